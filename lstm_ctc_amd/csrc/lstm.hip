// lstm.hip — the sequential part of the peephole/projection LSTM (forward and BPTT) for gfx950.
//
// Replaces the tf.while_loop that tf.nn.dynamic_rnn builds around tf.contrib.rnn.LSTMCell
// (mobvoi/lstm_ctc nnet/bilstm.py:125-188; semantics SURVEY.md App. A.1/A.2) and its gradient.
//
// What is sequential and what is not.  With x_t.Kx hoisted over all T (one big GEMM) and the
// projection folded into the recurrent weights (R = proj.Kh, so the state that recurs is the
// pre-projection output m'), one step is
//     z_t = zx_t + m'_{t-1} . R          [B,N] x [N,4N]      <- the only dependent GEMM
//     gates, c_t, m'_t                     elementwise on [B,N]
// and one BPTT step is   dm'_{rec} = dz_{t'} . R^T   [B,4N] x [4N,N]  followed by the gate
// derivatives.  Everything else (projection, input/weight gradients) is batched over T.
//
// Kernel: one launch per time step covering BOTH directions (blockIdx.z), 256 threads = 4 waves.
// A workgroup owns a 16*NTL-column slice of the step GEMM for up to 64 batch rows; the K dimension
// is split across the 4 waves (v_mfma_f32_16x16x4_f32, exact f32), partial tiles meet in LDS, and
// the epilogue applies the gate math for the units the slice covers.  Both GEMM operands are kept
// K-MAJOR in memory (the previous step's m'/dz is also written transposed, [K][Bpad]), so every
// MFMA fragment load is a run of 16 consecutive floats straight from L2 - no LDS staging.
//
// Column layout ("gate-interleaved"): column c' = (n/8)*32 + g*8 + (n%8) for gate g of unit n, so
// the 4 gates of 8 consecutive units are one 128-byte run.  The host keeps kernels/biases in this
// layout permanently (checkpoint I/O converts to TF's [i|j|f|o] blocks).
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int NTHREADS = 256;
constexpr int NWAVES = 4;

struct DirFwd {
    float *zx;          // [T,B,4N] in: pre-activations (x part + bias); out: activated gates
    const float *R;     // [N,4N]
    const float *w_f, *w_i, *w_o;
    float *cs, *hs;     // [T,B,N]
    float *hT;          // [2][N][Bpad] transposed m' ping-pong
    int reverse;
};
struct FwdArgs {
    DirFwd d[2];
    const int *seq_len;
    int T, B, N, Bpad, step;
    float forget_bias;
};

struct DirBwd {
    float *gates;       // [T,B,4N] in: activated gates; out: dz
    const float *RT;    // [4N,N]
    const float *w_f, *w_i, *w_o;
    const float *cs;    // [T,B,N]
    const float *dh;    // [T,B,N] gradient w.r.t. m'_t from the layer output
    float *dc;          // [B,N] carried cell gradient
    float *dzT;         // [2][4N][Bpad] transposed dz ping-pong
    int reverse;
};
struct BwdArgs {
    DirBwd d[2];
    const int *seq_len;
    int T, B, N, Bpad, step;
};

// K16 operand layout.  Both step-GEMM operands are stored so that ONE 16-byte load per lane feeds FOUR MFMAs:
// element (k, c) of a [K, C] operand lives at  ((k>>4)*4 + (k&3)) * C*4 + c*4 + ((k>>2)&3), i.e. [K/16][lk][C][q]
// with k = 16*blk + 4*q + lk.  MFMA 16x16x4 k-slot lk of "quad q" then covers k = 16*blk + 4*q + lk for both
// operands, and lane (li = lane&15, lk = lane>>4) fetches its four quads' values as one float4.
__host__ __device__ inline size_t k16_index(int k, int c, int C)
{
    return ((size_t)((k >> 4) * 4 + (k & 3)) * C + c) * 4 + ((k >> 2) & 3);
}

// acc[mt][nt] += sum_k A[k][row0 + ..] * W[k][col0 + ..] over the 16-blocks [bbeg, bend) (wave-uniform).
// Two register buffers of CB 16-blocks alternate: while one feeds the MFMAs the other is refilled from L2.
// sched_barrier pins the issue order load -> mma -> load -> mma (the scheduler otherwise sinks the loads
// next to their use and the prefetch distance collapses); both mma's are unconditional so no load can be
// sunk into a branch; refill indices are clamped (a redundant re-load at the tail) to stay branch-free.
constexpr int NBUF = 4;
template <int MT, int NTL>
struct Frag {
    float4 a[MT], w[NTL];
    __device__ __forceinline__ void load(const float *__restrict__ ap, const float *__restrict__ wp, size_t ablk,
                                         size_t wblk, int blk)
    {
#pragma unroll
        for (int m = 0; m < MT; ++m) a[m] = *reinterpret_cast<const float4 *>(ap + (size_t)blk * ablk + m * 64);
#pragma unroll
        for (int n = 0; n < NTL; ++n) w[n] = *reinterpret_cast<const float4 *>(wp + (size_t)blk * wblk + n * 64);
    }
    __device__ __forceinline__ void mma(f32x4 (&acc)[MT][NTL]) const
    {
        // quad-major order: consecutive MFMAs hit different accumulators (dependent-accumulator latency of
        // v_mfma_f32_16x16x4_f32 is 40 cycles vs 32 issue)
#define LC_QUAD(Q)                                                                                          \
        _Pragma("unroll") for (int m = 0; m < MT; ++m)                                                      \
            _Pragma("unroll") for (int n = 0; n < NTL; ++n)                                                 \
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].Q, w[n].Q, acc[m][n], 0, 0, 0);
        LC_QUAD(x) LC_QUAD(y) LC_QUAD(z) LC_QUAD(w)
#undef LC_QUAD
    }
};

// acc[mt][nt] += sum_k A[k][row0 + ..] * W[k][col0 + ..] over the 16-blocks [bbeg, bend) (wave-uniform).
// A ring of NBUF register buffers (one 16-block each) keeps NBUF-1 blocks of loads in flight under the
// MFMAs.  The main loop is straight-line code: sched_barrier pins the issue order (the scheduler otherwise
// sinks the loads next to their use and the prefetch distance collapses), refill indices are clamped (a
// redundant re-load at the tail) and the leftover < NBUF blocks run in a plain tail loop, so no load ever
// sits behind a branch and the compiler's counted vmcnt waits survive the back edge.
template <int MT, int NTL>
__device__ __forceinline__ void kslice_mfma(const float *__restrict__ A16, int ldA, const float *__restrict__ W16,
                                            int ldW, int row0, int col0, int bbeg, int bend, int lane,
                                            f32x4 (&acc)[MT][NTL])
{
    const int li = lane & 15, lk = lane >> 4;
    const size_t ablk = (size_t)16 * ldA, wblk = (size_t)16 * ldW;       // floats per 16-block
    const float *ap = A16 + (size_t)bbeg * ablk + ((size_t)lk * ldA + row0 + li) * 4;
    const float *wp = W16 + (size_t)bbeg * wblk + ((size_t)lk * ldW + col0 + li) * 4;
    const int nb = bend - bbeg;
    const int nmain = nb / NBUF * NBUF;          // blocks handled by the 4-buffer ring
    if (nmain > 0) {
        Frag<MT, NTL> f0, f1, f2, f3;
        static_assert(NBUF == 4, "ring is written out for 4 buffers");
        f0.load(ap, wp, ablk, wblk, 0);
        f1.load(ap, wp, ablk, wblk, min(1, nmain - 1));
        f2.load(ap, wp, ablk, wblk, min(2, nmain - 1));
#define LC_RING_STEP(FL, FM, OFF)                                              \
        FL.load(ap, wp, ablk, wblk, min(base + (OFF) + 3, nmain - 1));          \
        __builtin_amdgcn_sched_barrier(0);                                      \
        FM.mma(acc);                                                            \
        __builtin_amdgcn_sched_barrier(0);
#define LC_RING_ROUND(O) LC_RING_STEP(f3, f0, O) LC_RING_STEP(f0, f1, O + 1) LC_RING_STEP(f1, f2, O + 2) LC_RING_STEP(f2, f3, O + 3)
        int base = 0;
        // 16 blocks per iteration: the compiler drains vmcnt at every loop back edge, so long bodies matter
        for (; base + 16 <= nmain; base += 16) {
            LC_RING_ROUND(0) LC_RING_ROUND(4) LC_RING_ROUND(8) LC_RING_ROUND(12)
        }
        for (; base < nmain; base += NBUF) {
            LC_RING_ROUND(0)
        }
#undef LC_RING_ROUND
#undef LC_RING_STEP
    }
    for (int blk = nmain; blk < nb; ++blk) {   // leftover 16-blocks
        Frag<MT, NTL> t;
        t.load(ap, wp, ablk, wblk, blk);
        t.mma(acc);
    }
}

// [K, C] row-major -> K16 layout (once per call, off the sequential path)
__global__ __launch_bounds__(256) void pack_k16_kernel(const float *__restrict__ W, int K, int C, float *__restrict__ W16)
{
    const size_t total = (size_t)K * C;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i / C), c = (int)(i % C);
        W16[k16_index(k, c, C)] = W[i];
    }
}

// Writes this wave's partial tile into LDS: part[wave][row][col], row pitch LDP.
template <int MT, int NTL, int LDP>
__device__ __forceinline__ void spill_partial(float *part, int wave, int lane, const f32x4 (&acc)[MT][NTL])
{
    // 16x16 C layout: col = lane&15, row = (lane>>4)*4 + reg
    float *p = part + (size_t)wave * (MT * 16) * LDP;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NTL; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                p[(m * 16 + (lane >> 4) * 4 + r) * LDP + n * 16 + (lane & 15)] = acc[m][n][r];
}

// ------------------------------------------------------------------------------ forward step
// grid: (N/8, ceil(B/(16*MT)), ndir).  Each workgroup: 8 units x 4 gates = 32 columns.
template <int MT>
__global__ __launch_bounds__(NTHREADS) void lstm_fwd_step_kernel(FwdArgs p)
{
    constexpr int NTL = 2, LDP = 40;
    __shared__ float part[NWAVES * MT * 16 * LDP];
    const DirFwd &d = p.d[blockIdx.z];
    const int N = p.N, B = p.B, G = 4 * N;
    const int t = d.reverse ? (p.T - 1 - p.step) : p.step;
    const int tprev = d.reverse ? t + 1 : t - 1;
    const bool first = p.step == 0;
    const int blk = blockIdx.x, row0 = blockIdx.y * MT * 16;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float *hTprev = d.hT + (size_t)((p.step + 1) & 1) * N * p.Bpad;
    float *hTnext = d.hT + (size_t)(p.step & 1) * N * p.Bpad;

    f32x4 acc[MT][NTL];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NTL; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (!first) {
        const int nblk = N / 16, per = (nblk + NWAVES - 1) / NWAVES;          // 16-blocks of K per wave
        const int bbeg = min(wave * per, nblk), bend = min(bbeg + per, nblk);
        kslice_mfma<MT, NTL>(hTprev, p.Bpad, d.R, G, row0, blk * 32, bbeg, bend, lane, acc);
    }
    spill_partial<MT, NTL, LDP>(part, wave, lane, acc);
    __syncthreads();
    // epilogue: (row, unit) pairs
    for (int idx = threadIdx.x; idx < MT * 16 * 8; idx += NTHREADS) {
        const int i = idx & 7, r = idx >> 3, b = row0 + r;
        if (b >= B) continue;
        const int n = blk * 8 + i;
        float z[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < NWAVES; ++w) s += part[(w * MT * 16 + r) * LDP + g * 8 + i];
            z[g] = s;
        }
        float *zrow = d.zx + ((size_t)t * B + b) * G + blk * 32 + i;
        const size_t so = ((size_t)t * B + b) * N + n;
        if (t >= p.seq_len[b]) {   // dynamic_rnn: zero output; zero state stands in for "not started / frozen"
            zrow[0] = 0.f; zrow[8] = 0.f; zrow[16] = 0.f; zrow[24] = 0.f;
            d.cs[so] = 0.f; d.hs[so] = 0.f;
            hTnext[k16_index(n, b, p.Bpad)] = 0.f;
            continue;
        }
        const float cp = first ? 0.f : d.cs[((size_t)tprev * B + b) * N + n];
        const float zi = z[0] + zrow[0], zj = z[1] + zrow[8], zf = z[2] + zrow[16], zo = z[3] + zrow[24];
        const float ia = lc_sigmoid(zi + (d.w_i ? d.w_i[n] * cp : 0.f));
        const float fa = lc_sigmoid(zf + p.forget_bias + (d.w_f ? d.w_f[n] * cp : 0.f));
        const float ja = lc_tanh(zj);
        const float cn = fa * cp + ia * ja;
        const float oa = lc_sigmoid(zo + (d.w_o ? d.w_o[n] * cn : 0.f));
        const float h = oa * lc_tanh(cn);
        zrow[0] = ia; zrow[8] = ja; zrow[16] = fa; zrow[24] = oa;
        d.cs[so] = cn; d.hs[so] = h;
        hTnext[k16_index(n, b, p.Bpad)] = h;
    }
}

// ------------------------------------------------------------------------------ backward step
// grid: (N/16, ceil(B/(16*MT)), ndir).  Each workgroup: 16 units.
template <int MT>
__global__ __launch_bounds__(NTHREADS) void lstm_bwd_step_kernel(BwdArgs p)
{
    constexpr int NTL = 1, LDP = 17;
    __shared__ float part[NWAVES * MT * 16 * LDP];
    const DirBwd &d = p.d[blockIdx.z];
    const int N = p.N, B = p.B, G = 4 * N;
    // BPTT visits the steps in the opposite order of the forward recurrence
    const int t = d.reverse ? p.step : (p.T - 1 - p.step);
    const int tprev = d.reverse ? t + 1 : t - 1;      // the step whose state fed step t in the forward pass
    const bool has_prev = d.reverse ? (t + 1 < p.T) : (t > 0);
    const bool first = p.step == 0;
    const int n0 = blockIdx.x * 16, row0 = blockIdx.y * MT * 16;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float *dzTprev = d.dzT + (size_t)((p.step + 1) & 1) * G * p.Bpad;
    float *dzTnext = d.dzT + (size_t)(p.step & 1) * G * p.Bpad;

    f32x4 acc[MT][NTL];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (!first) {
        const int nblk = G / 16, per = (nblk + NWAVES - 1) / NWAVES;
        const int bbeg = min(wave * per, nblk), bend = min(bbeg + per, nblk);
        kslice_mfma<MT, NTL>(dzTprev, p.Bpad, d.RT, N, row0, n0, bbeg, bend, lane, acc);
    }
    spill_partial<MT, NTL, LDP>(part, wave, lane, acc);
    __syncthreads();
    for (int idx = threadIdx.x; idx < MT * 16 * 16; idx += NTHREADS) {
        const int i = idx & 15, r = idx >> 4, b = row0 + r;
        if (b >= B) continue;
        const int n = n0 + i;
        const int cbase = (n >> 3) * 32 + (n & 7);
        float *grow = d.gates + ((size_t)t * B + b) * G + cbase;
        if (t >= p.seq_len[b]) {   // masked step: no gradient, carries pass through (they are zero there)
            grow[0] = 0.f; grow[8] = 0.f; grow[16] = 0.f; grow[24] = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) dzTnext[k16_index(cbase + g * 8, b, p.Bpad)] = 0.f;
            continue;
        }
        float dh = d.dh[((size_t)t * B + b) * N + n];
#pragma unroll
        for (int w = 0; w < NWAVES; ++w) dh += part[(w * MT * 16 + r) * LDP + i];
        const float ia = grow[0], ja = grow[8], fa = grow[16], oa = grow[24];
        const float cn = d.cs[((size_t)t * B + b) * N + n];
        const float cp = has_prev ? d.cs[((size_t)tprev * B + b) * N + n] : 0.f;
        const float tc = lc_tanh(cn);
        const float do_pre = dh * tc * oa * (1.f - oa);
        float dcn = d.dc[(size_t)b * N + n] + dh * oa * (1.f - tc * tc);
        if (d.w_o) dcn += do_pre * d.w_o[n];
        const float di_pre = dcn * ja * ia * (1.f - ia);
        const float dj_pre = dcn * ia * (1.f - ja * ja);
        const float df_pre = dcn * cp * fa * (1.f - fa);
        float dcp = dcn * fa;
        if (d.w_i) dcp += di_pre * d.w_i[n];
        if (d.w_f) dcp += df_pre * d.w_f[n];
        d.dc[(size_t)b * N + n] = dcp;
        grow[0] = di_pre; grow[8] = dj_pre; grow[16] = df_pre; grow[24] = do_pre;
        dzTnext[k16_index(cbase + 0, b, p.Bpad)] = di_pre;
        dzTnext[k16_index(cbase + 8, b, p.Bpad)] = dj_pre;
        dzTnext[k16_index(cbase + 16, b, p.Bpad)] = df_pre;
        dzTnext[k16_index(cbase + 24, b, p.Bpad)] = do_pre;
    }
}

// Peephole gradients, batched over all frames (not on the sequential path):
//   dw_i[n] = sum_{t,b} dz_i * c_prev,  dw_f[n] = sum dz_f * c_prev,  dw_o[n] = sum dz_o * c_t
// grid: (N/64 rounded up, nsplit); atomics on [3][N].
__global__ __launch_bounds__(256) void peephole_grad_kernel(const float *__restrict__ dz, const float *__restrict__ cs,
                                                            int T, int B, int N, int reverse,
                                                            float *__restrict__ dpeep)
{
    __shared__ float red[3][4][64];
    const int n = blockIdx.x * 64 + (threadIdx.x & 63);
    const int sub = threadIdx.x >> 6;
    const int G = 4 * N;
    float ai = 0.f, af = 0.f, ao = 0.f;
    if (n < N) {
        const int cbase = (n >> 3) * 32 + (n & 7);
        const long long rows = (long long)T * B;
        for (long long row = blockIdx.y * 4 + sub; row < rows; row += (long long)gridDim.y * 4) {
            const int t = (int)(row / B);
            const float *g = dz + row * G + cbase;
            const float c = cs[row * N + n];
            const int tp = reverse ? t + 1 : t - 1;
            const float cp = (tp >= 0 && tp < T) ? cs[(row + (long long)(tp - t) * B) * N + n] : 0.f;
            ai += g[0] * cp; af += g[16] * cp; ao += g[24] * c;
        }
    }
    red[0][sub][threadIdx.x & 63] = ai; red[1][sub][threadIdx.x & 63] = af; red[2][sub][threadIdx.x & 63] = ao;
    __syncthreads();
    if (sub == 0 && n < N) {
        const int l = threadIdx.x;
        atomicAdd(&dpeep[0 * N + n], red[1][0][l] + red[1][1][l] + red[1][2][l] + red[1][3][l]);   // w_f
        atomicAdd(&dpeep[1 * N + n], red[0][0][l] + red[0][1][l] + red[0][2][l] + red[0][3][l]);   // w_i
        atomicAdd(&dpeep[2 * N + n], red[2][0][l] + red[2][1][l] + red[2][2][l] + red[2][3][l]);   // w_o
    }
}

inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
// rows of the transposed [K][Bpad] state buffers: covers every row a workgroup row-tile touches
inline int bpad(int B) { return B <= 16 ? 16 : (B <= 64 ? ((B + 31) & ~31) : ((B + 63) & ~63)); }

}  // namespace

extern "C" size_t lc_lstm_fwd_workspace_bytes(int B, int N, int ndir)
{
    return (size_t)ndir * (al256((size_t)2 * N * bpad(B) * sizeof(float)) + al256((size_t)N * 4 * N * sizeof(float)));
}
extern "C" size_t lc_lstm_bwd_workspace_bytes(int B, int N, int ndir)
{
    return (size_t)ndir * (al256((size_t)2 * 4 * N * bpad(B) * sizeof(float)) + al256((size_t)B * N * sizeof(float)) +
                           al256((size_t)N * 4 * N * sizeof(float)));
}

extern "C" int lc_lstm_fwd(const lc_lstm_fwd_dir_t *dirs, int ndir, const int *seq_len, int T, int B, int N,
                           float forget_bias, void *workspace, size_t workspace_bytes, lc_stream_t stream)
{
    LC_CHECK_ARG(dirs && seq_len && workspace, "lc_lstm_fwd: null pointer");
    LC_CHECK_ARG(ndir == 1 || ndir == 2, "lc_lstm_fwd: ndir must be 1 or 2");
    LC_CHECK_ARG(T > 0 && B > 0 && N > 0 && N % 16 == 0, "lc_lstm_fwd: need T,B > 0 and num_neurons %% 16 == 0 (N=%d)", N);
    if (workspace_bytes < lc_lstm_fwd_workspace_bytes(B, N, ndir)) {
        lc_set_error("lc_lstm_fwd: workspace too small");
        return LC_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    FwdArgs a;
    a.seq_len = seq_len; a.T = T; a.B = B; a.N = N; a.Bpad = bpad(B); a.forget_bias = forget_bias;
    char *w = (char *)workspace;
    for (int i = 0; i < ndir; ++i) {
        LC_CHECK_ARG(dirs[i].zx && dirs[i].R && dirs[i].cs && dirs[i].hs, "lc_lstm_fwd: null pointer in dirs[%d]", i);
        a.d[i].zx = dirs[i].zx;
        a.d[i].w_f = dirs[i].w_f; a.d[i].w_i = dirs[i].w_i; a.d[i].w_o = dirs[i].w_o;
        a.d[i].cs = dirs[i].cs; a.d[i].hs = dirs[i].hs; a.d[i].reverse = dirs[i].reverse;
        a.d[i].hT = (float *)w;
        const size_t hbytes = al256((size_t)2 * N * a.Bpad * sizeof(float));
        // pad rows of hT (b >= B) are never written by the kernel but are read as MFMA operands
        if (hipMemsetAsync(w, 0, hbytes, s) != hipSuccess) {
            lc_set_error("lc_lstm_fwd: memset failed");
            return LC_ELAUNCH;
        }
        w += hbytes;
        float *R16 = (float *)w;
        w += al256((size_t)N * 4 * N * sizeof(float));
        hipLaunchKernelGGL(pack_k16_kernel, dim3(1024), dim3(256), 0, s, dirs[i].R, N, 4 * N, R16);
        a.d[i].R = R16;
    }
    if (ndir == 1) a.d[1] = a.d[0];
    LC_CHECK_LAUNCH("pack_k16");
    const int mt = a.Bpad >= 64 ? 4 : a.Bpad / 16;
    dim3 grid(N / 8, lc_cdiv(B, 16 * mt), ndir), block(NTHREADS);
    for (int step = 0; step < T; ++step) {
        a.step = step;
        switch (mt) {
        case 1: hipLaunchKernelGGL(lstm_fwd_step_kernel<1>, grid, block, 0, s, a); break;
        case 2: hipLaunchKernelGGL(lstm_fwd_step_kernel<2>, grid, block, 0, s, a); break;
        case 3: hipLaunchKernelGGL(lstm_fwd_step_kernel<3>, grid, block, 0, s, a); break;
        default: hipLaunchKernelGGL(lstm_fwd_step_kernel<4>, grid, block, 0, s, a); break;
        }
    }
    LC_CHECK_LAUNCH("lstm_fwd_step");
    return LC_OK;
}

extern "C" int lc_lstm_bwd(const lc_lstm_bwd_dir_t *dirs, int ndir, const int *seq_len, int T, int B, int N,
                           void *workspace, size_t workspace_bytes, lc_stream_t stream)
{
    LC_CHECK_ARG(dirs && seq_len && workspace, "lc_lstm_bwd: null pointer");
    LC_CHECK_ARG(ndir == 1 || ndir == 2, "lc_lstm_bwd: ndir must be 1 or 2");
    LC_CHECK_ARG(T > 0 && B > 0 && N > 0 && N % 16 == 0, "lc_lstm_bwd: need T,B > 0 and num_neurons %% 16 == 0 (N=%d)", N);
    if (workspace_bytes < lc_lstm_bwd_workspace_bytes(B, N, ndir)) {
        lc_set_error("lc_lstm_bwd: workspace too small");
        return LC_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    BwdArgs a;
    a.seq_len = seq_len; a.T = T; a.B = B; a.N = N; a.Bpad = bpad(B);
    char *w = (char *)workspace;
    for (int i = 0; i < ndir; ++i) {
        LC_CHECK_ARG(dirs[i].gates && dirs[i].RT && dirs[i].cs && dirs[i].dh, "lc_lstm_bwd: null pointer in dirs[%d]", i);
        a.d[i].gates = dirs[i].gates;
        a.d[i].w_f = dirs[i].w_f; a.d[i].w_i = dirs[i].w_i; a.d[i].w_o = dirs[i].w_o;
        a.d[i].cs = dirs[i].cs; a.d[i].dh = dirs[i].dh; a.d[i].reverse = dirs[i].reverse;
        const size_t zbytes = al256((size_t)2 * 4 * N * a.Bpad * sizeof(float)) + al256((size_t)B * N * sizeof(float));
        if (hipMemsetAsync(w, 0, zbytes, s) != hipSuccess) {
            lc_set_error("lc_lstm_bwd: memset failed");
            return LC_ELAUNCH;
        }
        a.d[i].dzT = (float *)w; w += al256((size_t)2 * 4 * N * a.Bpad * sizeof(float));
        a.d[i].dc = (float *)w; w += al256((size_t)B * N * sizeof(float));
        float *RT16 = (float *)w;
        w += al256((size_t)N * 4 * N * sizeof(float));
        hipLaunchKernelGGL(pack_k16_kernel, dim3(1024), dim3(256), 0, s, dirs[i].RT, 4 * N, N, RT16);
        a.d[i].RT = RT16;
    }
    if (ndir == 1) a.d[1] = a.d[0];
    LC_CHECK_LAUNCH("pack_k16");
    // 32-row tiles: (N/16) x (B/32) x ndir workgroups of [32 x 16] outputs - 256 of them at N=1024, B=64
    const int mt = a.Bpad >= 32 ? 2 : 1;
    dim3 grid(N / 16, lc_cdiv(B, 16 * mt), ndir), block(NTHREADS);
    for (int step = 0; step < T; ++step) {
        a.step = step;
        if (mt == 1) hipLaunchKernelGGL(lstm_bwd_step_kernel<1>, grid, block, 0, s, a);
        else hipLaunchKernelGGL(lstm_bwd_step_kernel<2>, grid, block, 0, s, a);
    }
    LC_CHECK_LAUNCH("lstm_bwd_step");
    // peephole gradients (batched)
    for (int i = 0; i < ndir; ++i) {
        if (dirs[i].dpeep && dirs[i].w_f) {
            dim3 g2(lc_cdiv(N, 64), 64);
            hipLaunchKernelGGL(peephole_grad_kernel, g2, dim3(256), 0, s, dirs[i].gates, dirs[i].cs, T, B, N,
                               dirs[i].reverse, dirs[i].dpeep);
        }
    }
    LC_CHECK_LAUNCH("peephole_grad");
    return LC_OK;
}
