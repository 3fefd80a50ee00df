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
constexpr int MAXMT = 4;       // up to 64 batch rows per workgroup
constexpr int CHUNK = 4;       // k-quads fetched per prefetch group

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

// acc[mt][nt] += AT[k][rows] * W[k][cols] over k in [kbeg,kend) (multiple of 4).
template <int MT, int NTL>
__device__ __forceinline__ void kslice_mfma(const float *__restrict__ AT, int ldA, const float *__restrict__ W,
                                            int ldW, int row0, int col0, int kbeg, int kend, int lane,
                                            f32x4 (&acc)[MT][NTL])
{
    const int li = lane & 15, lk = lane >> 4;
    const float *ap = AT + (size_t)(kbeg + lk) * ldA + row0 + li;
    const float *wp = W + (size_t)(kbeg + lk) * ldW + col0 + li;
    const size_t astep = (size_t)4 * ldA, wstep = (size_t)4 * ldW;
    float a[CHUNK][MT], w[CHUNK][NTL];
    int k = kbeg;
    // full chunks, software-prefetched one chunk ahead
    const int nfull = (kend - kbeg) / (4 * CHUNK);
    if (nfull > 0) {
#pragma unroll
        for (int c = 0; c < CHUNK; ++c) {
#pragma unroll
            for (int m = 0; m < MT; ++m) a[c][m] = ap[c * astep + m * 16];
#pragma unroll
            for (int n = 0; n < NTL; ++n) w[c][n] = wp[c * wstep + n * 16];
        }
        for (int it = 0; it < nfull; ++it) {
            float a2[CHUNK][MT], w2[CHUNK][NTL];
            ap += CHUNK * astep; wp += CHUNK * wstep;
            if (it + 1 < nfull) {
#pragma unroll
                for (int c = 0; c < CHUNK; ++c) {
#pragma unroll
                    for (int m = 0; m < MT; ++m) a2[c][m] = ap[c * astep + m * 16];
#pragma unroll
                    for (int n = 0; n < NTL; ++n) w2[c][n] = wp[c * wstep + n * 16];
                }
            }
#pragma unroll
            for (int c = 0; c < CHUNK; ++c)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NTL; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][m], w[c][n], acc[m][n], 0, 0, 0);
            if (it + 1 < nfull) {
#pragma unroll
                for (int c = 0; c < CHUNK; ++c) {
#pragma unroll
                    for (int m = 0; m < MT; ++m) a[c][m] = a2[c][m];
#pragma unroll
                    for (int n = 0; n < NTL; ++n) w[c][n] = w2[c][n];
                }
            }
        }
        k += nfull * 4 * CHUNK;
    }
    for (; k < kend; k += 4) {   // tail quads
        float at[MT], wt[NTL];
#pragma unroll
        for (int m = 0; m < MT; ++m) at[m] = ap[m * 16];
#pragma unroll
        for (int n = 0; n < NTL; ++n) wt[n] = wp[n * 16];
        ap += astep; wp += wstep;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NTL; ++n)
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(at[m], wt[n], acc[m][n], 0, 0, 0);
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
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *hTprev = d.hT + (size_t)((p.step + 1) & 1) * N * p.Bpad;
    float *hTnext = d.hT + (size_t)(p.step & 1) * N * p.Bpad;

    f32x4 acc[MT][NTL];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NTL; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (!first) {
        const int kq = (N / 4 + NWAVES - 1) / NWAVES * 4;           // K slice per wave, multiple of 4
        const int kbeg = min(wave * kq, N), kend = min(kbeg + kq, N);
        kslice_mfma<MT, NTL>(hTprev, p.Bpad, d.R, G, row0, blk * 32, kbeg, kend, lane, acc);
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
            hTnext[(size_t)n * p.Bpad + b] = 0.f;
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
        hTnext[(size_t)n * p.Bpad + b] = h;
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
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *dzTprev = d.dzT + (size_t)((p.step + 1) & 1) * G * p.Bpad;
    float *dzTnext = d.dzT + (size_t)(p.step & 1) * G * p.Bpad;

    f32x4 acc[MT][NTL];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (!first) {
        const int kq = (G / 4 + NWAVES - 1) / NWAVES * 4;
        const int kbeg = min(wave * kq, G), kend = min(kbeg + kq, G);
        kslice_mfma<MT, NTL>(dzTprev, p.Bpad, d.RT, N, row0, n0, kbeg, kend, lane, acc);
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
            for (int g = 0; g < 4; ++g) dzTnext[(size_t)(cbase + g * 8) * p.Bpad + b] = 0.f;
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
        dzTnext[(size_t)(cbase + 0) * p.Bpad + b] = di_pre;
        dzTnext[(size_t)(cbase + 8) * p.Bpad + b] = dj_pre;
        dzTnext[(size_t)(cbase + 16) * p.Bpad + b] = df_pre;
        dzTnext[(size_t)(cbase + 24) * p.Bpad + b] = do_pre;
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
inline int bpad(int B) { return B <= 64 ? ((B + 15) & ~15) : ((B + 63) & ~63); }

}  // namespace

extern "C" size_t lc_lstm_fwd_workspace_bytes(int B, int N, int ndir)
{
    return (size_t)ndir * al256((size_t)2 * N * bpad(B) * sizeof(float));
}
extern "C" size_t lc_lstm_bwd_workspace_bytes(int B, int N, int ndir)
{
    return (size_t)ndir * (al256((size_t)2 * 4 * N * bpad(B) * sizeof(float)) + al256((size_t)B * N * sizeof(float)));
}

extern "C" int lc_lstm_fwd(const lc_lstm_fwd_dir_t *dirs, int ndir, const int *seq_len, int T, int B, int N,
                           float forget_bias, void *workspace, size_t workspace_bytes, lc_stream_t stream)
{
    LC_CHECK_ARG(dirs && seq_len && workspace, "lc_lstm_fwd: null pointer");
    LC_CHECK_ARG(ndir == 1 || ndir == 2, "lc_lstm_fwd: ndir must be 1 or 2");
    LC_CHECK_ARG(T > 0 && B > 0 && N > 0 && N % 8 == 0, "lc_lstm_fwd: need T,B > 0 and num_neurons %% 8 == 0 (N=%d)", N);
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
        a.d[i].zx = dirs[i].zx; a.d[i].R = dirs[i].R;
        a.d[i].w_f = dirs[i].w_f; a.d[i].w_i = dirs[i].w_i; a.d[i].w_o = dirs[i].w_o;
        a.d[i].cs = dirs[i].cs; a.d[i].hs = dirs[i].hs; a.d[i].reverse = dirs[i].reverse;
        a.d[i].hT = (float *)w;
        w += al256((size_t)2 * N * a.Bpad * sizeof(float));
    }
    if (ndir == 1) a.d[1] = a.d[0];
    // pad rows of hT (b >= B) are never written by the kernel but are read as MFMA operands
    if (hipMemsetAsync(workspace, 0, lc_lstm_fwd_workspace_bytes(B, N, ndir), s) != hipSuccess) {
        lc_set_error("lc_lstm_fwd: memset failed");
        return LC_ELAUNCH;
    }
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
        a.d[i].gates = dirs[i].gates; a.d[i].RT = dirs[i].RT;
        a.d[i].w_f = dirs[i].w_f; a.d[i].w_i = dirs[i].w_i; a.d[i].w_o = dirs[i].w_o;
        a.d[i].cs = dirs[i].cs; a.d[i].dh = dirs[i].dh; a.d[i].reverse = dirs[i].reverse;
        a.d[i].dzT = (float *)w; w += al256((size_t)2 * 4 * N * a.Bpad * sizeof(float));
        a.d[i].dc = (float *)w; w += al256((size_t)B * N * sizeof(float));
    }
    if (ndir == 1) a.d[1] = a.d[0];
    if (hipMemsetAsync(workspace, 0, lc_lstm_bwd_workspace_bytes(B, N, ndir), s) != hipSuccess) {
        lc_set_error("lc_lstm_bwd: memset failed");
        return LC_ELAUNCH;
    }
    const int mt = a.Bpad >= 64 ? 4 : a.Bpad / 16;
    dim3 grid(N / 16, lc_cdiv(B, 16 * mt), ndir), block(NTHREADS);
    for (int step = 0; step < T; ++step) {
        a.step = step;
        switch (mt) {
        case 1: hipLaunchKernelGGL(lstm_bwd_step_kernel<1>, grid, block, 0, s, a); break;
        case 2: hipLaunchKernelGGL(lstm_bwd_step_kernel<2>, grid, block, 0, s, a); break;
        case 3: hipLaunchKernelGGL(lstm_bwd_step_kernel<3>, grid, block, 0, s, a); break;
        default: hipLaunchKernelGGL(lstm_bwd_step_kernel<4>, grid, block, 0, s, a); break;
        }
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
