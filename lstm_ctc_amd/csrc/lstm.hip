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
// Schedules.  ONE persistent launch per call wherever the recurrent weights can stay in registers for all T steps: one XCD
// per (direction, 16-row group) for fp32 N <= 512 and bf16 N <= 1024 ("persistent recurrence" below), a PAIR of XCDs per
// (direction, 32-row group) with the step GEMM split along K for the fp32 1024-unit layers ("persistent recurrence over
// XCD pairs"); lstm_pair_x3.inc holds their split-operand (bf16x3) forms.  Every other shape, and the re-run of a step whose
// persistent launch reported a failure, takes the launch train:
// Kernel: one launch per time step covering BOTH directions (blockIdx.z) - for the big fp32 forward case one launch
// per direction and step, the two directions as independent chains on two streams - 256 threads = 4 waves.
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
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));

static unsigned long long *g_lstm_dbg = nullptr;
// 16-byte buffer accesses with explicit cache-policy bits (the XCD-pair schedule's system-scope hand-off)
typedef int x_i32x4 __attribute__((ext_vector_type(4)));
__device__ f32x4 x_buffer_load_b128(x_i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ void x_buffer_store_b128(f32x4 v, x_i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.v4f32");
typedef float x_f32x2v __attribute__((ext_vector_type(2)));
__device__ x_f32x2v x_buffer_load_b64(x_i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v2f32");
__device__ float x_buffer_load_b32(x_i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.f32");
__device__ void x_buffer_store_b32(float v, x_i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.f32");
__device__ void x_buffer_store_b64(x_f32x2v v, x_i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.v2f32");
__device__ void x_buffer_store_b16(short v, x_i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.i16");

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
    unsigned short *hs16;   // optional bf16 copy of hs (bf16 persistent kernel only; elsewhere a cast behind the recurrence)
};
struct FwdArgs {
    DirFwd d[2];
    const int *seq_len;
    int T, B, N, Bpad, step;
    int row_base;              // first batch row of this launch
    float forget_bias;
    unsigned long long *dbg;   // optional phase timestamps (s_memtime) of workgroup 0, for tools/probe.py
};
#define LC_STAMP(k)                                                                               \
    do {                                                                                          \
        if (p.dbg && blockIdx.x == 7 && blockIdx.y == 0 && blockIdx.z == 0 && lane == 0)          \
            p.dbg[(p.step * 4 + wave) * 8 + (k)] = __builtin_amdgcn_s_memtime();                  \
    } while (0)

struct DirBwd {
    float *gates;       // [T,B,4N] in: activated gates; out: dz
    const float *RT;    // [4N,N]
    const float *w_f, *w_i, *w_o;
    const float *cs;    // [T,B,N]
    const float *dh;    // [T,B,N] gradient w.r.t. m'_t from the layer output
    float *dc;          // [B,N] carried cell gradient
    float *dzT;         // [2][4N][Bpad] transposed dz ping-pong
    int reverse;
    unsigned short *dz16;   // optional bf16 copy of dz (bf16 persistent kernel only; elsewhere a cast behind the recurrence)
};
struct BwdArgs {
    DirBwd d[2];
    const int *seq_len;
    int T, B, N, Bpad, step;
    int row_base;
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
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// BF = bf16 operands (config c5): the same 16 bytes per lane are 8 consecutive-k bf16 values and feed ONE
// v_mfma_f32_16x16x32_bf16 (a "block" is then 32 k deep, layout [K/32][lk][C][8]; byte addressing is unchanged).
template <int MT, int NTL, bool BF>
struct Frag {
    float4 a[MT], w[NTL];
    __device__ __forceinline__ void load(const float *__restrict__ ap, const float *__restrict__ wp, unsigned oa,
                                         unsigned ow)
    {
#pragma unroll
        for (int m = 0; m < MT; ++m) a[m] = *reinterpret_cast<const float4 *>(ap + oa + m * 64);
#pragma unroll
        for (int n = 0; n < NTL; ++n) w[n] = *reinterpret_cast<const float4 *>(wp + ow + n * 64);
    }
    __device__ __forceinline__ void mma(f32x4 (&acc)[MT][NTL]) const
    {
        if constexpr (BF) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NTL; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[m]),
                                                                        __builtin_bit_cast(bf16x8, w[n]), acc[m][n],
                                                                        0, 0, 0);
        } else {
        // quad-major order: consecutive MFMAs hit different accumulators (dependent-accumulator latency of
        // v_mfma_f32_16x16x4_f32 is 40 cycles vs 32 issue)
#define LC_QUAD(Q)                                                                                          \
        _Pragma("unroll") for (int m = 0; m < MT; ++m)                                                      \
            _Pragma("unroll") for (int n = 0; n < NTL; ++n)                                                 \
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].Q, w[n].Q, acc[m][n], 0, 0, 0);
        LC_QUAD(x) LC_QUAD(y) LC_QUAD(z) LC_QUAD(w)
#undef LC_QUAD
        }
    }
};

// acc[mt][nt] += sum_k A[k][row0 + ..] * W[k][col0 + ..] over the 16-blocks [bbeg, bend) (wave-uniform).
// A ring of NBUF register buffers (one 16-block each) keeps NBUF-1 blocks of loads in flight under the
// MFMAs.  The main loop is straight-line code: sched_barrier pins the issue order (the scheduler otherwise
// sinks the loads next to their use and the prefetch distance collapses), refill indices are clamped (a
// redundant re-load at the tail) and the leftover < NBUF blocks run in a plain tail loop, so no load ever
// sits behind a branch and the compiler's counted vmcnt waits survive the back edge.
template <int MT, int NTL, bool BF, class Pre>
__device__ __forceinline__ void kslice_mfma(const float *__restrict__ A16, int ldA, const float *__restrict__ W16,
                                            int ldW, int row0, int col0, int bbeg, int bend, int rot, int lane,
                                            f32x4 (&acc)[MT][NTL], Pre &&issue_epilogue_loads)
{
    // The A operand (previous m' / dz) is the SAME 256 KB for every workgroup of a direction; if all of them
    // walk it in the same order every CU of an XCD asks the same L2 channel for the same line at the same time
    // (measured: 42 instead of 32 cycles per MFMA).  Each workgroup therefore starts its K walk at a different
    // 16-block (rot) and wraps around; the sum order differs per column slice but is fixed for a given shape.
    const int li = lane & 15, lk = lane >> 4;
    const unsigned ablk = 16u * ldA, wblk = 16u * ldW;       // floats per 16-block (operands are < 2^31 floats)
    const float *ap = A16 + (size_t)bbeg * ablk + ((size_t)lk * ldA + row0 + li) * 4;
    const float *wp = W16 + (size_t)bbeg * wblk + ((size_t)lk * ldW + col0 + li) * 4;
    const int nb = bend - bbeg;
    rot = nb > 0 ? rot % nb : 0;
    // Running physical block of the next refill, wrapping at nb: three scalar ops per step.  (A 64-bit
    // index*stride per load cost ~20 SALU per step that could not hide behind the matrix pipe.)
    int pb = rot;
    unsigned oa = (unsigned)pb * ablk, ow = (unsigned)pb * wblk;
    auto advance = [&]() {
        const bool wrap = (pb + 1 == nb);
        pb = wrap ? 0 : pb + 1;
        oa = wrap ? 0u : oa + ablk;
        ow = wrap ? 0u : ow + wblk;
    };
    const int nmain = nb / NBUF * NBUF;          // blocks handled by the 4-buffer ring
    if (nmain > 0) {
        Frag<MT, NTL, BF> f0, f1, f2, f3;
        constexpr int NM = (BF ? 1 : 4) * MT * NTL, NLD = MT + NTL, NPAIR = NM < NLD ? NM : NLD;
        static_assert(NBUF == 4, "ring is written out for 4 buffers");
        f0.load(ap, wp, oa, ow); advance();
        f1.load(ap, wp, oa, ow); advance();
        f2.load(ap, wp, oa, ow); advance();
        // vmcnt retires in order: issued ahead of the ring these (HBM-cold) loads would stall its first wait
        issue_epilogue_loads();
        // One scheduling region per step: the refill's loads are slotted one per MFMA gap (an MFMA occupies the
        // pipe for 32 cycles but issues in 4, so the loads ride along for free).  Refills past the end wrap
        // around to valid blocks (redundant, never consumed), which keeps the body branch-free.
#define LC_RING_STEP(FL, FM)                                                   \
        FL.load(ap, wp, oa, ow);                                                \
        advance();                                                              \
        FM.mma(acc);                                                            \
        _Pragma("unroll") for (int q_ = 0; q_ < NPAIR; ++q_) {                  \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                  \
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                  \
        }                                                                       \
        if constexpr (NM > NPAIR) __builtin_amdgcn_sched_group_barrier(0x008, NM - NPAIR, 0);  \
        if constexpr (NLD > NPAIR) __builtin_amdgcn_sched_group_barrier(0x020, NLD - NPAIR, 0); \
        __builtin_amdgcn_sched_barrier(0);
#define LC_RING_ROUND LC_RING_STEP(f3, f0) LC_RING_STEP(f0, f1) LC_RING_STEP(f1, f2) LC_RING_STEP(f2, f3)
        int base = 0;
        // 16 blocks per iteration: the compiler drains vmcnt at every loop back edge, so long bodies matter
        for (; base + 16 <= nmain; base += 16) {
            LC_RING_ROUND LC_RING_ROUND LC_RING_ROUND LC_RING_ROUND
        }
        for (; base < nmain; base += NBUF) {
            LC_RING_ROUND
        }
#undef LC_RING_ROUND
#undef LC_RING_STEP
    }
    if (nmain == 0) issue_epilogue_loads();
    for (int blk = nmain; blk < nb; ++blk) {   // leftover 16-blocks
        Frag<MT, NTL, BF> t;
        const int j = blk + rot, pj = j >= nb ? j - nb : j;
        t.load(ap, wp, (unsigned)pj * ablk, (unsigned)pj * wblk);
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

// bf16 operand layout [K/32][lk][C][8]: element (k, c) at (((k>>5)*4 + ((k>>3)&3)) * C + c) * 8 + (k&7)  (K % 32 == 0)
__host__ __device__ inline size_t k32_index(int k, int c, int C)
{
    return ((size_t)((k >> 5) * 4 + ((k >> 3) & 3)) * C + c) * 8 + (k & 7);
}
__device__ __forceinline__ unsigned short lc_bf16_bits(float x) { return __builtin_bit_cast(unsigned short, (__bf16)x); }

__global__ __launch_bounds__(256) void pack_k32_bf16_kernel(const float *__restrict__ W, int K, int C,
                                                            unsigned short *__restrict__ W32)
{
    const size_t total = (size_t)K * C;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i / C), c = (int)(i % C);
        W32[k32_index(k, c, C)] = lc_bf16_bits(W[i]);
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
template <int MT, bool BF>
__global__ __launch_bounds__(NTHREADS) void lstm_fwd_step_kernel(FwdArgs p)
{
    constexpr int NTL = 2, LDP = 40;
    __shared__ float part[NWAVES * MT * 16 * LDP];
    const DirFwd &d = p.d[blockIdx.z];
    const int N = p.N, B = p.B, G = 4 * N;
    const int t = d.reverse ? (p.T - 1 - p.step) : p.step;
    const int tprev = d.reverse ? t + 1 : t - 1;
    const bool first = p.step == 0;
    const int blk = blockIdx.x, row0 = p.row_base + blockIdx.y * MT * 16;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t hstride = (size_t)N * p.Bpad / (BF ? 2 : 1);          // floats per ping-pong half
    const float *hTprev = d.hT + (size_t)((p.step + 1) & 1) * hstride;
    float *hTnext = d.hT + (size_t)(p.step & 1) * hstride;
    auto store_h = [&](int n, int b, float v) {
        if constexpr (BF) reinterpret_cast<unsigned short *>(hTnext)[k32_index(n, b, p.Bpad)] = lc_bf16_bits(v);
        else hTnext[k16_index(n, b, p.Bpad)] = v;
    };
    LC_STAMP(0);

    // Epilogue operands do not depend on the step GEMM: fetch them now so their latency hides under it.
    constexpr int EPI = (MT * 16 * 8 + NTHREADS - 1) / NTHREADS;
    const int ei = threadIdx.x & 7, en = blk * 8 + ei;
    float pzx[EPI][4], pcp[EPI];
    bool pin[EPI];
    int plen[EPI];
    const float wi = d.w_i ? d.w_i[en] : 0.f, wf = d.w_f ? d.w_f[en] : 0.f, wo = d.w_o ? d.w_o[en] : 0.f;
    // All of these are issued unconditionally (row index clamped) so that nothing at kernel start waits for a
    // memory round trip: whether a row is active (t < seq_len) is only decided in the epilogue.
#pragma unroll
    for (int j = 0; j < EPI; ++j) {
        const int idx = threadIdx.x + j * NTHREADS, b = row0 + (idx >> 3);
        pin[j] = idx < MT * 16 * 8 && b < B;
        plen[j] = p.seq_len[min(b, B - 1)];
    }
    auto issue_epilogue_loads = [&]() {
#pragma unroll
        for (int j = 0; j < EPI; ++j) {
            const int idx = threadIdx.x + j * NTHREADS, b = min(row0 + (idx >> 3), B - 1);
            const float *zrow = d.zx + ((size_t)t * B + b) * G + blk * 32 + ei;
#pragma unroll
            for (int g = 0; g < 4; ++g) pzx[j][g] = zrow[8 * g];
            pcp[j] = first ? 0.f : d.cs[((size_t)tprev * B + b) * N + en];
        }
    };

    f32x4 acc[MT][NTL];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NTL; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (!first) {
        const int nblk = N / (BF ? 32 : 16), per = (nblk + NWAVES - 1) / NWAVES;          // K blocks per wave
        const int bbeg = min(wave * per, nblk), bend = min(bbeg + per, nblk);
        kslice_mfma<MT, NTL, BF>(hTprev, p.Bpad, d.R, G, row0, blk * 32, bbeg, bend, (int)(blockIdx.x >> 3), lane, acc,
                             issue_epilogue_loads);
    } else {
        issue_epilogue_loads();
    }
    LC_STAMP(1);
    spill_partial<MT, NTL, LDP>(part, wave, lane, acc);
    __syncthreads();
    LC_STAMP(2);
    // epilogue: (row, unit) pairs.  Branch-free arithmetic for all of a thread's pairs first (independent chains the
    // scheduler can interleave), masked rows selected to zero at the end (dynamic_rnn: zero output; zero state stands
    // in for "not started / frozen"), then the stores.
    float oia[EPI], oja[EPI], ofa[EPI], ooa[EPI], ocn[EPI], oh[EPI];
#pragma unroll
    for (int j = 0; j < EPI; ++j) {
        const int idx = threadIdx.x + j * NTHREADS, r = min(idx >> 3, MT * 16 - 1);
        float z[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float sacc = pzx[j][g];
#pragma unroll
            for (int w = 0; w < NWAVES; ++w) sacc += part[(w * MT * 16 + r) * LDP + g * 8 + ei];
            z[g] = sacc;
        }
        // explicit fma placement: every (row, unit) pair - whichever unrolled slot j it lands in - must go through
        // the same roundings, or identical utterances in different batch rows drift apart (the random-init
        // recurrence amplifies one ulp to 1e-3 within 1000 steps)
        const float cp = pcp[j];
        const float ia = lc_sigmoid(__builtin_fmaf(wi, cp, z[0]));
        const float fa = lc_sigmoid(__builtin_fmaf(wf, cp, z[2] + p.forget_bias));
        const float ja = lc_tanh(z[1]);
        const float cn = __builtin_fmaf(fa, cp, ia * ja);
        const float oa = lc_sigmoid(__builtin_fmaf(wo, cn, z[3]));
        const float h = oa * lc_tanh(cn);
        const bool act = t < plen[j];
        oia[j] = act ? ia : 0.f; oja[j] = act ? ja : 0.f; ofa[j] = act ? fa : 0.f; ooa[j] = act ? oa : 0.f;
        ocn[j] = act ? cn : 0.f; oh[j] = act ? h : 0.f;
    }
#pragma unroll
    for (int j = 0; j < EPI; ++j) {
        if (!pin[j]) continue;
        const int idx = threadIdx.x + j * NTHREADS, r = idx >> 3, b = row0 + r;
        float *zrow = d.zx + ((size_t)t * B + b) * G + blk * 32 + ei;
        const size_t so = ((size_t)t * B + b) * N + en;
        zrow[0] = oia[j]; zrow[8] = oja[j]; zrow[16] = ofa[j]; zrow[24] = ooa[j];
        d.cs[so] = ocn[j]; d.hs[so] = oh[j];
        store_h(en, b, oh[j]);
    }
    LC_STAMP(3);
}

// ------------------------------------------------------------------------------ backward step
// grid: (N/16, ceil(B/(16*MT)), ndir).  Each workgroup: 16 units.
template <int MT, bool BF>
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
    const int n0 = blockIdx.x * 16, row0 = p.row_base + blockIdx.y * MT * 16;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t zstride = (size_t)G * p.Bpad / (BF ? 2 : 1);          // floats per ping-pong half
    const float *dzTprev = d.dzT + (size_t)((p.step + 1) & 1) * zstride;
    float *dzTnext = d.dzT + (size_t)(p.step & 1) * zstride;
    auto store_dz = [&](int c, int b, float v) {
        if constexpr (BF) reinterpret_cast<unsigned short *>(dzTnext)[k32_index(c, b, p.Bpad)] = lc_bf16_bits(v);
        else dzTnext[k16_index(c, b, p.Bpad)] = v;
    };

    // Epilogue operands do not depend on the step GEMM: fetch them now so their latency hides under it.
    constexpr int EPI = (MT * 16 * 16 + NTHREADS - 1) / NTHREADS;
    const int ei = threadIdx.x & 15, en = n0 + ei;
    const int cbase = (en >> 3) * 32 + (en & 7);
    float pg[EPI][4], pdh[EPI], pcn[EPI], pcp[EPI], pdc[EPI];
    bool pin[EPI];
    int plen[EPI];
    const float wi = d.w_i ? d.w_i[en] : 0.f, wf = d.w_f ? d.w_f[en] : 0.f, wo = d.w_o ? d.w_o[en] : 0.f;
#pragma unroll
    for (int j = 0; j < EPI; ++j) {
        const int idx = threadIdx.x + j * NTHREADS, b = row0 + (idx >> 4);
        pin[j] = idx < MT * 16 * 16 && b < B;
        plen[j] = p.seq_len[min(b, B - 1)];
    }
    auto issue_epilogue_loads = [&]() {   // unconditional, row index clamped: see the forward kernel
#pragma unroll
        for (int j = 0; j < EPI; ++j) {
            const int idx = threadIdx.x + j * NTHREADS, b = min(row0 + (idx >> 4), B - 1);
            const float *grow = d.gates + ((size_t)t * B + b) * G + cbase;
#pragma unroll
            for (int g = 0; g < 4; ++g) pg[j][g] = grow[8 * g];
            pdh[j] = d.dh[((size_t)t * B + b) * N + en];
            pcn[j] = d.cs[((size_t)t * B + b) * N + en];
            pcp[j] = has_prev ? d.cs[((size_t)tprev * B + b) * N + en] : 0.f;
            pdc[j] = d.dc[(size_t)b * N + en];
        }
    };

    f32x4 acc[MT][NTL];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (!first) {
        const int nblk = G / (BF ? 32 : 16), per = (nblk + NWAVES - 1) / NWAVES;
        const int bbeg = min(wave * per, nblk), bend = min(bbeg + per, nblk);
        kslice_mfma<MT, NTL, BF>(dzTprev, p.Bpad, d.RT, N, row0, n0, bbeg, bend, (int)(blockIdx.x >> 3) * 5,
                             lane, acc, issue_epilogue_loads);
    } else {
        issue_epilogue_loads();
    }
    spill_partial<MT, NTL, LDP>(part, wave, lane, acc);
    __syncthreads();
    // branch-free arithmetic first, masked rows (no gradient; the carried dc passes through) selected at the end
    float odi[EPI], odj[EPI], odf[EPI], odo[EPI], odc[EPI];
#pragma unroll
    for (int j = 0; j < EPI; ++j) {
        const int idx = threadIdx.x + j * NTHREADS, r = min(idx >> 4, MT * 16 - 1);
        float dh = pdh[j];
#pragma unroll
        for (int w = 0; w < NWAVES; ++w) dh += part[(w * MT * 16 + r) * LDP + ei];
        const float ia = pg[j][0], ja = pg[j][1], fa = pg[j][2], oa = pg[j][3];
        const float cn = pcn[j], cp = pcp[j];
        const float tc = lc_tanh(cn);                       // explicit fma placement: see the forward kernel
        const float do_pre = dh * tc * oa * (1.f - oa);
        const float dcn = __builtin_fmaf(do_pre, wo, __builtin_fmaf(dh * oa, __builtin_fmaf(-tc, tc, 1.f), pdc[j]));
        const float di_pre = dcn * ja * ia * (1.f - ia);
        const float dj_pre = dcn * ia * __builtin_fmaf(-ja, ja, 1.f);
        const float df_pre = dcn * cp * fa * (1.f - fa);
        const bool act = t < plen[j];
        odc[j] = act ? __builtin_fmaf(df_pre, wf, __builtin_fmaf(di_pre, wi, dcn * fa)) : pdc[j];
        odi[j] = act ? di_pre : 0.f; odj[j] = act ? dj_pre : 0.f; odf[j] = act ? df_pre : 0.f; odo[j] = act ? do_pre : 0.f;
    }
#pragma unroll
    for (int j = 0; j < EPI; ++j) {
        if (!pin[j]) continue;
        const int idx = threadIdx.x + j * NTHREADS, r = idx >> 4, b = row0 + r;
        float *grow = d.gates + ((size_t)t * B + b) * G + cbase;
        d.dc[(size_t)b * N + en] = odc[j];
        grow[0] = odi[j]; grow[8] = odj[j]; grow[16] = odf[j]; grow[24] = odo[j];
        store_dz(cbase + 0, b, odi[j]);
        store_dz(cbase + 8, b, odj[j]);
        store_dz(cbase + 16, b, odf[j]);
        store_dz(cbase + 24, b, odo[j]);
    }
}

// Per-unit parameter gradients, batched over all frames (not on the sequential path), one pass over dz:
//   dbias[c]  += sum_{t,b} dz[t,b,c]                                  (the LSTM bias, all four gates)
//   dw_i[n]   += sum dz_i * c_prev,  dw_f[n] += sum dz_f * c_prev,  dw_o[n] += sum dz_o * c_t   (peepholes)
// Deterministic two-stage reduce (no float atomics): grid (N/64 rounded up, UPG_SPLITS) writes part[split][7][N],
// unit_param_fold_kernel adds the splits in index order into the (+=) outputs.  Either output may be NULL.
constexpr int UPG_SPLITS = 64;
__global__ __launch_bounds__(256) void unit_param_grad_kernel(const float *__restrict__ dz, const float *__restrict__ cs,
                                                              int T, int B, int N, int reverse, int want_peep,
                                                              float *__restrict__ part)
{
    __shared__ float red[7][4][64];
    const int n = blockIdx.x * 64 + (threadIdx.x & 63);
    const int sub = threadIdx.x >> 6;
    const int G = 4 * N;
    float ai = 0.f, af = 0.f, ao = 0.f, bi = 0.f, bj = 0.f, bf = 0.f, bo = 0.f;
    const int cbase = (min(n, N - 1) >> 3) * 32 + (min(n, N - 1) & 7);
    if (n < N) {
        const long long rows = (long long)T * B;
        for (long long row = blockIdx.y * 4 + sub; row < rows; row += (long long)gridDim.y * 4) {
            const int t = (int)(row / B);
            const float *g = dz + row * G + cbase;
            const float gi = g[0], gj = g[8], gf = g[16], go = g[24];
            bi += gi; bj += gj; bf += gf; bo += go;
            if (want_peep) {
                const float c = cs[row * N + n];
                const int tp = reverse ? t + 1 : t - 1;
                const float cp = (tp >= 0 && tp < T) ? cs[(row + (long long)(tp - t) * B) * N + n] : 0.f;
                ai += gi * cp; af += gf * cp; ao += go * c;
            }
        }
    }
    const int l = threadIdx.x & 63;
    red[0][sub][l] = ai; red[1][sub][l] = af; red[2][sub][l] = ao;
    red[3][sub][l] = bi; red[4][sub][l] = bj; red[5][sub][l] = bf; red[6][sub][l] = bo;
    __syncthreads();
    if (sub == 0 && n < N) {
        float *o = part + (size_t)blockIdx.y * 7 * N + n;
#pragma unroll
        for (int k = 0; k < 7; ++k) o[(size_t)k * N] = (red[k][0][l] + red[k][1][l]) + (red[k][2][l] + red[k][3][l]);
    }
}
__global__ __launch_bounds__(256) void unit_param_fold_kernel(const float *__restrict__ part, int nsplit, int N,
                                                              float *__restrict__ dpeep, float *__restrict__ dbias)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float a[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int sp = 0; sp < nsplit; ++sp)
#pragma unroll
        for (int k = 0; k < 7; ++k) a[k] += part[((size_t)sp * 7 + k) * N + n];
    if (dpeep) {
        dpeep[0 * N + n] += a[1];   // w_f
        dpeep[1 * N + n] += a[0];   // w_i
        dpeep[2 * N + n] += a[2];   // w_o
    }
    if (dbias) {
        const int cbase = (n >> 3) * 32 + (n & 7);
        dbias[cbase + 0] += a[3]; dbias[cbase + 8] += a[4]; dbias[cbase + 16] += a[5]; dbias[cbase + 24] += a[6];
    }
}
// ------------------------------------------------------------------------------ persistent recurrence (small models)
// For N <= 512 (the reference's own recipes train 320-unit layers: egs/wsj/run_wsj_phn.sh:17) a step GEMM is a
// fraction of a microsecond and the per-step launch train above is bound by the kernel boundary (4.5 / 5.7 us per
// forward / backward step at N = 320).  Utterances are independent through the recurrence, so the batch is cut
// into row groups of <= 16 rows and each (direction, row group) is given to ONE XCD: the whole recurrence of that
// group runs inside one launch, the (up to 32) workgroups of the XCD each keep their column slice of R (R^T)
// resident in REGISTERS for all T steps, like their (row, unit) cell state / cell gradient, and the only thing
// that crosses workgroups per step is the [16, N] state (the [16, 4N] dz), exchanged through the XCD's own L2.
// There is no barrier and no flag - the exchanged data carry their own generation.  Forward: the state of four consecutive
// units of a row is one 16-byte fragment, gathered from a quad of producer threads and written by one lane in one plain
// store (it stays in this XCD's L2).  Backward (4x the data): the four gate derivatives of a (row, unit) are one 16-byte
// store.  Either way a fragment is exactly one consumer lane's MFMA operand of a block, and the generation bit of the step
// sits in the lowest mantissa bit of its values (the exchanged copy only).  A consumer wave requests its K slice with
// L1-bypassing loads and re-requests it until the first dword of every fragment shows the generation it needs.  Two buffers alternate: a workgroup
// can only be writing step s+1 after it has read every workgroup's step-s output, i.e. after every workgroup finished
// reading step s-1's.  No agent-scope cache maintenance is involved: producers and consumers share one L2.  A
// workgroup learns which XCD it runs on from HW_REG_XCC_ID (correctness never depends on the dispatcher's placement:
// the group IS the XCD the workgroup finds itself on); surplus workgroups exit at once.  Every spin is bounded: on a
// timeout the wave raises ctl->fail (from then on every wait of the launch ends at its first poll); the verify kernel
// launched behind every persistent launch then overwrites every output row of the call with NaN and sets the sticky
// status word the host reads at its next sync point (lstm_ctc_hip.h).  Measured (MI355X, us per step, forward / backward): N = 256 2.0 /
// 2.7, N = 320 2.5 / 3.1, N = 512 4.0 / 4.3 - launch train 3.9 / 4.9, 4.45 / 5.7, 5.3 / 7.2.  Tried on the way: an
// atomic arrival counter (device-scope atomics leave the XCD's L2: 1.2 us per barrier), a per-workgroup flag line
// (the s_waitcnt vmcnt(0) for the store acknowledgement alone is 0.85 us), a one-fragment probe ahead of the full
// request (a second serial round trip).  LC_LSTM_PERSISTENT=0 forces the launch train.
constexpr int P_THREADS = 256;
constexpr int P_GRID = 512;                 // 64 candidates per XCD; the first `nwg` of each claim a column slice
constexpr int P_MAXN = 512;
constexpr unsigned P_SPIN_LIMIT = 1u << 21; // polls per wait (LC_LSTM_SPIN_LIMIT overrides it, for tests)
constexpr unsigned P_ABORT_CHECK = 63;      // a spinning wave looks at ctl->fail every 64 polls
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct PCtl {                                // first LC_LSTM_STATUS_OFFSET bytes: zeroed by the host before every launch
    unsigned claim[8];                       // workgroups that took a slice, per XCD
    int fail;                                // a bounded spin ran out somewhere in THIS launch (peers stop when they see it)
    int pad_[7];
    int sticky;                              // at LC_LSTM_STATUS_OFFSET: never cleared by the library (lstm_ctc_hip.h)
};
static_assert(offsetof(PCtl, sticky) == LC_LSTM_STATUS_OFFSET, "status word offset is part of the C ABI");
constexpr size_t P_CTL_BYTES = 256;          // control block at the start of the workspace in EVERY schedule
// After every persistent launch (256 workgroups, one word read each when all is well): the launch is BAD if a wait ran
// out anywhere (ctl->fail) or if an XCD that should hold a group saw fewer than nwg workgroups (a placement that
// skips an XCC id would otherwise leave that group's rows unwritten without any wait ever timing out).  A bad launch
// sets the sticky status word and gets EVERY output row of the call overwritten with NaN - row groups on other XCDs
// may well have completed, but the caller is promised "all NaN", not "some rows stale".
struct PVerifyArgs {
    PCtl *ctl;
    int nused, nwg, nout;
    float *out[2];
    unsigned short *out16[2];                // optional bf16 shadow of the same output (hs_bf16 / dz_bf16) or NULL
    size_t count;                            // elements per output
    size_t count16;                          // halfwords per shadow: count (bf16 shadows) or 3 * count (the x3 shadow of dz)
};
__global__ __launch_bounds__(256) void persist_verify_kernel(PVerifyArgs a)
{
    bool bad = a.ctl->fail != 0;
    for (int x = 0; x < a.nused; ++x) bad |= a.ctl->claim[x] < (unsigned)a.nwg;
    if (!bad) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) a.ctl->sticky = 1;
    const float nan = __builtin_nanf("");
    for (int o = 0; o < a.nout; ++o)
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < a.count; i += (size_t)gridDim.x * blockDim.x)
            a.out[o][i] = nan;
    for (int o = 0; o < a.nout; ++o)         // bf16 NaN in every term: the shadow feeds the next GEMM directly
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; a.out16[o] && i < a.count16;
             i += (size_t)gridDim.x * blockDim.x)
            a.out16[o][i] = 0x7fc0;
}
struct PGeom {
    int T, B, N, ndir;
    int gpd;                                 // row groups per direction (8 / ndir)
    int rpg;                                 // batch rows per group (<= 16)
    int upw;                                 // units per workgroup (a multiple of 4)
    int UP;                                  // = upw (forward: 4 * UP local columns)
    int nwg;                                 // workgroups per XCD that hold a slice
};
struct PFwdArgs {
    DirFwd d[2];                             // hT unused
    const int *seq_len;
    PGeom g;
    float forget_bias;
    unsigned spin_limit;
    PCtl *ctl;
    float *hT;                               // [8 XCDs][2 buffers] of [N / 4][16 rows][4 units] state fragments (bf16 kernel: 4-byte granules)
    unsigned long long *dbg;                 // optional s_memtime stamps [T][8] of one workgroup (tools/persist_probe.py)
};
#define LC_PSTAMP(k)                                                                           \
    do {                                                                                       \
        if (p.dbg && xcc == 0 && slot == 0 && threadIdx.x == 0) p.dbg[step * 8 + (k)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
struct PBwdArgs {
    DirBwd d[2];                             // dc, dzT unused
    const int *seq_len;
    PGeom g;
    unsigned spin_limit;
    PCtl *ctl;
    float *dzT;                              // [8 XCDs][2][4N * 16], K order of the kernel comment
    float *upg[2];                           // per direction: [B batch rows][7][N] partial bias / peephole gradients, or NULL
    unsigned long long *dbg;
};

__device__ __forceinline__ int p_xcc_id()
{
    int v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}
// One unsuccessful poll: back off, then say whether to keep waiting - not beyond `limit` polls, and not once any
// workgroup of the launch has given up (ctl->fail; looked at on the first unsuccessful poll of a wait and every 64 polls
// after it: an L2 round trip of its own).  A wave whose wait failed keeps stepping through the time loop - its peers stop
// publishing, so every later wait of anybody ends at its first look at ctl->fail - rather than leaving it (a uniform
// exit needs an LDS flag and a test after the step barrier: measured +3-20 % per step on the latency-bound kernels).
#ifndef LC_P_BACKOFF
#define LC_P_BACKOFF 1                       // s_sleep units (64 cycles) between two looks of a wait (development knob)
#endif
#ifndef LC_P_FIRSTLOOK
#define LC_P_FIRSTLOOK 0                     // s_sleep units before the FIRST look of a step's first fetch (development knob)
#endif
// A wave-uniform pointer pinned to scalar registers.  The tensors' base pointers live in the kernel argument (`p.d[dirx]`, runtime
// index): under register pressure the compiler re-reads them with VECTOR loads where they are used, and the wait for such a
// pointer is a wait for every request in front of it in the in-order queue.
template <typename T>
using p_global = __attribute__((address_space(1))) T;      // (a pointer rebuilt from integers would otherwise be generic: flat_*)
template <typename T>
__device__ __forceinline__ p_global<T> *p_uniform(T *ptr)
{
    const unsigned long long b = (unsigned long long)ptr;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
    return (p_global<T> *)(((unsigned long long)hi << 32) | lo);
}
#ifndef LC_P_DEV_SKIP_SAVED
#define LC_P_DEV_SKIP_SAVED 0                // development (timing only, wrong results): the bf16 BPTT without its saved-tensor stores
#endif
#ifndef LC_P_DEV_SKIP_OPS
#define LC_P_DEV_SKIP_OPS 0                  // development (timing only, wrong results): the bf16 BPTT without its saved-operand loads
#endif
#ifndef LC_P_F32_AHEAD
#define LC_P_F32_AHEAD 3                     // fp32 single-XCD kernels: 1 = next step's operands behind the multiplies, 2 = behind the publication
#endif
#ifndef LC_P_OPS_AHEAD
#define LC_P_OPS_AHEAD 1                     // the bf16 BPTT requests a step's saved operands one step ahead (0: at the top of the step)
#endif
__device__ __forceinline__ void p_first_look_delay()
{
    if constexpr (LC_P_FIRSTLOOK > 0) __builtin_amdgcn_s_sleep(LC_P_FIRSTLOOK);
}
__device__ __forceinline__ bool p_keep_waiting(unsigned &n, unsigned limit, const PCtl *ctl)
{
    __builtin_amdgcn_s_sleep(LC_P_BACKOFF);
    if (++n > limit) return false;
    if ((n & P_ABORT_CHECK) == 1 && __hip_atomic_load(&ctl->fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
        return false;
    return true;
}
// A wait ran out: tell the peers (they stop within 64 polls) and the host (sticky status word).
__device__ __forceinline__ void p_report_failure(PCtl *ctl)
{
    __hip_atomic_store(&ctl->fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&ctl->sticky, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// buffer descriptor over [uniform_ptr, uniform_ptr + bytes): wave-uniform base, 32-bit per-lane / scalar offsets
__device__ __forceinline__ x_i32x4 x_rsrc(const void *uniform_ptr, unsigned bytes)
{
    const unsigned long long b = (unsigned long long)uniform_ptr;
    const x_i32x4 r = {__builtin_amdgcn_readfirstlane((int)(unsigned)b),
                       __builtin_amdgcn_readfirstlane((int)((b >> 32) & 0xffffu)),
                       __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000};
    return r;
}
__device__ __forceinline__ f32x4 p_load_nt(const float *p)
{
    return __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p));
}
// K-walk order of a wave: slot j of its NB register slots holds block p_blk(j) of the wave's nval blocks, rotated by
// the workgroup's slice number - all workgroups of an XCD read the SAME buffer at the same moment, and walking it in
// the same order sends every CU to the same L2 channel at once.
template <bool RAGGED>
__device__ __forceinline__ int p_blk(int j, int rot, int nval)
{
    const int r = j + rot;
    const int b = r >= nval ? r - nval : r;
    return (!RAGGED || j < nval) ? b : 0;
}
// Backward dz: a lane's fragment (16 bytes) is exactly one producer thread's 16-byte store - the four gate
// derivatives of one (row, unit).  Freshness travels in the data: the lowest mantissa bit of EACH of the four values is
// a generation bit (the exchanged copy only: <= 1 ulp on an operand of the recurrent product; the saved dz is exact).
// Two buffers alternate, so the value a slot held before this step's write is the one written two steps earlier - the
// generation bit gen(tag) = ((tag + 1) >> 1) & 1 flips between consecutive writes to the same slot and is 1 for the
// first write into either (zero-initialised) buffer.  Every dword CARRIES the bit, but consumers test only the first one
// (p_frag_stale below; state_stale in the XCD-pair kernels): the protocol therefore ASSUMES single-copy atomicity of an
// aligned 16-byte store against an aligned 16-byte load through one L2 - a dwordx4 access of a lane is one 16-byte
// piece of one 128-byte line transaction on gfx950, never split across requests.  A torn fragment would go unnoticed;
// the long-chain oracle tests (T = 1000, every exchange schedule) are what would show one.
__device__ __forceinline__ unsigned p_gen_bit(unsigned tag) { return ((tag + 1u) >> 1) & 1u; }
// A dz fragment is ONE 16-byte store of one producer thread, so its first dword tells whether the whole fragment is the
// generation asked for (the tags in the other three dwords are not looked at: seven VALU instructions per fragment instead
// of two, 8-11 % of a BPTT step at N = 320 / 512 - every one of them at its full issue cost).
__device__ __forceinline__ unsigned p_frag_stale(const f32x4 &v, unsigned gen) { return (__float_as_uint(v.x) ^ gen) & 1u; }
__device__ __forceinline__ f32x4 p_with_lsb_tag(float x, float y, float z, float w, unsigned gen)
{
    return (f32x4){__uint_as_float((__float_as_uint(x) & ~1u) | gen), __uint_as_float((__float_as_uint(y) & ~1u) | gen),
                   __uint_as_float((__float_as_uint(z) & ~1u) | gen), __uint_as_float((__float_as_uint(w) & ~1u) | gen)};
}
// Slots [LO, HI) of the wave's slice: request (unless PREISSUED: the caller already did, once) and vote until fresh.
template <int NB, int LO, int HI, bool RAGGED, bool PREISSUED>
__device__ __forceinline__ bool p_fetch_lsb(const float *blk0, int lk, int li, int nval, int rot, int rows, unsigned tag,
                                            unsigned limit, const PCtl *ctl, f32x4 (&a)[NB])
{
    unsigned n = 0;
    const float *base = blk0 + ((size_t)lk * 16 + li) * 4;
    bool issue = !PREISSUED;
    if (!PREISSUED) p_first_look_delay();
    for (;;) {
        asm volatile("" ::: "memory");
        if (issue) {
#pragma unroll
            for (int j = LO; j < HI; ++j) a[j] = p_load_nt(base + (size_t)p_blk<RAGGED>(j, rot, nval) * 256);
        }
        issue = true;
        unsigned stale = 0;                      // branch-free: one wait for all requests, one vote
#pragma unroll
        for (int j = LO; j < HI; ++j) stale |= (!RAGGED || j < nval) ? p_frag_stale(a[j], tag) : 0u;
        if (__builtin_amdgcn_ballot_w64(stale != 0 && li < rows) == 0) return true;
        if (!p_keep_waiting(n, limit, ctl)) return false;
    }
}
// acc += dz slots [LO, HI) x R^T fragments (registers); one accumulator per quad: no dependent back-to-back MFMAs.
template <int NB, int LO, int HI, bool RAGGED>
__device__ __forceinline__ void p_mma_bwd(const f32x4 (&a)[NB], const f32x4 (&w)[NB], int nval, f32x4 &acc0, f32x4 &acc1,
                                          f32x4 &acc2, f32x4 &acc3)
{
#pragma unroll
    for (int j = LO; j < HI; ++j) {
        const f32x4 aj = (!RAGGED || j < nval) ? a[j] : (f32x4){0.f, 0.f, 0.f, 0.f};
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(aj.x, w[j].x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(aj.y, w[j].y, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(aj.z, w[j].z, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(aj.w, w[j].w, acc3, 0, 0, 0);
    }
}

// grid: P_GRID x 1; dynamic LDS: partial tiles [4][16][4*UP] (the R slice lives in registers).
// PER = 16-blocks of K per wave (ceil(N / 64)); the column slice is NTILE = ceil(PER / 2) MFMA tiles wide.
template <int PER, bool RAGGED>
__global__ __launch_bounds__(P_THREADS) void lstm_fwd_persist_kernel(PFwdArgs p)
{
    constexpr int NTILE = (PER + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) float p_lds[];
    __shared__ int s_slot;
    const PGeom &g = p.g;
    const int xcc = p_xcc_id();
    if (xcc >= g.ndir * g.gpd) return;
    if (threadIdx.x == 0)
        s_slot = (int)__hip_atomic_fetch_add(&p.ctl->claim[xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int slot = s_slot;
    const int dirx = xcc / g.gpd, grp = xcc % g.gpd;
    const int rows_here = min(g.rpg, g.B - grp * g.rpg);
    if (slot >= g.nwg || rows_here <= 0) return;
    const DirFwd &d = p.d[dirx];
    const int N = g.N, G = 4 * N, B = g.B, T = g.T, UP = g.UP;
    constexpr int ncols = NTILE * 16;
    const int u0 = slot * g.upw, nu = min(g.upw, N - u0);
    float *part = p_lds;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, lk = lane >> 4;
    // exchange loads of a lane whose row does not exist in this group (li >= rows_here: B = 32 gives 8 rows per group, c1's
    // 4) go to row 0's fragment instead of their own: the same address as another lane of the instruction - no extra line -
    // and with 8 rows or fewer the second 128-byte line of every (block, lk) run is never fetched: half the L2 -> CU bytes of
    // a transfer-bound exchange (round 5; the rows' products are discarded as before)
    const int lir = li < rows_here ? li : 0;
    const int i = threadIdx.x >> 4, uu = threadIdx.x & 15;        // this thread's (row, unit) pair
    const bool valid = i < rows_here && uu < nu;
    const int b = min(grp * g.rpg + i, B - 1), n = min(u0 + uu, N - 1);
    const int len = valid ? p.seq_len[b] : 0;
    const float wi = d.w_i ? d.w_i[n] : 0.f, wf = d.w_f ? d.w_f[n] : 0.f, wo = d.w_o ? d.w_o[n] : 0.f;
    const int nkb = N / 16, per = (nkb + NWAVES - 1) / NWAVES;
    const int kb0 = min(wave * per, nkb), kb1 = min(kb0 + per, nkb);
    float *hTg = p.hT + (size_t)xcc * 2 * N * 16 * 2;
    const size_t zcol = (size_t)(n >> 3) * 32 + (n & 7);
    // State exchange (round 2, as in the XCD-pair kernel): fragment [n / 16][lk = (n >> 2) & 3][row] = the state of four
    // consecutive units of one row, 16 bytes = one consumer lane's MFMA operand of a block (k slot lk, quads = units 4 lk ..
    // 4 lk + 3), published by ONE lane in ONE store (the four producer threads are a quad: three DPP moves) with the
    // generation bit of the step in every value's lowest mantissa bit (the exchanged copy only: <= 1 ulp on an operand of
    // the recurrent product; everything saved is exact).  A consumer loads one fragment per block and checks one bit.  Round
    // 1's 8-byte {value, step} granules cost two loads, eight tag operations and four moves per block: ~100 cycles of issue
    // per block and step on a wave whose instructions all cost their full time.
    const size_t hfrag = ((size_t)((n >> 4) * 4 + ((n >> 2) & 3)) * 16 + i) * 4;
    float cprev = 0.f;
    bool failed = false;
    // This wave's K slice of the workgroup's columns of R, as MFMA fragments, stays in REGISTERS for the whole call
    // (one wave per SIMD owns 512 VGPRs per lane: PER * NTILE * 4 <= 128 of them): slot j holds block p_blk(j) of the
    // rotated K walk, columns c*16 + li = gate (c*16+li)/UP of unit u0 + (c*16+li)%UP; padding columns and the slots
    // past a ragged range hold zeros.  No LDS traffic for weights, and the LDS stays free for co-resident GEMMs.
    const int nval = kb1 - kb0, rot = nval > 0 ? slot % nval : 0;
    f32x4 wreg[PER][NTILE];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int kb = min(kb0 + p_blk<RAGGED>(j, rot, max(nval, 1)), nkb - 1);
#pragma unroll
        for (int c = 0; c < NTILE; ++c) {
            const int col = c * 16 + li, gate = col / UP, u2 = col - gate * UP, nn = min(u0 + u2, N - 1);
            const bool okc = u2 < nu && j < nval;
            const float *src = d.R + (size_t)(16 * kb + 4 * lk) * G + (nn >> 3) * 32 + gate * 8 + (nn & 7);   // k = 16 kb + 4 lk + q
            wreg[j][c] = (f32x4){okc ? src[0] : 0.f, okc ? src[(size_t)G] : 0.f, okc ? src[(size_t)2 * G] : 0.f,
                                 okc ? src[(size_t)3 * G] : 0.f};
        }
    }
    p_global<float> *const zx_p = p_uniform(d.zx), *const cs_p = p_uniform(d.cs), *const hs_p = p_uniform(d.hs);
    const bool rev = __builtin_amdgcn_readfirstlane((int)d.reverse) != 0;
    float nz[4];                             // LC_P_F32_AHEAD: the next step's gate pre-activations, requested a step ahead
    auto request_z = [&](int s) {
        const int t_ = rev ? (T - 1 - s) : s;
#pragma unroll
        for (int q = 0; q < 4; ++q) nz[q] = zx_p[((size_t)t_ * B + b) * G + zcol + 8 * q];
    };
    if (LC_P_F32_AHEAD) request_z(0);
    __syncthreads();
    for (int step = 0; step < T; ++step) {
        const int t = rev ? (T - 1 - step) : step;
        LC_PSTAMP(0);
        p_global<float> *zrow = zx_p + ((size_t)t * B + b) * G + zcol;
        float z[4];
        if (!LC_P_F32_AHEAD) {
#pragma unroll
            for (int q = 0; q < 4; ++q) z[q] = zrow[8 * q];             // does not depend on the recurrence
        }
        f32x4 acc[NTILE], acd[NTILE];          // two accumulators per tile: a dependent MFMA costs 40 cycles, an independent one 32
#pragma unroll
        for (int c = 0; c < NTILE; ++c) { acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f}; acd[c] = acc[c]; }
        if (step > 0 && kb0 < kb1) {
            // this wave's K slice of the previous state: fragments of generation `step` (written during step - 1)
            const float *hp = hTg + (size_t)((step + 1) & 1) * N * 16 + (size_t)kb0 * 256;
            f32x4 a[PER];
            if (!p_fetch_lsb<PER, 0, PER, RAGGED, false>(hp, lk, lir, nval, rot, rows_here, p_gen_bit((unsigned)step), p.spin_limit,
                                                         p.ctl, a)) failed = true;
            LC_PSTAMP(1);
            if (LC_P_F32_AHEAD == 3) {
#pragma unroll
                for (int q = 0; q < 4; ++q) asm volatile("v_mov_b32 %0, %1" : "=v"(z[q]) : "v"(nz[q]));
                request_z(min(step + 1, T - 1));
                // (no scheduling barrier here: the scheduler sinks the requests behind the multiplies, and pinned in front of them
                // the forward step is back at 2.20 us - profiles/r5_persist_probe_ahead.txt)
            }
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const f32x4 aj = (!RAGGED || j < nval) ? a[j] : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < NTILE; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(aj.x, wreg[j][c].x, acc[c], 0, 0, 0);
#pragma unroll
                for (int c = 0; c < NTILE; ++c) acd[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(aj.y, wreg[j][c].y, acd[c], 0, 0, 0);
#pragma unroll
                for (int c = 0; c < NTILE; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(aj.z, wreg[j][c].z, acc[c], 0, 0, 0);
#pragma unroll
                for (int c = 0; c < NTILE; ++c) acd[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(aj.w, wreg[j][c].w, acd[c], 0, 0, 0);
            }
#pragma unroll
            for (int c = 0; c < NTILE; ++c) acc[c] += acd[c];
        }
        LC_PSTAMP(2);
        if (LC_P_F32_AHEAD && (LC_P_F32_AHEAD != 3 || !(step > 0 && kb0 < kb1))) {   // (the copy is an instruction of its own: see the bf16 BPTT kernel)
#pragma unroll
            for (int q = 0; q < 4; ++q) asm volatile("v_mov_b32 %0, %1" : "=v"(z[q]) : "v"(nz[q]));
            if (LC_P_F32_AHEAD == 1 || LC_P_F32_AHEAD == 3) request_z(min(step + 1, T - 1));
        }
        // 16x16 C layout: col = lane & 15, row = (lane >> 4) * 4 + r
#pragma unroll
        for (int c = 0; c < NTILE; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) part[(size_t)(wave * 16 + lk * 4 + r) * ncols + c * 16 + li] = acc[c][r];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int w = 0; w < NWAVES; ++w) z[q] += part[(size_t)(w * 16 + i) * ncols + q * UP + min(uu, UP - 1)];
        const float ia = lc_sigmoid(__builtin_fmaf(wi, cprev, z[0]));
        const float fa = lc_sigmoid(__builtin_fmaf(wf, cprev, z[2] + p.forget_bias));
        const float ja = lc_tanh(z[1]);
        const float cn = __builtin_fmaf(fa, cprev, ia * ja);
        const float oa = lc_sigmoid(__builtin_fmaf(wo, cn, z[3]));
        const bool act = t < len;
        const float h = act ? oa * lc_tanh(cn) : 0.f;
        cprev = act ? cn : 0.f;
        // the state the other workgroups wait for goes out first, the saved activations after it
        {
            const int hv = (int)((__float_as_uint(h) & ~1u) | p_gen_bit((unsigned)step + 1u));
            const f32x4 frag = {__int_as_float(__builtin_amdgcn_mov_dpp(hv, 0x00, 0xf, 0xf, true)),
                                __int_as_float(__builtin_amdgcn_mov_dpp(hv, 0x55, 0xf, 0xf, true)),
                                __int_as_float(__builtin_amdgcn_mov_dpp(hv, 0xaa, 0xf, 0xf, true)),
                                __int_as_float(__builtin_amdgcn_mov_dpp(hv, 0xff, 0xf, 0xf, true))};
            if (valid && (uu & 3) == 0) *reinterpret_cast<f32x4 *>(hTg + (size_t)(step & 1) * N * 16 + hfrag) = frag;
        }
        LC_PSTAMP(3);
        if (LC_P_F32_AHEAD == 2) request_z(min(step + 1, T - 1));
        if (valid) {
            const size_t so = ((size_t)t * B + b) * N + n;
            zrow[0] = act ? ia : 0.f; zrow[8] = act ? ja : 0.f; zrow[16] = act ? fa : 0.f; zrow[24] = act ? oa : 0.f;
            cs_p[so] = cprev;
            hs_p[so] = h;
        }
        __syncthreads();                       // `part` is rewritten by the next step
        LC_PSTAMP(4);
    }
    if (failed && lane == 0) p_report_failure(p.ctl);            // persist_verify_kernel turns the outputs into NaN
}

// grid: P_GRID x 1; dynamic LDS: partial tiles [4][16][16] (the R^T slice lives in registers).
// NQ = ceil(16-blocks of K per wave / 4), K = 4N: this wave's whole slice of dz (4 * NQ fragments) sits in registers.
// K order of the exchange buffer: k = 16 * (n / 4) + 4 * gate + n % 4, so that the four gate derivatives of one
// (row, unit) are the four quad slots of ONE lane's fragment - a producer thread publishes its pair with one 16-byte
// store.  The rows of R^T are permuted to match while they are copied into LDS.
template <int NQ, bool RAGGED>
__global__ __launch_bounds__(P_THREADS) void lstm_bwd_persist_kernel(PBwdArgs p)
{
    extern __shared__ __attribute__((aligned(16))) float p_lds[];
    __shared__ int s_slot;
    const PGeom &g = p.g;
    const int xcc = p_xcc_id();
    if (xcc >= g.ndir * g.gpd) return;
    if (threadIdx.x == 0)
        s_slot = (int)__hip_atomic_fetch_add(&p.ctl->claim[xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int slot = s_slot;
    const int dirx = xcc / g.gpd, grp = xcc % g.gpd;
    const int rows_here = min(g.rpg, g.B - grp * g.rpg);
    if (slot >= g.nwg || rows_here <= 0) return;
    const DirBwd &d = p.d[dirx];
    const int N = g.N, G = 4 * N, B = g.B, T = g.T;
    const int u0 = slot * g.upw, nu = min(g.upw, N - u0);
    float *part = p_lds;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int lir = li < rows_here ? li : 0;      // exchange loads of rows that do not exist go to row 0's fragment (see the forward kernel)
    const int i = threadIdx.x >> 4, uu = threadIdx.x & 15;
    const bool valid = i < rows_here && uu < nu;
    const int b = min(grp * g.rpg + i, B - 1), n = min(u0 + uu, N - 1);
    const int len = valid ? p.seq_len[b] : 0;
    const float wi = d.w_i ? d.w_i[n] : 0.f, wf = d.w_f ? d.w_f[n] : 0.f, wo = d.w_o ? d.w_o[n] : 0.f;
    const int nkb = G / 16, per = (nkb + NWAVES - 1) / NWAVES;
    const int kb0 = min(wave * per, nkb), kb1 = min(kb0 + per, nkb);
    float *dzTg = p.dzT + (size_t)xcc * 2 * G * 16;
    const int cbase = (n >> 3) * 32 + (n & 7);
    const size_t pubidx = ((size_t)n * 16 + i) * 4;                // [(n/4)*4 + n%4][row][4 gates]
    float dc = 0.f;
    float ug[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // bias / peephole gradient sums of this (row, unit): see the XCD-pair BPTT
    bool failed = false;
    // This wave's K slice of the workgroup's columns of R^T, as MFMA fragments, in REGISTERS for the whole call (4 * NQ
    // <= 32 fragments = 128 VGPRs; see the forward kernel).  Exchange position k = 16*kb + 4*q + lk is gate q of unit
    // 4*kb + lk, i.e. row (n/8)*32 + q*8 + n%8 of R^T.
    const int nval = kb1 - kb0, rot = nval > 0 ? (slot * 5) % nval : 0;
    f32x4 wreg[4 * NQ];
#pragma unroll
    for (int j = 0; j < 4 * NQ; ++j) {
        const int kb = min(kb0 + p_blk<RAGGED>(j, rot, max(nval, 1)), nkb - 1);
        const int kn = 4 * kb + lk;
        const bool okc = li < nu && j < nval;
        const float *src = d.RT + (size_t)((kn >> 3) * 32 + (kn & 7)) * N + min(u0 + li, N - 1);
        wreg[j] = (f32x4){okc ? src[0] : 0.f, okc ? src[(size_t)8 * N] : 0.f, okc ? src[(size_t)16 * N] : 0.f,
                          okc ? src[(size_t)24 * N] : 0.f};
    }
    p_global<float> *const gates_p = p_uniform(d.gates);
    p_global<const float> *const dh_p = p_uniform(d.dh), *const cs_p = p_uniform(d.cs);
    const bool rev = __builtin_amdgcn_readfirstlane((int)d.reverse) != 0;
    float nop[7];                            // gates i j f o, dh, c, c_prev as requested (LC_P_F32_AHEAD: for the step after)
    auto request_operands = [&](int s) {     // 7 loads, all unconditional, nothing computed from them here
        const int t_ = rev ? s : (T - 1 - s);
        const int tp_ = rev ? min(t_ + 1, T - 1) : max(t_ - 1, 0);       // (no frame before the first: any valid address, zeroed at the use)
        p_global<const float> *gr = gates_p + ((size_t)t_ * B + b) * G + cbase;
        const size_t so_ = ((size_t)t_ * B + b) * N + n;
        nop[0] = gr[0]; nop[1] = gr[8]; nop[2] = gr[16]; nop[3] = gr[24];
        nop[4] = dh_p[so_]; nop[5] = cs_p[so_]; nop[6] = cs_p[((size_t)tp_ * B + b) * N + n];
    };
    if (LC_P_F32_AHEAD) request_operands(0);
    __syncthreads();
    for (int step = 0; step < T; ++step) {
        const int t = rev ? step : (T - 1 - step);
        const bool has_prev = rev ? (t + 1 < T) : (t > 0);
        LC_PSTAMP(0);
        p_global<float> *grow = gates_p + ((size_t)t * B + b) * G + cbase;
        if (!LC_P_F32_AHEAD) request_operands(step);
        float ia, ja, fa, oa, dh, cn, cp;
        auto take_operands = [&]() {         // (the copy is an instruction of its own: see the bf16 BPTT kernel)
            float cv[7];
#pragma unroll
            for (int q = 0; q < 7; ++q) asm volatile("v_mov_b32 %0, %1" : "=v"(cv[q]) : "v"(nop[q]));
            ia = cv[0]; ja = cv[1]; fa = cv[2]; oa = cv[3]; dh = cv[4]; cn = cv[5]; cp = has_prev ? cv[6] : 0.f;
        };
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;     // one per quad: no dependent back-to-back MFMAs
        if (step > 0 && kb0 < kb1) {
            // this wave's whole K slice of the previous dz (one wave per SIMD: the register file holds it), tagged
            // with the generation bit of `step`
            const float *ap = dzTg + (size_t)((step + 1) & 1) * G * 16 + (size_t)kb0 * 256;
            // Two phases: the wave polls only the first C0 slots of its slice (a half / a quarter of the polling traffic);
            // once they are fresh the rest is requested once and flies under the first slots' multiplies, then is checked
            // (all producers publish within a fraction of a microsecond of each other) and re-requested if need be.
            // (At NB = 32 the rest goes out in two requests of 12 slots: at most 24 fragments are live, which keeps the
            // kernel at <= 360 VGPRs - a GEMM wave (152) then still fits on the same SIMD for the weight-gradient overlap.)
            // (round 5, operands out of the queue: a first chunk of a quarter instead of a half at NB < 32 - no gain, 2.58 against 2.52 us)
            constexpr int NB = 4 * NQ, C0 = NB <= 8 ? NB : (NB >= 32 ? NB / 4 : NB / 2), C1 = NB >= 32 ? C0 + (NB - C0) / 2 : NB;
            f32x4 a[NB];
            const unsigned tag = p_gen_bit((unsigned)step);
            const float *base = ap + ((size_t)lk * 16 + lir) * 4;
            if (!p_fetch_lsb<NB, 0, C0, RAGGED, false>(ap, lk, lir, nval, rot, rows_here, tag, p.spin_limit, p.ctl, a)) failed = true;
            LC_PSTAMP(1);
            if constexpr (C0 < NB) {
#pragma unroll
                for (int j = C0; j < C1; ++j) a[j] = p_load_nt(base + (size_t)p_blk<RAGGED>(j, rot, nval) * 256);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (C0 == NB) { if (LC_P_F32_AHEAD == 3) { take_operands(); request_operands(min(step + 1, T - 1)); } }
            p_mma_bwd<NB, 0, C0, RAGGED>(a, wreg, nval, acc0, acc1, acc2, acc3);
            if constexpr (C0 < NB) {
                if (!p_fetch_lsb<NB, C0, C1, RAGGED, true>(ap, lk, lir, nval, rot, rows_here, tag, p.spin_limit, p.ctl, a)) failed = true;
                if constexpr (C1 < NB) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = C1; j < NB; ++j) a[j] = p_load_nt(base + (size_t)p_blk<RAGGED>(j, rot, nval) * 256);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (C1 == NB) { if (LC_P_F32_AHEAD == 3) { take_operands(); request_operands(min(step + 1, T - 1)); } }
                p_mma_bwd<NB, C0, C1, RAGGED>(a, wreg, nval, acc0, acc1, acc2, acc3);
                if constexpr (C1 < NB) {
                    if (!p_fetch_lsb<NB, C1, NB, RAGGED, true>(ap, lk, lir, nval, rot, rows_here, tag, p.spin_limit, p.ctl, a)) failed = true;
                    if (LC_P_F32_AHEAD == 3) { take_operands(); request_operands(min(step + 1, T - 1)); }
                    p_mma_bwd<NB, C1, NB, RAGGED>(a, wreg, nval, acc0, acc1, acc2, acc3);
                }
            }
        }
        LC_PSTAMP(2);
        if (LC_P_F32_AHEAD == 3 && step > 0 && kb0 < kb1) {
            // (taken and requested again in front of the last multiplies)
        } else if (LC_P_F32_AHEAD) {
            take_operands();
            if (LC_P_F32_AHEAD == 1 || LC_P_F32_AHEAD == 3) request_operands(min(step + 1, T - 1));
        } else {
            ia = nop[0]; ja = nop[1]; fa = nop[2]; oa = nop[3]; dh = nop[4]; cn = nop[5]; cp = has_prev ? nop[6] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) part[(wave * 16 + lk * 4 + r) * 16 + li] = (acc0[r] + acc1[r]) + (acc2[r] + acc3[r]);
        __syncthreads();
#pragma unroll
        for (int w = 0; w < NWAVES; ++w) dh += part[(w * 16 + i) * 16 + uu];
        const float tc = lc_tanh(cn);                       // explicit fma placement: see the forward step kernel
        const float do_pre = dh * tc * oa * (1.f - oa);
        const float dcn = __builtin_fmaf(do_pre, wo, __builtin_fmaf(dh * oa, __builtin_fmaf(-tc, tc, 1.f), dc));
        const float di_pre = dcn * ja * ia * (1.f - ia);
        const float dj_pre = dcn * ia * __builtin_fmaf(-ja, ja, 1.f);
        const float df_pre = dcn * cp * fa * (1.f - fa);
        const bool act = t < len;
        const float odi = act ? di_pre : 0.f, odj = act ? dj_pre : 0.f, odf = act ? df_pre : 0.f, odo = act ? do_pre : 0.f;
        dc = act ? __builtin_fmaf(df_pre, wf, __builtin_fmaf(di_pre, wi, dcn * fa)) : dc;
        ug[0] = __builtin_fmaf(odi, cp, ug[0]); ug[1] = __builtin_fmaf(odf, cp, ug[1]); ug[2] = __builtin_fmaf(odo, cn, ug[2]);
        ug[3] += odi; ug[4] += odj; ug[5] += odf; ug[6] += odo;
        // what the other workgroups wait for goes out first (one 16-byte store), the saved dz after the arrival
        if (valid)
            *reinterpret_cast<f32x4 *>(dzTg + (size_t)(step & 1) * G * 16 + pubidx) =
                p_with_lsb_tag(odi, odj, odf, odo, p_gen_bit((unsigned)step + 1u));
        LC_PSTAMP(3);
        if (LC_P_F32_AHEAD == 2) request_operands(min(step + 1, T - 1));
        if (valid) { grow[0] = odi; grow[8] = odj; grow[16] = odf; grow[24] = odo; }
        __syncthreads();                       // `part` is rewritten by the next step
        LC_PSTAMP(4);
    }
    if (valid && p.upg[dirx]) {
        float *o = p.upg[dirx] + (size_t)b * 7 * N + n;
#pragma unroll
        for (int k = 0; k < 7; ++k) o[(size_t)k * N] = ug[k];
    }
    if (failed && lane == 0) p_report_failure(p.ctl);
}

// ------------------------------------------------------------------------------ persistent recurrence, bf16 operands
// Config c5 (N = 1024, bf16 step-GEMM operands): R in bf16 is 8 MB per direction - half of ONE XCD's register file
// (32 CUs x 4 SIMDs x 512 VGPRs x 64 lanes x 4 bytes = 16 MB).  So the same schedule as above carries over with one XCD
// per (direction, row group of 16 rows): every wave keeps its K slice of the workgroup's 32 units (128 gate columns
// forward, 32 columns of R^T backward) as bf16 MFMA fragments in 256 VGPRs for all T steps, and the step GEMM runs on
// v_mfma_f32_16x16x32_bf16.  Exchange (round 4): in both passes the PRODUCER does the nearest-even rounding the consumers
// of the per-step kernels do on load and publishes MFMA-ready bf16 pieces - forward 8-byte granules of a quad of units, two
// of which are a consumer lane's operand of a K32-block, backward 16-byte pieces of a unit pair's four gate derivatives each -
// and freshness is a sentinel over four buffers (comments in front of p_fetch_hq / p_fetch_pc).  The launch train it
// replaces is launch-bound at 7.6 / 9.7 us per step for 0.5 us of MFMA work.
__device__ __forceinline__ unsigned p_cvt_pk_bf16(float lo, float hi)       // bf16(lo) | bf16(hi) << 16, nearest even
{
    typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
    bf16x2v r;
    r[0] = (__bf16)lo; r[1] = (__bf16)hi;
    return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ bf16x8 p_pack_bf16(float e0, float e1, float e2, float e3, float e4, float e5, float e6, float e7)
{
    bf16x8 r;
    r[0] = (__bf16)e0; r[1] = (__bf16)e1; r[2] = (__bf16)e2; r[3] = (__bf16)e3;
    r[4] = (__bf16)e4; r[5] = (__bf16)e5; r[6] = (__bf16)e6; r[7] = (__bf16)e7;
    return r;
}
// Forward exchange: the consumers round the state to bf16 anyway, so the producer does it (same nearest-even rounding) and
// publishes the state as MFMA-ready 16-byte pieces [kb][lk][row] = the eight units 32 kb + 8 lk .. of one row (a consumer
// lane's whole A operand of a K32-block: one load per block, a wave-wide load = 8 whole cache lines, nothing to unpack).  A
// piece is two 8-byte stores - the quads of four consecutive units (adjacent lanes) gather theirs with two DPP moves -, so
// a consumer looks at the first dword of each half.  Freshness as in the BPTT below: four buffers in turn, "not arrived" is
// the sentinel ff..ff, re-armed two steps ahead by the producer.  (Round 2 / 3: 4-byte {bf16, 16-bit step} granules - twice
// the bytes through the CU's L1 / TA path, two loads and eight unpack operations per block.)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <int NBK, bool RAGGED>
__device__ __forceinline__ bool p_fetch_hq(const char *blk0, int lk, int li, int nval, int rot, int rows, unsigned limit,
                                           const PCtl *ctl, bf16x8 (&a)[NBK])
{
    unsigned n = 0;
    const char *base = blk0 + (lk * 16 + li) * 16;
    p_first_look_delay();
    for (;;) {
        u32x4 r[NBK];
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < NBK; ++j)
            r[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(base + (size_t)p_blk<RAGGED>(j, rot, nval) * 1024));
        unsigned stale = 0;
#pragma unroll
        for (int j = 0; j < NBK; ++j) {
            stale |= (!RAGGED || j < nval) ? (unsigned)(r[j].x == 0xffffffffu || r[j].z == 0xffffffffu) : 0u;
            a[j] = __builtin_bit_cast(bf16x8, r[j]);
        }
        if (__builtin_amdgcn_ballot_w64(stale != 0 && li < rows) == 0) return true;
        if (!p_keep_waiting(n, limit, ctl)) return false;
    }
}

// grid: P_GRID x 1; dynamic LDS: partial tiles [4][16][4*UP].  PERB = K32-blocks per wave (ceil(N / 128)), PPT =
// (row, unit) pairs per thread (units per workgroup / 16), NT = 4 * UP / 16 column tiles.
// SHONLY: the fp32 hs is not written - every consumer of hs in a c5 step reads its bf16 shadow (the projection product, dR,
// dproj), and a store instruction less per step is time in a kernel that is bound by its CU's vector-memory path (round 6:
// profiles/r6_c5_bptt_pmc.txt).  Instantiated for the full-width case only (N = 1024); hs16 must be given.
template <int PERB, int PPT, bool RAGGED, bool SHONLY = false>
__global__ __launch_bounds__(P_THREADS) void lstm_fwd_persist_bf16_kernel(PFwdArgs p)
{
    constexpr int NT = 4 * PPT;                  // UP = 16 * PPT units -> 64 * PPT columns
    constexpr int ncols = NT * 16, UP = 16 * PPT;
    // Two pairs per thread are ADJACENT units (2 uu, 2 uu + 1) and the saved tensors move as 8 bytes per lane; the gate
    // pre-activations of a step are requested ONE STEP AHEAD, behind the multiplies of the step before (see the BPTT kernel).
    // (one step ahead only where the weight slice lives in AGPRs - the full-width instantiation: at 832 - 992 units it shares
    // the VGPR half with everything else and the eight request registers are one too many: a spill in the time loop)
    constexpr bool ADJ = PPT == 2, AHEAD = LC_P_OPS_AHEAD && ADJ && PERB * 4 * PPT == 64;
    extern __shared__ __attribute__((aligned(16))) float p_lds[];
    __shared__ int s_slot;
    const PGeom &g = p.g;
    const int xcc = p_xcc_id();
    if (xcc >= g.ndir * g.gpd) return;
    if (threadIdx.x == 0)
        s_slot = (int)__hip_atomic_fetch_add(&p.ctl->claim[xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int slot = s_slot;
    const int dirx = xcc / g.gpd, grp = xcc % g.gpd;
    const int rows_here = min(g.rpg, g.B - grp * g.rpg);
    if (slot >= g.nwg || rows_here <= 0) return;
    const DirFwd &d = p.d[dirx];
    const int N = g.N, G = 4 * N, B = g.B, T = g.T;
    const int u0 = slot * g.upw, nu = min(g.upw, N - u0);
    float *part = p_lds;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int lir = li < rows_here ? li : 0;      // exchange loads of rows that do not exist go to row 0's piece (see lstm_fwd_persist_kernel)
    const int i = threadIdx.x >> 4, uu = threadIdx.x & 15;
    const int b = min(grp * g.rpg + i, B - 1);
    const int nkb = N / 32, per = (nkb + NWAVES - 1) / NWAVES;
    const int kb0 = min(wave * per, nkb), kb1 = min(kb0 + per, nkb);
    const int nval = kb1 - kb0, rot = nval > 0 ? slot % nval : 0;
    const size_t bufb = (size_t)N * 32;      // one exchange buffer: [N / 32 blocks][4][16 rows] pieces of 16 bytes
    char *hTg = reinterpret_cast<char *>(p.hT) + (size_t)xcc * 4 * bufb;
    bool valid[PPT];
    int nn[PPT], len[PPT];
    float wi[PPT], wf[PPT], wo[PPT], cprev[PPT];
    size_t zcol[PPT], hidx[PPT];
#pragma unroll
    for (int pp = 0; pp < PPT; ++pp) {
        const int uL = ADJ ? 2 * uu + pp : uu + 16 * pp;
        valid[pp] = i < rows_here && uL < nu;
        // (ADJ: the pair is clamped as a whole to an even in-row unit - its tensors move as 8-byte pairs at nn[0], and a
        // pair clamped per unit would start at the odd unit N - 1 and read 4 bytes past the row / the tensor)
        nn[pp] = ADJ ? min(u0 + 2 * uu, N - 2) + pp : min(u0 + uL, N - 1);
        len[pp] = valid[pp] ? p.seq_len[b] : 0;
        wi[pp] = d.w_i ? d.w_i[nn[pp]] : 0.f; wf[pp] = d.w_f ? d.w_f[nn[pp]] : 0.f; wo[pp] = d.w_o ? d.w_o[nn[pp]] : 0.f;
        cprev[pp] = 0.f;
        zcol[pp] = (size_t)(nn[pp] >> 3) * 32 + (nn[pp] & 7);
        // byte offset of the quad's half piece: piece [n / 32][lk = (n >> 3) & 3][row], half (n >> 2) & 1
        hidx[pp] = (((size_t)(nn[pp] >> 5) * 4 + ((nn[pp] >> 3) & 3)) * 16 + i) * 16 + ((nn[pp] >> 2) & 1) * 8;
    }
    // weights: slot j = block p_blk(j) of the rotated walk; lane (li = column, lk): k = 32*kb + 8*lk + e, e = 0..7
    // AREG: the full-width instantiation (N = 1024: 64 fragments = 256 registers per lane) keeps the slice in the AGPR half
    // of the register file as 128-bit TUPLES that are born there (ds_read_b128 into an "=a" operand: a fragment packed in
    // VGPRs and constrained to "a" afterwards is kept as four scattered AGPRs and gathered in front of every use) and feeds
    // them to asm MFMAs directly.  Through the intrinsic the compiler copies every fragment to VGPRs first
    // (v_accvgpr_read) and hoists the copies: 92 spilled VGPRs and ~85 scratch accesses in the time loop.
    constexpr bool AREG = PERB * NT == 64;
    bf16x8 wreg[PERB][NT];
#pragma unroll
    for (int j = 0; j < PERB; ++j) {
        const int kb = min(kb0 + p_blk<RAGGED>(j, rot, max(nval, 1)), nkb - 1);
#pragma unroll
        for (int c = 0; c < NT; ++c) {
            const int col = c * 16 + li, gate = col / UP, u2 = col - gate * UP, n2 = min(u0 + u2, N - 1);
            const bool okc = u2 < nu && j < nval;
            const float *src = d.R + (size_t)(32 * kb + 8 * lk) * G + (n2 >> 3) * 32 + gate * 8 + (n2 & 7);
            float e[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) e[q] = src[(size_t)q * G];        // address always valid (clamped): no branch per element
#pragma unroll
            for (int q = 0; q < 8; ++q) e[q] = okc ? e[q] : 0.f;
            if constexpr (AREG) {
                bf16x8 *bounce = reinterpret_cast<bf16x8 *>(part) + threadIdx.x;
                *bounce = p_pack_bf16(e[0], e[1], e[2], e[3], e[4], e[5], e[6], e[7]);
                asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)"
                             : "=a"(wreg[j][c]) : "v"((unsigned)(size_t)bounce) : "memory");
            } else {
                wreg[j][c] = p_pack_bf16(e[0], e[1], e[2], e[3], e[4], e[5], e[6], e[7]);
            }
        }
    }
    bool failed = false;
    p_global<float> *const zx_p = p_uniform(d.zx), *const cs_p = p_uniform(d.cs), *const hs_p = p_uniform(d.hs);
    p_global<unsigned short> *const hs16_p = p_uniform(d.hs16);
    const bool rev = __builtin_amdgcn_readfirstlane((int)d.reverse) != 0;
    f32x2 nz[4];                             // ADJ: the pair's four gate pre-activations as requested (AHEAD: for the step after)
    auto request_z = [&](int s) {
        const int t_ = rev ? (T - 1 - s) : s;
        typedef p_global<const f32x2> *v2p;
#pragma unroll
        for (int q = 0; q < 4; ++q) nz[q] = *(v2p)(zx_p + ((size_t)t_ * B + b) * G + zcol[0] + 8 * q);
    };
    // (one pair per thread - up to 512 units: the same, with four dword requests)
    constexpr bool AHEAD1 = LC_P_OPS_AHEAD && PPT == 1;
    float nz1[4];
    auto request_z1 = [&](int s) {
        const int t_ = rev ? (T - 1 - s) : s;
#pragma unroll
        for (int q = 0; q < 4; ++q) nz1[q] = zx_p[((size_t)t_ * B + b) * G + zcol[0] + 8 * q];
    };
    if constexpr (AHEAD) request_z(0);
    if constexpr (AHEAD1) request_z1(0);
    __syncthreads();
    for (int step = 0; step < T; ++step) {
        const int t = rev ? (T - 1 - step) : step;
        float z[PPT][4];
        if constexpr (!ADJ && !AHEAD1) {
#pragma unroll
            for (int pp = 0; pp < PPT; ++pp)
#pragma unroll
                for (int q = 0; q < 4; ++q) z[pp][q] = zx_p[((size_t)t * B + b) * G + zcol[pp] + 8 * q];
        } else if constexpr (ADJ && !AHEAD) {
            request_z(step);
        }
        auto take_z = [&]() {
            if constexpr (ADJ) {
                // (AHEAD: the copy out of the request registers is an instruction of its own - see the BPTT kernel)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x2 cv;
                    if constexpr (AHEAD) asm volatile("v_mov_b64 %0, %1" : "=v"(cv) : "v"(nz[q]));
                    else cv = nz[q];
                    z[0][q] = cv.x; z[PPT - 1][q] = cv.y;
                }
                if constexpr (AHEAD) request_z(min(step + 1, T - 1));    // (unconditional: a branch here would cost the counted waits)
            }
            if constexpr (AHEAD1) {
#pragma unroll
                for (int q = 0; q < 4; ++q) asm volatile("v_mov_b32 %0, %1" : "=v"(z[0][q]) : "v"(nz1[q]));
                request_z1(min(step + 1, T - 1));
            }
        };
        f32x4 acc[NT];
#pragma unroll
        for (int c = 0; c < NT; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (step > 0 && nval > 0) {
            const char *hp = hTg + (size_t)((step + 3) & 3) * bufb + (size_t)kb0 * 1024;      // the previous step's pieces
            bf16x8 a[PERB];
            if (!p_fetch_hq<PERB, RAGGED>(hp, lk, lir, nval, rot, rows_here, p.spin_limit, p.ctl, a)) failed = true;
            if (AHEAD && LC_P_OPS_AHEAD == 2) take_z();              // in front of the multiplies
            if constexpr (AREG) {
                if constexpr (RAGGED) {        // slots past the wave's blocks: zero weights, and a FINITE operand to go with them
#pragma unroll
                    for (int j = 0; j < PERB; ++j)
                        if (j >= nval) a[j] = p_pack_bf16(0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f);
                }
                // the operands and accumulators pass THROUGH the wait-state asm: whatever VALU wrote them is ahead of it, the
                // MFMAs behind it (the hazard recogniser does not look into inline asm)
                asm volatile("s_nop 7"
                             : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                               "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]),
                               "+v"(acc[7]));
#pragma unroll
                for (int j = 0; j < PERB; ++j)
#pragma unroll
                    for (int c = 0; c < NT; ++c)
                        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(a[j]), "a"(wreg[j][c]));
                asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 7"
                             : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]),
                               "+v"(acc[7]));
            } else {
#pragma unroll
                for (int j = 0; j < PERB; ++j) {
                    const bf16x8 aj = (!RAGGED || j < nval) ? a[j] : p_pack_bf16(0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int c = 0; c < NT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aj, wreg[j][c], acc[c], 0, 0, 0);
                }
            }
        }
        if (!(AHEAD && LC_P_OPS_AHEAD == 2 && step > 0 && nval > 0)) take_z();
#pragma unroll
        for (int c = 0; c < NT; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) part[(size_t)(wave * 16 + lk * 4 + r) * ncols + c * 16 + li] = acc[c][r];
        __syncthreads();
        float oia[PPT], oja[PPT], ofa[PPT], ooa[PPT], oh[PPT];
        unsigned oh16[PPT];                      // bf16(h) in the low half
#pragma unroll
        for (int pp = 0; pp < PPT; ++pp) {
            const int uL = min(ADJ ? 2 * uu + pp : uu + 16 * pp, UP - 1);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int w = 0; w < NWAVES; ++w) z[pp][q] += part[(size_t)(w * 16 + i) * ncols + q * UP + uL];
            const float cp = cprev[pp];
            const float ia = lc_sigmoid(__builtin_fmaf(wi[pp], cp, z[pp][0]));
            const float fa = lc_sigmoid(__builtin_fmaf(wf[pp], cp, z[pp][2] + p.forget_bias));
            const float ja = lc_tanh(z[pp][1]);
            const float cn = __builtin_fmaf(fa, cp, ia * ja);
            const float oa = lc_sigmoid(__builtin_fmaf(wo[pp], cn, z[pp][3]));
            const bool act = t < len[pp];
            oh[pp] = act ? oa * lc_tanh(cn) : 0.f;
            cprev[pp] = act ? cn : 0.f;
            oia[pp] = act ? ia : 0.f; oja[pp] = act ? ja : 0.f; ofa[pp] = act ? fa : 0.f; ooa[pp] = act ? oa : 0.f;
            // the state the other workgroups wait for goes out first: the quad's four rounded values (adjacent lanes =
            // consecutive units of one row: quad_perm [1, 0, 3, 2], then [2, 3, 2, 3]), stored by the quad's first lane,
            // which also re-arms the half piece of the buffer two steps ahead
            if constexpr (ADJ) {
                // the thread's two units are one dword of the quad's half piece, the adjacent lane's (quad_perm [1, 0, 3, 2]) the
                // other; the even lane stores and re-arms.  oh16[0] = bf16(h of 2 uu) | bf16(h of 2 uu + 1) << 16.
                if (pp == PPT - 1) {
                    oh16[0] = p_cvt_pk_bf16(oh[0], oh[pp]);
                    oh16[0] = oh16[0] == 0xffffffffu ? 0xfffeffffu : oh16[0];          // never the sentinel (see below)
                    const unsigned hi2 = (unsigned)__builtin_amdgcn_mov_dpp((int)oh16[0], 0xB1, 0xf, 0xf, true);
                    if (valid[0] && (uu & 1) == 0) {
                        *reinterpret_cast<u32x2 *>(hTg + (size_t)(step & 3) * bufb + hidx[0]) = (u32x2){oh16[0], hi2};
                        *reinterpret_cast<u32x2 *>(hTg + (size_t)((step + 2) & 3) * bufb + hidx[0]) = (u32x2){0xffffffffu, 0xffffffffu};
                    }
                }
            } else {
                const float nb = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(oh[pp]), 0xB1, 0xf, 0xf, true));
                oh16[pp] = p_cvt_pk_bf16(oh[pp], nb);
                // a dword of two all-ones NaNs would read as the "not arrived" sentinel and end in the bounded spin's time-out
                // instead of reaching "nan loss detected": any other NaN bits do
                oh16[pp] = oh16[pp] == 0xffffffffu ? 0xfffeffffu : oh16[pp];
                const unsigned hi2 = (unsigned)__builtin_amdgcn_mov_dpp((int)oh16[pp], 0xEE, 0xf, 0xf, true);
                if (valid[pp] && (uu & 3) == 0) {
                    *reinterpret_cast<u32x2 *>(hTg + (size_t)(step & 3) * bufb + hidx[pp]) = (u32x2){oh16[pp], hi2};
                    *reinterpret_cast<u32x2 *>(hTg + (size_t)((step + 2) & 3) * bufb + hidx[pp]) = (u32x2){0xffffffffu, 0xffffffffu};
                }
            }
        }
        if constexpr (ADJ) {
            if (valid[0]) {                      // (nu is a multiple of 4: a pair is valid or not as a whole)
                typedef p_global<f32x2> *v2p;
                p_global<float> *zrow = zx_p + ((size_t)t * B + b) * G + zcol[0];
                const size_t so = ((size_t)t * B + b) * N + nn[0];
                *(v2p)(zrow) = (f32x2){oia[0], oia[PPT - 1]}; *(v2p)(zrow + 8) = (f32x2){oja[0], oja[PPT - 1]};
                *(v2p)(zrow + 16) = (f32x2){ofa[0], ofa[PPT - 1]}; *(v2p)(zrow + 24) = (f32x2){ooa[0], ooa[PPT - 1]};
                *(v2p)(cs_p + so) = (f32x2){cprev[0], cprev[PPT - 1]};
                if constexpr (!SHONLY) *(v2p)(hs_p + so) = (f32x2){oh[0], oh[PPT - 1]};
                if (SHONLY || hs16_p) *(p_global<unsigned> *)(hs16_p + so) = oh16[0];  // the projection GEMM's shadow operand, no cast pass
            }
        } else {
#pragma unroll
            for (int pp = 0; pp < PPT; ++pp) {
                if (valid[pp]) {
                    p_global<float> *zrow = zx_p + ((size_t)t * B + b) * G + zcol[pp];
                    const size_t so = ((size_t)t * B + b) * N + nn[pp];
                    zrow[0] = oia[pp]; zrow[8] = oja[pp]; zrow[16] = ofa[pp]; zrow[24] = ooa[pp];
                    cs_p[so] = cprev[pp];
                    hs_p[so] = oh[pp];
                    if (hs16_p) hs16_p[so] = (unsigned short)oh16[pp];  // the projection GEMM's shadow operand, no cast pass
                }
            }
        }
        __syncthreads();                       // `part` is rewritten by the next step
    }
    if (failed && lane == 0) p_report_failure(p.ctl);
}

// Backward.  The PRODUCER rounds: the consumers multiply bf16 anyway, so a thread rounds its four gate derivatives (nearest
// even, v_cvt_pk_bf16_f32 - the instruction the consumers used) and, with the adjacent lane's (the other unit of the pair),
// publishes ONE 16-byte piece [unit 2 q: i j f o | unit 2 q + 1: i j f o] = a consumer lane's whole bf16 A operand of a
// 32-block: exchange position k = 32 kb + 8 lk + 4 s + gate belongs to unit 8 kb + 2 lk + s, pieces ordered [kb][lk][row] - a
// wave-wide load of a block covers 8 whole cache lines.  (Round 3 exchanged the fp32 (row, unit) fragments and converted on
// the consumer side: twice the bytes through every CU's L1 / TA path - 256 KB per step at N = 1024 against its 64 bytes per
// clock: the 4000 cycles the rest-of-the-chunks phase took - and a conversion per block.)  Freshness without touching a
// value: FOUR buffers in turn, a piece that has not arrived reads as the SENTINEL ff..ff (host memset; no rounding of a
// finite value gives the halfword ffff, and the producers rewrite a dword of two all-ones NaNs to fffeffff - still NaNs).  A
// producer that has seen every workgroup's step s - 1 (its own fetch of step s) re-arms its pieces of the buffer step s + 2
// will use - the one that held step s - 2, which nobody reads any more.  Between that store and the first look any
// consumer takes at that buffer (its fetch of step s + 3) lie the producer's vmcnt(0) waits of step s + 1 (gfx9 counts
// stores in vmcnt: the re-arming is acknowledged by the L2) and its publication of step s + 1, which the consumer has seen.
template <int NBK, bool RAGGED, bool PREISSUED>
__device__ __forceinline__ bool p_fetch_pc(const char *blk0, int lk, int li, int j0, int nval, int rot, int rows,
                                           unsigned limit, const PCtl *ctl, u32x4 (&raw)[NBK], unsigned &looks)
{
    unsigned n = 0;
    const char *base = blk0 + (lk * 16 + li) * 16;
    bool issue = !PREISSUED;
    if (!PREISSUED) p_first_look_delay();
    for (;;) {
        asm volatile("" ::: "memory");
        if (issue) {
#pragma unroll
            for (int j = 0; j < NBK; ++j)
                raw[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(base + (size_t)p_blk<RAGGED>(j0 + j, rot, nval) * 1024));
        }
        issue = true;
        unsigned stale = 0;
#pragma unroll
        for (int j = 0; j < NBK; ++j) stale |= (!RAGGED || j0 + j < nval) ? (unsigned)(raw[j].x == 0xffffffffu) : 0u;
        ++looks;                                 // (a register; tools/persist_probe.py prints looks per step - a pointer to it would be a stack slot)
        if (__builtin_amdgcn_ballot_w64(stale != 0 && li < rows) == 0) return true;
        if (!p_keep_waiting(n, limit, ctl)) return false;
    }
}

// NCH = ceil(K32-blocks per wave / 8) (= ceil(N / 256)); PPT = pairs per thread; NTB = PPT column tiles (units).
// The slice is walked in chunks of CS blocks (one 16-byte piece per block and lane).
// SHONLY: the fp32 dz is not written (dz16 must be given) - see lstm_fwd_persist_bf16_kernel; N = 1024 only.
#ifndef LC_P_SHADOW_QUAD
#define LC_P_SHADOW_QUAD 1                    // 0: the shadow-only BPTT stores four dwords per lane (A / B builds)
#endif
template <int NCH, int PPT, bool RAGGED, bool SHONLY = false>
__global__ __launch_bounds__(P_THREADS) void lstm_bwd_persist_bf16_kernel(PBwdArgs p)
{
    constexpr int NTB = PPT, ncols = NTB * 16, NBK = 8 * NCH;
    // Two pairs per thread are ADJACENT units (2 uu, 2 uu + 1): every saved tensor moves as 8 bytes per lane - a wave's request
    // is whole 128-byte runs, half as many instructions through the CU's address path (round 5: with the pairs 16 units apart
    // the 14 dword requests + 16 dword / short stores per thread and step cost 0.67 + 0.25 us of a 3.15 us step) - and the
    // pair's exchange piece is one thread's own.
    constexpr bool ADJ = PPT == 2;
    extern __shared__ __attribute__((aligned(16))) float p_lds[];
    __shared__ int s_slot;
    const PGeom &g = p.g;
    const int xcc = p_xcc_id();
    if (xcc >= g.ndir * g.gpd) return;
    if (threadIdx.x == 0)
        s_slot = (int)__hip_atomic_fetch_add(&p.ctl->claim[xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int slot = s_slot;
    const int dirx = xcc / g.gpd, grp = xcc % g.gpd;
    const int rows_here = min(g.rpg, g.B - grp * g.rpg);
    if (slot >= g.nwg || rows_here <= 0) return;
    const DirBwd &d = p.d[dirx];
    const int N = g.N, G = 4 * N, B = g.B, T = g.T;
    const int u0 = slot * g.upw, nu = min(g.upw, N - u0);
    float *part = p_lds;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int lir = li < rows_here ? li : 0;      // exchange loads of rows that do not exist go to row 0's piece (see lstm_fwd_persist_kernel)
    const int i = threadIdx.x >> 4, uu = threadIdx.x & 15;
    const int b = min(grp * g.rpg + i, B - 1);
    const int nkb = G / 32, per = (nkb + NWAVES - 1) / NWAVES;
    const int kb0 = min(wave * per, nkb), kb1 = min(kb0 + per, nkb);
    const int nval = kb1 - kb0, rot = nval > 0 ? (slot * 5) % nval : 0;
    const size_t bufb = (size_t)N * 128;     // one exchange buffer: [N / 8 blocks][4 unit pairs][16 rows] pieces of 16 bytes
    char *dzTg = reinterpret_cast<char *>(p.dzT) + (size_t)xcc * 4 * bufb;
    bool valid[PPT];
    int nn[PPT], len[PPT], cbase[PPT];
    float wi[PPT], wf[PPT], wo[PPT], dc[PPT];
    float ug[PPT][7];                        // bias / peephole gradient sums of this thread's (row, unit) pairs (XCD-pair BPTT)
    size_t pubidx[PPT];                      // byte offset of the pair's piece (the even unit's lane stores it)
#pragma unroll
    for (int pp = 0; pp < PPT; ++pp) {
        const int uL = ADJ ? 2 * uu + pp : uu + 16 * pp;
        valid[pp] = i < rows_here && uL < nu;
        // (ADJ: the pair is clamped as a whole to an even in-row unit - its tensors move as 8-byte pairs at nn[0], and a
        // pair clamped per unit would start at the odd unit N - 1 and read 4 bytes past the row / the tensor)
        nn[pp] = ADJ ? min(u0 + 2 * uu, N - 2) + pp : min(u0 + uL, N - 1);
        len[pp] = valid[pp] ? p.seq_len[b] : 0;
        wi[pp] = d.w_i ? d.w_i[nn[pp]] : 0.f; wf[pp] = d.w_f ? d.w_f[nn[pp]] : 0.f; wo[pp] = d.w_o ? d.w_o[nn[pp]] : 0.f;
        dc[pp] = 0.f;
#pragma unroll
        for (int k = 0; k < 7; ++k) ug[pp][k] = 0.f;
        cbase[pp] = (nn[pp] >> 3) * 32 + (nn[pp] & 7);
        pubidx[pp] = (((size_t)(nn[pp] >> 3) * 4 + ((nn[pp] >> 1) & 3)) * 16 + i) * 16;
    }
    // weights: slot j = block p_blk(j); lane (li = column = unit u0 + c*16 + li, lk): k = 32*kb + 8*lk + e,
    // e = 4*s + gate -> unit 8*kb + 2*lk + s -> row (n/8)*32 + gate*8 + n%8 of R^T
    constexpr bool AREG = NBK * NTB == 64;
    bf16x8 wreg[NBK][NTB];
#pragma unroll
    for (int j = 0; j < NBK; ++j) {
        const int kb = min(kb0 + p_blk<RAGGED>(j, rot, max(nval, 1)), nkb - 1);
#pragma unroll
        for (int c = 0; c < NTB; ++c) {
            const bool okc = c * 16 + li < nu && j < nval;
            const int colg = min(u0 + c * 16 + li, N - 1);
            float e[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int kn = 8 * kb + 2 * lk + (q >> 2), kg = q & 3;
                e[q] = d.RT[(size_t)((kn >> 3) * 32 + kg * 8 + (kn & 7)) * N + colg];
                e[q] = okc ? e[q] : 0.f;
            }
            if constexpr (AREG) {        // as in the forward kernel: the fragment is born as an AGPR tuple
                bf16x8 *bounce = reinterpret_cast<bf16x8 *>(part) + threadIdx.x;
                *bounce = p_pack_bf16(e[0], e[1], e[2], e[3], e[4], e[5], e[6], e[7]);
                asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)"
                             : "=a"(wreg[j][c]) : "v"((unsigned)(size_t)bounce) : "memory");
            } else {
                wreg[j][c] = p_pack_bf16(e[0], e[1], e[2], e[3], e[4], e[5], e[6], e[7]);
            }
        }
    }
    bool failed = false;
    // The saved tensors a step's gate math reads (gates, dh, c, c of the step before) come from HBM, and gfx9 returns loads in
    // order: requested at the top of their step (rounds 1 - 4) they stood in front of the exchange pieces in the queue, and the
    // look at chunk 0 - always fresh at the first attempt (tools/persist_probe.py counts looks) - waited their round trip first:
    // without them a look takes 2200 cycles, with them 3300 - 3700 (N = 1024; timing builds, profiles/r5_bf16_bptt_operands.txt).
    // With two or more chunks per step they are requested ONE STEP AHEAD, behind the multiplies of the step before (the round
    // trip passes under reduce, gate math and publication); with one chunk (N <= 512) that place delays the publication by more
    // than it saves, and they stay at the top.  Requested behind the look instead (a hand-counted wait that leaves them in
    // flight): no gain - tools/dropped/dropped_r5_ops_behind_look.diff.
    p_global<float> *const gates_p = p_uniform(d.gates);
    p_global<const float> *const dh_p = p_uniform(d.dh), *const cs_p = p_uniform(d.cs);
    p_global<unsigned short> *const dz16_p = p_uniform(d.dz16);
    const bool rev = __builtin_amdgcn_readfirstlane((int)d.reverse) != 0;
    constexpr bool AHEAD = LC_P_OPS_AHEAD && NBK > 16;
    static_assert(!AHEAD || ADJ, "more than one chunk means more than 512 units: two adjacent pairs per thread");
    float ia[PPT], ja[PPT], fa[PPT], oa[PPT], dh[PPT], cn[PPT], cp[PPT];       // the step in hand
    f32x2 nv[7];                             // ADJ: the pair's i, j, f, o, dh, c, c_prev as requested (AHEAD: for the step after)
    auto request_operands = [&](int s) {     // 7 (adjacent pairs) or 7 PPT loads, all unconditional, nothing computed from them here:
        const int t_ = rev ? s : (T - 1 - s);                                       // any arithmetic would wait for the request at once
        const int tp_ = rev ? min(t_ + 1, T - 1) : max(t_ - 1, 0);      // (no frame before the first: any valid address, zeroed at the use)
        if constexpr (ADJ) {                     // both units of the pair at once (cbase, nn even: 8-byte aligned)
            typedef p_global<const f32x2> *v2p;
            p_global<const float> *grow = gates_p + ((size_t)t_ * B + b) * G + cbase[0];
            const size_t so = ((size_t)t_ * B + b) * N + nn[0];
            nv[0] = *(v2p)(grow); nv[1] = *(v2p)(grow + 8); nv[2] = *(v2p)(grow + 16); nv[3] = *(v2p)(grow + 24);
            nv[4] = *(v2p)(dh_p + so); nv[5] = *(v2p)(cs_p + so);
            nv[6] = *(v2p)(cs_p + ((size_t)tp_ * B + b) * N + nn[0]);
            if (LC_P_DEV_SKIP_OPS) { nv[0] = nv[1] = nv[2] = nv[3] = (f32x2){0.5f, 0.5f}; nv[4] = (f32x2){0.01f, 0.01f}; nv[5] = nv[6] = (f32x2){0.1f, 0.1f}; }
        } else {
#pragma unroll
            for (int pp = 0; pp < PPT; ++pp) {
                p_global<const float> *grow = gates_p + ((size_t)t_ * B + b) * G + cbase[pp];
                const size_t so = ((size_t)t_ * B + b) * N + nn[pp];
                ia[pp] = grow[0]; ja[pp] = grow[8]; fa[pp] = grow[16]; oa[pp] = grow[24];
                dh[pp] = dh_p[so];
                cn[pp] = cs_p[so];
                cp[pp] = cs_p[((size_t)tp_ * B + b) * N + nn[pp]];
            }
        }
    };
    if constexpr (AHEAD) request_operands(0);
    __syncthreads();
    for (int step = 0; step < T; ++step) {
        const int t = rev ? step : (T - 1 - step);
        const bool has_prev = rev ? (t + 1 < T) : (t > 0);
        LC_PSTAMP(0);
        if constexpr (!AHEAD) request_operands(step);
        auto take_operands = [&]() {
            if constexpr (ADJ) {
                // AHEAD: the copy out of the request registers is an instruction of its own - left to itself the compiler keeps the
                // step's values where they arrived, requests into fresh registers and closes the loop with wait + copy straight
                // behind the requests: an exposed HBM round trip per step
                f32x2 cv[7];
#pragma unroll
                for (int k = 0; k < 7; ++k) {
                    if constexpr (AHEAD) asm volatile("v_mov_b64 %0, %1" : "=v"(cv[k]) : "v"(nv[k]));
                    else cv[k] = nv[k];
                }
                ia[0] = cv[0].x; ia[PPT - 1] = cv[0].y; ja[0] = cv[1].x; ja[PPT - 1] = cv[1].y; fa[0] = cv[2].x; fa[PPT - 1] = cv[2].y;
                oa[0] = cv[3].x; oa[PPT - 1] = cv[3].y; dh[0] = cv[4].x; dh[PPT - 1] = cv[4].y; cn[0] = cv[5].x; cn[PPT - 1] = cv[5].y;
                cp[0] = cv[6].x; cp[PPT - 1] = cv[6].y;
            }
#pragma unroll
            for (int pp = 0; pp < PPT; ++pp) cp[pp] = has_prev ? cp[pp] : 0.f;
            if constexpr (AHEAD) request_operands(min(step + 1, T - 1));     // (unconditional: a branch here would cost the counted waits)
        };
        f32x4 acc[NTB][2];
#pragma unroll
        for (int c = 0; c < NTB; ++c) { acc[c][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[c][1] = acc[c][0]; }
        if (step > 0 && nval > 0) {
            const char *ap = dzTg + (size_t)((step + 3) & 3) * bufb + (size_t)kb0 * 1024;      // the previous step's pieces
            const char *base = ap + (lk * 16 + lir) * 16;
#ifndef LC_BF16_BWD_CS
#define LC_BF16_BWD_CS 8                       // round 5: with the operands out of the queue the smaller first chunk wins (2.78 -> 2.57 us; 16 before)
#endif
            constexpr int CS = AREG ? LC_BF16_BWD_CS : 8, NCHK = NBK / CS;
            u32x4 raw[CS];
            // chunk 0 is polled; every later chunk is requested once its predecessor has been copied out, and flies under
            // the predecessor's multiplies.  (Measured alternatives of round 2, with fp32 fragments: rings of request buffers,
            // loop-free chunks with a redo, polled passes, one-dword probes - all slower: with all 32 workgroups of the XCD
            // pulling their slices at once the walk runs at the L2's / the CU's L1 delivery rate, not at a latency.)
            unsigned nlooks = 0, more = 0;
            if (!p_fetch_pc<CS, RAGGED, false>(ap, lk, lir, 0, nval, rot, rows_here, p.spin_limit, p.ctl, raw, nlooks)) failed = true;
            LC_PSTAMP(1);
            if (p.dbg && xcc == 0 && slot == 0 && threadIdx.x == 0) p.dbg[step * 8 + 5] = nlooks;
#pragma unroll
            for (int ch = 0; ch < NCHK; ++ch) {
                bf16x8 a[CS];
#pragma unroll
                for (int j = 0; j < CS; ++j) {
                    a[j] = __builtin_bit_cast(bf16x8, raw[j]);
                    if (RAGGED && ch * CS + j >= nval) a[j] = p_pack_bf16(0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f);
                }
                if (ch + 1 < NCHK) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < CS; ++j)
                        raw[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(
                            base + (size_t)p_blk<RAGGED>((ch + 1) * CS + j, rot, nval) * 1024));
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (AHEAD && LC_P_OPS_AHEAD == 2 && ch + 1 == NCHK) take_operands();     // in front of the last multiplies
                if constexpr (AREG) {
                    static_assert(!AREG || (CS >= 8 && NTB == 2), "operand lists below");
                    asm volatile("s_nop 7" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]),     // VALU-packed operands -> asm MFMA
                                             "+v"(a[CS - 4]), "+v"(a[CS - 3]), "+v"(a[CS - 2]), "+v"(a[CS - 1]),
                                             "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[NTB - 1][0]), "+v"(acc[NTB - 1][1]));
#pragma unroll
                    for (int j = 0; j < CS; ++j)
#pragma unroll
                        for (int c = 0; c < NTB; ++c)
                            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0"
                                         : "+v"(acc[c][j & 1]) : "v"(a[j]), "a"(wreg[ch * CS + j][c]));
                } else {
#pragma unroll
                    for (int j = 0; j < CS; ++j)
#pragma unroll
                        for (int c = 0; c < NTB; ++c)
                            acc[c][j & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j], wreg[ch * CS + j][c], acc[c][j & 1], 0, 0, 0);
                }
                if (ch + 1 < NCHK)
                    if (!p_fetch_pc<CS, RAGGED, true>(ap, lk, lir, (ch + 1) * CS, nval, rot, rows_here, p.spin_limit, p.ctl, raw, more)) failed = true;
            }
            if constexpr (AREG)                  // MFMA results -> VALU / LDS reads (no hazard recogniser for asm)
                asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 7"
                             : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[NTB - 1][0]), "+v"(acc[NTB - 1][1]));
        }
        LC_PSTAMP(2);
        if (!(AHEAD && LC_P_OPS_AHEAD == 2 && step > 0 && nval > 0)) take_operands();
#pragma unroll
        for (int c = 0; c < NTB; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) part[(wave * 16 + lk * 4 + r) * ncols + c * 16 + li] = acc[c][0][r] + acc[c][1][r];
        __syncthreads();
        float odi[PPT], odj[PPT], odf[PPT], odo[PPT];
        unsigned rij[PPT], rfo[PPT];             // the rounded derivatives: bf16(i) | bf16(j) << 16, bf16(f) | bf16(o) << 16
#pragma unroll
        for (int pp = 0; pp < PPT; ++pp) {
            const int uL = min(ADJ ? 2 * uu + pp : uu + 16 * pp, ncols - 1);
            float dhh = dh[pp];
#pragma unroll
            for (int w = 0; w < NWAVES; ++w) dhh += part[(w * 16 + i) * ncols + uL];
            const float tc = lc_tanh(cn[pp]);
            const float do_pre = dhh * tc * oa[pp] * (1.f - oa[pp]);
            const float dcn = __builtin_fmaf(do_pre, wo[pp], __builtin_fmaf(dhh * oa[pp], __builtin_fmaf(-tc, tc, 1.f), dc[pp]));
            const float di_pre = dcn * ja[pp] * ia[pp] * (1.f - ia[pp]);
            const float dj_pre = dcn * ia[pp] * __builtin_fmaf(-ja[pp], ja[pp], 1.f);
            const float df_pre = dcn * cp[pp] * fa[pp] * (1.f - fa[pp]);
            const bool act = t < len[pp];
            odi[pp] = act ? di_pre : 0.f; odj[pp] = act ? dj_pre : 0.f; odf[pp] = act ? df_pre : 0.f; odo[pp] = act ? do_pre : 0.f;
            dc[pp] = act ? __builtin_fmaf(df_pre, wf[pp], __builtin_fmaf(di_pre, wi[pp], dcn * fa[pp])) : dc[pp];
            // what the other workgroups wait for goes out first: the pair's piece (adjacent lanes = the two units of a pair:
            // quad_perm [1, 0, 3, 2]), stored by the even unit's lane; its piece of the buffer two steps ahead is re-armed
            rij[pp] = p_cvt_pk_bf16(odi[pp], odj[pp]);
            rij[pp] = rij[pp] == 0xffffffffu ? 0xfffeffffu : rij[pp];      // never the sentinel: two all-ones NaNs stay NaNs
            rfo[pp] = p_cvt_pk_bf16(odf[pp], odo[pp]);
            if constexpr (ADJ) {                 // the pair's piece is this thread's own: it leaves once its second unit is done
                if (pp == PPT - 1 && valid[0]) {
                    *reinterpret_cast<u32x4 *>(dzTg + (size_t)(step & 3) * bufb + pubidx[0]) = (u32x4){rij[0], rfo[0], rij[pp], rfo[pp]};
                    *reinterpret_cast<u32x4 *>(dzTg + (size_t)((step + 2) & 3) * bufb + pubidx[0]) =
                        (u32x4){0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
                }
            } else {
                const u32x4 piece = {rij[pp], rfo[pp], (unsigned)__builtin_amdgcn_mov_dpp((int)rij[pp], 0xB1, 0xf, 0xf, true),
                                     (unsigned)__builtin_amdgcn_mov_dpp((int)rfo[pp], 0xB1, 0xf, 0xf, true)};
                if (valid[pp] && !(uu & 1)) {
                    *reinterpret_cast<u32x4 *>(dzTg + (size_t)(step & 3) * bufb + pubidx[pp]) = piece;
                    *reinterpret_cast<u32x4 *>(dzTg + (size_t)((step + 2) & 3) * bufb + pubidx[pp]) =
                        (u32x4){0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
                }
            }
            ug[pp][0] = __builtin_fmaf(odi[pp], cp[pp], ug[pp][0]); ug[pp][1] = __builtin_fmaf(odf[pp], cp[pp], ug[pp][1]);
            ug[pp][2] = __builtin_fmaf(odo[pp], cn[pp], ug[pp][2]);
            ug[pp][3] += odi[pp]; ug[pp][4] += odj[pp]; ug[pp][5] += odf[pp]; ug[pp][6] += odo[pp];
        }
        if constexpr (ADJ && SHONLY && LC_P_SHADOW_QUAD) {
            // Shadow only, as WHOLE 64-byte lines.  A lane holds the four gate words of its unit pair, and the four lanes of a
            // quad (uu & 3) hold the four pairs of one 8-unit block, whose shadow is one 64-byte run [i x 8 | j x 8 | f x 8 | o x 8]:
            // a 4 x 4 transpose across the quad (two DPP butterfly stages) gives lane q the 16 bytes of gate q, and ONE store
            // instruction writes 16 full lines per wave - 16 L2 write requests instead of 4 instructions x 16 partial ones
            // (round 6, profiles/r6_c5_bptt_pmc.txt: the requests are what the CU's vector-memory path pays for).
            const unsigned r0 = rij[0], r1 = rij[PPT - 1], q0 = rfo[0], q1 = rfo[PPT - 1];
            unsigned w[4] = {(r0 & 0xffffu) | (r1 << 16), (r0 >> 16) | (r1 & 0xffff0000u),
                             (q0 & 0xffffu) | (q1 << 16), (q0 >> 16) | (q1 & 0xffff0000u)};
            const bool odd = uu & 1, hi = uu & 2;
#pragma unroll
            for (int k = 0; k < 4; k += 2) {         // lane ^ 1, register ^ 1
                const unsigned send = odd ? w[k] : w[k + 1];
                const unsigned recv = (unsigned)__builtin_amdgcn_mov_dpp((int)send, 0xB1, 0xf, 0xf, true);     // quad_perm [1,0,3,2]
                w[k] = odd ? recv : w[k];
                w[k + 1] = odd ? w[k + 1] : recv;
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {            // lane ^ 2, register ^ 2
                const unsigned send = hi ? w[k] : w[k + 2];
                const unsigned recv = (unsigned)__builtin_amdgcn_mov_dpp((int)send, 0x4E, 0xf, 0xf, true);     // quad_perm [2,3,0,1]
                w[k] = hi ? recv : w[k];
                w[k + 2] = hi ? w[k + 2] : recv;
            }
            if (valid[0] && !LC_P_DEV_SKIP_SAVED) {      // (N = 1024: every workgroup has whole blocks of 8 units)
                p_global<unsigned short> *g16 = dz16_p + ((size_t)t * B + b) * G + (size_t)(nn[0] >> 3) * 32 + (uu & 3) * 8;
                *(p_global<u32x4> *)(g16) = (u32x4){w[0], w[1], w[2], w[3]};
            }
        } else if constexpr (ADJ) {
            if (valid[0] && !LC_P_DEV_SKIP_SAVED) {      // (nu is a multiple of 4: a pair is valid or not as a whole)
                typedef p_global<f32x2> *v2p;
                p_global<float> *grow = gates_p + ((size_t)t * B + b) * G + cbase[0];
                if constexpr (!SHONLY) {
                    *(v2p)(grow) = (f32x2){odi[0], odi[PPT - 1]}; *(v2p)(grow + 8) = (f32x2){odj[0], odj[PPT - 1]};
                    *(v2p)(grow + 16) = (f32x2){odf[0], odf[PPT - 1]}; *(v2p)(grow + 24) = (f32x2){odo[0], odo[PPT - 1]};
                }
                if (SHONLY || dz16_p) {                                   // dX = dz . Kx^T reads this shadow: no cast pass
                    typedef p_global<unsigned> *u1p;
                    p_global<unsigned short> *g16 = dz16_p + ((size_t)t * B + b) * G + cbase[0];
                    const unsigned r0 = rij[0], r1 = rij[PPT - 1], q0 = rfo[0], q1 = rfo[PPT - 1];
                    *(u1p)(g16) = (r0 & 0xffffu) | (r1 << 16); *(u1p)(g16 + 8) = (r0 >> 16) | (r1 & 0xffff0000u);
                    *(u1p)(g16 + 16) = (q0 & 0xffffu) | (q1 << 16); *(u1p)(g16 + 24) = (q0 >> 16) | (q1 & 0xffff0000u);
                }
            }
        } else {
#pragma unroll
            for (int pp = 0; pp < PPT; ++pp) {
                if (valid[pp] && !LC_P_DEV_SKIP_SAVED) {
                    p_global<float> *grow = gates_p + ((size_t)t * B + b) * G + cbase[pp];
                    grow[0] = odi[pp]; grow[8] = odj[pp]; grow[16] = odf[pp]; grow[24] = odo[pp];
                    if (dz16_p) {                                   // dX = dz . Kx^T reads this shadow: no cast pass
                        p_global<unsigned short> *g16 = dz16_p + ((size_t)t * B + b) * G + cbase[pp];
                        g16[0] = (unsigned short)rij[pp]; g16[8] = (unsigned short)(rij[pp] >> 16);
                        g16[16] = (unsigned short)rfo[pp]; g16[24] = (unsigned short)(rfo[pp] >> 16);
                    }
                }
            }
        }
        LC_PSTAMP(3);
        __syncthreads();                       // `part` is rewritten by the next step
        LC_PSTAMP(4);
    }
    if (p.upg[dirx]) {
#pragma unroll
        for (int pp = 0; pp < PPT; ++pp)
            if (valid[pp]) {
                float *o = p.upg[dirx] + (size_t)b * 7 * N + nn[pp];
#pragma unroll
                for (int k = 0; k < 7; ++k) o[(size_t)k * N] = ug[pp][k];
            }
    }
    if (failed && lane == 0) p_report_failure(p.ctl);
}

// ------------------------------------------------------------------------------ persistent recurrence over XCD pairs
// Config c4 (fp32, N = 1024, both directions, <= 64 rows): R is 16 MB per direction - ONE XCD's whole register file - so
// the single-XCD schedule above cannot hold it, and the launch train re-streams it from the Infinity Cache every step
// (48 MB read per forward launch, 62 MB per backward launch: 11.5 / 15.4 us per step against a 6.8 us MFMA floor).
// Here each (direction, half of the batch rows) gets TWO XCDs and the step GEMM is split along K between them:
//
//   XCD x:  direction x >> 2, rows [32 r, 32 r + 32) with r = (x >> 1) & 1, unit half u = x & 1, partner x ^ 1.
//   The XCD owns the recurrent state of ITS 512 units and keeps the 512 ROWS of R that multiply them - all 4096 gate
//   columns - resident in registers (8 MB: 256 VGPRs per lane of every wave, for all T steps).  Workgroup `slot` of
//   32 owns gate columns of 16 units of its own half and of the 16 units of the partner's half with the same number:
//   z[rows, those 128 columns] = h_own_half . R[own rows, columns] is a PARTIAL sum; the half that belongs to the
//   partner's units travels to the partner XCD's workgroup with the same slot (a 1:1 hand-off of 4 KB, system-scope
//   16-byte stores / loads, generation bit in every dword), the other half meets the partner's contribution here.
//   What every workgroup of an XCD needs each step is therefore only the state of the XCD's OWN 512 units, exchanged
//   through the XCD's own L2 exactly as in the single-XCD schedule; nothing is broadcast across XCDs.
//
// Latency hiding: the 32 rows of an XCD are two independent groups of 16 (one MFMA row tile each).  Per step a workgroup
// runs  MFMA(A) - send(A) - MFMA(B) - send(B) - receive(A) - gates(A) - publish(A) - receive(B) - gates(B) - publish(B):
// group A's partial sums cross the fabric while group B multiplies, and B's state of the previous step has been in L2
// for a whole group time before it is needed.  Per workgroup and step: 2 x 256 v_mfma_f32_16x16x4_f32 per wave
// (2 x 3.4 us), exact fp32 products.  The partial sums travel as exact 8-byte {value, step} granules.  The state copies the
// XCD's workgroups exchange carry their freshness as a generation bit in the lowest mantissa bit, like the dz fragments of
// the single-XCD BPTT (<= 1 ulp on an operand of the recurrent product, step-dependent; everything SAVED is exact): with
// 8-byte state granules the 16 instead of 8 requests per lane and half step cost 3500 of 22900 cycles per step (measured).
// Consequence: bit-exact mirror symmetry between the two directions holds for an utterance whose start is a multiple of 4
// steps into the reverse walk, 1e-6-level agreement otherwise (test_full_size_c4_properties checks both).
constexpr int X_LDP = 132;                   // row pitch (floats) of a wave's [16 x 128] partial tile in LDS
constexpr int X_SYS = 17;                    // aux bits sc0 | sc1: system scope (coherent across the XCDs' L2s)
struct XFwdArgs {
    DirFwd d[2];                             // hT unused; the two "direction slots" (4 XCDs each) of the launch
    const int *seq_len;
    int row_base[2];                         // first batch row of each slot's 64-row block (see pair_blocks)
    int T, B;                                // B: rows of the WHOLE batch = the frame stride of every tensor
    float forget_bias;
    unsigned spin_limit;
    PCtl *ctl;
    float *hx;                               // [8 XCDs][2 groups][2 buffers][512 units x 16 rows] granules, [unit / 4][row][4]
    float *px;                               // [8 XCDs (receiver)][2 groups][2 buffers][32 slots][16 rows][16 units][4 gates] granules
    unsigned long long *dbg;                 // optional s_memtime stamps [T][16] of one workgroup (tools/pair_probe.py)
};
#define LC_XSTAMP(k)                                                                           \
    do {                                                                                       \
        if (p.dbg && xcc == 0 && slot == 0 && threadIdx.x == 0) p.dbg[step * 16 + (k)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
constexpr size_t X_HX_FLOATS = (size_t)8 * 2 * 2 * 512 * 16;
constexpr size_t X_PX_FLOATS = (size_t)8 * 2 * 2 * 32 * 16 * 16 * 4 * 2;
constexpr int X_HXBUF = 512 * 16;            // floats per (group, buffer) of hx
constexpr int X_PXBLK = 32 * 16 * 16 * 32;   // bytes per (receiver, group, buffer) of px

// grid: P_GRID x 1 (64 candidates per XCD, the first 32 claim a slot); dynamic LDS: two partial-tile buffers.
//
// Instruction stream of a wave.  A "half step" k multiplies group X = k & 1 (step s = k >> 1): 8 chunks of 32 MFMAs (one
// 16-unit block of K each).  An MFMA occupies the matrix pipe for 32 cycles but the wave only ~4 cycles to issue it, so
// everything else of the schedule - the whole post-processing of the OTHER group's previous product (LDS reduce of the four
// waves' partial tiles, hand-off of the partner's share, receipt of the partner's contribution, gate math, publishing the
// new state, storing the saved activations) - is placed BETWEEN the chunks of this group's MFMAs and rides in their shadow,
// loads one chunk ahead of their use.  (First version, post-processing after the MFMAs: 31100 cycles per step of which
// 20700 MFMA - s_memtime stamps, tools/pair_probe.py.)
struct XGroup {                              // per row group, in registers
    float z[4];                              // x_t . Kx + b of this thread's (row, unit): requested a half step ahead
    float zloc[4];                           // this XCD's partial sums
    float cprev;
};
// NKB: 16-unit K blocks per wave = N / 128 (N = 640 .. 1024 in steps of 128): an XCD owns HALF = 64 NKB units, 4 NKB
// workgroups hold a slice (16 units of either half each), a wave's K slice is 16 NKB units = 32 NKB weight registers.  The
// exchange buffers keep their N = 1024 strides for every width.
template <int NKB>
__global__ __launch_bounds__(P_THREADS) void lstm_fwd_pair_kernel(XFwdArgs p)
{
    extern __shared__ __attribute__((aligned(16))) float p_lds[];
    __shared__ int s_slot, s_fail;
    const int xcc = p_xcc_id();
    if (xcc >= 8) return;
    if (threadIdx.x == 0) {
        s_fail = 0;
        s_slot = (int)__hip_atomic_fetch_add(&p.ctl->claim[xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const int slot = s_slot;
    constexpr int HALF = 64 * NKB, NSLOT = 4 * NKB;
    if (slot >= NSLOT) return;
    const int dirx = xcc >> 2, rh = (xcc >> 1) & 1, uh = xcc & 1;
    const DirFwd &d = p.d[dirx];
    constexpr int N = 128 * NKB, G = 4 * N;
    const int B = p.B, T = p.T;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int i = threadIdx.x >> 4, ul = threadIdx.x & 15;        // this thread's (row of a group, unit) pair
    const int n = uh * HALF + slot * 16 + ul;                      // its unit (own half)
    const float wi = d.w_i ? d.w_i[n] : 0.f, wf = d.w_f ? d.w_f[n] : 0.f, wo = d.w_o ? d.w_o[n] : 0.f;
    const size_t zcol = (size_t)(n >> 3) * 32 + (n & 7);
    int brow[2], len[2];
    bool valid[2];
#pragma unroll
    for (int sg = 0; sg < 2; ++sg) {
        const int b = p.row_base[dirx] + rh * 32 + sg * 16 + i;
        valid[sg] = b < B;
        brow[sg] = min(b, B - 1);
        len[sg] = valid[sg] ? p.seq_len[brow[sg]] : 0;
    }
    // R rows of this wave's K slice (own-half units 128 w .. 128 w + 127), the workgroup's 128 columns, as MFMA B fragments:
    // block kb (16 units), quad q, column tile c: lane (li = column, lk) holds R[unit 16 kb + 4 lk + q][column], i.e. the
    // exchange buffer's granule (unit / 4, row) = 4 consecutive units is one A lane's float4.  Column c * 16 + li of the
    // workgroup: gate c >> 1 of unit li, own half for even c, partner's half for odd c.
    struct XW { float x, y, z, w; };
    XW wreg[NKB][8];                         // [kb][c], members = quads: 32 NKB registers (a0-a255 after allocation at NKB = 8)
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        const int k0 = uh * HALF + wave * (16 * NKB) + kb * 16 + 4 * lk;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int unit = ((c & 1) ? (1 - uh) : uh) * HALF + slot * 16 + li;
            const float *src = d.R + (size_t)k0 * G + (unit >> 3) * 32 + (c >> 1) * 8 + (unit & 7);
            wreg[kb][c].x = src[0]; wreg[kb][c].y = src[(size_t)G];
            wreg[kb][c].z = src[(size_t)2 * G]; wreg[kb][c].w = src[(size_t)3 * G];
        }
    }
    float *hxme = p.hx + (size_t)xcc * 2 * 2 * X_HXBUF;
    const x_i32x4 px_rs = x_rsrc(p.px, (unsigned)(X_PX_FLOATS * sizeof(float)));
    const int pxcell = ((slot * 16 + i) * 16 + ul) * 32;           // this thread's 32-byte cell (4 granules) in a [slot][row][unit] block
    // The state exchange of this XCD through one buffer descriptor.  A granule [unit / 4][row] (4 units = 16 bytes = one
    // consumer lane's MFMA fragment) is published by ONE lane in ONE 16-byte store: the four producer threads of a granule
    // are the four lanes of a quad, three DPP moves gather their values, the quad's first lane stores (the others' store
    // offset lies outside the descriptor: dropped, no branch).  A consumer therefore checks ONE generation bit per fragment
    // instead of four (56 -> 16 VALU instructions per half step, each at its full issue cost next to f32 MFMAs).
    const x_i32x4 hx_rs = x_rsrc(hxme, (unsigned)(2 * 2 * X_HXBUF * sizeof(float)));
    const int hx_vo = (lk * 16 + li) * 16;                                          // bytes
    const int hx_so = __builtin_amdgcn_readfirstlane(wave * NKB * 1024);           // + ((group * 2 + buffer) * X_HXBUF + kb * 256) * 4
    const int hx_pub = (ul & 3) == 0 ? ((slot * 4 + (ul >> 2)) * 16 + i) * 16 : 0x7ffffff0;
    constexpr int X_NT = 2;                                                         // aux: nt
    auto publish_state = [&](int sg, int s, float h) {
        const int v = (int)((__float_as_uint(h) & ~1u) | p_gen_bit((unsigned)s + 1u));
        const f32x4 g = {__int_as_float(__builtin_amdgcn_mov_dpp(v, 0x00, 0xf, 0xf, true)),
                         __int_as_float(__builtin_amdgcn_mov_dpp(v, 0x55, 0xf, 0xf, true)),
                         __int_as_float(__builtin_amdgcn_mov_dpp(v, 0xaa, 0xf, 0xf, true)),
                         __int_as_float(__builtin_amdgcn_mov_dpp(v, 0xff, 0xf, 0xf, true))};
        x_buffer_store_b128(g, hx_rs, hx_pub, (sg * 2 + (s & 1)) * (X_HXBUF * 4), 0);
    };
    XGroup grp[2];
    grp[0].cprev = grp[1].cprev = 0.f;
    f32x4 a[NKB];                            // the multiplying group's previous state (MFMA A fragments)
    bool failed = false;
    int step = 0;                            // (for the stamp macro)

    // zx / cs / hs of a (row, unit) through buffer descriptors: a per-lane byte offset per group + a scalar frame offset
    // (pair_geom keeps T * B * 4N * 4 below 2^32; offsets are unsigned 32-bit quantities): no 64-bit address arithmetic in the time loop - every instruction in
    // it costs its full issue time (f32 MFMAs and VALU do not overlap on a SIMD: tools/ubench/mfma_agpr_rate.hip)
    const x_i32x4 zx_rs = x_rsrc(d.zx, (unsigned)((size_t)T * B * G * sizeof(float)));
    const x_i32x4 cs_rs = x_rsrc(d.cs, (unsigned)((size_t)T * B * N * sizeof(float)));
    const x_i32x4 hs_rs = x_rsrc(d.hs, (unsigned)((size_t)T * B * N * sizeof(float)));
    int zvo[2], svo[2];
#pragma unroll
    for (int sg = 0; sg < 2; ++sg) { zvo[sg] = (brow[sg] * G + (int)zcol) * 4; svo[sg] = (brow[sg] * N + n) * 4; }
    auto load_zx = [&](int sg, int s) {      // requested early: does not depend on the recurrence
        const int zo = (int)((unsigned)(d.reverse ? (T - 1 - s) : s) * (unsigned)(B * G * 4));   // < 2^32: pair_geom
#pragma unroll
        for (int g = 0; g < 4; ++g) grp[sg].z[g] = x_buffer_load_b32(zx_rs, zvo[sg] + 32 * g, zo, 0);
    };
    // gate math of group sg at step s from z + zsum, publish the new state, store the saved activations
    auto gates = [&](int sg, int s, const float (&zsum)[4]) {
        const int t = d.reverse ? (T - 1 - s) : s;
        const float cp = grp[sg].cprev;
        const float ia = lc_sigmoid(__builtin_fmaf(wi, cp, grp[sg].z[0] + zsum[0]));
        const float fa = lc_sigmoid(__builtin_fmaf(wf, cp, (grp[sg].z[2] + zsum[2]) + p.forget_bias));
        const float ja = lc_tanh(grp[sg].z[1] + zsum[1]);
        const float cn = __builtin_fmaf(fa, cp, ia * ja);
        const float oa = lc_sigmoid(__builtin_fmaf(wo, cn, grp[sg].z[3] + zsum[3]));
        const bool act = t < len[sg];
        const float h = act ? oa * lc_tanh(cn) : 0.f;
        grp[sg].cprev = act ? cn : 0.f;
        // what the XCD's workgroups wait for goes out first: own-half unit (16 slot + ul) -> granule [unit / 4][row][unit % 4],
        // tag s + 1 in buffer s & 1
        publish_state(sg, s, h);
        if (valid[sg]) {
            float *zrow = d.zx + ((size_t)t * B + brow[sg]) * G + zcol;
            const size_t so = ((size_t)t * B + brow[sg]) * N + n;
            zrow[0] = act ? ia : 0.f; zrow[8] = act ? ja : 0.f; zrow[16] = act ? fa : 0.f; zrow[24] = act ? oa : 0.f;
            d.cs[so] = grp[sg].cprev;
            d.hs[so] = h;
        }
    };
    // previous state of group sg (published in step s - 1, tag s), this wave's K slice: 8 fragments per lane
    auto request_state = [&](int sg, int s) {
        const int so = hx_so + (sg * 2 + ((s + 1) & 1)) * (X_HXBUF * 4);
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) a[kb] = x_buffer_load_b128(hx_rs, hx_vo, so + kb * 1024, X_NT);
    };
    auto state_stale = [&](int s, const f32x4 (&av)[NKB]) {
        unsigned stale = 0;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) stale |= __float_as_uint(av[kb].x) ^ p_gen_bit((unsigned)s);   // one store per fragment
        return __builtin_amdgcn_ballot_w64((stale & 1u) != 0) != 0;
    };

    // ---- step 0: no recurrent product
    load_zx(0, 0);
    load_zx(1, 0);
    {
        const float zero[4] = {0.f, 0.f, 0.f, 0.f};
        gates(0, 0, zero);
        gates(1, 0, zero);
    }
    if (T > 1) { load_zx(0, 1); load_zx(1, 1); }
    __syncthreads();
    // ---- half steps k = 2 .. 2T-1: MFMA of group X = k & 1 at step s = k >> 1; in its shadow the post-processing of group
    //      Y = X ^ 1 at step sy = (k - 1) >> 1 (from k = 3 on; Y's step 0 needed none)
    // (the group index is a compile-time constant of each half step: per-group state stays in registers)
    // POST: the other group has a previous product to post-process (every half step but the very first) - a compile-time
    // property, so that the micro-steps below are straight-line code between the MFMAs.
    auto half = [&](auto XC, auto PC, int s) -> bool {
        constexpr int X = decltype(XC)::value, Y = X ^ 1;
        constexpr bool POST = decltype(PC)::value;
        const int k = 2 * s + X, sy = (k - 1) >> 1;
        step = s;
        float *partX = p_lds + (size_t)X * (NWAVES * 16 * X_LDP), *partY = p_lds + (size_t)Y * (NWAVES * 16 * X_LDP);
        if (X == 0) LC_XSTAMP(0); else LC_XSTAMP(8);
        {   // this group's previous state: requested at the end of the last half step where possible; poll until fresh
            if (k == 2) request_state(X, s);
            if (state_stale(s, a)) {             // first look outside the loop (counted waits: see the BPTT kernel)
                unsigned nspin = 0;
                for (;;) {
                    if (!p_keep_waiting(nspin, p.spin_limit, p.ctl)) { failed = true; break; }
                    asm volatile("" ::: "memory");
                    request_state(X, s);
                    if (!state_stale(s, a)) break;
                }
            }
        }
        if (X == 0) LC_XSTAMP(1); else LC_XSTAMP(9);
        f32x4 acc[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // VALU-written accumulators -> asm MFMA: no hazard recogniser for asm, so the wait states are written out - and the
        // accumulators pass THROUGH the statement, or the zeroing (trivially rematerialisable) is re-emitted right in front
        // of the MFMA that reads it (it was, once the micro-steps raised the register pressure: garbage results)
        asm volatile("s_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]),
                                 "+v"(acc[6]), "+v"(acc[7]) : : "memory");
        // The 256 weight registers live in the ACCUMULATION half of the register file (a0-a255 are the only place left for
        // them beside ~120 working VGPRs) and feed the MFMA's B operand from there DIRECTLY: through the intrinsic the
        // compiler copies each one to a VGPR first (v_accvgpr_read + a dependent MFMA: 40 instead of 32 cycles per MFMA
        // and no issue slot left for anything else - measured), hence the asm with "a" operands.
        //
        // Y's post-processing as MICRO-STEPS, one behind each MFMA: an MFMA holds the matrix pipe for 32 cycles and the wave
        // for ~4, and the wave cannot issue the next MFMA before the pipe is free - so ~28 cycles (five simple VALU
        // instructions, or one transcendental and one more) ride free behind EVERY MFMA, but no more: a 500-cycle piece
        // behind a chunk of 32 MFMAs (the first version of this schedule) leaves the pipe idle for 470 of them.
        float pl[4][NWAVES], pr[4][NWAVES], zrem[4], zsum[4] = {0.f, 0.f, 0.f, 0.f};
        f32x4 rv0 = {0.f, 0.f, 0.f, 0.f}, rv1 = rv0;
        float gx = 0.f, ge = 0.f, ia = 0.f, fa = 0.f, ja = 0.f, cn = 0.f, oa = 0.f, th = 0.f, hh = 0.f;
        bool act = false;
        const float tag_y = __uint_as_float((unsigned)sy);               // tag of the partial sums of Y's step sy (>= 1)
        const int pxblk = (sy + 1) & 1;
        const int pxsend = (((xcc ^ 1) * 2 + Y) * 2 + pxblk) * X_PXBLK + pxcell, pxrecv = ((xcc * 2 + Y) * 2 + pxblk) * X_PXBLK + pxcell;
        const int ty = d.reverse ? (T - 1 - sy) : sy;
        const unsigned tyu = (unsigned)sy;
        auto late = [&]() {
            return __builtin_amdgcn_ballot_w64(((__float_as_uint(rv0.y) ^ tyu) | (__float_as_uint(rv0.w) ^ tyu) |
                                                (__float_as_uint(rv1.y) ^ tyu) | (__float_as_uint(rv1.w) ^ tyu)) != 0) != 0;
        };
        // chunks of the receive / sum / gate-math pieces: 4, 5, 6 of 8 at N = 1024 (the partner's share, sent in chunk 1, has
        // three chunk times to arrive); with fewer chunks they move up and a late share is polled for
        constexpr int C_RECV = NKB - 4 > 2 ? NKB - 4 : 2, C_SUM = C_RECV + 1, C_GATE = C_RECV + 2;
        static_assert(C_GATE < NKB, "five chunks at least");
        auto mic = [&](auto KC, int m) {          // m = 0..31: the MFMA of chunk KC it follows (a constant once unrolled)
            constexpr int KB = decltype(KC)::value;
            XGroup &q = grp[Y];
            if constexpr (KB == 0 && POST) {      // Y's partial tiles (written before the last barrier): one LDS read each
                const int g = m >> 3, w = (m >> 1) & 3;
                if (m & 1) pr[g][w] = partY[(size_t)(w * 16 + i) * X_LDP + g * 32 + 16 + ul];
                else pl[g][w] = partY[(size_t)(w * 16 + i) * X_LDP + g * 32 + ul];
            } else if constexpr (KB == 1 && POST) {   // reduce; the partner's share goes out (system scope, tag per value)
                if (m < 8) {
                    const int g = m >> 1;
                    if (m & 1) zrem[g] = (pr[g][0] + pr[g][1]) + (pr[g][2] + pr[g][3]);
                    else q.zloc[g] = (pl[g][0] + pl[g][1]) + (pl[g][2] + pl[g][3]);
                } else if (m == 8) {
                    x_buffer_store_b128((f32x4){zrem[0], tag_y, zrem[1], tag_y}, px_rs, pxsend, 0, X_SYS);
                } else if (m == 9) {
                    x_buffer_store_b128((f32x4){zrem[2], tag_y, zrem[3], tag_y}, px_rs, pxsend + 16, 0, X_SYS);
                }
            } else if constexpr (KB == C_RECV && POST) {
                if (m == 0) rv0 = x_buffer_load_b128(px_rs, pxrecv, 0, X_SYS);
                else if (m == 1) rv1 = x_buffer_load_b128(px_rs, pxrecv + 16, 0, X_SYS);
            } else if constexpr (KB == C_SUM && POST) {
                if (m == 0) {                     // the partner's contribution: normally there by now
                    if (late()) {
                        unsigned nspin = 0;
                        for (;;) {
                            if (!p_keep_waiting(nspin, p.spin_limit, p.ctl)) { failed = true; break; }
                            rv0 = x_buffer_load_b128(px_rs, pxrecv, 0, X_SYS);
                            rv1 = x_buffer_load_b128(px_rs, pxrecv + 16, 0, X_SYS);
                            if (!late()) break;
                        }
                    }
                } else if (m == 1) {
                    zsum[0] = q.zloc[0] + rv0.x; zsum[1] = q.zloc[1] + rv0.z;
                    zsum[2] = q.zloc[2] + rv1.x; zsum[3] = q.zloc[3] + rv1.z;
                }
            } else if constexpr (KB == C_GATE && POST) {   // the gate math of `gates`, operation for operation, one piece per MFMA
                const float cp = q.cprev;
                if (m == 0) gx = __builtin_fmaf(wi, cp, q.z[0] + zsum[0]);
                else if (m == 1) ge = __builtin_amdgcn_exp2f(-1.44269504088896341f * gx);
                else if (m == 2) ia = __builtin_amdgcn_rcpf(1.0f + ge);
                else if (m == 3) gx = __builtin_fmaf(wf, cp, (q.z[2] + zsum[2]) + p.forget_bias);
                else if (m == 4) ge = __builtin_amdgcn_exp2f(-1.44269504088896341f * gx);
                else if (m == 5) fa = __builtin_amdgcn_rcpf(1.0f + ge);
                else if (m == 6) { gx = q.z[1] + zsum[1]; ge = __builtin_amdgcn_exp2f(-2.88539008177792681f * fabsf(gx)); }
                else if (m == 7) ja = copysignf((1.0f - ge) * __builtin_amdgcn_rcpf(1.0f + ge), gx);
                else if (m == 8) { cn = __builtin_fmaf(fa, cp, ia * ja); gx = __builtin_fmaf(wo, cn, q.z[3] + zsum[3]); }
                else if (m == 9) ge = __builtin_amdgcn_exp2f(-1.44269504088896341f * gx);
                else if (m == 10) oa = __builtin_amdgcn_rcpf(1.0f + ge);
                else if (m == 11) ge = __builtin_amdgcn_exp2f(-2.88539008177792681f * fabsf(cn));
                else if (m == 12) th = copysignf((1.0f - ge) * __builtin_amdgcn_rcpf(1.0f + ge), cn);
                else if (m == 13) { act = ty < len[Y]; hh = act ? oa * th : 0.f; q.cprev = act ? cn : 0.f; }
                else if (m == 14) {               // what the XCD's workgroups wait for goes out first
                    publish_state(Y, sy, hh);
                } else if (m == 16 || m == 18 || m == 20) {
                    if (valid[Y]) {
                        const int zo = (int)((unsigned)ty * (unsigned)(B * G * 4)), so = ty * B * N * 4;
                        if (m == 16) {
                            x_buffer_store_b32(act ? ia : 0.f, zx_rs, zvo[Y], zo, 0);
                            x_buffer_store_b32(act ? ja : 0.f, zx_rs, zvo[Y] + 32, zo, 0);
                        } else if (m == 18) {
                            x_buffer_store_b32(act ? fa : 0.f, zx_rs, zvo[Y] + 64, zo, 0);
                            x_buffer_store_b32(act ? oa : 0.f, zx_rs, zvo[Y] + 96, zo, 0);
                        } else {
                            x_buffer_store_b32(q.cprev, cs_rs, svo[Y], so, 0);
                            x_buffer_store_b32(hh, hs_rs, svo[Y], so, 0);
                        }
                    }
                } else if (m == 22) {
                    if (sy + 1 < T) load_zx(Y, sy + 1);            // next step's pre-activations of Y
                }
            } else if constexpr (KB == C_GATE && !POST) {
                if (m == 0 && sy + 1 < T) load_zx(Y, sy + 1);
            }
        };
#define LC_XMFMA(ACC, A, W) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "a"(W))
#define LC_XCHUNK(KB)                                                                                              \
    _Pragma("unroll") for (int c = 0; c < 8; ++c) {                                                                \
        LC_XMFMA(acc[c], a[KB].x, wreg[KB][c].x); mic(std::integral_constant<int, KB>(), c);                       \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
    }                                                                                                              \
    _Pragma("unroll") for (int c = 0; c < 8; ++c) {                                                                \
        LC_XMFMA(acc[c], a[KB].y, wreg[KB][c].y); mic(std::integral_constant<int, KB>(), 8 + c);                   \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
    }                                                                                                              \
    _Pragma("unroll") for (int c = 0; c < 8; ++c) {                                                                \
        LC_XMFMA(acc[c], a[KB].z, wreg[KB][c].z); mic(std::integral_constant<int, KB>(), 16 + c);                  \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
    }                                                                                                              \
    _Pragma("unroll") for (int c = 0; c < 8; ++c) {                                                                \
        LC_XMFMA(acc[c], a[KB].w, wreg[KB][c].w); mic(std::integral_constant<int, KB>(), 24 + c);                  \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
    }
        LC_XCHUNK(0) LC_XCHUNK(1) LC_XCHUNK(2) LC_XCHUNK(3) LC_XCHUNK(4)
        if constexpr (NKB > 5) { LC_XCHUNK(5) }
        if constexpr (NKB > 6) { LC_XCHUNK(6) }
        if constexpr (NKB > 7) { LC_XCHUNK(7) }
#undef LC_XCHUNK
#undef LC_XMFMA
        // the asm MFMAs are invisible to the compiler's hazard recogniser: cover the last result's latency by hand
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]),
                                                         "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]) : : "memory");
        if (X == 0) LC_XSTAMP(2); else LC_XSTAMP(10);
        // 16x16 C layout: col = lane & 15, row = (lane >> 4) * 4 + r
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) partX[(size_t)(wave * 16 + lk * 4 + r) * X_LDP + c * 16 + li] = acc[c][r];
        // Y's next product starts the next half step and needs the state published just now by every workgroup of the XCD
        if (k + 1 < 2 * T) request_state(Y, (k + 1) >> 1);
        if (failed) s_fail = 1;
        __syncthreads();
        if (X == 0) LC_XSTAMP(3); else LC_XSTAMP(11);
        return s_fail == 0;
    };
    {
        const std::integral_constant<int, 0> g0;
        const std::integral_constant<int, 1> g1;
        const std::true_type post;
        bool ok = T > 1 && half(g0, std::false_type(), 1) && half(g1, post, 1);
        for (int s = 2; ok && s < T; ++s) ok = half(g0, post, s) && half(g1, post, s);
    }
    // ---- the last product (group 1, step T - 1) has nobody's MFMAs to hide behind
    if (T > 1 && !s_fail) {
        const int Y = 1, sy = T - 1;
        const float *partY = p_lds + (size_t)Y * (NWAVES * 16 * X_LDP);
        const float tag_y = __uint_as_float((unsigned)sy);
        const int pxblk = (sy + 1) & 1;
        const int pxsend = (((xcc ^ 1) * 2 + Y) * 2 + pxblk) * X_PXBLK + pxcell, pxrecv = ((xcc * 2 + Y) * 2 + pxblk) * X_PXBLK + pxcell;
        float zrem[4], zsum[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {          // the same association as inside the loop: every step rounds alike
            const float *q = partY + (size_t)i * X_LDP + g * 32 + ul;
            grp[Y].zloc[g] = (q[0] + q[(size_t)16 * X_LDP]) + (q[(size_t)32 * X_LDP] + q[(size_t)48 * X_LDP]);
            zrem[g] = (q[16] + q[(size_t)16 * X_LDP + 16]) + (q[(size_t)32 * X_LDP + 16] + q[(size_t)48 * X_LDP + 16]);
        }
        x_buffer_store_b128((f32x4){zrem[0], tag_y, zrem[1], tag_y}, px_rs, pxsend, 0, X_SYS);
        x_buffer_store_b128((f32x4){zrem[2], tag_y, zrem[3], tag_y}, px_rs, pxsend + 16, 0, X_SYS);
        unsigned nspin = 0;
        const unsigned ty = (unsigned)sy;
        f32x4 rv0, rv1;
        for (;;) {
            rv0 = x_buffer_load_b128(px_rs, pxrecv, 0, X_SYS);
            rv1 = x_buffer_load_b128(px_rs, pxrecv + 16, 0, X_SYS);
            if (__builtin_amdgcn_ballot_w64(((__float_as_uint(rv0.y) ^ ty) | (__float_as_uint(rv0.w) ^ ty) |
                                             (__float_as_uint(rv1.y) ^ ty) | (__float_as_uint(rv1.w) ^ ty)) != 0) == 0) break;
            if (!p_keep_waiting(nspin, p.spin_limit, p.ctl)) { failed = true; break; }
        }
        zsum[0] = grp[Y].zloc[0] + rv0.x; zsum[1] = grp[Y].zloc[1] + rv0.z;
        zsum[2] = grp[Y].zloc[2] + rv1.x; zsum[3] = grp[Y].zloc[3] + rv1.z;
        gates(Y, sy, zsum);
        if (__syncthreads_or(failed ? 1 : 0)) s_fail = 1;
    }
    if (s_fail && threadIdx.x == 0) p_report_failure(p.ctl);     // persist_verify_kernel turns the outputs into NaN
}

// ---- BPTT on XCD pairs: dm'_rec[rows, N] = dz_{t'}[rows, 4N] . R^T, again split along K: an XCD keeps the 2048 ROWS of
// R^T that belong to the gate columns of ITS 512 units (all 1024 output units: 8 MB in registers) and multiplies the dz it
// produced itself (exchanged inside the XCD as 16-byte (row, unit) fragments with generation bits, as in the single-XCD
// BPTT); workgroup `slot` owns the output units 16 slot .. 16 slot + 15 of both halves, keeps the partial sums of its
// own half and sends those of the partner's half to the partner's workgroup `slot` (8-byte {value, step} granules).
// Same half-step pipeline as the forward kernel: MFMA of one row group, the other group's post-processing in its shadow.
constexpr int XB_LDP = 36;                   // row pitch (floats) of a wave's [16 x 32] partial tile in LDS
typedef float x_f32x2 __attribute__((ext_vector_type(2)));
struct XBwdArgs {
    DirBwd d[2];                             // dc, dzT unused
    const int *seq_len;
    int row_base[2];                         // as in XFwdArgs
    int T, B;
    unsigned spin_limit;
    PCtl *ctl;
    float *dzx;                              // [8 XCDs][2 groups][2 buffers][512 units][16 rows][4 gates]
    float *px;                               // [8 XCDs (receiver)][2 groups][2 buffers][32 slots][16 rows][16 units] granules
    float *upg[2];                           // per direction: [64 batch rows][7][N] partial bias / peephole gradients, or NULL
    unsigned long long *dbg;
};
constexpr int XB_DZBUF = 512 * 16 * 4;       // floats per (group, buffer) of dzx
constexpr size_t XB_DZX_FLOATS = (size_t)8 * 2 * 2 * XB_DZBUF;
constexpr int XB_PXBLK = 32 * 16 * 16 * 8;   // bytes per (receiver, group, buffer) of px
constexpr size_t XB_PX_FLOATS = (size_t)8 * 2 * 2 * XB_PXBLK / 4;

struct XBGroup {                             // per row group, in registers: the operands of the gate derivatives, requested
    float ia, ja, fa, oa, dh, cn, cp;        // a half step ahead, and the carried cell gradient
    float dc;
    float dloc;                              // this XCD's partial sum for this thread's (row, unit)
};
// NKB as in the forward kernel (N = 128 NKB); a wave's K slice is NB = 4 NKB blocks of (4 units x 4 gates).
template <int NKB>
__global__ __launch_bounds__(P_THREADS) void lstm_bwd_pair_kernel(XBwdArgs p)
{
    extern __shared__ __attribute__((aligned(16))) float p_lds[];
    __shared__ int s_slot, s_fail;
    const int xcc = p_xcc_id();
    if (xcc >= 8) return;
    if (threadIdx.x == 0) {
        s_fail = 0;
        s_slot = (int)__hip_atomic_fetch_add(&p.ctl->claim[xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const int slot = s_slot;
    constexpr int HALF = 64 * NKB, NSLOT = 4 * NKB, NB = 4 * NKB;
    if (slot >= NSLOT) return;
    const int dirx = xcc >> 2, rh = (xcc >> 1) & 1, uh = xcc & 1;
    const DirBwd &d = p.d[dirx];
    constexpr int N = 128 * NKB, G = 4 * N;
    const int B = p.B, T = p.T;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int i = threadIdx.x >> 4, ul = threadIdx.x & 15;        // this thread's (row of a group, unit) pair
    const int n = uh * HALF + slot * 16 + ul;                      // its unit (own half)
    const float wi = d.w_i ? d.w_i[n] : 0.f, wf = d.w_f ? d.w_f[n] : 0.f, wo = d.w_o ? d.w_o[n] : 0.f;
    const int cbase = (n >> 3) * 32 + (n & 7);
    int brow[2], len[2];
    bool valid[2];
#pragma unroll
    for (int sg = 0; sg < 2; ++sg) {
        const int b = p.row_base[dirx] + rh * 32 + sg * 16 + i;
        valid[sg] = b < B;
        brow[sg] = min(b, B - 1);
        len[sg] = valid[sg] ? p.seq_len[brow[sg]] : 0;
    }
    // R^T rows of this wave's K slice: own-half units 128 w .. 128 w + 127, four gates each; block kb = 4 units x 4 gates:
    // lane (li = output column, lk) holds for quad q R^T[(unit 4 kb + lk, gate q)][output unit]; output column tile 0 =
    // own-half units 16 slot + li, tile 1 = the partner's.  (The exchange fragment of (row, unit) is its four gate
    // derivatives = one A lane's float4.)
    struct XW { float x, y, z, w; };
    XW wreg[NB][2];                          // 32 NKB registers (a0-a255 at NKB = 8)
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
        const int ku = uh * HALF + wave * (16 * NKB) + kb * 4 + lk;
        const float *src0 = d.RT + (size_t)((ku >> 3) * 32 + (ku & 7)) * N;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int ou = (c ? (1 - uh) : uh) * HALF + slot * 16 + li;
            wreg[kb][c].x = src0[ou]; wreg[kb][c].y = src0[(size_t)8 * N + ou];
            wreg[kb][c].z = src0[(size_t)16 * N + ou]; wreg[kb][c].w = src0[(size_t)24 * N + ou];
        }
    }
    float *dzme = p.dzx + (size_t)xcc * 2 * 2 * XB_DZBUF;
    const x_i32x4 px_rs = x_rsrc(p.px, (unsigned)(XB_PX_FLOATS * sizeof(float)));
    const int pxcell = ((slot * 16 + i) * 16 + ul) * 8;            // this thread's granule in a [slot][row][unit] block
    // The exchange buffers of this XCD through ONE buffer descriptor: a fragment request is a per-lane byte offset (one
    // VGPR for all 48 requests of a half step) + a wave-uniform scalar offset + an immediate.  (With flat pointers the
    // compiler kept a 64-bit address pair per request live across the loop - 58 VGPRs - and spilled `len[]`, whose
    // reload in the middle of the MFMA stream cost a vmcnt(0) behind the operand refills.)
    const x_i32x4 dz_rs = x_rsrc(dzme, (unsigned)(2 * 2 * XB_DZBUF * sizeof(float)));
    const int dz_vo = (lk * 16 + li) * 16;                                         // bytes
    const int dz_so = __builtin_amdgcn_readfirstlane(wave * NKB * 4096);           // + ((group * 2 + buffer) * XB_DZBUF + kb * 256) * 4
    const int dz_pub = ((slot * 16 + ul) * 16 + i) * 16;                           // this thread's (row, unit) fragment, bytes
    constexpr int X_NT = 2;                                                        // aux: nt
    XBGroup grp[2];
    grp[0].dc = grp[1].dc = 0.f;
    // The multiplying group's previous dz (MFMA A fragments): 16 of the wave's 32 blocks at a time - a block's register is
    // re-requested with block + 16 as soon as its MFMAs have read it (all 32 held through the MFMA phase is 128 VGPRs:
    // with them the kernel spilled to scratch, 53 instead of 10 us per step).  The freshness poll at the start of a half
    // step covers all 32 (t[] holds blocks 16.. only for their tags), so the re-requests need no check of their own: a
    // check in the MFMA stream means control flow, and behind control flow the compiler waits with vmcnt(0) - for the
    // youngest request, i.e. a full round trip per block (measured: 50 instead of 39 cycles per MFMA).
    f32x4 a[2][16];                          // operand rings of the two row groups (compile-time index everywhere)
    // Bias / peephole gradients of this thread's two (row, unit) pairs, summed over the steps as they are produced: 7 adds per
    // half step instead of a pass over dz and cs afterwards (2.4 ms per c4 step: 1.5 GB read per layer and direction).  Written
    // out per batch row at the end; unit_param_fold_kernel adds the 64 rows in index order (deterministic).
    float ug[2][7] = {{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}};
    auto uacc = [&](int sg, float gi, float gj, float gf, float go, float cp, float cn) {
        ug[sg][0] = __builtin_fmaf(gi, cp, ug[sg][0]); ug[sg][1] = __builtin_fmaf(gf, cp, ug[sg][1]);
        ug[sg][2] = __builtin_fmaf(go, cn, ug[sg][2]);
        ug[sg][3] += gi; ug[sg][4] += gj; ug[sg][5] += gf; ug[sg][6] += go;
    };
    bool failed = false;
    int step = 0;

    auto tof = [&](int s) { return d.reverse ? s : (T - 1 - s); };           // BPTT visits the frames in the opposite order
    // saved activations / incoming gradient of a (row, unit): one descriptor per tensor, a per-lane byte offset per group and
    // a scalar frame offset (pair_geom keeps T * B * 4N * 4 below 2^32) - no 64-bit address arithmetic in the time loop
    const x_i32x4 g_rs = x_rsrc(d.gates, (unsigned)((size_t)T * B * G * sizeof(float)));
    const x_i32x4 dh_rs = x_rsrc(d.dh, (unsigned)((size_t)T * B * N * sizeof(float)));
    const x_i32x4 cs_rs = x_rsrc(d.cs, (unsigned)((size_t)T * B * N * sizeof(float)));
    int gvo[2], svo[2];
#pragma unroll
    for (int sg = 0; sg < 2; ++sg) { gvo[sg] = (brow[sg] * G + cbase) * 4; svo[sg] = (brow[sg] * N + n) * 4; }
    auto load_gates = [&](int sg, int s) {
        const int go = (int)((unsigned)tof(s) * (unsigned)(B * G * 4));          // < 2^32: pair_geom
        grp[sg].ia = x_buffer_load_b32(g_rs, gvo[sg], go, 0); grp[sg].ja = x_buffer_load_b32(g_rs, gvo[sg] + 32, go, 0);
        grp[sg].fa = x_buffer_load_b32(g_rs, gvo[sg] + 64, go, 0); grp[sg].oa = x_buffer_load_b32(g_rs, gvo[sg] + 96, go, 0);
    };
    auto load_rest = [&](int sg, int s) {
        const int t = tof(s);
        const int tprev = d.reverse ? t + 1 : t - 1;
        const bool has_prev = d.reverse ? (t + 1 < T) : (t > 0);
        grp[sg].dh = x_buffer_load_b32(dh_rs, svo[sg], t * B * N * 4, 0);
        grp[sg].cn = x_buffer_load_b32(cs_rs, svo[sg], t * B * N * 4, 0);
        const float cpv = x_buffer_load_b32(cs_rs, svo[sg], (has_prev ? tprev : t) * B * N * 4, 0);
        grp[sg].cp = has_prev ? cpv : 0.f;
    };
    auto load_operands = [&](int sg, int s) { load_gates(sg, s); load_rest(sg, s); };
    // gate derivatives of group sg at step s; drec = recurrent part of dm'
    auto derivs = [&](int sg, int s, float drec) {
        const int t = tof(s);
        const XBGroup &q = grp[sg];
        const float dh = q.dh + drec;
        const float tc = lc_tanh(q.cn);                      // explicit fma placement: see the forward step kernel
        const float do_pre = dh * tc * q.oa * (1.f - q.oa);
        const float dcn = __builtin_fmaf(do_pre, wo, __builtin_fmaf(dh * q.oa, __builtin_fmaf(-tc, tc, 1.f), q.dc));
        const float di_pre = dcn * q.ja * q.ia * (1.f - q.ia);
        const float dj_pre = dcn * q.ia * __builtin_fmaf(-q.ja, q.ja, 1.f);
        const float df_pre = dcn * q.cp * q.fa * (1.f - q.fa);
        const bool act = t < len[sg];
        const float odi = act ? di_pre : 0.f, odj = act ? dj_pre : 0.f, odf = act ? df_pre : 0.f, odo = act ? do_pre : 0.f;
        uacc(sg, odi, odj, odf, odo, q.cp, q.cn);
        grp[sg].dc = act ? __builtin_fmaf(df_pre, wf, __builtin_fmaf(di_pre, wi, dcn * q.fa)) : q.dc;
        // what the XCD's workgroups wait for goes out first: fragment [unit][row][4 gates], tag s + 1 in buffer s & 1
        x_buffer_store_b128(p_with_lsb_tag(odi, odj, odf, odo, p_gen_bit((unsigned)s + 1u)), dz_rs, dz_pub,
                            (sg * 2 + (s & 1)) * (XB_DZBUF * 4), 0);
        if (valid[sg]) {
            const int go = (int)((unsigned)t * (unsigned)(B * G * 4));
            x_buffer_store_b32(odi, g_rs, gvo[sg], go, 0); x_buffer_store_b32(odj, g_rs, gvo[sg] + 32, go, 0);
            x_buffer_store_b32(odf, g_rs, gvo[sg] + 64, go, 0); x_buffer_store_b32(odo, g_rs, gvo[sg] + 96, go, 0);
        }
    };
    // blocks 0..15 of group SG's operand (dz of step s - 1, generation s): normally prefetched behind the MFMAs of the
    // previous half step (LC_XB below); this full request serves the very first half step and the retry path
    auto request_dz = [&](auto SG, int s) {
        constexpr int sg = decltype(SG)::value;
        const int so = dz_so + (sg * 2 + ((s + 1) & 1)) * (XB_DZBUF * 4);
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) a[sg][kb] = x_buffer_load_b128(dz_rs, dz_vo, so + kb * 1024, X_NT);
    };
    auto dz_stale = [&](auto SG, int s) {
        constexpr int sg = decltype(SG)::value;
        unsigned stale = 0;
        const unsigned gen = p_gen_bit((unsigned)s);
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) stale |= __float_as_uint(a[sg][kb].x) ^ gen;     // one store per fragment: see `mic`
        return __builtin_amdgcn_ballot_w64((stale & 1u) != 0) != 0;
    };

    // ---- step 0: no recurrent term
    load_operands(0, 0);
    load_operands(1, 0);
    derivs(0, 0, 0.f);
    derivs(1, 0, 0.f);
    __syncthreads();
    auto half = [&](auto XC, auto PC, int s) -> bool {
        constexpr int X = decltype(XC)::value, Y = X ^ 1;
        constexpr bool POST = decltype(PC)::value;       // the other group has a product to post-process (all but the first)
        const int k = 2 * s + X, sy = (k - 1) >> 1;
        step = s;
        float *partX = p_lds + (size_t)X * (NWAVES * 16 * XB_LDP), *partY = p_lds + (size_t)Y * (NWAVES * 16 * XB_LDP);
        if (X == 0) LC_XSTAMP(0); else LC_XSTAMP(8);
        {   // blocks 0..15 were requested behind the last blocks of the previous half step: normally all there and fresh
            if (k == 2) request_dz(XC, s);
            if (dz_stale(XC, s)) {
                unsigned nspin = 0;
                for (;;) {
                    if (!p_keep_waiting(nspin, p.spin_limit, p.ctl)) { failed = true; break; }
                    asm volatile("" ::: "memory");
                    request_dz(XC, s);
                    if (!dz_stale(XC, s)) break;
                }
            }
        }
        if (X == 0) LC_XSTAMP(1); else LC_XSTAMP(9);
        f32x4 acc[2][2];                       // [tile][even / odd quad]: two chains per tile
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[c][0] = acc[c][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // VALU-written accumulators -> asm MFMA (no hazard recogniser for asm); they pass through the statement so that the
        // zeroing cannot be rematerialised behind it (see the forward kernel)
        asm volatile("s_nop 7" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]) : : "memory");
        const int aso = dz_so + (X * 2 + ((s + 1) & 1)) * (XB_DZBUF * 4);
        const unsigned agen = p_gen_bit((unsigned)s);
        // the OTHER group's operand for the next half step (its dz of step sn - 1 is published at block 16 of this one)
        const int sn = (k + 1) >> 1, nso = dz_so + (Y * 2 + ((sn + 1) & 1)) * (XB_DZBUF * 4);
        // ---- everything but the MFMAs, as MICRO-STEPS one behind each MFMA (~28 free cycles each: forward kernel):
        //      * this group's operand ring: a refilled block (16..31) is checked just before its turn - a late one (the
        //        first 16 were fresh thousands of cycles ago: it does not happen) is re-requested until it is fresh -,
        //        block b + 16 is requested into the register of block b, and behind blocks 24..31 go the requests for
        //        blocks 0..15 of the other group's next operand;
        //      * the other group's post-processing: LDS reduce, hand-off of the partner's share, receipt of the partner's
        //        contribution, gate derivatives operation by operation, publish, stores.
        float pl[NWAVES], pr[NWAVES];
        x_f32x2 rv = {0.f, 0.f};
        float dh = 0.f, tc = 0.f, ge = 0.f, psum = 0.f, do_pre = 0.f, dcn = 0.f, di_pre = 0.f, dj_pre = 0.f, df_pre = 0.f;
        float odi = 0.f, odj = 0.f, odf = 0.f, odo = 0.f;
        const float tag_y = __uint_as_float((unsigned)sy);
        const int pxblk = (sy + 1) & 1;
        const int pxsend = (((xcc ^ 1) * 2 + Y) * 2 + pxblk) * XB_PXBLK + pxcell, pxrecv = ((xcc * 2 + Y) * 2 + pxblk) * XB_PXBLK + pxcell;
        const int ty = tof(sy);
        const bool acty = ty < len[Y];
        // where the pieces sit among the NB blocks (N = 1024: receive 10, check 14, derivatives 15, publish 16, stores 17,
        // the other group's next operand behind blocks 24..31): the publish stays at the middle, the requests for the other
        // group's operand - which every workgroup of the XCD publishes around ITS middle - at the last eight blocks
        constexpr int B_PUB = NB / 2, B_DER = B_PUB - 1, B_CHK = B_PUB - 2, B_RCV = B_PUB - 6 > 5 ? B_PUB - 6 : 5,
                      B_STO = B_PUB + 1, B_NXT = NB - 8;
        static_assert(B_RCV < B_CHK && B_STO < B_NXT + 8 && B_NXT >= B_PUB, "pair BPTT: block schedule");
        auto mic = [&](auto BC, int m) {          // m = 0..7: the MFMA of block BC it follows (a constant once unrolled)
            constexpr int b = decltype(BC)::value;
            XBGroup &q = grp[Y];
            if constexpr (b >= 15 && b + 1 < NB) {
                if (m == 0) {
                    constexpr int kb = b + 1;
                    // (a fragment is ONE 16-byte store of one producer thread: its first dword tells; every instruction
                    // here costs its full issue time - f32 MFMAs and VALU do not overlap on a SIMD, tools/ubench)
                    if (__builtin_amdgcn_ballot_w64(((__float_as_uint(a[X][kb & 15].x) ^ agen) & 1u) != 0) != 0) {
                        unsigned nspin = 0;
                        for (;;) {
                            if (!p_keep_waiting(nspin, p.spin_limit, p.ctl)) { failed = true; break; }
                            a[X][kb & 15] = x_buffer_load_b128(dz_rs, dz_vo, aso + kb * 1024, X_NT);
                            if (__builtin_amdgcn_ballot_w64(((__float_as_uint(a[X][kb & 15].x) ^ agen) & 1u) != 0) == 0) break;
                        }
                    }
                }
            }
            if constexpr (b >= B_NXT) {
                if (m == 1) a[Y][2 * (b - B_NXT)] = x_buffer_load_b128(dz_rs, dz_vo, nso + 2 * (b - B_NXT) * 1024, X_NT);
                else if (m == 2) a[Y][2 * (b - B_NXT) + 1] = x_buffer_load_b128(dz_rs, dz_vo, nso + (2 * (b - B_NXT) + 1) * 1024, X_NT);
            }
            if constexpr (POST) {
                if constexpr (b == 0) {             // Y's partial tiles (written before the last barrier)
                    if (m < 4) pl[m] = partY[(size_t)(m * 16 + i) * XB_LDP + ul];
                } else if constexpr (b == 1) {
                    if (m < 4) pr[m] = partY[(size_t)(m * 16 + i) * XB_LDP + 16 + ul];
                } else if constexpr (b == 2) {      // the operands of Y's gate derivatives (used from block 14 on)
                    if (m == 0) load_gates(Y, sy);
                    else if (m == 2) load_rest(Y, sy);
                } else if constexpr (b == 3) {
                    if (m == 0) q.dloc = (pl[0] + pl[1]) + (pl[2] + pl[3]);
                    else if (m == 1) psum = (pr[0] + pr[1]) + (pr[2] + pr[3]);
                } else if constexpr (b == 4) {      // the partner's share goes out
                    if (m == 0) x_buffer_store_b64((x_f32x2){psum, tag_y}, px_rs, pxsend, 0, X_SYS);
                } else if constexpr (b == B_RCV) {
                    if (m == 0) rv = x_buffer_load_b64(px_rs, pxrecv, 0, X_SYS);
                } else if constexpr (b == B_CHK) {  // the partner's contribution: normally there by now
                    if (m == 0) {
                        // first look OUTSIDE any loop: a wait inside a loop makes the wait-count pass assume the loop's own
                        // (youngest) request at the header too, i.e. vmcnt(0) - which also waits for the operand refills
                        // behind `rv`, the last of them one block old: ~700 cycles of idle matrix pipe per half step
                        if (__builtin_amdgcn_ballot_w64(__float_as_uint(rv.y) != (unsigned)sy) != 0) {
                            unsigned nspin = 0;
                            for (;;) {
                                if (!p_keep_waiting(nspin, p.spin_limit, p.ctl)) { failed = true; break; }
                                rv = x_buffer_load_b64(px_rs, pxrecv, 0, X_SYS);
                                if (__builtin_amdgcn_ballot_w64(__float_as_uint(rv.y) != (unsigned)sy) == 0) break;
                            }
                        }
                    } else if (m == 1) dh = q.dh + (q.dloc + rv.x);
                    else if (m == 2) ge = __builtin_amdgcn_exp2f(-2.88539008177792681f * fabsf(q.cn));        // lc_tanh(q.cn), in
                    else if (m == 3) tc = copysignf((1.0f - ge) * __builtin_amdgcn_rcpf(1.0f + ge), q.cn);     // two pieces
                } else if constexpr (b == B_DER) {  // explicit fma placement: see the forward step kernel
                    if (m == 1) do_pre = dh * tc * q.oa * (1.f - q.oa);
                    else if (m == 2) dcn = __builtin_fmaf(do_pre, wo, __builtin_fmaf(dh * q.oa, __builtin_fmaf(-tc, tc, 1.f), q.dc));
                    else if (m == 3) di_pre = dcn * q.ja * q.ia * (1.f - q.ia);
                    else if (m == 4) dj_pre = dcn * q.ia * __builtin_fmaf(-q.ja, q.ja, 1.f);
                    else if (m == 5) df_pre = dcn * q.cp * q.fa * (1.f - q.fa);
                    else if (m == 6) { odi = acty ? di_pre : 0.f; odj = acty ? dj_pre : 0.f; odf = acty ? df_pre : 0.f; odo = acty ? do_pre : 0.f; }
                    else if (m == 7) {
                        uacc(Y, odi, odj, odf, odo, q.cp, q.cn);
                        q.dc = acty ? __builtin_fmaf(df_pre, wf, __builtin_fmaf(di_pre, wi, dcn * q.fa)) : q.dc;
                    }
                } else if constexpr (b == B_PUB) {
                    // what the XCD's workgroups wait for goes out first: fragment [unit][row][4 gates], generation sy + 1
                    if (m == 1)
                        x_buffer_store_b128(p_with_lsb_tag(odi, odj, odf, odo, p_gen_bit((unsigned)sy + 1u)), dz_rs, dz_pub,
                                            (Y * 2 + (sy & 1)) * (XB_DZBUF * 4), 0);
                } else if constexpr (b == B_STO) {
                    if (valid[Y]) {
                        const int go = (int)((unsigned)ty * (unsigned)(B * G * 4));
                        if (m == 1) { x_buffer_store_b32(odi, g_rs, gvo[Y], go, 0); x_buffer_store_b32(odj, g_rs, gvo[Y] + 32, go, 0); }
                        else if (m == 2) { x_buffer_store_b32(odf, g_rs, gvo[Y] + 64, go, 0); x_buffer_store_b32(odo, g_rs, gvo[Y] + 96, go, 0); }
                    }
                }
            }
            if constexpr (b + 16 < NB) {            // block b + 16 into the register block b has just been read from
                if (m == 7) a[X][b & 15] = x_buffer_load_b128(dz_rs, dz_vo, aso + (b + 16) * 1024, X_NT);
            }
        };
#define LC_XMFMA(ACC, A, W) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "a"(W))
#define LC_XM(KB, M, ACC, A, W)                                                                                    \
    LC_XMFMA(ACC, A, W); mic(std::integral_constant<int, KB>(), M); __builtin_amdgcn_sched_barrier(0);
        // block KB out of a[X][KB & 15]
#define LC_XB(KB)                                                                                                  \
    LC_XM(KB, 0, acc[0][0], a[X][(KB) & 15].x, wreg[KB][0].x) LC_XM(KB, 1, acc[1][0], a[X][(KB) & 15].x, wreg[KB][1].x) \
    LC_XM(KB, 2, acc[0][1], a[X][(KB) & 15].y, wreg[KB][0].y) LC_XM(KB, 3, acc[1][1], a[X][(KB) & 15].y, wreg[KB][1].y) \
    LC_XM(KB, 4, acc[0][0], a[X][(KB) & 15].z, wreg[KB][0].z) LC_XM(KB, 5, acc[1][0], a[X][(KB) & 15].z, wreg[KB][1].z) \
    LC_XM(KB, 6, acc[0][1], a[X][(KB) & 15].w, wreg[KB][0].w) LC_XM(KB, 7, acc[1][1], a[X][(KB) & 15].w, wreg[KB][1].w)
        LC_XB(0) LC_XB(1) LC_XB(2) LC_XB(3) LC_XB(4) LC_XB(5) LC_XB(6) LC_XB(7)
        LC_XB(8) LC_XB(9) LC_XB(10) LC_XB(11) LC_XB(12) LC_XB(13) LC_XB(14) LC_XB(15)
        LC_XB(16) LC_XB(17) LC_XB(18) LC_XB(19)
        if constexpr (NKB > 5) { LC_XB(20) LC_XB(21) LC_XB(22) LC_XB(23) }
        if constexpr (NKB > 6) { LC_XB(24) LC_XB(25) LC_XB(26) LC_XB(27) }
        if constexpr (NKB > 7) { LC_XB(28) LC_XB(29) LC_XB(30) LC_XB(31) }
#undef LC_XB
#undef LC_XM
#undef LC_XMFMA
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 7" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]) : : "memory");
        if (X == 0) LC_XSTAMP(2); else LC_XSTAMP(10);
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                partX[(size_t)(wave * 16 + lk * 4 + r) * XB_LDP + c * 16 + li] = acc[c][0][r] + acc[c][1][r];
        if (failed) s_fail = 1;
        __syncthreads();
        if (X == 0) LC_XSTAMP(3); else LC_XSTAMP(11);
        return s_fail == 0;
    };
    {
        const std::integral_constant<int, 0> g0;
        const std::integral_constant<int, 1> g1;
        const std::true_type post;
        bool ok = T > 1 && half(g0, std::false_type(), 1) && half(g1, post, 1);
        for (int s = 2; ok && s < T; ++s) ok = half(g0, post, s) && half(g1, post, s);
    }
    if (T > 1 && !s_fail) {                    // the last product (group 1, step T - 1)
        const int Y = 1, sy = T - 1;
        const float *q = p_lds + (size_t)Y * (NWAVES * 16 * XB_LDP) + (size_t)i * XB_LDP + ul;
        const float tag_y = __uint_as_float((unsigned)sy);
        const int pxblk = (sy + 1) & 1;
        const int pxsend = (((xcc ^ 1) * 2 + Y) * 2 + pxblk) * XB_PXBLK + pxcell, pxrecv = ((xcc * 2 + Y) * 2 + pxblk) * XB_PXBLK + pxcell;
        grp[Y].dloc = (q[0] + q[(size_t)16 * XB_LDP]) + (q[(size_t)32 * XB_LDP] + q[(size_t)48 * XB_LDP]);
        const float drem = (q[16] + q[(size_t)16 * XB_LDP + 16]) + (q[(size_t)32 * XB_LDP + 16] + q[(size_t)48 * XB_LDP + 16]);
        x_buffer_store_b64((x_f32x2){drem, tag_y}, px_rs, pxsend, 0, X_SYS);
        unsigned nspin = 0;
        x_f32x2 rv;
        for (;;) {
            rv = x_buffer_load_b64(px_rs, pxrecv, 0, X_SYS);
            if (__builtin_amdgcn_ballot_w64(__float_as_uint(rv.y) != (unsigned)sy) == 0) break;
            if (!p_keep_waiting(nspin, p.spin_limit, p.ctl)) { failed = true; break; }
        }
        load_operands(Y, sy);
        derivs(Y, sy, grp[Y].dloc + rv.x);
        if (__syncthreads_or(failed ? 1 : 0)) s_fail = 1;
    }
    if (p.upg[dirx]) {                         // rows past B carried zeros all along: every one of the 64 row slots is written
#pragma unroll
        for (int sg = 0; sg < 2; ++sg) {
            float *o = p.upg[dirx] + (size_t)(rh * 32 + sg * 16 + i) * 7 * N + n;
#pragma unroll
            for (int k = 0; k < 7; ++k) o[(size_t)k * N] = ug[sg][k];
        }
    }
    if (s_fail && threadIdx.x == 0) p_report_failure(p.ctl);
}

#include "lstm_pair_x3.inc"

inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
// The persistent schedules hard-wire the MI355X SPX topology (8 XCCs x 32 CUs, workgroup -> XCC round robin, one
// slice per CU): anything else runs the launch train.  Cached per device.
inline bool persist_device_ok()
{
    static std::mutex mu;
    static int state[16] = {0};                       // 0 = unknown, 1 = ok, -1 = no
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) { (void)hipGetLastError(); return false; }
    std::lock_guard<std::mutex> lock(mu);
    if (state[dev] == 0) {
        hipDeviceProp_t prop;
        int xccs = 0;
        const bool ok = hipGetDeviceProperties(&prop, dev) == hipSuccess &&
                        hipDeviceGetAttribute(&xccs, hipDeviceAttributeNumberOfXccs, dev) == hipSuccess &&
                        strncmp(prop.gcnArchName, "gfx950", 6) == 0 && prop.multiProcessorCount == 256 && xccs == 8;
        if (!ok) (void)hipGetLastError();
        state[dev] = ok ? 1 : -1;
    }
    return state[dev] == 1;
}
inline unsigned persist_spin_limit()
{
    const long v = lc_option(LC_OPT_LSTM_SPIN_LIMIT, P_SPIN_LIMIT);   // read per call (tests force the timeout path)
    return v < 0 ? 0u : (unsigned)v;
}
thread_local int g_last_sched = 0;
// Geometry of the persistent schedule, or false when the shape does not qualify (then the launch train runs).
inline bool persist_geom(int T, int B, int N, int ndir, bool bwd, PGeom &g, size_t &lds_bytes)
{
    // read per call: the tests compare the two schedules, and a failed launch is re-run with the override set
    if (lc_option(LC_OPT_LSTM_PERSISTENT, 1) == 0 || N > P_MAXN || N % 16 != 0 || T < 4 || !persist_device_ok()) return false;
    g.T = T; g.B = B; g.N = N; g.ndir = ndir;
    g.gpd = 8 / ndir;
    g.rpg = lc_cdiv(B, g.gpd);
    if (g.rpg > 16) return false;
    g.upw = (lc_cdiv(N, 32) + 3) & ~3;
    g.UP = g.upw;
    g.nwg = lc_cdiv(N, g.upw);
    // LDS holds only the partial tiles; the request is nevertheless more than half a CU's LDS so that two slices can
    // never share a CU (and its matrix pipe) while other CUs idle - 76 KB stay free for co-resident GEMM workgroups
    (void)bwd;
    lds_bytes = 84 * 1024;
    return true;
}
inline bool persist_geom_bf16(int T, int B, int N, int ndir, PGeom &g, size_t &lds_bytes)
{
    if (lc_option(LC_OPT_LSTM_PERSISTENT, 1) == 0 || N > 1024 || N % 32 != 0 || T < 4 || !persist_device_ok()) return false;
    g.T = T; g.B = B; g.N = N; g.ndir = ndir;
    g.gpd = 8 / ndir;
    g.rpg = lc_cdiv(B, g.gpd);
    if (g.rpg > 16) return false;
    g.upw = (lc_cdiv(N, 32) + 3) & ~3;
    g.UP = g.upw;
    g.nwg = lc_cdiv(N, g.upw);
    lds_bytes = 84 * 1024;
    return true;
}
// The XCD-pair schedule (fp32, 1024-unit layers: config c4).  One launch drives two "direction slots" of four XCDs, 64
// batch rows each.  Both directions of a layer: slot d = direction d, and batches of more than 64 rows run as
// ceil(B / 64) launches back to back over 64-row blocks (the tensors keep their [T, B, *] layout: a slot has a row base,
// B stays the frame stride).  One direction (lc_lstm_fwd with ndir = 1): the two slots are two 64-row blocks of the same
// direction, 128 rows per launch.
// widths the pair kernels are instantiated for: N = 128 NKB, NKB = 5 .. 8 (below 640 the single-XCD schedule holds R)
inline bool pair_width(int N) { return N == 640 || N == 768 || N == 896 || N == 1024; }
inline bool pair_x3_width(int N) { return N == 768 || N == 1024; }     // ... and for the split-operand kernels (whole 32-blocks)
inline bool persist_x3_width(int N) { return N >= 64 && N <= 512 && N % 64 == 0; }     // single-XCD kernels, split operands
inline bool pair_geom(int T, int B, int N, int ndir)
{
    // (the kernels address their [T, B, 4N] tensors with unsigned 32-bit scalar frame offsets)
    return lc_option(LC_OPT_LSTM_PERSISTENT, 1) != 0 && pair_width(N) && (ndir == 1 || ndir == 2) && B >= 1 && T >= 4 &&
           (unsigned long long)T * B * 4 * N * sizeof(float) <= 0xffffffffull && persist_device_ok();
}
inline int pair_rows_per_launch(int ndir) { return ndir == 2 ? 64 : 128; }
inline size_t pair_fwd_ws_bytes() { return P_CTL_BYTES + (X_HX_FLOATS + X_PX_FLOATS) * sizeof(float); }
inline size_t pair_bwd_ws_bytes() { return P_CTL_BYTES + (XB_DZX_FLOATS + XB_PX_FLOATS) * sizeof(float); }
inline size_t pair_bwd_x3_ws_bytes() { return P_CTL_BYTES + (X3B_DZX_FLOATS + XB_PX_FLOATS) * sizeof(float); }     // producer-split pieces: 1.5 x
inline size_t persist_ws_bytes(int N, bool bwd)
{
    if (N > 1024 || N % 16 != 0) return 0;
    return P_CTL_BYTES + (size_t)8 * 2 * (bwd ? 4 * N : 2 * N) * 16 * sizeof(float);
}
inline size_t persist_bwd_x3_ws_bytes(int N) { return P_CTL_BYTES + (size_t)8 * 2 * x3p_bwd_buf_bytes(N); }     // producer-split pieces: 1.5 x
inline size_t upg_part_bytes(int N) { return al256((size_t)2 * UPG_SPLITS * 7 * N * sizeof(float)); }   // x 2: both directions of the XCD-pair BPTT at once

// rows of the transposed [K][Bpad] state buffers: covers every row a workgroup row-tile touches
inline int bpad(int B) { return B <= 16 ? 16 : (B <= 64 ? ((B + 31) & ~31) : ((B + 63) & ~63)); }

// Clears what a persistent launch needs cleared: the per-launch part of the control block (NOT the sticky status word)
// and the exchange buffers behind it.
inline bool persist_clear(void *workspace, size_t ws_bytes, hipStream_t s, int fill = 0)
{
    // (fill = 0xff: the bf16 BPTT's exchange buffers start out as "not arrived" sentinels)
    return hipMemsetAsync(workspace, 0, LC_LSTM_STATUS_OFFSET, s) == hipSuccess &&
           hipMemsetAsync((char *)workspace + P_CTL_BYTES, fill, ws_bytes - P_CTL_BYTES, s) == hipSuccess;
}

template <class K, class A>
inline bool persist_launch(K kernel, size_t lds, hipStream_t s, const A &args)
{
    if (hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return false;
    hipLaunchKernelGGL(kernel, dim3(P_GRID), dim3(P_THREADS), lds, s, args);
    return true;
}

}  // namespace

// Development hook (not part of the product surface): device buffer of [T][4 waves][8] s_memtime stamps.
extern "C" void lc_debug_set_lstm_stamps(unsigned long long *buf) { g_lstm_dbg = buf; }
extern "C" int lc_debug_last_lstm_schedule(void) { return g_last_sched; }

// Workspace layout: [control block P_CTL_BYTES][schedule-specific buffers]; the backward one ends with the partial
// sums of the bias / peephole gradient reduce.
extern "C" size_t lc_lstm_fwd_workspace_bytes(int B, int N, int ndir)
{
    size_t need = P_CTL_BYTES + (size_t)ndir * (al256((size_t)2 * N * bpad(B) * sizeof(float)) +
                                                al256((size_t)N * 4 * N * sizeof(float)));
    if (al256(persist_ws_bytes(N, false)) > need) need = al256(persist_ws_bytes(N, false));
    if (pair_width(N) && al256(pair_fwd_ws_bytes()) > need) need = al256(pair_fwd_ws_bytes());
    return need;
}
static size_t lstm_bwd_main_bytes(int B, int N, int ndir)
{
    size_t need = P_CTL_BYTES + (size_t)ndir * (al256((size_t)2 * 4 * N * bpad(B) * sizeof(float)) +
                                                al256((size_t)B * N * sizeof(float)) +
                                                al256((size_t)N * 4 * N * sizeof(float)));
    if (al256(persist_ws_bytes(N, true)) > need) need = al256(persist_ws_bytes(N, true));
    if (pair_width(N) && al256(pair_bwd_ws_bytes()) > need) need = al256(pair_bwd_ws_bytes());
    if (pair_x3_width(N) && al256(pair_bwd_x3_ws_bytes()) > need) need = al256(pair_bwd_x3_ws_bytes());
    if (persist_x3_width(N) && al256(persist_bwd_x3_ws_bytes(N)) > need) need = al256(persist_bwd_x3_ws_bytes(N));
    return need;
}
extern "C" size_t lc_lstm_bwd_workspace_bytes(int B, int N, int ndir)
{
    return lstm_bwd_main_bytes(B, N, ndir) + upg_part_bytes(N);
}

// Second in-order queue (per device) for the two-stream forward schedule, plus the fork / join events.  Two internal
// streams of different priority are kept so that one can always be picked whose priority differs from the caller's
// stream - streams of different priority never share a hardware queue.  Creation is serialised; a call only reads
// its device's entry afterwards (two host threads driving the same device's two-stream schedule at once would share
// the fork / join events: callers that do that must serialise those calls themselves - see lstm_ctc_hip.h).
struct DirStreams {
    hipStream_t s_hi, s_lo;
    int prio_hi;
    hipEvent_t fork, join;
};
static DirStreams *dir_streams(hipStream_t caller, hipStream_t *second)
{
    static DirStreams pool[16];
    static int state[16] = {0};                     // 0 = untried, 1 = ready, -1 = unavailable
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    DirStreams &d = pool[dev];
    {
        std::lock_guard<std::mutex> lock(mu);
        if (state[dev] == 0) {
            int lo = 0, hi = 0;
            const bool ok = hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && lo != hi &&
                            hipStreamCreateWithPriority(&d.s_hi, hipStreamNonBlocking, hi) == hipSuccess &&
                            hipStreamCreateWithPriority(&d.s_lo, hipStreamNonBlocking, lo) == hipSuccess &&
                            hipEventCreateWithFlags(&d.fork, hipEventDisableTiming) == hipSuccess &&
                            hipEventCreateWithFlags(&d.join, hipEventDisableTiming) == hipSuccess;
            d.prio_hi = hi;
            state[dev] = ok ? 1 : -1;
            if (!ok) (void)hipGetLastError();
        }
        if (state[dev] != 1) return nullptr;
    }
    int prio = 0;
    if (hipStreamGetPriority(caller, &prio) != hipSuccess) { (void)hipGetLastError(); prio = 0; }
    *second = (prio == d.prio_hi) ? d.s_lo : d.s_hi;
    return &d;
}

// Packs a [K, C] row-major weight into the step-GEMM operand layout (f32 K16 or bf16 K32), once per call.
static void pack_operand(bool bf, const float *W, int K, int C, void *dst, hipStream_t s)
{
    if (bf) hipLaunchKernelGGL(pack_k32_bf16_kernel, dim3(1024), dim3(256), 0, s, W, K, C, (unsigned short *)dst);
    else hipLaunchKernelGGL(pack_k16_kernel, dim3(1024), dim3(256), 0, s, W, K, C, (float *)dst);
}

template <bool BF>
static void launch_fwd_step(int mt, dim3 grid, hipStream_t s, const FwdArgs &a)
{
    dim3 block(NTHREADS);
    switch (mt) {
    case 1: hipLaunchKernelGGL((lstm_fwd_step_kernel<1, BF>), grid, block, 0, s, a); break;
    case 2: hipLaunchKernelGGL((lstm_fwd_step_kernel<2, BF>), grid, block, 0, s, a); break;
    case 3: hipLaunchKernelGGL((lstm_fwd_step_kernel<3, BF>), grid, block, 0, s, a); break;
    default: hipLaunchKernelGGL((lstm_fwd_step_kernel<4, BF>), grid, block, 0, s, a); break;
    }
}

// x3: the step product as fp32-on-bf16x3 (lstm_pair_x3.inc) where a split-operand kernel exists for the shape - the XCD-pair
// schedule at N = 768 / 1024, the single-XCD schedule at N = 64 .. 512 in steps of 64 -; every other shape, and the launch-train fall-back, runs the fp32 kernels (same arithmetic in
// another summation order).
static int lstm_fwd_impl(bool bf, bool x3, const char *who, const lc_lstm_fwd_dir_t *dirs, int ndir, const int *seq_len, int T,
                         int B, int N, float forget_bias, void *workspace, size_t workspace_bytes, lc_stream_t stream)
{
    LC_CHECK_ARG(dirs && seq_len && workspace, "%s: null pointer", who);
    LC_CHECK_ARG(ndir == 1 || ndir == 2, "%s: ndir must be 1 or 2", who);
    LC_CHECK_ARG(T > 0 && B > 0 && N > 0 && N % (bf ? 32 : 16) == 0, "%s: need T,B > 0 and num_neurons %% %d == 0 (N=%d)",
                 who, bf ? 32 : 16, N);
    if (workspace_bytes < lc_lstm_fwd_workspace_bytes(B, N, ndir)) {
        lc_set_error("%s: workspace too small", who);
        return LC_EWORKSPACE;
    }
    for (int i = 0; i < ndir; ++i)
        LC_CHECK_ARG(dirs[i].zx && dirs[i].R && dirs[i].cs && dirs[i].hs, "%s: null pointer in dirs[%d]", who, i);
    hipStream_t s = (hipStream_t)stream;
    if (!bf && pair_geom(T, B, N, ndir)) {
        XFwdArgs xa;
        for (int i = 0; i < 2; ++i) {
            const lc_lstm_fwd_dir_t &di = dirs[ndir == 2 ? i : 0];
            xa.d[i].zx = di.zx; xa.d[i].R = di.R;
            xa.d[i].w_f = di.w_f; xa.d[i].w_i = di.w_i; xa.d[i].w_o = di.w_o;
            xa.d[i].cs = di.cs; xa.d[i].hs = di.hs; xa.d[i].hT = nullptr; xa.d[i].reverse = di.reverse;
        }
        xa.seq_len = seq_len; xa.T = T; xa.B = B; xa.forget_bias = forget_bias;
        xa.spin_limit = persist_spin_limit();
        xa.ctl = (PCtl *)workspace;
        xa.hx = (float *)((char *)workspace + P_CTL_BYTES);
        xa.px = xa.hx + X_HX_FLOATS;
        xa.dbg = g_lstm_dbg;
        for (int r0 = 0; r0 < B; r0 += pair_rows_per_launch(ndir)) {       // 64-row blocks (per slot), back to back
            xa.row_base[0] = r0;
            xa.row_base[1] = ndir == 2 ? r0 : r0 + 64;                     // (a slot whose block starts past B idles: all rows masked)
            if (!persist_clear(workspace, pair_fwd_ws_bytes(), s)) {
                lc_set_error("%s: memset failed", who);
                return LC_ELAUNCH;
            }
            const bool use_x3 = x3 && pair_x3_width(N);
            const bool launched = use_x3 ? (N == 1024 ? persist_launch(lstm_fwd_pair_x3_kernel<8>, x3_fwd_lds_bytes(8), s, xa)
                                                      : persist_launch(lstm_fwd_pair_x3_kernel<6>, x3_fwd_lds_bytes(6), s, xa))
                                : N == 1024 ? persist_launch(lstm_fwd_pair_kernel<8>, (size_t)84 * 1024, s, xa)
                                : N == 896 ? persist_launch(lstm_fwd_pair_kernel<7>, (size_t)84 * 1024, s, xa)
                                : N == 768 ? persist_launch(lstm_fwd_pair_kernel<6>, (size_t)84 * 1024, s, xa)
                                           : persist_launch(lstm_fwd_pair_kernel<5>, (size_t)84 * 1024, s, xa);
            if (!launched) {
                lc_set_error("%s: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed for the XCD-pair kernel", who);
                (void)hipGetLastError();
                return LC_ELAUNCH;
            }
            PVerifyArgs va;
            va.ctl = xa.ctl; va.nused = 8; va.nwg = N / 32; va.nout = ndir;
            va.out[0] = dirs[0].hs; va.out[1] = dirs[ndir - 1].hs; va.count = va.count16 = (size_t)T * B * N;
            va.out16[0] = va.out16[1] = nullptr;
            hipLaunchKernelGGL(persist_verify_kernel, dim3(256), dim3(256), 0, s, va);
            LC_CHECK_LAUNCH("lstm_fwd_pair");
        }
        g_last_sched = x3 && pair_x3_width(N) ? 6 : 5;
        return LC_OK;
    }
    PFwdArgs pa;
    size_t lds = 0;
    const bool persist = bf ? persist_geom_bf16(T, B, N, ndir, pa.g, lds) : persist_geom(T, B, N, ndir, false, pa.g, lds);
    if (persist) {
        for (int i = 0; i < ndir; ++i) {
            pa.d[i].zx = dirs[i].zx; pa.d[i].R = dirs[i].R;
            pa.d[i].w_f = dirs[i].w_f; pa.d[i].w_i = dirs[i].w_i; pa.d[i].w_o = dirs[i].w_o;
            pa.d[i].cs = dirs[i].cs; pa.d[i].hs = dirs[i].hs; pa.d[i].hT = nullptr; pa.d[i].reverse = dirs[i].reverse;
            pa.d[i].hs16 = bf ? dirs[i].hs_bf16 : nullptr;
        }
        if (ndir == 1) pa.d[1] = pa.d[0];
        pa.seq_len = seq_len; pa.forget_bias = forget_bias;
        pa.spin_limit = persist_spin_limit();
        pa.ctl = (PCtl *)workspace;
        pa.hT = (float *)((char *)workspace + P_CTL_BYTES);
        pa.dbg = bf ? nullptr : g_lstm_dbg;
        if (!persist_clear(workspace, persist_ws_bytes(N, false), s, bf ? 0xff : 0)) {
            lc_set_error("%s: memset failed", who);
            return LC_ELAUNCH;
        }
        bool ok = false;
        const bool px3 = !bf && x3 && persist_x3_width(N);
        if (px3) {                             // split-operand forward kernel: whole 32-blocks per wave, whole tiles per workgroup
#define LC_PFX(NB) case NB: ok = persist_launch(lstm_fwd_persist_x3_kernel<NB>, lds, s, pa); break;
            switch (N / 32) { LC_PFX(2) LC_PFX(4) LC_PFX(6) LC_PFX(8) LC_PFX(10) LC_PFX(12) LC_PFX(14) LC_PFX(16) }
#undef LC_PFX
        } else if (!bf) {
            const int per = lc_cdiv(N / 16, NWAVES);
#define LC_PFWD(PER)                                                                                                   \
    case PER:                                                                                                          \
        ok = (N % 64) ? persist_launch(lstm_fwd_persist_kernel<PER, true>, lds, s, pa)                                 \
                      : persist_launch(lstm_fwd_persist_kernel<PER, false>, lds, s, pa);                               \
        break;
            switch (per) { LC_PFWD(1) LC_PFWD(2) LC_PFWD(3) LC_PFWD(4) LC_PFWD(5) LC_PFWD(6) LC_PFWD(7) LC_PFWD(8) }
#undef LC_PFWD
        } else {
            const int perb = lc_cdiv(N / 32, NWAVES);
#define LC_PFB(PERB, PPT)                                                                                              \
    case PERB:                                                                                                         \
        ok = (N % 128) ? persist_launch(lstm_fwd_persist_bf16_kernel<PERB, PPT, true>, lds, s, pa)                     \
                       : persist_launch(lstm_fwd_persist_bf16_kernel<PERB, PPT, false>, lds, s, pa);                   \
        break;
            // shadow_only (every direction, each with its hs_bf16): the fp32 hs is not written - the full-width kernel only
            bool shonly = N == 1024;
            for (int i = 0; i < ndir; ++i) shonly = shonly && dirs[i].shadow_only && dirs[i].hs_bf16;
            if (shonly) ok = persist_launch(lstm_fwd_persist_bf16_kernel<8, 2, false, true>, lds, s, pa);
            else
            switch (perb) { LC_PFB(1, 1) LC_PFB(2, 1) LC_PFB(3, 1) LC_PFB(4, 1) LC_PFB(5, 2) LC_PFB(6, 2) LC_PFB(7, 2) LC_PFB(8, 2) }
#undef LC_PFB
        }
        if (!ok) {
            lc_set_error("%s: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed for the persistent kernel", who);
            (void)hipGetLastError();
            return LC_ELAUNCH;
        }
        PVerifyArgs va;
        va.ctl = pa.ctl; va.nused = ndir * pa.g.gpd; va.nwg = pa.g.nwg; va.nout = ndir;
        va.out[0] = dirs[0].hs; va.out[1] = dirs[ndir - 1].hs; va.count = va.count16 = (size_t)T * B * N;
        va.out16[0] = bf ? (unsigned short *)dirs[0].hs_bf16 : nullptr;
        va.out16[1] = bf ? (unsigned short *)dirs[ndir - 1].hs_bf16 : nullptr;
        hipLaunchKernelGGL(persist_verify_kernel, dim3(256), dim3(256), 0, s, va);
        LC_CHECK_LAUNCH(bf ? "lstm_fwd_persist_bf16" : "lstm_fwd_persist");
        g_last_sched = (bf ? 2 : px3 ? 7 : 1) | ((int)bf << 16);
        return LC_OK;
    }
    FwdArgs a;
    a.seq_len = seq_len; a.T = T; a.B = B; a.N = N; a.Bpad = bpad(B); a.forget_bias = forget_bias;
    a.dbg = g_lstm_dbg;
    char *w = (char *)workspace + P_CTL_BYTES;
    for (int i = 0; i < ndir; ++i) {
        a.d[i].zx = dirs[i].zx;
        a.d[i].w_f = dirs[i].w_f; a.d[i].w_i = dirs[i].w_i; a.d[i].w_o = dirs[i].w_o;
        a.d[i].cs = dirs[i].cs; a.d[i].hs = dirs[i].hs; a.d[i].reverse = dirs[i].reverse;
        a.d[i].hT = (float *)w;
        const size_t hbytes = al256((size_t)2 * N * a.Bpad * sizeof(float));
        // pad rows of hT (b >= B) are never written by the kernel but are read as MFMA operands
        if (hipMemsetAsync(w, 0, hbytes, s) != hipSuccess) {
            lc_set_error("%s: memset failed", who);
            return LC_ELAUNCH;
        }
        w += hbytes;
        pack_operand(bf, dirs[i].R, N, 4 * N, w, s);
        a.d[i].R = (const float *)w;
        w += al256((size_t)N * 4 * N * sizeof(float));
    }
    if (ndir == 1) a.d[1] = a.d[0];
    LC_CHECK_LAUNCH("pack_operand");
    a.row_base = 0;
    // Two-stream schedule for the big fp32 bidirectional case (c4: N >= 1024, 33..64 rows): one in-order chain of
    // 32-row-tile launches per direction, the reverse direction on a second stream.  The chains drift apart, so on
    // every CU one direction's MFMA phase runs under the other's launch boundary / operand latency / gate epilogue:
    // 13.5 vs 14.8 us per step pair.  Per-row arithmetic is unchanged (same K order), so results are bit-identical.
    // 57.6 vs 68.5 ms per c4 step.  Not for small steps (two launches per step make N = 320 / 512 host-bound: 8.5 vs
    // 4.4 us) nor for bf16 (its 8 us step leaves the host < 4 us per launch: the c5 step got 159 vs 153.5 ms).  The second stream is created with HIGH priority so that it can
    // never share a hardware queue with the caller's stream: two streams on one queue serialise (measured 2x).
    if (!bf && ndir == 2 && a.Bpad == 64 && N >= 1024) {
        hipStream_t s2 = nullptr;
        DirStreams *ds = dir_streams(s, &s2);
        if (ds) {
            (void)hipEventRecord(ds->fork, s);
            (void)hipStreamWaitEvent(s2, ds->fork, 0);
            FwdArgs a0 = a, a1 = a;
            a0.d[1] = a0.d[0];
            a1.d[0] = a.d[1];
            a1.dbg = nullptr;
            dim3 g1(N / 8, lc_cdiv(B, 32), 1);
            for (int step = 0; step < T; ++step) {
                a0.step = a1.step = step;
                launch_fwd_step<false>(2, g1, s, a0);
                launch_fwd_step<false>(2, g1, s2, a1);
            }
            (void)hipEventRecord(ds->join, s2);
            (void)hipStreamWaitEvent(s, ds->join, 0);
            LC_CHECK_LAUNCH("lstm_fwd_step");
            g_last_sched = 3 | (2 << 8);
            return LC_OK;
        }
    }
    // Row tile: 64 rows per workgroup when that already fills the chip (N = 1024: 256 workgroups; more, smaller
    // tiles re-stream R and measured 18.5 vs 15.1 us), halved while the grid would leave CUs idle (N = 320 / 512 at
    // B = 32: 5.4 -> 4.7 and 6.3 -> 5.3 us per step).
    int mt = a.Bpad >= 64 ? 4 : a.Bpad / 16;
    while (mt > 1 && (long long)(N / 8) * lc_cdiv(B, 16 * mt) * ndir < 200) mt = (mt + 1) / 2;
    dim3 grid(N / 8, lc_cdiv(B, 16 * mt), ndir);
    for (int step = 0; step < T; ++step) {
        a.step = step;
        if (bf) launch_fwd_step<true>(mt, grid, s, a);
        else launch_fwd_step<false>(mt, grid, s, a);
    }
    LC_CHECK_LAUNCH("lstm_fwd_step");
    g_last_sched = 4 | (mt << 8) | ((int)bf << 16);
    return LC_OK;
}

static int lstm_bwd_impl(bool bf, bool x3, const char *who, const lc_lstm_bwd_dir_t *dirs, int ndir, const int *seq_len, int T,
                         int B, int N, void *workspace, size_t workspace_bytes, lc_stream_t stream)
{
    LC_CHECK_ARG(dirs && seq_len && workspace, "%s: null pointer", who);
    LC_CHECK_ARG(ndir == 1 || ndir == 2, "%s: ndir must be 1 or 2", who);
    LC_CHECK_ARG(T > 0 && B > 0 && N > 0 && N % (bf ? 32 : 16) == 0, "%s: need T,B > 0 and num_neurons %% %d == 0 (N=%d)",
                 who, bf ? 32 : 16, N);
    if (workspace_bytes < lc_lstm_bwd_workspace_bytes(B, N, ndir)) {
        lc_set_error("%s: workspace too small", who);
        return LC_EWORKSPACE;
    }
    for (int i = 0; i < ndir; ++i)
        LC_CHECK_ARG(dirs[i].gates && dirs[i].RT && dirs[i].cs && dirs[i].dh, "%s: null pointer in dirs[%d]", who, i);
    hipStream_t s = (hipStream_t)stream;
    float *upg_part = (float *)((char *)workspace + lstm_bwd_main_bytes(B, N, ndir));
    PBwdArgs pa;
    size_t plds = 0;
    const bool pair = !bf && pair_geom(T, B, N, ndir);
    const bool persist = !pair && (bf ? persist_geom_bf16(T, B, N, ndir, pa.g, plds) : persist_geom(T, B, N, ndir, true, pa.g, plds));
    if (pair) {
        XBwdArgs xa;
        for (int i = 0; i < 2; ++i) {
            const lc_lstm_bwd_dir_t &di = dirs[ndir == 2 ? i : 0];
            xa.d[i].gates = di.gates; xa.d[i].RT = di.RT;
            xa.d[i].w_f = di.w_f; xa.d[i].w_i = di.w_i; xa.d[i].w_o = di.w_o;
            xa.d[i].cs = di.cs; xa.d[i].dh = di.dh; xa.d[i].dc = nullptr; xa.d[i].dzT = nullptr;
            xa.d[i].reverse = di.reverse;
            // split-operand kernel: the x3 shadow of dz (lc_lstm_bwd_x3), written by the producers that split dz anyway -
            // while its byte offsets fit the 32-bit buffer addressing of the kernel
            xa.d[i].dz16 = (x3 && pair_x3_width(N) && (unsigned long long)T * B * 12 * N * 2 <= 0x7fffffffull) ? di.dz_bf16 : nullptr;
        }
        xa.seq_len = seq_len; xa.T = T; xa.B = B;
        xa.spin_limit = persist_spin_limit();
        xa.ctl = (PCtl *)workspace;
        const bool use_x3 = x3 && pair_x3_width(N);
        xa.dzx = (float *)((char *)workspace + P_CTL_BYTES);
        xa.px = xa.dzx + (use_x3 ? X3B_DZX_FLOATS : XB_DZX_FLOATS);
        for (int i = 0; i < 2; ++i) {
            const lc_lstm_bwd_dir_t &di = dirs[ndir == 2 ? i : 0];
            const bool want = (di.dpeep && di.w_f) || di.dbias;
            xa.upg[i] = want ? upg_part + (size_t)i * UPG_SPLITS * 7 * N : nullptr;
        }
        xa.dbg = g_lstm_dbg;
        for (int r0 = 0; r0 < B; r0 += pair_rows_per_launch(ndir)) {       // 64-row blocks (per slot), back to back
            xa.row_base[0] = r0;
            xa.row_base[1] = ndir == 2 ? r0 : r0 + 64;
            if (!persist_clear(workspace, use_x3 ? pair_bwd_x3_ws_bytes() : pair_bwd_ws_bytes(), s)) {
                lc_set_error("%s: memset failed", who);
                return LC_ELAUNCH;
            }
            const bool launched = use_x3 ? (N == 1024 ? persist_launch(lstm_bwd_pair_x3_kernel<8>, x3_bwd_lds_bytes(8), s, xa)
                                                      : persist_launch(lstm_bwd_pair_x3_kernel<6>, x3_bwd_lds_bytes(6), s, xa))
                                : N == 1024 ? persist_launch(lstm_bwd_pair_kernel<8>, (size_t)84 * 1024, s, xa)
                                : N == 896 ? persist_launch(lstm_bwd_pair_kernel<7>, (size_t)84 * 1024, s, xa)
                                : N == 768 ? persist_launch(lstm_bwd_pair_kernel<6>, (size_t)84 * 1024, s, xa)
                                           : persist_launch(lstm_bwd_pair_kernel<5>, (size_t)84 * 1024, s, xa);
            if (!launched) {
                lc_set_error("%s: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed for the XCD-pair kernel", who);
                (void)hipGetLastError();
                return LC_ELAUNCH;
            }
            PVerifyArgs va;
            va.ctl = xa.ctl; va.nused = 8; va.nwg = N / 32; va.nout = ndir;
            va.out[0] = dirs[0].gates; va.out[1] = dirs[ndir - 1].gates; va.count = (size_t)T * B * 4 * N;
            // the x3 shadow the producers wrote themselves (no split pass follows it: schedule-word bit 18) is poisoned with
            // the fp32 dz - a caller of the C ABI that ignores the status word must not find finite terms of a failed launch
            va.count16 = 3 * va.count;
            va.out16[0] = (unsigned short *)xa.d[0].dz16; va.out16[1] = (unsigned short *)xa.d[ndir - 1].dz16;
            hipLaunchKernelGGL(persist_verify_kernel, dim3(256), dim3(256), 0, s, va);
            LC_CHECK_LAUNCH("lstm_bwd_pair");
            // the kernel left the block's per-row partials of the bias / peephole gradients: fold them into (+=) the outputs
            for (int i = 0; i < 2; ++i) {
                const lc_lstm_bwd_dir_t &di = dirs[ndir == 2 ? i : 0];
                if (xa.upg[i])
                    hipLaunchKernelGGL(unit_param_fold_kernel, dim3(lc_cdiv(N, 256)), dim3(256), 0, s, xa.upg[i], UPG_SPLITS, N,
                                       (di.dpeep && di.w_f) ? di.dpeep : nullptr, di.dbias);
            }
            LC_CHECK_LAUNCH("unit_param_fold");
        }
        g_last_sched = (x3 && pair_x3_width(N) ? 6 : 5) | (1 << 17) | ((xa.d[0].dz16 ? 1 : 0) << 18);
        return LC_OK;
    } else if (persist) {
        for (int i = 0; i < ndir; ++i) {
            pa.d[i].gates = dirs[i].gates; pa.d[i].RT = dirs[i].RT;
            pa.d[i].w_f = dirs[i].w_f; pa.d[i].w_i = dirs[i].w_i; pa.d[i].w_o = dirs[i].w_o;
            pa.d[i].cs = dirs[i].cs; pa.d[i].dh = dirs[i].dh; pa.d[i].dc = nullptr; pa.d[i].dzT = nullptr;
            pa.d[i].dz16 = bf ? dirs[i].dz_bf16 : nullptr;
            pa.d[i].reverse = dirs[i].reverse;
        }
        // split-operand kernel: dz_bf16 is the x3 shadow of dz (lc_lstm_bwd_x3), written by the producers that split dz anyway -
        // while its byte offsets fit the 32-bit buffer addressing of the kernel
        const bool px3 = !bf && x3 && persist_x3_width(N);
        const bool shadow3 = px3 && (unsigned long long)T * B * 12 * N * 2 <= 0x7fffffffull;
        if (shadow3)
            for (int i = 0; i < ndir; ++i) pa.d[i].dz16 = (unsigned short *)dirs[i].dz_bf16;
        if (ndir == 1) pa.d[1] = pa.d[0];
        pa.seq_len = seq_len;
        pa.spin_limit = persist_spin_limit();
        pa.ctl = (PCtl *)workspace;
        pa.dzT = (float *)((char *)workspace + P_CTL_BYTES);
        for (int i = 0; i < 2; ++i) {            // per-row partial bias / peephole gradients: [B][7][N] per direction
            const bool want = i < ndir && ((dirs[i].dpeep && dirs[i].w_f) || dirs[i].dbias);
            pa.upg[i] = want ? upg_part + (size_t)i * B * 7 * N : nullptr;
        }
        if (ndir == 1) pa.upg[1] = pa.upg[0];
        pa.dbg = g_lstm_dbg;
        if (!persist_clear(workspace, px3 ? persist_bwd_x3_ws_bytes(N) : persist_ws_bytes(N, true), s, bf ? 0xff : 0)) {
            lc_set_error("%s: memset failed", who);
            return LC_ELAUNCH;
        }
        bool ok = false;
        if (px3) {                              // split-operand BPTT: the producers split, pieces of 32-blocks
#define LC_PBX(NBW) case NBW: ok = persist_launch(lstm_bwd_persist_x3_kernel<NBW>, plds, s, pa); break;
            switch (N / 32) { LC_PBX(2) LC_PBX(4) LC_PBX(6) LC_PBX(8) LC_PBX(10) LC_PBX(12) LC_PBX(14) LC_PBX(16) }
#undef LC_PBX
        } else if (bf) {
            const int nch = lc_cdiv(N, 256);
#define LC_PBB(NCH, PPT)                                                                                               \
    case NCH:                                                                                                          \
        ok = (N % 256) ? persist_launch(lstm_bwd_persist_bf16_kernel<NCH, PPT, true>, plds, s, pa)                     \
                       : persist_launch(lstm_bwd_persist_bf16_kernel<NCH, PPT, false>, plds, s, pa);                   \
        break;
            bool shonly = N == 1024;            // shadow_only: the fp32 dz is not written (see lc_lstm_fwd_bf16)
            for (int i = 0; i < ndir; ++i) shonly = shonly && dirs[i].shadow_only && dirs[i].dz_bf16;
            if (shonly) ok = persist_launch(lstm_bwd_persist_bf16_kernel<4, 2, false, true>, plds, s, pa);
            else
            switch (nch) { LC_PBB(1, 1) LC_PBB(2, 1) LC_PBB(3, 2) LC_PBB(4, 2) }
#undef LC_PBB
        } else {
            const int nq = lc_cdiv(lc_cdiv(4 * N / 16, NWAVES), 4);
#define LC_PBWD(NQ)                                                                                                    \
    case NQ:                                                                                                           \
        ok = (N % 64) ? persist_launch(lstm_bwd_persist_kernel<NQ, true>, plds, s, pa)                                 \
                      : persist_launch(lstm_bwd_persist_kernel<NQ, false>, plds, s, pa);                               \
        break;
            switch (nq) { LC_PBWD(1) LC_PBWD(2) LC_PBWD(3) LC_PBWD(4) LC_PBWD(5) LC_PBWD(6) LC_PBWD(7) LC_PBWD(8) }
#undef LC_PBWD
        }
        if (!ok) {
            lc_set_error("%s: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed for the persistent kernel", who);
            (void)hipGetLastError();
            return LC_ELAUNCH;
        }
        PVerifyArgs va;
        va.ctl = pa.ctl; va.nused = ndir * pa.g.gpd; va.nwg = pa.g.nwg; va.nout = ndir;
        va.out[0] = dirs[0].gates; va.out[1] = dirs[ndir - 1].gates; va.count = (size_t)T * B * 4 * N;
        va.count16 = shadow3 ? 3 * va.count : va.count;      // split-operand kernel: the x3 shadow of dz its producers wrote
        va.out16[0] = (bf || shadow3) ? (unsigned short *)dirs[0].dz_bf16 : nullptr;
        va.out16[1] = (bf || shadow3) ? (unsigned short *)dirs[ndir - 1].dz_bf16 : nullptr;
        hipLaunchKernelGGL(persist_verify_kernel, dim3(256), dim3(256), 0, s, va);
        LC_CHECK_LAUNCH("lstm_bwd_persist");
        g_last_sched = (bf ? 2 : px3 ? 7 : 1) | ((int)bf << 16) | (1 << 17) | ((shadow3 && dirs[0].dz_bf16 ? 1 : 0) << 18);
        for (int i = 0; i < ndir; ++i)           // the kernel left per-row partials ([B][7][N]): only the fold remains
            if (pa.upg[i])
                hipLaunchKernelGGL(unit_param_fold_kernel, dim3(lc_cdiv(N, 256)), dim3(256), 0, s, pa.upg[i], B, N,
                                   (dirs[i].dpeep && dirs[i].w_f) ? dirs[i].dpeep : nullptr, dirs[i].dbias);
        LC_CHECK_LAUNCH("unit_param_fold");
        return LC_OK;
    } else {
        BwdArgs a;
        a.seq_len = seq_len; a.T = T; a.B = B; a.N = N; a.Bpad = bpad(B); a.row_base = 0;
        char *w = (char *)workspace + P_CTL_BYTES;
        for (int i = 0; i < ndir; ++i) {
            a.d[i].gates = dirs[i].gates;
            a.d[i].w_f = dirs[i].w_f; a.d[i].w_i = dirs[i].w_i; a.d[i].w_o = dirs[i].w_o;
            a.d[i].cs = dirs[i].cs; a.d[i].dh = dirs[i].dh; a.d[i].reverse = dirs[i].reverse;
            const size_t zbytes = al256((size_t)2 * 4 * N * a.Bpad * sizeof(float)) + al256((size_t)B * N * sizeof(float));
            if (hipMemsetAsync(w, 0, zbytes, s) != hipSuccess) {
                lc_set_error("%s: memset failed", who);
                return LC_ELAUNCH;
            }
            a.d[i].dzT = (float *)w; w += al256((size_t)2 * 4 * N * a.Bpad * sizeof(float));
            a.d[i].dc = (float *)w; w += al256((size_t)B * N * sizeof(float));
            pack_operand(bf, dirs[i].RT, 4 * N, N, w, s);
            a.d[i].RT = (const float *)w;
            w += al256((size_t)N * 4 * N * sizeof(float));
        }
        if (ndir == 1) a.d[1] = a.d[0];
        LC_CHECK_LAUNCH("pack_operand");
        // 32-row tiles: (N/16) x (B/32) x ndir workgroups of [32 x 16] outputs - 256 of them at N=1024, B=64
        // ... and 16-row tiles when that grid would leave most of the 256 CUs idle (N = 320 / 512 at B = 32: 40 / 64
        // workgroups -> 80 / 128; measured 7.4 -> 5.9 and 9.2 -> 7.0 us per step)
        int mt = a.Bpad >= 32 ? 2 : 1;
        if ((long long)(N / 16) * lc_cdiv(B, 32) * ndir < 200) mt = 1;
        dim3 grid(N / 16, lc_cdiv(B, 16 * mt), ndir), block(NTHREADS);
        // (the two-stream schedule of the forward pass does not pay here: 81-92 vs 80 ms per c4 step - the BPTT step
        // moves twice the operand bytes through L2 and gains nothing from interleaving)
        for (int step = 0; step < T; ++step) {
            a.step = step;
            if (bf) {
                if (mt == 1) hipLaunchKernelGGL((lstm_bwd_step_kernel<1, true>), grid, block, 0, s, a);
                else hipLaunchKernelGGL((lstm_bwd_step_kernel<2, true>), grid, block, 0, s, a);
            } else {
                if (mt == 1) hipLaunchKernelGGL((lstm_bwd_step_kernel<1, false>), grid, block, 0, s, a);
                else hipLaunchKernelGGL((lstm_bwd_step_kernel<2, false>), grid, block, 0, s, a);
            }
        }
        LC_CHECK_LAUNCH("lstm_bwd_step");
        g_last_sched = 4 | (mt << 8) | ((int)bf << 16) | (1 << 17);
    }
    // bias and peephole gradients (batched, f32, one pass over dz, deterministic two-stage reduce)
    for (int i = 0; i < ndir; ++i) {
        float *dpeep = (dirs[i].dpeep && dirs[i].w_f) ? dirs[i].dpeep : nullptr;
        if (dpeep || dirs[i].dbias) {
            dim3 g2(lc_cdiv(N, 64), UPG_SPLITS);
            hipLaunchKernelGGL(unit_param_grad_kernel, g2, dim3(256), 0, s, dirs[i].gates, dirs[i].cs, T, B, N,
                               dirs[i].reverse, dpeep ? 1 : 0, upg_part);
            hipLaunchKernelGGL(unit_param_fold_kernel, dim3(lc_cdiv(N, 256)), dim3(256), 0, s, upg_part, UPG_SPLITS, N,
                               dpeep, dirs[i].dbias);
        }
    }
    LC_CHECK_LAUNCH("unit_param_grad");
    return LC_OK;
}

extern "C" int lc_lstm_fwd(const lc_lstm_fwd_dir_t *dirs, int ndir, const int *seq_len, int T, int B, int N,
                           float forget_bias, void *workspace, size_t workspace_bytes, lc_stream_t stream)
{
    return lstm_fwd_impl(false, false, "lc_lstm_fwd", dirs, ndir, seq_len, T, B, N, forget_bias, workspace, workspace_bytes, stream);
}
extern "C" int lc_lstm_fwd_x3(const lc_lstm_fwd_dir_t *dirs, int ndir, const int *seq_len, int T, int B, int N,
                              float forget_bias, void *workspace, size_t workspace_bytes, lc_stream_t stream)
{
    return lstm_fwd_impl(false, true, "lc_lstm_fwd_x3", dirs, ndir, seq_len, T, B, N, forget_bias, workspace, workspace_bytes, stream);
}
extern "C" int lc_lstm_fwd_bf16(const lc_lstm_fwd_dir_t *dirs, int ndir, const int *seq_len, int T, int B, int N,
                                float forget_bias, void *workspace, size_t workspace_bytes, lc_stream_t stream)
{
    const int rc = lstm_fwd_impl(true, false, "lc_lstm_fwd_bf16", dirs, ndir, seq_len, T, B, N, forget_bias, workspace, workspace_bytes, stream);
    if (rc != LC_OK || T <= 0 || B <= 0) return rc;
    // hs_bf16: the persistent kernel writes it next to hs; every other schedule gets a cast behind the recurrence
    if ((lc_debug_last_lstm_schedule() & 0xff) != 2)
        for (int i = 0; i < ndir; ++i)
            if (dirs[i].hs_bf16) {
                const int rc2 = lc_cast_bf16(dirs[i].hs, T * B, N, N, dirs[i].hs_bf16, N, nullptr, 0, stream);
                if (rc2 != LC_OK) return rc2;
            }
    return LC_OK;
}
extern "C" int lc_lstm_bwd(const lc_lstm_bwd_dir_t *dirs, int ndir, const int *seq_len, int T, int B, int N,
                           void *workspace, size_t workspace_bytes, lc_stream_t stream)
{
    return lstm_bwd_impl(false, false, "lc_lstm_bwd", dirs, ndir, seq_len, T, B, N, workspace, workspace_bytes, stream);
}
extern "C" int lc_lstm_bwd_x3(const lc_lstm_bwd_dir_t *dirs, int ndir, const int *seq_len, int T, int B, int N,
                              void *workspace, size_t workspace_bytes, lc_stream_t stream)
{
    const int rc = lstm_bwd_impl(false, true, "lc_lstm_bwd_x3", dirs, ndir, seq_len, T, B, N, workspace, workspace_bytes, stream);
    if (rc != LC_OK || T <= 0 || B <= 0) return rc;
    // dz_bf16 here = the x3 shadow of dz ([T * B, 12 N] bf16, lc_split_bf16x3 layout): written by the split-operand pair
    // kernel's producers (bit 18 of the schedule word); every other schedule gets the split pass behind the recurrence
    if (!((lc_debug_last_lstm_schedule() >> 18) & 1))
        for (int i = 0; i < ndir; ++i)
            if (dirs[i].dz_bf16) {
                const int rc2 = lc_split_bf16x3(dirs[i].gates, T * B, 4 * N, 4 * N, dirs[i].dz_bf16, 12 * N, stream);
                if (rc2 != LC_OK) return rc2;
            }
    return LC_OK;
}
extern "C" int lc_lstm_bwd_bf16(const lc_lstm_bwd_dir_t *dirs, int ndir, const int *seq_len, int T, int B, int N,
                                void *workspace, size_t workspace_bytes, lc_stream_t stream)
{
    const int rc = lstm_bwd_impl(true, false, "lc_lstm_bwd_bf16", dirs, ndir, seq_len, T, B, N, workspace, workspace_bytes, stream);
    if (rc != LC_OK || T <= 0 || B <= 0) return rc;
    if ((lc_debug_last_lstm_schedule() & 0xff) != 2)        // see lc_lstm_fwd_bf16
        for (int i = 0; i < ndir; ++i)
            if (dirs[i].dz_bf16) {
                const int rc2 = lc_cast_bf16(dirs[i].gates, T * B, 4 * N, 4 * N, dirs[i].dz_bf16, 4 * N, nullptr, 0, stream);
                if (rc2 != LC_OK) return rc2;
            }
    return LC_OK;
}
