// ctc.hip — CTC loss / gradient / greedy decode for gfx950 (MI355X).
//
// Replaces tf.nn.ctc_loss, tf.nn.ctc_greedy_decoder (CPU-only kernels in TF 1.8) as called at
// mobvoi/lstm_ctc nnet/graph.py:109-114 and :138-142.  Semantics: SURVEY.md Appendix A.3/A.4.
//
// Three launches per loss call, all HBM/latency bound, no MFMA:
//   1. ctc_row_stats   one 16/64-lane group per (t,b) frame: max, log-sum-exp, first argmax (one memory round trip).
//   2. ctc_scan        one workgroup per DIRECTION of an utterance (alpha: t ascending, beta: t descending; the two
//                      run on different CUs).  The 2L+1 lattice lives in registers, cut into up to 4 segments of
//                      64 x PPL positions, one wave each, pipelined: a wave runs 16 steps behind the neighbour its
//                      inputs come from, which publishes the two cut-adjacent positions of every step in LDS (one
//                      lgkmcnt-only barrier per 16 steps).  Inside a wave the u-1/u-2 (u+1/u+2) neighbours move
//                      with one DPP wave shift each.  Log2 domain (v_exp_f32 / v_log_f32 are base 2) on RAW logits
//                      (sum_t lse_t is added to the loss in double), rows re-centred every 8 steps with a
//                      per-segment offset kept in double.  Logit gathers are prefetched 8 steps ahead.
//   3. ctc_grad        one wave per (t,b) frame: posterior mass per class from alpha+beta-logp (per-segment
//                      offsets), label positions through wave-private LDS float atomics, blank positions through a
//                      wave reduction; grad = softmax - posterior.  All independent loads are issued up front.
#include "common.h"

#define LC_NEG (-1.0e30f)
#define LC_LOG2E 1.4426950408889634f
#define LC_LN2 0.6931471805599453

// ------------------------------------------------------------------------------ row stats
// G lanes per (t, b) frame.  The frame's logits are requested before seq_len[b] is known (their address does not
// depend on it), so a group's life is one global-memory round trip, not two.
template <int G>
__global__ void ctc_row_stats_kernel(const float *__restrict__ logits, int T, int B, int V,
                                     const int *__restrict__ seq_len, float *__restrict__ rmax,
                                     float *__restrict__ rlse, int *__restrict__ argmax)
{
    const int rows = T * B;
    const int gid = (int)((blockIdx.x * (size_t)blockDim.x + threadIdx.x) / G);
    const int l = threadIdx.x % G;
    if (gid >= rows) return;
    const int b = gid % B, t = gid / B;
    const float *x = logits + (size_t)gid * V;
    const float x0 = l < V ? x[l] : -INFINITY;                  // V <= G: the whole frame
    const int len = seq_len[b];
    if (t >= len) {
        if (l == 0) {
            if (rmax) { rmax[gid] = 0.f; rlse[gid] = 0.f; }
            if (argmax) argmax[gid] = -1;
        }
        return;
    }
    float m = x0;
    int am = l < V ? l : 0x7fffffff;
    for (int k = l + G; k < V; k += G) {
        float v = x[k];
        if (v > m) { m = v; am = k; }
    }
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) {
        float m2 = __shfl_xor(m, o, G);
        int a2 = __shfl_xor(am, o, G);
        if (m2 > m || (m2 == m && a2 < am)) { m = m2; am = a2; }
    }
    if (argmax && l == 0) argmax[gid] = am;
    if (rmax) {
        float s = l < V ? expf(x0 - m) : 0.f;
        for (int k = l + G; k < V; k += G) s += expf(x[k] - m);
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, G);
        if (l == 0) { rmax[gid] = m; rlse[gid] = m + logf(s); }
    }
}

// ------------------------------------------------------------------------------ scan
// log2-domain log-sum-exp on the raw transcendental pipes (v_exp_f32 = 2^x, v_log_f32 = log2 x);
// arguments of exp2 are <= 0, "log zero" is the finite sentinel LC_NEG (absorbs every finite addend).
__device__ __forceinline__ float lse2_2(float a, float b)
{
    // max + log2(1 + 2^-|a-b|): one exp instead of two
    return fmaxf(a, b) + __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(-fabsf(a - b)));
}
__device__ __forceinline__ float lse3_2(float a, float b, float c)
{
    const float m = fmaxf(fmaxf(a, b), c);
    return m + __builtin_amdgcn_logf(__builtin_amdgcn_exp2f(a - m) + __builtin_amdgcn_exp2f(b - m) +
                                     __builtin_amdgcn_exp2f(c - m));
}

// Buffer addressing for the scan: a wave-uniform row pointer (SGPR pair, advanced with two scalar adds per step)
// plus a per-lane byte offset that never changes.  The flat-pointer form cost four 64-bit VALU adds and a dozen
// SALU multiplies per step, on a wave whose step time IS its instruction count (one wave per SIMD: ~5 cycles
// per instruction).
typedef int ctc_i32x4 __attribute__((ext_vector_type(4)));
typedef float ctc_f32x4 __attribute__((ext_vector_type(4)));
typedef float ctc_f32x2 __attribute__((ext_vector_type(2)));
__device__ float lc_ctc_buffer_load_f32(ctc_i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.f32");
__device__ void lc_ctc_buffer_store_f32(float v, ctc_i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.f32");
__device__ void lc_ctc_buffer_store_f32x2(ctc_f32x2 v, ctc_i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.v2f32");
__device__ void lc_ctc_buffer_store_f32x4(ctc_f32x4 v, ctc_i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.v4f32");

__device__ __forceinline__ ctc_i32x4 ctc_rsrc(const void *uniform_ptr)
{
    const unsigned long long b = (unsigned long long)uniform_ptr;
    const int lo = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    const int hi = __builtin_amdgcn_readfirstlane((int)((b >> 32) & 0xffffu));
    const ctc_i32x4 r = {lo, hi, -1, 0x00020000};
    return r;
}

// every lane stores its PPL positions: rows are 64*PPL floats wide, so no bounds check (and no exec-mask branch)
template <int PPL>
__device__ __forceinline__ void store_row(float *uniform_dst, const float (&v)[PPL], int lane)
{
    const ctc_i32x4 rs = ctc_rsrc(uniform_dst);
    if constexpr (PPL == 1) {
        lc_ctc_buffer_store_f32(v[0], rs, lane * 4, 0, 0);
    } else if constexpr (PPL == 2) {
        const ctc_f32x2 x = {v[0], v[1]};
        lc_ctc_buffer_store_f32x2(x, rs, lane * 8, 0, 0);
    } else {
#pragma unroll
        for (int c = 0; c < PPL / 4; ++c) {
            const ctc_f32x4 x = {v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]};
            lc_ctc_buffer_store_f32x4(x, rs, (lane * PPL + 4 * c) * 4, 0, 0);
        }
    }
}

constexpr int CTC_RING = 8;     // logit gathers are issued this many time steps ahead
constexpr int CTC_NORM = 8;     // the lattice row is re-centred (row max -> ~0) every CTC_NORM steps

// One alpha (DIR=0, t ascending) or beta (DIR=1, t descending) recursion over one utterance, one wave.
//   alpha_t[u] = e_t[u] + LSE(alpha_{t-1}[u], alpha_{t-1}[u-1], skip[u] ? alpha_{t-1}[u-2])
//   beta_t[u]  = LSE(g[u], g[u+1], skip[u] ? g[u+2]),  g = beta_{t+1} + e_{t+1}        (TF's beta: no emission at t)
// with e_t[u] = x[t, l'_u] * log2(e): RAW logits - the softmax normaliser sum_t lse_t is a per-utterance
// constant added to the loss afterwards.  Rows are stored re-centred: true log2 value = stored +
// coff[row group]; the offsets are kept in double per group of CTC_NORM rows.  The loop body is
// straight-line code (vector loads only, no per-step scalars), so the compiler's counted vmcnt keeps
// the gathers CTC_RING steps in flight.
__device__ __forceinline__ float ld_off(const ctc_i32x4 &row_rsrc, unsigned byte_off)
{
    return lc_ctc_buffer_load_f32(row_rsrc, (int)byte_off, 0, 0);
}

// Several waves per recursion.  The lattice of one direction is cut into NW contiguous segments of 64*PPL
// positions, one wave each.  Dependencies run one way only (alpha: towards higher positions, beta: towards lower), so
// the waves form a pipeline: the wave downstream of a cut runs one chunk of CTC_RING steps BEHIND its upstream
// neighbour, which publishes the two positions next to the cut for every step of a chunk in LDS; one workgroup
// barrier per chunk separates "published" from "consumed" (a per-step hand-shake was tried first and cost more
// than the split saved).  Every wave keeps its own re-centring offset; a published value is converted into the
// consumer's frame with the difference of the two offsets (both constant within a chunk).
constexpr int CTC_PIPE = 2;         // ring passes (of CTC_RING steps) per pipeline iteration, i.e. per barrier
struct CtcHand {                    // one cut: two iteration buffers
    float2 v[2][CTC_PIPE * CTC_RING][64];   // per step and LANE of the publisher (no exec-mask games: every lane writes)
    double coff[2][CTC_PIPE];               // the publisher's offset during each ring pass (constant within one)
};

template <int PPL, int DIR, bool GUARD, bool HASIN, bool HASOUT>
__device__ __forceinline__ void ctc_ring_pass(int s0, const float *__restrict__ xb, size_t rowstride, int Tb, int lane,
                                              const unsigned (&cls)[PPL], const bool (&valid)[PPL],
                                              const bool (&skip)[PPL], float *__restrict__ rows_out, int srow,
                                              double *__restrict__ coff_out, int coff_stride,
                                              float (&px)[CTC_RING][PPL], float (&a)[PPL], double &coff, float &mpend,
                                              const CtcHand *hin, CtcHand *hout, int cbuf, int q)
{
    const int hs = q * CTC_RING;      // first hand-off slot of this ring pass inside the iteration buffer
    // running row pointers (wave-uniform): gather row of step s + CTC_RING, lattice row of step s
    const long long gstep = DIR == 0 ? (long long)rowstride : -(long long)rowstride;
    const long long sstep = DIR == 0 ? (long long)srow : -(long long)srow;
    const float *grow = xb + (size_t)(DIR == 0 ? s0 + CTC_RING : Tb - 1 - s0 - CTC_RING) * rowstride;   // unguarded only
    float *srowp = rows_out + (size_t)(DIR == 0 ? s0 : Tb - 1 - s0) * srow;
    // boundary values of this chunk's steps, in this wave's frame (b1: the position next to the cut, b2: the next one)
    float b1[CTC_RING], b2[CTC_RING];
    if constexpr (HASIN) {
        const float delta = (float)(hin->coff[cbuf][q] - coff);
#pragma unroll
        for (int r = 0; r < CTC_RING; ++r) {
            if constexpr (PPL == 1) {      // the two positions sit in two lanes of the publisher
                b1[r] = hin->v[cbuf][hs + r][DIR == 0 ? 63 : 0].x + delta;
                b2[r] = hin->v[cbuf][hs + r][DIR == 0 ? 62 : 1].x + delta;
            } else {                        // .y = nearest to the cut, .x = the one behind it
                const float2 t2 = hin->v[cbuf][hs + r][DIR == 0 ? 63 : 0];
                b1[r] = t2.y + delta;
                b2[r] = t2.x + delta;
            }
            // "log zero" stays the sentinel whatever the frames are
            b1[r] = fmaxf(b1[r], LC_NEG);
            b2[r] = fmaxf(b2[r], LC_NEG);
        }
    }
    if (HASOUT && lane == 0) hout->coff[cbuf][q] = coff;
#pragma unroll
    for (int r = 0; r < CTC_RING; ++r) {
        const int s = s0 + r;
        if (!GUARD || s < Tb) {
            float e[PPL];
#pragma unroll
            for (int j = 0; j < PPL; ++j) e[j] = px[r][j] * LC_LOG2E;
            {   // refill this ring slot with the row CTC_RING steps ahead (guarded pass: clamped, a redundant load)
                const float *rowp = grow;
                if (GUARD) {
                    const int sn = min(s + CTC_RING, Tb - 1);
                    rowp = xb + (size_t)(DIR == 0 ? sn : Tb - 1 - sn) * rowstride;
                }
                const ctc_i32x4 rs = ctc_rsrc(rowp);
#pragma unroll
                for (int j = 0; j < PPL; ++j) px[r][j] = ld_off(rs, cls[j]);
                grow += gstep;
            }
            if ((r % CTC_NORM) == 0 && lane == 0) coff_out[(size_t)(s / CTC_NORM) * coff_stride] = coff;   // this row group
            if (DIR == 0) {
                if constexpr (HASOUT) {   // the row the downstream wave's step s reads: alpha_{s-1} next to the cut
                    if constexpr (PPL == 1) hout->v[cbuf][hs + r][lane] = make_float2(a[0], 0.f);
                    else hout->v[cbuf][hs + r][lane] = make_float2(a[PPL - 2], a[PPL - 1]);
                }
                const float f1 = HASIN ? b1[r] : LC_NEG, f2 = HASIN ? b2[r] : LC_NEG;
                float p1, p2;
                if constexpr (PPL == 1) {
                    p1 = lc_wave_shr1(a[0], f1);
                    p2 = lc_wave_shr1(p1, f2);
                } else {
                    p1 = lc_wave_shr1(a[PPL - 1], f1);
                    p2 = lc_wave_shr1(a[PPL - 2], f2);
                }
                float n[PPL];
#pragma unroll
                for (int j = 0; j < PPL; ++j) {
                    const float s1 = (j >= 1) ? a[j >= 1 ? j - 1 : 0] : p1;
                    const float s2 = (j >= 2) ? a[j >= 2 ? j - 2 : 0] : (j == 1 ? p1 : p2);
                    if ((PPL % 2 == 0) && (j % 2 == 0)) n[j] = lse2_2(a[j], s1) + e[j];   // blanks never skip
                    else n[j] = lse3_2(a[j], s1, skip[j] ? s2 : LC_NEG) + e[j];
                }
#pragma unroll
                for (int j = 0; j < PPL; ++j) a[j] = n[j];
                store_row<PPL>(srowp, a, lane);
            } else {
                store_row<PPL>(srowp, a, lane);
                float g[PPL];
#pragma unroll
                for (int j = 0; j < PPL; ++j) g[j] = a[j] + e[j];
                if constexpr (HASOUT) {   // g of this step next to the cut: .y = nearest (position 0 of this wave)
                    if constexpr (PPL == 1) hout->v[cbuf][hs + r][lane] = make_float2(g[0], 0.f);
                    else hout->v[cbuf][hs + r][lane] = make_float2(g[1], g[0]);
                }
                const float f1 = HASIN ? b1[r] : LC_NEG, f2 = HASIN ? b2[r] : LC_NEG;
                float n1, n2;
                if constexpr (PPL == 1) {
                    n1 = lc_wave_shl1(g[0], f1);
                    n2 = lc_wave_shl1(n1, f2);
                } else {
                    n1 = lc_wave_shl1(g[0], f1);
                    n2 = lc_wave_shl1(g[1], f2);
                }
                float n[PPL];
#pragma unroll
                for (int j = 0; j < PPL; ++j) {
                    const float s1 = (j + 1 < PPL) ? g[j + 1 < PPL ? j + 1 : 0] : n1;
                    const float s2 = (j + 2 < PPL) ? g[j + 2 < PPL ? j + 2 : 0] : (j + 2 == PPL ? n1 : n2);
                    if ((PPL % 2 == 0) && (j % 2 == 0)) n[j] = lse2_2(g[j], s1);
                    else n[j] = lse3_2(g[j], s1, skip[j] ? s2 : LC_NEG);
                }
#pragma unroll
                for (int j = 0; j < PPL; ++j) a[j] = n[j];
            }
            srowp += sstep;
            // Re-centring, kept off the dependent chain: the row max is taken one step before the group
            // boundary (its wave reduction overlaps the next step) and subtracted at the boundary.
            if ((r % CTC_NORM) == CTC_NORM - 2) {
                float m = LC_NEG;
#pragma unroll
                for (int j = 0; j < PPL; ++j) m = fmaxf(m, valid[j] ? a[j] : LC_NEG);
                m = lc_wave_max(m);
                mpend = (m < -1.0e29f) ? 0.f : m;
            }
            if ((r % CTC_NORM) == CTC_NORM - 1) {
#pragma unroll
                for (int j = 0; j < PPL; ++j) a[j] = fmaxf(a[j] - mpend, LC_NEG);
                coff += (double)mpend;
            }
        }
    }
}

// Chunk barrier of the pipeline: only the LDS hand-off has to be ordered, so wait for lgkmcnt alone - a
// __syncthreads() would also drain vmcnt, i.e. the logit gathers that are deliberately in flight CTC_RING steps ahead.
template <int NW>
__device__ __forceinline__ void ctc_chunk_barrier()
{
    if constexpr (NW > 1) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
}

// One wave's whole recursion.  A pipeline iteration is CTC_PIPE ring passes followed by one barrier: `lag` idle
// iterations, the iterations made of unguarded passes only (one straight-line loop body), the tail iterations
// (guarded passes), then idle iterations until every wave of the workgroup has done `total` barriers.
template <int PPL, int NW, int DIR, bool HASIN, bool HASOUT>
__device__ __forceinline__ void ctc_wave_loop(int lag, int total, int npass, const float *__restrict__ xb,
                                              size_t rowstride, int Tb, int lane, const unsigned (&cls)[PPL],
                                              const bool (&valid)[PPL], const bool (&skip)[PPL],
                                              float *__restrict__ rows_out, int srow, double *__restrict__ coff_out,
                                              float (&px)[CTC_RING][PPL], float (&a)[PPL], double &coff,
                                              const CtcHand *hin, CtcHand *hout)
{
    float mpend = 0.f;
    const int nun = max(0, Tb / CTC_RING - 1);        // ring passes whose every prefetch is in range
    const int nit = (npass + CTC_PIPE - 1) / CTC_PIPE;
    for (int i = 0; i < lag; ++i) ctc_chunk_barrier<NW>();
    int it = 0;
    for (; (it + 1) * CTC_PIPE <= nun; ++it) {
#pragma unroll
        for (int q = 0; q < CTC_PIPE; ++q)
            ctc_ring_pass<PPL, DIR, false, HASIN, HASOUT>((it * CTC_PIPE + q) * CTC_RING, xb, rowstride, Tb, lane, cls,
                                                          valid, skip, rows_out, srow, coff_out, NW, px, a, coff, mpend,
                                                          hin, hout, it & 1, q);
        ctc_chunk_barrier<NW>();
    }
    for (; it < nit; ++it) {
        for (int q = 0; q < CTC_PIPE; ++q) {
            const int cc = it * CTC_PIPE + q;
            if (cc < npass)
                ctc_ring_pass<PPL, DIR, true, HASIN, HASOUT>(cc * CTC_RING, xb, rowstride, Tb, lane, cls, valid, skip,
                                                             rows_out, srow, coff_out, NW, px, a, coff, mpend, hin, hout,
                                                             it & 1, q);
        }
        ctc_chunk_barrier<NW>();
    }
    for (int i = lag + nit; i < total; ++i) ctc_chunk_barrier<NW>();
}

// Workgroup = one direction of one utterance (blockIdx.x = 2*b + direction): NW scan waves + one wave for
// sum_t lse_t (idle in the beta workgroup).  Alpha and beta of an utterance sit in different workgroups, hence
// on different CUs: with both in one workgroup the 2*NW scan waves shared 4 SIMDs and halved each other's issue rate.
template <int PPL, int NW>
__global__ __launch_bounds__((NW + 1) * 64) void ctc_scan_kernel(
    const float *__restrict__ logits, int T, int B, int V, const int *__restrict__ labels,
    const int *__restrict__ offs, const int *__restrict__ seq_len, const float *__restrict__ rlse,
    float *__restrict__ alpha, float *__restrict__ beta, int srow, double *__restrict__ coffa,
    double *__restrict__ coffb, int ngroups, float *__restrict__ loss, double *__restrict__ logp2_out,
    int *__restrict__ status)
{
    __shared__ double lse_sum;
    __shared__ double fin[2];                                   // alpha[U-1], alpha[U-2] of the last row, true log2
    __shared__ CtcHand hand[NW > 1 ? NW - 1 : 1];              // one per cut between neighbouring segments
    const int b = blockIdx.x >> 1;
    const int dir = blockIdx.x & 1;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int off0 = offs[b];
    const int L = offs[b + 1] - off0;
    const int Tb = min(seq_len[b], T);
    const int U = 2 * L + 1;
    if (L > Tb || Tb <= 0) {   // ignore_longer_outputs_than_inputs=True: utterance skipped
        if (threadIdx.x == 0 && dir == 0) { loss[b] = 0.f; logp2_out[b] = 0.0; status[b] = 1; }
        return;
    }
    {   // tf.nn.ctc_loss: InvalidArgument for a label outside [0, V-1).  The host wrapper validates; at the C ABI such
        // an utterance gets a NaN loss and a zero gradient, and its labels never index anything.
        int bad = 0;
        for (int i = threadIdx.x; i < L; i += blockDim.x) {
            const int lab = labels[off0 + i];
            bad |= (lab < 0 || lab >= V - 1);
        }
        if (__syncthreads_or(bad)) {
            if (threadIdx.x == 0 && dir == 0) { loss[b] = __builtin_nanf(""); logp2_out[b] = 0.0; status[b] = 1; }
            return;
        }
    }
    const bool is_alpha = dir == 0 && wave < NW, is_beta = dir == 1 && wave < NW;
    const int seg = wave < NW ? wave : 0;                      // which segment of the lattice this wave owns
    const int ubase = seg * 64 * PPL;
    const int blank = V - 1;
    unsigned cls[PPL];   // byte offset of the lattice position's class within a logits row
    bool valid[PPL], skip[PPL];
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
        const int u = ubase + lane * PPL + j;
        valid[j] = u < U && (is_alpha || is_beta);
        const bool odd = (u & 1) && valid[j];
        const int lab = odd ? labels[off0 + (u >> 1)] : blank;
        cls[j] = (unsigned)lab * 4u;
        // Keep the gather index in a VGPR the compiler cannot prove uniform: a uniform (blank) address would
        // become an s_load, and scalar loads retire out of order - every lgkmcnt(0) would then also wait for
        // the prefetch issued a moment ago (measured: 2.4x slower scan).
        asm volatile("" : "+v"(cls[j]));
        if (is_alpha)    // alpha: may u be entered from u-2 ?
            skip[j] = odd && u >= 3 && lab != labels[off0 + ((u - 3) >> 1)];
        else             // beta: may u move on to u+2 ?
            skip[j] = odd && (u + 2 < U) && lab != labels[off0 + ((u + 1) >> 1)];
    }
    const size_t rowstride = (size_t)B * V;
    const float *xb = logits + (size_t)b * V;
    const int nchunks = (Tb + CTC_RING - 1) / CTC_RING;
    float a[PPL];
    double coff = 0.0;
    float px[CTC_RING][PPL];
    if (is_alpha || is_beta) {
        // virtual row before the first step.  alpha: all mass on u = 0, so the generic step yields
        // alpha_0 = (e[0], e[1], -inf, ...); beta: the two final positions
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
            const int u = ubase + lane * PPL + j;
            if (is_alpha) a[j] = (u == 0) ? 0.f : LC_NEG;
            else a[j] = (u < U && u >= U - 2) ? 0.f : LC_NEG;
        }
#pragma unroll
        for (int r = 0; r < CTC_RING; ++r) {
            const int sn = min(r, Tb - 1);
            const int t = is_alpha ? sn : Tb - 1 - sn;
            const ctc_i32x4 rs = ctc_rsrc(xb + (size_t)t * rowstride);
#pragma unroll
            for (int j = 0; j < PPL; ++j) px[r][j] = ld_off(rs, cls[j]);
        }
    } else {
        // last wave: sum_t lse_t, the softmax normaliser the raw-logit recursions left out (alpha workgroup only)
        double acc = 0.0;
        if (dir == 0)
            for (int t = lane; t < Tb; t += 64) acc += (double)rlse[(size_t)t * B + b];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (lane == 0) { lse_sum = acc; fin[0] = -1.0e300; fin[1] = -1.0e300; }
    }
    float *arow = alpha + (size_t)b * T * srow + ubase;
    float *brow = beta + (size_t)b * T * srow + ubase;
    double *ca = coffa + (size_t)b * ngroups * NW + seg, *cb = coffb + (size_t)b * ngroups * NW + seg;
    // the pipeline: alpha flows towards higher segments (segment k lags k chunks), beta towards lower ones
    const int total = (nchunks + CTC_PIPE - 1) / CTC_PIPE + NW - 1;
#define LC_LOOP(DIR, HASIN, HASOUT, LAG, ROWS, COFF, HIN, HOUT)                                                      \
    ctc_wave_loop<PPL, NW, DIR, HASIN, HASOUT>(LAG, total, nchunks, xb, rowstride, Tb, lane, cls, valid, skip, ROWS, \
                                               srow, COFF, px, a, coff, HIN, HOUT)
    if (is_alpha) {
        const CtcHand *hin = &hand[seg > 0 ? seg - 1 : 0];                  // cut k lies between segments k and k+1
        CtcHand *hout = &hand[seg < NW - 1 ? seg : 0];
        if (NW == 1) LC_LOOP(0, false, false, 0, arow, ca, hin, hout);
        else if (seg == 0) LC_LOOP(0, false, true, 0, arow, ca, hin, hout);
        else if (seg == NW - 1) LC_LOOP(0, true, false, seg, arow, ca, hin, hout);
        else LC_LOOP(0, true, true, seg, arow, ca, hin, hout);
    } else if (is_beta) {
        const CtcHand *hin = &hand[seg < NW - 1 ? seg : 0];
        CtcHand *hout = &hand[seg > 0 ? seg - 1 : 0];
        if (NW == 1) LC_LOOP(1, false, false, 0, brow, cb, hin, hout);
        else if (seg == NW - 1) LC_LOOP(1, false, true, 0, brow, cb, hin, hout);
        else if (seg == 0) LC_LOOP(1, true, false, NW - 1, brow, cb, hin, hout);
        else LC_LOOP(1, true, true, NW - 1 - seg, brow, cb, hin, hout);
    } else {
        for (int i = 0; i < total; ++i) ctc_chunk_barrier<NW>();
    }
#undef LC_LOOP
    __syncthreads();
    if (is_alpha) {
        // true log2 values of alpha[U-1], alpha[U-2] in the last row, from whichever wave owns them
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
            const int u = ubase + lane * PPL + j;
            if (u == U - 1) fin[0] = (a[j] < -1.0e29f) ? -1.0e300 : (double)a[j] + coff;
            if (u == U - 2) fin[1] = (a[j] < -1.0e29f) ? -1.0e300 : (double)a[j] + coff;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && dir == 0) {
        const double v1 = fin[0], v2 = (U < 2) ? -1.0e300 : fin[1];
        const double m = v1 > v2 ? v1 : v2;
        if (m < -1.0e299) {   // no valid path (TF: loss = +inf, gradient = softmax)
            loss[b] = INFINITY; logp2_out[b] = 0.0; status[b] = 2;
        } else {
            const double lo = v1 > v2 ? v2 : v1;
            const float d = (lo < -1.0e299) ? -INFINITY : (float)(lo - m);
            const double lp = m + (double)__builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(d));
            loss[b] = (float)(lse_sum - lp * LC_LN2);
            logp2_out[b] = lp; status[b] = 0;
        }
    }
}

// ------------------------------------------------------------------------------ gradient
__global__ __launch_bounds__(256) void ctc_grad_kernel(
    const float *__restrict__ logits, int T, int B, int V, const int *__restrict__ labels,
    const int *__restrict__ offs, const int *__restrict__ seq_len, const float *__restrict__ rlse,
    const float *__restrict__ alpha, const float *__restrict__ beta, int srow,
    const double *__restrict__ coffa, const double *__restrict__ coffb, int ngroups, int nw, int seglen,
    const double *__restrict__ logp2, const int *__restrict__ status, float *__restrict__ grad)
{
    // One wave per (t, b) frame; a wave's life is a chain of global-memory round trips (~1 us each), so everything
    // whose address does not depend on loaded data is requested up front and the chain is two trips long:
    //   trip 1: status / length / label range / log p / lse / logits / alpha and beta rows / alpha's offsets
    //   trip 2: beta's offsets (indexed by Tb - 1 - t) and the labels of this lane's lattice positions
    // The per-class bins are private to the wave (LDS operations of one wave execute in order): no barriers.
    extern __shared__ __attribute__((aligned(16))) float bins_all[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long rows = (long long)T * B;
    const long long row = min((long long)blockIdx.x * 4 + wave, rows - 1);
    const bool inrange = (long long)blockIdx.x * 4 + wave < rows;
    const int b = (int)(row % B), t = (int)(row / B);
    const int st = status[b];
    const int Tb = min(seq_len[b], T);
    const int off0 = offs[b], off1 = offs[b + 1];
    const double lpd = logp2[b];
    const float lse = rlse[row];
    const float *x = logits + (size_t)row * V;
    const float x0 = lane < V ? x[lane] : 0.f;                               // class `lane` (V <= 64: the whole row)
    const size_t ro = ((size_t)b * T + t) * srow;
    float *bins = bins_all + wave * V;
    for (int k = lane; k < V; k += 64) bins[k] = 0.f;
    const double *ca = coffa + ((size_t)b * ngroups + t / CTC_NORM) * nw;
    double cav[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) cav[w] = ca[min(w, nw - 1)];
    const bool live = inrange && t < Tb && st != 1;
    const int U = 2 * (off1 - off0) + 1;
    float blank_acc = 0.f;
    if (live && st == 0) {
        // stored rows are re-centred per lattice segment (one scan wave each): log2(alpha*beta/p) = a + b +
        // (offset_a + offset_b - log2 p), the bracket summed in double per segment
        const double *cb = coffb + ((size_t)b * ngroups + (Tb - 1 - t) / CTC_NORM) * nw;
        float lps[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) lps[w] = (float)(lpd - cav[w] - cb[min(w, nw - 1)]);
        // four consecutive positions per lane and pass: rows are 64*PPL*NW floats wide and positions >= U hold the
        // "log zero" sentinel (exp2 -> 0), so whole float4s are read
        for (int u0 = 4 * lane; u0 < U; u0 += 256) {
            const float4 av = *reinterpret_cast<const float4 *>(alpha + ro + u0);
            const float4 bv = *reinterpret_cast<const float4 *>(beta + ro + u0);
            const int l1 = (u0 + 1 < U) ? labels[off0 + (u0 >> 1)] : 0;
            const int l3 = (u0 + 3 < U) ? labels[off0 + (u0 >> 1) + 1] : 0;
            const int w = u0 / seglen;                                       // seglen % 4 == 0: one segment per float4
            const float lp = w == 0 ? lps[0] : (w == 1 ? lps[1] : (w == 2 ? lps[2] : lps[3]));
            const float e0 = __builtin_amdgcn_exp2f(av.x + bv.x - lp), e1 = __builtin_amdgcn_exp2f(av.y + bv.y - lp);
            const float e2 = __builtin_amdgcn_exp2f(av.z + bv.z - lp), e3 = __builtin_amdgcn_exp2f(av.w + bv.w - lp);
            blank_acc += e0 + ((u0 + 2 < U) ? e2 : 0.f);
            if (u0 + 1 < U) atomicAdd(&bins[l1], e1);
            if (u0 + 3 < U) atomicAdd(&bins[l3], e3);
        }
    }
    blank_acc = lc_wave_sum(blank_acc);
    if (inrange) {
        float *g = grad + (size_t)row * V;
        if (!live) {
            for (int k = lane; k < V; k += 64) g[k] = 0.f;
        } else {
            for (int k = lane; k < V; k += 64) {
                const float xv = (k == lane) ? x0 : x[k];
                const float y = expf(xv - lse);
                const float p = (st == 2) ? 0.f : ((k == V - 1) ? blank_acc : bins[k]);
                g[k] = y - p;
            }
        }
    }
}

// ------------------------------------------------------------------------------ greedy collapse
__global__ __launch_bounds__(256) void ctc_collapse_kernel(const int *__restrict__ argmax, int T, int B,
                                                           int V, const int *__restrict__ seq_len,
                                                           int *__restrict__ tokens, int *__restrict__ out_len)
{
    __shared__ int wsum[4];
    __shared__ int base_s;
    const int b = blockIdx.x;
    const int Tb = min(seq_len[b], T);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int blank = V - 1;
    if (threadIdx.x == 0) base_s = 0;
    __syncthreads();
    for (int t0 = 0; t0 < Tb; t0 += 256) {
        const int t = t0 + threadIdx.x;
        int k = -1;
        bool emit = false;
        if (t < Tb) {
            k = argmax[(size_t)t * B + b];
            const int prev = (t > 0) ? argmax[(size_t)(t - 1) * B + b] : -1;
            emit = (k != blank) && (k != prev);
        }
        const unsigned long long m = __ballot(emit);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(m);
        __syncthreads();
        int wbase = base_s;
        for (int w = 0; w < wave; ++w) wbase += wsum[w];
        if (emit) tokens[(size_t)b * T + wbase + before] = k;
        __syncthreads();
        if (threadIdx.x == 0) base_s += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) out_len[b] = base_s;
}

// ------------------------------------------------------------------------------ C ABI
static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
// lattice positions per lane of the scan kernel chosen for S = 2L+1 positions, and the row pitch that goes with it
// (every lane stores its positions unconditionally, so a row is 64 * PPL floats)
// scan geometry for S = 2L+1 lattice positions: NW waves per direction, PPL positions per lane; a row is
// 64 * PPL * NW floats (every lane stores its positions unconditionally)
static inline int ctc_nw(int S) { return S <= 64 ? 1 : S <= 128 ? 2 : 4; }
static inline int ctc_ppl(int S) { return S <= 256 ? 1 : S <= 512 ? 2 : S <= 1024 ? 4 : 8; }
static inline int ctc_srow(int max_label_len)
{
    const int S = 2 * max_label_len + 1;
    return 64 * ctc_ppl(S) * ctc_nw(S);
}

extern "C" size_t lc_ctc_workspace_bytes(int T, int B, int V, int max_label_len)
{
    (void)V;
    const size_t rows = (size_t)T * B;
    const size_t lat = align256(rows * ctc_srow(max_label_len) * sizeof(float));
    const size_t ng = (size_t)(T + CTC_NORM - 1) / CTC_NORM;
    return 2 * align256(rows * sizeof(float)) + 2 * align256((size_t)B * sizeof(double)) +
           2 * align256((size_t)B * ng * 4 * sizeof(double)) + 2 * lat;
}

static void launch_row_stats(const float *logits, int T, int B, int V, const int *seq_len, float *rmax,
                             float *rlse, int *argmax, hipStream_t s)
{
    const size_t rows = (size_t)T * B;
    if (V <= 32) {
        const int G = 16;
        hipLaunchKernelGGL(ctc_row_stats_kernel<G>, dim3(lc_cdiv(rows * G, 256)), dim3(256), 0, s, logits, T, B, V,
                           seq_len, rmax, rlse, argmax);
    } else {
        const int G = 64;
        hipLaunchKernelGGL(ctc_row_stats_kernel<G>, dim3(lc_cdiv(rows * G, 256)), dim3(256), 0, s, logits, T, B, V,
                           seq_len, rmax, rlse, argmax);
    }
}

extern "C" int lc_ctc_loss(const float *logits, int T, int B, int V, const int *labels,
                           const int *label_offsets, const int *seq_len, int max_label_len, float *loss,
                           float *grad, void *workspace, size_t workspace_bytes, lc_stream_t stream)
{
    LC_CHECK_ARG(logits && labels && label_offsets && seq_len && loss && workspace, "lc_ctc_loss: null pointer");
    LC_CHECK_ARG(T > 0 && B > 0 && V >= 2 && max_label_len >= 0, "lc_ctc_loss: bad shape T=%d B=%d V=%d L=%d", T, B, V, max_label_len);
    const int S = 2 * max_label_len + 1;
    LC_CHECK_ARG(S <= 64 * 32, "lc_ctc_loss: label length %d exceeds the supported maximum 1023", max_label_len);
    if (workspace_bytes < lc_ctc_workspace_bytes(T, B, V, max_label_len)) {
        lc_set_error("lc_ctc_loss: workspace too small (%zu < %zu)", workspace_bytes,
                     lc_ctc_workspace_bytes(T, B, V, max_label_len));
        return LC_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    const size_t rows = (size_t)T * B;
    const int srow = ctc_srow(max_label_len);
    char *w = (char *)workspace;
    float *rmax = (float *)w; w += align256(rows * sizeof(float));
    float *rlse = (float *)w; w += align256(rows * sizeof(float));
    const int ngroups = (T + CTC_NORM - 1) / CTC_NORM;
    double *logp2 = (double *)w; w += align256((size_t)B * sizeof(double));
    int *status = (int *)w; w += align256((size_t)B * sizeof(double));
    double *coffa = (double *)w; w += align256((size_t)B * ngroups * 4 * sizeof(double));      // [B][groups][NW <= 4]
    double *coffb = (double *)w; w += align256((size_t)B * ngroups * 4 * sizeof(double));
    const size_t lat = align256(rows * srow * sizeof(float));
    float *alpha = (float *)w; w += lat;
    float *beta = (float *)w;

    launch_row_stats(logits, T, B, V, seq_len, rmax, rlse, nullptr, s);
    LC_CHECK_LAUNCH("ctc_row_stats");
    const int nw = ctc_nw(S), ppl = ctc_ppl(S);
#define LC_SCAN(PPL, NW)                                                                                      \
    hipLaunchKernelGGL((ctc_scan_kernel<PPL, NW>), dim3(2 * B), dim3((NW + 1) * 64), 0, s, logits, T, B, V, labels, \
                       label_offsets, seq_len, rlse, alpha, beta, srow, coffa, coffb, ngroups, loss, logp2, status)
    if (nw == 1) LC_SCAN(1, 1);
    else if (nw == 2) LC_SCAN(1, 2);
    else if (ppl == 1) LC_SCAN(1, 4);
    else if (ppl == 2) LC_SCAN(2, 4);
    else if (ppl == 4) LC_SCAN(4, 4);
    else LC_SCAN(8, 4);
#undef LC_SCAN
    LC_CHECK_LAUNCH("ctc_scan");
    if (grad) {
        hipLaunchKernelGGL(ctc_grad_kernel, dim3(lc_cdiv(rows, 4)), dim3(256), 4 * V * sizeof(float), s, logits, T,
                           B, V, labels, label_offsets, seq_len, rlse, alpha, beta, srow, coffa, coffb, ngroups, nw, 64 * ppl, logp2, status, grad);
        LC_CHECK_LAUNCH("ctc_grad");
    }
    return LC_OK;
}

extern "C" int lc_ctc_greedy(const float *logits, int T, int B, int V, const int *seq_len, int *tokens,
                             int *out_len, int *argmax_workspace, lc_stream_t stream)
{
    LC_CHECK_ARG(logits && seq_len && tokens && out_len && argmax_workspace, "lc_ctc_greedy: null pointer");
    LC_CHECK_ARG(T > 0 && B > 0 && V >= 2, "lc_ctc_greedy: bad shape");
    hipStream_t s = (hipStream_t)stream;
    launch_row_stats(logits, T, B, V, seq_len, nullptr, nullptr, argmax_workspace, s);
    LC_CHECK_LAUNCH("ctc_row_argmax");
    hipLaunchKernelGGL(ctc_collapse_kernel, dim3(B), dim3(256), 0, s, argmax_workspace, T, B, V, seq_len, tokens,
                       out_len);
    LC_CHECK_LAUNCH("ctc_collapse");
    return LC_OK;
}

extern "C" int lc_edit_distance_host(const int *hyp, int hyp_stride, const int *hyp_len, const int *truth,
                                     const int *truth_offsets, int B, int *dist)
{
    LC_CHECK_ARG(hyp && hyp_len && truth_offsets && dist && B >= 0, "lc_edit_distance_host: bad argument");
    for (int b = 0; b < B; ++b) {
        const int *h = hyp + (size_t)b * hyp_stride;
        const int *r = truth + truth_offsets[b];
        const int n = hyp_len[b], m = truth_offsets[b + 1] - truth_offsets[b];
        int *row = new int[m + 1];
        for (int j = 0; j <= m; ++j) row[j] = j;
        for (int i = 1; i <= n; ++i) {
            int diag = row[0];
            row[0] = i;
            for (int j = 1; j <= m; ++j) {
                const int sub = diag + (h[i - 1] != r[j - 1]);
                const int del = row[j] + 1, ins = row[j - 1] + 1;
                diag = row[j];
                const int v = sub < del ? sub : del;
                row[j] = v < ins ? v : ins;
            }
        }
        dist[b] = row[m];
        delete[] row;
    }
    return LC_OK;
}
