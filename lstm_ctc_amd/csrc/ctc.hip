// ctc.hip — CTC loss / gradient / greedy decode for gfx950 (MI355X).
//
// Replaces tf.nn.ctc_loss, tf.nn.ctc_greedy_decoder (CPU-only kernels in TF 1.8) as called at
// mobvoi/lstm_ctc nnet/graph.py:109-114 and :138-142.  Semantics: SURVEY.md Appendix A.3/A.4.
//
// Default path (V <= 128): two launches of ctc_mm_kernel ("meet in the middle", see the section of that name below).
// Wider alphabets take the round-1 path: three launches per loss call, all HBM/latency bound, no MFMA:
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
#include <stdlib.h>

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

// ====================================================================================== meet-in-the-middle CTC
// The default loss / gradient path (V <= 128; lattices of <= 256 positions at any batch size, of <= 1024 positions
// for B > 128 - mm_supported): TWO launches of the same scan, no separate row-stats or gradient
// pass, beta never stored for the frames alpha covers (and vice versa):
//
//   phase 1   the alpha workgroup of an utterance walks t = 0 .. M-1 (M = T_b / 2) and stores its rows; the beta
//             workgroup walks t = T_b-1 .. M and stores ITS rows - one lattice buffer [T][S], alpha rows below M, beta rows
//             from M on: T rows written once, S wide (stores are bounds-checked by the buffer descriptor).  Meanwhile the
//             fifth wave of each workgroup computes log-sum-exp(logits[t,:]) for the frames of its own phase 2.
//   phase 2   (kernel boundary = the hand-over; no cross-workgroup wait, no co-residency assumption): both recursions
//             resume from their carried row.  log p = LSE_u(alpha_{M-1}[u] + beta_{M-1}[u]) is available at once, so the
//             alpha workgroup, at t = M .. T_b-1, multiplies its fresh alpha_t with the STORED beta_t (prefetched like
//             the logit gathers) and the beta workgroup, at t = M-1 .. 0, its fresh beta_t with the stored alpha_t:
//             every lattice position's posterior is one plain, conflict-free ds_write into the frame's LDS row at the
//             position's rank in CLASS order (positions sorted by class once per launch), and the fifth wave, one
//             pipeline iteration behind the last scan wave, turns each finished frame into grad[t,:] = softmax -
//             posterior: an in-place wave prefix sum of the row (DPP), class mass = difference of two prefix values.
//             (LDS float atomics per class were measured first: a wave-wide ds_add_f32 retires in ~440 cycles, more
//             than a scan step, and throttled the chain - 273 vs 168 us at the c4 shape.  They remain only for lattices
//             of more than 256 positions, whose rows would not fit the LDS.)
//
// Chain length per utterance: T_b steps in all (as before: alpha and beta ran T_b steps each, concurrently), HBM
// traffic per (utterance, frame): logits read by both scans out of one L2 (the two workgroups of an utterance are
// given block indices that map to the same XCD) + once more by the gradient wave, S floats of lattice written and read
// once, V floats of gradient written: about 12V + 8S bytes against the algorithmic 8V + 8S.
constexpr int MM_IT = CTC_PIPE * CTC_RING;           // steps per pipeline iteration (= per barrier)

struct MmArgs {
    const float *logits;
    int T, B, V;
    const int *labels, *offs, *seq_len;
    float *lat;             // [B][T][srow]
    int srow;
    double *coff;           // [B][2][ngroups][4] row-group offsets of the stored rows (per direction, per segment)
    int ngroups;
    float *carry;           // [B][2][cw] the row each recursion resumes from
    int cw;
    double *carry_off;      // [B][2][4]
    float *rlse;            // [T*B] log-sum-exp of every frame (natural log)
    double *lsepart;        // [B][2] sum_t lse_t over the frames of each workgroup's phase 2
    float *loss, *grad;
    unsigned long long *dbg;    // development hook (lc_debug_set_ctc_stamps): s_memtime stamps of workgroup 0, or NULL
    int lse2;               // 1: no frame statistics in phase 1 - phase 2's frame waves take the log-sum-exp of a frame where they
                            // take its softmax (one logits read less), and the loss is folded by two float atomic adds
};
static unsigned long long *g_ctc_dbg = nullptr;
// stamp k of wave w at pipeline iteration `it` of phase PH: dbg[((PH-1)*5 + w) * 4096 + it * 8 + k]
#define LC_CSTAMP(ph, w, it, k)                                                                                  \
    do {                                                                                                         \
        if (p.dbg && blockIdx.x == 0 && lane == 0 && wave <= (w) && (it) < 512)                                  \
            p.dbg[(((ph) - 1) * 5 + (w)) * 4096 + (it) * 8 + (k)] = __builtin_amdgcn_s_memtime();               \
    } while (0)

__device__ __forceinline__ ctc_i32x4 ctc_rsrc_n(const void *uniform_ptr, unsigned bytes)
{
    const unsigned long long b = (unsigned long long)uniform_ptr;
    const int lo = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    const int hi = __builtin_amdgcn_readfirstlane((int)((b >> 32) & 0xffffu));
    const ctc_i32x4 r = {lo, hi, __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000};
    return r;
}
__device__ float lc_ctc_buffer_load_f32x(ctc_i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.f32");
__device__ ctc_f32x2 lc_ctc_buffer_load_f32x2(ctc_i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v2f32");
__device__ ctc_f32x4 lc_ctc_buffer_load_f32x4(ctc_i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");

// this lane's PPL positions of a lattice row; positions beyond `bytes` are dropped (stores) / read as 0 (loads)
template <int PPL>
__device__ __forceinline__ void mm_store_row(float *uniform_dst, unsigned bytes, const float (&v)[PPL], int lane)
{
    const ctc_i32x4 rs = ctc_rsrc_n(uniform_dst, bytes);
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
template <int PPL>
__device__ __forceinline__ void mm_load_row(const float *uniform_src, unsigned bytes, float (&v)[PPL], int lane)
{
    const ctc_i32x4 rs = ctc_rsrc_n(uniform_src, bytes);
    if constexpr (PPL == 1) {
        v[0] = lc_ctc_buffer_load_f32x(rs, lane * 4, 0, 0);
    } else if constexpr (PPL == 2) {
        const ctc_f32x2 x = lc_ctc_buffer_load_f32x2(rs, lane * 8, 0, 0);
        v[0] = x[0]; v[1] = x[1];
    } else {
#pragma unroll
        for (int c = 0; c < PPL / 4; ++c) {
            const ctc_f32x4 x = lc_ctc_buffer_load_f32x4(rs, (lane * PPL + 4 * c) * 4, 0, 0);
            v[4 * c] = x[0]; v[4 * c + 1] = x[1]; v[4 * c + 2] = x[2]; v[4 * c + 3] = x[3];
        }
    }
}

struct MmPh2 {                  // per ring pass, phase 2 only
    float lpA, lpB;             // log2 p - own offset - partner offset, for the partner's two row groups of this pass
    int rsplit;                 // steps r <= rsplit lie in the first of them
    float *binrow;              // bins (or sorted posterior cells) of the pass' first step
    int brow;                   // floats per step
};

// One ring pass (CTC_RING steps) of one scan wave.  Local step s of the phase: time t = t0 + s (DIR 0) / t0 - s (DIR 1);
// x0 / row0 point at local step 0's logits row / lattice row and move with the same sign.  PH2 = false: as the legacy
// scan, rows stored (bounds-checked).  PH2 = true: nothing is stored; the partner's stored row of the same frame
// (pv, prefetched CTC_RING steps ahead like the gathers) meets the fresh row and the posteriors go to the LDS bins.
template <int PPL, int DIR, bool GUARD, bool HASIN, bool HASOUT, bool PH2, bool SORTED>
__device__ __forceinline__ void mm_ring_pass(int s0, int n, const float *__restrict__ x0, size_t rowstride, int lane,
                                             const unsigned (&cls)[PPL], const bool (&valid)[PPL],
                                             const bool (&skip)[PPL], float *__restrict__ row0, int srow,
                                             unsigned rowbytes, double *__restrict__ coff_out, int coff_stride,
                                             float (&px)[CTC_RING][PPL], float (&pv)[CTC_RING][PPL], float (&a)[PPL],
                                             double &coff, float &mpend, const CtcHand *hin, CtcHand *hout, int cbuf,
                                             int q, const MmPh2 &ph, const unsigned (&binoff)[PPL])
{
    const int hs = q * CTC_RING;
    const long long gstep = DIR == 0 ? (long long)rowstride : -(long long)rowstride;
    const long long sstep = DIR == 0 ? (long long)srow : -(long long)srow;
    const float *grow = x0 + (long long)(s0 + CTC_RING) * gstep;              // unguarded only
    float *srowp = row0 + (long long)s0 * sstep;                               // lattice row of step s
    float b1[CTC_RING], b2[CTC_RING];
    if constexpr (HASIN) {
        const float delta = (float)(hin->coff[cbuf][q] - coff);
#pragma unroll
        for (int r = 0; r < CTC_RING; ++r) {
            if constexpr (PPL == 1) {
                b1[r] = hin->v[cbuf][hs + r][DIR == 0 ? 63 : 0].x + delta;
                b2[r] = hin->v[cbuf][hs + r][DIR == 0 ? 62 : 1].x + delta;
            } else {
                const float2 t2 = hin->v[cbuf][hs + r][DIR == 0 ? 63 : 0];
                b1[r] = t2.y + delta;
                b2[r] = t2.x + delta;
            }
            b1[r] = fmaxf(b1[r], LC_NEG);
            b2[r] = fmaxf(b2[r], LC_NEG);
        }
    }
    if (HASOUT && lane == 0) hout->coff[cbuf][q] = coff;
#pragma unroll
    for (int r = 0; r < CTC_RING; ++r) {
        const int s = s0 + r;
        if (!GUARD || s < n) {
            float e[PPL], pw[PPL];
#pragma unroll
            for (int j = 0; j < PPL; ++j) { e[j] = px[r][j] * LC_LOG2E; pw[j] = PH2 ? pv[r][j] : 0.f; }
            if constexpr (PH2 && SORTED && PPL > 1) {
                // Only the label positions' partner values are used below.  Left alone, the compiler narrows the row load
                // to the used components - a dwordx3 at a 4-byte offset for PPL = 4 - and that load is SLOW (measured: the
                // phase-2 ring pass 17150 instead of 8080 cycles per 16 steps).  Naming all components here, at their
                // point of use, keeps the aligned full-width load without moving its wait.
#pragma unroll
                for (int j = 0; j < PPL; ++j) asm volatile("" : "+v"(pw[j]));
            }
            {   // refill this ring slot with the rows CTC_RING steps ahead (guarded pass: clamped, a redundant load)
                const float *rowp = grow;
                const float *prow = srowp + (long long)CTC_RING * sstep;
                if (GUARD) {
                    const int sn = min(s + CTC_RING, n - 1);
                    rowp = x0 + (long long)sn * gstep;
                    prow = row0 + (long long)sn * sstep;
                }
                const ctc_i32x4 rs = ctc_rsrc(rowp);
#pragma unroll
                for (int j = 0; j < PPL; ++j) px[r][j] = ld_off(rs, cls[j]);
                if constexpr (PH2) mm_load_row<PPL>(prow, rowbytes, pv[r], lane);
                grow += gstep;
            }
            if (!PH2 && (r % CTC_NORM) == 0 && lane == 0) coff_out[(size_t)(s / CTC_NORM) * coff_stride] = coff;
            // posterior of this frame's positions: 2^(own + partner - lp), own = alpha_t (after the update) / beta_t (before)
            auto emit = [&]() {
                if constexpr (PH2) {
                    const float lp = (r <= ph.rsplit) ? ph.lpA : ph.lpB;
                    float *br = ph.binrow + r * ph.brow;
#pragma unroll
                    for (int j = 0; j < PPL; ++j) {
                        if constexpr (SORTED) {      // every LABEL position into its own cell (class order); blank = 1 - sum
                            if (PPL > 1 && (j % 2) == 0) continue;                // an even position is a blank
                            // (positions beyond the lattice carry garbage that may be huge: their cells sort last, but
                            // they share a lane's cell vector with real positions in the prefix sum - keep them at 0)
                            *reinterpret_cast<float *>(reinterpret_cast<char *>(br) + binoff[j]) =
                                valid[j] ? __builtin_amdgcn_exp2f(a[j] + pw[j] - lp) : 0.f;
                        } else {
                            if (PPL > 1 && (j % 2) == 0) continue;                // blanks: 1 - sum of the label classes
                            const float gam = __builtin_amdgcn_exp2f(a[j] + pw[j] - lp);
                            const bool lab = valid[j] && (PPL > 1 || (lane & 1));
                            __hip_atomic_fetch_add(reinterpret_cast<float *>(reinterpret_cast<char *>(br) + binoff[j]),
                                                   lab ? gam : 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
                }
            };
            if (DIR == 0) {
                if constexpr (HASOUT) {
                    // (every lane writes: an exec-masked two-lane store was measured at +40 cycles per step of the chain)
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
                float nn[PPL];
#pragma unroll
                for (int j = 0; j < PPL; ++j) {
                    const float s1 = (j >= 1) ? a[j >= 1 ? j - 1 : 0] : p1;
                    const float s2 = (j >= 2) ? a[j >= 2 ? j - 2 : 0] : (j == 1 ? p1 : p2);
                    if ((PPL % 2 == 0) && (j % 2 == 0)) nn[j] = lse2_2(a[j], s1) + e[j];
                    else nn[j] = lse3_2(a[j], s1, skip[j] ? s2 : LC_NEG) + e[j];
                }
#pragma unroll
                for (int j = 0; j < PPL; ++j) a[j] = nn[j];
                if constexpr (PH2) emit();
                else mm_store_row<PPL>(srowp, rowbytes, a, lane);
            } else {
                if constexpr (PH2) emit();
                else mm_store_row<PPL>(srowp, rowbytes, a, lane);
                float g[PPL];
#pragma unroll
                for (int j = 0; j < PPL; ++j) g[j] = a[j] + e[j];
                if constexpr (HASOUT) {
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
                float nn[PPL];
#pragma unroll
                for (int j = 0; j < PPL; ++j) {
                    const float s1 = (j + 1 < PPL) ? g[j + 1 < PPL ? j + 1 : 0] : n1;
                    const float s2 = (j + 2 < PPL) ? g[j + 2 < PPL ? j + 2 : 0] : (j + 2 == PPL ? n1 : n2);
                    if ((PPL % 2 == 0) && (j % 2 == 0)) nn[j] = lse2_2(g[j], s1);
                    else nn[j] = lse3_2(g[j], s1, skip[j] ? s2 : LC_NEG);
                }
#pragma unroll
                for (int j = 0; j < PPL; ++j) a[j] = nn[j];
            }
            srowp += sstep;
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

// every wave of the workgroup, the gradient wave included, meets here once per pipeline iteration
__device__ __forceinline__ void mm_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

struct MmPartner {              // phase 2: the partner workgroup's row-group offsets, [group][4] (its segment order)
    const double *coffp;
    double logp;                // log2 p of the utterance
    float *bins;
    int brow, nslot;
    unsigned long long *dbg;    // stamps of this wave (NULL: none): [iteration][8]
};

// One scan wave's share of a phase: `lag` idle iterations, the iterations of unguarded passes, the guarded tail, idle
// iterations up to `total` barriers.
template <int PPL, int NW, int DIR, bool HASIN, bool HASOUT, bool PH2, bool SORTED>
__device__ __forceinline__ void mm_wave_loop(int lag, int total, int n, const float *__restrict__ x0, size_t rowstride,
                                             int lane, int seg, const unsigned (&cls)[PPL], const bool (&valid)[PPL],
                                             const bool (&skip)[PPL], float *__restrict__ row0, int srow,
                                             unsigned rowbytes, double *__restrict__ coff_out,
                                             float (&px)[CTC_RING][PPL], float (&pv)[CTC_RING][PPL], float (&a)[PPL],
                                             double &coff, const CtcHand *hin, CtcHand *hout, const MmPartner &pt,
                                             const unsigned (&binoff)[PPL])
{
    float mpend = 0.f;
    const int npass = (n + CTC_RING - 1) / CTC_RING;
    const int nun = max(0, n / CTC_RING - 1);
    const int nit = (npass + CTC_PIPE - 1) / CTC_PIPE;
    // Partner offsets (phase 2): my local step s meets the row the partner stored at ITS local step n-1-s, i.e. row
    // group (n-1-s) >> 3: within a pass the group index drops by one after step r = (n-1) & 7.  The offsets are
    // fetched a pass ahead (vector loads: a scalar load would retire out of order with the LDS traffic).
    double oA = 0.0, oB = 0.0, oC = 0.0;
    int gnext = 0;
    auto poff = [&](int g) {
        int idx = max(g, 0) * 4 + seg;
        asm volatile("" : "+v"(idx));
        return pt.coffp[idx];
    };
    if constexpr (PH2) {
        const int g0 = (n - 1) >> 3;
        oA = poff(g0); oB = poff(g0 - 1); oC = poff(g0 - 2);
        gnext = g0 - 3;
    }
    MmPh2 ph;
    ph.rsplit = PH2 ? ((n - 1) & 7) : 0;
    ph.brow = pt.brow;
    auto prep = [&](int cc) {
        if constexpr (PH2) {
            ph.lpA = (float)(pt.logp - coff - oA);
            ph.lpB = (float)(pt.logp - coff - oB);
            ph.binrow = pt.bins + (size_t)((cc * CTC_RING) % pt.nslot) * pt.brow;
            oA = oB; oB = oC; oC = poff(gnext);
            gnext -= 1;
        }
    };
    for (int i = 0; i < lag; ++i) mm_barrier();
    int it = 0;
    for (; (it + 1) * CTC_PIPE <= nun; ++it) {
        if (pt.dbg && lane == 0 && it < 512) pt.dbg[it * 8 + 0] = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int q = 0; q < CTC_PIPE; ++q) {
            prep(it * CTC_PIPE + q);
            mm_ring_pass<PPL, DIR, false, HASIN, HASOUT, PH2, SORTED>((it * CTC_PIPE + q) * CTC_RING, n, x0, rowstride, lane, cls,
                                                               valid, skip, row0, srow, rowbytes, coff_out, 4, px, pv, a,
                                                               coff, mpend, hin, hout, it & 1, q, ph, binoff);
        }
        if (pt.dbg && lane == 0 && it < 512) pt.dbg[it * 8 + 1] = __builtin_amdgcn_s_memtime();
        mm_barrier();
        if (pt.dbg && lane == 0 && it < 512) pt.dbg[it * 8 + 2] = __builtin_amdgcn_s_memtime();
    }
    for (; it < nit; ++it) {
        for (int q = 0; q < CTC_PIPE; ++q) {
            const int cc = it * CTC_PIPE + q;
            if (cc < npass) {
                prep(cc);
                mm_ring_pass<PPL, DIR, true, HASIN, HASOUT, PH2, SORTED>(cc * CTC_RING, n, x0, rowstride, lane, cls, valid, skip,
                                                                  row0, srow, rowbytes, coff_out, 4, px, pv, a, coff, mpend,
                                                                  hin, hout, it & 1, q, ph, binoff);
            }
        }
        mm_barrier();
    }
    for (int i = lag + nit; i < total; ++i) mm_barrier();
}

// The fifth ("frame") wave works on FOUR frames at a time: lane group g = lane >> 4 owns one frame, its 16 lanes hold
// KG = ceil(V / 16) classes each (class l + 16 k), reductions over a frame are 4 DPP row rotations (all-reduce inside a
// row of 16 lanes).  A pipeline iteration (16 frames) is 4 such passes; the loads of the NEXT iteration are issued
// before the current one is worked on, so the wave never waits for a memory round trip inside an iteration.
__device__ __forceinline__ float lc_row16_allmax(float v)
{
#define LC_ROR(n) __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x120 + n, 0xf, 0xf, false))
    v = fmaxf(v, LC_ROR(1)); v = fmaxf(v, LC_ROR(2)); v = fmaxf(v, LC_ROR(4)); v = fmaxf(v, LC_ROR(8));
    return v;
}
__device__ __forceinline__ float lc_row16_allsum(float v)
{
    v += LC_ROR(1); v += LC_ROR(2); v += LC_ROR(4); v += LC_ROR(8);
#undef LC_ROR
    return v;
}
// One frame wave per scan wave: a scan wave keeps its SIMD ~80 % busy (five transcendentals per step), so a single frame
// wave sharing one of the SIMDs took two scan iterations per iteration of its own (s_memtime: 7500 vs 4100 cycles) and
// every scan wave waited for it at the barrier.  With NF = NW frame waves each SIMD hosts one scan wave and one frame wave
// doing 1/NF of the frame work: frame wave f owns the iteration's frames [16 f / NF, 16 (f + 1) / NF) from first to last.
constexpr int mm_nf(int nw) { return nw; }
// posterior cells per frame row of the class-sorted path (positions <= 256): one per label position
constexpr int mm_cells(int grow) { return grow / 2 < 64 ? 64 : grow / 2; }
constexpr int mm_brow(int ppl, int nw) { return mm_cells(64 * nw * ppl) + (ppl == 1 ? 64 : 0); }
template <int KG, int NP>
struct MmFrames {
    float x[NP][KG];     // logits of this wave's NP passes (4 frames each)
    float lse[NP];       // phase 2: their log-sum-exp
};
// frames of local steps [base, base + 16) of a range of n frames starting at time tt0 (DIR sign): issue the loads
template <int KG, int NP, bool WITH_LSE>
__device__ __forceinline__ void mm_frames_load(MmFrames<KG, NP> &f, int base, int n, int tt0, int dir, const float *xb,
                                               size_t rowstride, const float *rlse, int B, int b, int V, int lane)
{
    const int g = lane >> 4, l = lane & 15;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int s = max(min(base + 4 * q + g, n - 1), 0), t = dir == 0 ? tt0 + s : tt0 - s;
        const float *row = xb + (size_t)t * rowstride;
#pragma unroll
        for (int k = 0; k < KG; ++k) f.x[q][k] = row[min(l + 16 * k, V - 1)];
        if constexpr (WITH_LSE) f.lse[q] = rlse[(size_t)t * B + b];
    }
}

// blockIdx -> (utterance, direction): the two workgroups of an utterance get indices 8 apart, i.e. (round-robin
// dispatch, observed not promised) the same XCD and one L2 for the logits rows both gather.  Speed only.
template <int PPL, int NW, int KG, int PH>
__global__ __launch_bounds__((NW + mm_nf(NW)) * 64) void ctc_mm_kernel(MmArgs p)
{
    extern __shared__ __attribute__((aligned(16))) float mm_bins[];        // phase 2: [NSLOT][brow]
    __shared__ CtcHand hand[NW > 1 ? NW - 1 : 1];
    __shared__ double seglp[4];
    __shared__ double s_logp;
    constexpr int NSLOT = MM_IT * (NW + 1);
    constexpr bool PH2 = PH >= 2;                        // PH == 3: phase 2 that takes the frames' log-sum-exp itself (MmArgs::lse2)
    constexpr bool LSE2 = PH == 3;
    constexpr int NF = mm_nf(NW), NP = 4 / NF, FR = MM_IT / NF;     // frame waves; passes / frames per wave and iteration
    __shared__ double s_lsum[NF];
    constexpr int GROW = 64 * NW * PPL;                  // lattice positions the scan waves cover
    constexpr bool SORTED = GROW <= 256;                 // posterior cells in class order (else: per-class ds_add bins)
    // SORTED: one cell per LABEL position (the odd ones), in class order; the blank class is 1 - their sum.  Half the
    // LDS, half the deposits and half the prefix-sum work of a cell per position - and the LDS is what decides how many
    // workgroups of the many-utterance geometry are resident.
    constexpr int CW = mm_cells(GROW);                   // cells per frame row (a multiple of 64)
    constexpr int CPL = CW / 64;                         // cells per lane of the frame wave's prefix sum
    constexpr int BROW = mm_brow(PPL, NW);               // row pitch: PPL == 1 adds dump cells for the blank lanes
    __shared__ unsigned char s_cls[SORTED ? GROW : 1];   // class of every position (255: beyond the lattice)
    __shared__ unsigned short s_perm[SORTED ? GROW : 1]; // rank of every position in class order
    __shared__ int s_cnt[SORTED ? 128 : 1], s_first[SORTED ? 128 : 1];   // positions per class / first cell of a class
    const int bi = blockIdx.x;
    const int b = (bi >> 4) * 8 + (bi & 7), dir = (bi >> 3) & 1;
    if (b >= p.B) return;
    const int T = p.T, B = p.B, V = p.V;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int off0 = p.offs[b], L = p.offs[b + 1] - off0, Tb = min(p.seq_len[b], T), U = 2 * L + 1;
    const size_t rowstride = (size_t)B * V;
    bool skipped = (L > Tb || Tb <= 0);            // ignore_longer_outputs_than_inputs=True: loss 0, gradient 0
    int bad = 0;                                   // a label outside [0, V-1): NaN loss, zero gradient (see the legacy scan)
    if (!skipped) {
        for (int i = threadIdx.x; i < L; i += blockDim.x) {
            const int lab = p.labels[off0 + i];
            bad |= (lab < 0 || lab >= V - 1);
        }
        bad = __syncthreads_or(bad);
    }
    if (skipped || bad) {
        if (PH2) {
            if (threadIdx.x == 0 && dir == 0) p.loss[b] = bad ? __builtin_nanf("") : 0.f;
            if (p.grad) {                          // each workgroup clears its half of the frames
                const int ta = dir == 0 ? T / 2 : 0, tb = dir == 0 ? T : T / 2;
                for (long long i = threadIdx.x; i < (long long)(tb - ta) * V; i += blockDim.x)
                    p.grad[((size_t)(ta + i / V) * B + b) * V + i % V] = 0.f;
            }
        }
        return;
    }
    const int M = Tb / 2;
    // phase 1: alpha t = 0..M-1, beta t = Tb-1..M;  phase 2: alpha t = M..Tb-1, beta t = M-1..0
    const int n1 = dir == 0 ? M : Tb - M, n2 = dir == 0 ? Tb - M : M;
    const int n = PH2 ? n2 : n1;
    const int t0 = PH2 ? (dir == 0 ? M : M - 1) : (dir == 0 ? 0 : Tb - 1);
    const bool scan = wave < NW;
    const int fw = scan ? 0 : wave - NW;                // frame wave number
    const int fbase = fw * FR;                          // its first frame within an iteration
    const int seg = scan ? wave : 0, ubase = seg * 64 * PPL;
    const int blank = V - 1;
    const float *xb = p.logits + (size_t)b * V;
    const float *x0 = xb + (long long)max(t0, 0) * (long long)rowstride;
    float *latb = p.lat + (size_t)b * T * p.srow;
    float *row0 = latb + (long long)max(t0, 0) * p.srow + ubase;
    // bytes of this wave's segment that hold lattice positions, rounded up to whole per-lane vectors (a vector store that
    // straddles the limit must not be dropped as a whole); the row pitch leaves room for the round-up
    const unsigned rowbytes = (unsigned)((max(0, min(U - ubase, 64 * PPL)) + PPL - 1) / PPL * PPL) * 4u;
    double *coffme = p.coff + ((size_t)(b * 2 + dir) * p.ngroups) * 4;
    const double *coffpt = p.coff + ((size_t)(b * 2 + (dir ^ 1)) * p.ngroups) * 4;
    const int nitp = ((n + CTC_RING - 1) / CTC_RING + CTC_PIPE - 1) / CTC_PIPE;
    // barriers per wave: phase 2's frame waves lag the last scan wave by one iteration (they consume its deposits);
    // phase 1's (frame statistics only) do not
    const int total = nitp + NW - (PH2 ? 0 : 1);
    const int brow = SORTED ? BROW : ((V + 15) & ~15) + 64;
    unsigned cls[PPL], binoff[PPL];
    bool valid[PPL], skip[PPL];
    float a[PPL], px[CTC_RING][PPL], pv[CTC_RING][PPL];
    double coff = 0.0;
    if (scan) {
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
            const int u = ubase + lane * PPL + j;
            valid[j] = u < U;
            const bool odd = (u & 1) && valid[j];
            const int lab = odd ? p.labels[off0 + (u >> 1)] : blank;
            cls[j] = (unsigned)lab * 4u;
            asm volatile("" : "+v"(cls[j]));          // keep the gather a vector load (see the legacy scan)
            binoff[j] = odd ? (unsigned)lab * 4u : (unsigned)(((V + 15) & ~15) + lane) * 4u;
            if (dir == 0) skip[j] = odd && u >= 3 && lab != p.labels[off0 + ((u - 3) >> 1)];
            else skip[j] = odd && (u + 2 < U) && lab != p.labels[off0 + ((u + 1) >> 1)];
        }
    }
    bool nopath = false;
    if constexpr (!PH2) {
        if (scan) {
#pragma unroll
            for (int j = 0; j < PPL; ++j) {
                const int u = ubase + lane * PPL + j;
                if (dir == 0) a[j] = (u == 0) ? 0.f : LC_NEG;
                else a[j] = (u < U && u >= U - 2) ? 0.f : LC_NEG;
            }
        }
    } else {
        // resume: own carried row, and log2 p = LSE_u(alpha_{M-1}[u] + beta_{M-1}[u]) from both carried rows (the same
        // arithmetic in both workgroups of the utterance, so both use the same value)
        float m = LC_NEG, ssum = 0.f;
        if (scan) {
            float pa[PPL];
            mm_load_row<PPL>(p.carry + (size_t)(b * 2 + dir) * p.cw + ubase, 64 * PPL * 4, a, lane);
            mm_load_row<PPL>(p.carry + (size_t)(b * 2 + (dir ^ 1)) * p.cw + ubase, 64 * PPL * 4, pa, lane);
            coff = p.carry_off[(b * 2 + dir) * 4 + seg];
            float v[PPL];
#pragma unroll
            for (int j = 0; j < PPL; ++j) { v[j] = valid[j] ? fmaxf(a[j] + pa[j], LC_NEG) : LC_NEG; m = fmaxf(m, v[j]); }
            m = lc_wave_max(m);
#pragma unroll
            for (int j = 0; j < PPL; ++j) ssum += __builtin_amdgcn_exp2f(v[j] - m);
            ssum = lc_wave_sum(ssum);
            if (lane == 0)
                seglp[seg] = (m < -1.0e29f) ? -1.0e300
                                            : (double)(m + __builtin_amdgcn_logf(ssum)) + coff +
                                                  p.carry_off[(b * 2 + (dir ^ 1)) * 4 + seg];
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            double mx = -1.0e300;
            for (int w = 0; w < NW; ++w) mx = seglp[w] > mx ? seglp[w] : mx;
            double lp = mx;
            if (mx > -1.0e299) {
                double acc = 0.0;
                for (int w = 0; w < NW; ++w) acc += seglp[w] > -1.0e299 ? exp2(seglp[w] - mx) : 0.0;
                lp = mx + log2(acc);
            }
            s_logp = lp;
            if (dir == 0 && !LSE2) {              // TF: no valid path => loss = +inf (and the gradient is the softmax)
                const double lsesum = p.lsepart[b * 2] + p.lsepart[b * 2 + 1];
                p.loss[b] = lp > -1.0e299 ? (float)(lsesum - lp * LC_LN2) : INFINITY;
            }
        }
        for (int i = threadIdx.x; i < NSLOT * brow; i += blockDim.x) mm_bins[i] = 0.f;    // (SORTED: cells nobody owns)
        __syncthreads();
        nopath = !(s_logp > -1.0e299);
        if (!p.grad) return;
        if constexpr (SORTED) {
            // label positions in class order, once per launch (a counting sort over the labels staged in LDS); label
            // positions beyond the lattice keep their own index (>= L: behind every real cell)
            for (int u = threadIdx.x; u < GROW; u += blockDim.x)
                s_cls[u] = u >= U ? 255 : ((u & 1) ? (unsigned char)p.labels[off0 + (u >> 1)] : (unsigned char)blank);
            __syncthreads();
            for (int c = threadIdx.x; c < V; c += blockDim.x) {
                int cnt = 0;
#pragma unroll 8
                for (int i = 0; i < L; ++i) cnt += s_cls[2 * i + 1] == c;
                s_cnt[c] = cnt;                                  // (blank: no label carries it -> 0 cells)
            }
            __syncthreads();
            for (int c = threadIdx.x; c < V; c += blockDim.x) {
                int first = 0;
                for (int w = 0; w < c; ++w) first += s_cnt[w];
                s_first[c] = first;
            }
            __syncthreads();
            for (int u = threadIdx.x; u < GROW; u += blockDim.x) {
                int rank = u >> 1;
                if (u < U && (u & 1)) {
                    const int c = s_cls[u];
                    int before = 0;
#pragma unroll 8
                    for (int i = 0; i < (u >> 1); ++i) before += s_cls[2 * i + 1] == c;
                    rank = s_first[c] + before;
                }
                s_perm[u] = (unsigned short)rank;
            }
            __syncthreads();
            if (scan) {
#pragma unroll
                for (int j = 0; j < PPL; ++j) {
                    const int u = ubase + lane * PPL + j;
                    // blank lanes (PPL == 1 only): a dump cell of their own behind the row (one shared cell serialises)
                    binoff[j] = (u & 1) ? (unsigned)s_perm[u] * 4u : (unsigned)(CW + lane) * 4u;
                }
            }
        }
    }
    if (scan) {
        // the scan waves are the critical path: the frame wave (which shares a SIMD with one of them) only gets the
        // issue slots they leave
        __builtin_amdgcn_s_setprio(2);
        // prime the rings with local steps 0..CTC_RING-1 (clamped)
#pragma unroll
        for (int r = 0; r < CTC_RING; ++r) {
            const int sn = max(min(r, n - 1), 0);
            const long long so = dir == 0 ? sn : -sn;
            const ctc_i32x4 rs = ctc_rsrc(x0 + so * (long long)rowstride);
#pragma unroll
            for (int j = 0; j < PPL; ++j) px[r][j] = ld_off(rs, cls[j]);
            if constexpr (PH2) mm_load_row<PPL>(row0 + so * p.srow, rowbytes, pv[r], lane);
            else {
#pragma unroll
                for (int j = 0; j < PPL; ++j) pv[r][j] = 0.f;
            }
        }
        MmPartner pt;
        pt.coffp = coffpt; pt.logp = PH2 ? s_logp : 0.0; pt.bins = mm_bins; pt.brow = brow; pt.nslot = NSLOT;
        pt.dbg = (p.dbg && blockIdx.x == 0) ? p.dbg + ((PH2 ? 1 : 0) * 5 + wave) * 4096 : nullptr;
        if (PH2 && nopath) pt.logp = 1.0e300;          // 2^(x - huge) = 0: no posterior mass anywhere, gradient = softmax
        double *cme = coffme + seg;
#define LC_MM(DIR, HASIN, HASOUT, LAG, HIN, HOUT)                                                                    \
    mm_wave_loop<PPL, NW, DIR, HASIN, HASOUT, PH2, SORTED>(LAG, total, n, x0, rowstride, lane, seg, cls, valid, skip, row0,    \
                                                   p.srow, rowbytes, cme, px, pv, a, coff, HIN, HOUT, pt, binoff)
        if (dir == 0) {
            const CtcHand *hin = &hand[seg > 0 ? seg - 1 : 0];
            CtcHand *hout = &hand[seg < NW - 1 ? seg : 0];
            if (NW == 1) LC_MM(0, false, false, 0, hin, hout);
            else if (seg == 0) LC_MM(0, false, true, 0, hin, hout);
            else if (seg == NW - 1) LC_MM(0, true, false, seg, hin, hout);
            else LC_MM(0, true, true, seg, hin, hout);
        } else {
            const CtcHand *hin = &hand[seg < NW - 1 ? seg : 0];
            CtcHand *hout = &hand[seg > 0 ? seg - 1 : 0];
            if (NW == 1) LC_MM(1, false, false, 0, hin, hout);
            else if (seg == NW - 1) LC_MM(1, false, true, 0, hin, hout);
            else if (seg == 0) LC_MM(1, true, false, NW - 1, hin, hout);
            else LC_MM(1, true, true, NW - 1 - seg, hin, hout);
        }
#undef LC_MM
        if constexpr (!PH2) {            // hand the recursion over to phase 2
            mm_store_row<PPL>(p.carry + (size_t)(b * 2 + dir) * p.cw + ubase, 64 * PPL * 4, a, lane);
            if (lane == 0) p.carry_off[(b * 2 + dir) * 4 + seg] = coff;
        }
    } else if (!PH2 && p.lse2) {
        // frame wave, phase 1, nothing to do: phase 2 takes the statistics (a finished wave leaves the barrier's count)
        if (dir == 0 && fw == 0 && lane == 0) p.loss[b] = 0.f;      // the two atomic adds of phase 2 start from here
    } else if constexpr (!PH2) {
        // frame wave, phase 1: log-sum-exp of the frames of this workgroup's phase 2 (n2 of them from time tt0), 16 per
        // pipeline iteration, the rest after the last barrier.  The iteration body is branch-free and the two register
        // sets alternate (no copies): the loads issued at the top of an iteration are first read in the NEXT one, a
        // whole scan iteration later, so the wave never sits out a memory round trip between two barriers.
        const int tt0 = dir == 0 ? M : M - 1;
        const int g = lane >> 4, l = lane & 15;
        double acc = 0.0;                                   // per lane group: the frames it owned
        const ctc_i32x4 lse_rs = ctc_rsrc_n(p.rlse, (unsigned)min((size_t)T * B * 4, (size_t)0xfffffff0u));
        MmFrames<KG, NP> fa, fb;
        auto step = [&](MmFrames<KG, NP> &cur, MmFrames<KG, NP> &nxt, int base) {
            mm_frames_load<KG, NP, false>(nxt, base + MM_IT + fbase, n2, tt0, dir, xb, rowstride, p.rlse, B, b, V, lane);
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const int s = base + fbase + 4 * q + g, t = dir == 0 ? tt0 + s : tt0 - s;
                float m = -INFINITY;
#pragma unroll
                for (int k = 0; k < KG; ++k) m = fmaxf(m, (l + 16 * k < V) ? cur.x[q][k] : -INFINITY);
                m = lc_row16_allmax(m);
                float sum = 0.f;
#pragma unroll
                for (int k = 0; k < KG; ++k)
                    sum += (l + 16 * k < V) ? __builtin_amdgcn_exp2f((cur.x[q][k] - m) * LC_LOG2E) : 0.f;
                sum = lc_row16_allsum(sum);
                const float lse = m + (float)LC_LN2 * __builtin_amdgcn_logf(sum);
                const bool live = s < n2;
                acc += live ? (double)lse : 0.0;
                // one lane per live frame stores; the others aim beyond the descriptor's range (dropped, no branch)
                lc_ctc_buffer_store_f32(lse, lse_rs, (live && l == 0) ? (int)(((size_t)t * B + b) * 4) : -16, 0, 0);
            }
        };
        const int nwork = (n2 + MM_IT - 1) / MM_IT, npair = min(nwork, total) / 2;
        mm_frames_load<KG, NP, false>(fa, fbase, n2, tt0, dir, xb, rowstride, p.rlse, B, b, V, lane);
        int k = 0;
        for (int pr = 0; pr < npair; ++pr, k += 2) {
            step(fa, fb, k * MM_IT);
            mm_barrier();
            step(fb, fa, (k + 1) * MM_IT);
            mm_barrier();
        }
        // leftovers: (work + barrier) while both remain, then whichever is left
        bool in_a = true;
        for (; k < max(nwork, total); ++k) {
            if (k < nwork) {
                if (in_a) step(fa, fb, k * MM_IT); else step(fb, fa, k * MM_IT);
                in_a = !in_a;
            }
            if (k < total) mm_barrier();
        }
        // the four lane groups' sums (lanes 0, 16, 32, 48 hold one each)
        acc = __shfl(acc, 0, 64) + __shfl(acc, 16, 64) + __shfl(acc, 32, 64) + __shfl(acc, 48, 64);
        if (lane == 0) s_lsum[fw] = acc;
    } else {
        // frame wave, phase 2: grad[t,:] = softmax(logits[t,:]) - posterior, one pipeline iteration behind the last
        // scan wave.  SORTED: the frame's row holds every position's posterior in class order; part A turns each of
        // the iteration's 16 rows into its inclusive prefix sum in place (whole wave per row: CPL cells per lane, DPP
        // wave scan), part B (four frames per pass, 16 lanes each) takes class mass = prefix[last] - prefix[first - 1].
        // Otherwise (> 256 positions): label classes come from the ds_add bins (cleared for their next use), blank =
        // 1 - their sum.
        const int g = lane >> 4, l = lane & 15;
        int cfirst[KG], clast[KG];                     // SORTED: cell range of this lane's classes (last < first: none)
        if constexpr (SORTED) {
#pragma unroll
            for (int k = 0; k < KG; ++k) {
                const int c = min(l + 16 * k, V - 1);
                cfirst[k] = c == V - 1 ? 0 : s_first[c];            // blank: the range of ALL label cells (1 - their sum)
                clast[k] = (l + 16 * k < V) ? (c == V - 1 ? L - 1 : s_first[c] + s_cnt[c] - 1) : -1;
            }
        }
        const ctc_i32x4 grad_rs = ctc_rsrc_n(p.grad + (size_t)b * V, (unsigned)min(((size_t)T * B - b) * V * 4, (size_t)0xfffffff0u));
        MmFrames<KG, NP> fa, fb;
        constexpr bool lse2 = LSE2;
        double lacc = 0.0;                             // lse2: sum of the log-sum-exp of the frames this lane group owned
        // iteration j (this wave's FR frames of local steps [16 j, 16 j + 16)); branch-free: frames past the end are
        // clamped for the loads and their stores aim beyond the descriptor's range.  Register sets alternate as in phase 1.
        auto step = [&](MmFrames<KG, NP> &cur, MmFrames<KG, NP> &nxt, int j) {
            LC_CSTAMP(2, NW, j, 0);
            mm_frames_load<KG, NP, !lse2>(nxt, (j + 1) * MM_IT + fbase, n, t0, dir, xb, rowstride, p.rlse, B, b, V, lane);
            LC_CSTAMP(2, NW, j, 1);
            if constexpr (lse2) {
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    float m = -INFINITY;
#pragma unroll
                    for (int k = 0; k < KG; ++k) m = fmaxf(m, (l + 16 * k < V) ? cur.x[q][k] : -INFINITY);
                    m = lc_row16_allmax(m);
                    float sum = 0.f;
#pragma unroll
                    for (int k = 0; k < KG; ++k)
                        sum += (l + 16 * k < V) ? __builtin_amdgcn_exp2f((cur.x[q][k] - m) * LC_LOG2E) : 0.f;
                    sum = lc_row16_allsum(sum);
                    cur.lse[q] = m + (float)LC_LN2 * __builtin_amdgcn_logf(sum);
                    lacc += (j * MM_IT + fbase + 4 * q + g < n) ? (double)cur.lse[q] : 0.0;
                }
            }
            if constexpr (SORTED) {
                // part A (rows past the end: harmless garbage).  All 16 rows are read before any is written back: the
                // rows are independent, but only this order lets the compiler interleave their DPP chains (measured:
                // row after row 258 cycles per row - the chain latency - vs the ~20 instructions it is).
                float c[FR][CPL];
#pragma unroll
                for (int r = 0; r < FR; ++r) {
                    const float *row = mm_bins + (size_t)((j * MM_IT + fbase + r) % NSLOT) * BROW + lane * CPL;
                    if constexpr (CPL == 4) {
                        const float4 v = *reinterpret_cast<const float4 *>(row);
                        c[r][0] = v.x; c[r][1] = v.y; c[r][2] = v.z; c[r][3] = v.w;
                    } else if constexpr (CPL == 2) {
                        const float2 v = *reinterpret_cast<const float2 *>(row);
                        c[r][0] = v.x; c[r][1] = v.y;
                    } else {
                        c[r][0] = row[0];
                    }
                }
#pragma unroll
                for (int r = 0; r < FR; ++r) {
#pragma unroll
                    for (int q = 1; q < CPL; ++q) c[r][q] += c[r][q - 1];
                    float v = c[r][CPL - 1];                  // inclusive scan over the wave (DPP, zero fill)
#define LC_SCAN_STEP(ctrl, rowmask)                                                                              \
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rowmask, 0xf, true))
                    LC_SCAN_STEP(0x111, 0xf); LC_SCAN_STEP(0x112, 0xf); LC_SCAN_STEP(0x114, 0xf);
                    LC_SCAN_STEP(0x118, 0xf); LC_SCAN_STEP(0x142, 0xa); LC_SCAN_STEP(0x143, 0xc);
#undef LC_SCAN_STEP
                    const float excl = lc_wave_shr1(v, 0.f);      // the lanes below (a shift, not v - own: no cancellation)
#pragma unroll
                    for (int q = 0; q < CPL; ++q) c[r][q] += excl;
                }
#pragma unroll
                for (int r = 0; r < FR; ++r) {
                    float *row = mm_bins + (size_t)((j * MM_IT + fbase + r) % NSLOT) * BROW + lane * CPL;
                    if constexpr (CPL == 4) *reinterpret_cast<float4 *>(row) = make_float4(c[r][0], c[r][1], c[r][2], c[r][3]);
                    else if constexpr (CPL == 2) *reinterpret_cast<float2 *>(row) = make_float2(c[r][0], c[r][1]);
                    else row[0] = c[r][0];
                }
            }
            LC_CSTAMP(2, NW, j, 2);
            // part B: all LDS reads of this wave's passes first, then the arithmetic, then the stores
            float bv[NP][KG], labsum[NP];
            bool live[NP];
            int rowoff[NP];
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const int s = j * MM_IT + fbase + 4 * q + g, t = dir == 0 ? t0 + s : t0 - s;
                live[q] = s < n;
                rowoff[q] = (int)((size_t)max(t, 0) * B * V * 4);
                float *br = mm_bins + (size_t)((live[q] ? s : 0) % NSLOT) * brow;
                labsum[q] = 0.f;
#pragma unroll
                for (int k = 0; k < KG; ++k) {
                    const int c = l + 16 * k;
                    if constexpr (SORTED) {
                        const float hi = br[max(clast[k], 0)], lo = br[max(cfirst[k] - 1, 0)];
                        bv[q][k] = clast[k] < cfirst[k] ? 0.f : hi - (cfirst[k] > 0 ? lo : 0.f);
                    } else {
                        bv[q][k] = (c < V && live[q]) ? br[c] : 0.f;
                        if (c < V && live[q]) br[c] = 0.f;
                        labsum[q] += (c < V - 1) ? bv[q][k] : 0.f;
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                if constexpr (!SORTED) labsum[q] = lc_row16_allsum(labsum[q]);
#pragma unroll
                for (int k = 0; k < KG; ++k) {
                    const int c = l + 16 * k;
                    const float y = __builtin_amdgcn_exp2f((cur.x[q][k] - cur.lse[q]) * LC_LOG2E);
                    const float post = nopath ? 0.f : (c == V - 1 ? 1.f - (SORTED ? bv[q][k] : labsum[q]) : bv[q][k]);
                    lc_ctc_buffer_store_f32(y - post, grad_rs, (live[q] && c < V) ? rowoff[q] + c * 4 : -16, 0, 0);
                }
            }
            LC_CSTAMP(2, NW, j, 3);
        };
        for (int i = 0; i < NW; ++i) mm_barrier();             // the lag behind the last scan wave
        mm_frames_load<KG, NP, !lse2>(fa, fbase, n, t0, dir, xb, rowstride, p.rlse, B, b, V, lane);
        int j = 0;
        for (; j + 2 <= nitp; j += 2) {
            step(fa, fb, j);
            mm_barrier();
            LC_CSTAMP(2, NW, j, 4);
            step(fb, fa, j + 1);
            mm_barrier();
            LC_CSTAMP(2, NW, j + 1, 4);
        }
        if (j < nitp) {
            step(fa, fb, j);
            mm_barrier();
        }
        if constexpr (lse2) {
            lacc = __shfl(lacc, 0, 64) + __shfl(lacc, 16, 64) + __shfl(lacc, 32, 64) + __shfl(lacc, 48, 64);
            if (lane == 0) s_lsum[fw] = lacc;
        }
    }
    if (!PH2 && !p.lse2) {               // the frame waves' partial sums of lse_t
        __syncthreads();
        if (threadIdx.x == 0) {
            double tot = 0.0;
            for (int f = 0; f < NF; ++f) tot += s_lsum[f];
            p.lsepart[b * 2 + dir] = tot;
        }
    }
    if constexpr (LSE2) {
        // loss = sum_t lse_t - ln p~: each workgroup adds its share with ONE float atomic - two addends onto the zero phase 1
        // left commute exactly, so the result does not depend on which workgroup comes first.  Both workgroups know ln p~
        // (s_logp, above) and each subtracts HALF of it in double before rounding: the two addends are then of the loss's own
        // magnitude, not of sum_t lse_t's (thousands at T = 1000, where rounding each half to fp32 cost 2.4e-4 absolute -
        // ADVICE round 4), and the result is within ~2 ulp of the loss like the frame-statistics path's single rounding.
        __syncthreads();
        if (threadIdx.x == 0) {
            double tot = 0.0;
            for (int f = 0; f < NF; ++f) tot += s_lsum[f];
            const float add = s_logp > -1.0e299 ? (float)(tot - 0.5 * s_logp * LC_LN2) : dir == 0 ? INFINITY : (float)tot;
            __hip_atomic_fetch_add(p.loss + b, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (PH2 && dir == 0 && Tb < T) {     // frames beyond the utterance: zero gradient (whole workgroup, after its scan)
        for (long long i = threadIdx.x; i < (long long)(T - Tb) * V; i += blockDim.x)
            p.grad[((size_t)(Tb + i / V) * B + b) * V + i % V] = 0.f;
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

// geometry of the meet-in-the-middle path: scan waves per direction / lattice positions per lane for S = 2L+1 positions.
// Few utterances (2B workgroups <= the 256 CUs): the widest pipeline, one position per lane - the chain latency is what
// counts.  Many utterances: every CU is busy anyway and what counts is the instruction total per utterance-step: one
// wave with four positions per lane (no hand-offs, no barriers between scan waves) when the lattice fits.
static inline void mm_geometry(int S, int B, int &ppl, int &nw)
{
    // (measured at B = 512, round 3: two waves x 2 positions per lane shorten a workgroup's iteration from 9900 to 7000
    // cycles but cost more SIMD time per utterance-step in all - 448 vs 340 us: at 2B >= the 1024 SIMDs the call is bound
    // by instruction issue, not by the chain)
    if (B > 128 && S <= 256) { ppl = 4; nw = 1; return; }
    nw = ctc_nw(S);
    ppl = ctc_ppl(S);
}
// (the gradient rows are addressed with 32-bit offsets through one buffer descriptor)
static inline bool mm_supported(int V, int S, int T, int B)
{
    // Lattices of more than 256 positions leave the class-sorted cells for per-class LDS atomics, and there the three-
    // kernel path is the faster one whenever the chain latency counts (B <= 128: 228 vs 289 us at L = 200, 366 vs 478 at
    // L = 450, MI355X, T = 1000) and, beyond 1024 positions, always (8 positions per lane: 603 vs 1078 us at L = 600, B = 64;
    // 4055 vs 4924 at B = 512 - and the meet-in-the-middle instantiation spilled).  With many utterances and 257..1024
    // positions the two-launch path wins (1245 vs 1452 us at L = 200, B = 512).
    if (S > 1024 || (S > 256 && B <= 128)) return false;
    return V <= 128 && (size_t)T * B * V * sizeof(float) < 0xfffffff0ull;
}
struct MmLayout {
    size_t lat, coff, carry, carry_off, rlse, lsepart, total;
    int srow, ngroups, cw;
};
static inline MmLayout mm_layout(int T, int B, int max_label_len)
{
    MmLayout m;
    const int S = 2 * max_label_len + 1;
    int ppl, nw;
    mm_geometry(S, B, ppl, nw);
    m.srow = (S + 15) & ~15;
    m.ngroups = T / CTC_NORM + 2;
    m.cw = nw * 64 * ppl;
    size_t o = 0;
    m.lat = o; o += align256((size_t)B * T * m.srow * sizeof(float));
    m.coff = o; o += align256((size_t)B * 2 * m.ngroups * 4 * sizeof(double));
    m.carry = o; o += align256((size_t)B * 2 * m.cw * sizeof(float));
    m.carry_off = o; o += align256((size_t)B * 2 * 4 * sizeof(double));
    m.rlse = o; o += align256((size_t)T * B * sizeof(float));
    m.lsepart = o; o += align256((size_t)B * 2 * sizeof(double));
    m.total = o;
    return m;
}

static size_t ctc_legacy_workspace_bytes(int T, int B, int max_label_len)
{
    const size_t rows = (size_t)T * B;
    const size_t lat = align256(rows * ctc_srow(max_label_len) * sizeof(float));
    const size_t ng = (size_t)(T + CTC_NORM - 1) / CTC_NORM;
    return 2 * align256(rows * sizeof(float)) + 2 * align256((size_t)B * sizeof(double)) +
           2 * align256((size_t)B * ng * 4 * sizeof(double)) + 2 * lat;
}

// Development hook (not part of the product surface): device buffer of [2 phases][5 waves][512 iterations][8] s_memtime
// stamps of workgroup 0 (tools/ctc_stamps.py).
extern "C" void lc_debug_set_ctc_stamps(unsigned long long *buf) { g_ctc_dbg = buf; }

extern "C" size_t lc_ctc_workspace_bytes(int T, int B, int V, int max_label_len)
{
    if (mm_supported(V, 2 * max_label_len + 1, T, B)) return mm_layout(T, B, max_label_len).total;
    return ctc_legacy_workspace_bytes(T, B, max_label_len);
}

template <int PPL, int NW, int KG>
static int mm_launch(const MmArgs &a, int B, int V, hipStream_t s)
{
    const int nblk = 2 * ((B + 7) / 8 * 8);
    const int grow = 64 * NW * PPL;
    const size_t lds2 = (size_t)MM_IT * (NW + 1) * (grow <= 256 ? mm_brow(PPL, NW) : ((V + 15) & ~15) + 64) * sizeof(float);
    if (hipFuncSetAttribute((const void *)ctc_mm_kernel<PPL, NW, KG, 2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds2) != hipSuccess) {
        (void)hipGetLastError();
        lc_set_error("lc_ctc_loss: hipFuncSetAttribute(MaxDynamicSharedMemorySize = %zu) failed", lds2);
        return LC_ELAUNCH;
    }
    if (a.lse2 && hipFuncSetAttribute((const void *)ctc_mm_kernel<PPL, NW, KG, 3>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds2) != hipSuccess) {
        (void)hipGetLastError();
        lc_set_error("lc_ctc_loss: hipFuncSetAttribute(MaxDynamicSharedMemorySize = %zu) failed", lds2);
        return LC_ELAUNCH;
    }
    hipLaunchKernelGGL((ctc_mm_kernel<PPL, NW, KG, 1>), dim3(nblk), dim3((NW + mm_nf(NW)) * 64), 0, s, a);
    if (a.lse2) hipLaunchKernelGGL((ctc_mm_kernel<PPL, NW, KG, 3>), dim3(nblk), dim3((NW + mm_nf(NW)) * 64), lds2, s, a);
    else hipLaunchKernelGGL((ctc_mm_kernel<PPL, NW, KG, 2>), dim3(nblk), dim3((NW + mm_nf(NW)) * 64), lds2, s, a);
    LC_CHECK_LAUNCH("ctc_mm");
    return LC_OK;
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
    if (mm_supported(V, S, T, B)) {
        const MmLayout m = mm_layout(T, B, max_label_len);
        char *w0 = (char *)workspace;
        MmArgs a;
        a.logits = logits; a.T = T; a.B = B; a.V = V;
        a.labels = labels; a.offs = label_offsets; a.seq_len = seq_len;
        a.lat = (float *)(w0 + m.lat); a.srow = m.srow;
        a.coff = (double *)(w0 + m.coff); a.ngroups = m.ngroups;
        a.carry = (float *)(w0 + m.carry); a.cw = m.cw;
        a.carry_off = (double *)(w0 + m.carry_off);
        a.rlse = (float *)(w0 + m.rlse);
        a.lsepart = (double *)(w0 + m.lsepart);
        a.loss = loss; a.grad = grad;
        a.dbg = g_ctc_dbg;
        // (many utterances and a gradient to write: the frame statistics move to where the softmax is taken; a loss-only call
        // has no phase-2 frame pass to move them to; measured 364 -> 349 us at B = 512, 847 -> 826 at 1024, 243 -> 262 at 256)
        a.lse2 = (grad != nullptr && lc_option(LC_OPT_CTC_LSE2, B >= 512 ? 1 : 0) != 0) ? 1 : 0;
        int ppl, nw;
        mm_geometry(S, B, ppl, nw);
#ifdef LC_CTC_DEV      /* development builds (tools/ctc_dev_build.sh): three instantiations instead of fifty */
#define LC_MMK(PPL, NW) mm_launch<PPL, NW, 3>(a, B, V, s)
        if (nw == 1 && ppl == 4) return LC_MMK(4, 1);
        if (nw == 4 && ppl == 1) return LC_MMK(1, 4);
        lc_set_error("lc_ctc_loss: this development build holds two geometries only");
        return LC_EINVAL;
#undef LC_MMK
#endif
#define LC_MMK(PPL, NW)                                                                                          \
    (V <= 48 ? mm_launch<PPL, NW, 3>(a, B, V, s) : V <= 80 ? mm_launch<PPL, NW, 5>(a, B, V, s) : mm_launch<PPL, NW, 8>(a, B, V, s))
#ifndef LC_CTC_DEV
        if (nw == 1 && ppl == 1) return LC_MMK(1, 1);
        if (nw == 1) return LC_MMK(4, 1);
        if (nw == 2) return LC_MMK(1, 2);
        if (ppl == 1) return LC_MMK(1, 4);
        if (ppl == 2) return LC_MMK(2, 4);
        return LC_MMK(4, 4);
#endif
#undef LC_MMK
    }
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
#ifdef LC_CTC_DEV
    (void)nw; (void)ppl;
    LC_SCAN(8, 4);
#else
    if (nw == 1) LC_SCAN(1, 1);
    else if (nw == 2) LC_SCAN(1, 2);
    else if (ppl == 1) LC_SCAN(1, 4);
    else if (ppl == 2) LC_SCAN(2, 4);
    else if (ppl == 4) LC_SCAN(4, 4);
    else LC_SCAN(8, 4);
#endif
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
