// ctc.hip — CTC loss / gradient / greedy decode for gfx950 (MI355X).
//
// Replaces tf.nn.ctc_loss, tf.nn.ctc_greedy_decoder (CPU-only kernels in TF 1.8) as called at
// mobvoi/lstm_ctc nnet/graph.py:109-114 and :138-142.  Semantics: SURVEY.md Appendix A.3/A.4.
//
// Three launches per loss call, all HBM/latency bound, no MFMA:
//   1. ctc_row_stats   one 16/64-lane group per (t,b) row: max, log-sum-exp, first argmax.
//   2. ctc_scan        one workgroup per utterance, wave 0 = alpha (t ascending), wave 1 = beta
//                      (t descending), run concurrently.  The 2L+1 lattice lives in registers,
//                      PPL consecutive positions per lane; the u-1/u-2 (u+1/u+2) neighbours that
//                      cross a lane boundary move with one DPP wave shift each.  Log2 domain
//                      (v_exp_f32 / v_log_f32 are base 2), emissions taken as (x - rowmax) so the
//                      lattice values stay small; the constant sum_t (max_t - lse_t) is added back
//                      to the loss in double.  Logit gathers are prefetched RING steps ahead.
//   3. ctc_grad        one wave per (t,b) row: posterior mass per class from alpha+beta-logp,
//                      label positions through LDS float atomics, blank positions through a wave
//                      reduction; grad = softmax - posterior.
#include "common.h"

#define LC_NEG (-1.0e30f)
#define LC_LOG2E 1.4426950408889634f
#define LC_LN2 0.6931471805599453

// ------------------------------------------------------------------------------ row stats
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
    if (t >= seq_len[b]) {
        if (l == 0) {
            if (rmax) { rmax[gid] = 0.f; rlse[gid] = 0.f; }
            if (argmax) argmax[gid] = -1;
        }
        return;
    }
    const float *x = logits + (size_t)gid * V;
    float m = -INFINITY;
    int am = 0x7fffffff;
    for (int k = l; k < V; k += G) {
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
        float s = 0.f;
        for (int k = l; k < V; k += G) s += expf(x[k] - m);
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

template <int PPL, int DIR, bool GUARD>
__device__ __forceinline__ void ctc_ring_pass(int s0, const float *__restrict__ xb, size_t rowstride, int Tb, int lane,
                                              const unsigned (&cls)[PPL], const bool (&valid)[PPL],
                                              const bool (&skip)[PPL], float *__restrict__ rows_out, int srow,
                                              double *__restrict__ coff_out, float (&px)[CTC_RING][PPL],
                                              float (&a)[PPL], double &coff, float &mpend)
{
    // running row pointers (wave-uniform): gather row of step s + CTC_RING, lattice row of step s
    const long long gstep = DIR == 0 ? (long long)rowstride : -(long long)rowstride;
    const long long sstep = DIR == 0 ? (long long)srow : -(long long)srow;
    const float *grow = xb + (size_t)(DIR == 0 ? s0 + CTC_RING : Tb - 1 - s0 - CTC_RING) * rowstride;   // unguarded only
    float *srowp = rows_out + (size_t)(DIR == 0 ? s0 : Tb - 1 - s0) * srow;
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
            if ((r % CTC_NORM) == 0 && lane == 0) coff_out[s / CTC_NORM] = coff;   // offset of this row group
            if (DIR == 0) {
                float p1, p2;
                if constexpr (PPL == 1) {
                    p1 = lc_wave_shr1(a[0], LC_NEG);
                    p2 = lc_wave_shr1(p1, LC_NEG);
                } else {
                    p1 = lc_wave_shr1(a[PPL - 1], LC_NEG);
                    p2 = lc_wave_shr1(a[PPL - 2], LC_NEG);
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
                float n1, n2;
                if constexpr (PPL == 1) {
                    n1 = lc_wave_shl1(g[0], LC_NEG);
                    n2 = lc_wave_shl1(n1, LC_NEG);
                } else {
                    n1 = lc_wave_shl1(g[0], LC_NEG);
                    n2 = lc_wave_shl1(g[1], LC_NEG);
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
                for (int j = 0; j < PPL; ++j) a[j] -= mpend;
                coff += (double)mpend;
            }
        }
    }
}

template <int PPL, int DIR>
__device__ __forceinline__ void ctc_recursion(const float *__restrict__ xb, size_t rowstride, int Tb, int lane,
                                              const unsigned (&cls)[PPL], const bool (&valid)[PPL],
                                              const bool (&skip)[PPL], float *__restrict__ rows_out, int srow,
                                              double *__restrict__ coff_out, float (&a)[PPL], double &coff)
{
    float px[CTC_RING][PPL];
#pragma unroll
    for (int r = 0; r < CTC_RING; ++r) {
        const int sn = min(r, Tb - 1);
        const int t = DIR == 0 ? sn : Tb - 1 - sn;
        const ctc_i32x4 rs = ctc_rsrc(xb + (size_t)t * rowstride);
#pragma unroll
        for (int j = 0; j < PPL; ++j) px[r][j] = ld_off(rs, cls[j]);
    }
    coff = 0.0;
    float mpend = 0.f;
    int s0 = 0;
    for (; s0 + 2 * CTC_RING <= Tb; s0 += CTC_RING)      // every prefetch of these passes is in range
        ctc_ring_pass<PPL, DIR, false>(s0, xb, rowstride, Tb, lane, cls, valid, skip, rows_out, srow, coff_out, px, a,
                                       coff, mpend);
    for (; s0 < Tb; s0 += CTC_RING)
        ctc_ring_pass<PPL, DIR, true>(s0, xb, rowstride, Tb, lane, cls, valid, skip, rows_out, srow, coff_out, px, a,
                                      coff, mpend);
}

template <int PPL>
__global__ __launch_bounds__(192) void ctc_scan_kernel(
    const float *__restrict__ logits, int T, int B, int V, const int *__restrict__ labels,
    const int *__restrict__ offs, const int *__restrict__ seq_len, const float *__restrict__ rlse,
    float *__restrict__ alpha, float *__restrict__ beta, int srow, double *__restrict__ coffa,
    double *__restrict__ coffb, int ngroups, float *__restrict__ loss, double *__restrict__ logp2_out,
    int *__restrict__ status)
{
    __shared__ double lse_sum;
    const int b = blockIdx.x;
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int off0 = offs[b];
    const int L = offs[b + 1] - off0;
    const int Tb = min(seq_len[b], T);
    const int U = 2 * L + 1;
    if (L > Tb || Tb <= 0) {   // ignore_longer_outputs_than_inputs=True: utterance skipped
        if (threadIdx.x == 0) { loss[b] = 0.f; logp2_out[b] = 0.0; status[b] = 1; }
        return;
    }
    const int blank = V - 1;
    unsigned cls[PPL];   // byte offset of the lattice position's class within a logits row
    bool valid[PPL], skip[PPL];
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
        const int u = lane * PPL + j;
        valid[j] = u < U;
        const bool odd = (u & 1) && valid[j];
        const int lab = odd ? labels[off0 + (u >> 1)] : blank;
        cls[j] = (unsigned)lab * 4u;
        // Keep the gather index in a VGPR the compiler cannot prove uniform: a uniform (blank) address would
        // become an s_load, and scalar loads retire out of order - every lgkmcnt(0) would then also wait for
        // the prefetch issued a moment ago (measured: 2.4x slower scan).
        asm volatile("" : "+v"(cls[j]));
        if (wave == 0)   // alpha: may u be entered from u-2 ?
            skip[j] = odd && u >= 3 && lab != labels[off0 + ((u - 3) >> 1)];
        else             // beta: may u move on to u+2 ?
            skip[j] = odd && (u + 2 < U) && lab != labels[off0 + ((u + 1) >> 1)];
    }
    const size_t rowstride = (size_t)B * V;
    const float *xb = logits + (size_t)b * V;
    float a[PPL];
    double coff = 0.0;
    if (wave == 0) {
        // virtual row t = -1: all mass on u = 0, so the generic step yields alpha_0 = (e[0], e[1], -inf, ...)
#pragma unroll
        for (int j = 0; j < PPL; ++j) a[j] = (lane * PPL + j == 0) ? 0.f : LC_NEG;
        ctc_recursion<PPL, 0>(xb, rowstride, Tb, lane, cls, valid, skip, alpha + (size_t)b * T * srow, srow,
                              coffa + (size_t)b * ngroups, a, coff);
    } else if (wave == 1) {
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
            const int u = lane * PPL + j;
            a[j] = (valid[j] && u >= U - 2) ? 0.f : LC_NEG;
        }
        ctc_recursion<PPL, 1>(xb, rowstride, Tb, lane, cls, valid, skip, beta + (size_t)b * T * srow, srow,
                              coffb + (size_t)b * ngroups, a, coff);
    } else {
        // wave 2: sum_t lse_t, the softmax normaliser the raw-logit recursion left out
        double acc = 0.0;
        for (int t = lane; t < Tb; t += 64) acc += (double)rlse[(size_t)t * B + b];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (lane == 0) lse_sum = acc;
    }
    __syncthreads();
    if (wave == 0) {
        // log2 p (raw-logit domain) = LSE(alpha[U-1], alpha[U-2]) at t = Tb-1, plus the re-centring offset
        float v1 = -INFINITY, v2 = -INFINITY;
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
            const int u = lane * PPL + j;
            if (u == U - 1) v1 = a[j];
            if (u == U - 2) v2 = a[j];
        }
        v1 = lc_wave_max(v1);
        v2 = lc_wave_max(v2);
        if (U < 2) v2 = LC_NEG;
        const float lp2 = lse2_2(v1, v2);
        if (lane == 0) {
            if (lp2 < -1.0e29f) {   // no valid path (TF: loss = +inf, gradient = softmax)
                loss[b] = INFINITY; logp2_out[b] = 0.0; status[b] = 2;
            } else {
                const double lp = (double)lp2 + coff;
                loss[b] = (float)(lse_sum - lp * LC_LN2);
                logp2_out[b] = lp; status[b] = 0;
            }
        }
    }
}

// ------------------------------------------------------------------------------ gradient
__global__ __launch_bounds__(256) void ctc_grad_kernel(
    const float *__restrict__ logits, int T, int B, int V, const int *__restrict__ labels,
    const int *__restrict__ offs, const int *__restrict__ seq_len, const float *__restrict__ rlse,
    const float *__restrict__ alpha, const float *__restrict__ beta, int srow,
    const double *__restrict__ coffa, const double *__restrict__ coffb, int ngroups,
    const double *__restrict__ logp2, const int *__restrict__ status, float *__restrict__ grad)
{
    extern __shared__ __attribute__((aligned(16))) float bins_all[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave;
    const bool inrange = row < T * B;
    const int b = inrange ? row % B : 0, t = inrange ? row / B : 0;
    const int st = inrange ? status[b] : 1;
    const int Tb = inrange ? min(seq_len[b], T) : 0;
    const bool live = inrange && t < Tb && st != 1;
    float *bins = bins_all + wave * V;
    for (int k = lane; k < V; k += 64) bins[k] = 0.f;
    __syncthreads();
    float blank_acc = 0.f;
    if (live && st == 0) {
        const int off0 = offs[b];
        const int U = 2 * (offs[b + 1] - off0) + 1;
        // stored rows are re-centred: log2(alpha*beta/p) = a + b + (offset_a + offset_b - log2 p), summed in double
        const float lp = (float)(logp2[b] - coffa[(size_t)b * ngroups + t / CTC_NORM] -
                                 coffb[(size_t)b * ngroups + (Tb - 1 - t) / CTC_NORM]);
        const size_t ro = ((size_t)b * T + t) * srow;
        for (int u = lane; u < U; u += 64) {
            const float e = __builtin_amdgcn_exp2f(alpha[ro + u] + beta[ro + u] - lp);
            if (u & 1) atomicAdd(&bins[labels[off0 + (u >> 1)]], e);
            else blank_acc += e;
        }
    }
    blank_acc = lc_wave_sum(blank_acc);
    __syncthreads();
    if (inrange) {
        float *g = grad + (size_t)row * V;
        if (!live) {
            for (int k = lane; k < V; k += 64) g[k] = 0.f;
        } else {
            const float *x = logits + (size_t)row * V;
            const float lse = rlse[row];
            for (int k = lane; k < V; k += 64) {
                const float y = expf(x[k] - lse);
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
static inline int ctc_ppl(int S) { return S <= 64 ? 1 : S <= 128 ? 2 : S <= 256 ? 4 : S <= 512 ? 8 : S <= 1024 ? 16 : 32; }
static inline int ctc_srow(int max_label_len) { return 64 * ctc_ppl(2 * max_label_len + 1); }

extern "C" size_t lc_ctc_workspace_bytes(int T, int B, int V, int max_label_len)
{
    (void)V;
    const size_t rows = (size_t)T * B;
    const size_t lat = align256(rows * ctc_srow(max_label_len) * sizeof(float));
    const size_t ng = (size_t)(T + CTC_NORM - 1) / CTC_NORM;
    return 2 * align256(rows * sizeof(float)) + 2 * align256((size_t)B * sizeof(double)) +
           2 * align256((size_t)B * ng * sizeof(double)) + 2 * lat;
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
    double *coffa = (double *)w; w += align256((size_t)B * ngroups * sizeof(double));
    double *coffb = (double *)w; w += align256((size_t)B * ngroups * sizeof(double));
    const size_t lat = align256(rows * srow * sizeof(float));
    float *alpha = (float *)w; w += lat;
    float *beta = (float *)w;

    launch_row_stats(logits, T, B, V, seq_len, rmax, rlse, nullptr, s);
    LC_CHECK_LAUNCH("ctc_row_stats");
#define LC_SCAN(PPL)                                                                                       \
    hipLaunchKernelGGL(ctc_scan_kernel<PPL>, dim3(B), dim3(192), 0, s, logits, T, B, V, labels, label_offsets, \
                       seq_len, rlse, alpha, beta, srow, coffa, coffb, ngroups, loss, logp2, status)
    switch (ctc_ppl(S)) {
    case 1: LC_SCAN(1); break;
    case 2: LC_SCAN(2); break;
    case 4: LC_SCAN(4); break;
    case 8: LC_SCAN(8); break;
    case 16: LC_SCAN(16); break;
    default: LC_SCAN(32); break;
    }
#undef LC_SCAN
    LC_CHECK_LAUNCH("ctc_scan");
    if (grad) {
        hipLaunchKernelGGL(ctc_grad_kernel, dim3(lc_cdiv(rows, 4)), dim3(256), 4 * V * sizeof(float), s, logits, T,
                           B, V, labels, label_offsets, seq_len, rlse, alpha, beta, srow, coffa, coffb, ngroups, logp2, status, grad);
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
