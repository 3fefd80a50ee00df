// misc.hip — the bandwidth-bound helpers around the GEMMs: dropout scaling, column sums, transpose,
// the fused high-rank (MoE) head combine, the fused L2 + global-norm clip + optimizer update, and
// the posterior transform of nnet-forward.  All are HBM-bound streaming kernels (no MFMA).
#include "common.h"

namespace {

// ------------------------------------------------------------------------------ dropout scale
// DropoutWrapper(output_keep_prob) — mobvoi/lstm_ctc nnet/bilstm.py:128,149.
__global__ void dropout_scale_kernel(const float *__restrict__ x, long long rows, int P, int ldx, float keep,
                                     float inv_keep, uint32_t seed, uint32_t stream_id, float *__restrict__ y,
                                     int ldy, int accumulate)
{
    const long long total = rows * P;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / P;
        const int p = (int)(i - r * P);
        const float f = lc_dropout_factor(seed, stream_id, (uint64_t)i, keep, inv_keep);
        const float v = x[r * ldx + p] * f;
        float *dst = y + r * ldy + p;
        *dst = accumulate ? *dst + v : v;
    }
}

// The same with 16-byte accesses (P, ldx, ldy multiples of 4, aligned pointers): a thread scales four consecutive columns
// of one row (6.9 TB/s on the c4 layer outputs: [64000, 1024] windows of a [64000, 2048] buffer).
// SHADOW: also write the bf16 (round-to-nearest-even) copy of the result to y16 [rows, P] with row pitch ld16 - the
// operand shadow the next product of the bf16 path (config c5) reads, made in the pass that already touches the tensor.
template <bool SHADOW>
__global__ __launch_bounds__(256) void dropout_scale_vec_kernel(const float *__restrict__ x, long long rows, int P, int ldx,
                                                                float keep, float inv_keep, uint32_t seed, uint32_t stream_id,
                                                                float *__restrict__ y, int ldy, int accumulate,
                                                                unsigned short *__restrict__ y16, int ld16)
{
    // flat walk over the (row, column quad) pairs: every lane has work whatever P is (with one row per workgroup a
    // 320-wide layer kept 80 of 256 threads busy: 1 TB/s, 0.9 ms of a c2 step); one division per thread, then carries
    const int P4 = P / 4;
    const long long first = (long long)blockIdx.x * 256 + threadIdx.x, stride = (long long)gridDim.x * 256;
    const long long total = rows * P4;
    if (first >= total) return;
    long long r = first / P4;
    int c = (int)(first - r * P4);
    const long long dr = stride / P4;
    const int dc = (int)(stride - dr * P4);
    for (long long i = first; i < total; i += stride) {
        const int p = c * 4;
        const float4 v = *reinterpret_cast<const float4 *>(x + r * ldx + p);
        const uint64_t e = (uint64_t)r * P + p;
        float4 o = {v.x * lc_dropout_factor(seed, stream_id, e, keep, inv_keep),
                    v.y * lc_dropout_factor(seed, stream_id, e + 1, keep, inv_keep),
                    v.z * lc_dropout_factor(seed, stream_id, e + 2, keep, inv_keep),
                    v.w * lc_dropout_factor(seed, stream_id, e + 3, keep, inv_keep)};
        float4 *dst = reinterpret_cast<float4 *>(y + r * ldy + p);
        if (accumulate) { const float4 q = *dst; o.x += q.x; o.y += q.y; o.z += q.z; o.w += q.w; }
        *dst = o;
        if constexpr (SHADOW) {
            typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
            typedef float f32x2_t __attribute__((ext_vector_type(2)));
            union { bf16x2_t h; unsigned u; } lo, hi;
            lo.h = __builtin_convertvector((f32x2_t){o.x, o.y}, bf16x2_t);       // v_cvt_pk_bf16_f32: RNE, as lc_cast_bf16
            hi.h = __builtin_convertvector((f32x2_t){o.z, o.w}, bf16x2_t);
            *reinterpret_cast<uint2 *>(y16 + r * ld16 + p) = make_uint2(lo.u, hi.u);
        }
        r += dr;
        c += dc;
        if (c >= P4) { c -= P4; ++r; }
    }
}

// ------------------------------------------------------------------------------ column sums
// Deterministic two-stage reduce: row slab blockIdx.y of a 64-column group -> part[slab][N]; the fold kernel adds
// the slabs in index order (no float atomics: the same call gives the same bits every time).
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float *__restrict__ x, long long rows, int N, int ldx,
                                                             float *__restrict__ part)
{
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int sub = threadIdx.x >> 6;
    float acc = 0.f;
    if (c < N)
        for (long long r = blockIdx.y * 4 + sub; r < rows; r += (long long)gridDim.y * 4) acc += x[r * ldx + c];
    red[sub][threadIdx.x & 63] = acc;
    __syncthreads();
    if (sub == 0 && c < N)
        part[(size_t)blockIdx.y * N + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ __launch_bounds__(256) void colsum_fold_kernel(const float *__restrict__ part, int nslab, int N,
                                                          float *__restrict__ out, int accumulate)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N) return;
    float acc = 0.f;
    for (int k = 0; k < nslab; ++k) acc += part[(size_t)k * N + c];
    out[c] = accumulate ? out[c] + acc : acc;
}
constexpr int COLSUM_SLABS = 128;

// ------------------------------------------------------------------------------ transpose
__global__ __launch_bounds__(256) void transpose_kernel(const float *__restrict__ in, int rows, int cols,
                                                        float *__restrict__ out)
{
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8)
        if (r0 + i < rows && c0 + tx < cols) tile[i][tx] = in[(size_t)(r0 + i) * cols + c0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < cols && r0 + tx < rows) out[(size_t)(c0 + i) * rows + r0 + tx] = tile[tx][i];
}

// ------------------------------------------------------------------------------ MoE combine
// create_moe — nnet/moe.py:29-72: one wave per row r.
//   pi = softmax_E(a[r,:]);  z = tanh(q[r,e*V+v]) (written back over q);  logits[r,v] = sum_e pi_e*mpi_e * tau*z*mz
__global__ __launch_bounds__(256) void moe_fwd_kernel(const float *__restrict__ a, float *__restrict__ q,
                                                      long long R, int E, int V, float tau, float keep, float inv_keep,
                                                      uint32_t seed, float *__restrict__ logits, float *__restrict__ pi)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];   // [4][E] effective gate weights
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *pe = sm + wave * E;
    for (long long r = blockIdx.x * 4ll + wave; r < R; r += (long long)gridDim.x * 4) {
        const float *ar = a + r * E;
        float mx = -INFINITY;
        for (int e = lane; e < E; e += 64) mx = fmaxf(mx, ar[e]);
        mx = lc_wave_max(mx);
        float s = 0.f;
        for (int e = lane; e < E; e += 64) s += expf(ar[e] - mx);
        s = lc_wave_sum(s);
        for (int e = lane; e < E; e += 64) {
            const float p = expf(ar[e] - mx) / s;
            pi[r * E + e] = p;
            pe[e] = p * (keep < 1.f ? lc_dropout_factor(seed, 1000u, (uint64_t)(r * E + e), keep, inv_keep) : 1.f);
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        float *qr = q + r * (long long)E * V;
        for (int v = lane; v < V; v += 64) {
            float acc = 0.f;
            for (int e = 0; e < E; ++e) {
                const long long idx = (long long)e * V + v;
                const float z = lc_tanh(qr[idx]);
                qr[idx] = z;
                const float mz = keep < 1.f ? lc_dropout_factor(seed, 1001u, (uint64_t)(r * (long long)E * V + idx), keep, inv_keep) : 1.f;
                acc += pe[e] * (tau * z * mz);
            }
            logits[r * V + v] = acc;
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
}

// Backward of the combine: zt (tanh values, in q) -> dq in place; da [R,E].
__global__ __launch_bounds__(256) void moe_bwd_kernel(const float *__restrict__ pi, float *__restrict__ q,
                                                      const float *__restrict__ dlogits, long long R, int E, int V,
                                                      float tau, float keep, float inv_keep, uint32_t seed,
                                                      float *__restrict__ da)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];   // [4][E] dpe
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *dpe = sm + wave * E;
    for (long long r = blockIdx.x * 4ll + wave; r < R; r += (long long)gridDim.x * 4) {
        float *qr = q + r * (long long)E * V;
        const float *g = dlogits + r * V;
        const float *pr = pi + r * E;
        for (int e = 0; e < E; ++e) {
            const float mpi = keep < 1.f ? lc_dropout_factor(seed, 1000u, (uint64_t)(r * E + e), keep, inv_keep) : 1.f;
            const float p = pr[e];
            float part = 0.f;
            for (int v = lane; v < V; v += 64) {
                const long long idx = (long long)e * V + v;
                const float mz = keep < 1.f ? lc_dropout_factor(seed, 1001u, (uint64_t)(r * (long long)E * V + idx), keep, inv_keep) : 1.f;
                const float z = qr[idx];
                const float gv = g[v];
                part += gv * tau * z * mz;
                qr[idx] = gv * p * mpi * mz * tau * (1.f - z * z);
            }
            part = lc_wave_sum(part) * mpi;
            if (lane == 0) dpe[e] = part;
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        float dot = 0.f;
        for (int e = lane; e < E; e += 64) dot += dpe[e] * pr[e];
        dot = lc_wave_sum(dot);
        for (int e = lane; e < E; e += 64) da[r * E + e] = pr[e] * (dpe[e] - dot);
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
}

// ------------------------------------------------------------------------------ optimizer
// nnet/graph.py:183-200.  Pass 1: g += l2*theta (first n_decay elements), per-block sum of g^2.
__global__ __launch_bounds__(256) void l2_sumsq_kernel(const float *__restrict__ params, float *__restrict__ grads,
                                                       size_t n, size_t n_decay, float l2, double *__restrict__ partial)
{
    __shared__ double red[4];
    double acc = 0.0;
    if ((n & 3) == 0 && (n_decay & 3) == 0 && ((((uintptr_t)params) | ((uintptr_t)grads)) & 15) == 0) {
        // the flat buffers keep every tensor 16-byte aligned: four elements per thread and access
        const size_t nq = n >> 2, dq = n_decay >> 2;
        for (size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x; q < nq; q += (size_t)gridDim.x * blockDim.x) {
            float4 g = reinterpret_cast<const float4 *>(grads)[q];
            if (q < dq) {
                const float4 w = reinterpret_cast<const float4 *>(params)[q];
                g.x += l2 * w.x; g.y += l2 * w.y; g.z += l2 * w.z; g.w += l2 * w.w;
                reinterpret_cast<float4 *>(grads)[q] = g;
            }
            acc += ((double)g.x * (double)g.x + (double)g.y * (double)g.y) + ((double)g.z * (double)g.z + (double)g.w * (double)g.w);
        }
    } else {
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
            float g = grads[i];
            if (i < n_decay) { g += l2 * params[i]; grads[i] = g; }
            acc += (double)g * (double)g;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// Pass 2 (one block): norm and clip scale.
__global__ __launch_bounds__(256) void norm_finish_kernel(const double *__restrict__ partial, int nblocks,
                                                          float clip_norm, float *__restrict__ norm_out)
{
    __shared__ double red[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) acc += partial[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double norm = sqrt(red[0] + red[1] + red[2] + red[3]);
        norm_out[0] = (float)norm;
        // tf.clip_by_global_norm: g * clip / max(norm, clip); a non-finite norm poisons the update like TF does
        norm_out[1] = (clip_norm > 0.f) ? (float)((double)clip_norm / fmax(norm, (double)clip_norm)) : 1.f;
    }
}
// Pass 3: update.  optimizer 0 sgd, 1 momentum(0.9), 2 adam (TF: theta -= lr_t * m / (sqrt(v) + eps)).
__global__ __launch_bounds__(256) void update_kernel(float *__restrict__ params, const float *__restrict__ grads,
                                                     size_t n, int optimizer, float lr, float lr_t,
                                                     float *__restrict__ state, const float *__restrict__ norm_out,
                                                     const int *__restrict__ guard)
{
    if (guard && *guard != 0) return;          // the step reported a failure: leave parameters and slots untouched
    const float scale = norm_out[1];
    if (optimizer == 2 && (n & 3) == 0 && ((((uintptr_t)params) | ((uintptr_t)grads) | ((uintptr_t)state)) & 15) == 0) {
        // Adam on four elements per thread and access (28 bytes move per parameter: the kernel is a pure stream)
        const size_t nq = n >> 2;
        float4 *P4 = reinterpret_cast<float4 *>(params), *M4 = reinterpret_cast<float4 *>(state), *V4 = reinterpret_cast<float4 *>(state + n);
        const float4 *G4 = reinterpret_cast<const float4 *>(grads);
        for (size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x; q < nq; q += (size_t)gridDim.x * blockDim.x) {
            const float4 g4 = G4[q];
            float4 th = P4[q], m4 = M4[q], v4 = V4[q];
            const float gs[4] = {g4.x * scale, g4.y * scale, g4.z * scale, g4.w * scale};
            float *t = &th.x, *mm = &m4.x, *vv = &v4.x;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float m = 0.9f * mm[k] + 0.1f * gs[k];
                const float v = 0.999f * vv[k] + 0.001f * gs[k] * gs[k];
                mm[k] = m; vv[k] = v;
                t[k] -= lr_t * m / (sqrtf(v) + 1e-8f);
            }
            M4[q] = m4; V4[q] = v4; P4[q] = th;
        }
        return;
    }
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float g = grads[i] * scale;
        float th = params[i];
        if (optimizer == 0) th -= lr * g;
        else if (optimizer == 1) {
            const float acc = 0.9f * state[i] + g;
            state[i] = acc;
            th -= lr * acc;
        } else {
            const float m = 0.9f * state[i] + 0.1f * g;
            const float v = 0.999f * state[n + i] + 0.001f * g * g;
            state[i] = m; state[n + i] = v;
            th -= lr_t * m / (sqrtf(v) + 1e-8f);
        }
        params[i] = th;
    }
}

// ------------------------------------------------------------------------------ posteriors
// nnet/graph.py:236 softmax(smooth*logits); bin/nnet-forward.py:87-91 log, minus class prior.
__global__ __launch_bounds__(256) void posteriors_kernel(const float *__restrict__ logits, long long rows, int V,
                                                         float smooth, int apply_softmax, int apply_log,
                                                         const float *__restrict__ prior, float *__restrict__ out)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (long long r = blockIdx.x * 4ll + wave; r < rows; r += (long long)gridDim.x * 4) {
        const float *x = logits + r * V;
        float *o = out + r * V;
        if (!apply_softmax) {
            for (int k = lane; k < V; k += 64) o[k] = x[k] - (prior ? prior[k] : 0.f);
            continue;
        }
        float mx = -INFINITY;
        for (int k = lane; k < V; k += 64) mx = fmaxf(mx, smooth * x[k]);
        mx = lc_wave_max(mx);
        float s = 0.f;
        for (int k = lane; k < V; k += 64) s += expf(smooth * x[k] - mx);
        s = lc_wave_sum(s);
        for (int k = lane; k < V; k += 64) {
            float y = expf(smooth * x[k] - mx) / s;
            if (apply_log) y = logf(y);      // numpy.log(softmax) as the reference does (-inf on underflow)
            o[k] = y - (prior ? prior[k] : 0.f);
        }
    }
}

// ------------------------------------------------------------------------------ label smoothing
// KL label-smoothing regulariser of nnet/bilstm.py:255-269: loss += w * sum_{rows,k} p*(log p - log q) over
// ALL rows (padded frames included, as the reference sums the whole [B,T,V] tensor); q uniform or a class
// prior given as log q.  dlogits += w * p_k * ((log p_k - log q_k) - KL_row).
__global__ __launch_bounds__(256) void label_smooth_kernel(const float *__restrict__ logits, long long rows, int V,
                                                           const float *__restrict__ logq, float weight,
                                                           double *__restrict__ loss_acc, float *__restrict__ dlogits)
{
    __shared__ double red[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float lu = -logf((float)V);
    double acc = 0.0;
    for (long long r = blockIdx.x * 4ll + wave; r < rows; r += (long long)gridDim.x * 4) {
        const float *x = logits + r * V;
        float mx = -INFINITY;
        for (int k = lane; k < V; k += 64) mx = fmaxf(mx, x[k]);
        mx = lc_wave_max(mx);
        float s = 0.f;
        for (int k = lane; k < V; k += 64) s += expf(x[k] - mx);
        const float lse = mx + logf(lc_wave_sum(s));
        float kl = 0.f;
        for (int k = lane; k < V; k += 64) {
            const float lp = x[k] - lse;
            kl += expf(lp) * (lp - (logq ? logq[k] : lu));
        }
        kl = lc_wave_sum(kl);
        if (dlogits) {
            float *g = dlogits + r * V;
            for (int k = lane; k < V; k += 64) {
                const float lp = x[k] - lse;
                g[k] += weight * expf(lp) * ((lp - (logq ? logq[k] : lu)) - kl);
            }
        }
        acc += (double)kl;
    }
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss_acc, (double)weight * (red[0] + red[1] + red[2] + red[3]));
}

inline int stream_grid(long long work_items, int per_block)
{
    long long g = (work_items + per_block - 1) / per_block;
    return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

}  // namespace

extern "C" int lc_dropout_scale(const float *x, int rows, int P, int ldx, float keep, uint32_t seed,
                                uint32_t stream_id, float *y, int ldy, int accumulate, lc_stream_t stream)
{
    LC_CHECK_ARG(x && y && rows >= 0 && P > 0 && keep > 0.f && keep <= 1.f, "lc_dropout_scale: bad argument");
    if (rows == 0) return LC_OK;
    if (P % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0) {
        const long long quads = (long long)rows * (P / 4);
        const long long nb = (quads + 255) / 256;
        hipLaunchKernelGGL((dropout_scale_vec_kernel<false>), dim3((unsigned)(nb < 16384 ? nb : 16384)), dim3(256), 0, (hipStream_t)stream, x,
                           (long long)rows, P, ldx, keep, 1.0f / keep, seed, stream_id, y, ldy, accumulate, nullptr, 0);
        LC_CHECK_LAUNCH("dropout_scale");
        return LC_OK;
    }
    hipLaunchKernelGGL(dropout_scale_kernel, dim3(stream_grid((long long)rows * P, 256)), dim3(256), 0,
                       (hipStream_t)stream, x, (long long)rows, P, ldx, keep, 1.0f / keep, seed, stream_id, y, ldy,
                       accumulate);
    LC_CHECK_LAUNCH("dropout_scale");
    return LC_OK;
}

extern "C" int lc_dropout_scale_bf16(const float *x, int rows, int P, int ldx, float keep, uint32_t seed,
                                     uint32_t stream_id, float *y, int ldy, int accumulate, uint16_t *y16, int ld16,
                                     lc_stream_t stream)
{
    LC_CHECK_ARG(x && y && y16 && rows >= 0 && P > 0 && keep > 0.f && keep <= 1.f, "lc_dropout_scale_bf16: bad argument");
    LC_CHECK_ARG(P % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ld16 % 4 == 0 && ld16 >= P &&
                     (((uintptr_t)x | (uintptr_t)y) & 15) == 0 && ((uintptr_t)y16 & 7) == 0,
                 "lc_dropout_scale_bf16: P and the row pitches must be multiples of 4, x / y 16-byte and y16 8-byte aligned");
    if (rows == 0) return LC_OK;
    const long long quads = (long long)rows * (P / 4);
    const long long nb = (quads + 255) / 256;
    hipLaunchKernelGGL((dropout_scale_vec_kernel<true>), dim3((unsigned)(nb < 16384 ? nb : 16384)), dim3(256), 0, (hipStream_t)stream, x,
                       (long long)rows, P, ldx, keep, 1.0f / keep, seed, stream_id, y, ldy, accumulate,
                       (unsigned short *)y16, ld16);
    LC_CHECK_LAUNCH("dropout_scale_bf16");
    return LC_OK;
}

extern "C" size_t lc_colsum_workspace_bytes(int N) { return (size_t)COLSUM_SLABS * (N > 0 ? N : 0) * sizeof(float); }

extern "C" int lc_colsum(const float *x, int rows, int N, int ldx, float *out, int accumulate, void *workspace,
                         size_t workspace_bytes, lc_stream_t stream)
{
    LC_CHECK_ARG(x && out && rows >= 0 && N > 0 && ldx >= N, "lc_colsum: bad argument");
    hipStream_t s = (hipStream_t)stream;
    if (rows == 0) {
        if (!accumulate && hipMemsetAsync(out, 0, sizeof(float) * N, s) != hipSuccess) {
            lc_set_error("lc_colsum: memset failed");
            return LC_ELAUNCH;
        }
        return LC_OK;
    }
    int ny = lc_cdiv(rows, 4 * 64);
    if (ny > COLSUM_SLABS) ny = COLSUM_SLABS;
    const bool staged = workspace && workspace_bytes >= (size_t)ny * N * sizeof(float) && ny > 1;
    if (!staged) {          // one slab per column group, written straight to its place by the fold
        LC_CHECK_ARG(workspace && workspace_bytes >= (size_t)N * sizeof(float),
                     "lc_colsum: workspace of at least N floats needed (%zu bytes given)", workspace_bytes);
        ny = 1;
    }
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(lc_cdiv(N, 64), ny), dim3(256), 0, s, x, (long long)rows, N, ldx,
                       (float *)workspace);
    hipLaunchKernelGGL(colsum_fold_kernel, dim3(lc_cdiv(N, 256)), dim3(256), 0, s, (const float *)workspace, ny, N, out,
                       accumulate);
    LC_CHECK_LAUNCH("colsum");
    return LC_OK;
}

extern "C" int lc_transpose(const float *in, int rows, int cols, float *out, lc_stream_t stream)
{
    LC_CHECK_ARG(in && out && rows > 0 && cols > 0, "lc_transpose: bad argument");
    hipLaunchKernelGGL(transpose_kernel, dim3(lc_cdiv(cols, 32), lc_cdiv(rows, 32)), dim3(256), 0,
                       (hipStream_t)stream, in, rows, cols, out);
    LC_CHECK_LAUNCH("transpose");
    return LC_OK;
}

extern "C" int lc_moe_combine_fwd(const float *a, float *q, int R, int E, int V, float tau, float keep,
                                  uint32_t seed, float *logits, float *pi, lc_stream_t stream)
{
    LC_CHECK_ARG(a && q && logits && pi && R >= 0 && E > 0 && V > 0 && keep > 0.f && keep <= 1.f,
                 "lc_moe_combine_fwd: bad argument");
    if (R == 0) return LC_OK;
    hipLaunchKernelGGL(moe_fwd_kernel, dim3(stream_grid(R, 4)), dim3(256),
                       4 * E * sizeof(float), (hipStream_t)stream, a, q, (long long)R, E, V, tau, keep, 1.0f / keep,
                       seed, logits, pi);
    LC_CHECK_LAUNCH("moe_fwd");
    return LC_OK;
}

extern "C" int lc_moe_combine_bwd(const float *pi, float *q, const float *dlogits, int R, int E, int V, float tau,
                                  float keep, uint32_t seed, float *da, lc_stream_t stream)
{
    LC_CHECK_ARG(pi && q && dlogits && da && R >= 0 && E > 0 && V > 0 && keep > 0.f && keep <= 1.f,
                 "lc_moe_combine_bwd: bad argument");
    if (R == 0) return LC_OK;
    hipLaunchKernelGGL(moe_bwd_kernel, dim3(stream_grid(R, 4)), dim3(256), 4 * E * sizeof(float), (hipStream_t)stream,
                       pi, q, dlogits, (long long)R, E, V, tau, keep, 1.0f / keep, seed, da);
    LC_CHECK_LAUNCH("moe_bwd");
    return LC_OK;
}

static const int kOptBlocks = 1024;
extern "C" size_t lc_optimizer_workspace_bytes(size_t n)
{
    (void)n;
    return kOptBlocks * sizeof(double);
}

extern "C" int lc_optimizer_step(float *params, float *grads, size_t n, size_t n_decay, float l2, float clip_norm,
                                 int optimizer, float lr, int step, float *state, float *norm_out, const int *guard,
                                 void *workspace, size_t workspace_bytes, lc_stream_t stream)
{
    LC_CHECK_ARG(params && grads && norm_out && workspace, "lc_optimizer_step: null pointer");
    LC_CHECK_ARG(optimizer >= 0 && optimizer <= 2 && (optimizer == 0 || state), "lc_optimizer_step: bad optimizer/state");
    LC_CHECK_ARG(n_decay <= n && step >= 1, "lc_optimizer_step: bad n_decay/step");
    if (workspace_bytes < lc_optimizer_workspace_bytes(n)) {
        lc_set_error("lc_optimizer_step: workspace too small");
        return LC_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    size_t nb = (n + 256 * 8 - 1) / (256 * 8);
    if (nb < 1) nb = 1;
    if (nb > (size_t)kOptBlocks) nb = kOptBlocks;
    hipLaunchKernelGGL(l2_sumsq_kernel, dim3((unsigned)nb), dim3(256), 0, s, params, grads, n, n_decay, l2,
                       (double *)workspace);
    hipLaunchKernelGGL(norm_finish_kernel, dim3(1), dim3(256), 0, s, (const double *)workspace, (int)nb, clip_norm,
                       norm_out);
    // Adam bias correction as tf.train.AdamOptimizer: lr_t = lr * sqrt(1-b2^t) / (1-b1^t)
    const double lr_t = (double)lr * sqrt(1.0 - pow(0.999, (double)step)) / (1.0 - pow(0.9, (double)step));
    hipLaunchKernelGGL(update_kernel, dim3((unsigned)nb), dim3(256), 0, s, params, grads, n, optimizer, lr, (float)lr_t,
                       state, norm_out, guard);
    LC_CHECK_LAUNCH("optimizer_step");
    return LC_OK;
}

extern "C" int lc_label_smoothing(const float *logits, int rows, int V, const float *log_q, float weight,
                                  double *loss_acc, float *dlogits, lc_stream_t stream)
{
    LC_CHECK_ARG(logits && loss_acc && rows >= 0 && V > 0, "lc_label_smoothing: bad argument");
    if (rows == 0) return LC_OK;
    hipLaunchKernelGGL(label_smooth_kernel, dim3(stream_grid(rows, 4)), dim3(256), 0, (hipStream_t)stream, logits,
                       (long long)rows, V, log_q, weight, loss_acc, dlogits);
    LC_CHECK_LAUNCH("label_smoothing");
    return LC_OK;
}

extern "C" int lc_posteriors(const float *logits, int rows, int V, float smooth, int apply_softmax, int apply_log,
                             const float *log_prior, float *out, lc_stream_t stream)
{
    LC_CHECK_ARG(logits && out && rows >= 0 && V > 0, "lc_posteriors: bad argument");
    if (rows == 0) return LC_OK;
    hipLaunchKernelGGL(posteriors_kernel, dim3(stream_grid(rows, 4)), dim3(256), 0, (hipStream_t)stream, logits,
                       (long long)rows, V, smooth, apply_softmax, apply_log, log_prior, out);
    LC_CHECK_LAUNCH("posteriors");
    return LC_OK;
}

// ---- development hook: a foreign resident kernel (tests/test_gpu_coresidency.py) ---------------------------------------------
// `blocks` workgroups of 256 threads that do nothing but stay resident for `microseconds` (wall clock: s_memrealtime counts
// the 100 MHz constant clock), each holding `lds_bytes` of LDS.  What a collective's kernel that waits for a slower peer looks
// like to the dispatcher: CUs that the next persistent recurrence - one workgroup per CU on every CU of an XCD, 84+ KB of the
// CU's 160 KB of LDS each - cannot have until it leaves (with lds_bytes >= 80 KB; an idle kernel WITHOUT a resource footprint
// slips in beside a persistent workgroup and rehearses nothing).
__global__ __launch_bounds__(256) void debug_spin_kernel(unsigned long long ticks)
{
    extern __shared__ float spin_lds[];
    if (ticks == ~0ull) spin_lds[threadIdx.x] = 0.f;              // (keeps the allocation referenced)
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
extern "C" int lc_debug_spin(int blocks, int microseconds, int lds_bytes, lc_stream_t stream)
{
    LC_CHECK_ARG(blocks > 0 && blocks <= 4096 && microseconds >= 0 && microseconds <= 5000000 && lds_bytes >= 0 &&
                     lds_bytes <= 160 * 1024,
                 "lc_debug_spin: bad argument");
    if (hipFuncSetAttribute((const void *)debug_spin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) {
        (void)hipGetLastError();
        lc_set_error("lc_debug_spin: hipFuncSetAttribute(MaxDynamicSharedMemorySize, %d) failed", lds_bytes);
        return LC_ELAUNCH;
    }
    hipLaunchKernelGGL(debug_spin_kernel, dim3(blocks), dim3(256), (size_t)lds_bytes, (hipStream_t)stream,
                       (unsigned long long)microseconds * 100ull);
    LC_CHECK_LAUNCH("debug_spin");
    return LC_OK;
}
