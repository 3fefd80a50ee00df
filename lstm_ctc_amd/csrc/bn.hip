// bn.hip — tf.layers.batch_normalization as create_logits_lstm applies it (mobvoi/lstm_ctc nnet/lstm.py:271-294):
// rank-3 input => TF's non-fused path, moments over EVERY [B,T] position (padded frames included), population
// variance, epsilon 1e-3, y = (x - mean) * rsqrt(var + eps) * gamma + beta; moving averages (momentum 0.99) are
// updated by the UPDATE_OPS the train op depends on (nnet/graph.py:194-196) and replace the batch moments when
// is_training is false.
//
// All of it is HBM-bound streaming over a [rows, C] row-major activation matrix (rows = T*B): two passes forward
// (column moments, then normalise) and two backward (column sums of dy and dy*xhat, then dx).  Column sums are
// accumulated in double (per-thread float partials over <= rows/512 rows, one double partial per row slab, slabs
// folded in index order - deterministic, no atomics) so that var = E[x^2] - mean^2 keeps float32 accuracy for
// activations with |mean| >> std.
#include "common.h"

namespace {

// part[slab][0][c] = sum_r a(r,c), part[slab][1][c] = sum_r b(r,c) over the rows of slab blockIdx.y;
// grid (ceil(C/64), ny), 256 threads = 4 row phases x 64 cols
template <class F>
__device__ __forceinline__ void column_sums2(long long rows, int C, double *__restrict__ acc, F &&value)
{
    __shared__ float red[2][4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int sub = threadIdx.x >> 6;
    float s0 = 0.f, s1 = 0.f;
    if (c < C)
        for (long long r = blockIdx.y * 4 + sub; r < rows; r += (long long)gridDim.y * 4) {
            float a, b;
            value(r, c, a, b);
            s0 += a;
            s1 += b;
        }
    red[0][sub][threadIdx.x & 63] = s0;
    red[1][sub][threadIdx.x & 63] = s1;
    __syncthreads();
    if (sub == 0 && c < C) {
        const int l = threadIdx.x;
        double *part = acc + (size_t)(1 + blockIdx.y) * 2 * C;
        part[c] = (double)red[0][0][l] + red[0][1][l] + red[0][2][l] + red[0][3][l];
        part[C + c] = (double)red[1][0][l] + red[1][1][l] + red[1][2][l] + red[1][3][l];
    }
}

__global__ __launch_bounds__(256) void bn_moments_partial_kernel(const float *__restrict__ x, long long rows, int C,
                                                                 int ldx, double *__restrict__ acc)
{
    column_sums2(rows, C, acc, [&](long long r, int c, float &a, float &b) {
        const float v = x[r * ldx + c];
        a = v;
        b = v * v;
    });
}

// acc[0 .. 2C) = sum over the slabs, in index order
__global__ __launch_bounds__(256) void bn_fold_kernel(double *__restrict__ acc, int nslab, int C)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * C) return;
    double a = 0.0;
    for (int k = 0; k < nslab; ++k) a += acc[(size_t)(1 + k) * 2 * C + i];
    acc[i] = a;
}

__global__ __launch_bounds__(256) void bn_moments_finish_kernel(const double *__restrict__ acc, long long rows, int C,
                                                                float *__restrict__ mean, float *__restrict__ var)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double m = acc[c] / (double)rows;
    const double v = acc[C + c] / (double)rows - m * m;
    mean[c] = (float)m;
    var[c] = (float)(v > 0.0 ? v : 0.0);
}

// y = (x - mean) * rsqrt(var + eps) * gamma + beta
__global__ __launch_bounds__(256) void bn_apply_kernel(const float *__restrict__ x, long long rows, int C, int ldx,
                                                       const float *__restrict__ mean, const float *__restrict__ var,
                                                       const float *__restrict__ gamma, const float *__restrict__ beta,
                                                       float eps, float *__restrict__ y, int ldy)
{
    const long long total = rows * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / C;
        const int c = (int)(i % C);
        const float inv = rsqrtf(var[c] + eps) * gamma[c];
        y[r * ldy + c] = (x[r * ldx + c] - mean[c]) * inv + beta[c];
    }
}

__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float *__restrict__ x, const float *__restrict__ dy,
                                                             long long rows, int C, int ldx, int lddy,
                                                             const float *__restrict__ mean,
                                                             const float *__restrict__ var, float eps,
                                                             double *__restrict__ acc)
{
    column_sums2(rows, C, acc, [&](long long r, int c, float &a, float &b) {
        const float g = dy[r * lddy + c];
        const float xhat = (x[r * ldx + c] - mean[c]) * rsqrtf(var[c] + eps);
        a = g;
        b = g * xhat;
    });
}

// dx = gamma * inv * (dy - mean_r(dy) - xhat * mean_r(dy*xhat))   (training)
// dx = gamma * inv * dy                                           (inference statistics)
// dgamma = sum dy*xhat, dbeta = sum dy (written by the threads of row 0)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float *__restrict__ x, const float *__restrict__ dy,
                                                           long long rows, int C, int ldx, int lddy,
                                                           const float *__restrict__ mean,
                                                           const float *__restrict__ var,
                                                           const float *__restrict__ gamma, float eps, int training,
                                                           const double *__restrict__ acc, float *__restrict__ dx,
                                                           int lddx, float *__restrict__ dgamma,
                                                           float *__restrict__ dbeta)
{
    const long long total = rows * C;
    const double invr = 1.0 / (double)rows;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / C;
        const int c = (int)(i % C);
        const float inv = rsqrtf(var[c] + eps);
        const float xhat = (x[r * ldx + c] - mean[c]) * inv;
        const float g = dy[r * lddy + c];
        float d = g;
        if (training) d = g - (float)(acc[c] * invr) - xhat * (float)(acc[C + c] * invr);
        dx[r * lddx + c] = gamma[c] * inv * d;
        if (r == 0) {
            dbeta[c] = (float)acc[c];
            dgamma[c] = (float)acc[C + c];
        }
    }
}

// assign_moving_average: v -= (v - value) * (1 - momentum)
__global__ __launch_bounds__(256) void bn_moving_kernel(float *__restrict__ mm, float *__restrict__ mv,
                                                        const float *__restrict__ mean, const float *__restrict__ var,
                                                        int C, float one_minus_momentum)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    mm[c] -= (mm[c] - mean[c]) * one_minus_momentum;
    mv[c] -= (mv[c] - var[c]) * one_minus_momentum;
}

// dynamic_rnn's length mask on a time-major [T*B, C] matrix: rows with t >= seq_len[b] become zero
__global__ __launch_bounds__(256) void length_mask_kernel(float *__restrict__ x, long long rows, int B, int C, int ldx,
                                                          const int *__restrict__ seq_len)
{
    const long long total = rows * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / C;
        const int c = (int)(i % C);
        if ((int)(r / B) >= seq_len[r % B]) x[r * ldx + c] = 0.f;
    }
}

inline int stream_blocks(long long total)
{
    long long b = (total + 255) / 256;
    return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}
inline int reduce_rows_grid(long long rows)
{
    int ny = lc_cdiv(rows, 4 * 64);
    return ny > 128 ? 128 : (ny < 1 ? 1 : ny);
}

}  // namespace

extern "C" size_t lc_bn_workspace_bytes(int C) { return (size_t)(1 + 128) * 2 * (C > 0 ? C : 0) * sizeof(double); }

extern "C" int lc_bn_moments(const float *x, int rows, int C, int ldx, float *mean, float *var, void *workspace,
                             size_t workspace_bytes, lc_stream_t stream)
{
    LC_CHECK_ARG(x && mean && var && rows > 0 && C > 0 && ldx >= C, "lc_bn_moments: bad argument");
    if (!workspace || workspace_bytes < lc_bn_workspace_bytes(C)) {
        lc_set_error("lc_bn_moments: workspace too small");
        return LC_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    const int ny = reduce_rows_grid(rows);
    hipLaunchKernelGGL(bn_moments_partial_kernel, dim3(lc_cdiv(C, 64), ny), dim3(256), 0, s, x,
                       (long long)rows, C, ldx, (double *)workspace);
    hipLaunchKernelGGL(bn_fold_kernel, dim3(lc_cdiv(2 * C, 256)), dim3(256), 0, s, (double *)workspace, ny, C);
    hipLaunchKernelGGL(bn_moments_finish_kernel, dim3(lc_cdiv(C, 256)), dim3(256), 0, s, (const double *)workspace,
                       (long long)rows, C, mean, var);
    LC_CHECK_LAUNCH("bn_moments");
    return LC_OK;
}

extern "C" int lc_bn_apply(const float *x, int rows, int C, int ldx, const float *mean, const float *var,
                           const float *gamma, const float *beta, float eps, float *y, int ldy, lc_stream_t stream)
{
    LC_CHECK_ARG(x && mean && var && gamma && beta && y && rows >= 0 && C > 0 && ldx >= C && ldy >= C && eps > 0.f,
                 "lc_bn_apply: bad argument");
    if (rows == 0) return LC_OK;
    hipLaunchKernelGGL(bn_apply_kernel, dim3(stream_blocks((long long)rows * C)), dim3(256), 0, (hipStream_t)stream, x,
                       (long long)rows, C, ldx, mean, var, gamma, beta, eps, y, ldy);
    LC_CHECK_LAUNCH("bn_apply");
    return LC_OK;
}

extern "C" int lc_bn_bwd(const float *x, const float *dy, int rows, int C, int ldx, int lddy, const float *mean,
                         const float *var, const float *gamma, float eps, int training, float *dx, int lddx,
                         float *dgamma, float *dbeta, void *workspace, size_t workspace_bytes, lc_stream_t stream)
{
    LC_CHECK_ARG(x && dy && mean && var && gamma && dx && dgamma && dbeta && rows > 0 && C > 0 && ldx >= C &&
                     lddy >= C && lddx >= C && eps > 0.f,
                 "lc_bn_bwd: bad argument");
    if (!workspace || workspace_bytes < lc_bn_workspace_bytes(C)) {
        lc_set_error("lc_bn_bwd: workspace too small");
        return LC_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    const int ny = reduce_rows_grid(rows);
    hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(lc_cdiv(C, 64), ny), dim3(256), 0, s, x, dy,
                       (long long)rows, C, ldx, lddy, mean, var, eps, (double *)workspace);
    hipLaunchKernelGGL(bn_fold_kernel, dim3(lc_cdiv(2 * C, 256)), dim3(256), 0, s, (double *)workspace, ny, C);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(stream_blocks((long long)rows * C)), dim3(256), 0, s, x, dy,
                       (long long)rows, C, ldx, lddy, mean, var, gamma, eps, training, (const double *)workspace, dx,
                       lddx, dgamma, dbeta);
    LC_CHECK_LAUNCH("bn_bwd");
    return LC_OK;
}

extern "C" int lc_bn_update_moving(float *moving_mean, float *moving_var, const float *mean, const float *var, int C,
                                   float momentum, lc_stream_t stream)
{
    LC_CHECK_ARG(moving_mean && moving_var && mean && var && C > 0 && momentum >= 0.f && momentum <= 1.f,
                 "lc_bn_update_moving: bad argument");
    hipLaunchKernelGGL(bn_moving_kernel, dim3(lc_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, moving_mean,
                       moving_var, mean, var, C, 1.0f - momentum);
    LC_CHECK_LAUNCH("bn_update_moving");
    return LC_OK;
}

extern "C" int lc_length_mask(float *x, int T, int B, int C, int ldx, const int *seq_len, lc_stream_t stream)
{
    LC_CHECK_ARG(x && seq_len && T >= 0 && B > 0 && C > 0 && ldx >= C, "lc_length_mask: bad argument");
    if (T == 0) return LC_OK;
    const long long rows = (long long)T * B;
    hipLaunchKernelGGL(length_mask_kernel, dim3(stream_blocks(rows * C)), dim3(256), 0, (hipStream_t)stream, x, rows, B, C,
                       ldx, seq_len);
    LC_CHECK_LAUNCH("length_mask");
    return LC_OK;
}
