// gemm.hip — float32 GEMM on the gfx950 f32 MFMA pipe (v_mfma_f32_32x32x2_f32, exact f32).
//
// C[M,N] = alpha * op(A)[M,K] * op(B)[K,N] + beta * C (+ bias[N]).  Row-major storage.
// Replaces tf.matmul / tf.nn.xw_plus_b (mobvoi/lstm_ctc nnet/bilstm.py:249, nnet/moe.py:43,58) and
// the batched (hoisted-over-T) halves of the LSTMCell kernel matmul (bilstm.py:129-136) and of
// their gradients (tf.gradients, nnet/graph.py:190).
//
// Tiling: 128x128x16 block tile, 256 threads = 4 waves in 2x2, each wave owns a 64x64 patch as
// 2x2 MFMA 32x32 tiles (64 accumulator VGPRs).  Both operand tiles are kept K-MAJOR in LDS
// (As[k][m], Bs[k][n]) so the one-float-per-lane MFMA fragments (lane -> row l&31, k = l>>5) are
// conflict-free ds_read_b32; a k-minor source (NN's A, NT's B) is transposed on the LDS store.
// Register-staged global prefetch of tile k+1 overlaps the 32 MFMAs of tile k; two LDS buffers,
// one barrier per K step.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BM = 128, BN = 128, BK = 16, NT = 256;

// Loads one 128x16 operand tile into 2 float4 registers per thread.
// KMAJOR: the stored matrix has k along rows (element (k, c) at src[k*ld + c]).
template <bool KMAJOR>
__device__ __forceinline__ void tile_load(const float *__restrict__ src, int ld, int c0, int cmax, int k0,
                                          int kmax, bool vec_ok, float4 (&r)[2])
{
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int f = threadIdx.x + NT * i;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (KMAJOR) {
            const int k = k0 + f / 32, c = c0 + 4 * (f % 32);
            if (k < kmax) {
                const float *p = src + (size_t)k * ld + c;
                if (vec_ok && c + 3 < cmax) v = *reinterpret_cast<const float4 *>(p);
                else {
                    if (c < cmax) v.x = p[0];
                    if (c + 1 < cmax) v.y = p[1];
                    if (c + 2 < cmax) v.z = p[2];
                    if (c + 3 < cmax) v.w = p[3];
                }
            }
        } else {
            const int c = c0 + f % 128, k = k0 + 4 * (f / 128);
            if (c < cmax) {
                const float *p = src + (size_t)c * ld + k;
                if (vec_ok && k + 3 < kmax) v = *reinterpret_cast<const float4 *>(p);
                else {
                    if (k < kmax) v.x = p[0];
                    if (k + 1 < kmax) v.y = p[1];
                    if (k + 2 < kmax) v.z = p[2];
                    if (k + 3 < kmax) v.w = p[3];
                }
            }
        }
        r[i] = v;
    }
}

template <bool KMAJOR>
__device__ __forceinline__ void tile_store(float *__restrict__ lds /*[BK][128]*/, const float4 (&r)[2])
{
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int f = threadIdx.x + NT * i;
        if (KMAJOR) {
            *reinterpret_cast<float4 *>(lds + (f / 32) * 128 + 4 * (f % 32)) = r[i];
        } else {
            const int c = f % 128, k = 4 * (f / 128);
            lds[(k + 0) * 128 + c] = r[i].x;
            lds[(k + 1) * 128 + c] = r[i].y;
            lds[(k + 2) * 128 + c] = r[i].z;
            lds[(k + 3) * 128 + c] = r[i].w;
        }
    }
}

template <bool TA, bool TB>
__global__ __launch_bounds__(NT) void gemm_f32_kernel(int M, int N, int K, float alpha,
                                                      const float *__restrict__ A, int lda,
                                                      const float *__restrict__ B, int ldb, float beta,
                                                      float *__restrict__ C, int ldc,
                                                      const float *__restrict__ bias, int vecA, int vecB)
{
    __shared__ __attribute__((aligned(16))) float As[2][BK * BM];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK * BN];
    // XCD-aware remap: consecutive tile ids (sharing a B column panel / A row panel) stay on one XCD's L2
    const int nbm = (M + BM - 1) / BM, nbn = (N + BN - 1) / BN;
    const int nwg = nbm * nbn;
    int bid = blockIdx.x;
    {
        const int q = nwg / 8, rr = nwg % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
    }
    // grouped order: the ~64 tiles in flight on one XCD form an 8(m) x 8(n) patch, so each A row
    // panel and each B column panel is shared 8 ways out of that XCD's L2
    constexpr int GROUP_M = 8;
    const int gsz = GROUP_M * nbn;
    const int first_m = (bid / gsz) * GROUP_M;
    const int gm = min(nbm - first_m, GROUP_M);
    const int bm = first_m + (bid % gsz) % gm, bn = (bid % gsz) / gm;
    const int m0 = bm * BM, n0 = bn * BN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // A operand: TA -> stored [K,M] (k-major); else stored [M,K] (k-minor)
    // B operand: TB -> stored [N,K] (k-minor); else stored [K,N] (k-major)
    float4 ra[2], rb[2];
    const int nk = (K + BK - 1) / BK;
    tile_load<TA>(A, lda, m0, M, 0, K, vecA, ra);
    tile_load<!TB>(B, ldb, n0, N, 0, K, vecB, rb);
    tile_store<TA>(As[0], ra);
    tile_store<!TB>(Bs[0], rb);
    __syncthreads();
    const int lr = lane & 31, lk = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            tile_load<TA>(A, lda, m0, M, (kt + 1) * BK, K, vecA, ra);
            tile_load<!TB>(B, ldb, n0, N, (kt + 1) * BK, K, vecB, rb);
        }
        const float *as = As[cur] + wm * 64 + lr;
        const float *bs = Bs[cur] + wn * 64 + lr;
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const int krow = (2 * kk + lk) * 128;
            const float a0 = as[krow], a1 = as[krow + 32];
            const float b0 = bs[krow], b1 = bs[krow + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            tile_store<TA>(As[cur ^ 1], ra);
            tile_store<!TB>(Bs[cur ^ 1], rb);
        }
        __syncthreads();
    }
    // epilogue: lane holds C[row = (r&3) + 8*(r>>2) + 4*(lane>>5)][col = lane&31] of each 32x32 tile
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + lr;
            if (col >= N) continue;
            const float bv = bias ? bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (row < M) {
                    float *c = C + (size_t)row * ldc + col;
                    float v = alpha * acc[i][j][r] + bv;
                    if (beta != 0.f) v += beta * *c;
                    *c = v;
                }
            }
        }
}

inline bool aligned16(const void *p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace

extern "C" int lc_gemm_f32(int ta, int tb, int M, int N, int K, float alpha, const float *A, int lda,
                           const float *B, int ldb, float beta, float *C, int ldc, const float *bias,
                           lc_stream_t stream)
{
    LC_CHECK_ARG(A && B && C, "lc_gemm_f32: null pointer");
    LC_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, "lc_gemm_f32: negative dimension");
    if (M == 0 || N == 0) return LC_OK;
    LC_CHECK_ARG(lda >= (ta ? M : K) && ldb >= (tb ? K : N) && ldc >= N, "lc_gemm_f32: leading dimension too small");
    hipStream_t s = (hipStream_t)stream;
    const int vecA = aligned16(A) && (lda % 4 == 0);
    const int vecB = aligned16(B) && (ldb % 4 == 0);
    const long long nwg = (long long)lc_cdiv(M, BM) * lc_cdiv(N, BN);
    LC_CHECK_ARG(nwg < (1ll << 31), "lc_gemm_f32: grid too large");
    dim3 grid((unsigned)nwg), block(NT);
#define LC_GEMM(TA, TB)                                                                                        \
    hipLaunchKernelGGL((gemm_f32_kernel<TA, TB>), grid, block, 0, s, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, \
                       bias, vecA, vecB)
    if (!ta && !tb) LC_GEMM(false, false);
    else if (ta && !tb) LC_GEMM(true, false);
    else if (!ta && tb) LC_GEMM(false, true);
    else LC_GEMM(true, true);
#undef LC_GEMM
    LC_CHECK_LAUNCH("gemm_f32");
    return LC_OK;
}
