// gemm_skinny.hip - the affine head's products: a tall activation matrix against a NARROW weight (num_targets = 44 / 72
// columns; mobvoi/lstm_ctc nnet/bilstm.py:249 `tf.nn.xw_plus_b`, nnet/lstm.py:105, and its weight gradient in
// tf.gradients, nnet/graph.py:190).  The general kernels pad 44 columns to a 128- or 256-wide tile and spend three to
// six times the matrix work (0.40 ms for 64000 x 44 x 2048 where the operand stream alone is 0.10 ms of HBM time); here the
// column tile is 16 NCT <= 80 wide (44 -> 48, 72 -> 80):
//
//   NN:  C[M, N] = alpha A[M, K] B[K, N] + beta C + bias     A streamed once from HBM, row-major, 16 bytes per lane and
//        k16-block straight into the MFMA's A operand (lane (row, lk) holds k = 4 lk .. 4 lk + 3: v_mfma_f32_16x16x4_f32 q
//        takes element q); B - the weight, L2-resident - staged per 128-deep K chunk in LDS in FRAGMENT order (one
//        ds_read_b128 per k16-block and column tile), two buffers; a wave owns 32 rows (two row tiles), a workgroup 128.
//   TN:  C[M, N] = alpha A^T B + ...   A stored [K, M] (the activations, K = T * B rows), B stored [K, N] (dlogits): a workgroup
//        takes a 128-column block of A and a slab of K; both chunks go through LDS in fragment order (A transposed on the
//        way: a k-row's 128 consecutive columns are loaded as float4 and scattered to the lanes that own them); the slabs'
//        partial products are reduced in a second pass in a fixed order (deterministic).
//
// ROUND = the c5 semantics (`compute_dtype = bf16`): both operands rounded to bf16 (nearest even) on load; the products of
// two bf16 values are exact in fp32, so the fp32 MFMA on the rounded values IS the bf16 product with fp32 accumulation.
#include "common.h"
#include "gemm_epi.h"

typedef float f32x4s __attribute__((ext_vector_type(4)));

namespace {

constexpr int SK_ROWS = 128, SK_KC = 128, SK_NT = 256;

struct SkinnyArgs {
    int M, N, K;
    float alpha, beta;
    const float *A; int lda;
    const float *B; int ldb;
    float *C; int ldc;
    const float *bias;
    int kchunk;            // TN: reduction rows per blockIdx.y slab
    float *slab;           // TN: [gridDim.y][M][N] partial products (nullptr: one slab, written straight to C)
};

template <bool ROUND>
__device__ __forceinline__ float sk_val(float x)
{
    if constexpr (ROUND) return (float)(__bf16)x;
    return x;
}

// B chunk [kc rows of K][N] -> LDS in fragment order: Bl[kb][c][lane = lk * 16 + li][q] = B[k0 + 16 kb + 4 lk + q][16 c + li]
template <int NCT, bool ROUND>
__device__ __forceinline__ void sk_stage_b(const SkinnyArgs &p, int k0, int kc, float *Bl)
{
    constexpr int W = 16 * NCT;
    for (int idx = threadIdx.x; idx < SK_KC * W; idx += SK_NT) {
        const int k = idx / W, n = idx - k * W;
        const float v = (k < kc && n < p.N) ? sk_val<ROUND>(p.B[(size_t)(k0 + k) * p.ldb + n]) : 0.f;
        const int kb = k >> 4, lk = (k >> 2) & 3, q = k & 3, c = n >> 4, li = n & 15;
        Bl[(((kb * NCT + c) * 64) + lk * 16 + li) * 4 + q] = v;
    }
}

template <int NCT, bool ROUND>
__global__ __launch_bounds__(SK_NT) void gemm_skinny_nn_kernel(SkinnyArgs p)
{
    __shared__ __attribute__((aligned(16))) float Bl[2][(SK_KC / 16) * NCT * 64 * 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int m0 = blockIdx.x * SK_ROWS + wave * 32;
    const float *arow[2];
    bool rok[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const int r = m0 + rt * 16 + li;
        rok[rt] = r < p.M;
        arow[rt] = p.A + (size_t)min(r, p.M - 1) * p.lda + 4 * lk;
    }
    f32x4s acc[2][NCT];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int c = 0; c < NCT; ++c) acc[rt][c] = (f32x4s){0.f, 0.f, 0.f, 0.f};
    const int nchunk = (p.K + SK_KC - 1) / SK_KC;
    // A of a chunk: 8 k16-blocks x 2 row tiles of 16 bytes per lane, requested a whole chunk ahead
    f32x4s a[2][SK_KC / 16][2];
    auto load_a = [&](int ch, f32x4s (&dst)[SK_KC / 16][2]) {
        const int k0 = ch * SK_KC;
#pragma unroll
        for (int kb = 0; kb < SK_KC / 16; ++kb)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const int k = k0 + kb * 16 + 4 * lk;
                // (K is a multiple of 4 and rows are 16-byte aligned: the host checks; a block past K reads zeros)
                dst[kb][rt] = (k < p.K) ? *reinterpret_cast<const f32x4s *>(arow[rt] + k0 + kb * 16) : (f32x4s){0.f, 0.f, 0.f, 0.f};
            }
    };
    load_a(0, a[0]);
    sk_stage_b<NCT, ROUND>(p, 0, min(SK_KC, p.K), Bl[0]);
    __syncthreads();
#pragma unroll 1
    for (int ch = 0; ch < nchunk; ch += 2) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int cc = ch + half;
            if (cc >= nchunk) break;
            if (cc + 1 < nchunk) {
                load_a(cc + 1, a[half ^ 1]);
                sk_stage_b<NCT, ROUND>(p, (cc + 1) * SK_KC, min(SK_KC, p.K - (cc + 1) * SK_KC), Bl[half ^ 1]);
            }
            const float *bl = Bl[half];
#pragma unroll
            for (int kb = 0; kb < SK_KC / 16; ++kb) {
                f32x4s b[NCT];
#pragma unroll
                for (int c = 0; c < NCT; ++c) b[c] = *reinterpret_cast<const f32x4s *>(bl + ((kb * NCT + c) * 64 + lane) * 4);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    f32x4s av = a[half][kb][rt];
                    if constexpr (ROUND) av = (f32x4s){sk_val<true>(av.x), sk_val<true>(av.y), sk_val<true>(av.z), sk_val<true>(av.w)};
#pragma unroll
                    for (int c = 0; c < NCT; ++c) {
                        acc[rt][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, b[c].x, acc[rt][c], 0, 0, 0);
                        acc[rt][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, b[c].y, acc[rt][c], 0, 0, 0);
                        acc[rt][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, b[c].z, acc[rt][c], 0, 0, 0);
                        acc[rt][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, b[c].w, acc[rt][c], 0, 0, 0);
                    }
                }
            }
            __syncthreads();                       // the other buffer is staged, this one may be overwritten
        }
    }
    (void)rok;
    // C layout of a 16 x 16 tile: column = lane & 15, row = 4 (lane >> 4) + r
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            const int col = 16 * c + li;
            if (col >= p.N) continue;
            const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + rt * 16 + 4 * lk + r;
                if (row < p.M) {
                    float *cp = p.C + (size_t)row * p.ldc + col;
                    float v = p.alpha * acc[rt][c][r] + bv;
                    if (p.beta != 0.f) v += p.beta * *cp;
                    *cp = v;
                }
            }
        }
}

// TN: A stored [K, M], B stored [K, N]; blockIdx.x = 128-column block of A (= 128 rows of C), blockIdx.y = K slab.
template <int NCT, bool ROUND>
__global__ __launch_bounds__(SK_NT) void gemm_skinny_tn_kernel(SkinnyArgs p)
{
    constexpr int KC = 64;                                   // reduction rows per staged chunk
    __shared__ __attribute__((aligned(16))) float Al[2][(KC / 16) * 8 * 64 * 4];      // [kb][row tile of the workgroup][lane][q]
    __shared__ __attribute__((aligned(16))) float Bl[2][(KC / 16) * NCT * 64 * 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int m0 = blockIdx.x * SK_ROWS;
    const int kbeg = blockIdx.y * p.kchunk, kend = min(p.K, kbeg + p.kchunk);
    f32x4s acc[2][NCT];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int c = 0; c < NCT; ++c) acc[rt][c] = (f32x4s){0.f, 0.f, 0.f, 0.f};
    // staging: a chunk of A is KC k-rows x 128 columns = KC * 32 float4, 8 per thread at KC = 64: thread -> (k-row, column quad)
    auto stage = [&](int k0, int buf) {
#pragma unroll
        for (int it = 0; it < KC * 32 / SK_NT; ++it) {
            const int idx = it * SK_NT + threadIdx.x, k = idx >> 5, mq = idx & 31;       // columns 4 mq .. 4 mq + 3
            const int kk = k0 + k, m = m0 + 4 * mq;
            f32x4s v = {0.f, 0.f, 0.f, 0.f};
            if (kk < kend) {
                const float *src = p.A + (size_t)kk * p.lda + m;
                if (m + 3 < p.M) v = *reinterpret_cast<const f32x4s *>(src);
                else {
                    if (m < p.M) v.x = src[0];
                    if (m + 1 < p.M) v.y = src[1];
                    if (m + 2 < p.M) v.z = src[2];
                }
            }
            const int kb = k >> 4, klk = (k >> 2) & 3, q = k & 3;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ml = 4 * mq + e, rtile = ml >> 4, mli = ml & 15;
                Al[buf][(((kb * 8 + rtile) * 64) + klk * 16 + mli) * 4 + q] = sk_val<ROUND>(v[e]);
            }
        }
        constexpr int W = 16 * NCT;
        for (int idx = threadIdx.x; idx < KC * W; idx += SK_NT) {
            const int k = idx / W, n = idx - k * W, kk = k0 + k;
            const float v = (kk < kend && n < p.N) ? sk_val<ROUND>(p.B[(size_t)kk * p.ldb + n]) : 0.f;
            const int kb = k >> 4, klk = (k >> 2) & 3, q = k & 3, c = n >> 4, nli = n & 15;
            Bl[buf][(((kb * NCT + c) * 64) + klk * 16 + nli) * 4 + q] = v;
        }
    };
    const int nchunk = (kend - kbeg + KC - 1) / KC;
    if (nchunk > 0) stage(kbeg, 0);
    __syncthreads();
#pragma unroll 1
    for (int ch = 0; ch < nchunk; ++ch) {
        const int buf = ch & 1;
        if (ch + 1 < nchunk) stage(kbeg + (ch + 1) * KC, buf ^ 1);
#pragma unroll
        for (int kb = 0; kb < KC / 16; ++kb) {
            f32x4s b[NCT];
#pragma unroll
            for (int c = 0; c < NCT; ++c) b[c] = *reinterpret_cast<const f32x4s *>(&Bl[buf][((kb * NCT + c) * 64 + lane) * 4]);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const f32x4s av = *reinterpret_cast<const f32x4s *>(&Al[buf][((kb * 8 + wave * 2 + rt) * 64 + lane) * 4]);
#pragma unroll
                for (int c = 0; c < NCT; ++c) {
                    acc[rt][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, b[c].x, acc[rt][c], 0, 0, 0);
                    acc[rt][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, b[c].y, acc[rt][c], 0, 0, 0);
                    acc[rt][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, b[c].z, acc[rt][c], 0, 0, 0);
                    acc[rt][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, b[c].w, acc[rt][c], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    float *out = p.slab ? p.slab + (size_t)blockIdx.y * p.M * p.N : nullptr;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            const int col = 16 * c + li;
            if (col >= p.N) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wave * 32 + rt * 16 + 4 * lk + r;
                if (row >= p.M) continue;
                if (out) out[(size_t)row * p.N + col] = acc[rt][c][r];
                else {
                    float *cp = p.C + (size_t)row * p.ldc + col;
                    float v = p.alpha * acc[rt][c][r] + (p.bias ? p.bias[col] : 0.f);
                    if (p.beta != 0.f) v += p.beta * *cp;
                    *cp = v;
                }
            }
        }
}

// C = alpha * sum_s slab[s] + beta * C + bias
__global__ __launch_bounds__(256) void skinny_reduce_kernel(const float *__restrict__ slab, int nslab, int M, int N, float alpha,
                                                            float beta, float *__restrict__ C, int ldc, const float *__restrict__ bias)
{
    const size_t total = (size_t)M * N;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(e / N), col = (int)(e % N);
        float s = slab[e];
        for (int k = 1; k < nslab; ++k) s += slab[(size_t)k * total + e];
        float *c = C + (size_t)row * ldc + col;
        float o = alpha * s + (bias ? bias[col] : 0.f);
        if (beta != 0.f) o += beta * *c;
        *c = o;
    }
}

inline int sk_tn_slabs(int M, int K)
{
    const int blocks = lc_cdiv(M, SK_ROWS);
    int s = lc_cdiv(512, blocks);                    // two workgroups per CU
    const int cap = K / 512 > 0 ? K / 512 : 1;       // slabs at least 512 rows deep
    s = s < cap ? s : cap;
    return s < 1 ? 1 : (s > 64 ? 64 : s);
}

}  // namespace

// Whether (and with how much workspace) the skinny kernels take a product: N <= 80 output columns against >= 4096 rows of
// work, operands aligned for 16-byte row loads.  gemm.hip asks before it picks a general kernel.
bool lc_skinny_takes(int ta, int tb, int M, int N, int K, const float *A, int lda, const float *B, int ldb)
{
    if (tb || N < 1 || N > 80) return false;
    if (!ta) return M >= 4096 && K >= 64 && K % 4 == 0 && lda % 4 == 0 && (((uintptr_t)A) & 15) == 0;
    return K >= 4096 && M >= 128 && lda % 4 == 0 && (((uintptr_t)A) & 15) == 0;
}
size_t lc_skinny_workspace_bytes(int ta, int tb, int M, int N, int K)
{
    if (!ta || tb || N > 80) return 0;
    const int s = sk_tn_slabs(M, K);
    return s > 1 ? (size_t)s * M * N * sizeof(float) : 0;
}
int lc_skinny_launch(bool round_bf16, int ta, int M, int N, int K, float alpha, const float *A, int lda, const float *B, int ldb,
                     float beta, float *C, int ldc, const float *bias, void *workspace, size_t workspace_bytes, hipStream_t s)
{
    SkinnyArgs p;
    p.M = M; p.N = N; p.K = K; p.alpha = alpha; p.beta = beta;
    p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.C = C; p.ldc = ldc; p.bias = bias;
    p.kchunk = K; p.slab = nullptr;
    const int nct = (N + 15) / 16;
#define LC_SK(KERNEL, GRID)                                                                                  \
    do {                                                                                                     \
        switch (nct * 2 + (round_bf16 ? 1 : 0)) {                                                            \
        case 2: hipLaunchKernelGGL((KERNEL<1, false>), GRID, dim3(SK_NT), 0, s, p); break;                   \
        case 3: hipLaunchKernelGGL((KERNEL<1, true>), GRID, dim3(SK_NT), 0, s, p); break;                    \
        case 4: hipLaunchKernelGGL((KERNEL<2, false>), GRID, dim3(SK_NT), 0, s, p); break;                   \
        case 5: hipLaunchKernelGGL((KERNEL<2, true>), GRID, dim3(SK_NT), 0, s, p); break;                    \
        case 6: hipLaunchKernelGGL((KERNEL<3, false>), GRID, dim3(SK_NT), 0, s, p); break;                   \
        case 7: hipLaunchKernelGGL((KERNEL<3, true>), GRID, dim3(SK_NT), 0, s, p); break;                    \
        case 8: hipLaunchKernelGGL((KERNEL<4, false>), GRID, dim3(SK_NT), 0, s, p); break;                   \
        case 9: hipLaunchKernelGGL((KERNEL<4, true>), GRID, dim3(SK_NT), 0, s, p); break;                    \
        case 10: hipLaunchKernelGGL((KERNEL<5, false>), GRID, dim3(SK_NT), 0, s, p); break;                  \
        default: hipLaunchKernelGGL((KERNEL<5, true>), GRID, dim3(SK_NT), 0, s, p); break;                   \
        }                                                                                                    \
    } while (0)
    if (!ta) {
        LC_SK(gemm_skinny_nn_kernel, dim3((unsigned)lc_cdiv(M, SK_ROWS)));
        LC_CHECK_LAUNCH("gemm_skinny_nn");
        return LC_OK;
    }
    int nslab = sk_tn_slabs(M, K);
    if (nslab > 1 && (!workspace || workspace_bytes < (size_t)nslab * M * N * sizeof(float))) nslab = 1;
    p.kchunk = lc_cdiv(lc_cdiv(K, nslab), 16) * 16;
    nslab = lc_cdiv(K, p.kchunk);
    p.slab = nslab > 1 ? (float *)workspace : nullptr;
    LC_SK(gemm_skinny_tn_kernel, dim3((unsigned)lc_cdiv(M, SK_ROWS), (unsigned)nslab));
    LC_CHECK_LAUNCH("gemm_skinny_tn");
    if (nslab > 1) {
        int g = (int)(((size_t)M * N + 255) / 256);
        if (g > 1024) g = 1024;
        hipLaunchKernelGGL(skinny_reduce_kernel, dim3(g), dim3(256), 0, s, (const float *)workspace, nslab, M, N, alpha, beta, C, ldc, bias);
        LC_CHECK_LAUNCH("gemm_skinny_tn (slab reduction)");
    }
    return LC_OK;
#undef LC_SK
}
