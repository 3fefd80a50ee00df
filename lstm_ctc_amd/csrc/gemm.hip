// gemm.hip — float32 GEMM on the gfx950 f32 MFMA pipe (v_mfma_f32_32x32x2_f32, exact f32).
//
// C[M,N] = alpha * op(A)[M,K] * op(B)[K,N] + beta * C (+ bias[N]).  Row-major storage.
// Replaces tf.matmul / tf.nn.xw_plus_b (mobvoi/lstm_ctc nnet/bilstm.py:249, nnet/moe.py:43,58) and
// the batched (hoisted-over-T) halves of the LSTMCell kernel matmul (bilstm.py:129-136) and of
// their gradients (tf.gradients, nnet/graph.py:190).
//
// Tiling: 128x128x16 block tile, 256 threads = 4 waves in 2x2, each wave owns a 64x64 patch as
// 2x2 MFMA 32x32 tiles (64 accumulator registers).  Both operand tiles are kept K-MAJOR in LDS
// (As[k][m], Bs[k][n]) so the one-float-per-lane MFMA fragments (lane -> row l&31, k = l>>5) are
// conflict-free ds_read_b32; a k-minor source (NN's A, NT's B) is transposed on the LDS store.
// Register-staged global prefetch TWO tiles deep (tile k+1 waits in registers for the LDS buffer tile k-1
// frees, tile k+2 is in flight), two LDS buffers, one barrier per K step; fragment reads are
// software-pipelined one k-pair ahead of the MFMAs behind counted lgkmcnt waits.
//   FAST variant: M,N multiples of 128, K multiple of 16, 16-byte aligned rows - no bounds checks, loads are
//   buffer_load_dwordx4 from a wave-uniform tile origin + a constant per-thread offset (every hot GEMM of
//   the c2-c5 configs); the generic variant handles edges (layer-0 K=40, head N=V).
// Split-K: tall-K products with few output tiles (the weight gradients X^T.dZ, K = T*B) are cut along K
// across blockIdx.z into per-slice slabs in a caller-provided workspace and reduced deterministically.
//
// Where the time goes (MI355X, 64000x4096x2048, rocprofv3 PMC + ablation builds, round 1):
//   MFMA pipe busy 85.5 % of GRBM_GUI_ACTIVE before / ~90 % after the two-deep prefetch (hipBLASLt's
//   MT128x128x64 kernel: 94 %), same 2.36-2.38 GHz effective clock for both, so the gap is pipeline, not DVFS.
//   Ablations of the old single-stage loop: no global loads +5 %, no LDS stores/barrier +2 %, no LDS reads
//   +1.5 %, MFMA-only skeleton (incl. the 1 GB C store) 145 TF = what hipBLASLt reaches on this shape.
//   1 / 2 / 4 workgroups per CU: 110 / 129 / 133 TF.  BK = 32, tile-order and s_setprio variants: no gain.
#include "common.h"
#include "gemm_epi.h"
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
// buffer_load_dwordx4 through the raw LLVM intrinsic, bound by its asm label (this toolchain's
// __builtin_amdgcn_raw_buffer_load_b128 lowers to a single-dword load)
__device__ f32x4v lc_raw_buffer_load_f32x4(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
// (integer flavour for the bf16 shadows: extracting the lanes of the f32 flavour through per-element bit casts makes
// this toolchain narrow the load to one dword)
__device__ i32x4 lc_raw_buffer_load_i32x4(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4i32");

namespace {

#ifndef LC_GEMM_BK
#define LC_GEMM_BK 16
#endif
#ifndef LC_GEMM_GROUP_M
#define LC_GEMM_GROUP_M 8
#endif
constexpr int BM = 128, BN = 128, BK = LC_GEMM_BK, NT = 256;

// One float4 of a 128 x BK operand tile per call: thread-slot f = threadIdx.x + NT*I of BK*32 slots.
// KMAJOR: the stored matrix has k along rows (element (k, c) at src[k*ld + c]).
// (Named registers, not arrays: float4 arrays passed by reference end up in scratch.)
template <bool KMAJOR, bool FAST, int I>
__device__ __forceinline__ float4 tile_load1(const float *__restrict__ src, int ld, int c0, int cmax, int k0,
                                             int kmax, bool vec_ok)
{
    const int f = threadIdx.x + NT * I;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (FAST) {
        if (KMAJOR) v = *reinterpret_cast<const float4 *>(src + (size_t)(k0 + f / 32) * ld + c0 + 4 * (f % 32));
        else v = *reinterpret_cast<const float4 *>(src + (size_t)(c0 + f % 128) * ld + k0 + 4 * (f / 128));
    } else if (KMAJOR) {
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
    return v;
}

template <bool KMAJOR, int I>
__device__ __forceinline__ void tile_store1(float *__restrict__ lds /*[BK][128]*/, const float4 v)
{
    const int f = threadIdx.x + NT * I;
    if (KMAJOR) {
        *reinterpret_cast<float4 *>(lds + (f / 32) * 128 + 4 * (f % 32)) = v;
    } else {
        const int c = f % 128, k = 4 * (f / 128);
        lds[(k + 0) * 128 + c] = v.x;
        lds[(k + 1) * 128 + c] = v.y;
        lds[(k + 2) * 128 + c] = v.z;
        lds[(k + 3) * 128 + c] = v.w;
    }
}

// NLD = BK/8 float4 per thread per operand tile, held in named registers r0..r3
struct TileRegs { float4 r0, r1, r2, r3; };
template <bool KMAJOR, bool FAST>
__device__ __forceinline__ void tile_load(const float *__restrict__ src, int ld, int c0, int cmax, int k0, int kmax,
                                          bool vec_ok, TileRegs &t)
{
    t.r0 = tile_load1<KMAJOR, FAST, 0>(src, ld, c0, cmax, k0, kmax, vec_ok);
    t.r1 = tile_load1<KMAJOR, FAST, 1>(src, ld, c0, cmax, k0, kmax, vec_ok);
    if constexpr (BK >= 32) {
        t.r2 = tile_load1<KMAJOR, FAST, 2>(src, ld, c0, cmax, k0, kmax, vec_ok);
        t.r3 = tile_load1<KMAJOR, FAST, 3>(src, ld, c0, cmax, k0, kmax, vec_ok);
    }
}
template <bool KMAJOR>
__device__ __forceinline__ void tile_store(float *__restrict__ lds, const TileRegs &t)
{
    tile_store1<KMAJOR, 0>(lds, t.r0);
    tile_store1<KMAJOR, 1>(lds, t.r1);
    if constexpr (BK >= 32) {
        tile_store1<KMAJOR, 2>(lds, t.r2);
        tile_store1<KMAJOR, 3>(lds, t.r3);
    }
}

// FAST-path loads go through a buffer descriptor rebuilt per K step from a wave-uniform 64-bit base (two SALU
// adds) plus a per-thread 32-bit byte offset that never changes: no per-step 64-bit VALU address arithmetic and
// one VGPR per load slot instead of a pointer pair.  The per-tile extent (128 rows or BK rows of ld floats) is what
// has to fit 32 bits, not the matrix.
__device__ __forceinline__ float4 buf_load16(const float *uniform_base, unsigned voff_bytes)
{
    const unsigned long long b = (unsigned long long)uniform_base;
    // {base[31:0], base[47:32] | stride 0, num_records = 4 GB - 1, gfx9 raw-buffer dword 3}
    const i32x4 rsrc = {(int)(unsigned)b, (int)((b >> 32) & 0xffffu), -1, 0x00020000};
    const f32x4v v = lc_raw_buffer_load_f32x4(rsrc, (int)voff_bytes, 0, 0);
    return make_float4(v.x, v.y, v.z, v.w);
}
// byte offset of thread-slot f = tid + NT*I inside a 128 x BK operand tile whose origin is the descriptor base
template <bool KMAJOR, int I>
__device__ __forceinline__ unsigned tile_voff(int ld)
{
    const int f = threadIdx.x + NT * I;
    return KMAJOR ? (unsigned)(((f / 32) * ld + 4 * (f % 32)) * 4) : (unsigned)(((f % 128) * ld + 4 * (f / 128)) * 4);
}
struct TileOff { unsigned v0, v1, v2, v3; };
template <bool KMAJOR>
__device__ __forceinline__ TileOff tile_offsets(int ld)
{
    TileOff o;
    o.v0 = tile_voff<KMAJOR, 0>(ld);
    o.v1 = tile_voff<KMAJOR, 1>(ld);
    o.v2 = BK >= 32 ? tile_voff<KMAJOR, 2>(ld) : 0u;
    o.v3 = BK >= 32 ? tile_voff<KMAJOR, 3>(ld) : 0u;
    return o;
}
__device__ __forceinline__ void tile_load_buf(const float *uniform_base, const TileOff &o, TileRegs &t)
{
    t.r0 = buf_load16(uniform_base, o.v0);
    t.r1 = buf_load16(uniform_base, o.v1);
    if constexpr (BK >= 32) {
        t.r2 = buf_load16(uniform_base, o.v2);
        t.r3 = buf_load16(uniform_base, o.v3);
    }
}

struct GemmArgs {
    EpiArgs epi;
    int M, N, K;
    float alpha, beta;
    const float *A; int lda;
    const float *B; int ldb;
    float *C; int ldc;
    const float *bias;
    int vecA, vecB;
    int kchunk;        // K range per blockIdx.z slice (multiple of BK); == K when not split
    float *slab;       // split-K: [gridDim.z][slab_rows][slab_ld] partial products (this launch's origin), else nullptr
    size_t slab_slice; // floats per K slice of the slab
    int slab_ld;
    // gemm_f32g_kernel<false, false, SEG2>: the reduction continues over a second operand pair (same leading dimensions)
    // behind the first k1 indices - see gemm_bf16g_kernel's SEG2
    const float *A2, *B2;
    int k1;
};

// Epilogue shared by the f32 and bf16 kernels (the 32x32 C/D layout does not depend on the operand type):
// lane holds C[row = (r&3) + 8*(r>>2) + 4*(lane>>5)][col = lane&31] of each 32x32 tile.
template <bool FAST>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs &p, const f32x16 (&acc)[2][2], int m0, int n0, int wm,
                                              int wn, int lr, int lk)
{
    const int M = p.M, N = p.N;
    if (p.slab) {
        float *S = p.slab + (size_t)blockIdx.z * p.slab_slice;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = n0 + wn * 64 + j * 32 + lr;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                    if (FAST || (row < M && col < N)) S[(size_t)row * p.slab_ld + col] = acc[i][j][r];
                }
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + lr;
            if (!FAST && col >= N) continue;
            const float bv = p.bias ? p.bias[col] : 0.f;
            unsigned est;
            int ecm;
            epi_column(p.epi, col, est, ecm);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (FAST || row < M) {
                    float *c = p.C + (size_t)row * p.ldc + col;
                    float v = p.alpha * acc[i][j][r] + bv;
                    if (p.beta != 0.f) v += p.beta * *c;
                    epi_store(p.epi, c, v, row, col, est, ecm);
                }
            }
        }
}

// Tile order: XCD-aware remap + grouped order (see the f32 kernel); returns (bm, bn).
__device__ __forceinline__ void gemm_tile_order(int M, int N, int &bm, int &bn)
{
    const int nbm = (M + BM - 1) / BM, nbn = (N + BN - 1) / BN;
    const int nwg = nbm * nbn;
    int bid = blockIdx.x;
    {
        const int q = nwg / 8, rr = nwg % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
    }
    constexpr int GROUP_M = LC_GEMM_GROUP_M;
    const int gsz = GROUP_M * nbn;
    const int first_m = (bid / gsz) * GROUP_M;
    const int gm = min(nbm - first_m, GROUP_M);
    bm = first_m + (bid % gsz) % gm;
    bn = (bid % gsz) / gm;
}

#ifndef LC_GEMM_MINWAVES
#define LC_GEMM_MINWAVES 2
#endif
template <bool TA, bool TB, bool FAST>
__global__ __launch_bounds__(NT, LC_GEMM_MINWAVES) void gemm_f32_kernel(GemmArgs p)
{
    __shared__ __attribute__((aligned(16))) float lds[2 * BK * BM + 2 * BK * BN];
    float *As0 = lds, *Bs0 = lds + 2 * BK * BM;
    const int M = p.M, N = p.N;
    // XCD-aware remap: a contiguous chunk of the tile order per XCD (block b runs on XCD b % 8) ...
    const int nbm = (M + BM - 1) / BM, nbn = (N + BN - 1) / BN;
    const int nwg = nbm * nbn;
    int bid = blockIdx.x;
    {
        const int q = nwg / 8, rr = nwg % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
    }
    // ... and a grouped order inside it: the ~64 tiles in flight on one XCD form an 8(m) x 8(n) patch, so
    // each A row panel and each B column panel is shared 8 ways out of that XCD's L2
    constexpr int GROUP_M = LC_GEMM_GROUP_M;
    const int gsz = GROUP_M * nbn;
    const int first_m = (bid / gsz) * GROUP_M;
    const int gm = min(nbm - first_m, GROUP_M);
    const int bm = first_m + (bid % gsz) % gm, bn = (bid % gsz) / gm;
    const int m0 = bm * BM, n0 = bn * BN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int kbeg = blockIdx.z * p.kchunk;
    const int kend = min(p.K, kbeg + p.kchunk);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // A operand: TA -> stored [K,M] (k-major); else stored [M,K] (k-minor)
    // B operand: TB -> stored [N,K] (k-minor); else stored [K,N] (k-major)
    // Register-staged prefetch, TWO tiles deep: while tile kt feeds the MFMAs out of LDS, tile kt+1 sits in one
    // register set (loaded during step kt-1, written to the idle LDS buffer at the end of step kt) and the loads
    // of tile kt+2 go out into the other set.  One tile of distance did not cover a first-touch HBM miss
    // (ablation: the loads alone cost 5 % of the kernel).  The two sets alternate, so the loop is unrolled by two.
    TileRegs ra0, rb0, ra1, rb1;
    const int nk = (kend - kbeg + BK - 1) / BK;
    // FAST path: wave-uniform tile origins + constant per-thread offsets (see buf_load16)
    const float *abase = TA ? p.A + (size_t)kbeg * p.lda + m0 : p.A + (size_t)m0 * p.lda + kbeg;
    const float *bbase = !TB ? p.B + (size_t)kbeg * p.ldb + n0 : p.B + (size_t)n0 * p.ldb + kbeg;
    const size_t astep = TA ? (size_t)BK * p.lda : (size_t)BK, bstep = !TB ? (size_t)BK * p.ldb : (size_t)BK;
    TileOff oa, ob;
    if constexpr (FAST) { oa = tile_offsets<TA>(p.lda); ob = tile_offsets<!TB>(p.ldb); }
    auto load_tile = [&](int j, TileRegs &a, TileRegs &b) {
        if constexpr (FAST) {
            tile_load_buf(abase + (size_t)j * astep, oa, a);
            tile_load_buf(bbase + (size_t)j * bstep, ob, b);
        } else {
            tile_load<TA, FAST>(p.A, p.lda, m0, M, kbeg + j * BK, kend, p.vecA, a);
            tile_load<!TB, FAST>(p.B, p.ldb, n0, N, kbeg + j * BK, kend, p.vecB, b);
        }
    };
    load_tile(0, ra0, rb0);
    load_tile(min(1, nk - 1), ra1, rb1);
    tile_store<TA>(As0, ra0);
    tile_store<!TB>(Bs0, rb0);
    __syncthreads();
    const int lr = lane & 31, lk = lane >> 5;
    // One K step: CUR = LDS buffer holding tile KT; (LA, LB) receive tile KT+2; (SA, SB) hold tile KT+1.
    // Branch-free: indices past the end are clamped (a redundant re-load / re-store of the last tile).
#define LC_GEMM_STEP(KT, CUR, LA, LB, SA, SB)                                                               \
    {                                                                                                       \
        load_tile(min((KT) + 2, nk - 1), LA, LB);                                                           \
        const float *as = As0 + (CUR) * BK * BM + wm * 64 + lr + lk * 128;                                  \
        const float *bs = Bs0 + (CUR) * BK * BN + wn * 64 + lr + lk * 128;                                  \
        /* software-pipelined fragments: the ds_reads of step kk+1 are in flight under the 4 MFMAs of kk */ \
        float a0 = as[0], a1 = as[32], b0 = bs[0], b1 = bs[32];                                             \
        _Pragma("unroll") for (int kk = 0; kk < BK / 2; ++kk) {                                             \
            float na0 = 0.f, na1 = 0.f, nb0 = 0.f, nb1 = 0.f;                                               \
            if (kk + 1 < BK / 2) {                                                                          \
                na0 = as[(kk + 1) * 256]; na1 = as[(kk + 1) * 256 + 32];                                    \
                nb0 = bs[(kk + 1) * 256]; nb1 = bs[(kk + 1) * 256 + 32];                                    \
            }                                                                                               \
            __builtin_amdgcn_sched_barrier(0);                                                              \
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);                   \
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);                   \
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);                   \
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);                   \
            __builtin_amdgcn_sched_barrier(0);                                                              \
            a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;                                                         \
        }                                                                                                   \
        tile_store<TA>(As0 + ((CUR) ^ 1) * BK * BM, SA);                                                    \
        tile_store<!TB>(Bs0 + ((CUR) ^ 1) * BK * BN, SB);                                                   \
        __syncthreads();                                                                                    \
    }
    int kt = 0;
    for (; kt + 2 <= nk; kt += 2) {
        LC_GEMM_STEP(kt, 0, ra0, rb0, ra1, rb1)
        LC_GEMM_STEP(kt + 1, 1, ra1, rb1, ra0, rb0)
    }
    if (kt < nk) LC_GEMM_STEP(kt, 0, ra0, rb0, ra1, rb1)
#undef LC_GEMM_STEP
    gemm_epilogue<FAST>(p, acc, m0, n0, wm, wn, lr, lk);
}

// ------------------------------------------------------------------------------------------------------------
// bf16-operand variant (BASELINE config c5: "bf16 MFMA gate GEMMs", fp32 accumulate / outputs / state).
// The operands stay fp32 in HBM; the loader rounds them to bf16 (round-to-nearest-even, v_cvt_pk_bf16_f32) on the
// way into LDS, so the product is  sum_k bf16(A[m,k]) * bf16(B[k,n])  accumulated in fp32 by
// v_mfma_f32_32x32x16_bf16 (16x the f32 MFMA rate).  Same 128x128 block tile / 2x2 waves / epilogue as above;
// BK = 32, LDS tiles are [row][k] with k contiguous and an 80-byte row pitch (32 bf16 + 8 pad): the 16-byte
// fragment reads (lane -> row l&31, k-octet l>>5) of 16 consecutive rows then fall into 16 disjoint 4-bank groups.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int HBK = 32, HP = 40;          // k per tile, row pitch in bf16 elements

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi)
{
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, v);
}

struct HTileRegs { float4 r0, r1, r2, r3; };

// k-minor source (stored [C, K], k contiguous): slot f = tid + 256*I -> row f/8, k-quad f%8
// k-major source (stored [K, C], c contiguous): thread -> 4(k) x 4(c) block: k-quad tid/32, c-quad tid%32.  (The
// transposing 8-byte LDS stores are 8-way bank-conflicted this way; k-quad-fastest removes the conflict but cuts the
// global loads into 128-byte pieces and measured 15-20 % slower - the loader is bound by the loads, not by LDS.)
template <bool KMAJOR, bool FAST>
__device__ __forceinline__ void htile_load(const float *__restrict__ src, int ld, int c0, int cmax, int k0, int kmax,
                                           bool vec_ok, HTileRegs &t)
{
    float4 *r[4] = {&t.r0, &t.r1, &t.r2, &t.r3};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int c, k;
        if (KMAJOR) { k = k0 + 4 * (threadIdx.x / 32) + i; c = c0 + 4 * (threadIdx.x % 32); }
        else { const int f = threadIdx.x + NT * i; c = c0 + f / 8; k = k0 + 4 * (f % 8); }
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (FAST) {
            v = KMAJOR ? *reinterpret_cast<const float4 *>(src + (size_t)k * ld + c)
                       : *reinterpret_cast<const float4 *>(src + (size_t)c * ld + k);
        } else if (KMAJOR) {
            if (k < kmax) {
                const float *q = src + (size_t)k * ld + c;
                if (vec_ok && c + 3 < cmax) v = *reinterpret_cast<const float4 *>(q);
                else {
                    if (c < cmax) v.x = q[0];
                    if (c + 1 < cmax) v.y = q[1];
                    if (c + 2 < cmax) v.z = q[2];
                    if (c + 3 < cmax) v.w = q[3];
                }
            }
        } else {
            if (c < cmax) {
                const float *q = src + (size_t)c * ld + k;
                if (vec_ok && k + 3 < kmax) v = *reinterpret_cast<const float4 *>(q);
                else {
                    if (k < kmax) v.x = q[0];
                    if (k + 1 < kmax) v.y = q[1];
                    if (k + 2 < kmax) v.z = q[2];
                    if (k + 3 < kmax) v.w = q[3];
                }
            }
        }
        *r[i] = v;
    }
}

// per-thread byte offsets of the four float4 slots inside a 128 x 32 operand tile (see htile_load)
template <bool KMAJOR>
__device__ __forceinline__ TileOff htile_offsets(int ld)
{
    unsigned v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = threadIdx.x + NT * i;
        v[i] = KMAJOR ? (unsigned)(((4 * (threadIdx.x / 32) + i) * ld + 4 * (threadIdx.x % 32)) * 4)
                      : (unsigned)(((f / 8) * ld + 4 * (f % 8)) * 4);
    }
    TileOff o = {v[0], v[1], v[2], v[3]};
    return o;
}

template <bool KMAJOR>
__device__ __forceinline__ void htile_store(unsigned short *__restrict__ lds /*[128][HP]*/, const HTileRegs &t)
{
    if (KMAJOR) {     // registers hold rows k..k+3 of columns c..c+3: transpose, 4 bf16 (8 bytes) per column
        const int kq = threadIdx.x / 32, c = 4 * (threadIdx.x % 32);
        uint2 *d0 = reinterpret_cast<uint2 *>(lds + (c + 0) * HP + 4 * kq);
        uint2 *d1 = reinterpret_cast<uint2 *>(lds + (c + 1) * HP + 4 * kq);
        uint2 *d2 = reinterpret_cast<uint2 *>(lds + (c + 2) * HP + 4 * kq);
        uint2 *d3 = reinterpret_cast<uint2 *>(lds + (c + 3) * HP + 4 * kq);
        *d0 = make_uint2(pack_bf16(t.r0.x, t.r1.x), pack_bf16(t.r2.x, t.r3.x));
        *d1 = make_uint2(pack_bf16(t.r0.y, t.r1.y), pack_bf16(t.r2.y, t.r3.y));
        *d2 = make_uint2(pack_bf16(t.r0.z, t.r1.z), pack_bf16(t.r2.z, t.r3.z));
        *d3 = make_uint2(pack_bf16(t.r0.w, t.r1.w), pack_bf16(t.r2.w, t.r3.w));
    } else {
        const float4 *r[4] = {&t.r0, &t.r1, &t.r2, &t.r3};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = threadIdx.x + NT * i, c = f / 8, kq = f % 8;
            *reinterpret_cast<uint2 *>(lds + c * HP + 4 * kq) =
                make_uint2(pack_bf16(r[i]->x, r[i]->y), pack_bf16(r[i]->z, r[i]->w));
        }
    }
}

template <bool TA, bool TB, bool FAST>
__global__ __launch_bounds__(NT, 2) void gemm_bf16_kernel(GemmArgs p)
{
    __shared__ __attribute__((aligned(16))) unsigned short lds[4 * 128 * HP];     // A0 A1 B0 B1: 40 KB
    unsigned short *As0 = lds, *Bs0 = lds + 2 * 128 * HP;
    const int M = p.M, N = p.N;
    int bm, bn;
    gemm_tile_order(M, N, bm, bn);
    const int m0 = bm * BM, n0 = bn * BN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int kbeg = blockIdx.z * p.kchunk;
    const int kend = min(p.K, kbeg + p.kchunk);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Two-tile-deep register prefetch and (FAST path) buffer loads from a wave-uniform tile origin, as in the f32
    // kernel - with the MFMAs 16x faster the loads are what this kernel waits for.
    HTileRegs ra0, rb0, ra1, rb1;
    const int nk = (kend - kbeg + HBK - 1) / HBK;
    const float *abase = TA ? p.A + (size_t)kbeg * p.lda + m0 : p.A + (size_t)m0 * p.lda + kbeg;
    const float *bbase = !TB ? p.B + (size_t)kbeg * p.ldb + n0 : p.B + (size_t)n0 * p.ldb + kbeg;
    const size_t astep = TA ? (size_t)HBK * p.lda : (size_t)HBK, bstep = !TB ? (size_t)HBK * p.ldb : (size_t)HBK;
    TileOff oa, ob;
    if constexpr (FAST) { oa = htile_offsets<TA>(p.lda); ob = htile_offsets<!TB>(p.ldb); }
    // (Starting each workgroup's K walk at a different tile, as the recurrent step kernel does, was tried here: it
    // costs the L2 sharing of the A/B panels between the 8x8 tiles of an XCD patch and measured 20-40 % slower.)
    auto load_tile = [&](int j, HTileRegs &a, HTileRegs &b) {
        if constexpr (FAST) {
            const float *ab = abase + (size_t)j * astep, *bb = bbase + (size_t)j * bstep;
            a.r0 = buf_load16(ab, oa.v0); a.r1 = buf_load16(ab, oa.v1); a.r2 = buf_load16(ab, oa.v2); a.r3 = buf_load16(ab, oa.v3);
            b.r0 = buf_load16(bb, ob.v0); b.r1 = buf_load16(bb, ob.v1); b.r2 = buf_load16(bb, ob.v2); b.r3 = buf_load16(bb, ob.v3);
        } else {
            htile_load<TA, FAST>(p.A, p.lda, m0, M, kbeg + j * HBK, kend, p.vecA, a);
            htile_load<!TB, FAST>(p.B, p.ldb, n0, N, kbeg + j * HBK, kend, p.vecB, b);
        }
    };
    load_tile(0, ra0, rb0);
    load_tile(min(1, nk - 1), ra1, rb1);
    htile_store<TA>(As0, ra0);
    htile_store<!TB>(Bs0, rb0);
    __syncthreads();
    const int lr = lane & 31, lk = lane >> 5;
#define LC_HGEMM_STEP(KT, CUR, LA, LB, SA, SB)                                                              \
    {                                                                                                       \
        load_tile(min((KT) + 2, nk - 1), LA, LB);                                                           \
        const unsigned short *as = As0 + (CUR) * 128 * HP + (wm * 64 + lr) * HP + lk * 8;                   \
        const unsigned short *bs = Bs0 + (CUR) * 128 * HP + (wn * 64 + lr) * HP + lk * 8;                   \
        bf16x8 a[2][2], b[2][2];                                                                            \
        _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                     \
            a[s][0] = *reinterpret_cast<const bf16x8 *>(as + s * 16);                                       \
            a[s][1] = *reinterpret_cast<const bf16x8 *>(as + 32 * HP + s * 16);                             \
            b[s][0] = *reinterpret_cast<const bf16x8 *>(bs + s * 16);                                       \
            b[s][1] = *reinterpret_cast<const bf16x8 *>(bs + 32 * HP + s * 16);                             \
        }                                                                                                   \
        _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                     \
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][0], b[s][0], acc[0][0], 0, 0, 0);      \
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][0], b[s][1], acc[0][1], 0, 0, 0);      \
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][1], b[s][0], acc[1][0], 0, 0, 0);      \
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][1], b[s][1], acc[1][1], 0, 0, 0);      \
        }                                                                                                   \
        htile_store<TA>(As0 + ((CUR) ^ 1) * 128 * HP, SA);                                                  \
        htile_store<!TB>(Bs0 + ((CUR) ^ 1) * 128 * HP, SB);                                                 \
        __syncthreads();                                                                                    \
    }
    int kt = 0;
    for (; kt + 2 <= nk; kt += 2) {
        LC_HGEMM_STEP(kt, 0, ra0, rb0, ra1, rb1)
        LC_HGEMM_STEP(kt + 1, 1, ra1, rb1, ra0, rb0)
    }
    if (kt < nk) LC_HGEMM_STEP(kt, 0, ra0, rb0, ra1, rb1)
#undef LC_HGEMM_STEP
    gemm_epilogue<FAST>(p, acc, m0, n0, wm, wn, lr, lk);
}

// ------------------------------------------------------------------------------------------------------------
// bf16 SHADOW operands (config c5, second stage): both operands already bf16 in memory with k contiguous -
// A[M][K], B[N][K] ("NT" form) - so the loader is a plain 16-byte copy into LDS: no conversion, no transposition,
// half the bytes of the converting loader above.  lc_cast_bf16 makes the shadows (natural and transposed) from
// the fp32 tensors.  128x128x64 tile, LDS rows of 64 k + 8 pad (144-byte pitch: the 16-byte fragment reads of 16
// consecutive rows hit 16 disjoint 4-bank groups), two-tile-deep register prefetch, same epilogue.
constexpr int SBK = 64, SP = 72;
typedef unsigned int u32x4s __attribute__((ext_vector_type(4)));
struct STileRegs { uint4 r0, r1, r2, r3; };

__device__ __forceinline__ uint4 buf_load16u(const void *uniform_base, unsigned voff_bytes)
{
    const unsigned long long b = (unsigned long long)uniform_base;
    const i32x4 rsrc = {(int)(unsigned)b, (int)((b >> 32) & 0xffffu), -1, 0x00020000};
    const i32x4 v = lc_raw_buffer_load_i32x4(rsrc, (int)voff_bytes, 0, 0);
    return make_uint4((unsigned)v.x, (unsigned)v.y, (unsigned)v.z, (unsigned)v.w);
}

// unit u = tid + 256*i of the 128 x 64 tile: row u/8, k-octet u%8 (8 bf16 = 16 bytes)
template <bool FAST>
__device__ __forceinline__ void stile_load(const unsigned short *__restrict__ src, int ld, int r0, int rmax, int k0,
                                           int kmax, const unsigned short *uniform_origin, STileRegs &t)
{
    uint4 *r[4] = {&t.r0, &t.r1, &t.r2, &t.r3};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int u = threadIdx.x + NT * i, row = u / 8, oct = u % 8;
        if constexpr (FAST) {
            *r[i] = buf_load16u(uniform_origin, (unsigned)((row * ld + oct * 8) * 2));
        } else {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            const int gr = r0 + row, gk = k0 + oct * 8;
            if (gr < rmax && gk < kmax) {                       // K % 8 == 0 and ld % 8 == 0: whole units only
                v = *reinterpret_cast<const uint4 *>(src + (size_t)gr * ld + gk);
            }
            *r[i] = v;
        }
    }
}
__device__ __forceinline__ void stile_store(unsigned short *__restrict__ lds /*[128][SP]*/, const STileRegs &t)
{
    const uint4 *r[4] = {&t.r0, &t.r1, &t.r2, &t.r3};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int u = threadIdx.x + NT * i, row = u / 8, oct = u % 8;
        *reinterpret_cast<uint4 *>(lds + row * SP + oct * 8) = *r[i];
    }
}

struct SGemmArgs {
    GemmArgs g;                       // A / B unused (fp32 pointers); sizes, C, bias, alpha/beta, kchunk, slab
    const unsigned short *A, *B;      // bf16 bits
    // gemm_bf16g_kernel<.., SEG2>: the reduction continues over a SECOND operand pair (same leading dimensions, k
    // contiguous) behind the first k1 indices - sum_k A[m,k] B[n,k] + sum_k A2[m,k] B2[n,k] in one pass over C
    const unsigned short *A2, *B2;
    int k1;
};

template <bool FAST>
__global__ __launch_bounds__(NT, 2) void gemm_bf16s_kernel(SGemmArgs sp)
{
    __shared__ __attribute__((aligned(16))) unsigned short lds[4 * 128 * SP];     // A0 A1 B0 B1: 73.7 KB
    const GemmArgs &p = sp.g;
    unsigned short *As0 = lds, *Bs0 = lds + 2 * 128 * SP;
    const int M = p.M, N = p.N;
    int bm, bn;
    gemm_tile_order(M, N, bm, bn);
    const int m0 = bm * BM, n0 = bn * BN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int kbeg = blockIdx.z * p.kchunk;
    const int kend = min(p.K, kbeg + p.kchunk);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    STileRegs ra0, rb0, ra1, rb1;
    const int nk = (kend - kbeg + SBK - 1) / SBK;
    const unsigned short *abase = sp.A + (size_t)m0 * p.lda + kbeg, *bbase = sp.B + (size_t)n0 * p.ldb + kbeg;
    auto load_tile = [&](int j, STileRegs &a, STileRegs &b) {
        stile_load<FAST>(sp.A, p.lda, m0, M, kbeg + j * SBK, kend, abase + (size_t)j * SBK, a);
        stile_load<FAST>(sp.B, p.ldb, n0, N, kbeg + j * SBK, kend, bbase + (size_t)j * SBK, b);
    };
    load_tile(0, ra0, rb0);
    load_tile(min(1, nk - 1), ra1, rb1);
    stile_store(As0, ra0);
    stile_store(Bs0, rb0);
    __syncthreads();
    const int lr = lane & 31, lk = lane >> 5;
#define LC_SGEMM_STEP(KT, CUR, LA, LB, SA, SB)                                                              \
    {                                                                                                       \
        load_tile(min((KT) + 2, nk - 1), LA, LB);                                                           \
        const unsigned short *as = As0 + (CUR) * 128 * SP + (wm * 64 + lr) * SP + lk * 8;                   \
        const unsigned short *bs = Bs0 + (CUR) * 128 * SP + (wn * 64 + lr) * SP + lk * 8;                   \
        bf16x8 a0 = *reinterpret_cast<const bf16x8 *>(as), a1 = *reinterpret_cast<const bf16x8 *>(as + 32 * SP); \
        bf16x8 b0 = *reinterpret_cast<const bf16x8 *>(bs), b1 = *reinterpret_cast<const bf16x8 *>(bs + 32 * SP); \
        _Pragma("unroll") for (int q = 0; q < SBK / 16; ++q) {                                              \
            bf16x8 na0 = a0, na1 = a1, nb0 = b0, nb1 = b1;                                                  \
            if (q + 1 < SBK / 16) {                                                                         \
                na0 = *reinterpret_cast<const bf16x8 *>(as + (q + 1) * 16);                                 \
                na1 = *reinterpret_cast<const bf16x8 *>(as + 32 * SP + (q + 1) * 16);                       \
                nb0 = *reinterpret_cast<const bf16x8 *>(bs + (q + 1) * 16);                                 \
                nb1 = *reinterpret_cast<const bf16x8 *>(bs + 32 * SP + (q + 1) * 16);                       \
            }                                                                                               \
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);                \
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);                \
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);                \
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);                \
            a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;                                                         \
        }                                                                                                   \
        stile_store(As0 + ((CUR) ^ 1) * 128 * SP, SA);                                                      \
        stile_store(Bs0 + ((CUR) ^ 1) * 128 * SP, SB);                                                      \
        __syncthreads();                                                                                    \
    }
    int kt = 0;
    for (; kt + 2 <= nk; kt += 2) {
        LC_SGEMM_STEP(kt, 0, ra0, rb0, ra1, rb1)
        LC_SGEMM_STEP(kt + 1, 1, ra1, rb1, ra0, rb0)
    }
    if (kt < nk) LC_SGEMM_STEP(kt, 0, ra0, rb0, ra1, rb1)
#undef LC_SGEMM_STEP
    gemm_epilogue<FAST>(p, acc, m0, n0, wm, wn, lr, lk);
}

// ---- the same product on 256 x 256 x 64 tiles, operands DMA'd straight into LDS (buffer_load ... lds) -----------------
// The 128 x 128 kernel above reads 16 KB of LDS per 128 MFMA cycles and CU (four 64 x 64 wave tiles), stages every tile
// through registers and a ds_write pass, and tops out at 0.33 of the bf16 MFMA peak.  Here a workgroup is 8 waves in a
// 2 (M) x 4 (N) grid, each with a 128 x 64 tile (4 x 2 MFMA tiles of 32 x 32 x 16: 128 accumulator registers): 6 fragment
// reads per 8 MFMAs instead of 4 per 4, two waves per SIMD to cover each other's LDS latency, one workgroup per CU.
// Both operands are k-contiguous (NT form), so a tile is 256 rows of 128 bytes and 16-byte granules go from global memory
// to LDS unchanged: one `buffer_load_dwordx4 ... lds` per wave moves 8 rows (1 KB), no staging registers, no ds_write.
// An LDS-DMA writes lane l's granule at base + 16 l, so the image is linear and un-padded; bank conflicts of the
// ds_read_b128 fragment reads (lane = row, serviced in the 16-lane groups of MI355X_MICROARCH.md) are avoided by
// swizzling on the SOURCE side instead: LDS granule slot s of row r holds k-octet s ^ ((r >> 1) & 7).  Two LDS buffers
// (128 KB): tile t+1 is requested before tile t is multiplied and waited for at the barrier that ends tile t.
// Only launched on whole tiles (M, N multiples of 256, K chunks multiples of 64).
constexpr int GBM = 256, GBN = 256, GBK = 64, GNT = 512;
constexpr int G_OPERAND_BYTES = 256 * GBK * 2;               // one operand tile in LDS
__device__ __forceinline__ void gemm_tile_order_big(int M, int N, int &bm, int &bn, int bid = blockIdx.x)
{
    const int nbm = M / GBM, nbn = N / GBN;
    const int nwg = nbm * nbn;
    {   // consecutive tiles on one XCD (workgroups are dealt round-robin to the 8 XCDs)
        const int q = nwg / 8, rr = nwg % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
    }
    constexpr int GROUP_M = 4;                                // 4 x nbn patches: A panels shared out of one L2
    const int gsz = GROUP_M * nbn;
    const int first_m = (bid / gsz) * GROUP_M;
    const int gm = min(nbm - first_m, GROUP_M);
    bm = first_m + (bid % gsz) % gm;
    bn = (bid % gsz) / gm;
}

// ACOL / BCOL: the operand is stored K-MAJOR in memory (A as [K][M], B as [K][N]: the natural layout of an activation
// whose ROWS are the reduction index - the weight gradients X^T dZ of a train step) instead of k-contiguous.  Such a tile
// is 64 k-rows of 512 bytes; it goes to LDS by the same DMA (a wave instruction moves two k-rows) and the MFMA fragment -
// 8 consecutive k of one column per lane - comes out of it with the transposing read ds_read_b64_tr_b16 (two per
// fragment): a 16-lane group hands in the addresses of four k-rows x four 8-byte pieces and each lane receives one
// COLUMN of that 4 x 16 block.  The four k-rows of a group are 512 bytes apart, i.e. on the same banks: the source-side
// swizzle (LDS piece s of k-row k holds the row's piece s ^ 4 (k & 3)) spreads them over four disjoint 32-byte bank
// chunks, and over the other four for the second 16-lane group of a 32-lane half.  With this form the transposed bf16
// shadow copies of X, hs and dz (7 ms of a c5 step) are not needed at all.  A k-major operand's K tail needs no handling:
// rows past K lie beyond the buffer descriptor's range and arrive as zeros.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
// The two transposing reads of one fragment (k 0..3 and 4..7 of the lane's k-octet, 4 k-rows = 2048 bytes apart) as
// inline asm: through the builtin the compiler cannot tell the read from the LDS-DMA fill of the OTHER buffer (the
// intrinsic carries no memory operand) and puts `s_waitcnt vmcnt(0)` in front of the first fragment read of every k tile -
// the prefetch of the next tile became synchronous (959 instead of 1200 TFLOP/s on the dKx shape).  The price of asm:
// the waits are written by hand (tr16_wait: the fragments pass THROUGH the statement, so their consumers cannot be moved
// in front of it).
template <int OFF>
__device__ __forceinline__ void tr16_pair(unsigned lds_addr, i32x2 &lo, i32x2 &hi)
{
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"
                 : "=&v"(lo), "=&v"(hi)
                 : "v"(lds_addr), "n"(OFF), "n"(OFF + 2048));
}
typedef int i32x4g __attribute__((ext_vector_type(4)));
template <int OFF>
__device__ __forceinline__ void lds_read128(unsigned lds_addr, i32x4g &v)      // a k-contiguous fragment, same asm discipline
{
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(v) : "v"(lds_addr), "n"(OFF));
}
__device__ __forceinline__ bf16x8 frag_cast(i32x4g v)
{
    union { i32x4g i; bf16x8 b; } u;
    u.i = v;
    return u.b;
}
__device__ __forceinline__ bf16x8 tr16_join(i32x2 lo, i32x2 hi)
{
    union { i32x2 h[2]; bf16x8 v; } u;
    u.h[0] = lo; u.h[1] = hi;
    return u.v;
}
// PERSIST (unsplit products with an even number of k tiles: the forward products of a train step, 15.6 tiles per CU at c5):
// the grid is one workgroup per CU and every workgroup walks tiles blockIdx.x, + gridDim.x, ...  The LAST fill of a tile's k
// loop - redundant in the one-tile form - requests k tile 0 of the NEXT tile instead, so a tile's launch, its descriptor /
// address set-up and the exposed latency of its first fill (a few us of a ~55 us tile) run under the previous tile's last
// MFMAs, and the C stores of the finished tile (fire-and-forget) drain under the next tile's first k tiles.
// SEG2 (NT form, unsplit): two operand pairs share one accumulator - the dX of a bidirectional layer, dz_fwd . Kx_fwd^T +
// dz_bwd . Kx_bwd^T (nnet/bilstm.py:190-203: both cells read the same concatenated input), as ONE K = 2 x 4N product: C is
// written once instead of written, read back and written again by a second beta = 1 product (1.34 against 0.95 ms at c5's
// sizes, profiles/r6_gemm_persist_ab.txt).
template <bool ACOL, bool BCOL, bool PERSIST = false, bool SEG2 = false>
__global__ __launch_bounds__(GNT, 1) void gemm_bf16g_kernel(SGemmArgs sp)
{
    __shared__ __attribute__((aligned(1024))) unsigned char lds[4 * G_OPERAND_BYTES];      // A0 B0 A1 B1
    const GemmArgs &p = sp.g;
    int bm, bn;
    gemm_tile_order_big(p.M, p.N, bm, bn);
    int m0 = bm * GBM, n0 = bn * GBN;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int kbeg = blockIdx.z * p.kchunk;
    const int kend = min(p.K, kbeg + p.kchunk);
    const int nk = (kend - kbeg + GBK - 1) / GBK;         // (a ragged last chunk: k-major operands only, see above)

    // k-contiguous operand: rows = tile rows, the descriptor spans everything behind the tile origin.  k-major operand:
    // rows = k, the descriptor ends with the last valid k-row of this K chunk, so the tail of a ragged chunk reads as zero.
    const unsigned a_bytes = ACOL ? (unsigned)(((size_t)(kend - kbeg - 1) * p.lda + GBM) * 2) : 0x7fffffffu;
    const unsigned b_bytes = BCOL ? (unsigned)(((size_t)(kend - kbeg - 1) * p.ldb + GBN) * 2) : 0x7fffffffu;
    auto rsrc_a = [&](int m0_) {
        return __builtin_amdgcn_make_buffer_rsrc(
            (void *)(ACOL ? sp.A + (size_t)kbeg * p.lda + m0_ : sp.A + (size_t)m0_ * p.lda + kbeg), 0, (int)a_bytes, 0x00020000);
    };
    auto rsrc_b = [&](int n0_) {
        return __builtin_amdgcn_make_buffer_rsrc(
            (void *)(BCOL ? sp.B + (size_t)kbeg * p.ldb + n0_ : sp.B + (size_t)n0_ * p.ldb + kbeg), 0, (int)b_bytes, 0x00020000);
    };
    __amdgpu_buffer_rsrc_t ra = rsrc_a(m0), rb = rsrc_b(n0);
    static_assert(!SEG2 || (!ACOL && !BCOL && !PERSIST), "two operand pairs: NT form, one tile per workgroup");
    const int nk1 = SEG2 ? sp.k1 / GBK : nk;
    const __amdgpu_buffer_rsrc_t ra2 = SEG2 ? __builtin_amdgcn_make_buffer_rsrc((void *)(sp.A2 + (size_t)m0 * p.lda), 0,
                                                                                0x7fffffff, 0x00020000) : ra;
    const __amdgpu_buffer_rsrc_t rb2 = SEG2 ? __builtin_amdgcn_make_buffer_rsrc((void *)(sp.B2 + (size_t)n0 * p.ldb), 0,
                                                                                0x7fffffff, 0x00020000) : rb;
    // fill: wave w moves pieces 4 w .. 4 w + 3 (1 KB each) of each operand.  k-contiguous: lane l of piece c fills LDS
    // granule l of the piece = row 8 c + l / 8, slot l % 8, with the row's k-octet (l % 8) ^ ((row >> 1) & 7).  k-major:
    // a piece is two k-rows; lane l fills k-row 2 c + l / 32, slot l % 32, with the row's 16-byte piece (l % 32) ^ 4 (k & 3).
    int voa[4], vob[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + (lane >> 3), oct = (lane & 7) ^ ((row >> 1) & 7);
        const int krow = (wave * 4 + i) * 2 + (lane >> 5), kcol = ((lane & 31) ^ ((krow & 3) << 2)) * 8;
        voa[i] = ACOL ? (krow * p.lda + kcol) * 2 : (row * p.lda + oct * 8) * 2;
        vob[i] = BCOL ? (krow * p.ldb + kcol) * 2 : (row * p.ldb + oct * 8) * 2;
    }
    const int kstep_a = ACOL ? GBK * p.lda * 2 : GBK * 2, kstep_b = BCOL ? GBK * p.ldb * 2 : GBK * 2;   // bytes per k tile
#define LC_GFILL_FROM(RA, RB, KT, BUF)                                                                                 \
    {                                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                  \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(RA, lds + (BUF) * 2 * G_OPERAND_BYTES + (wave * 4 + i) * 1024, 16, \
                                                     voa[i], (KT) * kstep_a, 0, 0);                                    \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                  \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(RB, lds + ((BUF) * 2 + 1) * G_OPERAND_BYTES + (wave * 4 + i) * 1024, \
                                                     16, vob[i], (KT) * kstep_b, 0, 0);                                \
    }
#define LC_GFILL(KT, BUF)                                                                                              \
    {                                                                                                                  \
        const int kt_ = (KT);                                                                                          \
        if (SEG2 && kt_ >= nk1) LC_GFILL_FROM(ra2, rb2, kt_ - nk1, BUF) else LC_GFILL_FROM(ra, rb, kt_, BUF)           \
    }
    f32x16 acc[4][2];

    const int lr = lane & 31, lk = lane >> 5;
    // fragment (row, k-octet 2 q + lk) sits in slot (2 q + lk) ^ ((row >> 1) & 7); rows of a wave's fragments differ by
    // multiples of 32, so the swizzle term is the lane's own ((lr >> 1) & 7) for all of them
    const int fl = (lr >> 1) & 7;
    int so[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) so[q] = ((2 * q + lk) ^ fl) * 16;
    const int arow = (wm * 128 + lr) * 128, brow = (wn * 64 + lr) * 128;
    // k-major fragment addresses: 16-lane group g = lane / 16 covers columns 16 (g & 1) .. + 15 of a 32-column MFMA tile and
    // k-octet g / 2; within it lane (r = l / 4, c = l % 4) hands in k-row r, 8-byte piece c.  Tile t's 64 bytes of k-row k
    // sit in pieces 4 (t ^ (k & 3)) .. + 3 (the fill's swizzle; t = tile number within the 256 columns), and k & 3 = r for
    // every fragment of the walk (all other k terms are multiples of 4).
    const int tg = lane >> 4, tr = (lane & 15) >> 2, tc = lane & 3;
    static_assert(!(ACOL && !BCOL), "built forms: NT (both k-contiguous), TN (both K-major), NN (A k-contiguous, B K-major)");
    const int tbase = ((tg >> 1) * 8 + tr) * 512 + ((tg & 1) * 2 + (tc >> 1)) * 16 + (tc & 1) * 8;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds;
    unsigned toa[4], tob[2];                    // LDS byte addresses of this lane's fragments in buffer 0
#pragma unroll
    for (int i = 0; i < 4; ++i) toa[i] = lds0 + tbase + (((wm * 4 + i) ^ tr) * 4) * 16;
#pragma unroll
    for (int j = 0; j < 2; ++j) tob[j] = lds0 + G_OPERAND_BYTES + tbase + (((wn * 2 + j) ^ tr) * 4) * 16;
#define LC_GFRAG(PTR) (*reinterpret_cast<const bf16x8 *>(PTR))
#define LC_GFRAG_A(I, Q) LC_GFRAG(as + arow + (I) * 4096 + so[Q])
#define LC_GFRAG_B(J, Q) LC_GFRAG(bs + brow + (J) * 4096 + so[Q])
    // K-major form: k tile in buffer BUF; the fragments of k-step Q + 1 are requested before the MFMAs of k-step Q
#define LC_TLOAD(Q, FA, FB)                                                                                            \
    {                                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) tr16_pair<(Q) * 8192>(toa[i] + tbuf, FA[i][0], FA[i][1]);        \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) tr16_pair<(Q) * 8192>(tob[j] + tbuf, FB[j][0], FB[j][1]);        \
    }
#define LC_TWAIT(FA, FB)                                                                                               \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                                \
                 : "+v"(FA[0][0]), "+v"(FA[0][1]), "+v"(FA[1][0]), "+v"(FA[1][1]), "+v"(FA[2][0]), "+v"(FA[2][1]),      \
                   "+v"(FA[3][0]), "+v"(FA[3][1]), "+v"(FB[0][0]), "+v"(FB[0][1]), "+v"(FB[1][0]), "+v"(FB[1][1]));
#define LC_TMMA(FA, FB)                                                                                                \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                      \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                  \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr16_join(FA[i][0], FA[i][1]), tr16_join(FB[j][0], FB[j][1]), \
                                                               acc[i][j], 0, 0, 0);
#define LC_TCOMPUTE(BUF)                                                                                               \
    {                                                                                                                  \
        const unsigned tbuf = (BUF) * 2 * G_OPERAND_BYTES;                                                             \
        i32x2 f0a[4][2], f0b[2][2], f1a[4][2], f1b[2][2];                                                              \
        LC_TLOAD(0, f0a, f0b) LC_TWAIT(f0a, f0b)                                                                       \
        LC_TLOAD(1, f1a, f1b) LC_TMMA(f0a, f0b) LC_TWAIT(f1a, f1b)                                                     \
        LC_TLOAD(2, f0a, f0b) LC_TMMA(f1a, f1b) LC_TWAIT(f0a, f0b)                                                     \
        LC_TLOAD(3, f1a, f1b) LC_TMMA(f0a, f0b) LC_TWAIT(f1a, f1b)                                                     \
        LC_TMMA(f1a, f1b)                                                                                              \
    }
    // NN form (A k-contiguous, B K-major: forward products on the natural shadows of activations AND weights): A's
    // fragments by ds_read_b128, B's by the transposing reads - all from asm, one wait discipline
    unsigned maa[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) maa[q] = lds0 + arow + so[q];
#define LC_MLOAD(Q, FA, FB)                                                                                            \
    {                                                                                                                  \
        lds_read128<0>(maa[Q] + tbuf, FA[0]); lds_read128<4096>(maa[Q] + tbuf, FA[1]);                                 \
        lds_read128<8192>(maa[Q] + tbuf, FA[2]); lds_read128<12288>(maa[Q] + tbuf, FA[3]);                             \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) tr16_pair<(Q) * 8192>(tob[j] + tbuf, FB[j][0], FB[j][1]);        \
    }
#define LC_MWAIT(FA, FB)                                                                                               \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                                \
                 : "+v"(FA[0]), "+v"(FA[1]), "+v"(FA[2]), "+v"(FA[3]), "+v"(FB[0][0]), "+v"(FB[0][1]), "+v"(FB[1][0]),  \
                   "+v"(FB[1][1]));
#define LC_MMMA(FA, FB)                                                                                                \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                      \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                  \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cast(FA[i]), tr16_join(FB[j][0], FB[j][1]),        \
                                                               acc[i][j], 0, 0, 0);
#define LC_MCOMPUTE(BUF)                                                                                               \
    {                                                                                                                  \
        const unsigned tbuf = (BUF) * 2 * G_OPERAND_BYTES;                                                             \
        i32x4g m0a[4], m1a[4];                                                                                         \
        i32x2 m0b[2][2], m1b[2][2];                                                                                    \
        LC_MLOAD(0, m0a, m0b) LC_MWAIT(m0a, m0b)                                                                       \
        LC_MLOAD(1, m1a, m1b) LC_MMMA(m0a, m0b) LC_MWAIT(m1a, m1b)                                                     \
        LC_MLOAD(2, m0a, m0b) LC_MMMA(m1a, m1b) LC_MWAIT(m0a, m0b)                                                     \
        LC_MLOAD(3, m1a, m1b) LC_MMMA(m0a, m0b) LC_MWAIT(m1a, m1b)                                                     \
        LC_MMMA(m1a, m1b)                                                                                              \
    }
#define LC_GCOMPUTE(BUF)                                                                                               \
    {                                                                                                                  \
        const unsigned char *as = lds + (BUF) * 2 * G_OPERAND_BYTES;                                                   \
        const unsigned char *bs = lds + ((BUF) * 2 + 1) * G_OPERAND_BYTES;                                             \
        bf16x8 a[4], b[2];                                                                                             \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) a[i] = LC_GFRAG_A(i, 0);                                         \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) b[j] = LC_GFRAG_B(j, 0);                                         \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                                \
            bf16x8 na[4], nb[2];                                                                                       \
            if (q + 1 < 4) {                                                                                           \
                _Pragma("unroll") for (int i = 0; i < 4; ++i) na[i] = LC_GFRAG_A(i, (q + 1) & 3);                      \
                _Pragma("unroll") for (int j = 0; j < 2; ++j) nb[j] = LC_GFRAG_B(j, (q + 1) & 3);                      \
            }                                                                                                          \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                              \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                          \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);               \
            if (q + 1 < 4) {                                                                                           \
                _Pragma("unroll") for (int i = 0; i < 4; ++i) a[i] = na[i];                                            \
                _Pragma("unroll") for (int j = 0; j < 2; ++j) b[j] = nb[j];                                            \
            }                                                                                                          \
        }                                                                                                              \
    }
    // (NT form: the compiler drains the LDS-DMA requests in front of a barrier because its own ds_reads follow; the K-major
    // form's reads are asm, so the drain is written out)
#define LC_GSYNC()                                                                                                     \
    {                                                                                                                  \
        if constexpr (ACOL || BCOL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                   \
        __syncthreads();                                                                                               \
    }
#define LC_GDO(BUF)                                                                                                    \
    {                                                                                                                  \
        if constexpr (ACOL) LC_TCOMPUTE(BUF) else if constexpr (BCOL) LC_MCOMPUTE(BUF) else LC_GCOMPUTE(BUF)            \
    }
    const int ntiles = (p.M / GBM) * (p.N / GBN);
    LC_GFILL(0, 0)
    LC_GSYNC()
    for (int vb = blockIdx.x;;) {                 // PERSIST: this workgroup's tiles; otherwise one pass
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    int nm0 = m0, nn0 = n0;
    bool more = false;
    __amdgpu_buffer_rsrc_t ra_n = ra, rb_n = rb;
    if constexpr (PERSIST) {
        const int vn = vb + (int)gridDim.x;
        more = vn < ntiles;
        if (more) {
            int bm_, bn_;
            gemm_tile_order_big(p.M, p.N, bm_, bn_, vn);
            nm0 = bm_ * GBM; nn0 = bn_ * GBN;
            ra_n = rsrc_a(nm0); rb_n = rsrc_b(nn0);
        }
    }
    int kt = 0;
    for (; kt + 2 <= nk; kt += 2) {
        LC_GFILL(min(kt + 1, nk - 1), 1)
        LC_GDO(0)
        LC_GSYNC()
        if (PERSIST && kt + 2 >= nk) {            // (nk even: launch condition) k tile 0 of the next tile, or a harmless refill
            LC_GFILL_FROM(ra_n, rb_n, 0, 0)
        } else {
            LC_GFILL(min(kt + 2, nk - 1), 0)
        }
        LC_GDO(1)
        LC_GSYNC()
    }
    if (kt < nk) LC_GDO(0)
#undef LC_GCOMPUTE
#undef LC_GFRAG
#undef LC_GFRAG_A
#undef LC_GFRAG_B
#undef LC_TLOAD
#undef LC_TWAIT
#undef LC_TMMA
#undef LC_TCOMPUTE
#undef LC_MLOAD
#undef LC_MWAIT
#undef LC_MMMA
#undef LC_MCOMPUTE
    if (p.slab) {
        float *S = p.slab + (size_t)blockIdx.z * p.slab_slice;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = n0 + wn * 64 + j * 32 + lr;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                    S[(size_t)row * p.slab_ld + col] = acc[i][j][r];
                }
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + lr;
            const float bv = p.bias ? p.bias[col] : 0.f;
            unsigned est;
            int ecm;
            epi_column(p.epi, col, est, ecm);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                float *c = p.C + (size_t)row * p.ldc + col;
                float v = p.alpha * acc[i][j][r] + bv;
                if (p.beta != 0.f) v += p.beta * *c;
                epi_store(p.epi, c, v, row, col, est, ecm);
            }
        }
    if constexpr (!PERSIST) break;
    if (!more) break;
    vb += (int)gridDim.x;
    m0 = nm0; n0 = nn0; ra = ra_n; rb = rb_n;
    }
#undef LC_GSYNC
#undef LC_GDO
#undef LC_GFILL
#undef LC_GFILL_FROM
}

// ---- f32 products on 256 x 256 x 32 tiles with LDS-DMA operands ------------------------------------------------------
// The 128 x 128 f32 kernel above sits at 0.86-0.90 of the MFMA peak, and tools/ubench/mfma_agpr_rate.hip shows why nothing
// more comes out of its schedule: f32 MFMAs do not overlap with VALU or vector-memory issue on a SIMD (a VALU instruction
// behind an MFMA costs its full ~5 cycles, a global load ~15, from the same wave or from another one) - every staging
// instruction of a tile is paid in matrix-pipe time.  So this kernel stages LESS per MFMA: a 256 x 256 tile moves half
// the operand bytes per flop of a 128 x 128 one, and the operands go from global memory into LDS by DMA
// (`buffer_load_dwordx4 ... lds`, 1 KB per wave-instruction: no staging registers, no ds_write pass, no address VALU in
// the loop - 8 vector-memory instructions per wave and 32-deep k tile for 128 MFMAs of 64 cycles).
//   * 8 waves, 2 (M) x 4 (N), wave tile 128 x 64 = 4 x 2 tiles of v_mfma_f32_32x32x2_f32 (128 accumulator registers);
//   * an operand whose k index is contiguous in memory (A of NN / NT, B of NT) is staged like the bf16 kernel's: 256 rows
//     of 128 bytes, 16-byte granule slot s of row r holding k-quad s ^ ((r >> 1) & 7); a lane reads the quad 2q + lk
//     of its row with one ds_read_b128 and feeds four MFMAs from it - MFMA t of group q then multiplies the k pair
//     {8q + t, 8q + 4 + t}: a permutation of the k walk that both operands follow;
//   * an operand stored k-major (A of TN, B of NN / TN) is staged as 32 rows of 1 KB (one wave-instruction per k row,
//     linear image) and read with ds_read_b32 at row 8q + 4 lk + t: lanes = consecutive columns, conflict-free;
//   * two LDS buffers (128 KB), tile t+1 requested before tile t is multiplied, one barrier per k tile.
// Launched on whole tiles only (M, N multiples of 256, K chunks multiples of 32) when tiles x slices fill >= 90 % of whole
// rounds of 256 CUs.
constexpr int FGBK = 32;
template <bool ACOL, bool BCOL, bool SEG2 = false>
__global__ __launch_bounds__(GNT, 1) void gemm_f32g_kernel(GemmArgs p)
{
    static_assert(!SEG2 || (!ACOL && !BCOL), "two operand pairs: NT form");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[4 * 256 * FGBK * 4];      // A0 B0 A1 B1, 32 KB each
    constexpr int OPB = 256 * FGBK * 4;
    int bm, bn;
    gemm_tile_order_big(p.M, p.N, bm, bn);
    const int m0 = bm * GBM, n0 = bn * GBN;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int kbeg = blockIdx.z * p.kchunk;
    const int kend = min(p.K, kbeg + p.kchunk);
    const int nk = (kend - kbeg) / FGBK;
    const float *abase = ACOL ? p.A + (size_t)kbeg * p.lda + m0 : p.A + (size_t)m0 * p.lda + kbeg;
    const float *bbase = BCOL ? p.B + (size_t)kbeg * p.ldb + n0 : p.B + (size_t)n0 * p.ldb + kbeg;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void *)abase, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void *)bbase, 0, 0x7fffffff, 0x00020000);
    const int nk1 = SEG2 ? p.k1 / FGBK : nk;
    const __amdgpu_buffer_rsrc_t ra2 = SEG2 ? __builtin_amdgcn_make_buffer_rsrc((void *)(p.A2 + (size_t)m0 * p.lda), 0,
                                                                                0x7fffffff, 0x00020000) : ra;
    const __amdgpu_buffer_rsrc_t rb2 = SEG2 ? __builtin_amdgcn_make_buffer_rsrc((void *)(p.B2 + (size_t)n0 * p.ldb), 0,
                                                                                0x7fffffff, 0x00020000) : rb;
    // fill: wave w moves pieces 4 w .. 4 w + 3 (1 KB each) of each operand's tile
    int voa[4], vob[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int piece = wave * 4 + i;
        if (ACOL) voa[i] = (piece * p.lda + 4 * lane) * 4;                      // k row `piece`, columns 4 l .. 4 l + 3
        else { const int row = piece * 8 + (lane >> 3); voa[i] = (row * p.lda + ((lane & 7) ^ ((row >> 1) & 7)) * 4) * 4; }
        if (BCOL) vob[i] = (piece * p.ldb + 4 * lane) * 4;
        else { const int row = piece * 8 + (lane >> 3); vob[i] = (row * p.ldb + ((lane & 7) ^ ((row >> 1) & 7)) * 4) * 4; }
    }
    const int ka = ACOL ? FGBK * p.lda * 4 : FGBK * 4, kb = BCOL ? FGBK * p.ldb * 4 : FGBK * 4;   // bytes per k tile
#define LC_FFILL_FROM(RA, RB, KT, BUF)                                                                                 \
    {                                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                  \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(RA, lds + (BUF) * 2 * OPB + (wave * 4 + i) * 1024, 16, voa[i],     \
                                                     (KT) * ka, 0, 0);                                                 \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                  \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(RB, lds + ((BUF) * 2 + 1) * OPB + (wave * 4 + i) * 1024, 16,       \
                                                     vob[i], (KT) * kb, 0, 0);                                         \
    }
#define LC_FFILL(KT, BUF)                                                                                              \
    {                                                                                                                  \
        const int kt_ = (KT);                                                                                          \
        if (SEG2 && kt_ >= nk1) LC_FFILL_FROM(ra2, rb2, kt_ - nk1, BUF) else LC_FFILL_FROM(ra, rb, kt_, BUF)           \
    }
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int lr = lane & 31, lk = lane >> 5;
    const int fl = (lr >> 1) & 7;
    // per-lane byte offsets of the fragments inside an operand tile; the q / t / tile parts are immediates
    int soq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) soq[q] = ((2 * q + lk) ^ fl) * 16;
    const int arow = ACOL ? (4 * lk * 256 + wm * 128 + lr) * 4 : (wm * 128 + lr) * 128;
    const int brow = BCOL ? (4 * lk * 256 + wn * 64 + lr) * 4 : (wn * 64 + lr) * 128;
#define LC_FCOMPUTE(BUF)                                                                                               \
    {                                                                                                                  \
        const unsigned char *as = lds + (BUF) * 2 * OPB + arow;                                                        \
        const unsigned char *bs = lds + ((BUF) * 2 + 1) * OPB + brow;                                                  \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                                \
            float a[4][4], b[2][4];                                                                                    \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                            \
                if (ACOL) {                                                                                            \
                    _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                      \
                        a[i][t] = *reinterpret_cast<const float *>(as + ((8 * q + t) * 256 + i * 32) * 4);             \
                } else {                                                                                               \
                    const f32x4v v = *reinterpret_cast<const f32x4v *>(as + i * 4096 + soq[q]);                        \
                    a[i][0] = v.x; a[i][1] = v.y; a[i][2] = v.z; a[i][3] = v.w;                                        \
                }                                                                                                      \
            }                                                                                                          \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                            \
                if (BCOL) {                                                                                            \
                    _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                      \
                        b[j][t] = *reinterpret_cast<const float *>(bs + ((8 * q + t) * 256 + j * 32) * 4);             \
                } else {                                                                                               \
                    const f32x4v v = *reinterpret_cast<const f32x4v *>(bs + j * 4096 + soq[q]);                        \
                    b[j][0] = v.x; b[j][1] = v.y; b[j][2] = v.z; b[j][3] = v.w;                                        \
                }                                                                                                      \
            }                                                                                                          \
            _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                              \
                _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                          \
                    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                      \
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][t], b[j][t], acc[i][j], 0, 0, 0);        \
        }                                                                                                              \
    }
    LC_FFILL(0, 0)
    __syncthreads();                            // (the compiler drains the LDS-DMA requests in front of a barrier)
    int kt = 0;
    for (; kt + 2 <= nk; kt += 2) {
        LC_FFILL(min(kt + 1, nk - 1), 1)
        LC_FCOMPUTE(0)
        __syncthreads();
        LC_FFILL(min(kt + 2, nk - 1), 0)
        LC_FCOMPUTE(1)
        __syncthreads();
    }
    if (kt < nk) LC_FCOMPUTE(0)
#undef LC_FFILL
#undef LC_FFILL_FROM
#undef LC_FCOMPUTE
    if (p.slab) {
        float *S = p.slab + (size_t)blockIdx.z * p.slab_slice;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = n0 + wn * 64 + j * 32 + lr;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                    S[(size_t)row * p.slab_ld + col] = acc[i][j][r];
                }
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + lr;
            const float bv = p.bias ? p.bias[col] : 0.f;
            unsigned est;
            int ecm;
            epi_column(p.epi, col, est, ecm);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                float *c = p.C + (size_t)row * p.ldc + col;
                float v = p.alpha * acc[i][j][r] + bv;
                if (p.beta != 0.f) v += p.beta * *c;
                epi_store(p.epi, c, v, row, col, est, ecm);
            }
        }
}

// fp32 [rows, C] -> bf16 copies: nat[rows][ldnat] (same orientation) and / or tr[C][ldtr] (transposed), either may be
// NULL.  64 x 64 tiles through LDS so that both outputs are written in 128-byte runs.
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float *__restrict__ x, int rows, int C, int ldx,
                                                        unsigned short *__restrict__ nat, int ldnat,
                                                        unsigned short *__restrict__ tr, int ldtr)
{
    __shared__ unsigned short tile[64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;          // 4 row phases
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        unsigned short v = 0;
        if (r < rows && c < C) {
            v = __builtin_bit_cast(unsigned short, (__bf16)x[(size_t)r * ldx + c]);
            if (nat) nat[(size_t)r * ldnat + c] = v;
        }
        tile[i][tx] = v;
    }
    if (!tr) return;
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < C && r < rows) tr[(size_t)c * ldtr + r] = tile[tx][i];
    }
}

// The same on whole 64 x 64 tiles with 16-byte accesses (rows, C multiples of 64; ldx % 4 == 0, ldnat % 4 == 0, ldtr % 8 == 0,
// aligned pointers): a thread converts a 2 x 4 patch (two rows, one float4 each), stores the same-orientation copy as 8
// bytes per row, and packs the two rows of each column into one dword of the transposed tile in LDS ([c][row pair], 128
// bytes per column, cell (c, rp) at slot rp ^ (((c >> 2) & 7) << 2): 2-way on the dword writes - free - and conflict-free
// for the 16-byte reads); the transposed copy leaves as 16 bytes (8 rows) per thread.  The element-wise kernel above moves
// 2-4 bytes per lane and instruction and reaches 3.7 TB/s; this one is the c5 path (every shadow of a [T*B, *] tensor).
__device__ __forceinline__ unsigned cast_pack2(float lo, float hi)
{
    return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)lo) |
           ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)hi) << 16);
}
template <bool NAT, bool TR>
__global__ __launch_bounds__(256) void cast_bf16_vec_kernel(const float *__restrict__ x, int ldx,
                                                            unsigned short *__restrict__ nat, int ldnat,
                                                            unsigned short *__restrict__ tr, int ldtr)
{
    __shared__ __attribute__((aligned(16))) unsigned tile[64 * 32];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int cq = threadIdx.x & 15, rp = threadIdx.x >> 4;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rpair = rp + 16 * i;
        const size_t r = (size_t)r0 + 2 * rpair;
        const int c = c0 + 4 * cq;
        const float4 v0 = *reinterpret_cast<const float4 *>(x + r * ldx + c);
        const float4 v1 = *reinterpret_cast<const float4 *>(x + (r + 1) * ldx + c);
        if constexpr (NAT) {
            *reinterpret_cast<uint2 *>(nat + r * ldnat + c) = make_uint2(cast_pack2(v0.x, v0.y), cast_pack2(v0.z, v0.w));
            *reinterpret_cast<uint2 *>(nat + (r + 1) * ldnat + c) = make_uint2(cast_pack2(v1.x, v1.y), cast_pack2(v1.z, v1.w));
        }
        if constexpr (TR) {
            const int slot = rpair ^ ((cq & 7) << 2);          // (c >> 2) & 7 = cq & 7 for the four columns 4 cq + j
            tile[(4 * cq + 0) * 32 + slot] = cast_pack2(v0.x, v1.x);
            tile[(4 * cq + 1) * 32 + slot] = cast_pack2(v0.y, v1.y);
            tile[(4 * cq + 2) * 32 + slot] = cast_pack2(v0.z, v1.z);
            tile[(4 * cq + 3) * 32 + slot] = cast_pack2(v0.w, v1.w);
        }
    }
    if constexpr (TR) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = (threadIdx.x >> 3) + 32 * i, g = threadIdx.x & 7;
            const uint4 v = *reinterpret_cast<const uint4 *>(&tile[c * 32 + ((4 * g) ^ (((c >> 2) & 7) << 2))]);
            *reinterpret_cast<uint4 *>(tr + (size_t)(c0 + c) * ldtr + r0 + 8 * g) = v;
        }
    }
}

// C = alpha * sum_s slab[s] + beta*C + bias
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float *__restrict__ slab, int nslices, int M, int N,
                                                            float alpha, float beta, float *__restrict__ C, int ldc,
                                                            const float *__restrict__ bias)
{
    const size_t total = (size_t)M * N / 4;     // N % 4 == 0 guaranteed by the caller
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t e = i * 4;
        const int row = (int)(e / N), col = (int)(e % N);
        float4 s = *reinterpret_cast<const float4 *>(slab + e);
        for (int k = 1; k < nslices; ++k) {
            const float4 t = *reinterpret_cast<const float4 *>(slab + (size_t)k * M * N + e);
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        float *c = C + (size_t)row * ldc + col;
        float o[4] = {alpha * s.x, alpha * s.y, alpha * s.z, alpha * s.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (bias) o[q] += bias[col + q];
            if (beta != 0.f) o[q] += beta * c[q];
            c[q] = o[q];
        }
    }
}

inline bool aligned16(const void *p) { return (((uintptr_t)p) & 15) == 0; }

// Number of K slices for tall-K products with few output tiles (the weight gradients, K = T*B).  The f32 kernel runs
// 3 workgroups per CU (768 slots on the chip) and needs all three to keep the MFMA pipe ~94 % busy (PMC: 89 % with
// two); so pick the slice count that makes tiles * slices fill whole rounds of 768 with the least waste, slices
// at least 1024 deep, at most 32.  (512 tiles -> 3 slices = two full rounds; 256 -> 3; 64 -> 12; 30 -> 25.)
inline int pick_splitk(int M, int N, int K)
{
    const long long tiles = (long long)lc_cdiv(M, BM) * lc_cdiv(N, BN);
    if (tiles >= 1536 || K < 4096 || (N % 4) != 0) return 1;
    const long long slots = 768;
    int best = 1;
    double best_eff = 0.0;
    for (int s = 1; s <= 32 && (long long)K / s >= 1024; ++s) {
        const long long wg = tiles * s, rounds = (wg + slots - 1) / slots;
        const double eff = (double)wg / (double)(rounds * slots);
        if (eff > best_eff + 1e-9) { best_eff = eff; best = s; }
    }
    return best;
}

// The same for the 256 x 256 kernel (one workgroup per CU: rounds of 256); 0 = the shape is not eligible for it.
inline int pick_splitk_big(int M, int N, int K, int bk = GBK)
{
    if (M <= 0 || N <= 0 || M % GBM || N % GBN || K < bk || K % bk) return 0;
    const long long tiles = (long long)(M / GBM) * (N / GBN);
    if (tiles >= 512 || K < 4096) return 1;
    int best = 1;
    double best_eff = 0.0;
    for (int c = 1; c <= 32 && K / c >= 1024; ++c) {
        const long long wg = tiles * c, rounds = (wg + 255) / 256;
        const double eff = (double)wg / (double)(rounds * 256);
        if (eff > best_eff + 1e-9) { best_eff = eff; best = c; }
    }
    return best;
}

}  // namespace

// Compute units of the current device (cached per device index): the grid of the persistent tile walk.
static int lc_num_cus()
{
    static int cached[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
    if (cached[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached[dev] = n;
    }
    return cached[dev];
}
// Tail of an unsplit product on the 256 x 256 kernels (one workgroup per CU): `row_tiles x col_tiles` tiles take
// ceil(tiles / CUs) rounds, and the last round of the big products is 44 - 81 % full (c4 zx: 250 x 16 = 15.6 rounds, dX: 250 x 8 =
// 7.8; c2 zx: 125 x 5 = 2.44).  Returns how many row tiles the big kernel should take so that its rounds are (>= 97 %) WHOLE;
// the rows behind them go to the 128 x 128 kernel (two workgroups per CU, quarter-size tiles: the tail's work spreads over the
// whole chip instead of idling part of it for a full tile time).  Cost model in big-kernel rounds: a quarter tile is 0.25 x
// 1.1 of a round per CU (+ 0.1 for the second launch); the split must win by 0.05 rounds.
static int gemm_whole_round_row_tiles(int row_tiles, int col_tiles, int cus_override = 0)
{
    const int cus = cus_override > 0 ? cus_override : lc_num_cus();
    // Default OFF (LC_GEMM_TAIL=1 / option gemm_tail switches it on): measured +0.3 ... 0.9 % on c4 and +2 % on c2
    // (profiles/r6_gemm_tail_ab.txt), but the rows behind the cut are summed in another order than the rows in front of it, so
    // an utterance's arithmetic would depend on the batch it sits in - which the full-size parity properties rely on
    // (tests/test_gpu_model.py::test_full_size_c4_gradient_additivity: at the reference's initialisation a T = 1000 recurrence
    // amplifies that last-bit difference to per cent).  Not worth it.
    if (lc_option(LC_OPT_GEMM_TAIL, 0) == 0 || row_tiles <= 1 || col_tiles <= 0) return row_tiles;
    const long long tiles = (long long)row_tiles * col_tiles;
    if (tiles % cus == 0 || tiles < 2ll * cus) return row_tiles;
    double best_t = (double)((tiles + cus - 1) / cus) - 0.05;
    int best = row_tiles;
    for (int keep = row_tiles - 1, it = 0; keep >= 1 && it < 512; --keep, ++it) {
        const long long big = (long long)keep * col_tiles, rem = big % cus;
        if (big < cus) break;
        if (rem != 0 && rem * 100 < (long long)cus * 97) continue;
        const double t = (double)((big + cus - 1) / cus) + (double)(row_tiles - keep) * col_tiles * 4 * 0.275 / cus + 0.1;
        if (t < best_t) { best_t = t; best = keep; }
    }
    return best;
}
extern "C" int lc_debug_gemm_whole_round_row_tiles(int row_tiles, int col_tiles, int cus)
{
    return gemm_whole_round_row_tiles(row_tiles, col_tiles, cus);
}
// gemm_bf16g_kernel's persistent tile walk: unsplit products, an even number of k tiles, more tiles than CUs, a CU count
// the XCD-aware tile order stays valid for (a multiple of 8: workgroup b's tiles b, b + grid, ... stay on XCD b % 8).
static bool bf16g_persist_ok(long long tiles, int nsl, int K)
{
    const int cus = lc_num_cus();
    // default OFF: measured in round 6 (profiles/r6_gemm_persist_ab.txt) the walk is worth +5 % on the zx NT shape and -1 ... -9 %
    // on the others, -0.5 ms on a c5 step: the per-tile cost it hides (launch, set-up, first fill) is not what a tile loses -
    // the C stores of 256 CUs arriving together are.  LC_GEMM_BF16_PERSIST=1 switches it on.
    return lc_option(LC_OPT_GEMM_BF16_PERSIST, 0) != 0 && nsl <= 1 && K % (2 * GBK) == 0 && cus % 8 == 0 && tiles > cus;
}

// ---- the one-shot fused epilogue (lstm_ctc_hip.h: lc_gemm_next_epilogue) -----------------------------------------------
static const EpiArgs EPI_NONE = {1.f, 1.f, 0u, 0u, 1, nullptr, 0, 0, 0};
static thread_local EpiArgs g_epi_next = EPI_NONE;
EpiArgs lc_epi_take()                          // every lc_gemm_* entry consumes the pending epilogue, whatever happens next
{
    const EpiArgs e = g_epi_next;
    g_epi_next = EPI_NONE;
    return e;
}
static EpiArgs epi_take() { return lc_epi_take(); }
extern "C" int lc_gemm_next_epilogue(const lc_gemm_epilogue_t *e)
{
    if (!e) { g_epi_next = EPI_NONE; return LC_OK; }
    LC_CHECK_ARG(e->keep > 0.f && e->keep <= 1.f && (e->keep >= 1.f || e->drop_width > 0),
                 "lc_gemm_next_epilogue: keep must be in (0, 1] and drop_width > 0");
    LC_CHECK_ARG(!e->c_bf16 || e->ldc_bf16 > 0, "lc_gemm_next_epilogue: ldc_bf16 must be positive");
    EpiArgs a = EPI_NONE;
    a.keep = e->keep; a.inv_keep = 1.0f / e->keep; a.seed = e->seed; a.stream0 = e->stream0;
    a.P = e->keep < 1.f ? e->drop_width : 1;
    a.c16 = (unsigned short *)e->c_bf16; a.ldc16 = e->ldc_bf16;
    LC_CHECK_ARG(!e->shadow_only || e->c_bf16, "lc_gemm_next_epilogue: shadow_only needs c_bf16");
    a.skip_c = e->shadow_only ? 1 : 0;
    g_epi_next = a;
    return LC_OK;
}

extern "C" size_t lc_gemm_workspace_bytes(int M, int N, int K)
{
    const int s = std::max(pick_splitk(M, N, K), std::max(pick_splitk_big(M, N, K), pick_splitk_big(M, N, K, FGBK)));
    return s > 1 ? (size_t)s * M * N * sizeof(float) : 0;
}

// One kernel launch over the sub-problem whose origin is (A, B, C, bias, slab) as given.
static void gemm_launch_part(bool bf16, bool fast, int ta, int tb, GemmArgs p, int nsl, hipStream_t s)
{
    const long long nwg = (long long)lc_cdiv(p.M, BM) * lc_cdiv(p.N, BN);
    dim3 grid((unsigned)nwg, 1, (unsigned)(nsl > 1 ? nsl : 1)), block(NT);
#define LC_GEMM(TA, TB)                                                                          \
    do {                                                                                         \
        if (bf16) {                                                                              \
            if (fast) hipLaunchKernelGGL((gemm_bf16_kernel<TA, TB, true>), grid, block, 0, s, p);    \
            else hipLaunchKernelGGL((gemm_bf16_kernel<TA, TB, false>), grid, block, 0, s, p);        \
        } else {                                                                                 \
            if (fast) hipLaunchKernelGGL((gemm_f32_kernel<TA, TB, true>), grid, block, 0, s, p);     \
            else hipLaunchKernelGGL((gemm_f32_kernel<TA, TB, false>), grid, block, 0, s, p);         \
        }                                                                                        \
    } while (0)
    if (!ta && !tb) LC_GEMM(false, false);
    else if (ta && !tb) LC_GEMM(true, false);
    else if (!ta && tb) LC_GEMM(false, true);
    else LC_GEMM(true, true);
#undef LC_GEMM
}

static int gemm_launch(bool bf16, const char *who, int ta, int tb, int M, int N, int K, float alpha, const float *A,
                       int lda, const float *B, int ldb, float beta, float *C, int ldc, const float *bias,
                       void *workspace, size_t workspace_bytes, lc_stream_t stream)
{
    const EpiArgs epi = epi_take();
    LC_CHECK_ARG(A && B && C, "%s: null pointer", who);
    LC_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, "%s: negative dimension", who);
    if (M == 0 || N == 0) return LC_OK;
    LC_CHECK_ARG(lda >= (ta ? M : K) && ldb >= (tb ? K : N) && ldc >= N, "%s: leading dimension too small", who);
    hipStream_t s = (hipStream_t)stream;
    const int bk = bf16 ? HBK : BK;
    if (epi_active(epi)) { workspace = nullptr; workspace_bytes = 0; }      // a fused epilogue lives in the product kernel: no K split
    GemmArgs p;
    p.epi = epi;
    p.A2 = p.B2 = nullptr; p.k1 = 0;
    p.M = M; p.N = N; p.K = K; p.alpha = alpha; p.beta = beta;
    p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.C = C; p.ldc = ldc; p.bias = bias;
    p.vecA = aligned16(A) && (lda % 4 == 0);
    p.vecB = aligned16(B) && (ldb % 4 == 0);
    LC_CHECK_ARG((long long)lc_cdiv(M, BM) * lc_cdiv(N, BN) < (1ll << 31), "%s: grid too large", who);
    // whole 256 x 256 tiles, 32-deep K chunks, enough of them for half the chip: the LDS-DMA kernel
    // (LC_GEMM_F32_BIG: 0 = never, 2 = whenever the shape is eligible - the tests use it on small shapes -, default: when
    // tiles x slices cover half the chip)
    const int big_mode = (int)lc_option(LC_OPT_GEMM_F32_BIG, 1);
    if (!bf16 && big_mode != 0 && p.vecA && p.vecB && K >= FGBK && K % FGBK == 0) {
        // interior of whole 256 x 256 tiles on the big kernel, ragged right / bottom edges (T * B is a multiple of 256 only
        // for every fourth T at B = 64) as strips on the 128 x 128 kernel; K is split only for exact shapes (the tall-K
        // weight gradients, whose M and N are layer widths)
        int Mi = M / GBM * GBM;
        const int Ni = N / GBN * GBN;
        const bool exact = Mi == M && Ni == N;
        int nb = exact ? pick_splitk_big(M, N, K, FGBK) : 1;
        if (nb > 1 && (!workspace || workspace_bytes < (size_t)nb * M * N * sizeof(float))) nb = 1;
        if (nb <= 1) Mi = gemm_whole_round_row_tiles(Mi / GBM, Ni / GBN) * GBM;      // whole rounds; the rest as a bottom strip
        const long long tiles = (long long)(Mi / GBM) * (Ni / GBN);
        const long long spanA = ta ? (long long)K * lda * 4 + 4ll * GBM : (long long)(GBM - 1) * lda * 4 + 4ll * K;
        const long long spanB = tb ? (long long)(GBN - 1) * ldb * 4 + 4ll * K : (long long)K * ldb * 4 + 4ll * GBN;
        // one workgroup per CU: the last round of 256 must be nearly full (32000 x 1280 = 625 tiles fill 2.44 rounds and run
        // 11 % slower here than on the 128 x 128 kernel's 768 slots)
        const long long wgs = tiles * (nb > 1 ? nb : 1), rounds = (wgs + 255) / 256;
        const bool fills = wgs >= 128 && wgs * 10 >= rounds * 256 * 9;
        if (tiles > 0 && (fills || big_mode == 2) && spanA < 0x7fffffffll && spanB < 0x7fffffffll) {
            GemmArgs q = p;
            q.M = Mi; q.N = Ni;
            q.kchunk = nb > 1 ? lc_cdiv(lc_cdiv(K, nb), FGBK) * FGBK : K;
            if (nb > 1) nb = lc_cdiv(K, q.kchunk);
            q.slab = nb > 1 ? (float *)workspace : nullptr;
            q.slab_slice = (size_t)M * N;
            q.slab_ld = N;
            const dim3 grid((unsigned)tiles, 1, (unsigned)(nb > 1 ? nb : 1)), block(GNT);
            if (ta && !tb) hipLaunchKernelGGL((gemm_f32g_kernel<true, true>), grid, block, 0, s, q);
            else if (ta) hipLaunchKernelGGL((gemm_f32g_kernel<true, false>), grid, block, 0, s, q);
            else if (!tb) hipLaunchKernelGGL((gemm_f32g_kernel<false, true>), grid, block, 0, s, q);
            else hipLaunchKernelGGL((gemm_f32g_kernel<false, false>), grid, block, 0, s, q);
            p.kchunk = K > 0 ? K : 1; p.slab = nullptr; p.slab_slice = 0; p.slab_ld = N;      // the strips: unsplit
            if (Ni < N) {                               // right strip: all M rows, columns [Ni, N)
                q = p;
                q.N = N - Ni;
                q.B = tb ? B + (size_t)Ni * ldb : B + Ni;
                q.C = C + Ni;
                q.epi = epi_block(p.epi, 0, Ni);
                q.bias = bias ? bias + Ni : nullptr;
                q.vecB = aligned16(q.B) && (ldb % 4 == 0);
                gemm_launch_part(false, false, ta, tb, q, 1, s);
            }
            if (Mi < M) {                               // bottom strip: rows [Mi, M), columns [0, Ni)
                q = p;
                q.M = M - Mi; q.N = Ni;
                q.A = ta ? A + Mi : A + (size_t)Mi * lda;
                q.C = C + (size_t)Mi * ldc;
                q.epi = epi_block(p.epi, Mi, 0);
                q.vecA = aligned16(q.A) && (lda % 4 == 0);
                // (a tail of whole 128 x 128 tiles - the rows cut off for whole rounds - takes the kernel without bounds checks)
                const bool fast = q.M % BM == 0 && q.N % BN == 0 && K % BK == 0 && q.vecA && q.vecB;
                gemm_launch_part(false, fast, ta, tb, q, 1, s);
            }
            LC_CHECK_LAUNCH(who);
            if (nb > 1) {
                const size_t quads = (size_t)M * N / 4;
                int g = (int)((quads + 255) / 256);
                if (g > 2048) g = 2048;
                hipLaunchKernelGGL(splitk_reduce_kernel, dim3(g), dim3(256), 0, s, (float *)workspace, nb, M, N, alpha, beta, C, ldc, bias);
                LC_CHECK_LAUNCH("splitk_reduce");
            }
            return LC_OK;
        }
    }
    int nsl = pick_splitk(M, N, K);
    if (nsl > 1 && (!workspace || workspace_bytes < (size_t)nsl * M * N * sizeof(float))) nsl = 1;   // no slab: unsplit
    p.kchunk = nsl > 1 ? lc_cdiv(lc_cdiv(K, nsl), bk) * bk : (K > 0 ? K : 1);
    if (nsl > 1) nsl = lc_cdiv(K, p.kchunk);
    p.slab = nsl > 1 ? (float *)workspace : nullptr;
    p.slab_slice = (size_t)M * N;
    p.slab_ld = N;
    // The no-bounds-check kernel needs whole 128x128 tiles; a ragged M or N edge (the MoE head's N = E*V, a last
    // partial batch) is peeled off into one or two strips for the generic kernel instead of demoting the whole
    // product to it.
    const bool kfast = (K % bk == 0) && K > 0 && p.vecA && p.vecB && (p.kchunk % bk == 0);
    const int Mi = M / BM * BM, Ni = N / BN * BN;
    // Peeling only pays when the interior launch fills the chip on its own: three back-to-back launches of a handful of
    // workgroups each (a 320 x 320 weight gradient: 128 + 96 + 64) took 359 us where one generic launch takes ~120.
    const bool ragged = Mi < M || Ni < N;
    const long long interior_wgs = (long long)(Mi / BM) * (Ni / BN) * (nsl > 1 ? nsl : 1);
    if (kfast && Mi > 0 && Ni > 0 && !(ragged && interior_wgs < 256)) {
        GemmArgs q = p;
        q.M = Mi; q.N = Ni;
        gemm_launch_part(bf16, true, ta, tb, q, nsl, s);
        if (Ni < N) {                                   // right strip: all M rows, columns [Ni, N)
            q = p;
            q.N = N - Ni;
            q.B = tb ? B + (size_t)Ni * ldb : B + Ni;
            q.C = C + Ni;
            q.epi = epi_block(p.epi, 0, Ni);
            q.bias = bias ? bias + Ni : nullptr;
            q.vecB = aligned16(q.B) && (ldb % 4 == 0);
            if (q.slab) q.slab += Ni;
            gemm_launch_part(bf16, false, ta, tb, q, nsl, s);
        }
        if (Mi < M) {                                   // bottom strip: rows [Mi, M), columns [0, Ni)
            q = p;
            q.M = M - Mi; q.N = Ni;
            q.A = ta ? A + Mi : A + (size_t)Mi * lda;
            q.C = C + (size_t)Mi * ldc;
            q.epi = epi_block(p.epi, Mi, 0);
            q.vecA = aligned16(q.A) && (lda % 4 == 0);
            if (q.slab) q.slab += (size_t)Mi * N;
            gemm_launch_part(bf16, false, ta, tb, q, nsl, s);
        }
    } else {
        gemm_launch_part(bf16, false, ta, tb, p, nsl, s);
    }
    LC_CHECK_LAUNCH(who);
    if (nsl > 1) {
        const size_t quads = (size_t)M * N / 4;
        int g = (int)((quads + 255) / 256);
        if (g > 2048) g = 2048;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(g), dim3(256), 0, s, p.slab, nsl, M, N, alpha, beta, C, ldc, bias);
        LC_CHECK_LAUNCH("splitk_reduce");
    }
    return LC_OK;
}

extern "C" int lc_gemm_f32(int ta, int tb, int M, int N, int K, float alpha, const float *A, int lda,
                           const float *B, int ldb, float beta, float *C, int ldc, const float *bias,
                           void *workspace, size_t workspace_bytes, lc_stream_t stream)
{
    return gemm_launch(false, "lc_gemm_f32", ta, tb, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, workspace,
                       workspace_bytes, stream);
}

// C = alpha * (A1 B1^T + A2 B2^T) + beta C + bias in one pass over C, float32 (see lc_gemm_bf16_nt2): the dX of a
// bidirectional layer.  Whole 256 x 256 tiles: gemm_f32g_kernel<false, false, true> walks K1 then K2 on one accumulator; ragged
// edges and shapes the big kernel does not take: the two products in sequence (beta / bias with the first, the fused epilogue -
// it finishes the FINAL value - with the second).
extern "C" int lc_gemm_f32_nt2(int M, int N, int K1, int K2, float alpha, const float *A1, const float *A2, int lda,
                               const float *B1, const float *B2, int ldb, float beta, float *C, int ldc, const float *bias,
                               lc_stream_t stream)
{
    const EpiArgs epi = epi_take();
    const char *who = "lc_gemm_f32_nt2";
    LC_CHECK_ARG(A1 && A2 && B1 && B2 && C, "%s: null pointer", who);
    LC_CHECK_ARG(M >= 0 && N >= 0 && K1 > 0 && K2 > 0, "%s: bad dimension", who);
    if (M == 0 || N == 0) return LC_OK;
    LC_CHECK_ARG(lda >= std::max(K1, K2) && ldb >= std::max(K1, K2) && ldc >= N, "%s: leading dimension too small", who);
    hipStream_t s = (hipStream_t)stream;
    auto sequence = [&](int m, int n, const float *a1, const float *a2, const float *b1, const float *b2, float *c,
                        const float *bi, const EpiArgs &e) -> int {
        int rc = gemm_launch(false, who, 0, 1, m, n, K1, alpha, a1, lda, b1, ldb, beta, c, ldc, bi, nullptr, 0, stream);
        if (rc != LC_OK) return rc;
        g_epi_next = e;                          // (taken by the call below)
        return gemm_launch(false, who, 0, 1, m, n, K2, alpha, a2, lda, b2, ldb, 1.f, c, ldc, nullptr, nullptr, 0, stream);
    };
    const int Mi = M / GBM * GBM, Ni = N / GBN * GBN;
    const long long tiles = (long long)(Mi / GBM) * (Ni / GBN), rounds = (tiles + 255) / 256;
    const int big_mode = (int)lc_option(LC_OPT_GEMM_F32_BIG, 1);
    const bool vec = aligned16(A1) && aligned16(A2) && aligned16(B1) && aligned16(B2) && lda % 4 == 0 && ldb % 4 == 0;
    const bool fills = tiles >= 128 && tiles * 10 >= rounds * 256 * 9;
    const long long span = (long long)(GBM - 1) * std::max(lda, ldb) * 4 + 4ll * std::max(K1, K2);
    if (!(big_mode != 0 && vec && tiles > 0 && (fills || big_mode == 2) && K1 % FGBK == 0 && K2 % FGBK == 0 &&
          span < 0x7fffffffll))
        return sequence(M, N, A1, A2, B1, B2, C, bias, epi);
    GemmArgs q;
    q.epi = epi;
    q.M = Mi; q.N = Ni; q.K = K1 + K2; q.alpha = alpha; q.beta = beta;
    q.A = A1; q.lda = lda; q.B = B1; q.ldb = ldb; q.C = C; q.ldc = ldc; q.bias = bias;
    q.vecA = q.vecB = 1;
    q.kchunk = K1 + K2; q.slab = nullptr; q.slab_slice = 0; q.slab_ld = N;
    q.A2 = A2; q.B2 = B2; q.k1 = K1;
    hipLaunchKernelGGL((gemm_f32g_kernel<false, false, true>), dim3((unsigned)tiles, 1, 1), dim3(GNT), 0, s, q);
    LC_CHECK_LAUNCH(who);
    if (Ni < N) {                                   // right strip: all M rows, columns [Ni, N)
        const int rc = sequence(M, N - Ni, A1, A2, B1 + (size_t)Ni * ldb, B2 + (size_t)Ni * ldb, C + Ni,
                                bias ? bias + Ni : nullptr, epi_block(epi, 0, Ni));
        if (rc != LC_OK) return rc;
    }
    if (Mi < M) {                                   // bottom strip: rows [Mi, M), columns [0, Ni)
        const int rc = sequence(M - Mi, Ni, A1 + (size_t)Mi * lda, A2 + (size_t)Mi * lda, B1, B2, C + (size_t)Mi * ldc, bias,
                                epi_block(epi, Mi, 0));
        if (rc != LC_OK) return rc;
    }
    return LC_OK;
}
extern "C" int lc_gemm_bf16(int ta, int tb, int M, int N, int K, float alpha, const float *A, int lda,
                            const float *B, int ldb, float beta, float *C, int ldc, const float *bias,
                            void *workspace, size_t workspace_bytes, lc_stream_t stream)
{
    return gemm_launch(true, "lc_gemm_bf16", ta, tb, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, workspace,
                       workspace_bytes, stream);
}

extern "C" int lc_cast_bf16(const float *x, int rows, int C, int ldx, uint16_t *nat, int ldnat, uint16_t *tr, int ldtr,
                            lc_stream_t stream)
{
    LC_CHECK_ARG(x && (nat || tr) && rows >= 0 && C > 0 && ldx >= C, "lc_cast_bf16: bad argument");
    LC_CHECK_ARG((!nat || ldnat >= C) && (!tr || ldtr >= rows), "lc_cast_bf16: leading dimension too small");
    if (rows == 0) return LC_OK;
    if (rows % 64 == 0 && C % 64 == 0 && ldx % 4 == 0 && aligned16(x) && (!nat || (ldnat % 4 == 0 && (((uintptr_t)nat) & 7) == 0)) &&
        (!tr || (ldtr % 8 == 0 && aligned16(tr)))) {
        const dim3 grid(C / 64, rows / 64), block(256);
        hipStream_t s = (hipStream_t)stream;
        if (nat && tr) hipLaunchKernelGGL((cast_bf16_vec_kernel<true, true>), grid, block, 0, s, x, ldx, nat, ldnat, tr, ldtr);
        else if (nat) hipLaunchKernelGGL((cast_bf16_vec_kernel<true, false>), grid, block, 0, s, x, ldx, nat, ldnat, tr, ldtr);
        else hipLaunchKernelGGL((cast_bf16_vec_kernel<false, true>), grid, block, 0, s, x, ldx, nat, ldnat, tr, ldtr);
        LC_CHECK_LAUNCH("cast_bf16 (64 x 64 tiles)");
        return LC_OK;
    }
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(lc_cdiv(C, 64), lc_cdiv(rows, 64)), dim3(256), 0, (hipStream_t)stream, x, rows,
                       C, ldx, nat, ldnat, tr, ldtr);
    LC_CHECK_LAUNCH("cast_bf16");
    return LC_OK;
}

// A2 / B2 / K2: optional second operand pair whose product is accumulated into the same C (lc_gemm_bf16_nt2); the epilogue
// `epi` has been taken by the caller.
static int gemm_bf16_nt_impl(const EpiArgs &epi, int M, int N, int K, float alpha, const uint16_t *A, int lda, const uint16_t *B,
                             int ldb, float beta, float *C, int ldc, const float *bias, void *workspace,
                             size_t workspace_bytes, lc_stream_t stream, const uint16_t *A2 = nullptr,
                             const uint16_t *B2 = nullptr, int K2 = 0)
{
    if (epi_active(epi)) { workspace = nullptr; workspace_bytes = 0; }      // fused epilogue: no K split
    LC_CHECK_ARG(A && B && C, "lc_gemm_bf16_nt: null pointer");
    LC_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, "lc_gemm_bf16_nt: negative dimension");
    if (M == 0 || N == 0) return LC_OK;
    LC_CHECK_ARG(lda >= K && ldb >= K && ldc >= N, "lc_gemm_bf16_nt: leading dimension too small");
    LC_CHECK_ARG(K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && aligned16(A) && aligned16(B),
                 "lc_gemm_bf16_nt: K, lda, ldb must be multiples of 8 and the operands 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    SGemmArgs sp;
    GemmArgs &p = sp.g;
    p.M = M; p.N = N; p.K = K; p.alpha = alpha; p.beta = beta;
    p.A = nullptr; p.lda = lda; p.B = nullptr; p.ldb = ldb; p.C = C; p.ldc = ldc; p.bias = bias;
    p.vecA = p.vecB = 1;
    p.epi = epi;
    sp.A = A; sp.B = B;
    LC_CHECK_ARG((long long)lc_cdiv(M, BM) * lc_cdiv(N, BN) < (1ll << 31), "lc_gemm_bf16_nt: grid too large");
    // whole 256 x 256 tiles and 64-deep K chunks: the LDS-DMA kernel, one workgroup per CU; ragged right / bottom edges (T * B
    // is a multiple of 256 only for every fourth T at B = 64) go to the 128 x 128 kernel as strips; K is split only for
    // exact shapes (the tall-K weight gradients)
    sp.A2 = sp.B2 = nullptr; sp.k1 = 0;
    const bool big_off = lc_option(LC_OPT_GEMM_BF16_BIG, 1) == 0;
    int Mb = M / GBM * GBM;
    const int Nb = N / GBN * GBN;
    const bool seg2 = A2 != nullptr;
    if (!seg2 && !(Mb == M && Nb == N && pick_splitk_big(M, N, K) > 1))   // unsplit: whole rounds, the rest as a bottom strip
        Mb = gemm_whole_round_row_tiles(Mb / GBM, Nb / GBN) * GBM;
    const bool big_ok = !big_off && Mb > 0 && Nb > 0 && K >= GBK && K % GBK == 0 &&
        (long long)(GBM - 1) * lda * 2 + 2ll * std::max(K, K2) < 0x7fffffffll &&
        (long long)(GBN - 1) * ldb * 2 + 2ll * std::max(K, K2) < 0x7fffffffll && (!seg2 || (K2 >= GBK && K2 % GBK == 0));
    if (seg2 && !big_ok) {
        // no whole 256 x 256 tiles (or the big kernel is switched off): the two products one after the other, the second
        // accumulating - beta and bias with the first, the fused epilogue (it masks the FINAL value) with the second
        int rc = gemm_bf16_nt_impl(EPI_NONE, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, workspace, workspace_bytes, stream);
        if (rc != LC_OK) return rc;
        return gemm_bf16_nt_impl(epi, M, N, K2, alpha, A2, lda, B2, ldb, 1.f, C, ldc, nullptr, nullptr, 0, stream);
    }
    if (big_ok) {
        const long long tiles = (long long)(Mb / GBM) * (Nb / GBN);
        int nsl = (Mb == M && Nb == N && !seg2) ? pick_splitk_big(M, N, K) : 1;      // tall-K weight gradients: fill whole rounds of 256 CUs
        if (nsl > 1 && (!workspace || workspace_bytes < (size_t)nsl * M * N * sizeof(float))) nsl = 1;
        SGemmArgs q = sp;
        q.g.M = Mb; q.g.N = Nb;
        q.g.kchunk = nsl > 1 ? lc_cdiv(lc_cdiv(K, nsl), GBK) * GBK : K;
        if (nsl > 1) nsl = lc_cdiv(K, q.g.kchunk);
        q.g.slab = nsl > 1 ? (float *)workspace : nullptr;
        q.g.slab_slice = (size_t)M * N;
        q.g.slab_ld = N;
        if (seg2) {
            q.A2 = A2; q.B2 = B2; q.k1 = K;
            q.g.K = K + K2; q.g.kchunk = K + K2;
            hipLaunchKernelGGL((gemm_bf16g_kernel<false, false, false, true>), dim3((unsigned)tiles, 1, 1), dim3(GNT), 0, s, q);
        } else if (bf16g_persist_ok(tiles, nsl, K))
            hipLaunchKernelGGL((gemm_bf16g_kernel<false, false, true>), dim3((unsigned)lc_num_cus(), 1, 1), dim3(GNT), 0, s, q);
        else
            hipLaunchKernelGGL((gemm_bf16g_kernel<false, false>), dim3((unsigned)tiles, 1, (unsigned)(nsl > 1 ? nsl : 1)), dim3(GNT), 0, s, q);
        p.kchunk = K; p.slab = nullptr; p.slab_slice = 0; p.slab_ld = N;       // the strips: unsplit, bounds-checked kernel
        auto strip = [&](const SGemmArgs &r0) {
            const long long nwg = (long long)lc_cdiv(r0.g.M, BM) * lc_cdiv(r0.g.N, BN);
            if (!seg2) {
                // (a tail of whole 128 x 128 tiles - the rows cut off for whole rounds - takes the kernel without bounds checks)
                if (r0.g.M % BM == 0 && r0.g.N % BN == 0 && K % SBK == 0)
                    hipLaunchKernelGGL((gemm_bf16s_kernel<true>), dim3((unsigned)nwg, 1, 1), dim3(NT), 0, s, r0);
                else
                    hipLaunchKernelGGL((gemm_bf16s_kernel<false>), dim3((unsigned)nwg, 1, 1), dim3(NT), 0, s, r0);
                return;
            }
            // two operand pairs on a strip: the first product (beta, bias, no epilogue), then the second accumulating into it
            SGemmArgs r = r0;
            const EpiArgs e = r.g.epi;
            r.g.epi = EPI_NONE;
            hipLaunchKernelGGL((gemm_bf16s_kernel<false>), dim3((unsigned)nwg, 1, 1), dim3(NT), 0, s, r);
            r.A = A2 + (r0.A - A); r.B = B2 + (r0.B - B);
            r.g.K = K2; r.g.kchunk = K2; r.g.beta = 1.f; r.g.bias = nullptr; r.g.epi = e;
            hipLaunchKernelGGL((gemm_bf16s_kernel<false>), dim3((unsigned)nwg, 1, 1), dim3(NT), 0, s, r);
        };
        if (Nb < N) {                                   // right strip: all M rows, columns [Nb, N)
            q = sp;
            q.g.N = N - Nb;
            q.B = B + (size_t)Nb * ldb;
            q.g.C = C + Nb;
            q.g.epi = epi_block(p.epi, 0, Nb);
            q.g.bias = bias ? bias + Nb : nullptr;
            strip(q);
        }
        if (Mb < M) {                                   // bottom strip: rows [Mb, M), columns [0, Nb)
            q = sp;
            q.g.M = M - Mb; q.g.N = Nb;
            q.A = A + (size_t)Mb * lda;
            q.g.C = C + (size_t)Mb * ldc;
            q.g.epi = epi_block(p.epi, Mb, 0);
            strip(q);
        }
        LC_CHECK_LAUNCH("lc_gemm_bf16_nt (256 x 256 tiles)");
        if (nsl > 1) {
            const size_t quads = (size_t)M * N / 4;
            int g = (int)((quads + 255) / 256);
            if (g > 2048) g = 2048;
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3(g), dim3(256), 0, s, (float *)workspace, nsl, M, N, alpha, beta, C, ldc, bias);
            LC_CHECK_LAUNCH("splitk_reduce");
        }
        return LC_OK;
    }
    int nsl = pick_splitk(M, N, K);
    if (nsl > 1 && (!workspace || workspace_bytes < (size_t)nsl * M * N * sizeof(float))) nsl = 1;
    p.kchunk = nsl > 1 ? lc_cdiv(lc_cdiv(K, nsl), SBK) * SBK : (K > 0 ? K : 1);
    if (nsl > 1) nsl = lc_cdiv(K, p.kchunk);
    p.slab = nsl > 1 ? (float *)workspace : nullptr;
    p.slab_slice = (size_t)M * N;
    p.slab_ld = N;
    auto launch = [&](bool fast, const SGemmArgs &q) {
        const long long nwg = (long long)lc_cdiv(q.g.M, BM) * lc_cdiv(q.g.N, BN);
        dim3 grid((unsigned)nwg, 1, (unsigned)(nsl > 1 ? nsl : 1)), block(NT);
        if (fast) hipLaunchKernelGGL((gemm_bf16s_kernel<true>), grid, block, 0, s, q);
        else hipLaunchKernelGGL((gemm_bf16s_kernel<false>), grid, block, 0, s, q);
    };
    const bool kfast = K > 0 && (K % SBK == 0) && (p.kchunk % SBK == 0);
    const int Mi = M / BM * BM, Ni = N / BN * BN;
    if (kfast && Mi > 0 && Ni > 0) {                 // interior without bounds checks, ragged edges as strips
        SGemmArgs q = sp;
        q.g.M = Mi; q.g.N = Ni;
        launch(true, q);
        if (Ni < N) {
            q = sp;
            q.g.N = N - Ni;
            q.B = B + (size_t)Ni * ldb;
            q.g.C = C + Ni;
            q.g.epi = epi_block(p.epi, 0, Ni);
            q.g.bias = bias ? bias + Ni : nullptr;
            if (q.g.slab) q.g.slab += Ni;
            launch(false, q);
        }
        if (Mi < M) {
            q = sp;
            q.g.M = M - Mi; q.g.N = Ni;
            q.A = A + (size_t)Mi * lda;
            q.g.C = C + (size_t)Mi * ldc;
            q.g.epi = epi_block(p.epi, Mi, 0);
            if (q.g.slab) q.g.slab += (size_t)Mi * N;
            launch(false, q);
        }
    } else {
        launch(false, sp);
    }
    LC_CHECK_LAUNCH("lc_gemm_bf16_nt");
    if (nsl > 1) {
        const size_t quads = (size_t)M * N / 4;
        int g = (int)((quads + 255) / 256);
        if (g > 2048) g = 2048;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(g), dim3(256), 0, s, p.slab, nsl, M, N, alpha, beta, C, ldc, bias);
        LC_CHECK_LAUNCH("splitk_reduce");
    }
    return LC_OK;
}

extern "C" int lc_gemm_bf16_nt(int M, int N, int K, float alpha, const uint16_t *A, int lda, const uint16_t *B, int ldb,
                               float beta, float *C, int ldc, const float *bias, void *workspace,
                               size_t workspace_bytes, lc_stream_t stream)
{
    const EpiArgs epi = epi_take();
    return gemm_bf16_nt_impl(epi, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, workspace, workspace_bytes, stream);
}
// C = alpha * (A1 B1^T + A2 B2^T) + beta C + bias in ONE pass over C: the input gradient of a bidirectional layer,
// dz_fwd . Kx_fwd^T + dz_bwd . Kx_bwd^T (both cells read the same concatenated input, nnet/bilstm.py:190-203).
extern "C" int lc_gemm_bf16_nt2(int M, int N, int K1, int K2, float alpha, const uint16_t *A1, const uint16_t *A2, int lda,
                                const uint16_t *B1, const uint16_t *B2, int ldb, float beta, float *C, int ldc,
                                const float *bias, lc_stream_t stream)
{
    const EpiArgs epi = epi_take();
    LC_CHECK_ARG(A2 && B2, "lc_gemm_bf16_nt2: null pointer");
    LC_CHECK_ARG(K1 > 0 && K2 > 0 && K2 % 8 == 0 && lda >= K2 && ldb >= K2 && aligned16(A2) && aligned16(B2),
                 "lc_gemm_bf16_nt2: K1, K2 > 0, K2 a multiple of 8 within the leading dimensions, operands 16-byte aligned");
    return gemm_bf16_nt_impl(epi, M, N, K1, alpha, A1, lda, B1, ldb, beta, C, ldc, bias, nullptr, 0, stream, A2, B2, K2);
}

// C[M,N] = alpha * A^T B (+ beta C + bias) with BOTH bf16 operands K-MAJOR: A stored [K][M], B stored [K][N] - the weight
// gradients of a train step (X^T dZ, hs^T dZ, hs^T dY) on the NATURAL bf16 shadows of the activations, no transposed
// copies.  gemm_bf16g_kernel<true, true>: whole 256 x 256 tiles only (M, N multiples of 256; the callers' M and N are layer
// widths), any K >= 1 (the K tail is zero-filled by the buffer descriptor), split along K like lc_gemm_bf16_nt.
static int gemm_bf16_kmajor(bool acol, const char *who, int M, int N, int K, float alpha, const uint16_t *A, int lda,
                            const uint16_t *B, int ldb, float beta, float *C, int ldc, const float *bias, void *workspace,
                            size_t workspace_bytes, lc_stream_t stream)
{
    const EpiArgs epi = epi_take();
    if (epi_active(epi)) { workspace = nullptr; workspace_bytes = 0; }      // fused epilogue: no K split
    LC_CHECK_ARG(A && B && C, "%s: null pointer", who);
    LC_CHECK_ARG(M > 0 && N > 0 && K > 0, "%s: empty product", who);
    LC_CHECK_ARG(M % GBM == 0 && N % GBN == 0, "%s: M and N must be multiples of 256 (got %d x %d)", who, M, N);
    LC_CHECK_ARG(lda >= (acol ? M : K) && ldb >= N && ldc >= N, "%s: leading dimension too small", who);
    LC_CHECK_ARG(acol || K % GBK == 0, "%s: K must be a multiple of 64 (the k-contiguous operand has no zero-filled tail)", who);
    LC_CHECK_ARG(lda % 8 == 0 && ldb % 8 == 0 && aligned16(A) && aligned16(B),
                 "%s: lda, ldb must be multiples of 8 and the operands 16-byte aligned", who);
    hipStream_t s = (hipStream_t)stream;
    SGemmArgs sp;
    GemmArgs &p = sp.g;
    p.M = M; p.N = N; p.K = K; p.alpha = alpha; p.beta = beta;
    p.A = nullptr; p.lda = lda; p.B = nullptr; p.ldb = ldb; p.C = C; p.ldc = ldc; p.bias = bias;
    p.vecA = p.vecB = 1;
    p.epi = epi;
    sp.A = A; sp.B = B;
    int nsl = pick_splitk_big(M, N, K);
    if (nsl > 1 && (!workspace || workspace_bytes < (size_t)nsl * M * N * sizeof(float))) nsl = 1;
    p.kchunk = nsl > 1 ? lc_cdiv(lc_cdiv(K, nsl), GBK) * GBK : K;
    if (nsl > 1) nsl = lc_cdiv(K, p.kchunk);
    // 32-bit byte offsets inside a K chunk: (k tile) * 64 rows * ld * 2 bytes (K-major operands); rows * ld (k-contiguous)
    LC_CHECK_ARG((long long)p.kchunk * (acol && lda > ldb ? lda : ldb) * 2 < 0x7fffffffll &&
                     (acol || (long long)(GBM - 1) * lda * 2 + 2ll * K < 0x7fffffffll), "%s: operand too large", who);
    p.slab = nsl > 1 ? (float *)workspace : nullptr;
    p.slab_slice = (size_t)M * N;
    p.slab_ld = N;
    const long long tiles = (long long)(M / GBM) * (N / GBN);
    const dim3 grid((unsigned)tiles, 1, (unsigned)(nsl > 1 ? nsl : 1));
    const bool persist = bf16g_persist_ok(tiles, nsl, K);
    const dim3 pgrid((unsigned)lc_num_cus(), 1, 1);
    if (acol) {
        if (persist) hipLaunchKernelGGL((gemm_bf16g_kernel<true, true, true>), pgrid, dim3(GNT), 0, s, sp);
        else hipLaunchKernelGGL((gemm_bf16g_kernel<true, true>), grid, dim3(GNT), 0, s, sp);
    } else {
        if (persist) hipLaunchKernelGGL((gemm_bf16g_kernel<false, true, true>), pgrid, dim3(GNT), 0, s, sp);
        else hipLaunchKernelGGL((gemm_bf16g_kernel<false, true>), grid, dim3(GNT), 0, s, sp);
    }
    LC_CHECK_LAUNCH(who);
    if (nsl > 1) {
        const size_t quads = (size_t)M * N / 4;
        int g = (int)((quads + 255) / 256);
        if (g > 2048) g = 2048;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(g), dim3(256), 0, s, (float *)workspace, nsl, M, N, alpha, beta, C, ldc, bias);
        LC_CHECK_LAUNCH("splitk_reduce");
    }
    return LC_OK;
}

extern "C" int lc_gemm_bf16_tn(int M, int N, int K, float alpha, const uint16_t *A, int lda, const uint16_t *B, int ldb,
                               float beta, float *C, int ldc, const float *bias, void *workspace,
                               size_t workspace_bytes, lc_stream_t stream)
{
    return gemm_bf16_kmajor(true, "lc_gemm_bf16_tn", M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, workspace,
                            workspace_bytes, stream);
}
// C[M,N] = alpha * A B (+ beta C + bias): A k-contiguous [M][K], B K-MAJOR [K][N] - the forward products X . Kx and hs . proj
// on the natural shadows of the activation AND of the weight (no transposed weight copies).  M, N multiples of 256, K of 64.
extern "C" int lc_gemm_bf16_nn(int M, int N, int K, float alpha, const uint16_t *A, int lda, const uint16_t *B, int ldb,
                               float beta, float *C, int ldc, const float *bias, void *workspace,
                               size_t workspace_bytes, lc_stream_t stream)
{
    return gemm_bf16_kmajor(false, "lc_gemm_bf16_nn", M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, workspace,
                            workspace_bytes, stream);
}
