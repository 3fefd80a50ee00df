// gemm_x3.hip - fp32 products on the bf16 matrix cores: every fp32 operand is split EXACTLY into three bf16 terms
// (a = a_h + a_m + a_l: 3 x 8 significand bits = the 24 of an fp32) and the product is accumulated in fp32 from the six
// term pairs whose weight is at least 2^-16 of the leading one,
//     a b ~= a_h b_h + a_h b_m + a_m b_h + a_h b_l + a_l b_h + a_m b_m        (dropped: a_m b_l + a_l b_m + a_l b_l <= 2^-25 |a b|),
// each bf16 x bf16 product being exact in fp32.  What is lost against an fp32 multiply is a quarter of an fp32 ulp per
// product - below the rounding of the fp32 accumulation both forms share (tests/test_gpu_ops.py measures both against
// float64).  On gfx950 one v_mfma_f32_32x32x16_bf16 does 16 k-steps in 32 cycles, one v_mfma_f32_32x32x2_f32 2 k-steps in 64
// (MI355X_MICROARCH.md: 2.5 PFLOP/s bf16 against 157 TFLOP/s fp32): 16 k of an fp32 product are six bf16 MFMAs = 192
// cycles here and eight fp32 MFMAs = 512 cycles there - 2.67 x less matrix-pipe time -, and the
// operand traffic per MFMA is HALF of the plain bf16 kernel's (six products share the three + three term tiles).  What
// the kernels then run at is the power limit of a dense bf16 MFMA stream (~1.3 PFLOP/s: DESIGN.md section 3f), i.e.
// ~1.55 x the fp32 kernels.
//
// Operand format ("x3 shadow", written by lc_split_bf16x3): row-major, k in tiles of 16: row r holds, for k tile t, 48
// bf16 = [hi 16 | mid 16 | lo 16] at element offset 48 t - so one k tile of one row is 96 contiguous bytes, and K is
// padded with zeros to a multiple of 16.  Both operands k-contiguous (NT form): C = alpha A B^T + beta C + bias.
//
// Kernel: 256 x 256 tile, 8 waves (2 x 4), wave tile 128 x 64 = 4 x 2 MFMA tiles x 6 term pairs = 48 MFMAs per 16-deep k
// tile; operands go global -> LDS by DMA (`buffer_load_dwordx4 ... lds`), LDS image per stage and operand
// [row][term][32 bytes] = the memory image of the k tile, the two 16-byte halves of a term swapped for rows with bit 3 set
// (source-side swizzle: the 16-lane groups of a ds_read_b128 are then conflict-free); three stages of 48 KB: the fill of tile t + 2 is
// in flight while tile t is multiplied (a stage takes about as long to arrive as to multiply).  Ragged M / N: the buffer
// descriptors end with the matrix, so rows past M / N arrive as zeros, and the epilogue masks them.
#include "common.h"
#include "gemm_epi.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int XBM = 256, XBN = 256, XNT = 512;          // k tile: 16
constexpr int X_OPER = 256 * 96;            // one operand tile: 256 rows x 3 terms x 16 bf16 = 24 KB
constexpr int X_STAGE = 2 * X_OPER;         // 48 KB

struct X3Args {
    EpiArgs epi;
    int M, N, nk;                           // nk = k tiles of 16
    float alpha, beta;
    const unsigned short *A; int lda;       // x3 shadows, leading dimensions in bf16 elements (>= 48 nk)
    const unsigned short *B; int ldb;
    float *C; int ldc;
    const float *bias;
    int K, kchunk;                          // TN form: reduction rows, rows per blockIdx.z slice (multiple of 16)
    float *slab;                            // TN form, split along K: [gridDim.z][M][N] partial products, else nullptr
};

__device__ __forceinline__ void x3_tile_order(int M, int N, int &bm, int &bn)
{
    const int nbm = (M + XBM - 1) / XBM, nbn = (N + XBN - 1) / XBN;
    const int nwg = nbm * nbn;
    int bid = blockIdx.x;
    {   // consecutive tiles on one XCD (workgroups are dealt round-robin to the 8 XCDs)
        const int q = nwg / 8, rr = nwg % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
    }
    constexpr int GROUP_M = 4;              // 4 x nbn patches: A panels shared out of one L2
    const int gsz = GROUP_M * nbn;
    const int first_m = (bid / gsz) * GROUP_M;
    const int gm = min(nbm - first_m, GROUP_M);
    bm = first_m + (bid % gsz) % gm;
    bn = (bid % gsz) / gm;
}

typedef int i32x4g __attribute__((ext_vector_type(4)));
template <int OFF>
__device__ __forceinline__ void x3_read(unsigned lds_addr, i32x4g &v)
{
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(v) : "v"(lds_addr), "n"(OFF));
}
__device__ __forceinline__ bf16x8 x3_frag(i32x4g v)
{
    union { i32x4g i; bf16x8 b; } u;
    u.i = v;
    return u.b;
}

__global__ __launch_bounds__(XNT, 1) void gemm_x3_kernel(X3Args p)
{
    __shared__ __attribute__((aligned(1024))) unsigned char lds[3 * X_STAGE];
    int bm, bn;
    x3_tile_order(p.M, p.N, bm, bn);
    const int m0 = bm * XBM, n0 = bn * XBN;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const long long a_left = (long long)(p.M - m0) * p.lda * 2, b_left = (long long)(p.N - n0) * p.ldb * 2;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(p.A + (size_t)m0 * p.lda), 0, (int)(a_left < 0x7fffffffll ? a_left : 0x7fffffffll), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(p.B + (size_t)n0 * p.ldb), 0, (int)(b_left < 0x7fffffffll ? b_left : 0x7fffffffll), 0x00020000);
    // fill: an operand stage is 256 rows x 96 bytes = 1536 granules of 16 bytes = 24 pieces of 1 KB; wave w moves pieces
    // 3 w .. 3 w + 2 of each operand.  Lane l of piece q fills LDS granule g = 64 q + l = (row g / 6, term (g % 6) / 2, half
    // g % 2) with the row's k-half (g % 2) ^ ((row >> 3) & 1) of that term: the six lanes of a row read its 96 contiguous
    // bytes of the k tile (one or two cache lines per row and instruction).
    int voa[3], vob[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int g = (wave * 3 + c) * 64 + lane, row = g / 6, idx = g - row * 6;
        const int term = idx >> 1, half = (idx & 1) ^ ((row >> 3) & 1);
        voa[c] = (row * p.lda + term * 16 + half * 8) * 2;
        vob[c] = (row * p.ldb + term * 16 + half * 8) * 2;
    }
#define LC_XFILL(KT, SOFF)                                                                                             \
    {                                                                                                                  \
        _Pragma("unroll") for (int c = 0; c < 3; ++c)                                                                  \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, lds + (SOFF) + (wave * 3 + c) * 1024, 16, voa[c], (KT) * 96, 0, 0); \
        _Pragma("unroll") for (int c = 0; c < 3; ++c)                                                                  \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, lds + (SOFF) + X_OPER + (wave * 3 + c) * 1024, 16, vob[c],    \
                                                     (KT) * 96, 0, 0);                                                 \
    }
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int lr = lane & 31, lk = lane >> 5;
    // fragment (row, k-half lk) of term t: row * 96 + t * 32 + (lk ^ ((row >> 3) & 1)) * 16; the rows of a wave's fragments
    // differ by multiples of 32, so the swizzle bit is the lane's own (lr >> 3) & 1.  Banks (64 x 4 bytes; a ds_read_b128 is
    // served 16 lanes = 256 bytes at a time): 16 consecutive rows of one (term, half) sit at 96 r + const, i.e. on the
    // 16-byte granules 6 r mod 16 = {0, 6, 12, 2, 8, 14, 4, 10} twice over - the half swap of rows 8-15 moves the second
    // eight to the odd granules: conflict-free (PMC: SQ_LDS_BANK_CONFLICT was half of SQ_LDS_IDX_ACTIVE with bit 2)
    const int fh = (lk ^ ((lr >> 3) & 1)) * 16;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds;
    const unsigned aaddr = lds0 + (wm * 128 + lr) * 96 + fh, baddr = lds0 + X_OPER + (wn * 64 + lr) * 96 + fh;
    // The fragment reads are inline asm (x3_read) and so are their waits: with C++ reads the compiler cannot tell a read of
    // stage s from the LDS-DMA fill of another stage and drains ALL fills (vmcnt(0)) in front of the first read of every k
    // tile - with three stages the point is that the fill of tile t + 2 stays in flight across the barrier that ends tile t.
    // The fragments pass THROUGH each wait statement ("+v"), so no MFMA can be moved in front of its wait.
#define LC_XREAD_A(T, F) { x3_read<(T) * 32>(aaddr + cs, F[0]); x3_read<(T) * 32 + 3072>(aaddr + cs, F[1]);           \
                           x3_read<(T) * 32 + 6144>(aaddr + cs, F[2]); x3_read<(T) * 32 + 9216>(aaddr + cs, F[3]); }
#define LC_XREAD_B(T, F) { x3_read<(T) * 32>(baddr + cs, F[0]); x3_read<(T) * 32 + 3072>(baddr + cs, F[1]); }
#define LC_XWAIT(N, FA, FB)                                                                                            \
    asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(FA[0]), "+v"(FA[1]), "+v"(FA[2]), "+v"(FA[3]), "+v"(FB[0]), "+v"(FB[1]));
#define LC_XMMA(FA, FB)                                                                                                \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                      \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                  \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x3_frag(FA[i]), x3_frag(FB[j]), acc[i][j], 0, 0, 0);
    const int nk = p.nk;
    LC_XFILL(0, 0)
    if (nk > 1) {
        LC_XFILL(1, X_STAGE)
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");          // (6 fills per wave and stage) tile 0 has landed
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // (raw s_barrier, not __syncthreads: its workgroup fence would drain every outstanding fill - vmcnt(0) - again; the
    // waits that matter are written out: the wave's own fills by vmcnt, its fragment reads by lgkmcnt(0) above)
    __builtin_amdgcn_s_barrier();
    unsigned cs = 0, fs = 2 * X_STAGE;                            // stage being multiplied / being filled (byte offsets)
    for (int kt = 0; kt < nk; ++kt) {
#ifndef LC_X3_NOFILL
        if (kt + 2 < nk) LC_XFILL(kt + 2, fs)
#endif
#ifndef LC_X3_NOMMA
        {   // the small term pairs first: they meet the accumulator before the leading pair of the same k tile does
            i32x4g ah[4], am[4], al[4], bh[2], bm_[2], bl[2];
            LC_XREAD_A(0, ah) LC_XREAD_B(2, bl)
            LC_XREAD_A(2, al) LC_XREAD_B(0, bh)
            LC_XREAD_A(1, am) LC_XREAD_B(1, bm_)
            LC_XWAIT(12, ah, bl)
            LC_XMMA(ah, bl)
            __builtin_amdgcn_sched_barrier(0);                    // (or the scheduler hoists the next wait over these MFMAs)
            LC_XWAIT(6, al, bh)
            LC_XMMA(al, bh)
            __builtin_amdgcn_sched_barrier(0);
            LC_XWAIT(0, am, bm_)
            LC_XMMA(am, bm_)
            LC_XMMA(ah, bm_)
            LC_XMMA(am, bh)
            LC_XMMA(ah, bh)
        }
#endif
        if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // tile kt + 1 has landed, kt + 2 in flight
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        cs = cs == 2 * X_STAGE ? 0 : cs + X_STAGE;
        fs = fs == 2 * X_STAGE ? 0 : fs + X_STAGE;
    }
#undef LC_XFILL
#undef LC_XREAD_A
#undef LC_XREAD_B
#undef LC_XWAIT
#undef LC_XMMA
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + lr;
            if (col >= p.N) continue;
            const float bv = p.bias ? p.bias[col] : 0.f;
            unsigned est;
            int ecm;
            epi_column(p.epi, col, est, ecm);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (row < p.M) {
                    float *c = p.C + (size_t)row * p.ldc + col;
                    float v = p.alpha * acc[i][j][r] + bv;
                    if (p.beta != 0.f) v += p.beta * *c;
#ifdef LC_X3_NOSTORE                       // ablation build (tools/x3_dev_build.sh): what the C store costs
                    if (v == 12345.678f)
#endif
                    epi_store(p.epi, c, v, row, col, est, ecm);
                }
            }
        }
}

// TN form: C = alpha A^T B + ... with BOTH operands K-major - A the x3 shadow of X [K, M], B that of dZ [K, N]: the weight
// gradients X^T dZ (K = T * B rows) on the very shadows the forward / dX products read as row operands.  In the x3 layout a
// k-ROW's 256 columns of a tile are 16 column tiles x 96 bytes = 1536 contiguous bytes, and one term's 16 columns of a
// k-row are 32 contiguous bytes - exactly what ds_read_b64_tr_b16 wants: a 16-lane group hands in four k-rows x four
// 8-byte pieces and every lane receives the four k of ONE column (two reads per fragment: k 0..3 and 4..7 of the lane's
// k-octet).  LDS image per stage and operand: 32 slots of 1 KB = (k-row, half of the 256 columns); a slot is ONE DMA
// instruction whose lanes 4 r .. 4 r + 47 (r = k-row & 3) carry the half-row's 768 bytes and whose other lanes point past
// the buffer descriptor (zeros, no fetch): the data of k-row r start 64 r bytes into the slot, so the four k-rows x two
// column halves a 32-lane read touches fall on the eight different 32-byte bank chunks.  Two stages of 64 KB.  Columns
// past M / N are never zeroed - column m of A only ever meets row m of C, which the epilogue does not store; rows past K
// lie beyond the descriptor and arrive as zeros.
typedef int i32x2x __attribute__((ext_vector_type(2)));
constexpr int XT_OPER = 32 * 1024, XT_STAGE = 2 * XT_OPER;
template <int OFF>
__device__ __forceinline__ void x3_tr_pair(unsigned lds_addr, i32x2x &lo, i32x2x &hi)
{
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"
                 : "=&v"(lo), "=&v"(hi)
                 : "v"(lds_addr), "n"(OFF), "n"(OFF + 8192));       // k-rows + 4: four k-rows x two slots further
}
__device__ __forceinline__ bf16x8 x3_tr_join(i32x2x lo, i32x2x hi)
{
    union { i32x2x h[2]; bf16x8 v; } u;
    u.h[0] = lo; u.h[1] = hi;
    return u.v;
}
__global__ __launch_bounds__(XNT, 1) void gemm_x3_tn_kernel(X3Args p)
{
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * XT_STAGE];
    int bm, bn;
    x3_tile_order(p.M, p.N, bm, bn);
    const int m0 = bm * XBM, n0 = bn * XBN;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int kbeg = blockIdx.z * p.kchunk, kend = min(p.K, kbeg + p.kchunk);
    const int nk = (kend - kbeg + 15) / 16;
    // descriptors: from the tile's first column tile in k-row kbeg to the end of k-row kend - 1 of the whole matrix
    const unsigned short *ab = p.A + (size_t)kbeg * p.lda + (m0 / 16) * 48, *bb = p.B + (size_t)kbeg * p.ldb + (n0 / 16) * 48;
    const long long a_left = ((long long)(kend - kbeg) * p.lda - (m0 / 16) * 48) * 2;
    const long long b_left = ((long long)(kend - kbeg) * p.ldb - (n0 / 16) * 48) * 2;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void *)ab, 0, (int)a_left, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void *)bb, 0, (int)b_left, 0x00020000);
    // wave w fills slots 4 w .. 4 w + 3 of each operand: k-row 2 w + c / 2, column half c % 2
    int voa[4], vob[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int kr = wave * 2 + (c >> 1), half = c & 1, q = lane - 4 * (kr & 3);
        const bool live = q >= 0 && q < 48;
        voa[c] = live ? kr * p.lda * 2 + half * 768 + q * 16 : 0x7ffffff0;
        vob[c] = live ? kr * p.ldb * 2 + half * 768 + q * 16 : 0x7ffffff0;
    }
    const int kstep_a = 16 * p.lda * 2, kstep_b = 16 * p.ldb * 2;          // bytes per k tile
#define LC_TFILL(KT, SOFF)                                                                                             \
    {                                                                                                                  \
        _Pragma("unroll") for (int c = 0; c < 4; ++c)                                                                  \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, lds + (SOFF) + (wave * 4 + c) * 1024, 16, voa[c],             \
                                                     (KT) * kstep_a, 0, 0);                                            \
        _Pragma("unroll") for (int c = 0; c < 4; ++c)                                                                  \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, lds + (SOFF) + XT_OPER + (wave * 4 + c) * 1024, 16, vob[c],   \
                                                     (KT) * kstep_b, 0, 0);                                            \
    }
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int lr = lane & 31, lk = lane >> 5;
    // transposing reads: 16-lane group g = lane / 16 covers columns 16 (g & 1) .. + 15 of a 32-column MFMA tile and k-octet
    // g / 2; in it lane (r = (l % 16) / 4, c = l % 4) hands in k-row 8 (g / 2) + r (+ 4 for the second read), 8-byte piece c.
    // Column tile 8 wm + 2 I + (g & 1) of A is tile 2 I + (g & 1) of column half wm; of B: 4 wn + 2 J + (g & 1).
    const int tg = lane >> 4, tr = (lane & 15) >> 2, tc = lane & 3;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds;
    const unsigned tlane = lds0 + ((tg >> 1) * 8 + tr) * 2048 + 64 * tr + (tg & 1) * 96 + tc * 8;
    const unsigned aaddr = tlane + wm * 1024, baddr = tlane + XT_OPER + (wn >> 1) * 1024 + (wn & 1) * 384;
#define LC_TREAD_A(T, F) { x3_tr_pair<(T) * 32>(aaddr + cs, F[0][0], F[0][1]); x3_tr_pair<(T) * 32 + 192>(aaddr + cs, F[1][0], F[1][1]); \
                           x3_tr_pair<(T) * 32 + 384>(aaddr + cs, F[2][0], F[2][1]); x3_tr_pair<(T) * 32 + 576>(aaddr + cs, F[3][0], F[3][1]); }
#define LC_TREAD_B(T, F) { x3_tr_pair<(T) * 32>(baddr + cs, F[0][0], F[0][1]); x3_tr_pair<(T) * 32 + 192>(baddr + cs, F[1][0], F[1][1]); }
#define LC_TWAIT(N, FA, FB)                                                                                            \
    asm volatile("s_waitcnt lgkmcnt(" #N ")"                                                                          \
                 : "+v"(FA[0][0]), "+v"(FA[0][1]), "+v"(FA[1][0]), "+v"(FA[1][1]), "+v"(FA[2][0]), "+v"(FA[2][1]),      \
                   "+v"(FA[3][0]), "+v"(FA[3][1]), "+v"(FB[0][0]), "+v"(FB[0][1]), "+v"(FB[1][0]), "+v"(FB[1][1]));
#define LC_TMMA(FA, FB)                                                                                                \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                      \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                  \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x3_tr_join(FA[i][0], FA[i][1]),                        \
                                                               x3_tr_join(FB[j][0], FB[j][1]), acc[i][j], 0, 0, 0);
    LC_TFILL(0, 0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    unsigned cs = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) LC_TFILL(kt + 1, cs ^ XT_STAGE)
        {
            i32x2x ah[4][2], am[4][2], al[4][2], bh[2][2], bm_[2][2], bl[2][2];
            // (lgkmcnt counts to 15: twelve reads per group, at most two groups outstanding)
            LC_TREAD_A(0, ah) LC_TREAD_B(2, bl)
            LC_TREAD_A(2, al) LC_TREAD_B(0, bh)
            LC_TWAIT(12, ah, bl)
            LC_TMMA(ah, bl)
            __builtin_amdgcn_sched_barrier(0);
            LC_TREAD_A(1, am) LC_TREAD_B(1, bm_)
            LC_TWAIT(12, al, bh)
            LC_TMMA(al, bh)
            __builtin_amdgcn_sched_barrier(0);
            LC_TWAIT(0, am, bm_)
            LC_TMMA(am, bm_)
            LC_TMMA(ah, bm_)
            LC_TMMA(am, bh)
            LC_TMMA(ah, bh)
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // tile kt + 1 has landed (this wave's part of it)
        __builtin_amdgcn_s_barrier();
        cs ^= XT_STAGE;
    }
#undef LC_TFILL
#undef LC_TREAD_A
#undef LC_TREAD_B
#undef LC_TWAIT
#undef LC_TMMA
    if (p.slab) {
        float *S = p.slab + (size_t)blockIdx.z * p.M * p.N;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = n0 + wn * 64 + j * 32 + lr;
                if (col >= p.N) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                    if (row < p.M) S[(size_t)row * p.N + col] = acc[i][j][r];
                }
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + lr;
            if (col >= p.N) continue;
            const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (row < p.M) {
                    float *c = p.C + (size_t)row * p.ldc + col;
                    float v = p.alpha * acc[i][j][r] + bv;
                    if (p.beta != 0.f) v += p.beta * *c;
                    *c = v;
                }
            }
        }
}

// C = alpha * sum_s slab[s] + beta * C + bias over the [nslices][M][N] partial products of a K-split launch
__global__ __launch_bounds__(256) void x3_reduce_kernel(const float *__restrict__ slab, int nslices, int M, int N, float alpha,
                                                        float beta, float *__restrict__ C, int ldc,
                                                        const float *__restrict__ bias)
{
    const size_t total = (size_t)M * N;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(e / N), col = (int)(e % N);
        float s = slab[e];
        for (int k = 1; k < nslices; ++k) s += slab[(size_t)k * total + e];
        float *c = C + (size_t)row * ldc + col;
        float o = alpha * s;
        if (bias) o += bias[col];
        if (beta != 0.f) o += beta * *c;
        *c = o;
    }
}

// x [rows, cols] fp32 -> x3 shadow: a thread splits 8 consecutive k of a row (two float4 in, one 16-byte store per term);
// columns past `cols` up to the next multiple of 16 are written as zeros.
__device__ __forceinline__ void x3_split(float a, unsigned short &h, unsigned short &m, unsigned short &l)
{
    const __bf16 bh = (__bf16)a;
    const float r1 = a - (float)bh;             // exact
    const __bf16 bm = (__bf16)r1;
    const float r2 = r1 - (float)bm;            // exact
    const __bf16 bl = (__bf16)r2;
    h = __builtin_bit_cast(unsigned short, bh);
    m = __builtin_bit_cast(unsigned short, bm);
    l = __builtin_bit_cast(unsigned short, bl);
}
__global__ __launch_bounds__(256) void split_x3_kernel(const float *__restrict__ x, long long rows, int cols, int ldx,
                                                       unsigned short *__restrict__ out, int ldo, int vec)
{
    const int octs = (cols + 15) / 16 * 2;                         // 8-column groups per row (padded)
    const long long total = rows * octs;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const long long r = t / octs;
        const int o = (int)(t - r * octs), c0 = o * 8;
        float v[8];
        const float *src = x + r * ldx + c0;
        if (vec && c0 + 8 <= cols) {
            const float4 v0 = *reinterpret_cast<const float4 *>(src), v1 = *reinterpret_cast<const float4 *>(src + 4);
            v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w; v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = c0 + q < cols ? src[q] : 0.f;
        }
        unsigned short h[8], m[8], l[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) x3_split(v[q], h[q], m[q], l[q]);
        unsigned short *dst = out + r * ldo + (o >> 1) * 48 + (o & 1) * 8;
        auto pack = [](const unsigned short *s) {
            return make_uint4(s[0] | ((unsigned)s[1] << 16), s[2] | ((unsigned)s[3] << 16), s[4] | ((unsigned)s[5] << 16),
                              s[6] | ((unsigned)s[7] << 16));
        };
        *reinterpret_cast<uint4 *>(dst) = pack(h);
        *reinterpret_cast<uint4 *>(dst + 16) = pack(m);
        *reinterpret_cast<uint4 *>(dst + 32) = pack(l);
    }
}

}  // namespace

extern "C" int lc_split_bf16x3(const float *x, int rows, int cols, int ldx, uint16_t *out, int ldo, lc_stream_t stream)
{
    LC_CHECK_ARG(rows >= 0 && cols >= 0, "lc_split_bf16x3: negative dimension");
    if (rows == 0 || cols == 0) return LC_OK;
    LC_CHECK_ARG(x && out, "lc_split_bf16x3: null pointer");
    const int kp = (cols + 15) / 16 * 16;
    LC_CHECK_ARG(ldx >= cols && ldo >= 3 * kp && ldo % 8 == 0 && (((uintptr_t)out) & 15) == 0,
                 "lc_split_bf16x3: ldo must be >= 3 * roundup(cols, 16) and a multiple of 8, out 16-byte aligned");
    const int vec = (((uintptr_t)x) & 15) == 0 && ldx % 4 == 0;
    const long long total = (long long)rows * (kp / 8);
    long long g = (total + 255) / 256;
    if (g > 16384) g = 16384;
    hipLaunchKernelGGL(split_x3_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, (long long)rows, cols, ldx,
                       out, ldo, vec);
    LC_CHECK_LAUNCH("lc_split_bf16x3");
    return LC_OK;
}

extern "C" int lc_gemm_bf16x3_nt(int M, int N, int K, float alpha, const uint16_t *A, int lda, const uint16_t *B, int ldb,
                                 float beta, float *C, int ldc, const float *bias, lc_stream_t stream)
{
    const EpiArgs epi = lc_epi_take();
    LC_CHECK_ARG(M >= 0 && N >= 0 && K > 0, "lc_gemm_bf16x3_nt: bad dimension");
    if (M == 0 || N == 0) return LC_OK;
    LC_CHECK_ARG(A && B && C, "lc_gemm_bf16x3_nt: null pointer");
    const int nk = (K + 15) / 16;
    LC_CHECK_ARG(lda >= 48 * nk && ldb >= 48 * nk && ldc >= N, "lc_gemm_bf16x3_nt: leading dimension too small (x3 shadows: 3 * roundup(K, 16))");
    LC_CHECK_ARG(lda % 8 == 0 && ldb % 8 == 0 && (((uintptr_t)A) & 15) == 0 && (((uintptr_t)B) & 15) == 0,
                 "lc_gemm_bf16x3_nt: lda, ldb must be multiples of 8 and the operands 16-byte aligned");
    LC_CHECK_ARG(255ll * lda * 2 + 96ll * nk < 0x7fffffffll && 255ll * ldb * 2 + 96ll * nk < 0x7fffffffll,
                 "lc_gemm_bf16x3_nt: operand rows too long");
    const long long tiles = (long long)lc_cdiv(M, XBM) * lc_cdiv(N, XBN);
    LC_CHECK_ARG(tiles < (1ll << 31), "lc_gemm_bf16x3_nt: grid too large");
    X3Args p;
    p.epi = epi;
    p.M = M; p.N = N; p.nk = nk; p.alpha = alpha; p.beta = beta;
    p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.C = C; p.ldc = ldc; p.bias = bias;
    p.K = K; p.kchunk = K; p.slab = nullptr;
    hipLaunchKernelGGL(gemm_x3_kernel, dim3((unsigned)tiles), dim3(XNT), 0, (hipStream_t)stream, p);
    LC_CHECK_LAUNCH("lc_gemm_bf16x3_nt");
    return LC_OK;
}

// K slices of the TN form: fill whole rounds of 256 CUs, slices at least 1024 rows deep, at most 32 - and never fewer than
// the 2 GB reach of a slice's buffer descriptor asks for: a slice of an operand is kchunk rows of `ld` bf16 (ld = the larger
// leading dimension), addressed with 32-bit byte offsets from the slice's own base
static int x3_tn_slices(int M, int N, int K, long long ld)
{
    const long long tiles = (long long)lc_cdiv(M, XBM) * lc_cdiv(N, XBN);
    // rows one descriptor covers, whole k tiles; 1.75 GB: the kernel parks unused lanes at offset 0x7ffffff0, which must stay
    // beyond the slice
    const long long reach = (0x70000000ll / (ld * 2)) / 16 * 16;
    const int need = reach > 0 ? (int)((K + reach - 1) / reach) : 1 << 30;
    int nsl = 1;
    if (tiles < 256 && K >= 4096) {
        double best = 0.0;
        for (int c = 1; c <= 32 && K / c >= 1024; ++c) {
            const long long wg = tiles * c, rounds = (wg + 255) / 256;
            const double eff = (double)wg / (double)(rounds * 256);
            if (c >= need && eff > best + 1e-9) { best = eff; nsl = c; }
        }
    }
    return nsl > need ? nsl : need;
}
static long long x3_min_ld(int M, int N) { return 3ll * ((((M > N ? M : N) + 15) / 16) * 16); }
extern "C" size_t lc_gemm_bf16x3_tn_workspace_bytes_ld(int M, int N, int K, int lda, int ldb)
{
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    long long ld = lda > ldb ? lda : ldb;
    if (ld < x3_min_ld(M, N)) ld = x3_min_ld(M, N);
    const int nsl = x3_tn_slices(M, N, K, ld);
    return nsl > 1 ? (size_t)nsl * M * N * sizeof(float) : 0;
}
extern "C" size_t lc_gemm_bf16x3_tn_workspace_bytes(int M, int N, int K)
{
    return lc_gemm_bf16x3_tn_workspace_bytes_ld(M, N, K, 0, 0);          // operands with their minimal leading dimensions
}

extern "C" int lc_gemm_bf16x3_tn(int M, int N, int K, float alpha, const uint16_t *A, int lda, const uint16_t *B, int ldb,
                                 float beta, float *C, int ldc, const float *bias, void *workspace, size_t workspace_bytes,
                                 lc_stream_t stream)
{
    (void)lc_epi_take();                                     // weight gradients: no activation epilogue; a pending one is dropped
    LC_CHECK_ARG(M >= 0 && N >= 0 && K > 0, "lc_gemm_bf16x3_tn: bad dimension");
    if (M == 0 || N == 0) return LC_OK;
    LC_CHECK_ARG(A && B && C, "lc_gemm_bf16x3_tn: null pointer");
    const int mp = (M + 15) / 16 * 16, np = (N + 15) / 16 * 16;
    LC_CHECK_ARG(lda >= 3 * mp && ldb >= 3 * np && ldc >= N,
                 "lc_gemm_bf16x3_tn: leading dimension too small (x3 shadows: 3 * roundup(columns, 16))");
    LC_CHECK_ARG(lda % 8 == 0 && ldb % 8 == 0 && (((uintptr_t)A) & 15) == 0 && (((uintptr_t)B) & 15) == 0,
                 "lc_gemm_bf16x3_tn: lda, ldb must be multiples of 8 and the operands 16-byte aligned");
    // a tile's k-row is read as 1536 bytes from its first column tile: the last tile of a ragged M / N reads past the row
    // end into the next row (harmless, see the kernel) - but never past the allocation's last row, which the descriptor ends
    const long long tiles = (long long)lc_cdiv(M, XBM) * lc_cdiv(N, XBN);
    LC_CHECK_ARG(tiles < 65536, "lc_gemm_bf16x3_tn: too many tiles");
    const long long ldmax = lda > ldb ? lda : ldb;
    int nsl = x3_tn_slices(M, N, K, ldmax);
    if (nsl > 1 && (!workspace || workspace_bytes < (size_t)nsl * M * N * sizeof(float))) nsl = 1;
    X3Args p;
    p.epi = {1.f, 1.f, 0u, 0u, 1, nullptr, 0, 0, 0};
    p.M = M; p.N = N; p.nk = 0; p.alpha = alpha; p.beta = beta;
    p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.C = C; p.ldc = ldc; p.bias = bias;
    p.K = K;
    p.kchunk = nsl > 1 ? lc_cdiv(lc_cdiv(K, nsl), 16) * 16 : (K + 15) / 16 * 16;
    if (nsl > 1) nsl = lc_cdiv(K, p.kchunk);
    p.slab = nsl > 1 ? (float *)workspace : nullptr;
    LC_CHECK_ARG((long long)p.kchunk * (lda > ldb ? lda : ldb) * 2 < 0x7fffffffll,
                 "lc_gemm_bf16x3_tn: a K slice of an operand exceeds 2 GB: pass the workspace of "
                 "lc_gemm_bf16x3_tn_workspace_bytes_ld(M, N, K, lda, ldb) so that K is split far enough");
    hipLaunchKernelGGL(gemm_x3_tn_kernel, dim3((unsigned)tiles, 1, (unsigned)nsl), dim3(XNT), 0, (hipStream_t)stream, p);
    LC_CHECK_LAUNCH("lc_gemm_bf16x3_tn");
    if (nsl > 1) {
        long long g = ((long long)M * N + 255) / 256;
        if (g > 4096) g = 4096;
        hipLaunchKernelGGL(x3_reduce_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const float *)workspace, nsl,
                           M, N, alpha, beta, C, ldc, bias);
        LC_CHECK_LAUNCH("lc_gemm_bf16x3_tn (K-slice reduction)");
    }
    return LC_OK;
}
