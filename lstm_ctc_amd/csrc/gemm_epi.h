// gemm_epi.h - the one-shot fused epilogue shared by the product kernels (gemm.hip, gemm_x3.hip).
#pragma once
#include "common.h"

// Optional fused epilogue of ONE product (lc_gemm_next_epilogue): the DropoutWrapper mask of the layer output the GEMM
// produces - element (R, C) of the WHOLE output matrix is scaled by the factor of (seed, stream0 + C / P, R * P + C % P), the
// very factor lc_dropout_scale applies to the column windows of width P - and / or a bf16 (RNE) shadow of the result.
// row0 / col0: origin of this launch's block inside the whole matrix (the strips of a ragged product).
struct EpiArgs {
    float keep, inv_keep;       // keep >= 1: no mask
    unsigned seed, stream0;
    int P;
    unsigned short *c16;        // shadow of THIS launch's block (same origin as C), or nullptr
    int ldc16;
    int row0, col0;
    int skip_c;                 // non-zero (needs c16): the fp32 C is NOT stored - every consumer reads the shadow
};
__device__ __forceinline__ void epi_column(const EpiArgs &e, int col, unsigned &stream, int &cm)
{
    const int c = col + e.col0;
    const int q = e.keep < 1.f ? c / e.P : 0;
    stream = e.stream0 + (unsigned)q;
    cm = c - q * e.P;
}
__device__ __forceinline__ float epi_value(const EpiArgs &e, float v, int row, int col, unsigned stream, int cm)
{
    if (e.keep < 1.f) v *= lc_dropout_factor(e.seed, stream, (uint64_t)(row + e.row0) * e.P + cm, e.keep, e.inv_keep);
    if (e.c16) e.c16[(size_t)row * e.ldc16 + col] = __builtin_bit_cast(unsigned short, (__bf16)v);
    return v;
}
// the store of one finished element: C (unless the epilogue says nobody reads it) and, inside epi_value, the shadow
__device__ __forceinline__ void epi_store(const EpiArgs &e, float *c, float v, int row, int col, unsigned stream, int cm)
{
    const float r = epi_value(e, v, row, col, stream, cm);
    if (!e.skip_c) *c = r;
}
// host side: the pending epilogue of the calling thread, taken (and cleared) by every lc_gemm_* entry (gemm.hip)
EpiArgs lc_epi_take();
static inline bool epi_active(const EpiArgs &e) { return e.keep < 1.f || e.c16 != nullptr; }
static inline EpiArgs epi_block(EpiArgs e, int row0, int col0)      // the epilogue of a sub-block whose origin is (row0, col0)
{
    e.row0 += row0; e.col0 += col0;
    if (e.c16) e.c16 += (size_t)row0 * e.ldc16 + col0;
    return e;
}
