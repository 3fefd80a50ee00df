// common.h — shared helpers for the gfx950 kernels behind include/lstm_ctc_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/lstm_ctc_hip.h"

#define LC_WAVE 64

void lc_set_error(const char *fmt, ...);
// development switches (error.cpp): per-thread lc_set_option() override > LC_* environment variable > dflt
enum { LC_OPT_LSTM_PERSISTENT = 0, LC_OPT_LSTM_SPIN_LIMIT, LC_OPT_GEMM_F32_BIG, LC_OPT_GEMM_BF16_BIG, LC_OPT_CTC_LSE2,
       LC_OPT_GEMM_BF16_PERSIST, LC_OPT_GEMM_TAIL };
long lc_option(int opt, long dflt);

#define LC_CHECK_ARG(cond, ...)                                   \
    do {                                                          \
        if (!(cond)) { lc_set_error(__VA_ARGS__); return LC_EINVAL; } \
    } while (0)

#define LC_CHECK_LAUNCH(what)                                                        \
    do {                                                                             \
        hipError_t e_ = hipGetLastError();                                           \
        if (e_ != hipSuccess) {                                                      \
            lc_set_error("%s: %s", what, hipGetErrorString(e_));                     \
            return LC_ELAUNCH;                                                       \
        }                                                                            \
    } while (0)

static inline int lc_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// Counter-based dropout mask shared bit-for-bit with oracle/oracle.py:dropout_mask.
// Returns 1/keep with probability keep, else 0.
__host__ __device__ inline uint32_t lc_fmix32(uint32_t h)
{
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
__host__ __device__ inline float lc_dropout_factor(uint32_t seed, uint32_t stream, uint64_t idx,
                                                   float keep, float inv_keep)
{
    uint32_t h = (seed * 0x9E3779B1u) ^ (stream * 0x85EBCA77u + 0x165667B1u);
    uint32_t v = lc_fmix32((uint32_t)idx ^ h);
    v = lc_fmix32(v ^ ((uint32_t)(idx >> 32) + 0x27D4EB2Fu));
    float u = (float)(v >> 8) * (1.0f / 16777216.0f);
    return u < keep ? inv_keep : 0.0f;
}

// ---- wave-level helpers (wave = 64 lanes on gfx950) ----
// lane i receives x from lane i-1; lane 0 receives fill.   (DPP wave_shr:1)
__device__ __forceinline__ float lc_wave_shr1(float x, float fill)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(x),
                                                      0x138, 0xf, 0xf, false));
}
// lane i receives x from lane i+1; lane 63 receives fill.  (DPP wave_shl:1)
__device__ __forceinline__ float lc_wave_shl1(float x, float fill)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(x),
                                                      0x130, 0xf, 0xf, false));
}
// Wave-wide reductions on the VALU (DPP row shifts + row broadcasts, result read from lane 63) -
// no LDS crossbar round trips (ds_bpermute costs ~60 cycles per hop and stalls the wave on lgkmcnt).
#define LC_DPP_STEP(OP, v, ctrl, rowmask)                                                                   \
    v = OP(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), ctrl, rowmask, \
                                                         0xf, false)))
__device__ __forceinline__ float lc_addf(float a, float b) { return a + b; }
__device__ __forceinline__ float lc_wave_max(float v)
{
    LC_DPP_STEP(fmaxf, v, 0x111, 0xf);   // row_shr:1
    LC_DPP_STEP(fmaxf, v, 0x112, 0xf);   // row_shr:2
    LC_DPP_STEP(fmaxf, v, 0x114, 0xf);   // row_shr:4
    LC_DPP_STEP(fmaxf, v, 0x118, 0xf);   // row_shr:8   -> lane 15 of every row holds the row max
    LC_DPP_STEP(fmaxf, v, 0x142, 0xa);   // row_bcast:15 into rows 1,3
    LC_DPP_STEP(fmaxf, v, 0x143, 0xc);   // row_bcast:31 into rows 2,3 -> lane 63 holds the wave max
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float lc_wave_sum(float v)
{
    // shifted-in lanes must contribute 0 for a sum: bound_ctrl=true feeds 0 from invalid source lanes
#define LC_DPP_ADD(ctrl, rowmask)                                                                            \
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rowmask, 0xf, true))
    LC_DPP_ADD(0x111, 0xf);
    LC_DPP_ADD(0x112, 0xf);
    LC_DPP_ADD(0x114, 0xf);
    LC_DPP_ADD(0x118, 0xf);
    LC_DPP_ADD(0x142, 0xa);
    LC_DPP_ADD(0x143, 0xc);
#undef LC_DPP_ADD
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// Gate non-linearities on the hardware transcendental pipe (v_exp_f32 / v_rcp_f32, 1 ulp each): absolute error
// ~1.5e-7, i.e. the rounding level of the fp32 gate values themselves (Eigen's vectorised exp / tanh that TF-1.8
// evaluates are rational approximations of the same quality).  The ocml expf / tanhf / IEEE-division forms used at
// first were ~250 VALU instructions per (row, unit) with a divergent branch inside tanhf: 3500 of the forward step
// kernel's ~11000 cycles (s_memtime stamps); these are ~25.
__device__ __forceinline__ float lc_sigmoid(float x)
{
    // 1 / (1 + e^-x); saturates cleanly: e^(+big) = inf -> rcp = 0, e^(-big) = 0 -> 1
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}
__device__ __forceinline__ float lc_tanh(float x)
{
    const float t = __builtin_amdgcn_exp2f(-2.88539008177792681f * fabsf(x));     // e^(-2|x|) in [0, 1]
    return copysignf((1.0f - t) * __builtin_amdgcn_rcpf(1.0f + t), x);
}
