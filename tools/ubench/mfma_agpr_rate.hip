// Microbenchmark (development tool): rate of the pair kernels' MFMA stream - inline-asm v_mfma_f32_16x16x4_f32 with the B
// operand in an AGPR, 8 (forward) or 4 (BPTT) accumulator chains, one wave per SIMD on every CU - in s_memtime ticks per
// MFMA and in ns (HIP events), with 0 / 2 / 5 VALU instructions behind every MFMA.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_agpr_rate mfma_agpr_rate.hip && ./mfma_agpr_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC, int NVALU>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *t, int iters, const float *src)
{
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    float w[32], a[8], v = src[threadIdx.x];
    for (int i = 0; i < 32; ++i) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(w[i]) : "v"(src[threadIdx.x + 64 * i]));
    for (int i = 0; i < 8; ++i) a[i] = src[threadIdx.x + 8 * i];
    asm volatile("s_nop 7" ::: "memory");
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 256; ++m) {
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[m % NACC]) : "v"(a[m & 7]), "a"(w[m & 31]));
#pragma unroll
            for (int q = 0; q < NVALU; ++q) v = __builtin_fmaf(v, 1.0001f, 0.5f);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 7" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = v;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = t1 - t0;
}
// One wave per SIMD, NLDS ds_read_b32 (or, MEM = 1, buffer-less global_load_dword) instructions behind every MFMA: do LDS /
// vector-memory instructions ride under an f32 MFMA where VALU instructions do not?
template <int NLDS, int MEM>
__global__ __launch_bounds__(256) void k3(float *out, unsigned long long *t, int iters, const float *src)
{
    __shared__ float lds[4096];
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    float w[32], a[8];
    for (int i = 0; i < 32; ++i) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(w[i]) : "v"(src[threadIdx.x + 64 * i]));
    for (int i = 0; i < 8; ++i) a[i] = src[threadIdx.x + 8 * i];
    lds[threadIdx.x] = a[0];
    __syncthreads();
    const unsigned la = (unsigned)(threadIdx.x * 4);
    const float *gp = src + threadIdx.x;
    float tmp = 0.f;
    asm volatile("s_nop 7" ::: "memory");
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 256; ++m) {
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[m % 8]) : "v"(a[m & 7]), "a"(w[m & 31]));
#pragma unroll
            for (int q = 0; q < NLDS; ++q) {
                if (MEM) asm volatile("global_load_dword %0, %1, off" : "=v"(tmp) : "v"(gp) : "memory");
                else asm volatile("ds_read_b32 %0, %1" : "=v"(tmp) : "v"(la) : "memory");
            }
            if ((m & 15) == 15) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 7" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s2 = tmp;
    for (int i = 0; i < 8; ++i) s2 += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s2;
    if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = t1 - t0;
}
// Two waves per SIMD (512 threads): waves 0-3 issue 128 MFMAs with NVALU instructions behind each, waves 4-7 128 bare MFMAs;
// the SIMD's matrix pipe sees 256 MFMAs per iteration.  Ticks per iteration of wave 0 (8192 = the pipe alone).
template <int NVALU>
__global__ __launch_bounds__(512) void k2(float *out, unsigned long long *t, int iters, const float *src)
{
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    float w[32], a[8], v = src[threadIdx.x];
    for (int i = 0; i < 32; ++i) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(w[i]) : "v"(src[threadIdx.x + 64 * i]));
    for (int i = 0; i < 8; ++i) a[i] = src[threadIdx.x + 8 * i];
    asm volatile("s_nop 7" ::: "memory");
    const bool worker = threadIdx.x < 256;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (worker) {
#pragma unroll
            for (int m = 0; m < 128; ++m) {
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[m % 8]) : "v"(a[m & 7]), "a"(w[m & 31]));
#pragma unroll
                for (int q = 0; q < NVALU; ++q) v = __builtin_fmaf(v, 1.0001f, 0.5f);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int m = 0; m < 128; ++m) {
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[m % 8]) : "v"(a[m & 7]), "a"(w[m & 31]));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 7" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s2 = v;
    for (int i = 0; i < 8; ++i) s2 += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 512 + threadIdx.x] = s2;
    if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = t1 - t0;
}
// The same question for the bf16 MFMA (v_mfma_f32_32x32x16_bf16, 8 passes = 32 cycles... measured below) and the f32 32x32x2
// (16 passes): NVALU dependent-free VALU instructions behind every MFMA, one wave per SIMD.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NVALU, int F32>
__global__ __launch_bounds__(256) void k4(float *out, unsigned long long *t, int iters, const float *src)
{
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)src[threadIdx.x + i]; b[i] = (__bf16)src[threadIdx.x + 8 + i]; }
    float fa = src[threadIdx.x], fb = src[threadIdx.x + 1];
    float v[4] = {src[threadIdx.x], src[threadIdx.x + 1], src[threadIdx.x + 2], src[threadIdx.x + 3]};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 64; ++m) {
            if (F32) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[m & 3], 0, 0, 0);
            else acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < NVALU; ++q) v[q & 3] = __builtin_fmaf(v[q & 3], 1.0001f, 0.5f);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s2 = v[0] + v[1] + v[2] + v[3];
    for (int i = 0; i < 4; ++i) s2 += acc[i][0] + acc[i][5];
    out[blockIdx.x * 256 + threadIdx.x] = s2;
    if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = t1 - t0;
}
int main()
{
    float *out, *src; unsigned long long *t, h;
    hipMalloc(&out, 4 * 256 * 1024); hipMalloc(&src, 4 * 65536); hipMemset(src, 0, 4 * 65536); hipMalloc(&t, 8);
    const int iters = 200;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
#define RUN(NACC, NV)                                                                                          \
    for (int grid : {1, 256}) {                                                                                \
        hipLaunchKernelGGL((k<NACC, NV>), dim3(grid), dim3(256), 0, 0, out, t, iters, src);                    \
        hipEventRecord(e0); hipLaunchKernelGGL((k<NACC, NV>), dim3(grid), dim3(256), 0, 0, out, t, iters, src); \
        hipEventRecord(e1); hipDeviceSynchronize();                                                            \
        float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);                \
        printf("chains %d, %d VALU per MFMA, grid %3d: %.2f ticks per MFMA, %.2f ns per MFMA (tick = %.3f ns)\n", NACC, NV, \
               grid, (double)h / (256.0 * iters), ms * 1e6 / (256.0 * iters), ms * 1e6 / (double)h);            \
    }
    RUN(8, 0) RUN(8, 2) RUN(8, 5) RUN(8, 8) RUN(4, 0) RUN(4, 5)
#define RUN2(NV)                                                                                               \
    {                                                                                                          \
        hipLaunchKernelGGL((k2<NV>), dim3(256), dim3(512), 0, 0, out, t, iters, src);                          \
        hipDeviceSynchronize(); hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);                                    \
        printf("two waves per SIMD, %2d VALU behind each of wave 0-3's MFMAs: %.0f ticks per iteration of 256 MFMAs per SIMD\n", \
               NV, (double)h / iters);                                                                         \
    }
    RUN2(0) RUN2(5) RUN2(10) RUN2(20)
#define RUN3(NL, MEM)                                                                                          \
    {                                                                                                          \
        hipLaunchKernelGGL((k3<NL, MEM>), dim3(256), dim3(256), 0, 0, out, t, iters, src);                     \
        hipDeviceSynchronize(); hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);                                    \
        printf("%d %s behind every MFMA: %.2f ticks per MFMA\n", NL, MEM ? "global_load_dword" : "ds_read_b32",  \
               (double)h / (256.0 * iters));                                                                   \
    }
    RUN3(1, 0) RUN3(2, 0) RUN3(4, 0) RUN3(1, 1) RUN3(2, 1)
#define RUN4(NV, F32)                                                                                          \
    {                                                                                                          \
        hipLaunchKernelGGL((k4<NV, F32>), dim3(256), dim3(256), 0, 0, out, t, iters, src);                     \
        hipDeviceSynchronize(); hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);                                    \
        printf("%s, %d independent VALU behind every MFMA: %.2f ticks per MFMA\n",                              \
               F32 ? "v_mfma_f32_32x32x2_f32" : "v_mfma_f32_32x32x16_bf16", NV, (double)h / (64.0 * iters));      \
    }
    RUN4(0, 0) RUN4(4, 0) RUN4(8, 0) RUN4(0, 1) RUN4(4, 1) RUN4(8, 1)
    return 0;
}
