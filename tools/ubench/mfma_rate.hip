// Microbenchmark: issue rate of v_mfma_f32_16x16x4_f32 / 32x32x2 from one wave per SIMD (development tool).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k16(float *out, unsigned long long *t, int iters, float a, float b)
{
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 32 / NACC; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = t1 - t0;
}
__global__ __launch_bounds__(256) void k32(float *out, unsigned long long *t, int iters, float a, float b)
{
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][5];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = t1 - t0;
}
// LSTM-like: 4 A float4 x 2 W float4 from memory, quad-major, optional refill loads in the loop
template <int LOADS>
__global__ __launch_bounds__(256) void klstm(float *out, unsigned long long *t, int iters, const float4 *src)
{
    f32x4 acc[4][2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
    float4 a[4], w[2];
    const float4 *p = src + threadIdx.x;
    for (int i = 0; i < 4; ++i) a[i] = p[i * 256];
    for (int i = 0; i < 2; ++i) w[i] = p[(4 + i) * 256];
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        float4 na[4], nw[2];
        if (LOADS) {
            const float4 *q = p + (size_t)((it & 7) + 1) * 6 * 256;
            for (int i = 0; i < 4; ++i) na[i] = q[i * 256];
            for (int i = 0; i < 2; ++i) nw[i] = q[(4 + i) * 256];
        }
#define Q(c) for (int m = 0; m < 4; ++m) for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].c, w[n].c, acc[m][n], 0, 0, 0);
        Q(x) Q(y) Q(z) Q(w)
        if (LOADS) { for (int i = 0; i < 4; ++i) a[i] = na[i]; for (int i = 0; i < 2; ++i) w[i] = nw[i]; }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = t1 - t0;
}
int main()
{
    float *out; unsigned long long *t, h;
    hipMalloc(&out, 4 * 256 * 1024); hipMalloc(&t, 8);
    const int iters = 1000;
    for (int grid : {1, 256, 512}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
#define RUN(name, kern, nm, cyc)                                                                          \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, t, iters, 1.0f, 0.5f);                  \
        hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, t, iters, 1.0f, 0.5f); \
        hipEventRecord(e1); hipDeviceSynchronize();                                                        \
        { float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);          \
          printf("grid %3d %-18s ticks/mfma %.1f  wall ns/mfma %.2f (ideal %d cyc)\n", grid, name,         \
                 (double)h / (iters * nm), ms * 1e6 / (iters * nm), cyc); }
        RUN("16x16x4 acc=8", k16<8>, 32, 32)
        RUN("16x16x4 acc=4", k16<4>, 32, 32)
        RUN("16x16x4 acc=2", k16<2>, 32, 32)
        RUN("32x32x2 acc=4", k32, 16, 64)
        { float4 *src; hipMalloc(&src, 16 * 256 * 64); hipMemset(src, 0, 16 * 256 * 64);
#define RUN2(name, kern)                                                                                  \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, t, iters, src);                         \
        hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, t, iters, src);     \
        hipEventRecord(e1); hipDeviceSynchronize();                                                        \
        { float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);          \
          printf("grid %3d %-18s ticks/mfma %.1f  wall ns/mfma %.2f\n", grid, name, (double)h / (iters * 32), ms * 1e6 / (iters * 32)); }
          RUN2("lstm-like noload", klstm<0>)
          RUN2("lstm-like loads", klstm<1>)
          hipFree(src); }
    }
    return 0;
}
