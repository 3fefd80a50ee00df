// Microbenchmark: issue cost / dependent latency of plain and transcendental VALU ops, one wave (dev tool).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int MODE>
__global__ void k(float *out, unsigned long long *t, int iters, float seed)
{
    float a = seed + threadIdx.x, b = seed * 2 + threadIdx.x, c = seed * 3, d = seed * 4;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) { REP16(a = __builtin_amdgcn_exp2f(a);) }                       // dependent exp chain
        if (MODE == 1) { REP16(a = __builtin_amdgcn_exp2f(a); b = __builtin_amdgcn_exp2f(b); c = __builtin_amdgcn_exp2f(c); d = __builtin_amdgcn_exp2f(d);) }  // 4 independent
        if (MODE == 2) { REP16(a = a * 1.0001f + 0.5f;) }                              // dependent fma chain
        if (MODE == 3) { REP16(a = a * 1.0001f + 0.5f; b = b * 1.0001f + 0.5f; c = c * 1.0001f + 0.5f; d = d * 1.0001f + 0.5f;) }
        if (MODE == 4) { REP16(a = __builtin_amdgcn_logf(a);) }
        if (MODE == 5) { REP16(a = fmaxf(a, b) + __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(-fabsf(a - b)));) }   // lse2 chain
        if (MODE == 6) { REP16(a = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x138, 0xf, 0xf, false)) + 1.0f;) }   // dpp shift + add
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = a + b + c + d;
    if (threadIdx.x == 0) t[0] = t1 - t0;
}
int main()
{
    float *out; unsigned long long *t, h; hipMalloc(&out, 1024); hipMalloc(&t, 8);
    const int iters = 2000;
    const char *names[] = {"exp dep chain", "exp 4 indep", "fma dep chain", "fma 4 indep", "log dep chain", "lse2 dep chain (6 ops)", "dpp wave_shr + add chain (2 ops)"};
    const int nops[] = {16, 64, 16, 64, 16, 16, 16};
#define RUN(M) hipLaunchKernelGGL(k<M>, dim3(1), dim3(64), 0, 0, out, t, iters, 0.3f); hipDeviceSynchronize(); \
    hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost); printf("%-34s %.1f cycles per op-group\n", names[M], (double)h / (iters * nops[M]));
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6)
    return 0;
}
