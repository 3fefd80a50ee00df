// Microbenchmark (dev tool): how fast one XCD's L2 feeds k of its CUs when every CU streams the SAME 128 KB (what the
// single-XCD recurrences' exchange reads do every step) or its OWN 128 KB, with the loads of the product kernels
// (global_load_dwordx4 nt, 4 waves x 32 instructions of 1 KB).  Answers whether the ~31 bytes / clock / CU seen inside
// lstm_bwd_persist_bf16_kernel is the CU's own vector-memory path or the L2 serving 32 readers of the same lines.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/l2_read_rate.hip -o tools/ubench/l2_read_rate && tools/ubench/l2_read_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int KB128 = 128 * 1024;
// grid = 8 * 32 workgroups of 256 threads, 96 KB of dynamic LDS each (one workgroup per CU); workgroup b lands on XCD b % 8.
// Only XCD 0's first `active` workgroups read; MODE 0: the same buffer, 1: buffer (b / 8), 2: same buffer, plain (cacheable) loads.
template <int MODE, int NBLK>
__global__ __launch_bounds__(256) void reader(const char *buf, unsigned long long *ticks, unsigned *sink, int active, int iters)
{
    extern __shared__ char lds[];
    const int b = blockIdx.x, slot = b >> 3;
    if ((b & 7) != 0 || slot >= active) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const char *base = buf + (MODE == 1 ? (size_t)slot * KB128 : 0) + (size_t)wave * NBLK * 1024 + lane * 16;
    const int rot = (slot * 5) % NBLK;                                   // the kernels' staggered walk
    unsigned off[NBLK];
#pragma unroll
    for (int j = 0; j < NBLK; ++j) off[j] = (unsigned)((j + rot) % NBLK) * 1024u;
    unsigned acc = 0;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        u32x4 r[NBLK];
#pragma unroll
        for (int j = 0; j < NBLK; ++j) {
            const u32x4 *p = reinterpret_cast<const u32x4 *>(base + off[j]);
            r[j] = MODE == 2 ? *p : __builtin_nontemporal_load(p);
        }
#pragma unroll
        for (int j = 0; j < NBLK; ++j) acc ^= r[j].x ^ r[j].w;           // (both ends of the 16 bytes are consumed)
        asm volatile("" ::: "memory");
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) ticks[slot * 4 + wave] = t1 - t0;
    sink[b * 256 + threadIdx.x] = acc;
    if (lds[threadIdx.x] == 77) sink[0] = 1;
}
template <int MODE, int NBLK>
static void run(const char *name, const char *buf, unsigned long long *ticks, unsigned *sink, int active)
{
    const int iters = 400, nblk = NBLK;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(reader<MODE, NBLK>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipMemset(ticks, 0, 32 * 4 * 8);
    hipLaunchKernelGGL((reader<MODE, NBLK>), dim3(256), dim3(256), 96 * 1024, 0, buf, ticks, sink, active, 20);   // warm the L2
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((reader<MODE, NBLK>), dim3(256), dim3(256), 96 * 1024, 0, buf, ticks, sink, active, iters);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[128]; (void)hipMemcpy(h, ticks, sizeof h, hipMemcpyDeviceToHost);
    double worst = 0, sum = 0; int n = 0;
    for (int i = 0; i < active * 4; ++i) { if ((double)h[i] > worst) worst = (double)h[i]; sum += (double)h[i]; ++n; }
    const double bytes = (double)nblk * 4 * 1024 * iters;                 // per workgroup
    // s_memtime ticks are the stamps of tools/persist_probe.py ("cycles"); ticks per us of the slowest wave against the events' time
    printf("%-34s CUs %2d  KB/pass %3d  %.3f ms  %.0f ticks/us  %5.0f ticks/pass  per CU %5.1f B/tick (mean) %5.1f (slowest)  XCD total %6.0f B/tick\n", name,
           active, nblk * 4, ms, worst / (ms * 1e3), sum / n / iters, bytes / (sum / n), bytes / worst, active * bytes / worst);
}
template <int MODE, int NBLK>
static void sweep(const char *name, const char *buf, unsigned long long *ticks, unsigned *sink)
{
    for (int k : {1, 32}) run<MODE, NBLK>(name, buf, ticks, sink, k);
}
// The BPTT kernels' pattern: NCH dependent chunks of CL loads per wave per pass over a footprint of NCH * CL * 4 KB (the
// kernels: 2 x 16 over 128 KB).  LAYOUT 0: wave w owns a contiguous run of NCH * CL blocks; LAYOUT 1: blocks dealt round-robin
// to the waves (a chunk = contiguous bytes).
template <int LAYOUT, int NCH, int CL>
__global__ __launch_bounds__(256) void reader2(const char *buf, unsigned long long *ticks, unsigned *sink, int active, int iters)
{
    extern __shared__ char lds[];
    const int b = blockIdx.x, slot = b >> 3;
    if ((b & 7) != 0 || slot >= active) return;
    constexpr int PER = NCH * CL;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const char *base = buf + lane * 16;
    const int rot = (slot * 5) % PER;
    unsigned acc = 0;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll 1
        for (int c = 0; c < NCH; ++c) {
            u32x4 r[CL];
#pragma unroll
            for (int j = 0; j < CL; ++j) {
                int q = c * CL + j + rot; q = q >= PER ? q - PER : q;
                const int blk = LAYOUT == 0 ? wave * PER + q : q * 4 + wave;
                r[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(base + (size_t)blk * 1024));
            }
#pragma unroll
            for (int j = 0; j < CL; ++j) acc ^= r[j].x ^ r[j].w;
            asm volatile("" ::: "memory");
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) ticks[slot * 4 + wave] = t1 - t0;
    sink[b * 256 + threadIdx.x] = acc;
    if (lds[threadIdx.x] == 77) sink[0] = 1;
}
template <int LAYOUT, int NCH, int CL>
static void run2(const char *name, const char *buf, unsigned long long *ticks, unsigned *sink, int active)
{
    const int iters = 400;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(reader2<LAYOUT, NCH, CL>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipMemset(ticks, 0, 32 * 4 * 8);
    hipLaunchKernelGGL((reader2<LAYOUT, NCH, CL>), dim3(256), dim3(256), 96 * 1024, 0, buf, ticks, sink, active, 20);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((reader2<LAYOUT, NCH, CL>), dim3(256), dim3(256), 96 * 1024, 0, buf, ticks, sink, active, iters);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[128]; (void)hipMemcpy(h, ticks, sizeof h, hipMemcpyDeviceToHost);
    double sum = 0; int n = 0;
    for (int i = 0; i < active * 4; ++i) { sum += (double)h[i]; ++n; }
    printf("%-40s footprint %3d KB = %d chunks of %2d loads / wave  CUs %2d  %.3f ms  %5.0f ticks per chunk  per CU %5.1f B/tick\n", name,
           NCH * CL * 4, NCH, CL, active, ms, sum / n / iters / NCH, (double)NCH * CL * 4096.0 * iters / (sum / n));
}
// The exchange itself, untimed protocol but real traffic: per pass every workgroup STORES its 4 KB piece of the 128 KB buffer
// (nt, 16 bytes per thread - what the producers publish), then reads all 128 KB as 2 chunks of 16 loads per wave; four
// buffers in turn as in the kernels.  WRITE = false: the same walk over the four buffers without the stores.
template <bool WRITE>
__global__ __launch_bounds__(256) void exchanger(char *buf, unsigned long long *ticks, unsigned *sink, int active, int iters)
{
    extern __shared__ char lds[];
    const int b = blockIdx.x, slot = b >> 3;
    if ((b & 7) != 0 || slot >= active) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int rot = (slot * 5) % 32;
    unsigned acc = 0;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        char *cur = buf + (size_t)(it & 3) * KB128;
        if (WRITE) {
            u32x4 v = {(unsigned)it, acc, (unsigned)slot, 7u};
            __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(cur + (size_t)slot * 4096 + threadIdx.x * 16));
        }
        const char *base = cur + lane * 16;
#pragma unroll 1
        for (int c = 0; c < 2; ++c) {
            u32x4 r[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                int q = c * 16 + j + rot; q = q >= 32 ? q - 32 : q;
                r[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(base + (size_t)(wave * 32 + q) * 1024));
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) acc ^= r[j].x ^ r[j].w;
            asm volatile("" ::: "memory");
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) ticks[slot * 4 + wave] = t1 - t0;
    sink[b * 256 + threadIdx.x] = acc;
    if (lds[threadIdx.x] == 77) sink[0] = 1;
}
template <bool WRITE>
static void run3(const char *name, char *buf, unsigned long long *ticks, unsigned *sink, int active)
{
    const int iters = 400;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(exchanger<WRITE>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipMemset(ticks, 0, 32 * 4 * 8);
    hipLaunchKernelGGL((exchanger<WRITE>), dim3(256), dim3(256), 96 * 1024, 0, buf, ticks, sink, active, 20);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((exchanger<WRITE>), dim3(256), dim3(256), 96 * 1024, 0, buf, ticks, sink, active, iters);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[128]; (void)hipMemcpy(h, ticks, sizeof h, hipMemcpyDeviceToHost);
    double sum = 0; int n = 0;
    for (int i = 0; i < active * 4; ++i) { sum += (double)h[i]; ++n; }
    printf("%-64s CUs %2d  %.3f ms  %5.0f ticks per pass (128 KB)  per CU %5.1f B/tick\n", name, active, ms, sum / n / iters, 131072.0 * iters / (sum / n));
}
template <int NCH, int CL>
static void foot(const char *buf, unsigned long long *ticks, unsigned *sink)
{
    for (int k : {1, 32}) run2<0, NCH, CL>("wave owns a contiguous run (the kernels)", buf, ticks, sink, k);
    run2<1, NCH, CL>("blocks dealt round-robin to the waves", buf, ticks, sink, 32);
}
int main()
{
    char *buf; unsigned long long *ticks; unsigned *sink;
    (void)hipMalloc(&buf, 32 * KB128); (void)hipMemset(buf, 1, 32 * KB128);
    (void)hipMalloc(&ticks, 32 * 4 * 8); (void)hipMalloc(&sink, 256 * 256 * 4);
    for (int k : {1, 2, 4, 8, 16, 32}) run<0, 32>("same 128 KB, nt loads", buf, ticks, sink, k);
    for (int k : {1, 2, 4, 8, 16, 32}) run<1, 32>("own 128 KB each, nt loads", buf, ticks, sink, k);
    for (int k : {1, 2, 4, 8, 16, 32}) run<2, 32>("same 128 KB, plain loads (L1)", buf, ticks, sink, k);
    // one dependent batch of n loads per wave (4 n KB per workgroup and pass): latency against throughput
    sweep<0, 1>("same buffer, nt, n loads / wave", buf, ticks, sink); sweep<0, 2>("same buffer, nt, n loads / wave", buf, ticks, sink);
    sweep<0, 4>("same buffer, nt, n loads / wave", buf, ticks, sink); sweep<0, 8>("same buffer, nt, n loads / wave", buf, ticks, sink);
    sweep<0, 16>("same buffer, nt, n loads / wave", buf, ticks, sink); sweep<0, 32>("same buffer, nt, n loads / wave", buf, ticks, sink);
    sweep<2, 1>("same buffer, plain, n loads / wave", buf, ticks, sink); sweep<2, 2>("same buffer, plain, n loads / wave", buf, ticks, sink);
    sweep<2, 4>("same buffer, plain, n loads / wave", buf, ticks, sink); sweep<2, 8>("same buffer, plain, n loads / wave", buf, ticks, sink);
    sweep<2, 16>("same buffer, plain, n loads / wave", buf, ticks, sink);
    // where the rate falls: cyclic footprints of 32 .. 512 KB per workgroup, in dependent chunks of 8 / 16 loads per wave
    foot<1, 8>(buf, ticks, sink); foot<2, 8>(buf, ticks, sink); foot<3, 8>(buf, ticks, sink); foot<4, 8>(buf, ticks, sink);
    foot<1, 16>(buf, ticks, sink); foot<2, 16>(buf, ticks, sink); foot<3, 16>(buf, ticks, sink); foot<4, 16>(buf, ticks, sink); foot<8, 16>(buf, ticks, sink);
    for (int k : {1, 32}) run3<false>("four 128 KB buffers in turn, 2 chunks of 16 loads, no stores", buf, ticks, sink, k);
    for (int k : {1, 32}) run3<true>("the same, every workgroup stores its 4 KB piece first (nt)", buf, ticks, sink, k);
    return 0;
}
