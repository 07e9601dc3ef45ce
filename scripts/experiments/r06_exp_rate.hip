// Round 6 probe: issue rate of v_exp_f32 / v_rcp_f32 against v_fma_f32 on gfx950 (cycles per wave64 instruction on one SIMD), with 1, 2 and 4 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/exp_rate scripts/experiments/r06_exp_rate.hip && /tmp/exp_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    float a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = 0.001f * (threadIdx.x + j);
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[j]));
            else if (OP == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(a[j]));
            else if (OP == 2) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[j]));
            else asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(*reinterpret_cast<double *>(&a[j & ~1])));
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += a[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)(t1 - t0) * 0.f;
    if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long *>(out)[1 << 20] = t1 - t0;
}
template <int OP> void run(const char *name, float *out, int waves_per_simd)
{
    const int iters = 4096, threads = 256, blocks = 256 * waves_per_simd;   // 4 waves a workgroup = 1 per SIMD per workgroup
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long cyc; hipMemcpy(&cyc, reinterpret_cast<long long *>(out) + (1 << 20), 8, hipMemcpyDeviceToHost);
    printf("%-12s %d wave(s)/SIMD: %.1f us, %.2f counter ticks per instruction per wave (the counter runs at 100 MHz: x clock / 1e8), %.2f ns per instruction per SIMD\n", name, waves_per_simd, ms * 1e3,
           (double)cyc / (iters * 8.0), ms * 1e6 / (iters * 8.0 * waves_per_simd));
}
int main()
{
    float *out; hipMalloc(&out, (1 << 23) + 64);
    for (int w : {1, 2, 4}) { run<0>("v_fma_f32", out, w); run<1>("v_exp_f32", out, w); run<2>("v_rcp_f32", out, w); run<3>("v_pk_fma_f32", out, w); }
    return 0;
}
