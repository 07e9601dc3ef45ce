// Dev-only: issue rate of vector f64 add / mul / fma and dependent-chain latency on gfx950, one or two waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o f64_rate f64_rate.hip && ./f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE, int CHAINS>
__global__ __launch_bounds__(64) void k(double *out, double a, double b, int iters)
{
    double v[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) v[c] = a + threadIdx.x + c;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            if (MODE == 0) v[c] = v[c] + b;
            else if (MODE == 1) v[c] = v[c] * b;
            else v[c] = __builtin_fma(v[c], b, a);
        }
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += v[c];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int MODE, int CHAINS>
void run(const char *name, int blocks)
{
    double *out; hipMalloc(&out, 1 << 24);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, CHAINS><<<blocks, 64>>>(out, 1.0, 1.0000001, 100);
    hipEventRecord(e0);
    k<MODE, CHAINS><<<blocks, 64>>>(out, 1.0, 1.0000001, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ops_per_wave = (double)iters * CHAINS;
    printf("%-4s chains %2d  blocks %5d: %.3f ms  -> %.1f ns per wave-instruction (%.1f cycles at 2.4 GHz)\n", name, CHAINS, blocks, ms, ms * 1e6 / ops_per_wave, ms * 1e6 / ops_per_wave * 2.4);
    hipFree(out);
}
int main()
{
    for (int blocks : {1024, 2048}) {   // one / two waves per SIMD
        run<0, 1>("add", blocks); run<0, 8>("add", blocks);
        run<1, 1>("mul", blocks); run<1, 8>("mul", blocks);
        run<2, 1>("fma", blocks); run<2, 8>("fma", blocks);
    }
    return 0;
}
