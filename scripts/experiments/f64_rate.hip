// Dev-only: issue rate of vector f64 add / mul / fma and dependent-chain latency on gfx950, one / two / four waves per SIMD.
// Measured (MI355X): a dependent operation every ~9 cycles, independent ones every ~5-6 from a lone wave, ~4.8 per SIMD with two or four waves.
// (The first version ran one round per loop trip and measured the loop: 38 / 10 cycles.  Sixteen rounds per trip.)
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o f64_rate f64_rate.hip && ./f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE, int CHAINS>
__global__ __launch_bounds__(64) void k(double *out, double a, double b, int iters)
{
    double v[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) v[c] = a + threadIdx.x + c;
    for (int i = 0; i < iters; i += 16) {                    // 16 rounds per trip: the loop's own scalar instructions and branch are 1/16 of a round
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                if (MODE == 0) v[c] = v[c] + b;
                else if (MODE == 1) v[c] = v[c] * b;
                else v[c] = __builtin_fma(v[c], b, a);
            }
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
    double *out; (void)hipMalloc(&out, 1 << 24);
    const int iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE, CHAINS><<<blocks, 64>>>(out, 1.0, 1.0000001, 100);
    (void)hipEventRecord(e0);
    k<MODE, CHAINS><<<blocks, 64>>>(out, 1.0, 1.0000001, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double ops_per_wave = (double)iters * CHAINS;
    printf("%-4s chains %2d  blocks %5d: %.3f ms  -> %.1f ns per wave-instruction (%.1f cycles at 2.4 GHz)\n", name, CHAINS, blocks, ms, ms * 1e6 / ops_per_wave, ms * 1e6 / ops_per_wave * 2.4);
    (void)hipFree(out);
}
int main()
{
    for (int blocks : {1024, 2048, 4096}) {   // one / two / four waves per SIMD
        run<0, 1>("add", blocks); run<0, 2>("add", blocks); run<0, 4>("add", blocks); run<0, 8>("add", blocks);
        run<1, 1>("mul", blocks); run<1, 8>("mul", blocks);
        run<2, 1>("fma", blocks); run<2, 8>("fma", blocks);
    }
    return 0;
}
