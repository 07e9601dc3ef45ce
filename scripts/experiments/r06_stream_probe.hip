// Round 6 probe: the memory pattern of glu_bwd (three bf16 streams read -- gate, up = gate + I of the same row, dh -- two written) with NO arithmetic, under launch / loop variants:
// what does the pattern itself allow?  T = 32 768 rows, I = 8 192 (the C3 shape: 2.68 GB per launch).
// hipcc --offload-arch=gfx950 -O3 -o /tmp/stream_probe scripts/experiments/r06_stream_probe.hip && /tmp/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
using u4 = __attribute__((ext_vector_type(4))) unsigned;
template <int U, bool NT>
__global__ __launch_bounds__(256) void k(const unsigned short *gu, const unsigned short *dh, unsigned short *dgu, size_t T, int I)
{
    const int per_row = I / 8;
    const size_t total = T * per_row, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < total; i0 += stride * U) {
        u4 g[U], u[U], d[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const size_t i = i0 + k * stride < total ? i0 + k * stride : i0;
            const size_t t = i / per_row; const int c = (int)(i % per_row) * 8;
            g[k] = *reinterpret_cast<const u4 *>(gu + t * 2 * I + c);
            u[k] = *reinterpret_cast<const u4 *>(gu + t * 2 * I + I + c);
            d[k] = *reinterpret_cast<const u4 *>(dh + t * I + c);
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const size_t i = i0 + k * stride;
            if (i >= total) break;
            const size_t t = i / per_row; const int c = (int)(i % per_row) * 8;
            const u4 a = g[k] ^ d[k], b = u[k] ^ d[k];
            if (NT) { __builtin_nontemporal_store(a, reinterpret_cast<u4 *>(dgu + t * 2 * I + c)); __builtin_nontemporal_store(b, reinterpret_cast<u4 *>(dgu + t * 2 * I + I + c)); }
            else { *reinterpret_cast<u4 *>(dgu + t * 2 * I + c) = a; *reinterpret_cast<u4 *>(dgu + t * 2 * I + I + c) = b; }
        }
    }
}
// the same bytes as two plain copies' worth of contiguous streams (what a memcpy-like kernel gets): in [5 units] -> 3 read, 2 written
template <int U>
__global__ __launch_bounds__(256) void flat(const u4 *a, const u4 *b, const u4 *c, u4 *o1, u4 *o2, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += stride * U) {
        u4 x[U], y[U], z[U];
#pragma unroll
        for (int k = 0; k < U; ++k) { const size_t i = i0 + k * stride < n ? i0 + k * stride : i0; x[k] = a[i]; y[k] = b[i]; z[k] = c[i]; }
#pragma unroll
        for (int k = 0; k < U; ++k) { const size_t i = i0 + k * stride; if (i >= n) break; o1[i] = x[k] ^ z[k]; o2[i] = y[k] ^ z[k]; }
    }
}
template <typename F> float timed(F f)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 10.f;
}
int main()
{
    const size_t T = 32768; const int I = 8192;
    unsigned short *gu, *dh, *dgu;
    hipMalloc(&gu, T * 2 * I * 2); hipMalloc(&dh, T * I * 2); hipMalloc(&dgu, T * 2 * I * 2);
    hipMemset(gu, 1, T * 2 * I * 2); hipMemset(dh, 2, T * I * 2);
    const double gb = T * I * 2.0 * 5 / 1e9;
    for (int grid : {4096, 2048, 1024, 16384, 65536}) {
        const float a = timed([&] { hipLaunchKernelGGL((k<1, false>), dim3(grid), dim3(256), 0, 0, gu, dh, dgu, T, I); });
        const float b = timed([&] { hipLaunchKernelGGL((k<2, false>), dim3(grid), dim3(256), 0, 0, gu, dh, dgu, T, I); });
        const float c = timed([&] { hipLaunchKernelGGL((k<4, false>), dim3(grid), dim3(256), 0, 0, gu, dh, dgu, T, I); });
        const float d = timed([&] { hipLaunchKernelGGL((k<1, true>), dim3(grid), dim3(256), 0, 0, gu, dh, dgu, T, I); });
        const float e = timed([&] { hipLaunchKernelGGL((k<4, true>), dim3(grid), dim3(256), 0, 0, gu, dh, dgu, T, I); });
        const size_t n = T * (size_t)I / 8;
        const float f = timed([&] { hipLaunchKernelGGL((flat<1>), dim3(grid), dim3(256), 0, 0, (const u4 *)gu, (const u4 *)gu + n, (const u4 *)dh, (u4 *)dgu, (u4 *)dgu + n, n); });
        printf("grid %5d: 1 chunk %.0f us (%.2f TB/s)  2 chunks %.0f (%.2f)  4 chunks %.0f (%.2f)  1 chunk, nt stores %.0f (%.2f)  4 chunks, nt stores %.0f (%.2f)  five flat streams %.0f (%.2f)\n",
               grid, a * 1e3, gb / a, b * 1e3, gb / b, c * 1e3, gb / c, d * 1e3, gb / d, e * 1e3, gb / e, f * 1e3, gb / f);
    }
    return 0;
}
