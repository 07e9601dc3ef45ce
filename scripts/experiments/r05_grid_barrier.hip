// Round 5 experiment: what does a dependent stage cost INSIDE one persistent kernel on MI355X, against a dependent kernel launch in a replayed graph (4.7-5 us measured in
// the decode step)?  A stage = every workgroup reads a 4 KB vector another stage wrote, does a 10-column x 2048 GEMV slice against its own weights, writes its outputs;
// between stages a grid barrier: one atomic add per workgroup on a monotonic counter, then a spin on it (bounded: a stuck barrier ends the kernel instead of hanging the box).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/gb scripts/experiments/r05_grid_barrier.hip && /tmp/gb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int kSpinMax = 1 << 22;

__device__ __forceinline__ bool grid_barrier(unsigned *counter, unsigned target)
{
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __threadfence();
        atomicAdd(counter, 1u);
        int spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) { if (++spins > kSpinMax) { ok = false; break; } }
    }
    __syncthreads();
    return ok;
}

// mode 0: barriers only; mode 1: barrier + every workgroup reads the 4 KB vector and writes 10 outputs (no weights); mode 2: + a 10 x 2048 bf16 weight slice per workgroup
__global__ __launch_bounds__(256) void persistent(unsigned *counter, float *x0, float *x1, const unsigned short *w, int stages, int mode, int *bad)
{
    __shared__ float s_part[4];
    const int G = gridDim.x;
    for (int s = 0; s < stages; ++s) {
        float *xin = (s & 1) ? x1 : x0, *xout = (s & 1) ? x0 : x1;
        if (mode >= 1) {
            // 2048-vector, device-scope loads (another CU wrote it in the stage before)
            float acc[10] = {0};
            for (int i = threadIdx.x; i < 2048; i += 256) {
                const float v = __hip_atomic_load(xin + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (mode >= 2) {
                    const unsigned short *wr = w + ((size_t)(s % 8) * G * 10 + (size_t)blockIdx.x * 10) * 2048 + i;
#pragma unroll
                    for (int c = 0; c < 10; ++c) acc[c] += v * __uint_as_float((unsigned)wr[(size_t)c * 2048] << 16);
                } else {
#pragma unroll
                    for (int c = 0; c < 10; ++c) acc[c] += v;
                }
            }
#pragma unroll
            for (int c = 0; c < 10; ++c) {
                float a = acc[c];
                for (int d = 32; d > 0; d >>= 1) a += __shfl_down(a, d, 64);
                if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = a;
                __syncthreads();
                if (threadIdx.x == 0) {
                    const int col = blockIdx.x * 10 + c;
                    if (col < 2048) __hip_atomic_store(xout + col, (s_part[0] + s_part[1] + s_part[2] + s_part[3]) * 1e-3f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __syncthreads();
            }
        }
        if (!grid_barrier(counter, (unsigned)(s + 1) * G)) { if (threadIdx.x == 0) *bad = 1; return; }
    }
}

__global__ __launch_bounds__(256) void one_stage(float *xin, float *xout, const unsigned short *w, int s, int mode)
{
    __shared__ float s_part[4];
    const int G = gridDim.x;
    float acc[10] = {0};
    for (int i = threadIdx.x; i < 2048; i += 256) {
        const float v = xin[i];
        if (mode >= 2) {
            const unsigned short *wr = w + ((size_t)(s % 8) * G * 10 + (size_t)blockIdx.x * 10) * 2048 + i;
#pragma unroll
            for (int c = 0; c < 10; ++c) acc[c] += v * __uint_as_float((unsigned)wr[(size_t)c * 2048] << 16);
        } else {
#pragma unroll
            for (int c = 0; c < 10; ++c) acc[c] += v;
        }
    }
#pragma unroll
    for (int c = 0; c < 10; ++c) {
        float a = acc[c];
        for (int d = 32; d > 0; d >>= 1) a += __shfl_down(a, d, 64);
        if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = a;
        __syncthreads();
        if (threadIdx.x == 0) { const int col = blockIdx.x * 10 + c; if (col < 2048) xout[col] = (s_part[0] + s_part[1] + s_part[2] + s_part[3]) * 1e-3f; }
        __syncthreads();
    }
}

int main()
{
    const int G = 256, stages = 200;
    unsigned *counter; float *x0, *x1; unsigned short *w; int *bad;
    hipMalloc(&counter, 4); hipMalloc(&x0, 8192); hipMalloc(&x1, 8192); hipMalloc(&bad, 4);
    const size_t wn = (size_t)8 * 512 * 10 * 2048;      // (for the larger of the two grids)
    hipMalloc(&w, wn * 2);
    std::vector<unsigned short> hw(wn, 0x3c00); hipMemcpy(w, hw.data(), wn * 2, hipMemcpyHostToDevice);
    std::vector<float> hx(2048, 1.0f); hipMemcpy(x0, hx.data(), 8192, hipMemcpyHostToDevice); hipMemcpy(x1, hx.data(), 8192, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {256, 512}) for (int mode = 0; mode < 3; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipMemset(counter, 0, 4); hipMemset(bad, 0, 4);
            hipEventRecord(e0);
            hipLaunchKernelGGL(persistent, dim3(grid), dim3(256), 0, 0, counter, x0, x1, w, stages, mode, bad);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
        }
        int hb; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
        printf("persistent kernel, %d workgroups, mode %d: %.2f us per stage%s\n", grid, mode, best * 1000.f / stages, hb ? "  (A BARRIER GAVE UP)" : "");
    }
    // the same stages as dependent kernels in a replayed graph
    for (int mode = 1; mode < 3; ++mode) {
        hipStream_t st; hipStreamCreate(&st);
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
        for (int s = 0; s < stages; ++s) hipLaunchKernelGGL(one_stage, dim3(G), dim3(256), 0, st, (s & 1) ? x1 : x0, (s & 1) ? x0 : x1, w, s, mode);
        hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0, st); hipGraphLaunch(ge, st); hipEventRecord(e1, st); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
        }
        printf("graph of dependent kernels, %d workgroups, mode %d: %.2f us per stage\n", G, mode, best * 1000.f / stages);
    }
    return 0;
}
