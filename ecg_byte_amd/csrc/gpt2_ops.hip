// gpt2_ops.hip -- the pieces of the GPT-2 block that the Llama / Gemma blocks do not have (BASELINE config C1's model), MI355X.
//
// Reference behaviour (vendored transformers 4.46.0.dev0, paths relative to the reference root):
//   nn.LayerNorm (ln_1, ln_2, ln_f)   transformers/src/transformers/models/gpt2/modeling_gpt2.py:593-595,  eps = layer_norm_epsilon
//   Conv1D bias                       transformers/src/transformers/pytorch_utils.py:87-113 (addmm(bias, x, weight))
//   NewGELUActivation ("gelu_new")    transformers/src/transformers/activations.py:  0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))
// All HBM-bound row passes: bf16 tensors, fp32 arithmetic, 16-byte accesses, one wave per row for the normalisations.
#include <hip/hip_runtime.h>

#include <string>

#include "tokenizer.hpp"

namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) unsigned short;

__device__ __forceinline__ float bf2f(unsigned short u) { return __uint_as_float((unsigned)u << 16); }
__device__ __forceinline__ unsigned short f2bf(float f)
{
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

unsigned grid_for(size_t items, unsigned per_block)
{
    size_t b = (items + per_block - 1) / per_block;
    return (unsigned)std::max<size_t>(1, std::min<size_t>(b, 256 * 16));
}

// y = (x - mean) * rstd * w + b, one wave per row.  If `res`: x := a + res first (the residual add of the block, written to sum_out).
// mean / rstd (fp32 per row) are saved for the backward.  Two passes over the row in registers would need H known at compile time;
// the row is re-read from L1/L2 instead (H <= a few thousand).
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const unsigned short *a, const unsigned short *res, const unsigned short *w,
                                                            const unsigned short *b, unsigned short *y, unsigned short *sum_out,
                                                            float *mean_out, float *rstd_out, size_t rows, int H, float eps)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t r = wave; r < rows; r += n_waves) {
        float s1 = 0.f;
        for (int c = lane * 8; c < H; c += 64 * 8) {
            bf16x8 v = *reinterpret_cast<const bf16x8 *>(a + r * H + c);
            if (res) {
                const bf16x8 u = *reinterpret_cast<const bf16x8 *>(res + r * H + c);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = f2bf(bf2f(v[j]) + bf2f(u[j]));
                *reinterpret_cast<bf16x8 *>(sum_out + r * H + c) = v;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) s1 += bf2f(v[j]);
        }
        const float mean = wave_sum(s1) / (float)H;
        const unsigned short *px = res ? sum_out + r * H : a + r * H;
        float s2 = 0.f;
        for (int c = lane * 8; c < H; c += 64 * 8) {
            const bf16x8 v = *reinterpret_cast<const bf16x8 *>(px + c);
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = bf2f(v[j]) - mean; s2 += d * d; }
        }
        const float rs = rsqrtf(wave_sum(s2) / (float)H + eps);
        if (lane == 0) { if (mean_out) mean_out[r] = mean; if (rstd_out) rstd_out[r] = rs; }
        for (int c = lane * 8; c < H; c += 64 * 8) {
            const bf16x8 v = *reinterpret_cast<const bf16x8 *>(px + c);
            const bf16x8 g = *reinterpret_cast<const bf16x8 *>(w + c), bb = *reinterpret_cast<const bf16x8 *>(b + c);
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = f2bf((bf2f(v[j]) - mean) * rs * bf2f(g[j]) + bf2f(bb[j]));
            *reinterpret_cast<bf16x8 *>(y + r * H + c) = o;
        }
    }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)) [+ dres], g = dy * w;  dw(fp32) += sum_rows dy * xhat;  db(fp32) += sum_rows dy.
// One wave per row; a lane keeps its share of dw / db in registers over all the rows of its wave (columns lane*8 + 512 k, k < NC),
// then one LDS reduction per block and one global atomic per column per block.
template <int NC>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const unsigned short *x, const unsigned short *w, const float *mean, const float *rstd,
                                                            const unsigned short *dy, const unsigned short *dres, unsigned short *dx,
                                                            float *dw, float *db, size_t rows, int H, float *partials)
{
    extern __shared__ float s_acc[];   // 2 * H floats
    for (int c = threadIdx.x; c < 2 * H; c += blockDim.x) s_acc[c] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    float wf[NC][8], aw[NC][8], ab[NC][8];
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        const int c = k * 512 + lane * 8;
        bf16x8 vw = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        if (c < H) vw = *reinterpret_cast<const bf16x8 *>(w + c);
#pragma unroll
        for (int j = 0; j < 8; ++j) { wf[k][j] = bf2f(vw[j]); aw[k][j] = 0.f; ab[k][j] = 0.f; }
    }
    for (size_t r = wave; r < rows; r += n_waves) {
        const float mu = mean[r], rs = rstd[r];
        bf16x8 vx[NC], vg[NC];
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const int c = k * 512 + lane * 8;
            if (c < H) {
                vx[k] = *reinterpret_cast<const bf16x8 *>(x + r * H + c);
                vg[k] = *reinterpret_cast<const bf16x8 *>(dy + r * H + c);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float g = bf2f(vg[k][j]) * wf[k][j], xh = (bf2f(vx[k][j]) - mu) * rs;
                    sg += g; sgx += g * xh;
                }
            }
        }
        sg = wave_sum(sg) / (float)H;
        sgx = wave_sum(sgx) / (float)H;
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const int c = k * 512 + lane * 8;
            if (c < H) {
                bf16x8 vr = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0}, o;
                if (dres) vr = *reinterpret_cast<const bf16x8 *>(dres + r * H + c);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float gy = bf2f(vg[k][j]), xh = (bf2f(vx[k][j]) - mu) * rs;
                    float d = rs * (gy * wf[k][j] - sg - xh * sgx);
                    if (dres) d += bf2f(vr[j]);
                    o[j] = f2bf(d);
                    aw[k][j] += gy * xh;
                    ab[k][j] += gy;
                }
                *reinterpret_cast<bf16x8 *>(dx + r * H + c) = o;
            }
        }
    }
    if (partials) {   // the four waves add in wave order; the block's [dw | db] row goes to partials[block], added in block order by the caller
        for (int wv = 0; wv < 4; ++wv) {
            if ((int)(threadIdx.x >> 6) == wv) {
#pragma unroll
                for (int k = 0; k < NC; ++k) {
                    const int c = k * 512 + lane * 8;
                    if (c < H) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) { s_acc[c + j] += aw[k][j]; s_acc[H + c + j] += ab[k][j]; }
                    }
                }
            }
            __syncthreads();
        }
        for (int c = threadIdx.x; c < 2 * H; c += blockDim.x) partials[(size_t)blockIdx.x * 2 * H + c] = s_acc[c];
        return;
    }
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        const int c = k * 512 + lane * 8;
        if (c < H) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { atomicAdd(&s_acc[c + j], aw[k][j]); atomicAdd(&s_acc[H + c + j], ab[k][j]); }
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < H; c += blockDim.x) { atomicAdd(dw + c, s_acc[c]); atomicAdd(db + c, s_acc[H + c]); }
}

__device__ __forceinline__ float gelu_new_f(float x)
{
    const float k = 0.7978845608028654f;     // sqrt(2 / pi)
    return 0.5f * x * (1.0f + tanhf(k * (x + 0.044715f * x * x * x)));
}
__device__ __forceinline__ float gelu_new_grad(float x)
{
    const float k = 0.7978845608028654f;
    const float u = k * (x + 0.044715f * x * x * x), th = tanhf(u);
    return 0.5f * (1.0f + th) + 0.5f * x * (1.0f - th * th) * k * (1.0f + 3.0f * 0.044715f * x * x);
}

// u[r, c] += bias[c] in place (Conv1D's bias);  ACT: h[r, c] = gelu_new(u[r, c]) as well (u keeps the pre-activation for the backward)
template <bool ACT>
__global__ __launch_bounds__(256) void bias_act_kernel(unsigned short *u, const unsigned short *bias, unsigned short *h, size_t rows, int N)
{
    const size_t per_row = (size_t)N / 8, total = rows * per_row;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % per_row) * 8;
        bf16x8 v = reinterpret_cast<bf16x8 *>(u)[i];
        const bf16x8 b = *reinterpret_cast<const bf16x8 *>(bias + c);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v[j] = f2bf(bf2f(v[j]) + bf2f(b[j]));
            if (ACT) o[j] = f2bf(gelu_new_f(bf2f(v[j])));
        }
        reinterpret_cast<bf16x8 *>(u)[i] = v;
        if (ACT) reinterpret_cast<bf16x8 *>(h)[i] = o;
    }
}

// dpre = dh * gelu_new'(pre)
__global__ __launch_bounds__(256) void gelu_new_bwd_kernel(const unsigned short *pre, const unsigned short *dh, unsigned short *dpre, size_t n8)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        const bf16x8 p = reinterpret_cast<const bf16x8 *>(pre)[i], g = reinterpret_cast<const bf16x8 *>(dh)[i];
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = f2bf(bf2f(g[j]) * gelu_new_grad(bf2f(p[j])));
        reinterpret_cast<bf16x8 *>(dpre)[i] = o;
    }
}

// out[c] (fp32) += sum over rows of dy[r, c]: a bias gradient.  Block b takes a slab of rows; thread t the 8 columns t*8 + 2048 k.
__global__ __launch_bounds__(256) void colsum_kernel(const unsigned short *dy, float *out, size_t rows, int N, size_t rows_per_block, float *partials)
{
    const size_t r0 = (size_t)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    for (int c = threadIdx.x * 8; c < N; c += 256 * 8) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (size_t r = r0; r < r1; ++r) {
            const bf16x8 v = *reinterpret_cast<const bf16x8 *>(dy + r * N + c);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += bf2f(v[j]);
        }
        if (partials) {
#pragma unroll
            for (int j = 0; j < 8; ++j) partials[(size_t)blockIdx.x * N + c + j] = acc[j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) atomicAdd(out + c + j, acc[j]);
        }
    }
}

int launched(const char *what)
{
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return ECGB_OK;
    ecgb::set_error(std::string(what) + ": " + hipGetErrorString(e));
    return ECGB_ERR_HIP;
}

}  // namespace

extern "C" int ecgb_layernorm_fwd(const void *x_dev, const void *residual_dev, const void *w_dev, const void *b_dev, void *y_dev,
                                  void *sum_out_dev, float *mean_dev, float *rstd_dev, size_t rows, int hidden, float eps, void *stream)
{
    if (!x_dev || !w_dev || !b_dev || !y_dev || hidden <= 0 || hidden % 8 || (residual_dev && !sum_out_dev)) {
        ecgb::set_error("ecgb_layernorm_fwd: bad argument (hidden % 8 == 0; a residual needs sum_out)");
        return ECGB_ERR_INVALID;
    }
    if (rows == 0) return ECGB_OK;
    hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(grid_for(rows, 4)), dim3(256), 0, (hipStream_t)stream, (const unsigned short *)x_dev,
                       (const unsigned short *)residual_dev, (const unsigned short *)w_dev, (const unsigned short *)b_dev, (unsigned short *)y_dev,
                       (unsigned short *)sum_out_dev, mean_dev, rstd_dev, rows, hidden, eps);
    return launched("layernorm_fwd_kernel");
}

extern "C" int ecgb_partial_rows_sum_f32(const float *partials_dev, int n_rows, int n, long long ld, float *dst_dev, void *stream);

extern "C" size_t ecgb_layernorm_bwd_scratch_floats(size_t rows, int hidden)
{
    return (size_t)std::min<unsigned>(grid_for(rows, 16), 1024) * 2 * (size_t)hidden;
}

extern "C" int ecgb_layernorm_bwd(const void *x_dev, const void *w_dev, const float *mean_dev, const float *rstd_dev, const void *dy_dev,
                                  const void *dres_dev, void *dx_dev, float *dw_dev, float *db_dev, size_t rows, int hidden, float *scratch_dev,
                                  void *stream)
{
    if (!x_dev || !w_dev || !mean_dev || !rstd_dev || !dy_dev || !dx_dev || !dw_dev || !db_dev || hidden <= 0 || hidden % 8 || hidden > 4096) {
        ecgb::set_error("ecgb_layernorm_bwd: bad argument (hidden % 8 == 0, hidden <= 4096)");
        return ECGB_ERR_INVALID;
    }
    if (rows == 0) return ECGB_OK;
    const dim3 grid(std::min<unsigned>(grid_for(rows, 16), 1024));
    const size_t lds = 2 * (size_t)hidden * sizeof(float);
#define ECGB_LN(NC_) hipLaunchKernelGGL(layernorm_bwd_kernel<NC_>, grid, dim3(256), lds, (hipStream_t)stream, (const unsigned short *)x_dev, \
        (const unsigned short *)w_dev, mean_dev, rstd_dev, (const unsigned short *)dy_dev, (const unsigned short *)dres_dev, (unsigned short *)dx_dev, \
        dw_dev, db_dev, rows, hidden, scratch_dev)
    if (hidden <= 1024) ECGB_LN(2); else if (hidden <= 2048) ECGB_LN(4); else ECGB_LN(8);
#undef ECGB_LN
    if (scratch_dev) {        // per-workgroup [dw | db] rows, added in workgroup order (the same bits every launch); null: atomics
        if (int rc = ecgb_partial_rows_sum_f32(scratch_dev, (int)grid.x, hidden, 2ll * hidden, dw_dev, stream)) return rc;
        if (int rc = ecgb_partial_rows_sum_f32(scratch_dev + hidden, (int)grid.x, hidden, 2ll * hidden, db_dev, stream)) return rc;
    }
    return launched("layernorm_bwd_kernel");
}

extern "C" int ecgb_bias_act(void *u_dev, const void *bias_dev, void *h_dev, size_t rows, int n, int gelu_new, void *stream)
{
    if (!u_dev || !bias_dev || n <= 0 || n % 8 || (gelu_new && !h_dev)) { ecgb::set_error("ecgb_bias_act: bad argument (n % 8 == 0)"); return ECGB_ERR_INVALID; }
    if (rows == 0) return ECGB_OK;
    const dim3 grid(grid_for(rows * (size_t)(n / 8), 256));
    if (gelu_new) hipLaunchKernelGGL(bias_act_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, (unsigned short *)u_dev, (const unsigned short *)bias_dev, (unsigned short *)h_dev, rows, n);
    else hipLaunchKernelGGL(bias_act_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, (unsigned short *)u_dev, (const unsigned short *)bias_dev, (unsigned short *)nullptr, rows, n);
    return launched("bias_act_kernel");
}

extern "C" int ecgb_gelu_new_bwd(const void *pre_dev, const void *dh_dev, void *dpre_dev, size_t n, void *stream)
{
    if (!pre_dev || !dh_dev || !dpre_dev || n % 8) { ecgb::set_error("ecgb_gelu_new_bwd: bad argument (n % 8 == 0)"); return ECGB_ERR_INVALID; }
    if (n == 0) return ECGB_OK;
    hipLaunchKernelGGL(gelu_new_bwd_kernel, dim3(grid_for(n / 8, 256)), dim3(256), 0, (hipStream_t)stream, (const unsigned short *)pre_dev,
                       (const unsigned short *)dh_dev, (unsigned short *)dpre_dev, n / 8);
    return launched("gelu_new_bwd_kernel");
}

extern "C" size_t ecgb_colsum_scratch_floats(size_t rows, int n)
{
    return std::min<size_t>(1024, (rows + 31) / 32) * (size_t)n;
}

extern "C" int ecgb_colsum(const void *dy_dev, float *out_dev, size_t rows, int n, float *scratch_dev, void *stream)
{
    if (!dy_dev || !out_dev || n <= 0 || n % 8) { ecgb::set_error("ecgb_colsum: bad argument (n % 8 == 0)"); return ECGB_ERR_INVALID; }
    if (rows == 0) return ECGB_OK;
    const size_t blocks = std::min<size_t>(1024, (rows + 31) / 32), rpb = (rows + blocks - 1) / blocks;
    const unsigned grid = (unsigned)((rows + rpb - 1) / rpb);
    hipLaunchKernelGGL(colsum_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const unsigned short *)dy_dev, out_dev, rows, n, rpb, scratch_dev);
    if (scratch_dev)          // per-workgroup partial rows added in workgroup order; null: atomics
        if (int rc = ecgb_partial_rows_sum_f32(scratch_dev, (int)grid, n, (long long)n, out_dev, stream)) return rc;
    return launched("colsum_kernel");
}
