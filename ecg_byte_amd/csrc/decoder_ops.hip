// decoder_ops.hip -- the HBM-bound pieces of the causal-LM step on MI355X (gfx950): embedding,
// RMSNorm, RoPE, SwiGLU, residual add, cross-entropy, Adam, gradient-norm; forward and backward.
// bf16 tensors in HBM, fp32 arithmetic in registers, 16-byte (8 x bf16) accesses.
//
// Reference behaviour (vendored transformers 4.46.0.dev0, paths relative to the reference root):
//   LlamaRMSNorm.forward              transformers/src/transformers/models/llama/modeling_llama.py:67-72
//   apply_rotary_pos_emb/rotate_half  modeling_llama.py:193-224 (half-split layout)
//   LlamaMLP.forward (SiLU gate)      modeling_llama.py:238-258
//   ForCausalLMLoss                   transformers/src/transformers/loss/loss_utils.py:32-47
//   Adam(weight_decay = L2)           ecg_byte/main.py:262-264 ; clip_grad_norm_ ecg_byte/runners/train.py:26
#include <hip/hip_bf16.h>
#include <hip/hip_runtime.h>
#include <atomic>

#include <string>

#include "glu_math.hpp"
#include "tokenizer.hpp"

namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) unsigned short;

__device__ __forceinline__ float bf2f(unsigned short u) { return __uint_as_float((unsigned)u << 16); }
__device__ __forceinline__ unsigned short f2bf(float f)
{
    __hip_bfloat16 b = __float2bfloat16(f);   // round to nearest even, NaN stays NaN
    return *reinterpret_cast<unsigned short *>(&b);
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
    return v;
}

int ok_or(hipError_t e, const char *what)
{
    if (e == hipSuccess) return ECGB_OK;
    ecgb::set_error(std::string(what) + ": " + hipGetErrorString(e));
    return ECGB_ERR_HIP;
}

unsigned grid_for(size_t items, unsigned per_block)
{
    size_t b = (items + per_block - 1) / per_block;
    return (unsigned)std::max<size_t>(1, std::min<size_t>(b, 256 * 16));
}

// Workgroups of the element-wise streams over [tokens, inter] (glu_fwd / glu_bwd): round 6 measured the PATTERN alone (three streams read, two written, no
// arithmetic: scripts/experiments/r06_stream_probe.hip) at 4.7 TB/s with 4 096 workgroups walking the tensor in a grid-stride loop and 5.6 TB/s with 65 536 of
// them, two trips each -- many short workgroups keep the mix of reads and writes at the memory side even; the kernels themselves are the same.
size_t g_stream_grid_cap = (size_t)1 << 20;
unsigned stream_grid(size_t items, unsigned per_block)
{
    size_t b = (items + per_block - 1) / per_block;
    return (unsigned)std::max<size_t>(1, std::min<size_t>(b, g_stream_grid_cap));
}

// ---- greedy token choice ---------------------------------------------------------------------
// out[r] = index of the first maximum of x[r, 0:n] (torch.argmax's rule for ties; a NaN counts as the maximum, as it does there).  One workgroup per row, no
// workspace: torch's own argmax over a 260 000-wide row is a two-pass reduction over a semaphore buffer it allocates per call -- three launches with the
// slice and the fp32 copy in front of it, and state a replayed HIP graph must not depend on.
__device__ __forceinline__ unsigned argmax_key(unsigned short b)
{
    const unsigned u = b;
    if ((u & 0x7FFFu) > 0x7F80u) return 0xFFFFu;            // NaN: above everything
    return (u & 0x8000u) ? (0x7FFFu - (u & 0x7FFFu)) : (0x8000u | u);   // monotone in the bf16 value (-0 just below +0: torch compares them equal -- first index wins there, see below)
}
// One row over `parts` workgroups (a decode step's single row of 256 000 logits through ONE workgroup was 30 us: 512 KB through one CU, four memory round trips;
// gridDim.y = parts).  Each part's best key goes into the row's slot of g_argmax_key by atomic max -- the maximum of the keys is what it is whatever the order --
// and the part that arrives last writes the index and leaves slot and ticket as it found them (zero) for the next launch.
constexpr int kArgmaxSlots = 4096;
__device__ unsigned long long g_argmax_key[kArgmaxSlots];
__device__ unsigned g_argmax_ticket[kArgmaxSlots];
__global__ __launch_bounds__(1024) void argmax_rows_kernel(const unsigned short *x, long long ld, int n, long long *out, int slot0)
{
    __shared__ unsigned long long s_best[16];
    const unsigned short *row = x + (long long)blockIdx.x * ld;
    const int parts = (int)gridDim.y, part = (int)blockIdx.y;
    // one 64-bit key per candidate: value key above, ~index below -- the maximum of the keys is the greatest value at the lowest index
    unsigned long long best = 0;
    auto take = [&](unsigned short b, int i) {
        unsigned k = argmax_key(b);
        if (k == 0x7FFFu) k = 0x8000u;                        // -0 == +0
        const unsigned long long key = ((unsigned long long)k << 32) | (unsigned)(0x7FFFFFFF - i);
        best = key > best ? key : best;
    };
    const int head = (int)((16 - ((uintptr_t)row & 15)) & 15) / 2;     // elements before the first 16-byte boundary (rows of an odd-width matrix)
    if (part == 0) for (int i = threadIdx.x; i < min(head, n); i += 1024) take(row[i], i);
    const int body = n > head ? (n - head) / 8 : 0;
    const int per = (body + parts - 1) / parts, b_lo = min(body, part * per), b_hi = min(body, b_lo + per);      // this part's 16-byte pieces
    constexpr int U = 8;                                      // 16-byte pieces in flight per thread
    for (int c0 = b_lo + (int)threadIdx.x; c0 < b_hi; c0 += 1024 * U) {
        bf16x8 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const bf16x8 *>(row + head + (long long)min(c0 + 1024 * u, b_hi - 1) * 8);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = c0 + 1024 * u;
            if (c < b_hi) {
#pragma unroll
                for (int j = 0; j < 8; ++j) take((unsigned short)v[u][j], head + c * 8 + j);
            }
        }
    }
    if (part == parts - 1) for (int i = head + body * 8 + threadIdx.x; i < n; i += 1024) take(row[i], i);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const unsigned long long o = __shfl_xor(best, d, 64);
        best = o > best ? o : best;
    }
    if ((threadIdx.x & 63) == 0) s_best[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < 16; ++w) best = s_best[w] > best ? s_best[w] : best;
        if (parts == 1) { out[blockIdx.x] = (long long)(0x7FFFFFFF - (unsigned)(best & 0xFFFFFFFFu)); return; }
        const int slot = slot0 + (int)blockIdx.x;
        atomicMax(&g_argmax_key[slot], best);
        __threadfence();
        if (atomicAdd(&g_argmax_ticket[slot], 1u) == (unsigned)parts - 1u) {                 // the last part in
            const unsigned long long all = atomicExch(&g_argmax_key[slot], 0ull);
            g_argmax_ticket[slot] = 0u;
            out[blockIdx.x] = (long long)(0x7FFFFFFF - (unsigned)(all & 0xFFFFFFFFu));
        }
    }
}

// ---- embedding -------------------------------------------------------------------------------
// out[t, :] = table[ids[t], :] * scale   (scale = 1 for Llama; sqrt(hidden) for Gemma)
__global__ __launch_bounds__(256) void embed_fwd_kernel(const long long *ids, const unsigned short *table, unsigned short *out,
                                                        size_t T, int H, float scale)
{
    const int per_row = H / 8;
    const size_t total = T * per_row;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t t = i / per_row;
        const int c = (int)(i % per_row);
        bf16x8 v = *reinterpret_cast<const bf16x8 *>(table + (size_t)ids[t] * H + c * 8);
        if (scale != 1.0f) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = f2bf(bf2f(v[j]) * scale);
        }
        *reinterpret_cast<bf16x8 *>(out + t * H + c * 8) = v;
    }
}

// grad_table(fp32)[ids[t], :] += dout[t, :] * scale
__global__ __launch_bounds__(256) void embed_bwd_kernel(const long long *ids, const unsigned short *dout, float *grad_table,
                                                        size_t T, int H, float scale)
{
    const int per_row = H / 8;
    const size_t total = T * per_row;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t t = i / per_row;
        const int c = (int)(i % per_row);
        const bf16x8 v = *reinterpret_cast<const bf16x8 *>(dout + t * H + c * 8);
        float *g = grad_table + (size_t)ids[t] * H + c * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) atomicAdd(g + j, bf2f(v[j]) * scale);
    }
}

// table(bf16)[id, :] += scale * sum of dout[t, :] over the tokens t with ids[t] == id, WITHOUT atomics and without an fp32 table: the tokens
// arrive sorted by id (ids_sorted, order = the stable argsort), workgroup i owns the run of equal ids that STARTS at sorted position i (the
// others return at once) and adds the run's rows in sorted order, in fp32, on top of what the table row already holds (the tied lm_head's
// gradient) -- the same bits every launch.  skip_id: nn.Embedding's padding_idx, whose row takes no gradient from the lookup
// (modeling_llama.py:889); it is also by far the longest run of a left-padded batch.  Negative ids are skipped too: the callers replace the
// id of every position whose attention mask is 0 by -1 (the gradient of such a row is exactly zero), so that the padding run -- thousands of
// rows that ONE workgroup would sum serially at the tail of the backward -- costs nothing whether or not the model declares a padding_idx.
__global__ __launch_bounds__(256) void embed_bwd_sorted_kernel(const long long *ids_sorted, const long long *order, const unsigned short *dout,
                                                               unsigned short *table, size_t T, int H, float scale, long long skip_id)
{
    const size_t i = blockIdx.x;
    const long long id = ids_sorted[i];
    if (id == skip_id || id < 0 || (i > 0 && ids_sorted[i - 1] == id)) return;   // id < 0: a row the caller masked out (attention mask 0: its gradient is exactly zero)
    size_t end = i + 1;
    while (end < T && ids_sorted[end] == id) ++end;
    for (int c = threadIdx.x * 8; c < H; c += 256 * 8) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        size_t j = i;
        for (; j + 4 <= end; j += 4) {                       // four rows in flight
            bf16x8 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const bf16x8 *>(dout + (size_t)order[j + u] * H + c);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] += bf2f(v[u][k]);
        }
        for (; j < end; ++j) {
            const bf16x8 v = *reinterpret_cast<const bf16x8 *>(dout + (size_t)order[j] * H + c);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += bf2f(v[k]);
        }
        unsigned short *dst = table + (size_t)id * H + c;
        bf16x8 o = *reinterpret_cast<const bf16x8 *>(dst);
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = f2bf(bf2f(o[k]) + acc[k] * scale);
        *reinterpret_cast<bf16x8 *>(dst) = o;
    }
}

// ---- RMSNorm ---------------------------------------------------------------------------------
// One wave per row.  y = w * bf16(x * rsqrt(mean(x^2) + eps))  (Llama; gemma: (1 + w), fp32 product)
// Optionally fuses the residual add that precedes it: x = a + b is written to `sum_out`.
template <bool GEMMA>
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const unsigned short *a, const unsigned short *b,
                                                          const unsigned short *w, unsigned short *y,
                                                          unsigned short *sum_out, float *rstd, size_t rows, int H, float eps)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t r = wave; r < rows; r += n_waves) {
        const unsigned short *pa = a + r * H;
        float ss = 0.f;
        for (int c = lane * 8; c < H; c += 64 * 8) {
            bf16x8 v = *reinterpret_cast<const bf16x8 *>(pa + c);
            if (b) {
                const bf16x8 u = *reinterpret_cast<const bf16x8 *>(b + r * H + c);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = f2bf(bf2f(v[j]) + bf2f(u[j]));
                *reinterpret_cast<bf16x8 *>(sum_out + r * H + c) = v;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float f = bf2f(v[j]); ss += f * f; }
        }
        ss = wave_sum(ss);
        const float rs = rsqrtf(ss / (float)H + eps);
        if (lane == 0 && rstd) rstd[r] = rs;
        const unsigned short *px = b ? sum_out + r * H : pa;
        for (int c = lane * 8; c < H; c += 64 * 8) {
            const bf16x8 v = *reinterpret_cast<const bf16x8 *>(px + c);
            const bf16x8 g = *reinterpret_cast<const bf16x8 *>(w + c);
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (GEMMA) o[j] = f2bf(bf2f(v[j]) * rs * (1.0f + bf2f(g[j])));
                else o[j] = f2bf(bf2f(f2bf(bf2f(v[j]) * rs)) * bf2f(g[j]));
            }
            *reinterpret_cast<bf16x8 *>(y + r * H + c) = o;
        }
    }
}

// rmsnorm_fwd_kernel for a decode step with adapters: n_a / 4 workgroups per row, each redoes the norm (4 KiB) and takes four rows of A; after the normalised row y is in LDS, its product with the site's stacked LoRA
// down-projection follows in the same launch -- t[r] = bf16(scale * y . A[r, :]) for the n_a rows of A ([n_a, H] bf16).  The lanes walk y and A[r] in the pieces
// and the order of the few-row GEMM (gemm_nt_skinny_kernel: lane l takes the 16-byte pieces at 8 l + 512 u, sums them in ascending k, then the butterfly), so t is
// bit for bit what ecgb_gemm_nt_bf16(y, A, alpha = scale) returns: one launch less per adapter site and token.
template <bool GEMMA>
__global__ __launch_bounds__(256) void rmsnorm_lora_fwd_kernel(const unsigned short *a, const unsigned short *b, const unsigned short *w, unsigned short *y,
                                                               unsigned short *sum_out, float *rstd, int H, float eps, const unsigned short *lora_a,
                                                               long long lda, int n_a, float lora_scale, unsigned short *t_out, long long ldt)
{
    extern __shared__ __align__(16) unsigned short s_y[];     // the normalised row
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t r = blockIdx.x;
    const bool first = blockIdx.y == 0;                        // the workgroup that also writes y, the residual sum and rstd
    // This wave's row of A and (wave 0) the norm's weights are asked for before anything else: behind the norm and its barrier they were a third memory round trip
    // of a kernel that is nothing but round trips (7.0 us a launch in the decode step, 4.7 of it the launch).  H <= 4096: eight pieces a lane.
    constexpr int kPieces = 8;
    const int ra_row = blockIdx.y * 4 + wave;
    bf16x8 apre[kPieces], gpre[kPieces];
    const bool pre = H <= kPieces * 512;
    if (pre) {
#pragma unroll
        for (int u = 0; u < kPieces; ++u) {
            const int k = lane * 8 + u * 512;
            if (k < H && ra_row < n_a) apre[u] = *reinterpret_cast<const bf16x8 *>(lora_a + (long long)ra_row * lda + k);
            if (k < H && wave == 0) gpre[u] = *reinterpret_cast<const bf16x8 *>(w + k);
        }
    }
    if (wave == 0) {                                           // (rmsnorm_fwd_kernel's arithmetic)
        const unsigned short *pa = a + r * H;
        float ss = 0.f;
        for (int c = lane * 8; c < H; c += 64 * 8) {
            bf16x8 v = *reinterpret_cast<const bf16x8 *>(pa + c);
            if (b) {
                const bf16x8 u = *reinterpret_cast<const bf16x8 *>(b + r * H + c);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = f2bf(bf2f(v[j]) + bf2f(u[j]));
                if (first) *reinterpret_cast<bf16x8 *>(sum_out + r * H + c) = v;
                *reinterpret_cast<bf16x8 *>(s_y + c) = v;     // (the summed row waits here for the second sweep)
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float f = bf2f(v[j]); ss += f * f; }
        }
        ss = wave_sum(ss);
        const float rs = rsqrtf(ss / (float)H + eps);
        if (first && lane == 0 && rstd) rstd[r] = rs;
#pragma unroll
        for (int u = 0; u < kPieces; ++u) {
            const int c = lane * 8 + u * 512;
            if (c >= H && pre) break;
            if (c >= H) break;
            const bf16x8 v = b ? *reinterpret_cast<const bf16x8 *>(s_y + c) : *reinterpret_cast<const bf16x8 *>(pa + c);
            const bf16x8 g = pre ? gpre[u] : *reinterpret_cast<const bf16x8 *>(w + c);
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (GEMMA) o[j] = f2bf(bf2f(v[j]) * rs * (1.0f + bf2f(g[j])));
                else o[j] = f2bf(bf2f(f2bf(bf2f(v[j]) * rs)) * bf2f(g[j]));
            }
            if (first) *reinterpret_cast<bf16x8 *>(y + r * H + c) = o;
            *reinterpret_cast<bf16x8 *>(s_y + c) = o;
        }
        for (int c = lane * 8 + kPieces * 512; c < H; c += 64 * 8) {      // (rows longer than the pieces held: as before)
            const bf16x8 v = b ? *reinterpret_cast<const bf16x8 *>(s_y + c) : *reinterpret_cast<const bf16x8 *>(pa + c);
            const bf16x8 g = *reinterpret_cast<const bf16x8 *>(w + c);
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (GEMMA) o[j] = f2bf(bf2f(v[j]) * rs * (1.0f + bf2f(g[j])));
                else o[j] = f2bf(bf2f(f2bf(bf2f(v[j]) * rs)) * bf2f(g[j]));
            }
            if (first) *reinterpret_cast<bf16x8 *>(y + r * H + c) = o;
            *reinterpret_cast<bf16x8 *>(s_y + c) = o;
        }
    }
    __syncthreads();
    {
        const int ra = ra_row;                              // one row of A per wave: the workgroups of a row (grid.y) redo the norm and share out the rows of A
        if (ra >= n_a) return;
        const unsigned short *pa = lora_a + (long long)ra * lda;
        float acc = 0.f;
        if (pre) {
#pragma unroll
            for (int u = 0; u < kPieces; ++u) {
                const int k = lane * 8 + u * 512;
                if (k >= H) break;
                const bf16x8 vb = apre[u];
                const bf16x8 va = *reinterpret_cast<const bf16x8 *>(s_y + k);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc += bf2f(va[j]) * bf2f(vb[j]);
            }
        } else
        for (int k = lane * 8; k < H; k += 512) {
            const bf16x8 vb = *reinterpret_cast<const bf16x8 *>(pa + k);
            const bf16x8 va = *reinterpret_cast<const bf16x8 *>(s_y + k);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += bf2f(va[j]) * bf2f(vb[j]);
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d, 64);
        if (lane == 0) t_out[r * ldt + ra] = f2bf(acc * lora_scale);
    }
}

// dx = rs * (dy*w' - xhat * mean(dy*w'*xhat)) [+ dres];   dw(fp32) += sum_rows dy * xhat
// w' = w (Llama) or 1 + w (Gemma).  One wave per row; dw accumulated per block in LDS then atomics.
template <bool GEMMA>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const unsigned short *x, const unsigned short *w,
                                                          const float *rstd, const unsigned short *dy,
                                                          const unsigned short *dres, unsigned short *dx, float *dw,
                                                          size_t rows, int H, float *partials)
{
    extern __shared__ float s_dw[];   // H floats
    for (int c = threadIdx.x; c < H; c += blockDim.x) s_dw[c] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t r = wave; r < rows; r += n_waves) {
        const float rs = rstd[r];
        float dot = 0.f;
        for (int c = lane * 8; c < H; c += 64 * 8) {
            const bf16x8 vx = *reinterpret_cast<const bf16x8 *>(x + r * H + c);
            const bf16x8 vg = *reinterpret_cast<const bf16x8 *>(dy + r * H + c);
            const bf16x8 vw = *reinterpret_cast<const bf16x8 *>(w + c);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float wf = GEMMA ? 1.0f + bf2f(vw[j]) : bf2f(vw[j]);
                dot += bf2f(vg[j]) * wf * bf2f(vx[j]) * rs;
            }
        }
        dot = wave_sum(dot) / (float)H;
        for (int c = lane * 8; c < H; c += 64 * 8) {
            const bf16x8 vx = *reinterpret_cast<const bf16x8 *>(x + r * H + c);
            const bf16x8 vg = *reinterpret_cast<const bf16x8 *>(dy + r * H + c);
            const bf16x8 vw = *reinterpret_cast<const bf16x8 *>(w + c);
            bf16x8 vr;
            if (dres) vr = *reinterpret_cast<const bf16x8 *>(dres + r * H + c);
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xh = bf2f(vx[j]) * rs;
                const float g = bf2f(vg[j]);
                const float wf = GEMMA ? 1.0f + bf2f(vw[j]) : bf2f(vw[j]);
                float d = rs * (g * wf - xh * dot);
                if (dres) d += bf2f(vr[j]);
                o[j] = f2bf(d);
                atomicAdd(&s_dw[c + j], g * xh);
            }
            *reinterpret_cast<bf16x8 *>(dx + r * H + c) = o;
        }
    }
    __syncthreads();
    if (!dw) return;   // frozen norm weights: only the input gradient was wanted
    if (partials) {   // launched with ONE wave per workgroup then: the LDS adds above happened in program order; the block's row is added in block order
        for (int c = threadIdx.x; c < H; c += blockDim.x) partials[(size_t)blockIdx.x * H + c] = s_dw[c];
        return;
    }
    for (int c = threadIdx.x; c < H; c += blockDim.x) atomicAdd(dw + c, s_dw[c]);
}

// The same for H = NC * 512 (2048, 4096): a lane owns NC runs of 8 columns, keeps the row's x and dy in registers (one
// pass over memory) and its share of dw in fp32 registers across all the rows its wave processes; LDS and global atomics
// happen once per wave / block instead of once per element (the kernel above spends its time in LDS atomics: 1.1 TB/s).
template <bool GEMMA, int NC>
__global__ __launch_bounds__(256) void rmsnorm_bwd_rows_kernel(const unsigned short *x, const unsigned short *w,
                                                               const float *rstd, const unsigned short *dy,
                                                               const unsigned short *dres, unsigned short *dx, float *dw,
                                                               size_t rows, float *partials)
{
    constexpr int H = NC * 512;
    __shared__ float s_dw[H];
    for (int c = threadIdx.x; c < H; c += blockDim.x) s_dw[c] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    float wf[NC][8], acc[NC][8];
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        const bf16x8 vw = *reinterpret_cast<const bf16x8 *>(w + k * 512 + lane * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) { wf[k][j] = GEMMA ? 1.0f + bf2f(vw[j]) : bf2f(vw[j]); acc[k][j] = 0.f; }
    }
    for (size_t r = wave; r < rows; r += n_waves) {
        const float rs = rstd[r];
        bf16x8 vx[NC], vg[NC], vr[NC];
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            vx[k] = *reinterpret_cast<const bf16x8 *>(x + r * H + k * 512 + lane * 8);
            vg[k] = *reinterpret_cast<const bf16x8 *>(dy + r * H + k * 512 + lane * 8);
            if (dres) vr[k] = *reinterpret_cast<const bf16x8 *>(dres + r * H + k * 512 + lane * 8);
        }
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < NC; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) dot += bf2f(vg[k][j]) * wf[k][j] * bf2f(vx[k][j]) * rs;
        dot = wave_sum(dot) / (float)H;
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xh = bf2f(vx[k][j]) * rs;
                const float g = bf2f(vg[k][j]);
                float d = rs * (g * wf[k][j] - xh * dot);
                if (dres) d += bf2f(vr[k][j]);
                o[j] = f2bf(d);
                acc[k][j] += g * xh;
            }
            *reinterpret_cast<bf16x8 *>(dx + r * H + k * 512 + lane * 8) = o;
        }
    }
    if (!dw) return;                                         // frozen norm weights (LoRA): only the input gradient was wanted
    if (partials) {                                          // the four waves add in wave order, the block's row goes to partials[block]
        for (int w = 0; w < 4; ++w) {
            if ((int)(threadIdx.x >> 6) == w) {
#pragma unroll
                for (int k = 0; k < NC; ++k)
#pragma unroll
                    for (int j = 0; j < 8; ++j) s_dw[k * 512 + lane * 8 + j] += acc[k][j];
            }
            __syncthreads();
        }
        for (int c = threadIdx.x; c < H; c += blockDim.x) partials[(size_t)blockIdx.x * H + c] = s_dw[c];
        return;
    }
#pragma unroll
    for (int k = 0; k < NC; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) atomicAdd(&s_dw[k * 512 + lane * 8 + j], acc[k][j]);
    __syncthreads();
    for (int c = threadIdx.x; c < H; c += blockDim.x) atomicAdd(dw + c, s_dw[c]);
}

// ---- RoPE --------------------------------------------------------------------------------------
// x: [T, n_heads, D] (token-major, heads contiguous); cos/sin: [T, D/2] fp32.  In place.
// forward:  (x1, x2) -> (x1*c - x2*s, x2*c + x1*s);  backward (INVERSE): (g1*c + g2*s, g2*c - g1*s)
template <bool INVERSE>
__global__ __launch_bounds__(256) void rope_kernel(unsigned short *x, const float *cs, const float *sn, size_t T,
                                                   int n_heads, int D, size_t row_stride)
{
    const int half = D / 2;
    const size_t total = T * n_heads * (half / 8 > 0 ? half / 8 : 1);
    const int per_head = half / 8;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % per_head);
        const size_t th = i / per_head;
        const int h = (int)(th % n_heads);
        const size_t t = th / n_heads;
        unsigned short *p = x + t * row_stride + (size_t)h * D + c * 8;
        bf16x8 a = *reinterpret_cast<bf16x8 *>(p), b = *reinterpret_cast<bf16x8 *>(p + half);
        const float *pc = cs + t * half + c * 8, *ps = sn + t * half + c * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            // HF holds cos/sin in the activation dtype (bf16): modeling_llama.py:163-165
            const float cc = bf2f(f2bf(pc[j])), ss = bf2f(f2bf(ps[j]));
            const float x1 = bf2f(a[j]), x2 = bf2f(b[j]);
            if (!INVERSE) { a[j] = f2bf(x1 * cc - x2 * ss); b[j] = f2bf(x2 * cc + x1 * ss); }
            else { a[j] = f2bf(x1 * cc + x2 * ss); b[j] = f2bf(x2 * cc - x1 * ss); }
        }
        *reinterpret_cast<bf16x8 *>(p) = a;
        *reinterpret_cast<bf16x8 *>(p + half) = b;
    }
}

// A decode step's RoPE and KV-cache append in one launch: the new token's q and k heads are rotated in place in qkv (rope_kernel's arithmetic), and its rotated k and its v
// go to row len - 1 of the cache (k | v, row stride `width` = 2 * n_kv * D) -- ecgb_rope followed by the cache copy / ecgb_kv_append, the same bits.
// qkv [B, (n_q + 2 n_kv) * D]; len: *len_dev when given (a replayed graph), else len_arg.
__global__ __launch_bounds__(256) void rope_append_kernel(unsigned short *qkv, const float *cs, const float *sn, int B, int n_q, int n_kv, int D, size_t row_stride,
                                                          unsigned short *cache, long long cap, int len_arg, const int *len_dev)
{
    const int half = D / 2, per_head = half / 8;
    const long long row = (len_dev ? *len_dev : len_arg) - 1;
    const int width = 2 * n_kv * D;
    const int n_rope = B * (n_q + n_kv) * per_head, n_v = B * (n_kv * D / 8);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_rope + n_v; i += gridDim.x * blockDim.x) {
        if (i < n_rope) {
            const int c = i % per_head, th = i / per_head, h = th % (n_q + n_kv), t = th / (n_q + n_kv);
            unsigned short *p = qkv + (size_t)t * row_stride + (size_t)h * D + c * 8;
            bf16x8 a = *reinterpret_cast<bf16x8 *>(p), b = *reinterpret_cast<bf16x8 *>(p + half);
            const float *pc = cs + (size_t)t * half + c * 8, *ps = sn + (size_t)t * half + c * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float cc = bf2f(f2bf(pc[j])), ss = bf2f(f2bf(ps[j]));
                const float x1 = bf2f(a[j]), x2 = bf2f(b[j]);
                a[j] = f2bf(x1 * cc - x2 * ss); b[j] = f2bf(x2 * cc + x1 * ss);
            }
            *reinterpret_cast<bf16x8 *>(p) = a;
            *reinterpret_cast<bf16x8 *>(p + half) = b;
            if (h >= n_q) {
                unsigned short *kc = cache + ((long long)t * cap + row) * width + (size_t)(h - n_q) * D + c * 8;
                *reinterpret_cast<bf16x8 *>(kc) = a;
                *reinterpret_cast<bf16x8 *>(kc + half) = b;
            }
        } else {
            const int j = i - n_rope, per_row = n_kv * D / 8, t = j / per_row, c = j % per_row;
            *reinterpret_cast<bf16x8 *>(cache + ((long long)t * cap + row) * width + n_kv * D + c * 8) =
                *reinterpret_cast<const bf16x8 *>(qkv + (size_t)t * row_stride + (size_t)(n_q + n_kv) * D + c * 8);
        }
    }
}

// ---- SwiGLU / GeGLU ------------------------------------------------------------------------------
// gu: [T, 2*I] with gate in [:, :I] and up in [:, I:]  (one fused gate|up projection);  h: [T, I]
template <bool GELU_TANH>
__device__ __forceinline__ float act(float g) { return ecgb::glu_act<GELU_TANH>(g); }
template <bool GELU_TANH>
__device__ __forceinline__ float act_grad(float g) { return ecgb::glu_act_grad<GELU_TANH>(g); }

template <bool GELU_TANH>
__global__ __launch_bounds__(256) void glu_fwd_kernel(const unsigned short *gu, unsigned short *h, size_t T, int I)
{
    const int per_row = I / 8;
    const size_t total = T * per_row;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t t = i / per_row;
        const int c = (int)(i % per_row) * 8;
        const bf16x8 g = *reinterpret_cast<const bf16x8 *>(gu + t * 2 * I + c);
        const bf16x8 u = *reinterpret_cast<const bf16x8 *>(gu + t * 2 * I + I + c);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = f2bf(bf2f(f2bf(act<GELU_TANH>(bf2f(g[j])))) * bf2f(u[j]));
        *reinterpret_cast<bf16x8 *>(h + t * I + c) = o;
    }
}

template <bool GELU_TANH>
__global__ __launch_bounds__(256) void glu_bwd_kernel(const unsigned short *gu, const unsigned short *dh, unsigned short *dgu,
                                                      size_t T, int I)
{
    const int per_row = I / 8;
    const size_t total = T * per_row;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t t = i / per_row;
        const int c = (int)(i % per_row) * 8;
        const bf16x8 g = *reinterpret_cast<const bf16x8 *>(gu + t * 2 * I + c);
        const bf16x8 u = *reinterpret_cast<const bf16x8 *>(gu + t * 2 * I + I + c);
        const bf16x8 d = *reinterpret_cast<const bf16x8 *>(dh + t * I + c);
        bf16x8 og, ou;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float gf = bf2f(g[j]), uf = bf2f(u[j]), df = bf2f(d[j]);
            og[j] = f2bf(df * uf * act_grad<GELU_TANH>(gf));
            ou[j] = f2bf(df * act<GELU_TANH>(gf));
        }
        *reinterpret_cast<bf16x8 *>(dgu + t * 2 * I + c) = og;
        *reinterpret_cast<bf16x8 *>(dgu + t * 2 * I + I + c) = ou;
    }
}

// ---- dropout (LoRA input dropout, peft LoraLayer) ---------------------------------------------------
__device__ __forceinline__ unsigned mix32(unsigned long long seed, unsigned long long idx)
{
    unsigned long long z = seed + idx * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (unsigned)((z ^ (z >> 31)) >> 32);
}
__global__ __launch_bounds__(256) void dropout_kernel(const unsigned short *x, unsigned short *o, size_t n8, float p, float inv_keep,
                                                      unsigned long long seed)
{
    const unsigned thr = (unsigned)(p * 4294967296.0);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        bf16x8 v = reinterpret_cast<const bf16x8 *>(x)[i];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool keep = mix32(seed, i * 8 + j) >= thr;
            v[j] = keep ? f2bf(bf2f(v[j]) * inv_keep) : (unsigned short)0;
        }
        reinterpret_cast<bf16x8 *>(o)[i] = v;
    }
}

// ---- elementwise add (residual) -----------------------------------------------------------------
__global__ __launch_bounds__(256) void add_kernel(const unsigned short *a, const unsigned short *b, unsigned short *o, size_t n8)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        const bf16x8 x = reinterpret_cast<const bf16x8 *>(a)[i], y = reinterpret_cast<const bf16x8 *>(b)[i];
        bf16x8 z;
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] = f2bf(bf2f(x[j]) + bf2f(y[j]));
        reinterpret_cast<bf16x8 *>(o)[i] = z;
    }
}

// ---- cross-entropy over a chunk of rows ------------------------------------------------------------
// logits: [rows, V] bf16 (upcast to fp32, loss_utils.py:36).  labels[r] = target of row r or -100.
// Writes per-row loss (0 for ignored rows), accumulates sum_loss; overwrites logits IN PLACE with
// dlogits = (softmax - onehot) * inv_count  (0 for ignored rows).  One workgroup per row.
__global__ __launch_bounds__(256) void ce_fwd_bwd_kernel(unsigned short *logits, const long long *labels, float *row_loss,
                                                         float *sum_loss, const float *inv_count_ptr, size_t rows, int V,
                                                         size_t ld)
{
    __shared__ float s_red[4];

    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (size_t r = blockIdx.x; r < rows; r += gridDim.x) {
        unsigned short *p = logits + r * ld;
        const long long lab = labels[r];
        const bool valid = lab >= 0 && lab < V;
        const int V8 = V & ~7;
        float m = -INFINITY;
        for (int c = threadIdx.x * 8; c < V8; c += 256 * 8) {
            const bf16x8 v = *reinterpret_cast<const bf16x8 *>(p + c);
#pragma unroll
            for (int j = 0; j < 8; ++j) m = fmaxf(m, bf2f(v[j]));
        }
        for (int c = V8 + threadIdx.x; c < V; c += 256) m = fmaxf(m, bf2f(p[c]));
        m = wave_max(m);
        if (lane == 0) s_red[wv] = m;
        __syncthreads();
        m = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
        __syncthreads();
        float s = 0.f;
        for (int c = threadIdx.x * 8; c < V8; c += 256 * 8) {
            const bf16x8 v = *reinterpret_cast<const bf16x8 *>(p + c);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += __expf(bf2f(v[j]) - m);
        }
        for (int c = V8 + threadIdx.x; c < V; c += 256) s += __expf(bf2f(p[c]) - m);
        s = wave_sum(s);
        if (lane == 0) s_red[wv] = s;
        __syncthreads();
        s = s_red[0] + s_red[1] + s_red[2] + s_red[3];
        const float lse = m + logf(s);
        const float inv_count = *inv_count_ptr;
        if (threadIdx.x == 0) {
            const float l = valid ? lse - bf2f(p[lab]) : 0.f;
            row_loss[r] = l;
            if (valid && sum_loss) atomicAdd(sum_loss, l * inv_count);      // (null: the caller sums row_loss in order)
        }
        __syncthreads();
        const float scale = valid ? inv_count : 0.f;
        for (int c = threadIdx.x * 8; c < V8; c += 256 * 8) {
            bf16x8 v = *reinterpret_cast<const bf16x8 *>(p + c);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float g = __expf(bf2f(v[j]) - lse);
                if (c + j == lab) g -= 1.f;
                v[j] = f2bf(g * scale);
            }
            *reinterpret_cast<bf16x8 *>(p + c) = v;
        }
        for (int c = V8 + threadIdx.x; c < V; c += 256) {
            float g = __expf(bf2f(p[c]) - lse);
            if (c == lab) g -= 1.f;
            p[c] = f2bf(g * scale);
        }
        for (int c = V + threadIdx.x; c < (int)ld; c += 256) p[c] = 0;   // padded vocabulary columns carry no gradient
        __syncthreads();
    }
}

// The same with the row held in REGISTERS: one read of the logits and one write of the gradient, the algorithmic minimum (the kernel above
// reads a 265 KB row three times -- maximum, sum, gradient; the second and third come from L2, but the three dependent sweeps with two exp passes
// ran at 0.31 of the HBM rate).  512 threads per row, thread t owns the 16-byte chunks t, t + 512, ... (MAXC of them: 40 cover 163 840
// columns -- Llama's 132 608 --: 160 packed registers of the 256 a thread has at two waves per SIMD; with 1 024 threads x 20 chunks and a 128-register
// budget hipcc spilled 100-600 registers whatever the sweeps looked like.  Wider vocabularies (Gemma) stay on the kernel above).  Statistics as
// before: block maximum, then the sum of exp(v - max) in a fixed order (wave shuffles, then the eight wave sums in wave order): the same bits every launch.
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
__device__ __forceinline__ unsigned pack2_bf16(float lo, float hi)     // one v_cvt_pk_bf16_f32 (round to nearest even, NaN stays NaN)
{
    using bf2_t = __attribute__((ext_vector_type(2))) __bf16;
    bf2_t v;
    v[0] = (__bf16)lo; v[1] = (__bf16)hi;
    return __builtin_bit_cast(unsigned, v);
}
template <int MAXC>
__global__ __launch_bounds__(512) void ce_fwd_bwd_reg_kernel(unsigned short *logits, const long long *labels, float *row_loss,
                                                             float *sum_loss, const float *inv_count_ptr, size_t rows, int V, size_t ld)
{
    __shared__ float s_red[2][8];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const float inv_count = *inv_count_ptr;
    for (size_t r = blockIdx.x; r < rows; r += gridDim.x) {
        // per row, everything below is derived from values hipcc cannot see through: hoisted out of this loop, the forty chunk offsets, their
        // range tests and address pairs lived across the whole row and spilled (660 scalar, 170 vector registers)
        int tid = (int)threadIdx.x;
        asm volatile("" : "+v"(tid), "+s"(V), "+s"(ld));
        tid &= 511;                                           // (range known again: 32-bit offsets from the row's scalar base, no 64-bit address pair per chunk)
        unsigned short *p = logits + r * ld;
        const long long lab = labels[r];
        const bool valid = lab >= 0 && lab < V;
        u32x4 v[MAXC];                                        // packed pairs stay packed: element-wise edits of a bf16x8 made hipcc keep every 16-bit lane in a register of its own
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {                      // every load issued before the first use; chunks past the row's end re-read its first chunk
            const int c = (i * 512 + tid) * 8;
            v[i] = *reinterpret_cast<const u32x4 *>(p + (c < (int)ld ? c : 0));
        }
        // columns >= V (the padded vocabulary, chunks past the row) become -inf ONCE, in the packed registers: they then drop out of the maximum, add
        // exp(-inf) = 0 to the sum and get the gradient 0 with no per-element test in the sweeps (only the chunk that holds column V is partial)
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            if ((i + 1) * 4096 <= V) continue;                // (uniform) every lane's chunk of this round is whole
            const int lim = V - (i * 512 + tid) * 8;
            if (lim < 8) {
#pragma unroll
                for (int d = 0; d < 4; ++d)
                    v[i][d] = (2 * d < lim ? v[i][d] & 0xFFFFu : 0xFF80u) | (2 * d + 1 < lim ? v[i][d] & 0xFFFF0000u : 0xFF800000u);
            }
        }
        float m = -INFINITY;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
#pragma unroll
            for (int d = 0; d < 4; ++d) m = fmaxf(m, fmaxf(__uint_as_float(v[i][d] << 16), __uint_as_float(v[i][d] & 0xFFFF0000u)));
        }
        m = wave_max(m);
        if (lane == 0) s_red[0][wv] = m;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < 8; ++w) m = fmaxf(m, s_red[0][w]);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
#pragma unroll
            for (int d = 0; d < 4; ++d) { s += __expf(__uint_as_float(v[i][d] << 16) - m); s += __expf(__uint_as_float(v[i][d] & 0xFFFF0000u) - m); }
            if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // bounds how many expansions hipcc keeps in flight (registers)
        }
        s = wave_sum(s);
        if (lane == 0) s_red[1][wv] = s;
        __syncthreads();
        s = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) s += s_red[1][w];
        const float lse = m + logf(s);
        if (tid == 0) {
            const float l = valid ? lse - bf2f(p[lab]) : 0.f;        // p[lab] is read before any thread has stored: the stores follow the barrier below
            row_loss[r] = l;
            if (valid && sum_loss) atomicAdd(sum_loss, l * inv_count);
        }
        __syncthreads();
        const float scale = valid ? inv_count : 0.f;
        const int lab_c = valid ? (int)lab : -1;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = (i * 512 + tid) * 8;
            u32x4 o;
#pragma unroll
            for (int d = 0; d < 4; ++d)
                o[d] = pack2_bf16(__expf(__uint_as_float(v[i][d] << 16) - lse) * scale, __expf(__uint_as_float(v[i][d] & 0xFFFF0000u) - lse) * scale);
            if ((unsigned)(lab_c - i * 4096) < 4096u) {              // (uniform) the label column lies in this round of chunks: one lane owns it
                const int j = lab_c - c;
                if ((unsigned)j < 8u) {
                    const unsigned w = v[i][j >> 1];
                    const float x = (j & 1) ? __uint_as_float(w & 0xFFFF0000u) : __uint_as_float(w << 16);
                    const unsigned g = pack2_bf16((__expf(x - lse) - 1.f) * scale, 0.f) & 0xFFFFu;
#pragma unroll
                    for (int d = 0; d < 4; ++d)
                        if (d == (j >> 1)) o[d] = (j & 1) ? (o[d] & 0xFFFFu) | (g << 16) : (o[d] & 0xFFFF0000u) | g;
                }
            }
            if ((i + 1) * 4096 <= (int)ld) *reinterpret_cast<u32x4 *>(p + c) = o;      // (uniform) whole round inside the row
            else if (c < (int)ld) *reinterpret_cast<u32x4 *>(p + c) = o;
            if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// number of labels != -100 -> *inv_count = 1 / max(count, 1)   (mean reduction, loss_utils.py:24-29)
__global__ __launch_bounds__(256) void count_labels_kernel(const long long *labels, size_t n, int V, float *inv_count)
{
    __shared__ unsigned s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    unsigned c = 0;
    for (size_t i = threadIdx.x; i < n; i += blockDim.x) c += (labels[i] >= 0 && labels[i] < V) ? 1u : 0u;
    atomicAdd(&s_cnt, c);
    __syncthreads();
    if (threadIdx.x == 0) *inv_count = 1.0f / (float)(s_cnt ? s_cnt : 1u);
}

// ---- optimizer -------------------------------------------------------------------------------------
// sum of squares of a bf16 or fp32 gradient tensor, accumulated into *acc (fp32)
template <typename T>
__global__ __launch_bounds__(256) void sumsq_kernel(const T *g, size_t n, float *acc)
{
    __shared__ float s_red[4];
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float f;
        if constexpr (sizeof(T) == 2) f = bf2f(g[i]); else f = g[i];
        s += f * f;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(acc, s_red[0] + s_red[1] + s_red[2] + s_red[3]);
}

// The same over a LIST of bf16 tensors in one launch (one launch per parameter tensor is ~100 launches of mostly tiny
// work per step): block b sums chunk b = elements [chunk_off[b], chunk_off[b] + 2^20) of tensor chunk_tensor[b].
__global__ __launch_bounds__(256) void sumsq_multi_kernel(const unsigned short *const *ptrs, const unsigned long long *counts,
                                                          const int *chunk_tensor, const unsigned long long *chunk_off, float *acc,
                                                          float *partials)
{
    __shared__ float s_red[4];
    const int t = chunk_tensor[blockIdx.x];
    const unsigned long long off = chunk_off[blockIdx.x];
    const unsigned long long n = min((unsigned long long)(1u << 20), counts[t] - off);
    const unsigned short *g = ptrs[t] + off;
    float s = 0.f;
    const unsigned long long n8 = ((reinterpret_cast<uintptr_t>(g) & 15) == 0) ? (n & ~7ull) : 0ull;
    for (unsigned long long i = (unsigned long long)threadIdx.x * 8; i < n8; i += 256 * 8) {
        const bf16x8 v = *reinterpret_cast<const bf16x8 *>(g + i);
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float f = bf2f(v[j]); s += f * f; }
    }
    for (unsigned long long i = n8 + threadIdx.x; i < n; i += 256) { const float f = bf2f(g[i]); s += f * f; }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float b = s_red[0] + s_red[1] + s_red[2] + s_red[3];
        if (partials) partials[blockIdx.x] = b;              // summed in chunk order by ordered_sum_kernel: the same bits every step
        else atomicAdd(acc, b);
    }
}

// *acc += sum of v[0..n) in a fixed order (one workgroup: strided partial sums, then the usual tree): the gradient norm must not depend
// on the order in which workgroups happened to finish -- the clip factor multiplies every gradient.
__global__ __launch_bounds__(256) void ordered_sum_kernel(const float *v, int n, float *acc)
{
    __shared__ float s_red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += v[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) *acc += (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

// dw[c] += sum over b < n_blocks of partials[b][c], in block order (the weight gradients of the norms: per-workgroup partial rows)
// (16 columns x 16 row groups per workgroup: row group g adds rows g, g + 16, ... in order, the 16 group sums are added in group order -- a fixed
// association, and 32 loads per thread instead of 512 in a row: one thread per column took 0.12 ms per call, 4 ms of a step)
__global__ __launch_bounds__(256) void partial_rows_sum_kernel(const float *partials, int n_blocks, int H, long long ld, float *dw)
{
    __shared__ float s_p[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + tx;
    float s = 0.f;
    if (c < H)
        for (int b = ty; b < n_blocks; b += 16) s += partials[(size_t)b * ld + c];
    s_p[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < H) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += s_p[g][tx];
        dw[c] += t;
    }
}

// torch.optim.Adam with weight_decay as L2 (main.py:262-264), preceded by clip_grad_norm_(1.0)
// (train.py:26): clip = min(1, max_norm / (sqrt(*sumsq) + 1e-6)).  Moments fp32, params bf16.
template <typename G>
__global__ __launch_bounds__(256) void adam_kernel(unsigned short *p, const G *g, float *m, float *v, size_t n,
                                                   const float *sumsq, float max_norm, float lr, float b1, float b2,
                                                   float eps, float wd, float bc1, float bc2)
{
    const float norm = sqrtf(*sumsq);
    const float clip = fminf(1.0f, max_norm / (norm + 1e-6f));
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float gf;
        if constexpr (sizeof(G) == 2) gf = bf2f(g[i]); else gf = g[i];
        const float pf = bf2f(p[i]);
        gf = gf * clip + wd * pf;
        const float mm = b1 * m[i] + (1.f - b1) * gf;
        const float vv = b2 * v[i] + (1.f - b2) * gf * gf;
        m[i] = mm; v[i] = vv;
        const float denom = sqrtf(vv / bc2) + eps;
        p[i] = f2bf(pf - lr * (mm / bc1) / denom);
    }
}


// The same update over a LIST of bf16 parameter tensors in one launch (a full fine-tune step was ~100 launches of adam_kernel): block b takes chunk b = elements
// [chunk_off[b], chunk_off[b] + 2^20) of tensor chunk_tensor[b] -- sumsq_multi_kernel's chunk table.  Eight elements per lane and trip where the four pointers allow
// 16-byte accesses (the single-tensor kernel moves 2 and 4 bytes per lane and instruction); element by element the arithmetic of adam_kernel: the same bits.
__global__ __launch_bounds__(256) void adam_multi_kernel(unsigned short *const *ps, const unsigned short *const *gs, float *const *ms, float *const *vs,
                                                         const unsigned long long *counts, const int *chunk_tensor, const unsigned long long *chunk_off,
                                                         const float *sumsq, float max_norm, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2)
{
    using u4 = __attribute__((ext_vector_type(4))) unsigned;
    using f4 = __attribute__((ext_vector_type(4))) float;
    const float norm = sqrtf(*sumsq);
    const float clip = fminf(1.0f, max_norm / (norm + 1e-6f));
    const int t = chunk_tensor[blockIdx.x];
    const unsigned long long off = chunk_off[blockIdx.x];
    const unsigned long long n = min((unsigned long long)(1u << 20), counts[t] - off);
    unsigned short *p = ps[t] + off;
    const unsigned short *g = gs[t] + off;
    float *m = ms[t] + off, *v = vs[t] + off;
    auto one = [&](float pf, float gf, float &mm, float &vv) {
        gf = gf * clip + wd * pf;
        mm = b1 * mm + (1.f - b1) * gf;
        vv = b2 * vv + (1.f - b2) * gf * gf;
        const float denom = sqrtf(vv / bc2) + eps;
        return f2bf(pf - lr * (mm / bc1) / denom);
    };
    const bool wide = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) == 0;
    const unsigned long long n8 = wide ? (n & ~7ull) : 0ull;
    for (unsigned long long i = (unsigned long long)threadIdx.x * 8; i < n8; i += 256 * 8) {
        const u4 pw = *reinterpret_cast<const u4 *>(p + i), gw = *reinterpret_cast<const u4 *>(g + i);
        float mm[8], vv[8];
        *reinterpret_cast<f4 *>(mm) = *reinterpret_cast<const f4 *>(m + i); *reinterpret_cast<f4 *>(mm + 4) = *reinterpret_cast<const f4 *>(m + i + 4);
        *reinterpret_cast<f4 *>(vv) = *reinterpret_cast<const f4 *>(v + i); *reinterpret_cast<f4 *>(vv + 4) = *reinterpret_cast<const f4 *>(v + i + 4);
        u4 out;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const unsigned short lo = one(bf2f((unsigned short)(pw[w] & 0xFFFFu)), bf2f((unsigned short)(gw[w] & 0xFFFFu)), mm[2 * w], vv[2 * w]);
            const unsigned short hi = one(bf2f((unsigned short)(pw[w] >> 16)), bf2f((unsigned short)(gw[w] >> 16)), mm[2 * w + 1], vv[2 * w + 1]);
            out[w] = (unsigned)lo | ((unsigned)hi << 16);
        }
        *reinterpret_cast<f4 *>(m + i) = *reinterpret_cast<const f4 *>(mm); *reinterpret_cast<f4 *>(m + i + 4) = *reinterpret_cast<const f4 *>(mm + 4);
        *reinterpret_cast<f4 *>(v + i) = *reinterpret_cast<const f4 *>(vv); *reinterpret_cast<f4 *>(v + i + 4) = *reinterpret_cast<const f4 *>(vv + 4);
        *reinterpret_cast<u4 *>(p + i) = out;
    }
    for (unsigned long long i = n8 + threadIdx.x; i < n; i += 256) {
        float mm = m[i], vv = v[i];
        p[i] = one(bf2f(p[i]), bf2f(g[i]), mm, vv);
        m[i] = mm; v[i] = vv;
    }
}

// out[c][r] = in[r][c]  (bf16), 64x64 tiles through LDS
__global__ __launch_bounds__(256) void transpose_kernel(const unsigned short *in, unsigned short *out, int R, int Cc)
{
    __shared__ unsigned short tile[64][66];
    const int tiles_c = (Cc + 63) / 64, tiles_r = (R + 63) / 64;
    for (int t = blockIdx.x; t < tiles_c * tiles_r; t += gridDim.x) {
        const int tr = t / tiles_c, tc = t % tiles_c;
        for (int k = threadIdx.x; k < 64 * 64; k += 256) {
            const int r = k / 64, c = k % 64;
            const int gr = tr * 64 + r, gc = tc * 64 + c;
            tile[r][c] = (gr < R && gc < Cc) ? in[(size_t)gr * Cc + gc] : 0;
        }
        __syncthreads();
        for (int k = threadIdx.x; k < 64 * 64; k += 256) {
            const int c = k / 64, r = k % 64;
            const int gr = tr * 64 + r, gc = tc * 64 + c;
            if (gr < R && gc < Cc) out[(size_t)gc * R + gr] = tile[r][c];
        }
        __syncthreads();
    }
}

// many small matrices in ONE launch (the LoRA adapters' transposed copies: 2 per site, 128 of a few KB each per step -- launch-bound as single
// calls): block b transposes 64x64 tile b - tile_off[t] of matrix t, t found by bisection in the tile prefix sums.
__global__ __launch_bounds__(256) void transpose_multi_kernel(const unsigned short *const *src, unsigned short *const *dst, const int *rows,
                                                              const int *cols, const int *tile_off, int n)
{
    __shared__ unsigned short tile[64][66];
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tile_off[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const int t = lo, R = rows[t], Cc = cols[t];
    const unsigned short *in = src[t];
    unsigned short *out = dst[t];
    const int local = (int)blockIdx.x - tile_off[t], tiles_c = (Cc + 63) / 64;
    const int tr = local / tiles_c, tc = local % tiles_c;
    for (int k = threadIdx.x; k < 64 * 64; k += 256) {
        const int r = k / 64, c = k % 64;
        const int gr = tr * 64 + r, gc = tc * 64 + c;
        tile[r][c] = (gr < R && gc < Cc) ? in[(size_t)gr * Cc + gc] : 0;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 64 * 64; k += 256) {
        const int c = k / 64, r = k % 64;
        const int gr = tr * 64 + r, gc = tc * 64 + c;
        if (gr < R && gc < Cc) out[(size_t)gc * R + gr] = tile[r][c];
    }
}

// batched, strided variant: matrix z = (zo, zi) of R x Cc elements, input rows ld_in apart, output rows ld_out apart
__global__ __launch_bounds__(256) void transpose_strided_kernel(const unsigned short *in, unsigned short *out, int R, int Cc,
                                                                long long ld_in, long long ld_out, int inner,
                                                                long long outer_in, long long inner_in, long long outer_out,
                                                                long long inner_out)
{
    __shared__ unsigned short tile[64][66];
    const int zo = blockIdx.y / inner, zi = blockIdx.y % inner;
    const unsigned short *src = in + zo * outer_in + zi * inner_in;
    unsigned short *dst = out + zo * outer_out + zi * inner_out;
    const int tiles_c = (Cc + 63) / 64, tiles_r = (R + 63) / 64;
    for (int t = blockIdx.x; t < tiles_c * tiles_r; t += gridDim.x) {
        const int tr = t / tiles_c, tc = t % tiles_c;
        for (int k = threadIdx.x; k < 64 * 64; k += 256) {
            const int r = k / 64, c = k % 64;
            const int gr = tr * 64 + r, gc = tc * 64 + c;
            tile[r][c] = (gr < R && gc < Cc) ? src[(long long)gr * ld_in + gc] : 0;
        }
        __syncthreads();
        for (int k = threadIdx.x; k < 64 * 64; k += 256) {
            const int c = k / 64, r = k % 64;
            const int gr = tr * 64 + r, gc = tc * 64 + c;
            if (gr < R && gc < Cc) dst[(long long)gc * ld_out + gr] = tile[r][c];
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float *in, unsigned short *out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = f2bf(in[i]);
}

// out[i] = bf16(sum_s slabs[s * stride + i]) (+ out[i] when ACC): the K-slices of a split weight-gradient product, summed in slice
// order.  Four elements per thread step: 16-byte loads per slab, one 8-byte store.
template <bool ACC>
__global__ __launch_bounds__(256) void sum_slabs_kernel(const float *slabs, long long stride, int n_slabs, unsigned short *out, size_t n4)
{
    using f4 = __attribute__((ext_vector_type(4))) float;
    using us4 = __attribute__((ext_vector_type(4))) unsigned short;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        f4 v = *reinterpret_cast<const f4 *>(slabs + 4 * i);
        for (int s = 1; s < n_slabs; ++s) {
            const f4 w = *reinterpret_cast<const f4 *>(slabs + (long long)s * stride + 4 * i);
            v += w;
        }
        us4 o;
        if (ACC) {
            o = *reinterpret_cast<const us4 *>(out + 4 * i);
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] += bf2f(o[t]);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) o[t] = f2bf(v[t]);
        *reinterpret_cast<us4 *>(out + 4 * i) = o;
    }
}

// ---- attention softmax (scores materialised) -------------------------------------------------------
// One wave per row i of one (batch, head): keys j <= i with mask[b, j] != 0.
__global__ __launch_bounds__(256) void softmax_causal_fwd_kernel(unsigned short *scores, const float *mask, size_t rows_total,
                                                                 int n_heads, int S, float scale)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t row = wave; row < rows_total; row += n_waves) {
        const size_t bh = row / S;
        const int i = (int)(row % S);
        const float *mk = mask + (bh / n_heads) * S;
        unsigned short *p = scores + row * S;
        const int lim = i + 1;                       // keys 0..i
        float m = -INFINITY;
        for (int c = lane * 8; c < lim; c += 64 * 8) {
            const bf16x8 v = *reinterpret_cast<const bf16x8 *>(p + c);
#pragma unroll
            for (int j = 0; j < 8; ++j) if (c + j < lim && mk[c + j] != 0.f) m = fmaxf(m, bf2f(v[j]) * scale);
        }
        m = wave_max(m);
        float sum = 0.f;
        for (int c = lane * 8; c < lim; c += 64 * 8) {
            const bf16x8 v = *reinterpret_cast<const bf16x8 *>(p + c);
#pragma unroll
            for (int j = 0; j < 8; ++j) if (c + j < lim && mk[c + j] != 0.f) sum += __expf(bf2f(v[j]) * scale - m);
        }
        sum = wave_sum(sum);
        const float inv = sum > 0.f ? 1.f / sum : 0.f;
        for (int c = lane * 8; c < S; c += 64 * 8) {
            bf16x8 v = *reinterpret_cast<const bf16x8 *>(p + c);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool vis = (c + j < lim) && mk[c + j] != 0.f;
                v[j] = vis ? f2bf(__expf(bf2f(v[j]) * scale - m) * inv) : (unsigned short)0;
            }
            *reinterpret_cast<bf16x8 *>(p + c) = v;
        }
    }
}

// dS = scale * P * (dP - sum_j P*dP), in place on dp.  One wave per row.
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const unsigned short *P, unsigned short *dP, size_t rows_total, int S,
                                                          float scale)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t row = wave; row < rows_total; row += n_waves) {
        const int i = (int)(row % S);
        const int lim8 = ((i + 1) + 7) & ~7;         // P is zero beyond key i
        const unsigned short *p = P + row * S;
        unsigned short *d = dP + row * S;
        float dot = 0.f;
        for (int c = lane * 8; c < lim8; c += 64 * 8) {
            const bf16x8 a = *reinterpret_cast<const bf16x8 *>(p + c), g = *reinterpret_cast<const bf16x8 *>(d + c);
#pragma unroll
            for (int j = 0; j < 8; ++j) dot += bf2f(a[j]) * bf2f(g[j]);
        }
        dot = wave_sum(dot);
        for (int c = lane * 8; c < S; c += 64 * 8) {
            const bf16x8 a = *reinterpret_cast<const bf16x8 *>(p + c);
            bf16x8 g = *reinterpret_cast<const bf16x8 *>(d + c);
#pragma unroll
            for (int j = 0; j < 8; ++j) g[j] = f2bf(scale * bf2f(a[j]) * (bf2f(g[j]) - dot));
            *reinterpret_cast<bf16x8 *>(d + c) = g;
        }
    }
}

}  // namespace

#define ECGB_CHECK_LAUNCH(name) return ok_or(hipGetLastError(), name)

int g_ce_in_registers = 1;   // ecgb_ce_fwd_bwd: rows that fit (ld <= 163 840) are held in registers (one read, one write); 0 = the three-sweep kernel (A/B, tests)
extern "C" int ecgb_set_ce_in_registers(int on) { g_ce_in_registers = on ? 1 : 0; return ECGB_OK; }

extern "C" int ecgb_argmax_bf16(const void *x_dev, long long ld, int rows, int n, int64_t *out_dev, void *stream)
{
    if (!x_dev || !out_dev || rows <= 0 || n <= 0 || ld < n) { ecgb::set_error("ecgb_argmax_bf16: bad argument"); return ECGB_ERR_INVALID; }
    // few rows of many columns (generate: one row of the vocabulary per sequence): a row over up to 32 workgroups.  Every launch takes the next block of 16 slots (256
    // blocks: launches in flight on different streams do not meet unless 256 of them are; a captured launch keeps its block on every replay)
    static std::atomic<unsigned> next_block{0};
    const unsigned parts = rows <= 16 ? (unsigned)std::max(1, std::min(32, n / 8192)) : 1u;
    const int slot0 = parts > 1 ? (int)((next_block.fetch_add(1u) % (unsigned)(kArgmaxSlots / 16)) * 16u) : 0;
    hipLaunchKernelGGL(argmax_rows_kernel, dim3((unsigned)rows, parts), dim3(1024), 0, (hipStream_t)stream, (const unsigned short *)x_dev, ld, n, (long long *)out_dev, slot0);
    ECGB_CHECK_LAUNCH("argmax_rows");
}

namespace {
__global__ void rope_table_kernel(const long long *pos, int n, const float *inv_freq, int half, float *cos_out, float *sin_out)
{
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n * half; k += gridDim.x * blockDim.x) {
        const float fr = (float)pos[k / half] * inv_freq[k % half];
        cos_out[k] = cosf(fr);
        sin_out[k] = sinf(fr);
    }
}
}  // namespace

extern "C" int ecgb_rope_table(const int64_t *pos_dev, int n, const float *inv_freq_dev, int half, float *cos_out_dev, float *sin_out_dev, void *stream)
{
    if (!pos_dev || !inv_freq_dev || !cos_out_dev || !sin_out_dev || n <= 0 || half <= 0) { ecgb::set_error("ecgb_rope_table: bad argument"); return ECGB_ERR_INVALID; }
    hipLaunchKernelGGL(rope_table_kernel, dim3((unsigned)std::min(64, (n * half + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const long long *)pos_dev, n, inv_freq_dev,
                       half, cos_out_dev, sin_out_dev);
    ECGB_CHECK_LAUNCH("rope_table");
}

namespace {
__global__ void decode_advance_kernel(const long long *next, int batch, long long *tok, long long *pos, long long *col, int *n_dev, long long *out, long long out_ld,
                                      float *mask, long long mask_ld, long long *unfinished, long long pad_id, const long long *eos, int n_eos, int *epoch)
{
    for (int b = threadIdx.x; b < batch; b += blockDim.x) {
        long long t = next[b];
        if (n_eos > 0) {
            const long long u = unfinished[b];
            t = t * u + pad_id * (1 - u);                       // (the reference's arithmetic: generation/utils.py:3210)
            bool is_eos = false;
            for (int e = 0; e < n_eos; ++e) is_eos = is_eos || t == eos[e];
            unfinished[b] = u * (is_eos ? 0 : 1);
        }
        const long long c = col[b];
        out[(long long)b * out_ld + c] = t;
        mask[(long long)b * mask_ld + c] = 1.0f;
        tok[b] = t;
        pos[b] += 1;
        col[b] = c + 1;
    }
    if (threadIdx.x == 0) { *n_dev += 1; if (epoch) *epoch += 1; }
}
}  // namespace

extern "C" int ecgb_decode_advance(const int64_t *next_dev, int batch, int64_t *tok_dev, int64_t *pos_dev, int64_t *col_dev, int *n_dev, int64_t *out_dev, long long out_ld,
                                   float *mask_dev, long long mask_ld, int64_t *unfinished_dev, long long pad_id, const int64_t *eos_dev, int n_eos, void *stream)
{
    if (!next_dev || !tok_dev || !pos_dev || !col_dev || !n_dev || !out_dev || !mask_dev || batch <= 0 || n_eos < 0 || (n_eos > 0 && (!eos_dev || !unfinished_dev))) {
        ecgb::set_error("ecgb_decode_advance: bad argument");
        return ECGB_ERR_INVALID;
    }
    hipLaunchKernelGGL(decode_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const long long *)next_dev, batch, (long long *)tok_dev, (long long *)pos_dev,
                       (long long *)col_dev, n_dev, (long long *)out_dev, out_ld, mask_dev, mask_ld, (long long *)unfinished_dev, pad_id, (const long long *)eos_dev, n_eos, (int *)nullptr);
    ECGB_CHECK_LAUNCH("decode_advance");
}

extern "C" int ecgb_decode_advance_e(const int64_t *next_dev, int batch, int64_t *tok_dev, int64_t *pos_dev, int64_t *col_dev, int *n_dev, int64_t *out_dev, long long out_ld,
                                     float *mask_dev, long long mask_ld, int64_t *unfinished_dev, long long pad_id, const int64_t *eos_dev, int n_eos, int *epoch_dev, void *stream)
{
    if (!next_dev || !tok_dev || !pos_dev || !col_dev || !n_dev || !out_dev || !mask_dev || !epoch_dev || batch <= 0 || n_eos < 0 || (n_eos > 0 && (!eos_dev || !unfinished_dev))) {
        ecgb::set_error("ecgb_decode_advance_e: bad argument");
        return ECGB_ERR_INVALID;
    }
    hipLaunchKernelGGL(decode_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const long long *)next_dev, batch, (long long *)tok_dev, (long long *)pos_dev,
                       (long long *)col_dev, n_dev, (long long *)out_dev, out_ld, mask_dev, mask_ld, (long long *)unfinished_dev, pad_id, (const long long *)eos_dev, n_eos, epoch_dev);
    ECGB_CHECK_LAUNCH("decode_advance_e");
}

extern "C" int ecgb_embed_fwd(const int64_t *ids_dev, const void *table_dev, void *out_dev, size_t tokens, int hidden,
                              float scale, void *stream)
{
    if (hidden % 8) { ecgb::set_error("ecgb_embed_fwd: hidden must be a multiple of 8"); return ECGB_ERR_INVALID; }
    hipLaunchKernelGGL(embed_fwd_kernel, dim3(grid_for(tokens * (hidden / 8), 256)), dim3(256), 0, (hipStream_t)stream,
                       (const long long *)ids_dev, (const unsigned short *)table_dev, (unsigned short *)out_dev, tokens, hidden, scale);
    ECGB_CHECK_LAUNCH("embed_fwd");
}

extern "C" int ecgb_embed_bwd(const int64_t *ids_dev, const void *dout_dev, float *grad_table_dev, size_t tokens, int hidden,
                              float scale, void *stream)
{
    if (hidden % 8) { ecgb::set_error("ecgb_embed_bwd: hidden must be a multiple of 8"); return ECGB_ERR_INVALID; }
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(grid_for(tokens * (hidden / 8), 256)), dim3(256), 0, (hipStream_t)stream,
                       (const long long *)ids_dev, (const unsigned short *)dout_dev, grad_table_dev, tokens, hidden, scale);
    ECGB_CHECK_LAUNCH("embed_bwd");
}

namespace {
// The same for H = NC * 512 (2048: both model families): the row stays in registers between the sum of squares and the scaling -- the kernel above reads the
// residual sum back from memory right after storing it (a fifth more load traffic and a store -> load round trip inside every row) -- and all of the row's loads
// are in flight at once.  Same arithmetic, same bits.
template <bool GEMMA, int NC>
__global__ __launch_bounds__(256) void rmsnorm_fwd_rows_kernel(const unsigned short *a, const unsigned short *b, const unsigned short *w, unsigned short *y,
                                                               unsigned short *sum_out, float *rstd, size_t rows, float eps)
{
    constexpr int H = NC * 512;
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    bf16x8 g[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) g[k] = *reinterpret_cast<const bf16x8 *>(w + k * 512 + lane * 8);
    for (size_t r = wave; r < rows; r += n_waves) {
        bf16x8 v[NC], u[NC];
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            v[k] = *reinterpret_cast<const bf16x8 *>(a + r * H + k * 512 + lane * 8);
            if (b) u[k] = *reinterpret_cast<const bf16x8 *>(b + r * H + k * 512 + lane * 8);
        }
        float ss = 0.f;
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            if (b) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[k][j] = f2bf(bf2f(v[k][j]) + bf2f(u[k][j]));
                *reinterpret_cast<bf16x8 *>(sum_out + r * H + k * 512 + lane * 8) = v[k];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float f = bf2f(v[k][j]); ss += f * f; }
        }
        ss = wave_sum(ss);
        const float rs = rsqrtf(ss / (float)H + eps);
        if (lane == 0 && rstd) rstd[r] = rs;
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (GEMMA) o[j] = f2bf(bf2f(v[k][j]) * rs * (1.0f + bf2f(g[k][j])));
                else o[j] = f2bf(bf2f(f2bf(bf2f(v[k][j]) * rs)) * bf2f(g[k][j]));
            }
            *reinterpret_cast<bf16x8 *>(y + r * H + k * 512 + lane * 8) = o;
        }
    }
}
}  // namespace

extern "C" int ecgb_embed_bwd_sorted(const int64_t *ids_sorted_dev, const int64_t *order_dev, const void *dout_dev, void *grad_table_dev,
                                    size_t tokens, int hidden, float scale, int64_t skip_id, void *stream)
{
    if (!ids_sorted_dev || !order_dev || !dout_dev || !grad_table_dev || hidden % 8) { ecgb::set_error("ecgb_embed_bwd_sorted: bad argument (hidden % 8)"); return ECGB_ERR_INVALID; }
    if (tokens == 0) return ECGB_OK;
    hipLaunchKernelGGL(embed_bwd_sorted_kernel, dim3((unsigned)tokens), dim3(256), 0, (hipStream_t)stream, (const long long *)ids_sorted_dev,
                       (const long long *)order_dev, (const unsigned short *)dout_dev, (unsigned short *)grad_table_dev, tokens, hidden, scale,
                       (long long)skip_id);
    ECGB_CHECK_LAUNCH("embed_bwd_sorted");
}

namespace { int g_rms_fwd_rows = 1; }
// A/B switch (tests, tuning): 1 (default) the register-resident forward for hidden 2048, 0 the generic kernel.  Same bits.
extern "C" int ecgb_set_rmsnorm_fwd_rows(int on) { g_rms_fwd_rows = on ? 1 : 0; return ECGB_OK; }

extern "C" int ecgb_rmsnorm_fwd(const void *x_dev, const void *residual_dev, const void *w_dev, void *y_dev, void *sum_out_dev,
                                float *rstd_dev, size_t rows, int hidden, float eps, int gemma, void *stream)
{
    if (hidden % 8 || (residual_dev && !sum_out_dev)) { ecgb::set_error("ecgb_rmsnorm_fwd: bad arguments"); return ECGB_ERR_INVALID; }
    const dim3 grid(hidden == 2048 && g_rms_fwd_rows ? stream_grid(rows, 4) : grid_for(rows, 4));
    if (hidden == 2048 && g_rms_fwd_rows) {
        if (gemma)
            hipLaunchKernelGGL((rmsnorm_fwd_rows_kernel<true, 4>), grid, dim3(256), 0, (hipStream_t)stream, (const unsigned short *)x_dev, (const unsigned short *)residual_dev,
                               (const unsigned short *)w_dev, (unsigned short *)y_dev, (unsigned short *)sum_out_dev, rstd_dev, rows, eps);
        else
            hipLaunchKernelGGL((rmsnorm_fwd_rows_kernel<false, 4>), grid, dim3(256), 0, (hipStream_t)stream, (const unsigned short *)x_dev, (const unsigned short *)residual_dev,
                               (const unsigned short *)w_dev, (unsigned short *)y_dev, (unsigned short *)sum_out_dev, rstd_dev, rows, eps);
        ECGB_CHECK_LAUNCH("rmsnorm_fwd");
    }
    if (gemma)
        hipLaunchKernelGGL(rmsnorm_fwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned short *)x_dev,
                           (const unsigned short *)residual_dev, (const unsigned short *)w_dev, (unsigned short *)y_dev,
                           (unsigned short *)sum_out_dev, rstd_dev, rows, hidden, eps);
    else
        hipLaunchKernelGGL(rmsnorm_fwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned short *)x_dev,
                           (const unsigned short *)residual_dev, (const unsigned short *)w_dev, (unsigned short *)y_dev,
                           (unsigned short *)sum_out_dev, rstd_dev, rows, hidden, eps);
    ECGB_CHECK_LAUNCH("rmsnorm_fwd");
}

extern "C" int ecgb_rmsnorm_lora_fwd(const void *x_dev, const void *residual_dev, const void *w_dev, void *y_dev, void *sum_out_dev, float *rstd_dev, size_t rows,
                                     int hidden, float eps, int gemma, const void *lora_a_dev, long long lda, int n_a, float lora_scale, void *t_dev, long long ldt,
                                     void *stream)
{
    if (hidden % 512 || (residual_dev && !sum_out_dev) || !lora_a_dev || !t_dev || n_a <= 0 || lda % 8 || rows == 0 || rows > 65535 || ((uintptr_t)lora_a_dev & 15)) {
        ecgb::set_error("ecgb_rmsnorm_lora_fwd: bad arguments (hidden % 512, 16-byte aligned A)");
        return ECGB_ERR_INVALID;
    }
    const size_t lds = (size_t)hidden * 2;
    if (gemma)
        hipLaunchKernelGGL(rmsnorm_lora_fwd_kernel<true>, dim3((unsigned)rows, (unsigned)((n_a + 3) / 4)), dim3(256), lds, (hipStream_t)stream, (const unsigned short *)x_dev,
                           (const unsigned short *)residual_dev, (const unsigned short *)w_dev, (unsigned short *)y_dev, (unsigned short *)sum_out_dev, rstd_dev, hidden, eps,
                           (const unsigned short *)lora_a_dev, lda, n_a, lora_scale, (unsigned short *)t_dev, ldt);
    else
        hipLaunchKernelGGL(rmsnorm_lora_fwd_kernel<false>, dim3((unsigned)rows, (unsigned)((n_a + 3) / 4)), dim3(256), lds, (hipStream_t)stream, (const unsigned short *)x_dev,
                           (const unsigned short *)residual_dev, (const unsigned short *)w_dev, (unsigned short *)y_dev, (unsigned short *)sum_out_dev, rstd_dev, hidden, eps,
                           (const unsigned short *)lora_a_dev, lda, n_a, lora_scale, (unsigned short *)t_dev, ldt);
    ECGB_CHECK_LAUNCH("rmsnorm_lora_fwd");
}

namespace { int g_rms_bwd_rows_per_wg = 64; }
// (tuning) rows per workgroup of ecgb_rmsnorm_bwd at hidden 2048 / 4096; changes the scratch size ecgb_rmsnorm_bwd_scratch_floats reports and the order of the dw sum
extern "C" int ecgb_set_rmsnorm_bwd_rows_per_wg(int n)
{
    if (n < 4 || n > 4096) { ecgb::set_error("ecgb_set_rmsnorm_bwd_rows_per_wg: 4..4096"); return ECGB_ERR_INVALID; }
    g_rms_bwd_rows_per_wg = n;
    return ECGB_OK;
}

namespace {
int g_rms_bwd_grid_cap = 2048;
size_t rms_bwd_grid(size_t rows) { return std::min<size_t>(std::max<size_t>(1, rows / (size_t)g_rms_bwd_rows_per_wg), (size_t)g_rms_bwd_grid_cap); }
}
extern "C" int ecgb_set_rmsnorm_bwd_grid_cap(int n)
{
    if (n < 1 || n > 65536) { ecgb::set_error("ecgb_set_rmsnorm_bwd_grid_cap: 1..65536"); return ECGB_ERR_INVALID; }
    g_rms_bwd_grid_cap = n;
    return ECGB_OK;
}

extern "C" size_t ecgb_rmsnorm_bwd_scratch_floats(size_t rows, int hidden)
{
    if (hidden != 2048 && hidden != 4096) return std::min<size_t>(std::max<size_t>(1, rows), 4096) * (size_t)hidden;   // one-wave workgroups
    return rms_bwd_grid(rows) * (size_t)hidden;
}

// dst[c] += sum over b < n_rows of partials[b * ld + c], c < n, in row order (per-workgroup partial sums of a weight / bias gradient)
extern "C" int ecgb_partial_rows_sum_f32(const float *partials_dev, int n_rows, int n, long long ld, float *dst_dev, void *stream)
{
    if (!partials_dev || !dst_dev || n_rows < 0 || n <= 0) { ecgb::set_error("ecgb_partial_rows_sum_f32: bad argument"); return ECGB_ERR_INVALID; }
    hipLaunchKernelGGL(partial_rows_sum_kernel, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, (hipStream_t)stream, partials_dev, n_rows, n, ld, dst_dev);
    ECGB_CHECK_LAUNCH("partial_rows_sum");
}

extern "C" int ecgb_rmsnorm_bwd(const void *x_dev, const void *w_dev, const float *rstd_dev, const void *dy_dev,
                                const void *dres_dev, void *dx_dev, float *dw_dev, size_t rows, int hidden, int gemma, float *scratch_dev,
                                void *stream)
{
    if (hidden % 8) { ecgb::set_error("ecgb_rmsnorm_bwd: hidden must be a multiple of 8"); return ECGB_ERR_INVALID; }
    if (hidden == 2048 || hidden == 4096) {   // rows per wave ~16: enough to amortise the end-of-kernel reduction, enough waves to fill the chip
        // frozen norm weights (dw NULL: the LoRA step) leave no per-workgroup partial row behind: one row a wave (round 6: 100 -> 94 us at [32 768, 2 048])
        const dim3 g2(dw_dev ? (unsigned)rms_bwd_grid(rows) : (unsigned)std::min<size_t>(std::max<size_t>(1, rows / 4), 8192));
#define ECGB_RMS_BWD_ROWS(G_, NC_) hipLaunchKernelGGL((rmsnorm_bwd_rows_kernel<G_, NC_>), g2, dim3(256), 0, (hipStream_t)stream, (const unsigned short *)x_dev, \
        (const unsigned short *)w_dev, rstd_dev, (const unsigned short *)dy_dev, (const unsigned short *)dres_dev, (unsigned short *)dx_dev, dw_dev, rows, scratch_dev)
        if (hidden == 2048) { if (gemma) ECGB_RMS_BWD_ROWS(true, 4); else ECGB_RMS_BWD_ROWS(false, 4); }
        else { if (gemma) ECGB_RMS_BWD_ROWS(true, 8); else ECGB_RMS_BWD_ROWS(false, 8); }
#undef ECGB_RMS_BWD_ROWS
        if (scratch_dev && dw_dev)      // per-block partial rows -> dw, in block order (scratch: ecgb_rmsnorm_bwd_scratch_floats; null: atomics)
            hipLaunchKernelGGL(partial_rows_sum_kernel, dim3((unsigned)((hidden + 15) / 16)), dim3(256), 0, (hipStream_t)stream, scratch_dev, (int)g2.x, hidden, (long long)hidden, dw_dev);
        ECGB_CHECK_LAUNCH("rmsnorm_bwd");
    }
    // scratch: one wave per workgroup (its LDS adds are then in program order) and per-workgroup rows added in order; without: four waves and atomics
    const dim3 grid(scratch_dev ? (unsigned)std::min<size_t>(std::max<size_t>(1, rows), 4096) : (unsigned)std::min<size_t>(std::max<size_t>(1, rows / 4), 1024));
    const dim3 block(scratch_dev ? 64 : 256);
    const size_t lds = (size_t)hidden * 4;
    if (gemma)
        hipLaunchKernelGGL(rmsnorm_bwd_kernel<true>, grid, block, lds, (hipStream_t)stream, (const unsigned short *)x_dev,
                           (const unsigned short *)w_dev, rstd_dev, (const unsigned short *)dy_dev, (const unsigned short *)dres_dev,
                           (unsigned short *)dx_dev, dw_dev, rows, hidden, scratch_dev);
    else
        hipLaunchKernelGGL(rmsnorm_bwd_kernel<false>, grid, block, lds, (hipStream_t)stream, (const unsigned short *)x_dev,
                           (const unsigned short *)w_dev, rstd_dev, (const unsigned short *)dy_dev, (const unsigned short *)dres_dev,
                           (unsigned short *)dx_dev, dw_dev, rows, hidden, scratch_dev);
    if (scratch_dev && dw_dev)
        hipLaunchKernelGGL(partial_rows_sum_kernel, dim3((unsigned)((hidden + 15) / 16)), dim3(256), 0, (hipStream_t)stream, scratch_dev, (int)grid.x, hidden, (long long)hidden, dw_dev);
    ECGB_CHECK_LAUNCH("rmsnorm_bwd");
}

extern "C" int ecgb_rope(void *x_dev, const float *cos_dev, const float *sin_dev, size_t tokens, int n_heads, int head_dim,
                         size_t row_stride, int inverse, void *stream)
{
    if (head_dim % 16) { ecgb::set_error("ecgb_rope: head_dim must be a multiple of 16"); return ECGB_ERR_INVALID; }
    const dim3 grid(stream_grid(tokens * n_heads * (head_dim / 16), 256));
    if (inverse)
        hipLaunchKernelGGL(rope_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, (unsigned short *)x_dev, cos_dev, sin_dev,
                           tokens, n_heads, head_dim, row_stride);
    else
        hipLaunchKernelGGL(rope_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, (unsigned short *)x_dev, cos_dev, sin_dev,
                           tokens, n_heads, head_dim, row_stride);
    ECGB_CHECK_LAUNCH("rope");
}

extern "C" int ecgb_rope_append(void *qkv_dev, const float *cos_dev, const float *sin_dev, int batch, int n_q_heads, int n_kv_heads, int head_dim, size_t row_stride,
                                void *cache_dev, long long capacity, int kv_len, const int *kv_len_dev, void *stream)
{
    if (!qkv_dev || !cos_dev || !sin_dev || !cache_dev || batch <= 0 || n_q_heads <= 0 || n_kv_heads <= 0 || head_dim % 16 || capacity <= 0 ||
        (!kv_len_dev && (kv_len <= 0 || kv_len > capacity)) || row_stride % 8) {
        ecgb::set_error("ecgb_rope_append: bad argument");
        return ECGB_ERR_INVALID;
    }
    const size_t items = (size_t)batch * ((size_t)(n_q_heads + n_kv_heads) * (head_dim / 16) + (size_t)n_kv_heads * head_dim / 8);
    hipLaunchKernelGGL(rope_append_kernel, dim3(grid_for(items, 256)), dim3(256), 0, (hipStream_t)stream, (unsigned short *)qkv_dev, cos_dev, sin_dev, batch, n_q_heads,
                       n_kv_heads, head_dim, row_stride, (unsigned short *)cache_dev, capacity, kv_len, kv_len_dev);
    ECGB_CHECK_LAUNCH("rope_append");
}

extern "C" int ecgb_glu_fwd(const void *gate_up_dev, void *h_dev, size_t tokens, int inter, int gelu_tanh, void *stream)
{
    if (inter % 8) { ecgb::set_error("ecgb_glu_fwd: intermediate size must be a multiple of 8"); return ECGB_ERR_INVALID; }
    const dim3 grid(stream_grid(tokens * (inter / 8), 256));
    if (gelu_tanh) hipLaunchKernelGGL(glu_fwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned short *)gate_up_dev, (unsigned short *)h_dev, tokens, inter);
    else hipLaunchKernelGGL(glu_fwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned short *)gate_up_dev, (unsigned short *)h_dev, tokens, inter);
    ECGB_CHECK_LAUNCH("glu_fwd");
}

extern "C" int ecgb_glu_bwd(const void *gate_up_dev, const void *dh_dev, void *dgate_up_dev, size_t tokens, int inter,
                            int gelu_tanh, void *stream)
{
    if (inter % 8) { ecgb::set_error("ecgb_glu_bwd: intermediate size must be a multiple of 8"); return ECGB_ERR_INVALID; }
    const dim3 grid(stream_grid(tokens * (inter / 8), 256));
    if (gelu_tanh) hipLaunchKernelGGL(glu_bwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned short *)gate_up_dev, (const unsigned short *)dh_dev, (unsigned short *)dgate_up_dev, tokens, inter);
    else hipLaunchKernelGGL(glu_bwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned short *)gate_up_dev, (const unsigned short *)dh_dev, (unsigned short *)dgate_up_dev, tokens, inter);
    ECGB_CHECK_LAUNCH("glu_bwd");
}

extern "C" int ecgb_set_stream_grid_cap(int n)
{
    if (n < 1) { ecgb::set_error("ecgb_set_stream_grid_cap: n >= 1"); return ECGB_ERR_INVALID; }
    g_stream_grid_cap = (size_t)n;
    return ECGB_OK;
}

extern "C" int ecgb_add_bf16(const void *a_dev, const void *b_dev, void *out_dev, size_t n, void *stream)
{
    if (n % 8) { ecgb::set_error("ecgb_add_bf16: n must be a multiple of 8"); return ECGB_ERR_INVALID; }
    hipLaunchKernelGGL(add_kernel, dim3(grid_for(n / 8, 256)), dim3(256), 0, (hipStream_t)stream, (const unsigned short *)a_dev,
                       (const unsigned short *)b_dev, (unsigned short *)out_dev, n / 8);
    ECGB_CHECK_LAUNCH("add_bf16");
}

extern "C" int ecgb_count_labels(const int64_t *labels_dev, size_t n, int vocab, float *inv_count_dev, void *stream)
{
    hipLaunchKernelGGL(count_labels_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const long long *)labels_dev, n, vocab, inv_count_dev);
    ECGB_CHECK_LAUNCH("count_labels");
}

extern "C" int ecgb_ce_fwd_bwd(void *logits_dev, const int64_t *labels_dev, float *row_loss_dev, float *sum_loss_dev,
                               const float *inv_count_dev, size_t rows, int vocab, size_t ld, void *stream)
{
    if (ld % 8) { ecgb::set_error("ecgb_ce_fwd_bwd: leading dimension must be a multiple of 8"); return ECGB_ERR_INVALID; }
    if (g_ce_in_registers && ld <= (size_t)40 * 512 * 8) {           // the row fits 512 threads x 40 chunks of 8: one read, one write
        hipLaunchKernelGGL(ce_fwd_bwd_reg_kernel<40>, dim3((unsigned)std::min<size_t>(std::max<size_t>(rows, 1), 4096)), dim3(512), 0,
                           (hipStream_t)stream, (unsigned short *)logits_dev, (const long long *)labels_dev, row_loss_dev, sum_loss_dev,
                           inv_count_dev, rows, vocab, ld);
        ECGB_CHECK_LAUNCH("ce_fwd_bwd (registers)");
    }
    hipLaunchKernelGGL(ce_fwd_bwd_kernel, dim3((unsigned)std::min<size_t>(std::max<size_t>(rows, 1), 4096)), dim3(256), 0,
                       (hipStream_t)stream, (unsigned short *)logits_dev, (const long long *)labels_dev, row_loss_dev, sum_loss_dev,
                       inv_count_dev, rows, vocab, ld);
    ECGB_CHECK_LAUNCH("ce_fwd_bwd");
}

extern "C" int ecgb_sumsq(const void *g_dev, size_t n, int is_fp32, float *acc_dev, void *stream)
{
    const dim3 grid(grid_for(n, 256 * 8));
    if (is_fp32) hipLaunchKernelGGL(sumsq_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float *)g_dev, n, acc_dev);
    else hipLaunchKernelGGL(sumsq_kernel<unsigned short>, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned short *)g_dev, n, acc_dev);
    ECGB_CHECK_LAUNCH("sumsq");
}

extern "C" int ecgb_sumsq_multi_bf16(const void *const *ptrs_dev, const unsigned long long *counts_dev, const int *chunk_tensor_dev,
                                    const unsigned long long *chunk_off_dev, int n_chunks, float *acc_dev, float *partials_dev, void *stream)
{
    if (n_chunks <= 0) return ECGB_OK;
    if (!ptrs_dev || !counts_dev || !chunk_tensor_dev || !chunk_off_dev || !acc_dev) { ecgb::set_error("ecgb_sumsq_multi_bf16: NULL argument"); return ECGB_ERR_INVALID; }
    hipLaunchKernelGGL(sumsq_multi_kernel, dim3((unsigned)n_chunks), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short *const *)ptrs_dev, counts_dev, chunk_tensor_dev, chunk_off_dev, acc_dev, partials_dev);
    if (partials_dev) hipLaunchKernelGGL(ordered_sum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials_dev, n_chunks, acc_dev);
    ECGB_CHECK_LAUNCH("sumsq_multi");
}

extern "C" int ecgb_adam_step(void *param_dev, const void *grad_dev, int grad_is_fp32, float *m_dev, float *v_dev, size_t n,
                              const float *sumsq_dev, float max_norm, float lr, float beta1, float beta2, float eps,
                              float weight_decay, int step, void *stream)
{
    const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
    const dim3 grid(grid_for(n, 256 * 4));
    if (grad_is_fp32)
        hipLaunchKernelGGL(adam_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (unsigned short *)param_dev, (const float *)grad_dev,
                           m_dev, v_dev, n, sumsq_dev, max_norm, lr, beta1, beta2, eps, weight_decay, bc1, bc2);
    else
        hipLaunchKernelGGL(adam_kernel<unsigned short>, grid, dim3(256), 0, (hipStream_t)stream, (unsigned short *)param_dev,
                           (const unsigned short *)grad_dev, m_dev, v_dev, n, sumsq_dev, max_norm, lr, beta1, beta2, eps, weight_decay, bc1, bc2);
    ECGB_CHECK_LAUNCH("adam_step");
}


extern "C" int ecgb_adam_multi_bf16(void *const *params_dev, const void *const *grads_dev, float *const *m_dev, float *const *v_dev, const unsigned long long *counts_dev,
                                    const int *chunk_tensor_dev, const unsigned long long *chunk_off_dev, int n_chunks, const float *sumsq_dev, float max_norm,
                                    float lr, float beta1, float beta2, float eps, float weight_decay, int step, void *stream)
{
    if (n_chunks <= 0) return ECGB_OK;
    if (!params_dev || !grads_dev || !m_dev || !v_dev || !counts_dev || !chunk_tensor_dev || !chunk_off_dev || !sumsq_dev) { ecgb::set_error("ecgb_adam_multi_bf16: NULL argument"); return ECGB_ERR_INVALID; }
    const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)n_chunks), dim3(256), 0, (hipStream_t)stream, (unsigned short *const *)params_dev,
                       (const unsigned short *const *)grads_dev, m_dev, v_dev, counts_dev, chunk_tensor_dev, chunk_off_dev, sumsq_dev, max_norm, lr, beta1, beta2, eps,
                       weight_decay, bc1, bc2);
    ECGB_CHECK_LAUNCH("adam_multi");
}

extern "C" int ecgb_transpose_bf16(const void *in_dev, void *out_dev, int rows, int cols, void *stream)
{
    const size_t tiles = (size_t)((rows + 63) / 64) * ((cols + 63) / 64);
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)std::min<size_t>(std::max<size_t>(tiles, 1), 256 * 16)), dim3(256), 0,
                       (hipStream_t)stream, (const unsigned short *)in_dev, (unsigned short *)out_dev, rows, cols);
    ECGB_CHECK_LAUNCH("transpose_bf16");
}

extern "C" int ecgb_transpose_multi_bf16(const void *const *src_dev, void *const *dst_dev, const int *rows_dev, const int *cols_dev,
                                         const int *tile_off_dev, int n, int total_tiles, void *stream)
{
    if (n <= 0 || total_tiles <= 0) return ECGB_OK;
    if (!src_dev || !dst_dev || !rows_dev || !cols_dev || !tile_off_dev) { ecgb::set_error("ecgb_transpose_multi_bf16: NULL argument"); return ECGB_ERR_INVALID; }
    hipLaunchKernelGGL(transpose_multi_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream, (const unsigned short *const *)src_dev,
                       (unsigned short *const *)dst_dev, rows_dev, cols_dev, tile_off_dev, n);
    ECGB_CHECK_LAUNCH("transpose_multi");
}

extern "C" int ecgb_f32_to_bf16(const float *in_dev, void *out_dev, size_t n, void *stream)
{
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(grid_for(n, 256 * 4)), dim3(256), 0, (hipStream_t)stream, in_dev, (unsigned short *)out_dev, n);
    ECGB_CHECK_LAUNCH("f32_to_bf16");
}

extern "C" int ecgb_sum_slabs_bf16(const float *slabs_dev, long long slab_stride, int n_slabs, void *out_dev, size_t n, int accumulate, void *stream)
{
    if (!slabs_dev || !out_dev || n_slabs < 1) { ecgb::set_error("ecgb_sum_slabs_bf16: bad argument"); return ECGB_ERR_INVALID; }
    if (n % 4 || slab_stride % 4 || ((uintptr_t)slabs_dev & 15) || ((uintptr_t)out_dev & 7)) {
        ecgb::set_error("ecgb_sum_slabs_bf16: n and slab_stride must be multiples of 4, slabs 16-byte and out 8-byte aligned");
        return ECGB_ERR_UNSUPPORTED;
    }
    if (n == 0) return ECGB_OK;
    const unsigned grid = grid_for(n / 4, 256);
    if (accumulate)
        hipLaunchKernelGGL(sum_slabs_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, slabs_dev, slab_stride, n_slabs, (unsigned short *)out_dev, n / 4);
    else
        hipLaunchKernelGGL(sum_slabs_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, slabs_dev, slab_stride, n_slabs, (unsigned short *)out_dev, n / 4);
    ECGB_CHECK_LAUNCH("sum_slabs");
}

extern "C" int ecgb_softmax_causal_fwd(void *scores_dev, const float *attn_mask_dev, int batch_heads, int n_heads, int seq,
                                       float scale, void *stream)
{
    if (seq % 8) { ecgb::set_error("ecgb_softmax_causal_fwd: seq must be a multiple of 8"); return ECGB_ERR_INVALID; }
    const size_t rows = (size_t)batch_heads * seq;
    hipLaunchKernelGGL(softmax_causal_fwd_kernel, dim3(grid_for(rows, 4)), dim3(256), 0, (hipStream_t)stream,
                       (unsigned short *)scores_dev, attn_mask_dev, rows, n_heads, seq, scale);
    ECGB_CHECK_LAUNCH("softmax_causal_fwd");
}

extern "C" int ecgb_softmax_bwd(const void *p_dev, void *dp_dev, int batch_heads, int seq, float scale, void *stream)
{
    if (seq % 8) { ecgb::set_error("ecgb_softmax_bwd: seq must be a multiple of 8"); return ECGB_ERR_INVALID; }
    const size_t rows = (size_t)batch_heads * seq;
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3(grid_for(rows, 4)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short *)p_dev, (unsigned short *)dp_dev, rows, seq, scale);
    ECGB_CHECK_LAUNCH("softmax_bwd");
}

extern "C" int ecgb_transpose_bf16_strided(const void *in_dev, void *out_dev, int rows, int cols, long long ld_in, long long ld_out,
                                           int batch, int inner, long long outer_in, long long inner_in, long long outer_out,
                                           long long inner_out, void *stream)
{
    if (batch <= 0 || inner <= 0 || batch > 65535) { ecgb::set_error("ecgb_transpose_bf16_strided: bad batch"); return ECGB_ERR_INVALID; }
    const size_t tiles = (size_t)((rows + 63) / 64) * ((cols + 63) / 64);
    hipLaunchKernelGGL(transpose_strided_kernel, dim3((unsigned)std::min<size_t>(std::max<size_t>(tiles, 1), 1024), (unsigned)batch),
                       dim3(256), 0, (hipStream_t)stream, (const unsigned short *)in_dev, (unsigned short *)out_dev, rows, cols,
                       ld_in, ld_out, inner, outer_in, inner_in, outer_out, inner_out);
    ECGB_CHECK_LAUNCH("transpose_bf16_strided");
}

extern "C" int ecgb_dropout_bf16(const void *x_dev, void *out_dev, size_t n, float p, uint64_t seed, void *stream)
{
    if (n % 8 || !(p >= 0.f && p < 1.f)) { ecgb::set_error("ecgb_dropout_bf16: n % 8 == 0 and 0 <= p < 1 required"); return ECGB_ERR_INVALID; }
    hipLaunchKernelGGL(dropout_kernel, dim3(grid_for(n / 8, 256)), dim3(256), 0, (hipStream_t)stream, (const unsigned short *)x_dev,
                       (unsigned short *)out_dev, n / 8, p, 1.0f / (1.0f - p), (unsigned long long)seed);
    ECGB_CHECK_LAUNCH("dropout_bf16");
}
