// decode.hip -- the decode step of generate() (one new token per sequence, batch <= 2) as SIX launches per layer instead of twelve (round 5).
//
// Reference: LlamaDecoderLayer.forward / GemmaDecoderLayer with a DynamicCache (modeling_llama.py:635-701, cache_utils.py:408-470), called per token by
// GenerationMixin._sample (generation/utils.py:3131) through ecg_byte/models/llm.py:26-37.
//
// Round 4's step ran twelve kernels per layer (rmsnorm + LoRA-down, q|k|v, RoPE + append, attention scores / values / combine, LoRA-down of o, o, rmsnorm, gate|up + GLU,
// LoRA-down of down, down): 122 us per Gemma-2B layer of which the weight streams themselves need 36 -- 40 us per layer went to kernels that move almost no bytes
// (profiles/r04/c5_kernel_stats.csv).  A dependent kernel boundary costs 1.2-1.9 us on this chip and a grid-wide barrier inside a launch 4 us or more
// (MI355X_MICROARCH.md, price table rows `boundary` / `barrier-xcd`): the step stays a chain of launches, cut at its all-to-all seams, with everything between two seams
// in ONE kernel:
//   1. decode_norm_gemv    residual add + RMSNorm + the site's LoRA down-projection t = scale * h A^T (every workgroup redoes them: a 4 KB row and 48 rows of A out of
//                          L2) + the projection (+ B t) -- q|k|v, and gate|up with the GLU
//   2. decode_attn_scores  RoPE of q and of the new key (rope_append's arithmetic), the cache append, and the scores of ALL query heads of a KV group against a
//                          split of the keys (the split's keys are read once for the group's heads: Gemma's eight query heads shared one KV head and read it eight times)
//   3. decode_attn_values  softmax statistics over the splits, P.V of the split for all heads of the group; the workgroup that arrives LAST (ticket) adds the splits'
//                          partial outputs in split order -- the combine launch
//   4. decode_gemv         o projection; its LoRA down-projection of the attention output in the prologue
//   5. decode_norm_gemv    (gate|up + GLU, above)
//   6. decode_lora_t + decode_gemv   down projection (its t over the 16 384-wide GLU output is a kernel of its own: redone per workgroup it would read A 128 times)
// Arithmetic: every sum is formed in the order the kernels it replaces form it (rmsnorm_lora_fwd_kernel, gemm_nt_skinny_kernel<2, 1 / 4>, gemm_nt_skinny_glu_kernel,
// rope_append_kernel, attn_decode_scores / values / combine): the same bits, checked op by op in tests/test_gpu_decode_fused.py.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>

#include <string>

#include "glu_math.hpp"
#include "tokenizer.hpp"

namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) unsigned short;

__device__ __forceinline__ float bf2f(unsigned short u) { return __uint_as_float((unsigned)u << 16); }
__device__ __forceinline__ unsigned short f2bf(float f)
{
    __hip_bfloat16 b = __float2bfloat16(f);   // round to nearest even
    return *reinterpret_cast<unsigned short *>(&b);
}
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

constexpr int kMaxRows = 2;          // sequences per step the fused kernels take
constexpr int kTRows = 64;           // rows of a site's stacked A (and columns of its B)

int check(hipError_t e, const char *what)
{
    if (e == hipSuccess) return ECGB_OK;
    ecgb::set_error(std::string(what) + ": " + hipGetErrorString(e));
    return ECGB_ERR_HIP;
}

// ---- shared pieces ------------------------------------------------------------------------------------------------------------------------------------------
// rmsnorm_lora_fwd_kernel's norm of row r by ONE wave: (x + delta) rounded to bf16, rs = rsqrt(mean square + eps), y = bf16(bf16(v rs) w) (Llama) or bf16(v rs (1 + w)) (Gemma).
// s_y: the row in LDS (first the sum, then the normalised row).  sum_out: where the residual sum goes (null: not written).
template <bool GEMMA>
__device__ __forceinline__ void norm_row(const unsigned short *pa, const unsigned short *pb, const unsigned short *w, unsigned short *s_y, unsigned short *sum_out, int H, float eps, int lane)
{
    float ss = 0.f;
    for (int c = lane * 8; c < H; c += 64 * 8) {
        bf16x8 v = *reinterpret_cast<const bf16x8 *>(pa + c);
        if (pb) {
            const bf16x8 u = *reinterpret_cast<const bf16x8 *>(pb + c);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = f2bf(bf2f(v[j]) + bf2f(u[j]));
            if (sum_out) *reinterpret_cast<bf16x8 *>(sum_out + c) = v;
        }
        *reinterpret_cast<bf16x8 *>(s_y + c) = v;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float f = bf2f(v[j]); ss += f * f; }
    }
    ss = wave_sum(ss);
    const float rs = rsqrtf(ss / (float)H + eps);
    for (int c = lane * 8; c < H; c += 64 * 8) {
        const bf16x8 v = *reinterpret_cast<const bf16x8 *>(s_y + c);
        const bf16x8 g = *reinterpret_cast<const bf16x8 *>(w + c);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (GEMMA) o[j] = f2bf(bf2f(v[j]) * rs * (1.0f + bf2f(g[j])));
            else o[j] = f2bf(bf2f(f2bf(bf2f(v[j]) * rs)) * bf2f(g[j]));
        }
        *reinterpret_cast<bf16x8 *>(s_y + c) = o;
    }
}

// t[m][ra] = bf16(scale * sum_k a[m][k] A[ra][k]) for the n_a used rows of A, the rest zero: one (row of A, sequence) pair per wave at a time, the sum in
// rmsnorm_lora_fwd_kernel's / gemm_nt_skinny_kernel's order (a lane takes the 16-byte pieces lane, lane + 64, ... of the row, in order; then the xor-shuffle tree).
__device__ __forceinline__ void lora_t_rows(const unsigned short *s_a, int lds_stride, int M, int K, const unsigned short *A, long long lda, int n_a, float scale,
                                            unsigned short *s_t, int wave, int n_waves, int lane)
{
    for (int i = threadIdx.x; i < M * kTRows; i += blockDim.x) s_t[i] = 0;
    __syncthreads();
    for (int idx = wave; idx < n_a * M; idx += n_waves) {
        const int ra = idx % n_a, m = idx / n_a;
        const unsigned short *pa = A + (long long)ra * lda;
        float acc = 0.f;
        for (int k = lane * 8; k < K; k += 512) {
            const bf16x8 vb = *reinterpret_cast<const bf16x8 *>(pa + k);
            const bf16x8 va = *reinterpret_cast<const bf16x8 *>(s_a + m * lds_stride + k);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += bf2f(va[j]) * bf2f(vb[j]);
        }
        acc = wave_sum(acc);
        if (lane == 0) s_t[m * kTRows + ra] = f2bf(acc * scale);
    }
    __syncthreads();
}

// One output column: sum_k a[m][k] W[n][k] over k in [k_lo, k_hi), four 16-byte pieces of the weight row in flight per lane, gemm_nt_skinny_kernel's order (piece
// after piece, the eight elements of a piece in order), activations out of LDS; then the LoRA pair (K2 = 64: lanes 0..7 hold one piece each) when `with_pair`.
template <int MR>
__device__ __forceinline__ void column_dot(float (&acc)[MR], const unsigned short *wrow, int k_lo, int k_hi, const unsigned short *s_a, int lds_stride, int M, int lane)
{
    constexpr int U = 4;
    for (int k0 = k_lo + lane * 8; k0 < k_hi; k0 += 512 * U) {
        bf16x8 vb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) vb[u] = *reinterpret_cast<const bf16x8 *>(wrow + min(k0 + 512 * u, k_hi - 8));
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = k0 + 512 * u;
            if (k < k_hi) {
                float fb[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) fb[j] = bf2f(vb[u][j]);
#pragma unroll
                for (int m = 0; m < MR; ++m) {
                    if (m < M) {
                        const bf16x8 va = *reinterpret_cast<const bf16x8 *>(s_a + m * lds_stride + k);
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[m] += bf2f(va[j]) * fb[j];
                    }
                }
            }
        }
    }
}
template <int MR>
__device__ __forceinline__ void pair_dot(float (&acc)[MR], const unsigned short *brow, const unsigned short *s_t, int M, int lane)
{
    for (int k = lane * 8; k < kTRows; k += 512) {
        const bf16x8 vb = *reinterpret_cast<const bf16x8 *>(brow + k);
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            if (m < M) {
                const bf16x8 va = *reinterpret_cast<const bf16x8 *>(s_t + m * kTRows + k);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[m] += bf2f(va[j]) * bf2f(vb[j]);
            }
        }
    }
}

struct NormGemvArgs {
    const unsigned short *x, *delta, *norm_w;    // [M, H] residual stream, what is still to be added to it (or null), the norm's weight
    unsigned short *x_out;                        // [M, H] x + delta (written by workgroup 0 when delta is given)
    const unsigned short *W;                      // [N, H] (GLU: [2 I, H], gate rows then up rows, N = I)
    long long ldw;
    const unsigned short *lA, *lB;                // the site's adapters: A [64, H] (n_a rows used), B [N (GLU: 2 I), 64]; null: none
    long long lda, ldb;
    unsigned short *y;                            // [M, ldy]
    long long ldy;
    int M, H, N, n_a, glu_I;
    float eps, lscale;
};

// 1 / 5: residual add + RMSNorm + LoRA down-projection + projection.  EPI 0: plain columns; 1 / 2: SiLU / tanh-GELU GLU (a wave computes gate column n and up column n + I).
// Four waves, COLS columns a wave.
template <bool GEMMA, int EPI, int COLS>
__global__ __launch_bounds__(256) void decode_norm_gemv_kernel(NormGemvArgs G)
{
    extern __shared__ __align__(16) unsigned short smem[];
    unsigned short *s_y = smem;                                  // [M][H]
    unsigned short *s_t = smem + kMaxRows * G.H;                 // [M][64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave < G.M)
        norm_row<GEMMA>(G.x + (size_t)wave * G.H, G.delta ? G.delta + (size_t)wave * G.H : nullptr, G.norm_w, s_y + wave * G.H,
                        (blockIdx.x == 0 && G.delta) ? G.x_out + (size_t)wave * G.H : nullptr, G.H, G.eps, lane);
    __syncthreads();
    if (G.lA) lora_t_rows(s_y, G.H, G.M, G.H, G.lA, G.lda, G.n_a, G.lscale, s_t, wave, 4, lane);
    const long long n0 = ((long long)blockIdx.x * 4 + wave) * COLS;
    if constexpr (EPI == 0) {
        float acc[COLS][kMaxRows];
#pragma unroll
        for (int c = 0; c < COLS; ++c) {
            const long long n = min(n0 + c, (long long)G.N - 1);
#pragma unroll
            for (int m = 0; m < kMaxRows; ++m) acc[c][m] = 0.f;
            column_dot<kMaxRows>(acc[c], G.W + n * G.ldw, 0, G.H, s_y, G.H, G.M, lane);
            if (G.lA) pair_dot<kMaxRows>(acc[c], G.lB + n * G.ldb, s_t, G.M, lane);
        }
#pragma unroll
        for (int c = 0; c < COLS; ++c)
#pragma unroll
            for (int m = 0; m < kMaxRows; ++m) acc[c][m] = wave_sum(acc[c][m]);
        if (lane == 0) {
#pragma unroll
            for (int c = 0; c < COLS; ++c)
                if (n0 + c < G.N)
                    for (int m = 0; m < G.M; ++m) G.y[(long long)m * G.ldy + n0 + c] = f2bf(acc[c][m]);
        }
    } else {
        float ag[COLS][kMaxRows], au[COLS][kMaxRows];
#pragma unroll
        for (int c = 0; c < COLS; ++c) {
            const long long n = min(n0 + c, (long long)G.N - 1);
#pragma unroll
            for (int m = 0; m < kMaxRows; ++m) { ag[c][m] = 0.f; au[c][m] = 0.f; }
            // gemm_nt_skinny_glu_kernel: two pieces of each of the two rows in flight, gate before up for every piece
            const unsigned short *bg = G.W + n * G.ldw, *bu = bg + (long long)G.glu_I * G.ldw;
            constexpr int U = 2;
            for (int k0 = lane * 8; k0 < G.H; k0 += 512 * U) {
                bf16x8 vg[U], vu[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int k = min(k0 + 512 * u, G.H - 8);
                    vg[u] = *reinterpret_cast<const bf16x8 *>(bg + k);
                    vu[u] = *reinterpret_cast<const bf16x8 *>(bu + k);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int k = k0 + 512 * u;
                    if (k < G.H) {
#pragma unroll
                        for (int m = 0; m < kMaxRows; ++m) {
                            if (m < G.M) {
                                const bf16x8 va = *reinterpret_cast<const bf16x8 *>(s_y + m * G.H + k);
#pragma unroll
                                for (int j = 0; j < 8; ++j) ag[c][m] += bf2f(va[j]) * bf2f(vg[u][j]);
#pragma unroll
                                for (int j = 0; j < 8; ++j) au[c][m] += bf2f(va[j]) * bf2f(vu[u][j]);
                            }
                        }
                    }
                }
            }
            if (G.lA) {
                pair_dot<kMaxRows>(ag[c], G.lB + n * G.ldb, s_t, G.M, lane);
                pair_dot<kMaxRows>(au[c], G.lB + (n + (long long)G.glu_I) * G.ldb, s_t, G.M, lane);
            }
        }
#pragma unroll
        for (int c = 0; c < COLS; ++c)
#pragma unroll
            for (int m = 0; m < kMaxRows; ++m) { ag[c][m] = wave_sum(ag[c][m]); au[c][m] = wave_sum(au[c][m]); }
        if (lane == 0) {
#pragma unroll
            for (int c = 0; c < COLS; ++c)
                if (n0 + c < G.N)
                    for (int m = 0; m < G.M; ++m) {
                        const unsigned short g16 = f2bf(ag[c][m]), u16 = f2bf(au[c][m]);
                        const float a = bf2f(f2bf(ecgb::glu_act<EPI == 2>(bf2f(g16))));
                        G.y[(long long)m * G.ldy + n0 + c] = f2bf(a * bf2f(u16));
                    }
        }
    }
}

struct GemvArgs {
    const unsigned short *a;                      // [M, K] activations
    long long lda_act;
    const unsigned short *W;                      // [N, K]
    long long ldw;
    const unsigned short *lA, *lB, *t_in;         // adapters: A [64, K] (t formed here), or t_in [M, 64] formed before; B [N, 64]
    long long lda, ldb;
    unsigned short *y;
    long long ldy;
    int M, K, N, n_a;
    float lscale;
};

// 4 / 6: a projection of a few rows.  KS = 1: a wave a column; KS = 4: the four waves of a workgroup share a column, a contiguous quarter of the contraction each
// (gemm_nt_skinny_kernel<2, 4>: long rows, few columns), their sums meet in LDS in wave order.
// STAGE: the activations go through LDS first (needed where the adapter's t is formed here: every wave reads them sixteen times); otherwise the lanes read them from
// global memory as gemm_nt_skinny_kernel does (L2-resident, a few KB) -- staging 32 KB of them per workgroup of a long-row projection cost more than the projection.
template <int KS, bool STAGE>
__global__ __launch_bounds__(256) void decode_gemv_kernel(GemvArgs G)
{
    extern __shared__ __align__(16) unsigned short smem[];
    unsigned short *s_t = smem;                                  // [M][64]
    unsigned short *s_stage = smem + kMaxRows * kTRows;          // [M][K] when STAGE
    __shared__ float s_part[4][kMaxRows];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, ks = wave % KS;
    const unsigned short *s_a = G.a;
    int a_stride = (int)G.lda_act;
    if constexpr (STAGE) {
        for (int i = threadIdx.x * 8; i < G.M * G.K; i += 256 * 8) {
            const int m = i / G.K, k = i % G.K;
            *reinterpret_cast<bf16x8 *>(s_stage + m * G.K + k) = *reinterpret_cast<const bf16x8 *>(G.a + (long long)m * G.lda_act + k);
        }
        __syncthreads();
        s_a = s_stage; a_stride = G.K;
        if (G.lA && !G.t_in) lora_t_rows(s_stage, G.K, G.M, G.K, G.lA, G.lda, G.n_a, G.lscale, s_t, wave, 4, lane);
    }
    const unsigned short *t_src = G.t_in ? G.t_in : s_t;       // (t formed before: read where it lies, as gemm_nt_skinny_kernel reads its second operand pair)
    const long long n = (long long)blockIdx.x * (4 / KS) + wave / KS;
    const bool live = n < G.N;
    const unsigned short *wrow = G.W + (live ? n : 0) * G.ldw;
    const int k_lo = ks * (G.K / KS), k_hi = k_lo + G.K / KS;
    float acc[kMaxRows];
#pragma unroll
    for (int m = 0; m < kMaxRows; ++m) acc[m] = 0.f;
    column_dot<kMaxRows>(acc, wrow, k_lo, k_hi, s_a, a_stride, G.M, lane);
    if ((G.lA || G.t_in) && ks == 0) pair_dot<kMaxRows>(acc, G.lB + (live ? n : 0) * G.ldb, t_src, G.M, lane);
#pragma unroll
    for (int m = 0; m < kMaxRows; ++m) acc[m] = wave_sum(acc[m]);
    if constexpr (KS > 1) {
        if (lane == 0) {
#pragma unroll
            for (int m = 0; m < kMaxRows; ++m) s_part[wave][m] = acc[m];
        }
        __syncthreads();
        if (ks == 0 && lane == 0) {
#pragma unroll
            for (int m = 0; m < kMaxRows; ++m)
#pragma unroll
                for (int t = 1; t < KS; ++t) acc[m] += s_part[wave + t][m];
        }
    }
    if (live && ks == 0 && lane == 0)
        for (int m = 0; m < G.M; ++m) G.y[(long long)m * G.ldy + n] = f2bf(acc[m]);
}

// 6a: t = bf16(scale * a A^T) of a wide site (K = 16 384): a wave a (row of A, sequence) pair, the four waves of a workgroup a contiguous quarter of the contraction each --
// gemm_nt_skinny_kernel<2, 4> on B = A with alpha = scale, the same bits.
__global__ __launch_bounds__(256) void decode_lora_t_kernel(const unsigned short *a, long long lda_act, const unsigned short *A, long long lda, int M, int K, int n_a, float scale,
                                                            unsigned short *t_out)
{
    __shared__ float s_part[4][kMaxRows];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ra = blockIdx.x;
    const int k_lo = wave * (K / 4), k_hi = k_lo + K / 4;
    float acc[kMaxRows];
#pragma unroll
    for (int m = 0; m < kMaxRows; ++m) acc[m] = 0.f;
    constexpr int U = 4;
    const unsigned short *b = A + (long long)ra * lda;
    for (int k0 = k_lo + lane * 8; k0 < k_hi; k0 += 512 * U) {
        bf16x8 vb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) vb[u] = *reinterpret_cast<const bf16x8 *>(b + min(k0 + 512 * u, k_hi - 8));
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = k0 + 512 * u;
            if (k < k_hi) {
#pragma unroll
                for (int m = 0; m < kMaxRows; ++m) {
                    if (m < M) {
                        const bf16x8 va = *reinterpret_cast<const bf16x8 *>(a + (long long)m * lda_act + k);
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[m] += bf2f(va[j]) * bf2f(vb[u][j]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int m = 0; m < kMaxRows; ++m) acc[m] = wave_sum(acc[m]);
    if (lane == 0) {
#pragma unroll
        for (int m = 0; m < kMaxRows; ++m) s_part[wave][m] = acc[m];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int m = 0; m < M; ++m) {
            float s = s_part[0][m];
            for (int t = 1; t < 4; ++t) s += s_part[t][m];
            t_out[m * kTRows + ra] = f2bf(s * scale);
        }
    }
    if (blockIdx.x == 0) {                                     // the unused rows of A are zero: so is their t
        for (int i = threadIdx.x; i < M * kTRows; i += 256)
            if (i % kTRows >= n_a) t_out[i] = 0;
    }
}

// ---- attention of one new token ---------------------------------------------------------------------------------------------------------------------------------
// scratch (floats): scores [B, Hq, cap] | stats [B, Hq, splits, 2] | partial outputs [B, Hq, splits, D] | tickets [B, Hkv] (unsigned, zero between launches)
constexpr int kAttnThreads = 256;
constexpr int kAttnMaxChunk = 1024;                 // keys per split (the launcher raises the number of splits to fit)
constexpr int kAttnMaxG = 8;                        // query heads per KV head

struct AttnArgs2 {
    unsigned short *qkv;          // [B, (Hq + 2 Hkv) D]: the step's projection; q and the new k are rotated in registers, v is read as it is
    long long ld_qkv;
    const float *cs, *sn;         // RoPE tables of the step's positions [B, D / 2]
    unsigned short *cache;        // [B, cap, 2 Hkv D]: keys | values
    long long cap;
    const float *mask;            // [B, mask_ld] by value (a key takes part iff != 0)
    long long mask_ld;
    float *scores, *stats, *partial;
    unsigned *tickets;
    unsigned short *o;            // [B, Hq D]
    const int *len_dev;           // keys valid AFTER the append (the new token's row is len - 1)
    int len_arg, Hq, Hkv, n_splits;
    float scale;
};

// 2: RoPE (q heads of the group and the new key: rope_append_kernel's arithmetic), cache append (split 0 of every (group, sequence)), scores of the split's keys for every
// head of the group + the split's maximum and sum of exponentials per head (attn_decode_scores_kernel's sums, head by head).  grid (splits, Hkv, B).
template <int D>
__global__ __launch_bounds__(kAttnThreads) void decode_attn_scores_kernel(AttnArgs2 A)
{
    constexpr int NW = kAttnThreads / 64, EPL = D / 64, KU = 8, half = D / 2;
    __shared__ __align__(16) unsigned short s_q[kAttnMaxG][D];       // rotated query heads of the group
    __shared__ __align__(16) unsigned short s_k[D];                  // the rotated new key
    __shared__ float s_sc[kAttnMaxG][kAttnMaxChunk];
    __shared__ float s_red[kAttnMaxG][2 * NW];
    const int sp = blockIdx.x, g = blockIdx.y, b = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = A.Hq / A.Hkv;
    const int len = A.len_dev ? *A.len_dev : A.len_arg, chunk = (len + (int)gridDim.x - 1) / (int)gridDim.x;
    const int k0 = sp * chunk, k1 = min(len, k0 + chunk);
    unsigned short *row = A.qkv + (long long)b * A.ld_qkv;
    const int width = 2 * A.Hkv * D;
    // ---- RoPE: piece c (eight elements) of the first half of a head with piece c of the second half
    for (int i = tid; i < (G + 1) * (half / 8); i += kAttnThreads) {
        const int c = i % (half / 8), h = i / (half / 8);                  // h < G: query head g G + h; h == G: the new key of KV head g
        const unsigned short *p = row + (long long)(h < G ? g * G + h : A.Hq + g) * D + c * 8;
        bf16x8 a = *reinterpret_cast<const bf16x8 *>(p), bb = *reinterpret_cast<const bf16x8 *>(p + half);
        const float *pc = A.cs + (size_t)b * half + c * 8, *ps = A.sn + (size_t)b * half + c * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float cc = bf2f(f2bf(pc[j])), ss = bf2f(f2bf(ps[j]));
            const float x1 = bf2f(a[j]), x2 = bf2f(bb[j]);
            a[j] = f2bf(x1 * cc - x2 * ss); bb[j] = f2bf(x2 * cc + x1 * ss);
        }
        unsigned short *dst = h < G ? s_q[h] : s_k;
        *reinterpret_cast<bf16x8 *>(dst + c * 8) = a;
        *reinterpret_cast<bf16x8 *>(dst + half + c * 8) = bb;
        if (h == G && sp == 0) {                                           // the append: rotated key
            unsigned short *kc = A.cache + ((long long)b * A.cap + (len - 1)) * width + (size_t)g * D + c * 8;
            *reinterpret_cast<bf16x8 *>(kc) = a;
            *reinterpret_cast<bf16x8 *>(kc + half) = bb;
        }
    }
    if (sp == 0) {                                                         // the append: the value as it is
        for (int i = tid; i < D / 8; i += kAttnThreads)
            *reinterpret_cast<bf16x8 *>(A.cache + ((long long)b * A.cap + (len - 1)) * width + (size_t)A.Hkv * D + (size_t)g * D + i * 8) =
                *reinterpret_cast<const bf16x8 *>(row + (long long)(A.Hq + A.Hkv + g) * D + i * 8);
    }
    __syncthreads();
    float qf[kAttnMaxG][EPL];
#pragma unroll
    for (int h = 0; h < kAttnMaxG; ++h)
#pragma unroll
        for (int t = 0; t < EPL; ++t) qf[h][t] = h < G ? bf2f(s_q[h][lane * EPL + t]) : 0.f;
    float kn[EPL];
#pragma unroll
    for (int t = 0; t < EPL; ++t) kn[t] = bf2f(s_k[lane * EPL + t]);
    const unsigned short *K = A.cache + (long long)b * A.cap * width + (long long)g * D;
    const float *mrow = A.mask + (long long)b * A.mask_ld;
    float m[kAttnMaxG];
#pragma unroll
    for (int h = 0; h < kAttnMaxG; ++h) m[h] = -INFINITY;
    for (int j0 = k0 + wave; j0 < k1; j0 += KU * NW) {
        float kf[KU][EPL], mk[KU];
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const int j = min(j0 + u * NW, k1 - 1);
            mk[u] = mrow[j];
            if (j == len - 1) {                                            // the new key: out of registers (its cache row is being written by another workgroup)
#pragma unroll
                for (int t = 0; t < EPL; ++t) kf[u][t] = kn[t];
            } else {
                const unsigned short *kr = K + (long long)j * width + lane * EPL;
                if constexpr (EPL == 4) {
                    const uint2 kv = *reinterpret_cast<const uint2 *>(kr);
                    kf[u][0] = __uint_as_float(kv.x << 16); kf[u][1] = __uint_as_float(kv.x & 0xFFFF0000u);
                    kf[u][2] = __uint_as_float(kv.y << 16); kf[u][3] = __uint_as_float(kv.y & 0xFFFF0000u);
                } else if constexpr (EPL == 2) {
                    const unsigned kv = *reinterpret_cast<const unsigned *>(kr);
                    kf[u][0] = __uint_as_float(kv << 16); kf[u][1] = __uint_as_float(kv & 0xFFFF0000u);
                } else {
                    kf[u][0] = bf2f(kr[0]);
                }
            }
        }
#pragma unroll
        for (int h = 0; h < kAttnMaxG; ++h) {
            if (h < G) {
                float part[KU];
#pragma unroll
                for (int u = 0; u < KU; ++u) {
                    if constexpr (EPL == 4) part[u] = qf[h][0] * kf[u][0] + qf[h][1] * kf[u][1] + qf[h][2] * kf[u][2] + qf[h][3] * kf[u][3];
                    else if constexpr (EPL == 2) part[u] = qf[h][0] * kf[u][0] + qf[h][1] * kf[u][1];
                    else part[u] = qf[h][0] * kf[u][0];
                }
#pragma unroll
                for (int d = 32; d > 0; d >>= 1)
#pragma unroll
                    for (int u = 0; u < KU; ++u) part[u] += __shfl_xor(part[u], d, 64);
                float *srow = A.scores + ((long long)b * A.Hq + g * G + h) * A.cap;
#pragma unroll
                for (int u = 0; u < KU; ++u) {
                    const int j = j0 + u * NW;
                    if (j < k1) {
                        const float sdot = (mk[u] != 0.f) ? part[u] * A.scale : -INFINITY;
                        if (lane == 0) { srow[j] = sdot; s_sc[h][j - k0] = sdot; }
                        m[h] = fmaxf(m[h], sdot);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int h = 0; h < kAttnMaxG; ++h)
        if (h < G && lane == 0) s_red[h][wave] = m[h];
    __syncthreads();
#pragma unroll
    for (int h = 0; h < kAttnMaxG; ++h) {
        if (h < G) {
            float mm = s_red[h][0];
#pragma unroll
            for (int w = 1; w < NW; ++w) mm = fmaxf(mm, s_red[h][w]);
            float l = 0.f;
            for (int j = k0 + tid; j < k1; j += kAttnThreads) l += (mm == -INFINITY) ? 0.f : __expf(s_sc[h][j - k0] - mm);
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) l += __shfl_xor(l, d, 64);
            if (lane == 0) s_red[h][NW + wave] = l;
            m[h] = mm;
        }
    }
    __syncthreads();
    if (tid < G) {
        float l = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) l += s_red[tid][NW + w];
        float mm = s_red[tid][0];
#pragma unroll
        for (int w = 1; w < NW; ++w) mm = fmaxf(mm, s_red[tid][w]);
        float *st = A.stats + (((long long)b * A.Hq + g * G + tid) * gridDim.x + sp) * 2;
        st[0] = mm;
        st[1] = l;
    }
}

// 3: softmax over the splits' statistics, P.V of the split's keys for every head of the group (attn_decode_values_kernel's sums), and -- the workgroup of a (group,
// sequence) that takes the last ticket -- the sum of the splits' partial outputs in split order (attn_decode_combine_kernel).  grid (splits, Hkv, B).
template <int D>
__global__ __launch_bounds__(kAttnThreads) void decode_attn_values_kernel(AttnArgs2 A)
{
    constexpr int TPR = D / 8, NS = kAttnThreads / TPR, VU = 4;
    __shared__ float s_part[NS * D];
    __shared__ unsigned s_last;
    const int sp = blockIdx.x, g = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
    const int G = A.Hq / A.Hkv, n_splits = gridDim.x;
    const int len = A.len_dev ? *A.len_dev : A.len_arg, chunk = (len + n_splits - 1) / n_splits;
    const int k0 = sp * chunk, k1 = min(len, k0 + chunk);
    const int width = 2 * A.Hkv * D;
    const unsigned short *V = A.cache + (long long)b * A.cap * width + (long long)A.Hkv * D + (long long)g * D;
    const int piece = tid % TPR, slice = tid / TPR;
    __shared__ float s_st[kAttnMaxG][2 * 64];                     // the splits' (maximum, sum) pairs of every head of the group
    for (int i = tid; i < G * n_splits * 2; i += kAttnThreads) {
        const int h = i / (n_splits * 2), r = i % (n_splits * 2);
        s_st[h][r] = A.stats[((long long)b * A.Hq + g * G + h) * n_splits * 2 + r];
    }
    __syncthreads();
    float mx[kAttnMaxG], inv[kAttnMaxG];
#pragma unroll
    for (int h = 0; h < kAttnMaxG; ++h) {
        mx[h] = -INFINITY; inv[h] = 0.f;
        if (h < G) {
            const float *st = s_st[h];
            float m = -INFINITY;
            for (int t = 0; t < n_splits; ++t) m = fmaxf(m, st[2 * t]);
            float l = 0.f;
            for (int t = 0; t < n_splits; ++t) l += (st[2 * t] == -INFINITY) ? 0.f : st[2 * t + 1] * __expf(st[2 * t] - m);
            mx[h] = m; inv[h] = l > 0.f ? 1.f / l : 0.f;
        }
    }
    float acc[kAttnMaxG][8];
#pragma unroll
    for (int h = 0; h < kAttnMaxG; ++h)
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[h][t] = 0.f;
    for (int j0 = k0 + slice; j0 < k1; j0 += VU * NS) {
        bf16x8 vv[VU];
#pragma unroll
        for (int u = 0; u < VU; ++u) vv[u] = *reinterpret_cast<const bf16x8 *>(V + (long long)min(j0 + u * NS, k1 - 1) * width + piece * 8);
#pragma unroll
        for (int h = 0; h < kAttnMaxG; ++h) {
            if (h < G) {
                const float *srow = A.scores + ((long long)b * A.Hq + g * G + h) * A.cap;
#pragma unroll
                for (int u = 0; u < VU; ++u) {
                    const int j = j0 + u * NS;
                    const float sc = srow[min(j, k1 - 1)];
                    const float e = (mx[h] == -INFINITY) ? 0.f : __expf(sc - mx[h]);
                    const float pj = (j < k1) ? bf2f(f2bf(e * inv[h])) : 0.f;
#pragma unroll
                    for (int t = 0; t < 8; ++t) acc[h][t] += pj * bf2f(vv[u][t]);
                }
            }
        }
    }
    for (int h = 0; h < G; ++h) {
#pragma unroll
        for (int t = 0; t < 8; ++t) s_part[slice * D + piece * 8 + t] = acc[h][t];
        __syncthreads();
        if (tid < D) {
            float sum = 0.f;
#pragma unroll
            for (int sl = 0; sl < NS; ++sl) sum += s_part[sl * D + tid];
            A.partial[(((long long)b * A.Hq + g * G + h) * n_splits + sp) * D + tid] = sum;
        }
        __syncthreads();
    }
    // ---- the last workgroup of the (group, sequence) to get here adds the splits in order
    __threadfence();
    if (tid == 0) s_last = atomicAdd(&A.tickets[b * A.Hkv + g], 1u) == (unsigned)(n_splits - 1) ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    for (int i = tid; i < G * D; i += kAttnThreads) {
        const int h = i / D, d = i % D;
        const float *p = A.partial + ((long long)b * A.Hq + g * G + h) * n_splits * D + d;
        float sum = 0.f;
        for (int t0 = 0; t0 < n_splits; t0 += 16) {              // sixteen splits' values in flight, added in split order
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = p[(long long)min(t0 + u, n_splits - 1) * D];
#pragma unroll
            for (int u = 0; u < 16; ++u) if (t0 + u < n_splits) sum += v[u];
        }
        A.o[((long long)b * A.Hq + g * G + h) * D + d] = f2bf(sum);
    }
    if (tid == 0) A.tickets[b * A.Hkv + g] = 0u;               // ready for the next launch (a replayed graph never clears the buffer itself)
}

}  // namespace

// ---- C ABI ----------------------------------------------------------------------------------------------------------------------------------------------------
static bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

// residual add + RMSNorm (Llama: bf16(bf16(x rs) w); gemma != 0: bf16(x rs (1 + w))) + optional LoRA branch + projection of M <= 2 rows:
// glu 0: y[M, N] = h W^T (+ t B^T);  glu 1 / 2 (SiLU / tanh-GELU): W = [gate rows; up rows] of N each, y[M, N] = act(gate) * up.
// x_out (with delta): the new residual stream x + delta.  ECGB_ERR_UNSUPPORTED outside M <= 2, H % 512 == 0, H <= 8192 (the caller runs the separate kernels).
extern "C" int ecgb_decode_norm_gemv(const void *x_dev, const void *delta_dev, const void *norm_w_dev, float eps, int gemma, int M, int H, void *x_out_dev,
                                     const void *w_dev, long long ldw, int N, const void *lora_a_dev, long long lda, int n_a, float lora_scale, const void *lora_b_dev,
                                     long long ldb, void *y_dev, long long ldy, int glu, void *stream)
{
    if (!x_dev || !norm_w_dev || !w_dev || !y_dev || M <= 0 || N <= 0 || H <= 0 || (delta_dev && !x_out_dev) || (lora_a_dev && !lora_b_dev)) {
        ecgb::set_error("ecgb_decode_norm_gemv: bad argument");
        return ECGB_ERR_INVALID;
    }
    if (M > kMaxRows || H % 512 || H > 8192 || ldw % 8 || (lora_a_dev && (lda % 8 || ldb % 8 || n_a <= 0 || n_a > kTRows)) || glu < 0 || glu > 2 ||
        !aligned16(x_dev) || !aligned16(delta_dev) || !aligned16(norm_w_dev) || !aligned16(w_dev) || !aligned16(lora_a_dev) || !aligned16(lora_b_dev) || !aligned16(x_out_dev)) {
        ecgb::set_error("ecgb_decode_norm_gemv: M <= 2, H a multiple of 512 (<= 8192), 16-byte aligned operands");
        return ECGB_ERR_UNSUPPORTED;
    }
    NormGemvArgs G;
    G.x = (const unsigned short *)x_dev; G.delta = (const unsigned short *)delta_dev; G.norm_w = (const unsigned short *)norm_w_dev; G.x_out = (unsigned short *)x_out_dev;
    G.W = (const unsigned short *)w_dev; G.ldw = ldw; G.lA = (const unsigned short *)lora_a_dev; G.lB = (const unsigned short *)lora_b_dev; G.lda = lda; G.ldb = ldb;
    G.y = (unsigned short *)y_dev; G.ldy = ldy; G.M = M; G.H = H; G.N = N; G.n_a = n_a; G.glu_I = N; G.eps = eps; G.lscale = lora_scale;
    const size_t lds = (size_t)(kMaxRows * H + kMaxRows * kTRows) * 2;
    hipStream_t st = (hipStream_t)stream;
    // columns a wave: enough waves to fill the chip twice over, as few launch rounds as that allows
    if (glu == 0) {
        const unsigned grid = (unsigned)((N + 7) / 8);
        if (gemma) hipLaunchKernelGGL((decode_norm_gemv_kernel<true, 0, 2>), dim3(grid), dim3(256), lds, st, G);
        else hipLaunchKernelGGL((decode_norm_gemv_kernel<false, 0, 2>), dim3(grid), dim3(256), lds, st, G);
    } else {
        const unsigned grid = (unsigned)((N + 7) / 8);
        if (gemma) {
            if (glu == 2) hipLaunchKernelGGL((decode_norm_gemv_kernel<true, 2, 2>), dim3(grid), dim3(256), lds, st, G);
            else hipLaunchKernelGGL((decode_norm_gemv_kernel<true, 1, 2>), dim3(grid), dim3(256), lds, st, G);
        } else {
            if (glu == 2) hipLaunchKernelGGL((decode_norm_gemv_kernel<false, 2, 2>), dim3(grid), dim3(256), lds, st, G);
            else hipLaunchKernelGGL((decode_norm_gemv_kernel<false, 1, 2>), dim3(grid), dim3(256), lds, st, G);
        }
    }
    return check(hipGetLastError(), "decode_norm_gemv_kernel launch");
}

// y[M, N] = a W^T (+ t B^T), M <= 2.  lora_a_dev: t = bf16(scale a A^T) is formed in the kernel (K <= 4096); t_dev: t formed before (ecgb_decode_lora_t); both null: no adapter.
extern "C" int ecgb_decode_gemv(const void *a_dev, long long lda_act, int M, int K, const void *w_dev, long long ldw, int N, const void *lora_a_dev, long long lda, int n_a,
                                float lora_scale, const void *t_dev, const void *lora_b_dev, long long ldb, void *y_dev, long long ldy, void *stream)
{
    if (!a_dev || !w_dev || !y_dev || M <= 0 || N <= 0 || K <= 0 || ((lora_a_dev || t_dev) && !lora_b_dev)) { ecgb::set_error("ecgb_decode_gemv: bad argument"); return ECGB_ERR_INVALID; }
    const bool split = (K >= 8192 && K % 32 == 0 && N <= 8192);         // gemm_nt_skinny_kernel's own rule: four waves a column
    if (M > kMaxRows || K % 512 || K > 32768 || (split && K % 2048) || ldw % 8 || lda_act % 8 || (lora_a_dev && !t_dev && K > 4096) || (lora_a_dev && (lda % 8 || n_a <= 0 || n_a > kTRows)) ||
        ((lora_a_dev || t_dev) && ldb % 8) || !aligned16(a_dev) || !aligned16(w_dev) || !aligned16(lora_a_dev) || !aligned16(lora_b_dev) || !aligned16(t_dev)) {
        ecgb::set_error("ecgb_decode_gemv: M <= 2, K a multiple of 512 (<= 32768), 16-byte aligned operands; t formed here only for K <= 4096");
        return ECGB_ERR_UNSUPPORTED;
    }
    GemvArgs G;
    G.a = (const unsigned short *)a_dev; G.lda_act = lda_act; G.W = (const unsigned short *)w_dev; G.ldw = ldw;
    G.lA = t_dev ? nullptr : (const unsigned short *)lora_a_dev; G.lB = (const unsigned short *)lora_b_dev; G.t_in = (const unsigned short *)t_dev; G.lda = lda; G.ldb = ldb;
    G.y = (unsigned short *)y_dev; G.ldy = ldy; G.M = M; G.K = K; G.N = N; G.n_a = n_a; G.lscale = lora_scale;
    const bool stage = G.lA != nullptr;                                   // (t formed in the kernel: the activations are read once per row of A)
    const size_t lds = (size_t)(kMaxRows * kTRows + (stage ? kMaxRows * K : 0)) * 2;
    hipStream_t st = (hipStream_t)stream;
    if (split) {
        if (stage) hipLaunchKernelGGL((decode_gemv_kernel<4, true>), dim3((unsigned)N), dim3(256), lds, st, G);
        else hipLaunchKernelGGL((decode_gemv_kernel<4, false>), dim3((unsigned)N), dim3(256), lds, st, G);
    } else {
        if (stage) hipLaunchKernelGGL((decode_gemv_kernel<1, true>), dim3((unsigned)((N + 3) / 4)), dim3(256), lds, st, G);
        else hipLaunchKernelGGL((decode_gemv_kernel<1, false>), dim3((unsigned)((N + 3) / 4)), dim3(256), lds, st, G);
    }
    return check(hipGetLastError(), "decode_gemv_kernel launch");
}

// t[M, 64] = bf16(scale * a A^T) for the n_a used rows of A [64, K] (the rest zero), K a multiple of 2048: the wide (down-projection) site of a decode step.
extern "C" int ecgb_decode_lora_t(const void *a_dev, long long lda_act, int M, int K, const void *lora_a_dev, long long lda, int n_a, float lora_scale, void *t_dev, void *stream)
{
    if (!a_dev || !lora_a_dev || !t_dev || M <= 0 || K <= 0 || n_a <= 0) { ecgb::set_error("ecgb_decode_lora_t: bad argument"); return ECGB_ERR_INVALID; }
    if (M > kMaxRows || K % 2048 || n_a > kTRows || lda % 8 || lda_act % 8 || !aligned16(a_dev) || !aligned16(lora_a_dev)) {
        ecgb::set_error("ecgb_decode_lora_t: M <= 2, K a multiple of 2048, 16-byte aligned operands");
        return ECGB_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL(decode_lora_t_kernel, dim3((unsigned)n_a), dim3(256), 0, (hipStream_t)stream, (const unsigned short *)a_dev, lda_act, (const unsigned short *)lora_a_dev, lda, M, K,
                       n_a, lora_scale, (unsigned short *)t_dev);
    return check(hipGetLastError(), "decode_lora_t_kernel launch");
}

// floats of scratch ecgb_decode_attn needs (scores, statistics, partial outputs, tickets); the tickets (the LAST batch * n_kv_heads words) must be zero before the first call
extern "C" size_t ecgb_decode_attn_scratch_floats(long long capacity, int batch, int n_q_heads, int n_kv_heads, int head_dim, int n_splits)
{
    return (size_t)batch * n_q_heads * ((size_t)capacity + (size_t)n_splits * 2 + (size_t)n_splits * head_dim) + (size_t)batch * n_kv_heads;
}

// One decode step's attention for `batch` sequences, from the step's raw q|k|v projection: RoPE on q and the new key, the key / value append at cache row len - 1, softmax(q K^T
// scale + mask) V over the len cached keys -- two launches (scores; values + combine).  len: *kv_len_dev when given (a replayed graph), else kv_len.  head_dim 64 / 128 / 256,
// n_q_heads / n_kv_heads <= 8.  The q and k parts of qkv are NOT rotated in place (the cache and the output are what leaves).
extern "C" int ecgb_decode_attn(void *qkv_dev, long long ld_qkv, const float *cos_dev, const float *sin_dev, void *cache_dev, long long capacity, const float *attn_mask_dev,
                                long long mask_ld, void *o_dev, int batch, int kv_len, const int *kv_len_dev, int n_q_heads, int n_kv_heads, int head_dim, float scale, int n_splits,
                                float *scratch_dev, size_t scratch_floats, void *stream)
{
    if (!qkv_dev || !cos_dev || !sin_dev || !cache_dev || !attn_mask_dev || !o_dev || !scratch_dev || batch <= 0 || n_q_heads <= 0 || n_kv_heads <= 0 || n_splits <= 0) {
        ecgb::set_error("ecgb_decode_attn: bad argument");
        return ECGB_ERR_INVALID;
    }
    const int G = n_q_heads / n_kv_heads;
    if (n_q_heads % n_kv_heads || G > kAttnMaxG || (head_dim != 64 && head_dim != 128 && head_dim != 256) || ld_qkv % 8 || !aligned16(qkv_dev) || !aligned16(cache_dev) ||
        (capacity + n_splits - 1) / n_splits > kAttnMaxChunk || n_splits > 64 || scratch_floats < ecgb_decode_attn_scratch_floats(capacity, batch, n_q_heads, n_kv_heads, head_dim, n_splits)) {
        ecgb::set_error("ecgb_decode_attn: head_dim 64 / 128 / 256, at most 8 query heads a KV head, at most 1024 keys a split, scratch of ecgb_decode_attn_scratch_floats()");
        return ECGB_ERR_UNSUPPORTED;
    }
    AttnArgs2 A;
    A.qkv = (unsigned short *)qkv_dev; A.ld_qkv = ld_qkv; A.cs = cos_dev; A.sn = sin_dev; A.cache = (unsigned short *)cache_dev; A.cap = capacity; A.mask = attn_mask_dev; A.mask_ld = mask_ld;
    A.scores = scratch_dev;
    A.stats = A.scores + (size_t)batch * n_q_heads * capacity;
    A.partial = A.stats + (size_t)batch * n_q_heads * n_splits * 2;
    A.tickets = reinterpret_cast<unsigned *>(A.partial + (size_t)batch * n_q_heads * n_splits * head_dim);
    A.o = (unsigned short *)o_dev; A.len_dev = kv_len_dev; A.len_arg = kv_len; A.Hq = n_q_heads; A.Hkv = n_kv_heads; A.n_splits = n_splits; A.scale = scale;
    const dim3 grid((unsigned)n_splits, (unsigned)n_kv_heads, (unsigned)batch);
    hipStream_t st = (hipStream_t)stream;
#define ECGB_DEC_ATTN(D_)                                                                                                    \
    do {                                                                                                                     \
        hipLaunchKernelGGL(decode_attn_scores_kernel<D_>, grid, dim3(kAttnThreads), 0, st, A);                               \
        hipLaunchKernelGGL(decode_attn_values_kernel<D_>, grid, dim3(kAttnThreads), 0, st, A);                               \
    } while (0)
    if (head_dim == 64) ECGB_DEC_ATTN(64);
    else if (head_dim == 128) ECGB_DEC_ATTN(128);
    else ECGB_DEC_ATTN(256);
#undef ECGB_DEC_ATTN
    return check(hipGetLastError(), "decode_attn kernels launch");
}
