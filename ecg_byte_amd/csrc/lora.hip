// lora.hip -- the LoRA adapter branch of a projection on MI355X (gfx950).
//
// Reference: peft 0.13 LoraLayer as ecg_byte/main.py:131-155 configures it (r 16, alpha 32, dropout 0.05 on
// q,k,v,o,gate,up,down; peft is neither vendored nor installed -- restated from its published formula):
//     y = x W^T + (alpha / r) * B ( A dropout(x) )              one independent dropout mask PER MODULE
// The adapters of a fused projection (q|k|v, gate|up) are stacked: A = [16-row sub-blocks, in], t = [T, 16 * n_sub], and every
// module ("block") draws its own mask from its own 16-bit field of one counter-based hash of the element index -- q, k and v
// see three independent masks of the same x for one hash evaluation.
//
// Both kernels are HBM-bound passes over a [T, in] activation with a rank-16 contraction on the matrix cores; they exist so
// that dropout never costs a pass of its own:
//   lora_down_kernel   t = scale/(1-p) * (mask_b . x) A_b^T   reads x ONCE: the MFMA operand fragments are loaded from global
//                      memory straight in operand layout (lane = row, 8 consecutive k), masked in registers; also stores the
//                      masked x of every block (what the backward's dA = dt^T (mask . x) contraction reads)
//   lora_dx_kernel     dx += scale/(1-p) * sum_b mask_b . (dt_b A_b)   one read-modify-write of dx: the rank-16 products land in
//                      MFMA accumulators whose lane holds four consecutive columns of one row, masked and added in place
#include <hip/hip_runtime.h>

#include <string>

#include "glu_math.hpp"
#include "tokenizer.hpp"

namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) short;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using us4 = __attribute__((ext_vector_type(4))) unsigned short;

__device__ __forceinline__ unsigned short f2bf(float f)
{
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

// 32 hash bits of (seed, element index): two independent 16-bit fields (murmur3's finaliser over a Weyl step)
__device__ __forceinline__ unsigned hash32(unsigned seed, unsigned idx)
{
    unsigned u = idx * 0x9E3779B1u + seed;
    u ^= u >> 15; u *= 0x85EBCA77u;
    u ^= u >> 13; u *= 0xC2B2AE3Du;
    u ^= u >> 16;
    return u;
}

// two fp32 -> packed bf16 pair, one v_cvt_pk_bf16_f32 (round to nearest even)
using bf2_t = __attribute__((ext_vector_type(2))) __bf16;
__device__ __forceinline__ unsigned pack_bf16(float a, float b)
{
    bf2_t v;
    v[0] = (__bf16)a; v[1] = (__bf16)b;
    return __builtin_bit_cast(unsigned, v);
}

constexpr int kLoraK = 64;        // row length of t / dt / A^T: up to four 16-wide sub-blocks, zero beyond the last one (the GEMM K-step)

struct LoraArgs {
    const unsigned short *x;      // [T, in]
    const unsigned short *A;      // lora_down: [>= 16 * n_sub, in];  lora_dx: A^T = [in, 64]
    const unsigned short *dt;     // lora_dx: [T, 64]
    unsigned short *t;            // lora_down: [T, 64]
    unsigned short *xd;           // lora_down: [n_fields, T, in] masked x per block, or NULL
    unsigned short *dx;           // lora_dx: [T, in], += in place
    const unsigned short *gu;     // lora_dx with the GLU backward behind it: gate|up [T, 2 in] of the forward ...
    unsigned short *dgu;          // ... and d(gate|up) [T, 2 in]: dx is then only read
    float *slab;                  // lora_da: [n_chunks, 16 * n_sub, in] fp32 partial products, one slab per chunk of rows
    int chunk_rows;               // lora_da: rows of x per chunk (a multiple of 64)
    int T, in;                    // (sub-block s belongs to block s * NF / NSUB: every block has the same number of sub-blocks)
    float scale;                  // alpha / r / (1 - p)
    unsigned thr;                 // keep iff field >= thr, thr = p * 65536 (0: no dropout)
    unsigned seed;
};

// the masks of one element: bit f = field f keeps it.  A site with ONE module (o, down: NF 1) spends a hash on an element PAIR -- elements 2k and 2k + 1 take the low
// and the high field of hash32(seed, k): half the integer work where the mask is replayed (the backward kernels, and the down-projection's input-gradient GEMM
// whose epilogue regenerates it with one wave per SIMD: gemm_w4.hip EPI 6 / 7 restates this rule).
template <int NF>
__device__ __forceinline__ unsigned keep_bits(const LoraArgs &L, unsigned idx)
{
    if (L.thr == 0) return 0xFu;
    if constexpr (NF == 1) {
        const unsigned h = hash32(L.seed, idx >> 1);
        return ((idx & 1u) ? (h >> 16) : (h & 0xFFFFu)) >= L.thr ? 1u : 0u;
    }
    const unsigned h0 = hash32(L.seed, idx);
    unsigned k = ((h0 & 0xFFFFu) >= L.thr ? 1u : 0u) | ((h0 >> 16) >= L.thr ? 2u : 0u);
    if constexpr (NF > 2) {
        const unsigned h1 = hash32(L.seed ^ 0x68E31DA4u, idx);
        k |= ((h1 & 0xFFFFu) >= L.thr ? 4u : 0u) | ((h1 >> 16) >= L.thr ? 8u : 0u);
    }
    return k;
}

// keep_bits of N consecutive elements from an EVEN index (NF 1: one hash per pair)
template <int NF, int N>
__device__ __forceinline__ void keep_run(const LoraArgs &L, unsigned idx0, unsigned (&keep)[N])
{
    if constexpr (NF == 1) {
#pragma unroll
        for (int e = 0; e < N; e += 2) {
            const unsigned h = L.thr ? hash32(L.seed, (idx0 >> 1) + (unsigned)(e >> 1)) : 0xFFFFFFFFu;
            keep[e] = (h & 0xFFFFu) >= L.thr ? 1u : 0u;
            keep[e + 1] = (h >> 16) >= L.thr ? 1u : 0u;
        }
    } else {
#pragma unroll
        for (int e = 0; e < N; ++e) keep[e] = keep_bits<NF>(L, idx0 + e);
    }
}

// One wave = 16 rows of x over 1 / KW of the columns; a workgroup = 4 waves = 64 / KW rows.  K loop in steps of 32 (one
// v_mfma_f32_16x16x32_bf16 per sub-block).  KW = 4 (in % 256 == 0): the four waves of a workgroup share 16 rows and take a quarter of the
// contraction each, their partial sums meet in LDS -- 32 768 rows are only 2 048 row groups, 8 waves per CU with two 16-byte loads in
// flight each: the pass ran at half the HBM rate.
// RG = 2: a wave takes two row groups (32 rows) against ONE load of the A fragments -- on the fused sites (q|k|v: three sub-blocks) the A
// fragments were three of every four 16-byte loads of the loop (L2 hits, but they are what the load pipe was busy with: 1.8 TB/s on x).
template <int NSUB, int NF, int KW, int RG = 1>
__global__ __launch_bounds__(256) void lora_down_kernel(LoraArgs L)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lm = lane & 15, lq = lane >> 4;
    int row[RG], rowc[RG];
    const unsigned short *xr[RG];
#pragma unroll
    for (int g = 0; g < RG; ++g) {
        row[g] = ((blockIdx.x * (4 / KW) + wave / KW) * RG + g) * 16 + lm;
        rowc[g] = row[g] < L.T ? row[g] : L.T - 1;                    // clamped: out-of-range rows are never stored
        xr[g] = L.x + (size_t)rowc[g] * L.in + 8 * lq;
    }
    f32x4 acc[RG][NSUB];
#pragma unroll
    for (int g = 0; g < RG; ++g)
#pragma unroll
        for (int s = 0; s < NSUB; ++s) acc[g][s] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const size_t plane = (size_t)L.T * L.in;
    const int k_lo = (wave % KW) * (L.in / KW), k_hi = k_lo + L.in / KW;
    for (int k0 = k_lo; k0 < k_hi; k0 += 64) {
        bf16x8 xv[RG][2], av[2][NSUB];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int g = 0; g < RG; ++g) xv[g][u] = *reinterpret_cast<const bf16x8 *>(xr[g] + k0 + 32 * u);
#pragma unroll
            for (int s = 0; s < NSUB; ++s)
                av[u][s] = *reinterpret_cast<const bf16x8 *>(L.A + (size_t)(16 * s + lm) * L.in + k0 + 32 * u + 8 * lq);
        }
#pragma unroll
        for (int g = 0; g < RG; ++g)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                unsigned keep[8];
                const unsigned idx0 = (unsigned)rowc[g] * (unsigned)L.in + (unsigned)(k0 + 32 * u + 8 * lq);
#pragma unroll
                for (int e = 0; e < 8; ++e) keep[e] = 0;
                keep_run<NF, 8>(L, idx0, keep);                               // (in % 64 == 0: idx0 is even)
                bf16x8 xm[NF];
#pragma unroll
                for (int f = 0; f < NF; ++f) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) xm[f][e] = ((keep[e] >> f) & 1u) ? xv[g][u][e] : (short)0;
                    if (L.xd && row[g] < L.T)
                        *reinterpret_cast<bf16x8 *>(L.xd + f * plane + (size_t)row[g] * L.in + k0 + 32 * u + 8 * lq) = xm[f];
                }
#pragma unroll
                for (int s = 0; s < NSUB; ++s)
                    acc[g][s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[u][s], xm[s * NF / NSUB], acc[g][s], 0, 0, 0);   // D'[c][row]: lane = row, regs = 4 columns
            }
    }
    if constexpr (KW > 1) {                                           // wave 0 of the row group adds the other waves' partial sums
        __shared__ f32x4 red[4][RG][NSUB][64];
        if (wave % KW) {
#pragma unroll
            for (int g = 0; g < RG; ++g)
#pragma unroll
                for (int s = 0; s < NSUB; ++s) red[wave][g][s][lane] = acc[g][s];
        }
        __syncthreads();
        if (wave % KW) return;
#pragma unroll
        for (int w = 1; w < KW; ++w)
#pragma unroll
            for (int g = 0; g < RG; ++g)
#pragma unroll
                for (int s = 0; s < NSUB; ++s) acc[g][s] += red[wave + w][g][s][lane];
    }
#pragma unroll
    for (int g = 0; g < RG; ++g) {
        if (row[g] >= L.T) continue;
#pragma unroll
        for (int s = 0; s < 4; ++s) {                                 // t is [T, 64]: the columns past the last sub-block are zero
            us4 v = (us4){0, 0, 0, 0};
            if (s < NSUB) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = f2bf(acc[g][s < NSUB ? s : 0][r] * L.scale);
            }
            *reinterpret_cast<us4 *>(L.t + (size_t)row[g] * kLoraK + 16 * s + 4 * lq) = v;
        }
    }
}

// One wave = 16 rows; it sweeps the columns of dx 64 at a time: four MFMAs per sub-block whose operand rows are permuted so that
// lane (row lm, quarter lq) ends with the 16 CONSECUTIVE columns c0 + 16 lq .. + 15 (MFMA q supplies columns 4q .. 4q + 3 of them):
// two 16-byte loads and stores per lane instead of eight 8-byte ones.  The rank-16 product of a sub-block uses half the K of the
// 16x16x32 MFMA (k-groups 2 and 3 of both fragments are zero).
// GLU != 0 (1 = SiLU, 2 = tanh-GELU; the down-projection site): dx is d(act(gate) * up) and nothing else reads it, so instead of writing
// the sum back the kernel goes on to d gate = dx * up * act'(gate), d up = dx * act(gate) (glu_bwd_kernel's arithmetic on the bf16-rounded sum:
// the same bits as the two separate kernels) and writes those: one read-modify-write pass over [T, in] less per layer.
template <int NSUB, int NF, int GLU = 0>
__global__ __launch_bounds__(256) void lora_dx_kernel(LoraArgs L)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lm = lane & 15, lq = lane >> 4;
    const int row = (blockIdx.x * 4 + wave) * 16 + lm;
    const int rowc = row < L.T ? row : L.T - 1;
    constexpr int kp = kLoraK;
    const bf16x8 zero = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
    bf16x8 dtf[NSUB];
#pragma unroll
    for (int s = 0; s < NSUB; ++s)
        dtf[s] = (lq < 2) ? *reinterpret_cast<const bf16x8 *>(L.dt + (size_t)rowc * kp + 16 * s + 8 * lq) : zero;
    unsigned short *dxr = L.dx + (size_t)rowc * L.in + 16 * lq;
    const int c_lo = blockIdx.y * (L.in / gridDim.y), c_hi = c_lo + L.in / gridDim.y;
    const int acol = 16 * (lm >> 2) + (lm & 3);                      // operand row lm of MFMA q stands for column c0 + acol + 4 q
    // The 64 rows of A^T a step needs are the same for the four waves of the workgroup: one cooperative copy into LDS per step (a thread
    // brings 32 bytes, the next step's rows in flight behind this step's work, one barrier per step) instead of four 16-byte gathers per
    // lane and step -- they were L2 hits, but half of the memory instructions of the loop (hoisting them out was worth 15 %).
    constexpr int kPitch = 72;                                       // shorts per staged row: 144 bytes, so that 16 rows spread over the banks
    __shared__ __attribute__((aligned(16))) unsigned short s_at[2][64 * kPitch];
    const int st_row = threadIdx.x >> 2, st_part = (threadIdx.x & 3) * 16;
    bf16x8 st0, st1;
    auto fetch_at = [&](int c0) {
        const unsigned short *src = L.A + (size_t)(c0 + st_row) * kp + st_part;
        st0 = *reinterpret_cast<const bf16x8 *>(src); st1 = *reinterpret_cast<const bf16x8 *>(src + 8);
    };
    auto store_at = [&](int buf) {
        unsigned short *dst = &s_at[buf][st_row * kPitch + st_part];
        *reinterpret_cast<bf16x8 *>(dst) = st0; *reinterpret_cast<bf16x8 *>(dst + 8) = st1;
    };
    fetch_at(c_lo);
    store_at(0);
    __syncthreads();
    for (int c0 = c_lo, it = 0; c0 < c_hi; c0 += 64, ++it) {
        using us8 = __attribute__((ext_vector_type(8))) unsigned short;
        const bool more = c0 + 64 < c_hi;
        fetch_at(more ? c0 + 64 : c0);                               // unconditional: the waits stay counted
        const unsigned short *at = s_at[it & 1];
        const us8 old0 = *reinterpret_cast<const us8 *>(dxr + c0), old1 = *reinterpret_cast<const us8 *>(dxr + c0 + 8);
        // GLU: gate / up of the same elements, requested here with dx (clamped row: unconditional, so that all six loads are in flight
        // under the hashing and the MFMAs -- behind the `row < T` test below they were issued after them, their latency exposed every step)
        us8 gv[2], uv[2];
        if constexpr (GLU != 0) {
            const size_t oc = (size_t)rowc * 2 * L.in + c0 + 16 * lq;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                gv[half] = *reinterpret_cast<const us8 *>(L.gu + oc + 8 * half);
                uv[half] = *reinterpret_cast<const us8 *>(L.gu + oc + L.in + 8 * half);
            }
        }
        float sum[16];
        unsigned keep[16];
        const unsigned idx0 = (unsigned)rowc * (unsigned)L.in + (unsigned)(c0 + 16 * lq);
#pragma unroll
        for (int e = 0; e < 16; ++e) { keep[e] = 0; sum[e] = 0.f; }
        keep_run<NF, 16>(L, idx0, keep);
#pragma unroll
        for (int s = 0; s < NSUB; ++s)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bf16x8 af = (lq < 2) ? *reinterpret_cast<const bf16x8 *>(at + (acol + 4 * q) * kPitch + 16 * s + 8 * lq) : zero;
                const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, dtf[s], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);   // D'[col][row]
#pragma unroll
                for (int e = 0; e < 4; ++e) sum[4 * q + e] += ((keep[4 * q + e] >> (s * NF / NSUB)) & 1u) ? d[e] : 0.f;
            }
        if (row < L.T) {
            // element pairs stay packed: word j of a half holds columns 2 j, 2 j + 1; one v_cvt_pk_bf16_f32 per pair (round to nearest even,
            // as every other kernel of the library converts) instead of an integer rounding sequence per element
            using u4 = __attribute__((ext_vector_type(4))) unsigned;
            const u4 oldw[2] = {__builtin_bit_cast(u4, old0), __builtin_bit_cast(u4, old1)};
            u4 v[2];
#pragma unroll
            for (int half = 0; half < 2; ++half)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    v[half][j] = pack_bf16(__uint_as_float(oldw[half][j] << 16) + sum[8 * half + 2 * j] * L.scale,
                                           __uint_as_float(oldw[half][j] & 0xFFFF0000u) + sum[8 * half + 2 * j + 1] * L.scale);
            if constexpr (GLU == 0) {
                *reinterpret_cast<u4 *>(dxr + c0) = v[0];
                *reinterpret_cast<u4 *>(dxr + c0 + 8) = v[1];
            } else {
                const size_t o = (size_t)row * 2 * L.in + c0 + 16 * lq;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const u4 g = __builtin_bit_cast(u4, gv[half]), u = __builtin_bit_cast(u4, uv[half]);
                    u4 og, ou;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float dg[2], du[2];
#pragma unroll
                        for (int hl = 0; hl < 2; ++hl) {
                            const float gf = __uint_as_float(hl ? g[j] & 0xFFFF0000u : g[j] << 16), uf = __uint_as_float(hl ? u[j] & 0xFFFF0000u : u[j] << 16);
                            const float df = __uint_as_float(hl ? v[half][j] & 0xFFFF0000u : v[half][j] << 16);
                            dg[hl] = df * uf * ecgb::glu_act_grad<GLU == 2>(gf);
                            du[hl] = df * ecgb::glu_act<GLU == 2>(gf);
                        }
                        og[j] = pack_bf16(dg[0], dg[1]);
                        ou[j] = pack_bf16(du[0], du[1]);
                    }
                    *reinterpret_cast<u4 *>(L.dgu + o + 8 * half) = og;
                    *reinterpret_cast<u4 *>(L.dgu + o + L.in + 8 * half) = ou;
                }
            }
        }
        if (more) store_at((it + 1) & 1);                            // its last readers passed the barrier that ended the previous step
        __syncthreads();
    }
}

// dA_b = scale/(1-p) * dt_b^T . (mask_b . x) for the stacked adapters of a site, from x itself: the masks are the hash lora_down drew,
// evaluated again on the operand fragments, so the forward keeps no masked copy of x per module (1.3 GB a layer at Llama-3.2-1B / batch 32,
// written once and read once) and the contraction reads x ONCE for all modules of the site.
// A workgroup = 256 columns of x over one chunk of rows; K-tiles of 64 rows.  x ([64 rows] x [256 columns], 512-byte rows) and dt ([64] x
// [64], 128-byte rows) go global -> LDS by LDS-DMA as they lie in memory (ring of two buffers, the next tile in flight under the hashing
// of this one); both MFMA operands are "8 consecutive rows of one column", gathered by ds_read_b64_tr_b16 (x: gemm_tn_kernel_tr's layout and
// swizzle; dt: 16-byte chunk c of row r at c ^ 2 f(r), f = bit 1 of r | bit 3 of r << 1, which spreads the 8 rows a 32-lane half addresses over all
// 64 banks).  Wave w owns columns 64 w .. 64 w + 63.  Each chunk's product goes to its own fp32 slab; the slabs are summed in chunk order
// afterwards (ecgb_sum_slabs_bf16): the gradient is the same bits every time.
template <int NSUB, int NF>
__global__ __launch_bounds__(256, 2) void lora_da_kernel(LoraArgs L)
{
    constexpr int kXBytes = 64 * 512, kDtBytes = 64 * 128, kBuf = kXBytes + kDtBytes;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];     // 2 x 40 KB
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int lm = lane & 15, lq = lane >> 4;
    const int col0 = blockIdx.x * 256;
    const int row_lo = blockIdx.y * L.chunk_rows, row_hi = min(L.T, row_lo + L.chunk_rows);
    const int KT = (row_hi - row_lo + 63) / 64;
    const int last_rows = row_hi - (row_lo + (KT - 1) * 64);               // rows of the chunk's last tile that exist (1..64)
    // LDS-DMA: scalar running pointer to the tile + constant per-lane byte offsets (a second set for the one partial tile x can end with:
    // rows past the end re-read the last row; their dt elements are zeroed below)
    unsigned offX[8], offXt[8], offD[2], offDt[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = (wave * 8 + i) * 2 + (lane >> 5);
        const int chunk = (lane & 31) ^ (((r & 7) << 1) ^ (r & 8));
        const long long c = min(col0 + chunk * 8, L.in - 8);                // clamped columns are never stored
        offX[i] = (unsigned)(((long long)r * L.in + c) * 2);
        offXt[i] = (unsigned)(((long long)min(r, last_rows - 1) * L.in + c) * 2);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (wave * 2 + i) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((((r >> 1) & 1) | (((r >> 3) & 1) << 1)) << 1);
        offD[i] = (unsigned)((r * kLoraK + chunk * 8) * 2);
        offDt[i] = (unsigned)((min(r, last_rows - 1) * kLoraK + chunk * 8) * 2);
    }
    const unsigned char *nextX = reinterpret_cast<const unsigned char *>(L.x) + (long long)row_lo * L.in * 2;
    const unsigned char *nextD = reinterpret_cast<const unsigned char *>(L.dt) + (long long)row_lo * kLoraK * 2;
    auto stage = [&](int kt, unsigned char *dst) {
        const bool tail = kt == KT - 1 && last_rows < 64;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(nextX + (tail ? offXt[i] : offX[i])),
                                             (__attribute__((address_space(3))) void *)(dst + (wave * 8 + i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(nextD + (tail ? offDt[i] : offD[i])),
                                             (__attribute__((address_space(3))) void *)(dst + kXBytes + (wave * 2 + i) * 1024), 16, 0, 0);
        nextX += (long long)64 * L.in * 2;
        nextD += 64 * kLoraK * 2;
    };
    stage(0, lds);
    if (KT > 1) { stage(1, lds + kBuf); asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // fragment addresses (see gemm_tn_kernel_tr): lane 16 g + 4 q + p addresses row 8 g + q (+ 4: second read), columns 4 p .. 4 p + 3
    using i2 = __attribute__((ext_vector_type(2))) int;
    using i4 = __attribute__((ext_vector_type(4))) int;
    const int tq = lm >> 2, tp = lm & 3;
    const int rb = 8 * lq + tq;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds;
    unsigned tabX0[4], tabX1[4], tabD[NSUB];
    {
        const int sw = (rb & 7) ^ ((rb & 8) >> 1);
        const unsigned base = lds0 + rb * 512 + (tp & 1) * 8 + (wave >> 1) * 256;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = (wave & 1) * 4 + j;
            tabX0[j] = base + (((2 * (m ^ sw)) + (tp >> 1)) << 4);
            tabX1[j] = base + (((2 * (m ^ 4 ^ sw)) + (tp >> 1)) << 4);
        }
        const int f = ((rb >> 1) & 1) | (((rb >> 3) & 1) << 1);
#pragma unroll
        for (int sb = 0; sb < NSUB; ++sb) tabD[sb] = lds0 + kXBytes + rb * 128 + (tp & 1) * 8 + ((((2 * sb) + (tp >> 1)) ^ (f << 1)) << 4);
    }
    f32x4 acc[NSUB][4];
#pragma unroll
    for (int sb = 0; sb < NSUB; ++sb)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[sb][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int kt = 0; kt < KT; ++kt) {
        // inline asm reads: hipcc would drain the tile in flight (vmcnt 0) in front of an LDS read it can see next to LDS-DMA.  It neither counts asm
        // reads in lgkmcnt nor knows their results are pending (it may copy a destination register as soon as the statement is over): a group of
        // reads and its wait are ONE asm statement, the results exist when it ends
        i4 xv[2][4], dv[2][NSUB];
        auto frags4 = [&](i4 (&f)[4], auto off) {
            constexpr int OFF = decltype(off)::value;
            i2 r0, r1, r2, r3, r4, r5, r6, r7;
            asm volatile("ds_read_b64_tr_b16 %0, %8 offset:%16\n\t"
                         "ds_read_b64_tr_b16 %1, %9 offset:%17\n\t"
                         "ds_read_b64_tr_b16 %2, %10 offset:%16\n\t"
                         "ds_read_b64_tr_b16 %3, %11 offset:%17\n\t"
                         "ds_read_b64_tr_b16 %4, %12 offset:%16\n\t"
                         "ds_read_b64_tr_b16 %5, %13 offset:%17\n\t"
                         "ds_read_b64_tr_b16 %6, %14 offset:%16\n\t"
                         "ds_read_b64_tr_b16 %7, %15 offset:%17\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
                         : "v"(tabX0[0]), "v"(tabX1[0]), "v"(tabX0[1]), "v"(tabX1[1]), "v"(tabX0[2]), "v"(tabX1[2]), "v"(tabX0[3]), "v"(tabX1[3]),
                           "n"(OFF), "n"(OFF + 4 * 512)
                         : "memory");
            f[0] = __builtin_shufflevector(r0, r1, 0, 1, 2, 3); f[1] = __builtin_shufflevector(r2, r3, 0, 1, 2, 3);
            f[2] = __builtin_shufflevector(r4, r5, 0, 1, 2, 3); f[3] = __builtin_shufflevector(r6, r7, 0, 1, 2, 3);
        };
        auto frag1 = [&](unsigned a, auto off) {
            constexpr int OFF = decltype(off)::value;
            i2 lo, hi;
            asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\t"
                         "ds_read_b64_tr_b16 %1, %2 offset:%4\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(lo), "=&v"(hi) : "v"(a), "n"(OFF), "n"(OFF + 4 * 128) : "memory");
            return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
        };
        frags4(xv[0], std::integral_constant<int, 0>{});
        frags4(xv[1], std::integral_constant<int, 32 * 512>{});
#pragma unroll
        for (int sb = 0; sb < NSUB; ++sb) {
            dv[0][sb] = frag1(tabD[sb], std::integral_constant<int, 0>{});
            dv[1][sb] = frag1(tabD[sb], std::integral_constant<int, 32 * 128>{});
        }
        __builtin_amdgcn_s_barrier();                                   // every wave holds its fragments: the buffer is free
        if (kt + 2 < KT) stage(kt + 2, lds + (kt & 1) * kBuf);
        const int trow0 = row_lo + kt * 64;
        if (kt == KT - 1 && last_rows < 64) {                           // (uniform) dt rows past the end contribute nothing
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int sb = 0; sb < NSUB; ++sb)
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const int r = 32 * ks + 8 * lq + 2 * w;
                        unsigned v = (unsigned)dv[ks][sb][w];
                        if (r >= last_rows) v &= 0xFFFF0000u;
                        if (r + 1 >= last_rows) v &= 0x0000FFFFu;
                        dv[ks][sb][w] = (int)v;
                    }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned col = (unsigned)(col0 + wave * 64 + 16 * j + lm);
                const unsigned idx0 = (unsigned)(trow0 + 32 * ks + 8 * lq) * (unsigned)L.in + col;
                i4 xm[NF];
                if (L.thr == 0) {
#pragma unroll
                    for (int f = 0; f < NF; ++f) xm[f] = xv[ks][j];
                } else {
#pragma unroll
                    for (int f = 0; f < NF; ++f) xm[f] = (i4){0, 0, 0, 0};
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const unsigned k0 = keep_bits<NF>(L, idx0 + (unsigned)(2 * w) * (unsigned)L.in);
                        const unsigned k1 = keep_bits<NF>(L, idx0 + (unsigned)(2 * w + 1) * (unsigned)L.in);
#pragma unroll
                        for (int f = 0; f < NF; ++f) {
                            const unsigned m = (((k0 >> f) & 1u) ? 0x0000FFFFu : 0u) | (((k1 >> f) & 1u) ? 0xFFFF0000u : 0u);
                            xm[f][w] = (int)((unsigned)xv[ks][j][w] & m);
                        }
                    }
                }
#pragma unroll
                for (int sb = 0; sb < NSUB; ++sb)
                    acc[sb][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, xm[sb * NF / NSUB]), __builtin_bit_cast(bf16x8, dv[ks][sb]),
                                                                         acc[sb][j], 0, 0, 0);
            }
        if (kt + 1 < KT) {                                              // the next tile has landed (the one after it may still fly)
            if (kt + 2 < KT) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const unsigned step = (kt & 1) ? (unsigned)-kBuf : (unsigned)kBuf;      // (40 KB is not a power of two: add, not xor)
#pragma unroll
            for (int j = 0; j < 4; ++j) { tabX0[j] += step; tabX1[j] += step; }
#pragma unroll
            for (int sb = 0; sb < NSUB; ++sb) tabD[sb] += step;
        }
    }
    // lane (lm, lq) of acc[sb][j]: adapter row 16 sb + lm, columns col0 + 64 wave + 16 j + 4 lq .. + 3
    float *out = L.slab + ((size_t)blockIdx.y * (16 * NSUB)) * L.in;
#pragma unroll
    for (int sb = 0; sb < NSUB; ++sb)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = col0 + wave * 64 + 16 * j + 4 * lq;
            if (c < L.in) *reinterpret_cast<f32x4 *>(out + (size_t)(16 * sb + lm) * L.in + c) = acc[sb][j] * L.scale;
        }
}

int check_common(const char *who, int T, int in, int n_sub, int n_fields, float p)
{
    if (T <= 0 || in <= 0 || in % 64 || n_sub < 1 || n_sub > 4 || n_fields < 1 || n_fields > n_sub || n_sub % n_fields || !(p >= 0.f && p < 1.f)) {
        ecgb::set_error(std::string(who) + ": T > 0, in % 64 == 0, 1..4 sub-blocks split evenly over 1..4 blocks, 0 <= p < 1 required");
        return ECGB_ERR_INVALID;
    }
    return ECGB_OK;
}

void fill(LoraArgs &L, int T, int in, float scale, float p, uint64_t seed)
{
    L.T = T; L.in = in;
    L.thr = (unsigned)(p * 65536.0f);
    L.scale = scale / (1.0f - (float)L.thr / 65536.0f);            // the keep probability the 16-bit threshold really gives
    L.seed = (unsigned)(seed ^ (seed >> 32));
}

#define ECGB_LORA_DISPATCH(KERNEL, GRID, ...)                                                                                  \
    do {                                                                                                                       \
        const int key = n_sub * 8 + n_fields;                                                                                  \
        if (key == 1 * 8 + 1) hipLaunchKernelGGL((KERNEL<1, 1 __VA_ARGS__>), GRID, dim3(256), 0, (hipStream_t)stream, L);      \
        else if (key == 2 * 8 + 1) hipLaunchKernelGGL((KERNEL<2, 1 __VA_ARGS__>), GRID, dim3(256), 0, (hipStream_t)stream, L); \
        else if (key == 2 * 8 + 2) hipLaunchKernelGGL((KERNEL<2, 2 __VA_ARGS__>), GRID, dim3(256), 0, (hipStream_t)stream, L); \
        else if (key == 3 * 8 + 1) hipLaunchKernelGGL((KERNEL<3, 1 __VA_ARGS__>), GRID, dim3(256), 0, (hipStream_t)stream, L); \
        else if (key == 3 * 8 + 3) hipLaunchKernelGGL((KERNEL<3, 3 __VA_ARGS__>), GRID, dim3(256), 0, (hipStream_t)stream, L); \
        else if (key == 4 * 8 + 1) hipLaunchKernelGGL((KERNEL<4, 1 __VA_ARGS__>), GRID, dim3(256), 0, (hipStream_t)stream, L); \
        else if (key == 4 * 8 + 2) hipLaunchKernelGGL((KERNEL<4, 2 __VA_ARGS__>), GRID, dim3(256), 0, (hipStream_t)stream, L); \
        else hipLaunchKernelGGL((KERNEL<4, 4 __VA_ARGS__>), GRID, dim3(256), 0, (hipStream_t)stream, L);                       \
    } while (0)
#define ECGB_COMMA ,

}  // namespace

extern "C" int ecgb_lora_down(const void *x_dev, const void *a_dev, void *t_dev, void *xd_dev, int T, int in, int n_sub, int n_fields,
                              float scale, float p, uint64_t seed, void *stream)
{
    if (!x_dev || !a_dev || !t_dev) { ecgb::set_error("ecgb_lora_down: NULL argument"); return ECGB_ERR_INVALID; }
    if (int rc = check_common("ecgb_lora_down", T, in, n_sub, n_fields, p)) return rc;
    LoraArgs L{};
    L.x = (const unsigned short *)x_dev; L.A = (const unsigned short *)a_dev; L.t = (unsigned short *)t_dev; L.xd = (unsigned short *)xd_dev;
    fill(L, T, in, scale, p, seed);
    if (in % 256 == 0 && T >= 8192 && n_sub > 1) {                   // fused sites: two row groups per wave share the A fragments
        const dim3 grid((unsigned)((T + 31) / 32));
        ECGB_LORA_DISPATCH(lora_down_kernel, grid, ECGB_COMMA 4 ECGB_COMMA 2);
    } else if (in % 256 == 0 && T >= 1024) {                         // row groups of 16 with the contraction split over the four waves
        const dim3 grid((unsigned)((T + 15) / 16));
        ECGB_LORA_DISPATCH(lora_down_kernel, grid, ECGB_COMMA 4);
    } else {
        const dim3 grid((unsigned)((T + 63) / 64));
        ECGB_LORA_DISPATCH(lora_down_kernel, grid, ECGB_COMMA 1);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { ecgb::set_error(std::string("lora_down_kernel: ") + hipGetErrorString(e)); return ECGB_ERR_HIP; }
    return ECGB_OK;
}

extern "C" int ecgb_lora_dx(const void *dt_dev, const void *at_dev, void *dx_dev, int T, int in, int n_sub, int n_fields,
                            float scale, float p, uint64_t seed, void *stream)
{
    if (!dt_dev || !at_dev || !dx_dev) { ecgb::set_error("ecgb_lora_dx: NULL argument"); return ECGB_ERR_INVALID; }
    if (int rc = check_common("ecgb_lora_dx", T, in, n_sub, n_fields, p)) return rc;
    LoraArgs L{};
    L.dt = (const unsigned short *)dt_dev; L.A = (const unsigned short *)at_dev; L.dx = (unsigned short *)dx_dev;
    fill(L, T, in, scale, p, seed);
    const unsigned row_blocks = (unsigned)((T + 63) / 64);
    unsigned col_split = 1;                                          // few rows: split the columns over blockIdx.y to fill the chip
    while (row_blocks * col_split < 4096 && (in / (int)(col_split * 2)) % 64 == 0 && col_split < 16) col_split *= 2;   // 16 workgroups per CU
    const dim3 grid(row_blocks, col_split);
    ECGB_LORA_DISPATCH(lora_dx_kernel, grid);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { ecgb::set_error(std::string("lora_dx_kernel: ") + hipGetErrorString(e)); return ECGB_ERR_HIP; }
    return ECGB_OK;
}

// ecgb_lora_dx followed by the GLU backward in one pass (the down-projection site: dx = d(act(gate) * up) [T, inter]):
//   d(gate|up) = glu_bwd(gate|up, dx + scale/(1-p) * sum_b mask_b . (dt_b A_b));  dx itself is only read.
extern "C" int ecgb_lora_dx_glu(const void *dt_dev, const void *at_dev, const void *dx_dev, const void *gate_up_dev, void *d_gate_up_dev, int T,
                                int inter, int n_sub, int n_fields, float scale, float p, uint64_t seed, int gelu_tanh, void *stream)
{
    if (!dt_dev || !at_dev || !dx_dev || !gate_up_dev || !d_gate_up_dev) { ecgb::set_error("ecgb_lora_dx_glu: NULL argument"); return ECGB_ERR_INVALID; }
    if (int rc = check_common("ecgb_lora_dx_glu", T, inter, n_sub, n_fields, p)) return rc;
    if (n_sub != 1 || n_fields != 1) { ecgb::set_error("ecgb_lora_dx_glu: one adapter block (the down projection)"); return ECGB_ERR_UNSUPPORTED; }
    LoraArgs L{};
    L.dt = (const unsigned short *)dt_dev; L.A = (const unsigned short *)at_dev; L.dx = (unsigned short *)const_cast<void *>(dx_dev);
    L.gu = (const unsigned short *)gate_up_dev; L.dgu = (unsigned short *)d_gate_up_dev;
    fill(L, T, inter, scale, p, seed);
    const unsigned row_blocks = (unsigned)((T + 63) / 64);
    unsigned col_split = 1;
    while (row_blocks * col_split < 4096 && (inter / (int)(col_split * 2)) % 64 == 0 && col_split < 16) col_split *= 2;
    const dim3 grid(row_blocks, col_split);
    if (gelu_tanh) hipLaunchKernelGGL((lora_dx_kernel<1, 1, 2>), grid, dim3(256), 0, (hipStream_t)stream, L);
    else hipLaunchKernelGGL((lora_dx_kernel<1, 1, 1>), grid, dim3(256), 0, (hipStream_t)stream, L);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { ecgb::set_error(std::string("lora_dx_kernel (glu): ") + hipGetErrorString(e)); return ECGB_ERR_HIP; }
    return ECGB_OK;
}

namespace {
// rows of x per chunk: about 512 workgroups (two per CU), never fewer than four K-tiles of 64 rows
int lora_da_chunk_rows(int T, int in)
{
    const int col_tiles = (in + 255) / 256;
    int n_chunks = 512 / col_tiles;                                  // (256 .. 1024 workgroups measured: 512 is best or within 5 % of it at in = 2048 and 8192)
    if (n_chunks < 1) n_chunks = 1;
    int rows = ((T + n_chunks - 1) / n_chunks + 63) / 64 * 64;
    if (rows < 256) rows = 256;
    return rows;
}
}  // namespace

extern "C" int ecgb_sum_slabs_bf16(const float *slabs_dev, long long slab_stride, int n_slabs, void *out_dev, size_t n, int accumulate, void *stream);

extern "C" size_t ecgb_lora_da_scratch_bytes(int T, int in, int n_sub)
{
    if (T <= 0 || in <= 0 || n_sub < 1 || n_sub > 4) return 0;
    const int rows = lora_da_chunk_rows(T, in);
    return (size_t)((T + rows - 1) / rows) * 16 * n_sub * in * sizeof(float);
}

// dA [16 n_sub, in] (bf16, the first rows of the stacked A's gradient) (+)= scale / (1 - p) * dt[:, :16 n_sub]^T . (mask . x)
extern "C" int ecgb_lora_da(const void *x_dev, const void *dt_dev, void *da_dev, int T, int in, int n_sub, int n_fields, float scale, float p,
                            uint64_t seed, int accumulate, void *scratch_dev, size_t scratch_bytes, void *stream)
{
    if (!x_dev || !dt_dev || !da_dev || !scratch_dev) { ecgb::set_error("ecgb_lora_da: NULL argument"); return ECGB_ERR_INVALID; }
    if (int rc = check_common("ecgb_lora_da", T, in, n_sub, n_fields, p)) return rc;
    if (scratch_bytes < ecgb_lora_da_scratch_bytes(T, in, n_sub)) { ecgb::set_error("ecgb_lora_da: scratch smaller than ecgb_lora_da_scratch_bytes()"); return ECGB_ERR_INVALID; }
    LoraArgs L{};
    L.x = (const unsigned short *)x_dev; L.dt = (const unsigned short *)dt_dev; L.slab = (float *)scratch_dev;
    fill(L, T, in, scale, p, seed);
    L.chunk_rows = lora_da_chunk_rows(T, in);
    const int n_chunks = (T + L.chunk_rows - 1) / L.chunk_rows;
    const dim3 grid((unsigned)((in + 255) / 256), (unsigned)n_chunks);
    constexpr unsigned kLds = 2 * (64 * 512 + 64 * 128);
    const int key = n_sub * 8 + n_fields;
    bool launched = false;
#define ECGB_DA_CASE(NS, NFI)                                                                                                    \
    if (key == NS * 8 + NFI) {                                                                                                   \
        const hipError_t ea = hipFuncSetAttribute((const void *)lora_da_kernel<NS, NFI>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds); \
        if (ea != hipSuccess) { ecgb::set_error(std::string("lora_da_kernel: hipFuncSetAttribute: ") + hipGetErrorString(ea)); return ECGB_ERR_HIP; } \
        hipLaunchKernelGGL((lora_da_kernel<NS, NFI>), grid, dim3(256), kLds, (hipStream_t)stream, L);                            \
        launched = true;                                                                                                         \
    }
    ECGB_DA_CASE(1, 1) else ECGB_DA_CASE(2, 1) else ECGB_DA_CASE(2, 2) else ECGB_DA_CASE(3, 1) else ECGB_DA_CASE(3, 3)
    else ECGB_DA_CASE(4, 1) else ECGB_DA_CASE(4, 2) else ECGB_DA_CASE(4, 4)
#undef ECGB_DA_CASE
    // no template case for this (n_sub, n_fields): nothing was launched, and summing the slabs would reduce uninitialised memory into A.grad
    if (!launched) { ecgb::set_error("ecgb_lora_da: unsupported (n_sub, n_fields) combination"); return ECGB_ERR_UNSUPPORTED; }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { ecgb::set_error(std::string("lora_da_kernel: ") + hipGetErrorString(e)); return ECGB_ERR_HIP; }
    return ecgb_sum_slabs_bf16((const float *)scratch_dev, (long long)16 * n_sub * in, n_chunks, da_dev, (size_t)16 * n_sub * in, accumulate, stream);
}
