// gemm.hip -- bf16 GEMM on the CDNA4 matrix cores (gfx950), the projections of the decoder.
//
//   C[M,N] = A[M,K] . B[N,K]^T          (both operands K-contiguous: y = x W^T with W = [out, in],
//                                        the layout of every nn.Linear in modeling_llama.py:227-258,273-395)
// fp32 accumulation in `v_mfma_f32_32x32x16_bf16`, bf16 (or fp32-accumulate) output.
// The two backward products are the same kernel on transposed operands (ecgb_transpose_bf16):
//   dX = dY . W      = NT(dY [M,N], W^T [K,N])        dW = dY^T . X = NT(dY^T [N,M], X^T [K,M]).
//
// Structure: 128x128 output tile per 256-thread workgroup (4 waves as 2x2, 64x64 each = 2x2 MFMA
// tiles of 32x32), K-step 64.  Operand tiles go global -> LDS with `global_load_lds_dwordx4`
// (no VGPR round trip), double buffered, the next K-tile in flight behind a counted vmcnt and raw
// s_barrier while the current one feeds the MFMAs.  LDS rows are 128 B (64 bf16); the 16-byte
// slot index is XOR-swizzled with the row ((row >> 1) & 7) on the SOURCE address side (the LDS
// side of an LDS-DMA is lane-linear), so the 32 rows a ds_read_b128 wave-instruction touches
// land on different banks.  Workgroups are remapped so that consecutive tiles of one C row-block
// share an XCD's L2.
#include <hip/hip_runtime.h>

#include <string>
#include <type_traits>

#include "glu_math.hpp"
#include "tokenizer.hpp"

namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) short;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int BK = 64;                            // K-step; LDS rows are BK bf16 = 128 bytes

struct GemmArgs {
    const unsigned short *A;    // [M, K], row stride lda
    const unsigned short *B;    // [N, K], row stride ldb
    void *C;                    // [M, N], row stride ldc (bf16, or fp32 when accumulate)
    int M, N, K;
    long long lda, ldb, ldc;
    long long batch_a, batch_b, batch_c;   // element strides between batch entries (blockIdx.z)
    // two-level batch addressing (attention heads): z = zo * inner + zi
    int inner;                              // 0 = plain batch strides above
    long long outer_a, inner_a, outer_b, inner_b, outer_c, inner_c;
    int div_a, div_b;
    int tiles_m, tiles_n;
    int group_m;                          // tile order inside an XCD's range: 0 = row-major, else blocks of group_m tile rows walked column by column (see tile_of)
    int accumulate_f32;         // 0: C (bf16) = ..;  1: C is fp32 and C += ..;  2: C is bf16 and C += ..
    float alpha;
    // K-concatenation (NT kernels, batch 1): C = alpha * ([A | A2] . [B | B2]^T) -- after the K / 64 tiles of (A, B) the loop runs
    // K2 / 64 tiles of (A2 [M, K2], B2 [N, K2]).  The LoRA branch t . B_lora^T of a projection rides in the projection's own launch.
    const unsigned short *A2, *B2;
    long long lda2, ldb2;
    int K2;
    // TN kernels with the contraction cut into K-slices: slice s stores its fp32 partial tile at C + s * split_stride (plain stores;
    // the caller sums the slabs in slice order -- fp32 atomics into one buffer cost 0.15-0.2 ms per launch on the chip-wide L2
    // miss path and made the weight gradients differ from run to run)
    long long split_stride = 0;
    // GLU epilogue (gemm_nt_kernel_m16p<.., EPI>): B is the fused gate|up weight [2 * glu_I, K] (gate rows, then up rows); an output
    // tile takes 128 gate rows and the 128 up rows of the SAME columns, interleaved in blocks of 16 so that a lane holds gate[j] and
    // up[j] side by side: H[M, glu_I] = act(gate) * up leaves in the epilogue, gate|up themselves go to C ([M, 2 * glu_I], or skipped
    // when C is null: inference needs only H).
    unsigned short *H = nullptr;
    long long ldh = 0;
    int glu_I = 0;
    // GLU backward in the NN kernel's epilogue (GB != 0): C is d(gate|up) [M, 2 glu_I], the product itself (d of act(gate) * up, [M, glu_I]) is never
    // written; GU = gate|up of the forward, row stride ldgu
    const unsigned short *GU = nullptr;
    long long ldgu = 0;
};

__device__ __forceinline__ unsigned short f2bf_rn(float f)
{
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

// Issue the LDS-DMA loads of one ROWS x 64 operand tile: ROWS/8 wave-instructions of 1 KiB (8 rows each),
// ROWS/8/NW per wave.  LDS image: row r at r*128 B, slot s (16 B) holds logical chunk s ^ ((r >> 1) & 7).
// GLU_I > 0 (run-time value, compile-time switch GLU): tile row r is gate row row0/2 + 16*(r>>5) + (r&15) when (r>>4) is even, else the
// up row GLU_I + the same.
// Tile index -> (tile row, tile column).  The workgroups an XCD runs at one time are ~32 consecutive indices of its range.  Row-major, those are 32 tiles of ONE
// tile row: the A panel is shared by all of them in the XCD's L2, but each reads a B panel of its own -- 1 MB per tile (K 2048) from beyond L2, 8 GB per
// gate|up projection (served by the Infinity Cache, but every byte beyond L2 costs energy, and an MFMA-dense loop is clocked by its power: MI355X_MICROARCH.md,
// DVFS give-back; the guide ranks "streamed data served from L2" first among what raises the clock).  With group_m = 8 the window is 8 tile rows x 4 columns: an A
// slice is shared by 4 workgroups, a B slice by 8 -- 0.375 MB per tile instead of 1.03.  The results do not depend on the order.  MEASURED SLOWER (see
// g_gemm_group_m): the default stays row by row; the switch is kept for A/B.
__device__ __forceinline__ void tile_of(int wgid, int tiles_m, int tiles_n, int group_m, int &tm, int &tn)
{
    if (group_m <= 1) { tm = wgid / tiles_n; tn = wgid % tiles_n; return; }
    const int per_group = group_m * tiles_n;
    const int gid = wgid / per_group, in = wgid % per_group;
    const int first = gid * group_m;
    const int gsz = min(tiles_m - first, group_m);
    tm = first + in % gsz;
    tn = in / gsz;
}

template <int ROWS, int NW, bool GLU = false>
__device__ __forceinline__ void stage_tile(const unsigned short *g, long long ld, int row0, int rows_valid, int k0,
                                           unsigned char *lds_tile, int wave, int lane, int glu_I = 0)
{
    constexpr int PER_WAVE = ROWS / 8 / NW;
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
        const int inst = wave * PER_WAVE + i;
        const int r = inst * 8 + (lane >> 3);          // row inside the tile
        const int slot = lane & 7;
        const int chunk = slot ^ ((r >> 1) & 7);
        int gr = row0 + r;
        if constexpr (GLU) gr = (row0 >> 1) + ((r >> 5) << 4) + (r & 15) + (((r >> 4) & 1) ? glu_I : 0);
        gr = gr < rows_valid ? gr : rows_valid - 1;    // clamp: out-of-range rows are never stored
        const unsigned short *src = g + (long long)gr * ld + k0 + chunk * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(lds_tile + inst * 1024), 16, 0, 0);
    }
}

// BM x BN output tile, WGM x WGN waves, each wave (BM/WGM) x (BN/WGN) = TM x TN MFMA tiles of 32x32.
//   <128,128,2,2>: 4 waves, 64 KiB LDS, 2 workgroups per CU -- small grids and ragged shapes
//   <256,256,2,4>: 8 waves, 128 KiB LDS, 1 workgroup per CU, half the L2->LDS bytes per FLOP -- the big projections
template <int BM, int BN, int WGM, int WGN, bool CAT = false>
__global__ __launch_bounds__(WGM *WGN * 64) void gemm_nt_kernel(GemmArgs G)
{
    constexpr int NW = WGM * WGN;
    constexpr int WTM = BM / WGM, WTN = BN / WGN;        // wave tile
    constexpr int TM = WTM / 32, TN = WTN / 32;          // MFMA tiles per wave
    constexpr int kABytes = BM * BK * 2, kBBytes = BN * BK * 2, kBufBytes = kABytes + kBBytes;
    constexpr int kLoadsPerTile = BM / 8 / NW + BN / 8 / NW;   // LDS-DMA instructions per wave per K-tile
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // [buf][A|B]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // XCD-aware remap: blocks b, b+8, ... share an XCD; give each XCD a contiguous run of tiles
    const int nwg = G.tiles_m * G.tiles_n;
    const int orig = blockIdx.x;
    const int q = nwg / 8, rr = nwg % 8, xcd = orig % 8;
    const int wgid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + orig / 8;
    int tm, tn;
    tile_of(wgid, G.tiles_m, G.tiles_n, G.group_m, tm, tn);
    const int row0 = tm * BM, col0 = tn * BN;
    long long off_a, off_b, off_c;
    if (G.inner) {
        const int zo = blockIdx.z / G.inner, zi = blockIdx.z % G.inner;
        off_a = zo * G.outer_a + (zi / G.div_a) * G.inner_a;
        off_b = zo * G.outer_b + (zi / G.div_b) * G.inner_b;
        off_c = zo * G.outer_c + zi * G.inner_c;
    } else {
        off_a = (long long)blockIdx.z * G.batch_a;
        off_b = (long long)blockIdx.z * G.batch_b;
        off_c = (long long)blockIdx.z * G.batch_c;
    }
    const unsigned short *A = G.A + off_a;
    const unsigned short *B = G.B + off_b;

    const int wr = wave / WGN, wc = wave % WGN;        // wave position in the WGM x WGN grid
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int KT1 = G.K / BK, KT = CAT ? KT1 + G.K2 / BK : KT1;
    // CAT: the K loop continues into (A2, B2).  A template flag, not a run-time test: the plain kernel keeps the exact address
    // arithmetic it had (the selects cost the 256x256 kernel 5 % when they sat in every launch).  Selects, not branches:
    // LDS-DMA issued on two sides of a branch makes hipcc drain vmcnt(0) at the join.
    auto stage_a = [&](int t, unsigned char *dst) {
        if constexpr (CAT) {
            const bool second = t >= KT1;
            stage_tile<BM, NW>(second ? G.A2 : A, second ? G.lda2 : G.lda, row0, G.M, (second ? t - KT1 : t) * BK, dst, wave, lane);
        } else {
            stage_tile<BM, NW>(A, G.lda, row0, G.M, t * BK, dst, wave, lane);
        }
    };
    auto stage_b = [&](int t, unsigned char *dst) {
        if constexpr (CAT) {
            const bool second = t >= KT1;
            stage_tile<BN, NW>(second ? G.B2 : B, second ? G.ldb2 : G.ldb, col0, G.N, (second ? t - KT1 : t) * BK, dst, wave, lane);
        } else {
            stage_tile<BN, NW>(B, G.ldb, col0, G.N, t * BK, dst, wave, lane);
        }
    };
    stage_a(0, lds);
    stage_b(0, lds + kABytes);
    const int lr = lane & 31, lh = lane >> 5;
    for (int kt = 0; kt < KT; ++kt) {
        unsigned char *cur = lds + (kt & 1) * kBufBytes;
        if (kt + 1 < KT) {
            unsigned char *nxt = lds + ((kt + 1) & 1) * kBufBytes;
            stage_a(kt + 1, nxt);
            stage_b(kt + 1, nxt + kABytes);
            // this wave's loads of tile kt have landed once only the kLoadsPerTile just issued are outstanding
            if constexpr (kLoadsPerTile == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if constexpr (kLoadsPerTile == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if constexpr (kLoadsPerTile == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();                            // ... and every other wave's
        const unsigned char *At = cur, *Bt = cur + kABytes;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8 a[TM], b[TN];
            const int chunk = ks * 2 + lh;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int ra = wr * WTM + i * 32 + lr;
                a[i] = *reinterpret_cast<const bf16x8 *>(At + ra * 128 + ((chunk ^ ((ra >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int rb = wc * WTN + j * 32 + lr;
                b[j] = *reinterpret_cast<const bf16x8 *>(Bt + rb * 128 + ((chunk ^ ((rb >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);   // (B.A^T)[n][m]: lane = m
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                            // everyone done reading `cur` before it is restaged
    }

    // epilogue.  The products above are B.A^T, so acc[i][j] holds a TRANSPOSED 32x32 tile: lane lr is output
    // row m = row0 + wr*WTM + i*32 + lr, register reg is output column n = col0 + wc*WTN + j*32 + (reg&3) + 8*(reg>>2) + 4*lh:
    // every group of four registers is four consecutive columns of one row -> one 8-byte (bf16) / 16-byte (fp32) store.
    const float alpha = G.alpha;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = row0 + wr * WTM + i * 32 + lr;
        if (r >= G.M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int c = col0 + wc * WTN + j * 32 + gq * 8 + 4 * lh;
                if (c + 3 < G.N && (G.ldc & 3) == 0) {
                    if (G.accumulate_f32 == 1) {
                        float4 *p = reinterpret_cast<float4 *>(reinterpret_cast<float *>(G.C) + off_c + (long long)r * G.ldc + c);
                        float4 v = *p;
                        v.x += acc[i][j][gq * 4 + 0] * alpha; v.y += acc[i][j][gq * 4 + 1] * alpha;
                        v.z += acc[i][j][gq * 4 + 2] * alpha; v.w += acc[i][j][gq * 4 + 3] * alpha;
                        *p = v;
                    } else {
                        using us4 = __attribute__((ext_vector_type(4))) unsigned short;
                        us4 *p = reinterpret_cast<us4 *>(reinterpret_cast<unsigned short *>(G.C) + off_c + (long long)r * G.ldc + c);
                        us4 v;
                        if (G.accumulate_f32 == 2) {   // bf16 C += alpha * A.B^T
                            const us4 o = *p;
#pragma unroll
                            for (int t = 0; t < 4; ++t) v[t] = f2bf_rn(__uint_as_float((unsigned)o[t] << 16) + acc[i][j][gq * 4 + t] * alpha);
                        } else {
#pragma unroll
                            for (int t = 0; t < 4; ++t) v[t] = f2bf_rn(acc[i][j][gq * 4 + t] * alpha);
                        }
                        *p = v;
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        if (c + t >= G.N) continue;
                        const float v = acc[i][j][gq * 4 + t] * alpha;
                        if (G.accumulate_f32 == 1) reinterpret_cast<float *>(G.C)[off_c + (long long)r * G.ldc + c + t] += v;
                        else {
                            unsigned short *q = reinterpret_cast<unsigned short *>(G.C) + off_c + (long long)r * G.ldc + c + t;
                            *q = f2bf_rn(G.accumulate_f32 == 2 ? __uint_as_float((unsigned)*q << 16) + v : v);
                        }
                    }
                }
            }
    }
}

// Epilogue of the 16x16x32 kernels: acc[i][j] holds D'[n = 4*lq + t][m = lm] of the 16x16 tile (i, j) of this wave.
// `stage` (the 256x256 kernels, whole interior tiles, plain bf16 store): the wave's 128 x 64 block goes through ITS OWN 16 KB of the operand
// buffers (all reads of them are over: every wave has passed its last lgkmcnt(0) + barrier) and leaves as sixteen 16-byte stores per lane, each
// instruction eight whole 128-byte lines -- registers-to-memory it was thirty-two 8-byte stores per lane, each instruction 32-byte pieces of
// sixteen different lines: the in-kernel timers (scripts/dev_prof_gemm.py) put the epilogue at 18 000 of a K = 2048 tile's 107 000 cycles,
// store-issue bound.  Row r of the block lies at r * 128 bytes, its 16-byte chunk c at c ^ ((r >> 1) & 7): the 8-byte writes of a 16-row MFMA
// tile and the row reads both run conflict-free.  Same values, same rounding (round to nearest even) as the direct path.
// GB != 0 (1 = SiLU, 2 = tanh-GELU; the down projection's input gradient in a full fine-tune): the block is d(act(gate) * up); instead of storing it
// the read phase fetches gate and up of the same elements and stores d gate = d * up * act'(gate), d up = d * act(gate) -- ecgb_glu_bwd's arithmetic on
// the bf16-rounded product (the same bits as the two kernels), one write and one read of [M, glu_I] less and no second launch.  Whole tiles only (the
// entry point checks).
template <int TM, int TN, int WTM, int WTN, int GB = 0>
__device__ __forceinline__ void store_tile_m16(const f32x4 (&acc)[TM][TN], const GemmArgs &G, int row0, int col0, int wr, int wc,
                                               int lm, int lq, long long off_c, unsigned char *stage = nullptr)
{
    const float alpha = G.alpha;
    if constexpr (WTM == 128 && WTN == 64) {
        if (GB != 0 || (stage && G.accumulate_f32 == 0 && row0 + 256 <= G.M && col0 + 256 <= G.N && (G.ldc & 7) == 0 && (off_c & 7) == 0)) {   // (uniform)
            using u2 = __attribute__((ext_vector_type(2))) unsigned;
            using u4 = __attribute__((ext_vector_type(4))) unsigned;
            using bf2 = __attribute__((ext_vector_type(2))) __bf16;
            unsigned char *blk = stage + (wr * 4 + wc) * (128 * 128);
            const int sw = (lm >> 1) & 7;                               // (row >> 1) & 7 of every row this lane writes (rows i * 16 + lm)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    bf2 lo, hi;
                    lo[0] = (__bf16)(acc[i][j][0] * alpha); lo[1] = (__bf16)(acc[i][j][1] * alpha);
                    hi[0] = (__bf16)(acc[i][j][2] * alpha); hi[1] = (__bf16)(acc[i][j][3] * alpha);
                    u2 v;
                    v[0] = __builtin_bit_cast(unsigned, lo); v[1] = __builtin_bit_cast(unsigned, hi);
                    const int slot = j * 4 + lq;                        // 8-byte slot of the row
                    *reinterpret_cast<u2 *>(blk + (i * 16 + lm) * 128 + ((((slot >> 1) ^ sw)) << 4) + (slot & 1) * 8) = v;
                }
            const int lane = lq * 16 + lm, rr = lane >> 3, c16 = lane & 7;
            unsigned short *dst = reinterpret_cast<unsigned short *>(G.C) + off_c + (long long)(row0 + wr * WTM + rr) * G.ldc + col0 + wc * WTN + c16 * 8;
            if constexpr (GB != 0) {
                const unsigned short *gsrc = G.GU + (long long)(row0 + wr * WTM + rr) * G.ldgu + col0 + wc * WTN + c16 * 8;
                auto pack = [](float a, float b) { bf2 v; v[0] = (__bf16)a; v[1] = (__bf16)b; return __builtin_bit_cast(unsigned, v); };
#pragma unroll
                for (int grp = 0; grp < 2; ++grp) {                     // eight rows' gate / up loads in flight at a time (the accumulators are dead by now)
                    u4 g[8], u[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const long long ro = (long long)(grp * 8 + k) * 8 * G.ldgu;
                        g[k] = *reinterpret_cast<const u4 *>(gsrc + ro);
                        u[k] = *reinterpret_cast<const u4 *>(gsrc + ro + G.glu_I);
                    }
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int r = (grp * 8 + k) * 8 + rr;
                        const u4 d = *reinterpret_cast<const u4 *>(blk + r * 128 + ((c16 ^ ((r >> 1) & 7)) << 4));
                        u4 og, ou;
#pragma unroll
                        for (int w = 0; w < 4; ++w) {
                            const float g0 = __uint_as_float(g[k][w] << 16), g1 = __uint_as_float(g[k][w] & 0xFFFF0000u);
                            const float u0 = __uint_as_float(u[k][w] << 16), u1 = __uint_as_float(u[k][w] & 0xFFFF0000u);
                            const float d0 = __uint_as_float(d[w] << 16), d1 = __uint_as_float(d[w] & 0xFFFF0000u);
                            og[w] = pack(d0 * u0 * ecgb::glu_act_grad<GB == 2>(g0), d1 * u1 * ecgb::glu_act_grad<GB == 2>(g1));
                            ou[w] = pack(d0 * ecgb::glu_act<GB == 2>(g0), d1 * ecgb::glu_act<GB == 2>(g1));
                        }
                        unsigned short *o = dst + (long long)(grp * 8 + k) * 8 * G.ldc;
                        *reinterpret_cast<u4 *>(o) = og;
                        *reinterpret_cast<u4 *>(o + G.glu_I) = ou;
                    }
                }
                return;
            }
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int r = it * 8 + rr;
                const u4 v = *reinterpret_cast<const u4 *>(blk + r * 128 + ((c16 ^ ((r >> 1) & 7)) << 4));
                *reinterpret_cast<u4 *>(dst + (long long)it * 8 * G.ldc) = v;
            }
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = row0 + wr * WTM + i * 16 + lm;
        if (r >= G.M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int c = col0 + wc * WTN + j * 16 + lq * 4;
            if (c + 3 < G.N && (G.ldc & 3) == 0) {
                if (G.accumulate_f32 == 3) {                      // a K-slice's fp32 partial tile, stored to its own slab (summed in slice order afterwards)
                    float4 v;
                    v.x = acc[i][j][0] * alpha; v.y = acc[i][j][1] * alpha; v.z = acc[i][j][2] * alpha; v.w = acc[i][j][3] * alpha;
                    *reinterpret_cast<float4 *>(reinterpret_cast<float *>(G.C) + off_c + (long long)r * G.ldc + c) = v;
                } else if (G.accumulate_f32 == 1) {
                    float4 *p = reinterpret_cast<float4 *>(reinterpret_cast<float *>(G.C) + off_c + (long long)r * G.ldc + c);
                    float4 v = *p;
                    v.x += acc[i][j][0] * alpha; v.y += acc[i][j][1] * alpha; v.z += acc[i][j][2] * alpha; v.w += acc[i][j][3] * alpha;
                    *p = v;
                } else {
                    using us4 = __attribute__((ext_vector_type(4))) unsigned short;
                    us4 *p = reinterpret_cast<us4 *>(reinterpret_cast<unsigned short *>(G.C) + off_c + (long long)r * G.ldc + c);
                    us4 v;
                    if (G.accumulate_f32 == 2) {
                        const us4 o = *p;
#pragma unroll
                        for (int t = 0; t < 4; ++t) v[t] = f2bf_rn(__uint_as_float((unsigned)o[t] << 16) + acc[i][j][t] * alpha);
                    } else {
#pragma unroll
                        for (int t = 0; t < 4; ++t) v[t] = f2bf_rn(acc[i][j][t] * alpha);
                    }
                    *p = v;
                }
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (c + t >= G.N) continue;
                    const float v = acc[i][j][t] * alpha;
                    if (G.accumulate_f32 == 3) reinterpret_cast<float *>(G.C)[off_c + (long long)r * G.ldc + c + t] = v;
                    else if (G.accumulate_f32 == 1) reinterpret_cast<float *>(G.C)[off_c + (long long)r * G.ldc + c + t] += v;
                    else {
                        unsigned short *q = reinterpret_cast<unsigned short *>(G.C) + off_c + (long long)r * G.ldc + c + t;
                        *q = f2bf_rn(G.accumulate_f32 == 2 ? __uint_as_float((unsigned)*q << 16) + v : v);
                    }
                }
            }
        }
    }
}

// Same structure on `v_mfma_f32_16x16x32_bf16` (one MFMA = 16x16 outputs x K 32; the chip holds a higher
// clock on this shape than on 32x32x16, MI355X_MICROARCH.md "DVFS give-back" item 7).  Operands swapped
// as above: D'[n][m] with m = lane & 15 on the lane and n = 4*(lane >> 4) + reg in the registers.
template <int BM, int BN, int WGM, int WGN, bool CAT = false>
__global__ __launch_bounds__(WGM *WGN * 64) void gemm_nt_kernel_m16(GemmArgs G)
{
    constexpr int NW = WGM * WGN;
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int kABytes = BM * BK * 2, kBBytes = BN * BK * 2, kBufBytes = kABytes + kBBytes;
    constexpr int kLoadsPerTile = BM / 8 / NW + BN / 8 / NW;
    static_assert(kLoadsPerTile == 8, "vmcnt literal below");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nwg = G.tiles_m * G.tiles_n;
    const int orig = blockIdx.x;
    const int q = nwg / 8, rr = nwg % 8, xcd = orig % 8;
    const int wgid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + orig / 8;
    int tm, tn;
    tile_of(wgid, G.tiles_m, G.tiles_n, G.group_m, tm, tn);
    const int row0 = tm * BM, col0 = tn * BN;
    long long off_a, off_b, off_c;
    if (G.inner) {
        const int zo = blockIdx.z / G.inner, zi = blockIdx.z % G.inner;
        off_a = zo * G.outer_a + (zi / G.div_a) * G.inner_a;
        off_b = zo * G.outer_b + (zi / G.div_b) * G.inner_b;
        off_c = zo * G.outer_c + zi * G.inner_c;
    } else {
        off_a = (long long)blockIdx.z * G.batch_a;
        off_b = (long long)blockIdx.z * G.batch_b;
        off_c = (long long)blockIdx.z * G.batch_c;
    }
    const unsigned short *A = G.A + off_a;
    const unsigned short *B = G.B + off_b;
    const int wr = wave / WGN, wc = wave % WGN;
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int KT1 = G.K / BK, KT = CAT ? KT1 + G.K2 / BK : KT1;
    // CAT: the K loop continues into (A2, B2).  A template flag, not a run-time test: the plain kernel keeps the exact address
    // arithmetic it had (the selects cost the 256x256 kernel 5 % when they sat in every launch).  Selects, not branches:
    // LDS-DMA issued on two sides of a branch makes hipcc drain vmcnt(0) at the join.
    auto stage_a = [&](int t, unsigned char *dst) {
        if constexpr (CAT) {
            const bool second = t >= KT1;
            stage_tile<BM, NW>(second ? G.A2 : A, second ? G.lda2 : G.lda, row0, G.M, (second ? t - KT1 : t) * BK, dst, wave, lane);
        } else {
            stage_tile<BM, NW>(A, G.lda, row0, G.M, t * BK, dst, wave, lane);
        }
    };
    auto stage_b = [&](int t, unsigned char *dst) {
        if constexpr (CAT) {
            const bool second = t >= KT1;
            stage_tile<BN, NW>(second ? G.B2 : B, second ? G.ldb2 : G.ldb, col0, G.N, (second ? t - KT1 : t) * BK, dst, wave, lane);
        } else {
            stage_tile<BN, NW>(B, G.ldb, col0, G.N, t * BK, dst, wave, lane);
        }
    };
    stage_a(0, lds);
    stage_b(0, lds + kABytes);
    const int lm = lane & 15, lq = lane >> 4;
    for (int kt = 0; kt < KT; ++kt) {
        unsigned char *cur = lds + (kt & 1) * kBufBytes;
        if (kt + 1 < KT) {
            unsigned char *nxt = lds + ((kt + 1) & 1) * kBufBytes;
            stage_a(kt + 1, nxt);
            stage_b(kt + 1, nxt + kABytes);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        const unsigned char *At = cur, *Bt = cur + kABytes;
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
            bf16x8 a[TM], b[TN];
            const int chunk = ks * 4 + lq;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int ra = wr * WTM + i * 16 + lm;
                a[i] = *reinterpret_cast<const bf16x8 *>(At + ra * 128 + ((chunk ^ ((ra >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int rb = wc * WTN + j * 16 + lm;
                b[j] = *reinterpret_cast<const bf16x8 *>(Bt + rb * 128 + ((chunk ^ ((rb >> 1) & 7)) << 4));
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    store_tile_m16<TM, TN, WTM, WTN>(acc, G, row0, col0, wr, wc, lm, lq, off_c);
}

// The 256x256 tile in four PHASES per K-tile, the two wave rows staggered by one barrier.
// A wave's 128x64 output is cut into quadrants (64 rows x 32 columns); a phase is
//     [LDS reads of the operand halves the quadrant needs | LDS-DMA issue | counted waits]  barrier
//     [16 MFMAs = one quadrant x K 64]                                                        barrier
// and the waves of row 1 run one barrier behind those of row 0, so on every SIMD one wave reads LDS while the
// other one issues MFMAs (cdna_hip_programming.md, "256^2 8-phase template": the per-phase interleave is the lever).
// Quadrant order (A0,B0) (A0,B1) (A1,B1) (A1,B0): phase 1 reads A0 and B0, phase 2 B1, phase 3 A1, phase 4 nothing.
// Reads are retired (lgkmcnt 0) before the phase's first barrier, so an LDS region may be restaged one phase after
// its last read: the B region of the buffer is refilled for tile t+2 in phase 3 of tile t, the A region in phase 4
// -- more than a whole K-tile ahead of their use.  Tile t+1 is waited for in phase 4 of tile t with the eight loads
// of tile t+2 left in flight (vmcnt 8, never 0 inside the loop) and first read one barrier later.
// GLU epilogue of the 16x16x32 kernels (EPI 1 = SiLU, 2 = tanh-GELU): tile (i, 2p) is gate, (i, 2p + 1) up of the same 16 columns.
// The arithmetic is glu_fwd_kernel's on the bf16-rounded projections (glu_math.hpp), so fused and unfused agree bit for bit.
template <int EPI, int TM, int TN, int WTM, int WTN>
__device__ __forceinline__ void store_tile_glu(const f32x4 (&acc)[TM][TN], const GemmArgs &G, int row0, int col0, int wr, int wc, int lm, int lq,
                                               unsigned char *stage = nullptr)
{
    using us4 = __attribute__((ext_vector_type(4))) unsigned short;
    const float alpha = G.alpha;
    unsigned short *C = reinterpret_cast<unsigned short *>(G.C);
    if constexpr (WTM == 128 && WTN == 64) {
        // whole interior tiles: gate, up and act(gate) * up of the wave's 128 rows x 32 columns leave through its own 16 KB of the operand buffers
        // (see store_tile_m16), 64 rows at a time: three arrays of 64 rows x 64 bytes, 16-byte chunk c of row r at c ^ ((r >> 2) & 3); twelve 16-byte
        // stores per lane and half instead of forty-eight 8-byte ones.  Same values, same rounding points as the direct path below.
        if (stage && row0 + 256 <= G.M && col0 + 256 <= G.N && (G.ldc & 7) == 0 && (G.ldh & 7) == 0 && (G.glu_I & 7) == 0) {   // (uniform)
            using u2 = __attribute__((ext_vector_type(2))) unsigned;
            using u4 = __attribute__((ext_vector_type(4))) unsigned;
            using bf2 = __attribute__((ext_vector_type(2))) __bf16;
            auto pack = [](float a, float b) { bf2 v; v[0] = (__bf16)a; v[1] = (__bf16)b; return __builtin_bit_cast(unsigned, v); };
            auto lo_f = [](unsigned w) { return __uint_as_float(w << 16); };
            auto hi_f = [](unsigned w) { return __uint_as_float(w & 0xFFFF0000u); };
            unsigned char *blk = stage + (wr * 4 + wc) * (128 * 128);
            const int sw = (lm >> 2) & 3;
            const int lane = lq * 16 + lm, rr = lane >> 2, c16 = lane & 3;
            const long long hcol = (col0 >> 1) + wc * (WTN / 2) + c16 * 8;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
#pragma unroll
                for (int i4 = 0; i4 < 4; ++i4)
#pragma unroll
                    for (int p = 0; p < TN / 2; ++p) {
                        const int i = half * 4 + i4;
                        u2 g, u, h;
#pragma unroll
                        for (int w = 0; w < 2; ++w) {
                            g[w] = pack(acc[i][2 * p][2 * w] * alpha, acc[i][2 * p][2 * w + 1] * alpha);
                            u[w] = pack(acc[i][2 * p + 1][2 * w] * alpha, acc[i][2 * p + 1][2 * w + 1] * alpha);
                            const unsigned a = pack(ecgb::glu_act<EPI == 2>(lo_f(g[w])), ecgb::glu_act<EPI == 2>(hi_f(g[w])));
                            h[w] = pack(lo_f(a) * lo_f(u[w]), hi_f(a) * hi_f(u[w]));
                        }
                        const int slot = p * 4 + lq;
                        unsigned char *dst = blk + (i4 * 16 + lm) * 64 + (((slot >> 1) ^ sw) << 4) + (slot & 1) * 8;
                        if (C) { *reinterpret_cast<u2 *>(dst) = g; *reinterpret_cast<u2 *>(dst + 4096) = u; }
                        *reinterpret_cast<u2 *>(dst + 8192) = h;
                    }
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int r = it * 16 + rr;
                    const unsigned char *src = blk + r * 64 + ((c16 ^ ((r >> 2) & 3)) << 4);
                    const long long grow = row0 + wr * WTM + half * 64 + r;
                    if (C) {
                        *reinterpret_cast<u4 *>(C + grow * G.ldc + hcol) = *reinterpret_cast<const u4 *>(src);
                        *reinterpret_cast<u4 *>(C + grow * G.ldc + G.glu_I + hcol) = *reinterpret_cast<const u4 *>(src + 4096);
                    }
                    *reinterpret_cast<u4 *>(G.H + grow * G.ldh + hcol) = *reinterpret_cast<const u4 *>(src + 8192);
                }
            }
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = row0 + wr * WTM + i * 16 + lm;
        if (r >= G.M) continue;
#pragma unroll
        for (int p = 0; p < TN / 2; ++p) {
            const int c = (col0 >> 1) + wc * (WTN / 2) + p * 16 + lq * 4;       // column of H = gate column of C
            us4 g, u, h;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                g[t] = f2bf_rn(acc[i][2 * p][t] * alpha);
                u[t] = f2bf_rn(acc[i][2 * p + 1][t] * alpha);
                const float a = __uint_as_float((unsigned)f2bf_rn(ecgb::glu_act<EPI == 2>(__uint_as_float((unsigned)g[t] << 16))) << 16);
                h[t] = f2bf_rn(a * __uint_as_float((unsigned)u[t] << 16));
            }
            if (C) {
                *reinterpret_cast<us4 *>(C + (long long)r * G.ldc + c) = g;
                *reinterpret_cast<us4 *>(C + (long long)r * G.ldc + G.glu_I + c) = u;
            }
            *reinterpret_cast<us4 *>(G.H + (long long)r * G.ldh + c) = h;
        }
    }
}

#ifdef ECGB_PROFILE
__device__ unsigned long long g_gemm_prof[12];    // [prologue, K loop, epilogue, tiles] cycles summed over the workgroups (wave 0) of gemm_nt_kernel_m16p; [4..7]: gemm_nn_kernel_m16p, [8..11]: gemm_tn_kernel_tr (K-tiles instead of tiles in [11])
#endif
template <int BM, int BN, int WGM, int WGN, bool CAT = false, int EPI = 0>
__global__ __launch_bounds__(WGM *WGN * 64) void gemm_nt_kernel_m16p(GemmArgs G)
{
    static_assert(BM == 256 && BN == 256 && WGM == 2 && WGN == 4, "phase schedule written for the 256x256 tile, 2x4 waves");
#ifdef ECGB_PROFILE
    const long long tp0 = clock64();
#endif
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int TM = WTM / 16, TN = WTN / 16;                 // 8 x 4 MFMA tiles per wave
    constexpr int kABytes = BM * BK * 2, kBBytes = BN * BK * 2, kBufBytes = kABytes + kBBytes;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;   // wave index in a scalar register: the LDS-DMA destinations (M0) are scalar arithmetic
    const int nwg = G.tiles_m * G.tiles_n;
    const int orig = blockIdx.x;
    const int q = nwg / 8, rr = nwg % 8, xcd = orig % 8;
    const int wgid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + orig / 8;
    int tm, tn;
    tile_of(wgid, G.tiles_m, G.tiles_n, G.group_m, tm, tn);
    const int row0 = tm * BM, col0 = tn * BN;
    long long off_a, off_b, off_c;
    if (G.inner) {
        const int zo = blockIdx.z / G.inner, zi = blockIdx.z % G.inner;
        off_a = zo * G.outer_a + (zi / G.div_a) * G.inner_a;
        off_b = zo * G.outer_b + (zi / G.div_b) * G.inner_b;
        off_c = zo * G.outer_c + zi * G.inner_c;
    } else {
        off_a = (long long)blockIdx.z * G.batch_a;
        off_b = (long long)blockIdx.z * G.batch_b;
        off_c = (long long)blockIdx.z * G.batch_c;
    }
    const unsigned short *A = G.A + off_a;
    const unsigned short *B = G.B + off_b;
    const int wr = wave / WGN, wc = wave % WGN;
    const int lm = lane & 15, lq = lane >> 4;
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int KT1 = G.K / BK, KT = CAT ? KT1 + G.K2 / BK : KT1;
    // CAT: the K loop continues into (A2, B2).  A template flag, not a run-time test: the plain kernel keeps the exact address
    // arithmetic it had (the selects cost the 256x256 kernel 5 % when they sat in every launch).  Selects, not branches:
    // LDS-DMA issued on two sides of a branch makes hipcc drain vmcnt(0) at the join.
    // The tiles of an operand are staged in K order, so the LDS-DMA address is a wave-uniform running pointer (scalar registers, + 128
    // bytes per tile) plus a per-lane byte offset fixed for the whole launch (row of the tile, swizzled chunk): the loop carries no
    // vector address arithmetic (global_load_lds v_off, s[base]).  CAT: a second pointer / offset set for (A2, B2), chosen per tile by
    // a wave-uniform select (no branch: LDS-DMA issued on two sides of a branch makes hipcc drain vmcnt(0) at the join).
    unsigned offA[4], offB[4], offA2[CAT ? 4 : 1], offB2[CAT ? 4 : 1];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        const int ra = min(row0 + r, G.M - 1) - row0;                                  // clamped: out-of-range rows are never stored
        int rb = min(col0 + r, G.N - 1) - col0;
        if constexpr (EPI != 0) rb = ((r >> 5) << 4) + (r & 15) + (((r >> 4) & 1) ? G.glu_I : 0);   // gate / up rows interleaved by 16
        offA[i] = (unsigned)(((long long)ra * G.lda + chunk * 8) * 2);
        offB[i] = (unsigned)(((long long)rb * G.ldb + chunk * 8) * 2);
        if constexpr (CAT) {
            offA2[i] = (unsigned)(((long long)ra * G.lda2 + chunk * 8) * 2);
            offB2[i] = (unsigned)(((long long)rb * G.ldb2 + chunk * 8) * 2);
        }
    }
    const int brow0 = EPI != 0 ? (col0 >> 1) : col0;
    const unsigned char *nextA = reinterpret_cast<const unsigned char *>(A + (long long)row0 * G.lda);
    const unsigned char *nextB = reinterpret_cast<const unsigned char *>(B + (long long)brow0 * G.ldb);
    const unsigned char *nextA2 = CAT ? reinterpret_cast<const unsigned char *>(G.A2 + (long long)row0 * G.lda2) : nullptr;
    const unsigned char *nextB2 = CAT ? reinterpret_cast<const unsigned char *>(G.B2 + (long long)brow0 * G.ldb2) : nullptr;
    auto stage_a = [&](int t, unsigned char *dst) {
        const bool second = CAT && t >= KT1;
        const unsigned char *base = second ? nextA2 : nextA;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + (second ? offA2[CAT ? i : 0] : offA[i])),
                                             (__attribute__((address_space(3))) void *)(dst + (wave * 4 + i) * 1024), 16, 0, 0);
        if (second) nextA2 += BK * 2; else nextA += BK * 2;
    };
    auto stage_b = [&](int t, unsigned char *dst) {
        const bool second = CAT && t >= KT1;
        const unsigned char *base = second ? nextB2 : nextB;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + (second ? offB2[CAT ? i : 0] : offB[i])),
                                             (__attribute__((address_space(3))) void *)(dst + (wave * 4 + i) * 1024), 16, 0, 0);
        if (second) nextB2 += BK * 2; else nextB += BK * 2;
    };

    // prologue: tiles 0 and 1 in flight, tile 0 complete
    stage_b(0, lds + kABytes);
    stage_a(0, lds);
    if (KT > 1) {
        stage_b(1, lds + kBufBytes + kABytes);
        stage_a(1, lds + kBufBytes);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();          // the stagger: row 1 runs one barrier behind row 0

#ifdef ECGB_PROFILE
    const long long tp1 = clock64();
#endif
    bf16x8 a[2][4], b[2][4];                             // [k-step][tile]: one A half (4 row tiles), both B halves (4 column tiles)
    for (int kt = 0; kt < KT; ++kt) {
        const unsigned char *At = lds + (kt & 1) * kBufBytes, *Bt = At + kABytes;
        unsigned char *nxt = lds + (kt & 1) * kBufBytes;  // tile kt+2 goes where tile kt lives
        const bool more = kt + 2 < KT;
        auto read_a = [&](int half) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ra = wr * WTM + (half * 4 + i) * 16 + lm, chunk = ks * 4 + lq;
                    a[ks][i] = *reinterpret_cast<const bf16x8 *>(At + ra * 128 + ((chunk ^ ((ra >> 1) & 7)) << 4));
                }
        };
        auto read_b = [&](int half) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int rb = wc * WTN + (half * 2 + j) * 16 + lm, chunk = ks * 4 + lq;
                    b[ks][half * 2 + j] = *reinterpret_cast<const bf16x8 *>(Bt + rb * 128 + ((chunk ^ ((rb >> 1) & 7)) << 4));
                }
        };
        auto mfma_quadrant = [&](int ah, int bh) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[ah * 4 + i][bh * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[ks][bh * 2 + j], a[ks][i], acc[ah * 4 + i][bh * 2 + j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_s_barrier();
        };
        // phase 1
        read_b(0);
        read_a(0);
        mfma_quadrant(0, 0);
        // phase 2
        read_b(1);
        mfma_quadrant(0, 1);
        // phase 3: the B region of this buffer was last read in phase 2
        read_a(1);
        if (more) stage_b(kt + 2, nxt + kABytes);
        mfma_quadrant(1, 1);
        // phase 4: the A region was last read in phase 3; tile kt+1 must be complete one barrier from here
        if (more) {
            stage_a(kt + 2, nxt);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        mfma_quadrant(1, 0);
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();          // barrier counts of the two rows match again
#ifdef ECGB_PROFILE
    const long long tp2 = clock64();
#endif
    if constexpr (EPI != 0) store_tile_glu<EPI, TM, TN, WTM, WTN>(acc, G, row0, col0, wr, wc, lm, lq, lds);
    else store_tile_m16<TM, TN, WTM, WTN>(acc, G, row0, col0, wr, wc, lm, lq, off_c, lds);
#ifdef ECGB_PROFILE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long tp3 = clock64();
    if (tid == 0) {
        atomicAdd(&g_gemm_prof[0], (unsigned long long)(tp1 - tp0)); atomicAdd(&g_gemm_prof[1], (unsigned long long)(tp2 - tp1));
        atomicAdd(&g_gemm_prof[2], (unsigned long long)(tp3 - tp2)); atomicAdd(&g_gemm_prof[3], 1ull);
    }
#endif
}

// Epilogue of the TN kernels: bf16 store, or fp32 atomics when the contraction is split over workgroups.
template <int TM, int TN, int WTM, int WTN>
__device__ __forceinline__ void store_tile_tn(const f32x4 (&acc)[TM][TN], const GemmArgs &G, int row0, int col0, int wr, int wc, int lm, int lq, int split)
{
    const float alpha = G.alpha;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = row0 + wr * WTM + i * 16 + lm;
        if (r >= G.N) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int c = col0 + wc * WTN + j * 16 + lq * 4;
            if (c + 3 >= G.K) continue;
            if (G.accumulate_f32) {   // split over the contraction: this slice's fp32 partial tile goes to its own slab
                float *p = reinterpret_cast<float *>(G.C) + (long long)split * G.split_stride + (long long)r * G.ldc + c;
                f32x4 v;
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = acc[i][j][t] * alpha;
                *reinterpret_cast<f32x4 *>(p) = v;
            } else {
                using us4 = __attribute__((ext_vector_type(4))) unsigned short;
                us4 v;
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = f2bf_rn(acc[i][j][t] * alpha);
                *reinterpret_cast<us4 *>(reinterpret_cast<unsigned short *>(G.C) + (long long)r * G.ldc + c) = v;
            }
        }
    }
}

// ---- persistent form of gemm_nt_kernel_m16p -------------------------------------------------------------------------------------------
// One workgroup per CU walks output tiles blockIdx.x, blockIdx.x + gridDim.x, ... (same tile -> XCD deal as the one-tile kernel).  A tile of the
// K = 2048 projections is 83 000 cycles of K loop between 4 800 of prologue (address set-up + the flight of the first K-tile) and 4 800 of
// epilogue, and with one workgroup per CU nothing overlaps them (scripts/dev_prof_gemm.py).  Here the K-tile stream runs on into the next
// tile: its first K-tile is staged during the last-but-one K-tile of this one (the ring position that would take K-tile KT), lands under the
// last K-tile and the epilogue, and the second one is issued right behind the epilogue; the epilogue stages through the OTHER buffer
// (8 KB per wave: half a block at a time).  Whole interior tiles only, no batch, plain bf16 store or the GLU epilogue (the launcher checks).
template <int EPI, int TM, int TN, int GB = 0>
__device__ __forceinline__ void store_block_half_staged(const f32x4 (&acc)[TM][TN], const GemmArgs &G, unsigned char *blk, int row0, int col0, int wr, int wc, int lm, int lq)
{
    using u2 = __attribute__((ext_vector_type(2))) unsigned;
    using u4 = __attribute__((ext_vector_type(4))) unsigned;
    using bf2 = __attribute__((ext_vector_type(2))) __bf16;
    auto pack = [](float a, float b) { bf2 v; v[0] = (__bf16)a; v[1] = (__bf16)b; return __builtin_bit_cast(unsigned, v); };
    const float alpha = G.alpha;
    const int lane = lq * 16 + lm;
    if constexpr (EPI == 0) {                                           // two passes of 64 rows x 128 bytes (layout of store_tile_m16)
        const int sw = (lm >> 1) & 7, rr = lane >> 3, c16 = lane & 7;
        unsigned short *dst = reinterpret_cast<unsigned short *>(G.C) + (long long)(row0 + wr * 128 + rr) * G.ldc + col0 + wc * 64 + c16 * 8;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int i = half * 4 + i4;
                    u2 v;
                    v[0] = pack(acc[i][j][0] * alpha, acc[i][j][1] * alpha); v[1] = pack(acc[i][j][2] * alpha, acc[i][j][3] * alpha);
                    const int slot = j * 4 + lq;
                    *reinterpret_cast<u2 *>(blk + (i4 * 16 + lm) * 128 + ((((slot >> 1) ^ sw)) << 4) + (slot & 1) * 8) = v;
                }
            if constexpr (GB != 0) {                                    // GLU backward on the staged product (see store_tile_m16): gate / up of the 64 rows in flight together
                const unsigned short *gsrc = G.GU + (long long)(row0 + wr * 128 + half * 64 + rr) * G.ldgu + col0 + wc * 64 + c16 * 8;
                u4 g[8], u[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    g[k] = *reinterpret_cast<const u4 *>(gsrc + (long long)k * 8 * G.ldgu);
                    u[k] = *reinterpret_cast<const u4 *>(gsrc + (long long)k * 8 * G.ldgu + G.glu_I);
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int r = k * 8 + rr;
                    const u4 d = *reinterpret_cast<const u4 *>(blk + r * 128 + ((c16 ^ ((r >> 1) & 7)) << 4));
                    u4 og, ou;
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const float g0 = __uint_as_float(g[k][w] << 16), g1 = __uint_as_float(g[k][w] & 0xFFFF0000u);
                        const float u0 = __uint_as_float(u[k][w] << 16), u1 = __uint_as_float(u[k][w] & 0xFFFF0000u);
                        const float d0 = __uint_as_float(d[w] << 16), d1 = __uint_as_float(d[w] & 0xFFFF0000u);
                        og[w] = pack(d0 * u0 * ecgb::glu_act_grad<GB == 2>(g0), d1 * u1 * ecgb::glu_act_grad<GB == 2>(g1));
                        ou[w] = pack(d0 * ecgb::glu_act<GB == 2>(g0), d1 * ecgb::glu_act<GB == 2>(g1));
                    }
                    unsigned short *o = dst + (long long)(half * 64 + k * 8) * G.ldc;
                    *reinterpret_cast<u4 *>(o) = og;
                    *reinterpret_cast<u4 *>(o + G.glu_I) = ou;
                }
            } else {
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int r = it * 8 + rr;
                    const u4 v = *reinterpret_cast<const u4 *>(blk + r * 128 + ((c16 ^ ((r >> 1) & 7)) << 4));
                    *reinterpret_cast<u4 *>(dst + (long long)(half * 64 + it * 8) * G.ldc) = v;
                }
            }
        }
    } else {                                                            // four passes of 32 rows: gate, up, act(gate) * up, 64 bytes a row each (layout of store_tile_glu)
        auto lo_f = [](unsigned w) { return __uint_as_float(w << 16); };
        auto hi_f = [](unsigned w) { return __uint_as_float(w & 0xFFFF0000u); };
        unsigned short *C = reinterpret_cast<unsigned short *>(G.C);
        const int sw = (lm >> 2) & 3, rr = lane >> 2, c16 = lane & 3;
        const long long hcol = (col0 >> 1) + wc * 32 + c16 * 8;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
                for (int p = 0; p < TN / 2; ++p) {
                    const int i = q4 * 2 + i2;
                    u2 g, u, h;
#pragma unroll
                    for (int w = 0; w < 2; ++w) {
                        g[w] = pack(acc[i][2 * p][2 * w] * alpha, acc[i][2 * p][2 * w + 1] * alpha);
                        u[w] = pack(acc[i][2 * p + 1][2 * w] * alpha, acc[i][2 * p + 1][2 * w + 1] * alpha);
                        const unsigned a = pack(ecgb::glu_act<EPI == 2>(lo_f(g[w])), ecgb::glu_act<EPI == 2>(hi_f(g[w])));
                        h[w] = pack(lo_f(a) * lo_f(u[w]), hi_f(a) * hi_f(u[w]));
                    }
                    const int slot = p * 4 + lq;
                    unsigned char *d = blk + (i2 * 16 + lm) * 64 + (((slot >> 1) ^ sw) << 4) + (slot & 1) * 8;
                    if (C) { *reinterpret_cast<u2 *>(d) = g; *reinterpret_cast<u2 *>(d + 2048) = u; }
                    *reinterpret_cast<u2 *>(d + 4096) = h;
                }
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int r = it * 16 + rr;
                const unsigned char *src = blk + r * 64 + ((c16 ^ ((r >> 2) & 3)) << 4);
                const long long grow = row0 + wr * 128 + q4 * 32 + r;
                if (C) {
                    *reinterpret_cast<u4 *>(C + grow * G.ldc + hcol) = *reinterpret_cast<const u4 *>(src);
                    *reinterpret_cast<u4 *>(C + grow * G.ldc + G.glu_I + hcol) = *reinterpret_cast<const u4 *>(src + 2048);
                }
                *reinterpret_cast<u4 *>(G.H + grow * G.ldh + hcol) = *reinterpret_cast<const u4 *>(src + 4096);
            }
        }
    }
}

template <bool CAT, int EPI>
__global__ __launch_bounds__(512) void gemm_nt_kernel_m16pp(GemmArgs G)
{
    constexpr int BM = 256, BN = 256, WGN = 4, WTM = 128, WTN = 64, TM = 8, TN = 4;
    constexpr int kABytes = BM * BK * 2, kBBytes = BN * BK * 2, kBufBytes = kABytes + kBBytes;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int nwg = G.tiles_m * G.tiles_n;
    const int q = nwg / 8, rr8 = nwg % 8;
    auto tile_origin = [&](int t, int &row0, int &col0) {              // the one-tile kernel's blockIdx.x -> tile map (XCD t % 8 owns a contiguous range)
        const int xcd = t % 8;
        const int wgid = (xcd < rr8 ? xcd * (q + 1) : rr8 * (q + 1) + (xcd - rr8) * q) + t / 8;
        int tm_, tn_;
        tile_of(wgid, G.tiles_m, G.tiles_n, G.group_m, tm_, tn_);
        row0 = tm_ * BM; col0 = tn_ * BN;
    };
    const int wr = wave / WGN, wc = wave % WGN;
    const int lm = lane & 15, lq = lane >> 4;
    const int KT1 = G.K / BK, KT = CAT ? KT1 + G.K2 / BK : KT1;
    unsigned offA[4], offB[4], offA2[CAT ? 4 : 1], offB2[CAT ? 4 : 1];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        int rb = r;
        if constexpr (EPI != 0) rb = ((r >> 5) << 4) + (r & 15) + (((r >> 4) & 1) ? G.glu_I : 0);   // gate / up rows interleaved by 16
        offA[i] = (unsigned)(((long long)r * G.lda + chunk * 8) * 2);
        offB[i] = (unsigned)(((long long)rb * G.ldb + chunk * 8) * 2);
        if constexpr (CAT) {
            offA2[i] = (unsigned)(((long long)r * G.lda2 + chunk * 8) * 2);
            offB2[i] = (unsigned)(((long long)rb * G.ldb2 + chunk * 8) * 2);
        }
    }
    const unsigned char *nextA = nullptr, *nextB = nullptr, *nextA2 = nullptr, *nextB2 = nullptr;
    auto point_a = [&](int row0) {
        nextA = reinterpret_cast<const unsigned char *>(G.A + (long long)row0 * G.lda);
        if constexpr (CAT) nextA2 = reinterpret_cast<const unsigned char *>(G.A2 + (long long)row0 * G.lda2);
    };
    auto point_b = [&](int col0) {
        const int brow0 = EPI != 0 ? (col0 >> 1) : col0;
        nextB = reinterpret_cast<const unsigned char *>(G.B + (long long)brow0 * G.ldb);
        if constexpr (CAT) nextB2 = reinterpret_cast<const unsigned char *>(G.B2 + (long long)brow0 * G.ldb2);
    };
    auto stage_a = [&](int t, unsigned char *dst) {
        const bool second = CAT && t >= KT1;
        const unsigned char *base = second ? nextA2 : nextA;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + (second ? offA2[CAT ? i : 0] : offA[i])),
                                             (__attribute__((address_space(3))) void *)(dst + (wave * 4 + i) * 1024), 16, 0, 0);
        if (second) nextA2 += BK * 2; else nextA += BK * 2;
    };
    auto stage_b = [&](int t, unsigned char *dst) {
        const bool second = CAT && t >= KT1;
        const unsigned char *base = second ? nextB2 : nextB;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + (second ? offB2[CAT ? i : 0] : offB[i])),
                                             (__attribute__((address_space(3))) void *)(dst + (wave * 4 + i) * 1024), 16, 0, 0);
        if (second) nextB2 += BK * 2; else nextB += BK * 2;
    };

    int tile = blockIdx.x, row0, col0;
    tile_origin(tile, row0, col0);
    point_a(row0); point_b(col0);
    int par = 0;                                                        // K-tile kt of the current tile lives in buffer (kt + par) & 1
    stage_b(0, lds + kABytes);
    stage_a(0, lds);
    stage_b(1, lds + kBufBytes + kABytes);                              // (KT >= 2: the launcher checks)
    stage_a(1, lds + kBufBytes);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (;;) {
        const int ntile = tile + gridDim.x;
        const bool has_next = ntile < nwg;
        int nrow0 = 0, ncol0 = 0;
        if (has_next) tile_origin(ntile, nrow0, ncol0);
        f32x4 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (wr == 1) __builtin_amdgcn_s_barrier();          // the stagger: row 1 runs one barrier behind row 0
        bf16x8 a[2][4], b[2][4];
        for (int kt = 0; kt < KT; ++kt) {
            const int cur = (kt + par) & 1;
            const unsigned char *At = lds + cur * kBufBytes, *Bt = At + kABytes;
            unsigned char *nxt = lds + cur * kBufBytes;      // K-tile kt + 2 of the stream goes where K-tile kt lives
            const bool own = kt + 2 < KT;                    // the stream's next K-tile belongs to this output tile ...
            const bool more = own || (kt + 2 == KT && has_next);   // ... or is the first one of the next (its second follows the epilogue)
            const int tnext = own ? kt + 2 : 0;
            auto read_a = [&](int half) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int ra = wr * WTM + (half * 4 + i) * 16 + lm, chunk = ks * 4 + lq;
                        a[ks][i] = *reinterpret_cast<const bf16x8 *>(At + ra * 128 + ((chunk ^ ((ra >> 1) & 7)) << 4));
                    }
            };
            auto read_b = [&](int half) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int rb = wc * WTN + (half * 2 + j) * 16 + lm, chunk = ks * 4 + lq;
                        b[ks][half * 2 + j] = *reinterpret_cast<const bf16x8 *>(Bt + rb * 128 + ((chunk ^ ((rb >> 1) & 7)) << 4));
                    }
            };
            auto mfma_quadrant = [&](int ah, int bh) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[ah * 4 + i][bh * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[ks][bh * 2 + j], a[ks][i], acc[ah * 4 + i][bh * 2 + j], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_s_barrier();
            };
            read_b(0);
            read_a(0);
            mfma_quadrant(0, 0);
            read_b(1);
            mfma_quadrant(0, 1);
            read_a(1);
            if (more) {
                if (!own) point_b(ncol0);
                stage_b(tnext, nxt + kABytes);
            }
            mfma_quadrant(1, 1);
            if (more) {
                if (!own) point_a(nrow0);
                stage_a(tnext, nxt);
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            mfma_quadrant(1, 0);
        }
        if (wr == 0) __builtin_amdgcn_s_barrier();          // barrier counts of the two rows match again; every read of the operand buffers is over
        // the buffer of this tile's last K-tile is free (the other one holds or awaits the next tile's first K-tile)
        store_block_half_staged<EPI, TM, TN>(acc, G, lds + ((KT - 1 + par) & 1) * kBufBytes + wave * 8192, row0, col0, wr, wc, lm, lq);
        if (!has_next) break;
        par = (KT + par) & 1;                                // the next tile's K-tile 0 sits where this tile's K-tile KT - 2 was
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // every wave is through with the staging buffer: it takes the next tile's second K-tile
        stage_b(1, lds + ((1 + par) & 1) * kBufBytes + kABytes);
        stage_a(1, lds + ((1 + par) & 1) * kBufBytes);
        tile = ntile; row0 = nrow0; col0 = ncol0;
    }
}

// ---- TN product for weight gradients: C[N,K] = A^T . B with A = dY [M,N] and B = X [M,K] (both ROW-major, the
// contraction index m is the slow one).  Same MFMA / LDS-image / epilogue scheme as gemm_nt_kernel_m16; only the
// staging differs: a thread loads 4 consecutive m-rows x 8 columns (four 16-byte loads, issued one K-tile ahead
// into registers) and writes them transposed -- eight 8-byte LDS writes, each 4 consecutive m of one output row --
// so the LDS image is again [output row][64 contraction elements] and no transposed copy ever exists in HBM.
template <int BN_, int BK_, int WGM, int WGN>
__global__ __launch_bounds__(WGM *WGN * 64) void gemm_tn_kernel_m16(GemmArgs G)
{
    // naming: output is [N rows (from A's columns)] x [K cols (from B's columns)], contraction over M in steps of 64
    constexpr int NW = WGM * WGN;
    constexpr int WTM = BN_ / WGM, WTN = BK_ / WGN;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int kABytes = BN_ * 64 * 2, kBBytes = BK_ * 64 * 2, kBufBytes = kABytes + kBBytes;
    static_assert(BN_ == 256 && BK_ == 256 && NW == 8, "one staging item per thread per operand");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nwg = G.tiles_m * G.tiles_n;
    const int orig = blockIdx.x;
    const int q = nwg / 8, rr = nwg % 8, xcd = orig % 8;
    const int wgid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + orig / 8;
    int tm, tn;
    tile_of(wgid, G.tiles_m, G.tiles_n, G.group_m, tm, tn);
    const int row0 = tm * BN_, col0 = tn * BK_;          // output row block (columns of A), output column block (columns of B)
    const unsigned short *A = G.A, *B = G.B;
    // staging item of this thread: columns c8*8 .. +7 of the operand's 256-column block, rows mg*4 .. +3 of the 64-row K-tile
    // 32 column chunks x 16 row groups = 512 items.  The 16 row groups of a column chunk sit on consecutive lanes and a wave
    // covers 4 column chunks, so that one transposing LDS write (below) touches 4 output rows x all 16 positions of a row
    // instead of 32 rows x 2 positions: with the chunk-major assignment every write was a 16-way bank conflict
    // (SQ_LDS_BANK_CONFLICT = 74 % of the LDS cycles).
    const int c8 = wave * 4 + (lane >> 4), mg = lane & 15;
    const int wr = wave / WGN, wc = wave % WGN;
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int KT_all = G.M / 64;
    const int per = (KT_all + (int)gridDim.y - 1) / (int)gridDim.y;
    const int kt_begin = (int)blockIdx.y * per;
    const int KT = min(KT_all, kt_begin + per);
    if (kt_begin >= KT) {                                 // an empty K-slice still owns a slab: zeros
        if (G.accumulate_f32) store_tile_tn<TM, TN, WTM, WTN>(acc, G, row0, col0, wr, wc, lane & 15, lane >> 4, (int)blockIdx.y);
        return;
    }
    bf16x8 ra[4], rb[4];
    auto load_items = [&](int kt) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const long long m = (long long)kt * 64 + mg * 4 + t;
            const int ca = min(row0 + c8 * 8, G.N - 8), cb = min(col0 + c8 * 8, G.K - 8);   // clamped columns are never stored
            ra[t] = *reinterpret_cast<const bf16x8 *>(A + m * G.lda + ca);
            rb[t] = *reinterpret_cast<const bf16x8 *>(B + m * G.ldb + cb);
        }
    };
    auto write_items = [&](unsigned char *buf) {
        // Element pair (2k, 2k+1) of the four loaded rows is one 32-bit register each; a byte permute gathers the low or
        // the high halves into the 8-byte transposed word.  Odd column chunks write their odd row first: a write
        // instruction then covers two even and two odd output rows, i.e. both halves of the 64 LDS banks.
        using u4 = __attribute__((ext_vector_type(4))) unsigned;
        using u2 = __attribute__((ext_vector_type(2))) unsigned;
        const unsigned odd = (unsigned)c8 & 1u;
        const unsigned sel0 = odd ? 0x07060302u : 0x05040100u, sel1 = odd ? 0x05040100u : 0x07060302u;
        u4 a4[4], b4[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { a4[t] = __builtin_bit_cast(u4, ra[t]); b4[t] = __builtin_bit_cast(u4, rb[t]); }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                const unsigned sel = pass ? sel1 : sel0;
                const int r = c8 * 8 + 2 * k + (int)(pass ? 1u - odd : odd);   // output row inside the block = operand column
                const int slot = (mg >> 1) ^ ((r >> 1) & 7);                    // 16-byte slot of contraction elements 8*(mg>>1) .. +7
                u2 wa, wb;
                wa[0] = __builtin_amdgcn_perm(a4[1][k], a4[0][k], sel); wa[1] = __builtin_amdgcn_perm(a4[3][k], a4[2][k], sel);
                wb[0] = __builtin_amdgcn_perm(b4[1][k], b4[0][k], sel); wb[1] = __builtin_amdgcn_perm(b4[3][k], b4[2][k], sel);
                *reinterpret_cast<u2 *>(buf + r * 128 + (slot << 4) + (mg & 1) * 8) = wa;
                *reinterpret_cast<u2 *>(buf + kABytes + r * 128 + (slot << 4) + (mg & 1) * 8) = wb;
            }
    };
    load_items(kt_begin);
    write_items(lds + (kt_begin & 1) * kBufBytes);
    const int lm = lane & 15, lq = lane >> 4;
    for (int kt = kt_begin; kt < KT; ++kt) {
        unsigned char *cur = lds + (kt & 1) * kBufBytes;
        if (kt + 1 < KT) load_items(kt + 1);              // in flight behind this tile's MFMAs
        __syncthreads();                                  // tile kt visible to every wave
        const unsigned char *At = cur, *Bt = cur + kABytes;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a[TM], b[TN];
            const int chunk = ks * 4 + lq;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int r = wr * WTM + i * 16 + lm;
                a[i] = *reinterpret_cast<const bf16x8 *>(At + r * 128 + ((chunk ^ ((r >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int r = wc * WTN + j * 16 + lm;
                b[j] = *reinterpret_cast<const bf16x8 *>(Bt + r * 128 + ((chunk ^ ((r >> 1) & 7)) << 4));
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
        if (kt + 1 < KT) write_items(lds + ((kt + 1) & 1) * kBufBytes);   // the other buffer: its last reader finished before the barrier above
    }
    store_tile_tn<TM, TN, WTM, WTN>(acc, G, row0, col0, wr, wc, lm, lq, (int)blockIdx.y);
}

// C[N, K] = alpha * A[M, N]^T . B[M, K] with NO register staging: both operand tiles ([64 contraction rows] x [256
// columns], 512-byte rows) go global -> LDS by LDS-DMA exactly as they lie in memory, and the MFMA fragments (8
// consecutive contraction elements of one column) are gathered by gfx950's transposing LDS read `ds_read_b64_tr_b16`
// (per 16 lanes: a block of 4 rows x 16 columns delivered column-major; lane 4q+p supplies the address of row q,
// columns 4p..4p+3) -- two reads per fragment.  16-byte chunk c of row r sits at chunk c ^ 2(r & 7) ^ (r & 8) of its row (applied
// on the global source address, the LDS side of the DMA is lane-linear), which spreads the 8 rows a 32-lane half reads
// over all 64 banks.  Schedule: the four phases and the stagger of gemm_nt_kernel_m16p.
template <int BN_, int BK_, int WGM, int WGN>
__global__ __launch_bounds__(WGM *WGN * 64) void gemm_tn_kernel_tr(GemmArgs G)
{
#ifdef ECGB_PROFILE
    const long long tp0 = clock64();
#endif
    const int orig = blockIdx.x, split = blockIdx.y, n_splits = gridDim.y;
    static_assert(BN_ == 256 && BK_ == 256 && WGM == 2 && WGN == 4, "phase schedule written for the 256x256 tile, 2x4 waves");
    constexpr int WTM = BN_ / WGM, WTN = BK_ / WGN;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int kABytes = 64 * BN_ * 2, kBBytes = 64 * BK_ * 2, kBufBytes = kABytes + kBBytes;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int nwg = G.tiles_m * G.tiles_n;
    const int q_ = nwg / 8, rr = nwg % 8, xcd = orig % 8;
    const int wgid = (xcd < rr ? xcd * (q_ + 1) : rr * (q_ + 1) + (xcd - rr) * q_) + orig / 8;
    int tm, tn;
    tile_of(wgid, G.tiles_m, G.tiles_n, G.group_m, tm, tn);
    const int row0 = tm * BN_, col0 = tn * BK_;          // output row block (columns of A), output column block (columns of B)
    const int wr = wave / WGN, wc = wave % WGN;
    const int lm = lane & 15, lq = lane >> 4;
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int KT_all = G.M / 64;
    const int per = (KT_all + n_splits - 1) / n_splits;
    const int kt_begin = split * per;
    const int KT = min(KT_all, kt_begin + per);
    if (kt_begin >= KT) {                                 // an empty K-slice still owns a slab: zeros
        if (G.accumulate_f32) store_tile_tn<TM, TN, WTM, WTN>(acc, G, row0, col0, wr, wc, lm, lq, split);
        return;
    }

    // LDS-DMA of one operand tile: 32 instructions of 1 KiB (2 rows of 512 B), 4 per wave.  Address = wave-uniform running pointer to
    // the next K-tile of the operand (tiles are staged in order: one scalar add per tile) + a per-lane 32-bit byte offset that never
    // changes (row inside the tile, swizzled chunk): no multiplies in the loop.
    unsigned offA[4], offB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int inst = wave * 4 + i;
        const int r = inst * 2 + (lane >> 5);                               // contraction row inside the tile
        const int chunk = (lane & 31) ^ (((r & 7) << 1) ^ (r & 8));         // the global chunk this LDS slot holds
        offA[i] = (unsigned)(((long long)r * G.lda + min(row0 + chunk * 8, G.N - 8)) * 2);     // clamped columns are never stored
        offB[i] = (unsigned)(((long long)r * G.ldb + min(col0 + chunk * 8, G.K - 8)) * 2);
    }
    const unsigned char *nextA = reinterpret_cast<const unsigned char *>(G.A) + (long long)kt_begin * 64 * G.lda * 2;
    const unsigned char *nextB = reinterpret_cast<const unsigned char *>(G.B) + (long long)kt_begin * 64 * G.ldb * 2;
    auto stage = [&](const unsigned char *&next, long long ld, const unsigned (&off)[4], unsigned char *dst) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(next + off[i]),
                                             (__attribute__((address_space(3))) void *)(dst + (wave * 4 + i) * 1024), 16, 0, 0);
        next += 64 * ld * 2;
    };
    stage(nextB, G.ldb, offB, lds + kABytes);
    stage(nextA, G.lda, offA, lds);
    if (kt_begin + 1 < KT) {
        stage(nextB, G.ldb, offB, lds + kBufBytes + kABytes);
        stage(nextA, G.lda, offA, lds + kBufBytes);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();

    // Fragment addresses.  A fragment (8 contraction rows x 16 columns) is two transposing reads of 4 rows each; lane (16g + 4q + p)
    // addresses row 8g + q (+4 for the second read), columns 4p..4p+3 of the fragment's 16: 16-byte chunk 2m + (p >> 1) of the row
    // for the fragment at columns 16m.., stored at chunk ^ 2s with s = (row & 7) ^ ((row & 8) >> 1) -- a per-lane constant that only
    // moves the low three bits of m.  So the byte address is  table[m & 7] + compile-time offset, with ONE table of 8 per-lane
    // entries for the A operand (the second read, 4 rows down, has s ^ 4: entry m ^ 4) and 4 + 4 for B; K-half, second read, operand
    // and nothing else go into the instruction's offset field, the buffer toggle (64 KiB: past the 16-bit field) is an XOR on the
    // 16 entries once per K-tile.  The loop had ~190 address VALU + ~100 register moves per 64 MFMAs before; the moves came from
    // assembling the 4-register operand out of two 2-register results element by element.
    using i2 = __attribute__((ext_vector_type(2))) int;
    using i4 = __attribute__((ext_vector_type(4))) int;
    const int tq = lm >> 2, tp = lm & 3;                                    // this lane's row / column group inside a 4 x 16 block
    unsigned tabA[8], tabB0[4], tabB1[4];
    {
        const int rb = 8 * lq + tq;
        const int sw = (rb & 7) ^ ((rb & 8) >> 1);
        const unsigned lane_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds + rb * 512 + (tp & 1) * 8;
        const int cb = tp >> 1;
#pragma unroll
        for (int m = 0; m < 8; ++m) tabA[m] = lane_base + wr * 256 + (((2 * (m ^ sw)) + cb) << 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = (wc & 1) * 4 + j;
            tabB0[j] = lane_base + (wc >> 1) * 256 + (((2 * (m ^ sw)) + cb) << 4);
            tabB1[j] = lane_base + (wc >> 1) * 256 + (((2 * (m ^ 4 ^ sw)) + cb) << 4);
        }
    }
#ifdef ECGB_PROFILE
    const long long tp1 = clock64();
#endif
    bf16x8 a[2][4], b[2][4];
    for (int it = 0, kt = kt_begin; kt < KT; ++kt, ++it) {
        unsigned char *nxt = lds + (it & 1) * kBufBytes;
        const bool more = kt + 2 < KT;
        // inline asm: next to LDS-DMA hipcc puts a vmcnt(0) in front of a transposing read it can see (it would drain the two tiles
        // in flight once per K-tile); ordering is by the explicit counted waits and barriers of the schedule
        auto frag = [&](unsigned t0, unsigned t1, auto off) {
            constexpr int OFF = decltype(off)::value;
            i2 lo, hi;
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(t0), "n"(OFF));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(t1), "n"(OFF + 2048));
            const i4 f = __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
            return __builtin_bit_cast(bf16x8, f);
        };
        auto read_a = [&](int half) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[0][i] = frag(tabA[half * 4 + i], tabA[(half * 4 + i) ^ 4], std::integral_constant<int, 0>{});
                a[1][i] = frag(tabA[half * 4 + i], tabA[(half * 4 + i) ^ 4], std::integral_constant<int, 32 * 512>{});
            }
        };
        auto read_b = [&](int half) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                b[0][half * 2 + j] = frag(tabB0[half * 2 + j], tabB1[half * 2 + j], std::integral_constant<int, kABytes>{});
                b[1][half * 2 + j] = frag(tabB0[half * 2 + j], tabB1[half * 2 + j], std::integral_constant<int, kABytes + 32 * 512>{});
            }
        };
        auto mfma_quadrant = [&](int ah, int bh) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[ah * 4 + i][bh * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[ks][bh * 2 + j], a[ks][i], acc[ah * 4 + i][bh * 2 + j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_s_barrier();
        };
        read_b(0);
        read_a(0);
        mfma_quadrant(0, 0);
        read_b(1);
        mfma_quadrant(0, 1);
        read_a(1);
        if (more) stage(nextB, G.ldb, offB, nxt + kABytes);
        mfma_quadrant(1, 1);
        if (more) {
            stage(nextA, G.lda, offA, nxt);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        mfma_quadrant(1, 0);
#pragma unroll
        for (int m = 0; m < 8; ++m) tabA[m] ^= (unsigned)kBufBytes;
#pragma unroll
        for (int j = 0; j < 4; ++j) { tabB0[j] ^= (unsigned)kBufBytes; tabB1[j] ^= (unsigned)kBufBytes; }
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();
#ifdef ECGB_PROFILE
    const long long tp2 = clock64();
#endif
    store_tile_tn<TM, TN, WTM, WTN>(acc, G, row0, col0, wr, wc, lm, lq, split);
#ifdef ECGB_PROFILE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long tp3 = clock64();
    if (threadIdx.x == 0) {
        atomicAdd(&g_gemm_prof[8], (unsigned long long)(tp1 - tp0)); atomicAdd(&g_gemm_prof[9], (unsigned long long)(tp2 - tp1));
        atomicAdd(&g_gemm_prof[10], (unsigned long long)(tp3 - tp2)); atomicAdd(&g_gemm_prof[11], (unsigned long long)(KT - kt_begin));
    }
#endif
}


// ---- NN product for input gradients: C[M, N] = alpha * A[M, K] . B[K, N] with B ROW-major (dX = dY . W with W = [out, in] as nn.Linear
// stores it): no transposed shadow copy of the weight.  The A side is gemm_nt_kernel_m16p's (rows of 64 contraction elements, LDS image
// [256 rows][128 B], ds_read_b128 fragments), the B side is gemm_tn_kernel_tr's ([64 contraction rows] x [256 columns] as it lies in memory,
// fragments by ds_read_b64_tr_b16 from the per-lane address tables), on the same four-phase staggered schedule.  Lane for lane the MFMAs see
// the operands of the NT kernel run on a transposed copy of B, so the results are the same bits (tests).
template <int BM, int BN, int WGM, int WGN, int GB = 0>
__global__ __launch_bounds__(WGM *WGN * 64) void gemm_nn_kernel_m16p(GemmArgs G)
{
    static_assert(BM == 256 && BN == 256 && WGM == 2 && WGN == 4, "phase schedule written for the 256x256 tile, 2x4 waves");
#ifdef ECGB_PROFILE
    const long long tp0 = clock64();
#endif
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int kABytes = BM * BK * 2, kBBytes = BN * BK * 2, kBufBytes = kABytes + kBBytes;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int nwg = G.tiles_m * G.tiles_n;
    const int orig = blockIdx.x;
    const int q = nwg / 8, rr = nwg % 8, xcd = orig % 8;
    const int wgid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + orig / 8;
    int tm, tn;
    tile_of(wgid, G.tiles_m, G.tiles_n, G.group_m, tm, tn);
    const int row0 = tm * BM, col0 = tn * BN;
    const int wr = wave / WGN, wc = wave % WGN;
    const int lm = lane & 15, lq = lane >> 4;
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // gridDim.y > 1: the contraction is cut into K-slices (few output tiles, a long contraction: the loss head's input gradient), each workgroup stores its
    // fp32 partial tile to its own slab (accumulate_f32 == 3), ecgb_sum_slabs_bf16 adds them in slice order
    const int KT_all = G.K / BK, n_splits = gridDim.y, split = blockIdx.y;
    const int kt_per = (KT_all + n_splits - 1) / n_splits, kt0 = split * kt_per;
    const int KT = min(KT_all, kt0 + kt_per) - kt0;
    if (KT <= 0) {                                         // an empty K-slice still owns a slab: zeros
        store_tile_m16<TM, TN, WTM, WTN>(acc, G, row0, col0, wr, wc, lm, lq, (long long)split * G.split_stride);
        return;
    }
    // LDS-DMA addresses: scalar running pointers + constant per-lane byte offsets (see gemm_nt_kernel_m16p / gemm_tn_kernel_tr)
    unsigned offA[4], offB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + (lane >> 3);                                // A: tile row, 8 rows of 128 B per instruction
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        const int ra = min(row0 + r, G.M - 1) - row0;
        offA[i] = (unsigned)(((long long)ra * G.lda + chunk * 8) * 2);
        const int rk = (wave * 4 + i) * 2 + (lane >> 5);                               // B: contraction row inside the tile, 2 rows of 512 B
        const int chunkb = (lane & 31) ^ (((rk & 7) << 1) ^ (rk & 8));
        offB[i] = (unsigned)(((long long)rk * G.ldb + min(col0 + chunkb * 8, G.N - 8)) * 2);      // clamped columns are never stored
    }
    const unsigned char *nextA = reinterpret_cast<const unsigned char *>(G.A + (long long)row0 * G.lda + (long long)kt0 * BK);
    const unsigned char *nextB = reinterpret_cast<const unsigned char *>(G.B + (long long)kt0 * BK * G.ldb);
    auto stage_a = [&](unsigned char *dst) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(nextA + offA[i]),
                                             (__attribute__((address_space(3))) void *)(dst + (wave * 4 + i) * 1024), 16, 0, 0);
        nextA += BK * 2;
    };
    auto stage_b = [&](unsigned char *dst) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(nextB + offB[i]),
                                             (__attribute__((address_space(3))) void *)(dst + (wave * 4 + i) * 1024), 16, 0, 0);
        nextB += (long long)BK * G.ldb * 2;
    };
    stage_b(lds + kABytes);
    stage_a(lds);
    if (KT > 1) {
        stage_b(lds + kBufBytes + kABytes);
        stage_a(lds + kBufBytes);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();

    // B fragment addresses: the tables of gemm_tn_kernel_tr (entry m & 7 of the fragment at columns 16 m.., + offset field)
    using i2 = __attribute__((ext_vector_type(2))) int;
    using i4 = __attribute__((ext_vector_type(4))) int;
    const int tq = lm >> 2, tp = lm & 3;
    unsigned tabB0[4], tabB1[4];
    {
        const int rb = 8 * lq + tq;
        const int sw = (rb & 7) ^ ((rb & 8) >> 1);
        const unsigned lane_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds + rb * 512 + (tp & 1) * 8;
        const int cb = tp >> 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = (wc & 1) * 4 + j;
            tabB0[j] = lane_base + (wc >> 1) * 256 + (((2 * (m ^ sw)) + cb) << 4);
            tabB1[j] = lane_base + (wc >> 1) * 256 + (((2 * (m ^ 4 ^ sw)) + cb) << 4);
        }
    }
#ifdef ECGB_PROFILE
    const long long tp1 = clock64();
#endif
    bf16x8 a[2][4], b[2][4];
    for (int kt = 0; kt < KT; ++kt) {
        const unsigned char *At = lds + (kt & 1) * kBufBytes;
        unsigned char *nxt = lds + (kt & 1) * kBufBytes;
        const bool more = kt + 2 < KT;
        auto read_a = [&](int half) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ra = wr * WTM + (half * 4 + i) * 16 + lm, chunk = ks * 4 + lq;
                    a[ks][i] = *reinterpret_cast<const bf16x8 *>(At + ra * 128 + ((chunk ^ ((ra >> 1) & 7)) << 4));
                }
        };
        auto fragb = [&](unsigned t0, unsigned t1, auto off) {
            constexpr int OFF = decltype(off)::value;
            i2 lo, hi;
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(t0), "n"(OFF));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(t1), "n"(OFF + 2048));
            const i4 f = __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
            return __builtin_bit_cast(bf16x8, f);
        };
        auto read_b = [&](int half) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                b[0][half * 2 + j] = fragb(tabB0[half * 2 + j], tabB1[half * 2 + j], std::integral_constant<int, kABytes>{});
                b[1][half * 2 + j] = fragb(tabB0[half * 2 + j], tabB1[half * 2 + j], std::integral_constant<int, kABytes + 32 * 512>{});
            }
        };
        auto mfma_quadrant = [&](int ah, int bh) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[ah * 4 + i][bh * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[ks][bh * 2 + j], a[ks][i], acc[ah * 4 + i][bh * 2 + j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_s_barrier();
        };
        read_b(0);
        read_a(0);
        mfma_quadrant(0, 0);
        read_b(1);
        mfma_quadrant(0, 1);
        read_a(1);
        if (more) stage_b(nxt + kABytes);
        mfma_quadrant(1, 1);
        if (more) {
            stage_a(nxt);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        mfma_quadrant(1, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) { tabB0[j] ^= (unsigned)kBufBytes; tabB1[j] ^= (unsigned)kBufBytes; }
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();
#ifdef ECGB_PROFILE
    const long long tp2 = clock64();
#endif
    store_tile_m16<TM, TN, WTM, WTN, GB>(acc, G, row0, col0, wr, wc, lm, lq, (long long)split * G.split_stride, lds);
#ifdef ECGB_PROFILE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long tp3 = clock64();
    if (threadIdx.x == 0) {
        atomicAdd(&g_gemm_prof[4], (unsigned long long)(tp1 - tp0)); atomicAdd(&g_gemm_prof[5], (unsigned long long)(tp2 - tp1));
        atomicAdd(&g_gemm_prof[6], (unsigned long long)(tp3 - tp2)); atomicAdd(&g_gemm_prof[7], 1ull);
    }
#endif
}

// Persistent form of gemm_nn_kernel_m16p (see gemm_nt_kernel_m16pp): whole interior tiles only, plain bf16 store or the GLU backward (GB).
// The B-side fragment tables toggle buffers once per K-tile, so they follow the K-tile stream across tiles by themselves.
template <int GB>
__global__ __launch_bounds__(512) void gemm_nn_kernel_m16pp(GemmArgs G)
{
    constexpr int BM = 256, BN = 256, WGN = 4, WTM = 128, TM = 8, TN = 4;
    constexpr int kABytes = BM * BK * 2, kBBytes = BN * BK * 2, kBufBytes = kABytes + kBBytes;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int nwg = G.tiles_m * G.tiles_n;
    const int q = nwg / 8, rr8 = nwg % 8;
    auto tile_origin = [&](int t, int &row0, int &col0) {
        const int xcd = t % 8;
        const int wgid = (xcd < rr8 ? xcd * (q + 1) : rr8 * (q + 1) + (xcd - rr8) * q) + t / 8;
        int tm_, tn_;
        tile_of(wgid, G.tiles_m, G.tiles_n, G.group_m, tm_, tn_);
        row0 = tm_ * BM; col0 = tn_ * BN;
    };
    const int wr = wave / WGN, wc = wave % WGN;
    const int lm = lane & 15, lq = lane >> 4;
    const int KT = G.K / BK;
    unsigned offA[4], offB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        offA[i] = (unsigned)(((long long)r * G.lda + chunk * 8) * 2);
        const int rk = (wave * 4 + i) * 2 + (lane >> 5);
        const int chunkb = (lane & 31) ^ (((rk & 7) << 1) ^ (rk & 8));
        offB[i] = (unsigned)(((long long)rk * G.ldb + chunkb * 8) * 2);              // the tile's first column goes into the base pointer
    }
    const unsigned char *nextA = nullptr, *nextB = nullptr;
    auto stage_a = [&](unsigned char *dst) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(nextA + offA[i]),
                                             (__attribute__((address_space(3))) void *)(dst + (wave * 4 + i) * 1024), 16, 0, 0);
        nextA += BK * 2;
    };
    auto stage_b = [&](unsigned char *dst) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(nextB + offB[i]),
                                             (__attribute__((address_space(3))) void *)(dst + (wave * 4 + i) * 1024), 16, 0, 0);
        nextB += (long long)BK * G.ldb * 2;
    };
    int tile = blockIdx.x, row0, col0;
    tile_origin(tile, row0, col0);
    nextA = reinterpret_cast<const unsigned char *>(G.A + (long long)row0 * G.lda);
    nextB = reinterpret_cast<const unsigned char *>(G.B + col0);
    int par = 0;
    stage_b(lds + kABytes);
    stage_a(lds);
    stage_b(lds + kBufBytes + kABytes);                                 // (KT >= 2: the launcher checks)
    stage_a(lds + kBufBytes);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    using i2 = __attribute__((ext_vector_type(2))) int;
    using i4 = __attribute__((ext_vector_type(4))) int;
    const int tq = lm >> 2, tp = lm & 3;
    unsigned tabB0[4], tabB1[4];
    {
        const int rb = 8 * lq + tq;
        const int sw = (rb & 7) ^ ((rb & 8) >> 1);
        const unsigned lane_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds + rb * 512 + (tp & 1) * 8;
        const int cb = tp >> 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = (wc & 1) * 4 + j;
            tabB0[j] = lane_base + (wc >> 1) * 256 + (((2 * (m ^ sw)) + cb) << 4);
            tabB1[j] = lane_base + (wc >> 1) * 256 + (((2 * (m ^ 4 ^ sw)) + cb) << 4);
        }
    }
    for (;;) {
        const int ntile = tile + gridDim.x;
        const bool has_next = ntile < nwg;
        int nrow0 = 0, ncol0 = 0;
        if (has_next) tile_origin(ntile, nrow0, ncol0);
        f32x4 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (wr == 1) __builtin_amdgcn_s_barrier();
        bf16x8 a[2][4], b[2][4];
        for (int kt = 0; kt < KT; ++kt) {
            const int cur = (kt + par) & 1;
            const unsigned char *At = lds + cur * kBufBytes;
            unsigned char *nxt = lds + cur * kBufBytes;
            const bool own = kt + 2 < KT;
            const bool more = own || (kt + 2 == KT && has_next);
            auto read_a = [&](int half) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int ra = wr * WTM + (half * 4 + i) * 16 + lm, chunk = ks * 4 + lq;
                        a[ks][i] = *reinterpret_cast<const bf16x8 *>(At + ra * 128 + ((chunk ^ ((ra >> 1) & 7)) << 4));
                    }
            };
            auto fragb = [&](unsigned t0, unsigned t1, auto off) {
                constexpr int OFF = decltype(off)::value;
                i2 lo, hi;
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(t0), "n"(OFF));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(t1), "n"(OFF + 2048));
                const i4 f = __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
                return __builtin_bit_cast(bf16x8, f);
            };
            auto read_b = [&](int half) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    b[0][half * 2 + j] = fragb(tabB0[half * 2 + j], tabB1[half * 2 + j], std::integral_constant<int, kABytes>{});
                    b[1][half * 2 + j] = fragb(tabB0[half * 2 + j], tabB1[half * 2 + j], std::integral_constant<int, kABytes + 32 * 512>{});
                }
            };
            auto mfma_quadrant = [&](int ah, int bh) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[ah * 4 + i][bh * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[ks][bh * 2 + j], a[ks][i], acc[ah * 4 + i][bh * 2 + j], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_s_barrier();
            };
            read_b(0);
            read_a(0);
            mfma_quadrant(0, 0);
            read_b(1);
            mfma_quadrant(0, 1);
            read_a(1);
            if (more) {
                if (!own) nextB = reinterpret_cast<const unsigned char *>(G.B + ncol0);
                stage_b(nxt + kABytes);
            }
            mfma_quadrant(1, 1);
            if (more) {
                if (!own) nextA = reinterpret_cast<const unsigned char *>(G.A + (long long)nrow0 * G.lda);
                stage_a(nxt);
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            mfma_quadrant(1, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) { tabB0[j] ^= (unsigned)kBufBytes; tabB1[j] ^= (unsigned)kBufBytes; }
        }
        if (wr == 0) __builtin_amdgcn_s_barrier();
        store_block_half_staged<0, TM, TN, GB>(acc, G, lds + ((KT - 1 + par) & 1) * kBufBytes + wave * 8192, row0, col0, wr, wc, lm, lq);
        if (!has_next) break;
        par = (KT + par) & 1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        stage_b(lds + ((1 + par) & 1) * kBufBytes + kABytes);
        stage_a(lds + ((1 + par) & 1) * kBufBytes);
        tile = ntile; row0 = nrow0; col0 = ncol0;
    }
}

// C[M, N] = alpha * A[M, K] . B[N, K]^T for M <= 8 (the decode step of generate(): one new token per sequence).  No MFMA
// tile to fill: the product is bound by reading B once.  One wave per output column n: the 64 lanes walk row n of B in
// 16-byte pieces (1 KiB per step, coalesced), multiply with the matching pieces of the M rows of A (L2-resident), reduce
// across the wave.  Same accumulate modes as the tiled kernels.
// KS = waves that share one output column, each with a contiguous 1/KS of the contraction (long rows, few columns: more
// waves in flight); their sums meet in LDS.
// MODE 0: the product.  MODE 1 / 2 (round 6, gemm_nt_skinny_lora_kernel): the adapter's down-projection t = scale * x A^T and the projection that consumes it in ONE launch --
// the launch's first workgroups run MODE 1 (the t columns: the same code and order of sums as a launch of their own; an element leaves as ONE 64-bit word, its bf16 bits under
// the step's epoch, in a store other XCDs can see), the others MODE 2 (the projection: a wave asks for the K2 words of a row -- a lane a word, past its XCD's L2 -- together with
// its first weight pieces, and looks at them behind its weight row; words that do not carry the epoch yet are asked for again).  Data and "it is there" in one word: one memory
// round trip, under the weight row's own, where flags beside the data were three in a row behind it (measured: 12.3 us for the o site against 4.6 + 5.2 as two launches).
// The epoch is a device word that grows by one a decode step (ecgb_decode_advance_e): nothing to reset, a replayed graph carries no per-step argument.
struct SkinnySync { unsigned long long *t64; const int *epoch; int pre_blocks; };
template <int MR, int KS, int MODE>
__device__ __forceinline__ void skinny_body(const GemmArgs &G, const long long blk, float (*s_part)[MR], const SkinnySync &Y)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, ks = wave % KS;
    const long long n = blk * (4 / KS) + wave / KS;
    const bool live = n < G.N;
    const unsigned short *b = G.B + (live ? n : 0) * G.ldb;
    const int k_lo = ks * (G.K / KS), k_hi = k_lo + G.K / KS;      // G.K % (8 * KS) == 0 (checked by the launcher)
    float acc[MR];
#pragma unroll
    for (int m = 0; m < MR; ++m) acc[m] = 0.f;
    unsigned long long tw[MR] = {};
    if constexpr (MODE == 2) {
        if (ks == 0) {
#pragma unroll
            for (int m = 0; m < MR; ++m)
                if (m < G.M) tw[m] = __hip_atomic_load(&Y.t64[(long long)m * G.K2 + min(lane, G.K2 - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // four 16-byte pieces of the weight row in flight per lane: with one, a long row (K = 16 384: 32 pieces per lane) is a chain
    // of 32 memory latencies
    constexpr int U = KS == 4 ? 8 : 4;                      // (a quarter of a 16 384-wide row: eight pieces, one round trip)
    for (int k0 = k_lo + lane * 8; k0 < k_hi; k0 += 512 * U) {
        bf16x8 vb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) vb[u] = *reinterpret_cast<const bf16x8 *>(b + min(k0 + 512 * u, k_hi - 8));   // past the end: re-read, not used
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = k0 + 512 * u;
            if (k < k_hi) {
                float fb[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) fb[j] = __uint_as_float((unsigned)(unsigned short)vb[u][j] << 16);
#pragma unroll
                for (int m = 0; m < MR; ++m) {
                    if (m < G.M) {
                        const bf16x8 va = *reinterpret_cast<const bf16x8 *>(G.A + (long long)m * G.lda + k);
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[m] += __uint_as_float((unsigned)(unsigned short)va[j] << 16) * fb[j];
                    }
                }
            }
        }
    }
    if (G.K2 > 0 && ks == 0) {                               // second operand pair (the LoRA branch of a decode step: K2 = 64, one piece for eight lanes)
        if constexpr (MODE == 2) {                           // t is being written by this launch's first workgroups: every word of the row must carry the step's epoch (K2 <= 64)
            const unsigned long long want = (unsigned long long)(unsigned)*Y.epoch;
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                if (m < G.M) {
                    while (!__all(lane >= G.K2 || (tw[m] >> 16) == want)) {
                        __builtin_amdgcn_s_sleep(1);
                        tw[m] = __hip_atomic_load(&Y.t64[(long long)m * G.K2 + min(lane, G.K2 - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
        }
        float tg[MR][8];                                     // MODE 2: lane L takes t[8 L .. 8 L + 7] from the lanes that hold them (every lane of the wave takes part in the shuffles)
        if constexpr (MODE == 2) {
#pragma unroll
            for (int m = 0; m < MR; ++m)
#pragma unroll
                for (int j = 0; j < 8; ++j) tg[m][j] = __uint_as_float(((unsigned)__shfl((int)(unsigned)tw[m], (lane * 8 + j) & 63, 64) & 0xFFFFu) << 16);
        }
        for (int k = lane * 8; k < G.K2; k += 512) {
            const bf16x8 vb = *reinterpret_cast<const bf16x8 *>(G.B2 + (live ? n : 0) * G.ldb2 + k);
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                if (m < G.M) {
                    if constexpr (MODE == 2) {               // ... and adds the products in the order of the plain form
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[m] += tg[m][j] * __uint_as_float((unsigned)(unsigned short)vb[j] << 16);
                    } else {
                        const bf16x8 va = *reinterpret_cast<const bf16x8 *>(G.A2 + (long long)m * G.lda2 + k);
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[m] += __uint_as_float((unsigned)(unsigned short)va[j] << 16) * __uint_as_float((unsigned)(unsigned short)vb[j] << 16);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) acc[m] += __shfl_xor(acc[m], d, 64);
    if constexpr (KS > 1) {
        if (lane == 0) {
#pragma unroll
            for (int m = 0; m < MR; ++m) s_part[wave][m] = acc[m];
        }
        __syncthreads();
        if (ks == 0 && lane == 0) {
#pragma unroll
            for (int m = 0; m < MR; ++m)
#pragma unroll
                for (int t = 1; t < KS; ++t) acc[m] += s_part[wave + t][m];
        }
    }
    if (live && ks == 0 && lane == 0) {
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            if (m >= G.M) break;
            const float v = acc[m] * G.alpha;
            if (G.accumulate_f32 == 1) reinterpret_cast<float *>(G.C)[(long long)m * G.ldc + n] += v;
            else {
                unsigned short *q = reinterpret_cast<unsigned short *>(G.C) + (long long)m * G.ldc + n;
                const unsigned short r = f2bf_rn(G.accumulate_f32 == 2 ? __uint_as_float((unsigned)*q << 16) + v : v);
                if constexpr (MODE == 1)                     // (G.C is not written in this mode: the element and the epoch as one word)
                    __hip_atomic_store(&Y.t64[(long long)m * G.N + n], ((unsigned long long)(unsigned)*Y.epoch << 16) | r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else *q = r;
            }
        }
    }
}
template <int MR, int KS>
__global__ __launch_bounds__(256) void gemm_nt_skinny_kernel(GemmArgs G)
{
    __shared__ float s_part[4][MR];
    skinny_body<MR, KS, 0>(G, blockIdx.x, s_part, SkinnySync{});
}
// P: the adapter's down-projection (C = t, N = the t columns), G: the projection with A2 = t; grid = Y.pre_blocks + G's blocks (the t columns FIRST: they are dispatched first
// and wait for nobody)
template <int MR, int KS>
__global__ __launch_bounds__(256) void gemm_nt_skinny_lora_kernel(GemmArgs G, GemmArgs P, SkinnySync Y)
{
    __shared__ float s_part[4][MR];
    if ((int)blockIdx.x < Y.pre_blocks) skinny_body<MR, KS, 1>(P, blockIdx.x, s_part, Y);
    else skinny_body<MR, KS, 2>(G, (long long)blockIdx.x - Y.pre_blocks, s_part, Y);
}

// The gate|up projection of a decode step with the GLU folded in: one wave per column n of H computes the gate column n and the up column n + glu_I (both weight rows
// stream by once, as in gemm_nt_skinny_kernel), rounds them to bf16 as the stored projection would be, and writes act(gate) * up -- store_tile_glu's arithmetic: the same
// bits as the few-row GEMM followed by ecgb_glu_fwd, one launch and one write / read of gate|up less per layer and token.  G.C (gate|up) may be null.
template <int MR, int EPI>
__global__ __launch_bounds__(256) void gemm_nt_skinny_glu_kernel(GemmArgs G)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long n = (long long)blockIdx.x * 4 + wave;
    const bool live = n < G.glu_I;
    const unsigned short *bg = G.B + (live ? n : 0) * G.ldb, *bu = bg + (long long)G.glu_I * G.ldb;
    float ag[MR], au[MR];
#pragma unroll
    for (int m = 0; m < MR; ++m) { ag[m] = 0.f; au[m] = 0.f; }
    constexpr int U = 4;                                   // 16-byte pieces of each of the two weight rows in flight per lane (round 5: four, a 2 048-wide row is ONE memory round trip; the same order of sums)
    auto fma8 = [&](const bf16x8 &va, const bf16x8 &vb, float &acc) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += __uint_as_float((unsigned)(unsigned short)va[j] << 16) * __uint_as_float((unsigned)(unsigned short)vb[j] << 16);
    };
    for (int k0 = lane * 8; k0 < G.K; k0 += 512 * U) {
        bf16x8 vg[U], vu[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = min(k0 + 512 * u, G.K - 8);
            vg[u] = *reinterpret_cast<const bf16x8 *>(bg + k);
            vu[u] = *reinterpret_cast<const bf16x8 *>(bu + k);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = k0 + 512 * u;
            if (k < G.K) {
#pragma unroll
                for (int m = 0; m < MR; ++m) {
                    if (m < G.M) {
                        const bf16x8 va = *reinterpret_cast<const bf16x8 *>(G.A + (long long)m * G.lda + k);
                        fma8(va, vg[u], ag[m]);
                        fma8(va, vu[u], au[m]);
                    }
                }
            }
        }
    }
    if (G.K2 > 0) {                                         // the LoRA pair behind the first, as in gemm_nt_skinny_kernel
        for (int k = lane * 8; k < G.K2; k += 512) {
            const bf16x8 vg = *reinterpret_cast<const bf16x8 *>(G.B2 + (live ? n : 0) * G.ldb2 + k);
            const bf16x8 vu = *reinterpret_cast<const bf16x8 *>(G.B2 + ((live ? n : 0) + (long long)G.glu_I) * G.ldb2 + k);
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                if (m < G.M) {
                    const bf16x8 va = *reinterpret_cast<const bf16x8 *>(G.A2 + (long long)m * G.lda2 + k);
                    fma8(va, vg, ag[m]);
                    fma8(va, vu, au[m]);
                }
            }
        }
    }
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { ag[m] += __shfl_xor(ag[m], d, 64); au[m] += __shfl_xor(au[m], d, 64); }
    if (live && lane == 0) {
        unsigned short *C = reinterpret_cast<unsigned short *>(G.C);
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            if (m >= G.M) break;
            const unsigned short g16 = f2bf_rn(ag[m] * G.alpha), u16 = f2bf_rn(au[m] * G.alpha);
            const float a = __uint_as_float((unsigned)f2bf_rn(ecgb::glu_act<EPI == 2>(__uint_as_float((unsigned)g16 << 16))) << 16);
            G.H[(long long)m * G.ldh + n] = f2bf_rn(a * __uint_as_float((unsigned)u16 << 16));
            if (C) { C[(long long)m * G.ldc + n] = g16; C[(long long)m * G.ldc + G.glu_I + n] = u16; }
        }
    }
}

// More than two rows and many columns (the batched decode step's gate||up and output-head products): with one column per
// wave every column re-reads the M activation pieces through L1 -- 8x the weight traffic at M = 8, and L1 bandwidth, not
// HBM, is the limit (2 TB/s).  Here a wave takes COLS columns: the activation pieces are loaded and unpacked once per
// K-step and used for all of them.
// KS = 4: the four waves of a workgroup share the same COLS columns and take a quarter of the contraction each (long rows,
// few columns), their sums meet in LDS.
template <int MR, int COLS, int KS>
__global__ __launch_bounds__(256) void gemm_nt_skinny_cols_kernel(GemmArgs G)
{
    static_assert(KS == 1 || KS == 4, "one wave per column group, or all four");
    __shared__ float s_part[KS == 4 ? 4 : 1][COLS][MR];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long n0 = (KS == 4 ? (long long)blockIdx.x : (long long)blockIdx.x * 4 + wave) * COLS;
    if (KS == 1 && n0 >= G.N) return;                       // (KS == 4: n0 is the same for the whole workgroup and < N by the grid size)
    const int k_lo = (KS == 4 ? wave : 0) * (G.K / KS), k_hi = k_lo + G.K / KS;
    float acc[COLS][MR];
#pragma unroll
    for (int c = 0; c < COLS; ++c)
#pragma unroll
        for (int m = 0; m < MR; ++m) acc[c][m] = 0.f;
    for (int k = k_lo + lane * 8; k < k_hi; k += 512) {
        bf16x8 vb[COLS], va[MR];
#pragma unroll
        for (int c = 0; c < COLS; ++c) vb[c] = *reinterpret_cast<const bf16x8 *>(G.B + min(n0 + c, (long long)G.N - 1) * G.ldb + k);
#pragma unroll
        for (int m = 0; m < MR; ++m) va[m] = *reinterpret_cast<const bf16x8 *>(G.A + (long long)min(m, G.M - 1) * G.lda + k);
        float fa[MR][8];
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int j = 0; j < 8; ++j) fa[m][j] = __uint_as_float((unsigned)(unsigned short)va[m][j] << 16);
#pragma unroll
        for (int c = 0; c < COLS; ++c) {
            float fb[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) fb[j] = __uint_as_float((unsigned)(unsigned short)vb[c][j] << 16);
#pragma unroll
            for (int m = 0; m < MR; ++m)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[c][m] += fa[m][j] * fb[j];
        }
    }
    if (G.K2 > 0 && (KS == 1 || wave == 0)) {               // second operand pair (see gemm_nt_skinny_kernel)
        for (int k = lane * 8; k < G.K2; k += 512) {
#pragma unroll
            for (int c = 0; c < COLS; ++c) {
                const bf16x8 vb = *reinterpret_cast<const bf16x8 *>(G.B2 + min(n0 + c, (long long)G.N - 1) * G.ldb2 + k);
#pragma unroll
                for (int m = 0; m < MR; ++m) {
                    const bf16x8 va = *reinterpret_cast<const bf16x8 *>(G.A2 + (long long)min(m, G.M - 1) * G.lda2 + k);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[c][m] += __uint_as_float((unsigned)(unsigned short)va[j] << 16) * __uint_as_float((unsigned)(unsigned short)vb[j] << 16);
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < COLS; ++c)
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) acc[c][m] += __shfl_xor(acc[c][m], d, 64);
    if constexpr (KS == 4) {
        if (lane == 0) {
#pragma unroll
            for (int c = 0; c < COLS; ++c)
#pragma unroll
                for (int m = 0; m < MR; ++m) s_part[wave][c][m] = acc[c][m];
        }
        __syncthreads();
        if (wave != 0) return;
        if (lane == 0) {
#pragma unroll
            for (int c = 0; c < COLS; ++c)
#pragma unroll
                for (int m = 0; m < MR; ++m) acc[c][m] += s_part[1][c][m] + s_part[2][c][m] + s_part[3][c][m];
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < COLS; ++c) {
            const long long n = n0 + c;
            if (n >= G.N) break;
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                if (m >= G.M) break;
                const float v = acc[c][m] * G.alpha;
                if (G.accumulate_f32 == 1) reinterpret_cast<float *>(G.C)[(long long)m * G.ldc + n] += v;
                else {
                    unsigned short *q = reinterpret_cast<unsigned short *>(G.C) + (long long)m * G.ldc + n;
                    *q = f2bf_rn(G.accumulate_f32 == 2 ? __uint_as_float((unsigned)*q << 16) + v : v);
                }
            }
        }
    }
}

// tile order of the 256x256 / 128x128 kernels inside an XCD's range (tile_of): 0 / 1 = row by row (default), g = blocks of g tile rows (A/B: ecgb_set_gemm_group_m).
// Measured (scripts/dev_gemm_group_m.py, the twelve GEMM shapes of a C3 layer, one MI355X, min of 3 rounds): 9.43 ms per layer row by row, 9.53 with blocks of 4,
// 9.73 with 8, 10.32 with 16 -- the long contractions lose most (NT down, K 8192: 0.817 -> 0.894 ms at 8).  Sharing both panels in L2 is not what these kernels lack.
int g_gemm_group_m = 0;
int g_gemm_tile = 0;   // 0 auto; 259 = 256x256 phased, one tile per workgroup (no persistent tile loop: A/B); forced: 128 = 128x128 tile, 256 = 256x256 phased on 16x16x32 MFMA (the default for big problems),
                       // 257 = 256x256 on 32x32x16 MFMA, 258 = 256x256 on 16x16x32 with one barrier pair per K-tile (the earlier kernels, kept for A/B)
// persistent tile loop (gemm_nt_kernel_m16pp): whole interior tiles, no batch, plain bf16 store, at least two rounds of tiles per CU
int n_cus()
{
    static int n = 0;
    if (!n) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    }
    return n;
}
// Persistent workgroups own a STATIC share of the tiles and fill every CU: a kernel of another stream that holds CUs while one runs (an RCCL
// collective under the backward of a data-parallel step) would leave the late-starting workgroups' shares to run after all the others.  The
// input-gradient GEMMs are the ones that run beside the gradient exchange: ecgb_set_gemm_backward_persistent(0) sends them to the one-tile kernels
// (parallel.GradAllReduce does so when it is active with more than one rank); the forward GEMMs never overlap a collective.
int g_nn_persist = 1;
bool persist_ok(const GemmArgs &G, int batch, bool glu)
{
    const int KT = G.K / BK + (G.K2 > 0 ? G.K2 / BK : 0);
    const long long tiles = (long long)(G.M / 256) * (G.N / 256);
    // (long contractions gain nothing -- the prologue is 0.5 % of a K = 16 384 tile -- and lose the dynamic dispatch's load balancing: measured +1.5 %)
    return g_gemm_tile != 259 && batch == 1 && !G.inner && G.M % 256 == 0 && G.N % 256 == 0 && KT >= 2 && KT <= 64 && G.accumulate_f32 == 0 && (G.ldc & 7) == 0 &&
           (!glu || ((G.ldh & 7) == 0 && (G.glu_I & 7) == 0)) && tiles >= 2 * n_cus();
}


int launch_gemm(GemmArgs &G, int batch, hipStream_t stream)
{
    // the 256x256 tile halves L2->LDS traffic per FLOP but needs enough tiles to fill the chip
    const long long tiles256 = (long long)((G.M + 255) / 256) * ((G.N + 255) / 256) * batch;
    const bool big = g_gemm_tile >= 256 || (g_gemm_tile == 0 && tiles256 >= 192 && G.M >= 256 && G.N >= 256);
    hipError_t e;
    if (big && g_gemm_tile != 257 && g_gemm_tile != 258) {
        constexpr int lds = 2 * (256 + 256) * BK * 2;
        G.tiles_m = (G.M + 255) / 256; G.tiles_n = (G.N + 255) / 256;
        if (persist_ok(G, batch, false)) {
            auto kern = G.K2 ? gemm_nt_kernel_m16pp<true, 0> : gemm_nt_kernel_m16pp<false, 0>;
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e == hipSuccess) hipLaunchKernelGGL(kern, dim3((unsigned)n_cus(), 1, 1), dim3(512), lds, stream, G);
        } else {
            auto kern = G.K2 ? gemm_nt_kernel_m16p<256, 256, 2, 4, true> : gemm_nt_kernel_m16p<256, 256, 2, 4>;
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e == hipSuccess)
                hipLaunchKernelGGL(kern, dim3((unsigned)(G.tiles_m * G.tiles_n), 1, (unsigned)batch), dim3(512), lds, stream, G);
        }
    } else if (big && g_gemm_tile == 258) {
        constexpr int lds = 2 * (256 + 256) * BK * 2;
        auto kern = G.K2 ? gemm_nt_kernel_m16<256, 256, 2, 4, true> : gemm_nt_kernel_m16<256, 256, 2, 4>;
        G.tiles_m = (G.M + 255) / 256; G.tiles_n = (G.N + 255) / 256;
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e == hipSuccess)
            hipLaunchKernelGGL(kern, dim3((unsigned)(G.tiles_m * G.tiles_n), 1, (unsigned)batch), dim3(512), lds, stream, G);
    } else if (big) {
        constexpr int lds = 2 * (256 + 256) * BK * 2;
        auto kern = G.K2 ? gemm_nt_kernel<256, 256, 2, 4, true> : gemm_nt_kernel<256, 256, 2, 4>;
        G.tiles_m = (G.M + 255) / 256; G.tiles_n = (G.N + 255) / 256;
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e == hipSuccess)
            hipLaunchKernelGGL(kern, dim3((unsigned)(G.tiles_m * G.tiles_n), 1, (unsigned)batch), dim3(512), lds, stream, G);
    } else {
        constexpr int lds = 2 * (128 + 128) * BK * 2;
        auto kern = G.K2 ? gemm_nt_kernel<128, 128, 2, 2, true> : gemm_nt_kernel<128, 128, 2, 2>;
        G.tiles_m = (G.M + 127) / 128; G.tiles_n = (G.N + 127) / 128;
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e == hipSuccess)
            hipLaunchKernelGGL(kern, dim3((unsigned)(G.tiles_m * G.tiles_n), 1, (unsigned)batch), dim3(256), lds, stream, G);
    }
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) { ecgb::set_error(std::string("gemm_nt_kernel: ") + hipGetErrorString(e)); return ECGB_ERR_HIP; }
    return ECGB_OK;
}

}  // namespace

// csrc/gemm_w4.hip: the four-wave kernel (128x128 wave tiles, one wave per SIMD) for plain NT products of whole 256x256 tiles
namespace ecgb {
bool gemm_w4_applies(const void *a_dev, long long lda, const void *b_dev, long long ldb, const void *c_dev, long long ldc, int M, int N, int K);
bool gemm_w4_span_ok(int lay, long long lda, long long ldb, long long K, long long extra_rows_b);      // 32-bit DMA offsets over the whole contraction (gemm_w4.hip)
int gemm_w4_launch(const void *a_dev, long long lda, const void *b_dev, long long ldb, void *c_dev, long long ldc, int M, int N, int K, float alpha, void *stream,
                   int epi, void *h_dev, long long ldh, const void *a2_dev, long long lda2, const void *b2_dev, long long ldb2, int K2,
                   const float *rope_cos, const float *rope_sin, int rope_cols, int lay, const void *gu_dev = nullptr, long long ldgu = 0,
                   const void *ldt_dev = nullptr, const void *lat_dev = nullptr, float lscale = 0.f, unsigned lthr = 0, unsigned lseed = 0);
}
namespace { int g_gemm_w4 = 1; }
extern "C" int ecgb_set_gemm_w4(int on)
{
    g_gemm_w4 = on < 0 ? 0 : on > 2 ? 2 : on;           // 2: every form the four-wave kernel has, also where the eight-wave one measured faster (tests)
    return ECGB_OK;
}

extern "C" int ecgb_set_gemm_group_m(int group_m)
{
    if (group_m < 0 || group_m > 64) { ecgb::set_error("ecgb_set_gemm_group_m: 0..64"); return ECGB_ERR_INVALID; }
    g_gemm_group_m = group_m;
    return ECGB_OK;
}

extern "C" int ecgb_set_gemm_backward_persistent(int on)
{
    g_nn_persist = on ? 1 : 0;
    return ECGB_OK;
}

extern "C" int ecgb_get_gemm_backward_persistent(void) { return g_nn_persist; }

extern "C" int ecgb_set_gemm_tile(int tile)
{
    if (tile != 0 && tile != 128 && (tile < 256 || tile > 259)) { ecgb::set_error("ecgb_set_gemm_tile: 0, 128, 256, 257, 258 or 259"); return ECGB_ERR_INVALID; }
    g_gemm_tile = tile;
    return ECGB_OK;
}

extern "C" int ecgb_gemm_nt_bf16(const void *a_dev, long long lda, const void *b_dev, long long ldb, void *c_dev, long long ldc,
                                 int M, int N, int K, float alpha, int accumulate_f32, int batch, long long batch_a,
                                 long long batch_b, long long batch_c, void *stream)
{
    if (!a_dev || !b_dev || !c_dev || M <= 0 || N <= 0 || K <= 0 || batch <= 0) {
        ecgb::set_error("ecgb_gemm_nt_bf16: bad argument");
        return ECGB_ERR_INVALID;
    }
    if (K % BK || lda % 8 || ldb % 8 || ((uintptr_t)a_dev & 15) || ((uintptr_t)b_dev & 15)) {
        ecgb::set_error("ecgb_gemm_nt_bf16: K must be a multiple of 64 and operands 16-byte aligned with strides % 8 == 0");
        return ECGB_ERR_UNSUPPORTED;
    }
    GemmArgs G{};
    G.group_m = g_gemm_group_m;
    G.A = (const unsigned short *)a_dev; G.B = (const unsigned short *)b_dev; G.C = c_dev;
    G.M = M; G.N = N; G.K = K; G.lda = lda; G.ldb = ldb; G.ldc = ldc;
    G.batch_a = batch_a; G.batch_b = batch_b; G.batch_c = batch_c;
    G.accumulate_f32 = accumulate_f32; G.alpha = alpha;
    G.A2 = G.B2 = nullptr; G.lda2 = G.ldb2 = 0; G.K2 = 0;
    G.inner = 0; G.outer_a = G.inner_a = G.outer_b = G.inner_b = G.outer_c = G.inner_c = 0; G.div_a = G.div_b = 1;
    if (M <= 8 && batch == 1 && K % 8 == 0) {   // a few rows: bandwidth-bound column-per-wave kernel (ecgb_gemm_nt_bf16_cat has the same dispatch)
        const bool split = (K >= 8192 && K % 32 == 0 && N <= 8192);      // long rows, few columns: four waves per column
        const dim3 grid(split ? (unsigned)N : (unsigned)((N + 3) / 4));
        if (M <= 2) {
            if (split) hipLaunchKernelGGL((gemm_nt_skinny_kernel<2, 4>), grid, dim3(256), 0, (hipStream_t)stream, G);
            else hipLaunchKernelGGL((gemm_nt_skinny_kernel<2, 1>), grid, dim3(256), 0, (hipStream_t)stream, G);
        } else {
            if (split) hipLaunchKernelGGL((gemm_nt_skinny_cols_kernel<8, 4, 4>), dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, G);
            else if (N >= 8192) hipLaunchKernelGGL((gemm_nt_skinny_cols_kernel<8, 4, 1>), dim3((unsigned)((N + 15) / 16)), dim3(256), 0, (hipStream_t)stream, G);
            else hipLaunchKernelGGL((gemm_nt_skinny_kernel<8, 1>), grid, dim3(256), 0, (hipStream_t)stream, G);
        }
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) { ecgb::set_error(std::string("gemm_nt_skinny_kernel: ") + hipGetErrorString(e)); return ECGB_ERR_HIP; }
        return ECGB_OK;
    }
    // whole 256x256 tiles, plain bf16 store, one problem, at least two tiles per CU: the four-wave kernel (the same bits; 1-6 % faster on the step's forward shapes)
    if (g_gemm_w4 && g_gemm_tile == 0 && batch == 1 && accumulate_f32 == 0 && ecgb::gemm_w4_span_ok(0, lda, ldb, K, 0) && ecgb::gemm_w4_applies(a_dev, lda, b_dev, ldb, c_dev, ldc, M, N, K))
        return ecgb::gemm_w4_launch(a_dev, lda, b_dev, ldb, c_dev, ldc, M, N, K, alpha, stream, 0, nullptr, 0, nullptr, 0, nullptr, 0, 0, nullptr, nullptr, 0, 0);
    return launch_gemm(G, batch, (hipStream_t)stream);
}

extern "C" int ecgb_gemm_nt_bf16_cat(const void *a_dev, long long lda, const void *b_dev, long long ldb, const void *a2_dev, long long lda2,
                                     const void *b2_dev, long long ldb2, int K2, void *c_dev, long long ldc, int M, int N, int K, float alpha,
                                     int accumulate_f32, void *stream)
{
    if (!a_dev || !b_dev || !a2_dev || !b2_dev || !c_dev || M <= 0 || N <= 0 || K <= 0 || K2 <= 0) {
        ecgb::set_error("ecgb_gemm_nt_bf16_cat: bad argument");
        return ECGB_ERR_INVALID;
    }
    if (K % BK || K2 % BK || lda % 8 || ldb % 8 || lda2 % 8 || ldb2 % 8 || (((uintptr_t)a_dev | (uintptr_t)b_dev | (uintptr_t)a2_dev | (uintptr_t)b2_dev) & 15)) {
        ecgb::set_error("ecgb_gemm_nt_bf16_cat: K and K2 must be multiples of 64 and operands 16-byte aligned with strides % 8 == 0");
        return ECGB_ERR_UNSUPPORTED;
    }
    GemmArgs G{};
    G.group_m = g_gemm_group_m;
    G.A = (const unsigned short *)a_dev; G.B = (const unsigned short *)b_dev; G.C = c_dev;
    G.M = M; G.N = N; G.K = K; G.lda = lda; G.ldb = ldb; G.ldc = ldc;
    G.batch_a = G.batch_b = G.batch_c = 0;
    G.accumulate_f32 = accumulate_f32; G.alpha = alpha;
    G.A2 = (const unsigned short *)a2_dev; G.B2 = (const unsigned short *)b2_dev; G.lda2 = lda2; G.ldb2 = ldb2; G.K2 = K2;
    G.inner = 0; G.outer_a = G.inner_a = G.outer_b = G.inner_b = G.outer_c = G.inner_c = 0; G.div_a = G.div_b = 1;
    if (M <= 8) {   // a few rows (a decode step with adapters): the column-per-wave kernels take the second pair behind the first, one launch
        const bool split = (K >= 8192 && K % 32 == 0 && N <= 8192);
        const dim3 grid(split ? (unsigned)N : (unsigned)((N + 3) / 4));
        if (M <= 2) {
            if (split) hipLaunchKernelGGL((gemm_nt_skinny_kernel<2, 4>), grid, dim3(256), 0, (hipStream_t)stream, G);
            else hipLaunchKernelGGL((gemm_nt_skinny_kernel<2, 1>), grid, dim3(256), 0, (hipStream_t)stream, G);
        } else {
            if (split) hipLaunchKernelGGL((gemm_nt_skinny_cols_kernel<8, 4, 4>), dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, G);
            else if (N >= 8192) hipLaunchKernelGGL((gemm_nt_skinny_cols_kernel<8, 4, 1>), dim3((unsigned)((N + 15) / 16)), dim3(256), 0, (hipStream_t)stream, G);
            else hipLaunchKernelGGL((gemm_nt_skinny_kernel<8, 1>), grid, dim3(256), 0, (hipStream_t)stream, G);
        }
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) { ecgb::set_error(std::string("gemm_nt_skinny_kernel (cat): ") + hipGetErrorString(e)); return ECGB_ERR_HIP; }
        return ECGB_OK;
    }
    if (g_gemm_w4 && g_gemm_tile == 0 && accumulate_f32 == 0 && ecgb::gemm_w4_span_ok(0, lda2, ldb2, K2, 0) && ecgb::gemm_w4_span_ok(0, lda, ldb, K, 0) &&
        ecgb::gemm_w4_applies(a_dev, lda, b_dev, ldb, c_dev, ldc, M, N, K))
        return ecgb::gemm_w4_launch(a_dev, lda, b_dev, ldb, c_dev, ldc, M, N, K, alpha, stream, 0, nullptr, 0, a2_dev, lda2, b2_dev, ldb2, K2, nullptr, nullptr, 0, 0);
    return launch_gemm(G, 1, (hipStream_t)stream);
}

// One launch for a decode step's adapter site (o, down): t = t_scale * x A_lora^T by the launch's first workgroups (into t64_dev: [M, K2] 64-bit words, an element's bf16 bits
// under the step's epoch), y = x W^T + t B_lora^T by the others (see skinny_body).  The bits of ecgb_gemm_nt_bf16(x, A_lora, alpha = t_scale) followed by
// ecgb_gemm_nt_bf16_cat.  M <= 2, K2 <= 64; epoch_dev: a device word that differs from one call on this site to the next (ecgb_decode_advance_e adds one a step).
extern "C" int ecgb_gemm_nt_bf16_lora_decode(const void *x_dev, long long ldx, const void *w_dev, long long ldw, const void *a_lora_dev, long long lda_lora, float t_scale,
                                             const void *b_lora_dev, long long ldb_lora, int K2, unsigned long long *t64_dev, void *y_dev, long long ldy, int M, int N, int K,
                                             const int *epoch_dev, void *stream)
{
    if (!x_dev || !w_dev || !a_lora_dev || !b_lora_dev || !t64_dev || !y_dev || !epoch_dev || M <= 0 || N <= 0 || K <= 0 || K2 <= 0) {
        ecgb::set_error("ecgb_gemm_nt_bf16_lora_decode: bad argument");
        return ECGB_ERR_INVALID;
    }
    if (M > 2 || K2 > 64 || K % BK || K2 % 8 || ldx % 8 || ldw % 8 || lda_lora % 8 || ldb_lora % 8 ||
        (((uintptr_t)x_dev | (uintptr_t)w_dev | (uintptr_t)a_lora_dev | (uintptr_t)b_lora_dev) & 15) || ((uintptr_t)t64_dev & 7)) {
        ecgb::set_error("ecgb_gemm_nt_bf16_lora_decode: one or two rows, K a multiple of 64, K2 <= 64, operands 16-byte aligned with strides % 8 == 0");
        return ECGB_ERR_UNSUPPORTED;
    }
    const bool split = (K >= 8192 && K % 32 == 0 && N <= 8192);            // (ecgb_gemm_nt_bf16's rule; K2 <= 8192 always: both products take the same kernel form)
    if (split != (K >= 8192 && K % 32 == 0)) { ecgb::set_error("ecgb_gemm_nt_bf16_lora_decode: the two products would take different kernels"); return ECGB_ERR_UNSUPPORTED; }
    GemmArgs G{}, P{};
    G.A = (const unsigned short *)x_dev; G.B = (const unsigned short *)w_dev; G.C = y_dev;
    G.M = M; G.N = N; G.K = K; G.lda = ldx; G.ldb = ldw; G.ldc = ldy; G.accumulate_f32 = 0; G.alpha = 1.f;
    G.A2 = nullptr; G.B2 = (const unsigned short *)b_lora_dev; G.lda2 = 0; G.ldb2 = ldb_lora; G.K2 = K2; G.div_a = G.div_b = 1;      // (A2: the words of t64_dev)
    P.A = (const unsigned short *)x_dev; P.B = (const unsigned short *)a_lora_dev; P.C = t64_dev;
    P.M = M; P.N = K2; P.K = K; P.lda = ldx; P.ldb = lda_lora; P.ldc = K2; P.accumulate_f32 = 0; P.alpha = t_scale; P.K2 = 0; P.div_a = P.div_b = 1;
    SkinnySync Y{t64_dev, epoch_dev, split ? K2 : (K2 + 3) / 4};
    const dim3 grid((unsigned)Y.pre_blocks + (split ? (unsigned)N : (unsigned)((N + 3) / 4)));
    if (split) hipLaunchKernelGGL((gemm_nt_skinny_lora_kernel<2, 4>), grid, dim3(256), 0, (hipStream_t)stream, G, P, Y);
    else hipLaunchKernelGGL((gemm_nt_skinny_lora_kernel<2, 1>), grid, dim3(256), 0, (hipStream_t)stream, G, P, Y);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { ecgb::set_error(std::string("gemm_nt_skinny_lora_kernel: ") + hipGetErrorString(e)); return ECGB_ERR_HIP; }
    return ECGB_OK;
}

extern "C" int ecgb_gemm_nt_bf16_heads(const void *a_dev, long long lda, const void *b_dev, long long ldb, void *c_dev,
                                       long long ldc, int M, int N, int K, float alpha, int accumulate_f32, int batch, int inner,
                                       long long outer_a, long long inner_a, int div_a, long long outer_b,
                                       long long inner_b, int div_b, long long outer_c, long long inner_c, void *stream)
{
    if (!a_dev || !b_dev || !c_dev || M <= 0 || N <= 0 || K <= 0 || batch <= 0 || inner <= 0 || div_a <= 0 || div_b <= 0) {
        ecgb::set_error("ecgb_gemm_nt_bf16_heads: bad argument");
        return ECGB_ERR_INVALID;
    }
    if (K % BK || lda % 8 || ldb % 8 || ((uintptr_t)a_dev & 15) || ((uintptr_t)b_dev & 15) || outer_a % 8 || inner_a % 8 ||
        outer_b % 8 || inner_b % 8) {
        ecgb::set_error("ecgb_gemm_nt_bf16_heads: K must be a multiple of 64 and every operand offset 16-byte aligned");
        return ECGB_ERR_UNSUPPORTED;
    }
    GemmArgs G{};
    G.group_m = g_gemm_group_m;
    G.A = (const unsigned short *)a_dev; G.B = (const unsigned short *)b_dev; G.C = c_dev;
    G.M = M; G.N = N; G.K = K; G.lda = lda; G.ldb = ldb; G.ldc = ldc;
    G.batch_a = G.batch_b = G.batch_c = 0;
    G.accumulate_f32 = accumulate_f32; G.alpha = alpha;
    G.A2 = G.B2 = nullptr; G.lda2 = G.ldb2 = 0; G.K2 = 0;
    G.inner = inner; G.outer_a = outer_a; G.inner_a = inner_a; G.div_a = div_a;
    G.outer_b = outer_b; G.inner_b = inner_b; G.div_b = div_b; G.outer_c = outer_c; G.inner_c = inner_c;
    return launch_gemm(G, batch, (hipStream_t)stream);
}

// dW-style product: C[N,K] (bf16, row stride ldc) = alpha * A^T . B, A = [M,N] row-major (lda), B = [M,K] row-major (ldb).
extern "C" int ecgb_gemm_tn_bf16(const void *a_dev, long long lda, const void *b_dev, long long ldb, void *c_dev, long long ldc,
                                 int M, int N, int K, float alpha, int splits, void *stream)
{
    if (!a_dev || !b_dev || !c_dev || M <= 0 || N <= 0 || K <= 0) { ecgb::set_error("ecgb_gemm_tn_bf16: bad argument"); return ECGB_ERR_INVALID; }
    if (M % 64 || N % 8 || K % 8 || N < 8 || K < 8 || lda % 8 || ldb % 8 || ldc % 4 || ((uintptr_t)a_dev & 15) || ((uintptr_t)b_dev & 15) ||
        ((uintptr_t)c_dev & (splits > 1 ? 15 : 7))) {
        ecgb::set_error("ecgb_gemm_tn_bf16: M % 64, N % 8, K % 8, 16-byte aligned operands required");
        return ECGB_ERR_UNSUPPORTED;
    }
    GemmArgs G{};
    G.group_m = g_gemm_group_m;
    G.A = (const unsigned short *)a_dev; G.B = (const unsigned short *)b_dev; G.C = c_dev;
    G.M = M; G.N = N; G.K = K; G.lda = lda; G.ldb = ldb; G.ldc = ldc;
    G.batch_a = G.batch_b = G.batch_c = 0;
    G.A2 = G.B2 = nullptr; G.lda2 = G.ldb2 = 0; G.K2 = 0;
    G.tiles_m = (N + 255) / 256; G.tiles_n = (K + 255) / 256;
    if (splits < 1 || splits > 64) { ecgb::set_error("ecgb_gemm_tn_bf16: splits must be 1..64"); return ECGB_ERR_INVALID; }
    if ((64 * lda + N) * 2 >= (1ll << 32) || (64 * ldb + K) * 2 >= (1ll << 32)) {      // per-lane byte offsets inside a K-tile are 32-bit
        ecgb::set_error("ecgb_gemm_tn_bf16: row strides above 2^24 elements are not supported");
        return ECGB_ERR_UNSUPPORTED;
    }
    // one K-slice, whole tiles, a long contraction: the four-wave kernel on the TN layout (the same bits; dW of the down projection 0.836 -> 0.785 ms, of gate|up 1.589 -> 1.574);
    // persistent, so not while a gradient exchange may hold CUs (g_nn_persist)
    if (splits == 1 && g_gemm_w4 && g_nn_persist && g_gemm_tile == 0 && N % 256 == 0 && K % 256 == 0 && (ldc & 7) == 0 && ((uintptr_t)c_dev & 15) == 0 &&
        ecgb::gemm_w4_span_ok(2, lda, ldb, M, 0) && ecgb::gemm_w4_applies(a_dev, 8, b_dev, 8, c_dev, ldc, N, K, M))
        return ecgb::gemm_w4_launch(a_dev, lda, b_dev, ldb, c_dev, ldc, N, K, M, alpha, stream, 0, nullptr, 0, nullptr, 0, nullptr, 0, 0, nullptr, nullptr, 0, 2);
    G.accumulate_f32 = splits > 1 ? 1 : 0; G.alpha = alpha;      // splits > 1: c_dev is fp32 [splits][N, ldc], one slab per K-slice
    G.split_stride = (long long)N * ldc;
    G.inner = 0; G.outer_a = G.inner_a = G.outer_b = G.inner_b = G.outer_c = G.inner_c = 0; G.div_a = G.div_b = 1;
    constexpr int lds = 2 * (256 + 256) * 64 * 2;
    auto kern = (g_gemm_tile == 258) ? gemm_tn_kernel_m16<256, 256, 2, 4> : gemm_tn_kernel_tr<256, 256, 2, 4>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(kern, dim3((unsigned)(G.tiles_m * G.tiles_n), (unsigned)splits), dim3(512), lds, (hipStream_t)stream, G);
        e = hipGetLastError();
    }
    if (e != hipSuccess) { ecgb::set_error(std::string("gemm_tn_kernel: ") + hipGetErrorString(e)); return ECGB_ERR_HIP; }
    return ECGB_OK;
}

// H[M, I] = act(gate) * up with gate | up = alpha * (A . B^T [+ A2 . B2^T]),  B = [2 I, K] (gate rows, then up rows): the gate|up
// projection of the MLP with the GLU in its epilogue (modeling_llama.py:227-258 `down_proj(act_fn(gate_proj(x)) * up_proj(x))`,
// modeling_gemma.py's GeGLU with gelu_tanh != 0).  c_dev ([M, 2 I] bf16, what the backward needs) may be null.  A2 / B2 ([M, K2],
// [2 I, K2]) as in ecgb_gemm_nt_bf16_cat, or null.  M % 1 free, I % 128 == 0, K % 64 == 0.
extern "C" int ecgb_gemm_nt_glu_bf16(const void *a_dev, long long lda, const void *b_dev, long long ldb, const void *a2_dev, long long lda2,
                                     const void *b2_dev, long long ldb2, int K2, void *c_dev, long long ldc, void *h_dev, long long ldh,
                                     int M, int inter, int K, float alpha, int gelu_tanh, void *stream)
{
    if (!a_dev || !b_dev || !h_dev || M <= 0 || inter <= 0 || K <= 0 || (K2 > 0 && (!a2_dev || !b2_dev))) {
        ecgb::set_error("ecgb_gemm_nt_glu_bf16: bad argument");
        return ECGB_ERR_INVALID;
    }
    if (K % BK || K2 % BK || inter % 128 || lda % 8 || ldb % 8 || lda2 % 8 || ldb2 % 8 || ldc % 4 || ldh % 4 ||
        (((uintptr_t)a_dev | (uintptr_t)b_dev | (uintptr_t)a2_dev | (uintptr_t)b2_dev) & 15) || (((uintptr_t)c_dev | (uintptr_t)h_dev) & 7)) {
        ecgb::set_error("ecgb_gemm_nt_glu_bf16: K % 64, inter % 128, 16-byte aligned operands, strides % 8 (outputs % 4) required");
        return ECGB_ERR_UNSUPPORTED;
    }
    if (M <= 8 && K % 8 == 0 && (K2 <= 0 || K2 % 8 == 0)) {       // a decode step: the few-row kernel with the GLU folded in (bound by reading the two weight rows once)
        GemmArgs G{};
        G.A = (const unsigned short *)a_dev; G.B = (const unsigned short *)b_dev; G.C = c_dev;
        G.M = M; G.N = 2 * inter; G.K = K; G.lda = lda; G.ldb = ldb; G.ldc = ldc; G.alpha = alpha;
        G.A2 = (const unsigned short *)a2_dev; G.B2 = (const unsigned short *)b2_dev; G.lda2 = lda2; G.ldb2 = ldb2; G.K2 = K2 > 0 ? K2 : 0;
        G.H = (unsigned short *)h_dev; G.ldh = ldh; G.glu_I = inter;
        const dim3 grid((unsigned)((inter + 3) / 4));
        if (M <= 2) {
            if (gelu_tanh) hipLaunchKernelGGL((gemm_nt_skinny_glu_kernel<2, 2>), grid, dim3(256), 0, (hipStream_t)stream, G);
            else hipLaunchKernelGGL((gemm_nt_skinny_glu_kernel<2, 1>), grid, dim3(256), 0, (hipStream_t)stream, G);
        } else {
            if (gelu_tanh) hipLaunchKernelGGL((gemm_nt_skinny_glu_kernel<8, 2>), grid, dim3(256), 0, (hipStream_t)stream, G);
            else hipLaunchKernelGGL((gemm_nt_skinny_glu_kernel<8, 1>), grid, dim3(256), 0, (hipStream_t)stream, G);
        }
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) { ecgb::set_error(std::string("gemm_nt_skinny_glu_kernel: ") + hipGetErrorString(e)); return ECGB_ERR_HIP; }
        return ECGB_OK;
    }
    // whole tiles, no second operand pair: the four-wave kernel with the same epilogue (the same bits).  Measured at [32768, 2048] -> 2 x 8192: 1.853 against 1.863 ms, h only
    // 1.669 against 1.703 -- the activation is vector work that a wave alone on its SIMD cannot hide (under the profiler its MFMA pipes are 0.57 busy, the plain kernel's
    // 0.65); with a LoRA pair behind it 1.952 against 1.935 in round 3 -- and 1.75 against 1.94 with round 4's K-tile schedule, so that form goes there too now.
    if (g_gemm_w4 && g_gemm_tile == 0 && (K2 <= 0 || ecgb::gemm_w4_span_ok(0, lda2, ldb2, K2, inter)) && (ldh & 7) == 0 && ((uintptr_t)h_dev & 15) == 0 && (!c_dev || ((ldc & 7) == 0 && ((uintptr_t)c_dev & 15) == 0)) &&
        ecgb::gemm_w4_span_ok(0, lda, ldb, K, inter) && ecgb::gemm_w4_applies(a_dev, lda, b_dev, ldb, h_dev, ldh, M, 2 * inter, K))
        return ecgb::gemm_w4_launch(a_dev, lda, b_dev, ldb, c_dev, ldc, M, 2 * inter, K, alpha, stream, gelu_tanh ? 2 : 1, h_dev, ldh, a2_dev, lda2, b2_dev, ldb2, K2, nullptr, nullptr, 0, 0);
    GemmArgs G{};
    G.group_m = g_gemm_group_m;
    G.A = (const unsigned short *)a_dev; G.B = (const unsigned short *)b_dev; G.C = c_dev;
    G.M = M; G.N = 2 * inter; G.K = K; G.lda = lda; G.ldb = ldb; G.ldc = ldc;
    G.batch_a = G.batch_b = G.batch_c = 0;
    G.accumulate_f32 = 0; G.alpha = alpha;
    G.A2 = (const unsigned short *)a2_dev; G.B2 = (const unsigned short *)b2_dev; G.lda2 = lda2; G.ldb2 = ldb2; G.K2 = K2 > 0 ? K2 : 0;
    G.inner = 0; G.outer_a = G.inner_a = G.outer_b = G.inner_b = G.outer_c = G.inner_c = 0; G.div_a = G.div_b = 1;
    G.H = (unsigned short *)h_dev; G.ldh = ldh; G.glu_I = inter;
    G.tiles_m = (M + 255) / 256; G.tiles_n = (2 * inter) / 256;
    constexpr int lds = 2 * (256 + 256) * BK * 2;
    using Kern = void (*)(GemmArgs);
    Kern kern = gelu_tanh ? (G.K2 ? (Kern)gemm_nt_kernel_m16p<256, 256, 2, 4, true, 2> : (Kern)gemm_nt_kernel_m16p<256, 256, 2, 4, false, 2>)
                          : (G.K2 ? (Kern)gemm_nt_kernel_m16p<256, 256, 2, 4, true, 1> : (Kern)gemm_nt_kernel_m16p<256, 256, 2, 4, false, 1>);
    unsigned grid = (unsigned)(G.tiles_m * G.tiles_n);
    if (persist_ok(G, 1, true)) {
        kern = gelu_tanh ? (G.K2 ? (Kern)gemm_nt_kernel_m16pp<true, 2> : (Kern)gemm_nt_kernel_m16pp<false, 2>)
                         : (G.K2 ? (Kern)gemm_nt_kernel_m16pp<true, 1> : (Kern)gemm_nt_kernel_m16pp<false, 1>);
        grid = (unsigned)n_cus();
    }
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(kern, dim3(grid, 1, 1), dim3(512), lds, (hipStream_t)stream, G);
        e = hipGetLastError();
    }
    if (e != hipSuccess) { ecgb::set_error(std::string("gemm_nt_kernel (glu): ") + hipGetErrorString(e)); return ECGB_ERR_HIP; }
    return ECGB_OK;
}

// C[M, N] = alpha * A[M, K] . B[K, N], B row-major (the input gradient dX = dY . W against the weight as nn.Linear stores it, [out, in]):
// what ecgb_gemm_nt_bf16 computes on a transposed copy of B, without the copy.  accumulate as in ecgb_gemm_nt_bf16.
extern "C" int ecgb_gemm_nn_bf16(const void *a_dev, long long lda, const void *b_dev, long long ldb, void *c_dev, long long ldc, int M, int N,
                                 int K, float alpha, int accumulate_f32, void *stream)
{
    if (!a_dev || !b_dev || !c_dev || M <= 0 || N <= 0 || K <= 0) { ecgb::set_error("ecgb_gemm_nn_bf16: bad argument"); return ECGB_ERR_INVALID; }
    if (K % BK || N % 8 || N < 8 || lda % 8 || ldb % 8 || ((uintptr_t)a_dev & 15) || ((uintptr_t)b_dev & 15) ||
        (64 * ldb + N) * 2 >= (1ll << 32) || (256 * lda + K) * 2 >= (1ll << 32)) {
        ecgb::set_error("ecgb_gemm_nn_bf16: K % 64, N % 8, 16-byte aligned operands with strides % 8 (below 2^24) required");
        return ECGB_ERR_UNSUPPORTED;
    }
    // long contractions of whole tiles, plain store: the four-wave kernel on the NN layout (the same bits; [32768, 16384] . [16384, 2048]: 1.63 against 1.72 ms).  It is
    // persistent: not while a gradient exchange may hold CUs (g_nn_persist, see above).
    if (g_gemm_w4 && g_nn_persist && g_gemm_tile == 0 && accumulate_f32 == 0 && (((uintptr_t)c_dev & 15) == 0) && (ldc & 7) == 0 && N % 256 == 0 &&
        ecgb::gemm_w4_span_ok(1, lda, ldb, K, 0) && ecgb::gemm_w4_applies(a_dev, lda, b_dev, ldb, c_dev, ldc, M, N, K))
        return ecgb::gemm_w4_launch(a_dev, lda, b_dev, ldb, c_dev, ldc, M, N, K, alpha, stream, 0, nullptr, 0, nullptr, 0, nullptr, 0, 0, nullptr, nullptr, 0, 1);
    GemmArgs G{};
    G.group_m = g_gemm_group_m;
    G.A = (const unsigned short *)a_dev; G.B = (const unsigned short *)b_dev; G.C = c_dev;
    G.M = M; G.N = N; G.K = K; G.lda = lda; G.ldb = ldb; G.ldc = ldc;
    G.batch_a = G.batch_b = G.batch_c = 0;
    G.accumulate_f32 = accumulate_f32; G.alpha = alpha;
    G.A2 = G.B2 = nullptr; G.lda2 = G.ldb2 = 0; G.K2 = 0;
    G.inner = 0; G.outer_a = G.inner_a = G.outer_b = G.inner_b = G.outer_c = G.inner_c = 0; G.div_a = G.div_b = 1;
    G.tiles_m = (M + 255) / 256; G.tiles_n = (N + 255) / 256;
    constexpr int lds = 2 * (256 + 256) * BK * 2;
    using Kern = void (*)(GemmArgs);
    Kern kern = gemm_nn_kernel_m16p<256, 256, 2, 4>;
    unsigned grid = (unsigned)(G.tiles_m * G.tiles_n);
    if (g_nn_persist && persist_ok(G, 1, false)) { kern = gemm_nn_kernel_m16pp<0>; grid = (unsigned)n_cus(); }
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(kern, dim3(grid, 1, 1), dim3(512), lds, (hipStream_t)stream, G);
        e = hipGetLastError();
    }
    if (e != hipSuccess) { ecgb::set_error(std::string("gemm_nn_kernel: ") + hipGetErrorString(e)); return ECGB_ERR_HIP; }
    return ECGB_OK;
}

// ecgb_gemm_nn_bf16 for few output tiles and a long contraction (the loss head's input gradient: 1 280 labelled rows x hidden 2048 over a vocabulary of
// 132 608): n_splits K-slices per output tile, each into its own fp32 slab [M, N] (slabs_dev: n_splits x M x N floats, 16-byte aligned), added in slice
// order into c_dev (bf16 [M, ldc]) -- no atomics, the same bits every launch.
extern "C" int ecgb_sum_slabs_bf16(const float *slabs_dev, long long slab_stride, int n_slabs, void *out_dev, size_t n, int accumulate, void *stream);
extern "C" int ecgb_gemm_nn_splitk_bf16(const void *a_dev, long long lda, const void *b_dev, long long ldb, void *c_dev, float *slabs_dev, int M, int N,
                                        int K, int n_splits, float alpha, void *stream)
{
    if (!a_dev || !b_dev || !c_dev || !slabs_dev || M <= 0 || N <= 0 || K <= 0 || n_splits < 1 || n_splits > 64) { ecgb::set_error("ecgb_gemm_nn_splitk_bf16: bad argument"); return ECGB_ERR_INVALID; }
    if (K % BK || N % 8 || N < 8 || lda % 8 || ldb % 8 || ((uintptr_t)a_dev & 15) || ((uintptr_t)b_dev & 15) || ((uintptr_t)slabs_dev & 15) ||
        (64 * ldb + N) * 2 >= (1ll << 32) || (256 * lda + K) * 2 >= (1ll << 32)) {
        ecgb::set_error("ecgb_gemm_nn_splitk_bf16: K % 64, N % 8, 16-byte aligned operands with strides % 8 required");
        return ECGB_ERR_UNSUPPORTED;
    }
    GemmArgs G{};
    G.group_m = g_gemm_group_m;
    G.A = (const unsigned short *)a_dev; G.B = (const unsigned short *)b_dev; G.C = slabs_dev;
    G.M = M; G.N = N; G.K = K; G.lda = lda; G.ldb = ldb; G.ldc = N;
    G.batch_a = G.batch_b = G.batch_c = 0;
    G.accumulate_f32 = 3; G.alpha = alpha; G.split_stride = (long long)M * N;
    G.A2 = G.B2 = nullptr; G.lda2 = G.ldb2 = 0; G.K2 = 0;
    G.inner = 0; G.outer_a = G.inner_a = G.outer_b = G.inner_b = G.outer_c = G.inner_c = 0; G.div_a = G.div_b = 1;
    G.tiles_m = (M + 255) / 256; G.tiles_n = (N + 255) / 256;
    constexpr int lds = 2 * (256 + 256) * BK * 2;
    auto kern = gemm_nn_kernel_m16p<256, 256, 2, 4>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(kern, dim3((unsigned)(G.tiles_m * G.tiles_n), (unsigned)n_splits, 1), dim3(512), lds, (hipStream_t)stream, G);
        e = hipGetLastError();
    }
    if (e != hipSuccess) { ecgb::set_error(std::string("gemm_nn_kernel (K-slices): ") + hipGetErrorString(e)); return ECGB_ERR_HIP; }
    return ecgb_sum_slabs_bf16(slabs_dev, (long long)M * N, n_splits, c_dev, (size_t)M * N, 0, stream);
}

// d(gate|up) [M, 2 inter] = ecgb_glu_bwd(gate|up, dY . W) with W = the down projection as stored ([hidden, inter] row-major): ecgb_gemm_nn_bf16 with the
// GLU backward in its epilogue (the product [M, inter] is never written).  Whole 256x256 tiles only: M % 256 == 0, inter % 256 == 0.
extern "C" int ecgb_gemm_nn_glu_bwd_bf16(const void *dy_dev, long long lddy, const void *w_dev, long long ldw, const void *gate_up_dev, long long ldgu,
                                         void *d_gate_up_dev, long long ldd, int M, int inter, int K, int gelu_tanh, void *stream)
{
    if (!dy_dev || !w_dev || !gate_up_dev || !d_gate_up_dev || M <= 0 || inter <= 0 || K <= 0) { ecgb::set_error("ecgb_gemm_nn_glu_bwd_bf16: bad argument"); return ECGB_ERR_INVALID; }
    if (K % BK || M % 256 || inter % 256 || lddy % 8 || ldw % 8 || ldgu % 8 || ldd % 8 || ((uintptr_t)dy_dev & 15) || ((uintptr_t)w_dev & 15) ||
        ((uintptr_t)gate_up_dev & 15) || ((uintptr_t)d_gate_up_dev & 15) || (64 * ldw + inter) * 2 >= (1ll << 32) || (256 * lddy + K) * 2 >= (1ll << 32)) {
        ecgb::set_error("ecgb_gemm_nn_glu_bwd_bf16: K % 64, M % 256, inter % 256, 16-byte aligned operands with strides % 8 required");
        return ECGB_ERR_UNSUPPORTED;
    }
    // whole tiles and a long enough contraction per workgroup: the four-wave kernel on the NN layout with the same epilogue (the same bits; round 4).  Persistent: not
    // while a gradient exchange may hold CUs (g_nn_persist).
    if (g_gemm_w4 && g_nn_persist && g_gemm_tile == 0 && ecgb::gemm_w4_span_ok(1, lddy, ldw, K, 0) && ecgb::gemm_w4_applies(dy_dev, lddy, w_dev, ldw, d_gate_up_dev, ldd, M, inter, K))
        return ecgb::gemm_w4_launch(dy_dev, lddy, w_dev, ldw, d_gate_up_dev, ldd, M, inter, K, 1.0f, stream, gelu_tanh ? 5 : 4, nullptr, 0, nullptr, 0, nullptr, 0, 0, nullptr, nullptr, 0, 1,
                                    gate_up_dev, ldgu);
    GemmArgs G{};
    G.group_m = g_gemm_group_m;
    G.A = (const unsigned short *)dy_dev; G.B = (const unsigned short *)w_dev; G.C = d_gate_up_dev;
    G.M = M; G.N = inter; G.K = K; G.lda = lddy; G.ldb = ldw; G.ldc = ldd;
    G.batch_a = G.batch_b = G.batch_c = 0;
    G.accumulate_f32 = 0; G.alpha = 1.0f;
    G.A2 = G.B2 = nullptr; G.lda2 = G.ldb2 = 0; G.K2 = 0;
    G.inner = 0; G.outer_a = G.inner_a = G.outer_b = G.inner_b = G.outer_c = G.inner_c = 0; G.div_a = G.div_b = 1;
    G.tiles_m = M / 256; G.tiles_n = inter / 256;
    G.glu_I = inter; G.GU = (const unsigned short *)gate_up_dev; G.ldgu = ldgu;
    constexpr int lds = 2 * (256 + 256) * BK * 2;
    using Kern = void (*)(GemmArgs);
    Kern kern = gelu_tanh ? (Kern)gemm_nn_kernel_m16p<256, 256, 2, 4, 2> : (Kern)gemm_nn_kernel_m16p<256, 256, 2, 4, 1>;
    unsigned grid = (unsigned)(G.tiles_m * G.tiles_n);
    if (g_nn_persist && persist_ok(G, 1, false) && (ldgu & 7) == 0) { kern = gelu_tanh ? (Kern)gemm_nn_kernel_m16pp<2> : (Kern)gemm_nn_kernel_m16pp<1>; grid = (unsigned)n_cus(); }
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(kern, dim3(grid, 1, 1), dim3(512), lds, (hipStream_t)stream, G);
        e = hipGetLastError();
    }
    if (e != hipSuccess) { ecgb::set_error(std::string("gemm_nn_kernel (GLU backward): ") + hipGetErrorString(e)); return ECGB_ERR_HIP; }
    return ECGB_OK;
}

// The same on the down-projection site of a LoRA fine-tune: the adapter's share of the input gradient joins the product before the GLU backward,
//   d(gate|up) = glu_bwd(gate|up, bf16(dY . W) + scale / (1 - p) * mask . (dt A)),   dt [M, 64] = dY . B_lora, A^T [inter, 64] (rank 16 in columns 0..15),
// the forward's dropout mask replayed from (seed, p) as ecgb_lora_down drew it: ecgb_gemm_nn_bf16 + ecgb_lora_dx_glu in one launch, the same bits, d(act(gate) * up)
// never written.  Four-wave kernel only: ECGB_ERR_UNSUPPORTED where it does not take the shape (the caller runs the two kernels).
extern "C" int ecgb_gemm_nn_glu_bwd_lora_bf16(const void *dy_dev, long long lddy, const void *w_dev, long long ldw, const void *gate_up_dev, long long ldgu,
                                              const void *dt_dev, const void *at_dev, void *d_gate_up_dev, long long ldd, int M, int inter, int K, int gelu_tanh,
                                              float scale, float p, uint64_t seed, void *stream)
{
    if (!dy_dev || !w_dev || !gate_up_dev || !dt_dev || !at_dev || !d_gate_up_dev || M <= 0 || inter <= 0 || K <= 0 || !(p >= 0.f && p < 1.f)) {
        ecgb::set_error("ecgb_gemm_nn_glu_bwd_lora_bf16: bad argument");
        return ECGB_ERR_INVALID;
    }
    if (K % BK || M % 256 || inter % 256 || lddy % 8 || ldw % 8 || ldgu % 8 || ldd % 8 || ((uintptr_t)dy_dev & 15) || ((uintptr_t)w_dev & 15) || ((uintptr_t)gate_up_dev & 15) ||
        ((uintptr_t)d_gate_up_dev & 15) || ((uintptr_t)dt_dev & 15) || ((uintptr_t)at_dev & 15) || (long long)M * inter > 0xFFFFFFFFll ||
        !(g_gemm_w4 && g_nn_persist && g_gemm_tile == 0 && ecgb::gemm_w4_span_ok(1, lddy, ldw, K, 0) && ecgb::gemm_w4_applies(dy_dev, lddy, w_dev, ldw, d_gate_up_dev, ldd, M, inter, K))) {
        ecgb::set_error("ecgb_gemm_nn_glu_bwd_lora_bf16: whole 256x256 tiles with at least the four-wave kernel's share of K-tiles per CU, 16-byte aligned operands");
        return ECGB_ERR_UNSUPPORTED;
    }
    const unsigned thr = (unsigned)(p * 65536.0f);                                   // as ecgb_lora_down / ecgb_lora_dx fill them (lora.hip)
    const float lscale = scale / (1.0f - (float)thr / 65536.0f);
    return ecgb::gemm_w4_launch(dy_dev, lddy, w_dev, ldw, d_gate_up_dev, ldd, M, inter, K, 1.0f, stream, gelu_tanh ? 7 : 6, nullptr, 0, nullptr, 0, nullptr, 0, 0, nullptr, nullptr, 0, 1,
                                gate_up_dev, ldgu, dt_dev, at_dev, lscale, thr, (unsigned)(seed ^ (seed >> 32)));
}

// The input gradient of a frozen projection with ONE LoRA module on it (o): dx = bf16(dY . W) + scale / (1 - p) * mask . (dt A) in one launch -- ecgb_gemm_nn_bf16 followed
// by ecgb_lora_dx (one block), the same bits, no read-modify-write pass over dx.  Four-wave kernel only: ECGB_ERR_UNSUPPORTED where it does not take the shape.
extern "C" int ecgb_gemm_nn_lora_bf16(const void *dy_dev, long long lddy, const void *w_dev, long long ldw, const void *dt_dev, const void *at_dev,
                                      void *dx_dev, long long lddx, int M, int in, int K, float scale, float p, uint64_t seed, void *stream)
{
    if (!dy_dev || !w_dev || !dt_dev || !at_dev || !dx_dev || M <= 0 || in <= 0 || K <= 0 || !(p >= 0.f && p < 1.f)) {
        ecgb::set_error("ecgb_gemm_nn_lora_bf16: bad argument");
        return ECGB_ERR_INVALID;
    }
    if (K % BK || M % 256 || in % 256 || lddy % 8 || ldw % 8 || lddx % 8 || ((uintptr_t)dy_dev & 15) || ((uintptr_t)w_dev & 15) || ((uintptr_t)dx_dev & 15) ||
        ((uintptr_t)dt_dev & 15) || ((uintptr_t)at_dev & 15) || (long long)M * in > 0xFFFFFFFFll ||
        !(g_gemm_w4 && g_nn_persist && g_gemm_tile == 0 && ecgb::gemm_w4_span_ok(1, lddy, ldw, K, 0) && ecgb::gemm_w4_applies(dy_dev, lddy, w_dev, ldw, dx_dev, lddx, M, in, K))) {
        ecgb::set_error("ecgb_gemm_nn_lora_bf16: whole 256x256 tiles with at least the four-wave kernel's share of K-tiles per CU, 16-byte aligned operands");
        return ECGB_ERR_UNSUPPORTED;
    }
    const unsigned thr = (unsigned)(p * 65536.0f);
    const float lscale = scale / (1.0f - (float)thr / 65536.0f);
    return ecgb::gemm_w4_launch(dy_dev, lddy, w_dev, ldw, dx_dev, lddx, M, in, K, 1.0f, stream, 8, nullptr, 0, nullptr, 0, nullptr, 0, 0, nullptr, nullptr, 0, 1,
                                nullptr, 0, dt_dev, at_dev, lscale, thr, (unsigned)(seed ^ (seed >> 32)));
}

#ifdef ECGB_PROFILE
extern "C" void ecgb_debug_gemm_profile(unsigned long long *out8, int reset)   // out8: 12 counters
{
    if (out8) (void)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_gemm_prof), 12 * sizeof(unsigned long long));
    if (reset) { unsigned long long z[12] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_prof), z, sizeof z); }
}
#endif
