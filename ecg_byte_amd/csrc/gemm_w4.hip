// C[M, N] = alpha * A[M, K] . B[N, K]^T (bf16 in, fp32 accumulate, bf16 out) on FOUR waves per workgroup, each owning a 128x128 quarter of a
// 256x256 tile -- one wave per SIMD, its 256 accumulators in the accumulation half of the register file.
//
// Why a second NT kernel (round 3).  The 8-wave kernels of gemm.hip (wave tile 128x64, two waves per SIMD) keep the MFMA pipes 0.61-0.70 busy at about
// 1.7 GHz; hipBLASLt's kernel for the same shapes (MT256x256x64, MIWT8_8, 256 threads; `scripts/prof_blas.sh`, `scripts/prof_gemm_pmc.sh`) keeps them 0.91
// busy at 1.65 GHz -- the clock does not pay for the idle cycles, so the gap is schedule, not power.  A 128x128 wave tile reads 32 LDS fragments per 128 MFMAs
// where a 128x64 one reads 24 per 64: a third less LDS traffic per FLOP, half the waves' worth of address arithmetic, ONE workgroup barrier per K-tile
// instead of eight.  Measured on the forward shapes of the C3 step (`scripts/dev_gemm_w4.py`): 1-6 % faster than the 8-wave kernels, 5-12 % behind
// hipBLASLt; what is left is the one rendezvous per K-tile (the four waves wait for the slowest wave's LDS-DMA: 8 % at K = 8192) -- see DESIGN.md section 7.
//
// This file is compiled WITHOUT -amdgpu-mfma-vgpr-form (Makefile): 256 accumulators + 128 fragment registers need both halves of the file.
//
// Persistent: one workgroup per CU, each walks its share of the tiles as ONE flat sequence of K-tiles, so the first two K-tiles of the next output tile are
// already in flight (and its first fragments in registers) while the finished tile is stored.  Per K-tile and wave: 16 LDS-DMA pieces (global_load_lds,
// 1 KiB each), 32 ds_read_b128, 128 MFMA 16x16x32, one s_barrier.  Same MFMA, same order over K as the 8-wave kernels: the same bits
// (tests/test_gpu_decoder_ops.py::test_gemm_nt_four_wave_kernel_same_bits).
#include <hip/hip_runtime.h>

#include <string>

#include "glu_math.hpp"
#include "tokenizer.hpp"

namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int BK = 64;
constexpr int kTileBytes = 256 * BK * 2;                 // one operand's K-tile: 256 rows of 128 bytes
constexpr int kBufBytes = 2 * kTileBytes;                // A then B
constexpr int kStageBytes = 6144;                        // per wave: one MFMA row of the finished tile on its way out (16 x 128 bf16; GLU: gate, up and act(gate) * up, 16 x 64 each)
constexpr int kLdsBytes = 2 * kBufBytes + 4 * kStageBytes;

struct W4Args {
    const unsigned short *A, *B;
    unsigned short *C;
    long long lda, ldb, ldc;
    int M, N, K;
    int tiles_m, tiles_n;
    int group_m;            // tile order: 0 / 1 row by row; g: blocks of g tile rows, column by column inside a block
    float alpha;
    const unsigned short *A2, *B2;   // CAT: a second operand pair contracted behind the first, C = alpha * (A B^T + A2 B2^T) (the LoRA branch: A2 = x A_lora^T, B2 = B_lora)
    long long lda2, ldb2;
    int K2;
    const float *rope_cos, *rope_sin;   // EPI 3: columns below rope_cols are heads of 64 that leave rotated (RoPE forward), row t with row t of the [M, 32] fp32 tables
    int rope_cols;
    unsigned short *H;      // GLU epilogue (EPI 1 / 2): act(gate) * up [M, glu_I]; B is [2 * glu_I, K], gate rows then up rows; C (gate|up, [M, 2 * glu_I]) may be null
    long long ldh;
    int glu_I;
    const unsigned short *GU;   // GLU BACKWARD epilogue (EPI 4 / 5, NN layout): the forward's gate|up [M, 2 * glu_I]; the product [M, glu_I] is d(act(gate) * up) and
    long long ldgu;             // C [M, 2 * glu_I] receives d gate | d up (ecgb_glu_bwd's arithmetic on the bf16-rounded product)
    const unsigned short *LDT, *LAT;   // EPI 6 / 7: the down-projection site's LoRA adapter joins the product before the GLU backward -- dt [M, 64] = dY . B_lora and A_lora^T
    float lscale;                      // [glu_I, 64] (rank 16 in columns 0..15), alpha / r / (1 - p), and the dropout mask of the forward replayed from (lseed, lthr):
    unsigned lthr, lseed;              // ecgb_lora_dx_glu's arithmetic on the bf16-rounded product
};

__device__ __forceinline__ unsigned pack2(float a, float b)
{
    using bf2 = __attribute__((ext_vector_type(2))) __bf16;
    bf2 v;
    v[0] = (__bf16)a; v[1] = (__bf16)b;
    return __builtin_bit_cast(unsigned, v);
}

// Tile of workgroup w in round `it`: the workgroups of a round take consecutive tiles of the tile order, the 32 workgroups of an XCD (w % 8) a contiguous run
// of them -- with the order in blocks of 8 tile rows (tile_rc) an XCD's 32 concurrent tiles are an 8 x 4 patch: 12 panels of A and B for 32 tiles instead of
// 33 (measured, [32768, 16384] x K 2048: 2.07 ms row by row, 1.87 / 1.70 / 1.62 / 1.60 ms with blocks of 2 / 4 / 8 / 16 -- the LDS-DMA of a K-tile has one K-tile
// of time to land, and the rendezvous waits for the slowest of 64 pieces: it is the L2 hit rate that decides how long).  8, not 16: level with 16 on the NT shapes
// over three boxes, and the NN product [32768, 16384] . [16384, 2048] (8 tile columns) takes 1.63 ms with blocks of 0 .. 8 and 2.31 with 16.
__device__ __forceinline__ int tile_of_round(int w, int it, int nwg) { return it * nwg + (w & 7) * (nwg >> 3) + (w >> 3); }
__device__ __forceinline__ void tile_rc(const W4Args &G, int tile, int &tm, int &tn)
{
    if (G.group_m > 1) {
        const int per = G.group_m * G.tiles_n, blk = tile / per, in = tile - blk * per;
        const int rows = min(G.group_m, G.tiles_m - blk * G.group_m);           // the last block may be shorter
        tn = in / rows;
        tm = blk * G.group_m + (in - tn * rows);
    } else {
        tm = tile / G.tiles_n;
        tn = tile - tm * G.tiles_n;
    }
}

// EPI 0: plain bf16 store.  EPI 1 (SiLU) / 2 (tanh-GELU): the MLP's gate|up projection with the GLU in the epilogue, as gemm.hip's EPI kernels do it -- a tile's 256
// columns are 8 groups of 16 gate columns and the 16 up columns of the same outputs (the B rows are fetched in that order), MFMA tile (i, 2p) is gate and
// (i, 2p + 1) up of the same 16 columns; glu_fwd_kernel's arithmetic on the bf16-rounded projections: the same bits fused and unfused, four-wave and eight-wave.
// CAT: the K loop continues into (A2, B2) -- K-tiles KT1 .. KT - 1 of every output tile come from the second pair (another scalar base, another set of per-lane
// offsets chosen by a wave-uniform select: one v_cndmask per DMA piece).
// NN: B is [K, N] row-major (the input gradient dX = dY . W against the weight as nn.Linear stores it): its K-tile lies in LDS as 64 contraction rows of 512 bytes, staged
// two rows per DMA piece, and the MFMA fragments (eight consecutive contraction elements of one column) are gathered by gfx950's transposing LDS read -- the layout, swizzle
// and fragment addresses of gemm_nn_kernel_m16p (gemm.hip), through the compiler's builtin so that hipcc carries the waits.  Plain store, no second pair.
// LAY 2 (TN: the weight gradient dW = dY^T . X): A is [K, M] row-major as well -- both operands staged as contraction rows and gathered by transposing reads
// (gemm_tn_kernel_tr's layout).  LAY 0: NT, LAY 1: NN.
// SCH: the schedule of a K-tile.  0 (round 3): ONE rendezvous per K-tile -- `s_waitcnt vmcnt(0)` + barrier between the two k-slices, all sixteen DMA pieces of
// K-tile g + 2 issued under k-slice 1: a piece has 0.5 .. 1 K-tile of time to land and every wave waits for its own slowest piece before the barrier.
// 1 (round 4): FOUR barriers per K-tile, each behind a wait for something long done (the shape of the library kernel's main loop, read off its disassembly:
// DESIGN.md section 7) -- the B region of the current buffer is free once every wave has read its k-slice-1 B fragments (barrier 1, MFMA 21) and the DMA of
// K-tile g + 2's B pieces starts right there, the A region after barrier 2 (MFMA 52); K-tile g + 1's B pieces are waited for with a COUNTED `vmcnt(18)` at
// MFMA 68 (barrier 3, then its k-slice-0 B fragments are read), its A pieces with `vmcnt(15)` at MFMA 105 (barrier 4): a piece has 0.85 .. 1.35 K-tiles to land
// and no wave ever waits for a piece it issued in the same K-tile.  Same MFMAs in the same order: the same bits.
template <int EPI, bool CAT, int LAY = 0, int SCH = 1>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_nt_w4_kernel(W4Args G)
{
    constexpr bool NN = LAY >= 1, TA = LAY == 2;              // B, A stored with the contraction index as the row
    static_assert(!NN || ((EPI == 0 || EPI >= 4) && !CAT), "the NN / TN forms are the plain product (NN: or the GLU backward behind it)");
    static_assert(EPI <= 8, "EPI 0..8");
    static_assert(EPI < 4 || LAY == 1, "the GLU backward epilogue belongs to the down projection's input gradient: NN layout");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int wr = wave >> 1, wc = wave & 1;
    const int lm = lane & 15, lq = lane >> 4;
    const int nwg = gridDim.x, w = blockIdx.x;
    const int n_tiles = G.tiles_m * G.tiles_n;
    const int KT1 = G.K / BK, KT = CAT ? KT1 + G.K2 / BK : KT1;
    // my tiles: rounds it = 0, 1, ... while tile_of_round(w, it) < n_tiles
    int my_tiles = 0;
    for (int it = 0; tile_of_round(w, it, nwg) < n_tiles; ++it) ++my_tiles;
    if (my_tiles == 0) return;
    const int total = my_tiles * KT;                       // K-tiles of the flat sequence (the host keeps it under 2^31)

    // ---- LDS-DMA: piece p = wave * 8 + i (i = 0..7) of an operand's K-tile is rows p*8 .. p*8+7; lane L brings the 16-byte chunk
    // (L & 7) ^ ((row >> 1) & 7) of row p*8 + (L >> 3) -- the swizzle sits on the global side, the LDS side of the DMA is linear.
    // (row >> 1) & 7 = ((L >> 4) + 4 * (i & 1)) & 7.
    // A piece is `buffer_load_dwordx4 v_lane, s[descriptor], s_piece offen lds` with M0 = where it lands: the descriptor's base is the K-tile's (scalar, advanced once
    // per K-tile), the per-lane offset (row L >> 3 of the piece, swizzled chunk; odd pieces: 8 rows further and the other swizzle) is fixed while the operand pair is,
    // and the scalar offset moves to piece pair i >> 1 -- no address arithmetic per piece, so a piece is TWO instructions (`s_add_u32 m0, m0, 1024` and the load), and
    // the schedule below puts each into an MFMA gap of its own.  (Round 3 wrote `s_mov m0; s_nop; global_load_lds v, s[base]` with the base formed per piece: five to six
    // scalar instructions and the load in ONE gap -- with one wave per SIMD every instruction of the wave takes its 4-5 issue cycles out of the 16 an MFMA gives, and the
    // matrix pipe idled behind each piece: MFMA busy 0.79 against 0.93 for the same loop without the pieces, 0.90 for the library kernel with them.)
    // Operands stored as contraction rows (NN's B, TN's A and B): piece i = contraction rows 2i, 2i + 1 of the wave's 16, eight per-lane offsets (the swizzle moves with the row).
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
    unsigned voffA[TA ? 8 : 2], voffB[NN ? 8 : 2];
    // (scalars) byte offset of piece pair q = i >> 1 from the descriptor's base, INCLUDING the K-tile's offset inside the operand: the descriptor is set once per
    // output tile (and per operand pair), a K-tile further is `s_add_u32 soff, soff, step` on each of these -- in place, one per MFMA gap, behind the offset's last use
    // (a descriptor advanced per K-tile cost a dozen scalar copies at the loop's tail: the loop-carried values of every path meet there).  Contraction-row operands
    // use entry 0 for all their pieces.
    unsigned ksA[TA ? 1 : 4], ksB[NN ? 1 : 4];
    unsigned pairB1 = 0, pairB2 = 0, pairB3 = 0, stepA2 = 0;     // what the entries are at K-tile 0 of the pair being staged (A: q * stepA2)
    auto lane_offsets = [&](long long lda_, long long ldb_) {
        // recomputed from the lane number wherever the staged pair changes (twice per output tile with a second pair), through an opaque copy: left visible, hipcc
        // kept both pairs' offsets alive across the K loop, spilled them, and its reloads at the loop's tail waited vmcnt(0) -- every DMA piece in flight -- per K-tile
        unsigned ln = (unsigned)lane;
        asm volatile("" : "+v"(ln));
        const unsigned lrow = ln >> 3, ch0 = ((ln & 7) ^ ((ln >> 4) & 7)) << 4, ch1 = ch0 ^ 64u;     // chunk (L & 7) ^ ((L >> 4) + 4 (i & 1)) & 7, times 16 bytes
        if constexpr (!TA) {
            const unsigned step = (unsigned)(lda_ * 16);                      // 8 rows
            voffA[0] = lrow * (unsigned)(lda_ * 2) + ch0; voffA[1] = lrow * (unsigned)(lda_ * 2) + step + ch1;
            stepA2 = 2 * step;
        }
        if constexpr (!NN) {
            const unsigned step = (unsigned)(ldb_ * 16);
            voffB[0] = lrow * (unsigned)(ldb_ * 2) + ch0; voffB[1] = lrow * (unsigned)(ldb_ * 2) + step + ch1;
            if constexpr (EPI == 0 || EPI >= 3) { pairB1 = 2 * step; pairB2 = 4 * step; pairB3 = 6 * step; }
            else {
                // GLU: tile row r = (wave * 8 + i) * 8 + (L >> 3) is weight row 16 * (r >> 5) + (r & 15) (+ glu_I: an up row) -- pieces 0, 1: gate rows 0 .. 15 of the
                // wave's 32, pieces 2, 3 the up rows of the same outputs, 4, 5 gate rows 16 .. 31, 6, 7 their up rows
                const unsigned glu = (unsigned)((long long)G.glu_I * ldb_ * 2);
                pairB1 = glu; pairB2 = 2 * step; pairB3 = 2 * step + glu;
            }
        }
    };
    // K-tile kt of the pair being staged
    auto soff_at = [&](unsigned kt, long long lda_, long long ldb_) {
        const unsigned ka = TA ? kt * (unsigned)(BK * lda_ * 2) : kt * (unsigned)(BK * 2), kb = NN ? kt * (unsigned)(BK * ldb_ * 2) : kt * (unsigned)(BK * 2);
        // (readfirstlane: wave-uniform by construction, but with a second operand pair hipcc carried these through vector registers and handed one to the
        // load's scalar-offset operand; this runs once per output tile and operand pair)
        auto u = [](unsigned x) { return (unsigned)__builtin_amdgcn_readfirstlane((int)x); };
        if constexpr (TA) ksA[0] = u(ka); else { ksA[0] = u(ka); ksA[TA ? 0 : 1] = u(ka + stepA2); ksA[TA ? 0 : 2] = u(ka + 2 * stepA2); ksA[TA ? 0 : 3] = u(ka + 3 * stepA2); }
        if constexpr (NN) ksB[0] = u(kb); else { ksB[0] = u(kb); ksB[NN ? 0 : 1] = u(kb + pairB1); ksB[NN ? 0 : 2] = u(kb + pairB2); ksB[NN ? 0 : 3] = u(kb + pairB3); }
    };
    lane_offsets(G.lda, G.ldb);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if constexpr (TA) {
            const int rk = i * 2 + (lane >> 5), rkt = wave * 16 + rk;
            voffA[i] = (unsigned)(((long long)rk * G.lda + ((lane & 31) ^ (((rkt & 7) << 1) ^ (rkt & 8))) * 8) * 2);
        }
        if constexpr (NN) {                                  // piece i: contraction rows rk, rk + 1 of the tile (512 bytes each), 16-byte chunk c of row rk at c ^ ((rk & 7) << 1 ^ (rk & 8))
            const int rk = i * 2 + (lane >> 5), rkt = wave * 16 + rk;
            const int chunkb = (lane & 31) ^ (((rkt & 7) << 1) ^ (rkt & 8));
            voffB[i] = (unsigned)(((long long)rk * G.ldb + chunkb * 8) * 2);
        }
    }
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds;
    const unsigned dma_base = lds_base + (unsigned)wave * 8192u;                 // this wave's eight pieces of an operand tile
    // the K-tile the next stage brings: (round, K-tile) of the flat sequence, the descriptors of its operands and the scalar offsets above.  Past the end of the
    // sequence they stay on the last K-tile: the loop body stages unconditionally (no branch inside the interleaved stretch), a repeated tile lands in a buffer
    // nobody reads again.
    int st_kt = 0, st_it = 0;
    u32x4 srdA = {0u, 0u, 0xFFFFFFFFu, 0x00020000u}, srdB = {0u, 0u, 0xFFFFFFFFu, 0x00020000u};     // raw buffers: base, no stride, no bound, 32-bit data format
    const unsigned char *st_a2 = nullptr, *st_b2 = nullptr;
    auto set_base = [](u32x4 &srd, const unsigned char *p) {
        const unsigned long long a = (unsigned long long)(uintptr_t)p;
        srd[0] = (unsigned)a; srd[1] = (unsigned)(a >> 32) & 0xFFFFu;
    };
    const unsigned kstepA = TA ? (unsigned)(BK * G.lda * 2) : (unsigned)(BK * 2), kstepB = NN ? (unsigned)(BK * G.ldb * 2) : (unsigned)(BK * 2);
    auto stage_first = [&]() {                                                  // start of round st_it's tile
        int tm, tn;
        tile_rc(G, tile_of_round(w, st_it, nwg), tm, tn);
        const long long ra = (long long)tm * 256 + wave * 64, rb = (EPI == 0 || EPI >= 3) ? (long long)tn * 256 + wave * 64 : (long long)tn * 128 + wave * 32;
        set_base(srdA, TA ? reinterpret_cast<const unsigned char *>(G.A + (long long)wave * 16 * G.lda + (long long)tm * 256)
                          : reinterpret_cast<const unsigned char *>(G.A + ra * G.lda));
        set_base(srdB, NN ? reinterpret_cast<const unsigned char *>(G.B + (long long)wave * 16 * G.ldb + (long long)tn * 256)       // this wave's 16 contraction rows of K-tile 0, the tile's columns
                          : reinterpret_cast<const unsigned char *>(G.B + rb * G.ldb));
        if constexpr (CAT) {
            st_a2 = reinterpret_cast<const unsigned char *>(G.A2 + ra * G.lda2);
            st_b2 = reinterpret_cast<const unsigned char *>(G.B2 + rb * G.ldb2);
            lane_offsets(G.lda, G.ldb);                                          // (back on the first pair)
        }
        soff_at(0u, G.lda, G.ldb);
    };
    // What the loop's tail does for the K-tile after the one just staged.  The K loop itself has already moved the scalar offsets one K-tile on (`dma_step_*`): nothing
    // is left to do unless the output tile or the operand pair changes, or the sequence has ended (the offsets go back: the last K-tile is staged again).
    auto stage_advance = [&](bool more) {
        if (!more) { soff_at((unsigned)(CAT && st_kt >= KT1 ? st_kt - KT1 : st_kt), CAT && st_kt >= KT1 ? G.lda2 : G.lda, CAT && st_kt >= KT1 ? G.ldb2 : G.ldb); return; }
        if (++st_kt == KT) {
            st_kt = 0;
            ++st_it;
            stage_first();
        } else if (CAT && st_kt == KT1) {
            set_base(srdA, st_a2); set_base(srdB, st_b2);
            if constexpr (CAT) { lane_offsets(G.lda2, G.ldb2); soff_at(0u, G.lda2, G.ldb2); }     // the second pair's strides: a wave-uniform branch once per tile
        }
    };
    // M0: `dma_dst` points it at this wave's first piece of an operand tile, `dma_next` moves it one piece on; each is one scalar instruction, and at least one other
    // instruction (an MFMA) stands between a write of M0 and the load that uses it
    constexpr bool NO_DMA = ((SCH >> 4) & 4) != 0;
    auto dma_dst_a = [&](unsigned buf_off) { if constexpr (!NO_DMA) asm volatile("s_mov_b32 m0, %0" :: "s"(dma_base + buf_off) : "memory"); };
    auto dma_dst_b = [&](unsigned buf_off) { if constexpr (!NO_DMA) asm volatile("s_mov_b32 m0, %0" :: "s"(dma_base + buf_off + (unsigned)kTileBytes) : "memory"); };
    auto dma_next = [&]() { if constexpr (!NO_DMA) asm volatile("s_add_u32 m0, m0, 0x400" ::: "memory", "scc"); };
    auto ld_a = [&](int i) {
        if constexpr (!NO_DMA) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" :: "v"(TA ? voffA[TA ? i : 0] : voffA[i & 1]), "s"(srdA), "s"(ksA[TA ? 0 : (i >> 1)]) : "memory");
    };
    auto ld_b = [&](int i) {
        if constexpr (!NO_DMA) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" :: "v"(NN ? voffB[NN ? i : 0] : voffB[i & 1]), "s"(srdB), "s"(ksB[NN ? 0 : (i >> 1)]) : "memory");
    };
    // one K-tile on (q: piece pair; contraction-row operands have the one entry).  Plain scalar adds, written behind the offset's last use in the schedule: as asm
    // with an in/out scalar operand hipcc's divergence analysis takes the result for a per-lane value ("illegal VGPR to SGPR copy")
    auto dma_step_a = [&](int q) {
        if constexpr (TA) { if (q == 3) ksA[0] += kstepA; }            // (the one entry serves all eight pieces: behind the last)
        else ksA[TA ? 0 : q] += (unsigned)(BK * 2);
    };
    auto dma_step_b = [&](int q) {
        if constexpr (NN) { if (q == 3) ksB[0] += kstepB; }
        else ksB[NN ? 0 : q] += (unsigned)(BK * 2);
    };
    // (prologue, and the round-3 schedule: a whole piece at once)
    auto dma_a = [&](unsigned buf_off, int i) {
        if constexpr (NO_DMA) return;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" :: "s"(dma_base + buf_off + (unsigned)i * 1024u) : "memory");
        ld_a(i);
    };
    auto dma_b = [&](unsigned buf_off, int i) {
        if constexpr (NO_DMA) return;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" :: "s"(dma_base + buf_off + (unsigned)kTileBytes + (unsigned)i * 1024u) : "memory");
        ld_b(i);
    };
    auto dma_step_all = [&]() {
#pragma unroll
        for (int q = 0; q < 4; ++q) { dma_step_a(q); dma_step_b(q); }
    };

    // ---- fragments: MFMA 16x16x32 operand = 16 rows x 32 k; lane (lm, lq) reads the 16-byte chunk ks*4 + lq of row (tile row) * 16 + lm.
    // (row >> 1) & 7 = (lm >> 1) & 7 for every row this lane reads: one swizzled offset per k-slice.
    const int sw = (lm >> 1) & 7;
    const unsigned fragA = (unsigned)((wr * 128 + lm) * 128 + ((lq ^ sw) << 4));
    const unsigned fragB = (unsigned)(kTileBytes + (wc * 128 + lm) * 128 + ((lq ^ sw) << 4));
    // NN / TN: a lane's transposing reads of the fragment at columns 16 m .. of a tile stored as contraction rows (k-slice 0; k-slice 1 is 32 rows further): lane
    // (16 g + 4 q + p) addresses row 8 g + q, chunk 2 m + (p >> 1) of the row, stored at chunk ^ 2 s with s = (row & 7) ^ ((row & 8) >> 1); the second read, 4 rows
    // down, has s ^ 4: the entry of fragment m ^ 4, 2048 bytes further (gemm_tn_kernel_tr has the derivation).
    unsigned tabA[TA ? 8 : 1], tabB[NN ? 8 : 1];
    if constexpr (NN) {
        const int tq = lm >> 2, tp = lm & 3, rbk = 8 * lq + tq, swb = (rbk & 7) ^ ((rbk & 8) >> 1), cb = tp >> 1;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            tabB[m] = (unsigned)(kTileBytes + rbk * 512 + (tp & 1) * 8 + wc * 256 + (((2 * (m ^ swb)) + cb) << 4));
            if constexpr (TA) tabA[m] = (unsigned)(rbk * 512 + (tp & 1) * 8 + wr * 256 + (((2 * (m ^ swb)) + cb) << 4));
        }
    }
    using s4 = __attribute__((ext_vector_type(4))) short;
    using s8 = __attribute__((ext_vector_type(8))) short;
    auto tr_frag = [&](const unsigned char *buf, unsigned t0, unsigned t1, int ks) {
        const unsigned char *p0 = buf + t0 + (ks ? 32 * 512 : 0), *p1 = buf + t1 + 2048 + (ks ? 32 * 512 : 0);
        const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4 *)(size_t)(unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char *)p0);
        const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4 *)(size_t)(unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char *)p1);
        const s8 f = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, f);
    };
    auto frag_a = [&](const unsigned char *buf, int ks, int i) {
        if constexpr (TA) return tr_frag(buf, tabA[i], tabA[i ^ 4], ks);
        else return *reinterpret_cast<const bf16x8 *>(buf + (fragA ^ (ks ? 64u : 0u)) + i * 2048);
    };
    auto frag_b = [&](const unsigned char *buf, int ks, int j) {
        if constexpr (NN) return tr_frag(buf, tabB[j], tabB[j ^ 4], ks);
        else return *reinterpret_cast<const bf16x8 *>(buf + (fragB ^ (ks ? 64u : 0u)) + j * 2048);
    };

    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // (asm: the accumulators stay in a0..a255 by constraint, the fragments in architectural registers -- as builtins hipcc put fragments into the accumulation
    // file and spilled 176 registers; the 64 MFMAs between two uses of one accumulator cover the dependent-issue distance the assembler does not see.  The empty
    // asm with a memory clobber pins the loads between the MFMAs where they are written.)
#define W4_MFMA(i, j, fa, fb) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(fb[j]), "v"(fa[i]))
#define W4_FENCE() asm volatile("" ::: "memory")

    // prologue: K-tiles 0 and 1 of the sequence in flight, 0 complete
    stage_first();
#pragma unroll
    for (int i = 0; i < 8; ++i) dma_b(0u, i);
#pragma unroll
    for (int i = 0; i < 8; ++i) dma_a(0u, i);
    if (total > 1) {
        dma_step_all(); stage_advance(true);
        // (B pieces, then A pieces: the issue order of every K-tile -- SCH 1's counted waits rely on it)
#pragma unroll
        for (int i = 0; i < 8; ++i) dma_b((unsigned)kBufBytes, i);
#pragma unroll
        for (int i = 0; i < 8; ++i) dma_a((unsigned)kBufBytes, i);
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (total > 2) { dma_step_all(); stage_advance(true); }
    __builtin_amdgcn_s_barrier();
    bf16x8 fa0[8], fb0[8], fa1[8], fb1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) fb0[j] = frag_b(lds, 0, j);
#pragma unroll
    for (int i = 0; i < 8; ++i) fa0[i] = frag_a(lds, 0, i);

    int g = 0;
    for (int it = 0; it < my_tiles; ++it) {
        // (two nested loops, not one flat one with the store under a test: with the store inside the K loop hipcc spilled the loop-carried fragments on every K-tile)
        for (int kt = 0; kt < KT; ++kt, ++g) {
            unsigned char *cur = lds + (g & 1) * kBufBytes, *nxt = lds + ((g + 1) & 1) * kBufBytes;
            const unsigned cur_off = (unsigned)(g & 1) * (unsigned)kBufBytes;
            if constexpr ((SCH & 15) == 1) {
                // 128 MFMAs, m = 0 .. 127: k-slice 0 (fa0, fb0) then k-slice 1 (fa1, fb1), row i of A fragments outer, B fragment j inner; what is issued behind MFMA m:
                //   1, 3 .. 15      fb1[0 .. 7]   <- cur (k-slice 1)                      20 / 21   lgkmcnt(0) / barrier 1: cur's B region is free
                //   23, 26 .. 35    DMA B pieces 0 .. 4 of K-tile g + 2 -> cur            25, 28, 31, 34, 37, 39, 41, 43   fa1[0 .. 7] <- cur
                //   (M0 is pointed at an operand's first piece at 22 / 60 and moved on behind every load, each in a gap of its own)
                //   51 / 52         lgkmcnt(0) / barrier 2: cur's A region is free        53, 56, 59   DMA B 5 .. 7      62, 65   DMA A 0, 1
                //   68 / 69         vmcnt(18) / barrier 3: K-tile g + 1's B has landed    70, 72 .. 84   fb0[0 .. 7] <- nxt (k-slice 0)
                //   86, 88, 90, 97, 101   DMA A 2 .. 6                                    105 / 106   vmcnt(15) / barrier 4: K-tile g + 1's A has landed
                //   107, 109 .. 121 fa0[0 .. 7] <- nxt                                    125   DMA A 7
                constexpr int DBG = SCH >> 4;      // timing-only builds (wrong results): 1 no barriers, 2 no DMA waits, 4 no DMA
                auto after = [&](const int m) __attribute__((always_inline)) {
                    W4_FENCE();
                    if (m <= 15 && (m & 1)) fb1[m >> 1] = frag_b(cur, 1, m >> 1);
                    else if (m == 20 || m == 51) __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0) -- as a builtin: hipcc's own wait insertion sees it and leaves out
                                                                                                //  the per-row waits it would otherwise put before every first use of a fragment
                    else if (m == 21 || m == 52 || m == 69 || m == 106) { if constexpr (!(DBG & 1)) __builtin_amdgcn_s_barrier(); }
                    else if (m == 22) dma_dst_b(cur_off);
                    else if (m == 23 || m == 26 || m == 29 || m == 32 || m == 35) ld_b((m - 23) / 3);
                    else if (m == 24 || m == 27 || m == 30 || m == 33 || m == 36 || m == 54 || m == 57) dma_next();
                    else if (m == 25 || m == 28 || m == 31 || m == 34) fa1[(m - 25) / 3] = frag_a(cur, 1, (m - 25) / 3);
                    else if (m == 37 || m == 39 || m == 41 || m == 43) fa1[4 + (m - 37) / 2] = frag_a(cur, 1, 4 + (m - 37) / 2);
                    else if (m == 53 || m == 56 || m == 59) ld_b(5 + (m - 53) / 3);
                    else if (m == 60) dma_dst_a(cur_off);
                    else if (m == 62 || m == 65) ld_a((m - 62) / 3);
                    else if (m == 63 || m == 66 || m == 87 || m == 89 || m == 95 || m == 99 || m == 103) dma_next();
                    else if (m == 68) { if constexpr (!(DBG & 2)) asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); }
                    else if (m >= 70 && m <= 84 && !(m & 1)) fb0[(m - 70) >> 1] = frag_b(nxt, 0, (m - 70) >> 1);
                    else if (m == 86 || m == 88 || m == 90) ld_a(2 + (m - 86) / 2);
                    else if (m == 97 || m == 101) ld_a(5 + (m - 97) / 4);
                    else if (m == 105) { if constexpr (!(DBG & 2)) asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); }
                    else if (m >= 107 && m <= 121 && (m & 1)) fa0[(m - 107) >> 1] = frag_a(nxt, 0, (m - 107) >> 1);
                    else if (m == 125) ld_a(7);
                    else if (m == 38 || m == 40 || m == 55 || m == 61) dma_step_b(m == 38 ? 0 : m == 40 ? 1 : m == 55 ? 2 : 3);      // behind the offset's last use
                    else if (m == 67 || m == 91 || m == 98 || m == 126) dma_step_a(m == 67 ? 0 : m == 91 ? 1 : m == 98 ? 2 : 3);
                    W4_FENCE();
                };
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j) { W4_MFMA(i, j, fa0, fb0); after(i * 8 + j); }
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j) { W4_MFMA(i, j, fa1, fb1); after(64 + i * 8 + j); }
            } else {
            // ---- k-slice 0 of K-tile g (fragments read a block ago), the fragments of k-slice 1 read underneath: one ds_read per four MFMAs
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    W4_MFMA(i, j, fa0, fb0);
                    if ((j & 3) == 3) {
                        const int r = i * 2 + (j >> 2);                       // 0..15: B fragments first (the first row of MFMAs needs them all)
                        W4_FENCE();
                        if (r < 8) fb1[r] = frag_b(cur, 1, r); else fa1[r - 8] = frag_a(cur, 1, r - 8);
                        W4_FENCE();
                    }
                }
            // every read of `cur` by this wave has returned; this wave's pieces of K-tile g + 1 have landed; then everyone's
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            // ---- k-slice 1; underneath, K-tile g + 2 starts into `cur` (16 pieces) and the fragments of K-tile g + 1's k-slice 0 are read (the last K-tile of a
            // tile reads the first fragments of the next tile): one of the 32 per two MFMAs, DMA pieces and reads alternating (all DMA first: 5-9 % slower)
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    W4_MFMA(i, j, fa1, fb1);
                    if (j & 1) {
                        const int slot = i * 4 + (j >> 1), r = slot >> 1;     // 32 slots: even = fragment read r, odd = DMA piece r (0..15)
                        W4_FENCE();
                        if (slot & 1) { if (r < 8) dma_a(cur_off, r); else dma_b(cur_off, r - 8); }
                        else { if (r < 8) fb0[r] = frag_b(nxt, 0, r); else fa0[r - 8] = frag_a(nxt, 0, r - 8); }
                        W4_FENCE();
                    }
                }
            }
            if constexpr ((SCH & 15) != 1) dma_step_all();
            stage_advance(g + 3 < total);
        }
        // ---- the tile is complete (the next tile's first two K-tiles are in flight, its first fragments in registers).  The wave's 128x128 block leaves through
        // 4 KiB of LDS of its own, one MFMA row of 16 x 128 at a time: a lane holds four consecutive columns of sixteen different rows, memory wants whole rows --
        // 8-byte writes in (row r at r * 256 bytes, its 16-byte chunk c at c ^ r: the writes of a 16 x 16 tile and the row reads both run at full LDS width), 16-byte
        // reads out, each store instruction four full 256-byte rows.  (Direct 8-byte stores from the accumulator layout: 12 us per tile, a quarter of a K = 2048 tile.)
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");                    // the last MFMAs' results before the accumulators are read
        int tm, tn;
        tile_rc(G, tile_of_round(w, it, nwg), tm, tn);
        using u2 = __attribute__((ext_vector_type(2))) unsigned;
        using u4 = __attribute__((ext_vector_type(4))) unsigned;
        unsigned char *stg = lds + 2 * kBufBytes + wave * kStageBytes;
        const float alpha = G.alpha;
        if constexpr (EPI == 0) {
            const int rr = lane >> 4, cc = lane & 15;                          // read side: row (of four per pass) and 16-byte chunk
            unsigned short *cdst = G.C + ((long long)tm * 256 + wr * 128 + rr) * G.ldc + (long long)tn * 256 + wc * 128 + cc * 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    u2 v;
                    v[0] = pack2(acc[i][j][0] * alpha, acc[i][j][1] * alpha);
                    v[1] = pack2(acc[i][j][2] * alpha, acc[i][j][3] * alpha);
                    const int c = j * 2 + (lq >> 1);
                    *reinterpret_cast<u2 *>(stg + lm * 256 + ((c ^ lm) << 4) + (lq & 1) * 8) = v;
                    acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int ps = 0; ps < 4; ++ps) {
                    const int r = ps * 4 + rr;
                    *reinterpret_cast<u4 *>(cdst + (long long)(i * 16 + ps * 4) * G.ldc) = *reinterpret_cast<const u4 *>(stg + r * 256 + ((cc ^ r) << 4));
                }
                asm volatile("" ::: "memory");
            }
        } else if constexpr (EPI >= 4) {
            // The down projection's input gradient with the GLU backward behind it (full fine-tune): the tile is d = d(act(gate) * up), rounded to bf16 as the plain
            // kernel stores it; on the way out a lane takes gate and up of its eight columns and writes d gate = d * up * act'(gate) and d up = d * act(gate) --
            // ecgb_glu_bwd's arithmetic, the same bits as the two kernels, one write and one read of [M, glu_I] less.
            auto lo_f = [](unsigned x) { return __uint_as_float(x << 16); };
            auto hi_f = [](unsigned x) { return __uint_as_float(x & 0xFFFF0000u); };
            constexpr bool GLU = EPI < 8, GELU = GLU && (EPI & 1) != 0, LORA = EPI >= 6;      // EPI 8: the adapter's share and a plain store (a single-module site without a GLU behind it: o)
            const int rr = lane >> 4, cc = lane & 15;
            const long long col = (long long)tn * 256 + wc * 128 + cc * 8, row0 = (long long)tm * 256 + wr * 128 + rr;
            unsigned short *cdst = G.C + row0 * G.ldc + col;
            const unsigned short *gsrc = GLU ? G.GU + row0 * G.ldgu + col : nullptr;
            const long long gI = G.glu_I;
            // EPI 6 / 7 (LoRA fine-tune): the adapter's share of the input gradient, scale * mask . (dt A), joins the tile in the accumulators' own layout -- one more MFMA
            // per 16 x 16 block (operand rows = 16 rows of A^T for the block's columns and 16 rows of dt, rank 16 in k 0..15, the rest zero: lora_dx_kernel's product,
            // the same k slots), the forward's dropout mask regenerated from the element index, and bf16(bf16(product) + kept * scale) staged instead of bf16(product):
            // ecgb_lora_dx_glu's bits without the product ever being written (0.64 ms of the layer's backward at [32768, 8192]).
            bf16x8 atf[LORA ? 8 : 1], dtf[LORA ? 3 : 1];                       // (dt: a ring with gate / up below)
            const unsigned short *dtp = nullptr;
            unsigned hrow = 0, hstep = 0;
            if constexpr (LORA) {
                const bf16x8 zero = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
                const unsigned short *atp = G.LAT + ((long long)tn * 256 + wc * 128 + lm) * 64 + 8 * (lq & 1);
                dtp = G.LDT + ((long long)tm * 256 + wr * 128 + lm) * 64 + 8 * (lq & 1);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const bf16x8 a = *reinterpret_cast<const bf16x8 *>(atp + k * 16 * 64);
                    atf[k] = lq < 2 ? a : zero;
                }
                // hash32(seed, k) = k * 0x9E3779B1 + seed, then the finaliser, k = (row * inter + col) / 2: the product is formed once per lane, a block further is an add
                hrow = (((unsigned)(tm * 256 + wr * 128 + lm) * (unsigned)G.glu_I + (unsigned)(tn * 256 + wc * 128 + lq * 4)) >> 1) * 0x9E3779B1u + G.lseed;
                hstep = 8u * (unsigned)G.glu_I * 0x9E3779B1u;
            }
            // gate / up of MFMA rows i + 1 and i + 2 are in flight while row i is worked on (a ring of three register sets: one row ahead the tile's store was a chain
            // of memory latencies, 14 us per tile with a wave alone on its SIMD)
            constexpr int AHEAD = LORA ? 1 : 2;
            u4 gq[AHEAD + 1][4], uq[AHEAD + 1][4];
            auto fetch = [&](int row16, int slot) {
#pragma unroll
                for (int ps = 0; ps < 4; ++ps) {
                    if constexpr (GLU) {
                        gq[slot][ps] = *reinterpret_cast<const u4 *>(gsrc + (long long)(row16 * 16 + ps * 4) * G.ldgu);
                        uq[slot][ps] = *reinterpret_cast<const u4 *>(gsrc + (long long)(row16 * 16 + ps * 4) * G.ldgu + gI);
                    }
                }
                if constexpr (LORA) dtf[slot] = *reinterpret_cast<const bf16x8 *>(dtp + row16 * 16 * 64);
            };
#pragma unroll
            for (int k = 0; k < AHEAD; ++k) fetch(k, k);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (i + AHEAD < 8) fetch(i + AHEAD, (i + AHEAD) % (AHEAD + 1));
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    u2 v;
                    v[0] = pack2(acc[i][j][0] * alpha, acc[i][j][1] * alpha);
                    v[1] = pack2(acc[i][j][2] * alpha, acc[i][j][3] * alpha);
                    if constexpr (LORA) {
                        const bf16x8 dti = lq < 2 ? dtf[i % (AHEAD + 1)] : (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
                        // D'[col][row]: lane = row lm, regs = columns 4 lq ..  Written out (with the wait states the result needs before a VALU instruction may read it:
                        // the compiler does not look inside) because the builtin's result was allocated in the accumulation half, all of which the tile holds --
                        // 32 accumulators went to scratch and back per tile.
                        f32x4 lo;
                        asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, 0\n\ts_nop 15\n\ts_nop 3" : "=v"(lo) : "v"(atf[j]), "v"(dti));
                        float o[4] = {lo_f(v[0]), hi_f(v[0]), lo_f(v[1]), hi_f(v[1])};
                        const unsigned hb = hrow + (unsigned)i * hstep;
#pragma unroll
                        for (int e = 0; e < 4; e += 2) {                      // keep_bits<1> of lora.hip: columns 2k, 2k + 1 take the low and the high field of hash32(seed, k)
                            unsigned u = hb + (unsigned)(j * 8 + (e >> 1)) * 0x9E3779B1u;
                            u ^= u >> 15; u *= 0x85EBCA77u;
                            u ^= u >> 13; u *= 0xC2B2AE3Du;
                            u ^= u >> 16;
                            // (lora_dx_kernel adds the kept product to 0.f first: a product accumulated from +0 is never -0, the same bits without)
                            o[e] = o[e] + ((u & 0xFFFFu) >= G.lthr ? lo[e] * G.lscale : 0.f);
                            o[e + 1] = o[e + 1] + ((u >> 16) >= G.lthr ? lo[e + 1] * G.lscale : 0.f);
                        }
                        v[0] = pack2(o[0], o[1]);
                        v[1] = pack2(o[2], o[3]);
                    }
                    const int c = j * 2 + (lq >> 1);
                    *reinterpret_cast<u2 *>(stg + lm * 256 + ((c ^ lm) << 4) + (lq & 1) * 8) = v;
                    acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const int slot = i % (AHEAD + 1);
#pragma unroll
                for (int ps = 0; ps < 4; ++ps) {
                    const int r = ps * 4 + rr;
                    const u4 d = *reinterpret_cast<const u4 *>(stg + r * 256 + ((cc ^ r) << 4));
                    unsigned short *o = cdst + (long long)(i * 16 + ps * 4) * G.ldc;
                    if constexpr (!GLU) {
                        *reinterpret_cast<u4 *>(o) = d;
                    } else {
                        u4 og, ou;
#pragma unroll
                        for (int x = 0; x < 4; ++x) {
                            const float g0 = lo_f(gq[slot][ps][x]), g1 = hi_f(gq[slot][ps][x]), u0 = lo_f(uq[slot][ps][x]), u1 = hi_f(uq[slot][ps][x]), d0 = lo_f(d[x]), d1 = hi_f(d[x]);
                            og[x] = pack2(d0 * u0 * ecgb::glu_act_grad<GELU>(g0), d1 * u1 * ecgb::glu_act_grad<GELU>(g1));
                            ou[x] = pack2(d0 * ecgb::glu_act<GELU>(g0), d1 * ecgb::glu_act<GELU>(g1));
                        }
                        *reinterpret_cast<u4 *>(o) = og;
                        *reinterpret_cast<u4 *>(o + gI) = ou;
                    }
                }
                asm volatile("" ::: "memory");
            }
        } else if constexpr (EPI == 3) {
            // the q|k|v projection with RoPE's forward in the epilogue (modeling_llama.py:151-176: q_embed = q * cos + rotate_half(q) * sin, in the activation dtype).  Staged as
            // EPI 0; on the way out a lane takes its own 16-byte chunk and the chunk 32 columns away in the same head of 64, and writes
            //     bf16(x1 * c - x2 * s)  (first half of the head)   /   bf16(x2 * c + x1 * s)  (second half)
            // from the bf16-rounded projection with cos / sin rounded to bf16 -- ecgb_rope's arithmetic on the stored tensor, the same bits, without storing and re-reading it.
            auto lo_f = [](unsigned x) { return __uint_as_float(x << 16); };
            auto hi_f = [](unsigned x) { return __uint_as_float(x & 0xFFFF0000u); };
            using f4 = __attribute__((ext_vector_type(4))) float;
            const int rr = lane >> 4, cc = lane & 15;
            const long long col = (long long)tn * 256 + wc * 128 + cc * 8;
            const bool rot = col < G.rope_cols, second = (cc & 4) != 0;
            unsigned short *cdst = G.C + ((long long)tm * 256 + wr * 128 + rr) * G.ldc + col;
            const long long trow0 = (long long)tm * 256 + wr * 128 + rr;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    u2 v;
                    v[0] = pack2(acc[i][j][0] * alpha, acc[i][j][1] * alpha);
                    v[1] = pack2(acc[i][j][2] * alpha, acc[i][j][3] * alpha);
                    const int c = j * 2 + (lq >> 1);
                    *reinterpret_cast<u2 *>(stg + lm * 256 + ((c ^ lm) << 4) + (lq & 1) * 8) = v;
                    acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int ps = 0; ps < 4; ++ps) {
                    const int r = ps * 4 + rr;
                    u4 own = *reinterpret_cast<const u4 *>(stg + r * 256 + ((cc ^ r) << 4));
                    if (rot) {
                        const u4 oth = *reinterpret_cast<const u4 *>(stg + r * 256 + (((cc ^ 4) ^ r) << 4));
                        const float *pc = G.rope_cos + (trow0 + i * 16 + ps * 4) * 32 + (cc & 3) * 8, *ps_ = G.rope_sin + (trow0 + i * 16 + ps * 4) * 32 + (cc & 3) * 8;
                        const f4 c0 = *reinterpret_cast<const f4 *>(pc), c1 = *reinterpret_cast<const f4 *>(pc + 4);
                        const f4 s0 = *reinterpret_cast<const f4 *>(ps_), s1 = *reinterpret_cast<const f4 *>(ps_ + 4);
#pragma unroll
                        for (int x = 0; x < 4; ++x) {
                            const unsigned cw = pack2(x < 2 ? c0[2 * x] : c1[2 * x - 4], x < 2 ? c0[2 * x + 1] : c1[2 * x - 3]);
                            const unsigned sw = pack2(x < 2 ? s0[2 * x] : s1[2 * x - 4], x < 2 ? s0[2 * x + 1] : s1[2 * x - 3]);
                            const float pa = lo_f(oth[x]) * lo_f(sw), pb = hi_f(oth[x]) * hi_f(sw);
                            own[x] = pack2(lo_f(own[x]) * lo_f(cw) + (second ? pa : -pa), hi_f(own[x]) * hi_f(cw) + (second ? pb : -pb));
                        }
                    }
                    *reinterpret_cast<u4 *>(cdst + (long long)(i * 16 + ps * 4) * G.ldc) = own;
                }
                asm volatile("" ::: "memory");
            }
        } else {
            // three arrays of 16 rows x 64 columns (gate, up, act(gate) * up: 2 KiB each), row r at r * 128 bytes, its 16-byte chunk c at c ^ (r & 7); out in whole
            // 128-byte rows, eight rows per store instruction
            auto lo_f = [](unsigned x) { return __uint_as_float(x << 16); };
            auto hi_f = [](unsigned x) { return __uint_as_float(x & 0xFFFF0000u); };
            const int rr = lane >> 3, cc = lane & 7;
            const long long hcol = (long long)tn * 128 + wc * 64 + cc * 8;       // column of H = gate column of C
            const long long grow0 = (long long)tm * 256 + wr * 128 + rr;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    u2 gt, up, h;
#pragma unroll
                    for (int x = 0; x < 2; ++x) {
                        gt[x] = pack2(acc[i][2 * p][2 * x] * alpha, acc[i][2 * p][2 * x + 1] * alpha);
                        up[x] = pack2(acc[i][2 * p + 1][2 * x] * alpha, acc[i][2 * p + 1][2 * x + 1] * alpha);
                        const unsigned a = pack2(ecgb::glu_act<EPI == 2>(lo_f(gt[x])), ecgb::glu_act<EPI == 2>(hi_f(gt[x])));
                        h[x] = pack2(lo_f(a) * lo_f(up[x]), hi_f(a) * hi_f(up[x]));
                    }
                    acc[i][2 * p] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    acc[i][2 * p + 1] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    const int c = p * 2 + (lq >> 1);
                    unsigned char *dst = stg + lm * 128 + ((c ^ (lm & 7)) << 4) + (lq & 1) * 8;
                    if (G.C) { *reinterpret_cast<u2 *>(dst) = gt; *reinterpret_cast<u2 *>(dst + 2048) = up; }
                    *reinterpret_cast<u2 *>(dst + 4096) = h;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int ps = 0; ps < 2; ++ps) {
                    const int r = ps * 8 + rr;
                    const unsigned char *src = stg + r * 128 + ((cc ^ (r & 7)) << 4);
                    const long long grow = grow0 + i * 16 + ps * 8;
                    if (G.C) {
                        *reinterpret_cast<u4 *>(G.C + grow * G.ldc + hcol) = *reinterpret_cast<const u4 *>(src);
                        *reinterpret_cast<u4 *>(G.C + grow * G.ldc + G.glu_I + hcol) = *reinterpret_cast<const u4 *>(src + 2048);
                    }
                    *reinterpret_cast<u4 *>(G.H + grow * G.ldh + hcol) = *reinterpret_cast<const u4 *>(src + 4096);
                }
                asm volatile("" ::: "memory");
            }
        }
    }
#undef W4_MFMA
#undef W4_FENCE
}

int g_w4_group_m = 8;
int g_w4_min_ktiles = 128;      // K-tiles per workgroup from which the entry points of gemm.hip send a problem here (round 3: 256; with round 4's schedule the qkv / o shapes gain too)
int g_w4_sched = 1;              // 1: four barriers per K-tile with counted waits (round 4); 0: one rendezvous per K-tile (round 3), kept for A/B

// CUs of the current device, rounded down to a multiple of the 8 XCDs (one persistent workgroup each); looked up per device, cached per device
int w4_cus()
{
    static int cached[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!cached[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached[dev] = (n & ~7) > 0 ? (n & ~7) : 8;
    }
    return cached[dev];
}

}  // namespace

namespace ecgb {
// The DMA descriptor of an operand is set ONCE per output tile and the K loop walks the contraction with 32-bit scalar offsets (soff_at / dma_step_*), so the largest
// byte offset a piece ever uses must stay below 2^32: a row operand (NT's A and B, NN's A) reaches 63 rows (+ `extra_rows_b`: the GLU forms' up rows) plus the whole
// contraction along the row; an operand stored as contraction rows (NN's B, TN's A and B: lay 1 / 2) reaches K + 15 rows.  Past that the eight-wave kernels
// (64-bit pointers per K-tile) take the problem.
bool gemm_w4_span_ok(int lay, long long lda, long long ldb, long long K, long long extra_rows_b)
{
    const long long lim = 0xFFFFFFFFll;
    const long long a = lay == 2 ? (K + 15) * lda * 2 + 512 : 63 * lda * 2 + K * 2 + 128;
    const long long b = lay >= 1 ? (K + 15) * ldb * 2 + 512 : (63 + extra_rows_b) * ldb * 2 + K * 2 + 128;
    return lda > 0 && ldb > 0 && K > 0 && a <= lim && b <= lim;
}

// For gemm.hip's dispatch: does the four-wave kernel take this problem?  Whole 256x256 tiles, operands 16-byte aligned, per-lane DMA offsets within 32 bits, and
// at least 256 K-tiles per workgroup (one persistent workgroup per CU with a static share of the tiles): measured on the step's shapes, the kernel gains 4-6 %
// where a workgroup has 512 K-tiles and more ([32768, 8192] -> 2048, the loss head, gate|up) and loses 0-3 % at 128-192 (qkv, o: the eight-wave kernels'
// second wave per SIMD hides the tile hand-over better).
bool gemm_w4_applies(const void *a_dev, long long lda, const void *b_dev, long long ldb, const void *c_dev, long long ldc, int M, int N, int K)
{
    const int g_w4_cus = w4_cus();
    return M > 0 && N > 0 && K > 0 && M % 256 == 0 && N % 256 == 0 && K % BK == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0 &&
           (((uintptr_t)a_dev | (uintptr_t)b_dev | (uintptr_t)c_dev) & 15) == 0 && (long long)63 * lda * 2 + 128 <= 0xFFFFFFFFll && (long long)63 * ldb * 2 + 128 <= 0xFFFFFFFFll &&
           (long long)(M / 256) * (N / 256) / g_w4_cus * (K / BK) >= g_w4_min_ktiles;
}

// epi 0: C [M, N] plain.  epi 1 / 2 (SiLU / tanh-GELU): N = 2 * inter, B = [gate rows; up rows], H [M, inter] = act(gate) * up, C (gate|up) may be null.
int gemm_w4_launch(const void *a_dev, long long lda, const void *b_dev, long long ldb, void *c_dev, long long ldc, int M, int N, int K, float alpha, void *stream,
                   int epi, void *h_dev, long long ldh, const void *a2_dev, long long lda2, const void *b2_dev, long long ldb2, int K2,
                   const float *rope_cos, const float *rope_sin, int rope_cols, int lay, const void *gu_dev, long long ldgu,
                   const void *ldt_dev = nullptr, const void *lat_dev = nullptr, float lscale = 0.f, unsigned lthr = 0, unsigned lseed = 0)
{
    W4Args G;
    G.LDT = (const unsigned short *)ldt_dev; G.LAT = (const unsigned short *)lat_dev; G.lscale = lscale; G.lthr = lthr; G.lseed = lseed;
    G.GU = (const unsigned short *)gu_dev; G.ldgu = ldgu;
    G.rope_cos = rope_cos; G.rope_sin = rope_sin; G.rope_cols = rope_cols;
    G.A2 = (const unsigned short *)a2_dev; G.B2 = (const unsigned short *)b2_dev; G.lda2 = lda2; G.ldb2 = ldb2; G.K2 = K2 > 0 ? K2 : 0;
    G.A = (const unsigned short *)a_dev; G.B = (const unsigned short *)b_dev; G.C = (unsigned short *)c_dev;
    G.lda = lda; G.ldb = ldb; G.ldc = ldc; G.M = M; G.N = N; G.K = K; G.tiles_m = M / 256; G.tiles_n = N / 256; G.alpha = alpha; G.group_m = g_w4_group_m;
    G.H = (unsigned short *)h_dev; G.ldh = ldh; G.glu_I = epi >= 4 ? N : N / 2;
    const int which = epi >= 4 ? 6 + epi : lay ? 7 + lay : epi + (G.K2 ? 4 : 0);      // (10, 11: the NN product with the GLU backward behind it; 12, 13: and the LoRA adapter's share; 14: the adapter's share and a plain store)
#define W4_KERNS(S) {gemm_nt_w4_kernel<0, false, 0, S>, gemm_nt_w4_kernel<1, false, 0, S>, gemm_nt_w4_kernel<2, false, 0, S>, gemm_nt_w4_kernel<3, false, 0, S>, \
                     gemm_nt_w4_kernel<0, true, 0, S>, gemm_nt_w4_kernel<1, true, 0, S>, gemm_nt_w4_kernel<2, true, 0, S>, gemm_nt_w4_kernel<3, true, 0, S>,     \
                     gemm_nt_w4_kernel<0, false, 1, S>, gemm_nt_w4_kernel<0, false, 2, S>, gemm_nt_w4_kernel<4, false, 1, S>, gemm_nt_w4_kernel<5, false, 1, S>, \
                     gemm_nt_w4_kernel<6, false, 1, S>, gemm_nt_w4_kernel<7, false, 1, S>, gemm_nt_w4_kernel<8, false, 1, S>}
    void (*const kerns[2][15])(W4Args) = {W4_KERNS(0), W4_KERNS(1)};
#undef W4_KERNS
    void (*kern)(W4Args) = kerns[g_w4_sched ? 1 : 0][which];
#ifdef ECGB_PROFILE
    if (g_w4_sched >= 16 && which == 0) {                                  // timing-only diagnostics of the plain NT kernel
        const int d = g_w4_sched >> 4;
        kern = d == 1 ? gemm_nt_w4_kernel<0, false, 0, 17> : d == 2 ? gemm_nt_w4_kernel<0, false, 0, 33> : d == 3 ? gemm_nt_w4_kernel<0, false, 0, 49> :
               d == 4 ? gemm_nt_w4_kernel<0, false, 0, 65> : d == 7 ? gemm_nt_w4_kernel<0, false, 0, 113> : kern;
    }
#endif
    // (the attribute is per device: set on every launch, as everywhere else in the library -- a cached flag would be wrong on a second GPU)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(kern, dim3((unsigned)w4_cus()), dim3(256), kLdsBytes, (hipStream_t)stream, G);
        e = hipGetLastError();
    }
    if (e != hipSuccess) { ecgb::set_error(std::string("gemm_nt_w4_kernel: ") + hipGetErrorString(e)); return ECGB_ERR_HIP; }
    return ECGB_OK;
}
}  // namespace ecgb

// 1 if the four-wave kernel's 32-bit DMA offsets cover an operand pair of these strides over a contraction of K (lay 0 NT, 1 NN, 2 TN), 0 otherwise: pure arithmetic,
// no device needed (tests/test_host.py pins the >= 4 GiB cases the dispatch must hand to the eight-wave kernels).
extern "C" int ecgb_gemm_w4_span_ok(int lay, long long lda, long long ldb, long long K) { return ecgb::gemm_w4_span_ok(lay, lda, ldb, K, 0) ? 1 : 0; }

extern "C" int ecgb_set_gemm_w4_sched(int s)
{
    // (17 / 33 / 49 / 65 / 113: timing-only builds WITHOUT barriers / DMA waits / DMA -- wrong results by design; they exist in the dev build only, `make prof`)
#ifdef ECGB_PROFILE
    const bool ok = s >= 0 && (s <= 1 || (s & 15) == 1);
#else
    const bool ok = s == 0 || s == 1;
#endif
    if (!ok) { ecgb::set_error("ecgb_set_gemm_w4_sched: 0 (one rendezvous per K-tile) or 1 (four barriers, counted waits)"); return ECGB_ERR_INVALID; }
    g_w4_sched = s;
    return ECGB_OK;
}

extern "C" int ecgb_set_gemm_w4_min_ktiles(int n)
{
    if (n < 1) { ecgb::set_error("ecgb_set_gemm_w4_min_ktiles: n >= 1"); return ECGB_ERR_INVALID; }
    g_w4_min_ktiles = n;
    return ECGB_OK;
}

extern "C" int ecgb_set_gemm_w4_group_m(int g)
{
    if (g < 0 || g > 1024) { ecgb::set_error("ecgb_set_gemm_w4_group_m: 0..1024"); return ECGB_ERR_INVALID; }
    g_w4_group_m = g;
    return ECGB_OK;
}

// The four-wave kernel by name (tests, A/B): ECGB_ERR_UNSUPPORTED where ecgb_gemm_nt_bf16 would fall back to the 8-wave kernels.
extern "C" int ecgb_gemm_nt_w4_bf16(const void *a_dev, long long lda, const void *b_dev, long long ldb, void *c_dev, long long ldc,
                                    int M, int N, int K, float alpha, void *stream)
{
    if (!a_dev || !b_dev || !c_dev || M <= 0 || N <= 0 || K <= 0) { ecgb::set_error("ecgb_gemm_nt_w4_bf16: bad argument"); return ECGB_ERR_INVALID; }
    (void)ecgb::gemm_w4_applies(a_dev, lda, b_dev, ldb, c_dev, ldc, M, N, K);          // (device properties)
    if (M % 256 || N % 256 || K % BK || lda % 8 || ldb % 8 || ldc % 8 || (((uintptr_t)a_dev | (uintptr_t)b_dev | (uintptr_t)c_dev) & 15) ||
        !ecgb::gemm_w4_span_ok(0, lda, ldb, K, 0)) {
        ecgb::set_error("ecgb_gemm_nt_w4_bf16: M, N multiples of 256, K of 64, operands 16-byte aligned with strides % 8 == 0");
        return ECGB_ERR_UNSUPPORTED;
    }
    return ecgb::gemm_w4_launch(a_dev, lda, b_dev, ldb, c_dev, ldc, M, N, K, alpha, stream, 0, nullptr, 0, nullptr, 0, nullptr, 0, 0, nullptr, nullptr, 0, 0, nullptr, 0, nullptr, nullptr, 0.f, 0, 0);
}

// The q|k|v projection with RoPE's forward in the epilogue: C = alpha * (A B^T [+ A2 B2^T]), then every head of 64 columns below rope_cols rotated with row t of the
// [M, 32] fp32 tables (ecgb_rope's arithmetic on the bf16-rounded projection: the same bits as the two calls).  Whole 256x256 tiles and one tile per CU at least;
// ECGB_ERR_UNSUPPORTED otherwise (the caller runs ecgb_gemm_nt_bf16[_cat] and ecgb_rope).
extern "C" int ecgb_gemm_nt_bf16_rope(const void *a_dev, long long lda, const void *b_dev, long long ldb, const void *a2_dev, long long lda2,
                                      const void *b2_dev, long long ldb2, int K2, void *c_dev, long long ldc, int M, int N, int K, float alpha,
                                      const float *rope_cos_dev, const float *rope_sin_dev, int rope_cols, void *stream)
{
    if (!a_dev || !b_dev || !c_dev || !rope_cos_dev || !rope_sin_dev || M <= 0 || N <= 0 || K <= 0 || (K2 > 0 && (!a2_dev || !b2_dev))) {
        ecgb::set_error("ecgb_gemm_nt_bf16_rope: bad argument");
        return ECGB_ERR_INVALID;
    }
    (void)ecgb::gemm_w4_applies(a_dev, lda, b_dev, ldb, c_dev, ldc, M, N, K);          // (device properties)
    if (M % 256 || N % 256 || K % BK || K2 % BK || rope_cols % 64 || rope_cols < 0 || rope_cols > N || lda % 8 || ldb % 8 || ldc % 8 || (K2 > 0 && (lda2 % 8 || ldb2 % 8)) ||
        (((uintptr_t)a_dev | (uintptr_t)b_dev | (uintptr_t)c_dev | (uintptr_t)a2_dev | (uintptr_t)b2_dev | (uintptr_t)rope_cos_dev | (uintptr_t)rope_sin_dev) & 15) ||
        !ecgb::gemm_w4_span_ok(0, lda, ldb, K, 0) || (K2 > 0 && !ecgb::gemm_w4_span_ok(0, lda2, ldb2, K2, 0)) || (long long)(M / 256) * (N / 256) < w4_cus()) {
        ecgb::set_error("ecgb_gemm_nt_bf16_rope: whole 256x256 tiles (one per CU at least), K % 64, 16-byte aligned operands required");
        return ECGB_ERR_UNSUPPORTED;
    }
    return ecgb::gemm_w4_launch(a_dev, lda, b_dev, ldb, c_dev, ldc, M, N, K, alpha, stream, 3, nullptr, 0, a2_dev, lda2, b2_dev, ldb2, K2 > 0 ? K2 : 0, rope_cos_dev, rope_sin_dev, rope_cols, 0, nullptr, 0);
}

// C = alpha * A . B with B [K, N] row-major on the four-wave kernel (tests, A/B; ecgb_gemm_nn_bf16 dispatches here for long contractions): whole tiles only.
extern "C" int ecgb_gemm_nn_w4_bf16(const void *a_dev, long long lda, const void *b_dev, long long ldb, void *c_dev, long long ldc,
                                    int M, int N, int K, float alpha, void *stream)
{
    if (!a_dev || !b_dev || !c_dev || M <= 0 || N <= 0 || K <= 0) { ecgb::set_error("ecgb_gemm_nn_w4_bf16: bad argument"); return ECGB_ERR_INVALID; }
    (void)ecgb::gemm_w4_applies(a_dev, lda, a_dev, lda, c_dev, ldc, M, N, K);          // (device properties)
    if (M % 256 || N % 256 || K % BK || lda % 8 || ldb % 8 || ldc % 8 || (((uintptr_t)a_dev | (uintptr_t)b_dev | (uintptr_t)c_dev) & 15) ||
        !ecgb::gemm_w4_span_ok(1, lda, ldb, K, 0)) {
        ecgb::set_error("ecgb_gemm_nn_w4_bf16: M, N multiples of 256, K of 64, operands 16-byte aligned with strides % 8 == 0");
        return ECGB_ERR_UNSUPPORTED;
    }
    return ecgb::gemm_w4_launch(a_dev, lda, b_dev, ldb, c_dev, ldc, M, N, K, alpha, stream, 0, nullptr, 0, nullptr, 0, nullptr, 0, 0, nullptr, nullptr, 0, 1, nullptr, 0);
}

// C[M, N] = alpha * A^T . B with A [K, M] and B [K, N] row-major (ecgb_gemm_tn_bf16's product: C = its [N, K] output, K here = its M) on the four-wave kernel, by name:
// whole 256x256 tiles, plain bf16 store.
extern "C" int ecgb_gemm_tn_w4_bf16(const void *a_dev, long long lda, const void *b_dev, long long ldb, void *c_dev, long long ldc,
                                    int M, int N, int K, float alpha, void *stream)
{
    if (!a_dev || !b_dev || !c_dev || M <= 0 || N <= 0 || K <= 0) { ecgb::set_error("ecgb_gemm_tn_w4_bf16: bad argument"); return ECGB_ERR_INVALID; }
    (void)ecgb::gemm_w4_applies(a_dev, 8, a_dev, 8, c_dev, ldc, M, N, K);              // (device properties)
    if (M % 256 || N % 256 || K % BK || lda % 8 || ldb % 8 || ldc % 8 || (((uintptr_t)a_dev | (uintptr_t)b_dev | (uintptr_t)c_dev) & 15) ||
        !ecgb::gemm_w4_span_ok(2, lda, ldb, K, 0)) {
        ecgb::set_error("ecgb_gemm_tn_w4_bf16: M, N multiples of 256, K of 64, operands 16-byte aligned with strides % 8 == 0");
        return ECGB_ERR_UNSUPPORTED;
    }
    return ecgb::gemm_w4_launch(a_dev, lda, b_dev, ldb, c_dev, ldc, M, N, K, alpha, stream, 0, nullptr, 0, nullptr, 0, nullptr, 0, 0, nullptr, nullptr, 0, 2, nullptr, 0);
}
