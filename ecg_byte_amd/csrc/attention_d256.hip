// attention_d256.hip -- head_dim 256 (Gemma) attention backward kernels, round 6: LDS reads the compiler can see, issued under the products of the group before.
// (The forward kernel of this head_dim is round 4's, in attention.hip: the same treatment did not pay there -- scripts/experiments/r06_attn_fwd_d256_v2.hip.txt.)
// Reference: see attention.hip (modeling_gemma.py:201-300 for this head layout).
#include "attention_common.inc"

namespace {

// ---- round 6: the head_dim-256 kernels with LDS reads the compiler can see -------------------------------------------------------------------------------------
// Round 4's kernels read LDS through asm statements that end in their own lgkmcnt(0) (a ds_read hipcc can see next to LDS-DMA gets a vmcnt(0) in front of it -- the DMA may
// write what the read reads, as far as it knows): every group of four fragments was an exposed LDS round trip in front of its four MFMAs on a SIMD that holds ONE wave.
// Here the LDS-DMA is the asm statement (lds_dma16): hipcc then knows of no LDS writer, the reads are ordinary loads that it counts (lgkmcnt(3) in front of a product: the
// three younger reads stay in flight), and each group's reads are issued BETWEEN the products of the group before (sched_group_barrier: a product, then the reads that fit under
// it -- a wave issues in order and is alone on its SIMD, reads in front of the products are issued while the matrix pipe has nothing left to do).  What the DMA writes is read
// only behind the kernels' own vmcnt(0) + barrier.  The arithmetic and its order are unchanged: the same bits as the round-4 kernels and the register-staged ones.
#define SG_MFMA(n) __builtin_amdgcn_sched_group_barrier(0x008, n, 0)
#define SG_DSR(n) __builtin_amdgcn_sched_group_barrier(0x100, n, 0)
#define SG_VMEM(n) __builtin_amdgcn_sched_group_barrier(0x020, n, 0)
// 1 KiB global -> LDS (16 bytes a lane from base + off, to lds + 16 lane), as an asm statement: with an LDS-DMA it can see pending, hipcc waits for lgkmcnt(0) in front of
// every use of an LDS read (the reads issued under the last product included); what the DMA writes is read only behind the kernel's own vmcnt(0) + barrier
__device__ __forceinline__ void lds_dma16(unsigned char *lds, const unsigned char *base, unsigned off)
{
    const unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(la), "v"(off), "s"(base) : "memory");   // (m0: nothing else of these kernels uses it)
}
// LDS reads hipcc can see, by LDS address: a row fragment (ds_read_b128), a transposed one (two ds_read_b64_tr_b16)
__device__ __forceinline__ bf16x8 lds_frag_a(unsigned a) { return *reinterpret_cast<const __attribute__((address_space(3))) bf16x8 *>((size_t)a); }
__device__ __forceinline__ bf16x8 lds_tr_frag_a(unsigned a, unsigned b)
{
    using s4 = __attribute__((ext_vector_type(4))) short;
    const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4 *)(size_t)a);
    const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4 *)(size_t)b);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ bf16x8 lds_frag_c(const unsigned char *p) { return *reinterpret_cast<const bf16x8 *>(p); }
__device__ __forceinline__ bf16x8 lds_tr_frag_c(const unsigned char *a, const unsigned char *b)
{
    using s4 = __attribute__((ext_vector_type(4))) short;
    const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4 *)const_cast<unsigned char *>(a));
    const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4 *)const_cast<unsigned char *>(b));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
__global__ __launch_bounds__(256) void attn_bwd_dq_d256_kernel(AttnArgs A)
{
    constexpr int D = 256, kRow = D * 2, kTile = 64 * kRow, PPW = 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];        // 2 x (K image, V image), the row's key mask, 64 tile flags
    float *lds_maskrow = reinterpret_cast<float *>(smem + 4 * kTile);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, lr = lane & 31, h = lane >> 5;
    int qblk, head_in, group;
    map_block((int)blockIdx.x, (A.S + 127) / 128, A.Hq / A.Hkv, A.B * A.Hkv, true, qblk, head_in, group);
    const int b = group / A.Hkv, g = group % A.Hkv, hq = g * (A.Hq / A.Hkv) + head_in;
    const int q0 = qblk * 128, qw0 = q0 + wave * 32;
    const int qi = qw0 + lr;
    const bool qvalid = qi < A.S;
    const long long rowbase = (long long)b * A.S;
    const unsigned short *Q = A.q + (long long)hq * D, *K = A.k + (long long)g * D, *V = A.v + (long long)g * D;
    const int k_end = min(A.S, q0 + 128);
    const int wave_qmax = qw0 + 31;
    const int last_tile = (k_end - 1) / 64;
    const int tail_rows = A.S - last_tile * 64;
    // piece i of this wave: tile rows r0 + 2 i (r0 = 16 wave + the lane's half), the lane's 16-byte slot lane & 31 of the LDS row takes global chunk slot ^ swizzle(row)
    const int r0 = wave * (2 * PPW) + (lane >> 5);
    // swz_f256(r0 + 2 i) = 4 (lane >> 5) + 8 (i & 1) + (i >> 1): the pieces' chunks differ by a constant XOR
    const unsigned chunk0 = (unsigned)((((lane & 31) ^ ((lane >> 5) << 2)) * 8) * 2);
    const unsigned ldk2 = (unsigned)(A.ldk * 2), ldv2 = (unsigned)(A.ldv * 2);
    const unsigned char *kb_next = reinterpret_cast<const unsigned char *>(K + rowbase * A.ldk), *vb_next = reinterpret_cast<const unsigned char *>(V + rowbase * A.ldv);
    const long long stepK = 128ll * A.ldk, stepV = 128ll * A.ldv;
    int t_next = 0;
    unsigned slot_next = 0;
    // The pieces' row offsets from the tile's first row, kept (the chunk is a constant XOR): a piece between two groups of MFMAs is then one XOR and the load (made where it is
    // issued each cost four vector instructions, and the groups are bound by what the wave can issue beside an MFMA: 5.7 vector instructions a product, EXPERIMENTS.md R6).
    // A tile that ends past the sequence re-reads its last row (masked as keys >= S): its offsets are made when it comes up, once a workgroup at most.
    unsigned offK[PPW], offV[PPW];
    auto piece_offsets = [&](int rmax) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const unsigned row = (unsigned)min(r0 + 2 * i, rmax), ch = chunk0 ^ (unsigned)((((i & 1) << 3) | (i >> 1)) << 4);
            offK[i] = row * ldk2 + ch; offV[i] = row * ldv2 + ch;
        }
    };
    piece_offsets(last_tile == 0 ? min(63, tail_rows - 1) : 63);
    // piece i of the NEXT tile's K (which = 0) / V (which = 1) image into the other stage
    auto issue_piece = [&](int which, int i) __attribute__((always_inline)) {
        lds_dma16(smem + slot_next + which * kTile + (wave * PPW + i) * 1024, which ? vb_next : kb_next, which ? offV[i] : offK[i]);
    };
    auto advance_next = [&]() {
        if (t_next < last_tile) {
            ++t_next; kb_next += stepK; vb_next += stepV;
            if (t_next == last_tile && tail_rows < 64) piece_offsets(tail_rows - 1);      // (uniform)
        }
        slot_next ^= 2 * kTile;
    };
#pragma unroll
    for (int i = 0; i < PPW; ++i) { issue_piece(0, i); issue_piece(1, i); }
    advance_next();
    bf16x8 qf[D / 16], dof[D / 16];
    float delta = 0.f;
    {
        bf16x8 of[D / 16];
        load_row_frags<D>(qf, Q, A.ldq, rowbase + qi, qvalid, h);
        load_row_frags<D>(dof, A.d_o + (long long)hq * D, A.ldo, rowbase + qi, qvalid, h);
        load_row_frags<D>(of, A.o + (long long)hq * D, A.ldo, rowbase + qi, qvalid, h);
#pragma unroll
        for (int ks = 0; ks < D / 16; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) delta += bf2f((unsigned short)dof[ks][j]) * bf2f((unsigned short)of[ks][j]);
    }
    delta += __shfl_xor(delta, 32, 64);
    const long long stat = ((long long)b * A.Hq + hq) * A.S + qi;
    if (qvalid && h == 0) A.delta[stat] = delta;
    const float lse = qvalid ? A.lse[stat] : INFINITY;
    f32x16 accQ[D / 32];
#pragma unroll
    for (int db = 0; db < D / 32; ++db) accQ[db] = splat16(0.f);
    const float sc = A.scale * kLog2e;
    lean_fill_mask<4>(lds_maskrow, A.mask + rowbase, A.S, (k_end + 63) & ~63);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // tile 0 and the row operands
    __syncthreads();
    const unsigned long long padbits = lean_pad_bits(lds_maskrow, (k_end + 63) & ~63);
    // this lane's row fragment (row lr, k-step 0) inside an image, as an offset: k-step ks is the offset ^ (ks << 5); row 32 + lr 16 KiB further; the V image kTile further
    const unsigned rbase = lr * kRow + ((h ^ swz_f256(lr)) << 4);
    // this lane's transposing reads inside a K image for d block 0, first (key 4 h + q) and second (8 keys further: f moves with bit 3 of the row): block db is the offset ^ (db << 6)
    unsigned tbaseA, tbaseB;
    {
        const int a = lr >> 4, q = (lr & 15) >> 2, p = lr & 3;
        const int cg = 2 * a + (p >> 1);                                 // 16-byte chunk inside the 64-byte group
        const int k1 = 4 * h + q, k2 = k1 + 8;
        tbaseA = k1 * kRow + ((((k1 & 3) << 2) | (cg ^ ((k1 >> 2) & 3))) << 4) + (p & 1) * 8;
        tbaseB = k2 * kRow + ((((k2 & 3) << 2) | (cg ^ ((k2 >> 2) & 3))) << 4) + (p & 1) * 8;      // = tbaseA ^ (4096 | 32): k1 < 8, and (k >> 2) & 3 moves by 2
        (void)tbaseB;
    }
    // The lane's read addresses are LDS ADDRESSES (the ring is the kernel's only LDS object and starts at address 0 -- checked below): the current stage is bit 16 of them,
    // a k-step / d block an XOR of bits 5 - 8, so that one register a kind, flipped once a trip, and one XOR a fragment address is all the arithmetic the reads need (with
    // the stage added at the reads: five vector instructions a group of four products, and the groups are bound by what a wave can issue beside an MFMA).  The second
    // transposing read (8 keys further) is the first ^ (4096 | 32): the row and bit 1 of the swizzle's chunk.
    if ((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem != 0u) __builtin_trap();
    unsigned sbase = rbase, tbase = tbaseA;
    // the four row fragments of group gp (k-steps 2 gp, 2 gp + 1) of key half kb: K, V, K, V
    auto load_sdp = [&](bf16x8 (&f)[4], int kb, int gp) __attribute__((always_inline)) {
        const unsigned a0 = (sbase ^ (unsigned)((2 * gp) << 5)) + kb * 32 * kRow, a1 = (sbase ^ (unsigned)((2 * gp + 1) << 5)) + kb * 32 * kRow;
        f[0] = lds_frag_a(a0); f[1] = lds_frag_a(a0 + kTile);
        f[2] = lds_frag_a(a1); f[3] = lds_frag_a(a1 + kTile);
    };
    // the four transposed K fragments [k-step s2][d block 2 dpair + j] of key half kb
    auto load_tr = [&](bf16x8 (&f)[2][2], int kb, int dpair) __attribute__((always_inline)) {
        const unsigned x0 = (unsigned)((2 * dpair) << 6), x1 = (unsigned)((2 * dpair + 1) << 6);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const unsigned off = (kb * 32 + s2 * 16) * kRow;
            f[s2][0] = lds_tr_frag_a((tbase ^ x0) + off, (tbase ^ x0 ^ 4128u) + off);
            f[s2][1] = lds_tr_frag_a((tbase ^ x1) + off, (tbase ^ x1 ^ 4128u) + off);
        }
    };
#ifdef ECGB_PROFILE
    unsigned long long prof_acc[7] = {};
    long long t_prof = clock64();
#endif
    for (int k0 = 0, it = 0; k0 < k_end; k0 += 64, ++it) {
        APROF(5);
        // Said once a trip: the accumulators live in the vector half (see the dK / dV kernel below)
#pragma unroll
        for (int db = 0; db < D / 32; ++db) asm volatile("" : "+v"(accQ[db]));
        const float *lds_mask = lds_maskrow + k0;
        if (k0 <= wave_qmax) {
            const bool need_mask = (k0 + 63 > qw0) || lean_tile_padded(padbits, it);
            // the tile's 64 keys' mask words as bits in scalar registers (a lane a key, one ballot), for the half-wave's keys 4 h + ...: a shift by 4 for the upper half
            const unsigned long long keybits = need_mask ? __ballot(lds_mask[lane] != 0.f) : ~0ull;
            bf16x8 cf[4];
            load_sdp(cf, 0, 0);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                f32x16 s = splat16(0.f), dp = splat16(0.f);
                bf16x8 ktf[2][2];
#pragma unroll
                for (int gp = 0; gp < D / 32; ++gp) {                     // a group AHEAD: the next group's fragments (or the first transposed ones) are asked for, then this group's products issue
                    bf16x8 nf[4];
                    __builtin_amdgcn_sched_barrier(0);
                    if (gp + 1 < D / 32) load_sdp(nf, kb, gp + 1);
                    else load_tr(ktf, kb, 0);
                    issue_piece(kb, gp);                                  // the next tile's K pieces under key half 0's products, its V pieces under key half 1's
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cf[0], qf[2 * gp], s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cf[1], dof[2 * gp], dp, 0, 0, 0);
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cf[2], qf[2 * gp + 1], s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cf[3], dof[2 * gp + 1], dp, 0, 0, 0);
                    // the order of issue: a product, then the reads that fit under it (a wave issues in order and is alone on its SIMD: reads in FRONT of the products are
                    // issued while the matrix pipe has nothing left to do)
                    if (gp + 1 < D / 32) {
                        SG_MFMA(1); SG_DSR(1); SG_MFMA(1); SG_DSR(1); SG_MFMA(1); SG_DSR(1); SG_VMEM(1); SG_MFMA(1); SG_DSR(1);
                    } else {
                        SG_MFMA(1); SG_DSR(2); SG_MFMA(1); SG_DSR(2); SG_MFMA(1); SG_DSR(2); SG_VMEM(1); SG_MFMA(1); SG_DSR(2);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (gp + 1 < D / 32) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) cf[j] = nf[j];
                    }
                }
                APROF(0);
                float ds[16];
                if (need_mask) {
                    const unsigned kbits = (unsigned)(keybits >> (kb * 32)) >> (4 * h);      // this half-wave's keys 4 h + ... of the key half
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int kl = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        const bool vis = (k0 + kl <= qi) & ((kbits & (1u << ((r & 3) + 8 * (r >> 2)))) != 0u);
                        const float pr = vis ? fast_exp2(__builtin_fmaf(s[r], sc, -lse)) : 0.f;
                        ds[r] = pr * (dp[r] - delta) * A.scale;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) ds[r] = fast_exp2(__builtin_fmaf(s[r], sc, -lse)) * (dp[r] - delta) * A.scale;
                }
                const bf16x8 dsf0 = frag_from_acc(&ds[0]), dsf1 = frag_from_acc(&ds[8]);
                APROF(1);
#pragma unroll
                for (int dpair = 0; dpair < D / 64; ++dpair) {
                    bf16x8 ntf[2][2];
                    __builtin_amdgcn_sched_barrier(0);
                    if (dpair + 1 < D / 64) load_tr(ntf, kb, dpair + 1);
                    else if (kb == 0) load_sdp(cf, 1, 0);            // (key half 1 of the same tile; the next tile's first group waits for the barrier)
                    accQ[2 * dpair] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf[0][0], dsf0, accQ[2 * dpair], 0, 0, 0);
                    accQ[2 * dpair + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf[0][1], dsf0, accQ[2 * dpair + 1], 0, 0, 0);
                    accQ[2 * dpair] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf[1][0], dsf1, accQ[2 * dpair], 0, 0, 0);
                    accQ[2 * dpair + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf[1][1], dsf1, accQ[2 * dpair + 1], 0, 0, 0);
                    if (dpair + 1 < D / 64) { SG_MFMA(1); SG_DSR(2); SG_MFMA(1); SG_DSR(2); SG_MFMA(1); SG_DSR(2); SG_MFMA(1); SG_DSR(2); }
                    else if (kb == 0) { SG_MFMA(1); SG_DSR(1); SG_MFMA(1); SG_DSR(1); SG_MFMA(1); SG_DSR(1); SG_MFMA(1); SG_DSR(1); }
                    __builtin_amdgcn_sched_barrier(0);
                    if (dpair + 1 < D / 64) {
#pragma unroll
                        for (int s2 = 0; s2 < 2; ++s2) { ktf[s2][0] = ntf[s2][0]; ktf[s2][1] = ntf[s2][1]; }
                    }
                }
                APROF(2);
            }
#ifdef ECGB_PROFILE
            prof_acc[6] += 1;
#endif
        } else {
#pragma unroll
            for (int i = 0; i < PPW; ++i) { issue_piece(0, i); issue_piece(1, i); }
        }
        advance_next();
        APROF(3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // this wave's pieces of tile it + 1 have landed
        APROF(4);
        __builtin_amdgcn_s_barrier();                                     // ... and everybody else's; all reads of tile it are done
        sbase ^= 2 * kTile; tbase ^= 2 * kTile;
    }
#ifdef ECGB_PROFILE
    if ((threadIdx.x & 63) == 0)
        for (int kk = 0; kk < 7; ++kk) atomicAdd(&g_attn_prof[(threadIdx.x >> 6) * 8 + kk], prof_acc[kk]);
#endif
    store_accT<D / 32>(accQ, A.dq + (long long)hq * D, A.lddq, rowbase + qi, qvalid, h, 1.f);
}

// 256 B global -> LDS (4 bytes a lane), see lds_dma16
__device__ __forceinline__ void lds_dma4(unsigned char *lds, const unsigned char *base, unsigned off)
{
    const unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" :: "s"(la), "v"(off), "s"(base) : "memory");
}

// head_dim 256 dK / dV (round 6): round 4's pair kernel (attention.hip's history: two workgroups of one launch, the dV pass -- S, dO^T . P -- and the dK pass -- S, dP, Q^T . dS --
// of a block of 128 keys, a wave 32 keys, looping over (query head of the group, 64-query tile) steps whose Q and dO tiles and row statistics land by LDS-DMA in a two-stage
// ring) with its LDS reads as ordinary loads issued a group of four products ahead (see the dQ kernel above).  The DMA being asm, hipcc knows of no LDS writer and the ring
// can stay one dynamic array with a rolled step loop.  The arithmetic of the register-staged
// body in its order: the same bits.
template <int WHICH>
__device__ __forceinline__ void attn_bwd_dkv_d256_body(const AttnArgs &A, const int kblk, const int b, const int g, const int hsplit)
{
    constexpr bool DO_V = WHICH != 2, DO_K = WHICH != 1;
    constexpr int D = 256, kRow = D * 2, kTile = 64 * kRow, PPW = 8, kStage = 2 * kTile + 512;   // Q image, dO image, lse and delta rows
    constexpr int NB = D / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];                         // 2 x kStage
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, lr = lane & 31, h = lane >> 5;
    const int G = A.Hq / A.Hkv;
    const int kk0 = kblk * 128;
    const int ki = kk0 + wave * 32 + lr;
    const bool kvalid = ki < A.S;
    const long long rowbase = (long long)b * A.S;
    const int wave_kmin = kk0 + wave * 32;
    const int t_begin = (kk0 / 64) * 64;                   // first query tile that can see this key block
    const int tiles_per_head = (A.S - t_begin + 63) / 64;
    const int heads_here = G / (A.head_splits > 1 ? A.head_splits : 1), head_lo = hsplit * heads_here;   // this workgroup's query heads of the group
    const int n_steps = heads_here * tiles_per_head;
    const int tail_rows = A.S - (t_begin + (tiles_per_head - 1) * 64);
    // piece i of this wave: tile rows r0 + 2 i, the lane's 16-byte slot lane & 31 of the LDS row takes global chunk slot ^ swz_f256(row) (see the dQ kernel)
    const int r0 = wave * (2 * PPW) + (lane >> 5);
    const unsigned chunk0 = (unsigned)((((lane & 31) ^ ((lane >> 5) << 2)) * 8) * 2);
    const unsigned ldq2 = (unsigned)(A.ldq * 2), ldo2 = (unsigned)(A.ldo * 2);
    // the step being fetched: its head and tile, the last row of the tile that exists (rows past the last query re-read it: switched off in the visibility test)
    int hq_next = 0, ti_next = 0, left_next = n_steps - 1;
    unsigned slot_next = 0;
    const unsigned char *q_next, *o_next, *stat_next;
    int rmax_next;
    auto point_next = [&]() {
        const int hq = g * G + head_lo + hq_next;
        const int t0n = t_begin + ti_next * 64;
        q_next = reinterpret_cast<const unsigned char *>(A.q + (long long)hq * D + (rowbase + t0n) * A.ldq);
        o_next = reinterpret_cast<const unsigned char *>(A.d_o + (long long)hq * D + (rowbase + t0n) * A.ldo);
        stat_next = reinterpret_cast<const unsigned char *>((wave ? A.delta : A.lse) + ((long long)b * A.Hq + hq) * A.S + t0n);   // wave 0 brings the lse row, wave 1 the delta row
        rmax_next = (ti_next == tiles_per_head - 1 && tail_rows < 64) ? tail_rows - 1 : 63;
    };
    point_next();
    auto issue_piece = [&](int which, int i) __attribute__((always_inline)) {      // which: 0 the Q image, 1 the dO image
        const unsigned off = (unsigned)min(r0 + 2 * i, rmax_next) * (which ? ldo2 : ldq2) + (chunk0 ^ (unsigned)((((i & 1) << 3) | (i >> 1)) << 4));
        lds_dma16(smem + slot_next + which * kTile + (wave * PPW + i) * 1024, which ? o_next : q_next, off);
    };
    auto issue_stats = [&]() __attribute__((always_inline)) {
        if (wave < 2) lds_dma4(smem + slot_next + 2 * kTile + wave * 256, stat_next, (unsigned)min(lane, rmax_next) * 4u);
    };
    auto advance_next = [&]() {                              // (past the last step the last one is fetched again)
        if (left_next > 0) {
            --left_next;
            if (++ti_next == tiles_per_head) { ti_next = 0; ++hq_next; }
            point_next();
        }
        slot_next = slot_next ? 0u : (unsigned)kStage;
    };
#pragma unroll
    for (int i = 0; i < PPW; ++i) { issue_piece(0, i); issue_piece(1, i); }
    issue_stats();
    advance_next();
    bf16x8 kf[D / 16], vf[DO_K ? D / 16 : 1];
    load_row_frags<D>(kf, A.k + (long long)g * D, A.ldk, rowbase + ki, kvalid, h);
    if constexpr (DO_K) load_row_frags<D>(vf, A.v + (long long)g * D, A.ldv, rowbase + ki, kvalid, h);
    const bool kvis = kvalid && A.mask[rowbase + (kvalid ? ki : 0)] != 0.f;
    f32x16 acc[NB];                                          // dV (WHICH 1) or dK (WHICH 2)
#pragma unroll
    for (int db = 0; db < NB; ++db) acc[db] = splat16(0.f);
    const float sc = A.scale * kLog2e;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // step 0 and the row operands
    __syncthreads();
    const unsigned rbase = lr * kRow + ((h ^ swz_f256(lr)) << 4);         // (offsets: see the dQ kernel)
    unsigned tbaseA, tbaseB;
    {
        const int a = lr >> 4, q = (lr & 15) >> 2, p = lr & 3;
        const int cg = 2 * a + (p >> 1);
        const int k1 = 4 * h + q, k2 = k1 + 8;
        tbaseA = k1 * kRow + ((((k1 & 3) << 2) | (cg ^ ((k1 >> 2) & 3))) << 4) + (p & 1) * 8;
        tbaseB = k2 * kRow + ((((k2 & 3) << 2) | (cg ^ ((k2 >> 2) & 3))) << 4) + (p & 1) * 8;
    }
    // group gp of query half qb: the dK pass takes k-steps 2 gp, 2 gp + 1 of the Q and the dO image (Q, dO, Q, dO), the dV pass k-steps 4 gp .. 4 gp + 3 of the Q image:
    // four fragments, four products either way
    auto load_grp = [&](bf16x8 (&f)[4], const unsigned char *cur, int qb, int gp) __attribute__((always_inline)) {
        unsigned rb = rbase;
        asm volatile("" : "+v"(rb));                          // (made here, one XOR a fragment pair: hipcc otherwise keeps every address in a register of its own)
        if constexpr (DO_K) {
            const unsigned a0 = (rb ^ (unsigned)((2 * gp) << 5)) + qb * 32 * kRow, a1 = (rb ^ (unsigned)((2 * gp + 1) << 5)) + qb * 32 * kRow;
            f[0] = lds_frag_c(cur + a0); f[1] = lds_frag_c(cur + a0 + kTile);
            f[2] = lds_frag_c(cur + a1); f[3] = lds_frag_c(cur + a1 + kTile);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) f[j] = lds_frag_c(cur + (rb ^ (unsigned)((4 * gp + j) << 5)) + qb * 32 * kRow);
        }
    };
    constexpr int NG = DO_K ? D / 32 : D / 64;               // groups of a query half's score products
    // the four transposed fragments [k-step s2][d block 2 dpair + j] of query half qb: from the dO image (dV: dO^T . P) or the Q image (dK: Q^T . dS)
    auto load_tr = [&](bf16x8 (&f)[2][2], const unsigned char *cur, int qb, int dpair) __attribute__((always_inline)) {
        const unsigned x0 = (unsigned)((2 * dpair) << 6), x1 = (unsigned)((2 * dpair + 1) << 6);
        unsigned ta = tbaseA, tb = tbaseB;
        asm volatile("" : "+v"(ta), "+v"(tb));
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const unsigned off = (qb * 32 + s2 * 16) * kRow + (DO_V ? kTile : 0);
            f[s2][0] = lds_tr_frag_c(cur + (ta ^ x0) + off, cur + (tb ^ x0) + off);
            f[s2][1] = lds_tr_frag_c(cur + (ta ^ x1) + off, cur + (tb ^ x1) + off);
        }
    };
    unsigned img = 0;
    int t0 = t_begin, hq_cur = 0;                            // the step at hand: its tile's first query, its head (of this workgroup's)
#ifdef ECGB_PROFILE
    unsigned long long prof_acc[7] = {};
    long long t_prof = clock64();
#endif
    for (int step = 0; step < n_steps; ++step) {
        const unsigned char *cur = smem + img;
        APROF(5);
        issue_stats();
        // Said once a step: the accumulators live in the vector half (this file is compiled with -amdgpu-mfma-vgpr-form: the products write them there, the softmax reads the
        // scores without copies, the wave's 128 fixed operand registers are read from the accumulation half directly).  Left alone hipcc carries some accumulators round the
        // loop in the other half and copies 32 registers in and out around every group of products, or spills.
#pragma unroll
        for (int db = 0; db < NB; ++db) asm volatile("" : "+v"(acc[db]));
        if (t0 + 63 >= wave_kmin) {                          // else: every query of the tile precedes every key of this wave
            const bool tail_tile = t0 + 64 > A.S;            // (uniform) the tile holds queries past the sequence
            const float *lds_lse = reinterpret_cast<const float *>(cur + 2 * kTile), *lds_delta = lds_lse + 64;
            bf16x8 cf[4];
            load_grp(cf, cur, 0, 0);
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                f32x16 s = splat16(0.f), dp = splat16(0.f);
                bf16x8 tf[2][2];
                f4v lse4[4], dl4[DO_K ? 4 : 1];
#pragma unroll
                for (int gp = 0; gp < NG; ++gp) {            // a group AHEAD: the next group's fragments (or the first transposed ones and the statistics) are asked for under this group's products
                    bf16x8 nf[4];
                    __builtin_amdgcn_sched_barrier(0);
                    if (gp + 1 < NG) load_grp(nf, cur, qb, gp + 1);
                    else load_tr(tf, cur, qb, 0);
                    if (gp == NG - 2) {                      // the lane's 16 queries are 4 runs of 4: 16-byte reads of their statistics, two groups ahead of the softmax
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {
                            lse4[g4] = *reinterpret_cast<const f4v *>(&lds_lse[qb * 32 + 8 * g4 + 4 * h]);
                            if constexpr (DO_K) dl4[g4] = *reinterpret_cast<const f4v *>(&lds_delta[qb * 32 + 8 * g4 + 4 * h]);
                        }
                    }
                    if constexpr (DO_K) {
                        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cf[0], kf[2 * gp], s, 0, 0, 0);
                        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cf[1], vf[2 * gp], dp, 0, 0, 0);
                        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cf[2], kf[2 * gp + 1], s, 0, 0, 0);
                        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cf[3], vf[2 * gp + 1], dp, 0, 0, 0);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cf[j], kf[4 * gp + j], s, 0, 0, 0);
                    }
                    if (gp == NG - 2) {
                        if (DO_K) { SG_MFMA(1); SG_DSR(3); SG_MFMA(1); SG_DSR(3); SG_MFMA(1); SG_DSR(3); SG_MFMA(1); SG_DSR(3); }
                        else { SG_MFMA(1); SG_DSR(2); SG_MFMA(1); SG_DSR(2); SG_MFMA(1); SG_DSR(2); SG_MFMA(1); SG_DSR(2); }
                    } else if (gp + 1 < NG) { SG_MFMA(1); SG_DSR(1); SG_MFMA(1); SG_DSR(1); SG_MFMA(1); SG_DSR(1); SG_MFMA(1); SG_DSR(1); }
                    else { SG_MFMA(1); SG_DSR(2); SG_MFMA(1); SG_DSR(2); SG_MFMA(1); SG_DSR(2); SG_MFMA(1); SG_DSR(2); }
                    __builtin_amdgcn_sched_barrier(0);
                    // the next step's pieces behind the groups: the Q image under query half 0, the dO image under query half 1 (the dV pass has four groups a half: two pieces each)
                    if constexpr (DO_K) issue_piece(qb, gp);
                    else { issue_piece(qb, 2 * gp); issue_piece(qb, 2 * gp + 1); }
                    if (gp + 1 < NG) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) cf[j] = nf[j];
                    }
                }
                APROF(0);
                float pr[16], ds[16];
                const bool diag = t0 + qb * 32 < wave_kmin + 32;              // some query of the block may precede some key of the wave
                if (!diag && !tail_tile) {                                      // (uniform) the plain case: every query of the half sees every key of the wave that is not padding
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float e = kvis ? fast_exp2(__builtin_fmaf(s[r], sc, -lse4[r >> 2][r & 3])) : 0.f;
                        pr[r] = e;
                        if constexpr (DO_K) ds[r] = e * (dp[r] - dl4[r >> 2][r & 3]) * A.scale;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ql = qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;     // query inside the tile
                        const bool vis = kvis & (!diag | (ki <= t0 + ql)) & (!tail_tile | (t0 + ql < A.S));
                        const float e = vis ? fast_exp2(__builtin_fmaf(s[r], sc, -lse4[r >> 2][r & 3])) : 0.f;
                        pr[r] = e;
                        if constexpr (DO_K) ds[r] = e * (dp[r] - dl4[r >> 2][r & 3]) * A.scale;
                    }
                }
                const bf16x8 f0 = frag_from_acc(DO_V ? &pr[0] : &ds[0]), f1 = frag_from_acc(DO_V ? &pr[8] : &ds[8]);
                if constexpr (DO_K) {
                    if (A.pbuf) {                                                // (uniform) the probabilities as the dV products' fragments: 2 x 1 KiB a wave and query half, coalesced
                        const long long piece = ((((((long long)b * A.Hq + (g * G + head_lo + hq_cur)) * ((A.S + 127) / 128) + kblk) * ((A.S + 63) / 64) + t0 / 64) * 4 + wave) * 2 + qb) * 2;
                        bf16x8 *dst = reinterpret_cast<bf16x8 *>(A.pbuf) + piece * 64 + lane;
                        dst[0] = frag_from_acc(&pr[0]);
                        dst[64] = frag_from_acc(&pr[8]);
                    }
                }
                APROF(1);
#pragma unroll
                for (int dpair = 0; dpair < D / 64; ++dpair) {
                    bf16x8 ntf[2][2];
                    __builtin_amdgcn_sched_barrier(0);
                    if (dpair + 1 < D / 64) load_tr(ntf, cur, qb, dpair + 1);
                    else if (qb == 0) load_grp(cf, cur, 1, 0);                  // (query half 1 of the same tile; the next step's first group waits for the barrier)
                    acc[2 * dpair] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[0][0], f0, acc[2 * dpair], 0, 0, 0);
                    acc[2 * dpair + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[0][1], f0, acc[2 * dpair + 1], 0, 0, 0);
                    acc[2 * dpair] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[1][0], f1, acc[2 * dpair], 0, 0, 0);
                    acc[2 * dpair + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[1][1], f1, acc[2 * dpair + 1], 0, 0, 0);
                    if (dpair + 1 < D / 64) { SG_MFMA(1); SG_DSR(2); SG_MFMA(1); SG_DSR(2); SG_MFMA(1); SG_DSR(2); SG_MFMA(1); SG_DSR(2); }
                    else if (qb == 0) { SG_MFMA(1); SG_DSR(1); SG_MFMA(1); SG_DSR(1); SG_MFMA(1); SG_DSR(1); SG_MFMA(1); SG_DSR(1); }
                    __builtin_amdgcn_sched_barrier(0);
                    if (dpair + 1 < D / 64) {
#pragma unroll
                        for (int s2 = 0; s2 < 2; ++s2) { tf[s2][0] = ntf[s2][0]; tf[s2][1] = ntf[s2][1]; }
                    }
                }
                APROF(2);
            }
#ifdef ECGB_PROFILE
            prof_acc[6] += 1;
#endif
        } else {
#pragma unroll
            for (int i = 0; i < PPW; ++i) { issue_piece(0, i); issue_piece(1, i); }
        }
        advance_next();
        APROF(3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        APROF(4);
        __builtin_amdgcn_s_barrier();
        img = img ? 0u : (unsigned)kStage;
        if (t0 + 64 >= t_begin + tiles_per_head * 64) { t0 = t_begin; ++hq_cur; } else t0 += 64;
    }
#ifdef ECGB_PROFILE
    if ((threadIdx.x & 63) == 0)
        for (int kk = 0; kk < 7; ++kk) atomicAdd(&g_attn_prof[(DO_K ? 0 : 32) + (threadIdx.x >> 6) * 8 + kk], prof_acc[kk]);
#endif
    if (A.head_splits > 1) {                                 // partial sums over this workgroup's heads: fp32 slabs, reduced in order afterwards
        const long long rows_all = (long long)A.B * A.S, slab = rows_all * A.Hkv * D;
        float *base = A.slab + (long long)hsplit * slab + ((long long)g * rows_all) * D;
        if constexpr (DO_K) store_accT_f32<NB>(acc, base, D, rowbase + ki, kvalid, h);
        if constexpr (DO_V) store_accT_f32<NB>(acc, base + (long long)A.head_splits * slab, D, rowbase + ki, kvalid, h);
        return;
    }
    if constexpr (DO_K) store_accT<NB>(acc, A.dk + (long long)g * D, A.lddk, rowbase + ki, kvalid, h, 1.f);
    if constexpr (DO_V) store_accT<NB>(acc, A.dv + (long long)g * D, A.lddv, rowbase + ki, kvalid, h, 1.f);
}
// head_dim 256 dV from the probabilities the dK pass left (A.pbuf): dV = sum over the group's heads and the query tiles of dO^T . P for a block of 128 keys, a wave 32 keys.
// No scores, no softmax, no K: a step is the tile's dO image by LDS-DMA (a ring of two), the wave's P fragments by two 16-byte loads a query half (a step ahead, in
// registers) and 32 products whose transposing reads issue under the products of the group before.  The products and their order are the pair kernel's dV pass: the same bits.
__global__ __launch_bounds__(256, 1) void attn_bwd_dv_d256_kernel(AttnArgs A)
{
    constexpr int D = 256, kRow = D * 2, kTile = 64 * kRow, PPW = 8, NB = D / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];                         // 2 x the dO image
    const int hs = A.head_splits > 1 ? A.head_splits : 1;
    int blk, head_in, group;
    map_block((int)blockIdx.x, ((A.S + 127) / 128) * hs, 1, A.B * A.Hkv, false, blk, head_in, group);
    const int hsplit = blk % hs, kblk = blk / hs, b = group / A.Hkv, g = group % A.Hkv;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, lr = lane & 31, h = lane >> 5;
    const int G = A.Hq / A.Hkv;
    const int kk0 = kblk * 128;
    const int ki = kk0 + wave * 32 + lr;
    const bool kvalid = ki < A.S;
    const long long rowbase = (long long)b * A.S;
    const int wave_kmin = kk0 + wave * 32;
    const int t_begin = (kk0 / 64) * 64;
    const int tiles_per_head = (A.S - t_begin + 63) / 64;
    const int heads_here = G / hs, head_lo = hsplit * heads_here;
    const int n_steps = heads_here * tiles_per_head;
    const int tail_rows = A.S - (t_begin + (tiles_per_head - 1) * 64);
    const int r0 = wave * (2 * PPW) + (lane >> 5);
    const unsigned chunk0 = (unsigned)((((lane & 31) ^ ((lane >> 5) << 2)) * 8) * 2);
    const unsigned ldo2 = (unsigned)(A.ldo * 2);
    int hq_next = 0, ti_next = 0, left_next = n_steps - 1;
    unsigned slot_next = 0;
    const unsigned char *o_next;
    const bf16x8 *p_next;                                    // the next step's P pieces of this wave
    int rmax_next, t0_next;
    auto point_next = [&]() {
        const int hq = g * G + head_lo + hq_next;
        t0_next = t_begin + ti_next * 64;
        o_next = reinterpret_cast<const unsigned char *>(A.d_o + (long long)hq * D + (rowbase + t0_next) * A.ldo);
        p_next = reinterpret_cast<const bf16x8 *>(A.pbuf) + ((((((long long)b * A.Hq + hq) * ((A.S + 127) / 128) + kblk) * ((A.S + 63) / 64) + t0_next / 64) * 4 + wave) * 4) * 64 + lane;
        rmax_next = (ti_next == tiles_per_head - 1 && tail_rows < 64) ? tail_rows - 1 : 63;
    };
    point_next();
    auto issue_piece = [&](int i) __attribute__((always_inline)) {
        const unsigned off = (unsigned)min(r0 + 2 * i, rmax_next) * ldo2 + (chunk0 ^ (unsigned)((((i & 1) << 3) | (i >> 1)) << 4));
        lds_dma16(smem + slot_next + (wave * PPW + i) * 1024, o_next, off);
    };
    bf16x8 pn[4] = {};                                       // the next step's fragments: [query half][k-step]; a step this wave skips left no P behind and asks for none
    auto load_p = [&]() __attribute__((always_inline)) {
        if (t0_next + 63 >= wave_kmin) {
#pragma unroll
            for (int j = 0; j < 4; ++j) pn[j] = p_next[j * 64];
        }
    };
    auto advance_next = [&]() {
        if (left_next > 0) {
            --left_next;
            if (++ti_next == tiles_per_head) { ti_next = 0; ++hq_next; }
            point_next();
        }
        slot_next ^= (unsigned)kTile;
    };
#pragma unroll
    for (int i = 0; i < PPW; ++i) issue_piece(i);
    load_p();
    advance_next();
    f32x16 acc[NB];
#pragma unroll
    for (int db = 0; db < NB; ++db) acc[db] = splat16(0.f);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned tbaseA, tbaseB;
    {
        const int a = lr >> 4, q = (lr & 15) >> 2, p = lr & 3;
        const int cg = 2 * a + (p >> 1);
        const int k1 = 4 * h + q, k2 = k1 + 8;
        tbaseA = k1 * kRow + ((((k1 & 3) << 2) | (cg ^ ((k1 >> 2) & 3))) << 4) + (p & 1) * 8;
        tbaseB = k2 * kRow + ((((k2 & 3) << 2) | (cg ^ ((k2 >> 2) & 3))) << 4) + (p & 1) * 8;
    }
    auto load_tr = [&](bf16x8 (&f)[2][2], const unsigned char *cur, int qb, int dpair) __attribute__((always_inline)) {
        const unsigned x0 = (unsigned)((2 * dpair) << 6), x1 = (unsigned)((2 * dpair + 1) << 6);
        unsigned ta = tbaseA, tb = tbaseB;
        asm volatile("" : "+v"(ta), "+v"(tb));
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const unsigned off = (qb * 32 + s2 * 16) * kRow;
            f[s2][0] = lds_tr_frag_c(cur + (ta ^ x0) + off, cur + (tb ^ x0) + off);
            f[s2][1] = lds_tr_frag_c(cur + (ta ^ x1) + off, cur + (tb ^ x1) + off);
        }
    };
    unsigned img = 0;
    int t0 = t_begin;
    for (int step = 0; step < n_steps; ++step) {
        const unsigned char *cur = smem + img;
#pragma unroll
        for (int db = 0; db < NB; ++db) asm volatile("" : "+v"(acc[db]));           // (see the pair kernel)
        bf16x8 pc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) pc[j] = pn[j];
        const bool active = t0 + 63 >= wave_kmin;
        load_p();                                            // the step after this one (its loads and this step's DMA are both waited for at the step's end)
        if (active) {
            bf16x8 tf[2][2];
            load_tr(tf, cur, 0, 0);
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
                for (int dpair = 0; dpair < D / 64; ++dpair) {
                    bf16x8 ntf[2][2];
                    const bool more = dpair + 1 < D / 64 || qb == 0;
                    __builtin_amdgcn_sched_barrier(0);
                    if (dpair + 1 < D / 64) load_tr(ntf, cur, qb, dpair + 1);
                    else if (qb == 0) load_tr(ntf, cur, 1, 0);
                    acc[2 * dpair] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[0][0], pc[2 * qb], acc[2 * dpair], 0, 0, 0);
                    acc[2 * dpair + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[0][1], pc[2 * qb], acc[2 * dpair + 1], 0, 0, 0);
                    acc[2 * dpair] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[1][0], pc[2 * qb + 1], acc[2 * dpair], 0, 0, 0);
                    acc[2 * dpair + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[1][1], pc[2 * qb + 1], acc[2 * dpair + 1], 0, 0, 0);
                    if (more) { SG_MFMA(1); SG_DSR(2); SG_MFMA(1); SG_DSR(2); SG_MFMA(1); SG_DSR(2); SG_MFMA(1); SG_DSR(2); }
                    __builtin_amdgcn_sched_barrier(0);
                    issue_piece(qb * 4 + dpair);             // the next step's dO pieces behind the groups
                    if (more) {
#pragma unroll
                        for (int s2 = 0; s2 < 2; ++s2) { tf[s2][0] = ntf[s2][0]; tf[s2][1] = ntf[s2][1]; }
                    }
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < PPW; ++i) issue_piece(i);
        }
        advance_next();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        img ^= (unsigned)kTile;
        t0 = t0 + 64 >= t_begin + tiles_per_head * 64 ? t_begin : t0 + 64;
    }
    if (A.head_splits > 1) {
        const long long rows_all = (long long)A.B * A.S, slab = rows_all * A.Hkv * D;
        float *base = A.slab + (long long)hsplit * slab + ((long long)g * rows_all) * D;
        store_accT_f32<NB>(acc, base + (long long)A.head_splits * slab, D, rowbase + ki, kvalid, h);
        return;
    }
    store_accT<NB>(acc, A.dv + (long long)g * D, A.lddv, rowbase + ki, kvalid, h, 1.f);
}
// the dK pass alone (with its P stores: A.pbuf), one workgroup a (key block, head split, batch row and KV head)
__global__ __launch_bounds__(256, 1) void attn_bwd_dk_d256_kernel(AttnArgs A)
{
    const int hs = A.head_splits > 1 ? A.head_splits : 1;
    int blk, head_in, group;
    map_block((int)blockIdx.x, ((A.S + 127) / 128) * hs, 1, A.B * A.Hkv, false, blk, head_in, group);
    attn_bwd_dkv_d256_body<2>(A, blk / hs, group / A.Hkv, group % A.Hkv, blk % hs);
}
__global__ __launch_bounds__(256, 1) void attn_bwd_dkv_pair_d256_kernel(AttnArgs A)
{
    const int hs = A.head_splits > 1 ? A.head_splits : 1;      // (the block order of attn_bwd_dkv_pair_kernel)
    int blk, head_in, group;
    map_block((int)blockIdx.x, ((A.S + 127) / 128) * 2 * hs, 1, A.B * A.Hkv, false, blk, head_in, group);
    const int pass = blk & 1, hsplit = (blk >> 1) % hs, kblk = (blk >> 1) / hs;
    if (pass) attn_bwd_dkv_d256_body<2>(A, kblk, group / A.Hkv, group % A.Hkv, hsplit);
    else attn_bwd_dkv_d256_body<1>(A, kblk, group / A.Hkv, group % A.Hkv, hsplit);
}

int launched256(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) return ECGB_OK;
    ecgb::set_error(std::string(what) + ": " + hipGetErrorString(e));
    return ECGB_ERR_HIP;
}
template <typename Kern>
int launch256(Kern kern, unsigned grid, int lds, void *stream, const AttnArgs &A, const char *what)
{
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) {
        ecgb::set_error(std::string(what) + ": hipFuncSetAttribute(" + std::to_string(lds) + " bytes of dynamic LDS): " + hipGetErrorString(e));
        return ECGB_ERR_HIP;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, (hipStream_t)stream, A);
    return launched256(what);
}

}  // namespace

#ifdef ECGB_PROFILE
// dev builds only (scripts/dev_prof_attn_d256.py): this file's phase timers
extern "C" void ecgb_debug_attn256_profile(unsigned long long *out64, int reset)
{
    if (out64) (void)hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_attn_prof), 64 * sizeof(unsigned long long));
    if (reset) { unsigned long long z[64] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_attn_prof), z, sizeof z); }
}
#endif

namespace ecgb_attn {
// dynamic LDS: the two-stage ring, the row's key mask, the tile flags
int launch_bwd_dq_d256(const AttnArgs &A, unsigned grid, int seq, void *stream)
{
    return launch256(attn_bwd_dq_d256_kernel, grid, 4 * 64 * 512 + 4 * ((seq + 63) & ~63) + 256, stream, A, "attn_bwd_dq_d256_kernel");
}
int launch_bwd_dk_then_dv_d256(const AttnArgs &A, unsigned grid_each, void *stream)
{
    const int rc = launch256(attn_bwd_dk_d256_kernel, grid_each, 2 * (2 * 64 * 512 + 512), stream, A, "attn_bwd_dk_d256_kernel");
    if (rc) return rc;
    return launch256(attn_bwd_dv_d256_kernel, grid_each, 2 * 64 * 512, stream, A, "attn_bwd_dv_d256_kernel");
}
int launch_bwd_dkv_pair_d256(const AttnArgs &A, unsigned grid, void *stream)
{
    return launch256(attn_bwd_dkv_pair_d256_kernel, grid, 2 * (2 * 64 * 512 + 512), stream, A, "attn_bwd_dkv_pair_d256_kernel");
}
}  // namespace ecgb_attn
