// attention.hip -- fused causal grouped-query attention for MI355X (gfx950): forward, dQ, dK/dV.
//
// Reference: LlamaSdpaAttention.forward + the 4-D causal/left-padding mask
// (transformers/src/transformers/models/llama/modeling_llama.py:526-614, 981-1100): key j is visible
// to query i iff j <= i and attention_mask[b, j] != 0; softmax in fp32, P cast to bf16 for P.V.
// Rows without a visible key (left-padding rows) produce zeros (don't-care in the reference).
//
// No S x S tensor ever exists in HBM.  One workgroup = 4 waves = 128 rows of one (batch, head);
// K/V (or Q/dO) tiles of 64 rows are staged in LDS, once plain and once transposed.
// The MFMA orientation is chosen so that the softmax row index sits on the LANE in every product:
//   S^T = K . Q^T           A = K rows from LDS, B = Q rows held in registers   -> acc[key rows][q lanes]
//   O^T = V^T . P^T         A = V^T rows from LDS, B = the S^T accumulator itself (cast to bf16)
// so row max / row sum / rescale are per-lane scalars (one cross-half shuffle each), and the
// accumulator of the first product is the B operand of the second with NO data movement: element
// j of a 32x32x16 B fragment built from accumulator registers 8s..8s+7 is k = 16s + 8(j>>2) + 4h + (j&3)
// (h = lane >> 5), and the transposed LDS tile is read in that same k order (two 8-byte reads).
// The backward kernels reuse the scheme with the roles permuted (lanes = queries for dQ, lanes = keys
// for dK/dV, which loops over the query heads of its KV group, so no atomics anywhere).
#include "attention_common.inc"

namespace {

template <int D, bool DMA = false>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs A)
{
    static_assert(!DMA || D == 64, "the LDS-DMA staging is written for 128-byte rows");
    // two K / V^T tile buffers: the tile after the one being multiplied is written while the others still compute,
    // one barrier per tile.  Dynamic LDS (4 x 128 D bytes + 4 bytes per key: 36 KB at head_dim 64 and S 1024, 139 KB at 256 / 2048).
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int kTile = 128 * D;                           // bytes of a 64-row tile, plain or transposed
    auto lds_k2 = [&](int i) { return smem + i * (DMA ? 2 * kTile : kTile); };
    auto lds_vt2 = [&](int i) { return DMA ? smem + i * 2 * kTile + kTile : smem + (2 + i) * kTile; };
    float *lds_maskrow = reinterpret_cast<float *>(smem + (DMA ? 6 : 4) * kTile);        // the batch row's key mask up to this block's last key
    const int wave = DMA ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : (int)(threadIdx.x >> 6), lane = threadIdx.x & 63, lr = lane & 31, h = lane >> 5;
    int qblk, head_in, group;
    map_block((int)blockIdx.x, (A.S + 127) / 128, A.Hq / A.Hkv, A.B * A.Hkv, true, qblk, head_in, group);
    const int b = group / A.Hkv, g = group % A.Hkv, hq = g * (A.Hq / A.Hkv) + head_in;
    const int q0 = qblk * 128;
    const int qi = q0 + wave * 32 + lr;
    const bool qvalid = qi < A.S;
    const long long rowbase = (long long)b * A.S;
    const unsigned short *Q = A.q + (long long)hq * D, *K = A.k + (long long)g * D, *V = A.v + (long long)g * D;
    bf16x8 qf[D / 16];
    load_row_frags<D>(qf, Q, A.ldq, rowbase + qi, qvalid, h);
    f32x16 accO[D / 32];
#pragma unroll
    for (int db = 0; db < D / 32; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) accO[db][r] = 0.f;
    float m = -INFINITY, l = 0.f;
    const float sc = A.scale * kLog2e;
    const int k_end = min(A.S, q0 + 128);
    const int wave_qmax = q0 + wave * 32 + 31;
    constexpr int NI = D / 64;                               // staging items per thread: a tile is 16 row groups x D / 8 chunks
    const int item = threadIdx.x & 127;
    const bool first_half = threadIdx.x < 128;
    const unsigned short *src = first_half ? K + rowbase * A.ldk : V + rowbase * A.ldv;
    const long long src_ld = first_half ? A.ldk : A.ldv;
    Stage4 st[NI];
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NI; ++i) stage_load<D>(st[i], src, src_ld, k0, A.S, item + 128 * i);
    };
    auto write_tile = [&](int buf, int k0) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (first_half) stage_write_plain<D>(lds_k2(buf), st[i], item + 128 * i); else stage_write_transposed(lds_vt2(buf), st[i], item + 128 * i);
        }
    };
    // DMA: a tile is 8 instructions of 1 KiB (8 rows of 128 B) per operand, two of each per wave; per-lane byte offsets inside a tile are
    // constants (whole tiles) or clamped to the last key (the one partial tile a sequence can end with); the tile's base is scalar
    const int last_tile = (k_end - 1) / 64;
    unsigned offK[2], offV[2], offKt[2], offVt[2];
    if constexpr (DMA) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = (wave * 2 + i) * 8 + (lane >> 3), slot = lane & 7;
            const int ck = slot ^ ((r >> 1) & 7), cv = slot ^ (((r >> 1) & 1) << 2);
            const int rt = min(r, A.S - 1 - last_tile * 64);                 // tail tile: rows past the last key re-read it (masked: keys >= S)
            offK[i] = (unsigned)(((long long)r * A.ldk + ck * 8) * 2);  offKt[i] = (unsigned)(((long long)rt * A.ldk + ck * 8) * 2);
            offV[i] = (unsigned)(((long long)r * A.ldv + cv * 8) * 2);  offVt[i] = (unsigned)(((long long)rt * A.ldv + cv * 8) * 2);
        }
    }
    auto issue_tile = [&](int t, int buf) {                  // UNCONDITIONAL (the caller clamps t): the waits below stay counted
        const bool tail = (t + 1) * 64 > A.S;
        const unsigned char *kb = reinterpret_cast<const unsigned char *>(K + (rowbase + (long long)t * 64) * A.ldk);
        const unsigned char *vb = reinterpret_cast<const unsigned char *>(V + (rowbase + (long long)t * 64) * A.ldv);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(kb + (tail ? offKt[i] : offK[i])),
                                             (__attribute__((address_space(3))) void *)(lds_k2(buf) + (wave * 2 + i) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(vb + (tail ? offVt[i] : offV[i])),
                                             (__attribute__((address_space(3))) void *)(lds_vt2(buf) + (wave * 2 + i) * 1024), 16, 0, 0);
        }
    };
    // key mask: the whole row goes to LDS once (keys >= S read as padded).  Fetched per tile by wave 0 right where it was written, the
    // load's memory latency (~1 700 cycles) sat in front of every tile's barrier: a quarter of the kernel.
    if constexpr (DMA) {
        issue_tile(0, 0);
        issue_tile(min(1, last_tile), 1);
        for (int i = threadIdx.x; i < ((k_end + 63) & ~63); i += 256) lds_maskrow[i] = (i < A.S) ? A.mask[rowbase + i] : 0.f;
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // tile 0 (and Q) landed; tile 1 may still be in flight
        __syncthreads();                                     // (drains: the price of one barrier with full waits, once per workgroup)
    } else {
        load_tile(0);
        for (int i = threadIdx.x; i < ((k_end + 63) & ~63); i += 256) lds_maskrow[i] = (i < A.S) ? A.mask[rowbase + i] : 0.f;
        write_tile(0, 0);
        __syncthreads();
    }
    // DMA: this lane's transposing-read address inside a V image for d block 0 / 1 (first read; the second is 8 keys = 1 KiB further down; the other
    // (kb, s2) row groups go into the instruction's offset field): 16-byte chunk c of key row r lies at c ^ 4 ((r >> 1) & 1)
    unsigned vtr[2] = {0, 0};
    if constexpr (DMA) {
        const int a = lr >> 4, q = (lr & 15) >> 2, p = lr & 3;
        const int key = 4 * h + q;
#pragma unroll
        for (int db = 0; db < 2; ++db)
            vtr[db] = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem + kTile + key * 128 +
                      (((db * 4 + 2 * a + (p >> 1)) ^ (((key >> 1) & 1) << 2)) << 4) + (p & 1) * 8;
    }
#ifdef ECGB_PROFILE
    unsigned long long prof_acc[7] = {};
    long long t_prof = clock64();
#endif
    for (int k0 = 0, it = 0; k0 < k_end; k0 += 64, ++it) {
        const bool more = k0 + 64 < k_end;
        const unsigned vimg = DMA ? (unsigned)((it % 3) * 2 * kTile) : 0u;
        if constexpr (DMA) issue_tile(min(it + 2, last_tile), (it + 2) % 3);      // two tiles ahead, into the buffer tile it - 1 left at the last barrier
        else if (more) load_tile(k0 + 64);   // next tile in flight behind the MFMAs
        APROF(0);
        const unsigned char *lds_k = lds_k2(DMA ? it % 3 : it & 1), *lds_vt = lds_vt2(DMA ? it % 3 : it & 1);
        const float *lds_mask = lds_maskrow + k0;
        if (k0 <= wave_qmax) {
            const bool lds_flag = __any(lds_mask[lane] == 0.f);          // the tile holds a padded / out-of-range key
            // masks only matter on tiles that touch the diagonal of this wave's rows or hold padded keys: two code paths chosen per
            // wave and tile (one `if` around the score loop alone is if-converted by hipcc: every tile then paid the ~130 compare /
            // select instructions of the masked form, a third of the loop's vector instructions)
            // (head_dim 128 / 256: one path with the choice inside -- two copies of the wider loop body cost more registers than a
            // wave has, and the MFMA share of a tile is 2-4 x larger there)
            const bool need_mask = (k0 + 63 > q0 + wave * 32) || lds_flag;
            auto tile = [&](auto mode_tag) {
                constexpr int MODE = decltype(mode_tag)::value;          // 0 = no masks, 1 = masks, 2 = decided here
                float p[2][16];
                float tmax = -INFINITY;
                f32x16 sacc[2];                                     // the two key halves' chains interleaved: an MFMA never
#pragma unroll                                                      // waits for the one issued just before it
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sacc[kb][r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < D / 16; ++ks)
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
                        sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_plain<D>(lds_k, kb * 32 + lr, ks, h), qf[ks], sacc[kb], 0, 0, 0);
                // DMA: the eight V^T fragments of the tile are read now, behind the issue of the S products (the wait inside runs under them)
                bf16x8 vfr[2][2][2];                                // [kb][s2][db]
                if constexpr (DMA) {
                    tr_frags4_wait<0, 16 * 128>(vfr[0], vtr[0] + vimg, vtr[0] + vimg + 1024, vtr[1] + vimg, vtr[1] + vimg + 1024);
                    tr_frags4_wait<32 * 128, 48 * 128>(vfr[1], vtr[0] + vimg, vtr[0] + vimg + 1024, vtr[1] + vimg, vtr[1] + vimg + 1024);
                }
                // max3() below is inline asm: hipcc's hazard recogniser does not look inside it, and a vector instruction that reads an
                // MFMA result needs 19 wait states after a 16-pass MFMA (nothing interlocks: the first version read stale registers now
                // and then -- a slightly different running maximum, a softmax that differed in the last bits from run to run).  The
                // accumulators pass through this statement, so everything after it is ordered behind the wait.
                if constexpr (MODE == 0) asm volatile("s_nop 15\n\ts_nop 3" : "+v"(sacc[0]), "+v"(sacc[1]));
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    const f32x16 &s = sacc[kb];
                    if (MODE == 1 || (MODE == 2 && need_mask)) {
                        float4 mk4[4];                                           // the lane's 16 keys are 4 runs of 4: four 16-byte reads
                        if constexpr (DMA) {
                            f4v mv[4];
                            const unsigned ma = (unsigned)(size_t)(__attribute__((address_space(3))) float *)const_cast<float *>(lds_mask) + 16 * h;
                            if (kb == 0) lds_rows4_wait<0>(mv, ma); else lds_rows4_wait<128>(mv, ma);
#pragma unroll
                            for (int g4 = 0; g4 < 4; ++g4) mk4[g4] = make_float4(mv[g4][0], mv[g4][1], mv[g4][2], mv[g4][3]);
                        } else {
#pragma unroll
                            for (int g4 = 0; g4 < 4; ++g4) mk4[g4] = *reinterpret_cast<const float4 *>(&lds_mask[kb * 32 + 8 * g4 + 4 * h]);
                        }
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int kl = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;       // key inside the tile
                            const float mkv = (r & 3) == 0 ? mk4[r >> 2].x : (r & 3) == 1 ? mk4[r >> 2].y : (r & 3) == 2 ? mk4[r >> 2].z : mk4[r >> 2].w;
                            const bool vis = (k0 + kl <= qi) & (mkv != 0.f);
                            const float v = vis ? s[r] : -INFINITY;
                            p[kb][r] = v;
                            tmax = fmaxf(tmax, v);
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; r += 2) { p[kb][r] = s[r]; p[kb][r + 1] = s[r + 1]; tmax = max3(s[r], s[r + 1], tmax); }
                    }
                }
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64)) * sc;      // p holds raw scores; sc > 0, so the maximum scales with them
                APROF(1);
                const float m_new = fmaxf(m, tmax);
                const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;   // no key visible yet: every p below is exp2(-inf) = 0
                const float alpha = fast_exp2(m - m_safe);                     // m = -inf -> 0 (accumulators are still zero then)
                float lsum = 0.f;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { const float e = fast_exp2(__builtin_fmaf(p[kb][r], sc, -m_safe)); p[kb][r] = e; lsum += e; }
                l = l * alpha + lsum;
                m = m_new;
                if (__any(alpha != 1.f)) {   // once the running maxima have settled the accumulators need no rescale
#pragma unroll
                    for (int db = 0; db < D / 32; ++db)
#pragma unroll
                        for (int r = 0; r < 16; ++r) accO[db][r] *= alpha;
                }
                APROF(2);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        const bf16x8 pf = frag_from_acc(&p[kb][8 * s2]);
#pragma unroll
                        for (int db = 0; db < D / 32; ++db)
                            accO[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(DMA ? vfr[kb][s2][db & 1] : frag_transposed(lds_vt, db * 32 + lr, kb, s2, h),
                                                                               pf, accO[db], 0, 0, 0);
                    }
                APROF(3);
            };
            if constexpr (D > 64) tile(std::integral_constant<int, 2>{});
            else if (need_mask) tile(std::integral_constant<int, 1>{});
            else tile(std::integral_constant<int, 0>{});
        }
        if constexpr (DMA) {
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // this wave's pieces of tile it + 1 have landed (tile it + 2's four may still fly)
            APROF(4);
            __builtin_amdgcn_s_barrier();                      // ... and everybody else's; all reads of tile it are done
        } else {
            if (more) write_tile((it + 1) & 1, k0 + 64);   // its last readers passed the barrier that ended the previous trip
            APROF(4);
            __syncthreads();
        }
        APROF(5);
#ifdef ECGB_PROFILE
        prof_acc[6] += 1;
#endif
    }
#ifdef ECGB_PROFILE
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 7; ++k) atomicAdd(&g_attn_prof[(threadIdx.x >> 6) * 8 + k], prof_acc[k]);
#endif
    const float lt = l + __shfl_xor(l, 32, 64);
    const float inv = lt > 0.f ? 1.f / lt : 0.f;
    store_accT<D / 32>(accO, A.o + (long long)hq * D, A.ldo, rowbase + qi, qvalid, h, inv);
    if (qvalid && h == 0) A.lse[((long long)b * A.Hq + hq) * A.S + qi] = lt > 0.f ? m + log2f(lt) : INFINITY;
}

// =====================================================================================================
// backward, dQ (and delta): 1-D grid of ceil(S/128) x Hq x B workgroups (map_block); lanes = queries
template <int D>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 3 x 128 D bytes + 4 bytes per key
    constexpr int kTile = 128 * D;
    unsigned char *lds_k = smem, *lds_v = smem + kTile, *lds_kt = smem + 2 * kTile;
    float *lds_maskrow = reinterpret_cast<float *>(smem + 3 * kTile);        // the batch row's key mask up to this block's last key
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 31, h = lane >> 5;
    int qblk, head_in, group;
    map_block((int)blockIdx.x, (A.S + 127) / 128, A.Hq / A.Hkv, A.B * A.Hkv, true, qblk, head_in, group);
    const int b = group / A.Hkv, g = group % A.Hkv, hq = g * (A.Hq / A.Hkv) + head_in;
    const int q0 = qblk * 128;
    const int qi = q0 + wave * 32 + lr;
    const bool qvalid = qi < A.S;
    const long long rowbase = (long long)b * A.S;
    const unsigned short *Q = A.q + (long long)hq * D, *K = A.k + (long long)g * D, *V = A.v + (long long)g * D;
    bf16x8 qf[D / 16], dof[D / 16], of[D / 16];
    load_row_frags<D>(qf, Q, A.ldq, rowbase + qi, qvalid, h);
    load_row_frags<D>(dof, A.d_o + (long long)hq * D, A.ldo, rowbase + qi, qvalid, h);
    load_row_frags<D>(of, A.o + (long long)hq * D, A.ldo, rowbase + qi, qvalid, h);
    float delta = 0.f;
#pragma unroll
    for (int ks = 0; ks < D / 16; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) delta += bf2f((unsigned short)dof[ks][j]) * bf2f((unsigned short)of[ks][j]);
    delta += __shfl_xor(delta, 32, 64);
    const long long stat = ((long long)b * A.Hq + hq) * A.S + qi;
    if (qvalid && h == 0) A.delta[stat] = delta;
    const float lse = qvalid ? A.lse[stat] : INFINITY;
    f32x16 accQ[D / 32];
#pragma unroll
    for (int db = 0; db < D / 32; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) accQ[db][r] = 0.f;
    const float sc = A.scale * kLog2e;
    const int k_end = min(A.S, q0 + 128);
    const int wave_qmax = q0 + wave * 32 + 31;
    constexpr int NI = D / 64;
    const int item = threadIdx.x & 127;
    const bool first_half = threadIdx.x < 128;
    const unsigned short *src = first_half ? K + rowbase * A.ldk : V + rowbase * A.ldv;
    const long long src_ld = first_half ? A.ldk : A.ldv;
    Stage4 st[NI];
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NI; ++i) stage_load<D>(st[i], src, src_ld, k0, A.S, item + 128 * i);
    };
    load_tile(0);
    for (int i = threadIdx.x; i < ((k_end + 63) & ~63); i += 256) lds_maskrow[i] = (i < A.S) ? A.mask[rowbase + i] : 0.f;   // see attn_fwd_kernel
    for (int k0 = 0; k0 < k_end; k0 += 64) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (first_half) { stage_write_plain<D>(lds_k, st[i], item + 128 * i); stage_write_transposed(lds_kt, st[i], item + 128 * i); }
            else stage_write_plain<D>(lds_v, st[i], item + 128 * i);
        }
        if (k0 + 64 < k_end) load_tile(k0 + 64);
        __syncthreads();
        if (k0 > wave_qmax) continue;
        const float *lds_mask = lds_maskrow + k0;
        const bool need_mask = (k0 + 63 > q0 + wave * 32) || __any(lds_mask[lane] == 0.f);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < D / 16; ++ks) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_plain<D>(lds_k, kb * 32 + lr, ks, h), qf[ks], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_plain<D>(lds_v, kb * 32 + lr, ks, h), dof[ks], dp, 0, 0, 0);
            }
            float ds[16];
            if (need_mask) {
                float4 mk4[4];
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) mk4[g4] = *reinterpret_cast<const float4 *>(&lds_mask[kb * 32 + 8 * g4 + 4 * h]);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kl = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const float mkv = (r & 3) == 0 ? mk4[r >> 2].x : (r & 3) == 1 ? mk4[r >> 2].y : (r & 3) == 2 ? mk4[r >> 2].z : mk4[r >> 2].w;
                    const bool vis = (k0 + kl <= qi) & (mkv != 0.f);
                    const float pr = vis ? fast_exp2(__builtin_fmaf(s[r], sc, -lse)) : 0.f;
                    ds[r] = pr * (dp[r] - delta) * A.scale;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) ds[r] = fast_exp2(__builtin_fmaf(s[r], sc, -lse)) * (dp[r] - delta) * A.scale;
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 dsf = frag_from_acc(&ds[8 * s2]);
#pragma unroll
                for (int db = 0; db < D / 32; ++db)
                    accQ[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_transposed(lds_kt, db * 32 + lr, kb, s2, h), dsf, accQ[db], 0, 0, 0);
            }
        }
    }
    store_accT<D / 32>(accQ, A.dq + (long long)hq * D, A.lddq, rowbase + qi, qvalid, h, 1.f);
}

// dQ with LDS-DMA staging (head_dim 64): K and V tiles through a ring of three buffers, two tiles ahead, counted waits (see attn_fwd_kernel<64, true>);
// K is read both ways from one image (row fragments for S, transposing reads for dS . K).  The arithmetic of attn_bwd_dq_kernel<64> except that the softmax scale
// multiplies dQ once at the store instead of every dS element: the same bits for head_dim 64's scale of 1/8 (a power of two commutes with the bf16 rounding of dS).
__global__ __launch_bounds__(256) void attn_bwd_dq_dma_kernel(AttnArgs A)
{
    constexpr int D = 64, kTile = 128 * D;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 3 x (K, V) + 4 bytes per key
    auto lds_k3 = [&](int i) { return smem + i * 2 * kTile; };
    auto lds_v3 = [&](int i) { return smem + i * 2 * kTile + kTile; };
    float *lds_maskrow = reinterpret_cast<float *>(smem + 6 * kTile);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, lr = lane & 31, h = lane >> 5;
    int qblk, head_in, group;
    map_block((int)blockIdx.x, (A.S + 127) / 128, A.Hq / A.Hkv, A.B * A.Hkv, true, qblk, head_in, group);
    const int b = group / A.Hkv, g = group % A.Hkv, hq = g * (A.Hq / A.Hkv) + head_in;
    const int q0 = qblk * 128;
    const int qi = q0 + wave * 32 + lr;
    const bool qvalid = qi < A.S;
    const long long rowbase = (long long)b * A.S;
    const unsigned short *Q = A.q + (long long)hq * D, *K = A.k + (long long)g * D, *V = A.v + (long long)g * D;
    bf16x8 qf[D / 16], dof[D / 16], of[D / 16];
    load_row_frags<D>(qf, Q, A.ldq, rowbase + qi, qvalid, h);
    load_row_frags<D>(dof, A.d_o + (long long)hq * D, A.ldo, rowbase + qi, qvalid, h);
    load_row_frags<D>(of, A.o + (long long)hq * D, A.ldo, rowbase + qi, qvalid, h);
    float delta = 0.f;
#pragma unroll
    for (int ks = 0; ks < D / 16; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) delta += bf2f((unsigned short)dof[ks][j]) * bf2f((unsigned short)of[ks][j]);
    delta += __shfl_xor(delta, 32, 64);
    const long long stat = ((long long)b * A.Hq + hq) * A.S + qi;
    if (qvalid && h == 0) A.delta[stat] = delta;
    const float lse = qvalid ? A.lse[stat] : INFINITY;
    f32x16 accQ[D / 32];
#pragma unroll
    for (int db = 0; db < D / 32; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) accQ[db][r] = 0.f;
    const float sc = A.scale * kLog2e;
    const int k_end = min(A.S, q0 + 128);
    const int wave_qmax = q0 + wave * 32 + 31;
    const int last_tile = (k_end - 1) / 64;
    unsigned offK[2], offV[2], offKt[2], offVt[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (wave * 2 + i) * 8 + (lane >> 3), c = (lane & 7) ^ swz_u(r);
        const int rt = min(r, A.S - 1 - last_tile * 64);                 // tail tile: rows past the last key re-read it (masked: keys >= S)
        offK[i] = (unsigned)(((long long)r * A.ldk + c * 8) * 2);  offKt[i] = (unsigned)(((long long)rt * A.ldk + c * 8) * 2);
        offV[i] = (unsigned)(((long long)r * A.ldv + c * 8) * 2);  offVt[i] = (unsigned)(((long long)rt * A.ldv + c * 8) * 2);
    }
    auto issue_tile = [&](int t, int buf) {                  // UNCONDITIONAL (the caller clamps t): the waits below stay counted
        const bool tail = (t + 1) * 64 > A.S;
        const unsigned char *kb = reinterpret_cast<const unsigned char *>(K + (rowbase + (long long)t * 64) * A.ldk);
        const unsigned char *vb = reinterpret_cast<const unsigned char *>(V + (rowbase + (long long)t * 64) * A.ldv);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(kb + (tail ? offKt[i] : offK[i])),
                                             (__attribute__((address_space(3))) void *)(lds_k3(buf) + (wave * 2 + i) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(vb + (tail ? offVt[i] : offV[i])),
                                             (__attribute__((address_space(3))) void *)(lds_v3(buf) + (wave * 2 + i) * 1024), 16, 0, 0);
        }
    };
    issue_tile(0, 0);
    issue_tile(min(1, last_tile), 1);
    for (int i = threadIdx.x; i < ((k_end + 63) & ~63); i += 256) lds_maskrow[i] = (i < A.S) ? A.mask[rowbase + i] : 0.f;   // see attn_fwd_kernel
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __syncthreads();
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
    unsigned trA[D / 32], trB[D / 32];                       // this lane's transposing-read offsets inside a K image (first / second read)
#pragma unroll
    for (int db = 0; db < D / 32; ++db) { trA[db] = lds0 + tr_off_u(db, lr, h, 0); trB[db] = lds0 + tr_off_u(db, lr, h, 8); }
    for (int k0 = 0, it = 0; k0 < k_end; k0 += 64, ++it) {
        issue_tile(min(it + 2, last_tile), (it + 2) % 3);
        const int buf = it % 3;
        const unsigned char *lds_k = lds_k3(buf), *lds_v = lds_v3(buf);
        const unsigned kimg = (unsigned)(buf * 2 * kTile);
        if (k0 <= wave_qmax) {
            const float *lds_mask = lds_maskrow + k0;
            const bool need_mask = (k0 + 63 > q0 + wave * 32) || __any(lds_mask[lane] == 0.f);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                f32x16 s, dp;
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
                for (int ks = 0; ks < D / 16; ++ks) {
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_plain_u(lds_k, kb * 32 + lr, ks, h), qf[ks], s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_plain_u(lds_v, kb * 32 + lr, ks, h), dof[ks], dp, 0, 0, 0);
                }
                bf16x8 ktf[2][2];                            // K^T fragments of this key half [s2][db]: read (and waited for) under the products above
                if (kb == 0) tr_frags4_wait<0, 16 * 128>(ktf, trA[0] + kimg, trB[0] + kimg, trA[1] + kimg, trB[1] + kimg);
                else tr_frags4_wait<32 * 128, 48 * 128>(ktf, trA[0] + kimg, trB[0] + kimg, trA[1] + kimg, trB[1] + kimg);
                float ds[16];
                if (need_mask) {
                    f4v mv[4];
                    const unsigned ma = (unsigned)(size_t)(__attribute__((address_space(3))) float *)const_cast<float *>(lds_mask) + 16 * h;
                    if (kb == 0) lds_rows4_wait<0>(mv, ma); else lds_rows4_wait<128>(mv, ma);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int kl = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        const float mkv = mv[r >> 2][r & 3];
                        const bool vis = (k0 + kl <= qi) & (mkv != 0.f);
                        const float pr = vis ? fast_exp2(__builtin_fmaf(s[r], sc, -lse)) : 0.f;
                        ds[r] = pr * (dp[r] - delta);                                // the softmax scale multiplies dQ once, at the store
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) ds[r] = fast_exp2(__builtin_fmaf(s[r], sc, -lse)) * (dp[r] - delta);
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const bf16x8 dsf = frag_from_acc(&ds[8 * s2]);
#pragma unroll
                    for (int db = 0; db < D / 32; ++db)
                        accQ[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf[s2][db], dsf, accQ[db], 0, 0, 0);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // this wave's pieces of tile it + 1 have landed (tile it + 2's four may still fly)
        __builtin_amdgcn_s_barrier();                        // ... and everybody else's; all reads of tile it are done
    }
    store_accT<D / 32>(accQ, A.dq + (long long)hq * D, A.lddq, rowbase + qi, qvalid, h, A.scale);
}

// =====================================================================================================
// backward, dK and dV: 1-D grid of ceil(S/128) x Hkv x B workgroups (map_block); lanes = keys; loops over the query heads of the group
// DS > 1 (head_dim 256): the d range of the dK / dV accumulators is split over DS workgroups (blockIdx.x % DS): K, V fragments and
// two full accumulator sets of head_dim 256 do not fit one wave's 512 registers; the S and dP products are recomputed per split.
// WHICH: 0 = dK and dV in one pass (head_dim 64 / 128); 1 = dV only, 2 = dK only -- at head_dim 256 the two gradients are two launches,
// each with full-width accumulators: the dV pass needs S alone (K fragments), the dK pass S and dP (K and V fragments), 1.25 x the MFMA
// work of a single pass instead of the 2.5 x of splitting the d range four ways (DS = 4, the first head_dim-256 version: 3.3 ms per
// layer at Gemma-2B's shape).
template <int D, int DS, int WHICH>
__device__ __forceinline__ void attn_bwd_dkv_body(const AttnArgs &A, const int block_x, const int b, const int g, const int hsplit = 0)
{
    constexpr bool DO_V = WHICH != 2, DO_K = WHICH != 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 4 x 128 D bytes + 512
    constexpr int kTile = 128 * D;
    unsigned char *lds_q = smem, *lds_do = smem + kTile, *lds_qt = smem + 2 * kTile, *lds_dot = smem + 3 * kTile;
    float *lds_lse = reinterpret_cast<float *>(smem + 4 * kTile), *lds_delta = lds_lse + 64;
    constexpr int NB = D / 32 / DS;                          // 32-wide d blocks this workgroup accumulates
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 31, h = lane >> 5;
    const int G = A.Hq / A.Hkv;
    const int kk0 = (block_x / DS) * 128, db0 = (block_x % DS) * NB;
    const int ki = kk0 + wave * 32 + lr;
    const bool kvalid = ki < A.S;
    const long long rowbase = (long long)b * A.S;
    bf16x8 kf[D / 16], vf[D / 16];
    load_row_frags<D>(kf, A.k + (long long)g * D, A.ldk, rowbase + ki, kvalid, h);
    if constexpr (DO_K) load_row_frags<D>(vf, A.v + (long long)g * D, A.ldv, rowbase + ki, kvalid, h);
    const bool kvis = kvalid && A.mask[rowbase + (kvalid ? ki : 0)] != 0.f;
    f32x16 accK[NB], accV[NB];
#pragma unroll
    for (int db = 0; db < NB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) { accK[db][r] = 0.f; accV[db][r] = 0.f; }
    const float sc = A.scale * kLog2e;
    const int wave_kmin = kk0 + wave * 32;
    constexpr int NI = D / 64;
    const int item = threadIdx.x & 127;
    const bool first_half = threadIdx.x < 128;            // first half stages Q, second half dO
    const int t_begin = (kk0 / 64) * 64;                   // first query tile that can see this key block
    const int tiles_per_head = (A.S - t_begin + 63) / 64;
    const int heads_here = G / (A.head_splits > 1 ? A.head_splits : 1), head_lo = hsplit * heads_here;   // this workgroup's query heads of the group
    const int n_steps = heads_here * tiles_per_head;
    auto src_of = [&](int step, int &t0, long long &stat) -> const unsigned short * {
        const int hq = g * G + head_lo + step / tiles_per_head;
        t0 = t_begin + (step % tiles_per_head) * 64;
        stat = ((long long)b * A.Hq + hq) * A.S;
        return first_half ? A.q + (long long)hq * D + rowbase * A.ldq : A.d_o + (long long)hq * D + rowbase * A.ldo;
    };
    const long long src_ld = first_half ? A.ldq : A.ldo;
    Stage4 st[NI];
    float lse_n = INFINITY, delta_n = 0.f;      // wave 0: the tile's row statistics; head_dim 64: fetched with the tile, one tile ahead
    auto load_step = [&](int step) {
        int t0; long long stat;
        const unsigned short *src = src_of(step, t0, stat);
#pragma unroll
        for (int i = 0; i < NI; ++i) stage_load<D>(st[i], src, src_ld, t0, A.S, item + 128 * i);
        if (D == 64 && threadIdx.x < 64) {
            const bool v = t0 + (int)threadIdx.x < A.S;
            lse_n = v ? A.lse[stat + t0 + threadIdx.x] : INFINITY;
            delta_n = v ? A.delta[stat + t0 + threadIdx.x] : 0.f;
        }
    };
    load_step(0);
    for (int step = 0; step < n_steps; ++step) {
        int t0; long long stat;
        (void)src_of(step, t0, stat);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (first_half) { stage_write_plain<D>(lds_q, st[i], item + 128 * i); if constexpr (DO_K) stage_write_transposed(lds_qt, st[i], item + 128 * i); }
            else { if constexpr (DO_K) stage_write_plain<D>(lds_do, st[i], item + 128 * i); if constexpr (DO_V) stage_write_transposed(lds_dot, st[i], item + 128 * i); }
        }
        if (threadIdx.x < 64) {
            if (D != 64) {                             // wider heads: measured slower a tile ahead (registers)
                const bool v = t0 + (int)threadIdx.x < A.S;
                lse_n = v ? A.lse[stat + t0 + threadIdx.x] : INFINITY;
                delta_n = v ? A.delta[stat + t0 + threadIdx.x] : 0.f;
            }
            lds_lse[threadIdx.x] = lse_n;
            lds_delta[threadIdx.x] = delta_n;
        }
        if (step + 1 < n_steps) load_step(step + 1);      // next tile in flight behind the MFMAs
        __syncthreads();
        if (t0 + 63 < wave_kmin) continue;        // every query of the tile precedes every key of this wave
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < D / 16; ++ks) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_plain<D>(lds_q, qb * 32 + lr, ks, h), kf[ks], s, 0, 0, 0);
                if constexpr (DO_K) dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_plain<D>(lds_do, qb * 32 + lr, ks, h), vf[ks], dp, 0, 0, 0);
            }
            float pr[16], ds[16];
            const bool diag = t0 + qb * 32 < wave_kmin + 32;              // some query of the block may precede some key of the wave
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {                                // the lane's 16 queries are 4 runs of 4: 16-byte reads
                const float4 lse4 = *reinterpret_cast<const float4 *>(&lds_lse[qb * 32 + 8 * g4 + 4 * h]);
                const float4 dl4 = *reinterpret_cast<const float4 *>(&lds_delta[qb * 32 + 8 * g4 + 4 * h]);
                const float lse_t[4] = {lse4.x, lse4.y, lse4.z, lse4.w}, dl_t[4] = {dl4.x, dl4.y, dl4.z, dl4.w};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int r = 4 * g4 + t;
                    const int ql = qb * 32 + t + 8 * g4 + 4 * h;             // query inside the tile
                    const bool vis = kvis & (!diag | (ki <= t0 + ql));
                    const float e = vis ? fast_exp2(__builtin_fmaf(s[r], sc, -lse_t[t])) : 0.f;
                    pr[r] = e;
                    ds[r] = e * (dp[r] - dl_t[t]) * A.scale;
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 pf = frag_from_acc(&pr[8 * s2]);
                const bf16x8 dsf = frag_from_acc(&ds[8 * s2]);
#pragma unroll
                for (int db = 0; db < NB; ++db) {
                    if constexpr (DO_V) accV[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_transposed(lds_dot, (db0 + db) * 32 + lr, qb, s2, h), pf, accV[db], 0, 0, 0);
                    if constexpr (DO_K) accK[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_transposed(lds_qt, (db0 + db) * 32 + lr, qb, s2, h), dsf, accK[db], 0, 0, 0);
                }
            }
        }
    }
    if (A.head_splits > 1) {                                 // partial sums over this workgroup's heads: fp32 slabs, reduced in order afterwards
        const long long rows_all = (long long)A.B * A.S, slab = rows_all * A.Hkv * D;
        float *base = A.slab + (long long)hsplit * slab + ((long long)g * rows_all) * D + db0 * 32;
        if constexpr (DO_K) store_accT_f32<NB>(accK, base, D, rowbase + ki, kvalid, h);
        if constexpr (DO_V) store_accT_f32<NB>(accV, base + (long long)A.head_splits * slab, D, rowbase + ki, kvalid, h);
        return;
    }
    if constexpr (DO_K) store_accT<NB>(accK, A.dk + (long long)g * D + db0 * 32, A.lddk, rowbase + ki, kvalid, h, 1.f);
    if constexpr (DO_V) store_accT<NB>(accV, A.dv + (long long)g * D + db0 * 32, A.lddv, rowbase + ki, kvalid, h, 1.f);
}

// dK and dV with LDS-DMA staging (head_dim 64): the Q and dO tiles of a step (one query head, 64 queries) and the tile's row statistics (log-sum-exp,
// delta: 256 bytes each, 4 bytes a lane) go global -> LDS through a ring of three buffers, two steps ahead; every wave issues the same five
// instructions per step (two Q pieces, two dO pieces, one statistics row: waves 0 / 1 bring lse / delta, 2 / 3 a copy nobody reads), so one counted
// wait serves all.  Each tile is read both ways from its one image.  A tile that ends past the sequence re-reads the last row (DMA cannot zero):
// those queries are switched off in the visibility test.  The arithmetic of attn_bwd_dkv_kernel<64, 1, 0> except that the softmax scale multiplies dK once at the
// store: the same bits for head_dim 64's scale of 1/8.
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_dma_kernel(AttnArgs A)
{
    constexpr int D = 64, kTile = 128 * D, kStep = 2 * kTile + 1024;      // Q image, dO image, four statistics rows
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 3 x kStep
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, lr = lane & 31, h = lane >> 5;
    int blk, head_in, group;
    map_block((int)blockIdx.x, (A.S + 127) / 128, 1, A.B * A.Hkv, false, blk, head_in, group);   // key block 0 sees every query: first
    const int b = group / A.Hkv, g = group % A.Hkv;
    const int G = A.Hq / A.Hkv;
    const int kk0 = blk * 128;
    const int ki = kk0 + wave * 32 + lr;
    const bool kvalid = ki < A.S;
    const long long rowbase = (long long)b * A.S;
    bf16x8 kf[D / 16], vf[D / 16];
    load_row_frags<D>(kf, A.k + (long long)g * D, A.ldk, rowbase + ki, kvalid, h);
    load_row_frags<D>(vf, A.v + (long long)g * D, A.ldv, rowbase + ki, kvalid, h);
    const bool kvis = kvalid && A.mask[rowbase + (kvalid ? ki : 0)] != 0.f;
    // query q sees this lane's key iff the key is real and ki <= q < S: one unsigned compare, (q - vis_lo) < vis_n with vis_lo = ki (S for a padded /
    // out-of-range key: vis_n = 0, never true).  (Causal test, key padding and the clamped rows of a tile past the end were two compares, an or and
    // scalar mask arithmetic per element: 133 of a step's 344 vector instructions.)
    const int vis_lo = kvis ? ki : A.S;
    const unsigned vis_n = (unsigned)(A.S - vis_lo);
    f32x16 accK[D / 32], accV[D / 32];
#pragma unroll
    for (int db = 0; db < D / 32; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) { accK[db][r] = 0.f; accV[db][r] = 0.f; }
    const float sc = A.scale * kLog2e;
    const int wave_kmin = kk0 + wave * 32;
    const int t_begin = (kk0 / 64) * 64;                   // first query tile that can see this key block
    const int tiles_per_head = (A.S - t_begin + 63) / 64;
    const int n_steps = G * tiles_per_head;
    const int tail_rows = A.S - (t_begin + (tiles_per_head - 1) * 64);    // rows of a head's last tile that exist (1..64)
    unsigned offQ[2], offO[2], offQt[2], offOt[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (wave * 2 + i) * 8 + (lane >> 3), c = (lane & 7) ^ swz_u(r);
        const int rt = min(r, tail_rows - 1);
        offQ[i] = (unsigned)(((long long)r * A.ldq + c * 8) * 2);  offQt[i] = (unsigned)(((long long)rt * A.ldq + c * 8) * 2);
        offO[i] = (unsigned)(((long long)r * A.ldo + c * 8) * 2);  offOt[i] = (unsigned)(((long long)rt * A.ldo + c * 8) * 2);
    }
    const unsigned offS = (unsigned)lane * 4u, offSt = (unsigned)min(lane, tail_rows - 1) * 4u;
    const float *stat_src = (wave & 1) ? A.delta : A.lse;
    auto issue_step = [&](int step, int buf) {               // UNCONDITIONAL (the caller clamps step): the waits below stay counted
        const int hq = g * G + step / tiles_per_head;
        const int ti = step % tiles_per_head;
        const int t0 = t_begin + ti * 64;
        const bool tail = ti == tiles_per_head - 1 && tail_rows < 64;
        const unsigned char *qb = reinterpret_cast<const unsigned char *>(A.q + (long long)hq * D + (rowbase + t0) * A.ldq);
        const unsigned char *ob = reinterpret_cast<const unsigned char *>(A.d_o + (long long)hq * D + (rowbase + t0) * A.ldo);
        const unsigned char *sb = reinterpret_cast<const unsigned char *>(stat_src + ((long long)b * A.Hq + hq) * A.S + t0);
        unsigned char *dst = smem + buf * kStep;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(qb + (tail ? offQt[i] : offQ[i])),
                                             (__attribute__((address_space(3))) void *)(dst + (wave * 2 + i) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(ob + (tail ? offOt[i] : offO[i])),
                                             (__attribute__((address_space(3))) void *)(dst + kTile + (wave * 2 + i) * 1024), 16, 0, 0);
        }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(sb + (tail ? offSt : offS)),
                                         (__attribute__((address_space(3))) void *)(dst + 2 * kTile + wave * 256), 4, 0, 0);
    };
    issue_step(0, 0);
    issue_step(min(1, n_steps - 1), 1);
    asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    __syncthreads();
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
    unsigned trA[D / 32], trB[D / 32];
#pragma unroll
    for (int db = 0; db < D / 32; ++db) { trA[db] = lds0 + tr_off_u(db, lr, h, 0); trB[db] = lds0 + tr_off_u(db, lr, h, 8); }
    const unsigned statA = lds0 + 2 * kTile + 16 * h;        // + 32 g4 + 128 qb: the lane's four runs of four queries
    for (int step = 0; step < n_steps; ++step) {
        issue_step(min(step + 2, n_steps - 1), (step + 2) % 3);
        const int buf = step % 3;
        const unsigned char *lds_q = smem + buf * kStep, *lds_do = lds_q + kTile;
        const unsigned img = (unsigned)(buf * kStep);
        const int t0 = t_begin + (step % tiles_per_head) * 64;
        if (t0 + 63 >= wave_kmin) {                            // else: every query of the tile precedes every key of this wave
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                f32x16 s, dp;
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
                for (int ks = 0; ks < D / 16; ++ks) {
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_plain_u(lds_q, qb * 32 + lr, ks, h), kf[ks], s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_plain_u(lds_do, qb * 32 + lr, ks, h), vf[ks], dp, 0, 0, 0);
                }
                // statistics and the transposed fragments of this query half: read (and waited for) under the products above
                f4v lse4[4], dl4[4];
                bf16x8 qtf[2][2], dotf[2][2];                // [s2][db]
                if (qb == 0) {
                    lds_rows4_wait<0>(lse4, statA + img);
                    lds_rows4_wait<256>(dl4, statA + img);
                    tr_frags4_wait<kTile, kTile + 16 * 128>(dotf, trA[0] + img, trB[0] + img, trA[1] + img, trB[1] + img);
                    tr_frags4_wait<0, 16 * 128>(qtf, trA[0] + img, trB[0] + img, trA[1] + img, trB[1] + img);
                } else {
                    lds_rows4_wait<128>(lse4, statA + img);
                    lds_rows4_wait<256 + 128>(dl4, statA + img);
                    tr_frags4_wait<kTile + 32 * 128, kTile + 48 * 128>(dotf, trA[0] + img, trB[0] + img, trA[1] + img, trB[1] + img);
                    tr_frags4_wait<32 * 128, 48 * 128>(qtf, trA[0] + img, trB[0] + img, trA[1] + img, trB[1] + img);
                }
                float pr[16], ds[16];
                const int qrel = t0 + 4 * h - vis_lo;                            // (query of element (g4, t)) - vis_lo = qrel + 32 qb + 8 g4 + t
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int r = 4 * g4 + t;
                        const bool vis = (unsigned)(qrel + (qb * 32 + 8 * g4 + t)) < vis_n;
                        const float e = vis ? fast_exp2(__builtin_fmaf(s[r], sc, -lse4[g4][t])) : 0.f;
                        pr[r] = e;
                        ds[r] = e * (dp[r] - dl4[g4][t]);                        // the softmax scale multiplies dK once, at the store
                    }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const bf16x8 pf = frag_from_acc(&pr[8 * s2]);
                    const bf16x8 dsf = frag_from_acc(&ds[8 * s2]);
#pragma unroll
                    for (int db = 0; db < D / 32; ++db) {
                        accV[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dotf[s2][db], pf, accV[db], 0, 0, 0);
                        accK[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf[s2][db], dsf, accK[db], 0, 0, 0);
                    }
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(5)" ::: "memory");     // this wave's pieces of step + 1 have landed (step + 2's five may still fly)
        __builtin_amdgcn_s_barrier();                        // ... and everybody else's; all reads of this step's images are done
    }
    store_accT<D / 32>(accK, A.dk + (long long)g * D, A.lddk, rowbase + ki, kvalid, h, A.scale);
    store_accT<D / 32>(accV, A.dv + (long long)g * D, A.lddv, rowbase + ki, kvalid, h, 1.f);
}

// =====================================================================================================
// Round 3: the head_dim-64 kernels again, with the softmax arithmetic moved into the MFMA accumulators ("lean" kernels, g_attn_dma == 2).
// The LDS-DMA kernels above are vector-issue bound, not MFMA bound (ISA count of attn_fwd_kernel<64, true>: ~200 vector instructions beside 16
// MFMAs per 32 x 64 tile = ~930 issue cycles against 512 of MFMA; dK/dV 211 beside 32).  What a score costs besides its exp is removed:
//   * the row operand held in registers is PRE-SCALED by scale * log2(e) (rounded to bf16 once per row, as the persistent-workgroup attention of the
//     CDNA guide does): the products are log2-domain scores, no multiply per score;
//   * the row constant of the softmax is the INITIAL ACCUMULATOR of the score product (MFMA takes C != D): forward -- minus the row's running
//     maximum, so p = exp2(acc) with no subtraction; backward -- the row's log-sum-exp with the register operand negated, acc = lse - s, p = exp2(-acc)
//     (the negation is the instruction's source modifier); dP starts from delta the same way: acc = delta - dP, and ds = -(p * acc): one multiply, the
//     sign goes into the scale applied once at the store;
//   * forward: the running maximum is DEFERRED (T13 of the guide).  A tile takes the fast path -- no maximum at all, 32 exp + 32 adds + 16 packs --
//     unless some row has not seen a key yet or some lane's tile sum exceeds 2^10 (a score more than ~2^5..2^10 above the row's reference);
//     then the exact path: row maximum, rescale of l and O, new reference.  Tiles that need the causal / padding mask always take the exact path.
// The bf16 rounding points of P and dS are the ones of the kernels above; what changes is the rounding of the pre-scaled operand (half an ulp
// of bf16 on q resp. k: the size of the input's own quantisation) and the order of the fp32 sums inside the MFMA.  lse / delta keep their meaning
// and layout, so every lean kernel can be mixed with the others (tests A/B each one alone).
// =====================================================================================================

// row operand fragments, pre-scaled: bf16(x * mul) (mul = +-scale * log2 e, or -1 for a sign flip, which is exact)
__device__ __forceinline__ void scale_row_frags(bf16x8 (&f)[4], float mul)
{
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        using u4 = __attribute__((ext_vector_type(4))) unsigned;
        u4 w;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            w[j] = pack_bf16(bf2f((unsigned short)f[ks][2 * j]) * mul, bf2f((unsigned short)f[ks][2 * j + 1]) * mul);
        f[ks] = __builtin_bit_cast(bf16x8, w);
    }
}
// the fragments negated (exact: the sign bit of each bf16): one XOR per dword where scale_row_frags(f, -1.f) spent an unpack, a multiply and a pack on every pair
__device__ __forceinline__ void negate_row_frags(bf16x8 (&f)[4])
{
    using u4 = __attribute__((ext_vector_type(4))) unsigned;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) f[ks] = __builtin_bit_cast(bf16x8, __builtin_bit_cast(u4, f[ks]) ^ 0x80008000u);
}

#ifndef ECGB_LEAN_DIAG
#define ECGB_LEAN_DIAG 0      // timing-only diagnostics of the forward kernel (wrong results): 1 no tile DMA in the loop, 2 no barrier, 4 no exp, 8 two of the eight P.V MFMAs, 16 four of the eight Q.K MFMAs, 32 no row sum
#endif
#ifndef ECGB_LEAN_FWD_WGS
#define ECGB_LEAN_FWD_WGS 3      // waves per SIMD the forward kernel is compiled for (168 registers; 2: 206).  Three: -9 ... -14 % (see the kernel)
#endif
#ifndef ECGB_LEAN_DQ_WGS
#define ECGB_LEAN_DQ_WGS 2
#endif
#ifndef ECGB_LEAN_DKV_WGS
#define ECGB_LEAN_DKV_WGS 2
#endif
#ifndef ECGB_LEAN_PRIO
#define ECGB_LEAN_PRIO 0      // 1: s_setprio 1 around the forward kernel's MFMA clusters (measured: no change, 0.248 against 0.250 ms)
#endif
#if ECGB_LEAN_PRIO
#define LEAN_PRIO(p) __builtin_amdgcn_s_setprio(p)
#else
#define LEAN_PRIO(p) do { } while (0)
#endif
constexpr int kLeanRing = 3;      // LDS ring depth of the lean kernels: tiles (kRing - 1) ahead (4 measured the same as 3: the flight time is not what a wave waits for)
template <int N> __device__ __forceinline__ void lean_wait_tiles() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }


// NW waves per workgroup (4 or 8) share every K / V tile.  Measured with the phase timers (scripts/dev_prof_attn.py, 4 waves x 32 rows, two workgroups per CU):
// 2 030 of a tile's 4 700 cycles went into ISSUING its four LDS-DMA pieces -- the vector memory path of a CU takes about 16 bytes per cycle, 32 KB per pair of
// workgroup tiles against 1 024 cycles of MFMA: the kernel was bound by the tile traffic, not by arithmetic.  So a tile serves twice the rows: eight waves,
// and -- grouped-query attention -- the waves of a workgroup are the HW = gcd(G, 8) query heads of the KV group times 8 / HW blocks of 32 rows (Llama: 4 heads
// x 64 rows, where both row blocks need exactly the same tiles: no wave idles at a barrier under the causal mask; one head: 256 rows).
template <int NW>
__global__ __launch_bounds__(NW * 64, ECGB_LEAN_FWD_WGS) void attn_fwd_lean_kernel(AttnArgs A)
{
    constexpr int D = 64, kTile = 128 * D, PPW = 8 / NW;      // PPW: 1 KiB pieces of a tile per wave and operand
    // Ring of kRing (K, V) tile pairs, kAhead = kRing - 1 tiles in flight, issued FIRST thing in a tile.  Measured (scripts/dev_attn_lean.py, C3 shape):
    // issuing tile it + 2 in the middle of tile it (in the vector-only stretch, where the guide prices an LDS-DMA issue lowest) cost 15 % of the kernel; a ring of
    // four (three tiles ahead) measured the same as three.
    constexpr int kRing = kLeanRing, kAhead = kRing - 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];     // kRing x (K, V) tiles + the row's key mask
    float *lds_maskrow = reinterpret_cast<float *>(smem + 2 * kRing * kTile);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, lr = lane & 31, h = lane >> 5;
    const int hw_log2 = A.lean_hw_log2, R = (NW >> hw_log2) * 32;          // heads per workgroup (log2), query rows per workgroup
    int qblk, head_in, group;
    map_block((int)blockIdx.x, (A.S + R - 1) / R, (A.Hq / A.Hkv) >> hw_log2, A.B * A.Hkv, true, qblk, head_in, group);
    const int b = group / A.Hkv, g = group % A.Hkv, hq = g * (A.Hq / A.Hkv) + (head_in << hw_log2) + (wave & ((1 << hw_log2) - 1));
    const int q0 = qblk * R, qw0 = q0 + (wave >> hw_log2) * 32;            // first row of the workgroup / of this wave
    const int qi = qw0 + lr;
    const bool qvalid = qi < A.S;
    const long long rowbase = (long long)b * A.S;
    const unsigned short *Q = A.q + (long long)hq * D, *K = A.k + (long long)g * D, *V = A.v + (long long)g * D;
    const int k_end = min(A.S, q0 + R);
    const int wave_qmax = qw0 + 31;
    const int last_tile = (k_end - 1) / 64;
    unsigned offK[PPW], offV[PPW], offKt[PPW], offVt[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int r = (wave * PPW + i) * 8 + (lane >> 3), slot = lane & 7;
        const int ck = slot ^ ((r >> 1) & 7), cv = slot ^ (((r >> 1) & 1) << 2);
        const int rt = min(r, A.S - 1 - last_tile * 64);
        offK[i] = (unsigned)(((long long)r * A.ldk + ck * 8) * 2);  offKt[i] = (unsigned)(((long long)rt * A.ldk + ck * 8) * 2);
        offV[i] = (unsigned)(((long long)r * A.ldv + cv * 8) * 2);  offVt[i] = (unsigned)(((long long)rt * A.ldv + cv * 8) * 2);
    }
    // The tile stream as RUNNING state (round 4): the next tile's global bases, its index and its ring slot advance by adds.  Formed from the tile index every trip
    // (64-bit multiplies, a division by the ring depth) the issue was 50 scalar instructions per trip, a fifth of the trip's instruction count -- and a wave issues
    // one instruction per ~4.5 cycles whatever its kind.  UNCONDITIONAL (past the last tile the last one is issued again): the waits below stay counted.
    const unsigned char *kb_next = reinterpret_cast<const unsigned char *>(K + rowbase * A.ldk), *vb_next = reinterpret_cast<const unsigned char *>(V + rowbase * A.ldv);
    const long long stepK = 128ll * A.ldk, stepV = 128ll * A.ldv;       // bytes from one 64-key tile to the next
    const bool tail_last = (last_tile + 1) * 64 > A.S;
    int t_next = 0;
    unsigned slot_next = 0;                                             // byte offset of the ring slot the next tile lands in
    auto issue_next = [&]() {
        const bool tail = tail_last && t_next == last_tile;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            unsigned ok = offK[i], ov = offV[i];
            if (tail) {                                          // (uniform, the one partial tile a sequence can end with) rows past the last key re-read it
                const int r = (wave * PPW + i) * 8 + (lane >> 3), slot = lane & 7, rt = min(r, A.S - 1 - last_tile * 64);
                ok = (unsigned)(((long long)rt * A.ldk + (slot ^ ((r >> 1) & 7)) * 8) * 2);
                ov = (unsigned)(((long long)rt * A.ldv + (slot ^ (((r >> 1) & 1) << 2)) * 8) * 2);
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(kb_next + ok),
                                             (__attribute__((address_space(3))) void *)(smem + slot_next + (wave * PPW + i) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(vb_next + ov),
                                             (__attribute__((address_space(3))) void *)(smem + slot_next + kTile + (wave * PPW + i) * 1024), 16, 0, 0);
        }
        if (t_next < last_tile) { ++t_next; kb_next += stepK; vb_next += stepV; }
        slot_next = slot_next == (kRing - 1) * 2 * kTile ? 0u : slot_next + 2 * kTile;
    };
#pragma unroll
    for (int t = 0; t < kAhead; ++t) issue_next();
    bf16x8 qf[4];
    load_row_frags<D>(qf, Q, A.ldq, rowbase + qi, qvalid, h);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // Q's fragments (younger than the tiles' pieces) are in: from here on the waits are counted
    scale_row_frags(qf, A.scale * kLog2e);                   // log2-domain scores straight out of the MFMA
    lean_fill_mask<NW>(lds_maskrow, A.mask + rowbase, A.S, (k_end + 63) & ~63);
    f32x16 accO[2] = {splat16(0.f), splat16(0.f)};
    float m = -INFINITY, l = 0.f;                            // m: the row's reference (a deferred running maximum); -inf = no key seen yet
    f32x16 cinit = splat16(0.f);                             // -m in all sixteen registers: the initial accumulator of the score products
    __syncthreads();
    const unsigned long long padbits = lean_pad_bits(lds_maskrow, (k_end + 63) & ~63);
    unsigned vtr[2];
    {
        const int a = lr >> 4, q = (lr & 15) >> 2, p = lr & 3;
        const int key = 4 * h + q;
#pragma unroll
        for (int db = 0; db < 2; ++db)
            vtr[db] = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem + kTile + key * 128 +
                      (((db * 4 + 2 * a + (p >> 1)) ^ (((key >> 1) & 1) << 2)) << 4) + (p & 1) * 8;
    }
    const unsigned lds_base_u = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
    unsigned kbase[4];                                       // this lane's K row fragment (key lr, k-step ks) inside a K image; key 32 + lr is 4 KiB further
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
        kbase[ks] = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem + lr * 128 + (((ks * 2 + h) ^ ((lr >> 1) & 7)) << 4);
#ifdef ECGB_PROFILE
    unsigned long long prof_acc[7] = {};
    long long t_prof = clock64();
#endif
    unsigned vimg = 0;                                       // byte offset of the ring slot of the current tile
    for (int k0 = 0, it = 0; k0 < k_end; k0 += 64, ++it) {
        const float *lds_mask = lds_maskrow + k0;
#if !(ECGB_LEAN_DIAG & 1)
        issue_next();                                        // FIRST thing in the tile: the pieces need their whole flight time (see kRing)
#endif
        APROF(0);
        if (k0 <= wave_qmax) {
            const bool need_mask = (k0 + 63 > qw0) || lean_tile_padded(padbits, it);
            f32x16 sacc[2];
            bf16x8 kfr[2][4];                                   // [kb][ks]: K row fragments, four per batch; the second batch and the V^T fragments land under MFMAs
            const unsigned kb0 = (unsigned)(lr * 128 + ((h ^ ((lr >> 1) & 7)) << 4)), kl0 = lds_base_u + vimg;      // k-step ks is the offset ^ (ks << 5): recomputed, not held
            lds_frags2x2_wait<0, 4096>(kfr[0][0], kfr[1][0], kfr[0][1], kfr[1][1], kb0 + kl0, (kb0 ^ 32u) + kl0);
            LEAN_PRIO(1);
            sacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[0][0], qf[0], cinit, 0, 0, 0);
            sacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[1][0], qf[0], cinit, 0, 0, 0);
            sacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[0][1], qf[1], sacc[0], 0, 0, 0);
            sacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[1][1], qf[1], sacc[1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            lds_frags2x2_wait<0, 4096>(kfr[0][2], kfr[1][2], kfr[0][3], kfr[1][3], (kb0 ^ 64u) + kl0, (kb0 ^ 96u) + kl0);
#if !(ECGB_LEAN_DIAG & 16)
#pragma unroll
            for (int ks = 2; ks < 4; ++ks)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
                    sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[kb][ks], qf[ks], sacc[kb], 0, 0, 0);
#endif
            __builtin_amdgcn_sched_barrier(0);
            bf16x8 vfr[2][2][2];                                // [kb][s2][db]: the tile's eight V^T fragments
            APROF(1);
            LEAN_PRIO(0);
            bf16x8 pf[2][2];                                    // [kb][s2]: P as the B operand of P.V, packed as it is produced (the fp32 values are not kept)
            using u4 = __attribute__((ext_vector_type(4))) unsigned;
            bool exact = need_mask;
            if (!need_mask) {                                   // (uniform) fast path: p = exp2(score - reference) is the accumulator's exp2
                // the row sum as FOUR chains of eight adds (one per fragment of P): as one chain of 32 dependent adds it was 13 % of the kernel -- with two waves
                // per SIMD nothing else of this wave can issue behind an add that waits for the previous one
                float ls[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        u4 w;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
#if ECGB_LEAN_DIAG & 4
                            const float e0 = sacc[kb][8 * s2 + 2 * j], e1 = sacc[kb][8 * s2 + 2 * j + 1];
#else
                            const float e0 = fast_exp2(sacc[kb][8 * s2 + 2 * j]), e1 = fast_exp2(sacc[kb][8 * s2 + 2 * j + 1]);
#endif
#if !(ECGB_LEAN_DIAG & 32)
                            ls[kb][s2] += e0; ls[kb][s2] += e1;
#endif
                            w[j] = pack_bf16(e0, e1);
                        }
                        pf[kb][s2] = __builtin_bit_cast(bf16x8, w);
                    }
                const float lsum = (ls[0][0] + ls[0][1]) + (ls[1][0] + ls[1][1]);
                // not (lsum <= 2^10) also catches +inf; a row that has seen no key yet has no reference: exact path
                exact = __any(!(lsum <= 1024.f) || m == -INFINITY);
                if (!exact) l += lsum;
            }
            if (exact) {
                float t = -INFINITY;
                if (need_mask) {
                    f4v mv[2][4];
                    const unsigned ma = (unsigned)(size_t)(__attribute__((address_space(3))) float *)const_cast<float *>(lds_mask) + 16 * h;
                    lds_rows4_wait<0>(mv[0], ma);
                    lds_rows4_wait<128>(mv[1], ma);
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int kl = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;       // key inside the tile
                            const bool vis = (k0 + kl <= qi) & (mv[kb][r >> 2][r & 3] != 0.f);
                            const float v = vis ? sacc[kb][r] : -INFINITY;
                            sacc[kb][r] = v;
                            t = fmaxf(t, v);
                        }
                } else {
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) t = fmaxf(t, sacc[kb][r]);
                }
                t = fmaxf(t, __shfl_xor(t, 32, 64));            // the row's maximum of (score - reference); -inf: no visible key in this tile
                // unseen row: the reference becomes the tile's maximum itself (if there is one); seen row: it only ever moves up
                const bool unseen = m == -INFINITY;
                const float d = (t == -INFINITY) ? 0.f : (unseen ? t : fmaxf(t, 0.f));
                const float alpha = unseen ? 1.f : fast_exp2(-d);   // (an unseen row's l and O are still zero)
                float ls[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        u4 w;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float e0 = fast_exp2(sacc[kb][8 * s2 + 2 * j] - d), e1 = fast_exp2(sacc[kb][8 * s2 + 2 * j + 1] - d);
                            ls[kb][s2] += e0; ls[kb][s2] += e1;
                            w[j] = pack_bf16(e0, e1);
                        }
                        pf[kb][s2] = __builtin_bit_cast(bf16x8, w);
                    }
                const float lsum = (ls[0][0] + ls[0][1]) + (ls[1][0] + ls[1][1]);
                l = l * alpha + lsum;
                if (t != -INFINITY) m = unseen ? t : m + d;
                if (__any(alpha != 1.f)) {
#pragma unroll
                    for (int db = 0; db < 2; ++db)
#pragma unroll
                        for (int r = 0; r < 16; ++r) accO[db][r] *= alpha;
                }
                if (__any(d != 0.f)) cinit = splat16(m == -INFINITY ? 0.f : -m);
            }
            APROF(2);
            LEAN_PRIO(1);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                if (kb == 0) tr_frags4_wait<0, 16 * 128>(vfr[0], vtr[0] + vimg, vtr[0] + vimg + 1024, vtr[1] + vimg, vtr[1] + vimg + 1024);
                else tr_frags4_wait<32 * 128, 48 * 128>(vfr[0], vtr[0] + vimg, vtr[0] + vimg + 1024, vtr[1] + vimg, vtr[1] + vimg + 1024);
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int db = 0; db < 2; ++db)
                        accO[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[0][s2][db], pf[kb][s2], accO[db], 0, 0, 0);
            }
            LEAN_PRIO(0);
            APROF(3);
        }
#if !(ECGB_LEAN_DIAG & 1)
        lean_wait_tiles<2 * PPW * (kAhead - 1)>();         // this wave's pieces of tile it + 1 have landed (the younger tiles' may still fly)
#endif
        APROF(4);
#if !(ECGB_LEAN_DIAG & 2)
        __builtin_amdgcn_s_barrier();                      // ... and everybody else's; all reads of tile it are done
#endif
        vimg = vimg == (kRing - 1) * 2 * kTile ? 0u : vimg + 2 * kTile;
        APROF(5);
#ifdef ECGB_PROFILE
        prof_acc[6] += 1;
#endif
    }
#ifdef ECGB_PROFILE
    if ((threadIdx.x & 63) == 0)
        for (int kk = 0; kk < 7; ++kk) atomicAdd(&g_attn_prof[(threadIdx.x >> 6) * 8 + kk], prof_acc[kk]);
#endif
    const float lt = l + __shfl_xor(l, 32, 64);
    const float inv = lt > 0.f ? 1.f / lt : 0.f;
    store_accT<2>(accO, A.o + (long long)hq * D, A.ldo, rowbase + qi, qvalid, h, inv);
    if (qvalid && h == 0) A.lse[((long long)b * A.Hq + hq) * A.S + qi] = lt > 0.f ? m + log2f(lt) : INFINITY;
}

// dQ, lean form: lanes = queries.  Register operands: -Q * scale * log2 e and -dO; initial accumulators: the row's lse and delta.
//   acc_s = lse - s,  p = exp2(-acc_s);   acc_dp = delta - dP;   dsn = p * acc_dp = -dS;   accQ += K^T . dsn  ->  dQ = -scale * accQ
template <int NW>
__global__ __launch_bounds__(NW * 64, ECGB_LEAN_DQ_WGS) void attn_bwd_dq_lean_kernel(AttnArgs A)
{
    constexpr int D = 64, kTile = 128 * D, PPW = 8 / NW;      // (waves, heads and rows of a workgroup: see attn_fwd_lean_kernel)
    constexpr int kRing = kLeanRing, kAhead = kRing - 1;   // (see attn_fwd_lean_kernel)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // kRing x (K, V) + 4 bytes per key
    float *lds_maskrow = reinterpret_cast<float *>(smem + 2 * kRing * kTile);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, lr = lane & 31, h = lane >> 5;
    const int hw_log2 = A.lean_hw_log2, R = (NW >> hw_log2) * 32;
    int qblk, head_in, group;
    map_block((int)blockIdx.x, (A.S + R - 1) / R, (A.Hq / A.Hkv) >> hw_log2, A.B * A.Hkv, true, qblk, head_in, group);
    const int b = group / A.Hkv, g = group % A.Hkv, hq = g * (A.Hq / A.Hkv) + (head_in << hw_log2) + (wave & ((1 << hw_log2) - 1));
    const int q0 = qblk * R, qw0 = q0 + (wave >> hw_log2) * 32;
    const int qi = qw0 + lr;
    const bool qvalid = qi < A.S;
    const long long rowbase = (long long)b * A.S;
    const unsigned short *Q = A.q + (long long)hq * D, *K = A.k + (long long)g * D, *V = A.v + (long long)g * D;
    const int k_end = min(A.S, q0 + R);
    const int wave_qmax = qw0 + 31;
    const int last_tile = (k_end - 1) / 64;
    unsigned offK[PPW], offV[PPW], offKt[PPW], offVt[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int r = (wave * PPW + i) * 8 + (lane >> 3), c = (lane & 7) ^ swz_u(r);
        const int rt = min(r, A.S - 1 - last_tile * 64);
        offK[i] = (unsigned)(((long long)r * A.ldk + c * 8) * 2);  offKt[i] = (unsigned)(((long long)rt * A.ldk + c * 8) * 2);
        offV[i] = (unsigned)(((long long)r * A.ldv + c * 8) * 2);  offVt[i] = (unsigned)(((long long)rt * A.ldv + c * 8) * 2);
    }
    // (the tile stream as running state: see attn_fwd_lean_kernel)
    const unsigned char *kb_next = reinterpret_cast<const unsigned char *>(K + rowbase * A.ldk), *vb_next = reinterpret_cast<const unsigned char *>(V + rowbase * A.ldv);
    const long long stepK = 128ll * A.ldk, stepV = 128ll * A.ldv;       // bytes from one 64-key tile to the next
    const bool tail_last = (last_tile + 1) * 64 > A.S;
    int t_next = 0;
    unsigned slot_next = 0;                                             // byte offset of the ring slot the next tile lands in
    auto issue_next = [&]() {
        const bool tail = tail_last && t_next == last_tile;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(kb_next + (tail ? offKt[i] : offK[i])),
                                             (__attribute__((address_space(3))) void *)(smem + slot_next + (wave * PPW + i) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(vb_next + (tail ? offVt[i] : offV[i])),
                                             (__attribute__((address_space(3))) void *)(smem + slot_next + kTile + (wave * PPW + i) * 1024), 16, 0, 0);
        }
        if (t_next < last_tile) { ++t_next; kb_next += stepK; vb_next += stepV; }
        slot_next = slot_next == (kRing - 1) * 2 * kTile ? 0u : slot_next + 2 * kTile;
    };
#pragma unroll
    for (int t = 0; t < kAhead; ++t) issue_next();
    bf16x8 qf[4], dof[4], of[4];
    load_row_frags<D>(qf, Q, A.ldq, rowbase + qi, qvalid, h);
    load_row_frags<D>(dof, A.d_o + (long long)hq * D, A.ldo, rowbase + qi, qvalid, h);
    load_row_frags<D>(of, A.o + (long long)hq * D, A.ldo, rowbase + qi, qvalid, h);
    float delta = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) delta += bf2f((unsigned short)dof[ks][j]) * bf2f((unsigned short)of[ks][j]);
    delta += __shfl_xor(delta, 32, 64);
    const long long stat = ((long long)b * A.Hq + hq) * A.S + qi;
    if (qvalid && h == 0) A.delta[stat] = delta;
    const float lse = qvalid ? A.lse[stat] : INFINITY;
    scale_row_frags(qf, -A.scale * kLog2e);
    negate_row_frags(dof);
    const f32x16 lseC = splat16(lse), delC = splat16(delta);
    f32x16 accQ[2] = {splat16(0.f), splat16(0.f)};
    lean_fill_mask<NW>(lds_maskrow, A.mask + rowbase, A.S, (k_end + 63) & ~63);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (the row loads above are younger than the tiles' pieces: one full wait, once)
    __syncthreads();
    const unsigned long long padbits = lean_pad_bits(lds_maskrow, (k_end + 63) & ~63);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
    unsigned trA[2], trB[2];
#pragma unroll
    for (int db = 0; db < 2; ++db) { trA[db] = lds0 + tr_off_u(db, lr, h, 0); trB[db] = lds0 + tr_off_u(db, lr, h, 8); }
    unsigned rbase[4];                                       // this lane's row fragment (row lr, k-step ks) inside an image; row 32 + lr is 4 KiB further
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) rbase[ks] = lds0 + lr * 128 + (((ks * 2 + h) ^ swz_u(lr)) << 4);
    unsigned kimg = 0;                                       // byte offset of the ring slot of the current tile
    for (int k0 = 0, it = 0; k0 < k_end; k0 += 64, ++it) {
        issue_next();
        if (k0 <= wave_qmax) {
            const float *lds_mask = lds_maskrow + k0;
            const bool need_mask = (k0 + 63 > qw0) || lean_tile_padded(padbits, it);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                bf16x8 kfr[4], vfr[4];                        // row fragments of this key half: K and V images side by side (+ kTile), two k-steps per batch
                if (kb == 0) lds_frags2x2_wait<0, kTile>(kfr[0], vfr[0], kfr[1], vfr[1], rbase[0] + kimg, rbase[1] + kimg);
                else lds_frags2x2_wait<4096, kTile + 4096>(kfr[0], vfr[0], kfr[1], vfr[1], rbase[0] + kimg, rbase[1] + kimg);
                f32x16 s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[0], qf[0], lseC, 0, 0, 0);
                f32x16 dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[0], dof[0], delC, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[1], qf[1], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[1], dof[1], dp, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (kb == 0) lds_frags2x2_wait<0, kTile>(kfr[2], vfr[2], kfr[3], vfr[3], rbase[2] + kimg, rbase[3] + kimg);
                else lds_frags2x2_wait<4096, kTile + 4096>(kfr[2], vfr[2], kfr[3], vfr[3], rbase[2] + kimg, rbase[3] + kimg);
#pragma unroll
                for (int ks = 2; ks < 4; ++ks) {
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[ks], qf[ks], s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[ks], dof[ks], dp, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                bf16x8 ktf[2][2];
                if (kb == 0) tr_frags4_wait<0, 16 * 128>(ktf, trA[0] + kimg, trB[0] + kimg, trA[1] + kimg, trB[1] + kimg);
                else tr_frags4_wait<32 * 128, 48 * 128>(ktf, trA[0] + kimg, trB[0] + kimg, trA[1] + kimg, trB[1] + kimg);
                float ds[16];
                if (need_mask) {
                    f4v mv[4];
                    const unsigned ma = (unsigned)(size_t)(__attribute__((address_space(3))) float *)const_cast<float *>(lds_mask) + 16 * h;
                    if (kb == 0) lds_rows4_wait<0>(mv, ma); else lds_rows4_wait<128>(mv, ma);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int kl = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        const bool vis = (k0 + kl <= qi) & (mv[r >> 2][r & 3] != 0.f);
                        ds[r] = vis ? fast_exp2(-s[r]) * dp[r] : 0.f;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) ds[r] = fast_exp2(-s[r]) * dp[r];
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const bf16x8 dsf = frag_from_acc(&ds[8 * s2]);
#pragma unroll
                    for (int db = 0; db < 2; ++db)
                        accQ[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf[s2][db], dsf, accQ[db], 0, 0, 0);
                }
            }
        }
        lean_wait_tiles<2 * PPW * (kAhead - 1)>();
        __builtin_amdgcn_s_barrier();
        kimg = kimg == (kRing - 1) * 2 * kTile ? 0u : kimg + 2 * kTile;
    }
    if (A.rope_cos) store_accT_rope_inv(accQ, A.dq + (long long)hq * D, A.lddq, rowbase + qi, qvalid, h, -A.scale, A.rope_cos + (rowbase + qi) * 32, A.rope_sin + (rowbase + qi) * 32);
    else store_accT<2>(accQ, A.dq + (long long)hq * D, A.lddq, rowbase + qi, qvalid, h, -A.scale);
}

// dK / dV, lean form: lanes = keys.  Register operands: -K * scale * log2 e and -V; initial accumulators: the tile's lse / delta rows as they lie in LDS
// (the lane's sixteen queries are four runs of four floats: four 16-byte reads ARE the sixteen accumulator registers).
//   acc_s = lse_q - s,  p = exp2(-acc_s);  acc_dp = delta_q - dP;  dsn = p * acc_dp = -dS;  accV += dO^T . p;  accK += Q^T . dsn  ->  dK = -scale * accK
// NW waves = NW x 32 keys per workgroup share every (Q, dO) tile (see attn_fwd_lean_kernel: the tile traffic, not the arithmetic, bounds the 4-wave kernel).
template <int NW>
__global__ __launch_bounds__(NW * 64, ECGB_LEAN_DKV_WGS) void attn_bwd_dkv_lean_kernel(AttnArgs A)
{
    constexpr int D = 64, kTile = 128 * D, PPW = 8 / NW, kStep = 2 * kTile + NW * 256;      // Q image, dO image, one 256-byte statistics row per wave
    constexpr int kRing = kLeanRing, kAhead = kRing - 1;   // (see attn_fwd_lean_kernel)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // kRing x kStep
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, lr = lane & 31, h = lane >> 5;
    // Under the causal mask key block j of a sequence has (n_blk - j) x as many query tiles as the last one: with one block per workgroup the 2 048 workgroups of the C3
    // shape carry 8 .. 64 steps each and the 512 slots of the chip were 58 % used (phase timers: 3 680 cycles per step and wave, 73 728 steps, 453 us).  A workgroup takes
    // the PAIR (j, n_blk - 1 - j): every workgroup the same work.
    const int n_blk = (A.S + NW * 32 - 1) / (NW * 32);
    int pair, head_in, group;
    map_block((int)blockIdx.x, (n_blk + 1) / 2, 1, A.B * A.Hkv, false, pair, head_in, group);
    const int b = group / A.Hkv, g = group % A.Hkv;
    const int G = A.Hq / A.Hkv;
    for (int half = 0; half < 2; ++half) {
    const int blk = half == 0 ? pair : n_blk - 1 - pair;
    if (half == 1) {
        if (blk == pair) break;                              // an odd number of blocks: the middle one once
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the clamped re-issues of the last step may still be landing in the ring
        __syncthreads();
    }
    const int kk0 = blk * (NW * 32);
    const int ki = kk0 + wave * 32 + lr;
    const bool kvalid = ki < A.S;
    const long long rowbase = (long long)b * A.S;
    const int wave_kmin = kk0 + wave * 32;
    const int t_begin = (kk0 / 64) * 64;
    const int tiles_per_head = (A.S - t_begin + 63) / 64;
    const int n_steps = G * tiles_per_head;
    const int tail_rows = A.S - (t_begin + (tiles_per_head - 1) * 64);
    unsigned offQ[PPW], offO[PPW], offQt[PPW], offOt[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int r = (wave * PPW + i) * 8 + (lane >> 3), c = (lane & 7) ^ swz_u(r);
        const int rt = min(r, tail_rows - 1);
        offQ[i] = (unsigned)(((long long)r * A.ldq + c * 8) * 2);  offQt[i] = (unsigned)(((long long)rt * A.ldq + c * 8) * 2);
        offO[i] = (unsigned)(((long long)r * A.ldo + c * 8) * 2);  offOt[i] = (unsigned)(((long long)rt * A.ldo + c * 8) * 2);
    }
    const unsigned offS = (unsigned)lane * 4u, offSt = (unsigned)min(lane, tail_rows - 1) * 4u;
    const float *stat_src = (wave & 1) ? A.delta : A.lse;
    auto issue_step = [&](int step, int buf) {
        const int hq = g * G + step / tiles_per_head;
        const int ti = step % tiles_per_head;
        const int t0 = t_begin + ti * 64;
        const bool tail = ti == tiles_per_head - 1 && tail_rows < 64;
        const unsigned char *qb = reinterpret_cast<const unsigned char *>(A.q + (long long)hq * D + (rowbase + t0) * A.ldq);
        const unsigned char *ob = reinterpret_cast<const unsigned char *>(A.d_o + (long long)hq * D + (rowbase + t0) * A.ldo);
        const unsigned char *sb = reinterpret_cast<const unsigned char *>(stat_src + ((long long)b * A.Hq + hq) * A.S + t0);
        unsigned char *dst = smem + buf * kStep;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(qb + (tail ? offQt[i] : offQ[i])),
                                             (__attribute__((address_space(3))) void *)(dst + (wave * PPW + i) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(ob + (tail ? offOt[i] : offO[i])),
                                             (__attribute__((address_space(3))) void *)(dst + kTile + (wave * PPW + i) * 1024), 16, 0, 0);
        }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(sb + (tail ? offSt : offS)),
                                         (__attribute__((address_space(3))) void *)(dst + 2 * kTile + wave * 256), 4, 0, 0);
    };
#pragma unroll
    for (int t = 0; t < kAhead; ++t) issue_step(min(t, n_steps - 1), t);
    bf16x8 kf[4], vf[4];
    load_row_frags<D>(kf, A.k + (long long)g * D, A.ldk, rowbase + ki, kvalid, h);
    load_row_frags<D>(vf, A.v + (long long)g * D, A.ldv, rowbase + ki, kvalid, h);
    const bool kvis = kvalid && A.mask[rowbase + (kvalid ? ki : 0)] != 0.f;
    scale_row_frags(kf, -A.scale * kLog2e);
    negate_row_frags(vf);
    const int vis_lo = kvis ? ki : A.S;
    const unsigned vis_n = (unsigned)(A.S - vis_lo);
    const bool wave_all_keys = __all(kvis);                  // (uniform) no padded / out-of-range key among this wave's 32
    f32x16 accK[2] = {splat16(0.f), splat16(0.f)}, accV[2] = {splat16(0.f), splat16(0.f)};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (the row loads above are younger than the steps' pieces: one full wait, once)
    __syncthreads();
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
    unsigned trA[2], trB[2];
#pragma unroll
    for (int db = 0; db < 2; ++db) { trA[db] = lds0 + tr_off_u(db, lr, h, 0); trB[db] = lds0 + tr_off_u(db, lr, h, 8); }
    const unsigned statA = lds0 + 2 * kTile + 16 * h;        // + 32 g4 + 128 qb: the lane's four runs of four queries
    unsigned rbase[4];                                       // this lane's row fragment (row lr, k-step ks) inside an image; row 32 + lr is 4 KiB further
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) rbase[ks] = lds0 + lr * 128 + (((ks * 2 + h) ^ swz_u(lr)) << 4);
#ifdef ECGB_PROFILE
    unsigned long long prof_acc[8] = {};
    long long t_prof = clock64();
#endif
    for (int step = 0; step < n_steps; ++step) {
        issue_step(min(step + kAhead, n_steps - 1), (step + kAhead) % kRing);
        APROF(0);
        const int buf = step % kRing;
        const unsigned img = (unsigned)(buf * kStep);
        const int t0 = t_begin + (step % tiles_per_head) * 64;
        if (t0 + 63 >= wave_kmin) {                            // else: every query of the tile precedes every key of this wave
            // (uniform) every (query, key) pair of this wave's 64 x 32 block is visible: queries all real and none before a key, keys all real
            const bool all_vis = wave_all_keys && t0 >= wave_kmin + 31 && t0 + 64 <= A.S;
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                f4v lse4[4], dl4[4];
                bf16x8 qfr[4], ofr[4];                        // row fragments of this query half: Q and dO images side by side (+ kTile), two k-steps per batch
                if (qb == 0) lds_stats_frags_wait<0, 0, kTile>(lse4, dl4, qfr[0], ofr[0], qfr[1], ofr[1], statA + img, rbase[0] + img, rbase[1] + img);
                else lds_stats_frags_wait<128, 4096, kTile + 4096>(lse4, dl4, qfr[0], ofr[0], qfr[1], ofr[1], statA + img, rbase[0] + img, rbase[1] + img);
                APROF(1);
                f32x16 c_s = __builtin_shufflevector(__builtin_shufflevector(lse4[0], lse4[1], 0, 1, 2, 3, 4, 5, 6, 7),
                                                     __builtin_shufflevector(lse4[2], lse4[3], 0, 1, 2, 3, 4, 5, 6, 7),
                                                     0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
                const f32x16 c_dp = __builtin_shufflevector(__builtin_shufflevector(dl4[0], dl4[1], 0, 1, 2, 3, 4, 5, 6, 7),
                                                            __builtin_shufflevector(dl4[2], dl4[3], 0, 1, 2, 3, 4, 5, 6, 7),
                                                            0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
                if (!all_vis) {                               // an invisible pair starts from +inf: p = exp2(-inf) = 0 exactly
                    const int qrel = t0 + 4 * h - vis_lo;     // (query of element r) - vis_lo = qrel + 32 qb + 8 (r >> 2) + (r & 3)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        c_s[r] = ((unsigned)(qrel + (qb * 32 + 8 * (r >> 2) + (r & 3))) < vis_n) ? c_s[r] : INFINITY;
                }
                f32x16 s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qfr[0], kf[0], c_s, 0, 0, 0);
                f32x16 dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ofr[0], vf[0], c_dp, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qfr[1], kf[1], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ofr[1], vf[1], dp, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (qb == 0) lds_frags2x2_wait<0, kTile>(qfr[2], ofr[2], qfr[3], ofr[3], rbase[2] + img, rbase[3] + img);
                else lds_frags2x2_wait<4096, kTile + 4096>(qfr[2], ofr[2], qfr[3], ofr[3], rbase[2] + img, rbase[3] + img);
#pragma unroll
                for (int ks = 2; ks < 4; ++ks) {
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qfr[ks], kf[ks], s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ofr[ks], vf[ks], dp, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                float pr[16], ds[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float e = fast_exp2(-s[r]);
                    pr[r] = e;
                    ds[r] = e * dp[r];
                }
                const bf16x8 pf0 = frag_from_acc(&pr[0]), pf1 = frag_from_acc(&pr[8]), dsf0 = frag_from_acc(&ds[0]), dsf1 = frag_from_acc(&ds[8]);
                APROF(2);
                {                                             // the transposed fragments one image at a time (dO, then Q): sixteen registers live instead of thirty-two
                    bf16x8 tf[2][2];                          // [s2][db]
                    if (qb == 0) tr_frags4_wait<kTile, kTile + 16 * 128>(tf, trA[0] + img, trB[0] + img, trA[1] + img, trB[1] + img);
                    else tr_frags4_wait<kTile + 32 * 128, kTile + 48 * 128>(tf, trA[0] + img, trB[0] + img, trA[1] + img, trB[1] + img);
#pragma unroll
                    for (int db = 0; db < 2; ++db) {
                        accV[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[0][db], pf0, accV[db], 0, 0, 0);
                        accV[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[1][db], pf1, accV[db], 0, 0, 0);
                    }
                    if (qb == 0) tr_frags4_wait<0, 16 * 128>(tf, trA[0] + img, trB[0] + img, trA[1] + img, trB[1] + img);
                    else tr_frags4_wait<32 * 128, 48 * 128>(tf, trA[0] + img, trB[0] + img, trA[1] + img, trB[1] + img);
#pragma unroll
                    for (int db = 0; db < 2; ++db) {
                        accK[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[0][db], dsf0, accK[db], 0, 0, 0);
                        accK[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[1][db], dsf1, accK[db], 0, 0, 0);
                    }
                }
                APROF(3);
            }
        }
        lean_wait_tiles<(2 * PPW + 1) * (kAhead - 1)>();
        APROF(4);
        __builtin_amdgcn_s_barrier();
        APROF(5);
#ifdef ECGB_PROFILE
        prof_acc[6] += 1;
#endif
    }
#ifdef ECGB_PROFILE
    if ((threadIdx.x & 63) == 0)
        for (int kk = 0; kk < 7; ++kk) atomicAdd(&g_attn_prof[(threadIdx.x >> 6) * 8 + kk], prof_acc[kk]);
#endif
    if (A.rope_cos) store_accT_rope_inv(accK, A.dk + (long long)g * D, A.lddk, rowbase + ki, kvalid, h, -A.scale, A.rope_cos + (rowbase + ki) * 32, A.rope_sin + (rowbase + ki) * 32);
    else store_accT<2>(accK, A.dk + (long long)g * D, A.lddk, rowbase + ki, kvalid, h, -A.scale);
    store_accT<2>(accV, A.dv + (long long)g * D, A.lddv, rowbase + ki, kvalid, h, 1.f);
    }
}

template <int D, int DS, int WHICH = 0>
__global__ __launch_bounds__(256, D == 64 ? 2 : 1) void attn_bwd_dkv_kernel(AttnArgs A)   // head_dim 64: two waves per SIMD (<= 256 registers)
{
    int blk, head_in, group;
    map_block((int)blockIdx.x, ((A.S + 127) / 128) * DS, 1, A.B * A.Hkv, false, blk, head_in, group);   // key block 0 sees every query: first
    attn_bwd_dkv_body<D, DS, WHICH>(A, blk, group / A.Hkv, group % A.Hkv);
}
// head_dim 256: the dV pass and the dK pass of a key block as two workgroups of ONE launch (even / odd blockIdx.x) -- Gemma's single KV
// head gives only S / 128 x batch key blocks, half a chip's worth at C5's shape
template <int D>
__global__ __launch_bounds__(256, 1) void attn_bwd_dkv_pair_kernel(AttnArgs A)
{
    // blocks of a (batch, KV head) group: (key block, head split, pass) with the key block slowest -- heaviest first
    const int hs = A.head_splits > 1 ? A.head_splits : 1;
    int blk, head_in, group;
    map_block((int)blockIdx.x, ((A.S + 127) / 128) * 2 * hs, 1, A.B * A.Hkv, false, blk, head_in, group);
    const int pass = blk & 1, hsplit = (blk >> 1) % hs, kblk = (blk >> 1) / hs;
    if (pass) attn_bwd_dkv_body<D, 1, 2>(A, kblk, group / A.Hkv, group % A.Hkv, hsplit);
    else attn_bwd_dkv_body<D, 1, 1>(A, kblk, group / A.Hkv, group % A.Hkv, hsplit);
}

// dk / dv [row][g * D + d] (bf16, row strides lddk / lddv) = sum over the head splits of the fp32 slabs, in split order
template <int D>
__global__ __launch_bounds__(256) void attn_dkv_reduce_kernel(AttnArgs A)
{
    const long long rows_all = (long long)A.B * A.S, slab = rows_all * A.Hkv * D;
    const long long n4 = slab / 4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * n4; i += (long long)gridDim.x * blockDim.x) {
        const int which = i >= n4;                           // 0 = dK, 1 = dV
        const long long e = (i - which * n4) * 4;            // element inside a slab: [g][row][d]
        const long long g = e / (rows_all * D), row = (e / D) % rows_all;
        const int d = (int)(e % D);
        using f4 = __attribute__((ext_vector_type(4))) float;
        f4 v = *reinterpret_cast<const f4 *>(A.slab + (long long)which * A.head_splits * slab + e);
        for (int hsp = 1; hsp < A.head_splits; ++hsp) v += *reinterpret_cast<const f4 *>(A.slab + ((long long)which * A.head_splits + hsp) * slab + e);
        using u2 = __attribute__((ext_vector_type(2))) unsigned;
        u2 o;
        o[0] = pack_bf16(v[0], v[1]); o[1] = pack_bf16(v[2], v[3]);
        unsigned short *dst = which ? A.dv + row * A.lddv : A.dk + row * A.lddk;
        *reinterpret_cast<u2 *>(dst + g * D + d) = o;
    }
}

// =====================================================================================================
// decode step: one query per (batch, head) against the KV cache.  grid (Hq, B), 16 waves per workgroup; head_dim 64 / 128 / 256.
// q [B, Hq*D]; k/v cache rows [B, cap] with row stride ld (elements), kv head g at + g*D; len = keys in the cache
// (incl. the new one); key j visible iff mask[b, j] != 0.  Reads the cache once, every row as one coalesced D*2-byte piece:
//   scores  a wave takes keys w, w+16, ...: its lanes read a key row together (D/64 elements each), multiply with the
//           query (registers) and add up across the wave; eight keys in flight per wave;
//   softmax max and sum over the workgroup, probabilities in LDS (fp32);
//   P.V     thread t owns one 16-byte piece of the value rows of its key slice (interleaved keys); the slices' partial
//           sums meet in LDS in a fixed order.  P is cast to bf16 before P.V, as SDPA does.
constexpr int kDecodeThreads = 1024;
template <int D>
__global__ __launch_bounds__(kDecodeThreads) void attn_decode_kernel(const unsigned short *q, const unsigned short *kc, const unsigned short *vc,
                                                                     long long ld, long long cap, const float *mask, long long mask_ld,
                                                                     unsigned short *o, int len_arg, const int *len_dev, int Hq, int Hkv, float scale)
{
    constexpr int NW = kDecodeThreads / 64, EPL = D / 64;   // waves, elements per lane of a key row
    constexpr int KU = 16, VU = 8;                      // key rows a wave / value rows a thread keeps in flight (memory latency, not bandwidth, is the cost)
    extern __shared__ float s_p[];                      // [len] scores / probabilities, later [kDecodeThreads * 8] partial outputs; [2 * NW] reductions
    const int len = len_dev ? *len_dev : len_arg;       // device-resident when the step is replayed from a captured graph
    const int rows_lds = len_dev ? (((int)cap + 3) & ~3) : ((len + 3) & ~3);
    float *s_red = s_p + max(rows_lds, kDecodeThreads * 8);   // behind the scores / the P.V partial sums, whichever is longer
    const int hq = blockIdx.x, b = blockIdx.y, g = hq / (Hq / Hkv), tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned short *qp = q + ((long long)b * Hq + hq) * D;
    float qf[EPL];
#pragma unroll
    for (int t = 0; t < EPL; ++t) qf[t] = bf2f(qp[lane * EPL + t]);
    const unsigned short *K = kc + (long long)b * cap * ld + (long long)g * D;
    const unsigned short *V = vc + (long long)b * cap * ld + (long long)g * D;
    const float *mrow = mask + (long long)b * mask_ld;

    // ---- scores
    float m = -INFINITY;
    for (int j0 = wave; j0 < len; j0 += KU * NW) {
        float part[KU], mk[KU];
#pragma unroll
        for (int u = 0; u < KU; ++u) mk[u] = mrow[min(j0 + u * NW, len - 1)];   // in flight together with the key rows
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const int j = j0 + u * NW;
            part[u] = 0.f;
            if (j < len) {
                const unsigned short *kr = K + (long long)j * ld + lane * EPL;
                if constexpr (EPL == 4) {
                    const uint2 kv = *reinterpret_cast<const uint2 *>(kr);
                    part[u] = qf[0] * __uint_as_float(kv.x << 16) + qf[1] * __uint_as_float(kv.x & 0xFFFF0000u) +
                              qf[2] * __uint_as_float(kv.y << 16) + qf[3] * __uint_as_float(kv.y & 0xFFFF0000u);
                } else if constexpr (EPL == 2) {
                    const unsigned kv = *reinterpret_cast<const unsigned *>(kr);
                    part[u] = qf[0] * __uint_as_float(kv << 16) + qf[1] * __uint_as_float(kv & 0xFFFF0000u);
                } else {
                    part[u] = qf[0] * bf2f(kr[0]);
                }
            }
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1)
#pragma unroll
            for (int u = 0; u < KU; ++u) part[u] += __shfl_xor(part[u], d, 64);
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const int j = j0 + u * NW;
            if (j < len) {
                const float sdot = (mk[u] != 0.f) ? part[u] * scale : -INFINITY;
                if (lane == 0) s_p[j] = sdot;
                m = fmaxf(m, sdot);
            }
        }
    }
    if (lane == 0) s_red[wave] = m;
    __syncthreads();
    m = s_red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) m = fmaxf(m, s_red[w]);

    // ---- softmax
    float l = 0.f;
    for (int j = tid; j < len; j += kDecodeThreads) {
        const float e = (m == -INFINITY) ? 0.f : __expf(s_p[j] - m);
        s_p[j] = e;
        l += e;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) l += __shfl_xor(l, d, 64);
    if (lane == 0) s_red[NW + wave] = l;
    __syncthreads();
    l = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) l += s_red[NW + w];
    const float inv = l > 0.f ? 1.f / l : 0.f;

    // ---- P.V: thread t owns 8 consecutive output dimensions (one 16-byte piece of a value row) and the keys of slice
    // t / (D / 8), interleaved; all of a thread's rows of a batch are in flight together
    constexpr int TPR = D / 8, NS = kDecodeThreads / TPR;      // threads per value row, key slices
    const int piece = tid % TPR, slice = tid / TPR;
    float acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = 0.f;
    for (int j0 = slice; j0 < len; j0 += VU * NS) {
        bf16x8 vv[VU];
#pragma unroll
        for (int u = 0; u < VU; ++u) {
            const int j = min(j0 + u * NS, len - 1);                    // past the end: a repeated row with probability 0
            vv[u] = *reinterpret_cast<const bf16x8 *>(V + (long long)j * ld + piece * 8);
        }
#pragma unroll
        for (int u = 0; u < VU; ++u) {
            const int j = j0 + u * NS;
            const float pj = (j < len) ? bf2f((unsigned short)(pack_bf16(s_p[j] * inv, 0.f) & 0xFFFFu)) : 0.f;
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[t] += pj * bf2f((unsigned short)vv[u][t]);
        }
    }
    __syncthreads();                                            // s_p is dead: the slices' partial sums go there
    float *s_part = s_p;                                        // [NS][D] floats (the launch sizes the buffer for it)
#pragma unroll
    for (int t = 0; t < 8; ++t) s_part[slice * D + piece * 8 + t] = acc[t];
    __syncthreads();
    if (tid < D) {
        float sum = 0.f;
#pragma unroll 8
        for (int sl = 0; sl < NS; ++sl) sum += s_part[sl * D + tid];
        o[((long long)b * Hq + hq) * D + tid] = (unsigned short)(pack_bf16(sum, 0.f) & 0xFFFFu);
    }
}

// ---- the same decode step with the keys of a head split over several workgroups (long caches, few heads: one CU
// sustains only ~64 GB/s at the memory latency, the chip has 256).  Three small kernels:
//   scores   grid (splits, Hq, B): the scores of the split's keys to scratch (fp32) + the split's max and sum of exponentials;
//   values   grid (splits, Hq, B): global max / sum from the splits' statistics, p_j = bf16(exp(s_j - max) / sum) exactly as
//            attn_decode_kernel forms it, partial P.V of the split's keys to scratch (fp32);
//   combine  grid (Hq, B): the partial outputs added in split order, cast to bf16.
// scratch (floats): scores [B, Hq, cap] | stats [B, Hq, splits, 2] | partial outputs [B, Hq, splits, D].
constexpr int kSplitThreads = 256;
constexpr int kSplitMaxChunk = 2048;                     // keys per split the scores kernel can hold (the launcher raises n_splits to fit)
template <int D>
__global__ __launch_bounds__(kSplitThreads) void attn_decode_scores_kernel(const unsigned short *q, const unsigned short *kc, long long ld, long long cap,
                                                                           const float *mask, long long mask_ld, float *scores, float *stats,
                                                                           int len_arg, int chunk_arg, const int *len_dev, int Hq, int Hkv, float scale)
{
    constexpr int NW = kSplitThreads / 64, EPL = D / 64, KU = 16;
    __shared__ float s_red[2 * NW];
    __shared__ float s_sc[kSplitMaxChunk];                // the split's scores once more: the sum of exponentials reads them here
    const int sp = blockIdx.x, hq = blockIdx.y, b = blockIdx.z, g = hq / (Hq / Hkv), tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // device-resident length (a replayed graph): the same split of the keys the host computes from a length it knows
    const int len = len_dev ? *len_dev : len_arg, chunk = len_dev ? (len + (int)gridDim.x - 1) / (int)gridDim.x : chunk_arg;
    const int k0 = sp * chunk, k1 = min(len, k0 + chunk);
    const unsigned short *qp = q + ((long long)b * Hq + hq) * D;
    float qf[EPL];
#pragma unroll
    for (int t = 0; t < EPL; ++t) qf[t] = bf2f(qp[lane * EPL + t]);
    const unsigned short *K = kc + (long long)b * cap * ld + (long long)g * D;
    const float *mrow = mask + (long long)b * mask_ld;
    float *srow = scores + ((long long)b * Hq + hq) * cap;
    float m = -INFINITY;
    for (int j0 = k0 + wave; j0 < k1; j0 += KU * NW) {
        float part[KU], mk[KU];
#pragma unroll
        for (int u = 0; u < KU; ++u) mk[u] = mrow[min(j0 + u * NW, k1 - 1)];
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const int j = j0 + u * NW;
            part[u] = 0.f;
            if (j < k1) {
                const unsigned short *kr = K + (long long)j * ld + lane * EPL;
                if constexpr (EPL == 4) {
                    const uint2 kv = *reinterpret_cast<const uint2 *>(kr);
                    part[u] = qf[0] * __uint_as_float(kv.x << 16) + qf[1] * __uint_as_float(kv.x & 0xFFFF0000u) +
                              qf[2] * __uint_as_float(kv.y << 16) + qf[3] * __uint_as_float(kv.y & 0xFFFF0000u);
                } else if constexpr (EPL == 2) {
                    const unsigned kv = *reinterpret_cast<const unsigned *>(kr);
                    part[u] = qf[0] * __uint_as_float(kv << 16) + qf[1] * __uint_as_float(kv & 0xFFFF0000u);
                } else {
                    part[u] = qf[0] * bf2f(kr[0]);
                }
            }
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1)
#pragma unroll
            for (int u = 0; u < KU; ++u) part[u] += __shfl_xor(part[u], d, 64);
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const int j = j0 + u * NW;
            if (j < k1) {
                const float sdot = (mk[u] != 0.f) ? part[u] * scale : -INFINITY;
                if (lane == 0) { srow[j] = sdot; s_sc[j - k0] = sdot; }
                m = fmaxf(m, sdot);
            }
        }
    }
    if (lane == 0) s_red[wave] = m;
    __syncthreads();
    m = s_red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) m = fmaxf(m, s_red[w]);
    float l = 0.f;
    for (int j = k0 + tid; j < k1; j += kSplitThreads) l += (m == -INFINITY) ? 0.f : __expf(s_sc[j - k0] - m);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) l += __shfl_xor(l, d, 64);
    if (lane == 0) s_red[NW + wave] = l;
    __syncthreads();
    if (tid == 0) {
        l = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) l += s_red[NW + w];
        float *st = stats + (((long long)b * Hq + hq) * gridDim.x + sp) * 2;
        st[0] = m;
        st[1] = l;
    }
}

template <int D>
__global__ __launch_bounds__(kSplitThreads) void attn_decode_values_kernel(const unsigned short *vc, long long ld, long long cap, const float *scores,
                                                                           const float *stats, float *partial, int len_arg, int chunk_arg, const int *len_dev,
                                                                           int Hq, int Hkv)
{
    constexpr int TPR = D / 8, NS = kSplitThreads / TPR, VU = 8;
    __shared__ float s_part[NS * D];
    const int sp = blockIdx.x, hq = blockIdx.y, b = blockIdx.z, g = hq / (Hq / Hkv), tid = threadIdx.x;
    const int n_splits = gridDim.x;
    const int len = len_dev ? *len_dev : len_arg, chunk = len_dev ? (len + n_splits - 1) / n_splits : chunk_arg;
    const int k0 = sp * chunk, k1 = min(len, k0 + chunk);
    const float *st = stats + ((long long)b * Hq + hq) * n_splits * 2;
    float m = -INFINITY;
    for (int t = 0; t < n_splits; ++t) m = fmaxf(m, st[2 * t]);
    float l = 0.f;
    for (int t = 0; t < n_splits; ++t) l += (st[2 * t] == -INFINITY) ? 0.f : st[2 * t + 1] * __expf(st[2 * t] - m);
    const float inv = l > 0.f ? 1.f / l : 0.f;
    const unsigned short *V = vc + (long long)b * cap * ld + (long long)g * D;
    const float *srow = scores + ((long long)b * Hq + hq) * cap;
    const int piece = tid % TPR, slice = tid / TPR;
    float acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = 0.f;
    for (int j0 = k0 + slice; j0 < k1; j0 += VU * NS) {
        bf16x8 vv[VU];
        float sc[VU];
#pragma unroll
        for (int u = 0; u < VU; ++u) {
            const int j = min(j0 + u * NS, k1 - 1);
            vv[u] = *reinterpret_cast<const bf16x8 *>(V + (long long)j * ld + piece * 8);
            sc[u] = srow[j];
        }
#pragma unroll
        for (int u = 0; u < VU; ++u) {
            const int j = j0 + u * NS;
            const float e = (m == -INFINITY) ? 0.f : __expf(sc[u] - m);
            const float pj = (j < k1) ? bf2f((unsigned short)(pack_bf16(e * inv, 0.f) & 0xFFFFu)) : 0.f;
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[t] += pj * bf2f((unsigned short)vv[u][t]);
        }
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) s_part[slice * D + piece * 8 + t] = acc[t];
    __syncthreads();
    if (tid < D) {
        float sum = 0.f;
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) sum += s_part[sl * D + tid];
        partial[(((long long)b * Hq + hq) * n_splits + sp) * D + tid] = sum;
    }
}

template <int D>
__global__ __launch_bounds__(D) void attn_decode_combine_kernel(const float *partial, unsigned short *o, int n_splits, int Hq)
{
    const int hq = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const float *p = partial + ((long long)b * Hq + hq) * n_splits * D + tid;
    float sum = 0.f;
    for (int t = 0; t < n_splits; ++t) sum += p[(long long)t * D];
    o[((long long)b * Hq + hq) * D + tid] = (unsigned short)(pack_bf16(sum, 0.f) & 0xFFFFu);
}

// ---- round 6: the four launches of a decode step's attention (rope_append, scores, values, combine: 4.8 + 11.2 + 5.3 + 4.8 us a layer in the replayed graph at the
// C5 shape, every one of them on the ~5 us floor of a dependent launch) as ONE.  grid (splits, Hq, B) as the split kernels; the workgroups of a (sequence, head) meet
// twice through counters in device memory:
//   0  RoPE of the head's q and of the KV head's new key in LDS (rope_append_kernel's arithmetic; the workgroup of split 0 of the group's first head writes the rotated
//      key and the value to cache row len - 1 -- nobody reads that row in this launch: the split that owns key len - 1 takes it from LDS / from qkv)
//   1  the split's scores into LDS, its maximum and sum of exponentials to `stats` (attn_decode_scores_kernel's sums);  wait until every split's pair is there (a pair holds kDecodeOneEmpty until stored)
//   2  softmax over the splits' statistics, P.V of the split's keys (attn_decode_values_kernel's sums) to `partial`;  counter B: the last to arrive adds the
//      partial outputs in split order (attn_decode_combine_kernel), puts the empty marks back and clears the counter for the next launch (a replayed graph never does itself).
// The same arithmetic in the same order: the same bits as the four launches (tests/test_gpu_decode_fused.py).  The wait in step 1 needs every workgroup of the
// launch resident at once: the host only takes this form for splits * Hq * B <= two workgroups a CU.
// "not stored yet" in the splits' statistics: a NaN with a payload no arithmetic produces (a sum or maximum that IS NaN -- garbage in -- is the canonical one and ends the wait like any value)
constexpr unsigned kDecodeOneEmpty = 0x7FC0DEADu;
struct DecodeOneArgs {
    const unsigned short *qkv;    // [B, (Hq + 2 Hkv) D]: the step's projection, not written (q and the new key are rotated into LDS)
    long long ld_qkv;
    const float *cs, *sn;         // RoPE tables of the step's positions [B, D / 2]
    unsigned short *cache;        // [B, cap, ld]: keys | values
    long long cap, ld;
    const float *mask;
    long long mask_ld;
    float *stats, *partial;       // [B, Hq, splits, 2] (kDecodeOneEmpty between launches), [B, Hq, splits, D]
    unsigned *cnt;                // [B, Hq, 2], zero between launches (word 1: the tickets of the second meeting)
    unsigned short *o;            // [B, Hq D]
    const int *len_dev;
    int len_arg, Hq, Hkv;
    float scale;
};

template <int D>
__global__ __launch_bounds__(kSplitThreads) void attn_decode_one_kernel(DecodeOneArgs A)
{
    constexpr int NW = kSplitThreads / 64, EPL = D / 64, KU = 16, half = D / 2;
    constexpr int TPR = D / 8, NS = kSplitThreads / TPR, VU = 8;
    __shared__ __attribute__((aligned(16))) unsigned short s_q[D], s_kn[D];
    __shared__ float s_red[2 * NW];
    __shared__ float s_sc[kSplitMaxChunk];
    __shared__ float s_part[NS * D];
    __shared__ float s_st[2 * 64];                         // the splits' (maximum, sum) pairs (splits <= 64)
    __shared__ unsigned s_last;
    const int sp = blockIdx.x, hq = blockIdx.y, b = blockIdx.z, G = A.Hq / A.Hkv, g = hq / G, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_splits = gridDim.x;
    const int len = A.len_dev ? *A.len_dev : A.len_arg, chunk = (len + n_splits - 1) / n_splits;
    const int k0 = sp * chunk, k1 = min(len, k0 + chunk);
    const unsigned short *row = A.qkv + (long long)b * A.ld_qkv;
    auto bf = [](float x) { return (unsigned short)(pack_bf16(x, 0.f) & 0xFFFFu); };
    // Memory between the workgroups of a (sequence, head): everything they exchange (statistics, partial outputs, counters) is written and read with agent-scope
    // RELAXED atomics -- stores that go through to the level the XCDs share, loads that do not stop at this XCD's L2 -- ordered by hand: a wave's stores are
    // complete (s_waitcnt vmcnt(0)) before its workgroup's counter moves.  Release / acquire orderings on the counters cost a write-back of the whole L2 per
    // arrival and an invalidate per poll: the launch took 24 us that way, as long as the four it replaces.
    auto ld_f = [](const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto st_f = [](float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    const unsigned short *K = A.cache + (long long)b * A.cap * A.ld + (long long)g * D;
    const unsigned short *V = A.cache + (long long)b * A.cap * A.ld + (long long)A.Hkv * D + (long long)g * D;
    const unsigned short *vnew = row + (long long)(A.Hq + A.Hkv + g) * D;
    const float *mrow = A.mask + (long long)b * A.mask_ld;
    const int piece = tid % TPR, slice = tid / TPR;
    // ---- the first batch of keys and values is asked for before anything else (a split of the C5 step is one batch: 61 keys): the loads fly under the RoPE and
    // the values' under the scores and the first meeting
    using kraw_t = uint2;                                                      // EPL elements of a key row as raw bits (EPL 4: 8 bytes, 2: 4, 1: 2)
    kraw_t kraw[KU];
    float mk0[KU];
    auto load_k = [&](int j0, kraw_t (&kr)[KU], float (&mk)[KU]) {
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const int j = min(j0 + u * NW, k1 - 1);
            mk[u] = mrow[j];
            const unsigned short *p = K + (long long)j * A.ld + lane * EPL;
            if constexpr (EPL == 4) kr[u] = *reinterpret_cast<const uint2 *>(p);
            else if constexpr (EPL == 2) { kr[u].x = *reinterpret_cast<const unsigned *>(p); kr[u].y = 0u; }
            else { kr[u].x = p[0]; kr[u].y = 0u; }
        }
    };
    bf16x8 vraw[VU];
    auto load_v = [&](int j0, bf16x8 (&vv)[VU]) {
#pragma unroll
        for (int u = 0; u < VU; ++u) {
            const int j = min(j0 + u * NS, k1 - 1);
            vv[u] = *reinterpret_cast<const bf16x8 *>((j == len - 1 ? vnew : V + (long long)j * A.ld) + piece * 8);
        }
    };
    if (k0 + wave < k1) load_k(k0 + wave, kraw, mk0);
    if (k0 + slice < k1) load_v(k0 + slice, vraw);
    // ---- 0: RoPE -- piece c (eight elements) of a head's first half with piece c of its second half
    if (tid < 2 * (half / 8)) {
        const int c = tid % (half / 8), h = tid / (half / 8);                  // h 0: the query head; 1: the new key of its KV head
        const unsigned short *p = row + (long long)(h == 0 ? hq : A.Hq + g) * D + c * 8;
        bf16x8 a = *reinterpret_cast<const bf16x8 *>(p), bb = *reinterpret_cast<const bf16x8 *>(p + half);
        const float *pc = A.cs + (size_t)b * half + c * 8, *ps = A.sn + (size_t)b * half + c * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float cc = bf2f(bf(pc[j])), ss = bf2f(bf(ps[j]));
            const float x1 = bf2f((unsigned short)a[j]), x2 = bf2f((unsigned short)bb[j]);
            a[j] = (short)bf(x1 * cc - x2 * ss); bb[j] = (short)bf(x2 * cc + x1 * ss);
        }
        unsigned short *dst = h == 0 ? s_q : s_kn;
        *reinterpret_cast<bf16x8 *>(dst + c * 8) = a;
        *reinterpret_cast<bf16x8 *>(dst + half + c * 8) = bb;
        if (h == 1 && sp == 0 && hq == g * G) {                                // the append: rotated key
            unsigned short *kc = A.cache + ((long long)b * A.cap + (len - 1)) * A.ld + (size_t)g * D + c * 8;
            *reinterpret_cast<bf16x8 *>(kc) = a;
            *reinterpret_cast<bf16x8 *>(kc + half) = bb;
        }
    }
    if (sp == 0 && hq == g * G && tid >= 128 && tid < 128 + D / 8)             // the append: the value as it is
        *reinterpret_cast<bf16x8 *>(A.cache + ((long long)b * A.cap + (len - 1)) * A.ld + (size_t)A.Hkv * D + (size_t)g * D + (tid - 128) * 8) =
            *reinterpret_cast<const bf16x8 *>(vnew + (tid - 128) * 8);
    __syncthreads();
    // ---- 1: scores (attn_decode_scores_kernel)
    float qf[EPL], kn[EPL];
#pragma unroll
    for (int t = 0; t < EPL; ++t) { qf[t] = bf2f(s_q[lane * EPL + t]); kn[t] = bf2f(s_kn[lane * EPL + t]); }
    float m = -INFINITY;
    for (int j0 = k0 + wave; j0 < k1; j0 += KU * NW) {
        if (j0 != k0 + wave) load_k(j0, kraw, mk0);
        float part[KU];
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const int j = j0 + u * NW;
            part[u] = 0.f;
            if (j < k1) {
                if (j == len - 1) {                                            // the new key: its cache row is being written by another workgroup
                    part[u] = qf[0] * kn[0];
#pragma unroll
                    for (int t = 1; t < EPL; ++t) part[u] += qf[t] * kn[t];
                } else if constexpr (EPL == 4) {
                    part[u] = qf[0] * __uint_as_float(kraw[u].x << 16) + qf[1] * __uint_as_float(kraw[u].x & 0xFFFF0000u) +
                              qf[2] * __uint_as_float(kraw[u].y << 16) + qf[3] * __uint_as_float(kraw[u].y & 0xFFFF0000u);
                } else if constexpr (EPL == 2) {
                    part[u] = qf[0] * __uint_as_float(kraw[u].x << 16) + qf[1] * __uint_as_float(kraw[u].x & 0xFFFF0000u);
                } else {
                    part[u] = qf[0] * bf2f((unsigned short)kraw[u].x);
                }
            }
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1)
#pragma unroll
            for (int u = 0; u < KU; ++u) part[u] += __shfl_xor(part[u], d, 64);
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const int j = j0 + u * NW;
            if (j < k1) {
                const float sdot = (mk0[u] != 0.f) ? part[u] * A.scale : -INFINITY;
                if (lane == 0) s_sc[j - k0] = sdot;
                m = fmaxf(m, sdot);
            }
        }
    }
    if (lane == 0) s_red[wave] = m;
    __syncthreads();
    m = s_red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) m = fmaxf(m, s_red[w]);
    float l = 0.f;
    for (int j = k0 + tid; j < k1; j += kSplitThreads) l += (m == -INFINITY) ? 0.f : __expf(s_sc[j - k0] - m);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) l += __shfl_xor(l, d, 64);
    if (lane == 0) s_red[NW + wave] = l;
    __syncthreads();
    float *st = A.stats + ((long long)b * A.Hq + hq) * n_splits * 2;
    unsigned *cnt = A.cnt + ((long long)b * A.Hq + hq) * 2;
    if (tid == 0) {
        l = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) l += s_red[NW + w];
        st_f(&st[2 * sp], m);
        st_f(&st[2 * sp + 1], l);
    }
    // ---- the first meeting: no counter.  A split's (maximum, sum) pair is kDecodeOneEmpty -- a NaN no sum produces -- until its workgroup has stored it (the scratch starts so and the last workgroup of
    // the launch puts it back): lane t polls pair t itself -- one round trip from the store to the values, where a counter cost an atomic, a poll and a load
    if (tid < 2 * n_splits) {
        const unsigned *sb = reinterpret_cast<const unsigned *>(st) + tid;
        unsigned v = __hip_atomic_load(sb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (v == kDecodeOneEmpty) { __builtin_amdgcn_s_sleep(1); v = __hip_atomic_load(sb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        s_st[tid] = __uint_as_float(v);
    }
    __syncthreads();
    // ---- 2: values (attn_decode_values_kernel)
    m = -INFINITY;
    for (int t = 0; t < n_splits; ++t) m = fmaxf(m, s_st[2 * t]);
    l = 0.f;
    for (int t = 0; t < n_splits; ++t) l += (s_st[2 * t] == -INFINITY) ? 0.f : s_st[2 * t + 1] * __expf(s_st[2 * t] - m);
    const float inv = l > 0.f ? 1.f / l : 0.f;
    float acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = 0.f;
    for (int j0 = k0 + slice; j0 < k1; j0 += VU * NS) {
        if (j0 != k0 + slice) load_v(j0, vraw);
        float sc[VU];
#pragma unroll
        for (int u = 0; u < VU; ++u) sc[u] = s_sc[min(j0 + u * NS, k1 - 1) - k0];
#pragma unroll
        for (int u = 0; u < VU; ++u) {
            const int j = j0 + u * NS;
            const float e = (m == -INFINITY) ? 0.f : __expf(sc[u] - m);
            const float pj = (j < k1) ? bf2f(bf(e * inv)) : 0.f;
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[t] += pj * bf2f((unsigned short)vraw[u][t]);
        }
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) s_part[slice * D + piece * 8 + t] = acc[t];
    __syncthreads();
    float *part_out = A.partial + ((long long)b * A.Hq + hq) * n_splits * D;
    if (tid < D) {
        float sum = 0.f;
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) sum += s_part[sl * D + tid];
        st_f(&part_out[(long long)sp * D + tid], sum);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // this wave's part of the partial output has arrived
    }
    __syncthreads();
    if (tid == 0) s_last = __hip_atomic_fetch_add(&cnt[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(n_splits - 1) ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    // ---- the last workgroup of the (sequence, head): the splits' partial outputs in split order (attn_decode_combine_kernel)
    if (tid < D) {
        float sum = 0.f;
        for (int t0 = 0; t0 < n_splits; t0 += 16) {                            // sixteen splits' values in flight, added in split order
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = ld_f(&part_out[(long long)min(t0 + u, n_splits - 1) * D + tid]);
#pragma unroll
            for (int u = 0; u < 16; ++u) if (t0 + u < n_splits) sum += v[u];
        }
        A.o[((long long)b * A.Hq + hq) * D + tid] = bf(sum);
    }
    // ready for the next launch: every workgroup of the (sequence, head) has read the statistics (it took its ticket after that)
    if (tid < 2 * n_splits) st_f(&st[tid], __uint_as_float(kDecodeOneEmpty));
    if (tid == 0) __hip_atomic_store(&cnt[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// head_dim 64 forward and backward: 2 = the lean kernels (LDS-DMA staging, softmax constants in the MFMA accumulators: the default), 1 = the round-2 LDS-DMA
// kernels, 0 = the register-staged kernels (tests / A-B).  Bits 8 / 9 / 10 send the forward / dQ / dK-dV kernel alone back to mode 1 (A-B of one kernel).
int g_attn_dma = 2;
// head_dim 256 backward: 0 (default) = the pair kernel (the dK pass and the dV pass each form the scores), 1 = the dK pass hands its probabilities to a dV kernel through the
// scratch (ecgb_attn_bwd_scratch_bytes then asks for batch * heads * seq^2 bf16 more).  Measured at the C5 shape: 270 + 131 us against the pair's 447, 0.66 against 0.68 ms
// for the whole backward -- not worth a transient S x S tensor in HBM by default (ecgb_set_attn_d256_pass_p; the same bits either way)
int g_attn_d256_pass_p = 0;

// waves per workgroup of the lean kernels: 4 (default; two workgroups per CU) or 8 (one workgroup per CU, a K / V tile serves twice the rows: measured 15 %
// slower end to end at the C3 shape although its tile loop is faster -- scripts/experiments/r03_attn_fwd_pingpong.hip.txt; kept for A-B: ecgb_set_attn_lean_waves)
int g_lean_waves = 4;
// forward / dQ lean kernels: HW = gcd(G, waves) query heads of a KV group per workgroup (a power of two), waves / HW blocks of 32 rows
// =====================================================================================================
// Round 4: head_dim 256 (Gemma) forward with LDS-DMA staging.  The register-staged kernel above spends more than half of a tile's 10 800 cycles moving the tile
// (phase timers at the C5 shape: 2 530 issuing the next tile's loads into registers, 1 170 - 2 750 writing them to LDS -- the transposed V image most -- and up to 2 100
// at the barrier behind the slower half) and keeps its 128 output accumulators in the register file's upper half as spill space of the forced VGPR form (445 registers:
// a copy out and back around every product).  Here
//   * K and V tiles (64 keys x 512 bytes each) go global -> LDS by LDS-DMA, 1 KiB (two rows) per wave-instruction, eight pieces per wave and operand, one tile ahead
//     into a ring of two (a trip is > 4 000 cycles of MFMA: one tile of flight time is enough, and two stages are what 160 KB hold);
//   * K lies as rows with 16-byte chunk c of row r at c ^ (r & 15): the sixteen rows sixteen lanes read as fragments fall on sixteen different bank groups;
//     V lies as it is in memory ([key][d]) with its 64-byte group w of key row r at w ^ (r & 7): the P.V operand is gathered by transposing reads whose eight key rows
//     then cover eight different groups.  Both swizzles are applied on the global side of the DMA;
//   * all LDS reads are asm statements with their own wait (a read hipcc can see next to LDS-DMA gets a vmcnt(0) in front of it);
//   * without the staging registers the kernel needs 328 registers instead of 445: 72 values live in the upper half between uses instead of 189.  (The P.V products written
//     out as asm statements on "+a" accumulators kept all 128 there, ran at the same speed and were WRONG now and then in output columns 0..31 -- a wait-state rule between
//     the statement's first MFMA and what hipcc places in front of it that 32 s_nops narrowed but did not close: the builtin stays.)
// The softmax arithmetic is attn_fwd_kernel's (running maximum per tile, scores scaled inside the exp2's fma): the same values up to the order of the fp32 sums.
// =====================================================================================================
__global__ __launch_bounds__(256) void attn_fwd_d256_kernel(AttnArgs A)
{
    constexpr int D = 256, kRow = D * 2, kTile = 64 * kRow, PPW = 8;      // bytes of a row / of a 64-key tile; 1 KiB pieces per wave and operand and tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // 2 x (K, V) tiles, the row's key mask, 64 tile flags
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
    const int tail_rows = A.S - last_tile * 64;                           // rows of the last tile that exist (64: whole)
    // piece i of this wave: tile rows (wave * 8 + i) * 2 + (lane >> 5), the lane's 16-byte slot lane & 31 of the LDS row takes global chunk slot ^ swizzle(row)
    unsigned offK[PPW], offV[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int r = (wave * PPW + i) * 2 + (lane >> 5), slot = lane & 31;
        offK[i] = (unsigned)(((long long)r * A.ldk + (slot ^ (r & 15)) * 8) * 2);
        offV[i] = (unsigned)(((long long)r * A.ldv + (slot ^ ((r & 7) << 2)) * 8) * 2);
    }
    const unsigned char *kb_next = reinterpret_cast<const unsigned char *>(K + rowbase * A.ldk), *vb_next = reinterpret_cast<const unsigned char *>(V + rowbase * A.ldv);
    const long long stepK = 128ll * A.ldk, stepV = 128ll * A.ldv;
    int t_next = 0;
    unsigned slot_next = 0;
    // piece i of the next tile's K and of its V image (round 6: a K piece behind each group of the score products, a V piece behind each group of the P.V products -- issued together at the top of the trip the sixteen pieces
    // held the wave for ~1 000 cycles with the matrix pipe idle; behind a group they run under its four MFMAs)
    auto issue_piece = [&](int which, int i) __attribute__((always_inline)) {      // which: 0 the K image, 1 the V image
        unsigned off = which ? offV[i] : offK[i];
        if (t_next == last_tile && tail_rows < 64) {                      // (uniform, once per workgroup at most) rows past the last key re-read it: masked as keys >= S
            const int r = (wave * PPW + i) * 2 + (lane >> 5), slot = lane & 31, rt = min(r, tail_rows - 1);
            off = which ? (unsigned)(((long long)rt * A.ldv + (slot ^ ((r & 7) << 2)) * 8) * 2) : (unsigned)(((long long)rt * A.ldk + (slot ^ (r & 15)) * 8) * 2);
        }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((which ? vb_next : kb_next) + off),
                                         (__attribute__((address_space(3))) void *)(smem + slot_next + which * kTile + (wave * PPW + i) * 1024), 16, 0, 0);
    };
    auto advance_next = [&]() { if (t_next < last_tile) { ++t_next; kb_next += stepK; vb_next += stepV; } slot_next ^= 2 * kTile; };
    auto issue_next = [&]() {                                             // UNCONDITIONAL (past the last tile the last one is issued again)
        if (t_next == last_tile && tail_rows < 64) {                      // (uniform, once per workgroup at most) rows past the last key re-read it: masked as keys >= S
#pragma unroll
            for (int i = 0; i < PPW; ++i) {
                const int r = (wave * PPW + i) * 2 + (lane >> 5), slot = lane & 31, rt = min(r, tail_rows - 1);
                const unsigned ok = (unsigned)(((long long)rt * A.ldk + (slot ^ (r & 15)) * 8) * 2), ov = (unsigned)(((long long)rt * A.ldv + (slot ^ ((r & 7) << 2)) * 8) * 2);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(kb_next + ok),
                                                 (__attribute__((address_space(3))) void *)(smem + slot_next + (wave * PPW + i) * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(vb_next + ov),
                                                 (__attribute__((address_space(3))) void *)(smem + slot_next + kTile + (wave * PPW + i) * 1024), 16, 0, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < PPW; ++i) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(kb_next + offK[i]),
                                                 (__attribute__((address_space(3))) void *)(smem + slot_next + (wave * PPW + i) * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(vb_next + offV[i]),
                                                 (__attribute__((address_space(3))) void *)(smem + slot_next + kTile + (wave * PPW + i) * 1024), 16, 0, 0);
            }
        }
        if (t_next < last_tile) { ++t_next; kb_next += stepK; vb_next += stepV; }
        slot_next ^= 2 * kTile;
    };
    issue_next();
    bf16x8 qf[D / 16];
    load_row_frags<D>(qf, Q, A.ldq, rowbase + qi, qvalid, h);
    lean_fill_mask<4>(lds_maskrow, A.mask + rowbase, A.S, (k_end + 63) & ~63);
    f32x16 accO[D / 32];
#pragma unroll
    for (int db = 0; db < D / 32; ++db) accO[db] = splat16(0.f);
    float m = -INFINITY, l = 0.f;
    const float sc = A.scale * kLog2e;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // tile 0 and Q
    __syncthreads();
    const unsigned long long padbits = lean_pad_bits(lds_maskrow, (k_end + 63) & ~63);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
    // this lane's K row fragment (key lr, k-step 0) inside a K image: k-step ks is the address ^ (ks << 5), key 32 + lr 16 KiB further (the swizzle repeats every 16 rows)
    const unsigned kbase = lr * kRow + ((h ^ (lr & 15)) << 4);      // (an offset: the XORs below must not see the LDS base)
    // this lane's first transposing read inside a V image for d block 0: block db is the address ^ (db << 6), the second read 8 keys = 4 KiB further, the (kb, s2) row groups go
    // into the offset field
    unsigned vbase;
    {
        const int a = lr >> 4, q = (lr & 15) >> 2, p = lr & 3;
        const int key = 4 * h + q;
        vbase = key * kRow + ((((key & 7) << 2) | (2 * a + (p >> 1))) << 4) + (p & 1) * 8;
    }
    unsigned img = 0;                                                     // byte offset of the current tile's ring slot
    for (int k0 = 0, it = 0; k0 < k_end; k0 += 64, ++it) {
        const float *lds_mask = lds_maskrow + k0;                         // (tile it + 1 goes into the other slot: its readers passed the barrier that ended the previous trip)
        if (k0 > wave_qmax) issue_next();
        if (k0 <= wave_qmax) {
            const bool need_mask = (k0 + 63 > qw0) || lean_tile_padded(padbits, it);
            f32x16 sacc[2] = {splat16(0.f), splat16(0.f)};
#pragma unroll
            for (int ks = 0; ks < D / 16; ks += 2) {                      // eight batches of four fragments (two k-steps x two key halves) and their four products
                bf16x8 f00, f01, f10, f11;
                lds_frags2x2_wait<0, 32 * kRow>(f00, f01, f10, f11, (kbase ^ (unsigned)(ks << 5)) + img + lds0, (kbase ^ (unsigned)((ks + 1) << 5)) + img + lds0);
                sacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f00, qf[ks], sacc[0], 0, 0, 0);
                sacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f01, qf[ks], sacc[1], 0, 0, 0);
                sacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f10, qf[ks + 1], sacc[0], 0, 0, 0);
                sacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f11, qf[ks + 1], sacc[1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                issue_piece(0, ks / 2);
                __builtin_amdgcn_sched_barrier(0);
            }
            float p[2][16];
            float tmax = -INFINITY;
            if (need_mask) {                                              // (uniform) the tile touches this wave's diagonal or holds padded keys
                f4v mv[2][4];
                const unsigned ma = (unsigned)(size_t)(__attribute__((address_space(3))) float *)const_cast<float *>(lds_mask) + 16 * h;
                lds_rows4_wait<0>(mv[0], ma);
                lds_rows4_wait<128>(mv[1], ma);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int kl = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;       // key inside the tile
                        const bool vis = (k0 + kl <= qi) & (mv[kb][r >> 2][r & 3] != 0.f);
                        const float v = vis ? sacc[kb][r] : -INFINITY;
                        p[kb][r] = v;
                        tmax = fmaxf(tmax, v);
                    }
            } else {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { p[kb][r] = sacc[kb][r]; tmax = fmaxf(tmax, sacc[kb][r]); }
            }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64)) * sc;            // p holds raw scores; sc > 0, so the maximum scales with them
            const float m_new = fmaxf(m, tmax);
            const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;      // no key visible yet: every p below is exp2(-inf) = 0
            const float alpha = fast_exp2(m - m_safe);                    // m = -inf -> 0 (accumulators are still zero then)
            float lsum = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float e = fast_exp2(__builtin_fmaf(p[kb][r], sc, -m_safe)); p[kb][r] = e; lsum += e; }
            l = l * alpha + lsum;
            m = m_new;
            if (__any(alpha != 1.f)) {                                    // once the running maxima have settled the accumulators need no rescale
#pragma unroll
                for (int db = 0; db < D / 32; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) accO[db][r] *= alpha;
            }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const bf16x8 pf0 = frag_from_acc(&p[kb][0]), pf1 = frag_from_acc(&p[kb][8]);
#pragma unroll
                for (int dp = 0; dp < D / 64; ++dp) {                     // pairs of d blocks: four transposed fragments (two k-steps x two blocks) and their four products
                    bf16x8 vfr[2][2];
                    const unsigned a0 = (vbase ^ (unsigned)((2 * dp) << 6)) + img + lds0 + kTile, a1 = (vbase ^ (unsigned)((2 * dp + 1) << 6)) + img + lds0 + kTile;
                    if (kb == 0) tr_frags4_wait<0, 16 * kRow>(vfr, a0, a0 + 8 * kRow, a1, a1 + 8 * kRow);
                    else tr_frags4_wait<32 * kRow, 48 * kRow>(vfr, a0, a0 + 8 * kRow, a1, a1 + 8 * kRow);
                    accO[2 * dp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[0][0], pf0, accO[2 * dp], 0, 0, 0);
                    accO[2 * dp + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[0][1], pf0, accO[2 * dp + 1], 0, 0, 0);
                    accO[2 * dp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[1][0], pf1, accO[2 * dp], 0, 0, 0);
                    accO[2 * dp + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[1][1], pf1, accO[2 * dp + 1], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    issue_piece(1, kb * 4 + dp);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            advance_next();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // this wave's pieces of tile it + 1 have landed
        __builtin_amdgcn_s_barrier();                                     // ... and everybody else's; all reads of tile it are done
        img ^= 2 * kTile;
    }
    const float lt = l + __shfl_xor(l, 32, 64);
    const float inv = lt > 0.f ? 1.f / lt : 0.f;
    store_accT<D / 32>(accO, A.o + (long long)hq * D, A.ldo, rowbase + qi, qvalid, h, inv);
    if (qvalid && h == 0) A.lse[((long long)b * A.Hq + hq) * A.S + qi] = lt > 0.f ? m + log2f(lt) : INFINITY;
}

// head_dim 256 dQ: attention_d256.hip (round 6; round 4's LDS-DMA kernel with asm reads is in the history).  The swizzle its images and the dK / dV kernel's share: 16-byte
// chunk c of row r lies at c ^ f(r), f(r) = (r & 3) << 2 | (r >> 2) & 3 (attention_common.inc) -- sixteen consecutive rows put one chunk on sixteen different bank groups, and
// the eight key rows of a transposing read put one 64-byte group on all four 64-byte positions of the bank cycle, twice each.

// head_dim 256 dK / dV pair kernel: attention_d256.hip (round 6; round 4's form with asm reads is in the history).

struct LeanGeom { int hw_log2; unsigned grid; };
LeanGeom lean_geom(int seq, int n_q_heads, int n_kv_heads, int batch)
{
    const int G = n_q_heads / n_kv_heads;
    int l2 = 0;
    while ((2 << l2) <= g_lean_waves && G % (2 << l2) == 0) ++l2;
    const int rows = (g_lean_waves >> l2) * 32;
    return {l2, (unsigned)((seq + rows - 1) / rows) * (unsigned)(n_q_heads >> l2) * (unsigned)batch};
}

int check_args(const AttnArgs &A, int D, const char *who)
{
    if (A.B <= 0 || A.S <= 0 || A.Hq <= 0 || A.Hkv <= 0 || A.Hq % A.Hkv) {
        ecgb::set_error(std::string(who) + ": bad shape");
        return ECGB_ERR_INVALID;
    }
    if (D != 64 && D != 128 && D != 256) {
        ecgb::set_error(std::string(who) + ": head_dim must be 64, 128 or 256");
        return ECGB_ERR_UNSUPPORTED;
    }
    if (A.ldq % 8 || A.ldk % 8 || A.ldv % 8 || A.ldo % 8) {
        ecgb::set_error(std::string(who) + ": row strides must be multiples of 8 elements");
        return ECGB_ERR_UNSUPPORTED;
    }
    if ((D == 64 ? 6 : 4) * 128 * D + 4 * (((long long)A.S + 63) & ~63ll) > 160 * 1024) {      // tile buffers + the row's key mask in LDS
        ecgb::set_error(std::string(who) + ": sequence too long for the key mask in LDS (head_dim 256: 8 192 keys, 64: 28 672)");
        return ECGB_ERR_UNSUPPORTED;
    }
    return ECGB_OK;
}

int launched(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) return ECGB_OK;
    ecgb::set_error(std::string(what) + ": " + hipGetErrorString(e));
    return ECGB_ERR_HIP;
}

// One kernel launch of the dispatch below: the dynamic-LDS attribute, then the launch; either failure is reported HERE with the kernel's name (a refused
// attribute used to skip the launch silently and leave it to a later hipGetLastError -- which does not see it: the outputs stayed unwritten, status OK).
template <typename Kern>
int launch_attn(Kern kern, dim3 grid, dim3 block, int lds, void *stream, const AttnArgs &A, const char *what)
{
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) {
        ecgb::set_error(std::string(what) + ": hipFuncSetAttribute(" + std::to_string(lds) + " bytes of dynamic LDS): " + hipGetErrorString(e));
        return ECGB_ERR_HIP;
    }
    hipLaunchKernelGGL(kern, grid, block, lds, (hipStream_t)stream, A);
    return launched(what);
}

}  // namespace

extern "C" int ecgb_attn_fwd(const void *q_dev, long long ldq, const void *k_dev, long long ldk, const void *v_dev, long long ldv,
                             const float *attn_mask_dev, void *o_dev, long long ldo, float *lse_dev, int batch, int seq,
                             int n_q_heads, int n_kv_heads, int head_dim, float scale, void *stream)
{
    AttnArgs A = {};
    A.q = (const unsigned short *)q_dev; A.k = (const unsigned short *)k_dev; A.v = (const unsigned short *)v_dev;
    A.ldq = ldq; A.ldk = ldk; A.ldv = ldv; A.mask = attn_mask_dev; A.o = (unsigned short *)o_dev; A.ldo = ldo; A.lse = lse_dev;
    A.B = batch; A.S = seq; A.Hq = n_q_heads; A.Hkv = n_kv_heads; A.scale = scale;
    int rc = check_args(A, head_dim, "ecgb_attn_fwd");
    if (rc) return rc;
    const dim3 grid((unsigned)((seq + 127) / 128) * (unsigned)n_q_heads * (unsigned)batch);   // 1-D: map_block() deals blocks to XCDs
    if (head_dim == 64) {
        const int lds = 6 * 128 * 64 + 4 * ((seq + 63) & ~63);
        if ((g_attn_dma & 3) == 2 && !(g_attn_dma & 0x100)) {
            const int ldl = 2 * kLeanRing * 128 * 64 + 4 * ((seq + 63) & ~63) + 256;   // K / V ring, the row's key mask, 64 tile flags
            const LeanGeom lg = lean_geom(seq, n_q_heads, n_kv_heads, batch);
            A.lean_hw_log2 = lg.hw_log2;
            auto kf = g_lean_waves == 8 ? attn_fwd_lean_kernel<8> : attn_fwd_lean_kernel<4>;
            return launch_attn(kf, dim3(lg.grid), dim3(g_lean_waves * 64), ldl, stream, A, "attn_fwd_lean_kernel");
        }
        if (g_attn_dma & 3) return launch_attn(attn_fwd_kernel<64, true>, grid, dim3(256), lds, stream, A, "attn_fwd_kernel<64, dma>");
        return launch_attn(attn_fwd_kernel<64>, grid, dim3(256), 4 * 128 * 64 + 4 * ((seq + 63) & ~63), stream, A, "attn_fwd_kernel<64>");
    }
    if (head_dim == 128) return launch_attn(attn_fwd_kernel<128>, grid, dim3(256), 4 * 128 * 128 + 4 * ((seq + 63) & ~63), stream, A, "attn_fwd_kernel<128>");
    // head_dim 256: LDS-DMA staging (round 4) where its two-stage ring, the key mask and the tile flags fit the CU's 160 KB and the per-lane DMA offsets 32 bits; mode 0 of
    // ecgb_set_attn_fwd_staging keeps the register-staged kernel (A/B, tests)
    const long long lds256 = 4ll * 64 * 512 + 4ll * ((seq + 63) & ~63) + 256;
    if ((g_attn_dma & 3) && lds256 <= 160 * 1024 && (ldk & 7) == 0 && (ldv & 7) == 0 && (((uintptr_t)k_dev | (uintptr_t)v_dev) & 15) == 0 && 64 * ldk * 2 + 512 <= 0xFFFFFFFFll && 64 * ldv * 2 + 512 <= 0xFFFFFFFFll)
        return launch_attn(attn_fwd_d256_kernel, grid, dim3(256), (int)lds256, stream, A, "attn_fwd_d256_kernel");
    return launch_attn(attn_fwd_kernel<256>, grid, dim3(256), 4 * 128 * 256 + 4 * ((seq + 63) & ~63), stream, A, "attn_fwd_kernel<256>");
}

namespace {
// head_dim 256 with few (batch, KV head, key block) workgroups: the causal key blocks differ 16-fold in work and one launch of <= 256
// workgroups takes as long as its heaviest one (key block 0).  Splitting a group's query heads over 2..4 workgroups gives the scheduler
// pieces to balance with (heaviest first); the price is fp32 partial slabs and a small ordered reduction.
int dkv_head_splits(int batch, int seq, int n_q_heads, int n_kv_heads, int head_dim)
{
    if (head_dim != 256) return 1;
    const int G = n_q_heads / n_kv_heads;
    const long long wgs = (long long)((seq + 127) / 128) * n_kv_heads * batch * 2;
    int hs = 1;
    while (wgs * hs < 1024 && hs * 2 <= 4 && G % (hs * 2) == 0) hs *= 2;
    return hs;
}
}  // namespace

namespace {
// head_dim 256 (round 6): room for the probabilities the dK pass hands to the dV kernel (AttnArgs::pbuf), bf16 over whole 128-key x 64-query tiles
size_t attn_bwd_p_bytes(int batch, int seq, int n_q_heads, int head_dim)
{
    if (head_dim != 256 || !g_attn_d256_pass_p) return 0;
    return (size_t)batch * n_q_heads * ((seq + 127) / 128) * ((seq + 63) / 64) * (128 * 64 * 2);
}
}  // namespace

extern "C" size_t ecgb_attn_bwd_scratch_bytes(int batch, int seq, int n_q_heads, int n_kv_heads, int head_dim)
{
    if (batch <= 0 || seq <= 0 || n_q_heads <= 0 || n_kv_heads <= 0 || n_q_heads % n_kv_heads) return 0;
    const int hs = dkv_head_splits(batch, seq, n_q_heads, n_kv_heads, head_dim);
    const size_t slabs = hs > 1 ? (size_t)2 * hs * batch * seq * n_kv_heads * head_dim * sizeof(float) : 0;
    return slabs + attn_bwd_p_bytes(batch, seq, n_q_heads, head_dim);
}

namespace {
int attn_bwd_impl(const void *q_dev, long long ldq, const void *k_dev, long long ldk, const void *v_dev, long long ldv,
                  const float *attn_mask_dev, const void *o_dev, const void *do_dev, long long ldo,
                  const float *lse_dev, float *delta_dev, void *dq_dev, long long lddq, void *dk_dev, long long lddk,
                  void *dv_dev, long long lddv, int batch, int seq, int n_q_heads, int n_kv_heads, int head_dim,
                  float scale, void *scratch_dev, size_t scratch_bytes, void *stream, const float *rope_cos, const float *rope_sin)
{
    AttnArgs A = {};
    A.rope_cos = rope_cos; A.rope_sin = rope_sin;
    A.q = (const unsigned short *)q_dev; A.k = (const unsigned short *)k_dev; A.v = (const unsigned short *)v_dev;
    A.ldq = ldq; A.ldk = ldk; A.ldv = ldv; A.mask = attn_mask_dev; A.o = (unsigned short *)const_cast<void *>(o_dev); A.ldo = ldo;
    A.lse = const_cast<float *>(lse_dev); A.d_o = (const unsigned short *)do_dev; A.delta = delta_dev;
    A.dq = (unsigned short *)dq_dev; A.dk = (unsigned short *)dk_dev; A.dv = (unsigned short *)dv_dev;
    A.lddq = lddq; A.lddk = lddk; A.lddv = lddv;
    A.B = batch; A.S = seq; A.Hq = n_q_heads; A.Hkv = n_kv_heads; A.scale = scale;
    int rc = check_args(A, head_dim, "ecgb_attn_bwd");
    if (rc) return rc;
    if (lddq % 4 || lddk % 4 || lddv % 4) { ecgb::set_error("ecgb_attn_bwd: gradient row strides must be multiples of 4"); return ECGB_ERR_UNSUPPORTED; }
    A.head_splits = dkv_head_splits(batch, seq, n_q_heads, n_kv_heads, head_dim);
    A.slab = (float *)scratch_dev;
    const size_t slab_bytes = A.head_splits > 1 ? (size_t)2 * A.head_splits * batch * seq * n_kv_heads * head_dim * sizeof(float) : 0;
    if (A.head_splits > 1 && (!scratch_dev || ((uintptr_t)scratch_dev & 15) || scratch_bytes < slab_bytes)) {
        ecgb::set_error("ecgb_attn_bwd: scratch for the partial dK / dV slabs (16-byte aligned; ecgb_attn_bwd_scratch_bytes() says how much) required for this shape");
        return ECGB_ERR_INVALID;
    }
    // (head_dim 256: with the whole of ecgb_attn_bwd_scratch_bytes() the dK pass hands its probabilities to the dV kernel; with the slabs' share alone the dV pass forms them again)
    const size_t p_bytes = attn_bwd_p_bytes(batch, seq, n_q_heads, head_dim);
    const bool pass_p = g_attn_d256_pass_p && p_bytes > 0 && scratch_dev && ((uintptr_t)scratch_dev & 15) == 0 && scratch_bytes >= slab_bytes + p_bytes;
    const unsigned nblk = (unsigned)((seq + 127) / 128);
    const dim3 gq(nblk * (unsigned)n_q_heads * (unsigned)batch);          // 1-D: map_block() deals blocks to XCDs
    const unsigned gk = nblk * (unsigned)n_kv_heads * (unsigned)batch;
    auto dq_generic = [&](auto kern, int D_, const char *what) { return launch_attn(kern, gq, dim3(256), 3 * 128 * D_ + 4 * ((seq + 63) & ~63), stream, A, what); };
    if (head_dim == 64) {
        const bool lean = (g_attn_dma & 3) == 2;
        if (g_attn_dma & 3) {
            const bool lq_lean = lean && !(g_attn_dma & 0x200);
            const int lq = (lq_lean ? 2 * kLeanRing : 6) * 128 * 64 + 4 * ((seq + 63) & ~63) + (lq_lean ? 256 : 0);
            const LeanGeom lg = lean_geom(seq, n_q_heads, n_kv_heads, batch);
            A.lean_hw_log2 = lg.hw_log2;
            auto kq = !lq_lean ? attn_bwd_dq_dma_kernel : g_lean_waves == 8 ? attn_bwd_dq_lean_kernel<8> : attn_bwd_dq_lean_kernel<4>;
            rc = launch_attn(kq, lq_lean ? dim3(lg.grid) : gq, dim3(lq_lean ? g_lean_waves * 64 : 256), lq, stream, A, lq_lean ? "attn_bwd_dq_lean_kernel" : "attn_bwd_dq_dma_kernel");
        } else {
            rc = dq_generic(attn_bwd_dq_kernel<64>, 64, "attn_bwd_dq_kernel<64>");
        }
        if (rc) return rc;
        if (g_attn_dma & 3) {
            const bool lk_lean = lean && !(g_attn_dma & 0x400);
            const int lk = lk_lean ? kLeanRing * (2 * 128 * 64 + g_lean_waves * 256) : 3 * (2 * 128 * 64 + 1024);
            auto kk = !lk_lean ? attn_bwd_dkv_dma_kernel : g_lean_waves == 8 ? attn_bwd_dkv_lean_kernel<8> : attn_bwd_dkv_lean_kernel<4>;
            const unsigned gkl = (unsigned)(((seq + g_lean_waves * 32 - 1) / (g_lean_waves * 32) + 1) / 2) * (unsigned)n_kv_heads * (unsigned)batch;   // key blocks in pairs (j, n - 1 - j)
            return launch_attn(kk, dim3(lk_lean ? gkl : gk), dim3(lk_lean ? g_lean_waves * 64 : 256), lk, stream, A, lk_lean ? "attn_bwd_dkv_lean_kernel" : "attn_bwd_dkv_dma_kernel");
        }
        return launch_attn(attn_bwd_dkv_kernel<64, 1, 0>, dim3(gk), dim3(256), 4 * 128 * 64 + 512, stream, A, "attn_bwd_dkv_kernel<64>");
    }
    if (head_dim == 128) {
        rc = dq_generic(attn_bwd_dq_kernel<128>, 128, "attn_bwd_dq_kernel<128>");
        if (rc) return rc;
        return launch_attn(attn_bwd_dkv_kernel<128, 1, 0>, dim3(gk), dim3(256), 4 * 128 * 128 + 512, stream, A, "attn_bwd_dkv_kernel<128>");
    }
    // head_dim 256: LDS-DMA staging where the two-stage ring fits (see ecgb_attn_fwd); mode 0 of ecgb_set_attn_fwd_staging keeps the register-staged kernels
    const long long lds256 = 4ll * 64 * 512 + 4ll * ((seq + 63) & ~63) + 256;
    const bool dma256 = (g_attn_dma & 3) && lds256 <= 160 * 1024 && (ldk & 7) == 0 && (ldv & 7) == 0 && (((uintptr_t)k_dev | (uintptr_t)v_dev) & 15) == 0 &&
                        64 * ldk * 2 + 512 <= 0xFFFFFFFFll && 64 * ldv * 2 + 512 <= 0xFFFFFFFFll;
    if (dma256) rc = ecgb_attn::launch_bwd_dq_d256(A, gq.x, seq, stream);
    else rc = dq_generic(attn_bwd_dq_kernel<256>, 256, "attn_bwd_dq_kernel<256>");
    if (rc) return rc;
    if ((g_attn_dma & 3) && (ldq & 7) == 0 && (ldo & 7) == 0 && (((uintptr_t)q_dev | (uintptr_t)do_dev) & 15) == 0 && 64 * ldq * 2 + 512 <= 0xFFFFFFFFll && 64 * ldo * 2 + 512 <= 0xFFFFFFFFll)
    {
        if (pass_p) {
            A.pbuf = reinterpret_cast<unsigned short *>(reinterpret_cast<unsigned char *>(scratch_dev) + slab_bytes);
            rc = ecgb_attn::launch_bwd_dk_then_dv_d256(A, gk * (unsigned)A.head_splits, stream);
        } else {
            rc = ecgb_attn::launch_bwd_dkv_pair_d256(A, gk * 2 * (unsigned)A.head_splits, stream);
        }
    }
    else
        rc = launch_attn(attn_bwd_dkv_pair_kernel<256>, dim3(gk * 2 * (unsigned)A.head_splits), dim3(256), 4 * 128 * 256 + 512, stream, A, "attn_bwd_dkv_pair_kernel<256>");
    if (rc) return rc;
    if (A.head_splits > 1) {
        hipLaunchKernelGGL(attn_dkv_reduce_kernel<256>, dim3(2048), dim3(256), 0, (hipStream_t)stream, A);
        return launched("attn_dkv_reduce_kernel<256>");
    }
    return ECGB_OK;
}
}  // namespace

extern "C" int ecgb_attn_bwd(const void *q_dev, long long ldq, const void *k_dev, long long ldk, const void *v_dev, long long ldv,
                             const float *attn_mask_dev, const void *o_dev, const void *do_dev, long long ldo,
                             const float *lse_dev, float *delta_dev, void *dq_dev, long long lddq, void *dk_dev, long long lddk,
                             void *dv_dev, long long lddv, int batch, int seq, int n_q_heads, int n_kv_heads, int head_dim,
                             float scale, void *scratch_dev, size_t scratch_bytes, void *stream)
{
    return attn_bwd_impl(q_dev, ldq, k_dev, ldk, v_dev, ldv, attn_mask_dev, o_dev, do_dev, ldo, lse_dev, delta_dev, dq_dev, lddq, dk_dev, lddk, dv_dev, lddv,
                         batch, seq, n_q_heads, n_kv_heads, head_dim, scale, scratch_dev, scratch_bytes, stream, nullptr, nullptr);
}

// ecgb_attn_bwd followed by RoPE's backward on dQ and dK (ecgb_rope, inverse), in the attention kernels' own stores: q and k are the ROTATED projections the forward
// saw, dq / dk come out as gradients of the unrotated ones.  rope_cos_dev / rope_sin_dev: [batch * seq, 32] fp32, the tables ecgb_rope takes.  head_dim 64 on the
// lean kernels with 16-byte aligned dq / dk rows only: ECGB_ERR_UNSUPPORTED otherwise, and the caller runs the two steps apart (the same bits either way).
extern "C" int ecgb_attn_bwd_rope(const void *q_dev, long long ldq, const void *k_dev, long long ldk, const void *v_dev, long long ldv,
                                  const float *attn_mask_dev, const void *o_dev, const void *do_dev, long long ldo,
                                  const float *lse_dev, float *delta_dev, void *dq_dev, long long lddq, void *dk_dev, long long lddk,
                                  void *dv_dev, long long lddv, const float *rope_cos_dev, const float *rope_sin_dev, int batch, int seq, int n_q_heads,
                                  int n_kv_heads, int head_dim, float scale, void *scratch_dev, size_t scratch_bytes, void *stream)
{
    if (!rope_cos_dev || !rope_sin_dev) { ecgb::set_error("ecgb_attn_bwd_rope: NULL table"); return ECGB_ERR_INVALID; }
    if (head_dim != 64 || (g_attn_dma & 0x703) != 2 || (lddq & 7) || (lddk & 7) || (((uintptr_t)dq_dev | (uintptr_t)dk_dev | (uintptr_t)rope_cos_dev | (uintptr_t)rope_sin_dev) & 15)) {
        ecgb::set_error("ecgb_attn_bwd_rope: head_dim 64 on the lean kernels with 16-byte aligned gradient rows only");
        return ECGB_ERR_UNSUPPORTED;
    }
    return attn_bwd_impl(q_dev, ldq, k_dev, ldk, v_dev, ldv, attn_mask_dev, o_dev, do_dev, ldo, lse_dev, delta_dev, dq_dev, lddq, dk_dev, lddk, dv_dev, lddv,
                         batch, seq, n_q_heads, n_kv_heads, head_dim, scale, scratch_dev, scratch_bytes, stream, rope_cos_dev, rope_sin_dev);
}

namespace {
int launch_attn_decode(const void *q_dev, const void *k_cache_dev, const void *v_cache_dev, long long ld, long long capacity,
                       const float *attn_mask_dev, long long mask_ld, void *o_dev, int batch, int kv_len, const int *kv_len_dev,
                       int n_q_heads, int n_kv_heads, int head_dim, float scale, void *stream)
{
    if (head_dim != 64 && head_dim != 128 && head_dim != 256) { ecgb::set_error("ecgb_attn_decode: head_dim must be 64, 128 or 256"); return ECGB_ERR_UNSUPPORTED; }
    if (batch <= 0 || (!kv_len_dev && (kv_len <= 0 || kv_len > capacity)) || n_q_heads % n_kv_heads || ld % 8) { ecgb::set_error("ecgb_attn_decode: bad shape"); return ECGB_ERR_INVALID; }
    const long long rows = kv_len_dev ? capacity : (long long)kv_len;
    const size_t lds = (std::max<size_t>((size_t)((rows + 3) & ~3ll), (size_t)kDecodeThreads * 8) + 2 * (kDecodeThreads / 64)) * 4;
    if (lds > 64 * 1024) { ecgb::set_error("ecgb_attn_decode: cache longer than ~16000 keys"); return ECGB_ERR_UNSUPPORTED; }
#define ECGB_DECODE(D_) hipLaunchKernelGGL(attn_decode_kernel<D_>, dim3((unsigned)n_q_heads, (unsigned)batch), dim3(kDecodeThreads), lds, (hipStream_t)stream, \
        (const unsigned short *)q_dev, (const unsigned short *)k_cache_dev, (const unsigned short *)v_cache_dev, ld, capacity, attn_mask_dev, mask_ld, \
        (unsigned short *)o_dev, kv_len, kv_len_dev, n_q_heads, n_kv_heads, scale)
    if (head_dim == 64) ECGB_DECODE(64); else if (head_dim == 128) ECGB_DECODE(128); else ECGB_DECODE(256);
#undef ECGB_DECODE
    return launched("attn_decode_kernel");
}

// cache[b, *len - 1, :] = src[b, col_off : col_off + width]   (the new token's roped K | V, position from device memory)
__global__ __launch_bounds__(256) void kv_append_kernel(const unsigned short *src, long long src_ld, long long col_off, int width,
                                                        unsigned short *cache, long long cap, const int *len_dev, int batch)
{
    const long long row = *len_dev - 1;
    const int per_row = width / 8;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < batch * per_row; i += gridDim.x * blockDim.x) {
        const int b = i / per_row, c = i % per_row;
        *reinterpret_cast<bf16x8 *>(cache + ((long long)b * cap + row) * width + c * 8) =
            *reinterpret_cast<const bf16x8 *>(src + (long long)b * src_ld + col_off + c * 8);
    }
}
}  // namespace

extern "C" int ecgb_attn_decode(const void *q_dev, const void *k_cache_dev, const void *v_cache_dev, long long ld, long long capacity,
                                const float *attn_mask_dev, long long mask_ld, void *o_dev, int batch, int kv_len, int n_q_heads,
                                int n_kv_heads, int head_dim, float scale, void *stream)
{
    return launch_attn_decode(q_dev, k_cache_dev, v_cache_dev, ld, capacity, attn_mask_dev, mask_ld, o_dev, batch, kv_len, nullptr,
                              n_q_heads, n_kv_heads, head_dim, scale, stream);
}

extern "C" size_t ecgb_attn_decode_split_scratch_bytes(long long capacity, int batch, int n_q_heads, int head_dim, int n_splits)
{
    if (capacity <= 0 || batch <= 0 || n_q_heads <= 0 || n_splits <= 0) return 0;
    // a split never holds more than kSplitMaxChunk keys: size for the splits the longest cache (capacity keys) can need
    const long long need_splits = (capacity + kSplitMaxChunk - 1) / kSplitMaxChunk;
    if (need_splits > n_splits) n_splits = (int)need_splits;
    return (size_t)batch * n_q_heads * ((size_t)capacity + (size_t)n_splits * (2 + (size_t)head_dim)) * sizeof(float);
}

namespace {
int launch_attn_decode_split(const char *who, const void *q_dev, const void *k_cache_dev, const void *v_cache_dev, long long ld, long long capacity,
                             const float *attn_mask_dev, long long mask_ld, void *o_dev, int batch, int kv_len, const int *kv_len_dev, int n_q_heads,
                             int n_kv_heads, int head_dim, float scale, int n_splits, void *scratch_dev, size_t scratch_bytes, void *stream)
{
    if (head_dim != 64 && head_dim != 128 && head_dim != 256) { ecgb::set_error(std::string(who) + ": head_dim must be 64, 128 or 256"); return ECGB_ERR_UNSUPPORTED; }
    if (!q_dev || !k_cache_dev || !v_cache_dev || !attn_mask_dev || !o_dev || !scratch_dev || batch <= 0 || (!kv_len_dev && (kv_len <= 0 || kv_len > capacity)) ||
        capacity <= 0 || n_q_heads <= 0 || n_kv_heads <= 0 || n_q_heads % n_kv_heads || ld % 8 || n_splits <= 0 || n_splits > 1024) {
        ecgb::set_error(std::string(who) + ": bad argument");
        return ECGB_ERR_INVALID;
    }
    if (scratch_bytes < ecgb_attn_decode_split_scratch_bytes(capacity, batch, n_q_heads, head_dim, n_splits)) {
        ecgb::set_error(std::string(who) + ": scratch too small (ecgb_attn_decode_split_scratch_bytes)");
        return ECGB_ERR_INVALID;
    }
    // the split count is final BEFORE the scratch is laid out: stats and partial are indexed by gridDim.x.  A length in device memory can grow to the capacity.
    const long long longest = kv_len_dev ? capacity : (long long)kv_len;
    if ((longest + n_splits - 1) / n_splits > kSplitMaxChunk) n_splits = (int)((longest + kSplitMaxChunk - 1) / kSplitMaxChunk);
    float *scores = (float *)scratch_dev;
    float *stats = scores + (size_t)batch * n_q_heads * (size_t)capacity;
    float *partial = stats + (size_t)batch * n_q_heads * (size_t)n_splits * 2;
    const int chunk = kv_len_dev ? 0 : (kv_len + n_splits - 1) / n_splits;
    const dim3 grid((unsigned)n_splits, (unsigned)n_q_heads, (unsigned)batch), gc((unsigned)n_q_heads, (unsigned)batch);
#define ECGB_SPLIT(D_) do { \
        hipLaunchKernelGGL(attn_decode_scores_kernel<D_>, grid, dim3(kSplitThreads), 0, (hipStream_t)stream, (const unsigned short *)q_dev, \
            (const unsigned short *)k_cache_dev, ld, capacity, attn_mask_dev, mask_ld, scores, stats, kv_len, chunk, kv_len_dev, n_q_heads, n_kv_heads, scale); \
        hipLaunchKernelGGL(attn_decode_values_kernel<D_>, grid, dim3(kSplitThreads), 0, (hipStream_t)stream, (const unsigned short *)v_cache_dev, \
            ld, capacity, scores, stats, partial, kv_len, chunk, kv_len_dev, n_q_heads, n_kv_heads); \
        hipLaunchKernelGGL(attn_decode_combine_kernel<D_>, gc, dim3(D_), 0, (hipStream_t)stream, partial, (unsigned short *)o_dev, n_splits, n_q_heads); \
    } while (0)
    if (head_dim == 64) ECGB_SPLIT(64); else if (head_dim == 128) ECGB_SPLIT(128); else ECGB_SPLIT(256);
#undef ECGB_SPLIT
    return launched("attn_decode_split kernels");
}
}  // namespace

extern "C" int ecgb_attn_decode_split(const void *q_dev, const void *k_cache_dev, const void *v_cache_dev, long long ld, long long capacity,
                                      const float *attn_mask_dev, long long mask_ld, void *o_dev, int batch, int kv_len, int n_q_heads,
                                      int n_kv_heads, int head_dim, float scale, int n_splits, void *scratch_dev, size_t scratch_bytes,
                                      void *stream)
{
    return launch_attn_decode_split("ecgb_attn_decode_split", q_dev, k_cache_dev, v_cache_dev, ld, capacity, attn_mask_dev, mask_ld, o_dev, batch, kv_len, nullptr,
                                    n_q_heads, n_kv_heads, head_dim, scale, n_splits, scratch_dev, scratch_bytes, stream);
}

extern "C" int ecgb_attn_decode_split_dyn(const void *q_dev, const void *k_cache_dev, const void *v_cache_dev, long long ld, long long capacity,
                                          const float *attn_mask_dev, long long mask_ld, void *o_dev, int batch, const int *kv_len_dev, int n_q_heads,
                                          int n_kv_heads, int head_dim, float scale, int n_splits, void *scratch_dev, size_t scratch_bytes,
                                          void *stream)
{
    if (!kv_len_dev) { ecgb::set_error("ecgb_attn_decode_split_dyn: NULL length pointer"); return ECGB_ERR_INVALID; }
    return launch_attn_decode_split("ecgb_attn_decode_split_dyn", q_dev, k_cache_dev, v_cache_dev, ld, capacity, attn_mask_dev, mask_ld, o_dev, batch, 0, kv_len_dev,
                                    n_q_heads, n_kv_heads, head_dim, scale, n_splits, scratch_dev, scratch_bytes, stream);
}

// floats of scratch ecgb_attn_decode_one needs (statistics, partial outputs, counters): before the first call the FIRST batch * n_q_heads * n_splits * 2 floats (the
// statistics) must hold the bit pattern 0x7FC0DEAD and the LAST batch * n_q_heads * 2 words (the counters) zero; every launch leaves them so
extern "C" size_t ecgb_attn_decode_one_scratch_floats(int batch, int n_q_heads, int head_dim, int n_splits)
{
    if (batch <= 0 || n_q_heads <= 0 || head_dim <= 0 || n_splits <= 0) return 0;
    return (size_t)batch * n_q_heads * ((size_t)n_splits * (2 + (size_t)head_dim) + 2);
}

// A decode step's RoPE + cache append + attention in ONE launch, from the step's raw q|k|v projection (q and k NOT rotated in place): the bits of ecgb_rope_append
// followed by ecgb_attn_decode_split[_dyn] with the same n_splits.  ECGB_ERR_UNSUPPORTED outside head_dim 64 / 128 / 256, n_splits <= 64, at most 2048 keys a split and
// n_splits * n_q_heads * batch <= two workgroups a CU (they wait for each other inside the launch: all of them must be resident).
extern "C" int ecgb_attn_decode_one(const void *qkv_dev, long long ld_qkv, const float *cos_dev, const float *sin_dev, void *cache_dev, long long ld, long long capacity,
                                    const float *attn_mask_dev, long long mask_ld, void *o_dev, int batch, int kv_len, const int *kv_len_dev, int n_q_heads, int n_kv_heads,
                                    int head_dim, float scale, int n_splits, float *scratch_dev, size_t scratch_floats, void *stream)
{
    if (!qkv_dev || !cos_dev || !sin_dev || !cache_dev || !attn_mask_dev || !o_dev || !scratch_dev || batch <= 0 || n_q_heads <= 0 || n_kv_heads <= 0 || n_splits <= 0 ||
        n_q_heads % n_kv_heads || capacity <= 0 || (!kv_len_dev && (kv_len <= 0 || kv_len > capacity))) {
        ecgb::set_error("ecgb_attn_decode_one: bad argument");
        return ECGB_ERR_INVALID;
    }
    const long long longest = kv_len_dev ? capacity : (long long)kv_len;
    int dev = 0, n_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 128;
    // (two workgroups of 256 threads and 20 KB of LDS a CU are resident whatever else the kernel needs: the launch's workgroups wait for each other)
    if ((head_dim != 64 && head_dim != 128 && head_dim != 256) || n_splits > 64 || (longest + n_splits - 1) / n_splits > kSplitMaxChunk ||
        (long long)n_splits * n_q_heads * batch > 2ll * n_cu || ld % 8 || ld_qkv % 8 || ((uintptr_t)qkv_dev & 15) || ((uintptr_t)cache_dev & 15) ||
        scratch_floats < ecgb_attn_decode_one_scratch_floats(batch, n_q_heads, head_dim, n_splits)) {
        ecgb::set_error("ecgb_attn_decode_one: head_dim 64 / 128 / 256, at most 64 splits of at most 2048 keys, at most two workgroups a CU, scratch of ecgb_attn_decode_one_scratch_floats()");
        return ECGB_ERR_UNSUPPORTED;
    }
    DecodeOneArgs A;
    A.qkv = (const unsigned short *)qkv_dev; A.ld_qkv = ld_qkv; A.cs = cos_dev; A.sn = sin_dev; A.cache = (unsigned short *)cache_dev; A.cap = capacity; A.ld = ld;
    A.mask = attn_mask_dev; A.mask_ld = mask_ld;
    A.stats = scratch_dev;
    A.partial = A.stats + (size_t)batch * n_q_heads * n_splits * 2;
    A.cnt = reinterpret_cast<unsigned *>(A.partial + (size_t)batch * n_q_heads * n_splits * head_dim);
    A.o = (unsigned short *)o_dev; A.len_dev = kv_len_dev; A.len_arg = kv_len; A.Hq = n_q_heads; A.Hkv = n_kv_heads; A.scale = scale;
    const dim3 grid((unsigned)n_splits, (unsigned)n_q_heads, (unsigned)batch);
    if (head_dim == 64) hipLaunchKernelGGL(attn_decode_one_kernel<64>, grid, dim3(kSplitThreads), 0, (hipStream_t)stream, A);
    else if (head_dim == 128) hipLaunchKernelGGL(attn_decode_one_kernel<128>, grid, dim3(kSplitThreads), 0, (hipStream_t)stream, A);
    else hipLaunchKernelGGL(attn_decode_one_kernel<256>, grid, dim3(kSplitThreads), 0, (hipStream_t)stream, A);
    return launched("attn_decode_one_kernel");
}

extern "C" int ecgb_attn_decode_dyn(const void *q_dev, const void *k_cache_dev, const void *v_cache_dev, long long ld, long long capacity,
                                    const float *attn_mask_dev, long long mask_ld, void *o_dev, int batch, const int *kv_len_dev,
                                    int n_q_heads, int n_kv_heads, int head_dim, float scale, void *stream)
{
    if (!kv_len_dev) { ecgb::set_error("ecgb_attn_decode_dyn: NULL length pointer"); return ECGB_ERR_INVALID; }
    return launch_attn_decode(q_dev, k_cache_dev, v_cache_dev, ld, capacity, attn_mask_dev, mask_ld, o_dev, batch, 0, kv_len_dev,
                              n_q_heads, n_kv_heads, head_dim, scale, stream);
}

extern "C" int ecgb_kv_append(const void *src_dev, long long src_ld, long long col_off, int width, void *cache_dev, long long capacity,
                              int batch, const int *kv_len_dev, void *stream)
{
    if (!src_dev || !cache_dev || !kv_len_dev || width <= 0 || width % 8 || src_ld % 8 || col_off % 8 || batch <= 0) {
        ecgb::set_error("ecgb_kv_append: bad argument");
        return ECGB_ERR_INVALID;
    }
    const int items = batch * (width / 8);
    hipLaunchKernelGGL(kv_append_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short *)src_dev, src_ld, col_off, width, (unsigned short *)cache_dev, capacity, kv_len_dev, batch);
    return launched("kv_append_kernel");
}

#ifdef ECGB_PROFILE
extern "C" void ecgb_debug_attn_profile(unsigned long long *out8, int reset)
{
    if (out8) (void)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_attn_prof), 64 * sizeof(unsigned long long));
    if (reset) { unsigned long long z[64] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_attn_prof), z, sizeof z); }
}
#endif

// head_dim 64, forward and backward: 2 = the lean LDS-DMA kernels (default), 1 = the round-2 LDS-DMA kernels, 0 = the register-staged kernels (tests, A/B);
// | 0x100 / 0x200 / 0x400 with mode 2: the forward / dQ / dK-dV kernel alone stays on mode 1
extern "C" int ecgb_set_attn_lean_waves(int waves)
{
    if (waves != 4 && waves != 8) { ecgb::set_error("ecgb_set_attn_lean_waves: 4 or 8"); return ECGB_ERR_INVALID; }
    g_lean_waves = waves;
    return ECGB_OK;
}
extern "C" int ecgb_set_attn_d256_pass_p(int on) { g_attn_d256_pass_p = on ? 1 : 0; return ECGB_OK; }

extern "C" int ecgb_set_attn_fwd_staging(int mode)
{
    if ((mode & 3) == 3 || (mode & ~0x703)) { ecgb::set_error("ecgb_set_attn_fwd_staging: mode 0, 1 or 2 (| 0x100 | 0x200 | 0x400)"); return ECGB_ERR_INVALID; }
    g_attn_dma = mode;
    return ECGB_OK;
}
