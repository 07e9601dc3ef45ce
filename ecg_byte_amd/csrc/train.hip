// train.hip -- BPE tokenizer training on MI355X (gfx950).
//
// Reference: rust_bpe.byte_pair_encoding, ecg_byte/rust_bpe/src/lib.rs:58-125, with
//   get_stats  lib.rs:28-48   histogram of ALL adjacent pairs (overlapping windows count),
//   arg-max    lib.rs:92-94   most frequent pair,
//   merge      lib.rs:10-26   left-to-right, non-overlapping replacement, ids compacted.
// Tie-break among equal counts: the reference's depends on hash-map iteration order and thread
// schedule; here it is DEFINED as the numerically smallest (left, right).
//
// The reference recounts every pair on every merge (two full passes over the ids per merge,
// hash-map inserts dominating).  Here the ids live in HBM as uint32 and the pair counts in a dense
// V x V table (V = 256 + num_merges) that is built once and then only PATCHED: a merge changes
// counts only next to its sites, so the rewrite pass emits ~5 count deltas per site.  Per merge:
//   1. arg-max: every row of the table keeps its maximum, only the rows a delta could have changed are read again (rowmax_kernel; the sharded form reads the
//      live V_cur x V part, argmax_partial_kernel)
//   2. survivor counts: each workgroup streams its contiguous range of tiles and leaves ONE record   (reads N_i)
//   3. rewrite + compaction + count deltas: each workgroup chains the records below its own (its output offset, the run parity entering its range), then walks its
//      range in order, carrying both   (reads N_i, writes N_{i+1})
// (The one-rank trainer's default since round 5 is train_seg.inc: every wave keeps its range of ids in a fixed slot and compacts inside it -- steps 2 and 3 become ONE pass,
//  ids 16 bits wide while the vocabulary fits.  The two-pass form below is what the sharded entry points run, and ecgb_set_bpe_train_form(2).)
// Runs of one symbol merged with itself ("aaa" -> "Xa") need the offset parity inside the run;
// tiles and per-thread spans are even-sized, so parity is carried by a "last non-uniform span"
// look-up instead of a full segmented scan.  No host synchronisation inside the merge loop: the
// chosen pair, the current length and the stop flag stay on the device.
#include <hip/hip_runtime.h>

#include <string>

#include "tokenizer.hpp"
#include <type_traits>

#ifndef ECGB_TRAIN_PER_THREAD
#define ECGB_TRAIN_PER_THREAD 8
#endif
#ifndef ECGB_RW_WAVES
#define ECGB_RW_WAVES 5
#endif

namespace {

constexpr int kThreads = 256;
constexpr int kPerThread = ECGB_TRAIN_PER_THREAD;       // even: a span of one repeated symbol keeps run parity
constexpr uint32_t kTile = kThreads * kPerThread;     // 2048 ids per tile
#ifndef ECGB_TRAIN_GRID
#define ECGB_TRAIN_GRID 1280                                 // five rewrite workgroups a CU (its registers and LDS allow five): every workgroup resident, one round
#endif
constexpr uint32_t kGrid = ECGB_TRAIN_GRID;            // workgroups of the two passes (a multiple of kThreads: the rewrite's lanes share out the records)
constexpr uint32_t kHashSlots = 2048;                 // per-workgroup LDS table of count deltas
constexpr uint32_t kEmpty = 0xFFFFFFFFu;

struct TrainState {
    uint64_t n_cur;        // ids in the current buffer (64-bit: the reference's corpus is up to 6e9 symbols, tokenizer_utils.py:79-93)
    uint64_t n_next;       // ids after the merge in flight
    uint32_t left, right;  // pair chosen for the merge in flight
    uint32_t new_id;
    uint32_t active;       // 0 once no pair is left (lib.rs:88-90) -- later merges are no-ops
    uint32_t done;         // merges performed
};

struct TileInfo {
    uint32_t count0;       // survivors if the run entering the tile has even length so far
    uint32_t flags;        // bit0: lead run has odd length in this tile (count1 = count0 - 1 ... see below)
                           // bit1: whole tile is `left` (l == r merges);  bit2: parity of trailing run of `left`
};

// What a shard of a corpus split over ranks knows about its neighbours for the merge in flight (ecgb_bpe_shard_*): the id before its
// first element, the three ids after its last one, and -- for merges of a symbol with itself -- the parity of the run of `left` that
// ends just before it.  A single-rank run has no neighbours: everything kEmpty / 0.
struct Halo {
    uint32_t prev;
    uint32_t next[3];
    uint32_t lead_par;
    uint32_t pad[3];
};

struct TrainArgs {
    const Halo *halo;      // never NULL
    long long *slab;       // sharded: count deltas go to this [6][V] slab (all-reduced, then applied by every rank); NULL: straight into the table
    TrainState *st;
    uint64_t *table;       // V x V pair counts (64-bit: one pair can occur more than 2^32 times in a corpus that long)
    uint32_t V;
    uint32_t *buf[2];      // ping-pong id buffers
    TileInfo *tiles;       // per RANGE of tiles (a workgroup's contiguous share of the buffer, tile_range): what TileInfo says of a tile, of the whole range
    uint64_t *partial;     // kGrid arg-max partials: 2 words each, count and ~key
    uint64_t *row_cnt;     // per row of the table: its largest count ...
    uint32_t *row_key;     //  ... and ~index of that cell (the smallest index among equals), exact unless row_dirty
    uint32_t *row_dirty;   // per row: not read yet since the table was built (rowmax_kernel reads it whole the first time)
    uint32_t *pairs_out;   // 2 x num_merges
    uint64_t n0;           // initial length
};

// A workgroup barrier that orders LDS only.  __syncthreads() also drains every global load and store in flight (s_waitcnt vmcnt(0) before s_barrier): in the tile
// loops below that is the NEXT tile on its way in and the last tile's survivors on their way out -- the loops never read global memory another thread of the
// workgroup wrote, so their barriers need not wait for either.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Workgroup b of G walks tiles [first, first + count): contiguous and balanced.  A workgroup that walks its tiles IN ORDER carries the output offset and the run parity from
// tile to tile itself, so the count pass leaves one record per workgroup, not per tile, and the rewrite chains at most G records whatever the corpus
// (per-tile records: a scan of 29 000 to 58 000 entries by one workgroup between the two passes, 9 to 16 % of a merge).
__device__ __forceinline__ void tile_range(uint32_t n_tiles, uint32_t b, uint32_t G, uint32_t &first, uint32_t &count)
{
    const uint32_t q = n_tiles / G, rem = n_tiles % G;
    first = b * q + min(b, rem);
    count = q + (b < rem ? 1u : 0u);
}

// ---- LDS delta table --------------------------------------------------------------------
// arg-max order: the larger count, then the larger ~index (= the smaller (left, right)); a count of 0 is no pair at all
__device__ __forceinline__ bool better(unsigned long long c, uint32_t k, unsigned long long bc, uint32_t bk)
{
    return c > bc || (c == bc && c != 0 && k > bk);
}

// A count delta goes into the table and nothing comes back: which rows' maxima it moved, rowmax_kernel finds out itself (marking the rows here -- comparing with the
// row's kept maximum, or with the add's old count -- measured the same and put loads and returning atomics into a flush that otherwise only fires and forgets).
struct Tab { uint64_t *table; };
__device__ __forceinline__ Tab tab_of(const TrainArgs &A) { return Tab{A.table}; }
__device__ __forceinline__ void table_add(const Tab &T, uint32_t key, int d)
{
#ifdef ECGB_DEV_NO_TABLE_ADD       // timing-only builds (wrong merges): what do the table's atomics cost?
    if (d == 0x7fffffff)
#endif
    atomicAdd(reinterpret_cast<unsigned long long *>(&T.table[key]), (unsigned long long)(long long)d);   // two's complement: -1 adds 2^64 - 1
}

// Every count delta of a merge has `left`, `right` or the new id as one of its two ids (a pair disappears only next to a site, a
// pair appears only next to a merged id), so the deltas of one merge fit six vectors of V: rows left / right / new, columns left /
// right / new.  slab_index is the (first-match) place of pair (a, b); slab_pair its inverse.
struct SlabKey { uint32_t l, r, x, V; };
__device__ __forceinline__ uint32_t slab_index(const SlabKey &K, uint32_t a, uint32_t b)
{
    if (a == K.l) return b;
    if (a == K.r) return K.V + b;
    if (a == K.x) return 2 * K.V + b;
    if (b == K.l) return 3 * K.V + a;
    if (b == K.r) return 4 * K.V + a;
    return 5 * K.V + a;                      // b == K.x
}

__device__ __forceinline__ void delta_add(uint32_t *s_key, int *s_val, const Tab &table, uint32_t key, int d)
{
    uint32_t h = (key * 2654435761u) >> 21;   // 11 bits
#pragma unroll 1                                  // (unrolled, the 32 probe chains of a span's walk were most of a 100 KB kernel: more than the instruction cache two CUs share)
    for (int probe = 0; probe < 16; ++probe) {
        const uint32_t prev = atomicCAS(&s_key[h], kEmpty, key);
        if (prev == kEmpty || prev == key) { atomicAdd(&s_val[h], d); return; }
        h = (h + 1) & (kHashSlots - 1);
    }
    table_add(table, key, d);                 // table full around h: go to memory (sharded runs pass the slab here, see rewrite_kernel)
}

// the same with the overflow going to the slab when the run is sharded
__device__ __forceinline__ void delta_add_s(uint32_t *s_key, int *s_val, const Tab &table, long long *slab, const SlabKey &K, uint32_t key, int d)
{
    uint32_t h = (key * 2654435761u) >> 21;
#pragma unroll 1
    for (int probe = 0; probe < 16; ++probe) {
        const uint32_t prev = atomicCAS(&s_key[h], kEmpty, key);
        if (prev == kEmpty || prev == key) { atomicAdd(&s_val[h], d); return; }
        h = (h + 1) & (kHashSlots - 1);
    }
    if (slab) atomicAdd(reinterpret_cast<unsigned long long *>(&slab[slab_index(K, key / K.V, key % K.V)]), (unsigned long long)(long long)d);
    else table_add(table, key, d);
}

__device__ __forceinline__ void delta_flush(uint32_t *s_key, int *s_val, const Tab &table, long long *slab = nullptr, SlabKey K = SlabKey{0, 0, 0, 0})
{
    lds_barrier();
    for (uint32_t i = threadIdx.x; i < kHashSlots; i += kThreads) {       // (all of a lane's adds issued before the first old count is looked at: the eight more registers
        const uint32_t k = s_key[i];                                      //  cost the rewrite a wave per SIMD -- built, 286 -> 335 ms over 4 000 merges)
        const int v = s_val[i];
        if (k != kEmpty && v != 0) {
            if (slab) atomicAdd(reinterpret_cast<unsigned long long *>(&slab[slab_index(K, k / K.V, k % K.V)]), (unsigned long long)(long long)v);
            else table_add(table, k, v);
        }
        s_key[i] = kEmpty;
        s_val[i] = 0;
    }
    lds_barrier();
}

// ---- 0. bytes -> ids, initial histogram ----------------------------------------------------
__global__ __launch_bounds__(kThreads) void init_kernel(TrainArgs A, const uint8_t *text)
{
    __shared__ uint32_t s_key[kHashSlots];
    __shared__ int s_val[kHashSlots];
    for (uint32_t i = threadIdx.x; i < kHashSlots; i += kThreads) { s_key[i] = kEmpty; s_val[i] = 0; }
    __syncthreads();
    const uint64_t n = A.n0;
    const uint32_t n_tiles = (uint32_t)((n + kTile - 1) / kTile);
    // A lane takes kPerThread consecutive bytes: the pairs of a stretch of one symbol are one pair over and over, counted in a register and added once (a quantised ECG
    // is mostly such stretches: one LDS add per pair was 2.8 ms for 1.2e8 symbols, every lane of a wave on the same few slots); the table leaves LDS once, at the end.
    for (uint32_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const uint64_t i0 = (uint64_t)t * kTile + (uint64_t)threadIdx.x * kPerThread;
        uint32_t a[kPerThread + 1];
#pragma unroll
        for (int k = 0; k <= kPerThread; ++k) a[k] = (i0 + k < n) ? (uint32_t)text[i0 + k] : kEmpty;
        uint32_t run_key = kEmpty;
        int run_n = 0;
#pragma unroll
        for (int k = 0; k < kPerThread; ++k) {
            if (i0 + k < n) A.buf[0][i0 + k] = a[k];                // lib.rs:72
            if (a[k] != kEmpty && a[k + 1] != kEmpty) {             // ids.windows(2)
                const uint32_t key = a[k] * A.V + a[k + 1];
                if (key == run_key) ++run_n;
                else {
                    if (run_n) delta_add(s_key, s_val, tab_of(A), run_key, run_n);
                    run_key = key; run_n = 1;
                }
            }
        }
        if (run_n) delta_add(s_key, s_val, tab_of(A), run_key, run_n);
    }
    delta_flush(s_key, s_val, tab_of(A));
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        A.st->n_cur = n; A.st->n_next = n; A.st->active = 1; A.st->done = 0;
        A.st->left = A.st->right = A.st->new_id = 0;
    }
}

// ---- 1. arg-max ------------------------------------------------------------------------------
// order: larger count first, then smaller (left, right) = larger ~index.  A zero count never wins.
__global__ __launch_bounds__(kThreads) void argmax_partial_kernel(TrainArgs A, uint32_t merge_index)
{
    __shared__ unsigned long long s_cnt[kThreads / 64];
    __shared__ uint32_t s_key[kThreads / 64];
    const uint32_t v_cur = 256u + merge_index;            // ids that can exist so far
    const uint64_t total = (uint64_t)v_cur * A.V;
    unsigned long long best = 0;                          // (count, ~index): larger count first, then smaller index
    uint32_t bkey = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (uint64_t)gridDim.x * kThreads) {
        const unsigned long long c = A.table[i];
        const uint32_t k = ~(uint32_t)i;
        if (better(c, k, best, bkey)) { best = c; bkey = k; }
    }
    for (int d = 32; d > 0; d >>= 1) {
        const unsigned long long oc = __shfl_down(best, d, 64);
        const uint32_t ok = __shfl_down(bkey, d, 64);
        if (better(oc, ok, best, bkey)) { best = oc; bkey = ok; }
    }
    if ((threadIdx.x & 63) == 0) { s_cnt[threadIdx.x >> 6] = best; s_key[threadIdx.x >> 6] = bkey; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kThreads / 64; ++w) if (better(s_cnt[w], s_key[w], best, bkey)) { best = s_cnt[w]; bkey = s_key[w]; }
        A.partial[2 * blockIdx.x] = best;
        A.partial[2 * blockIdx.x + 1] = bkey;
    }
}

// The one-rank trainer's arg-max, first half: a merge changes the counts of a few rows of the table (those of its two ids, of the new id, and of the ids that stood before a
// site), so each row keeps its own maximum (row_cnt, row_key); only the rows whose kept cell no longer holds the kept count and the newest id's row are read
// again, and every other row looks at its one cell in the newest id's column -- not the
// (256 + i) x V table of every merge: 145 MB at 4 000 merges, a tenth of a merge's time and the traffic that pushed the id buffers out of the memory-side cache.
// Workgroup b owns rows b, b + G, ..., a lane a row; partial[b] = the best of its rows, for tile_count_kernel's final reduction.
constexpr uint32_t kRowGrid = 64;
// COH: the table's cells were changed by atomics of THIS launch from other XCDs (the fused form of round 6: the row maxima behind the merge in the merge's own launch):
// they are read by loads that do not stop at this XCD's L2.
template <bool COH> __device__ __forceinline__ unsigned long long table_cell(const uint64_t *p)
{
    if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}
// workgroup `wg` of `n_wg`, NT threads: rows wg, wg + n_wg, ..., a lane a row; partial[wg] = the best of its rows
// `meet` (the fused form): called once, by every thread, between the loads of the rows' kept maxima (written by an earlier LAUNCH) and the first look at the table -- the
// wait for the other workgroups' count deltas, with the kept maxima's round trip under it.
struct NoMeet { __device__ __forceinline__ void operator()() const {} };
template <int NT, bool COH, class Meet = NoMeet>
__device__ __forceinline__ void rowmax_body(const TrainArgs &A, uint32_t merge_index, uint32_t wg, uint32_t n_wg, unsigned long long *s_cnt, uint32_t *s_key, uint32_t *s_rows, uint32_t *s_n,
                                            Meet meet = Meet())
{
    bool met = false;
    const uint32_t v_cur = 256u + merge_index;            // ids that can exist so far: rows and columns below v_cur
    const uint32_t newest = merge_index > 0 ? v_cur - 1u : kEmpty;    // the id the previous merge made: counts grew only in its row and its column
    unsigned long long my_best = 0;
    uint32_t my_key = 0;
    for (uint32_t row0 = wg; row0 < v_cur; row0 += n_wg * NT) {       // (one trip up to 16 384 ids)
        if (threadIdx.x == 0) *s_n = 0;
        const uint32_t row = row0 + threadIdx.x * n_wg;                                   // a lane a row: its kept maximum, or its name on the list of rows to read again
        unsigned long long oc = 0;
        uint32_t ok = 0, dirty = 0;
        if (row < v_cur) { oc = A.row_cnt[row]; ok = A.row_key[row]; dirty = A.row_dirty[row]; }
        if (!met) { meet(); met = true; }
        __syncthreads();                                                                  // (s_n; the fused form: every lane's share of the meeting is over)
        if (row < v_cur) {
            // Within one merge a cell only shrinks or only grows.  A shrinking cell lowers the row's maximum iff it IS the maximum: the kept cell no longer holds the kept
            // count.  A growing cell's pair holds the merge's new id: it lies in the newest id's row (all new: read) or column (one cell a row, looked at here).
            bool again = dirty || row == newest;                              // (row_dirty: not read yet since the table was built)
            // the kept cell and the row's cell in the newest id's column, asked for together (one after the other they were two memory round trips of a 7 us kernel)
            const uint32_t cell = row * A.V + (newest != kEmpty ? newest : 0u);
            const unsigned long long kept_now = table_cell<COH>(A.table + (oc != 0 ? ~ok : cell));
            const unsigned long long c = table_cell<COH>(A.table + cell);
            if (!again && oc != 0 && kept_now != oc) again = true;
            if (again) s_rows[atomicAdd(s_n, 1u)] = row;
            else {
                if (newest != kEmpty && better(c, ~cell, oc, ok)) { oc = c; ok = ~cell; A.row_cnt[row] = oc; A.row_key[row] = ok; }
                if (better(oc, ok, my_best, my_key)) { my_best = oc; my_key = ok; }
            }
        }
        __syncthreads();
        const uint32_t n_again = *s_n;
        for (uint32_t q = 0; q < n_again; ++q) {                                         // (a handful per merge over the whole grid)
            const uint32_t again = s_rows[q];
            unsigned long long best = 0;
            uint32_t bkey = 0;
            const uint64_t *cells = A.table + (size_t)again * A.V;
            // sixteen cells a lane in flight (four at a time a row of 4 256 counts was five memory round trips, one after the other)
            for (uint32_t c0 = threadIdx.x; c0 < v_cur; c0 += 16 * NT) {
                unsigned long long cnt[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) { const uint32_t c = c0 + (uint32_t)u * NT; cnt[u] = c < v_cur ? table_cell<COH>(cells + c) : 0ull; }
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const uint32_t k = ~(again * A.V + c0 + (uint32_t)u * NT);
                    if (better(cnt[u], k, best, bkey)) { best = cnt[u]; bkey = k; }
                }
            }
            for (int d = 32; d > 0; d >>= 1) {
                const unsigned long long oc = __shfl_down(best, d, 64);
                const uint32_t ok = __shfl_down(bkey, d, 64);
                if (better(oc, ok, best, bkey)) { best = oc; bkey = ok; }
            }
            if ((threadIdx.x & 63) == 0) { s_cnt[threadIdx.x >> 6] = best; s_key[threadIdx.x >> 6] = bkey; }
            __syncthreads();
            if (threadIdx.x == 0) {
                for (int w = 1; w < NT / 64; ++w) if (better(s_cnt[w], s_key[w], best, bkey)) { best = s_cnt[w]; bkey = s_key[w]; }
                A.row_cnt[again] = best; A.row_key[again] = bkey; A.row_dirty[again] = 0u;
                if (better(best, bkey, my_best, my_key)) { my_best = best; my_key = bkey; }
            }
            __syncthreads();
        }
    }
    for (int d = 32; d > 0; d >>= 1) {
        const unsigned long long oc = __shfl_down(my_best, d, 64);
        const uint32_t ok = __shfl_down(my_key, d, 64);
        if (better(oc, ok, my_best, my_key)) { my_best = oc; my_key = ok; }
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { s_cnt[threadIdx.x >> 6] = my_best; s_key[threadIdx.x >> 6] = my_key; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < NT / 64; ++w) if (better(s_cnt[w], s_key[w], my_best, my_key)) { my_best = s_cnt[w]; my_key = s_key[w]; }
        A.partial[2 * wg] = my_best; A.partial[2 * wg + 1] = my_key;
    }
}
__global__ __launch_bounds__(kThreads) void rowmax_kernel(TrainArgs A, uint32_t merge_index)
{
    __shared__ unsigned long long s_cnt[kThreads / 64];
    __shared__ uint32_t s_key[kThreads / 64];
    __shared__ uint32_t s_rows[kThreads];
    __shared__ uint32_t s_n;
    rowmax_body<kThreads, false>(A, merge_index, blockIdx.x, gridDim.x, s_cnt, s_key, s_rows, &s_n);
}

// One workgroup: commit the previous merge (length, counter), pick the pair of this one.
__global__ __launch_bounds__(kThreads) void argmax_final_kernel(TrainArgs A, uint32_t merge_index, uint32_t n_partials)
{
    __shared__ unsigned long long s_cnt[kThreads / 64];
    __shared__ uint32_t s_key[kThreads / 64];
    unsigned long long best = 0;
    uint32_t bkey = 0;
    for (uint32_t i = threadIdx.x; i < n_partials; i += kThreads) {
        const unsigned long long oc = A.partial[2 * i];
        const uint32_t ok = (uint32_t)A.partial[2 * i + 1];
        if (better(oc, ok, best, bkey)) { best = oc; bkey = ok; }
    }
    for (int d = 32; d > 0; d >>= 1) {
        const unsigned long long oc = __shfl_down(best, d, 64);
        const uint32_t ok = __shfl_down(bkey, d, 64);
        if (better(oc, ok, best, bkey)) { best = oc; bkey = ok; }
    }
    if ((threadIdx.x & 63) == 0) { s_cnt[threadIdx.x >> 6] = best; s_key[threadIdx.x >> 6] = bkey; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kThreads / 64; ++w) if (better(s_cnt[w], s_key[w], best, bkey)) { best = s_cnt[w]; bkey = s_key[w]; }
        TrainState *st = A.st;
        if (st->active && merge_index > 0) { st->n_cur = st->n_next; }   // commit merge_index - 1
        if (st->active && best != 0) {
            const uint32_t idx = ~bkey;
            st->left = idx / A.V;
            st->right = idx % A.V;
            st->new_id = 256u + merge_index;                                // lib.rs:97
            A.pairs_out[2 * merge_index] = st->left;
            A.pairs_out[2 * merge_index + 1] = st->right;
            st->done = merge_index + 1;
        } else {
            st->active = 0;                                                 // lib.rs:88-90: pairs.is_empty() -> break
        }
    }
}

// ---- helpers for the rewrite ------------------------------------------------------------------
// Per thread: kPerThread (8) consecutive ids (+1 before, +2 after).  `lead_par` = parity of the number of
// consecutive `left` ids immediately before the thread's first id (only used when left == right).
struct Span {
    uint32_t a[kPerThread + 3];   // a[0] = id before the span, a[1..kPerThread] the span, then the two after
};

__device__ __forceinline__ uint32_t halo_at(const Halo &h, int64_t i, uint64_t n)   // id at position i outside [0, n)
{
    if (i < 0) return i == -1 ? h.prev : kEmpty;
    const uint64_t d = (uint64_t)i - n;                 // (no indexing by d: an indexed member put the whole struct, 32 bytes a lane, into scratch memory -- written and
    return d == 0 ? h.next[0] : d == 1 ? h.next[1] : d == 2 ? h.next[2] : kEmpty;   //  read back by every lane of every launch: a quarter of the rewrite's HBM writes)
}

// A lane's span out of LDS.  A lane's 19 ids lie 64 bytes apart from its neighbour's: read straight from memory, every one of the 19 loads of a wave touches 64
// separate 64-byte segments (tile_count 40 us and rewrite 64 us per merge whatever the length: 1 TB/s).  So the workgroup brings the tile -- its kTile ids and the three
// around it -- with consecutive lanes on consecutive words, parks it in LDS with one pad word per 16 (lane t's ids start at word 17 t: the lanes' reads hit 64
// different banks), and every lane picks its span from there.  tile_park ends with a barrier; callers put one before the next tile_park (or re-use of s_ids).
constexpr uint32_t kStageWords = kTile + 3 + (kTile + 3) / kPerThread + 1;
// Staging in two halves, so that a workgroup's NEXT tile travels while it works on the current one (a tile staged when it is needed costs its memory latency every time:
// the rewrite took 87 us per merge on a corpus it could stream in 40): tile_fetch requests the tile's words into registers, tile_park writes them to LDS (barrier inside).
// The tile's own words travel sixteen bytes a lane (its start is aligned: buffers are, tiles are kTile words); the id before it and the two after by lanes 0, 1, 2.
static_assert(kPerThread % 4 == 0, "a lane fetches its share of a tile in 16-byte pieces");
constexpr int kFetchVecs = kPerThread / 4;
struct TileRegs { uint4 v[kFetchVecs]; uint32_t h; };
// position in the staged tile (0 = the id before the tile) of word c of a lane's k-th piece, and of the lane's neighbour word
__device__ __forceinline__ uint32_t fetch_pos(int k, int c) { return 1u + ((uint32_t)k * kThreads + threadIdx.x) * 4u + (uint32_t)c; }
__device__ __forceinline__ uint32_t fetch_pos_h() { return threadIdx.x == 0 ? 0u : kTile + threadIdx.x; }
__device__ __forceinline__ void tile_fetch(TileRegs &r, const uint32_t *src, uint64_t base, uint64_t n, const Halo &h)
{
    if (base >= 1 && base + kTile + 2 <= n) {         // (uniform) a tile with its three neighbours inside the buffer -- all but the first and the last one or two: nothing to check
        const uint4 *p = reinterpret_cast<const uint4 *>(src + base) + threadIdx.x;
#pragma unroll
        for (int k = 0; k < kFetchVecs; ++k) r.v[k] = p[(uint32_t)k * kThreads];
        r.h = threadIdx.x < 3 ? src[base + fetch_pos_h() - 1] : 0u;
        return;
    }
    // a boundary tile: word by word, each checked (the compiler branches around and waits for every load -- two or three tiles of a merge)
#pragma unroll
    for (int k = 0; k < kFetchVecs; ++k) {
        uint32_t w[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int64_t i = (int64_t)base + fetch_pos(k, c) - 1;
            w[c] = (i < (int64_t)n) ? src[i] : halo_at(h, i, n);
        }
        r.v[k] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    r.h = 0u;
    if (threadIdx.x < 3) {
        const int64_t i = (int64_t)base + fetch_pos_h() - 1;
        r.h = (i >= 0 && i < (int64_t)n) ? src[i] : halo_at(h, i, n);
    }
}
__device__ __forceinline__ void tile_park(uint32_t *s_ids, const TileRegs &r)
{
#pragma unroll
    for (int k = 0; k < kFetchVecs; ++k) {
        const uint32_t w[4] = {r.v[k].x, r.v[k].y, r.v[k].z, r.v[k].w};
#pragma unroll
        for (int c = 0; c < 4; ++c) { const uint32_t j = fetch_pos(k, c); s_ids[j + j / kPerThread] = w[c]; }
    }
    if (threadIdx.x < 3) { const uint32_t j = fetch_pos_h(); s_ids[j + j / kPerThread] = r.h; }
    lds_barrier();
}
__device__ __forceinline__ void span_from_lds(Span &s, const uint32_t *s_ids)
{
    const uint32_t t17 = threadIdx.x * (uint32_t)(kPerThread + 1);      // one pad word per span: the lanes' reads hit different banks
#pragma unroll
    for (int k = 0; k < kPerThread + 3; ++k) s.a[k] = s_ids[t17 + k + k / kPerThread];
}

// number of trailing `l` ids of the span (0..16)
__device__ __forceinline__ uint32_t trailing_l(const Span &s, uint32_t l)
{
    uint32_t c = 0;
#pragma unroll
    for (int k = kPerThread; k >= 1; --k) { if (s.a[k] == l && c == (uint32_t)(kPerThread - k)) ++c; }
    return c;
}

// Lead parity of each thread inside the workgroup for l == r merges: the trailing-run parity of
// the nearest previous thread whose span is not all `l` (spans are even-sized, so all-`l` spans
// pass parity through); `tile_par` if every previous thread of the tile is all `l`.
// Contains workgroup barriers: call from uniform control flow.
__device__ __forceinline__ uint32_t thread_lead_parity(uint32_t tail, uint32_t tile_par, uint32_t *s_wave, uint32_t *next_tile_par = nullptr)
{
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned long long m = __ballot(tail != kPerThread);
    const int last = m ? 63 - __clzll((long long)m) : 0;
    const uint32_t last_tail = __shfl(tail, last, 64);
    const unsigned long long before = m & ((1ull << lane) - 1ull);
    const int src = before ? 63 - __clzll((long long)before) : 0;
    const uint32_t src_tail = __shfl(tail, src, 64);
    if (lane == 0) s_wave[wv] = m ? (2u | (last_tail & 1u)) : 0u;
    lds_barrier();
    uint32_t par;
    if (before) par = src_tail & 1u;
    else {
        par = tile_par;
        for (int w = (int)wv - 1; w >= 0; --w)
            if (s_wave[w] & 2u) { par = s_wave[w] & 1u; break; }
    }
    if (next_tile_par) {                                     // what the NEXT tile is handed: the last span that is not all `l` decides, else the parity passes through
        uint32_t np = tile_par;
        for (int w = kThreads / 64 - 1; w >= 0; --w)
            if (s_wave[w] & 2u) { np = s_wave[w] & 1u; break; }
        *next_tile_par = np;
    }
    lds_barrier();
    return par;
}

// Classify the span.  Bit k of site_mask: a merge starts at element k; bit k of second_mask:
// element k is consumed as the right half of a site (k = 1..16).  `site_after`: element 17 starts
// a site.  `lead_par` is only used when l == r.
__device__ __forceinline__ void classify(const Span &s, uint32_t l, uint32_t r, uint32_t lead_par,
                                         uint32_t &site_mask, uint32_t &second_mask, uint32_t &site_after)
{
    // (the masks are built in locals, by selects: updated through the reference parameters under conditions they lived in scratch memory -- a store per set bit and
    // loads that wait on the same counter as the tile prefetch and the survivors' stores)
    uint32_t sm = 0, sc = 0;
    if (l != r) {
#pragma unroll
        for (int k = 1; k <= kPerThread; ++k) {
            sm |= ((s.a[k] == l && s.a[k + 1] == r) ? 1u : 0u) << k;
            sc |= ((s.a[k - 1] == l && s.a[k] == r) ? 1u : 0u) << k;
        }
        site_after = (s.a[kPerThread + 1] == l && s.a[kPerThread + 2] == r) ? 1u : 0u;
    } else {
        uint32_t q = lead_par;                                  // run offset parity of element 1
#pragma unroll
        for (int k = 1; k <= kPerThread; ++k) {
            const bool is_l = s.a[k] == l;
            sm |= ((is_l && q == 0 && s.a[k + 1] == l) ? 1u : 0u) << k;
            sc |= ((is_l && q != 0) ? 1u : 0u) << k;
            q = is_l ? (q ^ 1u) : 0u;
        }
        site_after = (s.a[kPerThread + 1] == l && q == 0 && s.a[kPerThread + 2] == l) ? 1u : 0u;
    }
    site_mask = sm; second_mask = sc;
}

// ---- 2. per-tile survivor counts (+ run facts of the tile for l == r merges) -------------------
// flags bit0: the run of `l` that starts the tile has odd length (then an odd entering run costs the
//             tile one more id);  bit1: the whole tile is `l` (even-sized: parity passes through);
//       bit2: parity of the run of `l` that ends the tile (for the next tile's lead parity).
// n_partials > 0 (the one-rank trainer): the final arg-max over argmax_partial_kernel's results is done HERE, by every workgroup for itself (a few hundred pairs), and workgroup 0
// commits it as argmax_final_kernel does -- one launch less per merge (6 us of 110).  Workgroups read nothing that workgroup 0 writes: the length is n_next of the merge being
// committed, the pair is their own reduction, and `active` only ever goes from 1 to 0, which their own reduction finds too.
__global__ __launch_bounds__(kThreads) void tile_count_kernel(TrainArgs A, uint32_t src_sel, uint32_t merge_index = 0, uint32_t n_partials = 0)
{
    __shared__ uint32_t s_wave[kThreads / 64];
    __shared__ uint32_t s_cnt[kThreads / 64];
    __shared__ uint32_t s_first[kThreads / 64];
    __shared__ uint32_t s_last[kThreads / 64];
    __shared__ uint32_t s_lead_extra;
    __shared__ uint32_t s_ids[kStageWords];
    __shared__ unsigned long long s_best[kThreads / 64];
    __shared__ uint32_t s_bkey[kThreads / 64];
    TrainState st = *A.st;
    if (!st.active) return;
    // The first tiles of the workgroup's range are requested before anything else: which ids are compared needs the pair, the ids themselves only the length (a range
    // at the end of a run is three or four tiles: asked for one after the other behind the arg-max, they were four memory round trips of a 10 us kernel).
    constexpr int kAhead = 4;                                                      // tiles requested together: a lane has 8 x 16 bytes in flight
    constexpr int kPreVecs = kAhead * kFetchVecs;
    const uint32_t *const src = A.buf[src_sel];
    const uint64_t n = (n_partials > 0 && merge_index > 0) ? st.n_next : st.n_cur;   // (the length merge_index - 1 left; the step-wise form has committed it already)
    const uint32_t n_tiles = (uint32_t)((n + kTile - 1) / kTile);
    uint32_t t_first, t_count;
    tile_range(n_tiles, blockIdx.x, gridDim.x, t_first, t_count);
    uint4 pre_v[kPreVecs];
    uint32_t pre_b[kPreVecs];
    uint32_t pre_n = 0;                                                            // leading tiles of the range that are whole and have an id before them
#pragma unroll
    for (int j = 0; j < kAhead; ++j) {
        const uint64_t base = (uint64_t)(t_first + j) * kTile;
        if (pre_n == (uint32_t)j && (uint32_t)j < t_count && base >= 1 && base + kTile <= n) pre_n = j + 1;
    }
#pragma unroll
    for (int j = 0; j < kAhead; ++j) {
        if ((uint32_t)j < pre_n) {                                                 // (uniform)
            const uint64_t base = (uint64_t)(t_first + j) * kTile;
            const uint4 *p = reinterpret_cast<const uint4 *>(src + base) + threadIdx.x;
#pragma unroll
            for (int k = 0; k < kFetchVecs; ++k) {
                pre_v[j * kFetchVecs + k] = p[(uint32_t)k * kThreads];
                pre_b[j * kFetchVecs + k] = 0u;
                if ((threadIdx.x & 63) == 0) pre_b[j * kFetchVecs + k] = src[base + ((uint64_t)k * kThreads + threadIdx.x) * 4u - 1u];
            }
        }
    }
    if (n_partials > 0) {
        unsigned long long best = 0;
        uint32_t bkey = 0;
        for (uint32_t i = threadIdx.x; i < n_partials; i += kThreads) {
            const unsigned long long oc = A.partial[2 * i];
            const uint32_t ok = (uint32_t)A.partial[2 * i + 1];
            if (better(oc, ok, best, bkey)) { best = oc; bkey = ok; }
        }
        for (int d = 32; d > 0; d >>= 1) {
            const unsigned long long oc = __shfl_down(best, d, 64);
            const uint32_t ok = __shfl_down(bkey, d, 64);
            if (better(oc, ok, best, bkey)) { best = oc; bkey = ok; }
        }
        if ((threadIdx.x & 63) == 0) { s_best[threadIdx.x >> 6] = best; s_bkey[threadIdx.x >> 6] = bkey; }
        __syncthreads();
        best = s_best[0]; bkey = s_bkey[0];
        for (int w = 1; w < kThreads / 64; ++w) if (better(s_best[w], s_bkey[w], best, bkey)) { best = s_best[w]; bkey = s_bkey[w]; }
        if (merge_index > 0) st.n_cur = st.n_next;                          // (the length merge_index - 1 left)
        const uint32_t idx = ~bkey;
        st.left = idx / A.V; st.right = idx % A.V;
        if (blockIdx.x == 0 && threadIdx.x == 0) {                          // commit, as argmax_final_kernel does
            TrainState *g = A.st;
            if (merge_index > 0) g->n_cur = st.n_cur;
            if (best != 0) {
                g->left = st.left; g->right = st.right;
                g->new_id = 256u + merge_index;                             // lib.rs:97
                A.pairs_out[2 * merge_index] = st.left;
                A.pairs_out[2 * merge_index + 1] = st.right;
                g->done = merge_index + 1;
            } else {
                g->active = 0;                                              // lib.rs:88-90: pairs.is_empty() -> break
            }
        }
        if (best == 0) return;
    }
    const uint32_t l = st.left, r = st.right;
    const bool same = (l == r);
    const Halo alone{kEmpty, {kEmpty, kEmpty, kEmpty}, 0, {0, 0, 0}};                                             // as if the shard stood alone: the rewrite adds what the neighbours change
    TileRegs tr;
    if (!same) {
        // Two different ids (nearly every merge): an id is consumed iff it is `r` and follows `l` -- a fact about two neighbours, whatever lane holds them.  So the
        // range streams through registers as it lies in memory (16 bytes a lane, the word before a lane's piece from the lane below), no LDS, no barrier until the one
        // sum at the end.  Through the staged spans of the l == r path this pass ran at 2.3 TB/s on a corpus in HBM.
        uint32_t dropped = 0;
        const uint32_t lane = threadIdx.x & 63;
        auto pairs_in = [&](const uint4 &v, uint32_t before) {                        // ids consumed among a lane's four: `r` right after `l`
            const uint32_t up = __shfl_up(v.w, 1, 64);
            const uint32_t a0 = lane == 0 ? before : up;
            return (a0 == l && v.x == r ? 1u : 0u) + (v.x == l && v.y == r ? 1u : 0u) + (v.y == l && v.z == r ? 1u : 0u) + (v.z == l && v.w == r ? 1u : 0u);
        };
#pragma unroll
        for (int j = 0; j < kAhead; ++j) {                                             // the tiles requested at the top
            if ((uint32_t)j < pre_n) {
#pragma unroll
                for (int k = 0; k < kFetchVecs; ++k) dropped += pairs_in(pre_v[j * kFetchVecs + k], pre_b[j * kFetchVecs + k]);
            }
        }
        auto whole_tiles = [&](uint64_t base, auto n_const) {                          // (one tile at a time, a trip was one memory latency long: 2.3 TB/s on a corpus in HBM)
            constexpr int NV = decltype(n_const)::value * kFetchVecs;
            const uint4 *p = reinterpret_cast<const uint4 *>(src + base) + threadIdx.x;
            uint4 v[NV];
            uint32_t before[NV];
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                v[k] = p[(uint32_t)k * kThreads];
                before[k] = 0u;
                if (lane == 0) before[k] = src[base + ((uint64_t)k * kThreads + threadIdx.x) * 4u - 1u];
            }
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const uint32_t up = __shfl_up(v[k].w, 1, 64);
                const uint32_t a0 = lane == 0 ? before[k] : up;
                dropped += (a0 == l && v[k].x == r ? 1u : 0u) + (v[k].x == l && v[k].y == r ? 1u : 0u) + (v[k].y == l && v[k].z == r ? 1u : 0u) +
                           (v[k].z == l && v[k].w == r ? 1u : 0u);
            }
        };
        uint32_t t = t_first + pre_n;
        const uint32_t t_end = t_first + t_count;
        for (; t < t_end; ) {
            const uint64_t base = (uint64_t)t * kTile;
            if (t + kAhead <= t_end && base >= 1 && base + (uint64_t)kAhead * kTile <= n) {      // (uniform) whole tiles with an id before them
                whole_tiles(base, std::integral_constant<int, kAhead>{});
                t += kAhead;
            } else if (base >= 1 && base + kTile <= n) {
                whole_tiles(base, std::integral_constant<int, 1>{});
                t += 1;
            } else {                                                                    // the first tile, a partial last one
                for (uint32_t k = threadIdx.x; k < kTile; k += kThreads) {
                    const uint64_t i = base + k;
                    if (i < n && i >= 1 && src[i - 1] == l && src[i] == r) ++dropped;
                }
                t += 1;
            }
        }
        for (int o = 32; o > 0; o >>= 1) dropped += __shfl_down(dropped, o, 64);
        if (lane == 0) s_cnt[threadIdx.x >> 6] = dropped;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t total = 0;
            for (int w = 0; w < kThreads / 64; ++w) total += s_cnt[w];
            const uint64_t lo = min((uint64_t)t_first * kTile, n), hi = min((uint64_t)(t_first + t_count) * kTile, n);
            A.tiles[blockIdx.x].count0 = (uint32_t)(hi - lo) - total;
            A.tiles[blockIdx.x].flags = 0u;
        }
        return;
    }
    if (t_count) tile_fetch(tr, src, (uint64_t)t_first * kTile, n, alone);
    // the range's record, built by thread 0 tile after tile exactly as the rewrite chains the ranges' records (entering parity 0): a tile of nothing but `l`
    // hands the parity on; the first tile that is not decides what an odd entering run costs the range (bit0); the last one decides the parity the range hands on (bit2)
    uint32_t r_count = 0, r_bit0 = 0, r_tp = 0, r_par = 0, r_seen = 0;
    for (uint32_t t = t_first; t < t_first + t_count; ++t) {
        Span s;
        tile_park(s_ids, tr);
        span_from_lds(s, s_ids);
        if (t + 1 < t_first + t_count) tile_fetch(tr, src, (uint64_t)(t + 1) * kTile, n, alone);               // the next tile's words travel under this tile's work
        uint32_t dropped = 0, tail = 0;
        if (!same) {
#pragma unroll
            for (int k = 1; k <= kPerThread; ++k) dropped += (s.a[k - 1] == l && s.a[k] == r) ? 1u : 0u;
        } else {
            tail = trailing_l(s, l);
            const uint32_t par = thread_lead_parity(tail, 0u, s_wave);   // as if the entering run were even
            uint32_t site_mask, second_mask, sa;
            classify(s, l, r, par, site_mask, second_mask, sa);
            dropped = __popc(second_mask);
        }
        uint32_t d = dropped;
        for (int o = 32; o > 0; o >>= 1) d += __shfl_down(d, o, 64);
        if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = d;
        uint32_t flags = 0;
        if (same) {
            const unsigned long long m = __ballot(tail != kPerThread);
            const int last = m ? 63 - __clzll((long long)m) : 0;
            const uint32_t last_tail = __shfl(tail, last, 64);
            if ((threadIdx.x & 63) == 0) {
                s_first[threadIdx.x >> 6] = m ? (uint32_t)__ffsll((long long)m) - 1u : 64u;
                s_last[threadIdx.x >> 6] = m ? (2u | (last_tail & 1u)) : 0u;
            }
            lds_barrier();
            uint32_t first_thread = kThreads;
            for (int w = 0; w < kThreads / 64; ++w)
                if (s_first[w] != 64u) { first_thread = w * 64 + s_first[w]; break; }
            if (threadIdx.x == first_thread) {
                uint32_t c = 0;
#pragma unroll
                for (int k = 1; k <= kPerThread; ++k) { if (s.a[k] == l && c == (uint32_t)(k - 1)) ++c; }
                s_lead_extra = c;                     // leading `l` ids inside the first non-uniform span
            }
            lds_barrier();
            if (threadIdx.x == 0) {
                const bool whole = (first_thread == kThreads);
                uint32_t tp = 0;
                for (int w = kThreads / 64 - 1; w >= 0; --w) if (s_last[w] & 2u) { tp = s_last[w] & 1u; break; }
                flags = (whole ? 2u : (s_lead_extra & 1u)) | (tp << 2);
            }
        } else {
            lds_barrier();
        }
        if (threadIdx.x == 0) {
            uint32_t total = 0;
            for (int w = 0; w < kThreads / 64; ++w) total += s_cnt[w];
            const uint32_t in_tile = (uint32_t)min((uint64_t)kTile, n - (uint64_t)t * kTile);
            uint32_t c = in_tile - total;
            if (same) {
                if (r_par && (flags & 1u)) c -= 1u;                      // odd run entering the tile + odd leading run: one more id is consumed
                if (!(flags & 2u)) {
                    if (!r_seen) { r_bit0 = flags & 1u; r_seen = 1u; }
                    r_par = r_tp = (flags >> 2) & 1u;
                }
            }
            r_count += c;
        }
        lds_barrier();
    }
    if (threadIdx.x == 0) {
        A.tiles[blockIdx.x].count0 = r_count;
        A.tiles[blockIdx.x].flags = same ? (r_seen ? (r_bit0 | (r_tp << 2)) : 2u) : 0u;    // (an empty range: nothing but `l`, vacuously -- the parity passes through)
    }
}

// ---- 4. rewrite + compaction + count deltas ---------------------------------------------------
#ifdef ECGB_TRAIN_TIMING      // dev builds: shader-clock cycles of every workgroup's phases, summed (scripts/dev_trainer_phases.py)
__device__ unsigned long long g_phase[16];
#define PH_MARK(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); if (threadIdx.x == 0) ph[i] += now_ - last_; last_ = now_; } while (0)
#define PH_DECL unsigned long long ph[15] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long last_ = __builtin_readcyclecounter()
#define PH_FLUSH do { if (threadIdx.x == 0) { for (int i_ = 0; i_ < 15; ++i_) atomicAdd(&g_phase[i_], ph[i_]); atomicAdd(&g_phase[15], 1ull); } } while (0)
#else
#define PH_MARK(i) do {} while (0)
#define PH_DECL do {} while (0)
#define PH_FLUSH do {} while (0)
#endif
__global__ __launch_bounds__(kThreads, ECGB_RW_WAVES) void rewrite_kernel(TrainArgs A, uint32_t src_sel)
{
    __shared__ uint32_t s_key[kHashSlots];
    __shared__ int s_val[kHashSlots];
    __shared__ uint32_t s_wave[kThreads / 64];
    __shared__ uint32_t s_wsum[kThreads / 64];
    __shared__ __attribute__((aligned(16))) uint32_t s_ids[kStageWords];   // the tile on its way in (tile_park), then its survivors on their way out
    PH_DECL;
    constexpr int kRec = kGrid / kThreads;
    static_assert(kGrid % kThreads == 0, "a lane takes kGrid / kThreads records");
    uint32_t cnt[kRec], flg[kRec];                        // the count pass's records, a lane's share (requested first: they need nothing the state says)
#pragma unroll
    for (int q = 0; q < kRec; ++q) {
        const uint32_t j = threadIdx.x * kRec + (uint32_t)q;
        cnt[q] = 0u; flg[q] = 2u;                           // (past the grid: nothing, and the parity passes through)
        if (j < gridDim.x) { const TileInfo ti = A.tiles[j]; cnt[q] = ti.count0; flg[q] = ti.flags; }
    }
    const TrainState st = *A.st;
    if (!st.active) return;
    for (uint32_t i = threadIdx.x; i < kHashSlots; i += kThreads) { s_key[i] = kEmpty; s_val[i] = 0; }
    __syncthreads();
    PH_MARK(0);
    const uint32_t *src = A.buf[src_sel];
    uint32_t *dst = A.buf[src_sel ^ 1u];
    const uint64_t n = st.n_cur;
    const uint32_t l = st.left, r = st.right, X = st.new_id, V = A.V;
    const Halo halo = *A.halo;
    const SlabKey SK{l, r, X, V};
    const uint32_t n_tiles = (uint32_t)((n + kTile - 1) / kTile);
    TileRegs tr;
    uint32_t t_first, t_count;
    tile_range(n_tiles, blockIdx.x, gridDim.x, t_first, t_count);
    if (t_count) tile_fetch(tr, src, (uint64_t)t_first * kTile, n, halo);
    // The range's output offset and the run parity entering it: every workgroup chains the (at most kGrid) records of the count pass for itself, under its first tile's
    // flight -- a scan kernel between the two passes was a launch of 5.5 us per merge for 16 KB of records.  A lane takes kGrid / kThreads consecutive records.  For
    // l == r a record's lead parity is the trailing-run parity of the nearest earlier record that is not all `l` (else what the left neighbour hands over), and an odd
    // entering run costs a record with an odd leading run one more id: thread_lead_parity's scheme one level up.  From here on the workgroup carries both itself.
    uint64_t run_off;
    uint32_t run_par;
    {
        __shared__ unsigned long long s_below[kThreads / 64];
        __shared__ unsigned long long s_own;
        __shared__ uint32_t s_own_par;
        const uint32_t G = gridDim.x, b = blockIdx.x;
        // a shard whose left neighbour ends in `left` while it starts with `right`: its first id is the second half of a site there
        if (threadIdx.x == 0 && l != r && n > 0 && halo.prev == l && src[0] == r) cnt[0] -= 1u;
        uint32_t lead[kRec];
#pragma unroll
        for (int q = 0; q < kRec; ++q) lead[q] = 0u;
        if (l == r) {                                               // (uniform)
            uint32_t tail = kPerThread;                             // as thread_lead_parity reads a span: kPerThread = all `l`, else the parity in bit 0
#pragma unroll
            for (int q = 0; q < kRec; ++q) if (!(flg[q] & 2u)) tail = (flg[q] >> 2) & 1u;
            uint32_t par = thread_lead_parity(tail, halo.lead_par, s_wave);
#pragma unroll
            for (int q = 0; q < kRec; ++q) {
                lead[q] = par;
                if (par && (flg[q] & 1u)) cnt[q] -= 1u;
                if (!(flg[q] & 2u)) par = (flg[q] >> 2) & 1u;
            }
        }
        unsigned long long below = 0;
#pragma unroll
        for (int q = 0; q < kRec; ++q) {
            const uint32_t j = threadIdx.x * kRec + (uint32_t)q;
            if (j < b) below += cnt[q];
            if (j == b) { s_own = cnt[q]; s_own_par = lead[q]; }
        }
        for (int d = 32; d > 0; d >>= 1) below += __shfl_down(below, d, 64);
        if ((threadIdx.x & 63) == 0) s_below[threadIdx.x >> 6] = below;
        __syncthreads();
        run_off = 0;
        for (int w = 0; w < kThreads / 64; ++w) run_off += s_below[w];
        run_par = s_own_par;
        if (b == G - 1 && threadIdx.x == 0) A.st->n_next = run_off + s_own;      // the length this merge leaves (nobody reads it before the next launch)
    }
    PH_MARK(9);
    // A tile's survivors leave one trip LATE, after the next tile has been parked: the counter of memory operations in flight retires in order, and the wait for the
    // next tile's words (requested a whole trip earlier) must find nothing younger than them in flight -- stores issued just before it would have to drain first
    // (the compiler cannot wait past them either: each store sits under its own lane mask, and it counts a skipped branch as no store).  So a trip takes the previous
    // tile's survivors from LDS into registers, parks its own tile, and only then sends them out; the last tile's follow the loop.
    // The merged pair's own count falls by one per site: as a count delta every lane of a wave with a site hit the SAME slot of the LDS table, one after the other (the
    // early merges, a site every few ids).  Each lane counts its sites' (l, r) pairs in a register instead; one delta per wave leaves before the final flush.
    const uint32_t key_lr = l * V + r;
    int n_lr = 0;
    uint32_t kept_prev = 0;
    uint64_t out_prev = 0;
    // Sixteen bytes a lane (four survivors in a row): a wave's store is 1 KB of consecutive words wherever the tile's output begins.  Four bytes a lane was 256 bytes
    // per store at an arbitrary word offset -- partly written 32- and 64-byte pieces at both ends of every one: WRITE_SIZE read 73 MB per merge for 46 MB of survivors
    // (profiles/r04/trainer_hbm.json).
    uint4 sv[kFetchVecs];
    auto survivors_take = [&]() {
#pragma unroll
        for (int k = 0; k < kFetchVecs; ++k) sv[k] = *reinterpret_cast<const uint4 *>(s_ids + (threadIdx.x + (uint32_t)k * kThreads) * 4u);
        lds_barrier();                                       // (everybody has its survivors in registers: s_ids is free for the next tile)
    };
    auto survivors_send = [&]() {
#pragma unroll
        for (int k = 0; k < kFetchVecs; ++k) {
            const uint32_t i = (threadIdx.x + (uint32_t)k * kThreads) * 4u;
            uint32_t *q = dst + out_prev + i;
            if (i + 3u < kept_prev) {
                typedef uint32_t u4_a4 __attribute__((ext_vector_type(4), aligned(4)));      // (one 16-byte store: the address is word-aligned only)
                *reinterpret_cast<u4_a4 *>(q) = u4_a4{sv[k].x, sv[k].y, sv[k].z, sv[k].w};
            } else {
                if (i < kept_prev) q[0] = sv[k].x;
                if (i + 1u < kept_prev) q[1] = sv[k].y;
                if (i + 2u < kept_prev) q[2] = sv[k].z;
            }
        }
    };
    for (uint32_t t = t_first; t < t_first + t_count; ++t) {
        survivors_take();
        PH_MARK(5);
        const uint64_t i0 = (uint64_t)t * kTile + threadIdx.x * kPerThread;
        const uint32_t tile_par = run_par;
        const uint64_t tile_off = run_off;
        Span s;
        tile_park(s_ids, tr);
        PH_MARK(1);
        survivors_send();
        PH_MARK(6);
        span_from_lds(s, s_ids);
        if (t + 1 < t_first + t_count) tile_fetch(tr, src, (uint64_t)(t + 1) * kTile, n, halo);                 // the next tile's words travel under this tile's work
        uint32_t lead_par = 0;
        if (l == r) lead_par = thread_lead_parity(trailing_l(s, l), tile_par, s_wave, &run_par);
        uint32_t site_mask, second_mask, sa;
        classify(s, l, r, lead_par, site_mask, second_mask, sa);
        // Is the element BEFORE the span (a[0]) a site?  For l != r directly; for l == r, a[0] (if it
        // is `l`) is the last id of the run counted by lead_par, so its run offset parity is
        // lead_par ^ 1: a site iff that is even and a[1] == l.
        uint32_t site0;
        if (l != r) site0 = (s.a[0] == l && s.a[1] == r) ? 1u : 0u;
        else site0 = (s.a[0] == l && lead_par == 1u && s.a[1] == l) ? 1u : 0u;
        site_mask |= site0;                 // bit 0 = element before the span
        site_mask |= sa << (kPerThread + 1);
        // local survivors
        const uint32_t valid = (i0 < n) ? (uint32_t)min((uint64_t)kPerThread, n - i0) : 0u;
        const uint32_t kept = __popc((((1u << valid) - 1u) << 1) & ~second_mask);
        uint32_t incl = kept;
        const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d, 64); if (lane >= (uint32_t)d) incl += o; }
        if (lane == 63) s_wsum[wv] = incl;
        PH_MARK(2);
        lds_barrier();                                     // (every lane has its span in registers: s_ids is free for the survivors)
        PH_MARK(3);
        uint32_t w_off = 0, tile_kept = 0;
        for (uint32_t w = 0; w < kThreads / 64; ++w) { if (w < wv) w_off += s_wsum[w]; tile_kept += s_wsum[w]; }
        uint32_t o = w_off + incl - kept;                    // inside the tile's output; it leaves LDS in one coalesced copy below
        // walk the span: write survivors, emit count deltas.  A span with no site in or next to it (site_mask covers elements 0..17) keeps all its ids and changes no
        // count -- nearly every span once the merged pairs are rare, which is most merges: kPerThread stores instead of the walk
        if ((site_mask | second_mask) == 0 && valid == (uint32_t)kPerThread) {
#pragma unroll
            for (int k = 1; k <= kPerThread; ++k) s_ids[o + (uint32_t)(k - 1)] = s.a[k];
        } else
#pragma unroll
        for (int k = 1; k <= kPerThread; ++k) {
            if ((uint32_t)k > valid) continue;
            const bool is_site = (site_mask >> k) & 1u;
            const bool is_second = (second_mask >> k) & 1u;
            const bool site_prev = (site_mask >> (k - 1)) & 1u;
            const bool site_next = (site_mask >> (k + 1)) & 1u;
            // old pair (a[k], a[k+1]) disappears iff a site starts at k-1, k or k+1
            if (s.a[k + 1] != kEmpty && (site_prev || is_site || site_next)) {
                const uint32_t key = s.a[k] * V + s.a[k + 1];
                if (key == key_lr) ++n_lr;                          // (the merged pair itself: every site's, counted in a register -- see n_lr)
                else delta_add_s(s_key, s_val, tab_of(A), A.slab, SK, key, -1);
            }
            if (!is_second) {
                const uint32_t val = is_site ? X : s.a[k];
                s_ids[o++] = val;
                // new pair (val, next kept value) counts iff this or the next kept element is a merged id
                const int kn = is_site ? k + 2 : k + 1;             // next kept element
                const uint32_t an = (kn <= kPerThread + 2) ? s.a[kn] : kEmpty;
                if (an != kEmpty) {
                    bool next_site;
                    if (kn <= kPerThread + 1) next_site = (site_mask >> kn) & 1u;
                    else {
                        // kn == kPerThread + 2 (only when the span's last element is a site): that element starts a site iff it and
                        // the one after it match; for l == r it sits at an even run offset (the site's two elements before it were even, odd)
                        const uint64_t i19 = i0 + kPerThread + 2;   // the id after a[kPerThread + 2]: src[i0 - 1 + kPerThread + 3]
                        const uint32_t a19 = (i19 < n) ? src[i19] : halo_at(halo, (int64_t)i19, n);
                        next_site = (an == l && a19 == r);
                    }
                    const uint32_t next_val = next_site ? X : an;
                    if (is_site || next_site) delta_add_s(s_key, s_val, tab_of(A), A.slab, SK, val * V + next_val, 1);
                }
            }
        }
        PH_MARK(4);
        kept_prev = tile_kept;
        out_prev = tile_off;
        run_off += tile_kept;
        // the count deltas stay in the workgroup's LDS table across ALL its tiles and leave once, after the loop (a flush per tile was two barriers, a sweep of the 2 048
        // slots and as many global atomics as the tile had distinct pairs -- the same pairs tile after tile, from 1 280 workgroups onto the same few hundred cells of the
        // table: every eighth tile still cost the early merges a quarter of their time); a table that fills up overflows into memory as before
        lds_barrier();                                     // (the survivors are all in s_ids)
        PH_MARK(7);
    }
    survivors_take();
    PH_MARK(10);
    survivors_send();
    PH_MARK(11);
    for (int d = 32; d > 0; d >>= 1) n_lr += __shfl_down(n_lr, d, 64);
    if ((threadIdx.x & 63) == 0 && n_lr) delta_add_s(s_key, s_val, tab_of(A), A.slab, SK, key_lr, -n_lr);
    delta_flush(s_key, s_val, tab_of(A), A.slab, SK);
    PH_MARK(8);
    PH_FLUSH;
}

// =================================================================================================================
// Sharded training: the corpus -- ONE string, tokenizer_utils.py:93, so merges do cross record joins --
// is cut into contiguous slices, one per rank.  Every rank keeps the whole V x V table (identical everywhere: the arg-max needs no
// exchange), rewrites its own slice, and per merge exchanges (1) an 8-word summary of its slice (all-gather: the ids at its ends
// and, for merges of a symbol with itself, how a run of that symbol leaves it) and (2) the count deltas of the merge (all-reduce of
// the 6 V-word slab).  The host only enqueues kernels and collectives; nothing is read back inside the loop.
constexpr int kSummaryWords = 8;     // [0] n  [1..3] first three ids  [4] last id  [5] slice is all `left`  [6] parity of its trailing run of `left`

__global__ __launch_bounds__(kThreads) void shard_convert_kernel(TrainArgs A, const uint8_t *text)
{
    const uint64_t n = A.n0;
    for (uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kThreads) A.buf[0][i] = text[i];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        A.st->n_cur = n; A.st->n_next = n; A.st->active = 1; A.st->done = 0;
        A.st->left = A.st->right = A.st->new_id = kEmpty;
    }
}

// initial histogram of the slice, the pair across its right end included (the LEFT rank of a pair counts it)
__global__ __launch_bounds__(kThreads) void shard_count_kernel(TrainArgs A)
{
    __shared__ uint32_t s_key[kHashSlots];
    __shared__ int s_val[kHashSlots];
    for (uint32_t i = threadIdx.x; i < kHashSlots; i += kThreads) { s_key[i] = kEmpty; s_val[i] = 0; }
    __syncthreads();
    const uint64_t n = A.n0;
    const Halo halo = *A.halo;
    const uint32_t n_tiles = (uint32_t)((n + kTile - 1) / kTile);
    for (uint32_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const uint64_t base = (uint64_t)t * kTile;
        for (uint32_t k = threadIdx.x; k < kTile; k += kThreads) {
            const uint64_t i = base + k;
            if (i < n) {
                const uint32_t b = (i + 1 < n) ? A.buf[0][i + 1] : halo.next[0];
                if (b != kEmpty) delta_add(s_key, s_val, tab_of(A), A.buf[0][i] * A.V + b, 1);
            }
        }
        delta_flush(s_key, s_val, tab_of(A));
    }
}

// One workgroup: the slice's summary for the merge in flight.
__global__ __launch_bounds__(kThreads) void shard_summary_kernel(TrainArgs A, uint32_t src_sel, long long *summary, int with_pair)
{
    __shared__ long long s_last_other[kThreads / 64];
    const TrainState st = *A.st;
    const uint64_t n = st.n_cur;
    const uint32_t *src = A.buf[src_sel];
    const uint32_t l = st.left;
    const bool same = with_pair && st.active && st.left == st.right;
    long long all_l = 0, tail_par = 0;
    if (same && n > 0) {
        // the last id of the slice that is not `left`, looked for from the end a tile's worth at a time (nearly always found in the first)
        __shared__ long long s_found;
        long long found = -1;
        for (uint64_t end = n; end > 0 && found < 0; ) {
            const uint64_t base = end > kTile ? end - kTile : 0;
            long long last_other = -1;
            for (uint64_t i = base + threadIdx.x; i < end; i += kThreads) if (src[i] != l) last_other = (long long)i;
            for (int d = 32; d > 0; d >>= 1) { const long long o = __shfl_down(last_other, d, 64); last_other = o > last_other ? o : last_other; }
            if ((threadIdx.x & 63) == 0) s_last_other[threadIdx.x >> 6] = last_other;
            __syncthreads();
            if (threadIdx.x == 0) {
                for (int w = 1; w < kThreads / 64; ++w) last_other = s_last_other[w] > last_other ? s_last_other[w] : last_other;
                s_found = last_other;
            }
            __syncthreads();
            found = s_found;
            end = base;
        }
        if (found >= 0) tail_par = (long long)((n - 1 - (uint64_t)found) & 1);
        else { tail_par = (long long)(n & 1); all_l = 1; }              // nothing but `left`: the run reaches into the slice before
    }
    if (threadIdx.x == 0) {
        summary[0] = (long long)n;
        for (int k = 0; k < 3; ++k) summary[1 + k] = (uint64_t)k < n ? (long long)src[k] : (long long)kEmpty;
        summary[4] = n > 0 ? (long long)src[n - 1] : (long long)kEmpty;
        summary[5] = all_l;
        summary[6] = tail_par;
        summary[7] = 0;
    }
}

// One thread: this rank's neighbours from the gathered summaries.
__global__ void shard_halo_kernel(TrainArgs A, Halo *halo, const long long *gathered, int rank, int world, int with_pair)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const TrainState st = *A.st;
    Halo h;
    h.prev = kEmpty;
    h.lead_par = 0;
    h.pad[0] = h.pad[1] = h.pad[2] = 0;
    uint32_t par = 0;
    for (int q = 0; q < rank; ++q) {
        const long long *g = gathered + (size_t)q * kSummaryWords;
        if (g[0] == 0) continue;                                        // an empty slice: invisible
        h.prev = (uint32_t)g[4];
        if (g[5]) par ^= (uint32_t)(g[0] & 1);                          // a slice of nothing but `left`: the run goes on through it
        else par = (uint32_t)g[6];
    }
    if (with_pair && st.active && st.left == st.right && h.prev == st.left) h.lead_par = par;
    int got = 0;
    for (int q = rank + 1; q < world && got < 3; ++q) {
        const long long *g = gathered + (size_t)q * kSummaryWords;
        for (int k = 0; k < 3 && got < 3 && k < g[0]; ++k) h.next[got++] = (uint32_t)g[1 + k];
    }
    for (; got < 3; ++got) h.next[got] = kEmpty;
    *halo = h;
}

// table += the all-reduced slab of the merge, slab = 0 for the next one
__global__ __launch_bounds__(kThreads) void slab_apply_kernel(TrainArgs A)
{
    const TrainState st = *A.st;
    const uint32_t V = A.V, l = st.left, r = st.right, x = st.new_id;
    for (uint32_t j = blockIdx.x * kThreads + threadIdx.x; j < 6 * V; j += gridDim.x * kThreads) {
        const long long v = A.slab[j];
        if (v == 0) continue;
        A.slab[j] = 0;
        const uint32_t part = j / V, o = j % V;
        const uint32_t fixed = (part % 3 == 0) ? l : (part % 3 == 1) ? r : x;
        const uint32_t a = part < 3 ? fixed : o, b = part < 3 ? o : fixed;
        if (st.active && fixed != kEmpty) atomicAdd(reinterpret_cast<unsigned long long *>(&A.table[(size_t)a * V + b]), (unsigned long long)v);
    }
}

// ---- final: commit the last merge and hand the ids out -----------------------------------------
__global__ __launch_bounds__(kThreads) void finish_kernel(TrainArgs A, uint32_t num_merges, uint32_t *ids_out,
                                                          uint64_t *n_ids_out, uint32_t *n_done_out)
{
    __shared__ uint64_t s_n;
    __shared__ uint32_t s_sel;
    if (threadIdx.x == 0) {
        const TrainState st = *A.st;
        // n_next of the last performed merge is committed here (no later argmax_final ran)
        const uint64_t n = (st.done > 0 && st.active) ? st.n_next : st.n_cur;
        s_n = n;
        s_sel = st.done & 1u;
        if (blockIdx.x == 0) { *n_ids_out = n; *n_done_out = st.done; }
    }
    __syncthreads();
    const uint32_t *src = A.buf[s_sel];
    for (uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x; i < s_n; i += (uint64_t)gridDim.x * kThreads) ids_out[i] = src[i];
}

#include "train_seg.inc"

int check_hip(hipError_t e, const char *what)
{
    if (e == hipSuccess) return ECGB_OK;
    ecgb::set_error(std::string(what) + ": " + hipGetErrorString(e));
    return ECGB_ERR_HIP;
}

inline size_t align256(size_t x) { return (x + 255) / 256 * 256; }
// positions are 64-bit; tile numbers and a range's survivor count (a record of the count pass) are 32-bit: 2^41 ids over kGrid (>= 1 024) ranges of 2 048-id tiles keep both below 2^32
constexpr size_t kMaxIds = (size_t)1 << 41;
static_assert(kMaxIds / kTile < (1ull << 32) && kMaxIds / kGrid + kTile < (1ull << 32), "32-bit tile numbers and range counts");

// An id buffer: n ids of 32 bits; in the slotted form (train_seg.inc) up to 4 kGrid slots of ceil(n / R) ids rounded up to 16 and 16 more, 16 ids of padding before the first and a tile
// and 16 ids after the last (whole-tile loads at a range's end run into the next slot, the last one's into the padding).
constexpr size_t kMaxRanges = 4 * (size_t)kGrid;                      // (a range a wave)
inline size_t buf_bytes(size_t n) { return (n + 32 * kMaxRanges + 2 * (size_t)kTile + 64) * 4; }
inline uint32_t slot_cap(size_t n, unsigned ranges) { return (uint32_t)(((n + ranges - 1) / ranges + 15) / 16 * 16 + 16); }

// carve the state of one trainer (or one shard) out of `scratch`
TrainArgs layout(void *scratch_dev, size_t n, uint32_t num_merges, Halo **halo_w, RangeSum **sums = nullptr)
{
    const size_t V = 256 + (size_t)num_merges;
    const size_t tiles = kGrid;                                         // (records are per range of tiles: one per workgroup of the grid)
    uint8_t *p = reinterpret_cast<uint8_t *>((reinterpret_cast<uintptr_t>(scratch_dev) + 255) / 256 * 256);
    TrainArgs A;
    A.st = reinterpret_cast<TrainState *>(p); p += align256(sizeof(TrainState));
    A.table = reinterpret_cast<uint64_t *>(p); p += align256(V * V * 8);
    A.buf[0] = reinterpret_cast<uint32_t *>(p); p += align256(buf_bytes(n));
    A.buf[1] = reinterpret_cast<uint32_t *>(p); p += align256(buf_bytes(n));
    A.tiles = reinterpret_cast<TileInfo *>(p); p += align256(tiles * sizeof(TileInfo));
    A.partial = reinterpret_cast<uint64_t *>(p); p += align256(kGrid * 16);
    A.row_cnt = reinterpret_cast<uint64_t *>(p); p += align256(V * 8);
    A.row_key = reinterpret_cast<uint32_t *>(p); p += align256(V * 4);
    A.row_dirty = reinterpret_cast<uint32_t *>(p); p += align256(V * 4);
    *halo_w = reinterpret_cast<Halo *>(p); p += align256(sizeof(Halo));
    A.halo = *halo_w;
    A.slab = reinterpret_cast<long long *>(p); p += align256(6 * V * 8);
    A.V = (uint32_t)V;
    A.pairs_out = reinterpret_cast<uint32_t *>(p); p += align256(2 * (size_t)num_merges * 4 + 8);   // (sharded runs keep the pairs here)
    if (sums) *sums = reinterpret_cast<RangeSum *>(p);
    p += align256(2 * kMaxRanges * sizeof(RangeSum));
    A.n0 = n;
    return A;
}

int g_train_fused = 0;      // the slotted forms: the next merge's row maxima inside the merge's launch (round 6); 0: a launch of their own (ecgb_set_bpe_train_fused)

template <typename IdT>
int train_slotted(TrainArgs A, RangeSum *sums, const uint8_t *text_dev, size_t n, uint32_t num_merges, unsigned grid, uint32_t *n_done_dev, uint32_t *ids_out_dev,
                  uint64_t *n_ids_dev, hipStream_t st)
{
    SegArgs<IdT> S;
    S.A = A;
    S.ranges = std::min<uint32_t>(kSegRanges, grid * (kThreads / 64));      // (a range a wave; the tests' small grids: four ranges a workgroup of the old form)
    S.cap = slot_cap(n, S.ranges);
    S.buf[0] = reinterpret_cast<IdT *>(A.buf[0]) + 16;
    S.buf[1] = reinterpret_cast<IdT *>(A.buf[1]) + 16;
    S.sum[0] = sums;
    S.sum[1] = sums + kMaxRanges;
    hipLaunchKernelGGL(seg_init_kernel<IdT>, dim3(grid), dim3(kThreads), 0, st, S, text_dev);
    const unsigned merge_grid = (S.ranges + kSegWaves - 1) / kSegWaves;
    // round 6: merge i + 1's row maxima in merge i's launch (its first workgroups, behind a meeting on per-workgroup flags) -- every workgroup of the launch must be
    // resident for that: one a CU at most.  ecgb_set_bpe_train_fused(0): the arg-max as a launch of its own between two merges (round 5).
    int dev = 0, n_cu = 0;
    const bool fused = g_train_fused && hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
                       merge_grid <= (unsigned)n_cu;
    if (fused) {
        const int rc = check_hip(hipMemsetAsync(A.tiles, 0, (size_t)merge_grid * sizeof(uint32_t), st), "hipMemsetAsync(flags)");
        if (rc) return rc;
    }
    const uint32_t n_row_wg = fused ? std::min<uint32_t>(kRowGrid, merge_grid) : kRowGrid;
    for (uint32_t i = 0; i < num_merges; ++i) {
        if (!fused || i == 0) hipLaunchKernelGGL(rowmax_kernel, dim3(n_row_wg), dim3(kThreads), 0, st, A, i);
        hipLaunchKernelGGL(seg_merge_kernel<IdT>, dim3(merge_grid), dim3(kSegThreads), 0, st, S, i, n_row_wg, fused ? n_row_wg : 0u);   // (with the final arg-max and the commit)
    }
    hipLaunchKernelGGL(seg_finish_kernel<IdT>, dim3(S.ranges), dim3(kThreads), 0, st, S, ids_out_dev, n_ids_dev, n_done_dev);
    return check_hip(hipGetLastError(), "bpe train launches");
}

}  // namespace

// workgroups of the count and rewrite passes: kGrid, or fewer for tests (ranges of many tiles on small corpora: every chain the passes carry across tiles and ranges).
// TEST HOOK: a process-wide value read once at the start of a run / at shard creation (a shard keeps its own copy) -- not synchronised, not for concurrent callers.
static unsigned g_train_grid = kGrid;
extern "C" int ecgb_set_bpe_train_grid(int workgroups)
{
    if (workgroups < 0 || workgroups > (int)kGrid) { ecgb::set_error("ecgb_set_bpe_train_grid: 0 (default) or 1 .. " + std::to_string(kGrid)); return ECGB_ERR_INVALID; }
    g_train_grid = workgroups == 0 ? kGrid : (unsigned)workgroups;
    return ECGB_OK;
}

// which merge step ecgb_bpe_train_hip runs: 0 = slotted ranges, one pass over the ids per merge (train_seg.inc), 16-bit ids when 256 + num_merges <= 65 536;
// 1 = the same with 32-bit ids whatever the vocabulary; 2 = round 4's count pass + rewrite over a globally compacted buffer (what the sharded form runs).
// Like the grid: a process-wide test and tuning hook read at the start of a run.
static int g_train_form = 0;
extern "C" int ecgb_set_bpe_train_fused(int on) { g_train_fused = on ? 1 : 0; return ECGB_OK; }
extern "C" int ecgb_set_bpe_train_form(int form)
{
    if (form < 0 || form > 2) { ecgb::set_error("ecgb_set_bpe_train_form: 0 (default), 1 or 2"); return ECGB_ERR_INVALID; }
    g_train_form = form;
    return ECGB_OK;
}

extern "C" size_t ecgb_bpe_train_scratch_bytes(size_t n, uint32_t num_merges)
{
    const size_t V = 256 + (size_t)num_merges;
    const size_t tiles = kGrid;                                         // (records are per range of tiles: one per workgroup of the grid)
    return align256(sizeof(TrainState)) + align256(V * V * 8) + 2 * align256(buf_bytes(n)) + align256(tiles * sizeof(TileInfo)) +
           align256(kGrid * 16) + align256(V * 8) + 2 * align256(V * 4) + align256(sizeof(Halo)) + align256(6 * V * 8) + align256(2 * (size_t)num_merges * 4 + 8) +
           align256(2 * kMaxRanges * sizeof(RangeSum)) + 1024;
}

extern "C" int ecgb_bpe_train_hip(const uint8_t *text_dev, size_t n, uint32_t num_merges, uint32_t *pairs_dev,
                                  uint32_t *n_done_dev, uint32_t *ids_out_dev, uint64_t *n_ids_dev,
                                  void *scratch_dev, size_t scratch_bytes, void *stream)
{
    if (!pairs_dev || !n_done_dev || !ids_out_dev || !n_ids_dev || !scratch_dev || (n && !text_dev)) {
        ecgb::set_error("ecgb_bpe_train_hip: NULL argument");
        return ECGB_ERR_INVALID;
    }
    if (n >= kMaxIds || num_merges > 65000u) {
        ecgb::set_error("ecgb_bpe_train_hip: text longer than 2^41 bytes or more than 65000 merges");
        return ECGB_ERR_UNSUPPORTED;
    }
    if (scratch_bytes < ecgb_bpe_train_scratch_bytes(n, num_merges)) {
        ecgb::set_error("ecgb_bpe_train_hip: scratch smaller than ecgb_bpe_train_scratch_bytes()");
        return ECGB_ERR_INVALID;
    }
    hipStream_t st = (hipStream_t)stream;
    const size_t V = 256 + (size_t)num_merges;
    Halo *halo_w;
    RangeSum *sums;
    TrainArgs A = layout(scratch_dev, n, num_merges, &halo_w, &sums);
    A.pairs_out = pairs_dev;
    A.slab = nullptr;                                                    // one rank: count deltas go straight into the table
    int rc = check_hip(hipMemsetAsync(A.table, 0, V * V * 8, st), "hipMemsetAsync(table)");
    if (rc) return rc;
    rc = check_hip(hipMemsetAsync(halo_w, 0xFF, 4 * sizeof(uint32_t), st), "hipMemsetAsync(halo)");   // prev, next[3] = kEmpty: no neighbours
    if (rc) return rc;
    rc = check_hip(hipMemsetAsync(reinterpret_cast<uint8_t *>(halo_w) + 16, 0, sizeof(Halo) - 16, st), "hipMemsetAsync(halo)");
    if (rc) return rc;
    rc = check_hip(hipMemsetAsync(A.row_cnt, 0, reinterpret_cast<uint8_t *>(A.row_dirty) - reinterpret_cast<uint8_t *>(A.row_cnt), st), "hipMemsetAsync(row maxima)");
    if (rc) return rc;
    rc = check_hip(hipMemsetAsync(A.row_dirty, 1, V * 4, st), "hipMemsetAsync(row flags)");      // every row is read when the arg-max first comes to it
    if (rc) return rc;
    const unsigned tile_grid = (unsigned)std::max<size_t>(1, std::min<size_t>(g_train_grid, (n + kTile - 1) / kTile));
    // a range's survivor count is a 32-bit record (TileInfo.count0): the static_assert above covers kGrid ranges, a smaller test grid must keep the bound itself
    if (n / tile_grid + kTile >= (1ull << 32)) { ecgb::set_error("ecgb_bpe_train_hip: " + std::to_string(n) + " ids over " + std::to_string(tile_grid) + " ranges overflow a range's 32-bit count (ecgb_set_bpe_train_grid is a test hook: use the default grid)"); return ECGB_ERR_UNSUPPORTED; }
    if (g_train_form != 2) {
        if (g_train_form == 0 && V <= 65536) return train_slotted<uint16_t>(A, sums, text_dev, n, num_merges, tile_grid, n_done_dev, ids_out_dev, n_ids_dev, st);
        return train_slotted<uint32_t>(A, sums, text_dev, n, num_merges, tile_grid, n_done_dev, ids_out_dev, n_ids_dev, st);
    }
    hipLaunchKernelGGL(init_kernel, dim3(tile_grid), dim3(kThreads), 0, st, A, text_dev);
    for (uint32_t i = 0; i < num_merges; ++i) {
        hipLaunchKernelGGL(rowmax_kernel, dim3(kRowGrid), dim3(kThreads), 0, st, A, i);
        hipLaunchKernelGGL(tile_count_kernel, dim3(tile_grid), dim3(kThreads), 0, st, A, i & 1u, i, kRowGrid);  // (with the final arg-max and the commit of merge i - 1)
        hipLaunchKernelGGL(rewrite_kernel, dim3(tile_grid), dim3(kThreads), 0, st, A, i & 1u);
    }
    hipLaunchKernelGGL(finish_kernel, dim3(tile_grid), dim3(kThreads), 0, st, A, num_merges, ids_out_dev, n_ids_dev,
                       n_done_dev);
    return check_hip(hipGetLastError(), "bpe train launches");
}

// ---- sharded training: host handle + step-wise entry points (include/ecgbyte.h) ---------------------------------------------------
struct ecgb_bpe_shard {
    TrainArgs A;
    Halo *halo_w;
    size_t n;
    uint32_t num_merges;
    unsigned tile_grid;
};

extern "C" ecgb_bpe_shard *ecgb_bpe_shard_create(size_t n_local, uint32_t num_merges, void *scratch_dev, size_t scratch_bytes)
{
    if (!scratch_dev || n_local >= kMaxIds || num_merges > 65000u || scratch_bytes < ecgb_bpe_train_scratch_bytes(n_local, num_merges)) {
        ecgb::set_error("ecgb_bpe_shard_create: bad argument or scratch smaller than ecgb_bpe_train_scratch_bytes()");
        return nullptr;
    }
    {
        const size_t tg = std::max<size_t>(1, std::min<size_t>(g_train_grid, (n_local + kTile - 1) / kTile));
        if (n_local / tg + kTile >= (1ull << 32)) { ecgb::set_error("ecgb_bpe_shard_create: ids per range overflow a range's 32-bit count (ecgb_set_bpe_train_grid is a test hook: use the default grid)"); return nullptr; }
    }
    ecgb_bpe_shard *h = new ecgb_bpe_shard;
    h->A = layout(scratch_dev, n_local, num_merges, &h->halo_w);
    h->n = n_local;
    h->num_merges = num_merges;
    h->tile_grid = (unsigned)std::max<size_t>(1, std::min<size_t>(g_train_grid, (n_local + kTile - 1) / kTile));
    return h;
}

extern "C" void ecgb_bpe_shard_destroy(ecgb_bpe_shard *h) { delete h; }

extern "C" void *ecgb_bpe_shard_table(ecgb_bpe_shard *h, size_t *n_words)
{
    if (n_words) *n_words = (size_t)h->A.V * h->A.V;
    return h->A.table;
}

extern "C" void *ecgb_bpe_shard_slab(ecgb_bpe_shard *h, size_t *n_words)
{
    if (n_words) *n_words = 6 * (size_t)h->A.V;
    return h->A.slab;
}

extern "C" int ecgb_bpe_shard_begin(ecgb_bpe_shard *h, const uint8_t *text_dev, long long *summary_dev, void *stream)
{
    if (!h || !summary_dev || (h->n && !text_dev)) { ecgb::set_error("ecgb_bpe_shard_begin: NULL argument"); return ECGB_ERR_INVALID; }
    hipStream_t st = (hipStream_t)stream;
    const size_t V = h->A.V;
    int rc = check_hip(hipMemsetAsync(h->A.table, 0, V * V * 8, st), "hipMemsetAsync(table)");
    if (rc) return rc;
    rc = check_hip(hipMemsetAsync(h->A.slab, 0, 6 * V * 8, st), "hipMemsetAsync(slab)");
    if (rc) return rc;
    hipLaunchKernelGGL(shard_convert_kernel, dim3(h->tile_grid), dim3(kThreads), 0, st, h->A, text_dev);
    hipLaunchKernelGGL(shard_summary_kernel, dim3(1), dim3(kThreads), 0, st, h->A, 0u, summary_dev, 0);
    return check_hip(hipGetLastError(), "bpe shard begin");
}

extern "C" int ecgb_bpe_shard_count(ecgb_bpe_shard *h, const long long *gathered_dev, int rank, int world, void *stream)
{
    if (!h || !gathered_dev || rank < 0 || rank >= world) { ecgb::set_error("ecgb_bpe_shard_count: bad argument"); return ECGB_ERR_INVALID; }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(shard_halo_kernel, dim3(1), dim3(64), 0, st, h->A, h->halo_w, gathered_dev, rank, world, 0);
    hipLaunchKernelGGL(shard_count_kernel, dim3(h->tile_grid), dim3(kThreads), 0, st, h->A);
    return check_hip(hipGetLastError(), "bpe shard count");
}

extern "C" int ecgb_bpe_shard_pick(ecgb_bpe_shard *h, uint32_t merge_index, long long *summary_dev, void *stream)
{
    if (!h || !summary_dev || merge_index >= h->num_merges) { ecgb::set_error("ecgb_bpe_shard_pick: bad argument"); return ECGB_ERR_INVALID; }
    hipStream_t st = (hipStream_t)stream;
    const size_t live = (size_t)(256 + merge_index) * h->A.V;
    const unsigned ag = (unsigned)std::max<size_t>(1, std::min<size_t>(kGrid, (live + kThreads * 8 - 1) / (kThreads * 8)));
    hipLaunchKernelGGL(argmax_partial_kernel, dim3(ag), dim3(kThreads), 0, st, h->A, merge_index);
    hipLaunchKernelGGL(argmax_final_kernel, dim3(1), dim3(kThreads), 0, st, h->A, merge_index, ag);
    hipLaunchKernelGGL(tile_count_kernel, dim3(h->tile_grid), dim3(kThreads), 0, st, h->A, merge_index & 1u, 0u, 0u);
    hipLaunchKernelGGL(shard_summary_kernel, dim3(1), dim3(kThreads), 0, st, h->A, merge_index & 1u, summary_dev, 1);
    return check_hip(hipGetLastError(), "bpe shard pick");
}

extern "C" int ecgb_bpe_shard_merge(ecgb_bpe_shard *h, uint32_t merge_index, const long long *gathered_dev, int rank, int world, void *stream)
{
    if (!h || !gathered_dev || rank < 0 || rank >= world || merge_index >= h->num_merges) { ecgb::set_error("ecgb_bpe_shard_merge: bad argument"); return ECGB_ERR_INVALID; }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(shard_halo_kernel, dim3(1), dim3(64), 0, st, h->A, h->halo_w, gathered_dev, rank, world, 1);
    hipLaunchKernelGGL(rewrite_kernel, dim3(h->tile_grid), dim3(kThreads), 0, st, h->A, merge_index & 1u);
    return check_hip(hipGetLastError(), "bpe shard merge");
}

extern "C" int ecgb_bpe_shard_apply(ecgb_bpe_shard *h, void *stream)
{
    if (!h) { ecgb::set_error("ecgb_bpe_shard_apply: NULL handle"); return ECGB_ERR_INVALID; }
    hipLaunchKernelGGL(slab_apply_kernel, dim3((6 * h->A.V + kThreads - 1) / kThreads), dim3(kThreads), 0, (hipStream_t)stream, h->A);
    return check_hip(hipGetLastError(), "bpe shard apply");
}

extern "C" int ecgb_bpe_shard_finish(ecgb_bpe_shard *h, uint32_t *pairs_dev, uint32_t *n_done_dev, uint32_t *ids_out_dev, uint64_t *n_ids_dev, void *stream)
{
    if (!h || !pairs_dev || !n_done_dev || !ids_out_dev || !n_ids_dev) { ecgb::set_error("ecgb_bpe_shard_finish: NULL argument"); return ECGB_ERR_INVALID; }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(finish_kernel, dim3(h->tile_grid), dim3(kThreads), 0, st, h->A, h->num_merges, ids_out_dev, n_ids_dev, n_done_dev);
    int rc = check_hip(hipGetLastError(), "bpe shard finish");
    if (rc) return rc;
    if (h->num_merges)
        rc = check_hip(hipMemcpyAsync(pairs_dev, h->A.pairs_out, 2 * (size_t)h->num_merges * 4, hipMemcpyDeviceToDevice, st), "hipMemcpyAsync(pairs)");
    return rc;
}

#ifdef ECGB_TRAIN_TIMING
extern "C" int ecgb_dev_train_phases(unsigned long long *out16)
{
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_phase), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    unsigned long long zero[16] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_phase), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}
#endif
