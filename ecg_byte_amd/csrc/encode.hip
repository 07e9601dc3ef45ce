// encode.hip -- quantise + greedy longest-match BPE encode on MI355X (gfx950).
//
// Reference behaviour reproduced bit-exactly (paths relative to the reference root):
//   normalize_all                 ecg_byte/utils/tokenizer_utils.py:14-19
//   rust_bpe.encode_text          ecg_byte/rust_bpe/src/lib.rs:149-193
//   per-sample front end          ecg_byte/data_loader.py:74-76
//
// ---- Quantiser ------------------------------------------------------------------------
// The reference maps x -> min(floor(clip((x-a)/d,0,1)*26),25) in float64.  Every step is a
// monotone non-decreasing function of x (IEEE rounding is monotone, d > 0), so the whole map
// is a 26-level staircase.  The host finds the 25 exact step positions by bisection over the
// float64 bit patterns USING the reference's own operation sequence (quantize_ref below), and
// the kernel classifies x by comparing against them: no fp64 division on the device, identical
// symbols for every input.  (Degenerate parameters, d <= 0 or non-finite, and callers that want
// the clipped float64 output take the literal-division kernel instead.)
//
// ---- Encoder ---------------------------------------------------------------------------
// Greedy longest match is a sequential chain i -> i + len(i).  Walking the trie from EVERY
// position would cost ~21 lookups per symbol on ECG streams; following only the chain costs
// ~1.15 -- and about 0.5 once a run of equal symbols is one step along a same-class chain of
// consecutive trie nodes (tokenizer.hpp has the node layout, walk_chunk / encode_flow_kernel the
// step).  One persistent workgroup per CU keeps the trie (8 B/node) and its bit tables in LDS:
//   stage   lanes read the float64 samples (coalesced 16-byte loads), classify them against the
//           staircase and write the symbol classes straight into LDS -- the symbol stream never
//           exists in HBM (for byte-stream input: LUT classify instead); a change map (bit p =
//           symbol p differs from p-1) gives the length of the run ahead with one find-first-set;
//   parse   speculative: the segment is cut into chunks, one lane each, parsed from the chunk
//           start before the true entry is known;
//   resolve which of the parsed tokens lie on the real chain;
//   emit    prefix sums give every real token its output slot.
// Two kernels share stage / walk: encode_flow_kernel (large batches: one wave = one record,
// claim-merge parse without re-walks, see its header) and encode_wg_kernel (small batches or
// very long tokens: one 256-lane workgroup = one record, 32768-symbol segments, re-parse from
// corrected entries to a fixed point).  No global memory on the dependent path of a step; trie
// nodes that do not fit LDS are read through L2.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <limits>
#include <string>
#include <type_traits>

#include "tokenizer.hpp"

namespace {

// ------------------------------------------------------------------------------------------
// Reference arithmetic of normalize_all, one operation per line (compiled with
// -ffp-contract=off so nothing is fused).  Returns the alphabet index; NaN -> 0.
inline int quantize_ref(double x, double a, double d)
{
    double nrm = (x - a) / d;
    double c = nrm;
    if (c < 0.0) c = 0.0;
    if (c > 1.0) c = 1.0;
    double s = std::floor(c * 26.0);
    if (s > 25.0) s = 25.0;
    return (s == s) ? (int)s : 0;
}

inline uint64_t ordered_key(double x)
{
    uint64_t u;
    std::memcpy(&u, &x, 8);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

inline double from_ordered_key(uint64_t k)
{
    uint64_t u = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    double x;
    std::memcpy(&x, &u, 8);
    return x;
}

struct QuantParams {
    double a;        // p1 - 0.5
    double d;        // ((p99 + 0.5) - (p1 - 0.5)) + 1e-6
    double scale;    // 26 / d, for the first guess of the level
    double nas;      // -a * scale: q = fma(x, scale, nas) in the staging loops of the encode kernels
    float amb_thr;   // a sample is ambiguous when |frac(q) - 0.5| exceeds this (staging loops of the encode kernels)
    double thr[28];  // thr[k] = smallest x whose level is >= k (k = 1..25); thr[0] = -inf, thr[26..27] = +inf
    int use_thresholds;
};

QuantParams make_quant_params(double p1, double p99)
{
    QuantParams q;
    q.a = p1 - 0.5;
    q.d = ((p99 + 0.5) - (p1 - 0.5)) + 1e-6;
    q.scale = 26.0 / q.d;
    q.nas = -(q.a * q.scale);
    const double inf = std::numeric_limits<double>::infinity();
    q.use_thresholds = (q.d > 0.0) && std::isfinite(q.d) && std::isfinite(q.a) && std::isfinite(q.scale);
    // beyond |a * scale| = 1e9 the rounding of fma(x, scale, nas) nears the 1e-4 band of level_f32: every group is then
    // declared ambiguous and takes the exact staircase (slow, exact)
    q.amb_thr = (std::fabs(q.a * q.scale) < 1e9) ? 0.5f - 1e-4f : -1.0f;
    q.thr[0] = -inf;
    q.thr[26] = q.thr[27] = inf;
    for (int k = 1; k <= 25; ++k) q.thr[k] = inf;
    if (!q.use_thresholds) return q;
    const uint64_t lo_all = ordered_key(-inf), hi_all = ordered_key(inf);
    for (int k = 1; k <= 25; ++k) {
        // smallest key in [lo_all, hi_all] with level >= k; level(+inf) = 25 so it exists
        uint64_t lo = lo_all, hi = hi_all;
        while (lo < hi) {
            uint64_t mid = lo + (hi - lo) / 2;
            if (quantize_ref(from_ordered_key(mid), q.a, q.d) >= k) hi = mid; else lo = mid + 1;
        }
        q.thr[k] = from_ordered_key(lo);
    }
    return q;
}

// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t level_from_thresholds(double x, double a, double scale,
                                                          const double *thr /* LDS, 28 entries */)
{
    double q = (x - a) * scale;
    int b = (q >= 0.0) ? ((q < 26.0) ? (int)q : 25) : 0;  // NaN -> 0
    while (x < thr[b]) --b;                                // thr[0] = -inf stops it
    while (b < 25 && x >= thr[b + 1]) ++b;
    return (uint32_t)b;
}

// Level of x without table lookups in the common case.  q = (x-a)*(26/d) differs from the value
// the reference floors, fl(clip(fl(fl(x-a)/d))*26), by a few ulp (< 1e-13 absolute, both round
// the same real number); so unless q lies within 1e-9 of one of the integers 1..25 the two
// floors agree and clamp(floor(q), 0, 25) IS the reference level.  Only the (rare) ambiguous
// inputs consult the exact staircase.  NaN -> 0, +-inf -> 25 / 0, as level_from_thresholds.
__device__ __forceinline__ uint32_t level_fast(double x, double a, double scale, const double *thr)
{
    const double q = (x - a) * scale;
    const double f = floor(q);
    const double t = q - f;
    int b = (q >= 0.0) ? ((q < 26.0) ? (int)f : 25) : 0;
    const bool ambiguous = (t < 1e-9 || t > 1.0 - 1e-9) && (q > 0.5) && (q < 25.5);
    if (ambiguous) b = (int)level_from_thresholds(x, a, scale, thr);
    return (uint32_t)b;
}

// Four levels at once, packed one per byte, for the staging loops of the encode kernels.  q is rounded to float first:
// for 0 <= q < 26 the rounding moves it by < 2e-6, so unless its fraction lies within 1e-4 of an integer, floor(q) is the
// same in float as in double and the argument of level_fast applies; clamping in float (NaN -> 0 by the IEEE max) and the
// truncating convert replace the double-precision floor / compares.  The (rare) group with an ambiguous sample is redone
// from the exact staircase: one branch per four samples.
__device__ __forceinline__ uint32_t level_f32(double x, double scale, double nas, float amb_thr, bool &amb)
{
    const float qf = (float)__builtin_fma(x, scale, nas);   // (x - a) * scale in one operation: |a * scale| < 1e9 keeps it within 1e-7 of it
    const float t = __builtin_amdgcn_fractf(qf);
    amb = amb || (fabsf(t - 0.5f) > amb_thr);               // 0.5 - 1e-4 (make_quant_params)
    return (uint32_t)(int)fminf(fmaxf(qf, 0.0f), 25.5f);
}

__device__ __forceinline__ uint32_t levels4(double2 v0, double2 v1, double a, double scale, double nas, float amb_thr, const double *thr)
{
    bool amb = false;
    uint32_t w = level_f32(v0.x, scale, nas, amb_thr, amb);
    w |= level_f32(v0.y, scale, nas, amb_thr, amb) << 8;
    w |= level_f32(v1.x, scale, nas, amb_thr, amb) << 16;
    w |= level_f32(v1.y, scale, nas, amb_thr, amb) << 24;
    if (amb) {
        w = level_from_thresholds(v0.x, a, scale, thr);
        w |= level_from_thresholds(v0.y, a, scale, thr) << 8;
        w |= level_from_thresholds(v1.x, a, scale, thr) << 16;
        w |= level_from_thresholds(v1.y, a, scale, thr) << 24;
    }
    return w;
}

// Records are rows: record r reads x + r*n and writes sym + r*sym_stride (blockIdx.y strides
// over records, blockIdx.x/threads over the row).  Vector variant: n % 4 == 0 and 16-byte
// aligned x, four samples per thread per iteration, one 4-byte store.
__global__ __launch_bounds__(256) void quantize_thr_kernel(const double *__restrict__ x, size_t n,
                                                           size_t rows, size_t sym_stride,
                                                           QuantParams qp, uint8_t *__restrict__ sym)
{
    __shared__ double thr[28];
    if (threadIdx.x < 28) thr[threadIdx.x] = qp.thr[threadIdx.x];
    __syncthreads();
    const size_t n4 = n / 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t r = blockIdx.y; r < rows; r += gridDim.y) {
        const double *xr = x + r * n;
        uint8_t *sr = sym + r * sym_stride;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
            const double2 v0 = *reinterpret_cast<const double2 *>(xr + 4 * i);
            const double2 v1 = *reinterpret_cast<const double2 *>(xr + 4 * i + 2);
            uint32_t w = level_from_thresholds(v0.x, qp.a, qp.scale, thr);
            w |= level_from_thresholds(v0.y, qp.a, qp.scale, thr) << 8;
            w |= level_from_thresholds(v1.x, qp.a, qp.scale, thr) << 16;
            w |= level_from_thresholds(v1.y, qp.a, qp.scale, thr) << 24;
            *reinterpret_cast<uint32_t *>(sr + 4 * i) = w;
        }
    }
}

// Scalar staircase variant for rows the vector variant cannot take (n % 4 != 0 / unaligned).
__global__ __launch_bounds__(256) void quantize_thr_scalar_kernel(const double *__restrict__ x, size_t n,
                                                                  size_t rows, size_t sym_stride,
                                                                  QuantParams qp, uint8_t *__restrict__ sym)
{
    __shared__ double thr[28];
    if (threadIdx.x < 28) thr[threadIdx.x] = qp.thr[threadIdx.x];
    __syncthreads();
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t r = blockIdx.y; r < rows; r += gridDim.y)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
            sym[r * sym_stride + i] = (uint8_t)level_from_thresholds(x[r * n + i], qp.a, qp.scale, thr);
}

// Literal operation sequence of tokenizer_utils.py:15-17 (IEEE fp64 division on the device).
__global__ __launch_bounds__(256) void quantize_exact_kernel(const double *__restrict__ x, size_t n,
                                                             size_t rows, size_t sym_stride,
                                                             double a, double d,
                                                             uint8_t *__restrict__ sym,
                                                             double *__restrict__ clipped)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t r = blockIdx.y; r < rows; r += gridDim.y)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
            double nrm = (x[r * n + i] - a) / d;
            double c = nrm;
            if (c < 0.0) c = 0.0;
            if (c > 1.0) c = 1.0;
            double s = floor(c * 26.0);
            if (s > 25.0) s = 25.0;
            if (clipped) clipped[r * n + i] = c;
            sym[r * sym_stride + i] = (s == s) ? (uint8_t)s : (uint8_t)0;
        }
}

// ------------------------------------------------------------------------------------------
struct EncodeArgs {
    const uint64_t *trie;     // packed nodes (global copy, all of them)
    uint32_t n_nodes;
    uint32_t n_lds_nodes;     // nodes [0, n_lds_nodes) are staged in LDS
    const uint32_t *runbits;  // per-node continuation / token bit tables (tokenizer.hpp), all of them staged in LDS
    uint32_t n_runwords;      // even
    const uint8_t *tok_len;   // token id -> length (flow kernel)
    uint32_t n_toklen;        // multiple of 8
    uint32_t chunk;           // flow kernel: symbols per lane-chunk; a segment is 64 chunks
    const uint8_t *lut;       // 256 B byte->class | 32 x u16 single_id | 32 B class->byte
    const double *signal;     // batch x n float64 samples            (INPUT_F64)
    const uint8_t *raw;       // batch x n raw bytes                  (INPUT_BYTES)
    QuantParams qp;
    uint16_t *ids_half;       // per resident stream slot: half-resolution id array of one segment
    uint32_t *ids_out;        // batch x ids_stride
    size_t ids_stride;
    uint32_t *counts;         // batch
    uint32_t n;               // symbols per stream
    uint32_t batch;
    uint32_t margin;          // symbols staged past the segment end (>= trie depth), multiple of 16
#ifdef ECGB_PROFILE
    unsigned long long *prof; // per workgroup: 8 words of phase timers / counters (dev builds only)
#endif
};

#ifdef ECGB_PROFILE
unsigned long long *g_prof_dev = nullptr;
#define PROF_STAMP(k) do { if (A.prof && (threadIdx.x & 63) == 0) { atomicAdd(&A.prof[blockIdx.x * 8 + (k)], (unsigned long long)(clock64() - t_prof)); } t_prof = clock64(); } while (0)
#define PROF_COUNT(k, v) do { if (A.prof && (threadIdx.x & 63) == 0) atomicAdd(&A.prof[blockIdx.x * 8 + (k)], (unsigned long long)(v)); } while (0)
#else
#define PROF_STAMP(k) do { } while (0)
#define PROF_COUNT(k, v) do { } while (0)
#endif

constexpr int INPUT_F64 = 0, INPUT_BYTES = 1;
constexpr uint32_t kLenBias = 30;          // flow kernel: a symbol byte >= kLenBias holds kLenBias + the length of the token that starts there
constexpr int kStageGroups = 5;             // groups of 4 samples a lane loads before it classifies the first (C2: 3 batches per segment)
constexpr uint32_t kListRegs = 8;          // flow kernel: token-list entries a lane keeps in registers between its two sweeps
constexpr size_t kFlowSlot = 8192;         // token-list entries (u32: position | id << 16) per resident wave of the flow kernel: one segment

// LDS byte address of word (i >> 5) of a table of 1 << SHIFT bytes per 32 entries: two instructions (the compiler's own
// shift / mask / add sequence for the same expression is three; the parse loop of the flow kernel issues it twice a trip).
typedef __attribute__((address_space(3))) const uint32_t lds_cu32;
template <int SHIFT>
__device__ __forceinline__ lds_cu32 *lds_word32(uint32_t i, uint32_t table_addr)
{
    uint32_t t, a;
    asm("v_lshrrev_b32 %0, 5, %1" : "=v"(t) : "v"(i));
    asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(a) : "v"(t), "n"(SHIFT), "v"(table_addr));
    return reinterpret_cast<lds_cu32 *>(a);
}
// the same for a table every wave shares: its address is a scalar operand
template <int SHIFT>
__device__ __forceinline__ lds_cu32 *lds_word32_uniform(uint32_t i, uint32_t table_addr)
{
    uint32_t t, a;
    asm("v_lshrrev_b32 %0, 5, %1" : "=v"(t) : "v"(i));
    asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(a) : "v"(t), "n"(SHIFT), "s"(table_addr));
    return reinterpret_cast<lds_cu32 *>(a);
}
__device__ __forceinline__ uint32_t lds_addr(const void *p)
{
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}

// clear bits [lo, hi) of a lane-owned run of mark words (segment-relative bit indices)
__device__ __forceinline__ void clear_bits(uint32_t *marks, uint32_t lo, uint32_t hi)
{
    if (lo >= hi) return;
    uint32_t w0 = lo >> 5, w1 = (hi - 1) >> 5;
    for (uint32_t w = w0; w <= w1; ++w) {
        uint32_t m = 0xFFFFFFFFu;
        if (w == w0) m &= 0xFFFFFFFFu << (lo & 31);
        if (w == w1) m &= 0xFFFFFFFFu >> (31 - ((hi - 1) & 31));
        marks[w] &= ~m;
    }
}

// LDS address of symbol position k.  Lanes own consecutive CHUNK-byte chunks, so lanes at equal
// offsets would all read the same LDS bank (CHUNK = 64 or 128 bytes = 16 or 32 banks apart): a
// 32-way conflict on every symbol read.  XOR-ing address bits [6:2] (the bank index) with the
// low five bits of the chunk number spreads 32 consecutive chunks over the 32 banks.  Groups of
// four positions stay contiguous and aligned; the map is a bijection on 2 * CHUNK-byte blocks.
template <int CHUNK>
__device__ __forceinline__ uint32_t swz(uint32_t k)
{
    if constexpr (CHUNK == 0) return k;   // no swizzle: the flow kernel picks a chunk length that spreads over the banks by itself
    else return k ^ (((k / CHUNK) & 31u) << 2);
}

// ---- stage: positions [seg_base, seg_base + stage_len) of one stream -> symbol classes in LDS.
// `lane`/`nlanes`: the cooperating lanes.  Positions >= n get the sentinel class (no trie node
// has that child bit, so a walk can never pass the end of the stream).
template <int CHUNK, int INPUT, bool VEC>
__device__ __forceinline__ void stage_symbols(uint8_t *sym, uint32_t stage_len, uint32_t n_here,
                                              const double *x, const uint8_t *t8, uint32_t lane,
                                              uint32_t nlanes, double qa, double qscale, double qnas, float qamb,
                                              const double *s_thr, const uint8_t *s_b2c)
{
    if (INPUT == INPUT_F64) {
        // whole groups of 4 samples inside the record: vector path, kStageGroups groups (2 x 16-byte loads each) in flight per lane
        // before the first use.  The last, partial batch runs the same code: groups past the end re-read the last
        // group and are not stored (one memory latency per batch of kStageGroups * nlanes groups, none per leftover group).
        const uint32_t vec_len = VEC ? (min(stage_len, n_here) & ~3u) : 0u;
        const uint32_t step = nlanes * 4;
        for (uint32_t k = lane * 4; k < vec_len; k += kStageGroups * step) {
            double2 v[2 * kStageGroups];
#pragma unroll
            for (int u = 0; u < kStageGroups; ++u) {
                const uint32_t ku = min(k + u * step, vec_len - 4);
                v[2 * u] = *reinterpret_cast<const double2 *>(x + ku);
                v[2 * u + 1] = *reinterpret_cast<const double2 *>(x + ku + 2);
            }
#pragma unroll
            for (int u = 0; u < kStageGroups; ++u) {
                const uint32_t w = levels4(v[2 * u], v[2 * u + 1], qa, qscale, qnas, qamb, s_thr);
                if (k + u * step < vec_len) *reinterpret_cast<uint32_t *>(sym + swz<CHUNK>(k + u * step)) = w;
            }
        }
        // the rest (record tail, odd/unaligned rows, sentinel padding) one sample at a time
        for (uint32_t kk = vec_len + lane * 4; kk < stage_len; kk += step) {
            uint32_t w = 0;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const uint32_t lv = (kk + t < n_here) ? level_fast(x[kk + t], qa, qscale, s_thr)
                                                      : ecgb::kOtherClass;
                w |= lv << (8 * t);
            }
            *reinterpret_cast<uint32_t *>(sym + swz<CHUNK>(kk)) = w;
        }
    } else {
        for (uint32_t k = lane; k < stage_len; k += nlanes)
            sym[swz<CHUNK>(k)] = (k < n_here) ? s_b2c[t8[k]] : (uint8_t)ecgb::kOtherClass;
    }
}

// ---- change map: bit k of `dmap` = (class of position k != class of position k - 1), bit 0 set.  A run of
// equal symbols is a run of zero bits, so the walk reads the length of the run ahead of it with one
// find-first-set instead of one trie step per symbol.
template <int CHUNK>
__device__ __forceinline__ void build_dmap(const uint8_t *sym, uint32_t *dmap, uint32_t stage_len, uint32_t lane, uint32_t nlanes)
{
    const uint32_t n_words = (stage_len + 31) >> 5;
    for (uint32_t w = lane; w < n_words; w += nlanes) {
        uint32_t prev = (w == 0) ? 0xFFu : sym[swz<CHUNK>(32 * w - 1)];   // 0xFF is no class: bit 0 of word 0 is set
        uint32_t bits = 0;
#pragma unroll
        for (uint32_t g = 0; g < 8; ++g) {
            const uint32_t v = *reinterpret_cast<const uint32_t *>(sym + swz<CHUNK>(32 * w + 4 * g));
            const uint32_t x = v ^ ((v << 8) | prev);                                  // byte t != 0 iff symbol t differs from t - 1
            const uint32_t nz = ((((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) >> 7) & 0x01010101u;   // bit 8t = byte t non-zero
            bits |= ((nz | (nz >> 7) | (nz >> 14) | (nz >> 21)) & 0xFu) << (4 * g);
            prev = v >> 24;
        }
        dmap[w] = bits;
    }
}

// what a walk needs besides the chunk bounds
struct WalkCtx {
    const uint8_t *sym;       // symbol classes of the segment (swizzled)
    uint32_t *marks;          // token-start bitmap of the segment
    const uint32_t *dmap;     // change map of the segment
    uint16_t *ids_half;
    const uint64_t *s_trie, *g_trie;
    const uint32_t *s_run;    // run bit tables
    uint32_t n_lds;
#ifdef ECGB_PROFILE
    unsigned long long *prof; // this workgroup's counters
#endif
};

// ---- walk: parse one chunk [s_rel, e_rel) (segment-relative) from `start_rel`, merging into
// whatever chain the chunk already holds.  Returns the segment-relative position at which the
// chain leaves the chunk, or `old_exit` when it lands on a position the existing chain marked.
// FIRST = speculative pass over a zeroed bitmap (nothing to clear, nothing to merge into).
//
// ONE flat loop, one trie step OR one token emission per trip, every lane back in step at the
// bottom.  (Written with `continue`, LLVM splits the two back edges into nested loops and a wave
// then waits for its slowest lane on every TOKEN, ~10x the trips; the single latch with a
// convergent no-op keeps it flat.)  Marks are set and cleared with no-return LDS atomics: nothing
// is on the dependent path but the reads at the top of a trip, all issued together.
//
// A step is one of (tokenizer.hpp describes the node layout):
//   branch        the symbol differs from the previous one (or the walk is at the root): child by bitmap + popcount;
//   continuation  the symbol repeats the previous one and the node is the head of its same-class chain: one node;
//   run           it repeats and the node is inside the chain: the chain is consecutive node ids, so the walk takes
//                 m = min(run length ahead, chain length below, 32) symbols at once; the deepest token-carrying node
//                 it passes comes from the token bit table (the node it lands on is examined by the next trip).
template <int CHUNK, bool FIRST, bool ALL_LDS>
__device__ __forceinline__ uint32_t walk_chunk(const WalkCtx &W, uint32_t s_rel, uint32_t e_rel,
                                               uint32_t start_rel, uint32_t old_exit)
{
    const uint8_t *sym = W.sym;
    uint32_t *marks = W.marks;
    if constexpr (!FIRST) {
        if (start_rel >= e_rel) {   // chain jumps over this chunk
            clear_bits(marks, s_rel, e_rel);
            return start_rel;
        }
        clear_bits(marks, s_rel, start_rel);
        if ((marks[start_rel >> 5] >> (start_rel & 31)) & 1u) return old_exit;   // already on the chain
    }
    uint32_t r = start_rel;        // token start
    uint32_t j = r, node = 0, best_j = r;
    uint32_t result = 0;
    bool done = false;
    while (!done) {
#ifdef ECGB_PROFILE
        if (W.prof && (threadIdx.x & 63) == (unsigned)(__ffsll((long long)__ballot(1)) - 1)) atomicAdd(&W.prof[FIRST ? 5 : 6], 1ull);
#endif
        const uint32_t s = sym[swz<CHUNK>(j)];
        uint64_t rec;
        if constexpr (ALL_LDS) rec = W.s_trie[node];
        else rec = (node < W.n_lds) ? W.s_trie[node] : W.g_trie[node];
        const uint32_t dk = j >> 5, rk = (node >> 5) * 2;
        const uint32_t d0 = W.dmap[dk], d1 = W.dmap[dk + 1];
        const uint2 r0 = *reinterpret_cast<const uint2 *>(W.s_run + rk), r1 = *reinterpret_cast<const uint2 *>(W.s_run + rk + 2);
        const uint32_t dw = __builtin_amdgcn_alignbit(d1, d0, j & 31);        // bit t: position j + t starts a new run
        const uint32_t cw = __builtin_amdgcn_alignbit(r1.x, r0.x, node & 31); // bit t: node + t has a continuation
        const uint32_t tw = __builtin_amdgcn_alignbit(r1.y, r0.y, node & 31); // bit t: node + t carries a token itself
        const uint32_t bm = (uint32_t)rec, hi = (uint32_t)(rec >> 32);
        const uint32_t fc = hi & 0xFFFFu;
        if (tw & 1u) best_j = j;                                              // the root carries none
        bool advanced = false;
        if (node != 0 && !(dw & 1u)) {              // the symbol repeats the one this node was entered by
            if (bm & ecgb::kContFlag) {
                advanced = true;
                if (bm & ecgb::kHeadFlag) {
                    node = fc + __popc(bm & ecgb::kBranchMask);
                    ++j;
                } else {
                    const uint32_t z = min((uint32_t)__ffs(dw) - 1u, 32u);               // symbols of this run ahead (>= 1)
                    const uint32_t ones = min((uint32_t)__ffs(~cw) - 1u, 32u);           // chain nodes from this one on that continue (>= 1)
                    const uint32_t m = min(z, ones);
                    const uint32_t passed = tw & (0xFFFFFFFFu >> (32u - m)) & ~1u;       // nodes node + 1 .. node + m - 1
                    if (passed) best_j = j + (31u - __clz(passed));
                    node += m;
                    j += m;
                }
            }
        } else {
            const uint32_t bit = 1u << s;
            if (bm & bit) {
                node = fc + __popc(bm & (bit - 1u));
                ++j;
                advanced = true;
            }
        }
        if (!advanced) {
            // emit the token [r, r + len)
            const uint32_t len = max(best_j - r, 1u);   // unmatched byte: lib.rs:186-189
            atomicOr(&marks[r >> 5], 1u << (r & 31));
            if (len >= 2) W.ids_half[r >> 1] = (uint16_t)(hi >> 16);   // the best token of the node the walk stopped at
            if constexpr (!FIRST) {   // drop marks of the old chain inside (r, r + len)
                uint32_t lo = r + 1;
                const uint32_t hi_pos = min(r + len, e_rel);
                while (lo < hi_pos) {
                    const uint32_t w = lo >> 5;
                    const uint32_t top = min(hi_pos, (w + 1) << 5);
                    uint32_t m = (top == ((w + 1) << 5)) ? 0xFFFFFFFFu : ((1u << (top & 31)) - 1u);
                    m &= 0xFFFFFFFFu << (lo & 31);
                    atomicAnd(&marks[w], ~m);
                    lo = top;
                }
            }
            r += len; j = r; node = 0; best_j = r;
            if (r >= e_rel) {
                result = r;
                done = true;
            } else if constexpr (!FIRST) {
                if ((marks[r >> 5] >> (r & 31)) & 1u) {   // re-synchronised with the old chain
                    result = old_exit;
                    done = true;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    return result;
}

// ---- emit helper: token ids of the marks in `bits` (word index `w_rel` of the segment).
template <int CHUNK, int INPUT>
__device__ __forceinline__ uint32_t emit_word(uint32_t bits, uint32_t w_rel, const uint32_t *marks,
                                              const uint8_t *sym, const uint16_t *ids_half,
                                              const uint16_t *s_single, const uint8_t *raw_seg,
                                              uint32_t seg_len, uint32_t carry_out_rel, uint32_t *out,
                                              uint32_t off, size_t ids_stride)
{
    while (bits) {
        const uint32_t t = __ffs(bits) - 1;
        bits &= bits - 1;
        const uint32_t r = w_rel * 32 + t;
        bool single;
        if (r + 1 < seg_len) single = (marks[(r + 1) >> 5] >> ((r + 1) & 31)) & 1u;
        else single = (carry_out_rel == r + 1);
        uint32_t id;
        if (single) {
            const uint32_t s = sym[swz<CHUNK>(r)];
            if (INPUT == INPUT_BYTES && s == ecgb::kOtherClass) id = raw_seg[r];
            else id = s_single[s];
        } else {
            id = ids_half[r >> 1];
        }
        if (off < ids_stride) out[off] = id;
        ++off;
    }
    return off;
}

// ==========================================================================================
// Kernel 1 (large batches): ONE WAVE = ONE STREAM.  A workgroup is W independent waves that share
// only the LDS tables; each wave takes streams b = wave_id, wave_id + total_waves, ... and parses
// them in segments of 64 chunks.  All synchronisation is wave-local (shuffles, ballots, program
// order of the wave's own LDS traffic): no workgroup barrier after start-up, so the waves of a CU
// drift into different phases and the HBM-bound staging of some overlaps the parsing of others.
//
// Per segment:
//   parse    lane c starts a greedy parse at the start of chunk c (lane 0: at the carry, which is the
//            one start known to lie on the real chain).  Every token start is CLAIMED with a returning
//            LDS atomic-or on the segment's bitmap; a parse that reaches a position somebody claimed
//            before has joined that parse and stops, otherwise it keeps going -- past its chunk,
//            through later chunks, to the end of the segment.  Every position is therefore parsed at
//            most once, nothing is ever re-walked, and a lane is idle only from its join to the end
//            of the loop.  The tokens of a trip (position, id; 0xFFFF for an unmatched byte) are appended
//            to the wave's token list in global memory: consecutive 4-byte entries, one coalesced
//            store per trip, a few hundred entries per segment (dense, L2-resident).
//   resolve  the claimed positions are the real chain plus the speculative prefixes that joined it.
//            One sweep of the list puts the token lengths (id -> length table) into the symbol
//            bytes, then lane c follows the length pointers through positions [64c, 64c+64) from a
//            guessed entry; entries are corrected from the left neighbour's exit until nothing
//            changes (a few LDS hops per pass, no trie walking).
//   emit     the real chain's bitmap and its per-word prefix counts go to LDS; a second sweep of the
//            list writes every entry that lies on the real chain to its output slot.
template <int INPUT, bool ALL_LDS, bool VEC, bool WIDE>
__global__ __launch_bounds__(1024) void encode_flow_kernel(EncodeArgs A)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t n_waves = blockDim.x >> 6;
    const uint32_t CH = A.chunk, SEG = 64u * CH, MARKW = SEG / 32;            // SEG is a multiple of 64
    constexpr uint32_t WPL = WIDE ? 4u : 2u, PW = 32u * WPL;                  // bitmap words / positions a lane owns outside the parse (WIDE: segments > 4096)
    const uint32_t sym_cap = SEG + A.margin;                                   // multiple of 16
    uint64_t *s_trie = reinterpret_cast<uint64_t *>(smem);
    double *s_thr = reinterpret_cast<double *>(s_trie + A.n_lds_nodes);        // 28
    uint16_t *s_single = reinterpret_cast<uint16_t *>(s_thr + 28);             // 32
    uint8_t *s_b2c = reinterpret_cast<uint8_t *>(s_single + 32);               // 256
    uint32_t *s_run = reinterpret_cast<uint32_t *>(s_b2c + 256);               // n_runwords (even)
    uint8_t *s_len = reinterpret_cast<uint8_t *>(s_run + A.n_runwords);        // n_toklen (multiple of 8)
    uint8_t *s_sym_all = s_len + A.n_toklen;
    uint32_t *s_marks_all = reinterpret_cast<uint32_t *>(s_sym_all + (size_t)n_waves * sym_cap);
    const uint32_t dmap_words = sym_cap / 32 + 2;                              // + the word a window read may touch past the end
    uint32_t *s_dmap_all = s_marks_all + n_waves * MARKW;

    const uint32_t tid = threadIdx.x, wave = tid >> 6, c = tid & 63;
    for (uint32_t i = tid; i < A.n_lds_nodes; i += blockDim.x) s_trie[i] = A.trie[i];
    for (uint32_t i = tid; i < A.n_runwords; i += blockDim.x) s_run[i] = A.runbits[i];
    for (uint32_t i = tid; i < A.n_toklen / 4; i += blockDim.x)
        reinterpret_cast<uint32_t *>(s_len)[i] = reinterpret_cast<const uint32_t *>(A.tok_len)[i];
    if (tid < 28) s_thr[tid] = A.qp.thr[tid];
    if (tid < 32) s_single[tid] = reinterpret_cast<const uint16_t *>(A.lut + 256)[tid];
    if (tid < 256) s_b2c[tid] = A.lut[tid];
    __syncthreads();   // the only workgroup barrier

    uint8_t *sym = s_sym_all + (size_t)wave * sym_cap;
    uint32_t *marks = s_marks_all + wave * MARKW;
    uint32_t *dmap = s_dmap_all + wave * dmap_words;
    const uint32_t gw = blockIdx.x * n_waves + wave, total_waves = gridDim.x * n_waves;
    // (the wave's list base as a scalar: list stores then take the base + 32-bit offset form)
    uint32_t *tok_list = reinterpret_cast<uint32_t *>(A.ids_half) + (size_t)__builtin_amdgcn_readfirstlane(gw) * kFlowSlot;
    const uint32_t n = A.n;
    const double qa = A.qp.a, qscale = A.qp.scale, qnas = A.qp.nas;
    const float qamb = A.qp.amb_thr;
#ifdef ECGB_PROFILE
    long long t_prof = clock64();
#endif

    for (uint32_t b = gw; b < A.batch; b += total_waves) {
        const size_t row = (size_t)b * n;
        uint32_t *out = A.ids_out + (size_t)b * A.ids_stride;
        uint32_t carry = 0, out_off = 0;
        for (uint32_t seg_base = 0; seg_base < n; seg_base += SEG) {
            const uint32_t seg_len = min(SEG, n - seg_base);
            const uint32_t stage_len = min(sym_cap, (n - seg_base + 16u) & ~15u);   // >= 1 sentinel past n
            // The look-ahead margin of the previous segment IS the head of this one: its symbol bytes are still in LDS, untouched (the resolve phase
            // writes lengths only at token starts < seg_len), so they move down inside LDS instead of being fetched and classified again -- the margin
            // was staged twice per segment: 6.5 % of the signal traffic at C2's 3 456-symbol segments (profiles/r02/hbm_pmc.json: 2.36 GB against 2.04).
            // Both stagings end at the same absolute position (SEG is a multiple of 16), so every sentinel this segment needs is in the moved bytes too.
            const uint32_t pre = (seg_base == 0) ? 0u : min(A.margin, stage_len);
            if (pre) {
                for (uint32_t k = c * 4; k < pre; k += 256) {
                    const uint32_t v = *reinterpret_cast<const uint32_t *>(sym + SEG + k);
                    *reinterpret_cast<uint32_t *>(sym + k) = v;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            const uint32_t n_rest = n - seg_base;
            stage_symbols<0, INPUT, VEC>(sym + pre, stage_len - pre, n_rest > pre ? n_rest - pre : 0u, A.signal + row + seg_base + pre,
                                         A.raw + row + seg_base + pre, c, 64, qa, qscale, qnas, qamb, s_thr, s_b2c);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            build_dmap<0>(sym, dmap, stage_len, c, 64);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            PROF_STAMP(0);

            // ---- chunk starts: equal numbers of RUNS per lane, not of symbols (a flat stretch costs a step per ~30
            // symbols, a flickering one a step per symbol).  Lane l counts the run starts in positions [64l, 64l+64),
            // a wave scan ranks them, and chunk k begins at the run start of rank k*R/64.  The claim bitmap is not in use
            // yet and holds the 64 starts for a moment.  Any start is valid (claims make the parse exact whatever the
            // starts are); the balance is what this buys.
            const uint32_t carry_rel = carry - seg_base;
            uint32_t my_start = c * CH;
            {
                const uint32_t p0 = PW * c;                                     // this lane's positions [p0, p0 + PW)
                uint32_t bw[WPL], nw[WPL], n_l = 0;
#pragma unroll
                for (uint32_t k = 0; k < WPL; ++k) {
                    const uint32_t q0 = p0 + 32 * k;
                    uint32_t v = (q0 < seg_len) ? dmap[WPL * c + k] : 0u;
                    if (q0 < seg_len && seg_len - q0 < 32) v &= (1u << (seg_len - q0)) - 1u;
                    bw[k] = v;
                    nw[k] = (uint32_t)__popc(v);
                    n_l += nw[k];
                }
                uint32_t incl = n_l;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t t = __shfl_up(incl, d, 64);
                    if (c >= (uint32_t)d) incl += t;
                }
                const uint32_t R = __shfl(incl, 63, 64), P = incl - n_l;
                marks[c] = my_start;                                            // fallback: equal symbol counts
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (n_l) {
                    const float inv = 64.0f / (float)R;
                    uint32_t k = max((uint32_t)ceilf((float)P * inv), 1u);
                    const uint32_t k_hi = min((uint32_t)ceilf((float)(P + n_l) * inv), 64u);
                    for (; k < k_hi; ++k) {
                        const uint32_t t = (k * R) >> 6;
                        if (t < P || t >= P + n_l) continue;
                        uint32_t want = t - P, w = bw[0], pos = 0;              // the want-th (0-based) set bit of the lane's words
#pragma unroll
                        for (uint32_t q = 0; q + 1 < WPL; ++q)
                            if (pos == 32 * q && want >= nw[q]) { want -= nw[q]; w = bw[q + 1]; pos = 32 * (q + 1); }
#pragma unroll
                        for (int sh = 16; sh >= 1; sh >>= 1) {
                            const uint32_t below = (uint32_t)__popc(w & ((1u << sh) - 1u));
                            if (want >= below) { want -= below; w >>= sh; pos += sh; }
                        }
                        marks[k] = p0 + pos;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                my_start = marks[c];
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            for (uint32_t w = c; w < MARKW; w += 64) marks[w] = 0;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();

            // ---- parse
            // A trip is one trie step, or -- when the step fails -- the emission of the token the walk has matched: its id
            // is the stopped-at node's best token (tokenizer.hpp), its length comes from the id -> length table, the next
            // token start is claimed, and the first symbol of the next token is taken on the spot (the root's children are
            // nodes 1 .. n_classes in class order), so the root is never visited and a token costs its steps past the
            // first symbol plus one.  The step itself is branch-free: "the symbol repeats" selects the continuation flag
            // instead of the symbol's class bit (kContFlag - 1 == kBranchMask makes the child formula the same), and inside
            // a same-class chain the step length is the distance to the first position that starts a new run or the first
            // chain node without a continuation.  All five LDS reads of a trip are issued together and waited for once.
            uint32_t n_list = 0;                                            // tokens in the wave's list (real chain + speculative prefixes)
            {
                uint32_t r = (c == 0) ? carry_rel : max(my_start, carry_rel);   // chunks the carry token covers start at the carry
                uint32_t j = r, node = 0;
                bool live = r < seg_len;
                uint32_t dmap_a = lds_addr(dmap), marks_a = lds_addr(marks);
                asm volatile("" : "+v"(dmap_a), "+v"(marks_a));             // (kept in registers, not recomputed every trip)
                const uint32_t run_a = __builtin_amdgcn_readfirstlane(lds_addr(s_run));
                uint32_t *tok_tail = tok_list;                                  // end of the list (wave-uniform)
                if (live) {
                    const uint32_t rbit = 1u << (r & 31);
                    const uint32_t claimed = atomicOr(&marks[r >> 5], rbit);
                    const uint32_t s0 = sym[r];
                    live = !(claimed & rbit);                                   // two lanes at the carry: one of them goes
                    if (INPUT == INPUT_F64 || s0 != ecgb::kOtherClass) { node = 1u + s0; j = r + 1u; }
                }
                while (live) {
#ifdef ECGB_PROFILE
                    if (A.prof && c == (unsigned)(__ffsll((long long)__ballot(1)) - 1)) atomicAdd(&A.prof[blockIdx.x * 8 + 5], 1ull);
#endif
                    uint64_t rec;
                    if constexpr (ALL_LDS) rec = s_trie[node];
                    else rec = (node < A.n_lds_nodes) ? s_trie[node] : A.trie[node];
                    lds_cu32 *dp = lds_word32<2>(j, dmap_a), *cp = lds_word32_uniform<3>(node, run_a);
                    const uint32_t d0 = dp[0], d1 = dp[1];
                    const uint32_t c0 = cp[0], c1 = cp[2];
                    const uint32_t s = sym[j];
                    const uint32_t bm = (uint32_t)rec, hi = (uint32_t)(rec >> 32);
                    // the length of the node's best token is fetched for every lane as soon as the node record is in (it is
                    // issued first and LDS answers in order): a trip that emits finds it there instead of waiting for it
                    const uint32_t id = hi >> 16;
                    uint32_t len;
                    if (INPUT == INPUT_F64) len = s_len[id];                             // every node below the root has a best token
                    else len = s_len[min(id, A.n_toklen - 1u)];
                    asm volatile("" :: "v"(d0), "v"(d1), "v"(c0), "v"(c1), "v"(s));   // one wait for the rest, here
                    const uint32_t dw = __builtin_amdgcn_alignbit(d1, d0, j & 31);       // bit t: position j + t starts a new run
                    const uint32_t cw = __builtin_amdgcn_alignbit(c1, c0, node & 31);    // bit t: node + t has a continuation
                    bool norep = dw & 1u;                                                // the symbol differs from the one this node was entered by
                    if (INPUT != INPUT_F64) norep = norep || node == 0;                  // (the root: after an unmatched byte)
                    const uint32_t width = norep ? s : 30u;                              // the class bit, or the continuation flag: bit 30
                    const bool ok = ((bm >> (width & 31u)) & 1u) != 0;
                    const unsigned long long em = __ballot(1) & ~__ballot(ok);           // the lanes that emit a token in this trip
                    const uint32_t child = (hi & 0xFFFFu) + (uint32_t)__popc(__builtin_amdgcn_ubfe(bm, 0u, width));   // children of the classes below
                    const bool inchain = !norep && (int32_t)bm >= 0;                     // not the head of its chain: the continuation is node + 1
                    const uint32_t stop = dw | ~cw;                                      // bit 0 is clear when the step is a chain step
                    const uint32_t m = inchain ? min((uint32_t)__ffs(stop) - 1u, 32u) : 1u;
                    const uint32_t nn = inchain ? node + m : child;
                    if (ok) {
                        node = nn;
                        j += m;
                    } else {
                        // emit the token that starts at r: appended to the wave's list (consecutive 4-byte entries, one
                        // coalesced store per trip)
                        if (INPUT != INPUT_F64 && id == ecgb::kNoToken) len = 1u;       // unmatched byte: lib.rs:186-189
                        tok_tail[__builtin_amdgcn_mbcnt_hi((uint32_t)(em >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)em, 0u))] = r | (id << 16);
                        r += len;
                        live = r < seg_len;
                        if (live) {
                            const uint32_t rbit = 1u << (r & 31);
                            const uint32_t claimed = atomicOr(const_cast<uint32_t *>((const uint32_t *)lds_word32<2>(r, marks_a)), rbit);
                            const uint32_t s1 = sym[r];
                            live = !(claimed & rbit);                                    // joined a parse that got here first
                            node = 0; j = r;
                            if (INPUT == INPUT_F64 || s1 != ecgb::kOtherClass) { node = 1u + s1; j = r + 1u; }
                        }
                    }
                    tok_tail += (uint32_t)__popcll(em);                                  // (a scalar pointer: the list grows by the tokens of this trip)
                    __builtin_amdgcn_wave_barrier();
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // list entries other lanes stored are read below (same CU, same L1)
            __builtin_amdgcn_wave_barrier();
            PROF_STAMP(1);

            // ---- resolve: lane c owns positions [PW*c, PW*c + PW)  (PW = 64, or 128 for segments longer than 4096)
            const uint32_t pbase = PW * c;
            const uint32_t s_blk = min(pbase, seg_len), e_blk = min(pbase + PW, seg_len);
            uint32_t first_claimed = e_blk;
#pragma unroll
            for (uint32_t k = WPL; k-- > 0;) {
                const uint32_t v = (WPL * c + k < MARKW) ? marks[WPL * c + k] : 0u;
                if (v) first_claimed = pbase + 32 * k + (uint32_t)__ffs(v) - 1u;
                n_list += (uint32_t)__popc(v);
            }
            // every claimed position got exactly one list entry (its claimant emitted the token that starts there), so the
            // length of the list is the population of the claim bitmap -- the parse loop keeps its own count in a scalar
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) n_list += (uint32_t)__shfl_xor(n_list, d, 64);
            // the first kListRegs * 64 entries of the list (all of it unless the segment is mostly two-symbol tokens) are read
            // once, all loads in flight together, and stay in registers for both sweeps
            uint32_t ent[kListRegs];
#pragma unroll
            for (uint32_t k = 0; k < kListRegs; ++k) ent[k] = (c + 64 * k < n_list) ? tok_list[c + 64 * k] : 0xFFFFFFFFu;
#pragma unroll
            for (uint32_t k = 0; k < kListRegs; ++k) {   // token lengths into the symbol bytes
                const uint32_t id = ent[k] >> 16;
                if (id != ecgb::kNoToken) sym[ent[k] & 0xFFFFu] = (uint8_t)(kLenBias + s_len[id]);
            }
            for (uint32_t i = c + 64 * kListRegs; i < n_list; i += 64) {
                const uint32_t e = tok_list[i], id = e >> 16;
                if (id != ecgb::kNoToken) sym[e & 0xFFFFu] = (uint8_t)(kLenBias + s_len[id]);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            PROF_STAMP(6);

            auto follow = [&](uint32_t p, unsigned long long &lo, unsigned long long &hi) {   // length pointers through this lane's block
                lo = 0; hi = 0;
                while (p < e_blk) {
                    const uint32_t rel = p - pbase;
                    if (!WIDE || rel < 64) lo |= 1ull << rel; else hi |= 1ull << (rel - 64);
                    const uint32_t bb = sym[p];
                    p += (bb < kLenBias) ? 1u : bb - kLenBias;
                }
                return p;
            };
            unsigned long long chain = 0, chain_hi = 0;
            uint32_t entry = (c == 0) ? carry_rel : first_claimed;
            entry = max(entry, s_blk);
            uint32_t my_exit = follow(entry, chain, chain_hi);
            PROF_STAMP(7);
            for (;;) {
                const uint32_t prev = __shfl_up(my_exit, 1, 64);
                const uint32_t want = (c == 0) ? carry_rel : prev;
                const bool changed = (want != entry) && (pbase < seg_len);   // lanes past the segment own nothing: they must not
                                                                             // pass the last exit along, one lane per pass
                if (!__any(changed)) break;
                PROF_COUNT(4, 1);
                if (changed) {
                    entry = want;
                    my_exit = follow(entry, chain, chain_hi);
                }
                __builtin_amdgcn_wave_barrier();
            }
            const uint32_t carry_out_rel = __shfl(my_exit, (int)((seg_len - 1u) / PW), 64);   // the exit of the last block of the segment
            PROF_STAMP(2);

            // ---- emit: the real chain's bitmap replaces the claim bitmap, its per-word prefix counts go where the change map
            // was; then the list is swept once more and every entry on the real chain lands in its output slot
            const uint32_t cnt = (uint32_t)(__popcll(chain) + __popcll(chain_hi));
            uint32_t incl = cnt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t t = __shfl_up(incl, d, 64);
                if (c >= (uint32_t)d) incl += t;
            }
            const uint32_t total = __shfl(incl, 63, 64);
            const uint32_t off = out_off + incl - cnt;
            {
                const uint32_t cw[4] = {(uint32_t)chain, (uint32_t)(chain >> 32), (uint32_t)chain_hi, (uint32_t)(chain_hi >> 32)};
                uint32_t run = off;
#pragma unroll
                for (uint32_t k = 0; k < WPL; ++k) {
                    if (WPL * c + k < MARKW) { marks[WPL * c + k] = cw[k]; dmap[WPL * c + k] = run; }
                    run += (uint32_t)__popc(cw[k]);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            auto place = [&](uint32_t e) {
                const uint32_t pos = e & 0xFFFFu;
                const uint32_t tb = marks[pos >> 5], bit = 1u << (pos & 31);
                if (!(tb & bit)) return;                                     // a speculative prefix, not on the real chain
                const uint32_t slot = dmap[pos >> 5] + (uint32_t)__popc(tb & (bit - 1u));
                uint32_t id = e >> 16;
                if (id == ecgb::kNoToken) {                                  // unmatched byte: its byte still holds the class
                    const uint32_t cls = sym[pos];
                    if (INPUT == INPUT_BYTES && cls == ecgb::kOtherClass) id = A.raw[row + seg_base + pos];
                    else id = s_single[cls];
                }
                if (slot < A.ids_stride) out[slot] = id;
            };
#pragma unroll
            for (uint32_t k = 0; k < kListRegs; ++k)
                if (c + 64 * k < n_list) place(ent[k]);
            for (uint32_t i = c + 64 * kListRegs; i < n_list; i += 64) place(tok_list[i]);
            out_off += total;
            carry = seg_base + carry_out_rel;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            PROF_STAMP(3);
        }
        if (c == 0) A.counts[b] = out_off;
    }
}

// ==========================================================================================
// Kernel 2 (small batches): ONE WORKGROUP OF 256 LANES = ONE STREAM, segments of 256 chunks of
// 128 symbols, so a single record still spreads over 4 waves and few records over many CUs.
// Stage, speculative chunk parse, re-parse from corrected entries to a fixed point (the stitch), emit -- with workgroup
// barriers between the phases.
constexpr int kChunk = 128;                 // symbols per lane-chunk (multiple of 32)
constexpr int kLanes = 256;                 // lanes (chunks) per stream per segment
constexpr int kSeg = kChunk * kLanes;       // symbols per segment = 32768
constexpr int kWordsPerChunk = kChunk / 32; // 4
constexpr int kMarkWords = kSeg / 32;       // 1024 words = 4 KiB
constexpr int kHalfPerSlot = kSeg / 2;      // u16 id slots per resident stream (global scratch)

template <int INPUT, bool ALL_LDS, bool VEC>
__global__ __launch_bounds__(kLanes) void encode_wg_kernel(EncodeArgs A)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t sym_cap = kSeg + A.margin;
    uint64_t *s_trie = reinterpret_cast<uint64_t *>(smem);
    double *s_thr = reinterpret_cast<double *>(s_trie + A.n_lds_nodes);        // 28
    uint16_t *s_single = reinterpret_cast<uint16_t *>(s_thr + 28);             // 32
    uint8_t *s_b2c = reinterpret_cast<uint8_t *>(s_single + 32);               // 256
    uint32_t *s_run = reinterpret_cast<uint32_t *>(s_b2c + 256);               // n_runwords (even)
    uint8_t *sym = reinterpret_cast<uint8_t *>(s_run + A.n_runwords);
    uint32_t *marks = reinterpret_cast<uint32_t *>(sym + sym_cap);
    uint32_t *exits = marks + kMarkWords;
    uint32_t *wsum = exits + kLanes;                                           // 4 wave totals
    uint32_t *dmap = wsum + 4;                                                 // sym_cap / 32 + 2 words

    const uint32_t c = threadIdx.x;
    for (uint32_t i = c; i < A.n_lds_nodes; i += kLanes) s_trie[i] = A.trie[i];
    for (uint32_t i = c; i < A.n_runwords; i += kLanes) s_run[i] = A.runbits[i];
    if (c < 28) s_thr[c] = A.qp.thr[c];
    if (c < 32) s_single[c] = reinterpret_cast<const uint16_t *>(A.lut + 256)[c];
    s_b2c[c] = A.lut[c];

    uint32_t *my = marks + c * kWordsPerChunk;
    uint16_t *ids_half = A.ids_half + (size_t)blockIdx.x * kHalfPerSlot;
#ifdef ECGB_PROFILE
    const WalkCtx W{sym, marks, dmap, ids_half, s_trie, A.trie, s_run, A.n_lds_nodes, A.prof ? A.prof + blockIdx.x * 8 : nullptr};
    long long t_prof = clock64();
#else
    const WalkCtx W{sym, marks, dmap, ids_half, s_trie, A.trie, s_run, A.n_lds_nodes};
#endif
    const uint32_t n = A.n;
    const double qa = A.qp.a, qscale = A.qp.scale, qnas = A.qp.nas;
    const float qamb = A.qp.amb_thr;

    for (uint32_t b = blockIdx.x; b < A.batch; b += gridDim.x) {
        const size_t row = (size_t)b * n;
        uint32_t *out = A.ids_out + (size_t)b * A.ids_stride;
        uint32_t carry = 0, out_off = 0;
        for (uint32_t seg_base = 0; seg_base < n; seg_base += kSeg) {
            const uint32_t seg_len = min((uint32_t)kSeg, n - seg_base);
            const uint32_t s_rel = min(c * (uint32_t)kChunk, seg_len), e_rel = min(s_rel + kChunk, seg_len);
            __syncthreads();   // previous segment fully emitted; (first time) trie + tables staged
            const uint32_t stage_len = min(sym_cap, (n - seg_base + 16u) & ~15u);
            stage_symbols<kChunk, INPUT, VEC>(sym, stage_len, n - seg_base, A.signal + row + seg_base,
                                      A.raw + row + seg_base, c, kLanes, qa, qscale, qnas, qamb, s_thr, s_b2c);
#pragma unroll
            for (int w = 0; w < kWordsPerChunk; ++w) my[w] = 0;
            __syncthreads();
            build_dmap<kChunk>(sym, dmap, stage_len, c, kLanes);
            __syncthreads();
            PROF_STAMP(0);

            uint32_t entry = s_rel, my_exit = s_rel;
            if (s_rel < e_rel)
                my_exit = walk_chunk<kChunk, true, ALL_LDS>(W, s_rel, e_rel, s_rel, s_rel);
            exits[c] = my_exit;
            PROF_STAMP(1);
            const uint32_t carry_rel = carry - seg_base;
            for (;;) {
                __syncthreads();
                const uint32_t want = (c == 0) ? carry_rel : exits[c - 1];
                const int changed = (want != entry);
                const int any = __syncthreads_or(changed);
                if (!any) break;
                PROF_COUNT(4, 1);
                if (changed) {
                    entry = want;
                    my_exit = walk_chunk<kChunk, false, ALL_LDS>(W, s_rel, e_rel, entry, my_exit);
                    exits[c] = my_exit;
                }
            }
            const uint32_t carry_out_rel = exits[kLanes - 1];
            PROF_STAMP(2);

            uint32_t cnt = 0;
#pragma unroll
            for (int w = 0; w < kWordsPerChunk; ++w) cnt += __popc(my[w]);
            uint32_t incl = cnt;
            const uint32_t lane = c & 63, wv = c >> 6;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t t = __shfl_up(incl, d, 64);
                if (lane >= (uint32_t)d) incl += t;
            }
            if (lane == 63) wsum[wv] = incl;
            __syncthreads();
            uint32_t base = 0, total = 0;
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) { const uint32_t t = wsum[k]; if (k < wv) base += t; total += t; }
            uint32_t off = out_off + base + incl - cnt;
            for (int w = 0; w < kWordsPerChunk; ++w)
                off = emit_word<kChunk, INPUT>(my[w], c * kWordsPerChunk + w, marks, sym, ids_half, s_single,
                                       A.raw + row + seg_base, seg_len, carry_out_rel, out, off, A.ids_stride);
            out_off += total;
            carry = seg_base + carry_out_rel;
            PROF_STAMP(3);
        }
        if (c == 0) A.counts[b] = out_off;
    }
}

#include "encode_long.inc"
#include "encode_pipe.inc"

// ==========================================================================================
// Kernel 4 (the GENERAL form of a tokenizer: more than 29 byte values in the merges, 65 535 or more trie nodes, token ids >= 65 535 -- what the packed
// 8-byte node cannot hold): lib.rs:149-193 as it is written, ONE LANE = ONE STREAM, the trie's edges in an open-addressing table in global memory
// (key (node << 8 | byte) + 1, tokenizer.cpp build_general), 32-bit ids.  Slow (a dependent L2 read per symbol and lane) and total: it exists so that
// `encode_text` has no merges list it refuses, not to be fast -- the a..z streams of the hot path never come here.
struct GeneralArgs {
    const uint64_t *keys;
    const uint32_t *child, *token;
    uint32_t cap_mask;
    const uint8_t *text;      // batch x n bytes (or alphabet indices, see `add`)
    uint32_t add;             // added to every input byte: 97 when the input is the quantiser's alphabet indices
    uint32_t *ids_out;
    size_t ids_stride;
    uint32_t *counts;
    uint32_t n, batch;
};

__device__ __forceinline__ uint64_t mix64_dev(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}

__global__ __launch_bounds__(64) void encode_general_kernel(GeneralArgs A)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= A.batch) return;
    const uint8_t *text = A.text + (size_t)b * A.n;
    uint32_t *out = A.ids_out + (size_t)b * A.ids_stride;
    uint32_t i = 0, cnt = 0;
    while (i < A.n) {
        uint32_t node = 0, j = i, best = ecgb::kNoToken32, blen = 0;
        while (j < A.n) {                                                   // lib.rs:170-181: walk while a child exists, remember the deepest token
            const uint64_t key = (((uint64_t)node << 8) | ((text[j] + A.add) & 0xFFu)) + 1;
            uint32_t h = (uint32_t)mix64_dev(key) & A.cap_mask;
            uint64_t k = A.keys[h];
            while (k != 0 && k != key) { h = (h + 1) & A.cap_mask; k = A.keys[h]; }
            if (k == 0) break;
            node = A.child[h];
            ++j;
            const uint32_t t = A.token[node];
            if (t != ecgb::kNoToken32) { best = t; blen = j - i; }
        }
        uint32_t id = best;
        if (best == ecgb::kNoToken32) { id = (text[i] + A.add) & 0xFFu; blen = 1; }   // lib.rs:186-189 (dead for byte input: every byte is a token)
        if (cnt < A.ids_stride) out[cnt] = id;
        ++cnt;
        i += blen;
    }
    A.counts[b] = cnt;
}

int check_hip(hipError_t e, const char *what)
{
    if (e == hipSuccess) return ECGB_OK;
    ecgb::set_error(std::string(what) + ": " + hipGetErrorString(e));
    return ECGB_ERR_HIP;
}

constexpr size_t kAlign = 256;
inline size_t align_up(size_t x) { return (x + kAlign - 1) / kAlign * kAlign; }
constexpr size_t kLdsCap = 160 * 1024;        // static + dynamic LDS of a workgroup
constexpr size_t kLdsStaticWg = 256;          // what hipcc allocates statically in encode_wg_kernel (__syncthreads_or)
constexpr size_t kLdsTablesFixed = 28 * 8 + 64 + 256;
constexpr size_t kMaxWaves = 16;
constexpr uint32_t kFlowChunks[] = {126, 124, 122, 118, 116, 114, 110, 108, 106, 102, 100, 98, 94, 92, 90, 86, 84, 82, 78, 76, 74, 70, 68, 66,
                                    62, 60, 58, 54, 52, 50, 46, 44, 42, 38, 36, 34};   // not multiples of 8: lanes at equal
                                                                                       // chunk offsets land in different LDS banks
int g_plan_mode = 0;   // 0 auto, 1 workgroup-per-stream, 2 wave-per-stream, 3 wave-per-stream with 8 waves per CU and segments up to 8064 symbols, 4 lane-per-chunk (encode_long_kernel; opt-in: measured slower), 5 = 0 (tests / tuning)

struct Plan {
    uint32_t chunk;    // chunk length of the flow kernel
    bool wave;         // wave-per-stream (flow) kernel
    unsigned grid, block;
    uint32_t margin, n_lds;
    size_t lds;
};

inline size_t flow_per_wave(uint32_t chunk, uint32_t margin)
{
    const size_t seg = 64 * (size_t)chunk, cap = seg + margin;
    return cap + seg / 8 + (cap / 32 + 2) * 4;   // symbols, claim bitmap, change map
}

// Large batches: wave-per-stream, 16 waves per CU (fewer only if the tables leave no room for a useful
// segment), the segment as long as LDS allows, one persistent workgroup per CU.  Small batches, or a
// tokenizer whose token lengths do not fit the flow kernel's byte encoding: workgroup-per-stream, so that
// few records still spread over many lanes and CUs.
Plan make_plan(const ecgb_tokenizer *tok, size_t batch)
{
    Plan p;
    const size_t cus = tok->n_cus > 0 ? (size_t)tok->n_cus : 256;
    const size_t n_nodes = tok->nodes.size();
    const bool flow_ok = !tok->tok_len.empty() && tok->max_depth + kLenBias <= 255 && tok->tok_len.size() <= 16384;
    const int pm = g_plan_mode >= 4 ? 0 : g_plan_mode;     // (4 / 5 choose encode_long_kernel or keep it out; below that, the segment kernels plan as in mode 0)
    p.wave = flow_ok && ((pm >= 2) || (pm == 0 && batch >= 2 * cus));
    p.chunk = 0;
    if (p.wave) {
        p.margin = (tok->max_depth + 1 + 15u) & ~15u;
        const size_t tables = kLdsTablesFixed + tok->runbits.size() * 4 + tok->tok_len.size();
        const size_t trie_bytes = n_nodes * 8;
        size_t waves = 8;
        p.chunk = 34;
        bool found = false;
        // 16 waves per CU with 3 456-symbol segments finish a round of records in ~1.0 time units, 8 waves per CU with
        // segments twice as long in ~0.62 (measured on C2: fewer loop trips per record, but half the waves to hide
        // latency): take whichever needs less time for this batch's rounds -- e.g. 8 waves when the batch holds at most
        // 8 records per CU, 16 waves when it holds 9..16.
        const size_t per_cu = (batch + cus - 1) / cus;
        const double t16 = (double)((per_cu + 15) / 16) * 1.0, t8 = (double)((per_cu + 7) / 8) * 0.62;
        const bool prefer8 = (pm == 3) || (pm == 0 && t8 < t16);
        for (size_t w : {(size_t)16, (size_t)12, (size_t)8}) {
            if (prefer8 && w != 8) continue;
            if (tables + trie_bytes >= kLdsCap) break;
            const size_t per_wave_max = (kLdsCap - tables - trie_bytes) / w;
            for (uint32_t ch : kFlowChunks) {
                if (ch > 62 && w != 8) continue;      // long segments only pay with few waves (and need a batch that fills two rounds)
                if (flow_per_wave(ch, p.margin) <= per_wave_max) { waves = w; p.chunk = ch; found = true; break; }
            }
            if (found) break;
        }
        const size_t per_wave = flow_per_wave(p.chunk, p.margin);
        const size_t left = (kLdsCap > tables + waves * per_wave) ? kLdsCap - tables - waves * per_wave : 0;
        p.n_lds = (uint32_t)std::min<size_t>(n_nodes, left / 8);
        p.block = (unsigned)(waves * 64);
        const size_t wgs = (batch + waves - 1) / waves;
        p.grid = (unsigned)std::max<size_t>(1, std::min(wgs, cus));
        p.lds = tables + (size_t)p.n_lds * 8 + waves * per_wave;
    } else {
        p.margin = (tok->max_depth + 1 + 255u) & ~255u;   // keeps the symbol buffer a whole number of swizzle blocks
        const size_t tables = kLdsTablesFixed + tok->runbits.size() * 4;
        const size_t fixed = tables + kSeg + p.margin + kMarkWords * 4 + kLanes * 4 + 16 + ((kSeg + p.margin) / 32 + 2) * 4;
        p.n_lds = (fixed >= kLdsCap - kLdsStaticWg) ? 0u : (uint32_t)std::min<size_t>(n_nodes, (kLdsCap - kLdsStaticWg - fixed) / 8);
        p.block = kLanes;
        p.grid = (unsigned)std::max<size_t>(1, std::min(batch, 2 * cus));
        p.lds = fixed + (size_t)p.n_lds * 8;
    }
    return p;
}

// encode_long_kernel (float64 records of up to 65 535 samples, every node in LDS): waves per CU and LDS bytes, 0 waves if it does not apply
struct LongPlan { unsigned waves; size_t lds; };
LongPlan make_long_plan(const ecgb_tokenizer *tok, size_t n)
{
    LongPlan p{0, 0};
    const bool flow_ok = !tok->tok_len.empty() && tok->max_depth + kLenBias <= 255 && tok->tok_len.size() <= 16384;
    if (!flow_ok || n == 0 || n > kLongMaxN || tok->n_classes > 29) return p;
    const size_t tables = tok->nodes.size() * 8 + kLdsTablesFixed + tok->runbits.size() * 4 + tok->tok_len.size() + 16;
    if (tables + 4 * kLongWaveLds > kLdsCap) return p;
    p.waves = (unsigned)std::min<size_t>(kMaxWaves, (kLdsCap - tables) / kLongWaveLds);
    p.lds = tables + p.waves * (size_t)kLongWaveLds;
    return p;
}

int launch_encode_long(const ecgb_tokenizer *tok, const LongPlan &lp, const double *signal, const QuantParams &qp, size_t batch, size_t n,
                       void *scratch, uint32_t *ids_out, size_t ids_stride, uint32_t *counts, hipStream_t stream)
{
    const size_t cus = tok->n_cus > 0 ? (size_t)tok->n_cus : 256;
    LongArgs A;
    A.trie = tok->nodes_dev; A.n_nodes = (uint32_t)tok->nodes.size();
    A.runbits = tok->runbits_dev; A.n_runwords = (uint32_t)tok->runbits.size();
    A.tok_len = tok->toklen_dev; A.n_toklen = (uint32_t)tok->tok_len.size();
    A.lut = tok->lut_dev;
    A.signal = signal; A.qp = qp;
    A.rle_cap = (uint32_t)long_rle_cap(n); A.list_cap = (uint32_t)long_list_cap(n);
    const size_t slots = cus * kMaxWaves;
    A.rle = reinterpret_cast<uint16_t *>(scratch);
    A.lists = reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(scratch) + align_up(slots * A.rle_cap * sizeof(uint16_t)));
    A.ids_out = ids_out; A.ids_stride = ids_stride; A.counts = counts;
    A.n = (uint32_t)n; A.batch = (uint32_t)batch;
#ifdef ECGB_PROFILE
    A.prof = g_prof_dev;
#endif
    const bool vec = (n % 2 == 0) && ((reinterpret_cast<uintptr_t>(signal) & 15u) == 0);
    void (*kern)(LongArgs) = vec ? encode_long_kernel<true> : encode_long_kernel<false>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lp.lds);
    if (e != hipSuccess) return check_hip(e, ("hipFuncSetAttribute(encode_long_kernel, " + std::to_string(lp.lds) + " bytes of LDS)").c_str());
    const size_t wgs = (batch + lp.waves - 1) / lp.waves;
    hipLaunchKernelGGL(kern, dim3((unsigned)std::max<size_t>(1, std::min(wgs, cus))), dim3(lp.waves * 64), lp.lds, stream, A);
    return check_hip(hipGetLastError(), "encode_long_kernel launch");
}

// encode_pipe_kernel: stagers and walkers in one workgroup of sixteen waves, sixteen records a round.  rho: the ratio of consecutive chunk lengths (ECGB_PIPE_RHO, dev).
int launch_encode_pipe(const ecgb_tokenizer *tok, const double *signal, const QuantParams &qp, size_t batch, size_t n, void *scratch, uint32_t *ids_out, size_t ids_stride,
                       uint32_t *counts, hipStream_t stream)
{
    const size_t cus = tok->n_cus > 0 ? (size_t)tok->n_cus : 256;
    PipeArgs P;
    LongArgs &A = P.L;
    A.trie = tok->nodes_dev; A.n_nodes = (uint32_t)tok->nodes.size();
    A.runbits = tok->runbits_dev; A.n_runwords = (uint32_t)tok->runbits.size();
    A.tok_len = tok->toklen_dev; A.n_toklen = (uint32_t)tok->tok_len.size();
    A.lut = tok->lut_dev;
    A.signal = signal; A.qp = qp;
    A.rle_cap = (uint32_t)long_rle_cap(n); A.list_cap = (uint32_t)long_list_cap(n);
    const size_t slots = cus * kMaxWaves;
    A.rle = reinterpret_cast<uint16_t *>(scratch);
    A.lists = reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(scratch) + align_up(slots * A.rle_cap * sizeof(uint16_t)));
    A.ids_out = ids_out; A.ids_stride = ids_stride; A.counts = counts;
    A.n = (uint32_t)n; A.batch = (uint32_t)batch;
#ifdef ECGB_PROFILE
    A.prof = g_prof_dev;
#endif
    // chunk k = blocks [sched[k], sched[k + 1]): lengths fall geometrically along the record (a lane that starts later has less time)
    const uint32_t nblk = (uint32_t)((n + kLongBlk - 1) / kLongBlk);
    const double rho = 1.0;      // measured at C2 (profiles/r05/README.md): 1.34 ms with equal chunks, 1.46 / 1.70 with 0.975 / 0.96 -- the walkers, not the stagers' head start, bound the kernel
    double w[64], tot = 0.0, cum = 0.0;
    for (int k = 0; k < 64; ++k) { w[k] = std::pow(rho, k); tot += w[k]; }
    for (int k = 0; k < 64; ++k) { P.sched[k] = (uint16_t)std::min<double>(nblk, std::floor(cum / tot * nblk + 0.5)); cum += w[k]; }
    P.sched[64] = (uint16_t)nblk;
    for (int k = 1; k <= 64; ++k) if (P.sched[k] < P.sched[k - 1]) P.sched[k] = P.sched[k - 1];
    const size_t tables = tok->nodes.size() * 8 + kLdsTablesFixed + tok->runbits.size() * 4 + tok->tok_len.size() + 16;
    const size_t lds = tables + 8 * ((kPipeStageLds + 15) & ~15u) + 8 * (size_t)kPipeWalkLds + 16 * (size_t)kPipeSlotLds;
    if (lds > kLdsCap) { ecgb::set_error("encode_pipe_kernel: the trie leaves no room for the pipeline's buffers"); return ECGB_ERR_UNSUPPORTED; }
    const bool vec = (n % 2 == 0) && ((reinterpret_cast<uintptr_t>(signal) & 15u) == 0);
    void (*kern)(PipeArgs) = vec ? encode_pipe_kernel<true> : encode_pipe_kernel<false>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return check_hip(e, ("hipFuncSetAttribute(encode_pipe_kernel, " + std::to_string(lds) + " bytes of LDS)").c_str());
    const size_t wgs = (batch + 15) / 16;
    hipLaunchKernelGGL(kern, dim3((unsigned)std::max<size_t>(1, std::min(wgs, cus))), dim3(1024), lds, stream, P);
    return check_hip(hipGetLastError(), "encode_pipe_kernel launch");
}

template <int INPUT>
int launch_encode(const ecgb_tokenizer *tok, const double *signal, const uint8_t *raw, const QuantParams &qp,
                  size_t batch, size_t n, uint16_t *ids_half, uint32_t *ids_out, size_t ids_stride,
                  uint32_t *counts, hipStream_t stream)
{
    const Plan pl = make_plan(tok, batch);
    if (pl.lds > kLdsCap) {
        ecgb::set_error("encode: trie depth needs more LDS margin than a CU has");
        return ECGB_ERR_UNSUPPORTED;
    }
    EncodeArgs A;
    A.trie = tok->nodes_dev;
    A.n_nodes = (uint32_t)tok->nodes.size();
    A.n_lds_nodes = pl.n_lds;
    A.runbits = tok->runbits_dev;
    A.n_runwords = (uint32_t)tok->runbits.size();
    A.tok_len = tok->toklen_dev;
    A.n_toklen = (uint32_t)tok->tok_len.size();
    A.chunk = pl.chunk;
    A.lut = tok->lut_dev;
    A.signal = signal;
    A.raw = raw;
    A.qp = qp;
    A.ids_half = ids_half;
    A.ids_out = ids_out;
    A.ids_stride = ids_stride;
    A.counts = counts;
    A.n = (uint32_t)n;
    A.batch = (uint32_t)batch;
    A.margin = pl.margin;
#ifdef ECGB_PROFILE
    A.prof = g_prof_dev;
#endif
    const bool all_lds = (pl.n_lds == A.n_nodes);
    const bool vec = (INPUT == INPUT_F64) && (n % 2 == 0) && ((reinterpret_cast<uintptr_t>(signal) & 15u) == 0);
    void (*kern)(EncodeArgs) = nullptr;
    if (pl.wave && pl.chunk > 64)
        kern = all_lds ? (vec ? encode_flow_kernel<INPUT, true, true, true> : encode_flow_kernel<INPUT, true, false, true>)
                       : (vec ? encode_flow_kernel<INPUT, false, true, true> : encode_flow_kernel<INPUT, false, false, true>);
    else if (pl.wave)
        kern = all_lds ? (vec ? encode_flow_kernel<INPUT, true, true, false> : encode_flow_kernel<INPUT, true, false, false>)
                       : (vec ? encode_flow_kernel<INPUT, false, true, false> : encode_flow_kernel<INPUT, false, false, false>);
    else
        kern = all_lds ? (vec ? encode_wg_kernel<INPUT, true, true> : encode_wg_kernel<INPUT, true, false>)
                       : (vec ? encode_wg_kernel<INPUT, false, true> : encode_wg_kernel<INPUT, false, false>);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds);
    if (e != hipSuccess) return check_hip(e, ("hipFuncSetAttribute(encode kernel, " + std::to_string(pl.lds) + " bytes of LDS)").c_str());
    hipLaunchKernelGGL(kern, dim3(pl.grid), dim3(pl.block), pl.lds, stream, A);
    return check_hip(hipGetLastError(), "encode kernel launch");
}

// grid for `rows` rows of `items` work items each: x covers a row (grid-stride), y the rows;
// about 8 workgroups per CU in total so the memory system stays full.
dim3 row_grid(size_t items, size_t rows)
{
    const size_t per_row = std::max<size_t>(1, (items + 255) / 256);
    const size_t gy = std::max<size_t>(1, std::min<size_t>(rows, 65535));
    const size_t want = 256 * 8;
    const size_t gx = std::max<size_t>(1, std::min<size_t>(per_row, (want + gy - 1) / gy));
    return dim3((unsigned)gx, (unsigned)gy);
}

// Quantise `rows` rows of n samples; row r's symbols go to sym + r*sym_stride.
int launch_quantize(const double *x, size_t n, size_t rows, size_t sym_stride, double p1, double p99,
                    uint8_t *sym, double *clipped, hipStream_t stream)
{
    QuantParams qp = make_quant_params(p1, p99);
    const bool vec = (n % 4 == 0) && (sym_stride % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15u) == 0) &&
                     ((reinterpret_cast<uintptr_t>(sym) & 3u) == 0);
    if (qp.use_thresholds && !clipped) {
        if (vec)
            hipLaunchKernelGGL(quantize_thr_kernel, row_grid(n / 4, rows), dim3(256), 0, stream, x, n, rows,
                               sym_stride, qp, sym);
        else
            hipLaunchKernelGGL(quantize_thr_scalar_kernel, row_grid(n, rows), dim3(256), 0, stream, x, n, rows,
                               sym_stride, qp, sym);
    } else {
        hipLaunchKernelGGL(quantize_exact_kernel, row_grid(n, rows), dim3(256), 0, stream, x, n, rows, sym_stride,
                           qp.a, qp.d, sym, clipped);
    }
    return check_hip(hipGetLastError(), "quantize kernel launch");
}

int launch_encode_general(const ecgb_tokenizer *tok, const uint8_t *text, uint32_t add, size_t batch, size_t n, uint32_t *ids_out, size_t ids_stride, uint32_t *counts, hipStream_t stream)
{
    GeneralArgs A;
    A.keys = tok->g_keys_dev; A.child = tok->g_child_dev; A.token = tok->g_token_dev;
    A.cap_mask = (uint32_t)(tok->g_keys.size() - 1);
    A.text = text; A.add = add; A.ids_out = ids_out; A.ids_stride = ids_stride; A.counts = counts;
    A.n = (uint32_t)n; A.batch = (uint32_t)batch;
    hipLaunchKernelGGL(encode_general_kernel, dim3((unsigned)((batch + 63) / 64)), dim3(64), 0, stream, A);
    return check_hip(hipGetLastError(), "encode_general_kernel launch");
}

int check_common(const ecgb_tokenizer *tok, size_t batch, size_t n, const void *ids, size_t ids_stride,
                 const void *counts, const void *scratch, size_t scratch_bytes, const char *who)
{
    if (!tok || !ids || !counts || !scratch || ids_stride == 0) {
        ecgb::set_error(std::string(who) + ": NULL or zero-sized argument");
        return ECGB_ERR_INVALID;
    }
    if (tok->general ? !tok->g_keys_dev : !tok->nodes_dev) {
        ecgb::set_error(std::string(who) + ": tokenizer handle has no device copy (no GPU at creation)");
        return ECGB_ERR_NODEVICE;
    }
    if (n >= 0x7FFF0000ull || batch >= 0x7FFFFFFFull) {
        ecgb::set_error(std::string(who) + ": stream longer than 2^31 symbols or batch too large");
        return ECGB_ERR_UNSUPPORTED;
    }
    if (scratch_bytes < ecgb_encode_scratch_bytes(tok, batch, n)) {
        ecgb::set_error(std::string(who) + ": scratch buffer smaller than ecgb_encode_scratch_bytes()");
        return ECGB_ERR_INVALID;
    }
    return ECGB_OK;
}

}  // namespace

#ifdef ECGB_PROFILE
extern "C" void ecgb_debug_set_profile_buffer(unsigned long long *dev) { g_prof_dev = dev; }
#endif

extern "C" int ecgb_set_encode_plan(int mode)
{
    if (mode < 0 || mode > 6) { ecgb::set_error("ecgb_set_encode_plan: mode must be 0..6"); return ECGB_ERR_INVALID; }
    g_plan_mode = mode;
    return ECGB_OK;
}

extern "C" int ecgb_quantizer_thresholds(double percentile_1, double percentile_99, double *thr25)
{
    if (!thr25) { ecgb::set_error("ecgb_quantizer_thresholds: NULL argument"); return ECGB_ERR_INVALID; }
    QuantParams qp = make_quant_params(percentile_1, percentile_99);
    if (!qp.use_thresholds) {
        ecgb::set_error("ecgb_quantizer_thresholds: degenerate percentiles (scale <= 0 or non-finite)");
        return ECGB_ERR_UNSUPPORTED;
    }
    for (int k = 1; k <= 25; ++k) thr25[k - 1] = qp.thr[k];
    return ECGB_OK;
}

extern "C" int ecgb_quantize_hip(const double *signal_dev, size_t n, double percentile_1,
                                 double percentile_99, uint8_t *sym_dev, double *clipped_dev,
                                 void *stream)
{
    if (n == 0) return ECGB_OK;
    if (!signal_dev || !sym_dev) { ecgb::set_error("ecgb_quantize_hip: NULL argument"); return ECGB_ERR_INVALID; }
    // one flat row; split long inputs into rows of 2^20 samples so the 2-D grid fills the chip
    const size_t row = 1u << 20;
    hipStream_t st = (hipStream_t)stream;
    const size_t full = n / row;
    int rc = ECGB_OK;
    if (full) rc = launch_quantize(signal_dev, row, full, row, percentile_1, percentile_99, sym_dev, clipped_dev, st);
    if (rc == ECGB_OK && n % row)
        rc = launch_quantize(signal_dev + full * row, n % row, 1, n % row, percentile_1, percentile_99,
                             sym_dev + full * row, clipped_dev ? clipped_dev + full * row : nullptr, st);
    return rc;
}

extern "C" size_t ecgb_encode_scratch_bytes(const ecgb_tokenizer *tok, size_t batch, size_t n_per_stream)
{
    // per resident stream slot (reused, L2-resident): flow kernel: CUs x 16 waves x a token list of up to 4096 4-byte
    // entries (the first few hundred are used); workgroup kernel: 2 x CUs x 16384 half-resolution u16 ids
    if (tok && tok->general) return align_up(batch * n_per_stream) + kAlign;                  // the quantised symbols of ecgb_quantize_encode_hip; the walk itself needs none
    const size_t cus = (tok && tok->n_cus > 0) ? (size_t)tok->n_cus : 256;
    const size_t a = cus * kMaxWaves * kFlowSlot * 2, b = 2 * cus * (size_t)kHalfPerSlot;   // in u16 units
    size_t bytes = std::max(a, b) * sizeof(uint16_t);
    // encode_long_kernel: per resident wave the record as run-length entries (2 B per run, <= one per symbol) and 64 token lists (4 B per token, one per symbol
    // at most plus the run-on margin); only what a record really has is touched (C2: 23 KB + 18 KB of 0.43 MB)
    if (tok && (g_plan_mode == 4 || g_plan_mode == 6) && n_per_stream <= kLongMaxN && make_long_plan(tok, n_per_stream).waves) {
        const size_t slots = cus * kMaxWaves;
        bytes = std::max(bytes, align_up(slots * long_rle_cap(n_per_stream) * sizeof(uint16_t)) + slots * long_list_cap(n_per_stream) * sizeof(uint32_t));
    }
    return align_up(bytes) + kAlign;
}

extern "C" int ecgb_encode_hip(const ecgb_tokenizer *tok, const uint8_t *text_dev, size_t batch,
                               size_t n_per_stream, uint32_t *ids_dev, size_t ids_stride,
                               uint32_t *counts_dev, void *scratch_dev, size_t scratch_bytes,
                               void *stream)
{
    if (batch == 0) return ECGB_OK;
    int rc = check_common(tok, batch, n_per_stream, ids_dev, ids_stride, counts_dev, scratch_dev,
                          scratch_bytes, "ecgb_encode_hip");
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (n_per_stream == 0) return check_hip(hipMemsetAsync(counts_dev, 0, batch * 4, st), "hipMemsetAsync");
    if (!text_dev) { ecgb::set_error("ecgb_encode_hip: NULL text"); return ECGB_ERR_INVALID; }
    if (tok->general) return launch_encode_general(tok, text_dev, 0u, batch, n_per_stream, ids_dev, ids_stride, counts_dev, st);
    uint16_t *half = reinterpret_cast<uint16_t *>(align_up(reinterpret_cast<uintptr_t>(scratch_dev)));
    QuantParams qp;
    std::memset(&qp, 0, sizeof(qp));
    return launch_encode<INPUT_BYTES>(tok, nullptr, text_dev, qp, batch, n_per_stream, half, ids_dev, ids_stride,
                                      counts_dev, st);
}

extern "C" int ecgb_quantize_encode_hip(const ecgb_tokenizer *tok, const double *signal_dev, size_t batch,
                                        size_t n_per_record, double percentile_1, double percentile_99,
                                        uint32_t *ids_dev, size_t ids_stride, uint32_t *counts_dev,
                                        void *scratch_dev, size_t scratch_bytes, void *stream)
{
    if (batch == 0) return ECGB_OK;
    int rc = check_common(tok, batch, n_per_record, ids_dev, ids_stride, counts_dev, scratch_dev,
                          scratch_bytes, "ecgb_quantize_encode_hip");
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (n_per_record == 0) return check_hip(hipMemsetAsync(counts_dev, 0, batch * 4, st), "hipMemsetAsync");
    if (!signal_dev) { ecgb::set_error("ecgb_quantize_encode_hip: NULL signal"); return ECGB_ERR_INVALID; }
    if (tok->general) {     // quantise into the scratch buffer (alphabet indices), then the general walk on index + 'a'
        uint8_t *sym = reinterpret_cast<uint8_t *>(align_up(reinterpret_cast<uintptr_t>(scratch_dev)));
        rc = ecgb_quantize_hip(signal_dev, batch * n_per_record, percentile_1, percentile_99, sym, nullptr, stream);
        if (rc) return rc;
        return launch_encode_general(tok, sym, 97u, batch, n_per_record, ids_dev, ids_stride, counts_dev, st);
    }
    const QuantParams qp = make_quant_params(percentile_1, percentile_99);
    if (!qp.use_thresholds) {
        ecgb::set_error("ecgb_quantize_encode_hip: degenerate percentiles (percentile_99 + 1 + 1e-6 <= percentile_1 "
                        "or non-finite); quantise with ecgb_quantize_hip and encode with ecgb_encode_hip instead");
        return ECGB_ERR_UNSUPPORTED;
    }
    uint16_t *half = reinterpret_cast<uint16_t *>(align_up(reinterpret_cast<uintptr_t>(scratch_dev)));
    // records that fit 16-bit positions, on request: a lane per long chunk of the whole record (encode_long_kernel); otherwise the segment kernels
    // (plan 4 only: measured slower than the segment kernels at every batch size -- encode_long.inc has the numbers -- so the automatic plan does not take it)
    if (g_plan_mode == 6 && make_long_plan(tok, n_per_record).waves) {
        rc = launch_encode_pipe(tok, signal_dev, qp, batch, n_per_record, half, ids_dev, ids_stride, counts_dev, st);
        if (rc != ECGB_ERR_UNSUPPORTED) return rc;
    }
    if (g_plan_mode == 4) {
        const LongPlan lp = make_long_plan(tok, n_per_record);
        if (lp.waves) return launch_encode_long(tok, lp, signal_dev, qp, batch, n_per_record, half, ids_dev, ids_stride, counts_dev, st);
        // (plan 4 on a record or trie the kernel does not take: the segment kernels, as in plan 0)
    }
    return launch_encode<INPUT_F64>(tok, signal_dev, nullptr, qp, batch, n_per_record, half, ids_dev, ids_stride,
                                    counts_dev, st);
}
