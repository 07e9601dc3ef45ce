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
// ~1.15.  So each stream is cut into 256-symbol chunks, one lane per chunk:
//   pass 0  every lane parses its chunk speculatively from the chunk start, setting one
//           bit per token start in an LDS bitmap, storing ids of tokens longer than one
//           symbol in a half-resolution id array, and recording where its chain leaves the chunk;
//   stitch  lane c re-parses from the exit of chunk c-1 until it lands on a position its own
//           speculative chain marked (greedy chains re-synchronise after ~40 symbols on
//           ECG data) -- iterated to a fixed point, so the result is the true chain whatever
//           the data (worst case: one iteration per chunk);
//   emit    popcount + group scan of the bitmap gives every token its output slot.
// The trie (8 B/node, breadth-first) sits in LDS as far as it fits; deeper nodes come from L2.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <limits>
#include <string>

#include "tokenizer.hpp"

namespace {

constexpr int kChunk = 256;                 // symbols per lane-chunk (multiple of 32)
constexpr int kLanes = 256;                 // lanes (chunks) per stream per segment
constexpr int kSeg = kChunk * kLanes;       // symbols per segment = 65536
constexpr int kWordsPerChunk = kChunk / 32; // 8
constexpr int kMarkWords = kSeg / 32;       // 2048 words = 8 KiB per stream in flight

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
    double thr[28];  // thr[k] = smallest x whose level is >= k (k = 1..25); thr[0] = -inf, thr[26..27] = +inf
    int use_thresholds;
};

QuantParams make_quant_params(double p1, double p99)
{
    QuantParams q;
    q.a = p1 - 0.5;
    q.d = ((p99 + 0.5) - (p1 - 0.5)) + 1e-6;
    q.scale = 26.0 / q.d;
    const double inf = std::numeric_limits<double>::infinity();
    q.use_thresholds = (q.d > 0.0) && std::isfinite(q.d) && std::isfinite(q.a) && std::isfinite(q.scale);
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

// sym[i] = level(x[i]); four samples per thread per iteration, one 4-byte store.
__global__ __launch_bounds__(256) void quantize_thr_kernel(const double *__restrict__ x, size_t n,
                                                           QuantParams qp, uint8_t *__restrict__ sym)
{
    __shared__ double thr[28];
    if (threadIdx.x < 28) thr[threadIdx.x] = qp.thr[threadIdx.x];
    __syncthreads();
    const size_t n4 = n / 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const double2 v0 = *reinterpret_cast<const double2 *>(x + 4 * i);
        const double2 v1 = *reinterpret_cast<const double2 *>(x + 4 * i + 2);
        uint32_t w = level_from_thresholds(v0.x, qp.a, qp.scale, thr);
        w |= level_from_thresholds(v0.y, qp.a, qp.scale, thr) << 8;
        w |= level_from_thresholds(v1.x, qp.a, qp.scale, thr) << 16;
        w |= level_from_thresholds(v1.y, qp.a, qp.scale, thr) << 24;
        *reinterpret_cast<uint32_t *>(sym + 4 * i) = w;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t i = n4 * 4 + threadIdx.x;
        sym[i] = (uint8_t)level_from_thresholds(x[i], qp.a, qp.scale, thr);
    }
}

// Literal operation sequence of tokenizer_utils.py:15-17 (IEEE fp64 division on the device).
__global__ __launch_bounds__(256) void quantize_exact_kernel(const double *__restrict__ x, size_t n,
                                                             double a, double d,
                                                             uint8_t *__restrict__ sym,
                                                             double *__restrict__ clipped)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        double nrm = (x[i] - a) / d;
        double c = nrm;
        if (c < 0.0) c = 0.0;
        if (c > 1.0) c = 1.0;
        double s = floor(c * 26.0);
        if (s > 25.0) s = 25.0;
        if (clipped) clipped[i] = c;
        sym[i] = (s == s) ? (uint8_t)s : (uint8_t)0;
    }
}

// raw bytes -> symbol classes (generic encode_text entry)
__global__ __launch_bounds__(256) void classify_kernel(const uint8_t *__restrict__ raw, size_t n,
                                                       const uint8_t *__restrict__ lut,
                                                       uint8_t *__restrict__ cls)
{
    __shared__ uint8_t s_lut[256];
    s_lut[threadIdx.x] = lut[threadIdx.x];
    __syncthreads();
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        cls[i] = s_lut[raw[i]];
}

// ------------------------------------------------------------------------------------------
struct EncodeArgs {
    const uint64_t *trie;     // packed nodes (global copy, all of them)
    uint32_t n_nodes;
    uint32_t n_lds_nodes;     // nodes [0, n_lds_nodes) are staged in LDS
    const uint8_t *lut;       // 256 B byte->class | 32 x u16 single_id | 32 B class->byte
    const uint8_t *cls;       // batch x n symbol classes
    const uint8_t *raw;       // batch x n raw bytes (only read for class kOtherClass); may be NULL
    uint16_t *ids_half;       // batch x half_stride scratch
    size_t half_stride;
    uint32_t *ids_out;        // batch x ids_stride
    size_t ids_stride;
    uint32_t *counts;         // batch
    uint32_t n;               // symbols per stream
    uint32_t batch;
};

// clear bits [lo, hi) of a lane-owned run of mark words (absolute bit indices in the segment)
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

// E streams per workgroup, kLanes lanes each.
template <int E>
__global__ __launch_bounds__(kLanes *E) void encode_kernel(EncodeArgs A)
{
    extern __shared__ __align__(16) unsigned char smem[];
    uint64_t *s_trie = reinterpret_cast<uint64_t *>(smem);
    uint32_t *s_marks_all = reinterpret_cast<uint32_t *>(s_trie + A.n_lds_nodes);
    uint32_t *s_exit_all = s_marks_all + E * kMarkWords;
    uint32_t *s_wsum_all = s_exit_all + E * kLanes;          // E x 4 wave totals
    uint16_t *s_single = reinterpret_cast<uint16_t *>(s_wsum_all + E * 4);  // 32 entries

    const int tid = threadIdx.x;
    const int g = tid / kLanes;          // stream slot inside the workgroup
    const int c = tid % kLanes;          // chunk index inside the segment
    for (uint32_t i = tid; i < A.n_lds_nodes; i += kLanes * E) s_trie[i] = A.trie[i];
    if (tid < 32) s_single[tid] = reinterpret_cast<const uint16_t *>(A.lut + 256)[tid];

    uint32_t *marks = s_marks_all + g * kMarkWords;   // this stream's bitmap (segment-relative bits)
    uint32_t *exits = s_exit_all + g * kLanes;
    uint32_t *wsum = s_wsum_all + g * 4;

    const uint32_t b = blockIdx.x * E + g;
    const bool live = b < A.batch;
    const uint32_t n = A.n;
    const uint8_t *cls = A.cls + (size_t)(live ? b : 0) * n;
    const uint8_t *raw = A.raw ? A.raw + (size_t)(live ? b : 0) * n : nullptr;
    uint16_t *ids_half = A.ids_half + (size_t)(live ? b : 0) * A.half_stride;
    uint32_t *out = A.ids_out + (size_t)(live ? b : 0) * A.ids_stride;
    const uint64_t *g_trie = A.trie;
    const uint32_t n_lds = A.n_lds_nodes;

    uint32_t carry = 0;     // true chain position entering the segment
    uint32_t out_off = 0;   // tokens emitted so far for this stream

    for (uint32_t seg_base = 0; seg_base < n; seg_base += kSeg) {
        const uint32_t seg_end = min(seg_base + (uint32_t)kSeg, n);
        const uint32_t s_c = min(seg_base + (uint32_t)c * kChunk, seg_end);   // chunk [s_c, e_c)
        const uint32_t e_c = min(s_c + (uint32_t)kChunk, seg_end);
        uint32_t *my = marks + c * kWordsPerChunk;   // lane-owned words; bit index = pos - seg_base
#pragma unroll
        for (int w = 0; w < kWordsPerChunk; ++w) my[w] = 0;
        __syncthreads();   // trie staged (first segment) / previous segment's emit done

        // Parse the chunk from `start`, merging into whatever chain the chunk already holds.
        // Returns the position at which the chain leaves the chunk.
        auto run = [&](uint32_t start, uint32_t old_exit) -> uint32_t {
            if (start >= e_c) {   // chain jumps over this chunk
                clear_bits(marks, s_c - seg_base, e_c - seg_base);
                return start;
            }
            clear_bits(marks, s_c - seg_base, start - seg_base);
            {   // already on the existing chain?
                uint32_t r = start - seg_base;
                if ((marks[r >> 5] >> (r & 31)) & 1u) return old_exit;
            }
            uint32_t p = start, j = start, node = 0, best_len = 0, best_tok = 0;
            for (;;) {
                const uint64_t rec = (node < n_lds) ? s_trie[node] : g_trie[node];
                const uint32_t tok = (uint32_t)(rec >> 48);
                if (j != p && tok != ecgb::kNoToken) { best_len = j - p; best_tok = tok; }
                bool adv = false;
                if (j < n) {
                    const uint32_t s = cls[j];
                    if (s < ecgb::kMaxClasses) {
                        const uint32_t bm = (uint32_t)rec, bit = 1u << s;
                        if (bm & bit) {
                            node = ((uint32_t)(rec >> 32) & 0xFFFFu) + __popc(bm & (bit - 1u));
                            ++j;
                            adv = true;
                        }
                    }
                }
                if (adv) continue;
                // emit the token [p, p + len)
                const uint32_t len = best_len ? best_len : 1u;   // unmatched byte: lib.rs:186-189
                const uint32_t r = p - seg_base;
                clear_bits(marks, r + 1, min(r + len, e_c - seg_base));
                marks[r >> 5] |= 1u << (r & 31);
                if (len >= 2) ids_half[p >> 1] = (uint16_t)best_tok;
                p += len; j = p; node = 0; best_len = 0;
                if (p >= e_c) return p;
                const uint32_t r2 = p - seg_base;
                if ((marks[r2 >> 5] >> (r2 & 31)) & 1u) return old_exit;   // re-synchronised
            }
        };

        uint32_t entry = s_c;
        uint32_t my_exit = s_c;
        if (live) my_exit = run(s_c, s_c);
        exits[c] = my_exit;
        // stitch to a fixed point: entry(c) must equal exit(c-1), entry(0) = carry
        for (;;) {
            __syncthreads();
            const uint32_t want = (c == 0) ? carry : exits[c - 1];
            const int changed = live && (want != entry);
            const int any = __syncthreads_or(changed);
            if (!any) break;
            if (changed) {
                entry = want;
                my_exit = run(entry, my_exit);
                exits[c] = my_exit;
            }
        }
        const uint32_t carry_out = exits[kLanes - 1];

        // emit: lane c owns the tokens that start in its chunk
        uint32_t cnt = 0;
#pragma unroll
        for (int w = 0; w < kWordsPerChunk; ++w) cnt += __popc(my[w]);
        uint32_t incl = cnt;
        const int lane = tid & 63, wv = c >> 6;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t t = __shfl_up(incl, d, 64);
            if (lane >= d) incl += t;
        }
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        uint32_t base = 0, total = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { uint32_t t = wsum[k]; if (k < wv) base += t; total += t; }
        uint32_t off = out_off + base + incl - cnt;
        if (live) {
            for (int w = 0; w < kWordsPerChunk; ++w) {
                uint32_t bits = my[w];
                while (bits) {
                    const uint32_t t = __ffs(bits) - 1;
                    bits &= bits - 1;
                    const uint32_t r = (uint32_t)c * kChunk + w * 32 + t;   // segment-relative
                    const uint32_t p = seg_base + r;
                    bool single;
                    if (p + 1 < seg_end) single = (marks[(r + 1) >> 5] >> ((r + 1) & 31)) & 1u;
                    else single = (carry_out == p + 1);
                    uint32_t id;
                    if (single) {
                        const uint32_t s = cls[p];
                        id = (s < ecgb::kMaxClasses) ? (uint32_t)s_single[s] : (raw ? (uint32_t)raw[p] : 0u);
                    } else {
                        id = ids_half[p >> 1];
                    }
                    if (off < A.ids_stride) out[off] = id;
                    ++off;
                }
            }
        }
        out_off += total;
        carry = carry_out;
        __syncthreads();   // marks/exits are rewritten by the next segment
    }
    if (live && c == 0) A.counts[b] = out_off;
}

size_t lds_bytes_for(int E, uint32_t n_lds_nodes)
{
    return (size_t)n_lds_nodes * 8 + (size_t)E * (kMarkWords * 4 + kLanes * 4 + 16) + 64;
}

int check_hip(hipError_t e, const char *what)
{
    if (e == hipSuccess) return ECGB_OK;
    ecgb::set_error(std::string(what) + ": " + hipGetErrorString(e));
    return ECGB_ERR_HIP;
}

constexpr size_t kAlign = 256;
inline size_t align_up(size_t x) { return (x + kAlign - 1) / kAlign * kAlign; }

int launch_encode(const ecgb_tokenizer *tok, const uint8_t *cls, const uint8_t *raw, size_t batch,
                  size_t n, uint16_t *ids_half, size_t half_stride, uint32_t *ids_out,
                  size_t ids_stride, uint32_t *counts, hipStream_t stream)
{
    EncodeArgs A;
    A.trie = tok->nodes_dev;
    A.n_nodes = (uint32_t)tok->nodes.size();
    A.lut = tok->lut_dev;
    A.cls = cls;
    A.raw = raw;
    A.ids_half = ids_half;
    A.half_stride = half_stride;
    A.ids_out = ids_out;
    A.ids_stride = ids_stride;
    A.counts = counts;
    A.n = (uint32_t)n;
    A.batch = (uint32_t)batch;
    // Streams per workgroup: 4 (1024 lanes, one workgroup per CU sharing one LDS trie) once
    // the batch fills the chip that way, else 1 so small batches still spread over CUs.
    const int E = (batch >= 1024) ? 4 : 1;
    const size_t lds_cap = 160 * 1024;
    const size_t fixed = lds_bytes_for(E, 0);
    uint32_t n_lds = (uint32_t)std::min<size_t>(A.n_nodes, (lds_cap - fixed) / 8);
    A.n_lds_nodes = n_lds;
    const size_t lds = lds_bytes_for(E, n_lds);
    const unsigned grid = (unsigned)((batch + E - 1) / E);
    hipError_t e;
    if (E == 4) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&encode_kernel<4>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return check_hip(e, "hipFuncSetAttribute(encode_kernel<4>)");
        hipLaunchKernelGGL(encode_kernel<4>, dim3(grid), dim3(kLanes * 4), lds, stream, A);
    } else {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&encode_kernel<1>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return check_hip(e, "hipFuncSetAttribute(encode_kernel<1>)");
        hipLaunchKernelGGL(encode_kernel<1>, dim3(grid), dim3(kLanes), lds, stream, A);
    }
    return check_hip(hipGetLastError(), "encode_kernel launch");
}

unsigned stream_grid(size_t work_items)
{
    size_t blocks = (work_items + 255) / 256;
    return (unsigned)std::max<size_t>(1, std::min<size_t>(blocks, 256 * 8));
}

int launch_quantize(const double *x, size_t n, double p1, double p99, uint8_t *sym, double *clipped,
                    hipStream_t stream)
{
    QuantParams qp = make_quant_params(p1, p99);
    const bool aligned = ((reinterpret_cast<uintptr_t>(x) & 15u) == 0) && ((reinterpret_cast<uintptr_t>(sym) & 3u) == 0);
    if (qp.use_thresholds && !clipped && aligned) {
        hipLaunchKernelGGL(quantize_thr_kernel, dim3(stream_grid(n / 4 + 1)), dim3(256), 0, stream, x, n, qp, sym);
    } else {
        hipLaunchKernelGGL(quantize_exact_kernel, dim3(stream_grid(n)), dim3(256), 0, stream, x, n, qp.a, qp.d,
                           sym, clipped);
    }
    return check_hip(hipGetLastError(), "quantize kernel launch");
}

int check_common(const ecgb_tokenizer *tok, size_t batch, size_t n, const void *ids, size_t ids_stride,
                 const void *counts, const void *scratch, size_t scratch_bytes, const char *who)
{
    if (!tok || !ids || !counts || !scratch || ids_stride == 0) {
        ecgb::set_error(std::string(who) + ": NULL or zero-sized argument");
        return ECGB_ERR_INVALID;
    }
    if (!tok->nodes_dev) {
        ecgb::set_error(std::string(who) + ": tokenizer handle has no device copy (no GPU at creation)");
        return ECGB_ERR_NODEVICE;
    }
    if (n >= 0x7FFFFFFFull || batch >= 0x7FFFFFFFull) {
        ecgb::set_error(std::string(who) + ": stream longer than 2^31-1 symbols or batch too large");
        return ECGB_ERR_UNSUPPORTED;
    }
    if (scratch_bytes < ecgb_encode_scratch_bytes(tok, batch, n)) {
        ecgb::set_error(std::string(who) + ": scratch buffer smaller than ecgb_encode_scratch_bytes()");
        return ECGB_ERR_INVALID;
    }
    return ECGB_OK;
}

}  // namespace

extern "C" int ecgb_quantize_hip(const double *signal_dev, size_t n, double percentile_1,
                                 double percentile_99, uint8_t *sym_dev, double *clipped_dev,
                                 void *stream)
{
    if (n == 0) return ECGB_OK;
    if (!signal_dev || !sym_dev) { ecgb::set_error("ecgb_quantize_hip: NULL argument"); return ECGB_ERR_INVALID; }
    return launch_quantize(signal_dev, n, percentile_1, percentile_99, sym_dev, clipped_dev,
                           (hipStream_t)stream);
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

extern "C" size_t ecgb_encode_scratch_bytes(const ecgb_tokenizer *, size_t batch, size_t n_per_stream)
{
    const size_t half_stride = (n_per_stream + 1) / 2;
    return align_up(batch * n_per_stream) + align_up(batch * half_stride * 2) + kAlign;
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
    uint8_t *cls = reinterpret_cast<uint8_t *>(align_up(reinterpret_cast<uintptr_t>(scratch_dev)));
    uint16_t *half = reinterpret_cast<uint16_t *>(cls + align_up(batch * n_per_stream));
    const size_t total = batch * n_per_stream;
    hipLaunchKernelGGL(classify_kernel, dim3(stream_grid(total)), dim3(256), 0, st, text_dev, total,
                       tok->lut_dev, cls);
    rc = check_hip(hipGetLastError(), "classify_kernel launch");
    if (rc) return rc;
    return launch_encode(tok, cls, text_dev, batch, n_per_stream, half, (n_per_stream + 1) / 2, ids_dev,
                         ids_stride, counts_dev, st);
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
    uint8_t *cls = reinterpret_cast<uint8_t *>(align_up(reinterpret_cast<uintptr_t>(scratch_dev)));
    uint16_t *half = reinterpret_cast<uint16_t *>(cls + align_up(batch * n_per_record));
    rc = launch_quantize(signal_dev, batch * n_per_record, percentile_1, percentile_99, cls, nullptr, st);
    if (rc) return rc;
    return launch_encode(tok, cls, nullptr, batch, n_per_record, half, (n_per_record + 1) / 2, ids_dev,
                         ids_stride, counts_dev, st);
}
