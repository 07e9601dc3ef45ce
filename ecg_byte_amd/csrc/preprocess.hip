// preprocess.hip -- the reference's offline ECG conditioning on MI355X (gfx950), float64 throughout:
//   ecgb_filtfilt_f64        advanced_ecg_filter   ecg_byte/utils/preprocess_utils.py:66-88   (scipy.signal.filtfilt chains)
//   ecgb_resample_cubic_f64  nsample_ecg           ecg_byte/utils/preprocess_utils.py:90-101  (scipy interp1d kind='cubic')
//   ecgb_wavelet_denoise_f64 wavelet_denoise       ecg_byte/utils/preprocess_utils.py:43-64   (pywt wavedec / threshold / waverec, db6)
// One-time work in the reference (12 worker processes, ~1 ms per filter and record); here a whole batch of records per launch.
//
// Every stage is a recursion along time (IIR state, tridiagonal sweep, nothing to tile), so the parallel axis is the SEQUENCE: one lane =
// one (record, lead) time series, 49 152 lanes for 4 096 twelve-lead records.  The records arrive as [record][time][lead] (what
// wfdb.rdsamp returns, preprocess_utils.py:126); intermediates live TIME-MAJOR PER WAVE, [wave][time][64 lanes]: the 64 lanes of a wave touch
// 512 contiguous bytes at every step and a wave's consecutive steps are consecutive 512-byte rows (one stream per wave for DRAM pages and the TLB;
// the first layout, [time][all sequences], put a wave's consecutive steps 393 KB apart).  Arithmetic follows the reference's libraries operation by operation where that is what fixes
// the bits (scipy's direct-form-II-transposed loop, its odd extension and initial conditions); -ffp-contract=off keeps a*b+c two roundings.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>
#include <type_traits>

#include "tokenizer.hpp"

namespace {

constexpr int kMaxTaps = 9;          // butter(4, band) is 8th order: 9 coefficients
constexpr int kMaxFilters = 4;

struct Filt {
    int nb;                          // coefficients of b and a (equal lengths, a[0] == 1: scipy normalises by a[0] before the loop)
    int edge;                        // 3 * max(len a, len b): filtfilt's default padlen
    double b[kMaxTaps], a[kMaxTaps], zi[kMaxTaps - 1];   // zi = scipy.signal.lfilter_zi(b, a): steady state of a unit step
};

struct FiltfiltArgs {
    const double *x;                 // [R, n, L]
    double *y;                       // [R, n, L]
    double *ext;                     // scratch [wave][n + 2 * max edge][64]: the forward pass's output over the extended signal
    double *mid;                     // scratch [wave][n][64]: a filter's result, the next filter's input
    int R, n, L, n_filters, ext_rows;   // ext_rows = n + 2 * max edge
    unsigned char *flags;            // optional [R]: set to 1 where a record's result holds a value that is not finite
    unsigned char *raw_flags;        // optional [R]: set to 1 where the record itself (x) holds one: process_instance's test of the raw record, preprocess_utils.py:134-136
    Filt f[kMaxFilters];
};

// scipy.signal._signaltools.filtfilt (method 'pad', padtype 'odd') around scipy's lfilter (sigtools DOUBLE_filt):
//   ext = [2 x[0] - x[e..1], x, 2 x[n-1] - x[n-2..n-1-e]];  y1 = lfilter(ext, z = zi * ext[0]);  y2 = lfilter(reverse(y1), z = zi * y1[-1]);
//   result = reverse(y2)[e : e + n]
// lfilter step:  y = z[0] + b[0] x;  z[k] = z[k+1] + x b[k+1] - y a[k+1]  (k < nb - 2);  z[nb-2] = x b[nb-1] - y a[nb-1].
// Round 3: the loop is the same recursion sample by sample (the same bits), but the SAMPLES move in blocks of kBlk: the next block's kBlk loads are issued
// (independent of each other and of the recursion) before the current block is filtered, and a block's results are stored together.  The first version loaded
// one sample, filtered it and stored it: hipcc cannot move a load above the previous store (`ext`, `mid` and the record may alias for all it knows), so every
// step paid a memory round trip -- 560 cycles per step against ~60 of arithmetic, one wave per SIMD (22.6 ms for 4 096 records).
#ifndef ECGB_PRE_BLK
#define ECGB_PRE_BLK 16
#endif
constexpr int kBlk = ECGB_PRE_BLK;
// The block length is a parameter per filter length (short filters could take longer blocks: two state variables instead of eight).  Measured with 32 for the
// notches: 6.66 ms either way -- the chain is at the HBM traffic of its sweeps, not at the latency of a block's loads -- so both are 16 (24 / 32 spill for the 9-tap filter).
#ifndef ECGB_PRE_BLK_SHORT
#define ECGB_PRE_BLK_SHORT 16
#endif
constexpr int kBlkShort = ECGB_PRE_BLK_SHORT;
constexpr int kBlkMax = kBlk > kBlkShort ? kBlk : kBlkShort;
template <int NB, int BLK, typename SRC, typename DST, typename FLUSH, typename PROBE>
__device__ __forceinline__ void filtfilt_one(const Filt &F, int n, SRC src, double *ext, size_t S, size_t seq, DST dst, FLUSH flush, PROBE probe)
{
    const int e = F.edge, N = n + 2 * e;
    double z[NB - 1];
    // probe(v) sees every sample of the source once it is USED (the first filter tests the raw record with it; a test at the load would wait for each load where
    // it is issued and undo the prefetch: 6.7 -> 8.0 ms): here for the samples that come through ext_at (blocks that touch the padding), below for whole blocks
    const double first = src(0), last = src(n - 1);
    auto ext_at = [&](int i) -> double {
        if (i < e) return 2.0 * first - src(e - i);
        if (i < e + n) { const double v = src(i - e); probe(v); return v; }
        return 2.0 * last - src(n - 2 - (i - e - n));
    };
    auto step = [&](double xi) -> double {
        const double y = z[0] + F.b[0] * xi;
#pragma unroll
        for (int k = 0; k < NB - 2; ++k) z[k] = z[k + 1] + xi * F.b[k + 1] - y * F.a[k + 1];
        z[NB - 2] = xi * F.b[NB - 1] - y * F.a[NB - 1];
        return y;
    };
    const double x0 = ext_at(0);
#pragma unroll
    for (int k = 0; k < NB - 1; ++k) z[k] = F.zi[k] * x0;
    double cur[BLK], nxt[BLK];
    // Every condition below is the same in all lanes (the lanes of a wave are at the same sample), and almost every block lies inside the signal: whole blocks take
    // straight-line code -- BLK loads at constant offsets, BLK steps, BLK stores in one basic block, so the scheduler can start a step's independent products under
    // the previous step's dependent chain -- and only the blocks that touch the padding or the end carry per-sample tests (written with the tests on every sample, the
    // loop was ~450 basic blocks per filter and 5 scalar instructions per load).
    // ---- forward over the extended signal -> ext
#pragma unroll
    for (int u = 0; u < BLK; ++u) cur[u] = ext_at(min(u, N - 1));
    bool cur_plain = false;                                  // cur holds samples of the source as loaded (a whole block inside the signal)
    for (int i0 = 0; i0 < N; i0 += BLK) {
        const int j0 = i0 + BLK;
        const bool nxt_plain = j0 >= e && j0 + BLK <= e + n;
        if (nxt_plain) {
#pragma unroll
            for (int u = 0; u < BLK; ++u) nxt[u] = src(j0 - e + u);                        // the next block's loads fly while this one is filtered
        } else {
#pragma unroll
            for (int u = 0; u < BLK; ++u) nxt[u] = ext_at(min(j0 + u, N - 1));
        }
        double out[BLK];
        if (i0 + BLK <= N) {
            if (cur_plain) {
#pragma unroll
                for (int u = 0; u < BLK; ++u) probe(cur[u]);
            }
#pragma unroll
            for (int u = 0; u < BLK; ++u) out[u] = step(cur[u]);
#pragma unroll
            for (int u = 0; u < BLK; ++u) ext[(size_t)(i0 + u) * S + seq] = out[u];
        } else {
#pragma unroll
            for (int u = 0; u < BLK; ++u) if (i0 + u < N) out[u] = step(cur[u]);
#pragma unroll
            for (int u = 0; u < BLK; ++u) if (i0 + u < N) ext[(size_t)(i0 + u) * S + seq] = out[u];
        }
#pragma unroll
        for (int u = 0; u < BLK; ++u) cur[u] = nxt[u];
        cur_plain = nxt_plain;
    }
    // ---- backward over ext -> dst (the middle n samples)
    const double y0 = ext[(size_t)(N - 1) * S + seq];
#pragma unroll
    for (int k = 0; k < NB - 1; ++k) z[k] = F.zi[k] * y0;
    auto rev_at = [&](int i) -> double { return ext[(size_t)(N - 1 - min(i, N - 1)) * S + seq]; };
#pragma unroll
    for (int u = 0; u < BLK; ++u) cur[u] = rev_at(u);
    for (int i0 = 0; i0 < N; i0 += BLK) {
        const int j0 = i0 + BLK;
        if (j0 + BLK <= N) {
#pragma unroll
            for (int u = 0; u < BLK; ++u) nxt[u] = ext[(size_t)(N - 1 - j0 - u) * S + seq];
        } else {
#pragma unroll
            for (int u = 0; u < BLK; ++u) nxt[u] = rev_at(j0 + u);
        }
        double out[BLK];
        const int top = N - 1 - i0 - e;                      // the block's positions: this one downwards
        if (i0 + BLK <= N && top < n && top - (BLK - 1) >= 0) {
#pragma unroll
            for (int u = 0; u < BLK; ++u) out[u] = step(cur[u]);
#pragma unroll
            for (int u = 0; u < BLK; ++u) dst(top - u, out[u], u, true);
        } else {
#pragma unroll
            for (int u = 0; u < BLK; ++u) if (i0 + u < N) out[u] = step(cur[u]);
#pragma unroll
            for (int u = 0; u < BLK; ++u) {
                const int pos = top - u;
                dst(pos, out[u], u, i0 + u < N && pos >= 0 && pos < n);
            }
        }
        flush(top, std::integral_constant<int, BLK>{});
#pragma unroll
        for (int u = 0; u < BLK; ++u) cur[u] = nxt[u];
    }
}

template <typename SRC, typename DST, typename FLUSH, typename PROBE>
__device__ __forceinline__ void filtfilt_dispatch(const Filt &F, int n, SRC src, double *ext, size_t S, size_t seq, DST dst, FLUSH flush, PROBE probe)
{
    switch (F.nb) {
    case 2: filtfilt_one<2, kBlkShort>(F, n, src, ext, S, seq, dst, flush, probe); break;
    case 3: filtfilt_one<3, kBlkShort>(F, n, src, ext, S, seq, dst, flush, probe); break;
    case 4: filtfilt_one<4, kBlkShort>(F, n, src, ext, S, seq, dst, flush, probe); break;
    case 5: filtfilt_one<5, kBlk>(F, n, src, ext, S, seq, dst, flush, probe); break;
    case 6: filtfilt_one<6, kBlk>(F, n, src, ext, S, seq, dst, flush, probe); break;
    case 7: filtfilt_one<7, kBlk>(F, n, src, ext, S, seq, dst, flush, probe); break;
    case 8: filtfilt_one<8, kBlk>(F, n, src, ext, S, seq, dst, flush, probe); break;
    default: filtfilt_one<9, kBlk>(F, n, src, ext, S, seq, dst, flush, probe); break;
    }
}

// PLANAR: the last filter's result goes out sequence by sequence, y[sequence][time] -- the layout the workgroup-per-sequence wavelet kernel reads with whole
// lines.  A lane's results are consecutive in time, 8 bytes a step; written lane by lane they would be 64 partial lines per store instruction, so a block of
// kBlk steps x 64 lanes crosses LDS and leaves as rows of kBlk consecutive doubles per sequence (whole lines).
template <bool PLANAR>
__global__ __launch_bounds__(64) void filtfilt_kernel(FiltfiltArgs A)
{
    __shared__ double tile[PLANAR ? kBlkMax * 65 : 1];
    const size_t S = (size_t)A.R * A.L;
    const size_t seq = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (seq >= S) return;
    const bool full_wave = ((size_t)blockIdx.x + 1) * 64 <= S;                    // (a partial last wave stores lane by lane)
    const int lane = threadIdx.x;
    const size_t r = seq / A.L, l = seq % A.L;
    const double *xin = A.x + (r * A.n) * A.L + l;
    double *yout = PLANAR ? A.y + seq * (size_t)A.n : A.y + (r * A.n) * A.L + l;
    const int L = A.L, n = A.n;
    double *ext = A.ext + (size_t)blockIdx.x * A.ext_rows * 64 + threadIdx.x;     // this wave's rows, this lane's column
    double *mid = A.mid + (size_t)blockIdx.x * A.n * 64 + threadIdx.x;
    bool bad = false;                                                             // a value that is not finite went out (flags: check_nan_inf's test)
    bool raw_bad = false;                                                         // ... or came in (the first filter's forward pass reads every sample of x)
    for (int k = 0; k < A.n_filters; ++k) {
        const bool first = k == 0, lastf = k == A.n_filters - 1;
        auto src_x = [&](int t) -> double { return xin[(size_t)t * L]; };
        auto probe_x = [&](double v) { raw_bad |= !isfinite(v); };
        auto no_probe = [](double) {};
        auto src_m = [&](int t) -> double { return mid[(size_t)t * 64]; };
        auto dst_y = [&](int t, double v, int u, bool ok) {
            if (ok) bad |= !isfinite(v);
            if (PLANAR) {
                if (full_wave) tile[u * 65 + lane] = v;
                else if (ok) yout[t] = v;
            } else if (ok) yout[(size_t)t * L] = v;
        };
        auto dst_m = [&](int t, double v, int, bool ok) { if (ok) mid[(size_t)t * 64] = v; };
        auto no_flush = [](int, auto) {};
        auto flush_y = [&](int top, auto blk_c) {                                    // the tile holds steps u = 0 .. blk - 1 of every lane: positions top - u
            if (!PLANAR || !full_wave) return;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            double *y0 = A.y + ((size_t)blockIdx.x * 64) * (size_t)n;
            constexpr int blk = decltype(blk_c)::value;
#pragma unroll 4
            for (int q = 0; q < blk; ++q) {
                const int idx = q * 64 + lane, sq = idx / blk, u = idx % blk, pos = top - u;
                if (pos >= 0 && pos < n) y0[(size_t)sq * n + pos] = tile[u * 65 + sq];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        };
        // a filter reads all of its input before the backward pass writes the first output sample, so `mid` can be source and
        // destination of the same filter
        if (first && lastf) filtfilt_dispatch(A.f[k], A.n, src_x, ext, 64, 0, dst_y, flush_y, probe_x);
        else if (first) filtfilt_dispatch(A.f[k], A.n, src_x, ext, 64, 0, dst_m, no_flush, probe_x);
        else if (lastf) filtfilt_dispatch(A.f[k], A.n, src_m, ext, 64, 0, dst_y, flush_y, no_probe);
        else filtfilt_dispatch(A.f[k], A.n, src_m, ext, 64, 0, dst_m, no_flush, no_probe);
    }
    if (A.flags && bad) A.flags[r] = 1;
    if (A.raw_flags && raw_bad) A.raw_flags[r] = 1;
}

// ---- cubic-spline resampling -----------------------------------------------------------------------------------------------------------
// scipy.interpolate.interp1d(t, y, kind='cubic') is the not-a-knot cubic spline through all n samples (make_interp_spline, k = 3, default
// boundary conditions), evaluated at m equally spaced targets over the same span (preprocess_utils.py:90-101).  The interpolant does not
// depend on how it is computed, so instead of scipy's banded B-spline solve this is the classic second-derivative form on the unit grid:
//   M[i-1] + 4 M[i] + M[i+1] = 6 (y[i+1] - 2 y[i] + y[i-1])            1 <= i <= n-2
//   not-a-knot:  M[0] = 2 M[1] - M[2],  M[n-1] = 2 M[n-2] - M[n-3]      => rows 1 and n-2 collapse to 6 M[1] = r[1], 6 M[n-2] = r[n-2]
// one Thomas sweep per sequence over rows 2 .. n-3 (the pivots 1 / (4 - c') are the same for every sequence and reach their fixed point
// 2 - sqrt(3) in double precision within ~20 rows), then  S(i + u) = y[i] + u (dy - (2 M[i] + M[i+1]) / 6) + u^2 M[i] / 2 + u^3 (M[i+1] - M[i]) / 6.
// Agreement with scipy: the two solves differ by rounding only (1e-13 of the signal's range measured, tests/test_gpu_preprocess.py).
constexpr int kMaxLeadMap = 32, kResampleTab = 32;
constexpr int kFuseMin = 64;      // shorter signals: back substitution and evaluation as two sweeps (the first version; n = 4 .. 7 have their own closed forms)
struct ResampleArgs {
    const double *x;                 // [R, n, L], or [R * L][n] (PLANAR_IN)
    double *y;                       // [R, m, L]
    double *M;                       // scratch [wave][n][64]
    int R, n, L, m;
    unsigned char *flags;            // optional [R]: 1 where a record's result holds a value that is not finite
    int use_map;
    double cp[kResampleTab];         // the Thomas sweep's pivots 1 / (4 - c'): the same for every sequence
    unsigned char out_lead[kMaxLeadMap];   // use_map: lead l of the input is lead out_lead[l] of the output (the MIMIC reorder, preprocess_utils.py:35-40)
};

template <bool PLANAR_IN>
__global__ __launch_bounds__(64) void resample_cubic_kernel(ResampleArgs A)
{
    const size_t S = (size_t)A.R * A.L;
    const size_t seq = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (seq >= S) return;
    const size_t r = seq / A.L, l = seq % A.L;
    const int n = A.n, m = A.m, L = A.L;
    const double *x = PLANAR_IN ? A.x + seq * (size_t)n : A.x + (r * n) * L + l;
    double *out = A.y + (r * m) * L + (A.use_map ? (size_t)A.out_lead[l] : l);
    double *M = A.M + (size_t)blockIdx.x * n * 64 + threadIdx.x;
    auto Y = [&](int i) -> double { return PLANAR_IN ? x[i] : x[(size_t)i * L]; };
    bool bad = false;
    auto rhs = [&](int i) -> double { return 6.0 * (Y(i + 1) - 2.0 * Y(i) + Y(i - 1)); };
    // c'[2 + k] = A.cp[k] (from the host: the same IEEE divisions; a row index is the same in every lane, so the pivot is one scalar load -- a table built in the
    // kernel and indexed by the row lived in scratch memory); c'[2 + k] = c'[2 + kTab - 1] beyond the table (fixed point)
    auto cprime = [&](int i) -> double { const int k = i - 2; return A.cp[k < kResampleTab ? k : kResampleTab - 1]; };
    const double M1 = rhs(1) / 6.0, Mn2 = rhs(n - 2) / 6.0;
    M[(size_t)1 * 64] = M1;
    M[(size_t)(n - 2) * 64] = Mn2;
    // Round 3: the sweeps and the evaluation move their samples in blocks of kBlk (loads of a block issued together, results stored together, the next
    // block's loads in flight meanwhile): same arithmetic in the same order, no memory round trip per step (see filtfilt_one).
    if (n >= 6) {                                            // rows 2 .. n-3
        double dprev = (rhs(2) - M1) * cprime(2);
        M[(size_t)2 * 64] = dprev;
        auto ycl = [&](int i) -> double { return Y(min(max(i, 0), n - 1)); };
        double yc[kBlk + 2], yn[kBlk + 2];
#pragma unroll
        for (int u = 0; u < kBlk + 2; ++u) yc[u] = ycl(3 - 1 + u);          // rows 3 .. 3 + kBlk - 1 need Y(2) .. Y(3 + kBlk)
        const double c_inf = A.cp[kResampleTab - 1];         // the pivots' fixed point (rows past the table)
        for (int i0 = 3; i0 <= n - 3; i0 += kBlk) {
            // (whole blocks inside the signal and past the pivot table: straight-line code, as in filtfilt_one)
            if (i0 + 2 * kBlk + 1 <= n - 1) {
#pragma unroll
                for (int u = 0; u < kBlk + 2; ++u) yn[u] = Y(i0 + kBlk - 1 + u);
            } else {
#pragma unroll
                for (int u = 0; u < kBlk + 2; ++u) yn[u] = ycl(i0 + kBlk - 1 + u);
            }
            double out[kBlk];
            if (i0 - 2 >= kResampleTab - 1 && i0 + kBlk - 1 <= n - 4) {
#pragma unroll
                for (int u = 0; u < kBlk; ++u) {
                    const double ri = 6.0 * (yc[u + 2] - 2.0 * yc[u + 1] + yc[u]);
                    dprev = (ri - dprev) * c_inf;
                    out[u] = dprev;
                }
#pragma unroll
                for (int u = 0; u < kBlk; ++u) M[(size_t)(i0 + u) * 64] = out[u];
            } else {
#pragma unroll
                for (int u = 0; u < kBlk; ++u) {
                    const int i = i0 + u;
                    if (i <= n - 3) {
                        double ri = 6.0 * (yc[u + 2] - 2.0 * yc[u + 1] + yc[u]);
                        if (i == n - 3) ri -= Mn2;
                        dprev = (ri - dprev) * cprime(i);    // cprime(i) = 1 / (4 - c'[i-1])
                        out[u] = dprev;
                    }
                }
#pragma unroll
                for (int u = 0; u < kBlk; ++u) if (i0 + u <= n - 3) M[(size_t)(i0 + u) * 64] = out[u];
            }
#pragma unroll
            for (int u = 0; u < kBlk + 2; ++u) yc[u] = yn[u];
        }
        double Mnext = dprev;                                // M[n-3]
        double mc[kBlk], mn[kBlk];
        auto mcl = [&](int i) -> double { return M[(size_t)max(i, 2) * 64]; };
        if (n >= kFuseMin) {
            // ---- back substitution and evaluation in ONE descending sweep: the spline on [i, i + 1] needs M[i] and M[i + 1], and the back substitution hands the
            // M's out from the top down -- so every block of rows is evaluated right where it is solved, from a per-lane column of an LDS tile (a row index is the same
            // in every lane; registers cannot be indexed by it), and M is never written to memory or read back: 36 bytes of traffic per sample instead of 52.  Each output
            // is the expression it was (same operands, same order); only the order in which the outputs are produced changes.
            __shared__ double mt[(kBlk + 2) * 64];
            double *col = mt + threadIdx.x;
            int jc = m - 1;                                  // next output, descending
            auto interval_of = [&](int j, double &sx) -> int {
                sx = m > 1 ? (double)j * (double)(n - 1) / (double)(m - 1) : 0.0;
                int i = (int)sx;
                return i > n - 2 ? n - 2 : i;
            };
            // outputs jc, jc - 1, ... whose interval is >= lo; M[i] is col[(i - base) * 64]
            auto eval_down = [&](int lo, int base) {
                constexpr int kE = 8;
                for (;;) {
                    double y0[kE], y1[kE], m0[kE], m1[kE], uu[kE];
                    int cnt = 0;
#pragma unroll
                    for (int u = 0; u < kE; ++u) {
                        double sx;
                        const int j = max(jc - u, 0);
                        const int i = interval_of(j, sx);
                        const bool ok = jc - u >= 0 && i >= lo;          // (intervals do not grow as j falls: the valid outputs are a prefix)
                        cnt += ok ? 1 : 0;
                        const int ic = max(i, lo);
                        uu[u] = sx - (double)i;
                        y0[u] = Y(ic); y1[u] = Y(ic + 1); m0[u] = col[(size_t)(ic - base) * 64]; m1[u] = col[(size_t)(ic + 1 - base) * 64];
                    }
#pragma unroll
                    for (int u = 0; u < kE; ++u) {
                        if (u >= cnt) continue;
                        const double c1 = (y1[u] - y0[u]) - (2.0 * m0[u] + m1[u]) / 6.0, c2 = 0.5 * m0[u], c3 = (m1[u] - m0[u]) / 6.0;
                        const double v = y0[u] + uu[u] * (c1 + uu[u] * (c2 + uu[u] * c3));
                        bad |= !isfinite(v);
                        out[(size_t)(jc - u) * L] = v;
                    }
                    jc -= cnt;
                    if (cnt < kE) break;
                }
            };
            // intervals n - 2 and n - 3: M[n-3] (the forward sweep's last value), M[n-2], M[n-1] = 2 M[n-2] - M[n-3]
            col[0] = dprev; col[64] = Mn2; col[128] = 2.0 * Mn2 - dprev;
            eval_down(n - 3, n - 3);
#pragma unroll
            for (int u = 0; u < kBlk; ++u) mc[u] = mcl(n - 4 - u);
            for (int i0 = n - 4; i0 >= 2; i0 -= kBlk) {
#pragma unroll
                for (int u = 0; u < kBlk; ++u) mn[u] = mcl(i0 - kBlk - u);
                const int base = i0 - kBlk + 1;
                col[(size_t)kBlk * 64] = Mnext;              // M[i0 + 1]
#pragma unroll
                for (int u = 0; u < kBlk; ++u) {
                    const int i = i0 - u;
                    if (i >= 2) { Mnext = mc[u] - cprime(i) * Mnext; col[(size_t)(kBlk - 1 - u) * 64] = Mnext; }
                }
                eval_down(max(base, 2), base);
#pragma unroll
                for (int u = 0; u < kBlk; ++u) mc[u] = mn[u];
            }
            // intervals 1 and 0: M[2] (the last value of the sweep), M[1], M[0] = 2 M[1] - M[2]
            col[128] = Mnext; col[64] = M1; col[0] = 2.0 * M1 - Mnext;
            eval_down(0, 0);
            if (A.flags && bad) A.flags[r] = 1;
            return;
        }
#pragma unroll
        for (int u = 0; u < kBlk; ++u) mc[u] = mcl(n - 4 - u);
        for (int i0 = n - 4; i0 >= 2; i0 -= kBlk) {
#pragma unroll
            for (int u = 0; u < kBlk; ++u) mn[u] = mcl(i0 - kBlk - u);     // rows below this block: not written by it
            double out[kBlk];
#pragma unroll
            for (int u = 0; u < kBlk; ++u) {
                const int i = i0 - u;
                if (i >= 2) { Mnext = mc[u] - cprime(i) * Mnext; out[u] = Mnext; }
            }
#pragma unroll
            for (int u = 0; u < kBlk; ++u) if (i0 - u >= 2) M[(size_t)(i0 - u) * 64] = out[u];
#pragma unroll
            for (int u = 0; u < kBlk; ++u) mc[u] = mn[u];
        }
    } else if (n == 5) {
        M[(size_t)2 * 64] = (rhs(2) - M1 - Mn2) / 4.0;
    }
    M[0] = 2.0 * M[(size_t)1 * 64] - M[(size_t)2 * 64];
    M[(size_t)(n - 1) * 64] = 2.0 * M[(size_t)(n - 2) * 64] - M[(size_t)(n - 3) * 64];
    for (int j0 = 0; j0 < m; j0 += kBlk) {
        double y0[kBlk], y1[kBlk], m0[kBlk], m1[kBlk], uu[kBlk];
#pragma unroll
        for (int u = 0; u < kBlk; ++u) {
            const int j = min(j0 + u, m - 1);
            const double sx = m > 1 ? (double)j * (double)(n - 1) / (double)(m - 1) : 0.0;
            int i = (int)sx;
            if (i > n - 2) i = n - 2;
            uu[u] = sx - (double)i;
            y0[u] = Y(i); y1[u] = Y(i + 1); m0[u] = M[(size_t)i * 64]; m1[u] = M[(size_t)(i + 1) * 64];
        }
#pragma unroll
        for (int u = 0; u < kBlk; ++u) {
            if (j0 + u >= m) continue;
            const double c1 = (y1[u] - y0[u]) - (2.0 * m0[u] + m1[u]) / 6.0, c2 = 0.5 * m0[u], c3 = (m1[u] - m0[u]) / 6.0;
            const double v = y0[u] + uu[u] * (c1 + uu[u] * (c2 + uu[u] * c3));
            bad |= !isfinite(v);
            out[(size_t)(j0 + u) * L] = v;
        }
    }
    if (A.flags && bad) A.flags[r] = 1;
}

// ---- wavelet denoising ---------------------------------------------------------------------------------------------------------------------
// preprocess_utils.py:43-64: coeffs = pywt.wavedec(x, 'db6', level=4) (mode 'symmetric'); threshold = median(|cD4|) / 0.6745 (0 when the
// median is 0); every detail band soft-thresholded (pywt.threshold: c * max(0, 1 - thr / |c|)) and zeroed where not finite or |c| <= 1e-10;
// pywt.waverec; NaN / inf -> 0.  PyWavelets is not installed anywhere this code runs: restated from its published algorithm, PARITY UNPINNED
// (tests compare with oracle/preprocess_ref.py, an independent numpy restatement, and check the transform's own identities).
//   analysis   cA[o] = sum_j dec_lo[j] xe[2 o + 1 - j],  cD likewise with dec_hi,  o < (N + 11) / 2;  xe = half-point symmetric extension
//   synthesis  x[t]  = sum_o cA[o] dec_lo[2 o + 1 - t] + sum_o cD[o] dec_hi[2 o + 1 - t],  t < 2 Nc - 10; a band one sample longer than its
//              detail band loses its last sample first (waverec)
__constant__ double kDb6Lo[12] = {-0.00107730108499558, 0.004777257511010651, 0.0005538422009938016, -0.031582039318031156,
                                  0.02752286553001629, 0.09750160558707936, -0.12976686756709563, -0.22626469396516913,
                                  0.3152503517092432, 0.7511339080215775, 0.4946238903983854, 0.11154074335008017};
constexpr int kWF = 12, kLevels = 4;

struct WaveletArgs {
    const double *x;                 // [R, n, L]
    double *y;                       // [R, n, L]
    double *work;                    // scratch [wave][rows][64]: per level l (1..4) bands A_l and D_l of len[l] + 2 rows
    long long rows;                  // rows per wave
    int R, n, L;
    int len[kLevels + 1];            // len[0] = n, len[l] = (len[l-1] + 11) / 2
    long long offA[kLevels + 1], offD[kLevels + 1];   // row offsets into `work`
    double epsilon;
};

// pywt.threshold(data, value, 'soft') = data * clip(1 - value / |data|, 0) followed by the reference's guard (preprocess_utils.py:57-59: zero where the result is
// not finite or |data| <= epsilon).  The division is the expensive part of the stage (a float64 division is ~40 instructions around v_rcp_f64: 1.5 of the workgroup
// kernel's 3.5 ms went here) and most detail coefficients lie under the threshold, where the quotient is >= 1 exactly when value >= |data| (IEEE division is monotone and
// exact at 1), the clip gives 0 and the product is data * 0 (its sign kept): those take no division.  Written `mag <= thr` so that a NaN on either side takes the general
// path and comes out as it always did (0 / 0, inf / inf and a NaN threshold included).
__device__ __forceinline__ double soft_threshold(double c, double mag, double thr, double epsilon)
{
    double t;
    if (mag <= thr) {
        t = c * 0.0;
    } else {
        double f = 1.0 - thr / mag;
        f = f < 0.0 ? 0.0 : f;                               // (NaN stays NaN and is zeroed by the test below)
        t = c * f;
    }
    return (isfinite(t) && mag > epsilon) ? t : 0.0;
}

__global__ __launch_bounds__(64) void wavelet_denoise_kernel(WaveletArgs A)
{
    const size_t S = (size_t)A.R * A.L;
    const size_t seq = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (seq >= S) return;
    const size_t r = seq / A.L, l0 = seq % A.L;
    const int L = A.L, n = A.n;
    const double *x = A.x + (r * n) * L + l0;
    double *out = A.y + (r * n) * L + l0;
    double *W = A.work + (size_t)blockIdx.x * A.rows * 64 + threadIdx.x;
    auto band = [&](long long off, int i) -> double & { return W[(size_t)(off + i) * 64]; };
    double lo[kWF], hi[kWF];
#pragma unroll
    for (int k = 0; k < kWF; ++k) { lo[k] = kDb6Lo[k]; hi[k] = ((k & 1) ? 1.0 : -1.0) * kDb6Lo[kWF - 1 - k]; }
    // Round 3: every loop below moves its samples in blocks (the loads of a block together, its results stored together): the sums are formed in the
    // order they had, so the values are the first version's; what is gone is one memory round trip per coefficient (see filtfilt_one) and three quarters of
    // the median's selection passes.
#ifndef ECGB_PRE_OB
#define ECGB_PRE_OB 8
#endif
    constexpr int kOB = ECGB_PRE_OB;                         // outputs per block
    // ---- wavedec: cA[o], cD[o] = sum_j f[j] xe[2 o + 1 - j]: a block of kOB outputs reads xe[2 o0 - 10 .. 2 (o0 + kOB - 1) + 1]
    for (int lev = 1; lev <= kLevels; ++lev) {
        const int N = A.len[lev - 1], Nc = A.len[lev];
        auto in_at = [&](int i) -> double {                  // symmetric extension: ... x1 x0 | x0 x1 ... x[N-1] | x[N-1] x[N-2] ...
            if (i < 0) i = -1 - i;
            else if (i >= N) i = 2 * N - 1 - i;
            i = min(max(i, 0), N - 1);                       // (blocks past the band's end read a valid sample and drop the result)
            return lev == 1 ? x[(size_t)i * L] : band(A.offA[lev - 1], i);
        };
        for (int o0 = 0; o0 < Nc; o0 += kOB) {
            double w[2 * kOB + kWF - 2];
#pragma unroll
            for (int u = 0; u < 2 * kOB + kWF - 2; ++u) w[u] = in_at(2 * o0 - (kWF - 2) + u);      // w[u] = xe[2 o0 - 10 + u]
            double sa[kOB], sd[kOB];
#pragma unroll
            for (int q = 0; q < kOB; ++q) {
                double a = 0.0, d = 0.0;
#pragma unroll
                for (int j = 0; j < kWF; ++j) {
                    const double v = w[2 * q + (kWF - 1) - j];                                       // xe[2 (o0 + q) + 1 - j]
                    a += lo[j] * v;
                    d += hi[j] * v;
                }
                sa[q] = a; sd[q] = d;
            }
#pragma unroll
            for (int q = 0; q < kOB; ++q)
                if (o0 + q < Nc) { band(A.offA[lev], o0 + q) = sa[q]; band(A.offD[lev], o0 + q) = sd[q]; }
        }
    }
    // ---- threshold from the median of |cD4|: the two middle order statistics by a radix select on the bit patterns (monotone for x >= 0), four bits a pass
    // (sixteen counters in registers: one sweep of the band settles four bits instead of one -- 16 sweeps per statistic instead of 63)
    const int n4 = A.len[kLevels];
    auto kth = [&](int k) -> double {
        unsigned long long prefix = 0;
        for (int shift = 60; shift >= 0; shift -= 4) {
            const unsigned long long mask = shift == 60 ? 0ull : ~((1ull << (shift + 4)) - 1);       // the bits already settled
            int cnt[16];
#pragma unroll
            for (int d = 0; d < 16; ++d) cnt[d] = 0;
            for (int i0 = 0; i0 < n4; i0 += kOB) {
                unsigned long long u8[kOB];
#pragma unroll
                for (int q = 0; q < kOB; ++q) u8[q] = (unsigned long long)__double_as_longlong(fabs(band(A.offD[kLevels], min(i0 + q, n4 - 1))));
#pragma unroll
                for (int q = 0; q < kOB; ++q) {
                    const bool in = (i0 + q < n4) && ((u8[q] & mask) == prefix);
                    const int dig = (int)((u8[q] >> shift) & 15ull);
#pragma unroll
                    for (int d = 0; d < 16; ++d) cnt[d] += (in && dig == d) ? 1 : 0;
                }
            }
            int dsel = 15;
            bool found = false;
#pragma unroll
            for (int d = 0; d < 16; ++d) {
                if (!found) {
                    if (k < cnt[d]) { dsel = d; found = true; }
                    else k -= cnt[d];
                }
            }
            prefix |= (unsigned long long)dsel << shift;
        }
        return __longlong_as_double((long long)prefix);
    };
    bool any_nan = false;                                    // np.median of a band that holds a NaN is NaN
    for (int i0 = 0; i0 < n4; i0 += kOB) {
        double v[kOB];
#pragma unroll
        for (int q = 0; q < kOB; ++q) v[q] = band(A.offD[kLevels], min(i0 + q, n4 - 1));
#pragma unroll
        for (int q = 0; q < kOB; ++q) any_nan |= isnan(v[q]);
    }
    const double med = any_nan ? __longlong_as_double(0x7ff8000000000000ll) : (n4 & 1) ? kth(n4 / 2) : 0.5 * (kth(n4 / 2 - 1) + kth(n4 / 2));
    const double thr = med == 0.0 ? 0.0 : med / 0.6745;
    for (int lev = 1; lev <= kLevels; ++lev)
        for (int i0 = 0; i0 < A.len[lev]; i0 += kOB) {
            double c[kOB];
#pragma unroll
            for (int q = 0; q < kOB; ++q) c[q] = band(A.offD[lev], min(i0 + q, A.len[lev] - 1));
#pragma unroll
            for (int q = 0; q < kOB; ++q) {
                const double mag = fabs(c[q]);
                c[q] = soft_threshold(c[q], mag, thr, A.epsilon);
            }
#pragma unroll
            for (int q = 0; q < kOB; ++q) if (i0 + q < A.len[lev]) band(A.offD[lev], i0 + q) = c[q];
        }
    // ---- waverec: x[t] = sum_o cA[o] lo[2 o + 1 - t] + cD[o] hi[2 o + 1 - t], o from t / 2 while 2 o + 1 - t < 12: a block of 2 kOB outputs (t0 even)
    // reads bands [t0 / 2 .. t0 / 2 + kOB + 4]
    int cur = A.len[kLevels];                                // length of the running approximation band (stored in A_lev)
    for (int lev = kLevels; lev >= 1; --lev) {
        const int Nd = A.len[lev];
        if (cur == Nd + 1) cur = Nd;                         // waverec drops the extra sample
        const int No = 2 * Nd - kWF + 2;
        for (int t0 = 0; t0 < No; t0 += 2 * kOB) {
            const int ob = t0 / 2;
            double ba[kOB + 5], bd[kOB + 5];
#pragma unroll
            for (int u = 0; u < kOB + 5; ++u) {
                const int o = min(ob + u, Nd - 1);
                ba[u] = band(A.offA[lev], o);
                bd[u] = band(A.offD[lev], o);
            }
            double res[2 * kOB];
#pragma unroll
            for (int q = 0; q < 2 * kOB; ++q) {
                const int t = t0 + q;
                double sa = 0.0, sd = 0.0;
#pragma unroll
                for (int u = 0; u < 6; ++u) {                // o = t / 2 + u, k = 2 o + 1 - t = 2 u + 1 - (t & 1) < 12
                    const int o = t / 2 + u, kk = 2 * u + 1 - (q & 1);
                    if (o < Nd) {
                        sa += ba[q / 2 + u] * lo[kk];
                        sd += bd[q / 2 + u] * hi[kk];
                    }
                }
                res[q] = sa + sd;
            }
#pragma unroll
            for (int q = 0; q < 2 * kOB; ++q) {
                const int t = t0 + q;
                if (t >= No) continue;
                if (lev > 1) band(A.offA[lev - 1], t) = res[q];
                else if (t < n) out[(size_t)t * L] = isfinite(res[q]) ? res[q] : 0.0;
            }
        }
        cur = No;
    }
}

// ---- the same stage with ONE WORKGROUP PER SEQUENCE (the path ecgb_wavelet_denoise_f64 takes whenever a sequence's bands fit in LDS) ------------------------
// The transform is a pair of FIR filters, not a recursion: every coefficient of a level is independent of its neighbours, so the parallel axis can be TIME and the
// sequence can stay on chip.  512 lanes share one sequence: the samples are read once into LDS, the four analysis levels, the median, the shrinkage and three of the
// four synthesis levels run LDS to LDS, and the last synthesis level writes the result -- 16 bytes of HBM traffic per sample where the lane-per-sequence kernel above
// moves ~92 (every band written and read back through its time-major scratch, the median's sweeps).  Each coefficient is the sum the kernel above forms, term by term
// in the same order, and the median is the same order statistic: the results are the same bits (tests/test_gpu_preprocess.py compares the two kernels).
// LDS: [x, later A2 D2 A3 D3 A4 D4 | A1 | D1] -- level 1 reads x and writes A1, D1; from level 2 on x is dead and the small bands take its place; synthesis writes
// A3', A2', A1' where A3, A2, A1 were.  80.1 KB at n = 5000: two workgroups per CU.
// The records arrive [record][time][lead]: a sequence's samples are 8 bytes every 96.  XCD k takes the k-th eighth of the sequences in order, so the twelve leads of a
// record run side by side on one XCD and share the record's lines in its L2 (HBM sees each line once; the 12-fold line traffic is L2 -> CU).
struct WaveletWgArgs {
    const double *x;
    double *y;
    int R, n, L;
    int len[kLevels + 1];
    int offA[kLevels + 1], offD[kLevels + 1];   // LDS offsets in doubles (offA[0] = x)
    double epsilon;
};

// lanes per sequence: 512 (eight waves; two workgroups per CU = four waves per SIMD to overlap the kernel's fourteen short phases: 3.05 ms with 256 lanes, 2.69 with 512,
// 4.31 with 1 024)
#ifndef ECGB_WAVELET_LANES
#define ECGB_WAVELET_LANES 512
#endif
constexpr int kWgLanes = ECGB_WAVELET_LANES;

// PLANAR: x and y are [sequence][time] (what ecgb_filtfilt_planar_f64 writes): the workgroup's loads and stores are whole lines
template <bool PLANAR>
__global__ __launch_bounds__(kWgLanes) void wavelet_denoise_wg_kernel(WaveletWgArgs A)
{
    extern __shared__ __align__(16) double s_w[];
    __shared__ unsigned long long s_key[2 * (kWgLanes / 64)];
    __shared__ int s_nan;
    const size_t S = (size_t)A.R * A.L;
    const size_t per_xcd = (S + 7) / 8;
    const size_t seq = (size_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((size_t)(blockIdx.x >> 3) >= per_xcd || seq >= S) return;
    const size_t r = seq / A.L, l0 = seq % A.L;
    const int L = A.L, n = A.n, tid = threadIdx.x;
    const double *x = PLANAR ? A.x + seq * (size_t)n : A.x + (r * n) * L + l0;
    double *out = PLANAR ? A.y + seq * (size_t)n : A.y + (r * n) * L + l0;
    const size_t stride = PLANAR ? 1 : (size_t)L;
    double lo[kWF], hi[kWF];
#pragma unroll
    for (int k = 0; k < kWF; ++k) { lo[k] = kDb6Lo[k]; hi[k] = ((k & 1) ? 1.0 : -1.0) * kDb6Lo[kWF - 1 - k]; }
    if (tid == 0) s_nan = 0;
    // the samples: kLoadBatch loads per lane in flight before the first is used (n = 5000: one memory round trip for the sequence)
    constexpr int kLoadBatch = (5120 + kWgLanes - 1) / kWgLanes;      // n = 5000 in one batch
    for (int i0 = tid; i0 < n; i0 += kLoadBatch * kWgLanes) {
        double v[kLoadBatch];
#pragma unroll
        for (int u = 0; u < kLoadBatch; ++u) v[u] = x[(size_t)min(i0 + u * kWgLanes, n - 1) * stride];
#pragma unroll
        for (int u = 0; u < kLoadBatch; ++u) if (i0 + u * kWgLanes < n) s_w[i0 + u * kWgLanes] = v[u];
    }
    __syncthreads();
    // ---- wavedec
    for (int lev = 1; lev <= kLevels; ++lev) {
        const int N = A.len[lev - 1], Nc = A.len[lev];
        const double *in = s_w + A.offA[lev - 1];
        double *ca = s_w + A.offA[lev], *cd = s_w + A.offD[lev];
        for (int o = tid; o < Nc; o += kWgLanes) {
            double a = 0.0, d = 0.0;
            if (2 * o - (kWF - 2) >= 0 && 2 * o + 1 < N) {  // the window lies inside the band: twelve reads at constant offsets, no index arithmetic
                const double2 *pw = reinterpret_cast<const double2 *>(in + 2 * o - (kWF - 2));   // (band offsets are even: 16-byte reads, lane after lane)
                double w[kWF];
#pragma unroll
                for (int j = 0; j < kWF / 2; ++j) { const double2 t2 = pw[j]; w[2 * j] = t2.x; w[2 * j + 1] = t2.y; }
#pragma unroll
                for (int j = 0; j < kWF; ++j) {
                    const double v = w[kWF - 1 - j];         // in[2 o + 1 - j]
                    a += lo[j] * v;
                    d += hi[j] * v;
                }
            } else {
#pragma unroll
                for (int j = 0; j < kWF; ++j) {
                    int i = 2 * o + 1 - j;                   // symmetric extension: ... x1 x0 | x0 x1 ... x[N-1] | x[N-1] x[N-2] ...
                    if (i < 0) i = -1 - i;
                    else if (i >= N) i = 2 * N - 1 - i;
                    i = min(max(i, 0), N - 1);
                    const double v = in[i];
                    a += lo[j] * v;
                    d += hi[j] * v;
                }
            }
            ca[o] = a; cd[o] = d;
        }
        __syncthreads();
    }
    // ---- median of |cD4|: the k-th order statistic is the largest value with at most k values below it.  |cD4| goes to LDS once (where A1 was: dead until the
    // synthesis rewrites it), every lane counts the values below its own (n4 broadcast reads) and the candidates meet in a maximum on the bit patterns (monotone
    // for x >= 0): across the wave by shuffles, across the four waves through LDS.  (With one LDS atomicMax per candidate -- a compare-and-swap loop on 64 bits, half
    // the band contending for one address -- this section was 1.3 of the kernel's 3.5 ms.)  The same order statistics as the radix select of the kernel above.
    const int n4 = A.len[kLevels];
    const double *d4 = s_w + A.offD[kLevels];
    double *mag4 = s_w + A.offA[1];
    {
        bool nan_here = false;
        for (int i = tid; i < n4; i += kWgLanes) { const double v = d4[i]; nan_here |= isnan(v); mag4[i] = fabs(v); }
        if (nan_here) s_nan = 1;
    }
    __syncthreads();
    const bool any_nan = s_nan != 0;
    unsigned long long cand_lo = 0ull, cand_hi = 0ull;
    if (!any_nan) {
        const int k_lo = (n4 & 1) ? n4 / 2 : n4 / 2 - 1, k_hi = n4 / 2;
        for (int i = tid; i < n4; i += kWgLanes) {
            const double v = mag4[i];
            int below = 0;
            int j = 0;
            for (; j + 4 <= n4; j += 4) {
                const double u0 = mag4[j], u1 = mag4[j + 1], u2 = mag4[j + 2], u3 = mag4[j + 3];
                below += (u0 < v ? 1 : 0) + (u1 < v ? 1 : 0) + (u2 < v ? 1 : 0) + (u3 < v ? 1 : 0);
            }
            for (; j < n4; ++j) below += mag4[j] < v ? 1 : 0;
            const unsigned long long key = (unsigned long long)__double_as_longlong(v);
            if (below <= k_lo && key > cand_lo) cand_lo = key;
            if (below <= k_hi && key > cand_hi) cand_hi = key;
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const unsigned long long o_lo = __shfl_xor(cand_lo, d, 64), o_hi = __shfl_xor(cand_hi, d, 64);
        cand_lo = o_lo > cand_lo ? o_lo : cand_lo;
        cand_hi = o_hi > cand_hi ? o_hi : cand_hi;
    }
    if ((tid & 63) == 0) { s_key[(tid >> 6) * 2] = cand_lo; s_key[(tid >> 6) * 2 + 1] = cand_hi; }
    __syncthreads();
    unsigned long long key_lo = s_key[0], key_hi = s_key[1];
#pragma unroll
    for (int w = 1; w < kWgLanes / 64; ++w) {
        key_lo = s_key[2 * w] > key_lo ? s_key[2 * w] : key_lo;
        key_hi = s_key[2 * w + 1] > key_hi ? s_key[2 * w + 1] : key_hi;
    }
    const double med_lo = __longlong_as_double((long long)key_lo), med_hi = __longlong_as_double((long long)key_hi);
    const double med = any_nan ? __longlong_as_double(0x7ff8000000000000ll) : (n4 & 1) ? med_lo : 0.5 * (med_lo + med_hi);
    const double thr = med == 0.0 ? 0.0 : med / 0.6745;
    for (int lev = 1; lev <= kLevels; ++lev) {
        double *cd = s_w + A.offD[lev];
        for (int i = tid; i < A.len[lev]; i += kWgLanes) {
            const double c = cd[i], mag = fabs(c);
            cd[i] = soft_threshold(c, mag, thr, A.epsilon);
        }
    }
    __syncthreads();
    // ---- waverec
    for (int lev = kLevels; lev >= 1; --lev) {
        const int Nd = A.len[lev];
        const int No = 2 * Nd - kWF + 2;
        const double *ca = s_w + A.offA[lev], *cd = s_w + A.offD[lev];
        double *dst = s_w + A.offA[lev - 1];
        // outputs 2 q and 2 q + 1 read the same six coefficients of each band (o = q + u; k = 2 u + 1 for the even output, 2 u for the odd one): a lane takes the pair
        for (int q = tid; 2 * q < No; q += kWgLanes) {
            double sa0 = 0.0, sd0 = 0.0, sa1 = 0.0, sd1 = 0.0;
            if (q + 5 < Nd) {
#pragma unroll
                for (int u = 0; u < 6; ++u) {
                    const double a = ca[q + u], d = cd[q + u];
                    sa0 += a * lo[2 * u + 1]; sd0 += d * hi[2 * u + 1];
                    sa1 += a * lo[2 * u];     sd1 += d * hi[2 * u];
                }
            } else {
#pragma unroll
                for (int u = 0; u < 6; ++u) {
                    if (q + u < Nd) {
                        const double a = ca[q + u], d = cd[q + u];
                        sa0 += a * lo[2 * u + 1]; sd0 += d * hi[2 * u + 1];
                        sa1 += a * lo[2 * u];     sd1 += d * hi[2 * u];
                    }
                }
            }
            const double res[2] = {sa0 + sd0, sa1 + sd1};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int t = 2 * q + h;
                if (t >= No) continue;
                if (lev > 1) dst[t] = res[h];
                else if (t < n) out[(size_t)t * stride] = isfinite(res[h]) ? res[h] : 0.0;
            }
        }
        __syncthreads();
    }
}

// ---- NaN / inf test of whole records (check_nan_inf's `np.isfinite(x).all()`, preprocess_utils.py:26-33, and process_instance's np.isnan(signal).any() on the raw
// record): flags[r] = 1 if record r holds a value that is not finite.  One pass at memory speed (torch.isfinite(x).all() on float64 writes a byte per value and reads
// it back: 1.8 ms per 2 GB here, 2.8 ms per record-wise test).  flags must be zero on entry.
__global__ __launch_bounds__(256) void nonfinite_records_kernel(const double *x, size_t per_record, unsigned chunks, unsigned char *flags)
{
    const unsigned rec = blockIdx.x / chunks, ch = blockIdx.x % chunks;
    const size_t per_chunk = (per_record + chunks - 1) / chunks;
    const size_t lo = (size_t)ch * per_chunk, hi = lo + per_chunk < per_record ? lo + per_chunk : per_record;
    const unsigned long long *p = reinterpret_cast<const unsigned long long *>(x) + (size_t)rec * per_record;
    bool bad = false;
    auto test = [&](unsigned long long v) { bad |= ((v >> 52) & 0x7FFull) == 0x7FFull; };                 // exponent all ones: inf or NaN
    using u64x2 = __attribute__((ext_vector_type(2))) unsigned long long;
    size_t i = lo;
    while (i < hi && ((uintptr_t)(p + i) & 15)) { if (threadIdx.x == 0) test(p[i]); ++i; }                // to a 16-byte boundary (records of an odd length)
    constexpr int U = 4;                                                                                   // 16-byte pieces in flight per lane
    for (; i + 2 * 256 * U <= hi; i += 2 * 256 * U) {
        u64x2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const u64x2 *>(p + i + 2 * (u * 256 + threadIdx.x));
#pragma unroll
        for (int u = 0; u < U; ++u) { test(v[u][0]); test(v[u][1]); }
    }
    for (i += threadIdx.x; i < hi; i += 256) test(p[i]);
    if (__any(bad) && (threadIdx.x & 63) == 0) flags[rec] = 1;
}

int g_wavelet_wg = 1;                                      // 0: the lane-per-sequence kernel on every shape (tests compare the two)
static inline size_t waves_of(size_t sequences) { return (sequences + 63) / 64; }   // scratch is laid out per wave of 64 sequences

}  // namespace

extern "C" size_t ecgb_wavelet_denoise_scratch_bytes(int records, int n, int leads)
{
    if (records <= 0 || n <= 0 || leads <= 0) return 0;
    size_t rows = 0;
    int len = n;
    for (int lev = 1; lev <= kLevels; ++lev) { len = (len + kWF - 1) / 2; rows += 2 * ((size_t)len + 2); }
    return waves_of((size_t)records * leads) * 64 * rows * sizeof(double);
}

// planar: x, y are [records * leads][n]; only the workgroup kernel reads that layout (scratch unused)
static int wavelet_denoise_impl(const double *x_dev, double *y_dev, int records, int n, int leads, double epsilon, double *scratch_dev, size_t scratch_bytes,
                                bool planar, void *stream)
{
    const char *fn = planar ? "ecgb_wavelet_denoise_planar_f64" : "ecgb_wavelet_denoise_f64";
    if (!x_dev || !y_dev || (!planar && !scratch_dev) || records <= 0 || n <= 0 || leads <= 0) { ecgb::set_error(std::string(fn) + ": bad argument"); return ECGB_ERR_INVALID; }
    if (n % 2) { ecgb::set_error(std::string(fn) + ": odd lengths reconstruct one sample long (the reference's assignment raises)"); return ECGB_ERR_UNSUPPORTED; }
    WaveletArgs A{};
    A.x = x_dev; A.y = y_dev; A.work = scratch_dev; A.R = records; A.n = n; A.L = leads; A.epsilon = epsilon;
    A.len[0] = n;
    long long row = 0;
    for (int lev = 1; lev <= kLevels; ++lev) {
        if (A.len[lev - 1] < kWF - 1) { ecgb::set_error(std::string(fn) + ": signal too short for four db6 levels"); return ECGB_ERR_UNSUPPORTED; }
        A.len[lev] = (A.len[lev - 1] + kWF - 1) / 2;
        A.offA[lev] = row; row += A.len[lev] + 2;
        A.offD[lev] = row; row += A.len[lev] + 2;
    }
    A.rows = row;
    if (!planar && scratch_bytes < ecgb_wavelet_denoise_scratch_bytes(records, n, leads)) { ecgb::set_error(std::string(fn) + ": scratch too small"); return ECGB_ERR_INVALID; }
    const size_t S = (size_t)records * leads;
    // one workgroup per sequence, bands in LDS, when they fit (n <= ~10 200); else one lane per sequence through the scratch
    WaveletWgArgs G{};
    G.x = x_dev; G.y = y_dev; G.R = records; G.n = n; G.L = leads; G.epsilon = epsilon;
    int small = 0;                                           // A2 D2 A3 D3 A4 D4 live where x was
    for (int lev = 0; lev <= kLevels; ++lev) G.len[lev] = A.len[lev];
    auto even = [](int v) { return (v + 1) & ~1; };          // bands start on 16 bytes
    for (int lev = 2; lev <= kLevels; ++lev) { G.offA[lev] = small; small += even(A.len[lev] + 2); G.offD[lev] = small; small += even(A.len[lev] + 2); }
    const int region0 = even(std::max(n, small));
    G.offA[0] = 0;
    G.offA[1] = region0;
    G.offD[1] = region0 + even(A.len[1] + 2);
    const size_t lds = (size_t)(region0 + 2 * even(A.len[1] + 2)) * sizeof(double);
    const size_t wgs = ((S + 7) / 8) * 8;
    const bool fits = lds <= 160 * 1024 - 64 && wgs <= 0x7FFFFFFFull;
    if (planar && !fits) { ecgb::set_error(std::string(fn) + ": the sequence's bands do not fit in LDS (n <= ~10 200)"); return ECGB_ERR_UNSUPPORTED; }
    if (planar || (g_wavelet_wg && fits)) {
        const void *kern = planar ? reinterpret_cast<const void *>(wavelet_denoise_wg_kernel<true>) : reinterpret_cast<const void *>(wavelet_denoise_wg_kernel<false>);
        if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            ecgb::set_error(std::string(fn) + ": cannot size the workgroup's LDS");
            return ECGB_ERR_HIP;
        }
        if (planar) hipLaunchKernelGGL(wavelet_denoise_wg_kernel<true>, dim3((unsigned)wgs), dim3(kWgLanes), lds, (hipStream_t)stream, G);
        else hipLaunchKernelGGL(wavelet_denoise_wg_kernel<false>, dim3((unsigned)wgs), dim3(kWgLanes), lds, (hipStream_t)stream, G);
    } else {
        hipLaunchKernelGGL(wavelet_denoise_kernel, dim3((unsigned)((S + 63) / 64)), dim3(64), 0, (hipStream_t)stream, A);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { ecgb::set_error(std::string("wavelet_denoise_kernel: ") + hipGetErrorString(e)); return ECGB_ERR_HIP; }
    return ECGB_OK;
}

extern "C" int ecgb_wavelet_denoise_f64(const double *x_dev, double *y_dev, int records, int n, int leads, double epsilon, double *scratch_dev,
                                        size_t scratch_bytes, void *stream)
{
    return wavelet_denoise_impl(x_dev, y_dev, records, n, leads, epsilon, scratch_dev, scratch_bytes, false, stream);
}

extern "C" int ecgb_wavelet_denoise_planar_f64(const double *x_dev, double *y_dev, int records, int n, int leads, double epsilon, void *stream)
{
    return wavelet_denoise_impl(x_dev, y_dev, records, n, leads, epsilon, nullptr, 0, true, stream);
}

extern "C" size_t ecgb_resample_cubic_scratch_bytes(int records, int n, int leads)
{
    if (records <= 0 || n <= 0 || leads <= 0) return 0;
    return waves_of((size_t)records * leads) * 64 * (size_t)n * sizeof(double);
}

static int resample_cubic_impl(const double *x_dev, double *y_dev, int records, int n, int leads, int m, const int *out_lead, double *scratch_dev,
                               size_t scratch_bytes, unsigned char *flags_dev, bool planar_in, void *stream)
{
    const char *fn = planar_in ? "ecgb_resample_cubic_planar_f64" : "ecgb_resample_cubic_f64";
    if (!x_dev || !y_dev || !scratch_dev || records <= 0 || leads <= 0 || m <= 0) { ecgb::set_error(std::string(fn) + ": bad argument"); return ECGB_ERR_INVALID; }
    if (n < 4) { ecgb::set_error(std::string(fn) + ": a cubic spline needs at least 4 samples (scipy raises too)"); return ECGB_ERR_INVALID; }
    const size_t S = (size_t)records * leads;
    if (scratch_bytes < ecgb_resample_cubic_scratch_bytes(records, n, leads)) { ecgb::set_error(std::string(fn) + ": scratch too small (ecgb_resample_cubic_scratch_bytes)"); return ECGB_ERR_INVALID; }
    ResampleArgs A{};
    A.x = x_dev; A.y = y_dev; A.M = scratch_dev; A.R = records; A.n = n; A.L = leads; A.m = m; A.flags = flags_dev;
    A.cp[0] = 0.25;
    for (int k = 1; k < kResampleTab; ++k) A.cp[k] = 1.0 / (4.0 - A.cp[k - 1]);
    if (out_lead) {
        if (leads > kMaxLeadMap) { ecgb::set_error(std::string(fn) + ": a lead map covers at most 32 leads"); return ECGB_ERR_UNSUPPORTED; }
        unsigned seen = 0;
        for (int l = 0; l < leads; ++l) {
            if (out_lead[l] < 0 || out_lead[l] >= leads || (seen >> out_lead[l] & 1u)) { ecgb::set_error(std::string(fn) + ": the lead map is not a permutation"); return ECGB_ERR_INVALID; }
            seen |= 1u << out_lead[l];
            A.out_lead[l] = (unsigned char)out_lead[l];
        }
        A.use_map = 1;
    }
    if (planar_in) hipLaunchKernelGGL(resample_cubic_kernel<true>, dim3((unsigned)((S + 63) / 64)), dim3(64), 0, (hipStream_t)stream, A);
    else hipLaunchKernelGGL(resample_cubic_kernel<false>, dim3((unsigned)((S + 63) / 64)), dim3(64), 0, (hipStream_t)stream, A);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { ecgb::set_error(std::string("resample_cubic_kernel: ") + hipGetErrorString(e)); return ECGB_ERR_HIP; }
    return ECGB_OK;
}

extern "C" int ecgb_resample_cubic_f64(const double *x_dev, double *y_dev, int records, int n, int leads, int m, double *scratch_dev,
                                       size_t scratch_bytes, void *stream)
{
    return resample_cubic_impl(x_dev, y_dev, records, n, leads, m, nullptr, scratch_dev, scratch_bytes, nullptr, false, stream);
}

extern "C" int ecgb_resample_cubic_planar_f64(const double *x_dev, double *y_dev, int records, int n, int leads, int m, const int *out_lead,
                                              double *scratch_dev, size_t scratch_bytes, unsigned char *flags_dev, void *stream)
{
    return resample_cubic_impl(x_dev, y_dev, records, n, leads, m, out_lead, scratch_dev, scratch_bytes, flags_dev, true, stream);
}

extern "C" size_t ecgb_filtfilt_scratch_bytes(int records, int n, int leads, int max_edge)
{
    if (records <= 0 || n <= 0 || leads <= 0 || max_edge < 0) return 0;
    return waves_of((size_t)records * leads) * 64 * ((size_t)n + 2 * (size_t)max_edge + (size_t)n) * sizeof(double);
}

static int filtfilt_impl(const double *x_dev, double *y_dev, int records, int n, int leads, int n_filters, const int *n_taps, const double *b, const double *a,
                         const double *zi, double *scratch_dev, size_t scratch_bytes, unsigned char *flags_dev, unsigned char *raw_flags_dev, bool planar,
                         void *stream)
{
    if (!x_dev || !y_dev || !n_taps || !b || !a || !zi || !scratch_dev || records <= 0 || n <= 0 || leads <= 0) {
        ecgb::set_error("ecgb_filtfilt_f64: bad argument");
        return ECGB_ERR_INVALID;
    }
    if (n_filters < 1 || n_filters > kMaxFilters) { ecgb::set_error("ecgb_filtfilt_f64: 1..4 filters per call"); return ECGB_ERR_UNSUPPORTED; }
    FiltfiltArgs A{};
    A.x = x_dev; A.y = y_dev; A.R = records; A.n = n; A.L = leads; A.n_filters = n_filters; A.flags = flags_dev; A.raw_flags = raw_flags_dev;
    int max_edge = 0;
    for (int k = 0; k < n_filters; ++k) {
        const int nb = n_taps[k];
        if (nb < 2 || nb > kMaxTaps) { ecgb::set_error("ecgb_filtfilt_f64: 2..9 coefficients per filter"); return ECGB_ERR_UNSUPPORTED; }
        Filt &F = A.f[k];
        F.nb = nb; F.edge = 3 * nb;
        if (n <= F.edge) { ecgb::set_error("ecgb_filtfilt_f64: the signal must be longer than padlen = 3 * taps (scipy raises too)"); return ECGB_ERR_INVALID; }
        const double a0 = a[k * kMaxTaps];
        if (a0 == 0.0) { ecgb::set_error("ecgb_filtfilt_f64: a[0] == 0"); return ECGB_ERR_INVALID; }
        for (int j = 0; j < nb; ++j) { F.b[j] = b[k * kMaxTaps + j] / a0; F.a[j] = a[k * kMaxTaps + j] / a0; }   // as lfilter does
        for (int j = 0; j < nb - 1; ++j) F.zi[j] = zi[k * (kMaxTaps - 1) + j];
        max_edge = std::max(max_edge, F.edge);
    }
    if (scratch_bytes < ecgb_filtfilt_scratch_bytes(records, n, leads, max_edge)) { ecgb::set_error("ecgb_filtfilt_f64: scratch too small"); return ECGB_ERR_INVALID; }
    const size_t S = (size_t)records * leads;
    A.ext = scratch_dev;
    A.ext_rows = n + 2 * max_edge;
    A.mid = scratch_dev + waves_of(S) * 64 * (size_t)A.ext_rows;
    if (planar) hipLaunchKernelGGL(filtfilt_kernel<true>, dim3((unsigned)((S + 63) / 64)), dim3(64), 0, (hipStream_t)stream, A);
    else hipLaunchKernelGGL(filtfilt_kernel<false>, dim3((unsigned)((S + 63) / 64)), dim3(64), 0, (hipStream_t)stream, A);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { ecgb::set_error(std::string("filtfilt_kernel: ") + hipGetErrorString(e)); return ECGB_ERR_HIP; }
    return ECGB_OK;
}

extern "C" int ecgb_filtfilt_f64(const double *x_dev, double *y_dev, int records, int n, int leads, int n_filters, const int *n_taps,
                                 const double *b, const double *a, const double *zi, double *scratch_dev, size_t scratch_bytes,
                                 void *stream)
{
    return filtfilt_impl(x_dev, y_dev, records, n, leads, n_filters, n_taps, b, a, zi, scratch_dev, scratch_bytes, nullptr, nullptr, false, stream);
}

extern "C" int ecgb_filtfilt_planar_f64(const double *x_dev, double *y_dev, int records, int n, int leads, int n_filters, const int *n_taps,
                                        const double *b, const double *a, const double *zi, double *scratch_dev, size_t scratch_bytes,
                                        unsigned char *flags_dev, unsigned char *raw_flags_dev, void *stream)
{
    if (x_dev == y_dev) { ecgb::set_error("ecgb_filtfilt_planar_f64: input and output have different layouts and must not alias"); return ECGB_ERR_INVALID; }
    return filtfilt_impl(x_dev, y_dev, records, n, leads, n_filters, n_taps, b, a, zi, scratch_dev, scratch_bytes, flags_dev, raw_flags_dev, true, stream);
}

extern "C" void ecgb_set_wavelet_workgroup_kernel(int on) { g_wavelet_wg = on ? 1 : 0; }

extern "C" int ecgb_nonfinite_records_f64(const double *x_dev, int records, size_t per_record, unsigned char *flags_dev, void *stream)
{
    if (!x_dev || !flags_dev || records <= 0 || per_record == 0) { ecgb::set_error("ecgb_nonfinite_records_f64: bad argument"); return ECGB_ERR_INVALID; }
    // enough workgroups to fill the chip whatever the record count: chunks of about 64 KiB, at least one per record
    size_t chunks = (per_record * 8 + 65535) / 65536;
    if (chunks < 1) chunks = 1;
    if ((size_t)records * chunks > 0x7FFFFFFFull) { ecgb::set_error("ecgb_nonfinite_records_f64: too many chunks"); return ECGB_ERR_UNSUPPORTED; }
    hipLaunchKernelGGL(nonfinite_records_kernel, dim3((unsigned)((size_t)records * chunks)), dim3(256), 0, (hipStream_t)stream, x_dev, per_record, (unsigned)chunks, flags_dev);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { ecgb::set_error(std::string("nonfinite_records_kernel: ") + hipGetErrorString(e)); return ECGB_ERR_HIP; }
    return ECGB_OK;
}
