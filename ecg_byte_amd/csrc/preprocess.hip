// preprocess.hip -- the reference's offline ECG conditioning on MI355X (gfx950), float64 throughout:
//   ecgb_filtfilt_f64        advanced_ecg_filter   ecg_byte/utils/preprocess_utils.py:66-88   (scipy.signal.filtfilt chains)
//   ecgb_resample_cubic_f64  nsample_ecg           ecg_byte/utils/preprocess_utils.py:90-101  (scipy interp1d kind='cubic')
//   ecgb_wavelet_denoise_f64 wavelet_denoise       ecg_byte/utils/preprocess_utils.py:43-64   (pywt wavedec / threshold / waverec, db6)
// One-time work in the reference (12 worker processes, ~1 ms per filter and record); here a whole batch of records per launch.
//
// Every stage is a recursion along time (IIR state, tridiagonal sweep, nothing to tile), so the parallel axis is the SEQUENCE: one lane =
// one (record, lead) time series, 49 152 lanes for 4 096 twelve-lead records.  The records arrive as [record][time][lead] (what
// wfdb.rdsamp returns, preprocess_utils.py:126); intermediates live TIME-MAJOR, [time][sequence], so that the 64 lanes of a wave touch
// 512 contiguous bytes at every step.  Arithmetic follows the reference's libraries operation by operation where that is what fixes
// the bits (scipy's direct-form-II-transposed loop, its odd extension and initial conditions); -ffp-contract=off keeps a*b+c two roundings.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>

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
    double *ext;                     // scratch [n + 2 * max edge][S]: the forward pass's output over the extended signal
    double *mid;                     // scratch [n][S]: a filter's result, the next filter's input
    int R, n, L, n_filters;
    Filt f[kMaxFilters];
};

// scipy.signal._signaltools.filtfilt (method 'pad', padtype 'odd') around scipy's lfilter (sigtools DOUBLE_filt):
//   ext = [2 x[0] - x[e..1], x, 2 x[n-1] - x[n-2..n-1-e]];  y1 = lfilter(ext, z = zi * ext[0]);  y2 = lfilter(reverse(y1), z = zi * y1[-1]);
//   result = reverse(y2)[e : e + n]
// lfilter step:  y = z[0] + b[0] x;  z[k] = z[k+1] + x b[k+1] - y a[k+1]  (k < nb - 2);  z[nb-2] = x b[nb-1] - y a[nb-1].
template <int NB, typename SRC, typename DST>
__device__ __forceinline__ void filtfilt_one(const Filt &F, int n, SRC src, double *ext, size_t S, size_t seq, DST dst)
{
    const int e = F.edge, N = n + 2 * e;
    double z[NB - 1];
    const double first = src(0), last = src(n - 1);
    auto ext_at = [&](int i) -> double {
        if (i < e) return 2.0 * first - src(e - i);
        if (i < e + n) return src(i - e);
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
    for (int i = 0; i < N; ++i) ext[(size_t)i * S + seq] = step(ext_at(i));
    const double y0 = ext[(size_t)(N - 1) * S + seq];
#pragma unroll
    for (int k = 0; k < NB - 1; ++k) z[k] = F.zi[k] * y0;
    for (int i = 0; i < N; ++i) {
        const double w = step(ext[(size_t)(N - 1 - i) * S + seq]);
        const int pos = N - 1 - i - e;
        if (pos >= 0 && pos < n) dst(pos, w);
    }
}

template <typename SRC, typename DST>
__device__ __forceinline__ void filtfilt_dispatch(const Filt &F, int n, SRC src, double *ext, size_t S, size_t seq, DST dst)
{
    switch (F.nb) {
    case 2: filtfilt_one<2>(F, n, src, ext, S, seq, dst); break;
    case 3: filtfilt_one<3>(F, n, src, ext, S, seq, dst); break;
    case 4: filtfilt_one<4>(F, n, src, ext, S, seq, dst); break;
    case 5: filtfilt_one<5>(F, n, src, ext, S, seq, dst); break;
    case 6: filtfilt_one<6>(F, n, src, ext, S, seq, dst); break;
    case 7: filtfilt_one<7>(F, n, src, ext, S, seq, dst); break;
    case 8: filtfilt_one<8>(F, n, src, ext, S, seq, dst); break;
    default: filtfilt_one<9>(F, n, src, ext, S, seq, dst); break;
    }
}

__global__ __launch_bounds__(64) void filtfilt_kernel(FiltfiltArgs A)
{
    const size_t S = (size_t)A.R * A.L;
    const size_t seq = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (seq >= S) return;
    const size_t r = seq / A.L, l = seq % A.L;
    const double *xin = A.x + (r * A.n) * A.L + l;
    double *yout = A.y + (r * A.n) * A.L + l;
    const int L = A.L;
    for (int k = 0; k < A.n_filters; ++k) {
        const bool first = k == 0, lastf = k == A.n_filters - 1;
        auto src_x = [&](int t) -> double { return xin[(size_t)t * L]; };
        auto src_m = [&](int t) -> double { return A.mid[(size_t)t * S + seq]; };
        auto dst_y = [&](int t, double v) { yout[(size_t)t * L] = v; };
        auto dst_m = [&](int t, double v) { A.mid[(size_t)t * S + seq] = v; };
        // a filter reads all of its input before the backward pass writes the first output sample, so `mid` can be source and
        // destination of the same filter
        if (first && lastf) filtfilt_dispatch(A.f[k], A.n, src_x, A.ext, S, seq, dst_y);
        else if (first) filtfilt_dispatch(A.f[k], A.n, src_x, A.ext, S, seq, dst_m);
        else if (lastf) filtfilt_dispatch(A.f[k], A.n, src_m, A.ext, S, seq, dst_y);
        else filtfilt_dispatch(A.f[k], A.n, src_m, A.ext, S, seq, dst_m);
    }
}

}  // namespace

extern "C" size_t ecgb_filtfilt_scratch_bytes(int records, int n, int leads, int max_edge)
{
    if (records <= 0 || n <= 0 || leads <= 0 || max_edge < 0) return 0;
    return (size_t)records * leads * ((size_t)n + 2 * (size_t)max_edge + (size_t)n) * sizeof(double);
}

extern "C" int ecgb_filtfilt_f64(const double *x_dev, double *y_dev, int records, int n, int leads, int n_filters, const int *n_taps,
                                 const double *b, const double *a, const double *zi, double *scratch_dev, size_t scratch_bytes,
                                 void *stream)
{
    if (!x_dev || !y_dev || !n_taps || !b || !a || !zi || !scratch_dev || records <= 0 || n <= 0 || leads <= 0) {
        ecgb::set_error("ecgb_filtfilt_f64: bad argument");
        return ECGB_ERR_INVALID;
    }
    if (n_filters < 1 || n_filters > kMaxFilters) { ecgb::set_error("ecgb_filtfilt_f64: 1..4 filters per call"); return ECGB_ERR_UNSUPPORTED; }
    FiltfiltArgs A{};
    A.x = x_dev; A.y = y_dev; A.R = records; A.n = n; A.L = leads; A.n_filters = n_filters;
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
    A.mid = scratch_dev + S * ((size_t)n + 2 * (size_t)max_edge);
    hipLaunchKernelGGL(filtfilt_kernel, dim3((unsigned)((S + 63) / 64)), dim3(64), 0, (hipStream_t)stream, A);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { ecgb::set_error(std::string("filtfilt_kernel: ") + hipGetErrorString(e)); return ECGB_ERR_HIP; }
    return ECGB_OK;
}
