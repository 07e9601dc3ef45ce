"""CPU restatement of the reference's offline conditioning (ecg_byte/utils/preprocess_utils.py:26-116) -- TEST INFRASTRUCTURE ONLY
(imported by tests/ and the golden generator; the product path in ecg_byte_amd/ never touches it).

The reference module itself cannot be imported here (it pulls wfdb and pywt at import time, neither is installed), so the call sites are
restated with the libraries it calls:
  advanced_ecg_filter  scipy.signal.iirnotch / butter / filtfilt with the reference's arguments          PINNED to scipy (installed)
  nsample_ecg          scipy.interpolate.interp1d(kind='cubic', fill_value='extrapolate')                PINNED to scipy
  segment_ecg, reorder_indices, check_nan_inf   plain numpy                                              literal
  wavelet_denoise      pywt.wavedec / threshold / waverec, db6, level 4, mode 'symmetric'                **parity unpinned**: PyWavelets is
                       absent, restated from its published algorithm (filter bank from the Daubechies-6 scaling coefficients, symmetric
                       half-point extension, periodic down/up-sampling phase as in pywt's convolution code)
"""
import numpy as np


def check_nan_inf(data):                                     # preprocess_utils.py:26-33
    if np.any(np.isnan(data)) or np.any(np.isinf(data)):
        data = np.nan_to_num(data, nan=0.0, posinf=0.0, neginf=0.0)
    return data


def reorder_indices(signals):                                # preprocess_utils.py:35-40
    current_order = ['I', 'II', 'III', 'aVR', 'aVF', 'aVL', 'V1', 'V2', 'V3', 'V4', 'V5', 'V6']
    desired_order = ['I', 'II', 'III', 'aVL', 'aVR', 'aVF', 'V1', 'V2', 'V3', 'V4', 'V5', 'V6']
    order_mapping = {lead: index for index, lead in enumerate(current_order)}
    return signals[:, [order_mapping[lead] for lead in desired_order]]


def advanced_ecg_filter(ecg_data, fs=500, notch_freqs=(50, 60), highcut=100.0):     # preprocess_utils.py:66-88
    from scipy import signal
    filtered_ecg = ecg_data.copy()
    for notch_freq in notch_freqs:
        b_notch, a_notch = signal.iirnotch(notch_freq, 30.0, fs)
        filtered_ecg = signal.filtfilt(b_notch, a_notch, filtered_ecg, axis=0)
    nyquist = 0.5 * fs
    b_band, a_band = signal.butter(4, [0.5 / nyquist, highcut / nyquist], btype='band')
    filtered_ecg = signal.filtfilt(b_band, a_band, filtered_ecg, axis=0)
    b_baseline, a_baseline = signal.butter(4, 0.05 / nyquist, btype='high')
    filtered_ecg = signal.filtfilt(b_baseline, a_baseline, filtered_ecg, axis=0)
    return filtered_ecg


def filtfilt_literal(b, a, x):
    """scipy.signal.filtfilt(b, a, x) for one 1-D series, written out (odd extension, lfilter_zi start, direct form II transposed both
    ways): what csrc/preprocess.hip restates.  Checked against scipy in tests/test_oracle_preprocess.py."""
    from scipy import signal
    b, a = np.asarray(b, np.float64) / a[0], np.asarray(a, np.float64) / a[0]
    nb = max(len(a), len(b))
    b = np.concatenate([b, np.zeros(nb - len(b))]); a = np.concatenate([a, np.zeros(nb - len(a))])
    e = 3 * nb
    zi = signal.lfilter_zi(b, a)
    ext = np.concatenate([2 * x[0] - x[e:0:-1], x, 2 * x[-1] - x[-2:-(e + 2):-1]])

    def lfilter(seq, z):
        z = z.copy()
        y = np.empty_like(seq)
        for i, xi in enumerate(seq):
            yi = z[0] + b[0] * xi
            for k in range(nb - 2):
                z[k] = z[k + 1] + xi * b[k + 1] - yi * a[k + 1]
            z[nb - 2] = xi * b[nb - 1] - yi * a[nb - 1]
            y[i] = yi
        return y
    y = lfilter(ext, zi * ext[0])
    y = lfilter(y[::-1], zi * y[-1])[::-1]
    return y[e:-e]


def nsample_ecg(ecg_data, orig_fs, target_fs):               # preprocess_utils.py:90-101
    from scipy import interpolate
    num_samples, num_leads = ecg_data.shape
    duration = num_samples / orig_fs
    t_original = np.linspace(0, duration, num_samples, endpoint=True)
    t_target = np.linspace(0, duration, int(num_samples * target_fs / orig_fs), endpoint=True)
    downsampled_data = np.zeros((len(t_target), num_leads))
    for lead in range(num_leads):
        f = interpolate.interp1d(t_original, ecg_data[:, lead], kind='cubic', bounds_error=False, fill_value="extrapolate")
        downsampled_data[:, lead] = f(t_target)
    return downsampled_data


def segment_ecg(ecg_data, text_data, seg_len):               # preprocess_utils.py:103-116
    num_segments = ecg_data.shape[0] // seg_len
    return (np.array([ecg_data[i * seg_len:(i + 1) * seg_len, :] for i in range(num_segments)]), [text_data] * num_segments)


# ---- wavelet_denoise: PyWavelets restated (PARITY UNPINNED: pywt is not installed; conventions from its documentation and C sources) --------
DB6_DEC_LO = np.array([-0.00107730108499558, 0.004777257511010651, 0.0005538422009938016, -0.031582039318031156, 0.02752286553001629,
                       0.09750160558707936, -0.12976686756709563, -0.22626469396516913, 0.3152503517092432, 0.7511339080215775,
                       0.4946238903983854, 0.11154074335008017])
DB6_DEC_HI = np.array([(-1.0) ** (k + 1) * DB6_DEC_LO[11 - k] for k in range(12)])      # quadrature mirror of the low-pass


def dwt_symmetric(x):
    """pywt.dwt(x, 'db6', mode='symmetric'): full convolution of the half-point symmetric extension with the decomposition filters,
    every second sample starting at index 1: (len(x) + 11) // 2 coefficients per band."""
    F = 12
    ext = np.concatenate([x[F - 2::-1] if len(x) >= F - 1 else None, x, x[:-F:-1]])      # x[10..0] | x | x[-1..-11]
    full_lo = np.convolve(ext, DB6_DEC_LO)[F - 1:]           # index i of the un-padded convolution = index i + (F - 1) here
    full_hi = np.convolve(ext, DB6_DEC_HI)[F - 1:]
    nc = (len(x) + F - 1) // 2
    return full_lo[1:2 * nc:2], full_hi[1:2 * nc:2]


def idwt_symmetric(ca, cd):
    """pywt.idwt(ca, cd, 'db6'): 2 len(ca) - 10 samples, x[t] = sum_o ca[o] dec_lo[2o + 1 - t] + cd[o] dec_hi[2o + 1 - t]."""
    n = 2 * len(ca) - 10
    x = np.zeros(n)
    for t in range(n):
        o = np.arange(t // 2, min(len(ca), (t + 12) // 2))
        k = 2 * o + 1 - t
        x[t] = (ca[o] * DB6_DEC_LO[k]).sum() + (cd[o] * DB6_DEC_HI[k]).sum()
    return x


def wavelet_denoise(ecg_data, level=4, epsilon=1e-10):       # preprocess_utils.py:43-64
    denoised = np.zeros_like(ecg_data)
    for i in range(ecg_data.shape[1]):
        a, details = ecg_data[:, i], []
        for _ in range(level):                                # wavedec
            a, d = dwt_symmetric(a)
            details.append(d)
        coeffs = [a] + details[::-1]                          # [cA4, cD4, cD3, cD2, cD1]
        median_abs = np.median(np.abs(coeffs[-level]))
        threshold = 0 if median_abs == 0 else median_abs / 0.6745

        def safe_threshold(c):
            with np.errstate(divide='ignore', invalid='ignore'):
                thresholded = c * np.clip(1 - threshold / np.abs(c), 0, None)           # pywt.threshold(c, threshold, mode='soft')
            return np.where(np.isfinite(thresholded) & (np.abs(c) > epsilon), thresholded, 0)
        new = [coeffs[0]] + [safe_threshold(c) for c in coeffs[1:]]
        a = new[0]
        for d in new[1:]:                                     # waverec
            if len(a) == len(d) + 1:
                a = a[:-1]
            a = idwt_symmetric(a, d)
        denoised[:, i] = a
    return np.nan_to_num(denoised, nan=0.0, posinf=0.0, neginf=0.0)
