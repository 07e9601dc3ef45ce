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


def segment_ecg(ecg_data, text_data, seg_len):               # preprocess_utils.py:103-116
    num_segments = ecg_data.shape[0] // seg_len
    return (np.array([ecg_data[i * seg_len:(i + 1) * seg_len, :] for i in range(num_segments)]), [text_data] * num_segments)
