"""Golden vectors for the offline conditioning, generated HERE from the libraries the reference calls with the
reference's own arguments (ecg_byte/utils/preprocess_utils.py:66-101: scipy.signal.iirnotch / butter / filtfilt, scipy.interpolate.interp1d);
the reference module cannot be imported (wfdb, pywt missing), so its call sites are spelled out below.  Run: python tests/golden/make_preprocess_golden.py
Writes tests/golden/preprocess_golden.npz (one 12-lead record of 640 samples at 500 Hz, float64: input, filtered, resampled to 250 Hz)."""
import os
import sys

import numpy as np
from scipy import interpolate, signal

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from ecg_byte_amd import synth  # noqa: E402


def advanced_ecg_filter(ecg_data, fs=500, notch_freqs=[50, 60], highcut=100.0):        # preprocess_utils.py:66-88, call for call
    filtered_ecg = ecg_data.copy()
    quality_factor = 30.0
    for notch_freq in notch_freqs:
        b_notch, a_notch = signal.iirnotch(notch_freq, quality_factor, fs)
        filtered_ecg = signal.filtfilt(b_notch, a_notch, filtered_ecg, axis=0)
    lowcut = 0.5
    nyquist = 0.5 * fs
    low = lowcut / nyquist
    high = highcut / nyquist
    order = 4
    b_band, a_band = signal.butter(order, [low, high], btype='band')
    filtered_ecg = signal.filtfilt(b_band, a_band, filtered_ecg, axis=0)
    baseline_cutoff = 0.05
    baseline_low = baseline_cutoff / nyquist
    b_baseline, a_baseline = signal.butter(order, baseline_low, btype='high')
    filtered_ecg = signal.filtfilt(b_baseline, a_baseline, filtered_ecg, axis=0)
    return filtered_ecg


def nsample_ecg(ecg_data, orig_fs, target_fs):                                         # preprocess_utils.py:90-101, call for call
    num_samples, num_leads = ecg_data.shape
    duration = num_samples / orig_fs
    t_original = np.linspace(0, duration, num_samples, endpoint=True)
    t_target = np.linspace(0, duration, int(num_samples * target_fs / orig_fs), endpoint=True)
    downsampled_data = np.zeros((len(t_target), num_leads))
    for lead in range(num_leads):
        f = interpolate.interp1d(t_original, ecg_data[:, lead], kind='cubic', bounds_error=False, fill_value="extrapolate")
        downsampled_data[:, lead] = f(t_target)
    return downsampled_data


def main():
    rng = np.random.default_rng(11)
    n = 640
    recs = synth.synth_ecg(1, n, seed=5)                    # [1, 12, n] float64, millivolt-like
    raw = np.ascontiguousarray(recs.transpose(0, 2, 1))     # wfdb layout [n, leads] per record
    raw = raw + 0.05 * np.sin(2 * np.pi * 50.0 * np.arange(n) / 500.0)[None, :, None] + 0.015 * rng.standard_normal(raw.shape) + 0.4   # mains hum, noise, offset
    filt = np.stack([advanced_ecg_filter(r) for r in raw])
    res = np.stack([nsample_ecg(f, 500, 250) for f in filt])
    np.savez_compressed(os.path.join(HERE, "preprocess_golden.npz"), raw=raw, filtered=filt, resampled=res)
    print("raw", raw.shape, "filtered", filt.shape, "resampled", res.shape)


if __name__ == "__main__":
    main()
