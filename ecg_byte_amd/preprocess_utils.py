"""Host mirror of the offline conditioning in ecg_byte/utils/preprocess_utils.py:26-113 on top of csrc/preprocess.hip.

Same function names and argument meaning as the reference; the arrays are float64 CUDA tensors and every function takes one record
`[n, leads]` (the reference's shape) or a batch `[records, n, leads]` -- the batch is what the kernels are built for (one lane per
(record, lead) series).  Filter design stays on the host with scipy.signal, the library the reference calls (`iirnotch`, `butter`,
`lfilter_zi`: a few dozen coefficients); everything that touches samples runs on the device.  No CPU fallback: without the HIP
library these functions raise.
"""
import ctypes as C
import functools

import numpy as np
import torch

from . import _lib

_MAX_TAPS = 9


def _L():
    return _lib.lib()


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _batch(x):
    if not (torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float64):
        raise TypeError("expected a float64 CUDA tensor [n, leads] or [records, n, leads]")
    _lib.require_current(x.device)
    return (x[None] if x.dim() == 2 else x).contiguous(), x.dim() == 2


def nonfinite_records(x, records=None):
    """uint8 [records]: 1 where a record of the float64 batch x (records consecutive, equal-sized pieces of it; default: its first dimension) holds a NaN or an
    infinity (ecgb_nonfinite_records_f64: one pass at memory speed, no temporaries)."""
    assert x.dtype == torch.float64 and x.is_contiguous()
    R = x.shape[0] if records is None else int(records)
    flags = torch.zeros(R, dtype=torch.uint8, device=x.device)
    if x.numel():
        _lib.check(_L().ecgb_nonfinite_records_f64(C.c_void_p(x.data_ptr()), R, x.numel() // R, C.c_void_p(flags.data_ptr()), _st()))
    return flags


def check_nan_inf(data, step_name):
    """preprocess_utils.py:26-33: NaN / inf -> 0 (with the reference's warning)."""
    if not bool(torch.isfinite(data).all()):
        print(f"Warning: NaN or inf values detected after {step_name}")
        data = torch.nan_to_num(data, nan=0.0, posinf=0.0, neginf=0.0)
    return data


def reorder_indices(signals):
    """preprocess_utils.py:35-40: MIMIC lead order (I, II, III, aVR, aVF, aVL, V1-6) -> (I, II, III, aVL, aVR, aVF, V1-6)."""
    current_order = ['I', 'II', 'III', 'aVR', 'aVF', 'aVL', 'V1', 'V2', 'V3', 'V4', 'V5', 'V6']
    desired_order = ['I', 'II', 'III', 'aVL', 'aVR', 'aVF', 'V1', 'V2', 'V3', 'V4', 'V5', 'V6']
    order_mapping = {lead: index for index, lead in enumerate(current_order)}
    new_indices = [order_mapping[lead] for lead in desired_order]
    return signals[..., new_indices]


def design_filters(fs=500, notch_freqs=(50, 60), highcut=100.0):
    """The four (b, a) pairs of advanced_ecg_filter, preprocess_utils.py:69-86, from the same scipy.signal calls."""
    from scipy import signal
    out = [signal.iirnotch(f, 30.0, fs) for f in notch_freqs]
    nyquist = 0.5 * fs
    out.append(signal.butter(4, [0.5 / nyquist, highcut / nyquist], btype='band'))
    out.append(signal.butter(4, 0.05 / nyquist, btype='high'))
    return out


def _pack_filters(chunk):
    """(n_taps, b, a, zi) as ecgb_filtfilt_f64 takes them: [filters][9] coefficient rows, zi = scipy.signal.lfilter_zi(b, a)."""
    from scipy import signal
    taps = (C.c_int * len(chunk))()
    b = np.zeros((len(chunk), _MAX_TAPS)); a = np.zeros((len(chunk), _MAX_TAPS)); zi = np.zeros((len(chunk), _MAX_TAPS - 1))
    for k, (bk, ak) in enumerate(chunk):
        bk, ak = np.atleast_1d(np.asarray(bk, np.float64)), np.atleast_1d(np.asarray(ak, np.float64))
        nt = max(len(bk), len(ak))
        if nt > _MAX_TAPS or nt < 2:
            raise ValueError("filters of 2..9 coefficients")
        taps[k] = nt
        b[k, :len(bk)] = bk; a[k, :len(ak)] = ak
        zi[k, :nt - 1] = signal.lfilter_zi(bk, ak)
    return taps, b, a, zi


def filtfilt(filters, x):
    """scipy.signal.filtfilt(b, a, x, axis=-2) for each (b, a) of `filters` in turn, on the device (ecgb_filtfilt_f64; at most four per
    launch, longer chains are cut)."""
    xb, single = _batch(x)
    R, n, leads = xb.shape
    y = xb
    for s0 in range(0, len(filters), 4):
        chunk = filters[s0:s0 + 4]
        taps, b, a, zi = _pack_filters(chunk)
        edge = 3 * max(taps)
        nbytes = _L().ecgb_filtfilt_scratch_bytes(R, n, leads, edge)
        scratch = torch.empty(nbytes // 8, dtype=torch.float64, device=xb.device)
        out = torch.empty_like(xb)
        _lib.check(_L().ecgb_filtfilt_f64(C.c_void_p(y.data_ptr()), C.c_void_p(out.data_ptr()), R, n, leads, len(chunk), taps,
                                          b.ctypes.data_as(C.c_void_p), a.ctypes.data_as(C.c_void_p), zi.ctypes.data_as(C.c_void_p),
                                          C.c_void_p(scratch.data_ptr()), nbytes, _st()))
        y = out
    return y[0] if single else y


_PLANAR_MAX_N = 10000       # the bands of a sequence live in LDS (ecgb_wavelet_denoise_planar_f64)


@functools.lru_cache(maxsize=8)
def _reference_chain(fs):
    """advanced_ecg_filter's four filters at sampling rate fs, designed and packed once (scipy's iirnotch / butter / lfilter_zi take ~0.4 ms of host time per call,
    and the fast path of condition_records is otherwise one launch sequence and one synchronisation)."""
    filters = design_filters(fs)
    taps, b, a, zi = _pack_filters(filters)
    return taps, b, a, zi, len(filters)


def _condition_planar(x, orig_fs, target_fs, out_lead):
    """The filter chain, the wavelet shrinkage and the resampling of condition_records' fast path with the intermediates sequence-major
    ([records * leads][n]; include/ecgbyte.h): the same three stages, the same bits as advanced_ecg_filter -> wavelet_denoise -> nsample_ecg, the lead
    reorder folded into the last store and the "not finite" tests -- of the stages' results and of the raw records -- into the kernels.
    Returns ([records, m, leads], flags [records] uint8: a stage wrote a value that is not finite, raw_flags [records] uint8: the record came in with one; the two
    are rows of one [2, records] tensor, so one copy brings both to the host)."""
    R, n, leads = x.shape
    taps, b, a, zi, n_filters = _reference_chain(orig_fs)
    nbytes = _L().ecgb_filtfilt_scratch_bytes(R, n, leads, 3 * max(taps))
    nres = _L().ecgb_resample_cubic_scratch_bytes(R, n, leads)
    scratch = torch.empty(max(nbytes, nres) // 8, dtype=torch.float64, device=x.device)
    both = torch.zeros(2, R, dtype=torch.uint8, device=x.device)
    flags, raw_flags = both[0], both[1]
    planar = torch.empty(R * leads * n, dtype=torch.float64, device=x.device)
    _lib.check(_L().ecgb_filtfilt_planar_f64(C.c_void_p(x.data_ptr()), C.c_void_p(planar.data_ptr()), R, n, leads, n_filters, taps,
                                             b.ctypes.data_as(C.c_void_p), a.ctypes.data_as(C.c_void_p), zi.ctypes.data_as(C.c_void_p),
                                             C.c_void_p(scratch.data_ptr()), nbytes, C.c_void_p(flags.data_ptr()), C.c_void_p(raw_flags.data_ptr()), _st()))
    _lib.check(_L().ecgb_wavelet_denoise_planar_f64(C.c_void_p(planar.data_ptr()), C.c_void_p(planar.data_ptr()), R, n, leads, 1e-10, _st()))
    m = int(n * target_fs / orig_fs)
    out = torch.empty((R, m, leads), dtype=torch.float64, device=x.device)
    lead_map = None
    if out_lead is not None:
        lead_map = (C.c_int * leads)(*out_lead)
    _lib.check(_L().ecgb_resample_cubic_planar_f64(C.c_void_p(planar.data_ptr()), C.c_void_p(out.data_ptr()), R, n, leads, m, lead_map,
                                                   C.c_void_p(scratch.data_ptr()), nres, C.c_void_p(flags.data_ptr()), _st()))
    return out, flags, raw_flags


def advanced_ecg_filter(ecg_data, fs=500, notch_freqs=[50, 60], highcut=100.0):
    """preprocess_utils.py:66-88: notch filters, Butterworth band-pass 0.5 Hz..highcut (order 4), high-pass 0.05 Hz (order 4), each
    applied forward and backward (filtfilt)."""
    return filtfilt(design_filters(fs, tuple(notch_freqs), highcut), ecg_data)


_planar_pipeline = True


def set_planar_pipeline(on=True):
    """Dev / tests: False sends condition_records through the three per-stage calls ([records, n, leads] between the stages) instead of the
    sequence-major pipeline (same bits)."""
    global _planar_pipeline
    _planar_pipeline = bool(on)


def set_wavelet_workgroup_kernel(on=True):
    """Dev / tests: False forms the wavelet stage with the lane-per-sequence kernel on every shape (default: one workgroup per sequence with the bands in LDS
    whenever they fit; same bits)."""
    _L().ecgb_set_wavelet_workgroup_kernel(1 if on else 0)


def wavelet_denoise(ecg_data, wavelet='db6', level=4, epsilon=1e-10):
    """preprocess_utils.py:43-64: db6 / level 4 wavelet shrinkage with the threshold median(|cD4|) / 0.6745 (ecgb_wavelet_denoise_f64;
    PyWavelets is absent: restated from its published algorithm, parity unpinned).  Only the reference's wavelet and level are built."""
    if wavelet != 'db6' or level != 4:
        raise NotImplementedError("the reference calls wavelet_denoise with its defaults (db6, level 4): only those are built")
    xb, single = _batch(ecg_data)
    R, n, leads = xb.shape
    out = torch.empty_like(xb)
    nbytes = _L().ecgb_wavelet_denoise_scratch_bytes(R, n, leads)
    scratch = torch.empty(nbytes // 8, dtype=torch.float64, device=xb.device)
    _lib.check(_L().ecgb_wavelet_denoise_f64(C.c_void_p(xb.data_ptr()), C.c_void_p(out.data_ptr()), R, n, leads, float(epsilon),
                                             C.c_void_p(scratch.data_ptr()), nbytes, _st()))
    return out[0] if single else out


def condition_records(signals, reorder=True, seg_len=1250, orig_fs=500, target_fs=250, return_kept=False):
    """process_instance, preprocess_utils.py:118-160, for a BATCH of raw records [records, n, 12] already on the device (reading wfdb
    files stays with the caller): MIMIC lead order, the filter chain, wavelet shrinkage, resampling to target_fs, whole segments of
    seg_len samples, NaN/inf -> 0 after every stage (check_nan_inf).

    A record whose RAW signal holds a NaN or an infinity is skipped, as the reference skips it (preprocess_utils.py:134-136 returns
    None before the first check_nan_inf can zero-fill it): it is not in the output.  The reference's second test, after the last
    check_nan_inf (157-159), can never fire -- the array has just been made finite -- and is reproduced by construction.
    Returns [kept records, segments, seg_len, 12]; with return_kept=True also the bool mask [records] of the records kept, so the
    caller can drop the matching text entries."""
    if signals.dim() != 3:
        raise ValueError("condition_records takes a batch [records, n, leads]")
    signals = signals.contiguous()
    n, leads = signals.shape[1], signals.shape[2]
    planar_ok = _planar_pipeline and n % 2 == 0 and 96 <= n <= _PLANAR_MAX_N and leads <= 32 and (not reorder or leads == 12) and signals.shape[0] > 0
    if planar_ok:
        # Fast path, sequence-major intermediates (see _condition_planar): the stages run back to back on EVERY record with no host synchronisation -- the test of the
        # raw records is read off the first filter's loads instead of a pass of its own (a record that holds NaN / inf costs its share of the work and is dropped
        # afterwards; the stages are per record, so the others are not touched) -- and the flags come to the host once, at the end.
        out_lead = None
        if reorder:
            new_indices = reorder_indices(torch.arange(12))        # output lead c = input lead new_indices[c]
            out_lead = [0] * 12
            for c, l in enumerate(new_indices.tolist()):
                out_lead[l] = c
        x, flags, raw_flags = _condition_planar(signals, orig_fs, target_fs, out_lead)
        host = torch.stack([flags, raw_flags]).cpu()
        kept = (host[1] == 0).to(signals.device)
        n_bad = int((host[1] != 0).sum())
        if n_bad:
            print(f"Warning: NaN values detected in {n_bad} record(s). Skipping these instances.")
        if not bool(((host[0] != 0) & (host[1] == 0)).any()):   # every kept record left every stage finite
            if n_bad:
                x = x[kept]
            seg, _ = segment_ecg(x, None, seg_len)
            return (seg, kept) if return_kept else seg
        signals = signals[kept] if n_bad else signals          # a stage overflowed: the literal sequence below, on the kept records
    else:
        kept = nonfinite_records(signals) == 0
        n_bad = int((~kept).sum().item())
        if n_bad:
            print(f"Warning: NaN values detected in {n_bad} record(s). Skipping these instances.")
            signals = signals[kept]
    if signals.shape[0] == 0:
        n_out = int(signals.shape[1] * target_fs / orig_fs)
        seg = signals.new_zeros((0, n_out // seg_len, seg_len, signals.shape[2]))
        return (seg, kept) if return_kept else seg
    # Per-stage path: every stage's "all finite" test stays a device scalar and is read ONCE at the end.
    # A non-finite value anywhere (overflow of a filter: not seen on real records) sends the batch through the literal sequence below, check_nan_inf after
    # every stage as the reference has it.  The lead permutation commutes with every per-lead stage: it is applied last, to the resampled half-size data.
    x = signals
    if not planar_ok:
        flags = []
        x = advanced_ecg_filter(x, fs=orig_fs); flags.append(nonfinite_records(x, 1))
        x = wavelet_denoise(x); flags.append(nonfinite_records(x, 1))
        x = nsample_ecg(x, orig_fs, target_fs); flags.append(nonfinite_records(x, 1))
        if not bool(torch.cat(flags).any()):
            if reorder:
                x = reorder_indices(x).contiguous()
            seg, _ = segment_ecg(x, None, seg_len)
            return (seg, kept) if return_kept else seg
    x = check_nan_inf(signals, "reading")
    if reorder:
        x = check_nan_inf(reorder_indices(x).contiguous(), "reordering")
    x = check_nan_inf(advanced_ecg_filter(x, fs=orig_fs), "advanced filtering")
    x = check_nan_inf(wavelet_denoise(x), "wavelet denoising")
    x = check_nan_inf(nsample_ecg(x, orig_fs, target_fs), "resampling")
    seg, _ = segment_ecg(x, None, seg_len)
    seg = check_nan_inf(seg, "segmentation")
    return (seg, kept) if return_kept else seg


def nsample_ecg(ecg_data, orig_fs, target_fs):
    """preprocess_utils.py:90-101: cubic-spline resampling (scipy interp1d kind='cubic' = not-a-knot spline through every sample) to
    int(n * target_fs / orig_fs) equally spaced instants over the same span (ecgb_resample_cubic_f64)."""
    xb, single = _batch(ecg_data)
    R, n, leads = xb.shape
    m = int(n * target_fs / orig_fs)
    out = torch.empty((R, m, leads), dtype=torch.float64, device=xb.device)
    scratch = torch.empty(_L().ecgb_resample_cubic_scratch_bytes(R, n, leads) // 8, dtype=torch.float64, device=xb.device)
    _lib.check(_L().ecgb_resample_cubic_f64(C.c_void_p(xb.data_ptr()), C.c_void_p(out.data_ptr()), R, n, leads, m,
                                            C.c_void_p(scratch.data_ptr()), scratch.numel() * 8, _st()))
    return out[0] if single else out


def segment_ecg(ecg_data, text_data, seg_len):
    """preprocess_utils.py:103-116: whole segments of seg_len samples, the tail dropped; the text repeated per segment.  One record
    [n, leads] -> ([segments, seg_len, leads], [text] * segments); a batch -> [records, segments, seg_len, leads]."""
    n = ecg_data.shape[-2]
    num_segments = n // seg_len
    seg = ecg_data[..., : num_segments * seg_len, :]
    seg = seg.reshape(*ecg_data.shape[:-2], num_segments, seg_len, ecg_data.shape[-1])
    return seg, [text_data] * num_segments


# ---- the dataset-preparation driver around the conditioning (preprocess_utils.py:168-226, preprocess/preprocess_ecg.py:28-48) ------------------------------------
# The reference runs process_instance per record in a process pool, twice: once to gather the global statistics Q1 consumes (compute_global_stats), once to write
# the segment files ECGTokenDataset reads (process_and_save_instance).  Here a batch of raw records is conditioned once on the device (condition_records); these two
# functions turn its output into the same statistics dict and the same files.  Reading the wfdb records stays with the caller.

def compute_global_stats(segment_batches, sample_size=100000, skipped=0):
    """preprocess_utils.py:168-214 over conditioned segments.  `segment_batches`: an iterable of arrays / tensors [records, segments, seg_len, leads] (what
    condition_records returns, any device; records the reference would skip are simply absent, their number goes into `skipped`).  Global minimum and maximum are
    reduced where the data lives; the percentile sample is the reference's: segments in order, each contributing `np.random.choice(seg.size, k, replace=False)` of
    its values until `sample_size` are collected (whole segments but the last) -- the same calls on numpy's global generator, so `np.random.seed` reproduces the
    reference's draw -- and `np.percentile(samples, 1 / 99)` on the host (100 000 doubles).  Returns the dict the reference saves as `{data}_dataset_stats.npy`."""
    global_min, global_max = np.inf, -np.inf
    samples, collected = [], 0
    for batch in segment_batches:
        t = batch if torch.is_tensor(batch) else torch.as_tensor(np.asarray(batch))
        if t.dim() != 4:
            raise ValueError("compute_global_stats: batches of [records, segments, seg_len, leads]")
        if t.numel() == 0:
            continue
        lo, hi = torch.aminmax(t)
        global_min, global_max = min(global_min, float(lo)), max(global_max, float(hi))
        if collected < sample_size:
            flat = t.reshape(-1, t.shape[2], t.shape[3])                     # instances in order, their segments in order (the reference's two loops)
            need = -(-(sample_size - collected) // (t.shape[2] * t.shape[3]))
            host = flat[:need].cpu().numpy()                                 # only the segments the sample can still draw from
            for seg in host:
                if collected >= sample_size:
                    break
                k = min(sample_size - collected, seg.size)
                idx = np.random.choice(seg.size, k, replace=False)           # preprocess_utils.py:200
                samples.append(seg.flat[idx])
                collected += k
    if not samples:
        raise ValueError("compute_global_stats: no segments")
    samples = np.concatenate(samples)
    return {"global_min": global_min, "global_max": global_max, "percentile_1": np.percentile(samples, 1), "percentile_99": np.percentile(samples, 99),
            "skipped_instances": int(skipped)}


def save_dataset_stats(path, stats):
    """preprocess/preprocess_ecg.py:36: the dict as a pickled .npy -- the file `--percentiles` names and ECGTokenDataset opens (data_loader.py:47)."""
    np.save(path, stats)


def process_and_save_batch(segments, texts, first_index, split_name, data, seg_len, root="./data", kept=None):
    """process_and_save_instance (preprocess_utils.py:216-226) for a conditioned batch: record r of `segments` [records, n_seg, seg_len, leads] is instance
    first_index + r of the split (with `kept`, the mask condition_records returns, r counts the KEPT records and the instance index skips the dropped ones, as
    the reference's enumeration does); segment j goes to {root}/{data}_{seg_len}/ecg/{split}/ecg_{i}_{j}.npy as the (leads, seg_len) transpose the reference
    saves, its text to .../text/{split}/text_{i}_{j}.json.  `texts`: one entry per ORIGINAL instance (the conversation / [type, question, answer] list).
    Returns the instance indices written."""
    import json
    import os
    ecg_dir = os.path.join(root, f"{data}_{seg_len}", "ecg", split_name)
    text_dir = os.path.join(root, f"{data}_{seg_len}", "text", split_name)
    os.makedirs(ecg_dir, exist_ok=True)
    os.makedirs(text_dir, exist_ok=True)
    seg = segments.cpu().numpy() if torch.is_tensor(segments) else np.asarray(segments)
    if seg.ndim != 4 or seg.shape[2] != seg_len:
        raise ValueError("process_and_save_batch: segments [records, n_seg, seg_len, leads]")
    keep = np.ones(seg.shape[0], dtype=bool) if kept is None else np.asarray(kept.cpu() if torch.is_tensor(kept) else kept, dtype=bool)
    originals = np.flatnonzero(keep)
    if originals.size != seg.shape[0]:
        raise ValueError("process_and_save_batch: `kept` must mark exactly the records present in `segments`")
    written = []
    for r, orig in enumerate(originals):
        i = first_index + int(orig)
        for j in range(seg.shape[1]):
            np.save(os.path.join(ecg_dir, f"ecg_{i}_{j}.npy"), seg[r, j, :, :].T)          # shape (leads, seg_len), as preprocess_utils.py:221-223 saves it
            with open(os.path.join(text_dir, f"text_{i}_{j}.json"), "w") as f:
                json.dump(texts[int(orig)], f)
        written.append(i)
    return written
