"""Dev-only: the offline conditioning on a batch of 12-lead, 5000-sample records: device time per stage against the
scipy path the reference runs (oracle/preprocess_ref.py, one core; its wavelet stage is the numpy restatement, slower than pywt would be)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from ecg_byte_amd import preprocess_utils as pp, synth
from oracle import preprocess_ref as P

R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
base = np.ascontiguousarray(synth.synth_ecg(64, 5000, seed=0).transpose(0, 2, 1))
x = np.concatenate([base] * (R // 64)) + 0.01 * np.random.default_rng(0).standard_normal((R, 5000, 12))
xd = torch.from_numpy(x).cuda()


def timed(f, n=3):
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): out = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n, out


tf, yf = timed(lambda: pp.advanced_ecg_filter(xd))
tw, yw = timed(lambda: pp.wavelet_denoise(yf))
tr, yr = timed(lambda: pp.nsample_ecg(yw, 500, 250))
tall, _ = timed(lambda: pp.condition_records(xd, reorder=True, seg_len=1250))
print(f"{R} records of 5000 x 12 float64 ({x.nbytes / 1e9:.2f} GB): filter chain {tf * 1e3:.1f} ms, wavelet {tw * 1e3:.1f} ms, resample {tr * 1e3:.1f} ms; "
      f"whole pipeline {tall * 1e3:.1f} ms = {R / tall:.0f} records/s")
t = time.perf_counter(); f = P.advanced_ecg_filter(x[0]); t1 = time.perf_counter() - t
t = time.perf_counter(); P.nsample_ecg(f, 500, 250); t2 = time.perf_counter() - t
print(f"CPU (scipy, 1 core, one record): filter chain {t1 * 1e3:.1f} ms, resample {t2 * 1e3:.1f} ms  -> {1 / (t1 + t2):.0f} records/s without the wavelet stage")

# the glue of condition_records, piece by piece
tk, _ = timed(lambda: torch.isfinite(xd).all(dim=2).all(dim=1).sum().item())
t1f, _ = timed(lambda: torch.isfinite(yf).all())
t1r, _ = timed(lambda: torch.isfinite(yr).all())
tro, _ = timed(lambda: pp.reorder_indices(yr).contiguous())
print(f"glue: raw-record finite test {tk * 1e3:.2f} ms, finite test of a stage's output {t1f * 1e3:.2f} ms (full size) / {t1r * 1e3:.2f} ms (resampled), lead reorder {tro * 1e3:.2f} ms; "
      f"stages {1e3 * (tf + tw + tr):.1f} ms + glue {1e3 * (tk + 2 * t1f + t1r + tro):.1f} ms against {tall * 1e3:.1f} ms measured")
