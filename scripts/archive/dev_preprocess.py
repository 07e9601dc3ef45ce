"""Dev-only: the offline conditioning on a batch of 12-lead, 5000-sample records: device time per stage against the
scipy path the reference runs (oracle/preprocess_ref.py, one core; its wavelet stage is the numpy restatement, slower than pywt would be)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ecg_byte_amd import _lib
if os.environ.get("ECGB_SO"):
    _lib.SO_PATH = os.path.join(os.path.dirname(_lib.SO_PATH), os.environ["ECGB_SO"])   # A/B against another build
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

# the sequence-major fast path, stage by stage
import ctypes as C
from ecg_byte_amd import _lib as L_
lib = L_.lib()
n, leads = 5000, 12
taps, b, a, zi = pp._pack_filters(pp.design_filters(500))
nbytes = lib.ecgb_filtfilt_scratch_bytes(R, n, leads, 3 * max(taps)); nres = lib.ecgb_resample_cubic_scratch_bytes(R, n, leads)
scratch = torch.empty(max(nbytes, nres) // 8, dtype=torch.float64, device="cuda")
flags = torch.zeros(R, dtype=torch.uint8, device="cuda")
planar = torch.empty(R * leads * n, dtype=torch.float64, device="cuda"); out = torch.empty((R, 2500, leads), dtype=torch.float64, device="cuda")
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
vp = lambda t: C.c_void_p(t.data_ptr())
t1, _ = timed(lambda: L_.check(lib.ecgb_filtfilt_planar_f64(vp(xd), vp(planar), R, n, leads, 4, taps, b.ctypes.data_as(C.c_void_p), a.ctypes.data_as(C.c_void_p), zi.ctypes.data_as(C.c_void_p), vp(scratch), nbytes, vp(flags), None, st())))
t2, _ = timed(lambda: L_.check(lib.ecgb_wavelet_denoise_planar_f64(vp(planar), vp(planar), R, n, leads, 1e-10, st())))
t3, _ = timed(lambda: L_.check(lib.ecgb_resample_cubic_planar_f64(vp(planar), vp(out), R, n, leads, 2500, None, vp(scratch), nres, vp(flags), st())))
t4, _ = timed(lambda: pp._condition_planar(xd, 500, 250, None))
t5, _ = timed(lambda: pp.nonfinite_records(xd))
print(f"sequence-major: filter chain {t1 * 1e3:.2f} ms, wavelet {t2 * 1e3:.2f} ms, resample {t3 * 1e3:.2f} ms; the three with their allocations {t4 * 1e3:.2f} ms; raw-record test {t5 * 1e3:.2f} ms")
