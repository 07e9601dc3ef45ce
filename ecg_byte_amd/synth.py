"""Seeded synthetic 12-lead ECG generator (SURVEY.md §8d "Synthetic encode inputs").

No dataset is reachable from the build/GPU boxes, so every parity test, fixture and
bench run draws its `(B, 12, L)` float64 signals from this generator.  Record `i` of a
given `seed` is a pure function of `(seed, i, L, fs)` -- independent of the batch size it
is drawn in -- so a shard of the batch on rank r sees exactly the records the
single-GPU run sees at the same indices.

Per record: heart rate U(55,100) bpm; per lead a gain U(0.3,1.5); per beat five Gaussian
bumps (P, Q, R, S, T); baseline wander 0.05*sin(2*pi*0.3*t + phi); white noise
N(0, 0.01^2).  Units mV.  Layout matches the reference's on-disk segments: float64,
C-order, leads first (reference: ecg_byte/utils/preprocess_utils.py:219-223).
"""
from __future__ import annotations

import numpy as np

LEADS = 12

# (amplitude mV, centre as fraction of the beat, width in seconds)
_BUMPS = (
    (0.15, 0.20, 0.025),   # P
    (-0.10, 0.345, 0.010),  # Q
    (1.00, 0.37, 0.012),   # R
    (-0.20, 0.395, 0.010),  # S
    (0.30, 0.60, 0.045),   # T
)


def _record(seed: int, index: int, L: int, fs: float) -> np.ndarray:
    rng = np.random.default_rng([int(seed), int(index)])
    hr = rng.uniform(55.0, 100.0)
    gains = rng.uniform(0.3, 1.5, size=LEADS)
    phi = rng.uniform(0.0, 2.0 * np.pi, size=LEADS)
    t0 = rng.uniform(0.0, 1.0)                      # beat phase offset
    noise = rng.standard_normal((LEADS, L))
    period = 60.0 / hr
    t = np.arange(L, dtype=np.float64) / fs
    tau = np.mod(t / period + t0, 1.0)              # fraction of the beat
    beat = np.zeros(L, dtype=np.float64)
    for amp, mu, sig in _BUMPS:
        d = (tau - mu) * period                      # seconds from bump centre
        beat += amp * np.exp(-0.5 * (d / sig) ** 2)
    wander = 0.05 * np.sin(2.0 * np.pi * 0.3 * t[None, :] + phi[:, None])
    return gains[:, None] * beat[None, :] + wander + 0.01 * noise


def synth_ecg(batch: int, L: int, seed: int = 0, fs: float | None = None,
              start: int = 0) -> np.ndarray:
    """Return `(batch, 12, L)` float64 records `start .. start+batch-1` of `seed`.

    `fs` defaults to L/10 Hz (10-second strips: L=1000 -> 100 Hz, L=5000 -> 500 Hz).
    """
    if fs is None:
        fs = L / 10.0
    out = np.empty((batch, LEADS, L), dtype=np.float64)
    for b in range(batch):
        out[b] = _record(seed, start + b, L, fs)
    return out


def synth_percentiles(L: int, seed: int = 0, n_samples: int = 100_000) -> dict:
    """Global percentile dict in the reference's on-disk shape
    (reference: ecg_byte/utils/preprocess_utils.py:197-210): 1st/99th percentile of
    `n_samples` values drawn from the same generator."""
    rng = np.random.default_rng([int(seed), 0xEC6])
    need = n_samples
    chunks = []
    idx = 1_000_000          # records disjoint from any batch index used in tests
    gmin, gmax = np.inf, -np.inf
    while need > 0:
        rec = _record(seed, idx, L, L / 10.0)
        gmin, gmax = min(gmin, rec.min()), max(gmax, rec.max())
        take = min(need, rec.size // 4)
        chunks.append(rec.flat[rng.choice(rec.size, take, replace=False)])
        need -= take
        idx += 1
    s = np.concatenate(chunks)
    return {
        "global_min": float(gmin),
        "global_max": float(gmax),
        "percentile_1": float(np.percentile(s, 1)),
        "percentile_99": float(np.percentile(s, 99)),
        "skipped_instances": 0,
    }
