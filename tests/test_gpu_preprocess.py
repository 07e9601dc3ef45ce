"""GPU parity of the offline conditioning kernels (csrc/preprocess.hip) against the oracle / scipy on the same inputs and against the
committed golden vectors.  float64: the filter chain is compared bit for bit with the written-out recursion of the oracle and within
1e-11 of its maximum with scipy's compiled loop (the recursion amplifies a last-bit difference of the two builds)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
scipy = pytest.importorskip("scipy")

from oracle import preprocess_ref as P  # noqa: E402

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "preprocess_golden.npz"))


@pytest.fixture(scope="module")
def pp():
    assert torch.cuda.is_available()
    from ecg_byte_amd import preprocess_utils
    return preprocess_utils


def _records(R, n, seed):
    from ecg_byte_amd import synth
    rng = np.random.default_rng(seed)
    x = np.ascontiguousarray(synth.synth_ecg(R, n, seed=seed).transpose(0, 2, 1))
    return x + 0.02 * rng.standard_normal(x.shape) + 0.05 * np.sin(2 * np.pi * 60.0 * np.arange(n) / 500.0)[None, :, None] + rng.uniform(-1, 1, (R, 1, 12))


def test_filter_chain_golden(pp):
    got = pp.advanced_ecg_filter(torch.from_numpy(G["raw"][0]).cuda()).cpu().numpy()
    want = G["filtered"][0]
    assert np.abs(got - want).max() <= 1e-11 * np.abs(want).max()


def test_each_filter_is_the_written_out_recursion_bit_for_bit(pp):
    """One filter at a time, three leads: the kernel against oracle.filtfilt_literal (the same operations in the same order)."""
    x = _records(1, 700, seed=3)[0][:, :3]
    xd = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    for b, a in pp.design_filters():
        got = pp.filtfilt([(b, a)], xd).cpu().numpy()
        for lead in range(3):
            assert np.array_equal(got[:, lead], P.filtfilt_literal(b, a, x[:, lead])), lead


@pytest.mark.parametrize("R,n", [(1, 5000), (37, 5000), (5, 28), (3, 1237)])
def test_filter_chain_batches_against_scipy(pp, R, n):
    """Full-length records (5000 samples at 500 Hz), a ragged batch, the shortest signal scipy accepts (padlen 27 < n)."""
    x = _records(R, n, seed=R)
    got = pp.advanced_ecg_filter(torch.from_numpy(x).cuda()).cpu().numpy()
    for r in range(R):
        want = P.advanced_ecg_filter(x[r])
        assert np.abs(got[r] - want).max() <= 1e-10 * max(np.abs(want).max(), 1e-3), r


def test_resample_golden_and_batches(pp):
    """The not-a-knot spline in second-derivative form against scipy's B-spline solve: the same interpolant, rounding apart (1e-11 of
    the signal's range allowed; 1e-13 typical).  Golden record, the reference's 5000 -> 2500, an odd ratio, the shortest inputs."""
    got = pp.nsample_ecg(torch.from_numpy(G["filtered"][0]).cuda(), 500, 250).cpu().numpy()
    want = G["resampled"][0]
    assert got.shape == want.shape and np.abs(got - want).max() <= 1e-11 * np.abs(want).max()
    for R, n, fs_in, fs_out in [(3, 5000, 500, 250), (2, 1001, 500, 360), (2, 4, 500, 1000), (1, 5, 500, 250), (2, 6, 100, 250), (1, 7, 500, 500)]:
        x = _records(R, n, seed=n)
        got = pp.nsample_ecg(torch.from_numpy(x).cuda(), fs_in, fs_out).cpu().numpy()
        for r in range(R):
            want = P.nsample_ecg(x[r], fs_in, fs_out)
            assert got[r].shape == want.shape
            assert np.abs(got[r] - want).max() <= 1e-11 * max(np.abs(want).max(), 1e-3), (n, r, np.abs(got[r] - want).max())


@pytest.mark.parametrize("R,n", [(1, 5000), (9, 5000), (2, 640), (3, 98)])
def test_wavelet_denoise_against_the_restatement(pp, R, n):
    """csrc/preprocess.hip against oracle.wavelet_denoise (an independent numpy restatement; PyWavelets itself is not available: parity
    unpinned) -- same decomposition, same median / threshold, same reconstruction to 1e-11 of the signal's range."""
    x = _records(R, n, seed=7 * R + n)
    got = pp.wavelet_denoise(torch.from_numpy(x).cuda()).cpu().numpy()
    for r in range(R):
        want = P.wavelet_denoise(x[r])
        assert np.abs(got[r] - want).max() <= 1e-11 * max(np.abs(want).max(), 1e-3), (r, np.abs(got[r] - want).max())


@pytest.mark.parametrize("R,n,L", [(9, 5000, 12), (70, 640, 12), (3, 98, 12), (5, 1000, 3), (2, 10000, 12), (1, 12000, 2)])
def test_wavelet_workgroup_kernel_is_the_lane_kernel_bit_for_bit(pp, R, n, L):
    """The stage has two kernels -- one workgroup per sequence with the bands in LDS, one lane per sequence through scratch memory (the only one for sequences
    too long for LDS: n = 12 000 takes it either way).  Same sums in the same order, the same order statistics: the same bits, NaN / constant / zero leads included."""
    x = _records(R, n, seed=R + n)[:, :, :L].copy()
    x[0, :, 0] = 0.0
    if L > 1:
        x[0, :, 1] = 0.5
    if R > 1:
        x[1, n // 3, L - 1] = np.nan
    xd = torch.from_numpy(x).cuda()
    try:
        pp.set_wavelet_workgroup_kernel(False)
        lane = pp.wavelet_denoise(xd).clone()
    finally:
        pp.set_wavelet_workgroup_kernel(True)
    wg = pp.wavelet_denoise(xd)
    assert torch.equal(lane.view(torch.int64), wg.view(torch.int64))
    buf = xd.clone()                                                                    # in place: x_dev == y_dev
    from ecg_byte_amd import _lib
    import ctypes as C
    lib = _lib.lib()
    nbytes = lib.ecgb_wavelet_denoise_scratch_bytes(R, n, L)
    scratch = torch.empty(nbytes // 8, dtype=torch.float64, device="cuda")
    _lib.check(lib.ecgb_wavelet_denoise_f64(C.c_void_p(buf.data_ptr()), C.c_void_p(buf.data_ptr()), R, n, L, 1e-10, C.c_void_p(scratch.data_ptr()), nbytes,
                                            C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    assert torch.equal(buf.view(torch.int64), wg.view(torch.int64))


def test_wavelet_denoise_edge_cases(pp):
    z = torch.zeros(2, 640, 12, dtype=torch.float64, device="cuda")
    assert bool((pp.wavelet_denoise(z) == 0).all())                                     # median 0 -> threshold 0 -> 0 / 0 guarded
    c = torch.full((640, 12), 0.25, dtype=torch.float64, device="cuda")
    assert float((pp.wavelet_denoise(c) - 0.25).abs().max()) < 1e-10
    xb = _records(1, 640, seed=2)[0]
    xb[100, 3] = np.nan                                                                 # np.median is NaN then: every detail of the lead is
    out = pp.wavelet_denoise(torch.from_numpy(xb).cuda()).cpu().numpy()                 # zeroed, the NaN's neighbourhood becomes 0
    want = P.wavelet_denoise(xb)
    assert np.isfinite(out).all() and np.abs(out - want).max() <= 1e-11 * np.abs(want).max()
    assert (out[95:106, 3] == 0).all() and (want[95:106, 3] == 0).all()
    from ecg_byte_amd._lib import EcgbError
    with pytest.raises(EcgbError):
        pp.wavelet_denoise(torch.zeros(641, 12, dtype=torch.float64, device="cuda"))


@pytest.mark.parametrize("R,n,reorder", [(16, 5000, True), (9, 5000, False), (7, 1000, True), (64, 640, True)])
def test_sequence_major_pipeline_is_the_three_stage_calls_bit_for_bit(pp, R, n, reorder):
    """condition_records' fast path keeps its intermediates [records * leads][n] (ecgb_filtfilt_planar_f64 -> ecgb_wavelet_denoise_planar_f64 ->
    ecgb_resample_cubic_planar_f64, lead reorder in the last store): the same bits as the per-stage calls with [records, n, leads] between them, whole and
    partial last waves of sequences."""
    x = torch.from_numpy(_records(R, n, seed=3 * R + n)).cuda()
    try:
        pp.set_planar_pipeline(False)
        want = pp.condition_records(x, reorder=reorder, seg_len=n // 4)
    finally:
        pp.set_planar_pipeline(True)
    got = pp.condition_records(x, reorder=reorder, seg_len=n // 4)
    assert got.shape == want.shape and torch.equal(got.view(torch.int64), want.view(torch.int64))
    # the stages one by one against the public per-stage functions
    planar_out, flags, raw_flags = pp._condition_planar(x, 500, 250, None)
    ref = pp.nsample_ecg(pp.wavelet_denoise(pp.advanced_ecg_filter(x)), 500, 250)
    assert torch.equal(planar_out.view(torch.int64), ref.view(torch.int64)) and not bool(flags.any()) and not bool(raw_flags.any())


def test_sequence_major_pipeline_flags_records_that_leave_a_stage_not_finite(pp):
    """The kernels of the fast path raise a record's flag where check_nan_inf's test (preprocess_utils.py:26-33) would have fired: condition_records then runs the
    literal stage-by-stage sequence and returns what the per-stage path returns."""
    x = _records(4, 1000, seed=11)
    x[2, 500, 4] = 1e308                                                                # finite on entry; the filter chain overflows around it
    xd = torch.from_numpy(x).cuda()
    _, flags, raw_flags = pp._condition_planar(xd, 500, 250, None)
    assert not bool(raw_flags.any())
    per_stage = pp.advanced_ecg_filter(xd)
    assert flags.cpu().tolist() == [0, 0, int(not bool(torch.isfinite(per_stage[2]).all())), 0]
    try:
        pp.set_planar_pipeline(False)
        want = pp.condition_records(xd, seg_len=250)
    finally:
        pp.set_planar_pipeline(True)
    got = pp.condition_records(xd, seg_len=250)
    assert torch.equal(got.view(torch.int64), want.view(torch.int64))
    # ... and with a record that comes in with a NaN beside the one that overflows: dropped, the overflowing one still sends the rest through the literal sequence
    x[0, 10, 0] = np.nan
    xd = torch.from_numpy(x).cuda()
    _, flags, raw_flags = pp._condition_planar(xd, 500, 250, None)
    assert raw_flags.cpu().tolist() == [1, 0, 0, 0] and flags.cpu().tolist()[1::2] == [0, 0]
    try:
        pp.set_planar_pipeline(False)
        want, want_kept = pp.condition_records(xd, seg_len=250, return_kept=True)
    finally:
        pp.set_planar_pipeline(True)
    got, kept = pp.condition_records(xd, seg_len=250, return_kept=True)
    assert kept.cpu().tolist() == want_kept.cpu().tolist() == [False, True, True, True]
    assert torch.equal(got.view(torch.int64), want.view(torch.int64))


def test_condition_records_is_the_reference_pipeline(pp):
    """process_instance's order of operations on a batch: reorder, filter chain, wavelet shrinkage, 500 -> 250 Hz, segments."""
    x = _records(3, 5000, seed=4)
    got = pp.condition_records(torch.from_numpy(x).cuda(), reorder=True, seg_len=1250).cpu().numpy()
    assert got.shape == (3, 2, 1250, 12)
    for r in range(3):
        y = P.reorder_indices(x[r])
        y = P.nsample_ecg(P.wavelet_denoise(P.advanced_ecg_filter(y)), 500, 250)
        want, _ = P.segment_ecg(y, None, 1250)
        assert np.abs(got[r] - want).max() <= 1e-9 * np.abs(want).max()


def test_records_with_nan_or_inf_in_the_raw_signal_are_skipped_like_the_reference(pp):
    """preprocess_utils.py:134-136: an instance whose raw signal holds NaN / inf is skipped before check_nan_inf could zero-fill it."""
    x = _records(4, 5000, seed=5)
    bad = x.copy()
    bad[1, 17, 3] = np.nan
    bad[3, 4000, 11] = np.inf
    clean = pp.condition_records(torch.from_numpy(x).cuda(), reorder=True, seg_len=1250)
    got, kept = pp.condition_records(torch.from_numpy(bad).cuda(), reorder=True, seg_len=1250, return_kept=True)
    assert kept.cpu().tolist() == [True, False, True, False]
    assert got.shape == (2, 2, 1250, 12) and bool(torch.isfinite(got).all())
    assert torch.equal(got, clean[[0, 2]])                                  # the kept records are untouched by their neighbours
    none, kept = pp.condition_records(torch.from_numpy(bad[[1, 3]]).cuda(), return_kept=True)
    assert none.shape == (0, 2, 1250, 12) and not bool(kept.any())


def test_short_signals_are_refused_like_scipy(pp):
    from ecg_byte_amd._lib import EcgbError
    with pytest.raises(EcgbError):
        pp.advanced_ecg_filter(torch.zeros(27, 12, dtype=torch.float64, device="cuda"))


def test_segment_reorder_nan(pp):
    x = torch.arange(2500 * 12, dtype=torch.float64, device="cuda").reshape(2500, 12)
    seg, txt = pp.segment_ecg(x, "q", 1000)
    ref, _ = P.segment_ecg(x.cpu().numpy(), "q", 1000)
    assert txt == ["q", "q"] and np.array_equal(seg.cpu().numpy(), ref)
    assert np.array_equal(pp.reorder_indices(x).cpu().numpy(), P.reorder_indices(x.cpu().numpy()))
    bad = x.clone(); bad[3, 2] = float("nan"); bad[4, 1] = float("inf")
    assert np.array_equal(pp.check_nan_inf(bad, "t").cpu().numpy(), P.check_nan_inf(bad.cpu().numpy()))


def test_nonfinite_records_is_isfinite_all():
    """ecgb_nonfinite_records_f64 (the finite tests of condition_records in one pass each) = ~torch.isfinite(x).all() per record: NaN of either sign, both infinities,
    denormals and huge finite values, at the ends of a record and of a chunk; one flag for a whole tensor."""
    from ecg_byte_amd import preprocess_utils as pp
    g = torch.Generator(device="cuda").manual_seed(4)
    x = torch.randn(37, 5000, 12, device="cuda", dtype=torch.float64, generator=g)
    x[3, 0, 0] = float("nan"); x[5, 4999, 11] = float("inf"); x[9, 2500, 7] = float("-inf"); x[11, 683, 3] = -float("nan")
    x[20, 1, 1] = 1.7e308; x[21, 2, 2] = 5e-324; x[22, 8191 // 12, 8191 % 12] = float("inf")
    want = ~torch.isfinite(x).reshape(37, -1).all(dim=1)
    assert torch.equal(pp.nonfinite_records(x) != 0, want)
    assert int(pp.nonfinite_records(x, 1)) == 1 and int(pp.nonfinite_records(x[12:20].contiguous(), 1)) == 0
    y = torch.randn(3, 7, device="cuda", dtype=torch.float64)                    # records shorter than a wave
    y[2, 6] = float("nan")
    assert pp.nonfinite_records(y).tolist() == [0, 0, 1]
