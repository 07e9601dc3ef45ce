"""Dev-only: the spline resampler of two builds of the library on the same records, bit for bit (ecg_byte_amd/libecgbyte_hip_old.so is the other build), and their times."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from ecg_byte_amd import _lib
    if os.environ.get("ECGB_SO"):
        _lib.SO_PATH = os.path.join(os.path.dirname(_lib.SO_PATH), os.environ["ECGB_SO"])
    import numpy as np, torch, hashlib
    from ecg_byte_amd import preprocess_utils as pp
    rng = np.random.default_rng(1)
    for (R, n, fs_in, fs_out) in [(70, 5000, 500, 250), (9, 1000, 500, 250), (5, 640, 500, 360), (3, 777, 250, 500), (2, 64, 500, 250), (2, 65, 500, 125), (1, 5000, 500, 499)]:
        x = torch.from_numpy(rng.standard_normal((R, n, 12))).cuda()
        y = pp.nsample_ecg(x, fs_in, fs_out)
        out, flags, _ = (None, None, None)
        print(R, n, fs_in, fs_out, tuple(y.shape), hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:16])
    x = torch.from_numpy(rng.standard_normal((4096, 5000, 12))).cuda()
    for _ in range(2): pp.nsample_ecg(x, 500, 250)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): pp.nsample_ecg(x, 500, 250)
    e1.record(); torch.cuda.synchronize()
    print("TIME %.3f ms" % (e0.elapsed_time(e1) / 5))
else:
    outs = []
    for so in ("libecgbyte_hip_old.so", ""):
        env = dict(os.environ, ECGB_SO=so)
        r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        lines = [l for l in r.stdout.splitlines() if l and not l.startswith("TIME")]
        print(so or "current", [l for l in r.stdout.splitlines() if l.startswith("TIME")], r.stderr[-300:] if r.returncode else "")
        outs.append(lines)
    print("same bits" if outs[0] == outs[1] and outs[0] else "DIFFERENT", len(outs[0]))
    if outs[0] != outs[1]:
        for a, b in zip(*outs): print(a, "|", b)
