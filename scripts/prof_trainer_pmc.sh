#!/bin/bash
# HBM traffic of the tokenizer trainer's three kernels on the bench leg's own corpus (CORPUS=c2): FETCH_SIZE and WRITE_SIZE in passes of their own (--pmc only with
# --kernel-trace), summarised per kernel and launch by scripts/pmc_summary.py -> gpurun_out/${P}_trainer_hbm.json (bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024, the
# guide's gfx950 reading of 16-byte-per-lane fetches; the trainer's fetches are 16-byte and 4-byte, so the doubled figure is an upper bound for it)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
P=${1:-r04}
export CORPUS=c2
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/${P}_trainer_$c -- python3 scripts/dev_trainer_prof.py > gpurun_out/${P}_trainer_$c.log 2>&1
done
python3 - <<PY > gpurun_out/${P}_trainer_hbm.json
import json, subprocess, sys
out = {}
for k in ("seg_merge_kernel", "seg_init_kernel", "rewrite_kernel", "tile_count_kernel", "rowmax_kernel", "init_kernel"):
    try:
        out[k] = json.loads(subprocess.run([sys.executable, "scripts/pmc_summary.py", k, "gpurun_out/${P}_trainer_FETCH_SIZE", "gpurun_out/${P}_trainer_WRITE_SIZE"], capture_output=True, text=True).stdout)["counters"]
    except Exception:
        continue          # (a kernel the form that ran does not launch)
tot = 0.0
for k, c in out.items():
    per = (2.0 * c.get("FETCH_SIZE", {}).get("per_launch_mean", 0.0) + c.get("WRITE_SIZE", {}).get("per_launch_mean", 0.0)) * 1024.0
    n = c.get("FETCH_SIZE", {}).get("n", 0) / 2          # dev_trainer_prof.py trains twice
    c["bytes_per_launch_upper"] = per; c["launches_per_run"] = n
    tot += per * n
out["bytes_per_run_upper"] = tot
print(json.dumps(out, indent=1))
PY
cat gpurun_out/${P}_trainer_hbm.json | head -60
find gpurun_out/${P}_trainer_FETCH_SIZE gpurun_out/${P}_trainer_WRITE_SIZE -type f -delete
