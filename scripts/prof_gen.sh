#!/bin/bash
# rocprofv3 kernel statistics of the C5 generate (Gemma-2B dims, LoRA r16 with LORA=1, 600-token prompt + 128 new tokens, batch 1) -> gpurun_out/prof_<tag>/
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
TAG=${1:-gen}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o gen -- python3 scripts/dev_gen_only.py 1 > gpurun_out/prof_$TAG.log 2> gpurun_out/prof_$TAG.err
find gpurun_out/prof_$TAG -type f ! -name "*kernel_stats.csv" -delete
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/prof_$TAG/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:22]:
    print(f"{r['Name'][:100]:100s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:8.1f} tot_ms={float(r['TotalDurationNs'])/1e6:8.1f}")
PY
