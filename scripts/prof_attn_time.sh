#!/bin/bash
# per-kernel average durations of the head_dim-64 attention kernels at the C3 shape (rocprofv3 kernel trace over scripts/dev_attn_time.py)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
rm -rf gpurun_out/attn_time
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/attn_time -- python3 scripts/dev_attn_time.py 4 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/attn_time/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "attn_" in row["Name"]:
            print(f"{row['Name'][:90]:90s} calls {row['Calls']:>5s}  avg {float(row['AverageNs']) / 1e3:8.1f} us")
PY
