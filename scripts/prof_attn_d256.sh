#!/bin/bash
# per-kernel average durations of the head_dim-256 attention kernels at the C5 shape, one rocprofv3 kernel trace per build of the library given as arguments
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for so in "$@"; do
  d=gpurun_out/attn_d256_$RANDOM
  ROUNDS=2 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 scripts/dev_attn_d256_ab.py $so > /dev/null 2>&1
  echo "== $so"
  python3 - $d <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "attn_" in row["Name"]:
            print(f"{row['Name'][:80]:80s} calls {row['Calls']:>5s}  avg {float(row['AverageNs']) / 1e3:8.1f} us")
PY
done
