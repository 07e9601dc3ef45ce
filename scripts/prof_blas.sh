#!/bin/bash
# What hipBLASLt (torch.matmul) launches for the C3 GEMM shapes, next to this repo's kernels: times first, then a kernel trace with registers / LDS per kernel.
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 300 python3 scripts/dev_blas_trace.py 2>&1 | grep -v amdgpu.ids
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/blas -- python3 scripts/dev_blas_trace.py > /dev/null 2>&1
f=$(find gpurun_out/blas -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'EOF'
import csv, sys, collections
seen = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "Cijk" in k or "gemm" in k.lower():
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        e = seen.setdefault(k, dict(n=0, t=0.0, r=r))
        e["n"] += 1; e["t"] += d
for k, e in seen.items():
    r = e["r"]
    print(k[:260]); print("    calls", e["n"], "avg us", round(e["t"] / e["n"], 1), {c: r[c] for c in r if any(w in c for w in ("Workgroup_Size", "Grid_Size", "LDS", "VGPR", "SGPR", "Scratch"))})
EOF
rm -rf gpurun_out/blas
