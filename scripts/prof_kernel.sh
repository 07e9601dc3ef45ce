#!/bin/bash
# SQ counters of every kernel whose name contains $1, from `python3 $2`:  bash scripts/prof_kernel.sh lora_dx scripts/dev_lora_dx_only.py
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
PAT=$1; SCRIPT=$2
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU --output-format csv -d gpurun_out/k_pmc1 -- python3 $SCRIPT > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/k_pmc2 -- python3 $SCRIPT > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/k_trace -- python3 $SCRIPT > /dev/null 2>&1
PAT=$PAT python3 - <<'PY'
import csv, glob, collections, os, re
pat = os.environ["PAT"]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/k_pmc*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if pat not in k: continue
        m = re.search(r"(\w+<[^>]*>)", k)
        name = (m.group(1) if m else k[:60]) + " grid" + row["Grid_Size"] + " vgpr" + row["VGPR_Count"]
        per[(name, row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
    for (name, d, c), v in per.items(): acc[name][c].append(v)
for name in sorted(acc):
    print(name)
    for c in sorted(acc[name]):
        v = acc[name][c]; print(f"   {c:28s} {sum(v)/len(v):16.0f}")
for f in glob.glob("gpurun_out/k_trace/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if pat in row["Name"]: print("   trace:", row["Name"][:90], "calls", row["Calls"], "avg us", float(row["AverageNs"]) / 1e3)
PY
find gpurun_out/k_pmc1 gpurun_out/k_pmc2 gpurun_out/k_trace -type f -delete
