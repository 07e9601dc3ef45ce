#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU --output-format csv -d gpurun_out/attn_pmc1 -- python3 ${ATTN_SCRIPT:-scripts/dev_attn_gemma.py} > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/attn_pmc2 -- python3 ${ATTN_SCRIPT:-scripts/dev_attn_gemma.py} > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/attn_pmc*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "attn_" not in k: continue
        import re
        name = re.search(r"(attn_\w+<[^>]*>)", k).group(1) + " grid" + row["Grid_Size"]
        per[(name, row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
    for (name, d, c), v in per.items(): acc[name][c].append(v)
for name in sorted(acc):
    print(name)
    for c in sorted(acc[name]):
        v = acc[name][c]; print(f"   {c:28s} {sum(v)/len(v):16.0f}")
PY
find gpurun_out/attn_pmc1 gpurun_out/attn_pmc2 -type f -delete
