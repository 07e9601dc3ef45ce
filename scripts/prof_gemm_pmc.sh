#!/bin/bash
# SQ counters of the eight-wave NT kernel, the four-wave kernel and hipBLASLt's kernel on one shape (SHAPE=M,N,K): MFMA busy, clock, LDS conflicts, instruction mix.
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/gp1 -- python3 scripts/dev_gemm_pmc.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/gp2 -- python3 scripts/dev_gemm_pmc.py > /dev/null 2>&1
python3 - <<'EOF'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/gp*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float); d = {}
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not (k.startswith("Cijk") or "gemm_nt" in k or "Custom" in k): continue
        name = k.replace("(anonymous namespace)::", "").replace("void ", "")[:60]
        per[(name, r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
        d[(name, r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for (n, _, c), v in per.items(): acc[n][c].append(v)
    if "gp1" in f:
        for (n, _), us in d.items(): dur[n].append(us)
for n in acc:
    c = {k: sum(v[2:]) / max(1, len(v[2:])) for k, v in acc[n].items()}
    us = sum(dur[n][2:]) / max(1, len(dur[n][2:])) if dur[n] else 0
    busy = c.get("SQ_BUSY_CYCLES", 0) / 32
    print(n)
    print(f"   {us:.1f} us  clock {busy / (us * 1e3) if us else 0:.2f} GHz  mfma_busy {c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024 / busy if busy else 0:.3f}  "
          f"valu/mfma {c.get('SQ_INSTS_VALU', 0) / max(1, c.get('SQ_INSTS_MFMA', 1)):.2f}  lds/mfma {c.get('SQ_INSTS_LDS', 0) / max(1, c.get('SQ_INSTS_MFMA', 1)):.3f}  "
          f"salu/mfma {c.get('SQ_INSTS_SALU', 0) / max(1, c.get('SQ_INSTS_MFMA', 1)):.2f}  vmem/mfma {c.get('SQ_INSTS_VMEM', 0) / max(1, c.get('SQ_INSTS_MFMA', 1)):.3f}  "
          f"lds_conflict/lds_active {c.get('SQ_LDS_BANK_CONFLICT', 0) / max(1, c.get('SQ_ACTIVE_INST_LDS', 1)):.3f}  lds_active_cycles/busy {c.get('SQ_ACTIVE_INST_LDS', 0) / 4 / 256 / busy if busy else 0:.3f}  "
          f"wait_any/wave_cycles {c.get('SQ_WAIT_ANY', 0) / max(1, c.get('SQ_WAVE_CYCLES', 1)):.3f}  wait_inst_lds/wave {c.get('SQ_WAIT_INST_LDS', 0) / max(1, c.get('SQ_WAVE_CYCLES', 1)):.3f}")
EOF
rm -rf gpurun_out/gp1 gpurun_out/gp2
