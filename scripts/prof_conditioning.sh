#!/bin/bash
# Kernel times and SQ counters of the conditioning kernels (sequence-major pipeline, 4096 records): duration, clock, vector / LDS instruction counts, wait fractions.
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cp0 -- python3 scripts/dev_conditioning_pmc.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d gpurun_out/cp1 -- python3 scripts/dev_conditioning_pmc.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM --output-format csv -d gpurun_out/cp2 -- python3 scripts/dev_conditioning_pmc.py > /dev/null 2>&1
cp $(ls gpurun_out/cp0/*/*kernel_stats.csv | head -1) gpurun_out/conditioning_kernel_stats.csv
python3 - <<'PY' | tee gpurun_out/conditioning_pmc.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/cp[12]/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float); d = {}
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not any(t in k for t in ("filtfilt", "wavelet", "resample")): continue
        name = k.replace("(anonymous namespace)::", "").replace("void ", "")[:48]
        per[(name, r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
        d[(name, r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for (n, _, c), v in per.items(): acc[n][c].append(v)
    if "cp1" in f:
        for (n, _), us in d.items(): dur[n].append(us)
for n in acc:
    c = {k: sum(v) / len(v) for k, v in acc[n].items()}
    us = sum(dur[n]) / len(dur[n]) if dur[n] else 0
    busy = c.get("SQ_BUSY_CYCLES", 0) / 32
    wc = max(1, c.get("SQ_WAVE_CYCLES", 1))
    print(n)
    print(f"   {us:.0f} us (counter pass)  clock {busy / (us * 1e3) if us else 0:.2f} GHz  waves {c.get('SQ_WAVES', 0):.0f}  valu {c.get('SQ_INSTS_VALU', 0) / 1e6:.1f} M  lds {c.get('SQ_INSTS_LDS', 0) / 1e6:.1f} M  salu {c.get('SQ_INSTS_SALU', 0) / 1e6:.1f} M  vmem {c.get('SQ_INSTS_VMEM', 0) / 1e6:.2f} M")
    print(f"   valu_active/busy(simd) {c.get('SQ_ACTIVE_INST_VALU', 0) / 4 / 1024 / busy if busy else 0:.3f}  lds_active/busy(cu) {c.get('SQ_ACTIVE_INST_LDS', 0) / 4 / 256 / busy if busy else 0:.3f}  lds_conflict/lds_active {c.get('SQ_LDS_BANK_CONFLICT', 0) / max(1, c.get('SQ_ACTIVE_INST_LDS', 1)):.3f}  "
          f"wait_any/wave_cycles {c.get('SQ_WAIT_ANY', 0) / wc:.3f}  wait_inst_any {c.get('SQ_WAIT_INST_ANY', 0) / wc:.3f}  wait_inst_lds {c.get('SQ_WAIT_INST_LDS', 0) / wc:.3f}  wave_cycles/busy/1024 {wc / 4 / 1024 / busy if busy else 0:.2f} waves per SIMD")
PY
rm -rf gpurun_out/cp0 gpurun_out/cp1 gpurun_out/cp2
