#!/bin/bash
# SQ counters of the dominant kernels of the C3 train step (full fine-tune leg), two --pmc passes + one kernel trace of the
# SAME command; per (kernel, grid) means -> gpurun_out/train_pmc_${TAG}.json.  Usage: bash scripts/prof_train_pmc.sh TAG [extra bench flags]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export ECGB_BENCH_WORKERS=1   # no fork pool under the profiler (a pool child hung one --pmc pass for ten minutes)
TAG=${1:-r03}; shift
CMD="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-c1 --no-c5 --no-lora-leg --no-extras --train-steps 2 $*"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/tp_pmc1 -- $CMD > gpurun_out/tp_pmc1.json 2> gpurun_out/tp_pmc1.err
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/tp_pmc2 -- $CMD > gpurun_out/tp_pmc2.json 2> gpurun_out/tp_pmc2.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tp_trace -- $CMD > gpurun_out/tp_trace.json 2> gpurun_out/tp_trace.err
TAG=$TAG python3 - <<'PY'
import csv, glob, collections, os, re, json
tag = os.environ["TAG"]
want = ("gemm_", "attn_", "ce_fwd", "rmsnorm", "adam", "encode_flow")
def short(k):
    k = k.replace("(anonymous namespace)::", "").replace("void ", "")
    return k.split("(")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
pmc_one, pmc_dur = {}, collections.defaultdict(list)
for f in glob.glob("gpurun_out/tp_pmc*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if not any(w in k for w in want): continue
        name = short(k) + " grid" + row["Grid_Size"] + " vgpr" + row["VGPR_Count"] + " lds" + row["LDS_Block_Size"]
        per[(name, row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
        if "pmc1" in f: pmc_one[(name, row["Dispatch_Id"])] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
    for (name, d, c), v in per.items(): acc[name][c].append(v)
for (name, d), us in pmc_one.items(): pmc_dur[name].append(us)
dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/tp_trace/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if not any(w in k for w in want): continue
        name = short(k) + " grid" + str(int(row["Grid_Size_X"]) * int(row["Grid_Size_Y"]) * int(row["Grid_Size_Z"]))
        dur[name].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
out = {}
for name in sorted(acc):
    c = {k: sum(v) / len(v) for k, v in acc[name].items()}
    c["launches_seen"] = max(len(v) for v in acc[name].values())
    key = name.split(" vgpr")[0]
    if key in dur: c["trace_avg_us"] = sum(dur[key]) / len(dur[key]); c["trace_calls"] = len(dur[key])
    # SQ_VALU_MFMA_BUSY_CYCLES: cycles, summed over the 1024 SIMDs (= 16 x the number of 16x16x32 bf16 MFMAs); SQ_BUSY_CYCLES: cycles, summed over the
    # 32 shader engines -- so SQ_BUSY_CYCLES / 32 is the dispatch's length in shader cycles: the clock it held, and the MFMA pipes' busy fraction
    pd = pmc_dur.get(name)
    if pd: c["pmc_pass_avg_us"] = sum(pd) / len(pd)
    if c.get("SQ_BUSY_CYCLES"): c["mfma_busy"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / (c["SQ_BUSY_CYCLES"] / 32)
    if c.get("SQ_INSTS_MFMA"): c["valu_per_mfma"] = c.get("SQ_INSTS_VALU", 0) / c["SQ_INSTS_MFMA"]
    if c.get("SQ_BUSY_CYCLES") and pd: c["clock_ghz_pmc_pass"] = c["SQ_BUSY_CYCLES"] / 32 / (c["pmc_pass_avg_us"] * 1e3)
    out[name] = c
json.dump(out, open(f"gpurun_out/train_pmc_{tag}.json", "w"), indent=1)
for name, c in out.items():
    print(name, {k: (round(v, 3) if isinstance(v, float) and v < 100 else int(v)) for k, v in c.items() if k in ("mfma_busy", "valu_per_mfma", "trace_avg_us", "clock_ghz_pmc_pass", "launches_seen")})
PY
for f in $(find gpurun_out/tp_trace -name "*kernel_stats.csv"); do cp $f gpurun_out/train_kernel_stats_${TAG}.csv; done
rm -rf gpurun_out/tp_pmc1 gpurun_out/tp_pmc2 gpurun_out/tp_trace
