#!/bin/bash
# SQ counters of the head_dim-256 attention kernels at the C5 shape, one --pmc pass per counter set and library build given as argument (files in ecg_byte_amd/):
# MFMA-pipe busy fraction, vector instructions per MFMA, the clock the dispatch held -> gpurun_out/attn_d256_pmc.json.  Usage: bash scripts/prof_attn_d256_pmc.sh lib.so [lib2.so ...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
rm -rf gpurun_out/ad_pmc*
for so in "$@"; do
  t=${so%.so}
  ROUNDS=1 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/ad_pmc1_$t -- python3 scripts/dev_attn_d256_ab.py $so > /dev/null 2> gpurun_out/ad_pmc1_$t.err
  ROUNDS=1 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/ad_pmc2_$t -- python3 scripts/dev_attn_d256_ab.py $so > /dev/null 2> gpurun_out/ad_pmc2_$t.err
done
python3 - "$@" <<'PY'
import csv, glob, collections, json, sys
out = {}
for so in sys.argv[1:]:
    t = so[:-3]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    durs = collections.defaultdict(list)
    for p in (1, 2):
        for f in glob.glob(f"gpurun_out/ad_pmc{p}_{t}/**/*counter_collection.csv", recursive=True):
            per = collections.defaultdict(float)
            for row in csv.DictReader(open(f)):
                k = row["Kernel_Name"]
                if "attn_" not in k or "reduce" in k: continue
                name = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
                per[(name, row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
                if p == 1: durs[(name, row["Dispatch_Id"])] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
            for (name, d, c), v in per.items(): acc[name][c].append(v)
    res = {}
    for name in sorted(acc):
        c = {k: sum(v) / len(v) for k, v in acc[name].items()}
        pd = [v for (n, d), v in durs.items() if n == name]
        c["pmc_pass_avg_us"] = sum(pd) / len(pd)
        # SQ_VALU_MFMA_BUSY_CYCLES: summed over the 1024 SIMDs; SQ_BUSY_CYCLES: summed over the 32 shader engines (scripts/prof_train_pmc.sh)
        c["mfma_busy"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / (c["SQ_BUSY_CYCLES"] / 32)
        c["valu_per_mfma"] = c.get("SQ_INSTS_VALU", 0) / c["SQ_INSTS_MFMA"]
        c["lds_per_mfma"] = c.get("SQ_INSTS_LDS", 0) / c["SQ_INSTS_MFMA"]
        c["clock_ghz_pmc_pass"] = c["SQ_BUSY_CYCLES"] / 32 / (c["pmc_pass_avg_us"] * 1e3)
        c["lds_bank_conflict_frac"] = c.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, c.get("SQ_LDS_IDX_ACTIVE", 0))
        res[name] = c
        print(so, name, {k: round(c[k], 3) for k in ("pmc_pass_avg_us", "mfma_busy", "valu_per_mfma", "lds_per_mfma", "clock_ghz_pmc_pass", "lds_bank_conflict_frac")})
    out[so] = res
json.dump(out, open("gpurun_out/attn_d256_pmc.json", "w"), indent=1)
PY
rm -rf gpurun_out/ad_pmc*
