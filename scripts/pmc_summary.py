"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel (substring match) and counter, mean per launch."""
import csv, glob, json, sys, collections
pat = sys.argv[1]
out = collections.defaultdict(list)
meta = {}
for d in sys.argv[2:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(float)
        for row in csv.DictReader(open(f)):
            if pat in row["Kernel_Name"]:
                per[(row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
                meta = {"vgpr": row["VGPR_Count"], "sgpr": row["SGPR_Count"], "lds": row["LDS_Block_Size"], "wg": row["Workgroup_Size"], "grid": row["Grid_Size"]}
        for (disp, name), v in per.items():
            out[name].append(v)
res = {k: {"per_launch_mean": sum(v) / len(v), "n": len(v), "min": min(v), "max": max(v)} for k, v in sorted(out.items())}
print(json.dumps({"kernel": pat, "launch": meta, "counters": res}, indent=1))
