"""Dev-only: copy one round of measurements from gpurun_out/ into profiles/rNN/ (bench line, rocprofv3 kernel stats, per-launch
HBM and SQ counters of the encode kernel).  Usage: python scripts/refresh_profiles.py r01 r1r bench_line_r.txt"""
import csv, glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd, prefix, bench_txt = sys.argv[1:4]
out = os.path.join(ROOT, "profiles", rnd)
G = os.path.join(ROOT, "gpurun_out")


def summ(dirs):
    return json.loads(subprocess.check_output([sys.executable, os.path.join(ROOT, "scripts", "pmc_summary.py"), "encode_flow_kernel"] + dirs))


stats = glob.glob(os.path.join(G, prefix + "_stats", "**", "*_kernel_stats.csv"), recursive=True)[0]
shutil.copy(stats, os.path.join(out, "bench_kernel_stats.csv"))
enc = [r for r in csv.DictReader(open(stats)) if "encode_flow" in r["Name"]][0]
kernel_us = float(enc["AverageNs"]) / 1e3
h = summ([os.path.join(G, prefix + "_FETCH_SIZE"), os.path.join(G, prefix + "_WRITE_SIZE")])
hb = {"kernel": "encode_flow_kernel<INPUT_F64>, 4096 records of 12x5000 per launch"}
hb.update(h["counters"])
json.dump(hb, open(os.path.join(out, "hbm_pmc.json"), "w"), indent=1)
traffic = (hb["FETCH_SIZE"]["per_launch_mean"] * 2 + hb["WRITE_SIZE"]["per_launch_mean"]) * 1024
sq = summ([os.path.join(G, prefix + "_sq1"), os.path.join(G, prefix + "_sq2")])
c = sq["counters"]
sq["kernel"] = "encode_flow_kernel<INPUT_F64>, 4096 records of 12x5000 per launch (sums over the chip)"
sq["derived"] = {
    "valu_insts_per_record": c["SQ_INSTS_VALU"]["per_launch_mean"] / 4096,
    "salu_insts_per_record": c["SQ_INSTS_SALU"]["per_launch_mean"] / 4096,
    "lds_insts_per_record": c["SQ_INSTS_LDS"]["per_launch_mean"] / 4096,
    "valu_busy_frac": c["SQ_ACTIVE_INST_VALU"]["per_launch_mean"] * 4 / (kernel_us * 1e-6 * 2.4e9 * 1024),
    "valu_busy_frac_how": "SQ_ACTIVE_INST_VALU (quad-cycles) x 4 / (%.1f us x 2.4 GHz x 1024 SIMDs)" % kernel_us,
}
json.dump(sq, open(os.path.join(out, "sq_pmc.json"), "w"), indent=1)
line = [x for x in open(os.path.join(G, bench_txt)) if x.startswith("{")][-1]
b = json.loads(line)
b["roofline"]["traffic"] = traffic
json.dump(b, open(os.path.join(out, "bench_line.json"), "w"), indent=1)
print("encode: %.4f ms/step, %.2f G tokens/s, frac %.3f, kernel avg %.1f us (%s calls), traffic %.3f GB" % (
    b["ms_per_step"], b["value"] / 1e9, b["roofline"]["frac"], kernel_us, enc["Calls"], traffic / 1e9))
print("train: %.1f ms, %.1f samples/s; lora %.1f ms" % (b["train"]["ms_per_step"], b["train"]["value"], b["train"]["lora_r16"]["ms_per_step"]))
print("cpu:", b["cpu_baseline"]["value"], b["cpu_baseline"]["sample"])
print({k: round(v, 3) if isinstance(v, float) else v for k, v in sq["derived"].items()})
print("FETCH KiB %.0f WRITE KiB %.0f" % (hb["FETCH_SIZE"]["per_launch_mean"], hb["WRITE_SIZE"]["per_launch_mean"]))
