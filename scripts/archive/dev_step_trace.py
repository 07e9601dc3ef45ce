"""Dev-only: one full fine-tune step of the bench as rocprofv3 saw it (a `--kernel-trace --output-format csv` directory of `bench.py --no-extras --no-c1 --no-c5 --no-cpu-baseline
--no-lora-leg --steps 2 --warmup 1 --train-steps 3`): launches, time in kernels, wall time, the kernels under 30 us by total, the gaps between kernels."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:70] for r in rows]
# last adam_multi marks the end of the last step; go back to the previous adam
idx = [i for i, n in enumerate(names) if n.startswith('adam_multi')]
a, b = idx[-2], idx[-1]
small = {}
tot = 0
for i in range(a + 1, b + 1):
    d = (int(rows[i]['End_Timestamp']) - int(rows[i]['Start_Timestamp'])) / 1e3
    tot += d
    if d < 30:
        k = names[i].split('<')[0].split('(')[0]
        small.setdefault(k, [0, 0.0]); small[k][0] += 1; small[k][1] += d
print("step: launches", b - a, "kernel us", round(tot), "wall us", (int(rows[b]['End_Timestamp']) - int(rows[a]['End_Timestamp'])) / 1e3)
for k, (n, t) in sorted(small.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{t:8.1f} us  {n:4d} x  {k}")
print("small kernels total us", round(sum(v[1] for v in small.values())), "count", sum(v[0] for v in small.values()))
# gaps
gaps = 0
for i in range(a + 1, b + 1):
    g = int(rows[i]['Start_Timestamp']) - int(rows[i - 1]['End_Timestamp'])
    if g > 0: gaps += g
print("gaps us", gaps / 1e3)
