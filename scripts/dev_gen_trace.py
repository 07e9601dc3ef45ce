"""Dev-only: one decode step of the C5 generate as rocprofv3 saw it -- reads a `rocprofv3 --kernel-trace --output-format csv` directory of `LORA=1 dev_gen_only.py`
and prints the kernels of one layer of a late token in launch order: duration and the gap to the kernel before (eager launches: the gaps are the host's; in the replayed graph
they are what the trace of `dev_gen_only.py 1 g` shows)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '') for r in rows]
# the last argmax but one marks the end of a token; take the launches between the two argmaxes before it
idx = [i for i, n in enumerate(names) if n.startswith('argmax_rows')]
a, b = idx[-3], idx[-2]
prev_end = int(rows[a]['End_Timestamp'])
tot = 0
for i in range(a + 1, b + 1):
    s, e = int(rows[i]['Start_Timestamp']), int(rows[i]['End_Timestamp'])
    if i - a <= 40 or i >= b - 6:
        print(f"{names[i][:60]:60s} {(e - s) / 1e3:7.1f} us   gap {(s - prev_end) / 1e3:6.1f} us   grid {rows[i].get('Grid_Size_X', '?')}")
    prev_end = e
    tot += e - s
print(f"token: {b - a} launches, {(int(rows[b]['End_Timestamp']) - int(rows[a]['End_Timestamp'])) / 1e3:.0f} us wall, {tot / 1e3:.0f} us in kernels")
