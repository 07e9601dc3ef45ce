import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
per = collections.defaultdict(list)
for r in rows:
    per[r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0]].append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))
for k, v in per.items():
    if len(v) < 4000: continue
    v.sort(); v = v[-4000:]           # second repetition
    d = [e - s for s, e in v]
    print(k, 'total ms', sum(d) / 1e6, ' per-merge us at merges 0,10,100,500,1000,2000,3999:', [round(d[i] / 1e3, 1) for i in (0, 10, 100, 500, 1000, 2000, 3999)], ' mean of 3100..3900:', round(sum(d[3100:3900]) / 800e3, 2))
# gaps: time between consecutive kernels in the second repetition
allk = sorted((s, e) for v in per.values() for s, e in v)
allk = allk[len(allk) // 2:]
busy = sum(e - s for s, e in allk); span = allk[-1][1] - allk[0][0]
print('second half: span ms', span / 1e6, 'busy ms', busy / 1e6, 'gaps ms', (span - busy) / 1e6, 'launches', len(allk))
