"""summarise the steady-state tail of a rocprofv3 kernel trace: per-kernel time over the last `ms` milliseconds"""
import csv, sys, collections
path, ms = sys.argv[1], float(sys.argv[2])
rows = list(csv.DictReader(open(path)))
ends = [int(r['End_Timestamp']) for r in rows]
t1 = max(ends); t0 = t1 - int(ms * 1e6)
agg = collections.defaultdict(lambda: [0, 0.0])
busy = 0.0
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if s < t0: continue
    k = r['Kernel_Name']
    agg[k][0] += 1; agg[k][1] += (e - s) / 1e3
    busy += (e - s) / 1e3
print(f"window {ms} ms, kernel busy {busy/1e3:.2f} ms ({busy/1e3/ms*100:.1f}%), {sum(v[0] for v in agg.values())} launches")
for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 25]:
    print(f"{us/1e3:8.3f} ms {n:5d}x {us/n:9.1f} us  {k[:110]}")
