"""per-kernel time inside N full steps of a bench.py trace; a step ends at k_nms_group_emit* (k_nms_group_compact before round 4).
python trace_steps.py trace.csv <steps> [<top rows> [<steps to skip at the end of the trace>]] -- the skip keeps the window inside
the timed region: behind it bench.py runs its single-stream loop, the operand-capture step and the roofline launches"""
import csv, sys, collections
path, nsteps = sys.argv[1], int(sys.argv[2])
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r['Start_Timestamp']))
marks = [int(r['End_Timestamp']) for r in rows if ('k_nms_group_emit' in r['Kernel_Name'] or 'k_nms_group_compact' in r['Kernel_Name'])]
skip = int(sys.argv[4]) if len(sys.argv) > 4 else 0
t0, t1 = marks[-nsteps - 1 - skip], marks[-1 - skip]
agg = collections.defaultdict(lambda: [0, 0.0]); busy = 0.0; n = 0
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if s < t0 or e > t1 + 2000000: continue
    if s > t1: continue
    agg[r['Kernel_Name']][0] += 1; agg[r['Kernel_Name']][1] += (e - s) / 1e3; busy += (e - s) / 1e3; n += 1
wall = (t1 - t0) / 1e6
print(f"{nsteps} steps: wall {wall/nsteps:.3f} ms/step, kernel busy {busy/1e3/nsteps:.3f} ms/step ({busy/1e3/wall*100:.1f}%), {n/nsteps:.0f} launches/step")
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
for k, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{us/1e3/nsteps:8.3f} ms/step {c/nsteps:6.1f}x {us/c:9.1f} us  {k[:120]}")
