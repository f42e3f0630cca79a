#!/bin/bash
# kernel trace of a captured detect(): the kernels of the LAST replay, and a check that none of them is a fill / memset
# kernel or a rocPRIM partition (those are what broke graph replay on ROCm 7.2)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/graph_tl; mkdir -p $O; cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O -o run -- python $R/scripts/graph_replay_once.py > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
T=$(ls $O/*kernel_trace.csv $O/*/*kernel_trace.csv 2>/dev/null | head -1)
python - "$T" <<'PY'
import csv, sys, re, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
stems = [i for i, r in enumerate(rows) if "k_stem" in r["Kernel_Name"]]
last = rows[stems[-1]:]                     # the last replay starts at its fused stem launch
names = collections.Counter()
for r in last:
    m = re.search(r"(k_\w+|rocprim\w*|\w*[Ff]ill\w*|\w*[Mm]emset\w*|igemm\w+|Cijk\w+)", r["Kernel_Name"])
    names[(m.group(1) if m else r["Kernel_Name"])[:60]] += 1
print("kernels of the last of three replays of one captured detect() (2 x 512 x 512): %d launches" % len(last))
for k, c in sorted(names.items(), key=lambda kv: -kv[1]):
    print("%4d x %s" % (c, k))
bad = [r["Kernel_Name"][:80] for r in last if re.search(r"fillBuffer|[Mm]emset|partition", r["Kernel_Name"])]
print("fill / memset / rocPRIM-partition kernels in the replay:", len(bad), bad[:3])
sys.exit(1 if bad else 0)
PY
rc=$?
rm -f $T
exit $rc
