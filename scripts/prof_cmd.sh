#!/bin/bash
# kernel-trace stats of an arbitrary python script: bash scripts/prof_cmd.sh <tag> <script.py> [args]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- python $R/"$@" > $OUT/run.log 2>&1 || { echo "rocprof run failed"; tail -5 $OUT/run.log; exit 1; }
S=$(ls $OUT/*kernel_stats.csv $OUT/*/*kernel_stats.csv 2>/dev/null | head -1)
cp $S $OUT/kernel_stats.csv
rm -f $OUT/*kernel_trace.csv $OUT/*/*kernel_trace.csv
python - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/kernel_stats.csv")))
for r in rows[:14]:
    print(f"{float(r['AverageNs'])/1e3:9.1f} us x{r['Calls']:>5}  {r['Name'][:110]}")
PY
