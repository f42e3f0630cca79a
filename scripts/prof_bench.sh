#!/bin/bash
# kernel-trace + stats of the default bench command; steady-state per-kernel table via trace_steps.py
# usage (on the GPU box): bash scripts/prof_bench.sh <tag> [bench args]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
# a profiler-preloaded process must never spawn ranks (bench.py refuses too): single-rank profiling only
for a in "$@"; do case "$prev$a" in --gpus[2-9]*|--gpus=[2-9]*|--gpus1[0-9]*) echo "prof_bench.sh: --gpus > 1 is refused under rocprofv3 (profile rank by rank)"; exit 2;; esac; prev=$a; done
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $OUT/bench.log 2>&1 || { echo "rocprof run failed"; tail -5 $OUT/bench.log; exit 1; }
T=$(ls $OUT/*kernel_trace.csv $OUT/*/*kernel_trace.csv 2>/dev/null | head -1)
S=$(ls $OUT/*kernel_stats.csv $OUT/*/*kernel_stats.csv 2>/dev/null | head -1)
# the window of five steps ends inside the timed region: behind it come the single-stream loop of a multi-stream run (3 + 10 steps)
# and the operand-capture step
SKIP=14; case " $* " in *" --streams 1 "*) SKIP=1;; esac
python $R/scripts/trace_steps.py $T 5 60 $SKIP > $OUT/steady.txt
cp $S $OUT/kernel_stats.csv
grep '^{' $OUT/bench.log > $OUT/line.json
rm -f $T   # the raw trace is large; the summaries are what gets kept
head -45 $OUT/steady.txt
