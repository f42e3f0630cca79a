#!/bin/bash
# one PMC pass over the default bench command: bash scripts/pmc_one.sh <tag> <kernel substring> <counter> [<counter> ...]
export TMPDIR=/tmp
cd /tmp
R=$GRAFT_REPO_ROOT
TAG=$1; KSUB=$2; shift; shift
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/p1 -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --streams 1 > $OUT/p1.log 2>&1 || echo "pass failed"
find $OUT/p1 -name "*kernel_trace.csv" -delete
python $R/scripts/pmc_summary.py $OUT "$KSUB"
