#!/bin/bash
# PMC passes (counters only + kernel trace) over an arbitrary python script, one pass per quoted counter set:
#   bash scripts/pmc_cmd.sh <tag> "<script.py args>" "<kernel substr> <kernel substr> ..." "<set 1>" "<set 2>" ...
# -> gpurun_out/pmc_<tag>/summary_<kernel>.txt (scripts/pmc_summary.py tables)
export TMPDIR=/tmp
cd /tmp
R=$GRAFT_REPO_ROOT
TAG=$1; CMD=$2; KERNELS=$3; shift; shift; shift
OUT=$R/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
i=0
for set in "$@"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python $R/$CMD > $OUT/p$i.log 2>&1 || { echo "pass $i ($set) failed"; tail -3 $OUT/p$i.log; exit 1; }
  # keep one kernel trace (durations, grid, LDS / VGPR / SGPR per dispatch), drop the others
  if [ $i -gt 1 ]; then find $OUT/p$i -name "*kernel_trace.csv" -delete; fi
done
for k in $KERNELS; do
  python $R/scripts/pmc_summary.py $OUT $k > $OUT/summary_$k.txt
  echo "== $k"; cat $OUT/summary_$k.txt
done
