#!/bin/bash
# same-box A/B of two builds of the library on the bench's roofline launches:
#   bash scripts/ab_bench.sh <lib_a.so> <lib_b.so> [rounds]
R=$GRAFT_REPO_ROOT
A=$1; B=$2; N=${3:-2}
for i in $(seq $N); do
  for L in $A $B; do
    S2A_LIB_PATH=$R/$L timeout -k 10 200 python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --streams 1 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$L', d['ms_per_step'], 'align_us', d['roofline']['avg_launch_us'], 'tower_us', d['roofline_conv_tower']['avg_launch_us'])"
  done
done
