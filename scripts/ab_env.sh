#!/bin/bash
# same-box A/B of an environment switch on the bench's roofline launches: bash scripts/ab_env.sh VAR A B [rounds]
R=$GRAFT_REPO_ROOT
V=$1; A=$2; B=$3; N=${4:-3}
for i in $(seq $N); do
  for X in $A $B; do
    env $V=$X timeout -k 10 200 python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --streams 1 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$V=$X', d['ms_per_step'], 'align_us', d['roofline']['avg_launch_us'], 'tower_us', d['roofline_conv_tower']['avg_launch_us'])"
  done
done
