#!/bin/bash
# same-box A/B of an MFMA-shape switch, alternating builds:  bash scripts/m16_ab.sh <switch> <value a> <value b>
#   S2A_CONV_M16 1 0 (convolutions: 16x16x32 everywhere vs 32x32x16)   S2A_CONV_M16 1 3 (narrow 3x3 layers as well vs full-width only)
#   S2A_DCN_M16 1 0 (AlignConv matrix waves)
cd $GRAFT_REPO_ROOT
SW=${1:-S2A_CONV_M16}; A=${2:-1}; B=${3:-0}
export S2A_ALLOW_MEASURE_BUILD=1
trap 'rm -f s2anet_amd/csrc/dcn_ops.o; make -C s2anet_amd/csrc -s' EXIT
for rep in 1 2; do
  for m in $A $B; do
    rm -f s2anet_amd/csrc/dcn_ops.o; make -C s2anet_amd/csrc -s EXTRA=-D$SW=$m 2>&1 | grep error
    echo "$SW=$m pyr: $(timeout -k 10 200 python scripts/bench_pyr.py 2>/dev/null | tr '\n' ' ')"
    echo "$SW=$m tail: $(timeout -k 10 200 python scripts/bench_tail.py 2>/dev/null | tail -1 | cut -c1-100)"
    echo "$SW=$m bench: $(timeout -k 10 300 python bench.py --no-cpu-baseline --no-ops --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('chips/s', d['value'], 'alignconv us', d['roofline']['avg_launch_us'], d['roofline']['frac'], 'tower us', d['roofline_conv_tower']['avg_launch_us'], d['roofline_conv_tower']['frac'])")  1 stream: $(timeout -k 10 300 python bench.py --no-cpu-baseline --no-ops --steps 20 --streams 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])")"
  done
done
