#!/bin/bash
# A/B of the tower convolution's MFMA shape on one box: 16x16x32 (default build) vs 32x32x16 (-DS2A_CONV_M16=0), alternating
cd $GRAFT_REPO_ROOT
export S2A_ALLOW_MEASURE_BUILD=1
trap 'rm -f s2anet_amd/csrc/dcn_ops.o; make -C s2anet_amd/csrc -s' EXIT
for rep in 1 2; do
  for m in 1 0; do
    rm -f s2anet_amd/csrc/dcn_ops.o; make -C s2anet_amd/csrc -s EXTRA=-DS2A_CONV_M16=$m 2>&1 | grep error
    echo "M16=$m pyr: $(timeout -k 10 200 python scripts/bench_pyr.py 2>/dev/null | head -1)"
    echo "M16=$m bench: $(timeout -k 10 300 python bench.py --no-cpu-baseline --no-ops --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline_conv_tower']['avg_launch_us'], d['roofline_conv_tower']['frac'])")"
  done
done
