#!/bin/bash
# same-box A/B of compile-time forms of k_conv_f16: pyramid tower launch + the bench step (one stream), alternating builds
cd $GRAFT_REPO_ROOT
export S2A_ALLOW_MEASURE_BUILD=1
trap 'rm -f s2anet_amd/csrc/dcn_ops.o; make -C s2anet_amd/csrc -s' EXIT
for rep in 1 2; do
  for fl in "$@"; do
    rm -f s2anet_amd/csrc/dcn_ops.o; make -C s2anet_amd/csrc -s EXTRA="$fl" 2>&1 | grep error
    echo "== [$fl] rep $rep"
    timeout -k 10 200 python scripts/bench_pyr.py 2>/dev/null | grep conv3x3 | cut -c1-100
    timeout -k 10 300 python bench.py --no-cpu-baseline --no-ops --streams 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('one stream', d['value'], d['ms_per_step'], 'tower', d['roofline_conv_tower']['avg_launch_us'])"
  done
done
