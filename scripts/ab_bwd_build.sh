#!/bin/bash
# same-box A/B of compile-time forms of the fused backward kernels: bash scripts/ab_bwd_build.sh <bwd16|bwd32> "<flags A>" "<flags B>" ...
cd $GRAFT_REPO_ROOT
export S2A_ALLOW_MEASURE_BUILD=1
W=$1; shift
trap 'rm -f s2anet_amd/csrc/dcn_bwd_ops.o; make -C s2anet_amd/csrc -s' EXIT
for rep in 1 2; do
  for fl in "$@"; do
    rm -f s2anet_amd/csrc/dcn_bwd_ops.o; make -C s2anet_amd/csrc -s EXTRA="$fl" 2>&1 | grep error
    echo "[$fl] rep $rep $(timeout -k 10 200 python scripts/bench_ops.py --which $W 2>/dev/null | grep '"op"' | cut -c60-190)"
  done
done
