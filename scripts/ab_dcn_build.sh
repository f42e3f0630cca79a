#!/bin/bash
# same-box A/B of compile-time forms of k_dcn_patch: bash scripts/ab_dcn_build.sh "<EXTRA flags A>" "<EXTRA flags B>" ...
# prints the pyramid launch (zeros / dense) and configs[1] as stated for every build, twice, alternating; restores the default build
cd $GRAFT_REPO_ROOT
export S2A_ALLOW_MEASURE_BUILD=1
trap 'rm -f s2anet_amd/csrc/dcn_ops.o; make -C s2anet_amd/csrc -s' EXIT
for rep in 1 2; do
  for fl in "$@"; do
    rm -f s2anet_amd/csrc/dcn_ops.o; make -C s2anet_amd/csrc -s EXTRA="$fl" 2>&1 | grep error
    echo "== [$fl] rep $rep"
    timeout -k 10 200 python scripts/pyr_power_probe.py --patch-only 2>/dev/null | cut -c1-100
    timeout -k 10 200 python scripts/bench_ops.py --which alignconv1 2>/dev/null | cut -c1-160
  done
done
