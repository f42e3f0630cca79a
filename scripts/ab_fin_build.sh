#!/bin/bash
# same-box A/B of compile-time forms of the NMS clean-up kernel inside the bench step: bash scripts/ab_fin_build.sh "<EXTRA A>" "<EXTRA B>" ...
cd $GRAFT_REPO_ROOT
export S2A_ALLOW_MEASURE_BUILD=1
trap 'rm -f s2anet_amd/csrc/rotated_ops.o; make -C s2anet_amd/csrc -s' EXIT
i=0
for fl in "$@"; do
  i=$((i+1))
  rm -f s2anet_amd/csrc/rotated_ops.o; make -C s2anet_amd/csrc -s EXTRA="$fl" 2>&1 | grep error
  echo "== [$fl]"
  bash scripts/prof_bench.sh finab$i --streams 1 --no-ops 2>&1 | grep -i "finish_seg" | cut -c1-100
done
