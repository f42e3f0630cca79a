#!/bin/bash
# A/B of the matrix-wave B-fragment pipeline of k_dcn_patch (S2A_MPIPE) on the pyramid launch, with ablations.
# usage: abl_mpipe.sh "<extra flags>" ...   (each argument = one build + one timing)
cd $GRAFT_REPO_ROOT
for a in "$@"; do
  rm -f s2anet_amd/csrc/dcn_ops.o
  make -C s2anet_amd/csrc -s EXTRA="$a" 2>&1 | grep -E "error" | head -3
  echo "[$a] $(timeout -k 10 200 python scripts/bench_pyr.py 2>&1 | grep alignconv_pyramid | cut -c1-160)"
done
