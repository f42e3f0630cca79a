#!/bin/bash
# timing-only ablations of the pyramid-packed 3x3 tower launch (k_conv_f16<9,4,2>): each argument = EXTRA flags of one build
cd $GRAFT_REPO_ROOT
for a in "$@"; do
  rm -f s2anet_amd/csrc/dcn_ops.o
  make -C s2anet_amd/csrc -s EXTRA="$a" 2>&1 | grep -E "error" | head -3
  echo "[$a] $(timeout -k 10 200 python scripts/bench_pyr.py 2>&1 | grep '"op"' | cut -c1-120 | tr '\n' ' ')"
done
