#!/bin/bash
# timing-only ablations of the patch-staged AlignConv kernel (rebuilds dcn_ops.o per variant)
cd $GRAFT_REPO_ROOT
for a in 0 1 2 3 4 6 7; do
  rm -f s2anet_amd/csrc/dcn_ops.o
  make -C s2anet_amd/csrc -s EXTRA=-DS2A_ABL=$a 2>&1 | grep -E "error" | head -3
  echo "ABL=$a $(timeout -k 10 200 python scripts/bench_ops.py --which align8 2>&1 | tail -1 | cut -c60-130)"
done
rm -f s2anet_amd/csrc/dcn_ops.o; make -C s2anet_amd/csrc -s
