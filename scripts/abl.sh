#!/bin/bash
# timing-only ablations of the AlignConv kernels on the pyramid launch (rebuilds dcn_ops.o per variant ON THE GPU BOX's
# copy).  S2A_ABL bits: 1 = no epilogue (plain kernel), 2 = loaders skip the blend, 4 = matrix waves skip the MFMAs,
# 8 = loaders skip their corner reads (ring-3 kernel), 16 = matrix waves skip their fragment reads (ring-3 kernel),
# 32 = filter fragments loaded once (plain kernel, S2A_MPIPE form)
cd $GRAFT_REPO_ROOT
for a in "$@"; do
  rm -f s2anet_amd/csrc/dcn_ops.o
  make -C s2anet_amd/csrc -s EXTRA=-DS2A_ABL=$a 2>&1 | grep -E "error" | head -3
  echo "ABL=$a $(timeout -k 10 200 python scripts/bench_pyr.py 2>&1 | grep alignconv_pyramid)"
done
