#!/bin/bash
# timing-only ablations of k_conv_wino_f16 on the pyramid launch (GPU box's copy): every argument is a value of S2A_WABL
#   bash scripts/abl_wino.sh 0 1 2 3 4 8
# bits: 1 = no patch DMA in the loop, 2 = no filter DMA in the loop, 4 = no MFMAs, 8 = no barriers in the loop
cd $GRAFT_REPO_ROOT
export S2A_ALLOW_MEASURE_BUILD=1   # the objects built below carry a measurement switch (s2anet_amd/_lib.py refuses them otherwise)
restore() { rm -f s2anet_amd/csrc/wino_ops.o; make -C s2anet_amd/csrc -s 2>&1 | grep -E "error" | head -3; }
trap restore EXIT
for a in "$@"; do
  rm -f s2anet_amd/csrc/wino_ops.o
  make -C s2anet_amd/csrc -s EXTRA="-DS2A_WABL=$a" 2>&1 | grep -E "error" | head -3
  echo "[S2A_WABL=$a] $(timeout -k 10 200 python scripts/bench_wino.py 1 2>&1 | grep '"wino"' | python -c "import sys,json; print(' '.join('%s %.1f' % (json.loads(l)['data'], json.loads(l)['us']) for l in sys.stdin))")"
done
