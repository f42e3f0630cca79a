#!/bin/bash
# compile-time A/B of rotated_ops on the GPU box's copy: every argument is the EXTRA flag string of one build
cd $GRAFT_REPO_ROOT
export S2A_ALLOW_MEASURE_BUILD=1   # the objects built below carry measurement switches (s2anet_amd/_lib.py refuses them otherwise)
restore() { rm -f s2anet_amd/csrc/rotated_ops.o; make -C s2anet_amd/csrc -s 2>&1 | grep -E "error" | head -3; }
trap restore EXIT
for flags in "$@"; do
  rm -f s2anet_amd/csrc/rotated_ops.o
  make -C s2anet_amd/csrc -s EXTRA="$flags" 2>&1 | grep -E "error" | head -3
  echo "[$flags] $(python scripts/nms_dense.py 2>&1 | grep '"op"') $(S2A_NMS_DEBUG=1 python scripts/nms_dense.py 2>&1 | grep '\[cull\]' | tail -1)"
done
