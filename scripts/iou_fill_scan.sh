#!/bin/bash
# box_iou_rotated 10 k x 10 k: how the forked zero-fill (workgroups, pacing) and the pair finder interact (timeline per setting)
cd $GRAFT_REPO_ROOT
export S2A_ALLOW_MEASURE_BUILD=1   # the objects built below carry measurement switches (s2anet_amd/_lib.py refuses them otherwise)
restore() { rm -f s2anet_amd/csrc/rotated_ops.o; make -C s2anet_amd/csrc -s 2>&1 | grep -E "error" | head -3; }
trap restore EXIT
rm -f s2anet_amd/csrc/rotated_ops.o; make -C s2anet_amd/csrc -s EXTRA="-DS2A_MEASURE" 2>&1 | grep -E "error" | head -3   # the fill switches exist in measurement builds only
# each argument: "<column-major finder 0|1> <fill workgroups, 0 = no fill> <s_sleep pace>"
for cfg in "$@"; do
  set -- $cfg
  echo "== cols=$1 fill_wgs=$2 pace=$3"
  S2A_IOU_CULL_COLS=$1 S2A_IOU_FILL_WGS=$2 S2A_IOU_FILL_PACE=$3 bash scripts/iou_timeline.sh 2>&1 | grep "k_iou_cull\|k_fill\|k_iou_heavy\|k_iou_scatter" | cut -c1-90
done
