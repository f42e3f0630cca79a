#!/bin/bash
# box_iou_rotated 10 k x 10 k: how the forked zero-fill (workgroups, pacing) and the pair finder interact (timeline per setting)
cd $GRAFT_REPO_ROOT
# each argument: "<column-major finder 0|1> <fill workgroups, 0 = no fill> <s_sleep pace>"
for cfg in "$@"; do
  set -- $cfg
  echo "== cols=$1 fill_wgs=$2 pace=$3"
  S2A_IOU_CULL_COLS=$1 S2A_IOU_FILL_WGS=$2 S2A_IOU_FILL_PACE=$3 bash scripts/iou_timeline.sh 2>&1 | grep "k_iou_cull\|k_fill\|k_iou_heavy\|k_iou_scatter" | cut -c1-90
done
