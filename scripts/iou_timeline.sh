#!/bin/bash
# kernel timeline of ONE box_iou_rotated call at 10 k x 10 k (which kernels overlap the forked zero-fill, and for how long)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/iou_tl; mkdir -p $O; cd /tmp
cat > /tmp/iou_once.py <<PY
import sys, torch
sys.path.insert(0, "$R")
import numpy as np
from scripts.bench_ops import rboxes
import s2anet_amd as S
rng = np.random.default_rng(1234)
a = torch.from_numpy(rboxes(rng, 10000)).cuda(); b = torch.from_numpy(rboxes(rng, 10000)).cuda()
for _ in range(12): S.box_iou_rotated(a, b)
torch.cuda.synchronize()
PY
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O -o run -- python /tmp/iou_once.py > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
T=$(ls $O/*kernel_trace.csv $O/*/*kernel_trace.csv 2>/dev/null | head -1)
python $R/scripts/timeline.py $T k_prep_boxes2 8 1
rm -f $T
