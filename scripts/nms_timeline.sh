#!/bin/bash
# kernel timeline of ONE ml_nms_rotated call at 200 k rows x 15 labels (launch gaps between the ~35 dependent kernels)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/nms_tl; mkdir -p $O; cd /tmp
cat > /tmp/nms_once.py <<PY
import sys, torch, numpy as np
sys.path.insert(0, "$R")
from scripts.bench_ops import rboxes
from s2anet_amd.rotated import ml_nms_rotated
rng = np.random.default_rng(1234)
n = 200000
d = torch.from_numpy(rboxes(rng, n)).cuda(); s = torch.from_numpy(rng.permutation(n).astype(np.float32) / n).cuda()
l = torch.from_numpy(rng.integers(0, 15, n).astype(np.float32)).cuda()
for _ in range(8): ml_nms_rotated(d, s, l, 0.5)
torch.cuda.synchronize()
PY
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O -o run -- python /tmp/nms_once.py > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
T=$(ls $O/*kernel_trace.csv $O/*/*kernel_trace.csv 2>/dev/null | head -1)
python $R/scripts/timeline.py $T k_nms_prep 5 1
rm -f $T
