#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2h; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -k "iou or assign" > $O/t1.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $O/t1.log | cut -c1-200
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 300 python scripts/bench_ops.py --which iou 2>&1 | grep box_iou | cut -c1-200
S2A_IOU_NO_GRID=1 timeout -k 10 300 python scripts/bench_ops.py --which iou 2>&1 | grep box_iou | cut -c1-200
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/tl -o run -- python $R/scripts/bench_ops.py --which iou > $O/tl.log 2>&1
T=$(ls $O/tl/*kernel_trace.csv $O/tl/*/*kernel_trace.csv 2>/dev/null | head -1)
python $R/scripts/timeline.py $T k_fill_zero 6 2
rm -f $T
