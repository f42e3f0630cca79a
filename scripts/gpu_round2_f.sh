#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2f; mkdir -p $O; cd $R
timeout -k 10 200 python scripts/nms_seg_diag.py 6000 15 500 2>&1 | grep mismatches
timeout -k 10 200 python scripts/nms_seg_diag.py 30000 3 700 2>&1 | grep mismatches
