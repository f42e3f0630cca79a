#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2c; mkdir -p $O; cd $R
for v in d4 c; do
timeout -k 10 60 python scripts/graph_diag.py $v > $O/diag_$v.log 2>&1; rc=$?
echo "== variant $v rc=$rc"; grep -v "amdgpu.ids" $O/diag_$v.log | grep -v "^\s" | head -9 | cut -c1-160
if [ $rc -ne 0 ]; then exit 1; fi
done
