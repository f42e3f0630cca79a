#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2h; mkdir -p $O; cd $R
echo "--- default"; timeout -k 10 300 python scripts/bench_ops.py --which align 2>&1 | grep alignconv | cut -c1-150
echo "--- NO_HALF"; S2A_DCN_NO_HALF=1 timeout -k 10 300 python scripts/bench_ops.py --which align 2>&1 | grep alignconv | cut -c1-150
bash scripts/prof_cmd.sh align scripts/bench_ops.py --which align > $O/prof_align.log 2>&1; grep "k_dcn\|k_nchw\|k_pack" $O/prof_align.log | cut -c1-150
