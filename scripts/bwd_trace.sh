#!/bin/bash
# kernel trace of the deform_conv backward benchmark (which kernels, how long)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/bwd_tl; mkdir -p $O; cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o run -- python $R/scripts/bench_ops.py --which bwd16 > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
S=$(ls $O/*kernel_stats.csv $O/*/*kernel_stats.csv 2>/dev/null | head -1)
head -12 $S | cut -c1-160
grep '"op"' $O/run.log
rm -f $O/*kernel_trace.csv $O/*/*kernel_trace.csv
