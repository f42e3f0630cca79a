#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6final; mkdir -p $O; cd $R
timeout -k 10 400 python -m pytest tests/test_gpu_ops.py -x -q -k "dcn_backward" > $O/bwd_tests.log 2>&1; echo "bwd tests rc=$?"; tail -2 $O/bwd_tests.log
echo "== ops report"; timeout -k 10 600 python scripts/bench_ops.py --which all > $O/ops_report.jsonl 2> $O/ops_report.err; echo "ops rc=$?"; wc -l $O/ops_report.jsonl
echo "== bench"; timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; head -c 300 $O/bench.json; echo
echo "== deform-conv backward: kernel trace + stamps of the f32 weight kernel"
bash scripts/prof_cmd.sh bwd scripts/bench_ops.py --which bwd > $O/dcn_backward_kernel_stats.txt 2>&1; grep '"op"' gpurun_out/prof_bwd/run.log | cut -c1-260 >> $O/dcn_backward_kernel_stats.txt; tail -3 $O/dcn_backward_kernel_stats.txt | cut -c1-200
bash scripts/bwd32_stamps.sh 2>&1 | grep -v amdgpu.ids > $O/dcn_backward_f32_weight_stamps.txt
find $R/gpurun_out -type f \( -name "*_counter_collection.csv" -o -name "*kernel_trace.csv" -o -name "*.db" -o -name "*.rocpd" -o -name "*_agent_info.csv" \) -delete
find $R/gpurun_out -type f -size +4M -delete
