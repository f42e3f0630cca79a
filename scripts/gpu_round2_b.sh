#!/bin/bash
# GPU call: NMS / IoU rework -- parity tests first, then timings
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2b; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -k "iou or nms or assign or multiclass or merge or voc" > $O/t1.log 2>&1; rc=$?; echo "rotated tests rc=$rc"; tail -12 $O/t1.log | cut -c1-200
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 300 python scripts/graph_diag.py c > $O/diag_c.log 2>&1; rc=$?; echo "graph diag c rc=$rc"; grep -v "^\s" $O/diag_c.log | grep -v amdgpu | tail -4 | cut -c1-160
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 300 python scripts/bench_ops.py --which nms > $O/ops_nms.jsonl 2>&1 && cat $O/ops_nms.jsonl
timeout -k 10 300 python scripts/bench_ops.py --which iou > $O/ops_iou.jsonl 2>&1 && cat $O/ops_iou.jsonl
timeout -k 10 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_ops.py -x -q -k "not (iou or nms or assign or multiclass or merge or voc)" > $O/t2.log 2>&1; rc=$?; echo "other tests rc=$rc"; tail -6 $O/t2.log | cut -c1-200
if [ $rc -ne 0 ]; then exit 1; fi
bash scripts/prof_cmd.sh nms200k scripts/bench_ops.py --which nms200k > $O/prof_nms.log 2>&1; tail -14 $O/prof_nms.log | cut -c1-150
bash scripts/prof_cmd.sh iou scripts/bench_ops.py --which iou > $O/prof_iou.log 2>&1; tail -9 $O/prof_iou.log | cut -c1-150
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['config']['detections_per_chip'])"
timeout -k 10 300 python bench.py --no-cpu-baseline --streams 1 --graph > $O/bench_graph.json 2> $O/bench_graph.err; echo "bench graph rc=$?"; python -c "
import json; d=json.load(open('$O/bench_graph.json')); print(d['value'], d['ms_per_step'])"
timeout -k 10 300 python bench.py --no-cpu-baseline --streams 1 > $O/bench_s1.json 2> $O/bench_s1.err; echo "bench s1 rc=$?"; python -c "
import json; d=json.load(open('$O/bench_s1.json')); print(d['value'], d['ms_per_step'])"
