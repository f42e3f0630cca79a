#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2e; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -k "iou or nms or assign or multiclass or merge or voc" > $O/t1.log 2>&1; rc=$?; echo "rotated tests rc=$rc"; tail -8 $O/t1.log | cut -c1-200
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 600 python -m pytest tests/test_gpu_e2e.py -x -q > $O/t2.log 2>&1; rc=$?; echo "e2e tests rc=$rc"; tail -5 $O/t2.log | cut -c1-200
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 300 python scripts/bench_ops.py --which nms > $O/ops_nms.jsonl 2>&1 && grep ml_nms $O/ops_nms.jsonl | cut -c1-120
bash scripts/prof_bench.sh r2s1 --streams 1 2>&1 | head -32 | cut -c1-150
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
