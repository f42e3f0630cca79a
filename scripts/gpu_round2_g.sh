#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2g; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -k "pyramid or alignconv or dcn or detector" > $O/t1.log 2>&1; rc=$?; echo "align tests rc=$rc"; tail -6 $O/t1.log | cut -c1-200
if [ $rc -ne 0 ]; then exit 1; fi
for m in 0 1 0 1; do echo "RING3=$m $(S2A_DCN_RING3=$m timeout -k 10 200 python scripts/bench_pyr.py 2>&1 | grep alignconv_pyramid)"; done
timeout -k 10 600 python -m pytest tests/test_gpu_e2e.py -x -q > $O/t2.log 2>&1; rc=$?; echo "e2e tests rc=$rc"; tail -3 $O/t2.log | cut -c1-200
for m in 0 1; do S2A_DCN_RING3=$m timeout -k 10 300 python bench.py --no-cpu-baseline --streams 1 > $O/bench_$m.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench_$m.json')); print('RING3=$m', d['value'], d['ms_per_step'], 'align us', d['roofline']['avg_launch_us'], 'frac', d['roofline']['frac'])"; done
