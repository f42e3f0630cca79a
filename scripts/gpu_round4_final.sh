#!/bin/bash
# end-of-round evidence (round 4): full GPU suite, smoke, bench counters -> traffic record, steady-state tables (3 streams and 1),
# the bench line, ops report, kernel list of a captured detect() replay, phase stamps of both AlignConv forms, the clock probe,
# the half-coordinate mode's end-to-end effect, a two-rank rehearsal on one card (gloo).  NMS / IoU kernels are unchanged since
# round 3 (their counter reports stay profiles/r03_*); scripts/gpu_round3_final.sh holds the recipe for those.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4final; mkdir -p $O; cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?; echo "gpu tests rc=$rc"; tail -3 $O/gpu_tests.log | cut -c1-200
if [ $rc -ne 0 ]; then exit 1; fi
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log | cut -c1-200
echo "== bench pmc"; bash scripts/pmc_bench.sh > $O/pmc_bench.log 2>&1; echo "pmc bench rc=$?"; cat gpurun_out/pmc_bench/traffic.json | head -c 900; echo
python -c "import json; json.dump(json.load(open('gpurun_out/pmc_bench/traffic.json')), open('profiles/r04_traffic.json', 'w'), indent=1)"   # the bench below reads this copy (publish_profiles.py writes the same file at home)
echo "== steady state"; bash scripts/prof_bench.sh r4final --no-ops > $O/prof_bench3.log 2>&1; head -3 $O/prof_bench3.log | cut -c1-160
bash scripts/prof_bench.sh r4final_s1 --streams 1 --no-ops > $O/prof_bench1.log 2>&1; head -3 $O/prof_bench1.log | cut -c1-160
echo "== graph replay kernels"; bash scripts/graph_trace.sh > $O/graph_replay_kernels.txt 2>&1; echo "graph trace rc=$?"; tail -1 $O/graph_replay_kernels.txt
echo "== ops report"; timeout -k 10 600 python scripts/bench_ops.py --which all > $O/ops_report.jsonl 2> $O/ops_report.err; echo "ops rc=$?"; wc -l $O/ops_report.jsonl
echo "== bench"; timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; head -c 400 $O/bench.json; echo
timeout -k 10 300 python bench.py --no-cpu-baseline --no-ops --streams 1 > $O/bench_s1.json 2>/dev/null; timeout -k 10 300 python bench.py --no-cpu-baseline --no-ops --streams 1 --graph > $O/bench_s1_graph.json 2>/dev/null
python -c "
import json
for f in ('bench_s1','bench_s1_graph'):
    d=json.load(open('$O/'+f+'.json')); print(f, d['value'], d['ms_per_step'])"
echo "== clock probe + AlignConv forms"; timeout -k 10 300 python scripts/pyr_power_probe.py > $O/pyr_power_probe.jsonl 2>/dev/null; cat $O/pyr_power_probe.jsonl
echo "== half-coordinate mode, end to end"; timeout -k 10 300 python scripts/half_mode_effect.py > $O/half_mode_effect.log 2>&1; tail -1 $O/half_mode_effect.log | cut -c1-300
echo "== two ranks on one card (gloo rehearsal)"; S2A_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --no-ops --streams 1 2>/dev/null | grep '^{' > $O/bench_2rank_gloo.json; head -c 300 $O/bench_2rank_gloo.json; echo
echo "== nms timeline"; bash scripts/nms_timeline.sh > $O/nms_timeline.txt 2>&1; grep -v rocprim $O/nms_timeline.txt | head -12 | cut -c1-110
echo "== phase stamps (diagnostic builds, restored afterwards)"; bash scripts/stamp_conv_run.sh > $O/conv_stamps.txt 2>&1; bash scripts/stamp_run.sh > $O/alignconv_stamps.txt 2>&1; bash scripts/stamp_sym_run.sh > $O/alignconv_sym_stamps.txt 2>&1; grep -v amdgpu.ids $O/alignconv_sym_stamps.txt | cut -c1-260
