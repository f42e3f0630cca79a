#!/bin/bash
# end-of-round evidence (round 3): full GPU suite, smoke, NMS / IoU counters and timelines, bench counters -> traffic record,
# steady-state tables, ops report, the bench line, the kernel list of a captured detect() replay, device-side NMS totals
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3final; mkdir -p $O; cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?; echo "gpu tests rc=$rc"; tail -3 $O/gpu_tests.log | cut -c1-200
if [ $rc -ne 0 ]; then exit 1; fi
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log | cut -c1-200
echo "== nms pmc"
bash scripts/pmc_cmd.sh nms200k "scripts/bench_ops.py --which nms200k" "k_nms_cull_lanes k_nms_heavy k_nms_tile_filter k_nms_round k_nms_sp_meta k_nms_pos_meta" \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" \
  "SQ_BUSY_CU_CYCLES SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU" > $O/nms_pmc.log 2>&1; echo "nms pmc rc=$?"
python scripts/nms_pmc_report.py gpurun_out/pmc_nms200k --json $O/nms_occupancy.json k_nms_cull_lanes "k_nms_heavy<false>" "k_nms_heavy<true>" k_nms_tile_filter k_nms_round k_nms_sp_meta k_nms_pos_meta > $O/nms_pmc_report.txt 2>&1; head -8 $O/nms_pmc_report.txt | cut -c1-170
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O/memcpy -o run -- python $R/scripts/bench_ops.py --which nms200k > $O/memcpy.log 2>&1); echo "memcpy rc=$?"
cat $O/memcpy/run_memory_copy_stats.csv 2>/dev/null | head -4
bash scripts/nms_timeline.sh > $O/nms_timeline.txt 2>&1; grep -v rocprim $O/nms_timeline.txt | head -8 | cut -c1-110
echo "== iou pmc + timeline"
bash scripts/pmc_cmd.sh iou10k "scripts/bench_ops.py --which iou10k" "k_iou_cull_lanes k_iou_heavy" \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" \
  "SQ_BUSY_CU_CYCLES SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU" > $O/iou_pmc.log 2>&1; echo "iou pmc rc=$?"
python scripts/nms_pmc_report.py gpurun_out/pmc_iou10k k_iou_cull_lanes k_iou_heavy k_fill_zero k_iou_scatter > $O/iou_pmc_report.txt 2>&1
bash scripts/iou_timeline.sh > $O/iou_timeline.txt 2>&1; cat $O/iou_timeline.txt | cut -c1-120
echo "== bench pmc"; bash scripts/pmc_bench.sh > $O/pmc_bench.log 2>&1; echo "pmc bench rc=$?"; cat gpurun_out/pmc_bench/traffic.json | head -c 900; echo
echo "== steady state"; bash scripts/prof_bench.sh r3final --no-ops > $O/prof_bench3.log 2>&1; head -3 $O/prof_bench3.log | cut -c1-160
bash scripts/prof_bench.sh r3final_s1 --streams 1 --no-ops > $O/prof_bench1.log 2>&1; head -3 $O/prof_bench1.log | cut -c1-160
echo "== graph replay kernels"; bash scripts/graph_trace.sh > $O/graph_replay_kernels.txt 2>&1; echo "graph trace rc=$?"; tail -1 $O/graph_replay_kernels.txt
echo "== backward trace"; bash scripts/bwd_trace.sh > $O/bwd_trace.txt 2>&1; tail -2 $O/bwd_trace.txt | cut -c1-200
echo "== ops report"; timeout -k 10 600 python scripts/bench_ops.py --which all > $O/ops_report.jsonl 2> $O/ops_report.err; echo "ops rc=$?"; wc -l $O/ops_report.jsonl
echo "== bench"; timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; head -c 400 $O/bench.json; echo
timeout -k 10 300 python bench.py --no-cpu-baseline --no-ops --streams 1 > $O/bench_s1.json 2>/dev/null; timeout -k 10 300 python bench.py --no-cpu-baseline --no-ops --streams 1 --graph > $O/bench_s1_graph.json 2>/dev/null
python -c "
import json
for f in ('bench_s1','bench_s1_graph'):
    d=json.load(open('$O/'+f+'.json')); print(f, d['value'], d['ms_per_step'])"
echo "== conv phase stamps (diagnostic builds, restored afterwards)"; bash scripts/stamp_conv_run.sh > $O/conv_stamps.txt 2>&1; bash scripts/stamp_conv1_run.sh >> $O/conv_stamps.txt 2>&1; bash scripts/stamp_run.sh > $O/alignconv_stamps.txt 2>&1; grep -v amdgpu.ids $O/conv_stamps.txt | cut -c1-220
echo "== nms device totals (measurement build, restored afterwards)"; bash scripts/nms_debug.sh > $O/nms_debug.txt 2>&1; bash scripts/nms_bench_debug.sh >> $O/nms_debug.txt 2>&1; cat $O/nms_debug.txt | cut -c1-200
