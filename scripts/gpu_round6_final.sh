#!/bin/bash
# end-of-round evidence (round 6): full GPU suite, smoke, NMS / IoU counters (occupancy record for the driver line) and timelines,
# bench counters -> traffic record, steady-state tables (3 streams and 1), ops report, the bench line, the RCCL world-of-one
# line, the half decode / NMS effect, the f16 fixture hashes, kernel list of a captured detect() replay.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6final; mkdir -p $O; cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?; echo "gpu tests rc=$rc"; tail -3 $O/gpu_tests.log | cut -c1-200
if [ $rc -ne 0 ]; then exit 1; fi
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log | cut -c1-200
SETS=("SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
      "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" \
      "SQ_BUSY_CU_CYCLES SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU")
echo "== nms pmc"
bash scripts/pmc_cmd.sh nms200k "scripts/bench_ops.py --which nms200k" "k_nms_cull_lanes k_nms_heavy k_nms_tile_filter k_nms_round k_nms_sp_meta k_spb_hist k_spb_scatter" "${SETS[@]}" > $O/nms_pmc.log 2>&1; echo "nms pmc rc=$?"
python scripts/nms_pmc_report.py gpurun_out/pmc_nms200k --json $O/nms_occupancy.json k_nms_cull_lanes "k_nms_heavy<false>" "k_nms_heavy<true>" k_nms_tile_filter k_nms_round k_nms_sp_meta k_spb_hist k_spb_scatter k_nms_spkeys > $O/nms_pmc_report.txt 2>&1; head -8 $O/nms_pmc_report.txt | cut -c1-170
bash scripts/nms_timeline.sh > $O/nms_timeline.txt 2>&1; grep -v rocprim $O/nms_timeline.txt | head -12 | cut -c1-110
echo "== iou pmc + timeline"
bash scripts/pmc_cmd.sh iou10k "scripts/bench_ops.py --which iou10k" "k_iou_cull_lanes k_iou_heavy" "${SETS[@]}" > $O/iou_pmc.log 2>&1; echo "iou pmc rc=$?"
python scripts/nms_pmc_report.py gpurun_out/pmc_iou10k --json $O/iou_occupancy.json k_iou_cull_lanes k_iou_heavy k_fill_zero k_iou_scatter > $O/iou_pmc_report.txt 2>&1
bash scripts/iou_timeline.sh > $O/iou_timeline.txt 2>&1; cat $O/iou_timeline.txt | cut -c1-120
python - <<PY
import json, hashlib
srcs = ["s2anet_amd/csrc/rotated_ops.hip", "s2anet_amd/csrc/rbox_geom.hpp"]
h = hashlib.sha256()
for p in srcs: h.update(open(p, "rb").read())
rec = {"sources": srcs, "source_sha16": h.hexdigest()[:16],
       "ml_nms_200k": json.load(open("$O/nms_occupancy.json")), "box_iou_10k": json.load(open("$O/iou_occupancy.json"))}
json.dump(rec, open("$O/ops_occupancy.json", "w"), indent=1)
json.dump(rec, open("profiles/r06_ops_occupancy.json", "w"), indent=1)     # the bench below reads this copy
PY
echo "== bench pmc"; bash scripts/pmc_bench.sh > $O/pmc_bench.log 2>&1; echo "pmc bench rc=$?"; cat gpurun_out/pmc_bench/traffic.json | head -c 900; echo
python -c "import json; json.dump(json.load(open('gpurun_out/pmc_bench/traffic.json')), open('profiles/r06_traffic.json', 'w'), indent=1)"
echo "== steady state"; bash scripts/prof_bench.sh r6final --no-ops > $O/prof_bench3.log 2>&1; head -3 $O/prof_bench3.log | cut -c1-160
bash scripts/prof_bench.sh r6final_s1 --streams 1 --no-ops > $O/prof_bench1.log 2>&1; head -3 $O/prof_bench1.log | cut -c1-160
echo "== graph replay kernels"; bash scripts/graph_trace.sh > $O/graph_replay_kernels.txt 2>&1; echo "graph trace rc=$?"; tail -1 $O/graph_replay_kernels.txt
echo "== ops report"; timeout -k 10 600 python scripts/bench_ops.py --which all > $O/ops_report.jsonl 2> $O/ops_report.err; echo "ops rc=$?"; wc -l $O/ops_report.jsonl
echo "== bench"; timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; head -c 400 $O/bench.json; echo
echo "== bench, RCCL world of one (side-stream gather)"; timeout -k 10 300 python bench.py --force-dist --streams 1 --no-cpu-baseline --no-ops 2>/dev/null | grep '^{' > $O/bench_rccl_world1.json; head -c 300 $O/bench_rccl_world1.json; echo
timeout -k 10 300 python bench.py --no-cpu-baseline --no-ops --streams 1 --graph 2>/dev/null | grep '^{' > $O/bench_s1_graph.json; python -c "
import json
d=json.load(open('$O/bench_s1_graph.json')); print('graph replay, 1 stream', d['value'], d['ms_per_step'])"
echo "== bench, 2 ranks sharing the card under gloo (rehearsal of the N > 1 code path: graph decision, per-rank CPU sets)"
S2A_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --streams 1 --steps 10 --no-cpu-baseline --no-ops 2>/dev/null | grep '^{' > $O/bench_2rank_gloo.json; head -c 300 $O/bench_2rank_gloo.json; echo
echo "== Winograd F(2,3) tower kernel against the direct one, same box (+ timing ablations)"
timeout -k 10 300 python scripts/bench_wino.py 3 2>/dev/null | grep kernel > $O/wino_ab.jsonl; cat $O/wino_ab.jsonl | cut -c1-160
echo "== half decode / NMS effect"; timeout -k 10 300 python scripts/half_nms_effect.py > $O/half_nms_effect.log 2>&1; tail -1 $O/half_nms_effect.log | cut -c1-400
echo "== clock probe"; timeout -k 10 300 python scripts/pyr_power_probe.py > $O/pyr_power_probe.jsonl 2>/dev/null; cat $O/pyr_power_probe.jsonl
echo "== deform-conv backward: kernel trace + stamps of the f32 weight kernel"
bash scripts/prof_cmd.sh bwd scripts/bench_ops.py --which bwd > $O/dcn_backward_kernel_stats.txt 2>&1; grep '"op"' gpurun_out/prof_bwd/run.log | cut -c1-260 >> $O/dcn_backward_kernel_stats.txt; tail -3 $O/dcn_backward_kernel_stats.txt | cut -c1-200
bash scripts/bwd32_stamps.sh 2>&1 | grep -v amdgpu.ids > $O/dcn_backward_f32_weight_stamps.txt; cat $O/dcn_backward_f32_weight_stamps.txt
cp gpurun_out/f16_fixture_hashes.json $O/ 2>/dev/null
# keep what comes home under 64 MiB: the raw per-dispatch counter / trace tables are summarised above
find $R/gpurun_out -type f \( -name "*_counter_collection.csv" -o -name "*kernel_trace.csv" -o -name "*.db" -o -name "*.rocpd" -o -name "*_agent_info.csv" \) -delete
find $R/gpurun_out -type f -size +4M -delete
du -sh $R/gpurun_out | cut -f1
