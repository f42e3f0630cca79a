#!/usr/bin/env python3
"""assemble the committed profiles/r02_* files from what scripts/gpu_round2_final.sh left under gpurun_out/ (run in the
build container, after the GPU call): python scripts/publish_profiles.py"""
import json, os, shutil, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles")
F = os.path.join(G, "r2final")
head = subprocess.run(["git", "-C", R, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()

# 1. HBM traffic record bench.py reports
t = json.load(open(os.path.join(G, "pmc_bench", "traffic.json")))
t["git_head"] = head
json.dump(t, open(os.path.join(P, "r02_traffic.json"), "w"), indent=1)

# 2. AlignConv / conv-tower counters of the bench command
def cat(path):
    return open(path).read() if os.path.exists(path) else ""
a = t["kernels"]["align_conv_pyramid"]
alg = 174592 * 512 * 2 + 256 * 2304 * 2 + 174592 * 20
with open(os.path.join(P, "r02_alignconv_pyramid_pmc.txt"), "w") as f:
    f.write(cat(os.path.join(G, "pmc_bench", "summary_k_dcn_patch.txt")))
    f.write("""
# k_dcn_patch<NHWC, anchors> as the default bench step launches it (ONE pyramid-packed launch, 174 592 positions, f16, 256 -> 256),
# rocprofv3 PMC passes over `python bench.py --steps 4 --warmup 2 --no-cpu-baseline` (scripts/pmc_bench.sh, tree %s).
# FETCH_SIZE / WRITE_SIZE in KiB; gfx950: FETCH_SIZE reads half of a wide coalesced read -> read bytes = 2 x %.0f KiB = %.1f MB,
# WRITE_SIZE exact -> %.1f MB; traffic per launch = %.1f MB vs %.1f MB algorithmic (in + out + filter + anchors) = %.2fx
# (round 1: 256.2 MB; the matrix waves' fragment prefetch of this round does not change the traffic; DESIGN.md section 4).
""" % (head, a["fetch_kib"], 2 * a["fetch_kib"] * 1024 / 1e6, a["write_kib"] * 1024 / 1e6, a["bytes"] / 1e6, alg / 1e6, a["bytes"] / alg))
shutil.copy(os.path.join(G, "pmc_bench", "summary_k_conv_f16_9_4.txt"), os.path.join(P, "r02_conv_tower_pmc.txt"))

# 3. NMS at 200 k rows: counters after the rework, memory-copy trace, kernel stats (the report is regenerated here so that
# the two instantiations of the dense pass -- 8-slot first launch, 24-slot redo -- are listed apart)
NMS_KERNELS = ["k_nms_cull", "k_nms_heavy<false>", "k_nms_heavy<true>", "k_nms_tile_filter", "k_nms_round",
               "k_nms_finish_segments", "k_nms_pos_meta"]
rep = subprocess.run([sys.executable, os.path.join(R, "scripts", "nms_pmc_report.py"), os.path.join(G, "pmc_nms200k"),
                      "--json", os.path.join(F, "nms_occupancy.json")] + NMS_KERNELS, capture_output=True, text=True)
assert rep.returncode == 0, rep.stderr
open(os.path.join(F, "nms_pmc_report.txt"), "w").write(rep.stdout)
nms_ms = [json.loads(l)["ms"] for l in open(os.path.join(F, "ops_report.jsonl")) if l.startswith("{") and '"ml_nms_rotated"' in l and '"n": 200000' in l]
with open(os.path.join(P, "r02_nms_200k_pmc.txt"), "w") as f:
    f.write("""# rotated ml-NMS at BASELINE configs[4] (200 000 rows x 15 labels, thr 0.5) AFTER the round-2 rework (tree %s); before:
# r02_nms_200k_pmc_before.txt (2.79 ms per call; cull 1231 us at 14 waves/CU, scan 719 us at 0.25 waves/CU, dense pass 456 us).
# Same command and counter sets: rocprofv3 --kernel-trace --pmc <set> -- python scripts/bench_ops.py --which nms200k
# (scripts/pmc_cmd.sh, report by scripts/nms_pmc_report.py).  Whole call (HIP events, un-profiled): %.2f ms.
# What changed: Morton-sorted 64-row blocks + bounding-box tile filter (83 %% of the tiles never tested), one wave per tile in the cull
# (boxes in registers, circles by v_readlane, no workgroup barrier: k_nms_cull_lanes), separating axes + IoU
# upper bound before the dense pass (7x fewer IoU evaluations), edge list resolved by parallel rounds (no suppression mask, no serial
# scan), wave / workgroup aggregated atomics.  Device -> host traffic: the memory-copy trace below shows host -> device uploads of the
# test inputs only (3 copies); no copy-engine transfer device -> host; the 8-byte keep count of the pybind-shaped entry point goes
# through a shader copy, the segmented entry point of the detector returns nothing to the host.
# Dense IoU pass: k_nms_heavy<false> gives a lane 8 candidate-point slots (16 KB of LDS per workgroup; 98 -> 58 us), k_nms_heavy<true>
# redoes the pairs that need the reference's 24.
""" % (head, nms_ms[0] if nms_ms else float("nan")))
    f.write(cat(os.path.join(F, "nms_pmc_report.txt")))
    f.write("\n# rocprofv3 --kernel-trace --memory-copy-trace --stats, memory copy stats of the same command:\n")
    f.write(cat(os.path.join(F, "memcpy", "run_memory_copy_stats.csv")))
    f.write("\n# per-kernel averages of one profiled run (scripts/prof_cmd.sh):\n")
    f.write(cat(os.path.join(F, "prof_nms.log")))

# 3b. box_iou_rotated at 10 k x 10 k: counters of the kernels of one call and its timeline
with open(os.path.join(P, "r02_iou_10k_pmc.txt"), "w") as f:
    f.write("""# box_iou_rotated at 10 000 x 10 000 (BASELINE configs[0] shape; 1.1 %% of the pairs overlap), tree %s.
# Counters: rocprofv3 --kernel-trace --pmc <set> -- python scripts/bench_ops.py --which iou10k (scripts/pmc_cmd.sh, report by
# scripts/nms_pmc_report.py).  Timeline of ONE call (scripts/iou_timeline.sh; q2 = the forked zero-fill stream) at the end.
# Reading (DESIGN.md section 4, "A saturating store stream ..."): the zero-fill is paced (s_sleep between its stores) so that the pair
# finder (k_iou_cull_lanes: circles by v_readlane, no LDS traffic in the first stage) and the exact pass (8 candidate-point slots per
# lane, VALU-bound) run beside it at their stand-alone speed; the call is bound by that chain, the 400 MB of stores are hidden.
""" % head)
    f.write(cat(os.path.join(F, "iou_pmc_report.txt")))
    f.write("\n# timeline of one call (us from the start of the call):\n")
    f.write(cat(os.path.join(F, "iou_timeline.txt")))

# 4. bench: line, steady-state tables, kernel stats
shutil.copy(os.path.join(F, "bench.json"), os.path.join(P, "r02_bench_line.json"))
for tag, out in (("r2final", "r02_bench_steady_state.txt"), ("r2final_s1", "r02_bench_steady_state_streams1.txt")):
    shutil.copy(os.path.join(G, "prof_" + tag, "steady.txt"), os.path.join(P, out))
shutil.copy(os.path.join(G, "prof_r2final", "kernel_stats.csv"), os.path.join(P, "r02_bench_kernel_stats.csv"))
shutil.copy(os.path.join(G, "prof_r2final", "line.json"), os.path.join(P, "r02_bench_line_under_rocprof.json"))

# 5. ops report, with the occupancy of the NMS kernels merged into the 200 k line
occ = json.load(open(os.path.join(F, "nms_occupancy.json")))
with open(os.path.join(P, "r02_ops_report.jsonl"), "w") as f:
    for line in open(os.path.join(F, "ops_report.jsonl")):
        line = line.strip()
        if not line.startswith("{"):
            continue
        d = json.loads(line)
        if d.get("op") == "ml_nms_rotated" and d.get("n") == 200000:
            d["occupancy"] = {k: {"waves_per_cu": v["waves_per_cu"], "pct_of_32": v["occupancy_pct"], "us": v["us"]} for k, v in occ.items()}
            d["occupancy_source"] = "profiles/r02_nms_200k_pmc.txt (rocprofv3 PMC: SQ_WAVE_CYCLES * 4 / kernel cycles / 256 CUs)"
        f.write(json.dumps(d) + "\n")
print("published for", head)
