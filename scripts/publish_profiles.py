#!/usr/bin/env python3
"""assemble the committed profiles/r03_* files from what scripts/gpu_round3_final.sh left under gpurun_out/ (run in the
build container, after the GPU call): python scripts/publish_profiles.py"""
import json, os, shutil, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles")
RN = "r03"
F = os.path.join(G, "r3final")
head = subprocess.run(["git", "-C", R, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()

# 1. HBM traffic record bench.py reports
t = json.load(open(os.path.join(G, "pmc_bench", "traffic.json")))
t["git_head"] = head
json.dump(t, open(os.path.join(P, RN + "_traffic.json"), "w"), indent=1)

# 2. AlignConv / conv-tower counters of the bench command
def cat(path):
    return open(path).read() if os.path.exists(path) else ""
a = t["kernels"]["align_conv_pyramid"]
alg = 174592 * 512 * 2 + 256 * 2304 * 2 + 174592 * 20
with open(os.path.join(P, RN + "_alignconv_pyramid_pmc.txt"), "w") as f:
    f.write(cat(os.path.join(G, "pmc_bench", "summary_k_dcn_patch.txt")))
    f.write("""
# k_dcn_patch<NHWC, anchors> as the default bench step launches it (ONE pyramid-packed launch, 174 592 positions, f16, 256 -> 256),
# rocprofv3 PMC passes over `python bench.py --steps 4 --warmup 2 --no-cpu-baseline` (scripts/pmc_bench.sh, tree %s).
# FETCH_SIZE / WRITE_SIZE in KiB; gfx950: FETCH_SIZE reads half of a wide coalesced read -> read bytes = 2 x %.0f KiB = %.1f MB,
# WRITE_SIZE exact -> %.1f MB; traffic per launch = %.1f MB vs %.1f MB algorithmic (in + out + filter + anchors) = %.2fx
# (rounds 1 and 2: 256.2 / 256.4 MB; the kernel is unchanged this round apart from the removal of its dominated variants).
""" % (head, a["fetch_kib"], 2 * a["fetch_kib"] * 1024 / 1e6, a["write_kib"] * 1024 / 1e6, a["bytes"] / 1e6, alg / 1e6, a["bytes"] / alg))
shutil.copy(os.path.join(G, "pmc_bench", "summary_k_conv_f16_9_4.txt"), os.path.join(P, RN + "_conv_tower_pmc.txt"))

# 3. NMS at 200 k rows: counters, memory-copy trace, timeline of one call, device-side totals of a measurement build
rep = cat(os.path.join(F, "nms_pmc_report.txt"))
nms_ms = [json.loads(l)["ms"] for l in open(os.path.join(F, "ops_report.jsonl")) if l.startswith("{") and '"ml_nms_rotated"' in l and '"n": 200000' in l]
with open(os.path.join(P, RN + "_nms_200k_pmc.txt"), "w") as f:
    f.write("""# rotated ml-NMS at BASELINE configs[4] (200 000 rows x 15 labels, thr 0.5), round 3 (tree %s); round 2: r02_nms_200k_pmc.txt
# (0.81 ms per call, cull 404 us).  Command and counter sets: rocprofv3 --kernel-trace --pmc <set> -- python scripts/bench_ops.py
# --which nms200k (scripts/pmc_cmd.sh, report by scripts/nms_pmc_report.py).  Whole call (HIP events, un-profiled): %.3f ms.
# What changed this round (DESIGN.md section 4, "Round-3 work on the kernels"): the cull runs in POSITION space (rows sorted by
# (label, Morton code); score ranks carried as a per-row key, so the score sort and the position sort run side by side on forked
# streams), an area-ratio test before the circle test, rotation by v_mov_dpp wave_ror instead of v_readlane broadcasts, an
# oversubscribed strided grid (8192 workgroups) instead of claimed tiles, the per-label scan folded into the meta kernels, and
# the keep list written by two small count / write kernels instead of the rocPRIM partition.
""" % (head, nms_ms[-1] if nms_ms else float("nan")))
    f.write(rep)
    f.write("\n# rocprofv3 --kernel-trace --memory-copy-trace --stats, memory copy stats of the same command:\n")
    f.write(cat(os.path.join(F, "memcpy", "run_memory_copy_stats.csv")))
    f.write("\n# device-side totals of a -DS2A_MEASURE build (scripts/nms_debug.sh, scripts/nms_bench_debug.sh; the shipped library has none of\n"
            "# these stamps): cycles per wave of the cull by phase, and tiles / pairs / edges / alive rows per round; the last two blocks are\n"
            "# the detector's segmented call (8 chips x 15 classes) without and with the spatial cull\n")
    f.write(cat(os.path.join(F, "nms_debug.txt")))
shutil.copy(os.path.join(F, "nms_timeline.txt"), os.path.join(P, RN + "_nms_200k_timeline.txt"))

# 3b. box_iou_rotated at 10 k x 10 k: counters of the kernels of one call and its timeline
with open(os.path.join(P, RN + "_iou_10k_pmc.txt"), "w") as f:
    f.write("""# box_iou_rotated at 10 000 x 10 000 (BASELINE configs[0] shape; 1.1 %% of the pairs overlap), tree %s.
# Counters: rocprofv3 --kernel-trace --pmc <set> -- python scripts/bench_ops.py --which iou10k (scripts/pmc_cmd.sh, report by
# scripts/nms_pmc_report.py).  Timeline of ONE call (scripts/iou_timeline.sh; q2 = the forked zero-fill stream) at the end.
# Unchanged against round 2 in structure: pair finder -> exact pass -> scatter on one stream, the paced zero-fill beside them; the
# round-3 attempts on it (two half-height finder launches, pace sweep, DPP rotation in the finder) measured no gain and are not in
# the tree (DESIGN.md section 4).
""" % head)
    f.write(cat(os.path.join(F, "iou_pmc_report.txt")))
    f.write("\n# timeline of one call (us from the start of the call):\n")
    f.write(cat(os.path.join(F, "iou_timeline.txt")))

# 3c. fused deformable-convolution backward: per-kernel stats of scripts/bwd_trace.sh; kernels of one captured detect() replay
shutil.copy(os.path.join(F, "bwd_trace.txt"), os.path.join(P, RN + "_dcn_backward_kernel_stats.txt"))
shutil.copy(os.path.join(F, "graph_replay_kernels.txt"), os.path.join(P, RN + "_graph_replay_kernels.txt"))
if os.path.exists(os.path.join(F, "conv_stamps.txt")):
    with open(os.path.join(P, RN + "_conv_phase_stamps.txt"), "w") as f:
        f.write("# in-kernel phase stamps (s_memtime, diagnostic builds -DS2A_STAMP=1; the shipped library has none) of the pyramid tower\n"
                "# convolution (scripts/stamp_conv_run.sh) and of two full-width 1x1 layers (scripts/stamp_conv1_run.sh), dense random data:\n"
                "# cycles per phase of a workgroup (median over workgroups) and the in-kernel clock they imply (DESIGN.md section 4)\n")
        f.write("".join(l for l in open(os.path.join(F, "conv_stamps.txt")) if "amdgpu.ids" not in l))
        if os.path.exists(os.path.join(F, "alignconv_stamps.txt")):
            f.write("# AlignConv (k_dcn_patch, P3 level at batch 8, dense random data; scripts/stamp_run.sh): matrix (consumer) and loader waves\n")
            f.write("".join(l for l in open(os.path.join(F, "alignconv_stamps.txt")) if "amdgpu.ids" not in l))

# 4. bench: line, steady-state tables, kernel stats
shutil.copy(os.path.join(F, "bench.json"), os.path.join(P, RN + "_bench_line.json"))
for tag, out in (("r3final", RN + "_bench_steady_state.txt"), ("r3final_s1", RN + "_bench_steady_state_streams1.txt")):
    shutil.copy(os.path.join(G, "prof_" + tag, "steady.txt"), os.path.join(P, out))
shutil.copy(os.path.join(G, "prof_r3final", "kernel_stats.csv"), os.path.join(P, RN + "_bench_kernel_stats.csv"))
shutil.copy(os.path.join(G, "prof_r3final", "line.json"), os.path.join(P, RN + "_bench_line_under_rocprof.json"))

# 5. ops report, with the occupancy of the NMS kernels merged into the 200 k line
occ = json.load(open(os.path.join(F, "nms_occupancy.json")))
with open(os.path.join(P, RN + "_ops_report.jsonl"), "w") as f:
    for line in open(os.path.join(F, "ops_report.jsonl")):
        line = line.strip()
        if not line.startswith("{"):
            continue
        d = json.loads(line)
        if d.get("op") == "ml_nms_rotated" and d.get("n") == 200000:
            d["occupancy"] = {k: {"waves_per_cu": v["waves_per_cu"], "pct_of_32": v["occupancy_pct"], "us": v["us"]} for k, v in occ.items()}
            d["occupancy_source"] = "profiles/" + RN + "_nms_200k_pmc.txt (rocprofv3 PMC: SQ_WAVE_CYCLES * 4 / kernel cycles / 256 CUs)"
        f.write(json.dumps(d) + "\n")
print("published for", head)
