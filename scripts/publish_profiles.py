#!/usr/bin/env python3
"""assemble the committed profiles/r06_* files from what scripts/gpu_round6_final.sh left under gpurun_out/ (run in the
build container, after the GPU call): python scripts/publish_profiles.py      (round 4's version of this script is in
the history at a1629f9)"""
import json, os, shutil, subprocess
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles")
RN = "r06"
F = os.path.join(G, "r6final")
head = subprocess.run(["git", "-C", R, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()


def cat(path):
    return "".join(l for l in open(path) if "amdgpu.ids" not in l) if os.path.exists(path) else ""


def copy(src, dst):
    if os.path.exists(src):
        shutil.copy(src, os.path.join(P, RN + dst))
    else:
        print("missing:", src)


# 1. HBM traffic record bench.py reports, and the counter tables behind it
t = json.load(open(os.path.join(G, "pmc_bench", "traffic.json")))
t["git_head"] = head
json.dump(t, open(os.path.join(P, RN + "_traffic.json"), "w"), indent=1)
a = t["kernels"]["align_conv_pyramid"]
alg = 174592 * 512 * 2 + 256 * 2304 * 2 + 174592 * 20
with open(os.path.join(P, RN + "_alignconv_pyramid_pmc.txt"), "w") as f:
    f.write(cat(os.path.join(G, "pmc_bench", "summary_k_dcn_patch.txt")))
    f.write("""
# k_dcn_patch<NHWC, anchors> as the default bench step launches it (ONE pyramid-packed launch, 174 592 positions, f16, 256 -> 256),
# rocprofv3 PMC passes over `python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-ops` (scripts/pmc_bench.sh, tree %s).
# FETCH_SIZE / WRITE_SIZE in KiB; gfx950: FETCH_SIZE reads half of a wide coalesced read -> read bytes = 2 x %.0f KiB = %.1f MB,
# WRITE_SIZE exact -> %.1f MB; traffic per launch = %.1f MB vs %.1f MB algorithmic (in + out + filter + anchors) = %.2fx
# (rounds 1-5: 256.2 / 256.4 / 256.5 / 247.7 / 247.9 MB; round 6 did not touch this kernel)
""" % (head, a["fetch_kib"], 2 * a["fetch_kib"] * 1024 / 1e6, a["write_kib"] * 1024 / 1e6, a["bytes"] / 1e6, alg / 1e6, a["bytes"] / alg))
copy(os.path.join(G, "pmc_bench", "summary_k_conv_f16_9_4.txt"), "_conv_tower_pmc.txt")

# 2. NMS / IoU counters (occupancy) and timelines
with open(os.path.join(P, RN + "_nms_200k_pmc.txt"), "w") as f:
    f.write("# ml_nms_rotated, 200 000 rows x 15 labels, round-6 pipeline (tree %s): rocprofv3 PMC passes over\n"
            "# `python scripts/bench_ops.py --which nms200k` (scripts/pmc_cmd.sh), report by scripts/nms_pmc_report.py\n" % head)
    f.write(cat(os.path.join(F, "nms_pmc_report.txt")))
    f.write("\n# kernel timeline of ONE call (scripts/nms_timeline.sh):\n")
    f.write(cat(os.path.join(F, "nms_timeline.txt")))
with open(os.path.join(P, RN + "_iou_10k_pmc.txt"), "w") as f:
    f.write("# box_iou_rotated, 10 000 x 10 000 (tree %s): PMC passes over `python scripts/bench_ops.py --which iou10k`\n" % head)
    f.write(cat(os.path.join(F, "iou_pmc_report.txt")))
    f.write("\n# kernel timeline of ONE call (scripts/iou_timeline.sh):\n")
    f.write(cat(os.path.join(F, "iou_timeline.txt")))
copy(os.path.join(F, "nms_timeline.txt"), "_nms_200k_timeline.txt")
occ = json.load(open(os.path.join(F, "ops_occupancy.json")))
occ["git_head"] = head
json.dump(occ, open(os.path.join(P, RN + "_ops_occupancy.json"), "w"), indent=1)

# 3. bench: line, steady-state tables, kernel stats, the captured-graph kernel list, the RCCL world-of-one line
copy(os.path.join(F, "bench.json"), "_bench_line.json")
for tag, out in (("r6final", "_bench_steady_state.txt"), ("r6final_s1", "_bench_steady_state_streams1.txt")):
    copy(os.path.join(G, "prof_" + tag, "steady.txt"), out)
copy(os.path.join(G, "prof_r6final", "kernel_stats.csv"), "_bench_kernel_stats.csv")
copy(os.path.join(G, "prof_r6final", "line.json"), "_bench_line_under_rocprof.json")
copy(os.path.join(F, "graph_replay_kernels.txt"), "_graph_replay_kernels.txt")
copy(os.path.join(F, "bench_rccl_world1.json"), "_bench_rccl_world1.json")

# 4. ops report, half decode / NMS effect, clock probe, fixture hashes
with open(os.path.join(P, RN + "_ops_report.jsonl"), "w") as f:
    for line in open(os.path.join(F, "ops_report.jsonl")):
        if line.startswith("{"):
            f.write(line)
copy(os.path.join(G, "half_nms_effect.json"), "_half_nms_effect.json")
copy(os.path.join(F, "pyr_power_probe.jsonl"), "_alignconv_clock_probe.jsonl")
copy(os.path.join(F, "f16_fixture_hashes.json"), "_f16_fixture_hashes.json")
with open(os.path.join(P, RN + "_dcn_backward_kernel_stats.txt"), "w") as f:
    f.write("# deform_conv backward at P3 x 8 (tree %s): `rocprofv3 --kernel-trace --stats` over `python scripts/bench_ops.py --which bwd`\n"
            "# (f32, f16, f16 with wild offsets: the average of k_dcn_bwd_input covers the last two), the op lines of that run, and the\n"
            "# in-kernel phase stamps of k_dcn_bwd_weight_f32 (scripts/bwd32_stamps.sh, -DS2A_MEASURE build)\n" % head)
    f.write(cat(os.path.join(F, "dcn_backward_kernel_stats.txt")))
    f.write("\n# k_dcn_bwd_weight_f32, cycles per 32-position tile (12.3 k of them MFMA):\n")
    f.write(cat(os.path.join(F, "dcn_backward_f32_weight_stamps.txt")))
copy(os.path.join(F, "bench_2rank_gloo.json"), "_bench_2rank_gloo_rehearsal.json")
copy(os.path.join(F, "bench_s1_graph.json"), "_bench_streams1_graph.json")
copy(os.path.join(F, "wino_ab.jsonl"), "_wino_ab_evidence_box.jsonl")
print("published for", head)
