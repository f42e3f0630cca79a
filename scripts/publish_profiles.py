#!/usr/bin/env python3
"""assemble the committed profiles/r04_* files from what scripts/gpu_round4_final.sh left under gpurun_out/ (run in the
build container, after the GPU call): python scripts/publish_profiles.py      (round 3's version of this script is in
the history at e5fc921; the NMS / IoU kernels are unchanged since, so their counter reports stay profiles/r03_*)"""
import json, os, shutil, subprocess
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles")
RN = "r04"
F = os.path.join(G, "r4final")
head = subprocess.run(["git", "-C", R, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()


def cat(path):
    return open(path).read() if os.path.exists(path) else ""


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
# (rounds 1-3: 256.2 / 256.4 / 256.5 MB; the kernel's main loop is unchanged this round: it gained the half-coordinate switch in its
# table build and a second, opt-in form of the launch beside it, k_dcn_sym -- r04_alignconv_forms.txt).
""" % (head, a["fetch_kib"], 2 * a["fetch_kib"] * 1024 / 1e6, a["write_kib"] * 1024 / 1e6, a["bytes"] / 1e6, alg / 1e6, a["bytes"] / alg))
shutil.copy(os.path.join(G, "pmc_bench", "summary_k_conv_f16_9_4.txt"), os.path.join(P, RN + "_conv_tower_pmc.txt"))

# 2. the two forms of the pyramid AlignConv launch: time on zero / sparse / dense data, phase stamps of both
with open(os.path.join(P, RN + "_alignconv_forms.txt"), "w") as f:
    f.write("# the pyramid-packed AlignConv launch (batch 8, five FPN levels, 174 592 positions) in its two forms, tree %s:\n"
            "#   k_dcn_patch  8 x 16 tiles, 4 matrix + 4 loader waves (the shipped default)\n"
            "#   k_dcn_sym    16 x 16 tiles, every wave blends and contracts (S2A_DCN_SYM=1)\n"
            "# (a) same launch on all-zero, ReLU-sparse and dense random data, 200 launches each (scripts/pyr_power_probe.py): equal\n"
            "#     instruction streams, so a time that moves with the data is the clock the chip holds, a time that does not is issue / latency\n" % head)
    f.write(cat(os.path.join(F, "pyr_power_probe.jsonl")))
    f.write("# (b) in-kernel phase stamps (s_memtime, -DS2A_STAMP=1 diagnostic builds; the shipped library has none), dense random data\n")
    f.write("# k_dcn_patch: P3 level at batch 8, configs[1] as stated, then the pyramid launch on dense and zero data with the in-kernel clock of a tile (scripts/stamp_run.sh):\n")
    f.write("".join(l for l in open(os.path.join(F, "alignconv_stamps.txt")) if "amdgpu.ids" not in l))
    f.write("# k_dcn_sym, the whole pyramid (scripts/stamp_sym_run.sh; cycles per stage interval = 1 024 cycles of MFMA per SIMD):\n")
    f.write("".join(l for l in open(os.path.join(F, "alignconv_sym_stamps.txt")) if "amdgpu.ids" not in l))
if os.path.exists(os.path.join(F, "conv_stamps.txt")):
    with open(os.path.join(P, RN + "_conv_phase_stamps.txt"), "w") as f:
        f.write("# in-kernel phase stamps of the pyramid tower convolution (scripts/stamp_conv_run.sh), dense random data\n")
        f.write("".join(l for l in open(os.path.join(F, "conv_stamps.txt")) if "amdgpu.ids" not in l))

# 3. bench: line, steady-state tables, kernel stats, the captured-graph kernel list, rehearsal of two ranks
shutil.copy(os.path.join(F, "bench.json"), os.path.join(P, RN + "_bench_line.json"))
for tag, out in (("r4final", RN + "_bench_steady_state.txt"), ("r4final_s1", RN + "_bench_steady_state_streams1.txt")):
    shutil.copy(os.path.join(G, "prof_" + tag, "steady.txt"), os.path.join(P, out))
shutil.copy(os.path.join(G, "prof_r4final", "kernel_stats.csv"), os.path.join(P, RN + "_bench_kernel_stats.csv"))
shutil.copy(os.path.join(G, "prof_r4final", "line.json"), os.path.join(P, RN + "_bench_line_under_rocprof.json"))
shutil.copy(os.path.join(F, "graph_replay_kernels.txt"), os.path.join(P, RN + "_graph_replay_kernels.txt"))
if os.path.exists(os.path.join(F, "bench_2rank_gloo.json")):
    shutil.copy(os.path.join(F, "bench_2rank_gloo.json"), os.path.join(P, RN + "_bench_2rank_gloo_rehearsal.json"))

# 4. ops report; half-coordinate mode records
with open(os.path.join(P, RN + "_ops_report.jsonl"), "w") as f:
    for line in open(os.path.join(F, "ops_report.jsonl")):
        if line.startswith("{"):
            f.write(line)
for src, dst in (("half_mode_effect.json", "_half_mode_effect.json"), ("half_coords_mode.json", "_half_coords_mode.json"),
                 ("half_path_deviation.json", "_half_path_deviation.json")):
    if os.path.exists(os.path.join(G, src)):
        shutil.copy(os.path.join(G, src), os.path.join(P, RN + dst))
if os.path.exists(os.path.join(F, "nms_timeline.txt")):
    shutil.copy(os.path.join(F, "nms_timeline.txt"), os.path.join(P, RN + "_nms_200k_timeline.txt"))
print("published for", head)
