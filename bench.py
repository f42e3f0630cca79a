#!/usr/bin/env python3
"""bench.py — throughput of the S2ANet dense-inference hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1 from a bare shell: this process launches N ranks itself (one fresh child process per GPU, before anything
here touches the GPU) and relays rank 0's JSON line; under ``python -m torch.distributed.run --nproc-per-node N``
(WORLD_SIZE already set) it simply is one of the ranks.

One "step" = one pass of the whole inference hot path over one batch of synthetic input PER GPU:
8 device-resident uint8 1024x1024 chips -> fused stem (/255, 7x7/2 conv, ReLU, max-pool: one kernel) -> fp16
R-50 trunk and FPN on this repo's own MFMA convolution kernels (bias / residual / ReLU fused; MIOpen only for FPN's
two tiny stride-2 extra levels) -> S2ANet head, every layer ONE launch for all five FPN levels (pyramid-packed):
fused anchor refine, AlignConv (anchors -> sampling -> 3x3 contraction on the matrix cores), cached ARF expansion +
ORConv with the rotation-invariant pooling in its epilogue, batched top-k / decode, segmented on-device rotated
ml-NMS -> padded detections [8,2000,6] + labels + counts on the device.  With N > 1 every rank runs its own 8 chips
(weak scaling: BASELINE.json configs[3] = batch 64 over 8 GPUs) and the step ends with ONE RCCL all-gather of the
padded detections.

Prints ONE JSON line (rank 0).  `value` = chips/s over all ranks, inputs resident in HBM.
`roofline` = the dominant hand-written kernel — the pyramid-packed AlignConv launch of the same batch, exactly as
the step issues it — timed live with HIP events on the stream it is launched on; `roofline.traffic` comes from the
recorded rocprofv3 PMC passes over this command (profiles/rNN_traffic.json, written by scripts/pmc_bench.sh) and is
null + `traffic_stale` when the kernel sources changed since.  `cpu_baseline` = the same pipeline for a few chips
on the host cores (oracle/pipeline.py: torch CPU convolutions + the oracle's AlignConv / ARF / pooling / decode +
the reference's own CPU ml_nms_rotated when oracle/_ref is present), with the per-stage seconds as fields.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

CHIP = 1024
BATCH_PER_GPU = 8
NUM_CLASSES = 15
PEAK_F16_TFLOPS = 2500.0   # dense MFMA f16/bf16, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0


def traffic_json():
    """newest profiles/rNN_traffic.json (one per round, written by scripts/pmc_bench.sh on the GPU box)"""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_traffic.json")))
    return found[-1] if found else None


def _sha16(paths):
    import hashlib
    h = hashlib.sha256()
    for p in paths:
        with open(os.path.join(ROOT, p), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def recorded_traffic(key, batch, pixels):
    """HBM bytes per launch of a kernel as the rocprofv3 PMC passes over this very command recorded them
    (scripts/pmc_bench.sh -> scripts/pmc_summary.py --json -> profiles/rNN_traffic.json: 2 x FETCH_SIZE as the gfx950
    correction prescribes + WRITE_SIZE, averaged over the kernel's dispatches).  A record is only valid for the
    kernel sources and launch shape it was taken on: otherwise (None, True) = unknown and stale."""
    try:
        with open(traffic_json()) as f:
            rec = json.load(f)["kernels"][key]
        if rec.get("batch") != batch or rec.get("pixels") != pixels:
            return None, True
        if rec.get("source_sha16") != _sha16(rec["sources"]):
            return None, True
        return float(rec["bytes"]), False
    except (OSError, KeyError, ValueError, TypeError):          # no record (TypeError: no file at all)
        return None, True


def recorded_occupancy(group):
    """occupancy / VALU-utilisation record of the rotated-NMS and rotated-IoU kernels from the rocprofv3 PMC passes of the
    round's evidence run (scripts/gpu_round6_final.sh -> scripts/nms_pmc_report.py --json -> profiles/rNN_ops_occupancy.json;
    `group` = "ml_nms_200k" | "box_iou_10k").  Valid only for the kernel sources it was taken on: otherwise
    {"stale": True} -- never a number measured on other code."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_ops_occupancy.json")))
    try:
        with open(found[-1]) as f:
            rec = json.load(f)
        if rec.get("source_sha16") != _sha16(rec["sources"]):
            return {"stale": True, "record": os.path.basename(found[-1])}
        out = dict(rec[group])
        out["record"] = os.path.basename(found[-1])
        out["unit"] = "achieved waves per CU (of 32), from SQ_WAVE_CYCLES / kernel cycles; valu_util_pct = share of SIMD issue cycles"
        return out
    except (OSError, KeyError, ValueError, TypeError, IndexError):
        return {"stale": True, "record": None}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)     # ~175 ms timed at 4.4 ms per step (>= 100 ms)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="chips per GPU per step")
    ap.add_argument("--candidates", type=int, default=5000,
                    help="(box,class) scores above 0.05 per chip the synthetic classifier is calibrated to")
    ap.add_argument("--dtype", default="f16", choices=["f16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ops", action="store_true", help="skip the `ops` object (BASELINE configs[0]/[1]/[4] as stated)")
    ap.add_argument("--no-fam-cls", action="store_true",
                    help="skip the FAM classification branch (unused at inference; the reference evaluates it)")
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured HIP graph")
    ap.add_argument("--no-graph", action="store_true",
                    help="never replay from a graph (with N > 1 ranks the default is decided by the measured host issue time)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) also for ONE rank and send the detections through "
                         "all_gather_into_tensor instead of the world-of-one copy: executes the real collective path "
                         "(backend, side stream, events) on a one-GPU box.  S2A_BENCH_FORCE_DIST=1 does the same")
    ap.add_argument("--streams", type=int, default=3,
                    help="batches in flight: step i runs on HIP stream i %% S (independent batches; S > 1 lets the "
                         "latency-bound NMS tail of one batch overlap the convolutions of the next)")
    return ap.parse_args()


def calibrate_cls_bias(model, imgs, target_per_chip, logit_std=1.5):
    """Random N(0,0.01) weights give the same score 0.01 everywhere (bias init -4.595, head.py:232)
    and no detection.  Give the synthetic classifier a realistic spread (scale odm_cls_head.weight so
    the logits have std `logit_std`) and shift its bias so that about `target_per_chip` (box, class)
    scores of the per-level top-k candidates exceed the 0.05 threshold — the post-processing then
    does real work on distinct scores."""
    head = model.head
    with torch.no_grad():
        x = imgs.to(next(model.parameters()).dtype).div_(255.0)
        p = model.features_to_pred(x)
        raw = torch.cat([l.float().reshape(-1) for l in p[2]])
        head.odm_cls_head.weight.mul_(logit_std / max(raw.std().item(), 1e-6))
        p = model.features_to_pred(x)
        _, logits = head.candidates(p, raw_logits=True)     # [B,n,C] pre-sigmoid, f32
        frac = min(max(target_per_chip / (logits.shape[1] * logits.shape[2]), 1e-5), 0.999)
        k = max(1, int(round(frac * logits.numel())))
        q = torch.topk(logits.reshape(-1), k)[0][-1].item()
        head.odm_cls_head.bias.add_(math.log(0.05 / 0.95) - q)   # uniform shift: ranking unchanged
        p = model.features_to_pred(x)
        _, scores = head.candidates(p)
        got = (scores > 0.05).sum().item() / scores.shape[0]
    return got


WARM_SECONDS = 0.25


def _time_launches(fn, iters=100):
    """HIP events on the launching stream (torch's current stream) around `iters` launches, after WARM_SECONDS of the same
    launches back to back: the operands of these measurements are built on the host, the GPU idles meanwhile and comes back
    at a low clock -- three warm-up launches measured the ramp (P3 x 8 AlignConv: 200-217 us cold, 167-177 us after 0.3 s;
    in-kernel stamps 1.69 GHz against 2.1-2.3 GHz, docs/HISTORY.md, round 4)"""
    t_w = time.perf_counter()
    n_w = 0
    while n_w < 3 or time.perf_counter() - t_w < WARM_SECONDS:
        fn()
        n_w += 1
        if n_w % 8 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 1e3 / iters


def capture_head_operands(model, imgs, max_candidates=None):
    """one step with S2ANetHead.capture armed: the pyramid-packed FPN features, the refined anchors and
    the level table that the step's AlignConv / conv-tower launches really see (None: per-level path).
    max_candidates as the timed step passes it: the same launch sequence (an uncapped call has 5 344 rows per segment
    and takes the spatial NMS path with its library sorts)"""
    model.head.capture = {}
    model.detect(imgs, max_candidates=max_candidates)
    torch.cuda.synchronize()
    cap, model.head.capture = model.head.capture, None
    return cap if cap else None


def measure_alignconv(model, batch, dtype, cap, iters=100):
    """dominant hand-written kernel of the path: the fused AlignConv launch exactly as the step issues it
    -- ONE pyramid-packed launch over the five FPN levels of the whole batch (21 824 positions per chip),
    on the step's own activations and refined anchors (the duration depends on the data: clocks).
    Per-unit figures: SURVEY 8(d) config 2 -- 2*256*2304 flop per position (19.33 GFLOP per P3 image)."""
    dev = next(model.parameters()).device
    C = O = 256
    es = 2 if dtype == torch.float16 else 4
    if cap is not None:
        from s2anet_amd import pyramid as P
        layout, x, anc = cap["layout"], cap["x"], cap["anchors"]
        wp = model.head.align_conv.packed_weight(dtype)
        sec = _time_launches(lambda: P.align_conv(layout, x, anc, wp, O), iters)
        npos = layout.pixels
        shape = "five FPN levels pyramid-packed, %d positions" % npos
        kname = "k_dcn_patch"
    else:           # f32 / per-level path: P3 only, synthetic operands of SURVEY 8(d) config 2
        from s2anet_amd.alignconv import align_conv_forward
        H = W = CHIP // 8
        g = torch.Generator(device="cpu").manual_seed(1234)
        x = torch.randn(batch, C, H, W, generator=g).to(dev, dtype).contiguous(memory_format=torch.channels_last)
        ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
        anc = torch.zeros(batch, H, W, 5)
        anc[..., 0] = xs * 8 + 3.5 + torch.randn(batch, H, W, generator=g) * 4
        anc[..., 1] = ys * 8 + 3.5 + torch.randn(batch, H, W, generator=g) * 4
        anc[..., 2:4] = 32 * torch.exp(torch.randn(batch, H, W, 2, generator=g) * 0.5)
        anc[..., 4] = (torch.rand(batch, H, W, generator=g) - 0.25) * math.pi
        anc = anc.to(dev)
        wp = model.head.align_conv.packed_weight(dtype)
        sec = _time_launches(lambda: align_conv_forward(x, anc, wp, 8, relu=True, packed=True, out_channels=O), iters)
        npos = batch * H * W
        shape = "P3 128x128"
        kname = "k_dcn_patch" if es == 2 else "k_dcn_mfma"
    flops = 2.0 * O * C * 9 * npos
    alg_bytes = npos * C * es + npos * O * es + O * C * 9 * es + npos * 5 * 4   # in + out + weight + anchors
    peak = PEAK_F16_TFLOPS if dtype == torch.float16 else PEAK_F32_TFLOPS
    ach = flops / sec / 1e12
    # HBM traffic per launch: a recorded measurement (PMC passes cannot run inside this process), see recorded_traffic
    traffic, stale = recorded_traffic("align_conv_pyramid", batch, npos) if (cap is not None and es == 2) else (None, True)
    probe = None
    if cap is not None:
        from s2anet_amd.alignconv import pack_weight
        g = torch.Generator(device="cpu").manual_seed(7)
        xd = torch.randn(x.shape, generator=g).to(x.device, x.dtype)
        wdn = pack_weight((torch.randn(O, C, 3, 3, generator=g) * 0.02).to(x.device).to(dtype), dtype)
        xz, wzn = torch.zeros_like(x), pack_weight(torch.zeros(O, C, 3, 3, device=x.device, dtype=dtype), dtype)
        probe = _clock_probe(lambda: P.align_conv(layout, xz, anc, wzn, O), lambda: P.align_conv(layout, xd, anc, wdn, O))
    return {
        "kernel": "%s (fused AlignConv: anchors -> sampling -> 3x3 contraction -> ReLU; %s, batch %d, %s)"
                  % (kname, shape, batch, "f16" if es == 2 else "f32"),
        "timing": "%d back-to-back launches of the step's own launch on one stream, alone on the GPU (HIP events); inside "
                  "the timed region the kernels of the batches in flight overlap and each takes longer" % iters,
        "bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
        "frac": round(ach / peak, 4), "traffic": traffic, "traffic_stale": stale,
        "avg_launch_us": round(sec * 1e6, 1),
        "clock_probe": probe,
        "flops_per_launch": flops,
        "hbm_algorithmic_bytes_per_launch": alg_bytes,
        "hbm_achieved_GBs": round(alg_bytes / sec / 1e9, 1),
        "hbm_frac_of_8TBs": round(alg_bytes / sec / 1e9 / PEAK_HBM_GBS, 4),
    }


def _clock_probe(launch_zero, launch_dense, iters=60):
    """the same launch on all-zero and on dense random operands (scripts/pyr_power_probe.py): equal instruction streams, so
    the two times differ only by the clock the chip holds -- a reader can split `frac` into kernel and DVFS"""
    return {"zeros_us": round(_time_launches(launch_zero, iters) * 1e6, 1),
            "dense_us": round(_time_launches(launch_dense, iters) * 1e6, 1),
            "note": "same launch, all-zero vs dense N(0,1) operands; the step's own (ReLU-sparse) data is avg_launch_us"}


def measure_conv_tower(model, cap, iters=100):
    """the kernel with the largest share of the step: the patch-staged 3x3 convolution of the head towers
    (256 -> 256 + bias + ReLU), as the step issues it -- one pyramid-packed launch on the step's FPN features"""
    from s2anet_amd import pyramid as P
    layout, x = cap["layout"], cap["x"]
    w, b, o = model.head.fam_reg_ls[0][0].packed_args()
    sec = _time_launches(lambda: P.conv3x3(layout, x, w, b, o, relu=True), iters)
    flops = 2.0 * 256 * 2304 * layout.pixels
    ach = flops / sec / 1e12
    traffic, stale = recorded_traffic("conv_tower_pyramid", layout.batch, layout.pixels)
    from s2anet_amd.fused import conv_pack_weight
    g = torch.Generator(device="cpu").manual_seed(7)
    xd = torch.randn(x.shape, generator=g).to(x.device, x.dtype)
    wd = conv_pack_weight((torch.randn(256, 256, 3, 3, generator=g) * 0.02).to(x.device).half())
    xz, wz = torch.zeros_like(x), torch.zeros_like(w)
    probe = _clock_probe(lambda: P.conv3x3(layout, xz, wz, b, o, relu=True), lambda: P.conv3x3(layout, xd, wd, b, o, relu=True))
    return {"kernel": "k_conv_f16<9,4,2> (head conv tower 3x3 256->256 + bias + ReLU; five FPN levels pyramid-packed, "
                      "%d positions, f16)" % layout.pixels,
            "timing": "%d back-to-back launches of the step's own launch, alone on the GPU" % iters,
            "bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach / PEAK_F16_TFLOPS, 4),
            "traffic": traffic, "traffic_stale": stale,
            "avg_launch_us": round(sec * 1e6, 1), "flops_per_launch": flops,
            "clock_probe": probe,
            "hbm_algorithmic_bytes_per_launch": int(layout.pixels * 512 * 2 + 256 * 2304 * 2)}


def _ops_inputs_rboxes(rng, n, span=1024.0):
    import numpy as np
    b = np.empty((n, 5), np.float32)
    b[:, :2] = rng.uniform(0, span, (n, 2))
    b[:, 2:4] = rng.uniform(4, 100, (n, 2))
    b[:, 4] = rng.uniform(-np.pi / 4, 3 * np.pi / 4, n)
    return b


def measure_ops(dev, with_cpu=True):
    """BASELINE configs[0], [1] and [4] AS STATED, on the record the driver keeps (outside the e2e timed region, about
    a second of GPU time): box_iou_rotated 10 k x 10 k, AlignConv forward on ONE [1,256,128,128]
    level in f16 and f32 (config 2 anchors), ml_nms_rotated on 200 k rows x 15 labels -- HIP events on the launching
    stream -- each beside the reference's CPU figure on a bounded sample (the reference's own CPU ops from oracle/_ref
    when present, 1 thread: they are serial loops; AlignConv has no CPU reference, the oracle port stands in)."""
    import numpy as np
    import s2anet_amd as S
    from s2anet_amd.alignconv import align_conv_forward, pack_weight
    rng = np.random.default_rng(1234)
    ops = {}
    # configs[0]
    n = 10000
    b1 = torch.from_numpy(_ops_inputs_rboxes(rng, n)).to(dev)
    b2 = torch.from_numpy(_ops_inputs_rboxes(rng, n)).to(dev)
    sec = _time_launches(lambda: S.box_iou_rotated(b1, b2), iters=20)
    byts = n * n * 4 + 2 * n * 20
    ops["box_iou_rotated_10k_x_10k"] = {
        "us": round(sec * 1e6, 1), "Gpairs_s": round(n * n / sec / 1e9, 1),
        # what binds the call is max(HBM write of N*M*4 B, the vector-ALU chain on the caller's stream): the zero-fill
        # (96 us = 4.2 TB/s) runs on a side stream BESIDE pair finder (60 us) -> exact pass (60 us), the scatter (21 us)
        # behind both (kernel timeline: scripts/iou_timeline.sh; utilisation of the two VALU-bound kernels 76 % / 81 %,
        # profiles/r03_iou_10k_pmc.txt) -- the chain, not the write, is the longer leg
        "bound": "max(hbm write of N*M*4 B, VALU chain finder -> exact pass -> scatter); the chain binds",
        "hbm_write_floor_us": round(byts / (PEAK_HBM_GBS * 1e9) * 1e6, 1),
        "valu_chain_us_recorded": {"pair_finder": 60, "exact_pass": 60, "scatter": 21, "zero_fill_beside": 96},
        "alg_bytes": byts, "achieved_GBs": round(byts / sec / 1e9, 1), "hbm_write_frac": round(byts / sec / 1e9 / PEAK_HBM_GBS, 4),
        "occupancy": recorded_occupancy("box_iou_10k")}
    # configs[0], second half: polyiou (DOTA_devkit/polyiou/csrc/polyiou.cpp:108-128) on the 10 k boxes as polygons,
    # 1 M (i, j) pairs of nearby boxes so that most of them overlap (f64 on the vector units, bit-exact by test)
    from s2anet_amd.formats import rbox_to_poly
    m = 1000000
    ii = torch.from_numpy(rng.integers(0, n, m)).to(dev)
    p1 = rbox_to_poly(b1)[ii].double()
    shift = torch.from_numpy(rng.normal(0, 8, (m, 1, 2))).to(dev)
    p2 = (p1.view(m, 4, 2) + shift).reshape(m, 8).contiguous()
    from s2anet_amd.rotated import polyiou_pairs
    overl = float((polyiou_pairs(p1, p2) > 0).double().mean())
    sec = _time_launches(lambda: polyiou_pairs(p1, p2), iters=10)
    ops["polyiou_1M_pairs"] = {"us": round(sec * 1e6, 1), "Mpairs_s": round(m / sec / 1e6, 1), "dtype": "f64",
                               "overlapping_fraction": round(overl, 3), "bound": "fp64 vector ALU (Sutherland-Hodgman per pair)"}
    del b1, b2, p1, p2, ii, shift
    # configs[1]: one P3 level, B = 1
    C = O = 256
    H = W = CHIP // 8
    g = torch.Generator(device="cpu").manual_seed(1234)
    x32 = torch.randn(1, C, H, W, generator=g)
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    anc = torch.zeros(1, H, W, 5)
    anc[..., 0] = xs * 8 + 3.5 + torch.randn(1, H, W, generator=g) * 4
    anc[..., 1] = ys * 8 + 3.5 + torch.randn(1, H, W, generator=g) * 4
    anc[..., 2:4] = 32 * torch.exp(torch.randn(1, H, W, 2, generator=g) * 0.5)
    anc[..., 4] = (torch.rand(1, H, W, generator=g) - 0.25) * math.pi
    anc = anc.to(dev)
    w32 = torch.randn(O, C, 3, 3, generator=g) * 0.01
    flops = 2.0 * O * C * 9 * H * W
    for name, dt, peak in (("f16", torch.float16, PEAK_F16_TFLOPS), ("f32", torch.float32, PEAK_F32_TFLOPS)):
        x = x32.to(dev, dt).contiguous(memory_format=torch.channels_last)
        wp = pack_weight(w32.to(dev, dt), dt)
        sec = _time_launches(lambda: align_conv_forward(x, anc, wp, 8, relu=True, packed=True, out_channels=O), iters=50)
        row = {"us": round(sec * 1e6, 1), "TFLOPs": round(flops / sec / 1e12, 1), "bound": "mfma",
               "mfma_frac": round(flops / sec / 1e12 / peak, 4), "peak_TFLOPs": peak}
        if name == "f32" and not os.environ.get("S2A_DCN_F32", "").startswith("m"):
            # the f32 forward runs on the 16-bit matrix instruction: every f32 product = six bf16 products (k_dcn_x3)
            row["peak_TFLOPs"] = round(PEAK_F16_TFLOPS / 6, 1)
            row["mfma_frac"] = round(flops / sec / 1e12 / (PEAK_F16_TFLOPS / 6), 4)
            row["note"] = ("f32 tensors as three bf16 planes: six 16-bit matrix products per f32 product, so the peak is a sixth of the "
                           "16-bit peak; against the f32 matrix instruction's own peak (%.1f TFLOP/s) the rate is %.2f"
                           % (PEAK_F32_TFLOPS, flops / sec / 1e12 / PEAK_F32_TFLOPS))
        ops["alignconv_1x256x128x128_" + name] = row
    # configs[4]
    n, nl = 200000, 15
    d = torch.from_numpy(_ops_inputs_rboxes(rng, n)).to(dev)
    sc = torch.from_numpy(((rng.permutation(n) + 1) / (n + 1) * 0.95 + 0.05).astype(np.float32)).to(dev)
    lab_h = rng.integers(0, nl, n)
    lab = torch.from_numpy(lab_h.astype(np.float32)).to(dev)
    keep = S.ml_nms_rotated(d, sc, lab, 0.5)
    sec = _time_launches(lambda: S.ml_nms_rotated(d, sc, lab, 0.5), iters=10)
    cnt = np.bincount(lab_h).astype(np.float64)
    pairs = float((cnt * (cnt - 1) / 2).sum())
    ops["ml_nms_rotated_200k_x_15"] = {"ms": round(sec * 1e3, 3), "keep": int(keep.numel()), "same_label_pairs": pairs,
                                       "Tpairs_s": round(pairs / sec / 1e12, 3), "occupancy": recorded_occupancy("ml_nms_200k")}
    # the size ONE image's post-processing hands the drop-in op (utils/bbox_nms_rotated.py:47): 5 000 rows x 15 labels --
    # a synchronous call (the host reads the count); settled by one launch (k_nms_small) since round 5
    n5 = 5000
    keep5 = S.ml_nms_rotated(d[:n5].contiguous(), sc[:n5].contiguous(), lab[:n5].contiguous(), 0.5)
    d5, s5, l5 = d[:n5].contiguous(), sc[:n5].contiguous(), lab[:n5].contiguous()
    sec5 = _time_launches(lambda: S.ml_nms_rotated(d5, s5, l5, 0.5), iters=20)
    ops["ml_nms_rotated_5k_x_15"] = {"us": round(sec5 * 1e6, 1), "keep": int(keep5.numel()),
                                     "note": "synchronous drop-in call incl. the host's wait for the count; one kernel launch"}
    # sizes BETWEEN the one-launch path and the big path: single class at 2 000 rows (one segment beyond
    # k_nms_small's 640 rows: skipped on the host now) and 12 000 rows x 15 labels (800 rows per label: known on the device only,
    # so the call pays the small kernel's launch + wait before the general path)
    d2, s2 = d[:2000].contiguous(), sc[:2000].contiguous()
    from s2anet_amd.rotated import nms_rotated_raw
    t2 = _time_launches(lambda: nms_rotated_raw(d2, s2, 0.5), iters=20)
    d12, s12, l12 = d[:12000].contiguous(), sc[:12000].contiguous(), lab[:12000].contiguous()
    t12 = _time_launches(lambda: S.ml_nms_rotated(d12, s12, l12, 0.5), iters=20)
    ops["nms_fallback_zone"] = {"nms_rotated_2k_us": round(t2 * 1e6, 1), "ml_nms_rotated_12k_x_15_us": round(t12 * 1e6, 1)}
    del d, sc, lab, d5, s5, l5, d2, s2, d12, s12, l12
    # SURVEY 8(f) rows 1 / 2 that got a speed pass in round 6: chip-merge polygon NMS at 20 000 polygons and assign_labels on
    # the 21 824 anchors of a chip
    try:
        from s2anet_amd.rotated import nms_poly, assign_labels
        rp = np.random.default_rng(77)
        polys = rbox_to_poly(torch.from_numpy(_ops_inputs_rboxes(rp, 20000, span=2048.0)).to(dev)).double()
        sc_p = torch.from_numpy((rp.permutation(20000) + 1.0) / 20001.0).to(dev)
        dp = torch.cat([polys, sc_p[:, None]], 1).contiguous()
        kp = nms_poly(dp, 0.3)
        tp = _time_launches(lambda: nms_poly(dp, 0.3), iters=10)
        ops["nms_poly_20k"] = {"ms": round(tp * 1e3, 3), "keep": int(kp.numel()), "dtype": "f64"}
        ra = np.random.default_rng(5)
        lev = []
        for st_ in (8, 16, 32, 64, 128):      # the grid anchors of a 1024^2 chip (models/anchors.py:75-126): centres x * s + 0.5 (s - 1), side 4 s
            n_ = CHIP // st_
            ys_, xs_ = np.meshgrid(np.arange(n_, dtype=np.float32), np.arange(n_, dtype=np.float32), indexing="ij")
            lev.append(np.stack([xs_ * st_ + 0.5 * (st_ - 1), ys_ * st_ + 0.5 * (st_ - 1), np.full_like(xs_, 4.0 * st_),
                                 np.full_like(xs_, 4.0 * st_), np.zeros_like(xs_)], -1).reshape(-1, 5))
        a = np.concatenate(lev).astype(np.float32)
        a[:, 4] = ra.uniform(-0.7, 2.3, a.shape[0])
        A = torch.from_numpy(a).to(dev)
        res = {}
        for ng in (32, 300):
            G = torch.from_numpy(_ops_inputs_rboxes(ra, ng)).to(dev)
            res["gts_%d_us" % ng] = round(_time_launches(lambda: assign_labels(A, G), iters=20) * 1e6, 1)
        ops["assign_labels_21824_anchors"] = res
        del dp, A
    except Exception as e:
        ops["nms_poly_20k"] = {"failed": repr(e)}
    # SURVEY 8(f) row 1 at the AlignConv shape: deform_conv backward, P3 x batch 8, f16 and f32, AlignConv-like offsets
    for dt, tag in ((torch.float16, "f16"), (torch.float32, "f32")):
        key = "deform_conv_backward_8x256x128x128_" + tag
        try:
            from s2anet_amd.alignconv import align_offsets
            from s2anet_amd.dcn import deform_conv_backward_input_cuda, deform_conv_backward_parameters_cuda, _fused_backward
            Bb = 8
            xb = torch.randn(Bb, C, H, W, generator=g).to(dev, dt)
            ancb = anc.expand(Bb, -1, -1, -1).contiguous().view(Bb, -1, 5)
            offb = align_offsets(ancb, (H, W), 8, 3).to(dt).contiguous()
            wb = w32.to(dev, dt)
            gob = torch.randn(Bb, O, H, W, generator=g).to(dev, dt)
            gi, goff, gw = torch.zeros_like(xb), torch.zeros_like(offb), torch.zeros_like(wb)
            args = (3, 3, 1, 1, 1, 1, 1, 1, 1, 1)
            t_in = _time_launches(lambda: deform_conv_backward_input_cuda(xb, offb, gob, gi, goff, wb, None, *args, Bb), iters=5)
            t_w = _time_launches(lambda: deform_conv_backward_parameters_cuda(xb, offb, gob, gw, None, None, *args, 1.0, Bb), iters=5)
            t_both = _time_launches(lambda: _fused_backward(xb, offb, wb, gob), iters=5)
            flop = 2.0 * O * C * 9 * Bb * H * W
            ops[key] = {
                "input_offset_ms": round(t_in * 1e3, 3), "weight_ms": round(t_w * 1e3, 3), "both_one_call_ms": round(t_both * 1e3, 3),
                "GFLOP_each": round(flop / 1e9, 1),
                "mfma_frac_one_call": round(2 * flop / t_both / 1e12 / (PEAK_F16_TFLOPS if dt == torch.float16 else PEAK_F32_TFLOPS), 4),
                "note": "fused kernels, no columns tensor, weight gradient without atomics (host-side tensor conversions included); "
                        "one call = what DeformConvFunction.backward runs"
                        + ("" if dt == torch.float16 else "; mfma_frac is of the f32 matrix instruction's peak: the column gradient runs on it, "
                           "the weight gradient on six bf16 products per f32 product (k_dcn_bwd_weight_x3)")}
            del xb, offb, gob, gi, goff, gw
        except Exception as e:          # a report, never a reason to lose the line
            ops[key] = {"failed": repr(e)}
    if with_cpu:
        try:
            ops["cpu_reference"] = _ops_cpu_figures()
        except Exception as e:      # a report, never a reason to lose the GPU numbers
            ops["cpu_reference"] = {"failed": repr(e)}
    return ops


def _ops_cpu_figures():
    """bounded CPU samples (~6 s in all) of the same ops on the host cores.  Every figure: one untimed warm-up call (page
    faults, OpenMP pool start-up, lazy imports), then the BEST of three timed calls; thread counts are pinned and stated
    (the reference's CPU ops and its SWIG polyiou are serial loops: 1 thread; the oracle port of the deformable
    convolution -- the reference has no CPU path for it -- runs OpenMP on `cores` threads)."""
    import ctypes
    import numpy as np
    import oracle
    from oracle import ref
    rng = np.random.default_rng(4321)
    out = {}

    def best_of_3(fn):
        fn()
        ts = []
        for _ in range(3):
            t = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t)
        return min(ts)

    def set_omp(nt):
        try:
            ctypes.CDLL("libgomp.so.1").omp_set_num_threads(int(nt))
        except OSError:
            pass

    def get_omp():
        try:
            return int(ctypes.CDLL("libgomp.so.1").omp_get_max_threads())
        except OSError:
            return None
    nth, omp_before = torch.get_num_threads(), get_omp()
    torch.set_num_threads(1)
    set_omp(1)
    try:
        f = ref.box_iou_rotated()
        m = 1000
        a, b = _ops_inputs_rboxes(rng, m), _ops_inputs_rboxes(rng, m)
        ta, tb = torch.from_numpy(a), torch.from_numpy(b)
        dt = best_of_3((lambda: f(ta, tb)) if f is not None else (lambda: oracle.box_iou_rotated(a, b, sort_mode=oracle.SORT_CPU)))
        out["box_iou_rotated"] = {"kind": "reference" if f is not None else "port", "sample": "%d x %d" % (m, m), "cores": 1,
                                  "s": round(dt, 4), "Mpairs_s": round(m * m / dt / 1e6, 3), "timing": "warm-up + best of 3"}
        fp = ref.polyiou()
        m = 20000
        P1 = oracle.rboxes_to_polys(a)[rng.integers(0, len(a), m)].astype(np.float64)
        P2 = P1 + np.repeat(rng.normal(0, 8, (m, 1, 2)), 4, 1).reshape(m, 8)
        if fp is not None:
            dt = best_of_3(lambda: [fp(P1[k], P2[k]) for k in range(m)])
            kind = "reference"
        else:
            dt = best_of_3(lambda: [oracle.polyiou(P1[k], P2[k]) for k in range(m)])
            kind = "port"
        out["polyiou"] = {"kind": kind, "sample": "%d pairs through the reference's SWIG module, one Python call per pair as "
                                                  "DOTA_devkit/ResultMerge_multi_process.py:62-123 calls it" % m, "cores": 1,
                          "s": round(dt, 4), "Mpairs_s": round(m / dt / 1e6, 4), "timing": "warm-up + best of 3"}
        fn = ref.ml_nms_rotated()
        m = 2000
        d = _ops_inputs_rboxes(rng, m)
        sc = ((rng.permutation(m) + 1) / (m + 1)).astype(np.float32)
        lab = rng.integers(0, 15, m).astype(np.float32)
        td, ts_, tl = torch.from_numpy(d), torch.from_numpy(sc), torch.from_numpy(lab)
        dt = best_of_3((lambda: fn(td, ts_, tl, 0.5)) if fn is not None else
                       (lambda: oracle.ml_nms_rotated(d, sc, lab, 0.5, rule=oracle.RULE_GE, sort_mode=oracle.SORT_CPU)))
        out["ml_nms_rotated"] = {"kind": "reference" if fn is not None else "port", "sample": "%d rows x 15 labels (quadratic)" % m,
                                 "cores": 1, "s": round(dt, 4), "timing": "warm-up + best of 3"}
    finally:
        torch.set_num_threads(nth)
    ncores = min(os.cpu_count() or 1, 16)
    set_omp(ncores)
    x = rng.standard_normal((1, 256, 32, 32)).astype(np.float32)
    w = (rng.standard_normal((256, 256, 3, 3)) * 0.01).astype(np.float32)
    off = rng.standard_normal((1, 18, 32, 32)).astype(np.float32)
    try:
        dt = best_of_3(lambda: oracle.deform_conv_forward(x, off, w))
    finally:
        if omp_before is not None:
            set_omp(omp_before)         # leave the OpenMP pool as it was found
    out["deform_conv_forward"] = {"kind": "port", "sample": "[1,256,32,32] f32 (the reference has no CPU path: oracle port, OpenMP)",
                                  "cores": ncores, "s": round(dt, 4), "GFLOPs": round(2 * 256 * 2304 * 1024 / dt / 1e9, 2),
                                  "timing": "warm-up + best of 3, OMP_NUM_THREADS pinned to `cores`"}
    return out


def cpu_baseline(seed, candidates, chips=3):
    """`chips` chips (one after the other, ~10-15 s) through the same pipeline on the host cores (fp32):
    oracle/pipeline.py = torch CPU convolutions for the carrier and the plain conv layers of the head, the oracle
    for AlignConv / ARF / pooling / decode, and the reference's own CPU ml_nms_rotated (oracle/_ref) when present."""
    import oracle
    from oracle import pipeline, ref
    from s2anet_amd.detector import S2ANet
    torch.manual_seed(seed)
    m = S2ANet(num_classes=NUM_CLASSES).eval()
    for mod in m.modules():
        if mod.__class__.__name__ == "BottleNeck":
            mod.bn3.weight.data.fill_(0.25)
    imgs = torch.randint(0, 256, (chips, 3, CHIP, CHIP), dtype=torch.uint8)
    ncores = min(os.cpu_count() or 1, 16)              # the GPU box's CPU share for one GPU
    torch.set_num_threads(ncores)
    try:
        import ctypes
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(ncores)
    except OSError:
        pass
    ref_nms = ref.ml_nms_rotated()
    # per leg: the carrier convolutions are torch's CPU kernels, AlignConv / ARF / pooling / decode the oracle port (the
    # reference has no CPU path for them), the NMS the reference's own CPU op when oracle/_ref holds it
    legs = {"conv": "torch-cpu", "ops": "port", "nms": "reference" if ref_nms is not None else "port"}
    kind = "mixed" if ref_nms is not None else "port"
    timers = {"oracle_ops_s": 0.0}
    t0 = time.perf_counter()
    t_nms, n_cand, n_keep = 0.0, 0, 0
    for ci in range(chips):
        _, levels = pipeline.forward_chip(m, imgs[ci:ci + 1], timers=timers)
        tp = time.perf_counter()
        scores, deltas, anc, _, _ = pipeline.select_candidates(levels, 2000)
        # same candidate load as the GPU run: threshold at the quantile that yields `candidates`
        sc = torch.from_numpy(scores)
        k = min(candidates, sc.numel() - 1)
        thr = torch.topk(sc.reshape(-1), k + 1)[0][-1].item()
        boxes = oracle.delta2bbox_rotated(anc, deltas)
        mask = sc > thr
        idx = mask.nonzero()
        cb = torch.from_numpy(boxes)[idx[:, 0]].contiguous()
        cs = sc[mask].contiguous()
        cl = idx[:, 1].float().contiguous()
        timers["oracle_ops_s"] += time.perf_counter() - tp
        t_nms0 = time.perf_counter()
        if ref_nms is not None:
            keep = ref_nms(cb, cs, cl, 0.5)
        else:
            keep = oracle.ml_nms_rotated(cb.numpy(), cs.numpy(), cl.numpy(), 0.5, rule=oracle.RULE_GE,
                                         sort_mode=oracle.SORT_CPU)
        t_nms += time.perf_counter() - t_nms0
        n_cand += int(cb.shape[0]); n_keep += len(keep)
    sec = time.perf_counter() - t0
    return {
        "value": round(chips / sec, 4), "unit": "chips/s", "cores": ncores, "kind": kind, "legs": legs,
        "sample": "%d chips 1024x1024 one after the other, fp32, %d NMS candidates per chip, kept %d per chip: torch-CPU "
                  "convolutions (%d threads) + oracle AlignConv/ARF/pooling/decode (OpenMP, %d threads) + %s ml_nms_rotated "
                  "(1 thread: the reference's CPU op is serial)"
                  % (chips, n_cand // chips, n_keep // chips, ncores, ncores, "reference CPU" if ref_nms is not None else "oracle"),
        "chips": chips, "total_s": round(sec, 3),
        "conv_s": round(sec - timers["oracle_ops_s"] - t_nms, 3), "conv_threads": ncores,
        "oracle_ops_s": round(timers["oracle_ops_s"], 3), "oracle_ops_threads": ncores,
        "nms_s": round(t_nms, 3), "nms_threads": 1, "nms_candidates_per_chip": n_cand // chips,
    }


# ------------------------------------------------------------------------------------------------ launcher
def pin_rank_cpus(local_rank, local_world):
    """give this rank a disjoint share of the CPUs the process may run on (N > 1: eight ranks each issue ~100 launches per
    4-5 ms step from Python; without this they migrate over each other's cores).  Called BEFORE anything touches the GPU.
    Returns the number of CPUs of the share, or None when the share would be empty / the call is not available."""
    if os.environ.get("S2A_BENCH_NO_AFFINITY") or local_world <= 1 or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        cpus = sorted(os.sched_getaffinity(0))
        share = len(cpus) // local_world
        if share < 1:
            return None
        mine = cpus[local_rank * share:(local_rank + 1) * share]
        os.sched_setaffinity(0, mine)
        return len(mine)
    except OSError:
        return None


def launch_ranks(n):
    """`python bench.py --gpus N` from a bare shell: start N fresh rank processes (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* in their environment, exactly what torch.distributed.run would set) and relay rank 0's stdout.  Nothing
    in THIS process has touched the GPU (importing torch does not), and the children are started, never exec'ed
    into.  Returns the exit code: non-zero if any rank failed (the others are then stopped by their PIDs)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool (RCCL needs it)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    import threading

    def relay():
        for line in procs[0].stdout:
            sys.stdout.write(line.decode("utf-8", "replace"))
            sys.stdout.flush()
    th = threading.Thread(target=relay, daemon=True)
    th.start()
    rc, live = 0, set(range(n))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print("bench.py: rank %d exited with %d; stopping the other ranks" % (r, code), file=sys.stderr)
                for o in sorted(live):
                    procs[o].terminate()
        time.sleep(0.05)
    th.join(timeout=10)
    return rc


class StubDetector:
    """S2A_BENCH_STUB=1 (tests/test_bench_launcher.py): stands in for the detector so that the launcher, the
    process-group bring-up, the step loop, the DetectionGather slots and the JSON line run on a box without a GPU
    (gloo, CPU tensors).  Its output is a fixed function of the rank; the line it produces is marked "stub"."""

    class _Head:
        max_per_img = 50

    def __init__(self, rank, batch, device):
        self.head = self._Head()
        g = torch.Generator().manual_seed(100 + rank)
        K = self.head.max_per_img
        self.counts = torch.randint(0, K + 1, (batch,), generator=g, dtype=torch.int32).to(device)
        self.dets = torch.rand(batch, K, 6, generator=g).to(device)
        self.labels = torch.randint(0, NUM_CLASSES, (batch, K), generator=g, dtype=torch.int32).to(device)
        for b in range(batch):
            self.dets[b, self.counts[b]:] = 0
            self.labels[b, self.counts[b]:] = -1
        self.ovf = torch.zeros(2, dtype=torch.int64, device=device)
        from s2anet_amd.gather import pack_detections
        self.wire = pack_detections(self.dets, self.labels, self.counts)

    def detect(self, imgs, max_candidates=None, return_overflow=False, dropped_total=None, return_wire=False):
        out = (self.dets, self.labels, self.counts)
        if return_overflow:
            out += (self.ovf,)
        return out + (self.wire,) if return_wire else out


def profiler_preloaded(env=None):
    """True when a GPU tool's library is loaded into this process before main() (rocprofv3 / rocprofiler-sdk /
    roctracer preloads): such a library has initialised the GPU already, and starting child processes from a
    GPU-initialised process is the hop that takes a box of this pool down."""
    env = os.environ if env is None else env
    if env.get("ROCP_TOOL_LIBRARIES") or env.get("ROCPROFILER_REGISTER_FORCE_LOAD") or env.get("HSA_TOOLS_LIB"):
        return True
    return any(k in env.get("LD_PRELOAD", "") for k in ("rocprof", "roctracer", "rocprofiler", "librocm-debug"))


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        if profiler_preloaded():
            print("bench.py: refusing to launch %d ranks from a process a GPU profiler is preloaded into (the GPU is "
                  "already initialised here). Profile multi-rank runs rank by rank: preset RANK / LOCAL_RANK / WORLD_SIZE / "
                  "MASTER_ADDR / MASTER_PORT in a clean shell and wrap rocprofv3 directly around each rank's "
                  "`python bench.py --gpus N`." % args.gpus, file=sys.stderr)
            sys.exit(4)
        sys.exit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert args.gpus == world, "--gpus must equal the number of launched ranks (WORLD_SIZE=%d)" % world
    cpus_per_rank = pin_rank_cpus(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)))   # before any GPU call
    stub = bool(os.environ.get("S2A_BENCH_STUB"))
    if stub:
        if os.environ.get("S2A_BENCH_STUB_FAIL_RANK") == str(rank):     # launcher test: a rank that dies at start-up
            sys.exit(3)
        dev = torch.device("cpu")
        args.streams, args.graph = 1, False
    else:
        if rank == 0 and world == 1 and not (args.no_cpu_baseline and args.no_ops):
            import oracle
            oracle.build()          # the checker / CPU-baseline library, BEFORE anything touches the GPU (it may run make)
        assert torch.cuda.is_available(), "bench.py needs an MI355X"
        ndev = torch.cuda.device_count()
        local_rank = local_rank % max(ndev, 1)     # rehearsal on a 1-GPU box: all ranks share cuda:0
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)

    def sync():
        if not stub:
            torch.cuda.synchronize()
    backend = None
    force_dist = (args.force_dist or bool(os.environ.get("S2A_BENCH_FORCE_DIST"))) and world == 1
    dist_on = world > 1 or force_dist
    if force_dist:
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # nccl == RCCL on ROCm.  S2A_BENCH_BACKEND=gloo is only for rehearsing the multi-rank code
        # path on a box with fewer GPUs than ranks (RCCL refuses two ranks on one device).
        backend = os.environ.get("S2A_BENCH_BACKEND", "gloo" if stub else "nccl")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    from s2anet_amd.gather import DetectionGather
    dtype = torch.float16 if args.dtype == "f16" else torch.float32
    if os.environ.get("S2A_BENCH_CUDNN_BENCHMARK"):     # A/B only: MIOpen's exhaustive find for the two library layers
        torch.backends.cudnn.benchmark = True
    B = args.batch
    if stub:
        model, imgs, got = StubDetector(rank, B, dev), None, 0.0
    else:
        from s2anet_amd.detector import build_synthetic_detector
        model = build_synthetic_detector(num_classes=NUM_CLASSES, seed=1234, dtype=dtype, device=dev,
                                         compute_fam_cls=not args.no_fam_cls)
        # one DISTINCT synthetic batch per batch in flight (and per rank): the streams do not share an input
        batches = []
        for k in range(max(args.streams, 1)):
            g = torch.Generator(device="cpu").manual_seed(1234 + rank + 7919 * k)
            t = torch.randint(0, 256, (B, 3, CHIP, CHIP), dtype=torch.uint8, generator=g).to(dev)
            batches.append(t.contiguous(memory_format=torch.channels_last))
        imgs = batches[0]
        got = calibrate_cls_bias(model, imgs, args.candidates)
    max_cand = int(min(B * 5344 * NUM_CLASSES, max(4 * args.candidates * B, 65536)))
    nslots = max(args.streams, 1)
    # one gather object (output buffers) and one overflow accumulator per batch in flight.  The NMS finish kernel writes
    # the wire buffer and adds the dropped-candidate count to the accumulator: no stock tensor op behind the detector.
    # With one batch at a time the all-gather goes to a side stream behind an event, so that batch i's gather overlaps
    # batch i+1's trunk; with several batches in flight each stream's gather already overlaps the other streams' work.
    # (RCCL only: ProcessGroupNCCL orders its collective behind the current stream with events and never blocks the host;
    # gloo's CUDA path blocks the host in wait() -- the rehearsal backend measured 72 vs 14 ms per step with the side
    # stream -- so the rehearsal keeps the gather on the compute stream)
    side = (dist_on and not stub and backend == "nccl" and args.streams <= 1
            and not os.environ.get("S2A_BENCH_NO_SIDE_GATHER"))
    gathers = [DetectionGather(world, B, model.head.max_per_img, dev, side_stream=side, force_collective=force_dist)
               for _ in range(nslots)] if dist_on else None
    dropped = [torch.zeros((1,), dtype=torch.int64, device=dev) for _ in range(nslots)]
    slot, do_gather, last_wire = [0], [True], [None]

    def detect_only():
        x = imgs if stub else batches[slot[0] % len(batches)]
        k = slot[0] % nslots
        out4 = model.detect(x, max_candidates=max_cand, dropped_total=dropped[k], return_wire=True)
        last_wire[0] = out4[3]
        return out4

    def finish(out4):
        """the exchange step behind a detect (eager even when detect is replayed from a graph: no collective is ever captured)"""
        if gathers is not None and do_gather[0]:
            return gathers[slot[0] % nslots](out4[3])
        return out4[:3]

    def step():
        return finish(detect_only())

    for _ in range(max(args.warmup, 1)):
        out = step()
    sync()
    streams = [torch.cuda.Stream(device=dev) for _ in range(args.streams)] if args.streams > 1 else None
    turn = [0]
    if streams:
        for k, st in enumerate(streams):                     # per-stream workspaces and allocator pools warm
            slot[0] = k
            with torch.cuda.stream(st):
                for _ in range(2):
                    out = step()
        sync()

    def eager_runner():
        if not streams:
            return step()
        slot[0] = turn[0] % len(streams)
        with torch.cuda.stream(streams[slot[0]]):
            step()
        turn[0] += 1

    def issue_and_step_ms(fn, n):
        """(host time for n calls of fn to RETURN, no synchronisation) and (time until the GPU has finished them), per call"""
        sync()
        t1 = time.perf_counter()
        for _ in range(n):
            fn()
        t2 = time.perf_counter()
        sync()
        return (t2 - t1) / n * 1e3, (time.perf_counter() - t1) / n * 1e3

    # N > 1: every rank issues its ~100 launches per step from its own Python thread.  If issuing alone takes more than half
    # a step, the step is replayed from a captured graph (one host call per batch); the all-gather stays eager behind it.
    graph_reason = None
    use_graph = bool(args.graph) and not stub
    if not stub and not args.graph and not args.no_graph and world > 1:
        iss, stp = issue_and_step_ms(eager_runner, 6)
        flag = torch.tensor([1.0 if iss > 0.5 * stp else 0.0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)           # one decision for the whole job
        use_graph = bool(flag.item() > 0)
        graph_reason = "eager host issue %.2f ms of a %.2f ms step on rank %d: %s" % (
            iss, stp, rank, "graph replay" if use_graph else "eager launches kept")
    runner = eager_runner
    if use_graph:
        try:
            nk = len(streams) if streams else 1
            graphs, statics = [], []
            for k in range(nk):
                slot[0] = k
                g_ = torch.cuda.CUDAGraph()
                if streams:
                    with torch.cuda.graph(g_, stream=streams[k]):
                        o4 = detect_only()
                else:
                    with torch.cuda.graph(g_):
                        o4 = detect_only()
                graphs.append(g_)
                statics.append(o4)
            sync()

            def runner():
                k = turn[0] % nk
                slot[0] = k
                # the replay rewrites the STATIC wire buffer: a side-stream gather of this slot's previous turn must have read it
                if streams:
                    with torch.cuda.stream(streams[k]):
                        if gathers is not None:
                            gathers[k].wait()
                        graphs[k].replay()
                        r = finish(statics[k])
                else:
                    if gathers is not None:
                        gathers[k].wait()
                    graphs[k].replay()
                    r = finish(statics[k])
                last_wire[0] = statics[k][3]
                turn[0] += 1
                return r
            for _ in range(nk):
                out = runner()
            sync()
        except Exception as e:      # a capture that fails must not cost the measurement: eager launches, reason on the line
            use_graph, runner = False, eager_runner
            graph_reason = "graph capture failed (%r): eager launches" % (e,)
            sync()
    args.graph = use_graph

    if dist_on:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        runner()
    t_issued = time.perf_counter()
    sync()
    if dist_on:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    host_issue_ms = (t_issued - t0) / args.steps * 1e3          # the host's share: all runner() calls returned, nothing waited for
    per_rank = None
    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
        # attribution of a scaling loss, OUTSIDE the timed region: every rank's step time without the collective
        # (same runner, same streams) and the collective alone on an otherwise idle GPU, gathered to rank 0
        if gathers is not None:
            sync()
            # clones: the attribution loop below gathers into its own object, but the timed slots stay untouched anyway
            out = tuple(t_.clone() for t_ in gathers[(slot[0]) % nslots].unpack())
        def loop_ms(fn, n):
            sync()
            t1 = time.perf_counter()
            for _ in range(n):
                fn()
            sync()
            return (time.perf_counter() - t1) / n * 1e3
        do_gather[0] = False
        compute_ms = loop_ms(runner, args.steps)
        do_gather[0] = True
        dist.barrier()
        w_ = last_wire[0]
        g_alone = DetectionGather(world, B, model.head.max_per_img, dev, side_stream=side, force_collective=force_dist)

        def gather_once():
            g_alone(w_)
            g_alone.wait()
        gather_once()
        gather_ms = loop_ms(gather_once, 20)
        mine = {"rank": rank, "step_ms": round(elapsed / args.steps * 1e3, 3), "compute_ms": round(compute_ms, 3),
                "gather_ms": round(gather_ms, 3)}
        allr = [None] * world
        dist.all_gather_object(allr, mine)
        per_rank = allr
    # the latency-optimal setting beside the throughput one, measured after the timed region: one batch at a time on the
    # default stream (what a scaling run with the side-stream gather uses); eager launches, same step
    single_stream = None
    if streams and not stub and world == 1 and not use_graph:
        slot[0] = 0
        for _ in range(3):
            step()
        iss1, ms1 = issue_and_step_ms(step, args.steps)
        single_stream = {"ms_per_step": round(ms1, 3), "chips_s": round(B / ms1 * 1e3, 1), "steps": args.steps,
                         "host_issue_ms_per_step": round(iss1, 3),
                         "note": "one batch at a time on one HIP stream (--streams 1), measured after the timed region"}
        # the same step replayed from a captured HIP graph: one host call per batch (what N > 1 ranks switch to when their
        # eager issue time exceeds half a step)
        try:
            g1 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g1):
                detect_only()
            g1.replay()
            issg, msg = issue_and_step_ms(g1.replay, args.steps)
            single_stream["graph_replay"] = {"ms_per_step": round(msg, 3), "chips_s": round(B / msg * 1e3, 1),
                                             "host_issue_ms_per_step": round(issg, 3)}
        except Exception as e:
            sync()
            single_stream["graph_replay"] = {"failed": repr(e)}
    # outside the timed region: the static candidate cap must not have cut a single row in any step (the reference
    # never drops a candidate, utils/bbox_nms_rotated.py:29-40) -- one host read
    n_dropped = int(sum(int(d.item()) for d in dropped))
    assert n_dropped == 0, "max_candidates=%d dropped %d NMS candidates: raise the cap" % (max_cand, n_dropped)

    counts = out[2].reshape(-1)
    # the label names the backend that was really initialised (never assumed)
    collective = None if not dist_on else {"nccl": "RCCL (torch.distributed nccl backend)"}.get(
        dist.get_backend(), "%s (rehearsal backend, not RCCL)" % dist.get_backend())
    result = {
        "metric": "1024x1024 DOTA chips/sec (R-50-FPN S2ANet inference)",
        "value": round(world * B * args.steps / elapsed, 2),
        "unit": "chips/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "host_issue_ms_per_step": round(host_issue_ms, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {
            "workload": "BASELINE configs[2]: full R-50-FPN S2ANet inference, batch %d of 1024x1024 synthetic "
                        "uint8 chips per GPU%s%s" % (B, "" if not dist_on else ", detections all-gathered over %s%s" % (
                            collective, " (world of ONE: the collective path executed, not a scaling run)" if force_dist else ""),
                                                     "" if args.streams == 1 else "; %d independent batches in flight on %d HIP streams"
                                                     % (args.streams, args.streams)),
            "chips_per_gpu_per_step": B, "global_batch": world * B, "num_classes": NUM_CLASSES,
            "weights": "seeded random init of the reference architecture; odm_cls bias calibrated",
            "nms_candidates_per_chip": round(got, 1), "detections_per_chip": round(counts.float().mean().item(), 1),
            "nms_candidate_cap": max_cand, "nms_candidates_dropped": n_dropped,
            "distinct_input_batches": 1 if stub else len(batches),
            "fam_cls_branch": not args.no_fam_cls, "hip_graph": bool(args.graph), "batches_in_flight": args.streams,
            "parallelism": "dp%d (one process per GPU)" % world, "collective_backend": collective,
            "cpus_per_rank": cpus_per_rank,
        },
    }
    if graph_reason is not None:
        result["config"]["graph_decision"] = graph_reason
    if single_stream is not None:
        result["single_stream"] = single_stream
    if per_rank is not None:
        # per rank: the timed step, the same step without the all-gather, and the all-gather alone (ms) -- measured after
        # the timed region; with world > 1 a loss shows up either as an uneven compute_ms or as a large gather_ms
        result["per_rank"] = per_rank
        result["config"]["gather_on_side_stream"] = bool(side)
    if stub:
        # the gathered batch must be the rank-major concatenation of what every rank's stub produced
        exp = [StubDetector(r, B, dev) for r in range(world)]
        ok = (torch.equal(out[0], torch.cat([e.dets for e in exp])) and torch.equal(out[1], torch.cat([e.labels for e in exp]))
              and torch.equal(out[2], torch.cat([e.counts for e in exp])))
        result["stub"] = True
        result["gather_equals_concatenation"] = bool(ok)
        result["data"] = "stub detector (launcher / gather rehearsal, not a measurement)"
    if rank == 0:
        if not stub:
            cap = capture_head_operands(model, imgs, max_cand)
            result["roofline"] = measure_alignconv(model, B, dtype, cap)
            if cap is not None:
                result["roofline_conv_tower"] = measure_conv_tower(model, cap)
        if world == 1 and not stub and not args.no_ops:
            result["ops"] = measure_ops(dev, with_cpu=not args.no_cpu_baseline)
        if world == 1 and not args.no_cpu_baseline and not stub:
            try:
                result["cpu_baseline"] = cpu_baseline(1234, args.candidates)
            except Exception as e:  # the baseline is a report, never a reason to lose the GPU number
                result["cpu_baseline"] = {"value": None, "unit": "chips/s", "cores": os.cpu_count(), "kind": "port",
                                          "sample": "failed: %r" % (e,)}
        print(json.dumps(result), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
