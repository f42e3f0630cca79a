"""Bounded, seeded slices of the randomised oracle checkers (scripts/fuzz_nms.py, fuzz_alignconv.py, fuzz_dcn_backward.py, fuzz_f_ops.py) so
that they are on the driver's record, and the one-launch small-input NMS under load: synchronous 5 000-row ml_nms_rotated
calls while two other streams run the head (the occupancy k_nms_small's cross-workgroup ticket meets in bench.py).
Reference: utils/ml_nms_rotated/src/nms_rotated_cuda.cu:74-137, models/dcn/src/deform_conv_cuda.cpp:262-489,
models/alignconv.py:30-98.  A failing case prints its seed and index: rerun scripts/fuzz_*.py <cases> <seed>."""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from conftest import rand_rboxes, distinct_scores  # noqa: E402

pytestmark = pytest.mark.gpu


def _run(case, seed, n, *args, **kw):
    rng = np.random.default_rng(seed)
    bad = []
    for c in range(n):
        ok, msg = case(rng, *args, **kw)
        if not ok:
            bad.append(f"seed {seed} case {c}: {msg}")
    assert not bad, "\n".join(bad)


def test_fuzz_ml_nms_rotated_slice():
    """14 random ml_nms_rotated calls (1 ... 9 000 rows, 1 ... 40 labels, piles, weird label values) + 3 batched
    detector-style calls: keep lists / detections == oracle"""
    from scripts import fuzz_nms
    _run(fuzz_nms.nms_case, 2026, 14, sizes=(1, 7, 63, 64, 65, 500, 3000, 4097, 9000))
    _run(fuzz_nms.batched_case, 606, 3, ns=(300, 2000))


def test_fuzz_pyramid_alignconv_slice():
    """4 random pyramid AlignConv launches (ragged levels, batches): half-tile tail on / off bit-identical, level 0 == the
    per-level entry point"""
    from scripts import fuzz_alignconv
    g, wp = fuzz_alignconv.setup()
    _run(fuzz_alignconv.align_case, 5, 4, g, wp, max_side=100)


def test_fuzz_dcn_backward_slice():
    """8 random deform_conv backward calls (f32 / f16, ragged images, tame to wild offsets, scale, non-zero gradient buffers)
    against the oracle; the weight gradient twice, bit-identical"""
    from scripts import fuzz_dcn_backward
    _run(fuzz_dcn_backward.bwd_case, 7, 8, max_h=20, max_w=28)


def test_fuzz_dcn_f32_forward_slice():
    """6 random float32 forward calls: the three-bf16-plane kernel (default since round 6) against the oracle (1e-4) and against
    the f32 matrix instruction's kernel (1e-5 of the largest output); ragged images, 1 ... 10 channel chunks, 64 ... 320 out
    channels, tame to wild offsets, both storage orders, operands scaled over 20 binades"""
    from scripts import fuzz_dcn_f32
    _run(fuzz_dcn_f32.fwd_case, 31, 6, max_h=24, max_w=30)


def test_fuzz_assign_labels_and_nms_poly_slice():
    """the two section-8(f) ops rewritten in round 6 (list forms): 10 random assign_labels calls (1 ... 9 000 anchors, 1 ... 1 025
    gts -- both sides of the list form's limit --, exact ties, invalid anchors, both gt_max_assign_all settings, other thresholds)
    and 10 random nms_poly calls (piles of identical polygons, reversed winding, score ties, thresholds 0 ... 0.95) == oracle"""
    from scripts import fuzz_f_ops
    _run(fuzz_f_ops.assign_case, 11, 10)
    _run(fuzz_f_ops.poly_case, 12, 10)


def _small_stats():
    from s2anet_amd import _lib
    a, b = ctypes.c_int64(0), ctypes.c_int64(0)
    _lib.check(_lib.lib().s2a_nms_small_stats(ctypes.byref(a), ctypes.byref(b)))
    return a.value, b.value


def test_small_nms_while_other_streams_run_the_head():
    """k_nms_small's last-workgroup merge assumes nothing about residency: its bounded poll either completes or reports, and
    the general path answers then.  Here the call runs while two other streams keep the CUs busy with the head (512-thread
    workgroups at 150 KB of LDS each: a small-NMS workgroup may have to wait for a CU): keep == oracle every time, and the
    counters say which path answered."""
    import s2anet_amd as S
    from s2anet_amd.detector import build_synthetic_detector
    from s2anet_amd.pyramid import PyramidLayout
    dev = torch.device("cuda:0")
    m = build_synthetic_detector(device=dev)
    layout = PyramidLayout(8, [(128, 128), (64, 64), (32, 32), (16, 16), (8, 8)], (8, 16, 32, 64, 128))
    g = torch.Generator().manual_seed(3)
    feats = [torch.randn(layout.pixels, 256, generator=g).to(dev).half() for _ in range(2)]
    rng = np.random.default_rng(11)
    n = 5000
    cases = []
    for _ in range(6):
        d, s = rand_rboxes(rng, n), distinct_scores(rng, n)
        lab = rng.integers(0, 15, n).astype(np.float32)
        cases.append((d, s, lab, oracle.ml_nms_rotated(d, s, lab, 0.5)))
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    t0, f0 = _small_stats()
    with torch.no_grad():
        for x, st in zip(feats, streams):            # warm the per-stream pools
            with torch.cuda.stream(st):
                m.head.forward_pyramid(layout, x)
        torch.cuda.synchronize()
        for d, s, lab, want in cases:
            for rep in range(3):                     # ~ 10 ms of head launches queued on each side stream
                for x, st in zip(feats, streams):
                    with torch.cuda.stream(st):
                        m.head.forward_pyramid(layout, x)
            got = S.ml_nms_rotated(torch.from_numpy(d).to(dev), torch.from_numpy(s).to(dev), torch.from_numpy(lab).to(dev), 0.5)
            assert np.array_equal(got.cpu().numpy(), want)
        torch.cuda.synchronize()
    t1, f1 = _small_stats()
    assert (t1 - t0) + (f1 - f0) == len(cases), (t1 - t0, f1 - f0)      # every call went through the small-path gate
    print(f"small path answered {t1 - t0} of {len(cases)} calls under load, general path {f1 - f0}")
