#!/usr/bin/env python3
"""Per-op timings of the hot-path kernels on one MI355X (BASELINE.json configs 1, 2, 5 and the
small ops).  HIP events on torch's current stream (the stream the C ABI launches on)."""
import argparse, json, math, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import s2anet_amd as S
from s2anet_amd.alignconv import align_conv_forward, pack_weight

dev = torch.device("cuda:0")

def timeit(fn, iters=20, warm=3, warm_s=0.25):
    # warm_s seconds of the same launches first: after host-side set-up the GPU comes back at a low clock
    t_w, n_w = time.time(), 0
    while n_w < warm or time.time() - t_w < warm_s:
        fn(); n_w += 1
        if n_w % 8 == 0: torch.cuda.synchronize()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3

def rboxes(rng, n, span=1024.0):
    b = np.empty((n, 5), np.float32)
    b[:, :2] = rng.uniform(0, span, (n, 2)); b[:, 2:4] = rng.uniform(4, 100, (n, 2)); b[:, 4] = rng.uniform(-np.pi/4, 3*np.pi/4, n)
    return b

def alignconv(batch, dtype, H=128, W=128, C=256, O=256, stride=8, sigma=0.5, jitter=4.0):
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(batch, C, H, W, generator=g).to(dev, dtype).contiguous(memory_format=torch.channels_last)
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    anc = torch.zeros(batch, H, W, 5)
    anc[..., 0] = xs * stride + 0.5*(stride-1) + torch.randn(batch, H, W, generator=g) * jitter
    anc[..., 1] = ys * stride + 0.5*(stride-1) + torch.randn(batch, H, W, generator=g) * jitter
    anc[..., 2:4] = 4 * stride * torch.exp(torch.randn(batch, H, W, 2, generator=g) * sigma)
    anc[..., 4] = (torch.rand(batch, H, W, generator=g) - 0.25) * math.pi
    anc = anc.to(dev)
    w = (torch.randn(O, C, 3, 3, generator=g) * 0.01).to(dev, dtype)
    wp = pack_weight(w, dtype)
    sec = timeit(lambda: align_conv_forward(x, anc, wp, stride, relu=True, packed=True, out_channels=O))
    flops = 2.0 * O * C * 9 * batch * H * W
    es = 2 if dtype == torch.float16 else 4
    byts = batch*H*W*(C+O)*es + O*C*9*es + batch*H*W*20
    peak = 2500.0 if es == 2 else 157.3
    return dict(op="alignconv_fused", sigma=sigma, dtype=str(dtype).split(".")[-1], batch=batch, hw=[H, W], us=round(sec*1e6, 1),
                tflops=round(flops/sec/1e12, 1), mfma_frac=round(flops/sec/1e12/peak, 4),
                alg_GBs=round(byts/sec/1e9, 1), hbm_frac=round(byts/sec/1e9/8000, 4))

def conv3(batch, hw, C=256, O=256, k=3, st=1):
    from s2anet_amd.fused import conv_f16, conv_pack_weight, bias_act_
    g = torch.Generator().manual_seed(1)
    x = torch.randn(batch, C, hw, hw, generator=g).to(dev).half().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(O, C, k, k, generator=g) * 0.02).to(dev).half()
    b = torch.randn(O, generator=g).to(dev).half()
    wp = conv_pack_weight(w)
    wcl = w.contiguous(memory_format=torch.channels_last)
    torch.backends.cudnn.benchmark = True
    pad = k // 2
    t_own = timeit(lambda: conv_f16(x, wp, b, O, k, st, True))
    t_mi = timeit(lambda: bias_act_(torch.nn.functional.conv2d(x, wcl, None, stride=st, padding=pad), b, None, True))
    ho = (hw - 1) // st + 1
    flops = 2.0 * O * C * k * k * batch * ho * ho
    byts = batch * (hw * hw * C + ho * ho * O) * 2
    return dict(op="conv%dx%d_f16+bias+relu" % (k, k), batch=batch, hw=hw, C=C, O=O, stride=st, own_us=round(t_own*1e6, 1),
                own_tflops=round(flops/t_own/1e12, 1), own_GBs=round(byts/t_own/1e9, 1),
                miopen_plus_epilogue_us=round(t_mi*1e6, 1))

def iou(n, m):
    rng = np.random.default_rng(1234)
    b1, b2 = torch.from_numpy(rboxes(rng, n)).to(dev), torch.from_numpy(rboxes(rng, m)).to(dev)
    out = S.box_iou_rotated(b1, b2)
    sec = timeit(lambda: S.box_iou_rotated(b1, b2), iters=10)
    byts = n*m*4 + (n+m)*20
    return dict(op="box_iou_rotated", n=n, m=m, us=round(sec*1e6, 1), Mpairs_s=round(n*m/sec/1e6, 1),
                nonzero_frac=round((out > 0).float().mean().item(), 4), alg_GBs=round(byts/sec/1e9, 1), hbm_frac=round(byts/sec/1e9/8000, 4))

def nms(n, nl=15):
    rng = np.random.default_rng(1234)
    d = torch.from_numpy(rboxes(rng, n)).to(dev)
    s = torch.from_numpy(((rng.permutation(n) + 1) / (n + 1) * 0.95 + 0.05).astype(np.float32)).to(dev)
    lab = torch.from_numpy(rng.integers(0, nl, n).astype(np.float32)).to(dev)
    k = S.ml_nms_rotated(d, s, lab, 0.5)
    sec = timeit(lambda: S.ml_nms_rotated(d, s, lab, 0.5), iters=20 if n <= 20000 else 5, warm=3)
    cnt = np.bincount(lab.cpu().numpy().astype(int))
    pairs = float((cnt.astype(np.float64) * (cnt - 1) / 2).sum())
    return dict(op="ml_nms_rotated", n=n, labels=nl, ms=round(sec*1e3, 3), keep=int(k.numel()),
                same_label_pairs=pairs, Gpairs_s=round(pairs/sec/1e9, 2))

def poly_ops(n_nms=20000, n_pairs=2000000):
    """chip-merge polygon NMS (py_cpu_nms_poly_fast) and pairwise polyiou on the GPU, with the oracle's CPU
    restatement (pinned bit-exact to the reference script / SWIG module) timed beside them on small samples"""
    import oracle
    rng = np.random.default_rng(77)
    res = []
    polys = oracle.rboxes_to_polys(rboxes(rng, n_nms, span=2048.0))
    dets = np.concatenate([polys, ((rng.permutation(n_nms) + 1.0) / (n_nms + 1.0))[:, None]], 1)
    d = torch.from_numpy(dets).to(dev)
    from s2anet_amd.rotated import nms_poly, polyiou_pairs
    keep = nms_poly(d, 0.3)
    sec = timeit(lambda: nms_poly(d, 0.3), iters=10)
    ns = 4000
    t0 = time.perf_counter(); kc = oracle.nms_poly(dets[:ns], 0.3); tc = time.perf_counter() - t0
    res.append(dict(op="nms_poly (chip merge)", n=n_nms, ms=round(sec * 1e3, 3), keep=int(keep.numel()),
                    cpu_port_s_at_n4000=round(tc, 3), cpu_keep_at_n4000=int(len(kc))))
    a = torch.from_numpy(oracle.rboxes_to_polys(rboxes(rng, n_pairs, span=300.0))).to(dev)
    b = torch.from_numpy(oracle.rboxes_to_polys(rboxes(rng, n_pairs, span=300.0))).to(dev)
    sec = timeit(lambda: polyiou_pairs(a, b), iters=10)
    res.append(dict(op="polyiou pairs (f64)", pairs=n_pairs, ms=round(sec * 1e3, 3), Mpairs_s=round(n_pairs / sec / 1e6, 1)))
    return res


def dcn_backward(batch=8, H=128, W=128, C=256, O=256, dtype=torch.float32, offsets="align"):
    """deform_conv backward (input + offset + weight gradients) at the P3 AlignConv shape.  offsets = "align": the offsets
    AlignConv really produces (SURVEY 8(d) config-2 anchors through get_offset); "wild": N(0, 2 px) on every tap"""
    from s2anet_amd.dcn import deform_conv_backward_input_cuda, deform_conv_backward_parameters_cuda
    g = torch.Generator().manual_seed(3)
    x = torch.randn(batch, C, H, W, generator=g).to(dev, dtype)
    if offsets == "align":
        from s2anet_amd.alignconv import AlignConv
        ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
        anc = torch.zeros(batch, H, W, 5)
        anc[..., 0] = xs * 8 + 3.5 + torch.randn(batch, H, W, generator=g) * 4
        anc[..., 1] = ys * 8 + 3.5 + torch.randn(batch, H, W, generator=g) * 4
        anc[..., 2:4] = 32 * torch.exp(torch.randn(batch, H, W, 2, generator=g) * 0.5)
        anc[..., 4] = (torch.rand(batch, H, W, generator=g) - 0.25) * math.pi
        from s2anet_amd.alignconv import align_offsets
        off = align_offsets(anc.to(dev).view(batch, -1, 5), (H, W), 8, 3).to(dtype).contiguous()
    else:
        off = (torch.randn(batch, 18, H, W, generator=g) * 2).to(dev, dtype)
    w = (torch.randn(O, C, 3, 3, generator=g) * 0.01).to(dev, dtype)
    go = torch.randn(batch, O, H, W, generator=g).to(dev, dtype)
    gi, goff, gw = torch.zeros_like(x), torch.zeros_like(off), torch.zeros_like(w)
    args = (3, 3, 1, 1, 1, 1, 1, 1, 1, 1)
    step = min(64, batch)
    t_in = timeit(lambda: deform_conv_backward_input_cuda(x, off, go, gi, goff, w, None, *args, step), iters=5, warm=2)
    t_w = timeit(lambda: deform_conv_backward_parameters_cuda(x, off, go, gw, None, None, *args, 1.0, step), iters=5, warm=2)
    from s2anet_amd.dcn import _fused_backward
    t_both = timeit(lambda: _fused_backward(x, off, w, go), iters=5, warm=2)     # what DeformConvFunction.backward runs
    flops = 2.0 * O * C * 9 * batch * H * W
    return dict(op="deform_conv backward", dtype=str(dtype).split(".")[-1], batch=batch, hw=[H, W], offsets=offsets,
                input_offset_ms=round(t_in * 1e3, 3), weight_ms=round(t_w * 1e3, 3), both_one_call_ms=round(t_both * 1e3, 3),
                gemm_tflops_each=round(flops / 1e12, 3),
                note="two fused kernels per call, no columns tensor: MFMA column gradient consumed in LDS; weight gradient contracted over the positions (f16: transposing LDS reads; f32: v_mfma_f32_16x16x4 / 32x32x2, 155 GFLOP each = 0.99 ms at the f32 MFMA peak)")


def assign(n_gt=300):
    """fused assign_labels on the 21 824 grid anchors of one chip vs box_iou_rotated + the stock max/compare ops"""
    import oracle
    from s2anet_amd.rotated import assign_labels
    rng = np.random.default_rng(5)
    a = np.concatenate([oracle.grid_anchors(1024 // s, 1024 // s, s).reshape(-1, 5) for s in (8, 16, 32, 64, 128)]).astype(np.float32)
    a[:, 4] = rng.uniform(-0.7, 2.3, a.shape[0])
    gt = rboxes(rng, n_gt)
    A, G = torch.from_numpy(a).to(dev), torch.from_numpy(gt).to(dev)
    t_f = timeit(lambda: assign_labels(A, G))
    def unfused():
        iou = S.box_iou_rotated(A, G)
        mx, am = iou.max(1)
        gm, ga = iou.max(0)
        return (iou == gm[None]).any(1), mx, am
    t_u = timeit(unfused)
    return dict(op="assign_labels", anchors=int(a.shape[0]), gts=n_gt, fused_us=round(t_f * 1e6, 1),
                iou_matrix_plus_maxes_us=round(t_u * 1e6, 1))


def cpu_baselines():
    """the reference's own CPU ops (oracle/_ref, built from /root/reference unmodified) timed on this
    box's host cores, single thread as the reference loops are serial; bounded samples"""
    import oracle
    from oracle import ref
    torch.set_num_threads(1)
    rng = np.random.default_rng(1234)
    out = []
    f = ref.box_iou_rotated()
    kind = "reference" if f is not None else "port"
    b1, b2 = rboxes(rng, 1500), rboxes(rng, 1500)
    t = time.time()
    if f is not None: f(torch.from_numpy(b1), torch.from_numpy(b2))
    else: oracle.box_iou_rotated(b1, b2, sort_mode=oracle.SORT_CPU)
    dt = time.time() - t
    out.append(dict(op="cpu box_iou_rotated", kind=kind, n=1500, m=1500, s=round(dt, 3), Mpairs_s=round(2.25 / dt, 3), cores=1))
    fp = ref.polyiou()
    P, Q = oracle.rboxes_to_polys(rboxes(rng, 20000, span=300)), oracle.rboxes_to_polys(rboxes(rng, 20000, span=300))
    t = time.time()
    if fp is not None:
        for i in range(20000): fp(P[i], Q[i])
        pk = "reference"
    else:
        oracle.polyiou(P, Q); pk = "port"
    dt = time.time() - t
    out.append(dict(op="cpu polyiou", kind=pk, pairs=20000, s=round(dt, 3), Kpairs_s=round(20 / dt, 1), cores=1))
    fn = ref.ml_nms_rotated()
    for n in (2000, 5000):
        d = rboxes(rng, n); sc = ((rng.permutation(n) + 1) / (n + 1)).astype(np.float32); lab = rng.integers(0, 15, n).astype(np.float32)
        t = time.time()
        if fn is not None: k = fn(torch.from_numpy(d), torch.from_numpy(sc), torch.from_numpy(lab), 0.5)
        else: k = oracle.ml_nms_rotated(d, sc, lab, 0.5, rule=oracle.RULE_GE, sort_mode=oracle.SORT_CPU)
        dt = time.time() - t
        out.append(dict(op="cpu ml_nms_rotated", kind="reference" if fn is not None else "port", n=n, s=round(dt, 3), keep=int(len(k)), cores=1))
    x = rng.standard_normal((1, 256, 32, 32)).astype(np.float32); w = (rng.standard_normal((256, 256, 3, 3)) * 0.01).astype(np.float32)
    off = (rng.standard_normal((1, 18, 32, 32))).astype(np.float32)
    ncores = min(os.cpu_count() or 1, 16)
    t = time.time(); oracle.deform_conv_forward(x, off, w); dt = time.time() - t
    out.append(dict(op="cpu deform_conv forward (oracle port, OpenMP; the reference has no CPU path)", kind="port",
                    shape=[1, 256, 32, 32], s=round(dt, 3), GFLOPs=round(2 * 256 * 2304 * 1024 / dt / 1e9, 2), cores=ncores))
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser(); ap.add_argument("--which", default="all"); a = ap.parse_args()
    res = []
    if a.which in ("all", "cpu"):
        res += cpu_baselines()
    if a.which in ("all", "align"):
        for b, dt in ((8, torch.float16), (1, torch.float16), (8, torch.float32), (1, torch.float32)):
            res.append(alignconv(b, dt))
        res.append(alignconv(8, torch.float16, 64, 64, stride=16))
    if a.which in ("all", "conv"):
        for hw in (128, 64, 32):
            res.append(conv3(8, hw))
    if a.which == "convbb":
        for (c, o, hw) in ((64, 64, 256), (128, 128, 128), (256, 256, 64), (512, 512, 32), (256, 256, 16), (256, 256, 8)):
            res.append(conv3(8, hw, c, o))
        for (c, o, hw, st) in ((64, 64, 256, 1), (64, 256, 256, 1), (256, 64, 256, 1), (256, 128, 256, 1), (128, 512, 128, 1),
                               (512, 128, 128, 1), (256, 512, 256, 2), (512, 256, 128, 1), (256, 1024, 64, 1), (1024, 256, 64, 1),
                               (1024, 512, 64, 1), (512, 2048, 32, 1), (2048, 512, 32, 1), (1024, 2048, 64, 2), (2048, 256, 32, 1)):
            res.append(conv3(8, hw, c, o, 1, st))
    if a.which == "alignconv1":
        res.append(alignconv(1, torch.float16))
    if a.which == "align8":
        res.append(alignconv(8, torch.float16))
    if a.which == "align8s":
        for sg in (0.0, 0.25, 0.5, 0.75):
            res.append(alignconv(8, torch.float16, sigma=sg))
    if a.which in ("all", "iou"):
        res.append(iou(10000, 10000)); res.append(iou(21824, 128))
    if a.which in ("all", "assign"):
        res.append(assign(300)); res.append(assign(32))
    if a.which in ("all", "bwd"):
        res.append(dcn_backward(8, dtype=torch.float32)); res.append(dcn_backward(8, dtype=torch.float16))
        res.append(dcn_backward(8, dtype=torch.float16, offsets="wild"))
    if a.which == "bwd32":
        res.append(dcn_backward(8, dtype=torch.float32))
    if a.which == "bwd16":
        res.append(dcn_backward(8, dtype=torch.float16))
    if a.which in ("all", "poly"):
        res += poly_ops()
    if a.which == "iou10k":
        print(json.dumps(iou(10000, 10000)))
    if a.which == "nms200k":
        res.append(nms(200000))
    if a.which in ("all", "nms"):
        for n in (5000, 20000, 80160, 200000):
            res.append(nms(n))
    for r in res: print(json.dumps(r))
