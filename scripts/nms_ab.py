#!/usr/bin/env python3
"""ml_nms_rotated at 200 k rows x 15 labels: GPU time per call (HIP events over back-to-back calls), host enqueue time per
call, and the A/B switches of the prelude (S2A_NMS_FORK) -- one process per setting (the switches are read per call, but a
fresh process keeps the side streams' history out of the comparison)."""
import os, subprocess, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    from scripts.bench_ops import rboxes
    from s2anet_amd.rotated import ml_nms_rotated
    n = int(sys.argv[2])
    rng = np.random.default_rng(1234)
    d = torch.from_numpy(rboxes(rng, n)).cuda()
    s = torch.from_numpy(((rng.permutation(n) + 1) / (n + 1) * 0.95 + 0.05).astype(np.float32)).cuda()
    l = torch.from_numpy(rng.integers(0, 15, n).astype(np.float32)).cuda()
    for _ in range(5): ml_nms_rotated(d, s, l, 0.5)
    torch.cuda.synchronize()
    iters = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(iters): k = ml_nms_rotated(d, s, l, 0.5)
    e1.record(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    # one call alone (nothing queued behind it)
    lat = []
    for _ in range(5):
        torch.cuda.synchronize(); a = time.perf_counter(); ml_nms_rotated(d, s, l, 0.5); torch.cuda.synchronize(); lat.append(time.perf_counter() - a)
    print(json.dumps(dict(n=n, env={k: v for k, v in os.environ.items() if k.startswith("S2A_")}, gpu_ms_per_call=round(e0.elapsed_time(e1) / iters, 4),
                          host_enqueue_ms_per_call=round((t1 - t0) / iters * 1e3, 4), wall_ms_per_call=round((t2 - t0) / iters * 1e3, 4),
                          single_call_wall_ms=round(min(lat) * 1e3, 4), keep=int(k.numel()))))
    sys.exit(0)
for n in (200000, 20000):
    for env in ({}, {"S2A_NMS_FORK": "0"}):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(n)], env=dict(os.environ, **env), capture_output=True, text=True)
        print(r.stdout.strip() or r.stderr[-500:])
