"""phase stamps of a full-width 1x1 layer (diagnostic build -DS2A_STAMP=1 -DS2A_STAMP_W2=3 only): scripts/stamp_conv1_run.sh"""
import os, sys, ctypes, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s2anet_amd import _lib
from s2anet_amd.fused import conv_f16, conv_pack_weight
g = torch.Generator().manual_seed(1)
for (B, C, H, W, O, with_res) in ((8, 1024, 64, 64, 256, False), (8, 256, 64, 64, 1024, True)):
    x = torch.relu(torch.randn(B, C, H, W, generator=g)).to("cuda").half().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(O, C, 1, 1, generator=g) * 0.03).to("cuda").half()
    b = torch.randn(O, generator=g).to("cuda").half()
    r = torch.randn(B, O, H, W, generator=g).to("cuda").half().contiguous(memory_format=torch.channels_last) if with_res else None
    wp = conv_pack_weight(w)
    out = torch.empty((B, O, H, W), dtype=torch.float16, device="cuda", memory_format=torch.channels_last)
    os.environ["S2A_CONV1_HALF"] = "0"
    for _ in range(20): conv_f16(x, wp, b, O, 1, 1, True, r, out=out)
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(50): conv_f16(x, wp, b, O, 1, 1, True, r, out=out)
    t1.record(); torch.cuda.synchronize()
    us = t0.elapsed_time(t1) / 50 * 1e3
    buf = np.zeros(4096 * 16, np.uint64)
    _lib.check(_lib.lib().s2a_debug_read_stamps(buf.ctypes.data_as(ctypes.c_void_p), buf.size))
    st = buf.reshape(4096, 16).astype(np.int64)
    nwg = B * H * W // 128 * (O // 256)
    st = st[:min(nwg, 4096)]
    st = st[st[:, 0] > 0]
    print("layer %d -> %d @%dx%d res=%s: %.1f us per launch (stamped build), %d workgroups, %d chunks" % (C, O, H, W, with_res, us, nwg, C // 64))
    for name, a in (("wave 0", st[:, :8]), ("wave 3", st[:, 8:])):
        d = lambda i, j: np.median(a[:, j] - a[:, i])
        print("  %s: start -> first barrier %d | barrier wait %d | chunk loop %d (compute %d, barrier %d) | acc -> LDS + barrier %d | residual / stores %d | total %d"
              % (name, d(0, 1), d(1, 2), d(2, 3), np.median(a[:, 6]), np.median(a[:, 7]), d(3, 4), d(4, 5), d(0, 5)))
    tot = np.median(st[:, 5] - st[:, 0])
    rounds = -(-nwg // 512)      # two workgroups per CU
    print("  workgroup lifetime %d cycles; %d round(s) of workgroups in %.1f us -> >= %.2f GHz in-kernel clock" % (tot, rounds, us, tot * rounds / us / 1e3))
