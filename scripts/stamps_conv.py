"""phase stamps of the pyramid-packed 3x3 tower kernel (diagnostic build -DS2A_STAMP=1 only): scripts/stamp_conv_run.sh"""
import os, sys, ctypes, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s2anet_amd import _lib
from s2anet_amd import pyramid as P
from s2anet_amd.pyramid import PyramidLayout
from s2anet_amd.fused import conv_pack_weight
dev = torch.device("cuda:0")
layout = PyramidLayout(8, [(128, 128), (64, 64), (32, 32), (16, 16), (8, 8)], (8, 16, 32, 64, 128))
g = torch.Generator().manual_seed(0)
x = torch.randn(layout.pixels, 256, generator=g).to(dev).half()
w = (torch.randn(256, 256, 3, 3, generator=g) * 0.02).to(dev).half()
b = torch.randn(256, generator=g).to(dev).half()
wp = conv_pack_weight(w)
for _ in range(20): P.conv3x3(layout, x, wp, b, 256, True)
torch.cuda.synchronize()
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(30): P.conv3x3(layout, x, wp, b, 256, True)
t1.record(); torch.cuda.synchronize()
print("us per launch (stamped build)", t0.elapsed_time(t1) / 30 * 1e3)
buf = np.zeros(4096 * 16, np.uint64)
_lib.check(_lib.lib().s2a_debug_read_stamps(buf.ctypes.data_as(ctypes.c_void_p), buf.size))
st = buf.reshape(4096, 16).astype(np.int64)
st = st[st[:, 0] > 0][:688]
for name, a in (("wave 0", st[:, :8]), ("wave 4", st[:, 8:])):
    d = lambda i, j: np.median(a[:, j] - a[:, i])
    print("%s: start->pre-barrier %d | first barrier wait %d | main loop %d (compute %d, barrier %d) | acc->LDS + barrier %d | stores %d | total %d"
          % (name, d(0, 1), d(1, 2), d(2, 3), np.median(a[:, 6]), np.median(a[:, 7]), d(3, 4), d(4, 5), d(0, 5)))
tot = st[:, 5] - st[:, 0]
print("workgroup total cycles: median %d  p10 %d  p90 %d; first start -> last end %d" % (np.median(tot), np.percentile(tot, 10), np.percentile(tot, 90), st[:, [5, 13]].max() - st[:, [0, 8]].min()))
order = np.argsort(st[:, 0]); s0 = st[order, 0] - st[:, 0].min()
print("start times (cycles) of workgroups 0, 255, 256, 511, 512, 687 in start order:", [int(s0[i]) for i in (0, 255, 256, 511, 512, min(687, len(s0) - 1))])
