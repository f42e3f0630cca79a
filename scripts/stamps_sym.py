"""phase stamps of the symmetric AlignConv kernel (k_dcn_sym; -DS2A_STAMP=1 build, scripts/stamp_sym_run.sh): the
pyramid-packed launch of the bench (batch 8, five levels), dense random data"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, '.')
os.environ["S2A_DCN_SYM"] = "1"
from s2anet_amd import _lib
from s2anet_amd import pyramid as P
from s2anet_amd.pyramid import PyramidLayout
from s2anet_amd.alignconv import pack_weight
dev = torch.device("cuda:0")
layout = PyramidLayout(8, [(128, 128), (64, 64), (32, 32), (16, 16), (8, 8)], (8, 16, 32, 64, 128))
g = torch.Generator().manual_seed(0)
x = torch.randn(layout.pixels, 256, generator=g).to(dev).half()
w = (torch.randn(256, 256, 3, 3, generator=g) * 0.02).to(dev).half()
pred = (torch.randn(layout.pixels, 64, generator=g) * 0.3).to(dev).half()
anc = P.fam_refine_anchors(layout, pred, 4.0)
wa = pack_weight(w, torch.float16)
for _ in range(5):
    P.align_conv(layout, x, anc, wa, 256)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30):
    P.align_conv(layout, x, anc, wa, 256)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 30 * 1e3
buf = np.zeros(4096 * 16, np.uint64)
_lib.check(_lib.lib().s2a_debug_read_stamps(buf.ctypes.data_as(ctypes.c_void_p), buf.size))
st = buf.reshape(4096, 16).astype(np.int64)
st = st[st[:, 0] > 0][:512]            # the P3 tiles (the first 512 workgroups)
def d(a, i, j): return np.median(a[:, j] - a[:, i])
for name, a in (("wave 0 (blends first)", st[:, :8]), ("wave 4 (contracts first)", st[:, 8:])):
    loop = d(a, 3, 4)
    print("%s: start->table barrier %d | wait %d | blend0+barrier %d | main loop %d; per interval %d = blend %d + mma %d + barrier wait %d + head (piece, filter requests, stamps) %d"
          % (name, d(a, 0, 1), d(a, 1, 2), d(a, 2, 3), loop, loop / 72, np.median(a[:, 6]) / 72, np.median(a[:, 5]) / 72, np.median(a[:, 7]) / 72,
             (loop - np.median(a[:, 6]) - np.median(a[:, 7]) - np.median(a[:, 5])) / 72))
tot = np.median(st[:, 4] - st[:, 0]) + 5000
print("us per launch (stamped build) %.1f; workgroup lifetime %d cycles; 688 tiles on 256 CUs = 3 rounds -> %.2f GHz in-kernel clock"
      % (us, tot, tot * 3 / us / 1e3))
