"""phase stamps of the pyramid AlignConv launch (diagnostic build -DS2A_STAMP=1 [-DS2A_STAMP_W2=<first loader wave>]);
env S2A_DCN_MW8=1 stamps the eight-matrix-wave form"""
import sys, os, ctypes, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s2anet_amd import _lib, pyramid as P
from s2anet_amd.pyramid import PyramidLayout
from s2anet_amd.alignconv import pack_weight
dev = torch.device("cuda:0")
# S2A_STAMP_LAYOUT="B:HxW,HxW,..." picks another pyramid (e.g. "2:64x64": 64 tiles on a quarter of the CUs)
_lay = os.environ.get("S2A_STAMP_LAYOUT")
if _lay:
    _b, _lv = _lay.split(":")
    _sizes = [tuple(int(v) for v in t.split("x")) for t in _lv.split(",")]
    layout = PyramidLayout(int(_b), _sizes, tuple(8 << i for i in range(len(_sizes))))
else:
    layout = PyramidLayout(8, [(128, 128), (64, 64), (32, 32), (16, 16), (8, 8)], (8, 16, 32, 64, 128))
g = torch.Generator().manual_seed(0)
x = torch.randn(layout.pixels, 256, generator=g).to(dev).half()
w = (torch.randn(256, 256, 3, 3, generator=g) * 0.02).to(dev).half()
if 'zeros' in sys.argv: x, w = torch.zeros_like(x), torch.zeros_like(w)
pred = (torch.randn(layout.pixels, 64, generator=g) * 0.3).to(dev).half()
anc = P.fam_refine_anchors(layout, pred, 4.0)
wa = pack_weight(w, torch.float16)
import time
WARM_S = float(os.environ.get('S2A_WARM_S', '0.05'))          # seconds of back-to-back launches before the timed ones
t_w = time.time()
while time.time() - t_w < WARM_S:
    for _ in range(50): P.align_conv(layout, x, anc, wa, 256)
    torch.cuda.synchronize()
torch.cuda.synchronize()
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(100): P.align_conv(layout, x, anc, wa, 256)
t1.record(); torch.cuda.synchronize()
us = t0.elapsed_time(t1) / 100 * 1e3
buf = np.zeros(4096 * 16, np.uint64)
_lib.check(_lib.lib().s2a_debug_read_stamps(buf.ctypes.data_as(ctypes.c_void_p), buf.size))
st = buf.reshape(4096, 16).astype(np.int64)
st = st[st[:, 0] > 0][:1024]
c = st[:, :8]; l = st[:, 8:]
def d(a, i, j): return np.median(a[:, j] - a[:, i])
print("layout %s  " % (_lay or "bench pyramid"), end=""); print("data: %s, %.2f s of launches before the timed ones  " % ("zeros" if "zeros" in sys.argv else "dense", WARM_S), end=""); print("form: %s   us per launch (stamped build) %.1f" % ("MW 8" if os.environ.get("S2A_DCN_MW8") == "1" else "MW 4", us))
print("matrix wave 0: start->pre#1 %d | #1 wait %d | #1->#2 %d | main loop %d (per stage %d) | epilogue %d | total %d" % (d(c,0,1), d(c,1,2), d(c,2,3), d(c,3,4), d(c,3,4) / 36, d(c,4,5), d(c,0,5)))
print("loader:        start->pre#1 %d | #1 wait %d | columns 0 %d | #2 wait %d | main loop %d: own work per stage %d, barrier wait per stage %d" % (d(l,0,1), d(l,1,2), d(l,2,6), d(l,6,3), d(l,3,4), np.median(l[:, 7]) / 36, np.median(l[:, 5]) / 36))
rt = np.median(c[:, 7] - c[:, 6])                      # 100 MHz ticks between the start and end stamps of matrix wave 0
print("in-kernel clock of a tile: %.2f GHz (%d cycles in %.2f us)" % (d(c, 0, 5) / rt / 10.0, d(c, 0, 5), rt / 100.0))
