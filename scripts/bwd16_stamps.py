#!/usr/bin/env python3
"""phase cycles of k_dcn_bwd_weight (f16; a -DS2A_MEASURE -DS2A_MEASURE_F16W build of dcn_bwd_ops.o: scripts/bwd32_stamps.sh f16)"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2anet_amd import _lib
from s2anet_amd.dcn import deform_conv_backward_parameters_cuda
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
B, C, H, W, O = 8, 256, 128, 128, 256
x = torch.randn(B, C, H, W, generator=g).to(dev).half()
off = (torch.randn(B, 18, H, W, generator=g) * 0.5).to(dev).half()
go = torch.randn(B, O, H, W, generator=g).to(dev).half()
gw = torch.zeros(O, C, 3, 3, device=dev)
args = (3, 3, 1, 1, 1, 1, 1, 1, 1, 1)
L = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 16)()
for _ in range(3):
    deform_conv_backward_parameters_cuda(x, off, go, gw, None, None, *args, 1.0, B)
torch.cuda.synchronize()
L.s2a_debug_bwd_stamps(buf)
n = 10
t0 = time.perf_counter()
for _ in range(n):
    deform_conv_backward_parameters_cuda(x, off, go, gw, None, None, *args, 1.0, B)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
L.s2a_debug_bwd_stamps(buf)
v = list(buf)
tiles, wgs = v[4] / n, v[5] / n
print("call %.3f ms; %d workgroups, %.1f tiles each" % (ms, wgs, tiles / wgs))
for name, c in (("loop-top barrier", v[0]), ("land + table", v[1]), ("requests of the next tile", v[2]), ("blend", v[3]), ("MFMA", v[7])):
    print("  %-28s %8.0f cycles per tile" % (name, c / n / tiles))
print("  whole kernel %.0f cycles per workgroup = %.0f per tile" % (v[6] / n / wgs, v[6] / n / tiles))
