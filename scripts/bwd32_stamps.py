#!/usr/bin/env python3
"""phase cycles of k_dcn_bwd_weight_f32 (an -DS2A_MEASURE build of dcn_bwd_ops.o; scripts/bwd32_stamps.sh)"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2anet_amd import _lib
from s2anet_amd.dcn import deform_conv_backward_parameters_cuda

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
B, C, H, W, O = 8, 256, 128, 128, 256
x = torch.randn(B, C, H, W, generator=g).to(dev)
off = (torch.randn(B, 18, H, W, generator=g) * 0.5).to(dev)
go = torch.randn(B, O, H, W, generator=g).to(dev)
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
for name, c in (("MFMA halves (wave 0)", v[0]), ("B1 wait (wave 0)", v[1]), ("B2 wait (wave 0)", v[2]),
                ("MFMA halves (wave 4)", v[11]), ("B1 wait (wave 4)", v[12]), ("B2 wait (wave 4)", v[13]),
                ("land + table (wave 8)", v[3]), ("  of it: wait for the loads", v[8]), ("  of it: land", v[9]),
                ("blend + requests (wave 8)", v[7]), ("  of it: requests", v[10])):
    print("  %-26s %8.0f cycles per tile" % (name, c / n / tiles))
print("  whole kernel %.0f cycles per workgroup = %.0f per tile -> %.2f GHz if the kernel is %.3f ms" % (
    v[6] / n / wgs, v[6] / n / tiles, v[6] / n / wgs / (ms * 1e-3) / 1e9, ms))
