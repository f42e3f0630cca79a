import sys, ctypes, numpy as np, torch
sys.path.insert(0, '.')
from s2anet_amd import _lib
import scripts.bench_ops as bo
BATCH = int(sys.argv[1]) if len(sys.argv) > 1 else 8
HW_ = int(sys.argv[2]) if len(sys.argv) > 2 else 128
r = bo.alignconv(BATCH, torch.float16, H=HW_, W=HW_)
print(r)
torch.cuda.synchronize()
buf = np.zeros(4096 * 16, np.uint64)
_lib.check(_lib.lib().s2a_debug_read_stamps(buf.ctypes.data_as(ctypes.c_void_p), buf.size))
st = buf.reshape(4096, 16).astype(np.int64)
st = st[st[:, 0] > 0][:1024]
c = st[:, :8]; l = st[:, 8:]
def d(a, i, j): return np.median(a[:, j] - a[:, i])
print("consumer: start->pre#1 %d | #1 wait %d | #1->#2 %d | main loop %d | epilogue %d | total %d" % (d(c,0,1), d(c,1,2), d(c,2,3), d(c,3,4), d(c,4,5), d(c,0,5)))
rt = np.median(c[:, 7] - c[:, 6])
print("in-kernel clock of a tile: %.2f GHz" % (np.median(c[:, 5] - c[:, 0]) / rt / 10.0))
print("loader:   start->pre#1 %d | #1 wait %d | produce0 %d | #2 wait %d | main loop %d" % (d(l,0,1), d(l,1,2), d(l,2,6), d(l,6,3), d(l,3,4)))
tot = np.median(c[:, 5] - c[:, 0])
rounds = -(-BATCH * 128 * 128 // 128 // 256) if BATCH > 1 else 1      # 8 x 16 tiles of the P3 level at batch 8 on 256 CUs, one workgroup per CU
print("workgroup lifetime %d cycles (36 stages of %d; 1024 of them MFMA); %d rounds of tiles in %.1f us -> %.2f GHz in-kernel clock"
      % (tot, d(c, 3, 4) / 36, rounds, r["us"], tot * rounds / r["us"] / 1e3))
