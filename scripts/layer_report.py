"""per-layer timing of one detect() step (hipEvents around every leaf conv-like module), with the
HBM and MFMA floors of each layer beside the measured time.  GPU only.
usage: python scripts/layer_report.py [--iters 20]"""
import argparse, collections, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
from s2anet_amd.detector import build_synthetic_detector
from s2anet_amd.alignconv import AlignConv
from s2anet_amd.orn import ORConv2d

ap = argparse.ArgumentParser(); ap.add_argument("--iters", type=int, default=20); ap.add_argument("--batch", type=int, default=8)
a = ap.parse_args()
dev = torch.device("cuda:0")
m = build_synthetic_detector(device=dev)
g = torch.Generator().manual_seed(1234)
imgs = torch.randint(0, 256, (a.batch, 3, 1024, 1024), dtype=torch.uint8, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
rec = collections.OrderedDict()
state = {"on": False}

def pre(name):
    def f(mod, inp):
        if not state["on"]: return
        e = torch.cuda.Event(enable_timing=True); e.record(); rec.setdefault(name, {"ev": [], "mod": mod})["ev"].append([e, None])
        rec[name]["in"] = tuple(inp[0].shape)
    return f
def post(name):
    def f(mod, inp, out):
        if not state["on"]: return
        e = torch.cuda.Event(enable_timing=True); e.record(); rec[name]["ev"][-1][1] = e
        rec[name]["out"] = tuple(out.shape) if torch.is_tensor(out) else None
    return f
for name, mod in m.named_modules():
    if isinstance(mod, (nn.Conv2d, AlignConv, nn.MaxPool2d)) and not (isinstance(mod, nn.Conv2d) and name.endswith("deform_conv")):
        mod.register_forward_pre_hook(pre(name)); mod.register_forward_hook(post(name))
for _ in range(5): m.detect(imgs)
torch.cuda.synchronize()
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
state["on"] = True
t0.record()
for _ in range(a.iters): m.detect(imgs)
t1.record(); torch.cuda.synchronize()
tot = t0.elapsed_time(t1) / a.iters
rows = []
for name, r in rec.items():
    # modules called once per FPN level appear 5x per step: split per input shape
    by = collections.defaultdict(list)
    us = [s.elapsed_time(e) * 1e3 for s, e in r["ev"]]
    ncall = len(us) // a.iters
    for i, u in enumerate(us): by[i % ncall].append(u)
    mod = r["mod"]
    for k, v in by.items():
        rows.append((name + (f"#{k}" if ncall > 1 else ""), sum(v) / len(v), mod))
print(f"step {tot*1e3:.0f} us (with event overhead); sum of hooked layers {sum(r[1] for r in rows):.0f} us")
# shapes need a second pass for per-call entries; recompute from hooks on a single run
shapes = {}
def sh_hook(name):
    cnt = collections.Counter()
    def f(mod, inp, out):
        k = cnt[name]; cnt[name] += 1
        shapes[(name, k)] = (tuple(inp[0].shape), tuple(out.shape) if torch.is_tensor(out) else None)
    return f
hs = [mod.register_forward_hook(sh_hook(name)) for name, mod in m.named_modules() if name in rec]
state["on"] = False
m.detect(imgs); torch.cuda.synchronize()
out = []
for nm, us, mod in rows:
    base, _, k = nm.partition("#"); k = int(k) if k else 0
    i, o = shapes[(base, k)]
    B, C, H, W = i
    if o is None: continue
    _, O, Ho, Wo = o
    ks = mod.kernel_size if isinstance(mod, nn.Conv2d) else (3, 3)
    ks = (ks, ks) if isinstance(ks, int) else ks
    flops = 0 if isinstance(mod, nn.MaxPool2d) else 2.0 * B * Ho * Wo * O * C * ks[0] * ks[1]
    byt = 2.0 * (B * C * H * W + B * O * Ho * Wo) + (0 if isinstance(mod, nn.MaxPool2d) else 2.0 * O * C * ks[0] * ks[1])
    floor = max(flops / 2.5e15, byt / 8e12) * 1e6
    out.append(dict(layer=nm, type=type(mod).__name__, inp=list(i), out=list(o), k=ks[0], us=round(us, 1), floor_us=round(floor, 1),
                    frac=round(floor / us, 3), bound="mfma" if flops / 2.5e15 > byt / 8e12 else "hbm", tflops=round(flops / us / 1e6, 1),
                    GBs=round(byt / us / 1e3, 1)))
for r in out:
    print(f"{r['us']:8.1f} us floor {r['floor_us']:6.1f} ({r['bound']}) frac {r['frac']:.2f} {r['tflops']:7.1f} TF {r['GBs']:7.1f} GB/s  k{r['k']} {r['inp']} -> {r['out'][1]}  {r['layer']} [{r['type']}]")
json.dump(out, open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "layer_report.json"), "w"))
