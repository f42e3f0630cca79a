import torch, time, sys
sys.path.insert(0,'.')
from s2anet_amd.fused import bias_act_
import torch.nn.functional as F
torch.backends.cudnn.benchmark = True
dev='cuda'
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
for (cin,cout,k,hw,stride) in [(256,256,3,128,1),(64,64,3,256,1),(64,256,1,256,1),(256,64,1,256,1),(512,128,1,128,1),(128,128,3,128,1),(3,64,7,1024,2)]:
    x = torch.randn(8,cin,hw,hw,device=dev,dtype=torch.half).contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout,cin,k,k,device=dev,dtype=torch.half).contiguous(memory_format=torch.channels_last)
    b = torch.randn(cout,device=dev,dtype=torch.half)
    pad = k//2
    t_conv = timeit(lambda: F.conv2d(x,w,None,stride,pad))
    t_ours = timeit(lambda: bias_act_(F.conv2d(x,w,None,stride,pad), b, None, True))
    t_stock = timeit(lambda: F.relu(F.conv2d(x,w,b,stride,pad)))
    try:
        t_fused = timeit(lambda: torch.miopen_convolution_relu(x,w,b,[stride,stride],[pad,pad],[1,1],1))
        y1 = torch.miopen_convolution_relu(x,w,b,[stride,stride],[pad,pad],[1,1],1); y2 = F.relu(F.conv2d(x,w,b,stride,pad))
        err = (y1.float()-y2.float()).abs().max().item()
    except Exception as e:
        t_fused = float('nan'); err = str(e)[:80]
    print(f"cin{cin} cout{cout} k{k} hw{hw}: conv {t_conv:.1f} conv+ours {t_ours:.1f} stock {t_stock:.1f} miopen_fused {t_fused:.1f} err {err}")
