"""Micro-benchmark of the local correlation kernels at the CRE shape (B=8, 64x64, C=256, r=5)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpnet_amd import functional as RF
B, h, w, C, r = 8, 64, 64, 256, 5
f1 = torch.randn(B, h, w, C, device="cuda").requires_grad_(True); f2 = torch.randn(B, h, w, C, device="cuda").requires_grad_(True)
go = torch.randn(B, h, w, 128, device="cuda")
def run(fn, n=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
tf = run(lambda: RF.LocalCorr.apply(f1, f2, r)[0])
out, _ = RF.LocalCorr.apply(f1, f2, r)
tb = run(lambda: torch.autograd.grad(out, [f1, f2], go, retain_graph=True))
print(f"corr fwd {tf*1e3:.1f} us  ({2*B*h*w*121*C/tf/1e9:.1f} TF)   bwd {tb*1e3:.1f} us")
