"""Micro-benchmark of rpnet_local_corr_split_bwd on two fp16 planes (the arithmetic of the training step) at the CRE shapes:
configs[1] (B = 8, 64 x 64, C = 256) and configs[4] (B = 4, 128 x 128).  RPNET_CORR_BWD_XCD=1: XCD-aware tile order."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpnet_amd import functional as RF
from rpnet_amd.hip import call, ptr, query
dev = "cuda:0"
for (B, h, w, C) in ((8, 64, 64, 256), (4, 128, 128, 256)):
    g = torch.Generator().manual_seed(3)
    f1 = torch.relu(torch.randn(B, h, w, C, generator=g)).to(dev); f2 = torch.relu(torch.randn(B, h, w, C, generator=g)).to(dev)
    sc = torch.tensor([2.0 ** -12], device=dev)
    p1 = RF.split_f16(f1, sc, want_scale=False, planes=2)[0]; p2 = RF.split_f16(f2, sc, want_scale=False, planes=2)[0]
    dcorr = torch.randn(B, h, w, 128, generator=g).to(dev); dcorr[..., 121:] = 0
    df1, df2 = torch.empty_like(f1), torch.empty_like(f2)
    add = torch.randn(B, h, w, C, generator=g).to(dev)
    wb = query("rpnet_local_corr_bwd_workspace_bytes", B, h, w, 128)
    ws = torch.empty(wb // 4 + 4, device=dev)
    fn = lambda: call("rpnet_local_corr_split_bwd", ptr(p1), ptr(p2), ptr(dcorr), ptr(df1), ptr(df2), B, h, w, C, 5, 128, 2, ptr(sc), ptr(sc), ptr(add), ptr(ws), wb)
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): fn()
    b.record(); torch.cuda.synchronize()
    print(f"corr bwd (both passes + transpose) B={B} {h}x{w} C={C}: {a.elapsed_time(b) / 20 * 1e3:.1f} us   checksum {float(df1.double().sum()):.6e} {float(df2.double().sum()):.6e}", flush=True)
