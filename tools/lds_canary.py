"""Does an LDS-DMA kernel ever write outside its own LDS allocation?  (one of the two cheap experiments on the
pooled-pass fault.)  An LDS canary (rpnet_debug_lds_canary: blocks that fill a few KB of LDS with a pattern and re-verify it for ~400 us)
runs on its own stream beside each LDS-DMA kernel form of the training step — the 3x3 forward / input-gradient patch kernel (152 - 156 KB
of LDS per block), the collapsed up_conv forms, the weight-gradient ring (144 KB) and the up_conv weight gradient — with an allocation
small enough to share their CUs (4 KB).  A changed word = a DMA write outside its workgroup's allocation.
Usage: python tools/lds_canary.py [rounds]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rpnet_amd.functional as RF  # noqa: E402
from rpnet_amd.functional import PackedWeight, _desc, split_f16  # noqa: E402
from rpnet_amd.hip import call, ptr, query  # noqa: E402

dev = "cuda:0"
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
g = torch.Generator().manual_seed(11)
side = torch.cuda.Stream()
sc = torch.tensor([2.0 ** -13], device=dev)


def planes(*shape):
    return split_f16(torch.randn(*shape, generator=g).to(dev), sc, want_scale=False, planes=2)[0]


def conv_launch(N, H, W, ci, co, ups=False):
    x = planes(N, H // (2 if ups else 1), W // (2 if ups else 1), ci)
    w = torch.randn(co, ci, 3, 3, generator=g).to(dev) * 0.05
    pw = PackedWeight(w)
    y = torch.empty(N, H, W, co, device=dev)
    if ups:
        pk = pw.up4_packs(2)
        d = _desc(x, None, pk[0], None, None, 0, y, None, N, H, W, 9, 1)
        d.split_planes = 2
        d.acc_scale_col, d.acc_scale_x = pk[2].data_ptr(), sc.data_ptr()
        d._keep = (x, pk, y)
        return lambda: call("rpnet_conv_up4", C.byref(d), 1)
    pk = pw.split_packs(2)
    d = _desc(x, None, pk[0], None, None, 0, y, None, N, H, W, 9, 0)
    d.split_planes = 2
    d.acc_scale_col, d.acc_scale_x = pk[2].data_ptr(), sc.data_ptr()
    d._keep = (x, pk, y)
    return lambda: call("rpnet_conv_fwd", C.byref(d))


def wgrad_launch(N, H, W, ci, co, up4=False):
    x = planes(N, H // (2 if up4 else 1), W // (2 if up4 else 1), ci)
    dy = planes(N, H, W, co)
    dw = torch.empty(co, ci, 3, 3, device=dev)
    d = _desc(x, None, None, None, None, 0, None, None, N, H, W, 9, 1 if up4 else 0, co_split=(co, 0), wgrad=True)
    d.split_planes = 2
    d.acc_scale_x, d.acc_scale_dy = ptr(sc), ptr(sc)
    if up4:
        wb = query("rpnet_conv_wgrad_up4_workspace_bytes", N, H, W, ci, co)
        ws = torch.empty(wb // 4 + 4, device=dev)
        d._keep = (x, dy, dw, ws)
        return lambda: call("rpnet_conv_wgrad_up4", C.byref(d), ptr(dy), ptr(dw), ptr(ws), wb)
    wb = query("rpnet_conv_wgrad_workspace_bytes", N, H, W, ci, co, 9)
    ws = torch.empty(wb // 4 + 4, device=dev)
    d._keep = (x, dy, dw, ws)
    return lambda: call("rpnet_conv_wgrad", C.byref(d), ptr(dy), ptr(dw), ci, 0, ci, ci, ptr(ws), wb)


FORMS = [("conv 3x3 256->256 @ 16x64x64 (patch kernel, 152 KB)", conv_launch(16, 64, 64, 256, 256)),
         ("conv 3x3 128->64 @ 16x128x128 (64-column form)", conv_launch(16, 128, 128, 128, 64)),
         ("up_conv forward 512->256 @ 16x64x64 (collapsed)", conv_launch(16, 64, 64, 512, 256, ups=True)),
         ("weight gradient ring 256->256 @ 16x64x64 (144 KB)", wgrad_launch(16, 64, 64, 256, 256)),
         ("weight gradient row-major 1024->1024 @ 16x16x16", wgrad_launch(16, 16, 16, 1024, 1024)),
         ("up_conv weight gradient 512->256 @ 16x64x64", wgrad_launch(16, 64, 64, 512, 256, up4=True))]
total_bad = 0
for name, fn in FORMS:
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    res = torch.zeros(3, device=dev, dtype=torch.int32)
    for _ in range(rounds):
        with torch.cuda.stream(side):
            for _ in range(6):
                fn()
        # 4 KB of LDS and few registers: the canary's blocks fit beside any of the forms above; 1024 blocks x ~400 us
        call("rpnet_debug_lds_canary", 1024, 4096, 40000, ptr(res))
        torch.cuda.synchronize()
    bad, blocks, sweeps = (int(v) for v in res.cpu())
    total_bad += bad
    print(f"{name}: {rounds} rounds, canary blocks {blocks}, verification sweeps {sweeps}, changed words {bad}", flush=True)
print("LDS canary:", "NO word changed — no LDS-DMA write outside its workgroup's allocation was seen" if total_bad == 0 else f"{total_bad} WORDS CHANGED")
