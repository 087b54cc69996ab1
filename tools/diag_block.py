"""Diagnostic: run-to-run determinism of ONE conv_block (two conv + BatchNorm + ReLU layers, optionally with the fused
2x2 max-pool) forward + backward on fixed inputs.   python tools/diag_block.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rpnet_amd.functional as RF  # noqa: E402
import rpnet_amd.modules as RM  # noqa: E402

dev = "cuda:0"
RM._F16_MIN_PIXELS = 0
RF.set_conv_math(os.environ.get("DIAG_MATH", "f16x2"))
RF.set_f16_active(True)


def one(blk, x, pool, seed_dz):
    cache = RF.WeightCache()
    for p in blk.parameters():
        p.grad = None
    xin = x.clone().requires_grad_(x.shape[-1] > 1)
    out = blk.forward_nhwc(RF.Operand(xin, scale=torch.full((1,), 2.0 ** -10, device=dev)) if x.shape[-1] > 1 else xin, cache,
                           out_split=True, pool=pool)
    z = out.x if not out.planes_only else None
    g = torch.Generator(device=dev).manual_seed(seed_dz)
    if z is None:
        raise SystemExit("planes-only output")
    dz = torch.randn(z.shape, device=dev, generator=g)
    z.backward(dz)
    torch.cuda.synchronize()
    return [p.grad.clone() for p in blk.parameters()] + ([xin.grad.clone()] if xin.grad is not None else [])


for (cin, cout, n, hw, pool) in ((1, 64, 4, 128, True), (1, 64, 8, 128, True), (64, 128, 4, 64, True), (64, 128, 8, 64, True),
                                 (128, 256, 8, 32, False), (64, 64, 4, 128, True)):
    torch.manual_seed(5)
    blk = RM.conv_block(cin, cout, "BatchNorm2d").to(dev).train()
    x = torch.randn(n, hw, hw, cin, device=dev).clamp_(-4, 4)
    ref = one(blk, x, pool, 7)
    bad = 0
    worst = 0.0
    for r in range(12):
        got = one(blk, x, pool, 7)
        for a, b in zip(ref, got):
            if not torch.equal(a, b):
                bad += 1
                worst = max(worst, float((a - b).abs().max() / (a.abs().max() + 1e-30)))
    print(f"conv_block {cin}->{cout} N={n} {hw}^2 pool={pool}: {bad} differing tensors in 12 repeats (worst {worst:.1e})", flush=True)
