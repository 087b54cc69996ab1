"""Stand-alone reproducer of round 4's pooled-pass fault (profiles/r04_pool_apply_fault.txt): ONE
LDS-DMA weight-gradient launch (conv_wgrad9_dma_kernel, rpnet_conv_wgrad) on a side stream beside the pooled BatchNorm-backward
apply pass (bn_bwd_apply_pool_split, rpnet_bn_bwd with pool_w) on the main stream, nothing else on the machine.  The apply pass's
output planes are compared with those of the same call run alone: any differing element is a wrong 2 x 2 window decision.

    RPNET_BN_POOL_DRAIN=0 RPNET_BN_POOL_ALONE=0 python tools/repro_pool_fault.py [repeats]     # the unguarded pass
    python tools/repro_pool_fault.py [repeats]                                                 # the library's defaults (both guards)

The two switches are read once per process (static initialisers in csrc/bn.hip), so each configuration is its own process:
tools/repro_pool_fault.sh runs the four combinations and writes profiles/r05_pool_fault_repro.txt."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rpnet_amd.functional as RF  # noqa: E402
from rpnet_amd.hip import call, ptr, query  # noqa: E402

dev = torch.device("cuda", 0)
repeats = int(sys.argv[1]) if len(sys.argv) > 1 else 200
g = torch.Generator(device="cpu").manual_seed(5)
# the Conv1-level pooled pass of the headline step: 16 images of 256 x 256, 64 channels, two statistic groups
N, H, W, Cc, groups = 16, 256, 256, 64, 2
y = (torch.randn(N, H, W, Cc, generator=g) * 0.7).to(dev)
dz = torch.randn(N, H // 2, W // 2, Cc, generator=g).to(dev)
gamma, beta = (0.8 + 0.4 * torch.rand(Cc, generator=g)).to(dev), ((torch.rand(Cc, generator=g) - 0.5) * 0.4).to(dev)
stats = torch.empty(4, groups, Cc, device=dev)
wsb = query("rpnet_bn_workspace_bytes", Cc, groups)
ws = torch.empty(wsb // 8 + 2, device=dev, dtype=torch.float64)
rm, rv, nbt = torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev), torch.zeros((), device=dev, dtype=torch.long)
call("rpnet_bn_stats", ptr(y), N, H * W, Cc, groups, ptr(gamma), ptr(beta), ptr(rm), ptr(rv), ptr(nbt), 0.1, 1e-5, ptr(stats[0]), ptr(stats[1]),
     ptr(stats[2]), ptr(stats[3]), ptr(ws), wsb)


def apply_pass():
    dys = torch.empty(2, N, H, W, Cc, device=dev, dtype=torch.float16)
    sdy = torch.empty(1, device=dev)
    dg, db = torch.empty(Cc, device=dev), torch.empty(Cc, device=dev)
    call("rpnet_bn_bwd", ptr(dz), ptr(y), ptr(gamma), ptr(stats[0]), ptr(stats[1]), ptr(stats[2]), ptr(stats[3]), None, ptr(dys), 2, ptr(sdy),
         ptr(dg), ptr(db), N, H * W, Cc, groups, 0, None, None, 0, W, ptr(ws), wsb, None, 0)
    return dys


# the weight gradient that runs beside it: 256 -> 256 at 64 x 64, batch 16 (an LDS-DMA launch of ~130 us: 147 KB of LDS per block)
M = (16, 64, 64)
xw = RF.split_f16(torch.randn(*M, 256, generator=g).to(dev), torch.tensor([2.0 ** -13], device=dev), want_scale=False, planes=2)[0]
dyw = RF.split_f16(torch.randn(*M, 256, generator=g).to(dev), torch.tensor([2.0 ** -13], device=dev), want_scale=False, planes=2)[0]
one = torch.tensor([1.0], device=dev)
dw = torch.empty(256, 256, 3, 3, device=dev)
wd = RF._desc(xw, None, None, None, None, 0, None, None, *M, 9, 0, co_split=(256, 0), wgrad=True)
wd.split_planes = 2
wd.acc_scale_x, wd.acc_scale_dy = ptr(one), ptr(one)
wwb = query("rpnet_conv_wgrad_workspace_bytes", *M, 256, 256, 9)
wws = torch.empty(wwb // 4 + 4, device=dev)
side = torch.cuda.Stream(dev)

ref = apply_pass()
torch.cuda.synchronize()
assert torch.equal(apply_pass(), ref), "the pass alone is not reproducible"
hits, wrong = 0, []
for r in range(repeats):
    with torch.cuda.stream(side):
        for _ in range(3):
            call("rpnet_conv_wgrad", C.byref(wd), ptr(dyw), ptr(dw), 256, 0, 256, 256, ptr(wws), wwb)
    out = apply_pass()
    torch.cuda.synchronize()
    nd = int((out.view(torch.int16) != ref.view(torch.int16)).sum())
    if nd:
        hits += 1
        wrong.append(nd)
print(f"RPNET_BN_POOL_DRAIN={os.environ.get('RPNET_BN_POOL_DRAIN', '1')} RPNET_BN_POOL_ALONE={os.environ.get('RPNET_BN_POOL_ALONE', '1')}: "
      f"{hits} of {repeats} apply passes beside a weight-gradient launch differ from the pass alone"
      + (f" (differing 16-bit values per hit: min {min(wrong)}, max {max(wrong)})" if wrong else ""))
