import copy, sys, torch
sys.path.insert(0, '.')
import rpnet_amd.functional as RF
from tests.helpers import rnd
from tests.test_gpu_ops import _mk_layer, nhwc
DEV = 'cuda:0'
RF.set_conv_math("f16x2")
def run(up4, N, H, W, cin, cout, groups):
    RF._UP4 = up4
    layer = _mk_layer(cin, cout, 3, 191)
    a = rnd(192, N, cin, H // 2, W // 2); go = rnd(194, N, cout, H, W)
    conv, bn = copy.deepcopy(layer[0]).to(DEV), copy.deepcopy(layer[1]).to(DEV).train()
    ag = nhwc(a).to(DEV).requires_grad_(True)
    sc = torch.tensor([2.0 ** (int(torch.ceil(torch.log2(a.abs().max())).item()) - 15)], device=DEV)
    z = RF.conv_bn_relu_op(RF.Operand(ag, scale=sc), conv, bn, RF.WeightCache(), True, groups=groups, upsample=True, out_split=False).x
    z.backward(nhwc(go).to(DEV)); torch.cuda.synchronize()
    return z.detach(), ag.grad
import torch.nn.functional as F
def ref64(N, H, W, cin, cout, groups):
    layer = _mk_layer(cin, cout, 3, 191)
    a = rnd(192, N, cin, H // 2, W // 2); go = rnd(194, N, cout, H, W)
    c_ref, b_ref = copy.deepcopy(layer[0]).double(), copy.deepcopy(layer[1]).double().train()
    ar = a.double().requires_grad_(True)
    xin = F.interpolate(ar, scale_factor=2, mode="nearest")
    per = N // groups
    ref = torch.cat([F.relu(b_ref(c_ref(xin[g * per:(g + 1) * per]))) for g in range(groups)], 0)
    ref.backward(go.double())
    return ar.grad.permute(0, 2, 3, 1).contiguous()
for case in [(4, 32, 64, 256, 128, 2), (4, 32, 64, 256, 128, 1), (1, 32, 64, 256, 128, 1)]:
    N, H, W, cin, cout, g = case
    z4, d4 = run(True, *case); z9, d9 = run(False, *case)
    r = ref64(*case).to(DEV)
    print(case, "da4 vs fp64", float((d4.double() - r).abs().max() / r.abs().max()), "da9 vs fp64", float((d9.double() - r).abs().max() / r.abs().max()))
    e = (d4 - d9).abs(); thr = 1e-5 * d9.abs().max()
    bad = (e > thr)
    print(case, "z err", float((z4 - z9).abs().max() / z9.abs().max()), "da err", float(e.max() / d9.abs().max()), "bad frac", float(bad.float().mean()))
    if bad.any():
        b = bad.any(-1)  # [N,h,w]
        print(" bad pixels per image:", b.flatten(1).sum(1).tolist())
        print(" bad rows (y):", b.any(0).any(-1).nonzero().flatten().tolist())
        print(" bad cols (x):", b.any(0).any(0).nonzero().flatten().tolist())
        print(" bad channels count:", int(bad.any(0).any(0).any(0).sum()), "of", cin)
