import copy, sys, torch
sys.path.insert(0, '.')
import torch.nn.functional as F
import rpnet_amd.functional as RF
from rpnet_amd import hip
from tests.helpers import rnd
from tests.test_gpu_ops import _mk_layer, nhwc
DEV = 'cuda:0'
RF.set_conv_math("f16x2")
RF._UP4 = False
def run(N, H, W, cin, cout, groups, ups, tile=0):
    RF.TUNE["tile"] = tile
    layer = _mk_layer(cin, cout, 3, 191)
    hs, ws = (H // 2, W // 2) if ups else (H, W)
    a = rnd(192, N, cin, hs, ws); go = rnd(194, N, cout, H, W)
    conv, bn = copy.deepcopy(layer[0]).to(DEV), copy.deepcopy(layer[1]).to(DEV).train()
    ag = nhwc(a).to(DEV).requires_grad_(True)
    sc = torch.tensor([2.0 ** (int(torch.ceil(torch.log2(a.abs().max())).item()) - 15)], device=DEV)
    chosen = []
    orig = RF.call
    def spy(name, *args):
        if name == "rpnet_conv_fwd":
            chosen.append(hip.query("rpnet_conv_tile_variant", args[0]))
        return orig(name, *args)
    RF.call = spy
    z = RF.conv_bn_relu_op(RF.Operand(ag, scale=sc), conv, bn, RF.WeightCache(), True, groups=groups, upsample=ups, out_split=False).x
    z.backward(nhwc(go).to(DEV)); torch.cuda.synchronize()
    RF.call = orig
    c_ref, b_ref = copy.deepcopy(layer[0]).double(), copy.deepcopy(layer[1]).double().train()
    ar = a.double().requires_grad_(True)
    xin = F.interpolate(ar, scale_factor=2, mode="nearest") if ups else ar
    per = N // groups
    ref = torch.cat([F.relu(b_ref(c_ref(xin[g * per:(g + 1) * per]))) for g in range(groups)], 0)
    ref.backward(go.double())
    r = ar.grad.permute(0, 2, 3, 1).to(DEV)
    zr = ref.permute(0, 2, 3, 1).to(DEV)
    e = (ag.grad.double() - r).abs().flatten(1).max(1)[0] / r.abs().max()
    print((N, H, W, cin, cout, groups, ups, tile), "variants", chosen, "z err", float((z.double() - zr).abs().max() / zr.abs().max()),
          "da err per image", [f"{float(v):.1e}" for v in e])
run(4, 32, 64, 256, 128, 2, True)
run(4, 32, 64, 256, 128, 1, True)
run(4, 32, 64, 256, 128, 2, False)
run(4, 32, 64, 128, 256, 2, False)
for t in (8, 10, 12, 13):   # variant t - 1 forced
    run(4, 32, 64, 256, 128, 2, True, t)
run(8, 32, 64, 256, 128, 2, True)
run(16, 32, 32, 1024, 512, 2, True)
run(16, 64, 64, 512, 256, 2, True)
