"""Where does the HIP path's gradient distance to the fp64 oracle come from?  Same episode, several loss variants, the
relative L2 error of selected parameter gradients for the HIP path and for the fp32 oracle (both against the fp64 oracle).
   python tools/diag_grads2.py [tag] [conv_math]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import rpnet_oracle as O  # noqa: E402
from tests.helpers import episode_tensors, load_cfg  # noqa: E402
from tests.test_gpu_model import build  # noqa: E402
import rpnet_amd.functional as RF  # noqa: E402
import rpnet_amd.modules as RM  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "m64_train"
RF.set_conv_math(sys.argv[2] if len(sys.argv) > 2 else "f32")
RM._F16_MIN_PIXELS = 0
g = dict(np.load(f"tests/golden/{tag}.npz"))
size, B, T, training, seed = (int(v) for v in g["meta"])
cfg = load_cfg(T)
cpu, _ = episode_tensors(seed, B, size)
gpu, _ = episode_tensors(seed, B, size, "cuda:0")
NAMES = ["cre.q.0.weight", "cre.q.1.weight", "cre.w_k.0.weight", "cre.w_q.0.weight", "cre.w_k.1.weight", "encoder.Up_conv4.conv.3.weight",
         "encoder.Up_conv4.conv.4.weight", "encoder.Up_conv4.conv.1.weight", "encoder.Up_conv4.conv.1.bias", "encoder.Up_conv4.conv.0.weight",
         "encoder.Up4.up.2.weight", "encoder.Up4.up.1.weight", "encoder.Up_conv5.conv.3.weight", "encoder.Conv3.conv.4.weight", "encoder.Conv5.conv.0.weight", "encoder.Conv3.conv.3.weight", "encoder.Conv1.conv.3.weight"]


KEYS = ("sq(refinement[0])", "all")


def variants(out, ql, dice_ce):
    last = dice_ce(out["output"], ql)
    refs = sum(dice_ce(v, ql) for v in out["refinement"].values())
    return {"dice_ce(output)": last, "sum dice_ce(refinement)": refs, "align": out["align_loss"] * 1.0,
            "sq(refinement[0])": out["refinement"][0].square().mean(), "all": last + refs + out["align_loss"]}


def oracle(dtype):
    res = {}
    for key in KEYS:
        P = {}
        for k, v in O.seeded_params(requires_grad=True).items():
            t = v.detach().to(dtype) if v.is_floating_point() else v.detach().clone()
            P[k] = t.clone().requires_grad_(v.requires_grad)
        c = lambda t: t.to(dtype)  # noqa: E731
        si, fg, bg, qi, ql, appr = cpu
        out = O.rp_net_forward(P, cfg, [[c(s) for s in w] for w in si], [[c(s) for s in w] for w in fg], [[c(s) for s in w] for w in bg],
                               [c(qi[0])], c(appr), True, align=True)
        variants(out, ql, O.dice_ce)[key].backward()
        res[key] = {n: P[n].grad.double() for n in NAMES}
    return res


def hip():
    res = {}
    for key in KEYS:
        net = build(cfg, True)
        si, fg, bg, qi, ql, appr = gpu
        out = net(si, fg, bg, qi, appr_query_labels=appr)
        variants(out, ql, RF.dice_ce)[key].backward()
        res[key] = {n: dict(net.named_parameters())[n].grad.double().cpu() for n in NAMES}
    return res


r64, r32, rh = oracle(torch.float64), oracle(torch.float32), hip()
for key in r64:
    print(f"--- loss = {key}")
    for n in NAMES:
        ref = r64[key][n]
        e = lambda a: float((a - ref).norm() / ref.norm())  # noqa: E731
        print(f"   {n:36s} HIP {e(rh[key][n]):.2e}   fp32 oracle {e(r32[key][n]):.2e}")
