"""Diagnostic (not a test): run-to-run determinism of the training step's gradients, per parameter, under stream layouts and
kernel switches toggled IN-PROCESS: python tools/diag_two_chains.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import episode_tensors, load_cfg  # noqa: E402
from tests.test_gpu_model import build, total_loss  # noqa: E402
import rpnet_amd.functional as RF  # noqa: E402
import rpnet_amd.modules as RM  # noqa: E402
from rpnet_amd.parallel import FlatGradBucket  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 16
RM._F16_MIN_PIXELS = 0
cfg = load_cfg(2)
(si, fg, bg, qi, ql, appr), _ = episode_tensors(91, 4, 128, "cuda:0", n_shots=1, n_ways=2)


def run(enc, asyncw):
    RM._ENC_STREAMS = enc
    net = build(cfg, True)
    bucket = FlatGradBucket(net) if asyncw else None
    RF.set_async_wgrad(asyncw)
    if bucket is not None:
        bucket.zero()
    out = net(si, fg, bg, qi, appr_query_labels=appr)
    total_loss(out, ql, 1.0).backward()
    if bucket is not None:
        bucket.allreduce()
    torch.cuda.synchronize()
    return out["output"].detach().clone(), {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}


def variant(name, enc, asyncw, **sw):
    saved = {}
    for k, v in sw.items():
        mod, attr = (RM, k[3:]) if k.startswith("RM_") else (RF, k[3:])
        if attr == "TUNE":
            saved[k] = dict(RF.TUNE)
            RF.TUNE.update(v)
        else:
            saved[k] = getattr(mod, attr)
            setattr(mod, attr, v)
    try:
        ref = run(enc, asyncw)
        nbad, first, worst = 0, {}, 0.0
        for _ in range(reps):
            got = run(enc, asyncw)
            bad = [n for n in ref[1] if not torch.equal(ref[1][n], got[1][n])]
            if bad or not torch.equal(ref[0], got[0]):
                nbad += 1
                key = f"{len(bad)}:{bad[-1] if bad else 'logits'}"
                first[key] = first.get(key, 0) + 1
                worst = max([worst] + [float((got[1][n] - ref[1][n]).abs().max() / (ref[1][n].abs().max() + 1e-30)) for n in bad])
        print(f"{name:58s} enc={enc} async={int(asyncw)}: {nbad:2d} of {reps} repeats differ (worst {worst:.1e}) {first}", flush=True)
    finally:
        for k, v in saved.items():
            mod, attr = (RM, k[3:]) if k.startswith("RM_") else (RF, k[3:])
            if attr == "TUNE":
                RF.TUNE.clear()
                RF.TUNE.update(v)
            else:
                setattr(mod, attr, v)


print("env:", {k: v for k, v in os.environ.items() if k.startswith("RPNET_") or k.startswith("AMD_")})
variant("single stream (no async wgrad, no CRE stream)", 0, False, RM__CRE_STREAMS_TRAIN=False)
variant("async wgrad only", 0, True, RM__CRE_STREAMS_TRAIN=False)
variant("CRE stream only", 0, False)
variant("default one-chain (async + CRE stream)", 0, True)
variant("two chains + async + CRE stream", 1, True)
variant("two chains, no async", 1, False)
variant("async wgrad only, no pool fusion", 0, True, RM__CRE_STREAMS_TRAIN=False, RF__POOL_FUSE=False)
variant("async wgrad only, register-staged wgrad (tune 8)", 0, True, RM__CRE_STREAMS_TRAIN=False, RF_TUNE={"wgrad": 8})
variant("async wgrad only, no DMA conv kernels", 0, True, RM__CRE_STREAMS_TRAIN=False, RF_TUNE={"tile": 0x10000})
variant("async wgrad only, defer 0", 0, True, RM__CRE_STREAMS_TRAIN=False, RF__WGRAD_DEFER=0)
variant("async wgrad only, no z skip", 0, True, RM__CRE_STREAMS_TRAIN=False, RM__ZSKIP=False)
variant("async wgrad only, no conv1 recompute", 0, True, RM__CRE_STREAMS_TRAIN=False, RF__CONV1_RECOMP=False)
variant("async wgrad only, record_stream instead of keep-alive", 0, True, RM__CRE_STREAMS_TRAIN=False, RF__KEEPALIVE=False)
