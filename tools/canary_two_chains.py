"""Run-to-run bit reproducibility of the training step in the configuration that showed round 4's pooled-pass fault most often
(profiles/r04_pool_apply_fault.txt, profiles/r05_pool_fault_repro.txt): 2-way 128^2, batch 4, the encoder's two calls as two chains
on two streams, weight gradients through autograd (no side stream) — and, second line, with the weight gradients on their side
stream.  Prints `<name>: K of N repeats differ`.  The process environment selects the guards (RPNET_BN_POOL_DRAIN / _ALONE, default
on) and the LDS form of the BatchNorm reductions (RPNET_BN_LDS=big raised the rate to 8 of 8); tests/test_gpu_model.py runs it
with the guards on and requires 0.    python tools/canary_two_chains.py [repeats]"""
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
# CANARY_KEEPALIVE=1 (round 6): every tensor the host code allocates through torch's factory functions during a step stays referenced until
# the step's device work has been synchronised — the caching allocator then cannot hand a block that one stream still reads to an
# allocation of another stream.  Same kernels, same streams, same timing: if the fault goes away with this switch it is a host-side
# lifetime race (the round-5 advisor's hypothesis); if it stays, it is not one of these tensors.
_KEEP = []
if os.environ.get("CANARY_KEEPALIVE") == "1":
    def _wrap(fn):
        def f(*a, **k):
            out = fn(*a, **k)
            _KEEP.append(out)
            return out
        return f
    for _n in ("empty", "zeros", "ones", "full", "empty_like", "zeros_like", "ones_like", "stack", "cat", "tensor"):
        setattr(torch, _n, _wrap(getattr(torch, _n)))
RM._F16_MIN_PIXELS = 0
RM._ENC_STREAMS = 1
if os.environ.get("CANARY_TILE"):        # force a tile variant of rpnet_conv_fwd (13 = variant 12: the 64-wide LDS-DMA form, 116 - 120 KB of LDS)
    RF.TUNE["tile"] = int(os.environ["CANARY_TILE"])
cfg = load_cfg(2)
(si, fg, bg, qi, ql, appr), _ = episode_tensors(91, 4, 128, "cuda:0", n_shots=1, n_ways=2)


def run(asyncw):
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
    res = out["output"].detach().clone(), {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    torch.cuda.synchronize()
    _KEEP.clear()
    return res


print("env:", {k: v for k, v in os.environ.items() if k.startswith("RPNET_")})
for name, asyncw in (("two chains, weight gradients through autograd", False), ("two chains, weight gradients on their side stream", True)):
    ref = run(asyncw)
    nbad, worst = 0, 0.0
    for _ in range(reps):
        got = run(asyncw)
        bad = [n for n in ref[1] if not torch.equal(ref[1][n], got[1][n])]
        if bad or not torch.equal(ref[0], got[0]):
            nbad += 1
            worst = max([worst] + [float((got[1][n] - ref[1][n]).abs().max() / (ref[1][n].abs().max() + 1e-30)) for n in bad])
    print(f"{name}: {nbad} of {reps} repeats differ (worst {worst:.1e})", flush=True)
