"""Diagnostic: does any kernel of the training step read memory nobody wrote?  Every torch.empty of the host layer is filled
with NaN (floats) / 0x7fff patterns first; a NaN in a gradient or output names the consumer.  DIAG_ENC=0|1 selects the encoder
stream layout.   python tools/diag_poison.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import episode_tensors, load_cfg  # noqa: E402
from tests.test_gpu_model import build, total_loss  # noqa: E402
import rpnet_amd.functional as RF  # noqa: E402
import rpnet_amd.modules as RM  # noqa: E402
from rpnet_amd.parallel import FlatGradBucket  # noqa: E402

_empty = torch.empty
_empty_like = torch.empty_like


def poisoned(*a, **k):
    t = _empty(*a, **k)
    if t.is_cuda and t.numel() and t.dtype in (torch.float32, torch.float64, torch.float16, torch.bfloat16):
        t.fill_(float("nan"))
    return t


def poisoned_like(*a, **k):
    t = _empty_like(*a, **k)
    if t.is_cuda and t.numel() and t.dtype.is_floating_point:
        t.fill_(float("nan"))
    return t


torch.empty = poisoned
torch.empty_like = poisoned_like
RM._F16_MIN_PIXELS = 0
RM._ENC_STREAMS = int(os.environ.get("DIAG_ENC", "0"))
cfg = load_cfg(2)
ways = int(os.environ.get("DIAG_WAYS", "2"))
(si, fg, bg, qi, ql, appr), _ = episode_tensors(91, 4, 128, "cuda:0", n_shots=1, n_ways=ways)
asyncw = os.environ.get("DIAG_ASYNC", "1") == "1"
net = build(cfg, True)
bucket = FlatGradBucket(net) if asyncw else None
RF.set_async_wgrad(asyncw)
for rep in range(2):
    if bucket is not None:
        bucket.zero()
    else:
        for p in net.parameters():
            p.grad = None
    out = net(si, fg, bg, qi, appr_query_labels=appr)
    loss = total_loss(out, ql, 1.0)
    loss.backward()
    if bucket is not None:
        bucket.allreduce()
    torch.cuda.synchronize()
    bad = [n for n, p in net.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    print(f"rep {rep} enc={RM._ENC_STREAMS} async={asyncw} ways={ways}: loss {float(loss):.6f} finite logits {bool(torch.isfinite(out['output']).all())}; "
          f"non-finite gradients: {len(bad)} {bad[:8]}", flush=True)
