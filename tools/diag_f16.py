"""Diagnostic (not a test): error of the one-plane fp16 arithmetic against the fp32 oracle (small sizes) and against the
fp32-equivalent f16x2 step (any size): per-iteration logit error / max |logit|, Dice, foreground fraction, loss,
gradient relative L2.   python tools/diag_f16.py [size B T ways]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import episode_tensors, load_cfg  # noqa: E402
from tests.test_gpu_model import build, total_loss  # noqa: E402
import rpnet_amd.functional as RF  # noqa: E402
import rpnet_amd.modules as RM  # noqa: E402

RM._F16_MIN_PIXELS = 0
size, B, T, ways = (int(v) for v in (sys.argv[1:5] + ["64", "2", "2", "2"][len(sys.argv) - 1:]))
cfg = load_cfg(T)
(si, fg, bg, qi, ql, appr), _ = episode_tensors(66 + size, B, size, "cuda:0", n_shots=1, n_ways=ways)


def dice(lg):
    pred = (lg.softmax(1)[:, 1] > 0.5).long()
    return float(2.0 * (pred * ql).sum() / (pred.sum() + ql.sum() + 1e-7)), float(pred.float().mean())


def run(math, forced=None):
    RF.set_conv_math(math)
    net = build(cfg, True)
    net.forced_masks = forced
    out = net(si, fg, bg, qi, appr_query_labels=appr)
    loss = total_loss(out, ql, cfg["align_loss_scaler"])
    loss.backward()
    torch.cuda.synchronize()
    return net, out, loss


ref_net, ref, ref_loss = run(os.environ.get("REF_MATH", "f16x2"))
FORCE = os.environ.get("FORCE") == "1"
for math in sys.argv[5:] or ["f16"]:
    forced = None
    if FORCE:   # teacher forcing: the reference run's masks into iterations 1 ..
        forced = {i: torch.nn.functional.avg_pool2d((ref["refinement"][i - 1].detach().softmax(1)[:, 1] > 0.5).float()[:, None], 4)[:, 0]
                  for i in range(1, T)}
    net, out, loss = run(math, forced)
    print(f"== {math} vs f16x2, {ways}-way {size}^2 B{B} T{T}: loss {loss.item():.6f} vs {ref_loss.item():.6f} "
          f"({abs(loss.item() - ref_loss.item()) / abs(ref_loss.item()):.2e})")
    for i in range(T):
        a, b = out["refinement"][i], ref["refinement"][i]
        e = float((a - b).abs().max() / b.abs().max())
        rms = float((a - b).square().mean().sqrt() / b.abs().max())
        fl = float(((a.softmax(1)[:, 1] > 0.5) != (b.softmax(1)[:, 1] > 0.5)).float().mean())
        print(f"  it {i}: flips {fl:.2e} logit max err {e:.2e} rms {rms:.2e}  dice {dice(a)[0]:.5f} vs {dice(b)[0]:.5f}  fg {dice(a)[1]:.5f} vs {dice(b)[1]:.5f}")
    worst = []
    for (n, p), (_, q) in zip(net.named_parameters(), ref_net.named_parameters()):
        if p.grad is not None and q.grad.norm() > 1e-6:
            worst.append((float((p.grad.double() - q.grad.double()).norm() / q.grad.double().norm()), n))
    worst.sort(reverse=True)
    print("  grad rel L2: worst", [(round(e, 4), n) for e, n in worst[:4]], "median", round(worst[len(worst) // 2][0], 4))
