"""Evaluation-call latency (batch 2, 256x256, T=10: the reference driver's call shape; `python tools/bench_eval.py 8` for
another batch), eager vs hipGraph replay."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.helpers import episode_tensors, load_cfg
from tests.test_gpu_model import build
from rpnet_amd.graph import GraphedEval
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
if os.environ.get("EVAL_CORR_PLANES") == "0":  # A/B: the correlation's planes by a split pass behind the kernel (round 5's eval path)
    import rpnet_amd.functional as _RF0
    _RF0._CORR_PRED_PLANES = False
if os.environ.get("EVAL_UP4") == "0":         # A/B: the two up_conv layers on the nine-tap form (round 5's eval path)
    import rpnet_amd.functional as _RF
    _RF._UP4 = False
cfg = load_cfg(10); net = build(cfg, False); g = GraphedEval(net)
(si, fg, bg, qi, ql, appr), _ = episode_tensors(5, B, 256, "cuda:0")
def run(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    te = run(lambda: net(si, fg, bg, qi, appr_query_labels=appr))
tg = run(lambda: g(si, fg, bg, qi, appr_query_labels=appr))
print(f"eval call batch {B}, 256x256, T=10: eager {te:.2f} ms ({B / te * 1e3:.0f} pairs/s), hipGraph replay {tg:.2f} ms ({te / tg:.2f}x)")
