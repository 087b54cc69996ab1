"""Per-parameter gradient comparison of the HIP path against a golden fixture (debug aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.helpers import episode_tensors, load_cfg
from tests.test_gpu_model import build, total_loss

tag = sys.argv[1] if len(sys.argv) > 1 else "m64_train"
g = dict(np.load(f"tests/golden/{tag}.npz"))
size, B, T, training, seed = (int(v) for v in g["meta"])
cfg = load_cfg(T)
(si, fg, bg, qi, ql, appr), ep = episode_tensors(seed, B, size, "cuda:0")
net = build(cfg, True)
out = net(si, fg, bg, qi, appr_query_labels=appr)
loss = total_loss(out, ql, 1.0)
loss.backward()
print("loss", loss.item(), float(g["loss"]))
from tests.helpers import rel_err
for i in range(T):
    print("refinement", i, rel_err(out["refinement"][i], g[f"refinement_{i}"]))
from rpnet_amd import functional as RF
net2 = build(cfg, True)
with torch.no_grad():
    x = torch.cat([si[0][0], qi[0]], 0).reshape(2 * B, size, size, 1)
    d4 = net2.encoder.forward_nhwc(x, RF.WeightCache(), groups=2).x
    print("supp_d4", rel_err(d4[:B].permute(0,3,1,2), g["supp_d4"]), "qry_d4", rel_err(d4[B:].permute(0,3,1,2), g["qry_d4"]))

params = dict(net.named_parameters())
for n, ref, head in zip(g["grad_names"], g["grad_norms"], g["grad_heads"]):
    n = str(n); gr = params[n].grad
    if gr is None or ref < 1e-4: continue
    k = min(32, gr.numel())
    e1 = abs(gr.double().norm().item() - ref) / ref
    e2 = (gr.flatten()[:k].cpu() - torch.from_numpy(head[:k])).abs().max().item() / (np.abs(head[:k]).max() + 1e-12)
    print(f"{n:40s} norm_rel {e1:.2e}  head_rel {e2:.2e}")
