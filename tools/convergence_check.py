"""Training under each convolution arithmetic from the same seed (tools/convergence_check.py [steps] [lr]): the loss
curves of f16x2 / bf16x3 / f32 (and the reduced-precision one-plane f16) must agree — to rounding for the first steps, statistically afterwards (a training
trajectory amplifies rounding like any chaotic system).  One JSON line."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import yaml
import rpnet_amd.functional as RF
import train_rpnet as T

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
lr = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-4
cfg = yaml.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "yamls", "example.yml")), Loader=yaml.FullLoader)
dev = torch.device("cuda", 0)
out = {"steps": steps, "lr": lr, "batch": 4, "size": 128}
import rpnet_amd.modules as RM
RM._F16_MIN_PIXELS = 0      # the fp16 planes at this small size too (the default keeps a batch this small on bf16 planes)
for math in ("f32", "bf16x3", "f16x2", "f16"):
    RF.set_conv_math(math)
    torch.manual_seed(1234)
    _, hist = T.train(cfg, steps, 4, 128, dev, lr=lr, log_every=0, seed=7)
    h = torch.tensor(hist)
    out[math] = {"first5": [round(v, 5) for v in hist[:5]], "mean_steps_20_40": round(h[20:40].mean().item(), 4),
                 "mean_last_20": round(h[-20:].mean().item(), 4), "min": round(h.min().item(), 4)}
print(json.dumps(out))
