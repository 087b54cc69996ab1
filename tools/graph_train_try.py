import os, sys, time
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/rpnet_amd") else os.getcwd())
import torch, yaml
import bench
import rpnet_amd.functional as RF
from rpnet_amd.parallel import FlatGradBucket
from rpnet_amd.graph import GraphedTrainStep
from rpnet_amd.functional import dice_ce
dev = torch.device("cuda", 0)
cfg = yaml.load(open(os.path.join(bench.ROOT, "yamls", "example.yml")), Loader=yaml.FullLoader); cfg["n_iter_refinement"] = 5
RF.set_async_wgrad(os.environ.get("RPNET_ASYNC_WGRAD", "1") == "1")
net = bench.build_model(cfg, dev); bucket = FlatGradBucket(net); inp = bench.make_inputs(1234, 8, 256, dev)
sc = cfg["align_loss_scaler"]
def loss_fn(out, ql):
    loss = dice_ce(out["output"], ql)
    for v in out["refinement"].values():
        loss = loss + dice_ce(v, ql)
    return loss + sc * out["align_loss"]
for _ in range(3): l0 = bench.step(net, bucket, inp, sc)
torch.cuda.synchronize(); ref = bucket.flat.clone(); l0 = l0.item()
t0 = time.perf_counter()
for _ in range(20): bench.step(net, bucket, inp, sc)
torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 20
g = GraphedTrainStep(net, bucket, loss_fn)
si, fg, bg, qi, ql, appr = inp
l1 = g(si, fg, bg, qi, ql, appr); torch.cuda.synchronize()
print("loss eager", l0, "graph", l1.item(), "grad rel diff", float((bucket.flat - ref).abs().max() / ref.abs().max()))
t0 = time.perf_counter()
for _ in range(20): g(si, fg, bg, qi, ql, appr)
torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 20
print(f"eager {te*1e3:.2f} ms/step ({8/te:.1f} pairs/s), graph replay {tg*1e3:.2f} ms/step ({8/tg:.1f} pairs/s)")
