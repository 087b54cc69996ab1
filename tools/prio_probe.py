"""Does a HIGH-PRIORITY main stream shorten the default-mode training step?  (HIP offers two priority levels on this stack: 0 and -1.)
The step of BASELINE configs[1] timed on the default stream and on a priority -1 stream, side streams at priority 0, alternating."""
import os, sys, time
import torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import rpnet_amd.functional as RF  # noqa: E402
from rpnet_amd.parallel import FlatGradBucket  # noqa: E402
dev = torch.device("cuda", 0)
cfg = yaml.load(open(os.path.join(ROOT, "yamls", "example.yml")), Loader=yaml.FullLoader)
RF.set_async_wgrad(True)
RF._MASK_SKIP = False
net = bench.build_model(cfg, dev)
bucket = FlatGradBucket(net)
inp = bench.make_inputs(1234, 8, 256, dev)
hi = torch.cuda.Stream(dev, priority=-1)


def timed(stream, n=20):
    with torch.cuda.stream(stream):
        for _ in range(3):
            bench.step(net, bucket, inp, cfg["align_loss_scaler"])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            bench.step(net, bucket, inp, cfg["align_loss_scaler"])
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3


for _ in range(3):
    a = timed(torch.cuda.default_stream(dev))
    b = timed(hi)
    print(f"default stream {a:.3f} ms   priority -1 main stream {b:.3f} ms", flush=True)
