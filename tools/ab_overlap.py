"""A/B of what runs beside what in the training step (one process = one variant; run the variants back to back on one box):
  RPNET_BN_LDS=big        the BatchNorm reduction passes with all per-thread sums in LDS at once (16 - 20 KB per block: such a
                          block does not fit beside a resident LDS-DMA convolution block, 147 - 156 KB of the CU's 160)
  RPNET_COMPUTE_PRIORITY=0  the step's main chain on the caller's (default-priority) stream instead of the high-priority
                          one of RF.use_compute_stream; the weight-gradient side streams are at the default priority either way
  RPNET_WGRAD_DEFER=0     async weight gradients launched in front of their layer's dgrad instead of behind it
  RPNET_CRE_STREAMS_TRAIN=0  both CRE branches (w_k, w_q) on one stream
  RPNET_DICE_MULTI=0      one dice_ce launch pair per loss term instead of the multi-tensor pair
  RPNET_ENC_STREAMS=0|1|2 the encoder's support / query calls as two chains on two streams: never / when they are separate
                          calls anyway (multi-shot, multi-way) / also for 1-way 1-shot
  AB_CONFIG=c2            BASELINE configs[2] (5-shot, batch 16)
  AB_CONFIG=c5            BASELINE configs[4] (2-way 512^2 T=10 batch 4, one fp16 plane) instead of configs[1]
Prints one line: variant, ms per step (wall clock around `steps` steps, synchronised on both sides), pairs/s.
Usage: python tools/ab_overlap.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, yaml
import bench
import rpnet_amd.functional as RF
from rpnet_amd.parallel import FlatGradBucket

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
cfg = yaml.load(open(os.path.join(bench.ROOT, "yamls", "example.yml")), Loader=yaml.FullLoader)
c5 = os.environ.get("AB_CONFIG") == "c5"
c2 = os.environ.get("AB_CONFIG") == "c2"        # BASELINE configs[2]: 5-shot, batch 16
cfg["n_iter_refinement"] = 10 if c5 else 5
RF.set_conv_math("f16" if c5 else "f16x2")
RF.set_async_wgrad(os.environ.get("RPNET_ASYNC_WGRAD", "1") == "1")
main = RF.use_compute_stream(dev)
prio = getattr(main, "priority", "?")
if True:
    net = bench.build_model(cfg, dev)
    bucket = FlatGradBucket(net)
    inp = bench.make_inputs(1234, 4 if c5 else (16 if c2 else 8), 512 if c5 else 256, dev, 5 if c2 else 1, 2 if c5 else 1)
    for _ in range(3):
        bench.step(net, bucket, inp, cfg["align_loss_scaler"])
    best = None
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            bench.step(net, bucket, inp, cfg["align_loss_scaler"])
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / steps
        best = ms if best is None else min(best, ms)
batch = 4 if c5 else (16 if c2 else 8)
r4 = " ".join(f"{k[6:].lower()}={v}" for k, v in sorted(os.environ.items())
              if k in ("RPNET_MASK_SKIP", "RPNET_WGRAD_KEEPALIVE", "RPNET_PACK_STREAM", "RPNET_CONV1_RECOMPUTE", "RPNET_BN_POOL_ALONE",
                       "RPNET_BN_POOL_DRAIN", "RPNET_BNBWD_FUSE"))
print(f"{'configs[4]' if c5 else ('configs[2]' if c2 else 'configs[1]')} [{r4 or 'round-4 defaults'}] enc_streams={os.environ.get('RPNET_ENC_STREAMS', 'default')} BN_LDS={os.environ.get('RPNET_BN_LDS', 'window')} main_priority={prio} "
      f"wgrad_defer={os.environ.get('RPNET_WGRAD_DEFER', '1')} cre_streams_train={os.environ.get('RPNET_CRE_STREAMS_TRAIN', '1')} "
      f"dice_multi={os.environ.get('RPNET_DICE_MULTI', '1')} async={os.environ.get('RPNET_ASYNC_WGRAD', '1')}: "
      f"best of 3 x {steps} steps {best:.3f} ms/step = {batch / best * 1e3:.1f} pairs/s", flush=True)
