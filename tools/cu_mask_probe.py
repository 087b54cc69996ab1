"""Does a CU mask on the weight-gradient streams act as the priority HIP's stream priorities do not give?  The training step of
BASELINE configs[1] (bench.step, default schedule) with the weight-gradient side stream (and, with REDUCE=1, its reduce stream) created by
hipExtStreamCreateWithCUMask on the first n of the 256 CU bits (KFD deals the bits round-robin over XCDs, then shader engines: n a multiple
of 32 takes n / 32 CUs of every shader engine of every XCD), against plain streams, alternating on one box.
Usage: python tools/cu_mask_probe.py [out.txt]      env: MASKS="256,224,192,160,128" ROUNDS=2 STEPS=10 REDUCE=0|1 B SIZE ITERS MATH"""
import ctypes as C
import os
import sys
import time

import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import rpnet_amd.functional as RF  # noqa: E402
from rpnet_amd.parallel import FlatGradBucket  # noqa: E402

out = open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
hip = C.CDLL("libamdhip64.so")


def masked_stream(n):
    words = (C.c_uint32 * 8)()
    for b in range(n):
        words[b // 32] |= 1 << (b % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask({n}) -> {rc}")
    back = (C.c_uint32 * 8)()
    hip.hipExtStreamGetCUMask(s, 8, back)
    return torch.cuda.ExternalStream(s.value, device=dev), [hex(w) for w in back]


cfg = yaml.load(open(os.path.join(ROOT, "yamls", "example.yml")), Loader=yaml.FullLoader)
cfg["n_iter_refinement"] = int(os.environ.get("ITERS", "5"))
if os.environ.get("MATH"):
    RF.set_conv_math(os.environ["MATH"])
RF._MASK_SKIP = False
RF.set_async_wgrad(True)
B, SIZE = int(os.environ.get("B", "8")), int(os.environ.get("SIZE", "256"))
WAYS = int(os.environ.get("WAYS", "1"))
net = bench.build_model(cfg, dev)
bucket = FlatGradBucket(net)
inp = bench.make_inputs(1234, B, SIZE, dev, 1, WAYS)
masks = [int(m) for m in os.environ.get("MASKS", "256,224,192,160,128").split(",")]
rounds, steps = int(os.environ.get("ROUNDS", "2")), int(os.environ.get("STEPS", "10"))
also_reduce = os.environ.get("REDUCE", "0") == "1"
streams = {}


def streams_of(m):
    """made on first use: every masked stream is a hardware queue of its own, and the legs in front of it run without it"""
    if m not in streams:
        if m >= 256:
            streams[m] = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev), "plain")
        else:
            s, got = masked_stream(m)
            r, _ = masked_stream(m) if also_reduce else (torch.cuda.Stream(device=dev), None)
            streams[m] = (s, r, " ".join(got))
    return streams[m]


print(f"# tools/cu_mask_probe.py: B={B} SIZE={SIZE} ways={WAYS} T={cfg['n_iter_refinement']} {RF.conv_math()}; the weight-gradient side stream"
      f"{' and its reduce stream' if also_reduce else ''} on the first n CU bits; {steps} timed steps per leg", file=out, flush=True)
ref = None
for rd in range(rounds):
    for m in masks:
        side, red, got = streams_of(m)
        RF._ASYNC["side"][dev] = side
        RF._ASYNC["side"][("reduce", dev)] = red
        for _ in range(3):
            bench.step(net, bucket, inp, cfg["align_loss_scaler"])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            bench.step(net, bucket, inp, cfg["align_loss_scaler"])
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        g = bucket.flat.double().abs().sum().item()
        if ref is None:
            ref = g
        print(f"round {rd}  n = {m:3d}  {ms:7.3f} ms/step  {B / ms * 1e3:7.1f} pairs/s   gradient checksum {'same' if g == ref else 'DIFFERS'}   mask read back: {got}",
              file=out, flush=True)
