#!/usr/bin/env python3
"""Headline benchmark: support/query pairs per second, forward+backward, 1-way 1-shot,
256x256, T=5 refinement iterations, batch 8 per GPU (BASELINE.json configs[1]; configs[3] =
the same per-GPU work on 8 GPUs with the RCCL gradient all-reduce -> weak scaling).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One step = RP_Net.forward (train mode, align loss on) + the harness loss
dice_ce(output) + sum_i dice_ce(refinement[i]) + align_loss (the reference has no train
loop) + backward + the flat-bucket gradient all-reduce.  Synthetic CT-shaped
episodes are generated once and are resident in HBM before the timed region; weights are
name-seeded random init (no dataset / checkpoint exists offline).

Arithmetic of the 3x3 convolutions (`--conv-math`, default bf16x3): every fp32 operand is carried as three
bf16 planes (an exact split) and multiplied on the bf16 matrix pipe with six partial products into fp32
accumulators — fp32-level accuracy (dropped terms <= 2^-23 |x*y|; tests/test_gpu_ops.py::test_split_conv_accuracy
measures it against fp64 next to the fp32-MFMA kernel) at 16/6 of the fp32 matrix rate.  `f32` selects the
v_mfma_f32_32x32x2_f32 kernels; the JSON line carries that figure too (`alt_math`).

Rank 0 prints ONE JSON line.  `roofline` is measured live: every C-ABI call of one extra
step is bracketed by HIP events on the launch stream; the dominant kernel is the implicit-GEMM
convolution (rpnet_conv_fwd: forward + dgrad launches).  The timed steps run with
the weight gradients on a second HIP stream (overlap on); the extra profiled step serialises the
streams so that every launch owns the GPU and its duration is the kernel's own
(RPNET_ASYNC_WGRAD=0 makes the timed steps serial too; profiles/ holds rocprofv3 stats of both).  `cpu_baseline` is
the CPU oracle in as-written mode (= the reference's operator sequence) on the host cores.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist
import yaml

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # same guide: v_mfma_f32_32x32x16_bf16 dense peak
# MFMA partial products issued per algorithmic fp32 multiply-add, and the resulting ceiling in algorithmic FLOPs
# (the fp16 matrix instruction v_mfma_f32_32x32x16_f16 has the bf16 one's rate)
MATH = {"f32": (1, PEAK_F32_MFMA_TFLOPS), "bf16x3": (6, PEAK_BF16_MFMA_TFLOPS / 6), "f16x2": (3, PEAK_BF16_MFMA_TFLOPS / 3)}
GF_PER_PAIR = {(256, 5): 675.4, (128, 1): 138.5}  # algorithmic fwd+bwd GFLOP per pair


def algorithmic_gf_per_pair(size, T, n_shots=1):
    """encoder 82.22 GF/image fwd at 256^2, CRE 10.12 GF/call, backward = 2x forward."""
    s = (size / 256.0) ** 2
    imgs, cre_calls = n_shots + 1, n_shots + T
    return 3.0 * (imgs * 82.22 + cre_calls * 10.12) * s


def build_model(cfg, dev):
    from rpnet_amd.modules import RP_Net
    from rpnet_amd.utils.seeding import seed_module_
    net = RP_Net(cfg={"align": True, "backbone": "UNet"}, backbone_cfg=cfg).to(dev)
    seed_module_(net)
    net.train()
    return net


def make_inputs(seed, B, size, dev, n_shots=1):
    from rpnet_amd.utils.synth import make_episode
    ep = make_episode(seed, B, size, n_shots=n_shots)
    t = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
    return ([[t(s) for s in ep["support_images"][0]]], [[t(s) for s in ep["support_fg"][0]]],
            [[t(s) for s in ep["support_bg"][0]]], [t(ep["query_images"])], t(ep["query_labels"]),
            t(ep["appr_query_labels"]))


def step(net, bucket, inp, scaler):
    from rpnet_amd.functional import dice_ce
    si, fg, bg, qi, ql, appr = inp
    bucket.zero()
    out = net(si, fg, bg, qi, appr_query_labels=appr)
    loss = dice_ce(out["output"], ql)
    for v in out["refinement"].values():
        loss = loss + dice_ce(v, ql)
    loss = loss + scaler * out["align_loss"]
    loss.backward()
    bucket.allreduce()
    return loss


def profile_step(net, bucket, inp, scaler):
    """One extra step with every C-ABI call bracketed by HIP events on the launch stream."""
    from rpnet_amd import hip
    records = []
    orig = hip.call

    def timed(name, *args):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        flops = nbytes = 0.0
        if name in ("rpnet_conv_fwd", "rpnet_conv_wgrad"):
            d = args[0]._obj
            m, ci, co = d.N * d.H * d.W, d.C0 + d.C1, d.Co0 + d.Co1
            flops = 2.0 * m * ci * co * d.taps
            esz = 2.0 * d.split_planes if d.split_planes else 4.0                # operand bytes per element
            nbytes = esz * ((m >> (2 * d.upsample)) * ci + d.taps * ci * co) + 4.0 * m * co   # read x, w; write y
        a.record()
        orig(name, *args)
        b.record()
        records.append((name, flops, nbytes, a, b))

    hip.call = timed
    import rpnet_amd.functional as RF
    RF.call = timed
    was_async = RF._ASYNC["on"]
    RF.set_async_wgrad(False)   # per-kernel durations need each launch to own the GPU: streams serialised here
    try:
        step(net, bucket, inp, scaler)
        torch.cuda.synchronize()
    finally:
        hip.call = orig
        RF.call = orig
        RF.set_async_wgrad(was_async)
    agg = {}
    for name, flops, nbytes, a, b in records:
        e = agg.setdefault(name, [0, 0.0, 0.0, 0.0])
        e[0] += 1
        e[1] += a.elapsed_time(b) * 1e-3
        e[2] += flops
        e[3] += nbytes
    return agg


def pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/r01_pmc_traffic.json: separate --pmc FETCH_SIZE / WRITE_SIZE runs of this very
    command, KiB units, read side doubled per the gfx950 note of MI355X_MICROARCH.md).  A PMC
    pass cannot run inside the timed process, so this is the offline figure or None."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if not os.path.exists(path):
        return None
    d = json.load(open(path))
    n = b = 0.0
    for k, v in d.items():
        if "conv_igemm" in k:
            n += v["launches"]
            b += v["launches"] * (v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"])
    return round(b / n) if n else None


def conv_accuracy_probe(dev):
    """max |err| / max |ref| of one 3x3 convolution (2x32x32, 256 -> 256) against an fp64 CPU reference under each
    arithmetic: the evidence that the default split arithmetic is fp32-accurate (same probe as
    tests/test_gpu_ops.py::test_split_conv_accuracy, forward only)."""
    import ctypes as C
    import torch.nn.functional as F
    import rpnet_amd.functional as RF
    from rpnet_amd.hip import call
    g = torch.Generator().manual_seed(7)
    x = torch.randn(2, 32, 32, 256, generator=g)
    w = torch.randn(256, 256, 3, 3, generator=g) * 0.05
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), padding=1).permute(0, 2, 3, 1)
    xd, wd = x.to(dev), w.to(dev)
    pw = RF.PackedWeight(wd)
    out = {}
    for name, planes in (("f32", 0), ("bf16x3", 3), ("f16x2", 2)):
        y = torch.empty(2, 32, 32, 256, device=dev)
        if planes == 2:      # fp16 planes of x / s, s = a power of two with max|x| / s <= 2^15 (here from the data itself)
            s_in = torch.tensor([2.0 ** (int(torch.ceil(torch.log2(x.abs().max())).item()) - 15)], device=dev)
            xs, sx = RF.split_f16(xd, s_in)
            wps, _, t_row, _ = pw.split_packs(2)
            d = RF._desc(xs, None, wps, None, None, 0, y, None, 2, 32, 32, 9, 0)
            d.split_planes, d.acc_scale_col, d.acc_scale_x = 2, t_row.data_ptr(), sx.data_ptr()
        elif planes:
            xs = RF.split_bf16(xd, planes)
            d = RF._desc(xs[0], None, pw.split_packs(planes)[0], None, None, 0, y, None, 2, 32, 32, 9, 0)
            d.split_planes = planes
        else:
            d = RF._desc(xd, None, pw.wp, None, None, 0, y, None, 2, 32, 32, 9, 0)
        call("rpnet_conv_fwd", C.byref(d))
        out[name] = float((y.double().cpu() - ref).abs().max() / ref.abs().max())
    return out


def cpu_baseline(cfg, size, T, seconds_budget=25.0, net=None, bucket=None, dev=None):
    """The CPU oracle in as-written mode (the reference's operator sequence: all-pairs
    correlation + grid_sample, explicit bilinear up-sampling in getFeatures, prototypes per
    iteration) fwd+bwd on the host cores, batch 1, same loss.  Bounded sample.  With `net` (the benched model: same
    seeded parameters) the same episode also goes through the HIP path under the arithmetic being benched and the two
    results are compared (`parity`) — the oracle in its checker role, at the headline image size."""
    from oracle import rpnet_oracle as O
    from rpnet_amd.utils.synth import make_episode
    ep = make_episode(1234, 1, size)
    t = torch.from_numpy
    si, fg, bg = [[t(ep["support_images"][0][0])]], [[t(ep["support_fg"][0][0])]], [[t(ep["support_bg"][0][0])]]
    qi, ql, appr = [t(ep["query_images"])], t(ep["query_labels"]), t(ep["appr_query_labels"])
    P = O.seeded_params(cfg["mask_refinement_correlation_radius"], requires_grad=True)

    last = {}

    def one():
        for p in P.values():
            p.grad = None
        out = O.rp_net_forward(P, cfg, si, fg, bg, qi, appr, True, align=True, as_written=True)
        loss = O.total_loss(out, ql, cfg["align_loss_scaler"])
        loss.backward()
        last["out"], last["loss"] = out["output"].detach(), loss.item()

    one()  # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        one()
        n += 1
        el = time.perf_counter() - t0
        if el > seconds_budget or n >= 8:
            break
    res = {"value": n / el, "unit": "pairs/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": f"{n} fwd+bwd steps of batch 1 at {size}x{size}, T={T}, oracle as-written mode "
                     f"(reference operator sequence), {el / n:.2f} s/step"}
    if net is not None:
        mv = lambda a: a.to(dev)  # noqa: E731
        loss = step(net, bucket, ([[mv(si[0][0])]], [[mv(fg[0][0])]], [[mv(bg[0][0])]], [mv(qi[0])], mv(ql), mv(appr)),
                    cfg["align_loss_scaler"])
        torch.cuda.synchronize()
        grads = {}
        for name in ("cre.w_k.0.weight", "cre.q.0.weight", "encoder.Up_conv4.conv.3.weight", "encoder.Conv5.conv.0.weight",
                     "encoder.Conv1.conv.3.weight"):
            a, b = dict(net.named_parameters())[name].grad.double().cpu(), P[name].grad.double()
            grads[name] = float((a - b).norm() / b.norm())
        res["parity"] = {"what": f"the same batch-1 {size}x{size} episode and seeded parameters through the HIP path (arithmetic of this "
                                 "run) against the CPU oracle: loss, and relative L2 error of five weight gradients (these are conditioned by "
                                 "ReLU / max-pool / 0.5-threshold switches: the fp32-MFMA kernels measure 2e-4 .. 6e-3 on the same comparison, "
                                 "three bf16 planes 2e-4 .. 7e-3)",
                         "loss_hip": round(loss.item(), 6), "loss_oracle": round(last["loss"], 6),
                         "loss_rel_err": abs(loss.item() - last["loss"]) / abs(last["loss"]), "weight_grad_rel_l2_err": grads}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="episodes (support/query pairs) per GPU")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--iters", type=int, default=5, help="T refinement iterations")
    ap.add_argument("--shots", type=int, default=1, help="support shots (5 with --batch 16 = BASELINE configs[2])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--conv-math", choices=sorted(MATH), default=None,
                    help="arithmetic of the 3x3 convolutions (default: the library's, bf16x3 = fp32-equivalent split)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback)"
    local_rank %= torch.cuda.device_count()   # > 1 rank per device only happens in the gloo plumbing test
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("RPNET_DIST_BACKEND", "nccl")   # "nccl" IS RCCL on ROCm; gloo only for 1-GPU plumbing tests
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    cfg = yaml.load(open(os.path.join(ROOT, "yamls", "example.yml")), Loader=yaml.FullLoader)
    cfg["n_iter_refinement"] = args.iters
    scaler = cfg["align_loss_scaler"]

    from rpnet_amd.parallel import FlatGradBucket, broadcast_parameters
    import rpnet_amd.functional as RF
    RF.set_async_wgrad(os.environ.get("RPNET_ASYNC_WGRAD", "1") == "1")   # weight gradients on a second HIP stream
    if args.conv_math:
        RF.set_conv_math(args.conv_math)
    math = RF.conv_math()
    products, peak = MATH[math]
    net = build_model(cfg, dev)
    broadcast_parameters(net)
    bucket = FlatGradBucket(net)
    inp = make_inputs(1234 + rank, args.batch, args.size, dev, args.shots)   # resident in HBM before timing

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(net, bucket, inp, scaler)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step(net, bucket, inp, scaler)
    fence()
    el = time.perf_counter() - t0
    if math == "f16x2" and not RF.f16_mode():
        # a call below the fp16 threshold (rpnet_amd.modules._F16_MIN_PIXELS: small, launch-bound episodes) ran on bf16 planes
        math = "bf16x3"
        products, peak = MATH[math]
    if world > 1:
        tt = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    assert torch.isfinite(loss).item()

    pairs = world * args.batch * args.steps
    value = pairs / el
    gf_pair = algorithmic_gf_per_pair(args.size, args.iters, args.shots)

    result = None
    # The profiled extra step contains the gradient all-reduce, so EVERY rank runs it (a collective
    # issued by rank 0 alone would never complete); only rank 0 reports.
    agg = profile_step(net, bucket, inp, scaler)
    alt = None
    if world == 1 and not args.no_cpu_baseline:
        # the same step under the other convolution arithmetics, for reference
        alt = {}
        for other in ("f32", "f16x2", "bf16x3"):
            if other == math:
                continue
            RF.set_conv_math(other)
            for _ in range(2):
                step(net, bucket, inp, scaler)
            fence()
            t1 = time.perf_counter()
            for _ in range(5):
                step(net, bucket, inp, scaler)
            fence()
            alt[other] = {"value": round(args.batch * 5 / (time.perf_counter() - t1), 3), "unit": "pairs/s", "steps": 5}
        RF.set_conv_math(math)
    if rank == 0:
        conv = agg.get("rpnet_conv_fwd", [0, 1e-9, 0.0, 0.0])
        wg = agg.get("rpnet_conv_wgrad", [0, 1e-9, 0.0, 0.0])
        achieved = conv[2] / conv[1] / 1e12
        kern_total = sum(v[1] for v in agg.values())
        result = {
            "metric": f"support/query pairs/sec (fwd+bwd, {args.shots}-shot {args.size}x{args.size}, T={args.iters})",
            "value": round(value, 3), "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * el / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "conv_math": {"f32": "v_mfma_f32_32x32x2_f32 on fp32 operands",
                          "bf16x3": "fp32 operands as 3 bf16 planes (exact split), 6 v_mfma_f32_32x32x16_bf16 partial "
                                    "products, fp32 accumulate: fp32-equivalent (dropped terms <= 2^-23 |x*y|)",
                          "f16x2": "fp32 operands as 2 fp16 planes of operand / (power-of-two scale from a rigorous bound: "
                                   "BatchNorm outputs and gradients, weights), 3 v_mfma_f32_32x32x16_f16 partial products, "
                                   "fp32 accumulate (dropped term <= 2^-22 |x*y|; measured error vs fp64 = the fp32 matrix "
                                   "instruction's); the local correlation the same way (block-local scale for its window gradients); operands without a bound (eval mode) on 3 bf16 planes"}[math],
            "config": {"workload": f"1-way {args.shots}-shot, {args.size}x{args.size}, T={args.iters}, batch {args.batch}/GPU "
                                   f"(BASELINE configs[{(1 if world == 1 else 3) if args.shots == 1 else 2}]), train mode, align loss on, "
                                   "loss = dice_ce(output)+sum dice_ce(refinement)+align_loss",
                       "global_batch": world * args.batch, "parallelism": f"dp{world}", "conv_math": math,
                       "grad_allreduce_mb": round(bucket.numel * 4 / 1e6, 1)},
            "roofline": {"bound": "mfma", "kernel": "rpnet_conv_fwd launches (conv forward + dgrad): conv_igemm"
                                                    + ("_kernel" if math == "f32" else "_split*_kernel"),
                         "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "traffic": pmc_traffic(),
                         "peak_basis": "157.3 TF dense fp32 MFMA" if math == "f32" else
                                       f"2500 TF dense bf16 / fp16 MFMA / {products} partial products per fp32 multiply-add "
                                       "(achieved counts ALGORITHMIC fp32 FLOPs, not issued MFMA FLOPs)",
                         "issued_mfma_tflops": round(achieved * products, 1),
                         "traffic_unit": "HBM bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, offline pass)",
                         "algorithmic_bytes_per_launch": round(conv[3] / max(conv[0], 1)),
                         "launches_per_step": conv[0], "avg_launch_ms": round(1e3 * conv[1] / max(conv[0], 1), 4),
                         "algorithmic_gflop_per_step": round(conv[2] / 1e9, 1),
                         "wgrad_tflops": round(wg[2] / wg[1] / 1e12, 2),
                         "whole_step_frac": round(value / world * gf_pair * 1e9 / (peak * 1e12), 4),
                         "whole_step_tflops": round(value / world * gf_pair / 1e3, 1),
                         "gflop_per_pair": round(gf_pair, 1),
                         "kernel_time_share": {k: round(v[1] / kern_total, 4) for k, v in
                                               sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]},
                         "sum_kernel_ms_per_step": round(1e3 * kern_total, 2),
                         "note": "per-kernel figures from one extra step with the two HIP streams serialised; "
                                 "value/ms_per_step measured with async weight gradients on"},
        }
        if alt:
            result["alt_math"] = alt
        if world == 1 and not args.no_cpu_baseline:
            result["conv_math_error_vs_fp64"] = {k: float(f"{v:.3g}") for k, v in conv_accuracy_probe(dev).items()}
        if world == 1 and not args.no_cpu_baseline and args.shots == 1:
            result["cpu_baseline"] = cpu_baseline(cfg, args.size, args.iters, net=net if args.shots == 1 else None, bucket=bucket, dev=dev)
            result["cpu_baseline"]["gpu_over_cpu"] = round(value / result["cpu_baseline"]["value"], 1)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
