#!/usr/bin/env python3
"""Training driver for RP-Net on the MI355X path.

The reference has NO train script (README "Train" section is empty); this driver
follows the hyper-parameters its yaml carries (yamls/example.yml:64-67,105-116: Adam, init_lr 1e-5,
weight_decay 1e-4, StepLR every `scheduler_step` epochs, n_iter_refinement = 4, loss dice_ce,
align_loss_scaler) and writes checkpoints in the format test_rpnet.py loads
(`{'epoch', 'state_dict'}`, test_rpnet.py:86-94).  Episodes come from the synthetic reader unless
real data is wired in (§8f.4).  One process per GPU; gradients are exchanged through the flat
bucket (RCCL all-reduce); BatchNorm statistics stay per rank like the reference (no SyncBN).

    python train_rpnet.py --yaml yamls/example.yml --steps 100
    tools/launch_ddp.sh 8 train --steps 100        (= python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 ...)
"""
import argparse
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from net.model import model_factory  # noqa: E402
import rpnet_amd.functional as RF  # noqa: E402
from rpnet_amd.functional import dice_ce  # noqa: E402
from rpnet_amd.parallel import FlatGradBucket, broadcast_parameters  # noqa: E402
from rpnet_amd.utils.synth import make_episode  # noqa: E402
from utils.util import load_yaml  # noqa: E402


def objective(out, labels, scaler):
    # dice_ce of the final output and of every refinement iteration's output, summed, + scaler * align_loss (on the GPU: ONE launch pair,
    # rpnet_amd.functional.objective -> rpnet_objective_fwd; the reference ships no training loop: this is what its paper describes)
    terms = [out["output"], *out["refinement"].values()]
    if terms[0].is_cuda:
        return RF.objective(terms, labels, out["align_loss"], scaler)
    loss = dice_ce(terms[0], labels)
    for v in terms[1:]:
        loss = loss + dice_ce(v, labels)
    return loss + scaler * out["align_loss"]


def episode(seed, batch, size, dev):
    ep = make_episode(seed, batch, size)
    t = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
    return ([[t(ep["support_images"][0][0])]], [[t(ep["support_fg"][0][0])]], [[t(ep["support_bg"][0][0])]],
            [t(ep["query_images"])], t(ep["query_labels"]), t(ep["appr_query_labels"]))


def train(config, steps, batch, size, dev, lr=None, log_every=10, out_dir=None, seed=0, steps_per_epoch=50, n_ways=1, n_shots=1):
    rank = dist.get_rank() if dist.is_initialized() else 0
    net = model_factory[config.get("net", "RP_Net")](pretrained_path=config.get("pretrained_path"),
                                                    cfg={"align": True, "backbone": config.get("backbone", "UNet")},
                                                    backbone_cfg=config).to(dev)
    broadcast_parameters(net)
    net.train()
    bucket = FlatGradBucket(net)
    import rpnet_amd.functional as RF
    RF.set_async_wgrad(True)             # weight gradients on a second stream, straight into the bucket
    caller_stream = None
    if torch.device(dev).type == "cuda":
        caller_stream = torch.cuda.current_stream(dev)
    params = [p for _, p in bucket.params]
    opt = torch.optim.Adam(params, lr=lr if lr is not None else config["init_lr"], weight_decay=config["weight_decay"])
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=config["scheduler_step"])
    scaler = config["align_loss_scaler"]
    history, t0 = [], time.time()
    # synthetic episodes are generated on the host (numpy): a few steps ahead, in worker threads, so that the GPU
    # step (~32 ms at batch 8) is not waiting for the generator (~70 ms)
    import concurrent.futures
    pool = concurrent.futures.ThreadPoolExecutor(max_workers=4)
    gen = lambda k: pool.submit(make_episode, seed + 1000 * rank + k, batch, size, n_shots=n_shots, n_ways=n_ways)  # noqa: E731
    pending = [gen(k) for k in range(min(4, steps))]

    def to_dev(ep):
        t = lambda a: torch.from_numpy(a).to(dev, non_blocking=True)  # noqa: E731
        return ([[t(s) for s in way] for way in ep["support_images"]], [[t(s) for s in way] for way in ep["support_fg"]],
                [[t(s) for s in way] for way in ep["support_bg"]], [t(ep["query_images"])], t(ep["query_labels"]),
                t(ep["appr_query_labels"]))

    for it in range(steps):
        si, fg, bg, qi, ql, appr = to_dev(pending.pop(0).result())
        if it + 4 < steps:
            pending.append(gen(it + 4))
        bucket.zero()                       # gradients live in the flat bucket: one memset instead of zero_grad
        out = net(si, fg, bg, qi, appr_query_labels=appr)
        loss = objective(out, ql, scaler)
        RF.backward(loss)                   # (cached gradient seed)
        bucket.allreduce()
        opt.step()
        history.append(loss.detach())       # no host sync per step
        if (it + 1) % steps_per_epoch == 0:
            sched.step()
            epoch = (it + 1) // steps_per_epoch
            if out_dir and rank == 0 and epoch % config.get("epoch_save", 1) == 0:
                os.makedirs(out_dir, exist_ok=True)
                torch.save({"epoch": epoch, "state_dict": net.state_dict()}, os.path.join(out_dir, f"{epoch:03d}.ckpt"))
        if rank == 0 and log_every and (it + 1) % log_every == 0:
            print(f"step {it + 1:5d}  loss {float(history[-1]):.4f}  ({(time.time() - t0) / (it + 1) * 1e3:.0f} ms/step)", flush=True)
    pool.shutdown(wait=False)
    history = [float(v) for v in history]
    if caller_stream is not None:        # hand the thread back on the stream it came with
        torch.cuda.current_stream(dev).synchronize()
        torch.cuda.set_stream(caller_stream)
    return net, history


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--yaml", default=os.path.join(ROOT, "yamls", "example.yml"))
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--lr", type=float, default=None)
    ap.add_argument("--out_dir", default=None)
    a = ap.parse_args()
    config, _ = load_yaml(a.yaml)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n_dev = torch.cuda.device_count()
    if n_dev == 0:
        raise SystemExit("train_rpnet.py needs MI355X GPUs (the HIP path has no CPU fallback)")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("RPNET_DIST_BACKEND", "nccl") == "nccl":
        if local_rank >= n_dev:      # RCCL hangs / fails with a duplicate-device error when two ranks share a GPU
            raise SystemExit(f"LOCAL_RANK {local_rank} but {n_dev} GPU(s): RCCL needs one device per rank "
                             "(RPNET_DIST_BACKEND=gloo: several ranks per device, plumbing tests only)")
    else:
        local_rank %= n_dev
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("RPNET_DIST_BACKEND", "nccl")      # "nccl" IS RCCL on ROCm; gloo for one-GPU plumbing tests
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    train(config, a.steps, a.batch or config["batch_size"], a.size, dev, lr=a.lr, out_dir=a.out_dir or config.get("out_dir"))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
