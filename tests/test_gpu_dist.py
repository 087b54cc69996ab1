"""Data-parallel path on the GPU: TWO ranks of the real RP_Net on one MI355X (both on cuda:0, gloo process group — RCCL
refuses two ranks on one device; the collective's arithmetic is the same sum), with everything that runs in the
multi-GPU bench switched on together: async weight gradients accumulated on side streams straight into the flat bucket,
the three-segment exchange launched from post-accumulate hooks during backward, join_side_streams in front of every
segment (the reference has no distributed code, test_rpnet.py:16 imports data_parallel and never uses it)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIZE, T, GLOBAL_B = 64, 2, 4


def _setup_model(force_active=None, T_=None):
    import rpnet_amd.functional as RF
    import rpnet_amd.modules as RM
    from rpnet_amd.modules import RP_Net
    from rpnet_amd.parallel import FlatGradBucket
    from rpnet_amd.utils.seeding import seed_module_
    from tests.helpers import load_cfg
    RM._F16_MIN_PIXELS = 0
    RF.set_conv_math("f16x2")
    RF.set_async_wgrad(True)
    cfg = load_cfg(T_ or T)
    net = RP_Net(cfg={"align": True, "backbone": "UNet"}, backbone_cfg=cfg).to("cuda:0")
    seed_module_(net)
    net.train()
    return net, FlatGradBucket(net, force_active=force_active), cfg


def _shard_step(net, bucket, cfg, lo, hi):
    from rpnet_amd.functional import dice_ce
    from tests.helpers import episode_tensors
    (si, fg, bg, qi, ql, appr), _ = episode_tensors(909, GLOBAL_B, SIZE, "cuda:0")
    sl = slice(lo, hi)
    bucket.zero()
    out = net([[si[0][0][sl]]], [[fg[0][0][sl]]], [[bg[0][0][sl]]], [qi[0][sl]], appr_query_labels=appr[sl])
    loss = dice_ce(out["output"], ql[sl])
    for v in out["refinement"].values():
        loss = loss + dice_ce(v, ql[sl])
    loss = loss + cfg["align_loss_scaler"] * out["align_loss"]
    loss.backward()
    return loss


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rpnet_amd.parallel import broadcast_parameters, shard_episodes
    net, bucket, cfg = _setup_model()
    broadcast_parameters(net)
    assert len(bucket.bounds) == 4 and len(bucket._hooks) == 2          # three segments, two hooks
    lo, hi = shard_episodes(GLOBAL_B, rank, world)
    _shard_step(net, bucket, cfg, lo, hi)
    launched = sorted(bucket._work)          # segments that went out from the hooks while backward was still running
    bucket.allreduce()
    torch.cuda.synchronize()
    q.put((rank, launched, bucket.flat.cpu().numpy()))       # by value: the worker may exit before the parent reads
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_rp_net_bucket_on_one_gpu():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=600) for _ in range(2)], key=lambda t: t[0])
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    (_, l0, f0), (_, l1, f1) = res
    f0, f1 = torch.from_numpy(f0), torch.from_numpy(f1)
    assert l0 == [1, 2] and l1 == [1, 2], (l0, l1)       # (c) decoder + cre and Conv5 segments left during backward
    assert torch.equal(f0, f1)                           # (a) both ranks hold the same averaged gradients
    # (b) = the mean of two single-process runs of the same shards (per-rank BatchNorm statistics, as in the exchange)
    from rpnet_amd.parallel import shard_episodes
    import rpnet_amd.functional as RF
    import rpnet_amd.modules as RM
    old = (RF.conv_math(), RM._F16_MIN_PIXELS, RF._ASYNC["on"])
    try:
        flats = []
        for r in range(2):
            net, bucket, cfg = _setup_model()
            _shard_step(net, bucket, cfg, *shard_episodes(GLOBAL_B, r, 2))
            bucket.allreduce()                           # no process group: joins the side streams only
            torch.cuda.synchronize()
            flats.append(bucket.flat.cpu().double())
    finally:
        RF.set_conv_math(old[0])
        RM._F16_MIN_PIXELS = old[1]
        RF.set_async_wgrad(old[2])
    want = (flats[0] + flats[1]) / 2
    err = float((f0.double() - want).abs().max() / want.abs().max())
    assert err <= 1e-6, err
    assert float(want.abs().max()) > 0


def _nccl_world1_worker(port, q, ways, size, B, T_, reps=5, greps=3, bench_objective=False):
    """ONE rank, backend nccl (= RCCL) on cuda:0, the bucket's exchange forced on: the real step — async weight gradients on
    the side streams, the CRE's second branch (and for 2-way the encoder's second chain) on their own stream, the
    three-segment exchange launched from post-accumulate hooks in autograd's thread — meets RCCL's stream semantics (the
    collective runs on RCCL's own stream behind an event recorded on the caller's CURRENT stream at call time; work.wait()
    orders the compute stream behind it).  A sum over one rank times 1/1 changes no bit, so the bucket must EQUAL the
    non-distributed step's, every time (five repetitions: a missing stream dependency is a race, not a constant)."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    import rpnet_amd.functional as RF
    import rpnet_amd.modules as RM
    from rpnet_amd.functional import dice_ce
    from rpnet_amd.graph import GraphedTrainStep
    from tests.helpers import episode_tensors
    net, bucket, cfg = _setup_model(force_active=True, T_=T_)
    assert RM._CRE_STREAMS_TRAIN and RF._ASYNC["on"] and RF._WGRAD_DEFER >= 1
    (si, fg, bg, qi, ql, appr), _ = episode_tensors(910, B, size, "cuda:0", n_ways=ways)

    def loss_fn(out, lab):
        if bench_objective:          # bench.py's / train_rpnet.py's form of the same objective (one launch pair)
            return RF.objective([out["output"], *out["refinement"].values()], lab, out["align_loss"], cfg["align_loss_scaler"])
        loss = dice_ce(out["output"], lab)
        for v in out["refinement"].values():
            loss = loss + dice_ce(v, lab)
        return loss + cfg["align_loss_scaler"] * out["align_loss"]

    # The HIP-graph form of the step is captured BEFORE the process group exists (no communicator, no watchdog thread yet):
    # a capture beside a live RCCL communicator ended in a segmentation fault inside hipStreamEndCapture in 3 of 12 runs
    # (profiles/r04_graph_capture_under_rccl.txt); captured first and replayed beside the communicator it completed 12 of 12
    # (profiles/r05_graph_capture_order.txt).  No collective is inside the capture: the exchange follows the replay.
    gts = GraphedTrainStep(net, bucket, loss_fn).capture(si, fg, bg, qi, ql, appr)
    torch.cuda.synchronize()
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    assert dist.get_backend() == "nccl"

    def one(active):
        bucket.force_active = active
        bucket.zero()
        loss_fn(net(si, fg, bg, qi, appr_query_labels=appr), ql).backward()
        launched = sorted(bucket._work)
        bucket.allreduce()
        torch.cuda.synchronize()
        return launched, bucket.flat.clone()

    one(False)                                   # warm-up (allocator, lazy initialisation)
    _, want = one(False)
    _, again = one(False)
    res = {"deterministic": bool(torch.equal(want, again)), "nonzero": float(want.abs().max()) > 0, "equal": [], "launched": []}
    for _ in range(reps):
        launched, got = one(True)
        res["launched"].append(launched)
        res["equal"].append(bool(torch.equal(got, want)))
    ones = torch.ones(4, device="cuda:0")
    dist.all_reduce(ones)
    res["ranks_seen"] = float(ones[0])
    q.put(dict(res, graph_equal=None))           # the eager result first: a crash below must not take it along
    bucket.force_active = True
    res["graph_equal"] = []
    for _ in range(greps):                       # replay + one all-reduce of the whole bucket behind it
        gts(si, fg, bg, qi, ql, appr)
        torch.cuda.synchronize()
        res["graph_equal"].append(bool(torch.equal(bucket.flat, want)))
    q.put(res)
    dist.destroy_process_group()


@pytest.mark.parametrize("ways,size,B,T_", [(1, 128, 4, 2), (2, 64, 2, 2)])
def test_rccl_world1_bucket_equals_plain_step(ways, size, B, T_):
    """RCCL's first contact with the side streams (one GPU is enough: a process group of one rank, FlatGradBucket(force_active=True))."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_world1_worker, args=(39500 + os.getpid() % 2000 + ways, q, ways, size, B, T_))
    p.start()
    res = q.get(timeout=900)
    assert res["deterministic"] and res["nonzero"], res
    assert all(l == [1, 2] for l in res["launched"]), res      # both early segments left from the hooks, during backward
    assert all(res["equal"]), res
    res = q.get(timeout=900)                     # the second message carries the graph-replay part
    p.join(120)
    assert p.exitcode == 0
    assert res["graph_equal"] and all(res["graph_equal"]), res
    assert res["ranks_seen"] == 1.0


def test_rccl_world1_canary_at_the_benched_size():
    """The canary the round-5 review asked for: BASELINE configs[1] at FULL size (batch 8, 256 x 256, T = 5, bench.py's objective) under a
    forced one-rank RCCL group — the three bucket segments launched from the hooks while the LDS-DMA kernels of the backward pass (and
    the guarded pooled BatchNorm passes) run, RCCL's own kernels sharing the CUs with them — 30 times, every bucket bit-identical to
    the non-distributed step's; then the pre-captured HIP-graph replay + ONE all-reduce, 30 times.  The one silent-corruption fault
    this code base has had (profiles/r05_pool_fault_repro.txt) was a co-residence fault; this is the schedule an 8-GPU job runs."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_world1_worker, args=(41500 + os.getpid() % 2000, q, 1, 256, 8, 5, 30, 30, True))
    p.start()
    res = q.get(timeout=1200)
    assert res["deterministic"] and res["nonzero"], res
    assert len(res["equal"]) == 30 and all(l == [1, 2] for l in res["launched"]), res
    assert all(res["equal"]), res
    res = q.get(timeout=1200)
    p.join(120)
    assert p.exitcode == 0
    assert len(res["graph_equal"]) == 30 and all(res["graph_equal"]), res
    assert res["ranks_seen"] == 1.0


def test_bench_forced_rccl_group_on_one_gpu():
    """bench.py with RPNET_BENCH_FORCE_DIST=1: a one-rank RCCL group, the exchange forced on — the line's `distributed` object
    is then produced by the N > 1 code (backend nccl, exposed time by HIP events)."""
    env = dict(os.environ, RPNET_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("RPNET_DIST_BACKEND", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--batch", "2", "--size", "128",
           "--iters", "2", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    d = res["distributed"]
    assert d["backend"].startswith("nccl") and d["rccl_ranks_seen"] == 1 and d["allreduce_exposed_ms"] >= 0
    # the replay form is captured BEFORE the group is created (profiles/r05_graph_capture_order.txt) and measured beside the
    # eager step; the line says which of the two it was issued as and carries the other
    assert d["issued"] in ("eager", "hip_graph_replay")
    other = d["other_issue_mode"]
    assert other is not None and {d["issued"], other["issued"]} == {"eager", "hip_graph_replay"} and other["ms_per_step"] > 0, d


def test_bench_two_ranks_on_one_gpu():
    """bench.py's N > 1 code path end to end (SELF-LAUNCHED: `python bench.py --gpus 2` with no rendezvous in the environment
    starts its own ranks; sharded seeds, barrier-fenced timing, max over ranks, every rank in the profiled step's
    collective) with two gloo ranks on the one GPU; small shapes."""
    env = dict(os.environ, RPNET_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--batch", "2", "--size", "128", "--iters", "2", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["steps"] == 2 and res["value"] > 0 and res["scaling"] == "weak"
    assert res["config"]["global_batch"] == 4 and res["config"]["parallelism"] == "dp2"
    # what a multi-GPU line must carry so that "did the collective library see N ranks" can be answered from the line
    dinfo = res["distributed"]
    assert dinfo["rccl_ranks_seen"] == 2 and dinfo["world_size"] == 2 and dinfo["backend"].startswith("gloo")
    assert dinfo["allreduce_exposed_ms"] >= 0 and len(dinfo["bucket_segments_mb"]) == 3
    # eager + overlapped exchange here; the graph-replay form is measured under RCCL only (test_bench_forced_rccl_group_on_one_gpu):
    # a HIP stream capture beside gloo's helper threads crashes or hangs on this ROCm (bench.py)
    assert dinfo["issued"] == "eager" and dinfo["other_issue_mode"] is None


def _train_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import rpnet_amd.modules as RM
    from tests.helpers import load_cfg
    from train_rpnet import train
    RM._F16_MIN_PIXELS = 0
    cfg = load_cfg(2)
    torch.manual_seed(rank)          # different seeds: only the broadcast makes the replicas equal
    net, hist = train(cfg, steps=4, batch=2, size=64, dev=torch.device("cuda:0"), lr=1e-3, log_every=0, seed=7)
    torch.cuda.synchronize()
    flat = torch.cat([p.detach().flatten() for p in net.parameters()]).cpu().numpy()
    q.put((rank, hist, flat))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_training_driver_on_one_gpu():
    """train_rpnet.py under two ranks (gloo, both on the one GPU): four Adam steps on per-rank episodes with the flat-bucket
    exchange -> both replicas hold bit-identical parameters afterwards (and they moved), the per-rank losses differ
    (different episodes)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 37500 + os.getpid() % 2000
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=900) for _ in range(2)], key=lambda t: t[0])
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    (_, h0, f0), (_, h1, f1) = res
    f0, f1 = torch.from_numpy(f0), torch.from_numpy(f1)
    assert torch.equal(f0, f1)
    assert all(map(lambda v: v == v, h0 + h1)) and h0 != h1
    from rpnet_amd.modules import RP_Net
    from tests.helpers import load_cfg
    torch.manual_seed(0)             # rank 0's initial parameters (what the broadcast replicated)
    ref = RP_Net(cfg={"align": True, "backbone": "UNet"}, backbone_cfg=load_cfg(2))
    start = torch.cat([p.detach().flatten() for p in ref.parameters()])
    assert float((f0 - start).abs().max()) > 1e-4       # the optimizer stepped (Adam, lr 1e-3, four steps)
