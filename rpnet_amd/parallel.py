"""Data-parallel gradient exchange: one process per GPU, episodes sharded across ranks,
replicated weights, ONE flat fp32 gradient buffer all-reduced per step (RCCL over xGMI
when the process group is "nccl"; gloo in the CPU tests).

The reference has no distributed code at all: episodes of RP_Net.forward
never mix (net/rp_net.py:287,320), BatchNorm statistics stay per rank (no SyncBN in the
reference), so the only exchange is the gradient sum.  Parameters that never receive a
gradient (cre.w_context.*, cre.out.*: constructed but unused, net/rp_net.py:60-64,70-74)
are left out of the buffer: 34 808 000 of the 34 972 800 parameters travel (139.2 MB).
"""
import os

import torch
import torch.distributed as dist

UNUSED_PREFIXES = ("cre.w_context.", "cre.out.")


class FlatGradBucket:
    """Gradients of `module` as views into one contiguous buffer.

    p.grad is pre-set to a view of the flat buffer, so autograd accumulates straight into it
    (no gather copy); `zero()` is one memset.  The exchange is cut into SEGMENTS at the `split_at` name
    prefixes so that most of it hides under the backward pass: backward reaches the parameters in reverse
    module order (cre.* and the decoder half Up_conv4..Up5 first, then Conv5, then Conv4..Conv1), so a
    segment is final as soon as the first gradient of the segment in front of it has been accumulated — a
    post-accumulate hook on that parameter launches the finished segment's all-reduce asynchronously (RCCL
    stream) while the rest of backward still runs.  With the default cuts (Conv5 = 41 % of the bytes, the
    decoder half + cre = 46 %) only the Conv1..Conv4 segment (13 %, 18 MB) is exchanged after backward.
    All are plain sums over identical buffers on every rank, so results do not depend on the overlap.
    """

    def __init__(self, module, skip_prefixes=UNUSED_PREFIXES, split_at=("encoder.Conv5.", "encoder.Up5."), force_active=None):
        """force_active: run the exchange code (hooks, segments, collectives, the 1/world scaling) also in a process group of
        ONE rank — the RCCL path (backend "nccl": its own stream, the event it records on the caller's current stream at call
        time, work.wait() back onto the compute stream) can then be exercised on a single GPU, where its result must equal
        the non-distributed step bit for bit (tests/test_gpu_dist.py); default: the RPNET_BUCKET_FORCE environment switch"""
        if isinstance(split_at, str):
            split_at = (split_at,)
        self.force_active = (os.environ.get("RPNET_BUCKET_FORCE", "0") == "1") if force_active is None else bool(force_active)
        # False while a step is being captured into a HIP graph (rpnet_amd.graph.GraphedTrainStep): the hooks then launch
        # nothing, the caller exchanges the whole bucket after the replay
        self.hooks_enabled = True
        self.params = [(n, p) for n, p in module.named_parameters()
                       if p.requires_grad and not n.startswith(tuple(skip_prefixes))]
        total = sum(p.numel() for _, p in self.params)
        p0 = self.params[0][1]
        self.flat = torch.zeros(total, device=p0.device, dtype=torch.float32)
        cuts, want = [], list(split_at or ())
        off = 0
        for n, p in self.params:
            if want and n.startswith(want[0]):
                if off > 0:
                    cuts.append(off)
                want.pop(0)
            k = p.numel()
            p.grad = self.flat[off:off + k].view_as(p)
            off += k
        self.numel = total
        self.cuts = cuts                                  # ascending offsets; segment i = [bounds[i], bounds[i+1])
        self.bounds = [0] + cuts + [total]
        self.split = cuts[-1] if cuts else None           # start of the last (first finished) segment
        self._work = {}                                   # segment index -> in-flight all-reduce
        self._hooks = []
        self._hook = None
        self._tail_work = None
        self._ensure_hooks()

    def _ensure_hooks(self):
        """Register the post-accumulate hooks that launch a finished segment's all-reduce — only once there IS an exchange (a process
        group of more than one rank, or force_active): a hook keeps its parameter's AccumulateGrad node alive from the stream it was
        registered on, which a single-GPU run that later steps on another stream (graph capture warm-up, the serialised profiling
        step) reports as "AccumulateGrad node's stream does not match" and pays for with a synchronisation."""
        if self._hooks or not self._active():
            return
        # the last parameter (module order) of segment i-1 is the FIRST gradient of that segment backward produces:
        # when it lands, segment i (everything behind it) is complete
        for i in range(1, len(self.bounds) - 1):
            last, o = None, 0
            for n, p in self.params:
                if o + p.numel() <= self.bounds[i]:
                    last = p
                o += p.numel()
            if last is not None and hasattr(last, "register_post_accumulate_grad_hook"):
                last._rpnet_autograd_grad = self          # keep this one on AccumulateGrad so that the hook fires (while wants_hooks())
                self._hooks.append(last.register_post_accumulate_grad_hook(self._launcher(i)))
        self._hook = self._hooks[-1] if self._hooks else None

    def wants_hooks(self):
        """True while a tagged parameter's gradient has to pass AccumulateGrad for its hook to launch an exchange (functional._direct)"""
        return self._active() and self.hooks_enabled

    def _active(self):
        return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or self.force_active)

    def _launcher(self, seg):
        def launch(_param=None):
            # segments finish back to front; launch this one and any later one that has not gone out yet
            if not self._active() or not self.hooks_enabled:
                return
            from .functional import join_side_streams
            join_side_streams(final=False)      # async weight gradients of the segment must have landed in the bucket
            for s in range(len(self.bounds) - 2, seg - 1, -1):
                if s not in self._work:
                    self._work[s] = dist.all_reduce(self.flat[self.bounds[s]:self.bounds[s + 1]], op=dist.ReduceOp.SUM,
                                                    async_op=True)
            self._tail_work = self._work.get(len(self.bounds) - 2)
        return launch

    def _launch_tail(self, _param=None):
        if len(self.bounds) > 2:
            self._launcher(len(self.bounds) - 2)()

    def zero(self):
        from .functional import reset_async
        reset_async()                # no side-stream accumulation may still be in flight into the buffer
        self._ensure_hooks()         # (a process group created, or force_active set, after construction)
        self._work = {}
        self._tail_work = None
        self.flat.zero_()

    def allreduce(self, async_op=False):
        """sum over ranks, then 1/world (mean gradient).  No-op without a process group."""
        from .functional import join_side_streams
        join_side_streams()          # also in single-process runs: the optimizer step reads the bucket next
        if not self._active():
            return None
        if not self._work:                                # nothing went out during backward (a replayed HIP graph): ONE collective
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        else:
            for s in range(len(self.bounds) - 1):         # whatever has not been launched during backward
                if s not in self._work:
                    dist.all_reduce(self.flat[self.bounds[s]:self.bounds[s + 1]], op=dist.ReduceOp.SUM)
        for w in self._work.values():
            w.wait()
        self._work = {}
        self._tail_work = None
        self.flat.mul_(1.0 / dist.get_world_size())
        return None


def shard_episodes(n_global, rank, world):
    """Contiguous, even split of the global batch of episodes; remainder to the low ranks."""
    base, rem = divmod(n_global, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_parameters(module, src=0):
    """Replicate rank `src`'s parameters and buffers (BN running stats) to all ranks."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src)
