"""Data-parallel gradient exchange: one process per GPU, episodes sharded across ranks,
replicated weights, ONE flat fp32 gradient buffer all-reduced per step (RCCL over xGMI
when the process group is "nccl"; gloo in the CPU tests).

The reference has no distributed code at all: episodes of RP_Net.forward
never mix (net/rp_net.py:287,320), BatchNorm statistics stay per rank (no SyncBN in the
reference), so the only exchange is the gradient sum.  Parameters that never receive a
gradient (cre.w_context.*, cre.out.*: constructed but unused, net/rp_net.py:60-64,70-74)
are left out of the buffer: 34 808 000 of the 34 972 800 parameters travel (139.2 MB).
"""
import torch
import torch.distributed as dist

UNUSED_PREFIXES = ("cre.w_context.", "cre.out.")


class FlatGradBucket:
    """Gradients of `module` as views into one contiguous buffer.

    p.grad is pre-set to a view of the flat buffer, so autograd accumulates straight into it
    (no gather copy); `zero()` is one memset, `allreduce()` is one collective + one scale.
    """

    def __init__(self, module, skip_prefixes=UNUSED_PREFIXES):
        self.params = [(n, p) for n, p in module.named_parameters()
                       if p.requires_grad and not n.startswith(tuple(skip_prefixes))]
        total = sum(p.numel() for _, p in self.params)
        p0 = self.params[0][1]
        self.flat = torch.zeros(total, device=p0.device, dtype=torch.float32)
        off = 0
        for _, p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            off += n
        self.numel = total

    def zero(self):
        self.flat.zero_()

    def allreduce(self, async_op=False):
        """sum over ranks, then 1/world (mean gradient).  No-op without a process group."""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return None
        world = dist.get_world_size()
        work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=async_op)
        if async_op:
            return work
        self.flat.mul_(1.0 / world)
        return None

    def finish(self, work):
        if work is not None:
            work.wait()
            self.flat.mul_(1.0 / dist.get_world_size())


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
