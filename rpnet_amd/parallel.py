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
    (no gather copy); `zero()` is one memset.  The exchange is split in two so that most of it
    hides under the backward pass: backward reaches the parameters in reverse module order
    (cre.* and the decoder half Up_conv4..Up5 first, Conv5..Conv1 last), so the TAIL of the buffer
    (everything from `split_at` on, ~46 % of the bytes) is final as soon as the first gradient of
    the HEAD has been accumulated — a post-accumulate hook on that parameter launches the tail's
    all-reduce asynchronously (RCCL stream) while the rest of backward still runs; `allreduce()`
    then only exposes the head's all-reduce.  Both are plain sums over identical buffers on every
    rank, so results do not depend on the overlap.
    """

    def __init__(self, module, skip_prefixes=UNUSED_PREFIXES, split_at="encoder.Up5."):
        self.params = [(n, p) for n, p in module.named_parameters()
                       if p.requires_grad and not n.startswith(tuple(skip_prefixes))]
        total = sum(p.numel() for _, p in self.params)
        p0 = self.params[0][1]
        self.flat = torch.zeros(total, device=p0.device, dtype=torch.float32)
        off, self.split = 0, None
        for n, p in self.params:
            if self.split is None and split_at and n.startswith(split_at):
                self.split = off
            k = p.numel()
            p.grad = self.flat[off:off + k].view_as(p)
            off += k
        self.numel = total
        self._tail_work = None
        self._hook = None
        if self.split:   # the last head parameter in module order is the FIRST head gradient backward produces
            head_last = None
            o = 0
            for n, p in self.params:
                if o + p.numel() <= self.split:
                    head_last = p
                o += p.numel()
            if head_last is not None and hasattr(head_last, "register_post_accumulate_grad_hook"):
                head_last._rpnet_autograd_grad = True    # keep this one on AccumulateGrad so that the hook fires
                self._hook = head_last.register_post_accumulate_grad_hook(self._launch_tail)

    def _active(self):
        return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    def _launch_tail(self, _param=None):
        if self._active() and self._tail_work is None and self.split:
            from .functional import join_side_streams
            join_side_streams()      # async weight gradients of the tail must have landed in the bucket
            self._tail_work = dist.all_reduce(self.flat[self.split:], op=dist.ReduceOp.SUM, async_op=True)

    def zero(self):
        self._tail_work = None
        self.flat.zero_()

    def allreduce(self, async_op=False):
        """sum over ranks, then 1/world (mean gradient).  No-op without a process group."""
        if not self._active():
            return None
        from .functional import join_side_streams
        join_side_streams()
        if self.split and self._tail_work is not None:     # tail already in flight (or done): only the head remains
            dist.all_reduce(self.flat[:self.split], op=dist.ReduceOp.SUM)
            self._tail_work.wait()
            self._tail_work = None
        else:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
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
