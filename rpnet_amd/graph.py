"""hipGraph replay of the evaluation forward.

The reference evaluates volumes in 2-slice batches with T = 10 refinement iterations
(test_rpnet.py:164,189-215): ~350 small-to-medium kernel launches per call, so the call is bound
by host launch latency, not by the GPU.  Every shape on that path is static per (batch, H, W), so
the whole `RP_Net.forward` (eval mode, no autograd) is captured once into a HIP graph and replayed:
inputs are copied into static buffers, one graph launch runs the pass, outputs are static tensors.
All kernels go through the same C ABI on the capturing stream; the library never allocates or
synchronises, which is what makes it capturable.  The weights are static for a captured graph, so the weight
packs and the folded BatchNorm affines are produced once in the warm-up and stay out of the replay
(`RP_Net.freeze_packs`).
"""
import torch


class GraphedEval:
    """`GraphedEval(net)(supp_imgs, fore_mask, back_mask, qry_imgs, appr_query_labels=...)` — same
    argument structure and output dict as `RP_Net.forward` for 1-way 1-shot; one graph per shape."""

    def __init__(self, net, warmup=2):
        self.net = net.eval()
        self.net.freeze_packs = True      # packs and folded BN affines are made in the warm-up, outside the graph
        self.net._cache.clear()
        self.warmup = warmup
        self._graphs = {}

    def _capture(self, key, si, fg, bg, qi, appr):
        static = [t.clone() for t in (si, fg, bg, qi, appr)]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():       # warm-up outside capture (allocator, lazy init)
            for _ in range(self.warmup):
                self.net([[static[0]]], [[static[1]]], [[static[2]]], [static[3]], appr_query_labels=static[4])
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g), torch.no_grad():
            out = self.net([[static[0]]], [[static[1]]], [[static[2]]], [static[3]], appr_query_labels=static[4])
        from . import functional as RF
        # the key of the captured forward's fp16 scale prediction (RF.pred_*): the replay checks ITS record, not another graph's
        self._graphs[key] = (g, static, out, RF.pred_last_key(si.device) if RF.f16_mode() else None)
        return self._graphs[key]

    def __call__(self, supp_imgs, fore_mask, back_mask, qry_imgs, registration_field=None, grid=None,
                 query_labels=None, appr_query_labels=None):
        if len(supp_imgs) != 1 or len(supp_imgs[0]) != 1 or len(qry_imgs) != 1:
            raise NotImplementedError("GraphedEval covers the 1-way 1-shot evaluation call")
        args = (supp_imgs[0][0].float(), fore_mask[0][0].float(), back_mask[0][0].float(), qry_imgs[0].float(),
                appr_query_labels.float())
        key = tuple(args[0].shape)
        entry = self._graphs.get(key) or self._capture(key, *args)
        g, static, out, pkey = entry
        for dst, src in zip(static, args):
            dst.copy_(src)
        g.replay()
        from . import functional as RF
        if pkey is not None and RF.pred_check_pending(args[0].device, pkey):
            # the captured forward runs on fp16 scales predicted from the previous call (RF.pred_*); a maximum above its
            # predicted bound invalidates this replay: redo the call eagerly on measured scales (the replay has already
            # turned the new maxima into the next replay's predictions)
            self.net._pred_redo = True
            try:
                with torch.no_grad():
                    return self.net(supp_imgs, fore_mask, back_mask, qry_imgs, appr_query_labels=appr_query_labels)
            finally:
                self.net._pred_redo = False
        return out


class GraphedTrainStep:
    """One training step (forward, loss, backward into the flat gradient bucket) captured into a HIP graph and replayed.

    The step of this network is ~470 kernel launches behind ~330 C-ABI calls and a Python autograd graph: the host needs
    10-11 ms to enqueue what the GPU runs in ~19 ms (tools/cpu_overhead.py) — no bound today, but every kernel speed-up
    eats the margin.  All shapes are static per (batch, H, W, T), the library neither allocates nor synchronises, the
    weight packs are made by kernels inside the step (so a replay re-packs the CURRENT weights), and the side streams of
    the asynchronous weight gradients fork from / join the capturing stream through events — so the whole step captures.
    Usage:
        g = GraphedTrainStep(net, bucket, loss_fn)          # loss_fn(out, labels) -> scalar tensor
        loss = g(supp_imgs, fore_mask, back_mask, qry_imgs, labels, appr_query_labels)   # replays; gradients in bucket.flat
    Inputs are copied into static buffers; the returned loss is a static tensor (clone it to keep it).  With a process
    group the captured step contains NO collective (the bucket's post-accumulate hooks are suspended while it is captured):
    the replay is followed by `bucket.allreduce()` — one all-reduce of the whole flat bucket on the collective library's
    stream, exposed behind the step instead of overlapped with backward (what eight eager Python processes on one shared
    host trade for one graph launch per step each).  `exposed`: a list that receives a HIP-event pair around that
    exchange (bench.py).  ORDER under a process group: call `capture(...)` BEFORE dist.init_process_group.  A stream capture in a
    process that already holds an RCCL communicator ended in a segmentation fault inside hipStreamEndCapture in 3 of 12 one-rank
    runs in round 4 (profiles/r04_graph_capture_under_rccl.txt; thread-local and global capture mode alike); a graph captured
    before the group exists and replayed beside it completed 12 of 12 runs with gradients equal to the eager step bit for bit
    (profiles/r05_graph_capture_order.txt) — bench.py and tests/test_gpu_dist.py use that order."""

    def __init__(self, net, bucket, loss_fn, warmup=2, exposed=None):
        self.net, self.bucket, self.loss_fn, self.warmup = net, bucket, loss_fn, warmup
        self.exposed = exposed
        self._graphs = {}

    @staticmethod
    def _flat(nested):
        return [t for way in nested for t in way]

    def _run(self, st):
        si, fg, bg, qi, ql, appr = st
        was = self.bucket.hooks_enabled
        self.bucket.hooks_enabled = False      # no collective inside the captured step
        try:
            self.bucket.zero()
            out = self.net(si, fg, bg, qi, appr_query_labels=appr)
            loss = self.loss_fn(out, ql)
            from . import functional as RF
            RF.backward(loss)                # (cached gradient seed: no fill launch in the captured step)
            RF.join_side_streams()           # the weight gradients of the side streams land before the graph ends
        finally:
            self.bucket.hooks_enabled = was
        return loss.detach()

    def _capture(self, key, args):
        si, fg, bg, qi, ql, appr = args
        st = ([[t.clone() for t in way] for way in si], [[t.clone() for t in way] for way in fg],
              [[t.clone() for t in way] for way in bg], [t.clone() for t in qi], ql.clone(), appr.clone())
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                        # warm-up outside capture (allocator, lazy init)
            for _ in range(self.warmup):
                self._run(st)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        # Under a process group on RCCL the collective library's watchdog THREAD polls its events (hipEventQuery) at any time;
        # in the default (global) capture mode that call from another thread is an error while this thread captures and takes
        # the process down ("operation not permitted when stream is capturing").  Thread-local mode restricts only this thread.
        # Under gloo (the one-GPU plumbing backend) neither mode is usable on this ROCm: thread-local mode ended three of five
        # captures in a segmentation fault inside hipStreamEndCapture, global mode hung five of five — refuse instead of crashing.
        import torch.distributed as dist
        ddp = dist.is_available() and dist.is_initialized()
        if ddp and dist.get_backend() != "nccl" and dist.get_world_size() > 1:
            raise RuntimeError("rpnet_amd.graph.GraphedTrainStep: HIP stream capture under a gloo process group is not supported "
                               "(gloo's helper threads touch the device while the step is captured); use the eager step or RCCL")
        mode = "thread_local" if ddp else "global"
        import os
        forced = os.environ.get("RPNET_GRAPH_CAPTURE_MODE")      # A/B: "global" | "thread_local" | "quiesce" (global after a pause)
        if forced == "quiesce" and ddp:
            # no collective in flight and the watchdog has seen the last one complete (it polls every 100 ms): nothing left for it
            # to query while the step is captured, so the global mode's prohibition cannot hit it
            import time
            torch.cuda.synchronize()
            time.sleep(0.5)
            mode = "global"
        elif forced in ("global", "thread_local"):
            mode = forced
        with torch.cuda.graph(g, capture_error_mode=mode):
            loss = self._run(st)
        self._graphs[key] = (g, st, loss)
        return self._graphs[key]

    @staticmethod
    def _key(net, supp_imgs, qry_imgs):
        return (len(supp_imgs), len(supp_imgs[0]), tuple(qry_imgs[0].shape), net.num_iter)

    def capture(self, supp_imgs, fore_mask, back_mask, qry_imgs, labels, appr_query_labels):
        """Capture the step for these shapes now (ahead of the first call: before a process group exists, see the class docstring)."""
        key = self._key(self.net, supp_imgs, qry_imgs)
        if key not in self._graphs:
            self._capture(key, (supp_imgs, fore_mask, back_mask, qry_imgs, labels, appr_query_labels))
        return self

    def __call__(self, supp_imgs, fore_mask, back_mask, qry_imgs, labels, appr_query_labels):
        args = (supp_imgs, fore_mask, back_mask, qry_imgs, labels, appr_query_labels)
        key = self._key(self.net, supp_imgs, qry_imgs)
        g, st, loss = self._graphs.get(key) or self._capture(key, args)
        for dn, sn in zip(st[:3], args[:3]):
            for dw, sw in zip(dn, sn):
                for d, s in zip(dw, sw):
                    d.copy_(s)
        st[3][0].copy_(qry_imgs[0])
        st[4].copy_(labels)
        st[5].copy_(appr_query_labels)
        g.replay()
        if self.bucket._active():            # the gradient exchange behind the replay (sum over ranks, 1 / world)
            self.bucket._work = {}
            if self.exposed is not None:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                self.bucket.allreduce()
                b.record()
                self.exposed.append((a, b))
            else:
                self.bucket.allreduce()
        return loss
