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
        self._graphs[key] = (g, static, out)
        return self._graphs[key]

    def __call__(self, supp_imgs, fore_mask, back_mask, qry_imgs, registration_field=None, grid=None,
                 query_labels=None, appr_query_labels=None):
        if len(supp_imgs) != 1 or len(supp_imgs[0]) != 1 or len(qry_imgs) != 1:
            raise NotImplementedError("GraphedEval covers the 1-way 1-shot evaluation call")
        args = (supp_imgs[0][0].float(), fore_mask[0][0].float(), back_mask[0][0].float(), qry_imgs[0].float(),
                appr_query_labels.float())
        key = tuple(args[0].shape)
        entry = self._graphs.get(key) or self._capture(key, *args)
        g, static, out = entry
        for dst, src in zip(static, args):
            dst.copy_(src)
        g.replay()
        from . import functional as RF
        if RF.pred_check_pending(args[0].device):
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
