"""Autograd bindings of the HIP kernels (host side of the C ABI).

Activations are fp32 NHWC tensors [N, H, W, C]; the module-boundary tensors of the
reference (images, masks, logits) keep their NCHW layout.  Every Function calls
librpnet_hip.so through rpnet_amd.hip — there is no torch-operator fallback.

The 3x3 convolutions (forward, input and weight gradients) and the local correlation run by default on split 16-bit
operands (set_conv_math / RPNET_CONV_MATH, see _MATH below): the kernel that produces a tensor also writes it as two
fp16 planes of tensor / scale (scale from a rigorous bound) or, where no bound exists, as three exact bf16 planes, which
is what the next convolution's operand loads read.  Those forms travel between layers in an explicit `Operand` (the fp32
tensor, its planes, its tensor scale) — never as attributes on tensors, which any view or reshape would drop silently.
"""
import collections
import ctypes as C
import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import hip
from .hip import ConvDesc, call, ptr, query

BN_EPS = 1e-5
BN_MOMENTUM = 0.1

# ---------------------------------------------------------------- async weight gradients
# Opt-in (bench.py / train_rpnet.py): the weight gradient of a conv only needs dy and the saved
# input, not the dgrad chain, so it is launched on a second HIP stream and accumulated straight
# into the parameter's existing .grad buffer (the flat bucket) instead of being returned to
# autograd; the backward pass's HBM-bound kernels then run beside MFMA work.  The two streams are
# joined by an engine callback at the end of backward (and before the bucket's early all-reduce).
_ASYNC = {"on": False, "side": {}, "pending": set(), "queued": False, "fifo": [], "keep": []}
# Tensors an async weight-gradient launch reads on the side streams (dy, its planes, the saved input, the operand scales,
# the split-K workspace) are KEPT ALIVE until the streams are joined at the end of the backward pass (join_side_streams)
# instead of being handed to the caching allocator's record_stream: every record costs an event record at free time and
# event queries at later allocations (~10 tensors x 27 launches per step: 9.7 -> 8.4 ms of host time per step, round 4), and
# 288 GB of HBM hold a backward pass's gradients without noticing (bench.py reports the step's peak allocation).

# Arithmetic of the 3x3 convolutions (forward, dgrad, wgrad): "f32" = v_mfma_f32_32x32x2_f32 on fp32 operands;
# "bf16x3" = every fp32 operand carried as three bf16 planes (exact split) and multiplied with six
# v_mfma_f32_32x32x16_bf16 partial products into fp32 accumulators (dropped terms <= 2^-23 |x*y|: fp32
# round-off level, 16/6 of the fp32 matrix rate); "f16x2" = two FP16 planes of operand / (power-of-two scale) and
# three v_mfma_f32_32x32x16_f16 products (dropped term 2^-22 |x*y|, 16/3 of the fp32 matrix rate) wherever a
# rigorous bound of the operand gives a scale that cannot overflow fp16 — the train-mode conv + BatchNorm layers:
# BatchNorm outputs, BatchNorm gradients, weights — and bf16x3 everywhere else (eval mode, vgg, correlation).
# "f16" = ONE fp16 plane of operand / scale (plain fp16 operands, fp32 accumulation, fp32 BatchNorm statistics): the
# reduced-precision arithmetic of BASELINE configs[4]; NOT fp32-equivalent (tolerances: DESIGN.md §6).
_MODES = {"f32": (0, False, 0), "bf16x3": (3, False, 0), "f16x2": (3, True, 2), "f16": (3, True, 1)}
_CORR16 = True
# eval mode on predicted scales: the correlation kernel writes its own fp16 planes (False: a split pass behind it; A/B switch of round 6)
_CORR_PRED_PLANES = True   # f16x2: the correlation on fp16 planes too (0: three bf16 planes)
_MATH = {}


def set_conv_math(mode):
    _MATH["mode"] = mode
    _MATH["planes"], _MATH["f16"], _MATH["f16_planes"] = _MODES[mode]
    _MATH["f16_on"] = True


set_conv_math(os.environ.get("RPNET_CONV_MATH", "f16x2"))


def conv_math():
    return _MATH["mode"]


import contextlib  # noqa: E402


@contextlib.contextmanager
def scope(conv_math=None, async_wgrad=None, mask_skip=None):
    """Activate a model's own options (rpnet_amd.schedule.Schedule) for the duration of ONE forward call; None = keep the
    process-wide default.  Restores the defaults afterwards (also on an exception): nothing leaks into the next model's call.
    The backward pass does not need the scope: every autograd node captured the options it needs when it was created."""
    global _MASK_SKIP
    saved = (_MATH["mode"], None, _ASYNC["on"], _MASK_SKIP)
    try:
        if conv_math is not None and conv_math != saved[0]:
            set_conv_math(conv_math)
        if async_wgrad is not None:
            _ASYNC["on"] = bool(async_wgrad)
        if mask_skip is not None:
            _MASK_SKIP = bool(mask_skip)
        yield
    finally:
        if _MATH["mode"] != saved[0]:
            set_conv_math(saved[0])
        # (the per-forward fp16-planes flag, set_f16_active, is the LAST call's decision until the next forward makes its own)
        _ASYNC["on"] = saved[2]
        _MASK_SKIP = saved[3]


def f16_mode():
    """fp16 planes in use: the mode is f16x2 and the current forward is large enough for it to pay (set_f16_active)"""
    return _MATH["f16"] and _MATH.get("f16_on", True)


def set_f16_active(on):
    """Per-forward switch of the fp16 planes inside the f16x2 mode: a single small episode is launch-bound and the
    fp16 path's extra small launches (tensor scales, split passes of pooled / concatenated inputs) cost more than its
    matrix work saves (batch 1 at 128^2: 6.3 vs 6.9 ms per step), so small calls stay on three bf16 planes."""
    _MATH["f16_on"] = bool(on)


def pack_planes():
    """planes of the operand packs the current forward will ask for: 0 (fp32 kernels), 3 (bf16), 2 / 1 (fp16, when active)"""
    if not _MATH["planes"]:
        return 0
    return _MATH["f16_planes"] if f16_mode() else _MATH["planes"]


def set_async_wgrad(on=True):
    """Opt-in "bucket mode" of the backward pass: weight gradients are launched on a second HIP stream and
    every parameter gradient of a conv+BN layer is accumulated by the producing kernel straight into the
    existing `param.grad` (the flat bucket of rpnet_amd.parallel) instead of being handed to autograd's
    AccumulateGrad (one torch add per parameter).  Parameter hooks then do not fire for those parameters;
    a parameter that must keep its hook is tagged `_rpnet_autograd_grad = True`."""
    _ASYNC["on"] = bool(on)


# diagnostic hook (None: off): a list that ConvBnRelu.backward fills with (tag, weight shape, clone) of its intermediate tensors
# in stream order (round 4's diagnostic of the pooled-pass fault compared them between runs of one step)
_TAPS = None


_TAPS_PIN = False      # True: keep the tensors themselves (no copy kernel: the step's timing stays what it is, their memory is not reused)


def _tap(tag, weight, *tensors):
    if _TAPS is not None:
        for i, t in enumerate(tensors):
            if t is not None:
                _TAPS.append((f"{tag}{i}", tuple(weight.shape), t.detach() if _TAPS_PIN else t.detach().clone()))


def _direct(p, on=None):
    """on: the weight-gradient option the autograd node captured at forward time (None: the process-wide switch)"""
    if not ((_ASYNC["on"] if on is None else on) and p is not None and p.grad is not None and p.grad.is_contiguous()):
        return False
    tag = getattr(p, "_rpnet_autograd_grad", False)
    if tag is False or tag is None:
        return True
    # tagged by a gradient bucket (rpnet_amd.parallel.FlatGradBucket): this parameter's AccumulateGrad hook launches a segment's
    # all-reduce — only while there is an exchange to launch (a process group of more than one rank, hooks not suspended by a graph
    # capture); otherwise the gradient goes straight into the bucket like every other (no AccumulateGrad add, no hook)
    wants = getattr(tag, "wants_hooks", None)
    return not (wants() if wants is not None else True)


def _accumulate_direct(p, g, on=None):
    """bucket mode: add a parameter gradient that a kernel could only WRITE (the first layer's direct weight gradient) into p.grad on the
    producing stream — ordered against the other chain's accumulation into the same parameter — and hand autograd nothing; otherwise g"""
    if not _direct(p, on):
        return g
    _order_wait(p.data_ptr())
    p.grad.add_(g)
    _order_done(p.data_ptr())
    return None


def use_compute_stream(device):
    """The stream the step's main chain runs on: the caller's current stream.  (Round 3 tried a high-priority main stream:
    +1 % alone, nothing on top of releasing every weight gradient behind its layer's dgrad — _WGRAD_DEFER — and removed.)"""
    return torch.cuda.current_stream(device)


def _reduce_stream(device):
    key = ("reduce", device)
    s = _ASYNC["side"].get(key)
    if s is None:
        s = _ASYNC["side"][key] = torch.cuda.Stream(device=device)
    return s


# Two chains of one forward / backward pass on two streams that share BatchNorm modules (the support and the query call of
# the encoder: RP_Net.forward): every launch that updates a module's running statistics (forward) or accumulates into its
# parameter gradients (backward) waits for the previous such launch of the SAME module on the other stream — the order of
# the reference's two calls is kept, no two launches write one buffer at a time.  Keyed by the parameter's address.
_ORDER = {"on": False, "ev": {}}


def order_begin(on):
    """RP_Net.forward: (re)start the per-module ordering for this pass"""
    _ORDER["on"] = bool(on)
    _ORDER["ev"].clear()


def _order_wait(key):
    if _ORDER["on"]:
        rec = _ORDER["ev"].get(key)
        if rec is not None:
            cur = torch.cuda.current_stream()
            if rec[1] != cur.cuda_stream:
                cur.wait_event(rec[0])


def _order_done(key):
    if _ORDER["on"]:
        cur = torch.cuda.current_stream()
        e = torch.cuda.Event()
        e.record(cur)
        _ORDER["ev"][key] = (e, cur.cuda_stream)


def _cre_stream(device):
    """the stream of the CRE's second branch (modules.ContextCorrelationEncoder.forward_masked, train mode)"""
    key = ("cre", device)
    s = _ASYNC["side"].get(key)
    if s is None:
        s = _ASYNC["side"][key] = torch.cuda.Stream(device=device)
    return s


def _pack_stream(device):
    """the stream of the per-step weight packing (WeightCache.prepack_async)"""
    key = ("pack", device)
    s = _ASYNC["side"].get(key)
    if s is None:
        s = _ASYNC["side"][key] = torch.cuda.Stream(device=device)
    return s


def _side_stream(device):
    s = _ASYNC["side"].get(device)
    if s is None:
        s = _ASYNC["side"][device] = torch.cuda.Stream(device=device)
    return s


def _defer_wgrad(launch):
    """Queue an async weight-gradient launch (ConvBnRelu.backward) for release behind a later dgrad; the end of the backward
    pass (engine callback) and every bucket operation flush the queue through join_side_streams."""
    _ASYNC["fifo"].append(launch)
    if not _ASYNC["queued"]:      # once per backward pass (reset_async re-arms it after a failed one)
        torch.autograd.Variable._execution_engine.queue_callback(join_side_streams)
        _ASYNC["queued"] = True


def _release_wgrads(keep):
    """Launch the queued weight gradients, oldest first, down to the `keep` newest: each launch waits for everything the
    main stream holds at this moment (the dgrad just enqueued)."""
    q = _ASYNC["fifo"]
    while len(q) > keep:
        q.pop(0)()


def join_side_streams(final=True):
    """Launch what is still queued, then make the current stream wait for every async weight-gradient launch issued so far.
    final=False (the gradient bucket's hooks, in the middle of a backward pass): the tensors kept alive for the side streams
    stay alive — only the CALLING stream has waited, and a block freed now could be handed to another stream of the pass (the
    CRE branch, the encoder's second chain) while a weight gradient still reads it."""
    _release_wgrads(0)
    for dev in list(_ASYNC["pending"]):
        torch.cuda.current_stream(dev).wait_stream(_ASYNC["side"][dev])
        red = _ASYNC["side"].get(("reduce", dev))
        if red is not None:
            torch.cuda.current_stream(dev).wait_stream(red)
        cre = _ASYNC["side"].get(("cre", dev))      # the CRE's second branch writes its BatchNorm parameter gradients there
        if cre is not None:
            torch.cuda.current_stream(dev).wait_stream(cre)
    if not final:
        return
    _ASYNC["pending"].clear()
    _ASYNC["keep"].clear()        # freed behind the waits above: whatever reuses their blocks is ordered after the side streams
    _ASYNC["queued"] = False


def reset_async():
    """Start of a forward pass / of a gradient-bucket operation: join whatever a previous backward left on the side
    streams (a backward that raised never ran its engine callback) and allow the next backward to queue its own join."""
    if _ASYNC["pending"] or _ASYNC["fifo"]:
        join_side_streams()
    _ASYNC["queued"] = False


# ---------------------------------------------------------------- which arithmetic actually ran
# Every convolution / correlation launch is counted by kind and by the arithmetic of its operands, so that a layer that
# quietly left the requested arithmetic (no operand bound, unsupported channel count) shows up: bench.py prints the
# counts of one step, tests assert them (reset_arith / arith_counts).
ARITH = collections.Counter()
_PLANE_NAME = {0: "f32", 1: "f16", 2: "f16x2", 3: "bf16x3"}


def reset_arith():
    ARITH.clear()


def arith_counts():
    """{kind: {arithmetic: launches}} since reset_arith(); kinds: conv3x3 (forward + input gradient), wgrad3x3,
    conv1x1, wgrad1x1, corr (forward), corr_bwd"""
    out = {}
    for (kind, arith), n in sorted(ARITH.items()):
        out.setdefault(kind, {})[arith] = n
    return out


def _cconv(name, d, *rest):
    """rpnet_conv_fwd / rpnet_conv_wgrad with the launch counted (the reduce-only phase of a two-phase weight
    gradient is not a second launch of the GEMM)"""
    wg = name == "rpnet_conv_wgrad"
    if not wg or rest[0] is not None:
        ARITH[(("wgrad" if wg else "conv") + ("3x3" if d.taps == 9 else "1x1"), _PLANE_NAME[d.split_planes])] += 1
    call(name, C.byref(d), *rest)


def _cup4(d, mode):
    """rpnet_conv_up4 with the launch counted (kind conv3x3_up4: four products per output instead of nine)"""
    ARITH[("conv3x3_up4", _PLANE_NAME[d.split_planes])] += 1
    call("rpnet_conv_up4", C.byref(d), mode)


def _up4_ok(pw, planes, upsample, x1, in_scale, xs0, N, H, W, cout):
    """does this launch take the collapsed form of an up_conv layer (see _UP4): two fp16 planes, a single unmasked source, a
    plain 3x3 layer whose shapes fit the kernel"""
    if not (_UP4 and upsample and planes in (1, 2) and x1 is None and in_scale is None and pw is not None and pw.taps == 9
            and pw.cin_pad == pw.cin and pw.cin % 32 == 0 and cout % 64 == 0):
        return False
    probe = ConvDesc()
    probe.N, probe.H, probe.W, probe.C0, probe.Co0, probe.split_planes = N, H, W, pw.cin, cout, planes
    probe.x0, probe.y0 = ptr(xs0), ptr(xs0)
    return bool(query("rpnet_conv_up4_supported", C.byref(probe), 1))


def up4_layer_ok(weight, planes, calls):
    """WeightCache.prepack's question: will EVERY encoder call of this forward run this up_conv layer on the collapsed form?
    calls: (N, H, W) of the layer's OUTPUT per call.  A layer that would fall back (maps below 16 x 16, ...) belongs into the batched
    nine-tap prepack, not into a per-layer pack on first use."""
    cout, cin = weight.shape[0], weight.shape[1]
    if not (_UP4 and planes in (1, 2) and cin % 32 == 0 and cout % 64 == 0):
        return False
    for N, H, W in calls:
        probe = ConvDesc()
        probe.N, probe.H, probe.W, probe.C0, probe.Co0, probe.split_planes = N, H, W, cin, cout, planes
        probe.x0 = probe.y0 = ptr(weight)
        if not query("rpnet_conv_up4_supported", C.byref(probe), 1):
            return False
    return True


# zeroed device scalars for rpnet_conv_desc.out_absmax (eval-mode f16 scales): one pool per device, handed out slot by slot,
# re-zeroed with ONE fill at the start of every RP_Net.forward (reset_absmax_pool); a call outside a forward that runs
# out of slots gets a fresh pool
_ABSMAX = {}
_ABSMAX_SLOTS = 256


def reset_absmax_pool(device):
    pool = _ABSMAX.get(device)
    if pool is None or pool[1] > 0:
        if pool is None:
            _ABSMAX[device] = [torch.zeros(_ABSMAX_SLOTS, device=device, dtype=torch.float32), 0]
        else:
            pool[0].zero_()
            pool[1] = 0


def _absmax_slot(device):
    pool = _ABSMAX.get(device)
    if pool is None or pool[1] >= _ABSMAX_SLOTS:
        # a fresh pool may be asked for on a side stream (the eval-mode CRE runs w_q there) while the main stream takes the
        # next slot without waiting: its zero fill must be complete for EVERY stream before the first slot is handed out,
        # or a later atomic max of another stream could land in front of it
        pool = _ABSMAX[device] = [torch.zeros(_ABSMAX_SLOTS, device=device, dtype=torch.float32), 0]
        torch.cuda.current_stream(device).synchronize()
    pool[1] += 1
    return pool[0][pool[1] - 1:pool[1]]


# ---- predicted fp16 scales of eval-mode layers (round 3).  Running statistics give no a-priori bound of an eval-mode
# BatchNorm output, so round 2 MEASURED every layer's maximum (out_absmax) and made the fp16 planes in a second pass over the
# tensor — one extra pass per layer and a launch-to-launch dependency, which is why small calls (the reference driver's
# batch of 2 slices) stayed on three bf16 planes.  Now: slot i of the absmax pool belongs to the same layer launch in every
# call of the same shape, so the maxima of the PREVIOUS call (x PRED_SAFETY, rounded up to a power of two) predict this
# call's bounds; the conv epilogue writes the fp16 planes of output / predicted scale directly (rpnet_conv_desc.y_split_scale)
# while out_absmax still measures the truth.  pred_end() compares: a maximum above its predicted bound (fp16 would have
# overflowed or lost its top bit) makes the caller redo the call on measured scales — counted in pred_stats().
PRED_SAFETY = float(os.environ.get("RPNET_EVAL_PRED_SAFETY", "4"))
_EVAL_PREDICT = True
_PRED = {}


def _pred_dev(device):
    """per device: the prediction records by call key, the record of the forward in progress, the tallies"""
    pd = _PRED.get(device)
    if pd is None:
        pd = _PRED[device] = {"by_key": {}, "cur": None, "last_key": None, "calls": 0, "predicted_calls": 0, "violations": 0}
    return pd


def _pred_state(device, key):
    """The record of ONE call shape (key = RP_Net.forward's pred_key: module, arithmetic, shapes, T): its predicted scales and
    bounds, the violation flag its comparison launch writes, whether a captured graph owns a pending comparison.  Keyed, so
    that two captured graphs (rpnet_amd.graph.GraphedEval of two shapes) or a graph beside eager calls of another shape never
    run on each other's scales or trip over each other's flags."""
    pd = _pred_dev(device)
    st = pd["by_key"].get(key)
    if st is None:
        st = pd["by_key"][key] = {"scale": torch.zeros(_ABSMAX_SLOTS, device=device), "bound": torch.zeros(_ABSMAX_SLOTS, device=device),
                                  "viol": torch.zeros(1, device=device, dtype=torch.int32), "n": 0, "active": False, "pending": False}
    return st


def pred_ready(device, key):
    """does a history of this call shape exist (an earlier eval call on this device had the same key)"""
    pd = _PRED.get(device)
    st = pd["by_key"].get(key) if pd is not None else None
    return _EVAL_PREDICT and st is not None and st["n"] > 0


def pred_begin(device, key, allow=True):
    """start of an eval-mode forward on fp16 planes (after reset_absmax_pool): predicted scales are used when an earlier
    call had this key and `allow` (False: the redo after a violation)"""
    pd = _pred_dev(device)
    ready = pred_ready(device, key)
    st = _pred_state(device, key)
    st["active"] = bool(allow and ready)
    pd["cur"] = st
    pd["calls"] += 1
    pd["predicted_calls"] += int(st["active"])
    return st["active"]


def pred_scale(device):
    """the predicted scale (device scalar) of the absmax slot handed out last, or None when this call measures"""
    pd = _PRED.get(device)
    st = pd["cur"] if pd is not None else None
    pool = _ABSMAX.get(device)
    if st is None or not st["active"] or pool is None or pool[1] > st["n"]:
        return None
    return st["scale"][pool[1] - 1:pool[1]]


def pred_end(device, key):
    """end of that forward: one launch compares the measured maxima with the bounds the call ran with (when it ran on
    predictions) and turns them into the next call's predictions.  Returns True when the call must be redone on measured
    scales (a maximum exceeded its predicted bound).  Inside a stream capture the comparison is recorded but not read:
    pred_check_pending(device, key) reads it after the replay."""
    pd = _pred_dev(device)
    st = _pred_state(device, key)
    pd["cur"], pd["last_key"] = None, key
    pool = _ABSMAX.get(device)
    n = pool[1] if pool is not None else 0
    ran_predicted = st["active"]
    if ran_predicted and n != st["n"]:
        raise RuntimeError(f"rpnet_amd: an eval call of key {key} took {n} absmax slots, its predecessor {st['n']}")
    st["active"] = False
    if n == 0:
        st["n"] = 0
        return False
    if ran_predicted:
        st["viol"].zero_()
    call("rpnet_predict_scales", ptr(pool[0]), ptr(st["bound"]), ptr(st["scale"]), n, PRED_SAFETY, 1 if ran_predicted else 0,
         ptr(st["viol"]))
    st["n"] = n
    if not ran_predicted:
        return False
    if torch.cuda.is_current_stream_capturing():
        st["pending"] = True       # stays set: EVERY replay of that graph ran on predictions and has to be checked
        return False
    bad = int(st["viol"].item()) > 0
    pd["violations"] += int(bad)
    if bad:
        st["n"] = 0                # the redo measures and rebuilds the history
    return bad


def pred_last_key(device):
    """the key of the eval forward that ended last on this device (GraphedEval remembers it with its captured graph)"""
    pd = _PRED.get(device)
    return pd["last_key"] if pd is not None else None


def pred_check_pending(device, key=None):
    """after replaying a captured eval forward of call key `key` that ran on predicted scales: did a maximum exceed its bound
    (the replay itself zeroed the flag and ran the comparison: the flag read here is this replay's)"""
    pd = _PRED.get(device)
    if pd is None:
        return False
    st = pd["by_key"].get(key if key is not None else pd["last_key"])
    if st is None or not st["pending"]:
        return False
    bad = int(st["viol"].item()) > 0
    pd["violations"] += int(bad)
    return bad


def pred_stats(device=None):
    """{calls, predicted_calls, violations} of the eval-mode fp16 prediction (all devices summed when device is None)"""
    out = {"calls": 0, "predicted_calls": 0, "violations": 0}
    for dv, pd in _PRED.items():
        if device is None or dv == device:
            for k in out:
                out[k] += pd[k]
    return out


def _empty(shape, like, dtype=torch.float32):
    return torch.empty(shape, device=like.device, dtype=dtype)


def _ws(nbytes, like):
    return torch.empty((max(int(nbytes), 16) + 7) // 8, device=like.device, dtype=torch.float64)


class Operand:
    """An NHWC fp32 activation together with the forms the operand loads of its consumers read:
      x      the fp32 tensor (what autograd differentiates; pooled / concatenated / masked consumers split it themselves)
      p16    fp16 planes [1 or 2, ...] of x / scale, written by the kernel that produced x (rpnet_bn_relu), or None
      pbf    three exact bf16 planes [3, ...] of x, or None
      scale  device scalar: the power-of-two tensor scale of the fp16 planes, from a rigorous bound of |x|
             (train-mode BatchNorm: |gamma| sqrt(n) + |beta|), or None when no bound is known — then a consumer in
             f16x2 / f16 mode runs on three bf16 planes, and the launch counters (arith_counts) show it.
    A tensor derived from x whose values are a subset of x's (max-pool, a batch slice, an alias) keeps the bound:
    `derive`."""
    __slots__ = ("x", "p16", "pbf", "scale", "planes_only", "deferred", "masked")

    def __init__(self, x, p16=None, pbf=None, scale=None, planes_only=False, deferred=None, masked=None):
        self.x, self.p16, self.pbf, self.scale = x, p16, pbf, scale
        # deferred (train-mode cre.q, conv_bn_relu_op(defer_act=True)): (y, batch scale, batch shift) — x is still UNWRITTEN, the
        # consumer (CosineMatchUp's fused launch) applies BatchNorm + ReLU and fills it
        self.deferred = deferred
        # masked: (mask tensor, mode, planes) — the operand planes of x * mask (mode 1) / x * (1 - mask) (mode 2) already exist
        # (written by the previous iteration's fused glue launch); a convolution gathering x with that very mask takes them
        self.masked = masked
        # planes_only: x is a shape-only placeholder for autograd (a zero-storage expanded tensor) — its fp32 values were
        # never written because the single consumer reads p16 (conv_bn_relu_op(z_unused=True)); reading x is an error
        self.planes_only = planes_only

    def values(self):
        """the fp32 tensor, for a consumer that reads its VALUES"""
        if self.planes_only:
            raise RuntimeError("rpnet_amd: this operand exists as fp16 planes only (its producer was told that its single "
                               "consumer is a 3x3 convolution on fp16 planes); a consumer asked for the fp32 values")
        return self.x

    @property
    def shape(self):
        return self.x.shape

    def derive(self, x):
        return Operand(x, scale=self.scale)


# async weight gradients go out behind a dgrad (ConvBnRelu.backward): 1 = their own layer's, d = the one d - 1 layers further
# down the chain (a deeper backlog of MFMA-bound work beside the chain's HBM-bound passes); 0 = in front of their own
_WGRAD_DEFER = 1
# A/B switch: split K for the eval-mode 3x3 convolutions whose grid covers half of the CUs or fewer
_EVAL_SPLITK = True
# A/B switch: BatchNorm + ReLU + MaxPool2d(2, 2) of the encoder levels whose output feeds only its pool in one pass
_POOL_FUSE = True
# A/B switch: the BatchNorm-backward apply pass of Conv1.conv.0 inside its direct weight gradient (rpnet_conv1_wgrad_bn)
_CONV1_BN_FUSE = True
# Conv1.conv.0 (Cin = 1) in training on fp16 planes: its pre-BatchNorm tensor is never written — the statistics launch only
# sums it, BatchNorm + ReLU and the backward's reduction pass / weight gradient make it again from the image (nine
# multiply-adds per value against eight bytes written and re-read; csrc/conv_first.hip).  _CONV1_RECOMP = False: A/B switch (module constant; tests / tools set it)
_CONV1_RECOMP = True
# up_conv (nn.Upsample(scale_factor=2) -> Conv2d 3x3, net/modules.py:61-75: Up5, Up4) on its COLLAPSED weights: a 3x3 convolution over a
# nearest-x2 up-sampled image reads a 2 x 2 block of source pixels per output pixel, so with the weights of coinciding taps added
# up front the layer needs 4 / 9 of its multiply-adds — forward, input gradient and weight gradient (csrc/conv_up4_dma.hip,
# rpnet_conv_up4; the two layers are 17 % of the step's FLOPs as the reference writes them).  Differs from the nine-product
# form by the rounding of the weight sums (2^-24 relative).  _UP4 = False: the nine-product form (A/B switch: module constant, tests / tools set it).
_UP4 = True
# w_k(x * mask) / w_q(x * (1 - mask)) (net/rp_net.py:275,283): output tiles whose masked input is zero on the tile and its halo
# (forward) or whose factor is zero on the tile (input gradient) skip their K loop (rpnet_conv_desc.skip_*) — same bits as the
# dense launch.  The support mask covers 2 - 15 % of the pixels, so most tiles of w_k go.  bench.py keeps the HEADLINE dense
# (the roofline accounting is algorithmic) and reports this as its own leg.  RPNET_MASK_SKIP=0 / Schedule.mask_skip = False: off
_MASK_SKIP = os.environ.get("RPNET_MASK_SKIP", "1") == "1"


_SKIP_STATS = None     # diagnostic (bench.py): a list that receives every launch's flag buffer (preset to 255 = "no tile here")


def _set_skip(d, mask, mode, halo, N, H, W, on=None):
    """lend a launch the mask and the flag scratch of the tile skip (only the LDS-DMA patch kernels use them); on: the option the
    autograd node captured at forward time (None: the process-wide switch)"""
    if (_MASK_SKIP if on is None else on) and mask is not None and mode in (1, 2):
        if _SKIP_STATS is not None:
            d._skip_ws = torch.full((max(N * H * W // 128, 16),), 255, device=mask.device, dtype=torch.uint8)
            _SKIP_STATS.append(d._skip_ws)
        else:
            d._skip_ws = torch.empty(max(N * H * W // 128, 16), device=mask.device, dtype=torch.uint8)
        d.skip_mask, d.skip_mode, d.skip_halo, d.skip_ws = ptr(mask), mode, halo, ptr(d._skip_ws)
        ARITH[("zero_tile_skip", "armed")] += 1


def as_operand(t):
    return t if (t is None or isinstance(t, Operand)) else Operand(t)


def split_bf16(x, planes, scale=None, mode=0):
    """x [..., C] fp32 (optionally * scale or * (1 - scale) per row) -> [planes, ..., C] bf16 planes with
    x = sum of the planes (rpnet_split_bf16)."""
    hip.require_gpu(x)
    x = x.contiguous()
    out = torch.empty((planes,) + tuple(x.shape), device=x.device, dtype=torch.bfloat16)
    c = x.shape[-1]
    call("rpnet_split_bf16", ptr(x), ptr(scale), mode if scale is not None else 0, ptr(out), x.numel() // c, c, planes)
    return out


def _split_operand(op, planes, scale=None, mode=0):
    """bf16 planes of a conv operand; an unmasked operand remembers its split (skip connections ask for it again)."""
    if scale is not None and mode:
        pre = op.masked
        if pre is not None and pre[0] is scale and pre[1] == mode and pre[2].shape[0] == planes and pre[2].dtype == torch.bfloat16:
            return pre[2]
        return split_bf16(op.values(), planes, scale, mode)
    if op.pbf is not None and op.pbf.shape[0] == planes and op.pbf.shape[1:] == op.x.shape:
        return op.pbf
    op.pbf = split_bf16(op.values(), planes)
    return op.pbf


# the 1x1 convolution over cat([corr, fm1]) on split planes too (gathering weight pack, per-source fp16 scales, single-tap
# split weight gradient); _CONV1X1_SPLIT = False keeps it on the fp32-MFMA kernels (A/B switch)
_CONV1X1_SPLIT = True


def _use_split(pw, x0, x1):
    """convolutions whose channel counts fit the split pack run on the 16-bit matrix pipe when enabled: the 3x3 layers
    (channel ranges in multiples of 32) and the 1x1 layer over two sources (its 121 + 256 channels are packed as 128 + 256)"""
    if not (_MATH["planes"] and pw is not None and x0.shape[-1] % 32 == 0 and (x1 is None or x1.shape[-1] % 32 == 0)):
        return False
    if pw.taps == 9:      # channel ranges in multiples of 32, or padded ranges through the gathering pack (mask_feature_map)
        return (pw.cin_pad == pw.cin and pw.cin % 32 == 0) or (pw.cin_pad != pw.cin and pw.cin_pad % 64 == 0)
    return _CONV1X1_SPLIT and pw.taps == 1 and pw.cin_pad % 64 == 0 and pw.cout % 64 == 0 and x0.shape[-1] % 64 == 0


def split_f16(x, s_a, s_b=None, mask=None, mode=0, want_scale=True, planes=None, a_is_bound=False):
    """fp16 planes of x * f(mask) / s, s = max(s_a, s_b) (device scalars) -> (planes [2 or 1, ...], s [1]) (rpnet_split_f16);
    a_is_bound: s_a is a measured bound of |x| instead (eval mode), s = its power-of-two scale"""
    hip.require_gpu(x)
    x = x.contiguous()
    planes = planes or _MATH["f16_planes"] or 2
    out = torch.empty((planes,) + tuple(x.shape), device=x.device, dtype=torch.float16)
    s = torch.empty(1, device=x.device, dtype=torch.float32) if want_scale else None
    c = x.shape[-1]
    call("rpnet_split_f16", ptr(x), ptr(mask), mode if mask is not None else 0, ptr(s_a), ptr(s_b), ptr(s), ptr(out),
         x.numel() // c, c, planes, 1 if a_is_bound else 0)
    return out, s


def _f16_sources(op0, op1, in_scale, in_mode, two_scales=False):
    """fp16 operand planes of a convolution's source(s) -> (planes0, planes1, scale, scale1), or None when a source
    carries no bound (then the caller runs on three bf16 planes).  scale1 is None when both sources share ONE tensor
    scale (`scale`).  A BatchNorm output arrives with its planes and scale (Operand.p16 / .scale, written by
    rpnet_bn_relu); pooled / masked / concatenated sources are split here from the fp32 tensor with the joint scale of
    their producers.  two_scales: the kernel takes a scale per source (the 1x1 convolution: rpnet_conv_desc.acc_scale_x1),
    so two sources that both arrive with planes are used as they are."""
    s0 = op0.scale
    s1 = op1.scale if op1 is not None else None
    if s0 is None or (op1 is not None and s1 is None):
        return None
    masked = in_scale is not None and in_mode
    fp = _MATH["f16_planes"]

    def ready(op):
        return op.p16 is not None and op.p16.shape[1:] == op.x.shape and op.p16.shape[0] == fp

    if op1 is None and not masked and ready(op0):
        return op0.p16, None, s0, None
    pre = op0.masked
    if (op1 is None and masked and pre is not None and pre[0] is in_scale and pre[1] == in_mode and pre[2].shape[0] == fp
            and pre[2].dtype == torch.float16):
        return pre[2], None, s0, None      # x * f(mask) / s0 split by the launch that made the mask (rpnet_refine_glue_fwd)
    if two_scales and op1 is not None and not masked and ready(op0) and ready(op1):
        return op0.p16, op1.p16, s0, s1
    xs0, s = split_f16(op0.values(), s0, s1, in_scale if masked else None, in_mode if masked else 0)
    xs1 = split_f16(op1.values(), s0, s1, want_scale=False)[0] if op1 is not None else None
    if op1 is None and not masked:
        op0.p16 = xs0            # s == s0: a later consumer of the same operand reuses the planes
    return xs0, xs1, s, None


# ------------------------------------------------------------------ weight packing
class PackedWeight:
    """Packed copies of one nn.Conv2d weight (see rpnet_pack_conv_weight)."""

    def __init__(self, weight, split=None, pool=None):
        cout, cin, kh, kw = weight.shape
        self._pool = pool         # WeightCache's store of PADDED pack buffers (their padding rows are written once, not per step)
        self.taps = kh * kw
        if split is None:
            self.off0, self.split, self.off1, self.cin_pad = 0, cin, cin, cin
        else:  # (first-source channels, padded first-source channels[, padded total]): concat [corr(121->128), fm1];
            # [features(64), mask(1->64)] of mask_feature_map: the weight's input channels n0.. land on packed rows n0_pad..
            n0, n0_pad = split[:2]
            self.off0, self.split, self.off1 = 0, n0, n0_pad
            self.cin_pad = split[2] if len(split) > 2 else n0_pad + (cin - n0)
        self.cout, self.cin = cout, cin
        self.has_wd = cin >= 32
        self._weight = weight
        self._wp = self._wd = self.wps = self.wds = None    # packed on first use (fp32 and split packs are exclusive per layer)

    def _pack_f32(self):
        w, n = self._weight, self.taps * self.cin_pad * self.cout
        mk = torch.zeros if self.cin_pad != self.cin else torch.empty
        self._wp = mk(n, device=w.device, dtype=torch.float32)
        self._wd = mk(n, device=w.device, dtype=torch.float32) if self.has_wd else None
        call("rpnet_pack_conv_weight", ptr(w), ptr(self._wp), ptr(self._wd), self.cout, self.cin, self.taps, self.off0,
             self.split, self.off1, self.cin_pad)

    @property
    def wp(self):
        if self._wp is None:
            self._pack_f32()
        return self._wp

    @property
    def wd(self):
        if self._wp is None:
            self._pack_f32()
        return self._wd

    def split_packs(self, planes):
        """split packs of the same weight (rpnet_pack_conv_weight_split), made on first use: planes == 3 -> (wp, wd) bf16
        planes; planes == 2 / 1 -> (wp, wd, row scale of wp [cout], row scale of wd [cin_pad]) fp16 planes of w / row scale."""
        pk = self.wps.get(planes) if self.wps else None
        if pk is None:
            w = self._weight
            pk = self.alloc_split(planes)
            call("rpnet_pack_conv_weight_split", ptr(w), ptr(pk[0]), ptr(pk[1]), self.cout, self.cin, self.taps,
                 self.off0, self.split, self.off1, self.cin_pad, planes, ptr(pk[2]) if planes <= 2 else None,
                 ptr(pk[3]) if planes <= 2 else None)
        return pk

    def up4_packs(self, planes):
        """packs of the COLLAPSED weights of an up_conv layer (nn.Upsample(2) -> Conv2d 3x3, net/modules.py:61-75): the nine taps
        summed onto the 2 x 2 source pixels each output phase really reads (rpnet_upconv_collapse_weights: [4 cout][cin][2][2]), then
        the ordinary four-tap pack -> (wp4, wd4, row scale of wp4 [4 cout], row scale of wd4 [cin]); fp16 planes (1 / 2)"""
        up = getattr(self, "_up4", None)
        if up is None:
            up = self._up4 = {}
        pk = up.get(planes)
        if pk is None:
            w = self._weight
            wc = torch.empty((4 * self.cout, self.cin, 2, 2), device=w.device, dtype=torch.float32)
            call("rpnet_upconv_collapse_weights", ptr(w), ptr(wc), self.cout, self.cin)
            n = 4 * self.cin * 4 * self.cout
            wp4 = torch.empty((planes, n), device=w.device, dtype=torch.float16)
            wd4 = torch.empty((planes, n), device=w.device, dtype=torch.float16)
            t4 = torch.empty(4 * self.cout, device=w.device, dtype=torch.float32)
            u4 = torch.empty(self.cin, device=w.device, dtype=torch.float32)
            call("rpnet_pack_conv_weight_split", ptr(wc), ptr(wp4), ptr(wd4), 4 * self.cout, self.cin, 4, 0, self.cin, self.cin, self.cin,
                 planes, ptr(t4), ptr(u4))
            pk = up[planes] = (wp4, wd4, t4, u4)
        return pk

    def alloc_split(self, planes):
        """buffers of the split pack (registered as this layer's pack; the caller fills them)"""
        if self.wps is None:
            self.wps = {}
        n = self.taps * self.cin_pad * self.cout
        padded = self.cin_pad != self.cin
        w = self._weight
        # a padded pack (the 1x1 convolution over cat([corr, fm1]): 377 -> 384 rows) keeps its buffers across steps: the pack kernel
        # writes the real rows only, the zero rows / unit scales of the padding are set once (three fill launches per step otherwise).
        # Safe because a layer's packs are written and read in stream order step after step (prepack waits for the caller's stream)
        key = (w.data_ptr(), tuple(w.shape), planes, self.cin_pad, self.off0, self.split, self.off1, str(w.device))
        if padded and self._pool is not None and key in self._pool:
            pk = self._pool[key]
            self.wps[planes] = pk
            return pk
        mk = torch.zeros if padded else torch.empty
        dt = torch.bfloat16 if planes == 3 else torch.float16
        wps, wds = mk((planes, n), device=w.device, dtype=dt), mk((planes, n), device=w.device, dtype=dt)
        if planes == 3:
            pk = (wps, wds)
        else:
            t = torch.empty(self.cout, device=w.device, dtype=torch.float32)
            # the kernel writes the scale of every real gathered row; only padding rows (none on the 3x3 layers) need a preset
            u = (torch.ones if padded else torch.empty)(self.cin_pad, device=w.device, dtype=torch.float32)
            pk = (wps, wds, t, u)
        if padded and self._pool is not None:
            self._pool[key] = pk
        self.wps[planes] = pk
        return pk


class WeightCache:
    """Per-forward cache of packed weights: cleared at the start of every RP_Net.forward, so
    weights are repacked once per step (never reused across optimizer steps)."""

    def __init__(self):
        self._d = {}
        self._ready = None          # (event, streams that already wait for it): the packs of prepack_async
        self._padded = {}           # buffers of PADDED split packs, kept across clear() (PackedWeight.alloc_split)

    def clear(self):
        self._d.clear()
        self._ready = None

    def prepack_async(self, weights, planes, device, up4=()):
        """prepack() on the pack stream, beside whatever the caller's stream does next (the first-layer convolution and its
        BatchNorm passes need no pack: 0.2 ms of HBM-bound packing at the head of every training step with nothing else on
        the machine); every stream that fetches a layer's pack afterwards (get) first waits for the event recorded here.
        The pack stream starts behind everything the caller's stream holds (the optimizer's update of the weights, the
        previous backward pass's reads of the old packs)."""
        main = torch.cuda.current_stream(device)
        ps = _pack_stream(device)
        ps.wait_stream(main)
        self._ready = None          # (the packs made below must not wait for an earlier call's event)
        with torch.cuda.stream(ps):
            self.prepack(weights, planes, up4)
            ev = torch.cuda.Event()
            ev.record(ps)
        self._ready = (ev, set(), torch.cuda.is_current_stream_capturing())

    def get(self, weight, split=None):
        if self._ready is not None and self._ready[2] and not torch.cuda.is_current_stream_capturing():
            # the event was recorded inside a stream capture (GraphedTrainStep / GraphedEval): it belongs to the graph, an eager
            # stream cannot wait for it — a replay's packs are ordered behind the replay on the stream that launched it
            self._ready = None
        if self._ready is not None:
            sid = hip.stream()
            if sid not in self._ready[1]:
                torch.cuda.current_stream(weight.device).wait_event(self._ready[0])
                self._ready[1].add(sid)
        # the split (channel ranges of the gathered sources) is part of what a pack IS: a layer fetched once without and
        # once with its ranges must not share one
        key = (weight.data_ptr(), weight._version, None if split is None else tuple(split))
        pw = self._d.get(key)
        if pw is None:
            pw = PackedWeight(weight.detach(), split, self._padded)
            self._d[key] = pw
        return pw

    def prepack(self, weights, planes, up4=()):
        """The split packs of many layers in ONE launch per kernel (rpnet_pack_conv_weights_split) instead of two
        launches per layer on first use: RP_Net.forward hands in the 3x3 weights of the whole model right after
        clearing the cache.  `planes`: 3 (bf16), 2 / 1 (fp16 planes with row scales).
        up4: the weights among them that belong to up_conv layers (nn.Upsample -> Conv2d): on two fp16 planes their forward and
        input gradient run on the COLLAPSED four-tap pack (PackedWeight.up4_packs, see _UP4), which is made here instead of
        the nine-tap pack (a launch that falls back to the nine-tap form packs it on first use)."""
        up4 = list(up4) if (_UP4 and planes in (1, 2)) else []
        if up4:
            skip = {id(w) for w in up4}
            weights = [w for w in weights if id(w) not in skip]
            for w in up4:
                self.get(w).up4_packs(planes)
        pws = [self.get(w) for w in weights]
        pws = [pw for pw in pws if pw.taps == 9 and pw.cin_pad == pw.cin and pw.cin % 32 == 0 and pw.cout % 32 == 0
               and not (pw.wps and planes in pw.wps)]
        for i in range(0, len(pws), hip.PACK_MAX):
            chunk = pws[i:i + hip.PACK_MAX]
            items = (hip.PackItem * len(chunk))()
            for it, pw in zip(items, chunk):
                bufs = pw.alloc_split(planes)
                it.w, it.wp, it.wd = ptr(pw._weight), ptr(bufs[0]), ptr(bufs[1])
                it.row_scale_wp, it.row_scale_wd = (ptr(bufs[2]), ptr(bufs[3])) if planes <= 2 else (None, None)
                it.cout, it.cin, it.taps = pw.cout, pw.cin, pw.taps
                it.cin_off0, it.cin_split, it.cin_off1, it.cin_pad = pw.off0, pw.split, pw.off1, pw.cin_pad
            call("rpnet_pack_conv_weights_split", items, len(chunk), planes)


    def materialize(self, weights, planes):
        """Make every pack the layers of `weights` will read EXIST NOW, on the current stream (split packs of `planes`
        planes, or the fp32 packs when planes == 0; layers prepack() already served cost nothing).  RP_Net.forward calls
        this before it forks the encoder's two chains onto two streams: a pack made lazily by the first chain's launch would
        be ordered on that chain's stream only, and the other chain — which gets the same cached PackedWeight — could read
        the freshly allocated buffer before the pack kernel has run."""
        for w in weights:
            pw = self.get(w)
            if planes and pw.taps == 9 and pw.cin_pad == pw.cin and pw.cin % 32 == 0 and pw.cout % 32 == 0:
                pw.split_packs(planes)
            else:
                pw.wp      # noqa: B018 (the property packs on first use)


# tuning / test override of the kernel variant, carried by every descriptor (rpnet_conv_desc.tune): 0 = the library's
# choice, v + 1 = tile variant v of the split forward kernels, 4 (weight gradient) = the 4-wave layout
# RPNET_CONV_DMA=0: the default tile policy without the LDS-DMA patch kernel (tune bit 16; A/B switch of round 3)
TUNE = {"tile": int(os.environ.get("RPNET_TUNE_TILE", "0")) + (0 if os.environ.get("RPNET_CONV_DMA", "1") == "1" else 0x10000),
        "wgrad": int(os.environ.get("RPNET_TUNE_WGRAD", "0"))}


def _desc(x0, x1, w, bias, in_scale, in_mode, y0, y1, N, H, W, taps, ups, groups=1, ep_scale=None, ep_shift=None,
          ep_relu=0, out_scale=None, out_mode=0, accumulate=0, co_split=None, wgrad=False):
    """wgrad: the descriptor goes to rpnet_conv_wgrad (its `tune` field then carries TUNE["wgrad"], not the tile variant of
    the forward kernels: the two entry points read the field differently)"""
    d = ConvDesc()
    d.x0, d.x1 = ptr(x0), ptr(x1)
    d.C0 = x0.shape[-1]
    d.C1 = x1.shape[-1] if x1 is not None else 0
    d.w, d.bias = ptr(w), ptr(bias)
    d.in_scale, d.in_scale_mode = ptr(in_scale), in_mode if in_scale is not None else 0
    d.y0, d.y1 = ptr(y0), ptr(y1)
    d.Co0 = y0.shape[-1] if co_split is None else co_split[0]
    d.Co1 = (y1.shape[-1] if y1 is not None else 0) if co_split is None else co_split[1]
    d.ep_scale, d.ep_shift, d.ep_relu = ptr(ep_scale), ptr(ep_shift), ep_relu
    d.out_scale, d.out_scale_mode, d.accumulate = ptr(out_scale), out_mode if out_scale is not None else 0, accumulate
    d.N, d.H, d.W, d.taps, d.upsample, d.groups = N, H, W, taps, ups, groups
    d.tune = TUNE["wgrad"] if wgrad else TUNE["tile"]
    return d


# ------------------------------------------------------------- conv + BN + ReLU
class ConvBnRelu(Function):
    """Conv2d(3x3 p1 | 1x1, bias) -> BatchNorm2d -> ReLU on NHWC activations
    (net/modules.py:47-49,66-69; net/rp_net.py:50-59,65-69).

    x0 (,x1): sources concatenated along C; `in_scale` [N,h,w] with mode 1 (x*s) / 2 (x*(1-s));
    `upsample`: nearest x2 in front of the conv; `groups`: BatchNorm statistic groups.
    """

    @staticmethod
    def forward(ctx, x0, x1, in_scale, weight, bias, gamma, beta, running_mean, running_var, nbt, pw, training,
                groups, upsample, in_mode, out_split, ops, produced):
        """ops = (Operand of x0, Operand of x1 or None): the planes / scales of the sources; `produced` (a dict) receives
        the planes / scale this launch wrote for the output ("p16", "pbf", "scale") — conv_bn_relu_op builds the
        output Operand from it."""
        hip.require_gpu(x0, weight)
        op0, op1 = ops
        N, Hs, Ws, _ = x0.shape
        H, W = (Hs * 2, Ws * 2) if upsample else (Hs, Ws)
        cout = weight.shape[0]
        first = weight.shape[1] == 1 and weight.shape[2] == 3  # Cin = 1 direct convolution
        z = _empty((N, H, W, cout), x0)
        if not training:
            aff = getattr(pw, "eval_affine", None)     # the folded BatchNorm lives as long as the layer's packed weights
            if aff is None:
                aff = (_empty((cout,), x0), _empty((cout,), x0))
                call("rpnet_bn_eval_affine", ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), BN_EPS, ptr(aff[0]),
                     ptr(aff[1]), cout)
                if pw is not None:
                    pw.eval_affine = aff
            scale, shift = aff
            # f16x2 / f16 in eval mode: running statistics give no a-priori bound of the output, so the launch measures
            # one — max |z| through rpnet_conv_desc.out_absmax — and the fp16 tensor scale (and, for a direct 3x3 /
            # correlation consumer, the planes) follow from it; an output that wants neither keeps the plain path.
            want16 = f16_mode() and cout % 32 == 0 and bool(out_split) and (_CORR16 or out_split != "corr")
            mx = _absmax_slot(x0.device) if want16 else None
            sp = pred_scale(x0.device) if want16 else None      # predicted scale of this launch (None: this call measures)
            np_out = _MATH["planes"] if (out_split in (True, "corr") and cout % 32 == 0 and not first and not want16) else 0
            zs = torch.empty((np_out, N, H, W, cout), device=x0.device, dtype=torch.bfloat16) if np_out else None
            z16 = None
            if first:
                call("rpnet_conv1_fwd", ptr(x0), ptr(weight), ptr(bias), ptr(z), ptr(scale), ptr(shift), N, H, W, cout, ptr(mx),
                     None, 1)
            elif _use_split(pw, x0, x1):
                f16 = _f16_sources(op0, op1, in_scale, in_mode, pw.taps == 1) if f16_mode() else None
                up4e = False
                if f16 is not None:      # fp16 planes of the sources (their scales measured by their own launches)
                    fp = _MATH["f16_planes"]
                    # up_conv in eval mode on its collapsed four-tap form too (round 6; training: round 5): the folded BatchNorm affine,
                    # ReLU and the measured maximum are the up4 kernel's plain epilogue; its output feeds a concatenation, which
                    # splits the fp32 tensor itself (out_split "scale": no planes wanted from this launch)
                    up4e = out_split in ("scale", False) and _up4_ok(pw, fp, upsample, x1, in_scale, f16[0], N, H, W, cout)
                if up4e:
                    pk4 = pw.up4_packs(fp)
                    d = _desc(f16[0], None, pk4[0], bias, None, 0, z, None, N, H, W, pw.taps, 1, 1, scale, shift, 1)
                    d.split_planes = fp
                    d.acc_scale_col, d.acc_scale_x = ptr(pk4[2]), ptr(f16[2])
                    d._keep = (f16, pk4)
                elif f16 is not None:
                    wps, _, t_row, _ = pw.split_packs(fp)
                    d = _desc(f16[0], f16[1], wps, bias, None, 0, z, None, N, H, W, pw.taps, upsample, 1, scale, shift, 1)
                    d.split_planes = fp
                    d.acc_scale_col, d.acc_scale_x, d.acc_scale_x1 = ptr(t_row), ptr(f16[2]), ptr(f16[3])
                    d._keep = f16
                    if pw.taps == 9 and x1 is None and not upsample:
                        _set_skip(d, in_scale, in_mode, 1, N, H, W)
                else:
                    np_ = _MATH["planes"]
                    d = _desc(_split_operand(op0, np_, in_scale, in_mode), None if x1 is None else _split_operand(op1, np_),
                              pw.split_packs(np_)[0], bias, None, 0, z, None, N, H, W, pw.taps, upsample, 1, scale, shift, 1)
                    d.split_planes = np_
                if up4e:
                    d.out_absmax = ptr(mx)
                else:
                    d.y_split, d.split_out_planes, d.out_absmax = ptr(zs), np_out, ptr(mx)
                if sp is not None and out_split in (True, "corr") and not up4e:
                    # the fp16 planes of output / predicted scale straight out of the epilogue: no second pass over z
                    fpo = _MATH["f16_planes"]
                    z16 = torch.empty((fpo, N, H, W, cout), device=x0.device, dtype=torch.float16)
                    d.y_split, d.split_out_planes, d.y_split_scale = ptr(z16), fpo, ptr(sp)
                if up4e:
                    _cup4(d, 1)
                else:
                    if _EVAL_SPLITK and d.split_planes == 2 and pw.taps == 9 and N * H * W * cout <= 128 * 256 * 64:
                        # a grid that would leave half of the CUs idle (batch-2 calls): lend the workspace that lets the launch
                        # cut its K range into parts (rpnet_conv_desc.splitk_ws)
                        nb = query("rpnet_conv_splitk_workspace_bytes", C.byref(d))
                        if nb:
                            d._ws = _ws(nb, x0)
                            d.splitk_ws, d.splitk_ws_bytes = ptr(d._ws), nb
                    _cconv("rpnet_conv_fwd", d)
            else:
                d = _desc(x0, x1, pw.wp, bias, in_scale, in_mode, z, None, N, H, W, pw.taps, upsample, 1, scale, shift, 1)
                d.y_split, d.split_out_planes, d.out_absmax = ptr(zs), np_out, ptr(mx)
                _cconv("rpnet_conv_fwd", d)
            if zs is not None:
                produced["pbf"] = zs      # written by the conv epilogue: no separate split pass in eval mode
            if want16 and sp is not None:
                if z16 is not None:
                    produced["p16"], produced["scale"] = z16, sp
                elif out_split in (True, "corr"):   # (first layer / fp32 kernels: a split pass, but no wait for the measured maximum)
                    produced["p16"], produced["scale"] = split_f16(z, sp, want_scale=False)[0], sp
                else:
                    produced["scale"] = sp
            elif want16:
                if out_split in (True, "corr"):     # planes and scale from the measured bound in one launch
                    produced["p16"], produced["scale"] = split_f16(z, mx, a_is_bound=True)
                else:
                    sz = torch.empty(1, device=x0.device, dtype=torch.float32)
                    call("rpnet_pow2_scale", ptr(mx), ptr(sz))
                    produced["scale"] = sz
            ctx.eval_mode = True
            return z
        # the first layer on fp16 planes whose only consumer reads the planes: y is summed, never written (see _CONV1_RECOMP)
        recomp = bool(first and _CONV1_RECOMP and f16_mode() and cout % 8 == 0 and 256 % (cout // 8) == 0 and out_split is True
                      and produced.get("z_unused") and not produced.get("pool_req") and _CONV1_BN_FUSE
                      and query("rpnet_conv1_stats_blocks", N, H, W, cout, groups) > 0)
        y = None if recomp else _empty((N, H, W, cout), x0)
        stats = _empty((4, groups, cout), x0)  # scale, shift, mean, invstd
        fused, xs, sx, sx1 = 0, None, None, None
        if first:      # batch statistics out of the same launch (one partial row per block and group)
            fused = query("rpnet_conv1_stats_blocks", N, H, W, cout, groups)
            part = torch.empty(groups * fused * cout * 2, device=x0.device, dtype=torch.float64) if fused else None
            call("rpnet_conv1_fwd", ptr(x0), ptr(weight), ptr(bias), ptr(y), None, None, N, H, W, cout, None, ptr(part), groups)
        else:
            f16 = _f16_sources(op0, op1, in_scale, in_mode, pw.taps == 1) if (f16_mode() and _use_split(pw, x0, x1)) else None
            if op0.planes_only and not (f16 is not None and op1 is None and in_scale is None and pw.cin % 64 == 0 and cout % 64 == 0):
                raise RuntimeError("rpnet_amd: a planes-only operand (conv_bn_relu_op(z_unused=True)) reached a convolution "
                                   "that cannot run forward AND weight gradient from its fp16 planes")
            if op1 is not None and op1.planes_only and not (f16 is not None and f16[1] is op1.p16 and pw.cin_pad % 64 == 0
                                                            and cout % 64 == 0 and x0.shape[-1] % 64 == 0):
                raise RuntimeError("rpnet_amd: a planes-only second source reached a convolution that does not read its "
                                   "fp16 planes as they are (forward and weight gradient)")
            up4 = False
            if f16 is not None:      # fp16 planes (two, or one in "f16" mode) with a tensor scale; weights with row scales
                xs, sx, sx1 = (f16[0], f16[1]), f16[2], f16[3]
                fp = _MATH["f16_planes"]
                up4 = _up4_ok(pw, fp, upsample, x1, in_scale, xs[0], N, H, W, cout)
                wps, t_row = (pw.up4_packs(fp)[0], pw.up4_packs(fp)[2]) if up4 else (pw.split_packs(fp)[0], pw.split_packs(fp)[2])
                d = _desc(xs[0], xs[1], wps, bias, None, 0, y, None, N, H, W, pw.taps, upsample, groups)
                d.split_planes = fp
                d.acc_scale_col, d.acc_scale_x, d.acc_scale_x1 = ptr(t_row), ptr(sx), ptr(sx1)
                if pw.taps == 9 and x1 is None and not upsample:
                    _set_skip(d, in_scale, in_mode, 1, N, H, W)
            elif _use_split(pw, x0, x1):
                np_ = _MATH["planes"]
                xs = (_split_operand(op0, np_, in_scale, in_mode), None if x1 is None else _split_operand(op1, np_))
                d = _desc(xs[0], xs[1], pw.split_packs(np_)[0], bias, None, 0, y, None, N, H, W, pw.taps, upsample, groups)
                d.split_planes = np_
            else:
                d = _desc(x0, x1, pw.wp, bias, in_scale, in_mode, y, None, N, H, W, pw.taps, upsample, groups)
            fused = query("rpnet_conv_up4_stats_blocks" if up4 else "rpnet_conv_stats_blocks", C.byref(d))
            if fused:  # batch statistics come out of the conv epilogue: y is not re-read
                part = torch.empty(groups * fused * cout * 2, device=x0.device, dtype=torch.float64)
                d.stats_partial = ptr(part)
            if up4:
                _cup4(d, 1)
            else:
                _cconv("rpnet_conv_fwd", d)
        _order_wait(gamma.data_ptr())      # the running statistics: after the other chain's update of this module
        if fused:
            call("rpnet_bn_stats_from_partial", ptr(part), fused, N, H * W, cout, groups, ptr(gamma), ptr(beta),
                 ptr(running_mean), ptr(running_var), ptr(nbt), BN_MOMENTUM, BN_EPS, ptr(stats[0]), ptr(stats[1]),
                 ptr(stats[2]), ptr(stats[3]))
        else:
            wsb = query("rpnet_bn_workspace_bytes", cout, groups)
            ws = _ws(wsb, x0)
            call("rpnet_bn_stats", ptr(y), N, H * W, cout, groups, ptr(gamma), ptr(beta), ptr(running_mean),
                 ptr(running_var), ptr(nbt), BN_MOMENTUM, BN_EPS, ptr(stats[0]), ptr(stats[1]), ptr(stats[2]), ptr(stats[3]),
                 ptr(ws), wsb)
        _order_done(gamma.data_ptr())
        # the output also as the operand planes of its consumer: out_split True = a 3x3 convolution reads it as is (fp16
        # planes with the tensor scale in f16x2 mode), "corr" = the local correlation (bf16 planes), "scale" = no planes
        # but the fp16 tensor scale (pooled / concatenated / masked 3x3 consumers split the fp32 tensor), False = none
        want16 = f16_mode() and cout % 32 == 0 and out_split in ((True, "scale", "corr") if _CORR16 else (True, "scale"))
        np_out = 0
        if out_split in (True, "corr") and cout % 32 == 0 and _MATH["planes"]:
            np_out = _MATH["f16_planes"] if want16 else _MATH["planes"]
        # pool: BatchNorm + ReLU + MaxPool2d(2, 2) in one pass (the output feeds nothing but its pool): possible when this
        # layer runs on split planes forward AND backward (its dy then exists as planes: the pooled backward writes those)
        pool = bool(produced.get("pool")) and np_out > 0 and out_split is True and xs is not None and not first and \
            H % 2 == 0 and W % 2 == 0 and pw.cin_pad % 64 == 0 and cout % 64 == 0 and x0.shape[-1] % 64 == 0
        produced["pooled"] = pool
        Hz, Wz = (H // 2, W // 2) if pool else (H, W)
        if pool:
            z = _empty((N, Hz, Wz, cout), x0)
        zs = torch.empty((np_out, N, Hz, Wz, cout), device=x0.device, dtype=torch.float16 if np_out <= 2 else torch.bfloat16) if np_out else None
        sz = torch.empty(1, device=x0.device, dtype=torch.float32) if want16 else None
        if produced.get("z_unused") and want16 and np_out and out_split in (True, "corr") and (pool or not produced.get("pool_req")):
            # the single consumer reads the fp16 planes: the fp32 form is never written, z is a zero-storage placeholder
            # of the right shape for autograd (a pool request that could not be fused keeps the fp32 form: the separate
            # max-pool reads it)
            z = torch.empty(1, device=x0.device, dtype=torch.float32).expand(N, Hz, Wz, cout)
            produced["planes_only"] = True
        defer = (bool(produced.get("defer_act")) and not pool and not np_out and not want16 and not recomp
                 and groups == 1 and not produced.get("planes_only"))
        if defer:
            # BatchNorm + ReLU are applied by the consumer's fused launch (CosineMatchUp, rpnet_refine_glue_fwd), which fills z
            produced["deferred"] = (y, stats[0], stats[1])
        elif recomp:
            if not (produced.get("planes_only") and want16 and np_out):
                raise RuntimeError("rpnet_amd: the first layer was run without its pre-BatchNorm tensor but its output is not planes-only")
            call("rpnet_conv1_bn_relu", ptr(x0), ptr(weight), ptr(bias), ptr(stats[0]), ptr(stats[1]), None, ptr(zs), np_out,
                 ptr(gamma), ptr(beta), ptr(sz), N, H, W, cout, groups)
            ARITH[("bn_relu", "first layer made again from the image")] += 1
        else:
            # the tensor scale comes out of the same launch: with the fp16 planes, or alone (np_out == 0, "scale")
            call("rpnet_bn_relu", ptr(y), ptr(stats[0]), ptr(stats[1]), None if produced.get("planes_only") else ptr(z), ptr(zs),
                 np_out, ptr(gamma), ptr(beta),
                 ptr(sz) if want16 else None, N, H * W, cout, groups, W if pool else 0, None, 0)
        if pool:
            ARITH[("bn_relu", "with the 2x2 max-pool")] += 1
        if want16 and np_out:
            produced["p16"] = zs          # the next convolution's operand, produced here instead of by a separate pass
        elif zs is not None:
            produced["pbf"] = zs
        if want16:
            produced["scale"] = sz
        _tap("fwd:y,stats", weight, y, stats)
        ctx.save_for_backward(x0, x1, in_scale, weight, gamma, y, stats)
        ctx.yshape = (N, H, W, cout)
        ctx.pw, ctx.cfg, ctx.eval_mode, ctx.pool = pw, (groups, upsample, in_mode, first), False, pool
        ctx.up4 = (not first) and up4
        ctx.bias, ctx.beta, ctx.xs, ctx.sx, ctx.sx1 = bias, beta, xs, sx, sx1
        ctx.opts = (_ASYNC["on"], _MASK_SKIP)       # the model's options when the node was made (rpnet_amd.schedule): backward reads these
        return z

    @staticmethod
    @once_differentiable
    def backward(ctx, dz):
        if ctx.eval_mode:
            raise NotImplementedError("rpnet_amd: backward through eval-mode BatchNorm is not implemented "
                                      "(the reference only evaluates under torch.no_grad, test_rpnet.py:163)")
        x0, x1, in_scale, weight, gamma, y, stats = ctx.saved_tensors
        async_on, skip_on = getattr(ctx, "opts", (None, None))
        pw = ctx.pw
        groups, upsample, in_mode, first = ctx.cfg
        dz = dz.contiguous()
        _tap("bwd_in:y,stats,dz", weight, y, stats, dz)
        N, H, W, cout = ctx.yshape
        wsb = query("rpnet_bn_workspace_bytes", cout, groups)
        ws = _ws(wsb, dz)
        beta, bias = ctx.beta, ctx.bias
        if y is None:
            # the first layer without its pre-BatchNorm tensor: reduction pass and weight gradient make y again from the image
            direct = _direct(gamma, async_on) and _direct(beta, async_on)
            dgamma, dbeta = (None, None) if direct else (_empty((cout,), dz), _empty((cout,), dz))
            rows = query("rpnet_conv1_bn_bwd_rows", N, H, W, cout, groups)
            part = torch.empty(groups * rows * cout * 2, device=dz.device, dtype=torch.float64)
            call("rpnet_conv1_bn_bwd_partial", ptr(x0), ptr(weight), ptr(bias), ptr(dz), ptr(stats), ptr(part), N, H, W, cout, groups)
            ARITH[("bn_bwd", "first layer made again from the image")] += 1
            if direct:
                _order_wait(gamma.data_ptr())
            call("rpnet_bn_bwd", ptr(dz), None, ptr(gamma), ptr(stats[0]), ptr(stats[1]), ptr(stats[2]), ptr(stats[3]), None, None, 0,
                 None, ptr(gamma.grad if direct else dgamma), ptr(beta.grad if direct else dbeta), N, H * W, cout, groups,
                 1 if direct else 0, ptr(part), None, rows, 0, ptr(ws), wsb, None, 0)
            if direct:
                _order_done(gamma.data_ptr())
            dw = torch.empty_like(weight)
            wb = query("rpnet_conv1_wgrad_workspace_bytes", N, H, W, cout)
            ws2 = _ws(wb, dz)
            coef = ws.data_ptr() + query("rpnet_bn_bwd_coef_offset", cout, groups)
            call("rpnet_conv1_wgrad_bn", ptr(x0), ptr(dz), None, ptr(stats), coef, ptr(dw), N, H, W, cout, groups, ptr(ws2), wb,
                 ptr(weight), ptr(bias))
            dw = _accumulate_direct(weight, dw, async_on)
            db = None if _direct(bias, async_on) else torch.zeros_like(gamma)
            return None, None, None, dw, db, dgamma, dbeta, None, None, None, None, None, None, None, None, None, None, None
        # which forms of dy the two consumers (wgrad, dgrad) want: split-bf16 planes and / or fp32
        np_ = ctx.xs[0].shape[0] if ctx.xs is not None else 0
        need_d = not first and (ctx.needs_input_grad[0] or (x1 is not None and ctx.needs_input_grad[1]))
        wsplit = bool(np_) and pw.cin_pad % 64 == 0 and cout % 64 == 0 and x0.shape[-1] % 64 == 0
        dsplit = bool(np_) and cout % 32 == 0 and need_d
        dys = torch.empty((np_,) + tuple(y.shape), device=y.device, dtype=torch.bfloat16) if (wsplit or dsplit) else None
        sdy = torch.empty(1, device=y.device, dtype=torch.float32) if (dys is not None and np_ <= 2) else None   # fp16: tensor scale
        dy = _empty(ctx.yshape, dz) if (first or not wsplit or (need_d and not dsplit)) else None
        # Conv1.conv.0 (Cin = 1) has no input gradient: its direct weight gradient forms dy itself from dz, y and the
        # coefficients of the reduction pass, so the apply pass (12 bytes per element of the largest tensor) is not run
        fuse1 = first and _CONV1_BN_FUSE and stats.is_contiguous()
        if fuse1:
            dy = None
        if ctx.pool and (dys is None or dz.shape[1] * 2 != H):
            raise RuntimeError("rpnet_amd: the pooled BatchNorm backward needs dy as split planes and the pooled gradient")
        direct = _direct(gamma, async_on) and _direct(beta, async_on)     # straight into the gradient bucket, no AccumulateGrad add
        dgamma, dbeta = (None, None) if direct else (_empty((cout,), y), _empty((cout,), y))
        ARITH[("bn_bwd", "own reduction pass")] += 1
        if direct:
            _order_wait(gamma.data_ptr())  # gamma.grad / beta.grad: after the other chain's accumulation into them
        call("rpnet_bn_bwd", ptr(dz), ptr(y), ptr(gamma), ptr(stats[0]), ptr(stats[1]), ptr(stats[2]), ptr(stats[3]),
             ptr(dy), ptr(dys), np_ if dys is not None else 0, ptr(sdy), ptr(gamma.grad if direct else dgamma),
             ptr(beta.grad if direct else dbeta), N, H * W, cout, groups, 1 if direct else 0, None, None, 0,
             W if ctx.pool else 0, ptr(ws), wsb, None, 0)
        if direct:
            _order_done(gamma.data_ptr())
        _tap("bn_bwd:dz,dy,dys,sdy,ws", weight, dz, dy, dys, sdy, ws)
        dw = torch.empty_like(weight)
        dx0 = dx1 = dscale = None
        if first:
            wb = query("rpnet_conv1_wgrad_workspace_bytes", N, H, W, cout)
            ws2 = _ws(wb, y)
            if fuse1:
                coef = ws.data_ptr() + query("rpnet_bn_bwd_coef_offset", cout, groups)
                call("rpnet_conv1_wgrad_bn", ptr(x0), ptr(dz), ptr(y), ptr(stats), coef, ptr(dw), N, H, W, cout, groups,
                     ptr(ws2), wb, None, None)
            else:
                call("rpnet_conv1_wgrad", ptr(x0), ptr(dy), ptr(dw), N, H, W, cout, ptr(ws2), wb)
            dw = _accumulate_direct(weight, dw, async_on)
        else:
            # same gather descriptor as the forward (sources, up-sampling, x*mask factor); dy is the other operand
            if wsplit:       # both wgrad operands as split planes (the x*mask factor is already in xs)
                dyp = dys
                d = _desc(ctx.xs[0], ctx.xs[1], None, None, None, 0, None, None, N, H, W, pw.taps, upsample,
                          co_split=(cout, 0), wgrad=True)
                d.split_planes = np_
                if np_ <= 2:
                    d.acc_scale_x, d.acc_scale_dy, d.acc_scale_x1 = ptr(ctx.sx), ptr(sdy), ptr(ctx.sx1)
            else:
                dyp = dy
                d = _desc(x0, x1, None, None, in_scale, in_mode, dy, None, N, H, W, pw.taps, upsample, wgrad=True)
            wb = query("rpnet_conv_wgrad_workspace_bytes", N, H, W, pw.cin_pad, cout, pw.taps)
            # the collapsed up_conv (see _UP4): sixteen tap products per source pixel instead of thirty-six
            wup4 = bool(getattr(ctx, "up4", False)) and wsplit and np_ in (1, 2) and bool(query("rpnet_conv_wgrad_up4_supported", C.byref(d)))
            if wup4:
                wb = query("rpnet_conv_wgrad_up4_workspace_bytes", N, H, W, pw.cin, cout)

            def wgrad(dy_ptr, dw_ptr, ws_t):
                """the weight-gradient entry point of this layer: GEMM phase (dw_ptr None), reduce phase (dy_ptr None) or both"""
                if wup4:
                    if dy_ptr is not None:
                        ARITH[("wgrad3x3_up4", _PLANE_NAME[d.split_planes])] += 1
                    call("rpnet_conv_wgrad_up4", C.byref(d), dy_ptr, dw_ptr, ptr(ws_t), wb)
                else:
                    _cconv("rpnet_conv_wgrad", d, dy_ptr, dw_ptr, pw.cin, pw.off0, pw.split, pw.off1, ptr(ws_t), wb)
            deferred = None
            if _direct(weight, async_on):
                # the stream this backward node runs on (the one that produced dy and will run this layer's dgrad), taken NOW:
                # a deferred launch may be released by a later node that runs on another stream (the CRE's second branch, the
                # encoder's second chain), and must still wait for THIS one
                prod = torch.cuda.current_stream(y.device)

                def launch_async():
                    dev = y.device
                    side, main = _side_stream(dev), prod
                    side.wait_stream(main)                   # dy, x are ready on the producing stream (deferred: and the dgrad is done)
                    d.accumulate = 1
                    with torch.cuda.stream(side):
                        ws2 = _ws(wb, y)
                        if wsplit:   # GEMM on the side stream, its HBM-bound reduce on a third one under the next layer's GEMM
                            wgrad(ptr(dyp), None, ws2)
                            red = _reduce_stream(dev)
                            red.wait_stream(side)
                            _ASYNC["keep"].append(ws2)      # read by the reduce; alive until join_side_streams
                            with torch.cuda.stream(red):
                                wgrad(None, ptr(weight.grad), ws2)
                        else:
                            wgrad(ptr(dyp), ptr(weight.grad), ws2)
                    # alive until join_side_streams (the saved tensors outlive this node anyway)
                    _ASYNC["keep"].append((x0, x1, in_scale, dy, dyp, ctx.sx, ctx.sx1, sdy, ctx.xs))
                    if not _ASYNC["queued"]:      # once per backward pass (reset_async re-arms it after a failed one)
                        torch.autograd.Variable._execution_engine.queue_callback(join_side_streams)
                        _ASYNC["queued"] = True
                    _ASYNC["pending"].add(dev)
                # _WGRAD_DEFER >= 1 (default 1): the launch goes out right BEHIND this layer's dgrad and waits for it, so that it
                # starts when the main chain enters the BatchNorm-backward passes of the layer below — every HBM-bound pass
                # of the chain then has an MFMA-bound partner on the machine.  Launched in front of the dgrad (=0) the two
                # GEMMs share the CUs, end together, and the passes behind them run alone.
                if _WGRAD_DEFER:
                    _defer_wgrad(launch_async)
                    deferred = True
                else:
                    launch_async()
                dw = None
            else:
                ws2 = _ws(wb, y)
                wgrad(ptr(dyp), ptr(dw), ws2)
            need0 = ctx.needs_input_grad[0]
            need1 = x1 is not None and ctx.needs_input_grad[1]
            if need0 or need1:
                c0, c1 = x0.shape[-1], (x1.shape[-1] if x1 is not None else 0)
                # dgrad = the same implicit GEMM on dy with the flipped/transposed weight pack
                need_s = in_scale is not None and ctx.needs_input_grad[2]   # soft_mask: the mask is differentiable
                up4 = bool(getattr(ctx, "up4", False)) and dsplit and np_ in (1, 2)
                if up4:
                    # the collapsed up_conv: the input gradient comes out at the SOURCE resolution (the 2 x 2 sum is in the launch)
                    pk4 = pw.up4_packs(np_)
                    g0, g1 = _empty(x0.shape, y), None
                    dd = _desc(dys, None, pk4[1], None, None, 0, g0, None, N, H, W, pw.taps, 1)
                    dd.split_planes = np_
                    dd.acc_scale_col, dd.acc_scale_x = ptr(pk4[3]), ptr(sdy)
                    up4 = bool(query("rpnet_conv_up4_supported", C.byref(dd), 2))
                if not up4:
                    g0 = _empty((N, H, W, c0), y)
                    g1 = _empty((N, H, W, c1), y) if x1 is not None else None
                if up4:
                    pass
                elif dsplit:
                    pk = pw.split_packs(np_)
                    dd = _desc(dys, None, pk[1], None, None, 0, g0, g1, N, H, W,
                               pw.taps, 0, out_scale=None if need_s else in_scale, out_mode=in_mode)
                    dd.split_planes = np_
                    if np_ <= 2:
                        dd.acc_scale_col, dd.acc_scale_x = ptr(pk[3]), ptr(sdy)
                        if pw.taps == 9 and x1 is None and not upsample and not need_s:
                            _set_skip(dd, in_scale, in_mode, 0, N, H, W, skip_on)
                else:
                    dd = _desc(dy, None, pw.wd, None, None, 0, g0, g1, N, H, W, pw.taps, 0,
                               out_scale=None if need_s else in_scale, out_mode=in_mode)
                if up4:
                    _cup4(dd, 2)
                else:
                    _cconv("rpnet_conv_fwd", dd)
                _tap("dgrad:g0,g1", weight, g0, g1)
                if deferred:
                    _release_wgrads(_WGRAD_DEFER - 1)
                    deferred = None
                if need_s:   # d(x*f(s)) -> dx = g*f(s), ds = +-<g, x>
                    gx, dscale = torch.empty_like(g0), torch.empty_like(in_scale)
                    call("rpnet_rowdot_scale", ptr(g0), ptr(x0), ptr(in_scale), ptr(gx), ptr(dscale), N * H * W, c0,
                         in_mode, 0)
                    g0 = gx
                if upsample and not up4:
                    h0 = _empty(x0.shape, y)
                    call("rpnet_upsample2_bwd", ptr(g0), ptr(h0), N, H, W, c0)
                    g0 = h0
                dx0 = g0 if need0 else None
                dx1 = g1 if need1 else None
            if deferred:                  # no input gradient wanted: no dgrad to wait for
                _release_wgrads(_WGRAD_DEFER - 1)
        # conv bias in front of a train-mode BatchNorm: the gradient is analytically zero
        db = None if _direct(bias, async_on) else torch.zeros_like(gamma)
        return dx0, dx1, dscale, dw, db, dgamma, dbeta, None, None, None, None, None, None, None, None, None, None, None


def conv_bn_relu_op(x0, conv, bn, cache, training, x1=None, in_scale=None, in_mode=0, groups=1, upsample=False, split=None,
                    out_split=True, z_unused=False, pool=False, defer_act=False):
    """Conv -> BatchNorm -> ReLU on Operands (tensors are wrapped: no planes, no bound); returns the output Operand.
    out_split: also write the output as the operand planes of its consumer (when the split arithmetic is on): True = a
    3x3 convolution reads it as is, "corr" = the local correlation, "scale" = no planes, only the fp16 tensor scale (a
    pooled / concatenated / masked 3x3 consumer splits the fp32 tensor itself), False = neither (1x1 consumers).
    z_unused: the caller guarantees that the ONLY consumer of the output is a 3x3 convolution that takes it as its single,
    unmasked source (conv_block's first layer): in train mode on fp16 planes the fp32 form is then not written at all
    (a third of the launch's bytes) and the returned Operand is `planes_only`.
    pool: the caller wants MaxPool2d(2, 2) of the output and nothing else of it (net/unet.py:442-448): returns the POOLED
    Operand — in train mode on split planes out of the BatchNorm + ReLU pass itself (rpnet_bn_relu(pool_w): the
    full-resolution activation is never written, the backward finds the window maxima again from the conv output),
    otherwise through the separate max-pool launch.
    defer_act (train mode, out_split False): BatchNorm + ReLU are NOT applied here — the returned Operand's tensor is unwritten
    and its `deferred` = (y, batch scale, batch shift); the consumer's fused launch applies them and fills the tensor
    (CosineMatchUp over cre.q's output: the refinement loop's glue, rpnet_refine_glue_fwd)."""
    op0, op1 = as_operand(x0), as_operand(x1)
    pw = cache.get(conv.weight, split) if (conv.weight.shape[1] >= 32 or split is not None) else None
    produced = {"z_unused": bool(z_unused) and training and conv.weight.shape[0] % 64 == 0,
                "pool": bool(pool) and training and _POOL_FUSE, "pool_req": bool(pool),
                "defer_act": bool(defer_act) and training and not out_split}
    z = ConvBnRelu.apply(op0.x, None if op1 is None else op1.x, in_scale, conv.weight, conv.bias, bn.weight, bn.bias,
                         bn.running_mean, bn.running_var, bn.num_batches_tracked if training else None, pw, training, groups,
                         1 if upsample else 0, in_mode, out_split, (op0, op1), produced)
    out = Operand(z, produced.get("p16"), produced.get("pbf"), produced.get("scale"), bool(produced.get("planes_only")),
                  produced.get("deferred"))
    if pool and not produced.get("pooled"):
        out = maxpool2(out)
    return out


def conv_bn_relu(x0, conv, bn, cache, training, **kw):
    """single-layer convenience: the output TENSOR of conv_bn_relu_op (its planes and scale are dropped)"""
    return conv_bn_relu_op(x0, conv, bn, cache, training, **kw).x


class ConvRelu(Function):
    """Conv2d(3x3, padding = dilation, bias) [-> ReLU], no BatchNorm: the (conv, relu) layers of
    vgg.Encoder (net/vgg.py:39-58).  x [N,H,W,Cg] NHWC with Cg = the packed (zero padded) Cin."""

    @staticmethod
    def forward(ctx, x, weight, bias, pw, relu, dilation):
        hip.require_gpu(x, weight)
        N, H, W, _ = x.shape
        cout = weight.shape[0]
        z = _empty((N, H, W, cout), x)
        xs = None
        if _use_split(pw, x, None):
            np_ = _MATH["planes"]
            xs = split_bf16(x, np_)
            d = _desc(xs, None, pw.split_packs(np_)[0], bias, None, 0, z, None, N, H, W, pw.taps, 0, ep_relu=1 if relu else 0)
            d.split_planes = np_
        else:
            d = _desc(x, None, pw.wp, bias, None, 0, z, None, N, H, W, pw.taps, 0, ep_relu=1 if relu else 0)
        d.dilation = dilation
        _cconv("rpnet_conv_fwd", d)
        ctx.save_for_backward(x, weight, z)
        ctx.pw, ctx.cfg, ctx.xs = pw, (relu, dilation), xs
        return z

    @staticmethod
    @once_differentiable
    def backward(ctx, dz):
        x, weight, z = ctx.saved_tensors
        pw, xs = ctx.pw, ctx.xs
        relu, dilation = ctx.cfg
        N, H, W, cout = z.shape
        dy, db = torch.empty_like(z), _empty((cout,), z)
        wb = query("rpnet_bias_relu_bwd_workspace_bytes", cout)
        ws = _ws(wb, z)
        call("rpnet_bias_relu_bwd", ptr(dz.contiguous()), ptr(z) if relu else None, ptr(dy), ptr(db), N * H * W, cout,
             ptr(ws), wb)
        dw = torch.empty_like(weight)
        dys = split_bf16(dy, xs.shape[0]) if xs is not None and cout % 32 == 0 else None
        if dys is not None and dilation <= 1 and pw.cin % 64 == 0 and cout % 64 == 0:
            d = _desc(xs, None, None, None, None, 0, dy, None, N, H, W, pw.taps, 0, wgrad=True)
            d.split_planes, dyp = xs.shape[0], dys
        else:
            d = _desc(x, None, None, None, None, 0, dy, None, N, H, W, pw.taps, 0, wgrad=True)
            dyp = dy
        d.Co0, d.dilation = cout, dilation
        wb2 = query("rpnet_conv_wgrad_workspace_bytes", N, H, W, pw.cin_pad, cout, pw.taps)
        ws2 = _ws(wb2, z)
        _cconv("rpnet_conv_wgrad", d, ptr(dyp), ptr(dw), pw.cin, pw.off0, pw.split, pw.off1, ptr(ws2), wb2)
        dx = None
        if ctx.needs_input_grad[0] and pw.has_wd:
            dx = _empty(x.shape, z)
            if dys is not None:
                dd = _desc(dys, None, pw.split_packs(xs.shape[0])[1], None, None, 0, dx, None, N, H, W, pw.taps, 0)
                dd.split_planes = xs.shape[0]
            else:
                dd = _desc(dy, None, pw.wd, None, None, 0, dx, None, N, H, W, pw.taps, 0)
            dd.dilation = dilation
            _cconv("rpnet_conv_fwd", dd)
        return dx, dw, db, None, None, None


class MaxPool3(Function):
    """nn.MaxPool2d(kernel_size=3, stride=s, padding=1) (net/vgg.py:23-29) on NHWC."""

    @staticmethod
    def forward(ctx, z, stride):
        N, H, W, Cc = z.shape
        out = _empty((N, (H - 1) // stride + 1, (W - 1) // stride + 1, Cc), z)
        call("rpnet_maxpool3_fwd", ptr(z), ptr(out), N, H, W, Cc, stride)
        ctx.save_for_backward(z)
        ctx.stride = stride
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dpool):
        (z,) = ctx.saved_tensors
        N, H, W, Cc = z.shape
        dz = torch.empty_like(z)
        call("rpnet_maxpool3_bwd", ptr(z), ptr(dpool.contiguous()), ptr(dz), N, H, W, Cc, ctx.stride)
        return dz, None


# ------------------------------------------------------------------------ gradient fan-in
def sum_n(tensors):
    """one-pass sum of up to 16 equally shaped contiguous fp32 GPU tensors (rpnet_sum_n)"""
    out = torch.empty_like(tensors[0])
    for i in range(0, len(tensors), 15):
        chunk = ([out] if i else []) + [t.contiguous() for t in tensors[i:i + 15]]
        arr = (C.c_void_p * len(chunk))(*[t.data_ptr() for t in chunk])
        call("rpnet_sum_n", arr, len(chunk), ptr(out), out.numel())
    return out


class FanOut(Function):
    """x -> n aliases of x for n consumers; the backward adds the n incoming gradients in ONE kernel (autograd's
    own fan-in is a chain of n - 1 pairwise adds: 3 (n - 1) tensor passes instead of n + 1).  Used for the query
    features, which feed the 2 T masked convolutions of the refinement loop (net/rp_net.py:275,281-312)."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.shape = x.shape
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    @once_differentiable
    def backward(ctx, *grads):
        live = [g for g in grads if g is not None]
        if not live:
            return None, None
        return (live[0] if len(live) == 1 else sum_n(live)), None


class SplitRows(Function):
    """x [n0 + n1, ...] -> (x[:n0], x[n0:]) (the support and query halves of one encoder call); the backward is one
    concatenation instead of autograd's zero-fill + copy + add per slice."""

    @staticmethod
    def forward(ctx, x, n0):
        ctx.n = (n0, x.shape[0] - n0)
        ctx.meta = (tuple(x.shape[1:]), x.device, x.dtype)
        return x[:n0], x[n0:]

    @staticmethod
    @once_differentiable
    def backward(ctx, g0, g1):
        tail, dev, dt = ctx.meta
        parts = [g if g is not None else torch.zeros((n,) + tail, device=dev, dtype=dt) for g, n in zip((g0, g1), ctx.n)]
        return torch.cat(parts, 0), None


class SplitFan(Function):
    """x [n0 + n1, ...] -> k0 aliases of x[:n0] followed by k1 aliases of x[n0:] (the support and query halves of one encoder call
    with their consumers: the two 3x3 convolutions of the support CRE call, the 2 T of the refinement loop).  The backward sums
    each half's incoming gradients in one pass STRAIGHT INTO its rows of the gradient of x: SplitRows' concatenation (a pass over
    the whole tensor) and autograd's pairwise add for the support half are gone."""

    @staticmethod
    def forward(ctx, x, n0, k0, k1):
        ctx.cfg = (n0, k0, k1)
        ctx.meta = (tuple(x.shape), x.device, x.dtype)
        a, b = x[:n0], x[n0:]
        return tuple(a.view_as(a) for _ in range(k0)) + tuple(b.view_as(b) for _ in range(k1))

    @staticmethod
    @once_differentiable
    def backward(ctx, *grads):
        n0, k0, k1 = ctx.cfg
        shape, dev, dt = ctx.meta
        out = torch.empty(shape, device=dev, dtype=dt)
        for dst, gs in ((out[:n0], grads[:k0]), (out[n0:], grads[k0:])):
            live = [g.contiguous() for g in gs if g is not None]
            if not live:
                dst.zero_()
                continue
            for i in range(0, len(live), 15):
                chunk = ([dst] if i else []) + live[i:i + 15]
                if len(chunk) == 1:
                    dst.copy_(chunk[0])
                else:
                    arr = (C.c_void_p * len(chunk))(*[t.data_ptr() for t in chunk])
                    call("rpnet_sum_n", arr, len(chunk), ptr(dst), dst.numel())
        return out, None, None, None


# ------------------------------------------------------------------------ pooling
class PoolSkip(Function):
    """z -> (alias of z for the U-Net skip connection, nn.MaxPool2d(2, 2) of z) (net/unet.py:449-455, 460, 464: x3 and x4 feed
    their pool AND the decoder's concatenation).  The backward routes the pooled gradient to each window's first maximum and
    adds the gradient that arrived through the skip connection in the same pass (rpnet_maxpool2_bwd's `skip`), instead of
    autograd's separate add over the full-resolution tensor."""

    @staticmethod
    def forward(ctx, z):
        N, H, W, Cc = z.shape
        out = _empty((N, H // 2, W // 2, Cc), z)
        call("rpnet_maxpool2_fwd", ptr(z), ptr(out), N, H, W, Cc)
        ctx.save_for_backward(z)
        return z.view_as(z), out

    @staticmethod
    @once_differentiable
    def backward(ctx, dskip, dpool):
        if dpool is None:
            return dskip
        (z,) = ctx.saved_tensors
        N, H, W, Cc = z.shape
        dz = torch.empty_like(z)
        call("rpnet_maxpool2_bwd", ptr(z), ptr(dpool.contiguous()), None if dskip is None else ptr(dskip.contiguous()), ptr(dz),
             N, H, W, Cc)
        return dz


def pool_skip(op):
    """(the operand for the skip connection, its 2 x 2 max-pool) of an Operand whose fp32 tensor feeds both: PoolSkip"""
    op = as_operand(op)
    z, pooled = PoolSkip.apply(op.values())
    return Operand(z, p16=op.p16, pbf=op.pbf, scale=op.scale), op.derive(pooled)


class MaxPool2(Function):
    """nn.MaxPool2d(2, 2) (net/unet.py:397) on NHWC."""

    @staticmethod
    def forward(ctx, z):
        N, H, W, Cc = z.shape
        out = _empty((N, H // 2, W // 2, Cc), z)
        call("rpnet_maxpool2_fwd", ptr(z), ptr(out), N, H, W, Cc)
        ctx.save_for_backward(z)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dpool):
        (z,) = ctx.saved_tensors
        N, H, W, Cc = z.shape
        dz = torch.empty_like(z)
        call("rpnet_maxpool2_bwd", ptr(z), ptr(dpool.contiguous()), None, ptr(dz), N, H, W, Cc)
        return dz


def maxpool2(op):
    """MaxPool2 on an Operand: the pooled values are a subset of the source's, so its fp16 tensor scale still bounds them"""
    op = as_operand(op)
    return op.derive(MaxPool2.apply(op.values()))


def mask_avgpool(mask, scale):
    """F.avg_pool2d(mask[:, None], scale) (net/rp_net.py:269-272): [B,H,W] -> [B,h,w]; no gradient."""
    B, H, W = mask.shape
    out = _empty((B, H // scale, W // scale), mask)
    call("rpnet_mask_avgpool", ptr(mask.contiguous().float()), ptr(out), B, H, W, scale)
    return out


# -------------------------------------------------------------------- correlation
CORR_STRIDE = 128  # (2*5+1)^2 = 121 window channels padded to a GEMM-friendly 128


def corr_stride(r):
    """channel stride of the correlation output for radius r: (2 r + 1)^2 window channels padded to a multiple of 64 (128 up
    to the yaml's radius 5; 192 / 256 for radius 6 / 7, which run on the fp32 window kernels)"""
    kk = (2 * r + 1) ** 2
    return CORR_STRIDE if kk <= CORR_STRIDE else (kk + 63) // 64 * 64


class LocalCorr(Function):
    """Correlation(fm1, fm2, r) (net/rp_net.py:153-181), NHWC, output [B,h,w,128] (zero padded).  With the split
    arithmetic on (and r = 5, C % 128 == 0) it runs on the bf16 matrix pipe from the split planes of fm1 / fm2.
    Second output: an alias of fm1 for its other consumer (the 1x1 convolution over cat([corr, fm1]), net/rp_net.py:81),
    so that both gradients of fm1 arrive in this backward and are summed by the kernel's own store instead of a
    separate autograd add over the tensor."""

    @staticmethod
    def forward(ctx, f1, f2, r, ops=None, produced=None):
        """produced (a dict, optional): receives the correlation's own fp16 planes and measured scale ("p16", "scale") when
        its consumer (the 1x1 convolution) reads planes"""
        ctx.set_materialize_grads(False)
        B, h, w, Cc = f1.shape
        CORR_STRIDE = corr_stride(r)          # (shadows the module constant: every launch below takes this radius' stride)
        corr = _empty((B, h, w, CORR_STRIDE), f1)
        np_ = _MATH["planes"] if (r == 5 and Cc % 128 == 0) else 0
        o1, o2 = ops if ops is not None else (Operand(f1), Operand(f2))
        if (np_ and o1.p16 is not None and o2.p16 is not None and o1.p16.shape[1:] == f1.shape
                and o2.p16.shape[1:] == f2.shape and o1.p16.shape[0] == o2.p16.shape[0]):
            # both inputs are BatchNorm outputs that came with fp16 planes and their tensor scales
            (f1s, s1), (f2s, s2) = (o1.p16, o1.scale), (o2.p16, o2.scale)
            np_ = f1s.shape[0]          # 2, or 1 in "f16" mode
            ARITH[("corr", _PLANE_NAME[np_])] += 1
            # the correlation has no a-priori bound: the launch measures max |corr|, its planes are scaled by that
            mx = _absmax_slot(f1.device) if (produced is not None and _CONV1X1_SPLIT) else None
            # eval mode on predicted scales (pred_scale: the previous call's maximum of this very launch x 4): the kernel writes the
            # correlation's planes itself; otherwise (training, a measuring call) a split pass on the measured bound
            sp = pred_scale(f1.device) if (mx is not None and _CORR_PRED_PLANES) else None
            cpl = torch.empty((np_, B, h, w, CORR_STRIDE), device=f1.device, dtype=torch.float16) if sp is not None else None
            call("rpnet_local_corr_split_fwd", ptr(f1s), ptr(f2s), ptr(corr), B, h, w, Cc, r, CORR_STRIDE, np_, ptr(s1), ptr(s2),
                 ptr(mx), ptr(cpl), ptr(sp))
            if cpl is not None:
                produced["p16"], produced["scale"] = cpl, sp
            elif mx is not None:
                produced["p16"], produced["scale"] = split_f16(corr, mx, planes=np_, a_is_bound=True)
            ctx.save_for_backward(f1s, f2s, s1, s2)
        elif np_:
            f1s, f2s = _split_operand(o1, np_), _split_operand(o2, np_)
            ARITH[("corr", _PLANE_NAME[np_])] += 1
            call("rpnet_local_corr_split_fwd", ptr(f1s), ptr(f2s), ptr(corr), B, h, w, Cc, r, CORR_STRIDE, np_, None, None, None, None, None)
            ctx.save_for_backward(f1s, f2s)
        else:
            # the fp32 kernels read VALUES: a planes-only operand (a 4-byte placeholder behind f1 / f2) must raise here,
            # not be read out of bounds
            v1, v2 = o1.values(), o2.values()
            ARITH[("corr", "f32")] += 1
            call("rpnet_local_corr_fwd", ptr(v1), ptr(v2), ptr(corr), B, h, w, Cc, r, CORR_STRIDE)
            ctx.save_for_backward(v1, v2)
        ctx.r, ctx.np_, ctx.shape = r, np_, tuple(f1.shape)
        return corr, f1.view_as(f1)

    @staticmethod
    @once_differentiable
    def backward(ctx, dcorr, d_alias):
        f1, f2 = ctx.saved_tensors[:2]
        s1, s2 = ctx.saved_tensors[2:] if ctx.np_ in (1, 2) else (None, None)
        B, h, w, Cc = ctx.shape
        if dcorr is None:
            return d_alias, None, None, None, None
        add = d_alias.contiguous() if d_alias is not None else None
        df1, df2 = _empty(ctx.shape, dcorr), _empty(ctx.shape, dcorr)
        CORR_STRIDE = corr_stride(ctx.r)
        wb = query("rpnet_local_corr_bwd_workspace_bytes", B, h, w, CORR_STRIDE)
        ws = _ws(wb, dcorr)
        ARITH[("corr_bwd", _PLANE_NAME[ctx.np_])] += 1
        if ctx.np_:
            call("rpnet_local_corr_split_bwd", ptr(f1), ptr(f2), ptr(dcorr.contiguous()), ptr(df1), ptr(df2), B, h, w, Cc,
                 ctx.r, CORR_STRIDE, ctx.np_, ptr(s1), ptr(s2), ptr(add), ptr(ws), wb)
        else:
            call("rpnet_local_corr_bwd", ptr(f1), ptr(f2), ptr(dcorr.contiguous()), ptr(df1), ptr(df2), B, h, w, Cc, ctx.r,
                 CORR_STRIDE, ptr(add), ptr(ws), wb)
        return df1, df2, None, None, None


def local_corr(f1, f2, r):
    """LocalCorr on Operands -> (Operand of corr, Operand of the alias of f1 for its other consumer): the correlation with
    its measured fp16 planes (when it ran on fp16 planes), the alias with f1's own planes and scale"""
    o1, o2 = as_operand(f1), as_operand(f2)
    produced = {}
    corr, alias = LocalCorr.apply(o1.x, o2.x, r, (o1, o2), produced)
    return Operand(corr, produced.get("p16"), None, produced.get("scale")), Operand(alias, o1.p16, o1.pbf, o1.scale, o1.planes_only)


# ------------------------------------------------------------------------ matcher
def mask_adjoint(masks, h, w):
    """masks [nmask,B,H,W] -> (am [B,nmask,h*w], msum [B,nmask]): the bilinear-adjoint pooling
    weights of getFeatures (net/rp_net.py:366-376), computed once per forward; no gradient."""
    nmask, B, H, W = masks.shape
    am, msum = _empty((B, nmask, h * w), masks), _empty((B, nmask), masks)
    call("rpnet_mask_adjoint", ptr(masks.contiguous()), ptr(am), ptr(msum), B, nmask, H, W, h, w)
    return am, msum


class MaskedPool(Function):
    """proto[b,k,:] = sum_q f[b,q,:]*am[b,k,q] / (msum[b,k] + 1e-5) (getFeatures, net/rp_net.py:366-376)."""

    @staticmethod
    def forward(ctx, f, am, msum):
        B, h, w, Cc = f.shape
        nmask = am.shape[1]
        proto = _empty((B, nmask, Cc), f)
        wb = query("rpnet_masked_pool_workspace_bytes", B, nmask, h * w, Cc)
        ws = _ws(wb, f)
        call("rpnet_masked_pool_fwd", ptr(f), ptr(am), ptr(msum), ptr(proto), B, nmask, h * w, Cc, ptr(ws), wb)
        ctx.save_for_backward(am, msum)
        ctx.shape = f.shape
        return proto

    @staticmethod
    @once_differentiable
    def backward(ctx, dproto):
        am, msum = ctx.saved_tensors
        B, h, w, Cc = ctx.shape
        df = _empty(ctx.shape, dproto)
        call("rpnet_masked_pool_bwd", ptr(dproto.contiguous()), ptr(am), ptr(msum), ptr(df), B, am.shape[1], h * w, Cc, 0)
        return df, None, None


# the refinement loop's glue as one launch per iteration (rpnet_refine_glue_fwd / _bwd, csrc/refine.hip): BatchNorm + ReLU of
# cre.q, cosine match, bilinear x4, softmax / threshold / 4x4 average and the next call's masked operand planes.
# _GLUE_FUSE = False: the separate launches of rounds 1 - 4 (A/B switch; same bits)
_GLUE_FUSE = True


def glue_supported(K, h, w, H, W, F, C=0, planes=0):
    return bool(_GLUE_FUSE and H == 4 * h and W == 4 * w and query("rpnet_refine_glue_supported", K, h, w, F, C, planes))


class CosineMatchUp(Function):
    """calDist x (1+Wa) -> stack -> F.interpolate(bilinear) (net/rp_net.py:301-303):
    f [B,h,w,C], proto [B,K,C] -> (logits [B,K,H,W], pred [B,K,h,w] (not differentiable here)).
    One launch where the shapes fit the fused kernel (glue_supported), and then optionally more of the loop's glue in the same
    launch — `extra` (a dict, or None):
      "deferred": (y, scale, shift)  f is still unwritten: f = relu(y * scale + shift) is formed and written here
                                     (conv_bn_relu_op(defer_act=True), net/rp_net.py:65-69)
      "mask": True, "soft": bool     also softmax(1)[:, 1] -> > 0.5 unless soft -> avg_pool2d(4) (net/rp_net.py:308-311);
                                     the next mask [B,h,w] comes back as extra["mask_out"]
      "x": tensor [B,h,w,Cx], "x_scale": device scalar or None, "planes": 1 / 2 (fp16) or 3 (bf16)
                                     also the operand planes of x * mask and x * (1 - mask) (net/rp_net.py:283) ->
                                     extra["xk"], extra["xq"] [planes,B,h,w,Cx]"""

    @staticmethod
    def forward(ctx, f, proto, H, W, scaler, extra=None):
        ctx.set_materialize_grads(False)      # `pred` is not differentiable: no zero tensor for it in backward
        B, h, w, Cc = f.shape
        K = proto.shape[1]
        proto = proto.contiguous()
        pred = _empty((B, K, h, w), f)
        logits = _empty((B, K, H, W), f)
        ex = extra if extra is not None else {}
        x = ex.get("x")
        planes = int(ex.get("planes") or 0) if x is not None else 0
        ctx.fused = glue_supported(K, h, w, H, W, Cc, x.shape[-1] if x is not None else 0, planes)
        if ctx.fused:
            d = ex.get("deferred")
            mask = _empty((B, h, w), f) if ex.get("mask") else None
            xk = xq = None
            if x is not None and mask is not None:
                dt = torch.bfloat16 if planes == 3 else torch.float16
                xk = torch.empty((planes,) + tuple(x.shape), device=f.device, dtype=dt)
                xq = torch.empty((planes,) + tuple(x.shape), device=f.device, dtype=dt)
            call("rpnet_refine_glue_fwd", ptr(d[0] if d else f), ptr(d[1]) if d else None, ptr(d[2]) if d else None, ptr(proto),
                 float(scaler), ptr(f) if d else None, ptr(pred), ptr(logits), ptr(mask), 1 if ex.get("soft") else 0,
                 ptr(x) if xk is not None else None, ptr(ex.get("x_scale")), ptr(xk), ptr(xq), planes, B, K, h, w, Cc,
                 x.shape[-1] if x is not None else 0)
            if mask is not None:
                ex["mask_out"], ex["xk"], ex["xq"] = mask, xk, xq
        else:
            if ex.get("deferred") or ex.get("mask"):
                raise RuntimeError("rpnet_amd: CosineMatchUp was asked for the fused glue on shapes the kernel does not take "
                                   "(check glue_supported before deferring the activation)")
            call("rpnet_cosine_match_fwd", ptr(f), ptr(proto), ptr(pred), B, K, h * w, Cc, float(scaler))
            call("rpnet_bilinear_up_fwd", ptr(pred), ptr(logits), B * K, h, w, H, W)
        ctx.save_for_backward(f, proto)
        ctx.cfg = (H, W, float(scaler))
        ctx.mark_non_differentiable(pred)
        return logits, pred

    @staticmethod
    @once_differentiable
    def backward(ctx, dlogits, _dpred):
        if dlogits is None:
            return None, None, None, None, None, None
        f, proto = ctx.saved_tensors
        H, W, scaler = ctx.cfg
        B, h, w, Cc = f.shape
        K = proto.shape[1]
        df, dproto = torch.empty_like(f), torch.empty_like(proto)
        if ctx.fused:
            wb = query("rpnet_refine_glue_bwd_workspace_bytes", B, K, h, w, Cc)
            ws = _ws(wb, f)
            call("rpnet_refine_glue_bwd", ptr(dlogits.contiguous()), ptr(f), ptr(proto), scaler, ptr(df), ptr(dproto), B, K, h, w,
                 Cc, ptr(ws), wb)
            return df, dproto, None, None, None, None
        dpred = _empty((B, K, h, w), f)
        call("rpnet_bilinear_up_bwd", ptr(dlogits.contiguous()), ptr(dpred), B * K, h, w, H, W)
        wb = query("rpnet_cosine_match_bwd_workspace_bytes", B, K, h * w, Cc)
        ws = _ws(wb, f)
        call("rpnet_cosine_match_bwd", ptr(f), ptr(proto), ptr(dpred), ptr(df), ptr(dproto), B, K, h * w, Cc, scaler, 0,
             ptr(ws), wb)
        return df, dproto, None, None, None, None


class SoftmaxPool(Function):
    """soft_mask: True — avg_pool2d(softmax(logits, 1)[:, 1], scale) kept differentiable (net/rp_net.py:308-311)."""

    @staticmethod
    def forward(ctx, logits, scale):
        B, K, H, W = logits.shape
        out = _empty((B, H // scale, W // scale), logits)
        call("rpnet_softmax_thresh_pool", ptr(logits), ptr(out), B, K, H, W, scale, 1)
        ctx.save_for_backward(logits)
        ctx.scale = scale
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dmask):
        (logits,) = ctx.saved_tensors
        B, K, H, W = logits.shape
        dl = torch.empty_like(logits)
        call("rpnet_softmax_pool_bwd", ptr(logits), ptr(dmask.contiguous()), ptr(dl), B, K, H, W, ctx.scale)
        return dl, None


def softmax_thresh_pool(logits, scale, soft):
    """softmax(1)[:,1] -> (>0.5) unless soft -> avg_pool2d(scale) (net/rp_net.py:308-311); no gradient."""
    B, K, H, W = logits.shape
    out = _empty((B, H // scale, W // scale), logits)
    call("rpnet_softmax_thresh_pool", ptr(logits), ptr(out), B, K, H, W, scale, 1 if soft else 0)
    return out


# ------------------------------------------------------------------------- losses
class DiceCE(Function):
    """dice_ce (net/rp_net.py:123-127); with_dice=False/ignore_index/per_sample give the
    F.cross_entropy(ignore_index=255) term of alignLoss (net/rp_net.py:438)."""

    @staticmethod
    def forward(ctx, logits, labels, with_dice, ignore_index, per_sample, sample_weight):
        B, K, H, W = logits.shape
        logits = logits.contiguous()
        labels = labels.contiguous()
        loss = _empty((), logits)
        stats = _empty(((B + 1) * (2 * K + 2),), logits)
        wb = query("rpnet_loss_workspace_bytes", B, K, H, W)
        ws = _ws(wb, logits)
        call("rpnet_dice_ce_fwd", ptr(logits), ptr(labels), ptr(loss), ptr(stats), B, K, H, W, int(with_dice),
             int(ignore_index), int(per_sample), ptr(sample_weight), ptr(ws), wb)
        ctx.save_for_backward(logits, labels, stats, sample_weight)
        ctx.cfg = (int(with_dice), int(ignore_index), int(per_sample))
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        logits, labels, stats, sample_weight = ctx.saved_tensors
        with_dice, ignore_index, per_sample = ctx.cfg
        B, K, H, W = logits.shape
        dl = torch.empty_like(logits)
        call("rpnet_dice_ce_bwd", ptr(logits), ptr(labels), ptr(stats), ptr(g.contiguous()), ptr(dl), B, K, H, W, with_dice,
             ignore_index, per_sample, ptr(sample_weight), 0)
        return dl, None, None, None, None, None


def dice_ce(logits, true, eps=1e-7):
    """Drop-in for net.rp_net.dice_ce on GPU tensors (eps is fixed at the reference's 1e-7)."""
    return DiceCE.apply(logits, true, True, -1, False, None)


_DICE_MULTI = True


class DiceCESum(Function):
    """sum_i dice_ce(logits_i, labels) over n <= 16 tensors of one shape in two launches (rpnet_dice_ce_multi_fwd) and one
    backward launch: the training objective evaluates dice_ce on the final output and on every refinement iteration's
    output, back to back at the end of the forward pass — 3 n launches of 7 - 11 us each with nothing beside them."""

    @staticmethod
    def forward(ctx, labels, *logits):
        B, K, H, W = logits[0].shape
        logits = tuple(t.contiguous() for t in logits)
        labels = labels.contiguous()
        n = len(logits)
        loss = _empty((n + 1,), logits[0])
        stats = _empty((n * (B + 1) * (2 * K + 2),), logits[0])
        wb = n * query("rpnet_loss_workspace_bytes", B, K, H, W)
        ws = _ws(wb, logits[0])
        arr = (C.c_void_p * n)(*[t.data_ptr() for t in logits])
        call("rpnet_dice_ce_multi_fwd", arr, n, ptr(labels), ptr(loss), ptr(stats), B, K, H, W, ptr(ws), wb)
        ctx.save_for_backward(labels, stats, *logits)
        ctx.per_tensor = loss[:n]
        return loss[n]

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        labels, stats, *logits = ctx.saved_tensors
        B, K, H, W = logits[0].shape
        n = len(logits)
        dl = torch.empty((n, B, K, H, W), device=logits[0].device, dtype=torch.float32)
        la = (C.c_void_p * n)(*[t.data_ptr() for t in logits])
        da = (C.c_void_p * n)(*[dl[i].data_ptr() for i in range(n)])
        call("rpnet_dice_ce_multi_bwd", la, da, n, ptr(labels), ptr(stats), ptr(g.contiguous()), B, K, H, W)
        return (None,) + tuple(dl.unbind(0))


def dice_ce_sum(logits, true):
    """sum of dice_ce(l, true) over the tensors of `logits` (one shape); every loss term bit-identical to dice_ce's"""
    logits = list(logits)
    total = None
    if not _DICE_MULTI:          # A/B switch: one dice_ce call per tensor, as before
        for t in logits:
            total = dice_ce(t, true) if total is None else total + dice_ce(t, true)
        return total
    for i in range(0, len(logits), 16):
        chunk = logits[i:i + 16]
        part = dice_ce(chunk[0], true) if len(chunk) == 1 else DiceCESum.apply(true, *chunk)
        total = part if total is None else total + part
    return total


class Objective(Function):
    """The training objective in two launches forward, one backward (rpnet_objective_fwd / _bwd):
    sum_i w_i dice_ce(logits_i, labels) + extra_scale * extra — w_i the multiplicity of tensor i in the caller's list (the final output IS
    the last refinement iteration's output, net/rp_net.py:314-337), extra the align loss (net/rp_net.py:394-440) with the yaml's
    align_loss_scaler.  Replaces DiceCESum + two scalar tensor operations + the autograd add of the duplicate's two gradients (and their
    backward nodes): same values, bit for bit."""

    @staticmethod
    def forward(ctx, labels, extra, extra_scale, weights, *logits):
        B, K, H, W = logits[0].shape
        logits = tuple(t.contiguous() for t in logits)
        labels = labels.contiguous()
        n = len(logits)
        loss = _empty((n + 1,), logits[0])
        stats = _empty((n * (B + 1) * (2 * K + 2),), logits[0])
        wb = n * query("rpnet_loss_workspace_bytes", B, K, H, W)
        ws = _ws(wb, logits[0])
        arr = (C.c_void_p * n)(*[t.data_ptr() for t in logits])
        warr = (C.c_float * n)(*[float(w) for w in weights])
        call("rpnet_objective_fwd", arr, warr, n, ptr(labels), ptr(extra), float(extra_scale), ptr(loss), ptr(stats), B, K, H, W,
             ptr(ws), wb)
        ctx.save_for_backward(labels, stats, *logits)
        ctx.cfg = (tuple(float(w) for w in weights), float(extra_scale), extra is not None and extra.requires_grad)
        ctx.per_tensor = loss[:n]
        return loss[n]

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        labels, stats, *logits = ctx.saved_tensors
        weights, extra_scale, want_extra = ctx.cfg
        B, K, H, W = logits[0].shape
        n = len(logits)
        dl = torch.empty((n, B, K, H, W), device=logits[0].device, dtype=torch.float32)
        dextra = _empty((), logits[0]) if want_extra else None
        la = (C.c_void_p * n)(*[t.data_ptr() for t in logits])
        da = (C.c_void_p * n)(*[dl[i].data_ptr() for i in range(n)])
        warr = (C.c_float * n)(*weights)
        call("rpnet_objective_bwd", la, da, warr, n, ptr(labels), ptr(stats), ptr(g.contiguous()), ptr(dextra), extra_scale, B, K, H, W)
        return (None, dextra, None, None) + tuple(dl.unbind(0))


def objective(logits, true, extra=None, extra_scale=1.0):
    """sum_i dice_ce(logits_i, true) + extra_scale * extra (extra: a scalar tensor, or None / a Python number for "no such term") —
    what train_rpnet.py and bench.py minimise; a tensor listed twice is evaluated once with weight 2"""
    uniq, weights = [], []
    for t in logits:
        for j, u in enumerate(uniq):
            if u is t:
                weights[j] += 1.0
                break
        else:
            uniq.append(t)
            weights.append(1.0)
    ex = extra if torch.is_tensor(extra) else None
    if not _DICE_MULTI or len(uniq) > 16 or not uniq[0].is_cuda:
        total = dice_ce_sum(logits, true)
        return total if ex is None else total + extra_scale * ex
    if ex is not None:
        ex = ex.reshape(())
        if ex.dtype != torch.float32:
            ex = ex.float()
    return Objective.apply(true, ex, float(extra_scale), tuple(weights), *uniq)


_SEEDS = {}


def backward(loss):
    """loss.backward() with a cached gradient seed (autograd otherwise fills a fresh ones_like(loss) every step: one launch)"""
    key = (loss.device, loss.dtype, tuple(loss.shape))
    seed = _SEEDS.get(key)
    if seed is None:
        seed = _SEEDS[key] = torch.ones(loss.shape, device=loss.device, dtype=loss.dtype)
    loss.backward(gradient=seed)


def argmax_masks(pred, want_keep=False):
    """pred [B,K,h,w] -> (masks [B,K,h*w] one-hot of argmax, counts [B,K][, keep [K,B] = 1 where counts > 0: the reference's
    per-episode skip of a way whose predicted mask is empty]) (net/rp_net.py:412-415,421)."""
    B, K, h, w = pred.shape
    masks, counts = _empty((B, K, h * w), pred), _empty((B, K), pred)
    keep = _empty((K, B), pred) if want_keep else None
    call("rpnet_argmax_masks", ptr(pred), ptr(masks), ptr(counts), ptr(keep), B, K, h * w)
    return (masks, counts, keep) if want_keep else (masks, counts)


def align_labels(fore, back):
    """1 = fore, 0 = back (wins), 255 = ignore (net/rp_net.py:433-436)."""
    lab = torch.empty(fore.shape, device=fore.device, dtype=torch.int64)
    call("rpnet_align_labels", ptr(fore.contiguous()), ptr(back.contiguous()), ptr(lab), fore.numel())
    return lab
