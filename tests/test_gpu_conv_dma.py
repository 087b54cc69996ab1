"""The LDS-DMA patch kernel (rpnet_amd/csrc/conv_split_dma.hip, tile variant 11 of rpnet_conv_fwd: 3x3 forward / input
gradient on two fp16 planes, four waves, operands by buffer_load ... lds) against
  * the register-staged 8-wave patch kernel (variant 7): same tiles, same K order, same MFMA sequence per accumulator, same
    epilogue -> the outputs must be BIT-IDENTICAL (forward, BatchNorm statistics, both input gradients);
  * the torch fp64 reference of the same layer (1e-3 bar of BASELINE.json; measured ~1e-6).
Shapes cover both patch widths (W % 32 == 0 -> 8 x 32 patches, else 16 x 16), image borders in every direction, two
concatenated sources, the nearest x2 up-sampling in the gather, two BatchNorm statistic groups, 128 / 256 / 384 output
channels and two input-gradient destinations."""
import copy
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from tests.helpers import rel_err, rnd
from tests.test_gpu_ops import _mk_layer, nchw, nhwc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def RF():
    from rpnet_amd import functional
    return functional


def _run(RF, monkeypatch, tile, layer, a, b, go, ups, groups):
    """conv-BN-ReLU of cat([a, b]) (fp16 planes: the sources carry a tensor scale) forward + backward with tile variant
    `tile` forced; -> (z, da, db, running_var, the variants rpnet_conv_fwd actually chose)"""
    from rpnet_amd import hip
    monkeypatch.setitem(RF.TUNE, "tile", tile + 1)
    if tile >= 0:       # a forced tile variant of rpnet_conv_fwd: the up-sampling layers on their nine-product form
        monkeypatch.setattr(RF, "_UP4", False)
    chosen = []
    orig = RF.call

    def spy(name, *args):
        if name == "rpnet_conv_fwd":
            chosen.append(hip.query("rpnet_conv_tile_variant", args[0]))
        return orig(name, *args)

    monkeypatch.setattr(RF, "call", spy)
    conv, bn = copy.deepcopy(layer[0]).to(DEV), copy.deepcopy(layer[1]).to(DEV).train()
    ag = nhwc(a).to(DEV).requires_grad_(True)
    bg = nhwc(b).to(DEV).requires_grad_(True) if b is not None else None
    bound = max(a.abs().max().item(), b.abs().max().item() if b is not None else 0.0)
    sc = torch.tensor([2.0 ** (int(torch.ceil(torch.log2(torch.tensor(bound))).item()) - 15)], device=DEV)
    oa = RF.Operand(ag, scale=sc)
    ob = RF.Operand(bg, scale=sc) if bg is not None else None
    z = RF.conv_bn_relu_op(oa, conv, bn, RF.WeightCache(), True, x1=ob, groups=groups, upsample=ups, out_split=False).x
    z.backward(nhwc(go).to(DEV))
    torch.cuda.synchronize()
    monkeypatch.setattr(RF, "call", orig)
    return z.detach(), ag.grad, (bg.grad if bg is not None else None), bn.running_var.clone(), conv.weight.grad, chosen


CASES = [
    # N, H, W, c0, c1, cout, ups
    (2, 32, 32, 128, 0, 128, False),     # 8 x 32 patches, 4 per image: top / bottom borders inside every patch column
    (1, 16, 64, 128, 128, 256, False),   # two sources, two column blocks, patches side by side (left / right borders)
    (2, 32, 64, 64, 0, 128, True),       # nearest x2 in the gather
    (3, 16, 48, 128, 128, 128, False),   # W % 32 != 0 -> 16 x 16 patches, odd image count (one statistic group)
    (2, 16, 16, 256, 0, 384, False),     # one 16 x 16 patch per image, 3 column blocks, 8 channel chunks
    (4, 8, 32, 64, 0, 128, False),       # two channel chunks (K = 18 steps): little more than prologue and tail of the DMA ring
]


@pytest.mark.parametrize("variant", [12, 14])
@pytest.mark.parametrize("N,H,W,c0,c1,cout,ups", CASES + [(2, 32, 32, 64, 0, 64, False), (1, 16, 32, 128, 64, 192, False)])
def test_dma_patch_kernel_64_wide_tiles(RF, monkeypatch, N, H, W, c0, c1, cout, ups, variant):
    """variants 12 / 14 (256 x 64 and 128 x 64 tiles of the same kernel: Cout = 64 layers, small grids): other tiles than any
    register-staged kernel, so the check is the fp64 reference (and the 8-wave / 4-wave patch kernels to 1e-5)"""
    old = RF.conv_math()
    RF.set_conv_math("f16x2")
    try:
        groups = 2 if N % 2 == 0 else 1
        layer = _mk_layer(c0 + c1, cout, 3, 91)
        hs, ws = (H // 2, W // 2) if ups else (H, W)
        a = rnd(92, N, c0, hs, ws)
        b = rnd(93, N, c1, hs, ws) if c1 else None
        go = rnd(94, N, cout, H, W)
        c_ref, b_ref = copy.deepcopy(layer[0]).double(), copy.deepcopy(layer[1]).double().train()
        ar = a.double().requires_grad_(True)
        br = b.double().requires_grad_(True) if c1 else None
        xin = torch.cat([ar, br], 1) if c1 else ar
        if ups:
            xin = F.interpolate(xin, scale_factor=2, mode="nearest")
        per = N // groups
        ref = torch.cat([F.relu(b_ref(c_ref(xin[g * per:(g + 1) * per]))) for g in range(groups)], 0)
        ref.backward(go.double())
        z0, da0, db0, rv0, dw0, ch0 = _run(RF, monkeypatch, 9, layer, a, b, go, ups, groups)
        z, da, db, rv, dw, ch = _run(RF, monkeypatch, variant, layer, a, b, go, ups, groups)
        assert ch == [variant, variant], ch
        assert rel_err(z, z0) < 1e-5 and rel_err(da, da0) < 1e-5
        assert rel_err(nchw(z), ref) < 1e-3
        assert rel_err(nchw(da), ar.grad) < 1e-3
        if c1:
            assert rel_err(nchw(db), br.grad) < 1e-3
        assert rel_err(rv, b_ref.running_var) < 1e-5
    finally:
        RF.set_conv_math(old)


@pytest.mark.parametrize("N,H,W,c0,c1,cout,ups", CASES)
def test_dma_patch_kernel_bit_identical_and_accurate(RF, monkeypatch, N, H, W, c0, c1, cout, ups):
    old = RF.conv_math()
    RF.set_conv_math("f16x2")
    try:
        groups = 2 if N % 2 == 0 else 1
        layer = _mk_layer(c0 + c1, cout, 3, 91)
        hs, ws = (H // 2, W // 2) if ups else (H, W)
        a = rnd(92, N, c0, hs, ws)
        b = rnd(93, N, c1, hs, ws) if c1 else None
        go = rnd(94, N, cout, H, W)
        # fp64 reference
        c_ref, b_ref = copy.deepcopy(layer[0]).double(), copy.deepcopy(layer[1]).double().train()
        ar = a.double().requires_grad_(True)
        br = b.double().requires_grad_(True) if c1 else None
        xin = torch.cat([ar, br], 1) if c1 else ar
        if ups:
            xin = F.interpolate(xin, scale_factor=2, mode="nearest")
        per = N // groups
        ref = torch.cat([F.relu(b_ref(c_ref(xin[g * per:(g + 1) * per]))) for g in range(groups)], 0)
        ref.backward(go.double())

        z7, da7, db7, rv7, dw7, ch7 = _run(RF, monkeypatch, 7, layer, a, b, go, ups, groups)
        z11, da11, db11, rv11, dw11, ch11 = _run(RF, monkeypatch, 11, layer, a, b, go, ups, groups)
        # forward launch + input-gradient launch, each on the forced variant (the input gradient only when its output
        # channels = the layer's input channels come in 128-wide tiles per destination)
        dgrad_ok = c0 % 128 == 0 and c1 % 128 == 0
        assert ch7[0] == 7 and ch11[0] == 11 and len(ch11) == 2, (ch7, ch11)
        assert (ch11[1] == 11) == dgrad_ok and (ch7[1] == 7) == dgrad_ok, (ch7, ch11)
        assert RF.arith_counts()["conv3x3"].get("f16x2", 0) >= 2
        assert torch.equal(z7, z11)
        assert torch.equal(da7, da11)
        assert torch.equal(rv7, rv11)
        if c1:
            assert torch.equal(db7, db11)
        assert rel_err(nchw(z11), ref) < 1e-3
        assert rel_err(nchw(da11), ar.grad) < 1e-3
        if c1:
            assert rel_err(nchw(db11), br.grad) < 1e-3
        assert rel_err(dw11, c_ref.weight.grad) < 1e-3
        assert rel_err(rv11, b_ref.running_var) < 1e-5
    finally:
        RF.set_conv_math(old)


def test_dma_patch_kernel_repeatable(RF, monkeypatch):
    """the counted waits of the DMA ring leave two K-steps of loads in flight across every barrier: a misplaced wait shows up
    as a rare stale tile, so the same launch is repeated and every result compared bit by bit"""
    old = RF.conv_math()
    RF.set_conv_math("f16x2")
    try:
        layer = _mk_layer(256, 256, 3, 95)
        a, go = rnd(96, 8, 256, 32, 32), rnd(97, 8, 256, 32, 32)
        first = _run(RF, monkeypatch, 11, layer, a, None, go, False, 2)
        for _ in range(5):
            again = _run(RF, monkeypatch, 11, layer, a, None, go, False, 2)
            assert torch.equal(first[0], again[0]) and torch.equal(first[1], again[1])
    finally:
        RF.set_conv_math(old)


@pytest.mark.parametrize("N,H,W,c0,c1,cout,ups", [
    (2, 32, 32, 128, 0, 128, False),     # power-of-two image, 4 tiles, split-K over 64 K-steps; one 32-pixel block per row
    (1, 16, 64, 128, 128, 256, False),   # two sources (a tile lies in one of them); two blocks per row
    (2, 32, 64, 64, 0, 128, True),       # nearest x2: the x strips come from the half-resolution source
    (3, 16, 48, 128, 128, 128, False),   # W not a power of two (division path), odd image count
    (2, 8, 8, 64, 0, 64, False),         # M = 128: four K-steps in all — prologue and tail of the ring only
    (4, 4, 4, 128, 0, 64, False),        # image rows of 4 pixels (the deepest level of a 64 x 64 episode): a DMA piece spans two rows
    (2, 4, 8, 64, 0, 64, False),         # H != W, rows of exactly one piece
    (5, 24, 40, 64, 0, 192, False),      # M = 4800 = 150 K-steps, image rows wider than a K-step and not a multiple of it
    (2, 64, 64, 64, 0, 64, False),       # one tile -> 32 splits of 8 steps: every split range begins and ends INSIDE an image column
    (1, 16, 128, 64, 0, 128, False),     # four blocks per row (interior blocks have both neighbours), 16-row columns
    (3, 8, 32, 64, 64, 64, False),       # odd image count, columns of 8 rows: more column changes than ring slots in a range
    (2, 2, 32, 64, 0, 64, False),        # two-row images: every step has a zero strip above or below
    (1, 1, 64, 64, 0, 64, False),        # one-row image: both neighbours zero in every step; two steps in all
])
def test_dma_weight_gradient_bit_identical_and_accurate(RF, monkeypatch, N, H, W, c0, c1, cout, ups):
    """conv_wgrad9_dma_kernel (conv_wgrad_split_dma.hip, tune 16) against the register-staged 12-wave kernel of round 2 (tune 8):
    same tiles, split-K plan and order of accumulation -> bit-identical dW.  The default of round 6 — the same kernel with its
    K-steps walked down the image columns (conv_wgrad_ring.hip; power-of-two images at least 32 pixels wide, no up-sampling) —
    sums a split's pixels in another order: equal to fp32 summation round-off.  All against the fp64 reference."""
    old = RF.conv_math()
    RF.set_conv_math("f16x2")
    try:
        groups = 2 if N % 2 == 0 else 1
        layer = _mk_layer(c0 + c1, cout, 3, 81)
        hs, ws = (H // 2, W // 2) if ups else (H, W)
        a = rnd(82, N, c0, hs, ws)
        b = rnd(83, N, c1, hs, ws) if c1 else None
        go = rnd(84, N, cout, H, W)
        c_ref, b_ref = copy.deepcopy(layer[0]).double(), copy.deepcopy(layer[1]).double().train()
        ar = a.double().requires_grad_(True)
        br = b.double().requires_grad_(True) if c1 else None
        xin = torch.cat([ar, br], 1) if c1 else ar
        if ups:
            xin = F.interpolate(xin, scale_factor=2, mode="nearest")
        per = N // groups
        ref = torch.cat([F.relu(b_ref(c_ref(xin[g * per:(g + 1) * per]))) for g in range(groups)], 0)
        ref.backward(go.double())
        monkeypatch.setitem(RF.TUNE, "wgrad", 8)
        dw_old = _run(RF, monkeypatch, -1, layer, a, b, go, ups, groups)[4]
        monkeypatch.setitem(RF.TUNE, "wgrad", 16)
        dw_row = _run(RF, monkeypatch, -1, layer, a, b, go, ups, groups)[4]
        monkeypatch.setitem(RF.TUNE, "wgrad", 0)
        dw_new = _run(RF, monkeypatch, -1, layer, a, b, go, ups, groups)[4]
        assert RF.arith_counts()["wgrad3x3"].get("f16x2", 0) >= 1
        assert torch.equal(dw_old, dw_row)
        # (rows of exactly one K-step, W == 32, or one-row images: the column-major order IS the row-major order — the ring kernel
        # then has to reproduce the row-major kernel's bits, which checks its strip ring, zero strips and waits against a known answer)
        ring = (not ups) and W >= 32 and (W & (W - 1)) == 0 and (H & (H - 1)) == 0
        if ring and (W == 32 or H == 1):
            assert torch.equal(dw_row, dw_new)
        elif ring:
            assert not torch.equal(dw_row, dw_new)          # another kernel ran (another summation order)
            assert rel_err(dw_new, dw_row) < 2e-6
            again = _run(RF, monkeypatch, -1, layer, a, b, go, ups, groups)[4]
            assert torch.equal(again, dw_new)               # (the ring's counted waits: a stale strip would show up here)
        else:
            assert torch.equal(dw_row, dw_new)
        assert rel_err(dw_new, c_ref.weight.grad) < 1e-3
        assert rel_err(dw_new, c_ref.weight.grad) < 1.05 * rel_err(dw_row, c_ref.weight.grad) + 1e-6
    finally:
        RF.set_conv_math(old)


@pytest.mark.parametrize("N,H,W,c0,c1,cout,ups", [
    (2, 32, 32, 128, 0, 128, False),
    (1, 16, 64, 128, 128, 256, False),   # two sources
    (2, 32, 64, 64, 0, 128, True),       # nearest x2, ONE 64-channel K chunk
    (3, 16, 48, 128, 128, 128, False),   # 16 x 16 patches
])
def test_dma_patch_kernel_one_plane(RF, monkeypatch, N, H, W, c0, c1, cout, ups):
    """variant 13: the LDS-DMA kernel on ONE fp16 plane (the f16 arithmetic of BASELINE configs[4]; a K-step = one tap of 64
    channels whose halves sit where the two planes of f16x2 do).  Same products as the 8-wave one-plane patch kernel (variant 7)
    in another order: equal to fp32 summation round-off; and within the f16 tolerance of the fp64 reference."""
    old = RF.conv_math()
    RF.set_conv_math("f16")
    try:
        groups = 2 if N % 2 == 0 else 1
        layer = _mk_layer(c0 + c1, cout, 3, 71)
        hs, ws = (H // 2, W // 2) if ups else (H, W)
        a = rnd(72, N, c0, hs, ws)
        b = rnd(73, N, c1, hs, ws) if c1 else None
        go = rnd(74, N, cout, H, W)
        z0, da0, db0, rv0, dw0, ch0 = _run(RF, monkeypatch, 7, layer, a, b, go, ups, groups)
        z, da, db, rv, dw, ch = _run(RF, monkeypatch, 13, layer, a, b, go, ups, groups)
        dgrad_ok = c0 % 128 == 0 and c1 % 128 == 0
        assert ch[0] == 13 and (ch[1] == 13) == dgrad_ok, ch
        assert RF.arith_counts()["conv3x3"].get("f16", 0) >= 2
        # (the input gradient passes the BatchNorm backward, whose dy leaves as ONE fp16 plane: a forward difference of one
        # fp32 ulp flips some of those roundings, each worth 2^-11 of an element — 3e-5 of the tensor's maximum measured)
        assert rel_err(z, z0) < 1e-5 and rel_err(da, da0) < 2e-4 and rel_err(rv, rv0) < 1e-5
        if c1:
            assert rel_err(db, db0) < 2e-4
        c_ref, b_ref = copy.deepcopy(layer[0]).double(), copy.deepcopy(layer[1]).double().train()
        ar = a.double().requires_grad_(True)
        br = b.double().requires_grad_(True) if c1 else None
        xin = torch.cat([ar, br], 1) if c1 else ar
        if ups:
            xin = F.interpolate(xin, scale_factor=2, mode="nearest")
        per = N // groups
        ref = torch.cat([F.relu(b_ref(c_ref(xin[g * per:(g + 1) * per]))) for g in range(groups)], 0)
        assert rel_err(nchw(z), ref) < 1e-2          # fp16 operands (F16_LOGIT_TOL class): measured ~2e-3
    finally:
        RF.set_conv_math(old)


@pytest.mark.parametrize("N,H,W,c0,c1,cout", [
    (2, 16, 64, 128, 0, 128),      # one 64-pixel step per image row: every step has both a left and a right border
    (1, 8, 128, 64, 64, 64),       # two sources, two steps per row (border on one side each)
    (2, 64, 64, 64, 0, 192),       # square images, three column tiles
    (4, 4, 256, 128, 0, 64),       # four steps per row, top / bottom borders in every second row
])
def test_dma_weight_gradient_one_plane(RF, monkeypatch, N, H, W, c0, c1, cout):
    """conv_wgrad9_dma_kernel<.., K64>: the weight gradient on ONE fp16 plane (f16 arithmetic of BASELINE configs[4]) through the
    LDS-DMA kernel — a K-step = 64 pixels of an image row, the halves where the two planes of f16x2 sit.  Same products as the
    register-staged one-plane kernel (tune 8) in another order: equal to fp32 summation round-off; and within the f16
    error of that kernel against the fp64 reference."""
    old = RF.conv_math()
    RF.set_conv_math("f16")
    try:
        groups = 2 if N % 2 == 0 else 1
        layer = _mk_layer(c0 + c1, cout, 3, 61)
        a = rnd(62, N, c0, H, W)
        b = rnd(63, N, c1, H, W) if c1 else None
        go = rnd(64, N, cout, H, W)
        c_ref, b_ref = copy.deepcopy(layer[0]).double(), copy.deepcopy(layer[1]).double().train()
        ar = a.double().requires_grad_(True)
        br = b.double().requires_grad_(True) if c1 else None
        xin = torch.cat([ar, br], 1) if c1 else ar
        per = N // groups
        ref = torch.cat([F.relu(b_ref(c_ref(xin[g * per:(g + 1) * per]))) for g in range(groups)], 0)
        ref.backward(go.double())
        monkeypatch.setitem(RF.TUNE, "wgrad", 8)
        dw_old = _run(RF, monkeypatch, -1, layer, a, b, go, False, groups)[4]
        monkeypatch.setitem(RF.TUNE, "wgrad", 16)           # round 5's row-major K order
        dw_row = _run(RF, monkeypatch, -1, layer, a, b, go, False, groups)[4]
        monkeypatch.setitem(RF.TUNE, "wgrad", 0)            # round 6: K-steps down the image columns (conv_wgrad_ring.hip)
        dw_new = _run(RF, monkeypatch, -1, layer, a, b, go, False, groups)[4]
        assert RF.arith_counts()["wgrad3x3"].get("f16", 0) >= 1
        assert not torch.equal(dw_old, dw_new)               # another kernel ran (another summation order)
        if W == 64 or H == 1:   # one 64-pixel step per image row: column-major IS row-major — the ring kernel must reproduce the bits
            assert torch.equal(dw_row, dw_new)
        else:
            assert not torch.equal(dw_row, dw_new)
        assert rel_err(dw_new, dw_old) < 2e-6 and rel_err(dw_row, dw_old) < 2e-6
        # (one fp16 plane of dy behind a BatchNorm backward: per-element roundings of 2^-11 of the TENSOR maximum — the
        # error against fp64 is that of the arithmetic, the same for both kernels)
        e_new, e_old = rel_err(dw_new, c_ref.weight.grad), rel_err(dw_old, c_ref.weight.grad)
        assert e_new < 1.02 * e_old + 1e-5 and e_new < 0.1, (e_new, e_old)
    finally:
        RF.set_conv_math(old)


@pytest.mark.parametrize("N,H,W,c0,c1,cout,ups", [
    (1, 16, 16, 160, 352, 512, False),    # 8 tiles -> 8 parts of two 32-channel chunks; part 2 straddles the two sources
    (4, 16, 16, 1024, 0, 1024, False),    # the Conv5 layer of a batch-2 eval call: 64 tiles -> 4 parts
    (2, 32, 32, 1024, 0, 512, True),      # nearest x2 in the gather (Up5 of that call at batch 1): 64 tiles -> 4 parts
    (1, 32, 64, 128, 128, 128, False),    # 16 tiles, 8 chunks -> 4 parts of two chunks
])
def test_dma_patch_kernel_split_k(RF, N, H, W, c0, c1, cout, ups):
    """rpnet_conv_desc.splitk_ws: eval-mode conv + folded BatchNorm + ReLU on a grid of a quarter of the CUs or fewer, K range
    cut into parts (fp32 partial tiles + the reduce launch that does the epilogue) against one block per tile: equal to
    fp32 summation round-off, max |output| and the fp64 reference as before."""
    old, old_sk = RF.conv_math(), RF._EVAL_SPLITK
    RF.set_conv_math("f16x2")
    try:
        layer = _mk_layer(c0 + c1, cout, 3, 51)
        hs, ws = (H // 2, W // 2) if ups else (H, W)
        a = rnd(52, N, c0, hs, ws)
        b = rnd(53, N, c1, hs, ws) if c1 else None
        c_ref, b_ref = copy.deepcopy(layer[0]).double(), copy.deepcopy(layer[1]).double().eval()
        xin = torch.cat([a, b], 1).double() if c1 else a.double()
        if ups:
            xin = F.interpolate(xin, scale_factor=2, mode="nearest")
        with torch.no_grad():
            ref = F.relu(b_ref(c_ref(xin)))
        bound = max(a.abs().max().item(), b.abs().max().item() if c1 else 0.0)
        sc = torch.tensor([2.0 ** (int(torch.ceil(torch.log2(torch.tensor(bound))).item()) - 15)], device=DEV)
        outs, lent = {}, {}
        for on in (False, True):
            RF._EVAL_SPLITK = on
            conv, bn = copy.deepcopy(layer[0]).to(DEV), copy.deepcopy(layer[1]).to(DEV).eval()
            seen, orig = [], RF.call

            def spy(name, *args):
                if name == "rpnet_conv_fwd":
                    seen.append(int(args[0]._obj.splitk_ws_bytes))
                return orig(name, *args)

            RF.call = spy
            try:
                with torch.no_grad():
                    oa = RF.Operand(nhwc(a).to(DEV), scale=sc)
                    ob = RF.Operand(nhwc(b).to(DEV), scale=sc) if c1 else None
                    op = RF.conv_bn_relu_op(oa, conv, bn, RF.WeightCache(), False, x1=ob, upsample=ups, out_split=True)
                    torch.cuda.synchronize()
            finally:
                RF.call = orig
            outs[on], lent[on] = (op.x.clone(), op.p16.clone() if op.p16 is not None else None, op.scale.clone()), seen
        assert lent[False] == [0] and lent[True][0] >= 2 * N * H * W * cout * 4, lent
        z0, p0, s0 = outs[False]
        z1, p1, s1 = outs[True]
        assert not torch.equal(z0, z1) and rel_err(z1, z0) < 1e-5       # fp32 sums of up to 9216 products in another order
        assert torch.equal(s0, s1)                          # the same power-of-two scale from the measured maximum
        if p0 is not None:
            assert rel_err(p1.float().sum(0), p0.float().sum(0)) < 1e-5
        assert rel_err(nchw(z1), ref) < 1e-3
    finally:
        RF._EVAL_SPLITK = old_sk
        RF.set_conv_math(old)


UP4_CASES = [
    (2, 32, 32, 128, 128),      # low resolution 16 x 16: one 16 x 16 patch per image, 128-wide column tiles (one per phase)
    (2, 16, 64, 64, 64),        # low resolution 8 x 32: 8 x 32 patches, 64-wide tiles, two channel chunks
    (4, 32, 64, 256, 128),      # 16 x 32: two patches per image (top / bottom borders inside the image), 8 chunks; 32 blocks in the
                                # input gradient: every start chunk / start phase of the rotated K loop, with wrap-around
    (1, 64, 64, 128, 256),      # 32 x 32: four patches, two column tiles per phase
    (3, 32, 32, 128, 192),      # 64-wide column tiles (192 = 3 x 64 per phase), 24 K chunks in the input gradient; odd image count
    (8, 32, 32, 512, 256),      # the input gradient in 128-wide tiles (256 blocks), 16 chunks per phase
]


@pytest.mark.parametrize("N,H,W,cin,cout", [UP4_CASES[0], UP4_CASES[2], UP4_CASES[5]])
def test_upconv_collapsed_kernels_one_plane(RF, N, H, W, cin, cout):
    """the same forward / input-gradient kernels on ONE fp16 plane (the f16 arithmetic of BASELINE configs[4]; 64-channel K-steps):
    against the nine-product one-plane kernels to 1e-3 of the tensor's maximum (both round their operands to fp16; the collapsed
    weights are rounded AFTER the taps are added) and against float64 to 1e-2 (fp16 operands: 11 significand bits)"""
    from rpnet_amd.hip import call, ptr, query
    g = torch.Generator().manual_seed(78)
    conv = torch.nn.Conv2d(cin, cout, 3, padding=1)
    with torch.no_grad():
        conv.weight.copy_((torch.rand(conv.weight.shape, generator=g) - 0.5) * (2.0 / (cin * 9) ** 0.5))
        conv.bias.copy_((torch.rand(cout, generator=g) - 0.5) * 0.2)
    a, go = rnd(292, N, cin, H // 2, W // 2), rnd(294, N, cout, H, W)
    cr = copy.deepcopy(conv).double()
    ar = a.double().requires_grad_(True)
    yr = cr(F.interpolate(ar, scale_factor=2, mode="nearest"))
    yr.backward(go.double())
    w, b = conv.weight.detach().to(DEV), conv.bias.detach().to(DEV)
    sx = torch.tensor([2.0 ** -13], device=DEV)
    xs = RF.split_f16(nhwc(a).to(DEV), sx, want_scale=False, planes=1)[0]
    dys = RF.split_f16(nhwc(go).to(DEV), sx, want_scale=False, planes=1)[0]
    pw = RF.PackedWeight(w)
    wp9, wd9, t9, u9 = pw.split_packs(1)
    wp4, wd4, t4, u4 = pw.up4_packs(1)
    out = {}
    for up4 in (True, False):
        y = torch.empty(N, H, W, cout, device=DEV)
        d = RF._desc(xs, None, wp4 if up4 else wp9, b, None, 0, y, None, N, H, W, 9, 1)
        d.split_planes = 1
        d.acc_scale_col, d.acc_scale_x = ptr(t4 if up4 else t9), ptr(sx)
        if up4:
            assert query("rpnet_conv_up4_supported", C.byref(d), 1)
            call("rpnet_conv_up4", C.byref(d), 1)
        else:
            call("rpnet_conv_fwd", C.byref(d))
        if up4:
            dx = torch.empty(N, H // 2, W // 2, cin, device=DEV)
            dd = RF._desc(dys, None, wd4, None, None, 0, dx, None, N, H, W, 9, 1)
            dd.split_planes = 1
            dd.acc_scale_col, dd.acc_scale_x = ptr(u4), ptr(sx)
            assert query("rpnet_conv_up4_supported", C.byref(dd), 2)
            call("rpnet_conv_up4", C.byref(dd), 2)
        else:
            gh = torch.empty(N, H, W, cin, device=DEV)
            dd = RF._desc(dys, None, wd9, None, None, 0, gh, None, N, H, W, 9, 0)
            dd.split_planes = 1
            dd.acc_scale_col, dd.acc_scale_x = ptr(u9), ptr(sx)
            call("rpnet_conv_fwd", C.byref(dd))
            dx = torch.empty(N, H // 2, W // 2, cin, device=DEV)
            call("rpnet_upsample2_bwd", ptr(gh), ptr(dx), N, H, W, cin)
        out[up4] = (y, dx)
    assert rel_err(out[True][0], out[False][0]) < 1e-3 and rel_err(out[True][1], out[False][1]) < 1e-3
    assert rel_err(nchw(out[True][0]), yr) < 1e-2 and rel_err(nchw(out[True][1]), ar.grad) < 1e-2
    # weight gradient on one plane (64-pixel K-steps, the halves in the two plane slots)
    dws = {}
    for up4 in (True, False):
        dw = torch.full((cout, cin, 3, 3), float("nan"), device=DEV)
        d = RF._desc(xs, None, None, None, None, 0, None, None, N, H, W, 9, 1, co_split=(cout, 0), wgrad=True)
        d.split_planes = 1
        d.acc_scale_x, d.acc_scale_dy = ptr(sx), ptr(sx)
        if up4:
            assert query("rpnet_conv_wgrad_up4_supported", C.byref(d))
            wb = query("rpnet_conv_wgrad_up4_workspace_bytes", N, H, W, cin, cout)
            ws = torch.empty(wb // 4 + 4, device=DEV)
            call("rpnet_conv_wgrad_up4", C.byref(d), ptr(dys), ptr(dw), ptr(ws), wb)
        else:
            wb = query("rpnet_conv_wgrad_workspace_bytes", N, H, W, cin, cout, 9)
            ws = torch.empty(wb // 4 + 4, device=DEV)
            call("rpnet_conv_wgrad", C.byref(d), ptr(dys), ptr(dw), cin, 0, cin, cin, ptr(ws), wb)
        dws[up4] = dw
    assert rel_err(dws[True], dws[False]) < 1e-4 and rel_err(dws[True], cr.weight.grad) < 1e-2


@pytest.mark.parametrize("N,H,W,cin,cout", UP4_CASES)
def test_upconv_collapsed_kernels_against_nine_tap_form_and_fp64(RF, N, H, W, cin, cout):
    """rpnet_conv_up4 (csrc/conv_up4_dma.hip) through the C ABI, without a BatchNorm behind it (no ReLU decisions that a 1e-7
    difference could flip): nn.Upsample(2) -> Conv2d 3x3 (net/modules.py:66-67) with the nine taps collapsed onto the 2 x 2 source
    pixels an output phase reads.  Forward (+ bias, + fused BatchNorm statistics) and input gradient (at the source resolution)
    against the nine-product kernels (rpnet_conv_fwd with the up-sampling in its gather, rpnet_upsample2_bwd) to 1e-5 of the tensor's
    maximum — the two differ by the rounding of the weight sums — and against torch in float64 to the 1e-3 bar (measured: 1e-6)."""
    from rpnet_amd.hip import call, ptr, query
    g = torch.Generator().manual_seed(77)
    conv = torch.nn.Conv2d(cin, cout, 3, padding=1)
    with torch.no_grad():
        conv.weight.copy_((torch.rand(conv.weight.shape, generator=g) - 0.5) * (2.0 / (cin * 9) ** 0.5))
        conv.bias.copy_((torch.rand(cout, generator=g) - 0.5) * 0.2)
    a, go = rnd(192, N, cin, H // 2, W // 2), rnd(194, N, cout, H, W)
    # fp64
    cr = copy.deepcopy(conv).double()
    ar = a.double().requires_grad_(True)
    yr = cr(F.interpolate(ar, scale_factor=2, mode="nearest"))
    yr.backward(go.double())
    w, b = conv.weight.detach().to(DEV), conv.bias.detach().to(DEV)
    x, dy = nhwc(a).to(DEV), nhwc(go).to(DEV)
    sx = torch.tensor([2.0 ** -13], device=DEV)
    sdy = torch.tensor([2.0 ** -13], device=DEV)
    xs = RF.split_f16(x, sx, want_scale=False, planes=2)[0]
    dys = RF.split_f16(dy, sdy, want_scale=False, planes=2)[0]
    pw = RF.PackedWeight(w)
    wp9, wd9, t9, u9 = pw.split_packs(2)
    wp4, wd4, t4, u4 = pw.up4_packs(2)
    groups = 2 if N % 2 == 0 else 1

    def fwd(up4):
        y = torch.empty(N, H, W, cout, device=DEV)
        d = RF._desc(xs, None, wp4 if up4 else wp9, b, None, 0, y, None, N, H, W, 9, 1, groups)
        d.split_planes = 2
        d.acc_scale_col, d.acc_scale_x = ptr(t4 if up4 else t9), ptr(sx)
        rows = query("rpnet_conv_up4_stats_blocks" if up4 else "rpnet_conv_stats_blocks", C.byref(d))
        part = torch.zeros(groups * max(rows, 1) * cout * 2, device=DEV, dtype=torch.float64)
        if rows:
            d.stats_partial = ptr(part)
        if up4:
            assert query("rpnet_conv_up4_supported", C.byref(d), 1)
            call("rpnet_conv_up4", C.byref(d), 1)
        else:
            call("rpnet_conv_fwd", C.byref(d))
        return y, part.reshape(groups, max(rows, 1), cout, 2).sum(1), rows

    def dgrad(up4):
        if up4:
            dx = torch.empty(N, H // 2, W // 2, cin, device=DEV)
            d = RF._desc(dys, None, wd4, None, None, 0, dx, None, N, H, W, 9, 1)
            d.split_planes = 2
            d.acc_scale_col, d.acc_scale_x = ptr(u4), ptr(sdy)
            assert query("rpnet_conv_up4_supported", C.byref(d), 2)
            call("rpnet_conv_up4", C.byref(d), 2)
            return dx
        gh = torch.empty(N, H, W, cin, device=DEV)
        d = RF._desc(dys, None, wd9, None, None, 0, gh, None, N, H, W, 9, 0)
        d.split_planes = 2
        d.acc_scale_col, d.acc_scale_x = ptr(u9), ptr(sdy)
        call("rpnet_conv_fwd", C.byref(d))
        dx = torch.empty(N, H // 2, W // 2, cin, device=DEV)
        call("rpnet_upsample2_bwd", ptr(gh), ptr(dx), N, H, W, cin)
        return dx

    y4, st4, rows4 = fwd(True)
    y9, st9, rows9 = fwd(False)
    assert rows4 == (N // groups) * H * W // 256
    assert rel_err(y4, y9) < 1e-5 and rel_err(nchw(y4), yr) < 1e-3
    # the fused statistics: per group and channel (sum, sum of squares) of the launch's own output
    per = N // groups
    yg = y4.double().reshape(groups, per * H * W, cout)
    assert rel_err(st4[..., 0], yg.sum(1)) < 1e-9 and rel_err(st4[..., 1], (yg * yg).sum(1)) < 1e-9
    dx4, dx9 = dgrad(True), dgrad(False)
    assert rel_err(dx4, dx9) < 1e-5 and rel_err(nchw(dx4), ar.grad) < 1e-3
    e4, e9 = rel_err(nchw(y4), yr), rel_err(nchw(y9), yr)
    assert e4 < 2 * e9 + 1e-6, (e4, e9)

    # weight gradient (csrc/conv_wgrad_up4.hip): sixteen tap products per source pixel, the phases summed back onto the nine taps
    def wgrad(up4, two_phase=False):
        dw = torch.full((cout, cin, 3, 3), float("nan"), device=DEV)
        d = RF._desc(xs, None, None, None, None, 0, None, None, N, H, W, 9, 1, co_split=(cout, 0), wgrad=True)
        d.split_planes = 2
        d.acc_scale_x, d.acc_scale_dy = ptr(sx), ptr(sdy)
        if up4:
            assert query("rpnet_conv_wgrad_up4_supported", C.byref(d))
            wb = query("rpnet_conv_wgrad_up4_workspace_bytes", N, H, W, cin, cout)
            ws = torch.empty(wb // 4 + 4, device=DEV)
            if two_phase:
                call("rpnet_conv_wgrad_up4", C.byref(d), ptr(dys), None, ptr(ws), wb)
                call("rpnet_conv_wgrad_up4", C.byref(d), None, ptr(dw), ptr(ws), wb)
            else:
                call("rpnet_conv_wgrad_up4", C.byref(d), ptr(dys), ptr(dw), ptr(ws), wb)
        else:
            wb = query("rpnet_conv_wgrad_workspace_bytes", N, H, W, cin, cout, 9)
            ws = torch.empty(wb // 4 + 4, device=DEV)
            call("rpnet_conv_wgrad", C.byref(d), ptr(dys), ptr(dw), cin, 0, cin, cin, ptr(ws), wb)
        return dw

    dw4, dw9 = wgrad(True), wgrad(False)
    assert rel_err(dw4, dw9) < 1e-5 and rel_err(dw4, cr.weight.grad) < 1e-3
    assert torch.equal(wgrad(True, two_phase=True), dw4)


@pytest.mark.parametrize("N,H,W,cin,cout", UP4_CASES[:3])
def test_upconv_layer_on_the_collapsed_form(RF, monkeypatch, N, H, W, cin, cout):
    """the same through the layer (conv + BatchNorm + ReLU, forward and backward, RF.conv_bn_relu_op(upsample=True)): both launches
    counted as collapsed; output, running statistics and weight gradient against the nine-product form and float64; the input
    gradient in relative L2 (one pre-activation within 1e-7 of zero may take the other side of its ReLU between two roundings of
    the same layer, which moves a 3 x 3 neighbourhood of the gradient by 1e-2 of its maximum: DESIGN.md section 4)."""
    from tests.helpers import rel_l2
    old = RF.conv_math()
    RF.set_conv_math("f16x2")
    try:
        groups = 2 if N % 2 == 0 else 1
        layer = _mk_layer(cin, cout, 3, 191)
        a = rnd(192, N, cin, H // 2, W // 2)
        go = rnd(194, N, cout, H, W)
        c_ref, b_ref = copy.deepcopy(layer[0]).double(), copy.deepcopy(layer[1]).double().train()
        ar = a.double().requires_grad_(True)
        xin = F.interpolate(ar, scale_factor=2, mode="nearest")
        per = N // groups
        ref = torch.cat([F.relu(b_ref(c_ref(xin[g * per:(g + 1) * per]))) for g in range(groups)], 0)
        ref.backward(go.double())
        RF.reset_arith()
        monkeypatch.setattr(RF, "_UP4", True)
        z4, da4, _, rv4, dw4, _ = _run(RF, monkeypatch, -1, layer, a, None, go, True, groups)
        counts = RF.arith_counts()
        assert counts.get("conv3x3_up4", {}).get("f16x2", 0) == 2, counts       # forward + input gradient on the collapsed form
        monkeypatch.setattr(RF, "_UP4", False)
        z9, da9, _, rv9, dw9, _ = _run(RF, monkeypatch, -1, layer, a, None, go, True, groups)
        assert da4.shape == da9.shape == (N, H // 2, W // 2, cin)
        assert rel_err(z4, z9) < 1e-5 and rel_err(rv4, rv9) < 1e-6
        assert rel_err(nchw(z4), ref) < 1e-3 and rel_err(rv4, b_ref.running_var) < 1e-5
        assert rel_l2(da4, da9) < 2e-3 and rel_l2(nchw(da4), ar.grad) < 2e-3
        assert rel_l2(dw4, dw9) < 2e-3 and rel_l2(dw4, c_ref.weight.grad) < 2e-3
    finally:
        RF.set_conv_math(old)
