"""The pieces of the f16x2 arithmetic one by one, through the C ABI: fp16 planes of operand / power-of-two
scale — reconstruction accuracy, the scales' rigor under adversarial data (an outlier 1e6 standard deviations out must not
overflow fp16), the weight row scales, the one-pass gradient fan-in."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def RF():
    from rpnet_amd import functional
    return functional


def _planes_value(planes, scale):
    """[2, ...] fp16 container -> fp64 value of (h + l) * s"""
    return planes.view(torch.float16).double().sum(0) * scale.double()


def test_split_f16_reconstruction_and_joint_scale(RF):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(3, 8, 8, 64, generator=g).to(DEV) * 3.0
    mask = torch.rand(3, 8, 8, generator=g).to(DEV)
    s_a, s_b = torch.tensor([2.0 ** -12], device=DEV), torch.tensor([2.0 ** -10], device=DEV)
    for m, mode in ((None, 0), (mask, 1), (mask, 2)):
        planes, s = RF.split_f16(x, s_a, s_b, m, mode)
        assert s.item() == 2.0 ** -10                                  # max of the two producer scales, exact
        want = x.double() if mode == 0 else (x * (mask if mode == 1 else 1 - mask)[..., None]).double()
        got = _planes_value(planes, s)
        # 22 significand bits for values whose residual stays normal in fp16; absolute floor 2^-25 * s below
        assert ((got - want).abs() <= want.abs() * 2.0 ** -21 + 2.0 ** -24 * s.item()).all()
    planes, s = RF.split_f16(x, s_a)                                   # single source: its own scale
    assert s.item() == 2.0 ** -12 and torch.isfinite(planes.view(torch.float16).float()).all()


def test_bn_relu_scale_is_rigorous_under_outliers(RF):
    """|z| <= |gamma| sqrt(n) + |beta| for ANY batch: one activation a million standard deviations out still fits fp16"""
    N, H, W, Cc = 2, 16, 16, 64
    g = torch.Generator().manual_seed(2)
    conv = torch.nn.Conv2d(Cc, Cc, 3, padding=1).to(DEV)
    bn = torch.nn.BatchNorm2d(Cc).to(DEV).train()
    with torch.no_grad():
        bn.weight.copy_(0.5 + torch.rand(Cc, generator=g).to(DEV))
        bn.bias.copy_(torch.randn(Cc, generator=g).to(DEV))
    x = torch.randn(N, H, W, Cc, generator=g).to(DEV)
    x[0, 3, 3, :] = 1e6                                                # an outlier pixel in every channel
    RF.set_conv_math("f16x2")
    try:
        zo = RF.conv_bn_relu_op(x, conv, bn, RF.WeightCache(), True)
        z, planes, s = zo.x, zo.p16, zo.scale
        n = N * H * W
        bound = (bn.weight.abs() * n ** 0.5 + bn.bias.abs()).max().item()
        assert s.item() * 2.0 ** 15 >= bound and s.item() * 2.0 ** 15 < 2.0001 * bound      # power of two just above the bound
        h = planes.view(torch.float16).float()
        assert torch.isfinite(h).all() and h.abs().max() <= 2.0 ** 15
        assert z.max().item() <= bound
        got = _planes_value(planes, s)
        assert ((got - z.double()).abs() <= z.double().abs() * 2.0 ** -21 + 2.0 ** -24 * s.item()).all()
        # the gradient side: a huge upstream gradient element must not overflow either
        go = torch.randn(N, H, W, Cc, generator=g).to(DEV) * 1e-3
        go[1, 5, 5, :] = 1e5
        x2 = x.clone().requires_grad_(True)
        z2 = RF.conv_bn_relu(RF.conv_bn_relu_op(x2, conv, bn, RF.WeightCache(), True), conv, bn, RF.WeightCache(), True)
        z2.backward(go)
        assert torch.isfinite(x2.grad).all() and torch.isfinite(conv.weight.grad).all()
    finally:
        RF.set_conv_math("f32")


def test_weight_pack_row_scales(RF):
    g = torch.Generator().manual_seed(3)
    w = (torch.randn(64, 64, 3, 3, generator=g) * torch.logspace(-4, 1, 64)[:, None, None, None]).to(DEV)   # rows over 5 decades
    pw = RF.PackedWeight(w)
    wps, wds, t, u = pw.split_packs(2)
    rowmax = w.abs().amax(dim=(1, 2, 3))
    colmax = w.abs().amax(dim=(0, 2, 3))
    assert ((t * 2.0 ** 15 >= rowmax) & (t * 2.0 ** 15 < 2.0001 * rowmax)).all()
    assert ((u * 2.0 ** 15 >= colmax) & (u * 2.0 ** 15 < 2.0001 * colmax)).all()
    h = wps.view(torch.float16).float()
    assert torch.isfinite(h).all() and h.abs().max() <= 2.0 ** 15
    # wp layout [plane][tap][cin/32][cout][32]: reconstruct and compare with w / t
    val = wps.view(torch.float16).double().sum(0).reshape(9, 2, 64, 32)           # [tap][cin block][cout][cin in block]
    rec = val.permute(2, 1, 3, 0).reshape(64, 64, 9) * t.double()[:, None, None]   # [cout][cin][tap]
    want = w.double().reshape(64, 64, 9)
    assert ((rec - want).abs() <= want.abs() * 2.0 ** -21 + 2.0 ** -24 * t.double()[:, None, None]).all()


@pytest.mark.parametrize("n,numel", [(1, 1000), (2, 4099), (10, 8 * 64 * 64 * 4), (16, 333), (21, 5000)])
def test_sum_n(RF, n, numel):
    g = torch.Generator().manual_seed(n)
    ts = [torch.randn(numel, generator=g).to(DEV) for _ in range(n)]
    got = RF.sum_n(ts)
    want = torch.stack([t.double() for t in ts]).sum(0)
    assert (got.double() - want).abs().max() <= 1e-5 * max(1.0, want.abs().max().item())


def test_fan_in_functions_match_autograd(RF):
    x = torch.randn(6, 4, 4, 8, device=DEV, requires_grad=True)
    a, b = RF.SplitRows.apply(x, 2)
    us = RF.FanOut.apply(b, 3)
    loss = (a * 2).sum() + sum((k + 1) * (u ** 2).sum() for k, u in enumerate(us))
    loss.backward()
    xr = x.detach().clone().requires_grad_(True)
    (xr[:2] * 2).sum().add(sum((k + 1) * (xr[2:] ** 2).sum() for k in range(3))).backward()
    assert torch.allclose(x.grad, xr.grad, rtol=1e-6, atol=1e-6)
    # unused outputs: their gradients are None
    y = torch.randn(4, 3, device=DEV, requires_grad=True)
    u0, u1 = RF.FanOut.apply(y, 2)
    u1.sum().backward()
    assert torch.equal(y.grad, torch.ones_like(y))


def test_f16_threshold_switch(RF):
    """small calls stay on three bf16 planes (RF.set_f16_active): same results either way, within the parity bar"""
    import yaml
    from rpnet_amd import modules as RM
    from rpnet_amd.modules import RP_Net
    from rpnet_amd.utils.seeding import seed_module_
    from tests.helpers import EXAMPLE_YAML, episode_tensors
    cfg = yaml.load(open(EXAMPLE_YAML), Loader=yaml.FullLoader)
    cfg["n_iter_refinement"] = 2
    (si, fg, bg, qi, ql, appr), _ = episode_tensors(91, 2, 64, DEV)
    net = RP_Net(cfg={"align": True, "backbone": "UNet"}, backbone_cfg=cfg).to(DEV)
    seed_module_(net)
    net.train()
    RF.set_conv_math("f16x2")
    old = RM._F16_MIN_PIXELS
    try:
        outs = []
        for thr, active in ((0, True), (1 << 30, False)):
            RM._F16_MIN_PIXELS = thr
            with torch.no_grad():
                outs.append(net(si, fg, bg, qi, appr_query_labels=appr)["output"])
            assert RF.f16_mode() == active
        assert (outs[0] - outs[1]).abs().max() <= 1e-4 * outs[1].abs().max()
    finally:
        RM._F16_MIN_PIXELS = old
        RF.set_conv_math("f32")


# ------------------------------------------------------------------------------------------------------------------
# "f16": ONE fp16 plane (plain fp16 operands, fp32 accumulation, fp32 BatchNorm statistics) — BASELINE configs[4].
# The reference is fp32-only (net/rp_net.py:160,166-171,179), so the oracle is the fp32 reference on identical inputs
# and the tolerance is stated here (SURVEY.md §8a "fp16 (c5)"; measured numbers in DESIGN.md §6):
#   * every iteration's logits within 1e-2 of max |logit| (= 20) GIVEN THE SAME INPUT MASK (teacher forcing, SURVEY.md
#     §7.3: the loop feeds a hard 0.5 threshold back, so a single flipped pixel changes the next iteration's input);
#     measured 2e-3 .. 4e-3;
#   * pixels whose thresholded prediction differs from the fp32 path's: <= 3e-4 of the pixels per iteration at 256^2 and
#     above (measured 1.0e-4 .. 1.7e-4 at 512^2), <= 2e-3 at 64^2 / 128^2 (5 .. 12 pixels: the mask boundary is a larger
#     share of a small image);
#   * Dice / foreground fraction per iteration within 1e-3 wherever one pixel is less than that (256^2 batch 8 and
#     512^2 batch 4: free-running through T = 5 at configs[1]'s shape, teacher-forced through T = 10 at configs[4]'s);
#   * loss within 1e-2 relative.
# Free-running at configs[4]'s size the RANDOM-WEIGHT model's loop is not contractive (its Dice falls from 0.69 to 0.33
# over the ten iterations under every arithmetic) and amplifies the few flipped pixels: Dice deviates by up to 7e-3
# after iteration 3 (measured in round 2) — bounded by 2e-2 in the test and stated, not hidden.
F16_LOGIT_TOL = 1e-2
F16_DICE_TOL = 1e-3
F16_FLIP_TOL = 3e-4


@pytest.fixture
def f16_single(RF):
    from rpnet_amd import modules as RM
    old, old_min = RF.conv_math(), RM._F16_MIN_PIXELS
    RF.set_conv_math("f16")
    RM._F16_MIN_PIXELS = 0
    yield RF
    RM._F16_MIN_PIXELS = old_min
    RF.set_conv_math(old)


def _pred(logits):
    return logits.softmax(1)[:, 1] > 0.5


def _dice(logits, ql):
    pred = _pred(logits).long()
    return float(2.0 * (pred * ql).sum() / (pred.sum() + ql.sum() + 1e-7)), float(pred.float().mean())


def _teacher_masks(ref_refinement, T, scale=4):
    """the masks the REFERENCE run fed into iterations 1 .. T-1 (net/rp_net.py:308-311)"""
    return {i: torch.nn.functional.avg_pool2d(_pred(ref_refinement[i - 1].detach()).float()[:, None], scale)[:, 0].to(DEV)
            for i in range(1, T)}


def _logit_err(got, want):
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    return float((got - want).abs().max() / want.abs().max())


def test_f16_single_plane_layer_vs_fp32(f16_single):
    """conv3x3 + BatchNorm + ReLU (x2, the second on a concatenation) under the one-plane arithmetic against torch fp32:
    output within a few fp16 ulps of the tensor maximum, input and weight gradients to 5e-3 relative L2; every launch
    counted as f16.  BatchNorm beta = 3 keeps the pre-activations away from the ReLU switch: with fp16 operands a
    fraction ~1e-3 of standard-normal pre-activations changes sign, and ONE switched term moves a gradient element by
    1 / sqrt(9 C) = 3 % of its typical size — the conditioning of the function, not an error of the kernels (the
    fp32-equivalent arithmetics pass the same comparison at 1e-3 without the offset: test_conv_bn_relu)."""
    import copy
    import torch.nn as nn
    RF = f16_single
    from tests.helpers import rel_err, rnd
    torch.manual_seed(3)
    c1, b1 = nn.Conv2d(64, 128, 3, padding=1), nn.BatchNorm2d(128)
    c2, b2 = nn.Conv2d(192, 128, 3, padding=1), nn.BatchNorm2d(128)
    with torch.no_grad():
        b1.bias.fill_(3.0)
        b2.bias.fill_(3.0)
    x = rnd(5, 4, 64, 32, 32)
    go = rnd(6, 4, 128, 32, 32)
    ref_mods = [copy.deepcopy(m) for m in (c1, b1, c2, b2)]
    xr = x.clone().requires_grad_(True)
    a = torch.relu(ref_mods[1](ref_mods[0](xr)))
    zr = torch.relu(ref_mods[3](ref_mods[2](torch.cat([xr, a], 1))))
    zr.backward(go)
    for m in (c1, b1, c2, b2):
        m.to(DEV)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_(True)
    cache = RF.WeightCache()
    RF.reset_arith()
    # the network's first layer gets its input scale from a BatchNorm; a raw tensor carries none: hand one in
    s_in = torch.tensor([2.0 ** (int(np.ceil(np.log2(float(x.abs().max())))) - 15)], device=DEV)
    xo = RF.Operand(xd, scale=s_in)
    a = RF.conv_bn_relu_op(xo, c1, b1, cache, True, out_split="scale")
    z = RF.conv_bn_relu(xo, c2, b2, cache, True, x1=a)
    z.backward(go.permute(0, 2, 3, 1).contiguous().to(DEV))
    counts = RF.arith_counts()
    assert counts["conv3x3"] == {"f16": 4} and counts["wgrad3x3"] == {"f16": 2}, counts       # 2 forward + 2 dgrad, 2 wgrad
    def l2(a, b):
        a, b = a.double().cpu(), b.double().cpu()
        return float((a - b).norm() / b.norm())

    assert rel_err(z.permute(0, 3, 1, 2), zr) < 4e-3
    assert l2(xd.grad.permute(0, 3, 1, 2), xr.grad) < 5e-3 and rel_err(xd.grad.permute(0, 3, 1, 2), xr.grad) < 2e-2
    assert l2(c1.weight.grad, ref_mods[0].weight.grad) < 5e-3 and l2(c2.weight.grad, ref_mods[2].weight.grad) < 5e-3
    assert l2(b2.weight.grad, ref_mods[3].weight.grad) < 5e-3


@pytest.mark.parametrize("size,B,T", [(64, 2, 3), (128, 1, 3)])
def test_f16_two_way_vs_fp32_oracle(f16_single, size, B, T):
    """2-way 1-shot (the configs[4] shape class) under the one-plane fp16 arithmetic against the fp32 oracle composed
    from the reference's own pieces: teacher-forced logits, threshold flips and loss at the stated tolerance."""
    from oracle import rpnet_oracle as O
    from tests.helpers import episode_tensors, load_cfg
    from tests.test_gpu_model import build, total_loss
    RF = f16_single
    cfg = load_cfg(T)
    (si, fg, bg, qi, ql, appr), _ = episode_tensors(66 + size, B, size, "cpu", n_shots=1, n_ways=2)
    P = O.seeded_params()
    with torch.no_grad():
        ref = O.rp_net_forward(P, cfg, si, fg, bg, qi, appr, True, align=True)
        ref_loss = O.total_loss(ref, ql, cfg["align_loss_scaler"])
    net = build(cfg, True)
    net.forced_masks = _teacher_masks(ref["refinement"], T)
    mv = lambda t: t.to(DEV)  # noqa: E731
    RF.reset_arith()
    out = net([[mv(s) for s in w] for w in si], [[mv(s) for s in w] for w in fg], [[mv(s) for s in w] for w in bg],
              [mv(qi[0])], appr_query_labels=mv(appr))
    loss = total_loss(out, mv(ql), cfg["align_loss_scaler"])
    loss.backward()
    counts = RF.arith_counts()
    assert set(counts["conv3x3"]) == {"f16"} and set(counts["wgrad3x3"]) == {"f16"} and set(counts["corr"]) == {"f16"}, counts
    assert out["output"].shape == (B, 3, size, size)
    for i in range(T):
        got, want = out["refinement"][i].detach().cpu(), ref["refinement"][i]
        assert _logit_err(got, want) <= F16_LOGIT_TOL, f"logits, iteration {i}: {_logit_err(got, want):.2e}"
        flips = float((_pred(got) != _pred(want)).float().mean())
        assert flips <= 2e-3, f"iteration {i}: {flips:.2e} of the pixels flipped"
    assert abs(loss.item() - ref_loss.item()) <= 1e-2 * abs(ref_loss.item())
    assert all(torch.isfinite(p.grad).all() for p in net.parameters() if p.grad is not None)


F16_YARD_EPS = (5e-4, 1.5e-3, 5e-3)     # relative input perturbations: 2^-11 (one fp16 rounding of the image) and two steps up


@pytest.mark.parametrize("size,B,T", [(64, 2, 3), (128, 1, 3)])
def test_f16_gradients_vs_perturbation_yardstick(f16_single, size, B, T):
    """Gradients of the one-plane fp16 step with the yardstick of tests/test_gpu_model.py::test_gradients_vs_fp64_yardstick
    at fp16 scale: the reference point is the oracle (2-way, composed from the reference's pieces) in float64 through torch's own
    device kernels (tests/helpers.py oracle_step(device): nothing of librpnet_hip.so; on the box's host cores these 13 oracle steps
    were the two slowest tests of the suite and scaled with whatever else the shared host was doing); the yardstick is
    how far the ORACLE's own gradients move (relative L2 per tensor, maximum over four draws) when every image pixel is
    perturbed by eps relative, with eps the smallest of F16_YARD_EPS (starting at 2^-11, one fp16 rounding) whose median
    effect on the oracle's logits is at least a third of the fp16 path's own forward deviation — i.e. a perturbation that
    disturbs the forward pass no more than fp16 operands do.  Requirement: every gradient tensor of the fp16 step lies
    within 3 yardsticks of the oracle's.  (The round-2 test only asked for cosine > 0.9.)"""
    from oracle import rpnet_oracle as O
    from tests.helpers import episode_tensors, load_cfg, oracle_step, rel_err
    from tests.test_gpu_model import build, total_loss
    RF = f16_single
    cfg = load_cfg(T)
    inputs, _ = episode_tensors(66 + size, B, size, "cpu", n_shots=1, n_ways=2)
    si, fg, bg, qi, ql, appr = inputs
    g0, l0, o0 = oracle_step(cfg, inputs, dtype=torch.float64, device=DEV)
    net = build(cfg, True)
    mv = lambda t: t.to(DEV)  # noqa: E731
    out = net([[mv(s) for s in w] for w in si], [[mv(s) for s in w] for w in fg], [[mv(s) for s in w] for w in bg],
              [mv(qi[0])], appr_query_labels=mv(appr))
    total_loss(out, mv(ql), cfg["align_loss_scaler"]).backward()
    assert set(RF.arith_counts()["conv3x3"]) == {"f16"}
    fwd16 = rel_err(out["refinement"][0].detach(), o0["refinement"][0].detach())
    assert fwd16 <= F16_LOGIT_TOL
    for eps in F16_YARD_EPS:
        yard, moves = {}, []
        for draw in range(4):
            gp, _, op = oracle_step(cfg, inputs, dtype=torch.float64, noise=(300 + draw, eps), device=DEV)
            moves.append(rel_err(op["refinement"][0].detach(), o0["refinement"][0].detach()))
            for n, v in gp.items():
                nrm = float(g0[n].norm())
                if nrm >= 1e-4:
                    yard[n] = max(yard.get(n, 0.0), float((v - g0[n]).norm()) / nrm)
        mid = sorted(moves)[len(moves) // 2]
        if mid >= fwd16 / 3.0:
            break
    assert mid <= 3.0 * fwd16 or eps == F16_YARD_EPS[0], (eps, mid, fwd16)     # not inflated either
    report = []
    for n, p in net.named_parameters():
        if p.grad is None or n not in yard:
            continue
        e = float((p.grad.double().cpu() - g0[n].double()).norm() / g0[n].double().norm())
        report.append((e / yard[n], n, e, yard[n]))
    report.sort(reverse=True)
    print(f"f16 gradients at {size}^2: eps {eps:g}, forward deviation {fwd16:.1e} (perturbed oracle {mid:.1e}); worst err / yardstick",
          [(round(r, 2), n, f"{a:.1e}", f"{b:.1e}") for r, n, a, b in report[:3]])
    for ratio, n, e, y in report:
        assert e <= 3.0 * y, f"{n}: fp16 step {e:.2e} from the oracle's gradient, yardstick (eps = {eps:g}) {y:.2e}"


def test_f16x2_two_way_512_vs_fp32_oracle(RF):
    """configs[4]'s image size and class count under the fp32-EQUIVALENT arithmetic against the CPU oracle itself (2-way 1-shot,
    512x512, batch 1, T = 2; oracle in algorithmic mode: the all-pairs correlation of the as-written mode is a 1 GB tensor per
    call here): logits of both iterations (the second teacher-forced with the oracle's mask), Dice, loss at the fp32 bar.
    test_config5_full_size_f16 uses the f16x2 step as the yardstick of the fp16 tolerance at full size: this pins that
    yardstick to the reference's arithmetic at the same size."""
    from oracle import rpnet_oracle as O
    from rpnet_amd import modules as RM
    from tests.helpers import episode_tensors, load_cfg, rel_err
    from tests.test_gpu_model import build, total_loss
    old, old_min = RF.conv_math(), RM._F16_MIN_PIXELS
    RF.set_conv_math("f16x2")
    RM._F16_MIN_PIXELS = 0
    try:
        T, B, size = 2, 1, 512
        cfg = load_cfg(T)
        (si, fg, bg, qi, ql, appr), _ = episode_tensors(577, B, size, "cpu", n_shots=1, n_ways=2)
        P = O.seeded_params()
        with torch.no_grad():
            ref = O.rp_net_forward(P, cfg, si, fg, bg, qi, appr, True, align=True, as_written=False)
            ref_loss = O.total_loss(ref, ql, cfg["align_loss_scaler"])
        net = build(cfg, True)
        net.forced_masks = _teacher_masks(ref["refinement"], T)
        mv = lambda t: t.to(DEV)  # noqa: E731
        RF.reset_arith()
        with torch.no_grad():
            out = net([[mv(s) for s in w] for w in si], [[mv(s) for s in w] for w in fg], [[mv(s) for s in w] for w in bg],
                      [mv(qi[0])], appr_query_labels=mv(appr))
            loss = total_loss(out, mv(ql), cfg["align_loss_scaler"])
        counts = RF.arith_counts()
        assert set(counts["conv3x3"]) == {"f16x2"} and set(counts["corr"]) == {"f16x2"}, counts
        assert out["output"].shape == (B, 3, size, size)
        for i in range(T):
            got, want = out["refinement"][i].cpu(), ref["refinement"][i]
            assert rel_err(got, want) < 1e-3, f"logits, iteration {i}: {rel_err(got, want):.2e}"
            assert float((_pred(got) != _pred(want)).float().mean()) <= 2e-5          # <= 5 of 262144 pixels on the threshold
            (d_g, f_g), (d_r, f_r) = _dice(got, ql), _dice(want, ql)
            assert abs(d_g - d_r) <= 1e-3 and abs(f_g - f_r) <= 1e-3
        assert abs(loss.item() - ref_loss.item()) <= 1e-3 * abs(ref_loss.item())
    finally:
        RM._F16_MIN_PIXELS = old_min
        RF.set_conv_math(old)


def test_f16_dice_at_config1_shape(f16_single):
    """configs[1]'s shape (1-way 1-shot, 256x256, T=5, batch 8), FREE-RUNNING, one fp16 plane against the fp32-equivalent
    arithmetic (f16x2, itself held to the reference's golden vectors at this size): Dice and foreground fraction of
    every iteration within 1e-3 (measured 3e-5)."""
    from tests.helpers import episode_tensors, load_cfg
    from tests.test_gpu_model import build
    RF = f16_single
    cfg = load_cfg(5)
    (si, fg, bg, qi, ql, appr), _ = episode_tensors(1234, 8, 256, DEV)
    outs = {}
    for math in ("f16x2", "f16"):
        RF.set_conv_math(math)
        net = build(cfg, True)
        with torch.no_grad():
            outs[math] = net(si, fg, bg, qi, appr_query_labels=appr)["refinement"]
    for i in range(5):
        (d_g, f_g), (d_r, f_r) = _dice(outs["f16"][i], ql), _dice(outs["f16x2"][i], ql)
        assert abs(d_g - d_r) <= F16_DICE_TOL and abs(f_g - f_r) <= F16_DICE_TOL, (i, d_g, d_r, f_g, f_r)
    assert _logit_err(outs["f16"][0], outs["f16x2"][0]) <= F16_LOGIT_TOL


def test_config5_full_size_f16(f16_single):
    """BASELINE configs[4] at its real size: 2-way 1-shot, 512x512, T=10, batch 4 per GPU, one fp16 plane.  The
    reference would allocate a 1.07 GB all-pairs correlation per sample and CRE call here (net/rp_net.py:158-161); the
    local-window kernel must not.  Size-independent properties + the same step under the fp32-equivalent arithmetic
    (f16x2, itself held to the reference at 64^2..256^2) as the full-size yardstick of the stated fp16 tolerance:
    teacher-forced logits / flips / Dice per iteration, free-running Dice within the cascade bound."""
    from tests.helpers import episode_tensors, load_cfg
    from tests.test_gpu_model import build, total_loss
    RF = f16_single
    T, B, size = 10, 4, 512
    cfg = load_cfg(T)
    (si, fg, bg, qi, ql, appr), _ = episode_tensors(555, B, size, DEV, n_shots=1, n_ways=2)

    def run(math, forced=None):
        RF.set_conv_math(math)
        torch.cuda.reset_peak_memory_stats()
        net = build(cfg, True)
        net.forced_masks = forced
        RF.reset_arith()
        out = net(si, fg, bg, qi, appr_query_labels=appr)
        loss = total_loss(out, ql, cfg["align_loss_scaler"])
        loss.backward()
        torch.cuda.synchronize()
        return net, out, loss, RF.arith_counts(), torch.cuda.max_memory_allocated()

    net, out, loss, counts, peak = run("f16")
    assert out["output"].shape == (B, 3, size, size) and len(out["refinement"]) == T
    assert torch.isfinite(loss) and all(torch.isfinite(p.grad).all() for p in net.parameters() if p.grad is not None)
    assert (out["output"].abs() <= 20.0 + 1e-3).all()
    assert set(counts["conv3x3"]) == {"f16"} and set(counts["wgrad3x3"]) == {"f16"}, counts
    assert set(counts["corr"]) == {"f16"} and set(counts["corr_bwd"]) == {"f16"}, counts
    sd = net.state_dict()
    assert int(sd["encoder.Conv1.conv.1.num_batches_tracked"]) == 2            # one support call (8 images) + the query call
    assert int(sd["cre.w_k.1.num_batches_tracked"]) == 2 + T                    # one CRE call per way + T query calls
    # 12 CRE calls x 4 samples x 1.07 GB of all-pairs scores (plus their grid_sample batches) would be > 51 GB
    assert peak < 40e9, f"peak allocation {peak / 1e9:.1f} GB"
    ref_net, ref, ref_loss, _, _ = run("f16x2")
    # free-running: the loop's own masks (cascade of the few flipped pixels through the random-weight model: see the header)
    for i in range(T):
        (d_g, f_g), (d_r, f_r) = _dice(out["refinement"][i], ql), _dice(ref["refinement"][i], ql)
        assert abs(d_g - d_r) <= (F16_DICE_TOL if i < 2 else 2e-2) and abs(f_g - f_r) <= (F16_DICE_TOL if i < 2 else 5e-3), (i, d_g, d_r)
    assert abs(loss.item() - ref_loss.item()) <= 2e-2 * abs(ref_loss.item())
    # teacher-forced: the fp32-equivalent run's masks into every iteration -> the stated tolerance, iteration by iteration
    _, tf, tf_loss, _, _ = run("f16", forced=_teacher_masks(ref["refinement"], T))
    for i in range(T):
        got, want = tf["refinement"][i], ref["refinement"][i]
        assert _logit_err(got, want) <= F16_LOGIT_TOL, f"logits, iteration {i}: {_logit_err(got, want):.2e}"
        assert float((_pred(got) != _pred(want)).float().mean()) <= F16_FLIP_TOL
        (d_g, f_g), (d_r, f_r) = _dice(got, ql), _dice(want, ql)
        assert abs(d_g - d_r) <= F16_DICE_TOL and abs(f_g - f_r) <= F16_DICE_TOL, (i, d_g, d_r, f_g, f_r)
    assert abs(tf_loss.item() - ref_loss.item()) <= 1e-2 * abs(ref_loss.item())
    # gradients against the fp32-equivalent step at FULL size: a sanity check of direction only (an oracle run of this size does
    # not fit a test); the bound with a yardstick is test_f16_gradients_vs_perturbation_yardstick at 64^2 / 128^2
    g16 = {n: p.grad for n, p in net.named_parameters() if p.grad is not None}
    for n, p in ref_net.named_parameters():
        if p.grad is not None and p.grad.norm() > 1e-6:
            a, b = g16[n].double().flatten(), p.grad.double().flatten()
            assert float(torch.dot(a, b) / (a.norm() * b.norm())) > 0.9, n


def test_config5_trained_weights_free_running_dice(f16_single):
    """north_star: "<= 1e-3 Dice deviation from reference" for the fp16 configuration.  On RANDOM weights the refinement loop is not
    contractive at configs[4]'s size and amplifies the ~1e-4 of thresholded pixels the one-plane arithmetic moves
    (test_config5_full_size_f16: free-running bound 2e-2 there).  The bar is about a model that has been trained: 200 Adam steps of
    train_rpnet.train on synthetic 2-way 512^2 episodes (T = 10, batch 4, fp32-equivalent f16x2 arithmetic), then configs[4]'s call
    (2-way 1-shot, 512^2, T = 10, batch 4) FREE-RUNNING — every iteration on the loop's own thresholded mask — under f16 and under
    f16x2, in train mode (the benched step's BatchNorm) and eval mode (the reference driver's): per-iteration Dice and
    foreground fraction within 1e-3 (measured 2e-5 / 5e-5; tools/trained_f16_dice.py, profiles/r04_trained_f16_dice.json)."""
    import tools.trained_f16_dice as TD
    dev = torch.device(DEV)
    net, hist = TD.train_weights(200, 512, 1e-4, dev, n_ways=2, iters=10)
    assert hist[-1] < 0.6 * hist[0], (hist[0], hist[-1])          # it did train
    for mode in (True, False):
        res = TD.free_running(net, TD.CASES["configs4_2way_512_T10_B4"], dev, mode)
        dd, df = TD.deviations(res)
        assert max(dd) <= F16_DICE_TOL and max(df) <= F16_DICE_TOL, (mode, dd, df)
        assert min(a[0] for a in res["f16x2"]) > 0.2               # a segmentation, not an empty prediction


