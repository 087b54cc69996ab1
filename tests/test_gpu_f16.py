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
        z = RF.conv_bn_relu(x, conv, bn, RF.WeightCache(), True)
        planes, s = z._rp_split16
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
        z2 = RF.conv_bn_relu(RF.conv_bn_relu(x2, conv, bn, RF.WeightCache(), True), conv, bn, RF.WeightCache(), True)
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
