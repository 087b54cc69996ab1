"""GPU parity, model level: rpnet_amd.RP_Net (HIP path) against the golden vectors the
reference produced (tests/golden/*.npz) and against the CPU oracle on the same seeded
inputs; full-size runs are checked through size-independent properties."""
import os

import numpy as np
import pytest
import torch

from tests.helpers import episode_tensors, in_checksum, load_cfg, rel_err, rel_l2

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-3   # BASELINE.json north_star: outputs within 1e-3 relative fp32


def build(cfg, training):
    from rpnet_amd.modules import RP_Net
    from rpnet_amd.utils.seeding import seed_module_
    net = RP_Net(cfg={"align": True, "backbone": "UNet"}, backbone_cfg=cfg).to(DEV)
    seed_module_(net)
    net.train(training)
    return net


def total_loss(out, ql, scaler):
    from rpnet_amd.functional import dice_ce
    loss = dice_ce(out["output"], ql)
    for v in out["refinement"].values():
        loss = loss + dice_ce(v, ql)
    return loss + scaler * out["align_loss"]


def nchw(x):
    return x.permute(0, 3, 1, 2)


def test_unet_and_cre_vs_oracle():
    """encoder d4 (train: 2 statistic groups == 2 reference calls; eval) and one CRE call."""
    from oracle import rpnet_oracle as O
    from rpnet_amd import functional as RF
    cfg = load_cfg(1)
    (si, fg, bg, qi, ql, appr), _ = episode_tensors(77, 2, 32)
    for training in (True, False):
        net = build(cfg, training)
        P = O.seeded_params()
        with torch.no_grad():
            ref_s = O.unet_d4(P, si[0][0], training)
            ref_q = O.unet_d4(P, qi[0], training)
            x = torch.cat([si[0][0], qi[0]], 0).to(DEV)
            d4 = net.encoder.forward_nhwc(x.reshape(4, 32, 32, 1), RF.WeightCache(), groups=2 if training else 1).x
            assert rel_err(nchw(d4[:2]), ref_s) < TOL and rel_err(nchw(d4[2:]), ref_q) < TOL
            m = RF.mask_avgpool(fg[0][0].to(DEV), 4)
            got = net.cre.forward_masked(d4[:2].contiguous(), m, RF.WeightCache())
            sm = torch.nn.functional.avg_pool2d(fg[0][0][:, None], 4)
            ref = O.cre(P, ref_s * sm, ref_s * (1 - sm), training, 5, False)
            assert rel_err(nchw(got), ref) < TOL
        if training:
            sd = net.state_dict()
            for k in ("encoder.Conv1.conv.1.running_mean", "encoder.Conv5.conv.4.running_var", "encoder.Up_conv4.conv.4.running_var",
                      "cre.q.1.running_mean"):
                assert rel_err(sd[k], P[k]) < 1e-4, k
            assert int(sd["encoder.Conv3.conv.1.num_batches_tracked"]) == 2 and int(sd["cre.w_k.1.num_batches_tracked"]) == 1


@pytest.fixture(params=["bf16x3", "f16x2", "f32"])
def conv_math(request):
    """default arithmetic of the 3x3 convolutions (3-plane split-bf16, fp32-equivalent) and the fp32-MFMA kernels"""
    from rpnet_amd import functional as RF
    from rpnet_amd import modules as RM
    old, old_min = RF.conv_math(), RM._F16_MIN_PIXELS
    RF.set_conv_math(request.param)
    RM._F16_MIN_PIXELS = 0          # the fp16 planes at every size (the default keeps small calls on bf16x3)
    yield request.param
    RM._F16_MIN_PIXELS = old_min
    RF.set_conv_math(old)


def _meta(g):
    """(size, B, T, training, seed, n_ways, n_shots) of a model fixture (the 1-way 1-shot fixtures store the first five)"""
    m = [int(v) for v in g["meta"]]
    return tuple(m) if len(m) == 7 else tuple(m) + (1, 1)


YARD_EPS = 4e-7      # relative input perturbation of the yardstick: moves the fp64 forward as much as the HIP path deviates
YARD_FLOOR = 5e-5    # relative L2: fp32 round-off of reductions over ~1e5 terms, where the yardstick itself is at round-off level
_YARD_CACHE = {}     # tag -> fp64 oracle results (shared by the three arithmetics and by the two tests that use them)


def _yardstick(tag, g):
    """(fp64 gradients, fp64 loss, fp64 logits of iteration 0, {parameter: yardstick}, forward movements of the draws) of
    fixture `tag`: the CPU oracle in FLOAT64 on the fixture's inputs, and how far ITS gradients move (relative L2, maximum
    over the draws) when every image pixel is perturbed by YARD_EPS relative.  Six draws at 64^2 / 128^2, two at 256^2
    (each is one fp64 forward + backward on the host)."""
    from tests.helpers import oracle_step
    if tag not in _YARD_CACHE:
        size, B, T, _, seed, n_ways, n_shots = _meta(g)
        cfg = load_cfg(T)
        mfm = tag.rsplit("_", 1)[1] if tag.startswith("m64_train_") else None
        if mfm:
            cfg["mask_feature_map"] = mfm
        cpu_inputs, _ = episode_tensors(seed, B, size, n_shots=n_shots, n_ways=n_ways)
        # (the float64 oracle through torch's own device kernels: tests/helpers.py oracle_step(device); on the host cores these
        # runs were 620 of the suite's 848 s in round 4)
        g64, l64, o64 = oracle_step(cfg, cpu_inputs, dtype=torch.float64, device=DEV)
        yard, fwd_moves = {}, []
        for draw in range(6 if size < 256 else 2):
            gp, _, op = oracle_step(cfg, cpu_inputs, dtype=torch.float64, noise=(100 + draw, YARD_EPS), device=DEV)
            fwd_moves.append(rel_err(op["refinement"][0].detach(), o64["refinement"][0].detach()))
            for n, v in gp.items():
                nrm = float(g64[n].norm())
                if nrm >= 1e-4:
                    yard[n] = max(yard.get(n, 0.0), float((v - g64[n]).norm()) / nrm)
        _YARD_CACHE[tag] = (g64, l64, o64["refinement"][0].detach(), yard, fwd_moves)
    return _YARD_CACHE[tag]


@pytest.mark.parametrize("tag", ["m64_train", "m64_eval", "m128_train", "m256_train", "m64_train_x", "m64_train_x2", "m64_train_x3"])
def test_model_vs_golden(golden, tag, conv_math):
    g = golden(tag)
    size, B, T, training, seed = (int(v) for v in g["meta"])
    training = bool(training)
    cfg = load_cfg(T)
    mfm = tag.rsplit("_", 1)[1] if tag.count("_") == 2 else None
    if mfm:                      # mask_feature_map 'x' / 'x2' / 'x3' (net/unet.py:401-414,437-449): the reference's own outputs
        cfg["mask_feature_map"] = mfm
    (si, fg, bg, qi, ql, appr), ep = episode_tensors(seed, B, size, DEV)
    assert np.allclose(in_checksum(ep), g["in_checksum"], rtol=0, atol=1e-6), "synthetic inputs drifted"
    s_out, s_d4, s_f = (int(v) for v in g["strides"])
    net = build(cfg, training)
    net.taps = {}
    from rpnet_amd import functional as RF
    RF.reset_arith()
    with torch.set_grad_enabled(training):
        out = net(si, fg, bg, qi, appr_query_labels=appr)
        loss = total_loss(out, ql, cfg["align_loss_scaler"])
    assert out["output"] is out["refinement"][T - 1]
    # no silent change of arithmetic: every 3x3 convolution and every correlation of this forward (train AND eval mode)
    # ran what was asked for
    counts = RF.arith_counts()
    allowed = {conv_math} | ({"bf16x3"} if (mfm == "x" and conv_math == "f16x2") else set())   # 'x': Conv1 reads the raw image (no bound)
    assert set(counts["conv3x3"]) <= allowed and conv_math in counts["conv3x3"] and set(counts["corr"]) == {conv_math}, counts
    # stage-boundary tensors of SURVEY.md §3.2 against what the reference's forward hooks captured (strided fixtures)
    tp = net.taps
    # both norms: max |err| / max |ref| (what the north star's "1e-3 relative" bounds) and relative L2 (which small-magnitude
    # regions of a feature map cannot hide behind its largest value)
    for err in (rel_err, rel_l2):
        assert err(tp["supp_d4"][0, 0][:, ::s_d4], g["supp_d4"]) < TOL and err(tp["qry_d4"][:, ::s_d4], g["qry_d4"]) < TOL
        assert err(tp["supp_fts"][0][0][..., ::s_f, ::s_f], g["supp_fts"]) < TOL
        assert err(tp["protos"], g["protos"]) < TOL
        for i in range(T):
            assert err(tp[f"inter_{i}"][..., ::s_f, ::s_f], g[f"inter_{i}"]) < TOL, f"inter[{i}]"
            assert err(out["refinement"][i][..., ::s_out, ::s_out], g[f"refinement_{i}"]) < TOL, f"refinement[{i}]"
    flips = 0
    for i in range(T):
        got = out["refinement"][i]
        assert rel_err(got[..., ::s_out, ::s_out], g[f"refinement_{i}"]) < TOL, f"refinement[{i}]"
        p = got.softmax(1)[:, 1]
        assert abs(float((p > 0.5).float().mean()) - float(g[f"fg_frac_{i}"])) <= 1e-3
        pred = (p > 0.5).long()
        dice = 2.0 * (pred * ql).sum() / (pred.sum() + ql.sum() + 1e-7)
        assert abs(float(dice) - float(g[f"dice_{i}"])) <= 1e-3, f"Dice deviation at iteration {i}"   # north_star bar
        nm = torch.nn.functional.avg_pool2d((p > 0.5).float()[:, None], 4)
        flips += int((nm.cpu() != torch.from_numpy(g[f"next_mask_{i}"])).sum())
    assert rel_err(loss, g["loss"]) < TOL
    if training:
        assert rel_err(out["align_loss"], g["align_loss"]) < TOL
        loss.backward()
        unused = set(str(u) for u in g["unused"])
        params = dict(net.named_parameters())
        worst = 0.0
        for n, ref, head in zip(g["grad_names"], g["grad_norms"], g["grad_heads"]):
            n = str(n)
            gr = params[n].grad
            if n in unused:
                assert gr is None, n
                continue
            if ref < 1e-4:          # conv biases in front of train-mode BN: analytically zero
                assert gr.abs().max() < 1e-4
                continue
            # Gradient NORMS against the reference's fixtures.  The encoder gradients are conditioned by ReLU / max-pool /
            # threshold switches (tests/test_oracle_conditioning.py), so their bound is not a constant but the measured
            # conditioning of THIS episode: the reference's fp32 gradients and the HIP path's each lie within 3 yardsticks of
            # the fp64 oracle's (test_gradients_vs_fp64_yardstick holds the HIP path to that, element-wise), hence their norms
            # within 6 of each other.  The smooth CRE block is held to 1e-3 (norms) and 4e-3 (first 32 elements).
            enc = n.startswith("encoder.")
            e = abs(gr.double().norm().item() - ref) / ref
            worst = max(worst, e)
            if enc:
                y = _yardstick(tag, g)[3][n]
                assert e <= 6.0 * y + 1e-6, f"grad norm {n}: rel {e:.2e}, yardstick {y:.2e}"
                continue
            assert e < 1e-3, f"grad norm {n}: rel {e:.2e}"
            k = min(32, gr.numel())
            hd = torch.from_numpy(head[:k])
            he = (gr.flatten()[:k].cpu() - hd).abs().max() / (hd.abs().max() + 1e-12)
            # (CRE heads: 32 elements of a BatchNorm gamma / first filter; measured <= 2.6e-3 over the seven fixtures x three arithmetics)
            assert he < 4e-3, f"grad head {n}: rel {he:.2e}"
        sd = net.state_dict()
        for k in g:
            if k.startswith("sd."):
                assert rel_err(sd[k[3:]].float(), g[k]) < 1e-4, k
    print(f"{tag}: threshold flips vs reference {flips}")


def test_teacher_forced_iterations(golden, conv_math):
    """Each refinement iteration reproduced independently from the reference's own mask
    (hard-threshold flips cannot compound)."""
    from rpnet_amd import functional as RF
    g = golden("m64_train")
    size, B, T, _, seed = (int(v) for v in g["meta"])
    cfg = load_cfg(T)
    (si, fg, bg, qi, ql, appr), _ = episode_tensors(seed, B, size, DEV)
    net = build(cfg, True)
    with torch.no_grad():
        out = net(si, fg, bg, qi, appr_query_labels=appr)          # also builds BN state like the reference step
        net2 = build(cfg, True)
        cache = RF.WeightCache()
        x = torch.cat([si[0][0], qi[0]], 0).reshape(2 * B, size, size, 1)
        d4 = net2.encoder.forward_nhwc(x, cache, groups=2).x
        sm = RF.mask_avgpool(fg[0][0], 4)
        sf = net2.cre.forward_masked(d4[:B].contiguous(), sm, cache)
        am, msum = RF.mask_adjoint(torch.stack([bg[0][0], fg[0][0]], 0), size // 4, size // 4)
        protos = RF.MaskedPool.apply(sf, am, msum)
        assert rel_err(protos, g["protos"]) < TOL
        mask = RF.mask_avgpool(appr, 4)
        for i in range(T):
            inter = net2.cre.forward_masked(d4[B:].contiguous(), mask, cache)
            logits, _ = RF.CosineMatchUp.apply(inter, protos, size, size, 20.0)
            assert rel_err(logits, g[f"refinement_{i}"]) < TOL
            assert rel_err(nchw(inter), g[f"inter_{i}"]) < TOL
            mask = torch.from_numpy(g[f"next_mask_{i}"])[:, 0].to(DEV)   # the REFERENCE's mask
    assert rel_err(out["output"], g["output"]) < TOL


def test_full_size_properties():
    """BASELINE configs[1] shape (1-way 1-shot, 256x256, T=5, batch 8): size-independent checks."""
    cfg = load_cfg(5)
    (si, fg, bg, qi, ql, appr), _ = episode_tensors(1234, 8, 256, DEV)
    net = build(cfg, True)
    out = net(si, fg, bg, qi, appr_query_labels=appr)
    loss = total_loss(out, ql, 1.0)
    loss.backward()
    assert out["output"].shape == (8, 2, 256, 256) and len(out["refinement"]) == 5
    assert torch.isfinite(loss) and all(torch.isfinite(p.grad).all() for p in net.parameters() if p.grad is not None)
    assert (out["output"].abs() <= 20.0 + 1e-3).all()                         # logits are 20 * cosine
    sd = net.state_dict()
    assert int(sd["encoder.Conv1.conv.1.num_batches_tracked"]) == 2            # support call + query call
    assert int(sd["cre.w_k.1.num_batches_tracked"]) == 6                        # 1 + T CRE calls
    assert int(sd["cre.out.1.num_batches_tracked"]) == 0                        # never used
    g1 = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    # the default arithmetic at the default threshold: every 3x3 layer (forward, input and weight gradient) and every
    # correlation of the configs[1] step ran on two fp16 planes; only the 1x1 convolution is on the fp32 kernels
    from rpnet_amd import functional as RF
    assert RF.conv_math() == "f16x2"               # the library default (tests/conftest.py restores it around every test)
    RF.reset_arith()
    out_c = net(si, fg, bg, qi, appr_query_labels=appr)
    total_loss(out_c, ql, 1.0).backward()
    counts = RF.arith_counts()
    # (Up5 / Up4 — forward, input gradient, weight gradient — on their collapsed four-product form: RF._UP4)
    assert counts["conv3x3"] == {"f16x2": 50} and counts["conv3x3_up4"] == {"f16x2": 4}, counts
    assert counts["wgrad3x3"] == {"f16x2": 25} and counts["wgrad3x3_up4"] == {"f16x2": 2}, counts
    assert counts["corr"] == {"f16x2": 6} and counts["corr_bwd"] == {"f16x2": 6}, counts
    # (Conv1.conv.0 of both... of the one encoder launch: its reduction pass makes the pre-BatchNorm tensor again from the image)
    assert counts["bn_bwd"] == {"own reduction pass": 33, "first layer made again from the image": 1}, counts
    # episodes are independent given fixed BN statistics: in eval mode a batch equals its halves
    net.eval()
    with torch.no_grad():
        full = net(si, fg, bg, qi, appr_query_labels=appr)["output"]
        half = net([[si[0][0][:4]]], [[fg[0][0][:4]]], [[bg[0][0][:4]]], [qi[0][:4]], appr_query_labels=appr[:4])["output"]
    assert rel_err(full[:4], half) < 1e-5
    # determinism: same step twice -> bit-identical gradients (no atomics anywhere)
    net2 = build(cfg, True)
    out2 = net2(si, fg, bg, qi, appr_query_labels=appr)
    total_loss(out2, ql, 1.0).backward()
    for n, p in net2.named_parameters():
        if p.grad is not None:
            assert torch.equal(p.grad, g1[n]), n


def test_eval_call_full_size_all_paths_agree():
    """The reference driver's call (test_rpnet.py:189-215: eval mode, 2 slices, 256x256, T = 10) at full size under the default
    arithmetic — fp16 planes on predicted scales, split K on the 16^2 / 32^2 levels, the two CRE convolutions on two
    streams — against the same call on the fp32 matrix instruction (no planes, no scales, no split), eager and replayed
    from its HIP graph, over four consecutive calls (the first measures, the others predict)."""
    from rpnet_amd import functional as RF
    from rpnet_amd.graph import GraphedEval
    cfg = load_cfg(10)
    eps = [episode_tensors(seed, 2, 256, DEV)[0] for seed in (51, 52, 53, 54)]
    RF.set_conv_math("f32")
    ref_net = build(cfg, False)
    with torch.no_grad():
        want = [ref_net(si, fg, bg, qi, appr_query_labels=appr) for (si, fg, bg, qi, ql, appr) in eps]
        want = [[o["refinement"][i].clone() for i in range(10)] for o in want]
    RF.set_conv_math("f16x2")
    net = build(cfg, False)
    lent, orig = [], RF.call

    def spy(name, *args):
        if name == "rpnet_conv_fwd":
            lent.append(int(bool(args[0]._obj.splitk_ws)))
        return orig(name, *args)

    RF.call = spy
    base = RF.pred_stats()
    try:
        with torch.no_grad():
            RF.reset_arith()
            got = [net(si, fg, bg, qi, appr_query_labels=appr) for (si, fg, bg, qi, ql, appr) in eps]
            got = [[o["refinement"][i].clone() for i in range(10)] for o in got]
    finally:
        RF.call = orig
    st = RF.pred_stats()
    assert set(RF.arith_counts()["conv3x3"]) == {"f16x2"}
    assert st["predicted_calls"] - base["predicted_calls"] == 3 and st["violations"] == base["violations"], (base, st)
    # per call: Conv5 (two layers, M = 1024) and Up_conv5's 1024 -> 512 layer at M = 4096 borrow the split-K workspace (Up5 itself runs
    # on the collapsed up_conv form since round 6: rpnet_conv_up4, four K-steps per channel chunk, no split)
    assert sum(lent) == 4 * 3, sum(lent)
    for w, g in zip(want, got):
        # teacher-free: the hard threshold of the fed-back mask may flip pixels between arithmetics; iteration 0 has no
        # feedback, the later ones are compared through the share of pixels whose class differs
        assert rel_err(g[0], w[0]) < 2e-5
        for i in range(1, 10):
            flips = (g[i].argmax(1) != w[i].argmax(1)).float().mean().item()
            assert flips < 2e-3, (i, flips)
    graphed = GraphedEval(net)
    for (si, fg, bg, qi, ql, appr), g in zip(eps[2:], got[2:]):
        out = graphed(si, fg, bg, qi, appr_query_labels=appr)
        for i in range(10):
            assert rel_err(out["refinement"][i], g[i]) <= 1e-6, i


@pytest.mark.parametrize("math", ["f16x2", "bf16x3", "f16"])
def test_bn_relu_maxpool_in_one_pass(monkeypatch, math):
    """rpnet_bn_relu(pool_w) / rpnet_bn_bwd(pool_w): x1 and x2 of the encoder feed nothing but their MaxPool2d(2, 2)
    (net/unet.py:442-448), so in training the pooled operand planes come out of the BatchNorm + ReLU pass and the backward
    finds the window maxima again from the conv output.  Same logits (bit-identical: max and ReLU commute) and the same
    gradients as BatchNorm + ReLU, max-pool and operand split as three launches; two of the 25 BatchNorm passes take it."""
    from rpnet_amd import functional as RF
    from rpnet_amd import modules as RM
    RM._F16_MIN_PIXELS = 0
    RF.set_conv_math(math)
    cfg = load_cfg(2)
    (si, fg, bg, qi, ql, appr), _ = episode_tensors(322, 4, 128, DEV)
    grads, outs = {}, {}
    for fuse in (False, True):
        monkeypatch.setattr(RF, "_POOL_FUSE", fuse)
        net = build(cfg, True)
        RF.reset_arith()
        out = net(si, fg, bg, qi, appr_query_labels=appr)
        total_loss(out, ql, 1.0).backward()
        bnr = dict(RF.arith_counts().get("bn_relu", {}))
        bnr.pop("first layer made again from the image", None)
        assert bnr == ({"with the 2x2 max-pool": 2} if fuse else {}), RF.arith_counts()
        outs[fuse] = out["output"].detach().clone()
        grads[fuse] = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    assert torch.equal(outs[True], outs[False])
    for n, gr in grads[False].items():
        assert rel_l2(grads[True][n], gr) < 1e-5 or float(gr.abs().max()) < 1e-6, n


def test_conv1_weight_gradient_from_bn_output_gradient(monkeypatch):
    """rpnet_conv1_wgrad_bn: Conv1.conv.0 (Cin = 1, no input gradient) forms dy inside its direct weight gradient from dz,
    y and the coefficients of rpnet_bn_bwd's reduction pass (dy == dy_split == NULL) instead of reading the output of a
    separate apply pass: same gradients (two BatchNorm groups, so both coefficient rows are exercised)."""
    from rpnet_amd import functional as RF
    from rpnet_amd import modules as RM
    RM._F16_MIN_PIXELS = 0
    cfg = load_cfg(2)
    (si, fg, bg, qi, ql, appr), _ = episode_tensors(323, 2, 64, DEV)
    grads = {}
    for fuse in (False, True):
        monkeypatch.setattr(RF, "_CONV1_BN_FUSE", fuse)
        net = build(cfg, True)
        out = net(si, fg, bg, qi, appr_query_labels=appr)
        total_loss(out, ql, 1.0).backward()
        grads[fuse] = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    for n, gr in grads[False].items():
        tol = 1e-5 if n.startswith("encoder.Conv1.conv.0") or n.startswith("encoder.Conv1.conv.1") else 0.0
        if tol:
            assert rel_l2(grads[True][n], gr) < tol or float(gr.abs().max()) < 1e-6, n
        else:
            assert torch.equal(grads[True][n], gr), n


def test_five_shot_extension_vs_composed_oracle():
    """BASELINE config 3 shape class (multi-shot): no reference behaviour (net/rp_net.py:275,288
    raise for n_shots > 1); pinned by the oracle composed from the reference's own pieces."""
    from oracle import rpnet_oracle as O
    cfg = load_cfg(2)
    (si, fg, bg, qi, ql, appr), _ = episode_tensors(55, 2, 64, "cpu", n_shots=3)
    P = O.seeded_params()
    with torch.no_grad():
        ref = O.rp_net_forward(P, cfg, si, fg, bg, qi, appr, True, align=False)
    net = build(cfg, True)
    mv = lambda t: t.to(DEV)  # noqa: E731
    with torch.no_grad():
        out = net([[mv(s) for s in w] for w in si], [[mv(s) for s in w] for w in fg], [[mv(s) for s in w] for w in bg],
                  [mv(qi[0])], appr_query_labels=mv(appr))
    for i in range(2):
        assert rel_err(out["refinement"][i], ref["refinement"][i]) < TOL


@pytest.mark.parametrize("tag", ["m64_5shot", "m64_2way", "m64_2way2shot"])
def test_extension_rows_vs_composed_reference(golden, tag, conv_math):
    """BASELINE configs[2] / configs[4] shape classes (multi-shot, multi-way): the reference's forward has no behaviour there
    (net/rp_net.py:275,288 fail), so SURVEY.md §8a defines it as the composition of the reference's OWN encoder, cre,
    getFeatures, getPrototype, calDist and alignLoss per (way, shot); tests/golden/gen_golden.py::gen_composed runs exactly
    that on the imported reference and stores stage taps, logits, align loss, loss, every gradient norm / head and the
    BatchNorm buffers.  The HIP path against those fixtures, under all three arithmetics, all gradients."""
    g = golden(tag)
    size, B, T, _, seed, n_ways, n_shots = _meta(g)
    cfg = load_cfg(T)
    (si, fg, bg, qi, ql, appr), ep = episode_tensors(seed, B, size, DEV, n_shots=n_shots, n_ways=n_ways)
    assert np.allclose(in_checksum(ep), g["in_checksum"], rtol=0, atol=1e-6), "synthetic inputs drifted"
    s_d4 = int(g["strides"][1])
    net = build(cfg, True)
    net.taps = {}
    from rpnet_amd import functional as RF
    RF.reset_arith()
    out = net(si, fg, bg, qi, appr_query_labels=appr)
    loss = total_loss(out, ql, cfg["align_loss_scaler"])
    counts = RF.arith_counts()
    assert set(counts["conv3x3"]) == {conv_math} and set(counts["corr"]) == {conv_math}, counts
    assert out["output"].shape == (B, 1 + n_ways, size, size) and out["output"] is out["refinement"][T - 1]
    tp = net.taps
    for err in (rel_err, rel_l2):
        assert err(tp["supp_d4"][:, :, :, ::s_d4], g["supp_d4"]) < TOL and err(tp["qry_d4"][:, ::s_d4], g["qry_d4"]) < TOL
        for wa in range(n_ways):
            for s_ in range(n_shots):
                assert err(tp["supp_fts"][wa][s_], g["supp_fts"][wa, s_]) < TOL, (wa, s_)
        assert err(tp["protos"], g["protos"]) < TOL
        for i in range(T):
            assert err(tp[f"inter_{i}"], g[f"inter_{i}"]) < TOL, f"inter[{i}]"
            assert err(out["refinement"][i], g[f"refinement_{i}"]) < TOL, f"refinement[{i}]"
    for i in range(T):
        p = out["refinement"][i].softmax(1)[:, 1]
        assert abs(float((p > 0.5).float().mean()) - float(g[f"fg_frac_{i}"])) <= 1e-3
        pred = (p > 0.5).long()
        dice = 2.0 * (pred * ql).sum() / (pred.sum() + ql.sum() + 1e-7)
        assert abs(float(dice) - float(g[f"dice_{i}"])) <= 1e-3, f"Dice deviation at iteration {i}"
    assert rel_err(loss, g["loss"]) < TOL and rel_err(out["align_loss"], g["align_loss"]) < TOL
    loss.backward()
    unused = set(str(u) for u in g["unused"])
    params = dict(net.named_parameters())
    yard = _yardstick(tag, g)[3]
    for n, ref, head in zip(g["grad_names"], g["grad_norms"], g["grad_heads"]):
        n = str(n)
        gr = params[n].grad
        if n in unused:
            assert gr is None, n
            continue
        if ref < 1e-4:
            assert gr.abs().max() < 1e-4
            continue
        e = abs(gr.double().norm().item() - ref) / ref
        if n.startswith("encoder."):     # conditioned by ReLU / max-pool switches: bound = this episode's measured conditioning (test_model_vs_golden)
            assert e <= 6.0 * yard[n] + 1e-6, f"grad norm {n}: rel {e:.2e}, yardstick {yard[n]:.2e}"
            continue
        assert e < 1e-3, f"grad norm {n}: rel {e:.2e}"
        k = min(32, gr.numel())
        hd = torch.from_numpy(head[:k])
        he = (gr.flatten()[:k].cpu() - hd).abs().max() / (hd.abs().max() + 1e-12)
        # (a spot check of 32 elements behind the norm check above: the CRE's parameters collect the gradients of
        # n_ways * n_shots + T calls here, each with its own ReLU switches, and a switched element moves single entries by ~1e-2 of
        # the head's maximum while the norm stays within 1e-3: measured 4.6e-3 / 1.1e-2 on m64_5shot under bf16x3; the 1-shot
        # fixtures hold 4e-3)
        assert he < 2e-2, f"grad head {n}: rel {he:.2e}"
    sd = net.state_dict()
    for k in g:
        if k.startswith("sd."):
            assert rel_err(sd[k[3:]].float(), g[k]) < 1e-4, k
    assert int(sd["cre.w_k.1.num_batches_tracked"]) == n_ways * n_shots + T


def test_soft_mask_training_vs_oracle():
    """soft_mask: True — the fed-back mask stays differentiable (net/rp_net.py:309): forward and
    gradients against the CPU oracle (autograd through softmax -> avg_pool -> x*mask)."""
    from oracle import rpnet_oracle as O
    cfg = load_cfg(3)
    cfg["soft_mask"] = True
    (si, fg, bg, qi, ql, appr), _ = episode_tensors(91, 2, 64, "cpu")
    P = O.seeded_params(requires_grad=True)
    ref = O.rp_net_forward(P, cfg, si, fg, bg, qi, appr, True)
    ref_loss = O.total_loss(ref, ql)
    ref_loss.backward()
    net = build(cfg, True)
    mv = lambda t: t.to(DEV)  # noqa: E731
    out = net([[mv(si[0][0])]], [[mv(fg[0][0])]], [[mv(bg[0][0])]], [mv(qi[0])], appr_query_labels=mv(appr))
    loss = total_loss(out, mv(ql), 1.0)
    loss.backward()
    for i in range(3):
        assert rel_err(out["refinement"][i], ref["refinement"][i]) < TOL
    assert rel_err(loss, ref_loss) < TOL
    for n, p in net.named_parameters():
        if p.grad is None or P[n].grad.norm() < 1e-4:
            continue
        a, b = p.grad.double().cpu().norm().item(), P[n].grad.double().norm().item()
        tol = 1e-2 if n.startswith("encoder.") else 2e-3
        assert abs(a - b) < tol * b, f"{n}: {a} vs {b}"
    # the mask gradient really flows: hard-mask gradients differ
    cfg_h = load_cfg(3)
    net_h = build(cfg_h, True)
    out_h = net_h([[mv(si[0][0])]], [[mv(fg[0][0])]], [[mv(bg[0][0])]], [mv(qi[0])], appr_query_labels=mv(appr))
    total_loss(out_h, mv(ql), 1.0).backward()
    assert rel_err(net_h.cre.q[0].weight.grad, net.cre.q[0].weight.grad) > 1e-3


def test_eval_driver_end_to_end():
    """The evaluation loop of the reference's driver shape (tools/eval_driver.py) over the synthetic
    reader, eval mode, T = n_test_iter_refinement: runs through the HIP path and yields Dice values."""
    from dataset.few_shot_reader import FewshotRegReader
    from tools.eval_driver import evaluate
    cfg = load_cfg()
    cfg["n_iter_refinement"] = cfg["n_test_iter_refinement"]
    ds = FewshotRegReader("/nonexistent", cfg["eval_set_name"], cfg, mode="eval", n_volumes=2, n_slices=4, size=128)
    net = build(cfg, False)
    aff, few, ref = evaluate(net, ds, cfg, n_items=2)
    assert len(few["Liver"]) == 2 and all(0.0 <= d <= 1.0 for d in few["Liver"])
    assert sorted(ref["Liver"].keys()) == list(range(10))
    assert few["Liver"] == [r for r in ref["Liver"][9]]          # output == refinement[T-1]


def test_eval_driver_over_nrrd_volumes(tmp_path):
    """the same loop fed by the real readers: NRRD volumes on disk -> FewshotVolumeReader / SliceReader ->
    HIP registration pre-step -> RP_Net (T = 10) -> Dice per volume (SURVEY §8f rows 1, 2, 4 chained)"""
    from dataset.few_shot_reader import FewshotRegReader
    from rpnet_amd.utils.volume_reader import FewshotRegReader as RealReader, write_synthetic_dataset
    from tools.eval_driver import evaluate
    data_dir, set_name, csv_dir = write_synthetic_dataset(str(tmp_path), n_volumes=3, classes=("Liver",), shape=(14, 72, 68), seed=5)
    cfg = load_cfg()
    cfg.update(n_iter_refinement=cfg["n_test_iter_refinement"], class_csv_dir=csv_dir, eval_classes=["Liver"], crop_size=[64, 64],
               k=4, use_registration_loss=True, do_deformable=False)
    ds = FewshotRegReader(data_dir, set_name, cfg, mode="eval")
    assert isinstance(ds, RealReader) and len(ds) == 3
    net = build(cfg, False)
    aff, few, ref = evaluate(net, ds, cfg)
    assert len(few["Liver"]) == 3 and all(0.0 <= d <= 1.0 for d in few["Liver"])
    assert all(d is not None and d > 0.2 for d in aff["Liver"])       # the affine-warped support label overlaps the organ
    assert sorted(ref["Liver"].keys()) == list(range(10)) and few["Liver"] == list(ref["Liver"][9])


def test_reference_driver_launcher(tmp_path):
    """tools/run_reference_driver.py: a driver script with the two lines that stop the reference's literal test_rpnet.py on a
    one-GPU ROCm box — `os.environ['CUDA_VISIBLE_DEVICES'] = '1'` in front of `import torch` (:3) and the tensorboard import
    (:27) — still sees the GPU and runs the evaluation loop on the HIP path when started through the launcher."""
    import subprocess
    import sys
    script = tmp_path / "driver_like_the_reference.py"
    script.write_text(
        "import os\n"
        "os.environ['CUDA_VISIBLE_DEVICES'] = '1'\n"
        "import torch\n"
        "from torch.utils.tensorboard import SummaryWriter\n"
        "from net.model import model_factory\n"
        "from dataset.few_shot_reader import FewshotRegReader\n"
        "from utils.util import load_yaml\n"
        "from tools.eval_driver import evaluate\n"
        "import sys\n"
        "config, args = load_yaml(sys.argv[1])\n"
        "config['n_iter_refinement'] = 2\n"
        "assert torch.cuda.device_count() >= 1, 'the device mask took effect'\n"
        "w = SummaryWriter('runs'); w.add_scalar('x', 1.0, 0); w.close()\n"
        "ds = FewshotRegReader('/nonexistent', config['eval_set_name'], config, mode='eval', n_volumes=2, n_slices=4, size=64)\n"
        "net = model_factory['RP_Net'](pretrained_path=None, cfg={'align': True, 'backbone': 'UNet'}, backbone_cfg=config).cuda()\n"
        "net.eval()\n"
        "aff, few, ref = evaluate(net, ds, config, n_items=1)\n"
        "print('DRIVER_OK', len(few['Liver']))\n")
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "run_reference_driver.py"), str(script),
                        os.path.join(root, "yamls", "example.yml")], capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0 and "DRIVER_OK 1" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])
    # and WITHOUT the launcher the same script loses the device (this is what the literal file does on a one-GPU box)
    env = dict(os.environ, PYTHONPATH=root)
    r2 = subprocess.run([sys.executable, str(script), os.path.join(root, "yamls", "example.yml")], capture_output=True, text=True,
                        timeout=900, cwd=str(tmp_path), env=env)
    assert r2.returncode != 0


def test_two_way_extension_vs_composed_oracle():
    """BASELINE config 5 shape class (2-way, fp32 here): no reference behaviour; oracle composed from the
    reference's own pieces, incl. gradients of the well-conditioned CRE block and the align loss."""
    from oracle import rpnet_oracle as O
    cfg = load_cfg(2)
    (si, fg, bg, qi, ql, appr), _ = episode_tensors(66, 2, 64, "cpu", n_shots=1, n_ways=2)
    P = O.seeded_params(requires_grad=True)
    ref = O.rp_net_forward(P, cfg, si, fg, bg, qi, appr, True, align=True)
    ref_loss = ref["refinement"][1].square().mean() + ref["align_loss"]
    ref_loss.backward()
    net = build(cfg, True)
    mv = lambda t: t.to(DEV)  # noqa: E731
    out = net([[mv(s) for s in w] for w in si], [[mv(s) for s in w] for w in fg], [[mv(s) for s in w] for w in bg],
              [mv(qi[0])], appr_query_labels=mv(appr))
    assert out["output"].shape == (2, 3, 64, 64)
    for i in range(2):
        assert rel_err(out["refinement"][i], ref["refinement"][i]) < TOL
    assert rel_err(out["align_loss"], ref["align_loss"]) < TOL
    (out["refinement"][1].square().mean() + out["align_loss"]).backward()
    for n in ("cre.q.0.weight", "cre.w_k.0.weight", "cre.q.1.bias"):
        a, b = dict(net.named_parameters())[n].grad.double().cpu(), P[n].grad.double()
        assert (a - b).norm() < 5e-3 * b.norm(), n


@pytest.mark.parametrize("async_wgrad,math,prepack", [(False, "f16x2", True), (True, "f16x2", True), (True, "f32", True),
                                                        (True, "f16x2", False), (False, "bf16x3", False)])
def test_encoder_two_chains_match_one_stream(async_wgrad, math, prepack):
    """2-way (support call: 2 B images, query call: B): the encoder's two calls as two chains on two HIP streams, their
    BatchNorm modules' running statistics and parameter gradients ordered by events (RF.order_begin), against the same
    step on one stream: same kernels in the same order per buffer -> logits, every gradient and every BatchNorm buffer
    bit-identical; with and without the weight gradients on their side stream, two runs of the two-chain step.
    Also under the arithmetics / switches in which NO prepack launch makes the weight packs up front (fp32 kernels,
    RPNET_PREPACK=0): both chains share the cached packs, which RP_Net.forward then materialises in front of the fork
    (WeightCache.materialize) — a pack made lazily on one chain's stream would be a race for the other."""
    import rpnet_amd.functional as RF
    import rpnet_amd.modules as RM
    from rpnet_amd.parallel import FlatGradBucket
    cfg = load_cfg(2)
    (si, fg, bg, qi, ql, appr), _ = episode_tensors(91, 4, 128, DEV, n_shots=1, n_ways=2)
    was, was_pp = RM._ENC_STREAMS, RM._PREPACK
    RF.set_conv_math(math)
    RM._F16_MIN_PIXELS = 0
    RM._PREPACK = prepack
    res = []
    try:
        for enc in (0, 1, 1):
            RM._ENC_STREAMS = enc
            net = build(cfg, True)
            bucket = FlatGradBucket(net) if async_wgrad else None
            RF.set_async_wgrad(async_wgrad)
            if bucket is not None:
                bucket.zero()
            out = net(si, fg, bg, qi, appr_query_labels=appr)
            total_loss(out, ql, 1.0).backward()
            if bucket is not None:
                bucket.allreduce()
            torch.cuda.synchronize()
            res.append((out["output"].detach().clone(), {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None},
                        {n: b.clone() for n, b in net.named_buffers()}))
    finally:
        RM._ENC_STREAMS, RM._PREPACK = was, was_pp
        RF.set_async_wgrad(False)
    for other in res[1:]:
        assert torch.equal(res[0][0], other[0])
        for n, g in res[0][1].items():
            assert torch.equal(g, other[1][n]), n
        for n, b in res[0][2].items():
            assert torch.equal(b, other[2][n]), n


@pytest.mark.parametrize("name,size,B,T,ways,math,repeats", [("configs[1]", 256, 8, 5, 1, "f16x2", 30), ("configs[4]", 512, 4, 10, 2, "f16", 10)])
def test_default_schedule_repeats_bit_for_bit(name, size, B, T, ways, math, repeats):
    """The canary of round 4's pooled-pass fault (profiles/r04_pool_apply_fault.txt: in roughly one step of four a pooled
    BatchNorm-backward pass that shared its CUs with an LDS-DMA weight-gradient block put gradients on the wrong pixel of their
    2 x 2 window — silently; only a flaky small test gave it away).  The DEFAULT schedule of the benched step — weight gradients
    on their side stream released behind their layer's dgrad, the CRE's second branch on its own stream, (configs[4]) the
    encoder's two calls as two chains — at the benched size, `repeats` times: every gradient and every BatchNorm buffer of
    every repeat must have the bits of the SAME step run on one stream.  Same kernels, same order per buffer: any difference
    is a race or a hardware-visible hazard between co-resident kernels, and any new pass / tile variant / LDS change that
    brings one back fails here."""
    import rpnet_amd.functional as RF
    import rpnet_amd.modules as RM
    from rpnet_amd.parallel import FlatGradBucket
    cfg = load_cfg(T)
    (si, fg, bg, qi, ql, appr), _ = episode_tensors(77, B, size, DEV, n_shots=1, n_ways=ways)
    RF.set_conv_math(math)
    was = (RM._ENC_STREAMS, RM._CRE_STREAMS_TRAIN, RF._MASK_SKIP)
    net = build(cfg, True)
    bucket = FlatGradBucket(net)
    state0 = {k: v.clone() for k, v in net.state_dict().items()}

    dev = torch.device(DEV)
    side_keys = (dev, ("reduce", dev), ("cre", dev), ("pack", dev))

    def one(multi_stream):
        net.load_state_dict(state0)          # the running statistics back to the start: every step sees the same state
        # the SAME code path both times (weight gradients accumulated by their kernels straight into the bucket); the one-stream
        # reference aliases every side stream to the caller's stream, so that the launches run one after the other in issue order
        RF.set_async_wgrad(True)
        saved = {k: RF._ASYNC["side"].pop(k, None) for k in side_keys}
        if not multi_stream:
            for k in side_keys:
                RF._ASYNC["side"][k] = torch.cuda.current_stream(dev)
        else:
            RF._ASYNC["side"].update({k: v for k, v in saved.items() if v is not None})
        RM._CRE_STREAMS_TRAIN = was[1] if multi_stream else False
        RM._ENC_STREAMS = was[0] if multi_stream else 0
        bucket.zero()
        out = net(si, fg, bg, qi, appr_query_labels=appr)
        total_loss(out, ql, 1.0).backward()
        bucket.allreduce()
        torch.cuda.synchronize()
        if not multi_stream:
            for k in side_keys:
                RF._ASYNC["side"].pop(k, None)
            RF._ASYNC["side"].update({k: v for k, v in saved.items() if v is not None})
        return bucket.flat.clone(), {n: b.clone() for n, b in net.named_buffers()}, out["output"].detach().clone()

    try:
        ref_flat, ref_buf, ref_out = one(False)
        assert torch.isfinite(ref_flat).all() and float(ref_flat.abs().max()) > 0
        bad = []
        for r in range(repeats):
            flat, buf, out = one(True)
            if not torch.equal(flat, ref_flat):
                nd = int((flat != ref_flat).sum())
                where = [n for n, p in net.named_parameters() if p.grad is not None and not torch.equal(p.grad, p.grad)]   # (NaN check)
                bad.append((r, nd, float((flat - ref_flat).abs().max()), where))
            assert torch.equal(out, ref_out), (name, r, "logits")
            for n in ref_buf:
                assert torch.equal(buf[n], ref_buf[n]), (name, r, n)
        assert not bad, f"{name}: {len(bad)} of {repeats} multi-stream steps differ from the one-stream step: {bad[:5]}"
    finally:
        RM._ENC_STREAMS, RM._CRE_STREAMS_TRAIN, RF._MASK_SKIP = was
        RF.set_async_wgrad(False)


def test_pooled_pass_guards_hold_where_the_fault_was_most_frequent():
    """tools/canary_two_chains.py in its own process (the switches are read once per process): the configuration in which the
    pooled BatchNorm-backward fault showed in 8 of 8 steps WITHOUT the guards — two encoder chains on two streams, the large LDS
    form of the BatchNorm reductions (RPNET_BN_LDS=big) — with the library's guards ON (RPNET_BN_POOL_DRAIN / RPNET_BN_POOL_ALONE,
    csrc/bn.hip): every repeat must have the bits of the first run.  (profiles/r05_pool_fault_repro.txt has the same command with
    the guards off, the stand-alone kernel pair, and the load-return probe.)"""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RPNET_BN_POOL_DRAIN", "RPNET_BN_POOL_ALONE")}
    env["RPNET_BN_LDS"] = "big"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "canary_two_chains.py"), "12"], capture_output=True, text=True, env=env,
                       cwd=ROOT, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if "repeats differ" in ln]
    assert r.returncode == 0 and len(lines) == 2, r.stdout[-2000:] + r.stderr[-2000:]
    for ln in lines:
        assert ": 0 of 12 repeats differ" in ln, ln


@pytest.mark.parametrize("math,ways", [("f16x2", 1), ("f16", 2)])
def test_first_layer_without_its_pre_batchnorm_tensor(math, ways):
    """Conv1.conv.0 (Cin = 1) in training on fp16 planes: its pre-BatchNorm tensor is never written — statistics, BatchNorm +
    ReLU, the backward's reduction pass and the weight gradient each make it again from the image (RF._CONV1_RECOMP,
    csrc/conv_first.hip).  One definition of a value -> the forward is BIT-identical to the step that stores the tensor
    (logits, BatchNorm buffers) and so is every gradient of every other layer; the layer's own three gradients come from
    another order of the same sums (1e-5).  Launch counters show that the path ran."""
    import rpnet_amd.functional as RF
    import rpnet_amd.modules as RM
    cfg = load_cfg(2)
    (si, fg, bg, qi, ql, appr), _ = episode_tensors(93, 2, 64, DEV, n_ways=ways)
    RF.set_conv_math(math)
    RM._F16_MIN_PIXELS = 0
    was = RF._CONV1_RECOMP
    res = []
    try:
        for on in (False, True):
            RF._CONV1_RECOMP = on
            net = build(cfg, True)
            RF.reset_arith()
            out = net(si, fg, bg, qi, appr_query_labels=appr)
            total_loss(out, ql, 1.0).backward()
            torch.cuda.synchronize()
            counts = RF.arith_counts()
            assert (("first layer made again from the image" in counts.get("bn_relu", {})) == on), counts
            assert (("first layer made again from the image" in counts.get("bn_bwd", {})) == on), counts
            res.append((out["output"].detach().clone(), {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None},
                        {n: b.clone() for n, b in net.named_buffers()}))
    finally:
        RF._CONV1_RECOMP = was
    (lo0, g0, b0), (lo1, g1, b1) = res
    assert torch.equal(lo0, lo1)
    for n in b0:
        assert torch.equal(b0[n], b1[n]), n
    own = ("encoder.Conv1.conv.0.weight", "encoder.Conv1.conv.1.weight", "encoder.Conv1.conv.1.bias")
    for n in g0:
        if n in own:
            assert rel_l2(g1[n], g0[n]) <= 1e-5, (n, rel_l2(g1[n], g0[n]))
        else:
            assert torch.equal(g0[n], g1[n]), n


def test_config3_full_size_properties():
    """BASELINE configs[2]: 1-way 5-shot, 256x256, T=5, batch 16 — full-size forward + backward through
    the multi-shot path: shapes, finiteness, BatchNorm update counts, prototype = mean over shots."""
    cfg = load_cfg(5)
    (si, fg, bg, qi, ql, appr), _ = episode_tensors(4321, 16, 256, DEV, n_shots=5)
    net = build(cfg, True)
    out = net(si, fg, bg, qi, appr_query_labels=appr)
    loss = total_loss(out, ql, 1.0)
    loss.backward()
    assert out["output"].shape == (16, 2, 256, 256) and torch.isfinite(loss)
    assert all(torch.isfinite(p.grad).all() for p in net.parameters() if p.grad is not None)
    sd = net.state_dict()
    assert int(sd["encoder.Conv1.conv.1.num_batches_tracked"]) == 2       # one support call (80 images) + one query call
    assert int(sd["cre.w_k.1.num_batches_tracked"]) == 5 + 5               # one CRE call per shot + T query calls
    torch.cuda.synchronize()


@pytest.mark.parametrize("fp16_planes", [False, True])
def test_odd_shapes_vs_oracle(fp16_planes):
    """Ragged everything: 96x96 images (24x24 feature maps: not powers of two -> the division paths of
    the wgrad strips), batch 3 (statistic groups of 3 images, M tails in every tile variant), T=2.  The default f16x2
    arithmetic keeps a call this small on three bf16 planes; fp16_planes forces its fp16 planes (train AND eval mode, whose
    scales are measured) through the same ragged shapes."""
    from oracle import rpnet_oracle as O
    from rpnet_amd import functional as RF
    from rpnet_amd import modules as RM
    if fp16_planes:
        RM._F16_MIN_PIXELS = 0          # restored by tests/conftest.py
    cfg = load_cfg(2)
    (si, fg, bg, qi, ql, appr), _ = episode_tensors(123, 3, 96, "cpu")
    P = O.seeded_params(requires_grad=True)
    ref = O.rp_net_forward(P, cfg, si, fg, bg, qi, appr, True)
    ref_loss = O.total_loss(ref, ql)
    ref_loss.backward()
    net = build(cfg, True)
    mv = lambda t: t.to(DEV)  # noqa: E731
    out = net([[mv(si[0][0])]], [[mv(fg[0][0])]], [[mv(bg[0][0])]], [mv(qi[0])], appr_query_labels=mv(appr))
    loss = total_loss(out, mv(ql), 1.0)
    loss.backward()
    for i in range(2):
        assert rel_err(out["refinement"][i], ref["refinement"][i]) < TOL
    assert rel_err(loss, ref_loss) < TOL
    for n, p in net.named_parameters():
        if p.grad is None or P[n].grad.norm() < 1e-4:
            continue
        a, b = p.grad.double().cpu().norm().item(), P[n].grad.double().norm().item()
        assert abs(a - b) < (1e-2 if n.startswith("encoder.") else 2e-3) * b, f"{n}: {a} vs {b}"
    for k in ("encoder.Conv2.conv.4.running_var", "cre.w_q.1.running_mean"):
        assert rel_err(net.state_dict()[k], P[k]) < 1e-4
    # eval mode on the same ragged shapes (running statistics as the train step above left them, on both sides)
    net.eval()
    RF.reset_arith()
    with torch.no_grad():
        ev = net([[mv(si[0][0])]], [[mv(fg[0][0])]], [[mv(bg[0][0])]], [mv(qi[0])], appr_query_labels=mv(appr))
        ref_ev = O.rp_net_forward(P, cfg, si, fg, bg, qi, appr, False)
    assert set(RF.arith_counts()["conv3x3"]) == {"f16x2" if fp16_planes else "bf16x3"}
    for i in range(2):
        assert rel_err(ev["refinement"][i], ref_ev["refinement"][i]) < TOL


def test_training_driver_learns_and_checkpoints(tmp_path):
    """train_rpnet.py (the driver the reference lacks): a few Adam steps on synthetic episodes lower the
    loss, and the checkpoint it writes loads back through the reference's checkpoint convention."""
    from train_rpnet import train
    cfg = load_cfg(2)
    torch.manual_seed(0)
    net, hist = train(cfg, steps=24, batch=4, size=64, dev=torch.device(DEV), lr=1e-3, log_every=0,
                      out_dir=str(tmp_path), steps_per_epoch=12)
    assert all(np.isfinite(hist))
    assert np.mean(hist[-6:]) < np.mean(hist[:6]) - 0.05, hist
    ck = torch.load(tmp_path / "002.ckpt", map_location="cpu")
    assert ck["epoch"] == 2 and len(ck["state_dict"]) == 147
    net2 = build(cfg, False)
    state = net2.state_dict(); state.update(ck["state_dict"]); net2.load_state_dict(state)   # test_rpnet.py:90-94
    assert torch.equal(net2.state_dict()["cre.q.0.weight"].cpu(), ck["state_dict"]["cre.q.0.weight"])


def test_gradient_fan_in_levels_agree():
    """The gradient fan-in forms (rpnet_amd.modules._FANIN: 0 = autograd's pairwise adds, 1 = RF.FanOut / RF.SplitRows, 2 = the
    default: both halves of the encoder output summed straight into their rows — RF.SplitFan — and the skip-connection gradients
    added inside the max-pool backward — RF.PoolSkip): the same step under all three.  Logits and BatchNorm buffers are equal bit
    for bit (the forward pass is the same launches); a sum of two gradients is the same fp32 number in either order, a sum of the 2 T
    gradients of the query features is not, and the encoder below amplifies that 1e-7 (2.6e-6 at Conv1.conv.0) — every parameter
    gradient within 2e-5 relative L2 of level 0's, those ABOVE the
    query-feature fan-in (nothing sums more than two terms below it) bit for bit between levels 1 and 2."""
    import rpnet_amd.modules as RM
    cfg = load_cfg(3)
    (si, fg, bg, qi, ql, appr), _ = episode_tensors(203, 4, 128, DEV)
    was = RM._FANIN
    res = []
    try:
        for level in (0, 1, 2):
            RM._FANIN = level
            net = build(cfg, True)
            out = net(si, fg, bg, qi, appr_query_labels=appr)
            total_loss(out, ql, 1.0).backward()
            torch.cuda.synchronize()
            res.append((out["output"].detach().clone(), {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None},
                        {n: b.clone() for n, b in net.named_buffers()}))
    finally:
        RM._FANIN = was
    for other in res[1:]:
        assert torch.equal(res[0][0], other[0])
        for n, b in res[0][2].items():
            assert torch.equal(b, other[2][n]), n
        assert set(other[1]) == set(res[0][1])
        for n, g in res[0][1].items():
            e = float((g.double() - other[1][n].double()).norm() / g.double().norm().clamp_min(1e-30))
            assert e <= 2e-5, (n, e)
    for n, g in res[1][1].items():
        if n.startswith("cre."):
            assert torch.equal(g, res[2][1][n]), n


def test_async_weight_gradients_match():
    """Opt-in async weight gradients (second HIP stream, accumulated straight into the flat bucket) give the
    same gradients as the autograd-returned ones, incl. the parameters used T+1 times (CRE)."""
    import rpnet_amd.functional as RF
    from rpnet_amd.parallel import FlatGradBucket
    cfg = load_cfg(3)
    (si, fg, bg, qi, ql, appr), _ = episode_tensors(202, 4, 128, DEV)
    grads = []
    for mode in (False, True):
        net = build(cfg, True)
        bucket = FlatGradBucket(net)
        bucket.zero()
        RF.set_async_wgrad(mode)
        try:
            out = net(si, fg, bg, qi, appr_query_labels=appr)
            total_loss(out, ql, 1.0).backward()
        finally:
            RF.set_async_wgrad(False)
        torch.cuda.synchronize()
        grads.append(bucket.flat.clone())
    assert torch.isfinite(grads[1]).all() and grads[0].abs().max() > 0
    assert rel_err(grads[1], grads[0]) < 1e-5
    # second step on the same bucket: zero() really clears what the side stream accumulated
    assert float((grads[1] - grads[0]).abs().max()) < 1e-5 * float(grads[0].abs().max())


@pytest.mark.parametrize("fp16_planes", [False, True])
def test_graphed_eval_matches_eager(fp16_planes):
    """hipGraph replay of the evaluation forward (batch 2, T = 10 like test_rpnet.py) is bit-identical to the
    eager call, also after the static input buffers are refilled with a different batch.  fp16_planes: the eval-mode
    fp16 path (measured tensor scales through the zeroed out_absmax slots, re-zeroed inside the captured graph)
    instead of the three bf16 planes a call this small gets by default."""
    from rpnet_amd.graph import GraphedEval
    from rpnet_amd import functional as RF
    from rpnet_amd import modules as RM
    if fp16_planes:
        RM._F16_MIN_PIXELS = 0          # restored by tests/conftest.py
    cfg = load_cfg(10)
    net = build(cfg, False)
    graphed = GraphedEval(net)
    for seed in (31, 32):
        (si, fg, bg, qi, ql, appr), _ = episode_tensors(seed, 2, 128, DEV)
        with torch.no_grad():
            ref = net(si, fg, bg, qi, appr_query_labels=appr)["output"].clone()
        RF.reset_arith()
        out = graphed(si, fg, bg, qi, appr_query_labels=appr)
        assert len(out["refinement"]) == 10
        # eager and replayed calls differ only in the power-of-two fp16 scales predicted from their own previous call
        # (RF.pred_*): exact scaling, identical roundings except below fp16's normal range (2^-25 of the tensor maximum)
        assert rel_err(out["output"], ref) <= 1e-6
    assert len(graphed._graphs) == 1
    with torch.no_grad():
        RF.reset_arith()
        net(si, fg, bg, qi, appr_query_labels=appr)
    assert set(RF.arith_counts()["conv3x3"]) == {"f16x2" if fp16_planes else "bf16x3"}


@pytest.mark.parametrize("tag", ["m64_train", "m128_train", "m256_train"])
def test_gradients_vs_fp64_yardstick(golden, tag, conv_math):
    _gradients_vs_fp64_yardstick(golden, tag, conv_math)


@pytest.mark.parametrize("tag", ["m64_5shot", "m64_2way"])
def test_extension_gradients_vs_fp64_yardstick(golden, tag, conv_math):
    """the same yardstick on the multi-shot / multi-way episodes of the composed-reference fixtures: EVERY parameter gradient
    of the extension rows' path (per-(way, shot) CRE calls, prototype means over shots / ways, the encoder's two calls as two
    chains) element-wise against the fp64 oracle — whose fp32 form tests/golden/gen_golden.py::gen_composed pins to the reference's own pieces"""
    _gradients_vs_fp64_yardstick(golden, tag, conv_math)


def _gradients_vs_fp64_yardstick(golden, tag, conv_math):
    """Backward parity with a reproducible yardstick instead of a loose constant.  Reference point: the CPU oracle in
    FLOAT64 on the same inputs.  Yardstick: how far the fp64 oracle's OWN gradients move when every image pixel is
    perturbed by YARD_EPS = 4e-7 relative (which moves the fp64 logits by 3e-6 .. 7e-6: the size of the HIP path's forward
    deviation from fp64 — both asserted below, so the yardstick is neither inflated nor starved) —
    the maximum over six independent draws (two at 256^2, the headline image size), per parameter tensor, relative L2.  Requirement: the HIP gradient is
    no further from the fp64 gradient than 3 x that movement.
    Why a perturbation and not "the fp32 oracle's distance to fp64": the distance is made of DISCRETE events — a
    pre-activation within the forward error of zero takes the other side of its ReLU (or max-pool / arg-max) — each
    worth ~ |dz| / ||dy|| = 1e-3 of every upstream gradient at these sizes (round 2's per-layer diagnostic showed one: the
    deviation enters at a single BatchNorm beta).  Whether an implementation hits such an event on a given episode is
    chance proportional to its forward error; the perturbation draws sample exactly that chance at the HIP path's error
    level (tests/test_oracle_conditioning.py is the CPU-only statement of the same sensitivity)."""
    g = golden(tag)
    size, B, T, _, seed, n_ways, n_shots = _meta(g)
    cfg = load_cfg(T)
    g64, l64, logits64, yard, fwd_moves = _yardstick(tag, g)
    net = build(cfg, True)
    (si, fg, bg, qi, ql, appr), _ = episode_tensors(seed, B, size, DEV, n_shots=n_shots, n_ways=n_ways)
    out = net(si, fg, bg, qi, appr_query_labels=appr)
    loss = total_loss(out, ql, cfg["align_loss_scaler"])
    loss.backward()
    assert abs(loss.item() - l64.item()) < 1e-5 * abs(l64.item())
    fwd_hip = rel_err(out["refinement"][0].detach(), logits64)
    # the perturbation is the right size: it moves the fp64 forward (median of the draws) between a third of and three
    # times what the HIP path deviates
    mid = sorted(fwd_moves)[len(fwd_moves) // 2]
    assert fwd_hip < 2e-5 and fwd_hip / 3.0 <= mid <= 3.0 * fwd_hip, (fwd_hip, fwd_moves)
    report = []
    for n, p in net.named_parameters():
        if p.grad is None:
            continue
        ref = g64[n]
        nrm = float(ref.norm())
        if nrm < 1e-4:              # conv biases in front of a train-mode BatchNorm: analytically zero
            assert float(p.grad.abs().max()) < 1e-4, n
            continue
        e_hip = float((p.grad.double().cpu() - ref).norm()) / nrm
        report.append((e_hip / yard[n], n, e_hip, yard[n]))
    report.sort(reverse=True)
    print(f"{tag} [{conv_math}]: forward deviation {fwd_hip:.1e} (perturbed fp64: {max(fwd_moves):.1e}); worst err_HIP / yardstick:",
          [(round(r, 2), n, f"{a:.1e}", f"{b:.1e}") for r, n, a, b in report[:3]], "median ratio", round(report[len(report) // 2][0], 2))
    # + YARD_FLOOR: where the function is smooth (the CRE's 1x1 block: yardstick = the perturbation's own 3e-6) what remains is
    # fp32 round-off of the reductions themselves — a BatchNorm beta gradient at 256^2 is a sum of B h w T = 41 000 fp32 terms,
    # sqrt(41 000) x 6e-8 = 1.2e-5 relative; measured 2.5e-5 on cre.q.1.bias with yardstick 5e-6 — not conditioning
    for ratio, n, e_hip, y in report:
        assert e_hip <= 3.0 * y + YARD_FLOOR, f"{n}: HIP {e_hip:.2e} from the fp64 gradient, yardstick (fp64 oracle under a {YARD_EPS} input perturbation) {y:.2e}"


@pytest.mark.parametrize("tag", ["cb", "up"])
def test_blocks_vs_reference_fixture(golden, tag, conv_math):
    """conv_block / up_conv (net/modules.py:42-75) against the REFERENCE's own outputs, gradients and BatchNorm buffers
    (tests/golden/blocks.npz: 3 -> 8 / 4 -> 8 channels, 10x12 images, two train calls + one eval call).  The HIP
    kernels tile channels by 32 / 64, so the fixture's layer is embedded in a 64 -> 64 one whose extra weights, biases
    and BatchNorm affines are zero: the real channels see exactly the reference's arithmetic."""
    from rpnet_amd.modules import conv_block, up_conv
    from rpnet_amd.utils.seeding import seeded_tensor
    g = golden("blocks")
    cin, cout = (3, 8) if tag == "cb" else (4, 8)
    m = (conv_block if tag == "cb" else up_conv)(64, 64, "BatchNorm2d")
    sd = m.state_dict()
    with torch.no_grad():
        for k, v in sd.items():
            ref = seeded_tensor(f"blk_{tag}.{k}", torch.empty(g[f"{tag}_sd2.{k}"].shape, dtype=v.dtype))
            if v.dim() == 0:
                v.copy_(ref)
                continue
            if "running_var" not in k:
                v.zero_()
            idx = tuple(slice(0, s) for s in ref.shape)
            v[idx] = ref
    m.to(DEV).train()

    def pad(x, c):
        out = torch.zeros(x.shape[0], c, *x.shape[2:])
        out[:, :x.shape[1]] = torch.as_tensor(x)
        return out.to(DEV)

    x1 = pad(g[f"{tag}_x1"], 64).requires_grad_(True)
    y1 = m(x1)
    assert rel_err(y1[:, :cout], g[f"{tag}_y1"]) < TOL and float(y1[:, cout:].abs().max()) == 0.0
    y1.backward(pad(g[f"{tag}_go"], 64))
    assert rel_err(x1.grad[:, :cin], g[f"{tag}_gx"]) < TOL
    for n, p in m.named_parameters():
        ref = torch.from_numpy(g[f"{tag}_g.{n}"])
        got = p.grad[tuple(slice(0, s) for s in ref.shape)]
        if n in ("conv.0.bias", "conv.3.bias", "up.1.bias"):
            # conv bias in front of a train-mode BatchNorm: round-off in the reference (1e-5), analytically zero here
            assert float(got.abs().max()) < 1e-5, n
        else:
            assert rel_err(got, ref) < TOL, n
    y2 = m(pad(g[f"{tag}_x2"], 64))
    assert rel_err(y2[:, :cout], g[f"{tag}_y2"]) < TOL
    for k, v in m.state_dict().items():
        if "running" in k:
            assert rel_err(v[:cout], g[f"{tag}_sd2.{k}"]) < 1e-4, k
        elif "num_batches" in k:
            assert int(v) == int(g[f"{tag}_sd2.{k}"]) == 2
    m.eval()
    with torch.no_grad():
        ye = m(pad(g[f"{tag}_x1"], 64))
    assert rel_err(ye[:, :cout], g[f"{tag}_yeval"]) < TOL


def test_eval_fp16_planes_on_predicted_scales():
    """Eval-mode calls (the reference's only entry point, test_rpnet.py:189-215) on fp16 planes whose tensor scales are
    PREDICTED from the previous call's measured maxima (RF.pred_*): the first call of a shape measures, later ones predict;
    results equal the three-bf16-plane arithmetic to fp32 round-off; an input whose activations outgrow the predicted bounds
    (images x 30) is detected, counted and redone on measured scales."""
    from rpnet_amd import functional as RF
    from rpnet_amd import modules as RM
    old, old_min = RF.conv_math(), RM._F16_MIN_PIXELS
    RM._F16_MIN_PIXELS = 0
    try:
        cfg = load_cfg(3)
        outs = {}
        for math in ("bf16x3", "f16x2"):
            RF.set_conv_math(math)
            net = build(cfg, False)
            base = RF.pred_stats()
            res = []
            for seed in (41, 42, 43):
                (si, fg, bg, qi, ql, appr), _ = episode_tensors(seed, 2, 128, DEV)
                with torch.no_grad():
                    RF.reset_arith()
                    res.append(net(si, fg, bg, qi, appr_query_labels=appr)["output"].clone())
                assert set(RF.arith_counts()["conv3x3"]) == {math}
            outs[math] = res
            if math == "f16x2":
                st = RF.pred_stats()
                assert st["calls"] - base["calls"] == 3 and st["predicted_calls"] - base["predicted_calls"] == 2, (base, st)
                assert st["violations"] == base["violations"]
                # activations 30 x larger than the history: beyond the safety factor of 4 -> violation -> redo, still right
                (si, fg, bg, qi, ql, appr), _ = episode_tensors(43, 2, 128, DEV)
                big = lambda t: [[30.0 * t[0][0]]]  # noqa: E731
                with torch.no_grad():
                    got = net(big(si), fg, bg, [30.0 * qi[0]], appr_query_labels=appr)["output"].clone()
                st2 = RF.pred_stats()
                assert st2["violations"] == st["violations"] + 1, (st, st2)
                RF.set_conv_math("bf16x3")
                ref_net = build(cfg, False)
                with torch.no_grad():
                    want = ref_net(big(si), fg, bg, [30.0 * qi[0]], appr_query_labels=appr)["output"]
                assert rel_err(got, want) < 1e-4
        for a, b in zip(outs["f16x2"], outs["bf16x3"]):
            assert rel_err(a, b) < 1e-4
    finally:
        RM._F16_MIN_PIXELS = old_min
        RF.set_conv_math(old)


def test_eval_split_k_matches_one_block_per_tile():
    """Eval-mode calls at batch 2: the 3x3 convolutions whose 256 x 64 tiles cover half of the CUs or fewer cut their K range
    into parts (rpnet_conv_desc.splitk_ws: fp32 partial tiles, summed in a fixed order by a second launch that does the
    epilogue — bias, folded BatchNorm, ReLU, max |output|, the fp16 planes on the predicted scale).  Same results as one
    block per tile up to the order of the fp32 sums, on measured (first call) and predicted (later calls) scales."""
    from rpnet_amd import functional as RF
    from rpnet_amd import modules as RM
    old, old_min, old_sk = RF.conv_math(), RM._F16_MIN_PIXELS, RF._EVAL_SPLITK
    RM._F16_MIN_PIXELS = 0
    RF.set_conv_math("f16x2")
    try:
        cfg = load_cfg(3)
        outs, used = {}, {}
        for on in (False, True):
            RF._EVAL_SPLITK = on
            net = build(cfg, False)
            lent = []
            orig = RF.call

            def spy(name, *args):
                if name == "rpnet_conv_fwd":
                    lent.append(int(bool(args[0]._obj.splitk_ws)))
                return orig(name, *args)

            RF.call = spy
            try:
                res = []
                for seed in (41, 42, 43):
                    (si, fg, bg, qi, ql, appr), _ = episode_tensors(seed, 2, 128, DEV)
                    with torch.no_grad():
                        o = net(si, fg, bg, qi, appr_query_labels=appr)
                    res.append([o["output"].clone()] + [o["refinement"][i].clone() for i in sorted(o["refinement"])])
            finally:
                RF.call = orig
            outs[on], used[on] = res, sum(lent)
        assert used[False] == 0 and used[True] >= 3 * 8, used       # 128^2 at batch 2: the CRE convolutions (M = 2048) and the deep levels
        for a, b in zip(outs[True], outs[False]):
            for x, y in zip(a, b):
                assert rel_err(x, y) < 2e-5
    finally:
        RM._F16_MIN_PIXELS, RF._EVAL_SPLITK = old_min, old_sk
        RF.set_conv_math(old)


@pytest.mark.parametrize("shots,ways,B,size", [(1, 1, 4, 128), (5, 1, 2, 64), (1, 2, 2, 128)])
def test_graphed_train_step_matches_eager(shots, ways, B, size):
    """rpnet_amd.graph.GraphedTrainStep: the whole training step (forward, harness loss, backward with the weight gradients on
    side streams, into the flat bucket) captured into a HIP graph — loss and every gradient bit-identical to the eager step,
    also on a second episode copied into the static inputs (the replay re-packs weights and re-measures nothing stale);
    1-way 1-shot and the two extension shapes of BASELINE configs[2] / configs[4] (5-shot, 2-way)."""
    import rpnet_amd.functional as RF
    from rpnet_amd.graph import GraphedTrainStep
    from rpnet_amd.parallel import FlatGradBucket
    cfg = load_cfg(2)
    net = build(cfg, True)
    bucket = FlatGradBucket(net)
    RF.set_async_wgrad(True)
    try:
        g = GraphedTrainStep(net, bucket, lambda out, ql: total_loss(out, ql, cfg["align_loss_scaler"]))
        for seed in (71, 72):
            (si, fg, bg, qi, ql, appr), _ = episode_tensors(seed, B, size, DEV, n_shots=shots, n_ways=ways)
            loss_g = g(si, fg, bg, qi, ql, appr).clone()
            torch.cuda.synchronize()
            grads_g = bucket.flat.clone()
            bucket.zero()
            out = net(si, fg, bg, qi, appr_query_labels=appr)
            loss_e = total_loss(out, ql, cfg["align_loss_scaler"])
            loss_e.backward()
            bucket.allreduce()
            torch.cuda.synchronize()
            assert torch.equal(loss_g, loss_e.detach()) and grads_g.abs().max() > 0
            assert torch.equal(grads_g, bucket.flat)
        assert len(g._graphs) == 1
    finally:
        RF.set_async_wgrad(False)


def test_two_models_with_their_own_schedules_in_one_process():
    """rpnet_amd.schedule: two RP_Net instances of ONE process on different arithmetic / stream layout / zero-tile
    skip — model A on the process defaults with async weight gradients into its bucket, model B with Schedule(conv_math="f32",
    async_wgrad=False, mask_skip=False, cre_streams_train=False, fanin=0) — forward and backward INTERLEAVED (A fwd, B fwd, A bwd,
    B bwd).  Each model's launches run on its own arithmetic (arith_counts), each model's result is bit-identical to that model run
    alone, and the process-wide defaults are what they were."""
    import rpnet_amd.functional as RF
    import rpnet_amd.modules as RM
    from rpnet_amd.parallel import FlatGradBucket
    from rpnet_amd.schedule import Schedule
    RM._F16_MIN_PIXELS = 0
    RF.set_conv_math("f16x2")
    RF.set_async_wgrad(False)
    cfg = load_cfg(2)
    (si, fg, bg, qi, ql, appr), _ = episode_tensors(77, 2, 64, DEV)

    def make(kind):
        net = build(cfg, True)
        if kind == "A":
            net.schedule.async_wgrad = True
            return net, FlatGradBucket(net)
        net.schedule = Schedule(conv_math="f32", async_wgrad=False, mask_skip=False, cre_streams_train=False, fanin=0)
        for m in (net.cre, net.encoder):
            object.__setattr__(m, "schedule", net.schedule)
        return net, None

    def grads(net, bucket):
        if bucket is not None:
            bucket.allreduce()
        torch.cuda.synchronize()
        return {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}

    def alone(kind):
        net, bucket = make(kind)
        if bucket is not None:
            bucket.zero()
        out = net(si, fg, bg, qi, appr_query_labels=appr)
        total_loss(out, ql, 1.0).backward()
        return out["output"].detach().clone(), grads(net, bucket)

    ref_a, ref_b = alone("A"), alone("B")
    (na, ba), (nb, _) = make("A"), make("B")
    ba.zero()
    base = (RF.conv_math(), RF._ASYNC["on"], RF._MASK_SKIP)
    RF.reset_arith()
    oa = na(si, fg, bg, qi, appr_query_labels=appr)
    ca = RF.arith_counts()
    RF.reset_arith()
    ob = nb(si, fg, bg, qi, appr_query_labels=appr)
    cb = RF.arith_counts()
    assert (RF.conv_math(), RF._ASYNC["on"], RF._MASK_SKIP) == base
    assert set(ca["conv3x3"]) == {"f16x2"} and set(cb["conv3x3"]) == {"f32"}, (ca, cb)
    RF.reset_arith()
    total_loss(oa, ql, 1.0).backward()          # A's backward after B's forward: its nodes carry A's options
    wa = RF.arith_counts()
    RF.reset_arith()
    total_loss(ob, ql, 1.0).backward()
    wb = RF.arith_counts()
    assert set(wa["wgrad3x3"]) == {"f16x2"} and set(wb["wgrad3x3"]) == {"f32"}, (wa, wb)
    ga, gb = grads(na, ba), grads(nb, None)
    assert torch.equal(oa["output"], ref_a[0]) and torch.equal(ob["output"], ref_b[0])
    assert ga.keys() == ref_a[1].keys() and all(torch.equal(ga[k], ref_a[1][k]) for k in ga)
    assert gb.keys() == ref_b[1].keys() and all(torch.equal(gb[k], ref_b[1][k]) for k in gb)
    # the two arithmetics agree to the bar of the fixtures (they are both fp32-equivalent)
    assert rel_err(oa["output"], ob["output"]) < TOL
