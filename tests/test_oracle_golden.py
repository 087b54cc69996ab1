"""CPU: the oracle restatement against the committed golden vectors (which
tests/golden/gen_golden.py produced from the reference itself)."""
import numpy as np
import pytest
import torch

from oracle import rpnet_oracle as O
from tests.helpers import episode_tensors, in_checksum, load_cfg, rel_err, rnd

TOL = 1e-4  # fp32, different summation orders


def test_param_inventory():
    shapes = O.param_shapes()
    assert len(shapes) == 147
    n = sum(int(np.prod(s)) for k, s in shapes.items() if not k.endswith(("running_mean", "running_var", "num_batches_tracked")))
    assert n == 34972800


def test_correlation_known_answers(golden):
    g = golden("ops")
    for tag in ("corr_small", "corr_r5"):
        b, c, h, w, r = g[tag + "_dims"]
        f1 = rnd(11, b, c, h, w).requires_grad_(True)
        f2 = rnd(12, b, c, h, w).requires_grad_(True)
        go = rnd(13, b, (2 * r + 1) ** 2, h, w)
        for fn in (O.local_correlation, O.correlation_as_written):
            out = fn(f1, f2, int(r))
            g1, g2 = torch.autograd.grad(out, [f1, f2], go)
            assert rel_err(out, g[tag + "_out"]) < 1e-5
            assert rel_err(g1, g[tag + "_g1"]) < 1e-5 and rel_err(g2, g[tag + "_g2"]) < 1e-5


def test_matcher_known_answers(golden):
    g = golden("ops")
    fts, masks = torch.from_numpy(g["gf_fts"]), torch.from_numpy(g["gf_masks"])
    for fn in (O.get_features_as_written, O.get_features_adjoint):
        out = torch.cat([fn(fts, m) for m in masks], 0)
        assert rel_err(out, g["gf_out"]) < 1e-5
        assert out[2].abs().max() == 0                          # empty mask -> 0 / 1e-5
    gf = [O.get_features_adjoint(fts, m) for m in masks]
    fg, bg = O.get_prototype([[gf[0], gf[1]]], [[gf[1], gf[2]]])
    assert rel_err(fg[0], g["proto_fg"]) < 1e-5 and rel_err(bg, g["proto_bg"]) < 1e-5
    q = torch.from_numpy(g["cd_q"])
    assert rel_err(O.cal_dist(q, fg[0]), g["cd_fg"]) < 1e-5
    assert O.cal_dist(q, torch.zeros(1, 8)).abs().max() == 0   # zero prototype
    assert O.cal_dist(q, fg[0])[0, 0, 0] == 0                   # zero feature vector


def test_losses_known_answers(golden):
    g = golden("ops")
    lg = torch.from_numpy(g["dce_logits"]).requires_grad_(True)
    lab = torch.from_numpy(g["dce_labels"])
    loss = O.dice_ce(lg, lab)
    assert rel_err(loss, g["dce_loss"]) < 1e-6
    assert rel_err(torch.autograd.grad(loss, lg)[0], g["dce_grad"]) < 1e-5
    qf = torch.from_numpy(g["al_qf"]).requires_grad_(True)
    sf = torch.from_numpy(g["al_sf"]).requires_grad_(True)
    fm = torch.from_numpy(g["al_fm"])
    al = O.align_loss(qf, torch.from_numpy(g["al_pred"]), sf, fm, 1 - fm)
    gq, gs = torch.autograd.grad(al, [qf, sf])
    assert rel_err(al, g["al_loss"]) < 1e-6 and rel_err(gq, g["al_gq"]) < 1e-5 and rel_err(gs, g["al_gs"]) < 1e-5


@pytest.mark.parametrize("tag,fn", [("cb", O.conv_block), ("up", O.up_conv)])
def test_blocks(golden, tag, fn):
    from rpnet_amd.utils.seeding import seeded_tensor
    g = golden("blocks")
    pre = f"blk_{tag}"
    P = {f"{pre}.{k[len(tag) + 5:]}": None for k in g if k.startswith(f"{tag}_sd2.")}
    for k in g:
        if k.startswith(f"{tag}_sd2."):
            name = f"{pre}.{k[len(tag) + 5:]}"
            P[name] = seeded_tensor(name, torch.from_numpy(np.asarray(g[k])))
    x1 = torch.from_numpy(g[f"{tag}_x1"]).requires_grad_(True)
    for n in P:
        if P[n].is_floating_point() and "running" not in n:
            P[n].requires_grad_(True)
    y1 = fn(P, pre, x1, True)
    names = [k[len(tag) + 3:] for k in g if k.startswith(f"{tag}_g.")]
    grads = torch.autograd.grad(y1, [x1] + [P[f"{pre}.{n}"] for n in names], torch.from_numpy(g[f"{tag}_go"]))
    y2 = fn(P, pre, torch.from_numpy(g[f"{tag}_x2"]), True)
    assert rel_err(y1, g[f"{tag}_y1"]) < 1e-5 and rel_err(y2, g[f"{tag}_y2"]) < 1e-5
    assert rel_err(grads[0], g[f"{tag}_gx"]) < 1e-4
    for n, gr in zip(names, grads[1:]):
        ref = torch.from_numpy(g[f"{tag}_g.{n}"])
        assert (gr - ref).norm() < 1e-4 * ref.norm() + 1e-5, n
    for k in g:
        if k.startswith(f"{tag}_sd2."):
            assert rel_err(P[f"{pre}.{k[len(tag) + 5:]}"], g[k]) < 1e-5, k     # stats after TWO train calls
    assert rel_err(fn(P, pre, x1, False), g[f"{tag}_yeval"]) < 1e-5


@pytest.mark.parametrize("tag", ["m64_train", "m64_eval", "m128_train", "m64_train_x", "m64_train_x2", "m64_train_x3"])
def test_model_vs_golden(golden, tag):
    g = golden(tag)
    size, B, T, training, seed = (int(v) for v in g["meta"])
    cfg = load_cfg(T)
    if tag.count("_") == 2:                     # m64_train_x2: mask_feature_map = 'x2' (net/unet.py:401-414,437-449)
        cfg["mask_feature_map"] = tag.rsplit("_", 1)[1]
    (si, fg, bg, qi, ql, appr), ep = episode_tensors(seed, B, size)
    assert np.allclose(in_checksum(ep), g["in_checksum"], rtol=0, atol=1e-6), "synthetic inputs drifted"
    s_out, s_d4, s_f = (int(v) for v in g["strides"])
    P = O.seeded_params(cfg["mask_refinement_correlation_radius"], requires_grad=bool(training),
                        mask_feature_map=cfg["mask_feature_map"])
    taps = {}
    with torch.set_grad_enabled(bool(training)):
        out = O.rp_net_forward(P, cfg, si, fg, bg, qi, appr, bool(training), taps=taps)
        loss = O.total_loss(out, ql, cfg["align_loss_scaler"])
    assert rel_err(taps["qry_d4"][:, ::s_d4], g["qry_d4"]) < TOL
    assert rel_err(taps["supp_fts"][0, 0][..., ::s_f, ::s_f], g["supp_fts"]) < TOL
    assert rel_err(taps["protos"], g["protos"]) < TOL
    for i in range(T):
        assert rel_err(out["refinement"][i][..., ::s_out, ::s_out], g[f"refinement_{i}"]) < TOL
        assert rel_err(taps[f"inter_{i}"][0][..., ::s_f, ::s_f], g[f"inter_{i}"]) < TOL
    assert rel_err(out["output"][..., ::s_out, ::s_out], g["output"]) < TOL
    assert rel_err(loss, g["loss"]) < 1e-5
    if training:
        assert rel_err(torch.as_tensor(out["align_loss"]), g["align_loss"]) < 1e-5
        loss.backward()
        for n, ref in zip(g["grad_names"], g["grad_norms"]):
            gr = P[str(n)].grad
            if str(n) in set(str(u) for u in g["unused"]):
                assert gr is None                               # cre.w_context / cre.out never used
                continue
            if ref > 1e-4:
                assert abs(gr.double().norm().item() - ref) < 2e-3 * ref, n
        for k in g:
            if k.startswith("sd."):
                assert rel_err(P[k[3:]], g[k]) < 1e-5, k


@pytest.mark.parametrize("tag", ["m64_5shot", "m64_2way", "m64_2way2shot"])
def test_extension_rows_vs_composed_reference(golden, tag):
    """SURVEY.md §8c fixture set 4: the oracle's multi-shot / multi-way branch (rp_net_forward, the per-(way, shot) CRE
    calls) against the fixtures tests/golden/gen_golden.py::gen_composed wrote by calling the REFERENCE's own encoder, cre,
    getFeatures, getPrototype, calDist and alignLoss per (way, shot) — stage taps, logits, align loss, loss, every gradient
    norm, BatchNorm buffers (whose update order the way-major / shot-minor CRE calls fix)."""
    g = golden(tag)
    size, B, T, _, seed, n_ways, n_shots = (int(v) for v in g["meta"])
    cfg = load_cfg(T)
    (si, fg, bg, qi, ql, appr), ep = episode_tensors(seed, B, size, n_shots=n_shots, n_ways=n_ways)
    assert np.allclose(in_checksum(ep), g["in_checksum"], rtol=0, atol=1e-6), "synthetic inputs drifted"
    s_d4 = int(g["strides"][1])
    P = O.seeded_params(cfg["mask_refinement_correlation_radius"], requires_grad=True)
    taps = {}
    out = O.rp_net_forward(P, cfg, si, fg, bg, qi, appr, True, taps=taps)
    loss = O.total_loss(out, ql, cfg["align_loss_scaler"])
    assert out["output"].shape == (B, 1 + n_ways, size, size)
    assert rel_err(taps["supp_d4"][:, :, :, ::s_d4], g["supp_d4"]) < TOL and rel_err(taps["qry_d4"][:, ::s_d4], g["qry_d4"]) < TOL
    assert tuple(taps["supp_fts"].shape[:3]) == (n_ways, n_shots, B) and rel_err(taps["supp_fts"], g["supp_fts"]) < TOL
    assert rel_err(taps["protos"], g["protos"]) < TOL
    for i in range(T):
        assert rel_err(taps[f"inter_{i}"][0], g[f"inter_{i}"]) < TOL
        assert rel_err(out["refinement"][i], g[f"refinement_{i}"]) < TOL
    assert rel_err(loss, g["loss"]) < 1e-5 and rel_err(torch.as_tensor(out["align_loss"]), g["align_loss"]) < 1e-5
    loss.backward()
    unused = set(str(u) for u in g["unused"])
    for n, ref in zip(g["grad_names"], g["grad_norms"]):
        gr = P[str(n)].grad
        if str(n) in unused:
            assert gr is None
        elif ref > 1e-4:
            assert abs(gr.double().norm().item() - ref) < 2e-3 * ref, n
    for k in g:
        if k.startswith("sd."):
            assert rel_err(P[k[3:]], g[k]) < 1e-5, k
    assert int(P["cre.w_k.1.num_batches_tracked"]) == n_ways * n_shots + T and int(P["encoder.Conv1.conv.1.num_batches_tracked"]) == 2


def test_teacher_forced_iteration(golden):
    """Feeding the reference's own masks reproduces each iteration independently."""
    g = golden("m64_train")
    size, B, T, training, seed = (int(v) for v in g["meta"])
    cfg = load_cfg(T)
    (si, fg, bg, qi, ql, appr), _ = episode_tensors(seed, B, size)
    P = O.seeded_params(cfg["mask_refinement_correlation_radius"])
    forced = {i + 1: torch.from_numpy(g[f"next_mask_{i}"]) for i in range(T - 1)}
    with torch.no_grad():
        out = O.rp_net_forward(P, cfg, si, fg, bg, qi, appr, True, align=False, forced_masks=forced)
    for i in range(T):
        assert rel_err(out["refinement"][i], g[f"refinement_{i}"]) < TOL


def test_vgg_encoder_vs_golden(golden):
    """oracle restatement of vgg.Encoder against the reference's own output (name-seeded weights)."""
    from rpnet_amd.modules import Encoder
    from rpnet_amd.utils.seeding import seeded_tensor
    g = golden("vgg")
    enc = Encoder(3, None)
    P = {f"vgg.{k}": seeded_tensor(f"vgg.{k}", v) for k, v in enc.state_dict().items()}
    with torch.no_grad():
        y = O.vgg_encoder(P, torch.from_numpy(g["x"]))
    assert y.shape == (1, 512, 8, 8) and rel_err(y, g["y"]) < 1e-5
