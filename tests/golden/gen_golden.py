#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REFERENCE itself.

Runs ONLY in the build container (needs /root/reference; never on the GPU box).
It imports uci-cbcl/RP-Net with the absent third-party modules stubbed (none is
touched on the UNet hot path), fills the model with the name-seeded
parameters of rpnet_amd.utils.seeding, runs it on seeded synthetic episodes and
stores inputs-by-seed + expected outputs as small .npz files.  While doing so it
also checks oracle/rpnet_oracle.py against the reference op by op and end to end
and fails loudly on a mismatch, so a committed fixture set implies a pinned oracle.

    python tests/golden/gen_golden.py          # rewrites tests/golden/*.npz
"""
import os
import sys
import unittest.mock as mock

sys.dont_write_bytecode = True
for m in ["torchvision", "torchvision.models", "torchvision.models.resnet", "pydicom", "SimpleITK",
          "cv2", "skimage", "skimage.measure", "nrrd", "nibabel", "torchviz"]:
    sys.modules[m] = mock.MagicMock()
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

import importlib.util  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402
import yaml  # noqa: E402


def _load_reference():
    """Import /root/reference/net as package `refnet` without shadowing the repo's own net/."""
    import types
    pkg = types.ModuleType("refnet")
    pkg.__path__ = ["/root/reference/net"]
    sys.modules["refnet"] = pkg
    # utils.util is a dead import of net/unet.py:11 -> stub it too (drags in pydicom etc.)
    sys.modules.setdefault("utils", mock.MagicMock())
    sys.modules.setdefault("utils.util", mock.MagicMock())
    mods = {}
    for name in ["vgg", "modules", "unet", "rp_net"]:
        spec = importlib.util.spec_from_file_location(f"refnet.{name}", f"/root/reference/net/{name}.py",
                                                      submodule_search_locations=None)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[f"refnet.{name}"] = mod
        spec.loader.exec_module(mod)
        mods[name] = mod
    return mods


REF = _load_reference()
from oracle import rpnet_oracle as O  # noqa: E402
from rpnet_amd.utils.seeding import seed_module_  # noqa: E402
from rpnet_amd.utils.synth import make_episode  # noqa: E402

torch.set_num_threads(8)
CFG = yaml.load(open("/root/reference/yamls/example.yml"), Loader=yaml.FullLoader)


def close(a, b, tol, what):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    err = (a - b).abs().max().item()
    ref = b.abs().max().item() + 1e-12
    assert err <= tol * max(ref, 1.0), f"oracle != reference at {what}: max-abs {err:.3e} (ref scale {ref:.3e})"
    return err


def save(name, **arrs):
    out = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()}
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"  wrote {name}.npz  ({os.path.getsize(os.path.join(HERE, name + '.npz')) / 1e6:.2f} MB)")


def rnd(seed, *shape):
    return torch.from_numpy(np.random.RandomState(seed).standard_normal(shape).astype(np.float32))


# ------------------------------------------------------------------------ op level
def gen_ops():
    R = REF["rp_net"]
    fx = {}
    # Correlation known answers (+grads)
    for tag, (b, c, h, w, r) in {"corr_small": (2, 16, 12, 10, 3), "corr_r5": (1, 64, 16, 16, 5)}.items():
        f1 = rnd(11, b, c, h, w).requires_grad_(True)
        f2 = rnd(12, b, c, h, w).requires_grad_(True)
        go = rnd(13, b, (2 * r + 1) ** 2, h, w)
        ref = R.Correlation(f1, f2, r=r)
        g1, g2 = torch.autograd.grad(ref, [f1, f2], go)
        o1 = O.local_correlation(f1, f2, r)
        og1, og2 = torch.autograd.grad(o1, [f1, f2], go)
        close(o1, ref, 2e-6, tag); close(og1, g1, 2e-6, tag + ".g1"); close(og2, g2, 2e-6, tag + ".g2")
        close(O.correlation_as_written(f1, f2, r), ref, 1e-6, tag + ".as_written")
        fx.update({f"{tag}_dims": np.array([b, c, h, w, r]), f"{tag}_out": ref, f"{tag}_g1": g1, f"{tag}_g2": g2})
    # getFeatures / getPrototype / calDist incl. empty mask and zero vector
    net = R.RP_Net.__new__(R.RP_Net)
    fts = rnd(21, 1, 8, 6, 5)
    masks = (rnd(22, 3, 1, 24, 20) > 0.3).float()
    masks[2] = 0  # empty mask
    gf = [net.getFeatures(fts, m) for m in masks]
    for i, m in enumerate(masks):
        close(O.get_features_as_written(fts, m), gf[i], 1e-6, f"getFeatures{i}")
        close(O.get_features_adjoint(fts, m), gf[i], 2e-6, f"getFeatures_adjoint{i}")
    fgp, bgp = net.getPrototype([[gf[0], gf[1]]], [[gf[1], gf[2]]])
    q = rnd(23, 1, 8, 6, 5)
    q[:, :, 0, 0] = 0  # zero vector -> cosine 0
    d_fg = net.calDist(q, fgp[0]); d_bg = net.calDist(q, bgp)
    d_zero = net.calDist(q, torch.zeros(1, 8))
    close(O.cal_dist(q, fgp[0]), d_fg, 1e-6, "calDist")
    fx.update({"gf_fts": fts, "gf_masks": masks, "gf_out": torch.cat(gf, 0), "proto_fg": fgp[0], "proto_bg": bgp,
               "cd_q": q, "cd_fg": d_fg, "cd_bg": d_bg, "cd_zero": d_zero})
    # dice_ce
    lg = rnd(31, 2, 2, 16, 12).requires_grad_(True)
    lab = (rnd(32, 2, 16, 12) > 0.5).long()
    l_ref = R.dice_ce(lg, lab)
    (gl,) = torch.autograd.grad(l_ref, lg)
    l_o = O.dice_ce(lg, lab)
    close(l_o, l_ref, 1e-6, "dice_ce"); close(torch.autograd.grad(l_o, lg)[0], gl, 1e-6, "dice_ce.grad")
    fx.update({"dce_logits": lg, "dce_labels": lab, "dce_loss": l_ref, "dce_grad": gl})
    # alignLoss: normal + skip-way (prediction all background)
    qf = rnd(41, 1, 8, 6, 5).requires_grad_(True)
    sf = rnd(42, 1, 1, 8, 6, 5).requires_grad_(True)
    fm = (rnd(43, 1, 1, 24, 20) > 0.2).float()
    pred = rnd(44, 1, 2, 6, 5)
    al = net.alignLoss(qf, pred, sf, fm, 1 - fm)
    gq, gs = torch.autograd.grad(al, [qf, sf])
    close(O.align_loss(qf, pred, sf, fm, 1 - fm), al, 1e-6, "alignLoss")
    pred_bg = pred.clone(); pred_bg[:, 0] = 10; pred_bg[:, 1] = -10
    al_skip = net.alignLoss(qf, pred_bg, sf, fm, 1 - fm)
    assert al_skip == 0 and O.align_loss(qf, pred_bg, sf, fm, 1 - fm) == 0
    fx.update({"al_qf": qf, "al_sf": sf, "al_fm": fm, "al_pred": pred, "al_loss": al, "al_gq": gq, "al_gs": gs})
    save("ops", **fx)


def gen_blocks():
    """conv_block / up_conv at small channel counts, train & eval, BN running stats
    after one and two calls (net/modules.py:42-75)."""
    M = REF["modules"]
    fx = {}
    for tag, mod, cin, cout in [("cb", M.conv_block, 3, 8), ("up", M.up_conv, 4, 8)]:
        m = mod(cin, cout, "BatchNorm2d")
        sub = "conv" if tag == "cb" else "up"
        pre = f"blk_{tag}"
        sd = {f"{pre}.{k}": v for k, v in m.state_dict().items()}
        from rpnet_amd.utils.seeding import seed_state_dict
        m.load_state_dict({k[len(pre) + 1:]: v for k, v in seed_state_dict(sd).items()})
        P = {k: v.clone() for k, v in seed_state_dict(sd).items()}
        x1 = rnd(51, 2, cin, 10, 12).requires_grad_(True)
        x2 = rnd(52, 2, cin, 10, 12)
        m.train()
        y1 = m(x1)
        go = rnd(53, *y1.shape)
        grads = torch.autograd.grad(y1, [x1] + list(m.parameters()), go)
        y2 = m(x2)
        f = O.conv_block if tag == "cb" else O.up_conv
        oy1 = f(P, pre, x1, True); oy2 = f(P, pre, x2, True)
        close(oy1, y1, 1e-5, pre + ".y1"); close(oy2, y2, 1e-5, pre + ".y2")
        for k, v in m.state_dict().items():
            close(P[f"{pre}.{k}"], v, 1e-6, f"{pre}.{k}")
        m.eval()
        ye = m(x1)
        close(f(P, pre, x1, False), ye, 1e-5, pre + ".eval")
        fx.update({f"{tag}_x1": x1, f"{tag}_x2": x2, f"{tag}_go": go, f"{tag}_y1": y1, f"{tag}_y2": y2,
                   f"{tag}_yeval": ye, f"{tag}_gx": grads[0]})
        for (n, _), g in zip(m.named_parameters(), grads[1:]):
            fx[f"{tag}_g.{n}"] = g
        for k, v in m.state_dict().items():
            fx[f"{tag}_sd2.{k}"] = v
    save("blocks", **fx)


# --------------------------------------------------------------------- model level
def build_ref(cfg):
    net = REF["rp_net"].RP_Net(cfg={"align": True, "backbone": "UNet"}, backbone_cfg=cfg)
    seed_module_(net)
    return net


def to_t(ep):
    t = lambda a: torch.from_numpy(a)  # noqa: E731
    return ([[t(s) for s in way] for way in ep["support_images"]], [[t(s) for s in way] for way in ep["support_fg"]],
            [[t(s) for s in way] for way in ep["support_bg"]], [t(ep["query_images"])], t(ep["query_labels"]),
            t(ep["appr_query_labels"]))


def gen_model(tag, size, B, T, training, seed, stride=1, d4_stride=1, fts_stride=1, mask_feature_map=None):
    cfg = dict(CFG)
    cfg["n_iter_refinement"] = T
    if mask_feature_map is not None:      # net/unet.py:401-414,437-449: the mask as one more input channel of Conv1 / 2 / 3
        cfg["mask_feature_map"] = mask_feature_map
    net = build_ref(cfg)
    net.train(training)
    ep = make_episode(seed, B, size)
    si, fg, bg, qi, ql, appr = to_t(ep)
    caps = {"enc": [], "cre": []}
    h1 = net.encoder.register_forward_hook(lambda m, i, o: caps["enc"].append(o["d4"]))
    h2 = net.cre.register_forward_hook(lambda m, i, o: caps["cre"].append(o))
    with torch.set_grad_enabled(training):
        out = net(si, fg, bg, qi, appr_query_labels=appr)
        loss = O.total_loss(out, ql, cfg["align_loss_scaler"])  # harness objective on reference outputs
    h1.remove(); h2.remove()
    assert torch.equal(out["output"], out["refinement"][T - 1])  # final pass is recomputation
    fx = {"meta": np.array([size, B, T, int(training), seed]),
          "in_checksum": np.array([float(ep["query_images"].astype(np.float64).sum()),
                                   float(ep["support_images"][0][0].astype(np.float64).sum()),
                                   float(ep["appr_query_labels"].sum()), float(ep["support_fg"][0][0].sum())]),
          "output": out["output"][..., ::stride, ::stride], "loss": loss,
          "align_loss": torch.as_tensor(out["align_loss"]).float(),
          "supp_d4": caps["enc"][0][:, ::d4_stride], "qry_d4": caps["enc"][1][:, ::d4_stride],
          "supp_fts": caps["cre"][0][..., ::fts_stride, ::fts_stride],
          "strides": np.array([stride, d4_stride, fts_stride])}
    for i in range(T):
        fx[f"refinement_{i}"] = out["refinement"][i][..., ::stride, ::stride]
        fx[f"inter_{i}"] = caps["cre"][1 + i][..., ::fts_stride, ::fts_stride]
        p = out["refinement"][i].softmax(1)[:, 1]
        fx[f"fg_frac_{i}"] = (p > 0.5).float().mean()
        fx[f"next_mask_{i}"] = torch.nn.functional.avg_pool2d((p > 0.5).float().unsqueeze(1), 4)
        pred = (p > 0.5).long()
        fx[f"dice_{i}"] = 2.0 * (pred * ql).sum() / (pred.sum() + ql.sum() + 1e-7)
    # prototypes via the reference's own getFeatures/getPrototype
    protos = []
    for e in range(B):
        f = net.getFeatures(caps["cre"][0][[e]], fg[0][0][[e]]); b_ = net.getFeatures(caps["cre"][0][[e]], bg[0][0][[e]])
        fgp, bgp = net.getPrototype([[f]], [[b_]])
        protos.append(torch.cat([bgp, fgp[0]], 0))
    fx["protos"] = torch.stack(protos, 0)
    if training:
        loss.backward()
        names, norms, heads = [], [], []
        for n, p in net.named_parameters():
            names.append(n)
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            norms.append(g.double().norm().item())
            heads.append(torch.nn.functional.pad(g.flatten()[:32], (0, max(0, 32 - g.numel()))))
        fx["grad_names"] = np.array(names)
        fx["grad_norms"] = np.array(norms)
        fx["grad_heads"] = torch.stack(heads, 0)
        fx["unused"] = np.array([n for n, p in net.named_parameters() if p.grad is None])
        for k, v in net.state_dict().items():
            if "running" in k or "num_batches" in k:
                fx["sd." + k] = v
    # ---- pin the oracle end to end (both modes) on this very case
    for as_written in (True, False):
        P = O.seeded_params(cfg["mask_refinement_correlation_radius"], requires_grad=training,
                            mask_feature_map=cfg["mask_feature_map"])
        taps = {}
        with torch.set_grad_enabled(training):
            o = O.rp_net_forward(P, cfg, si, fg, bg, qi, appr, training, align=True, as_written=as_written, taps=taps)
            ol = O.total_loss(o, ql, cfg["align_loss_scaler"])
        w = f"{tag}[as_written={as_written}]"
        close(taps["supp_d4"].reshape(caps["enc"][0].shape), caps["enc"][0], 1e-4, w + ".supp_d4")
        close(taps["qry_d4"], caps["enc"][1], 1e-4, w + ".qry_d4")
        close(taps["supp_fts"][0, 0], caps["cre"][0], 1e-4, w + ".supp_fts")
        flips = 0
        for i in range(T):
            flips += ((o["refinement"][i].softmax(1)[:, 1] > 0.5) != (out["refinement"][i].softmax(1)[:, 1] > 0.5)).sum().item()
            close(o["refinement"][i], out["refinement"][i], 1e-4, w + f".refinement[{i}]")
        close(o["output"], out["output"], 1e-4, w + ".output")
        close(ol, loss, 1e-5, w + ".loss")
        if training:
            ol.backward()
            for n, p in net.named_parameters():
                if p.grad is None:
                    assert P[n].grad is None, n
                    continue
                # conv biases in front of a train-mode BatchNorm have an analytically ZERO
                # gradient (the reference's values there are ~1e-7 round-off): absolute floor.
                e = (P[n].grad - p.grad).double().norm().item()
                sib = dict(net.named_parameters()).get(n.rsplit(".", 1)[0] + ".weight")
                floor = 1e-5 * max(1.0, sib.grad.double().norm().item() if sib is not None and sib.grad is not None else 1.0)
                assert e < 2e-3 * p.grad.double().norm().item() + floor, f"{w} grad {n}: abs {e:.2e}"
            for k, v in net.state_dict().items():
                if "running" in k or "num_batches" in k:
                    close(P[k], v, 1e-5, w + "." + k)
        print(f"  oracle == reference on {w} (threshold flips: {flips})")
    save(tag, **fx)


# ------------------------------------------------- extension rows: multi-shot / multi-way, composed from the reference's pieces
def composed_forward(net, cfg, si, fg, bg, qi, appr, caps):
    """SURVEY.md §8a "Extension rows": the reference's forward (net/rp_net.py:226-350) fails for n_shots > 1 or n_ways > 1
    (:275 feeds shot [0][0] only, :288 then indexes a [1, 1, ...] tensor), so the expected behaviour is DEFINED as the
    composition of the reference's OWN modules and methods — net.encoder, net.cre, net.getFeatures, net.getPrototype,
    net.calDist, net.alignLoss, all called here on the reference object — with the one generalisation of :271-275: the CRE
    runs once per (way, shot) on that shot's features with that shot's own pooled foreground mask, way-major / shot-minor
    (which fixes the order of the BatchNorm running-statistic updates).  Everything else is the reference's line, cited."""
    import torch.nn.functional as F
    n_ways, n_shots, n_queries = len(si), len(si[0]), len(qi)
    B = si[0][0].shape[0]
    img_size = qi[0].shape[-2:]
    imgs = torch.cat([torch.cat(way, dim=0) for way in si], dim=0)                      # :245
    img_fts = net.encoder(imgs, fg[0][0].unsqueeze(1))["d4"]                            # :248-249
    fts_size = img_fts.shape[-2:]
    supp_d4 = img_fts.view(n_ways, n_shots, B, -1, *fts_size)                           # :252
    qry_d4 = net.encoder(torch.cat(qi, dim=0), fg[0][0].unsqueeze(1))["d4"]             # :254-258
    qry_fts = qry_d4.view(n_queries, B, -1, *fts_size)                                  # :262
    fore = torch.stack([torch.stack(way, dim=0) for way in fg], dim=0)                  # :264-265
    back = torch.stack([torch.stack(way, dim=0) for way in bg], dim=0)                  # :266-267
    qry_mask = F.avg_pool2d(appr.unsqueeze(1), net.scale)                               # :269-270
    rows = []
    for wa in range(n_ways):                                                            # :271-275, per (way, shot)
        row = []
        for s in range(n_shots):
            sm = F.avg_pool2d(fore[wa][s].unsqueeze(1), net.scale)
            row.append(net.cre(supp_d4[wa][s] * sm, supp_d4[wa][s] * (1 - sm)))
        rows.append(torch.stack(row, 0))
    supp_fts = torch.stack(rows, 0)                                                     # Wa x Sh x B x C x h x w
    caps.update(supp_d4=supp_d4, qry_d4=qry_d4, supp_fts=supp_fts, inter=[], protos=None)

    def match(inter):
        outs, preds, protos = [], [], []
        for epi in range(B):
            fgf = [[net.getFeatures(supp_fts[wa, s, [epi]], fore[wa, s, [epi]]) for s in range(n_shots)]
                   for wa in range(n_ways)]                                             # :288-290
            bgf = [[net.getFeatures(supp_fts[wa, s, [epi]], back[wa, s, [epi]]) for s in range(n_shots)]
                   for wa in range(n_ways)]                                             # :291-293
            fgp, bgp = net.getPrototype(fgf, bgf)                                       # :297
            prototypes = [bgp] + fgp                                                    # :300
            protos.append(torch.cat(prototypes, 0))
            dist = [net.calDist(inter[:, epi], p) for p in prototypes]                  # :301
            pred = torch.stack(dist, dim=1)                                             # :302
            preds.append(pred)
            outs.append(F.interpolate(pred, size=img_size, mode="bilinear"))            # :303
        outs = torch.stack(outs, dim=1)                                                 # :305
        caps["protos"] = torch.stack(protos, 0)
        return outs.view(-1, *outs.shape[2:]), preds                                    # :306

    refinement = {}
    for i in range(net.num_iter):                                                       # :281
        inter = net.cre(qry_fts[0] * qry_mask, qry_fts[0] * (1 - qry_mask))[None]      # :283
        caps["inter"].append(inter[0])
        logits, _ = match(inter)
        outputs = logits.softmax(dim=1)[:, 1, ...]                                      # :308
        if net.backbone_cfg["soft_mask"] == False:  # noqa: E712                        # :309-310
            outputs = (outputs > 0.5).float()
        qry_mask = F.avg_pool2d(outputs.unsqueeze(1), net.scale)                        # :311
        refinement[i] = logits                                                          # :312
    output, preds = match(inter)                                                        # :320-337
    align_loss = 0
    if net.config["align"] and net.training:                                            # :340-343
        for epi in range(B):
            align_loss = align_loss + net.alignLoss(inter[:, epi], preds[epi], supp_fts[:, :, epi], fore[:, :, epi],
                                                    back[:, :, epi])
    return {"output": output, "align_loss": align_loss / B, "refinement": refinement}    # :348-350


def gen_composed(tag, size, B, T, seed, n_ways, n_shots, d4_stride=4):
    """Fixture set 4 of SURVEY.md §8c: the composed reference (composed_forward) on a multi-shot / multi-way episode —
    stage taps, logits per iteration, prototypes, align loss, loss, gradient norms / heads, BatchNorm buffers — and the
    oracle's multi-shot / multi-way branch (oracle/rpnet_oracle.py rp_net_forward) checked against it, both modes."""
    cfg = dict(CFG)
    cfg["n_iter_refinement"] = T
    net = build_ref(cfg)
    net.train(True)
    ep = make_episode(seed, B, size, n_shots=n_shots, n_ways=n_ways)
    si, fg, bg, qi, ql, appr = to_t(ep)
    caps = {}
    out = composed_forward(net, cfg, si, fg, bg, qi, appr, caps)
    loss = O.total_loss(out, ql, cfg["align_loss_scaler"])
    assert torch.equal(out["output"], out["refinement"][T - 1])
    # on a 1-way 1-shot episode the composition IS the reference's forward (checked once per run in __main__)
    fx = {"meta": np.array([size, B, T, 1, seed, n_ways, n_shots]),
          "in_checksum": np.array([float(ep["query_images"].astype(np.float64).sum()),
                                   float(ep["support_images"][0][0].astype(np.float64).sum()),
                                   float(ep["appr_query_labels"].sum()), float(ep["support_fg"][0][0].sum())]),
          "output": out["output"], "loss": loss, "align_loss": torch.as_tensor(out["align_loss"]).float(),
          "supp_d4": caps["supp_d4"][:, :, :, ::d4_stride], "qry_d4": caps["qry_d4"][:, ::d4_stride],
          "supp_fts": caps["supp_fts"], "protos": caps["protos"], "strides": np.array([1, d4_stride, 1])}
    for i in range(T):
        fx[f"refinement_{i}"] = out["refinement"][i]
        fx[f"inter_{i}"] = caps["inter"][i]
        p = out["refinement"][i].softmax(1)[:, 1]
        fx[f"fg_frac_{i}"] = (p > 0.5).float().mean()
        fx[f"next_mask_{i}"] = torch.nn.functional.avg_pool2d((p > 0.5).float().unsqueeze(1), 4)
        pred = (p > 0.5).long()
        fx[f"dice_{i}"] = 2.0 * (pred * ql).sum() / (pred.sum() + ql.sum() + 1e-7)
    loss.backward()
    names, norms, heads = [], [], []
    for n, p in net.named_parameters():
        names.append(n)
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        norms.append(g.double().norm().item())
        heads.append(torch.nn.functional.pad(g.flatten()[:32], (0, max(0, 32 - g.numel()))))
    fx["grad_names"], fx["grad_norms"], fx["grad_heads"] = np.array(names), np.array(norms), torch.stack(heads, 0)
    fx["unused"] = np.array([n for n, p in net.named_parameters() if p.grad is None])
    for k, v in net.state_dict().items():
        if "running" in k or "num_batches" in k:
            fx["sd." + k] = v
    for as_written in (True, False):
        P = O.seeded_params(cfg["mask_refinement_correlation_radius"], requires_grad=True)
        taps = {}
        o = O.rp_net_forward(P, cfg, si, fg, bg, qi, appr, True, align=True, as_written=as_written, taps=taps)
        ol = O.total_loss(o, ql, cfg["align_loss_scaler"])
        w = f"{tag}[as_written={as_written}]"
        close(taps["supp_d4"], caps["supp_d4"], 1e-4, w + ".supp_d4")
        close(taps["qry_d4"], caps["qry_d4"], 1e-4, w + ".qry_d4")
        close(taps["supp_fts"], caps["supp_fts"], 1e-4, w + ".supp_fts")
        close(taps["protos"], caps["protos"], 1e-4, w + ".protos")
        flips = 0
        for i in range(T):
            flips += ((o["refinement"][i].softmax(1)[:, 1] > 0.5) != (out["refinement"][i].softmax(1)[:, 1] > 0.5)).sum().item()
            close(taps[f"inter_{i}"][0], caps["inter"][i], 1e-4, w + f".inter[{i}]")
            close(o["refinement"][i], out["refinement"][i], 1e-4, w + f".refinement[{i}]")
        close(o["output"], out["output"], 1e-4, w + ".output")
        close(o["align_loss"], out["align_loss"], 1e-5, w + ".align_loss")
        close(ol, loss, 1e-5, w + ".loss")
        ol.backward()
        params = dict(net.named_parameters())
        for n, p in params.items():
            if p.grad is None:
                assert P[n].grad is None, n
                continue
            e = (P[n].grad - p.grad).double().norm().item()
            sib = params.get(n.rsplit(".", 1)[0] + ".weight")
            floor = 1e-5 * max(1.0, sib.grad.double().norm().item() if sib is not None and sib.grad is not None else 1.0)
            assert e < 2e-3 * p.grad.double().norm().item() + floor, f"{w} grad {n}: abs {e:.2e}"
        for k, v in net.state_dict().items():
            if "running" in k or "num_batches" in k:
                close(P[k], v, 1e-5, w + "." + k)
        print(f"  oracle == composed reference on {w} (threshold flips: {flips})")
    save(tag, **fx)


def check_composition_is_the_reference():
    """composed_forward on a 1-way 1-shot episode against the reference's own RP_Net.forward: bit-identical outputs, align
    loss and BatchNorm buffers — the composition adds nothing of its own where the reference has behaviour."""
    cfg = dict(CFG)
    cfg["n_iter_refinement"] = 2
    si, fg, bg, qi, ql, appr = to_t(make_episode(1001, 2, 64))
    a, b = build_ref(cfg), build_ref(cfg)
    a.train(True); b.train(True)
    oa = a(si, fg, bg, qi, appr_query_labels=appr)
    ob = composed_forward(b, cfg, si, fg, bg, qi, appr, {})
    assert torch.equal(oa["output"], ob["output"]) and torch.equal(oa["align_loss"], ob["align_loss"])
    for i in range(2):
        assert torch.equal(oa["refinement"][i], ob["refinement"][i])
    for (k, v), (_, u) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(v, u), k
    print("  composed_forward == RP_Net.forward (bit-identical) on a 1-way 1-shot episode")


def gen_vgg():
    enc = REF["vgg"].Encoder(3, None)
    sd = {f"vgg.{k}": v for k, v in enc.state_dict().items()}
    from rpnet_amd.utils.seeding import seed_state_dict
    enc.load_state_dict({k[4:]: v for k, v in seed_state_dict(sd).items()})
    x = rnd(61, 1, 3, 64, 64)
    with torch.no_grad():
        y = enc(x)
    save("vgg", x=x, y=y)


if __name__ == "__main__":
    assert os.path.isdir("/root/reference"), "gen_golden.py only runs where the reference is mounted"
    gen_ops()
    gen_blocks()
    gen_model("m64_train", 64, 2, 2, True, 1001)
    gen_model("m64_eval", 64, 2, 2, False, 1001)
    gen_model("m128_train", 128, 1, 1, True, 1002, d4_stride=4)      # BASELINE config 1
    gen_model("m256_train", 256, 2, 5, True, 1003, stride=8, d4_stride=16, fts_stride=4)
    for mfm in ("x", "x2", "x3"):      # the yaml's non-default mask_feature_map settings
        gen_model(f"m64_train_{mfm}", 64, 2, 2, True, 1004, d4_stride=4, mask_feature_map=mfm)
    gen_vgg()
    # SURVEY.md §8c fixture set 4: the extension rows (BASELINE configs[2] / configs[4] shape classes), composed reference
    check_composition_is_the_reference()
    gen_composed("m64_5shot", 64, 2, 2, 1005, n_ways=1, n_shots=5)
    gen_composed("m64_2way", 64, 2, 2, 1006, n_ways=2, n_shots=1)
    gen_composed("m64_2way2shot", 64, 2, 2, 1007, n_ways=2, n_shots=2)
    print("golden fixtures regenerated; oracle pinned against the reference")
