"""ORACLE — test infrastructure, not product code.

A from-scratch, CPU, plain-PyTorch fp32 restatement of the RP-Net data-parallel
hot path of uci-cbcl/RP-Net (encoder -> context-correlation encoder -> prototype
matcher -> T-step mask refinement loop, plus the losses the bench back-propagates).
It exists to CHECK the HIP path and to be timed as the CPU baseline
(`bench.py` cpu_baseline, kind "port").  Only `tests/`, `__graft_entry__.smoke()`
and `bench.py`'s cpu_baseline leg may import it; the product path under
`rpnet_amd/` never does and fails loudly when its HIP library is missing.

Parity status: PINNED.  The reference holds no tests or golden vectors of its own
(SURVEY.md §4), so the oracle is pinned against outputs of the reference itself,
imported in the build container by `tests/golden/gen_golden.py` (third-party
modules that are absent there are stubbed; none is touched on this path).  That
script writes the fixtures under `tests/golden/*.npz`; `tests/test_oracle_golden.py`
checks this file against them on every CPU test run.

Every function cites the reference lines it restates (paths relative to
/root/reference).  Parameters are a flat {state_dict key: tensor} dict with the
reference's 147 key names; BatchNorm buffers are updated in place like nn.BatchNorm2d.

Two modes:
  as_written=True   the reference's own operator sequence (all-pairs correlation +
                    grid_sample, explicit bilinear up-sampling in getFeatures, prototypes
                    recomputed every iteration, final pass recomputed).  This is what
                    "the reference CPU path" costs and is the timed CPU baseline.
  as_written=False  the algebraically identical forms the HIP path implements
                    (local-window correlation, adjoint-bilinear masked pooling,
                    prototypes hoisted out of the loop, final pass aliased).
"""
import math

import torch
import torch.nn.functional as F

BN_EPS = 1e-5       # nn.BatchNorm2d default (net/modules.py:48)
BN_MOMENTUM = 0.1   # nn.BatchNorm2d default


# --------------------------------------------------------------------------- blocks
def _bn_relu(P, prefix, x, training):
    """nn.BatchNorm2d + nn.ReLU (net/modules.py:48-49,51-52,68-69)."""
    if training:
        P[prefix + ".num_batches_tracked"] += 1
    y = F.batch_norm(x, P[prefix + ".running_mean"], P[prefix + ".running_var"],
                     P[prefix + ".weight"], P[prefix + ".bias"], training, BN_MOMENTUM, BN_EPS)
    return F.relu(y)


def conv_bn_relu(P, conv, bn, x, training, padding=1):
    """Conv2d(bias=True) -> BatchNorm2d -> ReLU (net/modules.py:47-49)."""
    y = F.conv2d(x, P[conv + ".weight"], P[conv + ".bias"], stride=1, padding=padding)
    return _bn_relu(P, bn, y, training)


def conv_block(P, name, x, training):
    """net/modules.py:42-58: two conv3x3+BN+ReLU; Sequential indices 0,1 and 3,4."""
    x = conv_bn_relu(P, name + ".conv.0", name + ".conv.1", x, training)
    return conv_bn_relu(P, name + ".conv.3", name + ".conv.4", x, training)


def up_conv(P, name, x, training):
    """net/modules.py:61-75: nearest x2 upsample -> conv3x3+BN+ReLU; indices 1,2."""
    x = F.interpolate(x, scale_factor=2, mode="nearest")
    return conv_bn_relu(P, name + ".up.1", name + ".up.2", x, training)


def unet_d4(P, x, training, prefix="encoder", mask=None, mask_feature_map=False):
    """U_Net.forward (net/unet.py:435-467).  mask_feature_map 'x' / 'x2' / 'x3' (:437-449): the mask [N,1,H,W] is concatenated
    as one more input channel of Conv1 / Conv2 (average-pooled by 2) / Conv3 (by 4); False / 'no': the mask is ignored."""
    e = prefix + "."
    if mask_feature_map == "x":
        x = torch.cat([x, mask], 1)                                                  # :437-438
    x1 = conv_block(P, e + "Conv1", x, training)
    x2 = F.max_pool2d(x1, 2, 2)
    if mask_feature_map == "x2":
        x2 = torch.cat([x2, F.avg_pool2d(mask, 2)], 1)                               # :443-444
    x2 = conv_block(P, e + "Conv2", x2, training)
    x3 = F.max_pool2d(x2, 2, 2)
    if mask_feature_map == "x3":
        x3 = torch.cat([x3, F.avg_pool2d(mask, 4)], 1)                               # :448-449
    x3 = conv_block(P, e + "Conv3", x3, training)
    return _unet_tail(P, e, x3, training)


def _unet_tail(P, e, x3, training):
    x4 = conv_block(P, e + "Conv4", F.max_pool2d(x3, 2, 2), training)
    x5 = conv_block(P, e + "Conv5", F.max_pool2d(x4, 2, 2), training)
    d5 = up_conv(P, e + "Up5", x5, training)
    d5 = conv_block(P, e + "Up_conv5", torch.cat((x4, d5), 1), training)
    d4 = up_conv(P, e + "Up4", d5, training)
    d4 = conv_block(P, e + "Up_conv4", torch.cat((x3, d4), 1), training)
    return d4


def vgg_encoder(P, x, prefix="vgg"):
    """vgg.Encoder.forward (net/vgg.py:8-58): (conv3x3, ReLU) blocks 2-2-3-3-3, MaxPool2d(3, 2, 1) x3,
    MaxPool2d(3, 1, 1), last block dilation 2 without its final ReLU.  Keys `features.<b>.<i>.*`."""
    plan = [(0, 2, 1, True), (2, 2, 1, True), (4, 3, 1, True), (6, 3, 1, True), (8, 3, 2, False)]
    for bi, (blk, n, dil, last_relu) in enumerate(plan):
        for i in range(n):
            k = f"{prefix}.features.{blk}.{2 * i}"
            x = F.conv2d(x, P[k + ".weight"], P[k + ".bias"], padding=dil, dilation=dil)
            if i != n - 1 or last_relu:
                x = F.relu(x)
        if bi < 3:
            x = F.max_pool2d(x, 3, 2, 1)
        elif bi == 3:
            x = F.max_pool2d(x, 3, 1, 1)
    return x


# --------------------------------------------------------------------- correlation
def correlation_as_written(fmap1, fmap2, r):
    """Correlation + coords_grid + bilinear_sampler, operator for operator
    (net/rp_net.py:130-181): all-pairs matmul / sqrt(C), then grid_sample of a
    (2r+1)^2 integer-offset window around every pixel, zeros padding,
    align_corners=True."""
    b, c, h, w = fmap1.shape
    corr = torch.matmul(fmap1.reshape(b, c, h * w).transpose(1, 2), fmap2.reshape(b, c, h * w))
    corr = corr.view(b, h, w, 1, h, w) / torch.sqrt(torch.tensor(c).float())
    corr = corr.view(-1, 1, h, w)
    ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    coords = torch.stack([xs, ys], dim=0).float()[None].repeat(b, 1, 1, 1).permute(0, 2, 3, 1)
    d = torch.linspace(-r, r, 2 * r + 1)
    delta = torch.stack(torch.meshgrid(d, d, indexing="ij"), dim=-1)
    coords_lvl = coords.reshape(b * h * w, 1, 1, 2) + delta.view(1, 2 * r + 1, 2 * r + 1, 2)
    xg, yg = coords_lvl.split([1, 1], dim=-1)
    grid = torch.cat([2 * xg / (w - 1) - 1, 2 * yg / (h - 1) - 1], dim=-1)
    out = F.grid_sample(corr, grid, align_corners=True)
    return out.view(b, h, w, -1).permute(0, 3, 1, 2).contiguous().float()


def local_correlation(fmap1, fmap2, r):
    """Closed form of net/rp_net.py:153-181:
    out[b, a*(2r+1)+c, y, x] = <fmap1[b,:,y,x], fmap2[b,:,y+(c-r),x+(a-r)]>/sqrt(C),
    zero outside the image.  First window axis `a` moves x, second `c` moves y,
    because `delta` is (dy,dx)-ordered but added to (x,y) coordinates (:169-175)."""
    b, c, h, w = fmap1.shape
    k = 2 * r + 1
    f2 = F.pad(fmap2, (r, r, r, r))
    out = fmap1.new_zeros(b, k * k, h, w)
    inv = 1.0 / math.sqrt(c)
    for a in range(k):
        for cc in range(k):
            out[:, a * k + cc] = (fmap1 * f2[:, :, cc:cc + h, a:a + w]).sum(1) * inv
    return out


def cre(P, fm1, fm2, training, radius, as_written, prefix="cre"):
    """ContextCorrelationEncoder.forward (net/rp_net.py:77-84); w_context/out unused."""
    fm1 = conv_bn_relu(P, prefix + ".w_k.0", prefix + ".w_k.1", fm1, training)
    fm2 = conv_bn_relu(P, prefix + ".w_q.0", prefix + ".w_q.1", fm2, training)
    corr = correlation_as_written(fm1, fm2, radius) if as_written else local_correlation(fm1, fm2, radius)
    return conv_bn_relu(P, prefix + ".q.0", prefix + ".q.1", torch.cat([corr, fm1], 1), training, padding=0)


# ------------------------------------------------------------------------- matcher
def get_features_as_written(fts, mask):
    """getFeatures (net/rp_net.py:366-376). fts [1,C,h,w], mask [1,H,W] -> [1,C]."""
    fts = F.interpolate(fts, size=mask.shape[-2:], mode="bilinear")
    return torch.sum(fts * mask[None], dim=(2, 3)) / (mask[None].sum(dim=(2, 3)) + 1e-5)


def bilinear_taps(out_size, in_size):
    """Source taps of F.interpolate(mode='bilinear', align_corners=False):
    src = max((dst + .5) * in/out - .5, 0); i0 = floor(src); i1 = min(i0 + 1, in - 1)."""
    scale = in_size / out_size
    dst = torch.arange(out_size, dtype=torch.float32)
    src = ((dst + 0.5) * scale - 0.5).clamp(min=0)
    i0 = src.floor().long().clamp(max=in_size - 1)
    i1 = (i0 + 1).clamp(max=in_size - 1)
    l1 = src - i0.float()
    return i0, i1, 1.0 - l1, l1


def bilinear_adjoint(mask, h, w):
    """U^T mask: adjoint of the bilinear up-sampler applied to a [..., H, W] map
    -> [..., h, w], so that sum(up(f) * m) == sum(f * U^T m)."""
    H, W = mask.shape[-2:]
    y0, y1, wy0, wy1 = bilinear_taps(H, h)
    x0, x1, wx0, wx1 = bilinear_taps(W, w)
    lead = mask.shape[:-2]
    m = mask.reshape(-1, H, W)
    rows = m.new_zeros(m.shape[0], h, W)
    rows.index_add_(1, y0, m * wy0[None, :, None])
    rows.index_add_(1, y1, m * wy1[None, :, None])
    out = m.new_zeros(m.shape[0], h, w)
    out.index_add_(2, x0, rows * wx0[None, None, :])
    out.index_add_(2, x1, rows * wx1[None, None, :])
    return out.reshape(*lead, h, w)


def get_features_adjoint(fts, mask):
    """Same value as get_features_as_written without materialising the up-sampled map."""
    h, w = fts.shape[-2:]
    am = bilinear_adjoint(mask, h, w)
    return torch.sum(fts * am[None], dim=(2, 3)) / (mask[None].sum(dim=(2, 3)) + 1e-5)


def get_prototype(fg_fts, bg_fts):
    """getPrototype (net/rp_net.py:379-391)."""
    n_ways, n_shots = len(fg_fts), len(fg_fts[0])
    fg = [sum(way) / n_shots for way in fg_fts]
    bg = sum([sum(way) / n_shots for way in bg_fts]) / n_ways
    return fg, bg


def cal_dist(fts, prototype, scaler=20):
    """calDist (net/rp_net.py:353-363)."""
    return F.cosine_similarity(fts, prototype[..., None, None], dim=1) * scaler


def align_loss(qry_fts, pred, supp_fts, fore_mask, back_mask):
    """alignLoss (net/rp_net.py:394-440).  qry_fts [N,C,h,w]; pred [N,1+Wa,h,w];
    supp_fts [Wa,Sh,C,h,w]; masks [Wa,Sh,H,W]."""
    n_ways, n_shots = len(fore_mask), len(fore_mask[0])
    pred_mask = pred.argmax(dim=1, keepdim=True)
    binary = [pred_mask == i for i in range(1 + n_ways)]
    skip = [i for i in range(n_ways) if binary[i + 1].sum() == 0]
    pm = torch.stack(binary, dim=1).float()
    protos = torch.sum(qry_fts.unsqueeze(1) * pm, dim=(0, 3, 4)) / (pm.sum((0, 3, 4)) + 1e-5)
    loss = 0
    for way in range(n_ways):
        if way in skip:
            continue
        pr = [protos[[0]], protos[[way + 1]]]
        for shot in range(n_shots):
            f = supp_fts[way, [shot]]
            sp = torch.stack([cal_dist(f, p) for p in pr], dim=1)
            sp = F.interpolate(sp, size=fore_mask.shape[-2:], mode="bilinear")
            lab = torch.full_like(fore_mask[way, shot], 255).long()
            lab[fore_mask[way, shot] == 1] = 1
            lab[back_mask[way, shot] == 1] = 0
            loss = loss + F.cross_entropy(sp, lab[None], ignore_index=255) / n_shots / n_ways
    return loss


# -------------------------------------------------------------------------- losses
def dice_loss_softmax(logits, true, eps=1e-7):
    """dice_loss_softmax, num_classes > 1 branch (net/rp_net.py:87-120)."""
    c = logits.shape[1]
    one_hot = torch.eye(c)[true].permute(0, 3, 1, 2).float()
    probas = F.softmax(logits, dim=1)
    inter = torch.sum(probas * one_hot, (0, 2, 3))
    card = torch.sum(probas + one_hot, (0, 2, 3))
    return 1 - (2.0 * inter / (card + eps)).mean()


def dice_ce(logits, true, eps=1e-7):
    """dice_ce (net/rp_net.py:123-127)."""
    return dice_loss_softmax(logits, true, eps) + F.cross_entropy(logits, true)


# ------------------------------------------------------------------------- forward
def rp_net_forward(P, cfg, supp_imgs, fore_mask, back_mask, qry_imgs, appr_query_labels,
                   training, align=True, as_written=False, taps=None, forced_masks=None):
    """RP_Net.forward, UNet backbone, use_relation_enc == 'relation'
    (net/rp_net.py:226-350).  `taps` (dict) collects stage-boundary tensors for the
    parity tests; `forced_masks` {i: [B,1,h,w]} teacher-forces the mask that enters
    iteration i (i >= 1) so hard-threshold flips cannot compound.

    Multi-shot / multi-way (no reference behaviour, SURVEY.md §8a extension rows):
    CRE runs once per (way, shot) on that shot's features with that shot's own pooled
    foreground mask, way-major / shot-minor.
    """
    n_ways, n_shots = len(supp_imgs), len(supp_imgs[0])
    B = supp_imgs[0][0].shape[0]
    img_size = qry_imgs[0].shape[-2:]
    scale = cfg.get("scale", 4)
    T = cfg["n_iter_refinement"]
    radius = cfg["mask_refinement_correlation_radius"]
    taps = taps if taps is not None else {}

    imgs = torch.cat([torch.cat(way, 0) for way in supp_imgs], 0)          # :245
    mfm = cfg.get("mask_feature_map", False)
    enc_mask = fore_mask[0][0].unsqueeze(1)                                # both encoder calls get the SUPPORT mask (:248,257)
    supp_d4 = unet_d4(P, imgs, training, mask=enc_mask, mask_feature_map=mfm)           # :248-249
    hw = supp_d4.shape[-2:]
    supp_d4 = supp_d4.view(n_ways, n_shots, B, -1, *hw)                    # :252
    qry_d4 = unet_d4(P, torch.cat(qry_imgs, 0), training, mask=enc_mask, mask_feature_map=mfm)   # :254-258
    qry_fts = qry_d4.view(len(qry_imgs), B, -1, *hw)
    taps["supp_d4"], taps["qry_d4"] = supp_d4, qry_d4

    fore = torch.stack([torch.stack(way, 0) for way in fore_mask], 0)      # :264-267
    back = torch.stack([torch.stack(way, 0) for way in back_mask], 0)
    qry_mask = F.avg_pool2d(appr_query_labels.unsqueeze(1), scale)         # :269-270

    if n_ways == 1 and n_shots == 1:
        sm = F.avg_pool2d(fore[0][0].unsqueeze(1), scale)                  # :271-272
        supp_fts = cre(P, supp_d4[0][0] * sm, supp_d4[0][0] * (1 - sm), training, radius,
                       as_written)[None, None]                             # :275
    else:
        rows = []
        for wa in range(n_ways):
            row = []
            for s in range(n_shots):
                sm = F.avg_pool2d(fore[wa][s].unsqueeze(1), scale)
                row.append(cre(P, supp_d4[wa][s] * sm, supp_d4[wa][s] * (1 - sm), training,
                               radius, as_written))
            rows.append(torch.stack(row, 0))
        supp_fts = torch.stack(rows, 0)
    taps["supp_fts"] = supp_fts

    getf = get_features_as_written if as_written else get_features_adjoint

    def prototypes(epi):
        fg = [[getf(supp_fts[wa, s, [epi]], fore[wa, s, [epi]]) for s in range(n_shots)]
              for wa in range(n_ways)]                                     # :288-293
        bg = [[getf(supp_fts[wa, s, [epi]], back[wa, s, [epi]]) for s in range(n_shots)]
              for wa in range(n_ways)]
        fgp, bgp = get_prototype(fg, bg)                                   # :297
        return [bgp] + fgp                                                 # :300

    def match(q, protos_per_epi):
        outs, preds = [], []
        for epi in range(B):
            dist = [cal_dist(q[:, epi], p) for p in protos_per_epi[epi]]  # :301
            pred = torch.stack(dist, 1)                                    # :302
            preds.append(pred)
            outs.append(F.interpolate(pred, size=img_size, mode="bilinear"))   # :303
        outs = torch.stack(outs, 1)
        return outs.view(-1, *outs.shape[2:]), preds                       # :305-306

    protos = None
    refinement = {}
    inter = qry_fts
    preds = None
    for i in range(T):                                                     # :281
        if forced_masks is not None and i in forced_masks:
            qry_mask = forced_masks[i]
        taps[f"qry_mask_{i}"] = qry_mask
        inter = cre(P, qry_fts[0] * qry_mask, qry_fts[0] * (1 - qry_mask), training, radius,
                    as_written)[None]                                      # :283
        taps[f"inter_{i}"] = inter
        if as_written or protos is None:
            protos = [prototypes(e) for e in range(B)]
        logits, preds = match(inter, protos)
        prob = logits.softmax(dim=1)[:, 1]                                 # :308
        if cfg["soft_mask"] == False:  # noqa: E712  (reference compares with ==, :309)
            prob = (prob > 0.5).float()                                    # :310
        qry_mask = F.avg_pool2d(prob.unsqueeze(1), scale)                  # :311
        refinement[i] = logits                                             # :312
    taps["protos"] = torch.stack([torch.cat(p, 0) for p in protos], 0) if protos else None

    # final pass (:314-343): identical inputs and ops to the last iteration
    if as_written or T == 0:
        protos = [prototypes(e) for e in range(B)]
        output, preds = match(inter, protos)
    else:
        output = refinement[T - 1]
    al = 0
    if align and training:                                                 # :340-343
        for epi in range(B):
            al = al + align_loss(inter[:, epi], preds[epi], supp_fts[:, :, epi],
                                 fore[:, :, epi], back[:, :, epi])
    return {"output": output, "align_loss": al / B, "refinement": refinement}


def total_loss(out, query_labels, align_loss_scaler=1.0):
    """The harness's training objective (the reference has no train loop, SURVEY.md
    §8d): dice_ce(output) + sum_i dice_ce(refinement[i]) + scaler * align_loss."""
    loss = dice_ce(out["output"], query_labels)
    for v in out["refinement"].values():
        loss = loss + dice_ce(v, query_labels)
    return loss + align_loss_scaler * out["align_loss"]


# ---------------------------------------------------------------------- parameters
def param_shapes(radius=5, in_ch=1, mask_feature_map=False):
    """The 147 state_dict entries RP_Net(backbone='UNet') creates
    (net/rp_net.py:209-221, net/unet.py:394-430): name -> shape, in module order.  mask_feature_map 'x' / 'x2' / 'x3' gives
    Conv1 / Conv2 / Conv3 one more input channel (net/unet.py:401-414)."""
    shapes = {}

    def conv(name, cin, cout, k):
        shapes[name + ".weight"] = (cout, cin, k, k)
        shapes[name + ".bias"] = (cout,)

    def bn(name, c):
        for leaf in ("weight", "bias", "running_mean", "running_var"):
            shapes[f"{name}.{leaf}"] = (c,)
        shapes[name + ".num_batches_tracked"] = ()

    def block(name, cin, cout):
        conv(name + ".conv.0", cin, cout, 3); bn(name + ".conv.1", cout)
        conv(name + ".conv.3", cout, cout, 3); bn(name + ".conv.4", cout)

    def up(name, cin, cout):
        conv(name + ".up.1", cin, cout, 3); bn(name + ".up.2", cout)

    f = [64, 128, 256, 512, 1024]
    block("encoder.Conv1", in_ch + (mask_feature_map == "x"), f[0])
    block("encoder.Conv2", f[0] + (mask_feature_map == "x2"), f[1])
    block("encoder.Conv3", f[1] + (mask_feature_map == "x3"), f[2])
    block("encoder.Conv4", f[2], f[3])
    block("encoder.Conv5", f[3], f[4])
    up("encoder.Up5", f[4], f[3]); block("encoder.Up_conv5", f[3] * 2, f[3])
    up("encoder.Up4", f[3], f[2]); block("encoder.Up_conv4", f[2] * 2, f[2])
    c = 256
    conv("cre.w_k.0", c, c, 3); bn("cre.w_k.1", c)
    conv("cre.w_q.0", c, c, 3); bn("cre.w_q.1", c)
    conv("cre.w_context.0", 2 * c, c, 1); bn("cre.w_context.1", c)
    conv("cre.q.0", c + (2 * radius + 1) ** 2, 64, 1); bn("cre.q.1", 64)
    conv("cre.out.0", 2 * c, 64, 1); bn("cre.out.1", 64)
    return shapes


def seeded_params(radius=5, requires_grad=False, mask_feature_map=False):
    from rpnet_amd.utils.seeding import seeded_tensor
    P = {}
    for name, shape in param_shapes(radius, mask_feature_map=mask_feature_map).items():
        leaf = name.rsplit(".", 1)[-1]
        like = torch.empty(shape, dtype=torch.int64 if leaf == "num_batches_tracked" else torch.float32)
        t = seeded_tensor(name, like)
        if requires_grad and t.is_floating_point() and not leaf.startswith("running"):
            t.requires_grad_(True)
        P[name] = t
    return P
