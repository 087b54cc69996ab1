"""Synthetic abdominal-CT-shaped episodes.

Stand-in for the reference's FewshotRegReader item contract
(/root/reference/dataset/few_shot_reader.py:599-647): fp32 images in [-1, 1]
(background -1.0 = the -1024 HU pad after utils/util.py:455-467 normalize), an
elliptical body with smooth texture and a few brighter organs, a support
foreground mask on one organ, the query as the same scene under a small random
affine, and `appr_query_labels` = the support mask under a slightly wrong
affine (what the registration pre-step hands the hot path).
numpy RandomState only, so the same seed gives the same bytes on every box.
"""
import numpy as np
from scipy import ndimage


def _ellipse(h, w, cy, cx, ry, rx, ang=0.0):
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    y -= cy
    x -= cx
    c, s = np.cos(ang), np.sin(ang)
    yr = c * y + s * x
    xr = -s * y + c * x
    return ((yr / ry) ** 2 + (xr / rx) ** 2) <= 1.0


def _affine(img, rot_deg, scale, shift, order):
    h, w = img.shape
    a = np.deg2rad(rot_deg)
    m = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]], dtype=np.float64) / scale
    c = np.array([h / 2.0, w / 2.0])
    off = c - m @ (c + np.asarray(shift, dtype=np.float64))
    cval = float(img.flat[0]) if order else 0.0
    return ndimage.affine_transform(img, m, offset=off, order=order, mode="constant", cval=cval)


def make_scene(rs, h, w):
    img = np.full((h, w), -1.0, dtype=np.float32)
    body = _ellipse(h, w, h / 2, w / 2, 0.42 * h, 0.36 * w)
    tex = ndimage.gaussian_filter(rs.randn(h, w).astype(np.float32), sigma=max(h / 32.0, 1.0))
    tex = 0.08 * tex / (np.abs(tex).max() + 1e-6)
    img[body] = (-0.55 + tex)[body]
    n_org = rs.randint(3, 7)
    masks = []
    for _ in range(n_org):
        cy = h / 2 + rs.uniform(-0.22, 0.22) * h
        cx = w / 2 + rs.uniform(-0.18, 0.18) * w
        ry = rs.uniform(0.06, 0.16) * h
        rx = rs.uniform(0.06, 0.16) * w
        m = _ellipse(h, w, cy, cx, ry, rx, rs.uniform(0, np.pi)) & body
        img[m] = rs.uniform(-0.45, -0.2)
        masks.append(m)
    img += 0.01 * rs.randn(h, w).astype(np.float32)
    fracs = [mk.mean() for mk in masks]
    ok = [i for i, f in enumerate(fracs) if 0.02 <= f <= 0.15]
    pick = ok[0] if ok else int(np.argmax(fracs))
    # later organs may overwrite earlier ones: the label is the picked ellipse itself
    return np.clip(img, -1, 1).astype(np.float32), masks[pick].astype(np.float32)


def make_episode(seed, batch, size, n_shots=1, n_ways=1):
    """Returns dict of numpy arrays following RP_Net.forward's input contract
    (/root/reference/net/rp_net.py:226-238):
      support_images [Wa][Sh] -> [B,1,H,W]; support_fg/bg [Wa][Sh] -> [B,H,W];
      query_images [B,1,H,W]; query_labels [B,H,W] int64; appr_query_labels [B,H,W] float.
    """
    rs = np.random.RandomState(seed)
    h = w = size
    supp = [[np.zeros((batch, 1, h, w), np.float32) for _ in range(n_shots)] for _ in range(n_ways)]
    fg = [[np.zeros((batch, h, w), np.float32) for _ in range(n_shots)] for _ in range(n_ways)]
    qry = np.zeros((batch, 1, h, w), np.float32)
    qlab = np.zeros((batch, h, w), np.int64)
    appr = np.zeros((batch, h, w), np.float32)
    for b in range(batch):
        img, lab = make_scene(rs, h, w)
        rot, sc = rs.uniform(-5, 5), rs.uniform(0.92, 1.08)
        sh = rs.uniform(-10, 10, size=2) * (h / 256.0)
        qry[b, 0] = _affine(img, rot, sc, sh, 1) + 0.01 * rs.randn(h, w).astype(np.float32)
        qlab[b] = (_affine(lab, rot, sc, sh, 0) > 0.5).astype(np.int64)
        # registration output: right transform perturbed a little
        appr[b] = (_affine(lab, rot + rs.uniform(-2, 2), sc * rs.uniform(0.97, 1.03),
                           sh + rs.uniform(-3, 3, size=2) * (h / 256.0), 0) > 0.5).astype(np.float32)
        for wa in range(n_ways):
            for s in range(n_shots):
                if wa == 0 and s == 0:
                    supp[wa][s][b, 0], fg[wa][s][b] = img, lab
                else:  # extra shots/ways: a jittered view of the same scene
                    r2, s2 = rs.uniform(-4, 4), rs.uniform(0.95, 1.05)
                    t2 = rs.uniform(-6, 6, size=2) * (h / 256.0)
                    supp[wa][s][b, 0] = _affine(img, r2, s2, t2, 1)
                    fg[wa][s][b] = (_affine(lab, r2, s2, t2, 0) > 0.5).astype(np.float32)
    qry = np.clip(qry, -1, 1).astype(np.float32)
    bg = [[1.0 - m for m in way] for way in fg]
    return {"support_images": supp, "support_fg": fg, "support_bg": bg, "query_images": qry,
            "query_labels": qlab, "appr_query_labels": appr}
