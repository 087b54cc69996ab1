"""Volume / slice episode readers over NRRD data: the host side that feeds the hot path from
real volumes, mirroring the reference's reader interface (dataset/few_shot_reader.py) — same class names, constructor
arguments, item keys, shapes and dtypes, same use of the `random` / `numpy.random` / `torch` generators in the same
order, so that a seeded reference run and a seeded run here pick the same volumes and slices.

  FewshotVolumeReader   few_shot_reader.py:232-398   pid lists (.csv / .npy), per-class csv of annotated volumes,
                                                     `<pid>_<roi>.nrrd` mask + `<pid>_clean.nrrd` image -> truncate,
                                                     pad to 16, annotated z-range, centre crop / pad, HU window
  FewshotSliceReader    few_shot_reader.py:438-588   k-block slice matching ('Squeeze & Excite' guided few-shot
                                                     segmentation, arXiv 1902.01314), train augmentation, the
                                                     registration pre-step and the optional mask channels
  FewshotRegReader      few_shot_reader.py:592-650   the item test_rpnet.py consumes

Where the reference runs `get_registration_field` (few_shot_reader.py:556-561) this reader calls the batched HIP path
(rpnet_amd/registration.py); it needs the GPU and fails loudly without one (no CPU fallback).  With
`use_registration_loss: False` nothing here touches the GPU.

The train-time augmentations use two packages that are absent offline: torchvision (RandomAffine, :27-48) and cv2
(getAffineTransform / warpAffine in brain_reader.py:248-294).  `random_affine` and `elastic_transform_all` restate
their documented behaviour (torchvision's tensor path: inverse affine matrix about the image centre, nearest
`grid_sample`, zero fill; cv2: three-point affine + bilinear / nearest warp) — parity of these two with the
third-party packages is UNPINNED (they cannot be imported here); everything else is pinned by
tests/golden/volume_reader.npz, produced by the reference's own classes (tests/golden/gen_golden_reader.py).
"""
import math
import os
import random

import numpy as np
import torch
import torch.nn.functional as F

from . import nrrd


# --------------------------------------------------------------------------------------------------- array helpers
def pad2factor(image, factor=16, pad_value=0):
    """reference utils/util.py:406-419"""
    return np.pad(image, [(0, -s % factor) for s in image.shape], "constant", constant_values=pad_value)


def normalize(img, minimum=-1024, maximum=3076):
    """reference utils/util.py:455-467"""
    out = np.array(img, copy=True)
    top = float(np.percentile(out, 100.0 - 0.5))
    out[out > top] = top
    out[out > maximum] = maximum
    out[out < minimum] = minimum
    return (out - minimum) / max(1, (maximum - minimum)) * 2 - 1


def keep_only_annotation_z_slices(img, mask):
    """[1,D,H,W] pair cut to the annotated z-range [first, last) — the last annotated slice is dropped, as in the
    reference (few_shot_reader.py:17-24: `d_min:d_max`)."""
    zs = np.nonzero(mask)[1]
    lo, hi = zs.min(), zs.max()
    return img[:, lo:hi], mask[:, lo:hi]


def crop(img, mask, crop_size, img_pad_value, mask_pad_value=0):
    """centre crop to at most crop_size, then pad symmetrically (extra pixel at the far end) up to crop_size
    (few_shot_reader.py:64-77)"""
    h, w = mask.shape[-2:]
    ch, cw = crop_size
    rh, rw = min(ch, h), min(cw, w)
    y0, x0 = h // 2 - rh // 2, w // 2 - rw // 2
    widths = [(0, 0), (0, 0), ((ch - rh) // 2, (ch - rh) - (ch - rh) // 2), ((cw - rw) // 2, (cw - rw) - (cw - rw) // 2)]
    win = (Ellipsis, slice(y0, y0 + rh), slice(x0, x0 + rw))
    return (np.pad(img[win], widths, mode="constant", constant_values=img_pad_value),
            np.pad(mask[win], widths, mode="constant", constant_values=mask_pad_value))


def make_support_query_same_size(support_images, support_labels, query_images, query_labels):
    """1-way 1-shot: pad support / query slices at the far end to a common H x W with each array's own minimum
    (few_shot_reader.py:80-106; the label pads take BOTH extents from the height axis, as the reference does)."""
    s_img, s_lab = support_images[0][0].numpy(), support_labels[0][0].numpy()
    q_img, q_lab = query_images.numpy(), query_labels.numpy()
    H, W = max(s_img.shape[2], q_img.shape[2]), max(s_img.shape[3], q_img.shape[3])

    def grow(a, dh, dw):
        return np.pad(a, [(0, 0)] * (a.ndim - 2) + [(0, dh), (0, dw)], "constant", constant_values=a.min())

    s_img = grow(s_img, H - s_img.shape[2], W - s_img.shape[3])
    q_img = grow(q_img, H - q_img.shape[2], W - q_img.shape[3])
    s_lab = grow(s_lab, H - s_lab.shape[1], W - s_lab.shape[1])
    q_lab = grow(q_lab, H - q_lab.shape[1], W - q_lab.shape[1])
    t = torch.from_numpy
    return [[t(s_img)]], [[t(s_lab)]], t(q_img), t(q_lab)


def compute_grid(img_size):
    """net/registration.py:171-187: [1, 2, H, W], channel 0 = x, channel 1 = y, coordinates 2 (j / (n - 1) - 0.5)"""
    h, w = img_size
    ys, xs = torch.meshgrid(torch.arange(0, h), torch.arange(0, w), indexing="ij")
    grid = torch.stack([xs, ys]).unsqueeze(0).float()
    grid[:, 0] = 2 * (grid[:, 0] / (w - 1) - 0.5)
    grid[:, 1] = 2 * (grid[:, 1] / (h - 1) - 0.5)
    return grid


# --------------------------------------------------------------------------------------------------- augmentations
def gamma_transform(img, gamma_range):
    """few_shot_reader.py:200-210 (`gamma_tansform`): a random power law on the [0,1]-mapped slice, one
    np.random.rand() draw"""
    img = (img + 1) / 2.0
    gamma = np.random.rand() * (gamma_range[1] - gamma_range[0]) + gamma_range[0]
    lo = img.min()
    span = img.max() - lo + 1e-5
    img = span * np.power((img - lo + 1e-5) * 1.0 / span, gamma) + lo
    return img * 2 - 1


gamma_tansform = gamma_transform          # the reference's spelling


def gamma_transform_with_label(img, label, gamma_range):
    """few_shot_reader.py:213-229: the power law inside the label region only"""
    return img * (1 - label) + gamma_transform(img, gamma_range) * label


gamma_tansform_with_label = gamma_transform_with_label


def _uniform(lo, hi):
    return float(torch.empty(1).uniform_(float(lo), float(hi)).item())


def random_affine(x, degrees, translate=None, scale=None, shear=None):
    """torchvision.transforms.RandomAffine(degrees, translate, scale, shear) on a [N,C,H,W] tensor, restated from its
    documentation (UNPINNED, see the module header): angle ~ U(-degrees, degrees); integer shift ~ round(U(-t W, t W));
    scale ~ U(scale); x-shear ~ U(-shear, shear); output pixel p samples the input at A^-1 p about the image centre,
    nearest neighbour, zeros outside.  Draw order: angle, tx, ty, scale, shear (torch generator)."""
    H, W = x.shape[-2:]
    deg = (-degrees, degrees) if np.isscalar(degrees) else degrees
    angle = _uniform(*deg)
    tx = ty = 0
    if translate is not None:
        tx = int(round(_uniform(-translate[0] * W, translate[0] * W)))
        ty = int(round(_uniform(-translate[1] * H, translate[1] * H)))
    s = _uniform(*scale) if scale is not None else 1.0
    shx = 0.0
    if shear is not None:
        sh = (-shear, shear) if np.isscalar(shear) else shear
        shx = _uniform(sh[0], sh[1])
    rot, sx = math.radians(angle), math.radians(shx)
    # forward map = T(translate) R(rot) Shear(sx) S(s) about the centre; rows of its inverse:
    a, b = math.cos(rot), -math.cos(rot) * math.tan(sx) - math.sin(rot)
    c, d = math.sin(rot), -math.sin(rot) * math.tan(sx) + math.cos(rot)
    m = [d / s, -b / s, 0.0, -c / s, a / s, 0.0]
    m[2] = m[0] * -tx + m[1] * -ty
    m[5] = m[3] * -tx + m[4] * -ty
    theta = torch.tensor(m, dtype=x.dtype).reshape(1, 2, 3)
    base = torch.empty(1, H, W, 3, dtype=x.dtype)
    base[..., 0] = torch.linspace(-W * 0.5 + 0.5, W * 0.5 - 0.5, W)
    base[..., 1] = torch.linspace(-H * 0.5 + 0.5, H * 0.5 - 0.5, H).unsqueeze(-1)
    base[..., 2] = 1
    grid = base.view(1, H * W, 3).bmm(theta.transpose(1, 2) / torch.tensor([0.5 * W, 0.5 * H], dtype=x.dtype)).view(1, H, W, 2)
    return F.grid_sample(x, grid.expand(x.shape[0], -1, -1, -1), mode="nearest", padding_mode="zeros", align_corners=False)


def random_transform(images, labels):
    """few_shot_reader.py:27-48: one RandomAffine(5, translate 0.2, scale 0.7-1.5) for the slice [1,1,H,W] (in [-1,1])
    and its label [1,H,W]; pixels that come out exactly 0 in the [0,1] image take the slice minimum."""
    images = (images + 1) / 2
    lo = images.min()
    both = random_affine(torch.cat([images, labels[None, ...]], dim=1), 5, translate=(0.2, 0.2), scale=(0.7, 1.5), shear=0)
    images, labels = both[:, [0]], both[:, 1]
    images[images == 0] = lo
    return images * 2 - 1, labels


def random_label_transform(labels):
    """few_shot_reader.py:51-61"""
    return random_affine(labels[None, None, ...], 5, translate=(0.02, 0.02), scale=(0.5, 1.5), shear=5)[:, 0]


def _three_point_affine(src, dst):
    """the 2x3 M with M [x, y, 1]^T = dst for three point pairs (cv2.getAffineTransform)"""
    A = np.concatenate([src, np.ones((3, 1))], axis=1).astype(np.float64)
    return np.linalg.solve(A, dst.astype(np.float64)).T


def elastic_transform_all(image, mask, alpha=1000, sigma=30, alpha_affine=0.04, padding_value=-1.0, random_state=None):
    """brain_reader.py:208-294: the same in-plane random affine (three corner points jittered by +-alpha_affine) and
    elastic displacement (Gaussian-smoothed uniform noise, sigma, times alpha) for every slice of image [1,D,H,W]
    and mask [num_class,D,H,W]; image bilinear with `padding_value` outside, mask nearest with 0 outside.  UNPINNED
    (cv2 absent; the reference draws from an unseeded RandomState)."""
    from scipy.ndimage import gaussian_filter, map_coordinates
    rs = random_state if random_state is not None else np.random.RandomState(None)
    plane = image.shape[2:]
    n_cls, D, Hh, Ww = mask.shape
    centre, half = np.float32(plane) // 2, min(plane) // 3
    pts1 = np.float32([centre + half, [centre[0] + half, centre[1] - half], centre - half])
    pts2 = pts1 + rs.uniform(-alpha_affine, alpha_affine, size=pts1.shape).astype(np.float32)
    M = _three_point_affine(pts1, pts2)
    Minv = np.linalg.inv(np.vstack([M, [0, 0, 1]]))[:2]
    dx = gaussian_filter(rs.rand(*plane) * 2 - 1, sigma) * alpha
    dy = gaussian_filter(rs.rand(*plane) * 2 - 1, sigma) * alpha
    xx, yy = np.meshgrid(np.arange(Ww), np.arange(Hh))
    src_x = Minv[0, 0] * xx + Minv[0, 1] * yy + Minv[0, 2]        # warpAffine: dst(x, y) = src(M^-1 (x, y))
    src_y = Minv[1, 0] * xx + Minv[1, 1] * yy + Minv[1, 2]
    warp_to = (yy + dy).reshape(-1, 1), (xx + dx).reshape(-1, 1)
    new_img, new_mask = np.zeros_like(image), np.zeros_like(mask)
    for z in range(D):
        aff = map_coordinates(image[0, z], (src_y, src_x), order=1, mode="constant", cval=padding_value)
        new_img[0, z] = map_coordinates(aff, warp_to, order=1, mode="constant", cval=padding_value).reshape(plane)
        for j in range(n_cls):
            if np.any(mask[j, z]):
                aff = map_coordinates(mask[j, z], (np.rint(src_y), np.rint(src_x)), order=0, mode="constant", cval=0)
                new_mask[j, z] = map_coordinates(aff, warp_to, order=0, mode="constant").reshape(plane)
    return new_img, new_mask


# ----------------------------------------------------------------------------------------------------------- readers
class FewshotVolumeReader(torch.utils.data.Dataset):
    """few_shot_reader.py:232-398.  config keys: class_csv_dir, train_classes / eval_classes, n_shot, n_way, num_slice,
    num_x, num_y, pad_value, HU_range, crop_size (default [256, 256]), do_elastic."""

    def __init__(self, data_dir, set_name, config, mode="train"):
        self.data_dir, self.cfg, self.mode = data_dir, config, mode
        self.class_csv_dir = config["class_csv_dir"]
        if set_name.endswith(".csv"):
            self.filenames = np.genfromtxt(set_name, dtype=str, delimiter="\n")
        elif set_name.endswith(".npy"):
            self.filenames = np.load(set_name)
        else:
            raise ValueError(f"set_name must be a .csv or .npy list of pids, got {set_name!r}")
        if mode not in ("train", "eval"):
            raise NotImplementedError(mode)
        self.classes = config["train_classes" if mode == "train" else "eval_classes"]
        self.read_data_meta()
        self.init_pairs()

    def read_data_meta(self):
        import pandas as pd
        wanted = set(np.atleast_1d(self.filenames).tolist())
        self.data_info, self.n_data = [], []
        for roi in self.classes:
            df = pd.read_csv(os.path.join(self.class_csv_dir, f"{roi}.csv"), dtype=str)
            rows = [{"pid": r["pid"], "z_start": r["z_start"], "z_end": r["z_end"]} for _, r in df.iterrows() if r["pid"] in wanted]
            self.data_info.append(rows)
            self.n_data.append(len(rows))

    def init_pairs(self):
        self.indices = [(c, i) for c in range(len(self.classes)) for i in range(self.n_data[c])]

    def __len__(self):
        return len(self.indices)

    def truncate_image(self, image):
        """first num_slice slices, central num_y x num_x window"""
        D, H, W = image.shape
        nx, ny = self.cfg["num_x"], self.cfg["num_y"]
        x1, x2 = max(0, W // 2 - nx // 2), min(W, W // 2 + nx // 2)
        y1, y2 = max(0, H // 2 - ny // 2), min(H, H // 2 + ny // 2)
        return image[:self.cfg["num_slice"], y1:y2, x1:x2]

    def load_image_and_mask(self, filename, roi_name):
        m, _ = nrrd.read(os.path.join(self.data_dir, f"{filename}_{roi_name}.nrrd"))
        mask = pad2factor(self.truncate_image(m.astype(np.float32)), factor=16, pad_value=0)[None, ...]
        v, _ = nrrd.read(os.path.join(self.data_dir, f"{filename}_clean.nrrd"))
        imgs = pad2factor(self.truncate_image(v), factor=16, pad_value=self.cfg["pad_value"])[np.newaxis, ...].astype(np.float32)
        imgs, mask = keep_only_annotation_z_slices(imgs, mask)
        imgs, mask = crop(imgs, mask, self.cfg.get("crop_size", [256, 256]), self.cfg.get("pad_value", -1024), 0)
        imgs = normalize(imgs, minimum=self.cfg["HU_range"][0], maximum=self.cfg["HU_range"][1])
        return {"image": imgs, "mask": mask}

    def __getitem__(self, idx, supp_idx=None):
        n_shots, n_ways = self.cfg["n_shot"], self.cfg["n_way"]
        c, q = self.indices[idx]
        pid = self.data_info[c][q]["pid"]
        others = [i for i in range(self.n_data[c]) if i != q]
        support = [(c, i) for i in random.choices(others, k=n_shots)]          # with replacement, python `random`
        if supp_idx is not None:
            support = [(c, supp_idx)]
        samples = [self.load_image_and_mask(self.data_info[ci][di]["pid"], self.classes[ci]) for ci, di in support]
        # every way serves the same shots (few_shot_reader.py:281-288)
        support_images = [[torch.from_numpy(samples[j]["image"]) for j in range(n_shots)] for _ in range(n_ways)]
        support_labels = [[torch.from_numpy(samples[j]["mask"]) for j in range(n_shots)] for _ in range(n_ways)]
        qs = self.load_image_and_mask(pid, self.classes[c])
        q_img, q_mask = qs["image"], qs["mask"]
        if self.mode == "train" and self.cfg["do_elastic"] and np.random.randint(2, size=1).item():
            q_img, q_mask = elastic_transform_all(q_img, q_mask)
        return {"support_images": support_images, "support_labels": support_labels,
                "query_images": [[torch.from_numpy(q_img)]], "query_labels": [[torch.from_numpy(q_mask)]],
                "class_id": c, "pid": pid, "supp_pids": support}


class FewshotSliceReader(torch.utils.data.Dataset):
    """few_shot_reader.py:438-588: the query volume is cut into k z-blocks; block j is matched with the slice at
    the centre of block j of the support volume.  eval: every query slice, in order, each with its block's support
    slice; train: one random slice per block, augmented, blocks shuffled."""

    def __init__(self, data_dir, set_name, config, mode="train"):
        self.cfg, self.k, self.mode = config, config["k"], mode
        self.fewshot_volume_reader = FewshotVolumeReader(data_dir, set_name, config, mode=mode)

    def __len__(self):
        return len(self.fewshot_volume_reader)

    def __getitem__(self, idx):
        from ..registration import get_registration_field
        vol = self.fewshot_volume_reader[idx]
        s_imgs, s_labs = vol["support_images"], vol["support_labels"]
        q_img, q_lab = vol["query_images"][0][0], vol["query_labels"][0][0]
        assert len(s_imgs) == 1
        depths = [v.shape[1] for v in s_imgs[0]] + [q_img.shape[1]]
        self.k = k = min([self.k] + depths)                     # sticks for later items, as in the reference
        s_pick = [np.floor(np.arange(n / k / 2, n, n / k)).astype(np.int32) for n in depths[:-1]]
        nq = depths[-1]
        q_edge = np.floor(np.array(np.arange(0, nq, nq / k).tolist() + [nq])).astype(np.int32)

        if self.mode == "train":
            sup_i = [[v[:, s_pick[i]].permute(1, 0, 2, 3).contiguous().expand(-1, 3, -1, -1).clone() for i, v in enumerate(s_imgs[0])]]
            sup_l = [[m[0, s_pick[i]].clone() for i, m in enumerate(s_labs[0])]]
            qs, ls = [], []
            for j in range(k):
                z = random.randint(q_edge[j], q_edge[j + 1] - 1)
                q, lab = q_img[:, z].clone(), q_lab[:, z].clone()
                if self.cfg["do_intaug"] and np.random.randint(2, size=1).item():
                    q = torch.from_numpy(gamma_transform(q.numpy(), self.cfg.get("gamma_range", [0.5, 1.5])))
                q, lab = random_transform(q[None, ...], lab)
                qs.append(q[0])
                ls.append(lab)
            qry_i = torch.cat(qs, dim=0).unsqueeze(1).expand(-1, 3, -1, -1)
            qry_l = torch.cat(ls, dim=0)
            order = np.arange(k)
            np.random.shuffle(order)
            qry_i, qry_l = qry_i[order], qry_l[order]
            sup_i, sup_l = [[sup_i[0][0][order]]], [[sup_l[0][0][order]]]
        else:
            test_shot = self.cfg.get("test_shot", self.cfg["n_shot"])
            qry_i = q_img.permute(1, 0, 2, 3).contiguous().expand(-1, 3, -1, -1)
            qry_l = q_lab[0]
            for i in range(len(s_imgs[0])):                       # the LAST support volume wins (:523-546)
                shots_i, shots_l = [], []
                for m in range(test_shot):
                    im, lb = [], []
                    for j in range(k):
                        n = int(q_edge[j + 1] - q_edge[j])
                        z = s_pick[i][j + (0 if j + m >= k else m)]
                        im.append(s_imgs[0][i][:, [z]].expand(n, 3, -1, -1))
                        lb.append(s_labs[0][i][0, [z]].expand(n, -1, -1))
                    shots_i.append(torch.cat(im, dim=0).unsqueeze(0))
                    shots_l.append(torch.cat(lb, dim=0).unsqueeze(0))
                shots_i, shots_l = torch.cat(shots_i, dim=0), torch.cat(shots_l, dim=0)
            sup_i, sup_l = [shots_i], [shots_l]

        sup_i, sup_l, qry_i, qry_l = make_support_query_same_size(sup_i, sup_l, qry_i, qry_l)

        if self.cfg.get("use_registration_loss", False):
            field, reg_pred, warped_src, aff_pred, aff_src = get_registration_field(
                qry_i, sup_i, sup_l, do_deformable=self.cfg.get("do_deformable", True))
            if self.cfg.get("use_registration_mask", False):
                sup_i[0][0] = torch.cat((sup_i[0][0], sup_l[0][0][:, None, ...]), dim=1)
                qry_i = torch.cat((qry_i, reg_pred), dim=1)
        else:
            field = reg_pred = aff_pred = None
            warped_src = aff_src = sup_i[0][0].numpy()
        return {"support_images": sup_i, "support_labels": sup_l, "query_images": qry_i, "query_labels": qry_l,
                "class_id": vol["class_id"], "registration_field": field,
                "support_images_3D": vol["support_images"], "support_labels_3D": vol["support_labels"],
                "query_images_3D": vol["query_images"], "query_labels_3D": vol["query_labels"],
                "warped_supp": torch.from_numpy(warped_src), "warped_supp_label": reg_pred,
                "affine_warped_supp": torch.from_numpy(aff_src), "affine_warped_supp_label": aff_pred,
                "pid": vol["pid"], "supp_pids": vol["supp_pids"]}


class FewshotRegReader(torch.utils.data.Dataset):
    """few_shot_reader.py:592-650 (needs `use_registration_loss: True`, like the reference): the support side is the
    AFFINE-warped support slice and label, `appr_query_labels` the fully warped label."""

    def __init__(self, data_dir, set_name, config, mode="train"):
        self.config, self.mode = config, mode
        self.fewshot_reader = FewshotSliceReader(data_dir, set_name, config, mode=mode)

    def __len__(self):
        return len(self.fewshot_reader)

    def __getitem__(self, idx):
        d = self.fewshot_reader[idx]
        if d["registration_field"] is None:
            raise TypeError("FewshotRegReader needs use_registration_loss: True (the reference iterates over "
                            "registration_field, few_shot_reader.py:601-602)")
        S, H, W = d["query_labels"].shape
        return {"support_images": [[d["affine_warped_supp"].unsqueeze(1)]],
                "support_labels": [[d["affine_warped_supp_label"][:, 0, ...]]],
                "query_images": d["query_images"][:, [0], ...], "query_labels": d["query_labels"],
                "appr_query_labels": (d["warped_supp_label"][:, 0, ...] > 0.5).float(),
                "class_id": d["class_id"], "registration_field": d["registration_field"],
                "support_images_3D": d["support_images_3D"], "support_labels_3D": d["support_labels_3D"],
                "query_images_3D": d["query_images_3D"], "query_labels_3D": d["query_labels_3D"],
                "grid": compute_grid((H, W)).repeat(S, 1, 1, 1),
                "original_support_images": d["support_images"], "original_support_labels": d["support_labels"],
                "warped_supp": d["warped_supp"], "pid": d["pid"], "supp_pids": d["supp_pids"]}


def train_collate(batch):
    return batch[0]


# ------------------------------------------------------------------------------------------ synthetic NRRD data set
def write_synthetic_dataset(root, n_volumes=3, classes=("Liver",), shape=(22, 44, 40), seed=0):
    """A tiny data set in the reference's on-disk layout (for tests and demos): `<pid>_clean.nrrd` int16 HU volumes
    [D,H,W], `<pid>_<roi>.nrrd` uint8 masks, `<root>/split/all.csv` (one pid per line) and
    `<root>/split/classes/<roi>.csv` (pid, z_start, z_end).  Returns (data_dir, set_name, class_csv_dir)."""
    rng = np.random.default_rng(seed)
    data_dir, split = os.path.join(root, "data"), os.path.join(root, "split")
    os.makedirs(data_dir, exist_ok=True)
    os.makedirs(os.path.join(split, "classes"), exist_ok=True)
    D, H, W = shape
    zz, yy, xx = np.meshgrid(np.arange(D), np.arange(H), np.arange(W), indexing="ij")
    pids = [f"case{v:03d}" for v in range(n_volumes)]
    rows = {roi: [] for roi in classes}
    for pid in pids:
        body = ((yy - H / 2) / (0.45 * H)) ** 2 + ((xx - W / 2) / (0.42 * W)) ** 2 < 1
        vol = np.where(body, 40.0, -1000.0) + rng.normal(0, 12, shape)
        for r, roi in enumerate(classes):
            cz, cy, cx = D * rng.uniform(0.4, 0.6), H * rng.uniform(0.4, 0.6), W * rng.uniform(0.35, 0.65)
            rz, ry, rx = D * rng.uniform(0.25, 0.35), H * rng.uniform(0.15, 0.25), W * rng.uniform(0.15, 0.25)
            organ = ((zz - cz) / rz) ** 2 + ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1
            vol = np.where(organ, 110.0 + 25 * r + rng.normal(0, 8, shape), vol)
            nrrd.write(os.path.join(data_dir, f"{pid}_{roi}.nrrd"), organ.astype(np.uint8))
            zs = np.nonzero(organ.any(axis=(1, 2)))[0]
            rows[roi].append((pid, int(zs.min()), int(zs.max())))
        vol[rng.random(shape) > 0.9995] = 3500.0                        # a few metal-like outliers above the HU window
        nrrd.write(os.path.join(data_dir, f"{pid}_clean.nrrd"), np.clip(vol, -1024, 4000).astype(np.int16))
    set_name = os.path.join(split, "all.csv")
    with open(set_name, "w") as f:
        f.write("\n".join(pids) + "\n")
    for roi in classes:
        with open(os.path.join(split, "classes", f"{roi}.csv"), "w") as f:
            f.write("pid,z_start,z_end\n" + "".join(f"{p},{a},{b}\n" for p, a, b in rows[roi]))
    return data_dir, set_name, os.path.join(split, "classes")
