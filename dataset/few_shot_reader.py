"""`dataset.few_shot_reader.FewshotRegReader` with the item contract test_rpnet.py consumes
(test_rpnet.py:70,166-184; reference dataset/few_shot_reader.py:592-650).

Two sources behind the one name:
  * `data_dir` holds `<pid>_clean.nrrd` / `<pid>_<roi>.nrrd` volumes (the reference's layout, few_shot_reader.py:314-321)
    -> rpnet_amd.utils.volume_reader.FewshotRegReader, the real volume / slice / registration readers;
  * otherwise (the private ABD-110 data is not available offline) SYNTHETIC volumes with the same keys, shapes and
    dtypes (rpnet_amd.utils.synth), so the evaluation loop runs end to end on the MI355X path.
The per-slice registration pre-step of the reference reader (few_shot_reader.py:556-566:
`use_registration_loss`, `do_deformable: False`) is `get_registration_field` = rpnet_amd.registration (one HIP
launch for all slices); on a GPU box the reader derives `appr_query_labels` from it exactly as the reference does
(:608), elsewhere it serves the generator's approximate labels.
"""
import os

import numpy as np
import torch

from rpnet_amd.registration import get_registration_field  # noqa: F401  (reference few_shot_reader.py:109, HIP path)
from rpnet_amd.utils.synth import make_episode
from rpnet_amd.utils import volume_reader as _vr
from rpnet_amd.utils.volume_reader import (FewshotSliceReader, FewshotVolumeReader, crop, elastic_transform_all,  # noqa: F401
                                           gamma_tansform, gamma_tansform_with_label, keep_only_annotation_z_slices,
                                           make_support_query_same_size, random_label_transform, random_transform,
                                           train_collate)


def _has_nrrd(data_dir):
    return bool(data_dir) and os.path.isdir(data_dir) and any(f.endswith(".nrrd") for f in os.listdir(data_dir))


class _VolumeInfo:
    def __init__(self, classes, n_vol):
        self.data_info = [[{"pid": f"synthetic_{c}_{i:03d}"} for i in range(n_vol)] for c in range(len(classes))]


class _SliceReader:
    def __init__(self, classes, n_vol):
        self.fewshot_volume_reader = _VolumeInfo(classes, n_vol)


class FewshotRegReader(torch.utils.data.Dataset):
    def __new__(cls, data_dir=None, set_name=None, config=None, mode="train", **kw):
        if _has_nrrd(data_dir):
            return _vr.FewshotRegReader(data_dir, set_name, config, mode=mode)      # not a subclass: __init__ below is skipped
        return super().__new__(cls)

    def __init__(self, data_dir, set_name, config, mode="train", n_volumes=4, n_slices=6, size=256):
        self.config, self.mode = config, mode
        self.classes = config.get("eval_classes" if mode == "eval" else "train_classes", ["Liver"])
        self.n_volumes, self.n_slices, self.size = n_volumes, n_slices, size
        self.fewshot_reader = _SliceReader(self.classes, n_volumes)

    def __len__(self):
        return len(self.classes) * self.n_volumes

    def __getitem__(self, idx):
        if idx >= len(self):
            raise IndexError(idx)
        class_id, vol = divmod(idx, self.n_volumes)
        S, H = self.n_slices, self.size
        ep = make_episode(7000 + idx, S, H)
        t = torch.from_numpy
        supp_img, supp_lab = t(ep["support_images"][0][0]), t(ep["support_fg"][0][0])
        ys, xs = torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, H), indexing="ij")
        grid = torch.stack([xs, ys], -1)[None].repeat(S, 1, 1, 1)
        supp_idx = (vol + 1) % self.n_volumes
        item = {"support_images": [[supp_img]], "support_labels": [[supp_lab]], "warped_supp": supp_img[:, 0],
                "query_images": t(ep["query_images"]), "query_labels": t(ep["query_labels"]),
                "appr_query_labels": t(ep["appr_query_labels"]), "grid": grid, "class_id": class_id,
                "pid": self.fewshot_reader.fewshot_volume_reader.data_info[class_id][vol]["pid"],
                "supp_pids": [(class_id, supp_idx)], "registration_field": None}
        if self.config.get("use_registration_loss", False) and torch.cuda.is_available():
            # reference few_shot_reader.py:556-566,582,608: warp the support label onto the query slice by slice
            field, reg_pred, warped_src, aff_pred, aff_src = get_registration_field(
                item["query_images"], item["support_images"], item["support_labels"],
                do_deformable=self.config.get("do_deformable", True))
            item.update({"registration_field": field, "warped_supp": torch.from_numpy(warped_src),
                         "warped_supp_label": reg_pred, "affine_warped_supp": torch.from_numpy(aff_src),
                         "affine_warped_supp_label": aff_pred, "appr_query_labels": (reg_pred[:, 0] > 0.5).float()})
        return item
