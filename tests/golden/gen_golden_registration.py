#!/usr/bin/env python3
"""Golden vectors of the registration pre-step from the REFERENCE itself.

Runs ONLY in the build container (needs /root/reference).  Imports net/registration.py and
dataset/few_shot_reader.py of uci-cbcl/RP-Net with the absent third-party modules stubbed (cv2, torchviz, nrrd,
nibabel, torchvision, SimpleITK: none is touched on this path), calls the reference's own
get_registration_field(query_images, support_images, support_labels, do_deformable=False) — the shipped
configuration (yamls/example.yml:99-101), which runs on the CPU in the reference — on seeded synthetic slices,
checks oracle/registration_oracle.py against it and writes tests/golden/registration.npz
(inputs by seed, expected thetas / warped labels / warped sources).

    python tests/golden/gen_golden_registration.py
"""
import importlib.machinery
import importlib.util
import os
import sys
import types
import unittest.mock as mock

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))


def stub(name):
    m = mock.MagicMock()
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__name__ = name
    sys.modules[name] = m


for _m in ["cv2", "torchviz", "nrrd", "nibabel", "torchvision", "torchvision.transforms", "SimpleITK", "pydicom",
           "skimage", "skimage.measure", "utils", "utils.util"]:
    stub(_m)
import matplotlib  # noqa: E402

matplotlib.use("Agg")
spec = importlib.util.spec_from_file_location("refreg", "/root/reference/net/registration.py")
refreg = importlib.util.module_from_spec(spec)
sys.modules["refreg"] = refreg
spec.loader.exec_module(refreg)
_net = types.ModuleType("net")
_net.registration = refreg
sys.modules["net"] = _net
sys.modules["net.registration"] = refreg
_pkg = types.ModuleType("refdata")
_pkg.__path__ = ["/root/reference/dataset"]
sys.modules["refdata"] = _pkg
stub("refdata.brain_reader")
spec = importlib.util.spec_from_file_location("refdata.few_shot_reader", "/root/reference/dataset/few_shot_reader.py")
fsr = importlib.util.module_from_spec(spec)
sys.modules["refdata.few_shot_reader"] = fsr
spec.loader.exec_module(fsr)

import numpy as np  # noqa: E402
import torch  # noqa: E402

# the repo's own modules, loaded by path (the name `net` is taken by the reference here)
for _name, _path in [("rp_synth", "rpnet_amd/utils/synth.py"), ("rp_regoracle", "oracle/registration_oracle.py")]:
    spec = importlib.util.spec_from_file_location(_name, os.path.join(REPO, _path))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_name] = mod
    spec.loader.exec_module(mod)
synth, RO = sys.modules["rp_synth"], sys.modules["rp_regoracle"]
torch.set_num_threads(8)

CASES = [("s64", 77, 3, 64), ("s128", 78, 2, 128), ("s96x", 79, 2, 96)]   # tag, seed, slices, size
out = {}
for tag, seed, S, size in CASES:
    ep = synth.make_episode(seed, S, size)
    supp = [[torch.from_numpy(ep["support_images"][0][0])]]
    lab = [[torch.from_numpy(ep["support_fg"][0][0])]]
    qry = torch.from_numpy(ep["query_images"])
    rf, reg_pred, warped_src, aff_pred, aff_src = fsr.get_registration_field(qry, supp, lab, do_deformable=False)
    thetas = torch.stack([r.affine_reg.theta.data[0] for r, _ in rf])
    o_th, o_reg, o_wsrc, o_areg, o_asrc = RO.get_registration_field(qry, supp, lab)
    e_th = (o_th - thetas).abs().max().item()
    e_src = max((o_wsrc - torch.from_numpy(warped_src)).abs().max().item(),
                (o_asrc - torch.from_numpy(aff_src)).abs().max().item())
    flips = int((o_reg != reg_pred).sum() + (o_areg != aff_pred).sum())
    print(f"{tag}: oracle vs reference: theta {e_th:.2e}, warped sources {e_src:.2e}, label flips {flips}")
    assert e_th < 1e-6 and e_src < 1e-5 and flips == 0, "oracle/registration_oracle.py does not reproduce the reference"
    out.update({f"{tag}_dims": np.array([seed, S, size]), f"{tag}_theta": thetas.numpy(),
                f"{tag}_base_grid": (torch.linspace(-1, 1, size) * (size - 1) / size).numpy(),   # F.affine_grid's, on this CPU
                f"{tag}_reg_pred": reg_pred.numpy().astype(np.uint8), f"{tag}_warped_src": np.asarray(warped_src, np.float32),
                f"{tag}_aff_pred": aff_pred.numpy().astype(np.uint8), f"{tag}_aff_src": np.asarray(aff_src, np.float32)})
np.savez_compressed(os.path.join(HERE, "registration.npz"), **out)
print("wrote registration.npz", os.path.getsize(os.path.join(HERE, "registration.npz")) / 1e6, "MB")
