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


# ------------------------------------------------------------------------- do_deformable: True (the demons stage)
# The reference's get_registration_field hard-codes device "cuda:0" for this branch (few_shot_reader.py:137-143); the
# container has no GPU, so the same call sequence (:133-180) is made here on the reference's own classes with
# use_GPU=False — AffineDemonsRegistration(use_diffeomorphic=True), two Adam(lr 0.01), GaussianRegulariser sigma [2,2],
# iters [50, 50] — a composed oracle in the sense of SURVEY.md §8c.
def reference_deformable(qry, supp, lab, iters=(50, 50)):
    src_all = (supp[0][0][:, 0].numpy() + 1) / 2.0
    dst_all = (qry[:, 0].numpy() + 1) / 2.0
    lab_all = lab[0][0].numpy()
    res = {k: [] for k in ("theta", "flow", "reg", "wsrc", "areg", "asrc")}
    for s in range(len(dst_all)):
        src = torch.from_numpy(src_all[s])[None, None]
        dst = torch.from_numpy(dst_all[s])[None, None]
        src_label = torch.from_numpy(lab_all[s])[None, None]
        size = src_all[s].shape
        registration = refreg.AffineDemonsRegistration(size, use_diffeomorphic=True, use_GPU=False, stop_shear=False)
        opts = [torch.optim.Adam(registration.affine_reg.parameters(), lr=0.01),
                torch.optim.Adam(registration.demons.parameters(), lr=0.01)]
        regulariser = refreg.GaussianRegulariser([1, 1], sigma=[2, 2], dtype=torch.float32, device="cpu")
        registration.train_registraion(src, dst, opts, regulariser=regulariser, iters=list(iters),
                                       regularise_displacement=False, verbose=False)
        grid = refreg.compute_grid(size)
        res["theta"].append(registration.affine_reg.theta.data[0].clone())
        res["flow"].append(registration.demons.flow.data[0].clone())
        res["reg"].append((registration(src_label, grid)[0, 0].data > 0.1).float())
        res["wsrc"].append(registration(src, grid)[0, 0].data * 2 - 1)
        res["areg"].append((registration.affine_reg(src_label)[0, 0].data > 0.1).float())
        res["asrc"].append(registration.affine_reg(src)[0, 0].data * 2 - 1)
    return {k: torch.stack(v) for k, v in res.items()}


DCASES = [("d64", 87, 2, 64), ("d96", 88, 1, 96)]
dout = {}
for tag, seed, S, size in DCASES:
    ep = synth.make_episode(seed, S, size)
    supp = [[torch.from_numpy(ep["support_images"][0][0])]]
    lab = [[torch.from_numpy(ep["support_fg"][0][0])]]
    qry = torch.from_numpy(ep["query_images"])
    ref = reference_deformable(qry, supp, lab)
    o_th, o_fl, o_reg, o_wsrc, o_areg, o_asrc = RO.get_registration_field_deformable(qry, supp, lab)
    e_fl = (o_fl - ref["flow"]).abs().max().item()
    e_src = (o_wsrc - ref["wsrc"]).abs().max().item()
    flips = int((o_reg[:, 0] != ref["reg"]).sum())
    print(f"{tag}: oracle vs reference (demons): theta {(o_th - ref['theta']).abs().max().item():.2e}, flow {e_fl:.2e} "
          f"(|flow| max {ref['flow'].abs().max().item():.3f}), warped source {e_src:.2e}, label flips {flips}")
    assert e_fl < 1e-6 and e_src < 1e-5 and flips == 0, "oracle demons stage does not reproduce the reference"
    dout.update({f"{tag}_dims": np.array([seed, S, size]), f"{tag}_theta": ref["theta"].numpy(), f"{tag}_flow": ref["flow"].numpy(),
                 f"{tag}_base_grid": (torch.linspace(-1, 1, size) * (size - 1) / size).numpy(),
                 f"{tag}_reg_pred": ref["reg"].numpy().astype(np.uint8), f"{tag}_warped_src": ref["wsrc"].numpy(),
                 f"{tag}_aff_pred": ref["areg"].numpy().astype(np.uint8), f"{tag}_aff_src": ref["asrc"].numpy()})
np.savez_compressed(os.path.join(HERE, "registration_demons.npz"), **dout)
print("wrote registration_demons.npz", os.path.getsize(os.path.join(HERE, "registration_demons.npz")) / 1e6, "MB")
