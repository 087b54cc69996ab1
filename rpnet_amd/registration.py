"""The per-slice registration pre-step of the reference's reader (dataset/few_shot_reader.py:109-198,
net/registration.py:316-357,474-502) for the shipped configuration `do_deformable: False`
(yamls/example.yml:99-101), batched: all slices of a volume are registered in ONE kernel launch
(rpnet_affine_register: a block per slice runs the 50 Adam steps on-chip), then warped in six more.

The reference does this slice by slice with ~20 small torch operators per Adam step — on the CPU in this
configuration (few_shot_reader.py:135-143) — and it dominates the wall-clock of real evaluation outside the
model.  No CPU fallback: the oracle (oracle/registration_oracle.py) is test infrastructure.
"""
import torch

from . import hip
from .hip import call, ptr

ADAM = dict(lr=0.01, beta1=0.9, beta2=0.999, eps=1e-8)     # torch.optim.Adam defaults, lr of few_shot_reader.py:147


def base_grid(n, device):
    """the 1-D base grid of F.affine_grid(align_corners=False), from the function the reference itself calls
    (so that coordinates that sit exactly on pixel centres round the same way, csrc/registration.hip)"""
    return (torch.linspace(-1, 1, n) * (n - 1) / n).to(device)


def affine_register(moving, fixed, iters=50):
    """moving, fixed [S,H,W] fp32 on the GPU, values in [0,1] -> (theta [S,2,3], final MSE [S])."""
    hip.require_gpu(moving, fixed)
    S, H, W = moving.shape
    theta = torch.empty((S, 2, 3), device=moving.device, dtype=torch.float32)
    loss = torch.empty((S,), device=moving.device, dtype=torch.float32)
    xs, ys = base_grid(W, moving.device), base_grid(H, moving.device)
    call("rpnet_affine_register", ptr(moving.contiguous()), ptr(fixed.contiguous()), ptr(xs), ptr(ys), ptr(theta), ptr(loss), S, H, W, iters,
         ADAM["lr"], ADAM["beta1"], ADAM["beta2"], ADAM["eps"])
    return theta, loss


def affine_warp(x, theta, threshold=-1.0, scale=1.0, shift=0.0):
    S, H, W = x.shape
    out = torch.empty_like(x)
    xs, ys = base_grid(W, x.device), base_grid(H, x.device)
    call("rpnet_affine_warp", ptr(x.contiguous()), ptr(theta), ptr(xs), ptr(ys), ptr(out), S, H, W, threshold, scale, shift)
    return out


def identity_grid_warp(x, threshold=-1.0, scale=1.0, shift=0.0):
    S, H, W = x.shape
    out = torch.empty_like(x)
    call("rpnet_identity_grid_warp", ptr(x.contiguous()), ptr(out), S, H, W, threshold, scale, shift)
    return out


def get_registration_field(query_images, support_images, support_labels, do_deformable=False, device="cuda:0"):
    """Signature and return tuple of dataset/few_shot_reader.py:109 (registration_field, py_reg_pred,
    warped_src_list, py_affine_reg_pred, affine_warped_src_list).  query_images [S,1,H,W] in [-1,1];
    support_images [[ [S,1,H,W] ]]; support_labels [[ [S,H,W] ]].  `registration_field` holds the per-slice affine
    matrices [S,2,3] (the reference returns [module, grid] pairs that nothing downstream reads:
    RP_Net.forward ignores its registration_field argument, net/rp_net.py:226).  Outputs are CPU tensors / numpy
    arrays like the reference's."""
    if do_deformable:
        raise NotImplementedError("do_deformable=True (50 demons iterations per slice, net/registration.py:291-313) is not "
                                  "built on MI355X; the shipped configuration is do_deformable: False (yamls/example.yml:101)")
    src = ((support_images[0][0][:, 0].float() + 1) / 2.0).to(device)
    dst = ((query_images[:, 0].float() + 1) / 2.0).to(device)
    lab = support_labels[0][0].float().to(device)
    theta, _ = affine_register(src, dst)
    aw_lab, aw_src = affine_warp(lab, theta), affine_warp(src, theta)
    py_reg_pred = identity_grid_warp(aw_lab, threshold=0.1)[:, None].cpu()
    warped_src = identity_grid_warp(aw_src, scale=2.0, shift=-1.0).cpu().numpy()
    py_affine_reg_pred = affine_warp(lab, theta, threshold=0.1)[:, None].cpu()
    affine_warped_src = affine_warp(src, theta, scale=2.0, shift=-1.0).cpu().numpy()
    return theta.cpu(), py_reg_pred, warped_src, py_affine_reg_pred, affine_warped_src
