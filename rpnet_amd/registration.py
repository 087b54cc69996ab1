"""The per-slice registration pre-step of the reference's reader (dataset/few_shot_reader.py:109-198,
net/registration.py:316-357,474-502), batched over the slices of a volume.  Affine stage (the whole pre-step in the
shipped configuration `do_deformable: False`, yamls/example.yml:99-101): ONE kernel launch registers all slices
(rpnet_affine_register: a block per slice runs the 50 Adam steps on-chip), six more warp them.  Deformable stage
(`do_deformable: True`, the reader's default when the key is absent): rpnet_demons_register advances the dense flow
fields of all slices together, 24 launches per Adam step enqueued from C (csrc/demons.hip).

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


def gaussian_kernel_2d(sigma=(2.0, 2.0)):
    """the smoothing kernel of GaussianRegulariser (net/registration.py:14-49,106-135): normalised 1-D Gaussians of
    2 ceil(2 sigma) + 1 taps, their outer product renormalised, in fp32"""
    import numpy as np

    def k1(s):
        n = int(2 * np.ceil(s * 2) + 1)
        x = np.linspace(-(n - 1) // 2, (n - 1) // 2, num=n)
        k = 1.0 / (s * np.sqrt(2 * np.pi)) * np.exp(-(x ** 2) / (2 * s ** 2))
        return k / np.sum(k)
    k = np.tensordot(k1(sigma[0]), k1(sigma[1]), 0)
    return torch.tensor(k / np.sum(k), dtype=torch.float32)


def demons_register(moving, fixed, iters=50, sigma=(2.0, 2.0)):
    """moving (affine-warped), fixed [S,H,W] fp32 on the GPU in [0,1] -> (flow [S,2,H,W], displacement [S,2,H,W] of the
    final flow, NCC [S] at the last evaluated flow)."""
    hip.require_gpu(moving, fixed)
    S, H, W = moving.shape
    kern = gaussian_kernel_2d(sigma)
    if kern.shape[0] != kern.shape[1]:
        raise NotImplementedError("anisotropic smoothing kernel sizes")
    kern = kern.to(moving.device)
    flow = torch.empty((S, 2, H, W), device=moving.device, dtype=torch.float32)
    disp, loss = torch.empty_like(flow), torch.empty((S,), device=moving.device, dtype=torch.float32)
    wb = hip.query("rpnet_demons_workspace_bytes", S, H, W)
    ws = torch.empty((wb,), device=moving.device, dtype=torch.uint8)
    call("rpnet_demons_register", ptr(moving.contiguous()), ptr(fixed.contiguous()), ptr(kern), kern.shape[0], ptr(flow), ptr(disp),
         ptr(loss), S, H, W, iters, ADAM["lr"], ADAM["beta1"], ADAM["beta2"], ADAM["eps"], ptr(ws), wb)
    return flow, disp, loss


def displacement_warp(x, disp, threshold=-1.0, scale=1.0, shift=0.0):
    S, H, W = x.shape
    out = torch.empty_like(x)
    call("rpnet_displacement_warp", ptr(x.contiguous()), ptr(disp), ptr(out), S, H, W, threshold, scale, shift)
    return out


def get_registration_field(query_images, support_images, support_labels, do_deformable=True, device="cuda:0"):
    """Signature and return tuple of dataset/few_shot_reader.py:109 (registration_field, py_reg_pred,
    warped_src_list, py_affine_reg_pred, affine_warped_src_list).  query_images [S,1,H,W] in [-1,1];
    support_images [[ [S,1,H,W] ]]; support_labels [[ [S,H,W] ]].  `registration_field` holds the per-slice affine
    matrices [S,2,3] — with do_deformable=True the pair (thetas, flows [S,2,H,W]) — where the reference returns
    [module, grid] pairs that nothing downstream reads (RP_Net.forward ignores its registration_field argument,
    net/rp_net.py:226).  Outputs are CPU tensors / numpy arrays like the reference's."""
    src = ((support_images[0][0][:, 0].float() + 1) / 2.0).to(device)
    dst = ((query_images[:, 0].float() + 1) / 2.0).to(device)
    lab = support_labels[0][0].float().to(device)
    theta, _ = affine_register(src, dst)
    aw_lab, aw_src = affine_warp(lab, theta), affine_warp(src, theta)
    if do_deformable:
        # few_shot_reader.py:135-143,152-161: 50 demons steps on the affine-warped source (net/registration.py:489-502)
        flow, disp, _ = demons_register(aw_src, dst)
        py_reg_pred = displacement_warp(aw_lab, disp, threshold=0.1)[:, None].cpu()
        warped_src = displacement_warp(aw_src, disp, scale=2.0, shift=-1.0).cpu().numpy()
        py_affine_reg_pred = affine_warp(lab, theta, threshold=0.1)[:, None].cpu()
        affine_warped_src = affine_warp(src, theta, scale=2.0, shift=-1.0).cpu().numpy()
        return (theta.cpu(), flow.cpu()), py_reg_pred, warped_src, py_affine_reg_pred, affine_warped_src
    py_reg_pred = identity_grid_warp(aw_lab, threshold=0.1)[:, None].cpu()
    warped_src = identity_grid_warp(aw_src, scale=2.0, shift=-1.0).cpu().numpy()
    py_affine_reg_pred = affine_warp(lab, theta, threshold=0.1)[:, None].cpu()
    affine_warped_src = affine_warp(src, theta, scale=2.0, shift=-1.0).cpu().numpy()
    return theta.cpu(), py_reg_pred, warped_src, py_affine_reg_pred, affine_warped_src
