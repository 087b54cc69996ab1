"""Shared helpers for the parity tests (not product code)."""
import numpy as np
import torch
import yaml

from rpnet_amd.utils.synth import make_episode

EXAMPLE_YAML = __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.dirname(
    __import__("os").path.abspath(__file__))), "yamls", "example.yml")


def load_cfg(T=None):
    cfg = yaml.load(open(EXAMPLE_YAML), Loader=yaml.FullLoader)
    if T is not None:
        cfg["n_iter_refinement"] = T
    return cfg


def episode_tensors(seed, B, size, device="cpu", n_shots=1, n_ways=1):
    ep = make_episode(seed, B, size, n_shots=n_shots, n_ways=n_ways)
    t = lambda a: torch.from_numpy(a).to(device)  # noqa: E731
    return ([[t(s) for s in way] for way in ep["support_images"]], [[t(s) for s in way] for way in ep["support_fg"]],
            [[t(s) for s in way] for way in ep["support_bg"]], [t(ep["query_images"])], t(ep["query_labels"]),
            t(ep["appr_query_labels"])), ep


def in_checksum(ep):
    return np.array([float(ep["query_images"].astype(np.float64).sum()),
                     float(ep["support_images"][0][0].astype(np.float64).sum()),
                     float(ep["appr_query_labels"].sum()), float(ep["support_fg"][0][0].sum())])


def rnd(seed, *shape):
    return torch.from_numpy(np.random.RandomState(seed).standard_normal(shape).astype(np.float32))


def rel_err(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def rel_l2(a, b):
    """|| a - b ||_2 / || b ||_2 — element-wise agreement in the aggregate (rel_err is max-abs over max-abs)"""
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-300)).item()


def oracle_step(cfg, inputs, dtype=torch.float32, input_scale=1.0, noise=None, device=None):
    """One forward + backward of the oracle (train mode, align loss on, the harness loss) in `dtype` on the seeded
    parameters; `input_scale` multiplies the images, `noise` = (seed, eps) multiplies every image pixel by
    1 + eps * u, u uniform in [-1, 1] (conditioning experiments).  Returns ({name: grad}, loss, out), on the CPU.
    device (default: the CPU): where torch executes the oracle's operators.  The oracle is plain PyTorch, so the float64
    yardsticks of the GPU suite may run it through torch's OWN device kernels (rocBLAS / aten, nothing of librpnet_hip.so): the
    same independent checker, minutes faster than on the host cores (round 4: 620 of the suite's 848 s were these runs)."""
    from oracle import rpnet_oracle as O
    import contextlib
    si, fg, bg, qi, ql, appr = inputs
    dv = torch.device(device) if device is not None else torch.device("cpu")
    P = {}
    for k, v in O.seeded_params(cfg["mask_refinement_correlation_radius"], requires_grad=True,
                                mask_feature_map=cfg.get("mask_feature_map", False)).items():
        t = v.detach().to(dtype) if v.is_floating_point() else v.detach().clone()
        P[k] = t.to(dv).clone().requires_grad_(v.requires_grad)
    c = lambda t: t.to(dtype).to(dv)  # noqa: E731
    if noise is not None:
        gen = torch.Generator().manual_seed(noise[0])
        jit = lambda t: (t.to(dtype) * (1.0 + noise[1] * (2.0 * torch.rand(t.shape, generator=gen, dtype=torch.float64) - 1.0)).to(dtype)).to(dv)  # noqa: E731
    else:
        jit = lambda t: c(t) * input_scale  # noqa: E731
    # the inputs are made OUTSIDE the context (the noise comes from a CPU generator)
    si_d, qi_d = [[jit(s) for s in w] for w in si], [jit(qi[0])]
    fg_d, bg_d, appr_d = [[c(s) for s in w] for w in fg], [[c(s) for s in w] for w in bg], c(appr)
    # (inside the context torch's factory functions — the arange / eye / linspace calls of the oracle — create on `dv`)
    with (torch.device(dv) if dv.type != "cpu" else contextlib.nullcontext()):
        out = O.rp_net_forward(P, cfg, si_d, fg_d, bg_d, qi_d, appr_d, True, align=True)
        loss = O.total_loss(out, ql.to(dv), cfg["align_loss_scaler"])
        loss.backward()
    if dv.type != "cpu":
        out = {k: (v.detach().cpu() if torch.is_tensor(v) else {i: t.detach().cpu() for i, t in v.items()} if isinstance(v, dict) else v)
               for k, v in out.items()}
    return {k: v.grad.cpu() for k, v in P.items() if v.requires_grad and v.grad is not None}, loss.detach().cpu(), out
