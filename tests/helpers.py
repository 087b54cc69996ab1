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
