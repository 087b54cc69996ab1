"""The conditioning of RP-Net's parameter gradients, measured on the CPU oracle itself (no GPU, no HIP code): the
yardstick behind the gradient tolerances of tests/test_gpu_model.py.

d loss / d encoder-weights passes through ~20 ReLU switches, four max-pool arg-maxes and the align loss's arg-max; a
relative perturbation of the input images at fp32 round-off level (2e-6) moves those gradients by 1e-4 .. 1e-2 relative,
and the fp32 oracle differs from the same computation in fp64 by about as much, while the smooth tail (cre.q) moves at
the perturbation's own size.  Element-wise gradient agreement with the reference beyond that level is therefore not a
property any fp32 implementation has — the reference's own CPU and GPU runs included."""
import torch

from tests.helpers import episode_tensors, load_cfg, oracle_step

ENC = ("encoder.Conv1.conv.3.weight", "encoder.Conv3.conv.0.weight", "encoder.Conv5.conv.0.weight", "encoder.Up_conv4.conv.3.weight")
SMOOTH = "cre.q.0.weight"


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def test_gradient_conditioning_of_the_oracle():
    cfg = load_cfg(2)
    inputs, _ = episode_tensors(4242, 2, 64)
    g32, l32, _ = oracle_step(cfg, inputs)
    g32p, l32p, _ = oracle_step(cfg, inputs, input_scale=1.0 + 2e-6)
    g64, l64, _ = oracle_step(cfg, inputs, dtype=torch.float64)
    assert abs(l32.item() - l64.item()) < 1e-5 * abs(l64.item())           # the forward is well conditioned ...
    moved = {n: _rel(g32p[n], g32[n]) for n in ENC + (SMOOTH,)}
    vs64 = {n: _rel(g32[n], g64[n]) for n in ENC + (SMOOTH,)}
    print("moved by a 2e-6 input perturbation:", moved)
    print("fp32 oracle vs fp64 oracle:        ", vs64)
    assert moved[SMOOTH] < 5e-5 and vs64[SMOOTH] < 5e-5                     # ... and so is the smooth tail of the backward
    # the encoder gradients are not: 50x .. 5000x amplification of the perturbation (and of fp32 round-off)
    assert max(moved[n] for n in ENC) > 1e-4 and max(vs64[n] for n in ENC) > 1e-4
    assert all(moved[n] < 5e-2 and vs64[n] < 5e-2 for n in ENC)
