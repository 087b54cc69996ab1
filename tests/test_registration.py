"""Registration pre-step (dataset/few_shot_reader.py:109-198 with do_deformable=False).
CPU: the oracle (oracle/registration_oracle.py) against the golden vectors the reference's own
get_registration_field produced (tests/golden/registration.npz, gen_golden_registration.py).
GPU: the HIP path (rpnet_amd/registration.py -> rpnet_affine_register / rpnet_affine_warp /
rpnet_identity_grid_warp through the C ABI) against the same vectors and the oracle."""
import numpy as np
import pytest
import torch

from rpnet_amd.utils.synth import make_episode

CASES = ["s64", "s128", "s96x"]


def _inputs(g, tag):
    seed, S, size = (int(v) for v in g[f"{tag}_dims"])
    ep = make_episode(seed, S, size)
    t = torch.from_numpy
    return [[t(ep["support_images"][0][0])]], [[t(ep["support_fg"][0][0])]], t(ep["query_images"])


@pytest.mark.parametrize("tag", CASES[:2])
def test_oracle_matches_reference_golden(golden, tag):
    """bit-level agreement (1e-6) on the CPU the vectors were made on (the build container, recognised by its base
    grid AND the presence of the reference tree); on another CPU model the reference itself only reproduces to ~1e-3 in
    theta (first Adam step = lr * sign of a near-zero gradient at a bilinear kink, see the GPU test below), so there
    the oracle is held to that spread."""
    import os
    from oracle import registration_oracle as RO
    g = golden("registration")
    supp, lab, qry = _inputs(g, tag)
    th, reg, wsrc, areg, asrc = RO.get_registration_field(qry, supp, lab)
    n = qry.shape[-1]
    home = os.path.isdir("/root/reference") and np.array_equal((torch.linspace(-1, 1, n) * (n - 1) / n).numpy(), g[f"{tag}_base_grid"])
    th_tol, src_tol, flip_tol = (1e-6, 1e-5, 0.0) if home else (2e-3, 2e-2, 0.01)
    assert np.abs(th.numpy() - g[f"{tag}_theta"]).max() < th_tol
    assert np.abs(wsrc.numpy() - g[f"{tag}_warped_src"]).max() < src_tol
    assert np.abs(asrc.numpy() - g[f"{tag}_aff_src"]).max() < src_tol
    assert (reg.numpy().astype(np.uint8) != g[f"{tag}_reg_pred"]).mean() <= flip_tol
    assert (areg.numpy().astype(np.uint8) != g[f"{tag}_aff_pred"]).mean() <= flip_tol


def test_reader_exports_and_no_cpu_fallback():
    import dataset.few_shot_reader as fsr
    assert callable(fsr.get_registration_field)
    if not torch.cuda.is_available():
        for deform in (False, True):
            with pytest.raises((RuntimeError, AssertionError)):
                fsr.get_registration_field(torch.zeros(1, 1, 8, 8), [[torch.zeros(1, 1, 8, 8)]], [[torch.zeros(1, 8, 8)]],
                                           do_deformable=deform)


DCASES = ["d64", "d96"]


@pytest.mark.parametrize("tag", DCASES[:1])
def test_oracle_demons_matches_reference_golden(golden, tag):
    """do_deformable: True — the oracle's demons stage against the reference's own classes (composed on the CPU by
    gen_golden_registration.py: the reference hard-codes cuda:0 for this branch).  Bit-level on the CPU the vectors
    were made on; elsewhere the optimisation's own sensitivity (a 1-ulp change moves the flow by ~1e-3 of a [-1,1]
    coordinate, measured) sets the bar."""
    import os
    from oracle import registration_oracle as RO
    g = golden("registration_demons")
    supp, lab, qry = _inputs(g, tag)
    th, fl, reg, wsrc, areg, asrc = RO.get_registration_field_deformable(qry, supp, lab)
    n = qry.shape[-1]
    home = os.path.isdir("/root/reference") and np.array_equal((torch.linspace(-1, 1, n) * (n - 1) / n).numpy(), g[f"{tag}_base_grid"])
    fl_tol, src_tol, flip_tol = (1e-6, 1e-5, 0.0) if home else (2e-2, 5e-2, 0.02)
    assert np.abs(fl.numpy() - g[f"{tag}_flow"]).max() < fl_tol
    assert np.abs(wsrc.numpy() - g[f"{tag}_warped_src"]).max() < src_tol
    assert (reg[:, 0].numpy().astype(np.uint8) != g[f"{tag}_reg_pred"]).mean() <= flip_tol


@pytest.mark.gpu
@pytest.mark.parametrize("tag", CASES)
def test_hip_registration_vs_reference_golden(golden, tag):
    """theta after 50 Adam steps within 1e-3 of the reference's (measured 2e-8 on s64, 8e-6 on s96x and 4e-4 on s128,
    the ill-conditioned case below, where the CPU oracle itself moves by 2e-4 between host CPU models; fp32 reduction
    order is the only difference), warped sources within 2e-2 of a [-1,1] image (a theta difference of 4e-4 is 0.03 px at
    128^2, times an edge of contrast up to 2; measured 6.7e-3 there, <= 1e-5 on the well-conditioned cases), thresholded labels differ in < 0.2 %
    of the pixels (a bilinear value within rounding of the 0.1 threshold).

    The optimisation starts at a kink of the bilinear interpolant (theta = identity puts every sample on a pixel
    centre) and its first step is lr * sign(gradient), so the result depends on how torch.linspace rounds the base
    grid on the HOST CPU (one column in 128 lands below its centre), and the reference is only reproducible to
    ~1e-3 in theta across CPU models even with the same grid (measured: EPYC 9575F vs the build container).  The HIP
    path takes the grid from the same library call and is deterministic; it is held to the golden vectors (1e-3)
    whenever this host's base grid equals the one they were made with, and to the oracle evaluated on this host
    with the reference's own cross-CPU spread (2e-3 in theta, 1 % of the label pixels)."""
    from oracle import registration_oracle as RO
    from rpnet_amd.registration import base_grid, get_registration_field
    g = golden("registration")
    supp, lab, qry = _inputs(g, tag)
    th, reg, wsrc, areg, asrc = get_registration_field(qry, supp, lab, do_deformable=False)
    n = reg.numel()
    if np.array_equal(base_grid(qry.shape[-1], "cpu").numpy(), g[f"{tag}_base_grid"]):
        assert np.abs(th.numpy() - g[f"{tag}_theta"]).max() < 1e-3
        assert np.abs(wsrc - g[f"{tag}_warped_src"]).max() < 2e-2 and np.abs(asrc - g[f"{tag}_aff_src"]).max() < 2e-2
        assert (reg.numpy().astype(np.uint8) != g[f"{tag}_reg_pred"]).sum() <= 2e-3 * n
        assert (areg.numpy().astype(np.uint8) != g[f"{tag}_aff_pred"]).sum() <= 2e-3 * n
    o_th, o_reg, o_wsrc, o_areg, o_asrc = RO.get_registration_field(qry, supp, lab)
    assert (th - o_th).abs().max() < 2e-3
    assert (reg != o_reg).sum() <= 1e-2 * n and (areg != o_areg).sum() <= 1e-2 * n


@pytest.mark.gpu
def test_hip_registration_pieces_vs_oracle():
    """the three entry points one by one on a non-square, odd-sized slice pair; empty label; zero iterations"""
    from oracle import registration_oracle as RO
    from rpnet_amd import registration as R
    g = torch.Generator().manual_seed(5)
    S, H, W = 2, 37, 52
    mov, fix = torch.rand(S, H, W, generator=g), torch.rand(S, H, W, generator=g)
    fix = 0.5 * fix + 0.5 * torch.roll(mov, (2, -3), (1, 2))
    th, loss = R.affine_register(mov.cuda(), fix.cuda())
    for s in range(S):
        ref = RO.affine_register(mov[s][None, None], fix[s][None, None])
        assert (th[s].cpu() - ref[0]).abs().max() < 2e-4
    theta = torch.tensor([[[0.9, 0.1, 0.05], [-0.08, 1.1, -0.02]], [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]]])
    out = R.affine_warp(mov.cuda(), theta.cuda()).cpu()
    out2 = R.identity_grid_warp(mov.cuda()).cpu()
    for s in range(S):
        assert (out[s] - RO.affine_warp(mov[s][None, None], theta[s][None])[0, 0]).abs().max() < 1e-5
        assert (out2[s] - RO.identity_grid_warp(mov[s][None, None])[0, 0]).abs().max() < 1e-5
    th0, _ = R.affine_register(mov.cuda(), fix.cuda(), iters=0)
    assert (th0.cpu() - torch.tensor([[1.0, 0, 0], [0, 1.0, 0]])).abs().max() == 0
    zero = torch.zeros(1, H, W).cuda()
    assert R.affine_warp(zero, theta[:1].cuda(), threshold=0.1).abs().max() == 0


@pytest.mark.gpu
def test_reader_uses_registration_on_gpu():
    """FewshotRegReader with use_registration_loss: appr_query_labels = warped support label > 0.5 (reference :608)"""
    from dataset.few_shot_reader import FewshotRegReader
    cfg = {"use_registration_loss": True, "do_deformable": False, "eval_classes": ["Liver"]}
    item = FewshotRegReader("/nonexistent", "test", cfg, mode="eval", n_volumes=2, n_slices=3, size=64)[0]
    assert item["appr_query_labels"].shape == item["query_labels"].shape
    assert item["warped_supp_label"].shape == (3, 1, 64, 64) and item["registration_field"].shape == (3, 2, 3)
    assert set(np.unique(item["appr_query_labels"].numpy())) <= {0.0, 1.0}


@pytest.mark.gpu
@pytest.mark.parametrize("tag", DCASES)
def test_hip_demons_stage_vs_oracle(golden, tag):
    """the demons stage alone, HIP vs oracle from the SAME affine-warped input (so that the comparison is not dominated
    by the affine stage's conditioning): flow after 1 and 2 steps within 5e-6 of a field of magnitude 0.01-0.02
    (measured 2e-7 - 8e-7: hand-derived backward of the ten-fold scaling-and-squaring chain, fp64 NCC moments, atomics
    order); after 50 steps within 5e-3 of a field of magnitude 0.03 (measured 9e-8 ... 6e-5 in most runs, above 1e-3 in
    about one run in ten: the order of the fp32 atomics changes and the optimisation amplifies rounding — one different
    ulp moved the CPU oracle's own flow by 9e-4 — as it does between two runs of the reference's own CUDA backward);
    NCC at the last evaluated flow within 1e-3; displacement and warp against the oracle's on the HIP flow."""
    from oracle import registration_oracle as RO
    from rpnet_amd import registration as R
    g = golden("registration_demons")
    supp, lab, qry = _inputs(g, tag)
    src, dst = ((supp[0][0][:, 0] + 1) / 2)[:1], ((qry[:, 0] + 1) / 2)[:1]
    mov = RO.affine_warp(src[None], RO.affine_register(src[None], dst[None]))
    h, w = src.shape[-2:]
    grid_t = RO.compute_grid(h, w).permute(0, 2, 3, 1).contiguous()
    for iters in (1, 2, 50):
        of = RO.demons_register(mov, dst[None], iters=iters)
        hf, hd, hl = R.demons_register(mov[0].cuda(), dst.cuda(), iters=iters)
        assert (hf.cpu() - of).abs().max() < (5e-6 if iters <= 2 else 5e-3), (tag, iters)
        assert of.abs().max() > 5e-3
    od = RO.diffeomorphic(hf.cpu(), grid_t)                   # scaling and squaring + warp of the HIP flow itself
    assert (hd.cpu() - od).abs().max() < 2e-6
    ow = RO.displacement_warp(mov, od, grid_t)
    assert (R.displacement_warp(mov[0].cuda(), hd).cpu() - ow[0]).abs().max() < 1e-5
    # NCC of the last evaluated flow = the one before the 50th update
    o49 = RO.demons_register(mov, dst[None], iters=49)
    l49 = RO.ncc(RO.displacement_warp(mov, RO.diffeomorphic(o49, grid_t), grid_t), dst[None])
    assert abs(hl.item() - l49.item()) < 1e-3 and hl.item() < -0.9
    f0, d0, _ = R.demons_register(mov[0].cuda(), dst.cuda(), iters=0)
    assert f0.abs().max() == 0 and d0.abs().max() == 0


@pytest.mark.gpu
@pytest.mark.parametrize("tag", DCASES)
def test_hip_deformable_registration_vs_reference_golden(golden, tag):
    """get_registration_field(do_deformable=True) end to end against the reference's classes: the affine stage's theta
    within 1e-3 / 2e-3 (same / other host CPU, see above); its rounding-level differences are amplified by the 50
    demons steps (measured: 1e-7 in theta -> 7e-4 in the flow), so the flow is held to 5e-3 of a field of magnitude 0.03
    on the golden vectors' host grid, the warped source to 1e-2, the thresholded labels to 1 % of the pixels."""
    from rpnet_amd.registration import base_grid, get_registration_field
    g = golden("registration_demons")
    supp, lab, qry = _inputs(g, tag)
    (th, fl), reg, wsrc, areg, asrc = get_registration_field(qry, supp, lab, do_deformable=True)
    assert tuple(fl.shape) == tuple(g[f"{tag}_flow"].shape) and reg.shape == (qry.shape[0], 1) + tuple(qry.shape[-2:])
    same = np.array_equal(base_grid(qry.shape[-1], "cpu").numpy(), g[f"{tag}_base_grid"])
    th_tol, fl_tol, src_tol = (1e-3, 5e-3, 1e-2) if same else (2e-3, 3e-2, 5e-2)
    assert np.abs(th.numpy() - g[f"{tag}_theta"]).max() < th_tol
    assert np.abs(fl.numpy() - g[f"{tag}_flow"]).max() < fl_tol
    assert np.abs(wsrc - g[f"{tag}_warped_src"]).max() < src_tol and np.abs(asrc - g[f"{tag}_aff_src"]).max() < src_tol
    assert (reg[:, 0].numpy().astype(np.uint8) != g[f"{tag}_reg_pred"]).mean() < 0.01
    assert (areg[:, 0].numpy().astype(np.uint8) != g[f"{tag}_aff_pred"]).mean() < 0.01


@pytest.mark.gpu
def test_reader_with_deformable_registration():
    from dataset.few_shot_reader import FewshotRegReader
    cfg = {"use_registration_loss": True, "do_deformable": True, "eval_classes": ["Liver"]}
    item = FewshotRegReader("/nonexistent", "test", cfg, mode="eval", n_volumes=2, n_slices=3, size=64)[0]
    theta, flow = item["registration_field"]
    assert theta.shape == (3, 2, 3) and flow.shape == (3, 2, 64, 64) and flow.abs().max() > 1e-3
    assert item["appr_query_labels"].shape == item["query_labels"].shape
    assert set(np.unique(item["appr_query_labels"].numpy())) <= {0.0, 1.0}
