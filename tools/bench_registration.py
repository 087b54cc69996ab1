"""Registration pre-step: slices/s of rpnet_amd.registration.get_registration_field on the
GPU (inputs resident in HBM, HIP events around the five launches) next to the CPU oracle (= the reference's
operator sequence for do_deformable: False) on a bounded sample of the same slices.  One JSON line.

    python tools/bench_registration.py [--slices 64] [--size 256] [--deformable]

--deformable adds a second JSON line for do_deformable: True (affine stage + 50 demons steps, rpnet_demons_register).
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpnet_amd import registration as R
from rpnet_amd.utils.synth import make_episode

ap = argparse.ArgumentParser()
ap.add_argument("--slices", type=int, default=64)
ap.add_argument("--size", type=int, default=256)
ap.add_argument("--cpu-slices", type=int, default=4)
ap.add_argument("--deformable", action="store_true")
args = ap.parse_args()
S, H = args.slices, args.size
ep = make_episode(4321, S, H)
t = torch.from_numpy
src = ((t(ep["support_images"][0][0])[:, 0] + 1) / 2).cuda()
dst = ((t(ep["query_images"])[:, 0] + 1) / 2).cuda()
lab = t(ep["support_fg"][0][0]).float().cuda()


def gpu_pass():
    theta, _ = R.affine_register(src, dst)
    aw_lab, aw_src = R.affine_warp(lab, theta), R.affine_warp(src, theta)
    return R.identity_grid_warp(aw_lab, threshold=0.1), R.identity_grid_warp(aw_src, scale=2.0, shift=-1.0)


for _ in range(2):
    gpu_pass()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5):
    gpu_pass()
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b) / 5
# the optimiser kernel alone
a.record()
for _ in range(5):
    R.affine_register(src, dst)
b.record()
torch.cuda.synchronize()
ms_reg = a.elapsed_time(b) / 5

from oracle import registration_oracle as RO
n = min(args.cpu_slices, S)
t0 = time.perf_counter()
RO.get_registration_field(t(ep["query_images"])[:n], [[t(ep["support_images"][0][0])[:n]]], [[t(ep["support_fg"][0][0])[:n]]])
cpu_s = (time.perf_counter() - t0) / n
gathers = 50.0 * 4 * H * H * S            # bilinear corner reads of the optimiser per pass
print(json.dumps({
    "metric": f"registration pre-step, slices/s ({H}x{H}, 50 Adam steps + 4 warps per slice, do_deformable: False)",
    "value": round(S / (ms * 1e-3), 1), "unit": "slices/s", "slices": S, "ms_per_pass": round(ms, 3),
    "affine_register_ms": round(ms_reg, 3), "blocks": S,
    "optimiser_gathers_per_s": round(gathers / (ms_reg * 1e-3) / 1e9, 1), "optimiser_gathers_unit": "G corner reads/s (L2-resident slices)",
    "cpu_baseline": {"value": round(1.0 / cpu_s, 3), "unit": "slices/s", "cores": torch.get_num_threads(), "kind": "port",
                     "sample": f"{n} slices through oracle/registration_oracle.py (the reference's operator sequence), {cpu_s:.2f} s/slice"},
    "gpu_over_cpu": round(S / (ms * 1e-3) * cpu_s, 1)}))

if args.deformable:
    def gpu_deformable():
        theta, _ = R.affine_register(src, dst)
        aw_lab, aw_src = R.affine_warp(lab, theta), R.affine_warp(src, theta)
        flow, disp, _ = R.demons_register(aw_src, dst)
        return R.displacement_warp(aw_lab, disp, threshold=0.1), R.displacement_warp(aw_src, disp, scale=2.0, shift=-1.0)

    gpu_deformable()
    torch.cuda.synchronize()
    a.record()
    for _ in range(3):
        gpu_deformable()
    b.record()
    torch.cuda.synchronize()
    ms_d = a.elapsed_time(b) / 3
    n = min(2, S)
    t0 = time.perf_counter()
    RO.get_registration_field_deformable(t(ep["query_images"])[:n], [[t(ep["support_images"][0][0])[:n]]], [[t(ep["support_fg"][0][0])[:n]]])
    cpu_d = (time.perf_counter() - t0) / n
    print(json.dumps({
        "metric": f"registration pre-step, slices/s ({H}x{H}, affine + 50 demons steps per slice, do_deformable: True)",
        "value": round(S / (ms_d * 1e-3), 1), "unit": "slices/s", "slices": S, "ms_per_pass": round(ms_d, 3),
        "launches_per_pass": 50 * 24 + 20, "note": "all slices advance together; backward scatter = fp32 atomics",
        "cpu_baseline": {"value": round(1.0 / cpu_d, 3), "unit": "slices/s", "cores": torch.get_num_threads(), "kind": "port",
                         "sample": f"{n} slices through the oracle's autograd restatement, {cpu_d:.2f} s/slice (the reference runs this branch on the GPU, slice by slice)"},
        "gpu_over_cpu": round(S / (ms_d * 1e-3) * cpu_d, 1)}))
