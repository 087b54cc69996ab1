#!/usr/bin/env python3
"""Run the reference's UNMODIFIED evaluation driver on an MI355X box against this repository:

    python tools/run_reference_driver.py /path/to/RP-Net/test_rpnet.py --yaml yamls/example.yml

`net.model`, `dataset.few_shot_reader`, `utils.util`, `net.registration` then resolve to this repository (it is first on
sys.path; the reference's directory is NOT added), i.e. to the HIP path.  Two lines of the literal file stop it on a one-GPU
ROCm box before any model code runs; both are handled here without touching the file:

* test_rpnet.py:3   `os.environ['CUDA_VISIBLE_DEVICES'] = '1'` in front of `import torch`.  HIP honours the variable, so a
  one-GPU lease would enumerate ZERO devices.  Device enumeration happens once, at the first use of the runtime: this launcher
  initialises it BEFORE the script runs, so the later assignment changes nothing.  (On a multi-GPU box the driver therefore
  runs on device 0 — or on the device HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES select in the launcher's environment.)
* test_rpnet.py:27  `from torch.utils.tensorboard import SummaryWriter`.  tensorboard is not installed in this image (and there
  is no network): if the import fails, a `torch.utils.tensorboard` with a SummaryWriter that accepts every call and writes
  nothing is registered (the driver only logs scalars there; its results go to stdout / log_eval).
"""
import os
import runpy
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def prepare(require_gpu=True):
    """the two workarounds; returns the names of those that were needed"""
    import torch
    did = []
    if torch.cuda.is_available():
        torch.cuda.init()                      # enumerate now: a CUDA_VISIBLE_DEVICES assignment made later has no effect
        torch.zeros(1, device="cuda")
        did.append("hip runtime initialised before the script (test_rpnet.py:3 neutralised)")
    elif require_gpu:
        raise SystemExit("run_reference_driver.py: no MI355X visible (the HIP path has no CPU fallback)")
    try:
        from torch.utils.tensorboard import SummaryWriter  # noqa: F401
    except Exception:       # ModuleNotFoundError: tensorboard; or a broken install
        class SummaryWriter:
            """accepts every call of torch.utils.tensorboard.SummaryWriter, writes nothing"""

            def __init__(self, *a, **k):
                pass

            def __getattr__(self, name):
                return lambda *a, **k: None

        mod = types.ModuleType("torch.utils.tensorboard")
        mod.SummaryWriter = SummaryWriter
        sys.modules["torch.utils.tensorboard"] = mod
        import torch.utils
        torch.utils.tensorboard = mod
        did.append("no-op torch.utils.tensorboard.SummaryWriter registered (test_rpnet.py:27: tensorboard is not installed)")
    return did


def main():
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    script = os.path.abspath(sys.argv[1])
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    for line in prepare(require_gpu=os.environ.get("RPNET_DRIVER_ALLOW_NO_GPU", "0") != "1"):
        print("[run_reference_driver]", line, file=sys.stderr)
    sys.argv = [script] + sys.argv[2:]
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
