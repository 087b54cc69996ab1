"""Per-model execution options of the MI355X path (round 6): which arithmetic a model's convolutions run on, how its step is laid out
on HIP streams, how gradient fan-in is done and whether the masked CRE convolutions skip their zero tiles — as attributes of the
RP_Net INSTANCE (`net.schedule`), so that two models (or two threads) of one process can differ.

Every field defaults to None = "the process-wide default at call time" (what `rpnet_amd.functional.set_conv_math`,
`set_async_wgrad` and the module-level constants of rpnet_amd.modules / rpnet_amd.functional say: tests and the A/B tools keep
working on those).  A field that is set wins for this model only: RP_Net.forward activates it for the duration of the call
(functional.scope) and every autograd node remembers, at forward time, the options its backward needs (weight-gradient stream,
zero-tile skip) — the backward of model A does not read what model B's forward left behind.

Nothing here has a reference counterpart: /root/reference runs on one stream in one arithmetic (torch fp32)."""
import dataclasses
from typing import Optional


@dataclasses.dataclass
class Schedule:
    conv_math: Optional[str] = None            # "f16x2" | "bf16x3" | "f32" | "f16"  (functional._MODES)
    async_wgrad: Optional[bool] = None         # weight gradients on a side stream, straight into param.grad (a flat bucket)
    mask_skip: Optional[bool] = None           # zero-tile skip of w_k(x * mask) / w_q(x * (1 - mask)), forward + input gradient
    cre_streams_train: Optional[bool] = None   # training: the CRE's w_q branch on its own stream beside w_k
    cre_streams_eval: Optional[bool] = None    # inference on few slices: the same for eval-mode calls
    enc_streams: Optional[int] = None          # training: the encoder's support / query calls as two chains (0 never, 1 auto, 2 always)
    fanin: Optional[int] = None                # gradient fan-in level 0 / 1 / 2 (modules._FANIN)
    f16_min_pixels: Optional[int] = None       # f16x2: encoder input pixels per call from which the fp16 planes are used
    f16_min_pixels_eval: Optional[int] = None  # the same for eval-mode calls on measured scales

    def get(self, name, default):
        v = getattr(self, name)
        return default if v is None else v

    def overrides(self):
        return {f.name: getattr(self, f.name) for f in dataclasses.fields(self) if getattr(self, f.name) is not None}
