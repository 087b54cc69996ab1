"""Diagnostic: which intermediate tensor of the backward pass differs first between two runs of the same step
(RF._TAPS: clones in stream order, no host synchronisation).   [RPNET_BN_LDS=big] python tools/diag_taps.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import episode_tensors, load_cfg  # noqa: E402
from tests.test_gpu_model import build, total_loss  # noqa: E402
import rpnet_amd.functional as RF  # noqa: E402
import rpnet_amd.modules as RM  # noqa: E402
from rpnet_amd.parallel import FlatGradBucket  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
RM._F16_MIN_PIXELS = 0
RM._CRE_STREAMS_TRAIN = os.environ.get("DIAG_CRE", "0") == "1"
RM._ENC_STREAMS = int(os.environ.get("DIAG_ENC", "0"))
ASYNCW = os.environ.get("DIAG_ASYNC", "1") == "1"
RF._TAPS_PIN = os.environ.get("DIAG_PIN", "1") == "1"
cfg = load_cfg(2)
(si, fg, bg, qi, ql, appr), _ = episode_tensors(91, 4, 128, "cuda:0", n_shots=1, n_ways=2)


def run():
    net = build(cfg, True)
    bucket = FlatGradBucket(net) if ASYNCW else None
    RF.set_async_wgrad(ASYNCW)
    if bucket is not None:
        bucket.zero()
    RF._TAPS = []
    out = net(si, fg, bg, qi, appr_query_labels=appr)
    total_loss(out, ql, 1.0).backward()
    if bucket is not None:
        bucket.allreduce()
    torch.cuda.synchronize()
    taps, RF._TAPS = RF._TAPS, None
    return taps, {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}


ref_t, ref_g = run()
print(f"{len(ref_t)} taps per step")
for r in range(reps):
    t, g = run()
    badp = [n for n in ref_g if not torch.equal(ref_g[n], g[n])]
    first = None
    nbad = 0
    for i, ((tag, shp, a), (tag2, shp2, b)) in enumerate(zip(ref_t, t)):
        if tag.endswith("ws4"):
            continue                      # the workspace has regions nobody writes
        if tag != tag2 or shp != shp2:
            first = first or (i, "ORDER", tag, tag2)
            break
        if a.dtype in (torch.float16, torch.bfloat16):
            same = torch.equal(a.view(torch.int16), b.view(torch.int16))
        else:
            same = torch.equal(a, b)
        if not same:
            nbad += 1
            if first is None:
                d = (a.float() - b.float()).abs()
                first = (i, tag, shp, f"{int((d > 0).sum())} of {a.numel()} elements differ, max {float(d.max()):.3e} (ref max {float(a.float().abs().max()):.3e})",
                         "first index", int(torch.nonzero(d.flatten() != 0)[0]) if bool((d != 0).any()) else -1)
    print(f"rep {r}: params differing {len(badp)} (last {badp[-1] if badp else '-'}); taps differing {nbad}; first: {first}", flush=True)
    if first is not None and first[1] != "ORDER":
        a, b = ref_t[first[0]][2], t[first[0]][2]
        if a.dtype in (torch.float16, torch.bfloat16):
            a, b = a.view(torch.float16), b.view(torch.float16)
        idx = torch.nonzero((a.flatten().view(torch.int16) if a.dtype == torch.float16 else a.flatten()) !=
                            (b.flatten().view(torch.int16) if b.dtype == torch.float16 else b.flatten())).flatten().tolist()
        shp = tuple(a.shape)
        print("       shape", shp, "differing flat indices:", idx[:40])
        # the window of the first differing element, recomputed on the host from the (pinned, equal) inputs of the pass
        i0 = first[0]
        tags = {ref_t[j][0]: ref_t[j][2] for j in range(max(0, i0 - 6), i0 + 1) if ref_t[j][1] == ref_t[i0][1]}
        yv, st, dzv = tags.get("bwd_in:y,stats,dz0"), tags.get("bwd_in:y,stats,dz1"), tags.get("bwd_in:y,stats,dz2")
        if yv is not None and len(shp) == 5 and yv.shape[1] == 2 * dzv.shape[1]:
            k = idx[0] % (a.numel() // shp[0])
            c = k % shp[4]; xx = (k // shp[4]) % shp[3]; yy = (k // (shp[4] * shp[3])) % shp[2]; n = k // (shp[4] * shp[3] * shp[2])
            oy, ox = yy // 2, xx // 2
            g = n // (shp[1] // st.shape[1])
            sc, sh = float(st[0, g, c]), float(st[1, g, c])
            win = [float(yv[n, 2 * oy + q // 2, 2 * ox + q % 2, c]) for q in range(4)]
            aff = [torch.tensor(v, dtype=torch.float32) * torch.tensor(sc) + torch.tensor(sh) for v in win]
            print(f"        window n={n} oy={oy} ox={ox} c={c} group={g}: scale {sc:.6g} shift {sc and sh:.6g}; y {win}; y*scale+shift {[float(v) for v in aff]}; "
                  f"dz {float(dzv[n, oy, ox, c]):.6g}")
            for q in range(4):
                kk = ((n * shp[2] + 2 * oy + q // 2) * shp[3] + 2 * ox + q % 2) * shp[4] + c
                print(f"          q={q}: ref plane0 {float(a.flatten()[kk]):.6g}  got plane0 {float(b.flatten()[kk]):.6g}")
        for k in idx[:12]:
            un = []
            rem = k
            for dim in reversed(shp):
                un.append(rem % dim)
                rem //= dim
            print("        ", k, tuple(reversed(un)), "ref", float(a.flatten()[k]), "got", float(b.flatten()[k]))
    if first is not None and first[1] != "ORDER":
        # context: the taps just before
        i = first[0]
        for j in range(max(0, i - 6), i + 1):
            print("      ", j, ref_t[j][0], ref_t[j][1])
