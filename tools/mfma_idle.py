"""Where the matrix pipe idles inside a training step: from a rocprofv3 rocpd database (--kernel-trace) of bench.py in its
default mode (weight gradients on side streams), the union of the intervals in which at least one MFMA-bound kernel
(convolution forward / dgrad, weight gradient, local correlation) is running, per step; the rest of the step's span is
"MFMA-idle" and is attributed to the kernels that ran then.  Steps are cut at the first-layer forward kernel
(conv1_fwd_kernel / conv1_fwd4_kernel: once per step).
Usage: python tools/mfma_idle.py trace.db [out.txt]"""
import collections, re, sqlite3, sys

db = sys.argv[1]
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
pick = lambda *names: next(n for n in names if n in cols)  # noqa: E731
cs, ce = pick("start", "start_ns", "begin"), pick("end", "end_ns", "stop")
rows = c.execute(f"select name, {cs}, {ce} from kernels order by {cs}").fetchall()
GEMM = re.compile(r"conv_igemm|conv_wgrad9|conv_wgrad1_split|local_corr_mfma|conv_igemm_split|conv_up4_dma|conv_wgrad_up4_kernel")
short = lambda n: re.sub(r"\(.*", "", n).replace("void ", "").replace("rpnet::", "")  # noqa: E731
starts = [i for i, r in enumerate(rows) if "conv1_fwd_kernel" in r[0] or "conv1_fwd4_kernel" in r[0]]
print(f"{len(rows)} kernel records, {len(starts)} steps (columns: {cols})", file=out)
for si in range(max(0, len(starts) - 4), len(starts) - 1):       # the last full steps
    seg = rows[starts[si]:starts[si + 1]]
    t0, t1 = seg[0][1], max(r[2] for r in seg)
    gem = sorted((r[1], r[2]) for r in seg if GEMM.search(r[0]))
    merged = []
    for a, b in gem:
        if merged and a <= merged[-1][1]:
            merged[-1][1] = max(merged[-1][1], b)
        else:
            merged.append([a, b])
    busy = sum(b - a for a, b in merged)
    # first backward kernel: the loss gradient
    tb = next((r[1] for r in seg if "dice_ce_bwd" in r[0]), t1)
    gaps = [(t0, merged[0][0])] + [(merged[i][1], merged[i + 1][0]) for i in range(len(merged) - 1)] + [(merged[-1][1], t1)]
    idle_f = sum(min(b, tb) - a for a, b in gaps if a < tb and b > a)
    idle_b = sum(b - max(a, tb) for a, b in gaps if b > tb and b > a)
    att = collections.Counter()
    none = 0
    for a, b in gaps:
        if b <= a:
            continue
        covered = 0
        for n, s, e in seg:
            if e <= a or s >= b or GEMM.search(n):
                continue
            att[short(n)] += min(e, b) - max(s, a)
        # time with no kernel at all: the gap minus the union of the others
        oth = sorted((max(s, a), min(e, b)) for n, s, e in seg if e > a and s < b and not GEMM.search(n))
        cur = a
        for s, e in oth:
            if s > cur:
                none += s - cur
            cur = max(cur, e)
        if cur < b:
            none += b - cur
    print(f"step {si}: span {(t1 - t0) / 1e6:.3f} ms (forward {(tb - t0) / 1e6:.3f}), MFMA-bound kernels running {busy / 1e6:.3f} ms, "
          f"idle {(t1 - t0 - busy) / 1e6:.3f} ms = forward {idle_f / 1e6:.3f} + backward {idle_b / 1e6:.3f}; no kernel at all {none / 1e6:.3f} ms",
          file=out)
    for n, v in att.most_common(14):
        print(f"    {v / 1e6:7.3f} ms  {n}", file=out)
    if si == len(starts) - 3:       # one step in detail: the longest MFMA-idle gaps, what ran in them and the GEMM in front
        big = sorted(((b - a, a, b) for a, b in gaps if b > a), reverse=True)[:28]
        for ln, a, b in sorted(big, key=lambda g: g[1]):
            inside = [short(n) for n, s_, e_ in seg if e_ > a and s_ < b and not GEMM.search(n)]
            before = next((short(n) for n, s_, e_ in reversed(seg) if GEMM.search(n) and e_ <= a + 1), "-")
            cnt = collections.Counter(inside)
            print(f"    gap at {(a - t0) / 1e6:7.3f} ms, {ln / 1e3:6.1f} us ({'bwd' if a >= tb else 'fwd'}) after {before[:40]}: "
                  + ", ".join(f"{k} x{v}" if v > 1 else k for k, v in cnt.most_common(6)), file=out)
            if ln > 400e3:          # a long one: every kernel in it, in order (start offset inside the gap, duration)
                for n, s_, e_ in seg:
                    if e_ > a and s_ < b and not GEMM.search(n):
                        print(f"        +{(s_ - a) / 1e3:7.1f} us  {(e_ - s_) / 1e3:6.1f} us  {short(n)[:90]}", file=out)
