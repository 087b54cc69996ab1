"""Timeline of ONE training step from a rocprofv3 rocpd database (--kernel-trace): every kernel with its start offset, duration and
stream, the step cut at the first-layer forward kernel; per stream the busy time of the forward and backward halves.
Usage: python tools/step_timeline.py trace.db [out.txt] [step index from the end, default 2]"""
import collections, re, sqlite3, sys

db = sys.argv[1]
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
back = int(sys.argv[3]) if len(sys.argv) > 3 else 2
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
pick = lambda *names: next(n for n in names if n in cols)  # noqa: E731
cs, ce = pick("start", "start_ns", "begin"), pick("end", "end_ns", "stop")
sid = pick("stream_id", "stream", "queue_id")
rows = c.execute(f"select name, {cs}, {ce}, {sid} from kernels order by {cs}").fetchall()
short = lambda n: re.sub(r"\(.*", "", n).replace("void ", "").replace("rpnet::", "")[:70]  # noqa: E731
starts = [i for i, r in enumerate(rows) if "conv1_fwd_kernel" in r[0] or "conv1_fwd4_kernel" in r[0]]
seg = rows[starts[-1 - back]:starts[-back]]
t0 = seg[0][1]
tb = next((r[1] for r in seg if "dice_ce_bwd" in r[0]), t0)
streams = sorted({r[3] for r in seg})
print(f"step span {(max(r[2] for r in seg) - t0) / 1e3:.1f} us, forward {(tb - t0) / 1e3:.1f} us, {len(seg)} kernels, streams {streams}", file=out)
for half, lo, hi in (("forward", t0, tb), ("backward", tb, max(r[2] for r in seg))):
    for s in streams:
        iv = sorted((max(r[1], lo), min(r[2], hi)) for r in seg if r[3] == s and r[2] > lo and r[1] < hi)
        busy, cur = 0, lo
        for a, b in iv:
            a = max(a, cur)
            if b > a:
                busy += b - a
                cur = b
        print(f"  {half:8s} stream {s}: {len(iv):4d} kernels, busy {busy / 1e3:8.1f} us of {(hi - lo) / 1e3:8.1f}", file=out)
print("   start_us     dur_us  stream  kernel", file=out)
for n, s, e, st in seg:
    print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:10.1f}  {str(st):>6s}  {'B ' if s >= tb else 'F '}{short(n)}", file=out)
