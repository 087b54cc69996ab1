"""Summarise a rocprofv3 rocpd database (--kernel-trace --stats) into the kernel-stats CSV
that is committed under profiles/ (the .db itself is scratch under gpurun_out/)."""
import sqlite3, sys
db, out = sys.argv[1], sys.argv[2]
c = sqlite3.connect(db)
rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows)
with open(out, "w") as f:
    f.write("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage\n")
    for n, k, s, a, mn, mx in rows:
        f.write(f"\"{n}\",{k},{s},{a:.1f},{mn},{mx},{100.0 * s / tot:.3f}\n")
print(f"{len(rows)} kernels, total {tot / 1e6:.2f} ms -> {out}")
