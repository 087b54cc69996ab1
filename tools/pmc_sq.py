"""Per-kernel wave-cycle breakdown from a rocprofv3 PMC pass (SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
SQ_WAIT_INST_LDS SQ_BUSY_CYCLES ...): WAIT_ANY = wave parked (s_waitcnt / barrier), WAIT_INST_ANY = issue stall (MFMA RAW /
pipe busy), ACTIVE_INST_ANY = issuing; the three are disjoint and add up to ~WAVE_CYCLES (MI355X_MICROARCH.md, PMC slots)."""
import collections, csv, json, sys
src, out = sys.argv[1], sys.argv[2]
d = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter(); seen = set()
for r in csv.DictReader(open(src)):
    k = r["Kernel_Name"].split("(")[0]
    d[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"]); n[k] += 1
res = {}
for k, c in d.items():
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    if wc <= 0: continue
    res[k] = {"launches": n[k], **{name: v / wc for name, v in c.items() if name != "SQ_WAVE_CYCLES"}, "SQ_WAVE_CYCLES_per_launch": wc / n[k]}
json.dump(res, open(out, "w"), indent=1)
for k, v in sorted(res.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES_per_launch"] * kv[1]["launches"])[:6]:
    print(k[:60], {a: round(b, 3) for a, b in v.items()})
