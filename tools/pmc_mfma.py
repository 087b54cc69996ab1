"""Per-kernel MFMA-busy fraction and effective clock from a rocprofv3 PMC pass
(SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE).  MFMA busy = busy cycles / (1024 SIMDs x kernel cycles);
GRBM_GUI_ACTIVE is summed over the 8 XCDs, so kernel cycles = GUI_ACTIVE / 8 (= 2.4 GHz x duration)."""
import collections, csv, json, sys
src, out = sys.argv[1], sys.argv[2]
d = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
dur = collections.defaultdict(float)
seen = set()
for r in csv.DictReader(open(src)):
    k = r["Kernel_Name"].split("(")[0]
    d[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r["Dispatch_Id"])
    if key not in seen:
        seen.add(key); n[k] += 1; dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
res = {}
for k, c in d.items():
    cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    if cyc <= 0: continue
    res[k] = {"launches": n[k], "avg_us": dur[k] / n[k] / 1e3, "mfma_busy": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * cyc),
              "clock_ghz": cyc / dur[k] if dur[k] else 0.0}
json.dump(res, open(out, "w"), indent=1)
for k, v in sorted(res.items(), key=lambda kv: -kv[1]["avg_us"] * kv[1]["launches"])[:8]:
    print(f"{k[:64]:64s} x{v['launches']:4d} avg {v['avg_us']:8.1f} us  MFMA busy {100*v['mfma_busy']:5.1f}%  clock {v['clock_ghz']:.2f} GHz")
