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
    clk = cyc / dur[k] if dur[k] else 0.0
    # box-independent figures first: MFMA busy cycles per launch (SQ_VALU_MFMA_BUSY_CYCLES counts cycles: 32 per 32x32x16 MFMA) and per
    # SIMD; the percentage and the clock are DERIVED from GRBM_GUI_ACTIVE / 8 / duration, which breaks for kernels of a few tens of
    # microseconds (round 4's file showed 3.2 - 7.9 GHz for them): a clock above the chip's 2.4 GHz is refused, not printed
    ok = 0.0 < clk <= 2.45
    res[k] = {"launches": n[k], "avg_us": dur[k] / n[k] / 1e3,
              "mfma_busy_cycles_per_launch": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / n[k],
              "mfma_busy_cycles_per_simd_per_launch": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / n[k] / 1024.0,
              "mfma_busy": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * cyc) if ok else None,
              "clock_ghz": clk if ok else None,
              "clock_note": None if ok else f"GUI_ACTIVE / 8 / duration = {clk:.2f} GHz is not a clock (short kernel): percentage withheld"}
json.dump(res, open(out, "w"), indent=1)
for k, v in sorted(res.items(), key=lambda kv: -kv[1]["avg_us"] * kv[1]["launches"])[:8]:
    pct = f"{100 * v['mfma_busy']:5.1f}% at {v['clock_ghz']:.2f} GHz" if v["mfma_busy"] is not None else "  (clock not derivable)"
    print(f"{k[:64]:64s} x{v['launches']:4d} avg {v['avg_us']:8.1f} us  MFMA busy cycles / SIMD / launch {v['mfma_busy_cycles_per_simd_per_launch']:10.0f}  = {pct}")
