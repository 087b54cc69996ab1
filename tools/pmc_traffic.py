"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs) into per-kernel HBM
traffic per launch.  Units/corrections per /opt/skills/guides/MI355X_MICROARCH.md §HBM:
counter values are KiB; on gfx950 FETCH_SIZE reads exactly half of a wide coalesced stream
(16 B/lane) -> the read side is doubled; WRITE_SIZE is taken as is (uncalibrated)."""
import collections, csv, json, sys
fetch_csv, write_csv, out = sys.argv[1], sys.argv[2], sys.argv[3]
def agg(path, name):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != name: continue
        k = r["Kernel_Name"].split("(")[0]
        d[k][0] += 1; d[k][1] += float(r["Counter_Value"])
    return d
f, w = agg(fetch_csv, "FETCH_SIZE"), agg(write_csv, "WRITE_SIZE")
res = {}
for k in sorted(set(f) | set(w), key=lambda k: -(f.get(k, [0, 0])[1] * 2 + w.get(k, [0, 0])[1])):
    nf, vf = f.get(k, [0, 0.0]); nw, vw = w.get(k, [0, 0.0])
    res[k] = {"launches": max(nf, nw), "fetch_bytes_per_launch": 2.0 * vf * 1024 / max(nf, 1),
              "write_bytes_per_launch": vw * 1024 / max(nw, 1)}
json.dump(res, open(out, "w"), indent=1)
tot = sum((v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"]) * v["launches"] for v in res.values())
print(f"{len(res)} kernels, total HBM traffic over the run {tot/1e9:.2f} GB")
for k, v in list(res.items())[:12]:
    print(f"{k[:70]:70s} x{v['launches']:4d} fetch {v['fetch_bytes_per_launch']/1e6:9.2f} MB write {v['write_bytes_per_launch']/1e6:9.2f} MB")
