"""Condense the rocprofv3 outputs of tools/collect_profiles.sh into the small files committed under profiles/:
   <tag>_kernel_stats.csv  (name, calls, total ms, average us, % of GPU time)
   <tag>_pmc_traffic.json  per kernel: launches, FETCH_SIZE / WRITE_SIZE per launch in bytes.  FETCH_SIZE is doubled
                           (gfx950 tallies 128-B read requests at 64 B: MI355X_MICROARCH.md, HBM / rocprofv3 section);
                           WRITE_SIZE is taken as reported; both are rocprofv3's KB (x 1024)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

out_dir, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(out_dir, "summary")
os.makedirs(dst, exist_ok=True)


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_names import short  # noqa: E402


stats = glob.glob(os.path.join(out_dir, "stats", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    agg = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(stats[0])):
        a = agg[short(r["Name"])]
        a[0] += int(r["Calls"])
        a[1] += float(r["TotalDurationNs"])
    total = sum(v[1] for v in agg.values())
    with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w") as f:
        f.write("kernel,calls,total_ms,avg_us,percent\n")
        for k, (c, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            f.write(f"{k},{c},{ns / 1e6:.3f},{ns / c / 1e3:.2f},{100 * ns / total:.2f}\n")

traffic = defaultdict(dict)
for which, scale in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
    files = glob.glob(os.path.join(out_dir, "pmc_" + which.split("_")[0].lower(), "**", "*counter_collection.csv"), recursive=True)
    if not files:
        continue
    acc = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(files[0])):
        if r.get("Counter_Name") != which:
            continue
        a = acc[short(r["Kernel_Name"])]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    for k, (n, v) in acc.items():
        traffic[k][which + "_bytes_per_launch"] = v / n * 1024.0 * scale
        traffic[k]["launches_" + which] = n
for k, d in traffic.items():
    if "FETCH_SIZE_bytes_per_launch" in d and "WRITE_SIZE_bytes_per_launch" in d:
        d["hbm_bytes_per_launch"] = d["FETCH_SIZE_bytes_per_launch"] + d["WRITE_SIZE_bytes_per_launch"]
if traffic:
    meta = {"_note": "FETCH_SIZE x2 (gfx950 correction), WRITE_SIZE as reported, KB -> bytes; separate --pmc passes of "
                     "`bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-stage-timing`"}
    meta.update({k: traffic[k] for k in sorted(traffic)})
    json.dump(meta, open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w"), indent=1)
if os.path.exists(os.path.join(out_dir, "bench.json")):
    txt = open(os.path.join(out_dir, "bench.json")).read().strip().splitlines()
    if txt:
        open(os.path.join(dst, f"{tag}_bench.json"), "w").write(txt[-1] + "\n")
print("summary in", dst, os.listdir(dst))
