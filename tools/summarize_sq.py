"""Condense the rocprofv3 passes of tools/collect_sq.sh into profiles-sized summaries: per kernel (top kernels by time) the
average duration and the SQ / TCC counters per launch, plus derived ratios (occupancy, VALU share of wave cycles, active
lanes per VALU instruction, L2 hit rate, HBM-side bytes).  FETCH_SIZE x2 per the gfx950 note of MI355X_MICROARCH.md."""
import csv, glob, json, os, sys
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
out_dir, tag = sys.argv[1], sys.argv[2]
from kernel_names import short
res = defaultdict(dict)
stats = glob.glob(os.path.join(out_dir, "stats", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    agg = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(stats[0])):
        a = agg[short(r["Name"])]
        a[0] += int(r["Calls"]); a[1] += float(r["TotalDurationNs"])
    for k, (c, nsum) in agg.items():
        res[k]["calls"] = c; res[k]["avg_us"] = nsum / c / 1e3; res[k]["total_ms"] = nsum / 1e6
for f in glob.glob(os.path.join(out_dir, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for r in csv.DictReader(open(f)):
        a = acc[short(r["Kernel_Name"])][r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"])
        if "VGPR_Count" in r and r["VGPR_Count"]:
            res[short(r["Kernel_Name"])]["vgpr"] = int(r["VGPR_Count"]); res[short(r["Kernel_Name"])]["sgpr"] = int(r.get("SGPR_Count") or 0)
            res[short(r["Kernel_Name"])]["lds"] = int(r.get("LDS_Block_Size") or 0); res[short(r["Kernel_Name"])]["scratch"] = int(r.get("Scratch_Size") or 0)
    for k, d in acc.items():
        for cn, (n, v) in d.items():
            res[k][cn] = v / n
top = sorted((k for k in res if "total_ms" in res[k]), key=lambda k: -res[k]["total_ms"])[:12]
out = {"_note": "per launch averages; SQ_* cycle counters are quad-cycles summed over waves (MI355X_MICROARCH.md); FETCH_SIZE x2 (KB)"}
for k in top:
    d = res[k]
    g = lambda n: d.get(n, float("nan"))
    e = {kk: (round(v, 3) if isinstance(v, float) else v) for kk, v in d.items()}
    wc = g("SQ_WAVE_CYCLES")
    if wc == wc and wc > 0:
        e["valu_active_frac_of_wave_cycles"] = round(g("SQ_ACTIVE_INST_VALU") / wc, 4)
        e["wait_inst_any_frac"] = round(g("SQ_WAIT_INST_ANY") / wc, 4)
        e["wait_any_frac"] = round(g("SQ_WAIT_ANY") / wc, 4)
        e["active_inst_any_frac"] = round(g("SQ_ACTIVE_INST_ANY") / wc, 4)
        e["avg_waves_in_flight_per_simd"] = round(wc / max(g("SQ_BUSY_CYCLES"), 1) , 3)
        e["valu_insts_per_wave"] = round(g("SQ_INSTS_VALU") / max(g("SQ_WAVES"), 1), 1)
    if g("SQ_THREAD_CYCLES_VALU") == g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU") > 0:
        e["active_lanes_per_valu_cycle"] = round(g("SQ_THREAD_CYCLES_VALU") / g("SQ_ACTIVE_INST_VALU") / 4, 2)
    if g("TCC_HIT_sum") == g("TCC_HIT_sum"):
        e["l2_hit_rate"] = round(g("TCC_HIT_sum") / max(g("TCC_HIT_sum") + g("TCC_MISS_sum"), 1), 4)
    if g("FETCH_SIZE") == g("FETCH_SIZE"):
        e["fabric_read_bytes"] = g("FETCH_SIZE") * 1024 * 2
    if g("WRITE_SIZE") == g("WRITE_SIZE"):
        e["fabric_write_bytes"] = g("WRITE_SIZE") * 1024
    out[k] = e
dst = os.path.join(out_dir, f"{tag}_sq_summary.json")
json.dump(out, open(dst, "w"), indent=1)
print("wrote", dst)
for k in top[:6]:
    e = out[k]
    print(k, {kk: e[kk] for kk in ("avg_us", "vgpr", "scratch", "valu_active_frac_of_wave_cycles", "wait_any_frac", "wait_inst_any_frac", "avg_waves_in_flight_per_simd", "valu_insts_per_wave", "active_lanes_per_valu_cycle", "l2_hit_rate") if kk in e})
