"""Merge one run's <tag>_pmc_traffic.json (tools/summarize_profiles.py) into profiles/pmc_traffic.json under its workload key,
so that bench.py only ever attaches counters to the workload they were collected on.
usage: python tools/merge_pmc.py <run_pmc_traffic.json> "<scene> <W>x<H> <spp>spp rebuild_blas=<0|1>" """
import json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles", "pmc_traffic.json")
run = json.load(open(sys.argv[1]))
key = sys.argv[2]
try:
    d = json.load(open(dst))
    if "workloads" not in d:
        d = {"workloads": {}}
except (OSError, ValueError):
    d = {"workloads": {}}
d["_note"] = ("fabric bytes per launch per kernel, keyed by the workload the rocprofv3 PMC passes ran (FETCH_SIZE x2 per the gfx950 note of "
              "MI355X_MICROARCH.md, WRITE_SIZE as reported, separate passes); bench.py looks its own workload up and reports null otherwise")
d["workloads"][key] = {k: v for k, v in run.items() if not k.startswith("_")}
d["workloads"][key]["_source"] = os.path.basename(sys.argv[1])
json.dump(d, open(dst, "w"), indent=1)
print("merged", key, "->", dst)
