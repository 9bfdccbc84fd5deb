#!/bin/bash
# A/B of the configs[3] leg (atrium 1080p 8 spp fp16, BLAS rebuilt per frame) under environment knobs, on the GPU box:
#   bash tools/gpu_c4_ab.sh <tag> "LPRT_RAY_SORT=0" "LPRT_RAY_SORT=3" ...
# prints ms_per_step and the traversal stage times of each variant (bench.py's own c4 leg, no CPU baseline).
TAG=$1; shift
mkdir -p gpurun_out
for V in "$@"; do
  OUT="gpurun_out/ab_${TAG}_$(echo "$V" | tr ' =/' '___' | tail -c 60).json"
  env $V python3 bench.py --no-cpu-baseline --steps 10 --warmup 2 --sustain-seconds 0 > "$OUT" 2> "$OUT.err" || { echo "FAILED: $V"; tail -5 "$OUT.err"; continue; }
  python3 - "$OUT" "$V" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c = d["c4"]; s = c["stage_ms_per_frame"]
keys = ("build_blas", "primary", "shade1", "trace_di1", "ray_sort", "trace_gi", "shade2", "trace_di2", "compose")
print(f"{sys.argv[2]:40s} c4 {c['ms_per_step']:7.2f} ms | " + " ".join(f"{k}={s.get(k, 0):.2f}" for k in keys) + f" | cornell f32 {d['ms_per_step']:.3f} f16 {d['fp16']['ms_per_step']:.3f}")
PY
done
