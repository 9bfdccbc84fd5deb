#!/bin/bash
# A/B of the Cornell legs (configs[1] fp32, configs[2] fp16) under environment knobs / library variants, on the GPU box:
#   bash tools/gpu_cornell_ab.sh <tag> "LPRT_X=0" "LPRT_LIBRARY=/root/repo/low-precision-raytracer_amd/csrc/variants/liblprt_v.so" ...
# prints ms per frame and the stage times named in STAGES (default: the denoiser's) of each variant.
TAG=$1; shift
STAGES=${STAGES:-svgf_temporal svgf_moments svgf_wavelet finalize}
mkdir -p gpurun_out
for V in "$@"; do
  OUT="gpurun_out/abc_${TAG}_$(echo "$V" | tr ' =/' '___' | tail -c 40).json"
  env $V python3 bench.py --no-cpu-baseline --no-c4 --steps 20 --warmup 3 --sustain-seconds 0 > "$OUT" 2> "$OUT.err" || { echo "FAILED: $V"; tail -5 "$OUT.err"; continue; }
  python3 - "$OUT" "$V" $STAGES <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for name, dd in (("f32", d), ("f16", d["fp16"])):
    s = dd["stage_ms_per_frame"]
    print(f"{sys.argv[2][-44:]:44s} {name} {dd['ms_per_step']:6.3f} ms | " + " ".join(f"{k}={s.get(k, 0):.3f}" for k in sys.argv[3:]))
PY
done
