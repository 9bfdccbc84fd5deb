#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root:  bash tools/collect_profiles.sh <tag>
# 1. the default bench.py run (configs[1] + [2] + the c4 leg)            -> gpurun_out/prof_<tag>/bench.json
# 2. rocprofv3 --kernel-trace --stats of that same command                 -> <tag>_kernel_stats.csv
# 3. per WORKLOAD, two --pmc passes (FETCH_SIZE / WRITE_SIZE cannot share a pass on gfx950) of bench.py restricted to it
#    -> <tag>_<workload>_pmc_traffic.json, merged into profiles/pmc_traffic.json under the workload's key (tools/merge_pmc.py)
# tools/summarize_profiles.py condenses 2 + 3 into the small files committed under profiles/.
set -o pipefail
TAG=${1:-run}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
timeout -k 10 500 python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- python3 "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/stats.log" 2>&1
cd "$ROOT" && python3 tools/summarize_profiles.py "$OUT" "$TAG"
pmc() { # <name> <workload key> <bench args...>
  local NAME=$1 KEY=$2; shift 2
  local W=$OUT/$NAME; mkdir -p "$W"
  cd /tmp
  timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$W/pmc_fetch" -o pmc -- python3 "$ROOT/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --no-stage-timing --no-c4 "$@" > "$W/pmc_fetch.log" 2>&1
  timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$W/pmc_write" -o pmc -- python3 "$ROOT/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --no-stage-timing --no-c4 "$@" > "$W/pmc_write.log" 2>&1
  cd "$ROOT" && python3 tools/summarize_profiles.py "$W" "${TAG}_$NAME" && python3 tools/merge_pmc.py "$W/summary/${TAG}_${NAME}_pmc_traffic.json" "$KEY"
  cp "$W/summary/${TAG}_${NAME}_pmc_traffic.json" "$OUT/summary/" 2>/dev/null
}
pmc cornell "cornell 1920x1080 4spp rebuild_blas=0"
pmc atrium "sponza_like 1920x1080 8spp rebuild_blas=1" --scene sponza_like --spp 8 --rebuild-blas
cp profiles/pmc_traffic.json "$OUT/summary/pmc_traffic.json"
ls "$OUT/summary"
