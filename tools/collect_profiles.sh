#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root:  bash tools/collect_profiles.sh <tag> [extra bench.py args]
# e.g. bash tools/collect_profiles.sh r01_k_atrium --scene sponza_like --spp 8 --rebuild-blas
# 1. bench.py default run            -> gpurun_out/prof_<tag>/bench.json
# 2. rocprofv3 --kernel-trace --stats of the same command -> kernel_stats.csv
# 3. two --pmc passes (FETCH_SIZE / WRITE_SIZE cannot share a pass on gfx950) -> pmc_fetch / pmc_write CSVs
# tools/summarize_profiles.py turns 2 + 3 into the small files committed under profiles/.
set -eo pipefail
TAG=${1:-run}
shift || true
EXTRA="$*"
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
ARGS="--steps 10 --warmup 2 --no-cpu-baseline $EXTRA"
timeout -k 10 400 python3 bench.py $EXTRA > "$OUT/bench.json" 2> "$OUT/bench.err"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- python3 "$ROOT/bench.py" $ARGS > "$OUT/stats.log" 2>&1
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o pmc -- python3 "$ROOT/bench.py" $ARGS --no-stage-timing > "$OUT/pmc_fetch.log" 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o pmc -- python3 "$ROOT/bench.py" $ARGS --no-stage-timing > "$OUT/pmc_write.log" 2>&1
cd "$ROOT"
python3 tools/summarize_profiles.py "$OUT" "$TAG"
