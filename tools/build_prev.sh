#!/bin/bash
# Builds the library of a commit (default HEAD) as low-precision-raytracer_amd/csrc/variants/liblprt_prev.so, for A/B runs of the
# working tree against it (tools/gpu_ab_hash.sh, tools/gpu_cornell_ab.sh, tools/gpu_c4_ab.sh):  bash tools/build_prev.sh [commit]
set -e
REV=${1:-HEAD}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
WT=$(mktemp -d /tmp/lprt_prev.XXXXXX)
git -C "$ROOT" worktree add -f "$WT" "$REV" -q
make -C "$WT/low-precision-raytracer_amd/csrc" -j8 2>&1 | grep -E "error|warning" || true
mkdir -p "$ROOT/low-precision-raytracer_amd/csrc/variants"
cp "$WT/low-precision-raytracer_amd/csrc/liblprt_hip.so" "$ROOT/low-precision-raytracer_amd/csrc/variants/liblprt_prev.so"
git -C "$ROOT" worktree remove --force "$WT"
ls -la "$ROOT/low-precision-raytracer_amd/csrc/variants/liblprt_prev.so"
