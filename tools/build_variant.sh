#!/bin/bash
# Builds low-precision-raytracer_amd/csrc/variants/liblprt_<name>.so with extra compiler flags for pipeline.hip / pipeline_refslab.hip
# (the translation units the traversal tuning macros live in):  bash tools/build_variant.sh <name> "<flags>"
# Use it with LPRT_LIBRARY=<path> (renderer.library_path) for A/B runs on the GPU box.
set -e
NAME=$1; FLAGS=$2
cd "$(dirname "$0")/../low-precision-raytracer_amd/csrc"
mkdir -p variants
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $COMMON $FLAGS -c pipeline.hip -o variants/pipeline_$NAME.o &
/opt/rocm/bin/hipcc $COMMON $FLAGS -c pipeline_refslab.hip -o variants/pipeline_refslab_$NAME.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/liblprt_$NAME.so api.o build.o rng.o svgf.o trace_kernels.o trace_kernels_refslab.o variants/pipeline_$NAME.o variants/pipeline_refslab_$NAME.o
rm -f variants/pipeline_$NAME.o variants/pipeline_refslab_$NAME.o
ls -la variants/liblprt_$NAME.so
