#!/bin/bash
# Builds low-precision-raytracer_amd/csrc/variants/liblprt_<name>.so with extra compiler flags for some translation units:
#   bash tools/build_variant.sh <name> "<flags>" ["<units>"]      units default to "pipeline pipeline_refslab" (the traversal
#   tuning macros); "svgf" for the denoiser's.  The other objects are the default build's (run make first).
# Use it with LPRT_LIBRARY=<path> (renderer.library_path) for A/B runs on the GPU box.
set -e
NAME=$1; FLAGS=$2; UNITS=${3:-pipeline pipeline_refslab}
cd "$(dirname "$0")/../low-precision-raytracer_amd/csrc"
mkdir -p variants
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function"
OBJS=""
for U in api build rng svgf trace_kernels trace_kernels_refslab pipeline pipeline_refslab; do
  if [[ " $UNITS " == *" $U "* ]]; then
    /opt/rocm/bin/hipcc $COMMON $FLAGS -c $U.hip -o variants/${U}_$NAME.o &
    OBJS="$OBJS variants/${U}_$NAME.o"
  else
    OBJS="$OBJS $U.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/liblprt_$NAME.so $OBJS
for U in $UNITS; do rm -f variants/${U}_$NAME.o; done
ls -la variants/liblprt_$NAME.so
