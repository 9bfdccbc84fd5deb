#!/bin/bash
# Bit-identity A/B of library variants on the GPU box: tools/gpu_frame_hash.py under each library, every frame size given in
# SIZES ("W H spp frames;..."), diffed against the first library's lines.
#   SIZES="333 187 2 4;640 360 2 2" bash tools/gpu_ab_hash.sh base.so new.so ...     ("default" = the in-tree liblprt_hip.so)
SIZES=${SIZES:-333 187 2 4}
mkdir -p gpurun_out/abh
IFS=';' read -ra SZ <<< "$SIZES"
for sz in "${SZ[@]}"; do
  n=$(echo $sz | tr ' ' x)
  first=""
  for lib in "$@"; do
    tag=$(basename $lib .so)
    if [ "$lib" = default ]; then env -u LPRT_LIBRARY python3 tools/gpu_frame_hash.py $sz > gpurun_out/abh/${tag}_$n.txt
    else LPRT_LIBRARY=$lib python3 tools/gpu_frame_hash.py $sz > gpurun_out/abh/${tag}_$n.txt; fi
    if [ -z "$first" ]; then first=$tag; else
      echo "$n $tag vs $first: $(diff gpurun_out/abh/${first}_$n.txt gpurun_out/abh/${tag}_$n.txt | grep '^<' | awk '{print $2":"$3}' | tr '\n' ' ')"
    fi
  done
done
