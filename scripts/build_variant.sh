#!/bin/bash
# Builds an experimental variant of the library next to the product one: scripts/build_variant.sh NAME [-DFLAG=...]
# -> sunray_amd/_variants/libsunray_hip_NAME.so. Select it at run time with SUNRAY_HIP_LIB=<path> (tuning only; the
# tests and the bench use the product library).
set -e
name=$1; shift
cd "$(dirname "$0")/.."
out=sunray_amd/_variants; mkdir -p $out/obj_$name
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fno-slp-vectorize -pthread"
objs=""
for s in kernels.hip post.hip bvh_gpu.hip api.cpp renderer.cpp multi_gpu.cpp gltf_load.cpp jpeg_decode.cpp host_prep.cpp bvh_build.cpp; do
  x=""; case $s in *.cpp) x="-x hip";; esac
  src=sunray_amd/csrc/$s; obj=$out/obj_$name/$s.o
  if [ "$s" = "kernels.hip" ] || [ ! -f $obj ] || [ $src -nt $obj ]; then /opt/rocm/bin/hipcc $F "$@" $x -c $src -o $obj 2>&1 | grep -E "error" || true; fi
  objs="$objs $obj"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libsunray_hip_$name.so $objs -pthread -lz -ldl
echo $out/libsunray_hip_$name.so
