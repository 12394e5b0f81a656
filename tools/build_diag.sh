#!/bin/bash
# Build timing-diagnostic variants of libmmhip.so into build_diag/ (results are WRONG by design; never shipped).
#   tools/build_diag.sh NAME "-DMACRO1 -DMACRO2"
set -e
cd "$(dirname "$0")/.."
name=$1; defs=$2
mkdir -p build_diag/obj_$name
objs=""
for f in mm_gemm mm_attn mm_rowwise mm_embed mm_optim mm_debug mm_comm mm_image mm_xattn; do
  o=multimeditron_amd/csrc/build/$f.o
  if [ $f = mm_gemm ] || { [ $f = mm_attn ] && [ -z "$MM_DIAG_GEMM_ONLY" ]; }; then
    o=build_diag/obj_$name/$f.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result $defs -c multimeditron_amd/csrc/$f.hip -o $o &
  fi
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_diag/libmmhip_$name.so $objs -ldl
echo build_diag/libmmhip_$name.so
