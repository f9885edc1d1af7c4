#!/bin/bash
# tools/build_variant.sh NAME [-DMACRO=VALUE ...]: another build of the HIP library
# with compile-time switches, as ffn_amd/csrc/libffn_hip_NAME.so (select it with
# FFN_AMD_LIB=...: same-box A/B runs of kernel variants in one gpurun call)
set -e
name=$1; shift
cd "$(dirname "$0")/../ffn_amd/csrc"
mkdir -p build/$name
for f in ffn_hip ffn_labels ffn_seeds; do
  if [ $f = ffn_hip ] || [ ! -f build/$f.o ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $f.hip -o build/$name/$f.o &
  else
    cp build/$f.o build/$name/$f.o
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libffn_hip_$name.so build/$name/ffn_hip.o build/$name/ffn_labels.o build/$name/ffn_seeds.o
echo built libffn_hip_$name.so
