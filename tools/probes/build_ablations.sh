#!/bin/bash
# Builds ablated variants of the library (timing probes only; results are wrong by construction).
set -e
cd "$(dirname "$0")/../../deepl-project_amd/csrc"
OUT=../../tools/probes/abl; mkdir -p $OUT
for v in NONE NO_DMA NO_BARRIER NO_MFMA NO_LDSREAD "NO_DMA -DTV_ABL_NO_LDSREAD" "NO_DMA -DTV_ABL_NO_BARRIER -DTV_ABL_NO_LDSREAD"; do
  name=$(echo "$v" | tr -d ' ' | sed 's/-DTV_ABL_/_/g')
  objs=""
  for f in *.hip; do
    o=/tmp/abl_${name}_${f%.hip}.o
    extra=""; [ "$f" = "attention.hip" ] && extra="-mllvm -amdgpu-mfma-vgpr-form=1"
    hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -DTV_ABL_$v $extra -c $f -o $o &
    objs="$objs $o"
  done
  wait
  hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/lib_$name.so $objs
  echo built $name
done
