#!/bin/bash
# Builds a variant of the library with extra -D flags on ONE source file (the other objects come from the in-tree build):
#   tools/probes/build_variant.sh pp igemm_nt.hip -DTV_HALO_PP=1      ->  tools/probes/abl/lib_pp.so
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/../../deepl-project_amd"
mkdir -p ../tools/probes/abl
extra=""; [ "$src" = "attention.hip" ] && extra="-mllvm -amdgpu-mfma-vgpr-form=1 -fno-slp-vectorize"
[ "$src" = "attention.hip" ] && [ -n "$TV_SLP" ] && extra="-mllvm -amdgpu-mfma-vgpr-form=1"   # TV_SLP=1: with the SLP vectorizer (A/B)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value $extra "$@" -c csrc/$src -o /tmp/var_${name}_${src%.hip}.o
objs=""
for f in csrc/*.hip; do
  b=$(basename ${f%.hip})
  if [ "$b.hip" = "$src" ]; then objs="$objs /tmp/var_${name}_$b.o"; else objs="$objs build/$b.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../tools/probes/abl/lib_$name.so $objs
echo built tools/probes/abl/lib_$name.so
