# A/B of attention builds (tools/probes/build_variant.sh <name> attention.hip -D...), per-kernel times; GPU box
mkdir -p gpurun_out
for v in $AB_VARIANTS; do
  echo "== $v"; TV_HIP_SO=tools/probes/abl/lib_$v.so python tools/probes/attn_kernels.py 64 $AB_SHAPES
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${AB_OUT:-ab_attn}.log
