# same-box A/B of library variants, weight-gradient column: bash tools/probes/ab_wgrad.sh "VARIANTS" "LAYER-FILTERS"
for rep in 1 2; do for v in $1; do
  if [ $v = default ]; then unset TV_HIP_SO; else export TV_HIP_SO=$PWD/tools/probes/abl/lib_$v.so; fi
  for l in $2; do echo -n "$v rep$rep: "; timeout -k 10 200 python tools/gemm_sweep.py --mb 64 --only "$l" 2>&1 | grep -v "^TOTAL\|amdgpu.ids\|^layer" | awk '{printf "%s wgrad %s | ", $1, $13} END {print ""}'; done
done; done
