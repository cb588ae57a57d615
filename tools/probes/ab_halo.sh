# same-box A/B of library variants on chosen layers: bash tools/probes/ab_halo.sh "VARIANTS" "LAYER-FILTERS" "HALO-MODES"
for rep in 1 2; do for v in $1; do
  if [ $v = default ]; then unset TV_HIP_SO; else export TV_HIP_SO=$PWD/tools/probes/abl/lib_$v.so; fi
  for h in $3; do for l in $2; do echo -n "$v rep$rep halo$h: "; timeout -k 10 200 python tools/gemm_sweep.py --mb 64 --halo $h --only "$l" 2>&1 | grep -v "^TOTAL\|amdgpu.ids\|^layer" | awk '{printf "%s fwd %s (%s ms) | ", $1, $7, $6} END {print ""}'; done; done
done; done
