#!/bin/bash
# Linear-layer tile configurations through the tuning hook (GPU box):  tools/probes/linear_cfg_sweep.sh > gpurun_out/linear_cfg.txt
cd "$(dirname "$0")/../.."
for cfg in "" 128,256,2,32 128,256,2,64 128,128,2,64 128,128,3,64 256,128,2,32 128,192,2,32 128,192,2,64 256,256,2,32; do
  echo "=== TV_AB_CFG=$cfg"
  TV_AB_CFG=$cfg python3 tools/probes/ab_lib.py 64 linear 2>&1 | grep -v amdgpu.ids
done
