import sys, os, json, ctypes
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/deepl-project_amd')
import torch, bench
from transvae.hip import _lib
lib=_lib.load()
lib.tv_set_igemm_dbg.argtypes=[ctypes.c_int]
for dbg,name in [(0,'normal'),(1,'no DMA after prologue'),(2,'no MFMA'),(3,'no DMA, no MFMA'),(4,'DMA only (no ds_read/MFMA)')]:
    lib.tv_set_igemm_dbg(dbg)
    r=bench.time_dominant_kernel(32,256,torch.device('cuda',0))
    print(f"{name:32s} {r['ms_per_launch']:.3f} ms  ({r['achieved']:.0f} TF/s equiv)")
