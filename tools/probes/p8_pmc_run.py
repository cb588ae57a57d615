"""The eight-phase GEMM loop alone on two linear layers (forward), 40 launches each: for a PMC pass."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "deepl-project_amd"))
import torch
from transvae.hip import ops, _lib as L
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
shapes = ((16384, 1536, 6144), (65536, 3072, 768))
if len(sys.argv) > 1:      # one shape: python p8_pmc_run.py M K N
    shapes = (tuple(int(v) for v in sys.argv[1:4]),)
for (M, K, N) in shapes:
    x = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    w = torch.randn(N, K, device=dev, generator=g) * K ** -0.5
    b = torch.randn(N, device=dev, generator=g) * 0.1
    for _ in range(40):
        ops.conv_forward(x, w, b, None, "linear", L.ACT_NONE, False)
torch.cuda.synchronize()
