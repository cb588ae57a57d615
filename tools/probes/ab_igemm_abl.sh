# ablation builds of the generic GEMM kernel (tools/probes/abl/lib_ig_*.so) on the linear layers: ms per layer and variant; GPU box
mkdir -p gpurun_out
{ echo "== NONE"; python tools/probes/ab_lib.py 64 linear
for n in NODMA NOLDS NOBAR NOMFMA NOEPI NODMALDS NODMALDSBAR; do echo "== $n"; TV_HIP_SO=tools/probes/abl/lib_ig_$n.so python tools/probes/ab_lib.py 64 linear; done; } 2>&1 | grep -v amdgpu.ids > gpurun_out/${AB_OUT:-ab_igemm_abl}.log
python - <<EOF
import re
runs={}; order=[]
cur=None
for l in open("gpurun_out/${AB_OUT:-ab_igemm_abl}.log"):
    if l.startswith("== "): cur=l.split()[1]; order.append(cur); continue
    m=re.match(r"linear\s+(\d+)->(\d+)\s+@(\d+)\s+(.+?)\s+([\d.]+) ms", l)
    if m and m.group(4).strip() in ("fwd","dgrad","fwd gelu","fwd+res"): runs.setdefault((m.group(1),m.group(2),m.group(3),m.group(4).strip()),{})[cur]=float(m.group(5))
print("%-28s"%"layer"+" ".join("%11s"%o for o in order))
for k,v in runs.items():
    print("%5s->%-5s@%-3s %-9s "%k+" ".join("%11.3f"%v.get(o,0) for o in order))
EOF
