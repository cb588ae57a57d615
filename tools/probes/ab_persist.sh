# A/B of the persistent tile walk of the generic 256 x 256 tile (tv_set_igemm_persist 0 / 1) on the linear layers; GPU box
mkdir -p gpurun_out
for p in ${AB_PAIR:-0 1} ${AB_PAIR:-0 1}; do echo "== persist $p"; TV_AB_PERSIST=$p python tools/probes/ab_lib.py 64 linear; done 2>&1 | grep -v amdgpu.ids > gpurun_out/${AB_OUT:-ab_persist}.log
python - <<EOF
import re
runs={}
cur=None
for l in open("gpurun_out/${AB_OUT:-ab_persist}.log"):
    if l.startswith("== persist"): cur=l.split()[2]; cur={"-1":"0","-2":"1"}.get(cur,cur); continue
    m=re.match(r"linear\s+(\d+)->(\d+)\s+@(\d+)\s+(.+?)\s+([\d.]+) ms", l)
    if m: runs.setdefault((m.group(1),m.group(2),m.group(3),m.group(4).strip()),{}).setdefault(cur,[]).append(float(m.group(5)))
ta=tb=0
for k,v in runs.items():
    a=min(v["0"]); b=min(v["1"]); ta+=a; tb+=b; print("%5s->%-5s@%-3s %-9s  %.3f -> %.3f  %+.1f%%"%(k+(a,b,(a/b-1)*100)))
print("sum %.3f -> %.3f  %+.1f%%"%(ta,tb,(ta/tb-1)*100))
EOF
