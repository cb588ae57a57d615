# same-box A/B of two builds on tools/probes/ab_lib.py <args>: AB_LIBS="path1 path2" (empty string = the in-tree build); GPU box
mkdir -p gpurun_out
for rep in 1 2; do for l in $AB_LIBS; do echo "== $l"; if [ "$l" = "tree" ]; then python tools/probes/ab_lib.py $AB_ARGS; else TV_HIP_SO=$l python tools/probes/ab_lib.py $AB_ARGS; fi; done; done 2>&1 | grep -v amdgpu.ids > gpurun_out/${AB_OUT:-ab_two}.log
python - <<EOF
import re
runs={}; order=[]
cur=None
for l in open("gpurun_out/${AB_OUT:-ab_two}.log"):
    if l.startswith("== "):
        cur=l.split()[1]
        if cur not in order: order.append(cur)
        continue
    m=re.match(r"(\S+)\s+(.+?)\s+([\d.]+) ms\s+(\d+) TF/s\s+sum (\S+) abs (\S+)", l)
    if m: runs.setdefault((m.group(1),m.group(2).strip()),{}).setdefault(cur,[]).append((float(m.group(3)),m.group(5),m.group(6)))
for k,v in runs.items():
    a=min(x[0] for x in v[order[0]]); b=min(x[0] for x in v[order[1]])
    same = v[order[0]][0][1:]==v[order[1]][0][1:]
    print("%-6s %-28s %.3f -> %.3f  %+.1f%%  %s"%(k+(a,b,(a/b-1)*100,"bits=" if same else "DIFF")))
EOF
