import os, sys, torch
ROOT=os.getcwd(); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "deepl-project_amd"))
from transvae.hip import ops
dev=torch.device("cuda:0")
def tm(fn, it=8):
    for _ in range(2): fn()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/it
for (mode, xs, cout) in [("c3s1",(64,256,256,192),192),("c3s1",(64,128,128,192),192),("linear",(65536,768),3072),("linear",(16384,1536),6144),("c3s1",(64,16,16,1536),1536)]:
    x=torch.randn(*xs,device=dev).to(torch.bfloat16)
    w=torch.randn(cout,*((3,3) if mode!="linear" else ()),xs[-1],device=dev)
    g=ops._Geo(mode,x,w)
    gy=torch.randn(*g.out_shape,device=dev).to(torch.bfloat16)
    t1=tm(lambda: ops.conv_wgrad(g,w,x,gy,True)); t0=tm(lambda: ops.conv_wgrad(g,w,x,gy,False))
    print(f"{mode} {xs}->{cout}: wgrad with bias {t1:.3f} ms, without {t0:.3f} ms")
