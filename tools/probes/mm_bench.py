import torch, time
dev=torch.device('cuda:0')
def tm(fn, it=10):
    for _ in range(3): fn()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/it
for (M,K,N) in [(65536,768,3072),(65536,3072,768),(65536,768,2304),(16384,1536,6144),(16384,6144,1536),(262144,384,1536),(262144,1536,384),(65536,768,768),(4194304,192,192)]:
    a=torch.randn(M,K,device=dev).to(torch.bfloat16); w=torch.randn(N,K,device=dev).to(torch.bfloat16)
    t=tm(lambda: torch.nn.functional.linear(a,w))
    print(f"M={M:8d} K={K:5d} N={N:5d}: {t:7.3f} ms {2*M*K*N/t/1e9:7.0f} TF/s")
for (M,K,N) in [(8192,8192,8192),(16384,8192,8192),(4194304,1728,192)]:
    a=torch.randn(M,K,device=dev).to(torch.bfloat16); w=torch.randn(N,K,device=dev).to(torch.bfloat16)
    t=tm(lambda: torch.nn.functional.linear(a,w))
    print(f"M={M:8d} K={K:5d} N={N:5d}: {t:7.3f} ms {2*M*K*N/t/1e9:7.0f} TF/s")
