import torch, sys
dev='cuda'
def timeit(fn,n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/n*1e3
for M in (401408,131072,100352):
  for K,N in ((256,256),(72,256),(256,32),(352,256)):
    x=torch.randn(M,K,device=dev); w=torch.randn(N,K,device=dev)
    xb,wb=x.bfloat16(),w.bfloat16()
    t32=timeit(lambda: x@w.t()); t16=timeit(lambda: xb@wb.t())
    g=torch.randn(M,N,device=dev); gb=g.bfloat16()
    S=64
    tw32=timeit(lambda: torch.bmm(g.view(S,M//S,N).transpose(1,2), x.view(S,M//S,K)).sum(0))
    tw16=timeit(lambda: torch.bmm(gb.view(S,M//S,N).transpose(1,2), xb.view(S,M//S,K)).sum(0))
    fl=2*M*K*N
    print(f"M={M} K={K} N={N}: fwd fp32 {t32:7.1f} us ({fl/t32/1e6:6.1f} TF)  bf16 {t16:7.1f} us ({fl/t16/1e6:6.1f} TF) | wgrad splitM fp32 {tw32:7.1f} bf16 {tw16:7.1f} us | bytes-bound bf16 {(M*K*2+M*N*2)/5e6:6.1f} us")
x=torch.randn(401408,256,device=dev)
print('cast fp32->bf16 401408x256', timeit(lambda: x.bfloat16()), 'us; copy', timeit(lambda: x.clone()),'us')
