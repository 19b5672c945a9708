import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, math, statistics
from mixermdm_amd import ops, load_library
lib = load_library(); d = torch.device("cuda:0")
def t(M,N,K,epi):
    x = torch.randn(M,K,device=d); w = torch.randn(N,K,device=d)/math.sqrt(K); b = torch.randn(N,device=d); out = torch.empty(M,N,device=d)
    extra = out if epi=="resid" else None
    res=[]
    for r in range(7):
        ops.linear(x,w,b,epi,extra,out=out)
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record()
        for _ in range(4): ops.linear(x,w,b,epi,extra,out=out)
        e1.record(); torch.cuda.synchronize(); res.append(e0.elapsed_time(e1)/4)
    ms=statistics.median(res); return ms*1e3, 2*M*N*K/ms/1e9
for M in [19200, 20480]:
  for N in [1024, 3072]:
    for K in [256, 512,1024,2048,4096]:
        for epi in ["bias","resid"]:
            us,tf=t(M,N,K,epi); print(f"M={M} N={N} K={K} {epi:5s} {us:8.1f} us {tf:6.1f} TF", flush=True)
