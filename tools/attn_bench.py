import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, statistics
from mixermdm_amd import ops, load_library
import os
ops.attention(torch.zeros(1,16,64,device="cuda:0"),torch.zeros(1,16,64,device="cuda:0"),torch.zeros(1,16,64,device="cuda:0"),1)
load_library().mmdmx_set_attn_ablate(int(os.environ.get("ABL","0")))
d = torch.device("cuda:0")
for nseq,T,H,dh,name in [(64,300,8,128,"d.sa"),(64,300,8,64,"m.sa")]:
    D=H*dh
    qkv = torch.randn(nseq,T,3*D,device=d)
    res=[]
    for r in range(7):
        ops.attention(qkv[...,:D],qkv[...,D:2*D],qkv[...,2*D:],H)
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4): ops.attention(qkv[...,:D],qkv[...,D:2*D],qkv[...,2*D:],H)
        e1.record(); torch.cuda.synchronize(); res.append(e0.elapsed_time(e1)/4)
    ms=statistics.median(res); fl=4*nseq*H*T*(T+1)*dh
    print(f"{name} nseq={nseq} T={T} H={H} dh={dh}: {ms*1e3:.1f} us  {fl/ms/1e9:.1f} TF/s")
