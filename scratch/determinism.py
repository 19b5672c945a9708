import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, math
from mixermdm_amd import ops
d = torch.device("cuda:0")
torch.manual_seed(0)
def check(name, fn, n=40):
    ref = fn().clone(); bad = 0; mx = 0.0
    for i in range(n):
        o = fn()
        if not torch.equal(o, ref):
            bad += 1; mx = max(mx, (o-ref).abs().max().item())
    print(f"{name}: {bad}/{n} runs differ, max diff {mx:.3e}", flush=True)
for (M,N,K) in [(256,1024,1024),(128,3072,1024),(19200,1024,1024),(1280,2048,1024),(64,65536,1024)]:
    x = torch.randn(M,K,device=d); w = torch.randn(N,K,device=d)/math.sqrt(K); b = torch.randn(N,device=d)
    check(f"gemm {M}x{N}x{K}", lambda: ops.linear(x,w,b))
for (nseq,T,H,dh) in [(8,32,8,128),(8,24,8,128),(64,300,8,128),(8,32,8,64),(64,300,8,64),(4,16,8,128)]:
    D=H*dh; qkv = torch.randn(nseq,T,3*D,device=d)
    check(f"attn nseq={nseq} T={T} dh={dh}", lambda: ops.attention(qkv[...,:D],qkv[...,D:2*D],qkv[...,2*D:],H))
print("--- T sweep")
for T in [48, 64, 65, 128, 256, 288, 290, 300, 304]:
    for nseq in [8, 64]:
        H,dh=8,128; D=H*dh; qkv = torch.randn(nseq,T,3*D,device=d)
        check(f"attn nseq={nseq} T={T} dh={dh}", lambda: ops.attention(qkv[...,:D],qkv[...,D:2*D],qkv[...,2*D:],H), n=10)
