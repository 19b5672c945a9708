import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, math
from mixermdm_amd import ops
d = torch.device("cuda:0"); torch.manual_seed(0)
nseq,T,H,dh = 64,64,8,128; D=H*dh
qkv = torch.randn(nseq,T,3*D,device=d)
f = lambda: ops.attention(qkv[...,:D],qkv[...,D:2*D],qkv[...,2*D:],H)
ref = f().clone()
# fp64 reference
q,k,v = [t.double().view(nseq,T,H,dh).transpose(1,2) for t in (qkv[...,:D],qkv[...,D:2*D],qkv[...,2*D:])]
z = torch.zeros(nseq,H,1,dh,dtype=torch.float64,device=d)
a = torch.softmax(q @ torch.cat([k,z],2).transpose(-1,-2)/math.sqrt(dh), -1)
gold = (a @ torch.cat([v,z],2)).transpose(1,2).reshape(nseq,T,D)
print("ref err vs fp64", (ref.double()-gold).abs().max().item())
for i in range(5):
    o = f()
    diff = (o != ref)
    idx = diff.nonzero()
    print("run", i, "ndiff", idx.shape[0], "err vs fp64", (o.double()-gold).abs().max().item())
    if idx.shape[0]:
        s_ = idx[:,0].unique(); q_ = idx[:,1].unique(); c_ = idx[:,2]
        print("  seqs", s_.tolist()[:20], " q rows", q_.tolist()[:40], " heads", (c_//dh).unique().tolist(), " d", (c_%dh).unique().tolist()[:40])
