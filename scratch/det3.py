import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, math
from mixermdm_amd import ops, load_library
lib=load_library()
d = torch.device("cuda:0"); torch.manual_seed(0)
nseq,T,H,dh = 64,64,8,128; D=H*dh
qkv = torch.randn(nseq,T,3*D,device=d)
f = lambda: ops.attention(qkv[...,:D],qkv[...,D:2*D],qkv[...,2*D:],H)
for dbg in [16, 8, 0]:
    lib.mmdmx_set_attn_dbg(dbg)
    sel = (lambda t: t.view(nseq,T,H,dh)[...,:4]) if (dbg & 24) else (lambda t: t)
    ref=sel(f()).clone(); bad=sum(0 if torch.equal(sel(f()),ref) else 1 for _ in range(20))
    o=sel(f()); print("   max diff", (o-ref).abs().max().item(), "n diff", (o!=ref).sum().item(), "lanes-groups agree:", (o[...,0:1]==o).all().item() if (dbg&24) else "")
    print("dbg",dbg,"differ",bad,"/20")
