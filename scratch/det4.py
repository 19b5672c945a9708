import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, math
from mixermdm_amd import ops, load_library
d = torch.device("cuda:0"); torch.manual_seed(0)
nseq,T,H,dh = 64,64,8,128; D=H*dh
qkv = torch.randn(nseq,T,3*D,device=d)
f = lambda: ops.attention(qkv[...,:D],qkv[...,D:2*D],qkv[...,2*D:],H)
ref=f().clone(); o=f()
diff=(o-ref)
nz=diff!=0
print("frac differing", nz.float().mean().item())
rel=(diff[nz].abs()/ref[nz].abs().clamp_min(1e-30))
print("rel diff quantiles", torch.quantile(rel.float().cpu()[:1000000], torch.tensor([0.1,0.5,0.9,0.99,1.0])).tolist())
# per (seq,head,qtile-wave) block: fraction differing
blk = nz.view(nseq, T//16, 16, H, dh).float().mean(dim=(2,4))   # [seq, wave-tile, head]
print("blocks fully identical:", (blk==0).sum().item(), "of", blk.numel(), " blocks fully different(>0.9):", (blk>0.9).sum().item(), " partial:", ((blk>0)&(blk<=0.9)).sum().item())
print("example block fracs", blk.flatten()[:24].tolist())
# within a differing block, is the ratio o/ref constant per row?
s,w,h = (blk>0.9).nonzero()[0].tolist()
ro = o[s, w*16:(w+1)*16, h*dh:(h+1)*dh]; rr = ref[s, w*16:(w+1)*16, h*dh:(h+1)*dh]
print("ratio per row (first 6 d):", (ro/rr)[:4,:6].tolist())
