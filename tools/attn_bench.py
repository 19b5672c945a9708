"""Stand-alone timing of the fp32 attention kernel at the step's shapes (GPU box), after a clock warm-up (the first ~30 launches after an idle
period run at lower clocks: without it this script read 246 us where the kernel takes 208).  ABL= ablation bits (mmdm_diag_set "attn_ablate",
diagnostic instantiation)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, statistics
from mixermdm_amd import ops, load_library
lib = load_library()
ops.attention(torch.zeros(1,16,64,device="cuda:0"),torch.zeros(1,16,64,device="cuda:0"),torch.zeros(1,16,64,device="cuda:0"),1)
lib.mmdm_diag_set(b"attn_ablate", int(os.environ.get("ABL","0")))
d = torch.device("cuda:0")
_w = torch.randn(4096, 4096, device=d)
for _ in range(40): ops.linear(_w, _w)          # clock ramp
shapes = [(64,300,8,128,"d.sa"),(64,300,8,64,"m.sa"),(128,300,8,128,"d.sa B=32"),(64,196,8,128,"single T=196")]
if os.environ.get("TAIL") == "1":      # what a perfect query / key tail could buy at configs[1]'s T = 196: the neighbouring lengths that have no tail
    shapes = [(64,T,8,128,"single T=%d" % T) for T in (176, 192, 196, 208, 240, 256, 300)]
for nseq,T,H,dh,name in shapes:
    D=H*dh
    qkv = torch.randn(nseq,T,3*D,device=d)
    line = f"{name} nseq={nseq} T={T} H={H} dh={dh}:"
    outs = []
    for nw in (4,):
        res=[]
        for r in range(7):
            o = ops.attention(qkv[...,:D],qkv[...,D:2*D],qkv[...,2*D:],H)
            e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4): ops.attention(qkv[...,:D],qkv[...,D:2*D],qkv[...,2*D:],H)
            e1.record(); torch.cuda.synchronize(); res.append(e0.elapsed_time(e1)/4)
        outs.append(o)
        ms=statistics.median(res); fl=4*nseq*H*T*(T+1)*dh
        line += f"  {ms*1e3:.1f} us  {fl/ms/1e9:.1f} TF/s"
    print(line, flush=True)
