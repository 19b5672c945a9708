import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, math, statistics
from mixermdm_amd import ops, load_library
lib = load_library()
d = torch.device("cuda:0")
shapes = [(19200,3072,1024,"qkv"),(19200,1024,1024,"out"),(19200,2048,1024,"ffn1"),(19200,1024,2048,"ffn2"),
          (19200,1536,512,"m.qkv"),(19200,512,512,"m.out"),(19200,1024,512,"m.ffn1"),(19200,512,1024,"m.ffn2"),(4096,4096,4096,"sq4k")]
# CFGS: comma-separated tile configurations (mmdm_diag_set "gemm_cfg"; -1 = the production dispatch), each optionally ":t" = row split of the
# fractional last round when it holds <= t/10 of the resident slots (mmdm_diag_set "gemm_tail"), e.g. CFGS=-1,-1:6,-1:10
cfgs = [c for c in os.environ.get("CFGS","0,1,4").split(",")]
def set_cfg(c):
    cfg, _, tail = c.partition(":")
    lib.mmdm_diag_set(b"gemm_cfg", int(cfg)); lib.mmdm_diag_set(b"gemm_tail", int(tail or 0))
lib.mmdm_diag_set(b"gemm_ablate", int(os.environ.get('ABL','0')))
only = sys.argv[1:]
rounds, reps = 7, 4
# clock ramp: after an idle period the first ~30 launches run at lower clocks (a 1 ms GEMM went 1133 -> 900 us over 40 back-to-back calls);
# warm the chip up before the first timed round, or the first configuration of a shape is measured ~10 % low
_w = torch.randn(4096, 4096, device=d)
for _ in range(60): ops.linear(_w, _w)
torch.cuda.synchronize()
for M,N,K,name in shapes:
    if only and name not in only: continue
    x = torch.randn(M,K,device=d); w = torch.randn(N,K,device=d)/math.sqrt(K); b = torch.randn(N,device=d)
    out = torch.empty(M,N,device=d)
    epi = {"qkv":"bias","out":"resid","ffn1":"gelu","ffn2":"resid","m.qkv":"bias","m.out":"resid","m.ffn1":"gelu","m.ffn2":"resid","sq4k":"bias"}[name]
    extra = out if epi=="resid" else None
    res = {c: [] for c in cfgs}
    for r in range(rounds):
        for c in cfgs:
            set_cfg(c)
            ops.linear(x,w,b,epi,extra,out=out)
            e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): ops.linear(x,w,b,epi,extra,out=out)
            e1.record(); torch.cuda.synchronize()
            res[c].append(e0.elapsed_time(e1)/reps)
    line = f"{name:7s} {M}x{N}x{K} {epi:5s}"
    for c in cfgs:
        ms = statistics.median(res[c]); line += f" | cfg{c}: {ms*1e3:7.1f}us {2*M*N*K/ms/1e9:6.1f}TF (min {2*M*N*K/min(res[c])/1e9:5.1f}..{2*M*N*K/max(res[c])/1e9:5.1f})"
    print(line, flush=True)
