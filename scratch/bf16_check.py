import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mixermdm_amd.sampler import Sampler
from mixermdm_amd.synthetic import synthetic_state_dict, synthetic_stats, synthetic_inputs, FULL_DIMS
sd = synthetic_state_dict(seed=0, std=0.02, bias_std=0.02, **FULL_DIMS); st = synthetic_stats()
B,T=2,64
cond,xT = synthetic_inputs(B,T)
res={}
for prec in ["fp32","bf16"]:
    s = Sampler(d_heads=8, m_heads=8, max_batch=B, max_frames=T, precision=prec, **FULL_DIMS)
    s.load_state_dict(sd); s.set_norm_stats(st["mean_hml"], st["std_hml"], st["mean_ih"], st["std_ih"]); s.prepare(); s.set_schedule("ddim50")
    s.begin(cond,xT); s.run(1, use_graph=False); a={k:v.clone() for k,v in s.state().items()}
    s.run(49, use_graph=True); b=s.state()["pred_xstart2"].clone()
    res[prec]=(a,b); s.close()
for k in ["model_out","pred_xstart2","x","x2"]:
    f,h=res["fp32"][0][k],res["bf16"][0][k]
    d=(f-h).abs(); print(f"step1 {k}: max {d.max().item():.3e} mean {d.mean().item():.3e} rel_rms {(d.pow(2).mean().sqrt()/f.pow(2).mean().sqrt()).item():.3e}")
f,h=res["fp32"][1],res["bf16"][1]; d=(f-h).abs()
print(f"ddim50 final: max {d.max().item():.3e} mean {d.mean().item():.3e} rel_rms {(d.pow(2).mean().sqrt()/f.pow(2).mean().sqrt()).item():.3e} finite {torch.isfinite(h).all().item()}")
