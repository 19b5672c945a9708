import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mixermdm_amd.sampler import Sampler
from mixermdm_amd.synthetic import synthetic_state_dict, synthetic_stats, synthetic_inputs, FULL_DIMS
B, T = 4, 300
sd = synthetic_state_dict(seed=0, std=0.02, bias_std=0.02, **FULL_DIMS); st = synthetic_stats(); cond, xT = synthetic_inputs(B, T)
res = {}
for prec in ["fp32", "fp32_split"]:
    s = Sampler(d_heads=8, m_heads=8, max_batch=B, max_frames=T, precision=prec, **FULL_DIMS)
    s.load_state_dict(sd); s.set_norm_stats(st["mean_hml"], st["std_hml"], st["mean_ih"], st["std_ih"]); s.prepare(); s.set_schedule("ddim20")
    s.begin(cond, xT)
    h = s.set_history(("influence_i1", "influence_i2", "out1", "out2", "out_influenced"), 1)
    s.run(1, use_graph=False); s.synchronize()
    res[prec] = {k: v[0].clone().cpu() for k, v in h.items()}
    res[prec]["model_out"] = s.state()["model_out"].clone().cpu()
    s.close()
rel = lambda a, b: ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()
for k in res["fp32"]:
    a, b = res["fp32_split"][k], res["fp32"][k]
    print(f"{k:16s} rel rms {rel(a,b):.3e}  max abs {(a-b).abs().max().item():.3e}  |ref| rms {b.pow(2).mean().sqrt().item():.3e}")
