"""Whole 1000-step loops at BASELINE configs[2] (B=16, T=300): wall time of sample() per precision mode and how far the modes end up apart."""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mixermdm_amd.sampler import Sampler
from mixermdm_amd.synthetic import synthetic_state_dict, synthetic_stats, synthetic_inputs, FULL_DIMS
B, T = 16, 300
sd = synthetic_state_dict(seed=0, std=0.02, **FULL_DIMS); st = synthetic_stats(); cond, xT = synthetic_inputs(B, T)
outs = {}
for prec in ["fp32", "fp32_split", "bf16", "bf16_fp8"]:
    s = Sampler(d_heads=8, m_heads=8, max_batch=B, max_frames=T, precision=prec, **FULL_DIMS)
    s.load_state_dict(sd); s.set_norm_stats(st["mean_hml"], st["std_hml"], st["mean_ih"], st["std_ih"]); s.prepare(); s.set_schedule("ddim1000")
    s.begin(cond, xT); s.run(2, True); s.synchronize()          # graph capture outside the timed call
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = s.sample(cond, xT, use_graph=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    outs[prec] = out.cpu()
    print(f"{prec:10s} 1000 steps, B={B}: {dt:6.2f} s -> {B/dt:.4f} motions/s, finite={bool(torch.isfinite(out).all())}, |out| rms {out.pow(2).mean().sqrt().item():.4f}", flush=True)
    s.close()
rel = lambda a, b: ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()
print("after 1000 steps, relative RMS vs fp32: fp32_split %.3e, bf16 %.3e, bf16_fp8 %.3e" % (rel(outs["fp32_split"], outs["fp32"]), rel(outs["bf16"], outs["fp32"]), rel(outs["bf16_fp8"], outs["fp32"])))
