"""scratch: which module of a low-precision handle gives other bits while ANOTHER handle samples beside it?"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mixermdm_amd.sampler import Sampler
from mixermdm_amd.synthetic import synthetic_state_dict, synthetic_stats, synthetic_inputs, FULL_DIMS
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
sd = synthetic_state_dict(seed=0, std=0.02, bias_std=0.0, **FULL_DIMS); st = synthetic_stats()
def fresh():
    t = Sampler(d_heads=8, m_heads=8, max_batch=1, max_frames=300, precision=prec, **FULL_DIMS)
    t.load_state_dict(sd); t.set_norm_stats(st["mean_hml"], st["std_hml"], st["mean_ih"], st["std_ih"]); t.prepare(); t.set_schedule("ddim50")
    return t
A, B = fresh(), fresh()
T = 181
g = torch.Generator().manual_seed(5)
n = 2
x1 = torch.randn(n, T, 524, generator=g).cuda(); x2 = torch.randn(n, T, 524, generator=g).cuda()
cond = torch.randn(n, 8 * 768, generator=g).cuda(); cond[1] = 0
cb, xb = synthetic_inputs(1, 263); cb, xb = cb.cuda(), xb.cuda()
def call(which):
    if which == 0: return A.module_forward(0, x1[..., :262].contiguous(), cond[:, 3 * 768:4 * 768].contiguous(), 500)
    if which == 1: return A.module_forward(1, x1, cond[:, :3 * 768].contiguous(), 500)
    return A.module_forward(2, x1, cond, 500, x2=x2)
for which in (0, 1, 2):
    ref = call(which).clone()
    same = all(torch.equal(call(which), ref) for _ in range(3))
    bad = 0
    for it in range(12):
        B.begin(cb, xb); B.run(50)              # queued on B's stream, returns at once
        o = call(which)                          # runs beside it
        bad += int(not torch.equal(o, ref))
        B.synchronize()
    print(prec, "module", which, "alone reproducible:", same, "| differs beside another handle:", bad, "of 12", flush=True)
