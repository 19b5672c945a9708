"""Estimate of what finer-grained stream mixing buys: N handles x (16/N) motions concurrently vs one handle x 16 motions."""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mixermdm_amd.sampler import Sampler
from mixermdm_amd.synthetic import synthetic_state_dict, synthetic_stats, synthetic_inputs, FULL_DIMS
T, steps = 300, 12
sd = synthetic_state_dict(seed=0, std=0.02, **FULL_DIMS)
st = synthetic_stats()
def mk(B):
    s = Sampler(d_heads=8, m_heads=8, max_batch=B, max_frames=T, **FULL_DIMS)
    s.load_state_dict(sd); s.set_norm_stats(st["mean_hml"], st["std_hml"], st["mean_ih"], st["std_ih"]); s.prepare(); s.set_schedule("ddim1000")
    c, x = synthetic_inputs(B, T); s.begin(c, x); s.run(2, True); s.synchronize()
    return s
for nh in [int(a) for a in (sys.argv[1:] or ["1", "2", "4"])]:
    hs = [mk(16 // nh) for _ in range(nh)]
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for k in range(steps):
            for h in hs: h.run(1, True)
        for h in hs: h.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps)
    print(f"{nh} handle(s) x B={16//nh}: {best*1e3:.2f} ms per 16-motion step", flush=True)
    for h in hs: h.close()
    del hs
