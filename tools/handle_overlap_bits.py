"""scratch: do shared handles give the parent's bits in fp32_split at the real sizes?  sequentially, then overlapped."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mixermdm_amd.sampler import Sampler
from mixermdm_amd.synthetic import synthetic_state_dict, synthetic_stats, synthetic_inputs, FULL_DIMS
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32_split"
sd = synthetic_state_dict(seed=0, std=0.02, bias_std=0.0, **FULL_DIMS); st = synthetic_stats()
s = Sampler(d_heads=8, m_heads=8, max_batch=1, max_frames=300, precision=prec, **FULL_DIMS)
s.load_state_dict(sd); s.set_norm_stats(st["mean_hml"], st["std_hml"], st["mean_ih"], st["std_ih"]); s.prepare()
if os.environ.get("TST0") == "1":
    from mixermdm_amd._lib import diag
    diag("bf16_tst", 0); diag("split_tst", 0)
NP = int(os.environ.get('NPOOL', '4'))
def fresh():
    t = Sampler(d_heads=8, m_heads=8, max_batch=1, max_frames=300, precision=prec, **FULL_DIMS)
    t.load_state_dict(sd); t.set_norm_stats(st["mean_hml"], st["std_hml"], st["mean_ih"], st["std_ih"]); t.prepare()
    return t
pool = [s] + [(fresh() if os.environ.get("INDEP") == "1" else s.share()) for _ in range(NP - 1)]
for p in pool: p.set_schedule("ddim50")
items = [tuple(t.cuda() for t in synthetic_inputs(1, T, seed_cond=T, seed_x=T + 1)) for T in (181, 97, 263, 140, 181, 97, 263, 140)]
ref = [s.sample(c, x) for c, x in items]
G = os.environ.get("PROBE_EAGER") != "1"
if os.environ.get("PRECAPTURE") == "1":          # every handle captures every shape alone first: the rounds below then only REPLAY
    for p in pool:
        for c, x in items[:4]:
            p.sample(c, x)
    torch.cuda.synchronize()
for rnd in range(3):
    outs = [torch.empty_like(x) for c, x in items]
    torch.cuda.synchronize()
    for i, (c, x) in enumerate(items):
        pool[i % NP].enqueue(c, x, outs[i], use_graph=G)
    torch.cuda.synchronize()
    print("overlapped round", rnd, ["eq" if torch.equal(o, r) else "DIFF %.2e" % (o - r).abs().max().item() for o, r in zip(outs, ref)], flush=True)
