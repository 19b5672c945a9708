"""One ragged batch of the evaluation caller under a profiler (GPU box): the first ragged batch bench.py --eval-items forms from its 64 synthetic
items (T uniform in [60, 300], seed 0; <= 4800 frames: 28 items, 4740 frames in a group of 4864 rows), a few eager ddim50 steps on one stream.
usage: python3 tools/ragged_step.py [steps, default 4] [graph, default 0]      (tools/profile_ragged.sh wraps it in rocprofv3 passes)"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mixermdm_amd.sampler import Sampler
from mixermdm_amd.synthetic import synthetic_state_dict, synthetic_stats, FULL_DIMS
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
graph = len(sys.argv) > 2 and sys.argv[2] == "1"
prec = os.environ.get("RAG_PRECISION", "fp32")
lens_all = [int(v) for v in np.random.RandomState(0).randint(60, 301, size=64)]
lens, rows = [], 0
for T in lens_all:
    if lens and rows + T > 4800:
        break
    lens.append(T); rows += T
sd = synthetic_state_dict(seed=0, std=0.02, bias_std=0.0, **FULL_DIMS); st = synthetic_stats()
s = Sampler(d_heads=8, m_heads=8, max_batch=max(len(lens), 17), max_frames=300, precision=prec, **FULL_DIMS)
s.load_state_dict(sd); s.set_norm_stats(st["mean_hml"], st["std_hml"], st["mean_ih"], st["std_ih"]); s.prepare(); s.set_schedule("ddim50")
g = torch.Generator().manual_seed(100)
cond = torch.randn(len(lens), 8 * 768, generator=g).cuda()
xs = [torch.randn(T, 524, generator=g).cuda() for T in lens]
s.begin_ragged(cond, xs, lens)
s.run(steps, use_graph=graph)
s.synchronize()
ok = bool(torch.isfinite(s.state()["x"]).all().item())
print(json.dumps({"items": len(lens), "frames": rows, "rows": s.rows, "steps": steps, "precision": prec, "outputs_finite": ok}))
s.close()
