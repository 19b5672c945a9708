import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch, numpy as np
from conftest import load_golden
from mixermdm_amd.sampler import Sampler
g, w, t = load_golden("single")
s = Sampler(d_latent=16, d_ff=32, d_layers=2, d_heads=int(g["H"]), single_only=True, cfg_scale=3.5, max_batch=2, max_frames=16)
s.load_state_dict({"denoiser1." + k: v for k, v in w("ind.").items()})
s.prepare()
cond, xT = t("cond"), t("x_T")
for strat in ["ddim50", "ddim20"]:
  s.set_schedule(strat)
  for graph in [False, True]:
    out = s.sample(cond, xT, use_graph=graph)
    d = np.abs(out.cpu().numpy() - g[f"loop:{strat}:output"])
    print(strat, graph, "sample()", d.mean(), d.max())
    s.begin(cond, xT); s.run(None, graph); st = s.state()
    print("   manual x finite", torch.isfinite(st["x"]).all().item(), "px", torch.isfinite(st["pred_xstart"]).all().item())
    d = np.abs(st["pred_xstart"].cpu().numpy() - g[f"loop:{strat}:output"]); print("   ", d.mean(), d.max())
