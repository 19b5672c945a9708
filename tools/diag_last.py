"""Diagnostic (GPU box): late ddim1000 steps, HIP vs oracle, teacher-forced; where are the out-of-tolerance elements?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
from conftest import fulldims_case
from mixermdm_amd.sampler import Sampler
from mixermdm_amd.synthetic import FULL_DIMS
from oracle import mixer as MX, schedule as OS
torch.set_num_threads(16)
torch.set_grad_enabled(False)
g, sd, W, stats, inp = fulldims_case()
cond, xT = inp["t300"]
mode = sys.argv[1] if len(sys.argv) > 1 else "fp32"
s = Sampler(d_heads=8, m_heads=8, max_batch=2, max_frames=300, precision=mode, **FULL_DIMS)
s.load_state_dict(sd); s.set_norm_stats(*[t.numpy() for t in stats]); s.prepare()
s.set_schedule("ddim1000")
s.begin(cond, xT)
s.run(980, use_graph=True)
st = s.state()
x, x2 = st["x"].cpu().clone(), st["x2"].cpu().clone()
sch = OS.make_schedule("cosine", 1000, "ddim1000")
spec = MX.MixerSpec(d_heads=8, m_heads=8)
ch = torch.arange(524)
for i in range(19, 12, -1):
    hist = {}
    rx, rx2, p1, p2 = MX.mixer_ddim_step(W, spec, stats, sch, 3.5, i, x, x2, cond, hist)
    st = s.state(); st["x"].copy_(x.cuda()); st["x2"].copy_(x2.cuda()); torch.cuda.synchronize(); s.seek(i); s.run(1, use_graph=True)
    st = s.state()
    line = f"i={i:3d}"
    for nm, ref in (("x", rx), ("x2", rx2), ("pred_xstart", p1), ("pred_xstart2", p2), ("model_out", None)):
        if ref is None:
            continue
        d = (st[nm].cpu() - ref).abs()
        bad = d > 2e-4 + 2e-4 * ref.abs()
        line += f" | {nm} {bad.float().mean():.1e} max {d.max():.1e}"
        if nm == "x" and bad.any():
            idx = bad.nonzero()
            pers = (idx[:, 2] >= 262).float().mean().item()
            c = idx[:, 2] % 262
            cls = [((c < 66)).float().mean().item(), ((c >= 66) & (c < 132)).float().mean().item(), ((c >= 132) & (c < 258)).float().mean().item(), (c >= 258).float().mean().item()]
            line += f" [person2 share {pers:.2f}; pos/vel/rot/feet {cls[0]:.2f}/{cls[1]:.2f}/{cls[2]:.2f}/{cls[3]:.2f}; frames {idx[:,1].min().item()}..{idx[:,1].max().item()}; ref rms {ref.pow(2).mean().sqrt():.2f} coef {sch.sqrt_recipm1_alphas_cumprod[i]:.3e}]"
    print(line, flush=True)
    ad = hist["align_diag"]
    for pp in range(2):
        d = ad[pp]
        print(f"      align person {pp+1}: rows(cond,uncond) disp_target {d['disp_target'].tolist()} disp_moved {d['disp_moved'].tolist()} w {d['w'].tolist()} reach {d['reach'].tolist()}")
    for pp, d in enumerate(hist["center_diag"]):
        print(f"      center person {pp+1}: across {d['across'].tolist()} fwd {d['fwd'].tolist()} w {d['w'].tolist()} reach {d['reach'].tolist()}")
    x, x2 = rx, rx2
