"""Diagnostic (GPU box): out-of-tolerance statistics of the HIP step against the reference golden at full dims, split by channel class
(rot6d channels go through the ill-conditioned Gram-Schmidt -> quaternion -> axis-angle round trip; the others do not)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
from conftest import fulldims_case
from mixermdm_amd.sampler import Sampler
from mixermdm_amd.synthetic import FULL_DIMS

g, sd, W, stats, inp = fulldims_case()
ch = torch.arange(524) % 262
rot = (ch >= 132) & (ch < 258)


def report(tag, got, ref):
    got, ref = got.cpu().double(), torch.as_tensor(ref).double()
    d = (got - ref).abs()
    bad = d > 2e-4 + 2e-4 * ref.abs()
    print(f"{tag:44s} all {bad.float().mean():.2e}  rot {bad[..., rot].float().mean():.2e} (max {d[..., rot].max():.2e})  "
          f"other {bad[..., ~rot].float().mean():.2e} (max {d[..., ~rot].max():.2e})", flush=True)


for mode in ("fp32", "fp32_split"):
    s = Sampler(d_heads=8, m_heads=8, max_batch=2, max_frames=300, precision=mode, **FULL_DIMS)
    s.load_state_dict(sd)
    s.set_norm_stats(*[t.numpy() for t in stats])
    s.prepare()
    x1, x2, cond, tt = inp["fwd"]
    report(mode + " Mixer.forward", s.module_forward(2, x1, cond, tt, x2=x2), g["fwd"])
    cb, xT, xb2 = inp["step"]
    s.set_schedule("ddim50")
    for i in (32, 0):
        s.begin(cb, xT)
        st = s.state(); st["x"].copy_(xT.cuda()); st["x2"].copy_(xb2.cuda()); torch.cuda.synchronize()
        s.seek(i)
        s.run(1, use_graph=True)
        st = s.state()
        for nm, key in (("x", "sample"), ("x2", "sample2"), ("pred_xstart2", "pred_xstart2")):
            report(f"{mode} ddim50 i={i} {key}", st[nm], g[f"ddim50:i{i}:{key}"])
    c300, x300 = inp["t300"]
    s.set_schedule("ddim1000")
    s.begin(c300, x300)
    s.run(1, use_graph=True)
    st = s.state()
    report(mode + " T300 i=999 sample", st["x"], g["ddim1000:T300:i999:sample"])
    report(mode + " T300 i=999 sample2", st["x2"], g["ddim1000:T300:i999:sample2"])
    xa, xb = inp["late"]
    st["x"].copy_(xa.cuda()); st["x2"].copy_(xb.cuda()); torch.cuda.synchronize()
    s.seek(3)
    s.run(1, use_graph=True)
    st = s.state()
    report(mode + " T300 i=3 sample", st["x"], g["ddim1000:T300:i3:sample"])
    report(mode + " T300 i=3 sample2", st["x2"], g["ddim1000:T300:i3:sample2"])
    s.close()
