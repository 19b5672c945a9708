"""Diagnostic (GPU box): per-row accuracy of Mixer.forward in the bf16 and bf16_fp8 modes against the oracle (no CFG combine)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
from conftest import fulldims_case
from mixermdm_amd.sampler import Sampler
from mixermdm_amd.synthetic import FULL_DIMS
from oracle import mixer as MX
torch.set_num_threads(16); torch.set_grad_enabled(False)
g, sd, W, stats, inp = fulldims_case()
x1, x2, cond, tt = inp["fwd"]
hist = {}
ref = MX.mixer_forward(W, MX.MixerSpec(d_heads=8, m_heads=8), stats, x1, torch.full((x1.shape[0],), tt, dtype=torch.long), cond, x2, hist)
rel = lambda a, b: ((a.cpu().double() - b.double()).pow(2).mean().sqrt() / b.double().pow(2).mean().sqrt()).item()
for mode in ("fp32", "bf16", "bf16_fp8"):
    s = Sampler(d_heads=8, m_heads=8, max_batch=2, max_frames=32, precision=mode, **FULL_DIMS)
    s.load_state_dict(sd); s.set_norm_stats(*[t.numpy() for t in stats]); s.prepare()
    out = s.module_forward(2, x1, cond, tt, x2=x2)
    B = x1.shape[0] // 2
    cfg = 3.5 * out[:B] - 2.5 * out[B:]
    cref = 3.5 * ref[:B] - 2.5 * ref[B:]
    print(f"{mode:9s} Mixer.forward rows rel RMS {rel(out, ref):.3e} | after the CFG combine (3.5 c - 2.5 u) {rel(cfg, cref):.3e} | |c-u|/|c| {rel(ref[:B], ref[B:]):.3f}")
    s.close()
