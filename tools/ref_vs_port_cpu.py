#!/usr/bin/env python3
"""The CPU baseline tied to the reference, once, in the BUILD container (needs /root/reference; never travels to the GPU box).

bench.py's `cpu_baseline` times the oracle (kind "port": oracle/mixer.py::mixer_ddim_step, a restatement of the reference path pinned by
tests/golden).  This script times the REFERENCE ITSELF -- MixerDiffusion.ddim_sample over ClassifierFreeSampleModelX2(Mixer(...)), imported
exactly as tests/golden/make_golden.py imports it (three import stubs, synthetic normaliser files) -- beside the port, on the same threads,
same weights, same inputs: B = 1, T = 300, full model sizes, one ddim1000 step (i = 999, both chains at x_T), and reports their distance.

    python tools/ref_vs_port_cpu.py [threads] [repeats]          -> one JSON line (quoted in BASELINE.md section 3)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden as G          # noqa: E402  (sets up the stubs, chdir()s into a temp cwd with the statistics files, puts /root/reference/src on sys.path)
sys.path.insert(0, ROOT)
import torch                     # noqa: E402
from mixermdm_amd.synthetic import synthetic_state_dict, synthetic_inputs, FULL_DIMS      # noqa: E402
from oracle import mixer as MX, schedule as OS                                          # noqa: E402
from oracle.layers import pe_table                                                      # noqa: E402

threads = int(sys.argv[1]) if len(sys.argv) > 1 else (os.cpu_count() or 1)
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
torch.set_num_threads(threads)
den = dict(latent_dim=1024, ff_size=2048, num_layers=8, num_heads=8, dropout=0.1)
d1 = G.in2INDenoiser(262, mode="individual", **den)
d2 = G.in2INDenoiser(262, mode="interaction", **den)
mix = G.Mixer(d1, d2, nfeats=262, latent_dim=512, ff_size=1024, text_dim=768, n_blocks=4, n_heads=8, mixing_mode=4, store_influence=True,
              force_influence_val=None, mode="eval", align=True)
sd = synthetic_state_dict(seed=0, std=0.02, bias_std=0.02, **FULL_DIMS)
mix.load_state_dict(sd, strict=False)
mix.eval()
cfg = G.ClassifierFreeSampleModelX2(mix, 3.5)
diff = G.make_diffusion("ddim1000")
cond, xT = synthetic_inputs(1, 300)
W = dict(sd)
W["sequence_pos_encoder.pe"] = pe_table(512)
W["denoiser1.sequence_pos_encoder.pe"] = pe_table(1024)
W["denoiser2.sequence_pos_encoder.pe"] = pe_table(1024)
stats = tuple(torch.from_numpy(G.STATS[k]) for k in ("mean_hml", "std_hml", "mean_ih", "std_ih"))
sch = OS.make_schedule("cosine", 1000, "ddim1000")
spec = MX.MixerSpec(d_heads=8, m_heads=8)


def ref_step():
    G.reset_hist(mix)
    with torch.no_grad():
        return diff.ddim_sample(cfg, xT, xT.clone(), torch.tensor([999]), clip_denoised=False, model_kwargs={"mask": None, "cond": cond})


def port_step():
    with torch.no_grad():
        return MX.mixer_ddim_step(W, spec, stats, sch, 3.5, 999, xT, xT.clone(), cond)


def best(fn):
    fn()                                  # first touch
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        ts.append(time.perf_counter() - t0)
    return min(ts), sorted(ts)[len(ts) // 2], out


r_min, r_med, r = ref_step, None, None
r_min, r_med, r = best(ref_step)
p_min, p_med, p = best(port_step)
dx = float((r["sample"] - p[0]).abs().max())
dx2 = float((r["sample2"] - p[1]).abs().max())
print(json.dumps({"what": "one ddim1000 step (i = 999), B = 1, T = 300, D = 1024 / 512, L = 8 / 4, fp32, PyTorch %s CPU" % torch.__version__,
                  "threads": threads, "repeats": reps,
                  "reference_s_per_step": {"min": round(r_min, 4), "median": round(r_med, 4)},
                  "port_s_per_step": {"min": round(p_min, 4), "median": round(p_med, 4)},
                  "port_over_reference": round(p_med / r_med, 3),
                  "max_abs_diff": {"sample": dx, "sample2": dx2}}))
