"""The low-precision handles store their stack weights in MFMA fragment order and run the packed GEMM kernels (gemm_splitw_kernel,
gemm_bf16w_kernel).  Model-level check: the same sampler steps with MMDM_NO_PACK=1 (weights left in rows / planes, staged kernels) must give
the same bits.  The switch is read once per process, so each side runs in a child process of its own (started with subprocess: fork + exec of
a fresh interpreter, never an exec from this GPU-initialised one)."""
import hashlib
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, hashlib, torch
sys.path.insert(0, %r)
from mixermdm_amd.sampler import Sampler
from mixermdm_amd.synthetic import synthetic_state_dict, synthetic_stats, synthetic_inputs, FULL_DIMS
from mixermdm_amd import load_library
prec = sys.argv[1]
sd = synthetic_state_dict(seed=0, std=0.02, bias_std=0.02, **FULL_DIMS)
st = synthetic_stats()
B, T = 2, 40
cond, xT = synthetic_inputs(B, T)
s = Sampler(d_heads=8, m_heads=8, max_batch=B, max_frames=T, precision=prec, **FULL_DIMS)
s.load_state_dict(sd); s.set_norm_stats(st["mean_hml"], st["std_hml"], st["mean_ih"], st["std_ih"]); s.prepare(); s.set_schedule("ddim50")
s.begin(cond, xT); s.run(2, use_graph=False)
h = hashlib.sha256()
for k in ("x", "x2", "model_out"):
    h.update(s.state()[k].contiguous().cpu().numpy().tobytes())
print("DIGEST", h.hexdigest(), bool(torch.isfinite(s.state()["x"]).all()))
'''


def _run(prec, no_pack):
    env = dict(os.environ)
    env.pop("MMDM_NO_PACK", None)
    env.pop("MMDM_SPLIT_NO_PACK", None)
    if no_pack:
        env["MMDM_NO_PACK"] = "1"
    r = subprocess.run([sys.executable, "-c", CHILD % ROOT, prec], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("DIGEST")][-1].split()
    assert line[2] == "True"
    return line[1]


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["fp32_split", "bf16", "bf16_fp8"])
def test_packed_weight_kernels_give_the_bits_of_the_staged_kernels_in_the_sampler(prec):
    assert _run(prec, no_pack=False) == _run(prec, no_pack=True)


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32_split", "bf16_fp8"])
def test_conditioning_projections_on_the_split_kernel_match_the_fp32_kernel(precision):
    """Low-precision handles run the AdaLN conditioning projections (silu(time + text) against [L n_ada 2D, D]) on the fp32-split kernel with the
    weights as two packed fp16 planes; MMDM_NO_SPLIT_COND=1 (read by mmdm_create: a child process here) keeps them on the fp32 MFMA kernel.
    Both are fp32-accurate, so the sampled chain of the mode must be as far from the fp32 handle's chain with the switch as without it (a free-running
    sampler amplifies any rounding difference: the yardstick is the mode's own distance from fp32, not zero)."""
    import os
    import subprocess
    import sys
    import tempfile
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, torch
sys.path.insert(0, %r)
from mixermdm_amd.sampler import Sampler
from mixermdm_amd.synthetic import synthetic_state_dict, synthetic_stats
DIMS = dict(d_latent=256, d_ff=512, d_layers=2, m_latent=128, m_ff=256, m_layers=2)
sd = synthetic_state_dict(seed=11, std=0.05, bias_std=0.02, **DIMS); st = synthetic_stats()
g = torch.Generator().manual_seed(5)
c, x = torch.randn(2, 8 * 768, generator=g).cuda(), torch.randn(2, 48, 524, generator=g).cuda()
outs = {}
for prec in ("fp32", %r):
    s = Sampler(d_heads=2, m_heads=2, max_batch=2, max_frames=64, precision=prec, **DIMS)
    s.load_state_dict(sd); s.set_norm_stats(st["mean_hml"], st["std_hml"], st["mean_ih"], st["std_ih"]); s.prepare(); s.set_schedule("ddim10")
    outs[prec] = s.sample(c, x).cpu()
    assert torch.isfinite(outs[prec]).all()
torch.save(outs, sys.argv[1])
print("COND_OK")
''' % (root, precision)
    dist = {}
    with tempfile.TemporaryDirectory() as td:
        for flag in ("0", "1"):
            path = os.path.join(td, "out%s.pt" % flag)
            env = dict(os.environ, MMDM_NO_SPLIT_COND=flag)
            r = subprocess.run([sys.executable, "-c", code, path], capture_output=True, text=True, timeout=600, env=env, cwd=root)
            assert r.returncode == 0 and "COND_OK" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])
            o = torch.load(path)
            ref, got = o["fp32"].double(), o[precision].double()
            dist[flag] = float((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    # "0": projections on the fp32-split kernel (shipped), "1": on the fp32 MFMA kernel
    print("distance from the fp32 chain after 10 steps:", precision, dist)
    assert dist["0"] <= 3.0 * dist["1"] + 1e-6, dist
