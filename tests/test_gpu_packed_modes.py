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
