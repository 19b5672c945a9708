"""Where does the HIP fp32 path's extra distance from float64 come from?  (CPU-only experiment; DESIGN.md section 3 / LAB_NOTES.md)

The float64 yardstick of tests/parity_tol.py shows the HIP step 4-10x further from the float64 oracle than the CPU fp32 oracle at matching
quantiles.  Hypothesis: v_mfma_f32_32x32x2_f32 accumulates an output element as ONE k-ordered fp32 fma chain (K = 1024-2048 terms), whereas
the CPU GEMM the oracle runs on (oneDNN / MKL) keeps 16-lane vector accumulators per k-block, i.e. many short chains summed at the end.
This script re-runs one oracle step with every F.linear replaced by a strict k-ordered fp32 chain (optionally cut into `--chunk`-term
partial sums added at the end, the two-level form the GEMM kernels could use) and reports the same quantile ratios.
"""
import argparse, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import torch.nn.functional as F

ap = argparse.ArgumentParser()
ap.add_argument("--T", type=int, default=48)
ap.add_argument("--chunk", type=int, default=0, help="0 = one chain over all of K; c > 0: chains of c terms, partial sums added in order")
ap.add_argument("--i", type=int, default=500)
args = ap.parse_args()

from oracle import mixer as MX, schedule as OS
from conftest import fulldims_case
from parity_tol import to64, _quant, channel_classes
from mixermdm_amd.synthetic import synthetic_inputs

g, sd, W, stats, inp = fulldims_case()
cond, xT = synthetic_inputs(1, args.T, seed_cond=41, seed_x=42)
x2 = torch.randn(1, args.T, 524, generator=torch.Generator().manual_seed(43))
sch = OS.make_schedule("cosine", 1000, "ddim1000")
spec = MX.MixerSpec(d_heads=8, m_heads=8)
torch.set_num_threads(8)
real_linear = F.linear


def chain_linear(x, w, b=None):
    """y[m, n] = (((b? no: sum first) ... strict k-ordered fp32 accumulation, products rounded once (fma), like the MFMA chain"""
    if x.dtype != torch.float32 or w.shape[1] < 64:
        return real_linear(x, w, b)
    shp = x.shape[:-1]
    x2d = x.reshape(-1, x.shape[-1])
    K = x2d.shape[1]
    c = args.chunk or K
    total = None
    for k0 in range(0, K, c):
        acc = torch.zeros(x2d.shape[0], w.shape[0], dtype=torch.float32)
        if k0 == 0 and b is not None:
            acc += b                       # the kernels start the accumulators from the bias
        for k in range(k0, min(K, k0 + c)):
            acc = torch.addcmul(acc, x2d[:, k:k + 1], w[:, k][None, :])          # one rounding per term on CPUs with FMA in addcmul? (mul + add: two roundings; upper bound)
        total = acc if total is None else total + acc
    return total.reshape(*shp, w.shape[0])


with torch.no_grad():
    t0 = time.time()
    ref32 = MX.mixer_ddim_step(W, spec, stats, sch, 3.5, args.i, xT, x2, cond, {})
    ref64 = MX.mixer_ddim_step(to64(W), spec, tuple(t.double() for t in stats), sch, 3.5, args.i, xT.double(), x2.double(), cond.double(), {})
    print(f"oracle fp32 + f64: {time.time() - t0:.1f} s", flush=True)
    F.linear = chain_linear
    t0 = time.time()
    seq = MX.mixer_ddim_step(W, spec, stats, sch, 3.5, args.i, xT, x2, cond, {})
    F.linear = real_linear
    print(f"k-ordered chain (chunk {args.chunk or 'K'}): {time.time() - t0:.1f} s", flush=True)
for nm, a, b, c in zip(("x", "x2", "pred_xstart", "pred_xstart2"), seq, ref32, ref64):
    for cname, sel in channel_classes(524).items():
        qs, qc = _quant((a.double() - c).abs()[..., sel]), _quant((b.double() - c).abs()[..., sel])
        print(f"{nm:13s} {cname:7s} chain-vs-f64 " + " ".join(f"{v:.2e}" for v in qs) + " | cpu32-vs-f64 " + " ".join(f"{v:.2e}" for v in qc) +
              " | ratio " + " ".join(f"{h / max(cc, 1e-6):.1f}" for h, cc in zip(qs, qc)))
