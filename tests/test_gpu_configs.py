"""GPU: the BASELINE.json configurations that had no test at their own sizes ("configs_untested"), and a free-running loop of
the headline model against the oracle.

  * configs[1]  single-person in2IN, D=1024/F=2048/L=8/H=8, T=196, ddim1000 (src/models/in2in.py:285-356, the single-chain sampler
                src/models/utils/gaussian_diffusion.py:946-1069): two steps at B=2 against oracle.mixer.cfg_single + ddim_update (fp32 and the
                float64 yardstick), and the full B=32 batch through size-independent properties;
  * configs[4]  the packed bf16 / fp8 GEMM kernels the bf16 and bf16_fp8 samplers launch, at M = 19 200 against float64 of the rounded /
                decoded operands with the instantiation asserted, and the 64-motions-per-GPU shard of the B=512 job in bf16_fp8;
  * configs[2]  one FREE-RUNNING ddim50 loop at the full model sizes, B=1, T=300 against the oracle (mean |d| <= 1e-4,
                99.9th percentile <= 1e-2), fp32 and fp32_split, next to the float64 yardstick of the same loop.
"""
import math
import os
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import mixer as MX            # noqa: E402  (checker only)
from oracle import schedule as OS         # noqa: E402
from oracle.layers import pe_table        # noqa: E402
from test_gpu_kernels import assert_close, rnd, dev   # noqa: E402
from parity_tol import STEP_TOL, YARD_FACTOR_LOOP, yardstick, to64, record, _quant      # noqa: E402

SINGLE = dict(d_latent=1024, d_ff=2048, d_layers=8)
T1 = 196


class _Threads:
    def __enter__(self):
        self.n = torch.get_num_threads()
        torch.set_num_threads(min(16, os.cpu_count() or 1))

    def __exit__(self, *a):
        torch.set_num_threads(self.n)


# ---------------------------------------------------------------------------------------------------
# configs[1]
# ---------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def single():
    from mixermdm_amd.sampler import Sampler
    from mixermdm_amd.synthetic import synthetic_state_dict
    sd = synthetic_state_dict(seed=7, std=0.02, bias_std=0.02, single_only=True, m_latent=0, m_ff=0, m_layers=0, **SINGLE)
    s = Sampler(d_heads=8, single_only=True, max_batch=32, max_frames=T1, **SINGLE)
    s.load_state_dict(sd)
    s.prepare()
    s.set_schedule("ddim1000")
    W = dict(sd)
    W["denoiser1.sequence_pos_encoder.pe"] = pe_table(1024)
    yield s, W
    s.close()


def test_configs1_single_person_two_ddim1000_steps_vs_oracle_at_full_dims(single):
    """B=2, T=196, D=1024, L=8: steps i = 999 and 998 of the single-chain sampler, free-running, against ClassifierFreeSampleModel +
    ddim_sample restated in the oracle -- no geometry on this path, so the plain step tolerance applies to every element."""
    from mixermdm_amd.synthetic import synthetic_inputs
    s, W = single
    cond, xT = synthetic_inputs(2, T1, seed_cond=71, seed_x=72, single=True)
    sch = OS.make_schedule("cosine", 1000, "ddim1000")
    W64 = to64(W)
    ref32, ref64 = [], []
    with _Threads(), torch.no_grad():
        x, xd = xT, xT.double()
        for i in (999, 998):
            ts = torch.full((2,), sch.timestep_map[i], dtype=torch.long)
            x0 = MX.cfg_single(W, "denoiser1.", "individual", 3.5, x, ts, cond, 8)
            x = MX.ddim_update(sch, i, x, x0)
            ref32.append({"x": x, "pred_xstart": x0})
            # the float64 step starts from the fp32 oracle's state of the previous step, like the HIP step below
            x0d = MX.cfg_single(W64, "denoiser1.", "individual", 3.5, xd, ts, cond.double(), 8)
            ref64.append({"x": MX.ddim_update(sch, i, xd, x0d), "pred_xstart": x0d})
            xd = x.double()
    s.begin(cond, xT)
    for k, i in enumerate((999, 998)):
        if k:           # teacher-forced from the oracle's state, so step 2 is compared on identical inputs
            s.state()["x"].copy_(ref32[0]["x"].to(s.device))
            torch.cuda.synchronize()
            s.seek(i)
        s.run(1, use_graph=True)
        st = s.state()
        out = {"x": st["x"].clone(), "pred_xstart": st["pred_xstart"].clone()}
        for nm in ("x", "pred_xstart"):
            assert_close(out[nm], ref32[k][nm], atol=STEP_TOL["atol"], rtol=STEP_TOL["rtol"], what=f"configs[1] i={i} {nm}")
        # no geometry on this path: every channel class (they are just output columns here) at the plain factor
        yardstick(out, ref32[k], ref64[k], f"configs[1] single-person T=196 B=2 i={i}", factor_posvel=3.0)


def test_configs1_full_batch_is_finite_deterministic_graph_equals_eager_and_row_independent(single):
    """The B=32, T=196 workload itself (M = 2*32*196 = 12 544 GEMM rows): finite; bit-identical across runs and between graph replay and
    eager launches; motion k of the batch == the same motion sampled alone."""
    from mixermdm_amd.synthetic import synthetic_inputs
    s, _ = single
    cond, xT = synthetic_inputs(32, T1, seed_cond=73, seed_x=74, single=True)

    def run(c, x, graph=True, steps=3):
        s.begin(c, x)
        s.run(steps, use_graph=graph)
        st = s.state()
        return st["x"].clone(), st["pred_xstart"].clone()

    a, b, c = run(cond, xT), run(cond, xT), run(cond, xT, graph=False)
    assert a[0].shape == (32, T1, 262)
    for u, v, w in zip(a, b, c):
        assert torch.isfinite(u).all() and torch.equal(u, v) and torch.equal(u, w)
    for k in (0, 13, 31):
        one = run(cond[k:k + 1], xT[k:k + 1])
        assert torch.equal(one[0][0], a[0][k]) and torch.equal(one[1][0], a[1][k]), k


# ---------------------------------------------------------------------------------------------------
# configs[4]: the packed kernels at the production shape, and the 64-motion shard
# ---------------------------------------------------------------------------------------------------
M_FULL = 19200
PACKED_SHAPES = [(3072, 1024, "bias"), (2048, 1024, "gelu"), (1024, 1024, "resid"), (1024, 2048, "resid"), (1536, 512, "bias"), (512, 1024, "resid")]


def _f64_ref(x, w, b, epi, r):
    y = F.linear(x.double(), w.double(), b.double())
    if epi == "gelu":
        y = F.gelu(y)
    elif epi == "resid":
        y = y + r.double()
    return y


@pytest.mark.parametrize("N,K,epi", PACKED_SHAPES)
def test_packed_bf16_production_gemm_vs_float64_of_the_rounded_operands(N, K, epi):
    """gemm_bf16w (weights in fragment order, W straight from global memory) -- the kernel the bf16 sampler and the bf16 projections of the
    bf16_fp8 sampler run -- at M = 19 200 against float64 of the bf16-rounded operands; the instantiation is asserted by name."""
    from mixermdm_amd import ops
    from mixermdm_amd._lib import load_library
    d = dev()
    x, w, b = rnd(301, M_FULL, K), rnd(302, N, K, scale=1 / math.sqrt(K)), rnd(303, N)
    r = rnd(304, M_FULL, N) if epi == "resid" else None
    xb, wb = ops.to_bf16(x.to(d)), ops.to_bf16(w.to(d))
    with _Threads():
        ref = _f64_ref(xb.float().cpu(), wb.float().cpu(), b, epi, r)
    got = ops.linear_bf16(xb, ops.pack_weight_frag(wb), b.to(d), epi, r.to(d) if r is not None else None, packed=True)
    kern = load_library().mmdm_last_gemm_kernel().decode()
    assert kern in {"gemm_bf16w<14,42>", "gemm_bf16w<14,41>"}, kern
    assert_close(got, ref.float(), atol=2e-5 * math.sqrt(K / 1024), rtol=1e-5, what=f"packed linear_bf16 19200x{N}x{K} {epi} on {kern}")


@pytest.mark.parametrize("N,K,epi", PACKED_SHAPES[:4])
def test_packed_fp8_production_gemm_vs_float64_of_the_decoded_operands(N, K, epi):
    """gemm_fp8w at M = 19 200 against float64 of the DECODED e4m3 operands x scales (exact statement about the quantised values)."""
    from mixermdm_amd import ops
    from mixermdm_amd._lib import load_library
    d = dev()
    x, w, b = rnd(311, M_FULL, K), rnd(312, N, K, scale=1 / math.sqrt(K)), rnd(313, N)
    r = rnd(314, M_FULL, N) if epi == "resid" else None
    xq, xs = ops.quantize_rows_fp8(x.to(d))
    wq, ws = ops.quantize_rows_fp8(w.to(d))
    with _Threads():
        xd, wd = xq.cpu().float().double() * xs.cpu().double()[:, None], wq.cpu().float().double() * ws.cpu().double()[:, None]
        ref = _f64_ref(xd, wd, b, epi, r)
    got = ops.linear_fp8(xq, xs, ops.pack_weight_frag(wq), ws, b.to(d), epi, r.to(d) if r is not None else None, packed=True)
    kern = load_library().mmdm_last_gemm_kernel().decode()
    assert kern == "gemm_fp8w<14,41>", kern
    assert_close(got, ref.float(), atol=2e-4, rtol=1e-4, what=f"packed linear_fp8 19200x{N}x{K} {epi} on {kern}")


def test_configs4_shard_of_64_motions_bf16_fp8_is_finite_deterministic_and_row_independent():
    """BASELINE configs[4] (B = 512 over 8 GPUs): the per-GPU shard of 64 motions at T=300 in the bf16_fp8 mode -- M = 76 800 GEMM rows.
    Finite, bit-identical across graph replays, and motions 0 / 29 / 63 equal the same motions sampled alone (per-row fp8 activation scales
    and per-row attention make a row independent of its batch)."""
    from mixermdm_amd.sampler import Sampler
    from mixermdm_amd.synthetic import synthetic_state_dict, synthetic_stats, synthetic_inputs, FULL_DIMS
    sd = synthetic_state_dict(seed=0, std=0.02, bias_std=0.02, **FULL_DIMS)
    st = synthetic_stats()
    B4, T = 64, 300
    cond, xT = synthetic_inputs(B4, T, seed_cond=81, seed_x=82)
    s = Sampler(d_heads=8, m_heads=8, max_batch=B4, max_frames=T, precision="bf16_fp8", **FULL_DIMS)
    s.load_state_dict(sd)
    s.set_norm_stats(st["mean_hml"], st["std_hml"], st["mean_ih"], st["std_ih"])
    s.prepare()
    s.set_schedule("ddim1000")

    def run(c, x, steps=2):
        s.begin(c, x)
        s.run(steps, use_graph=True)
        stt = s.state()
        return {k: stt[k].clone() for k in ("x", "x2", "pred_xstart2")}

    a, b = run(cond, xT), run(cond, xT)
    for k in a:
        assert a[k].shape == (B4, T, 524) and torch.isfinite(a[k]).all() and torch.equal(a[k], b[k]), k
    for k in (0, 29, 63):
        one = run(cond[k:k + 1], xT[k:k + 1])
        assert torch.equal(one["x"][0], a["x"][k]) and torch.equal(one["x2"][0], a["x2"][k]), k
    s.close()


# ---------------------------------------------------------------------------------------------------
# configs[2]: a free-running loop at the real sizes
# ---------------------------------------------------------------------------------------------------
def test_free_running_ddim1000_loop_at_full_dims_vs_oracle():
    """The headline's OWN loop, free-running: MixerDiffusion.ddim_sample_loop at ddim1000 (src/models/utils/gaussian_diffusion.py:1769-1899) from
    x_T to the motion -- 1000 steps, no teacher forcing -- at the full model sizes (D = 1024 / 512, 8 + 8 + 4 blocks), B = 1, T = 48 (the length is
    what bounds the oracle's CPU time: 0.19 s per step at T = 64 on the GPU box's host, a quarter of the whole GPU suite), HIP fp32 and fp32_split against the fp32 oracle's free-running loop with SURVEY 8c's
    end-to-end bound for 1000 steps: mean |d| <= 1e-3, 99.9th percentile <= 5e-2 (the survey's probe: a 1e-6 input perturbation grows to 1.7e-2
    over 1000 steps of this sampler).  The teacher-forced first / last 20 steps at T = 300 stay in tests/test_gpu_headline.py."""
    import time
    from conftest import fulldims_case
    from mixermdm_amd.sampler import Sampler
    from mixermdm_amd.synthetic import FULL_DIMS, synthetic_inputs
    g, sd, W, stats, inp = fulldims_case()
    T = 48
    cond, xT = synthetic_inputs(1, T)
    sch = OS.make_schedule("cosine", 1000, "ddim1000")
    spec = MX.MixerSpec(d_heads=8, m_heads=8)
    names = ("pred_xstart2", "x", "x2")
    t0 = time.time()
    with _Threads(), torch.no_grad():
        ref32 = dict(zip(names, MX.mixer_ddim_loop(W, spec, stats, sch, 3.5, xT, cond)))
    cpu_s = time.time() - t0
    for mode in ("fp32", "fp32_split"):
        s = Sampler(d_heads=8, m_heads=8, max_batch=1, max_frames=T, precision=mode, **FULL_DIMS)
        s.load_state_dict(sd)
        s.set_norm_stats(*[t.numpy() for t in stats])
        s.prepare()
        s.set_schedule("ddim1000")
        out_final = s.sample(cond, xT)
        st = s.state()
        out = {"pred_xstart2": out_final, "x": st["x"].clone(), "x2": st["x2"].clone()}
        figures = {"oracle_cpu_seconds": cpu_s}
        for nm in names:
            assert torch.isfinite(out[nm]).all(), (mode, nm)
            d = (out[nm].cpu().double() - ref32[nm].double()).abs()
            q = _quant(d, (0.5, 0.99, 0.999, 1.0))
            figures[nm] = {"mean": float(d.mean()), "p50": q[0], "p99": q[1], "p99.9": q[2], "max": q[3]}
            print(f"{mode} ddim1000 free-running {nm}: mean |d| {d.mean():.3e}, p99.9 {q[2]:.3e}, max {q[3]:.3e}")
        record(f"free-running ddim1000 B=1 T={T} [{mode}] vs fp32 oracle loop", kind="loop", **figures)
        for nm in names:
            assert figures[nm]["mean"] <= 1e-3 and figures[nm]["p99.9"] <= 5e-2, (mode, nm, figures[nm])
        s.close()


def test_free_running_ddim50_loop_at_full_dims_vs_oracle():
    """ddim50 from x_T to the motion, B=1, T=300, D=1024/512, 8+8+4 blocks, NO teacher forcing: HIP fp32 and fp32_split against the fp32
    oracle's own free-running loop with SURVEY 8c's end-to-end bound (mean |d| <= 1e-4, 99.9th percentile <= 1e-2), and against the
    float64 oracle's loop next to the fp32 oracle's distance from it (what 50 steps of fp32 rounding cost on this function)."""
    from conftest import fulldims_case
    from mixermdm_amd.sampler import Sampler
    from mixermdm_amd.synthetic import FULL_DIMS
    g, sd, W, stats, inp = fulldims_case()
    cond, xT = inp["t300"]
    sch = OS.make_schedule("cosine", 1000, "ddim50")
    spec = MX.MixerSpec(d_heads=8, m_heads=8)
    names = ("pred_xstart2", "x", "x2")
    with _Threads(), torch.no_grad():
        ref32 = dict(zip(names, MX.mixer_ddim_loop(W, spec, stats, sch, 3.5, xT, cond)))
        ref64 = dict(zip(names, MX.mixer_ddim_loop(to64(W), spec, tuple(t.double() for t in stats), sch, 3.5, xT.double(), cond.double())))
    for mode in ("fp32", "fp32_split"):
        s = Sampler(d_heads=8, m_heads=8, max_batch=1, max_frames=300, precision=mode, **FULL_DIMS)
        s.load_state_dict(sd)
        s.set_norm_stats(*[t.numpy() for t in stats])
        s.prepare()
        s.set_schedule("ddim50")
        out_final = s.sample(cond, xT)
        st = s.state()
        out = {"pred_xstart2": out_final, "x": st["x"].clone(), "x2": st["x2"].clone()}
        figures = {}
        for nm in names:
            assert torch.isfinite(out[nm]).all(), (mode, nm)
            d = (out[nm].cpu().double() - ref32[nm].double()).abs()
            q = _quant(d, (0.5, 0.99, 0.999, 1.0))
            figures[nm] = {"mean": float(d.mean()), "p50": q[0], "p99": q[1], "p99.9": q[2], "max": q[3]}
            print(f"{mode} ddim50 free-running {nm}: mean |d| {d.mean():.3e}, p99.9 {q[2]:.3e}, max {q[3]:.3e}")
        record(f"free-running ddim50 B=1 T=300 [{mode}] vs fp32 oracle loop", kind="loop", **figures)
        # 50 steps of compounded divergence, one sample of a chaotic process: the looser loop factor for every class
        yardstick(out, ref32, ref64, f"free-running ddim50 B=1 T=300 [{mode}]", factor=YARD_FACTOR_LOOP, factor_posvel=YARD_FACTOR_LOOP)
        for nm in names:
            assert figures[nm]["mean"] <= 1e-4 and figures[nm]["p99.9"] <= 1e-2, (mode, nm, figures[nm])
        if mode == "fp32":
            direct = out_final.clone()
        s.close()
    # The reference's own entry point for this call shape -- MixerMDM(cfg, ...)(batch) at B = 1, ddim50 (src/scripts/infer/mixermdm.py:73,
    # 117-124: the infer script; src/evaluation/datasets.py:100-116: forward_test per item) -- through the facade over configs/models/*.yaml:
    # the same kernels behind the reference API, so the motion is the Sampler's bit for bit and carries the figures asserted above; the
    # history lists have the reference's lengths and [2B, T, C] shapes.
    import os
    from mixermdm_amd.configs import get_config
    from mixermdm_amd.models import MixerMDM
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    m = MixerMDM(get_config(os.path.join(root, "configs", "models", "MixerMDM.yaml")), num_frames=300, sampling_strategy="ddim50", config_root=root)
    m.load_state_dict({"mixing." + k: v for k, v in sd.items()})
    m.set_norm_stats(*[t.numpy() for t in stats])
    m = m.to("cuda:0").eval()
    batch = {"cond": cond.cuda(), "x_T": xT.cuda(), "motion_lens": torch.tensor([[300]])}
    full = m(batch)
    assert torch.equal(full["output"], direct), "MixerMDM.forward differs from the Sampler it wraps"
    assert len(full["influence_i1"]) == 50 and full["influence_i1"][0].shape == (2, 300, 262) and len(full["out_influenced"]) == 50 and full["out1"][0].shape == (2, 300, 524)
    test = m.forward_test(batch)
    assert torch.equal(test["output"], direct) and set(test) == {"output", "influence_i1", "influence_i2"}
