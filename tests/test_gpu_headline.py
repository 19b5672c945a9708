"""GPU: the headline configuration VALUE-checked -- full model dimensions (D=1024/F=2048/L=8/H=8, mixer 512/1024/4/8), T=300, the
ddim1000 schedule -- in the native fp32 mode and in fp32_split, against (1) the CPU oracle and (2) outputs captured from the REFERENCE
itself at these sizes (tests/golden/fulldims.npz), plus float64 checks of the GEMM instantiations the B=16 step actually launches
(M = 19 200 rows), with the launched kernel asserted through the mmdm_last_gemm_kernel() debug getter.

Tolerance of a step (tests/parity_tol.py): all but 0.2 % of the elements within atol 2e-4 + rtol 2e-4 of the fp32 oracle, none beyond 5e-2
(a handful of near-degenerate joints amplify rounding through the rot6d -> quaternion round trip), the tolerance of an element never
smaller than 12 x what fp32 arithmetic costs the CPU oracle itself in that (sample, person, channel class) group -- measured against a
float64 run of the same oracle; and the float64 yardstick: the HIP path is as far from float64 as the CPU fp32 oracle is, within 3 x.
"""
import math
import os
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import mixer as MX            # noqa: E402  (checker only)
from oracle import schedule as OS         # noqa: E402
from test_gpu_kernels import assert_close, rnd, dev   # noqa: E402
from parity_tol import STEP_TOL, HARD_JOINTS, compare_step, yardstick, yardstick_sequence, to64, record      # noqa: E402,F401
from conftest import fulldims_case         # noqa: E402

MODES = ["fp32", "fp32_split"]
ORACLE_THREADS = 16          # B=1..2 GEMMs on the GPU box's 128-thread host run fastest on 16 threads (bench.py calibrates the same way)


class _Threads:
    def __enter__(self):
        self.n = torch.get_num_threads()
        torch.set_num_threads(min(ORACLE_THREADS, os.cpu_count() or 1))

    def __exit__(self, *a):
        torch.set_num_threads(self.n)


@pytest.fixture(scope="module")
def case():
    """(golden arrays, state dict, oracle weights incl. pe tables, stats tuple, seeded inputs) of fulldims.npz."""
    return fulldims_case()


@pytest.fixture(scope="module")
def case64(case):
    """The same weights and statistics in float64: the oracle run on them is the yardstick of tests/parity_tol.py::yardstick."""
    _, _, W, stats, _ = case
    return to64(W), tuple(t.double() for t in stats)


NAMES = ("x", "x2", "pred_xstart", "pred_xstart2")


def _step64(case64, sch, i, x, x2, cond):
    """One oracle step in float64 from the same (fp32-valued) inputs -> {state name: float64 tensor}."""
    W64, stats64 = case64
    with _Threads(), torch.no_grad():
        return dict(zip(NAMES, MX.mixer_ddim_step(W64, MX.MixerSpec(d_heads=8, m_heads=8), stats64, sch, 3.5, i, x.double(), x2.double(), cond.double(), {})))


@pytest.fixture(scope="module")
def samplers(case):
    """One handle per precision mode at the real sizes, B <= 2, T <= 300, with the fixture's statistics."""
    from mixermdm_amd.sampler import Sampler
    from mixermdm_amd.synthetic import FULL_DIMS
    g, sd, W, stats, _ = case
    made = {}
    for mode in MODES:
        s = Sampler(d_heads=8, m_heads=8, max_batch=2, max_frames=300, precision=mode, **FULL_DIMS)
        s.load_state_dict(sd)
        s.set_norm_stats(*[t.numpy() for t in stats])
        s.prepare()
        made[mode] = s
    yield made
    for s in made.values():
        s.close()


def _step(s, x, x2, i, names=("x", "x2", "pred_xstart", "pred_xstart2")):
    """One teacher-forced step of the begun call from chains (x, x2) at respaced index i -> cloned outputs."""
    _force(s, x, x2, i)
    s.run(1, use_graph=True)
    st = s.state()
    return {k: st[k].clone() for k in names}


def _denorm(stats, i):
    """(mean, std) per channel of the tensors that hold normalised poses: pred_xstart (chain 1: HumanML statistics per person, gaussian_diffusion.py:2052-2056)
    and pred_xstart2 (InterHuman statistics); None at i == 0, where both are the raw blend (quirk 6)."""
    if stats is None or i == 0:
        return None
    mean_h, std_h, mean_i, std_i = [torch.as_tensor(v).double().flatten() for v in stats]
    return {"pred_xstart": (mean_h.repeat(2), std_h.repeat(2)), "pred_xstart2": (mean_i.repeat(2), std_i.repeat(2))}


def _check_step(s, x, x2, i, refs, what, ref64, mode="fp32", stats=None):
    """refs: {state name: fp32 reference tensor} (the oracle's, or the reference's own captured output); ref64: the float64 oracle's
    outputs of the same step.  Element-wise parity with `refs` at the float64-derived tolerance, and the float64 yardstick
    (tests/parity_tol.py).  Returns (HIP outputs, worst out-of-tolerance fraction, ill-conditioned groups, yardstick entry)."""
    out = _step(s, x, x2, i)
    r64 = {k: ref64[k] for k in refs}
    # a turned joint (tests/parity_tol.py) is tolerated in the fp32_split mode only; the native-fp32 headline mode must show none
    worst, amplified, events = compare_step(out, refs, r64, what, hard_joints=HARD_JOINTS if mode == "fp32_split" else 0, denorm=_denorm(stats, i))
    return out, worst, amplified, yardstick(out, refs, r64, what, event=events > 0)


def _force(s, x, x2, i):
    """Teacher forcing: overwrite both chains of the begun call and continue from respaced step i."""
    st = s.state()
    st["x"].copy_(x.to(st["x"].device))
    st["x2"].copy_(x2.to(st["x2"].device))
    torch.cuda.synchronize()
    s.seek(i)


# ---------------------------------------------------------------------------------------------------
# (d) the reference itself at the real sizes
# ---------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def golden_f64(case, case64):
    """The float64 oracle's outputs of the fixture's steps: the yardstick the reference's own fp32 outputs (the expected VALUES) and the HIP
    outputs are both measured against."""
    g, _, W, stats, inp = case
    out = {}
    cb, xT, xb2 = inp["step"]
    s50 = OS.make_schedule("cosine", 1000, "ddim50")
    for i in (32, 0):
        out[f"ddim50:{i}"] = _step64(case64, s50, i, xT, xb2, cb)
    c300, x300 = inp["t300"]
    xa, xb = inp["late"]
    sch = OS.make_schedule("cosine", 1000, "ddim1000")
    out["t300:999"] = _step64(case64, sch, 999, x300, x300, c300)
    out["t300:3"] = _step64(case64, sch, 3, xa, xb, c300)
    return out


@pytest.mark.parametrize("mode", MODES)
def test_reference_golden_at_full_dims(case, samplers, golden_f64, mode):
    """Mixer.forward, ddim_sample at i=32 / i=0 (B=2, T=32) and two ddim1000 steps at T=300 (B=1): HIP == reference outputs captured at
    D=1024/512 with the seeded weights (tests/golden/make_golden.py::g_fulldims)."""
    g, _, _, _, inp = case
    s = samplers[mode]
    x1, x2, cond, tt = inp["fwd"]
    out = s.module_forward(2, x1, cond, tt, x2=x2)
    assert_close(out, torch.from_numpy(g["fwd"]), what=f"Mixer.forward [{mode}]", **STEP_TOL)
    cb, xT, xb2 = inp["step"]
    s.set_schedule("ddim50")
    s.begin(cb, xT)
    f64 = golden_f64
    for i in (32, 0):
        refs = {nm: torch.from_numpy(g[f"ddim50:i{i}:{key}"]) for nm, key in (("x", "sample"), ("x2", "sample2"), ("pred_xstart2", "pred_xstart2"))}
        _check_step(s, xT, xb2, i, refs, f"reference golden ddim50 i={i} B=2 T=32 [{mode}]", f64[f"ddim50:{i}"], mode)
    c300, x300 = inp["t300"]
    s.set_schedule("ddim1000")
    s.begin(c300, x300)
    _check_step(s, x300, x300, 999, {"x": torch.from_numpy(g["ddim1000:T300:i999:sample"]), "x2": torch.from_numpy(g["ddim1000:T300:i999:sample2"])},
                f"reference golden T=300 i=999 [{mode}]", f64["t300:999"], mode)
    xa, xb = inp["late"]
    _check_step(s, xa, xb, 3, {"x": torch.from_numpy(g["ddim1000:T300:i3:sample"]), "x2": torch.from_numpy(g["ddim1000:T300:i3:sample2"])},
                f"reference golden T=300 i=3 [{mode}]", f64["t300:3"], mode)


# ---------------------------------------------------------------------------------------------------
# (a) one full-dims DDIM step at T = 300, B = 2 against the oracle
# ---------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def oracle_t300_b2(case, case64):
    from mixermdm_amd.synthetic import synthetic_inputs
    _, _, W, stats, _ = case
    cond, xT = synthetic_inputs(2, 300, seed_cond=41, seed_x=42)
    x2 = rnd(43, 2, 300, 524)
    sch = OS.make_schedule("cosine", 1000, "ddim1000")
    with _Threads(), torch.no_grad():
        first = MX.mixer_ddim_step(W, MX.MixerSpec(d_heads=8, m_heads=8), stats, sch, 3.5, 999, xT, xT, cond)
        mid = MX.mixer_ddim_step(W, MX.MixerSpec(d_heads=8, m_heads=8), stats, sch, 3.5, 500, xT, x2, cond)
    return cond, xT, x2, (first, _step64(case64, sch, 999, xT, xT, cond)), (mid, _step64(case64, sch, 500, xT, x2, cond))


@pytest.mark.parametrize("mode", MODES)
def test_full_dims_T300_B2_step_vs_oracle(case, samplers, oracle_t300_b2, mode):
    cond, xT, x2, (first, f64a), (mid, f64b) = oracle_t300_b2
    stats = case[3]
    s = samplers[mode]
    s.set_schedule("ddim1000")
    s.begin(cond, xT)
    names = ("x", "x2", "pred_xstart", "pred_xstart2")
    _check_step(s, xT, xT, 999, dict(zip(names, first)), f"T=300 B=2 i=999 [{mode}]", f64a, mode, stats)
    _check_step(s, xT, x2, 500, dict(zip(names, mid)), f"T=300 B=2 i=500 [{mode}]", f64b, mode, stats)      # chains that differ, mid-schedule coefficients


# ---------------------------------------------------------------------------------------------------
# (b) ddim1000: first 20 and last 20 steps, teacher-forced, B = 1, T = 300
# ---------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def oracle_chains(case, case64, samplers):
    """Oracle trajectories: 20 free-running steps from x_T (i = 999..980), and 20 from the state the HIP fp32 sampler reaches at
    i = 19 after 980 steps of its own loop (i = 19..0, incl. the un-normalised i == 0 branch).  states[k] -> states[k+1] is one oracle
    step; every HIP mode is forced to states[k] before its step k, so one oracle pass serves both modes."""
    _, _, W, stats, inp = case
    cond, xT = inp["t300"]
    sch = OS.make_schedule("cosine", 1000, "ddim1000")
    spec = MX.MixerSpec(d_heads=8, m_heads=8)
    s = samplers["fp32"]
    s.set_schedule("ddim1000")
    s.begin(cond, xT)
    s.run(980, use_graph=True)
    st = s.state()
    late0 = (st["x"].cpu().clone(), st["x2"].cpu().clone())
    assert torch.isfinite(late0[0]).all() and torch.isfinite(late0[1]).all()
    chains = {}
    with _Threads(), torch.no_grad():
        for name, i0, (x, x2) in (("first", 999, (xT, xT)), ("last", 19, late0)):
            states = [(x, x2, None, None, None)]
            for k in range(20):
                nx, nx2, p1, p2 = MX.mixer_ddim_step(W, spec, stats, sch, 3.5, i0 - k, states[-1][0], states[-1][1], cond)
                f64 = _step64(case64, sch, i0 - k, states[-1][0], states[-1][1], cond)      # the same step from the same state, in float64
                states.append((nx, nx2, p1, p2, f64))
            chains[name] = (i0, states)
    return cond, xT, chains


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("which", ["first", "last"])
def test_ddim1000_teacher_forced_20_steps(case, samplers, oracle_chains, mode, which):
    cond, xT, chains = oracle_chains
    stats = case[3]
    i0, states = chains[which]
    s = samplers[mode]
    s.set_schedule("ddim1000")
    s.begin(cond, xT)
    worst, amplified, yards, events = 0.0, 0, [], 0
    for k in range(20):
        x, x2 = states[k][:2]
        rx, rx2, rp1, rp2, f64 = states[k + 1]
        out, w, a, y = _check_step(s, x, x2, i0 - k, {"x": rx, "x2": rx2, "pred_xstart": rp1, "pred_xstart2": rp2}, f"ddim1000 {which} i={i0 - k} [{mode}]", f64, mode, stats)
        worst = max(worst, w)
        amplified += int(a > 0)
        events += int(y["turned_joint_event"])
        yards.append(y)
    # native fp32 (the headline mode): none, ever (compare_step already refuses one there); fp32_split: at most one per 20-step sequence
    # (round 4 observed one with an intermediate kernel, at i = 15; the final kernels show none in either mode)
    assert events <= (1 if mode == "fp32_split" else 0), f"{events} steps of 20 with a turned joint [{mode}]"
    if which == "last":                       # quirk 6: the final step returns the raw (un-normalised) blend in both pred_xstart
        assert torch.equal(out["pred_xstart"], out["pred_xstart2"])
    med = yardstick_sequence(yards, f"ddim1000 {which} 20 teacher-forced steps [{mode}]")
    print(f"{mode} {which}: worst out-of-tolerance fraction over 20 steps {worst:.2e}; steps with an ill-conditioned group: {amplified}; "
          f"largest median HIP/CPU error ratio vs float64 {max(med.values()):.2f}")
    record(f"ddim1000 {which} 20 teacher-forced steps [{mode}]", kind="summary", worst_out_of_tol_fraction=worst, steps_with_an_ill_conditioned_group=amplified,
           steps_with_a_turned_joint=events, mode=mode)
    # the float64-derived tolerance must stay the exception, not the rule: observed 1 (last) / 0 (first) of 20 steps
    assert amplified <= 3


# ---------------------------------------------------------------------------------------------------
# (c) the GEMM instantiations of the B = 16 step (M = 19 200 rows), each against a float64 product
# ---------------------------------------------------------------------------------------------------
M_FULL = 19200        # 2 persons x 2B x T = 4 x 16 x 300 rows of a denoiser stack at BASELINE configs[2]
SHAPES = [(1024, 1024, "resid"), (1024, 2048, "resid"), (3072, 1024, "bias"), (2048, 1024, "gelu"), (2048, 1024, "bias"),
          (1024, 512, "gelu"), (512, 1024, "resid"), (512, 512, "resid"), (1536, 512, "bias")]


def _f64_ref(x, w, b, epi, r):
    y = F.linear(x.double(), w.double(), b.double())
    if epi == "gelu":
        y = F.gelu(y)
    elif epi == "resid":
        y = y + r.double()
    return y


@pytest.fixture(scope="module")
def gemm_operands():
    cache = {}

    def get(N, K, epi):
        if (N, K, epi) not in cache:
            x, w, b = rnd(301, M_FULL, K), rnd(302, N, K, scale=1 / math.sqrt(K)), rnd(303, N)
            r = rnd(304, M_FULL, N) if epi == "resid" else None
            with _Threads():
                ref = _f64_ref(x, w, b, epi, r)
            cache.clear()                     # one 19200 x 3072 float64 reference at a time is enough host memory
            cache[(N, K, epi)] = (x, w, b, r, ref)
        return cache[(N, K, epi)]
    return get


@pytest.mark.parametrize("N,K,epi", SHAPES)
def test_production_gemm_tiles_vs_float64(gemm_operands, N, K, epi):
    """mmdm_linear_f32 at M = 19 200 on the tiles the step uses; the launched instantiation is asserted, so a later change of the
    dispatch rules cannot silently move the production shapes onto an untested kernel."""
    from mixermdm_amd import ops
    from mixermdm_amd._lib import load_library
    x, w, b, r, ref = gemm_operands(N, K, epi)
    d = dev()
    rd = r.to(d) if r is not None else None
    got = ops.linear(x.to(d), w.to(d), b.to(d), epi, rd)
    kern = load_library().mmdm_last_gemm_kernel().decode()
    assert kern in PRODUCTION_F32, (N, K, epi, kern)
    assert_close(got, ref.float(), atol=2e-5 * math.sqrt(K / 1024), rtol=1e-5, what=f"linear 19200x{N}x{K} {epi} on {kern}")
    if epi == "resid":                        # in place, as the residual stream is updated
        ops.linear(x.to(d), w.to(d), b.to(d), epi, rd, out=rd)
        assert torch.equal(rd, got)


@pytest.mark.parametrize("N,K,epi", SHAPES)
def test_production_split_gemm_tiles_vs_float64(gemm_operands, N, K, epi):
    from mixermdm_amd import ops
    from mixermdm_amd._lib import load_library
    x, w, b, r, ref = gemm_operands(N, K, epi)
    d = dev()
    xs, ws = ops.split_f32(x.to(d)), ops.split_f32(w.to(d))
    got = ops.linear_split(xs, ws, b.to(d), epi, r.to(d) if r is not None else None)
    kern = load_library().mmdm_last_gemm_kernel().decode()
    assert all(k in PRODUCTION_SPLIT for k in kern.split("+")), (N, K, epi, kern)
    assert_close(got, ref.float(), atol=2e-5 * math.sqrt(K / 1024), rtol=1e-5, what=f"linear_split 19200x{N}x{K} {epi} on {kern}")
    if epi == "gelu":                         # two-plane output = the split of the fp32 output
        g3 = ops.linear_split(xs, ws, b.to(d), epi, split_out=True)
        assert torch.equal(g3, ops.split_f32(got))
    # the kernel the fp32-split sampler runs (weights in fragment order, W straight from global memory): same bits, and the instantiation is asserted
    wp = ops.split_pack_weight(ws)
    got_p = ops.linear_split(xs, wp, b.to(d), epi, r.to(d) if r is not None else None, packed=True)
    kern_p = load_library().mmdm_last_gemm_kernel().decode()
    assert all(k in PRODUCTION_SPLIT_PACKED for k in kern_p.split("+")), (N, K, epi, kern_p)
    assert torch.equal(got_p, got), f"packed fp32-split GEMM differs from the plane kernel: 19200x{N}x{K} {epi} on {kern_p}"
    if epi == "gelu":
        assert torch.equal(ops.linear_split(xs, wp, b.to(d), epi, split_out=True, packed=True), g3)


# instantiations the dispatch rules pick at M = 19 200 (update together with the rules in gemm_f32.hip / gemm_split.hip / gemm_bf16.hip)
PRODUCTION_F32 = {"gemm_pipe<22,22,16,5,vepi>", "gemm_pipe<22,22,16,5,scalar>", "gemm_pipe<22,21,16,4,vepi>", "gemm_pipe<22,21,16,4,scalar>"}
PRODUCTION_SPLIT = {"gemm_split<42,22>", "gemm_split<22,21>"}
PRODUCTION_SPLIT_PACKED = {"gemm_splitw<14,41>", "gemm_splitw<12,41>"}


def test_dispatch_puts_the_layer_gemms_on_the_large_tiles():
    """The four GEMMs that are 90 % of a step's FLOPs must not run on the small-shape fallbacks."""
    from mixermdm_amd import ops
    from mixermdm_amd._lib import load_library
    d = dev()
    lib = load_library()
    for N, K, epi, want in [(1024, 1024, "resid", "gemm_pipe<22,22,16,5"), (1024, 2048, "resid", "gemm_pipe<22,22,16,5"),
                            (3072, 1024, "bias", "gemm_pipe<22,22,16,5"), (2048, 1024, "gelu", "gemm_pipe<22,21,16,4")]:
        x, w = torch.zeros(M_FULL, K, device=d), torch.zeros(N, K, device=d)
        r = torch.zeros(M_FULL, N, device=d) if epi == "resid" else None
        ops.linear(x, w, None, epi, r)
        assert lib.mmdm_last_gemm_kernel().decode().startswith(want), (N, K, lib.mmdm_last_gemm_kernel())
