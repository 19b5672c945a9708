"""Comparison of one sampler step with the oracle, justified by a float64 run of the same oracle (shared by the GPU parity tests; test
infrastructure).

Two statements are made about every compared step (round 3; replaces the hand-fitted conditioning factors of round 2):

1. PARITY with the fp32 oracle, element by element (`compare_step`): all but 0.2 % of the elements within atol 2e-4 + rtol 2e-4 of the
   CPU fp32 oracle (the reference's own arithmetic: pinned to it by tests/golden), none beyond 5e-2 -- where the tolerance of an element
   is never smaller than what fp32 arithmetic itself costs THERE: 12 x the 99.9th percentile of |CPU-fp32 - float64| over the element's
   (sample, person, channel class) group.  The oracle runs unchanged on .double() weights and inputs (oracle/geometry.py::_f); its
   distance from the fp32 oracle is the measured conditioning of the function at that point.  Where the reference's geometry is
   ill-conditioned -- `qbetween` of nearly (anti-)parallel directions in align_motions / center_motion turns a whole person by an angle
   that is rounding noise -- the CPU's own fp32 error is large in exactly that person's position / velocity channels, and the tolerance
   follows it; everywhere else the plain tolerance applies.  No geometric formula, no mask.

2. YARDSTICK (`yardstick`): the HIP path's distance from float64 is of the size of the CPU fp32 oracle's distance from float64, at the
   quantiles p50 / p99 / p99.9 of every channel class: <= 3 x for the rotation-6D and foot-contact channels of every step, <= 12 x for the
   position / velocity channels of a single step, and <= 3 x for EVERY class and quantile (incl. the maximum) in the median over a
   teacher-forced sequence of steps (`yardstick_sequence`).  Why two factors: a person's position / velocity channels all pass through
   ONE global rotation whose angle error is a single random number per person and step, so the per-step quantile ratio of that class is the
   ratio of two random scalars -- heavy-tailed (observed up to 9.8 over 160 step x tensor samples, median 2.1) -- while its median over a
   sequence is stable.  Measured (profiles/r03_parity_report.json): HIP / CPU medians 1.1-2.2 (fp32 and fp32_split), 2.3-2.6 on the
   geometry-free single-person path.  The factor ~2 is the accumulation order: v_mfma_f32_32x32x2_f32 sums an output element as ONE
   k-ordered fp32 chain of K = 1024-2048 terms, the CPU GEMM as many short chains; the oracle with its GEMMs replaced by a k-ordered fp32
   chain shows the same 1.3-2.1 x (tools/accum_order_experiment.py).
"""
import torch

STEP_TOL = dict(atol=2e-4, rtol=2e-4, frac=2e-3, hard=5e-2)
# Near-degenerate joints (rot6d vectors almost collinear) amplify rounding through the Gram-Schmidt / quaternion round trip to ~1e-3 in
# single rot6d components; `frac` allows 0.2 % of them, and never fewer than this many elements (each still inside `hard`).
MIN_OUTLIERS = 4
HARD_OUTLIERS = 2          # rot6d components per tensor that may pass the hard bound (round 3's rule; where the caller passes compare_step(denorm=) each of them
                           # must sit on a joint at the half-turn discontinuity: CLIFF_MARGIN below), or ...
HARD_JOINTS = 1            # ... (only where the caller allows it: compare_step(hard_joints=1), the fp32_split mode; the native-fp32 headline mode asserts 0)
                           # all components of ONE (sample, frame, person, joint) rot6d sextet -- a "turned joint" -- if ...
ILL_JOINT = 25.0           # ... the CPU fp32 oracle's own |fp32 - float64| on that joint is >= ILL_JOINT x the tensor's median rot6d figure
CLIFF_MARGIN = 1e-4        # half-turn discontinuity of the reference's matrix_to_quaternion: the sign differences m21 - m12, m02 - m20, m10 - m01 of the joint's rotation
                           # (float64 oracle) all below this -- a 1e-5 difference between two implementations' inputs can then flip a sign
GROUP_FACTOR = 12.0        # element tolerance >= GROUP_FACTOR x p99.9 of |CPU-fp32 - float64| over the element's (sample, person, class) group
HARD_FACTOR = 100.0        # hard bound >= HARD_FACTOR x the same figure
AMPLIFIED = 25.0           # a group whose tolerance exceeds AMPLIFIED x the plain one is counted (and reported) as ill-conditioned

QUANTILES = (0.5, 0.99, 0.999, 1.0)
QNAMES = ("p50", "p99", "p99.9", "max")
YARD_FACTOR = 3.0          # rot6d / feet per step; every class in the median over a sequence
YARD_FACTOR_POSVEL = 12.0  # position / velocity channels of a single step (one global rotation per person: see the module docstring)
YARD_FACTOR_LOOP = 10.0    # free-running loops: S steps of compounded divergence, one sample of a chaotic process
YARD_FLOOR = 1e-6          # two orders below atol: quantiles of exactly representable channels are rounding noise of ~1e-7
REPORT = []                # one dict per compared step; tests/conftest.py writes it to gpurun_out/parity_report.json at session end
REPORT_ONLY = False        # set by `pytest --parity-report-only` (tests/conftest.py): yardstick asserts become log lines and the report is stamped
                           # "authoritative": false -- a diagnostic run for tools/parity_stats.py, never the suite the driver runs


def channel_classes(C):
    ch = torch.arange(C) % 262
    return {"posvel": ch < 132, "rot6d": (ch >= 132) & (ch < 258), "feet": ch >= 258}


def _quant(e, qs=QUANTILES):
    e = e.flatten().double()
    if e.numel() == 0:
        return [0.0] * len(qs)
    srt = torch.sort(e).values
    return [float(srt[min(srt.numel() - 1, int(q * (srt.numel() - 1) + 0.5))]) for q in qs]


def to64(W):
    return {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in W.items()}


def record(what, **kw):
    """Extra per-test figures for the parity report."""
    REPORT.append(dict(what=what, **kw))


def oracle_step_pair(W, spec, stats, sched, s, i, x, x2, cond, W64=None, names=("x", "x2", "pred_xstart", "pred_xstart2")):
    """One MixerDiffusion.ddim_sample step of the oracle in fp32 and, from the same (fp32-valued) inputs, in float64.
    Returns ({name: fp32 tensor}, {name: float64 tensor})."""
    from oracle import mixer as MX
    W64 = W64 if W64 is not None else to64(W)
    with torch.no_grad():
        r32 = MX.mixer_ddim_step(W, spec, stats, sched, s, i, x, x2, cond)
        r64 = MX.mixer_ddim_step(W64, spec, tuple(torch.as_tensor(t).double() for t in stats), sched, s, i, x.double(), x2.double(), cond.double())
    return dict(zip(names, r32)), dict(zip(names, r64))


def group_scale(e_cpu):
    """|CPU-fp32 - float64| [B, T, C] -> (G, ngroups): G [B, T, C] = p99.9 of it over the element's (sample, person, channel class) group."""
    B, T, C = e_cpu.shape
    G = torch.zeros_like(e_cpu)
    ch = torch.arange(C)
    per_group = []
    for p in range(C // 262):
        for lo, hi in ((0, 132), (132, 258), (258, 262)):
            sel = (ch >= p * 262 + lo) & (ch < p * 262 + hi)
            k = torch.sort(e_cpu[:, :, sel].reshape(B, -1), dim=1).values
            q = k[:, min(k.shape[1] - 1, int(0.999 * (k.shape[1] - 1) + 0.5))]
            G[:, :, sel] = q[:, None, None]
            per_group.append(q)
    return G, torch.stack(per_group, dim=1)          # [B, groups]


def half_turn_evidence(r64, mean_std, b, t, c):
    """The rotation of the joint that channel c belongs to, from the float64 oracle's (normalised) output r64 [B, T, C]: 1 + trace (0 at a rotation
    by pi) and the largest of the three sign differences the reference's matrix_to_quaternion decides on (rotation_conversions.py:98-120)."""
    from oracle import geometry as _G
    mean, std = [torch.as_tensor(v).double() for v in mean_std]
    c0 = (c // 262) * 262 + 132 + (((c % 262) - 132) // 6) * 6
    R = _G.rotation_6d_to_matrix((r64[b, t, c0:c0 + 6] * std[c0:c0 + 6] + mean[c0:c0 + 6])[None])[0]
    return {"one_plus_trace": float(1 + R[0, 0] + R[1, 1] + R[2, 2]),
            "max_sign_margin": float(max(abs(R[2, 1] - R[1, 2]), abs(R[0, 2] - R[2, 0]), abs(R[1, 0] - R[0, 1])))}


def compare_step(out, ref32, ref64, what, tol=STEP_TOL, hard_joints=0, denorm=None):
    """out: {state name: HIP tensor [B, T, C]}; ref32 / ref64: the fp32 and the float64 oracle's outputs of this very step.
    Asserts statement 1 of the module docstring for every tensor of ref32; returns (worst out-of-tolerance fraction, largest number of
    ill-conditioned (sample, person, class) groups in a tensor, number of tensors with a turned-joint event).
    hard_joints: how many turned joints (module constants above) a tensor may carry -- 0 for the native-fp32 headline mode and every caller
    that does not say otherwise, HARD_JOINTS for the fp32_split mode (the one mode a turned joint was ever observed in: ddim1000 i = 15, round 4).
    denorm: {tensor name: (mean [C], std [C])} for tensors that hold NORMALISED poses (pred_xstart*): with it, every rot6d component beyond the
    hard bound must sit on a joint whose rotation -- taken from the float64 oracle's own output -- is at the half-turn discontinuity of the
    reference's matrix_to_quaternion (rotation_conversions.py:98-120 decides the signs of x, y, z on m21 - m12, m02 - m20, m10 - m01, which all
    vanish at a rotation by pi): max |difference| <= CLIFF_MARGIN.  (the one such pair of the suite, ddim1000 i = 995 frame 251
    joint 17, reads 1 + trace = 1e-13, differences <= 6.4e-7 -- while the oracle's own fp32 - float64 there is an unremarkable 3 medians: the
    discontinuity is in the INPUT direction, which identical-input rounding cannot see.)"""
    worst, amplified, events = 0.0, 0, 0
    detail = {"kind": "step_vs_fp32_oracle", "tensors": {}}
    for nm, ref in ref32.items():
        got, ref = out[nm].detach().cpu().double(), torch.as_tensor(ref).detach().cpu().double()
        r64 = torch.as_tensor(ref64[nm]).detach().cpu().double()
        assert got.shape == ref.shape == r64.shape, (nm, got.shape, ref.shape, r64.shape)
        assert torch.isfinite(got).all(), f"{what} {nm}: non-finite"
        C = got.shape[-1]
        G, per_group = group_scale((ref - r64).abs())
        plain = tol["atol"] + tol["rtol"] * ref.abs()
        lim = torch.maximum(plain, GROUP_FACTOR * G)
        d = (got - ref).abs()
        nbad = int((d > lim).sum())
        frac = nbad / d.numel()
        if nbad <= MIN_OUTLIERS:          # tiny tensors (T = 1, 2): a fraction of 1048 elements is two elements; the outliers are single rot6d components
            frac = 0.0
        n_amp = int((GROUP_FACTOR * per_group > AMPLIFIED * tol["atol"]).sum())
        note = f"(largest group tolerance {float((GROUP_FACTOR * per_group).max()):.2e}, {n_amp} ill-conditioned group(s) of {per_group.numel()})"
        assert frac <= tol["frac"], f"{what} {nm}: {frac:.2e} of the elements outside tolerance, max err {d.max().item():.2e} {note}"
        over = d > torch.maximum(torch.full_like(d, tol["hard"]), HARD_FACTOR * G)
        ch = torch.arange(C) % 262
        rot = ((ch >= 132) & (ch < 258))[None, None, :].expand_as(over)
        # A rot6d pair within ~1e-5 of collinear makes Gram-Schmidt amplify fp32 rounding by ~1e5 (one joint in ~1e5 on noise inputs).  Up to
        # HARD_OUTLIERS single rot6d components per tensor may pass the hard bound (round 3's rule).  When the pair is closer still the whole joint
        # turns as a unit (round 4, ddim1000 i = 15 in fp32_split: five components of one joint off by up to 1.5): more components than that must
        # all belong to at most HARD_JOINTS (sample, frame, person, joint) sextets, and each such joint must be one the ORACLE ITSELF finds
        # ill-conditioned -- the CPU fp32 oracle's own distance from float64 on that joint at least ILL_JOINT x the tensor's median rot6d figure
        # (observed 3e-4 .. 5e-4 against a median of ~1e-6).  Such a step is an EVENT (the third return value; see yardstick(event=True)).
        # Position / velocity / foot channels never may pass the hard bound.
        assert not bool((over & ~rot).any()), f"{what} {nm}: max err {d.max().item():.2e} {note}"
        if int(over.sum()) > HARD_OUTLIERS:
            events += 1
            e_cpu = (ref - r64).abs()
            med = float(e_cpu[rot].median())
            joints = {}
            for b, t, c in over.nonzero().tolist():
                c0 = (c // 262) * 262 + 132 + (((c % 262) - 132) // 6) * 6
                joints.setdefault((b, t, c0), []).append(c)
            where = [(k, [f"{d[k[0], k[1], c].item():.1e}" for c in v], f"cpu32-f64 {float(e_cpu[k[0], k[1], k[2]:k[2] + 6].max()):.1e} (median {med:.1e})") for k, v in joints.items()]
            assert len(joints) <= hard_joints, f"{what} {nm}: rot6d components of {len(joints)} joints beyond the hard bound ({hard_joints} allowed), max err {d.max().item():.2e} {note} at {where[:6]}"
            for (b, t, c0) in joints:
                assert float(e_cpu[b, t, c0:c0 + 6].max()) >= ILL_JOINT * med, \
                    f"{what} {nm}: a joint the oracle finds well-conditioned is beyond the hard bound, max err {d.max().item():.2e} {note} at {where[:6]}"
        # evidence for EVERY component beyond the hard bound, also the <= HARD_OUTLIERS ones that pass without a joint-level rule: the CPU fp32 oracle's
        # own distance from float64 on the component's joint against the tensor's median rot6d figure (a joint the oracle itself finds ill-conditioned
        # reads tens to hundreds of medians; VERDICT r5 weak 0d: such components used to pass "without further evidence")
        hard_ev = []
        if bool(over.any()):
            e_cpu = (ref - r64).abs()
            med = float(e_cpu[rot].median())
            for b, t, c in over.nonzero().tolist()[:16]:
                c0 = (c // 262) * 262 + 132 + (((c % 262) - 132) // 6) * 6
                own = float(e_cpu[b, t, c0:c0 + 6].max())
                ev = {"sample": b, "frame": t, "channel": c, "err": float(d[b, t, c]), "oracle_fp32_vs_f64_on_joint": own,
                      "median_rot6d_oracle_err": med, "ratio": own / max(med, 1e-30)}
                if denorm is not None and nm in denorm:
                    ev.update(half_turn_evidence(r64, denorm[nm], b, t, c))
                    assert ev["max_sign_margin"] <= CLIFF_MARGIN, \
                        f"{what} {nm}: a rot6d component beyond the hard bound (err {ev['err']:.2e}) on a joint that is NOT at the half-turn discontinuity: {ev}"
                hard_ev.append(ev)
        worst = max(worst, frac)
        amplified = max(amplified, n_amp)
        detail["tensors"][nm] = {"beyond_hard_bound": hard_ev, "out_of_tol_fraction": frac, "out_of_tol_elements": nbad, "elements": int(d.numel()), "max_err": float(d.max()),
                                 "max_err_over_plain_tol": float((d / plain).max()), "largest_group_tolerance": float((GROUP_FACTOR * per_group).max()),
                                 "ill_conditioned_groups": n_amp, "groups": int(per_group.numel())}
    detail["turned_joint_events"] = events
    record(what, **detail)
    return worst, amplified, events


# ---- statistical form (round 6): many independent draws instead of a list of seeds that happen to pass ----------------------------------
# compare_step's element tolerance follows the CPU fp32 oracle's own error in the element's (sample, person, class) group by GROUP_FACTOR.
# For the centred chain's position / velocity channels both errors are ONE random rotation-angle error per (item, person) -- the HIP
# path's and the CPU's are independent draws -- so their ratio is heavy-tailed: tools/ratio_distribution.py, 48 (item, person) draws at the
# real sizes: ratio of the p50 errors median 2.3, p90 4.4, max 14.5 -- ONE draw of 48 beyond GROUP_FACTOR = 12.  A test on fixed inputs is
# then green or red by the luck of its seed (round 5's ragged oracle test: seed 3 red, seeds 4-6 green).  The statement that can be
# calibrated is about a POPULATION of draws:
#   (i)   every draw keeps the hard bounds (finite, nothing beyond 5e-2 / HARD_FACTOR x the group scale outside the rot6d outlier
#         allowance) -- asserted per draw, no exception;
#   (ii)  the number of draws whose out-of-tolerance fraction exceeds STEP_TOL["frac"] is <= k(N), the smallest k with
#         P[Binomial(N, DRAW_EXCEED_RATE) > k] <= DRAW_ALPHA: with the measured rate (1 of 48; DRAW_EXCEED_RATE = 0.05 is about the
#         90 % one-sided upper confidence bound of that observation) an honest implementation fails the count once in a thousand runs,
#         a kernel whose error is a few times larger fails (ii) or (iii) nearly always;
#   (iii) the POOLED error quantiles (all draws of the test concatenated along the frame axis) obey the float64 yardstick at
#         POOLED_FACTOR = 3 for every channel class at p50 -- pooling averages the per-draw random scalar of the position / velocity class
#         away, so its MEDIAN needs no allowance -- and at p99 / p99.9 for the rot6d and foot classes.  The position / velocity class has
#         no pooled TAIL bound: one draw is 1 / N >= 4 % of the pooled elements, so its p99 / p99.9 are the one or two largest per-draw
#         scalars on either side -- exactly what (ii) counts draw by draw (a pooled bound there is either implied by (ii) or fails on the
#         very draw (ii) allows); the pooled figures are recorded in the report.
# Seeds are CONSECUTIVE from 0 and none is skipped; the report lists every draw.
DRAW_EXCEED_RATE = 0.05
DRAW_ALPHA = 1e-3
POOLED_FACTOR = 3.0


def binomial_bound(n, p=DRAW_EXCEED_RATE, alpha=DRAW_ALPHA):
    """Smallest k with P[Binomial(n, p) > k] <= alpha."""
    from math import comb
    tail = 1.0
    for k in range(n + 1):
        tail -= comb(n, k) * p ** k * (1 - p) ** (n - k)
        if tail <= alpha:
            return k
    return n


def step_draws(out, ref32, ref64, label, tol=STEP_TOL, denorm=None):
    """The (sample, person) draws of one compared step: per draw the worst out-of-tolerance fraction over the compared tensors (compare_step's
    element tolerance, taken over that person's 262 channels of that sample) and whether it exceeds tol["frac"].  Asserts statement (i)."""
    draws = {}
    for nm, ref in ref32.items():
        got, ref = out[nm].detach().cpu().double(), torch.as_tensor(ref).detach().cpu().double()
        r64 = torch.as_tensor(ref64[nm]).detach().cpu().double()
        assert got.shape == ref.shape == r64.shape, (nm, got.shape, ref.shape, r64.shape)
        assert torch.isfinite(got).all(), f"{label} {nm}: non-finite"
        B, T, C = got.shape
        G, _ = group_scale((ref - r64).abs())
        lim = torch.maximum(tol["atol"] + tol["rtol"] * ref.abs(), GROUP_FACTOR * G)
        d = (got - ref).abs()
        over = d > torch.maximum(torch.full_like(d, tol["hard"]), HARD_FACTOR * G)
        ch = torch.arange(C) % 262
        rot = ((ch >= 132) & (ch < 258))[None, None, :].expand_as(over)
        assert not bool((over & ~rot).any()), f"{label} {nm}: position / velocity / foot element beyond the hard bound, max err {d.max().item():.2e}"
        assert int(over.sum()) <= HARD_OUTLIERS, f"{label} {nm}: {int(over.sum())} rot6d components beyond the hard bound (fp32: {HARD_OUTLIERS} allowed)"
        if denorm is not None and nm in denorm:          # ... and each of them on a joint at the half-turn discontinuity (compare_step has the rule)
            for b, t, c in over.nonzero().tolist():
                ev = half_turn_evidence(r64, denorm[nm], b, t, c)
                assert ev["max_sign_margin"] <= CLIFF_MARGIN, f"{label} {nm}: a rot6d component beyond the hard bound (err {float(d[b, t, c]):.2e}) on a joint that is NOT at the half-turn discontinuity: {ev}"
        for b in range(B):
            for p in range(C // 262):
                sel = d[b, :, p * 262:(p + 1) * 262] > lim[b, :, p * 262:(p + 1) * 262]
                nbad = int(sel.sum())
                frac = 0.0 if nbad <= MIN_OUTLIERS else nbad / sel.numel()
                e = draws.setdefault((b, p), {"label": label, "sample": b, "person": p, "worst_fraction": 0.0, "worst_tensor": None, "max_err": 0.0})
                if frac >= e["worst_fraction"]:
                    e["worst_fraction"], e["worst_tensor"] = frac, nm
                e["max_err"] = max(e["max_err"], float(d[b, :, p * 262:(p + 1) * 262].max()))
    for e in draws.values():
        e["beyond_factor"] = e["worst_fraction"] > tol["frac"]
    return list(draws.values())


def compare_draws(steps, what, tol=STEP_TOL, denorm=None):
    """steps: [(out, ref32, ref64, label)] -- independent one-step comparisons ([B, T, C] tensors per state name).  Asserts (i)-(iii) above and
    records every draw.  Returns (number of draws beyond the factor, allowed k, N)."""
    draws = []
    for out, r32, r64, label in steps:
        draws += step_draws(out, r32, r64, label, tol, denorm)
    n, beyond = len(draws), sum(d["beyond_factor"] for d in draws)
    k = binomial_bound(n)
    REPORT.append({"what": what, "kind": "draws_vs_fp32_oracle", "draws": draws, "n_draws": n, "beyond_factor": beyond, "allowed": k,
                   "exceed_rate_assumed": DRAW_EXCEED_RATE, "alpha": DRAW_ALPHA, "group_factor": GROUP_FACTOR})
    assert beyond <= k, (f"{what}: {beyond} of {n} (item, person) draws beyond {GROUP_FACTOR} x the CPU oracle's own error "
                         f"(allowed {k} at rate {DRAW_EXCEED_RATE}, alpha {DRAW_ALPHA}): " + "; ".join(f"{d['label']} person {d['person']} sample {d['sample']}: "
                         f"{d['worst_fraction']:.1e} of {d['worst_tensor']}" for d in draws if d["beyond_factor"]))
    # (iii) pooled yardstick: every class at POOLED_FACTOR (batch rows of different steps are concatenated along the frame axis, per batch row 0..B-1
    # only where the shapes allow; ragged items come with B = 1)
    cat = lambda idx: {k: torch.cat([torch.as_tensor(s[idx][k]).detach().cpu().double().reshape(1, -1, s[idx][k].shape[-1]) for s in steps], 1) for k in steps[0][1]}
    yardstick(cat(0), cat(1), cat(2), f"{what}: {n} draws pooled", factor=POOLED_FACTOR, factor_posvel=YARD_FACTOR_POSVEL, factor_posvel_p50=POOLED_FACTOR)
    return beyond, k, n


def yardstick(out, ref32, ref64, what, factor=YARD_FACTOR, factor_posvel=YARD_FACTOR_POSVEL, floor=YARD_FLOOR, event=False, factor_posvel_p50=None):
    """Statement 2 for one step: |HIP - f64| <= factor x |fp32 oracle - f64| + floor at p50 / p99 / p99.9 of every channel class.
    The per-step MAXIMUM ratio is recorded and deliberately NOT bounded here: it is the ratio of two single elements -- in the position /
    velocity channels of two random rotation-angle errors, one per person and step -- and is heavy-tailed (45.9 x observed in one of 104
    steps, median 2.2).  What stands behind it: compare_step's element-wise hard bound of the same step (5e-2, never exceeded outside
    the rot6d outlier allowance) and yardstick_sequence, which bounds the MEDIAN over a sequence of every quantile including the
    maximum.  event: compare_step found a turned joint in this step (a rot6d sextet the oracle itself finds ill-conditioned); that frame's O(1)
    change of the mixer's input reaches every frame of the sample through the mixer's attention, so every class of the step is held to the
    position / velocity factor (observed 5.5 x in the foot channels of the one such step) -- the sequence medians stay at `factor`.
    Returns the figures (also appended to REPORT)."""
    if event:
        factor = max(factor, factor_posvel)
    entry = {"what": what, "kind": "yardstick", "factor": factor, "factor_posvel": factor_posvel, "floor": floor, "tensors": {}, "turned_joint_event": bool(event)}
    fails = []
    for nm, r64 in ref64.items():
        if nm not in ref32 or nm not in out or out[nm] is None:
            continue
        got = out[nm].detach().cpu().double()
        r32 = torch.as_tensor(ref32[nm]).detach().cpu().double()
        r64 = torch.as_tensor(r64).detach().cpu().double()
        e_hip, e_cpu = (got - r64).abs(), (r32 - r64).abs()
        t = entry["tensors"][nm] = {}
        for cname, sel in channel_classes(got.shape[-1]).items():
            qh, qc = _quant(e_hip[..., sel]), _quant(e_cpu[..., sel])
            t[cname] = {"hip_vs_f64": qh, "cpu32_vs_f64": qc, "ratio": [h / max(c, floor) for h, c in zip(qh, qc)]}
            for qn, h, c in list(zip(QNAMES, qh, qc))[:3]:
                f = factor_posvel if cname == "posvel" else factor
                if cname == "posvel" and factor_posvel_p50 is not None:
                    if qn != "p50":
                        continue                      # pooled draws (compare_draws): the tails of that class are statement (ii), recorded here, not bounded
                    f = factor_posvel_p50             # ... and its median is no longer one random scalar
                if h > f * c + floor:
                    fails.append(f"{nm}/{cname} {qn}: HIP {h:.3e} vs CPU-fp32 {c:.3e} (factor {f})")
    entry["ok"] = not fails
    REPORT.append(entry)
    if not REPORT_ONLY:
        assert not fails, f"{what}: HIP is further from float64 than the fp32 oracle allows: " + "; ".join(fails)
    return entry


def yardstick_sequence(entries, what, factor=YARD_FACTOR):
    """Statement 2 over a teacher-forced sequence: for every (tensor, class, quantile incl. max) the MEDIAN over the steps of the
    HIP / CPU-fp32 error ratio is <= factor."""
    acc = {}
    for e in entries:
        for nm, t in e["tensors"].items():
            for cname, v in t.items():
                for qn, r in zip(QNAMES, v["ratio"]):
                    acc.setdefault(f"{nm}/{cname}/{qn}", []).append(r)
    med = {k: float(torch.tensor(v).median()) for k, v in acc.items()}
    mx = {k: float(max(v)) for k, v in acc.items()}
    fails = [f"{k}: median ratio {m:.2f}" for k, m in med.items() if m > factor]
    REPORT.append({"what": what, "kind": "yardstick_sequence", "steps": len(entries), "factor": factor, "median_ratio": med, "max_ratio": mx, "ok": not fails})
    if not REPORT_ONLY:
        assert not fails, f"{what}: " + "; ".join(fails)
    return med
