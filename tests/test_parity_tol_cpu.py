"""CPU: the float64-derived step comparison (tests/parity_tol.py) is neither vacuous nor brittle -- checked on the oracle itself at small sizes:
an fp32 evaluation with rounding-level noise passes, a real alignment error (one person turned by half a degree, a wrong foot-contact
channel, a mis-scaled rotation block) fails."""
import math
import pytest
import torch

from oracle import mixer as MX, schedule as OS
from oracle.layers import pe_table
from mixermdm_amd.synthetic import synthetic_state_dict, synthetic_inputs, synthetic_stats
from parity_tol import compare_step, yardstick, yardstick_sequence, oracle_step_pair

DIMS = dict(d_latent=64, d_ff=128, d_layers=2, m_latent=32, m_ff=64, m_layers=2)


@pytest.fixture(scope="module")
def pair():
    sd = synthetic_state_dict(seed=0, std=0.05, bias_std=0.02, **DIMS)
    W = dict(sd)
    W["sequence_pos_encoder.pe"], W["denoiser1.sequence_pos_encoder.pe"], W["denoiser2.sequence_pos_encoder.pe"] = pe_table(32), pe_table(64), pe_table(64)
    st = synthetic_stats()
    stats = tuple(st[k] for k in ("mean_hml", "std_hml", "mean_ih", "std_ih"))
    cond, xT = synthetic_inputs(2, 24)
    x2 = torch.randn(2, 24, 524, generator=torch.Generator().manual_seed(9))
    sch = OS.make_schedule("cosine", 1000, "ddim50")
    r32, r64 = oracle_step_pair(W, MX.MixerSpec(d_heads=4, m_heads=4), stats, sch, 3.5, 30, xT, x2, cond)
    return r32, r64, stats


def _noisy(r32, rel):
    g = torch.Generator().manual_seed(1)
    return {k: v * (1 + rel * torch.randn(v.shape, generator=g)) for k, v in r32.items()}


def test_rounding_level_noise_passes(pair):
    r32, r64, _ = pair
    out = _noisy(r32, 2e-7)
    worst, amp, ev = compare_step(out, r32, r64, "noise")
    assert worst <= 2e-3
    e = yardstick(out, r32, r64, "noise")
    assert e["ok"]
    assert max(yardstick_sequence([e, e, e], "noise x3").values()) <= 3.0


def test_a_person_turned_by_half_a_degree_fails(pair):
    """pred_xstart is in normalised HumanML3D space: de-normalise person 1, turn positions about Y by 0.5 degrees, normalise again."""
    r32, r64, stats = pair
    mean, std = stats[0], stats[1]
    out = {k: v.clone() for k, v in r32.items()}
    p = out["pred_xstart"][..., :66] * std[:66] + mean[:66]
    j = p.reshape(*p.shape[:-1], 22, 3)
    a = math.radians(0.5)
    x, z = j[..., 0].clone(), j[..., 2].clone()
    j[..., 0], j[..., 2] = math.cos(a) * x + math.sin(a) * z, -math.sin(a) * x + math.cos(a) * z
    out["pred_xstart"][..., :66] = (j.reshape(p.shape) - mean[:66]) / std[:66]
    with pytest.raises(AssertionError, match="pred_xstart"):
        compare_step(out, r32, r64, "turned person")


@pytest.mark.parametrize("lo,hi,what", [(258, 262, "feet"), (132, 258, "rot6d"), (262 + 66, 262 + 132, "velocities of person 2")])
def test_a_wrong_channel_block_fails(pair, lo, hi, what):
    r32, r64, _ = pair
    out = {k: v.clone() for k, v in r32.items()}
    out["x2"][..., lo:hi] += 1e-2
    with pytest.raises(AssertionError, match="x2"):
        compare_step(out, r32, r64, what)


def test_one_turned_joint_passes_only_where_the_oracle_itself_is_ill_conditioned(pair):
    """Hard-bound rule of compare_step: a whole rot6d sextet (one joint of one frame) far off is accepted only if the CPU fp32 oracle's own distance
    from float64 on that joint is >= ILL_JOINT x the tensor's median -- the signature of a near-collinear rot6d pair; the same deviation on a joint
    the oracle finds well-conditioned fails, and so do two such joints."""
    r32, r64, _ = pair
    b, t, c0 = 1, 7, 262 + 132 + 6 * 20                       # person 2, joint 20
    out = {k: v.clone() for k, v in r32.items()}
    out["pred_xstart"][b, t, c0:c0 + 5] += torch.tensor([0.1, 1.5, 0.3, 0.8, 0.2])
    with pytest.raises(AssertionError, match="well-conditioned"):
        compare_step(out, r32, r64, "turned joint, well-conditioned", hard_joints=1)
    ill32 = {k: v.clone() for k, v in r32.items()}
    ill32["pred_xstart"][b, t, c0:c0 + 6] += 4e-4              # the fp32 oracle itself is 4e-4 from float64 on this joint
    out2 = {k: v.clone() for k, v in ill32.items()}
    out2["pred_xstart"][b, t, c0:c0 + 5] += torch.tensor([0.1, 1.5, 0.3, 0.8, 0.2])
    compare_step(out2, ill32, r64, "turned joint, ill-conditioned", hard_joints=1)
    with pytest.raises(AssertionError, match="0 allowed"):       # the native-fp32 mode (and every caller that does not ask) allows none
        compare_step(out2, ill32, r64, "turned joint, ill-conditioned, strict")
    ill32["pred_xstart"][0, 3, c0:c0 + 6] += 4e-4
    out3 = {k: v.clone() for k, v in ill32.items()}
    out3["pred_xstart"][b, t, c0:c0 + 5] += 1.0
    out3["pred_xstart"][0, 3, c0:c0 + 5] += 1.0
    with pytest.raises(AssertionError, match="2 joints"):
        compare_step(out3, ill32, r64, "two turned joints", hard_joints=1)
    out4 = {k: v.clone() for k, v in r32.items()}                 # round 3's allowance: two single components, no evidence asked, no event
    out4["pred_xstart"][b, t, c0:c0 + 2] += 0.2
    assert compare_step(out4, r32, r64, "two components")[2] == 0
    assert compare_step(out2, {k: v.clone() for k, v in ill32.items()}, r64, "event", hard_joints=1)[2] == 1


def test_statistical_form_is_neither_brittle_nor_vacuous(pair):
    """compare_draws (round 6): rounding-level noise on every draw passes; ONE draw beyond the group factor (the measured 1-in-50 event) passes the
    count and is listed; the same deviation on MORE draws than the binomial bound allows fails the count; an error a few times the oracle's
    own on every draw (no single draw spectacular) fails the pooled yardstick; one element beyond the hard bound fails whatever the count says."""
    from parity_tol import compare_draws, binomial_bound, REPORT
    r32, r64, _ = pair
    assert [binomial_bound(n) for n in (12, 24, 48)] == [4, 5, 8]
    steps = [(_noisy(r32, 2e-7), r32, r64, f"step {i}") for i in range(6)]          # 6 steps x 2 samples x 2 persons = 24 draws
    beyond, k, n = compare_draws(steps, "noise x 24 draws")
    assert (beyond, k, n) == (0, 5, 24)
    assert REPORT[-2]["kind"] == "draws_vs_fp32_oracle" and len(REPORT[-2]["draws"]) == 24

    # the measured event: the fp32 oracle itself is ill-conditioned for ONE (sample, person) -- 4e-5 from float64 in its position / velocity
    # channels -- and the other implementation lands 14 x as far on the other side
    def ill(ref, out, b, p, steps_idx=None):
        ref, out = {k: v.clone() for k, v in ref.items()}, {k: v.clone() for k, v in out.items()}
        g2 = torch.Generator().manual_seed(100 + 2 * b + p)
        sgn = torch.sign(torch.randn(ref["pred_xstart"][b, :, :132].shape, generator=g2))
        ref["pred_xstart"][b, :, p * 262:p * 262 + 132] += 4e-5 * sgn
        out["pred_xstart"][b, :, p * 262:p * 262 + 132] -= 14 * 4e-5 * sgn
        return out, ref
    one = list(steps)
    o, r = ill(r32, steps[2][0], 1, 0)
    one[2] = (o, r, r64, "step 2")
    beyond, k, n = compare_draws(one, "one heavy-tail draw")
    assert beyond == 1
    many = []
    for (o, a, c, lab) in steps[:3]:
        o, a = ill(a, o, 0, 0)
        o, a = ill(a, o, 1, 1)
        many.append((o, a, c, lab))
    with pytest.raises(AssertionError, match="6 of 24"):
        compare_draws(many + steps[3:], "six draws beyond the factor")
    g = torch.Generator().manual_seed(3)
    e = {k: (r32[k].double() - r64[k]).abs() for k in r32}
    five = [({k: r32[k] + 5 * e[k].float() * torch.sign(torch.randn(r32[k].shape, generator=g)) for k in r32}, r32, r64, f"step {i}") for i in range(6)]
    with pytest.raises(AssertionError, match="further from float64"):
        compare_draws(five, "5 x the oracle's own error everywhere")
    hard = list(steps)
    bad = {k: v.clone() for k, v in steps[0][0].items()}
    bad["x"][0, 3, 10] += 0.2
    hard[0] = (bad, r32, r64, "step 0")
    with pytest.raises(AssertionError, match="hard bound"):
        compare_draws(hard, "a position element beyond the hard bound")


def test_a_component_beyond_the_hard_bound_needs_a_joint_at_the_half_turn_discontinuity(pair):
    """compare_step(denorm=): up to HARD_OUTLIERS rot6d components may pass the hard bound, but only on a joint whose rotation (float64 oracle) is at the
    half-turn discontinuity of the reference's matrix_to_quaternion (its three sign decisions vanish there).  A joint planted AT a rotation by pi
    passes with two wild components; the same two components on an ordinary joint fail once the evidence is asked for (and still pass without)."""
    from parity_tol import CLIFF_MARGIN
    r32, r64, stats = pair
    mean_h, std_h = stats[0].double(), stats[1].double()
    den = {"pred_xstart": (mean_h.repeat(2), std_h.repeat(2))}
    b, t, c0 = 0, 5, 132 + 6 * 7
    # a rotation by pi about a unit axis u: R = 2 u u^T - I (symmetric: all three sign differences are exactly 0); rot6d = first two rows, interleaved
    u = torch.tensor([0.6, 0.0, 0.8], dtype=torch.float64)
    R = 2 * torch.outer(u, u) - torch.eye(3, dtype=torch.float64)
    d6 = R[:2].reshape(6)[[0, 3, 1, 4, 2, 5]]
    planted = (d6 - mean_h[c0:c0 + 6]) / std_h[c0:c0 + 6]
    r32p, r64p = {k: v.clone() for k, v in r32.items()}, {k: v.clone() for k, v in r64.items()}
    r32p["pred_xstart"][b, t, c0:c0 + 6] = planted.to(r32p["pred_xstart"].dtype)
    r64p["pred_xstart"][b, t, c0:c0 + 6] = planted.to(r64p["pred_xstart"].dtype)
    out = {k: v.clone() for k, v in r32p.items()}
    out["pred_xstart"][b, t, c0 + 1] += 0.4
    out["pred_xstart"][b, t, c0 + 2] -= 0.3
    assert compare_step(out, r32p, r64p, "two components at the half turn", denorm=den)[2] == 0
    # the same excursion on an ordinary joint: allowed without the evidence (round 3's rule), refused with it
    out2 = {k: v.clone() for k, v in r32.items()}
    out2["pred_xstart"][b, t, c0 + 1] += 0.4
    out2["pred_xstart"][b, t, c0 + 2] -= 0.3
    assert compare_step(out2, r32, r64, "two components, no evidence asked")[2] == 0
    with pytest.raises(AssertionError, match="NOT at the half-turn"):
        compare_step(out2, r32, r64, "two components on an ordinary joint", denorm=den)
    assert CLIFF_MARGIN <= 1e-3
