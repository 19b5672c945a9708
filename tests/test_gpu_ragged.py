"""GPU: the reference's real call shapes, redesigned (SURVEY 8f-2 "variable T bucketing, many small batches").

The reference samples one item at a time (src/scripts/infer/mixermdm.py:184-188 ten times B = 1; src/evaluation/datasets.py:100-116 per item with
its own length).  Two mechanisms replace that loop here, and both must leave every motion's BITS unchanged:
  (a) several sampler handles over ONE weight set (mmdm_create_shared), each on its own stream -- items in flight side by side;
  (b) RAGGED batches (mmdm_begin_ragged): items of different lengths in one batch, per-sequence lengths as device data.
Parity with the oracle is inherited through bit-identity with the stand-alone calls (tested against the oracle and the reference goldens elsewhere),
and checked directly for one ragged step."""
import ctypes as C
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import mixer as MX            # noqa: E402  (checker only)
from oracle import schedule as OS         # noqa: E402
from oracle.layers import pe_table        # noqa: E402
from parity_tol import compare_draws, oracle_step_pair, to64      # noqa: E402

# head size 64 in every stack (the ragged attention instantiations cover 64 and 128)
DIMS = dict(d_latent=128, d_ff=256, d_layers=2, m_latent=128, m_ff=256, m_layers=2)
LENS = (40, 17, 64, 1, 33)


def small(max_batch=8, max_frames=64, mode=4, precision="fp32", single_only=False):
    from mixermdm_amd.sampler import Sampler
    from mixermdm_amd.synthetic import synthetic_state_dict, synthetic_stats
    sd = synthetic_state_dict(seed=7, std=0.05, bias_std=0.02, mixing_mode=mode, **DIMS)
    if single_only:
        sd = {k: v for k, v in sd.items() if k.startswith("denoiser1.")}
    st = synthetic_stats()
    s = Sampler(d_heads=2, m_heads=2, max_batch=max_batch, max_frames=max_frames, mixing_mode=mode, precision=precision, single_only=single_only, **DIMS)
    s.load_state_dict(sd)
    if not single_only:
        s.set_norm_stats(st["mean_hml"], st["std_hml"], st["mean_ih"], st["std_ih"])
    s.prepare()
    return s


def inputs(lens, width=524, cw=8 * 768, seed=0):
    g = torch.Generator().manual_seed(seed)
    cond = torch.randn(len(lens), cw, generator=g)
    xs = [torch.randn(t, width, generator=g) for t in lens]
    return cond, xs


# ---------------------------------------------------------------------------------------------------
# kernel level: ragged attention == the uniform kernel on every sequence alone
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dh", [64, 128])
@pytest.mark.parametrize("shift", [0, 3])
def test_ragged_attention_is_the_uniform_kernel_per_sequence(dh, shift):
    from mixermdm_amd._lib import load_library, check
    lib = load_library()
    H, D = 2, 2 * dh
    lens = [70, 5, 64, 129, 16, 300]          # shift 3: sequence s attends to sequence s + 3 of the SAME length pattern (a person pair)
    if shift:
        lens = lens[:3] + lens[:3]
    nseq, total = len(lens), sum(lens)
    off = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int32)
    g = torch.Generator().manual_seed(dh + shift)
    qkv = torch.randn(total + 7, 3 * D, generator=g).cuda()         # (rows past the last sequence exist and hold data: the kernel must not read them into a result)
    out = torch.full((total, D), float("nan"), device="cuda")
    d_off, d_len = torch.from_numpy(off).cuda(), torch.tensor(lens, dtype=torch.int32).cuda()
    p = lambda t, o=0: C.c_void_p(t.data_ptr() + 4 * o)
    check(lib.mmdm_attention_ragged_f32(p(qkv), 3 * D, p(qkv, D), 3 * D, p(qkv, 2 * D), 3 * D, p(out), D, nseq, p(d_off), p(d_len), max(lens), total, H, dh, shift, None))
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    for s, (o, t) in enumerate(zip(off, lens)):
        ks = (s + shift) % nseq
        ko = int(off[ks])
        assert lens[ks] == t
        ref = torch.empty(t, D, device="cuda")
        q = qkv[o:o + t].contiguous()
        kv = qkv[ko:ko + t].contiguous()
        check(lib.mmdm_attention_f32(p(q), 3 * D, p(kv, D), 3 * D, p(kv, 2 * D), 3 * D, p(ref), D, 1, t, t, H, dh, 0, None))
        torch.cuda.synchronize()
        assert torch.equal(out[o:o + t], ref), (s, t)


# ---------------------------------------------------------------------------------------------------
# (a) handles that share one weight set
# ---------------------------------------------------------------------------------------------------
def test_shared_handles_give_the_parents_bits_and_overlap_safely():
    s = small(max_batch=2, max_frames=64)
    s.set_schedule("ddim20")
    kids = [s.share(), s.share(max_batch=1, max_frames=48)]
    for k in kids:
        k.set_schedule("ddim20")
    conds, xs = zip(*[(torch.randn(1, 8 * 768, generator=torch.Generator().manual_seed(i)), torch.randn(1, T, 524, generator=torch.Generator().manual_seed(50 + i)))
                      for i, T in enumerate((40, 48, 33, 40, 17, 48))])
    ref = [s.sample(c, x) for c, x in zip(conds, xs)]
    # six calls dealt over three handles, nothing synchronised in between
    pool = [s] + kids
    pend = [pool[i % 3].sample_async(c, x) for i, (c, x) in enumerate(zip(conds, xs))]
    for (out, _, ev), r in zip(pend, ref):
        ev.synchronize()
        assert torch.equal(out, r)
    # a shared handle may run another schedule than its parent, and refuses weights of its own
    kids[0].set_schedule("ddim50")
    s.set_schedule("ddim20")
    assert torch.equal(s.sample(conds[0], xs[0]), ref[0])
    from mixermdm_amd._lib import MMDMError
    with pytest.raises(MMDMError, match="borrows its weights"):
        kids[0].load_state_dict({"denoiser1.out.linear.bias": torch.zeros(262)})
    # destroy order is free: the parent first, the children keep the weights alive
    s.close()
    k = kids[1]
    k.set_schedule("ddim20")
    assert torch.equal(k.sample(conds[2], xs[2]), ref[2])
    for k in kids:
        k.close()


# ---------------------------------------------------------------------------------------------------
# (b) ragged batches: every item == the same item sampled alone, bitwise
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision", ["fp32", "fp32_split", "bf16", "bf16_fp8"])
@pytest.mark.parametrize("mode", [4, 2, 3, 1])
def test_ragged_batch_items_equal_their_stand_alone_calls(precision, mode):
    if precision != "fp32" and mode != 4:
        pytest.skip("the mixing modes differ after the stacks: covered in fp32")
    s = small(mode=mode, precision=precision)
    s.set_schedule("ddim20")
    cond, xs = inputs(LENS, seed=mode)
    alone, alone_hist = [], []
    for b, x in enumerate(xs):
        out, hist = s.sample(cond[b:b + 1], x[None], history=("influence_i1", "influence_i2", "out_influenced"), history_every=5)
        alone.append(out[0])
        alone_hist.append(hist)
    items, hist, ev = s.sample_ragged_async(cond, xs, LENS, history=("influence_i1", "influence_i2", "out_influenced"), history_every=5)
    ev.synchronize()
    assert s.rows % 128 == 0 and s.rows >= sum(LENS)
    for b, ((o, t), it) in enumerate(zip(s.item_slices(), items)):
        assert torch.equal(it, alone[b]), (precision, mode, b, (it - alone[b]).abs().max().item())
        for k, v in hist.items():            # [slots, 2, rows, C] vs the stand-alone [slots, 2B = 2, T, C]
            assert torch.equal(v[:, :, o:o + t], alone_hist[b][k]), (k, b)
    # eager == graph, and a batch in another order gives the same motions
    perm = [3, 0, 4, 2, 1]
    items2 = s.sample_ragged(cond[perm], [xs[i] for i in perm], [LENS[i] for i in perm], use_graph=False)
    for j, i in enumerate(perm):
        assert torch.equal(items2[j], alone[i])
    s.close()


def test_ragged_single_person_sampler():
    s = small(single_only=True)
    s.set_schedule("ddim20")
    cond, xs = inputs(LENS, width=262, cw=768, seed=5)
    alone = [s.sample(cond[b:b + 1], x[None])[0] for b, x in enumerate(xs)]
    for it, ref in zip(s.sample_ragged(cond, xs, LENS), alone):
        assert torch.equal(it, ref)
    s.close()


def test_ragged_graphs_are_keyed_by_row_bucket_not_by_lengths():
    s = small()
    s.set_schedule("ddim20")
    a = (40, 17, 64, 1, 33)        # 155 frames -> 256 rows, longest item 64 -> one query tile
    b = (64, 60, 10, 50, 20)       # 204 frames -> 256 rows, one query tile: the SAME graph
    c = (30, 30, 30, 30)           # another B: another graph
    for lens in (a, b, a, c, b):
        cond, xs = inputs(lens, seed=sum(lens))
        ref = s.sample_ragged(cond, xs, lens, use_graph=False)
        got = s.sample_ragged(cond, xs, lens, use_graph=True)
        for x, y in zip(ref, got):
            assert torch.equal(x, y)
    cap, rep, cached = s.graph_stats()
    assert (cap, cached) == (2, 2) and rep == 5 * 20, (cap, rep, cached)
    # uniform calls keep their own keys
    cond, xs = inputs((33,), seed=1)
    s.sample(cond, xs[0][None])
    assert s.graph_stats()[0] == 3
    s.close()


def test_ragged_argument_errors():
    from mixermdm_amd._lib import MMDMError
    s = small(max_batch=4, max_frames=32)
    s.set_schedule("ddim20")
    cond, xs = inputs((8, 40), seed=2)
    with pytest.raises(MMDMError, match="max_frames"):
        s.begin_ragged(cond, xs, (8, 40))
    cond, xs = inputs((8,) * 5, seed=2)
    with pytest.raises(MMDMError, match="max_batch"):
        s.begin_ragged(cond, xs, (8,) * 5)
    with pytest.raises(ValueError):
        s.begin_ragged(cond[:2], xs[:3], (8, 8))
    s.close()


# ---------------------------------------------------------------------------------------------------
# the real model sizes: a ragged batch against the stand-alone calls (bitwise) and one ragged step against the ORACLE
# ---------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def full():
    from mixermdm_amd.sampler import Sampler
    from mixermdm_amd.synthetic import synthetic_state_dict, synthetic_stats, FULL_DIMS
    sd = synthetic_state_dict(seed=0, std=0.02, bias_std=0.02, **FULL_DIMS)
    st = synthetic_stats()
    made = {}

    def get(precision):
        if precision not in made:
            s = Sampler(d_heads=8, m_heads=8, max_batch=4, max_frames=300, precision=precision, **FULL_DIMS)
            s.load_state_dict(sd)
            s.set_norm_stats(st["mean_hml"], st["std_hml"], st["mean_ih"], st["std_ih"])
            s.prepare()
            made[precision] = s
        return made[precision]
    yield get, sd, st
    for s in made.values():
        s.close()


@pytest.mark.parametrize("precision", ["fp32", "fp32_split", "bf16_fp8"])
def test_full_size_ragged_batch_is_bitwise_the_stand_alone_calls(full, precision):
    get, _, _ = full
    s = get(precision)
    s.set_schedule("ddim50")
    lens = (299, 60, 133, 196)
    cond, xs = inputs(lens, seed=9)
    alone = []
    for b, x in enumerate(xs):
        s.begin(cond[b:b + 1], x[None])
        s.run(3)
        alone.append({k: v[0].clone() for k, v in s.state().items() if v is not None})
    s.begin_ragged(cond, xs, lens)
    s.run(3)
    st = s.state()
    for b, (o, t) in enumerate(s.item_slices()):
        for k in ("x", "x2", "pred_xstart", "pred_xstart2", "model_out"):
            assert torch.equal(st[k][o:o + t], alone[b][k]), (precision, b, k, (st[k][o:o + t] - alone[b][k]).abs().max().item())


@pytest.mark.parametrize("precision", ["fp32", "fp32_split", "bf16_fp8"])
def test_full_size_handles_side_by_side_keep_their_bits(full, precision):
    """Three handles over one weight set at the real sizes, eight calls dealt round-robin with nothing synchronised in between: every motion is
    the sequential loop's, bit for bit, in every precision mode, with nothing in the library serialising the handles (the second round only
    replays cached step graphs: the handles overlap for whole calls).  Round 5: this test was red for the low-precision modes until the cause was
    found -- on gfx950 a packed-fp32 VALU instruction (v_pk_*_f32) can transiently deliver a wrong result while its wave shares a SIMD
    with the packed-W GEMM kernels, and the geometry kernels' rotation round trip turned that bit into a turned joint; geometry.hip
    is built without those instructions (mixermdm_amd/build.py NO_PACKED_FP32; tools/canary.hip, tools/overlap_bisect.py)."""
    get, _, _ = full
    s = get(precision)
    s.set_schedule("ddim20")
    kids = [s.share(max_batch=1), s.share(max_batch=1)]
    for k in kids:
        k.set_schedule("ddim20")
    pool = [s] + kids
    items = []
    for i, T in enumerate((181, 97, 263, 140, 181, 97, 263, 140)):
        g = torch.Generator().manual_seed(70 + i)
        items.append((torch.randn(1, 8 * 768, generator=g).cuda(), torch.randn(1, T, 524, generator=g).cuda()))
    ref = [s.sample(c, x) for c, x in items]
    for rnd in range(2):
        outs = [torch.empty_like(x) for _, x in items]
        torch.cuda.synchronize()
        for i, (c, x) in enumerate(items):
            pool[i % 3].enqueue(c, x, outs[i])
        torch.cuda.synchronize()
        for i, (o, r) in enumerate(zip(outs, ref)):
            assert torch.equal(o, r), (precision, rnd, i, (o - r).abs().max().item())
    for k in kids:
        k.close()


@pytest.mark.parametrize("pair", [("fp32", "fp32_split"), ("fp32", "bf16_fp8"), ("bf16_fp8", "fp32_split")])
def test_full_size_handles_of_different_precisions_side_by_side_keep_their_bits(full, pair):
    """The pairs that actually broke in round 5's hunt (LAB_NOTES.md step 1: an fp32 handle as VICTIM beside a low-precision aggressor -- 11 of
    32 steps wrong beside fp32_split, 6 of 32 beside bf16 -- and two different low-precision modes side by side): two independent full-size
    handles, each with its own weights, eight calls dealt alternately with nothing synchronised in between, three rounds that only replay
    cached step graphs (the setting in which EVERY overlapped call was wrong before the fix): every motion is bitwise what its own handle
    samples alone.  What keeps it green: geometry.hip without packed-fp32 instructions for every handle, rowops_nopk.o for precision 1-3
    handles (mixermdm_amd/build.py)."""
    get, _, _ = full
    a, b = get(pair[0]), get(pair[1])
    for h in (a, b):
        h.set_schedule("ddim20")
    items = []
    for i, T in enumerate((181, 97, 263, 140, 181, 97, 263, 140)):
        g = torch.Generator().manual_seed(170 + i)
        items.append((torch.randn(1, 8 * 768, generator=g).cuda(), torch.randn(1, T, 524, generator=g).cuda()))
    pool = [a, b]
    ref = [pool[i % 2].sample(c, x) for i, (c, x) in enumerate(items)]        # (every handle has now captured its shapes: the rounds below replay)
    torch.cuda.synchronize()
    for rnd in range(3):
        outs = [torch.empty_like(x) for _, x in items]
        torch.cuda.synchronize()
        for i, (c, x) in enumerate(items):
            pool[i % 2].enqueue(c, x, outs[i])
        torch.cuda.synchronize()
        for i, (o, r) in enumerate(zip(outs, ref)):
            assert torch.equal(o, r), (pair, pair[i % 2], rnd, i, (o - r).abs().max().item())


def test_serialize_switch_and_graph_cache_do_not_evict_while_sharing(full):
    """ While handles share a weight set a graph cache must not evict: evicted execs cannot be destroyed beside another handle's live
    execs in this runtime, so eviction would park one exec per re-captured shape without bound.  A kid handle samples more distinct shapes than
    its cache capacity (8): nothing is re-captured on the second pass, nothing is parked while it lives; closing it parks exactly its execs."""
    get, _, _ = full
    s = get("fp32")
    s.set_schedule("ddim20")
    kid = s.share(max_batch=1, max_frames=64)
    kid.set_schedule("ddim20")
    parked0 = s.graphs_parked()
    g = torch.Generator().manual_seed(5)
    cond = torch.randn(1, 8 * 768, generator=g).cuda()
    shapes = list(range(20, 31))                                              # 11 distinct (1, T) shapes > graph_cap = 8
    for rnd in range(2):
        for T in shapes:
            x = torch.randn(1, T, 524, generator=g).cuda()
            kid.begin(cond, x)
            kid.run(2)
        cap, rep, n = kid.graph_stats()
        assert cap == len(shapes) and n == len(shapes), (rnd, cap, n)         # every shape captured once, all of them still cached
        assert s.graphs_parked() == parked0
    torch.cuda.synchronize()
    kid.close()
    assert s.graphs_parked() == parked0 + len(shapes)


_DENORM = None


def _oracle_setup(sd, st):
    W = dict(sd)
    W["sequence_pos_encoder.pe"] = pe_table(512)
    W["denoiser1.sequence_pos_encoder.pe"] = pe_table(1024)
    W["denoiser2.sequence_pos_encoder.pe"] = pe_table(1024)
    ostats = tuple(torch.as_tensor(st[k]) for k in ("mean_hml", "std_hml", "mean_ih", "std_ih"))
    mh, sh, mi, si = [v.double().flatten() for v in ostats]
    global _DENORM          # normalisation of the pose tensors of a step at i > 0 (tests/parity_tol.py: the half-turn evidence for a component beyond the hard bound)
    _DENORM = {"pred_xstart": (mh.repeat(2), sh.repeat(2)), "pred_xstart2": (mi.repeat(2), si.repeat(2))}
    return W, to64(W), ostats, OS.make_schedule("cosine", 1000, "ddim50"), MX.MixerSpec(d_heads=8, m_heads=8)


def test_full_size_ragged_step_against_the_oracle(full):
    """One DDIM step of ragged batches (T = 32, 24, 40) at D = 1024 / 512, L = 8 / 4 against the oracle's step on every item alone -- the ragged
    path's own parity statement, as a STATISTICAL one (tests/parity_tol.py, "statistical form"): four batches from the consecutive seeds 0-3,
    none selected (seed 3 holds the draw that made round 5 pick seed 4: item 2, person 0 at 10 x the CPU oracle's own error), 24 (item, person)
    draws; every draw keeps the hard bounds, at most binomial_bound(24) = 5 may exceed 12 x the CPU fp32 oracle's own error in their group, and
    the pooled error quantiles sit within 3 x the CPU fp32 oracle's for EVERY channel class."""
    get, sd, st = full
    s = get("fp32")
    s.set_schedule("ddim50")
    lens = (32, 24, 40)
    W, W64, ostats, sch, spec = _oracle_setup(sd, st)
    steps = []
    for seed in range(4):
        cond, xs = inputs(lens, seed=seed)
        s.begin_ragged(cond, xs, lens)
        s.run(1)
        got = {k: v.clone() for k, v in s.state().items() if v is not None}
        for b, (o, t) in enumerate(s.item_slices()):
            r32, r64 = oracle_step_pair(W, spec, ostats, sch, 3.5, 49, xs[b][None], xs[b][None], cond[b:b + 1], W64=W64)
            steps.append(({k: got[k][o:o + t][None].cpu() for k in r32}, r32, r64, f"ragged seed {seed} item {b} (T={t})"))
    beyond, k, n = compare_draws(steps, "ragged step, seeds 0-3 x 3 items x 2 persons [fp32]", denorm=_DENORM)
    assert n == 24


def test_full_size_uniform_step_draws_against_the_oracle(full):
    """The uniform twin of the test above: three B = 4, T = 32 batches from the consecutive seeds 0-2 (24 (sample, person) draws), one ddim50
    step each, the same three statements."""
    get, sd, st = full
    s = get("fp32")
    s.set_schedule("ddim50")
    W, W64, ostats, sch, spec = _oracle_setup(sd, st)
    steps = []
    for seed in range(3):
        g = torch.Generator().manual_seed(seed)
        cond, x = torch.randn(4, 8 * 768, generator=g), torch.randn(4, 32, 524, generator=g)
        s.begin(cond, x)
        s.run(1)
        got = {k: v.clone().cpu() for k, v in s.state().items() if v is not None}
        r32, r64 = oracle_step_pair(W, spec, ostats, sch, 3.5, 49, x, x, cond, W64=W64)
        steps.append(({k: got[k] for k in r32}, r32, r64, f"uniform seed {seed} (B=4, T=32)"))
    beyond, k, n = compare_draws(steps, "uniform step, seeds 0-2 x 4 samples x 2 persons [fp32]", denorm=_DENORM)
    assert n == 24


# ---------------------------------------------------------------------------------------------------
# the facade and the evaluation harness
# ---------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def model():
    import os
    from mixermdm_amd.configs import get_config
    from mixermdm_amd.models import MixerMDM
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = get_config(os.path.join(root, "configs", "models", "MixerMDM.yaml"))
    m = MixerMDM(cfg, sampling_strategy="ddim20", config_root=root)
    m.init_synthetic(seed=0)
    m = m.to("cuda:0")
    m.eval()
    return m


def _batches(lens, reps, seed=0):
    out = []
    for i, (t, r) in enumerate(zip(lens, reps)):
        g = torch.Generator().manual_seed(seed + i)
        out.append({"cond": torch.randn(r, 8 * 768, generator=g).cuda(), "x_T": torch.randn(r, t, 524, generator=g).cuda(),
                    "motion_lens": torch.tensor([t]), "text": ["x"] * r})
    return out


@pytest.mark.parametrize("mode", ["eval_intermediate", "eval"])
def test_sample_many_equals_the_references_loop(model, mode):
    lens, reps = (120, 47, 196, 64, 299), (1, 1, 2, 1, 1)
    batches = _batches(lens, reps)
    call = model.forward if mode == "eval" else model.forward_test
    ref = [call(dict(b)) for b in batches]
    ref = [{k: (v.clone() if torch.is_tensor(v) else [t.clone() for t in v]) for k, v in r.items()} for r in ref]
    for batching, kw in (("sequential", {}), ("inflight", dict(inflight=3)), ("ragged", dict(max_rows=600)), ("ragged", {})):
        got = model.sample_many([dict(b) for b in batches], mode=mode, batching=batching, **kw)
        for r, g, t, nb in zip(ref, got, lens, reps):
            assert g["output"].shape == (nb, t, 524)
            assert torch.equal(g["output"], r["output"]), (batching, t)
            assert set(g) == set(r)
            for k in r:
                if k == "output":
                    continue
                assert len(g[k]) == len(r[k]) == 20, (k, len(g[k]))
                for a, b in zip(g[k], r[k]):
                    assert a.shape == b.shape and torch.equal(a, b), (batching, k, t)
    # outputs only
    got = model.sample_many([dict(b) for b in batches], mode=mode, batching="ragged", keep_history=False)
    assert all(torch.equal(g["output"], r["output"]) and g["influence_i1"] == [] for g, r in zip(got, ref))


def test_evaluation_harness_batching_modes_agree(model):
    from mixermdm_amd.generation import generate_for_evaluation
    lens = (33, 120, 64, 196, 47, 150)
    items = [{"text": ("a",), "text_individual1": ("b",), "text_individual2": ("c",), "motion_lens": torch.tensor([t]),
              "cond": torch.randn(1, 8 * 768, generator=torch.Generator().manual_seed(i))} for i, t in enumerate(lens)]
    runs = {}
    for batching in ("sequential", "inflight", "ragged"):
        gen, mm = generate_for_evaluation(model, items, max_length=300, mm_idxs=(1, 4), mm_num_repeats=3, batching=batching, seed=11)
        assert len(gen) == len(items) and len(mm) == 2 and mm[0]["mm_motions"].shape == (3, 300, 2, 262)
        runs[batching] = (gen, mm)
    for batching in ("inflight", "ragged"):
        for a, b in zip(runs["sequential"][0], runs[batching][0]):
            assert np.array_equal(a["motion1"], b["motion1"]) and np.array_equal(a["motion2"], b["motion2"])
        for a, b in zip(runs["sequential"][1], runs[batching][1]):
            assert np.array_equal(a["mm_motions"], b["mm_motions"])


def test_serialize_handles_switch_runs_and_keeps_the_bits():
    """MMDM_SERIALIZE_HANDLES=1 (read by the first mmdm_create of a process: a child process here): every sampling call waits on the device for the
    previous one of any handle; results are the sequential ones, as without the switch."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, torch
sys.path.insert(0, %r)
from mixermdm_amd.sampler import Sampler
from mixermdm_amd.synthetic import synthetic_state_dict, synthetic_stats
DIMS = dict(d_latent=128, d_ff=256, d_layers=2, m_latent=128, m_ff=256, m_layers=2)
sd = synthetic_state_dict(seed=7, std=0.05, bias_std=0.02, **DIMS); st = synthetic_stats()
def make(prec):
    s = Sampler(d_heads=2, m_heads=2, max_batch=2, max_frames=64, precision=prec, **DIMS)
    s.load_state_dict(sd); s.set_norm_stats(st["mean_hml"], st["std_hml"], st["mean_ih"], st["std_ih"]); s.prepare(); s.set_schedule("ddim20")
    return s
a, b = make("fp32"), make("fp32_split")
items = []
for i, T in enumerate((40, 17, 64, 33)):
    g = torch.Generator().manual_seed(300 + i)
    items.append((torch.randn(1, 8 * 768, generator=g).cuda(), torch.randn(1, T, 524, generator=g).cuda()))
pool = [a, b]
ref = [pool[i %% 2].sample(c, x) for i, (c, x) in enumerate(items)]
for rnd in range(2):
    outs = [torch.empty_like(x) for _, x in items]
    torch.cuda.synchronize()
    for i, (c, x) in enumerate(items):
        pool[i %% 2].enqueue(c, x, outs[i])
    torch.cuda.synchronize()
    assert all(torch.equal(o, r) for o, r in zip(outs, ref)), rnd
print("SERIAL_OK")
''' % root
    env = dict(os.environ, MMDM_SERIALIZE_HANDLES="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0 and "SERIAL_OK" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])
