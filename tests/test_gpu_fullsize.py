"""GPU: BASELINE config 3 at its full size (B=16, T=300, D=1024/512, 8+8+4 layers) through size-independent properties
(the CPU oracle needs ~25 s per step at this size, so values are pinned at reduced B/T in test_gpu_sampler.py):
determinism, shard independence, history contracts and the reference's quirks, over consecutive ddim1000 steps."""
import pytest
import torch

pytestmark = pytest.mark.gpu

B, T, STEPS = 16, 300, 3


@pytest.fixture(scope="module")
def big():
    from mixermdm_amd.sampler import Sampler
    from mixermdm_amd.synthetic import synthetic_state_dict, synthetic_stats, synthetic_inputs, FULL_DIMS
    sd = synthetic_state_dict(seed=0, std=0.02, bias_std=0.02, **FULL_DIMS)
    st = synthetic_stats()
    s = Sampler(d_heads=8, m_heads=8, max_batch=B, max_frames=T, **FULL_DIMS)
    s.load_state_dict(sd)
    s.set_norm_stats(st["mean_hml"], st["std_hml"], st["mean_ih"], st["std_ih"])
    s.prepare()
    s.set_schedule("ddim1000")
    cond, xT = synthetic_inputs(B, T)
    yield s, cond, xT, st
    s.close()


def run(s, cond, xT, steps, graph=True, hist=None):
    s.begin(cond, xT)
    bufs = s.set_history(hist, 1) if hist else None
    s.run(steps, use_graph=graph)
    st = s.state()
    return {k: v.clone() for k, v in st.items()}, bufs


def test_full_size_steps_are_finite_deterministic_and_graph_equals_eager(big):
    s, cond, xT, _ = big
    a, _ = run(s, cond, xT, STEPS, graph=True)
    b, _ = run(s, cond, xT, STEPS, graph=True)
    c, _ = run(s, cond, xT, STEPS, graph=False)
    for k in ("x", "x2", "pred_xstart", "pred_xstart2", "model_out"):
        assert torch.isfinite(a[k]).all(), k
        assert torch.equal(a[k], b[k]) and torch.equal(a[k], c[k]), k
    assert not torch.equal(a["x"], a["x2"])             # the two chains diverge after the first step (different normalisation spaces)
    assert a["x"].shape == (B, T, 524)


def test_full_size_rows_are_independent_of_the_batch(big):
    """Motion k of the B=16 batch == the same motion sampled alone, bit for bit: sharding the batch over GPUs cannot change results."""
    s, cond, xT, _ = big
    full, _ = run(s, cond, xT, STEPS)
    for k in (0, 7, 15):
        one, _ = run(s, cond[k:k + 1], xT[k:k + 1], STEPS)
        assert torch.equal(one["x"][0], full["x"][k]) and torch.equal(one["x2"][0], full["x2"][k]), k


def test_full_size_history_contract_and_reference_quirks(big):
    s, cond, xT, st = big
    s.set_schedule("ddim20")            # 20 history slots (1.6 GB) instead of 1000 (80 GB) -- the contract does not depend on S
    out, h = run(s, cond, xT, 2, hist=("influence_i1", "influence_i2", "out1", "out2", "out_influenced"))
    s.set_schedule("ddim1000")
    n = 2 * B
    assert h["influence_i1"].shape == (20, n, T, 262)
    w = h["influence_i1"][:2]
    assert (w >= 0).all() and (w <= 1).all()
    # mode 4 expansion (mixermdm.py:767-784): 22 joint weights x3 (pos) | same (vel) | first 21 x6 (rot) | weight 22 x4 (feet)
    j = w[..., :66].reshape(2, n, T, 22, 3)
    assert (j == j[..., :1]).all() and torch.equal(w[..., :66], w[..., 66:132])
    r = w[..., 132:258].reshape(2, n, T, 21, 6)
    assert (r == r[..., :1]).all() and torch.equal(r[..., 0], j[..., :21, 0])
    assert (w[..., 258:] == w[..., 258:259]).all()
    o1, o2, mix = h["out1"][:2], h["out2"][:2], h["out_influenced"][:2]
    assert torch.count_nonzero(o1[..., 258:262]) == 0 and torch.count_nonzero(o1[..., 520:524]) == 0
    # blend is a per-element lerp between the two predictions
    lo, hi = torch.minimum(o1, o2), torch.maximum(o1, o2)
    assert ((mix >= lo - 1e-4) & (mix <= hi + 1e-4)).all()
    # interaction stream passes positions / velocities / feet through the alignment untouched (quirk 2): denormalised d2 output
    # rot6d block re-orthonormalised: first two rows of a rotation matrix (interleaved) -> unit norm, orthogonal
    rot = o2[0, :4, :, 132:258].reshape(4, T, 21, 3, 2)
    a1, a2 = rot[..., 0], rot[..., 1]
    assert (a1.norm(dim=-1) - 1).abs().max() < 1e-4 and (a2.norm(dim=-1) - 1).abs().max() < 1e-4
    assert (a1 * a2).sum(-1).abs().max() < 1e-4
    # chain 1 is re-centred every step (process_xstart): first-frame root XZ of pred_xstart maps to the origin in HML3D space
    mean, std = st["mean_hml"].cuda(), st["std_hml"].cuda()
    p = out["pred_xstart"][..., :262] * std + mean
    assert p[:, 0, 0].abs().max() < 1e-3 and p[:, 0, 2].abs().max() < 1e-3
    assert (p[:, :, 1:66:3].amin(dim=(1, 2))).abs().max() < 1e-3                                            # feet on the floor: min Y == 0


def test_full_size_fp32_split_mode_is_deterministic_batch_independent_and_tracks_fp32(big):
    """precision="fp32_split" at full size: bitwise deterministic, graph == eager, motion k independent of the batch it is sampled in (the
    hybrid 256x128 / 128x64 tiling sends a row to different kernels depending on M: both accumulate in the same order), and within fp32
    rounding noise of the native path after STEPS steps."""
    from mixermdm_amd.sampler import Sampler
    from mixermdm_amd.synthetic import synthetic_state_dict, synthetic_stats, FULL_DIMS
    s32, cond, xT, st = big
    s32.set_schedule("ddim1000")
    ref, _ = run(s32, cond, xT, STEPS)
    sd = synthetic_state_dict(seed=0, std=0.02, bias_std=0.02, **FULL_DIMS)
    s = Sampler(d_heads=8, m_heads=8, max_batch=B, max_frames=T, precision="fp32_split", **FULL_DIMS)
    s.load_state_dict(sd)
    s.set_norm_stats(st["mean_hml"], st["std_hml"], st["mean_ih"], st["std_ih"])
    s.prepare()
    s.set_schedule("ddim1000")
    a, _ = run(s, cond, xT, STEPS, graph=True)
    b, _ = run(s, cond, xT, STEPS, graph=False)
    # yardstick: how far the NATIVE path moves when x_T is perturbed at the fp32 rounding level (1 ulp ~ 6e-8 relative) -- the network
    # (26 blocks, softmax, LayerNorm, atan2 geometry) amplifies rounding noise by 2-3 orders of magnitude, whichever kernel produced it
    g = torch.Generator().manual_seed(99)
    noisy, _ = run(s32, cond, xT * (1 + 1.2e-7 * torch.randn(xT.shape, generator=g)), STEPS)
    # robust distance: the 99th percentile of |difference| relative to the RMS of the reference (a handful of near-degenerate joints can
    # cross an atan2 / quaternion branch in one run and not in the other, whichever kernels produced the rounding; an RMS would be theirs)
    def dist(u, v):
        d = (u - v).abs().flatten().float()
        return (torch.quantile(d[torch.randperm(d.numel(), generator=torch.Generator().manual_seed(0))[:2_000_000]], 0.99) / v.pow(2).mean().sqrt()).item()
    for k in ("x", "x2", "pred_xstart2"):
        assert torch.isfinite(a[k]).all() and torch.equal(a[k], b[k]), k
        rel, yard = dist(a[k].cpu(), ref[k].cpu()), dist(noisy[k].cpu(), ref[k].cpu())
        print(f"{k}: fp32_split vs fp32 p99 {rel:.3e}; fp32 with 1-ulp input noise vs fp32 p99 {yard:.3e}")
        assert rel < 4 * yard + 1e-7, (k, rel, yard)
        bad = ((a[k] - ref[k]).abs() > 2e-4 + 2e-4 * ref[k].abs()).float().mean().item()
        assert bad <= 2e-3, (k, bad)                                 # and the step tolerance of the parity tests, mode against mode
    assert not torch.equal(a["x"], ref["x"])                # the split kernels really ran
    for k in (3, 15):
        one, _ = run(s, cond[k:k + 1], xT[k:k + 1], STEPS)
        assert torch.equal(one["x"][0], a["x"][k]) and torch.equal(one["x2"][0], a["x2"][k]), k
    s.close()


def test_configs3_shard_of_32_motions_equals_its_motions_sampled_alone():
    """BASELINE configs[3] (B = 256 over 8 GPUs): the per-GPU shard of 32 motions, the batch `bench.py --gpus 8` runs -- M = 38 400 GEMM rows,
    5120 attention workgroups.  Finite, deterministic under graph replay, and motions 0 / 17 / 31 of the shard are bit-identical to the same
    motions sampled alone (so the 8-way split of the 256 requests cannot change a result), in fp32 and fp32_split."""
    from mixermdm_amd.sampler import Sampler
    from mixermdm_amd.synthetic import synthetic_state_dict, synthetic_stats, synthetic_inputs, FULL_DIMS
    sd = synthetic_state_dict(seed=0, std=0.02, bias_std=0.02, **FULL_DIMS)
    st = synthetic_stats()
    B3 = 32
    cond, xT = synthetic_inputs(B3, T, seed_cond=61, seed_x=62)
    for mode in ("fp32", "fp32_split"):
        s = Sampler(d_heads=8, m_heads=8, max_batch=B3, max_frames=T, precision=mode, **FULL_DIMS)
        s.load_state_dict(sd)
        s.set_norm_stats(st["mean_hml"], st["std_hml"], st["mean_ih"], st["std_ih"])
        s.prepare()
        s.set_schedule("ddim1000")
        a, _ = run(s, cond, xT, 2, graph=True)
        b, _ = run(s, cond, xT, 2, graph=True)
        for k in ("x", "x2", "pred_xstart2"):
            assert torch.isfinite(a[k]).all() and torch.equal(a[k], b[k]), (mode, k)
        assert a["x"].shape == (B3, T, 524)
        for k in (0, 17, 31):
            one, _ = run(s, cond[k:k + 1], xT[k:k + 1], 2)
            assert torch.equal(one["x"][0], a["x"][k]) and torch.equal(one["x2"][0], a["x2"][k]), (mode, k)
        s.close()
