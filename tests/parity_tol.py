"""Conditioning-aware comparison of one sampler step with the oracle (shared by the GPU parity tests; test infrastructure).

Tolerance of a step (STEP_TOL): all but 0.2 % of the elements within atol 2e-4 + rtol 2e-4, none beyond 5e-2; the tolerance of a person's
position / velocity channels is scaled by the conditioning factor of the global rotations that produced them.
"""
import torch

STEP_TOL = dict(atol=2e-4, rtol=2e-4, frac=2e-3, hard=5e-2)
# Near-degenerate joints (rot6d vectors almost collinear) amplify rounding through the Gram-Schmidt / quaternion round trip to ~1e-3 in
# single rot6d components; `frac` allows 0.2 % of them, and never fewer than this many elements (each still inside `hard`).
MIN_OUTLIERS = 4
HARD_OUTLIERS = 2

# Conditioning of the reference's two global rotations (geometry near its branch points needs a discriminant-aware comparison).
# Both are qbetween(u, v) of two directions and both are applied to whole position / velocity sequences:
#   * align_motions turns the individual model's motion by the angle between two root-displacement DIRECTIONS (alignment.py:84-101).  A
#     pre-geometry difference e in the root positions (HIP vs CPU rounding through 8-16 blocks: ~2e-5) turns the sequence by e / |disp| and
#     moves a position at distance `reach` from the pivot by reach * e / |disp|: that stays inside the 2e-4 tolerance only while
#     |disp| >= 0.1 reach.  The random-weight individual model barely moves its root (|disp| / reach = 0.04-0.07), so the factor
#     kappa = 0.1 reach / |disp| (>= 1) scales the tolerance of that person's position / velocity channels.
#   * qbetween is singular for anti-parallel directions: w = 1 + u.v -> 0 and |u x v| -> 0, and an fp32 rounding of w (1e-7) turns the
#     sequence by 2e-7 / sqrt(2 w): beyond the tolerance at 8 m reach once w < 3e-5.  center_motion (alignment.py:188-206) rotates every
#     person to face +Z, so person 2 of a pair facing each other -- where these trajectories sit for many late steps -- is exactly that
#     half turn.  kappa = sqrt(3e-4 / w) (10x margin on the figure above).
# Persons with kappa > KAPPA_MASK are on the branch point: their position / velocity channels are compared for sanity only (finite,
# |err| within the motion's own extent).  Rotation-6D and foot-contact channels are never affected and always meet the plain tolerance.
KAPPA_MASK = 25.0


def kappa(hist, B):
    """Oracle diagnostics of one step -> (kappa_chain1 [B, 2 persons], kappa_chain2 [B, 2]) tolerance factors >= 1."""
    k_align = torch.ones(B, 2, dtype=torch.float64)
    for p, d in enumerate(hist["align_diag"][-2:]):            # rows: B cond + B uncond of the CFG-doubled batch
        k = torch.maximum(0.1 * d["reach"] / torch.minimum(d["disp_target"], d["disp_moved"]).clamp_min(1e-12), torch.sqrt(3e-4 / d["w"].clamp_min(1e-12))).double()
        k_align[:, p] = torch.maximum(k[:B], k[B:]).clamp_min(1.0)
    k_center = torch.ones(B, 2, dtype=torch.float64)
    for p, d in enumerate(hist.get("center_diag", [])[-2:]):
        k = torch.maximum(torch.sqrt(3e-4 / d["w"].clamp_min(1e-12)), 0.01 * d["reach"] / (d["across"] * d["fwd"]).clamp_min(1e-12)).double()
        k_center[:, p] = k.clamp_min(1.0)
    return torch.maximum(k_align, k_center), k_align



def compare_step(out, refs, what, hist, tol=STEP_TOL):
    """out / refs: {state name: tensor [B, T, 524]}; hist: the oracle's diagnostics of this very step (``hist=`` of mixer_ddim_step).
    Returns (worst out-of-tolerance fraction, whether a person had to be masked)."""
    first = next(iter(refs.values()))
    B, T = first.shape[:2]
    k1, k2 = kappa(hist, B)
    worst, masked = 0.0, False
    for nm, ref in refs.items():
        got, ref = out[nm].detach().cpu().double(), torch.as_tensor(ref).detach().cpu().double()
        assert got.shape == ref.shape, (nm, got.shape, ref.shape)
        assert torch.isfinite(got).all(), f"{what} {nm}: non-finite"
        kap = k1 if nm in ("x", "pred_xstart") else k2
        scale = torch.ones(B, T, 524, dtype=torch.float64)
        for p in range(2):
            scale[:, :, p * 262:p * 262 + 132] = kap[:, p, None, None]
        keep = scale <= KAPPA_MASK
        masked |= not bool(keep.all())
        d = (got - ref).abs()
        bad = (d > scale * (tol["atol"] + tol["rtol"] * ref.abs())) & keep
        nbad = int(bad.sum())
        frac = nbad / max(1, int(keep.sum()))
        if nbad <= MIN_OUTLIERS:          # tiny tensors (T = 1, 2): a fraction of 1048 elements is two elements; the outliers are single rot6d components
            frac = 0.0
        note = f"(conditioning factors per [sample, person]: {[[round(v, 1) for v in r] for r in kap.tolist()]})"
        assert frac <= tol["frac"], f"{what} {nm}: {frac:.2e} of the elements outside tolerance, max err {d[keep].max().item():.2e} {note}"
        if keep.any():
            rel = torch.where(keep, d / scale, torch.zeros_like(d))
            over = rel > tol["hard"]
            # A rot6d pair within ~1e-5 of collinear makes Gram-Schmidt amplify fp32 rounding by ~1e5 (one joint in ~1e5 on noise inputs): at
            # most HARD_OUTLIERS rot6d components per tensor may pass the hard bound; position / velocity channels never may.
            ch = torch.arange(524) % 262
            rot = ((ch >= 132) & (ch < 258))[None, None, :].expand_as(over)
            worst_idx = tuple(int(v) for v in torch.nonzero(rel == rel.max())[0])
            assert not bool((over & ~rot).any()), f"{what} {nm}: max err {d[keep].max().item():.2e} at {worst_idx} {note}"
            assert int(over.sum()) <= HARD_OUTLIERS, f"{what} {nm}: {int(over.sum())} rot6d components beyond the hard bound, max err {d[keep].max().item():.2e} at {worst_idx} {note}"
        # masked persons sit on the rotation's branch point: any rounding turns them by an arbitrary angle about the pivot, so their
        # position / velocity channels can only be checked for sanity -- finite, and inside the motion's own extent
        assert d.max().item() <= 2.0 * ref.abs().max().item() + 1.0, f"{what} {nm}: an element on a branch point is off by {d.max().item():.2e}"
        worst = max(worst, frac)
    return worst, masked
