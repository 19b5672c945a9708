"""GPU: each HIP kernel (through the C ABI) against the CPU oracle on the same seeded inputs.

Tolerances (fp32): GEMM-based ops atol 2e-4 * sqrt(K/1024)-ish relative to unit-variance data -> stated per test;
geometry compared with the masked/percentile rule of tests/test_oracle_golden.py::close_frac.
"""
import math
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import layers as L          # noqa: E402  (checker only)
from oracle import geometry as G        # noqa: E402
from oracle import mixer as MX          # noqa: E402
from oracle import schedule as OS       # noqa: E402


def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def rnd(seed, *shape, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def assert_close(got, ref, atol, rtol, frac=0.0, hard=None, what=""):
    got = got.detach().cpu().double()
    ref = ref.detach().cpu().double()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert torch.isfinite(got).all(), what + ": non-finite output"
    d = (got - ref).abs()
    bad = d > atol + rtol * ref.abs()
    assert bad.double().mean().item() <= frac, f"{what}: {int(bad.sum())}/{bad.numel()} outside atol={atol} rtol={rtol}; max err {d.max().item():.3e}"
    if hard is not None:
        assert d.max().item() <= hard, f"{what}: max err {d.max().item():.3e} > {hard}"


# ---------------------------------------------------------------------------------------------------
# linear
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (300, 1024, 1024), (1200, 3072, 1024), (130, 262, 512), (77, 23, 512),
                                   (600, 1024, 2048), (1, 128, 768), (128, 2048, 1024)])
def test_linear_bias(M, N, K):
    from mixermdm_amd import ops
    x, w, b = rnd(1, M, K), rnd(2, N, K, scale=1 / math.sqrt(K)), rnd(3, N)
    ref = F.linear(x.double(), w.double(), b.double()).float()
    got = ops.linear(x.to(dev()), w.to(dev()), b.to(dev()))
    assert_close(got, ref, atol=2e-5, rtol=1e-5, what=f"linear {M}x{N}x{K}")


def test_linear_identity_asymmetric():
    """A = I against an asymmetric W catches a transposed C/D register map."""
    from mixermdm_amd import ops
    n = 256
    w = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 251) / 16.0
    got = ops.linear(torch.eye(n).to(dev()), w.to(dev()))
    assert torch.equal(got.cpu(), w.t().contiguous())


@pytest.mark.parametrize("epi", ["gelu", "silu", "resid", "pe"])
def test_linear_epilogues(epi):
    from mixermdm_amd import ops
    M, N, K, T = 330, 384, 256, 30
    x, w, b = rnd(4, M, K), rnd(5, N, K, scale=1 / math.sqrt(K)), rnd(6, N)
    y = F.linear(x.double(), w.double(), b.double())
    d = dev()
    if epi == "gelu":
        ref, got = F.gelu(y), ops.linear(x.to(d), w.to(d), b.to(d), "gelu")
    elif epi == "silu":
        ref, got = F.silu(y), ops.linear(x.to(d), w.to(d), b.to(d), "silu")
    elif epi == "resid":
        r = rnd(7, M, N)
        ref, got = y + r.double(), ops.linear(x.to(d), w.to(d), b.to(d), "resid", r.to(d))
    else:
        pe = rnd(8, 64, N)
        ref = y + pe.double()[torch.arange(M) % T]
        got = ops.linear(x.to(d), w.to(d), b.to(d), "pe", pe.to(d), period=T)
    assert_close(got, ref.float(), atol=2e-5, rtol=1e-5, what=epi)


def test_linear_residual_in_place():
    from mixermdm_amd import ops
    M, N, K = 200, 256, 128
    x, w, b, r = rnd(9, M, K), rnd(10, N, K, scale=0.1), rnd(11, N), rnd(12, M, N)
    ref = F.linear(x.double(), w.double(), b.double()) + r.double()
    rd = r.to(dev())
    ops.linear(x.to(dev()), w.to(dev()), b.to(dev()), "resid", rd, out=rd)
    assert_close(rd, ref.float(), atol=2e-5, rtol=1e-5)


def test_linear_strided_slices_k262():
    """motion_embed on a person slice of the [n,T,524] state (lda = 524, K = 262: no 16-byte alignment) and the
    final projection writing a 262-column half of a 524-wide output (ldc = 524)."""
    from mixermdm_amd import ops
    n, T, D = 3, 20, 128
    x = rnd(13, n * T, 524)
    w, b = rnd(14, D, 262, scale=0.06), rnd(15, D)
    xd = x.to(dev())
    for p in range(2):
        ref = F.linear(x[:, p * 262:(p + 1) * 262].double(), w.double(), b.double()).float()
        got = ops.linear(xd[:, p * 262:(p + 1) * 262], w.to(dev()), b.to(dev()))
        assert_close(got, ref, atol=2e-5, rtol=1e-5, what=f"embed person {p}")
    h = rnd(16, n * T, D)
    wo, bo = rnd(17, 262, D, scale=0.09), rnd(18, 262)
    out = torch.zeros(n * T, 524, device=dev())
    ops.linear(h.to(dev()), wo.to(dev()), bo.to(dev()), out=out[:, 262:])
    ref = F.linear(h.double(), wo.double(), bo.double()).float()
    assert_close(out[:, 262:], ref, atol=2e-5, rtol=1e-5)
    assert torch.count_nonzero(out[:, :262]).item() == 0


@pytest.mark.parametrize("M,N,K", [(777, 520, 1024), (130, 136, 128), (1000, 3000, 512), (129, 68, 2048), (19, 1028, 96)])
@pytest.mark.parametrize("epi", ["bias", "gelu", "resid", "pe"])
def test_linear_writes_only_its_window(M, N, K, epi):
    """Ragged M and N on the pipelined kernels and the small-launch forms (rows past M are dropped by the buffer resource's range check, columns past N by a lane
    mask): the result goes into an interior window of a sentinel-filled buffer (ldc > N, rows above and below), and nothing outside
    the window may change -- in particular not the rows below it, which a store clipped only by its vector offset would reach."""
    from mixermdm_amd import ops
    from mixermdm_amd._lib import load_library
    d = dev()
    x, w, b = rnd(31, M, K), rnd(32, N, K, scale=1 / math.sqrt(K)), rnd(33, N)
    y = F.linear(x.double(), w.double(), b.double())
    extra = None
    if epi == "gelu":
        y = F.gelu(y)
    elif epi == "resid":
        extra = rnd(34, M, N)
        y = y + extra.double()
    elif epi == "pe":
        extra = rnd(35, 40, N)
        y = y + extra.double()[torch.arange(M) % 40]
    PADR, PADC = 70, 12                                   # 12 floats keep the window's rows 16-byte aligned (the 16-byte epilogue stays eligible)
    buf = torch.full((M + 2 * PADR, N + 2 * PADC), 7.25, device=d)
    win = buf[PADR:PADR + M, PADC:PADC + N]
    ops.linear(x.to(d), w.to(d), b.to(d), epi, extra.to(d) if extra is not None else None, period=40 if epi == "pe" else 0, out=win)
    kern = load_library().mmdm_last_gemm_kernel().decode()
    assert kern.startswith(("gemm_pipe<", "gemm_s16<", "gemm_mix<")), kern          # the pipelined tiles, the 16 x 16-chain tiles, both in one launch
    assert_close(win, y.float(), atol=2e-5 * math.sqrt(max(1.0, K / 1024)), rtol=1e-5, what=f"{M}x{N}x{K} {epi} on {kern}")
    chk = buf.clone()
    chk[PADR:PADR + M, PADC:PADC + N] = 7.25
    assert bool((chk == 7.25).all()), f"{kern}: wrote outside its {M}x{N} window"


def test_linear_split_and_bf16_write_only_their_window():
    """Same for the fp32-split kernel (buffer-store fp32 rows, hybrid 256x128 + 128x64 tiling) at a ragged M: rows past M untouched."""
    from mixermdm_amd import ops
    d = dev()
    M, N, K = 19200 + 77, 1024, 512
    x, w, b = rnd(36, M, K), rnd(37, N, K, scale=1 / math.sqrt(K)), rnd(38, N)
    xs, ws = ops.split_f32(x.to(d)), ops.split_f32(w.to(d))
    ref = F.linear(x[-300:].double(), w.double(), b.double()).float()
    got = ops.linear_split(xs, ws, b.to(d))
    assert got.shape == (M, N)
    assert_close(got[-300:], ref, atol=2e-5, rtol=1e-5, what="linear_split tail rows")
    # the allocation behind `got` ends with row M-1: a write past it would have faulted or hit the caching allocator's next block; check the
    # block that follows in a fresh sentinel-filled arena instead
    arena = torch.full((M + 256, N), 3.5, device=d)
    import ctypes as C
    from mixermdm_amd._lib import load_library
    from mixermdm_amd.ops import EPI
    lib = load_library()
    rc = lib.mmdm_linear_split(C.c_void_p(xs.data_ptr()), K, M * K, C.c_void_p(ws.data_ptr()), K, N * K, C.c_void_p(b.to(d).data_ptr()),
                               C.c_void_p(arena.data_ptr()), N, 0, 0, M, N, K, EPI["bias"], None, 0, 0, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(arena[:M], got) and bool((arena[M:] == 3.5).all())


def test_linear_empty_and_errors():
    from mixermdm_amd import ops, MMDMError
    d = dev()
    assert ops.linear(torch.zeros(0, 64, device=d), torch.zeros(32, 64, device=d)).shape == (0, 32)
    with pytest.raises(MMDMError, match="needs `extra`"):
        ops.linear(torch.zeros(4, 64, device=d), torch.zeros(32, 64, device=d), None, "resid")


# ---------------------------------------------------------------------------------------------------
# AdaLN / cond / head
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("D", [16, 64, 512, 1024, 2048])
def test_adaln(D):
    from mixermdm_amd import ops
    nseq, T, rows = 6, 13, 3
    h, ss = rnd(20, nseq, T, D) * 3 + 0.5, rnd(21, rows, 2 * D)
    hn = F.layer_norm(h.double(), (D,), eps=1e-6)
    idx = torch.arange(nseq) % rows
    ref = hn * (1 + ss.double()[idx, None, :D]) + ss.double()[idx, None, D:]
    got = ops.adaln(h.to(dev()), ss.to(dev()), rows)
    assert_close(got, ref.float(), atol=2e-5, rtol=2e-5, what=f"adaln D={D}")


def test_adaln_strided_ss():
    from mixermdm_amd import ops
    nseq, T, D = 4, 7, 64
    h, big = rnd(22, nseq, T, D), rnd(23, nseq, 6 * D)
    ss = big[:, 2 * D:4 * D]
    ref = F.layer_norm(h, (D,), eps=1e-6) * (1 + ss[:, None, :D]) + ss[:, None, D:]
    got = ops.adaln(h.to(dev()), big.to(dev())[:, 2 * D:4 * D])
    assert_close(got, ref, atol=2e-5, rtol=2e-5)


def test_cond_silu_and_head_and_mean():
    from mixermdm_amd import ops
    d = dev()
    S, rows, D = 5, 7, 96
    tab, txt = rnd(24, S, D), rnd(25, rows, D)
    step = torch.tensor([3], dtype=torch.int32, device=d)
    assert_close(ops.cond_silu(tab.to(d), step, txt.to(d)), F.silu(tab[3] + txt), atol=1e-6, rtol=1e-5)
    h, w, b = rnd(26, 2, 9, 512), rnd(27, 23, 512, scale=0.05), rnd(28, 23)
    assert_close(ops.influence_head(h.to(d), w.to(d), b.to(d)), torch.sigmoid(F.linear(h, w, b)), atol=2e-6, rtol=1e-5)
    w1, b1 = rnd(29, 1, 512, scale=0.05), rnd(30, 1)
    assert_close(ops.influence_head(h.to(d), w1.to(d), b1.to(d)), torch.sigmoid(F.linear(h, w1, b1)), atol=2e-6, rtol=1e-5)
    assert_close(ops.mean_time(h.to(d)), h.mean(dim=1), atol=2e-6, rtol=1e-5)


# ---------------------------------------------------------------------------------------------------
# attention
# ---------------------------------------------------------------------------------------------------
def ref_attention(q, k, v, H, shift=0):
    nseq, Tq, HD = q.shape
    dh = HD // H
    Tk = k.shape[1]
    idx = (torch.arange(nseq) + shift) % nseq
    k, v = k[idx], v[idx]
    qh = q.double().view(nseq, Tq, H, dh).transpose(1, 2)
    kh = k.double().view(nseq, Tk, H, dh).transpose(1, 2)
    vh = v.double().view(nseq, Tk, H, dh).transpose(1, 2)
    z = torch.zeros(nseq, H, 1, dh, dtype=torch.float64)
    kh, vh = torch.cat([kh, z], 2), torch.cat([vh, z], 2)     # add_zero_attn
    a = torch.softmax(qh @ kh.transpose(-1, -2) / math.sqrt(dh), dim=-1)
    return (a @ vh).transpose(1, 2).reshape(nseq, Tq, HD).float()


@pytest.mark.parametrize("dh,H,Tq,Tk,nseq", [(128, 8, 300, 300, 2), (128, 2, 64, 64, 3), (128, 1, 1, 1, 1), (128, 2, 65, 129, 2),
                                              (64, 8, 300, 300, 2), (64, 3, 17, 200, 2), (8, 2, 12, 12, 4), (32, 4, 40, 33, 2), (16, 1, 5, 70, 1)])
def test_attention_matches_reference(dh, H, Tq, Tk, nseq):
    from mixermdm_amd import ops
    q, k, v = rnd(31, nseq, Tq, H * dh), rnd(32, nseq, Tk, H * dh), rnd(33, nseq, Tk, H * dh)
    got = ops.attention(q.to(dev()), k.to(dev()), v.to(dev()), H)
    assert_close(got, ref_attention(q, k, v, H), atol=3e-6, rtol=1e-5, what=f"attn dh={dh} Tq={Tq} Tk={Tk}")


def test_attention_packed_qkv_and_shift():
    """Q/K/V as column slices of one packed [nseq,T,3D] projection, and the interaction denoiser's cross-stream keys."""
    from mixermdm_amd import ops
    nseq, T, H, dh = 4, 50, 2, 64
    D = H * dh
    qkv = rnd(34, nseq, T, 3 * D)
    d = qkv.to(dev())
    got = ops.attention(d[..., :D], d[..., D:2 * D], d[..., 2 * D:], H)
    assert_close(got, ref_attention(qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:], H), atol=3e-6, rtol=1e-5)
    got = ops.attention(d[..., :D], d[..., D:2 * D], d[..., 2 * D:], H, kv_seq_shift=2)
    assert_close(got, ref_attention(qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:], H, shift=2), atol=3e-6, rtol=1e-5)


def test_attention_large_logits_rescale_path():
    """Online-softmax rescale branch: one key per chunk dominates (rule: force the rare branch)."""
    from mixermdm_amd import ops
    nseq, T, H, dh = 1, 200, 1, 128
    q, k, v = rnd(35, nseq, T, dh), rnd(36, nseq, T, dh), rnd(37, nseq, T, dh)
    k[0, 70] = q[0, 10] * 3.0      # spike in chunk 1
    k[0, 150] = q[0, 10] * 6.0     # bigger spike in chunk 2
    k[0, 5] = -q[0, 20] * 5.0      # all logits negative for some rows -> the zero key dominates
    got = ops.attention(q.to(dev()), k.to(dev()), v.to(dev()), H)
    assert_close(got, ref_attention(q, k, v, H), atol=5e-6, rtol=1e-5)


def test_attention_all_negative_logits_zero_key():
    from mixermdm_amd import ops
    q = torch.ones(1, 4, 64)
    k = -torch.ones(1, 9, 64) * 4
    v = rnd(38, 1, 9, 64)
    got = ops.attention(q.to(dev()), k.to(dev()), v.to(dev()), 1)
    assert_close(got, ref_attention(q, k, v, 1), atol=1e-6, rtol=1e-5)
    assert got.abs().max().item() < 1e-6      # everything attends to the zero key


@pytest.mark.parametrize("dh", [128, 64])
def test_attention_and_linear_are_bitwise_reproducible_at_full_occupancy(dh):
    """Several workgroups per CU / waves per SIMD: the regime where a missing MFMA->VALU wait state (hipcc pads that
    hazard only inside a basic block) once made the online-softmax running max differ run to run."""
    from mixermdm_amd import ops
    nseq, T, H = 64, 300, 8
    D = H * dh
    qkv = rnd(39, nseq, T, 3 * D).to(dev())
    ref = ops.attention(qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:], H)
    for _ in range(5):
        assert torch.equal(ops.attention(qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:], H), ref)
    x, w, b = rnd(40, 19200, 512).to(dev()), rnd(41, 1024, 512, scale=0.05).to(dev()), rnd(42, 1024).to(dev())
    ref = ops.linear(x, w, b, "gelu")
    for _ in range(3):
        assert torch.equal(ops.linear(x, w, b, "gelu"), ref)


def test_attention_unsupported_head_dim():
    from mixermdm_amd import ops, MMDMError
    x = torch.zeros(1, 4, 300, device=dev())       # 4..32, 64, 128: dedicated kernels; any other size <= 256: wave-per-query fallback
    with pytest.raises(MMDMError, match="head dim"):
        ops.attention(x, x, x, 1)
    x = torch.randn(2, 9, 2 * 24, device=dev())    # dh = 24 runs on the fallback, with the zero key
    ref = torch.nn.functional.scaled_dot_product_attention
    xs = x.cpu().view(2, 9, 2, 24).transpose(1, 2)
    z = torch.zeros(2, 2, 1, 24)
    want = ref(xs, torch.cat([xs, z], 2), torch.cat([xs, z], 2)).transpose(1, 2).reshape(2, 9, 48)
    assert_close(ops.attention(x, x, x, 2), want, atol=2e-5, rtol=1e-4, what="attention dh=24")


# ---------------------------------------------------------------------------------------------------
# geometry / blend / ddim
# ---------------------------------------------------------------------------------------------------
def stats_t(seed=3):
    from mixermdm_amd.synthetic import synthetic_stats
    s = synthetic_stats(seed)
    return s, torch.cat([s["mean_hml"], s["std_hml"], s["mean_ih"], s["std_ih"]])


def plausible(seed, n, T):
    """Normalised-space motions whose denormalised rot6d block is near a valid rotation (like real denoiser outputs)."""
    m = rnd(seed, n, T, 524) * 0.5
    aa = rnd(seed + 1, n, T, 2, 21, 3)
    r6 = G.matrix_to_rotation_6d(G.quaternion_to_matrix(G.axis_angle_to_quaternion(aa))).reshape(n, T, 2, 126)
    return m, r6


GEO = dict(atol=5e-5, rtol=1e-4, frac=2e-4, hard=5e-2)


@pytest.mark.parametrize("align", [True, False])
@pytest.mark.parametrize("kind", ["gaussian", "plausible"])
def test_mixer_pre(align, kind):
    from mixermdm_amd import ops
    n, T = 4, 37
    s, st = stats_t()
    o1, o2 = rnd(40, n, T, 524), rnd(41, n, T, 524)
    if kind == "plausible":
        _, r6 = plausible(42, n, T)
        for p in range(2):   # make the DEnormalised rot block a near-rotation
            o1[..., p * 262 + 132:p * 262 + 258] = (r6[:, :, p] - s["mean_hml"][132:258]) / s["std_hml"][132:258] + 0.01 * o1[..., p * 262 + 132:p * 262 + 258]
            o2[..., p * 262 + 132:p * 262 + 258] = (r6[:, :, p].flip(0) - s["mean_ih"][132:258]) / s["std_ih"][132:258] + 0.01 * o2[..., p * 262 + 132:p * 262 + 258]
    refs1, refs2 = [], []
    for p in range(2):
        a = o1[..., p * 262:(p + 1) * 262] * s["std_hml"] + s["mean_hml"]
        c = o2[..., p * 262:(p + 1) * 262] * s["std_ih"] + s["mean_ih"]
        if align:
            sa, sc = G.ih_to_smpl(a), G.ih_to_smpl(c)
            a, c = G.smpl_to_ih(G.align_motions(sc, sa)), G.smpl_to_ih(sc)
        refs1.append(a)
        refs2.append(c)
    g1, g2 = ops.mixer_pre(o1.to(dev()), o2.to(dev()), st.to(dev()), align)
    assert_close(g1, torch.cat(refs1, -1), what="out1", **GEO)
    assert_close(g2, torch.cat(refs2, -1), what="out2", **GEO)
    if align:
        assert torch.count_nonzero(g1[..., 258:262]).item() == 0 and torch.count_nonzero(g1[..., 520:524]).item() == 0   # quirk 1


def test_mixer_pre_golden_fixture(golden):
    """Directly against vectors captured from the reference: Mixer history out1/out2 given the denoiser outputs is not
    stored, so use the geometry fixture: align_motions + smpl_to_ih on its motions (stats = identity)."""
    from mixermdm_amd import ops
    g, _, t = golden("geometry")
    ident = torch.cat([torch.zeros(262), torch.ones(262), torch.zeros(262), torch.ones(262)])
    for name, m1, m2 in [("rand", "rand_a", "rand_b"), ("valid", "valid_c", "valid_d")]:
        tgt, mov = t(m1), t(m2)                       # motion1 = target (interaction), motion2 = moved (individual)
        o1 = torch.cat([mov, mov], -1)
        o2 = torch.cat([tgt, tgt], -1)
        g1, g2 = ops.mixer_pre(o1.to(dev()), o2.to(dev()), ident.to(dev()), True)
        assert_close(g1[..., :262], t(name + ":align_m2_ih"), what=name + " moved", **GEO)
        assert_close(g2[..., :262], t(m1 + ":smpl_to_ih"), what=name + " target", **GEO)


@pytest.mark.parametrize("mode", [1, 2, 3, 4])
def test_blend_cfg(mode):
    from mixermdm_amd import ops
    B, T = 3, 11
    n = 2 * B
    out1, out2 = rnd(50, n, T, 524), rnd(51, n, T, 524)
    Tw, nw = (T if mode in (2, 4) else 1), (23 if mode >= 3 else 1)
    w = torch.rand(2, n, Tw, nw, generator=torch.Generator().manual_seed(52))
    for force in [None, 0.25]:
        i = [MX.expand_influence(w[p] if Tw > 1 else w[p][:, 0], mode, T) for p in range(2)]
        if mode == 2:
            i = [x.expand(-1, -1, 262) for x in i]
        if mode == 1:
            i = [x.expand(-1, -1, 262) for x in i]
        if force is not None:
            i = [torch.ones_like(x) * force for x in i]
        mix = torch.cat([out2[..., :262] + i[0] * (out1[..., :262] - out2[..., :262]),
                         out2[..., 262:] + i[1] * (out1[..., 262:] - out2[..., 262:])], -1)
        ref = 3.5 * mix[:B] + (1 - 3.5) * mix[B:]
        mo, h1, h2, hm = ops.blend_cfg(out1.to(dev()), out2.to(dev()), w.to(dev()), mode, 3.5, force, want_hist=True)
        assert_close(mo, ref, atol=2e-6, rtol=1e-5, what=f"mode {mode}")
        assert_close(hm, mix, atol=1e-6, rtol=1e-6)
        assert_close(h1, i[0].contiguous(), atol=0, rtol=0)
        assert_close(h2, i[1].contiguous(), atol=0, rtol=0)


def test_blend_bad_mode():
    from mixermdm_amd import ops
    z = torch.zeros(2, 3, 524, device=dev())
    with pytest.raises(ValueError, match="Mixing mode not recognized"):
        ops.blend_cfg(z, z, torch.zeros(2, 2, 3, 23, device=dev()), 7, 3.5)


@pytest.mark.parametrize("i", [31, 1, 0])
@pytest.mark.parametrize("align", [True, False])
def test_xstart_ddim(i, align):
    from mixermdm_amd import ops
    from mixermdm_amd.schedule import make_schedule
    B, T = 4, 29          # B != 3
    s, st = stats_t()
    stats = (s["mean_hml"], s["std_hml"], s["mean_ih"], s["std_ih"])
    m, r6 = plausible(60, B, T)
    for p in range(2):
        m[..., p * 262 + 132:p * 262 + 258] = r6[:, :, p] + 0.02 * m[..., p * 262 + 132:p * 262 + 258]
    x, x2 = rnd(61, B, T, 524), rnd(62, B, T, 524)
    osch = OS.make_schedule("cosine", 1000, "ddim50")
    p1, p2 = MX.process_xstart(m, stats, i > 0, align)
    rx, rx2 = MX.ddim_update(osch, i, x, p1), MX.ddim_update(osch, i, x2, p2)
    coef = torch.from_numpy(make_schedule("cosine", 1000, "ddim50").device_coefficients()).to(dev())
    xd, x2d = x.to(dev()), x2.to(dev())
    step = torch.tensor([i], dtype=torch.int32, device=dev())
    g1, g2 = ops.xstart_ddim(m.to(dev()), st.to(dev()), coef, step, xd, x2d, align)
    tol = dict(atol=1e-4, rtol=2e-4, frac=2e-4, hard=0.2)
    assert_close(g1, p1, what="pred_xstart", **tol)
    assert_close(g2, p2, what="pred_xstart2", **tol)
    assert_close(xd, rx, what="sample", **tol)
    assert_close(x2d, rx2, what="sample2", **tol)
    if i == 0:
        assert torch.equal(g1.cpu(), m) and torch.equal(g2.cpu(), m)      # quirk 6: raw model output at t == 0


def test_center_motion_golden_fixture(golden):
    """xstart path against the reference's own center_motion -> smpl_to_ih vectors (identity stats, i > 0)."""
    from mixermdm_amd import ops
    g, _, t = golden("geometry")
    ident = torch.cat([torch.zeros(262), torch.ones(262), torch.zeros(262), torch.ones(262)]).to(dev())
    coef = torch.tensor([[1.0, 1.0], [1.0, 1.0], [1.0, 1.0], [0.0, 0.0]], device=dev())
    step = torch.tensor([1], dtype=torch.int32, device=dev())
    for a, b in [("rand_a", "rand_b"), ("valid_c", "valid_d")]:
        m = torch.cat([t(a), t(b)], -1)
        x, x2 = torch.zeros_like(m).to(dev()), torch.zeros_like(m).to(dev())
        p1, p2 = ops.xstart_ddim(m.to(dev()), ident, coef, step, x, x2, True)
        assert_close(p1[..., :262], t(a + ":center_ih"), what=a, **GEO)
        assert_close(p1[..., 262:], t(b + ":center_ih"), what=b, **GEO)
        assert torch.equal(p2.cpu(), m)


def test_cfg_ddim_single_chain():
    from mixermdm_amd import ops
    from mixermdm_amd.schedule import make_schedule
    B, T = 2, 10
    m, x = rnd(70, 2 * B, T, 262), rnd(71, B, T, 262)
    osch = OS.make_schedule("cosine", 1000, "ddim50")
    x0 = 3.5 * m[:B] + (1 - 3.5) * m[B:]
    ref = MX.ddim_update(osch, 17, x, x0)
    coef = torch.from_numpy(make_schedule("cosine", 1000, "ddim50").device_coefficients()).to(dev())
    xd = x.to(dev())
    p = ops.cfg_ddim(m.to(dev()), coef, torch.tensor([17], dtype=torch.int32, device=dev()), 3.5, xd)
    assert_close(p, x0, atol=2e-6, rtol=1e-5)
    assert_close(xd, ref, atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("dh,H,T,zero_key,causal", [(128, 2, 300, True, False), (64, 3, 77, True, False), (128, 1, 45, False, False), (64, 2, 130, False, True)])
def test_attention_with_bf16_plane_scores(dh, H, T, zero_key, causal):
    """Q K^T from exactly split bf16 planes (six MFMAs per block) must reproduce the fp32 attention to fp32 accuracy; with one plane
    (bf16 Q, K) it must match a reference that rounds Q and K to bf16."""
    from mixermdm_amd import ops
    import math
    n = 3
    qkv = rnd(17, n, T, 3 * H * dh)
    d = qkv.to(dev())
    q, k, v = d[..., :H * dh], d[..., H * dh:2 * H * dh], d[..., 2 * H * dh:]

    def ref(qr, kr, vr):
        sp = lambda t: t.reshape(n, -1, H, dh).transpose(1, 2).double()
        qs, ks, vs = sp(qr), sp(kr), sp(vr)
        if zero_key:
            z = torch.zeros(n, H, 1, dh, dtype=torch.float64)
            ks, vs = torch.cat([ks, z], 2), torch.cat([vs, z], 2)
        sc = (qs @ ks.transpose(-1, -2)) / math.sqrt(dh)
        if causal:
            sc = sc + torch.full((T, T), float("-inf"), dtype=torch.float64).triu_(1)
        return (torch.softmax(sc, -1) @ vs).transpose(1, 2).reshape(n, T, H * dh).float()

    qc, kc, vc = q.cpu(), k.cpu(), v.cpu()
    got3 = ops.attention_planes(ops.bf16_split3(q.contiguous()), ops.bf16_split3(k.contiguous()), v, H, zero_key=zero_key, causal=causal)
    native = ops.attention(q, k, v, H, zero_key=zero_key, causal=causal)
    want = ref(qc, kc, vc)
    assert_close(got3, want, atol=2e-5, rtol=1e-4, what="attention, split planes")
    e3, en = (got3.cpu() - want).abs().mean().item(), (native.cpu() - want).abs().mean().item()
    assert e3 <= 2 * en + 1e-8, (e3, en)
    got1 = ops.attention_planes(q.contiguous().bfloat16()[None], k.contiguous().bfloat16()[None], v, H, zero_key=zero_key, causal=causal)
    assert_close(got1, ref(qc.bfloat16().float(), kc.bfloat16().float(), vc), atol=2e-5, rtol=1e-4, what="attention, bf16 Q/K")


@pytest.mark.gpu
@pytest.mark.parametrize("dh,H,T,zero_key,causal,shift", [(128, 2, 300, True, False, 0), (64, 3, 77, True, False, 1), (128, 1, 45, False, False, 0), (64, 2, 130, False, True, 0),
                                                          (128, 8, 299, True, False, 2)])
def test_attention_on_the_fp16_split_planes_is_as_accurate_as_the_fp32_kernel(dh, H, T, zero_key, causal, shift):
    """The fp32-split mode's attention (Q, K, V as two fp16 planes each, column slices of one packed projection as the sampler passes them; scores and P.V
    from three fp16 MFMAs with hi / lo accumulators) against a float64 reference: error within 1.5x of the fp32 MFMA kernel's on the same inputs, with
    large-magnitude rows (scores up to +-40) included; two-plane output = the split of the fp32 output."""
    from mixermdm_amd import ops
    import math
    n = 4
    qkv = rnd(23, n, T, 3 * H * dh)
    qkv[1] *= 2.5                                   # peaked softmax rows
    d = qkv.to(dev())
    HD = H * dh
    q, k, v = d[..., :HD], d[..., HD:2 * HD], d[..., 2 * HD:]
    sp = lambda t: t.reshape(n, -1, H, dh).transpose(1, 2).double()
    qs, ks, vs = sp(qkv[..., :HD]), sp(qkv[..., HD:2 * HD]).roll(-shift, 0), sp(qkv[..., 2 * HD:]).roll(-shift, 0)
    if zero_key:
        z = torch.zeros(n, H, 1, dh, dtype=torch.float64)
        ks, vs = torch.cat([ks, z], 2), torch.cat([vs, z], 2)
    sc = (qs @ ks.transpose(-1, -2)) / math.sqrt(dh)
    if causal:
        sc = sc + torch.full((T, T), float("-inf"), dtype=torch.float64).triu_(1)
    want = (torch.softmax(sc, -1) @ vs).transpose(1, 2).reshape(n, T, HD)
    planes = ops.split_f32(d)                       # [2, n, T, 3 HD]: what the QKV GEMM writes with split_out
    got = ops.attention_split(planes[..., :HD], planes[..., HD:2 * HD], planes[..., 2 * HD:], H, kv_seq_shift=shift, zero_key=zero_key, causal=causal)
    native = ops.attention(q, k, v, H, kv_seq_shift=shift, zero_key=zero_key, causal=causal)
    es, en = (got.cpu().double() - want).abs(), (native.cpu().double() - want).abs()
    assert_close(got, want.float(), atol=2e-5, rtol=1e-4, what="attention on fp16 split planes")
    assert es.mean() <= 1.5 * en.mean() + 1e-9 and es.max() <= 2.0 * en.max() + 1e-7, (es.mean().item(), en.mean().item(), es.max().item(), en.max().item())
    got2 = ops.attention_split(planes[..., :HD], planes[..., HD:2 * HD], planes[..., 2 * HD:], H, kv_seq_shift=shift, zero_key=zero_key, causal=causal, split_out=True)
    assert torch.equal(got2, ops.split_f32(got))


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(300, 128, 64), (257, 192, 128), (1000, 384, 1024), (64, 512, 512), (5, 1152, 192)])
def test_linear_split_packed_is_bitwise_the_plane_kernel(M, N, K):
    """W in fragment order (mmdm_split_pack_weight) and fetched straight from global memory: every tile shape of the packed dispatch
    (128x128, 64x128 for N <= 512, 128x64 for N % 128 != 0), ragged M, every epilogue and both output forms against the plane kernel."""
    import mixermdm_amd as mm
    from mixermdm_amd import ops
    d = torch.device("cuda:0")
    g = torch.Generator().manual_seed(M * 7 + N)
    x, w, b, r = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    xs, ws = ops.split_f32(x.to(d)), ops.split_f32(w.to(d))
    wp = ops.split_pack_weight(ws)
    assert torch.equal(wp.flatten().sort().values, ws.flatten().sort().values)          # a permutation of the same elements
    for epi, extra in [("bias", None), ("gelu", None), ("silu", None), ("resid", r.to(d))]:
        want = ops.linear_split(xs, ws, b.to(d), epi, extra)
        got = ops.linear_split(xs, wp, b.to(d), epi, extra, packed=True)
        assert torch.equal(got, want), (epi, M, N, K)
    assert torch.equal(ops.linear_split(xs, wp, None, "gelu", split_out=True, packed=True), ops.linear_split(xs, ws, None, "gelu", split_out=True))
    # a row slice of the packed matrix that starts on a 32-row boundary is the same offset as in the plane layout (the K|V slice of a packed projection)
    if N >= 128:
        n0 = 64
        lib = mm.load_library()
        import ctypes as C
        out = torch.empty(M, N - n0, device=d)
        rc = lib.mmdm_linear_split_packed(C.c_void_p(xs.data_ptr()), K, M * K, C.c_void_p(wp.data_ptr() + n0 * K * 2), N * K, C.c_void_p(b.to(d)[n0:].contiguous().data_ptr()),
                                          C.c_void_p(out.data_ptr()), N - n0, 0, 0, M, N - n0, K, ops.EPI["bias"], None, 0, 0, None)
        assert rc == 0, lib.mmdm_last_error()
        torch.cuda.synchronize()
        assert torch.equal(out, ops.linear_split(xs, ws, b.to(d))[:, n0:])
    with pytest.raises(Exception):
        ops.linear_split(ops.split_f32(x[:, :48].contiguous().to(d)), ops.split_pack_weight(ops.split_f32(w[:, :48].contiguous().to(d))), packed=True)   # K % 64


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(1196, 1024, 1024), (1196, 2048, 1024), (1196, 3072, 1024), (3588, 1024, 2048), (240, 3072, 1024), (601, 1028, 512), (77, 1024, 96), (6, 512, 2048),
                                   (2400, 516, 1024)])
def test_small_launch_gemm_on_16x16_chains_is_bitwise_the_production_kernel(M, N, K):
    """gemm_s16_kernel (v_mfma_f32_16x16x4_f32: a quarter of a 32 x 32 block's chain; the dispatch of small launches, gemm_f32.hip) accumulates
    every output element in the production kernels' k order: every forced <blocks per wave, stages> form, every epilogue, ragged M and N
    against the 64 x 64-tile launch with the small-launch rules switched off -- and the automatic dispatch against both: s16 alone where the 64 x 64
    grid leaves half the CUs idle, gemm_mix_kernel (whole rounds on 64 x 64 tiles + the remaining rows on 16 x 16 chains, one launch) at M = 1196 / 3588."""
    import mixermdm_amd as mm
    from mixermdm_amd import ops
    lib, d = mm.load_library(), torch.device("cuda:0")
    g = torch.Generator().manual_seed(M * 11 + N)
    x, w, b = torch.randn(M, K, generator=g).to(d), (torch.randn(N, K, generator=g) / K ** 0.5).to(d), torch.randn(N, generator=g).to(d)
    r, pe = torch.randn(M, N, generator=g).to(d), torch.randn(299, N, generator=g).to(d)
    try:
        for epi, extra, period in [("bias", None, 0), ("gelu", None, 0), ("silu", None, 0), ("quickgelu", None, 0), ("sigmoid", None, 0), ("resid", r, 0), ("pe", pe, 299)]:
            lib.mmdm_diag_set(b"gemm_s16", 0)
            want = ops.linear(x, w, b, epi, extra, period)
            assert torch.isfinite(want).all()
            for cfg in (23, 13, 24, 14, -1):
                lib.mmdm_diag_set(b"gemm_s16", cfg)
                got = ops.linear(x, w, b, epi, extra, period, out=torch.full((M, N), float("nan"), device=d))
                assert torch.equal(got, want), (epi, cfg, M, N, K, float((got - want).abs().max()))
        # in place over the residual (the stack's `h += linear(...)`)
        lib.mmdm_diag_set(b"gemm_s16", 0)
        want = ops.linear(x, w, b, "resid", r.clone())
        lib.mmdm_diag_set(b"gemm_s16", -1)
        h = r.clone()
        assert torch.equal(ops.linear(x, w, b, "resid", h, out=h), want)
    finally:
        lib.mmdm_diag_set(b"gemm_s16", -1)
    ref = x.double() @ w.double().T + b.double()
    assert float((ops.linear(x, w, b) .double() - ref).abs().max()) < 2e-5 * (K / 1024) ** 0.5 * 4
