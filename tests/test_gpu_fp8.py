"""GPU: BASELINE configs[4] -- the fp8 (OCP e4m3) GEMM operands of the bf16 path.  Kernel-level checks are exact statements about the
quantised values (the fp8 bytes are decoded on the host and the float64 product of the DECODED operands is the reference); the
sampler-level check states the accuracy of the whole precision mode against the CPU oracle at the real model sizes."""
import math
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import mixer as MX            # noqa: E402  (checker only)
from oracle import schedule as OS         # noqa: E402
from test_gpu_kernels import assert_close, rnd, dev   # noqa: E402


def test_row_quantiser_is_rne_e4m3_with_absmax_scale():
    from mixermdm_amd import ops
    x = rnd(1, 300, 1024) * torch.logspace(-3, 2, 300)[:, None]
    x[7] = 0
    q, s = ops.quantize_rows_fp8(x.to(dev()))
    want_s = x.abs().amax(dim=1) / 448
    want_s[7] = 1.0
    assert_close(s, want_s, atol=0, rtol=1e-6, what="row scales")
    ref = (x / s.cpu()[:, None]).to(torch.float8_e4m3fn)             # torch's conversion: round to nearest even
    same = (q.cpu().view(torch.uint8) == ref.view(torch.uint8))
    assert same.float().mean().item() > 0.9999, same.float().mean().item()      # ties in x / s can differ by the fp32 division's last bit
    assert (q.cpu().float() - ref.float()).abs().max().item() <= 32        # and never by more than one e4m3 step
    assert bool(torch.isfinite(q.cpu().float()).all()) and q.cpu().float().abs().max().item() == 448


@pytest.mark.parametrize("M,N,K,epi", [(300, 512, 1024, "bias"), (19200, 3072, 1024, "bias"), (19200, 2048, 1024, "gelu"), (19200, 1024, 2048, "resid"),
                                       (130, 136, 64, "bias"), (1200, 1024, 512, "resid")])
def test_linear_fp8_vs_float64_of_the_decoded_operands(M, N, K, epi):
    from mixermdm_amd import ops
    from mixermdm_amd._lib import load_library
    d = dev()
    x, w, b = rnd(11, M, K), rnd(12, N, K, scale=1 / math.sqrt(K)), rnd(13, N)
    r = rnd(14, M, N) if epi == "resid" else None
    xq, xs = ops.quantize_rows_fp8(x.to(d))
    wq, ws = ops.quantize_rows_fp8(w.to(d))
    xd, wd = xq.cpu().float().double() * xs.cpu().double()[:, None], wq.cpu().float().double() * ws.cpu().double()[:, None]
    ref = F.linear(xd, wd, b.double())
    if epi == "gelu":
        ref = F.gelu(ref)
    if epi == "resid":
        ref = ref + r.double()
    got = ops.linear_fp8(xq, xs, wq, ws, b.to(d), epi, r.to(d) if r is not None else None)
    assert load_library().mmdm_last_gemm_kernel().decode().startswith("gemm_fp8<")
    # fp32 accumulation of K products of magnitude up to 448^2 whose sum cancels to O(1) after de-quantisation: 1e-4 of the output scale
    assert_close(got, ref.float(), atol=2e-4, rtol=1e-4, what=f"linear_fp8 {M}x{N}x{K} {epi}")
    # and the quantisation itself costs what e4m3 costs: a few percent of the output's scale against the un-quantised product
    full = F.linear(x.double(), w.double(), b.double())
    if epi == "bias":
        rel = ((got.cpu().double() - full).pow(2).mean().sqrt() / full.pow(2).mean().sqrt()).item()
        assert rel < 0.06, rel
    if epi == "gelu":         # fp8 output at unit scale = e4m3 rounding of the fp32 output
        g8 = ops.linear_fp8(xq, xs, wq, ws, b.to(d), epi, out_dtype=torch.float8_e4m3fn)
        want = got.clamp(-448, 448).to(torch.float8_e4m3fn)
        assert (g8.cpu().view(torch.uint8) == want.cpu().view(torch.uint8)).float().mean().item() > 0.9999
        gb = ops.linear_fp8(xq, xs, wq, ws, b.to(d), epi, out_dtype=torch.bfloat16)
        assert torch.equal(gb, got.bfloat16())


def test_linear_fp8_unit_scales_and_errors():
    from mixermdm_amd import ops, MMDMError
    d = dev()
    x = (torch.randint(-8, 9, (64, 128)).float() / 4).to(d)          # exactly representable in e4m3
    w = (torch.randint(-8, 9, (96, 128)).float() / 4).to(d)
    xq, wq = x.to(torch.float8_e4m3fn), w.to(torch.float8_e4m3fn)
    got = ops.linear_fp8(xq, None, wq, None)
    assert torch.equal(got, x @ w.t())                                  # small integers / 16: exact in fp32
    with pytest.raises(MMDMError, match="K %% 64|K % 64"):
        ops.linear_fp8(xq[:, :96].contiguous(), None, wq[:, :96].contiguous(), None)


@pytest.mark.parametrize("D", [512, 1024])
def test_adaln_fp8_is_the_row_quantised_adaln(D):
    from mixermdm_amd import ops
    nseq, T, rows = 6, 50, 3
    h, ss = rnd(20, nseq, T, D) * 3 + 0.5, rnd(21, rows, 2 * D)
    y = ops.adaln(h.to(dev()), ss.to(dev()), rows)
    q, s = ops.adaln_fp8(h.to(dev()), ss.to(dev()), rows)
    q2, s2 = ops.quantize_rows_fp8(y.reshape(-1, D))
    assert torch.equal(s, s2)
    assert (q.view(torch.uint8).reshape(-1, D) == q2.view(torch.uint8)).float().mean().item() > 0.9999
    deq = q.float().reshape(-1, D) * s[:, None]
    err = (deq - y.reshape(-1, D)).abs()
    assert (err <= y.reshape(-1, D).abs() * 2 ** -4 + s[:, None] * 2 ** -9 + 1e-12).all()      # half an e4m3 step (3 mantissa bits), sub-normal floor


# ---------------------------------------------------------------------------------------------------
# the precision mode as a whole, against the CPU oracle, at the real model sizes and the headline length
# ---------------------------------------------------------------------------------------------------
def _rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()


def test_bf16_and_fp8_modes_vs_oracle_at_full_dims_T300():
    """BASELINE configs[4]: precision "bf16" and "bf16_fp8" (fp8 e4m3 QKV / cross-attention input / FFN operands) at D=1024/512, T=300,
    B=2, against the ORACLE (fp32 CPU restatement pinned to the reference).  Stated tolerances, relative RMS over the whole tensor (the
    per-element STEP_TOL of the fp32 modes does not apply to 8- and 3-bit mantissas), set at ~1.5x what these random-weight networks
    measure:
      * Mixer.forward rows (the network evaluation itself, no guidance): bf16 2e-2 (measured 1.1e-2), fp8 1.5e-1 (measured 9.6e-2: every fp8
        GEMM output carries ~5 % of quantisation noise, 2^-4 / sqrt(3) per operand element on both operands, through 8 + 8 + 4 blocks);
      * one ddim1000 step from x_T (i = 999) and one mid-schedule step (i = 500): the DDIM chains within 1e-3 (bf16) / 3e-3 and 5e-3 (fp8);
        pred_xstart2 = the guided output 3.5 c - 2.5 u: bf16 3e-2 (measured 1.0e-2); fp8 only bounded by 1.0 -- measured 3.2e-1 / 5.4e-1:
        the guidance combine multiplies un-correlated evaluation noise by sqrt(3.5^2 + 2.5^2) = 4.3, which is what an 8-bit evaluation of a
        3.5-guided sampler costs.  An accuracy STATEMENT for configs[4], not an equivalence claim: the parity path is fp32."""
    from mixermdm_amd.sampler import Sampler
    from mixermdm_amd.synthetic import synthetic_state_dict, synthetic_stats, synthetic_inputs, FULL_DIMS
    from oracle.layers import pe_table
    sd = synthetic_state_dict(seed=0, std=0.02, bias_std=0.02, **FULL_DIMS)
    st = synthetic_stats()
    stats = (st["mean_hml"], st["std_hml"], st["mean_ih"], st["std_ih"])
    W = dict(sd)
    W["sequence_pos_encoder.pe"], W["denoiser1.sequence_pos_encoder.pe"], W["denoiser2.sequence_pos_encoder.pe"] = pe_table(512), pe_table(1024), pe_table(1024)
    B, T = 2, 300
    cond, xT = synthetic_inputs(B, T, seed_cond=51, seed_x=52)
    x2 = rnd(53, B, T, 524)
    sch = OS.make_schedule("cosine", 1000, "ddim1000")
    spec = MX.MixerSpec(d_heads=8, m_heads=8)
    n = 2 * B
    fx1, fx2, fc = rnd(54, n, T, 524), rnd(55, n, T, 524), rnd(56, n, 8 * 768)
    fc[B:] = 0
    nthr = torch.get_num_threads()
    torch.set_num_threads(min(16, nthr))
    with torch.no_grad():
        ref = {999: MX.mixer_ddim_step(W, spec, stats, sch, 3.5, 999, xT, xT, cond), 500: MX.mixer_ddim_step(W, spec, stats, sch, 3.5, 500, xT, x2, cond)}
        fwd = MX.mixer_forward(W, spec, stats, fx1, torch.full((n,), 640, dtype=torch.long), fc, fx2)
    torch.set_num_threads(nthr)
    # stated accuracy, not parity: every bound is ~1.5 x what the mode measures (bf16_fp8 guided output: 0.32-0.54 -- CFG 3.5 multiplies the
    # uncorrelated evaluation noise of the cond / uncond rows by 4.3)
    lim = {"bf16": {"fwd": 2e-2, "out": 3e-2, 999: 1e-3, 500: 1e-3}, "bf16_fp8": {"fwd": 1.5e-1, "out": 0.8, 999: 3e-3, 500: 5e-3}}
    seen = {}
    for mode in ("bf16", "bf16_fp8"):
        s = Sampler(d_heads=8, m_heads=8, max_batch=B, max_frames=T, precision=mode, **FULL_DIMS)
        s.load_state_dict(sd)
        s.set_norm_stats(*stats)
        s.prepare()
        e_fwd = _rel(s.module_forward(2, fx1, fc, 640, x2=fx2), fwd)
        print(f"{mode}: Mixer.forward rows rel RMS vs oracle {e_fwd:.3e}")
        assert e_fwd <= lim[mode]["fwd"], (mode, e_fwd)
        s.set_schedule("ddim1000")
        s.begin(cond, xT)
        for i, (a, b) in ((999, (xT, xT)), (500, (xT, x2))):
            stt = s.state()
            stt["x"].copy_(a.to(stt["x"].device)); stt["x2"].copy_(b.to(stt["x2"].device)); torch.cuda.synchronize()
            s.seek(i)
            s.run(1, use_graph=True)
            stt = s.state()
            rx, rx2, p1, p2 = ref[i]
            assert all(torch.isfinite(stt[k]).all() for k in ("x", "x2", "pred_xstart2"))
            e_out, e_x, e_x2 = _rel(stt["pred_xstart2"], p2), _rel(stt["x"], rx), _rel(stt["x2"], rx2)
            print(f"{mode} i={i}: rel RMS vs oracle: pred_xstart2 {e_out:.3e}, x {e_x:.3e}, x2 {e_x2:.3e}")
            assert e_out <= lim[mode]["out"] and e_x <= lim[mode][i] and e_x2 <= lim[mode][i], (mode, i, e_out, e_x, e_x2)
            seen[(mode, i)] = stt["pred_xstart2"].clone()
        # a short free-running loop stays finite under graph replay
        s.begin(cond, xT)
        s.run(5, use_graph=True)
        assert torch.isfinite(s.state()["x"]).all()
        s.close()
    assert not torch.equal(seen[("bf16", 999)], seen[("bf16_fp8", 999)])          # the fp8 kernels really ran


@pytest.mark.parametrize("nseq,Tq,Tk,H,dh,zero_key,causal", [(3, 70, 70, 2, 128, True, False), (2, 300, 300, 8, 128, True, False), (2, 33, 50, 4, 64, True, False),
                                                             (2, 64, 64, 4, 64, False, True), (1, 1, 1, 1, 128, True, False)])
def test_attention_bf16_vs_float64_of_the_rounded_operands(nseq, Tq, Tk, H, dh, zero_key, causal):
    """mmdm_attention_bf16 (configs[4] path): Q K^T and P.V on the bf16 matrix cores (V read transposed from its row-major LDS image by
    ds_read_b64_tr_b16), fp32 softmax / accumulation.  Reference: float64 attention of the SAME bf16-rounded Q, K, V, so the only error left
    is the rounding of the probabilities to bf16: each carries at most 2^-9 relative error, so |out - ref| <= 2^-9 * sum_k P_k |V_k| / l <=
    2^-9 * max|V| for every element -- the bound asserted here (measured: 0.5-1.8 % of the output's RMS at the worst element, which is a row
    dominated by one key; the deferred softmax reference of the kernel means that key's probability is no longer exactly 1).  The fp32 P.V
    form of the same kernel must stay within 4e-6."""
    from mixermdm_amd import ops
    d = dev()
    D = H * dh
    qb, kb, vb = (rnd(90 + i, nseq, T_, D).to(d).bfloat16() for i, T_ in enumerate((Tq, Tk, Tk)))
    Q = qb.double().view(nseq, Tq, H, dh).transpose(1, 2)
    K = kb.double().view(nseq, Tk, H, dh).transpose(1, 2)
    V = vb.double().view(nseq, Tk, H, dh).transpose(1, 2)
    S = Q @ K.transpose(-1, -2) / math.sqrt(dh)
    if causal:
        S = S.masked_fill(torch.triu(torch.ones(Tq, Tk, device=d, dtype=torch.bool), 1), float("-inf"))
    if zero_key:
        S = torch.cat([S, torch.zeros(nseq, H, Tq, 1, device=d, dtype=torch.float64)], -1)
        V = torch.cat([V, torch.zeros(nseq, H, 1, dh, device=d, dtype=torch.float64)], 2)
    ref = (S.softmax(-1) @ V).transpose(1, 2).reshape(nseq, Tq, D)
    rms = ref.pow(2).mean().sqrt().item()
    got = ops.attention_bf16(qb, kb, vb, H, zero_key=zero_key, causal=causal)
    assert torch.isfinite(got).all()
    err = (got.double() - ref).abs().max().item()
    print(f"attention_bf16 {nseq}x{Tq}x{Tk} H={H} dh={dh}: max err {err:.2e} = {err / max(rms, 1e-9) * 100:.2f} % of the output RMS; bound {2 ** -9 * vb.float().abs().max().item():.2e}")
    assert err <= 2.0 ** -9 * vb.float().abs().max().item() * 1.05 + 1e-5, (err, rms)
    exact = ops.attention_planes(qb[None], kb[None], vb.float(), H, zero_key=zero_key, causal=causal)
    assert (exact.double() - ref).abs().max().item() <= 4e-6
    gb = ops.attention_bf16(qb, kb, vb, H, zero_key=zero_key, causal=causal, out_dtype=torch.bfloat16)
    assert torch.equal(gb, got.bfloat16())


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(300, 256, 256), (19, 512, 512), (1000, 768, 1024), (4800, 1024, 2048), (520, 2048, 512), (130, 1152, 256)])
def test_packed_bf16_and_fp8_linear_are_bitwise_the_plane_kernels(M, N, K):
    """Weights in fragment order (mmdm_pack_weight_frag), W straight from global memory (gemm_bf16w_kernel): every epilogue and output form,
    ragged M, against the LDS-staged kernels -- the accumulators start the same way and k ascends the same way."""
    import mixermdm_amd as mm
    from mixermdm_amd import ops
    d = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5 * M + N)
    x, w, b, r = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    xb, wb = ops.to_bf16(x.to(d)), ops.to_bf16(w.to(d))
    wp = ops.pack_weight_frag(wb)
    assert torch.equal(wp.view(torch.int16).flatten().sort().values, wb.view(torch.int16).flatten().sort().values)
    lib = mm.load_library()
    for epi, extra in [("bias", None), ("gelu", None), ("silu", None), ("resid", r.to(d))]:
        for od in (torch.float32, torch.bfloat16):
            want = ops.linear_bf16(xb, wb, b.to(d), epi, extra, out_dtype=od)
            got = ops.linear_bf16(xb, wp, b.to(d), epi, extra, out_dtype=od, packed=True)
            assert lib.mmdm_last_gemm_kernel().decode() == ("gemm_bf16w<14,42>" if N > 1024 and N % 256 == 0 else "gemm_bf16w<14,41>")
            assert torch.equal(got.view(torch.int16 if od == torch.bfloat16 else torch.int32), want.view(torch.int16 if od == torch.bfloat16 else torch.int32)), (epi, od)
    xq, xs = ops.quantize_rows_fp8(x.to(d))
    wq, ws = ops.quantize_rows_fp8(w.to(d))
    wqp = ops.pack_weight_frag(wq)
    for epi, extra in [("bias", None), ("gelu", None), ("resid", r.to(d))]:
        for od in (torch.float32, torch.bfloat16, torch.float8_e4m3fn):
            want = ops.linear_fp8(xq, xs, wq, ws, b.to(d), epi, extra, out_dtype=od)
            got = ops.linear_fp8(xq, xs, wqp, ws, b.to(d), epi, extra, out_dtype=od, packed=True)
            assert lib.mmdm_last_gemm_kernel().decode() == "gemm_fp8w<14,41>"
            it = {torch.float32: torch.int32, torch.bfloat16: torch.int16, torch.float8_e4m3fn: torch.int8}[od]
            assert torch.equal(got.view(it), want.view(it)), (epi, od)
            if N % 256 == 0:            # the 128 x 256 form the large shards take (W fragments requested half a step ahead): same bits
                lib.mmdm_diag_set(b"bf16_cfg", 12)
                try:
                    wide = ops.linear_fp8(xq, xs, wqp, ws, b.to(d), epi, extra, out_dtype=od, packed=True)
                    assert lib.mmdm_last_gemm_kernel().decode() == "gemm_fp8w<14,42>"
                finally:
                    lib.mmdm_diag_set(b"bf16_cfg", -1)
                assert torch.equal(wide.view(it), want.view(it)), (epi, od, "128x256")
    with pytest.raises(Exception):
        ops.linear_bf16(xb, ops.pack_weight_frag(wb[:96].contiguous()), packed=True)       # N % 128


# ---------------------------------------------------------------------------------------------------
# round 6: the persistent fp8 GEMM (gemm_fp8p.hip) -- a tile's epilogue under the next tile's K loop -- is bitwise the packed kernel
# ---------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K,epi,od", [(19200, 3072, 1024, "bias", torch.bfloat16), (19200, 1024, 1024, "bias", torch.bfloat16),
                                           (19200, 2048, 1024, "gelu", torch.float8_e4m3fn), (8192, 1024, 1024, "bias", torch.float8_e4m3fn),
                                           (9344, 1152, 1024, "gelu", torch.bfloat16), (76800, 2048, 1024, "bias", torch.bfloat16)])
def test_persistent_fp8_linear_is_bitwise_the_packed_kernel(M, N, K, epi, od):
    """Same products in the same k order, same de-quantisation expression, activation and conversion: the persistent kernel's output equals
    gemm_bf16w_kernel<1, .>'s bit for bit -- at the step's shapes (M = 19 200: 3600 / 1200 / 2400 tiles over 512 workgroups, i.e. uneven tile
    counts per workgroup), at the smallest grid it takes (512 tiles: one tile per workgroup, nothing overlapped), at a tile count that is no
    multiple of 8 (73 x 9 = 657: XCD ranges of unequal length) and at configs[4]'s 64-motion shard (M = 76 800).  Repeated launches are stable
    (no stale LDS image / scale rows between tiles), also with a second launch of another shape in between."""
    from mixermdm_amd._lib import load_library, diag
    from mixermdm_amd import ops
    lib = load_library()
    d = torch.device("cuda:0")
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, K, generator=g).to(d)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(d)
    b = torch.randn(N, generator=g).to(d)
    xq, xs = ops.quantize_rows_fp8(x)
    wq, ws = ops.quantize_rows_fp8(w)
    wp = ops.pack_weight_frag(wq)
    diag("fp8p", 0)
    ref = ops.linear_fp8(xq, xs, wp, ws, b, epi, None, out_dtype=od, packed=True)
    assert "fp8w" in lib.mmdm_last_gemm_kernel().decode()
    try:
        diag("fp8p", 2)
        for rep in range(3):
            got = ops.linear_fp8(xq, xs, wp, ws, b, epi, None, out_dtype=od, packed=True)
            assert lib.mmdm_last_gemm_kernel().decode().startswith("gemm_fp8p<"), lib.mmdm_last_gemm_kernel()
            assert torch.equal(got.view(torch.uint8), ref.view(torch.uint8)), (rep, (got.float() - ref.float()).abs().max().item())
            ops.linear_fp8(xq[:8192], xs[:8192], wp, ws, b, epi, None, out_dtype=od, packed=True)         # another tile walk in between
        # a_const instead of per-row scales (the FFN's GELU tensor): same check
        r2 = None
        for on in (0, 2):
            diag("fp8p", on)
            o = torch.empty(M, N, device=d, dtype=od)
            import ctypes as C
            from mixermdm_amd._lib import check
            check(lib.mmdm_linear_fp8_packed(C.c_void_p(xq.data_ptr()), K, C.c_void_p(0), C.c_void_p(wp.data_ptr()), C.c_void_p(ws.data_ptr()), C.c_void_p(b.data_ptr()),
                                             C.c_void_p(o.data_ptr()), N, {torch.bfloat16: 1, torch.float8_e4m3fn: 2}[od], M, N, K, ops.EPI[epi], C.c_void_p(0), 0, 0,
                                             C.c_void_p(torch.cuda.current_stream().cuda_stream)))
            if r2 is None:
                r2 = o
            else:
                assert torch.equal(o.view(torch.uint8), r2.view(torch.uint8))
    finally:
        diag("fp8p", 0)
