"""Torch-tensor front ends of the stateless HIP kernels (include/mmdm.h section 1).

Every function takes contiguous fp32 CUDA(=HIP) tensors, launches on torch's current stream and returns a new
tensor.  No function here computes anything in Python: a non-CUDA tensor raises.
"""
import ctypes as C
import torch

from ._lib import load_library, check, EncoderLayerWeights

EPI = {"bias": 0, "gelu": 1, "resid": 2, "pe": 3, "silu": 4, "quickgelu": 5, "sigmoid": 6}
ATTN_NO_ZERO_KEY, ATTN_CAUSAL = 1, 2


def _p(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("mixermdm_amd ops run on the GPU only (no CPU fallback): got a CPU tensor")
        if t.dtype != torch.float32 and t.dtype != torch.int32:
            raise TypeError(f"fp32 tensors expected, got {t.dtype}")


def linear(x, weight, bias=None, epilogue="bias", extra=None, period=0, out=None):
    """y = x @ weight.T + bias with a fused epilogue; x [..., K] (last-dim stride 1, uniform row stride)."""
    _chk(x, weight, bias, extra)
    K = x.shape[-1]
    N = weight.shape[0]
    x2 = x.reshape(-1, K) if x.is_contiguous() else x
    assert x2.dim() == 2 and x2.stride(1) == 1 and weight.stride(1) == 1
    M = x2.shape[0]
    if out is None:
        out = torch.empty(M, N, device=x.device, dtype=torch.float32)
    ld_extra = extra.stride(0) if extra is not None else 0
    check(load_library().mmdm_linear_f32(_p(x2), x2.stride(0), _p(weight), weight.stride(0), _p(bias), _p(out), out.stride(0),
                                         M, N, K, EPI[epilogue], _p(extra), ld_extra, period, _stream()))
    return out.reshape(*x.shape[:-1], N) if x.is_contiguous() else out


def adaln(h, ss, ss_rows=None):
    """h [nseq, T, D]; ss [rows, 2D] (scale | shift); row(s) = s % ss_rows."""
    _chk(h, ss)
    nseq, T, D = h.shape
    h = h.contiguous()
    out = torch.empty_like(h)
    check(load_library().mmdm_adaln_f32(_p(h), _p(ss), ss.stride(0), ss_rows or ss.shape[0], _p(out), nseq, T, D, _stream()))
    return out


def attention(q, k, v, num_heads, kv_seq_shift=0, zero_key=True, causal=False):
    """q [nseq, Tq, H*dh], k/v [nseq, Tk, H*dh] (may be column slices of a packed projection); add_zero_attn semantics by default,
    plain softmax with zero_key=False (optionally causal)."""
    _chk(q, k, v)
    nseq, Tq, HD = q.shape
    Tk = k.shape[1]
    for t in (q, k, v):
        assert t.stride(2) == 1 and t.stride(0) == t.shape[1] * t.stride(1), "rows must be uniformly strided"
    out = torch.empty(nseq, Tq, HD, device=q.device, dtype=torch.float32)
    flags = (0 if zero_key else ATTN_NO_ZERO_KEY) | (ATTN_CAUSAL if causal else 0)
    check(load_library().mmdm_attention_opts(_p(q), q.stride(1), _p(k), k.stride(1), _p(v), v.stride(1), _p(out), HD, 0, flags,
                                             nseq, Tq, Tk, num_heads, HD // num_heads, kv_seq_shift, _stream()))
    return out


def layernorm(x, weight, bias, eps=1e-5):
    """nn.LayerNorm over the last dimension with affine parameters."""
    _chk(x, weight, bias)
    x = x.contiguous()
    out = torch.empty_like(x)
    D = x.shape[-1]
    check(load_library().mmdm_layernorm_f32(_p(x), _p(weight), _p(bias), _p(out), x.numel() // D, D, float(eps), _stream()))
    return out


def token_embed(table, tokens, pos):
    """table[tokens] + pos[:L]; tokens int32 [n, L] on the device."""
    _chk(table, tokens, pos)
    assert tokens.dtype == torch.int32 and tokens.is_contiguous() and table.is_contiguous() and pos.is_contiguous()
    n, L = tokens.shape
    D = table.shape[1]
    assert pos.shape[0] >= L and pos.shape[1] == D
    out = torch.empty(n, L, D, device=table.device, dtype=torch.float32)
    check(load_library().mmdm_token_embed_f32(_p(table), table.shape[0], _p(tokens), _p(pos), _p(out), n, L, D, _stream()))
    return out


def gather_rows(src, idx):
    """src [R, D][idx] with idx int32 [n] on the device."""
    _chk(src, idx)
    assert idx.dtype == torch.int32 and src.is_contiguous() and src.dim() == 2
    out = torch.empty(idx.shape[0], src.shape[1], device=src.device, dtype=torch.float32)
    check(load_library().mmdm_gather_rows_f32(_p(src), _p(idx), _p(out), idx.shape[0], src.shape[1], _stream()))
    return out


def encoder_layer_(x, w, num_heads, norm_first=False, activation="gelu", causal=False, eps=1e-5, workspace=None):
    """One nn.TransformerEncoderLayer on x [nseq, T, D] IN PLACE.  w: dict with in_proj_weight, in_proj_bias, out_proj.weight, out_proj.bias,
    linear1.weight/bias, linear2.weight/bias, norm1.weight/bias, norm2.weight/bias (contiguous fp32 device tensors)."""
    names = ("in_proj_weight", "in_proj_bias", "out_proj.weight", "out_proj.bias", "linear1.weight", "linear1.bias",
             "linear2.weight", "linear2.bias", "norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias")
    ts = [w[k] for k in names]
    _chk(x, *ts)
    assert x.is_contiguous() and all(t.is_contiguous() for t in ts)
    nseq, T, D = x.shape
    F = w["linear1.weight"].shape[0]
    lib = load_library()
    need = lib.mmdm_encoder_layer_workspace(nseq, T, D, F)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, device=x.device, dtype=torch.float32)
    cw = EncoderLayerWeights(*[t.data_ptr() for t in ts])
    check(lib.mmdm_encoder_layer_f32(_p(x), C.byref(cw), nseq, T, D, num_heads, F, int(norm_first), EPI[activation], int(causal), float(eps),
                                     _p(workspace), workspace.numel(), _stream()))
    return x


def cond_silu(time_tab, step_idx, txt):
    _chk(time_tab, step_idx, txt)
    out = torch.empty_like(txt)
    check(load_library().mmdm_cond_silu_f32(_p(time_tab), _p(step_idx), _p(txt), _p(out), txt.shape[0], txt.shape[1], _stream()))
    return out


def mixer_pre(o1, o2, stats, align=True):
    _chk(o1, o2, stats)
    n, T, _ = o1.shape
    out1, out2 = torch.empty_like(o1), torch.empty_like(o2)
    check(load_library().mmdm_mixer_pre_f32(_p(o1.contiguous()), _p(o2.contiguous()), _p(stats), _p(out1), _p(out2), n, T, int(align), _stream()))
    return out1, out2


def influence_head(h, weight, bias):
    _chk(h, weight, bias)
    D = h.shape[-1]
    rows = h.numel() // D
    nw = weight.shape[0]
    w = torch.empty(*h.shape[:-1], nw, device=h.device, dtype=torch.float32)
    check(load_library().mmdm_influence_head_f32(_p(h.contiguous()), _p(weight.contiguous()), _p(bias), _p(w), rows, D, nw, _stream()))
    return w


def mean_time(h):
    _chk(h)
    nseq, T, D = h.shape
    out = torch.empty(nseq, D, device=h.device, dtype=torch.float32)
    check(load_library().mmdm_mean_time_f32(_p(h.contiguous()), _p(out), nseq, T, D, _stream()))
    return out


def blend_cfg(out1, out2, w, mode, cfg_scale, force=None, want_hist=False):
    """out1/out2 [2B,T,524]; w [2, 2B, Tw, nw].  Returns model_out [B,T,524] (+ influence_i1, influence_i2, out_influenced)."""
    _chk(out1, out2, w)
    n, T, _ = out1.shape
    B = n // 2
    mo = torch.empty(B, T, 524, device=out1.device, dtype=torch.float32)
    h1 = h2 = hm = None
    if want_hist:
        h1 = torch.empty(n, T, 262, device=out1.device, dtype=torch.float32)
        h2 = torch.empty_like(h1)
        hm = torch.empty_like(out1)
    check(load_library().mmdm_blend_cfg_f32(_p(out1.contiguous()), _p(out2.contiguous()), _p(w.contiguous()), mode, int(force is not None),
                                            float(force or 0.0), float(cfg_scale), _p(mo), _p(h1), _p(h2), _p(hm), B, T, _stream()))
    return (mo, h1, h2, hm) if want_hist else mo


def xstart_ddim(model_out, stats, coef, step_idx, x, x2, align=True):
    """In place on x, x2.  Returns (pred_xstart, pred_xstart2)."""
    _chk(model_out, stats, coef, step_idx, x, x2)
    B, T, _ = model_out.shape
    S = coef.shape[1]
    p1, p2 = torch.empty_like(x), torch.empty_like(x)
    ws = torch.empty(2 * B, device=x.device, dtype=torch.float32)
    check(load_library().mmdm_xstart_ddim_f32(_p(model_out.contiguous()), _p(stats), _p(coef), S, _p(step_idx), _p(x), _p(x2), _p(p1), _p(p2),
                                              _p(ws), B, T, int(align), _stream()))
    return p1, p2


def cfg_ddim(m, coef, step_idx, cfg_scale, x):
    _chk(m, coef, step_idx, x)
    B, T, Cc = x.shape
    p = torch.empty_like(x)
    check(load_library().mmdm_cfg_ddim_f32(_p(m.contiguous()), _p(coef), coef.shape[1], _p(step_idx), float(cfg_scale), _p(x), _p(p), B, T, Cc, _stream()))
    return p


def gaussian_filter1d(x, sigma=1.0, truncate=4.0):
    """scipy.ndimage.gaussian_filter1d(x, sigma, axis=-2, mode="nearest") for x [..., T, C] on the GPU."""
    import numpy as np
    _chk(x)
    x = x.contiguous()
    T, Cc = x.shape[-2], x.shape[-1]
    n = x.numel() // (T * Cc) if x.numel() else 0
    radius = int(truncate * float(sigma) + 0.5)            # scipy: lw = int(truncate * sd + 0.5)
    k = np.arange(-radius, radius + 1)
    w = np.exp(-0.5 / (sigma * sigma) * k ** 2)            # scipy.ndimage._filters._gaussian_kernel1d, order 0
    w = w / w.sum()
    wd = torch.from_numpy(w).to(x.device)
    out = torch.empty_like(x)
    check(load_library().mmdm_gaussian_filter1d_f32(_p(x), _p(out), C.c_void_p(wd.data_ptr()), radius, n, T, Cc, _stream()))
    return out


def to_bf16(x):
    """fp32 -> bf16 (RNE) on the GPU; returns a torch.bfloat16 tensor of the same shape."""
    _chk(x)
    x = x.contiguous()
    out = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16)
    check(load_library().mmdm_f32_to_bf16(_p(x), C.c_void_p(out.data_ptr()), x.numel(), _stream()))
    return out


def pack_weight_frag(w):
    """A static bf16 or fp8 weight [N, K] -> the same bytes in MFMA fragment order (blocks of 32 rows x 32 bytes), for linear_bf16 /
    linear_fp8 (..., packed=True).  N % 32 == 0 and 32 | bytes per row; the result keeps shape and dtype so N and K can be read back."""
    if not w.is_cuda or w.dtype not in (torch.bfloat16, torch.float8_e4m3fn) or w.dim() != 2 or not w.is_contiguous():
        raise TypeError("pack_weight_frag expects a contiguous CUDA bfloat16 / float8_e4m3fn [N, K] tensor")
    N, K = w.shape
    rb = K * w.element_size()
    out = torch.empty_like(w)
    check(load_library().mmdm_pack_weight_frag(C.c_void_p(w.data_ptr()), rb, C.c_void_p(out.data_ptr()), N, rb, _stream()))
    return out


def linear_bf16(x, weight, bias=None, epilogue="bias", extra=None, period=0, out_dtype=torch.float32, packed=False):
    """y = x @ weight.T + bias with bf16 operands (x, weight torch.bfloat16), fp32 accumulation; fp32 or bf16 result.
    packed: weight comes from pack_weight_frag (W straight from global memory; bit-identical results)."""
    for t in (x, weight):
        if not t.is_cuda or t.dtype != torch.bfloat16:
            raise TypeError("linear_bf16 expects CUDA torch.bfloat16 operands")
    _chk(bias, extra)
    K, N = x.shape[-1], weight.shape[0]
    x2 = x.reshape(-1, K)
    M = x2.shape[0]
    out = torch.empty(M, N, device=x.device, dtype=out_dtype)
    if packed:
        check(load_library().mmdm_linear_bf16_packed(C.c_void_p(x2.data_ptr()), x2.stride(0), C.c_void_p(weight.data_ptr()), _p(bias),
                                                     C.c_void_p(out.data_ptr()), out.stride(0), int(out_dtype == torch.bfloat16), M, N, K, EPI[epilogue],
                                                     _p(extra), extra.stride(0) if extra is not None else 0, period, _stream()))
        return out.reshape(*x.shape[:-1], N)
    check(load_library().mmdm_linear_bf16(C.c_void_p(x2.data_ptr()), x2.stride(0), C.c_void_p(weight.data_ptr()), weight.stride(0), _p(bias),
                                          C.c_void_p(out.data_ptr()), out.stride(0), int(out_dtype == torch.bfloat16), M, N, K, EPI[epilogue],
                                          _p(extra), extra.stride(0) if extra is not None else 0, period, _stream()))
    return out.reshape(*x.shape[:-1], N)


SPLIT_SCALE = 2048.0


def split_f32(x):
    """Operand format of the fp32-split mode: [2, *x.shape] torch.float16 with x ~= out[0] + out[1] / 2048 (relative error <= 2^-22;
    out[0] = fp16(x), out[1] = fp16((x - out[0]) * 2048))."""
    _chk(x)
    x = x.contiguous()
    out = torch.empty(2, *x.shape, device=x.device, dtype=torch.float16)
    n = x.numel()
    check(load_library().mmdm_f32_split(_p(x), C.c_void_p(out.data_ptr()), n, n, _stream()))
    return out


def split_pack_weight(ws):
    """Split weights ws [2, N, K] (split_f32) -> the same elements in MFMA fragment order, for linear_split(..., packed=True).
    N % 32 == 0, K % 16 == 0; the result keeps the shape so that N and K can be read back from it."""
    if not ws.is_cuda or ws.dtype != torch.float16 or ws.dim() != 3 or ws.shape[0] != 2 or not ws.is_contiguous():
        raise TypeError("split_pack_weight expects a contiguous CUDA float16 [2, N, K] tensor (ops.split_f32)")
    _, N, K = ws.shape
    out = torch.empty_like(ws)
    check(load_library().mmdm_split_pack_weight(C.c_void_p(ws.data_ptr()), K, N * K, C.c_void_p(out.data_ptr()), N * K, N, K, _stream()))
    return out


def linear_split(xs, ws, bias=None, epilogue="bias", extra=None, period=0, split_out=False, packed=False):
    """fp32 y = x @ w.T + bias computed on the 16-bit matrix cores from two-way fp16 splits xs [2, M, K], ws [2, N, K] (split_f32).
    Returns fp32 [M, N], or its split [2, M, N] when split_out.  packed: ws comes from split_pack_weight (W straight from global memory
    in fragment order; bit-identical results)."""
    for t in (xs, ws):
        if not t.is_cuda or t.dtype != torch.float16 or t.shape[0] != 2 or not t.is_contiguous():
            raise TypeError("linear_split expects contiguous CUDA float16 [2, rows, K] operands (ops.split_f32)")
    _chk(bias, extra)
    _, M, K = xs.shape
    N = ws.shape[1]
    out = torch.empty((2, M, N) if split_out else (M, N), device=xs.device, dtype=torch.float16 if split_out else torch.float32)
    if packed:
        check(load_library().mmdm_linear_split_packed(C.c_void_p(xs.data_ptr()), K, M * K, C.c_void_p(ws.data_ptr()), N * K, _p(bias), C.c_void_p(out.data_ptr()), N,
                                                      M * N, int(split_out), M, N, K, EPI[epilogue], _p(extra), extra.stride(0) if extra is not None else 0, period, _stream()))
        return out
    check(load_library().mmdm_linear_split(C.c_void_p(xs.data_ptr()), K, M * K, C.c_void_p(ws.data_ptr()), K, N * K, _p(bias), C.c_void_p(out.data_ptr()), N,
                                           M * N, int(split_out), M, N, K, EPI[epilogue], _p(extra), extra.stride(0) if extra is not None else 0, period, _stream()))
    return out


def bf16_split3(x):
    """Exact 3-way bf16 split of an fp32 tensor (torch arithmetic): [3, *x.shape] torch.bfloat16 with x == out[0] + out[1] + out[2]; the
    operand format of attention_planes(NP = 3) (the split-mode GEMMs write it as their optional second output)."""
    x1 = x.bfloat16()
    r1 = x - x1.float()
    x2 = r1.bfloat16()
    return torch.stack([x1, x2, (r1 - x2.float()).bfloat16()])


def attention_planes(qp, kp, v, num_heads, kv_seq_shift=0, zero_key=True, causal=False):
    """Attention with Q K^T on the bf16 matrix cores.  qp [NP, nseq, Tq, H*dh], kp [NP, nseq, Tk, H*dh] torch.bfloat16 (NP = 3: exact three-way
    bf16 splits x = x1 + x2 + x3, see bf16_split3 -> fp32-accurate scores; NP = 1: bf16), v [nseq, Tk, H*dh] fp32 (may be a column slice).  Returns fp32 [nseq, Tq, H*dh]."""
    _chk(v)
    NP, nseq, Tq, HD = qp.shape
    Tk = kp.shape[2]
    assert qp.dtype == torch.bfloat16 and kp.dtype == torch.bfloat16 and qp.is_cuda and kp.is_cuda and kp.shape[0] == NP
    assert qp.stride(3) == 1 and kp.stride(3) == 1 and v.stride(2) == 1
    out = torch.empty(nseq, Tq, HD, device=v.device, dtype=torch.float32)
    flags = (0 if zero_key else ATTN_NO_ZERO_KEY) | (ATTN_CAUSAL if causal else 0)
    check(load_library().mmdm_attention_planes(C.c_void_p(qp.data_ptr()), qp.stride(2), qp.stride(0), C.c_void_p(kp.data_ptr()), kp.stride(2), kp.stride(0), NP,
                                               _p(v), v.stride(1), _p(out), HD, 0, flags, nseq, Tq, Tk, num_heads, HD // num_heads, kv_seq_shift, _stream()))
    return out


def attention_bf16(qb, kb, vb, num_heads, kv_seq_shift=0, zero_key=True, causal=False, out_dtype=torch.float32):
    """All-bf16 attention (BASELINE configs[4] path): qb [nseq, Tq, H*dh], kb / vb [nseq, Tk, H*dh] torch.bfloat16 (may be column slices of a
    packed projection copy); Q K^T and P.V on the bf16 matrix cores, fp32 softmax and accumulation.  Returns fp32 or bf16 [nseq, Tq, H*dh]."""
    for t in (qb, kb, vb):
        assert t.is_cuda and t.dtype == torch.bfloat16 and t.stride(2) == 1 and t.stride(0) == t.shape[1] * t.stride(1)
    nseq, Tq, HD = qb.shape
    Tk = kb.shape[1]
    out = torch.empty(nseq, Tq, HD, device=qb.device, dtype=out_dtype)
    flags = (0 if zero_key else ATTN_NO_ZERO_KEY) | (ATTN_CAUSAL if causal else 0)
    check(load_library().mmdm_attention_bf16(C.c_void_p(qb.data_ptr()), qb.stride(1), C.c_void_p(kb.data_ptr()), kb.stride(1), C.c_void_p(vb.data_ptr()), vb.stride(1),
                                             C.c_void_p(out.data_ptr()), HD, int(out_dtype == torch.bfloat16), flags, nseq, Tq, Tk, num_heads, HD // num_heads,
                                             kv_seq_shift, _stream()))
    return out


def attention_split(qs, ks, vs, num_heads, kv_seq_shift=0, zero_key=True, causal=False, split_out=False):
    """Attention of the fp32-split mode: qs [2, nseq, Tq, H*dh], ks / vs [2, nseq, Tk, H*dh] torch.float16 planes (split_f32; may be column
    slices of a packed projection written with split_out); scores and P.V from three fp16 MFMAs per block with hi / lo accumulators, fp32 softmax.
    Returns fp32 [nseq, Tq, H*dh] or its two planes."""
    for t in (qs, ks, vs):
        assert t.is_cuda and t.dtype == torch.float16 and t.shape[0] == 2 and t.stride(3) == 1 and t.stride(1) == t.shape[2] * t.stride(2)
    _, nseq, Tq, HD = qs.shape
    Tk = ks.shape[2]
    out = torch.empty((2, nseq, Tq, HD) if split_out else (nseq, Tq, HD), device=qs.device, dtype=torch.float16 if split_out else torch.float32)
    flags = (0 if zero_key else ATTN_NO_ZERO_KEY) | (ATTN_CAUSAL if causal else 0)
    check(load_library().mmdm_attention_split(C.c_void_p(qs.data_ptr()), qs.stride(2), qs.stride(0), C.c_void_p(ks.data_ptr()), ks.stride(2), ks.stride(0),
                                              C.c_void_p(vs.data_ptr()), vs.stride(2), vs.stride(0), C.c_void_p(out.data_ptr()), HD, 2 if split_out else 0, flags,
                                              nseq, Tq, Tk, num_heads, HD // num_heads, kv_seq_shift, _stream()))
    return out


def quantize_rows_fp8(x):
    """Row-wise OCP e4m3 quantisation of an fp32 [rows, K] tensor: returns (q uint8-viewed-as torch.float8_e4m3fn [rows, K], scale fp32 [rows])
    with x ~= q.float() * scale[:, None].  Per-output-channel weight quantisation is this on W [N, K]."""
    _chk(x)
    x2 = x.reshape(-1, x.shape[-1]).contiguous()
    rows, K = x2.shape
    q = torch.empty(rows, K, device=x.device, dtype=torch.uint8)
    scale = torch.empty(rows, device=x.device, dtype=torch.float32)
    check(load_library().mmdm_quantize_rows_fp8(_p(x2), x2.stride(0), C.c_void_p(q.data_ptr()), q.stride(0), _p(scale), rows, K, _stream()))
    return q.view(torch.float8_e4m3fn), scale


def linear_fp8(xq, x_scale, wq, w_scale, bias=None, epilogue="bias", extra=None, period=0, out_dtype=torch.float32, packed=False):
    """y = (xq * x_scale[:, None]) @ (wq * w_scale[:, None]).T + bias on the fp8 matrix instructions (fp32 accumulation, de-quantised in the
    epilogue).  xq [M, K], wq [N, K] torch.float8_e4m3fn; x_scale [M] / w_scale [N] fp32 or None; out fp32, bf16 or float8_e4m3fn (unit scale)."""
    for t in (xq, wq):
        if not t.is_cuda or t.dtype != torch.float8_e4m3fn:
            raise TypeError("linear_fp8 expects CUDA torch.float8_e4m3fn operands")
    _chk(bias, extra, x_scale, w_scale)
    K, N = xq.shape[-1], wq.shape[0]
    x2 = xq.reshape(-1, K)
    M = x2.shape[0]
    out = torch.empty(M, N, device=xq.device, dtype=out_dtype)
    mode = {torch.float32: 0, torch.bfloat16: 1, torch.float8_e4m3fn: 2}[out_dtype]
    if packed:          # wq from pack_weight_frag
        check(load_library().mmdm_linear_fp8_packed(C.c_void_p(x2.data_ptr()), x2.stride(0), _p(x_scale), C.c_void_p(wq.data_ptr()), _p(w_scale), _p(bias),
                                                    C.c_void_p(out.data_ptr()), out.stride(0), mode, M, N, K, EPI[epilogue],
                                                    _p(extra), extra.stride(0) if extra is not None else 0, period, _stream()))
        return out.reshape(*xq.shape[:-1], N)
    check(load_library().mmdm_linear_fp8(C.c_void_p(x2.data_ptr()), x2.stride(0), _p(x_scale), C.c_void_p(wq.data_ptr()), wq.stride(0), _p(w_scale), _p(bias),
                                         C.c_void_p(out.data_ptr()), out.stride(0), mode, M, N, K, EPI[epilogue],
                                         _p(extra), extra.stride(0) if extra is not None else 0, period, _stream()))
    return out.reshape(*xq.shape[:-1], N)


def adaln_fp8(h, ss, ss_rows=None):
    """AdaLN apply with an fp8 result: (q float8_e4m3fn [nseq, T, D], row_scale fp32 [nseq*T])."""
    _chk(h, ss)
    nseq, T, D = h.shape
    h = h.contiguous()
    q = torch.empty(nseq, T, D, device=h.device, dtype=torch.uint8)
    scale = torch.empty(nseq * T, device=h.device, dtype=torch.float32)
    check(load_library().mmdm_adaln_fp8(_p(h), _p(ss), ss.stride(0), ss_rows or ss.shape[0], C.c_void_p(q.data_ptr()), _p(scale), nseq, T, D, _stream()))
    return q.view(torch.float8_e4m3fn), scale
