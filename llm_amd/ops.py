"""Host-side mirror of the reference's operator interface for the hot path, on torch device tensors.

Same names and argument meaning as the ggml ops the backend serves (ggml/include/ggml.h):
  mul_mat(a, b)            ggml_mul_mat      a = weights [M rows of K] (block-quantized bytes), b = [N, K] f32 -> [N, M] f32
  mul_mat_id(as_, b, ids)  ggml_mul_mat_id
plus the two contract-level primitives quantize_q8_1 / dequantize.  Everything runs through the C ABI
of libmi355_kernels.so on the current torch stream; torch only owns the memory.
"""
import torch

from . import blocks as B
from . import kernels as K


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _u8(t):
    assert t.is_cuda and t.dtype == torch.uint8 and t.is_contiguous()
    return t


def _dst_rows(dst):
    return dst.stride(0) if dst.dim() > 1 else dst.numel()


def quantize_q8_1(x):
    """x: [..., K] f32 (last dim contiguous, up to 4-D) -> uint8 [rows, K/32*36] of block_q8_1."""
    assert x.is_cuda and x.dtype == torch.float32 and x.stride(-1) == 1 and 1 <= x.dim() <= 4
    k = x.shape[-1]
    shp = [1] * (4 - x.dim()) + list(x.shape)
    st = [0] * (4 - x.dim()) + list(x.stride())
    ne3, ne2, ne1 = shp[0], shp[1], shp[2]
    nb3, nb2, nb1 = (s * 4 for s in st[:3])
    rows = ne1 * ne2 * ne3
    y = torch.empty((rows, k // 32 * 36), dtype=torch.uint8, device=x.device)
    K.check(K.lib.mi355_quantize_row_q8_1(x.data_ptr(), y.data_ptr(), k, ne1, ne2, ne3, nb1, nb2, nb3, _stream()),
            "mi355_quantize_row_q8_1")
    return y


def quantize_act(x):
    """x: [rows, K] f32 contiguous -> (act_q int8 [rows, K], act_ds f32 [rows, K/32, 2])."""
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.is_contiguous()
    rows, k = x.shape
    aq = torch.empty((rows, k), dtype=torch.int8, device=x.device)
    ads = torch.empty((rows, k // 32, 2), dtype=torch.float32, device=x.device)
    K.check(K.lib.mi355_quantize_act(x.data_ptr(), aq.data_ptr(), ads.data_ptr(), k, rows, 1, 1, k * 4, 0, 0, _stream()),
            "mi355_quantize_act")
    return aq, ads


def dequantize(t, blocks, n):
    """blocks: uint8 device tensor holding n elements of type t -> f32 [n]."""
    _u8(blocks)
    assert blocks.numel() >= B.row_size(t, n)
    y = torch.empty(n, dtype=torch.float32, device=blocks.device)
    K.check(K.lib.mi355_dequantize_row(t, blocks.data_ptr(), y.data_ptr(), n, _stream()), "mi355_dequantize_row")
    return y


def mul_mat_vec_q(t, w, m, k, aq, ads, ncols, dst=None, w_row_stride=None, residual=None, sum_out=None):
    """Decode mat-vec on pre-quantized activations (act rows 0..ncols-1).  residual ([ncols, m] f32): also store
    sum_out = result + residual (sum_out may be dst -> only the sum is stored, or residual itself)."""
    _u8(w)
    if dst is None:
        dst = torch.empty((ncols, m), dtype=torch.float32, device=w.device)
    a = K.MmvqArgs()
    a.type, a.W, a.w_row_stride, a.w_ch_stride = t, w.data_ptr(), w_row_stride or B.row_size(t, k), 0
    a.act_q, a.act_ds, a.act_ch_stride = aq.data_ptr(), ads.data_ptr(), 0
    a.dst, a.dst_col_stride, a.dst_ch_stride = dst.data_ptr(), dst.stride(0), 0
    a.m, a.k, a.ncols, a.nchannels, a.ch_ratio = m, k, ncols, 1, 1
    if residual is not None:
        a.residual, a.sum = residual.data_ptr(), sum_out.data_ptr()
    K.check(K.lib.mi355_mul_mat_vec_q(a, _stream()), "mi355_mul_mat_vec_q")
    return dst


def mul_mat_vec_q_group(t, ws, ms, k, aq, ads, ncols):
    """Grouped decode mat-vec: several weight matrices (same type and k) on the same pre-quantized activations in
    ONE launch (mi355_mul_mat_vec_q_group).  ws: list of uint8 block tensors, ms: rows of each -> list of [ncols, m]."""
    outs, arr = [], (K.MmvqArgs * len(ws))()
    for i, (w, m) in enumerate(zip(ws, ms)):
        _u8(w)
        dst = torch.empty((ncols, m), dtype=torch.float32, device=w.device)
        a = arr[i]
        a.type, a.W, a.w_row_stride, a.w_ch_stride = t, w.data_ptr(), B.row_size(t, k), 0
        a.act_q, a.act_ds, a.act_ch_stride = aq.data_ptr(), ads.data_ptr(), 0
        a.dst, a.dst_col_stride, a.dst_ch_stride = dst.data_ptr(), dst.stride(0), 0
        a.m, a.k, a.ncols, a.nchannels, a.ch_ratio = m, k, ncols, 1, 1
        outs.append(dst)
    K.check(K.lib.mi355_mul_mat_vec_q_group(arr, len(ws), _stream()), "mi355_mul_mat_vec_q_group")
    return outs


def act_bf16(b):
    """f32 [n, k] (row stride free, 16-byte aligned) -> bf16 [n, k] as uint16 (mi355_act_bf16)."""
    assert b.is_cuda and b.dtype == torch.float32 and b.dim() == 2 and b.stride(1) == 1
    n, k = b.shape
    y = torch.empty((n, k), dtype=torch.int16, device=b.device)
    K.check(K.lib.mi355_act_bf16(b.data_ptr(), y.data_ptr(), k, n, 1, 1, b.stride(0) * 4, 0, 0, _stream()), "mi355_act_bf16")
    return y


def mul_mat_dq(t, w, b, m):
    """Prefill GEMM through the dequantize-in-LDS bf16 MFMA kernel: b = [n, k] f32 -> [n, m] f32."""
    _u8(w)
    n, k = b.shape
    y = act_bf16(b)
    dst = torch.empty((n, m), dtype=torch.float32, device=b.device)
    K.check(K.mul_mat_dq(t, w.data_ptr(), B.row_size(t, k), y.data_ptr(), k, dst.data_ptr(), m, m, k, n,
                                   _stream()), "mi355_mul_mat_dq")
    return dst


def mul_mat(t, w, b, m, workspace=None):
    """ggml_mul_mat: w = [m rows] blocks of type t (uint8), b = [n, k] f32 -> [n, m] f32."""
    _u8(w)
    assert b.is_cuda and b.dtype == torch.float32 and b.dim() == 2 and b.is_contiguous()
    n, k = b.shape
    need = K.lib.mi355_mul_mat_workspace_bytes(k, n)
    if workspace is None:
        workspace = torch.empty(max(need, 256), dtype=torch.uint8, device=b.device)
    dst = torch.empty((n, m), dtype=torch.float32, device=b.device)
    K.check(K.lib.mi355_mul_mat(t, w.data_ptr(), B.row_size(t, k), b.data_ptr(), dst.data_ptr(), m, k, n,
                                workspace.data_ptr(), workspace.numel(), _stream()), "mi355_mul_mat")
    return dst


def mul_mat_id(t, as_, b, ids, m, n_expert):
    """ggml_mul_mat_id for decode-sized batches: as_ = [n_expert][m rows] blocks, b = [n_tokens, b_ne1, k] f32,
    ids = [n_tokens, n_used] int32 (device) -> [n_tokens, n_used, m] f32."""
    _u8(as_)
    assert b.dim() == 3 and b.is_contiguous() and ids.dtype == torch.int32 and ids.dim() == 2
    n_tok, b_ne1, k = b.shape
    n_used = ids.shape[1]
    aq, ads = quantize_act(b.reshape(n_tok * b_ne1, k))
    dst = torch.empty((n_tok, n_used, m), dtype=torch.float32, device=b.device)
    a = K.MmvqArgs()
    rs = B.row_size(t, k)
    a.type, a.W, a.w_row_stride, a.w_ch_stride = t, as_.data_ptr(), rs, rs * m
    a.act_q, a.act_ds, a.act_ch_stride = aq.data_ptr(), ads.data_ptr(), b_ne1
    a.dst, a.dst_col_stride, a.dst_ch_stride = dst.data_ptr(), m, m
    a.m, a.k, a.ncols, a.nchannels, a.ch_ratio = m, k, 1, n_tok * n_used, 1
    a.ids, a.ids_stride, a.ids_nused, a.act_ne1, a.n_expert = ids.data_ptr(), ids.stride(0), n_used, b_ne1, n_expert
    K.check(K.lib.mi355_mul_mat_vec_q(a, _stream()), "mi355_mul_mat_vec_q(ids)")
    return dst


def flash_attn_ext(q, k, v, mask, type_kv, n_kv, scale, max_bias=0.0, logit_softcap=0.0, use_workspace=True, quant=False,
                   workspace_bytes=None, noalloc=False, bf16=False, dv=None):
    """ggml_flash_attn_ext, batch 1.  q: [n_head, n_q, d] f32; k, v: [n_head_kv, n_kv, row_bytes] uint8 rows of
    type_kv (F16 / BF16 / Q8_0 / Q4_0); mask: [n_q_pad, n_kv] f16 bits (int16) or None -> [n_q, n_head, dv] f32 (dv: V's head size
    where it differs from d, e.g. 192 / 128)."""
    assert q.is_cuda and q.dtype == torch.float32 and q.is_contiguous() and k.is_contiguous() and v.is_contiguous()
    n_head, n_q, d = q.shape
    n_head_kv = k.shape[0]
    d_k, d = d, (dv if dv else d)          # below, d is the head size of V and dst
    dst = torch.empty((n_q, n_head, d), dtype=torch.float32, device=q.device)
    a = K.FattnArgs()
    a.q, a.q_nb1, a.q_nb2, a.q_nb3 = q.data_ptr(), d_k * 4, n_q * d_k * 4, n_head * n_q * d_k * 4
    rb = k.shape[2] * k.element_size()
    a.k, a.type_k, a.k_nb1, a.k_nb2, a.k_nb3 = k.data_ptr(), type_kv, rb, rb * k.shape[1], rb * k.shape[1] * n_head_kv
    rb = v.shape[2] * v.element_size()
    a.v, a.type_v, a.v_nb1, a.v_nb2, a.v_nb3 = v.data_ptr(), type_kv, rb, rb * v.shape[1], rb * v.shape[1] * n_head_kv
    if mask is not None:
        assert mask.is_contiguous() and mask.element_size() == 2
        a.mask, a.m_nb1, a.m_nb2, a.m_nb3, a.m_ne2, a.m_ne3 = mask.data_ptr(), mask.shape[1] * 2, mask.numel() * 2, mask.numel() * 2, 1, 1
    a.dst, a.d_nb1, a.d_nb2, a.d_nb3 = dst.data_ptr(), d * 4, n_head * d * 4, n_q * n_head * d * 4
    a.d, a.n_q, a.n_kv, a.n_head, a.n_head_kv, a.ne3 = d_k, n_q, n_kv, n_head, n_head_kv, 1
    a.dv = d if dv else 0
    a.scale, a.max_bias, a.logit_softcap = scale, max_bias, logit_softcap
    ws, wsb = None, 0
    if noalloc:      # the entry point that allocates nothing: the split-KV scratch is a fixed array inside the library
        K.check(K.lib.mi355_flash_attn_ext_noalloc(a, _stream()), "mi355_flash_attn_ext_noalloc")
        return dst
    if use_workspace:
        wsb = workspace_bytes if workspace_bytes is not None else K.lib.mi355_flash_attn_ext_workspace_bytes(a)
        ws = torch.zeros(max(wsb, 16), dtype=torch.uint8, device=q.device)   # arrival counters must start at zero
    if quant:
        aq = torch.empty((n_q, n_head * d), dtype=torch.int8, device=q.device)
        ads = torch.empty((n_q, n_head * d // 32, 2), dtype=torch.float32, device=q.device)
        for _ in range(2):   # twice on the same workspace: the counters must have returned to zero
            K.check(K.lib.mi355_flash_attn_ext_quant(a, ws.data_ptr() if ws is not None else None, wsb, aq.data_ptr(), ads.data_ptr(),
                                                     _stream()), "mi355_flash_attn_ext_quant")
        return dst, aq, ads
    if bf16:         # prefill: the bf16 rows of the result from the same launch
        abf = torch.empty((n_q, n_head * d), dtype=torch.int16, device=q.device)
        K.check(K.lib.mi355_flash_attn_ext_bf16(a, ws.data_ptr() if ws is not None else None, wsb, abf.data_ptr(), _stream()), "mi355_flash_attn_ext_bf16")
        return dst, abf
    K.check(K.lib.mi355_flash_attn_ext(a, ws.data_ptr() if ws is not None else None, wsb, _stream()), "mi355_flash_attn_ext")
    return dst


def rope(x, pos, n_dims, mode, n_ctx_orig=8192, freq_base=10000.0, freq_scale=1.0, ext_factor=0.0, attn_factor=1.0,
         beta_fast=32.0, beta_slow=1.0, freq_factors=None):
    """ggml_rope_ext on x = [n_tokens, n_head, head_dim] f32 (contiguous), pos = [n_tokens] int32."""
    assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and pos.dtype == torch.int32
    dst = torch.empty_like(x)
    tx, td = K.Tensor.of(x), K.Tensor.of(dst)
    K.check(K.lib.mi355_rope_f32(tx, pos.data_ptr(), freq_factors.data_ptr() if freq_factors is not None else None, td, n_dims, mode,
                                 n_ctx_orig, freq_base, freq_scale, ext_factor, attn_factor, beta_fast, beta_slow, _stream()), "mi355_rope_f32")
    return dst


def qkv_rope_matvec(type_qk, type_v, wq, wk, wv, m_q, m_k, m_v, k, aq, ads, pos, head_dim, n_dims, mode, kv_dtype=torch.float16,
                    n_ctx_orig=8192, freq_base=10000.0, freq_scale=1.0, ext_factor=0.0, attn_factor=1.0, beta_fast=32.0,
                    beta_slow=1.0, freq_factors=None, type_k=None, norm=None):
    """Fused decode attention input (mi355_qkv_rope_matvec): -> (q f32 [m_q], k [m_k], v [m_v] of kv_dtype).
    type_k: type of wk when it differs from wq's (type_qk)."""
    q = torch.empty(m_q, dtype=torch.float32, device=wq.device)
    ko = torch.empty(m_k, dtype=kv_dtype, device=wq.device)
    vo = torch.empty(m_v, dtype=kv_dtype, device=wq.device)
    a = K.QkvRopeArgs()
    a.type_qk, a.type_v, a.wq, a.wk, a.wv = type_qk, type_v, wq.data_ptr(), wk.data_ptr(), wv.data_ptr()
    a.wq_row_stride = B.row_size(type_qk, k)
    a.wk_row_stride = B.row_size(type_k if type_k is not None else type_qk, k)
    a.type_k = type_k if type_k is not None and type_k != type_qk else 0
    a.wv_row_stride = B.row_size(type_v, k)
    a.m_q, a.m_k, a.m_v, a.k = m_q, m_k, m_v, k
    a.q_out, a.k_out, a.v_out = q.data_ptr(), ko.data_ptr(), vo.data_ptr()
    if norm is not None:     # (x [k], w [k], eps, out [k]): rms_norm(x) * w computed inside the launch, aq / ads unused
        a.norm_x, a.norm_w, a.norm_eps, a.norm_out = norm[0].data_ptr(), norm[1].data_ptr(), norm[2], norm[3].data_ptr()
    else:
        a.act_q, a.act_ds = aq.data_ptr(), ads.data_ptr()
    a.k_out_type = a.v_out_type = B.F16 if kv_dtype == torch.float16 else B.F32
    a.pos, a.freq_factors = pos.data_ptr(), freq_factors.data_ptr() if freq_factors is not None else None
    a.head_dim, a.n_dims, a.mode, a.n_ctx_orig = head_dim, n_dims, mode, n_ctx_orig
    a.freq_base, a.freq_scale, a.ext_factor, a.attn_factor, a.beta_fast, a.beta_slow = freq_base, freq_scale, ext_factor, attn_factor, beta_fast, beta_slow
    K.check(K.lib.mi355_qkv_rope_matvec(a, _stream()), "mi355_qkv_rope_matvec")
    return q, ko, vo


def glu_matvec(t, w_gate, w_up, m, k, aq, ads, glu_op=2, quant=True, keep=True, norm=None, ids=None, n_expert=0):
    """Fused decode feed-forward input (mi355_glu_matvec): -> (dst f32 [m], gate f32 [m] | None, up f32 [m] | None,
    out_q int8 [m] | None, out_ds f32 [m/32, 2] | None).  norm = (x [k] f32, w [k] f32, eps, out [k] f32): the
    activations are rms_norm(x) * w, computed inside the launch (aq / ads unused, may be None)."""
    _u8(w_gate); _u8(w_up)
    dev = w_gate.device
    ns = 1 if ids is None else ids.numel()       # ids: int32 [n_slots] expert of each slot; w_gate / w_up then hold n_expert matrices
    dst = torch.full((ns * m,), float("nan"), dtype=torch.float32, device=dev)
    g = torch.full((ns * m,), float("nan"), dtype=torch.float32, device=dev) if keep else None
    u = torch.full((ns * m,), float("nan"), dtype=torch.float32, device=dev) if keep else None
    oq = torch.zeros(ns * m, dtype=torch.int8, device=dev) if quant else None
    ods = torch.zeros((ns * m // 32, 2), dtype=torch.float32, device=dev) if quant else None
    a = K.GluMatvecArgs()
    a.type, a.glu_op, a.w_gate, a.w_up = t, glu_op, w_gate.data_ptr(), w_up.data_ptr()
    a.gate_row_stride = a.up_row_stride = B.row_size(t, k)
    a.m, a.k = m, k
    if norm is not None:
        a.norm_x, a.norm_w, a.norm_eps, a.norm_out = norm[0].data_ptr(), norm[1].data_ptr(), norm[2], norm[3].data_ptr()
    else:
        a.act_q, a.act_ds = aq.data_ptr(), ads.data_ptr()
    a.gate_out, a.up_out, a.dst = g.data_ptr() if keep else None, u.data_ptr() if keep else None, dst.data_ptr()
    a.out_q, a.out_ds = oq.data_ptr() if quant else None, ods.data_ptr() if quant else None
    if ids is not None:
        a.ids, a.n_slots, a.n_expert = ids.data_ptr(), ns, n_expert
        a.gate_expert_stride = a.up_expert_stride = B.row_size(t, k) * m
    K.check(K.lib.mi355_glu_matvec(a, _stream()), "mi355_glu_matvec")
    return dst, g, u, oq, ods


_dl_sync = {}


def decode_layer_sync(n_calls=1):
    """A zeroed block of n_calls x mi355_decode_layer_sync_bytes() on the current device (memset on the current stream)."""
    nb = K.lib.mi355_decode_layer_sync_bytes()
    key = (torch.cuda.current_device(), n_calls)
    buf = _dl_sync.get(key)
    if buf is None:
        buf = _dl_sync[key] = torch.empty(n_calls * nb, dtype=torch.uint8, device="cuda")
    buf.zero_()
    return buf, nb


def decode_layer(types, w, n_embd, k_o, n_ff, attn, residual, ffn_norm_w, eps=1e-5, glu_op=2, tail=None, keep=True, sync=None, bufs=None):
    """The persistent decode-layer launch (mi355_decode_layer).  types = (type_o, type_gu, type_d); w = (wo, w_gate, w_up, w_down)
    uint8 block tensors; attn = f32 [k_o] or (act_q, act_ds); tail = None | ("qkv", QkvRopeArgs with everything but norm_x set) |
    ("rows", MmvqArgs with everything but src_f32 set).  bufs: output tensors to use instead of fresh ones, by name (tests place
    them on top of each other the way ggml-alloc does).  -> dict of the f32 outputs (+ "sync": the sync block, error word last)."""
    dev = w[0].device
    bufs = bufs or {}
    f = lambda name, n: bufs[name] if name in bufs else torch.full((n,), float("nan"), dtype=torch.float32, device=dev)
    o = {"ffn_inp": f("ffn_inp", n_embd), "ffn_norm": f("ffn_norm", n_embd), "hidden": f("hidden", n_ff), "l_out": f("l_out", n_embd)}
    if keep:
        o.update({"attn_out": f("attn_out", n_embd), "gate": f("gate", n_ff), "up": f("up", n_ff), "down": f("down", n_embd)})
    a = K.DecodeLayerArgs()
    a.type_o, a.type_gu, a.type_d, a.glu_op = types[0], types[1], types[2], glu_op
    a.n_embd, a.k_o, a.n_ff = n_embd, k_o, n_ff
    a.wo, a.wo_row_stride = _u8(w[0]).data_ptr(), B.row_size(types[0], k_o)
    if isinstance(attn, tuple):
        a.attn_q, a.attn_ds = attn[0].data_ptr(), attn[1].data_ptr()
    else:
        a.attn_f32 = attn.data_ptr()
    a.residual, a.ffn_inp = residual.data_ptr(), o["ffn_inp"].data_ptr()
    a.ffn_norm_w, a.ffn_norm_out, a.norm_eps = ffn_norm_w.data_ptr(), o["ffn_norm"].data_ptr(), eps
    a.w_gate, a.w_up, a.gu_row_stride = _u8(w[1]).data_ptr(), _u8(w[2]).data_ptr(), B.row_size(types[1], n_embd)
    a.hidden = o["hidden"].data_ptr()
    a.w_down, a.down_row_stride = _u8(w[3]).data_ptr(), B.row_size(types[2], n_ff)
    a.l_out = o["l_out"].data_ptr()
    if keep:
        a.attn_out, a.gate_out, a.up_out, a.down_out = (o[n].data_ptr() for n in ("attn_out", "gate", "up", "down"))
    a.tail = K.TAIL_NONE
    if tail is not None and tail[0] == "qkv":
        a.tail, a.qkv = K.TAIL_QKV, tail[1]
        a.qkv.norm_x = o["l_out"].data_ptr()
    elif tail is not None:
        a.tail, a.out = K.TAIL_ROWS, tail[1]
        a.out.src_f32 = o["l_out"].data_ptr()
    if sync is None:
        sync = decode_layer_sync()[0]
    K.check(K.lib.mi355_decode_layer(a, sync.data_ptr(), _stream()), "mi355_decode_layer")
    o["sync"] = sync
    return o


def moe_down_matvec(t, w, m, k, n_expert, aq, ads, ids, weights, residual=None, in_place=False):
    """Fused expert feed-forward output of one decode token (mi355_moe_down_matvec): w = [n_expert][m rows] blocks, aq / ads =
    q8_1 of the [n_slots, k] activations, ids int32 [n_slots], weights f32 [n_slots]
    -> (y [n_slots, m] | None when in_place, weighted [n_slots, m], sum [m], sum + residual [m] | None)."""
    _u8(w)
    ns, dev = ids.numel(), w.device
    y = None if in_place else torch.full((ns, m), float("nan"), dtype=torch.float32, device=dev)
    yw = torch.full((ns, m), float("nan"), dtype=torch.float32, device=dev)
    sm = torch.full((m,), float("nan"), dtype=torch.float32, device=dev)
    ro = torch.full((m,), float("nan"), dtype=torch.float32, device=dev) if residual is not None else None
    a = K.MoeDownArgs()
    a.type, a.n_slots, a.n_expert, a.W = t, ns, n_expert, w.data_ptr()
    a.w_row_stride = B.row_size(t, k)
    a.w_expert_stride = a.w_row_stride * m
    a.m, a.k, a.act_q, a.act_ds, a.ids, a.weights = m, k, aq.data_ptr(), ads.data_ptr(), ids.data_ptr(), weights.data_ptr()
    a.out, a.weighted_out, a.sum_out = None if in_place else y.data_ptr(), yw.data_ptr(), sm.data_ptr()
    if residual is not None:
        a.residual, a.residual_out = residual.data_ptr(), ro.data_ptr()
    K.check(K.lib.mi355_moe_down_matvec(a, _stream()), "mi355_moe_down_matvec")
    return y, yw, sm, ro


def mul_mat_id_dq(t, as_, b, ids, m, n_expert):
    """ggml_mul_mat_id for prefill-sized batches (device-side expert bucketing + one bf16-MFMA GEMM launch):
    as_ = [n_expert][m rows] blocks, b = [n_tokens, b_ne1, k] f32, ids = [n_tokens, n_used] int32 -> [n_tokens, n_used, m] f32."""
    _u8(as_)
    assert b.dim() == 3 and b.is_contiguous() and ids.dtype == torch.int32 and ids.dim() == 2 and ids.is_contiguous()
    n_tok, b_ne1, k = b.shape
    n_used = ids.shape[1]
    rs = B.row_size(t, k)
    need = K.lib.mi355_mul_mat_id_dq_workspace_bytes(k, n_tok, b_ne1, n_used, n_expert)
    ws = torch.empty(need, dtype=torch.uint8, device=b.device)
    dst = torch.full((n_tok, n_used, m), float("nan"), dtype=torch.float32, device=b.device)
    K.check(K.lib.mi355_mul_mat_id_dq(t, as_.data_ptr(), rs, rs * m, b.data_ptr(), k * 4, b_ne1 * k * 4, ids.data_ptr(), n_used,
                                      dst.data_ptr(), m, k, n_expert, n_used, n_tok, b_ne1, ws.data_ptr(), need, _stream()),
            "mi355_mul_mat_id_dq")
    return dst


def mul_mat_id_skinny(t, as_, b, ids, m, n_expert):
    """ggml_mul_mat_id for small batches (device-side bucketing + skinny int8-MFMA kernels, every expert streamed once
    per 32 of its columns); same arguments and result layout as mul_mat_id_dq."""
    _u8(as_)
    assert b.dim() == 3 and b.is_contiguous() and ids.dtype == torch.int32 and ids.dim() == 2 and ids.is_contiguous()
    n_tok, b_ne1, k = b.shape
    n_used = ids.shape[1]
    rs = B.row_size(t, k)
    need = K.lib.mi355_mul_mat_id_skinny_workspace_bytes(t, k, n_tok, b_ne1, n_used, n_expert)
    ws = torch.empty(need, dtype=torch.uint8, device=b.device)
    dst = torch.full((n_tok, n_used, m), float("nan"), dtype=torch.float32, device=b.device)
    K.check(K.lib.mi355_mul_mat_id_skinny(t, as_.data_ptr(), rs, rs * m, b.data_ptr(), k * 4, b_ne1 * k * 4, ids.data_ptr(), n_used,
                                          dst.data_ptr(), m, k, n_expert, n_used, n_tok, b_ne1, ws.data_ptr(), need, _stream()),
            "mi355_mul_mat_id_skinny")
    return dst


def mul_mat_id_skinny_pair(t, as0, as1, b, ids, m, n_expert):
    """two expert tensors (ffn_up_exps, ffn_gate_exps) on the same b and ids in one skinny launch -> two [n_tokens, n_used, m]."""
    _u8(as0); _u8(as1)
    n_tok, b_ne1, k = b.shape
    n_used = ids.shape[1]
    rs = B.row_size(t, k)
    need = K.lib.mi355_mul_mat_id_skinny_workspace_bytes(t, k, n_tok, b_ne1, n_used, n_expert)
    ws = torch.empty(need, dtype=torch.uint8, device=b.device)
    d0 = torch.full((n_tok, n_used, m), float("nan"), dtype=torch.float32, device=b.device)
    d1 = torch.full((n_tok, n_used, m), float("nan"), dtype=torch.float32, device=b.device)
    K.check(K.lib.mi355_mul_mat_id_skinny_pair(t, as0.data_ptr(), as1.data_ptr(), rs, rs * m, b.data_ptr(), k * 4, b_ne1 * k * 4, ids.data_ptr(),
                                               n_used, d0.data_ptr(), d1.data_ptr(), m, k, n_expert, n_used, n_tok, b_ne1, ws.data_ptr(), need,
                                               _stream()), "mi355_mul_mat_id_skinny_pair")
    return d0, d1


def mul_mat_id_skinny_ex(t, as0, as1, b, ids, m, n_expert, ws, tables, flags):
    """mi355_mul_mat_id_skinny_ex on a caller-kept workspace and routing-table buffer (flags: 1 = the workspace already holds b's
    activation planes, 2 = `tables` already holds the tables of these ids); as1 None: one tensor -> dst0 (, dst1)."""
    n_tok, b_ne1, k = b.shape
    n_used = ids.shape[1]
    rs = B.row_size(t, k)
    d0 = torch.full((n_tok, n_used, m), float("nan"), dtype=torch.float32, device=b.device)
    d1 = torch.full((n_tok, n_used, m), float("nan"), dtype=torch.float32, device=b.device) if as1 is not None else None
    K.check(K.lib.mi355_mul_mat_id_skinny_ex(t, as0.data_ptr(), as1.data_ptr() if as1 is not None else None, rs, rs * m, b.data_ptr(), k * 4,
                                             b_ne1 * k * 4, ids.data_ptr(), n_used, d0.data_ptr(), d1.data_ptr() if d1 is not None else None, m, k,
                                             n_expert, n_used, n_tok, b_ne1, ws.data_ptr(), ws.numel(), tables.data_ptr() if tables is not None else None,
                                             flags, _stream()), "mi355_mul_mat_id_skinny_ex")
    return (d0, d1) if as1 is not None else d0


def mul_mat_skinny(t, w, b, m):
    """MUL_MAT for small batches through the int8-MFMA skinny kernel, in passes of 32 columns: b = [n, k] f32 -> [n, m]."""
    _u8(w)
    n, k = b.shape
    aq, ads = quantize_act(b)
    dst = torch.empty((n, m), dtype=torch.float32, device=b.device)
    for c0 in range(0, n, 32):
        nc = min(32, n - c0)
        K.check(K.lib.mi355_mul_mat_skinny(t, w.data_ptr(), B.row_size(t, k), aq[c0:].data_ptr(), ads[c0:].data_ptr(),
                                           dst[c0:].data_ptr(), m, m, k, nc, _stream()), "mi355_mul_mat_skinny")
    return dst


def quantize_act_q8k(x):
    """[n, k] f32 -> q8_K planes (int8 [n, k], f32 [n, k/256], int16 [n, k/16])."""
    n, k = x.shape
    q = torch.empty((n, k), dtype=torch.int8, device=x.device)
    d = torch.empty((n, k // 256), dtype=torch.float32, device=x.device)
    bs = torch.empty((n, k // 16), dtype=torch.int16, device=x.device)
    K.check(K.lib.mi355_quantize_act_q8k(x.data_ptr(), q.data_ptr(), d.data_ptr(), bs.data_ptr(), k, n, 1, 1, x.stride(0) * 4, 0, 0, _stream()),
            "mi355_quantize_act_q8k")
    return q, d, bs


def mul_mat_skinny_k(t, w, b, m, pass_cols=32):
    """K-quant MUL_MAT on q8_K activations through the digit-split int8-MFMA kernel, passes of 32 (or 64: two column tiles per
    workgroup) columns."""
    _u8(w)
    n, k = b.shape
    q, d, bs = quantize_act_q8k(b)
    dst = torch.empty((n, m), dtype=torch.float32, device=b.device)
    for c0 in range(0, n, pass_cols):
        K.check(K.lib.mi355_mul_mat_skinny_k(t, w.data_ptr(), B.row_size(t, k), q[c0:].data_ptr(), d[c0:].data_ptr(), bs[c0:].data_ptr(),
                                             dst[c0:].data_ptr(), m, m, k, min(pass_cols, n - c0), _stream()), "mi355_mul_mat_skinny_k")
    return dst


def mul_mat_skinny_k_group(t, ws, b, ms):
    """Up to four K-quant matrices of type t against the same [n <= 64, k] activations in one launch -> list of [n, m_i]."""
    import ctypes as C
    n, k = b.shape
    q, d, bs = quantize_act_q8k(b)
    cnt = len(ws)
    dsts = [torch.empty((n, m), dtype=torch.float32, device=b.device) for m in ms]
    Wp = (C.c_void_p * cnt)(*[_u8(w).data_ptr() for w in ws])
    rs = (C.c_int64 * cnt)(*[B.row_size(t, k)] * cnt)
    dp = (C.c_void_p * cnt)(*[x.data_ptr() for x in dsts])
    mm = (C.c_int64 * cnt)(*ms)
    K.check(K.lib.mi355_mul_mat_skinny_k_group(t, cnt, Wp, rs, dp, mm, mm, q.data_ptr(), d.data_ptr(), bs.data_ptr(), k, n, _stream()),
            "mi355_mul_mat_skinny_k_group")
    return dsts


def mul_mat_qk(t, w, b, m):
    """K-quant prefill MUL_MAT on q8_K activations (digit-split int8-MFMA tile kernel): b = [n, k] f32 -> [n, m]."""
    _u8(w)
    n, k = b.shape
    q, d, bs = quantize_act_q8k(b)
    dst = torch.empty((n, m), dtype=torch.float32, device=b.device)
    K.check(K.lib.mi355_mul_mat_qk(t, w.data_ptr(), B.row_size(t, k), q.data_ptr(), d.data_ptr(), bs.data_ptr(), dst.data_ptr(), m, m, k, n,
                                   _stream()), "mi355_mul_mat_qk")
    return dst


def _desc(t, type_id):
    """ggml-style descriptor of a torch tensor of up to 4 dims (last dim fastest), strides honoured"""
    d = K.Tensor()
    d.data, d.type = t.data_ptr(), type_id
    shape, stride = list(t.shape)[::-1], list(t.stride())[::-1]
    es = t.element_size()
    for i in range(4):
        d.ne[i] = shape[i] if i < len(shape) else 1
        d.nb[i] = stride[i] * es if i < len(shape) else (d.nb[i - 1] * d.ne[i - 1] if i else es)
    return d


def mul_mat_f(type_a, a, b, type_b=B.F32):
    """MUL_MAT on float weights (mi355_mul_mat_f): a [.., m, k] (f32 | f16 | bf16 bits as a torch tensor of matching width),
    b [.., n, k] f32 | f16 -> [.., n, m] f32; leading dims broadcast ggml-style."""
    assert a.is_cuda and b.is_cuda
    lead = list(b.shape[:-2])
    dst = torch.empty(lead + [b.shape[-2], a.shape[-2]], dtype=torch.float32, device=b.device)
    ta, tb, td = _desc(a, type_a), _desc(b, type_b), _desc(dst, B.F32)
    K.check(K.lib.mi355_mul_mat_f(ta, tb, td, _stream()), "mi355_mul_mat_f")
    return dst


def get_rows(t, w, ids, k):
    """GGML_OP_GET_ROWS (mi355_get_rows): w uint8 [nrows, row_bytes] blocks of type t (k values per row), ids int32 [n] or
    [b, n] -> f32 [.., n, k], row ids[.., i] dequantized."""
    assert w.is_cuda and ids.is_cuda and ids.dtype == torch.int32 and w.dtype == torch.uint8 and w.dim() == 2
    dst = torch.empty(list(ids.shape) + [k], dtype=torch.float32, device=w.device)
    ts = K.Tensor()
    ts.data, ts.type = w.data_ptr(), t
    ts.ne[0], ts.ne[1], ts.ne[2], ts.ne[3] = k, w.shape[0], 1, 1
    ts.nb[0], ts.nb[1] = B.TYPE_SIZE[t], w.stride(0)
    ts.nb[2] = ts.nb[3] = 0 if ids.dim() > 1 else w.stride(0) * w.shape[0]     # one table for every batch of ids
    K.check(K.lib.mi355_get_rows(ts, _desc(ids, B.I32), _desc(dst, B.F32), _stream()), "mi355_get_rows")
    return dst


def set_rows(src, ids, dst, type_dst):
    """GGML_OP_SET_ROWS (mi355_set_rows): src f32 [.., n, k], ids int64 [.., n], dst rows of type_dst (f32 / f16 tensors,
    or uint8 rows of q8_0 blocks); writes into dst and returns it."""
    ts, ti = _desc(src, B.F32), _desc(ids, B.I64)
    td = _desc(dst, type_dst)
    if type_dst in (B.Q8_0, B.Q4_0):  # uint8 rows of 34- / 18-byte blocks: extents in elements, nb[0] the block size
        td.ne[0] = dst.shape[-1] // B.TYPE_SIZE[type_dst] * 32
        td.nb[0] = B.TYPE_SIZE[type_dst]
    K.check(K.lib.mi355_set_rows(ts, ti, td, _stream()), "mi355_set_rows")
    return dst
