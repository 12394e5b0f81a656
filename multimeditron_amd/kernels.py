"""Tensor-level wrappers over the C ABI (include/mm_hip.h).  torch is used for device memory and the current
HIP stream only; every arithmetic step below is a libmmhip kernel.  No autograd here (see functional.py)."""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import (EPI_ACCUMULATE, EPI_BIAS, EPI_GELU_ERF, EPI_QUICK_GELU, EPI_RESIDUAL, GEMM_NN, GEMM_NT, GEMM_TN,
                   MM_BF16, MM_F32, call)

_DT = {torch.bfloat16: MM_BF16, torch.float32: MM_F32}


def dt(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"libmmhip supports bf16/f32 tensors, got {t.dtype}") from None


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.MMHipError("libmmhip kernels need device tensors (no CPU fallback)")
    return t.data_ptr()


def pad8(n: int) -> int:
    return (n + 7) // 8 * 8


# ------------------------------------------------------------------------------------------------ GEMM
def gemm(layout: int, a: torch.Tensor, b: torch.Tensor, M: int, N: int, K: int, out: Optional[torch.Tensor] = None,
         bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None, act: int = 0,
         accumulate: bool = False, ldc_pad: bool = False, sumsq: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Raw GEMM on 2-D row-major tensors (strides taken from .stride(0)).  Returns C [M, N] (a view of a padded
    buffer when ldc_pad).  sumsq: fp32 partial buffer -> mm_gemm_sumsq (the GEMM also leaves sum(C^2) there)."""
    assert a.dim() == 2 and b.dim() == 2 and a.stride(1) == 1 and b.stride(1) == 1
    if out is None:
        if ldc_pad:
            ld = (N + 63) // 64 * 64
            buf = torch.empty((M, ld), dtype=a.dtype, device=a.device)
            out = buf[:, :N]
        else:
            out = torch.empty((M, N), dtype=a.dtype, device=a.device)
    assert out.stride(1) == 1
    epi = act
    if bias is not None:
        epi |= EPI_BIAS
    if residual is not None:
        epi |= EPI_RESIDUAL
        assert residual.stride(-1) == 1
    if accumulate:
        epi |= EPI_ACCUMULATE
    if sumsq is not None:
        assert bias is None and residual is None and act == 0
        call("mm_gemm_sumsq", dt(a), layout, M, N, K, _p(a), a.stride(0), _p(b), b.stride(0), _p(out), out.stride(0), epi, _p(sumsq),
             sumsq.numel(), _stream())
        return out
    call("mm_gemm", dt(a), layout, M, N, K, _p(a), a.stride(0), _p(b), b.stride(0), _p(out), out.stride(0), _p(bias),
         _p(residual), residual.stride(0) if residual is not None else 0, epi, _stream())
    return out


def linear_fwd(x2d, w, bias=None, residual=None, act=0, ldc_pad=False):
    """y[M,N] = x[M,K] @ w[N,K]^T (+bias)(act)(+residual)."""
    M, K = x2d.shape
    N = w.shape[0]
    return gemm(GEMM_NT, x2d, w, M, N, K, bias=bias, residual=residual, act=act, ldc_pad=ldc_pad)


def linear_act_fwd(x2d, w, bias, act, residual=None):
    """Training forward of Linear + GELU in one launch: -> (pre, y) with pre = bf16(x @ w^T + bias) kept for backward and
    y = act(pre) (+ residual).  Bit-identical to linear_fwd + gelu_fwd (+ add).  bf16 and M > 16 only: the caller falls back to the
    separate launches otherwise (this function returns None then)."""
    M, K = x2d.shape
    N = w.shape[0]
    if x2d.dtype != torch.bfloat16 or M <= 16 or _os.environ.get("MM_FUSED_GELU", "1") == "0":
        return None
    assert x2d.stride(1) == 1 and w.stride(1) == 1
    pre = torch.empty((M, N), dtype=x2d.dtype, device=x2d.device)
    y = torch.empty((M, N), dtype=x2d.dtype, device=x2d.device)
    epi = act | (EPI_BIAS if bias is not None else 0) | (EPI_RESIDUAL if residual is not None else 0)
    call("mm_gemm_act_fwd", dt(x2d), M, N, K, _p(x2d), x2d.stride(0), _p(w), w.stride(0), _p(bias), _p(pre), pre.stride(0), _p(y),
         y.stride(0), _p(residual), residual.stride(0) if residual is not None else 0, epi, _stream())
    return pre, y


def linear_dgrad(dy2d, w, out=None):
    """dx[M,K] = dy[M,N] @ w[N,K]."""
    M = dy2d.shape[0]
    N, K = w.shape
    return gemm(GEMM_NN, dy2d, w, M, K, N, out=out)


def linear_wgrad(dy2d, x2d, out, accumulate, sumsq=None):
    """dw[N,K] (+)= dy[M,N]^T @ x[M,K].  sumsq: fp32 partial buffer -> the GEMM also leaves sum(dw^2) there (mm_gemm_sumsq)."""
    M, N = dy2d.shape
    K = x2d.shape[1]
    if dy2d.dtype != torch.bfloat16:
        sumsq = None
    return gemm(GEMM_TN, dy2d, x2d, N, K, M, out=out, accumulate=accumulate, sumsq=sumsq)


def gemm_sumsq_slots(layout, M, N, K):
    import ctypes
    n = ctypes.c_int64(0)
    call("mm_gemm_sumsq_slots", MM_BF16, layout, M, N, K, ctypes.byref(n))
    return n.value


import os as _os


def _fused_swiglu():      # A/B switch (tools/step_ab.py): MM_FUSED_SWIGLU=0 -> GEMM + separate SwiGLU kernels
    return _os.environ.get("MM_FUSED_SWIGLU", "1") != "0"


def gemm_swiglu_fwd(x2d, wgu, I):
    """(gu [M, 2I], act [M, I]) = fused gate|up GEMM + SwiGLU; None when the shape must take the two-launch form."""
    M, K = x2d.shape
    if x2d.dtype != torch.bfloat16 or (I % 128) or (K % 64) or M < 256 or not _fused_swiglu():
        return None
    gu = torch.empty((M, 2 * I), dtype=x2d.dtype, device=x2d.device)
    act = torch.empty((M, I), dtype=x2d.dtype, device=x2d.device)
    call("mm_gemm_swiglu_fwd", dt(x2d), M, I, K, _p(x2d), x2d.stride(0), _p(wgu), wgu.stride(0), _p(gu), gu.stride(0), _p(act),
         act.stride(0), _stream())
    return gu, act


def _fused_rope():         # A/B switch (tools/step_ab.py): MM_FUSED_ROPE=0 -> GEMM + separate RoPE kernel
    return _os.environ.get("MM_FUSED_ROPE", "1") != "0"


def gemm_rope_fwd(x2d, w, bias, rope_cols, D, cos, sin):
    """qkv [M, N] = x @ w^T (+ bias) with RoPE on the first rope_cols columns, in one launch; None when the shape must take the
    two-launch form (mm_gemm + mm_rope_apply: same bits)."""
    M, K = x2d.shape
    N = w.shape[0]
    if (x2d.dtype != torch.bfloat16 or D != 128 or (N % 128) or (rope_cols % 128) or M < 256 or not _fused_rope()
            or cos.dtype != torch.float32 or cos.shape[-1] != D // 2):
        return None
    out = torch.empty((M, N), dtype=x2d.dtype, device=x2d.device)
    call("mm_gemm_rope_fwd", dt(x2d), M, N, K, _p(x2d), x2d.stride(0), _p(w), w.stride(0), _p(bias), _p(out), out.stride(0), int(rope_cols),
         int(D), _p(cos), _p(sin), _stream())
    return out


# ---- decode-step fusions (M = batch <= 16 rows; see include/mm_hip.h) ----------------------------------------------------------
def decode_fits(M, K):
    """x (M rows of K bf16) must fit the weight-streaming kernel's LDS stage."""
    return K % 8 == 0 and M * K * 2 <= 143 * 1024        # csrc/mm_gemm.hip GEMV_X_LDS_MAX (160 KB minus the kernel's static reduction scratch)


def decode_gateup_swiglu(x2d, wgu, I, norm_w=None, eps=0.0):
    """act [M, I] = silu(x' Wg^T) * (x' Wu^T), x' = rmsnorm(x) * norm_w when norm_w is given."""
    M, K = x2d.shape
    act = torch.empty((M, I), dtype=x2d.dtype, device=x2d.device)
    call("mm_decode_gateup_swiglu", dt(x2d), M, I, K, _p(x2d), x2d.stride(0), _p(wgu), wgu.stride(0), _p(act), act.stride(0), _p(norm_w), float(eps),
         _stream())
    return act


def decode_qkv_rope_append(x2d, w, bias, Hq, Hkv, D, cos, sin, kcache, vcache, pos, norm_w=None, eps=0.0):
    """qkv [B, (Hq+2Hkv)*D] = x' @ w^T (+ bias) with RoPE on q / k and the append of roped k / v to cache[:, pos]."""
    M, K = x2d.shape
    assert kcache.stride(3) == 1 and kcache.stride(2) == D and vcache.stride(2) == D and kcache.stride(0) == vcache.stride(0)
    qkv = torch.empty((M, (Hq + 2 * Hkv) * D), dtype=x2d.dtype, device=x2d.device)
    call("mm_decode_qkv_rope_append", dt(x2d), M, Hq, Hkv, D, K, _p(x2d), x2d.stride(0), _p(w), w.stride(0), _p(bias), _p(qkv), qkv.stride(0),
         _p(cos), _p(sin), _p(kcache[:, pos]), _p(vcache[:, pos]), kcache.stride(0), _p(norm_w), float(eps), _stream())
    return qkv


def decode_linear(x2d, w, bias=None, residual=None, norm_w=None, eps=0.0, ldc_pad=False):
    """c [M, N] = x' @ w^T (+ bias) (+ residual), x' = rmsnorm(x) * norm_w when norm_w is given; ldc_pad: row stride padded to 64."""
    M, K = x2d.shape
    N = w.shape[0]
    ld = (N + 63) // 64 * 64 if ldc_pad else N
    buf = torch.empty((M, ld), dtype=x2d.dtype, device=x2d.device)
    call("mm_decode_linear", dt(x2d), M, N, K, _p(x2d), x2d.stride(0), _p(w), w.stride(0), _p(bias), _p(residual),
         residual.stride(0) if residual is not None else 0, _p(buf), ld, _p(norm_w), float(eps), _stream())
    return buf[:, :N] if ld != N else buf


def decode_fusions():      # A/B switch (tools/decode_bench.py): MM_DECODE_FUSED=0 -> the separate launches
    return _os.environ.get("MM_DECODE_FUSED", "1") != "0"


def gemm_swiglu_bwd(dy2d, wd, gu, I):
    """dgu [M, 2I] from dy [M, H], down_proj weight [H, I] and the saved pre-activations; None -> two-launch form."""
    M, H = dy2d.shape
    if dy2d.dtype != torch.bfloat16 or (I % 4) or not _fused_swiglu():
        return None
    dgu = torch.empty_like(gu)
    call("mm_gemm_swiglu_bwd", dt(dy2d), M, I, H, _p(dy2d), dy2d.stride(0), _p(wd), wd.stride(0), _p(gu), gu.stride(0), _p(dgu),
         dgu.stride(0), _stream())
    return dgu


def colsum(x2d, out, accumulate):
    call("mm_colsum", dt(x2d), _p(x2d), x2d.shape[0], x2d.shape[1], x2d.stride(0), _p(out), int(accumulate), _stream())
    return out


# ------------------------------------------------------------------------------------------------ splice
def splice_build_map(batch_idx, token_range, S, T):
    m = torch.empty(T, dtype=torch.int32, device=batch_idx.device if batch_idx is not None else "cuda")
    n = 0 if batch_idx is None else batch_idx.numel()
    call("mm_splice_build_map", _p(batch_idx) if n else None, _p(token_range) if n else None, n, S, T, _p(m), _stream())
    return m


_oor_state = {}          # device index -> [device flag, pinned host copy, event of the last copy]


def embed_check_ids(ids, vocab, block=False):
    """nn.Embedding's range check without a stall: a tiny kernel raises a sticky device flag for ids outside [0, vocab); the
    flag travels to pinned memory behind it, and the host looks at the copy of the PREVIOUS call (complete by then), or waits for
    this one with block=True.  Raises IndexError like torch (reference model.py:433: every id of the batch is embedded)."""
    dev = ids.device.index or 0
    st = _oor_state.get(dev)
    if st is None:
        st = _oor_state[dev] = [torch.zeros(1, dtype=torch.int32, device=ids.device), torch.zeros(1, dtype=torch.int32).pin_memory(), None]
    _embed_raise_if_flagged(st, wait=False)
    call("mm_embed_check_ids", _p(ids), ids.numel(), int(vocab), _p(st[0]), _stream())
    st[1].copy_(st[0], non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    st[2] = ev
    if block:
        _embed_raise_if_flagged(st, wait=True)


def embed_check_pending():
    """Wait for the outstanding id checks of every device and raise if one failed (called where the host synchronises anyway)."""
    for st in _oor_state.values():
        _embed_raise_if_flagged(st, wait=True)


def _embed_raise_if_flagged(st, wait):
    ev = st[2]
    if ev is None:
        return
    if wait:
        ev.synchronize()
    elif not ev.query():
        return
    st[2] = None
    if int(st[1][0]) != 0:
        st[0].zero_()
        st[1].zero_()
        raise IndexError("index out of range in self")


def embed_splice_fwd(emb, ids, proj, src_map):
    embed_check_ids(ids, emb.shape[0])
    T = ids.numel()
    H = emb.shape[1]
    out = torch.empty((T, H), dtype=emb.dtype, device=emb.device)
    call("mm_embed_splice_fwd", dt(emb), _p(emb), emb.shape[0], H, _p(ids), _p(proj), _p(src_map) if proj is not None else None, T,
         _p(out), _stream())
    return out


def embed_sort(ids, src_map, vocab, H):
    """Stable token order by id for the embedding gradient -> (order, skey) int32 (padded, see mm_embed_sort_sizes)."""
    import ctypes
    T = ids.numel()
    n_order, n_scr = ctypes.c_int64(0), ctypes.c_int64(0)
    call("mm_embed_sort_sizes", T, H, ctypes.byref(n_order), ctypes.byref(n_scr))
    order = torch.empty(n_order.value, dtype=torch.int32, device=ids.device)
    skey = torch.empty(n_order.value, dtype=torch.int32, device=ids.device)
    ws = torch.empty(max(T, 1), dtype=torch.int32, device=ids.device)
    call("mm_embed_sort", _p(ids), _p(src_map), T, vocab, _p(ws), _p(order), _p(skey), _stream())
    return order, skey


def embed_splice_bwd(dE, ids, src_map, batch_idx, token_range, S, dproj, demb, order=None, skey=None, accumulate=False):
    T, H = dE.shape
    n = 0 if batch_idx is None else batch_idx.numel()
    scratch = None
    if demb is not None:
        scratch = torch.empty(((T + 31) // 32) * 2 * H, dtype=torch.float32, device=dE.device)
    call("mm_embed_splice_bwd", dt(dE), _p(dE), H, _p(ids), _p(src_map), T, _p(batch_idx) if n else None,
         _p(token_range) if n else None, n, S, _p(dproj), _p(demb), demb.shape[0] if demb is not None else 0, _p(order), _p(skey),
         _p(scratch), int(accumulate), _stream())


# ------------------------------------------------------------------------------------------------ ViT glue
def patchify(pixels, ps, kpad, dtype):
    n, c, h, w = pixels.shape
    assert c == 3 and pixels.dtype == torch.float32 and pixels.is_contiguous()
    P = (h // ps) * (w // ps)
    out = torch.empty((n * P, kpad), dtype=dtype, device=pixels.device)
    call("mm_patchify", _DT[dtype], _p(pixels), n, h, w, ps, kpad, _p(out), _stream())
    return out


def vit_embed_fwd(patch_out, cls, pos, n, P):
    D = cls.numel()
    x = torch.empty((n, P + 1, D), dtype=patch_out.dtype, device=patch_out.device)
    call("mm_vit_embed_fwd", dt(patch_out), _p(patch_out), _p(cls), _p(pos), n, P, D, _p(x), _stream())
    return x


def vit_embed_bwd(dx, dcls, dpos, accumulate, want_dpatch=True):
    n, T, D = dx.shape
    P = T - 1
    dpatch = torch.empty((n * P, D), dtype=dx.dtype, device=dx.device) if want_dpatch else None
    call("mm_vit_embed_bwd", dt(dx), _p(dx), n, P, D, _p(dpatch), _p(dcls), _p(dpos), int(accumulate), _stream())
    return dpatch


def bcast_add(x, b):
    """y[n, ...] = x[n, ...] + b[...] (b broadcast over the leading axis)."""
    y = torch.empty_like(x)
    call("mm_bcast_add", dt(x), _p(x), _p(b), x.shape[0], b.numel(), _p(y), _stream())
    return y


def head_pad(x2d, nheads, d, dpad, inverse=False):
    """[rows, nheads*d] -> [rows, nheads*dpad] (zero-padded heads), or back with inverse=True."""
    rows = x2d.shape[0]
    assert x2d.is_contiguous() and x2d.shape[1] == nheads * (dpad if inverse else d)
    out = torch.empty((rows, nheads * (d if inverse else dpad)), dtype=x2d.dtype, device=x2d.device)
    call("mm_head_pad", dt(x2d), _p(x2d), rows, nheads, d, dpad, _p(out), int(inverse), _stream())
    return out


def drop_cls_fwd(x):
    n, T, D = x.shape
    out = torch.empty((n, T - 1, D), dtype=x.dtype, device=x.device)
    call("mm_drop_cls_fwd", dt(x), _p(x), n, T - 1, D, _p(out), _stream())
    return out


def drop_cls_bwd(dy):
    n, P, D = dy.shape
    out = torch.empty((n, P + 1, D), dtype=dy.dtype, device=dy.device)
    call("mm_drop_cls_bwd", dt(dy), _p(dy), n, P, D, _p(out), _stream())
    return out


def rows_select(src2d, row_map, n_dst):
    """dst[r] = src2d[row_map[r]] (zeros where row_map[r] < 0); row_map int32 on the device, n_dst known to the host."""
    D = src2d.shape[1]
    assert row_map.dtype == torch.int32 and row_map.is_contiguous() and row_map.numel() >= n_dst and src2d.stride(1) == 1
    out = torch.empty((n_dst, D), dtype=src2d.dtype, device=src2d.device)
    call("mm_rows_select", dt(src2d), _p(src2d), src2d.stride(0), _p(row_map), src2d.shape[0], n_dst, D, _p(out), D, _stream())
    return out


# ------------------------------------------------------------------------------------------------ norms
def rmsnorm_fwd(x2d, w, eps):
    M, H = x2d.shape
    y = torch.empty_like(x2d)
    rstd = torch.empty(M, dtype=torch.float32, device=x2d.device)
    call("mm_rmsnorm_fwd", dt(x2d), _p(x2d), _p(w), M, H, float(eps), _p(y), _p(rstd), _stream())
    return y, rstd


def norm_bwd_blocks(M):
    return _lib.lib().mm_norm_bwd_blocks(M)


def rmsnorm_bwd(dy2d, x2d, w, rstd, dres=None):
    M, H = x2d.shape
    dx = torch.empty_like(x2d)
    dwp = torch.empty((norm_bwd_blocks(M), H), dtype=torch.float32, device=x2d.device)
    call("mm_rmsnorm_bwd", dt(x2d), _p(dy2d), _p(x2d), _p(w), _p(rstd), M, H, _p(dx), _p(dwp), _p(dres), _stream())
    return dx, dwp


def layernorm_fwd(x2d, w, b, eps):
    M, H = x2d.shape
    y = torch.empty_like(x2d)
    mean = torch.empty(M, dtype=torch.float32, device=x2d.device)
    rstd = torch.empty(M, dtype=torch.float32, device=x2d.device)
    call("mm_layernorm_fwd", dt(x2d), _p(x2d), _p(w), _p(b), M, H, float(eps), _p(y), _p(mean), _p(rstd), _stream())
    return y, mean, rstd


def layernorm_bwd(dy2d, x2d, w, mean, rstd, dres=None):
    M, H = x2d.shape
    dx = torch.empty_like(x2d)
    nb = norm_bwd_blocks(M)
    dwp = torch.empty((nb, H), dtype=torch.float32, device=x2d.device)
    dbp = torch.empty((nb, H), dtype=torch.float32, device=x2d.device)
    call("mm_layernorm_bwd", dt(x2d), _p(dy2d), _p(x2d), _p(w), _p(mean), _p(rstd), M, H, _p(dx), _p(dwp), _p(dbp), _p(dres), _stream())
    return dx, dwp, dbp


def reduce_partials2(partial0, partial1, out0, out1, accumulate0, accumulate1):
    assert partial0.shape == partial1.shape and out0.dtype == out1.dtype
    call("mm_reduce_partials2", dt(out0), _p(partial0), _p(partial1), partial0.shape[0], partial0.shape[1], _p(out0), _p(out1), int(accumulate0),
         int(accumulate1), _stream())


def reduce_partials(partial, out, accumulate):
    call("mm_reduce_partials", dt(out), _p(partial), partial.shape[0], partial.shape[1], _p(out), int(accumulate), _stream())
    return out


# ------------------------------------------------------------------------------------------------ rope
def rope_table(position_ids, inv_freq, round_bf16):
    T = position_ids.numel()
    half = inv_freq.numel()
    cos = torch.empty((T, half), dtype=torch.float32, device=position_ids.device)
    sin = torch.empty_like(cos)
    call("mm_rope_table", _p(position_ids), _p(inv_freq), T, half, int(round_bf16), _p(cos), _p(sin), _stream())
    return cos, sin


def rope_apply_(x, T, nheads, D, ld, cos, sin, inverse=False):
    """in place on x (any tensor whose storage at data_ptr is [T, nheads, D] with row stride ld)."""
    call("mm_rope_apply", dt(x), _p(x), T, nheads, D, ld, _p(cos), _p(sin), int(inverse), _stream())
    return x


def rope_append_(qkv, B, Hq, Hkv, D, cos, sin, kcache, vcache, pos):
    """decode step: RoPE q/k in place + write roped k and v to cache[:, pos] ([B, Smax, Hkv, D] contiguous in the last two)."""
    assert kcache.stride(3) == 1 and kcache.stride(2) == D and vcache.stride(2) == D and kcache.stride(0) == vcache.stride(0)
    call("mm_rope_append", dt(qkv), _p(qkv), B, Hq, Hkv, D, qkv.stride(0), _p(cos), _p(sin), _p(kcache[:, pos]), _p(vcache[:, pos]),
         kcache.stride(0), _stream())
    return qkv


# ------------------------------------------------------------------------------------------------ attention
def _strides3(t):  # t: [B, S, H, D] view
    assert t.dim() == 4 and t.stride(3) == 1
    return t.stride(0), t.stride(1), t.stride(2)


def attn_fwd(q, k, v, key_mask, causal, scale):
    """q [B,Sq,Hq,D], k/v [B,Skv,Hkv,D] (strided views ok) -> out [B,Sq,Hq,D] contiguous, lse [B,Hq,Sq] f32."""
    B, Sq, Hq, D = q.shape
    Skv, Hkv = k.shape[1], k.shape[2]
    out = torch.empty((B, Sq, Hq, D), dtype=q.dtype, device=q.device)
    lse = torch.empty((B, Hq, Sq), dtype=torch.float32, device=q.device)
    call("mm_attn_fwd", dt(q), _p(q), _p(k), _p(v), B, Sq, Skv, Hq, Hkv, D, *_strides3(q), *_strides3(k), *_strides3(v),
         _p(key_mask), int(causal), float(scale), _p(out), _p(lse), _stream())
    return out, lse


def attn_bwd(q, k, v, out, dout, lse, key_mask, causal, scale, dq, dk, dv):
    """dq/dk/dv: preallocated views with the SAME strides as q/k/v."""
    B, Sq, Hq, D = q.shape
    Skv, Hkv = k.shape[1], k.shape[2]
    assert _strides3(dq) == _strides3(q) and _strides3(dk) == _strides3(k) and _strides3(dv) == _strides3(v)
    assert dout.is_contiguous() and out.is_contiguous()
    delta = torch.empty((B, Hq, Sq), dtype=torch.float32, device=q.device)
    call("mm_attn_bwd", dt(q), _p(q), _p(k), _p(v), _p(out), _p(dout), _p(lse), B, Sq, Skv, Hq, Hkv, D, *_strides3(q),
         *_strides3(k), *_strides3(v), _p(key_mask), int(causal), float(scale), _p(dq), _p(dk), _p(dv), _p(delta), _stream())


def attn_decode_supported(q_dtype, Hq, Hkv, D):
    return q_dtype == torch.bfloat16 and D in (64, 128) and Hq % Hkv == 0 and (Hq // Hkv) in (1, 2, 4, 7, 8)


_decode_sync = {}


def attn_decode(q, k, v, key_mask, scale):
    """One query token per sequence over a KV cache: q [B,Hq,D] (strided view ok), k/v [B,Skv,Hkv,D] -> out [B,Hq,D]."""
    B, Hq, D = q.shape
    Skv, Hkv = k.shape[1], k.shape[2]
    assert q.stride(2) == 1 and k.stride(3) == 1 and v.stride(3) == 1
    ns = _lib.lib().mm_attn_decode_splits(B, Hkv, Skv)
    ws = torch.empty(B * Hq * ns * (D + 2), dtype=torch.float32, device=q.device)
    out = torch.empty((B, Hq, D), dtype=q.dtype, device=q.device)
    # sync=None: the slices are merged by a second (tiny) launch.  The single-launch form (sync = arrival counters [B * Hkv]; the
    # slice that arrives last merges) is correct but SLOWER on MI355X in every protocol tried: __threadfence() per thread (round 2:
    # 8.2 vs 5.2 ms/token), one agent-scope release per workgroup (round 3: the partial kernel went from 18 to 75 us -- buffer_wbl2
    # with an L2 full of dirty lines), write-through record stores + one acquire by the last arriver (33 us against 18 + 8 for the
    # two launches: the merging workgroup starts only when the last slice is done and reads records that come from beyond L2).
    # MM_DECODE_MERGE=fused selects it.
    sync = None
    if _os.environ.get("MM_DECODE_MERGE", "launch") == "fused":
        sync = _decode_sync.get(q.device)
        if sync is None or sync.numel() < B * Hkv:
            sync = _decode_sync[q.device] = torch.zeros(max(64, B * Hkv), dtype=torch.int32, device=q.device)
    call("mm_attn_decode", dt(q), _p(q), _p(k), _p(v), B, Skv, Hq, Hkv, D, q.stride(0), q.stride(1), k.stride(0), k.stride(1),
         k.stride(2), v.stride(0), v.stride(1), v.stride(2), _p(key_mask), float(scale), _p(out), _p(ws), ns, _p(sync), _stream())
    return out


# ------------------------------------------------------------------------------------------------ small cross-attention + dropout
def xattn_supported(dtype, Nkv, D):
    """Shapes mm_xattn_* takes: up to 1024 keys (a query's whole score row lives in a wave's registers: 64 MFMA tiles; above 512 keys
    the kernel trades speed for it -- it spills -- which is fine at a cross-attention's size); bf16 head widths that are multiples of
    8 up to 512, any fp32 width whose score / query rows fit the wave-per-row kernel's LDS."""
    if Nkv > 1024:
        return False
    if dtype == torch.bfloat16:
        return D % 8 == 0 and D <= 512
    return dtype == torch.float32 and (2 * Nkv + 2 * D) * 4 <= 60000


def xattn_fwd(q, k, v, scale, drop_p=0.0, seed=0, offset=0):
    """q [n,Nq,H,D], k/v [n,Nkv,H,D] (strided views ok) -> out [n,Nq,H,D] contiguous, lse [n,H,Nq] f32; attention-probability
    dropout with probability drop_p from the Philox stream (seed, offset) (see include/mm_hip.h)."""
    n, Nq, H, D = q.shape
    Nkv = k.shape[1]
    out = torch.empty((n, Nq, H, D), dtype=q.dtype, device=q.device)
    lse = torch.empty((n, H, Nq), dtype=torch.float32, device=q.device)
    call("mm_xattn_fwd", dt(q), _p(q), _p(k), _p(v), n, Nq, Nkv, H, D, *_strides3(q), *_strides3(k), *_strides3(v), float(scale),
         float(drop_p), int(seed), int(offset), _p(out), _p(lse), _stream())
    return out, lse


def xattn_bwd(q, k, v, out, dout, lse, scale, drop_p, seed, offset, dq, dk, dv):
    """dq/dk/dv: preallocated views with the SAME strides as q/k/v.  Deterministic (no atomics)."""
    import ctypes
    n, Nq, H, D = q.shape
    Nkv = k.shape[1]
    assert _strides3(dq) == _strides3(q) and _strides3(dk) == _strides3(k) and _strides3(dv) == _strides3(v)
    assert dout.is_contiguous() and out.is_contiguous()
    nb = ctypes.c_int64(0)
    call("mm_xattn_ws_bytes", dt(q), n, Nq, Nkv, H, ctypes.byref(nb))
    ws = torch.empty(max(nb.value, 16), dtype=torch.uint8, device=q.device)
    call("mm_xattn_bwd", dt(q), _p(q), _p(k), _p(v), _p(out), _p(dout), _p(lse), n, Nq, Nkv, H, D, *_strides3(q), *_strides3(k),
         *_strides3(v), float(scale), float(drop_p), int(seed), int(offset), _p(dq), _p(dk), _p(dv), _p(ws), nb.value, _stream())


def dropout(x, p, seed, offset):
    """y = x * keep / (1 - p) with keep from the Philox stream (seed, offset); its own adjoint (apply it to dy)."""
    x = x.contiguous()
    y = torch.empty_like(x)
    call("mm_dropout", dt(x), _p(x), x.numel(), float(p), int(seed), int(offset), _p(y), _stream())
    return y


def dropout_mask(seed, offset, n, p, device="cuda"):
    """keep flags (uint8) of elements 0 .. n-1 of the Philox stream (seed, offset): what mm_dropout / mm_xattn_* apply (tests)."""
    m = torch.empty(n, dtype=torch.uint8, device=device)
    call("mm_dropout_mask", int(seed), int(offset), n, float(p), _p(m), _stream())
    return m


# ------------------------------------------------------------------------------------------------ activations
def swiglu_fwd(gu, I):
    M = gu.shape[0]
    out = torch.empty((M, I), dtype=gu.dtype, device=gu.device)
    call("mm_swiglu_fwd", dt(gu), _p(gu), M, I, _p(out), _stream())
    return out


def swiglu_bwd(gu, dout, I):
    dgu = torch.empty_like(gu)
    call("mm_swiglu_bwd", dt(gu), _p(gu), _p(dout), gu.shape[0], I, _p(dgu), _stream())
    return dgu


def gelu_fwd(x, kind):
    y = torch.empty_like(x)
    call("mm_gelu_fwd", dt(x), kind, _p(x), x.numel(), _p(y), _stream())
    return y


def gelu_bwd(x, dy, kind):
    dx = torch.empty_like(x)
    call("mm_gelu_bwd", dt(x), kind, _p(x), _p(dy), x.numel(), _p(dx), _stream())
    return dx


def add(a, b):
    y = torch.empty_like(a)
    call("mm_add", dt(a), _p(a), _p(b), a.numel(), _p(y), _stream())
    return y


# ------------------------------------------------------------------------------------------------ loss
def ce_fwd(logits2d, V, labels):
    """logits2d: [T, ld] view with V valid columns.  -> (loss_and_count [2] f32, lse [T])."""
    T = logits2d.shape[0]
    ld = logits2d.stride(0)
    lse = torch.empty(T, dtype=torch.float32, device=logits2d.device)
    loss_row = torch.empty(T, dtype=torch.float32, device=logits2d.device)
    out = torch.empty(2, dtype=torch.float32, device=logits2d.device)
    call("mm_ce_fwd", dt(logits2d), _p(logits2d), T, V, ld, _p(labels), _p(lse), _p(loss_row), _stream())
    call("mm_ce_reduce", _p(loss_row), _p(labels), T, _p(out), _stream())
    return out, lse


def ce_bwd(logits2d, V, labels, lse, loss_and_count, gscale, dlogits2d):
    T = logits2d.shape[0]
    ld = logits2d.stride(0)
    assert dlogits2d.stride(0) == ld
    call("mm_ce_bwd", dt(logits2d), _p(logits2d), T, V, ld, _p(labels), _p(lse), _p(loss_and_count), _p(gscale), _p(dlogits2d),
         _stream())
    return dlogits2d


def argmax_softmax(logits2d, V, temperature):
    rows = logits2d.shape[0]
    out = torch.empty(rows, dtype=torch.int64, device=logits2d.device)
    if V >= 16384 and _os.environ.get("MM_ARGMAX_SPLIT", "1") != "0":      # long rows: the vocabulary over many workgroups
        nb = _lib.lib().mm_argmax_softmax_ws_bytes(rows, V)
        ws = torch.empty((nb + 7) // 8, dtype=torch.int64, device=logits2d.device)
        call("mm_argmax_softmax_split", dt(logits2d), _p(logits2d), rows, V, logits2d.stride(0), float(temperature), _p(out), _p(ws), _stream())
        return out
    call("mm_argmax_softmax", dt(logits2d), _p(logits2d), rows, V, logits2d.stride(0), float(temperature), _p(out), _stream())
    return out


def expert_fuse(x, gate, idx, mode, backward=False, E=None):
    """MoE fusion (see mm_expert_fuse).  forward: x [E, n, L] -> [n, L] (mode 0) / [n, J, L] (mode 1); backward: x = dout ->
    dX [E, n, L] (zeros outside the listed experts)."""
    import ctypes
    J = len(idx)
    ix = (ctypes.c_int * J)(*[int(i) for i in idx])
    if not backward:
        E_, n, L = x.shape
        out = torch.empty((n, L) if mode == 0 else (n, J, L), dtype=x.dtype, device=x.device)
    else:
        E_ = int(E)
        n, L = x.shape[0], x.shape[-1]
        out = torch.zeros((E_, n, L), dtype=x.dtype, device=x.device)
    call("mm_expert_fuse", dt(x), int(backward), int(mode), _p(x), _p(gate), ix, J, E_, n, L, _p(out), _stream())
    return out


def decode_select(tok, finished, eos, out, col, next_ids):
    """device-side eos bookkeeping of one decode step (no host sync): see mm_decode_select."""
    call("mm_decode_select", _p(tok), _p(finished), int(eos), tok.numel(), _p(out), out.stride(0), int(col), _p(next_ids), _stream())


# ------------------------------------------------------------------------------------------------ optimizer
def gradnorm(flat_grads, max_norm):
    """flat_grads: list of 1-D flat gradient buffers -> device tensor [2] = (total_norm, clip_coef)."""
    nblk = 1024
    dev = flat_grads[0].device
    partial = torch.empty(nblk * len(flat_grads), dtype=torch.float32, device=dev)
    for i, g in enumerate(flat_grads):
        call("mm_gradnorm_partial", dt(g), _p(g), g.numel(), partial.data_ptr() + 4 * nblk * i, nblk, _stream())
    total = torch.empty(2, dtype=torch.float32, device=dev)
    call("mm_gradnorm_finish", _p(partial), nblk * len(flat_grads), float(max_norm), _p(total), _stream())
    return total


def gradnorm_partial(g, partial):
    """sum of squares of one flat gradient slice -> partial[:] (one float per workgroup; any stream)."""
    call("mm_gradnorm_partial", dt(g), _p(g), g.numel(), _p(partial), partial.numel(), _stream())


def gradnorm_finish(partial, max_norm):
    """-> device tensor [2] = (total_norm, clip_coef) from all the partial slots, summed in slot order (deterministic)."""
    total = torch.empty(2, dtype=torch.float32, device=partial.device)
    call("mm_gradnorm_finish", _p(partial), partial.numel(), float(max_norm), _p(total), _stream())
    return total


def adamw_step(p, g, master, m, v, lr, beta1, beta2, eps, wd, step, clip=None):
    call("mm_adamw_step", dt(p), _p(p), _p(g), _p(master), _p(m), _p(v), p.numel(), float(lr), float(beta1), float(beta2),
         float(eps), float(wd), int(step), _p(clip), _stream())


def adamw_step_split(p, g, lo, m, v, lr, beta1, beta2, eps, wd, step, clip=None):
    """AdamW on (bf16 parameter, int16 remainder) = the fp32 master in two halves (mm_adamw_step_split): 26 B per parameter."""
    assert p.dtype == torch.bfloat16 and g.dtype == torch.bfloat16 and lo.dtype == torch.int16
    call("mm_adamw_step_split", _p(p), _p(g), _p(lo), _p(m), _p(v), p.numel(), float(lr), float(beta1), float(beta2), float(eps), float(wd),
         int(step), _p(clip), _stream())


def master_split(master, p, lo):
    """fp32 master -> p = RNE(master) (bf16) and the int16 remainder lo."""
    call("mm_master_split", _p(master), master.numel(), _p(p), _p(lo), _stream())


def master_join(p, lo):
    """(bf16 parameter, int16 remainder) -> the fp32 master they encode."""
    out = torch.empty(p.numel(), dtype=torch.float32, device=p.device)
    call("mm_master_join", _p(p), _p(lo), p.numel(), _p(out), _stream())
    return out


def cast(src, dtype):
    dst = torch.empty(src.shape, dtype=dtype, device=src.device)
    call("mm_cast", dt(src), _DT[dtype], _p(src), _p(dst), src.numel(), _stream())
    return dst


# ------------------------------------------------------------------------------------------------ CU-masked streams
def cu_mask_words(n_enabled: int, ncu: int = 256, scheme: str = "hash"):
    """Bit mask (list of uint32 words) with `n_enabled` of `ncu` CUs on.  The disabled CUs are spread so that every XCD and shader
    engine loses about the same number whichever way the driver numbers CUs ("hash": the D smallest of (173 i) mod 256, a
    permutation that is even both over i mod 8 and over i // 32; "stride": every (ncu / D)-th)."""
    n_enabled = max(1, min(int(n_enabled), ncu))
    D = ncu - n_enabled
    if scheme == "stride":
        off = {int((k + 0.5) * ncu / D) for k in range(D)} if D else set()
    else:
        off = set(sorted(range(ncu), key=lambda i: ((i * 173) % 256, i))[:D])
    words = [0] * ((ncu + 31) // 32)
    for i in range(ncu):
        if i not in off:
            words[i // 32] |= 1 << (i % 32)
    return words


_masked_streams = {}


def masked_stream(n_enabled: int, scheme: str = "hash", tag: str = ""):
    """A torch stream object over a HIP stream whose kernels run on `n_enabled` CUs only (cached per process; never destroyed: the
    Trainer keeps them for its lifetime)."""
    import ctypes
    ncu = _lib.lib().mm_device_cu_count()
    key = (torch.cuda.current_device(), int(n_enabled), scheme, tag)
    st = _masked_streams.get(key)
    if st is None:
        words = cu_mask_words(n_enabled, ncu, scheme)
        arr = (ctypes.c_uint32 * len(words))(*words)
        out = ctypes.c_void_p()
        call("mm_stream_create_cu_mask", ctypes.cast(arr, ctypes.c_void_p), len(words), ctypes.cast(ctypes.pointer(out), ctypes.c_void_p))
        st = torch.cuda.ExternalStream(out.value)
        _masked_streams[key] = st
    return st


def priority_stream(priority: int, tag: str = ""):
    """A torch stream object over a HIP stream of the given priority (lower number = served first; cached per process)."""
    import ctypes
    key = (torch.cuda.current_device(), "prio", int(priority), tag)
    st = _masked_streams.get(key)
    if st is None:
        out = ctypes.c_void_p()
        call("mm_stream_create_priority", int(priority), ctypes.cast(ctypes.pointer(out), ctypes.c_void_p))
        st = _masked_streams[key] = torch.cuda.ExternalStream(out.value)
    return st


def stream_priority_range():
    import ctypes
    lo, hi = ctypes.c_int(), ctypes.c_int()
    call("mm_stream_priority_range", ctypes.cast(ctypes.pointer(lo), ctypes.c_void_p), ctypes.cast(ctypes.pointer(hi), ctypes.c_void_p))
    return lo.value, hi.value


def cu_probe(n_wg, threads, spin_ticks, stream=None):
    """-> int64 [n_wg, 2] (XCC id, HW_ID register) of a spinning launch on `stream` (default: the current stream)."""
    out = torch.zeros((n_wg, 2), dtype=torch.int32, device="cuda")
    s = stream.cuda_stream if stream is not None else _stream()
    if stream is not None:
        stream.wait_stream(torch.cuda.current_stream())
    call("mm_debug_cu_probe", _p(out), n_wg, threads, int(spin_ticks), s)
    if stream is not None:
        torch.cuda.current_stream().wait_stream(stream)
    return out.cpu().long() & 0xFFFFFFFF
