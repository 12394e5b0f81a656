"""torch.autograd.Function wrappers: each forward/backward is one or a few libmmhip calls.  Activations are
2-D token-major tensors [T, features] throughout (T = batch * seq), which is exactly what the GEMMs consume, so the
path contains no transposes or permutes.

Weight gradients are written straight into `param.grad` by the wgrad GEMM (overwrite on the first contribution of a
step, accumulate afterwards) instead of being returned to autograd: no extra pass over 8 B parameters, and the DP
trainer can point `.grad` at its flat all-reduce buckets.  See `grad_target`."""
from __future__ import annotations

from typing import Callable, Optional

import torch

from . import kernels as K
from ._lib import EPI_GELU_ERF, EPI_QUICK_GELU, GELU_KIND  # noqa: F401

# called as hook(param) right after a parameter's gradient for this step is complete (DP bucket scheduling)
_grad_ready_hook: Optional[Callable] = None


def set_grad_ready_hook(fn: Optional[Callable]):
    global _grad_ready_hook
    _grad_ready_hook = fn


def grad_target(p: torch.Tensor):
    """-> (grad buffer, accumulate?) for writing p's gradient in place."""
    if p.grad is None:
        view = getattr(p, "_mm_grad_view", None)
        if view is None and getattr(p, "_mm_flat", None) is not None:
            p._mm_flat.ensure_grad()          # packed model: gradients live in the flat buffer (fused groups need it)
            view = p._mm_grad_view
        p.grad = view if view is not None else torch.empty_like(p)
        fresh = True
    else:
        fresh = bool(getattr(p, "_mm_fresh", False))
    p._mm_fresh = False
    return p.grad, not fresh


def _ready(p):
    if _grad_ready_hook is not None:
        _grad_ready_hook(p)


# ---- deferred weight gradients ---------------------------------------------------------------------------------------
# The modality backward (ViT-L/14 on 4 images: ~700 short, latency-bound launches) runs at the very end of backward with
# nothing beside it.  Weight-gradient GEMMs have no consumer inside backward, so those of the LAST decoder layers to run
# (layers 0..n-1) are held back and launched on a side stream when backward reaches the embedding splice: they fill the
# CUs the modality backward leaves idle.  Same kernels, same operands, same single write per parameter.
_defer = None


def set_wgrad_deferral(stream, params, immediate=()):
    """stream: side HIP stream for the deferred GEMMs; params: the first Parameter of every deferred group (held weakly);
    immediate: first Parameters of groups whose wgrad GEMM is not held back but launched AT ONCE on the side stream (it then
    runs beside the input-gradient chain of the main stream instead of in front of it).  stream=None disables."""
    import weakref
    global _defer
    _defer = None if stream is None else {"stream": stream, "params": {id(p): weakref.ref(p) for p in params}, "items": [],
                                          "event": None, "now": {id(p): weakref.ref(p) for p in immediate}}


def _is_deferred(p):
    r = _defer["params"].get(id(p)) if _defer is not None else None
    return r is not None and r() is p        # the weak reference guards against id() reuse by a later model


def _wgrad_on_side(dy, x, wg):
    """Launch one wgrad GEMM on the side stream now, ordered after everything the main stream has been given so far."""
    from ._lib import get_option, lib
    d = _defer
    main, side = torch.cuda.current_stream(), d["stream"]
    ev0 = torch.cuda.Event()
    ev0.record(main)
    side.wait_event(ev0)
    persist = get_option("gemm_persist")
    lib().mm_set_option(b"gemm_persist", 0)          # one tile per workgroup: fills the CUs the main stream's tails leave
    try:
        with torch.cuda.stream(side):
            dy.record_stream(side)
            x.record_stream(side)
            g, acc = wg.grad_target()
            K.linear_wgrad(dy, x, g, acc, sumsq=wg.sumsq_slots())
            wg.ready()
            ev = torch.cuda.Event()
            ev.record(side)
    finally:
        lib().mm_set_option(b"gemm_persist", persist)
    d["event"] = ev


def _wgrad(dy, x, wg):
    """The weight-gradient GEMM of one parameter group: now on this stream, now on the side stream, or held back."""
    if not wg.requires_grad:
        return
    if _defer is not None:
        key = id(wg.params[0])
        r = _defer["now"].get(key)
        if r is not None and r() is wg.params[0]:
            _wgrad_on_side(dy, x, wg)
            return
        if _is_deferred(wg.params[0]):
            _defer["items"].append((dy, x, wg))
            return
    g, acc = wg.grad_target()
    K.linear_wgrad(dy, x, g, acc, sumsq=wg.sumsq_slots())
    wg.ready()


def flush_deferred_wgrads():
    """Launch what has been held back (no-op when nothing is pending); returns the event the compute stream must wait for
    before gradients are read (or None)."""
    d = _defer
    if d is None:
        return None
    if d["items"]:
        from ._lib import get_option, lib
        main, side = torch.cuda.current_stream(), d["stream"]
        side.wait_stream(main)
        import os
        persist = get_option("gemm_persist")         # restore what the trainer / the user had set, not a constant
        # one tile per workgroup: shares the chip with the other stream's kernels (MM_DEFER_PERSIST=1: keep the persistent grid)
        lib().mm_set_option(b"gemm_persist", 1 if os.environ.get("MM_DEFER_PERSIST", "0") == "1" else 0)
        try:
            with torch.cuda.stream(side):
                for dy, x, wg in d["items"]:
                    dy.record_stream(side)
                    x.record_stream(side)
                    g, acc = wg.grad_target()
                    K.linear_wgrad(dy, x, g, acc, sumsq=wg.sumsq_slots())
                    wg.ready()
                ev = torch.cuda.Event()
                ev.record(side)
        finally:
            lib().mm_set_option(b"gemm_persist", persist)
        d["items"].clear()
        d["event"] = ev
    ev, d["event"] = d["event"], None
    return ev


class ParamGroup:
    """One or several Parameters that are adjacent views of one flat buffer and act as a single GEMM operand
    (fused q/k/v, fused gate/up).  `.tensor()` is the fused [sum_out, in] weight (or [sum_out] bias)."""

    def __init__(self, params):
        self.params = list(params)

    def _fused(self, get):
        ts = [get(p) for p in self.params]
        if len(ts) == 1:
            return ts[0]
        first = ts[0]
        if any(t is None for t in ts):
            return None
        ptr = first.data_ptr()
        base = first.untyped_storage().data_ptr()
        for t in ts:
            if t.data_ptr() != ptr or not t.is_contiguous() or t.untyped_storage().data_ptr() != base:
                return None
            ptr += t.numel() * t.element_size()
        rows = sum(t.shape[0] for t in ts)
        shape = (rows,) + tuple(first.shape[1:])
        return first.as_strided(shape, first.stride())

    def tensor(self):
        t = self._fused(lambda p: p.data)
        if t is None:  # not packed (e.g. a freshly constructed CPU->GPU model before pack_parameters)
            raise RuntimeError("fused parameter group is not contiguous in memory: call model.pack_parameters()")
        return t

    @property
    def requires_grad(self):
        return any(p.requires_grad for p in self.params)

    def grad_target(self):
        tg = [grad_target(p) for p in self.params]
        if len(tg) == 1:
            return tg[0]
        accs = {a for _, a in tg}
        g = self._fused(lambda p: p.grad)
        if g is None or len(accs) != 1:
            raise RuntimeError("fused parameter group has non-contiguous .grad buffers")
        return g, accs.pop()

    def ready(self):
        for p in self.params:
            _ready(p)

    def sumsq_slots(self):
        """fp32 partial buffer the trainer attached to this group's first parameter (`_mm_ss`): the wgrad GEMM then also
        produces the group's sum of squares for the global gradient norm.  None = not requested."""
        return getattr(self.params[0], "_mm_ss", None)


def as_group(p):
    return p if isinstance(p, ParamGroup) else ParamGroup([p])


# --------------------------------------------------------------------------------------------------- linear
class LinearFn(torch.autograd.Function):
    """y = act(x @ W^T + b) + residual.  x, y 2-D.  W/b are ParamGroups (not autograd inputs); `dummy` is a
    requires-grad scalar that keeps the node in the graph when only the weights need gradients."""

    @staticmethod
    def forward(ctx, x, residual, dummy, wg: ParamGroup, bg: Optional[ParamGroup], act: int, ldc_pad: bool):
        w = wg.tensor()
        b = bg.tensor() if bg is not None else None
        pre = None
        if act and (x.requires_grad or wg.requires_grad):
            fused = K.linear_act_fwd(x, w, b, act, residual)      # one launch: the epilogue writes the pre-activation AND the output
            if fused is not None:
                pre, y = fused
            else:
                pre = K.linear_fwd(x, w, bias=b)                  # keep the pre-activation for backward
                y = K.gelu_fwd(pre, GELU_KIND[act])
                if residual is not None:
                    y = K.add(y, residual)
        else:
            y = K.linear_fwd(x, w, bias=b, residual=residual, act=act, ldc_pad=ldc_pad)
        ctx.wg, ctx.bg, ctx.act = wg, bg, act
        ctx.has_res = residual is not None
        ctx.save_for_backward(x, pre)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, pre = ctx.saved_tensors
        wg, bg = ctx.wg, ctx.bg
        if dy.stride(-1) != 1 or (dy.stride(0) % 8):
            dy = dy.contiguous()
        dres = dy if ctx.has_res else None
        if ctx.act:
            dy = K.gelu_bwd(pre, dy, GELU_KIND[ctx.act])
        if bg is not None and bg.requires_grad:
            g, acc = bg.grad_target()
            K.colsum(dy, g, acc)
            bg.ready()
        _wgrad(dy, x, wg)
        dx = K.linear_dgrad(dy, wg.tensor()) if ctx.needs_input_grad[0] else None
        return dx, dres, None, None, None, None, None


def linear(x, wg, bg=None, residual=None, act=0, ldc_pad=False, dummy=None):
    return LinearFn.apply(x, residual, dummy, as_group(wg), as_group(bg) if bg is not None else None, act, ldc_pad)


# --------------------------------------------------------------------------------------------------- norms
class RMSNormFn(torch.autograd.Function):
    """-> (y, x_res): x_res aliases x and is what the caller feeds to the residual add, so the gradient of the residual
    branch arrives HERE and is added inside the norm-backward kernel (no separate autograd accumulation pass)."""

    @staticmethod
    def forward(ctx, x, dummy, w, eps):
        y, rstd = K.rmsnorm_fwd(x, w.data, eps)
        ctx.w = w
        ctx.save_for_backward(x, rstd)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dres):
        x, rstd = ctx.saved_tensors
        w = ctx.w
        if dy is None:
            return dres, None, None, None
        dx, dwp = K.rmsnorm_bwd(dy.contiguous(), x, w.data, rstd, dres.contiguous() if dres is not None else None)
        if w.requires_grad:
            g, acc = grad_target(w)
            K.reduce_partials(dwp, g, acc)
            _ready(w)
        return dx, None, None, None


def rmsnorm(x, w, eps, dummy=None):
    return RMSNormFn.apply(x, dummy, w, eps)


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dummy, w, b, eps):
        y, mean, rstd = K.layernorm_fwd(x, w.data, b.data, eps)
        ctx.w, ctx.b = w, b
        ctx.save_for_backward(x, mean, rstd)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dres):
        x, mean, rstd = ctx.saved_tensors
        w, b = ctx.w, ctx.b
        if dy is None:
            return dres, None, None, None, None
        dx, dwp, dbp = K.layernorm_bwd(dy.contiguous(), x, w.data, mean, rstd, dres.contiguous() if dres is not None else None)
        if w.requires_grad and b.requires_grad:          # dw and db in one launch
            (gw, aw), (gb, ab) = grad_target(w), grad_target(b)
            K.reduce_partials2(dwp, dbp, gw, gb, aw, ab)
            _ready(w)
            _ready(b)
            return dx, None, None, None, None
        if w.requires_grad:
            g, acc = grad_target(w)
            K.reduce_partials(dwp, g, acc)
            _ready(w)
        if b.requires_grad:
            g, acc = grad_target(b)
            K.reduce_partials(dbp, g, acc)
            _ready(b)
        return dx, None, None, None, None


def layernorm(x, w, b, eps, dummy=None):
    return LayerNormFn.apply(x, dummy, w, b, eps)


# --------------------------------------------------------------------------------------------------- attention
class RopeAttentionFn(torch.autograd.Function):
    """qkv [T, (Hq+2Hkv)*D] (fused projection output, consumed in place) -> out [T, Hq*D].
    RoPE (optional) is applied in place to the q and k heads, then flash attention.  Returns roped k/v views via
    ctx-free attributes for KV caching (see `attention`)."""

    @staticmethod
    def forward(ctx, qkv, cos, sin, key_mask, B, S, Hq, Hkv, D, causal, scale):
        W = (Hq + 2 * Hkv) * D
        assert qkv.shape == (B * S, W) and qkv.is_contiguous()
        if cos is not None:
            K.rope_apply_(qkv, B * S, Hq + Hkv, D, W, cos, sin)
        q = qkv[:, : Hq * D].view(B, S, Hq, D)
        k = qkv[:, Hq * D:(Hq + Hkv) * D].view(B, S, Hkv, D)
        v = qkv[:, (Hq + Hkv) * D:].view(B, S, Hkv, D)
        out, lse = K.attn_fwd(q, k, v, key_mask, causal, scale)
        ctx.dims = (B, S, Hq, Hkv, D, causal, scale)
        ctx.save_for_backward(qkv, out, lse, cos, sin, key_mask)
        return out.view(B * S, Hq * D)

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse, cos, sin, key_mask = ctx.saved_tensors
        B, S, Hq, Hkv, D, causal, scale = ctx.dims
        W = (Hq + 2 * Hkv) * D
        q = qkv[:, : Hq * D].view(B, S, Hq, D)
        k = qkv[:, Hq * D:(Hq + Hkv) * D].view(B, S, Hkv, D)
        v = qkv[:, (Hq + Hkv) * D:].view(B, S, Hkv, D)
        dqkv = torch.zeros_like(qkv) if qkv.dtype == torch.float32 else torch.empty_like(qkv)
        dq = dqkv[:, : Hq * D].view(B, S, Hq, D)
        dk = dqkv[:, Hq * D:(Hq + Hkv) * D].view(B, S, Hkv, D)
        dv = dqkv[:, (Hq + Hkv) * D:].view(B, S, Hkv, D)
        K.attn_bwd(q, k, v, out, dout.contiguous().view(B, S, Hq, D), lse, key_mask, causal, scale, dq, dk, dv)
        if cos is not None:
            K.rope_apply_(dqkv, B * S, Hq + Hkv, D, W, cos, sin, inverse=True)
        return (dqkv,) + (None,) * 10


def rope_attention(qkv, cos, sin, key_mask, B, S, Hq, Hkv, D, causal, scale):
    return RopeAttentionFn.apply(qkv, cos, sin, key_mask, B, S, Hq, Hkv, D, causal, scale)


class QKVRopeAttentionFn(torch.autograd.Function):
    """The decoder's attention front half as ONE node: fused q|k|v projection with RoPE in its epilogue (mm_gemm_rope_fwd; the
    projection + mm_rope_apply when a shape does not qualify: same bits), then flash attention.  h [T, H] -> out [T, Hq*D].
    Backward = attention backward, inverse RoPE on d(q|k) in place, then the projection's wgrad / bias / dgrad (LinearFn's)."""

    @staticmethod
    def forward(ctx, h, dummy, wg: ParamGroup, bg: Optional[ParamGroup], cos, sin, key_mask, B, S, Hq, Hkv, D, causal, scale):
        w = wg.tensor()
        b = bg.tensor() if bg is not None else None
        W = (Hq + 2 * Hkv) * D
        qkv = K.gemm_rope_fwd(h, w, b, (Hq + Hkv) * D, D, cos, sin) if cos is not None else None
        if qkv is None:
            qkv = K.linear_fwd(h, w, bias=b)
            if cos is not None:
                K.rope_apply_(qkv, B * S, Hq + Hkv, D, W, cos, sin)
        q = qkv[:, : Hq * D].view(B, S, Hq, D)
        k = qkv[:, Hq * D:(Hq + Hkv) * D].view(B, S, Hkv, D)
        v = qkv[:, (Hq + Hkv) * D:].view(B, S, Hkv, D)
        out, lse = K.attn_fwd(q, k, v, key_mask, causal, scale)
        ctx.dims = (B, S, Hq, Hkv, D, causal, scale)
        ctx.wg, ctx.bg = wg, bg
        ctx.save_for_backward(h, qkv, out, lse, cos, sin, key_mask)
        return out.view(B * S, Hq * D)

    @staticmethod
    def backward(ctx, dout):
        h, qkv, out, lse, cos, sin, key_mask = ctx.saved_tensors
        B, S, Hq, Hkv, D, causal, scale = ctx.dims
        wg, bg = ctx.wg, ctx.bg
        W = (Hq + 2 * Hkv) * D
        q = qkv[:, : Hq * D].view(B, S, Hq, D)
        k = qkv[:, Hq * D:(Hq + Hkv) * D].view(B, S, Hkv, D)
        v = qkv[:, (Hq + Hkv) * D:].view(B, S, Hkv, D)
        dqkv = torch.zeros_like(qkv) if qkv.dtype == torch.float32 else torch.empty_like(qkv)
        dq = dqkv[:, : Hq * D].view(B, S, Hq, D)
        dk = dqkv[:, Hq * D:(Hq + Hkv) * D].view(B, S, Hkv, D)
        dv = dqkv[:, (Hq + Hkv) * D:].view(B, S, Hkv, D)
        K.attn_bwd(q, k, v, out, dout.contiguous().view(B, S, Hq, D), lse, key_mask, causal, scale, dq, dk, dv)
        if cos is not None:
            K.rope_apply_(dqkv, B * S, Hq + Hkv, D, W, cos, sin, inverse=True)
        if bg is not None and bg.requires_grad:
            g, acc = bg.grad_target()
            K.colsum(dqkv, g, acc)
            bg.ready()
        _wgrad(dqkv, h, wg)
        dh = K.linear_dgrad(dqkv, wg.tensor()) if ctx.needs_input_grad[0] else None
        return (dh,) + (None,) * 13


def qkv_rope_attention(h, wg, bg, cos, sin, key_mask, B, S, Hq, Hkv, D, causal, scale, dummy=None):
    return QKVRopeAttentionFn.apply(h, dummy, as_group(wg), as_group(bg) if bg is not None else None, cos, sin, key_mask, B, S, Hq, Hkv, D,
                                    causal, scale)


class HeadPadFn(torch.autograd.Function):
    """[rows, nheads*d] <-> [rows, nheads*dpad]: zero-pad (or strip) every head so that a head width the MFMA attention
    does not tile (SigLIP-so400m: 72) runs on the D = 64 / 128 kernels.  Zero columns add nothing to q.k and give zero
    columns of p.v, so the result is exact; the backward of a pad is a strip and vice versa."""

    @staticmethod
    def forward(ctx, x, nheads, d, dpad, inverse):
        ctx.meta = (nheads, d, dpad, inverse)
        return K.head_pad(x.contiguous(), nheads, d, dpad, inverse)

    @staticmethod
    def backward(ctx, dy):
        nheads, d, dpad, inverse = ctx.meta
        return K.head_pad(dy.contiguous(), nheads, d, dpad, not inverse), None, None, None, None


def head_pad(x, nheads, d, dpad):
    return HeadPadFn.apply(x, nheads, d, dpad, False)


def head_strip(x, nheads, d, dpad):
    return HeadPadFn.apply(x, nheads, d, dpad, True)


def attention_head_width(d: int, dtype) -> int:
    """Head width the attention kernels run a head of width d at (bf16 MFMA path: 64 or 128; fp32 path: any)."""
    if dtype == torch.float32 or d in (64, 128):
        return d
    if d > 128:
        raise ValueError(f"head_dim {d} > 128 is not supported by the bf16 attention kernels")
    return 64 if d < 64 else 128


# Philox streams of the dropout kernels: key = torch's seed (torch.manual_seed governs it), one counter offset per dropout site
# and call; backward regenerates a mask from the (seed, offset) its forward drew.
_dropout_calls = 0


def next_dropout_stream():
    global _dropout_calls
    _dropout_calls += 1
    return int(torch.initial_seed()) & 0x7FFFFFFFFFFFFFFF, _dropout_calls


class CrossAttentionFn(torch.autograd.Function):
    """Non-causal attention of Nq queries over Nkv keys/values with separate inputs (model/attention.py:79-96 between the
    projections): q2d [n*Nq, h*d], kv2d [n*Nkv, 2*h*d] (k | v, one fused projection output) -> [n*Nq, h*d].
    drop_p > 0: dropout on the attention probabilities (attention.py:40,91).  Up to 512 keys the one-pass kernel mm_xattn_* runs
    (any head width up to 512: the reference's recipes use 96 and 512); beyond that the flash kernels (d in {64, 128}, no dropout)."""

    @staticmethod
    def forward(ctx, q2d, kv2d, n, Nq, Nkv, heads, d, scale, drop_p):
        C = heads * d
        q = q2d.view(n, Nq, heads, d)
        k = kv2d[:, :C].view(n, Nkv, heads, d)
        v = kv2d[:, C:].view(n, Nkv, heads, d)
        small = K.xattn_supported(q2d.dtype, Nkv, d)
        seed = off = 0
        if small:
            if drop_p > 0.0:
                seed, off = next_dropout_stream()
            out, lse = K.xattn_fwd(q, k, v, scale, drop_p, seed, off)
        else:
            if drop_p > 0.0:
                raise NotImplementedError(f"attention dropout over {Nkv} keys: mm_xattn_* holds a query's scores in registers (<= 512 keys)")
            out, lse = K.attn_fwd(q, k, v, None, False, scale)
        ctx.dims = (n, Nq, Nkv, heads, d, scale, drop_p, seed, off, small)
        ctx.save_for_backward(q2d, kv2d, out, lse)
        return out.view(n * Nq, C)

    @staticmethod
    def backward(ctx, dout):
        q2d, kv2d, out, lse = ctx.saved_tensors
        n, Nq, Nkv, heads, d, scale, drop_p, seed, off, small = ctx.dims
        C = heads * d
        mk = torch.zeros_like if (q2d.dtype == torch.float32 and not small) else torch.empty_like
        dq2d, dkv2d = mk(q2d), mk(kv2d)
        args = (q2d.view(n, Nq, heads, d), kv2d[:, :C].view(n, Nkv, heads, d), kv2d[:, C:].view(n, Nkv, heads, d), out,
                dout.contiguous().view(n, Nq, heads, d), lse)
        grads = (dq2d.view(n, Nq, heads, d), dkv2d[:, :C].view(n, Nkv, heads, d), dkv2d[:, C:].view(n, Nkv, heads, d))
        if small:
            K.xattn_bwd(*args, scale, drop_p, seed, off, *grads)
        else:
            K.attn_bwd(*args, None, False, scale, *grads)
        return dq2d, dkv2d, None, None, None, None, None, None, None


def cross_attention(q2d, kv2d, n, Nq, Nkv, heads, d, scale, drop_p=0.0):
    return CrossAttentionFn.apply(q2d, kv2d, n, Nq, Nkv, heads, d, scale, float(drop_p))


class DropoutFn(torch.autograd.Function):
    """nn.Dropout (attention.py:41,98 `proj_drop`): y = x * keep / (1 - p); the mask is regenerated in backward."""

    @staticmethod
    def forward(ctx, x, p):
        ctx.stream = (p,) + next_dropout_stream()
        return K.dropout(x, *ctx.stream)

    @staticmethod
    def backward(ctx, dy):
        return K.dropout(dy, *ctx.stream), None


def dropout(x, p, training=True):
    return DropoutFn.apply(x, float(p)) if (training and p > 0.0) else x


class ExpertFuseFn(torch.autograd.Function):
    """Gating-weighted fusion of stacked expert features x [E, n, P, C] (image_modality_moe.py:163-205); `gate` [n, E] fp32 is
    the gating network's output and receives no gradient here (the gate is not part of this build)."""

    @staticmethod
    def forward(ctx, x, gate, idx, mode):
        E, n, P, C = x.shape
        ctx.meta = (tuple(idx), mode, E, P, C)
        ctx.save_for_backward(gate)
        out = K.expert_fuse(x.reshape(E, n, P * C), gate, idx, mode)
        return out.view(n, P, C) if mode == 0 else out.view(n, len(idx) * P, C)

    @staticmethod
    def backward(ctx, dout):
        (gate,) = ctx.saved_tensors
        idx, mode, E, P, C = ctx.meta
        n = dout.shape[0]
        d = dout.contiguous().view(n, P * C) if mode == 0 else dout.contiguous().view(n, len(idx), P * C)
        dx = K.expert_fuse(d, gate, idx, mode, backward=True, E=E)
        return dx.view(E, n, P, C), None, None, None


def expert_fuse(x, gate, idx, mode):
    return ExpertFuseFn.apply(x, gate, list(idx), mode)


# --------------------------------------------------------------------------------------------------- activations
class SwiGLUFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gu, I):
        ctx.I = I
        ctx.save_for_backward(gu)
        return K.swiglu_fwd(gu, I)

    @staticmethod
    def backward(ctx, dout):
        (gu,) = ctx.saved_tensors
        return K.swiglu_bwd(gu, dout.contiguous(), ctx.I), None


def swiglu(gu, I):
    return SwiGLUFn.apply(gu, I)


class SwiGLUMLPFn(torch.autograd.Function):
    """y = down_proj(silu(gate_proj(x)) * up_proj(x)) + residual as ONE autograd node (HF:llama:163-176 + the residual add of
    :317).  Forward: the SwiGLU is the epilogue of the fused gate|up GEMM (mm_gemm_swiglu_fwd), the residual the epilogue of
    down_proj.  Backward: d(gate|up) comes straight out of down_proj's dgrad GEMM (mm_gemm_swiglu_bwd); d(act) never exists
    in HBM.  Falls back to the separate swiglu kernels when a shape does not qualify (same arithmetic, bit-identical)."""

    @staticmethod
    def forward(ctx, x, residual, dummy, wgu: ParamGroup, wd: ParamGroup, I: int):
        fused = K.gemm_swiglu_fwd(x, wgu.tensor(), I)
        if fused is None:
            gu = K.linear_fwd(x, wgu.tensor())
            act = K.swiglu_fwd(gu, I)
        else:
            gu, act = fused
        y = K.linear_fwd(act, wd.tensor(), residual=residual)
        ctx.wgu, ctx.wd, ctx.I = wgu, wd, I
        ctx.has_res = residual is not None
        ctx.save_for_backward(x, gu, act)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gu, act = ctx.saved_tensors
        wgu, wd, I = ctx.wgu, ctx.wd, ctx.I
        if dy.stride(-1) != 1 or (dy.stride(0) % 8):
            dy = dy.contiguous()
        dres = dy if ctx.has_res else None

        wgrad = _wgrad
        wgrad(dy, act, wd)
        dgu = K.gemm_swiglu_bwd(dy, wd.tensor(), gu, I)
        if dgu is None:
            dgu = K.swiglu_bwd(gu, K.linear_dgrad(dy, wd.tensor()), I)
        wgrad(dgu, x, wgu)
        dx = K.linear_dgrad(dgu, wgu.tensor()) if ctx.needs_input_grad[0] else None
        return dx, dres, None, None, None, None


def swiglu_mlp(x, wgu, wd, I, residual=None, dummy=None):
    return SwiGLUMLPFn.apply(x, residual, dummy, as_group(wgu), as_group(wd), I)


class AddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        return K.add(a, b)

    @staticmethod
    def backward(ctx, d):
        return d, d


def add(a, b):
    return AddFn.apply(a, b)


# --------------------------------------------------------------------------------------------------- embed + splice
class EmbedSpliceFn(torch.autograd.Function):
    """model.py:433-444 in one pass: token embedding gather with modality rows spliced in."""

    @staticmethod
    def forward(ctx, proj, dummy, emb, ids, batch_idx, token_range, B, S):
        T = B * S
        ids = ids.reshape(-1).contiguous()
        src_map = None
        if proj is not None:
            src_map = K.splice_build_map(batch_idx, token_range, S, T)
            proj = proj.reshape(-1, proj.shape[-1])
            if proj.dtype != emb.dtype:   # model.py:443-444 casts the modality rows to the embedding dtype
                proj = K.cast(proj, emb.dtype)
            proj = proj.contiguous()
        out = K.embed_splice_fwd(emb.data, ids, proj, src_map)
        order = skey = None
        if emb.requires_grad and torch.is_grad_enabled():
            # the embedding gradient's summation order (a stable sort of the tokens by id) depends on ids and the splice
            # map only: built here, beside the forward, instead of at the tail of backward
            order, skey = K.embed_sort(ids, src_map, emb.shape[0], emb.shape[1])
        ctx.emb = emb
        ctx.S = S
        ctx.has_proj = proj is not None
        ctx.proj_rows = proj.shape[0] if proj is not None else 0
        ctx.save_for_backward(ids, src_map, batch_idx, token_range, order, skey)
        return out

    @staticmethod
    def backward(ctx, dE):
        ids, src_map, batch_idx, token_range, order, skey = ctx.saved_tensors
        emb = ctx.emb
        dE = dE.contiguous()
        if _defer is not None and _defer["items"]:      # the decoder's backward is done: release the held-back wgrads
            _defer["event"] = flush_deferred_wgrads()
        dproj = None
        if ctx.has_proj and ctx.needs_input_grad[0]:
            dproj = torch.empty((ctx.proj_rows, dE.shape[1]), dtype=dE.dtype, device=dE.device)
        demb = None
        acc = False
        if emb.requires_grad:
            demb, acc = grad_target(emb)
            if order is None:          # forward ran under no_grad bookkeeping (e.g. requires_grad flipped afterwards)
                order, skey = K.embed_sort(ids, src_map, emb.shape[0], emb.shape[1])
            if not acc:
                demb.zero_()           # rows no token touched must read 0 (the flat gradient buffer is never memset)
        K.embed_splice_bwd(dE, ids, src_map, batch_idx, token_range, ctx.S, dproj, demb, order, skey, accumulate=acc)
        if emb.requires_grad:
            _ready(emb)
        return dproj, None, None, None, None, None, None, None


def embed_splice(emb, ids, proj, batch_idx, token_range, B, S, dummy=None):
    return EmbedSpliceFn.apply(proj, dummy, emb, ids, batch_idx, token_range, B, S)


# --------------------------------------------------------------------------------------------------- ViT glue
class PatchEmbedFn(torch.autograd.Function):
    """conv(k=s=patch) as patchify + GEMM, then positions.  CLIP (HF:clip:138-218): no conv bias, CLS row prepended.
    SigLIP (HF:siglip SiglipVisionEmbeddings): conv bias, no CLS row (cls is None)."""

    @staticmethod
    def forward(ctx, pixels, dummy, w_conv, b_conv, cls, pos, ps):
        n = pixels.shape[0]
        Dv = w_conv.shape[0]
        kk = 3 * ps * ps
        kpad = (kk + 63) // 64 * 64
        dtype = w_conv.dtype
        patches = K.patchify(pixels.contiguous(), ps, kpad, dtype)
        wp = torch.zeros((Dv, kpad), dtype=dtype, device=w_conv.device)
        wp[:, :kk].copy_(w_conv.data.reshape(Dv, kk))
        po = K.linear_fwd(patches, wp, bias=b_conv.data if b_conv is not None else None)
        P = patches.shape[0] // n
        if cls is not None:
            x = K.vit_embed_fwd(po, cls.data, pos.data, n, P)
            T = P + 1
        else:
            x = K.bcast_add(po.view(n, P * Dv), pos.data.reshape(-1))
            T = P
        ctx.params = (w_conv, b_conv, cls, pos)
        ctx.meta = (n, P, Dv, kk, kpad)
        ctx.save_for_backward(patches)
        return x.view(n * T, Dv)

    @staticmethod
    def backward(ctx, dx):
        (patches,) = ctx.saved_tensors
        w_conv, b_conv, cls, pos = ctx.params
        n, P, Dv, kk, kpad = ctx.meta
        need_w = w_conv.requires_grad
        need_b = b_conv is not None and b_conv.requires_grad
        gc = gp = None
        accc = accp = False
        if cls is not None and cls.requires_grad:
            gc, accc = grad_target(cls)
        if pos.requires_grad:
            gp, accp = grad_target(pos)
        if not (need_w or need_b or gc is not None or gp is not None):
            return (None,) * 7
        if cls is not None:
            dx = dx.contiguous().view(n, P + 1, Dv)
            if gc is not None and gp is not None and accc != accp:   # kernel takes one flag: normalise by zeroing
                (gc if not accc else gp).zero_()
                accc = accp = True
            dpatch = K.vit_embed_bwd(dx, gc, gp, accc or accp, want_dpatch=need_w or need_b)
        else:
            dpatch = dx.contiguous().view(n * P, Dv)                 # no CLS row: the patch gradient IS dx
            if gp is not None:
                K.colsum(dpatch.view(n, P * Dv), gp.view(-1), accp)
        if need_b:
            g, acc = grad_target(b_conv)
            K.colsum(dpatch, g, acc)
            _ready(b_conv)
        if need_w:
            gw = torch.empty((Dv, kpad), dtype=dx.dtype, device=dx.device)
            K.linear_wgrad(dpatch, patches, gw, False)
            g, acc = grad_target(w_conv)
            if acc:
                g.add_(gw[:, :kk].reshape(g.shape))
            else:
                g.copy_(gw[:, :kk].reshape(g.shape))
            _ready(w_conv)
        for p in (cls, pos):
            if p is not None and p.requires_grad:
                _ready(p)
        return (None,) * 7


def patch_embed(pixels, w_conv, cls, pos, ps, dummy=None, b_conv=None):
    return PatchEmbedFn.apply(pixels, dummy, w_conv, b_conv, cls, pos, ps)


class DropClsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, n, T):
        return K.drop_cls_fwd(x.view(n, T, -1))

    @staticmethod
    def backward(ctx, dy):
        d = K.drop_cls_bwd(dy.contiguous())
        return d.view(-1, d.shape[-1]), None, None


def drop_cls(x, n, T):
    return DropClsFn.apply(x, n, T)


# --------------------------------------------------------------------------------------------------- loss
class CausalLMLossFn(torch.autograd.Function):
    """HF:loss/loss_utils.py:36-71 on the (padded-stride) logits: mean CE over labels != -100, labels pre-shifted."""

    @staticmethod
    def forward(ctx, logits2d, V, shift_labels):
        lc, lse = K.ce_fwd(logits2d, V, shift_labels)
        ctx.V = V
        ctx.save_for_backward(logits2d, shift_labels, lse, lc)
        return lc[0].clone()

    @staticmethod
    def backward(ctx, dloss):
        logits2d, labels, lse, lc = ctx.saved_tensors
        T = logits2d.shape[0]
        ld = logits2d.stride(0)
        buf = torch.empty((T, ld), dtype=logits2d.dtype, device=logits2d.device)
        d = buf[:, : logits2d.shape[1]]
        g = dloss.reshape(1).to(torch.float32).contiguous()
        K.ce_bwd(logits2d, ctx.V, labels, lse, lc, g, d)
        return d, None, None


class LossRows:
    """The rows of a [B*S] batch whose SHIFTED label is not -100, as the host knows them (built where the labels still are a
    host tensor: train/prefetch.py, or CausalLM.forward for host labels): `idx` int32 [n] ascending, `inv` int32 [B*S] (-1 at
    ignored rows), `labels` int64 [n] (the shifted labels of those rows), all on the device."""
    __slots__ = ("idx", "inv", "labels", "n", "total")

    def __init__(self, idx, inv, labels, n, total):
        self.idx, self.inv, self.labels, self.n, self.total = idx, inv, labels, int(n), int(total)

    @staticmethod
    def host_parts(labels_host: torch.Tensor):
        """-> (idx int32 [n], inv int32 [B*S], shifted labels int64 [n]) on the host; HF:loss/loss_utils.py:52-56 shift."""
        shift = torch.nn.functional.pad(labels_host, (0, 1), value=-100)[..., 1:].reshape(-1)
        keep = shift != -100
        idx = torch.nonzero(keep).reshape(-1).to(torch.int32)
        inv = torch.full((shift.numel(),), -1, dtype=torch.int32)
        inv[keep] = torch.arange(idx.numel(), dtype=torch.int32)
        return idx, inv, shift[keep].contiguous()

    @classmethod
    def from_host_labels(cls, labels_host, device):
        idx, inv, lab = cls.host_parts(labels_host)
        return cls(idx.to(device), inv.to(device), lab.to(device), idx.numel(), inv.numel())


class RowsSelectFn(torch.autograd.Function):
    """y = x[rows.idx]; backward scatters dy back into a [B*S, H] gradient with exact zeros at the ignored rows."""

    @staticmethod
    def forward(ctx, x2d, rows):
        ctx.rows = rows
        return K.rows_select(x2d, rows.idx, rows.n)

    @staticmethod
    def backward(ctx, dy):
        rows = ctx.rows
        return K.rows_select(dy.contiguous(), rows.inv, rows.total), None


def rows_select(x2d, rows):
    return RowsSelectFn.apply(x2d, rows)


def causal_lm_loss(logits2d, V, shift_labels):
    return CausalLMLossFn.apply(logits2d, V, shift_labels)
