"""Seeded random-shape sweeps of the two MFMA kernels families through the C ABI: every bf16 GEMM variant on ragged M/N/K in
all three layouts, and attention forward+backward on random (B, Sq, Skv, heads, D, causal, mask) incl. Sq != Skv, against
fp32 torch references.  Shapes are drawn once from a fixed seed, so failures reproduce."""
import random

import pytest
import torch
import torch.nn.functional as F  # noqa: F401

from tests.test_kernels_gpu import TOL, attn_ref, rel, rnd, K  # noqa: F401  (K: the kernels fixture)

pytestmark = pytest.mark.gpu
_rng = random.Random(20260131)


def _gemm_cases(n):
    out = []
    for _ in range(n):
        variant = _rng.choice([0, 1, 2, 3, 4, 5, 6])
        M = _rng.choice([1, 3, 17, 64, 129, 255, 256, 257, 511, 700, 1028])
        N = _rng.choice([8, 40, 72, 128, 130, 264, 512, 520, 1000])
        Kd = 8 * _rng.choice([1, 2, 7, 8, 9, 16, 25, 33, 64, 100])
        out.append((variant, _rng.choice(["NT", "NN", "TN"]), M, N, Kd, _rng.random() < 0.4))
    return out


@pytest.mark.parametrize("case", _gemm_cases(48))
def test_gemm_random(K, case):      # noqa: F811
    from multimeditron_amd._lib import lib
    variant, layout, M, N, Kd, with_epi = case
    dtype = torch.bfloat16
    pad8 = lambda n: (n + 7) // 8 * 8
    a, b = rnd((M, Kd), dtype, M * 7 + N), rnd((N, Kd), dtype, N * 3 + Kd)
    ref = a.float() @ b.float().t()

    def padded(x):
        r, c = x.shape
        o = torch.zeros(r, pad8(c), dtype=dtype)
        o[:, :c] = x
        return o.cuda()[:, :c]
    if layout == "NT":
        A, B, lay = padded(a), padded(b), 0
    elif layout == "NN":
        A, B, lay = padded(a), padded(b.t().contiguous()), 1
    else:
        A, B, lay = padded(a.t().contiguous()), padded(b.t().contiguous()), 2
    bias = res = None
    if with_epi:
        bias = rnd((N,), dtype, 5)
        rp = torch.zeros(M, (N + 63) // 64 * 64, dtype=dtype)
        rp[:, :N] = rnd((M, N), dtype, 6)
        res = rp.cuda()[:, :N]
        ref = ref + bias.float() + rp[:, :N].float()
    assert lib().mm_set_option(b"gemm_kernel", variant) == 0
    try:
        out = K.gemm(lay, A, B, M, N, Kd, bias=bias.cuda() if bias is not None else None, residual=res, ldc_pad=True)
        torch.cuda.synchronize()
    finally:
        lib().mm_set_option(b"gemm_kernel", 0)
    assert out.shape == (M, N)
    assert rel(out.float(), ref) < TOL[dtype], case


def _attn_cases(n):
    out = []
    for _ in range(n):
        D = _rng.choice([64, 128])
        Hkv = _rng.choice([1, 2, 3])
        G = _rng.choice([1, 2, 4, 7])
        Skv = _rng.choice([1, 5, 33, 64, 100, 129, 257, 300, 513])
        same = _rng.random() < 0.6
        Sq = Skv if same else _rng.choice([1, 2, 17, min(Skv, 40)])
        out.append((_rng.choice([1, 2, 3]), Sq, Skv, Hkv * G, Hkv, D, _rng.random() < 0.7, _rng.random() < 0.4))
    return out


@pytest.mark.parametrize("case", _attn_cases(40))
def test_attention_random(K, case):      # noqa: F811
    B, Sq, Skv, Hq, Hkv, D, causal, masked = case
    dtype = torch.bfloat16
    q, k, v, do = rnd((B, Sq, Hq, D), dtype, 1), rnd((B, Skv, Hkv, D), dtype, 2), rnd((B, Skv, Hkv, D), dtype, 3), rnd((B, Sq, Hq, D), dtype, 4)
    mask = None
    if masked and Skv > 2:
        mask = torch.ones(B, Skv, dtype=torch.long)
        mask[0, : max(1, Skv // 4)] = 0
    scale = D ** -0.5
    qf, kf, vf = (t.float().clone().requires_grad_(True) for t in (q, k, v))
    ref = attn_ref(qf, kf, vf, mask, causal, scale)
    ref.backward(do.float())
    qd, kd, vd = q.cuda(), k.cuda(), v.cuda()
    mg = mask.cuda() if mask is not None else None
    out, lse = K.attn_fwd(qd, kd, vd, mg, causal, scale)
    dq, dk, dv = torch.full_like(qd, float("nan")), torch.full_like(kd, float("nan")), torch.full_like(vd, float("nan"))
    K.attn_bwd(qd, kd, vd, out, do.cuda(), lse, mg, causal, scale, dq, dk, dv)
    torch.cuda.synchronize()
    # rows that see no key (masked prefix under the causal shift) are defined as 0 here and are garbage in the reference
    valid = torch.ones(B, Sq, dtype=torch.bool)
    shift = Skv - Sq
    for bb in range(B):
        for i in range(Sq):
            hi = min(Skv, i + shift + 1) if causal else Skv
            vis = (mask[bb, :hi].any() if mask is not None else True) if hi > 0 else False
            valid[bb, i] = bool(vis)
    tol = TOL[dtype]
    assert rel(out.float().cpu()[valid], ref.detach()[valid]) < tol, case
    assert torch.isfinite(dq.float()).all() and torch.isfinite(dk.float()).all() and torch.isfinite(dv.float()).all()
    if bool(valid.all()):
        def close(a, b):      # absolute floor: with a single visible key the reference gradient of q and k is exactly zero
            a, b = a.double().cpu(), b.double()
            return float((a - b).norm()) <= 3 * tol * float(b.norm()) + 1e-4
        assert close(dq.float(), qf.grad), case
        assert close(dk.float(), kf.grad), case
        assert close(dv.float(), vf.grad), case


# ---- row-wise kernels on the widths the supported models use (and a few odd ones) ---------------------------------------
def _norm_cases(n):
    Hs = [8, 64, 136, 768, 1024, 1152, 2048, 3584, 4096, 4304, 8192]
    return [(_rng.choice([1, 5, 16, 17, 77, 130, 1028]), _rng.choice(Hs), _rng.choice([torch.bfloat16, torch.float32]),
             _rng.random() < 0.5) for _ in range(n)]


@pytest.mark.parametrize("case", _norm_cases(24))
def test_norms_random(K, case):      # noqa: F811
    M, H, dtype, with_res = case
    x, dy = rnd((M, H), dtype, M + H), rnd((M, H), dtype, M * 3 + H)
    w, b = (1 + 0.1 * rnd((H,), torch.float32, 7)).to(dtype), rnd((H,), dtype, 8)
    dres = rnd((M, H), dtype, 9) if with_res else None
    tol = TOL[dtype]
    add = dres.float() if with_res else 0.0
    xf, wf = x.float().clone().requires_grad_(True), w.float().clone().requires_grad_(True)
    ref = wf * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6))
    ref.backward(dy.float())
    y, rstd = K.rmsnorm_fwd(x.cuda(), w.cuda(), 1e-6)
    assert rel(y.float(), ref.detach()) < tol, case
    dx, dwp = K.rmsnorm_bwd(dy.cuda(), x.cuda(), w.cuda(), rstd, dres.cuda() if with_res else None)
    dw = torch.empty(H, dtype=dtype, device="cuda")
    K.reduce_partials(dwp, dw, False)
    assert rel(dx.float(), xf.grad + add) < tol * 2, case
    assert float((dw.float().cpu() - wf.grad).norm()) <= tol * 3 * float(wf.grad.norm()) + 1e-3, case
    xf, wf, bf = (t.float().clone().requires_grad_(True) for t in (x, w, b))
    ref = F.layer_norm(xf, (H,), wf, bf, 1e-6)
    ref.backward(dy.float())
    y, mean, rstd = K.layernorm_fwd(x.cuda(), w.cuda(), b.cuda(), 1e-6)
    assert rel(y.float(), ref.detach()) < tol, case
    dx, dwp, dbp = K.layernorm_bwd(dy.cuda(), x.cuda(), w.cuda(), mean, rstd, dres.cuda() if with_res else None)
    dw, db = torch.empty(H, dtype=dtype, device="cuda"), torch.empty(H, dtype=dtype, device="cuda")
    K.reduce_partials(dwp, dw, False)
    K.reduce_partials(dbp, db, False)
    assert rel(dx.float(), xf.grad + add) < tol * 2, case
    assert float((dw.float().cpu() - wf.grad).norm()) <= tol * 3 * float(wf.grad.norm()) + 1e-3, case
    assert float((db.float().cpu() - bf.grad).norm()) <= tol * 3 * float(bf.grad.norm()) + 1e-3, case


@pytest.mark.parametrize("case", [(_rng.choice([1, 7, 33, 130]), _rng.choice([8, 136, 4304, 8192, 14336, 18944])) for _ in range(8)])
def test_swiglu_random(K, case):      # noqa: F811
    M, I = case
    dtype = torch.bfloat16
    gu, dout = rnd((M, 2 * I), dtype, M + I), rnd((M, I), dtype, M * I % 1000)
    guf = gu.float().clone().requires_grad_(True)
    ref = F.silu(guf[:, :I]) * guf[:, I:]
    ref.backward(dout.float())
    assert rel(K.swiglu_fwd(gu.cuda(), I).float(), ref.detach()) < TOL[dtype]
    assert rel(K.swiglu_bwd(gu.cuda(), dout.cuda(), I).float(), guf.grad) < TOL[dtype] * 2


@pytest.mark.parametrize("case", [(_rng.choice([1, 9, 40]), _rng.choice([2, 130, 1000, 32000, 128258, 152066]), _rng.choice([torch.bfloat16, torch.float32]))
                                  for _ in range(10)])
def test_cross_entropy_random(K, case):      # noqa: F811
    T, V, dtype = case
    ld = (V + 63) // 64 * 64
    logits = rnd((T, V), dtype, T + V, 2.0)
    labels = torch.randint(0, V, (T,), generator=torch.Generator().manual_seed(V))
    if T > 2:
        labels[::3] = -100
    lf = logits.float().clone().requires_grad_(True)
    kept = int((labels >= 0).sum())
    buf = torch.zeros(T, ld, dtype=dtype, device="cuda")
    buf[:, :V] = logits.cuda()
    lc, lse = K.ce_fwd(buf[:, :V], V, labels.cuda())
    assert int(lc[1]) == kept
    if kept:
        ref = F.cross_entropy(lf, labels, ignore_index=-100)
        ref.backward()
        assert abs(float(lc[0]) - float(ref)) < 1e-4 * max(1, abs(float(ref))), case
        d = torch.full_like(buf, float("nan"))
        K.ce_bwd(buf[:, :V], V, labels.cuda(), lse, lc, None, d[:, :V])
        assert rel(d[:, :V].float(), lf.grad) < (2e-5 if dtype == torch.float32 else 1e-2), case
        assert torch.all(d[:, V:] == 0)
    got = K.argmax_softmax(buf[:, :V], V, 0.1).cpu()
    assert torch.equal(got, torch.argmax(torch.softmax(logits / 0.1, dim=-1), dim=-1)), case
