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
