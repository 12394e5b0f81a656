#!/usr/bin/env python3
"""Attention micro-benchmark at the 8B step's shape (B=4,S=2048,Hq=32,Hkv=8,D=128 causal) + ViT shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimeditron_amd import kernels as K


def run(B, S, Hq, Hkv, D, causal, mask=False):
    W = (Hq + 2 * Hkv) * D
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = torch.randn(B * S, W, device="cuda", generator=g).to(torch.bfloat16)
    q = qkv[:, : Hq * D].view(B, S, Hq, D); k = qkv[:, Hq * D:(Hq + Hkv) * D].view(B, S, Hkv, D); v = qkv[:, (Hq + Hkv) * D:].view(B, S, Hkv, D)
    dqkv = torch.empty_like(qkv)
    dq = dqkv[:, : Hq * D].view(B, S, Hq, D); dk = dqkv[:, Hq * D:(Hq + Hkv) * D].view(B, S, Hkv, D); dv = dqkv[:, (Hq + Hkv) * D:].view(B, S, Hkv, D)
    do = torch.randn(B, S, Hq, D, device="cuda", generator=g).to(torch.bfloat16)
    sc = D ** -0.5
    km = torch.ones(B, S, dtype=torch.long, device="cuda") if mask else None     # all-ones key mask, as the collator sends
    out, lse = K.attn_fwd(q, k, v, km, causal, sc)
    K.attn_bwd(q, k, v, out, do, lse, km, causal, sc, dq, dk, dv)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    it = 5
    ev[0].record()
    for _ in range(it):
        out, lse = K.attn_fwd(q, k, v, km, causal, sc)
    ev[1].record()
    for _ in range(it):
        K.attn_bwd(q, k, v, out, do, lse, km, causal, sc, dq, dk, dv)
    ev[2].record()
    torch.cuda.synchronize()
    f = 4.0 * B * Hq * S * S * D * (0.5 if causal else 1.0)
    tf, tb = ev[0].elapsed_time(ev[1]) / it, ev[1].elapsed_time(ev[2]) / it
    print(f"B={B} S={S} Hq={Hq} Hkv={Hkv} D={D} causal={causal} mask={mask}: fwd {tf:.3f} ms ({f / tf / 1e9:.0f} TF/s)  bwd {tb:.3f} ms ({2.5 * f / tb / 1e9:.0f} TF/s algorithmic)", flush=True)


from multimeditron_amd._lib import lib as _rawlib


class _Checked:
    """mm_set_option with its return code checked (an unknown switch means the library is older than this script)."""
    def mm_set_option(self, name, value):
        rc = _rawlib().mm_set_option(name, value)
        assert rc == 0, (name, value, rc)
        return rc


def lib():
    return _Checked()


if "--ab-pair" in sys.argv:      # dK/dV: one key block per workgroup vs the balanced paired kernel (same process)
    for v in (0, 1, 0, 1):
        lib().mm_set_option(b"attn_dkv_pair", v)
        print("attn_dkv_pair", v)
        run(4, 2048, 32, 8, 128, True)
    run(2, 4096, 32, 8, 128, True)
    sys.exit(0)
if "--ab-fwdwaves" in sys.argv:  # forward: one 8-wave workgroup per CU vs two independent 4-wave workgroups (same process)
    for v in (8, 4, 8, 4):
        lib().mm_set_option(b"attn_fwd_waves", v)
        print("attn_fwd_waves", v)
        run(4, 2048, 32, 8, 128, True)
    run(2, 4096, 32, 8, 128, True)
    lib().mm_set_option(b"attn_fwd_waves", 8)
    run(2, 4096, 32, 8, 128, True)
    sys.exit(0)
if "--ab-pf" in sys.argv:        # D=128 forward: serialized fragment reads vs the prefetching kernel (same process, interleaved)
    for v in (0, 1, 0, 1):
        lib().mm_set_option(b"attn_fwd_pf", v)
        print("attn_fwd_pf", v)
        run(4, 2048, 32, 8, 128, True)
    run(2, 4096, 32, 8, 128, True)
    run(4, 2048, 28, 4, 128, True)
    run(4, 2048, 32, 8, 128, True, mask=True)
    sys.exit(0)
if "--ab-q" in sys.argv:         # D=128 forward: waves of a SIMD in phase (p kernel) vs out of phase (q kernel), interleaved
    B, S, Hq, Hkv, D = 2, 1000, 8, 2, 128            # bit-identity first (ragged length, key mask with holes)
    g = torch.Generator(device="cuda").manual_seed(1)
    q = torch.randn(B, S, Hq, D, device="cuda", generator=g).to(torch.bfloat16)
    k = torch.randn(B, S, Hkv, D, device="cuda", generator=g).to(torch.bfloat16)
    v = torch.randn(B, S, Hkv, D, device="cuda", generator=g).to(torch.bfloat16)
    km = (torch.rand(B, S, device="cuda", generator=g) > 0.1).long()
    for causal in (True, False):
        for mask in (None, km):
            outs = []
            for qv in (0, 1):
                lib().mm_set_option(b"attn_fwd_q", qv)
                o, lse = K.attn_fwd(q, k, v, mask, causal, D ** -0.5)
                outs.append((o.clone(), lse.clone()))
            print("fwd p vs q (row sums taken pairwise in q: last-bit differences):", causal, mask is not None, "max |dO|", float((outs[0][0].float() - outs[1][0].float()).abs().max()), "max |dlse|", float((outs[0][1] - outs[1][1]).abs().max()), flush=True)
    run(4, 2048, 32, 8, 128, True)
    for v_, pr, rd in ((0, 0, 4), (1, 1, 4), (1, 1, 6), (1, 1, 8), (1, 0, 8), (0, 0, 4), (1, 1, 4), (1, 1, 6), (1, 1, 8), (1, 0, 8)):
        lib().mm_set_option(b"attn_fwd_q", v_)
        lib().mm_set_option(b"attn_q_prio", pr)
        lib().mm_set_option(b"attn_q_rd", rd)
        print("attn_fwd_q", v_, "prio", pr, "rd", rd)
        run(4, 2048, 32, 8, 128, True)
    lib().mm_set_option(b"attn_q_prio", 1)
    run(2, 4096, 32, 8, 128, True)
    run(4, 2048, 28, 4, 128, True)
    run(4, 2048, 32, 8, 128, True, mask=True)
    sys.exit(0)
if "--ab-issue" in sys.argv:     # out-of-phase forward: 4 vs 8 waves issuing the K/V DMA (same process, interleaved)
    run(4, 2048, 32, 8, 128, True)
    for nw in (4, 8, 4, 8, 4, 8):
        lib().mm_set_option(b"attn_q_issue", nw)
        print("attn_q_issue", nw)
        run(4, 2048, 32, 8, 128, True)
    run(2, 4096, 32, 8, 128, True)
    run(4, 2048, 32, 8, 128, True, mask=True)
    sys.exit(0)
if "--ab-res" in sys.argv:       # dK/dV: the paired 8-wave kernel vs K / V fragments resident on four waves (attn_dkv_res), same process
    def grads(B, S, Hq, Hkv, causal, mask, seed):
        g = torch.Generator(device="cuda").manual_seed(seed)
        q = torch.randn(B, S, Hq, 128, device="cuda", generator=g).to(torch.bfloat16)
        k = torch.randn(B, S, Hkv, 128, device="cuda", generator=g).to(torch.bfloat16)
        v = torch.randn(B, S, Hkv, 128, device="cuda", generator=g).to(torch.bfloat16)
        do = torch.randn(B, S, Hq, 128, device="cuda", generator=g).to(torch.bfloat16)
        km = None
        if mask:
            km = torch.ones(B, S, dtype=torch.long, device="cuda")
            km[0, S - 37:] = 0
        out, lse = K.attn_fwd(q, k, v, km, causal, 128 ** -0.5)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        K.attn_bwd(q, k, v, out, do, lse, km, causal, 128 ** -0.5, dq, dk, dv)
        return dq.float(), dk.float(), dv.float()
    for (B, S, Hq, Hkv, causal, mask) in ((4, 2048, 32, 8, True, False), (2, 1000, 8, 2, True, True), (1, 333, 4, 4, False, False), (2, 640, 28, 4, True, False), (1, 2048, 2, 2, True, True)):
        lib().mm_set_option(b"attn_dkv_res", 0)
        ref = grads(B, S, Hq, Hkv, causal, mask, 5)
        for mode in (1, 2):              # 1: resident fragments; 2: + items pipelined inside the wave
            lib().mm_set_option(b"attn_dkv_res", mode)
            got = grads(B, S, Hq, Hkv, causal, mask, 5)
            rel = [float((a - b).norm() / (b.norm() + 1e-30)) for a, b in zip(got, ref)]
            print(f"B={B} S={S} Hq={Hq} Hkv={Hkv} causal={causal} mask={mask} attn_dkv_res={mode}: rel L2 of (dq, dk, dv) vs the pair kernel {rel[0]:.2e} {rel[1]:.2e} {rel[2]:.2e}  "
                  f"max |d dk| {float((got[1] - ref[1]).abs().max()):.3e}", flush=True)
            assert rel[0] == 0.0 and rel[1] < 4e-3 and rel[2] < 4e-3, rel
    for v_ in (0, 1, 2, 0, 1, 2, 0, 2):
        lib().mm_set_option(b"attn_dkv_res", v_)
        print("attn_dkv_res", v_)
        run(4, 2048, 32, 8, 128, True)
    lib().mm_set_option(b"attn_dkv_res", 2)
    lib().mm_set_option(b"attn_dkv_rd", 4)
    print("attn_dkv_res 2, attn_dkv_rd 4")
    run(4, 2048, 32, 8, 128, True)
    run(4, 2048, 32, 8, 128, True)
    lib().mm_set_option(b"attn_dkv_rd", 8)
    run(2, 4096, 32, 8, 128, True)
    run(4, 2048, 28, 4, 128, True)
    lib().mm_set_option(b"attn_dkv_res", 0)
    run(2, 4096, 32, 8, 128, True)
    run(4, 2048, 28, 4, 128, True)
    sys.exit(0)
if "--ab-dkv" in sys.argv:       # dK/dV fragment ring: 4 slots vs 8 (same process, interleaved)
    run(4, 2048, 32, 8, 128, True)
    for rd, late in ((4, 0), (8, 0), (8, 1), (4, 0), (8, 0), (8, 1), (8, 0), (8, 1)):
        lib().mm_set_option(b"attn_dkv_rd", rd)
        lib().mm_set_option(b"attn_dkv_late", late)
        print("attn_dkv_rd", rd, "late", late)
        run(4, 2048, 32, 8, 128, True)
    run(2, 4096, 32, 8, 128, True)
    run(4, 2048, 28, 4, 128, True)
    sys.exit(0)
if "--diag-q" in sys.argv:       # timing experiments on the out-of-phase forward (wrong results by design)
    lib().mm_set_option(b"attn_q_prio", 1)
    run(4, 2048, 32, 8, 128, True)
    for d in (0, 1, 2, 4, 8, 16, 4 | 8, 2 | 16, 1 | 16, 1 | 2 | 16, 4 | 8 | 16 | 1, 0):
        lib().mm_set_option(b"attn_diag", d)
        print("diag", d)
        run(4, 2048, 32, 8, 128, True)
    sys.exit(0)
if "--quick" in sys.argv:
    run(4, 2048, 32, 8, 128, True)
    run(4, 2048, 32, 8, 128, True)
    run(4, 2048, 32, 8, 128, True, mask=True)
    run(4, 2048, 32, 8, 128, True, mask=True)
    sys.exit(0)
for nw in (8, 4, 8, 4):
    lib().mm_set_option(b"attn_issue_waves", nw)
    print("attn_issue_waves", nw)
    run(4, 2048, 32, 8, 128, True)
run(4, 2048, 32, 8, 128, True)
run(4, 257, 16, 16, 64, False)
run(2, 4096, 32, 8, 128, True)
