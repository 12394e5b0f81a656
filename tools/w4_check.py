#!/usr/bin/env python3
"""The 4-wave hand-scheduled GEMM (gemm_bf16_w4_kernel, mm_set_option gemm_w4 = 1) against the 8-wave kernel (gemm_w4 = 0) on the
GPU box: BIT identity on plain / residual / accumulate / fused SwiGLU / SwiGLU-backward / RoPE shapes with ragged M and N, several
repetitions (a missing wait shows as rare wrong tiles), then interleaved timing on the step's NT / NN shapes.

    python tools/w4_check.py [--quick] [--time-only] [--check-only]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimeditron_amd import kernels as K
from multimeditron_amd._lib import lib

NT, NN, TN = 0, 1, 2
LN = {0: 'NT', 1: 'NN', 2: 'TN'}


def set_opt(name, v):
    assert lib().mm_set_option(name.encode(), v) == 0


def rnd(g, *s):
    return (torch.rand(*s, device="cuda", generator=g) * 2 - 1).to(torch.bfloat16)


def pad64(n):
    return (n + 63) // 64 * 64


def both(fn):
    out = []
    for mode in (0, 1):
        set_opt("gemm_w4", mode)
        out.append(fn())
    set_opt("gemm_w4", 1)
    torch.cuda.synchronize()
    return out


def same(a, b):
    if isinstance(a, (tuple, list)):
        return all(same(x, y) for x, y in zip(a, b))
    return torch.equal(a, b)


def check():
    g = torch.Generator(device="cuda").manual_seed(3)
    bad = 0
    set_opt("gemm_kernel", 3)             # every case on the 256x256 tile (small problems would pick a smaller one)
    cases = [(NT, 8192, 4096, 4096), (NT, 5112, 6144, 4096), (NT, 1000, 1544, 320), (NT, 2049, 4104, 1024), (NT, 8192, 4096, 14336),
             (NN, 8192, 4096, 6144), (NN, 5112, 4096, 4096), (NN, 1111, 1032, 448), (NN, 4096, 14336, 4096), (NN, 300, 520, 192),
             (TN, 4096, 4096, 8192), (TN, 6144, 4096, 8192), (TN, 1000, 1544, 320), (TN, 4096, 14336, 8192), (TN, 777, 2056, 1024),
             (TN, 4096, 4096, 5112), (TN, 6144, 4096, 5112), (TN, 520, 300, 203), (TN, 2048, 4104, 1001),
             (NT, 1000, 1544, 328), (NT, 2049, 4104, 1000), (NT, 4096, 4096, 4104), (NN, 300, 520, 200), (NN, 1111, 1032, 456), (NN, 5112, 4096, 128258)]
    for lay, M, N, Kd in cases:
        if lay != TN and Kd % 64:         # ragged K on a K-contiguous operand: what lies beyond K in a row must not be read (NaN x 0 = NaN)
            abuf = torch.full((M, pad64(Kd)), float("nan"), device="cuda", dtype=torch.bfloat16)
            abuf[:, :(Kd + 7) // 8 * 8] = 0          # (operands are read in 16-byte chunks: K % 8 != 0 needs zeros up to the chunk's end,
            abuf[:, :Kd] = rnd(g, M, Kd)             # as mm_ce_bwd leaves them for the lm_head gradient)
            a = abuf[:, :Kd]
        else:
            a = rnd(g, M, Kd) if lay != TN else rnd(g, Kd, pad64(M))[:, :M]
        if lay == NT and Kd % 64:
            bbuf = torch.full((N, pad64(Kd)), float("nan"), device="cuda", dtype=torch.bfloat16)
            bbuf[:, :(Kd + 7) // 8 * 8] = 0
            bbuf[:, :Kd] = rnd(g, N, Kd)
            b = bbuf[:, :Kd]
        else:
            b = rnd(g, N, Kd) if lay == NT else rnd(g, Kd, pad64(N))[:, :N]
        res = rnd(g, M, N)
        for kind in ("plain", "residual", "accumulate"):
            def run():
                c = torch.zeros(M, pad64(N), device="cuda", dtype=torch.bfloat16)[:, :N]
                if kind == "accumulate":
                    c.copy_(res)
                K.gemm(lay, a, b, M, N, Kd, out=c, residual=res if kind == "residual" else None, accumulate=kind == "accumulate")
                return c.clone()
            for rep in range(3):
                o0, o1 = both(run)
                ok = same(o0, o1)
                if not ok:
                    d = (o0.float() - o1.float()).abs()
                    rows = (d.amax(1) > 0).nonzero().flatten()
                    cols = (d.amax(0) > 0).nonzero().flatten()
                    print(f"  MISMATCH max {float(d.max()):.4g} at rows {rows[:4].tolist()}..{rows[-1:].tolist()} ({rows.numel()}) cols {cols[:4].tolist()}..{cols[-1:].tolist()} ({cols.numel()})")
                    bad += 1
                    break
            ref = (a.float() if lay != TN else a.float().t()) @ (b.float().t() if lay == NT else b.float())
            if kind != "plain":
                ref = ref + res.float()
            err = float((o1.float() - ref).norm() / ref.norm())
            print(f"{LN[lay]} M={M} N={N} K={Kd} {kind}: {'bit-identical' if ok else 'DIFFERENT'}  rel-L2 vs fp32 {err:.2e}", flush=True)
    # fused gate|up + SwiGLU (NT), SwiGLU backward on the down_proj dgrad (NN), q|k|v + RoPE (NT)
    for M, I, H in ((8192, 14336, 4096), (1000, 1024, 512), (5112, 2048, 4096)):
        x, wgu = rnd(g, M, H), rnd(g, 2 * I, H) * 0.05
        o0, o1 = both(lambda: tuple(t.clone() for t in K.gemm_swiglu_fwd(x, wgu, I)))
        ok = same(o0, o1)
        bad += not ok
        print(f"swiglu_fwd M={M} I={I} K={H}: {'bit-identical' if ok else 'DIFFERENT'}", flush=True)
        dy, wd, gu = rnd(g, M, H), rnd(g, H, I) * 0.05, o0[0]
        o0, o1 = both(lambda: K.gemm_swiglu_bwd(dy, wd, gu, I).clone())
        ok = same(o0, o1)
        bad += not ok
        print(f"swiglu_bwd M={M} I={I} H={H}: {'bit-identical' if ok else 'DIFFERENT'}", flush=True)
    for M, Hq, Hkv, H, bias in ((8192, 32, 8, 4096, False), (1000, 4, 2, 512, True), (5112, 28, 4, 3584, True)):
        N = (Hq + 2 * Hkv) * 128
        x, w = rnd(g, M, H), rnd(g, N, H) * 0.05
        bv = rnd(g, N) if bias else None
        cos = torch.rand(M, 64, device="cuda", generator=g)
        sin = torch.rand(M, 64, device="cuda", generator=g)
        o0, o1 = both(lambda: K.gemm_rope_fwd(x, w, bv, (Hq + Hkv) * 128, 128, cos, sin).clone())
        ok = same(o0, o1)
        bad += not ok
        print(f"rope_fwd M={M} N={N} K={H} bias={bias}: {'bit-identical' if ok else 'DIFFERENT'}", flush=True)
    set_opt("gemm_kernel", 0)
    print("CHECK", "FAILED" if bad else "OK", flush=True)
    return bad


def timing_scheds(scheds, quick):
    """interleaved timing of several schedules of the 4-wave kernel (gen_gemm_w4.py SCHEDS; 0 = the 8-wave kernel)"""
    T = 8192
    shapes = [(NT, T, 4096, 4096), (NT, T, 28672, 4096), (NT, T, 4096, 14336), (NN, T, 4096, 4096), (NN, T, 14336, 4096), (NN, T, 4096, 28672),
              (TN, 4096, 4096, T), (TN, 28672, 4096, T), (TN, 4096, 14336, T)]
    if quick:
        shapes = [shapes[0], shapes[3], shapes[6]]
    g = torch.Generator(device="cuda").manual_seed(0)
    tot = {m: 0.0 for m in scheds}
    for lay, M, N, Kd in shapes:
        a = rnd(g, M, Kd) if lay != TN else rnd(g, Kd, M)
        b = rnd(g, N, Kd) if lay == NT else rnd(g, Kd, N)
        c = torch.empty(M, pad64(N), device="cuda", dtype=torch.bfloat16)[:, :N]
        res = {m: [] for m in scheds}
        set_opt("gemm_w4", 0)
        ref = K.gemm(lay, a, b, M, N, Kd, out=c).clone()
        for mode in scheds:               # product schedules must give the 8-wave kernel's bits (diag ones, >= 100, are wrong by design)
            if 0 < mode < 100:
                set_opt("gemm_w4", mode)
                c.zero_()
                K.gemm(lay, a, b, M, N, Kd, out=c)
                if not torch.equal(c, ref):
                    print(f"  schedule {mode}: DIFFERENT from the 8-wave kernel", flush=True)
        for rnd_ in range(5):
            for mode in scheds:
                set_opt("gemm_w4", mode)
                K.gemm(lay, a, b, M, N, Kd, out=c)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    K.gemm(lay, a, b, M, N, Kd, out=c)
                e1.record()
                torch.cuda.synchronize()
                res[mode].append(e0.elapsed_time(e1) / 3)
        fl = 2.0 * M * N * Kd
        med = {m: sorted(v)[len(v) // 2] for m, v in res.items()}
        for m in med:
            tot[m] += med[m]
        print(f"{LN[lay]} M={M:6d} N={N:6d} K={Kd:6d}  " + "  ".join(f"s{m}: {fl / med[m] / 1e9:6.0f}" for m in scheds), flush=True)
    set_opt("gemm_w4", 1)
    print("TOTAL ms", {m: round(v, 3) for m, v in tot.items()}, flush=True)


def timing(quick):
    T = 8192
    shapes = [(NT, T, 6144, 4096), (NT, T, 4096, 4096), (NT, T, 28672, 4096), (NT, T, 4096, 14336), (NT, T, 128258, 4096),
              (NN, T, 4096, 6144), (NN, T, 4096, 4096), (NN, T, 4096, 28672), (NN, T, 14336, 4096)]
    if quick:
        shapes = shapes[:2] + shapes[5:7]
    g = torch.Generator(device="cuda").manual_seed(0)
    tot = {0: 0.0, 1: 0.0}
    for lay, M, N, Kd in shapes:
        a = rnd(g, M, Kd)
        b = rnd(g, N, Kd) if lay == NT else rnd(g, Kd, N)
        c = torch.empty(M, pad64(N), device="cuda", dtype=torch.bfloat16)[:, :N]
        res = {0: [], 1: []}
        for rnd_ in range(6):
            for mode in (0, 1):
                set_opt("gemm_w4", mode)
                K.gemm(lay, a, b, M, N, Kd, out=c)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    K.gemm(lay, a, b, M, N, Kd, out=c)
                e1.record()
                torch.cuda.synchronize()
                res[mode].append(e0.elapsed_time(e1) / 3)
        fl = 2.0 * M * N * Kd
        med = {m: sorted(v)[len(v) // 2] for m, v in res.items()}
        for m in med:
            tot[m] += med[m]
        print(f"{'NT' if lay == NT else 'NN'} M={M:6d} N={N:6d} K={Kd:6d}  8-wave {fl / med[0] / 1e9:7.1f} TF/s   4-wave {fl / med[1] / 1e9:7.1f} TF/s", flush=True)
    set_opt("gemm_w4", 1)
    print("TOTAL ms", tot, flush=True)


if __name__ == "__main__":
    gm = [a for a in sys.argv if a.startswith("--group-m=")]
    if gm:                                  # GROUP_M of the tile order (L2 / MALL locality of the panels)
        vals = [int(x) for x in gm[0][10:].split(",")]
        T = 8192
        g = torch.Generator(device="cuda").manual_seed(0)
        for lay, M, N, Kd in ((NT, T, 4096, 14336), (NT, T, 4096, 4096), (NT, T, 28672, 4096), (NN, T, 14336, 4096), (NN, T, 4096, 28672), (TN, 28672, 4096, T), (TN, 4096, 14336, T)):
            a = rnd(g, M, Kd) if lay != TN else rnd(g, Kd, M)
            b = rnd(g, N, Kd) if lay == NT else rnd(g, Kd, N)
            c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            t = {}
            for rep in range(5):
                for v in vals:
                    set_opt("gemm_w4_group_m", v)
                    K.gemm(lay, a, b, M, N, Kd, out=c)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(3):
                        K.gemm(lay, a, b, M, N, Kd, out=c)
                    e1.record()
                    torch.cuda.synchronize()
                    t.setdefault(v, []).append(e0.elapsed_time(e1) / 3)
            fl = 2.0 * M * N * Kd
            print(f"{LN[lay]} M={M:6d} N={N:6d} K={Kd:6d}  " + "  ".join(f"GROUP_M {v}: {fl / sorted(x)[len(x) // 2] / 1e9:6.0f}" for v, x in t.items()), flush=True)
        set_opt("gemm_w4_group_m", 8)
        sys.exit(0)
    if "--fused" in sys.argv:               # the fused-epilogue GEMMs of a decoder layer, 8-wave vs 4-wave kernel (option given as --opt=name:v0:v1)
        optn = [a for a in sys.argv if a.startswith("--opt=")]
        name, v0, v1 = (optn[0][6:].split(":") if optn else ("gemm_w4", "0", "1"))
        v0, v1 = int(v0), int(v1)
        g = torch.Generator(device="cuda").manual_seed(0)
        M, H, I = 8192, 4096, 14336
        x, wgu = rnd(g, M, H), rnd(g, 2 * I, H) * 0.05
        dy, wd = rnd(g, M, H), rnd(g, H, I) * 0.05
        gu = K.gemm_swiglu_fwd(x, wgu, I)[0]
        wq = rnd(g, 6144, H) * 0.05
        cos, sin = torch.rand(M, 64, device="cuda", generator=g), torch.rand(M, 64, device="cuda", generator=g)
        res = rnd(g, M, H)
        wo = rnd(g, H, H) * 0.05
        xi = rnd(g, M, I)
        wdn = rnd(g, H, I) * 0.05
        jobs = [("swiglu_fwd  NT 8192x28672x4096", 2.0 * M * 2 * I * H, lambda: K.gemm_swiglu_fwd(x, wgu, I)),
                ("swiglu_bwd  NN 8192x14336x4096", 2.0 * M * I * H, lambda: K.gemm_swiglu_bwd(dy, wd, gu, I)),
                ("qkv + rope  NT 8192x6144x4096", 2.0 * M * 6144 * H, lambda: K.gemm_rope_fwd(x, wq, None, 40 * 128, 128, cos, sin)),
                ("o_proj+res  NT 8192x4096x4096", 2.0 * M * H * H, lambda: K.gemm(NT, x, wo, M, H, H, residual=res)),
                ("down+res    NT 8192x4096x14336", 2.0 * M * H * I, lambda: K.gemm(NT, xi, wdn, M, H, I, residual=res))]
        for label, fl, fn in jobs:
            t = {v0: [], v1: []}
            for rep in range(5):
                for v in (v0, v1):
                    set_opt(name, v)
                    fn()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(3):
                        fn()
                    e1.record()
                    torch.cuda.synchronize()
                    t[v].append(e0.elapsed_time(e1) / 3)
            med = {v: sorted(x_)[len(x_) // 2] for v, x_ in t.items()}
            print(f"{label:34s} {name}={v0}: {fl / med[v0] / 1e9:6.0f} TF/s ({med[v0] * 1e3:6.0f} us)   {name}={v1}: {fl / med[v1] / 1e9:6.0f} TF/s ({med[v1] * 1e3:6.0f} us)", flush=True)
        set_opt(name, 1)
        sys.exit(0)
    stg = [a for a in sys.argv if a.startswith("--stagger=")]
    if stg:                                 # start delay per XCD slot (cycles): 0 = off
        vals = [int(x) for x in stg[0][10:].split(",")]
        sl = [a for a in sys.argv if a.startswith("--slots=")]
        if sl:
            set_opt("gemm_w4_stagger_slots", int(sl[0][8:]))
        T = 8192
        g = torch.Generator(device="cuda").manual_seed(0)
        for lay, M, N, Kd in ((NT, T, 4096, 4096), (NT, T, 4096, 1024), (NT, T, 6144, 4096), (NT, T, 28672, 4096), (NT, T, 4096, 14336), (NN, T, 14336, 4096), (TN, 28672, 4096, T), (TN, 4096, 4096, T)):
            a = rnd(g, M, Kd) if lay != TN else rnd(g, Kd, M)
            b = rnd(g, N, Kd) if lay == NT else rnd(g, Kd, N)
            c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            t = {}
            for rep in range(5):
                for v in vals:
                    set_opt("gemm_w4_stagger", v)
                    K.gemm(lay, a, b, M, N, Kd, out=c)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(3):
                        K.gemm(lay, a, b, M, N, Kd, out=c)
                    e1.record()
                    torch.cuda.synchronize()
                    t.setdefault(v, []).append(e0.elapsed_time(e1) / 3)
            fl = 2.0 * M * N * Kd
            print(f"{LN[lay]} M={M:6d} N={N:6d} K={Kd:6d}  " + "  ".join(f"{v}: {fl / sorted(x)[len(x) // 2] / 1e9:6.0f}" for v, x in t.items()), flush=True)
        set_opt("gemm_w4_stagger", 0)
        sys.exit(0)
    if "--rowmajor-ab" in sys.argv:         # accumulator-layout vs row-major epilogue of the 4-wave kernel, same process
        T = 8192
        g = torch.Generator(device="cuda").manual_seed(0)
        for lay, M, N, Kd, kind in ((NT, T, 4096, 4096, "plain"), (NT, T, 4096, 4096, "residual"), (NT, T, 6144, 4096, "plain"), (NT, T, 4096, 14336, "residual"),
                                    (NN, T, 4096, 4096, "plain"), (NN, T, 4096, 6144, "plain"), (TN, 4096, 4096, T, "plain"), (TN, 28672, 4096, T, "plain"),
                                    (TN, 4096, 4096, T, "accumulate")):
            a = rnd(g, M, Kd) if lay != TN else rnd(g, Kd, M)
            b = rnd(g, N, Kd) if lay == NT else rnd(g, Kd, N)
            res = rnd(g, M, N)
            c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            t, outs = {}, {}
            for rep in range(5):
                for mode in (0, 1, 2, 3):       # 0 round-3 pipelined epilogue, 1 wait-free accumulator layout, 2 wait-free row-major through LDS,
                    set_opt("gemm_w4_stream", 1 if mode else 0)          # 3 = 2 with the plain case by register lane exchange (the default)
                    set_opt("gemm_w4_rowmajor", 1 if mode >= 2 else 0)
                    set_opt("gemm_w4_shuffle", 1 if mode == 3 else 0)
                    kw = dict(residual=res if kind == "residual" else None, accumulate=kind == "accumulate")
                    c.copy_(res)
                    K.gemm(lay, a, b, M, N, Kd, out=c, **kw)
                    outs[mode] = c.clone()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(3):
                        K.gemm(lay, a, b, M, N, Kd, out=c, **kw)
                    e1.record()
                    torch.cuda.synchronize()
                    t.setdefault(mode, []).append(e0.elapsed_time(e1) / 3)
            fl = 2.0 * M * N * Kd
            med = {m: sorted(v)[len(v) // 2] for m, v in t.items()}
            print(f"{LN[lay]} M={M:6d} N={N:6d} K={Kd:6d} {kind:10s} pipe {fl / med[0] / 1e9:6.0f}  stream {fl / med[1] / 1e9:6.0f}  row-major {fl / med[2] / 1e9:6.0f}  shuffle {fl / med[3] / 1e9:6.0f} TF/s   "
                  f"{'bit-identical' if torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]) and torch.equal(outs[0], outs[3]) else 'DIFFERENT'}", flush=True)
        set_opt("gemm_w4_rowmajor", 0)
        set_opt("gemm_w4_stream", 1)
        sys.exit(0)
    sch = [a for a in sys.argv if a.startswith("--scheds=")]
    if sch:
        timing_scheds([int(x) for x in sch[0][9:].split(",")], "--quick" in sys.argv)
        sys.exit(0)
    rc = 0
    if "--time-only" not in sys.argv:
        rc = check()
    if "--check-only" not in sys.argv:
        timing("--quick" in sys.argv)
    sys.exit(1 if rc else 0)
