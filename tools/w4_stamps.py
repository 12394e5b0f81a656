#!/usr/bin/env python3
"""Diag library only (tools/build_diag.sh w4diag -DMM_W4_DIAG; MM_HIP_LIBRARY=...): cycles the 4-wave GEMM's waves spend at the
K-step barrier (schedule 121: s_memtime before / after every s_barrier), per workgroup and wave."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimeditron_amd import kernels as K
from multimeditron_amd._lib import lib

L = lib()
EPI = [0]                                # epi=0,1,2,4,7,8: timing-only epilogue variants (GemmArgs::diag_epi: 1 no global stores, 2 no LDS reads,
SHAPES = ((8192, 4096, 4096), (8192, 4096, 14336), (8192, 28672, 4096))        # 4 no LDS writes, 8 no epilogue at all)
for a in sys.argv[1:]:                   # e.g. gemm_w4_rowmajor=0 gemm_w4_stream=0: the round-3 pipelined epilogue
    k, v = a.split("=")
    if k == "shape":                     # shape=M,N,K[;M,N,K...]
        SHAPES = tuple(tuple(int(x) for x in t.split(",")) for t in v.split(";"))
        continue
    if k == "epi":
        EPI = [int(x) for x in v.split(",")]
        continue
    assert L.mm_set_option(k.encode(), int(v)) == 0, a
    print("option", a, flush=True)
g = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s: (torch.rand(*s, device="cuda", generator=g) * 2 - 1).to(torch.bfloat16)
buf = (ctypes.c_uint * (256 * 4 * 7))()
for M, N, Kd, epi in [(m, n, k, e) for (m, n, k) in SHAPES for e in EPI]:
    assert L.mm_set_option(b"gemm_w4_diag_epi", epi) == 0
    if len(EPI) > 1:
        print("diag_epi", epi, end=": ")
    a, b = r(M, Kd), r(N, Kd)
    c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    assert L.mm_set_option(b"gemm_w4", 121) == 0
    for _ in range(3):
        K.gemm(0, a, b, M, N, Kd, out=c)
    torch.cuda.synchronize()
    assert L.mm_w4_diag_read(buf, 1) == 0
    K.gemm(0, a, b, M, N, Kd, out=c)
    torch.cuda.synchronize()
    assert L.mm_w4_diag_read(buf, 1) == 0
    allw = torch.tensor(list(buf), dtype=torch.float64)
    t = allw[:256 * 16].view(256, 4, 4)
    gap = allw[256 * 16:256 * 24].view(256, 4, 2)[:, :, 0]
    epi = allw[256 * 24:].view(256, 4)
    nk = Kd // 64
    tiles = t[:, :, 3]
    wait, mx, loop = t[:, :, 0], t[:, :, 1], t[:, :, 2]
    print(f"NT M={M} N={N} K={Kd}: tiles/WG {tiles.mean():.2f}; loop cycles per K-step {float((loop / tiles / nk).mean()):.0f} (ideal 2048); "
          f"barrier cycles per K-step: mean {float((wait / tiles / nk).mean()):.0f}, by wave {[round(float((wait[:, w] / tiles[:, w] / nk).mean())) for w in range(4)]}, "
          f"first P of a tile {float((mx / tiles).mean()):.0f} cycles; between two tiles' loops {float((gap / (tiles - 1).clamp(min=1)).mean()):.0f} cycles "
          f"(a tile's loop {float((loop / tiles).mean()):.0f}); loop end -> last epilogue instruction issued {float((epi / tiles).mean()):.0f}", flush=True)
    if hasattr(L, "mm_w4_diag2_read"):
        b2 = (ctypes.c_uint * (256 * 4 * 8))()
        if L.mm_w4_diag2_read(b2, 1) == 0:
            d2 = torch.tensor(list(b2), dtype=torch.float64).view(256, 4, 8)
            if float(d2.sum()) > 0:
                per = (d2[:, :, :5] / tiles.unsqueeze(-1)).mean(dim=(0, 1))
                print(f"   register-exchange epilogue, cycles since the loop's end (mean): entry {per[0]:.0f}, half 0 accumulators read {per[1]:.0f}, half 0 stored {per[2]:.0f}, "
                      f"half 1 read {per[3]:.0f}, half 1 stored {per[4]:.0f}")
    for wg in (0, 1, 100):
        print(f"   WG {wg}: wait/K-step by wave {[round(float(wait[wg, w] / tiles[wg, w] / nk)) for w in range(4)]}  loop/K-step {[round(float(loop[wg, w] / tiles[wg, w] / nk)) for w in range(4)]}")
