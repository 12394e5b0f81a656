#!/usr/bin/env python3
"""GEMM micro-benchmark on the shapes of the 8B training step (run on the GPU box).
   python tools/gemm_bench.py [--quick]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimeditron_amd import kernels as K

T = 8192
SHAPES = {
    "NT": [(T, 6144, 4096), (T, 4096, 4096), (T, 28672, 4096), (T, 4096, 14336), (T, 128258, 4096)],
    "NN": [(T, 4096, 6144), (T, 4096, 4096), (T, 4096, 28672), (T, 14336, 4096), (T, 4096, 128258)],
    "TN": [(6144, 4096, T), (4096, 4096, T), (28672, 4096, T), (4096, 14336, T), (128258, 4096, T)],
}
LAY = {"NT": 0, "NN": 1, "TN": 2}


def pad64(n):
    return (n + 63) // 64 * 64


def operands(lay, M, N, Kd):
    g = torch.Generator(device="cuda").manual_seed(0)
    r = lambda *s: (torch.rand(*s, device="cuda", generator=g) * 2 - 1).to(torch.bfloat16)
    if lay == "NT":
        return r(M, Kd)[:, :Kd], r(N, Kd)
    if lay == "NN":
        a = torch.zeros(M, pad64(Kd), device="cuda", dtype=torch.bfloat16); a[:, :Kd] = r(M, Kd)
        return a[:, :Kd], r(Kd, N)
    a = torch.zeros(Kd, pad64(M), device="cuda", dtype=torch.bfloat16); a[:, :M] = r(Kd, M)
    return a[:, :M], r(Kd, N)


def set_opt(name, v):
    from multimeditron_amd._lib import lib
    assert lib().mm_set_option(name.encode(), v) == 0


def main():
    quick = "--quick" in sys.argv
    ab = [a for a in sys.argv if a.startswith("--ab=")]
    if ab:                                # same process, same device: interleaved A/B of one option, e.g. --ab=gemm_issue_waves:8:4
        name, v0, v1 = ab[0][5:].split(":")
        v0, v1 = int(v0), int(v1)
        tot = {v0: 0.0, v1: 0.0}
        for lay, shapes in SHAPES.items():
            for (M, N, Kd) in shapes:
                a, b = operands(lay, M, N, Kd)
                c = torch.empty(M, pad64(N), device="cuda", dtype=torch.bfloat16)[:, :N]
                res = {v0: [], v1: []}
                for rnd in range(6):
                    for mode in (v0, v1):
                        set_opt(name, mode)
                        K.gemm(LAY[lay], a, b, M, N, Kd, out=c)
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        for _ in range(3):
                            K.gemm(LAY[lay], a, b, M, N, Kd, out=c)
                        e1.record()
                        torch.cuda.synchronize()
                        res[mode].append(e0.elapsed_time(e1) / 3)
                fl = 2.0 * M * N * Kd
                med = {m: sorted(v)[len(v) // 2] for m, v in res.items()}
                for m in med:
                    tot[m] += med[m]
                print(f"{lay} M={M:6d} N={N:6d} K={Kd:6d}  {name}={v0}: {fl / med[v0] / 1e9:7.1f} TF/s   {name}={v1}: {fl / med[v1] / 1e9:7.1f} TF/s", flush=True)
        print("TOTAL ms", tot)
        return
    if "--decode" in sys.argv:            # one token per sequence (M = 4): weight-streaming kernel vs the tiled one, GB/s of W
        for (N, Kd) in [(6144, 4096), (4096, 4096), (28672, 4096), (4096, 14336), (128258, 4096)]:
            a, b = operands("NT", 4, N, Kd)
            c = torch.empty(4, pad64(N), device="cuda", dtype=torch.bfloat16)[:, :N]
            line = f"M=4 N={N:6d} K={Kd:5d} "
            for mode in (0, 1):
                set_opt("gemm_skinny", mode)
                for _ in range(3):
                    K.gemm(0, a, b, 4, N, Kd, out=c)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    K.gemm(0, a, b, 4, N, Kd, out=c)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 10
                line += f"  {'skinny' if mode else 'tiled '} {ms * 1e3:7.1f} us {N * Kd * 2 / ms / 1e6:7.0f} GB/s"
            print(line, flush=True)
        set_opt("gemm_skinny", 1)
        return
    if "--small" in sys.argv:             # ViT-L/14 (4 images = 1028 tokens) and projector shapes: every kernel variant, same process
        T2 = 1028
        shapes = {"NT": [(T2, 3072, 1024), (T2, 1024, 1024), (T2, 4096, 1024), (T2, 1024, 4096), (1024, 4096, 4096)],
                  "NN": [(T2, 1024, 3072), (T2, 1024, 1024), (T2, 1024, 4096), (T2, 4096, 1024), (1024, 4096, 4096)],
                  "TN": [(3072, 1024, T2), (1024, 1024, T2), (4096, 1024, T2), (1024, 4096, T2), (4096, 4096, 1024)]}
        names = {1: "v1", 2: "256x128", 4: "128x128", 5: "64x128", 6: "64x64"}
        tot = {m: 0.0 for m in names}
        for lay, shp in shapes.items():
            for (M, N, Kd) in shp:
                a, b = operands(lay, M, N, Kd)
                c = torch.empty(M, pad64(N), device="cuda", dtype=torch.bfloat16)[:, :N]
                res = {m: [] for m in names}
                for rnd in range(5):
                    for mode in names:
                        set_opt("gemm_kernel", mode)
                        K.gemm(LAY[lay], a, b, M, N, Kd, out=c)
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        for _ in range(5):
                            K.gemm(LAY[lay], a, b, M, N, Kd, out=c)
                        e1.record()
                        torch.cuda.synchronize()
                        res[mode].append(e0.elapsed_time(e1) / 5)
                med = {m: sorted(v)[len(v) // 2] for m, v in res.items()}
                for m in med:
                    tot[m] += med[m]
                print(f"{lay} M={M:5d} N={N:5d} K={Kd:5d}  " + "  ".join(f"{names[m]} {med[m] * 1e3:6.1f}us" for m in names), flush=True)
        set_opt("gemm_kernel", 0)
        print("TOTAL us", {names[m]: round(v * 1e3, 1) for m, v in tot.items()})
        return
    if "--ab-persist" in sys.argv:        # same process, same device: interleaved A/B of the persistent tile loop
        for lay, shapes in SHAPES.items():
            for (M, N, Kd) in shapes:
                a, b = operands(lay, M, N, Kd)
                c = torch.empty(M, pad64(N), device="cuda", dtype=torch.bfloat16)[:, :N]
                res = {0: [], 1: []}
                for rnd in range(6):
                    for mode in (0, 1):
                        set_opt("gemm_persist", mode)
                        K.gemm(LAY[lay], a, b, M, N, Kd, out=c)
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        for _ in range(3):
                            K.gemm(LAY[lay], a, b, M, N, Kd, out=c)
                        e1.record()
                        torch.cuda.synchronize()
                        res[mode].append(e0.elapsed_time(e1) / 3)
                fl = 2.0 * M * N * Kd
                med = {m: sorted(v)[len(v) // 2] for m, v in res.items()}
                print(f"{lay} M={M:6d} N={N:6d} K={Kd:6d}  one-tile-per-wg {fl / med[0] / 1e9:7.1f} TF/s   persistent {fl / med[1] / 1e9:7.1f} TF/s", flush=True)
        return
    tot_f = tot_t = 0.0
    for lay, shapes in SHAPES.items():
        for (M, N, Kd) in (shapes[:2] if quick else shapes):
            a, b = operands(lay, M, N, Kd)
            c = torch.empty(M, pad64(N), device="cuda", dtype=torch.bfloat16)[:, :N]
            for _ in range(2):
                K.gemm(LAY[lay], a, b, M, N, Kd, out=c)
            torch.cuda.synchronize()
            it = 5
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(it):
                K.gemm(LAY[lay], a, b, M, N, Kd, out=c)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / it
            fl = 2.0 * M * N * Kd
            tot_f += fl; tot_t += ms
            print(f"{lay} M={M:6d} N={N:6d} K={Kd:6d}  {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TF/s", flush=True)
    print(f"TOTAL {tot_t:.2f} ms  {tot_f / tot_t / 1e9:.1f} TF/s")


if __name__ == "__main__":
    main()
