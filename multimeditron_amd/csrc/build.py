#!/usr/bin/env python3
"""Build libmmhip.so (gfx950) in-tree:  python multimeditron_amd/csrc/build.py [--force]

Plain hipcc, one object per .hip, linked into multimeditron_amd/libmmhip.so.  The .so is git-ignored but
travels with the gpurun snapshot.  Cross-compiles without a GPU.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OUT = os.path.join(PKG, "libmmhip.so")
SOURCES = ["mm_gemm.hip", "mm_attn.hip", "mm_rowwise.hip", "mm_embed.hip", "mm_optim.hip", "mm_debug.hip", "mm_comm.hip", "mm_image.hip", "mm_xattn.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    gen, inc = os.path.join(HERE, "gen_gemm_w4.py"), os.path.join(HERE, "mm_gemm_w4.inc")
    if force or _stale(inc, [gen]):                     # the 4-wave GEMM's main loop (asm text) is generated
        subprocess.run([sys.executable, gen], check=True, stdout=subprocess.DEVNULL)
    headers = [os.path.join(HERE, "mm_common.h"), os.path.join(PKG, "..", "include", "mm_hip.h"), inc]
    jobs = []
    for src in SOURCES:
        s = os.path.join(HERE, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        # -Rpass-analysis=kernel-resource-usage: registers / scratch / LDS of every kernel, kept beside the object
        # (<obj>.resources.txt): tests/test_no_spills_cpu.py fails the build of a hot kernel that started to spill
        cmd = [HIPCC] + FLAGS + ["-Rpass-analysis=kernel-resource-usage", "-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
        remarks = [ln for ln in r.stderr.splitlines() if "remark:" in ln]
        import re
        ctx = re.compile(r"^\s*\d*\s*\|")               # the source excerpt clang prints under each remark
        other = [ln for ln in r.stderr.splitlines() if "remark:" not in ln and "-Rpass-analysis" not in ln and not ctx.match(ln)]
        if other:
            sys.stderr.write("\n".join(other) + "\n")
        if r.returncode != 0:
            raise subprocess.CalledProcessError(r.returncode, cmd)
        with open(o + ".resources.txt", "w") as f:
            f.write("\n".join(remarks) + "\n")

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(OUT, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
