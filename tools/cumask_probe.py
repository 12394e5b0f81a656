#!/usr/bin/env python3
"""Where do the workgroups of a CU-masked stream run?  For each mask (CUs enabled, scheme): distinct (XCC, SE, SH, CU) slots used
by 1024 spinning 512-thread workgroups, and how many per XCC.   python tools/cumask_probe.py"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimeditron_amd import kernels as K

torch.cuda.init()
x = torch.zeros(1, device="cuda")


def report(tag, stream):
    r = K.cu_probe(2048, 512, 20000, stream)             # 200 us each: every enabled CU gets several
    torch.cuda.synchronize()
    xcc = r[:, 0] & 0xF
    hw = r[:, 1]
    cu, sh, se = (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 0x7
    slots = set(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
    per = collections.Counter(s[0] for s in slots)
    print(f"{tag}: {len(slots)} distinct CUs; per XCC {[per.get(i, 0) for i in range(8)]}", flush=True)


report("unmasked stream", None)
for n in (248, 240, 224, 192, 128):
    for scheme in ("hash", "stride"):
        try:
            report(f"mask {n} CUs ({scheme})", K.masked_stream(n, scheme))
        except Exception as e:
            print(f"mask {n} ({scheme}): {type(e).__name__}: {e}")
