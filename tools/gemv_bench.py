#!/usr/bin/env python3
"""The decode step's weight-streaming kernels one by one at the 8B decoder's shapes (B rows of activations), each timed over L
different layers' weights (L x 436 MB: far beyond the 256 MB Infinity Cache, as in a real token) -- us per launch and TB/s of
weight bytes.   python tools/gemv_bench.py [B] [L]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimeditron_amd import kernels as K

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
L = int(sys.argv[2]) if len(sys.argv) > 2 else 16
H, I, Hq, Hkv, D, S = 4096, 14336, 32, 8, 128, 2048
dev = "cuda"
torch.manual_seed(0)
bf = torch.bfloat16
w_qkv = [torch.randn((Hq + 2 * Hkv) * D, H, device=dev, dtype=bf) * 0.02 for _ in range(L)]
w_o = [torch.randn(H, H, device=dev, dtype=bf) * 0.02 for _ in range(L)]
w_gu = [torch.randn(2 * I, H, device=dev, dtype=bf) * 0.02 for _ in range(L)]
w_d = [torch.randn(H, I, device=dev, dtype=bf) * 0.02 for _ in range(L)]
kc = [torch.randn(B, S + 8, Hkv, D, device=dev, dtype=bf) for _ in range(L)]
vc = [torch.randn(B, S + 8, Hkv, D, device=dev, dtype=bf) for _ in range(L)]
nw = torch.ones(H, device=dev, dtype=bf)
x = torch.randn(B, H, device=dev, dtype=bf)
act = torch.randn(B, I, device=dev, dtype=bf)
q = torch.randn(B, Hq, D, device=dev, dtype=bf)
cos = torch.randn(B, 64, device=dev)
sin = torch.randn(B, 64, device=dev)


def timed(name, fn, bytes_per_call, reps=6):
    for i in range(L):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        for i in range(L):
            fn(i)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (reps * L)
    print(f"{name:34s} {us:7.1f} us/launch   {bytes_per_call / us / 1e6:5.2f} TB/s   ({bytes_per_call / 1e6:.0f} MB)", flush=True)
    return us


nw2 = torch.ones(I, device=dev, dtype=bf)
from multimeditron_amd._lib import lib
for nt in (0, 1):
    lib().mm_set_option(b"gemv_nt", nt)
    print(f"-- weight loads: {'non-temporal' if nt else 'default policy'}")
    tot = 0.0
    tot += timed("norm + q|k|v + RoPE + append (N=6144)", lambda i: K.decode_qkv_rope_append(x, w_qkv[i], None, Hq, Hkv, D, cos, sin, kc[i], vc[i], S, norm_w=nw, eps=1e-5), w_qkv[0].numel() * 2)
    tot += timed("attention partial+merge (S=2048)", lambda i: K.attn_decode(q, kc[i][:, :S], vc[i][:, :S], None, D ** -0.5), 2 * B * S * Hkv * D * 2)
    tot += timed("o_proj + residual (N=4096)", lambda i: K.decode_linear(x, w_o[i], residual=x), w_o[0].numel() * 2)
    tot += timed("norm + gate|up + SwiGLU (N=2x14336)", lambda i: K.decode_gateup_swiglu(x, w_gu[i], I, norm_w=nw, eps=1e-5), w_gu[0].numel() * 2)
    tot += timed("down + residual (K=14336)", lambda i: K.decode_linear(act, w_d[i], residual=x), w_d[0].numel() * 2)
    print(f"sum of the five per layer: {tot:.1f} us -> x32 layers = {tot * 32 / 1e3:.2f} ms/token (+ lm_head, embedding, select)")
lib().mm_set_option(b"gemv_nt", 0)
for wgs in (1, 2, 3):
    lib().mm_set_option(b"gemv_wgs", wgs)
    print(f"-- persistent workgroups per CU: {wgs}")
    timed("norm + q|k|v + RoPE + append", lambda i: K.decode_qkv_rope_append(x, w_qkv[i], None, Hq, Hkv, D, cos, sin, kc[i], vc[i], S, norm_w=nw, eps=1e-5), w_qkv[0].numel() * 2)
    timed("norm + gate|up + SwiGLU", lambda i: K.decode_gateup_swiglu(x, w_gu[i], I, norm_w=nw, eps=1e-5), w_gu[0].numel() * 2)
    timed("down + residual", lambda i: K.decode_linear(act, w_d[i], residual=x), w_d[0].numel() * 2)
lib().mm_set_option(b"gemv_wgs", 0)
for wgs in (256, 512, 768, 1024, 1536):
    lib().mm_set_option(b"attn_decode_wgs", wgs)
    timed(f"attention, ~{wgs} workgroups", lambda i: K.attn_decode(q, kc[i][:, :S], vc[i][:, :S], None, D ** -0.5), 2 * B * S * Hkv * D * 2)
lib().mm_set_option(b"attn_decode_wgs", 768)
lib().mm_set_option(b"attn_decode_mfma", 0)
timed("attention, scores on the vector ALU (round-2 slice kernel)", lambda i: K.attn_decode(q, kc[i][:, :S], vc[i][:, :S], None, D ** -0.5), 2 * B * S * Hkv * D * 2)
lib().mm_set_option(b"attn_decode_mfma", 1)
lib().mm_set_option(b"gemv_stream", 0)
timed("round-2 kernel: plain o_proj", lambda i: K.linear_fwd(x, w_o[i]), w_o[0].numel() * 2)
timed("round-2 kernel: plain down", lambda i: K.linear_fwd(act, w_d[i]), w_d[0].numel() * 2)
timed("round-2 kernel: plain gate|up", lambda i: K.linear_fwd(x, w_gu[i]), w_gu[0].numel() * 2)
lib().mm_set_option(b"gemv_stream", 1)
