#!/usr/bin/env python3
"""MoE image modality at ViT-L/14 size: E experts on n images, forward + backward, experts on one stream vs one stream each.
   python tools/moe_bench.py [E] [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimeditron_amd.model.modalities import MOEImageConfig, MOEImageModality
from multimeditron_amd.nn import FlatParams

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
torch.manual_seed(0)


def gate(px):
    logits = px.float().mean(dim=(2, 3))[:, :1].repeat(1, E)
    return logits, logits.topk(1, dim=-1).indices, torch.softmax(logits, dim=-1)


for fusion in ("weighted_average", "cross_attn"):
    cfg = MOEImageConfig(hidden_size=4096, expert_clip_names=["openai/clip-vit-large-patch14"] * E, image_processor="openai/clip-vit-large-patch14",
                         top_k_experts=E, generalist_idx=E - 1, fusion_method=fusion, cross_attn_heads=8)
    m = MOEImageModality(cfg, dtype=torch.bfloat16, device="cuda", gating_network=gate)
    FlatParams([(k, p, "projector" if k.startswith("projector") else "encoder") for k, p in m.named_parameters()], "cuda", torch.bfloat16)
    for p in m.parameters():
        p.requires_grad_(True)
    px = torch.randn(n, 3, 224, 224, device="cuda")
    ref = None
    for streams, graph in (("0", "0"), ("1", "0"), ("1", "1"), ("0", "0"), ("1", "0"), ("1", "1")):
        os.environ["MM_MOE_STREAMS"] = streams
        os.environ["MM_MOE_TRAIN_GRAPH"] = graph        # trainable towers replayed from captured forward / backward graphs
        for it in range(6):
            if it == 3:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            y = m(px)
            y.backward(torch.ones_like(y))
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        same = "" if ref is None else f"  output identical to the sequential run: {bool(torch.equal(ref, y))}"
        ref = y.detach().clone() if ref is None else ref
        print(f"{fusion}: E={E} experts (ViT-L/14), n={n} images, MM_MOE_STREAMS={streams} MM_MOE_TRAIN_GRAPH={graph}: {ms:.2f} ms fwd+bwd{same}", flush=True)
    # the shipped alignment / end2end recipes: towers frozen, projector (and cross-attention) trainable
    m.freeze_modality_embedder()
    for graph in ("0", "1", "0", "1"):
        os.environ["MM_MOE_GRAPH"] = graph
        for it in range(4):
            if it == 2:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            y = m(px)
            y.backward(torch.ones_like(y))
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 2 * 1e3
        print(f"{fusion}: frozen towers, MM_MOE_GRAPH={graph}: {ms:.2f} ms fwd (+ projector / fusion bwd)", flush=True)
    del m
    torch.cuda.empty_cache()
