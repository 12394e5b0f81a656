"""Pins the oracle's MoE image modality (oracle/ref_cpu.py moe_image_modality / cross_attention) to vectors produced by the
REAL reference classes (MOEImageModality, CrossAttention: tools/make_golden.py moe_fixture) for all three fusions.  The gate
is the harness's stub (the reference's ResNet-50 GatingNetwork needs torchvision, absent here: that part is parity-unpinned);
its softmax weights are part of the fixture.  fp32; 2e-5 on outputs, 1e-4 on grads."""
import json
import os

import pytest
import torch
from safetensors.torch import load_file

from oracle import ref_cpu as R


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


@pytest.fixture(scope="module")
def moe(golden_dir):
    meta = json.load(open(os.path.join(golden_dir, "tiny_moe_clip.meta.json")))
    w = {k: t.float() for k, t in load_file(os.path.join(golden_dir, "tiny_moe_clip.weights.safetensors")).items()}
    v = load_file(os.path.join(golden_dir, "tiny_moe_clip.vectors.safetensors"))
    return meta, w, v


def test_stub_gate_contract(moe):
    meta, w, v = moe
    gw = torch.softmax(v["pixels"].float().mean(dim=(2, 3)) @ v["gate.w"].float().t() + v["gate.b"].float(), dim=-1)
    assert rel(gw, v["weighted_average.gate_weights"]) < 1e-6


@pytest.mark.parametrize("fusion", ["weighted_average", "sequence_append", "cross_attn"])
def test_moe_modality_outputs_and_grads(moe, fusion):
    meta, w, v = moe
    wg = {k: t.clone().requires_grad_(True) for k, t in w.items()}
    y = R.moe_image_modality(wg, v["pixels"].float(), v[f"{fusion}.gate_weights"], meta["vision"], meta["num_experts"], fusion,
                             generalist_idx=meta["generalist_idx"], heads=meta["cross_attn_heads"])
    assert y.shape == v[f"{fusion}.out"].shape
    assert rel(y, v[f"{fusion}.out"]) < 2e-5
    (y * v[f"{fusion}.dout"]).sum().backward()
    n = 0
    for key, ref in v.items():
        if not key.startswith(f"{fusion}.grad."):
            continue
        g = wg[key[len(fusion) + 6:]].grad
        assert g is not None, key
        assert float((g - ref).norm()) <= 1e-4 * float(ref.norm()) + 2e-6, key      # k_proj.bias is analytically 0: noise on both sides
        n += 1
    assert n >= 20
