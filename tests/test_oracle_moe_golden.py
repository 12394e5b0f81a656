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


def _load(golden_dir, name):
    meta = json.load(open(os.path.join(golden_dir, f"{name}.meta.json")))
    w = {k: t.float() for k, t in load_file(os.path.join(golden_dir, f"{name}.weights.safetensors")).items()}
    v = load_file(os.path.join(golden_dir, f"{name}.vectors.safetensors"))
    return meta, w, v


@pytest.fixture(scope="module")
def moe(golden_dir):
    return _load(golden_dir, "tiny_moe_clip")


@pytest.fixture(scope="module")
def moe_pep(golden_dir):
    return _load(golden_dir, "tiny_moe_clip_pep")


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


@pytest.mark.parametrize("fusion", ["weighted_average", "sequence_append", "cross_attn"])
def test_moe_pep_modality_outputs_and_grads(moe_pep, fusion):
    """MOEImageModalityPEP (image_modality_moe_pep.py): one projector per expert, fusion in the projected space."""
    meta, w, v = moe_pep
    assert meta["per_expert_projection"]
    wg = {k: t.clone().requires_grad_(True) for k, t in w.items()}
    y = R.moe_image_modality_pep(wg, v["pixels"].float(), v[f"{fusion}.gate_weights"], meta["vision"], meta["num_experts"], fusion,
                                 generalist_idx=meta["generalist_idx"], heads=meta["cross_attn_heads"])
    assert y.shape == v[f"{fusion}.out"].shape
    assert rel(y, v[f"{fusion}.out"]) < 2e-5
    (y * v[f"{fusion}.dout"]).sum().backward()
    n = 0
    for key, ref in v.items():
        if not key.startswith(f"{fusion}.grad."):
            continue
        g = wg[key[len(fusion) + 6:]].grad
        if g is None:
            # cross_attn reads only the generalist's queries and the specialists' keys/values: every expert still contributes;
            # a parameter the reference reports a gradient for must have one here
            raise AssertionError(key)
        assert float((g - ref).norm()) <= 1e-4 * float(ref.norm()) + 2e-6, key
        n += 1
    assert n >= 20
