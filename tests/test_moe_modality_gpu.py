"""MoE image modalities on the HIP path (GPU) against vectors produced by the REAL reference classes (MOEImageModality,
MOEImageModalityPEP + CrossAttention, eval mode; tools/make_golden.py moe_fixture): outputs of all three fusions and the gradients of the projector,
the cross-attention, and expert layers.  The gate is a stub with the reference gate's output contract (the ResNet-50 gate is
not part of this build: parity-unpinned).  fp32 path <= 1e-4 (outputs) / 1e-3 (grads); bf16 path <= 3e-2 / 6e-2."""
import json
import os

import pytest
import torch
from safetensors.torch import load_file

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope="module")
def moe(golden_dir):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    meta = json.load(open(os.path.join(golden_dir, "tiny_moe_clip.meta.json")))
    w = load_file(os.path.join(golden_dir, "tiny_moe_clip.weights.safetensors"))
    v = load_file(os.path.join(golden_dir, "tiny_moe_clip.vectors.safetensors"))
    return meta, w, v


@pytest.fixture(scope="module")
def moe_pep(golden_dir):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    meta = json.load(open(os.path.join(golden_dir, "tiny_moe_clip_pep.meta.json")))
    w = load_file(os.path.join(golden_dir, "tiny_moe_clip_pep.weights.safetensors"))
    v = load_file(os.path.join(golden_dir, "tiny_moe_clip_pep.vectors.safetensors"))
    return meta, w, v


def _build(meta, w, v, fusion, dtype, tmp):
    from multimeditron_amd.model.modalities import MOEImageConfig, MOEImageModality
    if meta.get("per_expert_projection"):
        from multimeditron_amd.model.modalities import MOEImageConfigPEP as MOEImageConfig, MOEImageModalityPEP as MOEImageModality
    from multimeditron_amd.nn import FlatParams
    E = meta["num_experts"]
    dirs = []
    for e in range(E):
        d = os.path.join(str(tmp), f"clip{e}")
        os.makedirs(d, exist_ok=True)
        json.dump({"vision_config": meta["vision"]}, open(os.path.join(d, "config.json"), "w"))
        dirs.append(d)
    gw_, gb_ = v["gate.w"].float().cuda(), v["gate.b"].float().cuda()

    def gate(px):      # the harness's stub gate: softmax(mean_hw(pixels) @ Wg^T + bg) -- torch ops on [n, 3]: test-side stand-in
        logits = px.float().mean(dim=(2, 3)) @ gw_.t() + gb_
        return logits, logits.topk(1, dim=-1).indices, torch.softmax(logits, dim=-1)

    cfg = MOEImageConfig(hidden_size=meta["hidden_size"], expert_clip_names=dirs, image_processor=dirs[0], gating_path="stub",
                         top_k_experts=E, generalist_idx=meta["generalist_idx"], fusion_method=fusion,
                         cross_attn_heads=meta["cross_attn_heads"])
    m = MOEImageModality(cfg, dtype=dtype, device="cuda", gating_network=gate)
    own = dict(m.named_parameters())
    with torch.no_grad():
        for k, p in own.items():
            p.copy_(w[k].to(dtype).reshape(p.shape))
    assert set(own) <= set(w)
    FlatParams([(k, p, "projector" if k.startswith("projector") else "encoder") for k, p in own.items()], "cuda", dtype)
    for p in m.parameters():
        p.requires_grad_(True)
    return m


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("fusion", ["weighted_average", "sequence_append", "cross_attn"])
def test_moe_fusions_match_reference(moe, tmp_path, fusion, dtype):
    meta, w, v = moe
    m = _build(meta, w, v, fusion, dtype, tmp_path)
    px = v["pixels"]
    y = m([px[i] for i in range(px.shape[0])])
    tol_o, tol_g = (1e-4, 1e-3) if dtype == torch.float32 else (3e-2, 6e-2)
    assert y.shape == v[f"{fusion}.out"].shape
    assert rel(y.float(), v[f"{fusion}.out"]) < tol_o
    y.backward(v[f"{fusion}.dout"].to(dtype).cuda())
    torch.cuda.synchronize()
    own = dict(m.named_parameters())
    n = 0
    for key, ref in v.items():
        if not key.startswith(f"{fusion}.grad.") or ref.dim() < 2:
            continue
        g = own[key[len(fusion) + 6:]].grad
        assert g is not None, key
        assert rel(g.float().reshape(ref.shape), ref) < tol_g, key
        n += 1
    assert n >= 10


def test_moe_gate_is_a_plug(moe, tmp_path):
    meta, w, v = moe
    m = _build(meta, w, v, "weighted_average", torch.float32, tmp_path)
    m.gating_network = None
    with pytest.raises(NotImplementedError):
        m([v["pixels"][0]])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("fusion", ["weighted_average", "sequence_append", "cross_attn"])
def test_moe_pep_fusions_match_reference(moe_pep, tmp_path, fusion, dtype):
    """MOEImageModalityPEP (reference image_modality_moe_pep.py): one projector per expert, fusion in the projected space."""
    meta, w, v = moe_pep
    m = _build(meta, w, v, fusion, dtype, tmp_path)
    assert type(m).__name__ == "MOEImageModalityPEP" and len(m.projectors) == meta["num_experts"]
    px = v["pixels"]
    y = m([px[i] for i in range(px.shape[0])])
    tol_o, tol_g = (1e-4, 1e-3) if dtype == torch.float32 else (3e-2, 6e-2)
    assert y.shape == v[f"{fusion}.out"].shape
    assert rel(y.float(), v[f"{fusion}.out"]) < tol_o
    y.backward(v[f"{fusion}.dout"].to(dtype).cuda())
    torch.cuda.synchronize()
    own = dict(m.named_parameters())
    n = 0
    for key, ref in v.items():
        if not key.startswith(f"{fusion}.grad.") or ref.dim() < 2:
            continue
        g = own[key[len(fusion) + 6:]].grad
        assert g is not None, key
        assert rel(g.float().reshape(ref.shape), ref) < tol_g, key
        n += 1
    assert n >= 10
