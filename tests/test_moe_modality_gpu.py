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
    return m.eval()          # the fixtures were made by the reference in eval mode: CrossAttention's dropouts are off


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


@pytest.mark.parametrize("pep", [False, True])
def test_shipped_cross_attn_recipes_construct_and_train(tmp_path, pep):
    """The two cross-attention recipes the reference ships (cookbook/sft/moe/*/attn/shared and .../attn/pep: five ViT-B/32 experts,
    hidden 4096, 8 cross-attention heads => head widths 768 / 8 = 96 and 4096 / 8 = 512) construct in bf16 and take a training
    step (train mode: both dropouts active).  Towers cut to 2 layers to keep the test small; widths, patch grid (P = 49), expert
    count and head counts are the recipes'."""
    from multimeditron_amd.model.modalities import AutoModality, MOEImageConfig, MOEImageConfigPEP
    from multimeditron_amd.model.presets import resolve_vision_config
    from multimeditron_amd.nn import FlatParams
    vis = dict(resolve_vision_config("openai/clip-vit-base-patch32"), num_hidden_layers=2)
    dirs = []
    for e in range(5):
        d = os.path.join(str(tmp_path), f"expert{e}")
        os.makedirs(d, exist_ok=True)
        json.dump({"vision_config": vis}, open(os.path.join(d, "config.json"), "w"))
        dirs.append(d)
    kw = dict(hidden_size=4096, expert_clip_names=dirs, image_processor=dirs[0], gating_path="stub", top_k_experts=5,
              generalist_idx=-1, fusion_method="cross_attn")
    cfg = (MOEImageConfigPEP if pep else MOEImageConfig)(**kw)
    assert cfg.model_type == ("moe_meditron_clip_pep" if pep else "moe_meditron_clip")
    assert AutoModality._cls("moe_meditron_clip_shared") is AutoModality._cls("moe_meditron_clip")      # the recipes' spelling

    def gate(px):
        logits = px.float().mean(dim=(2, 3)) @ torch.ones(3, 5, device=px.device) * torch.arange(5, device=px.device).float()
        return logits, logits.topk(1, dim=-1).indices, torch.softmax(logits, dim=-1)

    torch.manual_seed(0)
    m = AutoModality.model_from_config(cfg, dtype=torch.bfloat16, device="cuda", gating_network=gate)
    assert m.cross_attn.head_dim == (512 if pep else 96)
    with torch.no_grad():
        for k, p in m.named_parameters():
            torch.nn.init.normal_(p, std=0.02) if p.dim() > 1 else (torch.nn.init.ones_(p) if "norm" in k or "layrnorm" in k else torch.nn.init.zeros_(p))
    FlatParams([(k, p, "projector" if "projector" in k else "encoder") for k, p in m.named_parameters()], "cuda", torch.bfloat16)
    m.train()
    px = [torch.randn(3, 224, 224) for _ in range(3)]
    y = m(px)
    assert y.shape == (3, 49, 4096)
    y.float().square().mean().backward()
    torch.cuda.synchronize()
    for k, p in m.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad.float()).all(), k
    assert float(m.cross_attn.q_proj.weight.grad.float().abs().sum()) > 0


@pytest.mark.parametrize("pep", [False, True])
def test_frozen_towers_graph_replay_equals_eager(moe, moe_pep, tmp_path, pep, monkeypatch):
    """Frozen expert towers (the alignment / end2end recipes) run as one captured hipGraph per image count: bit-identical to the
    eager launches, across replays with new pixels and across image counts; the trainable projector / cross-attention behind them
    still get their gradients."""
    meta, w, v = moe_pep if pep else moe
    m = _build(meta, w, v, "cross_attn", torch.bfloat16, tmp_path)
    m.freeze_modality_embedder()
    px = v["pixels"]
    g = torch.Generator().manual_seed(3)
    batches = [[px[i] for i in range(px.shape[0])], [torch.randn_like(px[0], generator=g) for _ in range(px.shape[0])],
               [torch.randn_like(px[0], generator=g) for _ in range(2)], [px[i] for i in range(px.shape[0])]]
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("MM_MOE_GRAPH", mode)
        outs[mode] = [m(b).detach().clone() for b in batches]
    torch.cuda.synchronize()
    for a, b in zip(outs["0"], outs["1"]):
        assert torch.equal(a, b)
    assert torch.equal(outs["1"][0], outs["1"][3]) and not torch.equal(outs["1"][0], outs["1"][1])
    assert len(m.experts._mm_graphs) == 2                                    # two image counts
    monkeypatch.setenv("MM_MOE_GRAPH", "1")
    y = m(batches[0])
    y.float().square().mean().backward()
    trainable = [k for k, p in m.named_parameters() if p.requires_grad]
    assert trainable and all(dict(m.named_parameters())[k].grad is not None for k in trainable)
    assert all(p.grad is None for k, p in m.named_parameters() if k.startswith("experts."))
