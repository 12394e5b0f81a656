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


@pytest.mark.parametrize("pep", [False, True])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_trainable_towers_graph_replay_equals_eager(moe, moe_pep, tmp_path, pep, dtype, monkeypatch):
    """TRAINABLE expert towers (the `full` recipe) replayed from captured forward / backward hipGraphs (_TowerGraphs) against the
    eager launches: outputs and every gradient bit-identical (bf16; fp32: to 1e-6, its attention backward uses atomics), call after call -- the first backward (eager, through the capture's
    autograd graph), the captured `first gradient of the step` graph, its replay, and the `accumulate` graph (no reset between
    two backward passes); new pixels every call; the gradient-ready hook fires once per tower parameter per backward."""
    from multimeditron_amd import functional as Fm
    meta, w, v = moe_pep if pep else moe
    px = v["pixels"]
    g = torch.Generator().manual_seed(5)
    batches = [[px[i] for i in range(px.shape[0])]] + [[torch.randn_like(px[0], generator=g) for _ in range(px.shape[0])] for _ in range(4)]
    reset = [True, True, True, False, True]                                 # call 3 accumulates onto call 2's gradients
    got = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("MM_MOE_TRAIN_GRAPH", mode)
        m = _build(meta, w, v, "cross_attn", dtype, tmp_path / f"g{mode}")
        flat = next(iter(m.parameters()))._mm_flat
        tower = [p for k, p in m.named_parameters() if k.startswith("experts.")]
        fired = []
        Fm.set_grad_ready_hook(lambda p: fired.append(id(p)))
        try:
            rec = []
            for b, fresh in zip(batches, reset):
                if fresh:
                    flat.attach_grads(fresh=True)
                fired.clear()
                y = m(b)
                (y.float() * torch.linspace(-1, 1, y.numel(), device=y.device).view_as(y)).sum().backward()
                torch.cuda.synchronize()
                assert sorted(i for i in fired if i in {id(p) for p in tower}) == sorted(id(p) for p in tower)
                rec.append((y.detach().clone(), flat.grad.detach().clone()))
        finally:
            Fm.set_grad_ready_hook(None)
        got[mode] = rec
        if mode == "1":
            ent = [e for k, e in m.experts._mm_graphs.items() if k[0] == "train"]
            assert len(ent) == 1 and set(ent[0].bwd) == {True, False}      # both backward graphs were captured and replayed
    for i, ((y0, g0), (y1, g1)) in enumerate(zip(got["0"], got["1"])):
        assert torch.equal(y0, y1), i
        if dtype == torch.bfloat16:
            assert torch.equal(g0, g1), (i, float((g0.float() - g1.float()).abs().max()))
        else:       # the fp32 parity kernels' attention backward sums dq with atomics: equal to rounding, not bit for bit, run to run
            assert float((g0 - g1).norm()) <= 1e-6 * float(g0.norm()), (i, float((g0 - g1).abs().max()))
    assert not torch.equal(got["1"][1][1], got["1"][2][1])


def test_moe_full_recipe_trainer_steps_graph_equals_eager(tmp_path, monkeypatch):
    """The MoE `full` recipe (trainable towers) under MultimodalTrainer, optimiser overlapped with the next forward: three
    steps with the towers replayed from graphs against three steps of eager launches from the same weights -- same losses, same
    parameters bit for bit (bf16).  Covers what a replay must do by hand: the Trainer's per-block parameter-read hooks before the
    forward replay, the gradient-ready hook after the backward replay, `.grad` / overwrite-vs-accumulate state."""
    import bench
    from tests.test_training_config_cpu import ATTACH, make_tokenizer
    from multimeditron_amd.train import from_training_config
    llm = os.path.join(str(tmp_path), "llm")                 # head width 64 everywhere: what the bf16 attention kernels hold
    os.makedirs(llm, exist_ok=True)
    json.dump(dict(model_type="llama", hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
                   num_key_value_heads=1, head_dim=64, vocab_size=32, rms_norm_eps=1e-5, tie_word_embeddings=False,
                   rope_parameters={"rope_type": "default", "rope_theta": 10000.0}), open(os.path.join(llm, "config.json"), "w"))
    clips = []
    for i in range(3):
        d = os.path.join(str(tmp_path), f"clip{i}")
        os.makedirs(d, exist_ok=True)
        json.dump({"vision_config": dict(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2, image_size=32,
                                         patch_size=16)}, open(os.path.join(d, "config.json"), "w"))
        json.dump({"size": {"shortest_edge": 32}, "crop_size": {"height": 32, "width": 32}}, open(os.path.join(d, "preprocessor_config.json"), "w"))
        clips.append(d)
    recipe = {
        "base_llm": llm, "base_model": None, "attachment_token": ATTACH, "tokenizer_type": "llama", "token_size": 128,
        "loaders": [{"loader_type": "raw-image", "modality_type": "image"}],
        "modalities": [{"model_type": "moe_meditron_clip_shared", "image_processor": clips[0], "hidden_size": 128, "expert_clip_names": clips,
                        "generalist_idx": -1, "gating_path": "stub", "fusion_method": "cross_attn", "top_k_experts": 3, "cross_attn_heads": 2}],
        "training_mode": "FULL",
        "training_args": {"learning_rate": 1.0e-3, "bf16": True, "per_device_train_batch_size": 2, "gradient_accumulation_steps": 2,
                          "max_steps": 50, "max_grad_norm": 1.0, "lr_scheduler_type": "constant", "weight_decay": 0.01},
    }
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("MM_MOE_TRAIN_GRAPH", mode)
        torch.manual_seed(0)
        setup = from_training_config(recipe, make_tokenizer(), device="cuda", dtype="bfloat16")
        model, tr = setup.model, setup.trainer

        def gate(px):                    # test-side stand-in for the reference's ResNet gate (a plug: DESIGN.md section 7)
            logits = px.float().mean(dim=(2, 3)) @ torch.tensor([[1.0, 0.5, -0.5], [0.2, -0.3, 0.7], [-0.4, 0.9, 0.1]], device=px.device)
            return logits, logits.topk(1, dim=-1).indices, torch.softmax(logits, dim=-1)

        model.modalities_by_type["image"].gating_network = gate
        model.eval()                     # no dropout: the two runs must agree bit for bit
        model.train = lambda *_a, **_k: model
        vocab = model.config.vocab_size
        losses = []
        for step in range(6):            # 6 micro-batches = 3 optimiser steps (the second micro-batch of a step accumulates)
            batch, _ = bench.synthetic_batch(2, 24, 1, 4, vocab, (vocab - 3, vocab - 2, vocab - 1), 100 + step, "cpu", 32, collator_form=True)
            batch["input_ids"].clamp_(max=vocab - 1)
            batch["labels"] = torch.where(batch["labels"] >= 0, batch["labels"].clamp(max=vocab - 1), batch["labels"])
            losses.append(float(tr.training_step(batch)))
        tr.synchronize()
        torch.cuda.synchronize()
        mod = model.modalities_by_type["image"]
        if mode == "1":
            ent = [e for k, e in mod.experts._mm_graphs.items() if k[0] == "train"]
            assert len(ent) == 1 and set(ent[0].bwd) == {True, False}
        res[mode] = (losses, tr.flat.data.detach().clone())
        tr.close()
    assert res["0"][0] == res["1"][0], (res["0"][0], res["1"][0])
    assert torch.equal(res["0"][1], res["1"][1])
    assert all(l == l for l in res["0"][0])


@pytest.mark.parametrize("pep", [False, True])
def test_two_forwards_before_one_backward(moe, moe_pep, tmp_path, pep, monkeypatch):
    """the trainable towers' hipGraphs share their saved activations per pixel shape.  Two forwards of the same shape whose
    losses are summed before ONE backward must not let the second forward overwrite what the first one's backward reads: the second
    call takes the eager launches (its graph entry is `pending`), and the result equals the all-eager run bit for bit (bf16).  A second
    backward through a replayed forward raises instead of reading refilled buffers."""
    meta, w, v = moe_pep if pep else moe
    px = v["pixels"]
    g = torch.Generator().manual_seed(9)
    b0 = [px[i] for i in range(px.shape[0])]
    b1 = [torch.randn_like(px[0], generator=g) for _ in range(px.shape[0])]
    got = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("MM_MOE_TRAIN_GRAPH", mode)
        m = _build(meta, w, v, "weighted_average", torch.bfloat16, tmp_path / f"two{mode}")
        flat = next(iter(m.parameters()))._mm_flat
        for rep in range(2):                      # rep 0: first (eager-through-capture) backward; rep 1: replayed graphs
            flat.attach_grads(fresh=True)
            y0, y1 = m(b0), m(b1)
            wgt = torch.linspace(-1, 1, y0.numel(), device=y0.device).view_as(y0)
            ((y0.float() * wgt).sum() + (y1.float() * wgt.flip(0)).sum()).backward()
            torch.cuda.synchronize()
        got[mode] = (y0.detach().clone(), y1.detach().clone(), flat.grad.detach().clone())
        if mode == "1":
            ent = [e for k, e in m.experts._mm_graphs.items() if k[0] == "train"][0]
            assert ent.pending is None
            flat.attach_grads(fresh=True)
            y = m(b0)
            s = (y.float() * wgt).sum()
            s.backward(retain_graph=True)
            with pytest.raises(RuntimeError, match="saved activations are gone"):
                s.backward()
    for a, b in zip(got["0"], got["1"]):
        assert torch.equal(a, b), float((a.float() - b.float()).abs().max())
