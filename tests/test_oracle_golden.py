"""Pins oracle/ref_cpu.py to vectors produced by the REAL reference (tools/make_golden.py).

fp32 everywhere; tolerance 2e-5 relative-L2 on activations/logits, 1e-4 on grads; greedy ids exact.
"""
import pytest
import torch

from oracle import ref_cpu as R

MODELS = ["tiny_clip_llama", "tiny_clip_qwen2", "tiny_siglip_qwen2", "tiny_clip_llama_d128"]   # tiny_siglip_qwen2: BASELINE config 5 plug-in (SigLIP tower); tiny_clip_llama_d128: the headline attention geometry (head_dim 128, GQA 4) at S up to 330, produced by the reference


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


@pytest.fixture(scope="module", params=MODELS)
def gold(request, golden_dir):
    meta, w, v = R.load_golden(request.param, golden_dir)
    w = {k: t.float() for k, t in w.items()}
    return meta, w, v


@pytest.mark.parametrize("case", ["right", "left", "textonly", "interleaved4"])
def test_forward_stages_logits_loss(gold, case):
    meta, w, v = gold
    if case not in meta["cases"]:
        pytest.skip(f"{meta['name']} holds no '{case}' case")
    batch = R.golden_batch(v, case)
    stages = {}
    with torch.no_grad():
        logits, loss = R.multimodal_forward(w, batch, meta, stages)
    valid = batch["attention_mask"].bool()
    for name, t in stages.items():
        key = f"{case}.act.{name}"
        assert key in v, key
        ref = v[key]
        if name.startswith("llm") :
            assert rel(t[valid], ref[valid]) < 2e-5, name
        else:
            assert rel(t, ref) < 2e-5, name
    assert rel(logits[valid], v[f"{case}.logits"][valid]) < 2e-5
    assert abs(float(loss) - float(v[f"{case}.loss"])) < 2e-5 * max(1.0, abs(float(loss)))
    # arg-max agreement on valid rows
    assert torch.equal(logits[valid].argmax(-1), v[f"{case}.logits"][valid].argmax(-1))


def test_grads(gold):
    meta, w, v = gold
    wg = {k: t.clone().requires_grad_(True) for k, t in w.items()}
    batch = R.golden_batch(v, "right")
    _, loss = R.multimodal_forward(wg, batch, meta)
    loss.backward()
    checked = 0
    for key in v:
        if not key.startswith("right.grad."):
            continue
        name = key[len("right.grad."):]
        if name == "model.lm_head.weight" and meta["llm"].get("tie_word_embeddings"):
            continue  # tied: the reference reports the same (summed) grad under both names
        g = wg[name].grad
        assert g is not None, name
        # k_proj.bias grads are analytically zero (softmax shift invariance): absolute floor
        assert float((g - v[key]).norm()) <= 1e-4 * float(v[key].norm()) + 1e-7, name
        checked += 1
    assert checked > 20


@pytest.mark.parametrize("case", ["left", "textonly"])
@pytest.mark.parametrize("T", [0.1, 0.7])
def test_greedy_ids_bit_exact(gold, case, T):
    meta, w, v = gold
    if case not in meta["cases"]:
        pytest.skip(f"{meta['name']} holds no '{case}' case")
    ids = R.greedy_generate(w, R.golden_batch(v, case), meta, max_new_tokens=8, temperature=T)
    assert torch.equal(ids, v[f"{case}.greedy_T{T}"])
