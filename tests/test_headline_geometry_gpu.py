"""The HEADLINE layer geometry against the oracle (GPU) -- VERDICT r1 "Next round" item 2.

The committed reference fixtures are tiny (hidden 128); the kernels the benchmark actually runs (D = 128 attention fwd /
dQ / paired dK/dV, the 256x256 GEMM at K = 4096 / 14336, vocab 128 258 cross-entropy, llama3 RoPE, GQA 4 and 7, QKV
bias) used to be checked per kernel against fp32 torch only.  Here a 2-layer SLICE of the real models

    Llama-3.1-8B   (H 4096, 32/8 heads x 128, I 14336, vocab 128 258, llama3 RoPE) + 2-layer ViT-L/14 + MLP projector
    Qwen2-7B       (H 3584, 28/4 heads x 128 = GQA 7, I 18944, QKV bias, vocab 152 066) + 2-layer SigLIP-so400m/14@384

runs fwd + bwd in bf16 on the HIP path and is compared with oracle/ref_cpu.py in fp32 on the SAME bf16-rounded weights
(the oracle is pinned to the real reference by tests/test_oracle_golden.py, including a head_dim-128 / GQA-4 fixture).
B = 2, one image per sample, second row RIGHT-PADDED (key mask + ignored labels).  Criteria = test_config1_shapes_vs_oracle
and test_bf16_grads: logits rel-L2 <= 3e-2 on valid rows, |loss diff| <= 3e-2, every arg-max disagreement a near-tie of
the oracle's logits, gradients rel-L2 <= 6e-2 (decoder layer 1, projector, touched embedding rows, one ViT layer)."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _build(tmp, llm_name, clip_name, n_layers=2):
    from multimeditron_amd.model.model import MultimodalConfig, MultiModalModelForCausalLM
    from multimeditron_amd.model.modalities import ImageConfig, SiglipImageConfig
    from multimeditron_amd.model.presets import resolve_llm_config, resolve_vision_config
    llm = dict(resolve_llm_config(llm_name), num_hidden_layers=n_layers)
    vis = dict(resolve_vision_config(clip_name), num_hidden_layers=n_layers)
    d = os.path.join(str(tmp), "tower")
    os.makedirs(d, exist_ok=True)
    size = vis["image_size"]
    if vis.get("kind") == "siglip":
        json.dump(dict(vis, model_type="siglip_vision_model"), open(os.path.join(d, "config.json"), "w"))
        mod = SiglipImageConfig(hidden_size=llm["hidden_size"], clip_name=d)
    else:
        json.dump({"vision_config": vis}, open(os.path.join(d, "config.json"), "w"))
        json.dump({"size": {"shortest_edge": size}, "crop_size": {"height": size, "width": size}},
                  open(os.path.join(d, "preprocessor_config.json"), "w"))
        mod = ImageConfig(hidden_size=llm["hidden_size"], clip_name=d)
    torch.manual_seed(7)
    cfg = MultimodalConfig(vocab_size=llm["vocab_size"] + 2, modalities=[mod], llm_path="unused", dtype="bfloat16",
                           eos_token_idx=128009, hidden_size=llm["hidden_size"])
    m = MultiModalModelForCausalLM(cfg, device="cuda", llm_config=llm)
    g = torch.Generator(device="cuda").manual_seed(11)
    with torch.no_grad():            # default init leaves biases 0 and norms 1: perturb so those paths carry signal
        for n, p in m.named_parameters():
            if p.dim() == 1:
                p.add_((0.05 * torch.randn(p.shape, generator=g, device="cuda")).to(p.dtype))
    m.pack_parameters()
    m.unfreeze()
    return m, llm, vis


def _batch(B, S, P, vocab, img, pad_from, seed):
    """synthetic batch in the collator's form: 1 image per sample, row 1 right-padded from `pad_from`."""
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(0, vocab - 2, (B, S), generator=g)
    ids[:, :: 5] = ids[:, :: 5] % 40          # repeated ids: the embedding gradient sums many tokens per row
    labels = ids.clone()
    mask = torch.ones(B, S, dtype=torch.long)
    bi, tr, pix = [], [], []
    for b in range(B):
        s = 6 + 3 * b
        ids[b, s - 1], ids[b, s + P] = vocab - 2, vocab - 1
        ids[b, s:s + P] = 128002
        labels[b, s - 1:s + P + 1] = -100
        labels[b, :4] = -100
        bi += [b] * P
        tr += list(range(s, s + P))
        pix.append(torch.randn(3, img, img, generator=g).to(torch.bfloat16).float())
    mask[1, pad_from:] = 0
    ids[1, pad_from:] = 128009
    labels[1, pad_from:] = -100
    pos = (mask.cumsum(-1) - 1).masked_fill(mask == 0, 0)
    return dict(input_ids=ids, labels=labels, attention_mask=mask, position_ids=pos,
                processed_multimodal_inputs={"batch_idx": {"image": torch.tensor(bi)}, "token_range": {"image": torch.tensor(tr)},
                                             "stacked": {"image": pix}})


GEOMS = {
    "llama31_8b+vitl14": ("meta-llama/Llama-3.1-8B-Instruct", "openai/clip-vit-large-patch14", 512, 400),
    "qwen2_7b+siglip_so400m": ("Qwen/Qwen2-7B-Instruct", "google/siglip-so400m-patch14-384", 1024, 900),
}


@pytest.mark.parametrize("geom", sorted(GEOMS))
def test_layer_slice_fwd_bwd_vs_oracle(geom, tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import ref_cpu as R
    from tests.model_utils import to_device
    llm_name, clip_name, S, pad_from = GEOMS[geom]
    m, llm, vis = _build(tmp_path, llm_name, clip_name)
    V = llm["vocab_size"] + 2
    P = (vis["image_size"] // vis["patch_size"]) ** 2
    cb = _batch(2, S, P, V, vis["image_size"], pad_from, 3)
    gb = to_device(cb)
    out = m(input_ids=gb["input_ids"], attention_mask=gb["attention_mask"], position_ids=gb["position_ids"], labels=gb["labels"],
            processed_multimodal_inputs=gb["processed_multimodal_inputs"])
    out.loss.backward()
    torch.cuda.synchronize()
    got_logits = out.logits.float().cpu()
    params = dict(m.named_parameters())

    siglip = vis.get("kind") == "siglip"
    vit_layer = ("modalities_with_projection.0.feature_extractor." + ("" if siglip else "vision_model.")) + "encoder.layers.1."
    want = lambda n: (n.startswith("model.model.layers.1.") or ".projector." in n or n == "model.model.embed_tokens.weight"  # noqa: E731
                      or n.startswith(vit_layer) or n == "model.model.norm.weight")
    w = {}
    for n, p in params.items():
        t = p.detach().float().cpu()
        w[n] = t.requires_grad_(True) if want(n) else t
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    ref_logits, ref_loss = R.multimodal_forward(w, cb, {"vision": vis, "llm": llm})
    ref_loss.backward()
    ref_logits = ref_logits.detach()

    valid = cb["attention_mask"].bool()
    e = rel(got_logits[valid], ref_logits[valid])
    assert e < 3e-2, e
    assert abs(float(out.loss) - float(ref_loss)) < 3e-2, (float(out.loss), float(ref_loss))
    ga, ra = got_logits[valid].argmax(-1), ref_logits[valid].argmax(-1)
    rl = ref_logits[valid]
    gap = (rl.gather(-1, ra[:, None]) - rl.gather(-1, ga[:, None])).squeeze(-1)
    assert float((gap / rl.std(-1)).max()) < 0.1          # every disagreement is a near-tie of the oracle's own scores
    assert float((ga == ra).float().mean()) >= 0.85

    worst, checked = 0.0, 0
    for n, t in w.items():
        if not t.requires_grad:
            continue
        assert params[n].grad is not None, n
        g, r = params[n].grad.float().cpu(), t.grad
        if n == "model.model.embed_tokens.weight":        # touched rows only (the rest is exactly zero on both sides)
            rows = torch.unique(cb["input_ids"])
            untouched = torch.ones(g.shape[0], dtype=torch.bool)
            untouched[rows] = False
            assert float(g[untouched].abs().max()) == 0.0 and float(r[untouched].abs().max()) == 0.0
            g, r = g[rows], r[rows]
        err = rel(g, r)
        tol = 6e-2 if t.dim() >= 2 else 1e-1               # vectors (norm weights, biases): sums of many bf16 rows
        if n.endswith("k_proj.bias") and "feature_extractor" in n:   # ViT (no RoPE): analytically zero by softmax shift
            qb = w[n.replace("k_proj", "q_proj")].grad             # invariance, so both sides hold rounding noise only
            assert float(g.norm()) < 0.05 * float(qb.norm()) + 1e-6, n
            continue
        assert err < tol, (n, err)
        worst = max(worst, err)
        checked += 1
    assert checked >= 12 and worst > 0
