"""End-to-end optimisation parity (GPU): MultimodalTrainer (libmmhip fwd + bwd + fused clip/AdamW) against the CPU oracle
driven by torch autograd + torch.optim.AdamW + clip_grad_norm_ on the same batches.  fp32 path, FULL mode.
Tolerance: loss sequence |d| <= 2e-4, final parameters rel-L2 <= 2e-3 (3 steps, lr 1e-3)."""
import pytest
import torch

from oracle import ref_cpu as R
from tests.model_utils import build_from_golden, to_device

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["tiny_clip_llama", "tiny_clip_qwen2"])
@pytest.mark.parametrize("mode", ["FULL", "ALIGNMENT", "END2END", "LM_ONLY"])
@pytest.mark.parametrize("loss_rows", [True, False], ids=["labelled_rows", "all_rows"])
def test_training_steps_match_oracle(golden_dir, tmp_path, name, mode, loss_rows):
    """loss_rows: final norm + lm_head + loss on the labelled rows only (the Trainer's default) or on every row as HF computes
    them -- the oracle always does the latter; both must meet the same bounds."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from multimeditron_amd.train.trainer import MultimodalTrainer, TrainingMode
    meta, w, v = R.load_golden(name, golden_dir)
    model = build_from_golden(meta, w, tmp_path, "float32")
    tr = MultimodalTrainer(model, training_mode=TrainingMode[mode], learning_rate=1e-3, weight_decay=0.01, betas=(0.9, 0.95),
                           max_grad_norm=1.0, gradient_accumulation_steps=1, loss_rows_only=loss_rows)
    cases = ["right", "textonly", "interleaved4"]     # the text-only step leaves the vision tower without gradients
    losses = [float(tr.training_step(to_device(R.golden_batch(v, c)))) for c in cases]
    tr.synchronize()
    torch.cuda.synchronize()

    # oracle: same weights, torch autograd + AdamW on CPU
    tied = bool(meta["llm"].get("tie_word_embeddings"))
    wt = {k: t.float().clone().requires_grad_(True) for k, t in w.items() if not (tied and k == "model.lm_head.weight")}
    # reference trainer.py:132-144 / model.py:310-377: what each mode trains
    is_train = {"FULL": lambda k: True, "ALIGNMENT": lambda k: ".projector." in k,
                "END2END": lambda k: ".projector." in k or k.startswith("model."), "LM_ONLY": lambda k: k.startswith("model.")}[mode]
    trainable = {k: p for k, p in wt.items() if is_train(k)}
    for k, p in wt.items():
        p.requires_grad_(k in trainable)
    # HF Trainer's weight-decay grouping (tests/test_schedule_cpu.py checks the product's grouping against HF's own function):
    # everything except LayerNorm-module parameters and names matching bias | norm -- CLIP's 1-D class_embedding IS decayed
    import re
    no_decay = re.compile(r"bias|layernorm|layrnorm|rmsnorm|(?:^|\.)norm(?:$|\.)|_norm")
    decay = [p for k, p in trainable.items() if not no_decay.search(k.lower())]
    nodecay = [p for k, p in trainable.items() if no_decay.search(k.lower())]
    opt = torch.optim.AdamW([{"params": decay, "weight_decay": 0.01}, {"params": nodecay, "weight_decay": 0.0}], lr=1e-3,
                            betas=(0.9, 0.95), eps=1e-8)
    ref_losses = []
    for c in cases:
        opt.zero_grad(set_to_none=False)
        _, loss = R.multimodal_forward(wt, R.golden_batch(v, c), meta)
        if loss.requires_grad:     # ALIGNMENT + text-only batch: no trainable parameter on the path (zero gradients)
            loss.backward()
        torch.nn.utils.clip_grad_norm_(list(trainable.values()), 1.0)
        opt.step()
        ref_losses.append(float(loss))
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) < 2e-4, (losses, ref_losses)
    params = dict(model.named_parameters())
    for k, p in trainable.items():
        got = params[k].detach().float().cpu()
        err = float((got - p.detach()).norm() / (p.detach().norm() + 1e-12))
        assert err < 2e-3, (k, err)
    if mode == "ALIGNMENT":   # frozen parts untouched
        assert torch.equal(params["model.model.norm.weight"].cpu().float(), w["model.model.norm.weight"].float())
    if mode in ("END2END", "LM_ONLY"):
        k = next(n for n in params if "vision_model.encoder.layers.0.mlp.fc1.weight" in n)
        assert torch.equal(params[k].cpu().float(), w[k].float())          # the modality embedder stays frozen
    if mode == "LM_ONLY":
        k = next(n for n in params if ".projector.projection.0.weight" in n)
        assert torch.equal(params[k].cpu().float(), w[k].float())


@pytest.mark.parametrize("name", ["tiny_clip_llama", "tiny_siglip_qwen2"])
def test_optimizer_overlap_hooks_cover_every_block(golden_dir, tmp_path, name):
    """ADVICE r1 (trainer.py:245): the side-stream AdamW of step n is guarded, block by block, by forward pre-hooks in step
    n+1.  Every block whose parameters an image batch's forward reads must have FIRED its hook by the end of that forward
    (embed_tokens is read by the splice, patch/position embeddings by VisionEmbeddings.forward: neither module used to be
    __call__ed).  `_unfired` lists the blocks still pending when backward starts."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from multimeditron_amd.train.trainer import MultimodalTrainer, TrainingMode
    try:
        meta, w, v = R.load_golden(name, golden_dir)
    except FileNotFoundError:
        pytest.skip(f"fixture {name} not present")
    model = build_from_golden(meta, w, tmp_path, "bfloat16")
    tr = MultimodalTrainer(model, training_mode=TrainingMode.FULL, learning_rate=1e-3)
    assert all(mod is not None for mod, _ in tr._blocks), "a trainable block has no call site: its update would be waited eagerly"
    b = to_device(R.golden_batch(v, "right"))
    tr.training_step(b)                    # launches the overlapped update
    assert tr._pending, "the optimizer pipeline did not arm any block"
    tr.training_step(b)                    # forward must pop every block before backward starts
    assert tr._unfired == [], [type(m).__name__ for m, _ in tr._blocks if id(m) in tr._unfired]
    tr.synchronize()
    torch.cuda.synchronize()


@pytest.mark.parametrize("variant", ["fused", "early", "sweep"])
def test_gradnorm_assembly_equals_full_norm(golden_dir, tmp_path, monkeypatch, variant):
    """The global grad norm is assembled from the wgrad GEMMs' own sum-of-squares slots (mm_gemm_sumsq: decoder matrices,
    lm_head) plus a sweep of the rest ("fused", MM_FUSED_NORM=1), or from per-layer sweeps under backward ("early", an experiment
    kept behind MM_EARLY_NORM=1), or from one sweep ("sweep", the default since both alternatives measured slower).  Each must equal the norm of the complete flat gradient and be
    identical run to run (fixed slots, fixed summation order); deferred and immediate wgrads both take part."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    monkeypatch.setenv("MM_DEFER_WGRAD_LAYERS", "1")          # layer 0's wgrads deferred (side stream), layer 1's immediate
    monkeypatch.setenv("MM_FUSED_NORM", "1" if variant == "fused" else "0")
    monkeypatch.setenv("MM_EARLY_NORM", "1" if variant == "early" else "0")
    from multimeditron_amd.train.trainer import MultimodalTrainer, TrainingMode
    meta, w, v = R.load_golden("tiny_clip_llama", golden_dir)
    norms = []
    for rep in range(2):
        model = build_from_golden(meta, w, tmp_path / f"m{rep}", "bfloat16")
        tr = MultimodalTrainer(model, training_mode=TrainingMode.FULL, learning_rate=0.0, weight_decay=0.0, max_grad_norm=1.0)
        if variant == "fused":
            assert tr._ss is not None and tr._ss.numel() > 0 and not tr._norm_triggers
            fused_elems = sum(e - s for s, e, _ in tr.ranges) - sum(e - s for s, e, _ in tr._norm_chunks)
            assert fused_elems > 0.5 * sum(p.numel() for p in model.model.model.layers.parameters())
        elif variant == "early":
            assert tr._norm_triggers and tr._ss is None
        else:
            assert tr._ss is None and not tr._norm_triggers and len(tr._norm_chunks) == len(tr.ranges)
        spans = sorted((s, e) for s, e, _ in tr._norm_chunks)
        assert all(a[1] <= b[0] for a, b in zip(spans[:-1], spans[1:])), "chunks overlap"
        for _ in range(2):                                     # twice: slots are reused step after step
            tr.training_step(to_device(R.golden_batch(v, "right")))
        tr.synchronize()
        torch.cuda.synchronize()
        g = tr.flat.grad
        ref = torch.sqrt(sum((g[s:e].double() ** 2).sum() for s, e, _ in tr.ranges))
        got = tr.last_grad_norm[0].double()
        assert abs(float(got) - float(ref)) < 1e-5 * float(ref), (variant, float(got), float(ref))
        norms.append(tr.last_grad_norm.clone())
    assert torch.equal(norms[0], norms[1])


@pytest.mark.parametrize("dtype", ["bfloat16", "float32"])
def test_resume_is_bit_exact(golden_dir, tmp_path, dtype):
    """VERDICT r1 item 8 (reference cli/train.py:186-195, `trainer.train(resume_from_checkpoint=...)`): 4 optimiser steps ==
    2 steps + save_state + a NEW process-equivalent (fresh model object, fresh trainer, load_state) + 2 steps, bit for bit:
    parameters, fp32 master weights, AdamW moments, step counter, losses."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from multimeditron_amd.train.trainer import MultimodalTrainer, TrainingMode
    meta, w, v = R.load_golden("tiny_clip_llama", golden_dir)
    cases = ["right", "interleaved4", "textonly", "left"]
    kw = dict(training_mode=TrainingMode.FULL, learning_rate=1e-3, weight_decay=0.01, betas=(0.9, 0.95), max_grad_norm=1.0,
              max_steps=8, min_lr=1e-4, warmup_steps=1)

    def fresh():
        m = build_from_golden(meta, w, tmp_path / f"m{len(os.listdir(tmp_path))}", dtype)
        return m, MultimodalTrainer(m, **kw)

    import os
    m_a, t_a = fresh()
    la = [float(t_a.training_step(to_device(R.golden_batch(v, c)))) for c in cases]
    t_a.synchronize()
    torch.cuda.synchronize()

    m_b, t_b = fresh()
    lb = [float(t_b.training_step(to_device(R.golden_batch(v, c)))) for c in cases[:2]]
    ck = tmp_path / "checkpoint-2"
    t_b.save_state(str(ck))
    assert (ck / "trainer_state.json").exists() and (ck / "optimizer_state.safetensors").exists() and (ck / "config.json").exists()
    del m_b, t_b
    m_c, t_c = fresh()                        # a different random-free start would do too: everything comes from the checkpoint
    with torch.no_grad():
        for p in m_c.parameters():
            p.add_(1.0)                       # make sure the resumed run does not lean on the constructor's weights
    st = t_c.load_state(str(ck))
    assert st["global_step"] == 2 and t_c.step_count == 2
    lb += [float(t_c.training_step(to_device(R.golden_batch(v, c)))) for c in cases[2:]]
    t_c.synchronize()
    torch.cuda.synchronize()
    assert t_a.step_count == t_c.step_count == 4
    if dtype == "bfloat16":       # the throughput path: every kernel is deterministic, so the resumed run is bit-identical
        assert la == lb, (la, lb)
        for (k1, p1), (k2, p2) in zip(m_a.named_parameters(), m_c.named_parameters()):
            assert k1 == k2 and torch.equal(p1, p2), k1
        assert torch.equal(t_a.master, t_c.master) and torch.equal(t_a.m, t_c.m) and torch.equal(t_a.v, t_c.v)
    else:                         # the fp32 parity path sums dK/dV with float atomics (order varies run to run): equal to rounding
        assert max(abs(a - b) for a, b in zip(la, lb)) < 1e-5, (la, lb)
        for (k1, p1), (k2, p2) in zip(m_a.named_parameters(), m_c.named_parameters()):     # Adam divides by sqrt(v): an element whose
            assert k1 == k2 and float((p1 - p2).norm()) <= 1e-4 * float(p1.norm()) + 1e-7, k1   # gradient is rounding noise may move by ~lr
        assert float((t_a.master - t_c.master).norm()) < 1e-4 * float(t_a.master.norm())


@pytest.mark.parametrize("dtype", ["bfloat16", "float32"])
def test_loss_on_labelled_rows_equals_loss_on_all_rows(golden_dir, tmp_path, dtype):
    """Trainer.compute_loss with `loss_rows` (final norm, lm_head and cross-entropy on the rows whose shifted label is not -100)
    against the HF form (every row's logits, the loss ignores the -100 rows): same loss and the same gradient for EVERY parameter.
    fp32: <= 1e-6 relative (the lm_head wgrad sums the same products with the zero rows left out); bf16: <= 2e-3 of the gradient's
    norm (the K order of the wgrad changes where its partial sums round)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from multimeditron_amd.train.trainer import MultimodalTrainer, TrainingMode
    meta, w, v = R.load_golden("tiny_clip_llama", golden_dir)
    got = {}
    for rows in (True, False):
        model = build_from_golden(meta, w, tmp_path / f"rows{int(rows)}", dtype)
        tr = MultimodalTrainer(model, training_mode=TrainingMode.FULL, learning_rate=0.0, weight_decay=0.0, max_grad_norm=1.0,
                               loss_rows_only=rows, overlap_optimizer=False)
        losses = []
        for case in ("right", "interleaved4"):
            b = to_device(R.golden_batch(v, case))
            assert (b["labels"]._mm_loss_rows.n < b["labels"].numel()) and b["labels"]._mm_loss_rows.n > 0
            losses.append(float(tr.training_step(b)))
            tr.synchronize()
            torch.cuda.synchronize()
            losses.append(torch.cat([tr.flat.grad[s0:e0].detach().float() for s0, e0, _ in tr.ranges]))
        got[rows] = losses
    tol_loss, tol_grad = (1e-6, 1e-6) if dtype == "float32" else (2e-3, 2e-3)
    for a, b in zip(got[True], got[False]):
        if isinstance(a, float):
            assert abs(a - b) <= tol_loss * max(1.0, abs(b)), (a, b)
        else:
            assert float((a - b).norm()) <= tol_grad * float(b.norm()), float((a - b).norm() / b.norm())
            assert float(b.norm()) > 0
