"""BASELINE config 1 as a pipeline test: the reference's mock_dataset samples go through DataCollatorForMultimodal (CPU), the
collated batch through MultiModalModelForCausalLM.forward / generate and MultimodalTrainer.training_step on the GPU, and the
SAME collated batch through the CPU oracle.  fp32 model (the parity path): logits <= 1e-4 rel-L2 on non-pad rows, greedy ids
bit-exact; a few ALIGNMENT-mode steps on the bf16 model lower the loss."""
import copy

import pytest
import torch

from oracle import ref_cpu as R
from tests.model_utils import build_from_golden, to_device
from tests.test_collator_golden import env  # noqa: F401  (fixture: tokenizer, processor, chat template, mock images)
from multimeditron_amd.dataset.loader import AutoModalityLoader
from multimeditron_amd.model.data_loader import DataCollatorForMultimodal

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _collate(env, side, gen, threads=0):      # noqa: F811
    meta, vec, make_tok, proc, ct, imgdir = env
    coll = DataCollatorForMultimodal(tokenizer=make_tok(side), modality_processors={"image": proc},
                                     modality_loaders={"image": AutoModalityLoader.from_name("fs-image", base_path=imgdir)},
                                     attachment_token=meta["attachment_token"], chat_template=ct, add_generation_prompt=gen,
                                     num_threads=threads)
    return coll(copy.deepcopy(meta["samples_conv"]))


@pytest.mark.parametrize("side", ["right", "left"])
def test_collated_batch_forward_and_generate_match_oracle(env, golden_dir, tmp_path, side):      # noqa: F811
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    meta, w, v = R.load_golden("tiny_clip_llama", golden_dir)
    model = build_from_golden(meta, w, tmp_path, "float32")
    batch = _collate(env, side, gen=(side == "left"), threads=2)
    # the collator's token ids come from a 40-word vocabulary; the tiny model has 130 embedding rows: ids are in range
    assert int(batch["input_ids"].max()) < meta["vocab_size"]
    wf = {k: t.float() for k, t in w.items()}
    with torch.no_grad():
        ref_logits, ref_loss = R.multimodal_forward(wf, batch, meta)
        gb = to_device(batch)
        out = model(input_ids=gb["input_ids"], attention_mask=gb["attention_mask"], position_ids=gb["position_ids"],
                    labels=gb["labels"], processed_multimodal_inputs=gb["processed_multimodal_inputs"])
    valid = batch["attention_mask"].bool()
    assert rel(out.logits.cpu()[valid], ref_logits[valid]) < 1e-4
    assert abs(float(out.loss) - float(ref_loss)) < 1e-4
    if side == "left":      # generation prompt + left padding: what generate() is called with in the reference
        ids = model.generate(batch, max_new_tokens=6, temperature=0.1, do_sample=False)
        ref_ids = R.greedy_generate(wf, batch, meta, max_new_tokens=6, temperature=0.1)
        assert torch.equal(ids, ref_ids)


def test_alignment_training_on_collated_batches_lowers_the_loss(env, golden_dir, tmp_path):      # noqa: F811
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from multimeditron_amd.train.trainer import MultimodalTrainer, TrainingMode
    meta, w, v = R.load_golden("tiny_clip_llama", golden_dir)
    model = build_from_golden(meta, w, tmp_path, "bfloat16")
    trainer = MultimodalTrainer(model, training_mode=TrainingMode.ALIGNMENT, learning_rate=2e-3, weight_decay=0.0, max_steps=20)
    batch = to_device(_collate(env, "right", gen=False))
    losses = [float(trainer.training_step(batch)) for _ in range(12)]
    trainer.synchronize()
    assert all(torch.isfinite(torch.tensor(losses)))
    assert losses[-1] < losses[0] - 0.05, losses      # only the projector moves (ALIGNMENT), on one batch: a clear decrease
