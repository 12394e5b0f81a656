"""DevicePrefetcher: batches staged on a side stream arrive bit-identical to a plain `.to(device)`, in order,
with the all-ones flag computed on the host copy, and the model consumes them (pixel list -> one device stack)."""
import pytest
import torch

from oracle import ref_cpu as R
from tests.model_utils import build_from_golden, to_device

pytestmark = pytest.mark.gpu


def test_prefetched_batches_identical_and_ordered(golden_dir, tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from multimeditron_amd.train.prefetch import DevicePrefetcher
    meta, w, v = R.load_golden("tiny_clip_llama", golden_dir)
    cases = ["right", "interleaved4", "textonly", "left", "right", "interleaved4"]       # varying shapes: pinned slots are re-sized
    host = [R.golden_batch(v, c) for c in cases]
    got = list(DevicePrefetcher(iter(host), device="cuda"))
    assert len(got) == len(host)
    for h, g in zip(host, got):
        for k in ("input_ids", "labels", "attention_mask", "position_ids"):
            assert g[k].is_cuda and torch.equal(g[k].cpu(), h[k])
        assert g["attention_mask"]._mm_all_ones == bool(h["attention_mask"].all())
        pm_h, pm_g = h["processed_multimodal_inputs"], g["processed_multimodal_inputs"]
        for name in ("batch_idx", "token_range"):
            assert set(pm_g[name]) == set(pm_h[name])
            for t in pm_h[name]:
                assert torch.equal(pm_g[name][t].cpu(), pm_h[name][t])
        for t, vals in pm_h["stacked"].items():
            assert pm_g["stacked"][t].is_cuda and torch.equal(pm_g["stacked"][t].cpu(), torch.stack(list(vals)))


def test_model_on_prefetched_batch_equals_direct(golden_dir, tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from multimeditron_amd.train.prefetch import DevicePrefetcher
    meta, w, v = R.load_golden("tiny_clip_llama", golden_dir)
    model = build_from_golden(meta, w, tmp_path, "bfloat16")
    for case in ("right", "interleaved4"):          # a padded batch (mask kept) and an unpadded one (all-ones mask dropped)
        hb = R.golden_batch(v, case)
        db = to_device(hb)
        pb = next(DevicePrefetcher(iter([hb]), device="cuda"))
        with torch.no_grad():
            a = model(input_ids=db["input_ids"], attention_mask=db["attention_mask"], position_ids=db["position_ids"], labels=db["labels"],
                      processed_multimodal_inputs=db["processed_multimodal_inputs"])
            b = model(input_ids=pb["input_ids"], attention_mask=pb["attention_mask"], position_ids=pb["position_ids"], labels=pb["labels"],
                      processed_multimodal_inputs=pb["processed_multimodal_inputs"])
        valid = hb["attention_mask"].bool()
        assert torch.equal(a.logits.cpu()[valid], b.logits.cpu()[valid]) and torch.equal(a.loss, b.loss)
