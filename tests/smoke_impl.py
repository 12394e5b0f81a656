"""smoke(): one tiny forward+backward (+ one optimizer step) of the hot path on cuda:0, checked against the CPU oracle
on the same fixture inputs.  Called by __graft_entry__.smoke()."""
import os
import tempfile

import torch


def run_smoke():
    from oracle import ref_cpu as R
    from tests.model_utils import build_from_golden, to_device
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    meta, w, v = R.load_golden("tiny_clip_llama", golden)
    batch = R.golden_batch(v, "right")
    wf = {k: t.float() for k, t in w.items()}
    with torch.no_grad():
        ref_logits, ref_loss = R.multimodal_forward(wf, batch, meta)
    valid = batch["attention_mask"].bool()
    with tempfile.TemporaryDirectory() as tmp:
        for dtype, tol in (("float32", 1e-4), ("bfloat16", 3e-2)):
            model = build_from_golden(meta, w, os.path.join(tmp, dtype), dtype)
            model.unfreeze()
            gb = to_device(batch)
            out = model(input_ids=gb["input_ids"], attention_mask=gb["attention_mask"], position_ids=gb["position_ids"],
                        labels=gb["labels"], processed_multimodal_inputs=gb["processed_multimodal_inputs"])
            out.loss.backward()
            torch.cuda.synchronize()
            a, b = out.logits.float().cpu()[valid].double(), ref_logits[valid].double()
            err = float((a - b).norm() / b.norm())
            assert err < tol, (dtype, err)
            assert abs(float(out.loss) - float(ref_loss)) < max(tol, 1e-4) * 2, (dtype, float(out.loss), float(ref_loss))
            g = model.modalities_with_projection[0].projector.projection[4].weight.grad
            assert g is not None and torch.isfinite(g.float()).all() and float(g.float().abs().sum()) > 0
            print(f"smoke {dtype}: logits rel-L2 {err:.2e} loss {float(out.loss):.5f} (oracle {float(ref_loss):.5f})")
    print("smoke ok")
