#!/usr/bin/env python3
"""Collator throughput, reference vs this build, on the reference's mock_dataset samples (CPU; build container only:
the reference is imported read-only through tools/make_golden.py's shims).
    PYTHONDONTWRITEBYTECODE=1 python tools/collator_bench.py"""
import copy, os, sys, tempfile, time
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import make_golden as G          # installs the shims and imports the reference

torch.set_num_threads(1)
samples = [
    {"conversations": [{"role": "system", "content": "you are helpful"},
                       {"role": "user", "content": "<|attachment|> describe the image in detail"},
                       {"role": "assistant", "content": "a cat sitting on grass"}],
     "modalities": [{"type": "image", "value": "cat.jpg"}]},
    {"conversations": [{"role": "user", "content": "first <|attachment|> and second <|attachment|> what is this"},
                       {"role": "assistant", "content": "two pictures"}],
     "modalities": [{"type": "image", "value": "cat.jpg"}, {"type": "image", "value": "EPFL_campus_2017.jpg"}]},
] * 2

with tempfile.TemporaryDirectory() as tmp:
    G.make_clip_dir(tmp, 5)
    # reference
    proc = G.im.ImageProcessor(G.ImageConfig(hidden_size=128, clip_name=tmp))
    tok = G.make_tokenizer()
    ref = G.DataCollatorForMultimodal(tokenizer=tok, modality_processors={"image": proc},
                                      modality_loaders={"image": G.FileSystemImageLoader("/root/reference/mock_dataset")},
                                      attachment_token="<|attachment|>", chat_template=G.llama_spaced_template())
    # this build
    from multimeditron_amd.model.data_loader import DataCollatorForMultimodal as Mine
    from multimeditron_amd.model.modalities import ImageConfig as MyImageConfig
    from multimeditron_amd.model.modalities.image_modality import ImageProcessor as MyProc
    from multimeditron_amd.model.model import ChatTemplate as MyCT
    from multimeditron_amd.dataset.loader import FileSystemImageLoader as MyLoader
    ct = MyCT.llama()
    for role in ct.delimiters:
        ct.delimiters[role] = {"start": f"<|start_header_id|> {role} <|end_header_id|>", "end": "<|eot_id|>"}
    mine = Mine(tokenizer=G.make_tokenizer(), modality_processors={"image": MyProc(MyImageConfig(hidden_size=128, clip_name=tmp))},
                modality_loaders={"image": MyLoader("/root/reference/mock_dataset")}, attachment_token="<|attachment|>", chat_template=ct)

    def bench(c, n=20):
        c(copy.deepcopy(samples))
        t = time.perf_counter()
        for _ in range(n):
            b = c(copy.deepcopy(samples))
        return (time.perf_counter() - t) / n, b

    tr, br = bench(ref)
    tm, bm = bench(mine)
    mine.num_threads = 8
    tp, bp = bench(mine)
    samep = all(torch.equal(bp[k], bm[k]) for k in ("input_ids", "labels", "attention_mask", "position_ids")) and all(
        torch.equal(a, b) for a, b in zip(bp["processed_multimodal_inputs"]["stacked"]["image"], bm["processed_multimodal_inputs"]["stacked"]["image"]))
    print(f"this build with num_threads=8: {tp * 1e3:.1f} ms ({tr / tp:.2f}x the reference); identical to sequential: {samep}")
    same = all(torch.equal(br[k], bm[k]) for k in ("input_ids", "labels", "attention_mask", "position_ids"))
    print(f"batch of {len(samples)} samples / 6 images (1 CPU thread): reference {tr * 1e3:.1f} ms, this build {tm * 1e3:.1f} ms "
          f"({tr / tm:.2f}x); outputs identical: {same}")
