"""Device image preprocessing (mm_image_resample_h / mm_image_resample_v_norm through GpuClipPreprocessor) against the CPU
preprocessor (ClipImagePreprocessor = the reference's image processor, pinned by tests/test_collator_golden.py): bit-identical
fp32 pixels on the reference's test images and on random images, for the CLIP recipe (shortest edge + center crop) and the SigLIP
recipe (plain resize, no crop)."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu


def _cfgs():
    from multimeditron_amd.model.presets import resolve_preprocessor_config
    return {"clip224": resolve_preprocessor_config("openai/clip-vit-large-patch14", 224),
            "siglip384": resolve_preprocessor_config("google/siglip-so400m-patch14-384", 384)}


@pytest.mark.parametrize("recipe", ["clip224", "siglip384"])
def test_gpu_preprocessing_is_bit_identical(golden_dir, recipe):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from multimeditron_amd.dataset.gpu_image import GpuClipPreprocessor
    from multimeditron_amd.model.modalities.image_modality import ClipImagePreprocessor
    cfg = _cfgs()[recipe]
    cpu, gpu = ClipImagePreprocessor(cfg), GpuClipPreprocessor(cfg)
    rng = np.random.default_rng(5)
    images = [Image.open(os.path.join(golden_dir, "mock_dataset", n)) for n in ("cat.jpg", "EPFL_campus_2017.jpg")]
    for h, w in ((480, 640), (641, 479), (224, 224), (1000, 333), (60, 50), (300, 900), (384, 384)):
        images.append(Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)))
    images.append(Image.fromarray(rng.integers(0, 256, (200, 300), dtype=np.uint8)))          # greyscale: RGB conversion on the host
    ref = torch.stack([cpu(im) for im in images])
    got = gpu(images)
    torch.cuda.synchronize()
    assert got.shape == ref.shape and got.dtype == torch.float32
    assert torch.equal(got.cpu(), ref), float((got.cpu() - ref).abs().max())


def test_collator_with_gpu_preprocess_delivers_the_reference_pixels(golden_dir, tmp_path):
    """End to end: the collator with `ImageConfig(gpu_preprocess=True)` carries decoded uint8 images, DevicePrefetcher turns them
    into the pixel stack on its stream; ids / labels / splice indices equal the reference collator's fixture and the pixels equal
    the CPU path's bit for bit (the fixture's within its own 1e-6)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import copy
    import json
    from safetensors.torch import load_file
    from tests.test_collator_golden import _tokenizer_factory, _spaced_llama_template
    from multimeditron_amd.dataset.gpu_image import GpuClipPreprocessor
    from multimeditron_amd.dataset.loader import AutoModalityLoader
    from multimeditron_amd.model.data_loader import DataCollatorForMultimodal
    from multimeditron_amd.model.modalities import AutoModality, ImageConfig
    from multimeditron_amd.train.prefetch import DevicePrefetcher
    meta = json.load(open(os.path.join(golden_dir, "collator.meta.json")))
    vec = load_file(os.path.join(golden_dir, "collator.vectors.safetensors"))
    d = tmp_path / "clip"
    os.makedirs(d)
    json.dump({"vision_config": {"hidden_size": 128, "intermediate_size": 256, "num_hidden_layers": 2, "num_attention_heads": 2,
                                 "image_size": meta["image_size"], "patch_size": meta["patch_size"]}}, open(d / "config.json", "w"))
    json.dump({"size": {"shortest_edge": meta["image_size"]}, "crop_size": {"height": meta["image_size"], "width": meta["image_size"]}},
              open(d / "preprocessor_config.json", "w"))
    imgdir = os.path.join(golden_dir, "mock_dataset")
    batches = {}
    for gpu in (False, True):
        proc = AutoModality.preprocessor_from_name("meditron_clip", ImageConfig(hidden_size=128, clip_name=str(d), gpu_preprocess=gpu))
        coll = DataCollatorForMultimodal(tokenizer=_tokenizer_factory(meta)("right"), modality_processors={"image": proc},
                                         modality_loaders={"image": AutoModalityLoader.from_name("fs-image", base_path=imgdir)},
                                         attachment_token=meta["attachment_token"], chat_template=_spaced_llama_template(),
                                         add_generation_prompt=False)
        b = coll(copy.deepcopy(meta["samples_conv"]))
        pps = {"image": GpuClipPreprocessor(proc.preprocessor_config, device="cuda")} if gpu else None
        batches[gpu] = next(iter(DevicePrefetcher(iter([b]), device="cuda", image_preprocessors=pps)))
    torch.cuda.synchronize()
    cpu_b, gpu_b = batches[False], batches[True]
    for k in ("input_ids", "labels", "attention_mask", "position_ids"):
        assert torch.equal(gpu_b[k].cpu(), vec[f"conv_right_gen0.{k}"]), k
    px_cpu = cpu_b["processed_multimodal_inputs"]["stacked"]["image"]
    px_gpu = gpu_b["processed_multimodal_inputs"]["stacked"]["image"]
    assert px_gpu.is_cuda and px_gpu.shape == px_cpu.shape
    assert torch.equal(px_gpu, px_cpu)
    assert float((px_gpu.cpu() - vec["conv_right_gen0.pixels"]).abs().max()) < 1e-6
