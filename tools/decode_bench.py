#!/usr/bin/env python3
"""generate() timing on the 8B workload: prefill and per-token decode (run on the GPU box).
   python tools/decode_bench.py [B] [S] [new_tokens]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from multimeditron_amd.model.model import MultimodalConfig, MultiModalModelForCausalLM
from multimeditron_amd.model.modalities import ImageConfig
from multimeditron_amd.model.presets import resolve_llm_config, resolve_vision_config

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
S = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
N = int(sys.argv[3]) if len(sys.argv) > 3 else 33
llm_name, clip_name = "meta-llama/Llama-3.1-8B-Instruct", "openai/clip-vit-large-patch14"
llm, vis = resolve_llm_config(llm_name), resolve_vision_config(clip_name)
vocab = llm["vocab_size"] + 2
torch.manual_seed(0)
cfg = MultimodalConfig(vocab_size=vocab, modalities=[ImageConfig(hidden_size=llm["hidden_size"], clip_name=clip_name)],
                       llm_path=llm_name, dtype="bfloat16", eos_token_idx=vocab + 5, hidden_size=llm["hidden_size"])   # eos never hit
model = MultiModalModelForCausalLM(cfg, device="cuda")
model.pack_parameters()
batch, _ = bench.synthetic_batch(B, S, 1, 256, vocab, (llm["vocab_size"], llm["vocab_size"] + 1, 128002), 7, "cuda", 224)
batch["attention_mask"] = torch.ones(B, S, dtype=torch.long, device="cuda")


def run(n):
    torch.cuda.synchronize()
    t = time.perf_counter()
    ids = model.generate(batch, max_new_tokens=n, temperature=0.1, do_sample=False)
    torch.cuda.synchronize()
    return time.perf_counter() - t, ids


run(2)
t1, _ = run(1)
tn, ids = run(N)
per_tok = (tn - t1) / (N - 1)
wbytes = sum(p.numel() for p in model.model.parameters()) * 2
print(f"B={B} S={S}: prefill+1 token {t1 * 1e3:.1f} ms; decode {per_tok * 1e3:.2f} ms/token ({B / per_tok:.0f} tok/s); "
      f"weight stream {wbytes / per_tok / 1e12:.2f} TB/s of 8 (floor {wbytes / 8e12 * 1e3:.2f} ms/token)  ids {tuple(ids.shape)}")
