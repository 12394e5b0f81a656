#!/usr/bin/env python3
"""Same-box, same-process A/B of training-step variants selected by environment switches (box-to-box variance is +-3-5 %,
run-to-run on one box ~1 %: decisions need interleaved rounds in ONE process, guide rule 24).
    python tools/step_ab.py [--rounds 3] [--steps 8] "MM_FUSED_NORM=1" "MM_FUSED_NORM=0" ["A=1,B=2" ...]
The 8B model is built once; every variant gets a fresh MultimodalTrainer (the switches are read at construction / call time)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--steps", type=int, default=8)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("variants", nargs="+")
args = ap.parse_args()

from multimeditron_amd.model.model import MultimodalConfig, MultiModalModelForCausalLM
from multimeditron_amd.model.modalities import ImageConfig
from multimeditron_amd.model.presets import resolve_llm_config, resolve_vision_config
from multimeditron_amd.train.trainer import MultimodalTrainer, TrainingMode
from multimeditron_amd.train.prefetch import DevicePrefetcher

llm_name, clip_name, B, S, n_img = bench.WORKLOADS["llama31_8b_vitl14_s2048_b4"]
llm, vis = resolve_llm_config(llm_name), resolve_vision_config(clip_name)
vocab = llm["vocab_size"] + 2
torch.manual_seed(1234)
dev = torch.device("cuda", 0)
cfg = MultimodalConfig(vocab_size=vocab, modalities=[ImageConfig(hidden_size=llm["hidden_size"], clip_name=clip_name)], llm_path=llm_name,
                       dtype="bfloat16", eos_token_idx=128009, hidden_size=llm["hidden_size"])
model = MultiModalModelForCausalLM(cfg, device=dev)
model.pack_parameters()
host_batch, _ = bench.synthetic_batch(B, S, n_img, 256, vocab, (llm["vocab_size"], llm["vocab_size"] + 1, 128002), 1234, "cpu", 224,
                                      collator_form=True)


def endless():
    while True:
        yield host_batch


OPTION_DEFAULTS = {"adamw_blocks": 0, "attn_q_rd": 4, "attn_q_prio": 1, "attn_diag": 0, "gemm_kernel": 0, "gemm_small": -1,
                   "gemm_issue_waves": 4, "attn_issue_waves": 4, "attn_fwd_waves": 8}      # everything else is an on/off switch, default 1
res = {v: [] for v in args.variants}
for r in range(args.rounds):
    for v in args.variants:
        saved, opts = {}, []
        for kv in v.split(","):
            k, val = kv.split("=")
            if k.startswith("opt:"):            # a libmmhip switch (mm_set_option), e.g. opt:attn_fwd_pf=0
                from multimeditron_amd._lib import lib
                assert lib().mm_set_option(k[4:].encode(), int(val)) == 0, k
                opts.append(k[4:])
                continue
            saved[k] = os.environ.get(k)
            os.environ[k] = val
        tr = MultimodalTrainer(model, training_mode=TrainingMode.FULL, learning_rate=1e-4, weight_decay=0.01, max_grad_norm=1.0,
                               max_steps=1000, min_lr=3e-5)
        feed = DevicePrefetcher(endless(), device=dev)
        for _ in range(args.warmup):
            tr.training_step(next(feed))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            tr.training_step(next(feed))
        tr.synchronize()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        res[v].append(ms)
        print(f"round {r} {v}: {ms:.2f} ms/step", flush=True)
        tr.close()
        del tr, feed
        import gc
        gc.collect()
        for k in opts:                          # back to the library's default
            lib().mm_set_option(k.encode(), OPTION_DEFAULTS.get(k, 1))
        for k, val in saved.items():
            if val is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = val
        torch.cuda.empty_cache()
for v, xs in res.items():
    xs = sorted(xs)
    print(f"{v}: median {xs[len(xs) // 2]:.2f}  min {xs[0]:.2f}  all {[round(x, 1) for x in xs]}")
