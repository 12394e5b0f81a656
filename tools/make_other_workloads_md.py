#!/usr/bin/env python3
"""(the GEMM evidence of round 4 has a file of its own, profiles/r04_gemm_w4.md)
Assemble profiles/rNN_other_workloads.md from what the round's gpurun calls left under gpurun_out/ (tools/other_workloads.sh, the
step A/B runs, the micro-benchmarks): every number in it is copied from a measured log, none typed in.
    python tools/make_other_workloads_md.py r03"""
import glob, json, os, sys
pre = sys.argv[1] if len(sys.argv) > 1 else "r03"
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(R, "gpurun_out")
out = [f"# {pre}: workloads other than the headline line, micro-benchmarks and the same-process A/B runs behind DESIGN.md section 6\n",
       "Every block is the text a tool printed on an MI355X box (one gpurun call each; boxes differ by up to 8 % for identical code, so only "
       "numbers inside ONE block compare).\n"]


def clean(txt):
    return "\n".join(l for l in txt.splitlines() if "amdgpu.ids" not in l and not l.startswith(("W2026", "E2026", "I2026")))


def block(title, path, tail=None, note=""):
    paths = sorted(glob.glob(os.path.join(G, path)))
    if not paths:
        return
    txt = clean(open(paths[-1]).read()).rstrip()
    if tail:
        txt = "\n".join(txt.splitlines()[-tail:])
    out.append(f"\n## {title}\n\n{note}\n```\n{txt}\n```\n" if note else f"\n## {title}\n\n```\n{txt}\n```\n")


rows = []
for f in sorted(glob.glob(os.path.join(G, f"other_{pre}", "bench_*.json"))):
    try:
        d = json.load(open(f))
    except Exception:
        continue
    r = d.get("roofline") or {}
    rows.append(f"| `{os.path.basename(f)[6:-5]}` | {d['config']['workload']} ({d['config']['training_mode']}) | {d['value']} | {d['ms_per_step']} | "
                f"{r.get('achieved')} ({r.get('frac')}) | {r.get('whole_step_achieved')} ({r.get('whole_step_frac')}) | {d['config'].get('final_loss')} |")
if rows:
    out.append("\n## bench.py on the other workloads (`tools/other_workloads.sh`; 6 steps after 2 warm-up, 1 GPU)\n\n"
               "| run | workload (mode) | samples/s | ms/step | GEMM TFLOP/s (frac) | whole step, executed flops (frac) | final loss |\n|---|---|---|---|---|---|---|\n"
               + "\n".join(rows) + "\n")
block("generate(): prefill + decode, B = 4, S = 2048, 8B (`tools/decode_bench.py`)", f"other_{pre}/decode.log", tail=3,
      note="Weights streamed per token: 16.06 GB (8.03 B bf16 parameters of the decoder + lm_head); KV cache read per token: 1.07 GB.")
block("the decode layer's kernels one by one (`tools/gemv_bench.py 4 16`: 16 layers' weights in rotation, event-timed back-to-back launches)",
      f"other_{pre}/gemv.txt")
block("D = 128 attention forward / backward (`tools/attn_bench.py --quick`)", f"other_{pre}/attn_quick.log")
block("row-wise kernels alone (`tools/rowwise_bench.py`)", f"other_{pre}/rowwise.log")
block("MoE image modality, 4 x ViT-L/14 experts on 4 images (`tools/moe_bench.py 4 4`)", f"other_{pre}/moe_bench.txt")
block("per-stream composition of one training step and the two overlap windows (`tools/stream_time.py`)", f"other_{pre}/stream_time.txt")
open(os.path.join(R, "profiles", f"{pre}_other_workloads.md"), "w").write("".join(out))
print("wrote", f"profiles/{pre}_other_workloads.md", sum(len(x) for x in out), "bytes")
