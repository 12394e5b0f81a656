#!/usr/bin/env python3
"""Copy what tools/profile_round.sh (and tools/attn_pmc.sh) left under gpurun_out/ into profiles/ under a round prefix:
    python tools/publish_profile.py gpurun_out/prof_TAG r02 [gpurun_out/pmc_TAG]
-> profiles/r02_bench8b_kernel_stats.csv, _summary.md, _unprofiled_run.json, _profiled_run.json, r02_pmc_summary.md,
   r02_pmc_traffic.json (what bench.py reads for roofline.traffic), r02_attn_pmc.txt."""
import csv, json, os, shutil, sys

src, pre = sys.argv[1], sys.argv[2]
P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
shutil.copy(os.path.join(src, "kernel_stats.csv"), os.path.join(P, f"{pre}_bench8b_kernel_stats.csv"))
shutil.copy(os.path.join(src, "bench_unprofiled.json"), os.path.join(P, f"{pre}_bench8b_unprofiled_run.json"))
shutil.copy(os.path.join(src, "bench_profiled.json"), os.path.join(P, f"{pre}_bench8b_profiled_run.json"))
shutil.copy(os.path.join(src, "pmc_traffic.json"), os.path.join(P, f"{pre}_pmc_traffic.json"))
u = json.loads(open(os.path.join(src, "bench_unprofiled.json")).read())
pr = json.loads(open(os.path.join(src, "bench_profiled.json")).read())
rows = list(csv.DictReader(open(os.path.join(src, "kernel_stats.csv"))))
tot = sum(float(r["TotalDurationNs"]) for r in rows if "gemm_bf16_w4_kernel" in r["Name"])          # the roofline's kernel
n = sum(int(r["Calls"]) for r in rows if "gemm_bf16_w4_kernel" in r["Name"])
tot_all = sum(float(r["TotalDurationNs"]) for r in rows if "gemm_bf16_dma_kernel" in r["Name"] or "gemm_bf16_w4_kernel" in r["Name"])
n_all = sum(int(r["Calls"]) for r in rows if "gemm_bf16_dma_kernel" in r["Name"] or "gemm_bf16_w4_kernel" in r["Name"])
ru, rp = u["roofline"], pr["roofline"]
with open(os.path.join(P, f"{pre}_bench8b_summary.md"), "w") as f:
    f.write(f"# {pre}: rocprofv3 --kernel-trace --stats over `python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline` (tools/profile_round.sh)\n\n")
    f.write(f"Unprofiled run of the same build on the same box (`bench.py --steps {u['steps']} --warmup {u['warmup']}`): {u['ms_per_step']} ms/step = "
            f"{u['value']} samples/s; roofline of gemm_bf16_w4_kernel {ru['achieved']} TFLOP/s ({ru['frac']}); every bf16 GEMM launch "
            f"{ru['all_gemm_launches']['achieved']} TFLOP/s ({ru['all_gemm_launches']['frac']}); whole step {ru['whole_step_achieved']} TFLOP/s "
            f"({ru['whole_step_frac']}).\n")
    f.write(f"Profiled run: {pr['ms_per_step']} ms/step; average gemm_bf16_w4_kernel launch {rp['avg_launch_ms'] * 1e3:.1f} us from HIP events inside bench.py; the "
            f"same from the table below (all gemm_bf16_w4_kernel rows): {n} launches, {tot / 1e6:.1f} ms => {tot / n / 1e3:.1f} us.  Every bf16 GEMM launch "
            f"(+ gemm_bf16_dma_kernel rows): {rp['all_gemm_launches']['avg_launch_ms'] * 1e3:.1f} us from HIP events, {n_all} launches, {tot_all / 1e6:.1f} ms => "
            f"{tot_all / n_all / 1e3:.1f} us in the table.\n")
    f.write("Counts are over 5 steps (1 warm-up + 3 timed + the 1-step roofline pass); 'ms/step' = total / 5.  Kernels of different streams "
            "overlap (AdamW under the next forward, deferred wgrads beside the ViT backward), so the column sums to more than the step and "
            "the small ViT kernels show 5-10x their stand-alone duration (tools/rowwise_bench.py, tools/stream_time.py: "
            f"{pre}_other_workloads.md).\n\n```\n")
    f.write(open(os.path.join(src, "kernel_summary.txt")).read())
    f.write("```\n")
with open(os.path.join(P, f"{pre}_pmc_summary.md"), "w") as f:
    f.write(f"# {pre} PMC summary (separate rocprofv3 --pmc passes over `bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline`; "
            "tools/profile_round.sh, tools/pmc_summary.py)\n\n")
    f.write(open(os.path.join(src, "pmc_summary.md")).read())
    t = json.loads(open(os.path.join(src, "pmc_traffic.json")).read())
    f.write(f"\ngemm_bf16_w4_kernel HBM-side bytes per average launch (roofline.traffic): {t['bytes_per_launch'] / 1e9:.3f} GB "
            f"({t['read_bytes_per_launch'] / 1e9:.3f} read + {t['write_bytes_per_launch'] / 1e9:.3f} written) against "
            f"{ru['algorithmic_bytes_per_launch'] / 1e9:.3f} GB algorithmic; over every bf16 GEMM launch {t['all_gemm_launches']['bytes_per_launch'] / 1e9:.3f} GB; "
            f"kernel sources {t['kernel_source_sha']}.\n")
if len(sys.argv) > 3:
    with open(os.path.join(P, f"{pre}_attn_pmc.txt"), "w") as f:
        f.write(f"# {pre}: rocprofv3 --pmc passes over `python3 tools/attn_bench.py --quick` (tools/attn_pmc.sh): D=128 attention kernels, "
                "B=4 S=2048 32/8 heads causal, per launch.\n# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* are in quad-cycles summed over waves; "
                "SQ_VALU_MFMA_BUSY_CYCLES in cycles summed over SIMDs; GRBM_GUI_ACTIVE summed over the 8 XCDs.\n"
                "# MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024).\n")
        f.write(open(os.path.join(sys.argv[3], "summary.txt")).read())
print("published", pre)
