#!/usr/bin/env python3
"""Writes build_diag/epi_probe.hip: what the pieces of a GEMM epilogue cost a wave that is alone on its SIMD (a probe, not the product).
   python tools/probes/gen_epi_probe.py && hipcc --offload-arch=gfx950 -O3 -o build_diag/epi_probe build_diag/epi_probe.hip
256 workgroups x 4 waves; every variant runs its instruction sequence ITERS times between two s_memtime stamps (shader-clock cycles)."""
import os

def seq(var):
    L = []
    if var == "accread":            # 256 v_accvgpr_read
        for i in range(256): L.append(f"v_accvgpr_read_b32 v{100 + i % 64}, a{i}")
    elif var == "mov":              # 256 v_mov (the same as plain VALU)
        for i in range(256): L.append(f"v_mov_b32 v{100 + i % 64}, v{170 + i % 64}")
    elif var == "cvt":              # 128 v_cvt_pk_bf16_f32
        for i in range(128): L.append(f"v_cvt_pk_bf16_f32 v{100 + i % 64}, v{170 + i % 32}, v{202 + i % 32}")
    elif var == "swap":             # 64 v_permlane16_swap
        for i in range(64): L.append(f"v_permlane16_swap_b32 v{100 + 2 * (i % 32)}, v{101 + 2 * (i % 32)}")
    elif var == "store":            # 32 buffer_store_dwordx4, 16 rows x 64 B each (the shuffle epilogue's pattern), then wait
        for i in range(32): L.append(f"buffer_store_dwordx4 v[100:103], %[vo], %[desc], 0 offen offset:{(i % 4) * 64 + (i // 4) * 512}")
        L.append("s_waitcnt vmcnt(0)")
    elif var == "store_nowait":
        for i in range(32): L.append(f"buffer_store_dwordx4 v[100:103], %[vo], %[desc], 0 offen offset:{(i % 4) * 64 + (i // 4) * 512}")
    elif var == "store_rows":       # 32 stores of 8 rows x 128 B (row-major epilogue's pattern)
        for i in range(32): L.append(f"buffer_store_dwordx4 v[100:103], %[vo2], %[desc], 0 offen offset:{(i % 2) * 128 + (i // 2) * 256}")
    elif var == "store_rows_small":  # the same 32 stores over a quarter of the footprint (8 KB per wave): L2-resident lines
        for i in range(32): L.append(f"buffer_store_dwordx4 v[100:103], %[vo2], %[desc], 0 offen offset:{(i % 2) * 128 + ((i // 2) % 4) * 256}")
    elif var == "store_rows_nt":
        for i in range(32): L.append(f"buffer_store_dwordx4 v[100:103], %[vo2], %[desc], 0 offen offset:{(i % 2) * 128 + (i // 2) * 256} nt")
    elif var == "store_rows_sc":
        for i in range(32): L.append(f"buffer_store_dwordx4 v[100:103], %[vo2], %[desc], 0 offen offset:{(i % 2) * 128 + (i // 2) * 256} sc0 sc1")
    elif var == "lds_w":            # 64 ds_write_b128 from AGPRs
        for i in range(64): L.append(f"ds_write_b128 %[la], a[{4 * i}:{4 * i + 3}] offset:{(i % 8) * 1024}")
        L.append("s_waitcnt lgkmcnt(0)")
    elif var == "lds_r":
        for i in range(64): L.append(f"ds_read_b128 v[{100 + 4 * (i % 16)}:{103 + 4 * (i % 16)}], %[la] offset:{(i % 8) * 1024}")
        L.append("s_waitcnt lgkmcnt(0)")
    elif var == "shuffle_all":      # the shuffle epilogue of one tile: per pair of tiles 8 reads, 4 cvt, 2 swaps, 1 store
        for p in range(32):
            for r in range(8): L.append(f"v_accvgpr_read_b32 v{100 + 8 * (p % 4) + r}, a{8 * p + r}")
            for r in range(4): L.append(f"v_cvt_pk_bf16_f32 v{140 + 4 * (p % 4) + r}, v{100 + 8 * (p % 4) + 2 * r}, v{101 + 8 * (p % 4) + 2 * r}")
            L.append(f"v_permlane16_swap_b32 v{140 + 4 * (p % 4)}, v{142 + 4 * (p % 4)}")
            L.append(f"v_permlane16_swap_b32 v{141 + 4 * (p % 4)}, v{143 + 4 * (p % 4)}")
            L.append(f"buffer_store_dwordx4 v[{140 + 4 * (p % 4)}:{143 + 4 * (p % 4)}], %[vo], %[desc], 0 offen offset:{(p % 4) * 64 + (p // 4) * 512}")
    elif var == "shuffle_fresh":    # shuffle_all with the stores going to lines nobody has touched for 27 passes (soffset moves 32 MB per pass)
        for p in range(32):
            for r in range(8): L.append(f"v_accvgpr_read_b32 v{100 + 8 * (p % 4) + r}, a{8 * p + r}")
            for r in range(4): L.append(f"v_cvt_pk_bf16_f32 v{140 + 4 * (p % 4) + r}, v{100 + 8 * (p % 4) + 2 * r}, v{101 + 8 * (p % 4) + 2 * r}")
            L.append(f"v_permlane16_swap_b32 v{140 + 4 * (p % 4)}, v{142 + 4 * (p % 4)}")
            L.append(f"v_permlane16_swap_b32 v{141 + 4 * (p % 4)}, v{143 + 4 * (p % 4)}")
            L.append(f"buffer_store_dwordx4 v[{140 + 4 * (p % 4)}:{143 + 4 * (p % 4)}], %[vo2], %[desc], %[so] offen offset:{(p % 2) * 128 + (p // 2) * 256}")
    elif var == "shuffle_nostore":
        for p in range(32):
            for r in range(8): L.append(f"v_accvgpr_read_b32 v{100 + 8 * (p % 4) + r}, a{8 * p + r}")
            for r in range(4): L.append(f"v_cvt_pk_bf16_f32 v{140 + 4 * (p % 4) + r}, v{100 + 8 * (p % 4) + 2 * r}, v{101 + 8 * (p % 4) + 2 * r}")
            L.append(f"v_permlane16_swap_b32 v{140 + 4 * (p % 4)}, v{142 + 4 * (p % 4)}")
            L.append(f"v_permlane16_swap_b32 v{141 + 4 * (p % 4)}, v{143 + 4 * (p % 4)}")
    elif var == "mfma_then_read":   # 64 MFMAs then 256 accumulator reads: what the hand-over costs
        for i in range(64): L.append(f"v_mfma_f32_16x16x32_bf16 a[{4 * i}:{4 * i + 3}], v[100:103], v[104:107], a[{4 * i}:{4 * i + 3}]")
        for i in range(256): L.append(f"v_accvgpr_read_b32 v{110 + i % 64}, a{i}")
    elif var == "mfma":
        for i in range(64): L.append(f"v_mfma_f32_16x16x32_bf16 a[{4 * i}:{4 * i + 3}], v[100:103], v[104:107], a[{4 * i}:{4 * i + 3}]")
    return L

VARS = ["mov", "accread", "cvt", "swap", "store", "store_nowait", "store_rows", "store_rows_small", "store_rows_nt", "store_rows_sc", "lds_w", "lds_r", "shuffle_nostore", "shuffle_all", "mfma", "mfma_then_read"]

def main():
    os.makedirs("build_diag", exist_ok=True)
    out = ['''// GENERATED by tools/probes/gen_epi_probe.py -- a probe, not part of the product
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) int i32x4;
''']
    out.append("#define VCLOB " + ",".join(f'"v{i}"' for i in range(100, 240)))
    out.append("#define ACLOB " + ",".join(f'"a{i}"' for i in range(256)))
    for var in VARS:
        body = "\\n".join(seq(var))
        out.append(f'''
__global__ __launch_bounds__(256) void probe_{var}(char* win, int iters, long long* cyc) {{
  extern __shared__ uint4 lds[];
  const int t = threadIdx.x, l = t & 63, w = t >> 6;
  unsigned la = (unsigned)(w * 8192 + l * 16);
  i32x4 desc;
  {{ unsigned long long p = (unsigned long long)(win + (size_t)blockIdx.x * (1 << 20) + w * (1 << 18)); desc[0] = (int)p; desc[1] = (int)(p >> 32); desc[2] = 1 << 18; desc[3] = 0x00020000; }}
  desc[0] = __builtin_amdgcn_readfirstlane(desc[0]); desc[1] = __builtin_amdgcn_readfirstlane(desc[1]);
  desc[2] = __builtin_amdgcn_readfirstlane(desc[2]); desc[3] = __builtin_amdgcn_readfirstlane(desc[3]);
  unsigned vo = (unsigned)((l & 15) * 8192 + (l >> 4) * 16);        // 16 rows x 4 x 16 B
  unsigned vo2 = (unsigned)((l >> 3) * 8192 + (l & 7) * 16);        // 8 rows x 128 B
  lds[t] = uint4{{0, 0, 0, 0}};
  __syncthreads();
  long long t0, t1;
  asm volatile("s_memtime %[t0]\\ns_waitcnt lgkmcnt(0)\\n1:\\n{body}\\ns_sub_u32 %[it], %[it], 1\\ns_cmp_lg_u32 %[it], 0\\ns_cbranch_scc1 1b\\ns_memtime %[t1]\\ns_waitcnt vmcnt(0) lgkmcnt(0)\\n"
               : [t0] "=&s"(t0), [t1] "=&s"(t1), [it] "+s"(iters) : [la] "v"(la), [desc] "s"(desc), [vo] "v"(vo), [vo2] "v"(vo2) : "memory", "scc", VCLOB, ACLOB);
  if (l == 0) cyc[blockIdx.x * 4 + w] = t1 - t0;
}}''')
    for var in ["shuffle_nostore", "shuffle_all", "accread", "shuffle_fresh"]:
        body = "\\n".join(seq(var))
        inner = "\\n".join(seq("mfma"))
        out.append(f'''
__global__ __launch_bounds__(256) void probe_rare_{var}(char* win, int iters, long long* cyc) {{
  extern __shared__ uint4 lds[];
  const int t = threadIdx.x, l = t & 63, w = t >> 6;
  unsigned la = (unsigned)(w * 8192 + l * 16);
  i32x4 desc;
  {{ unsigned long long p = (unsigned long long)(win + (size_t)blockIdx.x * (1 << 17) + w * (1 << 15)); desc[0] = (int)p; desc[1] = (int)(p >> 32); desc[2] = 0x40000000; desc[3] = 0x00020000; }}
  desc[0] = __builtin_amdgcn_readfirstlane(desc[0]); desc[1] = __builtin_amdgcn_readfirstlane(desc[1]);
  desc[2] = __builtin_amdgcn_readfirstlane(desc[2]); desc[3] = __builtin_amdgcn_readfirstlane(desc[3]);
  unsigned vo = (unsigned)((l & 15) * 2048 + (l >> 4) * 16);        // (rare_* kernels: a wave's 32 KB = 16 rows x 2 KB, dense)
  unsigned vo2 = (unsigned)((l >> 3) * 4096 + (l & 7) * 16);        // 8 rows x 4 KB
  lds[t] = uint4{{0, 0, 0, 0}};
  __syncthreads();
  long long t0, t1, acc = 0;
  for (int it = 0; it < iters; ++it) {{
    int inner = 40;
    const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)(it % 28) << 25);      // + 32 MB per pass
    asm volatile("1:\\n{inner}\\ns_sub_u32 %[it], %[it], 1\\ns_cmp_lg_u32 %[it], 0\\ns_cbranch_scc1 1b\\ns_memtime %[t0]\\ns_waitcnt lgkmcnt(0)\\n{body}\\ns_memtime %[t1]\\ns_waitcnt lgkmcnt(0)\\n"
                 : [t0] "=&s"(t0), [t1] "=&s"(t1), [it] "+s"(inner) : [la] "v"(la), [desc] "s"(desc), [vo] "v"(vo), [vo2] "v"(vo2), [so] "s"(so) : "memory", "scc", VCLOB, ACLOB);
    acc += t1 - t0;
  }}
  if (l == 0) cyc[blockIdx.x * 4 + w] = acc;
}}''')
    out.append('''
typedef void (*kern_t)(char*, int, long long*);
static void run(const char* name, kern_t k, char* win, long long* cyc, int n_instr, int grid = 256) {
  const int iters = 200;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 64 * 1024, 0, win, iters, cyc);
  hipDeviceSynchronize();
  static long long h[1024];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0; long long mx = 0;
  const int nw = grid * 4;
  for (int i = 0; i < nw; ++i) { s += h[i]; if (h[i] > mx) mx = h[i]; }
  printf("%-18s grid %3d %4d instructions: %8.0f cycles per pass (mean over waves; max %8.0f) = %5.1f cycles per instruction\\n", name, grid, n_instr, s / nw / iters, (double)mx / iters, s / nw / iters / n_instr);
  if (hipGetLastError() != hipSuccess) { printf("HIP error\\n"); exit(1); }
}
int main() {
  char* win; long long* cyc;
  hipMalloc(&win, 1100ll << 20); hipMalloc(&cyc, 1024 * 8);
''')
    for var in VARS:
        out.append(f'  run("{var}", probe_{var}, win, cyc, {len(seq(var))});')
    for grid in (128, 64, 32, 8):
        out.append(f'  run("store_nowait", probe_store_nowait, win, cyc, 32, {grid});')
        out.append(f'  run("store_rows", probe_store_rows, win, cyc, 32, {grid});')
    for var in ["shuffle_nostore", "shuffle_all", "accread", "shuffle_fresh"]:
        out.append(f'  run("rare_{var}", probe_rare_{var}, win, cyc, {len(seq(var))});')
    out.append("  return 0;\n}\n")
    with open("build_diag/epi_probe.hip", "w") as f:
        f.write("\n".join(out))

main()
