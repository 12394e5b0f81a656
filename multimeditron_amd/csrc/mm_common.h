// Shared device/host helpers for libmmhip (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mm_hip.h"

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define MM_WAVE 64
#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

#define MM_CHECK_LAUNCH()                          \
  do {                                             \
    hipError_t e_ = hipGetLastError();             \
    if (e_ != hipSuccess) return MM_ERR_LAUNCH;    \
  } while (0)

__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16 x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float x) { return (bf16)x; }

// 16-byte vector of T (8 bf16 or 4 f32)
template <typename T> struct Vec16;
template <> struct Vec16<bf16> {
  static constexpr int N = 8;
  bf16x8 v;
  __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
  __device__ __forceinline__ void set(int i, float x) { v[i] = (bf16)x; }
};
template <> struct Vec16<float> {
  static constexpr int N = 4;
  f32x4 v;
  __device__ __forceinline__ float get(int i) const { return v[i]; }
  __device__ __forceinline__ void set(int i, float x) { v[i] = x; }
};

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
  return x;
}
__device__ __forceinline__ float wave_max(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o, 64));
  return x;
}
// block-wide sum for blockDim.x == 256 (4 waves); red = 8 floats of LDS
__device__ __forceinline__ float block_sum_256(float x, float* red) {
  x = wave_sum(x);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = x;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_max_256(float x, float* red) {
  x = wave_max(x);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = x;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// RoPE of one (x[d], x[d + D/2]) pair: out_lo = x_lo cos - x_hi sin, out_hi = x_hi cos + x_lo sin (HF:llama apply_rotary_pos_emb with
// rotate_half), written with ONE explicit rounding sequence (a product, then a fused multiply-add) so that the stand-alone kernels
// (mm_rope_apply, mm_rope_append) and the GEMM epilogue that fuses the rotation (mm_gemm_rope_fwd) give the same bits.
__device__ __forceinline__ float rope_lo(float lo, float hi, float c, float s) { return __builtin_fmaf(lo, c, -__fmul_rn(hi, s)); }
__device__ __forceinline__ float rope_hi(float lo, float hi, float c, float s) { return __builtin_fmaf(hi, c, __fmul_rn(lo, s)); }

// ---- LDS-DMA (buffer_load_dwordx4 ... lds) issued from inline asm -----------------------------------------------
// hipcc's waitcnt pass does not see these, so it neither drains them in front of ds_read_b64_tr_b16 nor counts them:
// completion is tracked by hand (s_waitcnt vmcnt(N) + s_barrier before any wave reads the bytes).  The descriptor is
// forced into SGPRs (wave-uniform by construction); out-of-range bytes are written as zeros by the hardware range check.
struct SRsrc { unsigned w0, w1, w2, w3; };
__device__ __forceinline__ SRsrc make_srsrc(const void* base, int64_t bytes) {
  if (bytes < 0) bytes = 0;
  const unsigned nb = bytes > 0xFFFFFFFFll ? 0xFFFFFFFFu : (unsigned)bytes;
  const uint64_t a = (uint64_t)base;
  SRsrc r;
  r.w0 = __builtin_amdgcn_readfirstlane((unsigned)a);
  r.w1 = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xFFFFu);
  r.w2 = __builtin_amdgcn_readfirstlane(nb);
  r.w3 = 0x00020000u;
  return r;
}
// one 1-KiB piece: lane l's 16 bytes land at LDS byte address lds_addr + 16*l (lds_addr wave-uniform)
__device__ __forceinline__ void lds_dma16(const SRsrc& r, unsigned voff, unsigned lds_addr) {
  u32x4 d = {r.w0, r.w1, r.w2, r.w3};
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 4\n\t"
      "buffer_load_dwordx4 %1, %3, 0 offen lds\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(lds_addr), "s"(d)
      : "memory");
}

static inline bool mm_aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
static inline int mm_elem_size(int dt) { return dt == MM_BF16 ? 2 : 4; }
