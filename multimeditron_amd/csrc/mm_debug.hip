// Diagnostics: expose raw MFMA / transposed-LDS-read lane maps so the host-side tests can pin the fragment
// layouts the GEMM and attention kernels rely on (guide: "check the map with exact integer data").
#include "mm_common.h"

namespace {
__global__ void dbg_tr_read_kernel(const bf16* img, const int* byte_addr, bf16* out) {
  __shared__ __attribute__((aligned(16))) bf16 lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = img[i];
  __syncthreads();
  bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, (char*)lds + byte_addr[threadIdx.x]));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
__global__ void dbg_mfma32_kernel(const bf16* a, const bf16* b, float* out) {
  bf16x8 fa = *(const bf16x8*)(a + threadIdx.x * 8), fb = *(const bf16x8*)(b + threadIdx.x * 8);
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) out[threadIdx.x * 16 + r] = acc[r];
}
__global__ void dbg_mfma16_kernel(const bf16* a, const bf16* b, float* out) {
  bf16x8 fa = *(const bf16x8*)(a + threadIdx.x * 8), fb = *(const bf16x8*)(b + threadIdx.x * 8);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[threadIdx.x * 4 + r] = acc[r];
}
}  // namespace

extern "C" int mm_debug_tr_read(const void* img_bf16_4096, const void* lane_byte_addr_i32_64, void* out_bf16_256, void* stream) {
  if (!img_bf16_4096 || !lane_byte_addr_i32_64 || !out_bf16_256) return MM_ERR_ARG;
  hipLaunchKernelGGL(dbg_tr_read_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const bf16*)img_bf16_4096, (const int*)lane_byte_addr_i32_64, (bf16*)out_bf16_256);
  MM_CHECK_LAUNCH();
  return MM_OK;
}
extern "C" int mm_debug_mfma(int shape, const void* a_frag_bf16_512, const void* b_frag_bf16_512, void* out_f32, void* stream) {
  if (!a_frag_bf16_512 || !b_frag_bf16_512 || !out_f32) return MM_ERR_ARG;
  if (shape == 32)
    hipLaunchKernelGGL(dbg_mfma32_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const bf16*)a_frag_bf16_512, (const bf16*)b_frag_bf16_512, (float*)out_f32);
  else if (shape == 16)
    hipLaunchKernelGGL(dbg_mfma16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const bf16*)a_frag_bf16_512, (const bf16*)b_frag_bf16_512, (float*)out_f32);
  else
    return MM_ERR_ARG;
  MM_CHECK_LAUNCH();
  return MM_OK;
}

// ---- streams restricted to a subset of the CUs (hipExtStreamCreateWithCUMask) ----------------------------------------
// The training step runs two HBM- / MFMA-saturating bursts BESIDE a latency-bound chain of small kernels (AdamW beside the next
// ViT forward; the deferred weight-gradient GEMMs beside the ViT backward).  A burst that may use every CU leaves the chain's
// workgroups waiting for a CU to drain (a 256x256 GEMM tile holds its CU for ~220 us); on a stream whose CU mask leaves a few CUs
// per XCD out, those CUs are always free for the chain.  mm_debug_cu_probe reports where workgroups of a stream really ran.
namespace {
__global__ void cu_probe_kernel(unsigned* out, long long spin) {
  const long long t0 = __builtin_amdgcn_s_memrealtime();
  while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < spin) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));       // HW_REG_XCC_ID
    out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));    // HW_REG_HW_ID: cu 11:8, sh 12, se 15:13
  }
}
}  // namespace

extern "C" int mm_stream_create_cu_mask(const unsigned* mask, int nwords, void** stream) {
  if (!mask || nwords <= 0 || !stream) return MM_ERR_ARG;
  hipStream_t s = nullptr;
  if (hipExtStreamCreateWithCUMask(&s, (uint32_t)nwords, mask) != hipSuccess) { (void)hipGetLastError(); return MM_ERR_UNSUPPORTED; }
  *stream = (void*)s;
  return MM_OK;
}
// a stream of the given HIP priority (hipDeviceGetStreamPriorityRange: numerically LOWER = served first; `least` is the lowest
// priority the device offers): the Trainer can put a saturating side burst BELOW the default stream's small kernels
extern "C" int mm_stream_priority_range(int* least, int* greatest) {
  if (!least || !greatest) return MM_ERR_ARG;
  return hipDeviceGetStreamPriorityRange(least, greatest) == hipSuccess ? MM_OK : MM_ERR_UNSUPPORTED;
}
extern "C" int mm_stream_create_priority(int priority, void** stream) {
  if (!stream) return MM_ERR_ARG;
  hipStream_t s = nullptr;
  if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, priority) != hipSuccess) { (void)hipGetLastError(); return MM_ERR_UNSUPPORTED; }
  *stream = (void*)s;
  return MM_OK;
}
extern "C" int mm_stream_destroy(void* stream) {
  if (!stream) return MM_ERR_ARG;
  return hipStreamDestroy((hipStream_t)stream) == hipSuccess ? MM_OK : MM_ERR_ARG;
}
extern "C" int mm_device_cu_count(void) {
  int d = 0;
  hipDeviceProp_t p;
  if (hipGetDevice(&d) != hipSuccess || hipGetDeviceProperties(&p, d) != hipSuccess) return -1;
  return p.multiProcessorCount;
}
// n_wg workgroups of `threads` threads, each holding its slot for `spin_ticks` of the 100 MHz counter; out[2 i] = XCC id register,
// out[2 i + 1] = HW_ID register of workgroup i's first wave
extern "C" int mm_debug_cu_probe(void* out_u32, int n_wg, int threads, int64_t spin_ticks, void* stream) {
  if (!out_u32 || n_wg <= 0 || threads <= 0 || threads > 1024 || spin_ticks < 0 || spin_ticks > 100000000ll) return MM_ERR_ARG;
  hipLaunchKernelGGL(cu_probe_kernel, dim3(n_wg), dim3(threads), 0, (hipStream_t)stream, (unsigned*)out_u32, spin_ticks);
  MM_CHECK_LAUNCH();
  return MM_OK;
}
