// Fused AdamW over flat parameter storage + global grad-norm (clip) -- replaces the reference's
// HF-Trainer/DeepSpeed CPU-offloaded Adam (config_alignment.yaml:38-59, deepspeed.json:5-23) with one
// HBM-bound pass: 16 B/param read (p-master, m, v f32 + g) and 14 B/param written.
#include <stdlib.h>

#include "mm_common.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void sumsq_kernel(const T* g, int64_t n, float* partial) {
  __shared__ float red[8];
  constexpr int VN = Vec16<T>::N;
  float s = 0.f;
  const int64_t nv = n / VN;
  const int64_t st = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * st < nv; i += 4 * st) {          // four independent 16-byte loads in flight per thread (one per iteration left the sweep at 5.1 TB/s)
    Vec16<T> v0 = *(const Vec16<T>*)(g + i * VN), v1 = *(const Vec16<T>*)(g + (i + st) * VN), v2 = *(const Vec16<T>*)(g + (i + 2 * st) * VN),
             v3 = *(const Vec16<T>*)(g + (i + 3 * st) * VN);
#pragma unroll
    for (int k = 0; k < VN; ++k) s += v0.get(k) * v0.get(k);
#pragma unroll
    for (int k = 0; k < VN; ++k) s += v1.get(k) * v1.get(k);
#pragma unroll
    for (int k = 0; k < VN; ++k) s += v2.get(k) * v2.get(k);
#pragma unroll
    for (int k = 0; k < VN; ++k) s += v3.get(k) * v3.get(k);
  }
  for (; i < nv; i += st) {
    Vec16<T> v = *(const Vec16<T>*)(g + i * VN);
#pragma unroll
    for (int k = 0; k < VN; ++k) s += v.get(k) * v.get(k);
  }
  if (blockIdx.x == 0)
    for (int64_t i = nv * VN + threadIdx.x; i < n; i += 256) s += to_f32(g[i]) * to_f32(g[i]);
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void gradnorm_finish_kernel(const float* partial, int nblk, float max_norm, float* total) {
  __shared__ float red[8];
  float s = 0.f;
  for (int i = threadIdx.x; i < nblk; i += 256) s += partial[i];
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) {
    const float nrm = sqrtf(s);
    total[0] = nrm;
    // torch.nn.utils.clip_grad_norm_: coef = max_norm / (norm + 1e-6), clamped to 1
    total[1] = max_norm > 0.f ? fminf(1.0f, max_norm / (nrm + 1e-6f)) : 1.0f;
  }
}

// 4 parameters per thread (16-byte f32 vectors), non-temporal loads/stores: the update streams 30 B/param once and
// must not evict the GEMM operand panels from L2 while it runs under the next step's forward on a side stream.
template <typename T, bool NT>
__global__ __launch_bounds__(256) void adamw_kernel(T* p, const T* g, float* master, float* m, float* v, int64_t n, float lr, float b1,
                                                    float b2, float eps, float wd, float bc1, float bc2, const float* clip) {
  const float c = clip ? clip[1] : 1.0f;
  const int64_t nv = n / 4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
    const f32x4 w4 = NT ? __builtin_nontemporal_load((const f32x4*)master + i) : ((const f32x4*)master)[i];
    const f32x4 m4 = NT ? __builtin_nontemporal_load((const f32x4*)m + i) : ((const f32x4*)m)[i];
    const f32x4 v4 = NT ? __builtin_nontemporal_load((const f32x4*)v + i) : ((const f32x4*)v)[i];
    float gi[4];
    if constexpr (sizeof(T) == 2) {
      const bf16x4 g4 = NT ? __builtin_nontemporal_load((const bf16x4*)g + i) : ((const bf16x4*)g)[i];
#pragma unroll
      for (int k = 0; k < 4; ++k) gi[k] = (float)g4[k] * c;
    } else {
      const f32x4 g4 = NT ? __builtin_nontemporal_load((const f32x4*)g + i) : ((const f32x4*)g)[i];
#pragma unroll
      for (int k = 0; k < 4; ++k) gi[k] = g4[k] * c;
    }
    f32x4 wo, mo, vo;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float mi = b1 * m4[k] + (1.f - b1) * gi[k];
      const float vi = b2 * v4[k] + (1.f - b2) * gi[k] * gi[k];
      float w = w4[k] * (1.f - lr * wd);
      w = w - lr * (mi / bc1) / (sqrtf(vi / bc2) + eps);
      mo[k] = mi; vo[k] = vi; wo[k] = w;
    }
    if (NT) {
      __builtin_nontemporal_store(mo, (f32x4*)m + i);
      __builtin_nontemporal_store(vo, (f32x4*)v + i);
      __builtin_nontemporal_store(wo, (f32x4*)master + i);
    } else {
      ((f32x4*)m)[i] = mo;
      ((f32x4*)v)[i] = vo;
      ((f32x4*)master)[i] = wo;
    }
    if constexpr (sizeof(T) == 2) {
      bf16x4 po;
#pragma unroll
      for (int k = 0; k < 4; ++k) po[k] = (bf16)wo[k];
      *((bf16x4*)p + i) = po;          // the parameters ARE re-read soon (next forward): default cache policy
    } else {
      *((f32x4*)p + i) = wo;
    }
  }
  if (blockIdx.x == 0)
    for (int64_t i = nv * 4 + threadIdx.x; i < n; i += 256) {
      const float gi = to_f32(g[i]) * c;
      const float mi = b1 * m[i] + (1.f - b1) * gi;
      const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
      float w = master[i] * (1.f - lr * wd);
      w = w - lr * (mi / bc1) / (sqrtf(vi / bc2) + eps);
      m[i] = mi; v[i] = vi; master[i] = w; p[i] = from_f32<T>(w);
    }
}

// AdamW with the fp32 master weight kept as (bf16 parameter, 16-bit remainder) instead of a separate fp32 copy (round 3).
// The bf16 parameter IS the high half of the master up to its rounding: master_bits = (p_bits << 16) + (int16) lo, with
// p = RNE(master) and lo = master_bits - (p_bits << 16) in [-0x8000, 0x8000].  One value of that range does not fit 16 bits
// (+0x8000: an exact tie that round-to-even resolved downwards, probability 2^-17 per update); it is stored as 0x7FFF, i.e. the
// master moves by one fp32 ulp there.  Traffic per parameter: read g 2 + p 2 + lo 2 + m 4 + v 4, write p 2 + lo 2 + m 4 + v 4
// = 26 B against 28 B with a separate fp32 master (and 2 B per parameter less memory).  The update arithmetic is adamw_kernel's.
template <bool NT>
__global__ __launch_bounds__(256) void adamw_split_kernel(bf16* p, const bf16* g, short* lo, float* m, float* v, int64_t n, float lr, float b1,
                                                          float b2, float eps, float wd, float bc1, float bc2, const float* clip) {
  const float c = clip ? clip[1] : 1.0f;
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  auto step = [&](float w, float gi, float& mi, float& vi) {
    mi = b1 * mi + (1.f - b1) * gi;
    vi = b2 * vi + (1.f - b2) * gi * gi;
    w = w * (1.f - lr * wd);
    return w - lr * (mi / bc1) / (sqrtf(vi / bc2) + eps);
  };
  auto join = [](bf16 pb, short l) { return __builtin_bit_cast(float, ((unsigned)__builtin_bit_cast(unsigned short, pb) << 16) + (unsigned)(int)l); };
  auto split = [](float w, bf16& pb, short& l) {
    pb = (bf16)w;                                                                       // round to nearest even (v_cvt_pk_bf16_f32)
    int d = (int)(__builtin_bit_cast(unsigned, w) - ((unsigned)__builtin_bit_cast(unsigned short, pb) << 16));
    l = (short)(d > 32767 ? 32767 : d);
  };
  const int64_t nv = n / 4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
    const bf16x4 p4 = ((const bf16x4*)p)[i];
    const s16x4 l4 = NT ? __builtin_nontemporal_load((const s16x4*)lo + i) : ((const s16x4*)lo)[i];
    const f32x4 m4 = NT ? __builtin_nontemporal_load((const f32x4*)m + i) : ((const f32x4*)m)[i];
    const f32x4 v4 = NT ? __builtin_nontemporal_load((const f32x4*)v + i) : ((const f32x4*)v)[i];
    const bf16x4 g4 = NT ? __builtin_nontemporal_load((const bf16x4*)g + i) : ((const bf16x4*)g)[i];
    f32x4 mo, vo;
    bf16x4 po;
    s16x4 lo4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float mi = m4[k], vi = v4[k];
      const float w = step(join(p4[k], l4[k]), (float)g4[k] * c, mi, vi);
      bf16 pb;
      short l;
      split(w, pb, l);
      mo[k] = mi; vo[k] = vi; po[k] = pb; lo4[k] = l;
    }
    if (NT) {
      __builtin_nontemporal_store(mo, (f32x4*)m + i);
      __builtin_nontemporal_store(vo, (f32x4*)v + i);
      __builtin_nontemporal_store(lo4, (s16x4*)lo + i);
    } else {
      ((f32x4*)m)[i] = mo;
      ((f32x4*)v)[i] = vo;
      ((s16x4*)lo)[i] = lo4;
    }
    ((bf16x4*)p)[i] = po;              // the parameters ARE re-read soon (next forward): default cache policy
  }
  if (blockIdx.x == 0)
    for (int64_t i = nv * 4 + threadIdx.x; i < n; i += 256) {
      float mi = m[i], vi = v[i];
      const float w = step(join(p[i], lo[i]), (float)g[i] * c, mi, vi);
      bf16 pb;
      short l;
      split(w, pb, l);
      m[i] = mi; v[i] = vi; p[i] = pb; lo[i] = l;
    }
}

// (p, lo) <-> fp32 master: checkpoints keep the fp32 form
__global__ __launch_bounds__(256) void master_split_kernel(const float* master, int64_t n, bf16* p, short* lo) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float w = master[i];
    const bf16 pb = (bf16)w;
    const int d = (int)(__builtin_bit_cast(unsigned, w) - ((unsigned)__builtin_bit_cast(unsigned short, pb) << 16));
    p[i] = pb;
    lo[i] = (short)(d > 32767 ? 32767 : d);
  }
}
__global__ __launch_bounds__(256) void master_join_kernel(const bf16* p, const short* lo, int64_t n, float* master) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    master[i] = __builtin_bit_cast(float, ((unsigned)__builtin_bit_cast(unsigned short, p[i]) << 16) + (unsigned)(int)lo[i]);
}

}  // namespace

extern "C" int mm_gradnorm_partial(int dtype, const void* g, int64_t n, float* partial, int nblk, void* stream) {
  if (!g || !partial || n < 0 || nblk <= 0) return MM_ERR_ARG;
  if (!mm_aligned16(g)) return MM_ERR_ALIGN;
  if (dtype == MM_BF16)
    hipLaunchKernelGGL(sumsq_kernel<bf16>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const bf16*)g, n, partial);
  else
    hipLaunchKernelGGL(sumsq_kernel<float>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const float*)g, n, partial);
  MM_CHECK_LAUNCH();
  return MM_OK;
}

extern "C" int mm_gradnorm_finish(const float* partial, int nblk, float max_norm, float* total, void* stream) {
  if (!partial || !total || nblk <= 0) return MM_ERR_ARG;
  hipLaunchKernelGGL(gradnorm_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, nblk, max_norm, total);
  MM_CHECK_LAUNCH();
  return MM_OK;
}

int g_adamw_blocks = 0;       // mm_set_option "adamw_blocks": grid cap of the update kernel (0 = MM_ADAMW_BLOCKS or 262144)

extern "C" int mm_adamw_step(int dtype, void* p, const void* g, float* master, float* m, float* v, int64_t n, float lr, float beta1,
                             float beta2, float eps, float weight_decay, int step, const float* clip, void* stream) {
  if (!p || !g || !master || !m || !v || n < 0 || step < 1) return MM_ERR_ARG;
  if (n == 0) return MM_OK;
  const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
  if (!mm_aligned16(p) || !mm_aligned16(g) || !mm_aligned16(master) || !mm_aligned16(m) || !mm_aligned16(v)) return MM_ERR_ALIGN;
  const int64_t nv4 = (n / 4 + 255) / 256;
  // grid cap: the update runs on a side stream under the next step's forward; MM_ADAMW_BLOCKS / "adamw_blocks" throttle how much of
  // the chip (and of HBM) it takes while the forward runs.  Round 4: 262 144 (one 4-element vector per thread for every tensor of the 8B
  // model, no grid-stride loop) instead of 2 048: stand-alone 5.1-5.9 -> 6.0-6.4 TB/s (tools/adamw_bench.py), the 8B step 347.6 -> 344.9 ms
  // median over six interleaved rounds (tools/step_ab.py "opt:adamw_blocks=0" "opt:adamw_blocks=262144").  (8 parameters per thread with
  // 16-byte p / g / remainder accesses: 3.8 instead of 6.2 TB/s -- m and v then go by 32-byte lane strides; removed.)
  static const int64_t env_cap = [] { const char* e = getenv("MM_ADAMW_BLOCKS"); const long c = e ? atol(e) : 0; return (int64_t)(c > 0 ? c : 262144); }();
  const int64_t cap = g_adamw_blocks > 0 ? g_adamw_blocks : env_cap;
  const unsigned nb = (unsigned)(nv4 < 1 ? 1 : (nv4 < cap ? nv4 : cap));
  static const bool nt = [] { const char* e = getenv("MM_ADAMW_NT"); return !e || e[0] != '0'; }();
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MM_BF16) {
    if (nt) hipLaunchKernelGGL((adamw_kernel<bf16, true>), dim3(nb), dim3(256), 0, st, (bf16*)p, (const bf16*)g, master, m, v, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2, clip);
    else hipLaunchKernelGGL((adamw_kernel<bf16, false>), dim3(nb), dim3(256), 0, st, (bf16*)p, (const bf16*)g, master, m, v, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2, clip);
  } else {
    hipLaunchKernelGGL((adamw_kernel<float, false>), dim3(nb), dim3(256), 0, st, (float*)p, (const float*)g, master, m, v, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2, clip);
  }
  MM_CHECK_LAUNCH();
  return MM_OK;
}

extern "C" int mm_adamw_step_split(void* p_bf16, const void* g_bf16, void* lo_i16, float* m, float* v, int64_t n, float lr, float beta1,
                                   float beta2, float eps, float weight_decay, int step, const float* clip, void* stream) {
  if (!p_bf16 || !g_bf16 || !lo_i16 || !m || !v || n < 0 || step < 1) return MM_ERR_ARG;
  if (n == 0) return MM_OK;
  const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
  if ((((uintptr_t)p_bf16) & 7) || (((uintptr_t)g_bf16) & 7) || (((uintptr_t)lo_i16) & 7) || !mm_aligned16(m) || !mm_aligned16(v)) return MM_ERR_ALIGN;
  const int64_t nv4 = (n / 4 + 255) / 256;
  static const int64_t env_cap = [] { const char* e = getenv("MM_ADAMW_BLOCKS"); const long c = e ? atol(e) : 0; return (int64_t)(c > 0 ? c : 262144); }();
  const int64_t cap = g_adamw_blocks > 0 ? g_adamw_blocks : env_cap;
  const unsigned nb = (unsigned)(nv4 < 1 ? 1 : (nv4 < cap ? nv4 : cap));
  static const bool nt = [] { const char* e = getenv("MM_ADAMW_NT"); return !e || e[0] != '0'; }();
  if (nt) hipLaunchKernelGGL((adamw_split_kernel<true>), dim3(nb), dim3(256), 0, (hipStream_t)stream, (bf16*)p_bf16, (const bf16*)g_bf16, (short*)lo_i16, m, v, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2, clip);
  else hipLaunchKernelGGL((adamw_split_kernel<false>), dim3(nb), dim3(256), 0, (hipStream_t)stream, (bf16*)p_bf16, (const bf16*)g_bf16, (short*)lo_i16, m, v, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2, clip);
  MM_CHECK_LAUNCH();
  return MM_OK;
}

extern "C" int mm_master_split(const float* master, int64_t n, void* p_bf16, void* lo_i16, void* stream) {
  if (n < 0 || (n > 0 && (!master || !p_bf16 || !lo_i16))) return MM_ERR_ARG;
  if (n == 0) return MM_OK;
  const unsigned nb = (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  hipLaunchKernelGGL(master_split_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, master, n, (bf16*)p_bf16, (short*)lo_i16);
  MM_CHECK_LAUNCH();
  return MM_OK;
}

extern "C" int mm_master_join(const void* p_bf16, const void* lo_i16, int64_t n, float* master, void* stream) {
  if (n < 0 || (n > 0 && (!master || !p_bf16 || !lo_i16))) return MM_ERR_ARG;
  if (n == 0) return MM_OK;
  const unsigned nb = (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  hipLaunchKernelGGL(master_join_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const bf16*)p_bf16, (const short*)lo_i16, n, master);
  MM_CHECK_LAUNCH();
  return MM_OK;
}
