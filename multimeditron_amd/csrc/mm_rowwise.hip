// HBM-bound row-wise / element-wise kernels of the decoder and vision tower:
// RMSNorm, LayerNorm, RoPE, SwiGLU, GELU, residual add, cross-entropy, arg-max.
// All: 16-byte vector accesses, fp32 math, wave-shuffle + small LDS reductions, one pass over HBM
// where the algorithm allows (rows up to 8K elements are kept in registers between the two sweeps).
#include "mm_common.h"

namespace {

constexpr int NORM_BWD_ROWS_PER_BLOCK = 16;   // 8192 rows -> 512 workgroups (2 per CU); dw partials stay small (nblk x H f32)

// ---------------------------------------------------------------- RMSNorm
// one block (256 threads) per row; row cached in registers (H <= 256*8*CH)
template <typename T, int CH>
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const T* x, const T* w, int M, int H, float eps, T* y, float* rstd) {
  __shared__ float red[8];
  constexpr int VN = Vec16<T>::N;
  const int row = blockIdx.x;
  const T* xr = x + (int64_t)row * H;
  T* yr = y + (int64_t)row * H;
  Vec16<T> xv[CH];
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int e = (c * 256 + threadIdx.x) * VN;
    if (e < H) {
      xv[c] = *(const Vec16<T>*)(xr + e);
#pragma unroll
      for (int i = 0; i < VN; ++i) { float f = xv[c].get(i); ss += f * f; }
    }
  }
  ss = block_sum_256(ss, red);
  const float rs = rsqrtf(ss / (float)H + eps);
  if (threadIdx.x == 0) rstd[row] = rs;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int e = (c * 256 + threadIdx.x) * VN;
    if (e < H) {
      Vec16<T> wv = *(const Vec16<T>*)(w + e), o;
#pragma unroll
      for (int i = 0; i < VN; ++i) {
        // HF: weight * hidden.to(input_dtype): the normalised value is rounded to the storage type first
        const float nrm = to_f32(from_f32<T>(xv[c].get(i) * rs));
        o.set(i, wv.get(i) * nrm);
      }
      *(Vec16<T>*)(yr + e) = o;
    }
  }
}

// block handles NORM_BWD_ROWS_PER_BLOCK rows; thread owns fixed columns -> dw partial in registers
template <typename T, int CH>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const T* dy, const T* x, const T* w, const float* rstd, int M, int H,
                                                          T* dx, float* dwp, const T* dres) {
  __shared__ float red[8];
  constexpr int VN = Vec16<T>::N;
  float dw[CH][VN];
  Vec16<T> wv[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int e = (c * 256 + threadIdx.x) * VN;
    if (e < H) wv[c] = *(const Vec16<T>*)(w + e);
#pragma unroll
    for (int i = 0; i < VN; ++i) dw[c][i] = 0.f;
  }
  const int r0 = blockIdx.x * NORM_BWD_ROWS_PER_BLOCK;
  const int r1 = min(M, r0 + NORM_BWD_ROWS_PER_BLOCK);
  for (int row = r0; row < r1; ++row) {
    const float rs = rstd[row];
    Vec16<T> xv[CH], gv[CH];
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int e = (c * 256 + threadIdx.x) * VN;
      if (e < H) {
        xv[c] = *(const Vec16<T>*)(x + (int64_t)row * H + e);
        gv[c] = *(const Vec16<T>*)(dy + (int64_t)row * H + e);
#pragma unroll
        for (int i = 0; i < VN; ++i) {
          const float xh = xv[c].get(i) * rs;
          const float g = gv[c].get(i);
          dw[c][i] += g * xh;
          dot += g * wv[c].get(i) * xh;
        }
      }
    }
    dot = block_sum_256(dot, red) / (float)H;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int e = (c * 256 + threadIdx.x) * VN;
      if (e < H) {
        Vec16<T> o, rv;
        if (dres) rv = *(const Vec16<T>*)(dres + (int64_t)row * H + e);
#pragma unroll
        for (int i = 0; i < VN; ++i) {
          const float xh = xv[c].get(i) * rs;
          o.set(i, rs * (gv[c].get(i) * wv[c].get(i) - xh * dot) + (dres ? rv.get(i) : 0.f));
        }
        *(Vec16<T>*)(dx + (int64_t)row * H + e) = o;
      }
    }
  }
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int e = (c * 256 + threadIdx.x) * VN;
    if (e < H)
#pragma unroll
      for (int i = 0; i < VN; ++i) dwp[(int64_t)blockIdx.x * H + e + i] = dw[c][i];
  }
}

// ---------------------------------------------------------------- LayerNorm
template <typename T, int CH>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const T* x, const T* w, const T* b, int M, int H, float eps, T* y,
                                                            float* mean, float* rstd) {
  __shared__ float red[8];
  constexpr int VN = Vec16<T>::N;
  const int row = blockIdx.x;
  const T* xr = x + (int64_t)row * H;
  Vec16<T> xv[CH];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int e = (c * 256 + threadIdx.x) * VN;
    if (e < H) {
      xv[c] = *(const Vec16<T>*)(xr + e);
#pragma unroll
      for (int i = 0; i < VN; ++i) s += xv[c].get(i);
    }
  }
  const float mu = block_sum_256(s, red) / (float)H;
  float v = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int e = (c * 256 + threadIdx.x) * VN;
    if (e < H)
#pragma unroll
      for (int i = 0; i < VN; ++i) { const float d = xv[c].get(i) - mu; v += d * d; }
  }
  const float rs = rsqrtf(block_sum_256(v, red) / (float)H + eps);
  if (threadIdx.x == 0) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int e = (c * 256 + threadIdx.x) * VN;
    if (e < H) {
      Vec16<T> wv = *(const Vec16<T>*)(w + e), bv = *(const Vec16<T>*)(b + e), o;
#pragma unroll
      for (int i = 0; i < VN; ++i) o.set(i, (xv[c].get(i) - mu) * rs * wv.get(i) + bv.get(i));
      *(Vec16<T>*)(y + (int64_t)row * H + e) = o;
    }
  }
}

template <typename T, int CH>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* dy, const T* x, const T* w, const float* mean,
                                                            const float* rstd, int M, int H, T* dx, float* dwp, float* dbp, const T* dres) {
  __shared__ float red[8];
  constexpr int VN = Vec16<T>::N;
  float dw[CH][VN], db[CH][VN];
  Vec16<T> wv[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int e = (c * 256 + threadIdx.x) * VN;
    if (e < H) wv[c] = *(const Vec16<T>*)(w + e);
#pragma unroll
    for (int i = 0; i < VN; ++i) { dw[c][i] = 0.f; db[c][i] = 0.f; }
  }
  const int r0 = blockIdx.x * NORM_BWD_ROWS_PER_BLOCK;
  const int r1 = min(M, r0 + NORM_BWD_ROWS_PER_BLOCK);
  for (int row = r0; row < r1; ++row) {
    const float rs = rstd[row], mu = mean[row];
    Vec16<T> xv[CH], gv[CH];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int e = (c * 256 + threadIdx.x) * VN;
      if (e < H) {
        xv[c] = *(const Vec16<T>*)(x + (int64_t)row * H + e);
        gv[c] = *(const Vec16<T>*)(dy + (int64_t)row * H + e);
#pragma unroll
        for (int i = 0; i < VN; ++i) {
          const float xh = (xv[c].get(i) - mu) * rs;
          const float g = gv[c].get(i);
          dw[c][i] += g * xh;
          db[c][i] += g;
          const float gw = g * wv[c].get(i);
          s1 += gw;
          s2 += gw * xh;
        }
      }
    }
    s1 = block_sum_256(s1, red) / (float)H;
    s2 = block_sum_256(s2, red) / (float)H;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int e = (c * 256 + threadIdx.x) * VN;
      if (e < H) {
        Vec16<T> o, rv;
        if (dres) rv = *(const Vec16<T>*)(dres + (int64_t)row * H + e);
#pragma unroll
        for (int i = 0; i < VN; ++i) {
          const float xh = (xv[c].get(i) - mu) * rs;
          o.set(i, rs * (gv[c].get(i) * wv[c].get(i) - s1 - xh * s2) + (dres ? rv.get(i) : 0.f));
        }
        *(Vec16<T>*)(dx + (int64_t)row * H + e) = o;
      }
    }
  }
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int e = (c * 256 + threadIdx.x) * VN;
    if (e < H)
#pragma unroll
      for (int i = 0; i < VN; ++i) {
        dwp[(int64_t)blockIdx.x * H + e + i] = dw[c][i];
        dbp[(int64_t)blockIdx.x * H + e + i] = db[c][i];
      }
  }
}

// out[h] (+)= sum_b partial[b][h].  16 columns per workgroup (H/16 workgroups: 256 for H = 4096, one per CU) x 64 row
// lanes, every lane's 16-byte loads of its rows all in flight at once, then a fixed-order tree through LDS (deterministic).
// The old shape (64 columns x 4 row lanes, 64 workgroups, 4-byte loads: 128 dependent rounds per thread) took 44 us on the
// decoder's 512 x 4096 partials (8.4 MB, L2/MALL resident); this one is bounded by one round of L2 latency.
// blockIdx.y selects one of two (partials, output) pairs: LayerNorm's dw and db leave backward in ONE launch (each launch of the
// ViT's backward chain costs ~90 us while the deferred weight-gradient GEMMs hold the CUs, whatever its own work is)
template <typename T>
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* p, int nblk, int H, T* out, int accumulate, const float* p1 = nullptr,
                                                              T* out1 = nullptr, int accumulate1 = 0) {
  __shared__ f32x4 red[64][4];
  if (blockIdx.y == 1) { p = p1; out = out1; accumulate = accumulate1; }
  const int c4 = threadIdx.x & 3, rl = threadIdx.x >> 2;
  const int h = blockIdx.x * 16 + c4 * 4;
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
  if (h < H) {                      // H % 4 == 0 (checked by the host entry)
    int b = rl;
    for (; b + 192 < nblk; b += 256) {
      const f32x4 v0 = *(const f32x4*)(p + (int64_t)b * H + h);
      const f32x4 v1 = *(const f32x4*)(p + (int64_t)(b + 64) * H + h);
      const f32x4 v2 = *(const f32x4*)(p + (int64_t)(b + 128) * H + h);
      const f32x4 v3 = *(const f32x4*)(p + (int64_t)(b + 192) * H + h);
      s0 += v0; s1 += v1; s2 += v2; s3 += v3;
    }
    for (; b < nblk; b += 64) s0 += *(const f32x4*)(p + (int64_t)b * H + h);
  }
  red[rl][c4] = (s0 + s1) + (s2 + s3);
  __syncthreads();
#pragma unroll
  for (int st = 32; st > 0; st >>= 1) {
    if (rl < st) red[rl][c4] += red[rl + st][c4];
    __syncthreads();
  }
  if (rl == 0 && h < H) {
    const f32x4 t = red[0][c4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v = t[i];
      if (accumulate) v += to_f32(out[h + i]);
      out[h + i] = from_f32<T>(v);
    }
  }
}

// ---------------------------------------------------------------- RoPE
__global__ void rope_table_kernel(const int64_t* pos, const float* inv_freq, int T, int half, int round_bf16, float* cs, float* sn) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)T * half) return;
  const int t = (int)(i / half), j = (int)(i % half);
  const float ang = (float)pos[t] * inv_freq[j];
  float c = cosf(ang), s = sinf(ang);
  if (round_bf16) { c = (float)(bf16)c; s = (float)(bf16)s; }
  cs[i] = c;
  sn[i] = s;
}

// x viewed [T, nheads, D] (row stride ld); thread handles 8 (bf16) / 4 (f32) consecutive j of one (t, head):
// out[j] = x[j]*c - x[j+half]*s ; out[j+half] = x[j+half]*c + x[j]*s   (inverse: s -> -s)
template <typename T>
__global__ void rope_apply_kernel(T* x, int Tn, int nheads, int D, int ld, const float* cs, const float* sn, int inverse) {
  constexpr int VN = Vec16<T>::N;
  const int half = D / 2;
  const int per_head = half / VN;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)Tn * nheads * per_head;
  if (i >= total) return;
  const int jc = (int)(i % per_head);
  const int h = (int)((i / per_head) % nheads);
  const int t = (int)(i / ((int64_t)per_head * nheads));
  T* p = x + (int64_t)t * ld + h * D + jc * VN;
  Vec16<T> a = *(Vec16<T>*)p, b = *(Vec16<T>*)(p + half), oa, ob;
  const float* c = cs + (int64_t)t * half + jc * VN;
  const float* s = sn + (int64_t)t * half + jc * VN;
#pragma unroll
  for (int k = 0; k < VN; ++k) {
    const float cc = c[k], ss = inverse ? -s[k] : s[k];
    oa.set(k, rope_lo(a.get(k), b.get(k), cc, ss));
    ob.set(k, rope_hi(a.get(k), b.get(k), cc, ss));
  }
  *(Vec16<T>*)p = oa;
  *(Vec16<T>*)(p + half) = ob;
}

// decode step: RoPE on the q and k heads of the fused projection output AND the append of the roped k / the v heads to
// the KV cache in the same pass (x [T, (Hq+2Hkv)*D] with row stride ld; kdst/vdst = the cache row of this step for
// sequence 0, dstride = elements between sequences)
template <typename T>
__global__ void rope_append_kernel(T* x, int Tn, int Hq, int Hkv, int D, int ld, const float* cs, const float* sn, T* kdst, T* vdst,
                                   int64_t dstride) {
  constexpr int VN = Vec16<T>::N;
  const int half = D / 2, per_head = half / VN, nh = Hq + 2 * Hkv;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)Tn * nh * per_head) return;
  const int jc = (int)(i % per_head);
  const int h = (int)((i / per_head) % nh);
  const int t = (int)(i / ((int64_t)per_head * nh));
  T* p = x + (int64_t)t * ld + h * D + jc * VN;
  Vec16<T> a = *(Vec16<T>*)p, b = *(Vec16<T>*)(p + half), oa = a, ob = b;
  if (h < Hq + Hkv) {
    const float* c = cs + (int64_t)t * half + jc * VN;
    const float* s = sn + (int64_t)t * half + jc * VN;
#pragma unroll
    for (int k = 0; k < VN; ++k) {
      oa.set(k, rope_lo(a.get(k), b.get(k), c[k], s[k]));
      ob.set(k, rope_hi(a.get(k), b.get(k), c[k], s[k]));
    }
    *(Vec16<T>*)p = oa;
    *(Vec16<T>*)(p + half) = ob;
  }
  if (h >= Hq) {
    T* d = (h < Hq + Hkv ? kdst + (h - Hq) * D : vdst + (h - Hq - Hkv) * D) + (int64_t)t * dstride + jc * VN;
    *(Vec16<T>*)d = oa;
    *(Vec16<T>*)(d + half) = ob;
  }
}

// ---------------------------------------------------------------- SwiGLU / GELU / add
template <typename T>
__global__ void swiglu_fwd_kernel(const T* gu, int M, int I, T* out) {
  constexpr int VN = Vec16<T>::N;
  const int per_row = I / VN;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)M * per_row) return;
  const int64_t m = i / per_row;
  const int c = (int)(i % per_row) * VN;
  Vec16<T> g = *(const Vec16<T>*)(gu + m * 2 * I + c), u = *(const Vec16<T>*)(gu + m * 2 * I + I + c), o;
#pragma unroll
  for (int k = 0; k < VN; ++k) {
    const float gf = g.get(k);
    // HF computes act_fn(gate) in the storage dtype, then multiplies: round silu(g) first
    const float sg = to_f32(from_f32<T>(gf / (1.0f + __expf(-gf))));
    o.set(k, sg * u.get(k));
  }
  *(Vec16<T>*)(out + m * I + c) = o;
}
template <typename T>
__global__ void swiglu_bwd_kernel(const T* gu, const T* dout, int M, int I, T* dgu) {
  constexpr int VN = Vec16<T>::N;
  const int per_row = I / VN;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)M * per_row) return;
  const int64_t m = i / per_row;
  const int c = (int)(i % per_row) * VN;
  Vec16<T> g = *(const Vec16<T>*)(gu + m * 2 * I + c), u = *(const Vec16<T>*)(gu + m * 2 * I + I + c);
  Vec16<T> d = *(const Vec16<T>*)(dout + m * I + c), dg, du;
#pragma unroll
  for (int k = 0; k < VN; ++k) {
    const float gf = g.get(k), sig = 1.0f / (1.0f + __expf(-gf));
    const float sg = gf * sig;
    const float dd = d.get(k);
    du.set(k, dd * sg);
    dg.set(k, dd * u.get(k) * (sig * (1.0f + gf * (1.0f - sig))));
  }
  *(Vec16<T>*)(dgu + m * 2 * I + c) = dg;
  *(Vec16<T>*)(dgu + m * 2 * I + I + c) = du;
}

template <typename T, int KIND, bool BWD>
__global__ void gelu_kernel(const T* x, const T* dy, int64_t n, T* y) {
  constexpr int VN = Vec16<T>::N;
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * VN;
  if (i >= n) return;
  auto f = [](float v) -> float {
    if (KIND == 0) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    if (KIND == 2) {
      const float u = 0.7978845608028654f * (v + 0.044715f * v * v * v);
      return 0.5f * v * (2.0f - 2.0f / (1.0f + __expf(2.0f * u)));
    }
    return v / (1.0f + __expf(-1.702f * v));
  };
  auto df = [](float v) -> float {
    if (KIND == 0) return 0.5f * (1.0f + erff(v * 0.70710678118654752440f)) + v * 0.39894228040143267794f * __expf(-0.5f * v * v);
    if (KIND == 2) {   // d/dv [0.5 v (1 + tanh u)], u = c (v + 0.044715 v^3)
      const float u = 0.7978845608028654f * (v + 0.044715f * v * v * v);
      const float th = 1.0f - 2.0f / (1.0f + __expf(2.0f * u));
      return 0.5f * (1.0f + th) + 0.5f * v * (1.0f - th * th) * 0.7978845608028654f * (1.0f + 3.0f * 0.044715f * v * v);
    }
    const float s = 1.0f / (1.0f + __expf(-1.702f * v));
    return s * (1.0f + 1.702f * v * (1.0f - s));
  };
  if (i + VN <= n) {
    Vec16<T> xv = *(const Vec16<T>*)(x + i), o;
    if (BWD) {
      Vec16<T> g = *(const Vec16<T>*)(dy + i);
#pragma unroll
      for (int k = 0; k < VN; ++k) o.set(k, g.get(k) * df(xv.get(k)));
    } else {
#pragma unroll
      for (int k = 0; k < VN; ++k) o.set(k, f(xv.get(k)));
    }
    *(Vec16<T>*)(y + i) = o;
  } else {
    for (int64_t k = i; k < n; ++k) y[k] = from_f32<T>(BWD ? to_f32(dy[k]) * df(to_f32(x[k])) : f(to_f32(x[k])));
  }
}

template <typename T>
__global__ void add_kernel(const T* a, const T* b, int64_t n, T* y) {
  constexpr int VN = Vec16<T>::N;
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * VN;
  if (i >= n) return;
  if (i + VN <= n) {
    Vec16<T> av = *(const Vec16<T>*)(a + i), bv = *(const Vec16<T>*)(b + i), o;
#pragma unroll
    for (int k = 0; k < VN; ++k) o.set(k, av.get(k) + bv.get(k));
    *(Vec16<T>*)(y + i) = o;
  } else {
    for (int64_t k = i; k < n; ++k) y[k] = from_f32<T>(to_f32(a[k]) + to_f32(b[k]));
  }
}

// ---------------------------------------------------------------- cross entropy
// one block per row; online (max, sum) in a single sweep of the logits row
template <typename T>
__global__ __launch_bounds__(256) void ce_fwd_kernel(const T* logits, int V, int ld, const int64_t* labels, float* lse, float* loss_row) {
  __shared__ float red[8];
  constexpr int VN = Vec16<T>::N;
  const int row = blockIdx.x;
  const T* p = logits + (int64_t)row * ld;
  float mx = -INFINITY, sm = 0.f;
  for (int e = threadIdx.x * VN; e < V; e += 256 * VN) {
    Vec16<T> v = *(const Vec16<T>*)(p + e);
    float lm = -INFINITY;
#pragma unroll
    for (int k = 0; k < VN; ++k)
      if (e + k < V) lm = fmaxf(lm, v.get(k));
    const float nm = fmaxf(mx, lm);
    float add = 0.f;
#pragma unroll
    for (int k = 0; k < VN; ++k)
      if (e + k < V) add += __expf(v.get(k) - nm);
    sm = sm * __expf(mx - nm) + add;
    mx = nm;
  }
  const float gmx = block_max_256(mx, red);
  const float part = (mx == -INFINITY) ? 0.f : sm * __expf(mx - gmx);
  const float tot = block_sum_256(part, red);
  if (threadIdx.x == 0) {
    const float l = gmx + logf(tot);
    lse[row] = l;
    const int64_t lab = labels[row];
    loss_row[row] = (lab >= 0 && lab < V) ? (l - to_f32(p[lab])) : 0.f;
  }
}

__global__ __launch_bounds__(256) void ce_reduce_kernel(const float* loss_row, const int64_t* labels, int T, float* out) {
  __shared__ float red[8];
  float s = 0.f, c = 0.f;
  for (int i = threadIdx.x; i < T; i += 256) {
    s += loss_row[i];
    c += (labels[i] >= 0) ? 1.f : 0.f;
  }
  s = block_sum_256(s, red);
  c = block_sum_256(c, red);
  if (threadIdx.x == 0) {
    out[0] = s / fmaxf(c, 1.f);
    out[1] = c;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const T* logits, int V, int ld, const int64_t* labels, const float* lse,
                                                     const float* lc, const float* gscale, T* dlogits) {
  constexpr int VN = Vec16<T>::N;
  const int row = blockIdx.x;
  const int64_t lab = labels[row];
  const bool live = lab >= 0 && lab < V;
  const float scale = live ? (gscale ? gscale[0] : 1.f) / fmaxf(lc[1], 1.f) : 0.f;
  const float l = lse[row];
  const T* p = logits + (int64_t)row * ld;
  T* d = dlogits + (int64_t)row * ld;
  for (int e = threadIdx.x * VN; e < ld; e += 256 * VN) {
    Vec16<T> o;
    if (live) {
      Vec16<T> v = *(const Vec16<T>*)(p + e);
#pragma unroll
      for (int k = 0; k < VN; ++k) {
        float g = 0.f;
        if (e + k < V) g = (__expf(v.get(k) - l) - ((e + k) == lab ? 1.f : 0.f)) * scale;
        o.set(k, g);
      }
    } else {
#pragma unroll
      for (int k = 0; k < VN; ++k) o.set(k, 0.f);
    }
    *(Vec16<T>*)(d + e) = o;
  }
}

// argmax(softmax(x / T)) in the logits dtype (model.py:607-621): softmax values are computed in fp32 then rounded
// to the storage type before comparison, so ties resolve exactly like torch.argmax over the softmax tensor
// (first index wins).
template <typename T>
__global__ __launch_bounds__(1024) void argmax_softmax_kernel(const T* logits, int V, int ld, float temperature, int64_t* out) {
  constexpr int NT = 1024, NW = NT / 64, VN = Vec16<T>::N;      // one row per block, 16 waves sweep the vocabulary three times
  __shared__ float red[NW];
  __shared__ float bestv[NW];
  __shared__ int besti[NW];
  const int row = blockIdx.x, w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const T* p = logits + (int64_t)row * ld;
  auto sc1 = [&](float x) { return to_f32(from_f32<T>(x / temperature)); };   // logits / T in the storage type
  // 16-byte loads (the row stride is a multiple of 16 bytes for the padded logits of the decoder; scalar loads otherwise and
  // for the ragged tail): each sweep was 125 dependent 2-byte loads per thread at V = 128 258 (115 us per token)
  const bool vec = ((ld * (int)sizeof(T)) & 15) == 0 && ((uintptr_t)logits & 15) == 0;
  const int Vv = vec ? V / VN * VN : 0;
  float mx = -INFINITY;
  for (int e = threadIdx.x * VN; e < Vv; e += NT * VN) {
    const Vec16<T> v = *(const Vec16<T>*)(p + e);
#pragma unroll
    for (int k = 0; k < VN; ++k) mx = fmaxf(mx, sc1(v.get(k)));
  }
  for (int e = Vv + threadIdx.x; e < V; e += NT) mx = fmaxf(mx, sc1(to_f32(p[e])));
  mx = wave_max(mx);
  if (l == 0) red[w] = mx;
  __syncthreads();
  mx = red[0];
#pragma unroll
  for (int i = 1; i < NW; ++i) mx = fmaxf(mx, red[i]);
  __syncthreads();
  float sm = 0.f;
  for (int e = threadIdx.x * VN; e < Vv; e += NT * VN) {
    const Vec16<T> v = *(const Vec16<T>*)(p + e);
#pragma unroll
    for (int k = 0; k < VN; ++k) sm += expf(sc1(v.get(k)) - mx);
  }
  for (int e = Vv + threadIdx.x; e < V; e += NT) sm += expf(sc1(to_f32(p[e])) - mx);
  sm = wave_sum(sm);
  if (l == 0) red[w] = sm;
  __syncthreads();
  sm = 0.f;
#pragma unroll
  for (int i = 0; i < NW; ++i) sm += red[i];
  float bv = -1.f;
  int bi = 0x7FFFFFFF;
  // (value desc, index asc): a thread meets its elements in ascending index order, so `>` keeps the first of equals
  for (int e = threadIdx.x * VN; e < Vv; e += NT * VN) {
    const Vec16<T> v = *(const Vec16<T>*)(p + e);
#pragma unroll
    for (int k = 0; k < VN; ++k) {
      const float pr = to_f32(from_f32<T>(expf(sc1(v.get(k)) - mx) / sm));
      if (pr > bv) { bv = pr; bi = e + k; }
    }
  }
  for (int e = Vv + threadIdx.x; e < V; e += NT) {
    const float pr = to_f32(from_f32<T>(expf(sc1(to_f32(p[e])) - mx) / sm));
    if (pr > bv || (pr == bv && e < bi)) { bv = pr; bi = e; }
  }
  // first index wins among equal probabilities: (value desc, index asc) order in every merge
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bv, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  if (l == 0) { bestv[w] = bv; besti[w] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < NW; ++i)
      if (bestv[i] > bv || (bestv[i] == bv && besti[i] < bi)) { bv = bestv[i]; bi = besti[i]; }
    out[row] = bi;
  }
}

// The same selection for a long row (the decoder's 128 258 logits: one 1024-thread block per row swept them three times in ~80 us
// per token) with the vocabulary cut into chunks of ARGMAX_CHUNK entries over many workgroups: chunk maxima and exp-sums, then --
// with the row's max and sum assembled from them in chunk order -- the best rounded probability of each chunk, merged across chunks
// by ONE 64-bit atomic max on (probability bits << 32 | 0x7FFFFFFF - index): larger probability first, smaller index among equals.
constexpr int ARGMAX_CHUNK = 4096;
template <typename T>
__global__ __launch_bounds__(256) void argmax_part_kernel(const T* logits, int V, int ld, float temperature, float* part, unsigned long long* packed) {
  __shared__ float red[4];
  const int c = blockIdx.x, row = blockIdx.y, C = gridDim.x;
  const T* p = logits + (int64_t)row * ld;
  auto sc1 = [&](float x) { return to_f32(from_f32<T>(x / temperature)); };
  const int e0 = c * ARGMAX_CHUNK, e1 = min(V, e0 + ARGMAX_CHUNK);
  float xv[ARGMAX_CHUNK / 256];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < ARGMAX_CHUNK / 256; ++i) {
    const int e = e0 + i * 256 + (int)threadIdx.x;
    xv[i] = e < e1 ? sc1(to_f32(p[e])) : -INFINITY;
    mx = fmaxf(mx, xv[i]);
  }
  mx = block_max_256(mx, red);
  float sm = 0.f;
#pragma unroll
  for (int i = 0; i < ARGMAX_CHUNK / 256; ++i) sm += (xv[i] == -INFINITY) ? 0.f : expf(xv[i] - mx);
  sm = block_sum_256(sm, red);
  if (threadIdx.x == 0) {
    part[((int64_t)row * C + c) * 2] = mx;
    part[((int64_t)row * C + c) * 2 + 1] = sm;
    if (c == 0) packed[row] = 0ull;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void argmax_pick_kernel(const T* logits, int V, int ld, float temperature, const float* part,
                                                          unsigned long long* packed) {
  __shared__ float bestv[4];
  __shared__ int besti[4];
  const int c = blockIdx.x, row = blockIdx.y, C = gridDim.x;
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const T* p = logits + (int64_t)row * ld;
  auto sc1 = [&](float x) { return to_f32(from_f32<T>(x / temperature)); };
  float mx = -INFINITY;
  for (int i = 0; i < C; ++i) mx = fmaxf(mx, part[((int64_t)row * C + i) * 2]);
  float sm = 0.f;
  for (int i = 0; i < C; ++i) {                              // chunk order: the same sum in every workgroup
    const float mi = part[((int64_t)row * C + i) * 2];
    if (mi != -INFINITY) sm += part[((int64_t)row * C + i) * 2 + 1] * expf(mi - mx);
  }
  const int e0 = c * ARGMAX_CHUNK, e1 = min(V, e0 + ARGMAX_CHUNK);
  float bv = -1.f;
  int bi = 0x7FFFFFFF;
#pragma unroll
  for (int i = 0; i < ARGMAX_CHUNK / 256; ++i) {             // ascending index per thread: `>` keeps the first of equals
    const int e = e0 + i * 256 + (int)threadIdx.x;
    if (e < e1) {
      const float pr = to_f32(from_f32<T>(expf(sc1(to_f32(p[e])) - mx) / sm));
      if (pr > bv) { bv = pr; bi = e; }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bv, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  if (l == 0) { bestv[w] = bv; besti[w] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 4; ++i)
      if (bestv[i] > bv || (bestv[i] == bv && besti[i] < bi)) { bv = bestv[i]; bi = besti[i]; }
    if (bv >= 0.f && bi != 0x7FFFFFFF) {
      const unsigned long long key = ((unsigned long long)__float_as_uint(bv) << 32) | (unsigned)(0x7FFFFFFF - bi);
      atomicMax(packed + row, key);
    }
  }
}

__global__ void argmax_unpack_kernel(const unsigned long long* packed, int rows, int V, int64_t* out) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < rows) {
    // a row whose probabilities are all NaN (NaN logits) issues no atomic max: packed stays 0 and the index would be 0x7FFFFFFF --
    // an out-of-range token id for the next embedding lookup.  Index 0 is what the one-launch kernel returns for such a row.
    const unsigned idx = 0x7FFFFFFFu - (unsigned)(packed[r] & 0xFFFFFFFFull);
    out[r] = idx < (unsigned)V ? (int64_t)idx : 0;
  }
}

// ---------------------------------------------------------------- MoE expert fusion (image_modality_moe.py:163-205)
// X [E, n, L] = the experts' token features (L = P*C), gate [n, E] fp32 = the gating network's softmax weights.
//   mode 0 (weighted_average, :170-176): out[n, L]    = sum_j  w(n, idx[j]) * X[idx[j], n, :]
//   mode 1 (cross_attn contexts, :190-199): out[n, j, L] = w'(n, j) * X[idx[j], n, :],  w' = softmax over the J listed experts
// fp32 arithmetic, one rounding.  Backward: dX[idx[j], n, :] = w * dout (mode 0: dout[n, :], mode 1: dout[n, j, :]).
constexpr int MOE_MAXJ = 16;
struct MoeIdx { int v[MOE_MAXJ]; };

template <typename T, bool BWD>
__global__ void expert_fuse_kernel(const T* X, const float* gate, MoeIdx idx, int J, int E, int n, int64_t L, int mode, T* out) {
  constexpr int VN = Vec16<T>::N;
  const int64_t per = L / VN;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)n * per) return;
  const int b = (int)(i / per);
  const int64_t c = (i % per) * VN;
  float w[MOE_MAXJ];
  float mx = -INFINITY, sm = 0.f;
  for (int j = 0; j < J; ++j) { w[j] = gate[(int64_t)b * E + idx.v[j]]; mx = fmaxf(mx, w[j]); }
  if (mode == 1) {
    for (int j = 0; j < J; ++j) { w[j] = expf(w[j] - mx); sm += w[j]; }
    for (int j = 0; j < J; ++j) w[j] /= sm;
  }
  if (!BWD) {
    if (mode == 0) {
      float acc[VN];
#pragma unroll
      for (int k = 0; k < VN; ++k) acc[k] = 0.f;
      for (int j = 0; j < J; ++j) {
        const Vec16<T> x = *(const Vec16<T>*)(X + ((int64_t)idx.v[j] * n + b) * L + c);
#pragma unroll
        for (int k = 0; k < VN; ++k) acc[k] += w[j] * x.get(k);
      }
      Vec16<T> o;
#pragma unroll
      for (int k = 0; k < VN; ++k) o.set(k, acc[k]);
      *(Vec16<T>*)(out + (int64_t)b * L + c) = o;
    } else {
      for (int j = 0; j < J; ++j) {
        const Vec16<T> x = *(const Vec16<T>*)(X + ((int64_t)idx.v[j] * n + b) * L + c);
        Vec16<T> o;
#pragma unroll
        for (int k = 0; k < VN; ++k) o.set(k, w[j] * x.get(k));
        *(Vec16<T>*)(out + ((int64_t)b * J + j) * L + c) = o;
      }
    }
  } else {      // X = dout, out = dX [E, n, L] (slices of the listed experts)
    for (int j = 0; j < J; ++j) {
      const Vec16<T> d = *(const Vec16<T>*)(X + (mode == 0 ? (int64_t)b * L : ((int64_t)b * J + j) * L) + c);
      Vec16<T> o;
#pragma unroll
      for (int k = 0; k < VN; ++k) o.set(k, w[j] * d.get(k));
      *(Vec16<T>*)(out + ((int64_t)idx.v[j] * n + b) * L + c) = o;
    }
  }
}

// ---------------------------------------------------------------- cast
template <typename S, typename D>
__global__ void cast_kernel(const S* s, D* d, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t k = i; k < n; k += stride) d[k] = from_f32<D>(to_f32(s[k]));
}

template <typename T> inline int norm_ch(int H) { return (H + 256 * Vec16<T>::N - 1) / (256 * Vec16<T>::N); }

}  // namespace

#define DISPATCH_CH(T, ch, ...)                            \
  switch (ch) {                                            \
    case 1: { constexpr int CH = 1; __VA_ARGS__; } break;  \
    case 2: { constexpr int CH = 2; __VA_ARGS__; } break;  \
    case 3: case 4: { constexpr int CH = 4; __VA_ARGS__; } break; \
    case 5: case 6: case 7: case 8: { constexpr int CH = 8; __VA_ARGS__; } break; \
    default: return MM_ERR_UNSUPPORTED;                    \
  }

extern "C" int mm_norm_bwd_blocks(int M) { return (M + NORM_BWD_ROWS_PER_BLOCK - 1) / NORM_BWD_ROWS_PER_BLOCK; }

extern "C" int mm_rmsnorm_fwd(int dtype, const void* x, const void* w, int M, int H, float eps, void* y, float* rstd, void* stream) {
  if (!x || !w || !y || !rstd || M < 0 || H <= 0) return MM_ERR_ARG;
  if (M == 0) return MM_OK;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MM_BF16) {
    if (H % 8) return MM_ERR_ALIGN;
    DISPATCH_CH(bf16, norm_ch<bf16>(H), hipLaunchKernelGGL((rmsnorm_fwd_kernel<bf16, CH>), dim3(M), dim3(256), 0, s, (const bf16*)x, (const bf16*)w, M, H, eps, (bf16*)y, rstd));
  } else {
    if (H % 4) return MM_ERR_ALIGN;
    DISPATCH_CH(float, norm_ch<float>(H), hipLaunchKernelGGL((rmsnorm_fwd_kernel<float, CH>), dim3(M), dim3(256), 0, s, (const float*)x, (const float*)w, M, H, eps, (float*)y, rstd));
  }
  MM_CHECK_LAUNCH();
  return MM_OK;
}

extern "C" int mm_rmsnorm_bwd(int dtype, const void* dy, const void* x, const void* w, const float* rstd, int M, int H, void* dx,
                              float* dwp, const void* dres, void* stream) {
  if (!dy || !x || !w || !rstd || !dx || !dwp || M < 0 || H <= 0) return MM_ERR_ARG;
  if (M == 0) return MM_OK;
  hipStream_t s = (hipStream_t)stream;
  const int nb = mm_norm_bwd_blocks(M);
  if (dtype == MM_BF16) {
    if (H % 8) return MM_ERR_ALIGN;
    DISPATCH_CH(bf16, norm_ch<bf16>(H), hipLaunchKernelGGL((rmsnorm_bwd_kernel<bf16, CH>), dim3(nb), dim3(256), 0, s, (const bf16*)dy, (const bf16*)x, (const bf16*)w, rstd, M, H, (bf16*)dx, dwp, (const bf16*)dres));
  } else {
    if (H % 4) return MM_ERR_ALIGN;
    DISPATCH_CH(float, norm_ch<float>(H), hipLaunchKernelGGL((rmsnorm_bwd_kernel<float, CH>), dim3(nb), dim3(256), 0, s, (const float*)dy, (const float*)x, (const float*)w, rstd, M, H, (float*)dx, dwp, (const float*)dres));
  }
  MM_CHECK_LAUNCH();
  return MM_OK;
}

extern "C" int mm_layernorm_fwd(int dtype, const void* x, const void* w, const void* b, int M, int H, float eps, void* y,
                                float* mean, float* rstd, void* stream) {
  if (!x || !w || !b || !y || !mean || !rstd || M < 0 || H <= 0) return MM_ERR_ARG;
  if (M == 0) return MM_OK;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MM_BF16) {
    if (H % 8) return MM_ERR_ALIGN;
    DISPATCH_CH(bf16, norm_ch<bf16>(H), hipLaunchKernelGGL((layernorm_fwd_kernel<bf16, CH>), dim3(M), dim3(256), 0, s, (const bf16*)x, (const bf16*)w, (const bf16*)b, M, H, eps, (bf16*)y, mean, rstd));
  } else {
    if (H % 4) return MM_ERR_ALIGN;
    DISPATCH_CH(float, norm_ch<float>(H), hipLaunchKernelGGL((layernorm_fwd_kernel<float, CH>), dim3(M), dim3(256), 0, s, (const float*)x, (const float*)w, (const float*)b, M, H, eps, (float*)y, mean, rstd));
  }
  MM_CHECK_LAUNCH();
  return MM_OK;
}

extern "C" int mm_layernorm_bwd(int dtype, const void* dy, const void* x, const void* w, const float* mean, const float* rstd,
                                int M, int H, void* dx, float* dwp, float* dbp, const void* dres, void* stream) {
  if (!dy || !x || !w || !mean || !rstd || !dx || !dwp || !dbp || M < 0 || H <= 0) return MM_ERR_ARG;
  if (M == 0) return MM_OK;
  hipStream_t s = (hipStream_t)stream;
  const int nb = mm_norm_bwd_blocks(M);
  if (dtype == MM_BF16) {
    if (H % 8) return MM_ERR_ALIGN;
    DISPATCH_CH(bf16, norm_ch<bf16>(H), hipLaunchKernelGGL((layernorm_bwd_kernel<bf16, CH>), dim3(nb), dim3(256), 0, s, (const bf16*)dy, (const bf16*)x, (const bf16*)w, mean, rstd, M, H, (bf16*)dx, dwp, dbp, (const bf16*)dres));
  } else {
    if (H % 4) return MM_ERR_ALIGN;
    DISPATCH_CH(float, norm_ch<float>(H), hipLaunchKernelGGL((layernorm_bwd_kernel<float, CH>), dim3(nb), dim3(256), 0, s, (const float*)dy, (const float*)x, (const float*)w, mean, rstd, M, H, (float*)dx, dwp, dbp, (const float*)dres));
  }
  MM_CHECK_LAUNCH();
  return MM_OK;
}

extern "C" int mm_reduce_partials2(int dtype, const float* partial0, const float* partial1, int nblk, int H, void* out0, void* out1, int accumulate0,
                                   int accumulate1, void* stream) {
  if (!partial0 || !partial1 || !out0 || !out1 || nblk < 0 || H <= 0) return MM_ERR_ARG;
  if ((H & 3) || !mm_aligned16(partial0) || !mm_aligned16(partial1)) return MM_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((H + 15) / 16, 2), block(256);
  if (dtype == MM_BF16)
    hipLaunchKernelGGL(reduce_partials_kernel<bf16>, grid, block, 0, s, partial0, nblk, H, (bf16*)out0, accumulate0, partial1, (bf16*)out1, accumulate1);
  else
    hipLaunchKernelGGL(reduce_partials_kernel<float>, grid, block, 0, s, partial0, nblk, H, (float*)out0, accumulate0, partial1, (float*)out1, accumulate1);
  MM_CHECK_LAUNCH();
  return MM_OK;
}

extern "C" int mm_reduce_partials(int dtype, const float* partial, int nblk, int H, void* out, int accumulate, void* stream) {
  if (!partial || !out || nblk < 0 || H <= 0) return MM_ERR_ARG;
  if ((H & 3) || !mm_aligned16(partial)) return MM_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((H + 15) / 16), block(256);
  if (dtype == MM_BF16)
    hipLaunchKernelGGL(reduce_partials_kernel<bf16>, grid, block, 0, s, partial, nblk, H, (bf16*)out, accumulate);
  else
    hipLaunchKernelGGL(reduce_partials_kernel<float>, grid, block, 0, s, partial, nblk, H, (float*)out, accumulate);
  MM_CHECK_LAUNCH();
  return MM_OK;
}

extern "C" int mm_rope_table(const int64_t* position_ids, const float* inv_freq, int T, int half, int round_bf16, float* cos_t,
                             float* sin_t, void* stream) {
  if (!position_ids || !inv_freq || !cos_t || !sin_t || T < 0 || half <= 0) return MM_ERR_ARG;
  if (T == 0) return MM_OK;
  const int64_t n = (int64_t)T * half;
  hipLaunchKernelGGL(rope_table_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, position_ids, inv_freq, T,
                     half, round_bf16, cos_t, sin_t);
  MM_CHECK_LAUNCH();
  return MM_OK;
}

extern "C" int mm_rope_apply(int dtype, void* x, int T, int nheads, int D, int ld, const float* cos_t, const float* sin_t, int inverse,
                             void* stream) {
  if (!x || !cos_t || !sin_t || T < 0 || nheads <= 0 || D <= 0) return MM_ERR_ARG;
  if (T == 0) return MM_OK;
  const int vn = dtype == MM_BF16 ? 8 : 4;
  if ((D / 2) % vn || (ld % vn) || !mm_aligned16(x)) return MM_ERR_ALIGN;
  const int64_t total = (int64_t)T * nheads * (D / 2 / vn);
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (dtype == MM_BF16)
    hipLaunchKernelGGL(rope_apply_kernel<bf16>, grid, block, 0, (hipStream_t)stream, (bf16*)x, T, nheads, D, ld, cos_t, sin_t, inverse);
  else
    hipLaunchKernelGGL(rope_apply_kernel<float>, grid, block, 0, (hipStream_t)stream, (float*)x, T, nheads, D, ld, cos_t, sin_t, inverse);
  MM_CHECK_LAUNCH();
  return MM_OK;
}

extern "C" int mm_rope_append(int dtype, void* x, int T, int Hq, int Hkv, int D, int ld, const float* cos_t, const float* sin_t,
                              void* kdst, void* vdst, int64_t dstride, void* stream) {
  if (!x || !cos_t || !sin_t || !kdst || !vdst || T < 0 || Hq <= 0 || Hkv <= 0 || D <= 0) return MM_ERR_ARG;
  if (dtype != MM_BF16 && dtype != MM_F32) return MM_ERR_UNSUPPORTED;
  if (T == 0) return MM_OK;
  const int vn = dtype == MM_BF16 ? 8 : 4;
  if ((D / 2) % vn || (ld % vn) || (dstride % vn) || !mm_aligned16(x) || !mm_aligned16(kdst) || !mm_aligned16(vdst)) return MM_ERR_ALIGN;
  const int64_t total = (int64_t)T * (Hq + 2 * Hkv) * (D / 2 / vn);
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (dtype == MM_BF16)
    hipLaunchKernelGGL(rope_append_kernel<bf16>, grid, block, 0, (hipStream_t)stream, (bf16*)x, T, Hq, Hkv, D, ld, cos_t, sin_t, (bf16*)kdst, (bf16*)vdst, dstride);
  else
    hipLaunchKernelGGL(rope_append_kernel<float>, grid, block, 0, (hipStream_t)stream, (float*)x, T, Hq, Hkv, D, ld, cos_t, sin_t, (float*)kdst, (float*)vdst, dstride);
  MM_CHECK_LAUNCH();
  return MM_OK;
}

extern "C" int mm_swiglu_fwd(int dtype, const void* gu, int M, int I, void* out, void* stream) {
  if (!gu || !out || M < 0 || I <= 0) return MM_ERR_ARG;
  if (M == 0) return MM_OK;
  const int vn = dtype == MM_BF16 ? 8 : 4;
  if (I % vn) return MM_ERR_ALIGN;
  const int64_t total = (int64_t)M * (I / vn);
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (dtype == MM_BF16)
    hipLaunchKernelGGL(swiglu_fwd_kernel<bf16>, grid, block, 0, (hipStream_t)stream, (const bf16*)gu, M, I, (bf16*)out);
  else
    hipLaunchKernelGGL(swiglu_fwd_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float*)gu, M, I, (float*)out);
  MM_CHECK_LAUNCH();
  return MM_OK;
}

extern "C" int mm_swiglu_bwd(int dtype, const void* gu, const void* dout, int M, int I, void* dgu, void* stream) {
  if (!gu || !dout || !dgu || M < 0 || I <= 0) return MM_ERR_ARG;
  if (M == 0) return MM_OK;
  const int vn = dtype == MM_BF16 ? 8 : 4;
  if (I % vn) return MM_ERR_ALIGN;
  const int64_t total = (int64_t)M * (I / vn);
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (dtype == MM_BF16)
    hipLaunchKernelGGL(swiglu_bwd_kernel<bf16>, grid, block, 0, (hipStream_t)stream, (const bf16*)gu, (const bf16*)dout, M, I, (bf16*)dgu);
  else
    hipLaunchKernelGGL(swiglu_bwd_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float*)gu, (const float*)dout, M, I, (float*)dgu);
  MM_CHECK_LAUNCH();
  return MM_OK;
}

template <typename T, bool BWD>
static int gelu_launch(int kind, const void* x, const void* dy, int64_t n, void* y, hipStream_t s) {
  constexpr int VN = Vec16<T>::N;
  dim3 grid((unsigned)((n / VN + 256) / 256)), block(256);
  if (kind == 0)
    hipLaunchKernelGGL((gelu_kernel<T, 0, BWD>), grid, block, 0, s, (const T*)x, (const T*)dy, n, (T*)y);
  else if (kind == 1)
    hipLaunchKernelGGL((gelu_kernel<T, 1, BWD>), grid, block, 0, s, (const T*)x, (const T*)dy, n, (T*)y);
  else
    hipLaunchKernelGGL((gelu_kernel<T, 2, BWD>), grid, block, 0, s, (const T*)x, (const T*)dy, n, (T*)y);
  MM_CHECK_LAUNCH();
  return MM_OK;
}

extern "C" int mm_gelu_fwd(int dtype, int kind, const void* x, int64_t n, void* y, void* stream) {
  if (!x || !y || n < 0 || kind < 0 || kind > 2) return MM_ERR_ARG;
  if (n == 0) return MM_OK;
  return dtype == MM_BF16 ? gelu_launch<bf16, false>(kind, x, nullptr, n, y, (hipStream_t)stream)
                          : gelu_launch<float, false>(kind, x, nullptr, n, y, (hipStream_t)stream);
}
extern "C" int mm_gelu_bwd(int dtype, int kind, const void* x, const void* dy, int64_t n, void* dx, void* stream) {
  if (!x || !dy || !dx || n < 0 || kind < 0 || kind > 2) return MM_ERR_ARG;
  if (n == 0) return MM_OK;
  return dtype == MM_BF16 ? gelu_launch<bf16, true>(kind, x, dy, n, dx, (hipStream_t)stream)
                          : gelu_launch<float, true>(kind, x, dy, n, dx, (hipStream_t)stream);
}

extern "C" int mm_add(int dtype, const void* a, const void* b, int64_t n, void* y, void* stream) {
  if (!a || !b || !y || n < 0) return MM_ERR_ARG;
  if (n == 0) return MM_OK;
  const int vn = dtype == MM_BF16 ? 8 : 4;
  dim3 grid((unsigned)((n / vn + 256) / 256)), block(256);
  if (dtype == MM_BF16)
    hipLaunchKernelGGL(add_kernel<bf16>, grid, block, 0, (hipStream_t)stream, (const bf16*)a, (const bf16*)b, n, (bf16*)y);
  else
    hipLaunchKernelGGL(add_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float*)a, (const float*)b, n, (float*)y);
  MM_CHECK_LAUNCH();
  return MM_OK;
}

extern "C" int mm_ce_fwd(int dtype, const void* logits, int T, int V, int ld, const int64_t* labels, float* lse, float* loss_row,
                         void* stream) {
  if (!logits || !labels || !lse || !loss_row || T < 0 || V <= 0 || ld < V) return MM_ERR_ARG;
  if (T == 0) return MM_OK;
  const int vn = dtype == MM_BF16 ? 8 : 4;
  if (ld % vn || !mm_aligned16(logits)) return MM_ERR_ALIGN;
  if (dtype == MM_BF16)
    hipLaunchKernelGGL(ce_fwd_kernel<bf16>, dim3(T), dim3(256), 0, (hipStream_t)stream, (const bf16*)logits, V, ld, labels, lse, loss_row);
  else
    hipLaunchKernelGGL(ce_fwd_kernel<float>, dim3(T), dim3(256), 0, (hipStream_t)stream, (const float*)logits, V, ld, labels, lse, loss_row);
  MM_CHECK_LAUNCH();
  return MM_OK;
}

extern "C" int mm_ce_reduce(const float* loss_row, const int64_t* labels, int T, float* out, void* stream) {
  if (!loss_row || !labels || !out || T < 0) return MM_ERR_ARG;
  hipLaunchKernelGGL(ce_reduce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, loss_row, labels, T, out);
  MM_CHECK_LAUNCH();
  return MM_OK;
}

extern "C" int mm_ce_bwd(int dtype, const void* logits, int T, int V, int ld, const int64_t* labels, const float* lse,
                         const float* loss_and_count, const float* gscale, void* dlogits, void* stream) {
  if (!logits || !labels || !lse || !loss_and_count || !dlogits || T < 0 || V <= 0 || ld < V) return MM_ERR_ARG;
  if (T == 0) return MM_OK;
  const int vn = dtype == MM_BF16 ? 8 : 4;
  if (ld % vn || !mm_aligned16(logits) || !mm_aligned16(dlogits)) return MM_ERR_ALIGN;
  if (dtype == MM_BF16)
    hipLaunchKernelGGL(ce_bwd_kernel<bf16>, dim3(T), dim3(256), 0, (hipStream_t)stream, (const bf16*)logits, V, ld, labels, lse,
                       loss_and_count, gscale, (bf16*)dlogits);
  else
    hipLaunchKernelGGL(ce_bwd_kernel<float>, dim3(T), dim3(256), 0, (hipStream_t)stream, (const float*)logits, V, ld, labels, lse,
                       loss_and_count, gscale, (float*)dlogits);
  MM_CHECK_LAUNCH();
  return MM_OK;
}

extern "C" int mm_argmax_softmax_ws_bytes(int rows, int V) {
  if (rows <= 0 || V <= 0) return 0;
  const int64_t C = (V + ARGMAX_CHUNK - 1) / ARGMAX_CHUNK;
  const int64_t b = (int64_t)rows * 8 + (int64_t)rows * C * 2 * 4;       // packed keys, then (max, sum) per chunk
  return b > 0x7FFFFFFF ? -1 : (int)b;
}

extern "C" int mm_argmax_softmax_split(int dtype, const void* logits, int rows, int V, int ld, float temperature, int64_t* out, void* ws,
                                       void* stream) {
  if (!logits || !out || !ws || rows < 0 || V <= 0 || ld < V || !(temperature > 0.f)) return MM_ERR_ARG;
  if (((uintptr_t)ws) & 7) return MM_ERR_ALIGN;
  if (rows == 0) return MM_OK;
  const int C = (V + ARGMAX_CHUNK - 1) / ARGMAX_CHUNK;
  unsigned long long* packed = (unsigned long long*)ws;
  float* part = (float*)(packed + rows);
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)C, (unsigned)rows), block(256);
  if (dtype == MM_BF16) {
    hipLaunchKernelGGL(argmax_part_kernel<bf16>, grid, block, 0, s, (const bf16*)logits, V, ld, temperature, part, packed);
    hipLaunchKernelGGL(argmax_pick_kernel<bf16>, grid, block, 0, s, (const bf16*)logits, V, ld, temperature, (const float*)part, packed);
  } else {
    hipLaunchKernelGGL(argmax_part_kernel<float>, grid, block, 0, s, (const float*)logits, V, ld, temperature, part, packed);
    hipLaunchKernelGGL(argmax_pick_kernel<float>, grid, block, 0, s, (const float*)logits, V, ld, temperature, (const float*)part, packed);
  }
  hipLaunchKernelGGL(argmax_unpack_kernel, dim3((rows + 63) / 64), dim3(64), 0, s, (const unsigned long long*)packed, rows, V, out);
  MM_CHECK_LAUNCH();
  return MM_OK;
}

extern "C" int mm_argmax_softmax(int dtype, const void* logits, int rows, int V, int ld, float temperature, int64_t* out, void* stream) {
  if (!logits || !out || rows < 0 || V <= 0 || ld < V || !(temperature > 0.f)) return MM_ERR_ARG;
  if (rows == 0) return MM_OK;
  if (dtype == MM_BF16)
    hipLaunchKernelGGL(argmax_softmax_kernel<bf16>, dim3(rows), dim3(1024), 0, (hipStream_t)stream, (const bf16*)logits, V, ld, temperature, out);
  else
    hipLaunchKernelGGL(argmax_softmax_kernel<float>, dim3(rows), dim3(1024), 0, (hipStream_t)stream, (const float*)logits, V, ld, temperature, out);
  MM_CHECK_LAUNCH();
  return MM_OK;
}

// generate()'s per-token bookkeeping on the device (reference model.py:618-625 does it on the host after a sync): a row that
// has already emitted eos keeps emitting eos, `finished` is updated, the chosen id goes to column `col` of the output and
// to `next_ids` (the next step's embedding lookup).
__global__ void decode_select_kernel(const int64_t* tok, unsigned char* finished, int64_t eos, int B, int64_t* out, int ld_out, int col,
                                     int64_t* next_ids) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int64_t t = finished[b] ? eos : tok[b];
  finished[b] = (unsigned char)(finished[b] | (t == eos));
  out[(int64_t)b * ld_out + col] = t;
  next_ids[b] = t;
}

extern "C" int mm_decode_select(const int64_t* tok, unsigned char* finished, int64_t eos, int B, int64_t* out, int ld_out, int col,
                                int64_t* next_ids, void* stream) {
  if (!tok || !finished || !out || !next_ids || B < 0 || col < 0 || col >= ld_out) return MM_ERR_ARG;
  if (B == 0) return MM_OK;
  hipLaunchKernelGGL(decode_select_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, tok, finished, eos, B, out, ld_out, col,
                     next_ids);
  MM_CHECK_LAUNCH();
  return MM_OK;
}

extern "C" int mm_expert_fuse(int dtype, int backward, int mode, const void* X, const float* gate, const int* idx, int J, int E, int n,
                              int64_t L, void* out, void* stream) {
  if (!X || !gate || !idx || !out || J <= 0 || J > MOE_MAXJ || E <= 0 || n < 0 || L <= 0 || mode < 0 || mode > 1) return MM_ERR_ARG;
  if (n == 0) return MM_OK;
  const int vn = dtype == MM_BF16 ? 8 : 4;
  if ((L % vn) || !mm_aligned16(X) || !mm_aligned16(out)) return MM_ERR_ALIGN;
  MoeIdx ix{};
  for (int j = 0; j < J; ++j) {
    if (idx[j] < 0 || idx[j] >= E) return MM_ERR_ARG;
    ix.v[j] = idx[j];
  }
  const int64_t total = (int64_t)n * (L / vn);
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MM_BF16) {
    if (backward) hipLaunchKernelGGL((expert_fuse_kernel<bf16, true>), grid, block, 0, s, (const bf16*)X, gate, ix, J, E, n, L, mode, (bf16*)out);
    else hipLaunchKernelGGL((expert_fuse_kernel<bf16, false>), grid, block, 0, s, (const bf16*)X, gate, ix, J, E, n, L, mode, (bf16*)out);
  } else if (dtype == MM_F32) {
    if (backward) hipLaunchKernelGGL((expert_fuse_kernel<float, true>), grid, block, 0, s, (const float*)X, gate, ix, J, E, n, L, mode, (float*)out);
    else hipLaunchKernelGGL((expert_fuse_kernel<float, false>), grid, block, 0, s, (const float*)X, gate, ix, J, E, n, L, mode, (float*)out);
  } else {
    return MM_ERR_UNSUPPORTED;
  }
  MM_CHECK_LAUNCH();
  return MM_OK;
}

extern "C" int mm_cast(int src_dtype, int dst_dtype, const void* src, void* dst, int64_t n, void* stream) {
  if (!src || !dst || n < 0) return MM_ERR_ARG;
  if (n == 0) return MM_OK;
  dim3 grid((unsigned)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (src_dtype == MM_F32 && dst_dtype == MM_BF16)
    hipLaunchKernelGGL((cast_kernel<float, bf16>), grid, block, 0, s, (const float*)src, (bf16*)dst, n);
  else if (src_dtype == MM_BF16 && dst_dtype == MM_F32)
    hipLaunchKernelGGL((cast_kernel<bf16, float>), grid, block, 0, s, (const bf16*)src, (float*)dst, n);
  else if (src_dtype == MM_F32 && dst_dtype == MM_F32)
    hipLaunchKernelGGL((cast_kernel<float, float>), grid, block, 0, s, (const float*)src, (float*)dst, n);
  else if (src_dtype == MM_BF16 && dst_dtype == MM_BF16)
    hipLaunchKernelGGL((cast_kernel<bf16, bf16>), grid, block, 0, s, (const bf16*)src, (bf16*)dst, n);
  else
    return MM_ERR_UNSUPPORTED;
  MM_CHECK_LAUNCH();
  return MM_OK;
}

extern "C" int mm_fill_zero(void* p, int64_t bytes, void* stream) {
  if (!p || bytes < 0) return MM_ERR_ARG;
  if (bytes == 0) return MM_OK;
  return hipMemsetAsync(p, 0, (size_t)bytes, (hipStream_t)stream) == hipSuccess ? MM_OK : MM_ERR_LAUNCH;
}

extern "C" int mm_version(void) { return 100; }
extern "C" const char* mm_error_string(int code) {
  switch (code) {
    case MM_OK: return "ok";
    case MM_ERR_ARG: return "invalid argument";
    case MM_ERR_ALIGN: return "alignment / leading-dimension requirement not met";
    case MM_ERR_UNSUPPORTED: return "unsupported dtype or shape";
    case MM_ERR_LAUNCH: return "kernel launch failed";
    default: return "unknown error";
  }
}
