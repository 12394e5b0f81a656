// Small cross-attention with attention-probability dropout: the core of `CrossAttention` of the MoE image modalities
// (reference model/attention.py:79-96: softmax(q k^T * scale) -> attn_drop -> @ v, between the q/k/v projections and `proj`),
// forward + backward, plus the element-wise dropout of its output projection (attention.py:97-99, `proj_drop`).
//
// Shapes of the reference's recipes (cookbook/sft/moe/*/attn/{shared,pep}): P = 49 generalist queries over (E-1)*P = 196
// specialist keys, 8 heads of width 96 (ViT-B/32 experts, C = 768) or 512 (per-expert projection, C = 4096): neither width is
// tiled by the flash kernels of mm_attn.hip (64 / 128), and the whole score row of a query fits a wave's registers, so this is
// NOT a flash kernel: one pass, no online softmax, any head width that is a multiple of 8 up to 512, up to 1024 keys (above 512: the 64-tile instantiation, which spills -- a cross-attention is too small for that to matter; round 4).
//
// bf16 (MFMA v_mfma_f32_16x16x32_bf16, fp32 accumulate), key-major like mm_attn.hip:
//   S^T[key][q] = K . Q^T      A = 16 keys x 32 d and B = 32 d x 16 queries, both straight from global memory (d contiguous)
//   O^T[d][q]   = V^T . P^T    B = P^T taken from the S^T accumulators (the lane already owns its query's keys), A = V^T by
//                              ds_read_b64_tr_b16 from a row-major LDS image of a 64-column slice of V
// backward, deterministic (no atomics): kernel 1 per query tile recomputes P, forms dS and dQ (same shape as the forward) and
// leaves P_drop^T and dS^T [key][q] in a workspace; kernel 2 per 64 keys sums dV = P_drop^T dO and dK = dS^T Q over the queries.
// Dropout: Philox4x32-10 keyed by (seed, offset); the keep bit of probability (row, key) is component (key & 3) of call
// row * KP/4 + key/4 (KP = keys rounded up to 16), so forward, backward, the fp32 kernels and mm_dropout_mask agree bit for bit.
//
// fp32 (parity path): plain wave-per-query kernels, exact fp32 chains, same dropout stream, no atomics either.
#include <stdlib.h>
#include <string.h>

#include "mm_common.h"

namespace {

// ---------------------------------------------------------------------------------------------- Philox4x32-10
struct U4 { unsigned x, y, z, w; };

__host__ __device__ inline unsigned mulhi32(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }

__host__ __device__ inline U4 philox4x32_10(unsigned long long call, unsigned long long offset, unsigned long long seed) {
  unsigned c0 = (unsigned)call, c1 = (unsigned)(call >> 32), c2 = (unsigned)offset, c3 = (unsigned)(offset >> 32);
  unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned h0 = mulhi32(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
    const unsigned h1 = mulhi32(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
    const unsigned n0 = h1 ^ c1 ^ k0, n1 = l1, n2 = h0 ^ c3 ^ k1, n3 = l0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return U4{c0, c1, c2, c3};
}

// keep <=> random word >= p * 2^32 (P(keep) = 1 - p up to 2^-32)
__host__ __device__ inline unsigned drop_threshold(float p) {
  const double t = (double)p * 4294967296.0;
  return t >= 4294967295.0 ? 0xFFFFFFFFu : (t <= 0.0 ? 0u : (unsigned)t);
}

struct XArgs {
  const void *q, *k, *v;
  int n, Nq, Nkv, H, D;
  int64_t q_sb, q_ss, q_sh, k_sb, k_ss, k_sh, v_sb, v_ss, v_sh;
  float scale, drop_p;
  unsigned long long seed, offset;
  void* out;      // [n, Nq, H, D] contiguous
  float* lse;     // [n, H, Nq]
  // backward
  const void* dout;   // [n, Nq, H, D] contiguous
  void *dq, *dk, *dv; // strides of q / k / v
  void *pT, *dsT;     // workspace [n, H, KP, NQP] (bf16 or f32), q contiguous
  int KP, NQP;
};

constexpr int VLD = 144;          // bytes per row of the 64-column LDS slice (128 + 16: rows stay 16-byte aligned)

// 8 contiguous bf16 of a row (or zeros)
__device__ __forceinline__ bf16x8 row8(const bf16* p, bool ok) {
  bf16x8 z;
#pragma unroll
  for (int j = 0; j < 8; ++j) z[j] = (bf16)0.f;
  return ok ? *(const bf16x8*)p : z;
}

// stage rows [0, nrows) x columns [c0, c0 + 64) of a strided bf16 matrix into the LDS slice (zeros outside the matrix)
__device__ __forceinline__ void stage_slice(char* sm, const bf16* base, int64_t row_stride, int row0, int nrows, int rows_valid, int c0, int D) {
  for (int i = threadIdx.x; i < nrows * 8; i += blockDim.x) {
    const int r = i >> 3, ch = i & 7;
    const bool ok = (row0 + r) < rows_valid && (c0 + ch * 8) < D;
    *(bf16x8*)(sm + r * VLD + ch * 16) = row8(base + (int64_t)(row0 + r) * row_stride + c0 + ch * 8, ok);
  }
}

// transposed operand fragment from the slice: lane (c = l & 15, g = l >> 4) receives column (col0 + c) of rows r_lo .. r_lo+3
// (elements 0-3) and r_hi .. r_hi+3 (elements 4-7)
__device__ __forceinline__ bf16x8 slice_frag_t(const char* sm, int r_lo, int r_hi, int col0) {
  const int c = threadIdx.x & 15;
  const int q = c >> 2, p = c & 3;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, sm + (r_lo + q) * VLD + (col0 + 4 * p) * 2));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, sm + (r_hi + q) * VLD + (col0 + 4 * p) * 2));
  bf16x8 o;
  o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3];
  o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = hi[3];
  return o;
}

// S^T tiles of one wave: acc[kt][r] = sum_d K[kt*16 + 4g + r][d] * Q[q][d]   (lane: q = column c, keys = rows 4g + r)
template <int NKT>
__device__ __forceinline__ void scores_t(f32x4 (&acc)[NKT], const bf16* Krows, int64_t k_ss, int Nkv, const bf16* qrow, bool qv, int D, int nkt) {
  const int l = threadIdx.x & 63, c = l & 15, g = l >> 4;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) acc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int d0 = 0; d0 < D; d0 += 32) {
    const int dd = d0 + 8 * g;
    const bf16x8 qf = row8(qrow + dd, qv && dd < D);
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      if (kt < nkt) {
        const int key = kt * 16 + c;
        const bf16x8 kf = row8(Krows + (int64_t)key * k_ss + dd, key < Nkv && dd < D);
        acc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, acc[kt], 0, 0, 0);
      }
    }
  }
}

// keep flags of the 4 keys kt*16 + 4g .. +3 of probability row `row` as inverse-keep-probability factors
__device__ __forceinline__ void keep4(float (&f)[4], int64_t row, int KP, int kt, int g, const XArgs& a, unsigned thr, float inv_keep) {
  const U4 r = philox4x32_10((unsigned long long)(row * (KP / 4) + kt * 4 + g), a.offset, a.seed);
  f[0] = r.x >= thr ? inv_keep : 0.f;
  f[1] = r.y >= thr ? inv_keep : 0.f;
  f[2] = r.z >= thr ? inv_keep : 0.f;
  f[3] = r.w >= thr ? inv_keep : 0.f;
}

// O^T[d][q] (or dQ^T) = X^T . W^T with W^T = the wave's key-major probabilities wf[kp] and X = V (or K) through the LDS slice;
// stores 4 consecutive d per lane.  All waves of the workgroup must call it (barriers inside).
template <int NKT>
__device__ __forceinline__ void apply_keys(char* sm, const bf16x8 (&wf)[NKT / 2], const bf16* Xrows, int64_t x_ss, int Nkv, int nkt, int D,
                                           bf16* orow, bool qv) {
  const int l = threadIdx.x & 63, c = l & 15, g = l >> 4;
  for (int dc = 0; dc < D; dc += 64) {
    __syncthreads();
    stage_slice(sm, Xrows, x_ss, 0, nkt * 16, Nkv, dc, D);
    __syncthreads();
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      if (dc + dt * 16 < D) {
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kp = 0; kp < NKT / 2; ++kp) {
          if (2 * kp < nkt) {
            const bf16x8 xf = slice_frag_t(sm, (2 * kp) * 16 + 4 * g, (2 * kp + 1) * 16 + 4 * g, dt * 16);
            o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, wf[kp], o, 0, 0, 0);     // D[d = 4g + r][q = c]
          }
        }
        const int d = dc + dt * 16 + 4 * g;
        if (qv && d < D) {
          bf16x4 ov;
#pragma unroll
          for (int r = 0; r < 4; ++r) ov[r] = (bf16)o[r];
          *(bf16x4*)(orow + d) = ov;
        }
      }
    }
  }
}

template <int NKT>
__global__ __launch_bounds__(256) void xattn_fwd_kernel(XArgs a) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6, c = l & 15, g = l >> 4;
  const int h = blockIdx.y, b = blockIdx.z;
  const int q = blockIdx.x * 64 + w * 16 + c;
  const bool qv = q < a.Nq;
  const int nkt = a.KP / 16;
  const bf16* qrow = (const bf16*)a.q + b * a.q_sb + h * a.q_sh + (int64_t)q * a.q_ss;
  const bf16* Kr = (const bf16*)a.k + b * a.k_sb + h * a.k_sh;
  const bf16* Vr = (const bf16*)a.v + b * a.v_sb + h * a.v_sh;
  f32x4 acc[NKT];
  scores_t<NKT>(acc, Kr, a.k_ss, a.Nkv, qrow, qv, a.D, nkt);
  float m = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = kt * 16 + 4 * g + r;
      const float s = (kt < nkt && key < a.Nkv) ? acc[kt][r] * a.scale : -INFINITY;
      acc[kt][r] = s;
      m = fmaxf(m, s);
    }
  m = fmaxf(m, __shfl_xor(m, 16, 64));
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float p = __expf(acc[kt][r] - m);      // exp(-inf) = 0 for the padding keys
      acc[kt][r] = p;
      sum += p;
    }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.0f / sum;
  if (qv && g == 0) a.lse[((int64_t)b * a.H + h) * a.Nq + q] = m + __logf(sum);
  const bool drop = a.drop_p > 0.f;
  const unsigned thr = drop_threshold(a.drop_p);
  const float inv_keep = drop ? 1.0f / (1.0f - a.drop_p) : 1.0f;
  const int64_t row = ((int64_t)b * a.H + h) * a.Nq + q;
  bf16x8 pf[NKT / 2];
#pragma unroll
  for (int kp = 0; kp < NKT / 2; ++kp) {
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int kt = 2 * kp + hf;
      float f[4] = {1.f, 1.f, 1.f, 1.f};
      if (drop && kt < nkt) keep4(f, qv ? row : 0, a.KP, kt, g, a, thr, inv_keep);
#pragma unroll
      for (int r = 0; r < 4; ++r) pf[kp][hf * 4 + r] = (bf16)(acc[kt][r] * inv * f[r]);
    }
  }
  bf16* orow = (bf16*)a.out + (((int64_t)b * a.Nq + q) * a.H + h) * a.D;
  apply_keys<NKT>(sm, pf, Vr, a.v_ss, a.Nkv, nkt, a.D, orow, qv);
}

// backward, kernel 1 (per 64-query tile): P, dP, dS; dQ = dS K; leaves P_drop^T and dS^T in the workspace
template <int NKT>
__global__ __launch_bounds__(256) void xattn_bwd_q_kernel(XArgs a) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6, c = l & 15, g = l >> 4;
  const int h = blockIdx.y, b = blockIdx.z;
  const int q = blockIdx.x * 64 + w * 16 + c;
  const bool qv = q < a.Nq;
  const int nkt = a.KP / 16;
  const bf16* qrow = (const bf16*)a.q + b * a.q_sb + h * a.q_sh + (int64_t)q * a.q_ss;
  const bf16* Kr = (const bf16*)a.k + b * a.k_sb + h * a.k_sh;
  const bf16* Vr = (const bf16*)a.v + b * a.v_sb + h * a.v_sh;
  const bf16* dorow = (const bf16*)a.dout + (((int64_t)b * a.Nq + q) * a.H + h) * a.D;
  const bf16* orow = (const bf16*)a.out + (((int64_t)b * a.Nq + q) * a.H + h) * a.D;
  f32x4 acc[NKT], dp[NKT];
  scores_t<NKT>(acc, Kr, a.k_ss, a.Nkv, qrow, qv, a.D, nkt);
  // dP_drop^T[key][q] = V . dO^T, and delta[q] = dO[q] . O[q] on the way (each lane group g covers its own d chunks)
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) dp[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float delta = 0.f;
  for (int d0 = 0; d0 < a.D; d0 += 32) {
    const int dd = d0 + 8 * g;
    const bool ok = qv && dd < a.D;
    const bf16x8 dof = row8(dorow + dd, ok), of = row8(orow + dd, ok);
#pragma unroll
    for (int j = 0; j < 8; ++j) delta = __builtin_fmaf((float)dof[j], (float)of[j], delta);
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      if (kt < nkt) {
        const int key = kt * 16 + c;
        const bf16x8 vf = row8(Vr + (int64_t)key * a.v_ss + dd, key < a.Nkv && dd < a.D);
        dp[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dof, dp[kt], 0, 0, 0);
      }
    }
  }
  delta += __shfl_xor(delta, 16, 64);
  delta += __shfl_xor(delta, 32, 64);
  const float lse = qv ? a.lse[((int64_t)b * a.H + h) * a.Nq + q] : 0.f;
  const bool drop = a.drop_p > 0.f;
  const unsigned thr = drop_threshold(a.drop_p);
  const float inv_keep = drop ? 1.0f / (1.0f - a.drop_p) : 1.0f;
  const int64_t row = ((int64_t)b * a.H + h) * a.Nq + q;
  bf16* pT = (bf16*)a.pT + ((int64_t)b * a.H + h) * a.KP * a.NQP;
  bf16* dsT = (bf16*)a.dsT + ((int64_t)b * a.H + h) * a.KP * a.NQP;
  bf16x8 dsf[NKT / 2];
#pragma unroll
  for (int kp = 0; kp < NKT / 2; ++kp) {
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int kt = 2 * kp + hf;
      float f[4] = {1.f, 1.f, 1.f, 1.f};
      if (drop && kt < nkt) keep4(f, qv ? row : 0, a.KP, kt, g, a, thr, inv_keep);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kt * 16 + 4 * g + r;
        float p = 0.f, ds = 0.f;
        if (qv && kt < nkt && key < a.Nkv) {
          p = __expf(acc[kt][r] * a.scale - lse);
          ds = p * (dp[kt][r] * f[r] - delta) * a.scale;
        }
        const bf16 pb = (bf16)(p * f[r]), dsb = (bf16)ds;
        dsf[kp][hf * 4 + r] = dsb;
        if (kt < nkt) {                                  // every (key < KP, q < NQP) cell is written: kernel 2 reads them all
          pT[(int64_t)key * a.NQP + q] = pb;
          dsT[(int64_t)key * a.NQP + q] = dsb;
        }
      }
    }
  }
  bf16* dqrow = (bf16*)a.dq + b * a.q_sb + h * a.q_sh + (int64_t)q * a.q_ss;
  apply_keys<NKT>(sm, dsf, Kr, a.k_ss, a.Nkv, nkt, a.D, dqrow, qv);
}

// backward, kernel 2 (per 64 keys; wave = 16 keys): dV^T[d][key] = dO^T . P_drop, dK^T[d][key] = Q^T . dS, summed over all queries
// in a fixed order.  Per 64-column slice the queries pass through LDS in blocks of 64 rows (dO and Q images, transposed reads).
__global__ __launch_bounds__(256) void xattn_bwd_kv_kernel(XArgs a) {
  extern __shared__ __attribute__((aligned(16))) char sm[];       // [64][VLD] dO slice | [64][VLD] Q slice
  char* smq = sm + 64 * VLD;
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6, c = l & 15, g = l >> 4;
  const int h = blockIdx.y, b = blockIdx.z;
  const int key0 = blockIdx.x * 64 + w * 16;
  const int key = key0 + c;                       // the lane's key as operand column
  const bool kv = key < a.KP;
  const bf16* Qr = (const bf16*)a.q + b * a.q_sb + h * a.q_sh;
  const bf16* dOr = (const bf16*)a.dout + ((int64_t)b * a.Nq * a.H + h) * a.D;
  const int64_t do_ss = (int64_t)a.H * a.D;
  const bf16* pT = (const bf16*)a.pT + (((int64_t)b * a.H + h) * a.KP + key) * a.NQP;
  const bf16* dsT = (const bf16*)a.dsT + (((int64_t)b * a.H + h) * a.KP + key) * a.NQP;
  bf16* dkrow = (bf16*)a.dk + b * a.k_sb + h * a.k_sh + (int64_t)key * a.k_ss;
  bf16* dvrow = (bf16*)a.dv + b * a.v_sb + h * a.v_sh + (int64_t)key * a.v_ss;
  for (int dc = 0; dc < a.D; dc += 64) {
    f32x4 dv[4], dk[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { dv[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dk[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    for (int q0 = 0; q0 < a.NQP; q0 += 64) {
      __syncthreads();
      stage_slice(sm, dOr, do_ss, q0, 64, a.Nq, dc, a.D);
      stage_slice(smq, Qr, a.q_ss, q0, 64, a.Nq, dc, a.D);
      __syncthreads();
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int qq = q0 + ks * 32 + 8 * g;
        const bf16x8 pf = row8(pT + qq, kv), sf = row8(dsT + qq, kv);       // B operands: [k = query][col = key]
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          if (dc + t * 16 < a.D) {
            const int r0 = ks * 32 + 8 * g;
            dv[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(slice_frag_t(sm, r0, r0 + 4, t * 16), pf, dv[t], 0, 0, 0);
            dk[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(slice_frag_t(smq, r0, r0 + 4, t * 16), sf, dk[t], 0, 0, 0);
          }
        }
      }
    }
    if (key < a.Nkv) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int d = dc + t * 16 + 4 * g;
        if (d < a.D) {
          bf16x4 ov, ok_;
#pragma unroll
          for (int r = 0; r < 4; ++r) { ov[r] = (bf16)dv[t][r]; ok_[r] = (bf16)dk[t][r]; }
          *(bf16x4*)(dvrow + d) = ov;
          *(bf16x4*)(dkrow + d) = ok_;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------- fp32 (parity path)
__device__ __forceinline__ float keep1(int64_t row, int KP, int key, const XArgs& a, unsigned thr, float inv_keep) {
  const U4 r = philox4x32_10((unsigned long long)(row * (KP / 4) + (key >> 2)), a.offset, a.seed);
  const unsigned x = (key & 3) == 0 ? r.x : ((key & 3) == 1 ? r.y : ((key & 3) == 2 ? r.z : r.w));
  return x >= thr ? inv_keep : 0.f;
}

// one wave per (query, head, image): scores in LDS.  mode 0 = forward; mode 1 = backward pass 1 (P_drop, dS rows into the
// workspace [n, H, KP, NQP] (key-major like the bf16 path), dQ)
__global__ __launch_bounds__(64) void xattn_f32_q_kernel(XArgs a, int mode) {
  extern __shared__ float smf[];   // [Nkv] p, [Nkv] ds, [D] q, [D] do
  float* sc = smf;
  float* dsb = smf + a.Nkv;
  float* qs = smf + 2 * a.Nkv;
  float* dos = qs + a.D;
  const int lane = threadIdx.x, D = a.D;
  const int qi = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const float* Q = (const float*)a.q + b * a.q_sb + h * a.q_sh + (int64_t)qi * a.q_ss;
  const float* K = (const float*)a.k + b * a.k_sb + h * a.k_sh;
  const float* V = (const float*)a.v + b * a.v_sb + h * a.v_sh;
  const int64_t orow = (((int64_t)b * a.Nq + qi) * a.H + h) * D;
  const int64_t row = ((int64_t)b * a.H + h) * a.Nq + qi;
  const bool drop = a.drop_p > 0.f;
  const unsigned thr = drop_threshold(a.drop_p);
  const float inv_keep = drop ? 1.0f / (1.0f - a.drop_p) : 1.0f;
  for (int d = lane; d < D; d += 64) {
    qs[d] = Q[d];
    if (mode) dos[d] = ((const float*)a.dout)[orow + d];
  }
  __syncthreads();
  if (mode == 0) {
    float mx = -INFINITY;
    for (int k = lane; k < a.Nkv; k += 64) {
      float s = 0.f;
      const float* kr = K + (int64_t)k * a.k_ss;
      for (int d = 0; d < D; ++d) s += qs[d] * kr[d];
      s *= a.scale;
      sc[k] = s;
      mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int k = lane; k < a.Nkv; k += 64) {
      const float p = expf(sc[k] - mx);
      sc[k] = p;
      sum += p;
    }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    for (int k = lane; k < a.Nkv; k += 64) sc[k] = sc[k] * inv * (drop ? keep1(row, a.KP, k, a, thr, inv_keep) : 1.f);
    __syncthreads();
    float* o = (float*)a.out + orow;
    for (int d = lane; d < D; d += 64) {
      float acc = 0.f;
      for (int k = 0; k < a.Nkv; ++k) acc += sc[k] * V[(int64_t)k * a.v_ss + d];
      o[d] = acc;
    }
    if (lane == 0) a.lse[row] = mx + logf(sum);
    return;
  }
  const float lse = a.lse[row];
  float dl = 0.f;
  for (int d = lane; d < D; d += 64) dl += dos[d] * ((const float*)a.out)[orow + d];
  dl = wave_sum(dl);
  float* pT = (float*)a.pT + ((int64_t)b * a.H + h) * a.KP * a.NQP;
  float* dsT = (float*)a.dsT + ((int64_t)b * a.H + h) * a.KP * a.NQP;
  for (int k = lane; k < a.Nkv; k += 64) {
    float s = 0.f, dpv = 0.f;
    const float* kr = K + (int64_t)k * a.k_ss;
    const float* vr = V + (int64_t)k * a.v_ss;
    for (int d = 0; d < D; ++d) { s += qs[d] * kr[d]; dpv += dos[d] * vr[d]; }
    const float f = drop ? keep1(row, a.KP, k, a, thr, inv_keep) : 1.f;
    const float p = expf(s * a.scale - lse);
    const float ds = p * (dpv * f - dl) * a.scale;
    dsb[k] = ds;
    pT[(int64_t)k * a.NQP + qi] = p * f;
    dsT[(int64_t)k * a.NQP + qi] = ds;
  }
  __syncthreads();
  float* dQ = (float*)a.dq + b * a.q_sb + h * a.q_sh + (int64_t)qi * a.q_ss;
  for (int d = lane; d < D; d += 64) {
    float acc = 0.f;
    for (int k = 0; k < a.Nkv; ++k) acc += dsb[k] * K[(int64_t)k * a.k_ss + d];
    dQ[d] = acc;
  }
}

// one wave per (key, head, image): dV[key] = sum_q P_drop[q][key] dO[q], dK[key] = sum_q dS[q][key] Q[q] in query order
__global__ __launch_bounds__(64) void xattn_f32_kv_kernel(XArgs a) {
  const int lane = threadIdx.x, D = a.D;
  const int k = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const float* pT = (const float*)a.pT + (((int64_t)b * a.H + h) * a.KP + k) * a.NQP;
  const float* dsT = (const float*)a.dsT + (((int64_t)b * a.H + h) * a.KP + k) * a.NQP;
  const float* Q = (const float*)a.q + b * a.q_sb + h * a.q_sh;
  const float* dO = (const float*)a.dout + ((int64_t)b * a.Nq * a.H + h) * D;
  float* dK = (float*)a.dk + b * a.k_sb + h * a.k_sh + (int64_t)k * a.k_ss;
  float* dV = (float*)a.dv + b * a.v_sb + h * a.v_sh + (int64_t)k * a.v_ss;
  for (int d = lane; d < D; d += 64) {
    float av = 0.f, ak = 0.f;
    for (int q = 0; q < a.Nq; ++q) {
      av += pT[q] * dO[(int64_t)q * a.H * D + d];
      ak += dsT[q] * Q[(int64_t)q * a.q_ss + d];
    }
    dV[d] = av;
    dK[d] = ak;
  }
}

// ---------------------------------------------------------------------------------------------- element-wise dropout
template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(const T* x, int64_t n, float p, unsigned long long seed, unsigned long long offset, T* y) {
  const unsigned thr = drop_threshold(p);
  const float inv_keep = 1.0f / (1.0f - p);
  const int64_t nc = (n + 3) / 4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nc; i += (int64_t)gridDim.x * 256) {
    const U4 r = philox4x32_10((unsigned long long)i, offset, seed);
    const unsigned rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t e = i * 4 + k;
      if (e < n) y[e] = from_f32<T>(rr[k] >= thr ? to_f32(x[e]) * inv_keep : 0.f);
    }
  }
}

__global__ __launch_bounds__(256) void dropout_mask_kernel(int64_t n, float p, unsigned long long seed, unsigned long long offset, unsigned char* mask) {
  const unsigned thr = drop_threshold(p);
  const int64_t nc = (n + 3) / 4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nc; i += (int64_t)gridDim.x * 256) {
    const U4 r = philox4x32_10((unsigned long long)i, offset, seed);
    const unsigned rr[4] = {r.x, r.y, r.z, r.w};
    for (int k = 0; k < 4; ++k)
      if (i * 4 + k < n) mask[i * 4 + k] = rr[k] >= thr ? 1 : 0;
  }
}

int pick_nkt(int KP) { const int t = KP / 16; return t <= 4 ? 4 : (t <= 8 ? 8 : (t <= 16 ? 16 : (t <= 32 ? 32 : 64))); }

int check(int dtype, int n, int Nq, int Nkv, int H, int D, float p) {
  if (n < 0 || Nq < 0 || Nkv <= 0 || H <= 0 || D <= 0 || !(p >= 0.f && p < 1.f)) return MM_ERR_ARG;
  if (dtype != MM_BF16 && dtype != MM_F32) return MM_ERR_UNSUPPORTED;
  if (dtype == MM_BF16 && ((D & 7) || D > 512)) return MM_ERR_UNSUPPORTED;
  if (Nkv > 1024) return MM_ERR_UNSUPPORTED;      // the score row of a query lives in registers: 64 MFMA tiles of 16 keys at most
  if (dtype == MM_F32 && (int64_t)(2 * Nkv + 2 * D) * 4 > 60000) return MM_ERR_UNSUPPORTED;
  return MM_OK;
}

}  // namespace

#define XA_STRIDES int64_t q_sb, int64_t q_ss, int64_t q_sh, int64_t k_sb, int64_t k_ss, int64_t k_sh, int64_t v_sb, int64_t v_ss, int64_t v_sh

extern "C" int mm_xattn_ws_bytes(int dtype, int n, int Nq, int Nkv, int H, int64_t* bytes) {
  if (!bytes || n < 0 || Nq < 0 || Nkv <= 0 || H <= 0) return MM_ERR_ARG;
  const int64_t KP = (Nkv + 31) / 32 * 32, NQP = (Nq + 63) / 64 * 64;
  *bytes = 2 * (int64_t)n * H * KP * NQP * mm_elem_size(dtype);
  return MM_OK;
}

extern "C" int mm_xattn_fwd(int dtype, const void* q, const void* k, const void* v, int n, int Nq, int Nkv, int H, int D, XA_STRIDES,
                            float scale, float drop_p, int64_t seed, int64_t offset, void* out, float* lse, void* stream) {
  const int rc = check(dtype, n, Nq, Nkv, H, D, drop_p);
  if (rc != MM_OK) return rc;
  if (n == 0 || Nq == 0) return MM_OK;
  if (!q || !k || !v || !out || !lse) return MM_ERR_ARG;
  XArgs a{};
  a.q = q; a.k = k; a.v = v; a.n = n; a.Nq = Nq; a.Nkv = Nkv; a.H = H; a.D = D;
  a.q_sb = q_sb; a.q_ss = q_ss; a.q_sh = q_sh; a.k_sb = k_sb; a.k_ss = k_ss; a.k_sh = k_sh; a.v_sb = v_sb; a.v_ss = v_ss; a.v_sh = v_sh;
  a.scale = scale; a.drop_p = drop_p; a.seed = (unsigned long long)seed; a.offset = (unsigned long long)offset;
  a.out = out; a.lse = lse;
  a.KP = (Nkv + 31) / 32 * 32;
  a.NQP = (Nq + 63) / 64 * 64;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MM_F32) {
    hipLaunchKernelGGL(xattn_f32_q_kernel, dim3(Nq, H, n), dim3(64), (size_t)(2 * Nkv + 2 * D) * 4, st, a, 0);
    MM_CHECK_LAUNCH();
    return MM_OK;
  }
  if ((q_ss | q_sh | q_sb | k_ss | k_sh | k_sb | v_ss | v_sh | v_sb) & 7) return MM_ERR_ALIGN;
  if (!mm_aligned16(q) || !mm_aligned16(k) || !mm_aligned16(v) || (((uintptr_t)out) & 7)) return MM_ERR_ALIGN;
  const dim3 grid((unsigned)(a.NQP / 64), (unsigned)H, (unsigned)n), block(256);
  const size_t lds = (size_t)a.KP * VLD;
#define XA_LAUNCH(KERN, T)                                                                                          \
  do {                                                                                                              \
    auto kfn = KERN<T>;                                                                                             \
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);              \
    hipLaunchKernelGGL(kfn, grid, block, lds, st, a);                                                               \
  } while (0)
  switch (pick_nkt(a.KP)) {
    case 4: XA_LAUNCH(xattn_fwd_kernel, 4); break;
    case 8: XA_LAUNCH(xattn_fwd_kernel, 8); break;
    case 16: XA_LAUNCH(xattn_fwd_kernel, 16); break;
    case 32: XA_LAUNCH(xattn_fwd_kernel, 32); break;
    default: XA_LAUNCH(xattn_fwd_kernel, 64); break;
  }
  MM_CHECK_LAUNCH();
  return MM_OK;
}

extern "C" int mm_xattn_bwd(int dtype, const void* q, const void* k, const void* v, const void* out, const void* dout, const float* lse,
                            int n, int Nq, int Nkv, int H, int D, XA_STRIDES, float scale, float drop_p, int64_t seed, int64_t offset,
                            void* dq, void* dk, void* dv, void* ws, int64_t ws_bytes, void* stream) {
  const int rc = check(dtype, n, Nq, Nkv, H, D, drop_p);
  if (rc != MM_OK) return rc;
  if (n == 0 || Nq == 0) return MM_OK;
  if (!q || !k || !v || !out || !dout || !lse || !dq || !dk || !dv || !ws) return MM_ERR_ARG;
  int64_t need = 0;
  mm_xattn_ws_bytes(dtype, n, Nq, Nkv, H, &need);
  if (ws_bytes < need) return MM_ERR_ARG;
  XArgs a{};
  a.q = q; a.k = k; a.v = v; a.n = n; a.Nq = Nq; a.Nkv = Nkv; a.H = H; a.D = D;
  a.q_sb = q_sb; a.q_ss = q_ss; a.q_sh = q_sh; a.k_sb = k_sb; a.k_ss = k_ss; a.k_sh = k_sh; a.v_sb = v_sb; a.v_ss = v_ss; a.v_sh = v_sh;
  a.scale = scale; a.drop_p = drop_p; a.seed = (unsigned long long)seed; a.offset = (unsigned long long)offset;
  a.out = (void*)out; a.lse = (float*)lse; a.dout = dout; a.dq = dq; a.dk = dk; a.dv = dv;
  a.KP = (Nkv + 31) / 32 * 32;
  a.NQP = (Nq + 63) / 64 * 64;
  a.pT = ws;
  a.dsT = (char*)ws + need / 2;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MM_F32) {
    hipLaunchKernelGGL(xattn_f32_q_kernel, dim3(Nq, H, n), dim3(64), (size_t)(2 * Nkv + 2 * D) * 4, st, a, 1);
    hipLaunchKernelGGL(xattn_f32_kv_kernel, dim3(Nkv, H, n), dim3(64), 0, st, a);
    MM_CHECK_LAUNCH();
    return MM_OK;
  }
  if ((q_ss | q_sh | q_sb | k_ss | k_sh | k_sb | v_ss | v_sh | v_sb) & 7) return MM_ERR_ALIGN;
  if (!mm_aligned16(q) || !mm_aligned16(k) || !mm_aligned16(v) || !mm_aligned16(dout) || !mm_aligned16(out) || !mm_aligned16(ws)) return MM_ERR_ALIGN;
  if ((((uintptr_t)dq) | ((uintptr_t)dk) | ((uintptr_t)dv)) & 7) return MM_ERR_ALIGN;
  {
    const dim3 grid((unsigned)(a.NQP / 64), (unsigned)H, (unsigned)n), block(256);
    const size_t lds = (size_t)a.KP * VLD;
    switch (pick_nkt(a.KP)) {
      case 4: XA_LAUNCH(xattn_bwd_q_kernel, 4); break;
      case 8: XA_LAUNCH(xattn_bwd_q_kernel, 8); break;
      case 16: XA_LAUNCH(xattn_bwd_q_kernel, 16); break;
      case 32: XA_LAUNCH(xattn_bwd_q_kernel, 32); break;
      default: XA_LAUNCH(xattn_bwd_q_kernel, 64); break;
    }
  }
#undef XA_LAUNCH
  hipLaunchKernelGGL(xattn_bwd_kv_kernel, dim3((unsigned)((Nkv + 63) / 64), (unsigned)H, (unsigned)n), dim3(256), (size_t)2 * 64 * VLD, st, a);
  MM_CHECK_LAUNCH();
  return MM_OK;
}

extern "C" int mm_dropout(int dtype, const void* x, int64_t n, float p, int64_t seed, int64_t offset, void* y, void* stream) {
  if (n < 0 || !(p >= 0.f && p < 1.f)) return MM_ERR_ARG;
  if (n == 0) return MM_OK;
  if (!x || !y) return MM_ERR_ARG;
  const int64_t nc = (n + 3) / 4;
  const unsigned nb = (unsigned)((nc + 255) / 256 > 4096 ? 4096 : (nc + 255) / 256);
  if (dtype == MM_BF16)
    hipLaunchKernelGGL(dropout_kernel<bf16>, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, n, p, (unsigned long long)seed, (unsigned long long)offset, (bf16*)y);
  else if (dtype == MM_F32)
    hipLaunchKernelGGL(dropout_kernel<float>, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const float*)x, n, p, (unsigned long long)seed, (unsigned long long)offset, (float*)y);
  else
    return MM_ERR_UNSUPPORTED;
  MM_CHECK_LAUNCH();
  return MM_OK;
}

extern "C" int mm_dropout_mask(int64_t seed, int64_t offset, int64_t n, float p, void* mask_u8, void* stream) {
  if (n < 0 || !(p >= 0.f && p < 1.f)) return MM_ERR_ARG;
  if (n == 0) return MM_OK;
  if (!mask_u8) return MM_ERR_ARG;
  const int64_t nc = (n + 3) / 4;
  const unsigned nb = (unsigned)((nc + 255) / 256 > 4096 ? 4096 : (nc + 255) / 256);
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, n, p, (unsigned long long)seed, (unsigned long long)offset, (unsigned char*)mask_u8);
  MM_CHECK_LAUNCH();
  return MM_OK;
}
