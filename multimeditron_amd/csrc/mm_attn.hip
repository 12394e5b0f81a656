// Flash-style attention for the Llama/Qwen2 decoder (causal, GQA, key-padding mask) and the CLIP ViT
// (non-causal), forward + backward, on v_mfma_f32_32x32x16_bf16.
//
// Layout idea (CDNA4): every product is issued "key-major" so that the quantity a row-softmax reduces
// over lives in a lane's OWN registers and the query index lives on the lane:
//   S^T[key][q] = K . Q^T          (A = K fragment from LDS, B = Q fragment held in registers)
//   O^T[d][q]  += V^T[d][key] . P^T (A = V^T by ds_read_b64_tr_b16 from a row-major V tile,
//                                    B = P^T taken straight from the S^T accumulators: no LDS, no shuffles)
// so max/sum over keys are 31 in-register ops + one cross-half exchange, the online-softmax rescale is one
// scalar per lane, and the backward dQ kernel has the same shape (dQ^T += K^T . dS^T).
// dK/dV use a second kernel, one workgroup per 128 keys, that keeps dK^T/dV^T of 32 keys per wave in
// accumulators while sweeping the query tiles (no atomics, deterministic).
// K tiles sit in LDS as [d/8][key][8] (ds_read_b128, conflict free); V / transposed operands as
// row-major tiles with a 32-byte XOR swizzle chosen so the transposed reads are conflict free.
//
// fp32 variants (parity path only) are plain wave-per-query-row kernels.
#include <stdlib.h>

#include "mm_common.h"
#include <type_traits>
#include <string.h>

namespace {

struct AttnArgs {
  const void *q, *k, *v;
  int B, Sq, Skv, Hq, Hkv;
  int64_t q_sb, q_ss, q_sh, k_sb, k_ss, k_sh, v_sb, v_ss, v_sh;
  const int64_t* kmask;
  int causal;
  float scale;
  void* out;
  float* lse;
  // backward
  const void* dout;
  const float* delta;
  void *dq, *dk, *dv;
  int diag;   // timing experiments on attn_fwd128q_kernel (results wrong by design): 1 no DMA, 2 no softmax VALU, 4 no QK^T MFMAs, 8 no PV MFMAs, 16 no LDS fragment reads
  int prio;   // s_setprio policy of the out-of-phase kernels: 0 none, 1 around every MFMA segment, 2 static for waves 4-7, 3 static for waves 0-3
};

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

template <int D> __device__ __forceinline__ int v_swz(int row) {
  if constexpr (D == 128) return (row & 3) << 1;
  else return ((row >> 1) & 1) << 1;
}

// ---- tile staging (256 threads) ---------------------------------------------------------------------------
// KC image: [D/8][ROWS][8]; loads 16 B per lane, 8 rows x 8 chunks (full 128-B lines) per wave instruction
template <int D, int ROWS>
struct KcStage {
  static constexpr int NI = ROWS * (D / 8) / 256;
  u32x4 r[NI];
  __device__ __forceinline__ void load(const bf16* base, int64_t row_stride, int row0, int nrows_valid) {
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    constexpr int HALVES = D / 64;  // 8-chunk groups per row
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int u = w * NI + i;
      const int rg = u / HALVES, dh = u % HALVES;
      const int row = rg * 8 + (l & 7), dc = dh * 8 + (l >> 3);
      u32x4 v = {0u, 0u, 0u, 0u};
      if (row0 + row < nrows_valid) v = *(const u32x4*)(base + (int64_t)(row0 + row) * row_stride + dc * 8);
      r[i] = v;
    }
  }
  __device__ __forceinline__ void store(char* tile) const {
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    constexpr int HALVES = D / 64;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int u = w * NI + i;
      const int rg = u / HALVES, dh = u % HALVES;
      const int row = rg * 8 + (l & 7), dc = dh * 8 + (l >> 3);
      *(u32x4*)(tile + (dc * ROWS + row) * 16) = r[i];
    }
  }
  // same registers into the row-major swizzled (KS) image; 2-way write conflicts only
  __device__ __forceinline__ void store_rowmajor(char* tile) const {
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    constexpr int HALVES = D / 64;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int u = w * NI + i;
      const int rg = u / HALVES, dh = u % HALVES;
      const int row = rg * 8 + (l & 7), c16 = dh * 8 + (l >> 3);
      *(u32x4*)(tile + row * (D * 2) + (((c16 >> 1) ^ v_swz<D>(row)) * 32) + (c16 & 1) * 16) = r[i];
    }
  }
};

// KS image: [ROWS][D] row-major with 32-byte-chunk XOR swizzle
template <int D, int ROWS>
struct KsStage {
  static constexpr int CPR = D / 8;  // 16-B chunks per row
  static constexpr int NI = ROWS * CPR / 256;
  u32x4 r[NI];
  __device__ __forceinline__ void load(const bf16* base, int64_t row_stride, int row0, int nrows_valid) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int idx = i * 256 + threadIdx.x;
      const int row = idx / CPR, c16 = idx % CPR;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (row0 + row < nrows_valid) v = *(const u32x4*)(base + (int64_t)(row0 + row) * row_stride + c16 * 8);
      r[i] = v;
    }
  }
  __device__ __forceinline__ void store(char* tile) const {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int idx = i * 256 + threadIdx.x;
      const int row = idx / CPR, c16 = idx % CPR;
      *(u32x4*)(tile + row * (D * 2) + (((c16 >> 1) ^ v_swz<D>(row)) * 32) + (c16 & 1) * 16) = r[i];
    }
  }
  // re-tile the same registers (loaded with KsStage::load's mapping) is not possible for KC; see callers
};

// A fragment (32 rows x 16 k) from a KC image: lane row = l&31, k = 8*(l>>5)+j
template <int ROWS>
__device__ __forceinline__ bf16x8 kc_frag(const char* tile, int rblk, int kstep) {
  const int l = threadIdx.x & 63;
  return *(const bf16x8*)(tile + ((kstep * 2 + (l >> 5)) * ROWS + rblk * 32 + (l & 31)) * 16);
}

// A fragment of the TRANSPOSE of a row-major KS image [rows = k index][D]: A[row = d][k], for d-block db
// (32 wide) and k rows kbase .. kbase+15 in the accumulator-compatible order
//   element j of lane half h  <->  k row  kbase + 8*(j>>2) + 4*h + (j&3)
template <int D>
__device__ __forceinline__ bf16x8 ks_frag_t(const char* tile, int db, int kbase) {
  const int l = threadIdx.x & 63;
  const int h = l >> 5, gi = (l >> 4) & 1, i = l & 15, q = i >> 2, p = i & 3;
  const int row1 = kbase + 4 * h + q;
  const int c32 = db * 2 + gi;
  const int sw = ((c32 ^ v_swz<D>(row1)) * 32) + p * 8;
  bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, tile + row1 * (D * 2) + sw));
  bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, tile + (row1 + 8) * (D * 2) + sw));
  bf16x8 o;
  o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3];
  o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = hi[3];
  return o;
}

// B fragment (k = d, n = row) straight from global: lane n = l&31, d = dstep*16 + 8*(l>>5) .. +7
__device__ __forceinline__ bf16x8 row_frag_global(const bf16* row_ptr_or_null, int dstep) {
  const int l = threadIdx.x & 63;
  bf16x8 z;
#pragma unroll
  for (int j = 0; j < 8; ++j) z[j] = (bf16)0.f;
  if (!row_ptr_or_null) return z;
  return *(const bf16x8*)(row_ptr_or_null + dstep * 16 + 8 * (l >> 5));
}

__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ============================================================================================================
// forward
// ============================================================================================================
template <int D>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnArgs a) {
  constexpr int BKV = 64, NDS = D / 16, NDB = D / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Kt = smem;                 // KC image [D/8][64][8]
  char* Vt = smem + BKV * D * 2;   // KS image [64][D]
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6, h = l >> 5;
  const int b = blockIdx.z, hq = blockIdx.y, hkv = hq / (a.Hq / a.Hkv);
  const int q0 = blockIdx.x * 128 + w * 32;
  const int qi = q0 + (l & 31);
  const int shift = a.Skv - a.Sq;  // causal: query i sees keys <= i + shift
  const bf16* Q = (const bf16*)a.q + b * a.q_sb + hq * a.q_sh;
  const bf16* K = (const bf16*)a.k + b * a.k_sb + hkv * a.k_sh;
  const bf16* V = (const bf16*)a.v + b * a.v_sb + hkv * a.v_sh;

  bf16x8 qf[NDS];
  {
    const bf16* qrow = qi < a.Sq ? Q + (int64_t)qi * a.q_ss : nullptr;
#pragma unroll
    for (int ds = 0; ds < NDS; ++ds) qf[ds] = row_frag_global(qrow, ds);
  }
  f32x16 o_acc[NDB];
#pragma unroll
  for (int i = 0; i < NDB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o_acc[i][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float sc = a.scale * LOG2E;

  int ntiles = (a.Skv + BKV - 1) / BKV;
  if (a.causal) {
    const int qmax = min(a.Sq - 1, (int)blockIdx.x * 128 + 127) + shift;
    ntiles = min(ntiles, qmax / BKV + 1);
    if (qmax < 0) ntiles = 0;
  }
  KcStage<D, BKV> ks;
  KsStage<D, BKV> vs;
  if (ntiles > 0) {
    ks.load(K, a.k_ss, 0, a.Skv);
    vs.load(V, a.v_ss, 0, a.Skv);
  }
  for (int t = 0; t < ntiles; ++t) {
    const int kv0 = t * BKV;
    __syncthreads();
    ks.store(Kt);
    vs.store(Vt);
    __syncthreads();
    if (t + 1 < ntiles) {
      ks.load(K, a.k_ss, kv0 + BKV, a.Skv);
      vs.load(V, a.v_ss, kv0 + BKV, a.Skv);
    }
    // key validity for this tile: one key per lane -> 64-bit ballot
    bool kvalid = (kv0 + l) < a.Skv;
    if (kvalid && a.kmask) kvalid = a.kmask[(int64_t)b * a.Skv + kv0 + l] != 0;
    const unsigned long long kbits = __ballot(kvalid);
    const bool need_mask = (kbits != ~0ull) || (a.causal && (kv0 + BKV - 1) > (q0 + shift));

    f32x16 s_acc[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s_acc[kb][r] = 0.f;
#pragma unroll
      for (int ds = 0; ds < NDS; ++ds)
        s_acc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kc_frag<BKV>(Kt, kb, ds), qf[ds], s_acc[kb], 0, 0, 0);
    }
    // online softmax in base 2: p = exp2(s*sc - m); the scale is folded into one FMA per element and the row max is
    // taken on the raw scores (sc > 0), so the unmasked path costs max + fma + v_exp + add (+ half a cvt) per element
    float mx = -INFINITY;
    if (need_mask) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kl = kb * 32 + acc_row(r, h);
          bool ok = (kbits >> kl) & 1ull;
          if (a.causal) ok = ok && (kv0 + kl) <= (qi + shift);
          const float tv = ok ? s_acc[kb][r] : -INFINITY;
          s_acc[kb][r] = tv;
          mx = fmaxf(mx, tv);
        }
    } else {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; r += 2) mx = fmaxf(mx, fmaxf(s_acc[kb][r], s_acc[kb][r + 1]));
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * sc;
    const float m_new = fmaxf(m_run, mx);
    const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_safe);
    float rs = 0.f;
    bf16x8 pf[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s_acc[kb][r], sc, -m_safe));
        rs += p;
        pf[kb][r >> 3][r & 7] = (bf16)p;
      }
    l_run = l_run * alpha + rs;
    m_run = m_new;
#pragma unroll
    for (int db = 0; db < NDB; ++db) {
#pragma unroll
      for (int r = 0; r < 16; ++r) o_acc[db][r] *= alpha;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int s = 0; s < 2; ++s)
          o_acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ks_frag_t<D>(Vt, db, kb * 32 + s * 16), pf[kb][s], o_acc[db], 0, 0, 0);
    }
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
  if (qi < a.Sq) {
    bf16* orow = (bf16*)a.out + (((int64_t)b * a.Sq + qi) * a.Hq + hq) * D;
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16)(o_acc[db][rg * 4 + e] * inv);
        *(bf16x4*)(orow + db * 32 + 8 * rg + 4 * h) = o;
      }
    if (h == 0) a.lse[((int64_t)b * a.Hq + hq) * a.Sq + qi] = l_tot > 0.f ? (m_run + log2f(l_tot)) * LN2 : INFINITY;
  }
}

// delta[b,h,q] = sum_d dO*O
// delta[b,hq,q] = sum_d O[b,q,hq,d] * dO[b,q,hq,d] (the softmax-backward row term).  16-byte loads: a row of D elements is
// read by D/VN lanes and reduced by shuffles (was: one wave per row with 2-byte loads, 3.2 TB/s on 134 MB).
template <typename T>
__global__ void attn_delta_kernel(const T* o, const T* dout, int B, int Sq, int Hq, int D, float* delta) {
  constexpr int VN = Vec16<T>::N;
  const int lpr = D / VN;                                  // lanes per row: 16 (bf16 D=128), 8 (bf16 D=64), ...
  const int64_t total = (int64_t)B * Sq * Hq;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool vec = (D % VN) == 0 && lpr <= 64 && (lpr & (lpr - 1)) == 0;
  if (!vec) {                                              // odd head widths (fp32 parity path): one wave per row
    const int64_t wave = tid >> 6;
    const int lane = threadIdx.x & 63;
    if (wave >= total) return;
    float s = 0.f;
    for (int d = lane; d < D; d += 64) s += to_f32(o[wave * D + d]) * to_f32(dout[wave * D + d]);
    s = wave_sum(s);
    if (lane == 0) {
      const int hq = (int)(wave % Hq);
      const int64_t bs = wave / Hq;
      delta[((bs / Sq) * Hq + hq) * Sq + bs % Sq] = s;
    }
    return;
  }
  const int64_t row = tid / lpr;
  const int c = (int)(tid % lpr);
  float s = 0.f;
  if (row < total) {
    const Vec16<T> a = *(const Vec16<T>*)(o + row * D + c * VN), d = *(const Vec16<T>*)(dout + row * D + c * VN);
#pragma unroll
    for (int k = 0; k < VN; ++k) s += a.get(k) * d.get(k);
  }
  for (int off = lpr >> 1; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if (row < total && c == 0) {
    const int hq = (int)(row % Hq);
    const int64_t bs = row / Hq;
    delta[((bs / Sq) * Hq + hq) * Sq + bs % Sq] = s;
  }
}

// ============================================================================================================
// backward: dQ   (same shape as forward; dQ^T[d][q] += K^T[d][key] . dS^T[key][q])
// ============================================================================================================
template <int D>
__global__ __launch_bounds__(256, (D == 128 ? 1 : 2)) void attn_bwd_dq_kernel(AttnArgs a) {
  constexpr int BKV = 64, NDS = D / 16, NDB = D / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Kc = smem;                      // K, KC image
  char* Vc = smem + BKV * D * 2;        // V, KC image
  char* Kr = smem + 2 * BKV * D * 2;    // K, row-major swizzled (for K^T fragments)
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6, h = l >> 5;
  const int b = blockIdx.z, hq = blockIdx.y, hkv = hq / (a.Hq / a.Hkv);
  const int q0 = blockIdx.x * 128 + w * 32;
  const int qi = q0 + (l & 31);
  const int shift = a.Skv - a.Sq;
  const bf16* Q = (const bf16*)a.q + b * a.q_sb + hq * a.q_sh;
  const bf16* K = (const bf16*)a.k + b * a.k_sb + hkv * a.k_sh;
  const bf16* V = (const bf16*)a.v + b * a.v_sb + hkv * a.v_sh;
  const bf16* dO = (const bf16*)a.dout + ((int64_t)b * a.Sq * a.Hq + hq) * D;

  bf16x8 qf[NDS], dof[NDS];
  {
    const bf16* qrow = qi < a.Sq ? Q + (int64_t)qi * a.q_ss : nullptr;
    const bf16* drow = qi < a.Sq ? dO + (int64_t)qi * a.Hq * D : nullptr;
#pragma unroll
    for (int ds = 0; ds < NDS; ++ds) {
      qf[ds] = row_frag_global(qrow, ds);
      dof[ds] = row_frag_global(drow, ds);
    }
  }
  const float sc = a.scale * LOG2E;
  float lse2 = INFINITY, dlt = 0.f;
  if (qi < a.Sq) {
    lse2 = a.lse[((int64_t)b * a.Hq + hq) * a.Sq + qi] * LOG2E;
    dlt = a.delta[((int64_t)b * a.Hq + hq) * a.Sq + qi];
  }
  f32x16 dq_acc[NDB];
#pragma unroll
  for (int i = 0; i < NDB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq_acc[i][r] = 0.f;

  int ntiles = (a.Skv + BKV - 1) / BKV;
  if (a.causal) {
    const int qmax = min(a.Sq - 1, (int)blockIdx.x * 128 + 127) + shift;
    ntiles = min(ntiles, qmax / BKV + 1);
    if (qmax < 0) ntiles = 0;
  }
  KcStage<D, BKV> kcs, vcs;
  if (ntiles > 0) {
    kcs.load(K, a.k_ss, 0, a.Skv);
    vcs.load(V, a.v_ss, 0, a.Skv);
  }
  for (int t = 0; t < ntiles; ++t) {
    const int kv0 = t * BKV;
    __syncthreads();
    kcs.store(Kc);
    vcs.store(Vc);
    kcs.store_rowmajor(Kr);
    __syncthreads();
    if (t + 1 < ntiles) {
      kcs.load(K, a.k_ss, kv0 + BKV, a.Skv);
      vcs.load(V, a.v_ss, kv0 + BKV, a.Skv);
    }
    bool kvalid = (kv0 + l) < a.Skv;
    if (kvalid && a.kmask) kvalid = a.kmask[(int64_t)b * a.Skv + kv0 + l] != 0;
    const unsigned long long kbits = __ballot(kvalid);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x16 s_acc, dp_acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s_acc[r] = 0.f; dp_acc[r] = 0.f; }
#pragma unroll
      for (int ds = 0; ds < NDS; ++ds) {
        s_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kc_frag<BKV>(Kc, kb, ds), qf[ds], s_acc, 0, 0, 0);
        dp_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kc_frag<BKV>(Vc, kb, ds), dof[ds], dp_acc, 0, 0, 0);
      }
      bf16x8 dsf[2];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kl = kb * 32 + acc_row(r, h);
        bool ok = (kbits >> kl) & 1ull;
        if (a.causal) ok = ok && (kv0 + kl) <= (qi + shift);
        const float p = ok ? exp2f(s_acc[r] * sc - lse2) : 0.f;
        const float dsv = p * (dp_acc[r] - dlt) * a.scale;
        dsf[r >> 3][r & 7] = (bf16)dsv;
      }
#pragma unroll
      for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int s = 0; s < 2; ++s)
          dq_acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ks_frag_t<D>(Kr, db, kb * 32 + s * 16), dsf[s], dq_acc[db], 0, 0, 0);
    }
  }
  if (qi < a.Sq) {
    bf16* drow = (bf16*)a.dq + b * a.q_sb + hq * a.q_sh + (int64_t)qi * a.q_ss;
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16)dq_acc[db][rg * 4 + e];
        *(bf16x4*)(drow + db * 32 + 8 * rg + 4 * h) = o;
      }
  }
}

// ============================================================================================================
// backward: dK, dV.  One workgroup = 128 keys of one (batch, kv head); wave w owns keys w*32..w*32+31 and keeps
// dK^T, dV^T [d][key] in accumulators while sweeping (q head in group) x (32-row query tiles).
//   S[q][key] = Q K^T, dP[q][key] = dO V^T     (A = Q / dO tile from LDS (KC image), B = K / V fragments in regs)
//   dV^T += dO^T . P, dK^T += Q^T . dS            (A = transposed reads of the row-major Q / dO images, B = accumulators)
// ============================================================================================================
template <int D>
__global__ __launch_bounds__(256, 1) void attn_bwd_dkv_kernel(AttnArgs a) {
  constexpr int BQ = 32, NDS = D / 16, NDB = D / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Qc = smem;                    // Q tile KC image [D/8][32][8]
  char* Qr = smem + BQ * D * 2;       // Q tile row-major swizzled
  char* Oc = smem + 2 * BQ * D * 2;   // dO tile KC
  char* Or = smem + 3 * BQ * D * 2;   // dO tile row-major
  float* rowc = (float*)(smem + 4 * BQ * D * 2);  // [2][32]: lse*log2e, delta
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6, h = l >> 5;
  const int b = blockIdx.z, hkv = blockIdx.y;
  const int G = a.Hq / a.Hkv;
  const int k0 = blockIdx.x * 128 + w * 32;
  const int ki = k0 + (l & 31);
  const int shift = a.Skv - a.Sq;
  const bf16* K = (const bf16*)a.k + b * a.k_sb + hkv * a.k_sh;
  const bf16* V = (const bf16*)a.v + b * a.v_sb + hkv * a.v_sh;
  bf16x8 kf[NDS], vf[NDS];
  {
    const bf16* krow = ki < a.Skv ? K + (int64_t)ki * a.k_ss : nullptr;
    const bf16* vrow = ki < a.Skv ? V + (int64_t)ki * a.v_ss : nullptr;
#pragma unroll
    for (int ds = 0; ds < NDS; ++ds) {
      kf[ds] = row_frag_global(krow, ds);
      vf[ds] = row_frag_global(vrow, ds);
    }
  }
  bool kvalid = ki < a.Skv;
  if (kvalid && a.kmask) kvalid = a.kmask[(int64_t)b * a.Skv + ki] != 0;
  const float sc = a.scale * LOG2E;
  f32x16 dk_acc[NDB], dv_acc[NDB];
#pragma unroll
  for (int i = 0; i < NDB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk_acc[i][r] = 0.f; dv_acc[i][r] = 0.f; }

  // first query tile that can see any key of this workgroup
  int qt0 = 0;
  if (a.causal) qt0 = max(0, (int)blockIdx.x * 128 - shift) / BQ;
  const int nqt = (a.Sq + BQ - 1) / BQ;
  const int per_head = max(0, nqt - qt0);
  const int niter = per_head * G;                  // (q head in group) x (query tile), flattened for prefetching
  const int64_t do_ss = (int64_t)a.Hq * D;
  KcStage<D, BQ> qcs, ocs;                         // each tile is written to both LDS images
  float rc = 0.f;
  auto prefetch = [&](int it) {
    const int g = it / per_head, qb = (qt0 + it % per_head) * BQ;
    const int hq = hkv * G + g;
    const bf16* Q = (const bf16*)a.q + b * a.q_sb + hq * a.q_sh;
    const bf16* dO = (const bf16*)a.dout + ((int64_t)b * a.Sq * a.Hq + hq) * D;
    qcs.load(Q, a.q_ss, qb, a.Sq);
    ocs.load(dO, do_ss, qb, a.Sq);
    if (threadIdx.x < 64) {
      const int qq = qb + (threadIdx.x & 31);
      const int64_t ro = ((int64_t)b * a.Hq + hq) * a.Sq + qq;
      if (threadIdx.x < 32) rc = qq < a.Sq ? a.lse[ro] * LOG2E : INFINITY;
      else rc = qq < a.Sq ? a.delta[ro] : 0.f;
    }
  };
  if (niter > 0) prefetch(0);
  for (int it = 0; it < niter; ++it) {
    const int qb = (qt0 + it % per_head) * BQ;
    __syncthreads();  // previous tile fully consumed
    qcs.store(Qc);
    qcs.store_rowmajor(Qr);
    ocs.store(Oc);
    ocs.store_rowmajor(Or);
    if (threadIdx.x < 64) rowc[threadIdx.x] = rc;
    __syncthreads();
    if (it + 1 < niter) prefetch(it + 1);
    f32x16 s_acc, dp_acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s_acc[r] = 0.f; dp_acc[r] = 0.f; }
#pragma unroll
    for (int ds = 0; ds < NDS; ++ds) {
      s_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kc_frag<BQ>(Qc, 0, ds), kf[ds], s_acc, 0, 0, 0);
      dp_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kc_frag<BQ>(Oc, 0, ds), vf[ds], dp_acc, 0, 0, 0);
    }
    bf16x8 pf[2], dsf[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ql = acc_row(r, h);
      bool ok = kvalid;
      if (a.causal) ok = ok && ki <= (qb + ql + shift);
#ifdef MM_ATTN_DIAG_NOSOFTMAX
      pf[r >> 3][r & 7] = (bf16)s_acc[r];
      dsf[r >> 3][r & 7] = (bf16)dp_acc[r];
#else
      const float p = ok ? exp2f(s_acc[r] * sc - rowc[ql]) : 0.f;
      const float dsv = p * (dp_acc[r] - rowc[32 + ql]) * a.scale;
      pf[r >> 3][r & 7] = (bf16)p;
      dsf[r >> 3][r & 7] = (bf16)dsv;
#endif
    }
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        dv_acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ks_frag_t<D>(Or, db, s * 16), pf[s], dv_acc[db], 0, 0, 0);
        dk_acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ks_frag_t<D>(Qr, db, s * 16), dsf[s], dk_acc[db], 0, 0, 0);
      }
  }
  if (ki < a.Skv) {
    bf16* dkrow = (bf16*)a.dk + b * a.k_sb + hkv * a.k_sh + (int64_t)ki * a.k_ss;
    bf16* dvrow = (bf16*)a.dv + b * a.v_sb + hkv * a.v_sh + (int64_t)ki * a.v_ss;
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        bf16x4 ok_, ov_;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          ok_[e] = (bf16)dk_acc[db][rg * 4 + e];
          ov_[e] = (bf16)dv_acc[db][rg * 4 + e];
        }
        *(bf16x4*)(dkrow + db * 32 + 8 * rg + 4 * h) = ok_;
        *(bf16x4*)(dvrow + db * 32 + 8 * rg + 4 * h) = ov_;
      }
  }
}

// ============================================================================================================
// fp32 parity kernels: one wave per (batch, head, query row); scores staged in LDS
// ============================================================================================================
__global__ __launch_bounds__(64) void attn_fwd_f32_kernel(AttnArgs a, int D) {
  extern __shared__ float sm[];  // [Skv] scores, then [D] q
  float* sc = sm;
  float* qs = sm + a.Skv;
  const int lane = threadIdx.x;
  const int qi = blockIdx.x, hq = blockIdx.y, b = blockIdx.z, hkv = hq / (a.Hq / a.Hkv);
  const int shift = a.Skv - a.Sq;
  const float* Q = (const float*)a.q + b * a.q_sb + hq * a.q_sh + (int64_t)qi * a.q_ss;
  const float* K = (const float*)a.k + b * a.k_sb + hkv * a.k_sh;
  const float* V = (const float*)a.v + b * a.v_sb + hkv * a.v_sh;
  for (int d = lane; d < D; d += 64) qs[d] = Q[d];
  __syncthreads();
  float mx = -INFINITY;
  for (int k = lane; k < a.Skv; k += 64) {
    bool ok = !(a.causal && k > qi + shift);
    if (ok && a.kmask) ok = a.kmask[(int64_t)b * a.Skv + k] != 0;
    float s = -INFINITY;
    if (ok) {
      s = 0.f;
      const float* kr = K + (int64_t)k * a.k_ss;
      for (int d = 0; d < D; ++d) s += qs[d] * kr[d];
      s *= a.scale;
    }
    sc[k] = s;
    mx = fmaxf(mx, s);
  }
  mx = wave_max(mx);
  const float ms = mx == -INFINITY ? 0.f : mx;
  float sum = 0.f;
  for (int k = lane; k < a.Skv; k += 64) {
    const float p = expf(sc[k] - ms);
    sc[k] = p;
    sum += p;
  }
  sum = wave_sum(sum);
  __syncthreads();
  const float inv = sum > 0.f ? 1.f / sum : 0.f;
  float* o = (float*)a.out + (((int64_t)b * a.Sq + qi) * a.Hq + hq) * D;
  for (int d = lane; d < D; d += 64) {
    float acc = 0.f;
    for (int k = 0; k < a.Skv; ++k) acc += sc[k] * V[(int64_t)k * a.v_ss + d];
    o[d] = acc * inv;
  }
  if (lane == 0) a.lse[((int64_t)b * a.Hq + hq) * a.Sq + qi] = sum > 0.f ? ms + logf(sum) : INFINITY;
}

__global__ __launch_bounds__(64) void attn_bwd_f32_kernel(AttnArgs a, int D) {
  extern __shared__ float sm[];  // [Skv] p, [Skv] ds, [D] q, [D] do
  float* pp = sm;
  float* dsb = sm + a.Skv;
  float* qs = sm + 2 * a.Skv;
  float* dos = qs + D;
  const int lane = threadIdx.x;
  const int qi = blockIdx.x, hq = blockIdx.y, b = blockIdx.z, hkv = hq / (a.Hq / a.Hkv);
  const int shift = a.Skv - a.Sq;
  const float* Q = (const float*)a.q + b * a.q_sb + hq * a.q_sh + (int64_t)qi * a.q_ss;
  const float* K = (const float*)a.k + b * a.k_sb + hkv * a.k_sh;
  const float* V = (const float*)a.v + b * a.v_sb + hkv * a.v_sh;
  const float* dO = (const float*)a.dout + (((int64_t)b * a.Sq + qi) * a.Hq + hq) * D;
  const float lse = a.lse[((int64_t)b * a.Hq + hq) * a.Sq + qi];
  const float dlt = a.delta[((int64_t)b * a.Hq + hq) * a.Sq + qi];
  for (int d = lane; d < D; d += 64) { qs[d] = Q[d]; dos[d] = dO[d]; }
  __syncthreads();
  for (int k = lane; k < a.Skv; k += 64) {
    bool ok = !(a.causal && k > qi + shift);
    if (ok && a.kmask) ok = a.kmask[(int64_t)b * a.Skv + k] != 0;
    float p = 0.f, dsv = 0.f;
    if (ok) {
      float s = 0.f, dp = 0.f;
      const float* kr = K + (int64_t)k * a.k_ss;
      const float* vr = V + (int64_t)k * a.v_ss;
      for (int d = 0; d < D; ++d) { s += qs[d] * kr[d]; dp += dos[d] * vr[d]; }
      p = expf(s * a.scale - lse);
      dsv = p * (dp - dlt) * a.scale;
    }
    pp[k] = p;
    dsb[k] = dsv;
  }
  __syncthreads();
  float* dQ = (float*)a.dq + b * a.q_sb + hq * a.q_sh + (int64_t)qi * a.q_ss;
  float* dK = (float*)a.dk + b * a.k_sb + hkv * a.k_sh;
  float* dV = (float*)a.dv + b * a.v_sb + hkv * a.v_sh;
  for (int d = lane; d < D; d += 64) {
    float acc = 0.f;
    const float qd = qs[d], dod = dos[d];
    for (int k = 0; k < a.Skv; ++k) {
      const float dsv = dsb[k];
      if (dsv != 0.f || pp[k] != 0.f) {
        acc += dsv * K[(int64_t)k * a.k_ss + d];
        atomicAdd(dK + (int64_t)k * a.k_ss + d, dsv * qd);
        atomicAdd(dV + (int64_t)k * a.v_ss + d, pp[k] * dod);
      }
    }
    dQ[d] = acc;
  }
}

// ============================================================================================================
// D = 128 fast path.  8 waves per workgroup (256 query rows / 256 keys), tiles streamed by LDS-DMA into a 2-deep ring
// with ONE barrier per tile, 2 waves per SIMD so one wave's softmax VALU overlaps the other's MFMAs.
// Every tile uses ONE LDS image that serves both row fragments (ds_read_b128) and transposed fragments
// (ds_read_b64_tr_b16): plain 256-byte rows, 16-byte chunk index XOR ((row&3)<<2 | (row>>2)&3); the DMA destination is
// linear in lane order, so the XOR is applied to the per-lane SOURCE chunk.
// ============================================================================================================
__device__ __forceinline__ int imgb_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ int imgb_off(int row, int ch) { return 256 * row + 16 * (ch ^ imgb_swz(row)); }

// DMA ROWS rows x 256 B (row stride in elements) starting at matrix row `row0`; 8 waves, ROWS/32 pieces (4 rows) each
template <int ROWS, int NW = 8, int NWV_UNUSED = 8>
__device__ __forceinline__ void imgb_dma(unsigned tile_lds, const SRsrc& rs, int64_t stride, int row0) {
  const int l = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int PPW = ROWS / (4 * NW);
  static_assert(PPW >= 1, "tile too small for this many waves");
  if (w >= NW) return;      // NW < waves in the workgroup: only the first NW waves issue (staggers SIMD partners)
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int pc = w * PPW + i;
    const int row = pc * 4 + (l >> 4);
    const int ch = (l & 15) ^ imgb_swz(row);
    const unsigned voff = (unsigned)(((int64_t)(row0 + row) * stride) * 2 + ch * 16);
    lds_dma16(rs, voff, tile_lds + pc * 1024);
  }
}
// same, issued by the NW waves of one TEAM: wt = this wave's index inside its team
template <int ROWS, int NW>
__device__ __forceinline__ void imgb_dma_team(unsigned tile_lds, const SRsrc& rs, int64_t stride, int row0, int wt) {
  const int l = threadIdx.x & 63;
  constexpr int PPW = ROWS / (4 * NW);
  static_assert(PPW >= 1, "tile too small for this many waves");
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int pc = wt * PPW + i;
    const int row = pc * 4 + (l >> 4);
    const int ch = (l & 15) ^ imgb_swz(row);
    const unsigned voff = (unsigned)(((int64_t)(row0 + row) * stride) * 2 + ch * 16);
    lds_dma16(rs, voff, tile_lds + pc * 1024);
  }
}
// row fragment (32 rows x 16 k): lane row = r0 + (l&31), k = kstep*16 + 8*(l>>5) .. +7
__device__ __forceinline__ bf16x8 imgb_rowfrag(const char* tile, int r0, int kstep) {
  const int l = threadIdx.x & 63;
  return *(const bf16x8*)(tile + imgb_off(r0 + (l & 31), kstep * 2 + (l >> 5)));
}
// fragment of the transposed tile: A[row = column db*32 + (l&31)][k = tile row], k rows kbase..kbase+15 in the
// accumulator-compatible order (element j of lane half h <-> tile row kbase + 8*(j>>2) + 4*h + (j&3))
__device__ __forceinline__ bf16x8 imgb_tfrag(const char* tile, int db, int kbase) {
  const int l = threadIdx.x & 63;
  const int h = l >> 5, gi = (l >> 4) & 1, i = l & 15, q = i >> 2, p = i & 3;
  const int row1 = kbase + 4 * h + q;
  const int ch = db * 4 + gi * 2 + (p >> 1);
  bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, tile + imgb_off(row1, ch) + 8 * (p & 1)));
  bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, tile + imgb_off(row1 + 8, ch) + 8 * (p & 1)));
  bf16x8 o;
  o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3];
  o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = hi[3];
  return o;
}
__device__ __forceinline__ SRsrc rows_rsrc(const bf16* base, int nrows, int64_t stride) {
  return make_srsrc(base, nrows > 0 ? ((int64_t)(nrows - 1) * stride + 128) * 2 : 0);
}

// single-instruction VALU helpers for the softmax of the D = 128 kernels (rocprofv3: the forward issued 10.8 VALU
// instructions per MFMA and was VALU-issue bound).  A plain fmaxf on MFMA results gets a canonicalising v_max in front of it
// (48 v_max + 8 v_max3 for a 32-value row maximum); __shfl_xor(.., 32) is a ds_bpermute round trip through the LDS.
__device__ __forceinline__ float vmax3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// max / sum of x with its partner lane (l ^ 32): one v_permlane32_swap
__device__ __forceinline__ float swap32_max(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return vmax3(__uint_as_float(r[0]), __uint_as_float(r[1]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float swap32_sum(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// maximum of the 32 scores a lane holds for its query row (two 32x32 accumulator tiles)
__device__ __forceinline__ float rowmax32(const f32x16& a, const f32x16& b) {
  float m0 = vmax3(a[0], a[1], a[2]), m1 = vmax3(b[0], b[1], b[2]);
#pragma unroll
  for (int r = 3; r < 15; r += 2) { m0 = vmax3(m0, a[r], a[r + 1]); m1 = vmax3(m1, b[r], b[r + 1]); }
  return vmax3(m0, m1, vmax3(a[15], b[15], b[15]));
}
// (A LAZY rescale -- guide T13: bring O and l to a new running maximum only when some row's maximum grew by more than a
// threshold, a wave-uniform branch -- was built and measured: no gain on this kernel (686 vs 676-689 TF/s: it is not VALU
// bound) and it costs the bit-exact causality / padding invariance of the forward, because the
// decision is taken per WAVE and so depends on the other 31 rows of the wave.  Not used.)

// Work order of the D = 128 kernels (1-D grid).  Under a causal mask a block's work grows with its position, and the
// hardware hands out workgroups in index order: with the heaviest blocks LAST the launch ends on a few long workgroups
// (62 vs 40 tile-times for 8 blocks x 128 heads on 512 slots).  Heaviest first fixes that.  Workgroup ids that are
// congruent mod 8 share an XCD (and its L2): the query heads of one key/value head are given consecutive turns on ONE
// XCD so its K/V rows are fetched into that L2 once per group.
__device__ __forceinline__ void attn_work_item(int id, int nblk, int B, int Hq, int Hkv, bool heavy_last, int& blk, int& b,
                                               int& hq) {
  const int nbh = B * Hq, G = Hq / Hkv;
  const int x = id / nbh, r = id % nbh;
  blk = heavy_last ? nblk - 1 - x : x;
  if (((B * Hkv) & 7) == 0) {
    const int xcd = r & 7, c = r >> 3;
    const int slot = xcd + 8 * (c / G), g = c % G;     // slot = b * Hkv + hkv
    b = slot / Hkv;
    hq = (slot % Hkv) * G + g;
  } else {
    b = r / Hq;
    hq = r % Hq;
  }
}

// NWV waves per workgroup (8: 256 query rows, one workgroup per CU; 4: 128 rows, two independent workgroups per CU)
template <int INW, int NWV>
__global__ __launch_bounds__(NWV * 64, 2) void attn_fwd128_kernel(AttnArgs a) {
  constexpr int QB = NWV * 32;
  constexpr int BKV = 64, NDS = 8, NDB = 4, TILE = BKV * 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 stages][K 16 KiB | V 16 KiB]
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6, h = l >> 5;
  int qblk, b, hq;
  attn_work_item(blockIdx.x, (a.Sq + QB - 1) / QB, a.B, a.Hq, a.Hkv, a.causal != 0, qblk, b, hq);
  const int hkv = hq / (a.Hq / a.Hkv);
  const int q0 = qblk * QB + w * 32;
  const int qi = q0 + (l & 31);
  const int shift = a.Skv - a.Sq;
  const bf16* Q = (const bf16*)a.q + b * a.q_sb + hq * a.q_sh;
  const SRsrc rk = rows_rsrc((const bf16*)a.k + b * a.k_sb + hkv * a.k_sh, a.Skv, a.k_ss);
  const SRsrc rv = rows_rsrc((const bf16*)a.v + b * a.v_sb + hkv * a.v_sh, a.Skv, a.v_ss);
  const unsigned lds0 = (unsigned)(uintptr_t)LDS_PTR(char, smem);

  bf16x8 qf[NDS];
  {
    const bf16* qrow = qi < a.Sq ? Q + (int64_t)qi * a.q_ss : nullptr;
#pragma unroll
    for (int ds = 0; ds < NDS; ++ds) qf[ds] = row_frag_global(qrow, ds);
  }
  f32x16 o_acc[NDB];
#pragma unroll
  for (int i = 0; i < NDB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o_acc[i][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float sc = a.scale * LOG2E;

  int ntiles = (a.Skv + BKV - 1) / BKV;
  if (a.causal) {
    const int qmax = min(a.Sq - 1, qblk * QB + QB - 1) + shift;
    ntiles = qmax < 0 ? 0 : min(ntiles, qmax / BKV + 1);
  }
  // per-lane source offsets of this wave's DMA pieces, computed ONCE (they were recomputed for every tile: 176 VALU
  // instructions per issuing wave and tile); a tile only adds its wave-uniform row offset
  constexpr int PPW = BKV / (4 * INW);
  const int wu = __builtin_amdgcn_readfirstlane(w);
  unsigned kofs[PPW], vofs[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int row = (wu * PPW + i) * 4 + (l >> 4);
    const int ch = (l & 15) ^ imgb_swz(row);
    kofs[i] = (unsigned)((int64_t)row * a.k_ss * 2 + ch * 16);
    vofs[i] = (unsigned)((int64_t)row * a.v_ss * 2 + ch * 16);
  }
  auto issue = [&](int t) {
    if (wu >= INW) return;                    // only the first INW waves issue (staggers the two waves of every SIMD)
    const unsigned st = lds0 + (unsigned)((t & 1) * 2 * TILE) + (unsigned)(wu * PPW) * 1024u;
    const unsigned tk = (unsigned)((int64_t)t * BKV * a.k_ss * 2), tv = (unsigned)((int64_t)t * BKV * a.v_ss * 2);
#pragma unroll
    for (int i = 0; i < PPW; ++i) lds_dma16(rk, kofs[i] + tk, st + i * 1024);
#pragma unroll
    for (int i = 0; i < PPW; ++i) lds_dma16(rv, vofs[i] + tv, st + TILE + i * 1024);
  };
  if (ntiles > 0) issue(0);
  for (int t = 0; t < ntiles; ++t) {
    const int kv0 = t * BKV;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#ifndef MM_ATTN_DIAG_FWD_NODMA
    if (t + 1 < ntiles) issue(t + 1);
#endif
    // wave-uniform skip: this wave's rows are all beyond Sq, or the whole tile lies above its causal diagonal
    if (q0 >= a.Sq || (a.causal && kv0 > q0 + 31 + shift)) continue;
    const char* Kt = smem + (t & 1) * 2 * TILE;
    const char* Vt = Kt + TILE;
    bool kvalid = (kv0 + l) < a.Skv;
    if (kvalid && a.kmask) kvalid = a.kmask[(int64_t)b * a.Skv + kv0 + l] != 0;
    const unsigned long long kbits = __ballot(kvalid);
    const bool need_mask = (kbits != ~0ull) || (a.causal && (kv0 + BKV - 1) > (q0 + shift));

    f32x16 s_acc[2];
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s_acc[kb][r] = 0.f;
#ifdef MM_ATTN_DIAG_FWD_NOQK
      s_acc[kb] = o_acc[kb];
#else
#pragma unroll
      for (int ds = 0; ds < NDS; ++ds)
        s_acc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(imgb_rowfrag(Kt, kb * 32, ds), qf[ds], s_acc[kb], 0, 0, 0);
#endif
    }
    __builtin_amdgcn_s_setprio(0);
    float mx = -INFINITY;
#ifdef MM_ATTN_DIAG_FWD_NOSM
    bf16x8 pf[2][2];
    float alpha = 1.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) pf[kb][r >> 3][r & 7] = (bf16)s_acc[kb][r];
    if (need_mask) l_run += 1.f;
#else
    if (need_mask) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kl = kb * 32 + acc_row(r, h);
          bool ok = (kbits >> kl) & 1ull;
          if (a.causal) ok = ok && (kv0 + kl) <= (qi + shift);
          s_acc[kb][r] = ok ? s_acc[kb][r] : -INFINITY;
        }
    }
    mx = swap32_max(rowmax32(s_acc[0], s_acc[1])) * sc;
    const float m_new = fmaxf(m_run, mx);
    const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_safe);
    float rs = 0.f;
    bf16x8 pf[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s_acc[kb][r], sc, -m_safe));
        rs += p;
        pf[kb][r >> 3][r & 7] = (bf16)p;
      }
    l_run = l_run * alpha + rs;
    m_run = m_new;
#endif
#ifdef MM_ATTN_DIAG_FWD_NOPV
#pragma unroll
    for (int r = 0; r < 16; ++r) o_acc[r & 3][r] = o_acc[r & 3][r] * alpha + (float)pf[r & 1][(r >> 3) & 1][r & 7] + (float)pf[(r + 1) & 1][(r >> 3) & 1][r & 7];
#else
#pragma unroll
    for (int db = 0; db < NDB; ++db) {
#pragma unroll
      for (int r = 0; r < 16; ++r) o_acc[db][r] *= alpha;
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int s = 0; s < 2; ++s)
          o_acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(imgb_tfrag(Vt, db, kb * 32 + s * 16), pf[kb][s], o_acc[db], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
#endif
  }
  const float l_tot = swap32_sum(l_run);
  const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
  if (qi < a.Sq) {
    bf16* orow = (bf16*)a.out + (((int64_t)b * a.Sq + qi) * a.Hq + hq) * 128;
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16)(o_acc[db][rg * 4 + e] * inv);
        *(bf16x4*)(orow + db * 32 + 8 * rg + 4 * h) = o;
      }
    if (h == 0) a.lse[((int64_t)b * a.Hq + hq) * a.Sq + qi] = l_tot > 0.f ? (m_run + log2f(l_tot)) * LN2 : INFINITY;
  }
}

// ---- D = 128 forward, 8 waves, fragments PREFETCHED --------------------------------------------------------------------
// rocprofv3 on attn_fwd128_kernel: MFMA busy 30 %, and halving its VALU count changed nothing.  The ISA showed why: at 246+
// VGPRs hipcc keeps ONE register quad for the K fragments, so the QK^T phase is `ds_read_b128 -> s_waitcnt lgkmcnt(0) ->
// v_mfma` sixteen times over: every MFMA (32 cycles) waits out a whole LDS round trip (>100 cycles), and the wave's partner on
// the SIMD is in the same phase.  This kernel (same tile, same arithmetic and rounding points, bit-identical results) fixes
// the two causes:
//   * LDS image (a) of the guide (8-row x 64-byte subtiles): every fragment address is a per-lane base + a CONSTANT, so the
//     16 K row reads hang off 2 base registers and the 32 V^T transposed reads off 2 more (was 8 + 8 registers and an
//     address add per read), and the DMA source offsets are 2 lane patterns + scalars (was 8 registers);
//   * the freed registers hold a 3-deep fragment ring: fragment i+2 is requested before MFMA i is issued, and
//     `sched_barrier(0)` between the (read, MFMA) chunks keeps hipcc from collapsing the ring again.
__device__ __forceinline__ int imga_off(int row, int ch) {      // byte offset of 16-byte chunk ch of row `row` in a 256-B-row tile
  return 2048 * (row >> 3) + 512 * (ch >> 2) + 64 * (row & 7) + 16 * ((ch & 3) ^ ((row >> 2) & 3));
}
struct ImgaBases { const char* kr[2]; const char* vt[2]; };     // per-lane bases inside stage 0 (K tile at 0, V tile at `tile_bytes`)
__device__ __forceinline__ ImgaBases imga_bases(const char* smem, int tile_bytes) {
  const int l = threadIdx.x & 63, h = l >> 5, r = l & 31;
  ImgaBases B;
#pragma unroll
  for (int a = 0; a < 2; ++a)          // row fragment of k-step ds (32x32x16 A operand): row r, chunk 2*ds + h; a = ds & 1
    B.kr[a] = smem + 2048 * (r >> 3) + 64 * (r & 7) + 16 * ((2 * a + h) ^ ((r >> 2) & 3));
  const int gi = (l >> 4) & 1, i = l & 15, q = i >> 2, p = i & 3;
  const int cl = gi * 2 + (p >> 1);    // chunk inside the 64-byte d block; rows kbase + 4h + q (first read) and + 8 (second)
  B.vt[0] = smem + tile_bytes + 64 * (4 * h + q) + 16 * (cl ^ h) + 8 * (p & 1);
  B.vt[1] = smem + tile_bytes + 2048 + 64 * (4 * h + q) + 16 * (cl ^ (h ^ 2)) + 8 * (p & 1);
  return B;
}
__device__ __forceinline__ bf16x8 imga_kfrag(const ImgaBases& B, int stage_off, int kb, int ds) {
  return *(const bf16x8*)(B.kr[ds & 1] + stage_off + 8192 * kb + 512 * (ds >> 1));
}
__device__ __forceinline__ bf16x8 imga_vfrag(const ImgaBases& B, int stage_off, int db, int kbase16) {   // kbase16 = first key / 16
  const int off = stage_off + 4096 * kbase16 + 512 * db;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, B.vt[0] + off));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, B.vt[1] + off));
  bf16x8 o;
  o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3];
  o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = hi[3];
  return o;
}

// K and V tiles (64 rows x 256 B each) of key tile t into stage `st` (LDS byte address), image (a), issued by waves 0-3:
// wave w moves pieces 4w .. 4w+3 of each tile; lk / lv = the two per-lane source patterns (see attn_fwd128p_kernel)
__device__ __forceinline__ void imga_lane_patterns(unsigned (&lk)[2], unsigned (&lv)[2], int64_t k_ss, int64_t v_ss) {
  const int l = threadIdx.x & 63;
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    const int rl = (l >> 2) & 7, cl = 4 * (l >> 5) + ((l & 3) ^ ((s2 << 1) | ((l >> 4) & 1)));
    lk[s2] = (unsigned)((int64_t)rl * k_ss * 2 + cl * 16);
    lv[s2] = (unsigned)((int64_t)rl * v_ss * 2 + cl * 16);
  }
}
// the same with NW issuing waves (4 or 8): wave w moves pieces w*PPW .. w*PPW + PPW - 1 of the K tile and of the V tile
template <int NW>
__device__ __forceinline__ void imga_issue_kv_n(int w, unsigned st, int row0, const SRsrc& rk, const SRsrc& rv, const unsigned (&lk)[2],
                                                const unsigned (&lv)[2], int64_t k_ss, int64_t v_ss, int tile_bytes) {
  constexpr int PPW = 16 / NW;
  if (w >= NW) return;
  st += (unsigned)(w * PPW) * 1024u;
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int pc = w * PPW + i, prow = row0 + 8 * (pc >> 1);          // wave-uniform
    lds_dma16(rk, lk[(pc >> 1) & 1] + (unsigned)((int64_t)prow * k_ss * 2 + (pc & 1) * 128), st + i * 1024);
  }
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int pc = w * PPW + i, prow = row0 + 8 * (pc >> 1);
    lds_dma16(rv, lv[(pc >> 1) & 1] + (unsigned)((int64_t)prow * v_ss * 2 + (pc & 1) * 128), st + tile_bytes + i * 1024);
  }
}
__device__ __forceinline__ void imga_issue_kv(int w, unsigned st, int row0, const SRsrc& rk, const SRsrc& rv, const unsigned (&lk)[2],
                                              const unsigned (&lv)[2], int64_t k_ss, int64_t v_ss, int tile_bytes) {
  if (w >= 4) return;
  st += (unsigned)w * 4096u;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int prow = row0 + 8 * (2 * w + (i >> 1));                 // first row of the piece (wave-uniform)
    lds_dma16(rk, lk[i >> 1] + (unsigned)((int64_t)prow * k_ss * 2 + (i & 1) * 128), st + i * 1024);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int prow = row0 + 8 * (2 * w + (i >> 1));
    lds_dma16(rv, lv[i >> 1] + (unsigned)((int64_t)prow * v_ss * 2 + (i & 1) * 128), st + tile_bytes + i * 1024);
  }
}

__global__ __launch_bounds__(512, 2) void attn_fwd128p_kernel(AttnArgs a) {
  constexpr int QB = 256, BKV = 64, NDS = 8, NDB = 4, TILE = BKV * 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [3 stages][K 16 KiB | V 16 KiB]
  const int l = threadIdx.x & 63, h = l >> 5;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int qblk, b, hq;
  attn_work_item(blockIdx.x, (a.Sq + QB - 1) / QB, a.B, a.Hq, a.Hkv, a.causal != 0, qblk, b, hq);
  const int hkv = hq / (a.Hq / a.Hkv);
  const int q0 = qblk * QB + w * 32;
  const int qi = q0 + (l & 31);
  const int shift = a.Skv - a.Sq;
  const bf16* Q = (const bf16*)a.q + b * a.q_sb + hq * a.q_sh;
  const SRsrc rk = rows_rsrc((const bf16*)a.k + b * a.k_sb + hkv * a.k_sh, a.Skv, a.k_ss);
  const SRsrc rv = rows_rsrc((const bf16*)a.v + b * a.v_sb + hkv * a.v_sh, a.Skv, a.v_ss);
  const unsigned lds0 = (unsigned)(uintptr_t)LDS_PTR(char, smem);
  const ImgaBases bases = imga_bases(smem, TILE);

  bf16x8 qf[NDS];
  {
    const bf16* qrow = qi < a.Sq ? Q + (int64_t)qi * a.q_ss : nullptr;
#pragma unroll
    for (int ds = 0; ds < NDS; ++ds) qf[ds] = row_frag_global(qrow, ds);
  }
  f32x16 o_acc[NDB];
#pragma unroll
  for (int i = 0; i < NDB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o_acc[i][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float sc = a.scale * LOG2E;

  int ntiles = (a.Skv + BKV - 1) / BKV;
  if (a.causal) {
    const int qmax = min(a.Sq - 1, qblk * QB + QB - 1) + shift;
    ntiles = qmax < 0 ? 0 : min(ntiles, qmax / BKV + 1);
  }
  // DMA: waves 0-3 issue (staggers the two waves of every SIMD); wave w moves pieces 4w .. 4w+3 of each 16-piece tile.  Piece
  // pc = 8 rows x 128 B: lane l reads row 8*(pc>>1) + ((l>>2)&7), chunk 8*(pc&1) + 4*(l>>5) + ((l&3) ^ (2*((pc>>1)&1) | (l>>4)&1)),
  // i.e. one of TWO lane patterns (pieces 4w, 4w+1 -> s = 0; 4w+2, 4w+3 -> s = 1) plus a wave-uniform offset.
  unsigned lk[2], lv[2];
  imga_lane_patterns(lk, lv, a.k_ss, a.v_ss);
  // 3-stage ring: the tiles of step t+2 are requested while t is computed.  With 2 stages (request t+1 after the barrier of t,
  // vmcnt(0) at the next barrier) a tile had ONE 32-MFMA step (~0.9 us) to arrive, about the loaded-chip LDS-DMA latency:
  // rocprofv3 still showed the waves parked 35 % of their cycles after the fragment prefetch went in.  An issuing wave has 8
  // DMA pieces per tile in flight, so `vmcnt(8)` retires tile t and leaves t+1 under way.
  constexpr int NST = 3;
  auto issue = [&](int t) {
    imga_issue_kv(w, lds0 + (unsigned)((t % NST) * 2 * TILE), t * BKV, rk, rv, lk, lv, a.k_ss, a.v_ss, TILE);
  };
  if (ntiles > 0) issue(0);
  if (ntiles > 1) issue(1);
  for (int t = 0; t < ntiles; ++t) {
    const int kv0 = t * BKV;
    if (t + 1 < ntiles) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t + 2 < ntiles) issue(t + 2);
    // wave-uniform skip: this wave's rows are all beyond Sq, or the whole tile lies above its causal diagonal
    if (q0 >= a.Sq || (a.causal && kv0 > q0 + 31 + shift)) continue;
    const int so = (t % NST) * 2 * TILE;
    bool kvalid = (kv0 + l) < a.Skv;
    if (kvalid && a.kmask) kvalid = a.kmask[(int64_t)b * a.Skv + kv0 + l] != 0;
    const unsigned long long kbits = __ballot(kvalid);
    const bool need_mask = (kbits != ~0ull) || (a.causal && (kv0 + BKV - 1) > (q0 + shift));

    // ---- S^T = K . Q^T: 16 MFMAs, K fragment i+2 requested before MFMA i
    f32x16 s_acc[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) s_acc[kb][r] = 0.f;
    constexpr int RD = 4;                                            // fragment ring depth: RD - 1 fragments in flight
    bf16x8 kf[RD];
#pragma unroll
    for (int j = 0; j < RD - 1; ++j) kf[j] = imga_kfrag(bases, so, 0, j);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (i + RD - 1 < 16) kf[(i + RD - 1) % RD] = imga_kfrag(bases, so, (i + RD - 1) >> 3, (i + RD - 1) & 7);
      s_acc[i >> 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[i % RD], qf[i & 7], s_acc[i >> 3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_setprio(0);
    bf16x8 vf[RD];                                                   // first V^T fragments travel under the softmax
#pragma unroll
    for (int j = 0; j < RD - 1; ++j) vf[j] = imga_vfrag(bases, so, j >> 2, j & 3);
    if (need_mask) {
      // key kl = c + 4h with c a compile-time constant per accumulator register: shift the key bits and the causal limit by
      // the lane's 4h ONCE, so every test is against an immediate (32 hoisted per-register key indices cost 14 spilled VGPRs)
      const unsigned long long kb2 = kbits >> (4 * h);
      const int dlim = a.causal ? (qi + shift - kv0 - 4 * h) : 0x7fffffff;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = kb * 32 + (r & 3) + 8 * (r >> 2);
          const bool ok = ((kb2 >> c) & 1ull) && c <= dlim;
          s_acc[kb][r] = ok ? s_acc[kb][r] : -INFINITY;
        }
    }
    const float mx = swap32_max(rowmax32(s_acc[0], s_acc[1])) * sc;
    const float m_new = fmaxf(m_run, mx);
    const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_safe);
    float rs = 0.f;
    bf16x8 pf[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s_acc[kb][r], sc, -m_safe));
        rs += p;
        pf[kb][r >> 3][r & 7] = (bf16)p;
      }
    l_run = l_run * alpha + rs;
    m_run = m_new;
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) o_acc[db][r] *= alpha;
    __builtin_amdgcn_sched_barrier(0);
    // ---- O^T += V^T . P^T: 16 MFMAs (d block db, key step ks = 16 keys), V^T fragment i+2 requested before MFMA i
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 16; ++i) {                                   // i = db * 4 + ks
      if (i + RD - 1 < 16) vf[(i + RD - 1) % RD] = imga_vfrag(bases, so, (i + RD - 1) >> 2, (i + RD - 1) & 3);
      o_acc[i >> 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[i % RD], pf[(i >> 1) & 1][i & 1], o_acc[i >> 2], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_setprio(0);
  }
  const float l_tot = swap32_sum(l_run);
  const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
  if (qi < a.Sq) {
    bf16* orow = (bf16*)a.out + (((int64_t)b * a.Sq + qi) * a.Hq + hq) * 128;
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16)(o_acc[db][rg * 4 + e] * inv);
        *(bf16x4*)(orow + db * 32 + 8 * rg + 4 * h) = o;
      }
    if (h == 0) a.lse[((int64_t)b * a.Hq + hq) * a.Sq + qi] = l_tot > 0.f ? (m_run + log2f(l_tot)) * LN2 : INFINITY;
  }
}

// ---- D = 128 forward, 8 waves, the two waves of a SIMD OUT OF PHASE ---------------------------------------------------------
// attn_fwd128p_kernel keeps the matrix pipe 33 % busy: its eight waves leave every barrier together, so the two waves of a SIMD
// are in the same segment at the same time -- QK^T beside QK^T (matrix pipe shared), softmax beside softmax (VALU shared,
// matrix pipe idle), PV beside PV.  Per key tile and wave the softmax is ~1470 VALU cycles against 2 x 512 MFMA cycles, so the
// SIMD's two waves need 2940 VALU + 2048 MFMA cycles and spend them one after the other (MI355X_MICROARCH.md, "Two waves per
// SIMD", items 1 and 5: what a barrier interval PAIRS is the lever).
// Here the halves run the same three segments per tile but the barrier sits at a different place in each:
//     late half  (waves 0-3, the DMA issuers):   b_k | S(k)        softmax(k)   PV(k)
//     early half (waves 4-7):                    b_k | softmax(k)  PV(k)        S(k+1)
// so one wave's MFMA segments face its partner's softmax.  The early half reads K(k+1) one interval ahead and both read V(k)
// in interval k, hence a 4-slot K/V ring: tile k+3 is requested at b_k into the slot of tile k-1 (last read before b_k by
// both halves); tile k+1 has landed before b_k (`vmcnt(8)`: only tile k+2 may still be in flight).  Same arithmetic, same
// rounding points, same per-row results as attn_fwd128p_kernel (bit-identical).
#ifdef MM_ATTN_QDIAG               // timing experiments (results wrong by design), tools/build_diag.sh: AttnArgs::diag bits
#define MM_QDIAG a.diag
#else
#define MM_QDIAG 0
#endif
template <int RD, int NW>              // RD: fragment ring depth (RD - 1 LDS reads in flight ahead of every MFMA); NW: waves issuing the K/V DMA (4 or 8)
__global__ __launch_bounds__(512, 2) void attn_fwd128q_kernel(AttnArgs a) {
  constexpr int QB = 256, BKV = 64, NDS = 8, NDB = 4, TILE = BKV * 256, NST = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [4 slots][K 16 KiB | V 16 KiB][key-valid bits: 8 B per key tile]
  const int l = threadIdx.x & 63, h = l >> 5;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int qblk, b, hq;
  attn_work_item(blockIdx.x, (a.Sq + QB - 1) / QB, a.B, a.Hq, a.Hkv, a.causal != 0, qblk, b, hq);
  const int hkv = hq / (a.Hq / a.Hkv);
  const int q0 = qblk * QB + w * 32;
  const int qi = q0 + (l & 31);
  const int shift = a.Skv - a.Sq;
  const bf16* Q = (const bf16*)a.q + b * a.q_sb + hq * a.q_sh;
  const SRsrc rk = rows_rsrc((const bf16*)a.k + b * a.k_sb + hkv * a.k_sh, a.Skv, a.k_ss);
  const SRsrc rv = rows_rsrc((const bf16*)a.v + b * a.v_sb + hkv * a.v_sh, a.Skv, a.v_ss);
  const unsigned lds0 = (unsigned)(uintptr_t)LDS_PTR(char, smem);
  const ImgaBases bases = imga_bases(smem, TILE);

  bf16x8 qf[NDS];
  {
    const bf16* qrow = qi < a.Sq ? Q + (int64_t)qi * a.q_ss : nullptr;
#pragma unroll
    for (int ds = 0; ds < NDS; ++ds) qf[ds] = row_frag_global(qrow, ds);
  }
  f32x16 o_acc[NDB];
#pragma unroll
  for (int i = 0; i < NDB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o_acc[i][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float sc = a.scale * LOG2E;

  int ntiles = (a.Skv + BKV - 1) / BKV;
  if (a.causal) {
    const int qmax = min(a.Sq - 1, qblk * QB + QB - 1) + shift;
    ntiles = qmax < 0 ? 0 : min(ntiles, qmax / BKV + 1);
  }
  // Key-valid bits (inside Skv and not masked out) of every tile, built once: a global load of the key mask INSIDE the tile
  // loop makes hipcc put `s_waitcnt vmcnt(0)` at the join in front of the first QK^T product, which drains the K/V DMA an
  // issuing wave requested a few instructions earlier (its whole latency exposed on every tile, on every SIMD).
  unsigned long long* kvbits = (unsigned long long*)(smem + NST * 2 * TILE);
  for (int t = w; t < ntiles; t += 8) {
    bool kvalid = (t * BKV + l) < a.Skv;
    if (kvalid && a.kmask) kvalid = a.kmask[(int64_t)b * a.Skv + t * BKV + l] != 0;
    const unsigned long long bits = __ballot(kvalid);
    if (l == 0) kvbits[t] = bits;
  }
  // ... and for the same reason the Q fragments are consumed once HERE: hipcc then waits for their loads now, not with a
  // `vmcnt(0)` in front of the first product of every tile
#pragma unroll
  for (int ds = 0; ds < NDS; ++ds) asm volatile("" ::"v"(qf[ds]));
  unsigned lk[2], lv[2];
  imga_lane_patterns(lk, lv, a.k_ss, a.v_ss);
  auto issue = [&](int t) {
    if (MM_QDIAG & 1) return;
    imga_issue_kv_n<NW>(w, lds0 + (unsigned)((t & (NST - 1)) * 2 * TILE), t * BKV, rk, rv, lk, lv, a.k_ss, a.v_ss, TILE);
  };
  // wave-uniform: the tile exists, this wave has rows, and the tile is not wholly above the wave's causal diagonal
  auto active = [&](int t) { return t < ntiles && q0 < a.Sq && !(a.causal && t * BKV > q0 + 31 + shift); };

  // state handed from S(t) to softmax(t) (for the early half: across a barrier)
  f32x16 s_acc[2];
  unsigned long long kbits = 0;
  bool need_mask = false;

  auto seg_s = [&](int t) {           // S^T = K(t) . Q^T: 16 MFMAs, K fragment i+3 requested before MFMA i
    const int kv0 = t * BKV, so = (t & (NST - 1)) * 2 * TILE;
    {
      const u32x2 kv = *(const u32x2*)(kvbits + t);                 // same address in every lane (LDS broadcast)
      kbits = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(kv[0]) |
              ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(kv[1]) << 32);     // (unsigned): the builtin returns int
    }
    need_mask = (kbits != ~0ull) || (a.causal && (kv0 + BKV - 1) > (q0 + shift));
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    bf16x8 kf[RD];
#pragma unroll
    for (int j = 0; j < RD - 1; ++j) kf[j] = (MM_QDIAG & 16) ? qf[j & 7] : imga_kfrag(bases, so, 0, j);
    if (a.prio == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (i + RD - 1 < 16 && !(MM_QDIAG & 16)) kf[(i + RD - 1) % RD] = imga_kfrag(bases, so, (i + RD - 1) >> 3, (i + RD - 1) & 7);
      // the first product of each chain takes C = 0 as an inline constant (no 32-register clear per tile)
      if (!(MM_QDIAG & 4)) s_acc[i >> 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[i % RD], qf[i & 7], (i & 7) ? s_acc[i >> 3] : zero16, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (a.prio == 1) __builtin_amdgcn_s_setprio(0);
  };

  auto seg_softmax_pv = [&](int t) {  // mask, online softmax, O rescale, then O^T += V(t)^T . P^T: 16 MFMAs
    const int kv0 = t * BKV, so = (t & (NST - 1)) * 2 * TILE;
    bf16x8 vf[RD];                                                   // first V^T fragments travel under the softmax
#pragma unroll
    for (int j = 0; j < RD - 1; ++j) vf[j] = (MM_QDIAG & 16) ? qf[j & 7] : imga_vfrag(bases, so, j >> 2, j & 3);
    if (need_mask) {
      const unsigned long long kb2 = kbits >> (4 * h);
      const int dlim = a.causal ? (qi + shift - kv0 - 4 * h) : 0x7fffffff;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = kb * 32 + (r & 3) + 8 * (r >> 2);
          const bool ok = ((kb2 >> c) & 1ull) && c <= dlim;
          s_acc[kb][r] = ok ? s_acc[kb][r] : -INFINITY;
        }
    }
    bf16x8 pf[2][2];
    if (MM_QDIAG & 2) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) pf[kb][r >> 3][r & 7] = (bf16)s_acc[kb][r];
    } else {
    const float mx = swap32_max(rowmax32(s_acc[0], s_acc[1])) * sc;
    const float m_new = fmaxf(m_run, mx);
    const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_safe);
    // packed fp32 (v_pk_fma_f32 / v_pk_add_f32: two elements per instruction, same IEEE results): the softmax is the longer
    // of the kernel's two pipes, so instruction count is time here
    f32x2 rs2 = {0.f, 0.f};
    const f32x2 sc2 = {sc, sc}, nm2 = {-m_safe, -m_safe};
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 sv = {s_acc[kb][r], s_acc[kb][r + 1]};
        const f32x2 t2 = __builtin_elementwise_fma(sv, sc2, nm2);
        f32x2 p2;
        p2[0] = __builtin_amdgcn_exp2f(t2[0]);
        p2[1] = __builtin_amdgcn_exp2f(t2[1]);
        rs2 += p2;
        pf[kb][r >> 3][r & 7] = (bf16)p2[0];
        pf[kb][r >> 3][(r & 7) + 1] = (bf16)p2[1];
      }
    const float rs = rs2[0] + rs2[1];
    l_run = l_run * alpha + rs;
    m_run = m_new;
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) o_acc[db][r] *= alpha;
    }
    __builtin_amdgcn_sched_barrier(0);
    if (a.prio == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 16; ++i) {                                   // i = db * 4 + ks
      if (i + RD - 1 < 16 && !(MM_QDIAG & 16)) vf[(i + RD - 1) % RD] = imga_vfrag(bases, so, (i + RD - 1) >> 2, (i + RD - 1) & 3);
      if (!(MM_QDIAG & 8)) o_acc[i >> 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[i % RD], pf[(i >> 1) & 1][i & 1], o_acc[i >> 2], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (a.prio == 1) __builtin_amdgcn_s_setprio(0);
  };

  if ((a.prio == 2 && w >= 4) || (a.prio == 3 && w < 4)) __builtin_amdgcn_s_setprio(1);
  // DMA: an issuing wave has PT = 32 / NW pieces per tile in flight.  With NW = 8 both halves carry half of the issue cost
  // (a piece costs its wave 100+ cycles of issue time beside the fragment reads); the waits below are no-ops for a wave that
  // issues nothing (NW = 4: waves 4-7).
  auto wait_tiles = [&](int in_flight) {          // at most `in_flight` tiles (0, 1 or 2) of this wave's pieces may still be under way
    if (NW == 8) {
      if (in_flight >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (in_flight == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      if (in_flight >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (in_flight == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  };
  if (ntiles > 0) issue(0);
  if (ntiles > 1) issue(1);
  if (ntiles > 2) issue(2);
  wait_tiles(ntiles > 2 ? 2 : (ntiles > 1 ? 1 : 0));                      // tile 0 has landed (the early half reads it right
  __builtin_amdgcn_s_barrier();                                            // after this opening barrier)
  if (w < 4) {
    // ---- late half
    for (int k = 0; k < ntiles; ++k) {
      wait_tiles(k + 2 < ntiles ? 1 : 0);                                  // tile k+1 has landed: the early half reads its K now
      __builtin_amdgcn_s_barrier();
      if (k + 3 < ntiles) issue(k + 3);
      if (active(k)) {
        seg_s(k);
        seg_softmax_pv(k);
      }
    }
  } else {
    // ---- early half: its barrier sits between S(k) and softmax(k)
    bool act = active(0);
    if (act) seg_s(0);
    for (int k = 0; k < ntiles; ++k) {
      wait_tiles(k + 2 < ntiles ? 1 : 0);
      __builtin_amdgcn_s_barrier();
      if (k + 3 < ntiles) issue(k + 3);
      if (act) seg_softmax_pv(k);
      act = active(k + 1);
      if (act) seg_s(k + 1);
    }
  }
  const float l_tot = swap32_sum(l_run);
  const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
  if (qi < a.Sq) {
    bf16* orow = (bf16*)a.out + (((int64_t)b * a.Sq + qi) * a.Hq + hq) * 128;
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16)(o_acc[db][rg * 4 + e] * inv);
        *(bf16x4*)(orow + db * 32 + 8 * rg + 4 * h) = o;
      }
    if (h == 0) a.lse[((int64_t)b * a.Hq + hq) * a.Sq + qi] = l_tot > 0.f ? (m_run + log2f(l_tot)) * LN2 : INFINITY;
  }
}

// dK/dV for D = 128: 4 waves x 32 keys, TWO workgroups per CU.  Register diet that makes 2 waves/SIMD fit: V fragments
// come from an LDS image of the workgroup's 128 keys (not registers) and the Q / dO tiles arrive by LDS-DMA (no staging
// registers) into a 2-deep ring, one barrier per query tile.
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv128_kernel(AttnArgs a) {
  constexpr int BQ = 32, NDS = 8, NDB = 4, QT = BQ * 256;            // 8 KiB per 32-row tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Vimg = smem;                                                   // 128 keys x 256 B
  char* ring = smem + 128 * 256;                                       // [2 stages][Q 8 KiB | dO 8 KiB]
  float* rowc = (float*)(smem + 128 * 256 + 4 * QT);                   // [2 stages][lse*log2e (32) | delta (32)]
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6, h = l >> 5;
  const int b = blockIdx.z, hkv = blockIdx.y;
  const int G = a.Hq / a.Hkv;
  const int kblk = blockIdx.x * 128;
  const int k0 = kblk + w * 32;
  const int ki = k0 + (l & 31);
  const int shift = a.Skv - a.Sq;
  const bf16* K = (const bf16*)a.k + b * a.k_sb + hkv * a.k_sh;
  const unsigned lds0 = (unsigned)(uintptr_t)LDS_PTR(char, smem);
  {
    const SRsrc rv = rows_rsrc((const bf16*)a.v + b * a.v_sb + hkv * a.v_sh, a.Skv, a.v_ss);
    imgb_dma<128, 4>(lds0, rv, a.v_ss, kblk);
  }
  bf16x8 kf[NDS];
  {
    const bf16* krow = ki < a.Skv ? K + (int64_t)ki * a.k_ss : nullptr;
#pragma unroll
    for (int ds = 0; ds < NDS; ++ds) kf[ds] = row_frag_global(krow, ds);
  }
  bool kvalid = ki < a.Skv;
  if (kvalid && a.kmask) kvalid = a.kmask[(int64_t)b * a.Skv + ki] != 0;
  const float sc = a.scale * LOG2E;
  f32x16 dk_acc[NDB], dv_acc[NDB];
#pragma unroll
  for (int i = 0; i < NDB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk_acc[i][r] = 0.f; dv_acc[i][r] = 0.f; }

  int qt0 = 0;
  if (a.causal) qt0 = max(0, kblk - shift) / BQ;
  const int nqt = (a.Sq + BQ - 1) / BQ;
  const int per_head = max(0, nqt - qt0);
  const int niter = per_head * G;
  const int64_t do_ss = (int64_t)a.Hq * 128;
  float rc = 0.f;
  auto issue = [&](int it) {
    const int g = it / per_head, qb = (qt0 + it % per_head) * BQ;
    const int hq = hkv * G + g;
    const SRsrc rq = rows_rsrc((const bf16*)a.q + b * a.q_sb + hq * a.q_sh, a.Sq, a.q_ss);
    const SRsrc rdo = rows_rsrc((const bf16*)a.dout + ((int64_t)b * a.Sq * a.Hq + hq) * 128, a.Sq, do_ss);
    const unsigned st = lds0 + 128 * 256 + (unsigned)((it & 1) * 2 * QT);
    imgb_dma<BQ, 4>(st, rq, a.q_ss, qb);
    imgb_dma<BQ, 4>(st + QT, rdo, do_ss, qb);
    if (threadIdx.x < 64) {
      const int qq = qb + (threadIdx.x & 31);
      const int64_t ro = ((int64_t)b * a.Hq + hq) * a.Sq + qq;
      if (threadIdx.x < 32) rc = qq < a.Sq ? a.lse[ro] * LOG2E : INFINITY;
      else rc = qq < a.Sq ? a.delta[ro] : 0.f;
    }
  };
  if (niter > 0) {
    issue(0);
    if (threadIdx.x < 64) rowc[threadIdx.x] = rc;
  }
  for (int it = 0; it < niter; ++it) {
    const int qb = (qt0 + it % per_head) * BQ;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (it + 1 < niter) issue(it + 1);
    const char* Qt = ring + (it & 1) * 2 * QT;
    const char* Ot = Qt + QT;
    const float* rcs = rowc + (it & 1) * 64;
    // wave-uniform skip: every key of this wave lies above the causal diagonal of every row of this query tile
    if (!(a.causal && k0 > qb + BQ - 1 + shift)) {
      f32x16 s_acc, dp_acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s_acc[r] = 0.f; dp_acc[r] = 0.f; }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ds = 0; ds < NDS; ++ds) {
        s_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(imgb_rowfrag(Qt, 0, ds), kf[ds], s_acc, 0, 0, 0);
        dp_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(imgb_rowfrag(Ot, 0, ds), imgb_rowfrag(Vimg, w * 32, ds), dp_acc, 0, 0, 0);
      }
      __builtin_amdgcn_s_setprio(0);
      bf16x8 pf[2], dsf[2];
      // masks only where they can bite: a padded/out-of-range key in this wave, or a tile that touches the diagonal
      const bool need_mask = (__ballot(kvalid) != ~0ull) || (a.causal && (k0 + 31) > (qb + shift));
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ql = acc_row(r, h);
        float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s_acc[r], sc, -rcs[ql]));
        if (need_mask) {
          bool ok = kvalid;
          if (a.causal) ok = ok && ki <= (qb + ql + shift);
          p = ok ? p : 0.f;
        }
        const float dsv = p * (dp_acc[r] - rcs[32 + ql]) * a.scale;
        pf[r >> 3][r & 7] = (bf16)p;
        dsf[r >> 3][r & 7] = (bf16)dsv;
      }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          dv_acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(imgb_tfrag(Ot, db, s * 16), pf[s], dv_acc[db], 0, 0, 0);
          dk_acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(imgb_tfrag(Qt, db, s * 16), dsf[s], dk_acc[db], 0, 0, 0);
        }
      __builtin_amdgcn_s_setprio(0);
    }
    // row constants of the NEXT tile: their ring slot was last read in iteration it-1, which every wave has left
    if (it + 1 < niter && threadIdx.x < 64) rowc[((it + 1) & 1) * 64 + threadIdx.x] = rc;
  }
  if (ki < a.Skv) {
    bf16* dkrow = (bf16*)a.dk + b * a.k_sb + hkv * a.k_sh + (int64_t)ki * a.k_ss;
    bf16* dvrow = (bf16*)a.dv + b * a.v_sb + hkv * a.v_sh + (int64_t)ki * a.v_ss;
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        bf16x4 ok_, ov_;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          ok_[e] = (bf16)dk_acc[db][rg * 4 + e];
          ov_[e] = (bf16)dv_acc[db][rg * 4 + e];
        }
        *(bf16x4*)(dkrow + db * 32 + 8 * rg + 4 * h) = ok_;
        *(bf16x4*)(dvrow + db * 32 + 8 * rg + 4 * h) = ov_;
      }
  }
}

// dK/dV for D = 128, balanced form (used when the GQA group size is even).  Under a causal mask key block j meets
// (nkb - j) query blocks, a 16:1 spread at S = 2048, and with B*Hkv*nkb = 512 items on 512 workgroup slots the launch
// lasts as long as block 0.  Here ONE workgroup of 8 waves owns the PAIR of key blocks (j, nkb-1-j), processed one after
// the other, so every workgroup does the same (nkb+1) block-units; its two 4-wave teams split the query heads of the
// group (team t: heads t*G/2 ..), each with its own Q/dO ring, and their dK/dV accumulators are summed through LDS at
// the end of a block (fixed order: deterministic, no atomics, no global partials).  One workgroup per CU, 2 waves/SIMD.
__global__ __launch_bounds__(512, 2) void attn_bwd_dkv128_pair_kernel(AttnArgs a) {
  constexpr int BQ = 32, NDS = 8, NDB = 4, QT = BQ * 256;            // 8 KiB per 32-row tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Vimg = smem;                                                   // 128 keys x 256 B
  const int l = threadIdx.x & 63, h = l >> 5;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave-uniform: LDS-DMA addresses live in SGPRs
  const int team = w >> 2, wt = w & 3, tt = threadIdx.x & 255;
  char* ring = smem + 128 * 256 + team * 4 * QT;                       // per team [2 stages][Q 8 KiB | dO 8 KiB]
  float* red = (float*)(smem + 128 * 256);                             // 64 KiB = both rings, reused for the team sum
  float* rowc = (float*)(smem + 128 * 256 + 8 * QT) + team * 128;      // per team [2 stages][lse*log2e (32) | delta (32)]
  const int b = blockIdx.z, hkv = blockIdx.y;
  const int G = a.Hq / a.Hkv, GH = G / 2;
  const int nkb = (a.Skv + 127) / 128;
  const int shift = a.Skv - a.Sq;
  const bf16* K = (const bf16*)a.k + b * a.k_sb + hkv * a.k_sh;
  const unsigned lds0 = (unsigned)(uintptr_t)LDS_PTR(char, smem);
  const unsigned ring0 = lds0 + 128 * 256 + (unsigned)(team * 4 * QT);
  const float sc = a.scale * LOG2E;
  const int nqt = (a.Sq + BQ - 1) / BQ;
  const int64_t do_ss = (int64_t)a.Hq * 128;

  for (int pass = 0; pass < 2; ++pass) {
    const int kb = pass == 0 ? (int)blockIdx.x : nkb - 1 - (int)blockIdx.x;
    if (pass == 1 && kb <= (int)blockIdx.x) break;                    // odd count: the middle block has no partner
    const int kblk = kb * 128;
    const int k0 = kblk + wt * 32;
    const int ki = k0 + (l & 31);
    {
      const SRsrc rv = rows_rsrc((const bf16*)a.v + b * a.v_sb + hkv * a.v_sh, a.Skv, a.v_ss);
      imgb_dma<128, 8>(lds0, rv, a.v_ss, kblk);
    }
    bf16x8 kf[NDS];
    {
      const bf16* krow = ki < a.Skv ? K + (int64_t)ki * a.k_ss : nullptr;
#pragma unroll
      for (int ds = 0; ds < NDS; ++ds) kf[ds] = row_frag_global(krow, ds);
    }
    bool kvalid = ki < a.Skv;
    if (kvalid && a.kmask) kvalid = a.kmask[(int64_t)b * a.Skv + ki] != 0;
    f32x16 dk_acc[NDB], dv_acc[NDB];
#pragma unroll
    for (int i = 0; i < NDB; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) { dk_acc[i][r] = 0.f; dv_acc[i][r] = 0.f; }

    int qt0 = 0;
    if (a.causal) qt0 = max(0, kblk - shift) / BQ;
    const int per_head = max(0, nqt - qt0);
    const int niter = per_head * GH;                                   // the same count for both teams
    float rc = 0.f;
    auto issue = [&](int it) {
      const int g = team * GH + it / per_head, qb = (qt0 + it % per_head) * BQ;
      const int hq = hkv * G + g;
      const SRsrc rq = rows_rsrc((const bf16*)a.q + b * a.q_sb + hq * a.q_sh, a.Sq, a.q_ss);
      const SRsrc rdo = rows_rsrc((const bf16*)a.dout + ((int64_t)b * a.Sq * a.Hq + hq) * 128, a.Sq, do_ss);
      const unsigned st = ring0 + (unsigned)((it & 1) * 2 * QT);
      imgb_dma_team<BQ, 4>(st, rq, a.q_ss, qb, wt);
      imgb_dma_team<BQ, 4>(st + QT, rdo, do_ss, qb, wt);
      if (tt < 64) {
        const int qq = qb + (tt & 31);
        const int64_t ro = ((int64_t)b * a.Hq + hq) * a.Sq + qq;
        if (tt < 32) rc = qq < a.Sq ? a.lse[ro] * LOG2E : INFINITY;
        else rc = qq < a.Sq ? a.delta[ro] : 0.f;
      }
    };
    if (niter > 0) {
      issue(0);
      if (tt < 64) rowc[tt] = rc;
    }
    for (int it = 0; it < niter; ++it) {
      const int qb = (qt0 + it % per_head) * BQ;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (it + 1 < niter) issue(it + 1);
      const char* Qt = ring + (it & 1) * 2 * QT;
      const char* Ot = Qt + QT;
      const float* rcs = rowc + (it & 1) * 64;
      if (!(a.causal && k0 > qb + BQ - 1 + shift)) {
        f32x16 s_acc, dp_acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s_acc[r] = 0.f; dp_acc[r] = 0.f; }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ds = 0; ds < NDS; ++ds) {
          s_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(imgb_rowfrag(Qt, 0, ds), kf[ds], s_acc, 0, 0, 0);
          dp_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(imgb_rowfrag(Ot, 0, ds), imgb_rowfrag(Vimg, wt * 32, ds), dp_acc, 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        bf16x8 pf[2], dsf[2];
        const bool need_mask = (__ballot(kvalid) != ~0ull) || (a.causal && (k0 + 31) > (qb + shift));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ql = acc_row(r, h);
          float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s_acc[r], sc, -rcs[ql]));
          if (need_mask) {
            bool ok = kvalid;
            if (a.causal) ok = ok && ki <= (qb + ql + shift);
            p = ok ? p : 0.f;
          }
          const float dsv = p * (dp_acc[r] - rcs[32 + ql]) * a.scale;
          pf[r >> 3][r & 7] = (bf16)p;
          dsf[r >> 3][r & 7] = (bf16)dsv;
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            dv_acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(imgb_tfrag(Ot, db, s * 16), pf[s], dv_acc[db], 0, 0, 0);
            dk_acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(imgb_tfrag(Qt, db, s * 16), dsf[s], dk_acc[db], 0, 0, 0);
          }
        __builtin_amdgcn_s_setprio(0);
      }
      if (it + 1 < niter && tt < 64) rowc[((it + 1) & 1) * 64 + tt] = rc;
    }
    // ---- sum the two teams' accumulators through LDS (team 1 writes, team 0 adds: fixed order), then team 0 stores
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                                   // every wave has left the rings and the V image
    float* mine = red + (wt * 64) * 64 + l;                            // [wave-in-team][64 values][lane]
    if (team == 1) {
#pragma unroll
      for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) mine[(db * 16 + r) * 64] = dk_acc[db][r];
    }
    __syncthreads();
    if (team == 0) {
#pragma unroll
      for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) dk_acc[db][r] += mine[(db * 16 + r) * 64];
    }
    __syncthreads();
    if (team == 1) {
#pragma unroll
      for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) mine[(db * 16 + r) * 64] = dv_acc[db][r];
    }
    __syncthreads();
    if (team == 0) {
#pragma unroll
      for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) dv_acc[db][r] += mine[(db * 16 + r) * 64];
      if (ki < a.Skv) {
        bf16* dkrow = (bf16*)a.dk + b * a.k_sb + hkv * a.k_sh + (int64_t)ki * a.k_ss;
        bf16* dvrow = (bf16*)a.dv + b * a.v_sb + hkv * a.v_sh + (int64_t)ki * a.v_ss;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            bf16x4 ok_, ov_;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              ok_[e] = (bf16)dk_acc[db][rg * 4 + e];
              ov_[e] = (bf16)dv_acc[db][rg * 4 + e];
            }
            *(bf16x4*)(dkrow + db * 32 + 8 * rg + 4 * h) = ok_;
            *(bf16x4*)(dvrow + db * 32 + 8 * rg + 4 * h) = ov_;
          }
      }
    }
    __syncthreads();                                                   // `red` is free again before the next pass's DMA
  }
}

// dK/dV for D = 128, paired key blocks, with prefetched fragments.  attn_bwd_dkv128_pair_kernel's ISA had the same disease as
// the forward (every MFMA behind its own ds_read + lgkmcnt(0), one register quad for all fragments) and a worse one: the 32 row
// constants (lse, delta) of a query tile were fetched by 17 ds_read2_b32, each followed by lgkmcnt(0) -- 17 dependent LDS
// round trips per 32-MFMA iteration; rocprofv3: waves parked 69 % of their cycles.  Here (same arithmetic and rounding, results
// bit-identical):
//   * all tiles (V and K images of the 128 keys, the teams' Q / dO rings) are image (a): base + constant addressing;
//   * K is an LDS image like V instead of 32 registers of fragments: that pays for a 4-deep fragment ring over the 48
//     fragments of an iteration (Q, K, dO, V row fragments for S^T and dP^T; dO^T, Q^T transposed fragments for dV, dK);
//   * the row constants are read as eight 16-byte vectors (rows 8g + 4h .. + 3 are consecutive), issued together.
#ifndef MM_DKV_DIAG                 // timing experiments on the paired dK/dV kernel (results wrong by design; tools/build_diag.sh):
#define MM_DKV_DIAG 0               // 1 no Q/dO DMA, 2 no softmax/dS VALU, 4 no S/dP MFMAs, 8 no dV/dK MFMAs, 16 no LDS fragment reads, 32 no barrier
#endif
template <int RD, bool DKV_LATE_ISSUE> // RD fragment ring slots: 8 (6 fragments in flight) or 4; DKV_LATE_ISSUE: DMA of the next tile after the S/dP products
__global__ __launch_bounds__(512, 2) void attn_bwd_dkv128_pairp_kernel(AttnArgs a) {
  constexpr int BQ = 32, NDB = 4, QT = BQ * 256, IMG = 128 * 256;     // 8 KiB per 32-row tile, 32 KiB per 128-key image
  extern __shared__ __attribute__((aligned(16))) char smem[];          // [V image | K image | 2 teams x 2 stages x (Q | dO) | row constants]
  const int l = threadIdx.x & 63, h = l >> 5;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave-uniform: LDS-DMA addresses live in SGPRs
  const int team = w >> 2, wt = w & 3, tt = threadIdx.x & 255;
  float* red = (float*)(smem + 2 * IMG);                               // 64 KiB = both rings, reused for the team sum
  float* rowc = (float*)(smem + 2 * IMG + 8 * QT) + team * 128;        // per team [2 stages][lse*log2e (32) | delta (32)]
  // 1-D grid, XCD-aware: the hardware deals consecutive workgroup ids to the 8 XCDs in turn, and the `npair` workgroups of one
  // (batch, KV head) stream the SAME Q / dO rows (G heads x Sq x 512 B).  With a (pair, head, batch) grid those workgroups sat
  // on 8 different XCDs, every L2 saw each tile once and the kernel pulled 4.5x its algorithmic bytes from beyond L2.  Here
  // group g = (b, hkv) lives on XCD g % 8 and its pairs fill that XCD's consecutive slots (needs B * Hkv % 8 == 0, else the
  // plain order).
  const int npair = (((a.Skv + 127) / 128) + 1) / 2, ngroup = a.B * a.Hkv;
  int pair_i, grp;
  if ((ngroup & 7) == 0) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    pair_i = slot % npair;
    grp = (slot / npair) * 8 + xcd;
  } else {
    pair_i = blockIdx.x % npair;
    grp = blockIdx.x / npair;
  }
  const int b = grp / a.Hkv, hkv = grp % a.Hkv;
  const int G = a.Hq / a.Hkv;
  const int nkb = (a.Skv + 127) / 128;
  const int shift = a.Skv - a.Sq;
  const unsigned lds0 = (unsigned)(uintptr_t)LDS_PTR(char, smem);
  const unsigned ring0 = lds0 + 2 * IMG + (unsigned)(team * 4 * QT);
  const float sc = a.scale * LOG2E;
  const int nqt = (a.Sq + BQ - 1) / BQ;
  const int64_t do_ss = (int64_t)a.Hq * 128;
  const ImgaBases bases = imga_bases(smem, 0);
  // per-lane bases: key images at this wave's 32 keys (row fragments), this team's ring (row + transposed fragments)
  const char* vrow[2] = {bases.kr[0] + 2048 * 4 * wt, bases.kr[1] + 2048 * 4 * wt};
  const char* ring_r[2] = {bases.kr[0] + 2 * IMG + team * 4 * QT, bases.kr[1] + 2 * IMG + team * 4 * QT};
  const char* ring_t[2] = {bases.vt[0] + 2 * IMG + team * 4 * QT, bases.vt[1] + 2 * IMG + team * 4 * QT};
  // DMA source patterns (image (a)): a 1-KiB piece = 8 rows x 128 B, lane pattern s = (piece >> 1) & 1
  unsigned lk[2], lv[2], lq, ldo;
  imga_lane_patterns(lk, lv, a.k_ss, a.v_ss);
  {
    const int rl = (l >> 2) & 7, cl = 4 * (l >> 5) + ((l & 3) ^ (((wt & 1) << 1) | ((l >> 4) & 1)));   // this wave's ring pieces: 2wt, 2wt+1
    lq = (unsigned)((int64_t)rl * a.q_ss * 2 + cl * 16);
    ldo = (unsigned)((int64_t)rl * do_ss * 2 + cl * 16);
  }

  for (int pass = 0; pass < 2; ++pass) {
    const int kb = pass == 0 ? pair_i : nkb - 1 - pair_i;
    if (pass == 1 && kb <= pair_i) break;                    // odd count: the middle block has no partner
    const int kblk = kb * 128;
    const int k0 = kblk + wt * 32;
    const int ki = k0 + (l & 31);
    {   // V and K images of the 128 keys: 32 pieces each, 4 per wave
      const SRsrc rv = rows_rsrc((const bf16*)a.v + b * a.v_sb + hkv * a.v_sh, a.Skv, a.v_ss);
      const SRsrc rk = rows_rsrc((const bf16*)a.k + b * a.k_sb + hkv * a.k_sh, a.Skv, a.k_ss);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int pc = w * 4 + i, prow = kblk + 8 * (pc >> 1);
        lds_dma16(rv, lv[i >> 1] + (unsigned)((int64_t)prow * a.v_ss * 2 + (i & 1) * 128), lds0 + pc * 1024);
        lds_dma16(rk, lk[i >> 1] + (unsigned)((int64_t)prow * a.k_ss * 2 + (i & 1) * 128), lds0 + IMG + pc * 1024);
      }
    }
    bool kvalid = ki < a.Skv;
    if (kvalid && a.kmask) kvalid = a.kmask[(int64_t)b * a.Skv + ki] != 0;
    f32x16 dk_acc[NDB], dv_acc[NDB];
#pragma unroll
    for (int i = 0; i < NDB; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) { dk_acc[i][r] = 0.f; dv_acc[i][r] = 0.f; }

    int qt0 = 0;
    if (a.causal) qt0 = max(0, kblk - shift) / BQ;
    const int per_head = max(0, nqt - qt0);
    // the group's (query head, query tile) items in head-major order, cut in two: team 0 takes items [0, niter), team 1
    // [niter, total).  Even groups: each team gets G/2 whole heads.  Odd groups (Qwen2-7B: 7): team 1 starts in the middle of
    // a head and, when `total` is odd, sits out the last step (it still meets the barrier).
    const int total = per_head * G, niter = (total + 1) / 2;           // the same step count for both teams
    const int item0 = team * niter;
    float rc = 0.f;
    auto issue = [&](int it) {
      const int idx = item0 + it;
      if (idx >= total || (MM_DKV_DIAG & 1)) return;
      const int g = idx / per_head, qb = (qt0 + idx % per_head) * BQ;
      const int hq = hkv * G + g;
      const SRsrc rq = rows_rsrc((const bf16*)a.q + b * a.q_sb + hq * a.q_sh, a.Sq, a.q_ss);
      const SRsrc rdo = rows_rsrc((const bf16*)a.dout + ((int64_t)b * a.Sq * a.Hq + hq) * 128, a.Sq, do_ss);
      const unsigned st = ring0 + (unsigned)((it & 1) * 2 * QT) + (unsigned)(wt * 2) * 1024u;
      const int prow = qb + 8 * wt;                                    // this wave's two pieces of each tile: rows 8wt .. 8wt+7
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        lds_dma16(rq, lq + (unsigned)((int64_t)prow * a.q_ss * 2 + i * 128), st + i * 1024);
        lds_dma16(rdo, ldo + (unsigned)((int64_t)prow * do_ss * 2 + i * 128), st + QT + i * 1024);
      }
      if (tt < 64) {
        const int qq = qb + (tt & 31);
        const int64_t ro = ((int64_t)b * a.Hq + hq) * a.Sq + qq;
        if (tt < 32) rc = qq < a.Sq ? a.lse[ro] * LOG2E : INFINITY;
        else rc = qq < a.Sq ? a.delta[ro] : 0.f;
      }
    };
    if (niter > 0) {
      issue(0);
      if (tt < 64) rowc[tt] = rc;
    }
    for (int it = 0; it < niter; ++it) {
      const int qb = (qt0 + (item0 + it) % per_head) * BQ;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (!(MM_DKV_DIAG & 32)) __builtin_amdgcn_s_barrier();
      // The next tile's DMA (4 pieces per wave) is NOT issued here: in front of the S / dP products it shares the LDS path with
      // their 32 fragment reads and costs 100-185 cycles per piece (timing build without the DMA: -25 % kernel time).  It is
      // issued after those products, in front of the VALU-only dS segment (MI355X_MICROARCH.md: 25-60 cycles there).  Its ring
      // slot was last read in iteration it - 1, which every wave of the team left before this barrier.
      bool issued = false;
      if (!DKV_LATE_ISSUE && it + 1 < niter) { issue(it + 1); issued = true; }
      const int so = (it & 1) * 2 * QT;
      const float* rcs = rowc + (it & 1) * 64;
      if (item0 + it < total && !(a.causal && k0 > qb + BQ - 1 + shift)) {
        // fragment j of the 48: j < 32: k-step j >> 2, kind j & 3 (0: Q rows, 1: K rows, 2: dO rows, 3: V rows);
        //                       j >= 32: d block (j - 32) >> 2, 16-query step ((j - 32) >> 1) & 1, kind (j & 1) (0: dO^T, 1: Q^T)
        auto frag = [&](int j) -> bf16x8 {
          if (MM_DKV_DIAG & 16) { bf16x8 z; for (int e = 0; e < 8; ++e) z[e] = (bf16)(float)(j + l); return z; }
          if (j < 32) {
            const int ds = j >> 2, kind = j & 3, a2 = ds & 1, off = 512 * (ds >> 1);
            if (kind == 0) return *(const bf16x8*)(ring_r[a2] + so + off);
            if (kind == 1) return *(const bf16x8*)(vrow[a2] + IMG + off);
            if (kind == 2) return *(const bf16x8*)(ring_r[a2] + so + QT + off);
            return *(const bf16x8*)(vrow[a2] + off);
          }
          const int jj = j - 32, db = jj >> 2, s16 = (jj >> 1) & 1, kind = jj & 1;
          const int off = so + (kind == 0 ? QT : 0) + 4096 * s16 + 512 * db;
          const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, ring_t[0] + off));
          const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, ring_t[1] + off));
          bf16x8 o;
          o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3];
          o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = hi[3];
          return o;
        };
        // S^T and dP^T take TWO fresh fragments per MFMA (K and Q rows, V and dO rows), so a 4-slot ring gave each product one
        // MFMA (32 cycles) of lookahead against an LDS round trip of 100+ cycles: rocprofv3 showed the waves parked 55 % of
        // their cycles with the LDS 30 % busy and no bank conflicts -- latency, not bandwidth.  8 slots, 6 fragments in flight
        // (3 MFMAs ahead); the 32 registers this costs are the row constants', which are now read AFTER the products.
        constexpr int LA = RD == 8 ? 6 : 4;
        f32x16 s_acc, dp_acc;
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        bf16x8 fr[RD];                                                 // ring: fragment j lives in fr[j % RD]
#pragma unroll
        for (int j = 0; j < LA; ++j) fr[j] = frag(j);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int m = 0; m < 16; ++m) {                                 // MFMA m uses fragments 2m (A) and 2m + 1 (B); once it is
          if (MM_DKV_DIAG & 4) { if (m < 2) { dp_acc = zero16; s_acc = zero16; } }
          else if (m & 1) dp_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[(2 * m) % RD], fr[(2 * m + 1) % RD], m > 1 ? dp_acc : zero16, 0, 0, 0);   // issued
          else s_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[(2 * m) % RD], fr[(2 * m + 1) % RD], m > 1 ? s_acc : zero16, 0, 0, 0);          // the slots
#pragma unroll
          for (int t = 0; t < 2; ++t) {                                // of MFMA m - 1 take the fragments of MFMA m + 3; the last
            const int jn = 2 * m + LA + t;                             // steps fetch the first four transposed fragments (32 .. 35)
            if (jn < 36) fr[jn % RD] = frag(jn);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
        if (DKV_LATE_ISSUE && it + 1 < niter) { issue(it + 1); issued = true; }
        // row constants of the lane's 16 query rows: rows 8g + 4h + (0..3), g = 0..3 -> 4 + 4 vectors of 16 bytes
        f32x4 lsev[4], dltv[4];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          lsev[g4] = *(const f32x4*)(rcs + 8 * g4 + 4 * h);
          dltv[g4] = *(const f32x4*)(rcs + 32 + 8 * g4 + 4 * h);
        }
        bf16x8 pf[2], dsf[2];
        if (MM_DKV_DIAG & 2) {
#pragma unroll
          for (int r = 0; r < 16; ++r) { pf[r >> 3][r & 7] = (bf16)s_acc[r]; dsf[r >> 3][r & 7] = (bf16)(dp_acc[r] + lsev[r >> 2][r & 3] + dltv[r >> 2][r & 3]); }
        } else {
        const bool need_mask = (__ballot(kvalid) != ~0ull) || (a.causal && (k0 + 31) > (qb + shift));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ql = acc_row(r, h);
          float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s_acc[r], sc, -lsev[r >> 2][r & 3]));
          if (need_mask) {
            bool ok = kvalid;
            if (a.causal) ok = ok && ki <= (qb + ql + shift);
            p = ok ? p : 0.f;
          }
          const float dsv = p * (dp_acc[r] - dltv[r >> 2][r & 3]) * a.scale;
          pf[r >> 3][r & 7] = (bf16)p;
          dsf[r >> 3][r & 7] = (bf16)dsv;
        }
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int m = 0; m < 16; ++m) {                                 // fragment 32 + m: d block m >> 2, query step (m >> 1) & 1
          const int db = m >> 2, s16 = (m >> 1) & 1;
          if (MM_DKV_DIAG & 8) { dk_acc[db][m] += (float)fr[(32 + m) % RD][0] + (float)dsf[s16][0] + (float)pf[s16][0]; }
          else if (m & 1) dk_acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[(32 + m) % RD], dsf[s16], dk_acc[db], 0, 0, 0);
          else dv_acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[(32 + m) % RD], pf[s16], dv_acc[db], 0, 0, 0);
          if (32 + m + 4 < 48) fr[(32 + m + 4) % RD] = frag(32 + m + 4);
          __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
      }
      if (!issued && it + 1 < niter) issue(it + 1);                     // a wave that skipped the tile (or the early-issue build)
      if (it + 1 < niter && tt < 64) rowc[((it + 1) & 1) * 64 + tt] = rc;
    }
    // ---- sum the two teams' accumulators through LDS (team 1 writes, team 0 adds: fixed order), then team 0 stores
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                                   // every wave has left the rings and the V image
    float* mine = red + (wt * 64) * 64 + l;                            // [wave-in-team][64 values][lane]
    if (team == 1) {
#pragma unroll
      for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) mine[(db * 16 + r) * 64] = dk_acc[db][r];
    }
    __syncthreads();
    if (team == 0) {
#pragma unroll
      for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) dk_acc[db][r] += mine[(db * 16 + r) * 64];
    }
    __syncthreads();
    if (team == 1) {
#pragma unroll
      for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) mine[(db * 16 + r) * 64] = dv_acc[db][r];
    }
    __syncthreads();
    if (team == 0) {
#pragma unroll
      for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) dv_acc[db][r] += mine[(db * 16 + r) * 64];
      if (ki < a.Skv) {
        bf16* dkrow = (bf16*)a.dk + b * a.k_sb + hkv * a.k_sh + (int64_t)ki * a.k_ss;
        bf16* dvrow = (bf16*)a.dv + b * a.v_sb + hkv * a.v_sh + (int64_t)ki * a.v_ss;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            bf16x4 ok_, ov_;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              ok_[e] = (bf16)dk_acc[db][rg * 4 + e];
              ov_[e] = (bf16)dv_acc[db][rg * 4 + e];
            }
            *(bf16x4*)(dkrow + db * 32 + 8 * rg + 4 * h) = ok_;
            *(bf16x4*)(dvrow + db * 32 + 8 * rg + 4 * h) = ov_;
          }
      }
    }
    __syncthreads();                                                   // `red` is free again before the next pass's DMA
  }
}

// dK/dV for D = 128 with the wave's K and V row fragments RESIDENT in registers (round 4; mm_set_option "attn_dkv_res").
// attn_bwd_dkv128_pairp_kernel is LDS-port-bound: 19.5 M LDS instructions per launch at B4 S2048 = 77 % of the port's cycles, 48 fragment
// reads per 32 MFMAs, 16 of them the SAME K / V rows of the wave's 32 keys in every iteration -- and all 256 registers of its
// two-waves-per-SIMD budget are in use (244 VGPRs), so those fragments cannot stay there.  Here a workgroup is ONE team of four waves, one
// wave per SIMD with the SIMD's 512 registers: 128 accumulators + 64 of resident K / V fragments + the transients; the team walks ALL the
// (query head, query tile) items of its key-block pair (the pair kernel's two teams take half each), so the grid is the same and there is
// no team sum at the end.  32 fragment reads per 32 MFMAs, four waves at the LDS port instead of eight.  Same products and roundings per
// item; the items of a key block are accumulated in ONE head-major sequence instead of two halves added at the end (low-order bits of
// dK / dV differ from the pair kernel's; every test compares against the oracle / the fixtures).
template <int RD>
__global__ __launch_bounds__(256) void attn_bwd_dkv128_res_kernel(AttnArgs a) {
  constexpr int BQ = 32, NDB = 4, QT = BQ * 256, IMG = 128 * 256;     // 8 KiB per 32-row tile, 32 KiB per 128-key image
  extern __shared__ __attribute__((aligned(16))) char smem[];          // [V image | K image | 2 stages x (Q | dO) | row constants]
  const int l = threadIdx.x & 63, h = l >> 5;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave-uniform: LDS-DMA addresses live in SGPRs
  const int wt = w, tt = threadIdx.x;
  float* rowc = (float*)(smem + 2 * IMG + 4 * QT);                     // [2 stages][lse*log2e (32) | delta (32)]
  // 1-D grid, XCD-aware: the hardware deals consecutive workgroup ids to the 8 XCDs in turn, and the `npair` workgroups of one
  // (batch, KV head) stream the SAME Q / dO rows (G heads x Sq x 512 B).  With a (pair, head, batch) grid those workgroups sat
  // on 8 different XCDs, every L2 saw each tile once and the kernel pulled 4.5x its algorithmic bytes from beyond L2.  Here
  // group g = (b, hkv) lives on XCD g % 8 and its pairs fill that XCD's consecutive slots (needs B * Hkv % 8 == 0, else the
  // plain order).
  const int npair = (((a.Skv + 127) / 128) + 1) / 2, ngroup = a.B * a.Hkv;
  int pair_i, grp;
  if ((ngroup & 7) == 0) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    pair_i = slot % npair;
    grp = (slot / npair) * 8 + xcd;
  } else {
    pair_i = blockIdx.x % npair;
    grp = blockIdx.x / npair;
  }
  const int b = grp / a.Hkv, hkv = grp % a.Hkv;
  const int G = a.Hq / a.Hkv;
  const int nkb = (a.Skv + 127) / 128;
  const int shift = a.Skv - a.Sq;
  const unsigned lds0 = (unsigned)(uintptr_t)LDS_PTR(char, smem);
  const unsigned ring0 = lds0 + 2 * IMG;
  const float sc = a.scale * LOG2E;
  const int nqt = (a.Sq + BQ - 1) / BQ;
  const int64_t do_ss = (int64_t)a.Hq * 128;
  const ImgaBases bases = imga_bases(smem, 0);
  // per-lane bases: key images at this wave's 32 keys (row fragments), this team's ring (row + transposed fragments)
  const char* vrow[2] = {bases.kr[0] + 2048 * 4 * wt, bases.kr[1] + 2048 * 4 * wt};
  const char* ring_r[2] = {bases.kr[0] + 2 * IMG, bases.kr[1] + 2 * IMG};
  const char* ring_t[2] = {bases.vt[0] + 2 * IMG, bases.vt[1] + 2 * IMG};
  // DMA source patterns (image (a)): a 1-KiB piece = 8 rows x 128 B, lane pattern s = (piece >> 1) & 1
  unsigned lk[2], lv[2], lq, ldo;
  imga_lane_patterns(lk, lv, a.k_ss, a.v_ss);
  {
    const int rl = (l >> 2) & 7, cl = 4 * (l >> 5) + ((l & 3) ^ (((wt & 1) << 1) | ((l >> 4) & 1)));   // this wave's ring pieces: 2wt, 2wt+1
    lq = (unsigned)((int64_t)rl * a.q_ss * 2 + cl * 16);
    ldo = (unsigned)((int64_t)rl * do_ss * 2 + cl * 16);
  }

  for (int pass = 0; pass < 2; ++pass) {
    const int kb = pass == 0 ? pair_i : nkb - 1 - pair_i;
    if (pass == 1 && kb <= pair_i) break;                    // odd count: the middle block has no partner
    const int kblk = kb * 128;
    const int k0 = kblk + wt * 32;
    const int ki = k0 + (l & 31);
    {   // V and K images of the 128 keys: 32 pieces each, 8 per wave (four waves)
      const SRsrc rv = rows_rsrc((const bf16*)a.v + b * a.v_sb + hkv * a.v_sh, a.Skv, a.v_ss);
      const SRsrc rk = rows_rsrc((const bf16*)a.k + b * a.k_sb + hkv * a.k_sh, a.Skv, a.k_ss);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int pc = w * 8 + i, prow = kblk + 8 * (pc >> 1);
        lds_dma16(rv, lv[(i >> 1) & 1] + (unsigned)((int64_t)prow * a.v_ss * 2 + (i & 1) * 128), lds0 + pc * 1024);
        lds_dma16(rk, lk[(i >> 1) & 1] + (unsigned)((int64_t)prow * a.k_ss * 2 + (i & 1) * 128), lds0 + IMG + pc * 1024);
      }
    }
    bool kvalid = ki < a.Skv;
    if (kvalid && a.kmask) kvalid = a.kmask[(int64_t)b * a.Skv + ki] != 0;
    f32x16 dk_acc[NDB], dv_acc[NDB];
#pragma unroll
    for (int i = 0; i < NDB; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) { dk_acc[i][r] = 0.f; dv_acc[i][r] = 0.f; }

    int qt0 = 0;
    if (a.causal) qt0 = max(0, kblk - shift) / BQ;
    const int per_head = max(0, nqt - qt0);
    // the group's (query head, query tile) items in head-major order: all of them
    const int total = per_head * G, niter = total;
    constexpr int item0 = 0;
    float rc = 0.f;
    auto issue = [&](int it) {
      const int idx = item0 + it;
      if (idx >= total || (MM_DKV_DIAG & 1)) return;
      const int g = idx / per_head, qb = (qt0 + idx % per_head) * BQ;
      const int hq = hkv * G + g;
      const SRsrc rq = rows_rsrc((const bf16*)a.q + b * a.q_sb + hq * a.q_sh, a.Sq, a.q_ss);
      const SRsrc rdo = rows_rsrc((const bf16*)a.dout + ((int64_t)b * a.Sq * a.Hq + hq) * 128, a.Sq, do_ss);
      const unsigned st = ring0 + (unsigned)((it & 1) * 2 * QT) + (unsigned)(wt * 2) * 1024u;
      const int prow = qb + 8 * wt;                                    // this wave's two pieces of each tile: rows 8wt .. 8wt+7
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        lds_dma16(rq, lq + (unsigned)((int64_t)prow * a.q_ss * 2 + i * 128), st + i * 1024);
        lds_dma16(rdo, ldo + (unsigned)((int64_t)prow * do_ss * 2 + i * 128), st + QT + i * 1024);
      }
      if (tt < 64) {
        const int qq = qb + (tt & 31);
        const int64_t ro = ((int64_t)b * a.Hq + hq) * a.Sq + qq;
        if (tt < 32) rc = qq < a.Sq ? a.lse[ro] * LOG2E : INFINITY;
        else rc = qq < a.Sq ? a.delta[ro] : 0.f;
      }
    };
    if (niter > 0) {
      issue(0);
      if (tt < 64) rowc[tt] = rc;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // the images (and the first tile) have landed ...
    __builtin_amdgcn_s_barrier();                                      // ... in every wave
    bf16x8 kres[8], vres[8];                                           // this wave's 32 keys: K / V rows x 16 d per fragment, the whole pass
#pragma unroll
    for (int ds = 0; ds < 8; ++ds) {
      kres[ds] = *(const bf16x8*)(vrow[ds & 1] + IMG + 512 * (ds >> 1));
      vres[ds] = *(const bf16x8*)(vrow[ds & 1] + 512 * (ds >> 1));
    }
    for (int it = 0; it < niter; ++it) {
      const int qb = (qt0 + (item0 + it) % per_head) * BQ;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (!(MM_DKV_DIAG & 32)) __builtin_amdgcn_s_barrier();
      // The next tile's DMA (4 pieces per wave) is NOT issued here: in front of the S / dP products it shares the LDS path with
      // their 32 fragment reads and costs 100-185 cycles per piece (timing build without the DMA: -25 % kernel time).  It is
      // issued after those products, in front of the VALU-only dS segment (MI355X_MICROARCH.md: 25-60 cycles there).  Its ring
      // slot was last read in iteration it - 1, which every wave of the team left before this barrier.
      bool issued = false;
      if (it + 1 < niter) { issue(it + 1); issued = true; }
      const int so = (it & 1) * 2 * QT;
      const float* rcs = rowc + (it & 1) * 64;
      if (item0 + it < total && !(a.causal && k0 > qb + BQ - 1 + shift)) {
        // fragment j of the 32 that come from LDS: j < 16: k-step j >> 1, kind j & 1 (0: Q rows, 1: dO rows);
        //                       j >= 16: d block (j - 16) >> 2, 16-query step ((j - 16) >> 1) & 1, kind (j & 1) (0: dO^T, 1: Q^T)
        auto frag = [&](int j) -> bf16x8 {
          if (j < 16) {
            const int ds = j >> 1, a2 = ds & 1, off = 512 * (ds >> 1);
            if ((j & 1) == 0) return *(const bf16x8*)(ring_r[a2] + so + off);
            return *(const bf16x8*)(ring_r[a2] + so + QT + off);
          }
          const int jj = j - 16, db = jj >> 2, s16 = (jj >> 1) & 1, kind = jj & 1;
          const int off = so + (kind == 0 ? QT : 0) + 4096 * s16 + 512 * db;
          const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, ring_t[0] + off));
          const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, ring_t[1] + off));
          bf16x8 o;
          o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3];
          o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = hi[3];
          return o;
        };
        // ONE fresh fragment per MFMA (Q rows for S^T = K Q^T, dO rows for dP^T = V dO^T; K / V rows are resident): RD slots, LA in flight
        constexpr int LA = RD == 8 ? 6 : 3;
        f32x16 s_acc, dp_acc;
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        bf16x8 fr[RD];                                                 // ring: fragment j lives in fr[j % RD]
#pragma unroll
        for (int j = 0; j < LA; ++j) fr[j] = frag(j);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int m = 0; m < 16; ++m) {                                 // MFMA m uses fragment m and the resident fragment of k-step m >> 1
          if (m & 1) dp_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[m % RD], vres[m >> 1], m > 1 ? dp_acc : zero16, 0, 0, 0);
          else s_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[m % RD], kres[m >> 1], m > 1 ? s_acc : zero16, 0, 0, 0);
          if (m + LA < 20) fr[(m + LA) % RD] = frag(m + LA);           // (the last steps fetch the first four transposed fragments, 16 .. 19)
          __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
        // row constants of the lane's 16 query rows: rows 8g + 4h + (0..3), g = 0..3 -> 4 + 4 vectors of 16 bytes
        f32x4 lsev[4], dltv[4];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          lsev[g4] = *(const f32x4*)(rcs + 8 * g4 + 4 * h);
          dltv[g4] = *(const f32x4*)(rcs + 32 + 8 * g4 + 4 * h);
        }
        bf16x8 pf[2], dsf[2];
        if (MM_DKV_DIAG & 2) {
#pragma unroll
          for (int r = 0; r < 16; ++r) { pf[r >> 3][r & 7] = (bf16)s_acc[r]; dsf[r >> 3][r & 7] = (bf16)(dp_acc[r] + lsev[r >> 2][r & 3] + dltv[r >> 2][r & 3]); }
        } else {
        const bool need_mask = (__ballot(kvalid) != ~0ull) || (a.causal && (k0 + 31) > (qb + shift));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ql = acc_row(r, h);
          float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s_acc[r], sc, -lsev[r >> 2][r & 3]));
          if (need_mask) {
            bool ok = kvalid;
            if (a.causal) ok = ok && ki <= (qb + ql + shift);
            p = ok ? p : 0.f;
          }
          const float dsv = p * (dp_acc[r] - dltv[r >> 2][r & 3]) * a.scale;
          pf[r >> 3][r & 7] = (bf16)p;
          dsf[r >> 3][r & 7] = (bf16)dsv;
        }
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int m = 0; m < 16; ++m) {                                 // fragment 16 + m: d block m >> 2, query step (m >> 1) & 1
          const int db = m >> 2, s16 = (m >> 1) & 1;
          if (m & 1) dk_acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[(16 + m) % RD], dsf[s16], dk_acc[db], 0, 0, 0);
          else dv_acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[(16 + m) % RD], pf[s16], dv_acc[db], 0, 0, 0);
          if (16 + m + 4 < 32) fr[(16 + m + 4) % RD] = frag(16 + m + 4);
          __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
      }
      if (!issued && it + 1 < niter) issue(it + 1);                     // a wave that skipped the tile (or the early-issue build)
      if (it + 1 < niter && tt < 64) rowc[((it + 1) & 1) * 64 + tt] = rc;
    }
    // ---- every wave stores the dK / dV rows of its 32 keys
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                                   // every wave has left the ring and the images (the next pass overwrites them)
    {
      if (ki < a.Skv) {
        bf16* dkrow = (bf16*)a.dk + b * a.k_sb + hkv * a.k_sh + (int64_t)ki * a.k_ss;
        bf16* dvrow = (bf16*)a.dv + b * a.v_sb + hkv * a.v_sh + (int64_t)ki * a.v_ss;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            bf16x4 ok_, ov_;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              ok_[e] = (bf16)dk_acc[db][rg * 4 + e];
              ov_[e] = (bf16)dv_acc[db][rg * 4 + e];
            }
            *(bf16x4*)(dkrow + db * 32 + 8 * rg + 4 * h) = ok_;
            *(bf16x4*)(dvrow + db * 32 + 8 * rg + 4 * h) = ov_;
          }
      }
    }
    __syncthreads();                                                   // `red` is free again before the next pass's DMA
  }
}

// The same with the items software-pipelined inside the wave (mm_set_option "attn_dkv_res" = 2): iteration `it` runs the S^T / dP^T products
// of item it + 1 INTERLEAVED with the exp / dS arithmetic of item it (one MFMA, then one of the 16 accumulator elements: the vector ALU works in
// the matrix pipe's shadow), then the dV / dK products of item it.  The ring has three stages (item it for the transposed fragments, it + 1 for
// the row fragments, it + 2 landing); fully masked (head, tile) items are not skipped (their p is 0: the sums are unchanged).
// 4 bytes per lane straight into LDS (lane l lands at lds_addr + 4 l): the row constants of the pipelined dK/dV kernel travel like its tiles,
// so that no compiler-visible load (and the s_waitcnt vmcnt(0) that would come with its use) sits inside the item loop
__device__ __forceinline__ void lds_dma4(const SRsrc& r, unsigned voff, unsigned lds_addr) {
  u32x4 d = {r.w0, r.w1, r.w2, r.w3};
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 4\n\t"
      "buffer_load_dword %1, %3, 0 offen lds\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(lds_addr), "s"(d)
      : "memory");
}

template <int RD>
__global__ __launch_bounds__(256) void attn_bwd_dkv128_resp_kernel(AttnArgs a) {
  constexpr int BQ = 32, NDB = 4, QT = BQ * 256, IMG = 128 * 256;     // 8 KiB per 32-row tile, 32 KiB per 128-key image
  constexpr int NST = 5, LAT = 4;                                      // ring stages; tiles requested ahead (tile it + LAT during iteration it)
  extern __shared__ __attribute__((aligned(16))) char smem[];          // [V image | K image | NST stages x (Q | dO) | NST x row constants (lse 32, -, delta 32, -)]
  const int l = threadIdx.x & 63, h = l >> 5;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave-uniform: LDS-DMA addresses live in SGPRs
  const int wt = w, tt = threadIdx.x;
  const unsigned lds0_early = (unsigned)(uintptr_t)LDS_PTR(char, smem);
  float* rowc = (float*)(smem + 2 * IMG + NST * 2 * QT);               // [NST stages][lse (32) | unused (32) | delta (32) | unused (32)] floats
  const unsigned rowc_lds = lds0_early + 2 * IMG + NST * 2 * QT;
  // 1-D grid, XCD-aware: the hardware deals consecutive workgroup ids to the 8 XCDs in turn, and the `npair` workgroups of one
  // (batch, KV head) stream the SAME Q / dO rows (G heads x Sq x 512 B).  With a (pair, head, batch) grid those workgroups sat
  // on 8 different XCDs, every L2 saw each tile once and the kernel pulled 4.5x its algorithmic bytes from beyond L2.  Here
  // group g = (b, hkv) lives on XCD g % 8 and its pairs fill that XCD's consecutive slots (needs B * Hkv % 8 == 0, else the
  // plain order).
  const int npair = (((a.Skv + 127) / 128) + 1) / 2, ngroup = a.B * a.Hkv;
  int pair_i, grp;
  if ((ngroup & 7) == 0) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    pair_i = slot % npair;
    grp = (slot / npair) * 8 + xcd;
  } else {
    pair_i = blockIdx.x % npair;
    grp = blockIdx.x / npair;
  }
  const int b = grp / a.Hkv, hkv = grp % a.Hkv;
  const int G = a.Hq / a.Hkv;
  const int nkb = (a.Skv + 127) / 128;
  const int shift = a.Skv - a.Sq;
  const unsigned lds0 = (unsigned)(uintptr_t)LDS_PTR(char, smem);
  const unsigned ring0 = lds0 + 2 * IMG;
  const float sc = a.scale * LOG2E;
  const int nqt = (a.Sq + BQ - 1) / BQ;
  const int64_t do_ss = (int64_t)a.Hq * 128;
  const ImgaBases bases = imga_bases(smem, 0);
  // per-lane bases: key images at this wave's 32 keys (row fragments), this team's ring (row + transposed fragments)
  const char* vrow[2] = {bases.kr[0] + 2048 * 4 * wt, bases.kr[1] + 2048 * 4 * wt};
  const char* ring_r[2] = {bases.kr[0] + 2 * IMG, bases.kr[1] + 2 * IMG};
  const char* ring_t[2] = {bases.vt[0] + 2 * IMG, bases.vt[1] + 2 * IMG};
  // DMA source patterns (image (a)): a 1-KiB piece = 8 rows x 128 B, lane pattern s = (piece >> 1) & 1
  unsigned lk[2], lv[2], lq, ldo;
  imga_lane_patterns(lk, lv, a.k_ss, a.v_ss);
  {
    const int rl = (l >> 2) & 7, cl = 4 * (l >> 5) + ((l & 3) ^ (((wt & 1) << 1) | ((l >> 4) & 1)));   // this wave's ring pieces: 2wt, 2wt+1
    lq = (unsigned)((int64_t)rl * a.q_ss * 2 + cl * 16);
    ldo = (unsigned)((int64_t)rl * do_ss * 2 + cl * 16);
  }

  for (int pass = 0; pass < 2; ++pass) {
    const int kb = pass == 0 ? pair_i : nkb - 1 - pair_i;
    if (pass == 1 && kb <= pair_i) break;                    // odd count: the middle block has no partner
    const int kblk = kb * 128;
    const int k0 = kblk + wt * 32;
    const int ki = k0 + (l & 31);
    {   // V and K images of the 128 keys: 32 pieces each, 8 per wave (four waves)
      const SRsrc rv = rows_rsrc((const bf16*)a.v + b * a.v_sb + hkv * a.v_sh, a.Skv, a.v_ss);
      const SRsrc rk = rows_rsrc((const bf16*)a.k + b * a.k_sb + hkv * a.k_sh, a.Skv, a.k_ss);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int pc = w * 8 + i, prow = kblk + 8 * (pc >> 1);
        lds_dma16(rv, lv[(i >> 1) & 1] + (unsigned)((int64_t)prow * a.v_ss * 2 + (i & 1) * 128), lds0 + pc * 1024);
        lds_dma16(rk, lk[(i >> 1) & 1] + (unsigned)((int64_t)prow * a.k_ss * 2 + (i & 1) * 128), lds0 + IMG + pc * 1024);
      }
    }
    bool kvalid = ki < a.Skv;
    if (kvalid && a.kmask) kvalid = a.kmask[(int64_t)b * a.Skv + ki] != 0;
    f32x16 dk_acc[NDB], dv_acc[NDB];
#pragma unroll
    for (int i = 0; i < NDB; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) { dk_acc[i][r] = 0.f; dv_acc[i][r] = 0.f; }

    int qt0 = 0;
    if (a.causal) qt0 = max(0, kblk - shift) / BQ;
    const int per_head = max(0, nqt - qt0);
    // the group's (query head, query tile) items in head-major order: all of them
    const int total = per_head * G, niter = total;
    constexpr int item0 = 0;
    // (head, tile) of the next item to request / to compute: counters instead of idx / per_head and idx % per_head -- two scalar divisions per
    // item are ~100 instructions that a wave alone on its SIMD cannot hide
    int is_g = 0, is_q = 0, cq = 0;
    auto issue = [&](int it) {
      const int idx = item0 + it;
      if (idx >= total || (MM_DKV_DIAG & 1)) return;
      const int g = is_g, qb = (qt0 + is_q) * BQ;
      if (++is_q == per_head) { is_q = 0; ++is_g; }
      const int hq = hkv * G + g;
      const SRsrc rq = rows_rsrc((const bf16*)a.q + b * a.q_sb + hq * a.q_sh, a.Sq, a.q_ss);
      const SRsrc rdo = rows_rsrc((const bf16*)a.dout + ((int64_t)b * a.Sq * a.Hq + hq) * 128, a.Sq, do_ss);
      const unsigned st = ring0 + (unsigned)((it % NST) * 2 * QT) + (unsigned)(wt * 2) * 1024u;
      const int prow = qb + 8 * wt;                                    // this wave's two pieces of each tile: rows 8wt .. 8wt+7
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        lds_dma16(rq, lq + (unsigned)((int64_t)prow * a.q_ss * 2 + i * 128), st + i * 1024);
        lds_dma16(rdo, ldo + (unsigned)((int64_t)prow * do_ss * 2 + i * 128), st + QT + i * 1024);
      }
      if (w == 0) {                                                    // the tile's 32 lse and 32 delta values: two 4-byte-per-lane DMAs (lanes >= 32 out of range)
        const SRsrc rl = make_srsrc((const float*)a.lse + ((int64_t)b * a.Hq + hq) * a.Sq, (int64_t)a.Sq * 4);      // rows >= Sq read 0 (their Q / dO rows are 0 too)
        const SRsrc rd = make_srsrc((const float*)a.delta + ((int64_t)b * a.Hq + hq) * a.Sq, (int64_t)a.Sq * 4);
        const unsigned vo = l < 32 ? (unsigned)(qb + l) * 4u : 0x80000000u;
        lds_dma4(rl, vo, rowc_lds + (unsigned)((it % NST) * 512));
        lds_dma4(rd, vo, rowc_lds + (unsigned)((it % NST) * 512 + 256));
      }
    };
#pragma unroll
    for (int j = 0; j < LAT; ++j)
      if (j < niter) issue(j);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // the images and the first tiles have landed ...
    __builtin_amdgcn_s_barrier();                                      // ... in every wave
    bf16x8 kres[8], vres[8];                                           // this wave's 32 keys: K / V rows x 16 d per fragment, the whole pass
#pragma unroll
    for (int ds = 0; ds < 8; ++ds) {
      kres[ds] = *(const bf16x8*)(vrow[ds & 1] + IMG + 512 * (ds >> 1));
      vres[ds] = *(const bf16x8*)(vrow[ds & 1] + 512 * (ds >> 1));
    }
    constexpr int LA = RD == 8 ? 6 : 3;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 s_cur = zero16, dp_cur = zero16;
    // row fragment j (< 16) of the tile in ring stage offset `sr`: k-step j >> 1, kind j & 1 (0: Q rows, 1: dO rows);
    // transposed fragment j (>= 16) of the tile at `stt`: d block (j - 16) >> 2, 16-query step ((j - 16) >> 1) & 1, kind j & 1 (0: dO^T, 1: Q^T)
    // (the stage offsets are added to the four lane bases ONCE per item -- rr0 / rr1 / rt0 / rt1 below --, so that every read is base + constant:
    // with `ring + stage + offset` per read the compiler spent two vector adds on each of the 32 transposed reads of an item)
    const char *rr0 = ring_r[0], *rr1 = ring_r[1], *rt0 = ring_t[0], *rt1 = ring_t[1];
    auto frag = [&](int j, int, int) -> bf16x8 {
      if (j < 16) {
        const int ds = j >> 1, off = 512 * (ds >> 1);
        const char* rb = (ds & 1) ? rr1 : rr0;
        if ((j & 1) == 0) return *(const bf16x8*)(rb + off);
        return *(const bf16x8*)(rb + QT + off);
      }
      const int jj = j - 16, db = jj >> 2, s16 = (jj >> 1) & 1, kind = jj & 1;
      const int off = (kind == 0 ? QT : 0) + 4096 * s16 + 512 * db;
      const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, rt0 + off));
      const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, rt1 + off));
      bf16x8 o;
      o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3];
      o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = hi[3];
      return o;
    };
    bf16x8 fr[RD];                                                     // ring: fragment j lives in fr[j % RD]
    if (niter > 0) {                                                   // pipeline prologue: S^T / dP^T of item 0
#pragma unroll
      for (int j = 0; j < LA; ++j) fr[j] = frag(j, 0, 0);
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        if (m & 1) dp_cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[m % RD], vres[m >> 1], m > 1 ? dp_cur : zero16, 0, 0, 0);
        else s_cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[m % RD], kres[m >> 1], m > 1 ? s_cur : zero16, 0, 0, 0);
        if (m + LA < 16) fr[(m + LA) % RD] = frag(m + LA, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    for (int it = 0; it < niter; ++it) {
      const int qb = (qt0 + cq) * BQ;
      if (++cq == per_head) cq = 0;
      // tile it + 1 has landed: what is younger than it in this wave's queue are tiles it + 2 .. it + LAT - 1 (4 DMA pieces each, 6 in wave 0)
      {
        const int young = min(niter, it + LAT) - min(niter, it + 2);
        if (young >= 2) { if (w == 0) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
        else if (young == 1) { if (w == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();                                    // ... in every wave, and every wave has left iteration it - 1 (stage (it + LAT) % NST is free)
      if (it + LAT < niter) issue(it + LAT);
      const bool nxt = it + 1 < niter;
      const int so_c = (it % NST) * 2 * QT, so_n = ((it + 1) % NST) * 2 * QT;
      rr0 = ring_r[0] + so_n; rr1 = ring_r[1] + so_n; rt0 = ring_t[0] + so_c; rt1 = ring_t[1] + so_c;
      const float* rcs = rowc + (it % NST) * 128;
      // row constants of the lane's 16 query rows: rows 8g + 4h + (0..3), g = 0..3 -> 4 + 4 vectors of 16 bytes (read four at a time inside the
      // element loop instead: 0.683 vs 0.670 ms, the reads then sit on the critical path)
      f32x4 lsev[4], dltv[4];
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        lsev[g4] = *(const f32x4*)(rcs + 8 * g4 + 4 * h) * LOG2E;
        dltv[g4] = *(const f32x4*)(rcs + 64 + 8 * g4 + 4 * h);
      }
      f32x16 s_nxt = zero16, dp_nxt = zero16;
      bf16x8 pf[2], dsf[2];
      // fragment sequence of this iteration: 0 .. 15 = row fragments of item it + 1 (when there is one), 16 .. 31 = transposed fragments of item it
      if (nxt) {
#pragma unroll
        for (int j = 0; j < LA; ++j) fr[j] = frag(j, so_n, so_c);
      }
      __builtin_amdgcn_s_setprio(1);
      // (branch-free bodies: a scalar branch per element ends the basic block, and the compiler then waits lgkmcnt(0) -- the whole fragment
      // ring -- in front of every MFMA; the mask is a select, and `nxt` picks one of two instantiations per iteration)
      const bool causal_b = a.causal != 0;
      const bool need_mask = (__ballot(kvalid) != ~0ull) || (a.causal && (k0 + 31) > (qb + shift));     // wave-uniform: most tiles need none
      auto bc = [&](auto nxt_c, auto mask_c) {
        constexpr bool NXT = decltype(nxt_c)::value, MASK = decltype(mask_c)::value;
#pragma unroll
        for (int m = 0; m < 16; ++m) {                                 // one product of item it + 1, then one accumulator element of item it
          if constexpr (NXT) {
            if (m & 1) dp_nxt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[m % RD], vres[m >> 1], m > 1 ? dp_nxt : zero16, 0, 0, 0);
            else s_nxt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[m % RD], kres[m >> 1], m > 1 ? s_nxt : zero16, 0, 0, 0);
            if (m + LA < 16) fr[(m + LA) % RD] = frag(m + LA, so_n, so_c);
          }
          if (m + LA >= 16 && m + LA < 16 + LA) fr[(m + LA) % RD] = frag(m + LA, so_n, so_c);   // the first LA transposed fragments of item it
          {
            const int r = m;
            float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s_cur[r], sc, -lsev[r >> 2][r & 3]));
            if constexpr (MASK) {
              const int ql = acc_row(r, h);
              const bool ok = kvalid && (!causal_b || ki <= (qb + ql + shift));
              p = ok ? p : 0.f;
            }
            const float dsv = p * (dp_cur[r] - dltv[r >> 2][r & 3]) * a.scale;
            pf[r >> 3][r & 7] = (bf16)p;
            dsf[r >> 3][r & 7] = (bf16)dsv;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      if (nxt) { if (need_mask) bc(std::true_type{}, std::true_type{}); else bc(std::true_type{}, std::false_type{}); }
      else { if (need_mask) bc(std::false_type{}, std::true_type{}); else bc(std::false_type{}, std::false_type{}); }
#pragma unroll
      for (int m = 0; m < 16; ++m) {                                   // fragment 16 + m: d block m >> 2, query step (m >> 1) & 1
        const int db = m >> 2, s16 = (m >> 1) & 1;
        if (m & 1) dk_acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[(16 + m) % RD], dsf[s16], dk_acc[db], 0, 0, 0);
        else dv_acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[(16 + m) % RD], pf[s16], dv_acc[db], 0, 0, 0);
        if (16 + m + LA < 32) fr[(16 + m + LA) % RD] = frag(16 + m + LA, so_n, so_c);      // (LA = 6 ahead: a transposed fragment is two LDS reads, ~200 cycles)
        __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_s_setprio(0);
      s_cur = s_nxt;
      dp_cur = dp_nxt;
    }
    // ---- every wave stores the dK / dV rows of its 32 keys
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                                   // every wave has left the ring and the images (the next pass overwrites them)
    {
      if (ki < a.Skv) {
        bf16* dkrow = (bf16*)a.dk + b * a.k_sb + hkv * a.k_sh + (int64_t)ki * a.k_ss;
        bf16* dvrow = (bf16*)a.dv + b * a.v_sb + hkv * a.v_sh + (int64_t)ki * a.v_ss;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            bf16x4 ok_, ov_;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              ok_[e] = (bf16)dk_acc[db][rg * 4 + e];
              ov_[e] = (bf16)dv_acc[db][rg * 4 + e];
            }
            *(bf16x4*)(dkrow + db * 32 + 8 * rg + 4 * h) = ok_;
            *(bf16x4*)(dvrow + db * 32 + 8 * rg + 4 * h) = ov_;
          }
      }
    }
    __syncthreads();                                                   // `red` is free again before the next pass's DMA
  }
}

// dQ for D = 128 with prefetched fragments (same treatment as attn_fwd128p_kernel: image (a), base + constant addressing, a
// 4-deep fragment ring; same arithmetic and rounding as attn_bwd_dq128_kernel, bit-identical results).  Per 32-key step the
// wave reads 16 row fragments (K for S^T = K.Q^T, V for dP^T = V.dO^T) and 8 transposed K fragments (dQ^T += K^T.dS^T) -- all
// 24 are one ring sequence, so the transposed fragments travel under the exponentials.
__global__ __launch_bounds__(512, 2) void attn_bwd_dq128p_kernel(AttnArgs a) {
  constexpr int BKV = 64, NDS = 8, NDB = 4, TILE = BKV * 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 stages][K 16 KiB | V 16 KiB]
  const int l = threadIdx.x & 63, h = l >> 5;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int qblk, b, hq;
  attn_work_item(blockIdx.x, (a.Sq + 255) / 256, a.B, a.Hq, a.Hkv, a.causal != 0, qblk, b, hq);
  const int hkv = hq / (a.Hq / a.Hkv);
  const int q0 = qblk * 256 + w * 32;
  const int qi = q0 + (l & 31);
  const int shift = a.Skv - a.Sq;
  const bf16* Q = (const bf16*)a.q + b * a.q_sb + hq * a.q_sh;
  const bf16* dO = (const bf16*)a.dout + ((int64_t)b * a.Sq * a.Hq + hq) * 128;
  const SRsrc rk = rows_rsrc((const bf16*)a.k + b * a.k_sb + hkv * a.k_sh, a.Skv, a.k_ss);
  const SRsrc rv = rows_rsrc((const bf16*)a.v + b * a.v_sb + hkv * a.v_sh, a.Skv, a.v_ss);
  const unsigned lds0 = (unsigned)(uintptr_t)LDS_PTR(char, smem);
  const ImgaBases bases = imga_bases(smem, 0);                   // kr: row fragments, vt: transposed fragments, both from tile offset 0

  bf16x8 qf[NDS], dof[NDS];
  {
    const bf16* qrow = qi < a.Sq ? Q + (int64_t)qi * a.q_ss : nullptr;
    const bf16* drow = qi < a.Sq ? dO + (int64_t)qi * a.Hq * 128 : nullptr;
#pragma unroll
    for (int ds = 0; ds < NDS; ++ds) {
      qf[ds] = row_frag_global(qrow, ds);
      dof[ds] = row_frag_global(drow, ds);
    }
  }
  const float sc = a.scale * LOG2E;
  float lse2 = INFINITY, dlt = 0.f;
  if (qi < a.Sq) {
    lse2 = a.lse[((int64_t)b * a.Hq + hq) * a.Sq + qi] * LOG2E;
    dlt = a.delta[((int64_t)b * a.Hq + hq) * a.Sq + qi];
  }
  f32x16 dq_acc[NDB];
#pragma unroll
  for (int i = 0; i < NDB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq_acc[i][r] = 0.f;

  int ntiles = (a.Skv + BKV - 1) / BKV;
  if (a.causal) {
    const int qmax = min(a.Sq - 1, qblk * 256 + 255) + shift;
    ntiles = qmax < 0 ? 0 : min(ntiles, qmax / BKV + 1);
  }
  unsigned lk[2], lv[2];
  imga_lane_patterns(lk, lv, a.k_ss, a.v_ss);
  constexpr int NST = 3;                                         // 3-stage ring, counted vmcnt: see attn_fwd128p_kernel
  if (ntiles > 0) imga_issue_kv(w, lds0, 0, rk, rv, lk, lv, a.k_ss, a.v_ss, TILE);
  if (ntiles > 1) imga_issue_kv(w, lds0 + 2 * TILE, BKV, rk, rv, lk, lv, a.k_ss, a.v_ss, TILE);
  for (int t = 0; t < ntiles; ++t) {
    const int kv0 = t * BKV;
    if (t + 1 < ntiles) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t + 2 < ntiles) imga_issue_kv(w, lds0 + (unsigned)(((t + 2) % NST) * 2 * TILE), (t + 2) * BKV, rk, rv, lk, lv, a.k_ss, a.v_ss, TILE);
    if (q0 >= a.Sq || (a.causal && kv0 > q0 + 31 + shift)) continue;
    const int so = (t % NST) * 2 * TILE;
    bool kvalid = (kv0 + l) < a.Skv;
    if (kvalid && a.kmask) kvalid = a.kmask[(int64_t)b * a.Skv + kv0 + l] != 0;
    const unsigned long long kbits = __ballot(kvalid);
    const bool need_mask = (kbits != ~0ull) || (a.causal && (kv0 + BKV - 1) > (q0 + shift));
    // fragment j of the 24 a 32-key step uses: j < 16: row fragment (j even: K, odd: V) of k-step j >> 1; j >= 16: transposed
    // K fragment of d block (j - 16) >> 1, 16-key step (j - 16) & 1.  The two 32-key steps of a tile are a ROLLED loop (the
    // step's LDS offset is folded into four base registers): unrolled, the kernel spilt 34 registers into the tile loop.
    constexpr int RD = 4;
#pragma unroll 1
    for (int kb = 0; kb < 2; ++kb) {
      const char* kr0 = bases.kr[0] + so + 8192 * kb;
      const char* kr1 = bases.kr[1] + so + 8192 * kb;
      const char* vt0 = bases.vt[0] + so + 8192 * kb;
      const char* vt1 = bases.vt[1] + so + 8192 * kb;
      auto frag = [&](int j) -> bf16x8 {
        if (j < 16) return *(const bf16x8*)(((j >> 1) & 1 ? kr1 : kr0) + (j & 1) * TILE + 512 * (j >> 2));
        const int off = 4096 * ((j - 16) & 1) + 512 * ((j - 16) >> 1);
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, vt0 + off));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, vt1 + off));
        bf16x8 o;
        o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3];
        o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = hi[3];
        return o;
      };
      f32x16 s_acc, dp_acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s_acc[r] = 0.f; dp_acc[r] = 0.f; }
      bf16x8 fr[RD];
#pragma unroll
      for (int j = 0; j < RD - 1; ++j) fr[j] = frag(j);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        fr[(j + RD - 1) % RD] = frag(j + RD - 1);
        if (j & 1) dp_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[j % RD], dof[j >> 1], dp_acc, 0, 0, 0);
        else s_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[j % RD], qf[j >> 1], s_acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_s_setprio(0);
      bf16x8 dsf[2];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s_acc[r], sc, -lse2));
        if (need_mask) {
          const int kl = kb * 32 + acc_row(r, h);
          bool ok = (kbits >> kl) & 1ull;
          if (a.causal) ok = ok && (kv0 + kl) <= (qi + shift);
          p = ok ? p : 0.f;
        }
        dsf[r >> 3][r & 7] = (bf16)(p * (dp_acc[r] - dlt) * a.scale);
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int j = 16; j < 24; ++j) {
        if (j + RD - 1 < 24) fr[(j + RD - 1) % RD] = frag(j + RD - 1);
        dq_acc[(j - 16) >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[j % RD], dsf[(j - 16) & 1], dq_acc[(j - 16) >> 1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_s_setprio(0);
    }
  }
  if (qi < a.Sq) {
    bf16* drow = (bf16*)a.dq + b * a.q_sb + hq * a.q_sh + (int64_t)qi * a.q_ss;
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16)dq_acc[db][rg * 4 + e];
        *(bf16x4*)(drow + db * 32 + 8 * rg + 4 * h) = o;
      }
  }
}

// dQ for D = 128: same shape as the forward fast path (8 waves, 256 query rows, K/V tiles by LDS-DMA, one barrier per
// tile).  The single K image serves the row fragments of S^T = K.Q^T and the transposed fragments of dQ^T += K^T.dS^T.
template <int INW>
__global__ __launch_bounds__(512, 2) void attn_bwd_dq128_kernel(AttnArgs a) {
  constexpr int BKV = 64, NDS = 8, NDB = 4, TILE = BKV * 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 stages][K 16 KiB | V 16 KiB]
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6, h = l >> 5;
  int qblk, b, hq;
  attn_work_item(blockIdx.x, (a.Sq + 255) / 256, a.B, a.Hq, a.Hkv, a.causal != 0, qblk, b, hq);
  const int hkv = hq / (a.Hq / a.Hkv);
  const int q0 = qblk * 256 + w * 32;
  const int qi = q0 + (l & 31);
  const int shift = a.Skv - a.Sq;
  const bf16* Q = (const bf16*)a.q + b * a.q_sb + hq * a.q_sh;
  const bf16* dO = (const bf16*)a.dout + ((int64_t)b * a.Sq * a.Hq + hq) * 128;
  const SRsrc rk = rows_rsrc((const bf16*)a.k + b * a.k_sb + hkv * a.k_sh, a.Skv, a.k_ss);
  const SRsrc rv = rows_rsrc((const bf16*)a.v + b * a.v_sb + hkv * a.v_sh, a.Skv, a.v_ss);
  const unsigned lds0 = (unsigned)(uintptr_t)LDS_PTR(char, smem);

  bf16x8 qf[NDS], dof[NDS];
  {
    const bf16* qrow = qi < a.Sq ? Q + (int64_t)qi * a.q_ss : nullptr;
    const bf16* drow = qi < a.Sq ? dO + (int64_t)qi * a.Hq * 128 : nullptr;
#pragma unroll
    for (int ds = 0; ds < NDS; ++ds) {
      qf[ds] = row_frag_global(qrow, ds);
      dof[ds] = row_frag_global(drow, ds);
    }
  }
  const float sc = a.scale * LOG2E;
  float lse2 = INFINITY, dlt = 0.f;
  if (qi < a.Sq) {
    lse2 = a.lse[((int64_t)b * a.Hq + hq) * a.Sq + qi] * LOG2E;
    dlt = a.delta[((int64_t)b * a.Hq + hq) * a.Sq + qi];
  }
  f32x16 dq_acc[NDB];
#pragma unroll
  for (int i = 0; i < NDB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq_acc[i][r] = 0.f;

  int ntiles = (a.Skv + BKV - 1) / BKV;
  if (a.causal) {
    const int qmax = min(a.Sq - 1, qblk * 256 + 255) + shift;
    ntiles = qmax < 0 ? 0 : min(ntiles, qmax / BKV + 1);
  }
  auto issue = [&](int t) {
    const unsigned st = lds0 + (unsigned)((t & 1) * 2 * TILE);
    imgb_dma<BKV, INW>(st, rk, a.k_ss, t * BKV);
    imgb_dma<BKV, INW>(st + TILE, rv, a.v_ss, t * BKV);
  };
  if (ntiles > 0) issue(0);
  for (int t = 0; t < ntiles; ++t) {
    const int kv0 = t * BKV;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t + 1 < ntiles) issue(t + 1);
    if (q0 >= a.Sq || (a.causal && kv0 > q0 + 31 + shift)) continue;
    const char* Kt = smem + (t & 1) * 2 * TILE;
    const char* Vt = Kt + TILE;
    bool kvalid = (kv0 + l) < a.Skv;
    if (kvalid && a.kmask) kvalid = a.kmask[(int64_t)b * a.Skv + kv0 + l] != 0;
    const unsigned long long kbits = __ballot(kvalid);
    const bool need_mask = (kbits != ~0ull) || (a.causal && (kv0 + BKV - 1) > (q0 + shift));
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x16 s_acc, dp_acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s_acc[r] = 0.f; dp_acc[r] = 0.f; }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ds = 0; ds < NDS; ++ds) {
        s_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(imgb_rowfrag(Kt, kb * 32, ds), qf[ds], s_acc, 0, 0, 0);
        dp_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(imgb_rowfrag(Vt, kb * 32, ds), dof[ds], dp_acc, 0, 0, 0);
      }
      __builtin_amdgcn_s_setprio(0);
      bf16x8 dsf[2];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s_acc[r], sc, -lse2));
        if (need_mask) {
          const int kl = kb * 32 + acc_row(r, h);
          bool ok = (kbits >> kl) & 1ull;
          if (a.causal) ok = ok && (kv0 + kl) <= (qi + shift);
          p = ok ? p : 0.f;
        }
        dsf[r >> 3][r & 7] = (bf16)(p * (dp_acc[r] - dlt) * a.scale);
      }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int s = 0; s < 2; ++s)
          dq_acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(imgb_tfrag(Kt, db, kb * 32 + s * 16), dsf[s], dq_acc[db], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
  }
  if (qi < a.Sq) {
    bf16* drow = (bf16*)a.dq + b * a.q_sb + hq * a.q_sh + (int64_t)qi * a.q_ss;
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16)dq_acc[db][rg * 4 + e];
        *(bf16x4*)(drow + db * 32 + 8 * rg + 4 * h) = o;
      }
  }
}

// ============================================================================================================
// KV-cache decode (ONE query token per sequence; reference model.py:595-602 with past_key_values).  The step reads the
// whole K and V cache once -- 2*B*Skv*Hkv*D*2 bytes, ~34 MB per layer at B=4, S=2048 on the 8B decoder -- against a few
// MFLOP, so it is laid out as an HBM stream: no MFMA, no LDS staging.  D/8 lanes share one key row (16 bytes each, one
// coalesced 256-B row per 16 lanes at D = 128), the G query heads of a key/value head ride on the same K/V bytes, a
// workgroup takes one slice of the keys of one (batch, kv head) with per-lane-group online softmax, and slices are
// merged by a second tiny kernel (flash-decoding split-K), all in a fixed order (deterministic).
// Partial record per (b, hq, split): [m (base-2 running max), l (sum), o[D]] in f32.
// ============================================================================================================
struct DecodeArgs {
  const void *q, *k, *v;
  int B, Skv, Hq, Hkv;
  int64_t q_sb, q_sh, k_sb, k_ss, k_sh, v_sb, v_ss, v_sh;
  const int64_t* kmask;
  float scale;
  void* out;
  float* ws;
  int nsplit, chunk;
  int* sync;            // [B*Hkv] arrival counters, zero on entry and left zero (NULL: merge in a second launch)
};

// element d of the merged output row from its nsplit partial records [m, l, o[D]] (in slice order: the result does not depend on
// who merges).  The records' loads are issued EIGHT slices at a time before anything is combined: written as one load per loop
// trip the merge was a chain of ~3 * nsplit dependent memory round trips (12.7 us for a few KB; 50 us inside the last-arriving
// workgroup of the single-launch form, whose records come from beyond L2).
template <int D>
__device__ __forceinline__ bf16 decode_merge_one(const float* rec, int nsplit, int d) {
  float mn = -INFINITY;
  for (int s0 = 0; s0 < nsplit; s0 += 8) {
    float mv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) mv[u] = __builtin_nontemporal_load(rec + (int64_t)min(s0 + u, nsplit - 1) * (D + 2));
#pragma unroll
    for (int u = 0; u < 8; ++u) mn = fmaxf(mn, mv[u]);
  }
  float lt = 0.f, ot = 0.f;
  for (int s0 = 0; s0 < nsplit; s0 += 8) {
    float mv[8], lv[8], ov[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float* r = rec + (int64_t)min(s0 + u, nsplit - 1) * (D + 2);
      mv[u] = __builtin_nontemporal_load(r);
      lv[u] = __builtin_nontemporal_load(r + 1);
      ov[u] = __builtin_nontemporal_load(r + 2 + d);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (s0 + u < nsplit) {
        const float al = mn == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mv[u] - mn);
        lt += lv[u] * al;
        ot += ov[u] * al;
      }
    }
  }
  return (bf16)(lt > 0.f ? ot / lt : 0.f);               // no visible key: 0, as in the prefill kernels
}

template <int D, int G>
__global__ __launch_bounds__(256) void attn_decode_partial_kernel(DecodeArgs a) {
  constexpr int LPK = D / 8;            // lanes per key row
  constexpr int KPI = 64 / LPK;         // keys per wave-instruction
  constexpr int U = 4;                  // key rows in flight per lane (x2: K and V).  8 (a whole 86-key slice of the 8B decoder at
                                        // S = 2048 in ONE batch of loads) measured the same 17 us per launch: not what bounds it
  __shared__ float red[4][G][D + 2];
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int sub = l % LPK, grp = l / LPK;
  const int split = blockIdx.x, hkv = blockIdx.y, b = blockIdx.z;
  const bf16* Kb = (const bf16*)a.k + b * a.k_sb + hkv * a.k_sh + sub * 8;
  const bf16* Vb = (const bf16*)a.v + b * a.v_sb + hkv * a.v_sh + sub * 8;
  const int kbeg = split * a.chunk, kend = min(a.Skv, kbeg + a.chunk);
  const float sc = a.scale * LOG2E;

  float qf[G][8], o[G][8], m[G], ls[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const bf16x8 qv = *(const bf16x8*)((const bf16*)a.q + b * a.q_sb + (hkv * G + g) * a.q_sh + sub * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) { qf[g][j] = (float)qv[j] * sc; o[g][j] = 0.f; }
    m[g] = -INFINITY;
    ls[g] = 0.f;
  }
  for (int k0 = kbeg + w * KPI * U; k0 < kend; k0 += 4 * KPI * U) {
    bf16x8 kf[U], vf[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int key = k0 + u * KPI + grp;
      ok[u] = key < kend;
      const int kk = ok[u] ? key : kbeg;                         // clamp: a valid address, result discarded
      kf[u] = *(const bf16x8*)(Kb + (int64_t)kk * a.k_ss);
      vf[u] = *(const bf16x8*)(Vb + (int64_t)kk * a.v_ss);
      if (ok[u] && a.kmask) ok[u] = a.kmask[(int64_t)b * a.Skv + key] != 0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float kx[8], vx[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { kx[j] = (float)kf[u][j]; vx[j] = (float)vf[u][j]; }
#pragma unroll
      for (int g = 0; g < G; ++g) {
        float sv = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) sv = __builtin_fmaf(qf[g][j], kx[j], sv);
#pragma unroll
        for (int off = LPK / 2; off > 0; off >>= 1) sv += __shfl_xor(sv, off, 64);
        if (ok[u]) {                                             // uniform over the LPK lanes of this key
          const float mn = fmaxf(m[g], sv);
          const float al = __builtin_amdgcn_exp2f(m[g] - mn), pv = __builtin_amdgcn_exp2f(sv - mn);
          ls[g] = ls[g] * al + pv;
          m[g] = mn;
#pragma unroll
          for (int j = 0; j < 8; ++j) o[g][j] = __builtin_fmaf(pv, vx[j], o[g][j] * al);
        }
      }
    }
  }
  // merge the KPI lane groups of the wave (same `sub`, different keys), then the 4 waves, in a fixed order
#pragma unroll
  for (int g = 0; g < G; ++g) {
#pragma unroll
    for (int off = LPK; off < 64; off <<= 1) {
      const float mo = __shfl_xor(m[g], off, 64), lo = __shfl_xor(ls[g], off, 64);
      const float mn = fmaxf(m[g], mo);
      const float a0 = mn == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m[g] - mn), a1 = mn == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mo - mn);
      ls[g] = ls[g] * a0 + lo * a1;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[g][j] = o[g][j] * a0 + __shfl_xor(o[g][j], off, 64) * a1;
      m[g] = mn;
    }
    if (grp == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) red[w][g][sub * 8 + j] = o[g][j];
      if (sub == 0) { red[w][g][D] = m[g]; red[w][g][D + 1] = ls[g]; }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < G * D; i += 256) {
    const int g = i / D, d = i % D;
    float mn = -INFINITY;
#pragma unroll
    for (int ww = 0; ww < 4; ++ww) mn = fmaxf(mn, red[ww][g][D]);
    float lt = 0.f, ot = 0.f;
#pragma unroll
    for (int ww = 0; ww < 4; ++ww) {
      const float al = mn == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(red[ww][g][D] - mn);
      lt += red[ww][g][D + 1] * al;
      ot += red[ww][g][d] * al;
    }
    float* rec = a.ws + (((int64_t)b * a.Hq + hkv * G + g) * a.nsplit + split) * (D + 2);
    if (a.sync) {          // read by another workgroup of this launch: write-through (sc1) stores, see the hand-off below
      __hip_atomic_store(rec + 2 + d, ot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (d == 0) {
        __hip_atomic_store(rec, mn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(rec + 1, lt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
      rec[2 + d] = ot;
      if (d == 0) { rec[0] = mn; rec[1] = lt; }
    }
  }
  if (a.sync == nullptr) return;
  // the slice that arrives LAST merges all slices of this (batch, kv head) -- in slice order, so the result does not
  // depend on which one that is -- and leaves the counter at zero for the next step
  // Hand-off per cdna_hip_programming.md Guideline 16 (counter form): the records are stored write-through (sc1), every storing
  // wave drains its stores, ONE lane takes a ticket; the last arriver makes ONE agent-scope acquire (invalidates this CU's L1)
  // before the workgroup reads the other slices' records.  (Round 2 had every thread run __threadfence() twice; an agent-scope
  // release per workgroup -- buffer_wbl2 with an L2 full of dirty lines -- still cost 4x the merge launch it replaced.)
  __shared__ int last;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {      // no agent-scope release: the records went out write-through (sc1) and every wave has drained its stores
    const int old = __hip_atomic_fetch_add(a.sync + b * a.Hkv + hkv, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last = old == a.nsplit - 1;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(a.sync + b * a.Hkv + hkv, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  if (!last) return;
  for (int i = threadIdx.x; i < G * D; i += 256) {
    const int g = i / D, d = i % D;
    const int64_t row = (int64_t)b * a.Hq + hkv * G + g;
    ((bf16*)a.out)[row * D + d] = decode_merge_one<D>(a.ws + row * a.nsplit * (D + 2), a.nsplit, d);
  }
}

// D = 128, G <= 4 query heads per key/value head: the same slice of the same (batch, kv head), with the scores on the MATRIX
// cores.  attn_decode_partial_kernel takes q.k as 8 FMAs per lane plus a 4-step cross-lane sum per (key, head): 64 ds_bpermute
// per 16 keys and wave, and its time did not move between 256 and 1536 workgroups (17 us per launch for 34 MB of cache: a
// throughput bound of its own making).  Here a wave takes 16 keys at a time:
//   S[16 keys x G] = K[16 x 128] Q^T: 4 x mfma_16x16x32 with K rows as the first operand (a lane loads 16 bytes of key l & 15) and
//   the G query rows as the second; lane (g = l & 15, kq = l >> 4) then holds the scores of keys 4 kq .. 4 kq + 3 for head g;
//   block softmax (max over the 16 keys: 4 registers and two cross-row exchanges), running (m, l) per head;
//   P V on the vector ALU with V rows as they lie in memory: lane (sub = l & 15, kq) holds 8 dims of keys 4 kq + u, and the
//   probabilities it needs sit in lane g of ITS OWN 16-lane row: one DPP row broadcast each (no LDS traffic).
// Same record format (m, l, o[D]) per slice: attn_decode_merge_kernel is unchanged.
__device__ __forceinline__ float row_bcast(float x, int lane_in_row) {          // value of lane `lane_in_row` of this lane's 16-lane row
  return __shfl(x, (int)(threadIdx.x & 48u) | lane_in_row, 64);
}
template <int N>
__device__ __forceinline__ float row_bcast_dpp(float x) {                        // the same as ONE v_mov_b32 dpp row_newbcast:N (gfx90a+)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x150 + N, 0xf, 0xf, false));
}

template <int G>
__global__ __launch_bounds__(256) void attn_decode_partial128_mfma_kernel(DecodeArgs a) {
  constexpr int D = 128;
  static_assert(G >= 1 && G <= 4, "row-broadcast unrolling and the LDS budget below are written for G <= 4");
  __shared__ float red_o[4][4][G][D];          // [wave][kq][head][d]
  __shared__ float red_ml[4][G][2];
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row = l & 15, kq = l >> 4;
  const int split = blockIdx.x, hkv = blockIdx.y, b = blockIdx.z;
  const bf16* Kb = (const bf16*)a.k + b * a.k_sb + hkv * a.k_sh;
  const bf16* Vb = (const bf16*)a.v + b * a.v_sb + hkv * a.v_sh;
  const int kbeg = split * a.chunk, kend = min(a.Skv, kbeg + a.chunk);
  const float sc = a.scale * LOG2E;
  bf16x8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    qf[ks] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    if (row < G) qf[ks] = *(const bf16x8*)((const bf16*)a.q + b * a.q_sb + (hkv * G + row) * a.q_sh + ks * 32 + kq * 8);
  }
  float o[G][8];
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int j = 0; j < 8; ++j) o[g][j] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  bf16x8 kf[4], vf[4];
  auto load_block = [&](int k0) {
    const int kk = min(k0 + row, kend - 1);                       // clamp: a valid address, the score is masked below
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kf[ks] = *(const bf16x8*)(Kb + (int64_t)kk * a.k_ss + ks * 32 + kq * 8);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int kv = min(k0 + 4 * kq + u, kend - 1);
      vf[u] = *(const bf16x8*)(Vb + (int64_t)kv * a.v_ss + row * 8);
    }
  };
  int k0 = kbeg + w * 16;
  if (k0 < kend) load_block(k0);
  for (; k0 < kend; k0 += 64) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[ks], qf[ks], acc, 0, 0, 0);
    float vx[4][8];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < 8; ++j) vx[u][j] = (float)vf[u][j];
    const int kcur = k0;
    if (k0 + 64 < kend) load_block(k0 + 64);                      // the next block streams in under the softmax and the P V sums
    float sv[4], bm = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = kcur + 4 * kq + r;
      bool ok = key < kend;
      if (ok && a.kmask) ok = a.kmask[(int64_t)b * a.Skv + key] != 0;
      sv[r] = ok ? acc[r] * sc : -INFINITY;
      bm = fmaxf(bm, sv[r]);
    }
    bm = fmaxf(bm, __shfl_xor(bm, 16, 64));
    bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
    const float mn = fmaxf(m_run, bm);
    const float alpha = mn == -INFINITY ? 1.f : __builtin_amdgcn_exp2f(m_run - mn);
    float p[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) p[r] = mn == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(sv[r] - mn);
    l_run = l_run * alpha + ((p[0] + p[1]) + (p[2] + p[3]));
    m_run = mn;
#define MM_DEC_HEAD(g)                                                              \
    if constexpr (g < G) {                                                          \
      const float ag = row_bcast_dpp<g>(alpha);                                     \
      const float p0 = row_bcast_dpp<g>(p[0]), p1 = row_bcast_dpp<g>(p[1]);         \
      const float p2 = row_bcast_dpp<g>(p[2]), p3 = row_bcast_dpp<g>(p[3]);         \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) {                               \
        float t = o[g][j] * ag;                                                     \
        t = __builtin_fmaf(p0, vx[0][j], t);                                        \
        t = __builtin_fmaf(p1, vx[1][j], t);                                        \
        t = __builtin_fmaf(p2, vx[2][j], t);                                        \
        o[g][j] = __builtin_fmaf(p3, vx[3][j], t);                                  \
      }                                                                             \
    }
    MM_DEC_HEAD(0) MM_DEC_HEAD(1) MM_DEC_HEAD(2) MM_DEC_HEAD(3)
#undef MM_DEC_HEAD
  }
  // this wave's partial: o per (kq row, head), (m, l) per head (l summed over the 4 rows; m is the same in all of them)
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int j = 0; j < 8; ++j) red_o[w][kq][g][row * 8 + j] = o[g][j];
  float lt = l_run + __shfl_xor(l_run, 16, 64);
  lt += __shfl_xor(lt, 32, 64);
  if (kq == 0 && row < G) { red_ml[w][row][0] = m_run; red_ml[w][row][1] = lt; }
  __syncthreads();
  for (int i = threadIdx.x; i < G * D; i += 256) {
    const int g = i / D, d = i % D;
    float mn = -INFINITY;
#pragma unroll
    for (int ww = 0; ww < 4; ++ww) mn = fmaxf(mn, red_ml[ww][g][0]);
    float ltot = 0.f, ot = 0.f;
#pragma unroll
    for (int ww = 0; ww < 4; ++ww) {
      const float al = mn == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(red_ml[ww][g][0] - mn);
      ltot += red_ml[ww][g][1] * al;
      ot += ((red_o[ww][0][g][d] + red_o[ww][1][g][d]) + (red_o[ww][2][g][d] + red_o[ww][3][g][d])) * al;
    }
    float* rec = a.ws + (((int64_t)b * a.Hq + hkv * G + g) * a.nsplit + split) * (D + 2);
    rec[2 + d] = ot;
    if (d == 0) { rec[0] = mn; rec[1] = ltot; }
  }
}

template <int D>
__global__ void attn_decode_merge_kernel(const float* ws, int nsplit, bf16* out) {
  const int row = blockIdx.x, d = threadIdx.x;          // row = b * Hq + hq
  out[(int64_t)row * D + d] = decode_merge_one<D>(ws + (int64_t)row * nsplit * (D + 2), nsplit, d);
}

int g_attn_fwd_waves = 8;     // waves per workgroup of the D=128 forward (mm_set_option "attn_fwd_waves": 8 or 4)
int g_attn_fwd_pf = 1;        // D=128 forward with prefetched fragments (attn_fwd128p_kernel; mm_set_option "attn_fwd_pf" 0 = the older kernel)
int g_attn_fwd_q = 1;         // D=128 forward with the two waves of a SIMD out of phase (attn_fwd128q_kernel; "attn_fwd_q" 0 = attn_fwd128p_kernel)
int g_attn_dkv_late = 0;      // paired dK/dV kernel: next tile's DMA issued after the S/dP products ("attn_dkv_late" 0 = at the barrier)
int g_attn_dkv_rd = 8;        // fragment ring slots of attn_bwd_dkv128_pairp_kernel ("attn_dkv_rd": 8 or 4)
int g_attn_q_issue = 4;       // waves issuing the K/V DMA in attn_fwd128q_kernel ("attn_q_issue": 4 or 8)
int g_attn_q_rd = 4;          // fragment ring depth of attn_fwd128q_kernel ("attn_q_rd": 4, 6 or 8)
int g_attn_diag = 0;          // AttnArgs::diag ("attn_diag")
int g_attn_q_prio = 1;        // s_setprio policy of the out-of-phase kernels (AttnArgs::prio; "attn_q_prio")
int g_attn_decode_mfma = 1;   // D = 128, G <= 4 decode slices with the scores on MFMA (mm_set_option "attn_decode_mfma"; 0 = attn_decode_partial_kernel)
int g_attn_decode_wgs = 768;  // workgroups mm_attn_decode_splits aims at (~3 per CU; mm_set_option "attn_decode_wgs")
int g_attn_dkv_res = 0;       // dK/dV with resident K / V fragments, one wave per SIMD (mm_set_option "attn_dkv_res")
int g_attn_dkv_pair = 1;      // balanced paired dK/dV kernel (mm_set_option "attn_dkv_pair"; 0 = one key block per workgroup)
int g_attn_issue_waves = 4;   // waves issuing the K/V DMA in the 8-wave D=128 kernels (mm_set_option "attn_issue_waves")

static bool attn_use_v1() {
  static const bool v1 = [] { const char* e = getenv("MM_ATTN_KERNEL"); return e && e[0] == 'v'; }();
  return v1;
}

template <int D>
int launch_bf16_fwd(const AttnArgs& a, hipStream_t s) {
  if (D == 128 && !attn_use_v1() && g_attn_fwd_pf && g_attn_fwd_q && g_attn_fwd_waves == 8 && a.Skv <= 256 * 1024) {
    const size_t lds = 4 * 2 * 64 * 256 + (size_t)((a.Skv + 63) / 64) * 8;       // K/V ring + key-valid bits
    const int64_t nwg = (int64_t)((a.Sq + 255) / 256) * a.Hq * a.B;
    if (nwg > 0x7FFFFFFF) return MM_ERR_ARG;
    auto launch = [&](auto kern) {
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(512), lds, s, a);
    };
    if (g_attn_q_rd == 8) launch(attn_fwd128q_kernel<8, 4>);
    else if (g_attn_q_rd == 6) launch(attn_fwd128q_kernel<6, 4>);
    else if (g_attn_q_issue == 8) launch(attn_fwd128q_kernel<4, 8>);
    else launch(attn_fwd128q_kernel<4, 4>);
    MM_CHECK_LAUNCH();
    return MM_OK;
  }
  if (D == 128 && !attn_use_v1() && g_attn_fwd_pf && g_attn_fwd_waves == 8) {
    const size_t lds = 3 * 2 * 64 * 256;
    const int64_t nwg = (int64_t)((a.Sq + 255) / 256) * a.Hq * a.B;
    if (nwg > 0x7FFFFFFF) return MM_ERR_ARG;
    (void)hipFuncSetAttribute((const void*)attn_fwd128p_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(attn_fwd128p_kernel, dim3((unsigned)nwg), dim3(512), lds, s, a);
    return MM_OK;
  }
  if (D == 128 && !attn_use_v1()) {
    const size_t lds = 2 * 2 * 64 * 256;
    const int qb = g_attn_fwd_waves * 32;
    const int64_t nwg = (int64_t)((a.Sq + qb - 1) / qb) * a.Hq * a.B;     // 1-D grid: attn_work_item orders the blocks
    if (nwg > 0x7FFFFFFF) return MM_ERR_ARG;
    dim3 grid((unsigned)nwg), block(g_attn_fwd_waves * 64);
#define MM_FWD128(...)                                                                                                  \
  do {                                                                                                                  \
    auto kfn = attn_fwd128_kernel<__VA_ARGS__>;                                                                         \
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                  \
    hipLaunchKernelGGL(kfn, grid, block, lds, s, a);                                                                    \
  } while (0)
    if (g_attn_fwd_waves == 4) MM_FWD128(4, 4);
    else if (g_attn_issue_waves == 4) MM_FWD128(4, 8);
    else MM_FWD128(8, 8);
#undef MM_FWD128
    return MM_OK;
  }
  const size_t lds = 2 * 64 * D * 2;
  dim3 grid((a.Sq + 127) / 128, a.Hq, a.B), block(256);
  hipLaunchKernelGGL(attn_fwd_kernel<D>, grid, block, lds, s, a);
  return MM_OK;
}
template <int D>
int launch_bf16_bwd(const AttnArgs& a, hipStream_t s) {
  if (D == 128 && !attn_use_v1()) {
    const size_t lds = 2 * 2 * 64 * 256;
    const int64_t nwg = (int64_t)((a.Sq + 255) / 256) * a.Hq * a.B;     // 1-D grid: attn_work_item orders the blocks
    if (nwg > 0x7FFFFFFF) return MM_ERR_ARG;
    dim3 grid((unsigned)nwg), block(512);
    if (g_attn_fwd_pf) {
      const size_t lds3 = 3 * 2 * 64 * 256;
      (void)hipFuncSetAttribute((const void*)attn_bwd_dq128p_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
      hipLaunchKernelGGL(attn_bwd_dq128p_kernel, grid, block, lds3, s, a);
    } else if (g_attn_issue_waves == 4) {
      (void)hipFuncSetAttribute((const void*)attn_bwd_dq128_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(attn_bwd_dq128_kernel<4>, grid, block, lds, s, a);
    } else {
      (void)hipFuncSetAttribute((const void*)attn_bwd_dq128_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(attn_bwd_dq128_kernel<8>, grid, block, lds, s, a);
    }
  } else {
    const size_t lds = 3 * 64 * D * 2;
    dim3 grid((a.Sq + 127) / 128, a.Hq, a.B), block(256);
    hipLaunchKernelGGL(attn_bwd_dq_kernel<D>, grid, block, lds, s, a);
  }
  if (D == 128 && !attn_use_v1()) {
    if ((((a.Hq / a.Hkv) & 1) == 0 || g_attn_fwd_pf) && g_attn_dkv_pair) {      // the prefetching pair kernel splits any group size
      const int nkb = (a.Skv + 127) / 128;
      dim3 grid((nkb + 1) / 2, a.Hkv, a.B), block(512);
      if (g_attn_fwd_pf) {
        const int64_t nwg = (int64_t)((nkb + 1) / 2) * a.Hkv * a.B;       // 1-D: the kernel deals (pair, head, batch) XCD-aware
        if (nwg > 0x7FFFFFFF) return MM_ERR_ARG;
        grid = dim3((unsigned)nwg);
        const size_t lds = 2 * 128 * 256 + 8 * 32 * 256 + 4 * 64 * sizeof(float);      // V + K images, rings, row constants
        auto launch = [&](auto kern) {
          (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
          hipLaunchKernelGGL(kern, grid, block, lds, s, a);
        };
        if (g_attn_dkv_res) {                                              // K / V fragments resident, four waves (attn_bwd_dkv128_res_kernel)
          if (g_attn_dkv_res == 2) {                                       // ... with the items pipelined inside the wave (three ring stages)
            const size_t lds_p = 2 * 128 * 256 + 5 * 2 * 32 * 256 + 5 * 512;
            if (g_attn_dkv_rd == 4) {                                     // ("attn_dkv_rd" 4: a 4-slot fragment ring, 3 in flight)
              auto kern = attn_bwd_dkv128_resp_kernel<4>;
              (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p);
              hipLaunchKernelGGL(kern, grid, dim3(256), lds_p, s, a);
            } else {
              auto kern = attn_bwd_dkv128_resp_kernel<8>;
              (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p);
              hipLaunchKernelGGL(kern, grid, dim3(256), lds_p, s, a);
            }
          } else {
            const size_t lds_r = 2 * 128 * 256 + 4 * 32 * 256 + 2 * 64 * sizeof(float);
            auto kern = attn_bwd_dkv128_res_kernel<8>;
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_r);
            hipLaunchKernelGGL(kern, grid, dim3(256), lds_r, s, a);
          }
        } else if (g_attn_dkv_rd == 4) launch(attn_bwd_dkv128_pairp_kernel<4, false>);
        else if (g_attn_dkv_late) launch(attn_bwd_dkv128_pairp_kernel<8, true>);
        else launch(attn_bwd_dkv128_pairp_kernel<8, false>);
      } else {
        const size_t lds = 128 * 256 + 8 * 32 * 256 + 4 * 64 * sizeof(float);
        (void)hipFuncSetAttribute((const void*)attn_bwd_dkv128_pair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(attn_bwd_dkv128_pair_kernel, grid, block, lds, s, a);
      }
    } else {
      const size_t lds = 128 * 256 + 4 * 32 * 256 + 2 * 64 * sizeof(float);
      dim3 grid((a.Skv + 127) / 128, a.Hkv, a.B), block(256);
      (void)hipFuncSetAttribute((const void*)attn_bwd_dkv128_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(attn_bwd_dkv128_kernel, grid, block, lds, s, a);
    }
  } else {
    const size_t lds = 4 * 32 * D * 2 + 64 * sizeof(float);
    dim3 grid((a.Skv + 127) / 128, a.Hkv, a.B), block(256);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<D>, grid, block, lds, s, a);
  }
  return MM_OK;
}

}  // namespace

int mm_attn_option(const char* name, int value) {   // reached through mm_set_option (mm_gemm.hip)
  if (!strcmp(name, "attn_dkv_pair")) { g_attn_dkv_pair = value != 0; return MM_OK; }
  if (!strcmp(name, "attn_dkv_res")) { if (value < 0 || value > 2) return MM_ERR_ARG; g_attn_dkv_res = value; return MM_OK; }
  if (!strcmp(name, "attn_decode_mfma")) { g_attn_decode_mfma = value != 0; return MM_OK; }
  if (!strcmp(name, "attn_decode_wgs")) { if (value < 1) return MM_ERR_ARG; g_attn_decode_wgs = value; return MM_OK; }
  if (!strcmp(name, "attn_fwd_pf")) { g_attn_fwd_pf = value != 0; return MM_OK; }
  if (!strcmp(name, "attn_fwd_q")) { g_attn_fwd_q = value != 0; return MM_OK; }
  if (!strcmp(name, "attn_q_prio")) { g_attn_q_prio = value; return MM_OK; }
  if (!strcmp(name, "attn_dkv_late")) { g_attn_dkv_late = value != 0; return MM_OK; }
  if (!strcmp(name, "attn_dkv_rd")) { g_attn_dkv_rd = value == 4 ? 4 : 8; return MM_OK; }
  if (!strcmp(name, "attn_q_issue")) { g_attn_q_issue = value == 4 ? 4 : 8; return MM_OK; }
  if (!strcmp(name, "attn_q_rd")) { g_attn_q_rd = value; return MM_OK; }
  if (!strcmp(name, "attn_diag")) { g_attn_diag = value; return MM_OK; }
  if (!strcmp(name, "attn_fwd_waves")) { if (value != 4 && value != 8) return MM_ERR_ARG; g_attn_fwd_waves = value; return MM_OK; }
  return MM_ERR_ARG;
}

extern "C" int mm_attn_set_issue_waves(int v) {
  if (v != 4 && v != 8) return MM_ERR_ARG;
  g_attn_issue_waves = v;
  return MM_OK;
}

static int check_common(int dtype, int B, int Sq, int Skv, int Hq, int Hkv, int D) {
  if (B < 0 || Sq < 0 || Skv < 0 || Hq <= 0 || Hkv <= 0 || Hq % Hkv || D <= 0) return MM_ERR_ARG;
  if (dtype == MM_BF16 && D != 64 && D != 128) return MM_ERR_UNSUPPORTED;
  if (dtype == MM_F32 && (D > 256 || Skv > 12000)) return MM_ERR_UNSUPPORTED;
  if (dtype != MM_BF16 && dtype != MM_F32) return MM_ERR_UNSUPPORTED;
  return MM_OK;
}

extern "C" int mm_attn_fwd(int dtype, const void* q, const void* k, const void* v, int B, int Sq, int Skv, int Hq, int Hkv, int D,
                           int64_t q_sb, int64_t q_ss, int64_t q_sh, int64_t k_sb, int64_t k_ss, int64_t k_sh, int64_t v_sb, int64_t v_ss,
                           int64_t v_sh, const int64_t* key_mask, int causal, float scale, void* out, float* lse, void* stream) {
  int rc = check_common(dtype, B, Sq, Skv, Hq, Hkv, D);
  if (rc) return rc;
  if (!q || !k || !v || !out || !lse) return MM_ERR_ARG;
  if (B == 0 || Sq == 0) return MM_OK;
  AttnArgs a{};
  a.prio = g_attn_q_prio;
  a.diag = g_attn_diag;
  a.q = q; a.k = k; a.v = v; a.B = B; a.Sq = Sq; a.Skv = Skv; a.Hq = Hq; a.Hkv = Hkv;
  a.q_sb = q_sb; a.q_ss = q_ss; a.q_sh = q_sh; a.k_sb = k_sb; a.k_ss = k_ss; a.k_sh = k_sh; a.v_sb = v_sb; a.v_ss = v_ss; a.v_sh = v_sh;
  a.kmask = key_mask; a.causal = causal; a.scale = scale; a.out = out; a.lse = lse;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MM_BF16) {
    if ((q_sb | q_ss | q_sh | k_sb | k_ss | k_sh | v_sb | v_ss | v_sh) & 7) return MM_ERR_ALIGN;
    if (!mm_aligned16(q) || !mm_aligned16(k) || !mm_aligned16(v) || !mm_aligned16(out)) return MM_ERR_ALIGN;
    rc = D == 128 ? launch_bf16_fwd<128>(a, s) : launch_bf16_fwd<64>(a, s);
  } else {
    const size_t lds = (size_t)(Skv + D) * sizeof(float);
    hipLaunchKernelGGL(attn_fwd_f32_kernel, dim3(Sq, Hq, B), dim3(64), lds, s, a, D);
  }
  MM_CHECK_LAUNCH();
  return rc;
}

extern "C" int mm_attn_bwd(int dtype, const void* q, const void* k, const void* v, const void* out, const void* dout, const float* lse,
                           int B, int Sq, int Skv, int Hq, int Hkv, int D, int64_t q_sb, int64_t q_ss, int64_t q_sh, int64_t k_sb,
                           int64_t k_ss, int64_t k_sh, int64_t v_sb, int64_t v_ss, int64_t v_sh, const int64_t* key_mask, int causal,
                           float scale, void* dq, void* dk, void* dv, float* delta, void* stream) {
  int rc = check_common(dtype, B, Sq, Skv, Hq, Hkv, D);
  if (rc) return rc;
  if (!q || !k || !v || !out || !dout || !lse || !dq || !dk || !dv || !delta) return MM_ERR_ARG;
  if (B == 0 || Sq == 0) return MM_OK;
  AttnArgs a{};
  a.prio = g_attn_q_prio;
  a.diag = g_attn_diag;
  a.q = q; a.k = k; a.v = v; a.B = B; a.Sq = Sq; a.Skv = Skv; a.Hq = Hq; a.Hkv = Hkv;
  a.q_sb = q_sb; a.q_ss = q_ss; a.q_sh = q_sh; a.k_sb = k_sb; a.k_ss = k_ss; a.k_sh = k_sh; a.v_sb = v_sb; a.v_ss = v_ss; a.v_sh = v_sh;
  a.kmask = key_mask; a.causal = causal; a.scale = scale; a.out = (void*)out; a.lse = (float*)lse;
  a.dout = dout; a.delta = delta; a.dq = dq; a.dk = dk; a.dv = dv;
  hipStream_t s = (hipStream_t)stream;
  const int64_t rows = (int64_t)B * Sq * Hq;
  if (dtype == MM_BF16) {
    if ((q_sb | q_ss | q_sh | k_sb | k_ss | k_sh | v_sb | v_ss | v_sh) & 7) return MM_ERR_ALIGN;
    hipLaunchKernelGGL(attn_delta_kernel<bf16>, dim3((unsigned)((rows * (D / 8) + 255) / 256)), dim3(256), 0, s, (const bf16*)out, (const bf16*)dout, B, Sq, Hq, D, delta);
    rc = D == 128 ? launch_bf16_bwd<128>(a, s) : launch_bf16_bwd<64>(a, s);
  } else {
    {
      const int lpr = D / 4;
      const bool vec = (D % 4) == 0 && lpr <= 64 && (lpr & (lpr - 1)) == 0;
      const int64_t nthr = vec ? rows * lpr : rows * 64;
      hipLaunchKernelGGL(attn_delta_kernel<float>, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, s, (const float*)out, (const float*)dout, B, Sq, Hq, D, delta);
    }
    const size_t lds = (size_t)(2 * Skv + 2 * D) * sizeof(float);
    hipLaunchKernelGGL(attn_bwd_f32_kernel, dim3(Sq, Hq, B), dim3(64), lds, s, a, D);
  }
  MM_CHECK_LAUNCH();
  return rc;
}

extern "C" int mm_attn_decode_splits(int B, int Hkv, int Skv) {
  if (B <= 0 || Hkv <= 0 || Skv <= 0) return 1;
  int n = (g_attn_decode_wgs + B * Hkv - 1) / (B * Hkv);          // ~3 workgroups per CU
  const int cap = (Skv + 63) / 64;                  // at least 64 keys per slice
  if (n > cap) n = cap;
  return n < 1 ? 1 : n;
}

extern "C" int mm_attn_decode(int dtype, const void* q, const void* k, const void* v, int B, int Skv, int Hq, int Hkv, int D,
                              int64_t q_sb, int64_t q_sh, int64_t k_sb, int64_t k_ss, int64_t k_sh, int64_t v_sb, int64_t v_ss,
                              int64_t v_sh, const int64_t* key_mask, float scale, void* out, float* workspace, int nsplit,
                              int* sync, void* stream) {
  if (B < 0 || Skv <= 0 || Hq <= 0 || Hkv <= 0 || Hq % Hkv || nsplit < 1) return MM_ERR_ARG;
  if (dtype != MM_BF16 || (D != 64 && D != 128)) return MM_ERR_UNSUPPORTED;
  const int G = Hq / Hkv;
  if (G != 1 && G != 2 && G != 4 && G != 7 && G != 8) return MM_ERR_UNSUPPORTED;
  if (!q || !k || !v || !out || !workspace) return MM_ERR_ARG;
  if (B == 0) return MM_OK;
  if ((q_sb | q_sh | k_sb | k_ss | k_sh | v_sb | v_ss | v_sh) & 7) return MM_ERR_ALIGN;
  if (!mm_aligned16(q) || !mm_aligned16(k) || !mm_aligned16(v)) return MM_ERR_ALIGN;
  DecodeArgs a{q, k, v, B, Skv, Hq, Hkv, q_sb, q_sh, k_sb, k_ss, k_sh, v_sb, v_ss, v_sh, key_mask, scale, out, workspace, nsplit, 0, sync};
  a.chunk = (Skv + nsplit - 1) / nsplit;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(nsplit, Hkv, B), block(256);
#define MM_DEC(DD, GG) hipLaunchKernelGGL((attn_decode_partial_kernel<DD, GG>), grid, block, 0, s, a)
#define MM_DEC_G(DD)                                                                                          \
  switch (G) { case 1: MM_DEC(DD, 1); break; case 2: MM_DEC(DD, 2); break; case 4: MM_DEC(DD, 4); break;     \
               case 7: MM_DEC(DD, 7); break; default: MM_DEC(DD, 8); break; }
  if (D == 128 && G <= 4 && !sync && g_attn_decode_mfma) {           // scores on the matrix cores (two-launch form only)
    switch (G) {
      case 1: hipLaunchKernelGGL(attn_decode_partial128_mfma_kernel<1>, grid, block, 0, s, a); break;
      case 2: hipLaunchKernelGGL(attn_decode_partial128_mfma_kernel<2>, grid, block, 0, s, a); break;
      default: hipLaunchKernelGGL(attn_decode_partial128_mfma_kernel<4>, grid, block, 0, s, a); break;
    }
  } else if (D == 128) { MM_DEC_G(128) } else { MM_DEC_G(64) }
#undef MM_DEC_G
#undef MM_DEC
  MM_CHECK_LAUNCH();
  if (sync) return MM_OK;
  if (D == 128) hipLaunchKernelGGL(attn_decode_merge_kernel<128>, dim3(B * Hq), dim3(128), 0, s, workspace, nsplit, (bf16*)out);
  else hipLaunchKernelGGL(attn_decode_merge_kernel<64>, dim3(B * Hq), dim3(64), 0, s, workspace, nsplit, (bf16*)out);
  MM_CHECK_LAUNCH();
  return MM_OK;
}
