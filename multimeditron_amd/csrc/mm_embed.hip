// Embedding gather + modality splice (the arithmetic the reference owns outright, model.py:433-444),
// ViT patchify / embedding assembly (HF:clip:138-218) and CLS drop (image_modality.py:133).
// All HBM-bound row movers: one wave per row, 16-byte lanes, a token's source row is chosen from a
// precomputed map so the spliced sequence is written exactly once.
#include "mm_common.h"

namespace {

__global__ void fill_i32_kernel(int32_t* p, int n, int32_t v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
// last writer wins like index_put: serialise duplicates by taking the max source index
__global__ void splice_map_kernel(const int64_t* bi, const int64_t* tr, int n_mod, int S, int T, int32_t* map) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_mod) return;
  const int64_t pos = bi[i] * (int64_t)S + tr[i];
  if (pos >= 0 && pos < T) atomicMax(&map[pos], i);
}

// grid-stride over rows; one wave per row
template <typename T>
__global__ __launch_bounds__(256) void embed_splice_fwd_kernel(const T* emb, int64_t vocab, int H, const int64_t* ids, const T* proj,
                                                               const int32_t* map, int Tn, T* out) {
  constexpr int VN = Vec16<T>::N;
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nw = gridDim.x * 4;
  for (int t = wave; t < Tn; t += nw) {
    const int32_t src = map ? map[t] : -1;
    const T* row;
    if (src >= 0) {
      row = proj + (int64_t)src * H;
    } else {
      int64_t id = ids[t];
      if (id < 0 || id >= vocab) id = 0;  // torch would raise; never read out of bounds
      row = emb + id * H;
    }
    T* o = out + (int64_t)t * H;
    for (int e = lane * VN; e < H; e += 64 * VN) *(Vec16<T>*)(o + e) = *(const Vec16<T>*)(row + e);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void splice_gather_bwd_kernel(const T* dE, int H, const int64_t* bi, const int64_t* tr, int n_mod,
                                                                int S, int Tn, const int32_t* map, T* dproj) {
  constexpr int VN = Vec16<T>::N;
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nw = gridDim.x * 4;
  for (int i = wave; i < n_mod; i += nw) {
    const int64_t pos = bi[i] * (int64_t)S + tr[i];
    T* o = dproj + (int64_t)i * H;
    // a duplicate position that lost the index_put race receives no gradient
    const bool live = pos >= 0 && pos < Tn && map[pos] == i;
    for (int e = lane * VN; e < H; e += 64 * VN) {
      Vec16<T> v;
      if (live) v = *(const Vec16<T>*)(dE + pos * H + e);
      else
#pragma unroll
        for (int k = 0; k < VN; ++k) v.set(k, 0.f);
      *(Vec16<T>*)(o + e) = v;
    }
  }
}

__device__ __forceinline__ void atomic_add_pair(bf16* p, float a, float b) {
  // packed bf16 atomic add (global_atomic_pk_add_bf16); p is 4-byte aligned
  bf16x2 v;
  v[0] = (bf16)a;
  v[1] = (bf16)b;
  __builtin_amdgcn_global_atomic_fadd_v2bf16((__attribute__((address_space(1))) bf16x2*)p, v);
}

template <typename T>
__global__ __launch_bounds__(256) void embed_scatter_bwd_kernel(const T* dE, int H, const int64_t* ids, const int32_t* map, int Tn,
                                                                T* demb, int64_t vocab) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nw = gridDim.x * 4;
  for (int t = wave; t < Tn; t += nw) {
    if (map && map[t] >= 0) continue;  // overwritten by a modality row: no gradient to the embedding
    const int64_t id = ids[t];
    if (id < 0 || id >= vocab) continue;
    const T* g = dE + (int64_t)t * H;
    T* d = demb + id * H;
    if constexpr (sizeof(T) == 4) {
      for (int e = lane; e < H; e += 64) atomicAdd((float*)d + e, to_f32(g[e]));
    } else {
      for (int e = lane * 2; e < H; e += 128) atomic_add_pair((bf16*)d + e, to_f32(g[e]), to_f32(g[e + 1]));
    }
  }
}

// ---- ViT glue --------------------------------------------------------------------------------------------
// patches[(i*P + py*g + px), c*ps*ps + y*ps + x] = pixels[i, c, py*ps + y, px*ps + x]
template <typename T>
__global__ void patchify_kernel(const float* pix, int n, int himg, int wimg, int ps, int kpad, T* out) {
  const int g = wimg / ps, gh = himg / ps;
  const int P = g * gh;
  const int64_t total = (int64_t)n * P * kpad;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int k = (int)(i % kpad);
  const int64_t rp = i / kpad;
  const int p = (int)(rp % P);
  const int img = (int)(rp / P);
  float v = 0.f;
  if (k < 3 * ps * ps) {
    const int c = k / (ps * ps), rem = k % (ps * ps), y = rem / ps, x = rem % ps;
    const int py = p / g, px = p % g;
    v = pix[(((int64_t)img * 3 + c) * himg + py * ps + y) * wimg + px * ps + x];
  }
  out[i] = from_f32<T>(v);
}

template <typename T>
__global__ void vit_embed_fwd_kernel(const T* patch_out, const T* cls, const T* pos, int n, int P, int D, T* x) {
  const int64_t total = (int64_t)n * (P + 1) * D;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int d = (int)(i % D);
  const int64_t r = i / D;
  const int tok = (int)(r % (P + 1));
  const int img = (int)(r / (P + 1));
  const float base = tok == 0 ? to_f32(cls[d]) : to_f32(patch_out[((int64_t)img * P + tok - 1) * D + d]);
  x[i] = from_f32<T>(base + to_f32(pos[(int64_t)tok * D + d]));
}

// dpos[tok,d] (+)= sum_img dx[img,tok,d]; dcls[d] (+)= sum_img dx[img,0,d]; dpatch_out = dx[:,1:,:]
template <typename T>
__global__ void vit_embed_bwd_kernel(const T* dx, int n, int P, int D, T* dpatch, T* dcls, T* dpos, int accumulate) {
  const int64_t total = (int64_t)(P + 1) * D;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int d = (int)(i % D);
  const int tok = (int)(i / D);
  float s = 0.f;
  for (int img = 0; img < n; ++img) {
    const T v = dx[((int64_t)img * (P + 1) + tok) * D + d];
    s += to_f32(v);
    if (tok > 0 && dpatch) dpatch[((int64_t)img * P + tok - 1) * D + d] = v;
  }
  if (dpos) dpos[i] = from_f32<T>(s + (accumulate ? to_f32(dpos[i]) : 0.f));
  if (tok == 0 && dcls) dcls[d] = from_f32<T>(s + (accumulate ? to_f32(dcls[d]) : 0.f));
}

template <typename T, bool BWD>
__global__ void drop_cls_kernel(const T* src, int n, int P, int D, T* dst) {
  constexpr int VN = Vec16<T>::N;
  // FWD: dst[n,P,D] = src[n,1+P,D][:,1:]   BWD: dst[n,1+P,D] = pad(src[n,P,D]) with a zero CLS row
  const int rows = BWD ? (P + 1) : P;
  const int per_row = D / VN;
  const int64_t total = (int64_t)n * rows * per_row;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % per_row) * VN;
  const int64_t r = i / per_row;
  const int tok = (int)(r % rows);
  const int img = (int)(r / rows);
  Vec16<T> v;
  if (!BWD) {
    v = *(const Vec16<T>*)(src + ((int64_t)img * (P + 1) + tok + 1) * D + c);
  } else if (tok == 0) {
#pragma unroll
    for (int k = 0; k < VN; ++k) v.set(k, 0.f);
  } else {
    v = *(const Vec16<T>*)(src + ((int64_t)img * P + tok - 1) * D + c);
  }
  *(Vec16<T>*)(dst + ((int64_t)img * rows + tok) * D + c) = v;
}

inline unsigned row_grid(int rows) { return (unsigned)max(1, min((rows + 3) / 4, 2048)); }

}  // namespace

extern "C" int mm_splice_build_map(const int64_t* batch_idx, const int64_t* token_range, int n_mod, int S, int T, int32_t* src_map,
                                   void* stream) {
  if (!src_map || T < 0 || n_mod < 0 || (n_mod > 0 && (!batch_idx || !token_range))) return MM_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (T > 0) hipLaunchKernelGGL(fill_i32_kernel, dim3((T + 255) / 256), dim3(256), 0, s, src_map, T, -1);
  if (n_mod > 0) hipLaunchKernelGGL(splice_map_kernel, dim3((n_mod + 255) / 256), dim3(256), 0, s, batch_idx, token_range, n_mod, S, T, src_map);
  MM_CHECK_LAUNCH();
  return MM_OK;
}

extern "C" int mm_embed_splice_fwd(int dtype, const void* emb, int64_t vocab, int H, const int64_t* ids, const void* proj,
                                   const int32_t* src_map, int T, void* out, void* stream) {
  if (!emb || !ids || !out || T < 0 || H <= 0 || vocab <= 0) return MM_ERR_ARG;
  if (src_map && !proj) return MM_ERR_ARG;
  if (T == 0) return MM_OK;
  const int vn = dtype == MM_BF16 ? 8 : 4;
  if (H % vn || !mm_aligned16(emb) || !mm_aligned16(out) || (proj && !mm_aligned16(proj))) return MM_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MM_BF16)
    hipLaunchKernelGGL(embed_splice_fwd_kernel<bf16>, dim3(row_grid(T)), dim3(256), 0, s, (const bf16*)emb, vocab, H, ids, (const bf16*)proj, src_map, T, (bf16*)out);
  else
    hipLaunchKernelGGL(embed_splice_fwd_kernel<float>, dim3(row_grid(T)), dim3(256), 0, s, (const float*)emb, vocab, H, ids, (const float*)proj, src_map, T, (float*)out);
  MM_CHECK_LAUNCH();
  return MM_OK;
}

extern "C" int mm_embed_splice_bwd(int dtype, const void* dE, int H, const int64_t* ids, const int32_t* src_map, int T,
                                   const int64_t* batch_idx, const int64_t* token_range, int n_mod, int S, void* dproj, void* demb,
                                   int64_t vocab, void* stream) {
  if (!dE || T < 0 || H <= 0) return MM_ERR_ARG;
  const int vn = dtype == MM_BF16 ? 8 : 4;
  if (H % vn || !mm_aligned16(dE)) return MM_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  if (dproj && n_mod > 0) {
    if (!batch_idx || !token_range || !src_map) return MM_ERR_ARG;
    if (dtype == MM_BF16)
      hipLaunchKernelGGL(splice_gather_bwd_kernel<bf16>, dim3(row_grid(n_mod)), dim3(256), 0, s, (const bf16*)dE, H, batch_idx, token_range, n_mod, S, T, src_map, (bf16*)dproj);
    else
      hipLaunchKernelGGL(splice_gather_bwd_kernel<float>, dim3(row_grid(n_mod)), dim3(256), 0, s, (const float*)dE, H, batch_idx, token_range, n_mod, S, T, src_map, (float*)dproj);
  }
  if (demb && T > 0) {
    if (!ids) return MM_ERR_ARG;
    if (dtype == MM_BF16)
      hipLaunchKernelGGL(embed_scatter_bwd_kernel<bf16>, dim3(row_grid(T)), dim3(256), 0, s, (const bf16*)dE, H, ids, src_map, T, (bf16*)demb, vocab);
    else
      hipLaunchKernelGGL(embed_scatter_bwd_kernel<float>, dim3(row_grid(T)), dim3(256), 0, s, (const float*)dE, H, ids, src_map, T, (float*)demb, vocab);
  }
  MM_CHECK_LAUNCH();
  return MM_OK;
}

extern "C" int mm_patchify(int dtype, const float* pixels, int n, int himg, int wimg, int ps, int kpad, void* patches, void* stream) {
  if (!pixels || !patches || n < 0 || ps <= 0 || himg < ps || wimg < ps || kpad < 3 * ps * ps) return MM_ERR_ARG;   // a ragged border is dropped, as a stride-ps "valid" conv does (SigLIP 384 / 14)
  if (n == 0) return MM_OK;
  const int64_t total = (int64_t)n * (himg / ps) * (wimg / ps) * kpad;
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (dtype == MM_BF16)
    hipLaunchKernelGGL(patchify_kernel<bf16>, grid, block, 0, (hipStream_t)stream, pixels, n, himg, wimg, ps, kpad, (bf16*)patches);
  else
    hipLaunchKernelGGL(patchify_kernel<float>, grid, block, 0, (hipStream_t)stream, pixels, n, himg, wimg, ps, kpad, (float*)patches);
  MM_CHECK_LAUNCH();
  return MM_OK;
}

extern "C" int mm_vit_embed_fwd(int dtype, const void* patch_out, const void* cls, const void* pos, int n, int P, int D, void* x,
                                void* stream) {
  if (!patch_out || !cls || !pos || !x || n < 0 || P <= 0 || D <= 0) return MM_ERR_ARG;
  if (n == 0) return MM_OK;
  const int64_t total = (int64_t)n * (P + 1) * D;
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (dtype == MM_BF16)
    hipLaunchKernelGGL(vit_embed_fwd_kernel<bf16>, grid, block, 0, (hipStream_t)stream, (const bf16*)patch_out, (const bf16*)cls, (const bf16*)pos, n, P, D, (bf16*)x);
  else
    hipLaunchKernelGGL(vit_embed_fwd_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float*)patch_out, (const float*)cls, (const float*)pos, n, P, D, (float*)x);
  MM_CHECK_LAUNCH();
  return MM_OK;
}

extern "C" int mm_vit_embed_bwd(int dtype, const void* dx, int n, int P, int D, void* dpatch_out, void* dcls, void* dpos, int accumulate,
                                void* stream) {
  if (!dx || n < 0 || P <= 0 || D <= 0) return MM_ERR_ARG;
  const int64_t total = (int64_t)(P + 1) * D;
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (dtype == MM_BF16)
    hipLaunchKernelGGL(vit_embed_bwd_kernel<bf16>, grid, block, 0, (hipStream_t)stream, (const bf16*)dx, n, P, D, (bf16*)dpatch_out, (bf16*)dcls, (bf16*)dpos, accumulate);
  else
    hipLaunchKernelGGL(vit_embed_bwd_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float*)dx, n, P, D, (float*)dpatch_out, (float*)dcls, (float*)dpos, accumulate);
  MM_CHECK_LAUNCH();
  return MM_OK;
}

template <bool BWD>
static int drop_cls_launch(int dtype, const void* src, int n, int P, int D, void* dst, void* stream) {
  if (!src || !dst || n < 0 || P <= 0 || D <= 0) return MM_ERR_ARG;
  if (n == 0) return MM_OK;
  const int vn = dtype == MM_BF16 ? 8 : 4;
  if (D % vn || !mm_aligned16(src) || !mm_aligned16(dst)) return MM_ERR_ALIGN;
  const int64_t total = (int64_t)n * (BWD ? P + 1 : P) * (D / vn);
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (dtype == MM_BF16)
    hipLaunchKernelGGL((drop_cls_kernel<bf16, BWD>), grid, block, 0, (hipStream_t)stream, (const bf16*)src, n, P, D, (bf16*)dst);
  else
    hipLaunchKernelGGL((drop_cls_kernel<float, BWD>), grid, block, 0, (hipStream_t)stream, (const float*)src, n, P, D, (float*)dst);
  MM_CHECK_LAUNCH();
  return MM_OK;
}
extern "C" int mm_drop_cls_fwd(int dtype, const void* src, int n, int P, int D, void* dst, void* stream) {
  return drop_cls_launch<false>(dtype, src, n, P, D, dst, stream);
}
extern "C" int mm_drop_cls_bwd(int dtype, const void* ddst, int n, int P, int D, void* dsrc, void* stream) {
  return drop_cls_launch<true>(dtype, ddst, n, P, D, dsrc, stream);
}

// ---- plug-in towers: learned positions without a CLS row, heads padded to a width the MFMA attention supports ----------
namespace {
template <typename T>
__global__ void bcast_add_kernel(const T* x, const T* b, int64_t L, int64_t total, T* y) {
  constexpr int VN = Vec16<T>::N;
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * VN;
  if (i >= total) return;
  const int64_t j = i % L;                 // L % VN == 0: a vector never straddles two images
  Vec16<T> xv = *(const Vec16<T>*)(x + i), bv = *(const Vec16<T>*)(b + j), o;
#pragma unroll
  for (int k = 0; k < VN; ++k) o.set(k, xv.get(k) + bv.get(k));
  *(Vec16<T>*)(y + i) = o;
}

// one thread = one 16-byte vector of the WIDE side [rows, nheads, dpad]
template <typename T, bool INVERSE>
__global__ void head_pad_kernel(const T* src, int64_t rows, int nheads, int d, int dpad, T* dst) {
  constexpr int VN = Vec16<T>::N;
  const int vpw = dpad / VN;                                   // vectors per padded head
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * nheads * vpw) return;
  const int c = (int)(i % vpw) * VN;
  const int64_t rh = i / vpw;                                  // row * nheads + head
  if (INVERSE) {
    if (c < d) *(Vec16<T>*)(dst + rh * d + c) = *(const Vec16<T>*)(src + rh * dpad + c);
  } else {
    Vec16<T> v;
#pragma unroll
    for (int k = 0; k < VN; ++k) v.set(k, 0.f);
    if (c < d) v = *(const Vec16<T>*)(src + rh * d + c);
    *(Vec16<T>*)(dst + rh * dpad + c) = v;
  }
}
}  // namespace

extern "C" int mm_bcast_add(int dtype, const void* x, const void* b, int n, int64_t L, void* y, void* stream) {
  if (!x || !b || !y || n < 0 || L <= 0) return MM_ERR_ARG;
  if (dtype != MM_BF16 && dtype != MM_F32) return MM_ERR_UNSUPPORTED;
  if (n == 0) return MM_OK;
  const int vn = dtype == MM_BF16 ? 8 : 4;
  if (L % vn || !mm_aligned16(x) || !mm_aligned16(b) || !mm_aligned16(y)) return MM_ERR_ALIGN;
  const int64_t total = (int64_t)n * L;
  dim3 grid((unsigned)((total / vn + 255) / 256)), block(256);
  if (dtype == MM_BF16)
    hipLaunchKernelGGL(bcast_add_kernel<bf16>, grid, block, 0, (hipStream_t)stream, (const bf16*)x, (const bf16*)b, L, total, (bf16*)y);
  else
    hipLaunchKernelGGL(bcast_add_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float*)x, (const float*)b, L, total, (float*)y);
  MM_CHECK_LAUNCH();
  return MM_OK;
}

extern "C" int mm_head_pad(int dtype, const void* src, int64_t rows, int nheads, int d, int dpad, void* dst, int inverse,
                           void* stream) {
  if (!src || !dst || rows < 0 || nheads <= 0 || d <= 0 || dpad < d) return MM_ERR_ARG;
  if (dtype != MM_BF16 && dtype != MM_F32) return MM_ERR_UNSUPPORTED;
  if (rows == 0) return MM_OK;
  const int vn = dtype == MM_BF16 ? 8 : 4;
  if (d % vn || dpad % vn || !mm_aligned16(src) || !mm_aligned16(dst)) return MM_ERR_ALIGN;
  const int64_t total = rows * nheads * (dpad / vn);
  if ((total + 255) / 256 > 0x7FFFFFFF) return MM_ERR_ARG;
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
#define MM_HP(T, INV) hipLaunchKernelGGL((head_pad_kernel<T, INV>), grid, block, 0, (hipStream_t)stream, (const T*)src, rows, nheads, d, dpad, (T*)dst)
  if (dtype == MM_BF16) { if (inverse) MM_HP(bf16, true); else MM_HP(bf16, false); }
  else { if (inverse) MM_HP(float, true); else MM_HP(float, false); }
#undef MM_HP
  MM_CHECK_LAUNCH();
  return MM_OK;
}
