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

// *flag = 1 when any id lies outside [0, vocab): the reference's nn.Embedding raises (device-side assert) for such a batch,
// also for ids under a modality span (model.py:433 embeds ALL of input_ids before the splice).  The lookup kernel below maps
// them to row 0 so that nothing is read out of bounds; this flag is how the host gets to raise (kernels.embed_check_ids).
__global__ __launch_bounds__(256) void embed_check_ids_kernel(const int64_t* ids, int Tn, int64_t vocab, int* flag) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t < Tn && (ids[t] < 0 || ids[t] >= vocab)) *flag = 1;
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

// ---- embedding gradient: deterministic, fp32-accumulated, one write per touched row ---------------------------------
// backward of `embedding(input_ids)` (model.py:433): demb[id] (+)= sum over the tokens t with ids[t] == id that were not
// overwritten by a modality row.  No float atomics (a packed-bf16 atomic rounds the running sum at every add and the
// order of adds differs between launches).  Instead:
//   1. mm_embed_sort (forward time: it depends only on ids and the splice map): a stable enumeration sort of the tokens
//      by id -- pos(t) = #{t' : (key[t'], t') < (key[t], t)} -- O(T^2) integer compares from LDS, no V-sized tables, the
//      same permutation on every launch and on every rank.  Spliced / out-of-range tokens get the key INVALID (last).
//   2. embed_reduce_chunks_kernel: the sorted positions are cut into chunks of 32; a wave owns (chunk, 16-byte column
//      slice), keeps its 32 gradient rows' slices in registers and walks them in ascending token order, summing each run
//      of equal ids in fp32.  A run that lies inside the chunk is rounded once and written to demb[id]; a run that
//      touches a chunk boundary shared with the same id goes to an fp32 scratch (two slots per chunk).
//   3. embed_merge_partials_kernel: for every id whose run crosses chunks, the chunk where the run begins adds the
//      partials in ascending chunk order and writes demb[id].
// So a token id repeated thousands of times (padding) is summed by many waves, and every sum has one fixed order.
constexpr int EMB_R = 32;                 // sorted positions per chunk
constexpr int32_t EMB_INVALID = 0x7fffffff;

__global__ void embed_key_kernel(const int64_t* ids, const int32_t* map, int Tn, int64_t vocab, int32_t* key, int32_t* skey, int Tpad) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < Tn) {
    const int64_t id = ids[t];
    key[t] = ((map && map[t] >= 0) || id < 0 || id >= vocab) ? EMB_INVALID : (int32_t)id;
  } else if (t < Tpad) {
    skey[t] = EMB_INVALID;                // padding behind the sorted keys: chunk tails read as "no token"
  }
}

// 64 tokens per workgroup (one per lane); the 4 waves split every staged block of keys; counts are merged through LDS
__global__ __launch_bounds__(256) void embed_rank_sort_kernel(const int32_t* key, int Tn, int32_t* order, int32_t* skey) {
  constexpr int CH = 8192;                // keys staged per round (32 KiB)
  __shared__ int32_t sk[CH];
  __shared__ int cnt[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int t = blockIdx.x * 64 + lane;
  const int32_t kt = t < Tn ? key[t] : EMB_INVALID;
  int c = 0;
  for (int base = 0; base < Tn; base += CH) {
    const int n = min(CH, Tn - base);
    __syncthreads();
    for (int i = threadIdx.x; i < CH; i += 256) sk[i] = i < n ? key[base + i] : EMB_INVALID;
    __syncthreads();
    const int q = (n + 3) / 4;            // this wave's quarter [w*q, min(n, w*q+q))
    const int j0 = w * q, j1 = min(n, j0 + q);
    for (int j = j0; j < j1; ++j) {       // sk[j] is a broadcast read (same address in every lane)
      const int32_t kj = sk[j];
      c += (kj < kt) | ((kj == kt) & (base + j < t));
    }
  }
  cnt[w][lane] = c;
  __syncthreads();
  if (w == 0 && t < Tn) {
    const int pos = cnt[0][lane] + cnt[1][lane] + cnt[2][lane] + cnt[3][lane];
    order[pos] = t;
    skey[pos] = kt;
  }
}

template <typename T>
__device__ __forceinline__ void emb_store_row(T* dst, const float (&acc)[Vec16<T>::N], int accumulate) {
  constexpr int VN = Vec16<T>::N;
  Vec16<T> o;
  if (accumulate) {
    const Vec16<T> old = *(const Vec16<T>*)dst;
#pragma unroll
    for (int k = 0; k < VN; ++k) o.set(k, old.get(k) + acc[k]);
  } else {
#pragma unroll
    for (int k = 0; k < VN; ++k) o.set(k, acc[k]);
  }
  *(Vec16<T>*)dst = o;
}

template <typename T>
__global__ __launch_bounds__(256) void embed_reduce_chunks_kernel(const T* dE, int H, const int32_t* order, const int32_t* skey,
                                                                  int nchunks, int nslices, T* demb, float* part, int accumulate) {
  constexpr int VN = Vec16<T>::N;
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wave >= nchunks * nslices) return;
  const int c = wave / nslices, sl = wave % nslices;
  const int col = (sl * 64 + lane) * VN;
  const bool live = col < H;                                   // the last slice may be ragged (H % (64*VN) != 0)
  const int p0 = c * EMB_R;
  const int32_t myk = lane < EMB_R ? skey[p0 + lane] : EMB_INVALID;
  const int32_t myo = lane < EMB_R ? order[p0 + lane] : 0;
  const int32_t kfirst = __builtin_amdgcn_readfirstlane(myk);
  if (kfirst == EMB_INVALID) return;                           // sorted: nothing valid from here on
  const int32_t kprev = c > 0 ? skey[p0 - 1] : -1;             // -1 never equals a key
  const int32_t knext = skey[p0 + EMB_R];                      // skey is padded by one chunk of INVALID
  Vec16<T> rows[EMB_R];
#pragma unroll
  for (int i = 0; i < EMB_R; ++i) {
    const int32_t ki = __shfl(myk, i, 64);
    const int32_t oi = __shfl(myo, i, 64);
    if (ki != EMB_INVALID && live) rows[i] = *(const Vec16<T>*)(dE + (int64_t)oi * H + col);
  }
  float acc[VN];
#pragma unroll
  for (int k = 0; k < VN; ++k) acc[k] = 0.f;
  int32_t run = kfirst;
  bool run_at_start = true;
  float* pc = part + ((int64_t)c * 2) * H + col;
  auto flush = [&](int32_t rk, bool at_start, bool at_end) {
    if (rk == EMB_INVALID || !live) return;
    const bool ts = at_start && kprev == rk, te = at_end && knext == rk;
    if (!ts && !te) {
      emb_store_row<T>(demb + (int64_t)rk * H + col, acc, accumulate);
    } else {
      float* d = pc + (ts ? 0 : (int64_t)H);                   // slot 0: run continues FROM the previous chunk; slot 1: it
#pragma unroll                                                 // begins here and continues INTO the next one
      for (int k = 0; k < VN; ++k) d[k] = acc[k];
    }
  };
#pragma unroll
  for (int i = 0; i < EMB_R; ++i) {
    const int32_t ki = __shfl(myk, i, 64);
    if (ki != run) {
      flush(run, run_at_start, false);
      run = ki;
      run_at_start = false;
#pragma unroll
      for (int k = 0; k < VN; ++k) acc[k] = 0.f;
    }
    if (ki != EMB_INVALID) {
#pragma unroll
      for (int k = 0; k < VN; ++k) acc[k] += rows[i].get(k);
    }
  }
  flush(run, run_at_start, true);
}

template <typename T>
__global__ __launch_bounds__(256) void embed_merge_partials_kernel(int H, const int32_t* skey, int nchunks, int nslices, T* demb,
                                                                   const float* part, int accumulate) {
  constexpr int VN = Vec16<T>::N;
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wave >= nchunks * nslices) return;
  const int c = wave / nslices, sl = wave % nslices;
  const int col = (sl * 64 + lane) * VN;
  if (col >= H) return;
  const int p0 = c * EMB_R;
  const int32_t kl = skey[p0 + EMB_R - 1];                     // key of the chunk's last run
  if (kl == EMB_INVALID || skey[p0 + EMB_R] != kl) return;     // the last run ends inside this chunk
  if (skey[p0] == kl && c > 0 && skey[p0 - 1] == kl) return;   // the run began in an earlier chunk: that chunk leads
  float acc[VN];
  const float* p = part + ((int64_t)c * 2 + 1) * H + col;
#pragma unroll
  for (int k = 0; k < VN; ++k) acc[k] = p[k];
  for (int cc = c + 1; cc < nchunks; ++cc) {                   // ascending chunk order: one fixed summation order
    const float* q = part + ((int64_t)cc * 2) * H + col;
#pragma unroll
    for (int k = 0; k < VN; ++k) acc[k] += q[k];
    const int q0 = cc * EMB_R;
    if (!(skey[q0 + EMB_R - 1] == kl && skey[q0 + EMB_R] == kl)) break;   // the run ends inside chunk cc
  }
  emb_store_row<T>(demb + (int64_t)kl * H + col, acc, accumulate);
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

extern "C" int mm_embed_check_ids(const int64_t* ids, int T, int64_t vocab, int* flag, void* stream) {
  if (!ids || !flag || T < 0 || vocab <= 0) return MM_ERR_ARG;
  if (T == 0) return MM_OK;
  hipLaunchKernelGGL(embed_check_ids_kernel, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ids, T, vocab, flag);
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

extern "C" int mm_embed_sort_sizes(int T, int H, int64_t* order_elems, int64_t* scratch_floats) {
  if (T < 0 || H <= 0 || !order_elems || !scratch_floats) return MM_ERR_ARG;
  const int64_t nchunks = (T + EMB_R - 1) / EMB_R;
  *order_elems = (nchunks + 1) * EMB_R;
  *scratch_floats = nchunks * 2 * (int64_t)H;
  return MM_OK;
}

extern "C" int mm_embed_sort(const int64_t* ids, const int32_t* src_map, int T, int64_t vocab, int32_t* key_ws, int32_t* order,
                             int32_t* skey, void* stream) {
  if (T < 0 || vocab <= 0 || (T > 0 && (!ids || !key_ws || !order || !skey))) return MM_ERR_ARG;
  if (T == 0) return MM_OK;
  hipStream_t s = (hipStream_t)stream;
  const int Tpad = ((T + EMB_R - 1) / EMB_R + 1) * EMB_R;
  hipLaunchKernelGGL(embed_key_kernel, dim3((Tpad + 255) / 256), dim3(256), 0, s, ids, src_map, T, vocab, key_ws, skey, Tpad);
  hipLaunchKernelGGL(embed_rank_sort_kernel, dim3((T + 63) / 64), dim3(256), 0, s, (const int32_t*)key_ws, T, order, skey);
  MM_CHECK_LAUNCH();
  return MM_OK;
}

extern "C" int mm_embed_splice_bwd(int dtype, const void* dE, int H, const int64_t* ids, const int32_t* src_map, int T,
                                   const int64_t* batch_idx, const int64_t* token_range, int n_mod, int S, void* dproj, void* demb,
                                   int64_t vocab, const int32_t* order, const int32_t* skey, float* scratch, int accumulate,
                                   void* stream) {
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
    if (!ids || !order || !skey || !scratch) return MM_ERR_ARG;      // mm_embed_sort's outputs (same ids / src_map / vocab)
    if (!mm_aligned16(demb) || !mm_aligned16(scratch)) return MM_ERR_ALIGN;
    (void)vocab;
    const int nchunks = (T + EMB_R - 1) / EMB_R;
    const int nslices = (H + 64 * vn - 1) / (64 * vn);
    const unsigned grid = (unsigned)(((int64_t)nchunks * nslices + 3) / 4);
    if (dtype == MM_BF16) {
      hipLaunchKernelGGL(embed_reduce_chunks_kernel<bf16>, dim3(grid), dim3(256), 0, s, (const bf16*)dE, H, order, skey, nchunks, nslices, (bf16*)demb, scratch, accumulate);
      hipLaunchKernelGGL(embed_merge_partials_kernel<bf16>, dim3(grid), dim3(256), 0, s, H, skey, nchunks, nslices, (bf16*)demb, (const float*)scratch, accumulate);
    } else {
      hipLaunchKernelGGL(embed_reduce_chunks_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)dE, H, order, skey, nchunks, nslices, (float*)demb, scratch, accumulate);
      hipLaunchKernelGGL(embed_merge_partials_kernel<float>, dim3(grid), dim3(256), 0, s, H, skey, nchunks, nslices, (float*)demb, (const float*)scratch, accumulate);
    }
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

// ---- loss rows: the rows of the last hidden state whose (shifted) label is not -100 ------------------------------------
// HF computes logits for every position and lets cross_entropy ignore the -100 ones (HF:loss/loss_utils.py:36-71); the loss
// and every gradient depend only on the labelled rows (an ignored row's dlogits are exactly zero), so the training step runs
// the final norm, lm_head and the loss on those rows alone.  One kernel serves both directions: dst[r] = src[map[r]], or a
// zero row where map[r] < 0 (forward: map = the labelled rows' indices; backward: map = the inverse, -1 at ignored rows).
namespace {
template <typename T>
__global__ void rows_select_kernel(const T* src, int64_t ld_src, const int* map, int n_src, int n_dst, int D, T* dst, int64_t ld_dst) {
  constexpr int VN = Vec16<T>::N;
  const int per_row = D / VN;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)n_dst * per_row) return;
  const int c = (int)(i % per_row) * VN;
  const int r = (int)(i / per_row);
  const int m = map[r];
  Vec16<T> v;
  if (m >= 0 && m < n_src) {
    v = *(const Vec16<T>*)(src + (int64_t)m * ld_src + c);
  } else {
#pragma unroll
    for (int k = 0; k < VN; ++k) v.set(k, 0.f);
  }
  *(Vec16<T>*)(dst + (int64_t)r * ld_dst + c) = v;
}
}  // namespace

extern "C" int mm_rows_select(int dtype, const void* src, int64_t ld_src, const int* map, int n_src, int n_dst, int D, void* dst,
                              int64_t ld_dst, void* stream) {
  if (!src || !dst || !map || n_src < 0 || n_dst < 0 || D <= 0) return MM_ERR_ARG;
  if (n_dst == 0) return MM_OK;
  const int vn = dtype == MM_BF16 ? 8 : 4;
  if (D % vn || ld_src % vn || ld_dst % vn || ld_src < D || ld_dst < D || !mm_aligned16(src) || !mm_aligned16(dst)) return MM_ERR_ALIGN;
  const int64_t total = (int64_t)n_dst * (D / vn);
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (dtype == MM_BF16)
    hipLaunchKernelGGL(rows_select_kernel<bf16>, grid, block, 0, (hipStream_t)stream, (const bf16*)src, ld_src, map, n_src, n_dst, D, (bf16*)dst, ld_dst);
  else
    hipLaunchKernelGGL(rows_select_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float*)src, ld_src, map, n_src, n_dst, D, (float*)dst, ld_dst);
  MM_CHECK_LAUNCH();
  return MM_OK;
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
