// CLIP / SigLIP image preprocessing on the device: resize (Pillow's bicubic resampling, bit for bit) -> center crop -> rescale ->
// normalise, from the decoded uint8 RGB image to the fp32 pixel tensor the vision tower reads.
//
// Replaces what the reference does on the CPU inside its collator: `AutoImageProcessor.from_pretrained(clip_name)(images=image)`
// (image_modality.py:77,88-93 -> HF CLIPImageProcessor: PIL resize BICUBIC, center_crop, rescale 1/255, normalize), SURVEY 8f-1
// "optional GPU image preprocessing".  HBM-bound integer/byte work: no MFMA, no LDS; every output element is an independent
// short dot product of bytes with fixed-point weights.
//
// Pillow (src/libImaging/Resample.c) resamples in two passes, horizontal then vertical, each pixel = clip8((2^21 + sum_k
// src[k] * w[k]) >> 22) with int32 weights w = round(filter weight * 2^22) that the HOST computes in double precision exactly as
// `precompute_coeffs` + `normalize_coeffs_8bpc` do (multimeditron_amd/dataset/gpu_image.py).  The intermediate image is uint8, as
// in Pillow: that rounding between the passes is part of the result.  Only what the crop window needs is computed.
#include "mm_common.h"

namespace {

__device__ __forceinline__ int clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// horizontal pass: tmp[r - r0][x][c] for source rows r in [r0, r1) and resized columns x in [left, left + cw)
__global__ __launch_bounds__(256) void image_resample_h_kernel(const uint8_t* src, int src_stride, int r0, int nrows, const int* xb,
                                                               const int* xk, int kx, int left, int cw, uint8_t* tmp) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;          // (row, x): one thread = one pixel, 3 channels
  if (i >= (int64_t)nrows * cw) return;
  const int r = (int)(i / cw), x = (int)(i % cw) + left;
  const int xmin = xb[2 * x], xn = xb[2 * x + 1];
  const int* k = xk + (int64_t)x * kx;
  const uint8_t* p = src + (int64_t)(r0 + r) * src_stride + (int64_t)xmin * 3;
  int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
  for (int t = 0; t < xn; ++t) {
    const int w = k[t];
    s0 += p[3 * t] * w;
    s1 += p[3 * t + 1] * w;
    s2 += p[3 * t + 2] * w;
  }
  uint8_t* o = tmp + i * 3;
  o[0] = (uint8_t)clip8(s0 >> 22);
  o[1] = (uint8_t)clip8(s1 >> 22);
  o[2] = (uint8_t)clip8(s2 >> 22);
}

// vertical pass over tmp + rescale + normalise: out[c][y][x] (CHW fp32), y in [0, ch), x in [0, cw)
__global__ __launch_bounds__(256) void image_resample_v_norm_kernel(const uint8_t* tmp, int r0, int cw, const int* yb, const int* yk, int ky,
                                                                    int top, int ch, float rescale, int do_rescale, float m0, float m1,
                                                                    float m2, float d0, float d1, float d2, int do_norm, float* out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)ch * cw) return;
  const int y = (int)(i / cw), x = (int)(i % cw);
  const int ymin = yb[2 * (y + top)], yn = yb[2 * (y + top) + 1];
  const int* k = yk + (int64_t)(y + top) * ky;
  const uint8_t* p = tmp + ((int64_t)(ymin - r0) * cw + x) * 3;
  int s[3] = {1 << 21, 1 << 21, 1 << 21};
  for (int t = 0; t < yn; ++t) {
    const int w = k[t];
    const uint8_t* q = p + (int64_t)t * cw * 3;
    s[0] += q[0] * w;
    s[1] += q[1] * w;
    s[2] += q[2] * w;
  }
  const float mean[3] = {m0, m1, m2}, sd[3] = {d0, d1, d2};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float v = (float)clip8(s[c] >> 22);
    // the CPU path: x * rescale_factor, then (x - mean) / std, each one float32 operation (no contraction into an FMA)
    if (do_rescale) v = __fmul_rn(v, rescale);
    if (do_norm) v = __fdiv_rn(__fsub_rn(v, mean[c]), sd[c]);
    out[((int64_t)c * ch + y) * cw + x] = v;
  }
}

}  // namespace

extern "C" int mm_image_resample_h(const void* src_u8, int src_h, int src_w, int src_row_stride, int r0, int nrows, const int* xbounds,
                                   const int* xcoef, int kx, int left, int cw, void* tmp_u8, void* stream) {
  if (!src_u8 || !xbounds || !xcoef || !tmp_u8 || src_h <= 0 || src_w <= 0 || src_row_stride < 3 * src_w || kx <= 0) return MM_ERR_ARG;
  if (r0 < 0 || nrows < 0 || r0 + nrows > src_h || left < 0 || cw < 0) return MM_ERR_ARG;
  const int64_t n = (int64_t)nrows * cw;
  if (n == 0) return MM_OK;
  hipLaunchKernelGGL(image_resample_h_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)src_u8,
                     src_row_stride, r0, nrows, xbounds, xcoef, kx, left, cw, (uint8_t*)tmp_u8);
  MM_CHECK_LAUNCH();
  return MM_OK;
}

extern "C" int mm_image_resample_v_norm(const void* tmp_u8, int r0, int nrows, int cw, const int* ybounds, const int* ycoef, int ky, int top,
                                        int ch, float rescale, int do_rescale, const float* mean3_host, const float* std3_host, int do_norm,
                                        float* out_chw, void* stream) {
  if (!tmp_u8 || !ybounds || !ycoef || !out_chw || ky <= 0 || nrows < 0 || cw < 0 || ch < 0 || top < 0) return MM_ERR_ARG;
  if (do_norm && (!mean3_host || !std3_host)) return MM_ERR_ARG;
  const int64_t n = (int64_t)ch * cw;
  if (n == 0) return MM_OK;
  const float m0 = do_norm ? mean3_host[0] : 0.f, m1 = do_norm ? mean3_host[1] : 0.f, m2 = do_norm ? mean3_host[2] : 0.f;
  const float d0 = do_norm ? std3_host[0] : 1.f, d1 = do_norm ? std3_host[1] : 1.f, d2 = do_norm ? std3_host[2] : 1.f;
  hipLaunchKernelGGL(image_resample_v_norm_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)tmp_u8,
                     r0, cw, ybounds, ycoef, ky, top, ch, rescale, do_rescale, m0, m1, m2, d0, d1, d2, do_norm, out_chw);
  MM_CHECK_LAUNCH();
  return MM_OK;
}
