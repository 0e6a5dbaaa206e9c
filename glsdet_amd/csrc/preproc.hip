// Image preprocessing on the device, drone flavour:
//   PIL `image.resize(size, Image.BICUBIC)` on uint8 RGB  (drone/models/core/utils.py:21-34)
//   -> float32, /255, -mean, /std, HWC -> CHW               (utils.py:46-50, yolo.py:134)
// Pillow resamples in two passes (horizontal, then vertical) with 8-bit intermediate storage and
// fixed-point coefficients (22 fractional bits, rounded half away from zero); both passes here
// use exactly that arithmetic on coefficient tables the host computes the way Pillow's
// precompute_coeffs / normalize_coeffs_8bpc do, so the result is bit-identical to PIL's.
#include "common.h"

namespace glsdet {

#define GLS_PIL_BITS 22

__device__ __forceinline__ unsigned char pil_clip8(int v) {
  v >>= GLS_PIL_BITS;                 // arithmetic shift, as clip8_lookups[in >> PRECISION_BITS]
  return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal pass: tmp[y][xx][c] = clip8(2^21 + sum_k src[y][xmin+k][c] * kk[xx][k])
__global__ __launch_bounds__(256) void pil_rows_kernel(const unsigned char* __restrict__ src, int in_h, int in_w,
                                                       const int* __restrict__ bounds, const int* __restrict__ kk,
                                                       int ksize, int out_w, unsigned char* __restrict__ tmp) {
  const long total = (long)in_h * out_w;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int xx = (int)(i % out_w), y = (int)(i / out_w);
    const int xmin = bounds[2 * xx], xmax = bounds[2 * xx + 1];
    const int* k = kk + (long)xx * ksize;
    const unsigned char* row = src + ((long)y * in_w + xmin) * 3;
    int s0 = 1 << (GLS_PIL_BITS - 1), s1 = s0, s2 = s0;
    for (int x = 0; x < xmax; ++x) {
      const int c = k[x];
      s0 += row[3 * x + 0] * c;
      s1 += row[3 * x + 1] * c;
      s2 += row[3 * x + 2] * c;
    }
    unsigned char* o = tmp + i * 3;
    o[0] = pil_clip8(s0);
    o[1] = pil_clip8(s1);
    o[2] = pil_clip8(s2);
  }
}

// vertical pass + normalisation + HWC -> CHW into a window of the fp32 NCHW batch tensor:
//   v = clip8(2^21 + sum_k tmp[ymin+k][x][c] * kk[yy][k]);  f = float(v) / 255.0f  (float32)
//   f = float(double(f) - mean[c]);  f = float(double(f) / std[c])     (numpy's in-place mixed ops)
struct NormArgs {
  double mean[3], stdv[3];
};
__global__ __launch_bounds__(256) void pil_cols_norm_kernel(const unsigned char* __restrict__ tmp, int out_w,
                                                            const int* __restrict__ bounds, const int* __restrict__ kk,
                                                            int ksize, int out_h, float* __restrict__ dst, long plane,
                                                            int dst_w, int off_y, int off_x, const NormArgs na) {
  const long total = (long)out_h * out_w;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % out_w), yy = (int)(i / out_w);
    const int ymin = bounds[2 * yy], ymax = bounds[2 * yy + 1];
    const int* k = kk + (long)yy * ksize;
    int s[3] = {1 << (GLS_PIL_BITS - 1), 1 << (GLS_PIL_BITS - 1), 1 << (GLS_PIL_BITS - 1)};
    for (int y = 0; y < ymax; ++y) {
      const unsigned char* p = tmp + ((long)(ymin + y) * out_w + x) * 3;
      const int c = k[y];
      s[0] += p[0] * c;
      s[1] += p[1] * c;
      s[2] += p[2] * c;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float f = (float)pil_clip8(s[c]) / 255.0f;
      f = (float)((double)f - na.mean[c]);
      f = (float)((double)f / na.stdv[c]);
      dst[c * plane + (long)(off_y + yy) * dst_w + off_x + x] = f;
    }
  }
}

}  // namespace glsdet

using namespace glsdet;

static int check_table(const int32_t* b, const int32_t* k, int ksize, const char* what) {
  if (!b || !k || ksize < 1) GLS_FAIL(GLSDET_E_ARG, "%s: null / empty coefficient table", what);
  return 0;
}

extern "C" int glsdet_pil_resize_normalize(const unsigned char* src, int32_t in_h, int32_t in_w,
                                           const int32_t* xbounds, const int32_t* xkk, int32_t xksize, int32_t out_w,
                                           const int32_t* ybounds, const int32_t* ykk, int32_t yksize, int32_t out_h,
                                           unsigned char* tmp, float* dst, int32_t dst_h, int32_t dst_w, int32_t off_y,
                                           int32_t off_x, const double* mean3, const double* std3, void* stream) {
  if (!src || !tmp || !dst || !mean3 || !std3) GLS_FAIL(GLSDET_E_ARG, "pil_resize_normalize: null argument");
  int rc;
  if ((rc = check_table(xbounds, xkk, xksize, "pil_resize_normalize(x)"))) return rc;
  if ((rc = check_table(ybounds, ykk, yksize, "pil_resize_normalize(y)"))) return rc;
  if (in_h < 1 || in_w < 1 || out_h < 1 || out_w < 1 || dst_h < 1 || dst_w < 1 || off_y < 0 || off_x < 0 ||
      off_y + out_h > dst_h || off_x + out_w > dst_w)
    GLS_FAIL(GLSDET_E_ARG, "pil_resize_normalize: the %dx%d result does not fit the %dx%d canvas at (%d,%d)", out_h, out_w,
             dst_h, dst_w, off_y, off_x);
  NormArgs na;
  for (int c = 0; c < 3; ++c) {
    na.mean[c] = mean3[c];
    na.stdv[c] = std3[c];
  }
  OpRecord op;
  op.kind = 1;
  op.flops = 0;
  op.bytes = 3.0 * in_h * in_w + 6.0 * in_h * out_w + 12.0 * out_h * out_w;
  op.name = "pil_bicubic_resize+normalize";
  op.launch = [=](hipStream_t st) -> int {
    long g1 = ((long)in_h * out_w + 255) / 256, g2 = ((long)out_h * out_w + 255) / 256;
    if (g1 > 65535) g1 = 65535;
    if (g2 > 65535) g2 = 65535;
    hipLaunchKernelGGL(pil_rows_kernel, dim3((unsigned)g1), dim3(256), 0, st, src, in_h, in_w, xbounds, xkk, xksize, out_w, tmp);
    hipLaunchKernelGGL(pil_cols_norm_kernel, dim3((unsigned)g2), dim3(256), 0, st, tmp, out_w, ybounds, ykk, yksize, out_h, dst,
                       (long)dst_h * dst_w, dst_w, off_y, off_x, na);
    GLS_HIP(hipGetLastError());
    return 0;
  };
  return submit(std::move(op), stream);
}
