// Image side of the UFPMP-Det second stage:
//   display_merge_result (ufp/ufpmp_det_eval.py:182-193): magnified crops of the source image
//   composited into one mosaic, then the mmdet test pipeline on it (Resize keep-ratio ->
//   Normalize(to_rgb) -> Pad(32), mmdet/datasets/pipelines/transforms.py:30,671,572).
// cv2.resize semantics are restated (cv2 is not available to pin them): uint8 INTER_LINEAR with
// 11-bit fixed-point coefficients for the crops, half-pixel float bilinear for the mosaic.
#include "common.h"

namespace glsdet {

struct LinTap {
  int i0, i1, c0, c1;      // source indices and 11-bit coefficients
};

// OpenCV per-axis INTER_LINEAR setup (fraction in float32, taps clamped at the borders)
__device__ __forceinline__ void lin_tap_f(int d, int dst, int src, int& i0, int& i1, float& f) {
  const double scale = 1.0 / ((double)dst / (double)src);
  float fx = (float)(((double)d + 0.5) * scale - 0.5);
  int s = (int)floorf(fx);
  fx -= (float)s;
  if (s < 0) { s = 0; fx = 0.f; }
  if (s >= src - 1) { s = src - 1; fx = 0.f; }
  i0 = s;
  i1 = min(s + 1, src - 1);
  f = fx;
}
__device__ __forceinline__ LinTap lin_tap_u8(int d, int dst, int src) {
  LinTap t;
  float f;
  lin_tap_f(d, dst, src, t.i0, t.i1, f);
  t.c0 = (int)rintf((1.0f - f) * 2048.0f);
  t.c1 = (int)rintf(f * 2048.0f);
  return t;
}

// canvas[y][x][c] (fp32, BGR like the source) = pixel of the chip that covers (x, y), else 0
__global__ __launch_bounds__(256) void ufp_mosaic_kernel(const unsigned char* __restrict__ img, int H, int W,
                                                         const float* __restrict__ chips, int n_chips,
                                                         float* __restrict__ canvas, int ch, int cw) {
  const long total = (long)ch * cw;
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (long)gridDim.x * blockDim.x) {
    const int x = (int)(p % cw), y = (int)(p / cw);
    float out[3] = {0.f, 0.f, 0.f};
    for (int k = 0; k < n_chips; ++k) {               // later chips overwrite earlier ones, as the slice assignments do
      const float* c = chips + 7 * k;
      const int x1 = (int)floorf(c[0]), y1 = (int)floorf(c[1]), w = (int)floorf(c[2]), h = (int)floorf(c[3]);
      const int nx = (int)floorf(c[4]), ny = (int)floorf(c[5]), s = (int)floorf(c[6]);
      if (w == 0 || h == 0) continue;
      if (x < nx || y < ny || x >= nx + w * s || y >= ny + h * s) continue;
      const int sw = min(w, W - x1), sh = min(h, H - y1);          // numpy clips the crop at the image border
      if (sw <= 0 || sh <= 0) continue;
      const int dx = x - nx, dy = y - ny;
      const unsigned char* base = img + ((long)y1 * W + x1) * 3;
      if (sw == w * s && sh == h * s) {                            // same size: cv2.resize copies
#pragma unroll
        for (int e = 0; e < 3; ++e) out[e] = (float)base[((long)dy * W + dx) * 3 + e];
        continue;
      }
      const LinTap tx = lin_tap_u8(dx, w * s, sw), ty = lin_tap_u8(dy, h * s, sh);
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        const int r0 = base[((long)ty.i0 * W + tx.i0) * 3 + e] * tx.c0 + base[((long)ty.i0 * W + tx.i1) * 3 + e] * tx.c1;
        const int r1 = base[((long)ty.i1 * W + tx.i0) * 3 + e] * tx.c0 + base[((long)ty.i1 * W + tx.i1) * 3 + e] * tx.c1;
        int v = (((ty.c0 * (r0 >> 4)) >> 16) + ((ty.c1 * (r1 >> 4)) >> 16) + 2) >> 2;
        out[e] = (float)(v < 0 ? 0 : (v > 255 ? 255 : v));
      }
    }
    float* o = canvas + p * 3;
    o[0] = out[0]; o[1] = out[1]; o[2] = out[2];
  }
}

// mosaic (fp32 HWC, BGR) -> bilinear resize to nh x nw -> RGB, (v - mean) * (1/std) -> zero-padded
// fp32 [3][ph][pw]
struct NormArgs3 {
  double mean[3], stdinv[3];
};
__global__ __launch_bounds__(256) void resize_norm_pad_kernel(const float* __restrict__ src, int h, int w, int nh, int nw,
                                                              float* __restrict__ dst, int ph, int pw, const NormArgs3 na) {
  const long total = (long)ph * pw;
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (long)gridDim.x * blockDim.x) {
    const int x = (int)(p % pw), y = (int)(p / pw);
    float v[3] = {0.f, 0.f, 0.f};
    if (x < nw && y < nh) {
      int x0, x1, y0, y1;
      float fx, fy;
      lin_tap_f(x, nw, w, x0, x1, fx);
      lin_tap_f(y, nh, h, y0, y1, fy);
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        const double a = (double)src[((long)y0 * w + x0) * 3 + e] * (1.0 - (double)fx) + (double)src[((long)y0 * w + x1) * 3 + e] * (double)fx;
        const double b = (double)src[((long)y1 * w + x0) * 3 + e] * (1.0 - (double)fx) + (double)src[((long)y1 * w + x1) * 3 + e] * (double)fx;
        const float bgr = (float)(a * (1.0 - (double)fy) + b * (double)fy);
        const int c = 2 - e;                                      // BGR -> RGB
        float f = (float)((double)bgr - na.mean[c]);
        f = (float)((double)f * na.stdinv[c]);
        v[c] = f;
      }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) dst[(long)c * ph * pw + p] = v[c];
  }
}

// The same pipeline on a uint8 source (the FIRST stage: `cv2.imread` frame -> mmcv.imresize): cv2.resize
// works in 11-bit fixed point on uint8 and rounds to uint8 before Normalize sees the pixel.
__global__ __launch_bounds__(256) void resize_norm_pad_u8_kernel(const unsigned char* __restrict__ src, int h, int w, int nh,
                                                                 int nw, float* __restrict__ dst, int ph, int pw,
                                                                 const NormArgs3 na) {
  const long total = (long)ph * pw;
  const bool same = nh == h && nw == w;                            // cv2.resize returns a copy
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (long)gridDim.x * blockDim.x) {
    const int x = (int)(p % pw), y = (int)(p / pw);
    float v[3] = {0.f, 0.f, 0.f};
    if (x < nw && y < nh) {
      LinTap tx, ty;
      if (!same) {
        tx = lin_tap_u8(x, nw, w);
        ty = lin_tap_u8(y, nh, h);
      }
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        int q;
        if (same) {
          q = src[((long)y * w + x) * 3 + e];
        } else {
          const int r0 = src[((long)ty.i0 * w + tx.i0) * 3 + e] * tx.c0 + src[((long)ty.i0 * w + tx.i1) * 3 + e] * tx.c1;
          const int r1 = src[((long)ty.i1 * w + tx.i0) * 3 + e] * tx.c0 + src[((long)ty.i1 * w + tx.i1) * 3 + e] * tx.c1;
          q = (((ty.c0 * (r0 >> 4)) >> 16) + ((ty.c1 * (r1 >> 4)) >> 16) + 2) >> 2;
          q = q < 0 ? 0 : (q > 255 ? 255 : q);
        }
        const int c = 2 - e;                                      // BGR -> RGB
        float f = (float)((double)(float)q - na.mean[c]);
        f = (float)((double)f * na.stdinv[c]);
        v[c] = f;
      }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) dst[(long)c * ph * pw + p] = v[c];
  }
}

}  // namespace glsdet

using namespace glsdet;

extern "C" int glsdet_ufp_mosaic(const unsigned char* img, int32_t H, int32_t W, const float* chips, int32_t n_chips,
                                 float* canvas, int32_t ch, int32_t cw, void* stream) {
  if (!img || !canvas || (n_chips > 0 && !chips)) GLS_FAIL(GLSDET_E_ARG, "ufp_mosaic: null argument");
  if (H < 1 || W < 1 || ch < 1 || cw < 1 || n_chips < 0) GLS_FAIL(GLSDET_E_ARG, "ufp_mosaic: bad sizes");
  OpRecord op;
  op.kind = 1;
  op.flops = 0;
  op.bytes = 12.0 * ch * cw + 3.0 * H * W;
  op.name = "ufp_mosaic";
  op.launch = [=](hipStream_t st) -> int {
    long g = ((long)ch * cw + 255) / 256;
    if (g > 65535) g = 65535;
    hipLaunchKernelGGL(ufp_mosaic_kernel, dim3((unsigned)g), dim3(256), 0, st, img, H, W, chips, n_chips, canvas, ch, cw);
    GLS_HIP(hipGetLastError());
    return 0;
  };
  return submit(std::move(op), stream);
}

extern "C" int glsdet_resize_normalize_pad(const float* src, int32_t h, int32_t w, int32_t nh, int32_t nw, float* dst,
                                           int32_t ph, int32_t pw, const double* mean_rgb, const double* std_rgb,
                                           void* stream) {
  if (!src || !dst || !mean_rgb || !std_rgb) GLS_FAIL(GLSDET_E_ARG, "resize_normalize_pad: null argument");
  if (h < 1 || w < 1 || nh < 1 || nw < 1 || ph < nh || pw < nw) GLS_FAIL(GLSDET_E_ARG, "resize_normalize_pad: bad sizes");
  NormArgs3 na;
  for (int c = 0; c < 3; ++c) {
    na.mean[c] = mean_rgb[c];
    na.stdinv[c] = 1.0 / std_rgb[c];
  }
  OpRecord op;
  op.kind = 1;
  op.flops = 0;
  op.bytes = 12.0 * h * w + 12.0 * ph * pw;
  op.name = "resize_normalize_pad";
  op.launch = [=](hipStream_t st) -> int {
    long g = ((long)ph * pw + 255) / 256;
    if (g > 65535) g = 65535;
    hipLaunchKernelGGL(resize_norm_pad_kernel, dim3((unsigned)g), dim3(256), 0, st, src, h, w, nh, nw, dst, ph, pw, na);
    GLS_HIP(hipGetLastError());
    return 0;
  };
  return submit(std::move(op), stream);
}

extern "C" int glsdet_resize_normalize_pad_u8(const unsigned char* src, int32_t h, int32_t w, int32_t nh, int32_t nw,
                                              float* dst, int32_t ph, int32_t pw, const double* mean_rgb,
                                              const double* std_rgb, void* stream) {
  if (!src || !dst || !mean_rgb || !std_rgb) GLS_FAIL(GLSDET_E_ARG, "resize_normalize_pad_u8: null argument");
  if (h < 1 || w < 1 || nh < 1 || nw < 1 || ph < nh || pw < nw) GLS_FAIL(GLSDET_E_ARG, "resize_normalize_pad_u8: bad sizes");
  NormArgs3 na;
  for (int c = 0; c < 3; ++c) {
    na.mean[c] = mean_rgb[c];
    na.stdinv[c] = 1.0 / std_rgb[c];
  }
  OpRecord op;
  op.kind = 1;
  op.flops = 0;
  op.bytes = 3.0 * h * w + 12.0 * ph * pw;
  op.name = "resize_normalize_pad_u8";
  op.launch = [=](hipStream_t st) -> int {
    long g = ((long)ph * pw + 255) / 256;
    if (g > 65535) g = 65535;
    hipLaunchKernelGGL(resize_norm_pad_u8_kernel, dim3((unsigned)g), dim3(256), 0, st, src, h, w, nh, nw, dst, ph, pw, na);
    GLS_HIP(hipGetLastError());
    return 0;
  };
  return submit(std::move(op), stream);
}
