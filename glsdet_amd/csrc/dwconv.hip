// Depthwise k x k convolution + folded BN + activation (the `dconv` half of the reference's
// DWConv, drone/models/base/baseConv.py:22-30; used by the phi='nano' models).  One MAC per
// output element and tap: purely HBM bound, so no MFMA -- one thread per (pixel, 16-byte
// channel chunk), taps accumulated in fp32, weights [tap][C] read through L1/L2.
#include "common.h"

namespace glsdet {

template <typename T> struct DwVec;
template <> struct DwVec<f16> { typedef f16x8 type; static constexpr int N = 8; };
template <> struct DwVec<float> { typedef f32x4 type; static constexpr int N = 4; };

struct DwArgs {
  const unsigned char* x; const unsigned char* w; const float* scale; const float* bias; unsigned char* y;
  long x_sn, x_sh, x_sw, y_sn, y_sh, y_sw;
  int N, H, W, C, Ho, Wo, R, S, stride, pad, act, dil;
};

template <typename T>
__global__ __launch_bounds__(256) void dwconv_kernel(const DwArgs a) {
  typedef typename DwVec<T>::type V;
  constexpr int VN = DwVec<T>::N;
  const int cchunks = a.C / VN;
  const long total = (long)a.N * a.Ho * a.Wo * cchunks;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % cchunks);
    long p = i / cchunks;
    const int wo = (int)(p % a.Wo);
    p /= a.Wo;
    const int ho = (int)(p % a.Ho);
    const int n = (int)(p / a.Ho);
    float acc[VN];
#pragma unroll
    for (int e = 0; e < VN; ++e) acc[e] = 0.f;
    for (int r = 0; r < a.R; ++r) {
      const int hi = ho * a.stride - a.pad + r * a.dil;
      if (hi < 0 || hi >= a.H) continue;
      for (int s = 0; s < a.S; ++s) {
        const int wi = wo * a.stride - a.pad + s * a.dil;
        if (wi < 0 || wi >= a.W) continue;
        const V xv = *reinterpret_cast<const V*>(a.x + (n * a.x_sn + hi * a.x_sh + wi * a.x_sw + cc * VN) * (long)sizeof(T));
        const V wv = *reinterpret_cast<const V*>(a.w + ((long)(r * a.S + s) * a.C + cc * VN) * (long)sizeof(T));
#pragma unroll
        for (int e = 0; e < VN; ++e) acc[e] += (float)xv[e] * (float)wv[e];
      }
    }
    V out;
#pragma unroll
    for (int e = 0; e < VN; ++e) {
      float v = acc[e] * a.scale[cc * VN + e] + a.bias[cc * VN + e];
      if (a.act == GLSDET_ACT_SILU) v = v / (1.0f + expf(-v));
      else if (a.act == GLSDET_ACT_RELU) v = fmaxf(v, 0.f);
      else if (a.act == GLSDET_ACT_LRELU) v = v > 0.f ? v : 0.1f * v;
      out[e] = (T)v;
    }
    *reinterpret_cast<V*>(a.y + (n * a.y_sn + ho * a.y_sh + wo * a.y_sw + cc * VN) * (long)sizeof(T)) = out;
  }
}

}  // namespace glsdet

using namespace glsdet;

extern "C" int glsdet_dwconv2d(const glsdet_conv_desc* d, void* stream) { return glsdet_dwconv2d_dilated(d, 1, stream); }

extern "C" int glsdet_dwconv2d_dilated(const glsdet_conv_desc* d, int32_t dilation, void* stream) {
  if (!d) GLS_FAIL(GLSDET_E_ARG, "dwconv2d: null descriptor");
  if (dilation < 1 || dilation > 8) GLS_FAIL(GLSDET_E_ARG, "dwconv2d: dilation 1..8");
  const glsdet_view &x = d->x, &y = d->y;
  int rc;
  if ((rc = check_view(x, "dwconv2d.x"))) return rc;
  if ((rc = check_view(y, "dwconv2d.y"))) return rc;
  if (x.dtype != y.dtype || x.c != y.c || x.c % 8) GLS_FAIL(GLSDET_E_ARG, "dwconv2d: x/y channels (multiple of 8) and dtype must match");
  if (d->R < 1 || d->S < 1 || d->R > 15 || d->S > 15 || d->stride < 1 || d->stride > 4 || d->pad < 0)
    GLS_FAIL(GLSDET_E_ARG, "dwconv2d: bad R/S/stride/pad");
  if (d->res.base) GLS_FAIL(GLSDET_E_ARG, "dwconv2d: residual not supported (it belongs to the pointwise conv)");
  const int Ho = (x.h + 2 * d->pad - dilation * (d->R - 1) - 1) / d->stride + 1, Wo = (x.w + 2 * d->pad - dilation * (d->S - 1) - 1) / d->stride + 1;
  if (y.n != x.n || y.h != Ho || y.w != Wo) GLS_FAIL(GLSDET_E_ARG, "dwconv2d: output extent mismatch");
  if (!d->w || !d->scale || !d->bias || (((uintptr_t)d->w) & 15)) GLS_FAIL(GLSDET_E_ARG, "dwconv2d: bad weight/scale/bias");
  if (d->act < 0 || d->act > 3) GLS_FAIL(GLSDET_E_ARG, "dwconv2d: bad act");
  DwArgs a;
  a.x = (const unsigned char*)x.base; a.w = (const unsigned char*)d->w; a.scale = d->scale; a.bias = d->bias;
  a.y = (unsigned char*)y.base;
  a.x_sn = x.sn; a.x_sh = x.sh; a.x_sw = x.sw; a.y_sn = y.sn; a.y_sh = y.sh; a.y_sw = y.sw;
  a.N = x.n; a.H = x.h; a.W = x.w; a.C = x.c; a.Ho = Ho; a.Wo = Wo;
  a.R = d->R; a.S = d->S; a.stride = d->stride; a.pad = d->pad; a.act = d->act; a.dil = dilation;
  const int dt = x.dtype;
  OpRecord op;
  op.kind = 0;
  op.flops = 2.0 * (double)x.n * Ho * Wo * x.c * d->R * d->S;
  op.bytes = ((double)x.n * x.h * x.w + (double)x.n * Ho * Wo) * x.c * dtype_size(dt);
  char nm[96];
  snprintf(nm, sizeof nm, "dwconv<%s> %dx%d s%d d%d c%d", dt ? "f32" : "f16", d->R, d->S, d->stride, dilation, x.c);
  op.name = nm;
  op.launch = [a, dt](hipStream_t st) -> int {
    const long items = (long)a.N * a.Ho * a.Wo * (a.C / (dt == GLSDET_F16 ? 8 : 4));
    long g = (items + 255) / 256;
    if (g > 8192) g = 8192;
    if (dt == GLSDET_F16) hipLaunchKernelGGL(dwconv_kernel<f16>, dim3((unsigned)g), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(dwconv_kernel<float>, dim3((unsigned)g), dim3(256), 0, st, a);
    GLS_HIP(hipGetLastError());
    return 0;
  };
  return submit(std::move(op), stream);
}
