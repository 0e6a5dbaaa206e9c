// HBM-bound helpers of the ResNet-50 / FPN / GFL / MPHead path (SURVEY section 8a rows A10,
// A11): image packing for the 7x7 stem, strided max pool, FPN nearest-upsample-add,
// GroupNorm (+ReLU) of the head towers and the MPHead proxy scores.  All are 16-byte-per-lane
// NHWC kernels; the contractions of this path go through glsdet_conv2d.
#include "common.h"

namespace glsdet {

template <typename T> struct V16;
template <> struct V16<f16> { typedef f16x8 type; static constexpr int N = 8; };
template <> struct V16<float> { typedef f32x4 type; static constexpr int N = 4; };

static inline unsigned rgrid(long work_items, long cap = 256L * 32) {
  long g = (work_items + 255) / 256;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (unsigned)g;
}

// ---------------------------------------------------------------- NCHW fp32 -> NHWC pack
// One thread per pixel: cin strided float loads (coalesced across the wave along W), the
// padded pixel leaves as 16-byte stores.
template <typename T>
__global__ __launch_bounds__(256) void nchw_pack_kernel(const float* __restrict__ img, int n, int cin, int H, int W,
                                                        unsigned char* y, long sn, long sh, long sw, int cy) {
  typedef typename V16<T>::type V;
  constexpr int VN = V16<T>::N;
  const long total = (long)n * H * W;
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (long)gridDim.x * blockDim.x) {
    const int w = (int)(p % W);
    const int h = (int)((p / W) % H);
    const int b = (int)(p / ((long)W * H));
    T* out = reinterpret_cast<T*>(y) + b * sn + h * sh + w * sw;
    const float* src = img + ((long)b * cin * H + h) * W + w;
    for (int c0 = 0; c0 < cy; c0 += VN) {
      V pk;
#pragma unroll
      for (int e = 0; e < VN; ++e) pk[e] = (c0 + e) < cin ? (T)src[(long)(c0 + e) * H * W] : (T)0.f;
      *reinterpret_cast<V*>(out + c0) = pk;
    }
  }
}

// ---------------------------------------------------------------- max pool k, stride, pad
template <typename T>
__global__ __launch_bounds__(256) void pool2d_kernel(const unsigned char* x, long xsn, long xsh, long xsw, int H, int W,
                                                     unsigned char* y, long ysn, long ysh, long ysw, int n, int Ho,
                                                     int Wo, int C, int k, int stride, int pad) {
  typedef typename V16<T>::type V;
  constexpr int VN = V16<T>::N;
  const int cchunks = C / VN;
  const long total = (long)n * Ho * Wo * cchunks;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % cchunks);
    long p = i / cchunks;
    const int wo = (int)(p % Wo);
    p /= Wo;
    const int ho = (int)(p % Ho);
    const int b = (int)(p / Ho);
    V m;
#pragma unroll
    for (int e = 0; e < VN; ++e) m[e] = (T)(-INFINITY);
    for (int dy = 0; dy < k; ++dy) {
      const int hh = ho * stride - pad + dy;
      if (hh < 0 || hh >= H) continue;
      for (int dx = 0; dx < k; ++dx) {
        const int ww = wo * stride - pad + dx;
        if (ww < 0 || ww >= W) continue;
        const V v = *reinterpret_cast<const V*>(x + (b * xsn + hh * xsh + ww * xsw + cc * VN) * (long)sizeof(T));
#pragma unroll
        for (int e = 0; e < VN; ++e) m[e] = v[e] > m[e] ? v[e] : m[e];
      }
    }
    *reinterpret_cast<V*>(y + (b * ysn + ho * ysh + wo * ysw + cc * VN) * (long)sizeof(T)) = m;
  }
}

// ---------------------------------------------------------------- fine += nearest(coarse)
// torch 'nearest' with size=: src = min((int)floorf(dst * scale), in - 1), scale = (float)in / out.
template <typename T>
__global__ __launch_bounds__(256) void upsample_add_kernel(const unsigned char* x, long xsn, long xsh, long xsw, int Hc,
                                                           int Wc, unsigned char* y, long ysn, long ysh, long ysw,
                                                           int n, int H, int W, int C, float scale_h, float scale_w) {
  typedef typename V16<T>::type V;
  constexpr int VN = V16<T>::N;
  const int cchunks = C / VN;
  const long total = (long)n * H * W * cchunks;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % cchunks);
    long p = i / cchunks;
    const int w = (int)(p % W);
    p /= W;
    const int h = (int)(p % H);
    const int b = (int)(p / H);
    const int hs = min((int)floorf((float)h * scale_h), Hc - 1);
    const int ws = min((int)floorf((float)w * scale_w), Wc - 1);
    const V a = *reinterpret_cast<const V*>(x + (b * xsn + hs * xsh + ws * xsw + cc * VN) * (long)sizeof(T));
    V* dst = reinterpret_cast<V*>(y + (b * ysn + h * ysh + w * ysw + cc * VN) * (long)sizeof(T));
    V d = *dst;
#pragma unroll
    for (int e = 0; e < VN; ++e) d[e] = (T)((float)d[e] + (float)a[e]);
    *dst = d;
  }
}

// ---------------------------------------------------------------- GroupNorm
// Pass 1: a workgroup owns a slice of one image's pixels; thread t always reads vector column
// t % vcols (256 % vcols == 0), so its elements all belong to ONE group; sum and sum of
// squares are kept in fp64 (no cancellation issue in E[x^2]-mean^2), folded per group in a
// fixed order and written as one partial per (image, slice, group).
#define GLS_GN_SPLIT 64
#define GLS_GN_SETS 16
// 1..16 independent tensors of the same C / groups / image count per launch pair (the cls and reg
// towers of all five pyramid levels): operands in one kernarg struct, set = blockIdx.z.
struct GnSet {
  const unsigned char* x;
  unsigned char* y;
  long xsn, xsh, xsw, ysn, ysh, ysw;
  const float *gamma, *beta;
  double* partial;
  float* mr;                    // [image][group] mean, rstd (gn_fold_kernel -> gn_apply_kernel)
  int H, W, nsplit, chunk, gb;
  int pstride, pre;             // partial slices per image in `partial`; pre: they were written by the producing conv
};
struct GnArgs {
  GnSet s[GLS_GN_SETS];
  int n, C, groups, act;
  float eps;
};

template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const GnArgs a) {
  typedef typename V16<T>::type V;
  constexpr int VN = V16<T>::N;
  __shared__ double s_sum[256], s_sq[256];
  const GnSet& S = a.s[blockIdx.z];
  const int b = blockIdx.y, z = blockIdx.x;
  if (z >= S.nsplit || S.pre) return;
  const int C = a.C, groups = a.groups, W = S.W;
  const int vcols = C / VN, rows = 256 / vcols;
  const int cc = threadIdx.x % vcols, r0 = threadIdx.x / vcols;
  const int N = S.H * W;
  const int pbeg = z * S.chunk, pend = min(N, pbeg + S.chunk);
  double sum = 0.0, sq = 0.0;
  for (int p = pbeg + r0; p < pend; p += rows) {
    const V v = *reinterpret_cast<const V*>(S.x + (b * S.xsn + (p / W) * S.xsh + (p % W) * S.xsw + cc * VN) * (long)sizeof(T));
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int e = 0; e < VN; ++e) {
      const float f = (float)v[e];
      s1 += f;
      s2 += f * f;
    }
    sum += (double)s1;
    sq += (double)s2;
  }
  s_sum[threadIdx.x] = sum;
  s_sq[threadIdx.x] = sq;
  __syncthreads();
  if ((int)threadIdx.x < groups) {
    const int g = threadIdx.x, cpg = C / groups, vpg = cpg / VN;       // vector columns per group
    double t = 0.0, q = 0.0;
    for (int r = 0; r < rows; ++r)
      for (int c = g * vpg; c < (g + 1) * vpg; ++c) {
        t += s_sum[r * vcols + c];
        q += s_sq[r * vcols + c];
      }
    double* o = S.partial + (((long)b * S.pstride + z) * groups + g) * 2;
    o[0] = t;
    o[1] = q;
  }
}

// Pass 2: fold the partial slices of one (set, image): thread t sums the slices z = t / groups, t / groups + R, ... of group
// t % groups (R = 256 / groups rows of threads), the rows are folded through LDS in a fixed order -> mean, rstd.
__global__ __launch_bounds__(256) void gn_fold_kernel(const GnArgs a) {
  __shared__ double s_t[256], s_q[256];
  const GnSet& S = a.s[blockIdx.y];
  const int b = blockIdx.x, groups = a.groups;
  const int R = 256 / groups;
  const int g = threadIdx.x % groups, r = threadIdx.x / groups;
  double t = 0.0, q = 0.0;
  if (r < R)
    for (int z = r; z < S.nsplit; z += R) {
      const double* o = S.partial + (((long)b * S.pstride + z) * groups + g) * 2;
      t += o[0];
      q += o[1];
    }
  s_t[threadIdx.x] = t;
  s_q[threadIdx.x] = q;
  __syncthreads();
  if ((int)threadIdx.x < groups) {
    for (int k = 1; k < R; ++k) {
      t += s_t[k * groups + g];
      q += s_q[k * groups + g];
    }
    const double cnt = (double)S.H * S.W * (a.C / groups);
    const double mean = t / cnt;
    double var = q / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    S.mr[((long)b * groups + g) * 2] = (float)mean;
    S.mr[((long)b * groups + g) * 2 + 1] = (float)(1.0 / sqrt(var + (double)a.eps));
  }
}

// Pass 3: y = act((x - mean) * rstd * gamma + beta).
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const GnArgs a) {
  typedef typename V16<T>::type V;
  constexpr int VN = V16<T>::N;
  __shared__ float s_mean[256], s_rstd[256];
  const GnSet& S = a.s[blockIdx.z];
  if ((int)blockIdx.x >= S.gb) return;
  const int b = blockIdx.y;
  const int C = a.C, groups = a.groups, W = S.W;
  const int N = S.H * W, cpg = C / groups;
  if ((int)threadIdx.x < groups) {
    s_mean[threadIdx.x] = S.mr[((long)b * groups + threadIdx.x) * 2];
    s_rstd[threadIdx.x] = S.mr[((long)b * groups + threadIdx.x) * 2 + 1];
  }
  __syncthreads();
  const int vcols = C / VN, rows = 256 / vcols;
  const int cc = threadIdx.x % vcols, r0 = threadIdx.x / vcols;
  const int g = cc * VN / cpg;
  float sc[VN], sh[VN];
#pragma unroll
  for (int e = 0; e < VN; ++e) {
    const float ga = S.gamma[cc * VN + e] * s_rstd[g];
    sc[e] = ga;
    sh[e] = S.beta[cc * VN + e] - s_mean[g] * ga;
  }
  for (int p = blockIdx.x * rows + r0; p < N; p += S.gb * rows) {
    const long ox = (b * S.xsn + (p / W) * S.xsh + (p % W) * S.xsw + cc * VN) * (long)sizeof(T);
    const long oy = (b * S.ysn + (p / W) * S.ysh + (p % W) * S.ysw + cc * VN) * (long)sizeof(T);
    V v = *reinterpret_cast<const V*>(S.x + ox);
#pragma unroll
    for (int e = 0; e < VN; ++e) {
      float f = (float)v[e] * sc[e] + sh[e];
      if (a.act == GLSDET_ACT_RELU) f = fmaxf(f, 0.f);
      v[e] = (T)f;
    }
    *reinterpret_cast<V*>(S.y + oy) = v;
  }
}

// ---------------------------------------------------------------- MPHead proxy scores
// Sixteen lanes per position (four positions per wave): |feat| by a shuffle reduction over the lanes' channel vectors
// (16-byte loads, the row of a position read once, coalesced); lane c < num_classes of the group then walks the proxies of
// class c in the position's dots row (hot in L1: the group has just touched it) for the softmax-weighted mean
// gamma * sum_j softmax(gamma s)_j s_j,  s_j = dots_j / max(|feat|, 1e-12)  (mp_head.py:105-121).  The first cut gave every
// position a whole wave (half of its lanes idle at 256 channels, one 512-byte row in flight per wave): 0.103 ms = 1.2 TB/s on
// the 179 k positions of the benchmark.
struct ProxyArgs {
  int P, nc, maxcnt;
  unsigned char cls_of[256];     // class of proxy k
  unsigned char first[256];      // first proxy of class c
  unsigned char count[256];      // proxies of class c
};
template <typename T>
__global__ __launch_bounds__(256) void proxy_scores_kernel(const unsigned char* f, long fsn, long fsh, long fsw,
                                                           const float* dots, long dsn, long dsh, long dsw, float* out,
                                                           long osn, long osh, long osw, int n, int H, int W, int C,
                                                           float gamma, const ProxyArgs a) {
  typedef typename V16<T>::type V;
  constexpr int VN = V16<T>::N;
  const int lane = threadIdx.x & 63, sub = lane & 15, grp = lane >> 4;
  const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  const long total = (long)n * H * W;
  for (long p0 = wave * 4; p0 < total; p0 += nwaves * 4) {
    const bool okp = p0 + grp < total;
    const long p = okp ? p0 + grp : total - 1;      // (every lane takes part in the shuffles)
    const int w = (int)(p % W);
    const int h = (int)((p / W) % H);
    const int b = (int)(p / ((long)W * H));
    const unsigned char* px = f + (b * fsn + h * fsh + w * fsw) * (long)sizeof(T);
    float sq = 0.f;
    for (int c = sub * VN; c < C; c += 16 * VN) {
      const V v = *reinterpret_cast<const V*>(px + c * (long)sizeof(T));
#pragma unroll
      for (int e = 0; e < VN; ++e) sq += (float)v[e] * (float)v[e];
    }
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) sq += __shfl_xor(sq, o, 64);
    const float inv = 1.0f / fmaxf(sqrtf(sq), 1e-12f);
    const float* d = dots + b * dsn + h * dsh + w * dsw;
    float* o = out + b * osn + h * osh + w * osw;
    for (int c = sub; c < a.nc; c += 16) {
      const int first = a.first[c], cnt = a.count[c];
      float m = -INFINITY;
      for (int j = 0; j < cnt; ++j) m = fmaxf(m, d[first + j] * inv * gamma);
      float den = 0.f, num = 0.f;
      for (int j = 0; j < cnt; ++j) {
        const float sj = d[first + j] * inv;
        const float e = expf(sj * gamma - m);
        den += e;
        num += e * sj;
      }
      if (okp) o[c] = num / den * gamma;
    }
  }
}

}  // namespace glsdet

using namespace glsdet;

extern "C" int glsdet_nchw_pack(const float* img, int32_t n, int32_t cin, int32_t H, int32_t W, const glsdet_view* y,
                                void* stream) {
  if (!img || !y) GLS_FAIL(GLSDET_E_ARG, "nchw_pack: null argument");
  int rc;
  if ((rc = check_view(*y, "nchw_pack.y"))) return rc;
  if (n < 1 || cin < 1 || y->n != n || y->h != H || y->w != W || y->c < cin || y->c % 8)
    GLS_FAIL(GLSDET_E_ARG, "nchw_pack: y must be [n,H,W,c>=cin], c a multiple of 8");
  const glsdet_view v = *y;
  OpRecord op;
  op.kind = 1;
  op.flops = 0;
  op.bytes = (double)n * H * W * (cin * 4.0 + v.c * dtype_size(v.dtype));
  op.name = "nchw_pack";
  op.launch = [=](hipStream_t st) -> int {
    const unsigned g = rgrid((long)n * H * W);
    if (v.dtype == GLSDET_F16)
      hipLaunchKernelGGL(nchw_pack_kernel<f16>, dim3(g), dim3(256), 0, st, img, n, cin, H, W, (unsigned char*)v.base, v.sn, v.sh, v.sw, v.c);
    else
      hipLaunchKernelGGL(nchw_pack_kernel<float>, dim3(g), dim3(256), 0, st, img, n, cin, H, W, (unsigned char*)v.base, v.sn, v.sh, v.sw, v.c);
    GLS_HIP(hipGetLastError());
    return 0;
  };
  return submit(std::move(op), stream);
}

extern "C" int glsdet_pool2d(const glsdet_view* x, const glsdet_view* y, int32_t k, int32_t stride, int32_t pad,
                             void* stream) {
  if (!x || !y) GLS_FAIL(GLSDET_E_ARG, "pool2d: null argument");
  int rc;
  if ((rc = check_view(*x, "pool2d.x"))) return rc;
  if ((rc = check_view(*y, "pool2d.y"))) return rc;
  if (k < 1 || k > 31 || stride < 1 || pad < 0 || 2 * pad > k) GLS_FAIL(GLSDET_E_ARG, "pool2d: bad k/stride/pad");
  if (x->dtype != y->dtype || x->n != y->n || x->c != y->c || x->c % 8 || y->h != (x->h + 2 * pad - k) / stride + 1 ||
      y->w != (x->w + 2 * pad - k) / stride + 1)
    GLS_FAIL(GLSDET_E_ARG, "pool2d: y must be [n,(h+2p-k)/s+1,(w+2p-k)/s+1,c] of x's dtype");
  const glsdet_view a = *x, b = *y;
  OpRecord op;
  op.kind = 2;
  op.flops = 0;
  op.bytes = ((double)a.n * a.h * a.w + (double)b.n * b.h * b.w) * a.c * dtype_size(a.dtype);
  op.name = "pool2d";
  op.launch = [=](hipStream_t st) -> int {
    const unsigned g = rgrid((long)b.n * b.h * b.w * (b.c * dtype_size(b.dtype) / 16));
    if (a.dtype == GLSDET_F16)
      hipLaunchKernelGGL(pool2d_kernel<f16>, dim3(g), dim3(256), 0, st, (const unsigned char*)a.base, a.sn, a.sh, a.sw, a.h, a.w, (unsigned char*)b.base, b.sn, b.sh, b.sw, b.n, b.h, b.w, b.c, k, stride, pad);
    else
      hipLaunchKernelGGL(pool2d_kernel<float>, dim3(g), dim3(256), 0, st, (const unsigned char*)a.base, a.sn, a.sh, a.sw, a.h, a.w, (unsigned char*)b.base, b.sn, b.sh, b.sw, b.n, b.h, b.w, b.c, k, stride, pad);
    GLS_HIP(hipGetLastError());
    return 0;
  };
  return submit(std::move(op), stream);
}

extern "C" int glsdet_upsample_add(const glsdet_view* coarse, const glsdet_view* fine, void* stream) {
  if (!coarse || !fine) GLS_FAIL(GLSDET_E_ARG, "upsample_add: null argument");
  int rc;
  if ((rc = check_view(*coarse, "upsample_add.coarse"))) return rc;
  if ((rc = check_view(*fine, "upsample_add.fine"))) return rc;
  if (coarse->dtype != fine->dtype || coarse->n != fine->n || coarse->c != fine->c || fine->c % 8)
    GLS_FAIL(GLSDET_E_ARG, "upsample_add: n, c, dtype must match; c a multiple of 8");
  const glsdet_view a = *coarse, b = *fine;
  OpRecord op;
  op.kind = 3;
  op.flops = 0;
  op.bytes = 3.0 * b.n * b.h * b.w * b.c * dtype_size(b.dtype);
  op.name = "upsample_add";
  op.launch = [=](hipStream_t st) -> int {
    const float sh = (float)a.h / (float)b.h, sw = (float)a.w / (float)b.w;
    const unsigned g = rgrid((long)b.n * b.h * b.w * (b.c * dtype_size(b.dtype) / 16));
    if (a.dtype == GLSDET_F16)
      hipLaunchKernelGGL(upsample_add_kernel<f16>, dim3(g), dim3(256), 0, st, (const unsigned char*)a.base, a.sn, a.sh, a.sw, a.h, a.w, (unsigned char*)b.base, b.sn, b.sh, b.sw, b.n, b.h, b.w, b.c, sh, sw);
    else
      hipLaunchKernelGGL(upsample_add_kernel<float>, dim3(g), dim3(256), 0, st, (const unsigned char*)a.base, a.sn, a.sh, a.sw, a.h, a.w, (unsigned char*)b.base, b.sn, b.sh, b.sw, b.n, b.h, b.w, b.c, sh, sw);
    GLS_HIP(hipGetLastError());
    return 0;
  };
  return submit(std::move(op), stream);
}

extern "C" int64_t glsdet_groupnorm_workspace_bytes(int32_t n, int32_t groups) {
  if (n < 1 || groups < 1) return 0;      // partial sums + (mean, rstd) per (image, group)
  return (int64_t)n * GLS_GN_SPLIT * groups * 2 * (int64_t)sizeof(double) + (int64_t)n * groups * 2 * (int64_t)sizeof(double);
}

static int groupnorm_sets(const glsdet_view* x, const glsdet_view* y, int32_t n_sets, int32_t groups,
                          const float* const* gamma, const float* const* beta, float eps, int32_t act,
                          void* stats, void* const* pre_stats, void* stream) {
  if (!x || !y || !gamma || !beta || !stats) GLS_FAIL(GLSDET_E_ARG, "groupnorm: null argument");
  if (n_sets < 1 || n_sets > GLS_GN_SETS) GLS_FAIL(GLSDET_E_ARG, "groupnorm: 1..%d sets", GLS_GN_SETS);
  if (act != GLSDET_ACT_NONE && act != GLSDET_ACT_RELU) GLS_FAIL(GLSDET_E_ARG, "groupnorm: act must be none or relu");
  if ((uintptr_t)stats & 7) GLS_FAIL(GLSDET_E_ALIGN, "groupnorm: stats must be 8-byte aligned");
  const int dt = x[0].dtype, C = x[0].c, nimg = x[0].n;
  const int vn = 16 / dtype_size(dt);
  if (groups < 1 || groups > 256 || C % groups || (C / groups) % vn || C / vn > 256 || 256 % (C / vn))
    GLS_FAIL(GLSDET_E_ARG, "groupnorm: need C %% groups == 0, (C/groups) %% %d == 0 and C/%d a divisor of 256 (C=%d groups=%d)",
             vn, vn, C, groups);
  GnArgs a = {};
  a.n = n_sets; a.C = C; a.groups = groups; a.act = act; a.eps = eps;
  const int rows = 256 / (C / vn);
  int max_split = 1, max_gb = 1;
  bool any_stats = false;
  OpRecord op;
  op.kind = 7;
  op.flops = op.bytes = 0;
  const long per_set = (long)nimg * GLS_GN_SPLIT * groups * 2 + (long)nimg * groups * 2;      // doubles (the tail holds mean, rstd as floats)
  for (int q = 0; q < n_sets; ++q) {
    int rc;
    if ((rc = check_view(x[q], "groupnorm.x"))) return rc;
    if ((rc = check_view(y[q], "groupnorm.y"))) return rc;
    if (!same_extent(x[q], y[q]) || x[q].dtype != dt || y[q].dtype != dt || x[q].c != C || x[q].n != nimg)
      GLS_FAIL(GLSDET_E_ARG, "groupnorm: set %d extent / dtype / channel mismatch", q);
    if (!gamma[q] || !beta[q]) GLS_FAIL(GLSDET_E_ARG, "groupnorm: null gamma/beta");
    GnSet& S = a.s[q];
    const int N = x[q].h * x[q].w;
    S.x = (const unsigned char*)x[q].base; S.y = (unsigned char*)y[q].base;
    S.xsn = x[q].sn; S.xsh = x[q].sh; S.xsw = x[q].sw;
    S.ysn = y[q].sn; S.ysh = y[q].sh; S.ysw = y[q].sw;
    S.gamma = gamma[q]; S.beta = beta[q];
    S.partial = (double*)stats + q * per_set;
    S.mr = (float*)((double*)stats + q * per_set + (long)nimg * GLS_GN_SPLIT * groups * 2);
    S.H = x[q].h; S.W = x[q].w;
    S.pstride = GLS_GN_SPLIT;
    S.pre = 0;
    S.nsplit = (N + 255) / 256;
    if (S.nsplit > GLS_GN_SPLIT) S.nsplit = GLS_GN_SPLIT;
    if (pre_stats && pre_stats[q]) {          // partials per 8 x 16 pixel tile, written by glsdet_conv2d_gnstats
      if ((uintptr_t)pre_stats[q] & 7) GLS_FAIL(GLSDET_E_ALIGN, "groupnorm: stats must be 8-byte aligned");
      S.partial = (double*)pre_stats[q];
      S.pre = 1;
      S.nsplit = S.pstride = ((x[q].h + 7) / 8) * ((x[q].w + 15) / 16) * 4;      // one slice per (tile, wave)
    }
    S.chunk = (N + S.nsplit - 1) / S.nsplit;
    S.gb = (N + rows * 8 - 1) / (rows * 8);
    if (S.gb > 1024) S.gb = 1024;
    if (!S.pre && S.nsplit > max_split) max_split = S.nsplit;
    if (S.gb > max_gb) max_gb = S.gb;
    if (!S.pre) any_stats = true;
    op.bytes += (S.pre ? 2.0 : 3.0) * nimg * N * C * dtype_size(dt);
  }
  op.name = !any_stats ? "groupnorm_multi(apply; stats by the convs)" : (n_sets > 1 ? "groupnorm_multi(stats+apply)" : "groupnorm(stats+apply)");
  op.launch = [=](hipStream_t st) -> int {
    if (dt == GLSDET_F16) {
      if (any_stats) hipLaunchKernelGGL(gn_stats_kernel<f16>, dim3(max_split, nimg, a.n), dim3(256), 0, st, a);
      hipLaunchKernelGGL(gn_fold_kernel, dim3(nimg, a.n), dim3(256), 0, st, a);
      hipLaunchKernelGGL(gn_apply_kernel<f16>, dim3(max_gb, nimg, a.n), dim3(256), 0, st, a);
    } else {
      if (any_stats) hipLaunchKernelGGL(gn_stats_kernel<float>, dim3(max_split, nimg, a.n), dim3(256), 0, st, a);
      hipLaunchKernelGGL(gn_fold_kernel, dim3(nimg, a.n), dim3(256), 0, st, a);
      hipLaunchKernelGGL(gn_apply_kernel<float>, dim3(max_gb, nimg, a.n), dim3(256), 0, st, a);
    }
    GLS_HIP(hipGetLastError());
    return 0;
  };
  return submit(std::move(op), stream);
}

extern "C" int glsdet_groupnorm_multi(const glsdet_view* x, const glsdet_view* y, int32_t n_sets, int32_t groups,
                                      const float* const* gamma, const float* const* beta, float eps, int32_t act,
                                      void* stats, void* stream) {
  return groupnorm_sets(x, y, n_sets, groups, gamma, beta, eps, act, stats, nullptr, stream);
}
extern "C" int glsdet_groupnorm_multi_pre(const glsdet_view* x, const glsdet_view* y, int32_t n_sets, int32_t groups,
                                          const float* const* gamma, const float* const* beta, float eps, int32_t act,
                                          void* stats, void* const* pre_stats, void* stream) {
  return groupnorm_sets(x, y, n_sets, groups, gamma, beta, eps, act, stats, pre_stats, stream);
}

extern "C" int glsdet_groupnorm(const glsdet_view* x, const glsdet_view* y, int32_t groups, const float* gamma,
                                const float* beta, float eps, int32_t act, void* stats, void* stream) {
  return glsdet_groupnorm_multi(x, y, 1, groups, &gamma, &beta, eps, act, stats, stream);
}

extern "C" int glsdet_proxy_scores(const glsdet_view* feat, const glsdet_view* dots, const int32_t* counts,
                                   int32_t num_classes, float gamma, const glsdet_view* out, void* stream) {
  if (!feat || !dots || !counts || !out) GLS_FAIL(GLSDET_E_ARG, "proxy_scores: null argument");
  int rc;
  if ((rc = check_view(*feat, "proxy_scores.feat"))) return rc;
  if ((rc = check_view(*dots, "proxy_scores.dots", false))) return rc;
  if ((rc = check_view(*out, "proxy_scores.out", false))) return rc;
  if (dots->dtype != GLSDET_F32 || out->dtype != GLSDET_F32) GLS_FAIL(GLSDET_E_ARG, "proxy_scores: dots/out must be fp32");
  if (num_classes < 1 || num_classes > 256) GLS_FAIL(GLSDET_E_ARG, "proxy_scores: bad num_classes");
  ProxyArgs pa = {};
  int P = 0;
  for (int c = 0; c < num_classes; ++c) {
    if (counts[c] < 1 || counts[c] > 64) GLS_FAIL(GLSDET_E_ARG, "proxy_scores: class %d has %d proxies (1..64)", c, counts[c]);
    if (P + counts[c] > 256) GLS_FAIL(GLSDET_E_ARG, "proxy_scores: more than 256 proxies");
    pa.first[c] = (unsigned char)P;
    if (counts[c] > pa.maxcnt) pa.maxcnt = counts[c];
    pa.count[c] = (unsigned char)counts[c];
    for (int k = 0; k < counts[c]; ++k) pa.cls_of[P + k] = (unsigned char)c;
    P += counts[c];
  }
  pa.P = P;
  pa.nc = num_classes;
  if (dots->c < P || out->c < num_classes || dots->n != feat->n || dots->h != feat->h || dots->w != feat->w ||
      out->n != feat->n || out->h != feat->h || out->w != feat->w || feat->c % 8)
    GLS_FAIL(GLSDET_E_ARG, "proxy_scores: extent mismatch (P=%d)", P);
  const glsdet_view a = *feat, d = *dots, o = *out;
  OpRecord op;
  op.kind = 8;
  op.flops = 0;
  op.bytes = (double)a.n * a.h * a.w * (a.c * dtype_size(a.dtype) + 4.0 * (P + num_classes));
  op.name = "proxy_scores";
  op.launch = [=](hipStream_t st) -> int {
    const unsigned g = rgrid((long)a.n * a.h * a.w * 16);
    if (a.dtype == GLSDET_F16)
      hipLaunchKernelGGL(proxy_scores_kernel<f16>, dim3(g), dim3(256), 0, st, (const unsigned char*)a.base, a.sn, a.sh, a.sw, (const float*)d.base, d.sn, d.sh, d.sw, (float*)o.base, o.sn, o.sh, o.sw, a.n, a.h, a.w, a.c, gamma, pa);
    else
      hipLaunchKernelGGL(proxy_scores_kernel<float>, dim3(g), dim3(256), 0, st, (const unsigned char*)a.base, a.sn, a.sh, a.sw, (const float*)d.base, d.sn, d.sh, d.sw, (float*)o.base, o.sn, o.sh, o.sw, a.n, a.h, a.w, a.c, gamma, pa);
    GLS_HIP(hipGetLastError());
    return 0;
  };
  return submit(std::move(op), stream);
}
