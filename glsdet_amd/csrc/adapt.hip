// Patch_Conv_NonLocal_adapt_new (drone/models/new/Non_local_family.py:272-357): the pieces around the non-local blocks
// whose shapes depend on DATA.  The reference finds its quadrant split with Python loops that synchronise with the device
// once per column / row (`if d.sum() > 0.5 * x.sum()`, :298-321); here one small kernel leaves the three split indices
// in device memory and every consumer (non-local windows: misc.hip nl_window; row masks / selects below) reads them
// there: no host round trip, the whole block stays capturable in a hipGraph.
#include "common.h"

namespace glsdet {

template <typename T>
__device__ __forceinline__ float ld1(const unsigned char* p, long off) { return (float)reinterpret_cast<const T*>(p)[off]; }

// One workgroup.  att: [n,H,W,>=1] view, channel 0 = the attention map.  split[0..2] = centroid_x (row split of the whole
// map), centroid_y of the rows above it, centroid_y of the rows from it on -- get_centroid's arithmetic: fp32 sums over the
// WHOLE batch, first index whose running sum exceeds half the total, rounded down to even, clamped to [4, size - 4].
template <typename T>
__global__ __launch_bounds__(1024) void attn_split_kernel(const unsigned char* att, long sn, long sh, long sw, int n, int H, int W,
                                                          int* __restrict__ split) {
  extern __shared__ float sm[];                 // rows [n][H], cols_top [n][W], cols_bot [n][W], scratch [2 * 1024]
  float* rows = sm;
  float* ctop = rows + n * H;
  float* cbot = ctop + n * W;
  float* red = cbot + n * W;
  __shared__ float s_thr;
  __shared__ int s_cx;
  const int tid = threadIdx.x;
  float mx = -INFINITY, mn = INFINITY;
  const long total_px = (long)n * H * W;
  for (long i = tid; i < total_px; i += 1024) {
    const int w = (int)(i % W), h = (int)((i / W) % H), b = (int)(i / ((long)W * H));
    const float v = ld1<T>(att, b * sn + h * sh + w * sw);
    mx = fmaxf(mx, v);
    mn = fminf(mn, v);
  }
  red[tid] = mx;
  red[1024 + tid] = mn;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if (tid < s) {
      red[tid] = fmaxf(red[tid], red[tid + s]);
      red[1024 + tid] = fminf(red[1024 + tid], red[1024 + tid + s]);
    }
    __syncthreads();
  }
  if (tid == 0) s_thr = red[1024] + 0.75f * (red[0] - red[1024]);       // min + 0.75 * (max - min), :325-327
  __syncthreads();
  const float thr = s_thr;
  auto masked = [&](int b, int h, int w) -> float {
    const float v = ld1<T>(att, b * sn + h * sh + w * sw);
    return v < thr ? 0.f : v;
  };
  for (int i = tid; i < n * H; i += 1024) {     // x.sum(3): per image and row
    const int b = i / H, h = i - b * H;
    float s = 0.f;
    for (int w = 0; w < W; ++w) s += masked(b, h, w);
    rows[i] = s;
  }
  __syncthreads();
  // first index with running sum over the batch > half the total; i // 2 * 2; clamp [4, size - 4]
  auto centroid = [&](const float* v, int len, int stride_b) -> int {   // v[b * stride_b + i]
    float total = 0.f;
    for (int b = 0; b < n; ++b)
      for (int i = 0; i < len; ++i) total += v[b * stride_b + i];
    int i = 0;
    float d[16];
    for (int b = 0; b < n && b < 16; ++b) d[b] = 0.f;
    for (i = 0; i < len; ++i) {
      float s = 0.f;
      for (int b = 0; b < n; ++b) {
        d[b & 15] += v[b * stride_b + i];
        s += d[b & 15];
      }
      if (s > 0.5f * total) break;
    }
    if (i == len) i = len - 1;                  // the Python loop variable after an unbroken loop
    i = i / 2 * 2;
    i = i < 4 ? 4 : i;
    return i > len - 4 ? len - 4 : i;
  };
  if (tid == 0) s_cx = centroid(rows, H, H);
  __syncthreads();
  const int cx = s_cx;
  for (int i = tid; i < n * W; i += 1024) {     // x.sum(2) of the rows above / from the split
    const int b = i / W, w = i - b * W;
    float st = 0.f, sb = 0.f;
    for (int h = 0; h < cx; ++h) st += masked(b, h, w);
    for (int h = cx; h < H; ++h) sb += masked(b, h, w);
    ctop[i] = st;
    cbot[i] = sb;
  }
  __syncthreads();
  if (tid == 0) {
    split[0] = cx;
    split[1] = centroid(ctop, W, W);
    split[2] = centroid(cbot, W, W);
    split[3] = 0;
  }
}

// Region masks / selects by the device-side split (indices >> shift: the regions of the stride-2 map).
// mode 0: y = a on the rows above the row split, 0 below; 1: y = a from the split on, 0 above; 2: y = row < split ? a : b;
// 3: y = a inside quadrant q (0 lt, 1 lb, 2 rt, 3 rb), 0 outside; 4: y = a inside quadrant q, y unchanged outside.
template <typename T>
__global__ __launch_bounds__(256) void rowsplit_kernel(const unsigned char* a, long asn, long ash, long asw, const unsigned char* b,
                                                       long bsn, long bsh, long bsw, unsigned char* y, long ysn, long ysh, long ysw,
                                                       int n, int H, int W, int cch, const int* __restrict__ split, int mode, int q,
                                                       int shift) {
  const int cx = split[0] >> shift, cyl = split[1] >> shift, cyr = split[2] >> shift;
  const long total = (long)n * H * W * cch;
  constexpr int VN = 16 / (int)sizeof(T);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int cc = (int)(i % cch);
    long p = i / cch;
    const int w = (int)(p % W);
    p /= W;
    const int h = (int)(p % H), img = (int)(p / H);
    const bool top = h < cx;
    const int quad = (top ? 0 : 1) + (w >= (top ? cyl : cyr) ? 2 : 0);
    uint4 v = {0u, 0u, 0u, 0u};
    const long ao = (img * asn + h * ash + w * asw + cc * VN) * (long)sizeof(T);
    const long yo = (img * ysn + h * ysh + w * ysw + cc * VN) * (long)sizeof(T);
    if (mode == 2) {
      v = top ? *reinterpret_cast<const uint4*>(a + ao)
              : *reinterpret_cast<const uint4*>(b + (img * bsn + h * bsh + w * bsw + cc * VN) * (long)sizeof(T));
    } else if (mode <= 1) {
      if ((mode == 0) == top) v = *reinterpret_cast<const uint4*>(a + ao);
    } else {
      if (quad == q) v = *reinterpret_cast<const uint4*>(a + ao);
      else if (mode == 4) continue;
    }
    *reinterpret_cast<uint4*>(y + yo) = v;
  }
}

// y[.., c] = m[.., 0] * x[.., c]   (attention_map * feat_patch, :355-356)
template <typename T>
__global__ __launch_bounds__(256) void scale_by_map_kernel(const unsigned char* x, long xsn, long xsh, long xsw, const unsigned char* m,
                                                           long msn, long msh, long msw, unsigned char* y, long ysn, long ysh, long ysw,
                                                           int n, int H, int W, int cch) {
  constexpr int VN = 16 / (int)sizeof(T);
  const long total = (long)n * H * W * cch;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int cc = (int)(i % cch);
    long p = i / cch;
    const int w = (int)(p % W);
    p /= W;
    const int h = (int)(p % H), img = (int)(p / H);
    const float g = (float)reinterpret_cast<const T*>(m)[img * msn + h * msh + w * msw];
    T v[VN];
    *reinterpret_cast<uint4*>(v) = *reinterpret_cast<const uint4*>(x + (img * xsn + h * xsh + w * xsw + cc * VN) * (long)sizeof(T));
#pragma unroll
    for (int e = 0; e < VN; ++e) v[e] = (T)(g * (float)v[e]);
    *reinterpret_cast<uint4*>(y + (img * ysn + h * ysh + w * ysw + cc * VN) * (long)sizeof(T)) = *reinterpret_cast<uint4*>(v);
  }
}

// mode 0: y[.., c] = a[.., c] * m[.., 0] + b[.., c] * m[.., 1]   (LSKblock: attn1 * sig[:, 0] + attn2 * sig[:, 1], LSK.py:46)
// mode 1: y = a * b elementwise                                   (LSKblock: x * attn, LSK.py:48)
template <typename T>
__global__ __launch_bounds__(256) void gate_kernel(const unsigned char* a, long asn, long ash, long asw, const unsigned char* b, long bsn,
                                                   long bsh, long bsw, const unsigned char* m, long msn, long msh, long msw,
                                                   unsigned char* y, long ysn, long ysh, long ysw, int n, int H, int W, int cch, int mode) {
  constexpr int VN = 16 / (int)sizeof(T);
  const long total = (long)n * H * W * cch;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int cc = (int)(i % cch);
    long p = i / cch;
    const int w = (int)(p % W);
    p /= W;
    const int h = (int)(p % H), img = (int)(p / H);
    T va[VN], vb[VN];
    *reinterpret_cast<uint4*>(va) = *reinterpret_cast<const uint4*>(a + (img * asn + h * ash + w * asw + cc * VN) * (long)sizeof(T));
    *reinterpret_cast<uint4*>(vb) = *reinterpret_cast<const uint4*>(b + (img * bsn + h * bsh + w * bsw + cc * VN) * (long)sizeof(T));
    if (mode == 0) {
      const T* mp = reinterpret_cast<const T*>(m) + img * msn + h * msh + w * msw;
      const float g0 = (float)mp[0], g1 = (float)mp[1];
#pragma unroll
      for (int e = 0; e < VN; ++e) va[e] = (T)((float)va[e] * g0 + (float)vb[e] * g1);
    } else {
#pragma unroll
      for (int e = 0; e < VN; ++e) va[e] = (T)((float)va[e] * (float)vb[e]);
    }
    *reinterpret_cast<uint4*>(y + (img * ysn + h * ysh + w * ysw + cc * VN) * (long)sizeof(T)) = *reinterpret_cast<uint4*>(va);
  }
}

static inline unsigned grid_of(long items) {
  long g = (items + 255) / 256;
  return (unsigned)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace glsdet

using namespace glsdet;

extern "C" int glsdet_attn_split(const glsdet_view* att, int32_t* split, void* stream) {
  if (!att || !split) GLS_FAIL(GLSDET_E_ARG, "attn_split: null argument");
  int rc;
  if ((rc = check_view(*att, "attn_split.att", false))) return rc;
  if (att->h < 8 || att->w < 8 || att->n < 1 || att->n > 16) GLS_FAIL(GLSDET_E_ARG, "attn_split: needs 1..16 images of at least 8 x 8");
  const size_t lds = (size_t)(att->n * (att->h + 2 * att->w) + 2048) * 4;
  if (lds > 60 * 1024) GLS_FAIL(GLSDET_E_ARG, "attn_split: map too large for one workgroup's sums");
  const glsdet_view a = *att;
  OpRecord op;
  op.kind = 2;
  op.flops = 0;
  op.bytes = 3.0 * a.n * a.h * a.w * dtype_size(a.dtype);
  op.name = "attn_split(threshold + centroids)";
  op.launch = [=](hipStream_t st) -> int {
    if (a.dtype == GLSDET_F16)
      hipLaunchKernelGGL(attn_split_kernel<f16>, dim3(1), dim3(1024), lds, st, (const unsigned char*)a.base, a.sn, a.sh, a.sw, a.n, a.h, a.w, split);
    else
      hipLaunchKernelGGL(attn_split_kernel<float>, dim3(1), dim3(1024), lds, st, (const unsigned char*)a.base, a.sn, a.sh, a.sw, a.n, a.h, a.w, split);
    GLS_HIP(hipGetLastError());
    return 0;
  };
  return submit(std::move(op), stream);
}

extern "C" int glsdet_rowsplit(const glsdet_view* a, const glsdet_view* b, const glsdet_view* y, const int32_t* split, int32_t mode,
                               void* stream) {
  const int q = (mode >> 4) & 3, shift = (mode >> 8) & 1;
  mode &= 15;
  if (!a || !y || !split || mode < 0 || mode > 4 || (mode == 2 && !b)) GLS_FAIL(GLSDET_E_ARG, "rowsplit: bad argument");
  int rc;
  if ((rc = check_view(*a, "rowsplit.a"))) return rc;
  if ((rc = check_view(*y, "rowsplit.y"))) return rc;
  if (!same_extent(*a, *y) || a->dtype != y->dtype) GLS_FAIL(GLSDET_E_ARG, "rowsplit: a / y mismatch");
  glsdet_view vb = *a;
  if (mode == 2) {
    if ((rc = check_view(*b, "rowsplit.b"))) return rc;
    if (!same_extent(*b, *y) || b->dtype != y->dtype) GLS_FAIL(GLSDET_E_ARG, "rowsplit: b / y mismatch");
    vb = *b;
  }
  const glsdet_view va = *a, vy = *y;
  OpRecord op;
  op.kind = 3;
  op.flops = 0;
  op.bytes = 2.0 * va.n * va.h * va.w * va.c * dtype_size(va.dtype);
  op.name = mode == 2 ? "rowsplit(select)" : (mode == 4 ? "rowsplit(merge quadrant)" : "rowsplit(mask)");
  op.launch = [=](hipStream_t st) -> int {
    const int vn = 16 / dtype_size(va.dtype);
    const unsigned g = grid_of((long)va.n * va.h * va.w * (va.c / vn));
    if (va.dtype == GLSDET_F16)
      hipLaunchKernelGGL(rowsplit_kernel<f16>, dim3(g), dim3(256), 0, st, (const unsigned char*)va.base, va.sn, va.sh, va.sw, (const unsigned char*)vb.base, vb.sn, vb.sh, vb.sw, (unsigned char*)vy.base, vy.sn, vy.sh, vy.sw, va.n, va.h, va.w, va.c / vn, split, mode, q, shift);
    else
      hipLaunchKernelGGL(rowsplit_kernel<float>, dim3(g), dim3(256), 0, st, (const unsigned char*)va.base, va.sn, va.sh, va.sw, (const unsigned char*)vb.base, vb.sn, vb.sh, vb.sw, (unsigned char*)vy.base, vy.sn, vy.sh, vy.sw, va.n, va.h, va.w, va.c / vn, split, mode, q, shift);
    GLS_HIP(hipGetLastError());
    return 0;
  };
  return submit(std::move(op), stream);
}

extern "C" int glsdet_scale_by_map(const glsdet_view* x, const glsdet_view* map, const glsdet_view* y, void* stream) {
  if (!x || !map || !y) GLS_FAIL(GLSDET_E_ARG, "scale_by_map: null argument");
  int rc;
  if ((rc = check_view(*x, "scale_by_map.x"))) return rc;
  if ((rc = check_view(*map, "scale_by_map.map", false))) return rc;
  if ((rc = check_view(*y, "scale_by_map.y"))) return rc;
  if (!same_extent(*x, *y) || x->dtype != y->dtype || map->dtype != x->dtype || map->n != x->n || map->h != x->h || map->w != x->w)
    GLS_FAIL(GLSDET_E_ARG, "scale_by_map: extent / dtype mismatch");
  const glsdet_view vx = *x, vm = *map, vy = *y;
  OpRecord op;
  op.kind = 3;
  op.flops = 0;
  op.bytes = 2.0 * vx.n * vx.h * vx.w * vx.c * dtype_size(vx.dtype);
  op.name = "scale_by_map";
  op.launch = [=](hipStream_t st) -> int {
    const int vn = 16 / dtype_size(vx.dtype);
    const unsigned g = grid_of((long)vx.n * vx.h * vx.w * (vx.c / vn));
    if (vx.dtype == GLSDET_F16)
      hipLaunchKernelGGL(scale_by_map_kernel<f16>, dim3(g), dim3(256), 0, st, (const unsigned char*)vx.base, vx.sn, vx.sh, vx.sw, (const unsigned char*)vm.base, vm.sn, vm.sh, vm.sw, (unsigned char*)vy.base, vy.sn, vy.sh, vy.sw, vx.n, vx.h, vx.w, vx.c / vn);
    else
      hipLaunchKernelGGL(scale_by_map_kernel<float>, dim3(g), dim3(256), 0, st, (const unsigned char*)vx.base, vx.sn, vx.sh, vx.sw, (const unsigned char*)vm.base, vm.sn, vm.sh, vm.sw, (unsigned char*)vy.base, vy.sn, vy.sh, vy.sw, vx.n, vx.h, vx.w, vx.c / vn);
    GLS_HIP(hipGetLastError());
    return 0;
  };
  return submit(std::move(op), stream);
}

extern "C" int glsdet_gate(const glsdet_view* a, const glsdet_view* b, const glsdet_view* map, const glsdet_view* y, int32_t mode,
                           void* stream) {
  if (!a || !b || !y || mode < 0 || mode > 1 || (mode == 0 && !map)) GLS_FAIL(GLSDET_E_ARG, "gate: bad argument");
  int rc;
  if ((rc = check_view(*a, "gate.a"))) return rc;
  if ((rc = check_view(*b, "gate.b"))) return rc;
  if ((rc = check_view(*y, "gate.y"))) return rc;
  if (!same_extent(*a, *y) || !same_extent(*b, *y) || a->dtype != y->dtype || b->dtype != y->dtype) GLS_FAIL(GLSDET_E_ARG, "gate: a / b / y mismatch");
  glsdet_view vm = *a;
  if (mode == 0) {
    if ((rc = check_view(*map, "gate.map", false))) return rc;
    if (map->n != y->n || map->h != y->h || map->w != y->w || map->c < 2 || map->dtype != y->dtype) GLS_FAIL(GLSDET_E_ARG, "gate: map must be [n,h,w,>=2] of y's dtype");
    vm = *map;
  }
  const glsdet_view va = *a, vb = *b, vy = *y;
  OpRecord op;
  op.kind = 3;
  op.flops = 0;
  op.bytes = 3.0 * va.n * va.h * va.w * va.c * dtype_size(va.dtype);
  op.name = mode == 0 ? "gate(a * m0 + b * m1)" : "gate(a * b)";
  op.launch = [=](hipStream_t st) -> int {
    const int vn = 16 / dtype_size(va.dtype);
    const unsigned g = grid_of((long)va.n * va.h * va.w * (va.c / vn));
    if (va.dtype == GLSDET_F16)
      hipLaunchKernelGGL(gate_kernel<f16>, dim3(g), dim3(256), 0, st, (const unsigned char*)va.base, va.sn, va.sh, va.sw, (const unsigned char*)vb.base, vb.sn, vb.sh, vb.sw, (const unsigned char*)vm.base, vm.sn, vm.sh, vm.sw, (unsigned char*)vy.base, vy.sn, vy.sh, vy.sw, va.n, va.h, va.w, va.c / vn, mode);
    else
      hipLaunchKernelGGL(gate_kernel<float>, dim3(g), dim3(256), 0, st, (const unsigned char*)va.base, va.sn, va.sh, va.sw, (const unsigned char*)vb.base, vb.sn, vb.sh, vb.sw, (const unsigned char*)vm.base, vm.sn, vm.sh, vm.sw, (unsigned char*)vy.base, vy.sn, vy.sh, vy.sw, va.n, va.h, va.w, va.c / vn, mode);
    GLS_HIP(hipGetLastError());
    return 0;
  };
  return submit(std::move(op), stream);
}
