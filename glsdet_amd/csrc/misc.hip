// HBM-bound helpers of the forward pass: Focus space-to-depth, SPP max pool, nearest
// resample / strided copy, and the non-local block (re-associated dot-product form).
// All are coalesced 16-byte-per-lane NHWC kernels; none of them is GEMM shaped enough to
// deserve MFMA (the non-local block is 0.35 % of the model's MACs before re-association).
#include <algorithm>
#include <vector>

#include "common.h"

namespace glsdet {

template <typename T> struct Vec16;
template <> struct Vec16<f16> { typedef f16x8 type; static constexpr int N = 8; };
template <> struct Vec16<float> { typedef f32x4 type; static constexpr int N = 4; };

// ---------------------------------------------------------------- Focus space-to-depth
// drone/models/base/darknet.py:15-21 : cat(TL, BL, TR, BR) on channels.
// One thread per output pixel: 2*cin float2 loads (a 2x2 patch of every channel), the
// 16-channel fp16 pixel leaves as two 16-byte stores (fp32: four).
template <typename T>
__global__ __launch_bounds__(256) void focus_pack_kernel(const float* __restrict__ img, int n, int cin, int H, int W,
                                                         unsigned char* y, long sn, long sh, long sw, int cy) {
  const int Ho = H >> 1, Wo = W >> 1;
  const long total = (long)n * Ho * Wo;
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (long)gridDim.x * blockDim.x) {
    const int wo = (int)(p % Wo);
    const int ho = (int)((p / Wo) % Ho);
    const int b = (int)(p / ((long)Wo * Ho));
    T* out = reinterpret_cast<T*>(y) + b * sn + ho * sh + wo * sw;
    if (cin == 3 && cy == 16) {
      float v[16];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float* r0 = img + (((long)b * 3 + c) * H + 2 * ho) * W + 2 * wo;
        const float2 t = *reinterpret_cast<const float2*>(r0);          // TL, TR
        const float2 u = *reinterpret_cast<const float2*>(r0 + W);      // BL, BR
        v[0 + c] = t.x;   // TL
        v[3 + c] = u.x;   // BL
        v[6 + c] = t.y;   // TR
        v[9 + c] = u.y;   // BR
      }
      v[12] = v[13] = v[14] = v[15] = 0.f;
      typedef typename Vec16<T>::type V;
      constexpr int VN = Vec16<T>::N;
#pragma unroll
      for (int q = 0; q < 16 / VN; ++q) {
        V pk;
#pragma unroll
        for (int e = 0; e < VN; ++e) pk[e] = (T)v[q * VN + e];
        reinterpret_cast<V*>(out)[q] = pk;
      }
      continue;
    }
    int ch = 0;
    // patch order: (dy,dx) = (0,0) TL, (1,0) BL, (0,1) TR, (1,1) BR
    for (int patch = 0; patch < 4; ++patch) {
      const int dy = patch & 1, dx = patch >> 1;
      for (int c = 0; c < cin; ++c, ++ch)
        out[ch] = (T)img[(((long)b * cin + c) * H + (2 * ho + dy)) * W + (2 * wo + dx)];
    }
    for (; ch < cy; ++ch) out[ch] = (T)0.0f;
  }
}

// ---------------------------------------------------------------- max pool k x k, s1
template <typename T>
__global__ __launch_bounds__(256) void maxpool_kernel(const unsigned char* x, long xsn, long xsh, long xsw,
                                                      unsigned char* y, long ysn, long ysh, long ysw, int n, int H,
                                                      int W, int C, int k) {
  typedef typename Vec16<T>::type V;
  constexpr int VN = Vec16<T>::N;
  const int cchunks = C / VN;
  const long total = (long)n * H * W * cchunks;
  const int r = k / 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % cchunks);
    long p = i / cchunks;
    const int w = (int)(p % W);
    p /= W;
    const int h = (int)(p % H);
    const int b = (int)(p / H);
    V m;
#pragma unroll
    for (int e = 0; e < VN; ++e) m[e] = (T)(-INFINITY);
    for (int dy = -r; dy <= r; ++dy) {
      const int hh = h + dy;
      if (hh < 0 || hh >= H) continue;
      for (int dx = -r; dx <= r; ++dx) {
        const int ww = w + dx;
        if (ww < 0 || ww >= W) continue;
        const V v = *reinterpret_cast<const V*>(x + (b * xsn + hh * xsh + ww * xsw + cc * VN) * (long)sizeof(T));
#pragma unroll
        for (int e = 0; e < VN; ++e) m[e] = v[e] > m[e] ? v[e] : m[e];
      }
    }
    *reinterpret_cast<V*>(y + (b * ysn + h * ysh + w * ysw + cc * VN) * (long)sizeof(T)) = m;
  }
}

// ---------------------------------------------------------------- SPP: the 5 / 9 / 13 max pools in one launch
// SPPBottleneck (drone/models/base/darknet.py:29,35): cat(x, pool5(x), pool9(x), pool13(x)).  A stride-1 max pool of 9 is
// pool5 of pool5, 13 is pool5 of pool9 (exact, -inf padding each time): a workgroup takes an 8 x 16 output tile of one
// 16-byte channel chunk, stages the (8+12) x (16+12) input patch in LDS and runs the three pools as separable row / column
// passes on shrinking regions, writing each pool's centre 8 x 16 to its output.  One read of x instead of three chained
// launches each re-reading the previous pool through L2.
template <typename T>
__global__ __launch_bounds__(256) void spp_pools_kernel(const unsigned char* x, long xsn, long xsh, long xsw, unsigned char* y5,
                                                        unsigned char* y9, unsigned char* y13, long ysn, long ysh, long ysw, int H,
                                                        int W, int cchunks, int tiles_x, int tiles_y) {
  typedef typename Vec16<T>::type V;
  constexpr int VN = Vec16<T>::N;
  constexpr int TH = 8, TW = 16, R = 6, PH = TH + 2 * R, PW = TW + 2 * R;
  __shared__ V sA[PH * PW], sB[PH * PW];
  int t = blockIdx.x;
  const int cc = t % cchunks;
  t /= cchunks;
  const int tx = t % tiles_x;
  t /= tiles_x;
  const int ty = t % tiles_y, b = t / tiles_y;
  const int y0 = ty * TH - R, x0 = tx * TW - R;            // image coordinates of patch (0, 0)
  V ninf;
#pragma unroll
  for (int e = 0; e < VN; ++e) ninf[e] = (T)(-INFINITY);
  auto vmax = [](V a, V c) { V r; for (int e = 0; e < VN; ++e) r[e] = a[e] > c[e] ? a[e] : c[e]; return r; };
  for (int q = threadIdx.x; q < PH * PW; q += 256) {
    const int py = q / PW, px = q - py * PW;
    const int h = y0 + py, w = x0 + px;
    V v = ninf;
    if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W)
      v = *reinterpret_cast<const V*>(x + (b * xsn + h * xsh + w * xsw + cc * VN) * (long)sizeof(T));
    sA[q] = v;
  }
  __syncthreads();
  unsigned char* outs[3] = {y5, y9, y13};
#pragma unroll
  for (int stage = 0; stage < 3; ++stage) {
    const int m = 2 * (stage + 1);                          // the valid region shrinks by 2 on every side per pool
    // rows: sB[py][px] = max over px-2..px+2 of sA, for py in [m-2, PH-m+2), px in [m, PW-m)
    for (int q = threadIdx.x; q < PH * PW; q += 256) {
      const int py = q / PW, px = q - py * PW;
      if (py >= m - 2 && py < PH - m + 2 && px >= m && px < PW - m) {
        V v = sA[q - 2];
        v = vmax(v, sA[q - 1]); v = vmax(v, sA[q]); v = vmax(v, sA[q + 1]); v = vmax(v, sA[q + 2]);
        sB[q] = v;
      }
    }
    __syncthreads();
    // columns, and -inf outside the image (the next pool pads with -inf again)
    for (int q = threadIdx.x; q < PH * PW; q += 256) {
      const int py = q / PW, px = q - py * PW;
      if (py >= m && py < PH - m && px >= m && px < PW - m) {
        V v = sB[q - 2 * PW];
        v = vmax(v, sB[q - PW]); v = vmax(v, sB[q]); v = vmax(v, sB[q + PW]); v = vmax(v, sB[q + 2 * PW]);
        const int h = y0 + py, w = x0 + px;
        const bool inside = (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
        sA[q] = inside ? v : ninf;
        if (inside && py >= R && py < R + TH && px >= R && px < R + TW)
          *reinterpret_cast<V*>(outs[stage] + (b * ysn + h * ysh + w * ysw + cc * VN) * (long)sizeof(T)) = v;
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------- channel max / mean
// SpatialAttention (Non_local_family.py:429-432).  One wave per pixel: lanes stride the
// channel chunks (16 B each), max and sum reduced across the wave with shuffles.
template <typename T>
__global__ __launch_bounds__(256) void channel_maxmean_kernel(const unsigned char* x, long xsn, long xsh, long xsw,
                                                              unsigned char* y, long ysn, long ysh, long ysw, int n,
                                                              int H, int W, int C) {
  typedef typename Vec16<T>::type V;
  constexpr int VN = Vec16<T>::N;
  const int lane = threadIdx.x & 63;
  const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  const long total = (long)n * H * W;
  for (long p = wave; p < total; p += nwaves) {
    const int w = (int)(p % W);
    const int h = (int)((p / W) % H);
    const int b = (int)(p / ((long)W * H));
    const unsigned char* px = x + (b * xsn + h * xsh + w * xsw) * (long)sizeof(T);
    float mx = -INFINITY, sm = 0.f;
    for (int c = lane * VN; c < C; c += 64 * VN) {
      const V v = *reinterpret_cast<const V*>(px + c * (long)sizeof(T));
#pragma unroll
      for (int e = 0; e < VN; ++e) {
        const float f = (float)v[e];
        mx = fmaxf(mx, f);
        sm += f;
      }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      mx = fmaxf(mx, __shfl_xor(mx, o, 64));
      sm += __shfl_xor(sm, o, 64);
    }
    if (lane == 0) {
      V out;
#pragma unroll
      for (int e = 0; e < VN; ++e) out[e] = (T)0.f;
      out[0] = (T)mx;
      out[1] = (T)(sm / (float)C);
      T* o = reinterpret_cast<T*>(y) + b * ysn + h * ysh + w * ysw;
      *reinterpret_cast<V*>(o) = out;
      if (VN == 4) {
        V z;
#pragma unroll
        for (int e = 0; e < VN; ++e) z[e] = (T)0.f;
        *reinterpret_cast<V*>(o + 4) = z;
      }
    }
  }
}

// ---------------------------------------------------------------- nearest resample / copy
template <typename T>
__global__ __launch_bounds__(256) void resample_kernel(const unsigned char* x, long xsn, long xsh, long xsw,
                                                       unsigned char* y, long ysn, long ysh, long ysw, int n, int Ho,
                                                       int Wo, int C, int f) {
  constexpr int VN = Vec16<T>::N;
  const int cchunks = C / VN;
  const long total = (long)n * Ho * Wo * cchunks;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % cchunks);
    long p = i / cchunks;
    const int w = (int)(p % Wo);
    p /= Wo;
    const int h = (int)(p % Ho);
    const int b = (int)(p / Ho);
    const uint4 v = *reinterpret_cast<const uint4*>(x + (b * xsn + (h / f) * xsh + (w / f) * xsw + cc * VN) * (long)sizeof(T));
    *reinterpret_cast<uint4*>(y + (b * ysn + h * ysh + w * ysw + cc * VN) * (long)sizeof(T)) = v;
  }
}


// the same strided copy for up to GLS_COPY_JOBS independent (source, destination) pairs of one launch: blockIdx.y = pair
// (the dense per-(image, quadrant) operand copies of the ResNet GL plug-in were 96 launches of 7 us per step)
#define GLS_COPY_JOBS 32
struct CopyJob {
  const unsigned char* x;
  unsigned char* y;
  long xsn, xsh, xsw, ysn, ysh, ysw;
  int n, h, w;
};
struct CopyManyArgs {
  CopyJob j[GLS_COPY_JOBS];
  int C;
};
template <typename T>
__global__ __launch_bounds__(256) void copy_many_kernel(CopyManyArgs a) {
  constexpr int VN = Vec16<T>::N;
  const CopyJob& jb = a.j[blockIdx.y];
  const int cchunks = a.C / VN;
  const long total = (long)jb.n * jb.h * jb.w * cchunks;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % cchunks);
    long p = i / cchunks;
    const int w = (int)(p % jb.w);
    p /= jb.w;
    const int h = (int)(p % jb.h);
    const int b = (int)(p / jb.h);
    const uint4 v = *reinterpret_cast<const uint4*>(jb.x + (b * jb.xsn + h * jb.xsh + w * jb.xsw + cc * VN) * (long)sizeof(T));
    *reinterpret_cast<uint4*>(jb.y + (b * jb.ysn + h * jb.ysh + w * jb.ysw + cc * VN) * (long)sizeof(T)) = v;
  }
}


// transposed form of the same: y[i] is a dense matrix [rows = channels][cols = pixels of x[i]] (row pitch ysw), i.e. the
// `.view(b, c, -1)` operand of the non-local products with the pixel index contiguous; 64 pixels x 64 channels per
// workgroup through LDS.  Columns [N, ceil_vec(N)) are written as zeros, later ones are left alone.
template <typename T>
__global__ __launch_bounds__(256) void transpose_many_kernel(CopyManyArgs a) {
  constexpr int VN = Vec16<T>::N;
  constexpr int PITCH = sizeof(T) == 2 ? 66 : 65;      // 33 / 65 words per tile row: the strided reads below hit distinct banks
  __shared__ T tile[64 * PITCH];
  const CopyJob& jb = a.j[blockIdx.z];
  const int N = jb.h * jb.w;
  const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  if (p0 >= N) return;                                  // whole-workgroup exit, before any barrier
  constexpr int CV = 64 / VN;                           // vectors per pixel row of the tile
  for (int i = threadIdx.x; i < 64 * CV; i += 256) {
    const int px = i / CV, cv = i % CV;
    const int p = p0 + px, c = c0 + cv * VN;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (p < N && c < a.C) v = *reinterpret_cast<const uint4*>(jb.x + ((long)(p / jb.w) * jb.xsh + (long)(p % jb.w) * jb.xsw + c) * (long)sizeof(T));
    const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
    for (int k = 0; k < VN; ++k) tile[px * PITCH + cv * VN + k] = e[k];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * CV; i += 256) {
    const int ch = i / CV, pg = i % CV;
    const int c = c0 + ch, p = p0 + pg * VN;
    if (c >= a.C || p >= N) continue;
    uint4 v;
    T* e = reinterpret_cast<T*>(&v);
#pragma unroll
    for (int k = 0; k < VN; ++k) e[k] = tile[(pg * VN + k) * PITCH + ch];
    *reinterpret_cast<uint4*>(jb.y + ((long)c * jb.ysw + p) * (long)sizeof(T)) = v;
  }
}

// ---------------------------------------------------------------- non-local block
// Gp[z][b][c1][c2] = sum_{j in slice z} phi[b,j,c1] * g[b,j,c2]   (16x16 block per workgroup;
// the position range is split over blockIdx.z for parallelism, partial sums are added in a
// fixed order by nl_fold_kernel -> deterministic)
// Up to GLS_NL_SETS independent blocks (the four quadrants of Patch_Conv_NonLocal) per launch:
// every kernel below takes the per-set operands in one kernarg struct and finds its set from
// the block index.
#define GLS_NL_SETS 4
struct NlSet {
  const unsigned char *x, *tpg;
  unsigned char* out;
  long xsn, xsh, xsw, tsn, tsh, tsw, osn, osh, osw;
  const float *wout, *bout;
  float *gram, *P;
  int H, W, nsplit, jchunk;
  float invN;
};
struct NlArgs {
  NlSet s[GLS_NL_SETS];
  int n, nimg, ci, cx, kc;
  // data-dependent quadrants (Patch_Conv_NonLocal_adapt_new): split = device int32 {row split, column split of the top
  // part, of the bottom part} written by glsdet_attn_split; every set's views then span the FULL FH x FW map and the
  // kernels cut their own window (sets in the order lt, lb, rt, rb).  nullptr: the static windows of the views.
  const int* split;
  int FH, FW, shift;          // shift 1: the windows live on the stride-2 map of the split's map (indices halved)
};
struct NlWin { int H, W, nsplit, jchunk; long xo, to, oo; float invN; };
__device__ __forceinline__ NlWin nl_window(const NlArgs& a, int q) {
  const NlSet& S = a.s[q];
  NlWin w = {S.H, S.W, S.nsplit, S.jchunk, 0, 0, 0, S.invN};
  if (a.split) {
    const int cx = a.split[0] >> a.shift, cyl = a.split[1] >> a.shift, cyr = a.split[2] >> a.shift;
    const bool bottom = q & 1, right = q >> 1;
    const int c = bottom ? cyr : cyl;
    const int r0 = bottom ? cx : 0, c0 = right ? c : 0;
    w.H = bottom ? a.FH - cx : cx;
    w.W = right ? a.FW - c : c;
    w.xo = r0 * S.xsh + c0 * S.xsw;
    w.to = r0 * S.tsh + c0 * S.tsw;
    w.oo = r0 * S.osh + c0 * S.osw;
    const int N = w.H * w.W;
    w.nsplit = 8;
    w.jchunk = (((N + 7) / 8) + 63) / 64 * 64;
    w.invN = 1.0f / (float)N;
  }
  return w;
}

template <typename T>
__global__ __launch_bounds__(256) void nl_gram_kernel(const NlArgs a) {
  __shared__ float ph[64][17], gg[64][17];
  const int q = blockIdx.z >> 3, z = blockIdx.z & 7;
  const NlSet& S = a.s[q];
  const NlWin wn = nl_window(a, q);
  if (z >= wn.nsplit) return;
  const int ci = a.ci;
  const int nb = (ci + 15) / 16;
  const int c1_0 = (blockIdx.x / nb) * 16, c2_0 = (blockIdx.x % nb) * 16;
  const int b = blockIdx.y;
  const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
  const int W = wn.W, N = wn.H * W;
  const int jbeg = z * wn.jchunk, jend = min(N, jbeg + wn.jchunk);
  const T* base = reinterpret_cast<const T*>(S.tpg) + b * S.tsn + wn.to;
  float acc = 0.f;
  constexpr int VN = 16 / (int)sizeof(T);          // channels per 16-byte chunk
  // whole 16-channel blocks of 16-byte-aligned rows: one vector load per thread and tile instead of 2-byte loads (thread t:
  // matrix t / 128 (phi | g), position (t % 128) / CPB, chunk (t % 128) % CPB; fp32: two passes)
  const bool vec = (ci % 16) == 0 && (S.tsw % VN) == 0 && (S.tsh % VN) == 0 && (S.tsn % VN) == 0 && (wn.to % VN) == 0 &&
                   ((uintptr_t)S.tpg & 15) == 0;
  for (int j0 = jbeg; j0 < jend; j0 += 64) {
    if (vec) {
      constexpr int CPB = 16 / VN;                 // chunks per 16-channel block: 2 (fp16) / 4 (fp32)
#pragma unroll
      for (int e = 0; e < CPB / 2; ++e) {
        const int t = threadIdx.x + e * 256;
        const int mat = t / (64 * CPB), r = t - mat * 64 * CPB;
        const int jj = r / CPB, ch = r - jj * CPB;
        const int j = j0 + jj;
        typename Vec16<T>::type v;
#pragma unroll
        for (int k = 0; k < VN; ++k) v[k] = (T)0.f;
        if (j < jend) {
          const T* px = base + (j / W) * S.tsh + (j % W) * S.tsw;
          v = *reinterpret_cast<const typename Vec16<T>::type*>(px + (mat ? 2 * ci + c2_0 : ci + c1_0) + ch * VN);
        }
        float(*dst)[17] = mat ? gg : ph;
#pragma unroll
        for (int k = 0; k < VN; ++k) dst[jj][ch * VN + k] = (float)v[k];
      }
    } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int idx = threadIdx.x + e * 256;
      const int jj = idx >> 4, cc = idx & 15;
      const int j = j0 + jj;
      float vp = 0.f, vg = 0.f;
      if (j < jend) {
        const T* px = base + (j / W) * S.tsh + (j % W) * S.tsw;
        if (c1_0 + cc < ci) vp = (float)px[ci + c1_0 + cc];
        if (c2_0 + cc < ci) vg = (float)px[2 * ci + c2_0 + cc];
      }
      ph[jj][cc] = vp;
      gg[jj][cc] = vg;
    }
    }
    __syncthreads();
#pragma unroll 16
    for (int jj = 0; jj < 64; ++jj) acc += ph[jj][ty] * gg[jj][tx];
    __syncthreads();
  }
  if (c1_0 + ty < ci && c2_0 + tx < ci)
    S.gram[(((long)z * a.nimg + b) * ci + c1_0 + ty) * ci + c2_0 + tx] = acc;
}

// P[b][co][c1] = (1/N) sum_c2 Wout[co][c2] * (sum_z Gp[z][b][c1][c2]).  Workgroup = (image,
// 16 output channels, 64 rows c1).  The Gram rows are walked in c2 chunks of kc (<= 128)
// columns: the partial slices of a chunk are summed into LDS with all 8 slice loads of an
// element in flight (fixed order -> deterministic); a wave owns 4 output channels, so its
// weight rows are wave-uniform (scalar loads) while lanes are the c1 rows (stride kc+1,
// conflict-free).  Any ci works: LDS holds one chunk, never the whole Gram.
#define GLS_FOLD_CO 16
#define GLS_NL_KC 128
__global__ __launch_bounds__(256) void nl_fold_kernel(const NlArgs a) {
  extern __shared__ float gs[];   // [64][kc+1]
  const int q = blockIdx.x / a.nimg, b = blockIdx.x - q * a.nimg;
  const NlSet& S = a.s[q];
  const float* __restrict__ Gp = S.gram;
  const float* __restrict__ wout = S.wout;
  float* P = S.P;
  const NlWin wn = nl_window(a, q);
  const int nsplit = wn.nsplit, nimg = a.nimg, ci = a.ci, cx = a.cx, kc = a.kc;
  const float invN = wn.invN;
  const int co0 = blockIdx.y * GLS_FOLD_CO, c1_0 = blockIdx.z * 64, ld = kc + 1;
  const long slice = (long)nimg * ci * ci;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* w[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) w[q] = wout + (long)min(co0 + wave * 4 + q, cx - 1) * ci;   // wave-uniform rows
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int c2_0 = 0; c2_0 < ci; c2_0 += kc) {
    const int kn = min(kc, ci - c2_0);
    for (int e = threadIdx.x; e < 64 * kc; e += 256) {
      const int r = e / kc, c = e - r * kc;
      float a = 0.f;
      if (c1_0 + r < ci && c < kn) {
        const float* g = Gp + ((long)b * ci + c1_0 + r) * ci + c2_0 + c;
        float v[8];
#pragma unroll
        for (int z = 0; z < 8; ++z) v[z] = z < nsplit ? g[z * slice] : 0.f;
        a = v[0];
#pragma unroll
        for (int z = 1; z < 8; ++z) a += v[z];
      }
      gs[r * ld + c] = a;
    }
    __syncthreads();
    const float* g = gs + lane * ld;
#pragma unroll 8
    for (int c = 0; c < kn; ++c) {
      const float gv = g[c];
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] += w[q][c2_0 + c] * gv;
    }
    __syncthreads();
  }
  const int c1 = c1_0 + lane;
  if (c1 < ci) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int co = co0 + wave * 4 + q;
      if (co < cx) P[((long)b * cx + co) * ci + c1] = acc[q] * invN;
    }
  }
}

// out[b,i,co] = x[b,i,co] + bout[co] + sum_c1 theta[b,i,c1] * P[b][co][c1]
// Workgroup = 64 pixels x 32 output channels of one image.  c1 is walked in chunks of kc:
// a theta tile [64][kc+1] and the P rows [32][kc] live in LDS; lanes are pixels (theta rows
// conflict-free), a wave owns 8 output channels (P reads are broadcasts), whose 8
// accumulators share each theta read and persist across the chunks.
#define GLS_APPLY_CO 32
template <typename T>
__global__ __launch_bounds__(256) void nl_apply_kernel(const NlArgs a) {
  extern __shared__ float th[];   // [64][kc+1] theta, then [GLS_APPLY_CO][kc] P rows
  const int q = blockIdx.y / a.nimg, b = blockIdx.y - q * a.nimg;
  const NlSet& S = a.s[q];
  const unsigned char *x = S.x, *tpg = S.tpg;
  unsigned char* out = S.out;
  const long xsn = S.xsn, xsh = S.xsh, xsw = S.xsw, tsn = S.tsn, tsh = S.tsh, tsw = S.tsw, osn = S.osn, osh = S.osh,
             osw = S.osw;
  const float* __restrict__ P = S.P;
  const float* __restrict__ bout = S.bout;
  const NlWin wn = nl_window(a, q);
  const int W = wn.W, ci = a.ci, cx = a.cx, kc = a.kc;
  const int N = wn.H * W;
  const int j0 = blockIdx.x * 64, co0 = blockIdx.z * GLS_APPLY_CO;
  if (j0 >= N) return;            // block-uniform: this set is smaller than the largest one
  const int ld = kc + 1;
  float* pr = th + 64 * ld;
  const int jj = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int j = j0 + jj;
  constexpr int VN = 16 / (int)sizeof(T);
  // 16-byte loads / stores where the views allow it (rows and windows on 16-byte boundaries)
  const bool vec_t = (tsw % VN) == 0 && (tsh % VN) == 0 && (tsn % VN) == 0 && (wn.to % VN) == 0 && ((uintptr_t)tpg & 15) == 0;
  const bool vec_o = (xsw % VN) == 0 && (xsh % VN) == 0 && (xsn % VN) == 0 && (wn.xo % VN) == 0 && ((uintptr_t)x & 15) == 0 &&
                     (osw % VN) == 0 && (osh % VN) == 0 && (osn % VN) == 0 && (wn.oo % VN) == 0 && ((uintptr_t)out & 15) == 0 &&
                     (cx % VN) == 0 && kc >= GLS_APPLY_CO;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int c0 = 0; c0 < ci; c0 += kc) {
    const int kn = min(kc, ci - c0);
    if (vec_t && (kn % VN) == 0) {                   // theta tile by 16-byte loads
      const int cpr = kn / VN;
      for (int idx = threadIdx.x; idx < 64 * cpr; idx += 256) {
        const int r = idx / cpr, ch = idx - r * cpr;
        const int jr = j0 + r;
        typename Vec16<T>::type v;
#pragma unroll
        for (int k = 0; k < VN; ++k) v[k] = (T)0.f;
        if (jr < N)
          v = *reinterpret_cast<const typename Vec16<T>::type*>(reinterpret_cast<const T*>(tpg) + b * tsn + wn.to + (jr / W) * tsh + (jr % W) * tsw + c0 + ch * VN);
#pragma unroll
        for (int k = 0; k < VN; ++k) th[r * ld + ch * VN + k] = (float)v[k];
      }
    } else
    for (int idx = threadIdx.x; idx < 64 * kn; idx += 256) {
      const int r = idx / kn, c = idx - r * kn;
      const int jr = j0 + r;
      float v = 0.f;
      if (jr < N) v = (float)(reinterpret_cast<const T*>(tpg) + b * tsn + wn.to + (jr / W) * tsh + (jr % W) * tsw)[c0 + c];
      th[r * ld + c] = v;
    }
    for (int idx = threadIdx.x; idx < GLS_APPLY_CO * kn; idx += 256) {
      const int r = idx / kn, c = idx - r * kn;
      pr[r * kc + c] = (co0 + r) < cx ? P[((long)b * cx + co0 + r) * ci + c0 + c] : 0.f;
    }
    __syncthreads();
    const float* t = th + jj * ld;
    const float* p0 = pr + grp * 8 * kc;
#pragma unroll 4
    for (int c = 0; c < kn; ++c) {
      const float tv = t[c];
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q] += tv * p0[q * kc + c];
    }
    __syncthreads();
  }
  if (vec_o) {
    // the accumulators go through LDS ([64 pixels][32 channels + 1], the theta tile's space: 64 * (kc + 1) floats with
    // kc >= 32) so that a thread adds x + bout to ONE 16-byte chunk of one pixel and stores it whole; lanes that hold pixels
    // wrote 64 two-byte stores per instruction before
    float* so = th;
#pragma unroll
    for (int q = 0; q < 8; ++q) so[jj * (GLS_APPLY_CO + 1) + grp * 8 + q] = acc[q];
    __syncthreads();
    constexpr int CPP = GLS_APPLY_CO / VN;          // chunks per pixel
    for (int idx = threadIdx.x; idx < 64 * CPP; idx += 256) {
      const int r = idx / CPP, ch = idx - r * CPP;
      const int jr = j0 + r, co = co0 + ch * VN;
      if (jr >= N || co >= cx) continue;
      const long px_x = b * xsn + wn.xo + (jr / W) * xsh + (jr % W) * xsw + co;
      const long px_o = b * osn + wn.oo + (jr / W) * osh + (jr % W) * osw + co;
      typename Vec16<T>::type v = *reinterpret_cast<const typename Vec16<T>::type*>(reinterpret_cast<const T*>(x) + px_x);
#pragma unroll
      for (int k = 0; k < VN; ++k) v[k] = (T)((float)v[k] + bout[co + k] + so[r * (GLS_APPLY_CO + 1) + ch * VN + k]);
      *reinterpret_cast<typename Vec16<T>::type*>(reinterpret_cast<T*>(out) + px_o) = v;
    }
    return;
  }
  if (j >= N) return;
  const long poff_x = b * xsn + wn.xo + (j / W) * xsh + (j % W) * xsw;
  const long poff_o = b * osn + wn.oo + (j / W) * osh + (j % W) * osw;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int co = co0 + grp * 8 + q;
    if (co < cx) {
      const float xv = (float)(reinterpret_cast<const T*>(x) + poff_x)[co];
      (reinterpret_cast<T*>(out) + poff_o)[co] = (T)(xv + bout[co] + acc[q]);
    }
  }
}

static inline unsigned grid_for(long work_items, int block = 256, long cap = 256L * 32) {
  long g = (work_items + block - 1) / block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (unsigned)g;
}

}  // namespace glsdet

using namespace glsdet;

extern "C" int glsdet_focus_pack(const float* img, int32_t n, int32_t cin, int32_t H, int32_t W,
                                 const glsdet_view* y, void* stream) {
  if (!img || !y) GLS_FAIL(GLSDET_E_ARG, "focus_pack: null argument");
  int rc;
  if ((rc = check_view(*y, "focus_pack.y"))) return rc;
  if (H % 2 || W % 2 || n < 1 || cin < 1) GLS_FAIL(GLSDET_E_ARG, "focus_pack: H,W must be even");
  if (y->n != n || y->h != H / 2 || y->w != W / 2 || y->c < 4 * cin)
    GLS_FAIL(GLSDET_E_ARG, "focus_pack: output extent mismatch");
  const glsdet_view v = *y;
  OpRecord op;
  op.kind = 1;
  op.flops = 0;
  op.bytes = (double)n * cin * H * W * 4 + (double)n * (H / 2) * (W / 2) * v.c * dtype_size(v.dtype);
  op.name = "focus_pack";
  op.launch = [=](hipStream_t st) -> int {
    const unsigned g = grid_for((long)n * (H / 2) * (W / 2));
    if (v.dtype == GLSDET_F16)
      hipLaunchKernelGGL(focus_pack_kernel<f16>, dim3(g), dim3(256), 0, st, img, n, cin, H, W, (unsigned char*)v.base, v.sn, v.sh, v.sw, v.c);
    else
      hipLaunchKernelGGL(focus_pack_kernel<float>, dim3(g), dim3(256), 0, st, img, n, cin, H, W, (unsigned char*)v.base, v.sn, v.sh, v.sw, v.c);
    GLS_HIP(hipGetLastError());
    return 0;
  };
  return submit(std::move(op), stream);
}

extern "C" int glsdet_maxpool2d(const glsdet_view* x, const glsdet_view* y, int32_t k, void* stream) {
  if (!x || !y) GLS_FAIL(GLSDET_E_ARG, "maxpool2d: null argument");
  int rc;
  if ((rc = check_view(*x, "maxpool2d.x"))) return rc;
  if ((rc = check_view(*y, "maxpool2d.y"))) return rc;
  if (!same_extent(*x, *y) || x->dtype != y->dtype) GLS_FAIL(GLSDET_E_ARG, "maxpool2d: x/y extent or dtype mismatch");
  if (k < 1 || !(k & 1) || k > 31) GLS_FAIL(GLSDET_E_ARG, "maxpool2d: k must be odd in [1,31]");
  if (x->c % 8) GLS_FAIL(GLSDET_E_ARG, "maxpool2d: channels must be a multiple of 8");
  const glsdet_view a = *x, b = *y;
  OpRecord op;
  op.kind = 2;
  op.flops = 0;
  op.bytes = 2.0 * a.n * a.h * a.w * a.c * dtype_size(a.dtype);
  op.name = "maxpool";
  op.launch = [=](hipStream_t st) -> int {
    const int vn = 16 / dtype_size(a.dtype);
    const unsigned g = grid_for((long)a.n * a.h * a.w * (a.c / vn));
    if (a.dtype == GLSDET_F16)
      hipLaunchKernelGGL(maxpool_kernel<f16>, dim3(g), dim3(256), 0, st, (const unsigned char*)a.base, a.sn, a.sh, a.sw, (unsigned char*)b.base, b.sn, b.sh, b.sw, a.n, a.h, a.w, a.c, k);
    else
      hipLaunchKernelGGL(maxpool_kernel<float>, dim3(g), dim3(256), 0, st, (const unsigned char*)a.base, a.sn, a.sh, a.sw, (unsigned char*)b.base, b.sn, b.sh, b.sw, a.n, a.h, a.w, a.c, k);
    GLS_HIP(hipGetLastError());
    return 0;
  };
  return submit(std::move(op), stream);
}

extern "C" int glsdet_spp_pools(const glsdet_view* x, const glsdet_view* y5, const glsdet_view* y9, const glsdet_view* y13, void* stream) {
  if (!x || !y5 || !y9 || !y13) GLS_FAIL(GLSDET_E_ARG, "spp_pools: null argument");
  int rc;
  if ((rc = check_view(*x, "spp_pools.x"))) return rc;
  const glsdet_view* ys[3] = {y5, y9, y13};
  for (int i = 0; i < 3; ++i) {
    if ((rc = check_view(*ys[i], "spp_pools.y"))) return rc;
    if (!same_extent(*x, *ys[i]) || ys[i]->dtype != x->dtype || ys[i]->sn != y5->sn || ys[i]->sh != y5->sh || ys[i]->sw != y5->sw)
      GLS_FAIL(GLSDET_E_ARG, "spp_pools: outputs must have x's extent and dtype and share one set of strides");
  }
  if (x->c % 8) GLS_FAIL(GLSDET_E_ARG, "spp_pools: channels must be a multiple of 8");
  const glsdet_view a = *x, b5 = *y5, b9 = *y9, b13 = *y13;
  OpRecord op;
  op.kind = 2;
  op.flops = 0;
  op.bytes = 4.0 * a.n * a.h * a.w * a.c * dtype_size(a.dtype);
  op.name = "spp_pools(5,9,13)";
  op.launch = [=](hipStream_t st) -> int {
    const int vn = 16 / dtype_size(a.dtype);
    const int tiles_x = (a.w + 15) / 16, tiles_y = (a.h + 7) / 8, cch = a.c / vn;
    const long g = (long)a.n * tiles_y * tiles_x * cch;
    if (g <= 0 || g > 0x7fffffffL) GLS_FAIL(GLSDET_E_ARG, "spp_pools: grid out of range");
    if (a.dtype == GLSDET_F16)
      hipLaunchKernelGGL(spp_pools_kernel<f16>, dim3((unsigned)g), dim3(256), 0, st, (const unsigned char*)a.base, a.sn, a.sh, a.sw, (unsigned char*)b5.base, (unsigned char*)b9.base, (unsigned char*)b13.base, b5.sn, b5.sh, b5.sw, a.h, a.w, cch, tiles_x, tiles_y);
    else
      hipLaunchKernelGGL(spp_pools_kernel<float>, dim3((unsigned)g), dim3(256), 0, st, (const unsigned char*)a.base, a.sn, a.sh, a.sw, (unsigned char*)b5.base, (unsigned char*)b9.base, (unsigned char*)b13.base, b5.sn, b5.sh, b5.sw, a.h, a.w, cch, tiles_x, tiles_y);
    GLS_HIP(hipGetLastError());
    return 0;
  };
  return submit(std::move(op), stream);
}

extern "C" int glsdet_channel_maxmean(const glsdet_view* x, const glsdet_view* y, void* stream) {
  if (!x || !y) GLS_FAIL(GLSDET_E_ARG, "channel_maxmean: null argument");
  int rc;
  if ((rc = check_view(*x, "channel_maxmean.x"))) return rc;
  if ((rc = check_view(*y, "channel_maxmean.y"))) return rc;
  if (x->dtype != y->dtype || y->n != x->n || y->h != x->h || y->w != x->w || y->c != 8 || x->c % 8)
    GLS_FAIL(GLSDET_E_ARG, "channel_maxmean: y must be [n,h,w,8] of x's dtype, x.c a multiple of 8");
  const glsdet_view a = *x, b = *y;
  OpRecord op;
  op.kind = 2;
  op.flops = 0;
  op.bytes = (double)a.n * a.h * a.w * (a.c + 8) * dtype_size(a.dtype);
  op.name = "channel_maxmean";
  op.launch = [=](hipStream_t st) -> int {
    const unsigned g = grid_for((long)a.n * a.h * a.w * 64);
    if (a.dtype == GLSDET_F16)
      hipLaunchKernelGGL(channel_maxmean_kernel<f16>, dim3(g), dim3(256), 0, st, (const unsigned char*)a.base, a.sn, a.sh, a.sw, (unsigned char*)b.base, b.sn, b.sh, b.sw, a.n, a.h, a.w, a.c);
    else
      hipLaunchKernelGGL(channel_maxmean_kernel<float>, dim3(g), dim3(256), 0, st, (const unsigned char*)a.base, a.sn, a.sh, a.sw, (unsigned char*)b.base, b.sn, b.sh, b.sw, a.n, a.h, a.w, a.c);
    GLS_HIP(hipGetLastError());
    return 0;
  };
  return submit(std::move(op), stream);
}

extern "C" int glsdet_resample_copy(const glsdet_view* x, const glsdet_view* y, int32_t factor, void* stream) {
  if (!x || !y) GLS_FAIL(GLSDET_E_ARG, "resample_copy: null argument");
  int rc;
  if ((rc = check_view(*x, "resample_copy.x"))) return rc;
  if ((rc = check_view(*y, "resample_copy.y"))) return rc;
  if (factor < 1 || factor > 8) GLS_FAIL(GLSDET_E_ARG, "resample_copy: bad factor %d", factor);
  if (x->dtype != y->dtype || y->n != x->n || y->c != x->c || y->h != x->h * factor || y->w != x->w * factor)
    GLS_FAIL(GLSDET_E_ARG, "resample_copy: extent mismatch");
  if (x->c % 8) GLS_FAIL(GLSDET_E_ARG, "resample_copy: channels must be a multiple of 8");
  const glsdet_view a = *x, b = *y;
  OpRecord op;
  op.kind = 3;
  op.flops = 0;
  op.bytes = ((double)a.n * a.h * a.w + (double)b.n * b.h * b.w) * a.c * dtype_size(a.dtype);
  op.name = factor == 1 ? "copy" : "upsample_nearest";
  op.launch = [=](hipStream_t st) -> int {
    const int vn = 16 / dtype_size(a.dtype);
    const unsigned g = grid_for((long)b.n * b.h * b.w * (b.c / vn));
    if (a.dtype == GLSDET_F16)
      hipLaunchKernelGGL(resample_kernel<f16>, dim3(g), dim3(256), 0, st, (const unsigned char*)a.base, a.sn, a.sh, a.sw, (unsigned char*)b.base, b.sn, b.sh, b.sw, b.n, b.h, b.w, b.c, factor);
    else
      hipLaunchKernelGGL(resample_kernel<float>, dim3(g), dim3(256), 0, st, (const unsigned char*)a.base, a.sn, a.sh, a.sw, (unsigned char*)b.base, b.sn, b.sh, b.sw, b.n, b.h, b.w, b.c, factor);
    GLS_HIP(hipGetLastError());
    return 0;
  };
  return submit(std::move(op), stream);
}

extern "C" int glsdet_copy_many(const glsdet_view* x, const glsdet_view* y, int32_t count, void* stream) {
  if (!x || !y) GLS_FAIL(GLSDET_E_ARG, "copy_many: null argument");
  if (count < 1 || count > 4096) GLS_FAIL(GLSDET_E_ARG, "copy_many: 1..4096 pairs, got %d", count);
  int rc;
  std::vector<CopyJob> jobs((size_t)count);
  double bytes = 0;
  for (int i = 0; i < count; ++i) {
    if ((rc = check_view(x[i], "copy_many.x"))) return rc;
    if ((rc = check_view(y[i], "copy_many.y"))) return rc;
    if (x[i].dtype != x[0].dtype || y[i].dtype != x[0].dtype || x[i].c != x[0].c || y[i].c != x[0].c)
      GLS_FAIL(GLSDET_E_ARG, "copy_many: pair %d: one dtype and one channel count per call", i);
    if (x[i].n != y[i].n || x[i].h != y[i].h || x[i].w != y[i].w) GLS_FAIL(GLSDET_E_ARG, "copy_many: pair %d: extent mismatch", i);
    CopyJob& j = jobs[(size_t)i];
    j.x = (const unsigned char*)x[i].base;
    j.y = (unsigned char*)y[i].base;
    j.xsn = x[i].sn, j.xsh = x[i].sh, j.xsw = x[i].sw, j.ysn = y[i].sn, j.ysh = y[i].sh, j.ysw = y[i].sw;
    j.n = x[i].n, j.h = x[i].h, j.w = x[i].w;
    const long px = (long)j.n * j.h * j.w;
    bytes += 2.0 * px * x[i].c * dtype_size(x[i].dtype);
  }
  if (x[0].c % 8) GLS_FAIL(GLSDET_E_ARG, "copy_many: channels must be a multiple of 8");
  const int C = x[0].c, dt = x[0].dtype;
  OpRecord op;
  op.kind = 3;
  op.flops = 0;
  op.bytes = bytes;
  op.name = "copy_many[" + std::to_string(count) + "]";
  op.launch = [=](hipStream_t st) -> int {
    const int vn = 16 / dtype_size(dt);
    // the pairs share the chip: about four workgroups per CU over one launch, at least one per pair
    for (int j0 = 0; j0 < count; j0 += GLS_COPY_JOBS) {
      const int nj = std::min(GLS_COPY_JOBS, count - j0);
      CopyManyArgs a = {};
      long big = 0;
      for (int i = 0; i < nj; ++i) {
        a.j[i] = jobs[(size_t)(j0 + i)];
        big = std::max(big, (long)a.j[i].n * a.j[i].h * a.j[i].w * (C / vn));
      }
      a.C = C;
      const unsigned per = (unsigned)std::max(1L, std::min((big + 255) / 256, (long)std::max(1, 2048 / nj)));
      if (dt == GLSDET_F16)
        hipLaunchKernelGGL(copy_many_kernel<f16>, dim3(per, nj), dim3(256), 0, st, a);
      else
        hipLaunchKernelGGL(copy_many_kernel<float>, dim3(per, nj), dim3(256), 0, st, a);
      GLS_HIP(hipGetLastError());
    }
    return 0;
  };
  return submit(std::move(op), stream);
}

extern "C" int glsdet_transpose_many(const glsdet_view* x, const glsdet_view* y, int32_t count, void* stream) {
  if (!x || !y) GLS_FAIL(GLSDET_E_ARG, "transpose_many: null argument");
  if (count < 1 || count > 4096) GLS_FAIL(GLSDET_E_ARG, "transpose_many: 1..4096 pairs, got %d", count);
  int rc;
  std::vector<CopyJob> jobs((size_t)count);
  double bytes = 0;
  for (int i = 0; i < count; ++i) {
    if ((rc = check_view(x[i], "transpose_many.x"))) return rc;
    if ((rc = check_view(y[i], "transpose_many.y"))) return rc;
    if (x[i].dtype != x[0].dtype || y[i].dtype != x[0].dtype || x[i].c != x[0].c)
      GLS_FAIL(GLSDET_E_ARG, "transpose_many: pair %d: one dtype and one channel count per call", i);
    const int vn = 16 / dtype_size(x[i].dtype);
    const long N = (long)x[i].h * x[i].w;
    if (x[i].n != 1 || y[i].n != 1 || y[i].h != 1) GLS_FAIL(GLSDET_E_ARG, "transpose_many: pair %d: one image -> one matrix [1,1,rows,cols]", i);
    if (y[i].w < x[i].c || y[i].c < (N + vn - 1) / vn * vn)
      GLS_FAIL(GLSDET_E_ARG, "transpose_many: pair %d: matrix %d x %d too small for %d channels x %ld pixels", i, y[i].w, y[i].c, x[i].c, N);
    CopyJob& j = jobs[(size_t)i];
    j.x = (const unsigned char*)x[i].base;
    j.y = (unsigned char*)y[i].base;
    j.xsn = x[i].sn, j.xsh = x[i].sh, j.xsw = x[i].sw, j.ysn = y[i].sn, j.ysh = y[i].sh, j.ysw = y[i].sw;
    j.n = 1, j.h = x[i].h, j.w = x[i].w;
    bytes += 2.0 * N * x[i].c * dtype_size(x[i].dtype);
  }
  if (x[0].c % 8) GLS_FAIL(GLSDET_E_ARG, "transpose_many: channels must be a multiple of 8");
  const int C = x[0].c, dt = x[0].dtype;
  OpRecord op;
  op.kind = 3;
  op.flops = 0;
  op.bytes = bytes;
  op.name = "transpose_many[" + std::to_string(count) + "]";
  op.launch = [=](hipStream_t st) -> int {
    for (int j0 = 0; j0 < count; j0 += GLS_COPY_JOBS) {
      const int nj = std::min(GLS_COPY_JOBS, count - j0);
      CopyManyArgs a = {};
      long big = 0;
      for (int i = 0; i < nj; ++i) {
        a.j[i] = jobs[(size_t)(j0 + i)];
        big = std::max(big, (long)a.j[i].h * a.j[i].w);
      }
      a.C = C;
      const dim3 grid((unsigned)((big + 63) / 64), (unsigned)((C + 63) / 64), (unsigned)nj);
      if (dt == GLSDET_F16)
        hipLaunchKernelGGL(transpose_many_kernel<f16>, grid, dim3(256), 0, st, a);
      else
        hipLaunchKernelGGL(transpose_many_kernel<float>, grid, dim3(256), 0, st, a);
      GLS_HIP(hipGetLastError());
    }
    return 0;
  };
  return submit(std::move(op), stream);
}

extern "C" int glsdet_nonlocal_multi(const glsdet_view* x, const glsdet_view* tpg, int32_t n_sets, int32_t ci,
                                     const float* const* wout, const float* const* bout, float* gram,
                                     const glsdet_view* out, void* stream) {
  if (!x || !tpg || !out || !wout || !bout || !gram) GLS_FAIL(GLSDET_E_ARG, "nonlocal: null argument");
  if (n_sets < 1 || n_sets > GLS_NL_SETS) GLS_FAIL(GLSDET_E_ARG, "nonlocal: 1..%d sets", GLS_NL_SETS);
  NlArgs a = {};
  a.n = n_sets;
  a.ci = ci;
  a.cx = x[0].c;
  a.nimg = x[0].n;
  a.kc = ci < GLS_NL_KC ? ci : GLS_NL_KC;
  OpRecord op;
  op.kind = 4;
  op.flops = op.bytes = 0;
  int maxN = 0;
  const int dt = x[0].dtype;
  const long per_set = (long)a.nimg * (8L * ci * ci + (long)a.cx * ci);      // floats of workspace per set
  for (int q = 0; q < n_sets; ++q) {
    int rc;
    if ((rc = check_view(x[q], "nonlocal.x", false))) return rc;
    if ((rc = check_view(tpg[q], "nonlocal.tpg", false))) return rc;
    if ((rc = check_view(out[q], "nonlocal.out", false))) return rc;
    if (!wout[q] || !bout[q]) GLS_FAIL(GLSDET_E_ARG, "nonlocal: null weight");
    if (!same_extent(x[q], out[q]) || x[q].dtype != dt || out[q].dtype != dt || tpg[q].dtype != dt)
      GLS_FAIL(GLSDET_E_ARG, "nonlocal: x/out extent or dtype mismatch");
    if (tpg[q].n != x[q].n || tpg[q].h != x[q].h || tpg[q].w != x[q].w || ci < 1 || tpg[q].c < 3 * ci)
      GLS_FAIL(GLSDET_E_ARG, "nonlocal: theta|phi|g view must be [n,h,w,>=3*ci]");
    if (x[q].c != a.cx || x[q].n != a.nimg) GLS_FAIL(GLSDET_E_ARG, "nonlocal: the sets must agree in n and channels");
    NlSet& S = a.s[q];
    const int N = x[q].h * x[q].w;
    S.x = (const unsigned char*)x[q].base; S.tpg = (const unsigned char*)tpg[q].base; S.out = (unsigned char*)out[q].base;
    S.xsn = x[q].sn; S.xsh = x[q].sh; S.xsw = x[q].sw;
    S.tsn = tpg[q].sn; S.tsh = tpg[q].sh; S.tsw = tpg[q].sw;
    S.osn = out[q].sn; S.osh = out[q].sh; S.osw = out[q].sw;
    S.wout = wout[q]; S.bout = bout[q];
    S.gram = gram + q * per_set;
    S.P = S.gram + (long)a.nimg * 8 * ci * ci;
    S.H = x[q].h; S.W = x[q].w;
    S.nsplit = N >= 1024 ? 8 : (N >= 256 ? 4 : 1);
    S.jchunk = (((N + S.nsplit - 1) / S.nsplit) + 63) / 64 * 64;
    S.invN = 1.0f / (float)N;
    if (N > maxN) maxN = N;
    // algorithmic count of the reference's two matmuls: 2 * (N*N*ci) MACs per image
    op.flops += 2.0 * 2.0 * (double)a.nimg * N * (double)N * ci;
    op.bytes += (double)a.nimg * N * (3.0 * ci + 2.0 * a.cx) * dtype_size(dt);
  }
  op.name = n_sets > 1 ? "nonlocal_multi(gram+fold+apply)" : "nonlocal(gram+fold+apply)";
  op.launch = [=](hipStream_t st) -> int {
    const int nb = (a.ci + 15) / 16;
    const dim3 g1(nb * nb, a.nimg, 8 * a.n), g2(a.nimg * a.n, (a.cx + GLS_FOLD_CO - 1) / GLS_FOLD_CO, (a.ci + 63) / 64),
        g3((maxN + 63) / 64, a.nimg * a.n, (a.cx + GLS_APPLY_CO - 1) / GLS_APPLY_CO);
    const size_t lds2 = (size_t)64 * (a.kc + 1) * 4;
    const size_t lds3 = ((size_t)64 * (a.kc + 1) + (size_t)GLS_APPLY_CO * a.kc) * 4;
    if (dt == GLSDET_F16) {
      hipLaunchKernelGGL(nl_gram_kernel<f16>, g1, dim3(256), 0, st, a);
      hipLaunchKernelGGL(nl_fold_kernel, g2, dim3(256), lds2, st, a);
      hipLaunchKernelGGL(nl_apply_kernel<f16>, g3, dim3(256), lds3, st, a);
    } else {
      hipLaunchKernelGGL(nl_gram_kernel<float>, g1, dim3(256), 0, st, a);
      hipLaunchKernelGGL(nl_fold_kernel, g2, dim3(256), lds2, st, a);
      hipLaunchKernelGGL(nl_apply_kernel<float>, g3, dim3(256), lds3, st, a);
    }
    GLS_HIP(hipGetLastError());
    return 0;
  };
  return submit(std::move(op), stream);
}

// Patch_Conv_NonLocal_adapt_new: the four quadrant blocks on windows whose extents live in DEVICE memory (`split`, written
// by glsdet_attn_split): x / out are the full maps, tpg[q] the full-map projections with quadrant q's weights.
extern "C" int glsdet_nonlocal_split(const glsdet_view* x, const glsdet_view* tpg, int32_t ci, const float* const* wout,
                                     const float* const* bout, float* gram, const glsdet_view* out, const int32_t* split,
                                     int32_t split_shift, void* stream) {
  if (!x || !tpg || !out || !wout || !bout || !gram || !split || split_shift < 0 || split_shift > 1)
    GLS_FAIL(GLSDET_E_ARG, "nonlocal_split: bad argument");
  int rc;
  if ((rc = check_view(*x, "nonlocal_split.x", false))) return rc;
  if ((rc = check_view(*out, "nonlocal_split.out", false))) return rc;
  if (!same_extent(*x, *out) || x->dtype != out->dtype) GLS_FAIL(GLSDET_E_ARG, "nonlocal_split: x/out mismatch");
  NlArgs a = {};
  a.n = 4; a.ci = ci; a.cx = x->c; a.nimg = x->n;
  a.kc = ci < GLS_NL_KC ? ci : GLS_NL_KC;
  a.split = split; a.FH = x->h; a.FW = x->w; a.shift = split_shift;
  const int dt = x->dtype;
  const long per_set = (long)a.nimg * (8L * ci * ci + (long)a.cx * ci);
  for (int q = 0; q < 4; ++q) {
    if ((rc = check_view(tpg[q], "nonlocal_split.tpg", false))) return rc;
    if (tpg[q].n != x->n || tpg[q].h != x->h || tpg[q].w != x->w || ci < 1 || tpg[q].c < 3 * ci || tpg[q].dtype != dt)
      GLS_FAIL(GLSDET_E_ARG, "nonlocal_split: theta|phi|g views must be [n,h,w,>=3*ci] of x's dtype");
    if (!wout[q] || !bout[q]) GLS_FAIL(GLSDET_E_ARG, "nonlocal_split: null weight");
    NlSet& S = a.s[q];
    S.x = (const unsigned char*)x->base; S.tpg = (const unsigned char*)tpg[q].base; S.out = (unsigned char*)out->base;
    S.xsn = x->sn; S.xsh = x->sh; S.xsw = x->sw;
    S.tsn = tpg[q].sn; S.tsh = tpg[q].sh; S.tsw = tpg[q].sw;
    S.osn = out->sn; S.osh = out->sh; S.osw = out->sw;
    S.wout = wout[q]; S.bout = bout[q];
    S.gram = gram + q * per_set;
    S.P = S.gram + (long)a.nimg * 8 * ci * ci;
    S.H = x->h; S.W = x->w; S.nsplit = 8; S.jchunk = 64; S.invN = 1.0f;        // overridden from `split` on the device
  }
  const int maxN = x->h * x->w;
  OpRecord op;
  op.kind = 4;
  op.flops = 2.0 * 2.0 * (double)a.nimg * maxN * (double)maxN / 4.0 * ci;      // four quadrants of ~N/4 positions each
  op.bytes = (double)a.nimg * maxN * (3.0 * ci + 2.0 * a.cx) * dtype_size(dt);
  op.name = "nonlocal_split(gram+fold+apply, device-side quadrants)";
  op.launch = [=](hipStream_t st) -> int {
    const int nb = (a.ci + 15) / 16;
    const dim3 g1(nb * nb, a.nimg, 8 * a.n), g2(a.nimg * a.n, (a.cx + GLS_FOLD_CO - 1) / GLS_FOLD_CO, (a.ci + 63) / 64),
        g3((maxN + 63) / 64, a.nimg * a.n, (a.cx + GLS_APPLY_CO - 1) / GLS_APPLY_CO);
    const size_t lds2 = (size_t)64 * (a.kc + 1) * 4;
    const size_t lds3 = ((size_t)64 * (a.kc + 1) + (size_t)GLS_APPLY_CO * a.kc) * 4;
    if (dt == GLSDET_F16) {
      hipLaunchKernelGGL(nl_gram_kernel<f16>, g1, dim3(256), 0, st, a);
      hipLaunchKernelGGL(nl_fold_kernel, g2, dim3(256), lds2, st, a);
      hipLaunchKernelGGL(nl_apply_kernel<f16>, g3, dim3(256), lds3, st, a);
    } else {
      hipLaunchKernelGGL(nl_gram_kernel<float>, g1, dim3(256), 0, st, a);
      hipLaunchKernelGGL(nl_fold_kernel, g2, dim3(256), lds2, st, a);
      hipLaunchKernelGGL(nl_apply_kernel<float>, g3, dim3(256), lds3, st, a);
    }
    GLS_HIP(hipGetLastError());
    return 0;
  };
  return submit(std::move(op), stream);
}

extern "C" int glsdet_nonlocal(const glsdet_view* x, const glsdet_view* tpg, int32_t ci, const float* wout,
                               const float* bout, float* gram, const glsdet_view* out, void* stream) {
  return glsdet_nonlocal_multi(x, tpg, 1, ci, &wout, &bout, gram, out, stream);
}
