// Focus + stem conv in ONE kernel:  y = act(bn(conv3x3(cat(TL, BL, TR, BR)(img))))
//   drone/models/base/darknet.py:10-21 (Focus.forward) -- the space-to-depth tensor is never materialised.
//
// The stem is the most bandwidth-bound layer of the network (12 -> 32 channels at half resolution: 2 x 144 MACs per
// output byte): as two launches it moved 103 MB (image) + 2 x 172 MB (packed tensor out and in) + 137 MB (output) per
// batch of 8 x 800 x 1344.  Here a workgroup reads the fp32 NCHW image patch of its output tile directly (float2 loads:
// the TL|TR resp. BL|BR pair of one packed pixel), builds the 16-channel packed patch in LDS, multiplies the nine taps
// on MFMA (K = 9 x 16, channels 12..15 zero) and writes NHWC: 103 + 137 MB, one launch.  A workgroup walks a strip of
// STRIP tiles along x with the weight tile resident in LDS.
// GEMM orientation, fragment layout and epilogue as in conv.hip; weights packed as for a 3x3 conv over 16 channels.
#include "conv_common.h"

namespace glsdet {

struct StemArgs {
  const float* img;            // [n][3][H][W] fp32
  const unsigned char* w;      // [cout_pad][kpad(3,3,16)]
  const float* scale;
  const float* bias;
  unsigned char* y;
  long y_sn, y_sh, y_sw;
  int n, H, W, Ho, Wo, cout, cout_pad, kpad, act;
  int tiles_x, tiles_y, strips_x;
};

template <typename T, int CO_T>
__global__ __launch_bounds__(256) void focus_stem_kernel(const StemArgs a) {
  constexpr int ES = (int)sizeof(T);
  constexpr int TH = 8, TW = 16, PH = TH + 2, PW = TW + 2, STRIP = 6;
  constexpr int PRS = 16 * ES + 16;                 // patch row: 16 channels + 16 B (odd number of 16-B units: conflict free)
  constexpr int KBY = 144 * ES;                     // bytes of one weight row
  constexpr int WRS = KBY + 16;
  constexpr int TM = CO_T / 32;
  constexpr int ORS = CO_T * ES + 16;
  constexpr int W_BYTES = CO_T * WRS, P_BYTES = PH * PW * PRS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sW = smem;
  unsigned char* sP = smem + W_BYTES;               // patch, later the staged output tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;

  int t = blockIdx.x;
  const int sx = t % a.strips_x;
  t /= a.strips_x;
  const int ty = t % a.tiles_y, img = t / a.tiles_y;
  const int ty0 = ty * TH;

  // resident weight tile
  for (int q = tid; q < CO_T * (KBY / 16); q += 256) {
    const int row = q / (KBY / 16), c = q - row * (KBY / 16);
    u32x4 v = {0u, 0u, 0u, 0u};
    if (row < a.cout_pad) v = *reinterpret_cast<const u32x4*>(a.w + ((long)row * a.kpad) * ES + c * 16);
    *reinterpret_cast<u32x4*>(sW + row * WRS + c * 16) = v;
  }
  f32x4 scv[TM][4], biv[TM][4];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int co = i * 32 + 8 * g + 4 * lh;
      scv[i][g] = f32x4{0.f, 0.f, 0.f, 0.f};
      biv[i][g] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (co < a.cout_pad) {
        scv[i][g] = *reinterpret_cast<const f32x4*>(a.scale + co);
        biv[i][g] = *reinterpret_cast<const f32x4*>(a.bias + co);
      }
    }
  const long plane = (long)a.H * a.W;
  const float* ibase = a.img + (long)img * 3 * plane;

  // image loads of one tile: NL float2 per thread (channel, image row, packed column), held in registers so that the
  // NEXT tile's loads are in flight while the current tile is multiplied and stored
  constexpr int NQ = 3 * (2 * PH) * PW, NL = (NQ + 255) / 256;
  float2 pre[NL];
  auto issue_loads = [&](int tx0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int q = tid + i * 256;
      const int px = q % PW, r = q / PW;
      const int iy_l = r % (2 * PH), c = r / (2 * PH);
      const int Y = ty0 - 1 + (iy_l >> 1), X = tx0 - 1 + px;   // packed coordinates; outside the packed image: zero padding
      pre[i] = float2{0.f, 0.f};
      if (q < NQ && (unsigned)Y < (unsigned)a.Ho && (unsigned)X < (unsigned)a.Wo)
        pre[i] = *reinterpret_cast<const float2*>(ibase + c * plane + (long)(2 * Y + (iy_l & 1)) * a.W + 2 * X);
    }
  };
  issue_loads(sx * STRIP * TW);
  for (int s = 0; s < STRIP; ++s) {
    const int tx = sx * STRIP + s;
    if (tx >= a.tiles_x) break;                     // uniform
    const int tx0 = tx * TW;
    __syncthreads();                                // the previous tile's store phase is done with sP
    // ---- packed patch: one float2 = the dx = 0 | 1 pair of one packed pixel
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int q = tid + i * 256;
      if (q < NQ) {
        const int px = q % PW, r = q / PW;
        const int iy_l = r % (2 * PH), c = r / (2 * PH);
        const int py = iy_l >> 1, dy = iy_l & 1;
        T* dst = reinterpret_cast<T*>(sP + (py * PW + px) * PRS);
        dst[dy * 3 + c] = (T)pre[i].x;              // TL (dy 0) / BL (dy 1): channels 0..2 / 3..5
        dst[6 + dy * 3 + c] = (T)pre[i].y;          // TR / BR: channels 6..8 / 9..11
      }
    }
    for (int q = tid; q < PH * PW; q += 256) {      // channels 12..15
      T* dst = reinterpret_cast<T*>(sP + q * PRS) + 12;
      dst[0] = dst[1] = dst[2] = dst[3] = (T)0.f;
    }
    if (s + 1 < STRIP && tx + 1 < a.tiles_x) issue_loads(tx0 + TW);
    __syncthreads();
    // ---- nine taps: wave w owns pixel block w (32 pixels = two tile rows), all cout rows
    f32x16 acc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.0f;
    const int pix = wave * 32 + l31;
    const int oy = pix >> 4, ox = pix & 15;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int r = tap / 3, c = tap - 3 * r;
      const unsigned char* brow = sP + ((oy + r) * PW + ox + c) * PRS;
#pragma unroll
      for (int kk = 0; kk < 16 * ES / 32; ++kk) {   // f16: one 32-byte step per tap, f32: two
        const u32x4 bf = *reinterpret_cast<const u32x4*>(brow + kk * 32 + lh * 16);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const u32x4 af = *reinterpret_cast<const u32x4*>(sW + (i * 32 + l31) * WRS + tap * 16 * ES + kk * 32 + lh * 16);
          MMA<T>::run(af, bf, acc[i]);
        }
      }
    }
    __syncthreads();                                // all waves are done reading the patch: it becomes the output stage
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int co_l = i * 32 + 8 * g + 4 * lh;
        const f32x4 xv = {acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]};
        const f32x4 yv = scale_bias_act4<T>(xv, scv[i][g], biv[i][g], a.act);
        const float v[4] = {yv[0], yv[1], yv[2], yv[3]};
        store4(sP + pix * ORS + co_l * ES, v, (T*)nullptr);
      }
    __syncthreads();
    constexpr int VO = 16 / ES, OCPR = CO_T / VO;
    for (int q = tid; q < 128 * OCPR; q += 256) {
      const int px_l = q / OCPR, cq = q - px_l * OCPR;
      const int ho = ty0 + (px_l >> 4), wo = tx0 + (px_l & 15), co = cq * VO;
      if (ho < a.Ho && wo < a.Wo && co < a.cout) {
        const long yo = (long)img * a.y_sn + (long)ho * a.y_sh + (long)wo * a.y_sw + co;
        *reinterpret_cast<u32x4*>(a.y + yo * (long)ES) = *reinterpret_cast<const u32x4*>(sP + px_l * ORS + cq * 16);
      }
    }
  }
}

template <typename T, int CO_T>
static int launch_stem(const StemArgs& a, hipStream_t st) {
  constexpr int ES = (int)sizeof(T);
  constexpr int patch = 10 * 18 * (16 * ES + 16), epi = 128 * (CO_T * ES + 16);
  constexpr int lds = CO_T * (144 * ES + 16) + (patch > epi ? patch : epi);
  if (lds > 64 * 1024) {
    static bool attr_set = false;
    if (!attr_set) {
      GLS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(focus_stem_kernel<T, CO_T>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      attr_set = true;
    }
  }
  const long grid = (long)a.n * a.tiles_y * a.strips_x;
  hipLaunchKernelGGL((focus_stem_kernel<T, CO_T>), dim3((unsigned)grid), dim3(256), lds, st, a);
  GLS_HIP(hipGetLastError());
  return 0;
}

}  // namespace glsdet

using namespace glsdet;

extern "C" int glsdet_focus_conv(const float* img, int32_t n, int32_t cin, int32_t H, int32_t W, const void* w, const float* scale,
                                 const float* bias, int32_t act, const glsdet_view* y, void* stream) {
  if (!img || !w || !scale || !bias || !y) GLS_FAIL(GLSDET_E_ARG, "focus_conv: null argument");
  if (cin != 3 || n < 1 || H < 2 || W < 2 || (H & 1) || (W & 1)) GLS_FAIL(GLSDET_E_ARG, "focus_conv: needs a 3-channel image with even H and W");
  if (((uintptr_t)img & 7) || ((uintptr_t)w | (uintptr_t)scale | (uintptr_t)bias) & 15) GLS_FAIL(GLSDET_E_ALIGN, "focus_conv: operand alignment");
  int rc;
  if ((rc = check_view(*y, "focus_conv.y"))) return rc;
  if (y->n != n || y->h != H / 2 || y->w != W / 2 || y->c % 8 || y->c > 64) GLS_FAIL(GLSDET_E_ARG, "focus_conv: output must be [n, H/2, W/2, <= 64 channels]");
  if (act < 0 || act > 5) GLS_FAIL(GLSDET_E_ARG, "focus_conv: bad act %d", act);
  StemArgs a;
  a.img = img; a.w = (const unsigned char*)w; a.scale = scale; a.bias = bias;
  a.y = (unsigned char*)y->base; a.y_sn = y->sn; a.y_sh = y->sh; a.y_sw = y->sw;
  a.n = n; a.H = H; a.W = W; a.Ho = H / 2; a.Wo = W / 2;
  a.cout = y->c; a.cout_pad = glsdet_conv_cout_pad(y->c); a.kpad = glsdet_conv_kpad(3, 3, 16, y->dtype); a.act = act;
  a.tiles_x = (a.Wo + 15) / 16; a.tiles_y = (a.Ho + 7) / 8; a.strips_x = (a.tiles_x + 5) / 6;
  const int dt = y->dtype, big = a.cout_pad > 32;
  OpRecord op;
  op.kind = 0;
  op.flops = 2.0 * (double)n * a.Ho * a.Wo * y->c * 108.0;
  op.bytes = (double)n * 3 * H * W * 4.0 + (double)n * a.Ho * a.Wo * y->c * dtype_size(dt);
  char nm[96];
  snprintf(nm, sizeof nm, "focus_stem<%s,%dx8x16> 3x3 s1 cin12 cout%d (fp32 NCHW image in)", dt ? "f32" : "f16", big ? 64 : 32, y->c);
  op.name = nm;
  op.launch = [a, dt, big](hipStream_t st) -> int {
    if (dt == GLSDET_F16) return big ? launch_stem<f16, 64>(a, st) : launch_stem<f16, 32>(a, st);
    return big ? launch_stem<float, 64>(a, st) : launch_stem<float, 32>(a, st);
  };
  return submit(std::move(op), stream);
}
