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
#include <stdlib.h>

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

  // image loads of one tile: thread q < PH * PW owns packed pixel q and loads its six float2 (channel x image row: the
  // TL|TR resp. BL|BR pairs), held in registers so that the NEXT tile's loads are in flight while the current tile is
  // multiplied and stored; the 16 packed channels then go to LDS as whole 16-byte chunks
  constexpr int NPP = PH * PW;
  const int ppy = tid / PW, ppx = tid - ppy * PW;
  float2 pre[6];
  auto issue_loads = [&](int tx0) __attribute__((always_inline)) {
    const int Y = ty0 - 1 + ppy, X = tx0 - 1 + ppx;          // packed coordinates; outside the packed image: zero padding
    const bool in = tid < NPP && (unsigned)Y < (unsigned)a.Ho && (unsigned)X < (unsigned)a.Wo;
    const float* p0 = ibase + (long)(2 * Y) * a.W + 2 * X;
#pragma unroll
    for (int i = 0; i < 6; ++i) {                            // i = dy * 3 + c
      pre[i] = float2{0.f, 0.f};
      if (in) pre[i] = *reinterpret_cast<const float2*>(p0 + (i % 3) * plane + (i / 3) * a.W);
    }
  };
  issue_loads(sx * STRIP * TW);
  for (int s = 0; s < STRIP; ++s) {
    const int tx = sx * STRIP + s;
    if (tx >= a.tiles_x) break;                     // uniform
    const int tx0 = tx * TW;
    __syncthreads();                                // the previous tile's store phase is done with sP
    // ---- packed patch: channels [TL c0-2 | BL c0-2 | TR c0-2 | BR c0-2 | 0 0 0 0] of pixel `tid`
    if (tid < NPP) {
      T v[16];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        v[i] = (T)pre[i].x;                           // TL (dy 0) / BL (dy 1): channels 0..2 / 3..5
        v[6 + i] = (T)pre[i].y;                       // TR / BR: channels 6..8 / 9..11
      }
      v[12] = v[13] = v[14] = v[15] = (T)0.f;
      unsigned char* dst = sP + tid * PRS;
#pragma unroll
      for (int k = 0; k < 16 * ES / 16; ++k) *reinterpret_cast<u32x4*>(dst + k * 16) = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(v) + k * 16);
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

// ------------------------------------------------------------------------------------------------
// Focus + stem conv + the FIRST downsampling conv (dark2.0: 3x3 stride 2, darknet.py:117-121) in one kernel.
// The stem output is the largest tensor of the network (8 x 400 x 672 x 32 fp16 = 138 MB), written once and read once by
// a layer that does 18 MACs per byte of it: as two launches (focus_stem 74 us + 3x3 s2 81 us) it costs 2 x 138 MB of the
// 447 MB both move.  Here the workgroup of an 8 x 16 OUTPUT tile of dark2.0
//   phase A  reads the fp32 image region of the tile's 17 x 33 stem-output halo (38 x 70 pixels, float2 loads), builds
//            the packed 19 x 35 x 16 patch, multiplies the nine stem taps for all 561 halo pixels (18 blocks of 32, five
//            per wave: the accumulators live in registers until ALL are done, because the result patch overwrites the
//            packed one), applies scale / bias / act, ZERO outside the stem's map, rounds to the storage type and writes
//            the patch with its columns DE-INTERLEAVED (conv_halo.hip, STR = 2);
//   phase B  runs the stride-2 ring loop over that resident patch (its weight DMAs were issued at kernel start);
//   epilogue as the halo kernels.
// Rounding points and k order are those of the two stand-alone kernels: bit-identical result (tests/test_hip_model.py).
struct Stem2Args {
  ConvArgs c;                  // the stride-2 conv: w, scale, bias, y, strides, H / W = the stem's map, Ho / Wo, Cout, act
  const float* img;            // [n][3][IH][IW] fp32
  const unsigned char* w1;     // stem weights [32][kpad(3,3,16)]
  const float* scale1;
  const float* bias1;
  int IH, IW, kpad1, act1;
};

template <typename T>
struct Stem2Geom {
  static constexpr int ES = (int)sizeof(T), C1 = 32, CO_T = 64;
  static constexpr int KB = C1 * ES;                           // one channel chunk: 64 B (f16) / 128 B (f32)
  static constexpr int RING = KB == 64 ? 4 : 3;
  static constexpr int SH = 17, SW = 33, NS = SH * SW;         // stem-output halo of an 8 x 16 stride-2 tile
  static constexpr int FH = SH + 2, FW = SW + 2, NF = FH * FW; // packed patch
  static constexpr int FRS = 16 * ES + 16, WRS = 144 * ES + 16, RS1 = KB + 16;
  static constexpr int RING_BYTES = RING * CO_T * KB, W1_BYTES = C1 * WRS;
  static constexpr int W1_OFF = RING_BYTES, B_OFF = RING_BYTES + W1_BYTES;
  static constexpr int F_BYTES = (NF + 40) * FRS, P1_BYTES = NS * RS1;        // (+40: the junk pixels 561..575 read past the patch)
  static constexpr int STAGE = B_OFF + (F_BYTES > P1_BYTES ? F_BYTES : P1_BYTES);
};

template <typename T>
__global__ __launch_bounds__(256) void focus_stem_down_kernel(const Stem2Args b, const int tiles_x, const int tiles_y) {
  using G = Stem2Geom<T>;
  const ConvArgs& a = b.c;
  constexpr int ES = G::ES, VEC = 16 / ES, KB = G::KB, RING = G::RING, CO_T = G::CO_T;
  constexpr int SW = G::SW, NS = G::NS, FH = G::FH, FW = G::FW, NF = G::NF, FRS = G::FRS, WRS = G::WRS, RS1 = G::RS1;
  constexpr int HALF = (SW + 1) / 2, PITCH = 2 * SW;
  constexpr int CPRW = KB / 16, RPL = 256 / KB, RPI = 64 / CPRW, NI = CO_T / RPI / 4;
  constexpr int A_BYTES = CO_T * KB;
  constexpr int TM = 1, TN = 2;                        // waves: 2 (cout) x 2 (pixels)
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  unsigned char* sW = smem + G::W1_OFF;
  unsigned char* sF = smem + G::B_OFF;                 // packed patch, then the stem-output patch P1
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int t = blockIdx.x;
  const int tx = t % tiles_x;
  t /= tiles_x;
  const int ty = t % tiles_y, img = t / tiles_y;
  const int ty0 = ty * 8, tx0 = tx * 16;               // output tile origin
  const int Y0 = 2 * ty0 - 2, X0 = 2 * tx0 - 2;        // packed (= stem map) coordinates of the packed patch's origin

  // ---- weight ring of the second conv: the first RING - 1 taps are requested now and land during phase A
  const auto wrs = gls_make_rsrc(a.w, a.w_bytes);
  unsigned wd[NI];
#pragma unroll
  for (int q = 0; q < NI; ++q) {
    const int row = RPI * (wave + 4 * q) + lane / CPRW;
    const int ch = (lane % CPRW) ^ ((row / RPL) & (CPRW - 1));
    wd[q] = row < a.cout_pad ? (unsigned)((row * a.kpad + ch * VEC) * ES) : GLS_OOB;
  }
  typedef __attribute__((address_space(3))) void* lds_ptr;
  int dg = 0, dslot = 0;
  unsigned dadd = 0;
  auto dma_next = [&]() __attribute__((always_inline)) {
    unsigned char* dst = smem + dslot * A_BYTES + wave * 1024;
#pragma unroll
    for (int q = 0; q < NI; ++q)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_ptr)(dst + q * 4096), 16, (int)(wd[q] + dadd), 0, 0, 0);
    dslot = dslot + 1 == RING ? 0 : dslot + 1;
    dadd += (unsigned)KB;
    if (++dg >= 9) dadd = GLS_OOB;
  };
#pragma unroll
  for (int g = 0; g < RING - 1; ++g) dma_next();

  // ---- phase A: image -> packed patch -> stem conv on the 17 x 33 halo
  {
    const long plane = (long)b.IH * b.IW;
    const float* ibase = b.img + (long)img * 3 * plane;
    constexpr int NQ = 3 * (2 * FH) * FW, NL = (NQ + 255) / 256;
    float2 pre[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int q = tid + i * 256;
      const int px = q % FW, r = q / FW;
      const int iy_l = r % (2 * FH), c = r / (2 * FH);
      const int Y = Y0 + (iy_l >> 1), X = X0 + px;
      pre[i] = float2{0.f, 0.f};
      if (q < NQ && (unsigned)Y < (unsigned)a.H && (unsigned)X < (unsigned)a.W && !(a.dbg & 1))
        pre[i] = *reinterpret_cast<const float2*>(ibase + c * plane + (long)(2 * Y + (iy_l & 1)) * b.IW + 2 * X);
    }
    for (int q = tid; q < G::C1 * (144 * ES / 16); q += 256) {
      const int row = q / (144 * ES / 16), c = q - row * (144 * ES / 16);
      *reinterpret_cast<u32x4*>(sW + row * WRS + c * 16) = *reinterpret_cast<const u32x4*>(b.w1 + ((long)row * b.kpad1) * ES + c * 16);
    }
    for (int q = tid; q < NF; q += 256) {              // channels 12..15
      T* dst = reinterpret_cast<T*>(sF + q * FRS) + 12;
      dst[0] = dst[1] = dst[2] = dst[3] = (T)0.f;
    }
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int q = tid + i * 256;
      if (q < NQ && !(a.dbg & 2)) {
        const int px = q % FW, r = q / FW;
        const int iy_l = r % (2 * FH), c = r / (2 * FH);
        const int py = iy_l >> 1, dy = iy_l & 1;
        T* dst = reinterpret_cast<T*>(sF + (py * FW + px) * FRS);
        dst[dy * 3 + c] = (T)pre[i].x;                // TL (dy 0) / BL (dy 1)
        dst[6 + dy * 3 + c] = (T)pre[i].y;            // TR / BR
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // (never __syncthreads(): it would drain the ring's DMAs)
    constexpr int NT = 5;                              // 18 pixel blocks of 32 over four waves
    f32x16 acc[NT];
    int boff[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.0f;
      const int p = (wave + 4 * i) * 32 + l31;
      const int sy = p / SW, sx = p - sy * SW;
      boff[i] = (sy * FW + sx) * FRS + lh * 16;
    }
    if (!(a.dbg & 4))
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int r = tap / 3, c = tap - 3 * r;
#pragma unroll
      for (int kk = 0; kk < 16 * ES / 32; ++kk) {
        const u32x4 af = *reinterpret_cast<const u32x4*>(sW + l31 * WRS + tap * 16 * ES + kk * 32 + lh * 16);
#pragma unroll
        for (int i = 0; i < NT; ++i) {
          if (wave + 4 * i < 18) {
            const u32x4 bf = *reinterpret_cast<const u32x4*>(sF + boff[i] + (r * FW + c) * FRS + kk * 32);
            MMA<T>::run(af, bf, acc[i]);
          }
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // every wave is done with the packed patch: its bytes become P1
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int p = (wave + 4 * i) * 32 + l31;
      const int sy = p / SW, sx = p - sy * SW;
      const int Ys = 2 * ty0 - 1 + sy, Xs = 2 * tx0 - 1 + sx;
      const bool inside = (unsigned)Ys < (unsigned)a.H && (unsigned)Xs < (unsigned)a.W;
      if (wave + 4 * i < 18 && p < NS && !(a.dbg & 8)) {
        const int slot = sy * SW + ((sx & 1) ? HALF + (sx >> 1) : (sx >> 1));
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int co = 8 * g + 4 * lh;
          const f32x4 sc = *reinterpret_cast<const f32x4*>(b.scale1 + co), bi = *reinterpret_cast<const f32x4*>(b.bias1 + co);
          const f32x4 xv = {acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]};
          f32x4 yv = scale_bias_act4<T>(xv, sc, bi, b.act1);
          if (!inside) yv = f32x4{0.f, 0.f, 0.f, 0.f};
          const float v[4] = {yv[0], yv[1], yv[2], yv[3]};
          store4(sF + slot * RS1 + co * ES, v, (T*)nullptr);
        }
      }
    }
  }

  // ---- phase B: 3x3 stride 2 over the resident patch
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
  const int wco = wave % 2, wpx = wave / 2;
  const int a_row = (wco * 32 + l31) * KB;
  int a_sw[KB / 32];
#pragma unroll
  for (int kk = 0; kk < KB / 32; ++kk) a_sw[kk] = ((2 * kk + lh) ^ ((l31 / RPL) & (CPRW - 1))) << 4;
  int b_off[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int pix = wpx * 64 + j * 32 + l31;
    int oy, ox;
    pix_to_xy16<PITCH>(pix, oy, ox);
    b_off[j] = G::B_OFF + (oy * PITCH + ox) * RS1 + lh * 16;
  }
  int g = 0, tap_off = 0, ts = 0;
  for (int tap = 0; tap < 9; ++tap) {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NI * (RING - 2)) : "memory");   // at tap 0 also: P1 is visible
    dma_next();
    const unsigned char* sA = smem + g * A_BYTES + a_row;
    if (!(a.dbg & 16))
#pragma unroll
    for (int kk = 0; kk < KB / 32; ++kk) {
      const u32x4 af = *reinterpret_cast<const u32x4*>(sA + a_sw[kk]);
      u32x4 bf[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const u32x4*>(smem + b_off[j] + tap_off + kk * 32);
#pragma unroll
      for (int j = 0; j < TN; ++j) MMA<T>::run(af, bf[j], acc[0][j]);
    }
    g = g + 1 == RING ? 0 : g + 1;
    ++ts;
    tap_off += (ts == 1) ? HALF * RS1 : (ts == 2 ? -(HALF - 1) * RS1 : (SW - 1) * RS1);
    ts = (ts == 3) ? 0 : ts;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

#pragma unroll
  for (int gq = 0; gq < 4; ++gq) {
    const int co_l = wco * 32 + 8 * gq + 4 * lh;
    f32x4 sc = {0.f, 0.f, 0.f, 0.f}, bi = {0.f, 0.f, 0.f, 0.f};
    if (co_l < a.cout_pad) {
      sc = *reinterpret_cast<const f32x4*>(a.scale + co_l);
      bi = *reinterpret_cast<const f32x4*>(a.bias + co_l);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int px_l = wpx * 64 + j * 32 + l31;
      const f32x4 xv = {acc[0][j][4 * gq], acc[0][j][4 * gq + 1], acc[0][j][4 * gq + 2], acc[0][j][4 * gq + 3]};
      const f32x4 yv = scale_bias_act4<T>(xv, sc, bi, a.act);
      const float v[4] = {yv[0], yv[1], yv[2], yv[3]};
      stage4<T, CO_T>(smem, px_l, co_l, v, false);
    }
  }
  __syncthreads();
  if (!(a.dbg & 32)) halo_store_tile<T, CO_T, PITCH>(smem, a, img, ty0, tx0, 0, tid);
}

template <typename T>
static int launch_stem_down(const Stem2Args& b, hipStream_t st) {
  using G = Stem2Geom<T>;
  constexpr int epi = epi_bytes<T>(64, 128, false);
  constexpr int lds = G::STAGE > epi ? G::STAGE : epi;
  static bool attr_set = false;
  if (!attr_set && lds > 64 * 1024) {
    GLS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(focus_stem_down_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_set = true;
  }
  const int tiles_x = (b.c.Wo + 15) / 16, tiles_y = (b.c.Ho + 7) / 8;
  const long grid = (long)b.c.N * tiles_x * tiles_y;
  if (grid <= 0 || grid > 0x7fffffffL) GLS_FAIL(GLSDET_E_ARG, "focus_conv_down: grid %ld out of range", grid);
  hipLaunchKernelGGL((focus_stem_down_kernel<T>), dim3((unsigned)grid), dim3(256), lds, st, b, tiles_x, tiles_y);
  GLS_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------
// ResNet stem: y = relu(bn(conv7x7 stride 2 pad 3 (img)))  straight from the fp32 NCHW image
// (ufp/mmdet/models/backbones/resnet.py:634-636).  As two launches (NCHW -> NHWC8 pack, then the generic kernel with 3
// channels padded to 8: K = 392, 1.7 TB/s) it cost 60 + 246 us of the MPDet step and moved 137 MB of packed image twice.
// Here K is laid out [7 rows][8 taps][4 channels] (tap 7 and channel 3 are zero weights): a 16-element K step is HALF A
// FILTER ROW -- four taps x four channels -- which in a patch stored [row][column][4 channels] is 32 CONTIGUOUS bytes
// starting at an even column, so every B fragment is one aligned ds_read_b128 and the im2col costs nothing.  The patch row
// pitch is 48 pixels so that the two pixel rows of a 32-lane MFMA block hit the same bank sets (conflict free).  Weights
// (64 x 224) resident in LDS, a strip of tiles per workgroup with the next tile's image loads in flight (focus_stem recipe).
struct RStemArgs {
  const float* img;            // [n][3][H][W] fp32
  const unsigned char* w;      // [64][7][8][4] elements of y's dtype
  const float* scale;
  const float* bias;
  unsigned char* y;
  long y_sn, y_sh, y_sw;
  int n, H, W, Ho, Wo, cout, act;
  int tiles_x, tiles_y, strips_x;
};

template <typename T>
__global__ __launch_bounds__(256) void resnet_stem_kernel(const RStemArgs a) {
  constexpr int ES = (int)sizeof(T);
  constexpr int TH = 8, TW = 16, PH = 2 * TH + 5, PW = 2 * TW + 5, PWP = 48, STRIP = 6, CO_T = 64;
  constexpr int PXB = 4 * ES;                         // bytes of a patch pixel: 4 channels
  constexpr int KROW = 224 * ES, WRS = KROW + 16;     // weight row: 7 x 8 x 4 elements
  constexpr int RB = 16 * ES;                         // bytes of one K step (4 taps x 4 channels)
  constexpr int ORS = CO_T * ES + 16;
  constexpr int W_BYTES = CO_T * WRS, P_BYTES = PH * PWP * PXB, E_BYTES = 128 * ORS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sW = smem;
  unsigned char* sP = smem + W_BYTES;                 // patch, later the staged output tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  int t = blockIdx.x;
  const int sx = t % a.strips_x;
  t /= a.strips_x;
  const int ty = t % a.tiles_y, img = t / a.tiles_y;
  const int ty0 = ty * TH;

  for (int q = tid; q < CO_T * (KROW / 16); q += 256) {
    const int row = q / (KROW / 16), c = q - row * (KROW / 16);
    *reinterpret_cast<u32x4*>(sW + row * WRS + c * 16) = *reinterpret_cast<const u32x4*>(a.w + (long)row * KROW + c * 16);
  }
  f32x4 scv[2][4], biv[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int co = i * 32 + 8 * g + 4 * lh;
      scv[i][g] = *reinterpret_cast<const f32x4*>(a.scale + co);
      biv[i][g] = *reinterpret_cast<const f32x4*>(a.bias + co);
    }
  const long plane = (long)a.H * a.W;
  const float* ibase = a.img + (long)img * 3 * plane;
  constexpr int NQ = 3 * PH * PW, NL = (NQ + 255) / 256;
  float pre[NL];
  auto issue_loads = [&](int tx0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int q = tid + i * 256;
      const int px = q % PW, r = q / PW;
      const int py = r % PH, c = r / PH;
      const int Y = 2 * ty0 - 3 + py, X = 2 * tx0 - 3 + px;
      pre[i] = 0.f;
      if (q < NQ && (unsigned)Y < (unsigned)a.H && (unsigned)X < (unsigned)a.W) pre[i] = ibase[c * plane + (long)Y * a.W + X];
    }
  };
  issue_loads(sx * STRIP * TW);
  for (int s = 0; s < STRIP; ++s) {
    const int tx = sx * STRIP + s;
    if (tx >= a.tiles_x) break;                       // uniform
    const int tx0 = tx * TW;
    __syncthreads();                                  // the previous tile's store phase is done with sP
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int q = tid + i * 256;
      if (q < NQ) {
        const int px = q % PW, r = q / PW;
        const int py = r % PH, c = r / PH;
        reinterpret_cast<T*>(sP + (py * PWP + px) * PXB)[c] = (T)pre[i];
      }
    }
    for (int q = tid; q < PH * PWP; q += 256) {       // channel 3 of every pixel, and the pad columns (tap 7 reads column PW)
      T* dst = reinterpret_cast<T*>(sP + q * PXB);
      dst[3] = (T)0.f;
      if (q % PWP >= PW) dst[0] = dst[1] = dst[2] = (T)0.f;
    }
    if (s + 1 < STRIP && tx + 1 < a.tiles_x) issue_loads(tx0 + TW);
    __syncthreads();
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.0f;
    const int pix = wave * 32 + l31;
    const int oy = pix >> 4, ox = pix & 15;
    const unsigned char* bbase = sP + ((2 * oy) * PWP + 2 * ox + 2 * lh) * PXB;     // lane half: taps s0 + 2 lh, + 1
    const unsigned char* abase = sW + l31 * WRS + lh * 16;
#pragma unroll
    for (int r = 0; r < 7; ++r) {
#pragma unroll
      for (int hs = 0; hs < 2; ++hs) {                // taps 0..3 / 4..7 of filter row r
#pragma unroll
        for (int kk = 0; kk < RB / 32; ++kk) {        // f16: one 32-byte step, f32: two
          // f16: 16 B = taps (s0 + 2 lh, + 1) x 4 channels; f32: the K step's 64 bytes are taken 32 at a time, 16 per lane half
          const unsigned char* bp = ES == 2 ? bbase + (r * PWP + 4 * hs) * PXB
                                            : sP + ((2 * oy + r) * PWP + 2 * ox + 4 * hs) * PXB + kk * 32 + lh * 16;
          const u32x4 bf = *reinterpret_cast<const u32x4*>(bp);
          const int aoff = (r * 2 + hs) * RB + (ES == 2 ? 0 : kk * 32);
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const u32x4 af = *reinterpret_cast<const u32x4*>(abase + i * 32 * WRS + aoff);
            MMA<T>::run(af, bf, acc[i]);
          }
        }
      }
    }
    __syncthreads();                                  // all waves are done reading the patch: it becomes the output stage
#pragma unroll
    for (int i = 0; i < 2; ++i)
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
  (void)P_BYTES; (void)E_BYTES;
}

template <typename T> struct StemVec;
template <> struct StemVec<f16> { typedef f16x8 type; };
template <> struct StemVec<float> { typedef f32x4 type; };

// The same stem with the 3x3 stride-2 max pool that follows it (resnet.py:637 `x = self.maxpool(x)`) in the epilogue: the
// 275 MB conv output is neither written nor read by a pooling pass (378 + 344 MB -> 172 MB).  A workgroup owns a 4 x 8
// tile of POOLED pixels = a 9 x 17 tile of conv pixels (1.2x recompute at the tile seams), computed as above into LDS
// (zero outside the conv map: the activation is ReLU, so a zero never wins a max that holds a real value), then every
// thread reduces one (pooled pixel, 8-channel chunk) over its 3 x 3 window.
template <typename T>
__global__ __launch_bounds__(256) void resnet_stem_pool_kernel(const RStemArgs a) {
  constexpr int ES = (int)sizeof(T);
  constexpr int QH = 4, QW = 8, CH = 2 * QH + 1, CW = 2 * QW + 1, NC = CH * CW;      // pooled tile, conv tile (153 pixels)
  constexpr int PH = 2 * (CH - 1) + 7, PW = 2 * (CW - 1) + 7, PWP = 48, STRIP = 6, CO_T = 64;
  constexpr int PXB = 4 * ES, KROW = 224 * ES, WRS = KROW + 16, RB = 16 * ES, ORS = CO_T * ES + 16;
  constexpr int W_BYTES = CO_T * WRS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sW = smem;
  unsigned char* sP = smem + W_BYTES;                 // patch, later the conv tile [160 slots][ORS]
  constexpr int B_BYTES = (PH * PWP * PXB > 160 * ORS ? PH * PWP * PXB : 160 * ORS);
  unsigned char* sSB = sP + B_BYTES;                  // scale | bias (64 floats each), parked once per workgroup
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int t = blockIdx.x;
  const int sx = t % a.strips_x;
  t /= a.strips_x;
  const int ty = t % a.tiles_y, img = t / a.tiles_y;
  const int qy0 = ty * QH;                            // pooled row origin; conv row origin = 2 qy0 - 1
  const int Hc = (a.H + 6 - 7) / 2 + 1, Wc = (a.W + 6 - 7) / 2 + 1;

  for (int q = tid; q < CO_T * (KROW / 16); q += 256) {
    const int row = q / (KROW / 16), c = q - row * (KROW / 16);
    *reinterpret_cast<u32x4*>(sW + row * WRS + c * 16) = *reinterpret_cast<const u32x4*>(a.w + (long)row * KROW + c * 16);
  }
  if (tid < 32) *reinterpret_cast<f32x4*>(sSB + tid * 16) = *reinterpret_cast<const f32x4*>((tid < 16 ? a.scale : a.bias - 64) + tid * 4);
  const long plane = (long)a.H * a.W;
  const float* ibase = a.img + (long)img * 3 * plane;
  // image loads of one tile: a thread owns NL slots of the PH x PWP patch (columns >= PW are padding) and loads the three
  // channels of each, held in registers so that the NEXT tile's loads are in flight while this one is multiplied and pooled;
  // a slot then goes to LDS as ONE store of its four channels (c0 c1 c2 0) instead of three 2-byte stores and a zero pass
  constexpr int NS = PH * PWP, NL = (NS + 255) / 256;
  float pre[NL][3];
  auto issue_loads = [&](int qx0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int q = tid + i * 256;
      const int py = q / PWP, px = q - py * PWP;
      const int Y = 2 * (2 * qy0 - 1) - 3 + py, X = 2 * (2 * qx0 - 1) - 3 + px;
      const bool in = q < NS && px < PW && (unsigned)Y < (unsigned)a.H && (unsigned)X < (unsigned)a.W;
      const float* p0 = ibase + (long)Y * a.W + X;
#pragma unroll
      for (int c = 0; c < 3; ++c) pre[i][c] = in ? p0[c * plane] : 0.f;
    }
  };
  issue_loads(sx * STRIP * QW);
  // ten (cout block, pixel block) tiles of 32 x 32 over four waves: tile k = wave + 4 i -> cout block k & 1, pixel block k >> 1
  constexpr int NT = 3;
  for (int s = 0; s < STRIP; ++s) {
    const int tx = sx * STRIP + s;
    if (tx >= a.tiles_x) break;                       // uniform
    const int qx0 = tx * QW;
    __syncthreads();                                  // the previous tile's pooling is done with sP
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int q = tid + i * 256;
      if (q < NS) {
        T v[4] = {(T)pre[i][0], (T)pre[i][1], (T)pre[i][2], (T)0.f};
        if constexpr (sizeof(T) == 2) *reinterpret_cast<unsigned long long*>(sP + q * PXB) = *reinterpret_cast<const unsigned long long*>(v);
        else *reinterpret_cast<u32x4*>(sP + q * PXB) = *reinterpret_cast<const u32x4*>(v);
      }
    }
    if (s + 1 < STRIP && tx + 1 < a.tiles_x) issue_loads(qx0 + QW);
    __syncthreads();
    f32x16 acc[NT];
    int boff[NT], aoff0[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.0f;
      const int k = wave + 4 * i;
      int p = (k >> 1) * 32 + l31;
      p = p < NC ? p : NC - 1;                        // junk lanes of the last block read a valid pixel
      const int cy = p / CW, cx = p - cy * CW;
      boff[i] = ((2 * cy) * PWP + 2 * cx) * PXB + (ES == 2 ? 2 * lh * PXB : lh * 16);
      aoff0[i] = ((k & 1) * 32 + l31) * WRS + lh * 16;
    }
#pragma unroll
    for (int r = 0; r < 7; ++r) {
#pragma unroll
      for (int hs = 0; hs < 2; ++hs) {
#pragma unroll
        for (int kk = 0; kk < RB / 32; ++kk) {
          const int bo = (r * PWP + 4 * hs) * PXB + (ES == 2 ? 0 : kk * 32);
          const int ao = (r * 2 + hs) * RB + (ES == 2 ? 0 : kk * 32);
#pragma unroll
          for (int i = 0; i < NT; ++i) {
            if (wave + 4 * i < 10) {
              const u32x4 bf = *reinterpret_cast<const u32x4*>(sP + boff[i] + bo);
              const u32x4 af = *reinterpret_cast<const u32x4*>(sW + aoff0[i] + ao);
              MMA<T>::run(af, bf, acc[i]);
            }
          }
        }
      }
    }
    __syncthreads();                                  // all waves are done reading the patch: it becomes the conv tile
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int k = wave + 4 * i;
      const int p = (k >> 1) * 32 + l31;
      if (k < 10 && p < NC) {
        const int cy = p / CW, cx = p - cy * CW;
        const int cr = 2 * qy0 - 1 + cy, cc = 2 * qx0 - 1 + cx;
        const bool inside = (unsigned)cr < (unsigned)Hc && (unsigned)cc < (unsigned)Wc;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int co = (k & 1) * 32 + 8 * g + 4 * lh;
          const f32x4 sc = *reinterpret_cast<const f32x4*>(sSB + co * 4), bi = *reinterpret_cast<const f32x4*>(sSB + 256 + co * 4);
          const f32x4 xv = {acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]};
          f32x4 yv = scale_bias_act4<T>(xv, sc, bi, GLSDET_ACT_RELU);
          if (!inside) yv = f32x4{0.f, 0.f, 0.f, 0.f};
          const float v[4] = {yv[0], yv[1], yv[2], yv[3]};
          store4(sP + p * ORS + co * ES, v, (T*)nullptr);
        }
      }
    }
    __syncthreads();
    {                                                 // one (pooled pixel, 16-byte channel chunk) per thread
      constexpr int VO = 16 / ES, OCPR = CO_T / VO;
      for (int q = tid; q < QH * QW * OCPR; q += 256) {
        const int pq = q / OCPR, cq = q - pq * OCPR;
        const int py = pq / QW, px = pq - py * QW;
        const int ho = qy0 + py, wo = qx0 + px;
        if (ho < a.Ho && wo < a.Wo) {
          typedef typename StemVec<T>::type V;
          V m = *reinterpret_cast<const V*>(sP + ((2 * py) * CW + 2 * px) * ORS + cq * 16);
#pragma unroll
          for (int d = 1; d < 9; ++d) {
            const V v = *reinterpret_cast<const V*>(sP + ((2 * py + d / 3) * CW + 2 * px + d % 3) * ORS + cq * 16);
            m = __builtin_elementwise_max(m, v);
          }
          const long yo = (long)img * a.y_sn + (long)ho * a.y_sh + (long)wo * a.y_sw + cq * VO;
          *reinterpret_cast<V*>(a.y + yo * (long)ES) = m;
        }
      }
    }
  }
}

template <typename T>
static int launch_rstem_pool(const RStemArgs& a, hipStream_t st) {
  constexpr int ES = (int)sizeof(T);
  constexpr int patch = 23 * 48 * 4 * ES, ctile = 160 * (64 * ES + 16);
  constexpr int lds = 64 * (224 * ES + 16) + (patch > ctile ? patch : ctile) + 512;
  static bool attr_set = false;
  if (!attr_set && lds > 64 * 1024) {
    GLS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(resnet_stem_pool_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_set = true;
  }
  const long grid = (long)a.n * a.tiles_y * a.strips_x;
  hipLaunchKernelGGL((resnet_stem_pool_kernel<T>), dim3((unsigned)grid), dim3(256), lds, st, a);
  GLS_HIP(hipGetLastError());
  return 0;
}

template <typename T>
static int launch_rstem(const RStemArgs& a, hipStream_t st) {
  constexpr int ES = (int)sizeof(T);
  constexpr int patch = 21 * 48 * 4 * ES, epi = 128 * (64 * ES + 16);
  constexpr int lds = 64 * (224 * ES + 16) + (patch > epi ? patch : epi);
  static bool attr_set = false;
  if (!attr_set && lds > 64 * 1024) {
    GLS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(resnet_stem_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_set = true;
  }
  const long grid = (long)a.n * a.tiles_y * a.strips_x;
  hipLaunchKernelGGL((resnet_stem_kernel<T>), dim3((unsigned)grid), dim3(256), lds, st, a);
  GLS_HIP(hipGetLastError());
  return 0;
}

}  // namespace glsdet

using namespace glsdet;

extern "C" int64_t glsdet_resnet_stem_weight_elems(void) { return 64 * 224; }

extern "C" int glsdet_resnet_stem_pool(const float* img, int32_t n, int32_t cin, int32_t H, int32_t W, const void* w, const float* scale,
                                       const float* bias, const glsdet_view* y, void* stream) {
  if (!img || !w || !scale || !bias || !y) GLS_FAIL(GLSDET_E_ARG, "resnet_stem_pool: null argument");
  if (cin != 3 || n < 1 || H < 7 || W < 7) GLS_FAIL(GLSDET_E_ARG, "resnet_stem_pool: needs a 3-channel image of at least 7 x 7");
  if (((uintptr_t)img & 3) || ((uintptr_t)w | (uintptr_t)scale | (uintptr_t)bias) & 15) GLS_FAIL(GLSDET_E_ALIGN, "resnet_stem_pool: operand alignment");
  int rc;
  if ((rc = check_view(*y, "resnet_stem_pool.y"))) return rc;
  const int Hc = (H + 6 - 7) / 2 + 1, Wc = (W + 6 - 7) / 2 + 1, Ho = (Hc + 2 - 3) / 2 + 1, Wo = (Wc + 2 - 3) / 2 + 1;
  if (y->n != n || y->h != Ho || y->w != Wo || y->c != 64) GLS_FAIL(GLSDET_E_ARG, "resnet_stem_pool: output must be [n, %d, %d, 64]", Ho, Wo);
  RStemArgs a;
  a.img = img; a.w = (const unsigned char*)w; a.scale = scale; a.bias = bias;
  a.y = (unsigned char*)y->base; a.y_sn = y->sn; a.y_sh = y->sh; a.y_sw = y->sw;
  a.n = n; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo; a.cout = 64; a.act = GLSDET_ACT_RELU;
  a.tiles_x = (Wo + 7) / 8; a.tiles_y = (Ho + 3) / 4; a.strips_x = (a.tiles_x + 5) / 6;
  const int dt = y->dtype;
  OpRecord op;
  op.kind = 0;
  op.flops = 2.0 * (double)n * Hc * Wc * 64 * 147.0;
  op.bytes = (double)n * 3 * H * W * 4.0 + (double)n * Ho * Wo * 64 * dtype_size(dt);
  char nm[112];
  snprintf(nm, sizeof nm, "resnet_stem_pool<%s> 7x7 s2 cin3 cout64 + relu + maxpool 3x3 s2 (fp32 NCHW image in)", dt ? "f32" : "f16");
  op.name = nm;
  op.launch = [a, dt](hipStream_t st) -> int { return dt == GLSDET_F16 ? launch_rstem_pool<f16>(a, st) : launch_rstem_pool<float>(a, st); };
  return submit(std::move(op), stream);
}

extern "C" int glsdet_resnet_stem(const float* img, int32_t n, int32_t cin, int32_t H, int32_t W, const void* w, const float* scale,
                                  const float* bias, int32_t act, const glsdet_view* y, void* stream) {
  if (!img || !w || !scale || !bias || !y) GLS_FAIL(GLSDET_E_ARG, "resnet_stem: null argument");
  if (cin != 3 || n < 1 || H < 7 || W < 7) GLS_FAIL(GLSDET_E_ARG, "resnet_stem: needs a 3-channel image of at least 7 x 7");
  if (((uintptr_t)img & 3) || ((uintptr_t)w | (uintptr_t)scale | (uintptr_t)bias) & 15) GLS_FAIL(GLSDET_E_ALIGN, "resnet_stem: operand alignment");
  int rc;
  if ((rc = check_view(*y, "resnet_stem.y"))) return rc;
  const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
  if (y->n != n || y->h != Ho || y->w != Wo || y->c != 64) GLS_FAIL(GLSDET_E_ARG, "resnet_stem: output must be [n, %d, %d, 64]", Ho, Wo);
  if (act < 0 || act > 5) GLS_FAIL(GLSDET_E_ARG, "resnet_stem: bad act %d", act);
  RStemArgs a;
  a.img = img; a.w = (const unsigned char*)w; a.scale = scale; a.bias = bias;
  a.y = (unsigned char*)y->base; a.y_sn = y->sn; a.y_sh = y->sh; a.y_sw = y->sw;
  a.n = n; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo; a.cout = 64; a.act = act;
  a.tiles_x = (Wo + 15) / 16; a.tiles_y = (Ho + 7) / 8; a.strips_x = (a.tiles_x + 5) / 6;
  const int dt = y->dtype;
  OpRecord op;
  op.kind = 0;
  op.flops = 2.0 * (double)n * Ho * Wo * 64 * 147.0;
  op.bytes = (double)n * 3 * H * W * 4.0 + (double)n * Ho * Wo * 64 * dtype_size(dt);
  char nm[96];
  snprintf(nm, sizeof nm, "resnet_stem<%s,64x8x16> 7x7 s2 cin3 cout64 (fp32 NCHW image in)", dt ? "f32" : "f16");
  op.name = nm;
  op.launch = [a, dt](hipStream_t st) -> int { return dt == GLSDET_F16 ? launch_rstem<f16>(a, st) : launch_rstem<float>(a, st); };
  return submit(std::move(op), stream);
}

extern "C" int glsdet_focus_conv_down(const float* img, int32_t n, int32_t cin, int32_t H, int32_t W, const void* w1, const float* scale1,
                                      const float* bias1, int32_t act1, int32_t c1, const void* w2, const float* scale2,
                                      const float* bias2, int32_t act2, const glsdet_view* y, void* stream) {
  if (!img || !w1 || !scale1 || !bias1 || !w2 || !scale2 || !bias2 || !y) GLS_FAIL(GLSDET_E_ARG, "focus_conv_down: null argument");
  if (cin != 3 || n < 1 || H < 4 || W < 4 || (H & 1) || (W & 1)) GLS_FAIL(GLSDET_E_ARG, "focus_conv_down: needs a 3-channel image with even H and W");
  if (((uintptr_t)img & 7) || ((uintptr_t)w1 | (uintptr_t)scale1 | (uintptr_t)bias1 | (uintptr_t)w2 | (uintptr_t)scale2 | (uintptr_t)bias2) & 15)
    GLS_FAIL(GLSDET_E_ALIGN, "focus_conv_down: operand alignment");
  int rc;
  if ((rc = check_view(*y, "focus_conv_down.y"))) return rc;
  const int Hs = H / 2, Ws = W / 2, Ho = (Hs + 2 - 3) / 2 + 1, Wo = (Ws + 2 - 3) / 2 + 1;
  if (c1 != 32 || y->n != n || y->h != Ho || y->w != Wo || y->c % 8 || y->c > 64)
    GLS_FAIL(GLSDET_E_ARG, "focus_conv_down: stem of 32 channels, output [n, %d, %d, <= 64 channels]", Ho, Wo);
  if (act1 < 0 || act1 > 5 || act2 < 0 || act2 > 5) GLS_FAIL(GLSDET_E_ARG, "focus_conv_down: bad act");
  Stem2Args b = {};
  ConvArgs& a = b.c;
  const int dt = y->dtype, es = dtype_size(dt);
  a.w = (const unsigned char*)w2; a.scale = scale2; a.bias = bias2;
  a.y = (unsigned char*)y->base; a.y_sn = y->sn; a.y_sh = y->sh; a.y_sw = y->sw;
  a.res = nullptr;
  a.N = n; a.H = Hs; a.W = Ws; a.Cin = 32; a.Ho = Ho; a.Wo = Wo; a.Cout = y->c; a.cout_pad = glsdet_conv_cout_pad(y->c);
  a.R = a.S = 3; a.stride = 2; a.pad = 1; a.act = act2; a.act_post = 0;
  a.kreal = 9 * 32; a.kpad = glsdet_conv_kpad(3, 3, 32, dt);
  a.w_bytes = (unsigned)((int64_t)a.cout_pad * a.kpad * es);
  a.M = n * Ho * Wo;
  b.img = img; b.w1 = (const unsigned char*)w1; b.scale1 = scale1; b.bias1 = bias1;
  b.IH = H; b.IW = W; b.kpad1 = glsdet_conv_kpad(3, 3, 16, dt); b.act1 = act1;
  a.dbg = getenv("GLSDET_STEM2_DBG") ? atoi(getenv("GLSDET_STEM2_DBG")) : 0;      // knock-out timing experiments (results invalid)
  OpRecord op;
  op.kind = 0;
  op.flops = 2.0 * (double)n * Hs * Ws * 32 * 108.0 + 2.0 * (double)n * Ho * Wo * y->c * 288.0;
  op.bytes = (double)n * 3 * H * W * 4.0 + (double)n * Ho * Wo * y->c * es;
  char nm[112];
  snprintf(nm, sizeof nm, "focus_stem_down<%s> 3x3 cin12 cout32 -> 3x3 s2 cout%d (fp32 NCHW image in)", dt ? "f32" : "f16", y->c);
  op.name = nm;
  op.launch = [b, dt](hipStream_t st) -> int { return dt == GLSDET_F16 ? launch_stem_down<f16>(b, st) : launch_stem_down<float>(b, st); };
  return submit(std::move(op), stream);
}

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
