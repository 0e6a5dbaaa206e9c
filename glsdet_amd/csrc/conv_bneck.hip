// Bottleneck front in ONE launch: y = act2(bn2(conv3x3(act1(bn1(conv1x1(x)))))) [+ res]
// (drone/models/base/darknet.py:61-64 `y = self.conv2(self.conv1(x)); if self.use_add: y = y + x`).
//
// The hidden tensor of a Bottleneck is written and read back by nobody else, and on the large early maps both convs are
// HBM / latency bound (round 2 op table: 1x1 64->64 @100x168 17 us + 3x3 64->64 30 us, 2 TB/s each).  Here the
// workgroup that owns an 8 x 16 output tile computes the 1x1 on the tile's 10 x 18 HALO (1.4x the pixels, a sixth of the
// 3x3's MFMAs) straight into the LDS patch the 3x3 reads:
//   phase A  x patch (180 pixels, one 64/128-byte channel chunk at a time) and the 1x1 weights -> LDS, registers staged;
//            D1[cm][192 px] on the MFMAs; scale / bias / act, ZERO outside the image (the 3x3 pads its INPUT with
//            zeros, not with act(bias)), rounded to the storage type, -> patch P1 (all CM channels per pixel);
//   phase B  the LDS-DMA weight ring of conv_halo_ring_kernel (conv_halo.hip) over the resident patch: no patch
//            exchange between channel chunks, taps x chunks back to back;
//   epilogue as the halo kernels (wide staging + residual, one rounding).
// The hidden values are rounded exactly as a stored tensor would be and both products run in the k order of the
// stand-alone kernels: the result equals the two-launch form bit for bit (tests/test_hip_model.py).
// LDS: [ring | W1 chunk] + [x chunk | P1] (the x chunk dies before P1 is written): CM 64 fp16 = 51 KB -> 3 workgroups / CU.
#include "conv_common.h"

namespace glsdet {

template <int N>
__device__ __forceinline__ void bn_wait_vm_barrier() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

template <typename T, int CM, int CO_T, int KB, int RING>
struct BneckGeom {
  static constexpr int ES = (int)sizeof(T);
  static constexpr int PH = 10, PW = 18, NSLOT = PH * PW;
  static constexpr int XS = KB + 16;                       // phase A rows (x chunk, W1 chunk)
  static constexpr int RS1 = CM * ES + 16;                 // P1 rows: all CM channels of a patch pixel
  static constexpr int RING_BYTES = RING * CO_T * KB;
  static constexpr int W1_BYTES = CM * XS;
  static constexpr int REG_A = ((RING_BYTES > W1_BYTES ? RING_BYTES : W1_BYTES) + 1023) / 1024 * 1024;
  static constexpr int XC_BYTES = 192 * XS, P1_BYTES = NSLOT * RS1;
  static constexpr int REG_B = XC_BYTES > P1_BYTES ? XC_BYTES : P1_BYTES;
  static constexpr int STAGE = REG_A + REG_B;
  // parked scale | bias of the 1x1 (CM * 8 bytes) and of the 3x3's cout tile (CO_T * 8 bytes): behind everything else
  static constexpr int epi_w = 128 * (CO_T * 4 + 16);
  static constexpr int SB_OFF = ((STAGE > epi_w ? STAGE : epi_w) + 15) / 16 * 16;
  static constexpr int LDS = SB_OFF + (CM + CO_T) * 8;
};

template <typename T, int CM, int CO_T, int KB, int RING>
__global__ __launch_bounds__(256) void conv_bneck_kernel(const BneckArgs b, const int tiles_x, const int tiles_y) {
  using G = BneckGeom<T, CM, CO_T, KB, RING>;
  const ConvArgs& a = b.c;
  constexpr int ES = G::ES, VEC = 16 / ES, KE = KB / ES;
  constexpr int PW = G::PW, NSLOT = G::NSLOT, XS = G::XS, RS1 = G::RS1;
  constexpr int CPRW = KB / 16, RPL = 256 / KB;
  constexpr int P1_OFF = G::REG_A, XC_OFF = G::REG_A, W1_OFF = 0;
  constexpr int WCO = 2, WPX = 2;
  constexpr int WT_CO = CO_T / WCO, WT_PX = 128 / WPX;
  constexpr int TM = WT_CO / 32, TN = WT_PX / 32;
  constexpr int A_BYTES = CO_T * KB;
  constexpr int RPI = 64 / CPRW, NI = CO_T / RPI / 4;
  static_assert(CO_T % (RPI * 4) == 0 && RING >= 3 && (CM * ES) % KB == 0 && CM % 32 == 0, "geometry");
  constexpr int NCH = CM * ES / KB;                      // channel chunks of the hidden tensor
  constexpr int NSTEPS = NCH * 9;

  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  int tile;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, local = bid >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
  }
  int rest = gls_div(tile, a.nco_mul, a.nco_sh);
  const int co0 = (tile - rest * a.n_co_tiles) * CO_T;
  const int r1 = gls_div(rest, a.tx_mul, a.tx_sh);
  const int tx0 = (rest - r1 * tiles_x) * 16;
  const int img = gls_div(r1, a.ty_mul, a.ty_sh);
  const int ty0 = (r1 - img * tiles_y) * 8;

  const auto xrs = gls_make_rsrc(a.x_lo, a.x_bytes);
  const auto wrs = gls_make_rsrc(a.w, a.w_bytes);
  const auto w0rs = gls_make_rsrc(b.w0, b.w0_bytes);
  // both convs' folded-BN scale / bias: requested first, parked in LDS once the first x chunk has landed anyway
  unsigned char* sSB0 = smem + G::SB_OFF;                  // [scale0 | bias0] (CM floats each)
  unsigned char* sSB = sSB0 + CM * 8;                      // [scale | bias] of this cout tile (CO_T floats each)
  f32x4 sbv = {0.f, 0.f, 0.f, 0.f};
  {
    constexpr int N0 = CM / 4, N1 = CO_T / 4;              // f32x4 pieces: scale0, bias0, scale, bias
    if (tid < N0) sbv = *reinterpret_cast<const f32x4*>(b.scale0 + tid * 4);
    else if (tid < 2 * N0) sbv = *reinterpret_cast<const f32x4*>(b.bias0 + (tid - N0) * 4);
    else if (tid < 2 * N0 + N1) { if (co0 + (tid - 2 * N0) * 4 < a.cout_pad) sbv = *reinterpret_cast<const f32x4*>(a.scale + co0 + (tid - 2 * N0) * 4); }
    else if (tid < 2 * N0 + 2 * N1) { if (co0 + (tid - 2 * N0 - N1) * 4 < a.cout_pad) sbv = *reinterpret_cast<const f32x4*>(a.bias + co0 + (tid - 2 * N0 - N1) * 4); }
  }

  // ------------------------------------------------------------------ phase A: the 1x1 on the halo
  {
    constexpr int NP = (NSLOT * CPRW + 255) / 256;       // x chunk pieces per thread
    constexpr int NW = (CM * CPRW + 255) / 256;          // W1 chunk pieces per thread
    constexpr int NCB = CM / 32;                         // cout blocks of the hidden tensor
    constexpr int WPB = 4 / NCB;                         // waves that share a cout block
    constexpr int NT = (6 + WPB - 1) / WPB;              // pixel blocks (of 32, six cover the 180 patch pixels) per wave
    static_assert(NCB == 1 || NCB == 2 || NCB == 4, "hidden channels: 32, 64 or 128");
    const int kc = tid % CPRW;
    unsigned poff[NP], woff[NW];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int q = tid + i * 256, pp = q / CPRW;
      const int py = pp / PW, px = pp - py * PW;
      const int hi = ty0 - 1 + py, wi = tx0 - 1 + px;
      const bool ok = pp < NSLOT && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
      poff[i] = ok ? a.x_off + (unsigned)(((long)img * a.x_sn + (long)hi * a.x_sh + (long)wi * a.x_sw + kc * VEC) * (long)ES) : GLS_OOB;
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int q = tid + i * 256, row = q / CPRW;
      woff[i] = row < CM ? (unsigned)((row * b.kpad0 + kc * VEC) * ES) : GLS_OOB;
    }
    f32x16 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.0f;
    const int cb = wave % NCB, pb0 = wave / NCB;
    const int nch0 = b.cin0 / KE;
    u32x4 rp[NP], rw[NW];
    auto load_chunk = [&](int cc) __attribute__((always_inline)) {
      const unsigned coff = (unsigned)(cc * KB);
#pragma unroll
      for (int i = 0; i < NP; ++i) rp[i] = gls_buf_load16(xrs, poff[i] + coff);
#pragma unroll
      for (int i = 0; i < NW; ++i) rw[i] = gls_buf_load16(w0rs, woff[i] + coff);
    };
    load_chunk(0);
    for (int cc = 0; cc < nch0; ++cc) {
      if (cc) __syncthreads();                           // the previous chunk has been multiplied by every wave
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int q = tid + i * 256;
        if (q < NSLOT * CPRW) *reinterpret_cast<u32x4*>(smem + XC_OFF + (q / CPRW) * XS + kc * 16) = rp[i];
      }
#pragma unroll
      for (int i = 0; i < NW; ++i) {
        const int q = tid + i * 256;
        if (q < CM * CPRW) *reinterpret_cast<u32x4*>(smem + W1_OFF + (q / CPRW) * XS + kc * 16) = rw[i];
      }
      if (cc == 0 && tid < (CM + CO_T) / 2) *reinterpret_cast<f32x4*>(sSB0 + tid * 16) = sbv;
      if (cc + 1 < nch0) load_chunk(cc + 1);
      __syncthreads();
      const unsigned char* sA = smem + W1_OFF + (cb * 32 + l31) * XS + lh * 16;
      const unsigned char* sB = smem + XC_OFF + l31 * XS + lh * 16;
#pragma unroll
      for (int kk = 0; kk < KB / 32; ++kk) {
        const u32x4 af = *reinterpret_cast<const u32x4*>(sA + kk * 32);
#pragma unroll
        for (int i = 0; i < NT; ++i) {
          const int pb = pb0 + i * WPB;
          if (pb < 6) {
            const u32x4 bf = *reinterpret_cast<const u32x4*>(sB + pb * 32 * XS + kk * 32);
            MMA<T>::run(af, bf, acc[i]);
          }
        }
      }
    }
    __syncthreads();                                     // every wave is done with the x chunk: its bytes become P1
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int pb = pb0 + i * WPB;
      const int p = pb * 32 + l31;
      const int py = p / PW, px = p - py * PW;
      const int hi = ty0 - 1 + py, wi = tx0 - 1 + px;
      const bool inside = (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
      if (pb < 6 && p < NSLOT) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int co = cb * 32 + 8 * g + 4 * lh;
          const f32x4 sc = *reinterpret_cast<const f32x4*>(sSB0 + co * 4), bi = *reinterpret_cast<const f32x4*>(sSB0 + CM * 4 + co * 4);
          const f32x4 xv = {acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]};
          f32x4 yv = scale_bias_act4<T>(xv, sc, bi, b.act0);
          if (!inside) yv = f32x4{0.f, 0.f, 0.f, 0.f};   // the 3x3's zero padding
          const float v[4] = {yv[0], yv[1], yv[2], yv[3]};
          store4(smem + P1_OFF + p * RS1 + co * ES, v, (T*)nullptr);
        }
      }
    }
  }
  bn_wait_vm_barrier<0>();                               // P1 is visible; the W1 bytes may become the ring

  // ------------------------------------------------------------------ phase B: the 3x3 from the resident patch
  unsigned wd[NI];
#pragma unroll
  for (int q = 0; q < NI; ++q) {
    const int row = RPI * (wave + 4 * q) + lane / CPRW;
    const int ch = (lane % CPRW) ^ ((row / RPL) & (CPRW - 1));
    const bool ok = (co0 + row) < a.cout_pad;
    wd[q] = ok ? (unsigned)(((co0 + row) * a.kpad + ch * VEC) * ES) : GLS_OOB;
  }
  typedef __attribute__((address_space(3))) void* lds_ptr;
  int dg = 0, dtap = 0, dslot = 0;
  unsigned dadd = 0;
  const unsigned tap_bytes = (unsigned)(CM * ES);
  const unsigned chunk_fix = (unsigned)KB - 9u * tap_bytes;
  auto dma_next = [&]() __attribute__((always_inline)) {
    unsigned char* dst = smem + dslot * A_BYTES + wave * 1024;
#pragma unroll
    for (int q = 0; q < NI; ++q)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_ptr)(dst + q * 4096), 16, (int)(wd[q] + dadd), 0, 0, 0);
    dslot = dslot + 1 == RING ? 0 : dslot + 1;
    dadd += tap_bytes;
    if (++dtap == 9) {
      dtap = 0;
      dadd += chunk_fix;
    }
    if (++dg >= NSTEPS) dadd = GLS_OOB;
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
  const int wco = wave % WCO, wpx = wave / WCO;
  const int a_row = (wco * WT_CO + l31) * KB;
  int a_sw[KB / 32];
#pragma unroll
  for (int kk = 0; kk < KB / 32; ++kk) a_sw[kk] = ((2 * kk + lh) ^ ((l31 / RPL) & (CPRW - 1))) << 4;
  int b_off[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int pix = wpx * WT_PX + j * 32 + l31;
    int oy, ox;
    pix_to_xy16<PW>(pix, oy, ox);
    b_off[j] = P1_OFF + (oy * PW + ox) * RS1 + lh * 16;
  }
#pragma unroll
  for (int g = 0; g < RING - 1; ++g) dma_next();
  int g = 0;
  for (int cc = 0; cc < NCH; ++cc) {
    int tap_off = cc * KB, ts = 0;
    for (int tap = 0; tap < 9; ++tap) {
      bn_wait_vm_barrier<NI * (RING - 2)>();             // tap g landed in every wave; slot g-1 is free
      dma_next();
      const unsigned char* sA = smem + g * A_BYTES + a_row;
#pragma unroll
      for (int kk = 0; kk < KB / 32; ++kk) {
        u32x4 af[TM], bf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const u32x4*>(sA + i * 32 * KB + a_sw[kk]);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const u32x4*>(smem + b_off[j] + tap_off + kk * 32);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) MMA<T>::run(af[i], bf[j], acc[i][j]);
      }
      g = g + 1 == RING ? 0 : g + 1;
      ++ts;
      tap_off += (ts == 3) ? (PW - 2) * RS1 : RS1;
      ts = (ts == 3) ? 0 : ts;
    }
  }
  bn_wait_vm_barrier<0>();                               // the zero fills of the tail have landed; all waves done reading

  const bool wide = sizeof(T) == 2 && a.res != nullptr;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int co_l = wco * WT_CO + i * 32 + 8 * gq + 4 * lh;
      const f32x4 sc = *reinterpret_cast<const f32x4*>(sSB + co_l * 4), bi = *reinterpret_cast<const f32x4*>(sSB + CO_T * 4 + co_l * 4);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int px_l = wpx * WT_PX + j * 32 + l31;
        const f32x4 xv = {acc[i][j][4 * gq], acc[i][j][4 * gq + 1], acc[i][j][4 * gq + 2], acc[i][j][4 * gq + 3]};
        const f32x4 yv = scale_bias_act4<T>(xv, sc, bi, a.act);
        const float v[4] = {yv[0], yv[1], yv[2], yv[3]};
        stage4<T, CO_T>(smem, px_l, co_l, v, wide);
      }
    }
  }
  __syncthreads();
  halo_store_tile<T, CO_T, PW>(smem, a, img, ty0, tx0, co0, tid);
}

template <typename T, int CM, int CO_T, int KB, int RING>
static int launch_bneck(const BneckArgs& b0, hipStream_t st) {
  using G = BneckGeom<T, CM, CO_T, KB, RING>;
  constexpr int ldsw = G::LDS;
  const int lds = G::LDS;
  auto kern = conv_bneck_kernel<T, CM, CO_T, KB, RING>;
  static int attr_lds = 64 * 1024;
  if (ldsw > attr_lds) {
    GLS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, ldsw));
    attr_lds = ldsw;
  }
  BneckArgs b = b0;
  ConvArgs& a = b.c;
  a.n_co_tiles = (a.Cout + CO_T - 1) / CO_T;
  const int tiles_x = (a.Wo + 15) / 16, tiles_y = (a.Ho + 7) / 8;
  gls_fastdiv(a.n_co_tiles, &a.nco_mul, &a.nco_sh);
  gls_fastdiv(tiles_x, &a.tx_mul, &a.tx_sh);
  gls_fastdiv(tiles_y, &a.ty_mul, &a.ty_sh);
  const long grid = (long)a.n_co_tiles * tiles_x * tiles_y * a.N;
  if (grid <= 0 || grid > 0x7fffffffL) GLS_FAIL(GLSDET_E_ARG, "bottleneck: grid %ld out of range", grid);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, st, b, tiles_x, tiles_y);
  GLS_HIP(hipGetLastError());
  return 0;
}

// hint 0: 128-byte channel chunks where the hidden rows allow them; 1: 64-byte chunks (less LDS, more barriers)
int conv_bneck_try(const BneckArgs& b, int dt, int hint, OpRecord* op) {
  const ConvArgs& a = b.c;
  const int es = dtype_size(dt);
  const int cm = a.Cin;
  if (a.R != 3 || a.S != 3 || a.stride != 1 || a.pad != 1 || a.Cout != cm) return 1;
  if (cm != 32 && cm != 64 && cm != 128) return 1;
  if (dt == GLSDET_F32 && cm == 128 && hint == 0) hint = 1;          // 128-byte chunks would need 144 KB
  const int kb = (hint == 1 || cm * es == 64) ? 64 : 128;
  if ((cm * es) % kb || (b.cin0 * es) % kb) return 1;
  if (dt == GLSDET_F32 && cm == 128 && kb == 128) return 1;
  char nm[112];
  snprintf(nm, sizeof nm, "conv_bneck<%s,cm%d,kb%d> 1x1 cin%d -> 3x3 s1 cout%d", dt ? "f32" : "f16", cm, kb, b.cin0, a.Cout);
  op->name = nm;
  op->launch = [b, dt, cm, kb](hipStream_t st) -> int {
    if (dt == GLSDET_F16) {
      if (cm == 32) return launch_bneck<f16, 32, 64, 64, 4>(b, st);
      if (cm == 64) return kb == 128 ? launch_bneck<f16, 64, 64, 128, 3>(b, st) : launch_bneck<f16, 64, 64, 64, 4>(b, st);
      return kb == 128 ? launch_bneck<f16, 128, 128, 128, 3>(b, st) : launch_bneck<f16, 128, 128, 64, 4>(b, st);
    }
    if (cm == 32) return kb == 128 ? launch_bneck<float, 32, 64, 128, 3>(b, st) : launch_bneck<float, 32, 64, 64, 4>(b, st);
    if (cm == 64) return kb == 128 ? launch_bneck<float, 64, 64, 128, 3>(b, st) : launch_bneck<float, 64, 64, 64, 4>(b, st);
    return launch_bneck<float, 128, 128, 64, 4>(b, st);
  };
  return 0;
}

}  // namespace glsdet
