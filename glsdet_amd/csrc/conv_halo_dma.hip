// Stride-1 k x k convolution (k = 3, 5, 7), fp16 -> fp16: PERSISTENT workgroups, every operand staged by
// LDS-DMA (`buffer_load_dwordx4 ... lds`) several steps ahead of its use, across tile boundaries.
//
// conv_halo.hip runs one output tile per workgroup: kernel-argument loads, address set-up, the first
// memory round trip, the epilogue's scale / bias loads and the dispatch of the next workgroup are all
// exposed once per tile -- measured 6.5 us of an 8.7 us tile on a 3x3 128->128 layer whose MFMAs take
// 2.2 us -- and inside the loop the weight tile of a tap is requested ONE tap (~250-500 cycles) ahead of
// an L2 round trip several times that.  Here
//   * one workgroup per CU walks its tiles; the operand stream is one flat list of STAGES over all of
//     them (a stage = one filter ROW of a channel chunk: k taps, k*TM*TN*KB/32 MFMAs per wave);
//   * the weight tiles of a stage (k x CO_T rows) travel global -> LDS by DMA, no registers, into a ring
//     of D stages: the DMA of stage g+D-1 is issued when stage g starts -- also when g+D-1 already
//     belongs to the NEXT tile;
//   * the input patch of a chunk ((8+k-1) x (16+k-1) pixels) travels the same way into one of two
//     buffers a whole chunk ahead, the tile's folded-BN scale / bias with its first patch;
//   * one raw s_barrier per stage and a COUNTED s_waitcnt vmcnt(N) (never 0: the younger DMAs stay in
//     flight across the barrier); the epilogue has its own LDS staging area and overlaps the DMAs of
//     the next tile.  (vmcnt also counts the epilogue's stores: they can only make a counted wait
//     longer, never too short -- at most N operations of ANY kind are outstanding after it, so at most
//     N loads, and loads complete in order.)
// An LDS-DMA instruction writes 64 lanes x 16 B = 1 KiB of consecutive LDS, so rows cannot be padded;
// bank conflicts are avoided by an XOR swizzle applied on the SOURCE side: 16-byte slot c of row r holds
// channel chunk c ^ f(r), f(r) = (r / rows-per-256-B) mod chunks-per-row, and a fragment read asks for
// slot kc ^ f(r).  Out-of-range lanes of a buffer DMA write zeros (measured: tools/probe/glds_probe.hip):
// zero padding of the patch border and of the cout tail needs no branch.
#include "conv_common.h"

namespace glsdet {

template <int N>
__device__ __forceinline__ void wait_vm_barrier() {
  // all but the newest N memory operations of this wave are done; then every wave of the group is here
#ifdef DMA_DBG_NOWAIT
  asm volatile("s_barrier" ::: "memory");
#else
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
#endif
}
__device__ __forceinline__ void wait_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // no vmcnt: the DMAs of the next tile stay in flight
}

template <int CO_T, int KS, int KB, int D, int NWV>
struct HaloDma {
  static constexpr int TH = 8, TW = 16, PX_T = 128;
  static constexpr int PH = TH + KS - 1, PW = TW + KS - 1, PROWS = PH * PW;
  static constexpr int CH = KB / 16;              // 16-byte chunks per row
  static constexpr int RPI = 64 / CH;             // rows one DMA wave-instruction fills
  static constexpr int RPL = 256 / KB;            // rows per 256 bytes (one sweep of the 64 banks)
  static constexpr int KE = KB / 2;               // fp16 channels per chunk
  static constexpr int AR = KS * CO_T;            // weight rows of a stage
  static constexpr int A_BYTES = AR * KB;
  static constexpr int NI_A = AR / RPI / NWV;     // DMA instructions per wave per stage
  static constexpr int PR = ((PROWS + RPI * NWV - 1) / (RPI * NWV)) * (RPI * NWV);
  static constexpr int P_BYTES = PR * KB;
  static constexpr int NI_P = PR / RPI / NWV;
  static constexpr int P_OFF = D * A_BYTES;
  static constexpr int E_OFF = P_OFF + 2 * P_BYTES;                    // epilogue staging
  static constexpr int ORS = CO_T * 2 + 16;
  static constexpr int SB_OFF = E_OFF + PX_T * ORS;                    // 2 x (scale[CO_T] | bias[CO_T]) fp32
  static constexpr int SB_BYTES = 1024;                                // one DMA instruction each for scale and bias
  static constexpr int LDS = SB_OFF + 2 * 2 * SB_BYTES;
  static_assert(AR % (RPI * NWV) == 0 && (NWV == 4 || NWV == 8) && D >= 2 && D - 1 <= KS && LDS <= 160 * 1024 && CO_T * 4 <= SB_BYTES, "stage shape");
};

template <int CO_T, int KS, int KB, int D, int NWV>
__global__ __launch_bounds__(64 * NWV) void conv_halo_dma_kernel(const ConvArgs a, const int tiles_x, const int tiles_y,
                                                            const int ntiles) {
  using G = HaloDma<CO_T, KS, KB, D, NWV>;
  using T = f16;
  using TO = f16;
  constexpr int TH = G::TH, TW = G::TW, PX_T = G::PX_T, PW = G::PW, PROWS = G::PROWS;
  constexpr int CH = G::CH, RPI = G::RPI, RPL = G::RPL, KE = G::KE;
  constexpr int A_BYTES = G::A_BYTES, NI_A = G::NI_A, P_BYTES = G::P_BYTES, NI_P = G::NI_P, P_OFF = G::P_OFF;
  constexpr int E_OFF = G::E_OFF, ORS = G::ORS, SB_OFF = G::SB_OFF, SB_BYTES = G::SB_BYTES;
  // 4 waves: 2 x 2 (each CO_T/2 x 64 pixels); 8 waves: CO_T/32 x (256/CO_T) waves of 32 cout rows --
  // two waves per SIMD cover each other's LDS and issue latencies, which a lone wave per SIMD cannot
  constexpr int WCO = NWV == 8 ? CO_T / 32 : 2, WPXN = NWV / WCO;
  constexpr int WT_CO = CO_T / WCO, WT_PX = PX_T / WPXN;
  constexpr int TM = WT_CO / 32, TN = WT_PX / 32;
  static_assert(TM >= 1 && TN >= 1 && WCO * WPXN == NWV, "wave grid");
  constexpr int NKK = KB / 32;
  constexpr int pad = KS / 2;

  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- this workgroup's tiles: round k covers tiles [k*nwg, (k+1)*nwg); inside a round XCD x (= bid & 7)
  // takes a contiguous run, cout tile fastest, so neighbours in x share halo and weights in that XCD's L2
  const int nwg = gridDim.x;                                    // a multiple of 8
  const int toff = (blockIdx.x & 7) * (nwg >> 3) + (blockIdx.x >> 3);
  const int mytiles = toff < ntiles ? (ntiles - toff + nwg - 1) / nwg : 0;
  if (mytiles == 0) return;
  struct Tile { int co0, tx0, ty0, img; };
  auto decode = [&](int k) __attribute__((always_inline)) {
    const int tile = k * nwg + toff;
    Tile t;
    t.co0 = (tile % a.n_co_tiles) * CO_T;
    int rest = tile / a.n_co_tiles;
    t.tx0 = (rest % tiles_x) * TW;
    rest /= tiles_x;
    t.ty0 = (rest % tiles_y) * TH;
    t.img = rest / tiles_y;
    return t;
  };

  const auto xrs = gls_make_rsrc(a.x_lo, a.x_bytes);
  const auto wrs = gls_make_rsrc(a.w, a.w_bytes);
  const auto srs = gls_make_rsrc(a.scale, (unsigned)a.cout_pad * 4u);
  const auto brs = gls_make_rsrc(a.bias, (unsigned)a.cout_pad * 4u);
  const int nchunks = a.Cin / KE;
  const int S = nchunks * KS;                      // stages per tile

  // ---- DMA source offsets of this lane (bytes): row = RPI * instruction + lane / CH, slot = lane % CH
  const int lrow = lane / CH, lslot = lane % CH;
  unsigned woff[NI_A], poff[NI_P];
  auto set_woff = [&](int co0) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < NI_A; ++q) {
      const int row = RPI * (wave + NWV * q) + lrow;              // sp * CO_T + co
      const int sp = row / CO_T, co = row % CO_T;
      const int kc = lslot ^ ((row / RPL) & (CH - 1));
      const bool ok = (co0 + co) < a.cout_pad;
      woff[q] = ok ? (unsigned)(((co0 + co) * a.kpad + sp * a.Cin + kc * 8) * 2) : GLS_OOB;
    }
  };
  auto set_poff = [&](const Tile& t) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < NI_P; ++q) {
      const int prow = RPI * (wave + NWV * q) + lrow;
      const int py = prow / PW, px = prow - py * PW;
      const int hi = t.ty0 - pad + py, wi = t.tx0 - pad + px;
      const int kc = lslot ^ ((prow / RPL) & (CH - 1));
      const bool ok = prow < PROWS && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
      poff[q] = ok ? a.x_off + (unsigned)(((long)t.img * a.x_sn + (long)hi * a.x_sh + (long)wi * a.x_sw + kc * 8) * 2L) : GLS_OOB;
    }
  };

  typedef __attribute__((address_space(3))) void* lds_ptr;
  // weight tiles of stage (cc, r) of the tile woff describes -> ring slot; live = false: zeros (keeps the count)
  auto dma_A = [&](int slot, int sa, bool live) __attribute__((always_inline)) {
    const int cc = sa / KS, r = sa - cc * KS;
    const unsigned add = (unsigned)((r * KS * a.Cin + cc * KE) * 2);
    unsigned char* dst = smem + slot * A_BYTES + wave * 1024;
#pragma unroll
    for (int q = 0; q < NI_A; ++q) {
      const unsigned v = (live && woff[q] != GLS_OOB) ? woff[q] + add : GLS_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_ptr)(dst + q * NWV * 1024), 16, (int)v, 0, 0, 0);
    }
  };
  auto dma_P = [&](int buf, int cc) __attribute__((always_inline)) {      // patch of chunk cc of the tile poff describes
    const unsigned add = (unsigned)(cc * KE * 2);
    unsigned char* dst = smem + P_OFF + buf * P_BYTES + wave * 1024;
#pragma unroll
    for (int q = 0; q < NI_P; ++q) {
      const unsigned v = poff[q] != GLS_OOB ? poff[q] + add : GLS_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_ptr)(dst + q * NWV * 1024), 16, (int)v, 0, 0, 0);
    }
  };
  auto dma_SB = [&](int buf, int co0) __attribute__((always_inline)) {    // wave 0: the tile's scale and bias
    if (wave == 0) {
      const unsigned v = (lane * 4 < CO_T && co0 + lane * 4 < a.cout_pad) ? (unsigned)((co0 + lane * 4) * 4) : GLS_OOB;
      unsigned char* dst = smem + SB_OFF + buf * 2 * SB_BYTES;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srs, (lds_ptr)dst, 16, (int)v, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(brs, (lds_ptr)(dst + SB_BYTES), 16, (int)v, 0, 0, 0);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  const int wco = wave % WCO, wpx = wave / WCO;
  const int l31 = lane & 31, lh = lane >> 5;
  // fragment addressing: byte offset of the row + swizzled slot of chunk (2 kk + lh)
  const int a_x = (l31 / RPL) & (CH - 1);                      // f(row) of the weight rows: the tap / tile offsets are multiples of 32 rows
  int a_row[TM], a_sl[NKK];
#pragma unroll
  for (int i = 0; i < TM; ++i) a_row[i] = (wco * WT_CO + i * 32 + l31) * KB;
#pragma unroll
  for (int kk = 0; kk < NKK; ++kk) a_sl[kk] = ((2 * kk + lh) ^ a_x) << 4;
  int b_row0[TN];                                              // patch row of the pixel at tap (0, 0)
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int pix = wpx * WT_PX + j * 32 + l31;
    int oy, ox;
    pix_to_xy16<PW>(pix, oy, ox);                               // odd rows rotated: the 16 lanes of a ds_read_b128 group stay distinct mod 16
    b_row0[j] = oy * PW + ox;
  }

  // ---- DMA cursors.  Weights: tile ordinal ka, stage sa of it; gd = global stage number of the next DMA.
  Tile tc = decode(0);                                          // the tile being computed
  int ka = 0, sa = 0, gd = 0;
  set_woff(tc.co0);
  set_poff(tc);
  auto issue_A = [&]() __attribute__((always_inline)) {
    dma_A(gd % D, sa, ka < mytiles);
    ++gd;
    if (++sa == S) {
      sa = 0;
      ++ka;
      if (ka < mytiles) set_woff(decode(ka).co0);
    }
  };

  // ---- prologue: scale / bias and patch of the first tile, stages 0 .. D-2
  dma_SB(0, tc.co0);
  dma_P(0, 0);
#pragma unroll
  for (int s = 0; s < D - 1; ++s) issue_A();

  int gs = 0, gc = 0;                                           // global stage / chunk number on the compute side
  for (int k = 0; k < mytiles; ++k) {
    int cc = 0, r = 0;
    for (int s = 0; s < S; ++s, ++gs) {
      wait_vm_barrier<(D - 2) * NI_A>();                        // stage gs (and its patch) landed, in every wave
      if (r == 0) {                                             // before the weights: a patch is then never younger than the stage that needs it
        if (cc + 1 < nchunks) {
          dma_P((gc + 1) & 1, cc + 1);
        } else if (k + 1 < mytiles) {                           // first chunk of the next tile
          const Tile tn = decode(k + 1);
          set_poff(tn);
          dma_SB((k + 1) & 1, tn.co0);
          dma_P((gc + 1) & 1, 0);
        }
      }
      issue_A();
      const unsigned char* sA = smem + (gs % D) * A_BYTES;
      const unsigned char* sP = smem + P_OFF + (gc & 1) * P_BYTES;
#pragma unroll
      for (int sp = 0; sp < KS; ++sp) {
        int b_off[TN], b_x[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int prow = b_row0[j] + r * PW + sp;
          b_off[j] = prow * KB;
          b_x[j] = (prow / RPL) & (CH - 1);
        }
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
          u32x4 af[TM], bf[TN];
#ifdef DMA_DBG_NOLDS
#pragma unroll
          for (int i = 0; i < TM; ++i) af[i] = u32x4{(unsigned)(a_row[i] + s), 1u, 2u, 3u};
#pragma unroll
          for (int j = 0; j < TN; ++j) bf[j] = u32x4{(unsigned)(b_off[j] + kk), 1u, 2u, 3u};
#else
#pragma unroll
          for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const u32x4*>(sA + sp * CO_T * KB + a_row[i] + a_sl[kk]);
#pragma unroll
          for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const u32x4*>(sP + b_off[j] + (((2 * kk + lh) ^ b_x[j]) << 4));
#endif
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) MMA<T>::run(af[i], bf[j], acc[i][j]);
        }
      }
      if (++r == KS) { r = 0; ++cc; ++gc; }
    }

    // ---- epilogue of tile k, in its own LDS area (the ring and the patch buffers already fill for tile k+1)
    const unsigned char* sb = smem + SB_OFF + (k & 1) * 2 * SB_BYTES;
    unsigned char* sE = smem + E_OFF;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int co_l = wco * WT_CO + i * 32 + 8 * g + 4 * lh;
        const f32x4 sc = *reinterpret_cast<const f32x4*>(sb + co_l * 4);             // zeros beyond cout_pad
        const f32x4 bi = *reinterpret_cast<const f32x4*>(sb + SB_BYTES + co_l * 4);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int px_l = wpx * WT_PX + j * 32 + l31;
          const f32x4 xv = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
          const f32x4 yv = scale_bias_act4<T>(xv, sc, bi, a.act);
          const float v[4] = {yv[0], yv[1], yv[2], yv[3]};
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][j][4 * g + e] = 0.0f;
          store4(sE + px_l * ORS + co_l * (int)sizeof(TO), v, (TO*)nullptr);
        }
      }
    }
    wait_lds_barrier();
    constexpr int VO = 16 / (int)sizeof(TO);
    constexpr int OCPR = CO_T / VO;
    for (int q = tid; q < PX_T * OCPR; q += 64 * NWV) {
      const int px_l = q / OCPR, cq = q - px_l * OCPR;
      int oy, ox;
      pix_to_xy16<PW>(px_l, oy, ox);
      const int ho = tc.ty0 + oy, wo = tc.tx0 + ox, co = tc.co0 + cq * VO;
      if (ho < a.Ho && wo < a.Wo && co < a.Cout) {
        u32x4 v = *reinterpret_cast<const u32x4*>(sE + px_l * ORS + cq * 16);
        if (a.res) {
          const long ro = (long)tc.img * a.r_sn + (long)ho * a.r_sh + (long)wo * a.r_sw + co;
          v = add_chunk(v, *reinterpret_cast<const u32x4*>(a.res + ro * (long)sizeof(TO)), (TO*)nullptr, a.act_post);
        }
        const long yo = (long)tc.img * a.y_sn + (long)ho * a.y_sh + (long)wo * a.y_sw + co;
        *reinterpret_cast<u32x4*>(a.y + yo * (long)sizeof(TO)) = v;
      }
    }
    if (k + 1 < mytiles) tc = decode(k + 1);
    // the staging area is written again only after the S >= 3 stage barriers of the next tile
  }
  // drain the zero-fill DMAs of the tail before the LDS is released
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int CO_T, int KS, int KB, int D, int NWV>
static int launch_halo_dma(const ConvArgs& a, hipStream_t st) {
  using G = HaloDma<CO_T, KS, KB, D, NWV>;
  auto kern = conv_halo_dma_kernel<CO_T, KS, KB, D, NWV>;
  static bool attr_set = false;
  if (!attr_set && G::LDS > 64 * 1024) {
    GLS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
    attr_set = true;
  }
  ConvArgs b = a;
  b.n_co_tiles = (a.Cout + CO_T - 1) / CO_T;
  const int tiles_x = (a.Wo + G::TW - 1) / G::TW, tiles_y = (a.Ho + G::TH - 1) / G::TH;
  const long ntiles = (long)b.n_co_tiles * tiles_x * tiles_y * a.N;
  if (ntiles <= 0 || ntiles > 0x3fffffffL) GLS_FAIL(GLSDET_E_ARG, "conv2d(halo dma): %ld tiles out of range", ntiles);
  // one persistent workgroup per CU (the LDS footprint allows no second one); fewer tiles: one round
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) GLS_FAIL(GLSDET_E_HIP, "device query failed");
    n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  const long cap = (long)n_cu * (G::LDS <= 80 * 1024 ? 2 : 1);
  long nwg = ntiles < cap ? ntiles : cap;
  nwg = (nwg + 7) / 8 * 8;
  hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(64 * NWV), G::LDS, st, b, tiles_x, tiles_y, (int)ntiles);
  GLS_HIP(hipGetLastError());
  return 0;
}

// tile_hint 6: 64-row cout tiles, 7: 128-row cout tiles.  Returns 1 when the kernel does not apply.
int conv_halo_dma_try(const ConvArgs& a, int xdt, int ydt, int hint, OpRecord* op) {
  if (hint != 6 && hint != 7) return 1;
  if (a.stride != 1 || a.R != a.S || (a.R != 3 && a.R != 5 && a.R != 7) || a.pad != a.R / 2) return 1;
  if (xdt != GLSDET_F16 || ydt != GLSDET_F16) return 1;
  const int co_t = hint == 6 ? 64 : 128;
  if (co_t == 128 && (a.cout_pad <= 64 || a.R == 7)) return 1;       // 7x7 with 128 rows does not fit the LDS
  // channel chunk: 64 channels (128-byte rows) where the LDS holds ring + patches + epilogue, else 32
  const int kb = 64;
  if ((a.Cin * 2) % kb) return 1;
  char nm[96];
  snprintf(nm, sizeof nm, "conv_halo_dma<f16,%dx8x16,kb%d> %dx%d s1 cin%d cout%d", co_t, kb, a.R, a.S, a.Cin, a.Cout);
  op->name = nm;
  const int ks = a.R;
  op->launch = [a, co_t, ks](hipStream_t st) -> int {
    if (co_t == 64) {
      if (ks == 3) return launch_halo_dma<64, 3, 64, 2, 4>(a, st);          // 70 KB of LDS: two workgroups per CU, out of phase
      if (ks == 5) return launch_halo_dma<64, 5, 64, 3, 4>(a, st);
      return launch_halo_dma<64, 7, 64, 3, 4>(a, st);
    }
    if (ks == 3) return launch_halo_dma<128, 3, 64, 3, 8>(a, st);
    return launch_halo_dma<128, 5, 64, 2, 8>(a, st);
  };
  return 0;
}

}  // namespace glsdet
