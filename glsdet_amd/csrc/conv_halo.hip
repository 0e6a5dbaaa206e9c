// Stride-1 k x k convolution (k = 3, 5, 7) with the input PATCH staged once per channel
// chunk instead of once per tap.
//
// The generic implicit-GEMM kernel (conv.hip) re-stages the im2col tile for every filter
// tap: for a k x k filter every input pixel travels L2 -> registers -> LDS k*k times and the
// LDS write port, not the MFMA pipe, bounds the loop.  Here a workgroup owns a TH x TW block
// of output pixels of one image; for each 128-byte channel chunk it stages the
// (TH+k-1) x (TW+k-1) input patch ONCE (zero filled outside the image) and then walks the
// k*k taps: the B fragment of a tap is the same LDS image read at a constant offset
// (r*PW + s) rows further on.  Only the 128 x 128-byte weight tile of the tap is staged per
// step (double buffered, next tap's loads in flight under the MFMAs).
//   LDS writes per MFMA drop by ~(1 + 1/(k*k)) / 2, global->LDS input traffic by ~k*k.
// GEMM orientation, fragment layout, epilogue: identical to conv.hip.
#include "conv_common.h"

namespace glsdet {


template <int KS, int TH, int TW>
struct HaloGeom {
  static constexpr int PH = TH + KS - 1, PW = TW + KS - 1;
  static constexpr int NP = (PH * PW * 8 + 255) / 256;     // 16-B chunks per thread, one patch
};

template <typename T, typename TO, int CO_T, int WCO, int KS, int TH, int TW, bool CH = false>
__global__ __launch_bounds__(256) void conv_halo_kernel(const ConvArgs a, const int tiles_x, const int tiles_y) {
  constexpr int KB = 128, RS = KB + 16;
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int KE = KB / (int)sizeof(T);          // channels per chunk
  constexpr int PX_T = TH * TW;
  constexpr int PH = HaloGeom<KS, TH, TW>::PH, PW = HaloGeom<KS, TH, TW>::PW;
  constexpr int NP = HaloGeom<KS, TH, TW>::NP;
  constexpr int NA = (CO_T * 8 + 255) / 256;
  constexpr int WPX = 4 / WCO;
  constexpr int WT_CO = CO_T / WCO, WT_PX = PX_T / WPX;
  constexpr int TM = WT_CO / 32, TN = WT_PX / 32;
  constexpr int A_BYTES = CO_T * RS;
  constexpr int PATCH_OFF = 2 * A_BYTES;
  static_assert(PX_T == 128 && TW == 16 && TM >= 1 && TN >= 1 && (WCO != 4 || CO_T == 128), "tile shape");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // tile order: cout tile fastest, then x, y, image; XCD-contiguous like conv.hip
  int tile;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, local = bid >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
  }
  int rest = gls_div(tile, a.nco_mul, a.nco_sh);
  const int co0 = (tile - rest * a.n_co_tiles) * CO_T;
  const int r1 = gls_div(rest, a.tx_mul, a.tx_sh);
  const int tx0 = (rest - r1 * tiles_x) * TW;
  const int img = gls_div(r1, a.ty_mul, a.ty_sh);
  const int ty0 = (r1 - img * tiles_y) * TH;

  const int kc = tid & 7, row0 = tid >> 3;         // 8 chunks per 128-B row, 32 rows per pass
  const auto xrs = gls_make_rsrc(a.x_lo, a.x_bytes);
  const auto wrs = gls_make_rsrc(a.w, a.w_bytes);
  // WCO == 4 ("wave-private weights"): each wave owns 32 cout rows x all 128 pixels and stages exactly
  // the weight rows it multiplies itself -- no other wave reads them, so the per-tap workgroup barrier
  // disappears (LDS operations of one wave execute in order); the waves meet only when the input
  // patch is exchanged, once per 128-byte channel chunk.
  constexpr bool WP = WCO == 4;
  unsigned wp[NA];                                  // byte offsets; GLS_OOB rows read as zeros
  int arow[NA];                                     // LDS row of chunk i
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int row = WP ? 32 * wave + (lane >> 3) + 8 * i : row0 + i * 32;
    arow[i] = row;
    const bool ok = row < CO_T && (co0 + row) < a.cout_pad;
    wp[i] = ok ? (unsigned)(((co0 + row) * a.kpad + kc * VEC) * (int)sizeof(T)) : GLS_OOB;
  }
  // patch chunk -> input byte offset (without the channel-chunk offset); outside the image: GLS_OOB
  unsigned poff[NP];
  const int pad = KS / 2;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int q = tid + i * 256;
    const int pp = q >> 3;
    const int py = pp / PW, px = pp - py * PW;
    const int hi = ty0 - pad + py, wi = tx0 - pad + px;
    const bool ok = pp < PH * PW && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
    poff[i] = ok ? a.x_off + (unsigned)(((long)img * a.x_sn + (long)hi * a.x_sh + (long)wi * a.x_sw + kc * VEC) * (long)sizeof(T))
                 : GLS_OOB;
  }

  u32x4 ra[NA], rp[NP];
  const int nchunks = a.Cin / KE;
  const int ntaps = KS * KS;
  const int nsteps = nchunks * ntaps;

  auto load_a = [&](unsigned kbyte) __attribute__((always_inline)) {      // kbyte: wave-uniform byte offset of the tap's weights
#pragma unroll
    for (int i = 0; i < NA; ++i) ra[i] = gls_buf_load16(wrs, wp[i] + kbyte);
  };
  auto store_a = [&](int buf) __attribute__((always_inline)) {
    unsigned char* sa = smem + buf * A_BYTES + kc * 16;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      if (CO_T % 32 == 0 || arow[i] < CO_T) *reinterpret_cast<u32x4*>(sa + arow[i] * RS) = ra[i];
    }
  };
  auto load_patch = [&](int cc) __attribute__((always_inline)) {
    const unsigned coff = (unsigned)(cc * KE * (int)sizeof(T));
#pragma unroll
    for (int i = 0; i < NP; ++i) rp[i] = gls_buf_load16(xrs, poff[i] + coff);
  };
  auto store_patch = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int q = tid + i * 256;
      if (q < PH * PW * 8) *reinterpret_cast<u32x4*>(smem + PATCH_OFF + (q >> 3) * RS + (q & 7) * 16) = rp[i];
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
  const int a_off = (wco * WT_CO + l31) * RS + lh * 16;
  int b_off[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int pix = wpx * WT_PX + j * 32 + l31;
    int oy, ox;
    pix_to_xy16<PW>(pix, oy, ox);
    b_off[j] = PATCH_OFF + (oy * PW + ox) * RS + lh * 16;
  }

  load_patch(0);
  load_a(0u);
  store_patch();
  store_a(0);
  __syncthreads();

  // The K loop is written for a SHORT instruction stream per tap -- measured: with one wave per SIMD
  // every scalar instruction and, above all, every branch of the loop body delays the next MFMA issue
  // (the empty loop cost 27 % of the 7x7 layer).  Nest: channel chunk / tap; the weight offset of the
  // next tap is the current one plus Cin elements; the last tap of a chunk -- the only place where the
  // patch is exchanged -- is peeled, so the common body has no data-dependent branch at all.
  const unsigned tap_stride = (unsigned)(a.Cin * (int)sizeof(T));
  int cur = 0;
  auto mma_tap = [&](int tap_off) __attribute__((always_inline)) {
    const unsigned char* sA = smem + cur * A_BYTES;
#pragma unroll
    for (int kk = 0; kk < KB / 32; ++kk) {
      u32x4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const u32x4*>(sA + a_off + i * 32 * RS + kk * 32);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const u32x4*>(smem + b_off[j] + tap_off + kk * 32);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) MMA<T>::run(af[i], bf[j], acc[i][j]);
    }
  };
  for (int cc = 0; cc < nchunks; ++cc) {
    unsigned kbyte = (unsigned)(cc * KE * (int)sizeof(T));
    int tap_off = 0, ts = 0;
    for (int tap = 0; tap < ntaps - 1; ++tap) {       // every tap but the chunk's last
      kbyte += tap_stride;
      load_a(kbyte);
      mma_tap(tap_off);
      store_a(cur ^ 1);
      if (!WP) __syncthreads();
      cur ^= 1;
      ++ts;
      tap_off += (ts == KS) ? (PW - KS + 1) * RS : RS;
      ts = (ts == KS) ? 0 : ts;
    }
    if (cc + 1 < nchunks) {                           // last tap, another chunk follows
      load_a((unsigned)((cc + 1) * KE * (int)sizeof(T)));
      load_patch(cc + 1);
      mma_tap(tap_off);
      store_a(cur ^ 1);
      __syncthreads();                                // every wave is done with the old patch
      store_patch();
      __syncthreads();
      cur ^= 1;
    } else {
      mma_tap(tap_off);
    }
  }

  // the epilogue stages the output tile in the LDS bytes the weight tiles and the patch occupy: no wave may
  // start it while another still reads its last fragments (without the per-tap barrier of the shared-weight
  // form nothing else orders them)
  __syncthreads();
  // ---- epilogue (as conv.hip; tile-local pixel -> (oy, ox))
  const bool wide = sizeof(TO) == 2 && a.res != nullptr;      // fp32 staging: the residual is added before the one rounding
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int co_l = wco * WT_CO + i * 32 + 8 * g + 4 * lh;
      f32x4 sc = {0.f, 0.f, 0.f, 0.f}, bi = {0.f, 0.f, 0.f, 0.f};
      if (co0 + co_l < a.cout_pad) {
        sc = *reinterpret_cast<const f32x4*>(a.scale + co0 + co_l);
        bi = *reinterpret_cast<const f32x4*>(a.bias + co0 + co_l);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int px_l = wpx * WT_PX + j * 32 + l31;
        const f32x4 xv = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
        const f32x4 yv = scale_bias_act4<T>(xv, sc, bi, a.act);
        const float v[4] = {yv[0], yv[1], yv[2], yv[3]};
        stage4<TO, CO_T>(smem, px_l, co_l, v, wide);
      }
    }
  }
  __syncthreads();
  halo_store_and_chain<T, TO, CO_T, PW, CH>(smem, a, img, ty0, tx0, co0, tid);
}

// ------------------------------------------------------------------------------------------------
// The same kernel with the weight tiles staged by LDS-DMA into a ring of RING taps (tile hints 8 / 9).
// PMC of the register-staged form on a 3x3 128->128 layer: a workgroup lives 15 us for 1.9 us of MFMAs per
// wave -- each tap (256 MFMA cycles) waits ~2000 cycles because the next tap's weight tile is requested one tap
// ahead of an L2 round trip that takes several, and hipcc turns any deeper REGISTER ring into vmcnt(0) waits.
// A `buffer_load ... lds` has no register destination: the wait is ours to count.  The tile of tap g+RING-1 is
// requested when tap g starts (`s_waitcnt vmcnt(NI*(RING-2))` + raw `s_barrier`); rows are 128 bytes, unpadded
// (a DMA writes 1 KiB of consecutive LDS), with the XOR swizzle slot = chunk ^ ((row >> 1) & 7) applied on the
// source address and in the fragment read.  The patch keeps its register-staged, padded form.
template <int N>
__device__ __forceinline__ void halo_wait_vm_barrier() {
  // lgkmcnt(0): this wave's own LDS writes (the patch it just stored) are done before it signals the barrier
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}
__device__ __forceinline__ void halo_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // never __syncthreads(): it would drain the ring (vmcnt(0))
}

// KB = bytes of channels per chunk: 128, or 64 (half the LDS per workgroup: more workgroups per CU, a barrier every 4 MFMAs)
// STR = 2 (3x3 only): the stride-2 downsampling convs.  The patch is (2*8+1) x (2*16+1) input pixels; its columns are
// stored DE-INTERLEAVED (even columns first, then the odd ones), so that the 16 pixels of a fragment read, which are two
// input columns apart, are consecutive LDS rows again: tap (r, s) of output pixel (oy, ox) is slot
// (2*oy + r) * PW + ox + (s & 1 ? HALF : s >> 1) -- a per-lane base plus a per-tap constant, exactly as for stride 1.
// byte offset of the parked scale | bias (CO_T * 8 bytes): behind the staging area AND this launch's epilogue tile
// (wide = fp32 staging of a residual layer)
template <typename T, typename TO, int CO_T, int KS, int RING, int KB, int STR, int GEO = 0>
__host__ __device__ constexpr int halo_ring_sb_off(bool wide) {
  constexpr int PH = (TileGeo<GEO>::TH - 1) * STR + KS, PW0 = (TileGeo<GEO>::TW - 1) * STR + KS;
  constexpr int PW = (GEO != 0 && PW0 % 2 == 0) ? PW0 + 1 : PW0;       // odd patch pitch (conv_common.h GeoMap)
  constexpr int stage = RING * CO_T * KB + PH * PW * (KB + 16);
  const int epi = epi_bytes<TO>(CO_T, 128, wide);
  return ((stage > epi ? stage : epi) + 15) / 16 * 16;
}

template <typename T, typename TO, int CO_T, int KS, int RING, int KB = 128, int STR = 1, bool CH = false, bool GN = false, int GEO = 0, bool M16_ = false>
__device__ __forceinline__ void conv_halo_ring_body(const ConvArgs& a, const int tiles_x, const int tiles_y, const int tile) {
  constexpr bool M16 = M16_ && sizeof(T) == 2;     // v_mfma_f32_16x16x32_f16 and its fragment maps (conv_common.h)
  constexpr int TH = TileGeo<GEO>::TH, TW = TileGeo<GEO>::TW, WCO = 2;
  static_assert(GEO == 0 || (!CH && !GN), "other tile geometries: no chained / GroupNorm form");
  constexpr int RS = KB + 16;
  constexpr int CPRW = KB / 16;                    // 16-byte chunks per row
  constexpr int RPL = 256 / KB;                    // rows per 256 bytes: the swizzle is f(row) = (row / RPL) % CPRW
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int KE = KB / (int)sizeof(T);
  constexpr int PX_T = 128;                        // 128 MFMA lanes; dead lanes (GeoMap) multiply pixel 0 again and store nothing
  constexpr int PH = (TH - 1) * STR + KS, PW0 = (TW - 1) * STR + KS;
  constexpr int PW = (GEO != 0 && PW0 % 2 == 0) ? PW0 + 1 : PW0;       // odd patch pitch for the other geometries (one spare column)
  constexpr int HALF = (PW + 1) / 2;               // STR 2: even columns occupy slots [0, HALF), odd ones [HALF, PW)
  constexpr int PITCH = STR * PW;                  // slots between two output rows
  static_assert(STR == 1 || (STR == 2 && KS == 3), "stride 2 is built for 3x3");
  constexpr int NP = (PH * PW * CPRW + 255) / 256;
  constexpr int WPX = 4 / WCO;
  constexpr int WT_CO = CO_T / WCO, WT_PX = PX_T / WPX;
  constexpr int TM = WT_CO / 32, TN = WT_PX / 32;
  constexpr int A_BYTES = CO_T * KB;               // one tap's weight tile, unpadded rows
  constexpr int RPI = 64 / CPRW;                   // rows one DMA instruction fills
  constexpr int NI = CO_T / RPI / 4;               // DMA instructions per wave and tap
  static_assert(CO_T % (RPI * 4) == 0, "whole DMA pieces");
  constexpr int PATCH_OFF = RING * A_BYTES;
  static_assert(RING >= 3 && TM >= 1 && TN >= 1, "ring");

  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int rest = gls_div(tile, a.nco_mul, a.nco_sh);
  const int co0 = (tile - rest * a.n_co_tiles) * CO_T;
  const int r1 = gls_div(rest, a.tx_mul, a.tx_sh);
  const int tx0 = (rest - r1 * tiles_x) * TW;
  const int img = gls_div(r1, a.ty_mul, a.ty_sh);
  const int ty0 = (r1 - img * tiles_y) * TH;

  // folded-BN scale / bias of the cout tile: requested first, parked in LDS behind the staging / epilogue areas when the
  // first patch has landed anyway (conv.hip: the epilogue then starts without a dependent global round trip)
  unsigned char* sSB = smem + halo_ring_sb_off<T, TO, CO_T, KS, RING, KB, STR, GEO>(sizeof(TO) == 2 && a.res != nullptr);
  f32x4 sb_s = {0.f, 0.f, 0.f, 0.f}, sb_b = {0.f, 0.f, 0.f, 0.f};
  if (tid < CO_T / 4 && co0 + tid * 4 < a.cout_pad) {
    sb_s = *reinterpret_cast<const f32x4*>(a.scale + co0 + tid * 4);
    sb_b = *reinterpret_cast<const f32x4*>(a.bias + co0 + tid * 4);
  }

  const int kc = tid % CPRW;
  const auto xrs = gls_make_rsrc(a.x_lo, a.x_bytes);
  const auto wrs = gls_make_rsrc(a.w, a.w_bytes);
  // DMA source offsets: instruction q of this wave fills rows 8 * (wave + 4q) .. +7; lane -> (row, slot)
  unsigned wd[NI];
#pragma unroll
  for (int q = 0; q < NI; ++q) {
    const int row = RPI * (wave + 4 * q) + lane / CPRW;
    const int ch = (lane % CPRW) ^ w_swz<KB, M16>(row);
    const bool ok = (co0 + row) < a.cout_pad;
    wd[q] = ok ? (unsigned)(((co0 + row) * a.kpad + ch * VEC) * (int)sizeof(T)) : GLS_OOB;
  }
  unsigned poff[NP];
  constexpr int pad = KS / 2;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int q = tid + i * 256;
    const int pp = q / CPRW;                       // patch slot
    const int py = pp / PW, ps = pp - py * PW;
    const int px = STR == 1 ? ps : (ps < HALF ? 2 * ps : 2 * (ps - HALF) + 1);
    const int hi = ty0 * STR - pad + py, wi = tx0 * STR - pad + px;
    const bool ok = pp < PH * PW && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
    poff[i] = ok ? a.x_off + (unsigned)(((long)img * a.x_sn + (long)hi * a.x_sh + (long)wi * a.x_sw + kc * VEC) * (long)sizeof(T))
                 : GLS_OOB;
  }
  u32x4 rp[NP];
  const int nchunks = a.Cin / KE;
  constexpr int ntaps = KS * KS;
  const int nsteps = nchunks * ntaps;

  typedef __attribute__((address_space(3))) void* lds_ptr;
  // weight tile of global tap g (chunk g / ntaps, tap g % ntaps) -> ring slot g % RING; beyond the end: zeros
  // DMA cursor, all scalar: ring slot, tap inside the chunk, byte offset of the tap's weights.  Beyond the last tap the
  // offset becomes 2^31: valid rows then read out of range (zeros); rows that are out of range themselves wrap to a
  // small offset and fetch garbage into weight rows >= cout_pad, whose outputs are never stored -- no select, no branch.
  int dg = 0, dtap = 0, dslot = 0;
  unsigned dadd = 0;
  const unsigned tap_bytes = (unsigned)(a.Cin * (int)sizeof(T));
  const unsigned chunk_fix = (unsigned)(KE * (int)sizeof(T)) - (unsigned)ntaps * tap_bytes;   // (mod 2^32)
  auto dma_next = [&]() __attribute__((always_inline)) {
    unsigned char* dst = smem + dslot * A_BYTES + wave * 1024;
#pragma unroll
    for (int q = 0; q < NI; ++q)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_ptr)(dst + q * 4096), 16, (int)(wd[q] + dadd), 0, 0, 0);
    dslot = dslot + 1 == RING ? 0 : dslot + 1;
    dadd += tap_bytes;
    if (++dtap == ntaps) {
      dtap = 0;
      dadd += chunk_fix;
    }
    if (++dg >= nsteps) dadd = GLS_OOB;
  };
  auto load_patch = [&](int cc) __attribute__((always_inline)) {
    const unsigned coff = (unsigned)(cc * KE * (int)sizeof(T));
#pragma unroll
    for (int i = 0; i < NP; ++i) rp[i] = gls_buf_load16(xrs, poff[i] + coff);
  };
  auto store_patch = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int q = tid + i * 256;
      if (q < PH * PW * CPRW) *reinterpret_cast<u32x4*>(smem + PATCH_OFF + (q / CPRW) * RS + (q % CPRW) * 16) = rp[i];
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
  // ---- 16x16x32 form: TM16 x TN16 blocks of 16 x 16 per wave; lane -> (weight row / pixel column n16, k slice sl)
  constexpr int TM16 = M16 ? WT_CO / 16 : 1, TN16 = M16 ? WT_PX / 16 : 1;
  f32x4 acc16[TM16][TN16];
  int a16_sw[KB / 64 > 0 ? KB / 64 : 1], b16_off[TN16];
  const int n16 = lane & 15, sl = lane >> 4;
  const int a16_row = (wco * WT_CO + m16_wrow(n16)) * KB;
  // row of the staged output tile of this lane's pixel in column block j (-1: dead lane), and its (oy, ox)
  auto pix16 = [&](int j, int& oy, int& ox) __attribute__((always_inline)) -> int {
    if constexpr (GEO == 0) {                      // 8 x 16: block = one tile row; the store phase decodes rows with pix_to_xy16
      oy = wpx * (WT_PX / 16) + j;
      ox = m16_px16(n16);
      constexpr int ROT = (16 - (PITCH & 15)) & 15;
      return oy * 16 + ((oy & 1) ? ((ox - ROT) & 15) : ox);
    } else {
      const int p = GeoMap16Holder<TH, TW, PITCH, 128>::map.pix[(wpx * (WT_PX / 16) + j) * 16 + n16];
      oy = p == 0xffff ? 0 : p / TW;
      ox = p == 0xffff ? 0 : p - oy * TW;
      return p == 0xffff ? -1 : p;
    }
  };
  if constexpr (M16) {
#pragma unroll
    for (int i = 0; i < TM16; ++i)
#pragma unroll
      for (int j = 0; j < TN16; ++j) acc16[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < KB / 64; ++kk) a16_sw[kk] = ((m16_chunk(sl) + 4 * kk) ^ w_swz<KB, true>(m16_wrow(n16))) << 4;
#pragma unroll
    for (int j = 0; j < TN16; ++j) {
      int oy, ox;
      pix16(j, oy, ox);
      b16_off[j] = PATCH_OFF + (oy * PITCH + ox) * RS + m16_chunk(sl) * 16;
    }
  }
  const int a_row = (wco * WT_CO + l31) * KB;      // + i * 32 * KB: the swizzle term (row >> 1) & 7 depends on l31 only
  int a_sw[KB / 32];
#pragma unroll
  for (int kk = 0; kk < KB / 32; ++kk) a_sw[kk] = ((2 * kk + lh) ^ ((l31 / RPL) & (CPRW - 1))) << 4;
  int b_off[TN];
  int prow[TN];                                    // row of the staged output tile this lane's pixel goes to (-1: dead lane)
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int pix = wpx * WT_PX + j * 32 + l31;
    int oy, ox;
    if constexpr (GEO == 0) {
      pix_to_xy16<PITCH>(pix, oy, ox);
      prow[j] = pix;
    } else {
      const int p = GeoMapHolder<TH, TW, PITCH, 128>::map.pix[pix];       // (stride 2: a tile row is two patch rows further on)
      prow[j] = p == 0xffff ? -1 : p;
      oy = p == 0xffff ? 0 : p / TW;
      ox = p == 0xffff ? 0 : p - oy * TW;
    }
    b_off[j] = PATCH_OFF + (oy * PITCH + ox) * RS + lh * 16;
  }

  load_patch(0);
#pragma unroll
  for (int g = 0; g < RING - 1; ++g) dma_next();
  store_patch();                                   // (the compiler waits for the patch registers here)
  if (tid < CO_T / 4) {
    *reinterpret_cast<f32x4*>(sSB + tid * 16) = sb_s;
    *reinterpret_cast<f32x4*>(sSB + CO_T * 4 + tid * 16) = sb_b;
  }

  int g = 0;                                       // ring slot of the tap being multiplied
  auto mma_tap = [&](int tap_off) __attribute__((always_inline)) {
    if constexpr (M16) {
      const unsigned char* sA = smem + g * A_BYTES + a16_row;
      asm volatile("s_setprio 1" ::: "memory");
#pragma unroll
      for (int kk = 0; kk < KB / 64; ++kk) {
        u32x4 af[TM16];
#pragma unroll
        for (int i = 0; i < TM16; ++i) af[i] = *reinterpret_cast<const u32x4*>(sA + i * 16 * KB + a16_sw[kk]);
#pragma unroll
        for (int jh = 0; jh < TN16 / 2; ++jh) {     // two column blocks at a time (fragment registers)
          u32x4 bf[2];
#pragma unroll
          for (int j = 0; j < 2; ++j) bf[j] = *reinterpret_cast<const u32x4*>(smem + b16_off[2 * jh + j] + tap_off + kk * 64);
#pragma unroll
          for (int i = 0; i < TM16; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) mma16(af[i], bf[j], acc16[i][2 * jh + j]);
        }
      }
      asm volatile("s_setprio 0" ::: "memory");
      return;
    }
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
  };
  for (int cc = 0; cc < nchunks; ++cc) {
    int tap_off = 0, ts = 0;
    for (int tap = 0; tap < ntaps - 1; ++tap) {     // every tap but the chunk's last
      halo_wait_vm_barrier<NI * (RING - 2)>();      // tap g landed in every wave (at tap 0 also: the patch is visible); slot g-1 is free
      dma_next();
      mma_tap(tap_off);
      g = g + 1 == RING ? 0 : g + 1;
      ++ts;
      if (STR == 1) {
        tap_off += (ts == KS) ? (PW - KS + 1) * RS : RS;
      } else {                                      // slots of s = 0, 1, 2: 0, HALF, 1; then the next patch row
        tap_off += (ts == 1) ? HALF * RS : (ts == 2 ? -(HALF - 1) * RS : (PW - 1) * RS);
      }
      ts = (ts == KS) ? 0 : ts;
    }
    halo_wait_vm_barrier<NI * (RING - 2)>();
    dma_next();
    const bool more = cc + 1 < nchunks;             // last tap of the chunk; another chunk follows: exchange the patch
    if (more) load_patch(cc + 1);
    mma_tap(tap_off);                               // (ONE instance of the MFMA block for both cases: no accumulator copies)
    g = g + 1 == RING ? 0 : g + 1;
    if (more) {
      halo_lds_barrier();                           // every wave is done with the old patch
      store_patch();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the next tap's barrier publishes it
    }
  }
  halo_wait_vm_barrier<0>();                        // the zero fills of the tail have landed; all waves done reading

  const bool wide = sizeof(TO) == 2 && a.res != nullptr;
  if constexpr (M16) {
#pragma unroll
    for (int i = 0; i < TM16; ++i) {
      const int co_l = wco * WT_CO + i * 16 + m16_wrow(4 * sl);    // D rows 4 sl .. + 3 = the weight rows of lanes 4 sl .. + 3
      const f32x4 sc = *reinterpret_cast<const f32x4*>(sSB + co_l * 4), bi = *reinterpret_cast<const f32x4*>(sSB + CO_T * 4 + co_l * 4);
#pragma unroll
      for (int j = 0; j < TN16; ++j) {
        const f32x4 yv = scale_bias_act4<T>(acc16[i][j], sc, bi, a.act);
        const float v[4] = {yv[0], yv[1], yv[2], yv[3]};
        int oy, ox;
        const int pr = pix16(j, oy, ox);
        if (GEO == 0 || pr >= 0) stage4<TO, CO_T>(smem, pr, co_l, v, wide);
      }
    }
  } else
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int co_l = wco * WT_CO + i * 32 + 8 * gq + 4 * lh;
      const f32x4 sc = *reinterpret_cast<const f32x4*>(sSB + co_l * 4), bi = *reinterpret_cast<const f32x4*>(sSB + CO_T * 4 + co_l * 4);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const f32x4 xv = {acc[i][j][4 * gq], acc[i][j][4 * gq + 1], acc[i][j][4 * gq + 2], acc[i][j][4 * gq + 3]};
        const f32x4 yv = scale_bias_act4<T>(xv, sc, bi, a.act);
        const float v[4] = {yv[0], yv[1], yv[2], yv[3]};
        if (GEO == 0 || prow[j] >= 0) stage4<TO, CO_T>(smem, prow[j], co_l, v, wide);      // staged in row-major pixel order
      }
    }
  }
  __syncthreads();
  halo_store_and_chain<T, TO, CO_T, PITCH, CH, GN, TH, TW>(smem, a, img, ty0, tx0, co0, tid);
}

template <typename T, typename TO, int CO_T, int KS, int RING, int KB = 128, int STR = 1, bool CH = false, bool GN = false, int GEO = 0, bool M16 = false>
__global__ __launch_bounds__(256) void conv_halo_ring_kernel(const ConvArgs a, const int tiles_x, const int tiles_y) {
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, local = bid >> 3;       // XCD-aware tile order
  conv_halo_ring_body<T, TO, CO_T, KS, RING, KB, STR, CH, GN, GEO, M16>(a, tiles_x, tiles_y,
                                                               (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local);
}

// Several independent 3x3 convs of one shape class (the quadrant convs of Patch_Conv / Patch_Conv_NonLocal and their
// l / r / t / b convs: each alone is a fraction of a round of workgroups) as ONE launch of the ring kernel; the
// argument blocks travel in the kernarg segment as for conv_igemm_multi_kernel (conv.hip), a workgroup finds its problem
// from the prefix of tile counts.
template <typename T, typename TO, int CO_T, int KS, int RING, int KB, int STR, int GEO = 0, bool M16 = false>
__global__ __launch_bounds__(256) void conv_halo_ring_multi_kernel(const HaloArgsN m) {
  int g = 0;
#pragma unroll
  for (int i = 1; i < GLS_MULTI; ++i)
    if (i < m.n && (int)blockIdx.x >= m.start[i]) g = i;
  conv_halo_ring_body<T, TO, CO_T, KS, RING, KB, STR, false, false, GEO, M16>(m.p[g], m.tx[g], m.ty[g], (int)blockIdx.x - m.start[g]);
}

// ------------------------------------------------------------------------------------------------
// 8-WAVE form of the ring kernel (tile hints 12 / 13; the round-1 verdict's "128 x 256 tile, two waves per SIMD, half the
// weight staging per MFMA"): one workgroup of 512 threads owns 128 cout rows x an 8 x 32 pixel tile; the waves are laid
// out 2 (cout) x 4 (pairs of pixel rows), each with the 64 x 64 register tile of the 128-row 4-wave form, so a tap's
// weight tile (one DMA ring slot) serves twice the pixels.  A 32-lane MFMA block is one full tile row of 32 pixels: no
// row rotation needed.  No residual / chained / GroupNorm forms.  Measured in DESIGN.md section 3.
// GEO8: 0 = 8 x 32 pixels, 1 = 10 x 24 (240 of the 256 lanes live), 2 = 6 x 42 (252): a 100 x 168 map takes 78 / 70 / 68 tiles
template <int GEO8> struct TileGeo8;
template <> struct TileGeo8<0> { static constexpr int TH = 8, TW = 32; };
template <> struct TileGeo8<1> { static constexpr int TH = 10, TW = 24; };
template <> struct TileGeo8<2> { static constexpr int TH = 6, TW = 42; };

// (amdgpu_waves_per_eu: the 64-byte-chunk forms fit two workgroups per CU by LDS, i.e. 4 waves per SIMD -- 128 registers; the
// 16x16x32 form holds 32 fragment registers per k step instead of 16 and the allocator otherwise settles at 130-150)
template <typename T, int KS, int RING, int KB, int GEO8 = 0, bool M16_ = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(KB == 64 ? 4 : 2))) void conv_halo_ring8_kernel(const ConvArgs a, const int tiles_x, const int tiles_y) {
  constexpr bool M16 = M16_ && sizeof(T) == 2;     // v_mfma_f32_16x16x32_f16 and its fragment maps (conv_common.h)
  constexpr int TH = TileGeo8<GEO8>::TH, TW = TileGeo8<GEO8>::TW, CO_T = 128, NWV = 8;
  constexpr int LIVE = TH * TW;                    // of 256 MFMA lanes
  constexpr int RS = KB + 16, CPRW = KB / 16, RPL = 256 / KB;
  constexpr int VEC = 16 / (int)sizeof(T), KE = KB / (int)sizeof(T);
  constexpr int PH = TH - 1 + KS, PW0 = TW - 1 + KS;
  constexpr int PW = (GEO8 != 0 && PW0 % 2 == 0) ? PW0 + 1 : PW0;      // odd patch pitch for the other geometries (GeoMap)
  constexpr int NP = (PH * PW * CPRW + 511) / 512;
  constexpr int TM = 2, TN = 2;
  constexpr int A_BYTES = CO_T * KB;
  constexpr int RPI = 64 / CPRW;                   // rows one DMA instruction fills
  constexpr int NI = CO_T / RPI / NWV;             // DMA instructions per wave and tap
  static_assert(CO_T % (RPI * NWV) == 0 && RING >= 3, "ring geometry");
  constexpr int PATCH_OFF = RING * A_BYTES;
  constexpr int ORS = CO_T * (int)sizeof(T) + 16;
  constexpr int STAGE = RING * A_BYTES + PH * PW * RS, EPI = 256 * ORS;
  constexpr int SB_OFF = ((STAGE > EPI ? STAGE : EPI) + 15) / 16 * 16;

  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int tile;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, local = bid >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
  }
  int rest = gls_div(tile, a.nco_mul, a.nco_sh);
  const int co0 = (tile - rest * a.n_co_tiles) * CO_T;
  const int r1 = gls_div(rest, a.tx_mul, a.tx_sh);
  const int tx0 = (rest - r1 * tiles_x) * TW;
  const int img = gls_div(r1, a.ty_mul, a.ty_sh);
  const int ty0 = (r1 - img * tiles_y) * TH;

  unsigned char* sSB = smem + SB_OFF;
  f32x4 sbv = {0.f, 0.f, 0.f, 0.f};
  if (tid < CO_T / 2 && co0 + (tid % (CO_T / 4)) * 4 < a.cout_pad)
    sbv = *reinterpret_cast<const f32x4*>((tid < CO_T / 4 ? a.scale : a.bias) + co0 + (tid % (CO_T / 4)) * 4);

  const int kc = tid % CPRW;
  const auto xrs = gls_make_rsrc(a.x_lo, a.x_bytes);
  const auto wrs = gls_make_rsrc(a.w, a.w_bytes);
  unsigned wd[NI];
#pragma unroll
  for (int q = 0; q < NI; ++q) {
    const int row = RPI * (wave + NWV * q) + lane / CPRW;
    const int ch = (lane % CPRW) ^ w_swz<KB, M16>(row);
    wd[q] = (co0 + row) < a.cout_pad ? (unsigned)(((co0 + row) * a.kpad + ch * VEC) * (int)sizeof(T)) : GLS_OOB;
  }
  unsigned poff[NP];
  constexpr int pad = KS / 2;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int q = tid + i * 512;
    const int pp = q / CPRW;
    const int py = pp / PW, px = pp - py * PW;
    const int hi = ty0 - pad + py, wi = tx0 - pad + px;
    const bool ok = pp < PH * PW && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
    poff[i] = ok ? a.x_off + (unsigned)(((long)img * a.x_sn + (long)hi * a.x_sh + (long)wi * a.x_sw + kc * VEC) * (long)sizeof(T)) : GLS_OOB;
  }
  u32x4 rp[NP];
  const int nchunks = a.Cin / KE;
  constexpr int ntaps = KS * KS;
  const int nsteps = nchunks * ntaps;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  int dg = 0, dtap = 0, dslot = 0;
  unsigned dadd = 0;
  const unsigned tap_bytes = (unsigned)(a.Cin * (int)sizeof(T));
  const unsigned chunk_fix = (unsigned)(KE * (int)sizeof(T)) - (unsigned)ntaps * tap_bytes;
  auto dma_next = [&]() __attribute__((always_inline)) {
    unsigned char* dst = smem + dslot * A_BYTES + wave * 1024;
#pragma unroll
    for (int q = 0; q < NI; ++q)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_ptr)(dst + q * NWV * 1024), 16, (int)(wd[q] + dadd), 0, 0, 0);
    dslot = dslot + 1 == RING ? 0 : dslot + 1;
    dadd += tap_bytes;
    if (++dtap == ntaps) {
      dtap = 0;
      dadd += chunk_fix;
    }
    if (++dg >= nsteps) dadd = GLS_OOB;
  };
  auto load_patch = [&](int cc) __attribute__((always_inline)) {
    const unsigned coff = (unsigned)(cc * KE * (int)sizeof(T));
#pragma unroll
    for (int i = 0; i < NP; ++i) rp[i] = gls_buf_load16(xrs, poff[i] + coff);
  };
  auto store_patch = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int q = tid + i * 512;
      if (q < PH * PW * CPRW) *reinterpret_cast<u32x4*>(smem + PATCH_OFF + (q / CPRW) * RS + (q % CPRW) * 16) = rp[i];
    }
  };
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
  const int wco = wave % 2, wpx = wave / 2;         // 2 cout halves x 4 pairs of pixel rows
  const int l31 = lane & 31, lh = lane >> 5;
  // ---- 16x16x32 form: 4 x 4 blocks of 16 x 16 per wave; lane -> (row / pixel column n16, k slice sl)
  constexpr int NB16 = M16 ? 4 : 1;
  f32x4 acc16[NB16][NB16];
  int a16_sw[KB / 64 > 0 ? KB / 64 : 1], b16_off[NB16];
  const int n16 = lane & 15, sl = lane >> 4;
  const int a16_row = (wco * 64 + m16_wrow(n16)) * KB;
  auto pix16 = [&](int j) __attribute__((always_inline)) -> int {     // row-major pixel of this lane in column block j (0xffff: dead)
    if constexpr (GEO8 == 0) return (wpx * 2 + (j >> 1)) * 32 + (j & 1) * 16 + m16_px16(n16);
    else return GeoMap16Holder<TH, TW, PW, 256>::map.pix[(wpx * 4 + j) * 16 + n16];
  };
  if constexpr (M16) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc16[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < KB / 64; ++kk) a16_sw[kk] = ((m16_chunk(sl) + 4 * kk) ^ w_swz<KB, true>(m16_wrow(n16))) << 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {                  // column block wpx * 4 + j of the tile's 16
      const int p = pix16(j);
      const int oy = p == 0xffff ? 0 : p / TW, ox = p == 0xffff ? 0 : p - oy * TW;
      b16_off[j] = PATCH_OFF + (oy * PW + ox) * RS + m16_chunk(sl) * 16;
    }
  }
  const int a_row = (wco * 64 + l31) * KB;
  int a_sw[KB / 32];
#pragma unroll
  for (int kk = 0; kk < KB / 32; ++kk) a_sw[kk] = ((2 * kk + lh) ^ ((l31 / RPL) & (CPRW - 1))) << 4;
  int b_off[TN];
  int prow[TN];                                    // row of the staged output tile (row-major pixel index; -1: dead lane)
#pragma unroll
  for (int j = 0; j < TN; ++j) {                  // lanes of pixel block wpx * 2 + j (8 x 32: tile row wpx * 2 + j, column l31)
    const int pix = (wpx * 2 + j) * 32 + l31;
    int p = pix;
    if constexpr (GEO8 != 0) p = GeoMapHolder<TH, TW, PW, 256>::map.pix[pix];
    prow[j] = p == 0xffff ? -1 : p;
    const int oy = p == 0xffff ? 0 : p / TW, ox = p == 0xffff ? 0 : p - oy * TW;
    b_off[j] = PATCH_OFF + (oy * PW + ox) * RS + lh * 16;
  }

  load_patch(0);
#pragma unroll
  for (int g = 0; g < RING - 1; ++g) dma_next();
  store_patch();
  if (tid < CO_T / 2) *reinterpret_cast<f32x4*>(sSB + tid * 16) = sbv;

  int g = 0;
  auto mma_tap = [&](int tap_off) __attribute__((always_inline)) {
    if constexpr (M16) {
      const unsigned char* sA = smem + g * A_BYTES + a16_row;
      asm volatile("s_setprio 1" ::: "memory");
#pragma unroll
      for (int kk = 0; kk < KB / 64; ++kk) {
        u32x4 af[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const u32x4*>(sA + i * 16 * KB + a16_sw[kk]);
#pragma unroll
        for (int jh = 0; jh < 2; ++jh) {            // two column blocks at a time: 24 fragment registers live, not 32
          u32x4 bf[2];
#pragma unroll
          for (int j = 0; j < 2; ++j) bf[j] = *reinterpret_cast<const u32x4*>(smem + b16_off[2 * jh + j] + tap_off + kk * 64);
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) mma16(af[i], bf[j], acc16[i][2 * jh + j]);
        }
      }
      asm volatile("s_setprio 0" ::: "memory");
      return;
    }
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
  };
  for (int cc = 0; cc < nchunks; ++cc) {
    int tap_off = 0, ts = 0;
    for (int tap = 0; tap < ntaps - 1; ++tap) {
      halo_wait_vm_barrier<NI * (RING - 2)>();
      dma_next();
      mma_tap(tap_off);
      g = g + 1 == RING ? 0 : g + 1;
      ++ts;
      tap_off += (ts == KS) ? (PW - KS + 1) * RS : RS;
      ts = (ts == KS) ? 0 : ts;
    }
    halo_wait_vm_barrier<NI * (RING - 2)>();
    dma_next();
    const bool more = cc + 1 < nchunks;
    if (more) load_patch(cc + 1);
    mma_tap(tap_off);
    g = g + 1 == RING ? 0 : g + 1;
    if (more) {
      halo_lds_barrier();
      store_patch();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  halo_wait_vm_barrier<0>();

  if constexpr (M16) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int co_l = wco * 64 + i * 16 + m16_wrow(4 * sl);       // D rows 4 sl .. 4 sl + 3 are the weight rows of lanes 4 sl .. + 3
      const f32x4 sc = *reinterpret_cast<const f32x4*>(sSB + co_l * 4), bi = *reinterpret_cast<const f32x4*>(sSB + CO_T * 4 + co_l * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 yv = scale_bias_act4<T>(acc16[i][j], sc, bi, a.act);
        const float v[4] = {yv[0], yv[1], yv[2], yv[3]};
        const int p = pix16(j);
        if (GEO8 == 0 || p != 0xffff) store4(smem + p * ORS + co_l * (int)sizeof(T), v, (T*)nullptr);
      }
    }
  } else
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int co_l = wco * 64 + i * 32 + 8 * gq + 4 * lh;
      const f32x4 sc = *reinterpret_cast<const f32x4*>(sSB + co_l * 4), bi = *reinterpret_cast<const f32x4*>(sSB + CO_T * 4 + co_l * 4);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const f32x4 xv = {acc[i][j][4 * gq], acc[i][j][4 * gq + 1], acc[i][j][4 * gq + 2], acc[i][j][4 * gq + 3]};
        const f32x4 yv = scale_bias_act4<T>(xv, sc, bi, a.act);
        const float v[4] = {yv[0], yv[1], yv[2], yv[3]};
        if (GEO8 == 0 || prow[j] >= 0) store4(smem + prow[j] * ORS + co_l * (int)sizeof(T), v, (T*)nullptr);
      }
    }
  }
  __syncthreads();
  constexpr int OCPR = CO_T / VEC;
  for (int q = tid; q < LIVE * OCPR; q += 512) {
    const int px_l = q / OCPR, cq = q - px_l * OCPR;
    const int oy = px_l / TW, ox = px_l - oy * TW;
    const int ho = ty0 + oy, wo = tx0 + ox, co = co0 + cq * VEC;
    if (ho < a.Ho && wo < a.Wo && co < a.Cout) {
      const long yo = (long)img * a.y_sn + (long)ho * a.y_sh + (long)wo * a.y_sw + co;
      *reinterpret_cast<u32x4*>(a.y + yo * (long)sizeof(T)) = *reinterpret_cast<const u32x4*>(smem + px_l * ORS + cq * 16);
    }
  }
}

template <typename T, int KS, int RING, int KB, int GEO8 = 0, bool M16 = false>
static int launch_halo_ring8(const ConvArgs& a, hipStream_t st) {
  constexpr int TH = TileGeo8<GEO8>::TH, TW = TileGeo8<GEO8>::TW;
  constexpr int PH = TH - 1 + KS, PW0 = TW - 1 + KS, PW = (GEO8 != 0 && PW0 % 2 == 0) ? PW0 + 1 : PW0, ORS = 128 * (int)sizeof(T) + 16;
  constexpr int stage = RING * 128 * KB + PH * PW * (KB + 16), epi = 256 * ORS;
  constexpr int lds = ((stage > epi ? stage : epi) + 15) / 16 * 16 + 128 * 8;
  static_assert(lds <= 160 * 1024, "LDS");
  auto kern = conv_halo_ring8_kernel<T, KS, RING, KB, GEO8, M16>;
  static bool attr_set = false;
  if (!attr_set) {
    GLS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_set = true;
  }
  ConvArgs b = a;
  b.n_co_tiles = (a.Cout + 127) / 128;
  const int tiles_x = (a.Wo + TW - 1) / TW, tiles_y = (a.Ho + TH - 1) / TH;
  gls_fastdiv(b.n_co_tiles, &b.nco_mul, &b.nco_sh);
  gls_fastdiv(tiles_x, &b.tx_mul, &b.tx_sh);
  gls_fastdiv(tiles_y, &b.ty_mul, &b.ty_sh);
  const long grid = (long)b.n_co_tiles * tiles_x * tiles_y * a.N;
  if (grid <= 0 || grid > 0x7fffffffL) GLS_FAIL(GLSDET_E_ARG, "conv2d(halo ring8): grid %ld out of range", grid);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds, st, b, tiles_x, tiles_y);
  GLS_HIP(hipGetLastError());
  return 0;
}

// fp16: the 16x16x32 MFMA form unless GLSDET_NO_M16 is set.  Read at every launch (an op of a recorded plan keeps what it
// was recorded with only through the environment of its replays: set it for the life of the process, as the tests that
// compare kernel families bit for bit do -- the two MFMA shapes round differently in the last place).
static bool use_m16() { return getenv("GLSDET_NO_M16") == nullptr; }
template <typename T, int KS, int RING, int KB, int GEO8 = 0>
static int ring8_any(const ConvArgs& a, hipStream_t st) {
  if constexpr (sizeof(T) == 2) {
    if (use_m16()) return launch_halo_ring8<T, KS, RING, KB, GEO8, true>(a, st);
  }
  return launch_halo_ring8<T, KS, RING, KB, GEO8, false>(a, st);
}

template <typename T>
static int halo_ring8_dispatch(const ConvArgs& a, bool k64, int geo, hipStream_t st) {
  if (geo == 0) {
    switch (a.R) {
      case 3: return k64 ? ring8_any<T, 3, 4, 64>(a, st) : ring8_any<T, 3, 3, 128>(a, st);
      case 5: return k64 ? ring8_any<T, 5, 4, 64>(a, st) : ring8_any<T, 5, 3, 128>(a, st);
      case 7: return k64 ? ring8_any<T, 7, 4, 64>(a, st) : ring8_any<T, 7, 3, 128>(a, st);
    }
  } else if (k64) {                // the other tile geometries: 64-byte channel chunks only (the form the tuner picks at 100 x 168)
    switch (a.R * 4 + geo) {
      case 3 * 4 + 1: return ring8_any<T, 3, 4, 64, 1>(a, st);
      case 3 * 4 + 2: return ring8_any<T, 3, 4, 64, 2>(a, st);
      case 5 * 4 + 1: return ring8_any<T, 5, 4, 64, 1>(a, st);
      case 5 * 4 + 2: return ring8_any<T, 5, 4, 64, 2>(a, st);
      case 7 * 4 + 1: return ring8_any<T, 7, 4, 64, 1>(a, st);
      case 7 * 4 + 2: return ring8_any<T, 7, 4, 64, 2>(a, st);
    }
  }
  GLS_FAIL(GLSDET_E_ARG, "conv2d(halo ring8): unsupported kernel size %d / geometry %d", a.R, geo);
}

// (the chained 1x1 exists for 3x3 stride 1 only: a CSP Bottleneck's conv2 -> the next Bottleneck's conv1)
// and so do the GroupNorm partials (glsdet_conv2d_gnstats: the 3x3 tower convs of GFLHead / MPHead)
template <typename T, typename TO, int CO_T, int KS, int RING, int KB = 128, int STR = 1, bool CH = false, bool GN = false, int GEO = 0>
static int launch_halo_ring(const ConvArgs& a, hipStream_t st) {
  if constexpr (GEO != 0) {
    if (a.w2 || a.gn_part) GLS_FAIL(GLSDET_E_ARG, "conv2d: chained / GroupNorm forms exist for 8 x 16 tiles only");
  }
  if constexpr (!CH && !GN && KS == 3 && STR == 1 && GEO == 0) {
    if (a.w2) return launch_halo_ring<T, TO, CO_T, KS, RING, KB, STR, true, false>(a, st);
    if constexpr (sizeof(T) == sizeof(TO)) {
      if (a.gn_part) return launch_halo_ring<T, TO, CO_T, KS, RING, KB, STR, false, true>(a, st);
    }
  }
  if (a.w2 && !CH) GLS_FAIL(GLSDET_E_ARG, "conv2d_chain: this halo kernel has no chained form");
  if (a.gn_part && !GN) GLS_FAIL(GLSDET_E_ARG, "conv2d_gnstats: this halo kernel has no statistics form");
  constexpr int ldsw = halo_ring_sb_off<T, TO, CO_T, KS, RING, KB, STR, GEO>(true) + CO_T * 8;
  int lds = halo_ring_sb_off<T, TO, CO_T, KS, RING, KB, STR, GEO>(sizeof(TO) == 2 && a.res != nullptr) + CO_T * 8;
  if (a.w2 && chain_lds_bytes<T>(CO_T, 128, a) > lds) lds = chain_lds_bytes<T>(CO_T, 128, a);
  auto kern = conv_halo_ring_kernel<T, TO, CO_T, KS, RING, KB, STR, CH, GN, GEO, false>;
  if constexpr (sizeof(T) == 2) {
    if (use_m16()) kern = conv_halo_ring_kernel<T, TO, CO_T, KS, RING, KB, STR, CH, GN, GEO, true>;
  }
  static int attr_lds = 64 * 1024;
  const int want_attr = lds > ldsw ? lds : ldsw;
  if (want_attr > attr_lds) {                      // (both MFMA forms: the switch is read per launch)
    GLS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo_ring_kernel<T, TO, CO_T, KS, RING, KB, STR, CH, GN, GEO, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, want_attr));
    if constexpr (sizeof(T) == 2)
      GLS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo_ring_kernel<T, TO, CO_T, KS, RING, KB, STR, CH, GN, GEO, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, want_attr));
    attr_lds = want_attr;
  }
  ConvArgs b = a;
  b.n_co_tiles = (a.Cout + CO_T - 1) / CO_T;
  const int tiles_x = (a.Wo + TileGeo<GEO>::TW - 1) / TileGeo<GEO>::TW, tiles_y = (a.Ho + TileGeo<GEO>::TH - 1) / TileGeo<GEO>::TH;
  gls_fastdiv(b.n_co_tiles, &b.nco_mul, &b.nco_sh);
  gls_fastdiv(tiles_x, &b.tx_mul, &b.tx_sh);
  gls_fastdiv(tiles_y, &b.ty_mul, &b.ty_sh);
  const long grid = (long)b.n_co_tiles * tiles_x * tiles_y * a.N;
  if (grid <= 0 || grid > 0x7fffffffL) GLS_FAIL(GLSDET_E_ARG, "conv2d(halo ring): grid %ld out of range", grid);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, st, b, tiles_x, tiles_y);
  GLS_HIP(hipGetLastError());
  return 0;
}

template <typename T, typename TO, int CO_T, int RING, int KB, int STR, int GEO = 0>
static int launch_halo_ring_multi(const ConvArgsN& m0, hipStream_t st) {
  constexpr int KS = 3;
  bool any_res = false;
  for (int i = 0; i < m0.n; ++i) any_res = any_res || m0.p[i].res != nullptr;
  constexpr int ldsw = halo_ring_sb_off<T, TO, CO_T, KS, RING, KB, STR, GEO>(true) + CO_T * 8;
  const int lds = halo_ring_sb_off<T, TO, CO_T, KS, RING, KB, STR, GEO>(sizeof(TO) == 2 && any_res) + CO_T * 8;
  auto kern = conv_halo_ring_multi_kernel<T, TO, CO_T, KS, RING, KB, STR, GEO, false>;
  if constexpr (sizeof(T) == 2) {
    if (use_m16()) kern = conv_halo_ring_multi_kernel<T, TO, CO_T, KS, RING, KB, STR, GEO, true>;
  }
  static bool attr_set = false;
  if (!attr_set && ldsw > 64 * 1024) {
    GLS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo_ring_multi_kernel<T, TO, CO_T, KS, RING, KB, STR, GEO, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, ldsw));
    if constexpr (sizeof(T) == 2)
      GLS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo_ring_multi_kernel<T, TO, CO_T, KS, RING, KB, STR, GEO, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, ldsw));
    attr_set = true;
  }
  HaloArgsN m = {};
  m.n = m0.n;
  long grid = 0;
  for (int i = 0; i < m.n; ++i) {
    ConvArgs& b = m.p[i];
    b = m0.p[i];
    b.n_co_tiles = (b.Cout + CO_T - 1) / CO_T;
    m.tx[i] = (b.Wo + TileGeo<GEO>::TW - 1) / TileGeo<GEO>::TW;
    m.ty[i] = (b.Ho + TileGeo<GEO>::TH - 1) / TileGeo<GEO>::TH;
    gls_fastdiv(b.n_co_tiles, &b.nco_mul, &b.nco_sh);
    gls_fastdiv(m.tx[i], &b.tx_mul, &b.tx_sh);
    gls_fastdiv(m.ty[i], &b.ty_mul, &b.ty_sh);
    m.start[i] = (int)grid;
    grid += (long)b.n_co_tiles * m.tx[i] * m.ty[i] * b.N;
  }
  for (int i = m.n; i <= GLS_MULTI; ++i) m.start[i] = (int)grid;
  if (grid <= 0 || grid > 0x7fffffffL) GLS_FAIL(GLSDET_E_ARG, "conv2d_multi(halo ring): grid %ld out of range", grid);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, st, m);
  GLS_HIP(hipGetLastError());
  return 0;
}

// The grouped form of the ring kernel for glsdet_conv2d_multi: hints 8 / 9 (128-byte channel chunks, 64- / 128-row cout
// tiles; stride 1) and 10 / 11 (64-byte chunks; stride 1 and 2), 3x3 only, no chained / GroupNorm forms.
// Returns 1 when it does not apply, 0 when `op` (name + launch) was filled in.
int conv_halo_multi_try(const ConvArgsN& m, int xdt, int ydt, int hint_in, OpRecord* op) {
  const int geo = (hint_in >= 0x100 && hint_in < 0x300) ? (hint_in >> 8) : 0;       // tile geometry: 8 x 16 / 10 x 12 / 6 x 21
  const int hint = geo ? (hint_in & 0xff) : hint_in;
  if (hint < 8 || hint > 11 || xdt != ydt) return 1;
  if (geo && m.p[0].stride != 1 && !(hint_in & 2)) return 1;         // stride 2: 64-byte chunks (hints 10 / 11)
  const int es = dtype_size(xdt);
  const bool k64 = hint >= 10;
  const ConvArgs& a0 = m.p[0];
  for (int i = 0; i < m.n; ++i) {
    const ConvArgs& a = m.p[i];
    if (a.w2 || a.gn_part || a.R != 3 || a.S != 3 || a.pad != 1 || (a.stride != 1 && a.stride != 2)) return 1;
    if (a.stride != a0.stride || a.Cin != a0.Cin || a.Cout != a0.Cout) return 1;
    if ((a.Cin * es) % (k64 ? 64 : 128)) return 1;
  }
  if (a0.stride == 2 && !k64) return 1;                 // the de-interleaved-patch form exists for 64-byte chunks
  if ((hint == 9 || hint == 11) && a0.cout_pad <= 64) return 1;
  const int co_t = (hint == 9 || hint == 11) ? 128 : 64;
  const int str = a0.stride;
  char nm[112];
  int gth, gtw;
  tile_geo_dims(geo, &gth, &gtw);
  snprintf(nm, sizeof nm, "conv_halo_ring%s%s_multi[%d]<%s,%dx%dx%d> 3x3 s%d cin%d cout%d", k64 ? "_k64" : "", str == 2 ? "_s2" : "", m.n,
           xdt ? "f32" : "f16", co_t, gth, gtw, str, a0.Cin, a0.Cout);
  op->name = nm;
  op->launch = [m, co_t, xdt, k64, str, geo](hipStream_t st) -> int {
#define GLS_HMG(T_, G_)                                                                                             \
    if (geo == G_ && str == 2)                                                                                      \
      return co_t == 128 ? launch_halo_ring_multi<T_, T_, 128, 4, 64, 2, G_>(m, st) : launch_halo_ring_multi<T_, T_, 64, 4, 64, 2, G_>(m, st); \
    if (geo == G_) {                                                                                                \
      if (k64) return co_t == 128 ? launch_halo_ring_multi<T_, T_, 128, 4, 64, 1, G_>(m, st) : launch_halo_ring_multi<T_, T_, 64, 4, 64, 1, G_>(m, st); \
      return co_t == 128 ? launch_halo_ring_multi<T_, T_, 128, 3, 128, 1, G_>(m, st) : launch_halo_ring_multi<T_, T_, 64, 3, 128, 1, G_>(m, st);        \
    }
#define GLS_HM(T_)                                                                                                  \
    GLS_HMG(T_, 1)                                                                                                  \
    GLS_HMG(T_, 2)                                                                                                  \
    if (str == 2) return co_t == 128 ? launch_halo_ring_multi<T_, T_, 128, 4, 64, 2>(m, st) : launch_halo_ring_multi<T_, T_, 64, 4, 64, 2>(m, st); \
    if (k64) return co_t == 128 ? launch_halo_ring_multi<T_, T_, 128, 4, 64, 1>(m, st) : launch_halo_ring_multi<T_, T_, 64, 4, 64, 1>(m, st);      \
    return co_t == 128 ? launch_halo_ring_multi<T_, T_, 128, 3, 128, 1>(m, st) : launch_halo_ring_multi<T_, T_, 64, 3, 128, 1>(m, st);
    if (xdt == GLSDET_F16) { GLS_HM(f16) }
    GLS_HM(float)
#undef GLS_HM
#undef GLS_HMG
  };
  return 0;
}

// the other tile geometries (GEO 1 = 10 x 12, 2 = 6 x 21): 128-byte chunks for 3x3 only, 64-byte chunks for 3x3 / 5x5 / 7x7
template <typename T, typename TO, int CO_T, int GEO>
static int halo_ring_geo(const ConvArgs& a, bool k64, hipStream_t st) {
  if (!k64) {
    if (a.R == 3) return launch_halo_ring<T, TO, CO_T, 3, 3, 128, 1, false, false, GEO>(a, st);
    GLS_FAIL(GLSDET_E_ARG, "conv2d(halo ring): tile geometry %d with 128-byte chunks exists for 3x3 only", GEO);
  }
  switch (a.R) {
    case 3: return launch_halo_ring<T, TO, CO_T, 3, 4, 64, 1, false, false, GEO>(a, st);
    case 5: return launch_halo_ring<T, TO, CO_T, 5, 4, 64, 1, false, false, GEO>(a, st);
    case 7: return launch_halo_ring<T, TO, CO_T, 7, 4, 64, 1, false, false, GEO>(a, st);
  }
  GLS_FAIL(GLSDET_E_ARG, "conv2d(halo ring): unsupported kernel size %d", a.R);
}

template <typename T, typename TO, int CO_T>
static int halo_ring_by_ks(const ConvArgs& a, hipStream_t st) {
  switch (a.R) {
    case 3: return launch_halo_ring<T, TO, CO_T, 3, 3>(a, st);
    case 5: return launch_halo_ring<T, TO, CO_T, 5, CO_T == 64 ? 4 : 3>(a, st);   // 64 rows: a fourth slot costs no workgroup per CU
    case 7: return launch_halo_ring<T, TO, CO_T, 7, CO_T == 64 ? 4 : 3>(a, st);
  }
  GLS_FAIL(GLSDET_E_ARG, "conv2d(halo ring): unsupported kernel size %d", a.R);
}

template <typename T, typename TO, int CO_T, int WCO, int KS, int TH, int TW, bool CH = false>
static int launch_halo(const ConvArgs& a, hipStream_t st) {
  if constexpr (!CH && KS == 3) {
    if (a.w2) return launch_halo<T, TO, CO_T, WCO, KS, TH, TW, true>(a, st);
  }
  if (a.w2 && !CH) GLS_FAIL(GLSDET_E_ARG, "conv2d_chain: this halo kernel has no chained form");
  constexpr int PH = TH + KS - 1, PW = TW + KS - 1;
  constexpr int stage = 2 * CO_T * 144 + PH * PW * 144;
  constexpr int epiw = epi_bytes<TO>(CO_T, TH * TW, true), ldsw = stage > epiw ? stage : epiw;
  const int epi = epi_bytes<TO>(CO_T, TH * TW, a.res != nullptr);
  int lds = stage > epi ? stage : epi;
  if (a.w2 && chain_lds_bytes<T>(CO_T, TH * TW, a) > lds) lds = chain_lds_bytes<T>(CO_T, TH * TW, a);
  auto kern = conv_halo_kernel<T, TO, CO_T, WCO, KS, TH, TW, CH>;
  static int attr_lds = 64 * 1024;
  const int want_attr = lds > ldsw ? lds : ldsw;
  if (want_attr > attr_lds) {
    GLS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, want_attr));
    attr_lds = want_attr;
  }
  ConvArgs b = a;
  b.n_co_tiles = (a.Cout + CO_T - 1) / CO_T;
  const int tiles_x = (a.Wo + TW - 1) / TW, tiles_y = (a.Ho + TH - 1) / TH;
  gls_fastdiv(b.n_co_tiles, &b.nco_mul, &b.nco_sh);
  gls_fastdiv(tiles_x, &b.tx_mul, &b.tx_sh);
  gls_fastdiv(tiles_y, &b.ty_mul, &b.ty_sh);
  const long grid = (long)b.n_co_tiles * tiles_x * tiles_y * a.N;
  if (grid <= 0 || grid > 0x7fffffffL) GLS_FAIL(GLSDET_E_ARG, "conv2d(halo): grid %ld out of range", grid);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, st, b, tiles_x, tiles_y);
  GLS_HIP(hipGetLastError());
  return 0;
}

template <typename T, typename TO, int CO_T, int WCO>
static int halo_by_ks(const ConvArgs& a, hipStream_t st) {
  switch (a.R) {
    case 3: return launch_halo<T, TO, CO_T, WCO, 3, 8, 16>(a, st);
    case 5: return launch_halo<T, TO, CO_T, WCO, 5, 8, 16>(a, st);
    case 7: return launch_halo<T, TO, CO_T, WCO, 7, 8, 16>(a, st);
  }
  GLS_FAIL(GLSDET_E_ARG, "conv2d(halo): unsupported kernel size %d", a.R);
}

template <typename T, typename TO, int CO_T>
static int halo_ring_k64_by_ks(const ConvArgs& a, hipStream_t st) {
  switch (a.R) {
    case 3: return launch_halo_ring<T, TO, CO_T, 3, 4, 64>(a, st);
    case 5: return launch_halo_ring<T, TO, CO_T, 5, 4, 64>(a, st);
    case 7: return launch_halo_ring<T, TO, CO_T, 7, 4, 64>(a, st);
  }
  GLS_FAIL(GLSDET_E_ARG, "conv2d(halo ring): unsupported kernel size %d", a.R);
}

// Returns 1 when the halo kernel does not apply (caller falls back to the generic kernel),
// 0 when `op` was filled in.
int conv_halo_try(const ConvArgs& a, int xdt, int ydt, int hint_in, OpRecord* op) {
  // tile geometry in bits 8..9 of a ring hint (8..13): 0x1xx = 10 x 12 (ring8: 10 x 24), 0x2xx = 6 x 21 (ring8: 6 x 42)
  const int geo = (hint_in >= 0x100 && hint_in < 0x300) ? (hint_in >> 8) : 0;
  const int hint = geo ? (hint_in & 0xff) : hint_in;
  if (geo && (hint < 8 || hint > 13)) return 1;
  if (hint == 12 || hint == 13) {      // 8-wave 128 x 256 form of the ring kernel (13: 64-byte channel chunks)
    const int es8 = dtype_size(xdt);
    if (a.w2 || a.res || a.gn_part || xdt != ydt || a.stride != 1 || a.R != a.S || (a.R != 3 && a.R != 5 && a.R != 7) || a.pad != a.R / 2 ||
        a.cout_pad <= 64 || (a.Cin * es8) % (hint == 13 ? 64 : 128))
      return 1;
    const bool k64 = hint == 13;
    if (geo && !k64) return 1;
    int gth, gtw;
    tile_geo8_dims(geo, &gth, &gtw);
    char nm8[96];
    snprintf(nm8, sizeof nm8, "conv_halo_ring8%s<%s,128x%dx%d> %dx%d s1 cin%d cout%d", k64 ? "_k64" : "", xdt ? "f32" : "f16", gth, gtw, a.R, a.S, a.Cin, a.Cout);
    op->name = nm8;
    op->launch = [a, xdt, k64, geo](hipStream_t st) -> int {
      return xdt == GLSDET_F16 ? halo_ring8_dispatch<f16>(a, k64, geo, st) : halo_ring8_dispatch<float>(a, k64, geo, st);
    };
    return 0;
  }
  if (hint == 1 || hint == 3 || (hint > 5 && hint != 8 && hint != 9 && hint != 10 && hint != 11)) return 1;          // hint 1 / explicit tile = the generic kernel
  const int es = dtype_size(xdt);
  if (a.w2 && (a.R != 3 || a.stride != 1)) return 1;      // chained 1x1: compiled into the 3x3 stride-1 forms only
  if (a.gn_part && (a.R != 3 || a.stride != 1 || hint < 8 || hint > 11 || a.w2 || a.res)) return 1;   // GN partials: ring forms only
  if (geo && (a.w2 || a.gn_part)) return 1;
  if (a.stride == 2) {            // 3x3 stride 2: the de-interleaved-patch form of the ring kernel, 64-byte channel chunks
    if (a.R != 3 || a.S != 3 || a.pad != 1 || xdt != ydt || (a.Cin * es) % 64 || (hint != 0 && hint != 10 && hint != 11)) return 1;
    if (hint == 11 && a.cout_pad <= 64) return 1;
    const long tiles = (long)((a.Ho + 7) / 8) * ((a.Wo + 15) / 16);
    if (hint == 0 && (double)tiles * 128.0 / ((double)a.Ho * a.Wo) > 1.30) return 1;
    const int co_t = (a.cout_pad <= 64 || hint != 11) ? 64 : 128;      // measured: 64-row tiles win unless the tuner says otherwise
    if (a.w2 && a.c2_0 / co_t != (a.c2_0 + a.cin2 - 1) / co_t) return 1;
    char nm[96];
    int sth, stw;
    tile_geo_dims(geo, &sth, &stw);
    snprintf(nm, sizeof nm, "conv_halo_ring_k64_s2<%s,%dx%dx%d> 3x3 s2 cin%d cout%d%s", xdt ? "f32" : "f16", co_t, sth, stw, a.Cin, a.Cout, a.w2 ? " +1x1" : "");
    op->name = nm;
    op->launch = [a, co_t, xdt, geo](hipStream_t st) -> int {
#define GLS_S2(T_, G_)                                                                                                                  \
      if (geo == G_) return co_t == 128 ? launch_halo_ring<T_, T_, 128, 3, 4, 64, 2, false, false, G_>(a, st)                               \
                                        : launch_halo_ring<T_, T_, 64, 3, 4, 64, 2, false, false, G_>(a, st);
      if (xdt == GLSDET_F16) { GLS_S2(f16, 1) GLS_S2(f16, 2) } else { GLS_S2(float, 1) GLS_S2(float, 2) }
#undef GLS_S2
      if (xdt == GLSDET_F16) return co_t == 128 ? launch_halo_ring<f16, f16, 128, 3, 4, 64, 2>(a, st) : launch_halo_ring<f16, f16, 64, 3, 4, 64, 2>(a, st);
      return co_t == 128 ? launch_halo_ring<float, float, 128, 3, 4, 64, 2>(a, st) : launch_halo_ring<float, float, 64, 3, 4, 64, 2>(a, st);
    };
    return 0;
  }
  if (a.stride != 1 || a.R != a.S || (a.R != 3 && a.R != 5 && a.R != 7) || a.pad != a.R / 2) return 1;
  if ((a.Cin * es) % ((hint == 10 || hint == 11) ? 64 : 128)) return 1;      // whole channel chunks (hint 10 works on 64-byte chunks)
  if (xdt != ydt) return 1;
  if (geo && !((hint == 8 || hint == 9) ? a.R == 3 : (hint == 10 || hint == 11))) return 1;
  // wasted MFMA work on partial tiles: prefer the flat-pixel kernel when it is large
  const long tiles = (long)((a.Ho + 7) / 8) * ((a.Wo + 15) / 16);
  const double waste = (double)tiles * 128.0 / ((double)a.Ho * a.Wo);
  if (hint != 2 && hint != 4 && hint != 5 && (hint < 8 || hint > 11) && waste > 1.30) return 1;      // a hint forces the halo kernel
  const bool ring = hint >= 8 && hint <= 11;       // weight tiles by LDS-DMA into a ring: 8 = 64-row, 9 = 128-row cout tiles,
  const bool ring_k64 = hint == 10 || hint == 11;  // 10 / 11 = 64- / 128-row tiles with 64-byte channel chunks
  if ((hint == 9 || hint == 11) && a.cout_pad <= 64) return 1;
  const int co_t = (a.cout_pad <= 64 || hint == 5 || hint == 8 || hint == 10) ? 64 : 128;     // hint 5 / 8 / 10: 64-row cout tiles also for wide layers
  const bool wpriv = hint == 4;                     // wave-private weight staging (128-row cout tile only)
  if (wpriv && co_t != 128) return 1;
  if (a.w2 && a.c2_0 / co_t != (a.c2_0 + a.cin2 - 1) / co_t) return 1;      // chained 1x1: its input channels in ONE cout tile
  char nm[96];
  int gth, gtw;
  tile_geo_dims(geo, &gth, &gtw);
  snprintf(nm, sizeof nm, "conv_halo%s<%s,%dx%dx%d> %dx%d s1 cin%d cout%d%s", wpriv ? "_wp" : (ring ? (ring_k64 ? "_ring_k64" : "_ring") : ""), xdt ? "f32" : "f16", co_t, gth, gtw, a.R,
           a.S, a.Cin, a.Cout, a.w2 ? " +1x1" : "");
  op->name = nm;
  op->launch = [a, co_t, xdt, wpriv, ring, ring_k64, geo](hipStream_t st) -> int {
    if (geo == 1) {
      if (xdt == GLSDET_F16) return co_t == 128 ? halo_ring_geo<f16, f16, 128, 1>(a, ring_k64, st) : halo_ring_geo<f16, f16, 64, 1>(a, ring_k64, st);
      return co_t == 128 ? halo_ring_geo<float, float, 128, 1>(a, ring_k64, st) : halo_ring_geo<float, float, 64, 1>(a, ring_k64, st);
    }
    if (geo == 2) {
      if (xdt == GLSDET_F16) return co_t == 128 ? halo_ring_geo<f16, f16, 128, 2>(a, ring_k64, st) : halo_ring_geo<f16, f16, 64, 2>(a, ring_k64, st);
      return co_t == 128 ? halo_ring_geo<float, float, 128, 2>(a, ring_k64, st) : halo_ring_geo<float, float, 64, 2>(a, ring_k64, st);
    }
    if (ring_k64) {
      if (xdt == GLSDET_F16) return co_t == 128 ? halo_ring_k64_by_ks<f16, f16, 128>(a, st) : halo_ring_k64_by_ks<f16, f16, 64>(a, st);
      return co_t == 128 ? halo_ring_k64_by_ks<float, float, 128>(a, st) : halo_ring_k64_by_ks<float, float, 64>(a, st);
    }
    if (ring) {
      if (xdt == GLSDET_F16) return co_t == 128 ? halo_ring_by_ks<f16, f16, 128>(a, st) : halo_ring_by_ks<f16, f16, 64>(a, st);
      return co_t == 128 ? halo_ring_by_ks<float, float, 128>(a, st) : halo_ring_by_ks<float, float, 64>(a, st);
    }
    if (wpriv) return xdt == GLSDET_F16 ? halo_by_ks<f16, f16, 128, 4>(a, st) : halo_by_ks<float, float, 128, 4>(a, st);
    if (xdt == GLSDET_F16) return co_t == 128 ? halo_by_ks<f16, f16, 128, 2>(a, st) : halo_by_ks<f16, f16, 64, 2>(a, st);
    return co_t == 128 ? halo_by_ks<float, float, 128, 2>(a, st) : halo_by_ks<float, float, 64, 2>(a, st);
  };
  return 0;
}

}  // namespace glsdet

