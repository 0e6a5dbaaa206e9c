// conv2d + folded BN + activation (+ residual) as an implicit GEMM on CDNA4 MFMA.
//
// GEMM view (operands swapped so that the contiguous NHWC channel run of the OUTPUT is
// register-local in the accumulator):
//     D[co][px] = sum_k  Wt[co][k] * Im2col[px][k],   k = (r*S + s)*Cin + ci
//   A operand = weight tile  [CO_T rows x KB bytes]  (rows contiguous in memory)
//   B operand = im2col tile  [PX_T rows x KB bytes]  (each 16-byte chunk = 8 f16 / 4 f32
//               consecutive input channels of ONE tap -> one coalesced 16-B global load,
//               zero-filled for padding / tails)
// Both tiles are staged global -> registers -> LDS (rows padded by 16 B: conflict-free
// ds_read_b128 fragments), double buffered, one barrier per K step, next tile's global
// loads in flight under the current tile's MFMAs.  Accumulation is fp32.
//   f16: v_mfma_f32_32x32x16_f16  (16 B of an A row x 16 B of a B row per lane)
//   f32: 4 x v_mfma_f32_32x32x2_f32 on the same 16-B fragments (k order permuted
//        identically for A and B, so the dot product is unchanged) -- exact f32.
// Epilogue: scale/bias/activation in fp32 on the accumulator, tile transposed through LDS,
// written (and the residual read) as full 16-byte channel chunks per pixel.
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "conv_common.h"

namespace glsdet {

template <int CO_T, int PX_T, int KB, typename TO>
constexpr int conv_lds_bytes(bool wide) {        // wide: fp32 staging of an fp16 output tile (residual layers)
  constexpr int stage = 2 * (CO_T + PX_T) * (KB + 16);
  const int epi = epi_bytes<TO>(CO_T, PX_T, wide);
  return (stage > epi ? stage : epi) + CO_T * 8;   // + scale | bias of the cout tile (fp32), parked in LDS for the epilogue
}

// XCD-aware tile order: the blocks that land on one XCD (blockIdx % 8 equal) walk a CONTIGUOUS run of the launch's tiles
// (cout tile fastest, then pixel tile, then -- grouped launches -- problem), so the operand rows they share stay in that
// XCD's L2.  In a grouped launch the run is cut from the concatenation of all problems: eight equal GEMMs land one per XCD
// (rocprofv3 PMC on config 3's 576 x 576 x 4224 Gram products, 64 x 64 tiles, eight per launch: 330 MB fetched per launch
// for 83 MB of operands while every XCD saw every problem).
__device__ __forceinline__ int xcd_tile(const int bid, const int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, local = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
}

// UT ("uniform tap"): Cin * sizeof(T) is a multiple of KB, so all chunks of a K step belong to ONE
// filter tap and the tap walk (kr, ks, channel base) is scalar: a K step then costs one VALU add
// per 16-byte chunk (plus the padding test when the conv pads) instead of ~10.
// CH: the chained 1x1 (glsdet_conv2d_chain) is compiled in.  A flag, not a runtime branch: with the chain code present
// every instantiation carried its accumulators and staged chunks (64x64 tile: 60 -> 89 VGPRs, 6 -> 3 waves per SIMD;
// the 128-row halo kernels 82-118 -> 238), which cost the plain launches 5-20 % (round 2 regression, found in the op table).
template <typename T, typename TO, int CO_T, int PX_T, int KB, int WCO, bool UT, bool CH = false, int NTHR = 256>
__device__ __forceinline__ void conv_igemm_body(const ConvArgs& a, const int tile) {
  constexpr int RS = KB + 16;                    // LDS row stride, bytes
  constexpr int VEC = 16 / (int)sizeof(T);       // elements per 16-B chunk
  constexpr int KE = KB / (int)sizeof(T);        // k elements per step
  constexpr int CPR = KB / 16;                   // chunks per tile row
  constexpr int NA = (CO_T * CPR + NTHR - 1) / NTHR;   // chunks per thread, weight tile
  constexpr int NB = (PX_T * CPR + NTHR - 1) / NTHR;   // chunks per thread, im2col tile
  constexpr int WPX = (NTHR / 64) / WCO;       // NTHR = 512: eight waves on a 128 x 256 tile (tile 128<<16|256), 64 x 64 each
  constexpr int WT_CO = CO_T / WCO, WT_PX = PX_T / WPX;
  constexpr int TM = WT_CO / 32, TN = WT_PX / 32;
  constexpr int STAGE = (CO_T + PX_T) * RS;
  static_assert(TM >= 1 && TN >= 1 && NTHR % CPR == 0 && (NTHR == 256 || !CH), "tile shape");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;

  const int ptile = gls_div(tile, a.nco_mul, a.nco_sh);
  const int co0 = (tile - ptile * a.n_co_tiles) * CO_T;
  const int px0 = ptile * PX_T;

  const int kc = tid % CPR;
  const int row0 = tid / CPR;                    // row of chunk i is row0 + i*(256/CPR)
  constexpr int ROWS_PER_PASS = NTHR / CPR;

  const auto xrs = gls_make_rsrc(a.x_lo, a.x_bytes);
  const auto wrs = gls_make_rsrc(a.w, a.w_bytes);
  // ---- weight tile: per-thread row offsets (bytes; GLS_OOB rows read as zeros)
  unsigned wp[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int row = row0 + i * ROWS_PER_PASS;
    const bool ok = row < CO_T && (co0 + row) < a.cout_pad;
    wp[i] = ok ? (unsigned)(((co0 + row) * a.kpad + kc * VEC) * (int)sizeof(T)) : GLS_OOB;
  }
  // ---- im2col tile: per-thread pixel coordinates
  int boff[NB];                                  // element offset from the view base (may be < 0)
  int hi0[NB], wi0[NB];
  bool pok[NB];
  const int HoWo = a.Ho * a.Wo;
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int row = row0 + i * ROWS_PER_PASS;
    const int p = px0 + row;
    pok[i] = row < PX_T && p < a.M;
    const int pp = pok[i] ? p : 0;
    if (a.x_lin) {              // 1x1 / stride 1 / no padding on a pixel-linear view: no division, always in range
      hi0[i] = wi0[i] = 0;
      boff[i] = (int)((long)pp * a.x_sw);
    } else {
      const int n = gls_div(pp, a.howo_mul, a.howo_sh), rem = pp - n * HoWo;
      const int ho = gls_div(rem, a.wo_mul, a.wo_sh), wo = rem - ho * a.Wo;
      hi0[i] = ho * a.stride - a.pad;
      wi0[i] = wo * a.stride - a.pad;
      boff[i] = (int)((long)n * a.x_sn + (long)hi0[i] * a.x_sh + (long)wi0[i] * a.x_sw);
    }
  }
  // ---- k position of this thread's chunk column: (r, s, c)
  int kr, ks, kci;
  {
    const int k = kc * VEC;
    const int tap = k / a.Cin;
    kci = k - tap * a.Cin;
    kr = tap / a.S;
    ks = tap - kr * a.S;
  }

  const int nsteps = (a.kreal + KE - 1) / KE;
  // register ring of three K-step tiles: the loads of step t+3 are issued while step t is
  // multiplied (a 64x64 tile's MFMA phase is ~0.1 us, one step of lookahead cannot cover an
  // L2/HBM round trip; the small-K / small-tile layers were latency bound on it)
  u32x4 ra0[NA], rb0[NB], ra1[NA], rb1[NB], ra2[NA], rb2[NB];

  // (Measured, tools/ab_lib.sh: issuing the loads unconditionally with padding steps -- so that the
  // compiler can count them and wait with vmcnt(N) -- was 5-20 % SLOWER than this branchy form on
  // the short-K layers and no faster on the long-K ones; the extra barriers cost more than the
  // exposed waits.)
  // UT state: scalar tap walk + per-chunk byte offsets that already hold everything per-thread
  int u_kr = 0, u_ks = 0, u_kci = 0;                     // wave-uniform
  unsigned ub[NB];                                       // x_off + (pixel origin + chunk column) bytes, or GLS_OOB
  const bool u_nopad = a.pad == 0 && (a.Ho - 1) * a.stride + a.R <= a.H && (a.Wo - 1) * a.stride + a.S <= a.W;
  if (UT) {
#pragma unroll
    for (int i = 0; i < NB; ++i)
      ub[i] = pok[i] ? a.x_off + (unsigned)((boff[i] + kc * VEC) * (int)sizeof(T)) : GLS_OOB;
  }
  auto gload = [&](int step, u32x4 (&ra)[NA], u32x4 (&rb)[NB]) __attribute__((always_inline)) {
    const bool live = step < nsteps && !((a.dbg & 1) && step > 0);
    if (UT) {
      const unsigned kbyte = live ? (unsigned)step * KB : GLS_OOB;       // scalar
#pragma unroll
      for (int i = 0; i < NA; ++i) ra[i] = gls_buf_load16(wrs, wp[i] + kbyte);
      const unsigned toff = live ? (unsigned)(((long)u_kr * a.x_sh + (long)u_ks * a.x_sw + u_kci) * (long)sizeof(T))
                                 : GLS_OOB;                             // scalar
      if (u_nopad) {
#pragma unroll
        for (int i = 0; i < NB; ++i) rb[i] = gls_buf_load16(xrs, ub[i] + toff);
      } else {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          const bool ok = (unsigned)(hi0[i] + u_kr) < (unsigned)a.H && (unsigned)(wi0[i] + u_ks) < (unsigned)a.W;
          rb[i] = gls_buf_load16(xrs, ok ? ub[i] + toff : GLS_OOB);
        }
      }
      u_kci += KE;
      if (u_kci == a.Cin) {
        u_kci = 0;
        if (++u_ks == a.S) {
          u_ks = 0;
          ++u_kr;
        }
      }
      return;
    }
    const unsigned kbyte = (unsigned)step * KB;
#pragma unroll
    for (int i = 0; i < NA; ++i) ra[i] = gls_buf_load16(wrs, live ? wp[i] + kbyte : GLS_OOB);   // GLS_OOB + small stays out of range
    const bool kval = live && kr < a.R;
    const int tapoff = (int)((long)kr * a.x_sh + (long)ks * a.x_sw) + kci;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int hi = hi0[i] + kr, wi = wi0[i] + ks;
      const bool ok = pok[i] && kval && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
      rb[i] = gls_buf_load16(xrs, ok ? a.x_off + (unsigned)((boff[i] + tapoff) * (int)sizeof(T)) : GLS_OOB);
    }
    kci += KE;
    if (a.Cin >= KE) {          // at most one tap boundary per step: selects, no loop
      const bool wrap = kci >= a.Cin;
      kci -= wrap ? a.Cin : 0;
      ks += wrap ? 1 : 0;
      const bool wrap2 = ks == a.S;
      ks = wrap2 ? 0 : ks;
      kr += wrap2 ? 1 : 0;
    } else {
      while (kci >= a.Cin) {
        kci -= a.Cin;
        if (++ks == a.S) {
          ks = 0;
          ++kr;
        }
      }
    }
  };
  auto lstore = [&](int buf, const u32x4 (&ra)[NA], const u32x4 (&rb)[NB]) __attribute__((always_inline)) {
    unsigned char* sa = smem + buf * STAGE + kc * 16;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int row = row0 + i * ROWS_PER_PASS;
      if (CO_T % ROWS_PER_PASS == 0 || row < CO_T) *reinterpret_cast<u32x4*>(sa + row * RS) = ra[i];
    }
    unsigned char* sb = sa + CO_T * RS;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int row = row0 + i * ROWS_PER_PASS;
      if (PX_T % ROWS_PER_PASS == 0 || row < PX_T) *reinterpret_cast<u32x4*>(sb + row * RS) = rb[i];
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
  const int b_off = CO_T * RS + (wpx * WT_PX + l31) * RS + lh * 16;

  // folded-BN scale / bias of the cout tile: requested FIRST (so that waiting for them waits for nothing else), parked in
  // LDS behind the staging area once step 0's operands have landed anyway -- the epilogue then starts without a dependent
  // global round trip (these launches are one round of workgroups whose life IS such a chain) and no register lives
  // through the K loop for it (holding them in registers cost the 64x64 tile two of its six waves per SIMD)
  unsigned char* sSB = smem + conv_lds_bytes<CO_T, PX_T, KB, TO>(sizeof(TO) == 2 && a.res != nullptr) - CO_T * 8;
  f32x4 sb_s = {0.f, 0.f, 0.f, 0.f}, sb_b = {0.f, 0.f, 0.f, 0.f};
  if (tid < CO_T / 4 && co0 + tid * 4 < a.cout_pad) {
    sb_s = *reinterpret_cast<const f32x4*>(a.scale + co0 + tid * 4);
    sb_b = *reinterpret_cast<const f32x4*>(a.bias + co0 + tid * 4);
  }
  gload(0, ra0, rb0);
  lstore(0, ra0, rb0);
  if (tid < CO_T / 4) {
    *reinterpret_cast<f32x4*>(sSB + tid * 16) = sb_s;
    *reinterpret_cast<f32x4*>(sSB + CO_T * 4 + tid * 16) = sb_b;
  }
  if (nsteps > 1) gload(1, ra1, rb1);
  if (nsteps > 2) gload(2, ra2, rb2);
  __syncthreads();

  // one K step: issue step t+3 into the set that held step t (already in LDS), multiply step t
  // from LDS buffer t&1, then move step t+1 (in flight for two steps) into the other buffer
  auto kstep = [&](int t, u32x4 (&fa)[NA], u32x4 (&fb)[NB], const u32x4 (&na)[NA], const u32x4 (&nb)[NB])
      __attribute__((always_inline)) {
    const int cur = t & 1;
    if (t + 3 < nsteps) gload(t + 3, fa, fb);
    const unsigned char* sbuf = smem + cur * STAGE;
    if (t < nsteps && !(a.dbg & 2)) {
#pragma unroll
      for (int kk = 0; kk < KB / 32; ++kk) {
        u32x4 af[TM], bf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
          af[i] = *reinterpret_cast<const u32x4*>(sbuf + a_off + i * 32 * RS + kk * 32);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          bf[j] = *reinterpret_cast<const u32x4*>(sbuf + b_off + j * 32 * RS + kk * 32);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) MMA<T>::run(af[i], bf[j], acc[i][j]);
      }
    }
    if (t + 1 < nsteps) lstore(cur ^ 1, na, nb);
    __syncthreads();
  };
  for (int t = 0; t < nsteps; t += 3) {
    kstep(t, ra0, rb0, ra1, rb1);
    if (t + 1 < nsteps) kstep(t + 1, ra1, rb1, ra2, rb2);
    if (t + 2 < nsteps) kstep(t + 2, ra2, rb2, ra0, rb0);
  }
  if (a.dbg & 4) return;
  // ---- epilogue: fp32 scale/bias/act, transpose through LDS, 16-B channel chunks out
  constexpr int ORS = CO_T * (int)sizeof(TO) + 16;
  const bool wide = sizeof(TO) == 2 && a.res != nullptr;      // fp32 staging: the residual is added before the one rounding
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int co_l = wco * WT_CO + i * 32 + 8 * g + 4 * lh;
      const f32x4 sc = *reinterpret_cast<const f32x4*>(sSB + co_l * 4), bi = *reinterpret_cast<const f32x4*>(sSB + CO_T * 4 + co_l * 4);
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
  constexpr int VO = 16 / (int)sizeof(TO);
  constexpr int OCPR = CO_T / VO;                // 16-B chunks per pixel row of the tile
  // chained 1x1 (glsdet_conv2d_chain): only the workgroup whose cout tile holds the chained conv's input channels
  const bool chain = CH && sizeof(T) == sizeof(TO) && a.w2 != nullptr && co0 <= a.c2_0 && a.c2_0 + a.cin2 <= co0 + CO_T;
  auto y2pix = [&](int px_l, bool& ok) -> long {
    const int p = px0 + px_l;
    ok = p < a.M;
    return ok ? gls_pix_off(p, HoWo, a.Wo, a.y2_sn, a.y2_sh, a.y2_sw, a.y2_lin, a) : 0;
  };
  if constexpr (CH && sizeof(T) == sizeof(TO)) if (chain && a.res) {
    // the FINAL tile (after the residual) must stand in LDS for the chained product: pass 1 computes every chunk in
    // registers (and stores it to y), pass 2 writes the chunks back in the TO row layout
    constexpr int NITC = (PX_T * OCPR + NTHR - 1) / NTHR;
    constexpr int ORSW = CO_T * 4 + 16;
    u32x4 fin[NITC];
    bool okc[NITC];
#pragma unroll
    for (int b = 0; b < NITC; ++b) {
      const int q = tid + b * NTHR;
      const int px_l = q / OCPR, cc = q - px_l * OCPR;
      const int p = px0 + px_l, co = co0 + cc * VO;
      okc[b] = q < PX_T * OCPR && p < a.M && co < a.Cout;
      fin[b] = u32x4{0u, 0u, 0u, 0u};
      if (okc[b]) {
        const long ro = gls_pix_off(p, HoWo, a.Wo, a.r_sn, a.r_sh, a.r_sw, a.r_lin, a) + co;
        const u32x4 rv = *reinterpret_cast<const u32x4*>(a.res + ro * (long)sizeof(TO));
        if (wide) {
          const f32x4 lo = *reinterpret_cast<const f32x4*>(smem + px_l * ORSW + cc * 32);
          const f32x4 hi = *reinterpret_cast<const f32x4*>(smem + px_l * ORSW + cc * 32 + 16);
          fin[b] = add_chunk_wide(lo, hi, rv, a.act_post);
        } else {
          fin[b] = add_chunk(*reinterpret_cast<const u32x4*>(smem + px_l * ORS + cc * 16), rv, (TO*)nullptr, a.act_post);
        }
        const long yo = gls_pix_off(p, HoWo, a.Wo, a.y_sn, a.y_sh, a.y_sw, a.y_lin, a) + co;
        *reinterpret_cast<u32x4*>(a.y + yo * (long)sizeof(TO)) = fin[b];
      }
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < NITC; ++b) {
      const int q = tid + b * NTHR;
      const int px_l = q / OCPR, cc = q - px_l * OCPR;
      if (q < PX_T * OCPR) *reinterpret_cast<u32x4*>(smem + px_l * ORS + cc * 16) = fin[b];
    }
    chain_1x1<T, CO_T, PX_T>(a, smem, smem + PX_T * ORS, co0, tid, y2pix);
    return;
  }
  if (wide) {                                    // residual loads batched four deep, fp32 add, one rounding
    constexpr int ORSW = CO_T * 4 + 16, NITW = (PX_T * OCPR + NTHR - 1) / NTHR, EBW = NITW < 4 ? NITW : 4;
    for (int it0 = 0; it0 < NITW; it0 += EBW) {
      u32x4 rv[EBW];
      long yo[EBW];
      bool ok[EBW];
#pragma unroll
      for (int b = 0; b < EBW; ++b) {
        const int q = tid + (it0 + b) * NTHR;
        const int px_l = q / OCPR, cc = q - px_l * OCPR;
        const int p = px0 + px_l, co = co0 + cc * VO;
        ok[b] = q < PX_T * OCPR && p < a.M && co < a.Cout;
        yo[b] = 0;
        if (ok[b]) {
          yo[b] = gls_pix_off(p, HoWo, a.Wo, a.y_sn, a.y_sh, a.y_sw, a.y_lin, a) + co;
          const long ro = gls_pix_off(p, HoWo, a.Wo, a.r_sn, a.r_sh, a.r_sw, a.r_lin, a) + co;
          rv[b] = *reinterpret_cast<const u32x4*>(a.res + ro * 2);
        }
      }
#pragma unroll
      for (int b = 0; b < EBW; ++b) {
        if (ok[b]) {
          const int q = tid + (it0 + b) * NTHR;
          const int px_l = q / OCPR, cc = q - px_l * OCPR;
          const f32x4 lo = *reinterpret_cast<const f32x4*>(smem + px_l * ORSW + cc * 32);
          const f32x4 hi = *reinterpret_cast<const f32x4*>(smem + px_l * ORSW + cc * 32 + 16);
          *reinterpret_cast<u32x4*>(a.y + yo[b] * 2) = add_chunk_wide(lo, hi, rv[b], a.act_post);
        }
      }
    }
    return;
  }
  // written in batches of EB chunks per thread: all residual loads of a batch are issued before the
  // first add / store, so a residual layer keeps EB x 16 B per thread in flight instead of one
  // (measured: +3...8 % on the residual 1x1 layers with 64-wide tiles; on the 128x128 tile, 8 chunks
  // per thread, the extra live registers cost 25 % -- there the chunks go one by one)
  constexpr int NIT = (PX_T * OCPR + NTHR - 1) / NTHR;
  if constexpr (NIT > 4) {
    for (int q = tid; q < PX_T * OCPR; q += NTHR) {
      const int px_l = q / OCPR, cc = q - px_l * OCPR;
      const int p = px0 + px_l, co = co0 + cc * VO;
      if (p < a.M && co < a.Cout) {
        u32x4 v = *reinterpret_cast<const u32x4*>(smem + px_l * ORS + cc * 16);
        if (a.res) {
          const long ro = gls_pix_off(p, HoWo, a.Wo, a.r_sn, a.r_sh, a.r_sw, a.r_lin, a) + co;
          v = add_chunk(v, *reinterpret_cast<const u32x4*>(a.res + ro * (long)sizeof(TO)), (TO*)nullptr, a.act_post);
        }
        const long yo = gls_pix_off(p, HoWo, a.Wo, a.y_sn, a.y_sh, a.y_sw, a.y_lin, a) + co;
        if (!CH || !a.y_skip) *reinterpret_cast<u32x4*>(a.y + yo * (long)sizeof(TO)) = v;
      }
    }
    if constexpr (CH && sizeof(T) == sizeof(TO)) {
      if (chain) chain_1x1<T, CO_T, PX_T>(a, smem, smem + PX_T * ORS, co0, tid, y2pix);
    }
    return;
  }
  constexpr int EB = NIT <= 4 ? NIT : 1;
  for (int it0 = 0; it0 < NIT; it0 += EB) {
    u32x4 rv[EB];
    long yo[EB];
    bool ok[EB];
#pragma unroll
    for (int b = 0; b < EB; ++b) {
      const int q = tid + (it0 + b) * NTHR;
      const int px_l = q / OCPR, cc = q - px_l * OCPR;
      const int p = px0 + px_l, co = co0 + cc * VO;
      ok[b] = q < PX_T * OCPR && p < a.M && co < a.Cout;
      yo[b] = 0;
      if (ok[b]) {
        yo[b] = gls_pix_off(p, HoWo, a.Wo, a.y_sn, a.y_sh, a.y_sw, a.y_lin, a) + co;
        if (a.res) {
          const long ro = gls_pix_off(p, HoWo, a.Wo, a.r_sn, a.r_sh, a.r_sw, a.r_lin, a) + co;
          rv[b] = *reinterpret_cast<const u32x4*>(a.res + ro * (long)sizeof(TO));
        }
      }
    }
#pragma unroll
    for (int b = 0; b < EB; ++b) {
      if (ok[b]) {
        const int q = tid + (it0 + b) * NTHR;
        const int px_l = q / OCPR, cc = q - px_l * OCPR;
        u32x4 v = *reinterpret_cast<const u32x4*>(smem + px_l * ORS + cc * 16);
        if (a.res) v = add_chunk(v, rv[b], (TO*)nullptr, a.act_post);
        if (!CH || !a.y_skip) *reinterpret_cast<u32x4*>(a.y + yo[b] * (long)sizeof(TO)) = v;
      }
    }
  }
  if constexpr (CH && sizeof(T) == sizeof(TO)) {
    if (chain) chain_1x1<T, CO_T, PX_T>(a, smem, smem + PX_T * ORS, co0, tid, y2pix);      // (chain && res returned above)
  }
}

template <typename T, typename TO, int CO_T, int PX_T, int KB, int WCO, bool UT, bool CH = false, int NTHR = 256>
__global__ __launch_bounds__(NTHR) void conv_igemm_kernel(const ConvArgs a) {
  conv_igemm_body<T, TO, CO_T, PX_T, KB, WCO, UT, CH, NTHR>(a, xcd_tile((int)blockIdx.x, (int)gridDim.x));
}

// Several independent convolutions of the SAME shape class (kernel size, stride, channels, dtypes:
// the four quadrant convs of the GL-fusion block, the cls/reg tower convs of one level, the per-image /
// per-quadrant GEMMs of the large-channel non-local block) as ONE
// launch: each is too small to fill 256 CUs on its own.  The argument blocks travel in the kernarg
// segment; a workgroup finds its problem from the prefix of tile counts (wave-uniform).
// (ConvArgsN: conv_common.h)
template <typename T, typename TO, int CO_T, int PX_T, int KB, int WCO, bool UT>
__global__ __launch_bounds__(256) void conv_igemm_multi_kernel(const ConvArgsN m) {
  const int L = xcd_tile((int)blockIdx.x, (int)gridDim.x);
  int g = 0;
#pragma unroll
  for (int i = 1; i < GLS_MULTI; ++i)
    if (i < m.n && L >= m.start[i]) g = i;
  conv_igemm_body<T, TO, CO_T, PX_T, KB, WCO, UT>(m.p[g], L - m.start[g]);
}

static_assert(sizeof(ConvArgsB) <= 4096 && sizeof(ConvArgsN) <= 4096, "the kernarg segment holds 4 KB");
// ... and up to GLS_BATCH problems of IDENTICAL geometry (ConvArgsB: they differ in their operand addresses only)
template <typename T, typename TO, int CO_T, int PX_T, int KB, int WCO, bool UT>
__global__ __launch_bounds__(256) void conv_igemm_batch_kernel(const ConvArgsB m) {
  const int L = xcd_tile((int)blockIdx.x, (int)gridDim.x);
  const int g = gls_div(L, m.tiles_mul, m.tiles_sh);
  ConvArgs a = m.base;
  const ConvPtrs& q = m.p[g];
  a.x = q.x; a.w = q.w; a.scale = q.scale; a.bias = q.bias; a.y = q.y; a.res = q.res;
  a.x_lo = q.x_lo; a.x_off = q.x_off; a.x_bytes = q.x_bytes; a.w_bytes = q.w_bytes;
  conv_igemm_body<T, TO, CO_T, PX_T, KB, WCO, UT>(a, L - g * m.tiles);
}

// ---- host side --------------------------------------------------------------------------
template <typename T, typename TO, int CO_T, int PX_T, int KB, int WCO, bool UT, bool CH = false, int NTHR = 256>
static int launch_conv(const ConvArgs& a, hipStream_t st) {
  if constexpr (!CH && UT && CO_T >= 64 && sizeof(T) == sizeof(TO) && NTHR == 256) {
    if (a.w2) return launch_conv<T, TO, CO_T, PX_T, KB, WCO, UT, true>(a, st);
  }
  if (a.w2 && !CH) GLS_FAIL(GLSDET_E_ARG, "conv2d_chain: this tile of the generic kernel has no chained form");
  int lds = conv_lds_bytes<CO_T, PX_T, KB, TO>(a.res != nullptr);
  if (a.w2 && chain_lds_bytes<T>(CO_T, PX_T, a) > lds) lds = chain_lds_bytes<T>(CO_T, PX_T, a);
  static int attr_lds = 64 * 1024;
  auto kern = conv_igemm_kernel<T, TO, CO_T, PX_T, KB, WCO, UT, CH, NTHR>;
  const int want_attr = lds > conv_lds_bytes<CO_T, PX_T, KB, TO>(true) ? lds : conv_lds_bytes<CO_T, PX_T, KB, TO>(true);
  if (want_attr > attr_lds) {
    GLS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, want_attr));
    attr_lds = want_attr;
  }
  ConvArgs b = a;
  b.n_co_tiles = (a.cout_pad + CO_T - 1) / CO_T;
  // tiles that would only cover the zero padding of cout_pad are never created
  if ((b.n_co_tiles - 1) * CO_T >= a.Cout) b.n_co_tiles = (a.Cout + CO_T - 1) / CO_T;
  gls_fastdiv(b.n_co_tiles, &b.nco_mul, &b.nco_sh);
  b.n_px_tiles = (a.M + PX_T - 1) / PX_T;
  const long grid = (long)b.n_co_tiles * b.n_px_tiles;
  if (grid <= 0 || grid > 0x7fffffffL) GLS_FAIL(GLSDET_E_ARG, "conv2d: grid %ld out of range", grid);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NTHR), lds, st, b);
  GLS_HIP(hipGetLastError());
  return 0;
}

template <typename T, typename TO, int CO_T, int PX_T, int KB, int WCO, bool UT>
static int launch_conv_multi(const ConvArgsN& m0, hipStream_t st) {
  bool any_res = false;
  for (int i = 0; i < m0.n; ++i) any_res = any_res || m0.p[i].res != nullptr;
  const int lds = conv_lds_bytes<CO_T, PX_T, KB, TO>(any_res);
  static bool attr_set = false;
  auto kern = conv_igemm_multi_kernel<T, TO, CO_T, PX_T, KB, WCO, UT>;
  if (!attr_set && conv_lds_bytes<CO_T, PX_T, KB, TO>(true) > 64 * 1024) {
    GLS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, conv_lds_bytes<CO_T, PX_T, KB, TO>(true)));
    attr_set = true;
  }
  ConvArgsN m = m0;
  long grid = 0;
  for (int i = 0; i < m.n; ++i) {
    ConvArgs& b = m.p[i];
    b.n_co_tiles = (b.cout_pad + CO_T - 1) / CO_T;
    if ((b.n_co_tiles - 1) * CO_T >= b.Cout) b.n_co_tiles = (b.Cout + CO_T - 1) / CO_T;
    gls_fastdiv(b.n_co_tiles, &b.nco_mul, &b.nco_sh);
    b.n_px_tiles = (b.M + PX_T - 1) / PX_T;
    m.start[i] = (int)grid;
    grid += (long)b.n_co_tiles * b.n_px_tiles;
  }
  for (int i = m.n; i <= GLS_MULTI; ++i) m.start[i] = (int)grid;
  if (grid <= 0 || grid > 0x7fffffffL) GLS_FAIL(GLSDET_E_ARG, "conv2d_multi: grid %ld out of range", grid);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, st, m);
  GLS_HIP(hipGetLastError());
  return 0;
}

template <typename T, typename TO, int CO_T, int PX_T, int KB, int WCO, bool UT>
static int launch_conv_batch(const ConvArgsB& m0, hipStream_t st) {
  const int lds = conv_lds_bytes<CO_T, PX_T, KB, TO>(m0.base.res != nullptr);
  static bool attr_set = false;
  auto kern = conv_igemm_batch_kernel<T, TO, CO_T, PX_T, KB, WCO, UT>;
  if (!attr_set && conv_lds_bytes<CO_T, PX_T, KB, TO>(true) > 64 * 1024) {
    GLS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, conv_lds_bytes<CO_T, PX_T, KB, TO>(true)));
    attr_set = true;
  }
  ConvArgsB m = m0;
  ConvArgs& b = m.base;
  b.n_co_tiles = (b.cout_pad + CO_T - 1) / CO_T;
  if ((b.n_co_tiles - 1) * CO_T >= b.Cout) b.n_co_tiles = (b.Cout + CO_T - 1) / CO_T;
  gls_fastdiv(b.n_co_tiles, &b.nco_mul, &b.nco_sh);
  b.n_px_tiles = (b.M + PX_T - 1) / PX_T;
  m.tiles = b.n_co_tiles * b.n_px_tiles;
  gls_fastdiv(m.tiles, &m.tiles_mul, &m.tiles_sh);
  const long grid = (long)m.tiles * m.n;
  if (grid <= 0 || grid > 0x7fffffffL) GLS_FAIL(GLSDET_E_ARG, "conv2d_multi: grid %ld out of range", grid);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, st, m);
  GLS_HIP(hipGetLastError());
  return 0;
}

template <typename T, typename TO>
static int dispatch_tile_batch(const ConvArgsB& m, int co_t, int px_t, int kb, hipStream_t st) {
  const bool ut = ((long)m.base.Cin * (long)sizeof(T)) % kb == 0;
#define GLS_CASE(CO, PX, WCO_)                                                                      \
  if (co_t == CO && px_t == PX) {                                                                   \
    if (ut) return kb == 128 ? launch_conv_batch<T, TO, CO, PX, 128, WCO_, true>(m, st)             \
                             : launch_conv_batch<T, TO, CO, PX, 64, WCO_, true>(m, st);             \
    return kb == 128 ? launch_conv_batch<T, TO, CO, PX, 128, WCO_, false>(m, st)                    \
                     : launch_conv_batch<T, TO, CO, PX, 64, WCO_, false>(m, st);                    \
  }
  GLS_CASE(128, 128, 2)
  GLS_CASE(64, 128, 2)
  GLS_CASE(64, 64, 2)
#undef GLS_CASE
  GLS_FAIL(GLSDET_E_ARG, "conv2d_multi: no kernel for tile %dx%d", co_t, px_t);
}

template <typename T, typename TO>
static int dispatch_tile_multi(const ConvArgsN& m, int co_t, int px_t, int kb, hipStream_t st) {
  const bool ut = ((long)m.p[0].Cin * (long)sizeof(T)) % kb == 0;
#define GLS_CASE(CO, PX, WCO_)                                                                      \
  if (co_t == CO && px_t == PX) {                                                                   \
    if (ut) return kb == 128 ? launch_conv_multi<T, TO, CO, PX, 128, WCO_, true>(m, st)             \
                             : launch_conv_multi<T, TO, CO, PX, 64, WCO_, true>(m, st);             \
    return kb == 128 ? launch_conv_multi<T, TO, CO, PX, 128, WCO_, false>(m, st)                    \
                     : launch_conv_multi<T, TO, CO, PX, 64, WCO_, false>(m, st);                    \
  }
  GLS_CASE(128, 128, 2)
  GLS_CASE(64, 128, 2)
  GLS_CASE(64, 64, 2)
#undef GLS_CASE
  GLS_FAIL(GLSDET_E_ARG, "conv2d_multi: no kernel for tile %dx%d", co_t, px_t);
}

template <typename T, typename TO>
static int dispatch_tile(const ConvArgs& a, int co_t, int px_t, int kb, hipStream_t st) {
  static const bool no_ut = getenv("GLSDET_NO_UT") != nullptr;            // A/B switch for measurements
  const bool ut = !no_ut && ((long)a.Cin * (long)sizeof(T)) % kb == 0;     // every K step inside one filter tap
#define GLS_CASE(CO, PX, WCO_)                                                                      \
  if (co_t == CO && px_t == PX) {                                                                   \
    if (ut) return kb == 128 ? launch_conv<T, TO, CO, PX, 128, WCO_, true>(a, st)                   \
                             : launch_conv<T, TO, CO, PX, 64, WCO_, true>(a, st);                   \
    return kb == 128 ? launch_conv<T, TO, CO, PX, 128, WCO_, false>(a, st)                          \
                     : launch_conv<T, TO, CO, PX, 64, WCO_, false>(a, st);                          \
  }
  GLS_CASE(128, 128, 2)
  GLS_CASE(64, 128, 2)
  GLS_CASE(32, 128, 1)
  GLS_CASE(64, 64, 2)
#undef GLS_CASE
  if (co_t == 128 && px_t == 256) {          // eight waves, 64 x 64 each (uniform-tap problems only)
    if (!ut || a.w2) GLS_FAIL(GLSDET_E_ARG, "conv2d: the 128x256 tile needs Cin to be a whole number of K steps and no chained conv");
    return kb == 128 ? launch_conv<T, TO, 128, 256, 128, 2, true, false, 512>(a, st) : launch_conv<T, TO, 128, 256, 64, 2, true, false, 512>(a, st);
  }
  GLS_FAIL(GLSDET_E_ARG, "conv2d: no kernel for tile %dx%d", co_t, px_t);
}

static void pick_tile(const ConvArgs& a, int elem, int hint, int* co_t, int* px_t, int* kb) {
  if (hint) {
    *co_t = hint >> 16;
    *px_t = hint & 0xff;
    if (*px_t == 0) *px_t = 256;        // co<<16 | 0: the eight-wave 128 x 256 tile (bits 8..10 of a tile hint are diagnostic switches)
  } else {
    // largest tile that still gives the chip >= 3 workgroups per CU; below that the layer
    // is latency bound and more, smaller workgroups win over MFMA density
    struct Cand { int co, px; };
    const Cand cands[] = {{128, 128}, {64, 128}, {64, 64}, {32, 128}};
    *co_t = 0;
    long best_blocks = -1;
    for (const Cand& c : cands) {
      if (a.w2 && (c.co < 64 || a.c2_0 / c.co != (a.c2_0 + a.cin2 - 1) / c.co)) continue;   // chained 1x1: inputs in one cout tile
      if (c.co > 32 && a.cout_pad <= c.co / 2) continue;             // mostly padding
      if (c.co == 32 && a.cout_pad > 32) continue;
      if (c.co == 128 && (a.cout_pad % 128) == 64 && a.cout_pad <= 320) continue;   // 192, 320: 64-wide tiles waste nothing
      const long blocks = (long)((a.cout_pad + c.co - 1) / c.co) * ((a.M + c.px - 1) / c.px);
      if (blocks >= 768) { *co_t = c.co; *px_t = c.px; break; }
      if (blocks > best_blocks) { best_blocks = blocks; *co_t = c.co; *px_t = c.px; }
    }
  }
  *kb = ((long)a.kreal * elem) % 128 == 0 ? 128 : 64;
  if (hint & 0x8000) *kb = 64;      // explicit 64-byte K steps: half the LDS / registers -> more workgroups per CU
}

}  // namespace glsdet

using namespace glsdet;

extern "C" int32_t glsdet_conv_kpad(int32_t R, int32_t S, int32_t cin, int32_t dtype) {
  const int es = dtype_size(dtype);
  const long kb = (long)R * S * cin * es;
  return (int32_t)(((kb + 127) / 128 * 128) / es);
}
extern "C" int32_t glsdet_conv_cout_pad(int32_t cout) { return (cout + 31) / 32 * 32; }
extern "C" int64_t glsdet_conv_weight_elems(int32_t cout, int32_t R, int32_t S, int32_t cin, int32_t dtype) {
  return (int64_t)glsdet_conv_cout_pad(cout) * glsdet_conv_kpad(R, S, cin, dtype);
}

// validate the descriptor and fill the kernel argument block
static int make_conv_args(const glsdet_conv_desc* d, int hint, ConvArgs& a, double* flops, double* bytes) {
  if (!d) GLS_FAIL(GLSDET_E_ARG, "conv2d: null descriptor");
  const glsdet_view &x = d->x, &y = d->y;
  int rc;
  if ((rc = check_view(x, "conv2d.x"))) return rc;
  if ((rc = check_view(y, "conv2d.y"))) return rc;
  if (x.dtype == GLSDET_F32 && y.dtype != GLSDET_F32)
    GLS_FAIL(GLSDET_E_ARG, "conv2d: f32 input needs f32 output");
  if (d->R < 1 || d->S < 1 || d->R > 15 || d->S > 15 || d->stride < 1 || d->stride > 4 || d->pad < 0)
    GLS_FAIL(GLSDET_E_ARG, "conv2d: bad R/S/stride/pad %d %d %d %d", d->R, d->S, d->stride, d->pad);
  if (x.c % 8 || y.c % 8) GLS_FAIL(GLSDET_E_ARG, "conv2d: channels must be multiples of 8 (%d,%d)", x.c, y.c);
  const int Ho = (x.h + 2 * d->pad - d->R) / d->stride + 1;
  const int Wo = (x.w + 2 * d->pad - d->S) / d->stride + 1;
  if (y.n != x.n || y.h != Ho || y.w != Wo)
    GLS_FAIL(GLSDET_E_ARG, "conv2d: output extent [%d,%d,%d] != expected [%d,%d,%d]", y.n, y.h, y.w, x.n, Ho, Wo);
  if (!d->w || !d->scale || !d->bias) GLS_FAIL(GLSDET_E_ARG, "conv2d: null weight/scale/bias");
  if (((uintptr_t)d->w | (uintptr_t)d->scale | (uintptr_t)d->bias) & 15)
    GLS_FAIL(GLSDET_E_ALIGN, "conv2d: weight/scale/bias must be 16-byte aligned");
  if (d->act < 0 || (d->act & 0xff) > 5 || (d->act & ~0x1ff)) GLS_FAIL(GLSDET_E_ARG, "conv2d: bad act %d", d->act);
  const bool has_res = d->res.base != nullptr;
  if (has_res) {
    if ((rc = check_view(d->res, "conv2d.res"))) return rc;
    if (!same_extent(d->res, y) || d->res.dtype != y.dtype)
      GLS_FAIL(GLSDET_E_ARG, "conv2d: residual must match the output extent and dtype");
  }
  const long M = (long)x.n * Ho * Wo;
  if (M <= 0 || M > 0x7fffffffL) GLS_FAIL(GLSDET_E_ARG, "conv2d: pixel count %ld out of range", M);

  a.x = (const unsigned char*)x.base;
  a.w = (const unsigned char*)d->w;
  a.scale = d->scale;
  a.bias = d->bias;
  a.y = (unsigned char*)y.base;
  a.res = has_res ? (const unsigned char*)d->res.base : nullptr;
  a.x_sn = x.sn; a.x_sh = x.sh; a.x_sw = x.sw;
  a.y_sn = y.sn; a.y_sh = y.sh; a.y_sw = y.sw;
  a.r_sn = d->res.sn; a.r_sh = d->res.sh; a.r_sw = d->res.sw;
  a.N = x.n; a.H = x.h; a.W = x.w; a.Cin = x.c;
  a.Ho = Ho; a.Wo = Wo; a.Cout = y.c; a.cout_pad = glsdet_conv_cout_pad(y.c);
  a.R = d->R; a.S = d->S; a.stride = d->stride; a.pad = d->pad;
  a.act = d->act & 0xff;
  a.act_post = 0;
  a.dbg = hint >= 0x10000 ? (hint >> 8) & 7 : 0;
  if ((d->act & GLSDET_ACT_RES_FIRST) && has_res) { a.act_post = a.act; a.act = GLSDET_ACT_NONE; }
  a.kreal = d->R * d->S * x.c;
  a.kpad = glsdet_conv_kpad(d->R, d->S, x.c, x.dtype);
  a.M = (int)M;
  gls_fastdiv(Ho * Wo, &a.howo_mul, &a.howo_sh);
  gls_fastdiv(Wo, &a.wo_mul, &a.wo_sh);
  a.n_co_tiles = a.n_px_tiles = 0;
  a.w2 = nullptr; a.scale2 = a.bias2 = nullptr; a.y2 = nullptr;
  a.y2_sn = a.y2_sh = a.y2_sw = 0;
  a.c2_0 = a.cin2 = a.cout2 = a.cout2_pad = a.kpad2 = a.act2 = a.y2_lin = 0;
  a.w2_bytes = 0;
  a.y_skip = 0;
  a.gn_part = nullptr; a.gn_cpg = a.gn_groups = 0;
  const int64_t xalloc = (const char*)x.alloc_hi - (const char*)x.alloc_lo;
  const int64_t wbytes = (int64_t)a.cout_pad * a.kpad * dtype_size(x.dtype);
  if (xalloc >= 0x7fffffffLL || wbytes >= 0x7fffffffLL)
    GLS_FAIL(GLSDET_E_ARG, "conv2d: operand allocation of %lld bytes exceeds the 2 GiB descriptor range", (long long)xalloc);
  a.x_lo = (const unsigned char*)x.alloc_lo;
  a.x_off = (unsigned)((const char*)x.base - (const char*)x.alloc_lo);
  a.x_bytes = (unsigned)xalloc;
  a.w_bytes = (unsigned)wbytes;
  a.x_lin = (d->R == 1 && d->S == 1 && d->stride == 1 && d->pad == 0 && x.sh == (int64_t)x.w * x.sw &&
             x.sn == (int64_t)x.h * x.sh) ? 1 : 0;
  a.y_lin = (y.sh == (int64_t)Wo * y.sw && y.sn == (int64_t)Ho * y.sh) ? 1 : 0;
  a.r_lin = (has_res && d->res.sh == (int64_t)Wo * d->res.sw && d->res.sn == (int64_t)Ho * d->res.sh) ? 1 : 0;

  const int xdt = x.dtype, ydt = y.dtype;
  *flops = 2.0 * (double)M * y.c * a.kreal;
  *bytes = (double)x.n * x.h * x.w * x.c * dtype_size(xdt) + (double)M * y.c * dtype_size(ydt) * (has_res ? 2 : 1) +
           (double)a.cout_pad * a.kpad * dtype_size(xdt);
  return 0;
}

// chained 1x1 (glsdet_conv2d_chain): validate, fill the argument block, add its work to the op's counters
static int add_chain(const glsdet_conv_desc* d, const glsdet_conv_chain* c, ConvArgs& a, double* flops, double* bytes) {
  if (!c->w2 || !c->scale2 || !c->bias2) GLS_FAIL(GLSDET_E_ARG, "conv2d_chain: null weight/scale/bias");
  if (((uintptr_t)c->w2 | (uintptr_t)c->scale2 | (uintptr_t)c->bias2) & 15) GLS_FAIL(GLSDET_E_ALIGN, "conv2d_chain: operands must be 16-byte aligned");
  int rc;
  if ((rc = check_view(c->y2, "conv2d_chain.y2"))) return rc;
  const glsdet_view& y = d->y;
  if (d->x.dtype != y.dtype || c->y2.dtype != y.dtype) GLS_FAIL(GLSDET_E_ARG, "conv2d_chain: x, y and y2 must share one dtype");
  if (c->y2.n != y.n || c->y2.h != y.h || c->y2.w != y.w || c->y2.c % 8) GLS_FAIL(GLSDET_E_ARG, "conv2d_chain: y2 must have y's pixel extent");
  const int es = dtype_size(y.dtype);
  if (c->cin2 < 1 || c->c0 < 0 || c->c0 + c->cin2 > y.c || (c->cin2 * es) % 32 || (c->c0 * es) % 16)
    GLS_FAIL(GLSDET_E_ARG, "conv2d_chain: input channel range [%d,%d) of y (%d channels) not usable", c->c0, c->c0 + c->cin2, y.c);
  if (c->act2 < 0 || c->act2 > 5) GLS_FAIL(GLSDET_E_ARG, "conv2d_chain: bad act2 %d", c->act2);
  a.cout2 = c->y2.c;
  a.cout2_pad = glsdet_conv_cout_pad(c->y2.c);
  if (a.cout2_pad > 128) GLS_FAIL(GLSDET_E_ARG, "conv2d_chain: at most 128 chained output channels");
  a.w2 = (const unsigned char*)c->w2;
  a.scale2 = c->scale2;
  a.bias2 = c->bias2;
  a.y2 = (unsigned char*)c->y2.base;
  a.y2_sn = c->y2.sn; a.y2_sh = c->y2.sh; a.y2_sw = c->y2.sw;
  a.c2_0 = c->c0; a.cin2 = c->cin2; a.act2 = c->act2;
  a.kpad2 = glsdet_conv_kpad(1, 1, c->cin2, y.dtype);
  a.w2_bytes = (unsigned)((int64_t)a.cout2_pad * a.kpad2 * es);
  a.y2_lin = (c->y2.sh == (int64_t)y.w * c->y2.sw && c->y2.sn == (int64_t)y.h * c->y2.sh) ? 1 : 0;
  *flops += 2.0 * (double)a.M * c->y2.c * c->cin2;
  *bytes += (double)a.M * c->y2.c * es + (double)a.cout2_pad * a.kpad2 * es;
  if (c->flags & ~GLSDET_CHAIN_SKIP_Y) GLS_FAIL(GLSDET_E_ARG, "conv2d_chain: unknown flags %d", c->flags);
  if (c->flags & GLSDET_CHAIN_SKIP_Y) {
    // y is not stored: legal only when the chained conv reads ALL of y (then one cout tile holds it) and nothing else is fused in
    if (c->c0 != 0 || c->cin2 != y.c || d->res.base) GLS_FAIL(GLSDET_E_ARG, "conv2d_chain: SKIP_Y needs c0 == 0, cin2 == y.c and no residual");
    a.y_skip = 1;
    *bytes -= (double)a.M * y.c * es;
  }
  return 0;
}
// the chained product runs in the workgroup whose cout tile holds its input channels: one tile must hold them all
static bool chain_fits(const ConvArgs& a, int co_t) {
  return a.w2 == nullptr || (a.c2_0 / co_t == (a.c2_0 + a.cin2 - 1) / co_t);
}

// validate the descriptor and build the op for `hint` (d->tile_hint is ignored here)
static int build_conv_op(const glsdet_conv_desc* d, int hint, OpRecord& op, const glsdet_conv_chain* chain = nullptr,
                         int gn_groups = 0, void* gn_stats = nullptr) {
  ConvArgs a;
  op.kind = 0;
  int rc = make_conv_args(d, hint, a, &op.flops, &op.bytes);
  if (rc) return rc;
  if (chain && (rc = add_chain(d, chain, a, &op.flops, &op.bytes))) return rc;
  if (gn_stats) {               // glsdet_conv2d_gnstats: GroupNorm partials of the stored output, halo ring kernels only
    const int vo = 16 / dtype_size(d->y.dtype);
    if (gn_groups < 1 || d->y.c % gn_groups || (d->y.c / gn_groups) % vo || 64 % (d->y.c / gn_groups) || ((uintptr_t)gn_stats & 7) ||
        d->x.dtype != d->y.dtype || d->res.base)
      GLS_FAIL(GLSDET_E_ARG, "conv2d_gnstats: need C %% groups == 0, whole 16-byte chunks per group, 8-byte aligned stats, no residual");
    a.gn_part = (double*)gn_stats;
    a.gn_groups = gn_groups;
    a.gn_cpg = d->y.c / gn_groups;
    if (hint < 8 || hint > 11) GLS_FAIL(GLSDET_E_ARG, "conv2d_gnstats: tile_hint must name a halo ring kernel (8..11)");
    if (conv_halo_try(a, d->x.dtype, d->y.dtype, hint, &op) == 0) {
      op.name += " +gn stats";
      return 0;
    }
    GLS_FAIL(GLSDET_E_ARG, "conv2d_gnstats: the halo ring kernel does not apply to this problem");
  }
  const glsdet_view &x = d->x, &y = d->y;
  const int xdt = x.dtype, ydt = y.dtype;
  // tile_hint: 0 auto, 1 generic kernel, 2 halo kernel, 4 halo kernel with wave-private weight staging,
  // 5 halo kernel with 64-row cout tiles also for wide layers,
  // 8 / 9 halo kernel with the weight tiles in an LDS-DMA ring (10 / 11: 64-byte channel chunks), 3 weight-stationary
  // 1x1 kernel (6 / 7 were the persistent LDS-DMA halo kernel, removed in round 2: 1.3-1.9x slower),
  // else co<<16|px (generic)
  if (hint == 3) {
    if (!a.w2 && conv1x1_ws_try(a, xdt, ydt, &op) == 0) return 0;
    GLS_FAIL(GLSDET_E_ARG, "conv2d: the weight-stationary 1x1 kernel does not apply to this problem");
  }
  if (hint == 6 || hint == 7) GLS_FAIL(GLSDET_E_ARG, "conv2d: tile_hint 6 / 7 (persistent LDS-DMA halo kernel) no longer exist");
  if (hint >= 16 && hint < 64 && !a.w2) {   // persistent LDS-DMA GEMM kernel for 1x1 convs (conv_gemm.hip), variant hint - 16
    if (conv_gemm_try(a, xdt, ydt, hint, &op) == 0) return 0;
    GLS_FAIL(GLSDET_E_ARG, "conv2d: the persistent 1x1 kernel (variant %d) does not apply to this problem", hint - 16);
  }
  if (conv_halo_try(a, xdt, ydt, hint, &op) == 0) return 0;
  if (hint == 2 || hint == 4 || hint == 5 || (hint >= 8 && hint <= 13) || (hint >= 0x100 && hint < 0x300))
    GLS_FAIL(GLSDET_E_ARG, "conv2d: the halo kernel does not apply to this problem");

  int co_t, px_t, kb;
  pick_tile(a, dtype_size(x.dtype), hint >= 0x10000 ? hint : 0, &co_t, &px_t, &kb);
  if (hint >= 0x10000 && co_t > 32 && a.cout_pad <= co_t / 2) GLS_FAIL(GLSDET_E_ARG, "conv2d: tile %dx%d is mostly padding here", co_t, px_t);
  if (co_t == 0 || (a.w2 && co_t == 32)) GLS_FAIL(GLSDET_E_ARG, "conv2d_chain: no tile of the generic kernel takes this chained problem");
  if (!chain_fits(a, co_t)) GLS_FAIL(GLSDET_E_ARG, "conv2d_chain: the chained input channels straddle two cout tiles of %d", co_t);
  // the chained form is compiled for K steps that stay inside one filter tap only (dispatch_tile: UT)
  if (a.w2 && (((long)a.Cin * dtype_size(xdt)) % kb != 0 || getenv("GLSDET_NO_UT") != nullptr))
    GLS_FAIL(GLSDET_E_ARG, "conv2d_chain: the generic kernel chains only when Cin is a whole number of K steps");
  char nm[112];
  snprintf(nm, sizeof nm, "conv_igemm<%s,%s,%dx%d,kb%d> %dx%d s%d cin%d cout%d%s", xdt ? "f32" : "f16",
           ydt ? "f32" : "f16", co_t, px_t, kb, d->R, d->S, d->stride, x.c, y.c, a.w2 ? " +1x1" : "");
  op.name = nm;
  op.launch = [a, co_t, px_t, kb, xdt, ydt](hipStream_t st) -> int {
    if (xdt == GLSDET_F16 && ydt == GLSDET_F16) return dispatch_tile<f16, f16>(a, co_t, px_t, kb, st);
    if (xdt == GLSDET_F16 && ydt == GLSDET_F32) return dispatch_tile<f16, float>(a, co_t, px_t, kb, st);
    return dispatch_tile<float, float>(a, co_t, px_t, kb, st);
  };
  return 0;
}

// The batched form: 9..GLS_BATCH problems whose argument blocks differ in the operand addresses only.
static int build_conv_batch_op(const glsdet_conv_desc* d, int32_t n, int hint, OpRecord& op) {
  if (hint && hint < 0x10000) GLS_FAIL(GLSDET_E_ARG, "conv2d_multi: more than %d descriptors run on the generic tiles only (hint %d)", GLS_MULTI, hint);
  ConvArgsB m = {};
  m.n = n;
  op.kind = 0;
  op.flops = op.bytes = 0;
  for (int i = 0; i < n; ++i) {
    ConvArgs a;
    memset((void*)&a, 0, sizeof a);             // (padding and the fields only launchers fill: compared below)
    double fl, by;
    int rc = make_conv_args(&d[i], hint, a, &fl, &by);
    if (rc) return rc;
    op.flops += fl;
    op.bytes += by;
    ConvPtrs& q = m.p[i];
    q.x = a.x; q.w = a.w; q.scale = a.scale; q.bias = a.bias; q.y = a.y; q.res = a.res;
    q.x_lo = a.x_lo; q.x_off = a.x_off; q.x_bytes = a.x_bytes; q.w_bytes = a.w_bytes; q._pad = 0;
    if (i == 0) { memcpy((void*)&m.base, &a, sizeof a); continue; }
    if (d[i].x.dtype != d[0].x.dtype || d[i].y.dtype != d[0].y.dtype || (a.res == nullptr) != (m.base.res == nullptr))
      GLS_FAIL(GLSDET_E_ARG, "conv2d_multi: descriptor %d is not of the shape class of descriptor 0", i);
    // everything but the addresses must be the same bytes
    ConvArgs t;
    memcpy((void*)&t, &a, sizeof a);
    const ConvArgs& b = m.base;
    t.x = b.x; t.w = b.w; t.scale = b.scale; t.bias = b.bias; t.y = b.y; t.res = b.res;
    t.x_lo = b.x_lo; t.x_off = b.x_off; t.x_bytes = b.x_bytes; t.w_bytes = b.w_bytes;
    if (memcmp(&t, &b, sizeof(ConvArgs)) != 0)
      GLS_FAIL(GLSDET_E_ARG, "conv2d_multi: more than %d descriptors must share one geometry (sizes, strides, epilogue); descriptor %d differs",
               GLS_MULTI, i);
  }
  if (m.base.w2 || m.base.gn_part) GLS_FAIL(GLSDET_E_ARG, "conv2d_multi: no chained / GroupNorm form");
  int co_t, px_t, kb;
  ConvArgs probe = m.base;
  const long Mtot = (long)m.base.M * n;
  probe.M = (int)(Mtot > 0x7fffffffL ? 0x7fffffffL : Mtot);
  pick_tile(probe, dtype_size(d[0].x.dtype), hint >= 0x10000 ? hint : 0, &co_t, &px_t, &kb);
  if (co_t == 32) { co_t = 64; px_t = 64; }
  if (px_t == 256) { co_t = 128; px_t = 128; }
  const int xdt = d[0].x.dtype, ydt = d[0].y.dtype;
  char nm[112];
  snprintf(nm, sizeof nm, "conv_igemm_batch[%d]<%s,%s,%dx%d,kb%d> %dx%d s%d cin%d cout%d", n, xdt ? "f32" : "f16",
           ydt ? "f32" : "f16", co_t, px_t, kb, d[0].R, d[0].S, d[0].stride, d[0].x.c, d[0].y.c);
  op.name = nm;
  op.launch = [m, co_t, px_t, kb, xdt, ydt](hipStream_t st) -> int {
    if (xdt == GLSDET_F16 && ydt == GLSDET_F16) return dispatch_tile_batch<f16, f16>(m, co_t, px_t, kb, st);
    if (xdt == GLSDET_F16 && ydt == GLSDET_F32) return dispatch_tile_batch<f16, float>(m, co_t, px_t, kb, st);
    return dispatch_tile_batch<float, float>(m, co_t, px_t, kb, st);
  };
  return 0;
}

static int build_conv_multi_op(const glsdet_conv_desc* d, int32_t n, int hint, OpRecord& op) {
  if (d && n > GLS_MULTI && n <= GLS_BATCH) return build_conv_batch_op(d, n, hint, op);
  if (!d || n < 1 || n > GLS_MULTI) GLS_FAIL(GLSDET_E_ARG, "conv2d_multi: 1..%d descriptors (up to %d of one geometry)", GLS_MULTI, GLS_BATCH);
  ConvArgsN m = {};
  m.n = n;
  op.kind = 0;
  op.flops = op.bytes = 0;
  long Mtot = 0;
  for (int i = 0; i < n; ++i) {
    double fl, by;
    int rc = make_conv_args(&d[i], hint, m.p[i], &fl, &by);
    if (rc) return rc;
    op.flops += fl;
    op.bytes += by;
    Mtot += m.p[i].M;
    const ConvArgs &a = m.p[i], &b = m.p[0];
    if (d[i].x.dtype != d[0].x.dtype || d[i].y.dtype != d[0].y.dtype || a.R != b.R || a.S != b.S || a.stride != b.stride ||
        a.pad != b.pad || a.Cin != b.Cin || a.Cout != b.Cout)
      GLS_FAIL(GLSDET_E_ARG, "conv2d_multi: descriptor %d is not of the shape class of descriptor 0", i);
  }
  if ((hint >= 8 && hint <= 11) || (hint >= 0x100 && hint < 0x300)) {          // the grouped ring kernel (conv_halo.hip; bits 8..9: tile geometry); no silent fall-back: the tuner asks per hint
    if (conv_halo_multi_try(m, d[0].x.dtype, d[0].y.dtype, hint, &op))
      GLS_FAIL(GLSDET_E_ARG, "conv2d_multi: the grouped halo kernel (hint %d) does not apply to this shape class", hint);
    return 0;
  }
  int co_t, px_t, kb;
  ConvArgs probe = m.p[0];
  probe.M = (int)(Mtot > 0x7fffffffL ? 0x7fffffffL : Mtot);
  pick_tile(probe, dtype_size(d[0].x.dtype), hint >= 0x10000 ? hint : 0, &co_t, &px_t, &kb);
  if (co_t == 32) { co_t = 64; px_t = 64; }
  const int xdt = d[0].x.dtype, ydt = d[0].y.dtype;
  char nm[112];
  snprintf(nm, sizeof nm, "conv_igemm_multi[%d]<%s,%s,%dx%d,kb%d> %dx%d s%d cin%d cout%d", n, xdt ? "f32" : "f16",
           ydt ? "f32" : "f16", co_t, px_t, kb, d[0].R, d[0].S, d[0].stride, d[0].x.c, d[0].y.c);
  op.name = nm;
  op.launch = [m, co_t, px_t, kb, xdt, ydt](hipStream_t st) -> int {
    if (xdt == GLSDET_F16 && ydt == GLSDET_F16) return dispatch_tile_multi<f16, f16>(m, co_t, px_t, kb, st);
    if (xdt == GLSDET_F16 && ydt == GLSDET_F32) return dispatch_tile_multi<f16, float>(m, co_t, px_t, kb, st);
    return dispatch_tile_multi<float, float>(m, co_t, px_t, kb, st);
  };
  return 0;
}

extern "C" int glsdet_conv2d_multi(const glsdet_conv_desc* d, int32_t n, void* stream) {
  OpRecord op;
  int rc = build_conv_multi_op(d, n, d ? d[0].tile_hint : 0, op);
  if (rc) return rc;
  return submit(std::move(op), stream);
}

// Times every candidate on the device and reports the fastest.  Each candidate: one warm launch
// (module load, function attributes), then TRIALS rounds over ALL candidates of REPS back-to-back
// launches; a candidate's time is the MINIMUM over its rounds.  (One round of five launches per
// candidate let a single hiccup -- a 20 us kernel measured at 2 ms now and then -- decide the variant
// for the life of the plan; interleaving the rounds also spreads clock drift over all candidates.)
static int time_variants(std::vector<OpRecord>& ops, const std::vector<int>& ids, hipStream_t st, int* best_id,
                         float* best_us, int* any) {
  constexpr int TRIALS = 3, REPS = 5;
  hipEvent_t e0, e1;
  GLS_HIP(hipEventCreate(&e0));
  GLS_HIP(hipEventCreate(&e1));
  std::vector<float> us(ops.size(), 1e30f);
  std::vector<char> dead(ops.size(), 0);
  for (size_t i = 0; i < ops.size(); ++i)
    if (ops[i].launch(st)) dead[i] = 1;
  for (int t = 0; t < TRIALS; ++t) {
    for (size_t i = 0; i < ops.size(); ++i) {
      if (dead[i]) continue;
      int rc = 0;
      (void)hipEventRecord(e0, st);
      for (int r = 0; r < REPS && !rc; ++r) rc = ops[i].launch(st);
      (void)hipEventRecord(e1, st);
      if (hipEventSynchronize(e1) != hipSuccess || rc) { dead[i] = 1; continue; }
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, e0, e1);
      const float u = ms * 1000.f / REPS;
      if (u < us[i]) us[i] = u;
    }
  }
  // play-off: candidates within 4 % of the fastest are measured again, longer (near-ties used to flip between runs and
  // moved a whole workload by up to 1.5 %: the same build picked ring vs ring_k64 for the tower convs on two boxes)
  float lo = 1e30f;
  for (size_t i = 0; i < ops.size(); ++i)
    if (!dead[i] && us[i] < lo) lo = us[i];
  std::vector<size_t> close;
  for (size_t i = 0; i < ops.size(); ++i)
    if (!dead[i] && us[i] <= 1.04f * lo) close.push_back(i);
  if (close.size() > 1) {
    constexpr int REPS2 = 12;
    for (size_t i : close) us[i] = 1e30f;
    for (int t = 0; t < 4; ++t)
      for (size_t i : close) {
        if (dead[i]) continue;
        int rc = 0;
        (void)hipEventRecord(e0, st);
        for (int r = 0; r < REPS2 && !rc; ++r) rc = ops[i].launch(st);
        (void)hipEventRecord(e1, st);
        if (hipEventSynchronize(e1) != hipSuccess || rc) { dead[i] = 1; continue; }
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const float u = ms * 1000.f / REPS2;
        if (u < us[i]) us[i] = u;
      }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *any = 0;
  for (size_t i = 0; i < ops.size(); ++i) {
    if (dead[i]) continue;
    *any = 1;
    if (us[i] < *best_us) { *best_us = us[i]; *best_id = ids[i]; }
  }
  return 0;
}

extern "C" int glsdet_conv2d_multi_tune(const glsdet_conv_desc* d, int32_t n, void* stream, int32_t* best_hint,
                                        float* best_us) {
  if (!d || !best_hint) GLS_FAIL(GLSDET_E_ARG, "conv2d_multi_tune: null argument");
  hipStream_t st = (hipStream_t)stream;
  const int hints[] = {(128 << 16) | 128, (64 << 16) | 128, (64 << 16) | 64, (64 << 16) | 64 | 0x8000,
                       (64 << 16) | 128 | 0x8000, (128 << 16) | 128 | 0x8000, 8, 9, 10, 11};
  std::vector<OpRecord> ops;
  std::vector<int> ids;
  const bool no_ring = getenv("GLSDET_NO_RING_MULTI") != nullptr;       // A/B switch for measurements
  std::vector<int> cand(hints, hints + sizeof(hints) / sizeof(hints[0]));
  if (!getenv("GLSDET_NO_TILE_GEO") && (d[0].stride == 1 || d[0].stride == 2) && d[0].R == 3 && !no_ring) {
    for (int geo = 1; geo <= 2; ++geo) {             // tile geometries with >= 3 % fewer tiles over the whole group
      int th, tw;
      tile_geo_dims(geo, &th, &tw);
      long t0 = 0, t1 = 0;
      for (int i = 0; i < n; ++i) {
        t0 += (long)((d[i].y.h + 7) / 8) * ((d[i].y.w + 15) / 16);
        t1 += (long)((d[i].y.h + th - 1) / th) * ((d[i].y.w + tw - 1) / tw);
      }
      if (t1 * 100 <= t0 * 97)
        for (int h : {8, 9, 10, 11})
          if (d[0].stride == 1 || h >= 10) cand.push_back((geo << 8) | h);
    }
  }
  for (int h : cand) {
    OpRecord op;
    if (no_ring && (h & 0xff) >= 8 && (h & 0xff) <= 11 && h < 0x10000) continue;
    if (build_conv_multi_op(d, n, h, op)) continue;
    ops.push_back(std::move(op));
    ids.push_back(h);
  }
  float best = 1e30f;
  int bh = 0, any = 0;
  int rc = time_variants(ops, ids, st, &bh, &best, &any);
  if (rc) return rc;
  if (!any) GLS_FAIL(GLSDET_E_ARG, "conv2d_multi_tune: no variant applies");
  *best_hint = bh;
  if (best_us) *best_us = best;
  set_error("");
  return 0;
}

extern "C" int glsdet_conv2d(const glsdet_conv_desc* d, void* stream) {
  OpRecord op;
  int rc = build_conv_op(d, d ? d->tile_hint : 0, op);
  if (rc) return rc;
  return submit(std::move(op), stream);
}

extern "C" int glsdet_conv2d_chain(const glsdet_conv_desc* d, const glsdet_conv_chain* c, void* stream) {
  if (!d || !c) GLS_FAIL(GLSDET_E_ARG, "conv2d_chain: null argument");
  OpRecord op;
  int rc = build_conv_op(d, d->tile_hint, op, c);
  if (rc) return rc;
  return submit(std::move(op), stream);
}

// Bottleneck front: the 1x1 (c1) recomputed on the halo of the 3x3 (c2) in one launch (conv_bneck.hip)
static int build_bneck_op(const glsdet_conv_desc* c1, const glsdet_conv_desc* c2, int hint, OpRecord& op) {
  if (!c1 || !c2) GLS_FAIL(GLSDET_E_ARG, "bottleneck: null descriptor");
  BneckArgs b;
  ConvArgs a1;
  double f1, b1, f2, b2;
  int rc;
  if ((rc = make_conv_args(c1, 0, a1, &f1, &b1))) return rc;
  if ((rc = make_conv_args(c2, 0, b.c, &f2, &b2))) return rc;
  const glsdet_view &m1 = c1->y, &m2 = c2->x;
  if (c1->R != 1 || c1->S != 1 || c1->stride != 1 || c1->pad != 0 || c1->res.base || (c1->act & GLSDET_ACT_RES_FIRST))
    GLS_FAIL(GLSDET_E_ARG, "bottleneck: the first conv must be a plain 1x1 (stride 1, no padding, no residual)");
  if (m1.n != m2.n || m1.h != m2.h || m1.w != m2.w || m1.c != m2.c || m1.dtype != m2.dtype)
    GLS_FAIL(GLSDET_E_ARG, "bottleneck: c1.y and c2.x must describe the same hidden tensor");
  if (c1->x.dtype != m1.dtype || c2->y.dtype != m1.dtype) GLS_FAIL(GLSDET_E_ARG, "bottleneck: x, the hidden tensor and y share one dtype");
  // in place is impossible: a workgroup reads the halo of x while its neighbours store their tiles of y
  const char *xlo = (const char*)c1->x.base, *ylo = (const char*)c2->y.base;
  const int es = dtype_size(m1.dtype);
  const int64_t xspan = ((int64_t)(c1->x.n - 1) * c1->x.sn + (int64_t)(c1->x.h - 1) * c1->x.sh + (int64_t)(c1->x.w - 1) * c1->x.sw + c1->x.c) * es;
  const int64_t yspan = ((int64_t)(c2->y.n - 1) * c2->y.sn + (int64_t)(c2->y.h - 1) * c2->y.sh + (int64_t)(c2->y.w - 1) * c2->y.sw + c2->y.c) * es;
  if (xlo < ylo + yspan && ylo < xlo + xspan) {
    // overlapping address ranges are fine only for disjoint channel slices of one pixel-interleaved buffer
    const bool same_geom = c1->x.sn == c2->y.sn && c1->x.sh == c2->y.sh && c1->x.sw == c2->y.sw;
    const int64_t d = (ylo - xlo) / es, sw = c1->x.sw;
    const bool disjoint = same_geom && (d > 0 ? (d >= c1->x.c && d + c2->y.c <= sw) : (d < 0 && -d >= c2->y.c && -d + c1->x.c <= sw));
    if (!disjoint) GLS_FAIL(GLSDET_E_ARG, "bottleneck: y overlaps x (the fused form cannot run in place)");
  }
  ConvArgs& a = b.c;
  a.x = a1.x; a.x_sn = a1.x_sn; a.x_sh = a1.x_sh; a.x_sw = a1.x_sw;
  a.x_lo = a1.x_lo; a.x_off = a1.x_off; a.x_bytes = a1.x_bytes; a.x_lin = 0;
  b.w0 = a1.w; b.scale0 = a1.scale; b.bias0 = a1.bias;
  b.cin0 = a1.Cin; b.kpad0 = a1.kpad; b.act0 = a1.act; b.w0_bytes = a1.w_bytes;
  if (glsdet_conv_cout_pad(m1.c) != m1.c) GLS_FAIL(GLSDET_E_ARG, "bottleneck: hidden channels must be a multiple of 32");
  op.kind = 0;
  op.flops = f1 + f2;
  op.bytes = b1 + b2 - 2.0 * (double)a.M * m1.c * es;      // the hidden tensor is neither written nor read
  if (conv_bneck_try(b, m1.dtype, hint, &op)) GLS_FAIL(GLSDET_E_ARG, "bottleneck: the fused kernel does not apply to this problem");
  return 0;
}
extern "C" int glsdet_bottleneck(const glsdet_conv_desc* c1, const glsdet_conv_desc* c2, int32_t hint, void* stream) {
  OpRecord op;
  int rc = build_bneck_op(c1, c2, hint, op);
  if (rc) return rc;
  return submit(std::move(op), stream);
}
extern "C" int glsdet_bottleneck_tune(const glsdet_conv_desc* c1, const glsdet_conv_desc* c2, void* stream, int32_t* best_hint,
                                      float* best_us) {
  if (!best_hint) GLS_FAIL(GLSDET_E_ARG, "bottleneck_tune: null argument");
  std::vector<OpRecord> ops;
  std::vector<int> ids;
  for (int h : {0, 1}) {
    OpRecord op;
    if (build_bneck_op(c1, c2, h, op)) continue;
    if (h == 1 && !ops.empty() && ops[0].name == op.name) continue;    // hint 1 fell back to the same kernel
    ops.push_back(std::move(op));
    ids.push_back(h);
  }
  float best = 1e30f;
  int bh = 0, any = 0;
  int rc = time_variants(ops, ids, (hipStream_t)stream, &bh, &best, &any);
  if (rc) return rc;
  if (!any) GLS_FAIL(GLSDET_E_ARG, "bottleneck_tune: the fused kernel does not apply");
  *best_hint = bh;
  if (best_us) *best_us = best;
  set_error("");
  return 0;
}

// conv + GroupNorm partials of its output (the tower convs of gfl_head.py:128-152: conv -> GN -> ReLU): the statistics
// pass of glsdet_groupnorm reads the tensor once more only to sum it; here the conv's store phase sums what it stores.
extern "C" int64_t glsdet_conv2d_gnstats_bytes(int32_t n, int32_t ho, int32_t wo, int32_t groups) {
  if (n < 1 || ho < 1 || wo < 1 || groups < 1) return 0;
  return (int64_t)n * ((ho + 7) / 8) * ((wo + 15) / 16) * 4 * groups * 2 * (int64_t)sizeof(double);
}
extern "C" int glsdet_conv2d_gnstats(const glsdet_conv_desc* d, int32_t groups, void* stats, void* stream) {
  if (!d || !stats) GLS_FAIL(GLSDET_E_ARG, "conv2d_gnstats: null argument");
  OpRecord op;
  int rc = build_conv_op(d, d->tile_hint ? d->tile_hint : 8, op, nullptr, groups, stats);
  if (rc) return rc;
  return submit(std::move(op), stream);
}
extern "C" int glsdet_conv2d_gnstats_tune(const glsdet_conv_desc* d, int32_t groups, void* stats, void* stream, int32_t* best_hint,
                                          float* best_us) {
  if (!d || !stats || !best_hint) GLS_FAIL(GLSDET_E_ARG, "conv2d_gnstats_tune: null argument");
  std::vector<OpRecord> ops;
  std::vector<int> ids;
  for (int h : {8, 9, 10, 11}) {
    OpRecord op;
    if (build_conv_op(d, h, op, nullptr, groups, stats)) continue;
    ops.push_back(std::move(op));
    ids.push_back(h);
  }
  float best = 1e30f;
  int bh = 0, any = 0;
  int rc = time_variants(ops, ids, (hipStream_t)stream, &bh, &best, &any);
  if (rc) return rc;
  if (!any) GLS_FAIL(GLSDET_E_ARG, "conv2d_gnstats_tune: no halo ring kernel applies");
  *best_hint = bh;
  if (best_us) *best_us = best;
  set_error("");
  return 0;
}

// Measure every kernel/tile variant that applies to this exact problem on the device (its
// real buffers; launches immediately, never recorded) and report the fastest hint.  Build-time
// only: it synchronises.  The conv writes its real output, so callers tune before the first
// real run (contents are overwritten by it).
static int conv_tune(const glsdet_conv_desc* d, const glsdet_conv_chain* c, void* stream, int32_t* best_hint, float* best_us);
extern "C" int glsdet_conv2d_tune(const glsdet_conv_desc* d, void* stream, int32_t* best_hint, float* best_us) {
  return conv_tune(d, nullptr, stream, best_hint, best_us);
}
extern "C" int glsdet_conv2d_chain_tune(const glsdet_conv_desc* d, const glsdet_conv_chain* c, void* stream, int32_t* best_hint,
                                        float* best_us) {
  if (!c) GLS_FAIL(GLSDET_E_ARG, "conv2d_chain_tune: null argument");
  return conv_tune(d, c, stream, best_hint, best_us);
}
static int conv_tune(const glsdet_conv_desc* d, const glsdet_conv_chain* c, void* stream, int32_t* best_hint, float* best_us) {
  if (!d || !best_hint) GLS_FAIL(GLSDET_E_ARG, "conv2d_tune: null argument");
  hipStream_t st = (hipStream_t)stream;
  // (6 / 7, the persistent LDS-DMA halo kernel, is not offered: slower than 8 / 9 on every layer measured)
  const int hints[] = {2, 4, 5, 8, 9, 10, 11, 12, 13, 3, 16, 20, 22, 24, 25, 29, (128 << 16) | 128, (64 << 16) | 128, (64 << 16) | 64, (32 << 16) | 128,
                       (64 << 16) | 64 | 0x8000, (64 << 16) | 128 | 0x8000, (128 << 16) | 128 | 0x8000, (128 << 16),
                       (128 << 16) | 0x8000};
  std::vector<OpRecord> ops;
  std::vector<int> ids;
  std::vector<int> cand(hints, hints + sizeof(hints) / sizeof(hints[0]));
  if (!c && (d->stride == 1 || (d->stride == 2 && d->R == 3)) && d->R == d->S && d->R >= 3 && !getenv("GLSDET_NO_TILE_GEO")) {
    // other tile geometries of the ring kernels (conv_common.h TileGeo): offered where they need >= 3 % fewer tiles
    const int Ho = d->y.h, Wo = d->y.w;
    auto tiles = [&](int th, int tw) { return (long)((Ho + th - 1) / th) * ((Wo + tw - 1) / tw); };
    for (int geo = 1; geo <= 2; ++geo) {
      int th, tw;
      tile_geo_dims(geo, &th, &tw);
      if (tiles(th, tw) * 100 <= tiles(8, 16) * 97)
        for (int h : {8, 9, 10, 11})
          if (d->stride == 1 || h >= 10) cand.push_back((geo << 8) | h);
      tile_geo8_dims(geo, &th, &tw);
      if (d->stride == 1 && tiles(th, tw) * 100 <= tiles(8, 32) * 97) cand.push_back((geo << 8) | 13);
    }
  }
  for (int h : cand) {
    OpRecord op;
    if (h >= 0x10000 && (h >> 16) == 128 && (h & 0xff) == 0 && (c || d->y.c <= 64)) continue;      // the eight-wave tile: wide layers, no chain
    if (build_conv_op(d, h, op, c)) continue;          // variant does not apply
    if (h >= 0x10000 && (h >> 16) == 32 && d->y.c > 32) continue;
    ops.push_back(std::move(op));
    ids.push_back(h);
  }
  float best = 1e30f;
  int bh = 0, any = 0;
  int rc = time_variants(ops, ids, st, &bh, &best, &any);
  if (rc) return rc;
  if (!any) GLS_FAIL(GLSDET_E_ARG, "conv2d_tune: no variant applies");
  *best_hint = bh;
  if (best_us) *best_us = best;
  set_error("");
  return 0;
}
