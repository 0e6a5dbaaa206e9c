// 1x1 convolution, weight stationary, persistent.
//
// 1x1 layers of this network are HBM/MALL bound (arithmetic intensity <= Cin/2 FLOP/B with
// Cin <= 256).  The generic kernel re-reads the input once per cout tile and runs load ->
// compute -> store strictly in sequence inside a workgroup.  Here a workgroup
//   * keeps its whole [CO_T x Cin] weight tile resident in LDS for its lifetime,
//   * walks pixel tiles of 64 in a persistent loop, treating (tile, 128-byte k chunk) as one
//     stream of chunks: the loads of chunk c+2 are issued before chunk c is multiplied,
//     chunk c+1 is written to the other LDS buffer after it -> input bytes are always in
//     flight, also across the epilogue of the previous tile,
//   * reads every input byte once per cout tile of 128 (one tile for Cout <= 128).
// Fragment layout / epilogue as in conv.hip.
#include <stdlib.h>

#include "conv_common.h"

namespace glsdet {

template <typename T, int CO_T>
__global__ __launch_bounds__(256) void conv1x1_ws_kernel(const ConvArgs a, const int n_px_tiles, const int a_rs) {
  constexpr int PX_T = 64, KB = 128, RS = KB + 16;
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int KE = KB / (int)sizeof(T);
  constexpr int NB = PX_T * 8 / 256;                  // 2 chunks of 16 B per thread per k chunk
  constexpr int WT_CO = CO_T / 2, TM = WT_CO / 32;    // waves: 2 (cout) x 2 (pixels), 32 px each
  constexpr int ORS = CO_T * (int)sizeof(T) + 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sA = smem;                                          // [CO_T][a_rs]
  unsigned char* sB = smem + CO_T * a_rs;                            // [2][PX_T][RS]
  unsigned char* sE = sB + 2 * PX_T * RS;                            // [PX_T][ORS]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool wide = sizeof(T) == 2 && a.res != nullptr;              // fp32 staging of the output tile (conv_common.h)
  const auto xrs = gls_make_rsrc(a.x_lo, a.x_bytes);
  const auto wrs = gls_make_rsrc(a.w, a.w_bytes);
  const int n_co = a.n_co_tiles;
  const int co_tile = blockIdx.x % n_co;
  const int co0 = co_tile * CO_T;
  const int first = blockIdx.x / n_co, stride = gridDim.x / n_co;    // my pixel tiles: first, first+stride, ...
  const int nk = a.Cin / KE;                                         // k chunks per tile
  const int kbytes = a.Cin * (int)sizeof(T);

  // ---- resident weight tile
  {
    const int cpr = kbytes / 16;
    for (int q = tid; q < CO_T * cpr; q += 256) {
      const int row = q / cpr, c = q - row * cpr;
      const unsigned off = (co0 + row) < a.cout_pad ? (unsigned)(((co0 + row) * a.kpad) * (int)sizeof(T) + c * 16) : GLS_OOB;
      *reinterpret_cast<u32x4*>(sA + row * a_rs + c * 16) = gls_buf_load16(wrs, off);
    }
  }

  const int kc = tid & 7, row0 = tid >> 3;            // rows row0, row0 + 32
  const int HoWo = a.Ho * a.Wo;
  // issue-side state (runs two chunks ahead of the compute side)
  int it_tile = first, it_k = 0;
  unsigned it_off[NB];
  auto set_issue_rows = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int p = it_tile * PX_T + row0 + i * 32;
      if (it_tile < n_px_tiles && p < a.M) {
        const int n = gls_div(p, a.howo_mul, a.howo_sh), rem = p - n * HoWo;
        const int ho = gls_div(rem, a.wo_mul, a.wo_sh), wo = rem - ho * a.Wo;
        it_off[i] = a.x_off + (unsigned)(((long)n * a.x_sn + (long)ho * a.x_sh + (long)wo * a.x_sw + kc * VEC) * (long)sizeof(T));
      } else {
        it_off[i] = GLS_OOB;
      }
    }
  };
  auto issue = [&](u32x4 (&r)[NB]) __attribute__((always_inline)) {
    const unsigned koff = (unsigned)(it_k * KB);
#pragma unroll
    for (int i = 0; i < NB; ++i) r[i] = gls_buf_load16(xrs, it_off[i] + koff);
    if (++it_k == nk) {
      it_k = 0;
      it_tile += stride;
      set_issue_rows();
    }
  };
  auto stash = [&](int buf, const u32x4 (&r)[NB]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NB; ++i)
      *reinterpret_cast<u32x4*>(sB + buf * (PX_T * RS) + (row0 + i * 32) * RS + kc * 16) = r[i];
  };

  const int wco = wave & 1, wpx = wave >> 1;
  const int l31 = lane & 31, lh = lane >> 5;
  const int a_off = (wco * WT_CO + l31) * a_rs + lh * 16;
  const int b_off = (wpx * 32 + l31) * RS + lh * 16;

  f32x16 acc[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.0f;
  // folded-BN scale / bias of this lane's 4 x 4 output channels per 32-row block: loaded once
  f32x4 scv[TM][4], biv[TM][4];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int co_l = wco * WT_CO + i * 32 + 8 * g + 4 * lh;
      scv[i][g] = biv[i][g] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (co0 + co_l < a.cout_pad) {
        scv[i][g] = *reinterpret_cast<const f32x4*>(a.scale + co0 + co_l);
        biv[i][g] = *reinterpret_cast<const f32x4*>(a.bias + co0 + co_l);
      }
    }

  int my_tiles = 0;
  for (int t = first; t < n_px_tiles; t += stride) ++my_tiles;
  const int total = my_tiles * nk;
  if (total == 0) return;

  // register ring of 4 chunk sets: chunks c+1 .. c+4 are in flight while chunk c is multiplied
  u32x4 r0[NB], r1[NB], r2[NB], r3[NB];
  set_issue_rows();
  issue(r0);                    // chunk 0
  stash(0, r0);
  issue(r1);                    // chunks 1..3 (past the end: read zeros)
  issue(r2);
  issue(r3);
  __syncthreads();

  int ck = 0, ctile = first;    // compute-side position
  // residual chunks of the tile being multiplied: requested when its LAST k chunk starts (before that step's input loads, so
  // that waiting for them leaves the newer input loads in flight) and consumed after the staging barrier -- one batched
  // round trip in the shadow of the MFMAs and the staging instead of one dependent trip per chunk in the store loop
  // (the 64 -> 256 / 128 -> 512 / 256 -> 1024 expansions of ResNet move 4x more residual + output bytes than input)
  constexpr int OCPR = CO_T / VEC, NQ = PX_T * OCPR / 256;
  u32x4 rres[NQ];
  long ryo[NQ];
  const int e_cq = tid % OCPR, e_px0 = tid / OCPR;      // my output chunks: pixels e_px0 + j * (256 / OCPR), channel chunk e_cq
  const bool e_cok = co0 + e_cq * VEC < a.Cout;
  auto issue_res = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      const int p = ctile * PX_T + e_px0 + j * (256 / OCPR);
      rres[j] = u32x4{0u, 0u, 0u, 0u};
      ryo[j] = -1;
      if (p < a.M && e_cok) {
        const int co = co0 + e_cq * VEC;
        ryo[j] = gls_pix_off(p, HoWo, a.Wo, a.y_sn, a.y_sh, a.y_sw, a.y_lin, a) + co;
        if (a.res) {
          const long ro = gls_pix_off(p, HoWo, a.Wo, a.r_sn, a.r_sh, a.r_sw, a.r_lin, a) + co;
          rres[j] = *reinterpret_cast<const u32x4*>(a.res + ro * (long)sizeof(T));
        }
      }
    }
  };
  auto step = [&](int c, u32x4 (&nxt)[NB], u32x4 (&freeset)[NB]) __attribute__((always_inline)) {
    // nxt holds chunk c+1; freeset held chunk c (already in LDS): chunk c+4 goes there
    if (ck == nk - 1) issue_res();
    issue(freeset);
    const unsigned char* bbuf = sB + (c & 1) * (PX_T * RS);
    const int kbase = ck * KB;
#pragma unroll
    for (int kk = 0; kk < KB / 32; ++kk) {
      u32x4 af[TM];
      const u32x4 bf = *reinterpret_cast<const u32x4*>(bbuf + b_off + kk * 32);
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const u32x4*>(sA + a_off + i * 32 * a_rs + kbase + kk * 32);
#pragma unroll
      for (int i = 0; i < TM; ++i) MMA<T>::run(af[i], bf, acc[i]);
    }
    stash((c + 1) & 1, nxt);
    if (++ck == nk) {           // tile finished: epilogue (loads of the next chunks stay in flight)
      ck = 0;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int co_l = wco * WT_CO + i * 32 + 8 * g + 4 * lh;
          const f32x4 sc = scv[i][g], bi = biv[i][g];
          const f32x4 xv = {acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]};
          const f32x4 yv = scale_bias_act4<T>(xv, sc, bi, a.act);
          const float v[4] = {yv[0], yv[1], yv[2], yv[3]};
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][4 * g + e] = 0.0f;
          stage4<T, CO_T>(sE, wpx * 32 + l31, co_l, v, wide);
        }
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < NQ; ++j) {
        const int px_l = e_px0 + j * (256 / OCPR);
        if (ryo[j] >= 0) {
          u32x4 v;
          if (wide) {                      // fp32 staging: residual add in fp32, one rounding
            constexpr int ORSW = CO_T * 4 + 16;
            const f32x4 lo = *reinterpret_cast<const f32x4*>(sE + px_l * ORSW + e_cq * 32);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(sE + px_l * ORSW + e_cq * 32 + 16);
            v = add_chunk_wide(lo, hi, rres[j], a.act_post);
          } else {
            v = *reinterpret_cast<const u32x4*>(sE + px_l * ORS + e_cq * 16);
            if (a.res) v = add_chunk(v, rres[j], (T*)nullptr, a.act_post);
          }
          *reinterpret_cast<u32x4*>(a.y + ryo[j] * (long)sizeof(T)) = v;
        }
      }
      ctile += stride;
    }
    __syncthreads();
  };
  for (int c = 0; c < total; c += 4) {
    step(c, r1, r0);
    if (c + 1 < total) step(c + 1, r2, r1);
    if (c + 2 < total) step(c + 2, r3, r2);
    if (c + 3 < total) step(c + 3, r0, r3);
  }
}

template <typename T, int CO_T>
static int launch_ws(const ConvArgs& a, hipStream_t st) {
  const int kbytes = a.Cin * (int)sizeof(T);
  const int a_rs = kbytes + 16;
  const int lds = CO_T * a_rs + 2 * 64 * 144 + epi_bytes<T>(CO_T, 64, a.res != nullptr);
  auto kern = conv1x1_ws_kernel<T, CO_T>;
  static int attr_lds = 0;
  if (lds > 64 * 1024 && lds > attr_lds) {
    GLS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_lds = lds;
  }
  ConvArgs b = a;
  b.n_co_tiles = (a.Cout + CO_T - 1) / CO_T;
  const int n_px_tiles = (a.M + 63) / 64;
  // workgroups resident per CU (LDS bound; 160 KiB per CU): each keeps ~3 chunks of 8 KB in flight, and an HBM-bound layer
  // wants ~50 KB per CU outstanding (6 TB/s x ~2 us / 256 CUs)
  static const int per_cu_env = getenv("GLSDET_WS_PER_CU") ? atoi(getenv("GLSDET_WS_PER_CU")) : 0;
  int per_cu = (160 * 1024) / (lds + 1024);
  if (per_cu > 6) per_cu = 6;
  if (per_cu < 1) per_cu = 1;
  if (per_cu_env > 0 && per_cu_env < per_cu) per_cu = per_cu_env;
  int groups = 256 * per_cu / b.n_co_tiles;          // pixel-tile walkers per cout tile
  if (groups > n_px_tiles) groups = n_px_tiles;
  if (groups < 1) groups = 1;
  hipLaunchKernelGGL(kern, dim3((unsigned)(groups * b.n_co_tiles)), dim3(256), lds, st, b, n_px_tiles, a_rs);
  GLS_HIP(hipGetLastError());
  return 0;
}

// hint 3 = this kernel.  Returns 1 when it does not apply.
int conv1x1_ws_try(const ConvArgs& a, int xdt, int ydt, OpRecord* op) {
  if (a.R != 1 || a.S != 1 || a.stride != 1 || a.pad != 0 || xdt != ydt) return 1;
  const int es = dtype_size(xdt);
  const int kbytes = a.Cin * es;
  if (kbytes % 128 || kbytes > 512) return 1;          // resident weight tile <= 128 x 528 B
  const int co_t = a.cout_pad <= 64 ? 64 : 128;
  char nm[96];
  snprintf(nm, sizeof nm, "conv1x1_ws<%s,%dx64> cin%d cout%d", xdt ? "f32" : "f16", co_t, a.Cin, a.Cout);
  op->name = nm;
  op->launch = [a, co_t, xdt](hipStream_t st) -> int {
    if (xdt == GLSDET_F16) return co_t == 128 ? launch_ws<f16, 128>(a, st) : launch_ws<f16, 64>(a, st);
    return co_t == 128 ? launch_ws<float, 128>(a, st) : launch_ws<float, 64>(a, st);
  };
  return 0;
}

}  // namespace glsdet
