// Shared by the convolution kernels (conv.hip: generic implicit GEMM; conv_halo.hip:
// stride-1 kxk with the input patch staged once per channel chunk).
#pragma once
#include "common.h"

namespace glsdet {

struct ConvArgs {
  const unsigned char* x;   // bytes
  const unsigned char* w;
  const float* scale;
  const float* bias;
  unsigned char* y;
  const unsigned char* res;
  long x_sn, x_sh, x_sw;    // element strides
  long y_sn, y_sh, y_sw;
  long r_sn, r_sh, r_sw;
  int N, H, W, Cin;
  int Ho, Wo, Cout, cout_pad;
  int R, S, stride, pad, act;
  int dbg;                  // diagnostic bits of tile_hint (timing experiments; results are then invalid)
  int act_post;             // activation applied AFTER the residual add (GLSDET_ACT_RES_FIRST), else 0
  int kreal, kpad;          // elements
  int M;                    // N*Ho*Wo
  int n_co_tiles, n_px_tiles;
  // buffer-descriptor view of the two streamed operands (bounds-checked loads: an offset
  // >= *_bytes returns zeros, which is how padding / tails are zero-filled branch-free)
  const unsigned char* x_lo;  // start of the allocation x lives in
  unsigned x_off;             // byte offset of element (0,0,0,0) from x_lo
  unsigned x_bytes;           // size of that allocation
  unsigned w_bytes;           // size of the packed weight buffer (from w)
  // 1 when offset(pixel p) == p * s_w for the view (dense NHWC or a channel slice of one):
  // the epilogue then needs no integer division per 16-byte chunk
  int y_lin, r_lin;
  int x_lin;                  // 1x1 / s1 / p0 conv on a pixel-linear input view
  // p / (Ho*Wo) and r / Wo without the ~25-instruction division sequence: q = umulhi(n, mul) >> sh, exact for
  // 0 <= n < 2^31 (host: gls_fastdiv); sh < 0 marks a divisor of 1
  unsigned howo_mul, wo_mul;
  int howo_sh, wo_sh;
  // the same for the tile decode of a workgroup (scalar, but ~25 SALU per division at the head of every workgroup):
  // n_co_tiles, and tiles_x / tiles_y of the pixel-tile kernels
  unsigned nco_mul, tx_mul, ty_mul;
  int nco_sh, tx_sh, ty_sh;
  // chained 1x1 conv (glsdet_conv2d_chain): y2 = act2(scale2 * W2 . y[.., c2_0 : c2_0 + cin2] + bias2), computed by the
  // workgroup from the tile it just stored (w2 == nullptr: none).  y2 has y's pixel geometry.
  const unsigned char* w2;
  const float* scale2;
  const float* bias2;
  unsigned char* y2;
  long y2_sn, y2_sh, y2_sw;
  int c2_0, cin2, cout2, cout2_pad, kpad2, act2, y2_lin;
  unsigned w2_bytes;
  int y_skip;                 // chained form only (GLSDET_CHAIN_SKIP_Y): y itself is read by nothing but the chained conv and is not stored
  // GroupNorm statistics of the output (glsdet_conv2d_gnstats): per (image, pixel tile, wave, group) the sum and the sum
  // of squares of the STORED values, fp64, at gn_part[(((img * tiles + tile) * 4 + wave) * gn_groups + group) * 2]; nullptr: none
  double* gn_part;
  int gn_cpg, gn_groups;
};

// Bottleneck front (glsdet_bottleneck, conv_bneck.hip): `c` describes the 3x3 (weights, epilogue, output, residual,
// Cin = hidden channels) except that its x fields address the INPUT of the 1x1 that produces the 3x3's input on the fly.
struct BneckArgs {
  ConvArgs c;
  const unsigned char* w0;    // packed 1x1 weights [cout_pad(hidden)][kpad0]
  const float* scale0;
  const float* bias0;
  int cin0, kpad0, act0;
  unsigned w0_bytes;
};

// host: mul, sh with floor(n / d) == umulhi(n, mul) >> sh for 0 <= n < 2^31, 2 <= d < 2^31; d == 1 -> sh = -1
inline void gls_fastdiv(int d, unsigned* mul, int* sh) {
  if (d <= 1) { *mul = 0; *sh = -1; return; }
  int l = 0;
  while ((1ll << l) < d) ++l;                                   // 2^l >= d
  *mul = (unsigned)(((1ull << (31 + l)) / (unsigned long long)d) + 1ull);   // < 2^32 because 2^l / d < 2
  *sh = l - 1;
}
__device__ __forceinline__ int gls_div(int n, unsigned mul, int sh) {
  return sh < 0 ? n : (int)(__umulhi((unsigned)n, mul) >> sh);
}

// element offset of flat output pixel p (+ channel) in a view
__device__ __forceinline__ long gls_pix_off(int p, int HoWo, int Wo, long sn, long sh, long sw, int lin, const ConvArgs& a) {
  if (lin) return (long)p * sw;
  const int n = gls_div(p, a.howo_mul, a.howo_sh), rem = p - n * HoWo;
  const int ho = gls_div(rem, a.wo_mul, a.wo_sh), wo = rem - ho * Wo;
  return (long)n * sn + (long)ho * sh + (long)wo * sw;
}

#define GLS_OOB 0x80000000u   // any offset >= 2^31 is out of range of every descriptor we build
typedef __attribute__((address_space(8))) void* gls_rsrc_t;
__device__ __forceinline__ auto gls_make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)bytes, 0x00020000);
}
template <typename R>
__device__ __forceinline__ u32x4 gls_buf_load16(R rsrc, unsigned voff) {
  return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, 0, 0));
}

template <typename T>
struct MMA;
template <>
struct MMA<f16> {
  static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a),
                                               __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};
template <>
struct MMA<float> {
  static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x16& c) {
    const f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0], bf[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1], bf[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(af[2], bf[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(af[3], bf[3], c, 0, 0, 0);
  }
};

// v_mfma_f32_16x16x32_f16 (fp16 kernels that are MFMA bound): the same FLOPs per cycle as 32x32x16, but the chip holds a
// higher clock under it -- tools/probe/mfma_loop.hip, the ring8 step on random operands, 8000 back-to-back launches:
// 32x32x16 1302-1323 TFLOP/s, 16x16x32 1418-1430, 16x16x32 + s_setprio 1562 (constant operands: 1835 / 1942: the clock, not
// the instruction count, is what bounds these loops).  A: lane -> row lane % 16, k slice lane / 16 (8 halves = one 16-byte
// chunk); B likewise; D: lane -> column lane % 16, rows 4 * (lane / 16) .. + 3 (row i = what lane i supplied as A: with
// the weight rows permuted over the lanes (m16_wrow) a lane's four values are the channels m16_wrow(4 * (lane / 16)) .. + 3).
__device__ __forceinline__ void mma16(const u32x4& a, const u32x4& b, f32x4& c) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// Fragment maps of the 16x16x32 form.  The LDS serves a ds_read_b128 in 16-lane groups that pair 8 lanes of k slice a
// (lanes {0-3, 12-15}: "half 0") with 8 lanes of slice b ({20-27}: "half 1" of the next 16), and c with d.  With the slices
// (a, b, c, d) = chunks (0, 2, 1, 3) of the 64-byte step the two halves of a group are 32 bytes apart, and a group is
// conflict free when each half holds 8 rows whose 16-byte columns (row * pitch / 16 mod 16) fill two residue classes mod 4
// once each, the same two classes in both halves when they are adjacent ({c, c + 1}), complementary ones otherwise.
//   position i = 0..7 of half h -> lane % 16
__host__ __device__ constexpr int m16_lane(int h, int i) { return h == 0 ? (i < 4 ? i : i + 8) : i + 4; }
//   lane % 16 -> (half, position)
__host__ __device__ constexpr int m16_half(int n) { return (n >= 4 && n < 12) ? 1 : 0; }
__host__ __device__ constexpr int m16_pos(int n) { return n < 4 ? n : (n < 12 ? n - 4 : n - 8); }
//   k slice lane / 16 -> 16-byte chunk of the 64-byte step
__host__ __device__ constexpr int m16_chunk(int sl) { return ((sl & 1) << 1) | (sl >> 1); }
//   weights (unpadded rows of KB bytes, filled by LDS-DMA): half 0 = rows 0..7, half 1 = rows 8..15 of the 16-row block;
//   the slot of chunk c in row r is c ^ swz: KB 128: (r / 2) & 7 (as for 32x32x16), KB 64: (r / 4) & 1
__host__ __device__ constexpr int m16_wrow(int n) { return m16_half(n) * 8 + m16_pos(n); }
template <int KB, bool M16>
__host__ __device__ constexpr int w_swz(int row) {
  return (M16 && KB == 64) ? ((row >> 2) & 1) : ((row / (256 / KB)) & (KB / 16 - 1));
}
//   patch rows (pitch KB + 16: 5 or 9 sixteen-byte columns) of 16 consecutive slots: half 0 = the even ones, half 1 = the odd
__host__ __device__ constexpr int m16_px16(int n) { return 2 * m16_pos(n) + m16_half(n); }

// SiLU: the exact-f32 instantiation uses the accurate expf, the fp16 one the native exp
template <typename T>
__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == GLSDET_ACT_SILU) {
    if (sizeof(T) == 4) return v / (1.0f + expf(-v));                  // exact-f32 mode: IEEE divide
    return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));              // fp16 mode: v_exp + v_rcp (1 ulp each)
  }
  if (act == GLSDET_ACT_RELU) return fmaxf(v, 0.0f);
  if (act == GLSDET_ACT_GELU) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
  if (act == GLSDET_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
  if (act == GLSDET_ACT_LRELU) return v > 0.0f ? v : 0.1f * v;
  return v;
}

// four outputs at once: scale, bias and activation on 4-vectors, so that hipcc can use the packed fp32 ALU
// (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two lanes of work per instruction).  Same operations in the same
// order as apply_act per element: results are bit-identical.
template <typename T>
__device__ __forceinline__ f32x4 scale_bias_act4(f32x4 x, f32x4 sc, f32x4 bi, int act) {
  x = x * sc + bi;
  if (act == GLSDET_ACT_SILU && sizeof(T) == 2) {
    const f32x4 t = x * -1.44269504088896340736f;              // exp(-v) = exp2(-v * log2 e), as __expf does
    f32x4 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1]), __builtin_amdgcn_exp2f(t[2]), __builtin_amdgcn_exp2f(t[3])};
    e = e + 1.0f;
    const f32x4 r = {__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1]), __builtin_amdgcn_rcpf(e[2]), __builtin_amdgcn_rcpf(e[3])};
    return x * r;
  }
  if (act == GLSDET_ACT_RELU) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    return __builtin_elementwise_max(x, z);
  }
  if (act == GLSDET_ACT_NONE) return x;
#pragma unroll
  for (int e = 0; e < 4; ++e) x[e] = apply_act<T>(x[e], act);
  return x;
}

// pack 4 fp32 -> 4 TO, stored at p (8 B for f16, 16 B for f32)
__device__ __forceinline__ void store4(unsigned char* p, const float (&v)[4], f16*) {
  f16x4 h = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
  *reinterpret_cast<f16x4*>(p) = h;
}
__device__ __forceinline__ void store4(unsigned char* p, const float (&v)[4], float*) {
  f32x4 h = {v[0], v[1], v[2], v[3]};
  *reinterpret_cast<f32x4*>(p) = h;
}
// 16-byte chunk (+)= residual chunk, in fp32; act_post != 0: the activation follows the add
__device__ __forceinline__ u32x4 add_chunk(u32x4 a, u32x4 b, f16*, int act_post) {
  f16x8 x = __builtin_bit_cast(f16x8, a), y = __builtin_bit_cast(f16x8, b);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float v = (float)x[i] + (float)y[i];
    if (act_post) v = apply_act<f16>(v, act_post);
    x[i] = (f16)v;
  }
  return __builtin_bit_cast(u32x4, x);
}
__device__ __forceinline__ u32x4 add_chunk(u32x4 a, u32x4 b, float*, int act_post) {
  f32x4 x = __builtin_bit_cast(f32x4, a), y = __builtin_bit_cast(f32x4, b);
  x += y;
  if (act_post) {
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = apply_act<float>(x[i], act_post);
  }
  return __builtin_bit_cast(u32x4, x);
}


// fp16 output with a residual: the epilogue stages the conv result in fp32 ("wide" tile rows of CO_T * 4 + 16 bytes)
// so that residual add (and, for GLSDET_ACT_RES_FIRST, the activation) act on the UNROUNDED value and the sum is
// rounded to fp16 ONCE -- act(conv*scale+bias) + res as one expression, which is what the fp16-storage emulation of
// the oracle models (round 2: the per-layer trace showed 28 % of the Bottleneck outputs one ulp off when the conv
// result was rounded before the add).  lo / hi: the 8 staged floats of a 16-byte output chunk.
__device__ __forceinline__ u32x4 add_chunk_wide(f32x4 lo, f32x4 hi, u32x4 r, int act_post) {
  const f16x8 y = __builtin_bit_cast(f16x8, r);
  f16x8 o;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float v = (i < 4 ? lo[i] : hi[i - 4]) + (float)y[i];
    if (act_post) v = apply_act<f16>(v, act_post);
    o[i] = (f16)v;
  }
  return __builtin_bit_cast(u32x4, o);
}
// one 4-value group of the accumulator -> the staged tile (wide: fp32 rows, see above)
template <typename TO, int CO_T>
__device__ __forceinline__ void stage4(unsigned char* tile, int px_l, int co_l, const float (&v)[4], bool wide) {
  if (sizeof(TO) == 2 && wide) store4(tile + px_l * (CO_T * 4 + 16) + co_l * 4, v, (float*)nullptr);
  else store4(tile + px_l * (CO_T * (int)sizeof(TO) + 16) + co_l * (int)sizeof(TO), v, (TO*)nullptr);
}
// bytes of the staged output tile of PX pixels
template <typename TO>
constexpr int epi_bytes(int co_t, int px, bool wide) { return px * (co_t * (wide ? 4 : (int)sizeof(TO)) + 16); }

// Tile-local pixel index -> (oy, ox) for 16-pixel-wide tiles.  A 32-lane MFMA subtile covers
// two pixel rows; the patch row pitch PW is not a multiple of 16 rows-of-144-bytes, so the
// second row would land on the bank sets of the first (2-way conflicts on ds_read_b128).
// Rotating the odd rows' pixel order by (-PW mod 16) makes lane m of either row hit bank
// set m: conflict free at no LDS cost.  The epilogue decodes with the same function.
template <int PW>
__device__ __forceinline__ void pix_to_xy16(int pix, int& oy, int& ox) {
  oy = pix >> 4;
  const int m = pix & 15;
  constexpr int ROT = (16 - (PW & 15)) & 15;
  ox = (oy & 1) ? ((m + ROT) & 15) : m;
}

// Tile geometries of the pixel-tile kernels (round 3).  A tile is TH x TW output pixels laid row-major onto the 128 (256)
// MFMA lanes of a workgroup; TH * TW may fall short of the lane count (dead lanes: they multiply pixel 0 again and store
// nothing).  8 x 16 wastes 28 % of the lanes on a 50 x 84 map and 47 % on a 13 x 21 quadrant; 10 x 12 and 6 x 21 tile those
// maps with 7-17 % (tile counts: 42 -> 35, 12 -> 10, 4 -> 3 per image).  GEO: 0 = 8 x 16, 1 = 10 x 12, 2 = 6 x 21.
template <int GEO> struct TileGeo;
template <> struct TileGeo<0> { static constexpr int TH = 8, TW = 16; };
template <> struct TileGeo<1> { static constexpr int TH = 10, TW = 12; };
template <> struct TileGeo<2> { static constexpr int TH = 6, TW = 21; };
inline void tile_geo_dims(int geo, int* th, int* tw) {
  *th = geo == 1 ? 10 : (geo == 2 ? 6 : 8);
  *tw = geo == 1 ? 12 : (geo == 2 ? 21 : 16);
}
inline void tile_geo8_dims(int geo, int* th, int* tw) {      // the 8-wave (256-lane) ring kernel: 8 x 32, 10 x 24, 6 x 42
  *th = geo == 1 ? 10 : (geo == 2 ? 6 : 8);
  *tw = geo == 1 ? 24 : (geo == 2 ? 42 : 32);
}
// tile-local pixel index -> (oy, ox); TW == 16 keeps the row rotation of pix_to_xy16
template <int TW, int PW>
__device__ __forceinline__ void pix_to_xy(int pix, int& oy, int& ox) {
  if constexpr (TW == 16) {
    pix_to_xy16<PW>(pix, oy, ox);
  } else {
    oy = pix / TW;                                  // (constant divisor)
    ox = pix - oy * TW;
  }
}

// Lane <-> pixel assignment of the other tile geometries.  The B fragment of an MFMA block is one ds_read_b128 per lane at
// patch slot (oy * PW + ox); the LDS serves such a read in groups of 16 lanes ({0-3, 12-15, 20-27} and {4-11, 16-19, 28-31}
// of each half wave) and a group is conflict free when its 16 slots differ mod 16 (row pitches of 80 / 144 bytes: slot * 5
// resp. * 9 sixteen-byte columns, a bijection mod 16).  With rows of 12, 21, 24 or 42 pixels laid onto the lanes in row-major
// order a group straddles patch rows and 43-55 % of the LDS cycles were bank conflicts (rocprofv3, 5x5 256 -> 256 @50x84),
// which ate the whole gain of the smaller tile count.  So: the patch pitch is made ODD (every row then starts at another
// slot residue), pixels are dealt to (group, position = slot mod 16) pairs -- the k-th pixel of a residue class goes to
// group k -- and position m of a group sits on the lane of that group with lane % 16 == m.  pix[q]: row-major pixel index of
// lane q (q = 32 * block + lane % 32), 0xffff = dead lane.  Built at compile time.
template <int TH, int TW, int PW, int NL>
struct GeoMap {
  unsigned short pix[NL];
  static constexpr int lane_of(int g, int m) {
    const int j = g / 2;
    int l = 0;
    if ((g & 1) == 0) l = (m < 4 || m >= 12) ? m : 16 + m;
    else l = (m >= 4 && m < 12) ? m : 16 + m;
    return j * 32 + l;
  }
  constexpr GeoMap() : pix{} {
    int cnt[16] = {};
    bool used[NL] = {};
    int over[NL] = {};
    int nover = 0;
    for (int q = 0; q < NL; ++q) pix[q] = 0xffff;
    for (int p = 0; p < TH * TW; ++p) {
      const int oy = p / TW, ox = p % TW, s = (oy * PW + ox) & 15;
      const int g = cnt[s]++;
      if (g < NL / 16) {
        const int q = lane_of(g, s);
        pix[q] = (unsigned short)p;
        used[q] = true;
      } else {
        over[nover++] = p;                          // more than NL / 16 pixels of one residue: a (rare) two-way conflict
      }
    }
    int qf = 0;
    for (int i = 0; i < nover; ++i) {
      while (used[qf]) ++qf;
      pix[qf] = (unsigned short)over[i];
      used[qf] = true;
    }
  }
};
template <int TH, int TW, int PW, int NL>
struct GeoMapHolder { static constexpr GeoMap<TH, TW, PW, NL> map{}; };

// ... and for the 16x16x32 form (m16_* above): pixels are dealt to HALVES of 16-lane MFMA column blocks.  A half takes one
// pixel of each of the 8 slot residues (mod 16) of a class pair -- type A: residues with s % 4 in {0, 1}, type B: {2, 3} --
// and a block is two halves of one type (adjacent classes: both halves the same pair).  pix[q]: row-major pixel index of lane
// column q = 16 * block + lane % 16, 0xffff = dead.  With an odd patch pitch every residue holds TH * TW / 16 pixels to within
// one, so the NL / 16 blocks take all of them; a surplus pixel (none for the geometries in use) goes to any free lane.
template <int TH, int TW, int PW, int NL>
struct GeoMap16 {
  unsigned short pix[NL];
  constexpr GeoMap16() : pix{} {
    constexpr int NB = NL / 16;
    int cnt[16] = {};
    int maxc[2] = {0, 0};
    for (int p = 0; p < TH * TW; ++p) {
      const int s = ((p / TW) * PW + p % TW) & 15, t = (s & 3) >> 1;
      ++cnt[s];
      if (cnt[s] > maxc[t]) maxc[t] = cnt[s];
    }
    // halves 0 .. maxc[0] - 1 of type A fill blocks 0 .. nbA - 1, type B the blocks behind them
    int nbA = (maxc[0] + 1) / 2;
    if (nbA + (maxc[1] + 1) / 2 > NB) nbA = NB / 2;
    bool used[NL] = {};
    int over[NL] = {};
    int nover = 0;
    int c2[16] = {};
    for (int q = 0; q < NL; ++q) pix[q] = 0xffff;
    for (int p = 0; p < TH * TW; ++p) {
      const int s = ((p / TW) * PW + p % TW) & 15, t = (s & 3) >> 1;
      const int k = c2[s]++;                                  // k-th pixel of this residue -> half k of its type
      const int blk = (t == 0 ? 0 : nbA) + k / 2, lim = t == 0 ? nbA : NB;
      const int i = ((s >> 2) << 1) | (s & 1);                // position of the residue inside its class pair: 0..7
      if (blk < lim) {
        const int q = blk * 16 + m16_lane(k & 1, i);
        pix[q] = (unsigned short)p;
        used[q] = true;
      } else {
        over[nover++] = p;
      }
    }
    int qf = 0;
    for (int i = 0; i < nover; ++i) {
      while (used[qf]) ++qf;
      pix[qf] = (unsigned short)over[i];
      used[qf] = true;
    }
  }
};
template <int TH, int TW, int PW, int NL>
struct GeoMap16Holder { static constexpr GeoMap16<TH, TW, PW, NL> map{}; };

// Store phase of the 8 x 16 pixel-tile kernels: the staged tile (rows of ORS bytes in LDS) -> global, 16-byte chunks,
// optional residual.  A thread's chunks are 256 / OCPR pixels apart -- one or two tile rows -- so its addresses advance
// by a constant: the 64-bit offsets are built once and stepped with one add per chunk (the per-chunk multiplies were
// ~10 vector instructions each in kernels that PMC shows issue bound), and the loop is fully unrolled.
// sum and sum of squares of one stored 16-byte chunk
__device__ __forceinline__ void gn_chunk_sums(u32x4 v, f16*, float& s1, float& s2) {
  const f16x8 h = __builtin_bit_cast(f16x8, v);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float f = (float)h[e];
    s1 += f;
    s2 += f * f;
  }
}
__device__ __forceinline__ void gn_chunk_sums(u32x4 v, float*, float& s1, float& s2) {
  const f32x4 h = __builtin_bit_cast(f32x4, v);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    s1 += h[e];
    s2 += h[e] * h[e];
  }
}
// GroupNorm partials of a tile (GN instantiations only): every thread owns ONE channel chunk (cq = tid % OCPR) of several
// pixels.  The lanes of a wave that hold chunks of the same group are folded by xor shuffles (fixed order, no LDS, no
// barrier) and one lane per (wave, group) writes the fp64 partial: four slices per tile, one per wave.
template <typename TO, int CO_T>
__device__ __forceinline__ void gn_tile_partials(const ConvArgs& a, int img, int ty0, int tx0, int co0, int tid, float s1, float s2) {
  constexpr int VO = 16 / (int)sizeof(TO), OCPR = CO_T / VO;
  const int cpv = a.gn_cpg / VO;                  // chunks per group: 1 (f16, 8 channels per group), 2, 4, ...
  for (int m = 1; m < cpv; m <<= 1) {
    s1 += __shfl_xor(s1, m);
    s2 += __shfl_xor(s2, m);
  }
#pragma unroll
  for (int m = OCPR; m < 64; m <<= 1) {
    s1 += __shfl_xor(s1, m);
    s2 += __shfl_xor(s2, m);
  }
  const int lane = tid & 63, wave = tid >> 6;
  if (lane < OCPR && lane % cpv == 0 && co0 + lane * VO < a.Cout) {
    const int tiles_x = (a.Wo + 15) / 16, tiles = tiles_x * ((a.Ho + 7) / 8);
    const int tile = (ty0 >> 3) * tiles_x + (tx0 >> 4);
    double* o = a.gn_part + ((((long)img * tiles + tile) * 4 + wave) * a.gn_groups + (co0 + lane * VO) / a.gn_cpg) * 2;
    o[0] = (double)s1;
    o[1] = (double)s2;
  }
}

template <typename TO, int CO_T, int PW, bool GN = false, int TH = 8, int TW = 16>
__device__ __forceinline__ void halo_store_tile(const unsigned char* stile, const ConvArgs& a, int img, int ty0, int tx0, int co0,
                                                int tid) {
  constexpr int PX_T = 128, ORS = CO_T * (int)sizeof(TO) + 16;
  constexpr int LIVE = TH * TW;                   // lanes beyond the tile's pixels store nothing
  constexpr int VO = 16 / (int)sizeof(TO);
  constexpr int OCPR = CO_T / VO;                 // chunks per pixel
  constexpr int PXS = 256 / OCPR;                 // pixels between two chunks of one thread (a multiple of 16)
  float gs1 = 0.f, gs2 = 0.f;                     // (GN) sums over this thread's stored chunks
  if (sizeof(TO) == 2 && a.res) {                 // wide staging (fp32 rows): add in fp32, round once
    constexpr int ORSW = CO_T * 4 + 16;
    for (int q = tid; q < LIVE * OCPR; q += 256) {
      const int px_l = q / OCPR, cq = q - px_l * OCPR;
      int oy, ox;
      pix_to_xy<TW, PW>(px_l, oy, ox);
      const int ho = ty0 + oy, wo = tx0 + ox, co = co0 + cq * VO;
      if (ho < a.Ho && wo < a.Wo && co < a.Cout) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(stile + px_l * ORSW + cq * 32);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(stile + px_l * ORSW + cq * 32 + 16);
        const long ro = (long)img * a.r_sn + (long)ho * a.r_sh + (long)wo * a.r_sw + co;
        const u32x4 v = add_chunk_wide(lo, hi, *reinterpret_cast<const u32x4*>(a.res + ro * 2), a.act_post);
        const long yo = (long)img * a.y_sn + (long)ho * a.y_sh + (long)wo * a.y_sw + co;
        *reinterpret_cast<u32x4*>(a.y + yo * 2) = v;
        if constexpr (GN) gn_chunk_sums(v, (TO*)nullptr, gs1, gs2);
      }
    }
    if constexpr (GN) gn_tile_partials<TO, CO_T>(a, img, ty0, tx0, co0, tid, gs1, gs2);
    return;
  }
  if constexpr (PXS % 16 != 0 || TW != 16) {      // fp32 output with 128-row tiles (8 pixels apart) / other tile geometries: plain form
    for (int q = tid; q < LIVE * OCPR; q += 256) {
      const int px_l = q / OCPR, cq = q - px_l * OCPR;
      int oy, ox;
      pix_to_xy<TW, PW>(px_l, oy, ox);
      const int ho = ty0 + oy, wo = tx0 + ox, co = co0 + cq * VO;
      if (ho < a.Ho && wo < a.Wo && co < a.Cout) {
        u32x4 v = *reinterpret_cast<const u32x4*>(stile + px_l * ORS + cq * 16);
        if (a.res) {
          const long ro = (long)img * a.r_sn + (long)ho * a.r_sh + (long)wo * a.r_sw + co;
          v = add_chunk(v, *reinterpret_cast<const u32x4*>(a.res + ro * (long)sizeof(TO)), (TO*)nullptr, a.act_post);
        }
        const long yo = (long)img * a.y_sn + (long)ho * a.y_sh + (long)wo * a.y_sw + co;
        *reinterpret_cast<u32x4*>(a.y + yo * (long)sizeof(TO)) = v;
        if constexpr (GN) gn_chunk_sums(v, (TO*)nullptr, gs1, gs2);
      }
    }
    if constexpr (GN) gn_tile_partials<TO, CO_T>(a, img, ty0, tx0, co0, tid, gs1, gs2);
    return;
  }
  constexpr int NCH = PX_T / (PXS > 0 ? PXS : 1); // chunks per thread
  constexpr int RSTEP = PXS / 16;                 // tile rows between them
  const int cq = tid % OCPR, pxl0 = tid / OCPR;   // first pixel of this thread: pxl0 < PXS
  const int co = co0 + cq * VO;
  const int m = pxl0 & 15, oy0 = pxl0 >> 4;
  constexpr int ROT = (16 - (PW & 15)) & 15;      // pix_to_xy16: odd rows are rotated
  const int ox_e = m, ox_o = (m + ROT) & 15;
  const bool cok = co < a.Cout;
  const long ybase = (long)img * a.y_sn + (long)(ty0 + oy0) * a.y_sh + (long)tx0 * a.y_sw + co;
  const long rbase = a.res ? (long)img * a.r_sn + (long)(ty0 + oy0) * a.r_sh + (long)tx0 * a.r_sw + co : 0;
  const long y_e = ybase + (long)ox_e * a.y_sw, y_o = ybase + (long)ox_o * a.y_sw;
  const long r_e = rbase + (long)ox_e * a.r_sw, r_o = rbase + (long)ox_o * a.r_sw;
  const long ystep = (long)RSTEP * a.y_sh, rstep = (long)RSTEP * a.r_sh;
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int oy = oy0 + k * RSTEP;
    const bool odd = ((oy0 + k * RSTEP) & 1) != 0;            // RSTEP even: the parity never changes
    const int ox = odd ? ox_o : ox_e;
    if (ty0 + oy < a.Ho && tx0 + ox < a.Wo && cok) {
      u32x4 v = *reinterpret_cast<const u32x4*>(stile + (pxl0 + k * PXS) * ORS + cq * 16);
      if (a.res) {
        const long ro = (odd ? r_o : r_e) + k * rstep;
        v = add_chunk(v, *reinterpret_cast<const u32x4*>(a.res + ro * (long)sizeof(TO)), (TO*)nullptr, a.act_post);
      }
      const long yo = (odd ? y_o : y_e) + k * ystep;
      *reinterpret_cast<u32x4*>(a.y + yo * (long)sizeof(TO)) = v;
      if constexpr (GN) gn_chunk_sums(v, (TO*)nullptr, gs1, gs2);
    }
  }
  if constexpr (GN) gn_tile_partials<TO, CO_T>(a, img, ty0, tx0, co0, tid, gs1, gs2);
}

// ---- chained 1x1 conv on the tile a workgroup has just produced (CSPLayer: conv1|conv2 -> m.0.conv1, Bottleneck i ->
// Bottleneck i+1 conv1).  `tile`: the FINAL output tile (after residual / post activation) in LDS, PX_T rows of
// ORS = CO_T * sizeof(TO) + 16 bytes, TO == T.  `wbuf`: LDS for W2 (cout2_pad rows of cin2 * sizeof(T) + 16 bytes), later
// reused for the staged y2 tile.  The product reads the stored (rounded) values in the k order of the stand-alone 1x1
// kernels, so the result equals a separate launch bit for bit.  All 256 threads call it; it starts and ends with a barrier.
// pix(px_l, &ok) -> element offset of tile pixel px_l in y2's view (ok = false: outside the image).
template <typename T, int CO_T, int PX_T, typename PixFn>
__device__ __forceinline__ void chain_1x1(const ConvArgs& a, unsigned char* tile, unsigned char* wbuf, int co0, int tid, PixFn pix) {
  constexpr int ES = (int)sizeof(T), VEC = 16 / ES;
  constexpr int ORS = CO_T * ES + 16;
  const int lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int rs2 = a.cin2 * ES + 16;                       // W2 row pitch in LDS
  const int cpr = a.cin2 * ES / 16;                       // 16-byte chunks per W2 row
  const auto wrs = gls_make_rsrc(a.w2, a.w2_bytes);
  for (int q = tid; q < a.cout2_pad * cpr; q += 256) {
    const int row = q / cpr, c = q - row * cpr;
    *reinterpret_cast<u32x4*>(wbuf + row * rs2 + c * 16) = gls_buf_load16(wrs, (unsigned)(row * a.kpad2 * ES + c * 16));
  }
  __syncthreads();
  constexpr int NPB = PX_T / 32, WPB = 4 / NPB;           // pixel blocks of 32; waves per pixel block
  const int pb = wave % NPB, rb0 = wave / NPB;
  const int nrb = a.cout2_pad / 32;
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.0f;
  const unsigned char* brow = tile + (pb * 32 + l31) * ORS + (a.c2_0 - co0) * ES + lh * 16;
  const int nk = a.cin2 * ES / 32;
  for (int kk = 0; kk < nk; ++kk) {
    const u32x4 bf = *reinterpret_cast<const u32x4*>(brow + kk * 32);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rb = rb0 + i * WPB;
      if (rb < nrb) {
        const u32x4 af = *reinterpret_cast<const u32x4*>(wbuf + (rb * 32 + l31) * rs2 + kk * 32 + lh * 16);
        MMA<T>::run(af, bf, acc[i]);
      }
    }
  }
  __syncthreads();                                        // every wave is done with W2: its LDS becomes the y2 stage
  const int ors2 = a.cout2_pad * ES + 16;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rb = rb0 + i * WPB;
    if (rb < nrb) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int co_l = rb * 32 + 8 * g + 4 * lh;
        const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale2 + co_l), bi = *reinterpret_cast<const f32x4*>(a.bias2 + co_l);
        const f32x4 xv = {acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]};
        const f32x4 yv = scale_bias_act4<T>(xv, sc, bi, a.act2);
        const float v[4] = {yv[0], yv[1], yv[2], yv[3]};
        store4(wbuf + (pb * 32 + l31) * ors2 + co_l * ES, v, (T*)nullptr);
      }
    }
  }
  __syncthreads();
  const int ocpr = a.cout2_pad / VEC;
  for (int q = tid; q < PX_T * ocpr; q += 256) {
    const int px_l = q / ocpr, cq = q - px_l * ocpr;
    bool ok;
    const long po = pix(px_l, ok);
    if (ok && cq * VEC < a.cout2)
      *reinterpret_cast<u32x4*>(a.y2 + (po + cq * VEC) * (long)ES) = *reinterpret_cast<const u32x4*>(wbuf + px_l * ors2 + cq * 16);
  }
}
// LDS bytes a kernel needs for the chained 1x1 on a CO_T x PX_T tile: the final tile + max(W2, staged y2)
template <typename T>
inline int chain_lds_bytes(int co_t, int px_t, const ConvArgs& a) {
  const int es = (int)sizeof(T);
  const int tile = px_t * (co_t * es + 16);
  const int w2 = a.cout2_pad * (a.cin2 * es + 16), st = px_t * (a.cout2_pad * es + 16);
  return tile + (w2 > st ? w2 : st);
}

// Store phase of the pixel-tile kernels when a chained 1x1 follows AND the layer has a residual: the final values are
// computed in registers (pass 1, stored to y), then written back to LDS in the TO row layout (pass 2) for chain_1x1.
template <typename TO, int CO_T, int PW>
__device__ __forceinline__ void halo_store_tile_keep(unsigned char* stile, const ConvArgs& a, int img, int ty0, int tx0, int co0,
                                                     int tid) {
  constexpr int PX_T = 128, ORS = CO_T * (int)sizeof(TO) + 16, ORSW = CO_T * 4 + 16;
  constexpr int VO = 16 / (int)sizeof(TO), OCPR = CO_T / VO, NIT = PX_T * OCPR / 256;
  const bool wide = sizeof(TO) == 2;
  u32x4 fin[NIT];
#pragma unroll
  for (int b = 0; b < NIT; ++b) {
    const int q = tid + b * 256;
    const int px_l = q / OCPR, cq = q - px_l * OCPR;
    int oy, ox;
    pix_to_xy16<PW>(px_l, oy, ox);
    const int ho = ty0 + oy, wo = tx0 + ox, co = co0 + cq * VO;
    fin[b] = u32x4{0u, 0u, 0u, 0u};
    if (ho < a.Ho && wo < a.Wo && co < a.Cout) {
      const long ro = (long)img * a.r_sn + (long)ho * a.r_sh + (long)wo * a.r_sw + co;
      const u32x4 rv = *reinterpret_cast<const u32x4*>(a.res + ro * (long)sizeof(TO));
      if (wide) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(stile + px_l * ORSW + cq * 32);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(stile + px_l * ORSW + cq * 32 + 16);
        fin[b] = add_chunk_wide(lo, hi, rv, a.act_post);
      } else {
        fin[b] = add_chunk(*reinterpret_cast<const u32x4*>(stile + px_l * ORS + cq * 16), rv, (TO*)nullptr, a.act_post);
      }
      const long yo = (long)img * a.y_sn + (long)ho * a.y_sh + (long)wo * a.y_sw + co;
      *reinterpret_cast<u32x4*>(a.y + yo * (long)sizeof(TO)) = fin[b];
    }
  }
  __syncthreads();
#pragma unroll
  for (int b = 0; b < NIT; ++b) {
    const int q = tid + b * 256;
    const int px_l = q / OCPR, cq = q - px_l * OCPR;
    *reinterpret_cast<u32x4*>(stile + px_l * ORS + cq * 16) = fin[b];
  }
}

// epilogue tail shared by the pixel-tile kernels: store the staged tile, then the chained 1x1 if this workgroup's cout tile
// holds its input channels
template <typename T, typename TO, int CO_T, int PW, bool CH, bool GN = false, int TH = 8, int TW = 16>
__device__ __forceinline__ void halo_store_and_chain(unsigned char* smem, const ConvArgs& a, int img, int ty0, int tx0, int co0,
                                                     int tid) {
  static_assert((TH == 8 && TW == 16) || (!CH && !GN), "chained / GroupNorm forms exist for 8 x 16 tiles only");
  if constexpr (TW != 16) {
    halo_store_tile<TO, CO_T, PW, false, TH, TW>(smem, a, img, ty0, tx0, co0, tid);
    return;
  }
  if constexpr (GN) {
    halo_store_tile<TO, CO_T, PW, true>(smem, a, img, ty0, tx0, co0, tid);
    return;
  }
  if constexpr (CH && sizeof(T) == sizeof(TO)) {
    const bool chain = a.w2 != nullptr && co0 <= a.c2_0 && a.c2_0 + a.cin2 <= co0 + CO_T;
    if (chain) {
      if (a.res) halo_store_tile_keep<TO, CO_T, PW>(smem, a, img, ty0, tx0, co0, tid);
      else if (!a.y_skip) halo_store_tile<TO, CO_T, PW>(smem, a, img, ty0, tx0, co0, tid);
      auto y2pix = [&](int px_l, bool& ok) -> long {
        int oy, ox;
        pix_to_xy16<PW>(px_l, oy, ox);
        const int ho = ty0 + oy, wo = tx0 + ox;
        ok = ho < a.Ho && wo < a.Wo;
        return (long)img * a.y2_sn + (long)ho * a.y2_sh + (long)wo * a.y2_sw;
      };
      chain_1x1<T, CO_T, 128>(a, smem, smem + 128 * (CO_T * (int)sizeof(TO) + 16), co0, tid, y2pix);
      return;
    }
  }
  halo_store_tile<TO, CO_T, PW>(smem, a, img, ty0, tx0, co0, tid);
}

// host-side dispatch of the halo kernel (conv_halo.hip); returns 1 if it does not apply
// several problems of one shape class in the kernarg segment (conv_igemm_multi_kernel, conv_halo_ring_multi_kernel)
#define GLS_MULTI 8       // 8 x sizeof(ConvArgs) = 2.2 KB of the 4 KB kernarg segment
struct ConvArgsN {
  ConvArgs p[GLS_MULTI];
  int start[GLS_MULTI + 1];
  int n;
};
// up to GLS_BATCH problems that differ ONLY in their operand addresses (the per-image, per-quadrant GEMMs of the ResNet GL
// plug-in: 32 equal windows per level): one full argument block + 72 bytes per problem instead of 280
#define GLS_BATCH 32
struct ConvPtrs {
  const unsigned char* x;
  const unsigned char* w;
  const float* scale;
  const float* bias;
  unsigned char* y;
  const unsigned char* res;
  const unsigned char* x_lo;
  unsigned x_off, x_bytes;
  unsigned w_bytes, _pad;
};
struct ConvArgsB {
  ConvArgs base;            // problem 0; n_co_tiles / n_px_tiles filled by the launcher
  ConvPtrs p[GLS_BATCH];
  int n, tiles;             // problems; tiles per problem
  unsigned tiles_mul;
  int tiles_sh;
};
struct HaloArgsN {
  ConvArgs p[GLS_MULTI];
  int start[GLS_MULTI + 1];
  int tx[GLS_MULTI], ty[GLS_MULTI];      // 8 x 16 tiles per row / column of each problem
  int n;
};
int conv_halo_try(const ConvArgs& a, int xdt, int ydt, int hint, OpRecord* op);
int conv_halo_multi_try(const ConvArgsN& m, int xdt, int ydt, int hint, OpRecord* op);
// fused Bottleneck front (conv_bneck.hip); hint 0 / 1 = 128- / 64-byte channel chunks; returns 1 if it does not apply
int conv_bneck_try(const BneckArgs& b, int dt, int hint, OpRecord* op);
// weight-stationary persistent 1x1 kernel (conv1x1.hip), tile_hint 3; returns 1 if it does not apply
int conv1x1_ws_try(const ConvArgs& a, int xdt, int ydt, OpRecord* op);
// persistent LDS-DMA GEMM kernel for 1x1 convs (conv_gemm.hip), tile_hint 16 + variant; returns 1 if it does not apply
int conv_gemm_try(const ConvArgs& a, int xdt, int ydt, int hint, OpRecord* op);

}  // namespace glsdet
