// Ceiling probe for the 8-wave halo ring kernel's step structure (tools/probe: built and run on the GPU box only):
//   per step: [s_waitcnt vmcnt + s_barrier] [one weight-tile DMA piece per wave] [TM+TN ds_read_b128 + TM*TN MFMA] x KK
// on LDS images of the production kernel's shapes (128 cout rows x KB bytes ring slots, a padded patch).  Variants by
// -D flags show what each element of the step costs and what a software-pipelined fragment read would buy.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_loop mfma_loop.hip [-DNO_DMA] [-DNO_BARRIER] [-DNO_WAIT] [-DPIPE] [-DSETPRIO] [-DWAVES=8] [-DKB=64]
//   -DRANDOM: LDS images and weights hold random fp16 values (the clock the chip holds depends on the operand data);
//   -DSHAPE16: v_mfma_f32_16x16x32_f16 (16 per step) instead of 32x32x16 (8 per step), KB = 64, conflict-free fragment maps
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#ifndef KB
#define KB 64
#endif
#ifndef WAVES
#define WAVES 8
#endif
#ifndef RING
#define RING 4
#endif
#ifndef NSTEPS
#define NSTEPS 196
#endif
constexpr int RS = KB + 16, CPRW = KB / 16, RPL = 256 / KB, RPI = 64 / CPRW;
constexpr int CO_T = 128, PH = 14, PW = 38;
constexpr int A_BYTES = CO_T * KB;
constexpr int NIW = (CO_T / RPI) / WAVES;         // DMA pieces per wave and tap; 0 when a tap's tile has fewer pieces than waves
constexpr int NI = NIW > 0 ? NIW : 1;            // (KB = 32: 4 pieces of 32 rows, issued by waves 0..3)
constexpr bool SOME_WAVES = NIW == 0;
constexpr int PATCH_OFF = RING * A_BYTES;
constexpr int LDS = PATCH_OFF + PH * PW * RS;
constexpr int KK = KB / 32;

__device__ __forceinline__ void mma(const u32x4& a, const u32x4& b, f32x16& c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

__global__ __launch_bounds__(WAVES * 64) void loop_kernel(const unsigned char* __restrict__ w, unsigned w_bytes, float* out, int nsteps) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const auto wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(w), (short)0, (int)w_bytes, 0x00020000);
#ifdef RANDOM
  for (int i = tid; i < LDS / 4; i += WAVES * 64) {
    unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const unsigned lo = (h & 0x83ffu) | ((13u + (h >> 10) % 3u) << 10), hi = ((h >> 16) & 0x83ffu) | ((13u + (h >> 26) % 3u) << 10);
    reinterpret_cast<unsigned*>(smem)[i] = lo | (hi << 16);
  }
#else
  for (int i = tid; i < LDS / 16; i += WAVES * 64) reinterpret_cast<u32x4*>(smem)[i] = u32x4{0x3c003c00u + (unsigned)i, 0x38003800u, 0x34003400u, 0x30003000u};
#endif
  __syncthreads();
  unsigned wd[NI > 0 ? NI : 1];
#pragma unroll
  for (int q = 0; q < NI; ++q) {
    const int row = RPI * (wave + WAVES * q) + lane / CPRW;
    const int ch = (lane % CPRW) ^ ((row / RPL) & (CPRW - 1));
    wd[q] = (unsigned)(row * 4096 + ch * 16) + (blockIdx.x & 7) * 64;
  }
  int dslot = 0;
  unsigned dadd = 0;
  auto dma_next = [&]() __attribute__((always_inline)) {
#ifndef NO_DMA
    unsigned char* dst = smem + dslot * A_BYTES + wave * 1024;
    if (!SOME_WAVES || wave < CO_T / RPI) {
#pragma unroll
    for (int q = 0; q < NI; ++q)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_ptr)(dst + q * WAVES * 1024), 16, (int)(wd[q] + dadd), 0, 0, 0);
    }
    dslot = dslot + 1 == RING ? 0 : dslot + 1;
    dadd = (dadd + 256) & 2047;
#endif
  };
  constexpr int TM = 2, TN = 2;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
  const int wco = wave % 2, wpx = (wave / 2) % 4;
  const int l31 = lane & 31, lh = lane >> 5;
  const int a_row = (wco * 64 + l31) * KB;
  int a_sw[KK];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) a_sw[kk] = ((2 * kk + lh) ^ ((l31 / RPL) & (CPRW - 1))) << 4;
  int b_off[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) b_off[j] = PATCH_OFF + ((wpx * 2 + j) * PW + l31) * RS + lh * 16;
#pragma unroll
  for (int g = 0; g < RING - 1; ++g) dma_next();
  int g = 0, tap_off = 0, ts = 0;
#ifdef SHAPE16
  // 16 x 16 x 32: lane n = lane % 16 -> operand row, lane / 16 -> one 16-byte k slice of the 64-byte step.  Conflict-free
  // ds_read_b128 (the 16-lane groups pair slice a with b and c with d): slices (a, b, c, d) = chunks (0, 2, 1, 3);
  //   weights (unpadded 64-byte rows): lanes {0-3, 12-15} -> rows 0..7, {4-11} -> rows 8..15, slot = chunk ^ ((row / 4) & 1);
  //   patch (80-byte pitch): lanes {0-3, 12-15} -> even pixels, {4-11} -> odd pixels
  static_assert(KB == 64 && WAVES == 8, "SHAPE16 probe: KB 64, 8 waves");
  f32x4 acc16[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc16[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int n16 = lane & 15, sl = lane >> 4;
  const int chunk = ((sl & 1) << 1) | (sl >> 1);
  const int arow = n16 < 4 ? n16 : (n16 >= 12 ? n16 - 8 : n16 + 4);
  const int bpix = n16 < 4 ? 2 * n16 : (n16 >= 12 ? 2 * (n16 - 8) : 2 * (n16 - 4) + 1);
  int a16[4], b16[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wco * 64 + i * 16 + arow;
    a16[i] = row * KB + ((chunk ^ ((row >> 2) & 1)) << 4);
    b16[i] = PATCH_OFF + ((wpx * 2 + (i >> 1)) * PW + (i & 1) * 16 + bpix) * RS + chunk * 16;
  }
  for (int s = 0; s < nsteps; ++s) {
#ifndef NO_BARRIER
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NI * (RING - 2)) : "memory");
#endif
    dma_next();
    const unsigned char* sA = smem + g * A_BYTES;
#ifdef SETPRIO
    asm volatile("s_setprio 1" ::: "memory");
#endif
    u32x4 af[4], bf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const u32x4*>(sA + a16[i]);
#pragma unroll
    for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const u32x4*>(smem + b16[j] + tap_off);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, af[i]), __builtin_bit_cast(f16x8, bf[j]), acc16[i][j], 0, 0, 0);
#ifdef SETPRIO
    asm volatile("s_setprio 0" ::: "memory");
#endif
    g = g + 1 == RING ? 0 : g + 1;
    ++ts;
    tap_off += (ts == 7) ? (PW - 7 + 1) * RS : RS;
    ts = (ts == 7) ? 0 : ts;
    if (tap_off >= (PH - 1) * PW * RS) tap_off = 0;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i & 1][j & 1][(i >> 1) * 8 + (j >> 1) * 4 + e] += acc16[i][j][e];
#elif defined(PIPE)
  // software pipelined: the fragments of step s + 1 are read while step s is multiplied (ring slot s + 1 is certified by the
  // barrier of step s: the wait allows one stage less in flight)
  u32x4 af[2][KK][TM], bf[2][KK][TN];
  auto rd = [&](int buf, int slot, int toff) __attribute__((always_inline)) {
    const unsigned char* sA = smem + slot * A_BYTES + a_row;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
      for (int i = 0; i < TM; ++i) af[buf][kk][i] = *reinterpret_cast<const u32x4*>(sA + i * 32 * KB + a_sw[kk]);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[buf][kk][j] = *reinterpret_cast<const u32x4*>(smem + b_off[j] + toff + kk * 32);
    }
  };
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NI * (RING - 2)) : "memory");
  rd(0, 0, 0);
  for (int s = 0; s < nsteps; s += 2) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#ifndef NO_BARRIER
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(NI * (RING - 3)) : "memory");
#endif
      dma_next();
      const int gn = g + 1 == RING ? 0 : g + 1;
      ++ts;
      const int tn = tap_off + ((ts == 7) ? (PW - 7 + 1) * RS : RS);
      rd(u ^ 1, gn, tn >= (PH - 1) * PW * RS ? 0 : tn);
#ifdef SETPRIO
      asm volatile("s_setprio 1" ::: "memory");
#endif
#pragma unroll
      for (int kk = 0; kk < KK; ++kk)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) mma(af[u][kk][i], bf[u][kk][j], acc[i][j]);
#ifdef SETPRIO
      asm volatile("s_setprio 0" ::: "memory");
#endif
      g = gn;
      tap_off = tn >= (PH - 1) * PW * RS ? 0 : tn;
      ts = (ts == 7) ? 0 : ts;
    }
  }
#else
  for (int s = 0; s < nsteps; ++s) {
#ifndef NO_BARRIER
#ifdef NO_WAIT
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NI * (RING - 2)) : "memory");
#endif
#endif
    dma_next();
    const unsigned char* sA = smem + g * A_BYTES + a_row;
#ifdef SETPRIO
    asm volatile("s_setprio 1" ::: "memory");
#endif
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      u32x4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const u32x4*>(sA + i * 32 * KB + a_sw[kk]);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const u32x4*>(smem + b_off[j] + tap_off + kk * 32);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) mma(af[i], bf[j], acc[i][j]);
    }
#ifdef SETPRIO
    asm volatile("s_setprio 0" ::: "memory");
#endif
    g = g + 1 == RING ? 0 : g + 1;
    ++ts;
    tap_off += (ts == 7) ? (PW - 7 + 1) * RS : RS;
    ts = (ts == 7) ? 0 : ts;
    if (tap_off >= (PH - 1) * PW * RS) tap_off = 0;
  }
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) sum += acc[i][j][e];
  if (sum == 123.456f) out[blockIdx.x * WAVES * 64 + tid] = sum;
}

int main(int argc, char** argv) {
  const int grid = argc > 1 ? atoi(argv[1]) : 512;
  unsigned char* w;
  float* out;
  const unsigned w_bytes = 128 * 4096 + 4096;
  hipMalloc(&w, w_bytes);
#ifdef RANDOM
  {
    std::vector<unsigned short> hw(w_bytes / 2);
    unsigned h = 12345u;
    for (auto& v : hw) { h = h * 1664525u + 1013904223u; v = (unsigned short)(((h >> 8) & 0x83ffu) | ((13u + (h >> 24) % 3u) << 10)); }
    hipMemcpy(w, hw.data(), w_bytes, hipMemcpyHostToDevice);
  }
#else
  hipMemset(w, 0x3c, w_bytes);
#endif
  hipMalloc(&out, (size_t)grid * WAVES * 64 * 4);
  hipFuncSetAttribute(reinterpret_cast<const void*>(loop_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int reps = argc > 2 ? atoi(argv[2]) : 20;         // (DVFS settles over ~2 s of back-to-back launches: pass ~10000)
  for (int it = 0; it < 3 + reps / 2; ++it) hipLaunchKernelGGL(loop_kernel, dim3(grid), dim3(WAVES * 64), LDS, 0, w, w_bytes, out, NSTEPS);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int it = 0; it < reps; ++it) hipLaunchKernelGGL(loop_kernel, dim3(grid), dim3(WAVES * 64), LDS, 0, w, w_bytes, out, NSTEPS);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps;
  const double flops = (double)grid * WAVES * NSTEPS * (KB / 32) * 4 * 2.0 * 32 * 32 * 16;
  printf("grid %d waves %d KB %d ring %d LDS %d B: %.1f us, %.0f TFLOP/s (%.1f %% of 2500)\n", grid, WAVES, KB, RING, LDS, us, flops / us / 1e6, flops / us / 1e6 / 25.0);
  return 0;
}
