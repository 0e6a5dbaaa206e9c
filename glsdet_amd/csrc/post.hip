// Per-anchor decode, class-max + threshold + compaction, and batched (per-class) NMS.
// HBM / latency bound integer-and-compare work: no MFMA; coalesced reads, LDS-staged
// tiles for the O(n^2) passes, 64-bit wave ballots as the suppression bitmask.
#include <stdlib.h>
#include <type_traits>
#include <string.h>

#include "common.h"

namespace glsdet {

#define GLS_MAX_LEVELS 8
struct DecodeArgs {
  const float* base[GLS_MAX_LEVELS];
  long sn[GLS_MAX_LEVELS], sh[GLS_MAX_LEVELS], sw[GLS_MAX_LEVELS];
  int H[GLS_MAX_LEVELS], W[GLS_MAX_LEVELS], start[GLS_MAX_LEVELS + 1];
  float stride_x[GLS_MAX_LEVELS], stride_y[GLS_MAX_LEVELS];
  int n_levels, nc, n, A, mode;
  float in_w, in_h;
  float* out;
  const float* scale;   // optional [n][4] per-image divisors of (x1,y1,x2,y2) (mmdet `rescale`), mode 1
};

__device__ __forceinline__ float sigmoidf_acc(float v) { return 1.0f / (1.0f + expf(-v)); }

// drone/models/core/utils_bbox.py:266-305 (mode 0) / mmdet yolox_head.py:298-308 (mode 1)
__global__ __launch_bounds__(256) void decode_kernel(const DecodeArgs a) {
  const long total = (long)a.n * a.A;
  const int F = 5 + a.nc;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / a.A), an = (int)(i - (long)b * a.A);
    int l = 0;
#pragma unroll
    for (int t = 1; t < GLS_MAX_LEVELS; ++t)
      if (t < a.n_levels && an >= a.start[t]) l = t;
    const int loc = an - a.start[l];
    const int gy = loc / a.W[l], gx = loc - gy * a.W[l];
    const float* p = a.base[l] + b * a.sn[l] + gy * a.sh[l] + gx * a.sw[l];
    float* o = a.out + i * F;
    const float sx = a.stride_x[l], sy = a.stride_y[l];
    const float cx = (p[0] + (float)gx) * sx, cy = (p[1] + (float)gy) * sy;
    const float w = expf(p[2]) * sx, h = expf(p[3]) * sy;
    if (a.mode == 0) {
      o[0] = cx / a.in_w;
      o[1] = cy / a.in_h;
      o[2] = w / a.in_w;
      o[3] = h / a.in_h;
    } else {
      float x1 = cx - w / 2, y1 = cy - h / 2, x2 = cx + w / 2, y2 = cy + h / 2;
      if (a.scale) {          // yolox_head.py:283-285  flatten_bboxes[..., :4] /= scale_factor
        const float* sf = a.scale + 4 * b;
        x1 /= sf[0]; y1 /= sf[1]; x2 /= sf[2]; y2 /= sf[3];
      }
      o[0] = x1; o[1] = y1; o[2] = x2; o[3] = y2;
    }
    for (int c = 4; c < F; ++c) o[c] = sigmoidf_acc(p[c]);
  }
}

// Counter reset as a kernel of our own: hipMemsetAsync nodes inside a captured graph were seen to
// scribble over their target when two graphs replay concurrently on two streams (stale fill
// patterns / pointers in the counters), so the plan contains no memset nodes at all.
__global__ __launch_bounds__(256) void reset_counters_kernel(int* cnt, int n, int* status) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) cnt[i] = 0;
  if (threadIdx.x == 0) *status = 0;
}

// ---------------------------------------------------------------------------- NMS
// workspace layout (per call): see nms_layout()
struct NmsWs {
  int* cnt;            // [n]   candidates found (unclamped); [n .. 2n): set by nms_fused_kernel for the images it finished
  float4* cbox;        // [n][max_cand]  candidate boxes (xyxy), arrival order
  float4* cext;        // [n][max_cand]  obj, cls_conf, cls_id, score
  int* canchor;        // [n][max_cand]
  float* cscore;       // [n][max_cand]  score again, contiguous (scalar-cache friendly)
  float4* sbox;        // [n][max_cand]  sorted by score desc
  float4* sext;        // [n][max_cand]
  unsigned long long* mask;  // [n][nw][max_cand]  (column-block major; the class-segmented path: row-major words per class)
  unsigned long long* skey;  // [n][max_cand]  class-segmented path: sort keys (class | ~score | anchor) in segment order
  int* seg;                  // [n][GLS_NMSF_SEG]  class-segmented path: segment table per image
};

// utils_bbox.py:380-385 (cxcywh->xyxy), :398 class max, :403 threshold
__global__ __launch_bounds__(256) void nms_filter_kernel(const float* __restrict__ pred, int n, int A, int nc, int box_mode,
                                                         float thr, int max_cand, NmsWs ws, int* status) {
  const long total = (long)n * A;
  const int F = 5 + nc;
  const long padded = (total + 63) / 64 * 64;     // whole waves stay in the loop (ballots below)
  for (long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x; i0 < padded; i0 += (long)gridDim.x * blockDim.x) {
    const bool live = i0 < total;
    const long i = live ? i0 : total - 1;
    const int b = (int)(i / A), an = (int)(i - (long)b * A);
    const float* p = pred + i * F;
    float best = p[5];
    int arg = 0;
    for (int c = 1; c < nc; ++c) {
      const float v = p[5 + c];
      if (v > best) { best = v; arg = c; }       // first maximum wins, like torch.max
    }
    const float obj = p[4];
    const float score = obj * best;
    const bool pass = live && score >= thr;
    // one atomic per wave and image instead of one per candidate: lanes of this wave that
    // pass and belong to the same image as the first passing lane share one fetch-add.
    unsigned long long todo = __ballot(pass);
    int slot = -1;
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const int lb = __shfl(b, leader, 64);
      const unsigned long long grp = __ballot(pass && b == lb) & todo;
      int base0 = 0;
      if ((int)(threadIdx.x & 63) == leader) base0 = atomicAdd(&ws.cnt[lb], __popcll(grp));
      base0 = __shfl(base0, leader, 64);
      if (pass && b == lb) slot = base0 + __popcll(grp & ((1ull << (threadIdx.x & 63)) - 1ull));
      todo &= ~grp;
    }
    if (pass) {
      if (slot < max_cand) {
        float4 bx;
        if (box_mode == 0) {
          bx.x = p[0] - p[2] / 2; bx.y = p[1] - p[3] / 2; bx.z = p[0] + p[2] / 2; bx.w = p[1] + p[3] / 2;
        } else {
          bx.x = p[0]; bx.y = p[1]; bx.z = p[2]; bx.w = p[3];
        }
        const long o = (long)b * max_cand + slot;
        ws.cbox[o] = bx;
        ws.cext[o] = make_float4(obj, best, (float)arg, score);
        ws.canchor[o] = an;
        ws.cscore[o] = score;
      } else {
        atomicOr(status, 1);
      }
    }
  }
}

// rank sort: position of candidate i = #candidates that come before it in
// (score desc, anchor asc) order.  Exact and deterministic.  Workgroup = 64 candidates x 4
// slices of the comparison range; tiles of 256 (score, anchor) pairs go through LDS and each
// thread scans its quarter of the tile (broadcast reads), partial ranks are summed in LDS.
__global__ __launch_bounds__(256) void nms_rank_kernel(int max_cand, NmsWs ws, int fused = 0) {
  __shared__ float s_score[256];
  __shared__ int s_anchor[256];
  __shared__ int s_part[256];
  const int b = blockIdx.y;
  if (fused && ws.cnt[gridDim.y + b]) return;      // nms_fused_kernel handled this image
  const int n = min(ws.cnt[b], max_cand);
  if (blockIdx.x * 64 >= n) return;
  const int il = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + il;
  const long base = (long)b * max_cand;
  const float my = i < n ? ws.cscore[base + i] : 0.f;
  const int mya = i < n ? ws.canchor[base + i] : 0;
  int rank = 0;
  for (int j0 = 0; j0 < n; j0 += 256) {
    const int j = j0 + threadIdx.x;
    s_score[threadIdx.x] = j < n ? ws.cscore[base + j] : -INFINITY;     // never ranks before anything
    s_anchor[threadIdx.x] = j < n ? ws.canchor[base + j] : 0x7fffffff;
    __syncthreads();
#pragma unroll 16
    for (int t = slice * 64; t < slice * 64 + 64; ++t) {
      const float s = s_score[t];
      rank += (s > my) || (s == my && s_anchor[t] < mya);
    }
    __syncthreads();
  }
  s_part[threadIdx.x] = rank;
  __syncthreads();
  if (slice == 0 && i < n) {
    rank = s_part[il] + s_part[64 + il] + s_part[128 + il] + s_part[192 + il];
    ws.sbox[base + rank] = ws.cbox[base + i];
    ws.sext[base + rank] = ws.cext[base + i];
  }
}

// maskT[cb][i] bit t : candidate j = cb*64+t (j > i) has the class of i and IoU(i,j) > thr
// (torchvision nms: inter / (area_i + area_j - inter) > thr, areas without +1).
// Stored column-block major so that the scan reads 64 consecutive rows of one column block
// as one coalesced 512-byte wave load.  The grid is fixed; each 64-thread block walks the
// (row block, column block) pairs of the ACTUAL candidate count (known only on the device).
__global__ __launch_bounds__(64) void nms_mask_kernel(int max_cand, int nw, float thr, NmsWs ws, float one = 0.f, int fused = 0) {
  const int b = blockIdx.y;
  if (fused && ws.cnt[gridDim.y + b]) return;
  const int n = min(ws.cnt[b], max_cand);
  const int nb = (n + 63) >> 6;
  __shared__ float4 cbx[64];
  __shared__ float ccls[64];
  const long base = (long)b * max_cand;
  const int t = threadIdx.x;
  for (int p = blockIdx.x; p < nb * nb; p += gridDim.x) {
    const int rb = p / nb, cb = p - rb * nb;
    if (cb < rb) continue;                       // block-uniform
    const int j = cb * 64 + t;
    __syncthreads();
    if (j < n) {
      cbx[t] = ws.sbox[base + j];
      ccls[t] = ws.sext[base + j].z;
    }
    __syncthreads();
    const int i = rb * 64 + t;
    if (i < n) {
      const float4 a = ws.sbox[base + i];
      const float acls = ws.sext[base + i].z;
      const float aarea = (a.z - a.x + one) * (a.w - a.y + one);      // one = 1: the '+1' pixel-area convention
      unsigned long long bits = 0;
      const int lim = min(64, n - cb * 64);
      for (int k = 0; k < lim; ++k) {
        const int jj = cb * 64 + k;
        if (jj <= i || ccls[k] != acls) continue;
        const float4 c = cbx[k];
        const float w = fmaxf(0.f, fminf(a.z, c.z) - fmaxf(a.x, c.x) + one);
        const float h = fmaxf(0.f, fminf(a.w, c.w) - fmaxf(a.y, c.y) + one);
        const float inter = w * h;
        const float iou = inter / (aarea + (c.z - c.x + one) * (c.w - c.y + one) - inter);
        if (iou > thr) bits |= 1ull << k;
      }
      ws.mask[((long)b * nw + cb) * max_cand + i] = bits;
    }
  }
}

__device__ __forceinline__ unsigned long long wave_or(unsigned long long v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v |= __shfl_xor(v, o, 64);
  return v;
}

// Greedy NMS as a fixed point.  keep[i] = !exists j<i : keep[j] && M[j][i] has exactly one
// solution (induction on i) and it is what the sequential scan produces.  Iterating
//     keep'[i] = !OR_{j<i} (keep[j] & M[j][i])
// from keep = all-ones fixes at least one more prefix element per sweep and usually
// converges in a handful of sweeps (longest suppression chain), each sweep fully parallel:
// wave w ORs the coalesced column block maskT[w][0 .. 64(w+1)) under the current keep bits.
// A sweep that changes nothing proves the fixed point.  One 1024-thread workgroup / image.
#define GLS_NMS_MAXW 512    // 512 * 64 = 32768 candidates max
__global__ __launch_bounds__(1024) void nms_scan_kernel(int max_cand, int nw, int max_det, NmsWs ws, float* dets,
                                                        int* count, int fused = 0) {
  __shared__ unsigned long long s_keep[GLS_NMS_MAXW];
  __shared__ int s_prefix[GLS_NMS_MAXW];
  __shared__ int s_changed;
  const int b = blockIdx.x;
  if (fused && ws.cnt[gridDim.x + b]) return;
  const int n = min(ws.cnt[b], max_cand);
  const long base = (long)b * max_cand;
  const unsigned long long* maskT = ws.mask + (long)b * nw * max_cand;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
  const int words = (n + 63) >> 6;
  for (int w = tid; w < words; w += blockDim.x) {
    const int valid = min(64, n - w * 64);
    s_keep[w] = valid == 64 ? ~0ull : ((1ull << valid) - 1ull);
  }
  __syncthreads();
  for (int sweep = 0; sweep <= n; ++sweep) {
    if (tid == 0) s_changed = 0;
    __syncthreads();
    for (int w = wave; w < words; w += nwaves) {
      unsigned long long acc = 0ull;
      const int jend = min(n, (w + 1) * 64);
      const unsigned long long* col = maskT + (long)w * max_cand;
      for (int j = lane; j < jend; j += 64) {
        const unsigned long long kw = s_keep[j >> 6];
        if ((kw >> (j & 63)) & 1ull) acc |= col[j];
      }
      acc = wave_or(acc);
      if (lane == 0) {
        const int valid = min(64, n - w * 64);
        const unsigned long long all = valid == 64 ? ~0ull : ((1ull << valid) - 1ull);
        const unsigned long long nk = all & ~acc;
        if (nk != s_keep[w]) {
          s_keep[w] = nk;              // in place: any mix of old/new inputs has the same fixed point
          s_changed = 1;
        }
      }
    }
    __syncthreads();
    if (!s_changed) break;
    __syncthreads();
  }
  // exclusive prefix of kept counts per word, then every candidate writes its own row
  if (tid == 0) {
    int run = 0;
    for (int w = 0; w < words; ++w) {
      s_prefix[w] = run;
      run += __popcll(s_keep[w]);
    }
    count[b] = min(run, max_det);
    count[gridDim.x + b] = run;
  }
  __syncthreads();
  for (int i = tid; i < n; i += blockDim.x) {
    const unsigned long long kw = s_keep[i >> 6];
    if ((kw >> (i & 63)) & 1ull) {
      const int pos = s_prefix[i >> 6] + __popcll(kw & ((1ull << (i & 63)) - 1ull));
      if (pos < max_det) {
        const float4 bx = ws.sbox[base + i];
        const float4 ex = ws.sext[base + i];
        float* d = dets + ((long)b * max_det + pos) * 7;
        d[0] = bx.x; d[1] = bx.y; d[2] = bx.z; d[3] = bx.w; d[4] = ex.x; d[5] = ex.y; d[6] = ex.z;
      }
    }
  }
}


// ---------------------------------------------------------------------------- class-segmented NMS (round 3)
// filter -> rank -> mask -> scan was 0.21 ms of a 2.7 ms serial step: an O(n^2) rank sort over the chip, an n x n IoU bitmask
// over ALL candidate pairs although only same-class pairs can suppress (utils_bbox.py:413-419 batched_nms), and a fixed-point
// scan that re-reads the whole mask once per sweep.  Now:
//   nms_sort_kernel (one 1024-thread workgroup per image): bitonic sort in LDS by (class, score desc, anchor asc) -- the
//      candidates of a class become one contiguous segment, in the order greedy NMS visits them; boxes / extras / keys are
//      stored in that order together with the segment table (start, length, mask offset, task prefix per class);
//   nms_cmask_kernel (chip wide): IoU bitmask of the same-class block pairs only (a tenth of the pairs for ten balanced
//      classes), a 64-thread workgroup per (class, row block, column block) task;
//   nms_cscan_kernel (one workgroup per image): greedy scan, one wave per class, blocked and right-looking -- inside a
//      64-candidate block the keep bits are resolved sequentially with v_readlane on the block's diagonal words, then the
//      kept rows' words are OR-ed into the dead masks of the later blocks: every mask word is read ONCE; the kept candidates
//      are compacted, bitonic-sorted by (score desc, anchor asc) -- the order torchvision's batched_nms returns and the one
//      the old path produces -- and written out.
// (First cut of the round: all three phases in ONE workgroup per image.  Correct, and slower than the old path -- 0.24 vs
// 0.21 ms: 200 k IoUs per image are too much for one CU.)  Same keep set and order as the old path by construction:
// suppression only acts inside a class, and inside a class both visit the candidates in (score desc, anchor asc) order.
// Images with more than GLS_NMSF_MAX candidates (or anchors / classes beyond the key's bit fields) are left to the old
// kernels, which skip the images handled here.
#define GLS_NMSF_MAX 4096
#define GLS_NMSF_SEG (256 + 256 + 260 + 260)      // ints per image: segment start, length, mask offset, task prefix
__device__ __forceinline__ unsigned f2ord(float f) {            // float -> unsigned with the same order (also for negative / zero scores)
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
// Bitonic sort of P = 1024 * E (key, value) pairs held E per thread (element i = e * 1024 + tid) by 1024 threads.  A plain
// LDS network with a barrier per stage cost ~0.9 us per stage (66 stages for 2048 elements: 61 us, rocprofv3); here a
// compare-exchange with stride j >= 1024 stays inside the thread's registers, j < 64 is a lane shuffle, and only the
// strides 64 .. 512 (14 of the 66 stages at P = 2048) go through LDS with barriers.  Ascending; keys must be distinct.
template <int E>
__device__ __forceinline__ void bitonic_sort_regs(unsigned long long (&key)[E], unsigned short (&val)[E], unsigned long long* xk,
                                                  unsigned short* xv, int tid) {
  constexpr int P = 1024 * E;
#pragma unroll 1
  for (int k = 2; k <= P; k <<= 1) {
#pragma unroll 1
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j >= 1024) {                              // partner in this thread (STATIC register indices: a runtime index would
        auto cx = [&](auto ea, auto eb) {           // send the arrays to scratch memory)
          constexpr int A_ = decltype(ea)::value, B_ = decltype(eb)::value;
          if constexpr (B_ < E) {
            const bool up = ((A_ * 1024 + tid) & k) == 0;
            if ((key[A_] > key[B_]) == up) {
              const unsigned long long t = key[A_]; key[A_] = key[B_]; key[B_] = t;
              const unsigned short tv = val[A_]; val[A_] = val[B_]; val[B_] = tv;
            }
          }
        };
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
        if (j == 1024) { cx(I0{}, I1{}); cx(I2{}, I3{}); }
        else { cx(I0{}, I2{}); cx(I1{}, I3{}); }
      } else if (j >= 64) {                         // partner in another wave: through LDS
#pragma unroll
        for (int e = 0; e < E; ++e) {
          xk[e * 1024 + tid] = key[e];
          xv[e * 1024 + tid] = val[e];
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const int i = e * 1024 + tid;
          const unsigned long long pk = xk[i ^ j];
          const unsigned short pv = xv[i ^ j];
          const bool keepmin = ((i & k) == 0) == ((i & j) == 0);
          if (keepmin ? (pk < key[e]) : (pk > key[e])) {
            key[e] = pk;
            val[e] = pv;
          }
        }
        __syncthreads();
      } else {                                      // partner lane: shuffles
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const int i = e * 1024 + tid;
          const unsigned lo = __shfl_xor((unsigned)key[e], j, 64), hi = __shfl_xor((unsigned)(key[e] >> 32), j, 64);
          const unsigned pv = __shfl_xor((unsigned)val[e], j, 64);
          const unsigned long long pk = ((unsigned long long)hi << 32) | lo;
          const bool keepmin = ((i & k) == 0) == ((i & j) == 0);
          if (keepmin ? (pk < key[e]) : (pk > key[e])) {
            key[e] = pk;
            val[e] = (unsigned short)pv;
          }
        }
      }
    }
  }
}
// sort the first P (1024, 2048 or 4096) entries of key / val (LDS) in place; all 1024 threads call it
__device__ __forceinline__ void bitonic_sort_kv(unsigned long long* key, unsigned short* val, int P, int tid) {
#define GLS_SORT_E(E_)                                                              \
  {                                                                                 \
    unsigned long long k_[E_];                                                      \
    unsigned short v_[E_];                                                          \
    for (int e = 0; e < E_; ++e) { k_[e] = key[e * 1024 + tid]; v_[e] = val[e * 1024 + tid]; } \
    __syncthreads();                                                                \
    bitonic_sort_regs<E_>(k_, v_, key, val, tid);                                   \
    for (int e = 0; e < E_; ++e) { key[e * 1024 + tid] = k_[e]; val[e * 1024 + tid] = v_[e]; } \
    __syncthreads();                                                                \
  }
  if (P <= 1024) GLS_SORT_E(1)
  else if (P <= 2048) GLS_SORT_E(2)
  else GLS_SORT_E(4)
#undef GLS_SORT_E
}
// exclusive prefix sum of v over the 1024 threads of a workgroup (every thread calls it); total in *tot.  Wave shuffles +
// ONE exchange of the 16 wave totals through LDS: a 1024-thread __syncthreads costs ~0.8 us here (rocprofv3: 66 barrier
// stages = 61 us), so the Hillis-Steele scans of the first cut (20 barriers each) were most of the kernels' time.
__device__ __forceinline__ int block_scan_excl(int v, int* s_w /* [16] */, int tid, int* tot) {
  const int lane = tid & 63, wave = tid >> 6;
  int x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int y = __shfl_up(x, o, 64);
    if (lane >= o) x += y;
  }
  __syncthreads();                                 // (s_w may still be read from the previous call)
  if (lane == 63) s_w[wave] = x;
  __syncthreads();
  int before = 0, all = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    const int t = s_w[w];
    all += t;
    if (w < wave) before += t;
  }
  *tot = all;
  return before + x - v;
}
__device__ __forceinline__ int scan256_excl(int v, int* s_w, int tid, int* tot) {      // v must be 0 for tid >= 256
  return block_scan_excl(tid < 256 ? v : 0, s_w, tid, tot);
}
__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int l) {
  const unsigned lo = __builtin_amdgcn_readlane((unsigned)v, l), hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), l);
  return ((unsigned long long)hi << 32) | lo;
}

__global__ __launch_bounds__(1024) void nms_sort_kernel(int max_cand, int key_ok, NmsWs ws, int dbg = 0) {
  __shared__ unsigned long long s_key[GLS_NMSF_MAX];
  __shared__ unsigned short s_idx[GLS_NMSF_MAX];
  __shared__ int s_seg[256], s_segn[256];
  const int b = blockIdx.x, nimg = gridDim.x, tid = threadIdx.x;
  const int n = min(ws.cnt[b], max_cand);
  int* done = ws.cnt + nimg;
  if (n > GLS_NMSF_MAX || !key_ok) {               // left to nms_rank / nms_mask / nms_scan
    if (tid == 0) done[b] = 0;
    return;
  }
  if (tid == 0) done[b] = 1;
  int* seg = ws.seg + (long)b * GLS_NMSF_SEG;
  int P = 1024;
  while (P < n) P <<= 1;
  const long base = (long)b * max_cand;
  for (int i = tid; i < P; i += 1024) {
    unsigned long long k = ~0ull - (unsigned)i;     // padding: distinct keys behind every real one (class byte 0xff)
    if (i < n) {
      const float4 ex = ws.cext[base + i];
      k = ((unsigned long long)(unsigned)(int)ex.z << 56) | ((unsigned long long)(~f2ord(ex.w)) << 24) | (unsigned long long)(unsigned)ws.canchor[base + i];
    }
    s_key[i] = k;
    s_idx[i] = (unsigned short)i;
  }
  if (tid < 256) s_seg[tid] = s_segn[tid] = 0;
  __syncthreads();
  if (!(dbg & 1)) bitonic_sort_kv(s_key, s_idx, P, tid);
  if (!(dbg & 2))
  for (int i = tid; i < n; i += 1024) {
    const int o = s_idx[i];
    ws.sbox[base + i] = ws.cbox[base + o];
    ws.sext[base + i] = ws.cext[base + o];
    ws.skey[base + i] = s_key[i];
    const int c = (int)(s_key[i] >> 56);
    if (i == 0 || (int)(s_key[i - 1] >> 56) != c) s_seg[c] = i;
    if (i == n - 1 || (int)(s_key[i + 1] >> 56) != c) s_segn[c] = i + 1;
  }
  __syncthreads();
  int len = 0, nb = 0;
  if (tid < 256) {
    len = s_segn[tid] - s_seg[tid];                // length (0 for absent classes: both zero)
    nb = (len + 63) >> 6;
    seg[tid] = s_seg[tid];
    seg[256 + tid] = len;
  }
  __shared__ int s_pre[256];
  int tot_m, tot_t;
  const int mo = scan256_excl(len * nb, s_pre, tid, &tot_m);             // mask word offset of the class
  const int to = scan256_excl(nb * (nb + 1) / 2, s_pre, tid, &tot_t);    // task prefix
  if (tid < 256) {
    seg[512 + tid] = mo;
    seg[512 + 260 + tid] = to;
  }
  if (tid == 0) {
    seg[512 + 256] = tot_m;
    seg[512 + 260 + 256] = tot_t;
  }
}

// one 64-thread workgroup per (class, row block rb <= column block cb) task of an image; the grid is fixed, a workgroup walks
// the tasks of the ACTUAL segment table (known only on the device)
__global__ __launch_bounds__(64) void nms_cmask_kernel(int max_cand, int nw, float thr, float one, NmsWs ws) {
  const int b = blockIdx.y, nimg = gridDim.y;
  if (!ws.cnt[nimg + b]) return;
  const int* seg = ws.seg + (long)b * GLS_NMSF_SEG;
  const int* toff = seg + 512 + 260;
  const int ntask = toff[256];
  __shared__ float4 cbx[64];
  const long base = (long)b * max_cand;
  unsigned long long* mask = ws.mask + (long)b * nw * max_cand;      // nw * max_cand >= sum n_c * nb_c words
  const int lane = threadIdx.x;
  for (int task = blockIdx.x; task < ntask; task += gridDim.x) {
    int lo = 0, hi = 255;                          // last class whose task prefix is <= task
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (toff[mid] <= task) lo = mid;
      else hi = mid - 1;
    }
    const int c = lo, s = seg[c], nc_ = seg[256 + c], nb = (nc_ + 63) >> 6;
    int local = task - toff[c], rb = 0;
    while (local >= nb - rb) {
      local -= nb - rb;
      ++rb;
    }
    const int cb = rb + local;
    __syncthreads();
    if (cb * 64 + lane < nc_) cbx[lane] = ws.sbox[base + s + cb * 64 + lane];
    __syncthreads();
    const int ri = rb * 64 + lane;                 // segment-local row
    if (ri < nc_) {
      const float4 a = ws.sbox[base + s + ri];
      const float aarea = (a.z - a.x + one) * (a.w - a.y + one);
      unsigned long long bits = 0;
      const int lim = min(64, nc_ - cb * 64);
      for (int k = 0; k < lim; ++k) {
        if (cb * 64 + k <= ri) continue;
        const float4 q = cbx[k];
        const float w = fmaxf(0.f, fminf(a.z, q.z) - fmaxf(a.x, q.x) + one);
        const float h = fmaxf(0.f, fminf(a.w, q.w) - fmaxf(a.y, q.y) + one);
        const float inter = w * h;
        const float iou = inter / (aarea + (q.z - q.x + one) * (q.w - q.y + one) - inter);
        if (iou > thr) bits |= 1ull << k;
      }
      mask[seg[512 + c] + (long)ri * nb + cb] = bits;
    }
  }
}

__global__ __launch_bounds__(1024) void nms_cscan_kernel(int max_cand, int nw, int max_det, NmsWs ws, float* dets, int* count, int dbg = 0) {
  __shared__ unsigned long long s_key2[GLS_NMSF_MAX];
  __shared__ unsigned short s_val[GLS_NMSF_MAX];
  __shared__ unsigned char s_flag[GLS_NMSF_MAX];
  __shared__ int s_scan[1024];
  const int b = blockIdx.x, nimg = gridDim.x;
  if (!ws.cnt[nimg + b]) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = min(ws.cnt[b], max_cand);
  const int* seg = ws.seg + (long)b * GLS_NMSF_SEG;
  const long base = (long)b * max_cand;
  const unsigned long long* mask = ws.mask + (long)b * nw * max_cand;
  __shared__ int s_cs[256], s_cn[256], s_cm[256];   // segment table of this image: start, length, mask word offset per class
  if (tid < 256) {
    s_cs[tid] = seg[tid];
    s_cn[tid] = seg[256 + tid];
    s_cm[tid] = seg[512 + tid];
  }
  for (int i = tid; i < n; i += 1024) s_flag[i] = 0;
  __syncthreads();
  // greedy scan, a wave per class
  if (!(dbg & 4))
  for (int c = wave; c < 256; c += 16) {
    const int nc_ = s_cn[c];
    if (nc_ == 0) continue;
    const int s = s_cs[c], nb = (nc_ + 63) >> 6;
    const unsigned long long* M = mask + s_cm[c];
    unsigned long long dead_reg = 0ull;            // lane L: OR of the kept earlier rows' words for column block L
    for (int wb = 0; wb < nb; ++wb) {
      const int ri = wb * 64 + lane;
      const bool valid = ri < nc_;
      const unsigned long long word = valid ? M[(long)ri * nb + wb] : 0ull;
      const int left = nc_ - wb * 64;
      unsigned long long alive = (left >= 64 ? ~0ull : ((1ull << left) - 1ull)) & ~readlane64(dead_reg, wb);
      for (int t = 0; t < 64; ++t)
        if ((alive >> t) & 1ull) alive &= ~readlane64(word, t);
      const bool kept = valid && ((alive >> lane) & 1ull);
      if (kept) s_flag[s + ri] = 1;
      for (int w2 = wb + 1; w2 < nb; ++w2) {
        const unsigned long long v = kept ? M[(long)ri * nb + w2] : 0ull;
        const unsigned long long r = wave_or(v);
        if (lane == w2) dead_reg |= r;
      }
    }
  }
  __syncthreads();
  // The kept candidates in class-grouped order are <= 256 runs, each already sorted by (score desc, anchor asc): the output
  // position of a kept candidate is the number of kept candidates with a smaller key -- its index inside its own run plus a
  // binary search in every other non-empty run (ten classes: ~70 LDS reads per candidate instead of a 55-stage sort).
  int P = 1024;
  while (P < n) P <<= 1;
  const int per = P >> 10;                         // consecutive positions per thread (1, 2 or 4)
  int mine = 0;
  for (int e = 0; e < per; ++e) {
    const int p = tid * per + e;
    if (p < n && s_flag[p]) ++mine;
  }
  int K;
  const int q0 = block_scan_excl(mine, s_scan, tid, &K);
  {
    int q = q0;
    for (int e = 0; e < per; ++e) {
      const int p = tid * per + e;
      if (p < n) {
        s_val[p] = (unsigned short)q;              // kept candidates before position p (= compacted index of p if kept)
        if (s_flag[p]) {
          s_key2[q] = ws.skey[base + p] & 0x00ffffffffffffffull;      // (score desc, anchor asc)
          ++q;
        }
      }
    }
  }
  __syncthreads();
  // run table: compacted start and length of every class with kept candidates
  __shared__ int s_rs[256], s_rl[256], s_nrun;
  if (tid < 256) {
    const int nc_ = s_cn[tid], st = s_cs[tid];
    int rs = 0, rl = 0;
    if (nc_ > 0) {
      rs = s_val[st];
      const int last = st + nc_ - 1;
      rl = s_val[last] + (s_flag[last] ? 1 : 0) - rs;
    }
    s_rs[tid] = rs;
    s_rl[tid] = rl;
  }
  __syncthreads();
  {                                                // pack the non-empty runs to the front
    const int a = tid < 256 ? s_rs[tid] : 0, l = tid < 256 ? s_rl[tid] : 0;
    int tot;
    const int m = scan256_excl(l > 0 ? 1 : 0, s_scan, tid, &tot);
    if (tid < 256 && l > 0) {
      s_rs[m] = a;
      s_rl[m] = l;
    }
    if (tid == 0) s_nrun = tot;
    __syncthreads();
  }
  if (tid == 0) {
    count[b] = min(K, max_det);
    count[nimg + b] = K;
  }
  const int nrun = s_nrun;
  if (!(dbg & 8))
  for (int e = 0; e < per; ++e) {
    const int p = tid * per + e;
    if (p < n && s_flag[p]) {
      const int q = s_val[p];
      const unsigned long long kq = s_key2[q];
      int pos = 0;
      for (int r = 0; r < nrun; ++r) {
        const int a = s_rs[r], l = s_rl[r];
        if (q >= a && q < a + l) {
          pos += q - a;                            // its own run
        } else {
          int lo = 0, hi = l;                      // elements of run r with a key below kq
          while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (s_key2[a + mid] < kq) lo = mid + 1;
            else hi = mid;
          }
          pos += lo;
        }
      }
      if (pos < max_det) {
        const float4 bx = ws.sbox[base + p];
        const float4 ex = ws.sext[base + p];
        float* d = dets + ((long)b * max_det + pos) * 7;
        d[0] = bx.x; d[1] = bx.y; d[2] = bx.z; d[3] = bx.w; d[4] = ex.x; d[5] = ex.y; d[6] = ex.z;
      }
    }
  }
}

// ---------------------------------------------------------------------------- GFL / MPHead
// gfl_head.py:380-471 per (image, level): sigmoid scores, (position, class) pairs above the
// threshold, Integral (softmax expectation over reg_max+1 bins) * stride, distance2bbox from
// the anchor centre, clamp to the image.  grid = (position blocks, image, level); a wave
// allocates its candidates with one atomic (prefix sum of the lanes' pass counts).
struct GflArgs {
  const float* cls[GLS_MAX_LEVELS];
  const float* reg[GLS_MAX_LEVELS];
  long csn[GLS_MAX_LEVELS], csh[GLS_MAX_LEVELS], csw[GLS_MAX_LEVELS];
  long rsn[GLS_MAX_LEVELS], rsh[GLS_MAX_LEVELS], rsw[GLS_MAX_LEVELS];
  int H[GLS_MAX_LEVELS], W[GLS_MAX_LEVELS];
  float stride[GLS_MAX_LEVELS];
  int n_levels, nc, reg_max, n;
  float in_h, in_w, thr;
  const float* img_hw;   // optional [n][2]
};

__global__ __launch_bounds__(256) void gfl_filter_kernel(const GflArgs a, int max_cand, NmsWs ws, int* status) {
  const int l = blockIdx.z, b = blockIdx.y;
  const int HW = a.H[l] * a.W[l];
  if ((int)(blockIdx.x * blockDim.x) >= HW) return;           // block-uniform
  const int pos = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = pos < HW;
  const int gy = live ? pos / a.W[l] : 0, gx = live ? pos - gy * a.W[l] : 0;
  const float* pc = a.cls[l] + b * a.csn[l] + gy * a.csh[l] + gx * a.csw[l];
  int npass = 0;
  if (live)
    for (int c = 0; c < a.nc; ++c) npass += sigmoidf_acc(pc[c]) > a.thr;
  // wave-inclusive prefix sum of npass
  const int lane = threadIdx.x & 63;
  int incl = npass;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o, 64);
    if (lane >= o) incl += v;
  }
  const int total = __shfl(incl, 63, 64);
  if (total == 0) return;                                       // wave-uniform
  const int slot_list = b * a.n_levels + l;
  int base0 = 0;
  if (lane == 0) base0 = atomicAdd(&ws.cnt[slot_list], total);
  base0 = __shfl(base0, 0, 64);
  if (npass == 0) return;
  int slot = base0 + incl - npass;
  // Integral (gfl_head.py:16-49) and distance2bbox (core/bbox/transforms.py:153-165)
  const float* pr = a.reg[l] + b * a.rsn[l] + gy * a.rsh[l] + gx * a.rsw[l];
  const int bins = a.reg_max + 1;
  float d[4];
  if (bins == 17 && (((uintptr_t)pr) & 15) == 0) {
    // reg_max = 16 (the GFL / MPDet configs): the 68 logits of the position are 17 aligned float4 -- all requested before the
    // first is used.  (The generic loop below walks them one dependent load after the other; with a few lanes of a wave
    // alive here that chain, ~200 loads long, was the critical path of the launch: 61 us for 7 MB.)
    float4 r4[17];
#pragma unroll
    for (int i = 0; i < 17; ++i) r4[i] = reinterpret_cast<const float4*>(pr)[i];
    const float* r = reinterpret_cast<const float*>(r4);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float m = r[s * 17];
#pragma unroll
      for (int k = 1; k < 17; ++k) m = fmaxf(m, r[s * 17 + k]);
      float den = 0.f, num = 0.f;
#pragma unroll
      for (int k = 0; k < 17; ++k) {
        const float e = expf(r[s * 17 + k] - m);
        den += e;
        num += e * (float)k;
      }
      d[s] = num / den * a.stride[l];
    }
  } else {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float* q = pr + s * bins;
      float m = q[0];
      for (int k = 1; k < bins; ++k) m = fmaxf(m, q[k]);
      float den = 0.f, num = 0.f;
      for (int k = 0; k < bins; ++k) {
        const float e = expf(q[k] - m);
        den += e;
        num += e * (float)k;
      }
      d[s] = num / den * a.stride[l];
    }
  }
  const float cx = (float)gx * a.stride[l], cy = (float)gy * a.stride[l];
  const float mh = a.img_hw ? a.img_hw[2 * b] : a.in_h, mw = a.img_hw ? a.img_hw[2 * b + 1] : a.in_w;
  float4 bx;
  bx.x = fminf(fmaxf(cx - d[0], 0.f), mw);
  bx.y = fminf(fmaxf(cy - d[1], 0.f), mh);
  bx.z = fminf(fmaxf(cx + d[2], 0.f), mw);
  bx.w = fminf(fmaxf(cy + d[3], 0.f), mh);
  for (int c = 0; c < a.nc; ++c) {
    const float sc = sigmoidf_acc(pc[c]);
    if (!(sc > a.thr)) continue;
    if (slot < max_cand) {
      const long o = (long)slot_list * max_cand + slot;
      ws.cbox[o] = bx;
      ws.cext[o] = make_float4(sc, sc, (float)c, sc);
      ws.canchor[o] = pos * a.nc + c;
      ws.cscore[o] = sc;
    } else {
      atomicOr(status, 1);
    }
    ++slot;
  }
}

// filter_scores_and_topk's top-k + the level concatenation + `rescale` (base_dense_head.py:
// 270-272): the nms_pre best of every level's sorted list become the image's candidates.
__global__ __launch_bounds__(256) void gfl_merge_kernel(int n_levels, int max_cand1, int nms_pre, int max_cand2, NmsWs w1,
                                                        NmsWs w2, const float* scale) {
  const int b = blockIdx.y;
  int start = 0, l = 0, take = 0;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  int total = 0;
  for (int t = 0; t < n_levels; ++t) total += min(min(w1.cnt[b * n_levels + t], max_cand1), nms_pre);
  if (i == 0) w2.cnt[b] = total;
  if (i >= total) return;
  for (l = 0; l < n_levels; ++l) {
    take = min(min(w1.cnt[b * n_levels + l], max_cand1), nms_pre);
    if (i < start + take) break;
    start += take;
  }
  const long src = (long)(b * n_levels + l) * max_cand1 + (i - start);
  const long dst = (long)b * max_cand2 + i;
  float4 bx = w1.sbox[src];
  if (scale) {
    const float* sf = scale + 4 * b;
    bx.x /= sf[0]; bx.y /= sf[1]; bx.z /= sf[2]; bx.w /= sf[3];
  }
  const float4 ex = w1.sext[src];
  w2.cbox[dst] = bx;
  w2.cext[dst] = ex;
  w2.canchor[dst] = i;
  w2.cscore[dst] = ex.w;
}

// ---------------------------------------------------------------------------- UFP back-mapping
// ufp/ufpmp_det_eval.py:282-296: a fine detection whose box lies in a chip's mosaic rectangle
// (intersection over the smaller area > iof_thr) is mapped back through the chip's magnification
// and offsets.  One candidate per (chip, detection) match; canchor = -(chip * max_det + det) so that,
// for equal scores, the later entry of the reference's list ranks first (its argsort()[::-1]).
__global__ __launch_bounds__(256) void ufp_backmap_kernel(const float* __restrict__ dets, const int* __restrict__ count,
                                                          int max_det, const float* __restrict__ chips, int n_chips,
                                                          float iof_thr, int max_cand, NmsWs ws, int* status) {
  const int nd = min(count[0], max_det);
  const long total = (long)n_chips * nd;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int k = (int)(i / nd), j = (int)(i - (long)k * nd);
    const float* c = chips + 7 * k;
    const float ox = floorf(c[0]), oy = floorf(c[1]), w = floorf(c[2]), h = floorf(c[3]);
    const float nx = floorf(c[4]), ny = floorf(c[5]), s = floorf(c[6]);
    const float* d = dets + (long)j * 7;
    const float rx2 = nx + w * s, ry2 = ny + h * s;
    const float l = fmaxf(d[0], nx), t = fmaxf(d[1], ny), r = fminf(d[2], rx2), b = fminf(d[3], ry2);
    if (l >= r || t >= b) continue;
    const float iof = (r - l) * (b - t) / fminf((d[2] - d[0]) * (d[3] - d[1]), (rx2 - nx) * (ry2 - ny));
    if (!(iof > iof_thr)) continue;
    const int slot = atomicAdd(&ws.cnt[0], 1);
    if (slot >= max_cand) {
      atomicOr(status, 1);
      continue;
    }
    const float bw = (d[2] - d[0]) / s, bh = (d[3] - d[1]) / s;
    const float bx = (d[0] - nx) / s + ox, by = (d[1] - ny) / s + oy;
    ws.cbox[slot] = make_float4(bx, by, bx + bw, by + bh);
    ws.cext[slot] = make_float4(d[4], d[5], d[6], d[4]);
    ws.canchor[slot] = -(k * max_det + j);
    ws.cscore[slot] = d[4];
  }
}

static inline long align_up(long v, long a) { return (v + a - 1) / a * a; }
static long nms_layout(int n, int max_cand, NmsWs* ws, char* base, bool with_mask = true) {
  const int nw = (max_cand + 63) / 64;
  long off = 0;
  auto take = [&](long bytes) { long o = off; off = align_up(off + bytes, 256); return o; };
  const long o_cnt = take((long)n * 8);            // [n] candidate counts + [n] "the fused kernel handled this image" flags
  const long o_cbox = take((long)n * max_cand * 16);
  const long o_cext = take((long)n * max_cand * 16);
  const long o_can = take((long)n * max_cand * 4);
  const long o_csc = take((long)n * max_cand * 4);
  const long o_sbox = take((long)n * max_cand * 16);
  const long o_sext = take((long)n * max_cand * 16);
  const long o_mask = with_mask ? take((long)n * max_cand * nw * 8) : 0;
  const long o_skey = take((long)n * max_cand * 8);
  const long o_seg = take((long)n * (256 + 256 + 260 + 260) * 4);
  if (ws && base) {
    ws->cnt = (int*)(base + o_cnt);
    ws->cbox = (float4*)(base + o_cbox);
    ws->cext = (float4*)(base + o_cext);
    ws->canchor = (int*)(base + o_can);
    ws->cscore = (float*)(base + o_csc);
    ws->sbox = (float4*)(base + o_sbox);
    ws->sext = (float4*)(base + o_sext);
    ws->mask = (unsigned long long*)(base + o_mask);
    ws->skey = (unsigned long long*)(base + o_skey);
    ws->seg = (int*)(base + o_seg);
  }
  return off;
}

}  // namespace glsdet

using namespace glsdet;

extern "C" int glsdet_yolox_decode(const glsdet_view* levels, int32_t n_levels, int32_t num_classes, int32_t in_h,
                                   int32_t in_w, const int32_t* strides, int32_t mode, float* out, int64_t out_elems,
                                   const float* scale_factors, void* stream) {
  if (!levels || !out || n_levels < 1 || n_levels > GLS_MAX_LEVELS) GLS_FAIL(GLSDET_E_ARG, "yolox_decode: bad levels");
  if (num_classes < 1 || mode < 0 || mode > 1) GLS_FAIL(GLSDET_E_ARG, "yolox_decode: bad num_classes/mode");
  DecodeArgs a = {};
  int A = 0;
  for (int l = 0; l < n_levels; ++l) {
    int rc;
    if ((rc = check_view(levels[l], "yolox_decode.level", false))) return rc;
    if (levels[l].dtype != GLSDET_F32) GLS_FAIL(GLSDET_E_ARG, "yolox_decode: levels must be fp32");
    if (levels[l].c < 5 + num_classes || levels[l].n != levels[0].n)
      GLS_FAIL(GLSDET_E_ARG, "yolox_decode: level %d has %d channels, need >= %d", l, levels[l].c, 5 + num_classes);
    a.base[l] = (const float*)levels[l].base;
    a.sn[l] = levels[l].sn; a.sh[l] = levels[l].sh; a.sw[l] = levels[l].sw;
    a.H[l] = levels[l].h; a.W[l] = levels[l].w;
    a.start[l] = A;
    A += levels[l].h * levels[l].w;
    // reference quirk (utils_bbox.py:285): stride = input_shape[0] / h for BOTH axes
    const float s = strides ? (float)strides[l] : (float)in_h / (float)levels[l].h;
    a.stride_x[l] = a.stride_y[l] = s;
  }
  a.start[n_levels] = A;
  a.n_levels = n_levels; a.nc = num_classes; a.n = levels[0].n; a.A = A; a.mode = mode;
  a.in_w = (float)in_w; a.in_h = (float)in_h;
  a.out = out;
  a.scale = mode == 1 ? scale_factors : nullptr;
  if (out_elems < (int64_t)a.n * A * (5 + num_classes)) GLS_FAIL(GLSDET_E_CAPACITY, "yolox_decode: output buffer too small");
  OpRecord op;
  op.kind = 5;
  op.flops = 0;
  op.bytes = (double)a.n * A * (5 + num_classes) * 8.0;
  op.name = "yolox_decode";
  op.launch = [a](hipStream_t st) -> int {
    long g = ((long)a.n * a.A + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(decode_kernel, dim3((unsigned)g), dim3(256), 0, st, a);
    GLS_HIP(hipGetLastError());
    return 0;
  };
  return submit(std::move(op), stream);
}

extern "C" int64_t glsdet_nms_workspace_bytes(int32_t n, int32_t A, int32_t max_cand) {
  (void)A;
  if (n < 1 || max_cand < 1) return 0;
  return nms_layout(n, max_cand, nullptr, nullptr);
}

extern "C" int glsdet_nms(const float* pred, int32_t n, int32_t A, int32_t num_classes, int32_t box_mode,
                          float conf_thres, float nms_thres, int32_t max_cand, int32_t max_det, float* dets,
                          int32_t* count, int32_t* status, void* wsp, int64_t ws_bytes, void* stream) {
  if (!pred || !dets || !count || !status || !wsp) GLS_FAIL(GLSDET_E_ARG, "nms: null argument");
  if (n < 1 || A < 1 || num_classes < 1 || max_cand < 1 || max_det < 1) GLS_FAIL(GLSDET_E_ARG, "nms: bad sizes");
  if (max_cand > GLS_NMS_MAXW * 64) GLS_FAIL(GLSDET_E_ARG, "nms: max_cand %d > %d", max_cand, GLS_NMS_MAXW * 64);
  if ((uintptr_t)wsp & 255) GLS_FAIL(GLSDET_E_ALIGN, "nms: workspace must be 256-byte aligned");
  NmsWs ws;
  const long need = nms_layout(n, max_cand, &ws, (char*)wsp);
  if (ws_bytes < need) GLS_FAIL(GLSDET_E_CAPACITY, "nms: workspace %ld < %ld bytes", (long)ws_bytes, need);
  const int nw = (max_cand + 63) / 64;
  OpRecord op;
  op.kind = 6;
  op.flops = 0;
  op.bytes = (double)n * A * (5 + num_classes) * 4.0;
  static const bool no_fused = getenv("GLSDET_NO_FUSED_NMS") != nullptr;       // A/B switch: the three-kernel path of rounds 1-2
  const int key_ok = (A < (1 << 24) && num_classes <= 255) ? 1 : 0;           // bit fields of the fused kernel's sort key
  op.name = no_fused ? "nms(filter+rank+mask+scan)" : "nms(filter + class sort + same-class mask + blocked scan)";
  op.launch = [=](hipStream_t st) -> int {
    hipLaunchKernelGGL(reset_counters_kernel, dim3(1), dim3(256), 0, st, ws.cnt, 2 * n, status);
    long g = ((long)n * A + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(nms_filter_kernel, dim3((unsigned)g), dim3(256), 0, st, pred, n, A, num_classes, box_mode,
                       conf_thres, max_cand, ws, status);
    int fused = 0;
    if (!no_fused) {
      const int dbg = getenv("GLSDET_NMS_DBG") ? atoi(getenv("GLSDET_NMS_DBG")) : 0;      // timing knock-outs (results invalid)
      hipLaunchKernelGGL(nms_sort_kernel, dim3(n), dim3(1024), 0, st, max_cand, key_ok, ws, dbg);
      hipLaunchKernelGGL(nms_cmask_kernel, dim3(512, n), dim3(64), 0, st, max_cand, nw, nms_thres, 0.f, ws);
      hipLaunchKernelGGL(nms_cscan_kernel, dim3(n), dim3(1024), 0, st, max_cand, nw, max_det, ws, dets, count, dbg);
      fused = 1;
    }
    if (!fused || max_cand > GLS_NMSF_MAX || !key_ok) {      // images the fused kernel cannot take (its flags decide per image)
      hipLaunchKernelGGL(nms_rank_kernel, dim3((max_cand + 63) / 64, n), dim3(256), 0, st, max_cand, ws, fused);
      hipLaunchKernelGGL(nms_mask_kernel, dim3(2048, n), dim3(64), 0, st, max_cand, nw, nms_thres, ws, 0.f, fused);
      hipLaunchKernelGGL(nms_scan_kernel, dim3(n), dim3(1024), 0, st, max_cand, nw, max_det, ws, dets, count, fused);
    }
    GLS_HIP(hipGetLastError());
    return 0;
  };
  return submit(std::move(op), stream);
}

extern "C" int64_t glsdet_gfl_workspace_bytes(int32_t n, int32_t n_levels, int32_t max_cand, int32_t nms_pre) {
  if (n < 1 || n_levels < 1 || n_levels > GLS_MAX_LEVELS || max_cand < 1 || nms_pre < 1) return 0;
  return nms_layout(n * n_levels, max_cand, nullptr, nullptr, false) + 256 +
         nms_layout(n, n_levels * nms_pre, nullptr, nullptr, true);
}

extern "C" int glsdet_gfl_detect(const glsdet_view* cls, const glsdet_view* reg, int32_t n_levels,
                                 const int32_t* strides, int32_t num_classes, int32_t reg_max, int32_t in_h,
                                 int32_t in_w, const float* img_hw, const float* scale_factors, float score_thr,
                                 int32_t nms_pre, float iou_thr, int32_t max_cand, int32_t max_det, float* dets,
                                 int32_t* count, int32_t* status, void* wsp, int64_t ws_bytes, void* stream) {
  if (!cls || !reg || !strides || !dets || !count || !status || !wsp) GLS_FAIL(GLSDET_E_ARG, "gfl_detect: null argument");
  if (n_levels < 1 || n_levels > GLS_MAX_LEVELS || num_classes < 1 || reg_max < 1 || reg_max > 63 || nms_pre < 1 ||
      max_cand < 1 || max_det < 1)
    GLS_FAIL(GLSDET_E_ARG, "gfl_detect: bad sizes");
  const int max_cand2 = n_levels * nms_pre;
  if (max_cand > GLS_NMS_MAXW * 64 || max_cand2 > GLS_NMS_MAXW * 64)
    GLS_FAIL(GLSDET_E_ARG, "gfl_detect: max_cand / n_levels*nms_pre above %d", GLS_NMS_MAXW * 64);
  if ((uintptr_t)wsp & 255) GLS_FAIL(GLSDET_E_ALIGN, "gfl_detect: workspace must be 256-byte aligned");
  GflArgs a = {};
  int maxhw = 0;
  for (int l = 0; l < n_levels; ++l) {
    int rc;
    if ((rc = check_view(cls[l], "gfl_detect.cls", false))) return rc;
    if ((rc = check_view(reg[l], "gfl_detect.reg", false))) return rc;
    if (cls[l].dtype != GLSDET_F32 || reg[l].dtype != GLSDET_F32) GLS_FAIL(GLSDET_E_ARG, "gfl_detect: levels must be fp32");
    if (cls[l].c < num_classes || reg[l].c < 4 * (reg_max + 1) || cls[l].n != cls[0].n || reg[l].n != cls[0].n ||
        reg[l].h != cls[l].h || reg[l].w != cls[l].w || strides[l] < 1)
      GLS_FAIL(GLSDET_E_ARG, "gfl_detect: level %d extent mismatch", l);
    a.cls[l] = (const float*)cls[l].base; a.reg[l] = (const float*)reg[l].base;
    a.csn[l] = cls[l].sn; a.csh[l] = cls[l].sh; a.csw[l] = cls[l].sw;
    a.rsn[l] = reg[l].sn; a.rsh[l] = reg[l].sh; a.rsw[l] = reg[l].sw;
    a.H[l] = cls[l].h; a.W[l] = cls[l].w;
    a.stride[l] = (float)strides[l];
    if (cls[l].h * cls[l].w > maxhw) maxhw = cls[l].h * cls[l].w;
  }
  const int n = cls[0].n;
  a.n_levels = n_levels; a.nc = num_classes; a.reg_max = reg_max; a.n = n;
  a.in_h = (float)in_h; a.in_w = (float)in_w; a.thr = score_thr; a.img_hw = img_hw;
  NmsWs w1, w2;
  const long need1 = nms_layout(n * n_levels, max_cand, &w1, (char*)wsp, false);
  const long off2 = align_up(need1, 256);
  const long need = off2 + nms_layout(n, max_cand2, &w2, (char*)wsp + off2, true);
  if (ws_bytes < need) GLS_FAIL(GLSDET_E_CAPACITY, "gfl_detect: workspace %ld < %ld bytes", (long)ws_bytes, need);
  const int nw2 = (max_cand2 + 63) / 64;
  double bytes = 0;
  for (int l = 0; l < n_levels; ++l) bytes += (double)n * a.H[l] * a.W[l] * (num_classes + 4.0 * (reg_max + 1)) * 4.0;
  OpRecord op;
  op.kind = 6;
  op.flops = 0;
  op.bytes = bytes;
  op.name = "gfl_detect(filter+topk+merge+nms)";
  op.launch = [=](hipStream_t st) -> int {
    hipLaunchKernelGGL(reset_counters_kernel, dim3(1), dim3(256), 0, st, w1.cnt, n * n_levels, status);
    hipLaunchKernelGGL(gfl_filter_kernel, dim3((maxhw + 255) / 256, n, n_levels), dim3(256), 0, st, a, max_cand, w1, status);
    hipLaunchKernelGGL(nms_rank_kernel, dim3((max_cand + 63) / 64, n * n_levels), dim3(256), 0, st, max_cand, w1);
    hipLaunchKernelGGL(gfl_merge_kernel, dim3((max_cand2 + 255) / 256, n), dim3(256), 0, st, n_levels, max_cand, nms_pre,
                       max_cand2, w1, w2, scale_factors);
    hipLaunchKernelGGL(nms_rank_kernel, dim3((max_cand2 + 63) / 64, n), dim3(256), 0, st, max_cand2, w2);
    hipLaunchKernelGGL(nms_mask_kernel, dim3(2048, n), dim3(64), 0, st, max_cand2, nw2, iou_thr, w2);
    hipLaunchKernelGGL(nms_scan_kernel, dim3(n), dim3(1024), 0, st, max_cand2, nw2, max_det, w2, dets, count);
    GLS_HIP(hipGetLastError());
    return 0;
  };
  return submit(std::move(op), stream);
}

// ---- detections -> fixed-capacity exchange record (the payload of the one all_gather) -------------
// out[img][0..cap) = the first min(count, cap) rows (score order), zero rows behind them; out[img][cap] = (count kept
// here, count before the cap, 0...).  One thread per float of the record.
__global__ __launch_bounds__(256) void pack_dets_kernel(const float* __restrict__ dets, const int* __restrict__ count,
                                                        int n, int max_det, int cap, float* __restrict__ out) {
  const long per = (long)(cap + 1) * 7;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n * per; i += (long)gridDim.x * 256) {
    const int img = (int)(i / per);
    const int r = (int)((i - img * per) / 7), f = (int)(i - img * per - r * 7);
    int c = count[img];
    c = c < max_det ? c : max_det;
    const int kept = c < cap ? c : cap;
    float v = 0.f;
    if (r < cap) {
      if (r < kept) v = dets[((long)img * max_det + r) * 7 + f];
    } else if (f == 0) {
      v = (float)kept;
    } else if (f == 1) {
      v = (float)count[n + img];
    }
    out[i] = v;
  }
}

extern "C" int glsdet_pack_detections(const float* dets, const int32_t* count, int32_t n, int32_t max_det, int32_t cap,
                                      float* out, void* stream) {
  if (!dets || !count || !out) GLS_FAIL(GLSDET_E_ARG, "pack_detections: null argument");
  if (n < 1 || max_det < 1 || cap < 1) GLS_FAIL(GLSDET_E_ARG, "pack_detections: bad sizes");
  OpRecord op;
  op.kind = 6;
  op.flops = 0;
  op.bytes = 2.0 * n * (cap + 1) * 28.0;
  op.name = "pack_detections";
  op.launch = [=](hipStream_t st) -> int {
    long g = ((long)n * (cap + 1) * 7 + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(pack_dets_kernel, dim3((unsigned)g), dim3(256), 0, st, dets, count, n, max_det, cap, out);
    GLS_HIP(hipGetLastError());
    return 0;
  };
  return submit(std::move(op), stream);
}

extern "C" int64_t glsdet_ufp_merge_workspace_bytes(int32_t max_cand) {
  if (max_cand < 1) return 0;
  return nms_layout(1, max_cand, nullptr, nullptr, true);
}

extern "C" int glsdet_ufp_backmap_merge(const float* dets, const int32_t* count, int32_t max_det, const float* chips,
                                        int32_t n_chips, float iof_thr, float nms_thr, int32_t max_cand, int32_t max_out,
                                        float* out, int32_t* out_count, int32_t* status, void* wsp, int64_t ws_bytes,
                                        void* stream) {
  if (!dets || !count || !out || !out_count || !status || !wsp || (n_chips > 0 && !chips))
    GLS_FAIL(GLSDET_E_ARG, "ufp_backmap_merge: null argument");
  if (max_det < 1 || n_chips < 0 || max_cand < 1 || max_out < 1 || max_cand > GLS_NMS_MAXW * 64)
    GLS_FAIL(GLSDET_E_ARG, "ufp_backmap_merge: bad sizes");
  if ((uintptr_t)wsp & 255) GLS_FAIL(GLSDET_E_ALIGN, "ufp_backmap_merge: workspace must be 256-byte aligned");
  NmsWs ws;
  const long need = nms_layout(1, max_cand, &ws, (char*)wsp, true);
  if (ws_bytes < need) GLS_FAIL(GLSDET_E_CAPACITY, "ufp_backmap_merge: workspace %ld < %ld bytes", (long)ws_bytes, need);
  const int nw = (max_cand + 63) / 64;
  OpRecord op;
  op.kind = 6;
  op.flops = 0;
  op.bytes = 28.0 * max_det * (n_chips > 0 ? n_chips : 1);
  op.name = "ufp_backmap+merge_nms";
  op.launch = [=](hipStream_t st) -> int {
    hipLaunchKernelGGL(reset_counters_kernel, dim3(1), dim3(256), 0, st, ws.cnt, 1, status);
    if (n_chips > 0) {
      long g = ((long)n_chips * max_det + 255) / 256;
      if (g > 4096) g = 4096;
      hipLaunchKernelGGL(ufp_backmap_kernel, dim3((unsigned)g), dim3(256), 0, st, dets, count, max_det, chips, n_chips, iof_thr,
                         max_cand, ws, status);
    }
    hipLaunchKernelGGL(nms_rank_kernel, dim3((max_cand + 63) / 64, 1), dim3(256), 0, st, max_cand, ws);
    hipLaunchKernelGGL(nms_mask_kernel, dim3(2048, 1), dim3(64), 0, st, max_cand, nw, nms_thr, ws, 1.0f);
    hipLaunchKernelGGL(nms_scan_kernel, dim3(1), dim3(1024), 0, st, max_cand, nw, max_out, ws, out, out_count);
    GLS_HIP(hipGetLastError());
    return 0;
  };
  return submit(std::move(op), stream);
}
