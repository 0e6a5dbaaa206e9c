// Per-anchor decode, class-max + threshold + compaction, and batched (per-class) NMS.
// HBM / latency bound integer-and-compare work: no MFMA; coalesced reads, LDS-staged
// tiles for the O(n^2) passes, 64-bit wave ballots as the suppression bitmask.
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
      o[0] = cx - w / 2;
      o[1] = cy - h / 2;
      o[2] = cx + w / 2;
      o[3] = cy + h / 2;
    }
    for (int c = 4; c < F; ++c) o[c] = sigmoidf_acc(p[c]);
  }
}

// ---------------------------------------------------------------------------- NMS
// workspace layout (per call): see nms_layout()
struct NmsWs {
  int* cnt;            // [n]   candidates found (unclamped)
  float4* cbox;        // [n][max_cand]  candidate boxes (xyxy), arrival order
  float4* cext;        // [n][max_cand]  obj, cls_conf, cls_id, score
  int* canchor;        // [n][max_cand]
  float4* sbox;        // [n][max_cand]  sorted by score desc
  float4* sext;        // [n][max_cand]
  unsigned long long* mask;  // [n][nw][max_cand]  (column-block major)
};

// utils_bbox.py:380-385 (cxcywh->xyxy), :398 class max, :403 threshold
__global__ __launch_bounds__(256) void nms_filter_kernel(const float* __restrict__ pred, int n, int A, int nc, int box_mode,
                                                         float thr, int max_cand, NmsWs ws, int* status) {
  const long total = (long)n * A;
  const int F = 5 + nc;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
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
    if (score >= thr) {
      const int slot = atomicAdd(&ws.cnt[b], 1);
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
      } else {
        atomicOr(status, 1);
      }
    }
  }
}

// rank sort: position of candidate i = #candidates that come before it in
// (score desc, anchor asc) order.  Exact, deterministic, O(n^2) compares from LDS.
__global__ __launch_bounds__(256) void nms_rank_kernel(int max_cand, NmsWs ws) {
  __shared__ float s_score[256];
  __shared__ int s_anchor[256];
  const int b = blockIdx.y;
  const int n = min(ws.cnt[b], max_cand);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.x * 256 >= n) return;
  const long base = (long)b * max_cand;
  float my = 0.f;
  int mya = 0;
  float4 bx, ex;
  if (i < n) {
    bx = ws.cbox[base + i];
    ex = ws.cext[base + i];
    my = ex.w;
    mya = ws.canchor[base + i];
  }
  int rank = 0;
  for (int j0 = 0; j0 < n; j0 += 256) {
    const int j = j0 + threadIdx.x;
    if (j < n) {
      s_score[threadIdx.x] = ws.cext[base + j].w;
      s_anchor[threadIdx.x] = ws.canchor[base + j];
    }
    __syncthreads();
    const int lim = min(256, n - j0);
    for (int t = 0; t < lim; ++t) {
      const float s = s_score[t];
      rank += (s > my) || (s == my && s_anchor[t] < mya);
    }
    __syncthreads();
  }
  if (i < n) {
    ws.sbox[base + rank] = bx;
    ws.sext[base + rank] = ex;
  }
}

// maskT[cb][i] bit t : candidate j = cb*64+t (j > i) has the class of i and IoU(i,j) > thr
// (torchvision nms: inter / (area_i + area_j - inter) > thr, areas without +1).
// Stored column-block major so that the scan reads 64 consecutive rows of one column block
// as one coalesced 512-byte wave load.  The grid is fixed; each 64-thread block walks the
// (row block, column block) pairs of the ACTUAL candidate count (known only on the device).
__global__ __launch_bounds__(64) void nms_mask_kernel(int max_cand, int nw, float thr, NmsWs ws) {
  const int b = blockIdx.y;
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
      const float aarea = (a.z - a.x) * (a.w - a.y);
      unsigned long long bits = 0;
      const int lim = min(64, n - cb * 64);
      for (int k = 0; k < lim; ++k) {
        const int jj = cb * 64 + k;
        if (jj <= i || ccls[k] != acls) continue;
        const float4 c = cbx[k];
        const float w = fmaxf(0.f, fminf(a.z, c.z) - fmaxf(a.x, c.x));
        const float h = fmaxf(0.f, fminf(a.w, c.w) - fmaxf(a.y, c.y));
        const float inter = w * h;
        const float iou = inter / (aarea + (c.z - c.x) * (c.w - c.y) - inter);
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

// Greedy scan, one 256-thread workgroup per image, removed-bits in LDS.  Per 64-candidate
// word: wave 0 resolves the word serially from its diagonal block held in registers
// (shuffles only, no memory latency on the dependent chain), then all four waves OR the
// rows of the newly kept candidates into the later words (coalesced, independent loads).
#define GLS_NMS_MAXW 512    // 512 * 64 = 32768 candidates max
__global__ __launch_bounds__(256) void nms_scan_kernel(int max_cand, int nw, int max_det, NmsWs ws, float* dets,
                                                       int* count) {
  __shared__ unsigned long long s_removed[GLS_NMS_MAXW];
  __shared__ unsigned long long s_kept;
  __shared__ int s_base;
  const int b = blockIdx.x;
  const int n = min(ws.cnt[b], max_cand);
  const long base = (long)b * max_cand;
  const unsigned long long* maskT = ws.mask + (long)b * nw * max_cand;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int words = (n + 63) >> 6;
  for (int w = tid; w < words; w += 256) s_removed[w] = 0ull;
  if (tid == 0) s_base = 0;
  for (int wi = 0; wi < words; ++wi) {
    __syncthreads();
    if (wave == 0) {
      const int i = wi * 64 + lane;
      const unsigned long long diag = (i < n) ? maskT[(long)wi * max_cand + i] : 0ull;
      unsigned long long cur = s_removed[wi];
      unsigned long long kept = 0;
      const int lim = min(64, n - wi * 64);
      for (int bit = 0; bit < lim; ++bit) {
        const unsigned long long row = __shfl(diag, bit, 64);   // wave-uniform source lane
        if (!((cur >> bit) & 1ull)) {
          kept |= 1ull << bit;
          cur |= row;
        }
      }
      const int pos0 = s_base;
      if (i < n && ((kept >> lane) & 1ull)) {
        const int pos = pos0 + __popcll(kept & ((1ull << lane) - 1ull));
        if (pos < max_det) {
          const float4 bx = ws.sbox[base + i];
          const float4 ex = ws.sext[base + i];
          float* d = dets + ((long)b * max_det + pos) * 7;
          d[0] = bx.x; d[1] = bx.y; d[2] = bx.z; d[3] = bx.w; d[4] = ex.x; d[5] = ex.y; d[6] = ex.z;
        }
      }
      if (lane == 0) {
        s_kept = kept;
        s_base = pos0 + __popcll(kept);
      }
    }
    __syncthreads();
    const unsigned long long kept = s_kept;
    const int i = wi * 64 + lane;
    const bool mine = i < n && ((kept >> lane) & 1ull);
    for (int w0 = wi + 1 + wave; w0 < words; w0 += 16) {
      unsigned long long v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int w = w0 + 4 * u;
        v[u] = (mine && w < words) ? maskT[(long)w * max_cand + i] : 0ull;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int w = w0 + 4 * u;
        const unsigned long long r = wave_or(v[u]);
        if (lane == 0 && w < words) s_removed[w] |= r;     // word w is touched by this wave only
      }
    }
  }
  __syncthreads();
  if (tid == 0) count[b] = min(s_base, max_det), count[gridDim.x + b] = s_base;
}

static inline long align_up(long v, long a) { return (v + a - 1) / a * a; }
static long nms_layout(int n, int max_cand, NmsWs* ws, char* base) {
  const int nw = (max_cand + 63) / 64;
  long off = 0;
  auto take = [&](long bytes) { long o = off; off = align_up(off + bytes, 256); return o; };
  const long o_cnt = take((long)n * 4);
  const long o_cbox = take((long)n * max_cand * 16);
  const long o_cext = take((long)n * max_cand * 16);
  const long o_can = take((long)n * max_cand * 4);
  const long o_sbox = take((long)n * max_cand * 16);
  const long o_sext = take((long)n * max_cand * 16);
  const long o_mask = take((long)n * max_cand * nw * 8);
  if (ws && base) {
    ws->cnt = (int*)(base + o_cnt);
    ws->cbox = (float4*)(base + o_cbox);
    ws->cext = (float4*)(base + o_cext);
    ws->canchor = (int*)(base + o_can);
    ws->sbox = (float4*)(base + o_sbox);
    ws->sext = (float4*)(base + o_sext);
    ws->mask = (unsigned long long*)(base + o_mask);
  }
  return off;
}

}  // namespace glsdet

using namespace glsdet;

extern "C" int glsdet_yolox_decode(const glsdet_view* levels, int32_t n_levels, int32_t num_classes, int32_t in_h,
                                   int32_t in_w, const int32_t* strides, int32_t mode, float* out, int64_t out_elems,
                                   void* stream) {
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
  op.name = "nms(filter+rank+mask+scan)";
  op.launch = [=](hipStream_t st) -> int {
    GLS_HIP(hipMemsetAsync(ws.cnt, 0, (size_t)n * 4, st));
    GLS_HIP(hipMemsetAsync(status, 0, 4, st));
    long g = ((long)n * A + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(nms_filter_kernel, dim3((unsigned)g), dim3(256), 0, st, pred, n, A, num_classes, box_mode,
                       conf_thres, max_cand, ws, status);
    hipLaunchKernelGGL(nms_rank_kernel, dim3((max_cand + 255) / 256, n), dim3(256), 0, st, max_cand, ws);
    hipLaunchKernelGGL(nms_mask_kernel, dim3(2048, n), dim3(64), 0, st, max_cand, nw, nms_thres, ws);
    hipLaunchKernelGGL(nms_scan_kernel, dim3(n), dim3(256), 0, st, max_cand, nw, max_det, ws, dets, count);
    GLS_HIP(hipGetLastError());
    return 0;
  };
  return submit(std::move(op), stream);
}
