// bbox COCOeval on the device: the per-(image, category) work of
//   COCOeval.computeIoU   drone/models/core/cocoeval.py:163-190  (bbIou of pycocotools' maskApi.c)
//   COCOeval.evaluateImg  drone/models/core/cocoeval.py:235-313  (greedy matching per IoU threshold
//                                                                 and area range)
// for every pair at once.  Everything is fp64 like the reference (python floats / C doubles) and
// the products are kept out of fused multiply-adds, so the IoU values -- and with them every
// `iou < thr` decision -- are the reference's bit for bit.
//
// evaluateImg's inner loop walks the ground truths in "ignored last" order and keeps the LAST best
// candidate (`ious < iou -> continue`, otherwise update), but stops at the first ignored one once a
// regular match exists.  That is: arg max (IoU, then position) over the regular region; only when
// that is empty, the same over the ignored region.  One wave handles one (pair, area range, IoU
// threshold) with its lanes across the ground truths; the detections stay sequential (each match
// removes a ground truth for the following ones).
#include "common.h"

namespace glsdet {

struct CocoArgs {
  const double* dt_box;    // [ND][4] x,y,w,h -- per pair contiguous, descending score (stable), <= maxDet
  const double* dt_area;   // [ND]
  const double* gt_box;    // [NG][4]
  const double* gt_area;   // [NG]
  const unsigned char* gt_flags;   // [NG] bit0 iscrowd, bit1 ignore, bit2 "annotation id is 0"
  const int* dt_off;       // [P+1]
  const int* gt_off;       // [P+1]
  const long long* iou_off;   // [P+1] offsets of the [D][G] blocks in ious
  const double* area_rng;  // [A][2]
  const double* iou_thr;   // [T]
  int P, A, T, ND, NG;
  double* ious;            // out [sum D*G], ORIGINAL ground-truth order (what computeIoU returns)
  int* gt_order;           // out [A][NG] position -> index inside the pair (ignored last, stable)
  unsigned char* gt_ignore;   // out [A][NG] by position
  int* n_regular;          // out [A][P]
  int* dt_match;           // out [A][T][ND] index inside the pair of the matched ground truth, -1 none
  unsigned char* dt_ignore;   // out [A][T][ND]
  int* gt_match;           // out [A][T][NG] by position: index of the matching detection, -1 none
};

__global__ __launch_bounds__(256) void coco_iou_kernel(CocoArgs a) {
  for (int p = blockIdx.x; p < a.P; p += gridDim.x) {
    const int d0 = a.dt_off[p], D = a.dt_off[p + 1] - d0, g0 = a.gt_off[p], G = a.gt_off[p + 1] - g0;
    double* o = a.ious + a.iou_off[p];
    const long n = (long)D * G;
    for (long e = threadIdx.x; e < n; e += blockDim.x) {
      const int d = (int)(e / G), g = (int)(e % G);
      const double* Db = a.dt_box + 4l * (d0 + d);
      const double* Gb = a.gt_box + 4l * (g0 + g);
      // bbIou (pycocotools common/maskApi.c): the same operations in the same order, unfused
      double v = 0.0;
      {
#pragma clang fp contract(off)
        const double da = Db[2] * Db[3], ga = Gb[2] * Gb[3];
        const double w = fmin(Db[2] + Db[0], Gb[2] + Gb[0]) - fmax(Db[0], Gb[0]);
        if (w > 0) {
          const double h = fmin(Db[3] + Db[1], Gb[3] + Gb[1]) - fmax(Db[1], Gb[1]);
          if (h > 0) {
            const double i = w * h;
            const double u = (a.gt_flags[g0 + g] & 1) ? da : da + ga - i;
            v = i / u;
          }
        }
      }
      o[e] = v;
    }
  }
}

// one thread per (pair, area range): stable "ignored last" order of the pair's ground truths
__global__ __launch_bounds__(256) void coco_order_kernel(CocoArgs a) {
  const long n = (long)a.P * a.A;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    const int p = (int)(e % a.P), ar = (int)(e / a.P);
    const int g0 = a.gt_off[p], G = a.gt_off[p + 1] - g0;
    const double lo = a.area_rng[2 * ar], hi = a.area_rng[2 * ar + 1];
    int* ord = a.gt_order + (long)ar * a.NG + g0;
    unsigned char* ig = a.gt_ignore + (long)ar * a.NG + g0;
    int k = 0;
    for (int pass = 0; pass < 2; ++pass) {
      for (int g = 0; g < G; ++g) {
        const double ga = a.gt_area[g0 + g];
        const int ign = ((a.gt_flags[g0 + g] & 2) || ga < lo || ga > hi) ? 1 : 0;
        if (ign == pass) {
          ord[k] = g;
          ig[k] = (unsigned char)ign;
          ++k;
        }
      }
      if (pass == 0) a.n_regular[(long)ar * a.P + p] = k;
    }
  }
}

// arg max of (value, position) over the wave; lanes without a candidate carry pos = -1
__device__ __forceinline__ void wave_argmax(double& v, int& pos) {
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) {
    const double ov = __shfl_xor(v, s, 64);
    const int op = __shfl_xor(pos, s, 64);
    if (op >= 0 && (pos < 0 || ov > v || (ov == v && op > pos))) {
      v = ov;
      pos = op;
    }
  }
}

// one wave per (pair, area range, threshold)
__global__ __launch_bounds__(64) void coco_match_kernel(CocoArgs a) {
  const int lane = threadIdx.x;
  const long n = (long)a.P * a.A * a.T;
  for (long e = blockIdx.x; e < n; e += gridDim.x) {
    const int p = (int)(e % a.P);
    const int t = (int)((e / a.P) % a.T);
    const int ar = (int)(e / ((long)a.P * a.T));
    const int d0 = a.dt_off[p], D = a.dt_off[p + 1] - d0, g0 = a.gt_off[p], G = a.gt_off[p + 1] - g0;
    const long at = (long)ar * a.T + t;
    int* dtm = a.dt_match + at * a.ND + d0;
    unsigned char* dti = a.dt_ignore + at * a.ND + d0;
    int* gtm = a.gt_match + at * a.NG + g0;
    const int* ord = a.gt_order + (long)ar * a.NG + g0;
    const unsigned char* gig = a.gt_ignore + (long)ar * a.NG + g0;
    const unsigned char* gfl = a.gt_flags + g0;
    const int nreg = a.n_regular[(long)ar * a.P + p];
    const double lo = a.area_rng[2 * ar], hi = a.area_rng[2 * ar + 1];
    const double thr = fmin(a.iou_thr[t], 1 - 1e-10);
    const double* iou = a.ious + a.iou_off[p];
    for (int g = lane; g < G; g += 64) gtm[g] = -1;
    __builtin_amdgcn_wave_barrier();
    for (int d = 0; d < D; ++d) {
      const double* row = iou + (long)d * G;
      int m = -1;
      for (int region = 0; region < 2 && m < 0; ++region) {
        const int b = region ? nreg : 0, eend = region ? G : nreg;
        double best = 0.0;
        int pos = -1;
        for (int g = b + lane; g < eend; g += 64) {
          const int gi = ord[g];
          if (gtm[g] >= 0 && !(gfl[gi] & 1)) continue;       // taken, and not a crowd
          const double v = row[gi];
          if (v < thr) continue;
          if (pos < 0 || v >= best) {                          // later position wins a tie
            best = v;
            pos = g;
          }
        }
        wave_argmax(best, pos);
        m = pos;
      }
      if (lane == 0) {
        // an unmatched detection (or one whose match carries annotation id 0: the reference tests
        // `dtm == 0` on the stored id) outside the area range is ignored
        const double da = a.dt_area[d0 + d];
        const bool outside = da < lo || da > hi;
        if (m >= 0) {
          const int gi = ord[m];
          dtm[d] = gi;
          gtm[m] = d;
          dti[d] = (unsigned char)(gig[m] | (((gfl[gi] & 4) && outside) ? 1 : 0));
        } else {
          dtm[d] = -1;
          dti[d] = outside ? 1 : 0;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
}

}  // namespace glsdet

using namespace glsdet;

extern "C" int glsdet_coco_match(const double* dt_box, const double* dt_area, const int32_t* dt_off, const double* gt_box,
                                 const double* gt_area, const unsigned char* gt_flags, const int32_t* gt_off,
                                 const int64_t* iou_off, int32_t n_pairs, int32_t n_dt, int32_t n_gt,
                                 const double* area_rng, int32_t n_area, const double* iou_thr, int32_t n_thr,
                                 double* ious, int32_t* gt_order, unsigned char* gt_ignore, int32_t* n_regular,
                                 int32_t* dt_match, unsigned char* dt_ignore, int32_t* gt_match, void* stream) {
  if (n_pairs < 0 || n_dt < 0 || n_gt < 0 || n_area < 1 || n_thr < 1) GLS_FAIL(GLSDET_E_ARG, "coco_match: bad sizes");
  if (!dt_off || !gt_off || !iou_off || !area_rng || !iou_thr || !n_regular) GLS_FAIL(GLSDET_E_ARG, "coco_match: null argument");
  if (n_dt > 0 && (!dt_box || !dt_area || !dt_match || !dt_ignore)) GLS_FAIL(GLSDET_E_ARG, "coco_match: null detection array");
  if (n_gt > 0 && (!gt_box || !gt_area || !gt_flags || !gt_order || !gt_ignore || !gt_match))
    GLS_FAIL(GLSDET_E_ARG, "coco_match: null ground-truth array");
  if (n_dt > 0 && n_gt > 0 && !ious) GLS_FAIL(GLSDET_E_ARG, "coco_match: null IoU buffer");
  if (n_pairs == 0) return 0;
  CocoArgs a;
  a.dt_box = dt_box; a.dt_area = dt_area; a.gt_box = gt_box; a.gt_area = gt_area; a.gt_flags = gt_flags;
  a.dt_off = dt_off; a.gt_off = gt_off; a.iou_off = (const long long*)iou_off; a.area_rng = area_rng; a.iou_thr = iou_thr;
  a.P = n_pairs; a.A = n_area; a.T = n_thr; a.ND = n_dt; a.NG = n_gt;
  a.ious = ious; a.gt_order = gt_order; a.gt_ignore = gt_ignore; a.n_regular = n_regular;
  a.dt_match = dt_match; a.dt_ignore = dt_ignore; a.gt_match = gt_match;
  OpRecord op;
  op.kind = 6;
  op.flops = 0;
  op.bytes = 40.0 * n_dt + 41.0 * n_gt;
  op.name = "coco_match(iou+order+match)";
  op.launch = [=](hipStream_t st) -> int {
    const unsigned gp = (unsigned)(a.P < 65535 ? a.P : 65535);
    hipLaunchKernelGGL(coco_iou_kernel, dim3(gp), dim3(256), 0, st, a);
    const long no = ((long)a.P * a.A + 255) / 256;
    hipLaunchKernelGGL(coco_order_kernel, dim3((unsigned)(no < 65535 ? no : 65535)), dim3(256), 0, st, a);
    const long nm = (long)a.P * a.A * a.T;
    hipLaunchKernelGGL(coco_match_kernel, dim3((unsigned)(nm < (1l << 20) ? nm : (1l << 20))), dim3(64), 0, st, a);
    GLS_HIP(hipGetLastError());
    return 0;
  };
  return submit(std::move(op), stream);
}
