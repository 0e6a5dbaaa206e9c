/* glsdet_hip.h -- C ABI of libglsdet_hip.so (MI355X / gfx950 only).
 *
 * Drop-in boundary for ONE path of WUTCM-Lab/GLSDet: the detection forward pass
 * (backbone -> PAFPN / GL-fusion neck -> decoupled head -> decode -> NMS).  The reference
 * is pure Python; the native kernels it reaches live in torch ATen/cuDNN and
 * torchvision.  Each entry point below replaces one such call site (reference file:line
 * given per function, paths relative to the reference root, drone/ = yolox-drone/).
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is a DEVICE pointer unless it says host.
 *  - activations are NHWC "views": base pointer + element strides for n/h/w, channels
 *    contiguous.  A view can therefore be a quadrant, a channel slice of a concat buffer
 *    or a row band of a larger tensor without any copy.
 *  - dtype: GLSDET_F16 (fp16 storage, fp32 accumulate) or GLSDET_F32 (exact-f32 MFMA).
 *  - `stream` is a hipStream_t passed as void*.  All entry points are asynchronous on
 *    that stream, never synchronise, never allocate, and may be stream-captured.
 *  - return value: 0 = ok, negative = GLSDET_E_* (nothing was launched); the text of
 *    the last error of the calling thread is available from glsdet_last_error().
 *  - every view carries the bounds of the allocation it lives in; each launch checks on
 *    the host that the extreme addresses it can touch lie inside (a kernel is never
 *    launched on operands that do not fit).
 */
#ifndef GLSDET_HIP_H
#define GLSDET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GLSDET_ABI_VERSION 13

enum { GLSDET_F16 = 0, GLSDET_F32 = 1 };
enum { GLSDET_ACT_NONE = 0, GLSDET_ACT_SILU = 1, GLSDET_ACT_RELU = 2, GLSDET_ACT_LRELU = 3,
       GLSDET_ACT_GELU = 4 /* exact erf form, nn.GELU() */, GLSDET_ACT_SIGMOID = 5,
       /* OR-ed into `act`: the residual is added BEFORE the activation, act(conv*scale+bias+res)
        * (ResNet Bottleneck: `out += identity; out = relu(out)`, ufp/mmdet/models/backbones/
        * resnet.py:292-301).  Without the flag the order is act(conv*scale+bias) + res
        * (YOLOX Bottleneck, drone/models/base/darknet.py:61-63).                            */
       GLSDET_ACT_RES_FIRST = 0x100 };
enum {
  GLSDET_OK = 0,
  GLSDET_E_ARG = -1,      /* inconsistent shapes / unsupported parameter            */
  GLSDET_E_BOUNDS = -2,   /* a view reaches outside its allocation                  */
  GLSDET_E_ALIGN = -3,    /* pointer / stride not 16-byte compatible                */
  GLSDET_E_HIP = -4,      /* HIP runtime error (text in glsdet_last_error)          */
  GLSDET_E_CAPACITY = -5  /* workspace too small                                    */
};

/* NHWC view.  Strides in ELEMENTS of `dtype`; channel stride is 1. */
typedef struct glsdet_view {
  void*   base;            /* address of element (n=0,h=0,w=0,c=0)                    */
  int64_t sn, sh, sw;      /* element strides                                         */
  int32_t n, h, w, c;      /* logical extent                                          */
  int32_t dtype;           /* GLSDET_F16 / GLSDET_F32                                 */
  int32_t _pad;
  void*   alloc_lo;        /* [alloc_lo, alloc_hi) = allocation the view lives in     */
  void*   alloc_hi;
} glsdet_view;

/* ---------------------------------------------------------------------------------
 * conv2d + folded BatchNorm + activation (+ residual add)          reference call sites:
 *   drone/models/base/baseConv.py:15-16   act(bn(conv(x)))            (BaseConv)
 *   drone/models/base/darknet.py:61-62    y = conv2(conv1(x)) + x     (Bottleneck add)
 *   drone/models/base/yolox.py:31-44      predictors, Conv2d + bias
 *   drone/models/block/non_local/Identity_Conv.py:27-84   dense kxk conv + bias
 *   (mmdet twin: mmcv ConvModule at ufp/mmdet/models/backbones/csp_darknet.py:39-47)
 * y[n,ho,wo,co] = act( scale[co] * sum_{r,s,ci} x[n,ho*stride-pad+r,wo*stride-pad+s,ci]
 *                                              * w[co,r,s,ci] + bias[co] ) (+ res[n,ho,wo,co])
 * Implicit GEMM on MFMA, NHWC, LDS-staged weight and im2col tiles, fused epilogue.
 *  w      : packed [cout_pad][kpad] elements of x.dtype, k = (r*S + s)*x.c + ci,
 *           kpad = round_up(R*S*x.c, 64 bytes worth of elements), cout_pad = round_up(y.c, 32),
 *           zero filled (see glsdet_conv_weight_elems).
 *  scale/bias : fp32 [cout_pad]  (BN folded: scale = gamma/sqrt(var+eps), bias = beta-mean*scale;
 *           plain conv: scale = 1, bias = conv bias)
 *  res    : optional (base == NULL for none), same dtype/extent as y.
 *  x.c, y.c must be multiples of 8; y.dtype may be F32 while x.dtype is F16 (predictors).
 */
typedef struct glsdet_conv_desc {
  glsdet_view x, y, res;
  const void*  w;
  const float* scale;
  const float* bias;
  int32_t R, S, stride, pad, act;
  int32_t tile_hint;       /* 0 auto | 1 generic | 2 halo (s1 kxk) | 4 halo, wave-private weights | 5 halo, 64-row cout tiles |
                            * 8 / 9 halo with the weight tiles in an LDS-DMA ring (64- / 128-row cout tiles) | 10 / 11 the same with
                            *   64-byte channel chunks (64 / 128 rows) | 12 / 13 the 8-wave form: 128 cout rows x 8 x 32 pixels per
                            *   512-thread workgroup (128- / 64-byte chunks; no residual) | 3 weight-stationary 1x1 |
                            * 0x100 | h, 0x200 | h (h = 8..11, 13): ring kernel h on 10 x 12 resp. 6 x 21 pixel tiles (h = 13:
                            *   10 x 24 resp. 6 x 42) instead of 8 x 16 (8 x 32) -- fewer wasted lanes on 50 x 84 / 25 x 42 maps |
                            * 16..31 the persistent LDS-DMA 1x1 kernel, variant hint - 16 (csrc/conv_gemm.hip: tile, ring
                            *   depth, K panel, weight-resident, single shot) |
                            * co_tile<<16|px_tile (|0x8000: 64-byte K steps); 128<<16|0 = the 8-wave 128 x 256 tile */
} glsdet_conv_desc;

int     glsdet_conv2d(const glsdet_conv_desc* d, void* stream);
/* Up to 8 independent convolutions of ONE shape class (same R, S, stride, pad, Cin, Cout and
 * dtypes; extents, strides, weights, residuals may differ) as one launch of the generic kernel:
 * the four quadrant convs of Patch_Conv (drone/models/block/non_local/Identity_Conv.py:298-301),
 * the cls / reg tower convs of one head level (base/yolox.py:62-75).  Each alone is too small
 * to fill the chip.  tile_hint of d[0] applies (0 = auto; 8..11 = the grouped LDS-DMA ring kernel for 3x3 problems, meaning
 * as for glsdet_conv2d, refused with GLSDET_E_ARG where it does not apply).  Results are those of n glsdet_conv2d.
 * `w` may point at an ACTIVATION matrix (rows of x.c elements at a pitch of glsdet_conv_kpad elements, zero padded):
 * the batched products of the non-local block at ResNet widths are such 1x1 "convs" with per-image weights.
 * n = 9 .. 32 (round 3): the BATCHED form -- all descriptors must describe ONE geometry (extents, strides, epilogue, residual
 * or not: they may differ in the operand addresses only; the 8 images x 4 quadrants of a plug-in level), the generic
 * tiles only (tile_hint 0 or co << 16 | px); anything else is refused with GLSDET_E_ARG. */
int     glsdet_conv2d_multi(const glsdet_conv_desc* d, int32_t n, void* stream);
/* as glsdet_conv2d_tune, for the group: fastest tile_hint of the one-launch form and its time */
int     glsdet_conv2d_multi_tune(const glsdet_conv_desc* d, int32_t n, void* stream, int32_t* best_hint, float* best_us);
/* Build-time autotune: times every kernel / tile variant that applies to exactly this
 * problem on the device (synchronises; never recorded into a plan) and returns the fastest
 * tile_hint.  The output view is written with the conv result. */
int     glsdet_conv2d_tune(const glsdet_conv_desc* d, void* stream, int32_t* best_hint, float* best_us);
/* A conv (exactly as glsdet_conv2d, residual included) followed IN THE SAME LAUNCH by a 1x1 conv + folded BN + act on a
 * channel range of its result -- the two back-to-back BaseConvs of a CSPLayer / Bottleneck chain
 * (drone/models/base/darknet.py:58-63: `y = conv2(conv1(x))`; :96-107: conv1 -> m[0].conv1):
 *     y  = glsdet_conv2d(d)
 *     y2 = act2( scale2 * sum_{c < cin2} y[.., c0 + c] * w2[co2][c] + bias2 )          (w2 packed as for a 1x1 conv)
 * The workgroup that produced a tile multiplies it from LDS while it is still there: the 1x1 costs no launch and no
 * re-read of y.  It consumes the STORED (rounded) values in the k order of the stand-alone kernels, so y2 equals what
 * glsdet_conv2d on y would give, bit for bit.  Limits: x, y, y2 one dtype; y2.c <= 128; the channels [c0, c0+cin2) must
 * lie inside one cout tile of the kernel that runs d (else GLSDET_E_ARG: callers fall back to two launches).          */
typedef struct glsdet_conv_chain {
  glsdet_view y2;          /* y's n, h, w                                                        */
  const void*  w2;         /* [cout_pad(y2.c)][kpad(1,1,cin2)] elements of y.dtype               */
  const float* scale2;
  const float* bias2;
  int32_t act2, c0, cin2;
  int32_t flags;           /* 0 or GLSDET_CHAIN_SKIP_Y (ABI 13; the field was padding before)      */
} glsdet_conv_chain;
/* y is read by nothing but the chained conv: it is not stored (its view is still validated).  Needs c0 == 0, cin2 == y.c,
 * no residual.  The stride-2 BaseConv in front of a CSPLayer (drone/models/base/darknet.py:174-195: `dark3 = Sequential(
 * Conv(.., 3, 2), CSPLayer(..))`) + the layer's conv1 | conv2 then move the 3x3's output through LDS only.          */
#define GLSDET_CHAIN_SKIP_Y 1
int     glsdet_conv2d_chain(const glsdet_conv_desc* d, const glsdet_conv_chain* c, void* stream);
int     glsdet_conv2d_chain_tune(const glsdet_conv_desc* d, const glsdet_conv_chain* c, void* stream, int32_t* best_hint,
                                 float* best_us);
/* Bottleneck front in ONE launch (drone/models/base/darknet.py:61-64 `y = self.conv2(self.conv1(x)); if self.use_add:
 * y = y + x`): c1 = the 1x1 (c1->y describes the hidden tensor; its base is not touched), c2 = the 3x3 stride 1 pad 1 on
 * that hidden tensor (c2->x must equal c1->y in extent and dtype), with c2's residual / activation as for glsdet_conv2d.
 * The workgroup of an 8 x 16 output tile recomputes the 1x1 on the tile's halo into LDS; the hidden values are rounded
 * to the storage dtype there, so y equals the two-launch form bit for bit.  Limits: hidden channels 32, 64 or 128 ==
 * c2->y.c; one dtype; y must not overlap x other than as a disjoint channel slice of the same pixel-interleaved buffer
 * (the fused form cannot run in place: GLSDET_E_ARG).  hint 0 / 1: 128- / 64-byte channel chunks.                      */
int     glsdet_bottleneck(const glsdet_conv_desc* c1, const glsdet_conv_desc* c2, int32_t hint, void* stream);
int     glsdet_bottleneck_tune(const glsdet_conv_desc* c1, const glsdet_conv_desc* c2, void* stream, int32_t* best_hint,
                               float* best_us);
/* Depthwise k x k conv + folded BN + act (`dconv` of DWConv, drone/models/base/baseConv.py:22-30;
 * mmcv DepthwiseSeparableConvModule).  Same descriptor; x.c == y.c; w = [R*S][x.c] elements of
 * x.dtype (tap-major), scale/bias fp32 [x.c]; res must be empty. */
int     glsdet_dwconv2d(const glsdet_conv_desc* d, void* stream);
/* the same with a dilation (LSKblock.conv_spatial: 7x7, dilation 3, padding 9, groups = dim; drone/models/lsk/LSK.py:31);
 * y extent = floor((x + 2 pad - dilation (k - 1) - 1) / stride) + 1.                                                  */
int     glsdet_dwconv2d_dilated(const glsdet_conv_desc* d, int32_t dilation, void* stream);
/* number of ELEMENTS of the packed weight buffer for (cout, R, S, cin, dtype)           */
int64_t glsdet_conv_weight_elems(int32_t cout, int32_t R, int32_t S, int32_t cin, int32_t dtype);
int32_t glsdet_conv_kpad(int32_t R, int32_t S, int32_t cin, int32_t dtype);
int32_t glsdet_conv_cout_pad(int32_t cout);

/* ---------------------------------------------------------------------------------
 * Focus space-to-depth + layout change      drone/models/base/darknet.py:15-21
 * img: NCHW fp32 [n,3,H,W] (contiguous)  ->  y: NHWC view [n,H/2,W/2,16]:
 * channels 0..11 = (TL, BL, TR, BR) x (c0,c1,c2), 12..15 = 0.
 */
int glsdet_focus_pack(const float* img, int32_t n, int32_t cin, int32_t H, int32_t W,
                      const glsdet_view* y, void* stream);

/* Focus AND its 3x3 BaseConv in one launch (drone/models/base/darknet.py:10-21: `self.conv(cat(TL, BL, TR, BR))`): the
 * fp32 NCHW image is read directly, the packed tensor never exists.  w / scale / bias: as glsdet_conv2d for a 3x3 conv
 * over 16 input channels (12 real), packed for y's dtype.  y: NHWC view [n, H/2, W/2, <= 64 channels].                */
int glsdet_focus_conv(const float* img, int32_t n, int32_t cin, int32_t H, int32_t W, const void* w, const float* scale,
                      const float* bias, int32_t act, const glsdet_view* y, void* stream);
/* Focus, its 3x3 BaseConv AND the first downsampling conv (CSPDarknet.stem -> dark2[0], drone/models/base/darknet.py:117-121,
 * 3x3 stride 2) in one launch: the stem's output tensor -- the largest of the network -- is neither written nor read; the
 * workgroup of an output tile computes the stem on the tile's halo into LDS.  w1 / scale1 / bias1: the stem (c1 = 32
 * output channels, packed as for glsdet_focus_conv); w2 / scale2 / bias2: a 3x3 conv over 32 channels; y: NHWC view
 * [n, (H/2 + 1) / 2, (W/2 + 1) / 2, <= 64 channels].  Equals glsdet_focus_conv + glsdet_conv2d(stride 2) bit for bit.     */
int glsdet_focus_conv_down(const float* img, int32_t n, int32_t cin, int32_t H, int32_t W, const void* w1, const float* scale1,
                           const float* bias1, int32_t act1, int32_t c1, const void* w2, const float* scale2,
                           const float* bias2, int32_t act2, const glsdet_view* y, void* stream);

/* max pool k x k, stride 1, pad k/2 (-inf padding)   drone/models/base/darknet.py:29,35 */
int glsdet_maxpool2d(const glsdet_view* x, const glsdet_view* y, int32_t k, void* stream);

/* The three pools of SPPBottleneck (darknet.py:29,35: kernel sizes 5, 9, 13, stride 1, pad k/2) in one launch: x is read once,
 * pool9 = pool5(pool5), pool13 = pool5(pool9) are formed in LDS.  y5 / y9 / y13: x's extent and dtype, one common set of
 * strides (the channel slices of the concat buffer the SPP conv2 reads).                                                */
int glsdet_spp_pools(const glsdet_view* x, const glsdet_view* y5, const glsdet_view* y9, const glsdet_view* y13, void* stream);

/* SpatialAttention front half (drone/models/new/Non_local_family.py:429-432): per pixel the max
 * and the mean over channels.  y: view [n,h,w,8] of x.dtype, channel 0 = max, 1 = mean, 2..7 = 0
 * (the 7x7 2->1 conv + sigmoid that follows is a glsdet_conv2d with GLSDET_ACT_SIGMOID).     */
int glsdet_channel_maxmean(const glsdet_view* x, const glsdet_view* y, void* stream);

/* nearest-neighbour resample by an integer factor (1 = strided copy, 2 = nn.Upsample(2))
 * drone/models/base/yolox.py:103,181,198 ; torch.cat is realised by views, not copies.   */
int glsdet_resample_copy(const glsdet_view* x, const glsdet_view* y, int32_t factor, void* stream);

/* `count` strided copies x[i] -> y[i] (same extents per pair, one dtype and channel count per call) in one launch per 32
 * pairs: the per-(image, quadrant) `.view(b, c, -1)` / `.permute().contiguous()` operand copies of the non-local block
 * (drone/models/new/Non_local_family.py:32-41) when it runs on a ResNet stage, where every product is a GEMM whose
 * second operand must be a dense matrix.                                                                    */
int glsdet_copy_many(const glsdet_view* x, const glsdet_view* y, int32_t count, void* stream);

/* The transposed form: x[i] is ONE image window [1,h,w,C], y[i] a dense matrix view [1,1,rows >= C, cols >= ceil8(h*w)]
 * (row pitch y[i].sw); y[i][c][p] = x[i][p / w][p % w][c].  This is `x.view(b, c, -1)` (Non_local_family.py:33-39) for
 * NHWC storage: the operand of the products that contract over the pixels (X^T X).  Columns beyond the pixels, rows
 * beyond C are left as they are.                                                                            */
int glsdet_transpose_many(const glsdet_view* x, const glsdet_view* y, int32_t count, void* stream);

/* ---------------------------------------------------------------------------------
 * Non-local block, dot-product form       drone/models/block/non_local/Identity_Conv.py:152-173
 *   out = x + Wout * ( (theta^T phi / N) g^T ) + bout      (NO softmax, divide by N)
 * `tpg` holds the three 1x1 projections [theta | phi | g] (each ci channels) of x, produced
 * by glsdet_conv2d with concatenated weights.  Evaluated in the re-associated order
 *   G = sum_j phi_j (x) g_j   [ci x ci],  out_i = x_i + (Wout G^T / N) theta_i + bout
 * which is the same bilinear form at O(N ci^2) instead of O(N^2 ci).
 *  wout : fp32 [cx][ci] row-major, bout: fp32 [cx];
 *  gram : fp32 workspace of n*(8*ci*ci + cx*ci) floats (partial Gram slices + folded matrix).
 */
int glsdet_nonlocal(const glsdet_view* x, const glsdet_view* tpg, int32_t ci,
                    const float* wout, const float* bout, float* gram,
                    const glsdet_view* out, void* stream);
/* 1..4 independent non-local blocks with the same n, channel counts and dtype (the four
 * quadrants of Patch_Conv_NonLocal, Identity_Conv.py:361-364) in one set of three launches.
 * x, tpg, out: arrays of n_sets views; wout, bout: arrays of n_sets device pointers;
 * gram: n_sets times the single-block workspace.                                           */
int glsdet_nonlocal_multi(const glsdet_view* x, const glsdet_view* tpg, int32_t n_sets, int32_t ci,
                          const float* const* wout, const float* const* bout, float* gram,
                          const glsdet_view* out, void* stream);

/* ---------------------------------------------------------------------------------
 * Patch_Conv_NonLocal_adapt_new: the data-dependent quadrant split (drone/models/new/Non_local_family.py:272-357)
 * The reference thresholds its SpatialAttention map (values under min + 0.75 (max - min) -> 0) and walks columns / rows in
 * Python until the running sum passes half of the total (`get_centroid`, :298-321: one device sync per step).  Here:
 *   glsdet_attn_split   att = [n,h,w,>=1] view whose channel 0 is the map (n <= 16, one split for the whole batch as in the
 *                       reference) -> split = DEVICE int32[4] {row split, column split above it, column split from it on, 0}
 *   glsdet_nonlocal_split  the four Non_local_Blocks (sets lt, lb, rt, rb) on the windows `split` describes: x / out full
 *                       maps, tpg[q] = [theta|phi|g] of the full map with quadrant q's weights, gram = 4 x the workspace of
 *                       glsdet_nonlocal
 *   glsdet_rowsplit     mode 0 / 1: y = a above / from the row split, zero elsewhere (the zero padding the top / bottom 3x3
 *                       convs see at the split); mode 2: y = row < split ? a : b (`torch.cat((t, b), dim=2)`);
 *                       mode 3 | q << 4: y = a inside quadrant q (0 lt, 1 lb, 2 rt, 3 rb), zero outside; mode 4 | q << 4: y = a
 *                       inside quadrant q, y untouched outside; | 1 << 8: the split indices halved (regions of the stride-2
 *                       map of Patch_Conv_NonLocal_adapt, :112-206, whose quadrant convs have stride 2)
 *   glsdet_scale_by_map y[..,c] = map[..,0] * x[..,c]                                  (:355-356)
 * Nothing returns to the host: the block is recordable / capturable like any other op sequence.                       */
int glsdet_attn_split(const glsdet_view* att, int32_t* split, void* stream);
int glsdet_nonlocal_split(const glsdet_view* x, const glsdet_view* tpg /*[4]*/, int32_t ci, const float* const* wout /*[4]*/,
                          const float* const* bout /*[4]*/, float* gram, const glsdet_view* out, const int32_t* split,
                          int32_t split_shift /* 0, or 1: windows on the stride-2 map */, void* stream);
int glsdet_rowsplit(const glsdet_view* a, const glsdet_view* b /*mode 2*/, const glsdet_view* y, const int32_t* split, int32_t mode,
                    void* stream);
int glsdet_scale_by_map(const glsdet_view* x, const glsdet_view* map, const glsdet_view* y, void* stream);

/* ---------------------------------------------------------------------------------
 * YOLOX decode        drone/models/core/utils_bbox.py:254-306  (mode 0, normalised cxcywh)
 *                     ufp/mmdet/models/dense_heads/yolox_head.py:298-308 (mode 1, xyxy px)
 * levels: fp32 views [n,H_l,W_l,>=5+nc] (channel order reg4, obj, cls..).
 * out: fp32 [n][A][5+nc] contiguous, A = sum H_l*W_l, level-major then row-major.
 * mode 0: sigmoid(obj,cls); cx=(x+gx)*s/in_w, cy=(y+gy)*s/in_h, w=exp()*s/in_w, h=exp()*s/in_h,
 *         s = in_h / H_l for both axes (reference quirk, utils_bbox.py:285).
 * mode 1: sigmoid(obj,cls); x1,y1,x2,y2 in input pixels, s_l = strides[l]; if scale_factors
 *         (device fp32 [n][4], may be NULL) is given the box is divided by it per image
 *         (mmdet `rescale=True`, yolox_head.py:283-285).
 */
int glsdet_yolox_decode(const glsdet_view* levels, int32_t n_levels, int32_t num_classes,
                        int32_t in_h, int32_t in_w, const int32_t* strides /*host, may be NULL*/,
                        int32_t mode, float* out, int64_t out_elems,
                        const float* scale_factors, void* stream);

/* ---------------------------------------------------------------------------------
 * class-max + score threshold + batched (per-class) NMS
 *   drone/models/core/utils_bbox.py:375-419  (torchvision.ops.boxes.batched_nms)
 *   ufp/mmdet/models/dense_heads/yolox_head.py:310-322 (mmcv.ops.batched_nms)
 * pred: fp32 [n][A][5+nc] as produced by glsdet_yolox_decode (either mode; mode 0 boxes
 *       are converted cxcywh->xyxy first, utils_bbox.py:380-385).
 * keep score = obj * max_c cls_c >= conf_thres; candidates visited by descending score
 * (ties: lower anchor index first); a candidate is dropped if an already kept one of the
 * SAME class has IoU > nms_thres (areas without +1).
 * dets : fp32 [n][max_det][7] = x1,y1,x2,y2,obj,cls_conf,cls_id in score order
 * count: int32 [2n]: [0,n) kept per image clamped to max_det, [n,2n) unclamped
 * ws   : workspace of glsdet_nms_workspace_bytes(n, A, max_cand) bytes
 * status: int32[1], bit0 set if some image had more than max_cand candidates (results for
 *         that image are then NOT the reference's: callers must treat it as an error).
 */
int64_t glsdet_nms_workspace_bytes(int32_t n, int32_t A, int32_t max_cand);
int glsdet_nms(const float* pred, int32_t n, int32_t A, int32_t num_classes, int32_t box_mode,
               float conf_thres, float nms_thres, int32_t max_cand, int32_t max_det,
               float* dets, int32_t* count, int32_t* status,
               void* ws, int64_t ws_bytes, void* stream);

/* The exchange record of the multi-GPU path: replaces the pickled result list that
 * collect_results_gpu pads and all_gathers (ufp/mmdet/apis/test.py:161-191).
 *   dets [n][max_det][7], count int32 [2n] as glsdet_nms / glsdet_gfl_detect leave them
 *   out  fp32 [n][cap+1][7]: rows 0..cap-1 = the first min(count, cap) detections in score order (zero rows
 *        behind them), row cap = (rows kept here, detections before any cap, 0, 0, 0, 0, 0).
 * One launch, recordable / capturable like every other op: the record is complete when the plan is.  */
int glsdet_pack_detections(const float* dets, const int32_t* count, int32_t n, int32_t max_det, int32_t cap,
                           float* out, void* stream);

/* ---------------------------------------------------------------------------------
 * ResNet-50 / FPN / GFL / MPHead helpers
 * --------------------------------------------------------------------------------- */
/* fp32 NCHW image -> NHWC view of the engine dtype, channels zero padded to y.c (>= cin,
 * multiple of 8): the input of the 7x7 s2 stem conv (ufp/mmdet/models/backbones/resnet.py:634). */
int glsdet_nchw_pack(const float* img, int32_t n, int32_t cin, int32_t H, int32_t W,
                     const glsdet_view* y, void* stream);

/* ResNet stem in one launch: y = act(scale * conv7x7 stride 2 pad 3 (img) + bias) straight from the fp32 NCHW image
 * (ufp/mmdet/models/backbones/resnet.py:634-636 `x = self.relu(self.norm1(self.conv1(x)))`), replacing glsdet_nchw_pack +
 * glsdet_conv2d over 3 channels padded to 8.  w: [64][7][8][4] elements of y's dtype = conv1.weight[co][c][r][s] at
 * [co][r][s][c], zero for s = 7 and c = 3 (glsdet_resnet_stem_weight_elems elements); scale / bias: folded BN, fp32 [64];
 * y: NHWC view [n, (H + 1) / 2, (W + 1) / 2, 64].  Same k order as the generic kernel minus its zero-padded channels:
 * results agree to fp32 summation-order noise.                                                                          */
int64_t glsdet_resnet_stem_weight_elems(void);
/* ... and with `x = self.maxpool(x)` (resnet.py:637: MaxPool2d(3, stride 2, padding 1)) in the epilogue: the stem's output
 * is never written; act is ReLU (a zero stands in for the pool's -inf padding).  y: [n, ((H+1)/2 + 1) / 2, ((W+1)/2 + 1) / 2, 64].
 * Equals glsdet_resnet_stem + glsdet_pool2d bit for bit.                                                                */
int glsdet_resnet_stem_pool(const float* img, int32_t n, int32_t cin, int32_t H, int32_t W, const void* w, const float* scale,
                            const float* bias, const glsdet_view* y, void* stream);
int glsdet_resnet_stem(const float* img, int32_t n, int32_t cin, int32_t H, int32_t W, const void* w, const float* scale,
                       const float* bias, int32_t act, const glsdet_view* y, void* stream);

/* nn.MaxPool2d(k, stride, pad) (resnet.py:598: k3 s2 p1), floor mode, -inf padding.
 * y extent must be floor((x + 2*pad - k)/stride) + 1.                                      */
int glsdet_pool2d(const glsdet_view* x, const glsdet_view* y, int32_t k, int32_t stride, int32_t pad,
                  void* stream);

/* FPN top-down step (ufp/mmdet/models/necks/fpn.py:165-175):
 *   fine += F.interpolate(coarse, size=fine.shape[2:], mode='nearest')
 * source index = min(floor(dst * (float)in/out), in-1) as torch computes it.  In place on fine. */
int glsdet_upsample_add(const glsdet_view* coarse, const glsdet_view* fine, void* stream);

/* nn.GroupNorm(groups, C) (+ optional ReLU) on an NHWC view, two kernels: fp64 partial sums
 * per (image, pixel slice, group) into `stats` (caller owned, glsdet_groupnorm_workspace_bytes
 * bytes, 8-byte aligned), then the fold + affine normalisation.  The conv towers of GFLHead /
 * MPHead: mmcv ConvModule(norm_cfg=GN32) = conv -> GN -> ReLU (gfl_head.py:128-152).
 * y may alias x.  gamma, beta: fp32 [C].  act: GLSDET_ACT_NONE or GLSDET_ACT_RELU.          */
int64_t glsdet_groupnorm_workspace_bytes(int32_t n, int32_t groups);
int glsdet_groupnorm(const glsdet_view* x, const glsdet_view* y, int32_t groups, const float* gamma,
                     const float* beta, float eps, int32_t act, void* stats, void* stream);
/* 1..16 tensors of the same n / C / groups in one launch pair (cls and reg tower outputs of all
 * pyramid levels); stats = n_sets times the single-tensor workspace. */
int glsdet_groupnorm_multi(const glsdet_view* x, const glsdet_view* y, int32_t n_sets, int32_t groups,
                           const float* const* gamma, const float* const* beta, float eps, int32_t act,
                           void* stats, void* stream);

/* The same, with the statistics of some sets already produced by the conv that wrote them: pre_stats[q] != NULL names
 * the partials glsdet_conv2d_gnstats wrote for x[q] (one slice per 8 x 16 pixel tile); those sets skip the statistics
 * pass (one read of the tensor less).  pre_stats == NULL or all NULL: glsdet_groupnorm_multi.                          */
int glsdet_groupnorm_multi_pre(const glsdet_view* x, const glsdet_view* y, int32_t n_sets, int32_t groups,
                               const float* const* gamma, const float* const* beta, float eps, int32_t act,
                               void* stats, void* const* pre_stats, void* stream);
/* glsdet_conv2d that ALSO writes the GroupNorm partial sums of the tensor it stores (gfl_head.py:128-152: every tower
 * conv is followed by GN(32) + ReLU): per (image, 8 x 16 pixel tile, group) the fp64 sum and sum of squares of the
 * stored values, summed in a fixed order.  3x3 stride 1, no residual, x.dtype == y.dtype, halo ring kernels only
 * (tile_hint 8..11, 0 = 8); anything else is GLSDET_E_ARG and the caller runs glsdet_conv2d + the two-pass GN.
 * stats: glsdet_conv2d_gnstats_bytes(n, ho, wo, groups) bytes, 8-byte aligned.                                         */
int64_t glsdet_conv2d_gnstats_bytes(int32_t n, int32_t ho, int32_t wo, int32_t groups);
int     glsdet_conv2d_gnstats(const glsdet_conv_desc* d, int32_t groups, void* stats, void* stream);
int     glsdet_conv2d_gnstats_tune(const glsdet_conv_desc* d, int32_t groups, void* stats, void* stream, int32_t* best_hint,
                                   float* best_us);

/* The two elementwise steps of LSKblock.forward (drone/models/lsk/LSK.py:46-48) on NHWC views of one dtype:
 *   mode 0: y[.., c] = a[.., c] * map[.., 0] + b[.., c] * map[.., 1]     (`attn1 * sig[:, 0] + attn2 * sig[:, 1]`)
 *   mode 1: y = a * b                                                     (`x * attn`; map may be NULL)
 * fp32 arithmetic, one rounding.  y may alias a or b.                                                                 */
int glsdet_gate(const glsdet_view* a, const glsdet_view* b, const glsdet_view* map, const glsdet_view* y, int32_t mode,
                void* stream);

/* MPHead.forward_proxy (ufp/mmdet/models/dense_heads/mp_head.py:105-121).
 *   feat : view [n,h,w,C] (the gfl_cls_conv output), engine dtype
 *   dots : fp32 view [n,h,w,>=P], feat . (proxy_k / max(|proxy_k|, 1e-12)) for the P proxies
 *          (a glsdet_conv2d with the normalised proxies as a 1x1 weight)
 *   counts: host int32 [nc], proxies per class (sum = P <= 64 per class and P <= 256 in all)
 *   out  : fp32 view [n,h,w,>=nc]:  gamma * sum_k softmax_k(gamma*s_k) * s_k  per class,
 *          s_k = dots_k / max(|feat|, 1e-12)                                              */
int glsdet_proxy_scores(const glsdet_view* feat, const glsdet_view* dots, const int32_t* counts,
                        int32_t num_classes, float gamma, const glsdet_view* out, void* stream);

/* GFL post-processing: gfl_head.py:380-471 (_get_bboxes_single) + base_dense_head.py:226-301
 * (_bbox_post_process) + core/utils/misc.py:119-165 (filter_scores_and_topk).
 *   cls : fp32 views [n,H_l,W_l,>=nc] raw class logits;  reg: fp32 views [n,H_l,W_l,>=4*(reg_max+1)]
 *   per level: score = sigmoid(cls) ; (position, class) pairs with score > score_thr ; the
 *   nms_pre best (ties: lower position*nc+class first) ; distances = Integral(reg) * stride ;
 *   box = (x*s - l, y*s - t, x*s + r, y*s + b) clamped to [0,img_w] x [0,img_h]
 *   (img_hw: device fp32 [n][2] = h,w per image, NULL = in_h,in_w) ; levels concatenated ;
 *   divided by scale_factors (device fp32 [n][4], may be NULL) ; per-class NMS (IoU > iou_thr
 *   suppresses) ; the first max_det in descending score order.
 *   dets : fp32 [n][max_det][7] = x1,y1,x2,y2,score,score,label ; count: int32 [2n] as glsdet_nms
 *   status bit0: a level held more than max_cand passing pairs (results invalid for that image)
 *   ws   : glsdet_gfl_workspace_bytes(n, n_levels, max_cand, nms_pre) bytes, 256-byte aligned   */
int64_t glsdet_gfl_workspace_bytes(int32_t n, int32_t n_levels, int32_t max_cand, int32_t nms_pre);
int glsdet_gfl_detect(const glsdet_view* cls, const glsdet_view* reg, int32_t n_levels,
                      const int32_t* strides /*host*/, int32_t num_classes, int32_t reg_max,
                      int32_t in_h, int32_t in_w, const float* img_hw, const float* scale_factors,
                      float score_thr, int32_t nms_pre, float iou_thr, int32_t max_cand, int32_t max_det,
                      float* dets, int32_t* count, int32_t* status,
                      void* ws, int64_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------
 * Image preprocessing, drone flavour:
 *   PIL `image.resize(size, Image.BICUBIC)` + `preprocess_input` + HWC->CHW
 *   (drone/models/core/utils.py:21-34,46-50; yolo.py:130-134), bit-identical to Pillow:
 *   two passes with uint8 intermediate and 22-bit fixed-point coefficients.
 * src : device uint8 [in_h][in_w][3] (RGB);  tmp: device uint8 [in_h][out_w][3] scratch
 * x/y bounds: device int32 [out][2] = (first source index, tap count); x/y kk: device int32
 *   [out][ksize] coefficients * 2^22 as Pillow's precompute_coeffs + normalize_coeffs_8bpc give
 *   them (glsdet_amd/preprocess.py computes the tables on the host)
 * dst : device fp32 image [3][dst_h][dst_w] (one image of an NCHW batch); the resized picture
 *   lands at (off_y, off_x) (letterbox paste), the rest of the canvas is the caller's.
 * mean3 / std3: host double[3]; value = f32(f64(f32(v/255f)) - mean) then f32(f64(.) / std), the
 *   mixed float32/float64 in-place arithmetic numpy performs in preprocess_input.          */
int glsdet_pil_resize_normalize(const unsigned char* src, int32_t in_h, int32_t in_w,
                                const int32_t* xbounds, const int32_t* xkk, int32_t xksize, int32_t out_w,
                                const int32_t* ybounds, const int32_t* ykk, int32_t yksize, int32_t out_h,
                                unsigned char* tmp, float* dst, int32_t dst_h, int32_t dst_w,
                                int32_t off_y, int32_t off_x, const double* mean3, const double* std3,
                                void* stream);

/* ---------------------------------------------------------------------------------
 * UFPMP-Det second stage (ufp/ufpmp_det_eval.py)
 * --------------------------------------------------------------------------------- */
/* display_merge_result (:182-193): chips = device fp32 [n_chips][7] = src_x, src_y, w, h, canvas_x,
 * canvas_y, magnification (floored like the reference); img: device uint8 [H][W][3] (BGR as cv2 reads
 * it); canvas: device fp32 [ch][cw][3], zero outside the chips.  Crops are magnified with cv2.resize's
 * uint8 INTER_LINEAR arithmetic (11-bit fixed point), restated -- cv2 is not available to pin it. */
int glsdet_ufp_mosaic(const unsigned char* img, int32_t H, int32_t W, const float* chips, int32_t n_chips,
                      float* canvas, int32_t ch, int32_t cw, void* stream);
/* mmdet test pipeline on an fp32 HWC BGR image: bilinear resize to nh x nw (half-pixel centres, edge
 * clamp), BGR->RGB, (v - mean) * (1/std), zero padding to ph x pw, HWC->CHW
 * (mmdet/datasets/pipelines/transforms.py:30,671,572).  mean/std: host double[3], RGB order.      */
int glsdet_resize_normalize_pad(const float* src, int32_t h, int32_t w, int32_t nh, int32_t nw, float* dst,
                                int32_t ph, int32_t pw, const double* mean_rgb, const double* std_rgb, void* stream);
/* the same on a uint8 HWC BGR frame (the first stage: cv2.imread -> mmcv.imresize): cv2.resize's uint8
 * INTER_LINEAR (11-bit fixed point, rounded to uint8) before Normalize.  An exact 2x downscale, which
 * OpenCV silently turns into INTER_AREA, is NOT special-cased. */
int glsdet_resize_normalize_pad_u8(const unsigned char* src, int32_t h, int32_t w, int32_t nh, int32_t nw, float* dst,
                                   int32_t ph, int32_t pw, const double* mean_rgb, const double* std_rgb,
                                   void* stream);
/* back-mapping + merge NMS (:282-300): dets = the fine detector's rows [max_det][7] (x1,y1,x2,y2,
 * score,score,label) with their count on the device; a row inside a chip's mosaic rectangle (IoF >
 * iof_thr) is mapped to source-image coordinates; per class greedy NMS with '+1' areas, a box
 * survives while IoU <= nms_thr (py_cpu_nms :149-178).  out: [max_out][7] in descending score
 * order, out_count int32[2] (kept clamped / unclamped), status bit0 = more than max_cand matches. */
int64_t glsdet_ufp_merge_workspace_bytes(int32_t max_cand);
int glsdet_ufp_backmap_merge(const float* dets, const int32_t* count, int32_t max_det, const float* chips,
                             int32_t n_chips, float iof_thr, float nms_thr, int32_t max_cand, int32_t max_out,
                             float* out, int32_t* out_count, int32_t* status, void* ws, int64_t ws_bytes,
                             void* stream);

/* ---------------------------------------------------------------------------------
 * bbox COCOeval: COCOeval.computeIoU + evaluateImg
 * (drone/models/core/cocoeval.py:163-190, 235-313; the protocol ufp/ufpmp_det_eval.py:333-338 runs
 * through pycocotools) for n_pairs (image, category) pairs in one call, all in fp64.
 *   dt_box [n_dt][4] x,y,w,h / dt_area [n_dt]: the pair's detections contiguous, in descending
 *       score order (stable), cut to the largest maxDets; dt_off int32 [n_pairs+1]
 *   gt_box [n_gt][4] / gt_area [n_gt] / gt_flags u8 [n_gt] (bit0 iscrowd, bit1 ignore, bit2 the
 *       annotation id is 0 -- the reference tests `dtm == 0` on stored ids); gt_off [n_pairs+1]
 *   iou_off int64 [n_pairs+1]: start of the pair's [D][G] block in `ious` (cumulative D*G)
 *   area_rng fp64 [n_area][2]; iou_thr fp64 [n_thr]
 * out (device, caller allocated; nothing needs clearing):
 *   ious       fp64 [iou_off[n_pairs]]   IoU, ground truths in annotation order (computeIoU)
 *   gt_order   int32 [n_area][n_gt]      position -> index inside the pair, ignored last (stable)
 *   gt_ignore  u8    [n_area][n_gt]      gtIgnore by position
 *   n_regular  int32 [n_area][n_pairs]   ground truths not ignored
 *   dt_match   int32 [n_area][n_thr][n_dt]  index inside the pair of the matched ground truth, -1
 *   dt_ignore  u8    [n_area][n_thr][n_dt]  dtIgnore (match ignored, or unmatched outside the range)
 *   gt_match   int32 [n_area][n_thr][n_gt]  by position: index of the matching detection, -1 */
int glsdet_coco_match(const double* dt_box, const double* dt_area, const int32_t* dt_off, const double* gt_box,
                      const double* gt_area, const unsigned char* gt_flags, const int32_t* gt_off,
                      const int64_t* iou_off, int32_t n_pairs, int32_t n_dt, int32_t n_gt,
                      const double* area_rng, int32_t n_area, const double* iou_thr, int32_t n_thr,
                      double* ious, int32_t* gt_order, unsigned char* gt_ignore, int32_t* n_regular,
                      int32_t* dt_match, unsigned char* dt_ignore, int32_t* gt_match, void* stream);

/* ---------------------------------------------------------------------------------
 * Plan: a recorded sequence of the calls above, replayed without Python in the loop and
 * capturable into one hipGraph (HIP streams + graphs instead of a tracing compiler).
 * Recording: glsdet_plan_begin(plan) makes every following entry-point call on this
 * thread append to the plan instead of launching; glsdet_plan_end() stops recording.
 */
typedef struct glsdet_plan glsdet_plan;
glsdet_plan* glsdet_plan_create(void);
void    glsdet_plan_destroy(glsdet_plan*);
int     glsdet_plan_begin(glsdet_plan*);
int     glsdet_plan_end(glsdet_plan*);
/* While recording: ops submitted under different non-zero branch ids (1..8) between two
 * branch-0 ops are declared independent of each other; replay runs them on side streams
 * forked from / joined into the main stream (parallel nodes of the captured hipGraph). */
int     glsdet_plan_set_branch(int32_t branch);
int32_t glsdet_plan_num_ops(const glsdet_plan*);
/* kind: 0 conv, 1 focus/pack, 2 pool, 3 resample/upsample-add, 4 nonlocal, 5 decode, 6 nms,
 * 7 groupnorm, 8 proxy scores; flops = 2*MACs */
int     glsdet_plan_op_info(const glsdet_plan*, int32_t i, int32_t* kind, double* flops,
                            double* bytes, char* name, int32_t name_cap);
int     glsdet_plan_run(glsdet_plan*, void* stream);               /* eager replay        */
int     glsdet_plan_capture(glsdet_plan*, void* stream);           /* build the hipGraph  */
int     glsdet_plan_launch(glsdet_plan*, void* stream);            /* hipGraphLaunch      */
/* eager replay with a hipEvent pair around every op; ms[i] accumulates milliseconds      */
int     glsdet_plan_run_timed(glsdet_plan*, void* stream, float* ms /*host, num_ops*/);

const char* glsdet_last_error(void);
int32_t     glsdet_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GLSDET_HIP_H */
