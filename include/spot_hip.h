/*
 * spot_hip.h - C ABI of the MI355X (gfx950) library for SPOT's tracker hot path.
 *
 * Conventions
 *   - Every entry point returns 0 on success or a negative code; the reason is
 *     available from spot_last_error() (thread-local text).  No C++ exception
 *     crosses this boundary.
 *   - All data pointers are DEVICE pointers unless a parameter says "host".
 *     The caller owns every buffer; nothing is retained past the call.
 *   - Work is enqueued on `stream` and not synchronised.
 *   - Token tensors are batch-first, row-major: [B, L, C].
 *
 * Each declaration cites the reference code it replaces (paths relative to the
 * VISION-SJTU/SPOT checkout).  The reference has no FFI for this path: its seam
 * is Python (torch.autograd.Function over compiled ops is its one precedent,
 * ltr/models/neck/dcn/deform_conv.py:12-94), so these are the functions a
 * ctypes binding in those Python modules calls.
 */
#ifndef SPOT_HIP_H
#define SPOT_HIP_H

#include <hip/hip_runtime_api.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPOT_HIP_VERSION 100

int spot_version(void);
const char* spot_last_error(void);

/* ---- pre-input --------------------------------------------------------- */

/* NormJitter + Normalize, ltr/models/pre_input/necessary_preprocess.py:54-67,86-91.
 * out = (clamp(float(images) * scale[b], 0, 1) - mean[c]) / std[c].
 * images [B,3,H,W] uint8, scale [B] (brightness/255; NULL = 1/255), mean3/std3 HOST
 * pointers to 3 floats, out [B,3,H,W] float.  16 pixels per lane when H*W is a multiple of 16 and both buffers are
 * 16-byte aligned, one otherwise (255x255 and 127x127 crops). */
int spot_preprocess_u8_f32(const unsigned char* images, const float* scale, const float* mean3,
                           const float* std3, float* out, int B, int C, int H, int W, hipStream_t stream);

/* PositionEmbeddingSine.forward (normalize = True), ltr/models/neck/position_encoding.py:27-45.  mask [B,H,W]
 * bytes (non-zero = padded cell) or NULL for no padding; out [B, H*W, 2*num_pos_feats] token rows: num_pos_feats
 * features of the row coordinate, then as many of the column coordinate (the reference's [B, 2F, H, W] map is the
 * transpose of each clip's rows).  H*W <= 4096. */
int spot_posenc_sine_f32(const unsigned char* mask, float* out, int B, int H, int W, int num_pos_feats,
                         float temperature, float scale, hipStream_t stream);

/* ---- pseudo-label path -------------------------------------------------- */

/* TranstSparseActor._decode_teacher_result, ltr/actor/transt_sparse_actor.py:57-98.
 * pred_logits [B,L,2], pred_boxes [B,L,4] (cx,cy,w,h in [0,1]), pred_ious [B,L],
 * window [L] (outer product of two float32 Hanning vectors).
 * -> best_bbox [B,4] (x0,y0,w,h px in the search crop; x0,y0 >= 0, w,h >= 1),
 *    best_score [B], best_conf [B], best_idx [B]. */
int spot_decode_teacher_f32(const float* pred_logits, const float* pred_boxes, const float* pred_ious,
                            const float* window, double window_penalty, int img_w, int img_h,
                            float* best_bbox, float* best_score, float* best_conf, int* best_idx,
                            int B, int L, hipStream_t stream);

/* SparseMiddleware.record_relay_results, ltr/data/middleware.py:49-152 with
 * recover_bbox_to_original / transform_image_to_crop, ltr/data/processing_utils.py:180-229.
 * aug_success[b] = best_conf[b] >= confidence_threshold (float32 compare, middleware.py:117).
 * box_extract [B,4], resize_factors [B] -> bbox_recovered [B,4], aug_anno [B,4], aug_success [B] (0/1). */
int spot_middleware_filter_f32(const float* best_bbox, const float* best_conf,
                               const float* unlabeled_box_extract, const float* unlabeled_resize_factors,
                               const float* aug_box_extract, const float* aug_resize_factors,
                               int search_sz, float confidence_threshold, float* bbox_recovered,
                               float* aug_anno, unsigned char* aug_success, int B, hipStream_t stream);

/* The two calls above fused into one launch (one workgroup per clip). */
int spot_decode_filter_f32(const float* pred_logits, const float* pred_boxes, const float* pred_ious,
                           const float* window, double window_penalty, int img_w, int img_h,
                           const float* unlabeled_box_extract, const float* unlabeled_resize_factors,
                           const float* aug_box_extract, const float* aug_resize_factors, int search_sz,
                           float confidence_threshold, float* best_bbox, float* best_score,
                           float* best_conf, int* best_idx, float* bbox_recovered, float* aug_anno,
                           unsigned char* aug_success, int B, int L, hipStream_t stream);

/* Keep/refill rule of LTRSparseTrainer._sparse_training, ltr/trainers/ltr_sparse_trainer.py:334-357:
 * kept rows in order, then cyclic repetition up to n_out rows.  gather_idx [n_out] (-1 everywhere when
 * nothing was kept: the caller skips the iteration, :328-332), n_keep [1].  n_in <= 12288. */
int spot_compact_indices(const unsigned char* aug_success, int n_in, int* gather_idx, int n_out,
                         int* n_keep, hipStream_t stream);

/* dst[r] = src[gather_idx[r]] for rows of row_bytes bytes (the torch.stack/cat of :342-357); a row whose index is
 * negative (nothing was kept) is filled with zero bytes. */
int spot_gather_rows(const void* src, void* dst, const int* gather_idx, int n_rows, long row_bytes,
                     hipStream_t stream);

/* TransT_Tracker._convert_score + window penalty + arg-max, pysot_toolkit/trackers/transt_tracker.py:21-32,164-171.
 * window [L] float64 (numpy hanning outer product), pscore = float64(score32 * float32(1-wp)) + window*wp.
 * best6 [B,6] float64 = (pscore, token index, cx, cy, w, h of that token, still normalised);
 * pscore_map [B,L] float64 or NULL. */
int spot_tracker_decode(const float* pred_logits, const float* pred_boxes, const double* window,
                        double window_penalty, double* best6, double* pscore_map, int B, int L,
                        hipStream_t stream);

/* TrackingMatcher.forward, ltr/models/loss/transt_matcher.py:24-109, as dense maps: pos_mask[b,t] = 1 for
 * tokens whose cell lies in the ground-truth box gt_cxcywh[b] (normalised cx,cy,w,h; float64 ceil arithmetic
 * as the reference's Python floats; cell ranges cut the grid the way its numpy slices do, negative bounds counting
 * from the end), neg_mask[b,t] = how many of the four rectangles of the ring of width gap around it (:92-95) hold
 * the token: 0 or 1 for a box on the map, up to 4 once a bound has wrapped (NULL to skip: the reference builds
 * negatives for the supervised tag only).  Both are [B, feat_sz*feat_sz] bytes. */
int spot_match_masks(const float* gt_cxcywh, int B, int feat_sz, int gap, unsigned char* pos_mask,
                     unsigned char* neg_mask, hipStream_t stream);

/* ---- teacher EMA --------------------------------------------------------- */

/* _update_teacher_model, ltr/actor/transt_sparse_actor.py:11-33, on flat storage:
 * teacher = student * (1 - keep_rate) + teacher * keep_rate. */
int spot_ema_lerp_f32(float* teacher, const float* student, long n, double keep_rate, hipStream_t stream);
/* same for int64 buffers (BatchNorm num_batches_tracked): float32 arithmetic, truncated. */
int spot_ema_lerp_i64(long long* teacher, const long long* student, int n, double keep_rate,
                      hipStream_t stream);

/* ---- boxes ---------------------------------------------------------------- */

/* box_iou_pair_xyxy / generalized_box_iou_pair_xyxy (+ box_cxcywh_to_xyxy when cxcywh != 0),
 * ltr/utils/box_ops.py:9-13,105-156.  boxes [P,4]; any of giou/iou/union_area may be NULL. */
int spot_box_iou_pair_f32(const float* boxes1, const float* boxes2, int cxcywh, float* giou, float* iou,
                          float* union_area, int P, hipStream_t stream);
/* gradient of (giou, iou) w.r.t. boxes1 (cxcywh) for upstream d_giou / d_iou (either may be NULL). */
int spot_box_giou_pair_bwd_f32(const float* boxes1_cxcywh, const float* boxes2_cxcywh, const float* d_giou,
                               const float* d_iou, float* d_boxes1, int P, hipStream_t stream);

/* Set criterion of the transformer tracker, SetCriterionTransT.forward with the token sets of TrackingMatcher given as
 * dense maps (ltr/models/tracking/transt_criterion.py:40-109,136-170; spot_match_masks writes pos / neg).
 * logits [B,L,2], boxes [B,L,4] cxcywh, iou_pred [B,L] or NULL (no IoU head: the unsupervised branch), gt [B,4] cxcywh,
 * pos [B,L] u8, neg [B,L] u8 multiplicities or NULL; w_obj / w_eos = the class weights (1, eos_coef).
 * losses[5] = (loss_ce, loss_bbox, loss_giou, iou, loss_iou_pred); norms[3] = (sum of class weights, max(#pos,1),
 * max(#neg,1)) is what the backward call divides by.  Two launches, no atomics: the same bits on every run.
 * ws: spot_transt_loss_ws_floats() floats. */
long spot_transt_loss_ws_floats(void);
int spot_transt_loss_f32(const float* logits, const float* boxes, const float* iou_pred, const float* gt,
                         const unsigned char* pos, const unsigned char* neg, float w_obj, float w_eos, int B, int L,
                         float* losses, float* norms, float* ws, hipStream_t stream);
/* Gradient of the four trained terms w.r.t. logits / boxes / iou_pred (any output may be NULL).  g_* point at the
 * upstream gradient of each term on the device (NULL = 0); the IoU target of the IoU head is a constant, as in
 * transt_criterion.py:92,103. */
int spot_transt_loss_bwd_f32(const float* logits, const float* boxes, const float* iou_pred, const float* gt,
                             const unsigned char* pos, const unsigned char* neg, float w_obj, float w_eos,
                             const float* norms, const float* g_ce, const float* g_bbox, const float* g_giou,
                             const float* g_iou_pred, int B, int L, float* d_logits, float* d_boxes, float* d_iou_pred,
                             hipStream_t stream);

/* ---- depthwise cross-correlation head ------------------------------------- */

/* xcorr_depthwise, ltr/models/neck/ocean_fusion.py:238-247.  x [planes,hx,wx], kernel [planes,hz,wz]
 * (planes = B*C) -> out [planes, hx-hz+1, wx-wz+1]. */
int spot_xcorr_depthwise_f32(const float* x, const float* kernel, float* out, int planes, int hx, int wx,
                             int hz, int wz, hipStream_t stream);
/* GroupDW.forward, ltr/models/neck/ocean_fusion.py:211-235: sum_i softmax(weight3)_i * xcorr(x_i, z_i) with
 * branch shapes (hx,wx)/(hz,wz), (hx-2,wx)/(hz-2,wz), (hx,wx-2)/(hz,wz-2) (matrix, :117-171). */
int spot_groupdw_f32(const float* x11, const float* x12, const float* x21, const float* z11,
                     const float* z12, const float* z21, const float* weight3, float* out, int planes,
                     int hx, int wx, int hz, int wz, hipStream_t stream);

/* ---- dense contractions (fp32 on the matrix cores) ------------------------- */

/* One GEMM problem: C[M,N] (+)= alpha * opA(A) * opB(B) + epilogue.  Carries nn.Linear forward
 * (F.linear: transt_fusion.py:151,234,240; transt_model.py:308-309), the packed in-/out-projections
 * of nn.MultiheadAttention (:121,172-175), the 1x1 input_proj conv (transt_model.py:49,274-275)
 * and all of their gradients.
 *   transA = 0: A[m*lda + k]        transA = 1: A[k*lda + m]
 *   transB = 0: B[n*ldb + k] (an nn.Linear weight [N,K])   transB = 1: B[k*ldb + n]
 *   batch > 1: problem z uses A + z*strideA, B + z*strideB, C + z*strideC, residual + z*strideRes.
 *   kseg > 0: the K range is cut into segments of kseg (multiple of 32); segment s reads A + s*segA,
 *             B + s*segB with k counted from 0 again (a reduction that runs over several batch slabs).
 *   splitk > 1: K is split over splitk workgroups that add atomically into C (set accumulate = 1 and
 *             clear C first); not combinable with relu / mask.
 *   rowsum: see the field.
 * Epilogue order: v = alpha*acc + bias[n] + rowbias[(m % rb_rows)*ldrb + n]; inference BatchNorm over rows (bn_*);
 *                 relu (relu = 1); dropout(drop_p, drop_seed); v = mask[m*ldmask+n] > 0 ? v : 0;
 *                 v += residual[m*ldres + n]; relu (relu = 2); C = v or C += v. */
typedef struct spot_gemm_desc {
    const float* A;
    const float* B;
    float* C;
    int M, N, K, batch;
    int transA, transB;
    long lda, ldb, ldc;
    long strideA, strideB, strideC;
    int kseg;
    long segA, segB;
    int splitk;
    float alpha;
    int accumulate;
    const float* bias;
    const float* rowbias;
    int rb_rows;
    long ldrb;
    int relu;
    const float* mask;
    long ldmask;
    const float* residual;
    long ldres, strideRes;
    float drop_p;        /* dropout after relu and before mask/residual: v = keep ? v/(1-p) : 0; 0 = off */
    unsigned drop_seed;  /* mask = f(drop_seed, global row z*M + m, column n): see spot_dropout_f32 */
    int tile;            /* workgroup tile: 0 = 64x64 or 128x128, chosen from the problem shape (bit-identical results, so
                          * the arithmetic never depends on the batch size); 64 / 128 force one.  32 = a 32x32 tile whose
                          * four waves split every 128-wide k step, for launches too small to fill the chip (one clip: the
                          * online trackers): deterministic, but another summation order; needs M, N % 32 == 0, K and every
                          * K segment % 128 == 0, aligned operands, no split-K / rowsum, else refused */
    int rb_cols;         /* rowbias covers columns n < rb_cols only (0 = all N columns) */
    const float* bn_gamma;  /* [M] each, or bn_gamma = null: inference-mode BatchNorm over the rows, applied after the */
    const float* bn_beta;   /* bias terms: v = v*a + (beta[m] - mean[m]*a), a = gamma[m] / sqrt(var[m] + bn_eps)      */
    const float* bn_mean;   /* (nn.BatchNorm2d in eval() behind a 1x1 convolution whose output channels are the rows: */
    const float* bn_var;    /* ltr/models/backbone/resnet.py:77-92 as the tracker and an eval() teacher run it)       */
    float bn_eps;
    float* rowsum;       /* [M] or null: rowsum[m] += sum_k opA(A)[m][k], added atomically (clear it first).  Only with
                          * transA = transB = 1, batch = 1: the weight-gradient call dW = dY^T X, where this is the bias
                          * gradient sum_rows dY (the sum nn.Linear's backward takes separately) */
    int precision;       /* 0 = exact fp32 products (the product path).  1 = "bf16x3", an experiment: each operand is cut
                          * into three bf16 pieces (exactly) and six of the nine piece products run on the bf16 matrix
                          * instruction with fp32 accumulation; error per product below 2^-23 |a||b|.  Only K and every
                          * K segment % 32 == 0, M % 128 == 0, N % 128 == 0, 16-byte aligned operands, no rowsum;
                          * anything else is refused */
} spot_gemm_desc;

int spot_gemm_f32(const spot_gemm_desc* desc /* host */, hipStream_t stream);
/* Two independent problems.  One launch when both ask for tile = 32 with the same transA / transB and batch = 1 (one clip:
 * FeatureFusionLayer.forward_post runs the same products on the template and on the search stream with different weights,
 * transt_fusion.py:204-244, and such launches last as long as their latencies); otherwise the two calls one after the
 * other.  The arithmetic of either problem is that of spot_gemm_f32. */
int spot_gemm_pair_f32(const spot_gemm_desc* desc0 /* host */, const spot_gemm_desc* desc1 /* host */, hipStream_t stream);

/* Bias-free 3x3 convolution, stride 1, dilation d = 1 or 2 with padding d, NCHW fp32 (conv2 of the ResNet
 * bottlenecks that do not stride: ltr/models/backbone/resnet.py:79-81; d = 2 in layer3's first block, :166), by
 * Winograd F(4x4, 3x3): weight / input transforms, 36 batched GEMMs on the matrix cores, output transform.  A
 * dilated convolution runs as d*d undilated ones on the polyphase sub-maps.  H and W multiples of 4*d.
 *   flip = 0: out[B,Cout,H,W] = conv2d(in[B,Cin,H,W], w[Cout,Cin,3,3])                      (forward)
 *   flip = 1: out[B,Cout,H,W] = conv_transpose2d(in[B,Cin,H,W], w[Cin,Cout,3,3])            (input gradient: in = dy,
 *             w = the forward weight as stored, Cin = its output channels, Cout = its input channels)
 * ws: spot_conv3x3_winograd_ws_floats(B, Cin, Cout, H, W) floats of scratch.  opts: see the struct.  Results differ from a direct
 * convolution by fp32 rounding of the transforms (~1e-6 relative to the largest term). */
/* Training-mode BatchNorm + ReLU of the convolution's INPUT, applied inside the input transform (bn1 -> relu -> conv2
 * of a bottleneck, ltr/models/backbone/resnet.py:77-81, when the normalised map has no other reader): in = the raw
 * output of the convolution in front, stats_ws = what spot_batchnorm_stats_f32(in, stats_ws, B, Cin, H*W) left.  The
 * call stores save_mean / save_rstd [Cin], moves the running statistics (or leaves the update in deferred_stats, see
 * spot_batchnorm_apply_deferred) and writes relu_sign (one byte per pack of four elements, as
 * spot_batchnorm_fwd_train_f32 does) for spot_batchnorm_bwd_f32: same values as the stand-alone forward, bit for bit. */
typedef struct spot_bn_train_in {
    const float* stats_ws;
    const float* gamma;
    const float* beta;
    float* running_mean;            /* with running_var, or both NULL */
    float* running_var;
    long long* num_batches_tracked; /* or NULL */
    float* save_mean;
    float* save_rstd;
    unsigned char* relu_sign;       /* B*Cin*H*W/4 bytes, or NULL when no backward follows */
    float* deferred_stats;          /* [2*Cin] or NULL */
    float eps, momentum;
} spot_bn_train_in;
typedef struct spot_winograd_opts {
    float* v_keep;          /* 36*Cin*B*(H/4)*(W/4) floats or NULL: receives the transformed input, for
                             * spot_conv3x3_winograd_wgrad_f32 to reuse                                                */
    float* u_keep;          /* 36*Cout*Cin floats or NULL: home of the transformed weights across calls                */
    int u_valid;            /* u_keep already holds the transform of these weights (frozen weights): not recomputed    */
    const float* bn_gamma;  /* [Cout] each, or bn_gamma = NULL: inference-mode BatchNorm of the output channels,       */
    const float* bn_beta;   /* y = y*a + (beta - mean*a), a = gamma / sqrt(var + bn_eps), then ReLU if relu != 0,      */
    const float* bn_mean;   /* applied in the output transform (nn.BatchNorm2d in eval() + ReLU behind conv2,          */
    const float* bn_var;    /* ltr/models/backbone/resnet.py:79-81, as the tracker and an eval() teacher run them)     */
    float bn_eps;
    int relu;
    const spot_bn_train_in* in_bn;  /* host pointer or NULL; dilation 1, flip 0 only */
} spot_winograd_opts;
long spot_conv3x3_winograd_ws_floats(int B, int Cin, int Cout, int H, int W);
int spot_conv3x3_winograd_f32(const float* in, const float* w, float* out, float* ws, const spot_winograd_opts* opts /* host, may be NULL */,
                              int B, int Cin, int Cout, int H, int W, int dilation, int flip, hipStream_t stream);

/* Weight gradient of the same convolution in the transformed domain: dw[Cout,Cin,3,3] (+)= G^T (sum over tiles of
 * (A dY A^T) (B^T d B)) G, the tile sum being 36 batched GEMMs.  x [B,Cin,H,W] is the forward input, dy [B,Cout,H,W]
 * the output gradient; accumulate = 1 adds into dw (a slice of a gradient buffer).  v_kept: the forward call's v_keep
 * buffer; with it x may be NULL and the input transform is not repeated. */
long spot_conv3x3_winograd_wgrad_ws_floats(int B, int Cin, int Cout, int H, int W);
int spot_conv3x3_winograd_wgrad_f32(const float* x, const float* v_kept, const float* dy, float* dw, float* ws, int B,
                                    int Cin, int Cout, int H, int W, int dilation, int accumulate, hipStream_t stream);

/* The same Winograd path for any map size, dilation 1 or 2 per axis and padding = pad * dilation with pad 0, 1 or 2
 * (the 3x3 convolutions of the Siamese head and backbone on 15x15 ... 63x63 maps, some dilated along one axis only:
 * ltr/models/neck/ocean_fusion.py:56-85,126-158, ltr/models/backbone/ocean_backbone.py:50-51): in [B,Cin,Hin,Win] ->
 * out [B,Cout,Hin+(2pad-2)dil_h,Win+(2pad-2)dil_w], bias [Cout] or NULL added on the way out.  flip = 1 with
 * pad' = 2 - pad on the output gradient gives the input gradient.  opts as for spot_conv3x3_winograd_f32: v_keep
 * (spot_conv3x3_winograd_general_v_floats(...) floats) receives the transformed input for the weight gradient,
 * u_keep / u_valid keep the transformed weights across calls, bn_* / relu fold inference BatchNorm (+ ReLU) of the
 * output channel into the output transform (applied after the bias).
 * The weight gradient takes x (or the kept v) and dy [B,Cout,Ho,Wo]; accumulate = 1 adds into dw. */
long spot_conv3x3_winograd_general_ws_floats(int B, int Cin, int Cout, int Hin, int Win, int pad, int dil_h, int dil_w);
long spot_conv3x3_winograd_general_v_floats(int B, int Cin, int Hin, int Win, int pad, int dil_h, int dil_w);
int spot_conv3x3_winograd_general_f32(const float* in, const float* w, const float* bias, float* out, float* ws,
                                      const spot_winograd_opts* opts /* host, may be NULL */, int B, int Cin, int Cout,
                                      int Hin, int Win, int pad, int dil_h, int dil_w, int flip, hipStream_t stream);
int spot_conv3x3_winograd_general_wgrad_f32(const float* x, const float* v_kept, const float* dy, float* dw, float* ws,
                                            int B, int Cin, int Cout, int Hin, int Win, int pad, int dil_h, int dil_w,
                                            int accumulate, hipStream_t stream);

/* Patch matrix of a convolution (the 7x7 / stride 2 stem and layer2's 3x3 / stride 2 of the ResNet,
 * ltr/models/backbone/resnet.py:61,79,195-196, and the 3x3 convolutions of the Siamese head on odd-sized maps,
 * ltr/models/neck/ocean_fusion.py:56-85,126-158, which then run as GEMMs: spot_amd/ops.py:ConvIm2colFn):
 * cols[B][Kpad][pitch], row (ci*kh + ky)*kw + kx holds x[b][ci][oy*stride - pad + ky*dilation_h][ox*stride - pad +
 * kx*dilation_w] at column oy*Wo + ox (zero outside the map); rows from C*kh*kw up to Kpad and columns from Ho*Wo
 * up to pitch are zero.  pitch >= Ho*Wo and a multiple of 4.
 * spot_col2im_f32 is its transpose as a gather: dx[B][C][H][W] = sum of the dcols entries that read that pixel. */
int spot_im2col_f32(const float* x, float* cols, int B, int C, int H, int W, int kh, int kw, int stride, int pad,
                    int dilation_h, int dilation_w, int Kpad, int pitch, hipStream_t stream);
int spot_col2im_f32(const float* dcols, float* dx, int B, int C, int H, int W, int kh, int kw, int stride, int pad,
                    int dilation_h, int dilation_w, int Kpad, int pitch, hipStream_t stream);

/* nn.MaxPool2d(kernel_size=3, stride=2, padding=1) of the ResNet stems (ltr/models/backbone/resnet.py:113,199;
 * ocean_backbone.py:97 on its odd 61x61 map) on `planes` = B*C maps of H x W -> Ho x Wo = ((H-1)/2+1) x ((W-1)/2+1).
 * Maps with even H and W a multiple of 8 take a vectorised kernel (4 outputs per thread), any other size one thread per
 * element.  winner [planes, Ho, Wo] (or null in inference) records which of the nine window positions held the maximum
 * (the first one in row-major order on ties, as ATen); the backward gathers dy through it: dx[planes, H, W] is written
 * once, no atomics. */
int spot_maxpool3x3s2_fwd_f32(const float* x, float* y, unsigned char* winner, long planes, int H, int W,
                              hipStream_t stream);
int spot_maxpool3x3s2_bwd_f32(const float* dy, const unsigned char* winner, float* dx, long planes, int H, int W,
                              hipStream_t stream);

/* ---- In-network strong augmentation of a crop batch (ltr/models/pre_input/transforms_in_net.py), NCHW fp32.
 * EllipseRotationAffine.forward's image part (:168-178 with :240-331): for samples with apply[b] != 0, a left/right
 * flip (lr_flip[b]) followed by a rotation by the integer angle (int)angle[b] degrees about the crop centre, nearest
 * sampling, pixels from outside replaced by fill[b][c]; other samples are copied.  theta_table [721][6]: the inverse
 * affine matrices of the integer angles -360..360 in centred pixel coordinates (host, double precision, as the
 * reference's _get_inverse_affine_matrix(center=[0.5, 0.5], angle=-a, ...) computes them).  out != in. */
int spot_rotate_flip_nearest_f32(const float* in, float* out, const float* theta_table, const float* angle,
                                 const unsigned char* lr_flip, const unsigned char* apply, const float* fill, int B, int C,
                                 int H, int W, hipStream_t stream);
/* Its box part (:180-214): xywh boxes [B,4] follow as the bounding boxes of their inscribed ellipses (360 sample
 * points) after the flip and the rotation by angle[b] degrees about (W/2, H/2); samples with apply[b] == 0 keep their
 * box; all outputs are clamped at 0. */
int spot_ellipse_rotate_boxes_f32(const float* bbox, const float* angle, const unsigned char* lr_flip,
                                  const unsigned char* apply, float* out, int B, int W, int H, hipStream_t stream);
/* x[p][:] *= scale[p] for planes p of HW floats, in place (16-byte accesses when HW is a multiple of 4): the per-channel weighting of
 * GrayScale.forward (:384-396) with scale = colour weight or 1 per (sample, channel). */
int spot_channel_scale_f32(float* x, const float* scale, long planes, int HW, hipStream_t stream);

/* nn.Dropout on a [rows, cols] tensor (transt_fusion.py:124,130-131,178,182,191-196) with a counter-based
 * mask keep(seed, row, col): out = keep ? x/(1-p) : 0.  The same (seed, row, col) function is used by the GEMM
 * epilogue, so the backward pass re-applies a forward mask by calling this on the incoming gradient. */
int spot_dropout_f32(const float* x, float* out, long rows, int cols, float p, unsigned seed,
                     hipStream_t stream);

/* out[n] += sum_m x[m*ld + n]  (bias gradients); out must be cleared by the caller. */
int spot_colsum_f32(const float* x, long ld, float* out, long rows, int N, hipStream_t stream);
/* out[r*C + c] = sum_b x[(b*R + r)*ld + c]: the sum over nb clips of a [nb*R, C] block of token rows, which is what
 * positional rows shared by every clip receive (q = k = src + pos, ltr/models/neck/transt_fusion.py:143-145).  C and ld
 * multiples of 4. */
int spot_batch_sum_f32(const float* x, long ld, int nb, long R, int C, float* out, hipStream_t stream);
/* out[c] = sum over clips and pixels of x [B,C,HW]: the bias gradient of a convolution (nn.Conv2d(bias=True) in the
 * Siamese head's towers, ltr/models/neck/ocean_fusion.py:56-85).  Fixed order, no atomics. */
int spot_channel_sum_f32(const float* x, int B, int C, long HW, float* out, hipStream_t stream);

/* ---- attention core ---------------------------------------------------------- */

/* LIMIT of every entry point in this section: head_dim == 32, i.e. d_model = 32 * nhead.  That is the one geometry the
 * reference's settings use (hidden_dim 256, nheads 8: ltr/train_settings/spot/spot_transt_002.py:84-86, _003.py:84-86,
 * _003_got10k.py:84-86, _005.py:84-86); one head's 32 channels are exactly one matrix-instruction tile, which the
 * register layout of the kernels is built on.  Any other head_dim (the reference's constructor default d_model 512 with
 * 8 heads gives 64) returns SPOT_ERR_ARG with a message (spot_last_error); nothing is computed and no other path is taken. */

/* softmax(scale * Q K^T + key_padding_mask) V per head, the middle of nn.MultiheadAttention as called at
 * ltr/models/neck/transt_fusion.py:145-148,212-213,218-219,223-230.  q [B,Lq,nhead*32] with row stride ldq
 * (so q/k/v may be column windows of a packed projection buffer), k,v [B,Lk,nhead*32], out [B,Lq,nhead*32];
 * lse [B,nhead,Lq] (log-sum-exp of the scaled scores, NULL to skip; needed by the backward);
 * key_padding_mask [B,Lk] bytes, non-zero = ignore key, or NULL.  head_dim must be 32.
 * drop_p > 0: dropout on the attention probabilities (nn.MultiheadAttention(dropout=...), :121,172-175) with a
 * counter-based mask of (drop_seed, clip, head, query, key).
 * drop_bits (NULL, or spot_mha_drop_bits_words(B, nhead, Lq, Lk) 32-bit words): with drop_p > 0 the forward also
 * writes the mask it drew, one bit per (query, key), for the backward call, which then reads the words with scalar
 * loads and spends one select per element on the mask instead of recomputing the hash (1024 x 1024, 40 clips:
 * backward 1137 -> 1040 us, forward 434 -> 440 us).  The buffer is 8-byte aligned; its word order is private to the
 * two calls.  A backward call given NULL regenerates the mask from the seed; both give the same gradients. */
long spot_mha_drop_bits_words(int B, int nhead, int Lq, int Lk);
int spot_mha_fwd_f32(const float* q, long ldq, const float* k, long ldk, const float* v, long ldv,
                     float* out, long ldo, float* lse, const unsigned char* key_padding_mask, int B,
                     int Lq, int Lk, int nhead, int head_dim, float scale, float drop_p, unsigned drop_seed,
                     unsigned* drop_bits, hipStream_t stream);

/* The same forward for grids too small to fill the chip (the online tracker: one clip): the keys are cut in nsplit
 * ranges, one workgroup each, and merged by a second kernel.  Inference only (no dropout, no lse).  ws:
 * spot_mha_fwd_splitk_ws_floats(B, Lq, nhead, nsplit) floats. */
long spot_mha_fwd_splitk_ws_floats(int B, int Lq, int nhead, int nsplit);
int spot_mha_fwd_splitk_f32(const float* q, long ldq, const float* k, long ldk, const float* v, long ldv, float* out,
                            long ldo, const unsigned char* key_padding_mask, float* ws, int nsplit, int B, int Lq, int Lk,
                            int nhead, int head_dim, float scale, hipStream_t stream);
/* Two such calls of one clip each (B = 1) in two launches instead of four: the attention blocks of a fusion layer's
 * template and search streams (transt_fusion.py:212-230).  ws0 / ws1: spot_mha_fwd_splitk_ws_floats(1, Lq, nhead, nsplit). */
int spot_mha_fwd_splitk_pair_f32(const float* q0, long ldq0, const float* k0, long ldk0, const float* v0, long ldv0,
                                 float* out0, long ldo0, const unsigned char* key_padding_mask0, float* ws0, int nsplit0,
                                 int Lq0, int Lk0, const float* q1, long ldq1, const float* k1, long ldk1, const float* v1,
                                 long ldv1, float* out1, long ldo1, const unsigned char* key_padding_mask1, float* ws1,
                                 int nsplit1, int Lq1, int Lk1, int nhead, int head_dim, float scale, hipStream_t stream);

/* spot_mha_fwd_splitk_f32 without the workspace argument, for callers that cannot allocate device memory: a frame
 * replayed under HIP graph capture, a serving loop or a harness that forbids hipMalloc.  It has no counterpart in the
 * reference.  The partial sums go to a static device array of SPOT_MHA_SPLITK_NOALLOC_MAX_FLOATS floats (8.5 MiB,
 * held by every GPU the library runs on until the process ends); a call needing more, i.e.
 * spot_mha_fwd_splitk_ws_floats(B, Lq, nhead, nsplit) > SPOT_MHA_SPLITK_NOALLOC_MAX_FLOATS, is rejected.  The call
 * makes no hipMalloc / hipFree.  Same results as spot_mha_fwd_splitk_f32 bit for bit.  Calls that overlap on one GPU
 * (two streams, two threads) share the array and race: serialise them.  The FIRST call on a device resolves the
 * array's address (hipGetDevice + hipGetSymbolAddress, which may load the code object): make one eager call per
 * device before any stream capture; a first call made while capturing can invalidate the capture. */
#define SPOT_MHA_SPLITK_NOALLOC_MAX_FLOATS 2228224L   /* 8 key ranges x 8 heads x 1024 queries x (32 + 2) */
int spot_mha_fwd_splitk_noalloc_f32(const float* q, long ldq, const float* k, long ldk, const float* v, long ldv,
                                    float* out, long ldo, const unsigned char* key_padding_mask, int nsplit, int B,
                                    int Lq, int Lk, int nhead, int head_dim, float scale, hipStream_t stream);

/* Gradient of spot_mha_fwd_f32 w.r.t. q, k, v given d_out.  delta_ws: no longer read or written (the row sums of
 * out * d_out are taken inside the kernel); may be NULL.  dq is accumulated with atomic adds and is cleared by this
 * call. */
int spot_mha_bwd_f32(const float* q, long ldq, const float* k, long ldk, const float* v, long ldv,
                     const float* out, long ldo, const float* d_out, long lddo, const float* lse,
                     const unsigned char* key_padding_mask, float* delta_ws, float* dq, long lddq,
                     float* dk, long lddk, float* dv, long lddv, int B, int Lq, int Lk, int nhead,
                     int head_dim, float scale, float drop_p, unsigned drop_seed, const unsigned* drop_bits,
                     hipStream_t stream);
/* Same contracts as spot_mha_fwd_f32 / spot_mha_bwd_f32, with the five matrix products taken in bf16
 * (operands rounded to bf16 in registers, fp32 accumulation, fp32 softmax): the "bf16 MFMA attention" training
 * configuration.  Inputs, outputs and saved statistics stay fp32. */
int spot_mha_fwd_bf16(const float* q, long ldq, const float* k, long ldk, const float* v, long ldv,
                      float* out, long ldo, float* lse, const unsigned char* key_padding_mask, int B, int Lq,
                      int Lk, int nhead, int head_dim, float scale, float drop_p, unsigned drop_seed,
                      unsigned* drop_bits, hipStream_t stream);
int spot_mha_bwd_bf16(const float* q, long ldq, const float* k, long ldk, const float* v, long ldv,
                      const float* out, long ldo, const float* d_out, long lddo, const float* lse,
                      const unsigned char* key_padding_mask, float* delta_ws, float* dq, long lddq, float* dk,
                      long lddk, float* dv, long lddv, int B, int Lq, int Lk, int nhead, int head_dim,
                      float scale, float drop_p, unsigned drop_seed, const unsigned* drop_bits, hipStream_t stream);
/* The mask the two calls above use, as bytes [B,nhead,Lq,Lk] (1 = kept); for host-side verification. */
int spot_mha_dropout_mask(unsigned char* mask, int B, int nhead, int Lq, int Lk, float drop_p,
                          unsigned drop_seed, hipStream_t stream);

/* ---- LayerNorm ----------------------------------------------------------------- */

/* nn.LayerNorm over the last dim (transt_fusion.py:127-128,185-190,26): y = (x-mean)*rstd*gamma + beta.
 * x,y [rows,C]; mean,rstd [rows] (NULL to skip).  C multiple of 4, <= 1024. */
int spot_layernorm_fwd_f32(const float* x, const float* gamma, const float* beta, float* y, float* mean,
                           float* rstd, long rows, int C, float eps, hipStream_t stream);
/* Two tensors of the same width in one launch, inference (no statistics kept): norm1x / norm2x of a fusion layer's two
 * streams (transt_fusion.py:214-244). */
int spot_layernorm_fwd_pair_f32(const float* x0, const float* gamma0, const float* beta0, float* y0, long rows0, float eps0,
                                const float* x1, const float* gamma1, const float* beta1, float* y1, long rows1, float eps1,
                                int C, hipStream_t stream);
/* dx, dgamma, dbeta (the latter two overwritten, or added to when accumulate != 0; either may be NULL).  ws: workspace of
 * spot_layernorm_bwd_ws_floats(rows, C) floats holding per-workgroup partial sums.
 * dx_dropped (NULL to skip): a second copy of dx behind the mask spot_dropout_f32(dx, drop_p, drop_seed) would apply -
 * the gradient behind the output dropout of the attention / feed-forward block whose result this LayerNorm normalised
 * (nn.Dropout at transt_fusion.py:149,152,214,232,241), which that block's backward would otherwise make with a pass
 * of its own over dx. */
long spot_layernorm_bwd_ws_floats(long rows, int C);
int spot_layernorm_bwd_f32(const float* dy, const float* x, const float* mean, const float* rstd,
                           const float* gamma, float* dx, float* dgamma, float* dbeta, float* ws, long rows,
                           int C, int accumulate, float* dx_dropped, float drop_p, unsigned drop_seed,
                           hipStream_t stream);

/* Gradients of the two calls above.  d_out [planes, hx-hz+1, wx-wz+1]; dx / dkernel may be NULL to skip. */
int spot_xcorr_depthwise_bwd_f32(const float* x, const float* kernel, const float* d_out, float* dx,
                                 float* dkernel, int planes, int hx, int wx, int hz, int wz,
                                 hipStream_t stream);
/* dsoftmax_partial [planes,3]: per-plane <d_out, xcorr(x_i, z_i)>; summed over planes it is the gradient w.r.t.
 * softmax(weight3), from which d weight3 = s * (ds - <s, ds>) (NULL to skip). */
int spot_groupdw_bwd_f32(const float* x11, const float* x12, const float* x21, const float* z11,
                         const float* z12, const float* z21, const float* weight3, const float* d_out,
                         float* dx11, float* dx12, float* dx21, float* dz11, float* dz12, float* dz21,
                         float* dsoftmax_partial, int planes, int hx, int wx, int hz, int wz,
                         hipStream_t stream);
/* The last step of the line above: d_weight3[3] from dsoftmax_partial [planes,3] and weight3 (softmax mixing of
 * GroupDW.forward, ltr/models/neck/ocean_fusion.py:217-236).  One launch, planes summed in a fixed order. */
int spot_groupdw_weight_grad_f32(const float* dsoftmax_partial, int planes, const float* weight3, float* d_weight3,
                                 hipStream_t stream);

/* ---- Deformable convolution v1 (ltr/models/neck/dcn/src/deform_conv_cuda_kernel.cu:189-275 deformable_im2col,
 * :277-369 deformable_col2im, :371-463 deformable_col2im_coord; called from deform_conv_cuda.cpp:149-482 through
 * ltr/models/neck/dcn/deform_conv.py:12-110).  x [B,C,H,W]; offset [B, deformable_groups*2*kh*kw, Ho, Wo] holds
 * (dy, dx) per tap; square stride / padding / dilation.  The sampled patch matrix cols[B][Kpad][pitch] has the layout
 * of spot_im2col_f32 (row (ci*kh + ky)*kw + kx, column oy*Wo + ox, zero beyond C*kh*kw rows and Ho*Wo columns), so the
 * products with the weights are spot_gemm_f32 calls (spot_amd/ops.py:DeformConvFn).
 * spot_deform_col2im_f32: from dcols (= W^T dy, same layout) the input gradient dx [B,C,H,W] (cleared here, then
 * accumulated with float atomics; may be NULL) and the offset gradient doffset (same shape as offset; may be NULL;
 * needs ws of spot_deform_col2im_ws_floats(...) floats). */
int spot_deform_im2col_f32(const float* x, const float* offset, float* cols, int B, int C, int H, int W, int kh, int kw,
                           int stride, int pad, int dilation, int deformable_groups, int Kpad, int pitch,
                           hipStream_t stream);
long spot_deform_col2im_ws_floats(int B, int C, int H, int W, int kh, int kw, int stride, int pad, int dilation,
                                  int deformable_groups);
int spot_deform_col2im_f32(const float* dcols, const float* x, const float* offset, float* dx, float* doffset, float* ws,
                           int B, int C, int H, int W, int kh, int kw, int stride, int pad, int dilation,
                           int deformable_groups, int Kpad, int pitch, hipStream_t stream);

/* ---- Crop + pad + resize of a frame on the device (pysot_toolkit/trackers/transt_tracker.py:41-101 get_subwindow,
 * ltr/data/processing_utils.py:8-91 sample_target: numpy slicing + cv2.resize on the host there).  The frame is uint8
 * HWC in device memory; the window [y0, y0+crop) x [x0, x0+crop) (x0 / y0 = the reference's context_xmin / ymin, may
 * be negative) is filled with `pad` where it leaves the frame and resized to out_sz x out_sz, written uint8 CHW
 * [3, out_sz, out_sz].  crop == out_sz copies; crop == 2 out_sz is the 2x2 box average; anything else OpenCV's 8-bit
 * INTER_LINEAR arithmetic (weights in 1/2048, see csrc/crop.hip).  The batch form reads B descriptors from device
 * memory and writes [B, 3, out_sz, out_sz]: one launch for a loader's batch. */
typedef struct spot_crop_desc {
    const unsigned char* image;   /* [H, W, 3] */
    int H, W;
    int x0, y0;                   /* top-left corner of the window in frame coordinates */
    int crop;                     /* side of the window */
    unsigned char pad[4];         /* fill colour per channel (4th byte unused) */
} spot_crop_desc;
int spot_crop_resize_u8(const spot_crop_desc* crop /* host */, unsigned char* out, int out_sz, hipStream_t stream);
int spot_crop_resize_batch_u8(const spot_crop_desc* table_dev, int B, unsigned char* out, int out_sz, hipStream_t stream);

/* ---- Training labels of the Siamese (Ocean) head for B boxes (x, y, w, h on the search patch), one launch
 * (ltr/models/loss/ocean_label_gen.py:44-66,124-233).  S = response-map side, cell (r, c) is centred at
 * ((c - S/2) stride + search_sz/2, (r - S/2) stride + search_sz/2).  Any output may be NULL.
 *   cls   [B,S,S]   1 within `radius` (L1, in cells) of the cell the box centre falls in, else 0
 *   reg   [B,S,S,4] float64 distances from the cell centre to the box's left, top, right, bottom edge
 *   reg_w [B,S,S]   1 where all four are > 0 (the centre lies inside the box)
 *   iou_w [B,S,S]   the same test against the box grown by `expand` times its size (sizes clamped to >= 1) per side */
int spot_ocean_labels_f32(const float* anno_xywh, float* cls, double* reg, float* reg_w, float* iou_w, int B, int S,
                          int stride, int search_sz, float expand, int radius, hipStream_t stream);

/* One frame of the Siamese online tracker (pysot_toolkit/trackers/ocean_tracker.py:71-104): from the classification
 * and align logits [S,S], the ltrb map [4,S,S], the float64 cosine window [S,S] and the target size scaled to the
 * crop (target_wh: two float64 values in device memory, so that a captured launch can be replayed with a new size),
 * the winning cell after score mixing (ratio), scale / aspect penalty (penalty_k) and window blend
 * (window_influence).  out8 (device, float64) = (row, col, x1, y1, x2, y2, penalty, mixed score) of that cell. */
int spot_ocean_tracker_decode(const float* cls_logits, const float* align_logits, const float* ltrb, const double* window,
                              const double* target_wh, double ratio, double penalty_k, double window_influence, int S,
                              int stride, int instance_size, double* out8, hipStream_t stream);

/* ---- BatchNorm2d fused with the ReLU / residual add that follow it in the ResNet bottleneck
 * (ltr/models/backbone/resnet.py:72-92,196-198; the Siamese head's conv-bn-relu stacks,
 * ltr/models/neck/ocean_fusion.py:56-85,126-158), NCHW fp32.  Planes are walked in packs of four floats when HW is a
 * multiple of 4 and every map pointer is 16-byte aligned, element by element otherwise (a pack is then one element).
 * Training forward: batch statistics over (N, HW) per channel; y = [relu]((x - mean) rstd gamma + beta [+ residual]);
 * running_mean / running_var updated with `momentum` (unbiased variance), *num_batches_tracked += 1 (all three may be
 * NULL); save_mean / save_rstd [C] for backward; relu_sign: one byte per pack (bit j = element j of the pack passed
 * the ReLU; NULL when no gradient is needed; forward and backward must see the same pack width).  deferred_stats [2C] (may be NULL): receives the batch mean and the
 * unbiased batch variance for a later spot_batchnorm_apply_deferred (pass the running buffers as NULL then).
 * ws: spot_batchnorm_ws_floats(N, C) floats. */
long spot_batchnorm_ws_floats(int N, int C);
/* The statistics pass of the training forward alone: per-slice partial sums into ws (spot_batchnorm_ws_floats(N, C)
 * floats), for a consumer that applies the normalisation itself (spot_winograd_opts.in_bn). */
int spot_batchnorm_stats_f32(const float* x, float* ws, int N, int C, int HW, hipStream_t stream);
int spot_batchnorm_fwd_train_f32(const float* x, const float* residual, const float* gamma, const float* beta,
                                 float* running_mean, float* running_var, long long* num_batches_tracked, float* y,
                                 float* save_mean, float* save_rstd, unsigned char* relu_sign, float* deferred_stats,
                                 float* ws, int N, int C, int HW, float eps, float momentum, int relu,
                                 hipStream_t stream);
/* One pending running-statistics update: running = (1 - momentum) running + momentum stats, tracked += 1. */
typedef struct spot_bn_deferred {
    float* running_mean;              /* [C] */
    float* running_var;               /* [C] */
    long long* num_batches_tracked;   /* may be NULL */
    const float* stats;               /* [2C]: batch mean, unbiased batch variance (deferred_stats of the forward) */
    int C;
    float momentum;
} spot_bn_deferred;
/* Applies n pending updates (table in device memory), in table order, in one launch. */
int spot_batchnorm_apply_deferred(const spot_bn_deferred* table_dev, int n, hipStream_t stream);
/* Inference forward with the running statistics (one pass). */
int spot_batchnorm_fwd_infer_f32(const float* x, const float* residual, const float* gamma, const float* beta,
                                 const float* running_mean, const float* running_var, float* y, int N, int C,
                                 int HW, float eps, int relu, hipStream_t stream);
/* Backward of the training forward.  dy is the gradient of the fused output; relu_sign as written by forward (NULL
 * if no ReLU was fused).  dx: gradient w.r.t. x; d_residual (may be NULL): gradient w.r.t. the residual input
 * (= dy gated by the ReLU); dgamma / dbeta [C] overwritten, or added to when accumulate != 0. */
int spot_batchnorm_bwd_f32(const float* dy, const float* x, const unsigned char* relu_sign, const float* save_mean,
                           const float* save_rstd, const float* gamma, float* dx, float* d_residual, float* dgamma,
                           float* dbeta, float* ws, int N, int C, int HW, int accumulate, hipStream_t stream);

/* ---- AdamW over flat parameter storage (torch.optim.AdamW semantics: decoupled weight decay, no amsgrad; the
 * reference's optimizer, ltr/train_settings/spot/spot_transt_003.py:148-157), one launch per step.
 * The four buffers are parallel flat arrays; `chunks` (device memory) lists the trainable ranges, at most
 * 4096 elements each, with the parameter group each belongs to. */
typedef struct spot_adamw_chunk {
    long offset;     /* first element */
    int count;       /* elements (<= 4096 keeps workgroups balanced) */
    int group;       /* index into the per-group arrays below, 0..3 */
} spot_adamw_chunk;
typedef struct spot_adamw_groups {
    float lr[4];
    float weight_decay[4];
    float beta1, beta2, eps;
    float bias_correction1;         /* 1 - beta1^step */
    float sqrt_bias_correction2;    /* sqrt(1 - beta2^step) */
} spot_adamw_groups;
int spot_adamw_f32(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                   const spot_adamw_chunk* chunks_dev, int n_chunks, const spot_adamw_groups* hyper,
                   hipStream_t stream);

/* torch.nn.utils.clip_grad_norm_(parameters, max_norm) over flat gradient storage, as the trainer applies it to the
 * Siamese variant after backward (ltr/trainers/ltr_sparse_trainer.py:75-80,126-131): norm_out[0] (device) = the L2 norm
 * of grads[0..n); grads *= max_norm / (norm + 1e-6) when that is below 1 and the norm is finite (max_norm <= 0: the norm
 * only).  Deterministic (no atomics).  ws: spot_clip_grad_norm_ws_floats() floats.  n a multiple of 4. */
long spot_clip_grad_norm_ws_floats(void);
int spot_clip_grad_norm_f32(float* grads, long n, float max_norm, float* norm_out, float* ws, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SPOT_HIP_H */
