/* ============================================================================
 * s2anet_hip.h — C ABI of libs2anet_hip.so: the S2ANet dense-inference hot path
 * as hand-written HIP kernels for MI355X (gfx950, wave64).
 *
 * This is the drop-in boundary: every entry point replaces one
 * function of the reference's six pybind11 extension modules and takes plain
 * device pointers and sizes — no torch types.  All pointers are DEVICE pointers
 * unless a parameter is named host_*; every call is asynchronous on `stream`
 * (a hipStream_t passed as void*) unless documented otherwise.  Tensors are
 * dense row-major ("contiguous") like the reference requires
 * (models/utils.py:51-56, deform_conv_cuda.cpp:168-170).
 *
 * Return value: 0 on success, negative S2A_E* on error; s2a_last_error() gives
 * the message (thread-local).  The reference raises c10::Error -> RuntimeError
 * for the same conditions (deform_conv_cuda.cpp:62-150, nms_rotated_cuda.cu:80-82);
 * the Python host side maps S2A_E* back to RuntimeError.
 *
 * Workspaces: ops that need scratch take (workspace, workspace_bytes); query the
 * size with the matching *_workspace_bytes().  Nothing in this library calls
 * hipMalloc/hipFree/hipDeviceSynchronize on the hot path, so every launch
 * sequence is hipGraph-capturable; the only host synchronisations are the
 * explicitly documented `host_count` read-backs of the reference-shaped NMS calls.
 * ==========================================================================*/
#ifndef S2ANET_HIP_H_
#define S2ANET_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define S2A_OK 0
#define S2A_EINVAL (-1)      /* bad shape / argument (reference: TORCH_CHECK / AT_ERROR) */
#define S2A_EWORKSPACE (-2)  /* workspace too small */
#define S2A_EHIP (-3)        /* HIP runtime error (launch failure, ...) */
#define S2A_ENOTIMPL (-4)

#define S2A_DTYPE_F32 0
#define S2A_DTYPE_F16 1
#define S2A_DTYPE_F64 2 /* ARF forward / backward (AT_DISPATCH_FLOATING_TYPES, ActiveRotatingFilter_cuda.cu:104,149) and the generic
                           deformable convolution + its three backward kernels (deform_conv_cuda_kernel.cu:258,352,450) */

#define S2A_LAYOUT_NCHW 0 /* reference layout (contiguous NCHW) */
#define S2A_LAYOUT_NHWC 1 /* channels-last storage of the same logical tensor */

typedef void* s2a_stream_t; /* hipStream_t */

const char* s2a_last_error(void);
const char* s2a_version(void);

/* ---------------------------------------------------------------------------
 * Rotated-box IoU.  Replaces  utils.box_iou_rotated.box_iou_rotated_cuda.box_iou_rotated
 * (utils/box_iou_rotated/src/box_iou_rotated.h:22-42, kernel box_iou_rotated_cuda.cu:14-101).
 * boxes1[N,5], boxes2[M,5] f32 (x_ctr,y_ctr,w,h,angle_rad) -> ious[N,M] f32.
 * Arithmetic follows the reference's __CUDACC__ branch of box_iou_rotated_utils.h
 * operation-for-operation (no FMA contraction).  N==0 or M==0 is a no-op.
 * ------------------------------------------------------------------------- */
size_t s2a_box_iou_rotated_workspace_bytes(int64_t n, int64_t m);
int s2a_box_iou_rotated(const float* boxes1, int64_t n, const float* boxes2, int64_t m,
                        float* ious, void* workspace, size_t workspace_bytes,
                        s2a_stream_t stream);

/* element-wise variant (pairs i<->i), used by tests and by the merge path */
int s2a_box_iou_rotated_pairs(const float* boxes1, const float* boxes2, int64_t n, float* ious,
                              s2a_stream_t stream);

/* polyiou.iou_poly (DOTA_devkit/polyiou/csrc/polyiou.cpp:108-128; SWIG module `polyiou`), element-wise
 * over n pairs of quadrilaterals: polys[n,8] f64 (x1,y1,...,x4,y4) -> ious[n] f64.  Double precision,
 * same operation order as the reference (bit-identical to it on the tested inputs). */
int s2a_polyiou_pairs(const double* polys1, const double* polys2, int64_t n, double* ious,
                      s2a_stream_t stream);

/* The search inside voc_eval (DOTA_devkit/dota_evaluation_task1.py:204-263): for every detection polygon
 * dets8[d] (8 doubles) of image det_image[d], the ground-truth polygon gts8[g], g in
 * [gt_offsets[img], gt_offsets[img+1]), with the largest iou_poly(GT, det) among those whose axis-aligned boxes
 * overlap it (:222-246) -> ovmax[d] (-inf when none) and argmax[d] (index into gts8, -1 when none; first maximum). */
int s2a_polyiou_match(const double* dets8, const int32_t* det_image, int64_t num_dets, const double* gts8,
                      const int64_t* gt_offsets, int64_t num_images, double* ovmax, int64_t* argmax,
                      s2a_stream_t stream);

/* Label assignment of the training side, fused: assign_labels(anchors[M,5], gt_boxes[N,5], imgs_size, pos_iou_thr,
 * neg_iou_thr, min_pos_iou_thr, gt_max_assign_all, filter_invalid_anchors, filter_invalid_ious)
 * (models/utils.py:33-147) -> assign_gt_ids[M] int64 (-2 ignore, -1 negative, >= 0 the gt index).  The
 * [M,N] IoU matrix of bbox_iou_rotated (utils/metrics.py:85-107) lives in the workspace only (same pipeline and
 * values as s2a_box_iou_rotated); three streaming passes over it replace the reference's tensor ops and its
 * Python loop over the gts.  Row arg-max ties: first index. */
size_t s2a_assign_labels_workspace_bytes(int64_t num_anchors, int64_t num_gts);
int s2a_assign_labels(const float* anchors, int64_t num_anchors, const float* gt_boxes, int64_t num_gts,
                      float img_h, float img_w, float pos_iou_thr, float neg_iou_thr, float min_pos_iou_thr,
                      int gt_max_assign_all, int filter_invalid_anchors, int filter_invalid_ious,
                      int64_t* assign_gt_ids, void* workspace, size_t workspace_bytes, s2a_stream_t stream);

/* Chip-merge polygon NMS: py_cpu_nms_poly_fast(dets[n,9] f64 = 8 polygon coordinates + score, thresh)
 * (DOTA_devkit/ResultMerge_multi_process.py:62-123) entirely on the device.  keep[] (n int64) receives
 * the surviving original indices in descending-score order, *count_dev their number; host_count as in
 * s2a_nms_rotated.  Score ties: ascending index. */
size_t s2a_nms_poly_workspace_bytes(int64_t n);
int s2a_nms_poly(const double* dets9, int64_t n, double thresh, int64_t* keep, int64_t* count_dev,
                 int64_t* host_count, void* workspace, size_t workspace_bytes, s2a_stream_t stream);

/* ---------------------------------------------------------------------------
 * Rotated NMS.  Replaces
 *   utils.nms_rotated.nms_rotated_cuda.nms_rotated(dets[N,5], scores[N], thr) -> int64[K]
 *     (utils/nms_rotated/src/nms_rotated.h:21-41, nms_rotated_cuda.cu:14-133)
 *   utils.ml_nms_rotated.ml_nms_rotated_cuda.ml_nms_rotated(dets, scores, labels, thr)
 *     (utils/ml_nms_rotated/src/nms_rotated.h:23-44, nms_rotated_cuda.cu:14-137)
 * Semantics of the reference GPU op: sort by score descending, suppress j by a kept,
 * higher-scored i when iou(i,j) > thr (strict), IoU == 0 across different labels;
 * keep[] = indices into the ORIGINAL order, in descending-score order.
 * Ties in score are broken by ascending original index (the reference leaves them to
 * torch's unstable sort).
 *
 * Everything — sort, pair finding, exact IoU, greedy resolution, compaction — runs on the
 * device; the reference's N x N/64 mask and its device->host copy (cuda.cu:109) do not exist here:
 * suppressing pairs are kept as a list.  s2a_nms_rotated_workspace_bytes() sizes the lists for
 * dense inputs; a SMALLER workspace (>= the fixed part + 48 KB) is accepted, and a call whose
 * lists fill up finishes on a slower memory-free kernel with the same keep list.
 * Speed limit of the label handling (results are unaffected): inputs of more than 4 096 rows per label are ordered
 * spatially by an own counting sort whose tables hold at most 8 192 DISTINCT labels and 65 536 (label, cell) buckets;
 * beyond either the same memory-free kernel settles the call -- exact, but an order of magnitude slower (12 000 distinct
 * labels at n = 9 000 are tested for the keep list, tests/test_gpu_ops.py::test_nms_order_b_counting_sort_labels_of_any_kind).
 * The reference's use (utils/bbox_nms_rotated.py:47: labels = class ids, 15 for DOTA) is far inside the limit.
 * labels may be NULL (single class).  keep must hold n int64.  *count_dev (device
 * int64) receives K; if host_count != NULL the call synchronises the stream once and
 * stores K there (the reference call shape needs K on the host to size its result).
 * ------------------------------------------------------------------------- */
size_t s2a_nms_rotated_workspace_bytes(int64_t n, int64_t max_segment_rows);
int s2a_ml_nms_rotated(const float* dets, const float* scores, const float* labels, int64_t n,
                       float iou_threshold, int64_t* keep, int64_t* count_dev,
                       int64_t* host_count, void* workspace, size_t workspace_bytes,
                       s2a_stream_t stream);
int s2a_nms_rotated(const float* dets, const float* scores, int64_t n, float iou_threshold,
                    int64_t* keep, int64_t* count_dev, int64_t* host_count, void* workspace,
                    size_t workspace_bytes, s2a_stream_t stream);

/* Synchronous calls (host_count != NULL) of the two ops above on at most 16 384 rows are settled by ONE kernel launch when the
 * labels split the rows into at most 64 segments of at most 640 rows and at most 4 096 pairs survive the cull of a segment;
 * otherwise -- and always for host_count == NULL -- by the general multi-launch path (same keep list either way, tested).
 * Counters of the two outcomes since the library was loaded (diagnostic; tests assert that the small path really ran). */
int s2a_nms_small_stats(int64_t* taken, int64_t* fell_back);

/* The same two ops on float64 boxes.  The reference dispatches the NMS kernels on the dtype of `dets`
 * (AT_DISPATCH_FLOATING_TYPES_AND_HALF, utils/nms_rotated/src/nms_rotated_cuda.cu:95-100,
 * utils/ml_nms_rotated/src/nms_rotated_cuda.cu:100-105; AT_DISPATCH_FLOATING_TYPES in the CPU files): on double boxes it
 * evaluates single_box_iou_rotated<double>, and keep decisions next to the threshold differ from the float32 evaluation.
 * dets[n,5], scores[n], labels[n] (NULL: single class) float64; the threshold stays a float as in the reference's
 * signature.  A plain N x N/64 mask form (API completeness, not a hot path): n < 260 k.  Not HIP-graph capturable. */
size_t s2a_nms_rotated_f64_workspace_bytes(int64_t n);
int s2a_nms_rotated_f64(const double* dets, const double* scores, const double* labels, int64_t n,
                        float iou_threshold, int64_t* keep, int64_t* count_dev, int64_t* host_count,
                        void* workspace, size_t workspace_bytes, s2a_stream_t stream);

/* Batched form used by the detector (one call for a whole batch of images):
 * segment_ids[n] int32 in [0, num_segments) — NMS runs independently inside each
 * segment (segment = image*num_classes + label); a NEGATIVE segment id marks a padding row
 * that is ignored (never compared, never kept; lets callers use static-size buffers); group_ids[n] int32 in [0,num_groups)
 * (group = image) decides the output grouping: keep_flags[n] uint8 (1 = survives) and,
 * if keep != NULL, per-group lists keep[g*max_per_group + r] (int32 original index, score
 * descending, -1 padded) with group_counts[g] = min(K_g, max_per_group).
 * No host synchronisation. */
int s2a_nms_rotated_segmented(const float* dets, const float* scores, const int32_t* segment_ids,
                              const int32_t* group_ids, int64_t n, int32_t num_segments,
                              int32_t num_groups, float iou_threshold, uint8_t* keep_flags,
                              int32_t* keep, int32_t* group_counts, int32_t max_per_group,
                              void* workspace, size_t workspace_bytes, s2a_stream_t stream);

/* s2a_nms_rotated_segmented followed, in the same launch sequence, by the output assembly of
 * multiclass_nms_rotated (utils/bbox_nms_rotated.py:47-64: dets = cat([bboxes, scores[:, None]]), labels, sorted by
 * score, cut to max_per_img) for a whole batch — no host synchronisation, no stock tensor ops behind the NMS:
 *   wire[g][r*7 .. r*7+6] = x, y, w, h, angle, score, label (as float) of the r-th kept row of group g in
 *   descending-score order; rows behind the last kept one are 0,0,0,0,0,0,-1; wire[g][max_per_group*7] = the number
 *   of rows written = min(K_g, max_per_group).  wire is float32 [num_groups][max_per_group*7 + 1]: also the
 *   all-gather wire format of the data-parallel detector (one buffer per rank, s2anet_amd/gather.py).
 *   labels_out int32 [num_groups][max_per_group] (-1 padded) and counts_out int32 [num_groups]: optional copies.
 * row_labels[n] int32 = class of every row (out_cls of s2a_multiclass_candidates).
 * cand_found (device int64, may be NULL) = the untruncated candidate count of s2a_multiclass_candidates whose first
 * n rows these are.  Rows at or behind min(*cand_found, n) are PADDING by contract: they are never kept and (on every
 * internal path) never take part in a comparison, whatever their segment id -- a caller that passes a smaller count
 * than it filled rows cuts its own input.  overflow_out (device int64[2], may be NULL) receives [found, found - n clamped at 0] and
 * *dropped_total (device int64, may be NULL) is incremented by the second number (the reference never drops a
 * candidate, so a caller with a static row cap must be able to prove that nothing was cut). */
int s2a_nms_rotated_segmented_dets(const float* dets, const float* scores, const int32_t* segment_ids,
                                   const int32_t* group_ids, const int32_t* row_labels, int64_t n,
                                   int32_t num_segments, int32_t num_groups, float iou_threshold,
                                   int32_t max_per_group, float* wire, int32_t* labels_out,
                                   int32_t* counts_out, const int64_t* cand_found, int64_t* overflow_out,
                                   int64_t* dropped_total, void* workspace, size_t workspace_bytes,
                                   s2a_stream_t stream);

/* Candidate selection of multiclass_nms_rotated (utils/bbox_nms_rotated.py:29-42) for a whole
 * batch: every (box, class) pair with score > score_thr, in row-major (image, box, class) order,
 * compacted into `cap` slots (first `cap` in that order if there are more; *count_dev holds the
 * untruncated count).  boxes[batch*n,5], scores[batch*n*num_classes] f32.  Unused slots become
 * padding rows: out_seg = out_grp = out_cls = -1, score -1 (ignored by
 * s2a_nms_rotated_segmented).  out_seg = image*num_classes + class, out_grp = image. */
size_t s2a_multiclass_candidates_workspace_bytes(int64_t total_scores);
int s2a_multiclass_candidates(const float* boxes, const float* scores, int64_t batch, int64_t n,
                              int64_t num_classes, float score_thr, int64_t cap, float* out_boxes,
                              float* out_scores, int32_t* out_seg, int32_t* out_grp, int32_t* out_cls,
                              int64_t* count_dev, void* workspace, size_t workspace_bytes,
                              s2a_stream_t stream);

/* ---------------------------------------------------------------------------
 * ORN.  Replaces  models.orn.orn_cuda.arf_forward(weight[O,I,nOri,kH,kW], indices u8
 * [nOri,kH,kW,nRot]) -> [O*nRot, I*nOri, kH, kW]
 * (models/orn/src/vision.cpp:7-12, cuda/ActiveRotatingFilter_cuda.cu:20-46,79-119).
 * dtype F32 or F16 (element size only matters: the op is a permuting copy).
 * ------------------------------------------------------------------------- */
int s2a_arf_forward(const void* weight, const uint8_t* indices, int64_t n_out, int64_t n_in,
                    int n_orientation, int kh, int kw, int n_rotation, int dtype, void* output,
                    s2a_stream_t stream);

/* orn_cuda.arf_backward(indices, gradOutput[O*nRot, I*nOri, kH, kW]) -> gradInput[O,I,nOri,kH,kW]
 * (models/orn/src/vision.cpp:9, cuda/ActiveRotatingFilter_cuda.cu:49-76,122-162): sum over the nRot
 * rotated copies, ascending k.  float32 (the reference dispatches float/double only). */
int s2a_arf_backward(const uint8_t* indices, const void* grad_output, int64_t n_out, int64_t n_in,
                     int n_orientation, int kh, int kw, int n_rotation, int dtype, void* grad_input,
                     s2a_stream_t stream);

/* orn_cuda.rie_forward(feature[B,C,1,1], nOrientation) -> (mainDirection uint8[B,C/nOri], aligned[B,C,1,1]) and
 * orn_cuda.rie_backward(mainDirection, gradOutput[B,C,1,1], nOrientation) -> gradInput (models/orn/src/vision.cpp:10-11,
 * RotationInvariantEncoding.h:11-34, cuda/RotationInvariantEncoding_cuda.cu:20-84): main direction = first index of
 * the strict maximum of each group of nOri values, group rotated so that it comes first; backward rotates back.
 * float32.  Not used by the S2ANet model itself (exported by the module the model imports). */
int s2a_rie_forward(const void* feature, int64_t batch, int64_t channels, int n_orientation, int dtype,
                    uint8_t* main_direction, void* aligned, s2a_stream_t stream);
int s2a_rie_backward(const uint8_t* main_direction, const void* grad_output, int64_t batch, int64_t features,
                     int n_orientation, int dtype, void* grad_input, s2a_stream_t stream);

/* RotationInvariantPooling.forward (models/orn/functions/rotation_invariant_pooling.py:19-27):
 * x[B,C,H,W] -> out[B,C/nOri,H,W] = max over each group of nOri consecutive channels. */
int s2a_rot_inv_pool(const void* x, int64_t batch, int64_t channels, int64_t hw, int n_orientation,
                     int dtype, int layout, void* out, s2a_stream_t stream);

/* ---------------------------------------------------------------------------
 * Deformable convolution v1 forward.  Replaces
 *   models.dcn.deform_conv_cuda.deform_conv_forward_cuda(input, weight, offset, output,
 *       columns, ones, kW, kH, dW, dH, padW, padH, dilationW, dilationH, group,
 *       deformable_group, im2col_step) -> int
 * (models/dcn/src/deform_conv_cuda.cpp:152-260; sampling kernel
 *  deform_conv_cuda_kernel.cu:83-114,189-242).  The reference's `columns` / `ones`
 * scratch tensors and im2col_step have no counterpart: sampling and the channel
 * contraction are fused (no columns buffer).  Argument order keeps the reference's
 * W-before-H convention.
 *   input  [B,C,H,W]                 dtype, `layout` storage
 *   weight [O, C/group, kH, kW]      dtype, contiguous
 *   offset [B, dg*2*kH*kW, Ho, Wo]   F32 (or dtype when offset_dtype says so), NCHW;
 *                                    channel 2t = dy, 2t+1 = dx of tap t (kernel.cu:221-222)
 *   output [B,O,Ho,Wo]               dtype, `layout` storage, caller-allocated
 * relu != 0 fuses AlignConv's ReLU (models/alignconv.py:97).
 * ------------------------------------------------------------------------- */
typedef struct s2a_dcn_params {
  int64_t batch, channels, height, width, out_channels;
  int kW, kH, dW, dH, padW, padH, dilationW, dilationH, group, deformable_group;
  int dtype;        /* S2A_DTYPE_* of input/weight/output */
  int offset_dtype; /* S2A_DTYPE_* of the offset tensor */
  int layout;       /* S2A_LAYOUT_* of input and output */
  int relu;
} s2a_dcn_params;

size_t s2a_deform_conv_workspace_bytes(const s2a_dcn_params* p);
int s2a_deform_conv_forward(const void* input, const void* weight, const void* offset,
                            void* output, const s2a_dcn_params* p, void* workspace,
                            size_t workspace_bytes, s2a_stream_t stream);

/* deform_conv_cuda.modulated_deform_conv_cuda_forward (models/dcn/src/deform_conv_cuda.cpp:491-570, kernel
 * deform_conv_cuda_kernel.cu:467-632; DCNv2): as s2a_deform_conv_forward with every sampled value multiplied by
 * mask[B, dg*kH*kW, Ho, Wo] and bias[O] (may be NULL) added after the contraction.  NCHW, any kernel size / stride /
 * padding / dilation / groups; offset and mask in the input's dtype.  One thread per output element (the S2ANet
 * model does not use this operator; it is exported by the module the model imports).  The backward is not built. */
int s2a_modulated_deform_conv_forward(const void* input, const void* weight, const void* bias, const void* offset,
                                      const void* mask, void* output, const s2a_dcn_params* p, s2a_stream_t stream);

/* AlignConv.get_offset (models/alignconv.py:30-87), batched: anchors[B,H*W,5] f32 (pixels,
 * radians) -> offset[B,2*k*k,H,W] f32 (k = 3). */
int s2a_align_offsets(const float* anchors, int64_t batch, int64_t height, int64_t width,
                      float stride, int ksize, float* offset, s2a_stream_t stream);

/* AlignConv.forward (models/alignconv.py:88-98) fully fused: refined anchors ->
 * sampling offsets (never materialised) -> bilinear sampling -> 3x3 contraction (MFMA)
 * -> ReLU.  x[B,C,H,W], anchors[B,H,W,5] f32, weight[O,C,3,3], out[B,O,H,W]. */
typedef struct s2a_align_params {
  int64_t batch, channels, height, width, out_channels;
  float stride;
  int dtype, layout, relu;
  int weight_packed; /* 1: `weight` is the output of s2a_dcn_pack_weight (cached by the caller at
                        inference, the weights do not change between forwards) */
} s2a_align_params;
size_t s2a_align_conv_workspace_bytes(const s2a_align_params* p);
/* weight[O,C,3,3] -> the layouts the fused kernels stream: stage-major [C/KC][9][O][KC]
 * (KC = 32 for f32, 64 for f16) and right behind it, for f16, a second copy in MFMA-fragment order; for f32, the filter split
 * into three bf16 planes, [C/16][9][plane][O][16] (hi + mid + lo = the f32 value exactly: the f32 forward runs on the 16-bit
 * matrix instruction, six plane products per f32 product -- the accuracy of f32 arithmetic for finite operands; an infinite
 * operand gives NaN where f32 arithmetic gives +-inf; S2A_DCN_F32=mfma32 selects the f32 instruction).
 * `packed` must hold s2a_dcn_packed_elems(O, C, dtype) elements (f16: 2 per weight, f32: 2.5 per weight). */
int64_t s2a_dcn_packed_elems(int64_t out_channels, int64_t channels, int dtype);
int s2a_dcn_pack_weight(const void* weight, int64_t out_channels, int64_t channels, int dtype,
                        void* packed, s2a_stream_t stream);
int s2a_align_conv_forward(const void* x, const float* anchors, const void* weight, void* out,
                           const s2a_align_params* p, void* workspace, size_t workspace_bytes,
                           s2a_stream_t stream);

/* ---------------------------------------------------------------------------
 * Head glue, batched and device-side.
 * s2a_delta2bbox_rotated: models/boxes.py:82-162 (is_encode_relative=True) + norm_angle
 *   (utils/general.py:925-929).  rois[n,5], deltas[n,5] f32 -> out[n,5].
 * s2a_fam_refine_anchors: gen_grid_anchors (models/anchors.py:75-126, 1 square anchor of
 *   side scale*stride per position) + fam_bbox_decode (models/head.py:27-52,
 *   wh_ratio_clip = 1e-6) straight from the NCHW/NHWC prediction map:
 *   bbox_pred[B,5,H,W] (dtype/layout) -> refined[B,H,W,5] f32.
 * ------------------------------------------------------------------------- */
/* level table of a pyramid-packed buffer (layout: see "Pyramid-packed head launches" below) */
typedef struct s2a_pyramid {
  int32_t n_levels;     /* 1..8 */
  int32_t height[8];
  int32_t width[8];
  float stride[8];      /* FPN stride of the level (AlignConv / anchor generation) */
} s2a_pyramid;

/* Candidate selection of get_bboxes for the whole batch on pyramid-packed predictions (models/head.py:684-717):
 * per level and image sigmoid -> max over classes -> top-k (max_per_level = 2000) only where H*W > k, levels
 * concatenated, final decode (wh_ratio_clip 16/1000).  cls[P,64] / reg[P,64] f16 (first num_classes / 5 columns),
 * anchors[P,5] f32 -> bboxes[B,n,5], scores[B,n,num_classes] f32 with n = s2a_pyramid_candidates_count; sel[B,n] int32
 * scratch receives the packed row of every candidate.  Ties at the k-th score: lowest positions first (the stock
 * top-k leaves them unspecified); candidates of a level come in position order.  Levels above 24576 positions that
 * need a top-k are refused (use the per-level path). */
int64_t s2a_pyramid_candidates_count(const s2a_pyramid* pyr, int64_t max_per_level);
int s2a_pyramid_candidates(const void* cls, const void* reg, const float* anchors, int64_t batch,
                           const s2a_pyramid* pyr, int num_classes, int64_t max_per_level, float wh_ratio_clip,
                           float* bboxes, float* scores, int32_t* sel, s2a_stream_t stream);

/* Output side (val.py:40-52): rotated_box_to_poly_single (utils/general.py:886-921) + cv2.boxPoints for a
 * whole batch: boxes rows (x,y,w,h,angle[,score...]) with row_stride floats -> polys[n,8] f32. */
int s2a_rbox_to_poly(const float* boxes, int64_t n, int64_t row_stride, float* polys, s2a_stream_t stream);
int s2a_delta2bbox_rotated(const float* rois, const float* deltas, int64_t n, float wh_ratio_clip,
                           float* out, s2a_stream_t stream);
int s2a_fam_refine_anchors(const void* bbox_pred, int64_t batch, int64_t height, int64_t width,
                           float stride, float anchor_scale, int dtype, int layout, float* refined,
                           s2a_stream_t stream);

/* Convolution epilogue for the conv layers of the head/carrier that MIOpen runs without fusion:
 * y[positions, channels] (channels-last storage) = act(y + bias[c] (+ residual)), in place.
 * Replaces the bias add / residual add / ReLU passes that follow every nn.Conv2d of
 * models/head.py:163-222 and models/backbone.py:37-83 (same arithmetic, one pass over HBM). */
int s2a_bias_act_nhwc(void* y, const void* bias, const void* residual, int64_t positions,
                      int64_t channels, int dtype, int relu, s2a_stream_t stream);
/* the same pass writing out[positions, channels] instead (out may equal y): lets the two library convolutions that
 * remain (FPN's stride-2 extra levels, models/neck.py:86-94) land in the pyramid-packed head buffer without a copy */
int s2a_bias_act_nhwc_to(const void* y, const void* bias, const void* residual, void* out, int64_t positions,
                         int64_t channels, int dtype, int relu, s2a_stream_t stream);

/* Regular convolutions, f16 channels-last, with bias / residual / ReLU fused — the conv towers of
 * S2ANetHead (models/head.py:163-222, nn.Conv2d + nn.ReLU pairs) and the 1x1 layers of the carrier,
 * on the same patch-staged MFMA structure as AlignConv.  ksize 3: pad 1, stride 1 or 2 (stride 2: O % 128 == 0).
 * ksize 1: pad 0, stride 1 or 2.  x[B,H,W,C] -> out[B,Ho,Wo,O] = relu?(conv(x) + bias (+ residual[B,Ho,Wo,O])).
 * weight_frag = s2a_conv_pack_weight_f16 of the [O,C,k,k] filter (O*C*k*k halfs, MFMA-fragment order);
 * bias[O] f16 or NULL; residual or NULL.  O must be a multiple of 64 (narrower heads: zero-pad the
 * filter), C a multiple of 64 or exactly 32 (then the filter given to s2a_conv_pack_weight_f16 is
 * zero-padded to 64 input channels). */
int s2a_conv_pack_weight_f16(const void* weight, int64_t out_channels, int64_t channels, int ksize,
                             void* packed, s2a_stream_t stream);
int s2a_conv_nhwc_f16(const void* x, const void* weight_frag, const void* bias, const void* residual,
                      void* out, int64_t batch, int64_t channels, int64_t height, int64_t width,
                      int64_t out_channels, int ksize, int stride, int relu, s2a_stream_t stream);

/* Training side of the deformable convolution: the three device functions the reference's
 * backward is composed of (models/dcn/src/deform_conv_cuda_kernel.cu): deformable_im2col (:244-276),
 * deformable_col2im (:332-370), deformable_col2im_coord (:431-464).  p->batch = the number of images of
 * the chunk (the reference's im2col_step / parallel_imgs); layouts as the reference: im [S,C,H,W],
 * offset [S, dg*2*kH*kW, Ho, Wo] (same dtype as im), columns [C*kH*kW, S*Ho*Wo]; NCHW only.
 * s2a_deformable_col2im ACCUMULATES into grad_im_acc [S,C,H,W] (float32 for float32 / float16 columns, float64 for
 * float64 columns; caller zeroes it, as deform_conv.py:88-90 does); s2a_deformable_col2im_coord overwrites grad_offset.
 * The GEMMs around them (deform_conv_cuda.cpp:323-324, :455-459 addmm_) are library GEMMs on the host side.
 * dtype S2A_DTYPE_F64 (the reference dispatches these kernels on double too: AT_DISPATCH_FLOATING_TYPES_AND_HALF,
 * deform_conv_cuda_kernel.cu:258,352,450): evaluated in double throughout -- s2a_deform_conv_forward takes it on its
 * generic NCHW kernel (offsets float64 as well), and torch.autograd.gradcheck ties backward to forward on it. */
int s2a_deformable_im2col(const void* im, const void* offset, void* columns, const s2a_dcn_params* p,
                          s2a_stream_t stream);
int s2a_deformable_col2im(const void* columns, const void* offset, void* grad_im_acc,
                          const s2a_dcn_params* p, s2a_stream_t stream);
int s2a_deformable_col2im_coord(const void* columns, const void* im, const void* offset, void* grad_offset,
                                const s2a_dcn_params* p, s2a_stream_t stream);

/* deform_conv_backward_input_cuda (models/dcn/src/deform_conv_cuda.cpp:262-374: gradInput and gradOffset) for f16 tensors
 * with the AlignConv geometry -- 3x3, stride 1, pad 1, dilation 1, one group, one deformable group, channels % 32 == 0,
 * out_channels % 16 == 0, out_channels <= 256 -- as ONE fused kernel per call: the column gradient W^T . gradOutput is
 * formed tile by tile on the matrix cores and consumed in LDS (offset gradient = contraction with the sampled corners,
 * input gradient = bilinear scatter into an LDS window, flushed with one atomic per touched cell); the reference's
 * `columns` [C*9, N] never exists.  All tensors NCHW f16 as the reference passes them; grad_input_f32 [S,C,H,W] is
 * ACCUMULATED (the caller zeroes it, deform_conv.py:88), grad_offset [S,18,H,W] f16 is overwritten. */
size_t s2a_deform_conv_backward_input_workspace_bytes(int64_t batch, int64_t channels, int64_t height, int64_t width,
                                                      int64_t out_channels);
int s2a_deform_conv_backward_input_f16(const void* input, const void* offset, const void* grad_output, const void* weight,
                                       float* grad_input_f32, void* grad_offset, int64_t batch, int64_t channels,
                                       int64_t height, int64_t width, int64_t out_channels, void* workspace,
                                       size_t workspace_bytes, s2a_stream_t stream);

/* The same entry for f32 tensors (same geometry limits), on the f32 matrix instruction: 4 x 8 position tiles, gradOutput
 * tile and column gradient in LDS, no `columns`.  grad_input [S,C,H,W] f32 is the caller's gradInput, ACCUMULATED in place
 * (deform_conv_cuda.cpp:334-340 adds into it through col2im's atomics); grad_offset [S,18,H,W] f32 is overwritten. */
size_t s2a_deform_conv_backward_input_f32_workspace_bytes(int64_t batch, int64_t channels, int64_t height, int64_t width,
                                                          int64_t out_channels);
int s2a_deform_conv_backward_input_f32(const float* input, const float* offset, const float* grad_output, const float* weight,
                                       float* grad_input, float* grad_offset, int64_t batch, int64_t channels,
                                       int64_t height, int64_t width, int64_t out_channels, void* workspace,
                                       size_t workspace_bytes, s2a_stream_t stream);

/* deform_conv_backward_parameters_cuda (models/dcn/src/deform_conv_cuda.cpp:376-489: gradWeight) for f16 tensors with the
 * AlignConv geometry (channels % 64 == 0, out_channels % 32 == 0, out_channels <= 256), fused: the sampled columns are
 * formed tile by tile in LDS (bilinear corners from an LDS patch, as the forward does) and contracted with gradOutput over
 * the POSITIONS on the matrix cores, both operands through gfx950's transposing LDS read; split-K over the position tiles:
 * every workgroup stores its partial block and one reduce kernel sums the slices in a fixed order into grad_weight_f32
 * [O,C,3,3] (ACCUMULATED, unscaled: the caller zeroes it and applies `scale`) -- no atomics, bit-identical from run to run. */
size_t s2a_deform_conv_backward_weight_workspace_bytes(int64_t batch, int64_t channels, int64_t height, int64_t width,
                                                       int64_t out_channels);
int s2a_deform_conv_backward_weight_f16(const void* input, const void* offset, const void* grad_output,
                                        float* grad_weight_f32, int64_t batch, int64_t channels, int64_t height,
                                        int64_t width, int64_t out_channels, void* workspace, size_t workspace_bytes,
                                        s2a_stream_t stream);

/* The same entry for f32 tensors (same geometry limits, 4 x 8 position tiles): grad_weight [O,C,3,3] f32 is the caller's
 * gradWeight, += scale * gradOutput x columns^T in place (deform_conv_cuda.cpp:455-459); deterministic as above.
 * Arithmetic: every f32 operand is split EXACTLY into three bf16 values (8 + 8 + 8 significand bits) and a product is the six
 * largest of the nine plane products, accumulated in f32 on the 16-bit matrix instruction -- the accuracy of f32 arithmetic
 * (|error| of a product <~ 3 * 2^-24 of it; the bilinear blend itself runs in f32), 2.7 x fewer matrix cycles than
 * v_mfma_f32_16x16x4_f32.  The environment switch S2A_BWD_F32_WEIGHT=mfma32 selects the kernel on the f32 instruction. */
size_t s2a_deform_conv_backward_weight_f32_workspace_bytes(int64_t batch, int64_t channels, int64_t height, int64_t width,
                                                           int64_t out_channels);
int s2a_deform_conv_backward_weight_f32(const float* input, const float* offset, const float* grad_output,
                                        float* grad_weight, float scale, int64_t batch, int64_t channels, int64_t height,
                                        int64_t width, int64_t out_channels, void* workspace, size_t workspace_bytes,
                                        s2a_stream_t stream);

/* Both gradients of the AlignConv-geometry deformable convolution in ONE call -- what DeformConvFunction.backward needs
 * (models/dcn/deform_conv.py:73-118 calls deform_conv_backward_input_cuda and deform_conv_backward_parameters_cuda on the same
 * tensors): the NHWC copies of input and gradOutput the fused kernels read are made once instead of twice.
 * dtype = S2A_DTYPE_F16 / S2A_DTYPE_F32: the type of input, offset, grad_output, weight [O,C,3,3] and grad_offset [S,18,H,W];
 * grad_input_f32 [S,C,H,W] (ACCUMULATED) and grad_weight_f32 [O,C,3,3] (+= scale * ...) are always f32.  Either may be NULL
 * to skip that gradient (grad_offset goes with grad_input).  Limits: channels % 64 == 0, out_channels % 32 == 0, <= 256. */
size_t s2a_deform_conv_backward_workspace_bytes(int dtype, int64_t batch, int64_t channels, int64_t height, int64_t width,
                                                int64_t out_channels);
int s2a_deform_conv_backward(int dtype, const void* input, const void* offset, const void* grad_output, const void* weight,
                             float* grad_input_f32, void* grad_offset, float* grad_weight_f32, float scale, int64_t batch,
                             int64_t channels, int64_t height, int64_t width, int64_t out_channels, void* workspace,
                             size_t workspace_bytes, s2a_stream_t stream);
/* The same call with gradInput as DeformConvFunction.backward owns it (deform_conv.py:88: zeros_like(input)): grad_input
 * [S,C,H,W] is a tensor of `dtype` and is OVERWRITTEN with the gradient (== the reference's accumulation into a zeroed
 * tensor) -- no f32 copy on the caller's side and no conversion pass behind the call.  The tiles' sums are accumulated in f32
 * (workspace, [S,H,W,C]: one pixel's 32 channels are one 128-byte atomic row) and rounded once.  Workspace:
 * s2a_deform_conv_backward_workspace_bytes. */
int s2a_deform_conv_backward_typed(int dtype, const void* input, const void* offset, const void* grad_output,
                                   const void* weight, void* grad_input, void* grad_offset, float* grad_weight_f32,
                                   float scale, int64_t batch, int64_t channels, int64_t height, int64_t width,
                                   int64_t out_channels, void* workspace, size_t workspace_bytes, s2a_stream_t stream);

/* A bottleneck's conv2 + conv3 in one launch (models/backbone.py:56-83 with the BatchNorms folded):
 *   out = relu(W3 . relu(conv3x3(x; W2) + b2) + b3 + residual)        x [B,H,W,64] -> out [B,H,W,256], f16 NHWC
 * The 64-map intermediate never leaves the workgroup (LDS); results are bit-identical to s2a_conv_nhwc_f16 (3x3,
 * ReLU) followed by s2a_conv_nhwc_f16 (1x1, residual, ReLU).  weight_frag / tail_weight_frag from
 * s2a_conv_pack_weight_f16 (ksize 3 / 1); residual may be NULL.  channels = mid_channels = 64, out_channels = 256.
 * chain_*: optionally the NEXT bottleneck's conv1 (1x1, 256 -> chain_channels = 64 or 128, + bias + ReLU) applied to
 * the finished output tile in the same launch: chain_out [B,H,W,chain_channels] = relu(Wc . out + bc), bit-identical
 * to a separate s2a_conv_nhwc_f16; all three pointers NULL = not computed. */
int s2a_conv3x3_tail1x1_f16(const void* x, const void* weight_frag, const void* bias, const void* tail_weight_frag,
                            const void* tail_bias, const void* residual, void* out, const void* chain_weight_frag,
                            const void* chain_bias, void* chain_out, int64_t chain_channels, int64_t batch, int64_t channels,
                            int64_t mid_channels, int64_t out_channels, int64_t height, int64_t width,
                            s2a_stream_t stream);

/* FPN top-down step in one launch (models/neck.py:67-79): out[B,H,W,O] = conv1x1(x[B,H,W,C]) + bias +
 * nearest-2x-upsample(coarse[B,H/2,W/2,O]); f16 channels-last, H and W even, C and O multiples of 64. */
int s2a_conv1x1_add_up2_f16(const void* x, const void* weight_frag, const void* bias, const void* coarse,
                            void* out, int64_t batch, int64_t channels, int64_t height, int64_t width,
                            int64_t out_channels, s2a_stream_t stream);

/* Pyramid-packed head launches.  The conv towers, AlignConv and prediction heads of S2ANetHead share
 * their filters over the FPN levels (models/head.py:261-265 maps forward_single over the levels), so
 * one launch can serve all of them: the levels sit back to back in ONE channels-last buffer
 * [sum_l B*H_l*W_l, C], level l = [B,H_l,W_l,C] at pixel offset sum_{k<l} B*H_k*W_k (anchors
 * [.,5] f32 and outputs packed the same way).  A workgroup finds its level from its tile index.
 *   s2a_pyramid_pixels            -> total pixel rows of the packed buffer (or -1: bad table)
 *   s2a_conv3x3_pyramid_f16       = s2a_conv_nhwc_f16(ksize 3) on every level
 *   s2a_align_conv_pyramid_f16    = s2a_align_conv_forward (f16, NHWC, weight from s2a_dcn_pack_weight)
 *                                   on every level, stride[l] = the level's anchor stride
 *   s2a_fam_refine_anchors_pyramid = s2a_fam_refine_anchors on every level; pred rows have
 *                                   row_stride f16 columns of which the first 5 are the deltas */
int64_t s2a_pyramid_pixels(const s2a_pyramid* pyr, int64_t batch);
int s2a_conv3x3_pyramid_f16(const void* x, const void* weight_frag, const void* bias, const void* residual,
                            void* out, int64_t batch, int64_t channels, int64_t out_channels, int relu,
                            const s2a_pyramid* pyr, s2a_stream_t stream);
/* A conv tower's last 3x3 layer + the 1x1 prediction head that reads it (fam_reg_ls / fam_cls_ls -> fam_reg_head /
 * fam_cls_head, models/head.py:163-213, :296-306) in one launch: head_out[P,64] (columns 0..31 written; the head's
 * <= 32 maps first) = head(relu?(conv3x3(x) + bias)) + head_bias, computed from the staged f16 tile as the separate
 * layers would.  out may be NULL when nothing else reads the tower (then it is never written).  O must be 256;
 * head_weight_frag = s2a_conv_pack_weight_f16 of the zero-padded [64,256,1,1] filter, head_bias >= 32 f16. */
int s2a_conv3x3_head_pyramid_f16(const void* x, const void* weight_frag, const void* bias, void* out,
                                 const void* head_weight_frag, const void* head_bias, void* head_out,
                                 int64_t batch, int64_t channels, int64_t out_channels, int relu,
                                 const s2a_pyramid* pyr, s2a_stream_t stream);
/* ORConv2d + RotationInvariantPooling in one launch (models/head.py:337-341): out[P,O] = conv3x3(x) + bias (no
 * activation) and pooled[P,O/8] = max over every run of 8 orientation channels of out, both pyramid-packed. */
int s2a_orconv_pool_pyramid_f16(const void* x, const void* weight_frag, const void* bias, void* out, void* pooled,
                                int64_t batch, int64_t channels, int64_t out_channels, const s2a_pyramid* pyr,
                                s2a_stream_t stream);
/* The same 3x3 / stride 1 / pad 1 convolution (bias, optional ReLU, optional orientation max-pool as
 * s2a_orconv_pool_pyramid_f16) in the Winograd F(2,3) minimal-filtering form along x: 6 C O multiply-adds per output
 * instead of 9 C O, f16 operands / f32 accumulation as the direct kernel (the reference leaves the algorithm to cuDNN,
 * models/head.py:163-222).  Results differ from s2a_conv3x3_pyramid_f16 by the rounding of the transformed operands
 * (one f16 rounding of d_i +- d_j and of the transformed filter): same tolerance against an f32 convolution.
 *   weight_wino = s2a_conv_wino_pack_weight_f16 of the [O,C,3,3] f16 filter (s2a_conv_wino_packed_elems halfs; O a
 *   multiple of 64, C a multiple of 32); pooled [P,O/8] or NULL; a one-level table serves a plain [B,H,W,C] tensor. */
int64_t s2a_conv_wino_packed_elems(int64_t out_channels, int64_t channels);
int s2a_conv_wino_pack_weight_f16(const void* weight, int64_t out_channels, int64_t channels, void* packed,
                                  s2a_stream_t stream);
int s2a_conv3x3_wino_pyramid_f16(const void* x, const void* weight_wino, const void* bias, void* out, void* pooled,
                                 int64_t batch, int64_t channels, int64_t out_channels, int relu,
                                 const s2a_pyramid* pyr, s2a_stream_t stream);
int s2a_align_conv_pyramid_f16(const void* x, const float* anchors, const void* weight_packed, void* out,
                               int64_t batch, int64_t channels, int64_t out_channels, int relu,
                               const s2a_pyramid* pyr, s2a_stream_t stream);
int s2a_fam_refine_anchors_pyramid(const void* pred, int64_t row_stride, int64_t batch, const s2a_pyramid* pyr,
                                   float anchor_scale, float* refined, s2a_stream_t stream);

/* Fused ResNet stem of the end-to-end config: uint8 image / divisor
 * (val.py:246-247) -> conv 7x7 / stride 2 / pad 3, 3 -> 64 maps + bias (BatchNorm folded) -> ReLU ->
 * max-pool 3x3 / stride 2 / pad 1 (models/backbone.py:112-117, :172-175) in one kernel.
 * image_u8[B,H,W,3] (channels-last uint8, W % 4 == 0) -> out[B,Hp,Wp,64] f16 with
 * Hc = (H-1)/2+1, Hp = (Hc-1)/2+1 (same for W).  weight_packed = s2a_stem_pack_weight_f16 of the
 * [64,3,7,7] f16 filter (s2a_stem_packed_elems() halfs); bias[64] f16 or NULL. */
int64_t s2a_stem_packed_elems(void);
int s2a_stem_pack_weight_f16(const void* weight, void* packed, s2a_stream_t stream);
int s2a_stem_u8_f16(const void* image_u8, const void* weight_packed, const void* bias, void* out,
                    int64_t batch, int64_t height, int64_t width, float divisor, s2a_stream_t stream);

/* 0 for a normal build; non-zero when an object was compiled with a measurement / ablation switch (-DS2A_MEASURE,
 * -DS2A_ABL=..., -DS2A_STAMP=1: such a build may skip work or print diagnostics). */
int s2a_build_flags(void);

/* Diagnostic builds only (-DS2A_STAMP=1): per-workgroup s_memtime phase stamps of the AlignConv
 * kernel; returns S2A_ENOTIMPL in a normal build. */
int s2a_debug_read_stamps(unsigned long long* host_dst, int64_t count);

#ifdef __cplusplus
}
#endif
#endif /* S2ANET_HIP_H_ */
