// TEST INFRASTRUCTURE ONLY (oracle/_ref recipe).  Host-compiles the reference's
// rotated-IoU geometry header with its __CUDACC__ (GPU swap-sort) branch
// selected, so that the oracle's restatement of the GPU branch can be pinned
// against the reference's own text.  The header is read in place from
// /root/reference (include path set by oracle/build_ref.py); nothing of it is
// copied here.  Reference: utils/ml_nms_rotated/src/box_iou_rotated_utils.h:209-226
// (swap sort), :317-322 (label test); utils/box_iou_rotated/src/box_iou_rotated_utils.h.
#include <cmath>
#include <cstdint>
#include <cassert>
#define __CUDACC__ 1
#define __host__
#define __device__
#define __forceinline__ inline
namespace ref_ml {
#include "box_iou_rotated_utils.h"  // resolved by -I to the reference tree (ml_nms_rotated flavour)
}
#undef __CUDACC__

extern "C" {
// 6 floats per box (x,y,w,h,a,label), ml-NMS flavour of the header
float ref_gpubranch_iou6(const float* b1, const float* b2) {
  return ref_ml::single_box_iou_rotated<float>(b1, b2);
}
void ref_gpubranch_iou6_pairs(const float* b1, const float* b2, int64_t n, float* out) {
  for (int64_t i = 0; i < n; i++) out[i] = ref_ml::single_box_iou_rotated<float>(b1 + 6 * i, b2 + 6 * i);
}
}
