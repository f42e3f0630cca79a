// Head glue of S2ANetHead for gfx950, batched and device-side:
//   delta2bbox_rotated  models/boxes.py:82-162  (+ norm_angle utils/general.py:925-929)
//   grid anchors        models/anchors.py:75-126 (one square anchor / position, angle 0)
//   fam_bbox_decode     models/head.py:27-52 (wh_ratio_clip = 1e-6), no per-image Python loop
//   AlignConv.get_offset models/alignconv.py:30-87, no per-image Python loop
// All elementwise, HBM-bound; float32 arithmetic in the reference's operation order.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>

#include "common.hpp"

namespace s2a {
namespace {

constexpr float kPi = 3.14159265358979323846f;

__device__ __forceinline__ float norm_angle(float a) {
  // (a - (-pi/4)) mod pi + (-pi/4), python floor-mod
  const float lo = -0.78539816339744830962f;
  float r = fmodf(a - lo, kPi);
  if (r != 0.0f && r < 0.0f) r += kPi;
  return r + lo;
}

__device__ __forceinline__ void decode_one(const float* roi, const float* d, float max_ratio,
                                           float* o) {
  float dw = fminf(fmaxf(d[2], -max_ratio), max_ratio);
  float dh = fminf(fmaxf(d[3], -max_ratio), max_ratio);
  float ca = cosf(roi[4]), sa = sinf(roi[4]);
  o[0] = d[0] * roi[2] * ca - d[1] * roi[3] * sa + roi[0];
  o[1] = d[0] * roi[2] * sa + d[1] * roi[3] * ca + roi[1];
  o[2] = roi[2] * expf(dw);
  o[3] = roi[3] * expf(dh);
  o[4] = norm_angle(kPi * d[4] + roi[4]);
}

__global__ void k_delta2bbox(const float* __restrict__ rois, const float* __restrict__ deltas,
                             int64_t n, float max_ratio, float* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float r[5], d[5], o[5];
#pragma unroll
  for (int k = 0; k < 5; k++) {
    r[k] = rois[5 * i + k];
    d[k] = deltas[5 * i + k];
  }
  decode_one(r, d, max_ratio, o);
#pragma unroll
  for (int k = 0; k < 5; k++) out[5 * i + k] = o[k];
}

template <typename T, bool NHWC>
__global__ void k_fam_refine(const T* __restrict__ pred, int64_t B, int64_t H, int64_t W,
                             float stride, float side, float max_ratio,
                             float* __restrict__ refined) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t HW = H * W;
  if (e >= B * HW) return;
  int64_t b = e / HW, p = e % HW;
  int64_t y = p / W, x = p % W;
  float d[5];
#pragma unroll
  for (int k = 0; k < 5; k++)
    d[k] = NHWC ? (float)pred[(b * HW + p) * 5 + k] : (float)pred[(b * 5 + k) * HW + p];
  float half = 0.5f * (stride - 1.0f);
  float roi[5] = {(float)x * stride + half, (float)y * stride + half, side, side, 0.0f};
  float o[5];
  decode_one(roi, d, max_ratio, o);
#pragma unroll
  for (int k = 0; k < 5; k++) refined[e * 5 + k] = o[k];
}

// pyramid-packed variant: pred rows [sum_l B*H_l*W_l][row_stride] f16 (first 5 columns = deltas), all
// FPN levels in one launch; a thread finds its level from the packed pixel index
struct RefineLevels {
  int n, batch;
  int H[8], W[8], pix0[8];
  float stride[8];
};
__global__ void k_fam_refine_pyramid(const _Float16* __restrict__ pred, int row_stride, RefineLevels lv,
                                     int64_t total, float anchor_scale, float max_ratio,
                                     float* __restrict__ refined) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  int H = lv.H[0], W = lv.W[0], p0 = 0;
  float stride = lv.stride[0];
#pragma unroll
  for (int i = 1; i < 8; i++)
    if (i < lv.n && e >= lv.pix0[i]) {
      H = lv.H[i]; W = lv.W[i]; p0 = lv.pix0[i]; stride = lv.stride[i];
    }
  const int64_t p = (e - p0) % ((int64_t)H * W);
  const int64_t y = p / W, x = p % W;
  float d[5];
#pragma unroll
  for (int k = 0; k < 5; k++) d[k] = (float)pred[e * row_stride + k];
  const float half = 0.5f * (stride - 1.0f), side = anchor_scale * stride;
  float roi[5] = {(float)x * stride + half, (float)y * stride + half, side, side, 0.0f};
  float o[5];
  decode_one(roi, d, max_ratio, o);
#pragma unroll
  for (int k = 0; k < 5; k++) refined[e * 5 + k] = o[k];
}

// one thread per (b, position); writes the 18 offset planes (coalesced over positions)
__global__ void k_align_offsets(const float* __restrict__ anchors, int64_t B, int64_t H, int64_t W,
                                float stride, int ks, float* __restrict__ offset) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t HW = H * W;
  if (e >= B * HW) return;
  int64_t b = e / HW, p = e % HW;
  float yc = (float)(p / W), xc = (float)(p % W);
  const float* a = anchors + e * 5;
  float x_ctr = a[0] / stride, y_ctr = a[1] / stride, w = a[2] / stride, h = a[3] / stride;
  float cs = cosf(a[4]), sn = sinf(a[4]);
  float dw = w / (float)ks, dh = h / (float)ks;
  int pad = (ks - 1) / 2;
  float* ob = offset + b * (2 * ks * ks) * HW + p;
  for (int ky = 0; ky < ks; ky++)
    for (int kx = 0; kx < ks; kx++) {
      float xx = (float)(kx - pad), yy = (float)(ky - pad);
      float x = dw * xx, y = dh * yy;
      float xr = cs * x - sn * y;
      float yr = sn * x + cs * y;
      float off_x = (xr + x_ctr) - (xc + xx);
      float off_y = (yr + y_ctr) - (yc + yy);
      int t = ky * ks + kx;
      ob[(int64_t)(2 * t) * HW] = off_y;
      ob[(int64_t)(2 * t + 1) * HW] = off_x;
    }
}

// y = act(y + bias[c] (+ residual)) in place, channels-last (the channel is the fastest
// dimension): one pass instead of the three (bias add, residual add, ReLU) the stock
// elementwise kernels make after every convolution.  8 halfs / 4 floats per lane (16 B).
template <typename T, int VEC, bool HOIST>
__global__ __launch_bounds__(256) void k_bias_act(T* __restrict__ y, const T* __restrict__ bias,
                                                  const T* __restrict__ res, int64_t nvec, int cvec,
                                                  int relu) {
  using V = T __attribute__((ext_vector_type(VEC)));
  const int64_t T0 = (int64_t)gridDim.x * blockDim.x;
  const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  // HOIST: the launch makes the total thread count a multiple of cvec (cvec is a power of two
  // <= 256 for every layer of the network), so a thread always meets the same channel group
  V b0;
  if (HOIST) b0 = reinterpret_cast<const V*>(bias)[i0 % cvec];
  auto apply = [&](V v, V b, V r) {
#pragma unroll
    for (int e = 0; e < VEC; e++) {
      float f = (float)v[e] + (float)b[e];
      if (res) f += (float)r[e];
      if (relu) f = fmaxf(f, 0.f);
      v[e] = (T)f;
    }
    return v;
  };
  int64_t i = i0;
  // four independent 16-byte streams per lane in flight
  for (; i + 3 * T0 < nvec; i += 4 * T0) {
    V v[4], r[4], b[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      v[k] = reinterpret_cast<V*>(y)[i + k * T0];
      if (res) r[k] = reinterpret_cast<const V*>(res)[i + k * T0];
      b[k] = HOIST ? b0 : reinterpret_cast<const V*>(bias)[(i + k * T0) % cvec];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) reinterpret_cast<V*>(y)[i + k * T0] = apply(v[k], b[k], r[k]);
  }
  for (; i < nvec; i += T0) {
    V v = reinterpret_cast<V*>(y)[i], r;
    if (res) r = reinterpret_cast<const V*>(res)[i];
    V b = HOIST ? b0 : reinterpret_cast<const V*>(bias)[i % cvec];
    reinterpret_cast<V*>(y)[i] = apply(v, b, r);
  }
}

}  // namespace
}  // namespace s2a

using namespace s2a;

extern "C" int s2a_bias_act_nhwc(void* y, const void* bias, const void* residual, int64_t positions,
                                 int64_t channels, int dtype, int relu, s2a_stream_t stream) {
  S2A_CHECK_ARG(positions >= 0 && channels > 0, "bias_act: bad shape");
  S2A_CHECK_ARG(dtype == S2A_DTYPE_F32 || dtype == S2A_DTYPE_F16, "bias_act: dtype");
  const int vec = dtype == S2A_DTYPE_F16 ? 8 : 4;
  S2A_CHECK_ARG(channels % vec == 0, "bias_act: channels must be a multiple of %d", vec);
  if (positions == 0) return S2A_OK;
  S2A_CHECK_ARG(y && bias, "bias_act: NULL tensor");
  S2A_CHECK_ARG(((uintptr_t)y % 16) == 0 && ((uintptr_t)bias % 16) == 0 && ((uintptr_t)residual % 16) == 0,
                "bias_act: tensors must be 16-byte aligned");
  const int64_t nvec = positions * channels / vec;
  const int cvec = (int)(channels / vec);
  unsigned g = (unsigned)std::min<int64_t>((nvec + 1023) / 1024, 256 * 8);
  if (g == 0) g = 1;
  const bool hoist = cvec <= 256 && (256 % cvec) == 0;   // then gridDim*256 is a multiple of cvec
  hipStream_t st = as_stream(stream);
#define S2A_BA(T, V, H) k_bias_act<T, V, H><<<g, 256, 0, st>>>((T*)y, (const T*)bias, (const T*)residual, nvec, cvec, relu)
  if (dtype == S2A_DTYPE_F16) {
    if (hoist) S2A_BA(_Float16, 8, true); else S2A_BA(_Float16, 8, false);
  } else {
    if (hoist) S2A_BA(float, 4, true); else S2A_BA(float, 4, false);
  }
#undef S2A_BA
  S2A_LAUNCH_CHECK();
  return S2A_OK;
}

extern "C" int s2a_delta2bbox_rotated(const float* rois, const float* deltas, int64_t n,
                                      float wh_ratio_clip, float* out, s2a_stream_t stream) {
  S2A_CHECK_ARG(n >= 0 && wh_ratio_clip > 0, "delta2bbox_rotated: bad argument");
  if (n == 0) return S2A_OK;
  S2A_CHECK_ARG(rois && deltas && out, "delta2bbox_rotated: NULL tensor");
  float max_ratio = (float)std::fabs(std::log((double)wh_ratio_clip));
  k_delta2bbox<<<(unsigned)((n + 255) / 256), 256, 0, as_stream(stream)>>>(rois, deltas, n, max_ratio, out);
  S2A_LAUNCH_CHECK();
  return S2A_OK;
}

extern "C" int s2a_fam_refine_anchors(const void* bbox_pred, int64_t batch, int64_t height,
                                      int64_t width, float stride, float anchor_scale, int dtype,
                                      int layout, float* refined, s2a_stream_t stream) {
  S2A_CHECK_ARG(batch >= 0 && height >= 0 && width >= 0 && stride > 0, "fam_refine_anchors: bad shape");
  const int64_t total = batch * height * width;
  if (total == 0) return S2A_OK;
  S2A_CHECK_ARG(bbox_pred && refined, "fam_refine_anchors: NULL tensor");
  hipStream_t st = as_stream(stream);
  const float max_ratio = (float)std::fabs(std::log(1e-6));  // head.py:48
  const float side = anchor_scale * stride;
  unsigned g = (unsigned)((total + 255) / 256);
  if (dtype == S2A_DTYPE_F32) {
    if (layout == S2A_LAYOUT_NHWC)
      k_fam_refine<float, true><<<g, 256, 0, st>>>((const float*)bbox_pred, batch, height, width, stride, side, max_ratio, refined);
    else
      k_fam_refine<float, false><<<g, 256, 0, st>>>((const float*)bbox_pred, batch, height, width, stride, side, max_ratio, refined);
  } else if (dtype == S2A_DTYPE_F16) {
    if (layout == S2A_LAYOUT_NHWC)
      k_fam_refine<_Float16, true><<<g, 256, 0, st>>>((const _Float16*)bbox_pred, batch, height, width, stride, side, max_ratio, refined);
    else
      k_fam_refine<_Float16, false><<<g, 256, 0, st>>>((const _Float16*)bbox_pred, batch, height, width, stride, side, max_ratio, refined);
  } else {
    S2A_CHECK_ARG(false, "fam_refine_anchors: dtype");
  }
  S2A_LAUNCH_CHECK();
  return S2A_OK;
}

extern "C" int s2a_fam_refine_anchors_pyramid(const void* pred, int64_t row_stride, int64_t batch,
                                              const s2a_pyramid* pyr, float anchor_scale, float* refined,
                                              s2a_stream_t stream) {
  S2A_CHECK_ARG(pyr && pyr->n_levels >= 1 && pyr->n_levels <= 8 && batch >= 0 && row_stride >= 5,
                "fam_refine_anchors_pyramid: bad argument");
  RefineLevels lv = {};
  lv.n = pyr->n_levels;
  lv.batch = (int)batch;
  int64_t pix = 0;
  for (int i = 0; i < lv.n; i++) {
    S2A_CHECK_ARG(pyr->height[i] > 0 && pyr->width[i] > 0 && pyr->stride[i] > 0, "fam_refine_anchors_pyramid: bad level");
    lv.H[i] = pyr->height[i]; lv.W[i] = pyr->width[i]; lv.stride[i] = pyr->stride[i]; lv.pix0[i] = (int)pix;
    pix += batch * pyr->height[i] * pyr->width[i];
    S2A_CHECK_ARG(pix < (1ll << 31), "fam_refine_anchors_pyramid: too many positions");
  }
  if (pix == 0) return S2A_OK;
  S2A_CHECK_ARG(pred && refined, "fam_refine_anchors_pyramid: NULL tensor");
  const float max_ratio = (float)std::fabs(std::log(1e-6));  // head.py:48
  k_fam_refine_pyramid<<<(unsigned)((pix + 255) / 256), 256, 0, as_stream(stream)>>>(
      (const _Float16*)pred, (int)row_stride, lv, pix, anchor_scale, max_ratio, refined);
  S2A_LAUNCH_CHECK();
  return S2A_OK;
}

// rotated_box_to_poly_single (utils/general.py:886-921) for a whole batch: angle in [-pi/4, 3pi/4) -> the
// OpenCV convention (degrees in [0, 90], edges swapped above 90), then cv2.boxPoints.  boxPoints is OpenCV's
// RotatedRect::points (third-party, not in the reference tree; OpenCV 4.x modules/core/src/types.cpp), restated:
// b = (float)cos(a)*0.5f, a = (float)sin(a)*0.5f in double-evaluated trig, corners in float.
namespace s2a {
namespace {
__global__ void k_rbox_to_poly(const float* __restrict__ boxes, int64_t n, int64_t row_stride,
                               float* __restrict__ polys) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* r = boxes + i * row_stride;
  const float x = r[0], y = r[1], w = r[2], h = r[3];
  double angle = (double)r[4];
  if (angle < 0) angle += 3.141592653589793;
  angle = (angle / 3.141592653589793) * 180.0;
  float e1 = w, e2 = h;
  if (angle > 90) { angle -= 90; e1 = h; e2 = w; }
  const float af = (float)angle;                      // RotatedRect stores the angle as float
  const double rad = (double)af * 3.141592653589793 / 180.0;
  const float b = (float)cos(rad) * 0.5f, a = (float)sin(rad) * 0.5f;
  float p[8];
  p[0] = x - a * e2 - b * e1;
  p[1] = y + b * e2 - a * e1;
  p[2] = x + a * e2 - b * e1;
  p[3] = y - b * e2 - a * e1;
  p[4] = 2 * x - p[0];
  p[5] = 2 * y - p[1];
  p[6] = 2 * x - p[2];
  p[7] = 2 * y - p[3];
#pragma unroll
  for (int k = 0; k < 8; k++) polys[i * 8 + k] = p[k];
}
}  // namespace
}  // namespace s2a

extern "C" int s2a_rbox_to_poly(const float* boxes, int64_t n, int64_t row_stride, float* polys, s2a_stream_t stream) {
  S2A_CHECK_ARG(n >= 0 && row_stride >= 5, "rbox_to_poly: bad argument");
  if (n == 0) return S2A_OK;
  S2A_CHECK_ARG(boxes && polys, "rbox_to_poly: NULL tensor");
  k_rbox_to_poly<<<(unsigned)((n + 255) / 256), 256, 0, as_stream(stream)>>>(boxes, n, row_stride, polys);
  S2A_LAUNCH_CHECK();
  return S2A_OK;
}

extern "C" int s2a_align_offsets(const float* anchors, int64_t batch, int64_t height, int64_t width,
                                 float stride, int ksize, float* offset, s2a_stream_t stream) {
  S2A_CHECK_ARG(batch >= 0 && height >= 0 && width >= 0 && stride > 0 && ksize > 0 && (ksize & 1),
                "align_offsets: bad argument");
  const int64_t total = batch * height * width;
  if (total == 0) return S2A_OK;
  S2A_CHECK_ARG(anchors && offset, "align_offsets: NULL tensor");
  k_align_offsets<<<(unsigned)((total + 255) / 256), 256, 0, as_stream(stream)>>>(
      anchors, batch, height, width, stride, ksize, offset);
  S2A_LAUNCH_CHECK();
  return S2A_OK;
}
