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
__global__ __launch_bounds__(256) void k_bias_act(const T* y, T* out, const T* __restrict__ bias,
                                                  const T* __restrict__ res, int64_t nvec, int cvec,
                                                  int relu) {      // out == y: in place (every element is read before it is written)
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
      v[k] = reinterpret_cast<const V*>(y)[i + k * T0];
      if (res) r[k] = reinterpret_cast<const V*>(res)[i + k * T0];
      b[k] = HOIST ? b0 : reinterpret_cast<const V*>(bias)[(i + k * T0) % cvec];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) reinterpret_cast<V*>(out)[i + k * T0] = apply(v[k], b[k], r[k]);
  }
  for (; i < nvec; i += T0) {
    V v = reinterpret_cast<const V*>(y)[i], r;
    if (res) r = reinterpret_cast<const V*>(res)[i];
    V b = HOIST ? b0 : reinterpret_cast<const V*>(bias)[i % cvec];
    reinterpret_cast<V*>(out)[i] = apply(v, b, r);
  }
}

}  // namespace
}  // namespace s2a

using namespace s2a;

static int bias_act_impl(const void* y, void* out, const void* bias, const void* residual, int64_t positions,
                         int64_t channels, int dtype, int relu, s2a_stream_t stream) {
  S2A_CHECK_ARG(positions >= 0 && channels > 0, "bias_act: bad shape");
  S2A_CHECK_ARG(dtype == S2A_DTYPE_F32 || dtype == S2A_DTYPE_F16, "bias_act: dtype");
  const int vec = dtype == S2A_DTYPE_F16 ? 8 : 4;
  S2A_CHECK_ARG(channels % vec == 0, "bias_act: channels must be a multiple of %d", vec);
  if (positions == 0) return S2A_OK;
  S2A_CHECK_ARG(y && bias, "bias_act: NULL tensor");
  S2A_CHECK_ARG(((uintptr_t)y % 16) == 0 && ((uintptr_t)bias % 16) == 0 && ((uintptr_t)residual % 16) == 0 &&
                ((uintptr_t)out % 16) == 0 && out, "bias_act: tensors must be 16-byte aligned");
  const int64_t nvec = positions * channels / vec;
  const int cvec = (int)(channels / vec);
  unsigned g = (unsigned)std::min<int64_t>((nvec + 1023) / 1024, 256 * 8);
  if (g == 0) g = 1;
  const bool hoist = cvec <= 256 && (256 % cvec) == 0;   // then gridDim*256 is a multiple of cvec
  hipStream_t st = as_stream(stream);
#define S2A_BA(T, V, H) k_bias_act<T, V, H><<<g, 256, 0, st>>>((const T*)y, (T*)out, (const T*)bias, (const T*)residual, nvec, cvec, relu)
  if (dtype == S2A_DTYPE_F16) {
    if (hoist) S2A_BA(_Float16, 8, true); else S2A_BA(_Float16, 8, false);
  } else {
    if (hoist) S2A_BA(float, 4, true); else S2A_BA(float, 4, false);
  }
#undef S2A_BA
  S2A_LAUNCH_CHECK();
  return S2A_OK;
}

extern "C" int s2a_bias_act_nhwc(void* y, const void* bias, const void* residual, int64_t positions,
                                 int64_t channels, int dtype, int relu, s2a_stream_t stream) {
  return bias_act_impl(y, y, bias, residual, positions, channels, dtype, relu, stream);
}

extern "C" int s2a_bias_act_nhwc_to(const void* y, const void* bias, const void* residual, void* out, int64_t positions,
                                    int64_t channels, int dtype, int relu, s2a_stream_t stream) {
  return bias_act_impl(y, out, bias, residual, positions, channels, dtype, relu, stream);
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

// ---------------------------------------------------------------- candidate selection of get_bboxes, all levels
// get_bboxes_single_img (models/head.py:684-717) for the whole batch on pyramid-packed predictions: per level and
// image, sigmoid -> max over classes -> top-k (k = 2000) only where H*W > k (:697-705); levels concatenated
// (:712-714); final decode (:717, wh_ratio_clip 16/1000).  Two kernels instead of ~25 stock launches:
//  k_pyr_keys + k_pyr_topk: one workgroup per (image, level): exact k-th largest key by a bitwise radix select on
//   16-bit keys held in LDS (key = order-preserving image of max_c logit; sigmoid is monotonic), then an index-order
//   compaction (ties at the threshold: lowest positions first) -> packed row indices;
//  k_pyr_gather: one thread per selected row: f16 sigmoid of the class logits (as the stock half sigmoid: float
//   math, rounded to half), decode of the box against its refined anchor.
namespace s2a {
namespace {
constexpr int kTopkThreads = 1024;
constexpr int kTopkMaxN = 24576;     // positions of one level of one image held in LDS as 16-bit keys (48 KB)

struct CandLevels {
  int n, batch, num_classes, k;
  int HW[8], pix0[8], out0[8];       // positions per image, first packed row, first output slot of the level
};

__device__ __forceinline__ unsigned short f16_key(_Float16 v) {
  unsigned short u = __builtin_bit_cast(unsigned short, v);
  return (u & 0x8000u) ? (unsigned short)~u : (unsigned short)(u | 0x8000u);
}

// key = order-preserving image of max_c logit for every packed row, one thread per row, the whole chip at once (one
// workgroup per (image, level) pulling its level's rows -- 2 MB of 128-byte lines for P3 -- through ONE CU took 40 us)
__global__ __launch_bounds__(256) void k_pyr_keys(const _Float16* __restrict__ cls, int64_t rows, int num_classes,
                                                  unsigned short* __restrict__ keys) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows) return;
  using V8 = __attribute__((ext_vector_type(8))) _Float16;
  const V8* r = reinterpret_cast<const V8*>(cls + i * 64);
  const int nvec = (num_classes + 7) / 8;
  _Float16 m = r[0][0];
  for (int v = 0; v < nvec; v++) {
    const V8 x = r[v];
#pragma unroll
    for (int c = 0; c < 8; c++)
      if (v * 8 + c < num_classes) m = x[c] > m ? x[c] : m;
  }
  keys[i] = f16_key(m);
}

__global__ __launch_bounds__(kTopkThreads) void k_pyr_topk(const _Float16* __restrict__ cls, CandLevels lv,
                                                           int64_t n_out, int32_t* __restrict__ sel,
                                                           const unsigned short* __restrict__ keys) {
  __shared__ unsigned short s_key[kTopkMaxN];
  __shared__ unsigned s_hist[256];
  __shared__ unsigned s_part[kTopkThreads];
  __shared__ unsigned s_T, s_need_eq;
  const int l = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int HW = lv.HW[l], k = lv.k;
  const int64_t row0 = (int64_t)lv.pix0[l] + (int64_t)b * HW;
  int32_t* out = sel + (int64_t)b * n_out + lv.out0[l];
  if (HW <= k || k <= 0) {                        // whole level (head.py:697: top-k only when H*W > k)
    for (int i = tid; i < HW; i += kTopkThreads) out[i] = (int32_t)(row0 + i);
    return;
  }
  // per-position max over the class maps: rows are 128-byte aligned, the classes sit in the first (num_classes + 7) / 8
  // 16-byte vectors; four positions per thread in flight (scalar 2-byte loads, one position at a time, took 165 us
  // for the 16 384 positions of a P3 level: the whole kernel was that loop)
  using V8 = __attribute__((ext_vector_type(8))) _Float16;
  const int nvec = (lv.num_classes + 7) / 8;
  if (keys) {                                      // precomputed by k_pyr_keys
    for (int i = tid; i < HW; i += kTopkThreads) s_key[i] = keys[row0 + i];
  } else if (nvec > 2) {                           // more than 16 classes: plain loop
    for (int i = tid; i < HW; i += kTopkThreads) {
      const _Float16* r = cls + (row0 + i) * 64;
      _Float16 m = r[0];
      for (int c = 1; c < lv.num_classes; c++) m = r[c] > m ? r[c] : m;
      s_key[i] = f16_key(m);
    }
  } else
  for (int i = tid; i < HW; i += 4 * kTopkThreads) {
    V8 v[4][2];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int iu = min(i + u * kTopkThreads, HW - 1);
      const V8* r = reinterpret_cast<const V8*>(cls + (row0 + iu) * 64);
      v[u][0] = r[0];
      v[u][1] = nvec > 1 ? r[1] : r[0];
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int iu = i + u * kTopkThreads;
      if (iu >= HW) break;
      _Float16 m = v[u][0][0];
#pragma unroll
      for (int c = 1; c < 16; c++) {
        const _Float16 x = c < 8 ? v[u][0][c] : v[u][1][c - 8];
        if (c < lv.num_classes) m = x > m ? x : m;
      }
      s_key[iu] = f16_key(m);
    }
  }
  __syncthreads();
  // exact k-th largest key by a bitwise radix select, most significant bit first: every thread keeps its contiguous
  // chunk of keys in registers and counts the keys that match the prefix found so far with the next bit set; one
  // barrier per bit (wave sums through DPP shuffles, 16 partials in LDS, double-buffered).  (Two byte-wide histogram
  // passes with LDS atomics took 50 us on clustered scores: most positions of a level share one high byte.)
  constexpr int kPerMax = kTopkMaxN / kTopkThreads;          // 24
  const int per = (HW + kTopkThreads - 1) / kTopkThreads;
  const int i0 = tid * per, i1 = min(HW, i0 + per);
  unsigned kreg[kPerMax];
#pragma unroll
  for (int j = 0; j < kPerMax; j++) kreg[j] = (i0 + j < i1) ? (unsigned)s_key[i0 + j] : 0x10000u;   // 0x10000: never matches
  unsigned prefix = 0, need = (unsigned)k;
  unsigned* s_cnt = s_hist;                                   // [2][16]
  for (int bit = 15; bit >= 0; bit--) {
    const unsigned cand = prefix | (1u << bit), mask = (0xffffu << bit) & 0xffffu;
    unsigned cnt = 0;
#pragma unroll
    for (int j = 0; j < kPerMax; j++) cnt += ((kreg[j] & (mask | 0x10000u)) == cand) ? 1u : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    unsigned* sc = s_cnt + (bit & 1) * 16;
    if ((tid & 63) == 0) sc[tid >> 6] = cnt;
    __syncthreads();
    unsigned tot = 0;
#pragma unroll
    for (int w = 0; w < kTopkThreads / 64; w++) tot += sc[w];
    if (tot >= need) prefix = cand;        // the k-th largest key has this bit set
    else need -= tot;                      // all keys with this bit set (under the prefix) are above it
  }
  if (tid == 0) { s_T = prefix; s_need_eq = need; }   // need = how many rows equal to the k-th key are taken
  __syncthreads();
  const unsigned T = s_T, need_eq = s_need_eq;
  // index-order compaction: contiguous chunk per thread, two running counts (selected so far, equals so far)
  unsigned gt = 0, eq = 0;
  for (int i = i0; i < i1; i++) {
    gt += s_key[i] > T;
    eq += s_key[i] == T;
  }
  s_part[tid] = (gt << 16) | eq;                   // both < 65536
  __syncthreads();
  // exclusive scan of the 1024 partials (Hillis-Steele in LDS; packed halves cannot carry: totals <= 32768)
  for (int off = 1; off < kTopkThreads; off <<= 1) {
    unsigned v = tid >= off ? s_part[tid - off] : 0u;
    __syncthreads();
    s_part[tid] += v;
    __syncthreads();
  }
  const unsigned incl = s_part[tid];
  unsigned gt0 = (incl >> 16) - gt, eq0 = (incl & 0xffffu) - eq;
  for (int i = i0; i < i1; i++) {
    const unsigned key = s_key[i];
    if (key > T) {
      out[gt0 + min(eq0, need_eq)] = (int32_t)(row0 + i);
      gt0++;
    } else if (key == T) {
      if (eq0 < need_eq) out[gt0 + eq0] = (int32_t)(row0 + i);
      eq0++;
    }
  }
}

__global__ void k_pyr_gather(const _Float16* __restrict__ cls, const _Float16* __restrict__ reg,
                             const float* __restrict__ anchors, const int32_t* __restrict__ sel, int64_t total,
                             int num_classes, float max_ratio, float* __restrict__ bboxes,
                             float* __restrict__ scores) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int64_t row = sel[e];
  const _Float16* c = cls + row * 64;
  for (int k = 0; k < num_classes; k++) {
    const float x = (float)c[k];
    scores[e * num_classes + k] = (float)(_Float16)(1.f / (1.f + expf(-x)));      // half sigmoid, then .float()
  }
  const _Float16* d16 = reg + row * 64;
  float r[5], d[5], o[5];
#pragma unroll
  for (int k = 0; k < 5; k++) {
    r[k] = anchors[row * 5 + k];
    d[k] = (float)d16[k];
  }
  decode_one(r, d, max_ratio, o);
#pragma unroll
  for (int k = 0; k < 5; k++) bboxes[e * 5 + k] = o[k];
}

int cand_levels(const s2a_pyramid* pyr, int64_t batch, int num_classes, int64_t k, CandLevels* lv, int64_t* n_out) {
  if (!pyr || pyr->n_levels < 1 || pyr->n_levels > 8) return -1;
  *lv = CandLevels{};
  lv->n = pyr->n_levels; lv->batch = (int)batch; lv->num_classes = num_classes; lv->k = (int)k;
  int64_t pix = 0, out = 0;
  for (int i = 0; i < lv->n; i++) {
    const int64_t hw = (int64_t)pyr->height[i] * pyr->width[i];
    if (hw < 1 || hw >= (1ll << 31)) return -1;
    if (k > 0 && hw > k && hw > kTopkMaxN) return -2;
    lv->HW[i] = (int)hw; lv->pix0[i] = (int)pix; lv->out0[i] = (int)out;
    pix += batch * hw;
    out += (k > 0 && hw > k) ? k : hw;
    if (pix >= (1ll << 31)) return -1;
  }
  *n_out = out;
  return 0;
}
}  // namespace
}  // namespace s2a

extern "C" int64_t s2a_pyramid_candidates_count(const s2a_pyramid* pyr, int64_t max_per_level) {
  CandLevels lv; int64_t n = 0;
  return cand_levels(pyr, 1, 1, max_per_level, &lv, &n) == 0 ? n : -1;
}

extern "C" int s2a_pyramid_candidates(const void* cls, const void* reg, const float* anchors, int64_t batch,
                                      const s2a_pyramid* pyr, int num_classes, int64_t max_per_level,
                                      float wh_ratio_clip, float* bboxes, float* scores, int32_t* sel,
                                      s2a_stream_t stream) {
  S2A_CHECK_ARG(batch >= 0 && num_classes >= 1 && num_classes <= 64 && wh_ratio_clip > 0, "pyramid_candidates: bad argument");
  CandLevels lv; int64_t n = 0;
  const int rc = cand_levels(pyr, batch, num_classes, max_per_level, &lv, &n);
  S2A_CHECK_ARG(rc != -2, "pyramid_candidates: a level with more than 24576 positions needs the per-level path");
  S2A_CHECK_ARG(rc == 0, "pyramid_candidates: bad level table");
  if (batch == 0 || n == 0) return S2A_OK;
  S2A_CHECK_ARG(cls && reg && anchors && bboxes && scores && sel, "pyramid_candidates: NULL tensor");
  hipStream_t st = as_stream(stream);
  const int64_t total = batch * n;
  // keys of all rows first, in the scores buffer (written only by k_pyr_gather afterwards) when it is large enough
  int64_t rows = 0;
  for (int i = 0; i < lv.n; i++) rows += batch * (int64_t)lv.HW[i];
  const unsigned short* keys = nullptr;
  if (rows * (int64_t)sizeof(unsigned short) <= total * num_classes * (int64_t)sizeof(float) && ((uintptr_t)cls % 16) == 0) {
    k_pyr_keys<<<(unsigned)((rows + 255) / 256), 256, 0, st>>>((const _Float16*)cls, rows, num_classes, (unsigned short*)scores);
    keys = (const unsigned short*)scores;
  }
  k_pyr_topk<<<dim3((unsigned)lv.n, (unsigned)batch), kTopkThreads, 0, st>>>((const _Float16*)cls, lv, n, sel, keys);
  const float max_ratio = (float)std::fabs(std::log((double)wh_ratio_clip));
  k_pyr_gather<<<(unsigned)((total + 255) / 256), 256, 0, st>>>((const _Float16*)cls, (const _Float16*)reg, anchors, sel,
                                                               total, num_classes, max_ratio, bboxes, scores);
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
