// Training-side kernels of the deformable convolution: the three device
// functions the reference's backward is built from --
//   deformable_im2col        models/dcn/src/deform_conv_cuda_kernel.cu:189-276   (sampled columns)
//   deformable_col2im        :278-370   (gradient w.r.t. the input: bilinear scatter, atomics)
//   deformable_col2im_coord  :372-464   (gradient w.r.t. the offsets)
// with get_gradient_weight / get_coordinate_weight of :116-187.  The host side
// (s2anet_amd/dcn.py, mirroring deform_conv_cuda.cpp:262-489) chunks the batch by im2col_step and runs
// the two plain GEMMs around them on the library, exactly as the reference does with addmm_.
// Layouts are the reference's: im [S,C,H,W], offset [S, dg*2*kh*kw, Ho, Wo], columns
// [C*kh*kw, S*Ho*Wo] (row = (c*kh + i)*kw + j, column = (s*Ho + h)*Wo + w).  First version: one thread
// per element, HBM-bound like the reference's dataflow; the input gradient always accumulates in f32
// (the reference adds in the storage type, half atomics included).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>

#include "common.hpp"

namespace s2a {
namespace {

struct BwdGeom {
  int C, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, dg, S, Ho, Wo;
};

template <typename T>
__device__ __forceinline__ float bilinear_at(const T* __restrict__ im, int W, int H, float h, float w) {
  // deformable_im2col_bilinear (:83-114); caller has checked h > -1 && w > -1 && h < H && w < W
  int h_low = (int)floorf(h), w_low = (int)floorf(w);
  int h_high = h_low + 1, w_high = w_low + 1;
  float lh = h - h_low, lw = w - w_low, hh = 1 - lh, hw = 1 - lw;
  float v1 = (h_low >= 0 && w_low >= 0) ? (float)im[h_low * W + w_low] : 0.f;
  float v2 = (h_low >= 0 && w_high <= W - 1) ? (float)im[h_low * W + w_high] : 0.f;
  float v3 = (h_high <= H - 1 && w_low >= 0) ? (float)im[h_high * W + w_low] : 0.f;
  float v4 = (h_high <= H - 1 && w_high <= W - 1) ? (float)im[h_high * W + w_high] : 0.f;
  return hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4;
}

// one thread per (c, s, h_col, w_col): writes the kh*kw column entries of that channel/position
template <typename T>
__global__ void k_def_im2col(int64_t n, const T* __restrict__ im, const T* __restrict__ offset, BwdGeom g,
                             T* __restrict__ col) {
  for (int64_t index = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; index < n;
       index += (int64_t)gridDim.x * blockDim.x) {
    const int w_col = (int)(index % g.Wo);
    const int h_col = (int)((index / g.Wo) % g.Ho);
    const int s = (int)((index / g.Wo / g.Ho) % g.S);
    const int c = (int)(index / g.Wo / g.Ho / g.S);
    const int dgi = c / (g.C / g.dg);
    const int h_in = h_col * g.stride_h - g.pad_h, w_in = w_col * g.stride_w - g.pad_w;
    const int64_t HoWo = (int64_t)g.Ho * g.Wo;
    const T* imp = im + ((int64_t)s * g.C + c) * g.H * g.W;
    const T* offp = offset + ((int64_t)s * g.dg + dgi) * 2 * g.kh * g.kw * HoWo + (int64_t)h_col * g.Wo + w_col;
    T* colp = col + ((int64_t)c * g.kh * g.kw) * g.S * HoWo + ((int64_t)s * g.Ho + h_col) * g.Wo + w_col;
    for (int i = 0; i < g.kh; i++)
      for (int j = 0; j < g.kw; j++) {
        const int t = i * g.kw + j;
        const float oh = (float)offp[(int64_t)(2 * t) * HoWo], ow = (float)offp[(int64_t)(2 * t + 1) * HoWo];
        const float h_im = h_in + i * g.dil_h + oh, w_im = w_in + j * g.dil_w + ow;
        float val = 0.f;
        if (h_im > -1 && w_im > -1 && h_im < g.H && w_im < g.W) val = bilinear_at(imp, g.W, g.H, h_im, w_im);
        colp[(int64_t)t * g.S * HoWo] = (T)val;
      }
  }
}

__device__ __forceinline__ float gradient_weight(float ah, float aw, int h, int w, int H, int W) {
  if (ah <= -1 || ah >= H || aw <= -1 || aw >= W) return 0.f;
  int hl = (int)floorf(ah), wl = (int)floorf(aw), hh = hl + 1, wh = wl + 1;
  float weight = 0.f;
  if (h == hl && w == wl) weight = (h + 1 - ah) * (w + 1 - aw);
  if (h == hl && w == wh) weight = (h + 1 - ah) * (aw + 1 - w);
  if (h == hh && w == wl) weight = (ah + 1 - h) * (w + 1 - aw);
  if (h == hh && w == wh) weight = (ah + 1 - h) * (aw + 1 - w);
  return weight;
}

// one thread per column entry (c, i, j, s, h_out, w_out): scatter into grad_im (f32, atomics)
template <typename T>
__global__ void k_def_col2im(int64_t n, const T* __restrict__ col, const T* __restrict__ offset, BwdGeom g,
                             float* __restrict__ grad_im) {
  for (int64_t index = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; index < n;
       index += (int64_t)gridDim.x * blockDim.x) {
    const int w_out = (int)(index % g.Wo);
    const int h_out = (int)((index / g.Wo) % g.Ho);
    const int s = (int)((index / g.Wo / g.Ho) % g.S);
    const int j = (int)((index / g.Wo / g.Ho / g.S) % g.kw);
    const int i = (int)((index / g.Wo / g.Ho / g.S / g.kw) % g.kh);
    const int c = (int)(index / g.Wo / g.Ho / g.S / g.kw / g.kh);
    const int dgi = c / (g.C / g.dg);
    const int64_t HoWo = (int64_t)g.Ho * g.Wo;
    const T* offp = offset + ((int64_t)s * g.dg + dgi) * 2 * g.kh * g.kw * HoWo + (int64_t)h_out * g.Wo + w_out;
    const int t = i * g.kw + j;
    const float oh = (float)offp[(int64_t)(2 * t) * HoWo], ow = (float)offp[(int64_t)(2 * t + 1) * HoWo];
    const float ch = h_out * g.stride_h - g.pad_h + i * g.dil_h + oh;
    const float cw = w_out * g.stride_w - g.pad_w + j * g.dil_w + ow;
    const float top = (float)col[index];
    const int cur_h = (int)ch, cur_w = (int)cw;      // truncation toward zero, as the reference (:318-319)
    for (int dy = -2; dy <= 2; dy++)
      for (int dx = -2; dx <= 2; dx++) {
        const int y = cur_h + dy, x = cur_w + dx;
        if (y >= 0 && y < g.H && x >= 0 && x < g.W && fabsf(ch - y) < 1 && fabsf(cw - x) < 1) {
          const float wgt = gradient_weight(ch, cw, y, x, g.H, g.W);
          atomicAdd(grad_im + (((int64_t)s * g.C + c) * g.H + y) * g.W + x, wgt * top);
        }
      }
  }
}

// The same scatter for the AlignConv geometry (3x3, stride 1, dilation 1): one workgroup = an 8 x 32 tile of
// output positions x 8 channels.  Its contributions land in a (8+2+2*6) x (32+2+2*6) window around the tile:
// they are summed in an LDS copy of that window (ds_add_f32) and the window is flushed once (one global
// atomic per touched cell instead of one per tap and corner: ~9x fewer, row-contiguous); samples that leave
// the window go straight to memory.  The four bilinear corners are addressed directly -- the reference's
// 5 x 5 search (:321-337) visits exactly the cells with |dy| < 1, |dx| < 1, i.e. these four.
constexpr int kC2TH = 8, kC2TW = 32, kC2Halo = 6, kC2Ch = 8;
constexpr int kC2PH = kC2TH + 2 + 2 * kC2Halo, kC2PW = kC2TW + 2 + 2 * kC2Halo;   // 22 x 46

template <typename T>
__global__ __launch_bounds__(256) void k_def_col2im_tiled(const T* __restrict__ col, const T* __restrict__ offset,
                                                          BwdGeom g, float* __restrict__ grad_im) {
  __shared__ float s_patch[kC2Ch][kC2PH * kC2PW];
  const int tid = threadIdx.x;
  const int txn = (g.Wo + kC2TW - 1) / kC2TW, tyn = (g.Ho + kC2TH - 1) / kC2TH;
  int t = blockIdx.x;
  const int tx = t % txn;
  t /= txn;
  const int ty = t % tyn, s = t / tyn;
  const int c0 = blockIdx.y * kC2Ch;
  const int h_out = ty * kC2TH + tid / kC2TW, w_out = tx * kC2TW + tid % kC2TW;
  const int oy = ty * kC2TH - g.pad_h - kC2Halo, ox = tx * kC2TW - g.pad_w - kC2Halo;   // window origin (input coords)
  for (int i = tid; i < kC2Ch * kC2PH * kC2PW; i += 256) (&s_patch[0][0])[i] = 0.f;
  __syncthreads();
  const int64_t HoWo = (int64_t)g.Ho * g.Wo;
  const bool live = h_out < g.Ho && w_out < g.Wo;
  if (live) {
    const int dgi = c0 / (g.C / g.dg);          // the 8 channels share a deformable group (C/dg % 8 == 0, host check)
    const T* offp = offset + ((int64_t)s * g.dg + dgi) * 18 * HoWo + (int64_t)h_out * g.Wo + w_out;
    for (int tap = 0; tap < 9; tap++) {
      const float ch = h_out - g.pad_h + tap / 3 + (float)offp[(int64_t)(2 * tap) * HoWo];
      const float cw = w_out - g.pad_w + tap % 3 + (float)offp[(int64_t)(2 * tap + 1) * HoWo];
      if (ch <= -1 || ch >= g.H || cw <= -1 || cw >= g.W) continue;      // get_gradient_weight: empty
      const int hl = (int)floorf(ch), wl = (int)floorf(cw);
      const float lh = ch - hl, lw = cw - wl;
      const float wgt[4] = {(1 - lh) * (1 - lw), (1 - lh) * lw, lh * (1 - lw), lh * lw};
      for (int cc = 0; cc < kC2Ch; cc++) {
        const int c = c0 + cc;
        if (c >= g.C) break;
        const float top = (float)col[(((int64_t)c * 9 + tap) * g.S + s) * HoWo + (int64_t)h_out * g.Wo + w_out];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int y = hl + (k >> 1), x = wl + (k & 1);
          if (y < 0 || y >= g.H || x < 0 || x >= g.W || wgt[k] == 0.f) continue;
          const int py = y - oy, px = x - ox;
          if (py >= 0 && py < kC2PH && px >= 0 && px < kC2PW)
            atomicAdd(&s_patch[cc][py * kC2PW + px], wgt[k] * top);
          else
            atomicAdd(grad_im + (((int64_t)s * g.C + c) * g.H + y) * g.W + x, wgt[k] * top);
        }
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < kC2Ch * kC2PH * kC2PW; i += 256) {
    const int cc = i / (kC2PH * kC2PW), r = i % (kC2PH * kC2PW);
    const float v = s_patch[cc][r];
    const int y = oy + r / kC2PW, x = ox + r % kC2PW, c = c0 + cc;
    if (v != 0.f && c < g.C && y >= 0 && y < g.H && x >= 0 && x < g.W)
      atomicAdd(grad_im + (((int64_t)s * g.C + c) * g.H + y) * g.W + x, v);
  }
}

template <typename T>
__device__ __forceinline__ float coordinate_weight(float ah, float aw, int H, int W, const T* __restrict__ im,
                                                   int dir) {
  if (ah <= -1 || ah >= H || aw <= -1 || aw >= W) return 0.f;
  int hl = (int)floorf(ah), wl = (int)floorf(aw), hh = hl + 1, wh = wl + 1;
  float weight = 0.f;
  if (dir == 0) {
    if (hl >= 0 && wl >= 0) weight += -1 * (wl + 1 - aw) * (float)im[hl * W + wl];
    if (hl >= 0 && wh <= W - 1) weight += -1 * (aw - wl) * (float)im[hl * W + wh];
    if (hh <= H - 1 && wl >= 0) weight += (wl + 1 - aw) * (float)im[hh * W + wl];
    if (hh <= H - 1 && wh <= W - 1) weight += (aw - wl) * (float)im[hh * W + wh];
  } else {
    if (hl >= 0 && wl >= 0) weight += -1 * (hl + 1 - ah) * (float)im[hl * W + wl];
    if (hl >= 0 && wh <= W - 1) weight += (hl + 1 - ah) * (float)im[hl * W + wh];
    if (hh <= H - 1 && wl >= 0) weight += -1 * (ah - hl) * (float)im[hh * W + wl];
    if (hh <= H - 1 && wh <= W - 1) weight += (ah - hl) * (float)im[hh * W + wh];
  }
  return weight;
}

// one thread per offset element (s, offset channel, h, w): sum over the channels of its deformable group
template <typename T>
__global__ void k_def_col2im_coord(int64_t n, const T* __restrict__ col, const T* __restrict__ im,
                                   const T* __restrict__ offset, BwdGeom g, T* __restrict__ grad_offset) {
  const int offset_channels = 2 * g.kh * g.kw * g.dg;
  const int cpg = g.C / g.dg;                 // image channels per deformable group
  for (int64_t index = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; index < n;
       index += (int64_t)gridDim.x * blockDim.x) {
    const int w = (int)(index % g.Wo);
    const int h = (int)((index / g.Wo) % g.Ho);
    const int c = (int)((index / g.Wo / g.Ho) % offset_channels);
    const int s = (int)(index / g.Wo / g.Ho / offset_channels);
    const int dgi = c / (2 * g.kh * g.kw);
    const int offset_c = c - dgi * 2 * g.kh * g.kw;
    const int t = offset_c / 2, dir = offset_c % 2;
    const int i = t / g.kw, j = t % g.kw;
    const int64_t HoWo = (int64_t)g.Ho * g.Wo;
    const T* offp = offset + ((int64_t)s * g.dg + dgi) * 2 * g.kh * g.kw * HoWo + (int64_t)h * g.Wo + w;
    const float oh = (float)offp[(int64_t)(2 * t) * HoWo], ow = (float)offp[(int64_t)(2 * t + 1) * HoWo];
    float inv_h = h * g.stride_h - g.pad_h + i * g.dil_h + oh;
    float inv_w = w * g.stride_w - g.pad_w + j * g.dil_w + ow;
    if (inv_h <= -1 || inv_w <= -1 || inv_h >= g.H || inv_w >= g.W) inv_h = inv_w = -2;
    float val = 0.f;
    for (int cc = 0; cc < cpg; cc++) {
      const int ch = dgi * cpg + cc;
      const T* imp = im + ((int64_t)s * g.C + ch) * g.H * g.W;
      const float cw = coordinate_weight(inv_h, inv_w, g.H, g.W, imp, dir);
      const int64_t col_pos = (((int64_t)ch * g.kh * g.kw + t) * g.S + s) * HoWo + (int64_t)h * g.Wo + w;
      val += cw * (float)col[col_pos];
    }
    grad_offset[index] = (T)val;
  }
}

int make_geom(const s2a_dcn_params* pp, BwdGeom* g, const char* who) {
  S2A_CHECK_ARG(pp != nullptr, "%s: NULL params", who);
  const s2a_dcn_params& p = *pp;
  S2A_CHECK_ARG(p.kW > 0 && p.kH > 0 && p.dW > 0 && p.dH > 0 && p.dilationW > 0 && p.dilationH > 0 &&
                p.deformable_group > 0, "%s: bad kernel geometry", who);
  S2A_CHECK_ARG(p.batch >= 0 && p.channels > 0 && p.height > 0 && p.width > 0, "%s: bad shape", who);
  S2A_CHECK_ARG(p.channels % p.deformable_group == 0, "input channels must divide deformable group size");
  S2A_CHECK_ARG(p.dtype == S2A_DTYPE_F32 || p.dtype == S2A_DTYPE_F16, "%s: dtype", who);
  S2A_CHECK_ARG(p.layout == S2A_LAYOUT_NCHW, "%s: NCHW only", who);
  const int64_t Ho = (p.height + 2 * p.padH - (p.dilationH * (p.kH - 1) + 1)) / p.dH + 1;
  const int64_t Wo = (p.width + 2 * p.padW - (p.dilationW * (p.kW - 1) + 1)) / p.dW + 1;
  S2A_CHECK_ARG(Ho >= 1 && Wo >= 1, "%s: output size is too small", who);
  S2A_CHECK_ARG(p.height < (1 << 15) && p.width < (1 << 15) && p.channels * p.kH * p.kW < (1ll << 31),
                "%s: shape too large", who);
  *g = BwdGeom{(int)p.channels, (int)p.height, (int)p.width, p.kH, p.kW, p.padH, p.padW, p.dH, p.dW,
               p.dilationH, p.dilationW, p.deformable_group, (int)p.batch, (int)Ho, (int)Wo};
  return S2A_OK;
}

inline unsigned grid_for(int64_t n) { return (unsigned)std::min<int64_t>((n + 255) / 256, 1 << 20); }

}  // namespace
}  // namespace s2a

using namespace s2a;

extern "C" int s2a_deformable_im2col(const void* im, const void* offset, void* columns, const s2a_dcn_params* p,
                                     s2a_stream_t stream) {
  BwdGeom g;
  int rc = make_geom(p, &g, "deformable_im2col");
  if (rc != S2A_OK) return rc;
  const int64_t n = (int64_t)g.C * g.S * g.Ho * g.Wo;
  if (n == 0) return S2A_OK;
  S2A_CHECK_ARG(im && offset && columns, "deformable_im2col: NULL tensor");
  hipStream_t st = as_stream(stream);
  if (p->dtype == S2A_DTYPE_F32)
    k_def_im2col<float><<<grid_for(n), 256, 0, st>>>(n, (const float*)im, (const float*)offset, g, (float*)columns);
  else
    k_def_im2col<_Float16><<<grid_for(n), 256, 0, st>>>(n, (const _Float16*)im, (const _Float16*)offset, g, (_Float16*)columns);
  S2A_LAUNCH_CHECK();
  return S2A_OK;
}

extern "C" int s2a_deformable_col2im(const void* columns, const void* offset, float* grad_im_f32,
                                     const s2a_dcn_params* p, s2a_stream_t stream) {
  BwdGeom g;
  int rc = make_geom(p, &g, "deformable_col2im");
  if (rc != S2A_OK) return rc;
  const int64_t n = (int64_t)g.C * g.kh * g.kw * g.S * g.Ho * g.Wo;
  if (n == 0) return S2A_OK;
  S2A_CHECK_ARG(columns && offset && grad_im_f32, "deformable_col2im: NULL tensor");
  hipStream_t st = as_stream(stream);
  const bool tiled = g.kh == 3 && g.kw == 3 && g.stride_h == 1 && g.stride_w == 1 && g.dil_h == 1 && g.dil_w == 1 &&
                     (g.C / g.dg) % kC2Ch == 0 && !getenv("S2A_COL2IM_SIMPLE");
  if (tiled) {
    dim3 grid((unsigned)((int64_t)g.S * ((g.Ho + kC2TH - 1) / kC2TH) * ((g.Wo + kC2TW - 1) / kC2TW)),
              (unsigned)((g.C + kC2Ch - 1) / kC2Ch));
    if (p->dtype == S2A_DTYPE_F32)
      k_def_col2im_tiled<float><<<grid, 256, 0, st>>>((const float*)columns, (const float*)offset, g, grad_im_f32);
    else
      k_def_col2im_tiled<_Float16><<<grid, 256, 0, st>>>((const _Float16*)columns, (const _Float16*)offset, g, grad_im_f32);
  } else if (p->dtype == S2A_DTYPE_F32)
    k_def_col2im<float><<<grid_for(n), 256, 0, st>>>(n, (const float*)columns, (const float*)offset, g, grad_im_f32);
  else
    k_def_col2im<_Float16><<<grid_for(n), 256, 0, st>>>(n, (const _Float16*)columns, (const _Float16*)offset, g, grad_im_f32);
  S2A_LAUNCH_CHECK();
  return S2A_OK;
}

extern "C" int s2a_deformable_col2im_coord(const void* columns, const void* im, const void* offset,
                                           void* grad_offset, const s2a_dcn_params* p, s2a_stream_t stream) {
  BwdGeom g;
  int rc = make_geom(p, &g, "deformable_col2im_coord");
  if (rc != S2A_OK) return rc;
  const int64_t n = (int64_t)g.S * 2 * g.kh * g.kw * g.dg * g.Ho * g.Wo;
  if (n == 0) return S2A_OK;
  S2A_CHECK_ARG(columns && im && offset && grad_offset, "deformable_col2im_coord: NULL tensor");
  hipStream_t st = as_stream(stream);
  if (p->dtype == S2A_DTYPE_F32)
    k_def_col2im_coord<float><<<grid_for(n), 256, 0, st>>>(n, (const float*)columns, (const float*)im,
                                                           (const float*)offset, g, (float*)grad_offset);
  else
    k_def_col2im_coord<_Float16><<<grid_for(n), 256, 0, st>>>(n, (const _Float16*)columns, (const _Float16*)im,
                                                              (const _Float16*)offset, g, (_Float16*)grad_offset);
  S2A_LAUNCH_CHECK();
  return S2A_OK;
}
