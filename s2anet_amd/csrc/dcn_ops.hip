// Deformable convolution v1 forward / AlignConv for MI355X (gfx950).
//
// Replaces:
//   deform_conv_forward_cuda        models/dcn/src/deform_conv_cuda.cpp:152-260
//   deformable_im2col_gpu_kernel    models/dcn/src/deform_conv_cuda_kernel.cu:189-242
//   deformable_im2col_bilinear      models/dcn/src/deform_conv_cuda_kernel.cu:83-114
//   AlignConv.get_offset / forward  models/alignconv.py:30-98
//
// The reference materialises a [C*9, B*H*W] `columns` matrix in HBM (1.2 GB at P3, B=8),
// zero-fills it, writes it with an uncoalesced NCHW gather and re-reads it in a cuBLAS GEMM.
// Here the op is ONE kernel per call and `columns` never exists:
//   * activations are read channels-last (NHWC): a bilinear corner is a contiguous run of
//     channels, so every gather instruction moves full 128-byte lines
//   * a workgroup owns 64 output positions x all (<=256) output channels; per (tap, channel
//     chunk) stage it gathers + blends the column tile straight into LDS and streams the
//     matching pre-packed weight tile into LDS
//   * the contraction runs on the matrix cores: v_mfma_f32_32x32x2_f32 (exact f32 FMA chain,
//     1e-4 parity with the f32 reference) or v_mfma_f32_32x32x16_f16 (f16 in, f32 accumulate)
//   * LDS tiles are double-buffered (one barrier per stage): next stage's global loads are
//     in flight under the current stage's MFMAs
//   * sampling coordinates come either from the reference's [B,18,H,W] offset tensor or —
//     AlignConv fused — directly from the refined anchors (the 18-channel tensor is never
//     written), with ReLU fused into the epilogue
// Sampling semantics (border rule h_im > -1 && < H, corner dropping, weight order) follow
// deform_conv_cuda_kernel.cu:97-112,:228 exactly.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "common.hpp"

namespace s2a {
namespace {

// S2A_ABL: compile-time ablation switches for timing experiments only (never set in a shipped build)
#ifndef S2A_ABL
#define S2A_ABL 0
#endif
#ifndef S2A_STAMP
#define S2A_STAMP 0
#endif
// measurement builds (-DS2A_MEASURE) only: S2A_DCN_DROP=x|w gives the wave-specialised kernel zero-record descriptors (the
// loads are dropped, the instruction stream stays; OUTPUTS ARE WRONG) -- never compiled into a shipped library
#ifdef S2A_MEASURE
#define S2A_DCN_DROP_SWITCH(xb, wb)                     \
  do {                                                  \
    const char* drop = getenv("S2A_DCN_DROP");          \
    if (drop && strchr(drop, 'x')) (xb) = 0;            \
    if (drop && strchr(drop, 'w')) (wb) = 0;            \
  } while (0)
#else
#define S2A_DCN_DROP_SWITCH(xb, wb) do {} while (0)
#endif
#ifndef S2A_STAMP_W2
#define S2A_STAMP_W2 4        // second stamped wave (0 is the first)
#endif
#if S2A_STAMP
// diagnostic build only: per-workgroup phase stamps (s_memtime) into a buffer nothing else reads
__device__ unsigned long long g_stamps[4096 * 16];
#define S2A_STAMP_AT(slot)                                                                   \
  do {                                                                                       \
    if (lane == 0 && (wave == 0 || wave == S2A_STAMP_W2))                                    \
      g_stamps[((blockIdx.x & 4095) * 16) + (wave ? 8 : 0) + (slot)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#define S2A_TIC() (t_tic = __builtin_amdgcn_s_memtime())
#define S2A_TOC(accv) (accv += __builtin_amdgcn_s_memtime() - t_tic)
#define S2A_STAMP_VAL(slot, val)                                                             \
  do {                                                                                       \
    if (lane == 0 && (wave == 0 || wave == S2A_STAMP_W2))                                    \
      g_stamps[((blockIdx.x & 4095) * 16) + (wave ? 8 : 0) + (slot)] = (val);                \
  } while (0)
#else
#define S2A_STAMP_AT(slot) do {} while (0)
#define S2A_TIC() do {} while (0)
#define S2A_TOC(accv) do {} while (0)
#define S2A_STAMP_VAL(slot, val) do {} while (0)
#endif
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

constexpr int kPos = 64;        // output positions per workgroup
constexpr int kMaxO = 256;      // output channels per workgroup
constexpr int kRowBytes = 144;  // LDS row: 128 B of K data + 16 B pad (conflict-free b128 reads)

// ------------------------------------------------------------------ layout helpers
template <typename T>
__global__ __launch_bounds__(256) void k_nchw_to_nhwc(const T* __restrict__ src, int64_t B, int C,
                                                      int64_t HW, T* __restrict__ dst) {
  // 32x32 tile transpose through LDS: reads coalesced along HW, writes coalesced along C
  __shared__ T tile[32][33];
  const int64_t b = blockIdx.z;
  const int64_t p0 = (int64_t)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    int c = c0 + r;
    int64_t p = p0 + tx;
    if (c < C && p < HW) tile[r][tx] = src[(b * C + c) * HW + p];
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    int64_t p = p0 + r;
    int c = c0 + tx;
    if (c < C && p < HW) dst[(b * HW + p) * C + c] = tile[tx][r];
  }
}

// weight [O][C][9] -> packed [C/KC][9][O][KC]  (KC = channels per stage; stage s = cc*9 + tap:
// the 9 taps of one channel chunk run back to back, so their overlapping corner pixels stay in L1)
template <typename T>
__global__ void k_pack_weight(const T* __restrict__ w, int O, int C, int KC, T* __restrict__ wp) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t total = (int64_t)O * C * 9;
  if (e >= total) return;
  int k = (int)(e % KC);
  int64_t r = e / KC;
  int o = (int)(r % O);
  r /= O;
  int t = (int)(r % 9);
  int cc = (int)(r / 9);
  wp[e] = w[((int64_t)o * C + cc * KC + k) * 9 + t];
}

// 2-D position tiles: a workgroup owns a 16-wide x (NPOS/16)-high patch of one image, so the
// footprint of its bilinear corners (patch + halo) is a few hundred pixels instead of a whole
// image row.  Returns the linear position b*H*W + y*W + x, or -1 outside the image / batch.
__device__ __forceinline__ int64_t tile_pos(int64_t tile, int pl, int th, int H, int W, int64_t HW,
                                            int64_t Ntot) {
  const int txn = (W + 15) / 16, tyn = (H + th - 1) / th;
  const int64_t b = tile / (txn * tyn);
  const int r = (int)(tile % (txn * tyn));
  const int y = (r / txn) * th + (pl >> 4), xq = (r % txn) * 16 + (pl & 15);
  const int64_t g = b * HW + (int64_t)y * W + xq;
  return (y < H && xq < W && g < Ntot) ? g : -1;
}

// ------------------------------------------------------------------ sampling table
struct Tap {
  int idx[4];   // global pixel index (b*H*W + h*W + w) of the 4 corners, 0 when dropped
  float w[4];   // bilinear weights hh*hw, hh*lw, lh*hw, lh*lw; 0 when dropped / outside
};

__device__ __forceinline__ Tap make_tap(float h_im, float w_im, int H, int W, int64_t img_pix0) {
  Tap t;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    t.idx[k] = (int)img_pix0;
    t.w[k] = 0.f;
  }
  if (h_im > -1 && w_im > -1 && h_im < H && w_im < W) {  // kernel.cu:228
    int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
    int h_high = h_low + 1, w_high = w_low + 1;
    float lh = h_im - h_low, lw = w_im - w_low;
    float hh = 1 - lh, hw = 1 - lw;
    if (h_low >= 0 && w_low >= 0) {
      t.idx[0] = (int)(img_pix0 + (int64_t)h_low * W + w_low);
      t.w[0] = hh * hw;
    }
    if (h_low >= 0 && w_high <= W - 1) {
      t.idx[1] = (int)(img_pix0 + (int64_t)h_low * W + w_high);
      t.w[1] = hh * lw;
    }
    if (h_high <= H - 1 && w_low >= 0) {
      t.idx[2] = (int)(img_pix0 + (int64_t)h_high * W + w_low);
      t.w[2] = lh * hw;
    }
    if (h_high <= H - 1 && w_high <= W - 1) {
      t.idx[3] = (int)(img_pix0 + (int64_t)h_high * W + w_high);
      t.w[3] = lh * lw;
    }
  }
  return t;
}

// sampling point of tap (ky,kx) at output (y,x), 3x3 / stride 1 / pad 1 / dilation 1
struct AnchorCtx {
  float x_ctr, y_ctr, dw, dh, cs, sn;
};
__device__ __forceinline__ AnchorCtx anchor_ctx(const float* a, float stride) {
  // models/alignconv.py:58-62
  AnchorCtx c;
  c.x_ctr = a[0] / stride;
  c.y_ctr = a[1] / stride;
  float w = a[2] / stride, h = a[3] / stride;
  c.cs = cosf(a[4]);
  c.sn = sinf(a[4]);
  c.dw = w / 3.0f;
  c.dh = h / 3.0f;
  return c;
}
__device__ __forceinline__ void anchor_offset(const AnchorCtx& c, int ky, int kx, float yc, float xc,
                                              float& off_y, float& off_x) {
  // models/alignconv.py:63-70 (same operation order as s2a_align_offsets)
  float xx = (float)(kx - 1), yy = (float)(ky - 1);
  float x = c.dw * xx, y = c.dh * yy;
  float xr = c.cs * x - c.sn * y;
  float yr = c.sn * x + c.cs * y;
  off_x = (xr + c.x_ctr) - (xc + xx);
  off_y = (yr + c.y_ctr) - (yc + yy);
}

// ------------------------------------------------------------------ fused MFMA kernel
template <typename T>
struct Traits;
template <>
struct Traits<float> {
  static constexpr int KC = 32;      // channels per stage (128 B per corner)
  static constexpr int VEC = 4;      // elements per 16-B vector
};
template <>
struct Traits<_Float16> {
  static constexpr int KC = 64;
  static constexpr int VEC = 8;
};

template <typename T>
struct Vec16;
template <>
struct Vec16<float> {
  using type = f32x4;
};
template <>
struct Vec16<_Float16> {
  using type = f16x8;
};

template <typename T>
__device__ __forceinline__ typename Vec16<T>::type blend(const typename Vec16<T>::type (&v)[4],
                                                         const float (&w)[4]);
template <>
__device__ __forceinline__ f32x4 blend<float>(const f32x4 (&v)[4], const float (&w)[4]) {
  f32x4 r;
#pragma unroll
  for (int e = 0; e < 4; e++) r[e] = w[0] * v[0][e] + w[1] * v[1][e] + w[2] * v[2][e] + w[3] * v[3][e];
  return r;
}
template <>
__device__ __forceinline__ f16x8 blend<_Float16>(const f16x8 (&v)[4], const float (&w)[4]) {
  f16x8 r;
#pragma unroll
  for (int e = 0; e < 8; e++)
    r[e] = (_Float16)(w[0] * (float)v[0][e] + w[1] * (float)v[1][e] + w[2] * (float)v[2][e] +
                      w[3] * (float)v[3][e]);
  return r;
}

// f16 blend in packed half arithmetic (v_pk_mul/v_pk_fma_f16): the reference's half path
// evaluates w1*v1 + w2*v2 + w3*v3 + w4*v4 in scalar_t = half as well (kernel.cu:110-112)
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;
__device__ __forceinline__ f16x8 blend_pk(const f16x8 (&v)[4], const float (&w)[4]) {
  f16x8 r;
  f16x2 w0 = {(_Float16)w[0], (_Float16)w[0]}, w1 = {(_Float16)w[1], (_Float16)w[1]};
  f16x2 w2 = {(_Float16)w[2], (_Float16)w[2]}, w3 = {(_Float16)w[3], (_Float16)w[3]};
#pragma unroll
  for (int e = 0; e < 4; e++) {
    f16x2 a0 = {v[0][2 * e], v[0][2 * e + 1]}, a1 = {v[1][2 * e], v[1][2 * e + 1]};
    f16x2 a2 = {v[2][2 * e], v[2][2 * e + 1]}, a3 = {v[3][2 * e], v[3][2 * e + 1]};
    // explicit FMAs: with plain `w1 * a1 + acc` the compiler is free to fuse EITHER product of a sum of two products
    // into the add, and picks differently between instantiations (1-ulp differences between tile shapes)
    f16x2 acc = w0 * a0;
    acc = __builtin_elementwise_fma(w1, a1, acc);
    acc = __builtin_elementwise_fma(w2, a2, acc);
    acc = __builtin_elementwise_fma(w3, a3, acc);
    r[2 * e] = acc[0];
    r[2 * e + 1] = acc[1];
  }
  return r;
}

// the same blend with the four weights still in binary16 (the sampling table stores them so): no conversions
__device__ __forceinline__ f16x8 blend_pk_h(const f16x8 (&v)[4], const _Float16 (&w)[4]) {
  f16x8 r;
  const f16x2 w0 = {w[0], w[0]}, w1 = {w[1], w[1]}, w2 = {w[2], w[2]}, w3 = {w[3], w[3]};
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const f16x2 a0 = {v[0][2 * e], v[0][2 * e + 1]}, a1 = {v[1][2 * e], v[1][2 * e + 1]};
    const f16x2 a2 = {v[2][2 * e], v[2][2 * e + 1]}, a3 = {v[3][2 * e], v[3][2 * e + 1]};
    f16x2 acc = w0 * a0;
    acc = __builtin_elementwise_fma(w1, a1, acc);
    acc = __builtin_elementwise_fma(w2, a2, acc);
    acc = __builtin_elementwise_fma(w3, a3, acc);
    r[2 * e] = acc[0];
    r[2 * e + 1] = acc[1];
  }
  return r;
}

// XCD-aware tile order: blockIdx round-robins over the 8 XCDs (private L2 each); give every XCD
// a contiguous run of position tiles so neighbouring tiles (which sample overlapping input rows)
// share one L2.  Bijective for any tile count (cdna guide T1).
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned n) {
  const unsigned q = n / 8, r = n % 8, x = bid % 8;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + bid / 8;
}

// Pyramid-packed launches: the FPN levels of one head layer share their filters, so all of them go
// through ONE launch.  The levels sit back to back in one NHWC buffer (level l = [B,H_l,W_l,C] at
// pixel offset pix0[l]); a workgroup finds its level from the tile index and rebinds its pointers
// and geometry -- from there on it is an ordinary single-level tile.  n <= 1: plain tensor.
constexpr int kMaxLevels = 8;
struct LevelTab {
  int n, batch;
  int H[kMaxLevels], W[kMaxLevels], tile0[kMaxLevels], pix0[kMaxLevels];
  float stride[kMaxLevels];
};

// SRC: 0 = offset tensor [B,18,H,W] f32, 1 = refined anchors [B,H,W,5] f32
// NPOS: output positions per workgroup (64: more workgroups for small inputs; 128: the weight tile
// is amortised over twice the positions)
template <typename T, bool OUT_NHWC, int SRC, int NPOS>
__global__ __launch_bounds__(256, 1) void k_dcn_mfma(const T* __restrict__ x,       // NHWC
                                                     const float* __restrict__ src,  // offsets | anchors
                                                     const T* __restrict__ wp,       // packed weights
                                                     T* __restrict__ out, int64_t Ntot, int C, int H,
                                                     int W, int O, float stride, int relu) {
  constexpr int KC = Traits<T>::KC;
  constexpr int VEC = Traits<T>::VEC;
  constexpr int NT = NPOS / 32;    // 32-wide position tiles per wave
  constexpr int ITEMS = NPOS / 32; // (position, 16-byte channel group) items per thread
  using V = typename Vec16<T>::type;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // [ taps NPOS*9*32 B | A0 | B0 | A1 | B1 ]
  Tap* s_tab = reinterpret_cast<Tap*>(smem);
  constexpr int kTabBytes = NPOS * 9 * 32;
  constexpr int kABytes = kMaxO * kRowBytes, kBBytes = NPOS * kRowBytes;
  char* s_buf = smem + kTabBytes;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t HW = (int64_t)H * W;
  const int64_t tile = xcd_remap(blockIdx.x, gridDim.x);
  const int o0 = blockIdx.y * kMaxO;
  const int Oloc = min(kMaxO, O - o0);
  const int CC = C / KC;
  const int nstage = 9 * CC;

  // ---- sampling table for this tile
  for (int e = tid; e < NPOS * 9; e += 256) {
    int pl = e / 9, t = e % 9;
    int64_t g = tile_pos(tile, pl, NPOS / 16, H, W, HW, Ntot);
    Tap tp;
    if (g >= 0) {
      int64_t b = g / HW, p = g % HW;
      int y = (int)(p / W), xq = (int)(p % W);
      int ky = t / 3, kx = t % 3;
      float off_y, off_x;
      if (SRC == 0) {
        const float* ob = src + (b * 18) * HW + p;
        off_y = ob[(int64_t)(2 * t) * HW];
        off_x = ob[(int64_t)(2 * t + 1) * HW];
      } else {
        AnchorCtx c = anchor_ctx(src + g * 5, stride);
        anchor_offset(c, ky, kx, (float)y, (float)xq, off_y, off_x);
      }
      float h_im = (float)(y - 1 + ky) + off_y;  // kernel.cu:226-227
      float w_im = (float)(xq - 1 + kx) + off_x;
      tp = make_tap(h_im, w_im, H, W, b * HW);
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        tp.idx[k] = 0;
        tp.w[k] = 0.f;
      }
    }
    s_tab[e] = tp;
  }
  __syncthreads();

  f32x16 acc[2][NT];
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < NT; b++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

  // per-thread staging registers
  V cv[ITEMS][4];
  float cw[ITEMS][4];
  V av[8];

  auto issue = [&](int s) {
    const int t = s % 9, cc = s / 9;
#pragma unroll
    for (int it = 0; it < ITEMS; it++) {
      int item = tid + 256 * it;
      int pl = item >> 3, q = item & 7;
      const Tap tp = s_tab[pl * 9 + t];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        cw[it][k] = tp.w[k];
        cv[it][k] = *reinterpret_cast<const V*>(x + (int64_t)tp.idx[k] * C + cc * KC + q * VEC);
      }
    }
    const T* wsrc = wp + ((int64_t)s * O + o0) * KC;
#pragma unroll
    for (int r = 0; r < 8; r++) {
      int idx = tid + 256 * r;
      int row = idx >> 3, q = idx & 7;
      if (row < Oloc) av[r] = *reinterpret_cast<const V*>(wsrc + (int64_t)row * KC + q * VEC);
    }
  };
  auto commit = [&](int buf) {
    char* A = s_buf + buf * (kABytes + kBBytes);
    char* Bm = A + kABytes;
#pragma unroll
    for (int it = 0; it < ITEMS; it++) {
      int item = tid + 256 * it;
      int pl = item >> 3, q = item & 7;
      if constexpr (sizeof(T) == 2)
        *reinterpret_cast<V*>(Bm + pl * kRowBytes + q * 16) = blend_pk(cv[it], cw[it]);
      else
        *reinterpret_cast<V*>(Bm + pl * kRowBytes + q * 16) = blend<T>(cv[it], cw[it]);
    }
#pragma unroll
    for (int r = 0; r < 8; r++) {
      int idx = tid + 256 * r;
      int row = idx >> 3, q = idx & 7;
      if (row < Oloc) *reinterpret_cast<V*>(A + row * kRowBytes + q * 16) = av[r];
    }
  };

  issue(0);
  commit(0);
  __syncthreads();

  const bool wave_active = wave * 64 < Oloc;
  for (int s = 0; s < nstage; s++) {
    const int buf = s & 1;
    if (s + 1 < nstage) issue(s + 1);
    if (wave_active) {
      const char* A = s_buf + buf * (kABytes + kBBytes);
      const char* Bm = A + kABytes;
      const char* wrow = A + (wave * 64 + (lane & 31)) * kRowBytes + (lane >> 5) * 16;
      const char* prow = Bm + (lane & 31) * kRowBytes + (lane >> 5) * 16;
#pragma unroll
      for (int kk = 0; kk < 4; kk++) {
        V wf[2], pf[NT];
#pragma unroll
        for (int h = 0; h < 2; h++) wf[h] = *reinterpret_cast<const V*>(wrow + h * 32 * kRowBytes + kk * 32);
#pragma unroll
        for (int h = 0; h < NT; h++) pf[h] = *reinterpret_cast<const V*>(prow + h * 32 * kRowBytes + kk * 32);
        if constexpr (sizeof(T) == 4) {
#pragma unroll
          for (int j = 0; j < 4; j++)
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
              for (int b = 0; b < NT; b++) {
                if constexpr (OUT_NHWC)
                  acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(pf[b][j], wf[a][j], acc[a][b], 0, 0, 0);
                else
                  acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[a][j], pf[b][j], acc[a][b], 0, 0, 0);
              }
        } else {
#pragma unroll
          for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < NT; b++) {
              if constexpr (OUT_NHWC)
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pf[b], wf[a], acc[a][b], 0, 0, 0);
              else
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[a], pf[b], acc[a][b], 0, 0, 0);
            }
        }
      }
    }
    if (s + 1 < nstage) commit(buf ^ 1);
    __syncthreads();
  }

  if (!wave_active) return;
  // ---- epilogue: ReLU + store.  acc[a][b] = (out-channel tile a, position tile b);
  // MFMA D layout: column = lane&31, row = (r&3)+8*(r>>2)+4*(lane>>5)
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < NT; b++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        float v = acc[a][b][r];
        if (relu) v = fmaxf(v, 0.f);
        int rowi = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if constexpr (OUT_NHWC) {
          // operands swapped: rows = positions, columns = out channels (128 B runs along channels)
          int64_t g = tile_pos(tile, 32 * b + rowi, NPOS / 16, H, W, HW, Ntot);
          int och = o0 + wave * 64 + 32 * a + (lane & 31);
          if (g >= 0) out[g * O + och] = (T)v;
        } else {
          int och = o0 + wave * 64 + 32 * a + rowi;
          int64_t g = tile_pos(tile, 32 * b + (lane & 31), NPOS / 16, H, W, HW, Ntot);
          if (g >= 0) {
            int64_t bi = g / HW, p = g % HW;
            out[(bi * O + och) * HW + p] = (T)v;
          }
        }
      }
}

// ------------------------------------------------------------------ wave-specialised variant
// 512 threads: waves 0-3 only issue LDS fragment reads + MFMAs, waves 4-7 only gather, blend and
// write LDS (one MFMA wave and one loader wave share each SIMD).  The loaders keep TWO stages of
// global loads in flight (two register sets, counted vmcnt by the compiler), so memory stays busy
// across the blend / LDS-write / barrier phases that stall the single-role kernel above.
// Loads use buffer addressing (32-bit voffset + scalar stage offset): no per-load 64-bit VALU math.
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

template <typename T, bool OUT_NHWC, int SRC>
__global__ __launch_bounds__(512, 2) void k_dcn_ws(const T* __restrict__ x, const float* __restrict__ src,
                                                   const T* __restrict__ wp, T* __restrict__ out,
                                                   int64_t Ntot, int C, int H, int W, int O, float stride,
                                                   int relu, unsigned x_bytes, unsigned wp_bytes) {
  constexpr int KC = Traits<T>::KC;
  constexpr int NPOS = 128, NT = 4, ITEMS = 4;
  constexpr int ES = (int)sizeof(T);
  using V = typename Vec16<T>::type;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  Tap* s_tab = reinterpret_cast<Tap*>(smem);
  constexpr int kTabBytes = NPOS * 9 * 32;
  constexpr int kABytes = kMaxO * kRowBytes, kBBytes = NPOS * kRowBytes;
  char* s_buf = smem + kTabBytes;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t HW = (int64_t)H * W;
  const int64_t tile = xcd_remap(blockIdx.x, gridDim.x);
  const int o0 = blockIdx.y * kMaxO;
  const int Oloc = min(kMaxO, O - o0);
  const int CC = C / KC;
  const int nstage = 9 * CC;

  for (int e = tid; e < NPOS * 9; e += 512) {
    int pl = e / 9, t = e % 9;
    int64_t g = tile_pos(tile, pl, NPOS / 16, H, W, HW, Ntot);
    Tap tp;
    if (g >= 0) {
      int64_t b = g / HW, p = g % HW;
      int y = (int)(p / W), xq = (int)(p % W);
      int ky = t / 3, kx = t % 3;
      float off_y, off_x;
      if (SRC == 0) {
        const float* ob = src + (b * 18) * HW + p;
        off_y = ob[(int64_t)(2 * t) * HW];
        off_x = ob[(int64_t)(2 * t + 1) * HW];
      } else {
        AnchorCtx c = anchor_ctx(src + g * 5, stride);
        anchor_offset(c, ky, kx, (float)y, (float)xq, off_y, off_x);
      }
      float h_im = (float)(y - 1 + ky) + off_y;
      float w_im = (float)(xq - 1 + kx) + off_x;
      tp = make_tap(h_im, w_im, H, W, b * HW);
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        tp.idx[k] = 0;
        tp.w[k] = 0.f;
      }
    }
    s_tab[e] = tp;
  }
  __syncthreads();

  if (wave < 4) {
    // ===================== MFMA waves =====================
    f32x16 acc[2][NT];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int b = 0; b < NT; b++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;
    const bool wave_active = wave * 64 < Oloc;
    __syncthreads();  // stage 0 is in LDS
    for (int s = 0; s < nstage; s++) {
      if (wave_active) {
        const char* A = s_buf + (s & 1) * (kABytes + kBBytes);
        const char* Bm = A + kABytes;
        const char* wrow = A + (wave * 64 + (lane & 31)) * kRowBytes + (lane >> 5) * 16;
        const char* prow = Bm + (lane & 31) * kRowBytes + (lane >> 5) * 16;
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
          V wf[2], pf[NT];
#pragma unroll
          for (int h = 0; h < 2; h++) wf[h] = *reinterpret_cast<const V*>(wrow + h * 32 * kRowBytes + kk * 32);
#pragma unroll
          for (int h = 0; h < NT; h++) pf[h] = *reinterpret_cast<const V*>(prow + h * 32 * kRowBytes + kk * 32);
          if constexpr (sizeof(T) == 4) {
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
              for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < NT; b++) {
                  if constexpr (OUT_NHWC)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(pf[b][j], wf[a][j], acc[a][b], 0, 0, 0);
                  else
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[a][j], pf[b][j], acc[a][b], 0, 0, 0);
                }
          } else {
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
              for (int b = 0; b < NT; b++) {
                if constexpr (OUT_NHWC)
                  acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pf[b], wf[a], acc[a][b], 0, 0, 0);
                else
                  acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[a], pf[b], acc[a][b], 0, 0, 0);
              }
          }
        }
      }
      __syncthreads();
    }
    if (!wave_active) return;
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int b = 0; b < NT; b++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          float v = acc[a][b][r];
          if (relu) v = fmaxf(v, 0.f);
          int rowi = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if constexpr (OUT_NHWC) {
            int64_t g = tile_pos(tile, 32 * b + rowi, NPOS / 16, H, W, HW, Ntot);
            int och = o0 + wave * 64 + 32 * a + (lane & 31);
            if (g >= 0) out[g * O + och] = (T)v;
          } else {
            int och = o0 + wave * 64 + 32 * a + rowi;
            int64_t g = tile_pos(tile, 32 * b + (lane & 31), NPOS / 16, H, W, HW, Ntot);
            if (g >= 0) {
              int64_t bi = g / HW, p = g % HW;
              out[(bi * O + och) * HW + p] = (T)v;
            }
          }
        }
  } else {
    // ===================== loader waves =====================
    const int L = tid - 256;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(x), 0, (int)x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(wp), 0, (int)wp_bytes, 0x00020000);
    unsigned voff[ITEMS][4];
    V cvA[ITEMS][4], cvB[ITEMS][4], avA[8], avB[8];
    const unsigned row_bytes = (unsigned)C * ES;
    const unsigned wvoff = (unsigned)L * 16;   // + r*4096 via the scalar offset

    auto issue = [&](int s, V (&cv)[ITEMS][4], V (&av)[8]) {
      const int t = s % 9, cc = s / 9;
      {  // every stage is a new tap: refresh the 16 corner offsets of this thread's items
#pragma unroll
        for (int it = 0; it < ITEMS; it++) {
          int item = L + 256 * it;
          int pl = item >> 3, q = item & 7;
          const Tap tp = s_tab[pl * 9 + t];
#pragma unroll
          for (int k = 0; k < 4; k++) voff[it][k] = (unsigned)tp.idx[k] * row_bytes + q * 16;
        }
      }
      const int soff = cc * KC * ES;
#pragma unroll
      for (int it = 0; it < ITEMS; it++)
#pragma unroll
        for (int k = 0; k < 4; k++) {
          u32x4 d = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)voff[it][k], soff, 0);
          cv[it][k] = __builtin_bit_cast(V, d);
        }
      const unsigned wbase = (unsigned)(((int64_t)s * O + o0) * KC * ES);
      // rows >= Oloc read past this workgroup's weight block: harmless (bounds-checked buffer load,
      // the LDS tile always has 256 rows, inactive MFMA waves never read them).  No predicate here:
      // a branch around loads makes hipcc fall back from counted vmcnt(N) to vmcnt(0).
#pragma unroll
      for (int r = 0; r < 8; r++) {
        u32x4 d = __builtin_amdgcn_raw_buffer_load_b128(rw, (int)wvoff, (int)(wbase + r * 4096), 0);
        av[r] = __builtin_bit_cast(V, d);
      }
    };
    auto commit = [&](int s, V (&cv)[ITEMS][4], V (&av)[8]) {
      const int t = s % 9;
      char* A = s_buf + (s & 1) * (kABytes + kBBytes);
      char* Bm = A + kABytes;
#pragma unroll
      for (int it = 0; it < ITEMS; it++) {
        int item = L + 256 * it;
        int pl = item >> 3, q = item & 7;
        const f32x4 w4 = *reinterpret_cast<const f32x4*>(&s_tab[pl * 9 + t].w[0]);
        const float cw[4] = {w4[0], w4[1], w4[2], w4[3]};
        if constexpr (sizeof(T) == 2)
          *reinterpret_cast<V*>(Bm + pl * kRowBytes + q * 16) = blend_pk(cv[it], cw);
        else
          *reinterpret_cast<V*>(Bm + pl * kRowBytes + q * 16) = blend<T>(cv[it], cw);
      }
#pragma unroll
      for (int r = 0; r < 8; r++) {
        int idx = L + 256 * r;
        int row = idx >> 3, q = idx & 7;
        *reinterpret_cast<V*>(A + row * kRowBytes + q * 16) = av[r];
      }
    };

    // Straight-line double-stage body (stage indices clamped instead of branches) so that the
    // compiler keeps exact vmcnt counts: when a set is committed, the other set's 24 loads stay
    // in flight.  Loop-entry and back-edge state are identical: "24 loads of set B outstanding".
    const int last = nstage - 1;
    issue(0, cvA, avA);
    issue(min(1, last), cvB, avB);
    commit(0, cvA, avA);
    __syncthreads();  // stage 0 is in LDS
    int s = 0;
    for (; s + 1 < nstage; s += 2) {
      issue(min(s + 2, last), cvA, avA);
      commit(s + 1, cvB, avB);
      __syncthreads();              // end of stage s
      issue(min(s + 3, last), cvB, avB);
      commit(min(s + 2, last), cvA, avA);   // when s+2 > last: rewrites the unread buffer, harmless
      __syncthreads();              // end of stage s+1
    }
    if (s < nstage) __syncthreads();  // odd stage count: the last stage's barrier
  }
}

// ------------------------------------------------------------------ patch-staged variant (f16)
// The vector-memory path of a CU sustains only ~30 B/clk for 16-B-per-lane gathers (measured: the
// loader side of k_dcn_ws takes 3.3 k cycles per stage even with every load dropped by a
// zero-record descriptor), and the 9 taps of one position re-read almost the same pixels.  So:
//   * the input patch around the 8x16 position tile (16x24 pixels x 64 channels = 48 KB) is
//     loaded ONCE per channel chunk into LDS; the 9 taps gather their corners from LDS
//     (ds_read_b128, ~10x the bandwidth of the global gather); corners outside the patch fall
//     back to global loads (rare for anchor-sized offsets)
//   * the weight tile never touches LDS: it is pre-packed in MFMA-fragment order and every MFMA
//     wave loads its own 8 KB per stage straight into registers (1 KB contiguous per instruction),
//     one stage ahead — which frees 72 KB of LDS for the double-buffered patch
// L1 traffic per stage drops from 96 KB to ~38 KB.
struct alignas(16) PTap {
  short y, x;        // top-left bilinear corner (h_low, w_low), image coordinates
  unsigned flags;    // bit 0: all four corners lie inside the LDS patch
  _Float16 w[4];     // hh*hw, hh*lw, lh*hw, lh*lw; 0 where the corner is dropped
};
constexpr int kPH = 16, kPW = 24, kHalo = 4;
constexpr int kPatchBytes = kPH * kPW * 128;
constexpr int kPatchLds = 128 * 9 * 16 + 2 * 128 * kRowBytes + 2 * kPatchBytes;  // 153600 B

// weight [O][C][9] f16 -> [stage = cc*9+t][och group of 64][mt 2][kk 4][lane 64][8 halfs]:
// lane l, element j of fragment (mt,kk) = W[g*64 + mt*32 + (l&31)][cc*64 + kk*16 + 8*(l>>5) + j][t]
__global__ void k_pack_weight_frag(const _Float16* __restrict__ w, int O, int C, _Float16* __restrict__ wp,
                                   int taps = 9) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t total = (int64_t)O * C * taps;
  if (e >= total) return;
  const int G = O / 64;
  int j = (int)(e & 7);
  int lane = (int)((e >> 3) & 63);
  int kk = (int)((e >> 9) & 3);
  int mt = (int)((e >> 11) & 1);
  int64_t r = e >> 12;
  int g = (int)(r % G);
  int st = (int)(r / G);
  int t = st % taps, cc = st / taps;
  int och = g * 64 + mt * 32 + (lane & 31);
  int k = cc * 64 + kk * 16 + 8 * (lane >> 5) + j;
  wp[e] = w[((int64_t)och * C + k) * taps + t];
}

// the same filter for the 16x16x32 matrix waves of k_dcn_patch: [stage][och group of 64][fragment f = 2a + ks (8)][lane 64][8 halfs],
// lane = (i = lane & 15, kg = lane >> 4): element j = W[g*64 + och_l(a, i)][cc*64 + 32 (kg & 1) + 16 ks + 8 (kg >> 1) + j][t] with
// och_l(a, i) = 32 (a >> 1) + 8 (i >> 2) + 4 (a & 1) + (i & 3).  One contiguous 1 KB load per fragment, and the D rows of a lane
// (4 kg + e of fragments a = 0..3) are out channels 8 kg .. 8 kg + 7 and 32 + 8 kg .. 32 + 8 kg + 7 of its pixel: the epilogue
// stores two 16-byte vectors per pixel straight from the accumulators (no LDS staging, no barrier)
__device__ __forceinline__ int frag16_och(int a, int i) { return 32 * (a >> 1) + 8 * (i >> 2) + 4 * (a & 1) + (i & 3); }
__global__ void k_pack_weight_frag16(const _Float16* __restrict__ w, int O, int C, _Float16* __restrict__ wp) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t total = (int64_t)O * C * 9;
  if (e >= total) return;
  const int G = O / 64;
  int j = (int)(e & 7);
  int lane = (int)((e >> 3) & 63);
  int f = (int)((e >> 9) & 7);
  int64_t r = e >> 12;
  int g = (int)(r % G);
  int st = (int)(r / G);
  int t = st % 9, cc = st / 9;
  int och = g * 64 + frag16_och(f >> 1, lane & 15);
  int k = cc * 64 + 32 * ((lane >> 4) & 1) + 16 * (f & 1) + 8 * (lane >> 5) + j;   // the k slots of the column fragments (bfrag)
  wp[e] = w[((int64_t)och * C + k) * 9 + t];
}

// TH = rows of the position tile: 8 (128 positions), or 4 (64 positions: half tiles for launches that fill the chip badly,
// e.g. one P3 level of one chip = 128 full tiles on 256 CUs; tile index = tile_base + block / 2, half = block & 1)
// (MW = 4 matrix waves, one per SIMD, 64 out channels x all positions each.  The form with EIGHT -- two per SIMD, 32 out channels
// each, twelve waves of <= 168 registers, the later patch chunks in two halves -- passed the same tests and measured 10 % slower,
// 241.7 against 219.8 us: DESIGN 4)
template <bool OUT_NHWC, int SRC, int TH = 8>
__device__ __forceinline__ void dcn_patch_tile(const _Float16* __restrict__ x_,
                                               const float* __restrict__ src_,
                                               const _Float16* __restrict__ wfrag,
                                               _Float16* __restrict__ out_, int64_t Ntot_, int C,
                                               int H_, int W_, int O, float stride_, int relu,
                                               unsigned x_bytes_, const LevelTab& lt, int tile_base, unsigned vb, unsigned nvb, const int tid) {
  using T = _Float16;
  using V = f16x8;
  constexpr int NPOS = TH * 16, NT = NPOS / 32, ITEMS = NPOS / 32;
  constexpr int kPHt = TH + 2 * kHalo;                 // patch rows
  constexpr int kPatchBytesT = kPHt * kPW * 128;
  constexpr int NPV = kPHt * kPW * 8 / 256;            // 16-byte patch vectors per loader thread (12 | 9)
  static_assert(TH == 8 || TH == 4, "tile height");
  constexpr int MW = 4, NTHR = 64 * (MW + 4);          // matrix waves (0 .. MW-1; the loaders are waves MW .. MW+3), threads
  constexpr int AH = 16 / MW;                          // 16-channel accumulator tiles per matrix wave
  static_assert(kPHt * kPW * 8 % 256 == 0, "patch vectors must divide among the loader threads");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  PTap* s_tab = reinterpret_cast<PTap*>(smem);
  char* s_B = smem + NPOS * 9 * 16;
  char* s_patch = s_B + 2 * NPOS * kRowBytes;

  const bool half_coords = (relu & 2) != 0;          // relu: bit 0 = ReLU epilogue, bit 1 = S2A_DCN_HALF_COORDS
  auto rh16 = [](float v) { return (float)(_Float16)v; };
  const int lane = tid & 63, wave = tid >> 6;                      // (waves 0-3 = matrix: the older half wins issue
                                                                   // arbitration; roles swapped measured 4 % slower)
  int64_t tile = xcd_remap(vb, nvb);
  const int half = TH == 4 ? (int)(tile & 1) : 0;
  if (TH == 4) tile >>= 1;
  tile += tile_base;
  const T* x = x_;
  const float* src = src_;
  T* out = out_;
  int64_t Ntot = Ntot_;
  int H = H_, W = W_;
  float stride = stride_;
  unsigned x_bytes = x_bytes_;
  if (SRC == 1 && lt.n > 1) {         // pyramid-packed levels (anchors [sum B*H*W, 5] packed alike)
    int t0 = 0, p0 = 0;
#pragma unroll
    for (int i = 0; i < kMaxLevels; i++)
      if (i < lt.n && tile >= lt.tile0[i]) {
        t0 = lt.tile0[i]; p0 = lt.pix0[i]; H = lt.H[i]; W = lt.W[i]; stride = lt.stride[i];
      }
    tile -= t0;
    Ntot = (int64_t)lt.batch * H * W;
    x += (int64_t)p0 * C;
    out += (int64_t)p0 * O;
    src += (int64_t)p0 * 5;
    x_bytes = (unsigned)(Ntot * C * 2);
  }
  const int64_t HW = (int64_t)H * W;
  const int txn = (W + 15) / 16, tyn = (H + 7) / 8;
  const int64_t bimg = tile / (txn * tyn);
  const int trem = (int)(tile % (txn * tyn));
  const int ty0 = (trem / txn) * 8 + 4 * half, tx0 = (trem % txn) * 16;
  const int oy = ty0 - kHalo, ox = tx0 - kHalo;
  const int o0 = blockIdx.y * kMaxO;
  const int Oloc = min(kMaxO, O - o0);
  const int CC = C / 64;
  const int nstage = 9 * CC;
  const int G = O / 64;
  const unsigned row_bytes = (unsigned)C * 2;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(x), 0, (int)x_bytes, 0x00020000);

  S2A_STAMP_AT(0);
  S2A_STAMP_VAL(6, __builtin_amdgcn_s_memrealtime());   // (matrix wave; 100 MHz: the in-kernel clock is d memtime / d memrealtime)
  // ---- loader waves: put the first patch in flight before anything else (its latency hides
  // under the table build).  patch element v = L + 256*i: pixel v>>3, 16-byte channel group v&7;
  // out-of-image pixels get an out-of-range offset -> the bounds-checked load returns zeros.
  const int L = tid - 64 * MW;
  auto patch_off = [&](int i) -> unsigned {
    int v = L + 256 * i, p = v >> 3, q = v & 7;
    int yy = oy + p / kPW, xx = ox + p % kPW;
    bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;
    return in ? (unsigned)((bimg * HW + (int64_t)yy * W + xx) * row_bytes + q * 16) : 0x80000000u;
  };
  unsigned pvoff[NPV];
  V pv[NPV];
  if (wave >= MW) {
#pragma unroll
    for (int i = 0; i < NPV; i++) {
      const unsigned o = patch_off(i);
      pvoff[i] = o;
      u32x4 d = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)o, 0, 0);
      pv[i] = __builtin_bit_cast(V, d);
    }
  }

  // ---- MFMA waves: first weight fragments in flight now as well (one memory latency for the
  // whole prologue instead of three in a row)
  const int g = min(o0 / 64 + (wave & 3), G - 1);
  const V* wf_base = reinterpret_cast<const V*>(wfrag) + lane;
  constexpr int WR = 2;                                // filter fragments per matrix wave and stage: 4 WR
  V wA[WR][4], wB[WR][4];
  // 16x16x32 MFMAs (lane maps and the reason: k_conv_f16; the 32x32x16 form of rounds 1-2 measured +1.7 %).
  // Lane = (i = lane & 15, kg = lane >> 4); fragment f = (16-channel tile f >> 1, k-step f & 1) in k_pack_weight_frag16
  // order; 16-position tile pt = tile row pt, its pixel for lane i = pix16 (conflict-free ds_read_b128 on the 144-byte
  // rows of the column tile)
  const int kg16 = lane >> 4, i16 = lane & 15;
  const int pix16 = (i16 >= 4 && i16 < 12) ? (((i16 - 4) >> 1) * 4 + (i16 & 1))
                                           : (((i16 & 3) >> 1) * 4 + 2 + (i16 & 1) + (i16 >= 12 ? 8 : 0));
  auto load_w = [&](int s, V (&wv)[WR][4]) {
    const V* p = wf_base + ((int64_t)s * G + g) * 8 * 64;        // k_pack_weight_frag16 order: 1 KB per fragment
#pragma unroll
    for (int f = 0; f < 4 * WR; f++) wv[f >> 2][f & 3] = p[f * 64];
  };
  if (wave < MW) load_w(0, wA);

  // ---- one table entry: tap t of tile position pl, anchor context c (SRC 1) or the offset maps (SRC 0)
  auto table_entry = [&](int pl, int t, const AnchorCtx& c) {
    int y = ty0 + (pl >> 4), xq = tx0 + (pl & 15);
    PTap tp;
    tp.y = (short)oy;
    tp.x = (short)ox;
    tp.flags = 1u;
#pragma unroll
    for (int k = 0; k < 4; k++) tp.w[k] = (_Float16)0.f;
    if (y < H && xq < W) {
      const int64_t p = (int64_t)y * W + xq;
      int ky = t / 3, kx = t % 3;
      float off_y, off_x;
      if (SRC == 0) {
        const float* ob = src + (bimg * 18) * HW + p;
        off_y = ob[(int64_t)(2 * t) * HW];
        off_x = ob[(int64_t)(2 * t + 1) * HW];
      } else {
        anchor_offset(c, ky, kx, (float)y, (float)xq, off_y, off_x);
      }
      // half_coords: the reference's scalar_t = Half instantiation (deform_conv.py:45-46 casts the offsets to half;
      // deform_conv_cuda_kernel.cu:221-228 h_im / w_im, :97-109 lh / lw / hh / hw and the four weights are Half
      // results): every one of them rounded to binary16.  Default: f32 coordinates.  ONE uniform branch per
      // entry: written as a rounding after each step the compiler computed both forms and selected per value
      // (30 of the entry's 150 instructions for a mode that is off)
      float h_im, w_im, lh = 0.f, lw = 0.f, hh = 0.f, hw = 0.f;
      int h_low = 0, w_low = 0;
      bool inside;
      if (!half_coords) {
        h_im = (float)(y - 1 + ky) + off_y;
        w_im = (float)(xq - 1 + kx) + off_x;
        inside = h_im > -1 && w_im > -1 && h_im < H && w_im < W;
        if (inside) {
          h_low = (int)floorf(h_im); w_low = (int)floorf(w_im);
          lh = h_im - h_low; lw = w_im - w_low;
          hh = 1 - lh; hw = 1 - lw;
        }
      } else {
        off_y = rh16(off_y); off_x = rh16(off_x);
        h_im = rh16((float)(y - 1 + ky) + off_y);
        w_im = rh16((float)(xq - 1 + kx) + off_x);
        inside = h_im > -1 && w_im > -1 && h_im < H && w_im < W;
        if (inside) {
          h_low = (int)floorf(h_im); w_low = (int)floorf(w_im);
          lh = rh16(h_im - h_low); lw = rh16(w_im - w_low);
          hh = rh16(1 - lh); hw = rh16(1 - lw);
        }
      }
      if (inside) {
        bool t_ok = h_low >= 0, b_ok = h_low + 1 <= H - 1, l_ok = w_low >= 0, r_ok = w_low + 1 <= W - 1;
        tp.w[0] = (_Float16)((t_ok && l_ok) ? hh * hw : 0.f);
        tp.w[1] = (_Float16)((t_ok && r_ok) ? hh * lw : 0.f);
        tp.w[2] = (_Float16)((b_ok && l_ok) ? lh * hw : 0.f);
        tp.w[3] = (_Float16)((b_ok && r_ok) ? lh * lw : 0.f);
        tp.y = (short)h_low;
        tp.x = (short)w_low;
        bool in = h_low >= oy && h_low + 1 <= oy + kPHt - 1 && w_low >= ox && w_low + 1 <= ox + kPW - 1;
        // bits 31..1: byte offset of the top-left corner inside a patch buffer (clamped for corners that left the
        // patch: those items are redone from global memory) -- the loader adds it instead of re-deriving it per stage
        const int py = min(max(h_low - oy, 0), kPHt - 2), px = min(max(w_low - ox, 0), kPW - 2);
        tp.flags = (in ? 1u : 0u) | ((unsigned)((py * kPW + px) * 128) << 1);
      }
    }
    s_tab[pl * 9 + t] = tp;
  };
  // ---- per-position anchor context (cos/sin once per position, not once per tap).  It lives in the SECOND column buffer:
  // the first one is written (first column tile) while the matrix waves still read contexts for the rest of the table
  AnchorCtx* s_ctx = reinterpret_cast<AnchorCtx*>(s_B + NPOS * kRowBytes);
  static_assert(NPOS * sizeof(AnchorCtx) <= NPOS * kRowBytes, "contexts fit a column buffer");
  if (SRC == 1 && tid < NPOS) {
    int y = ty0 + (tid >> 4), xq = tx0 + (tid & 15);
    AnchorCtx c = {0, 0, 0, 0, 1, 0};
    if (y < H && xq < W) c = anchor_ctx(src + (bimg * HW + (int64_t)y * W + xq) * 5, stride);
    s_ctx[tid] = c;
  }
  if (SRC == 1) __syncthreads();

  // ---- sampling table, TAP-major (entry e = tap e / NPOS of position e % NPOS): everyone builds the first NTHR entries
  // (taps 0 .. 3 of a full tile), which is all the loaders need for their first column tiles; the matrix waves, idle until
  // the first columns exist, build the rest under the loaders' first tile (the whole table in front of barrier #1 was
  // 2.85 k cycles with the matrix pipes and then the loaders waiting on each other: 9.9 k cycles before the first MFMA)
  auto table_entries = [&](int e0, int e1, int t0, int nthr) {
    for (int e = e0 + t0; e < e1; e += nthr) {
      const int pl = e % NPOS, t = e / NPOS;
      AnchorCtx c = {0, 0, 0, 0, 1, 0};
      if (SRC == 1) c = s_ctx[pl];
      table_entry(pl, t, c);
    }
  };
  constexpr int kTabFirst = NTHR < NPOS * 9 ? NTHR : NPOS * 9;        // 512 entries = taps 0 .. 3 (TH 8), 0 .. 7 (TH 4)
  static_assert(kTabFirst % NPOS == 0 && kTabFirst / NPOS >= 2, "the first round of entries covers whole taps, at least taps 0 and 1");
  table_entries(0, kTabFirst, tid, NTHR);
  if (wave >= MW) {  // first patch -> LDS
#pragma unroll
    for (int i = 0; i < NPV; i++) *reinterpret_cast<V*>(s_patch + (L + 256 * i) * 16) = pv[i];
  }
  S2A_STAMP_AT(1);
  __syncthreads();  // #1 table rows of the first taps + patch 0 ready
  S2A_STAMP_AT(2);
  if (wave < MW) table_entries(kTabFirst, NPOS * 9, tid, 64 * MW);   // (s_ctx is dead behind barrier #2)

  f32x4 acc16[AH][2 * NT];
  const bool wave_active = wave < MW && (wave & 3) * 64 < Oloc;
  if (wave < MW) {
    // ===================== MFMA waves =====================
#pragma unroll
    for (int a = 0; a < AH; a++)
#pragma unroll
      for (int b = 0; b < 2 * NT; b++)
#pragma unroll
        for (int r = 0; r < 4; r++) acc16[a][b][r] = 0.f;
    const int last = nstage - 1;
    __syncthreads();  // #2 stage 0 columns in LDS
    S2A_STAMP_AT(3);
    int s = 0;
    unsigned long long t_tic = 0, t_work = 0, t_wait = 0;
    (void)t_tic; (void)t_work; (void)t_wait;
    // B fragments one k-step AHEAD of the MFMAs that use them (two register sets), also across the stage boundary: the
    // stage's barrier sits in front of the LAST k-step, after all four reads of the current tile have been issued and
    // have landed -- behind it the next tile is sealed and its first fragments are requested under the last 2*NT MFMAs.
    // (The plain form read 2 k-steps and waited for them at once: two exposed LDS round trips per stage plus one behind
    // the barrier.)  Same MFMAs in the same order: bit-identical.
    V p0[NT], p1[NT];
    // (no branch on wave_active in here: a wave without out-channels computes on clamped filter fragments and drops the
    // result in the epilogue -- a branch would split the basic block and hipcc's waitcnt pass then waits for ALL LDS
    // reads at every join, the prefetched ones included)
    // step kk = (k-step kk >> 1 of 32 channels, half kk & 1 of the wave's 16-position tiles): NT reads and 4 NT MFMAs of
    // 16 cycles per step
    auto bfrag = [&](int st, int kk, V (&pf)[NT]) {
      if ((S2A_ABL & 128) && st > 0) return;           // timing only: the fragments of stage 0 serve every stage
      const char* prow = s_B + (st & 1) * (NPOS * kRowBytes) + ((kk & 1) * NT * 16 + pix16) * kRowBytes + (kg16 & 1) * 64 + (kg16 >> 1) * 16 + (kk >> 1) * 32;
#pragma unroll
      for (int h = 0; h < NT; h++) pf[h] = *reinterpret_cast<const V*>(prow + h * 16 * kRowBytes);
    };
    auto mma = [&](const V (&wv)[WR][4], int kk, const V (&pf)[NT]) {
      if (S2A_ABL & 4) return;
      const int ks = kk >> 1, bh = (kk & 1) * NT;
#pragma unroll
      for (int a = 0; a < AH; a++)
#pragma unroll
        for (int b = 0; b < NT; b++)
          acc16[a][bh + b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv[(a * 2 + ks) >> 2][(a * 2 + ks) & 3], pf[b], acc16[a][bh + b], 0, 0, 0);
    };
    auto stage = [&](int st, const V (&wv)[WR][4]) {
      bfrag(st, 1, p1); __builtin_amdgcn_sched_barrier(0);
      mma(wv, 0, p0);   __builtin_amdgcn_sched_barrier(0);
      bfrag(st, 2, p0); __builtin_amdgcn_sched_barrier(0);
      mma(wv, 1, p1);   __builtin_amdgcn_sched_barrier(0);
      bfrag(st, 3, p1); __builtin_amdgcn_sched_barrier(0);
      mma(wv, 2, p0);   __builtin_amdgcn_sched_barrier(0);
      if (!(S2A_ABL & 256)) __syncthreads();  // tile st+1 sealed; this tile's buffer is free for the loaders (all its reads have landed)
      bfrag(min(st + 1, last), 0, p0); __builtin_amdgcn_sched_barrier(0);   // (last stage: a dead re-read)
      mma(wv, 3, p1);   __builtin_amdgcn_sched_barrier(0);
    };
    bfrag(0, 0, p0);
    if (S2A_ABL & 32) load_w(1, wB);     // timing only: the filter fragments of stages 0 / 1 serve every stage
    for (; s + 1 < nstage; s += 2) {
      // (64, timing only: the loads stay but always fetch stages 1 / 0 -- L1 hits: separates issue cost from L2 latency)
      if (!(S2A_ABL & 32)) load_w((S2A_ABL & 64) ? 1 : s + 1, wB);
      stage(s, wA);
      if (!(S2A_ABL & 32)) load_w((S2A_ABL & 64) ? 0 : min(s + 2, last), wA);
      stage(s + 1, wB);
    }
    if (s < nstage) stage(s, wA);
  } else {
    // ===================== loader waves =====================
    auto patch_issue = [&](int cc) {
#pragma unroll
      for (int i = 0; i < NPV; i++) {
        u32x4 d = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)pvoff[i], cc * 128, 0);
        pv[i] = __builtin_bit_cast(V, d);
      }
    };
    auto patch_write = [&](int cc) {
      char* P = s_patch + (cc & 1) * kPatchBytesT;
#pragma unroll
      for (int i = 0; i < NPV; i++) *reinterpret_cast<V*>(P + (L + 256 * i) * 16) = pv[i];
    };
    // The operands of a stage (4 table entries, 16 corner vectors per thread) are FETCHED one stage ahead -- right
    // after the previous stage's columns have been written, so their LDS latency (table read, then the dependent
    // corner reads) runs under the barrier wait instead of in front of the blend.
    PTap tp[ITEMS];
    V c[ITEMS][4];
    auto fetch = [&](int s) {
      const int t = s % 9, cc = s / 9;
      const char* P = s_patch + (cc & 1) * kPatchBytesT + (L & 7) * 16;   // (item & 7 = L & 7 for every item)
#pragma unroll
      for (int it = 0; it < ITEMS; it++)      // one 16-byte read per entry
        tp[it] = __builtin_bit_cast(PTap, *reinterpret_cast<const u32x4*>(&s_tab[((L + 256 * it) >> 3) * 9 + t]));
#pragma unroll
      for (int it = 0; it < ITEMS; it++) {
        // out-of-patch items read a clamped (wrong) location here and are redone in produce()
        const char* b0 = P + (tp[it].flags >> 1);
        c[it][0] = *reinterpret_cast<const V*>(b0);
        c[it][1] = *reinterpret_cast<const V*>(b0 + 128);
        c[it][2] = *reinterpret_cast<const V*>(b0 + kPW * 128);
        c[it][3] = *reinterpret_cast<const V*>(b0 + kPW * 128 + 128);
      }
    };
    auto produce = [&](int s) {  // columns of stage s (operands fetched before) -> B[s&1]
      if (S2A_ABL & 2) return;
      const int cc = s / 9;
      char* Bm = s_B + (s & 1) * (NPOS * kRowBytes);
      bool any_out = false;
#pragma unroll
      for (int it = 0; it < ITEMS; it++) any_out |= !(tp[it].flags & 1u);
#pragma unroll
      for (int it = 0; it < ITEMS; it++) {
        const int item = L + 256 * it, pl = item >> 3, q = item & 7;
        const float cw[4] = {(float)tp[it].w[0], (float)tp[it].w[1], (float)tp[it].w[2], (float)tp[it].w[3]};
        *reinterpret_cast<V*>(Bm + pl * kRowBytes + q * 16) = blend_pk(c[it], cw);
      }
      if (any_out) {  // rare: a corner left the patch -> global gather for that (position, tap)
        for (int it = 0; it < ITEMS; it++) {
          if (tp[it].flags & 1u) continue;
          const int item = L + 256 * it, pl = item >> 3, q = item & 7;
          V g4[4];
#pragma unroll
          for (int k = 0; k < 4; k++) {
            int yy = min(max((int)tp[it].y + (k >> 1), 0), H - 1), xx = min(max((int)tp[it].x + (k & 1), 0), W - 1);
            unsigned vo = (unsigned)((bimg * HW + (int64_t)yy * W + xx) * row_bytes + q * 16);
            u32x4 d = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)vo, cc * 128, 0);
            g4[k] = __builtin_bit_cast(V, d);
          }
          const float cw[4] = {(float)tp[it].w[0], (float)tp[it].w[1], (float)tp[it].w[2], (float)tp[it].w[3]};
          *reinterpret_cast<V*>(Bm + pl * kRowBytes + q * 16) = blend_pk(g4, cw);
        }
      }
    };

    if (CC > 1) patch_issue(1);
    fetch(0);
    produce(0);
    if (nstage > 1) fetch(1);
    S2A_STAMP_AT(6);
    __syncthreads();  // #2 stage 0 columns in LDS
    S2A_STAMP_AT(3);
    unsigned long long t_tic = 0, t_work = 0, t_wait = 0;
    (void)t_tic; (void)t_work; (void)t_wait;
    for (int s = 0; s < nstage; s++) {
      const int sn = s + 1;          // stage produced while stage s is consumed
      S2A_TIC();
      if (sn < nstage) {
        const int t = sn % 9, cc = sn / 9;
        if (t == 4 && cc + 1 < CC) patch_write(cc + 1);   // loads issued >= 3 stages ago
        produce(sn);
        if (t == 8 && cc + 2 < CC) patch_issue(cc + 2);   // next-next chunk: lands during the next chunk
        if (sn + 1 < nstage) fetch(sn + 1);               // (chunk of stage sn+1: written >= 4 stages ago)
      }
      S2A_TOC(t_work); S2A_TIC();
      if (!(S2A_ABL & 256)) __syncthreads();
      S2A_TOC(t_wait);
    }
    S2A_STAMP_VAL(7, t_work);
    S2A_STAMP_VAL(5, t_wait);   // (loader slot 5 = 13 overall; its end stamp is not used)
  }

  S2A_STAMP_AT(4);
  // ===================== epilogue =====================
  auto out_pos = [&](int pos) -> int64_t {       // linear position b*H*W + y*W + x of tile position pos, -1 outside
    const int y = ty0 + (pos >> 4), xq = tx0 + (pos & 15);
    const int64_t gpos = bimg * HW + (int64_t)y * W + xq;
    return (y < H && xq < W && gpos < Ntot) ? gpos : -1;
  };
  if ((S2A_ABL & 1) && relu != 12345) return;
  if constexpr (OUT_NHWC) {
    // straight from the accumulators (filter rows permuted by k_pack_weight_frag16): lane (pixel pix16 of 16-position tile b,
    // kg16) holds out channels 8 kg16 .. +7 (fragments 0, 1) and 32 + 8 kg16 .. +7 (fragments 2, 3) of its wave's 64 -- two
    // 16-byte stores per pixel, no LDS staging, no barrier (the LDS-staged whole-row form measured +1 %).
    // ReLU on the rounded halves, two per instruction (rounding is monotonic and keeps zero and sign: max(round(v), 0) ==
    // round(max(v, 0)); a NaN becomes 0 either way), and the ReLU switch as ONE uniform branch around the tile: with
    // fmaxf + a per-value select on the f32 accumulators the epilogue was 800 instructions, 5.3 k of a tile's 87 k cycles
    auto store_tile = [&](auto relu_c) {
      constexpr bool kRelu = decltype(relu_c)::value;
      using h2 = __attribute__((ext_vector_type(2))) _Float16;
      // tile position 16 b + pix16 = (row ty0 + b, column tx0 + pix16): one address, then a row pitch per b
      auto pack8 = [&](int b, int hf) {
        V v8;
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const f32x4& a4 = acc16[2 * hf + (i >> 1)][b];
          h2 v = {(_Float16)a4[2 * (i & 1)], (_Float16)a4[2 * (i & 1) + 1]};
          if constexpr (kRelu) v = __builtin_elementwise_max(v, h2{(_Float16)0.f, (_Float16)0.f});
          v8[2 * i] = v[0];
          v8[2 * i + 1] = v[1];
        }
        return v8;
      };
      const int xq = tx0 + pix16;
      int64_t gp = bimg * HW + (int64_t)ty0 * W + xq;
      _Float16* orow = out + o0 + (wave & 3) * 64 + 8 * kg16 + gp * O;
      const int64_t pitch = (int64_t)W * O;
#pragma unroll
      for (int b = 0; b < 2 * NT; b++, gp += W, orow += pitch) {
        const bool ok = xq < W && ty0 + b < H && gp < Ntot;
#pragma unroll
        for (int hf = 0; hf < AH / 2; hf++)
          if (ok) *reinterpret_cast<V*>(orow + 32 * hf) = pack8(b, hf);
      }
    };
    if (wave_active) {
      if (relu & 1) store_tile(std::true_type{}); else store_tile(std::false_type{});
    }
    S2A_STAMP_AT(5);
    if (wave == 0) S2A_STAMP_VAL(7, __builtin_amdgcn_s_memrealtime());
  } else {
    if (!wave_active) return;
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int b = 0; b < 2 * NT; b++)
#pragma unroll
        for (int e = 0; e < 4; e++) {
          float v = acc16[a][b][e];
          if (relu & 1) v = fmaxf(v, 0.f);
          const int och = o0 + wave * 64 + 32 * (a >> 1) + 8 * kg16 + 4 * (a & 1) + e;
          const int64_t gp = out_pos(16 * b + pix16);
          if (gp >= 0) {
            int64_t bi = gp / HW, p = gp % HW;
            out[(bi * O + och) * HW + p] = (T)v;
          }
        }
  }
}

// the launch.  n_full = 0: one TH-row tile per workgroup.  n_full > 0 (TH 8 only): workgroups [0, n_full) take full 8 x 16
// tiles, the workgroups behind them take the remaining tiles as 4 x 16 HALF tiles (two per tile) -- the hardware hands out
// workgroups in index order, so the halves are what runs last.  Why: the launch is rounds of one workgroup per CU, every round
// as long as a tile (36.7 us), and the last round of the detector's pyramid holds 84 tiles for 256 CUs: as 168 half tiles
// (0.62 of a tile each) it costs 0.62 rounds instead of one.  (A fixed grid of workgroups walking the tiles in a loop -- with
// the arguments re-read and the thread index laundered per tile so that nothing is hoisted across the body -- was bit-identical
// and 3 % SLOWER, 227 against 220 us: it keeps the same 6-tile critical path and adds a barrier per tile.)
struct PatchArgs {
  const _Float16* x;
  const float* src;
  const _Float16* wfrag;
  _Float16* out;
  int64_t Ntot;
  int C, H, W, O;
  float stride;
  int relu;
  unsigned x_bytes;
  int tile_base;
  unsigned n_full;
  LevelTab lt;
};
template <bool OUT_NHWC, int SRC, int TH = 8>
__global__ __launch_bounds__(512, 2) void k_dcn_patch(PatchArgs a) {
  if constexpr (TH == 8 && OUT_NHWC && SRC == 1) {
    if (a.n_full != 0 && blockIdx.x >= a.n_full) {      // (uniform)
      dcn_patch_tile<OUT_NHWC, SRC, 4>(a.x, a.src, a.wfrag, a.out, a.Ntot, a.C, a.H, a.W, a.O, a.stride, a.relu, a.x_bytes, a.lt,
                                           (int)a.n_full, blockIdx.x - a.n_full, gridDim.x - a.n_full, threadIdx.x);
      return;
    }
  }
  dcn_patch_tile<OUT_NHWC, SRC, TH>(a.x, a.src, a.wfrag, a.out, a.Ntot, a.C, a.H, a.W, a.O, a.stride, a.relu, a.x_bytes, a.lt,
                                        a.tile_base, blockIdx.x, a.n_full ? a.n_full : gridDim.x, threadIdx.x);
}

// ------------------------------------------------------------------ regular convolutions (f16)
// The conv towers of S2ANetHead (models/head.py:163-222: fam_reg_ls, fam_cls_ls, odm_reg_ls,
// odm_cls_ls, or_conv — nine 256->256 3x3 convolutions per FPN level) are the patch-staged
// AlignConv with integer sampling points: no blend, no column tile — the MFMA waves read their B
// fragments straight out of the LDS patch, the weights come in fragment order from L2, and a
// barrier is only needed once per 64-channel chunk.  256 threads = 4 waves; <= 67.6 KB of LDS ->
// two workgroups per CU, so one tile's prologue/epilogue overlaps the other's MFMA loop.  Bias,
// residual and ReLU are fused into the LDS-staged epilogue (one pass over the output instead of
// conv + bias/add/ReLU kernels).  TAPS = 9: 3x3/stride 1/pad 1 on an 8x16 position tile with a
// one-pixel halo.  TAPS = 1: 1x1 (stride 1 or 2) on 128 consecutive output positions — a plain
// GEMM with the same pipeline (the bottleneck 1x1 layers and FPN laterals of the carrier).
// OG = 64-channel output groups per workgroup (4, 2 or 1): with fewer than four groups the waves
// split the 128 positions instead, so narrow layers still use all four MFMA waves.
constexpr int kCPW = 18;                                  // stride 1: 16 positions + 1 halo each side

// PH = 128-position blocks per workgroup (1: 8 x 16 tile, 256 threads, two workgroups per CU; 2: 16 x 16 tile,
// 512 threads, one workgroup per CU, filter through LDS -- the pyramid-packed 256 -> 256 towers)
// optional second results computed from the staged output tile in the epilogue
struct ConvExtra {
  _Float16* pool_out;          // [P, O/8]: max over runs of 8 channels (rotation-invariant pooling) or null
  const _Float16* head_w;      // 1x1 prediction head on the tile: fragment-order filter (<= 32 maps, zero-padded) or null
  const _Float16* head_b;      // its bias (>= 32 entries)
  _Float16* head_out;          // [P, 64] (columns 0..31 written)
  int store_main;              // 0: the tower's own output is not needed (only the head reads it)
  // TAIL kernels only: the 1x1 convolution that follows (a bottleneck's conv3, models/backbone.py:60-83) applied to
  // the staged tile: out2 = relu(W2 . relu(conv + bias) + bias2 + residual2), 256 maps
  const _Float16* tail_w;      // fragment-order 1x1 filter [256][64]
  const _Float16* tail_b;      // [256]
  const _Float16* tail_res;    // [P, 256] or null
  _Float16* tail_out;          // [P, 256]
  // ... and optionally the NEXT bottleneck's conv1 (1x1, 256 -> 64, + bias + ReLU) on the finished output tile
  const _Float16* chain_w;     // fragment-order 1x1 filter [chain_O][256] or null
  const _Float16* chain_b;     // [chain_O]
  _Float16* chain_out;         // [P, chain_O]
  int chain_O;                 // 64 | 128
};

// SD = spatial stride of the 3x3 form (1, or 2: the down-sampling conv2 of a stage's first bottleneck; output tile
// 4 x 16 positions from a 9 x 33-pixel patch, two 32-position tiles per wave)
// HT = 1: 4 x 16 tile of a stride-1 3x3 (64 positions) for maps so small that 8 x 16 tiles leave CUs idle or give every
// CU a single workgroup (one wave per SIMD: nothing covers the weight / patch round trips)
template <int TAPS, int OG, int PH = 1, int SD = 1, bool TAIL = false, int HT = 0>
struct ConvCfg {
  static constexpr int kTH = (SD == 2 || HT) ? 4 : 8 * PH;             // tile rows (TAPS 9)
  static constexpr int kPos = (SD == 2 || HT) ? 64 : 128 * PH;         // output positions per workgroup
  static constexpr int kWaves = 4 * PH;
  static constexpr int kPW = SD == 2 ? 33 : kCPW;                      // patch width in pixels
  static constexpr int kNT = (kPos / 32) * OG / 4 / PH;                // 32-position tiles per wave
  static constexpr int kPix = TAPS == 9 ? ((kTH - 1) * SD + 3) * kPW : kPos;   // patch pixels
  static constexpr int kDma = (kPix * 9 + 63) / 64;                   // 1 KB LDS-DMA pieces per patch
  static constexpr int kPatchBytes = kDma * 1024;
  static constexpr int kOutRowB = OG * 128 + 16;                      // staged output row (bytes)
  // PH = 2 (TAPS 9, OG 4): the filter of a tap (32 KB) is staged through LDS once per workgroup, two buffers
  static constexpr bool kWLds = PH == 2 && TAPS == 9;
  static constexpr int kWBuf = OG * 8192;
  static constexpr int kLoop0 = 2 * kPatchBytes + (kWLds ? 2 * kWBuf : 0);
  // (OG 1 with the filter through LDS: room for all nine taps behind the first patch buffer)
  static constexpr int kLoop = (kWLds && OG == 1 && kPatchBytes + 9 * 8192 > kLoop0) ? kPatchBytes + 9 * 8192 : kLoop0;
  static constexpr int kLds0 = (kLoop > kPos * kOutRowB) ? kLoop : kPos * kOutRowB;
  static constexpr int kTailRowB = 528;                                // staged 256-map row of the fused 1x1
  static constexpr int kLds = (TAIL && kPos * kTailRowB > kLds0) ? kPos * kTailRowB : kLds0;
  static constexpr int kBiasBytes = TAIL ? 1024 : 512;
  static constexpr int kJ = (kDma + kWaves - 1) / kWaves;             // DMA pieces per wave
};

template <int TAPS, int OG, int PH = 1, int SD = 1, bool TAIL = false, int HT = 0>
__global__ __launch_bounds__(256 * PH, PH == 1 ? 2 : 1) void k_conv_f16(const _Float16* __restrict__ x_,
                                                     const _Float16* __restrict__ wfrag,
                                                     const _Float16* __restrict__ bias,
                                                     const _Float16* __restrict__ residual_,
                                                     _Float16* __restrict__ out_, int64_t Ntot_, int C,
                                                     int H_, int W_, int Ho_, int Wo_, int cstride, int O,
                                                     int relu, unsigned x_bytes_, LevelTab lt, int res_up,
                                                     ConvExtra ex) {
  _Float16* pool_out_ = ex.pool_out;
  using T = _Float16;
  using V = f16x8;
  using Cfg = ConvCfg<TAPS, OG, PH, SD, TAIL, HT>;
  constexpr int NT = Cfg::kNT;        // 32-position tiles per wave
  static_assert(NT >= 1, "unsupported tile / group combination");
  static_assert(!HT || (TAPS == 9 && PH == 1 && SD == 1 && !TAIL), "half tiles: plain 3x3 / stride 1");
  static_assert(!TAIL || (TAPS == 9 && OG == 1 && SD == 1), "the fused 1x1 tail follows a 64-map 3x3/s1");
  constexpr int WPG = 4 / OG;         // waves per out-channel group (inside a 128-position block)
  constexpr int kThreads_ = 256 * PH;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave4 = wave & 3, blk = wave >> 2;     // wave inside its 128-position block, block index
  int64_t tile = xcd_remap(blockIdx.x, gridDim.x);
  const T* x = x_;
  const T* residual = residual_;
  T* out = out_;
  int64_t Ntot = Ntot_;
  int H = H_, W = W_, Ho = Ho_, Wo = Wo_;
  unsigned x_bytes = x_bytes_;
  if (TAPS == 9 && lt.n > 1) {        // pyramid-packed levels: rebind this workgroup to its level
    int t0 = 0, p0 = 0;
#pragma unroll
    for (int i = 0; i < kMaxLevels; i++)
      if (i < lt.n && tile >= lt.tile0[i]) {
        t0 = lt.tile0[i]; p0 = lt.pix0[i]; H = lt.H[i]; W = lt.W[i];
      }
    tile -= t0;
    Ho = H; Wo = W;
    Ntot = (int64_t)lt.batch * H * W;
    x += (int64_t)p0 * C;
    out += (int64_t)p0 * O;
    if (residual) residual += (int64_t)p0 * O;
    if (pool_out_) pool_out_ += (int64_t)p0 * (O / 8);
    if (ex.head_out) ex.head_out += (int64_t)p0 * 64;
    x_bytes = (unsigned)(Ntot * C * 2);
  }
  const int64_t HWo = (int64_t)Ho * Wo, HWi = (int64_t)H * W;
  // TAPS 9: 2-D tile of one image; TAPS 1: 128 consecutive output positions of the whole batch
  const int txn = (Wo + 15) / 16, tyn = (Ho + Cfg::kTH - 1) / Cfg::kTH;
  const int64_t bimg = TAPS == 9 ? tile / (txn * tyn) : 0;
  const int trem = TAPS == 9 ? (int)(tile % (txn * tyn)) : 0;
  const int ty0 = (trem / txn) * Cfg::kTH, tx0 = (trem % txn) * 16;
  const int64_t g0 = tile * Cfg::kPos;
  const int o0 = blockIdx.y * (64 * OG);
  const int Oloc = min(64 * OG, O - o0);
  const int CC = (C + 63) / 64, G = O / 64;
  const int qlim = C >= 64 ? 8 : C / 8;   // C = 32: half-filled chunk, the filter is zero-padded to 64 inputs
  const unsigned row_bytes = (unsigned)C * 2;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(x), 0, (int)x_bytes, 0x00020000);

  // Patch: global -> LDS by LDS-DMA (buffer_load ... lds).  One wave instruction writes 64 x 16 B
  // linearly; pixels sit 144 B apart (128 B of channels + a 16-byte pad chunk: conflict-free
  // ds_read_b128 fragments), linear slot v = pixel*9 + chunk; pad chunks and pixels outside the
  // image read an out-of-range offset (-> zeros).
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  unsigned pvoff[Cfg::kJ];
#pragma unroll
  for (int j = 0; j < Cfg::kJ; j++) {
    int v = (wave_u + Cfg::kWaves * j) * 64 + lane, p = v / 9, q = v % 9;
    bool in = q < qlim && p < Cfg::kPix;
    int64_t pix = 0;
    if (TAPS == 9) {
      int yy = ty0 * SD - 1 + p / Cfg::kPW, xx = tx0 * SD - 1 + p % Cfg::kPW;
      in = in && yy >= 0 && yy < H && xx >= 0 && xx < W;
      pix = bimg * HWi + (int64_t)yy * W + xx;
    } else {
      int64_t g = g0 + p;
      in = in && g < Ntot;
      int64_t bb = g / HWo, r = g % HWo;
      pix = bb * HWi + (r / Wo) * cstride * (int64_t)W + (r % Wo) * cstride;
    }
    pvoff[j] = in ? (unsigned)(pix * row_bytes + q * 16) : 0x80000000u;
  }
  // bias of this workgroup's out channels -> LDS (behind the tile buffers): the epilogue reads it
  // with ds_read instead of eight dependent global loads
  T* s_bias = reinterpret_cast<T*>(smem + Cfg::kLds);
  T bias_v = (T)0.f;
  if (bias && tid < Oloc) bias_v = bias[o0 + tid];   // in flight with the first patch / weights
  T tail_bias_v = (T)0.f;
  V wt[2][4];                                        // TAIL: this wave's 64 x 64 block of the 1x1 filter
  if constexpr (TAIL) {
    if (tid < 256) tail_bias_v = ex.tail_b[tid];
    // 16x16x32 fragments (f = 16-channel tile * 2 + k-step), as the stand-alone 1x1 takes them
    const V* tp = reinterpret_cast<const V*>(ex.tail_w) + (int64_t)wave4 * 8 * 64 + ((lane >> 4) & 1) * 128 + (lane >> 5) * 32 + (lane & 15);
#pragma unroll
    for (int f = 0; f < 8; f++) wt[f >> 2][f & 3] = tp[((f >> 2) * 4 + (f & 1)) * 64 + ((f >> 1) & 1) * 16];
  }
  auto patch_issue = [&](int cc) {
    char* P = smem + (cc & 1) * Cfg::kPatchBytes;
#pragma unroll
    for (int j = 0; j < Cfg::kJ; j++) {
      const int i = wave_u + Cfg::kWaves * j;
      if (i < Cfg::kDma)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(P + i * 1024), 16,
                                                 (int)pvoff[j], cc * 128, 0, 0);
    }
  };

  const int grp = wave4 / WPG, sub = wave4 % WPG;     // out-channel group, position sub-range (inside the block)
  const bool wave_active = grp * 64 < Oloc;
  const int g = min(o0 / 64 + grp, G - 1);
  const V* wf_base = reinterpret_cast<const V*>(wfrag) + lane;
  V wA[2][4], wB[2][4];
  // (every stride-1 3x3 launch and the full-width 1x1 launches; the narrower 1x1 launches stay on 32x32x16, as the chained conv1 of
  // the fused tail, which must agree with them bit for bit)
  // v_mfma_f32_16x16x32_f16 there (same-box: towers -5 ... -7 %, every full-width layer +3 % end to end)
  constexpr bool M16 = (TAPS == 9 && SD == 1) || (TAPS == 1 && OG == 4);
  constexpr int NB16 = 2 * NT;        // 16-position tiles per wave (8; 4 on the 64-position tiles)
  auto load_w = [&](int s, V (&wv)[2][4]) {
    if constexpr (M16) {
      // 16x16x32 fragments out of the same packed filter (lane maps below): fragment f = (16-channel tile f >> 1, k-step f & 1)
      const V* p = reinterpret_cast<const V*>(wfrag) + ((int64_t)s * G + g) * 8 * 64 + ((lane >> 4) & 1) * 128 + (lane >> 5) * 32 + (lane & 15);
#pragma unroll
      for (int f = 0; f < 8; f++) wv[f >> 2][f & 3] = p[((f >> 2) * 4 + (f & 1)) * 64 + ((f >> 1) & 1) * 16];
      return;
    }
    const V* p = wf_base + ((int64_t)s * G + g) * 8 * 64;
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int kk = 0; kk < 4; kk++) wv[a][kk] = p[(a * 4 + kk) * 64];
  };
  // per N-tile b: byte offset of this lane's position (tap (0,0)) and k-half inside the patch;
  // the rest of a fragment address is a compile-time immediate (tap, kk)
  // lane -> position inside a 32-position tile.  3x3 / stride 1: a tile is two patch rows of 16 pixels, 18 pixels =
  // 162 sixteen-byte slots apart; ds_read_b128 serves lanes {0-3,12-15,20-27} (and the three like groups) in one
  // cycle only if their slots differ mod 16, and the second row's x = 4..11 land 2 slots-classes off the first row's
  // -> every group was 2-way conflicted (SQ_LDS_BANK_CONFLICT = half of the LDS cycles).  Rotating the second
  // row's pixels by two lanes makes all sixteen classes distinct; the epilogue uses the same map.
  const int lp = (TAPS == 9 && SD == 1) ? ((lane & 16) | (((lane & 15) - ((lane >> 3) & 2)) & 15)) : (lane & 31);
  int fbase[NT];
#pragma unroll
  for (int b = 0; b < NT; b++) {
    int pl = 128 * blk + 32 * (sub * NT + b) + lp;
    int pix = TAPS == 9 ? SD * ((pl >> 4) * Cfg::kPW + (pl & 15)) : pl;
    fbase[b] = pix * kRowBytes + (lane >> 5) * 16;
  }
  // full-width layers (OG 4, 128 positions per wave): v_mfma_f32_16x16x32_f16 -- same flops per cycle and the same LDS reads
  // per flop as 32x32x16, but the chip holds a higher clock on it under load (MI355X_MICROARCH.md, clocks (7): measured here
  // 187 -> 173 us on the pyramid towers, same box); the wave's 64 x 128 outputs are 4 x 8 tiles of 16 out channels x 16
  // positions (3x3: one patch row of 16 pixels)
  f32x16 acc[2][NT];
  f32x4 acc16[M16 ? 4 : 1][M16 ? NB16 : 1];
  if constexpr (M16) {
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int b = 0; b < NB16; b++)
#pragma unroll
        for (int r = 0; r < 4; r++) acc16[a][b][r] = 0.f;
  } else {
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int b = 0; b < NT; b++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;
  }
  // 16x16x32 lane maps.  Lane l = (i = l & 15, kg = l >> 4).  Its eight k-values are the channel group c(ks, kg) =
  // {0, 4, 1, 5}[kg] + 2 ks of the 64-channel chunk (any order works as long as A and B agree): within the lane groups a
  // ds_read_b128 serves per cycle ({0-3,12-15,20-27}, ...) the two k-groups present then sit 64 B = 4 sixteen-byte slots
  // apart, and with the pixel map pix16 (i in 4..11 -> pixels = 0,1 mod 4, the others -> 2,3 mod 4; pixel pitch 9 slots)
  // all sixteen slots of a group differ -- conflict-free B reads.  The A fragment comes out of the SAME packed filter as
  // the 32x32x16 form: (out channel o, channel group c) is the 16 B at ((o>>5)*4 + (c>>1))*1 KB + ((c&1)*32 + (o&31))*16.
  const int kg16 = lane >> 4, i16 = lane & 15;
  const int pix16 = (i16 >= 4 && i16 < 12) ? (((i16 - 4) >> 1) * 4 + (i16 & 1))
                                           : (((i16 & 3) >> 1) * 4 + 2 + (i16 & 1) + (i16 >= 12 ? 8 : 0));
  const int t16 = 2 * sub * NT;                   // first 16-position tile of this wave inside its 128-position block
  const int fbase16 = (TAPS == 9 ? (8 * blk + t16) * Cfg::kPW + pix16 : 128 * blk + 16 * t16 + pix16) * kRowBytes + (kg16 & 1) * 64 + (kg16 >> 1) * 16;
  constexpr int kTile16 = (TAPS == 9 ? Cfg::kPW : 16) * kRowBytes;      // LDS distance between a wave's 16-position tiles
  const int abase16 = (kg16 & 1) * 2048 + ((kg16 >> 1) * 32 + i16) * 16;

  auto compute = [&](const char* P, int t, const V (&wv)[2][4]) {
    if (!wave_active) return;
    const int toff = ((t / 3) * Cfg::kPW + (t % 3)) * kRowBytes;
    if constexpr (M16) {
#pragma unroll
      for (int ks = 0; ks < 2; ks++) {
        V pf[NB16];
#pragma unroll
        for (int b = 0; b < NB16; b++) pf[b] = *reinterpret_cast<const V*>(P + fbase16 + b * kTile16 + toff + ks * 32);
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
          for (int b = 0; b < NB16; b++)
            acc16[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv[(a * 2 + ks) >> 2][(a * 2 + ks) & 3], pf[b], acc16[a][b], 0, 0, 0);
      }
      return;
    }
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
      V pf[NT];
#pragma unroll
      for (int b = 0; b < NT; b++) pf[b] = *reinterpret_cast<const V*>(P + fbase[b] + toff + kk * 32);
#pragma unroll
      for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < NT; b++)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wv[a][kk], pf[b], acc[a][b], 0, 0, 0);
    }
  };

  // fused tail: residual vectors of the whole 256-map tile.  With the filter through LDS (16 x 16 tiles) they are requested
  // right after the prologue's barrier: the 3x3 GEMM waits on LDS reads only, so the 128 KB are in flight under it and under
  // the first epilogue instead of in front of the second one (at kernel start they queue ahead of the patch and the
  // filter on the in-order memory path: measured slower).  Register-filter form: requested behind the second GEMM, as before.
  constexpr bool kEarlyRes = TAIL && Cfg::kWLds;
  constexpr int NI2 = TAIL ? Cfg::kPos * 32 / kThreads_ : 1;
  unsigned off2[NI2];
  V r2[NI2];
  auto tail_res_issue = [&]() {
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T*>(ex.tail_res ? ex.tail_res : ex.tail_out), 0, (int)((uint64_t)Ntot * 256 * 2), 0x00020000);
#pragma unroll
    for (int i = 0; i < NI2; i++) {
      const int idx = tid + kThreads_ * i, pos = idx >> 5, col = idx & 31;
      const int64_t gp = tile_pos(tile, pos, Cfg::kTH, Ho, Wo, HWo, Ntot);
      off2[i] = gp >= 0 ? (unsigned)((gp * 256 + col * 8) * 2) : 0x80000000u;
      r2[i] = __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(rr, (int)(ex.tail_res ? off2[i] : 0x80000000u), 0, 0));
    }
  };
  const int nstage = TAPS * CC, last = nstage - 1;
#if S2A_STAMP
  unsigned long long t_tic = 0, t_work = 0, t_wait = 0;
  if ((Cfg::kWLds && OG == 4) || (TAPS == 1 && OG == 4 && SD == 1)) S2A_STAMP_AT(0);
#endif
  if constexpr (Cfg::kWLds) {
    // 16 x 16 tile, filter through LDS: every tap's 32 KB (this workgroup's 256 out channels, fragment order =
    // contiguous) is DMA-ed once into one of two LDS buffers while the previous tap computes; all eight waves
    // read their A fragments from there (ds_read_b128, lane-linear).  Halves the filter bytes a CU pulls per
    // flop compared with two 8 x 16 workgroups; costs a barrier per tap.
    char* wb = smem + 2 * Cfg::kPatchBytes;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T*>(wfrag), 0, (int)((uint64_t)O * (uint64_t)(CC * 64) * 9 * 2), 0x00020000);
    const int wbase = (o0 / 64) * 8192;
    auto w_issue = [&](int s) {      // 8 * OG pieces of 1 KB over the eight waves
#pragma unroll
      for (int j = 0; j < OG; j++) {
        const int piece = wave_u * OG + j;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(wb + (s & 1) * Cfg::kWBuf + piece * 1024),
                                                 16, piece * 1024 + lane * 16, s * G * 8192 + wbase, 0, 0);
      }
    };
    auto compute_wl = [&](const char* P, int t, const char* Wb) {
      const int toff = ((t / 3) * Cfg::kPW + (t % 3)) * kRowBytes;
      if constexpr (M16) {
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
          V wv[4], pf[NB16];
#pragma unroll
          for (int a = 0; a < 4; a++)
            wv[a] = *reinterpret_cast<const V*>(Wb + grp * 8192 + abase16 + (a >> 1) * 4096 + ks * 1024 + (a & 1) * 256);
#pragma unroll
          for (int b = 0; b < NB16; b++)
            pf[b] = *reinterpret_cast<const V*>(P + fbase16 + b * kTile16 + toff + ks * 32);
#pragma unroll
          for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < NB16; b++)
              acc16[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv[a], pf[b], acc16[a][b], 0, 0, 0);
        }
        return;
      }
#pragma unroll
      for (int kk = 0; kk < 4; kk++) {
        V pf[NT], wv[2];
#pragma unroll
        for (int a = 0; a < 2; a++) wv[a] = *reinterpret_cast<const V*>(Wb + (((grp * 2 + a) * 4 + kk) * 64 + lane) * 16);
#pragma unroll
        for (int b = 0; b < NT; b++) pf[b] = *reinterpret_cast<const V*>(P + fbase[b] + toff + kk * 32);
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
          for (int b = 0; b < NT; b++)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wv[a], pf[b], acc[a][b], 0, 0, 0);
      }
    };
    if (OG == 1 && CC == 1) {
      // 64 input maps, one out-channel group: the WHOLE 3x3 filter (9 x 8 KB) goes into LDS at once (over the second
      // patch buffer, which a one-chunk layer never uses) -- one memory latency per tile instead of one per tap
      // (eight MFMAs per wave and tap cannot cover an L2 round trip), and no barrier inside the tile
      char* wall = smem + Cfg::kPatchBytes;
      patch_issue(0);
#pragma unroll
      for (int t = 0; t < 9; t++)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(wall + t * 8192 + wave_u * 1024),
                                                 16, wave_u * 1024 + lane * 16, t * G * 8192 + wbase, 0, 0);
      if (tid < 64) s_bias[tid] = bias_v;
      if constexpr (TAIL) { if (tid < 256) s_bias[64 + tid] = tail_bias_v; }
      __syncthreads();
      if constexpr (kEarlyRes) tail_res_issue();
#pragma unroll
      for (int t = 0; t < 9; t++) compute_wl(smem, t, wall + t * 8192);
      __syncthreads();
    } else {
    patch_issue(0);
    w_issue(0);
    if (tid < 64 * OG) s_bias[tid] = bias_v;
    if constexpr (TAIL) { if (tid < 256) s_bias[64 + tid] = tail_bias_v; }
    S2A_STAMP_AT(1);
    __syncthreads();
    S2A_STAMP_AT(2);
    // (Measured dead end, round 2: fragments one k-step ahead in two register sets across taps, the barrier in front of
    // the tap's last k-step with counted vmcnt, DMAs issued behind it -- bit-identical and within noise of this form,
    // 219-225 vs 222 us: at two waves per SIMD the partner wave already covers these waits.  Timing-only ablations of
    // that form: no filter DMA in the loop -7 %, no patch DMA 0 %, no barrier -3 %, neither -11 %.)
    for (int cc = 0; cc < CC; cc++) {
      const char* Pc = smem + (cc & 1) * Cfg::kPatchBytes;
#pragma unroll
      for (int t = 0; t < 9; t++) {
        const int s = cc * 9 + t;
        if (s + 1 < nstage) w_issue(s + 1);
        if (t == 0 && cc + 1 < CC) patch_issue(cc + 1);
        S2A_TIC();
        compute_wl(Pc, t, wb + (s & 1) * Cfg::kWBuf);
        S2A_TOC(t_work);
        S2A_TIC();
        __syncthreads();     // drains this tap's DMAs (vmcnt(0)) and frees the buffers they will overwrite next
        S2A_TOC(t_wait);
      }
    }
    S2A_STAMP_AT(3);
    S2A_STAMP_VAL(6, t_work);
    S2A_STAMP_VAL(7, t_wait);
    }
  } else {
  patch_issue(0);
  load_w(0, wA);
  if (tid < 64 * OG) s_bias[tid] = bias_v;
  if constexpr (TAIL) { if (tid < 256) s_bias[64 + tid] = tail_bias_v; }
#if S2A_STAMP
  if (TAPS == 1 && OG == 4 && SD == 1) S2A_STAMP_AT(1);
#endif
  __syncthreads();   // (the compiler drains the DMA with vmcnt(0) before the barrier)
#if S2A_STAMP
  if (TAPS == 1 && OG == 4 && SD == 1) S2A_STAMP_AT(2);
#endif
  if constexpr (TAPS == 9) {
    for (int cc = 0; cc < CC; cc++) {
      const int s0 = cc * 9;
      const char* Pc = smem + (cc & 1) * Cfg::kPatchBytes;
      // 9 taps, weight fragments double-buffered in registers (static indexing: unrolled by hand)
#define S2A_TAP(T_, WCUR, WNEXT)                                                     \
      load_w(min(s0 + (T_) + 1, last), WNEXT);                                          \
      compute(Pc, (T_), WCUR);                                                          \
      __builtin_amdgcn_sched_barrier(0); /* keep the next taps' loads from being hoisted (registers) */
      S2A_TAP(0, wA, wB)
      S2A_TAP(1, wB, wA)
      S2A_TAP(2, wA, wB)
      S2A_TAP(3, wB, wA)
      S2A_TAP(4, wA, wB)
      S2A_TAP(5, wB, wA)
      // next chunk's patch: issued here so that tap 6 still runs on weights loaded before the DMA
      // (vmcnt is in-order) and taps 6-8 cover its latency
      if (cc + 1 < CC) patch_issue(cc + 1);
      S2A_TAP(6, wA, wB)
      S2A_TAP(7, wB, wA)
      S2A_TAP(8, wA, wB)
#undef S2A_TAP
      __syncthreads();
      // after an odd number of taps the roles of wA/wB are swapped: copy back (8 v_movs per chunk)
#pragma unroll
      for (int a = 0; a < 2; a++)
#pragma unroll
        for (int kk = 0; kk < 4; kk++) wA[a][kk] = wB[a][kk];
    }
  } else {
    // one stage per 64-channel chunk: next chunk's tile and weights in flight under this one's MFMAs
#define S2A_STAGE(C_, WCUR, WNEXT)                                                   \
    if ((C_) + 1 < CC) patch_issue((C_) + 1);                                           \
    load_w(min((C_) + 1, last), WNEXT);                                                 \
    S2A_TIC();                                                                          \
    compute(smem + ((C_) & 1) * Cfg::kPatchBytes, 0, WCUR);                             \
    S2A_TOC(t_work); S2A_TIC();                                                         \
    __syncthreads();                                                                    \
    S2A_TOC(t_wait);
    int cc = 0;
    for (; cc + 1 < CC; cc += 2) {
      S2A_STAGE(cc, wA, wB)
      S2A_STAGE(cc + 1, wB, wA)
    }
    if (cc < CC) { S2A_STAGE(cc, wA, wB) }
#undef S2A_STAGE
#if S2A_STAMP
    if (TAPS == 1 && OG == 4 && SD == 1) { S2A_STAMP_AT(3); S2A_STAMP_VAL(6, t_work); S2A_STAMP_VAL(7, t_wait); }
#endif
  }
  }   // !kWLds

  // ---- epilogue: bias (+ residual) + ReLU; tile staged through LDS, rows stored 16 B per lane
  char* s_out = smem;
  // ReLU on the ROUNDED halves, two per instruction (rounding is monotonic and keeps zero: max(round(v), 0) == round(max(v, 0));
  // NaN -> 0 either way), and the ReLU switch as one uniform branch around the tile -- fmaxf plus a per-value select on the
  // f32 sums was 9 instructions per two values, 4.8 k of a tower tile's 113 k cycles (same-box A/B: pyramid tower launch
  // 195.9 -> 191.4 us on dense data)
  using h2e = __attribute__((ext_vector_type(2))) _Float16;
  using h4e = __attribute__((ext_vector_type(4))) _Float16;
  auto stage_tile = [&](auto relu_c) {
    constexpr bool kRelu = decltype(relu_c)::value;
    auto quad = [&](float v0, float v1, float v2, float v3, const h4e& bq) {
      h2e lo = {(_Float16)(v0 + (float)bq[0]), (_Float16)(v1 + (float)bq[1])};
      h2e hi = {(_Float16)(v2 + (float)bq[2]), (_Float16)(v3 + (float)bq[3])};
      if constexpr (kRelu) {
        lo = __builtin_elementwise_max(lo, h2e{(_Float16)0.f, (_Float16)0.f});
        hi = __builtin_elementwise_max(hi, h2e{(_Float16)0.f, (_Float16)0.f});
      }
      return h4e{lo[0], lo[1], hi[0], hi[1]};
    };
    if constexpr (M16) {
#pragma unroll
      for (int a = 0; a < 4; a++) {
        const int och = grp * 64 + 16 * a + 4 * kg16;       // D: row (out channel) = 4 (lane >> 4) + register, column = pixel
        const h4e bq = *reinterpret_cast<const h4e*>(s_bias + och);
#pragma unroll
        for (int b = 0; b < NB16; b++) {
          const int pos = 128 * blk + 16 * (t16 + b) + pix16;
          *reinterpret_cast<h4e*>(s_out + pos * Cfg::kOutRowB + och * 2) =
              quad(acc16[a][b][0], acc16[a][b][1], acc16[a][b][2], acc16[a][b][3], bq);
        }
      }
    } else {
#pragma unroll
      for (int a = 0; a < 2; a++)
#pragma unroll
        for (int rq = 0; rq < 4; rq++) {
          const int och = grp * 64 + 32 * a + 8 * rq + 4 * (lane >> 5);
          const h4e bq = *reinterpret_cast<const h4e*>(s_bias + och);
#pragma unroll
          for (int b = 0; b < NT; b++) {
            int pos = 128 * blk + 32 * (sub * NT + b) + lp;
            *reinterpret_cast<h4e*>(s_out + pos * Cfg::kOutRowB + och * 2) =
                quad(acc[a][b][rq * 4], acc[a][b][rq * 4 + 1], acc[a][b][rq * 4 + 2], acc[a][b][rq * 4 + 3], bq);
          }
        }
    }
  };
  if (wave_active) {
    if (relu && !residual) stage_tile(std::true_type{}); else stage_tile(std::false_type{});
  }
  __syncthreads();
#if S2A_STAMP
  if ((Cfg::kWLds && OG == 4) || (TAPS == 1 && OG == 4 && SD == 1)) S2A_STAMP_AT(4);
#endif
  if constexpr (TAIL) {
    // ---- fused 1x1 (64 -> 256) on the staged tile: wave w = out maps 64w..64w+63 x the 128 positions of its block,
    // B fragments from the staged rows (144-byte stride: conflict-free), accumulation order = the stand-alone 1x1's
    // (the second GEMM on 16x16x32 MFMAs, as the stand-alone 64 -> 256 1x1)
    f32x4 acc3s[4][8];
    {
#pragma unroll
      for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 8; b++)
#pragma unroll
          for (int r = 0; r < 4; r++) acc3s[a][b][r] = 0.f;
      const char* brow = s_out + (128 * blk + pix16) * Cfg::kOutRowB + (kg16 & 1) * 64 + (kg16 >> 1) * 16;
#pragma unroll
      for (int ks = 0; ks < 2; ks++)
#pragma unroll
        for (int bh = 0; bh < 8; bh += 4) {      // four position tiles at a time: the residual prefetch below needs the registers
          V pf[4];
#pragma unroll
          for (int b = 0; b < 4; b++) pf[b] = *reinterpret_cast<const V*>(brow + (bh + b) * 16 * Cfg::kOutRowB + ks * 32);
#pragma unroll
          for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 4; b++)
              acc3s[a][bh + b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wt[(a * 2 + ks) >> 2][(a * 2 + ks) & 3], pf[b], acc3s[a][bh + b], 0, 0, 0);
        }
    }
    // residual vectors of the whole tile in flight before the tile is re-staged (issuing them at kernel start was
    // slower: they queue ahead of the patch and the filters on the in-order memory path)
    if constexpr (!kEarlyRes) {
      const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<T*>(ex.tail_res ? ex.tail_res : ex.tail_out), 0, (int)((uint64_t)Ntot * 256 * 2), 0x00020000);
#pragma unroll
      for (int i = 0; i < NI2; i++) {
        const int idx = tid + kThreads_ * i, pos = idx >> 5, col = idx & 31;
        const int64_t gp = tile_pos(tile, pos, Cfg::kTH, Ho, Wo, HWo, Ntot);
        off2[i] = gp >= 0 ? (unsigned)((gp * 256 + col * 8) * 2) : 0x80000000u;
        r2[i] = __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(rr, (int)(ex.tail_res ? off2[i] : 0x80000000u), 0, 0));
      }
    }
    __syncthreads();                       // every wave has read its B fragments: the tile may be overwritten
    {
      using h4 = __attribute__((ext_vector_type(4))) _Float16;
#pragma unroll
      for (int a = 0; a < 4; a++) {
        const int och = wave4 * 64 + 16 * a + 4 * kg16;
        const h4 bq = *reinterpret_cast<const h4*>(s_bias + 64 + och);
#pragma unroll
        for (int b = 0; b < 8; b++) {
          h4 v4;
#pragma unroll
          for (int e = 0; e < 4; e++) v4[e] = (_Float16)(acc3s[a][b][e] + (float)bq[e]);
          *reinterpret_cast<h4*>(s_out + (128 * blk + 16 * b + pix16) * Cfg::kTailRowB + och * 2) = v4;
        }
      }
    }
    __syncthreads();
    // chained conv1: its 16 filter fragments are requested here, in front of the residual add / store pass (they were
    // loaded behind it, one exposed L2 round trip per tile in front of the third GEMM)
    const int O3 = ex.chain_O, MT = max(O3 / 32, 1);                // 2 | 4 m-tiles
    const int nper = (Cfg::kPos / 32) * MT / Cfg::kWaves;          // 32-position tiles per wave: 2 | 4
    const int mt = wave % MT, nt0 = (wave / MT) * nper;
    const int G3 = O3 / 64;
    V aw[16];
    if (ex.chain_w) {
      const V* cwp = reinterpret_cast<const V*>(ex.chain_w) + lane + ((mt >> 1) * 8 + (mt & 1) * 4) * 64;
#pragma unroll
      for (int c4 = 0; c4 < 4; c4++)
#pragma unroll
        for (int kk = 0; kk < 4; kk++) aw[c4 * 4 + kk] = cwp[(c4 * G3 * 8 + kk) * 64];
    }
#pragma unroll
    for (int i = 0; i < NI2; i++) {
      const int idx = tid + kThreads_ * i, pos = idx >> 5, col = idx & 31;
      V v = *reinterpret_cast<const V*>(s_out + pos * Cfg::kTailRowB + col * 16);
#pragma unroll
      for (int e = 0; e < 8; e += 2) {     // (ReLU on the rounded halves, two per instruction: see the epilogue above)
        h2e p2 = {(_Float16)((float)v[e] + (float)r2[i][e]), (_Float16)((float)v[e + 1] + (float)r2[i][e + 1])};
        p2 = __builtin_elementwise_max(p2, h2e{(_Float16)0.f, (_Float16)0.f});
        v[e] = p2[0];
        v[e + 1] = p2[1];
      }
      if (off2[i] != 0x80000000u) *reinterpret_cast<V*>(reinterpret_cast<char*>(ex.tail_out) + off2[i]) = v;
      if (ex.chain_w) *reinterpret_cast<V*>(s_out + pos * Cfg::kTailRowB + col * 16) = v;   // finished rows back to LDS
    }
    if (ex.chain_w) {
      // ---- the next block's conv1 on the finished 256-map tile: out3 = relu(W . y + b), 64 maps (same stage) or 128
      // (first block of the next stage).  wave = (m-tile, a run of 32-position tiles); B fragments from the staged
      // rows (528-byte stride: conflict-free), the 16 filter fragments of the m-tile straight from L2; K order =
      // the stand-alone 1x1 kernel's (chunk, k-step).
      f32x16 c2[4];
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) c2[j][r] = 0.f;
      __syncthreads();                     // the whole finished tile is in LDS
#pragma unroll
      for (int c4 = 0; c4 < 4; c4++)
#pragma unroll
        for (int kk = 0; kk < 4; kk++)
#pragma unroll
          for (int j = 0; j < 4; j++)
            if (j < nper) {
              const V bf = *reinterpret_cast<const V*>(s_out + (32 * (nt0 + j) + (lane & 31)) * Cfg::kTailRowB +
                                                       (c4 * 64 + kk * 16 + (lane >> 5) * 8) * 2);
              c2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aw[c4 * 4 + kk], bf, c2[j], 0, 0, 0);
            }
      __syncthreads();                     // every wave has read its B fragments: the tile area is free again
      // tile -> LDS rows (O3 * 2 + 16 bytes) -> whole rows stored 16 B per lane (8-byte stores straight from the MFMA
      // layout cost more than the GEMM)
      const int rowb = O3 * 2 + 16;
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (j < nper) {
#pragma unroll
          for (int rq = 0; rq < 4; rq++) {
            using h4 = __attribute__((ext_vector_type(4))) _Float16;
            const int och = mt * 32 + 8 * rq + 4 * (lane >> 5);
            const h4 bq = *reinterpret_cast<const h4*>(ex.chain_b + och);
            h2e lo = {(_Float16)(c2[j][rq * 4] + (float)bq[0]), (_Float16)(c2[j][rq * 4 + 1] + (float)bq[1])};
            h2e hi = {(_Float16)(c2[j][rq * 4 + 2] + (float)bq[2]), (_Float16)(c2[j][rq * 4 + 3] + (float)bq[3])};
            lo = __builtin_elementwise_max(lo, h2e{(_Float16)0.f, (_Float16)0.f});
            hi = __builtin_elementwise_max(hi, h2e{(_Float16)0.f, (_Float16)0.f});
            const h4 v4 = {lo[0], lo[1], hi[0], hi[1]};
            *reinterpret_cast<h4*>(s_out + (32 * (nt0 + j) + (lane & 31)) * rowb + och * 2) = v4;
          }
        }
      __syncthreads();
      const int vpr = O3 / 8;                                          // 16-byte vectors per row: 8 | 16
      for (int idx = tid; idx < Cfg::kPos * vpr; idx += kThreads_) {
        const int pos = idx / vpr, col = idx % vpr;
        const int64_t gp = tile_pos(tile, pos, Cfg::kTH, Ho, Wo, HWo, Ntot);
        if (gp >= 0)
          *reinterpret_cast<V*>(ex.chain_out + gp * O3 + col * 8) = *reinterpret_cast<const V*>(s_out + pos * rowb + col * 16);
      }
    }
    return;
  }
  constexpr int VPR = 8 * OG;                     // 16-byte vectors per output row
  constexpr int NI = (Cfg::kPos * VPR) / kThreads_;
  const bool relu_u = __builtin_amdgcn_readfirstlane(relu) != 0;
  if (residual) {
    // all residual vectors of the tile in flight at once (bounds-checked buffer loads: no branch
    // around a load, so the compiler does not wait for each one before issuing the next)
    // res_up: the residual is a half-resolution map [B,Ho/2,Wo/2,O] added through a nearest 2x
    // up-sampling (the FPN top-down pathway, models/neck.py:73-79) -- only its row index differs
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T*>(residual), 0, (int)((uint64_t)(res_up ? Ntot / 4 : Ntot) * O * 2), 0x00020000);
    unsigned off[NI];
    V r[NI];
    const int Wr = Wo >> 1;
    const int64_t HWr = (int64_t)(Ho >> 1) * Wr;
#pragma unroll
    for (int i = 0; i < NI; i++) {
      int idx = tid + kThreads_ * i, pos = idx / VPR, col = idx % VPR;
      int64_t gp = TAPS == 9 ? tile_pos(tile, pos, Cfg::kTH, Ho, Wo, HWo, Ntot) : (g0 + pos < Ntot ? g0 + pos : -1);
      const bool ok = gp >= 0 && col * 8 < Oloc;
      off[i] = ok ? (unsigned)((gp * O + o0 + col * 8) * 2) : 0x80000000u;
      unsigned roff = off[i];
      if (res_up && ok) {
        const int64_t bb = gp / HWo, rem = gp % HWo;
        const int yy = (int)(rem / Wo), xx = (int)(rem % Wo);
        roff = (unsigned)(((bb * HWr + (int64_t)(yy >> 1) * Wr + (xx >> 1)) * O + o0 + col * 8) * 2);
      }
      r[i] = __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(rr, (int)roff, 0, 0));
    }
#pragma unroll
    for (int i = 0; i < NI; i++) {
      int idx = tid + kThreads_ * i, pos = idx / VPR, col = idx % VPR;
      V v = *reinterpret_cast<const V*>(s_out + pos * Cfg::kOutRowB + col * 16);
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        h2e p2 = {(_Float16)((float)v[e] + (float)r[i][e]), (_Float16)((float)v[e + 1] + (float)r[i][e + 1])};
        if (relu_u) p2 = __builtin_elementwise_max(p2, h2e{(_Float16)0.f, (_Float16)0.f});   // (on the rounded halves: see above)
        v[e] = p2[0];
        v[e + 1] = p2[1];
      }
      if (off[i] != 0x80000000u) *reinterpret_cast<V*>(reinterpret_cast<char*>(out) + off[i]) = v;
    }
  } else {
#pragma unroll
    for (int i = 0; i < NI; i++) {
      int idx = tid + kThreads_ * i, pos = idx / VPR, col = idx % VPR;
      int64_t gp = TAPS == 9 ? tile_pos(tile, pos, Cfg::kTH, Ho, Wo, HWo, Ntot) : (g0 + pos < Ntot ? g0 + pos : -1);
      if (gp >= 0 && col * 8 < Oloc && ex.store_main)
        *reinterpret_cast<V*>(out + gp * O + o0 + col * 8) = *reinterpret_cast<const V*>(s_out + pos * Cfg::kOutRowB + col * 16);
    }
  }
  // optional: a 1x1 prediction head (<= 32 maps: fam_reg_head / fam_cls_head, models/head.py:205-213) applied to
  // the staged tile -- every wave takes 32 positions, B fragments straight from the staged rows (528-byte stride:
  // conflict-free), A fragments = the head's filter (16 KB, L2-resident), 16 MFMAs.  Needs the whole channel range
  // in this workgroup (OG = 4, O = 256).
  if constexpr (OG == 4) {
    if (ex.head_w) {
      f32x16 hacc;
#pragma unroll
      for (int r = 0; r < 16; r++) hacc[r] = 0.f;
      const int hpos = 32 * wave + (lane & 31);
      const V* hw = reinterpret_cast<const V*>(ex.head_w) + lane;
#pragma unroll
      for (int c4 = 0; c4 < 4; c4++)
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
          const V a = hw[(c4 * 8 + kk) * 64];                 // stage c4, m-tile 0 (maps 0..31), k-step kk
          const V b = *reinterpret_cast<const V*>(s_out + hpos * Cfg::kOutRowB + (c4 * 64 + kk * 16 + (lane >> 5) * 8) * 2);
          hacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, hacc, 0, 0, 0);
        }
      const int64_t gp = TAPS == 9 ? tile_pos(tile, hpos, Cfg::kTH, Ho, Wo, HWo, Ntot) : (g0 + hpos < Ntot ? g0 + hpos : -1);
      if (gp >= 0) {
        using h4 = __attribute__((ext_vector_type(4))) _Float16;
#pragma unroll
        for (int rq = 0; rq < 4; rq++) {
          const int och = 8 * rq + 4 * (lane >> 5);
          const h4 hb = *reinterpret_cast<const h4*>(ex.head_b + och);
          h4 v4;
#pragma unroll
          for (int e = 0; e < 4; e++) v4[e] = (_Float16)(hacc[rq * 4 + e] + (float)hb[e]);
          *reinterpret_cast<h4*>(ex.head_out + gp * 64 + och) = v4;
        }
      }
    }
  }
  // optional second output: rotation-invariant pooling of the tile just produced (max over each run of 8
  // orientation channels, models/orn/functions/rotation_invariant_pooling.py:19-27) straight from the staged
  // tile -- ORConv2d's output feeds both the regression tower (full tile, stored above) and, pooled, the
  // classification tower.  One item = one position x 8 pooled channels (128 B read, 16 B stored).
  if (pool_out_) {
    constexpr int GP = 8 * OG;                      // pooled channels of this workgroup's 64*OG outputs
    for (int item = tid; item < Cfg::kPos * (GP / 8); item += kThreads_) {
      const int pos = item / (GP / 8), q = item % (GP / 8);
      const int64_t gp = TAPS == 9 ? tile_pos(tile, pos, Cfg::kTH, Ho, Wo, HWo, Ntot) : (g0 + pos < Ntot ? g0 + pos : -1);
      if (gp < 0 || q * 64 >= Oloc) continue;
      V res;
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const V v = *reinterpret_cast<const V*>(s_out + pos * Cfg::kOutRowB + (q * 8 + e) * 16);
        _Float16 mx = v[0];
#pragma unroll
        for (int k = 1; k < 8; k++) mx = v[k] > mx ? v[k] : mx;
        res[e] = mx;
      }
      *reinterpret_cast<V*>(pool_out_ + gp * (O / 8) + o0 / 8 + q * 8) = res;
    }
  }
#if S2A_STAMP
  if ((Cfg::kWLds && OG == 4) || (TAPS == 1 && OG == 4 && SD == 1)) S2A_STAMP_AT(5);
#endif
}

// CU count of the CURRENT device (cached per device index: a process may drive devices of different size)
inline int device_cu_count(int* out) {
  static int cached[64] = {};
  int dev = 0;
  S2A_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || cached[dev] == 0) {
    hipDeviceProp_t prop;
    S2A_HIP(hipGetDeviceProperties(&prop, dev));
    const int n = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (dev >= 0 && dev < 64) __atomic_store_n(&cached[dev], n, __ATOMIC_RELAXED);   // (racing writers store the same value)
    *out = n;
    return S2A_OK;
  }
  *out = cached[dev];
  return S2A_OK;
}

// ------------------------------------------------------------------ f32 AlignConv / deformable conv on the 16-bit matrix instruction
// The f32 matrix instruction runs at 1/16 of the 16-bit rate (k_dcn_mfma<float>: 0.55 of ITS peak = 225 us for one
// [1,256,128,128] level).  An f32 value is EXACTLY the sum of three bf16 values (hi = rne(x), mid = rne(x - hi), lo = rne(x - hi -
// mid): 8 + 8 + 8 significand bits), and a product without its three smallest cross terms is six bf16 products accumulated
// in f32 (|error| <~ 3 * 2^-24 of the product: one f32 rounding) -- six v_mfma_f32_32x32x16_bf16 for eight 32x32x2 f32 ones of
// a quarter their length.  k_dcn_x3 is k_dcn_mfma's dataflow with that arithmetic: per stage (tap, 16 input channels) the
// four bilinear corners of every position are gathered from the NHWC input, blended in f32 (im2col_bilinear's order) and
// written to LDS as three bf16 planes; the filter was split into its planes when it was packed (k_pack_weight_x3); the waves
// read both operands with ds_read_b128 (48-byte rows: conflict-free) and run 6 MFMAs per output tile and stage.
// S2A_DCN_F32=mfma32 selects the f32 instruction's kernel (A/B, and the tests' cross-check).
constexpr int kX3KC = 16;                       // input channels per stage = one k-step of 32x32x16
constexpr int kX3Row = 48;                      // bytes per LDS row: 16 bf16 + 16 B pad (12 dwords: sixteen rows start in sixteen 4-bank groups)
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
template <int NPOS>
constexpr int x3_lds_bytes() { return NPOS * 9 * 32 + 2 * 3 * (kMaxO + NPOS) * kX3Row; }      // 110 592 / 147 456
// weight [O][C][9] f32 -> [stage s = (c / 16) * 9 + tap][plane][O][16] bf16
__global__ void k_pack_weight_x3(const float* __restrict__ w, int O, int C, __bf16* __restrict__ wp) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)O * C * 9) return;
  const int t = (int)(e % 9);
  const int64_t oc = e / 9;
  const int c = (int)(oc % C), o = (int)(oc / C);
  const int64_t st = (int64_t)(c / kX3KC) * 9 + t;
  __bf16 h, m, l;
  split3(w[e], h, m, l);
  const int64_t base = (st * 3 * O + o) * kX3KC + (c % kX3KC), plane = (int64_t)O * kX3KC;
  wp[base] = h;
  wp[base + plane] = m;
  wp[base + 2 * plane] = l;
}

template <bool OUT_NHWC, int SRC, int NPOS>
__global__ __launch_bounds__(512, 1) void k_dcn_x3(const float* __restrict__ x,       // NHWC
                                                   const float* __restrict__ src,     // offsets | anchors
                                                   const __bf16* __restrict__ wp,     // k_pack_weight_x3
                                                   float* __restrict__ out, int64_t Ntot, int C, int H,
                                                   int W, int O, float stride, int relu) {
  // eight waves, two per SIMD (one wave's blend / split / requests run under the other's MFMAs): wave = (64-out-channel group
  // wave & 3, position half wave >> 2)
  constexpr int NT = NPOS / 64;        // 32-wide position tiles per wave
  constexpr int kItems = NPOS * 4;     // (position, 4-channel group) items per stage: one per thread (NPOS 64: threads 0-255)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // [ taps NPOS*9*32 B | A0 (3 planes) | B0 (3 planes) | A1 | B1 ]
  Tap* s_tab = reinterpret_cast<Tap*>(smem);
  constexpr int kTabBytes = NPOS * 9 * 32;
  constexpr int kAPlane = kMaxO * kX3Row, kBPlane = NPOS * kX3Row;
  constexpr int kBufBytes = 3 * (kAPlane + kBPlane);
  char* s_buf = smem + kTabBytes;

  const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, phalf = tid >> 8;
  const int64_t HW = (int64_t)H * W;
  const int64_t tile = xcd_remap(blockIdx.x, gridDim.x);
  const int o0 = blockIdx.y * kMaxO;
  const int Oloc = min(kMaxO, O - o0);
  const int CC = C / kX3KC;
  const int nstage = 9 * CC;

  // ---- sampling table for this tile (as k_dcn_mfma)
  for (int e = tid; e < NPOS * 9; e += 512) {
    int pl = e / 9, t = e % 9;
    int64_t g = tile_pos(tile, pl, NPOS / 16, H, W, HW, Ntot);
    Tap tp;
    if (g >= 0) {
      int64_t b = g / HW, p = g % HW;
      int y = (int)(p / W), xq = (int)(p % W);
      int ky = t / 3, kx = t % 3;
      float off_y, off_x;
      if (SRC == 0) {
        const float* ob = src + (b * 18) * HW + p;
        off_y = ob[(int64_t)(2 * t) * HW];
        off_x = ob[(int64_t)(2 * t + 1) * HW];
      } else {
        AnchorCtx c = anchor_ctx(src + g * 5, stride);
        anchor_offset(c, ky, kx, (float)y, (float)xq, off_y, off_x);
      }
      float h_im = (float)(y - 1 + ky) + off_y;  // kernel.cu:226-227
      float w_im = (float)(xq - 1 + kx) + off_x;
      tp = make_tap(h_im, w_im, H, W, b * HW);
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        tp.idx[k] = 0;
        tp.w[k] = 0.f;
      }
    }
    s_tab[e] = tp;
  }
  __syncthreads();

  f32x16 acc[2][NT];
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < NT; b++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

  // per-thread staging registers, TWO sets: one wave per SIMD runs a stage's 24-48 MFMAs in 0.8-1.5 k cycles, less than an L2
  // round trip under load, so a stage's requests go out TWO stages ahead (one stage ahead: 202 us for one [1,256,128,128]
  // level against 227 on the f32 instruction; the matrix work is 46 us)
  struct Stage {
    f32x4 cv[4];
    float cw[4];
    bf16x8 av[3];                      // 3 planes x 256 rows x two 16-byte halves = 1 536 vectors
  };
  Stage ra, rb;

  auto issue = [&](int s, Stage& r_) {
    const int t = s % 9, cc = s / 9;
    if (tid < kItems) {
      const int pl = tid >> 2, q = tid & 3;
      const Tap tp = s_tab[pl * 9 + t];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        r_.cw[k] = tp.w[k];
        r_.cv[k] = *reinterpret_cast<const f32x4*>(x + (int64_t)tp.idx[k] * C + cc * kX3KC + q * 4);
      }
    }
    const __bf16* wsrc = wp + (int64_t)s * 3 * O * kX3KC;
#pragma unroll
    for (int r = 0; r < 3; r++) {
      int idx = tid + 512 * r;
      int plane = idx >> 9, row = (idx & 511) >> 1, q = idx & 1;
      if (row < Oloc) r_.av[r] = *reinterpret_cast<const bf16x8*>(wsrc + ((int64_t)plane * O + o0 + row) * kX3KC + q * 8);
    }
  };
  auto commit = [&](int buf, const Stage& r_) {
    char* A = s_buf + buf * kBufBytes;
    char* Bm = A + 3 * kAPlane;
    if (tid < kItems) {
      const int pl = tid >> 2, q = tid & 3;
      const f32x4 v = blend<float>(r_.cv, r_.cw);
      bf16x4 h, m, l;
#pragma unroll
      for (int j = 0; j < 4; j++) { __bf16 a, b2, c; split3(v[j], a, b2, c); h[j] = a; m[j] = b2; l[j] = c; }
      char* d = Bm + pl * kX3Row + q * 8;
      *reinterpret_cast<bf16x4*>(d) = h;
      *reinterpret_cast<bf16x4*>(d + kBPlane) = m;
      *reinterpret_cast<bf16x4*>(d + 2 * kBPlane) = l;
    }
#pragma unroll
    for (int r = 0; r < 3; r++) {
      int idx = tid + 512 * r;
      int plane = idx >> 9, row = (idx & 511) >> 1, q = idx & 1;
      if (row < Oloc) *reinterpret_cast<bf16x8*>(A + plane * kAPlane + row * kX3Row + q * 16) = r_.av[r];
    }
  };

  const bool wave_active = wave * 64 < Oloc;
  auto mfma_stage = [&](int buf) {
    if (!wave_active) return;
    const char* A = s_buf + buf * kBufBytes;
    const char* Bm = A + 3 * kAPlane;
    const char* wrow = A + (wave * 64 + (lane & 31)) * kX3Row + (lane >> 5) * 16;
    const char* prow = Bm + (phalf * NT * 32 + (lane & 31)) * kX3Row + (lane >> 5) * 16;
    bf16x8 wf[3][2], pf[3][NT];
#pragma unroll
    for (int p = 0; p < 3; p++) {
#pragma unroll
      for (int h = 0; h < 2; h++) wf[p][h] = *reinterpret_cast<const bf16x8*>(wrow + p * kAPlane + h * 32 * kX3Row);
#pragma unroll
      for (int h = 0; h < NT; h++) pf[p][h] = *reinterpret_cast<const bf16x8*>(prow + p * kBPlane + h * 32 * kX3Row);
    }
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int b = 0; b < NT; b++) {
        // (filter plane, column plane): the three small terms first
        constexpr int kPw[6] = {2, 0, 1, 1, 0, 0}, kPp[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int k = 0; k < 6; k++) {
          if constexpr (OUT_NHWC)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf[kPp[k]][b], wf[kPw[k]][a], acc[a][b], 0, 0, 0);
          else
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[kPw[k]][a], pf[kPp[k]][b], acc[a][b], 0, 0, 0);
        }
      }
  };
  // stage s: requests of stage s + 2 (into the set stage s used), MFMAs of stage s, operands of stage s + 1 into the other buffer.
  // Measured and not kept (batch 8 / one image, 914 / 142 us as shipped): the two waves of a SIMD taking the halves of a stage in
  // opposite order (one blends while the other holds the matrix pipe) 973 / 140 -- the branch doubles the loop body; stages tap
  // by tap with the corner addresses hoisted (nine table reads per tile instead of 144) 984 / 156 -- chunk by chunk the nine
  // taps of a 16-channel chunk re-touch the same 64-byte pieces of neighbouring pixels while they are still in the L1.
  auto step = [&](int s, Stage& r_issue, const Stage& r_commit) {
    if (s + 2 < nstage) issue(s + 2, r_issue);
    mfma_stage(s & 1);
    if (s + 1 < nstage) commit((s + 1) & 1, r_commit);
    __syncthreads();
  };
  issue(0, ra);
  if (nstage > 1) issue(1, rb);
  commit(0, ra);
  __syncthreads();
  for (int s = 0; s < nstage; s += 2) {
    step(s, ra, rb);
    if (s + 1 < nstage) step(s + 1, rb, ra);
  }

  if (!wave_active) return;
  // ---- epilogue: ReLU + store.  acc[a][b] = (out-channel tile a, position tile b);
  // MFMA D layout: column = lane&31, row = (r&3)+8*(r>>2)+4*(lane>>5)
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < NT; b++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        float v = acc[a][b][r];
        if (relu) v = fmaxf(v, 0.f);
        int rowi = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if constexpr (OUT_NHWC) {
          int64_t g = tile_pos(tile, 32 * (phalf * NT + b) + rowi, NPOS / 16, H, W, HW, Ntot);
          int och = o0 + wave * 64 + 32 * a + (lane & 31);
          if (g >= 0) out[g * O + och] = v;
        } else {
          int och = o0 + wave * 64 + 32 * a + rowi;
          int64_t g = tile_pos(tile, 32 * (phalf * NT + b) + (lane & 31), NPOS / 16, H, W, HW, Ntot);
          if (g >= 0) {
            int64_t bi = g / HW, p = g % HW;
            out[(bi * O + och) * HW + p] = v;
          }
        }
      }
}

template <int NPOS>
constexpr int mfma_lds_bytes() { return NPOS * 9 * 32 + 2 * (kMaxO + NPOS) * kRowBytes; }  // 110592 / 147456

// ------------------------------------------------------------------ generic fallback
// Any stride / padding / dilation / groups / deformable groups / channel count, NCHW only.
// One thread per output element; used when the shape is not the AlignConv fast-path shape.
// mask / bias: the modulated form (DCNv2: modulated_deformable_im2col_gpu_kernel, deform_conv_cuda_kernel.cu:570-632:
// every sampled value is multiplied by mask[b, dg, tap, ho, wo]; bias added after the contraction,
// deform_conv_cuda.cpp:566-568); both null = plain deformable convolution.
// arithmetic type of the generic kernel: the reference instantiates its kernels per scalar type
// (AT_DISPATCH_FLOATING_TYPES_AND_HALF, deform_conv_cuda_kernel.cu:258,352,450): double stays double; float and half compute in
// float here (half columns rounded once, as the fused f16 path)
template <typename T>
struct GenAcc { using type = float; };
template <>
struct GenAcc<double> { using type = double; };

template <typename T, typename TO>
__global__ __launch_bounds__(256) void k_dcn_generic(const T* __restrict__ x, const TO* __restrict__ offset,
                                                     const T* __restrict__ w, T* __restrict__ out,
                                                     s2a_dcn_params p, int Ho, int Wo,
                                                     const T* __restrict__ mask = nullptr,
                                                     const T* __restrict__ bias = nullptr) {
  using A = typename GenAcc<T>::type;
  const int64_t total = p.batch * p.out_channels * Ho * Wo;
  const int Cg = (int)(p.channels / p.group), Og = (int)(p.out_channels / p.group);
  const int cpdg = (int)(p.channels / p.deformable_group);
  const int H = (int)p.height, W = (int)p.width;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    int wo = (int)(e % Wo);
    int64_t r = e / Wo;
    int ho = (int)(r % Ho);
    r /= Ho;
    int o = (int)(r % p.out_channels);
    int64_t b = r / p.out_channels;
    int g = o / Og;
    A acc = 0;
    for (int cl = 0; cl < Cg; cl++) {
      int c = g * Cg + cl;
      int dg = c / cpdg;
      const T* plane = x + (b * p.channels + c) * (int64_t)H * W;
      const TO* offp = offset + (b * p.deformable_group + dg) * 2 * p.kH * p.kW * (int64_t)Ho * Wo;
      const T* maskp = mask ? mask + (b * p.deformable_group + dg) * p.kH * p.kW * (int64_t)Ho * Wo : nullptr;
      for (int i = 0; i < p.kH; i++)
        for (int j = 0; j < p.kW; j++) {
          int t = i * p.kW + j;
          A oh = (A)offp[((int64_t)(2 * t) * Ho + ho) * Wo + wo];
          A ow = (A)offp[((int64_t)(2 * t + 1) * Ho + ho) * Wo + wo];
          A h_im = (A)(ho * p.dH - p.padH + i * p.dilationH) + oh;
          A w_im = (A)(wo * p.dW - p.padW + j * p.dilationW) + ow;
          A v = 0;
          if (h_im > -1 && w_im > -1 && h_im < H && w_im < W) {
            int h_low = (int)floor(h_im), w_low = (int)floor(w_im);
            int h_high = h_low + 1, w_high = w_low + 1;
            A lh = h_im - h_low, lw = w_im - w_low, hh = 1 - lh, hw = 1 - lw;
            A v1 = 0, v2 = 0, v3 = 0, v4 = 0;
            if (h_low >= 0 && w_low >= 0) v1 = (A)plane[h_low * W + w_low];
            if (h_low >= 0 && w_high <= W - 1) v2 = (A)plane[h_low * W + w_high];
            if (h_high <= H - 1 && w_low >= 0) v3 = (A)plane[h_high * W + w_low];
            if (h_high <= H - 1 && w_high <= W - 1) v4 = (A)plane[h_high * W + w_high];
            v = hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4;
          }
          if (maskp) v *= (A)maskp[((int64_t)t * Ho + ho) * Wo + wo];
          if (sizeof(T) == 2) v = (A)(T)v;  // f16 columns, as the f16 fast path
          acc += (A)w[(((int64_t)o * Cg + cl) * p.kH + i) * p.kW + j] * v;
        }
    }
    if (bias) acc += (A)bias[o];
    if (p.relu) acc = acc > 0 ? acc : (A)0;
    out[e] = (T)acc;
  }
}

inline bool fast_path_ok(const s2a_dcn_params& p) {
  if (p.dtype == S2A_DTYPE_F64) return false;       // float64: the generic kernel (API completeness, not a hot path)
  int kc = p.dtype == S2A_DTYPE_F32 ? 32 : 64;
  return p.kW == 3 && p.kH == 3 && p.dW == 1 && p.dH == 1 && p.padW == 1 && p.padH == 1 &&
         p.dilationW == 1 && p.dilationH == 1 && p.group == 1 && p.deformable_group == 1 &&
         p.channels % kc == 0 && p.out_channels % 64 == 0 && p.offset_dtype == S2A_DTYPE_F32 &&
         p.batch * p.height * p.width < (1ll << 31);
}

inline size_t esize(int dtype) { return dtype == S2A_DTYPE_F64 ? 8 : (dtype == S2A_DTYPE_F32 ? 4 : 2); }

inline bool half_coords_requested() {
  const char* f = getenv("S2A_DCN_HALF_COORDS");
  return f && atoi(f) != 0;
}

template <typename T>
int launch_fast(const T* x_nhwc, const float* src, bool from_anchors, const T* wp, const T* wfrag,
                T* out, bool out_nhwc, int64_t B, int C, int H, int W, int O, float stride, int relu,
                hipStream_t st) {
  const int64_t Ntot = B * (int64_t)H * W;
  // 128-position tiles once they still give every CU >= 2 workgroups; else 64
  auto ntiles = [&](int npos) { return B * (int64_t)((W + 15) / 16) * ((H + npos / 16 - 1) / (npos / 16)); };
  const bool big = ntiles(128) >= 512;
  if constexpr (sizeof(T) == 4) {
    // f32: the three-bf16-plane kernel (its filter planes sit behind the f32 stage layout in the packed buffer) unless
    // S2A_DCN_F32=mfma32 asks for the f32 matrix instruction
    const char* sel = getenv("S2A_DCN_F32");
    if (!(sel && sel[0] == 'm')) {
      const __bf16* wx3 = reinterpret_cast<const __bf16*>(wp + (size_t)O * C * 9);
#define S2A_DCN_LAUNCH_X3(NHWC, SRC, NPOS)                                                        \
      do {                                                                                        \
        auto kern = k_dcn_x3<NHWC, SRC, NPOS>;                                                    \
        constexpr int lds = x3_lds_bytes<NPOS>();                                                 \
        dim3 grid((unsigned)ntiles(NPOS), (unsigned)((O + kMaxO - 1) / kMaxO));                   \
        S2A_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                          \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, lds));            \
        kern<<<grid, 512, lds, st>>>(x_nhwc, src, wx3, out, Ntot, C, H, W, O, stride, relu);      \
      } while (0)
#define S2A_DCN_PICK_X3(NHWC, SRC) do { if (big) S2A_DCN_LAUNCH_X3(NHWC, SRC, 128); else S2A_DCN_LAUNCH_X3(NHWC, SRC, 64); } while (0)
      if (out_nhwc) {
        if (from_anchors) S2A_DCN_PICK_X3(true, 1); else S2A_DCN_PICK_X3(true, 0);
      } else {
        if (from_anchors) S2A_DCN_PICK_X3(false, 1); else S2A_DCN_PICK_X3(false, 0);
      }
#undef S2A_DCN_PICK_X3
#undef S2A_DCN_LAUNCH_X3
      S2A_LAUNCH_CHECK();
      return S2A_OK;
    }
  }
#define S2A_DCN_LAUNCH(NHWC, SRC, NPOS)                                                           \
  do {                                                                                            \
    auto kern = k_dcn_mfma<T, NHWC, SRC, NPOS>;                                                   \
    constexpr int lds = mfma_lds_bytes<NPOS>();                                                   \
    dim3 grid((unsigned)ntiles(NPOS), (unsigned)((O + kMaxO - 1) / kMaxO));                       \
    S2A_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                              \
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds));                \
    kern<<<grid, 256, lds, st>>>(x_nhwc, src, wp, out, Ntot, C, H, W, O, stride, relu);           \
  } while (0)
  const uint64_t x_bytes = (uint64_t)Ntot * C * sizeof(T), w_bytes = (uint64_t)O * C * 9 * sizeof(T);
  const bool ws_ok = sizeof(T) == 2 && big && x_bytes < (1ull << 31) && w_bytes < (1ull << 31) && !getenv("S2A_DCN_NO_WS");
#define S2A_DCN_LAUNCH_WS(NHWC, SRC)                                                              \
  do {                                                                                            \
    auto kern = k_dcn_ws<T, NHWC, SRC>;                                                           \
    constexpr int lds = mfma_lds_bytes<128>();                                                    \
    dim3 grid((unsigned)ntiles(128), (unsigned)((O + kMaxO - 1) / kMaxO));                        \
    S2A_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                              \
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds));                \
    /* timing-only experiment (cdna guide, rocprof section): a zero-record descriptor drops every  \
       buffer load through it while the instruction stream stays; outputs are then wrong */       \
    unsigned xb = (unsigned)x_bytes, wb = (unsigned)w_bytes;                                      \
    S2A_DCN_DROP_SWITCH(xb, wb);                                                                  \
    kern<<<grid, 512, lds, st>>>(x_nhwc, src, wp, out, Ntot, C, H, W, O, stride, relu, xb, wb);   \
  } while (0)
  const char* variant = getenv("S2A_DCN_VARIANT");   // A/B switch for measurements: "ws" | "mfma"
  bool patch_ok = false;
  if constexpr (sizeof(T) == 2)
    patch_ok = ntiles(128) >= 128 && x_bytes < (1ull << 31) && !getenv("S2A_DCN_NO_WS") && wfrag != nullptr && C % 64 == 0 && H < 32000 && W < 32000 &&
               !(variant && (!strcmp(variant, "ws") || !strcmp(variant, "mfma")));
  const bool ws_use = ws_ok && !(variant && !strcmp(variant, "mfma"));
  // One 153 KB workgroup per CU: a launch whose 8 x 16 tiles fill the last round badly (one P3 level of ONE chip: 128
  // tiles on 256 CUs, BASELINE configs[1]) runs as 4 x 16 half tiles instead when that needs less time (bit-identical)
  int n_cu_fast = 0;
  {
    int rc_ = device_cu_count(&n_cu_fast);
    if (rc_ != S2A_OK) return rc_;
  }
  // rounds of one workgroup per CU; a half tile takes ~0.62 of a full one (measured: 128 tiles 44.5 -> 34.8 us as 256 half
  // tiles, but 256 tiles -- exactly one full round -- 50 -> 61 us as 512 half tiles)
  const int64_t rounds_full = (ntiles(128) + n_cu_fast - 1) / n_cu_fast, rounds_half = (2 * ntiles(128) + n_cu_fast - 1) / n_cu_fast;
  const bool half_tiles = 62 * rounds_half < 100 * rounds_full && !getenv("S2A_DCN_NO_HALF");
  // S2A_DCN_HALF_COORDS=1: sampling coordinates and weights rounded as the reference's Half instantiation rounds them
  // (f16 inputs only -- the f32 instantiation keeps f32 coordinates; built into the patch-staged kernel alone)
  const int hc_bit = half_coords_requested() && sizeof(T) == 2 ? 2 : 0;
  if (hc_bit && !patch_ok) {
    set_error("deform_conv: S2A_DCN_HALF_COORDS=1 needs the patch-staged f16 kernel (>= 128 position tiles, C %% 64 == 0)");
    return S2A_ENOTIMPL;
  }
#define S2A_DCN_LAUNCH_PATCH(NHWC, SRC)                                                           \
  do {                                                                                            \
    if constexpr (sizeof(T) == 2) {                                                               \
      if (half_tiles) {                                                                           \
        constexpr int kHalfLdsF = 64 * 9 * 16 + 2 * 64 * kRowBytes + 2 * (4 + 2 * kHalo) * kPW * 128; \
        auto kern = k_dcn_patch<NHWC, SRC, 4>;                                                    \
        dim3 grid((unsigned)(2 * ntiles(128)), (unsigned)((O + kMaxO - 1) / kMaxO));              \
        S2A_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                          \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, kHalfLdsF));      \
        kern<<<grid, 512, kHalfLdsF, st>>>(PatchArgs{x_nhwc, src, wfrag, out, Ntot, C, H, W, O, stride, (relu ? 1 : 0) | hc_bit, \
                                                     (unsigned)x_bytes, 0, 0u, LevelTab{}});           \
      } else {                                                                                    \
        auto kern = k_dcn_patch<NHWC, SRC>;                                                       \
        dim3 grid((unsigned)ntiles(128), (unsigned)((O + kMaxO - 1) / kMaxO));                    \
        S2A_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                          \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, kPatchLds));      \
        kern<<<grid, 512, kPatchLds, st>>>(PatchArgs{x_nhwc, src, wfrag, out, Ntot, C, H, W, O, stride, (relu ? 1 : 0) | hc_bit, \
                                                     (unsigned)x_bytes, 0, 0u, LevelTab{}});           \
      }                                                                                           \
    }                                                                                             \
  } while (0)
#define S2A_DCN_PICK(NHWC, SRC) do { if (patch_ok) S2A_DCN_LAUNCH_PATCH(NHWC, SRC); else if (ws_use) S2A_DCN_LAUNCH_WS(NHWC, SRC); else if (big) S2A_DCN_LAUNCH(NHWC, SRC, 128); else S2A_DCN_LAUNCH(NHWC, SRC, 64); } while (0)
  if (out_nhwc) {
    if (from_anchors) S2A_DCN_PICK(true, 1); else S2A_DCN_PICK(true, 0);
  } else {
    if (from_anchors) S2A_DCN_PICK(false, 1); else S2A_DCN_PICK(false, 0);
  }
#undef S2A_DCN_PICK
#undef S2A_DCN_LAUNCH_WS
#undef S2A_DCN_LAUNCH_PATCH
#undef S2A_DCN_LAUNCH
  S2A_LAUNCH_CHECK();
  return S2A_OK;
}

template <typename T>
int run_fast(const void* input, const float* src, bool from_anchors, const void* weight, void* output,
             int64_t B, int C, int H, int W, int O, int layout, float stride, int relu, void* ws,
             size_t ws_bytes, hipStream_t st, bool weight_packed = false) {
  constexpr int KC = Traits<T>::KC;
  S2A_CHECK_ARG(((uintptr_t)input % 16) == 0 && ((uintptr_t)output % 16) == 0 && ((uintptr_t)weight % 16) == 0 &&
                ((uintptr_t)ws % 16) == 0, "deform_conv: tensors must be 16-byte aligned");
  Carver cv(ws, ws_bytes);
  const size_t wel = (size_t)O * C * 9;
  const bool has_frag = sizeof(T) == 2;             // f16 packs both layouts back to back; f32: the stage layout, then the three bf16 planes
  T* wp = weight_packed ? const_cast<T*>((const T*)weight) : cv.take<T>(has_frag ? wel * 2 : wel * 5 / 2);
  T* xn = nullptr;
  if (layout == S2A_LAYOUT_NCHW) xn = cv.take<T>((size_t)B * C * H * W);
  if (!wp || (layout == S2A_LAYOUT_NCHW && !xn)) {
    set_error("deform_conv: workspace too small (%zu < %zu)", ws_bytes, cv.off);
    return S2A_EWORKSPACE;
  }
  const int64_t wtot = (int64_t)O * C * 9;
  if (!weight_packed) {
    k_pack_weight<T><<<(unsigned)((wtot + 255) / 256), 256, 0, st>>>((const T*)weight, O, C, KC, wp);
    if constexpr (sizeof(T) == 2) {
      k_pack_weight_frag16<<<(unsigned)((wtot + 255) / 256), 256, 0, st>>>((const _Float16*)weight, O, C, wp + wel);
    } else {
      k_pack_weight_x3<<<(unsigned)((wtot + 255) / 256), 256, 0, st>>>((const float*)weight, O, C, reinterpret_cast<__bf16*>(wp + wel));
    }
  }
  const T* wfrag = has_frag ? wp + wel : nullptr;
  const T* x_nhwc = (const T*)input;
  if (layout == S2A_LAYOUT_NCHW) {
    int64_t HW = (int64_t)H * W;
    dim3 g((unsigned)((HW + 31) / 32), (unsigned)((C + 31) / 32), (unsigned)B);
    k_nchw_to_nhwc<T><<<g, 256, 0, st>>>((const T*)input, B, C, HW, xn);
    x_nhwc = xn;
  }
  return launch_fast<T>(x_nhwc, src, from_anchors, wp, wfrag, (T*)output, layout == S2A_LAYOUT_NHWC, B, C,
                        H, W, O, stride, relu, st);
}

}  // namespace
}  // namespace s2a

using namespace s2a;

extern "C" size_t s2a_deform_conv_workspace_bytes(const s2a_dcn_params* p) {
  if (!p) return 0;
  size_t es = esize(p->dtype);
  size_t b = align_up((size_t)p->out_channels * p->channels * p->kH * p->kW * es * 3) + 256;      // packed filter: f16 two layouts, f32 2.5 x
  if (p->layout == S2A_LAYOUT_NCHW) b += align_up((size_t)p->batch * p->channels * p->height * p->width * es);
  return b;
}

extern "C" int s2a_deform_conv_forward(const void* input, const void* weight, const void* offset,
                                       void* output, const s2a_dcn_params* pp, void* workspace,
                                       size_t workspace_bytes, s2a_stream_t stream) {
  S2A_CHECK_ARG(pp != nullptr, "deform_conv: NULL params");
  const s2a_dcn_params p = *pp;
  // shape_check (deform_conv_cuda.cpp:62-150)
  S2A_CHECK_ARG(p.kW > 0 && p.kH > 0, "kernel size should be greater than zero, but got kH: %d kW: %d", p.kH, p.kW);
  S2A_CHECK_ARG(p.dW > 0 && p.dH > 0, "stride should be greater than zero, but got dH: %d dW: %d", p.dH, p.dW);
  S2A_CHECK_ARG(p.dilationW > 0 && p.dilationH > 0, "dilation should be greater than 0, but got dilationH: %d dilationW: %d", p.dilationH, p.dilationW);
  S2A_CHECK_ARG(p.group > 0 && p.deformable_group > 0, "deform_conv: group counts must be positive");
  S2A_CHECK_ARG(p.channels % p.group == 0 && p.out_channels % p.group == 0, "deform_conv: channels must divide groups");
  S2A_CHECK_ARG(p.channels % p.deformable_group == 0, "input channels must divide deformable group size");
  S2A_CHECK_ARG(p.dtype == S2A_DTYPE_F32 || p.dtype == S2A_DTYPE_F16 || p.dtype == S2A_DTYPE_F64, "deform_conv: dtype");
  S2A_CHECK_ARG(p.dtype == S2A_DTYPE_F64 ? p.offset_dtype == S2A_DTYPE_F64 : (p.offset_dtype == S2A_DTYPE_F32 || p.offset_dtype == p.dtype),
                "deform_conv: offset dtype (float64 input: float64 offsets; otherwise float32 or the input's)");
  const int64_t Ho = (p.height + 2 * p.padH - (p.dilationH * (p.kH - 1) + 1)) / p.dH + 1;
  const int64_t Wo = (p.width + 2 * p.padW - (p.dilationW * (p.kW - 1) + 1)) / p.dW + 1;
  S2A_CHECK_ARG(Ho >= 1 && Wo >= 1, "Given input size: (%lld x %lld x %lld). Calculated output size: (%lld x %lld x %lld). Output size is too small",
                (long long)p.channels, (long long)p.height, (long long)p.width, (long long)p.out_channels, (long long)Ho, (long long)Wo);
  S2A_CHECK_ARG(p.height >= p.kH && p.width >= p.kW, "input image is smaller than kernel");
  if (p.batch == 0) return S2A_OK;
  S2A_CHECK_ARG(input && weight && offset && output, "deform_conv: NULL tensor");
  hipStream_t st = as_stream(stream);
  if (fast_path_ok(p)) {
    if (p.dtype == S2A_DTYPE_F32)
      return run_fast<float>(input, (const float*)offset, false, weight, output, p.batch, (int)p.channels,
                             (int)p.height, (int)p.width, (int)p.out_channels, p.layout, 1.f, p.relu,
                             workspace, workspace_bytes, st);
    return run_fast<_Float16>(input, (const float*)offset, false, weight, output, p.batch, (int)p.channels,
                              (int)p.height, (int)p.width, (int)p.out_channels, p.layout, 1.f, p.relu,
                              workspace, workspace_bytes, st);
  }
  S2A_CHECK_ARG(p.layout == S2A_LAYOUT_NCHW, "deform_conv: the generic path supports NCHW only");
  const int64_t total = p.batch * p.out_channels * Ho * Wo;
  unsigned g = (unsigned)std::min<int64_t>((total + 255) / 256, 65535);
  if (p.dtype == S2A_DTYPE_F64)
    k_dcn_generic<double, double><<<g, 256, 0, st>>>((const double*)input, (const double*)offset, (const double*)weight, (double*)output, p, (int)Ho, (int)Wo);
  else if (p.dtype == S2A_DTYPE_F32)
    k_dcn_generic<float, float><<<g, 256, 0, st>>>((const float*)input, (const float*)offset, (const float*)weight, (float*)output, p, (int)Ho, (int)Wo);
  else if (p.offset_dtype == S2A_DTYPE_F32)
    k_dcn_generic<_Float16, float><<<g, 256, 0, st>>>((const _Float16*)input, (const float*)offset, (const _Float16*)weight, (_Float16*)output, p, (int)Ho, (int)Wo);
  else
    k_dcn_generic<_Float16, _Float16><<<g, 256, 0, st>>>((const _Float16*)input, (const _Float16*)offset, (const _Float16*)weight, (_Float16*)output, p, (int)Ho, (int)Wo);
  S2A_LAUNCH_CHECK();
  return S2A_OK;
}

extern "C" int s2a_modulated_deform_conv_forward(const void* input, const void* weight, const void* bias,
                                                 const void* offset, const void* mask, void* output,
                                                 const s2a_dcn_params* pp, s2a_stream_t stream) {
  S2A_CHECK_ARG(pp != nullptr, "modulated_deform_conv: NULL params");
  const s2a_dcn_params p = *pp;
  S2A_CHECK_ARG(p.kW > 0 && p.kH > 0 && p.dW > 0 && p.dH > 0 && p.dilationW > 0 && p.dilationH > 0,
                "modulated_deform_conv: kernel size, stride and dilation must be positive");
  S2A_CHECK_ARG(p.group > 0 && p.deformable_group > 0 && p.channels % p.group == 0 && p.out_channels % p.group == 0 &&
                p.channels % p.deformable_group == 0, "Input shape and kernel channels wont match");
  S2A_CHECK_ARG(p.dtype == S2A_DTYPE_F32 || p.dtype == S2A_DTYPE_F16, "modulated_deform_conv: dtype");
  S2A_CHECK_ARG(p.offset_dtype == p.dtype, "modulated_deform_conv: offset and mask share the input dtype");
  S2A_CHECK_ARG(p.layout == S2A_LAYOUT_NCHW, "modulated_deform_conv: NCHW only");
  const int64_t Ho = (p.height + 2 * p.padH - (p.dilationH * (p.kH - 1) + 1)) / p.dH + 1;
  const int64_t Wo = (p.width + 2 * p.padW - (p.dilationW * (p.kW - 1) + 1)) / p.dW + 1;
  S2A_CHECK_ARG(Ho >= 1 && Wo >= 1, "modulated_deform_conv: output size is too small");
  if (p.batch == 0) return S2A_OK;
  S2A_CHECK_ARG(input && weight && offset && mask && output, "modulated_deform_conv: NULL tensor");
  hipStream_t st = as_stream(stream);
  const int64_t total = p.batch * p.out_channels * Ho * Wo;
  unsigned g = (unsigned)std::min<int64_t>((total + 255) / 256, 65535);
  if (p.dtype == S2A_DTYPE_F32)
    k_dcn_generic<float, float><<<g, 256, 0, st>>>((const float*)input, (const float*)offset, (const float*)weight,
                                                   (float*)output, p, (int)Ho, (int)Wo, (const float*)mask, (const float*)bias);
  else
    k_dcn_generic<_Float16, _Float16><<<g, 256, 0, st>>>((const _Float16*)input, (const _Float16*)offset, (const _Float16*)weight,
                                                         (_Float16*)output, p, (int)Ho, (int)Wo, (const _Float16*)mask,
                                                         (const _Float16*)bias);
  S2A_LAUNCH_CHECK();
  return S2A_OK;
}

extern "C" size_t s2a_align_conv_workspace_bytes(const s2a_align_params* p) {
  if (!p) return 0;
  size_t es = esize(p->dtype);
  size_t b = align_up((size_t)p->out_channels * p->channels * 9 * es * 3) + 256;      // packed filter: f16 two layouts, f32 2.5 x
  if (p->layout == S2A_LAYOUT_NCHW) b += align_up((size_t)p->batch * p->channels * p->height * p->width * es);
  return b;
}

extern "C" int s2a_align_conv_forward(const void* x, const float* anchors, const void* weight, void* out,
                                      const s2a_align_params* pp, void* workspace,
                                      size_t workspace_bytes, s2a_stream_t stream) {
  S2A_CHECK_ARG(pp != nullptr, "align_conv: NULL params");
  const s2a_align_params p = *pp;
  S2A_CHECK_ARG(p.dtype == S2A_DTYPE_F32 || p.dtype == S2A_DTYPE_F16, "align_conv: dtype");
  S2A_CHECK_ARG(p.stride > 0, "align_conv: stride must be positive");
  const int kc = p.dtype == S2A_DTYPE_F32 ? 32 : 64;
  S2A_CHECK_ARG(p.channels % kc == 0 && p.out_channels % 64 == 0,
                "align_conv: channels must be a multiple of %d and out_channels of 64", kc);
  S2A_CHECK_ARG(p.height >= 3 && p.width >= 3, "input image is smaller than kernel");
  S2A_CHECK_ARG(p.batch * p.height * p.width < (1ll << 31), "align_conv: too many positions");
  if (p.batch == 0) return S2A_OK;
  S2A_CHECK_ARG(x && anchors && weight && out, "align_conv: NULL tensor");
  hipStream_t st = as_stream(stream);
  if (p.dtype == S2A_DTYPE_F32)
    return run_fast<float>(x, anchors, true, weight, out, p.batch, (int)p.channels, (int)p.height,
                           (int)p.width, (int)p.out_channels, p.layout, p.stride, p.relu, workspace,
                           workspace_bytes, st, p.weight_packed != 0);
  return run_fast<_Float16>(x, anchors, true, weight, out, p.batch, (int)p.channels, (int)p.height,
                            (int)p.width, (int)p.out_channels, p.layout, p.stride, p.relu, workspace,
                            workspace_bytes, st, p.weight_packed != 0);
}

extern "C" int s2a_dcn_pack_weight(const void* weight, int64_t out_channels, int64_t channels,
                                   int dtype, void* packed, s2a_stream_t stream) {
  S2A_CHECK_ARG(dtype == S2A_DTYPE_F32 || dtype == S2A_DTYPE_F16, "dcn_pack_weight: dtype");
  const int kc = dtype == S2A_DTYPE_F32 ? 32 : 64;
  S2A_CHECK_ARG(out_channels > 0 && channels > 0 && channels % kc == 0,
                "dcn_pack_weight: channels must be a multiple of %d", kc);
  S2A_CHECK_ARG(weight && packed, "dcn_pack_weight: NULL tensor");
  const int64_t wtot = out_channels * channels * 9;
  hipStream_t st = as_stream(stream);
  if (dtype == S2A_DTYPE_F32) {
    k_pack_weight<float><<<(unsigned)((wtot + 255) / 256), 256, 0, st>>>((const float*)weight, (int)out_channels, (int)channels, kc, (float*)packed);
    k_pack_weight_x3<<<(unsigned)((wtot + 255) / 256), 256, 0, st>>>((const float*)weight, (int)out_channels, (int)channels,
                                                                    reinterpret_cast<__bf16*>((float*)packed + wtot));
  } else {
    S2A_CHECK_ARG(out_channels % 64 == 0, "dcn_pack_weight: out_channels must be a multiple of 64");
    k_pack_weight<_Float16><<<(unsigned)((wtot + 255) / 256), 256, 0, st>>>((const _Float16*)weight, (int)out_channels, (int)channels, kc, (_Float16*)packed);
    k_pack_weight_frag16<<<(unsigned)((wtot + 255) / 256), 256, 0, st>>>((const _Float16*)weight, (int)out_channels, (int)channels, (_Float16*)packed + wtot);
  }
  S2A_LAUNCH_CHECK();
  return S2A_OK;
}

extern "C" int64_t s2a_dcn_packed_elems(int64_t out_channels, int64_t channels, int dtype) {
  // f16 holds two layouts back to back (stage-major for the LDS-staged kernels, MFMA-fragment order for the patch-staged kernel);
  // f32: the stage-major f32 layout, then the filter's three bf16 planes (k_dcn_x3): 1 + 1.5 elements per weight
  const int64_t wel = out_channels * channels * 9;
  return dtype == S2A_DTYPE_F16 ? wel * 2 : wel * 5 / 2;
}

namespace s2a {
int build_flags_dcn() {
  int f = S2A_ABL & 0xff;
  if (S2A_STAMP) f |= 1 << 9;
#ifdef S2A_MEASURE
  f |= 1 << 10;
#endif
  return f;
}
}  // namespace s2a

extern "C" int s2a_debug_read_stamps(unsigned long long* host_dst, int64_t count) {
#if S2A_STAMP
  S2A_HIP(hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(s2a::g_stamps), (size_t)count * 8));
  return S2A_OK;
#else
  (void)host_dst; (void)count;
  s2a::set_error("s2a_debug_read_stamps: not a diagnostic build");
  return S2A_ENOTIMPL;
#endif
}

namespace s2a {
namespace {
template <int TAPS, int OG, int PH = 1, int SD = 1, bool TAIL = false, int HT = 0>
int launch_conv(const _Float16* x, const _Float16* wfrag, const _Float16* bias, const _Float16* residual,
                _Float16* out, int64_t B, int C, int H, int W, int Ho, int Wo, int cstride, int O, int relu,
                hipStream_t st, const LevelTab* levels = nullptr, int64_t level_tiles = 0, int res_up = 0,
                ConvExtra ex = ConvExtra{nullptr, nullptr, nullptr, nullptr, 1}) {
  using Cfg = ConvCfg<TAPS, OG, PH, SD, TAIL, HT>;
  const int64_t Ntot = B * (int64_t)Ho * Wo;
  int64_t tiles = TAPS == 9 ? B * ((Wo + 15) / 16) * ((Ho + Cfg::kTH - 1) / Cfg::kTH) : (Ntot + Cfg::kPos - 1) / Cfg::kPos;
  LevelTab lt = {};
  if (levels) { lt = *levels; tiles = level_tiles; }
  dim3 grid((unsigned)tiles, (unsigned)((O + 64 * OG - 1) / (64 * OG)));
  auto kern = k_conv_f16<TAPS, OG, PH, SD, TAIL, HT>;
  S2A_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::kLds + Cfg::kBiasBytes));
  kern<<<grid, 256 * PH, Cfg::kLds + Cfg::kBiasBytes, st>>>(x, wfrag, bias, residual, out, Ntot, C, H, W, Ho, Wo, cstride, O, relu,
                                     (unsigned)((uint64_t)B * H * W * C * 2), lt, res_up, ex);
  S2A_LAUNCH_CHECK();
  return S2A_OK;
}
}  // namespace
}  // namespace s2a

extern "C" int s2a_conv_nhwc_f16(const void* x, const void* weight_frag, const void* bias, const void* residual,
                                 void* out, int64_t batch, int64_t channels, int64_t height, int64_t width,
                                 int64_t out_channels, int ksize, int stride, int relu, s2a_stream_t stream) {
  S2A_CHECK_ARG(batch >= 0 && channels > 0 && out_channels > 0 && height > 0 && width > 0, "conv: bad shape");
  S2A_CHECK_ARG(ksize == 3 || ksize == 1, "conv: kernel size must be 1 or 3");
  S2A_CHECK_ARG(stride == 1 || stride == 2, "conv: stride must be 1 or 2");
  S2A_CHECK_ARG(!(ksize == 3 && stride == 2) || out_channels % 128 == 0, "conv: 3x3 stride 2 needs out_channels % 128 == 0");
  S2A_CHECK_ARG((channels % 64 == 0 || channels == 32) && out_channels % 64 == 0,
                "conv: channels must be 32 or a multiple of 64, out_channels a multiple of 64");
  const uint64_t x_bytes = (uint64_t)batch * height * width * channels * 2;
  S2A_CHECK_ARG(x_bytes < (1ull << 31) && (ksize == 1 || (height < 32000 && width < 32000)),
                "conv: input too large for 32-bit offsets");
  S2A_CHECK_ARG(!residual || (uint64_t)batch * ((height - 1) / stride + 1) * ((width - 1) / stride + 1) * out_channels * 2 < (1ull << 31),
                "conv: output too large for the fused residual (32-bit offsets)");
  if (batch == 0) return S2A_OK;
  S2A_CHECK_ARG(x && weight_frag && out, "conv: NULL tensor");
  S2A_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)out % 16) == 0 && ((uintptr_t)weight_frag % 16) == 0 &&
                ((uintptr_t)bias % 8) == 0 && ((uintptr_t)residual % 16) == 0, "conv: tensors must be 16-byte aligned");
  hipStream_t st = as_stream(stream);
  const int Ho = (int)((height - 1) / stride + 1), Wo = (int)((width - 1) / stride + 1);   // k=1,p=0 / k=3,p=1 (s = 1, 2)
  const int og = out_channels % 256 == 0 ? 4 : (out_channels % 128 == 0 ? 2 : 1);
  const _Float16 *X = (const _Float16*)x, *Wf = (const _Float16*)weight_frag, *Bi = (const _Float16*)bias,
                 *R = (const _Float16*)residual;
  _Float16* Y = (_Float16*)out;
#define S2A_CONV(TAPS, OG_) launch_conv<TAPS, OG_>(X, Wf, Bi, R, Y, batch, (int)channels, (int)height, (int)width, Ho, Wo, stride, (int)out_channels, relu, st)
  if (ksize == 3 && stride == 2) {
    if (og == 4)
      return launch_conv<9, 4, 1, 2>(X, Wf, Bi, R, Y, batch, (int)channels, (int)height, (int)width, Ho, Wo, 2,
                                     (int)out_channels, relu, st);
    return launch_conv<9, 2, 1, 2>(X, Wf, Bi, R, Y, batch, (int)channels, (int)height, (int)width, Ho, Wo, 2,
                                   (int)out_channels, relu, st);
  }
  if (ksize == 3 && og < 4 && channels % 64 == 0) {
    // narrow 3x3 layers (the 64- and 128-map conv2 of the trunk's first stages): the four waves of an 8 x 16 tile
    // re-load the same filter fragments (4x / 2x the bytes into the CU); 16 x 16 tiles with the filter of each tap
    // staged once per workgroup through LDS when there are enough tiles to fill the chip.  S2A_CONV_PH_NARROW=1|2
    const int64_t tiles16 = batch * ((Wo + 15) / 16) * ((Ho + 15) / 16) * ((out_channels + 64 * og - 1) / (64 * og));
    int ph = tiles16 >= 256 ? 2 : 1;
    if (const char* f = getenv("S2A_CONV_PH_NARROW")) ph = atoi(f) == 2 ? 2 : 1;
    if (ph == 2) {
      if (og == 2)
        return launch_conv<9, 2, 2>(X, Wf, Bi, R, Y, batch, (int)channels, (int)height, (int)width, Ho, Wo, 1,
                                    (int)out_channels, relu, st);
      return launch_conv<9, 1, 2>(X, Wf, Bi, R, Y, batch, (int)channels, (int)height, (int)width, Ho, Wo, 1,
                                  (int)out_channels, relu, st);
    }
  }
  if (ksize == 3 && og == 4 && channels % 64 == 0) {
    // large maps (FPN's 3x3 on P3: 128^2 at batch 8): 16 x 16 tiles with the filter through LDS, as the pyramid-packed
    // towers -- when they still fill the chip twice.  S2A_CONV_PH=1|2
    const int64_t tiles16 = batch * ((Wo + 15) / 16) * ((Ho + 15) / 16) * (out_channels / 256);
    int ph = tiles16 >= 512 ? 2 : 1;
    if (const char* f = getenv("S2A_CONV_PH")) ph = atoi(f) == 2 ? 2 : 1;
    if (ph == 2)
      return launch_conv<9, 4, 2>(X, Wf, Bi, R, Y, batch, (int)channels, (int)height, (int)width, Ho, Wo, 1,
                                  (int)out_channels, relu, st);
  }
  if (ksize == 3 && og >= 2) {
    // small maps (32^2 at batch 8): fewer 8 x 16 workgroups than CUs -> 4 x 16 tiles (512 -> 512 on 32^2: 54 -> 44 us,
    // 256 -> 256 on 32^2: 28 -> 19 us; once every CU has a workgroup the smaller tile loses: 256 -> 256 on 64^2
    // 39.5 -> 43 us).  S2A_CONV3_HALF=0|1
    const int64_t wgs128 = batch * ((Wo + 15) / 16) * ((Ho + 7) / 8) * ((out_channels + 64 * og - 1) / (64 * og));
    bool half = wgs128 < 256;
    if (const char* f = getenv("S2A_CONV3_HALF")) half = atoi(f) != 0;
    if (half) {
      if (og == 4)
        return launch_conv<9, 4, 1, 1, false, 1>(X, Wf, Bi, R, Y, batch, (int)channels, (int)height, (int)width, Ho, Wo, 1,
                                                 (int)out_channels, relu, st);
      return launch_conv<9, 2, 1, 1, false, 1>(X, Wf, Bi, R, Y, batch, (int)channels, (int)height, (int)width, Ho, Wo, 1,
                                               (int)out_channels, relu, st);
    }
  }
  if (ksize == 3) return og == 4 ? S2A_CONV(9, 4) : (og == 2 ? S2A_CONV(9, 2) : S2A_CONV(9, 1));
  if (og == 4) {
    // small maps (64^2 / 32^2 at batch 8): 128-position tiles give at most one workgroup per CU, and one workgroup's
    // chunk pipeline is latency-bound (32 MFMAs per wave between two memory round trips) -- 64-position tiles fill
    // the chip (2048 -> 512 and 2048 -> 256 on 32^2: 35 -> 26 us, 34 -> 23 us; no gain once every CU has a workgroup).  (ConvCfg<1, 4, 1, 2>: SD = 2 selects the 64-position tile; the spatial
    // stride of a 1x1 is the cstride argument.)  S2A_CONV1_HALF=0|1
    const int64_t wgs128 = ((int64_t)batch * Ho * Wo + 127) / 128 * (out_channels / 256);
    bool half = wgs128 < 256;   // measured: a win only while the 128-position grid leaves CUs empty
    if (const char* f = getenv("S2A_CONV1_HALF")) half = atoi(f) != 0;
    if (half)
      return launch_conv<1, 4, 1, 2>(X, Wf, Bi, R, Y, batch, (int)channels, (int)height, (int)width, Ho, Wo, stride,
                                     (int)out_channels, relu, st);
  }
#ifdef S2A_MEASURE
  if (const char* f = getenv("S2A_CONV1_OG")) {     // measurement builds: narrower out-channel groups for the 1x1 layers
    const int cap = atoi(f);
    if (cap == 2 && og == 4) return S2A_CONV(1, 2);
    if (cap == 1) return S2A_CONV(1, 1);
  }
#endif
  return og == 4 ? S2A_CONV(1, 4) : (og == 2 ? S2A_CONV(1, 2) : S2A_CONV(1, 1));
#undef S2A_CONV
}

extern "C" int s2a_conv3x3_tail1x1_f16(const void* x, const void* weight_frag, const void* bias,
                                       const void* tail_weight_frag, const void* tail_bias, const void* residual,
                                       void* out, const void* chain_weight_frag, const void* chain_bias, void* chain_out,
                                       int64_t chain_channels, int64_t batch, int64_t channels, int64_t mid_channels,
                                       int64_t out_channels, int64_t height, int64_t width, s2a_stream_t stream) {
  S2A_CHECK_ARG(batch >= 0 && height > 0 && width > 0, "conv3x3_tail1x1: bad shape");
  S2A_CHECK_ARG(channels == 64 && mid_channels == 64 && out_channels == 256,
                "conv3x3_tail1x1: built for the 64 -> 64 -> 256 bottleneck tail");
  S2A_CHECK_ARG((uint64_t)batch * height * width * out_channels * 2 < (1ull << 31) && height < 32000 && width < 32000,
                "conv3x3_tail1x1: tensor too large for 32-bit offsets");
  if (batch == 0) return S2A_OK;
  S2A_CHECK_ARG(x && weight_frag && bias && tail_weight_frag && tail_bias && out, "conv3x3_tail1x1: NULL tensor");
  S2A_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)out % 16) == 0 && ((uintptr_t)weight_frag % 16) == 0 &&
                ((uintptr_t)tail_weight_frag % 16) == 0 && ((uintptr_t)bias % 8) == 0 && ((uintptr_t)tail_bias % 8) == 0 &&
                ((uintptr_t)residual % 16) == 0, "conv3x3_tail1x1: tensors must be 16-byte aligned");
  ConvExtra ex{};
  ex.store_main = 0;
  ex.tail_w = (const _Float16*)tail_weight_frag;
  ex.tail_b = (const _Float16*)tail_bias;
  ex.tail_res = (const _Float16*)residual;
  ex.tail_out = (_Float16*)out;
  if (chain_weight_frag || chain_bias || chain_out) {
    S2A_CHECK_ARG(chain_weight_frag && chain_bias && chain_out, "conv3x3_tail1x1: chain filter, bias and output go together");
    S2A_CHECK_ARG(chain_channels == 64 || chain_channels == 128, "conv3x3_tail1x1: the chained 1x1 has 64 or 128 maps");
    S2A_CHECK_ARG(((uintptr_t)chain_weight_frag % 16) == 0 && ((uintptr_t)chain_bias % 8) == 0 && ((uintptr_t)chain_out % 16) == 0,
                  "conv3x3_tail1x1: chain tensors must be 16-byte aligned");
    ex.chain_w = (const _Float16*)chain_weight_frag;
    ex.chain_b = (const _Float16*)chain_bias;
    ex.chain_out = (_Float16*)chain_out;
    ex.chain_O = (int)chain_channels;
  }
  int ph = batch * ((width + 15) / 16) * ((height + 15) / 16) >= 256 ? 2 : 1;
  if (const char* f = getenv("S2A_CONV_PH_NARROW")) ph = atoi(f) == 2 ? 2 : 1;
  if (ph == 2)
    return launch_conv<9, 1, 2, 1, true>((const _Float16*)x, (const _Float16*)weight_frag, (const _Float16*)bias, nullptr,
                                         (_Float16*)out, batch, 64, (int)height, (int)width, (int)height, (int)width, 1,
                                         64, 1, as_stream(stream), nullptr, 0, 0, ex);
  return launch_conv<9, 1, 1, 1, true>((const _Float16*)x, (const _Float16*)weight_frag, (const _Float16*)bias, nullptr,
                                       (_Float16*)out, batch, 64, (int)height, (int)width, (int)height, (int)width, 1, 64,
                                       1, as_stream(stream), nullptr, 0, 0, ex);
}

extern "C" int s2a_conv1x1_add_up2_f16(const void* x, const void* weight_frag, const void* bias, const void* coarse,
                                       void* out, int64_t batch, int64_t channels, int64_t height, int64_t width,
                                       int64_t out_channels, s2a_stream_t stream) {
  S2A_CHECK_ARG(batch >= 0 && channels > 0 && out_channels > 0 && height > 0 && width > 0, "conv_add_up2: bad shape");
  S2A_CHECK_ARG(height % 2 == 0 && width % 2 == 0, "conv_add_up2: the map must be exactly twice the coarse map");
  S2A_CHECK_ARG(channels % 64 == 0 && out_channels % 64 == 0, "conv_add_up2: channel counts must be multiples of 64");
  const uint64_t x_bytes = (uint64_t)batch * height * width * channels * 2;
  S2A_CHECK_ARG(x_bytes < (1ull << 31) && (uint64_t)batch * height * width * out_channels * 2 < (1ull << 31),
                "conv_add_up2: tensor too large for 32-bit offsets");
  if (batch == 0) return S2A_OK;
  S2A_CHECK_ARG(x && weight_frag && coarse && out, "conv_add_up2: NULL tensor");
  S2A_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)out % 16) == 0 && ((uintptr_t)weight_frag % 16) == 0 &&
                ((uintptr_t)bias % 2) == 0 && ((uintptr_t)coarse % 16) == 0, "conv_add_up2: tensors must be 16-byte aligned");
  hipStream_t st = as_stream(stream);
  const int og = out_channels % 256 == 0 ? 4 : (out_channels % 128 == 0 ? 2 : 1);
  const _Float16 *X = (const _Float16*)x, *Wf = (const _Float16*)weight_frag, *Bi = (const _Float16*)bias,
                 *R = (const _Float16*)coarse;
  _Float16* Y = (_Float16*)out;
#define S2A_CONVU(OG_) launch_conv<1, OG_>(X, Wf, Bi, R, Y, batch, (int)channels, (int)height, (int)width, (int)height, (int)width, 1, (int)out_channels, 0, st, nullptr, 0, 1)
  return og == 4 ? S2A_CONVU(4) : (og == 2 ? S2A_CONVU(2) : S2A_CONVU(1));
#undef S2A_CONVU
}

extern "C" int s2a_conv_pack_weight_f16(const void* weight, int64_t out_channels, int64_t channels, int ksize,
                                        void* packed, s2a_stream_t stream) {
  S2A_CHECK_ARG(ksize == 3 || ksize == 1, "conv_pack_weight: kernel size must be 1 or 3");
  S2A_CHECK_ARG(out_channels % 64 == 0 && channels % 64 == 0, "conv_pack_weight: channel counts must be multiples of 64");
  S2A_CHECK_ARG(weight && packed, "conv_pack_weight: NULL tensor");
  const int taps = ksize * ksize;
  const int64_t wtot = out_channels * channels * taps;
  k_pack_weight_frag<<<(unsigned)((wtot + 255) / 256), 256, 0, as_stream(stream)>>>(
      (const _Float16*)weight, (int)out_channels, (int)channels, (_Float16*)packed, taps);
  S2A_LAUNCH_CHECK();
  return S2A_OK;
}

// ------------------------------------------------------------------ pyramid-packed launches
namespace s2a {
namespace {
// tiles (tile_rows x 16 positions) per level, pixel offsets; returns the total tile count or -1
int64_t build_levels(const s2a_pyramid* pyr, int64_t batch, LevelTab* lt, int64_t* total_pix, int tile_rows = 8) {
  if (!pyr || pyr->n_levels < 1 || pyr->n_levels > kMaxLevels) return -1;
  *lt = LevelTab{};
  lt->n = pyr->n_levels;
  lt->batch = (int)batch;
  int64_t tiles = 0, pix = 0;
  for (int i = 0; i < pyr->n_levels; i++) {
    const int64_t H = pyr->height[i], W = pyr->width[i];
    if (H < 1 || W < 1 || H >= 32000 || W >= 32000) return -1;
    lt->H[i] = (int)H; lt->W[i] = (int)W; lt->stride[i] = pyr->stride[i];
    lt->tile0[i] = (int)tiles; lt->pix0[i] = (int)pix;
    tiles += batch * ((W + 15) / 16) * ((H + tile_rows - 1) / tile_rows);
    pix += batch * H * W;
    if (tiles >= (1ll << 31) || pix >= (1ll << 31)) return -1;
  }
  *total_pix = pix;
  return tiles;
}
}  // namespace
}  // namespace s2a

extern "C" int64_t s2a_pyramid_pixels(const s2a_pyramid* pyr, int64_t batch) {
  LevelTab lt; int64_t pix = 0;
  return build_levels(pyr, batch, &lt, &pix) < 0 ? -1 : pix;
}

static int conv3x3_pyramid_impl(const void* x, const void* weight_frag, const void* bias, const void* residual,
                                void* out, ConvExtra ex, int64_t batch, int64_t channels, int64_t out_channels,
                                int relu, const s2a_pyramid* pyr, s2a_stream_t stream);

extern "C" int s2a_conv3x3_pyramid_f16(const void* x, const void* weight_frag, const void* bias, const void* residual,
                                       void* out, int64_t batch, int64_t channels, int64_t out_channels,
                                       int relu, const s2a_pyramid* pyr, s2a_stream_t stream) {
  return conv3x3_pyramid_impl(x, weight_frag, bias, residual, out, ConvExtra{nullptr, nullptr, nullptr, nullptr, 1}, batch,
                              channels, out_channels, relu, pyr, stream);
}

extern "C" int s2a_conv3x3_head_pyramid_f16(const void* x, const void* weight_frag, const void* bias, void* out,
                                            const void* head_weight_frag, const void* head_bias, void* head_out,
                                            int64_t batch, int64_t channels, int64_t out_channels, int relu,
                                            const s2a_pyramid* pyr, s2a_stream_t stream) {
  S2A_CHECK_ARG(out_channels == 256, "conv3x3_head_pyramid: the fused 1x1 head needs a 256-channel tower");
  S2A_CHECK_ARG(head_weight_frag && head_bias && head_out, "conv3x3_head_pyramid: NULL head tensor");
  S2A_CHECK_ARG(((uintptr_t)head_weight_frag % 16) == 0 && ((uintptr_t)head_bias % 8) == 0 && ((uintptr_t)head_out % 16) == 0,
                "conv3x3_head_pyramid: misaligned head tensor");
  ConvExtra ex{nullptr, (const _Float16*)head_weight_frag, (const _Float16*)head_bias, (_Float16*)head_out, out != nullptr};
  return conv3x3_pyramid_impl(x, weight_frag, bias, nullptr, out ? out : head_out, ex, batch, channels, out_channels, relu,
                              pyr, stream);
}

extern "C" int s2a_orconv_pool_pyramid_f16(const void* x, const void* weight_frag, const void* bias, void* out,
                                           void* pooled, int64_t batch, int64_t channels, int64_t out_channels,
                                           const s2a_pyramid* pyr, s2a_stream_t stream) {
  S2A_CHECK_ARG(pooled != nullptr && ((uintptr_t)pooled % 16) == 0 && out_channels % 64 == 0,
                "orconv_pool_pyramid: pooled must be a 16-byte aligned buffer, out_channels a multiple of 64");
  return conv3x3_pyramid_impl(x, weight_frag, bias, nullptr, out, ConvExtra{(_Float16*)pooled, nullptr, nullptr, nullptr, 1},
                              batch, channels, out_channels, 0, pyr, stream);
}

static int conv3x3_pyramid_impl(const void* x, const void* weight_frag, const void* bias, const void* residual,
                                void* out, ConvExtra ex, int64_t batch, int64_t channels, int64_t out_channels,
                                int relu, const s2a_pyramid* pyr, s2a_stream_t stream) {
  S2A_CHECK_ARG(batch >= 0 && channels > 0 && out_channels > 0, "conv_pyramid: bad shape");
  S2A_CHECK_ARG((channels % 64 == 0 || channels == 32) && out_channels % 64 == 0,
                "conv_pyramid: channels must be 32 or a multiple of 64, out_channels a multiple of 64");
  LevelTab lt; int64_t pix = 0;
  // Full-width towers: 16 x 16 tiles on 512-thread workgroups with the filter staged once per workgroup through
  // LDS (ConvCfg::kWLds) -- 4-7 % faster than two 8 x 16 workgroups per CU, bit-identical.  S2A_CONV_PH=1|2: A/B switch
  int ph = (out_channels % 256 == 0 && channels % 64 == 0) ? 2 : 1;
  if (const char* f = getenv("S2A_CONV_PH")) ph = atoi(f) == 2 && out_channels % 256 == 0 && channels % 64 == 0 ? 2 : 1;
  const int64_t tiles = build_levels(pyr, batch, &lt, &pix, 8 * ph);
  S2A_CHECK_ARG(tiles >= 0, "conv_pyramid: bad level table (1..8 levels, positive sizes)");
  S2A_CHECK_ARG((uint64_t)pix * channels * 2 < (1ull << 31), "conv_pyramid: input too large for 32-bit offsets");
  if (batch == 0) return S2A_OK;
  S2A_CHECK_ARG(x && weight_frag && out, "conv_pyramid: NULL tensor");
  S2A_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)out % 16) == 0 && ((uintptr_t)weight_frag % 16) == 0 &&
                ((uintptr_t)bias % 8) == 0 && ((uintptr_t)residual % 16) == 0, "conv_pyramid: tensors must be 16-byte aligned");
  if (lt.n == 1) lt.n = 2, lt.tile0[1] = 0x7fffffff;   // keep the rebind path (n > 1) for a one-level table
  hipStream_t st = as_stream(stream);
  int og = out_channels % 256 == 0 ? 4 : (out_channels % 128 == 0 ? 2 : 1);
  if (const char* f = getenv("S2A_CONV_OG")) og = std::min(og, std::max(1, atoi(f)));   // A/B switch for measurements
  const _Float16 *X = (const _Float16*)x, *Wf = (const _Float16*)weight_frag, *Bi = (const _Float16*)bias,
                 *R = (const _Float16*)residual;
  _Float16* Y = (_Float16*)out;
  const ConvExtra Pq = ex;
#define S2A_CONVP(OG_) launch_conv<9, OG_>(X, Wf, Bi, R, Y, batch, (int)channels, lt.H[0], lt.W[0], lt.H[0], lt.W[0], 1, (int)out_channels, relu, st, &lt, tiles, 0, Pq)
  if (ph == 2 && og == 4)
    return launch_conv<9, 4, 2>(X, Wf, Bi, R, Y, batch, (int)channels, lt.H[0], lt.W[0], lt.H[0], lt.W[0], 1,
                                (int)out_channels, relu, st, &lt, tiles, 0, Pq);
  return og == 4 ? S2A_CONVP(4) : (og == 2 ? S2A_CONVP(2) : S2A_CONVP(1));
#undef S2A_CONVP
}

extern "C" int s2a_align_conv_pyramid_f16(const void* x, const float* anchors, const void* weight_packed, void* out,
                                          int64_t batch, int64_t channels, int64_t out_channels, int relu,
                                          const s2a_pyramid* pyr, s2a_stream_t stream) {
  S2A_CHECK_ARG(batch >= 0 && channels > 0 && out_channels > 0, "align_conv_pyramid: bad shape");
  S2A_CHECK_ARG(channels % 64 == 0 && out_channels % 64 == 0, "align_conv_pyramid: channels and out_channels must be multiples of 64");
  LevelTab lt; int64_t pix = 0;
  const int64_t tiles = build_levels(pyr, batch, &lt, &pix);
  S2A_CHECK_ARG(tiles >= 0, "align_conv_pyramid: bad level table (1..8 levels, positive sizes)");
  for (int i = 0; i < lt.n; i++) {
    S2A_CHECK_ARG(lt.stride[i] > 0, "align_conv: stride must be positive");
    S2A_CHECK_ARG(lt.H[i] >= 3 && lt.W[i] >= 3, "input image is smaller than kernel");
  }
  S2A_CHECK_ARG((uint64_t)pix * channels * 2 < (1ull << 31), "align_conv_pyramid: input too large for 32-bit offsets");
  if (batch == 0) return S2A_OK;
  S2A_CHECK_ARG(x && anchors && weight_packed && out, "align_conv_pyramid: NULL tensor");
  S2A_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)out % 16) == 0 && ((uintptr_t)weight_packed % 16) == 0,
                "align_conv_pyramid: tensors must be 16-byte aligned");
  if (lt.n == 1) lt.n = 2, lt.tile0[1] = 0x7fffffff;
  hipStream_t st = as_stream(stream);
  // s2a_dcn_pack_weight (f16) = stage-major layout followed by the MFMA-fragment layout
  const _Float16* wfrag = (const _Float16*)weight_packed + (size_t)out_channels * channels * 9;
  // (Forms of this launch that were built, tested bit-identical, measured on MI355X and removed again -- DESIGN.md 4 has the
  // numbers: a persistent workgroup per CU with the next tile's anchors / patch prefetched (245 vs 232 us: spills), a
  // three-slot column ring with loaders two stages ahead (2-8 % slower), two half-tile workgroups per CU (332 vs 282 us:
  // the filter streamed twice), half tiles for the last round in a second launch (-1.2 %).)
  const unsigned ogroups = (unsigned)((out_channels + kMaxO - 1) / kMaxO);
  const int relu_flags = (relu ? 1 : 0) | (half_coords_requested() ? 2 : 0);
  auto kern = k_dcn_patch<true, 1>;
  S2A_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kPatchLds));
  // the last round as half tiles when it would leave more than half of the CUs idle (see k_dcn_patch); S2A_DCN_HALF_TAIL=0|1 forces
  int ncu = 0;
  {
    int rc_ = device_cu_count(&ncu);
    if (rc_ != S2A_OK) return rc_;
  }
  const int64_t rem = tiles % ncu;
  bool half_tail = tiles > ncu && rem != 0 && 2 * rem <= ncu;
  if (const char* f = getenv("S2A_DCN_HALF_TAIL")) half_tail = atoi(f) != 0 && tiles > rem && rem != 0;
  const unsigned n_full = half_tail ? (unsigned)(tiles - rem) : 0u;
  const unsigned grid_x = half_tail ? (unsigned)(tiles - rem + 2 * rem) : (unsigned)tiles;
  kern<<<dim3(grid_x, ogroups), 512, kPatchLds, st>>>(PatchArgs{(const _Float16*)x, anchors, wfrag, (_Float16*)out, 0, (int)channels,
                                                      lt.H[0], lt.W[0], (int)out_channels, lt.stride[0], relu_flags, 0u, 0, n_full, lt});
  S2A_LAUNCH_CHECK();
  return S2A_OK;
}
