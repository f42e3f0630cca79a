// ORN ops for gfx950: active rotating filter expansion and rotation-invariant pooling.
//   ARF forward  : models/orn/src/cuda/ActiveRotatingFilter_cuda.cu:20-46, :79-119
//   RI pooling   : models/orn/functions/rotation_invariant_pooling.py:19-27
// Both are pure data movement (zero flops) => HBM-bound.  The reference ARF kernel
// scatters (one thread per weight element, nRot strided stores); here every thread owns
// one OUTPUT element and reads through the inverted index table held in LDS, so both the
// loads (nEntry-contiguous) and the stores (fully contiguous) coalesce.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include "common.hpp"

namespace s2a {
namespace {

template <typename T>
__global__ __launch_bounds__(256) void k_arf_forward(const T* __restrict__ w,
                                                     const uint8_t* __restrict__ idx, int64_t n_out,
                                                     int64_t n_in, int n_entry, int n_rot,
                                                     T* __restrict__ out) {
  // inv[k][t] = l such that idx[l][k] - 1 == t   (idx[:,k] is a permutation of 1..nEntry)
  __shared__ uint8_t inv[8 * 256];
  for (int e = threadIdx.x; e < n_entry * n_rot; e += blockDim.x) {
    int l = e / n_rot, k = e % n_rot;
    inv[k * n_entry + (int)idx[e] - 1] = (uint8_t)l;
  }
  __syncthreads();
  const int64_t total = n_out * n_rot * n_in * n_entry;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    int t = (int)(e % n_entry);
    int64_t r = e / n_entry;
    int64_t j = r % n_in;
    r /= n_in;
    int k = (int)(r % n_rot);
    int64_t i = r / n_rot;
    out[e] = w[(i * n_in + j) * n_entry + inv[k * n_entry + t]];
  }
}

// ARF backward (ActiveRotatingFilter_cuda.cu:49-76): gather-sum of the nRot rotated copies, one
// thread per filter-bank element, k ascending like the reference (plain adds: bit-identical)
template <typename T>
__global__ __launch_bounds__(256) void k_arf_backward(const T* __restrict__ gout,
                                                      const uint8_t* __restrict__ idx, int64_t n_out,
                                                      int64_t n_in, int n_entry, int n_rot,
                                                      T* __restrict__ gin) {
  __shared__ uint8_t s_idx[8 * 256];
  for (int e = threadIdx.x; e < n_entry * n_rot; e += blockDim.x) s_idx[e] = idx[e];
  __syncthreads();
  const int64_t total = n_out * n_in * n_entry;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    int l = (int)(e % n_entry);
    int64_t r = e / n_entry;
    int64_t j = r % n_in, i = r / n_in;
    T v = 0;
    for (int k = 0; k < n_rot; k++) {
      int t = (int)s_idx[l * n_rot + k] - 1;
      v = v + gout[((i * n_rot + k) * n_in + j) * n_entry + t];
    }
    gin[e] = v;
  }
}

__device__ __forceinline__ float nanmax(float m, float a) {
  // torch.max propagates NaN
  return (a > m || a != a) ? a : m;
}

template <typename T>
__device__ __forceinline__ float to_f(T v) { return (float)v; }

// NCHW: x[B,C,HW] -> out[B,C/nOri,HW]
template <typename T>
__global__ __launch_bounds__(256) void k_ripool_nchw(const T* __restrict__ x, int64_t B, int64_t G,
                                                     int64_t HW, int n_ori, T* __restrict__ out) {
  const int64_t total = B * G * HW;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    int64_t p = e % HW;
    int64_t bg = e / HW;  // = b*G + g ; channel block starts at bg*nOri
    const T* src = x + bg * n_ori * HW + p;
    T best = src[0];
    float m = to_f(best);
    for (int o = 1; o < n_ori; o++) {
      T v = src[(int64_t)o * HW];
      float f = to_f(v);
      if (f > m || f != f) {
        m = f;
        best = v;
      }
    }
    out[e] = best;
  }
}

// NHWC: x[B,HW,C] -> out[B,HW,C/nOri]; the nOri channels of a group are contiguous
template <typename T>
__global__ __launch_bounds__(256) void k_ripool_nhwc(const T* __restrict__ x, int64_t total_groups,
                                                     int n_ori, T* __restrict__ out) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total_groups;
       e += (int64_t)gridDim.x * blockDim.x) {
    const T* src = x + e * n_ori;
    T best = src[0];
    float m = to_f(best);
    for (int o = 1; o < n_ori; o++) {
      T v = src[o];
      float f = to_f(v);
      if (f > m || f != f) {
        m = f;
        best = v;
      }
    }
    out[e] = best;
  }
}

inline unsigned grid_cap(int64_t total, int threads = 256, unsigned cap = 256 * 8) {
  int64_t g = (total + threads - 1) / threads;
  return (unsigned)std::max<int64_t>(1, std::min<int64_t>(g, cap));
}

}  // namespace
}  // namespace s2a

using namespace s2a;

extern "C" int s2a_arf_forward(const void* weight, const uint8_t* indices, int64_t n_out,
                               int64_t n_in, int n_orientation, int kh, int kw, int n_rotation,
                               int dtype, void* output, s2a_stream_t stream) {
  S2A_CHECK_ARG(n_out >= 0 && n_in >= 0 && n_orientation > 0 && kh > 0 && kw > 0 && n_rotation > 0,
                "arf_forward: bad shape");
  const int n_entry = n_orientation * kh * kw;
  S2A_CHECK_ARG(n_entry <= 255, "arf_forward: nOrientation*kH*kW must be <= 255 (uint8 1-based index)");
  S2A_CHECK_ARG(n_rotation <= 8, "arf_forward: nRotation must be <= 8");
  S2A_CHECK_ARG(dtype == S2A_DTYPE_F32 || dtype == S2A_DTYPE_F16 || dtype == S2A_DTYPE_F64, "arf_forward: dtype");
  const int64_t total = n_out * n_rotation * n_in * n_entry;
  if (total == 0) return S2A_OK;  // reference returns the empty tensor (cuda.cu:100-103)
  S2A_CHECK_ARG(weight && indices && output, "arf_forward: NULL tensor");
  hipStream_t st = as_stream(stream);
  unsigned g = grid_cap(total);
  if (dtype == S2A_DTYPE_F32)
    k_arf_forward<float><<<g, 256, 0, st>>>((const float*)weight, indices, n_out, n_in, n_entry,
                                            n_rotation, (float*)output);
  else if (dtype == S2A_DTYPE_F64)
    k_arf_forward<double><<<g, 256, 0, st>>>((const double*)weight, indices, n_out, n_in, n_entry,
                                             n_rotation, (double*)output);
  else
    k_arf_forward<uint16_t><<<g, 256, 0, st>>>((const uint16_t*)weight, indices, n_out, n_in,
                                               n_entry, n_rotation, (uint16_t*)output);
  S2A_LAUNCH_CHECK();
  return S2A_OK;
}

extern "C" int s2a_arf_backward(const uint8_t* indices, const void* grad_output, int64_t n_out, int64_t n_in,
                                int n_orientation, int kh, int kw, int n_rotation, int dtype,
                                void* grad_input, s2a_stream_t stream) {
  S2A_CHECK_ARG(n_out >= 0 && n_in >= 0 && n_orientation > 0 && kh > 0 && kw > 0 && n_rotation > 0,
                "arf_backward: bad shape");
  const int n_entry = n_orientation * kh * kw;
  S2A_CHECK_ARG(n_entry <= 255 && n_rotation <= 8, "arf_backward: index table too large");
  S2A_CHECK_ARG(dtype == S2A_DTYPE_F32 || dtype == S2A_DTYPE_F64, "arf_backward: float32 / float64 (the reference's dispatch, cuda.cu:149)");
  const int64_t total = n_out * n_in * n_entry;
  if (total == 0) return S2A_OK;
  S2A_CHECK_ARG(indices && grad_output && grad_input, "arf_backward: NULL tensor");
  if (dtype == S2A_DTYPE_F64)
    k_arf_backward<double><<<grid_cap(total), 256, 0, as_stream(stream)>>>((const double*)grad_output, indices, n_out, n_in,
                                                                          n_entry, n_rotation, (double*)grad_input);
  else
    k_arf_backward<float><<<grid_cap(total), 256, 0, as_stream(stream)>>>((const float*)grad_output, indices, n_out, n_in,
                                                                         n_entry, n_rotation, (float*)grad_input);
  S2A_LAUNCH_CHECK();
  return S2A_OK;
}

// ---------------------------------------------------------------- rotation-invariant encoding
// orn_cuda.rie_forward / rie_backward (models/orn/src/cuda/RotationInvariantEncoding_cuda.cu:20-57,60-84; CPU twin
// cpu/RotationInvariantEncoding_cpu.cpp:6-45,47-76): per (batch, feature) group of nOri values the main direction
// = FIRST index of the strict maximum (start value -FLT_MAX), values rotated so that it comes first;
// backward rotates the gradient back.  One thread per group.
namespace s2a {
namespace {
__global__ void k_rie_forward(const float* __restrict__ f, int64_t groups, int n_ori, uint8_t* __restrict__ dir,
                              float* __restrict__ aligned) {
  for (int64_t gidx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; gidx < groups; gidx += (int64_t)gridDim.x * blockDim.x) {
    const float* src = f + gidx * n_ori;
    float mx = -3.402823466e+38F;
    int d = 0;              // (the reference leaves the direction uninitialised when nothing exceeds -FLT_MAX)
    for (int l = 0; l < n_ori; l++) {
      const float v = src[l];
      if (v > mx) { mx = v; d = l; }
    }
    dir[gidx] = (uint8_t)d;
    for (int l = 0; l < n_ori; l++) aligned[gidx * n_ori + (l - d + n_ori) % n_ori] = src[l];
  }
}

__global__ void k_rie_backward(const uint8_t* __restrict__ dir, const float* __restrict__ g, int64_t groups, int n_ori,
                               float* __restrict__ gin) {
  for (int64_t gidx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; gidx < groups; gidx += (int64_t)gridDim.x * blockDim.x) {
    const int d = dir[gidx];
    for (int l = 0; l < n_ori; l++) gin[gidx * n_ori + (l + d) % n_ori] = g[gidx * n_ori + l];
  }
}
}  // namespace
}  // namespace s2a

extern "C" int s2a_rie_forward(const void* feature, int64_t batch, int64_t channels, int n_orientation, int dtype,
                               uint8_t* main_direction, void* aligned, s2a_stream_t stream) {
  S2A_CHECK_ARG(batch >= 0 && channels >= 0 && n_orientation > 0 && n_orientation <= 255, "rie_forward: bad shape");
  S2A_CHECK_ARG(channels % n_orientation == 0, "rie_forward: channels %% nOrientation != 0");
  S2A_CHECK_ARG(dtype == S2A_DTYPE_F32, "rie_forward: float32 only (the reference dispatches float / double)");
  const int64_t groups = batch * (channels / n_orientation);
  if (groups == 0) return S2A_OK;
  S2A_CHECK_ARG(feature && main_direction && aligned, "rie_forward: NULL tensor");
  k_rie_forward<<<grid_cap(groups), 256, 0, as_stream(stream)>>>((const float*)feature, groups, n_orientation,
                                                                 main_direction, (float*)aligned);
  S2A_LAUNCH_CHECK();
  return S2A_OK;
}

extern "C" int s2a_rie_backward(const uint8_t* main_direction, const void* grad_output, int64_t batch, int64_t features,
                                int n_orientation, int dtype, void* grad_input, s2a_stream_t stream) {
  S2A_CHECK_ARG(batch >= 0 && features >= 0 && n_orientation > 0 && n_orientation <= 255, "rie_backward: bad shape");
  S2A_CHECK_ARG(dtype == S2A_DTYPE_F32, "rie_backward: float32 only (the reference dispatches float / double)");
  const int64_t groups = batch * features;
  if (groups == 0) return S2A_OK;
  S2A_CHECK_ARG(main_direction && grad_output && grad_input, "rie_backward: NULL tensor");
  k_rie_backward<<<grid_cap(groups), 256, 0, as_stream(stream)>>>(main_direction, (const float*)grad_output, groups,
                                                                  n_orientation, (float*)grad_input);
  S2A_LAUNCH_CHECK();
  return S2A_OK;
}

extern "C" int s2a_rot_inv_pool(const void* x, int64_t batch, int64_t channels, int64_t hw,
                                int n_orientation, int dtype, int layout, void* out,
                                s2a_stream_t stream) {
  S2A_CHECK_ARG(batch >= 0 && channels >= 0 && hw >= 0 && n_orientation > 0, "rot_inv_pool: bad shape");
  S2A_CHECK_ARG(channels % n_orientation == 0, "rot_inv_pool: channels %% nOrientation != 0");
  S2A_CHECK_ARG(dtype == S2A_DTYPE_F32 || dtype == S2A_DTYPE_F16, "rot_inv_pool: dtype");
  const int64_t G = channels / n_orientation;
  const int64_t total = batch * G * hw;
  if (total == 0) return S2A_OK;
  S2A_CHECK_ARG(x && out, "rot_inv_pool: NULL tensor");
  hipStream_t st = as_stream(stream);
  unsigned g = grid_cap(total);
  if (layout == S2A_LAYOUT_NCHW) {
    if (dtype == S2A_DTYPE_F32)
      k_ripool_nchw<float><<<g, 256, 0, st>>>((const float*)x, batch, G, hw, n_orientation, (float*)out);
    else
      k_ripool_nchw<_Float16><<<g, 256, 0, st>>>((const _Float16*)x, batch, G, hw, n_orientation, (_Float16*)out);
  } else {
    if (dtype == S2A_DTYPE_F32)
      k_ripool_nhwc<float><<<g, 256, 0, st>>>((const float*)x, total, n_orientation, (float*)out);
    else
      k_ripool_nhwc<_Float16><<<g, 256, 0, st>>>((const _Float16*)x, total, n_orientation, (_Float16*)out);
  }
  S2A_LAUNCH_CHECK();
  return S2A_OK;
}
