// polyiou on the GPU — DOTA_devkit/polyiou/csrc/polyiou.cpp:108-128 (iou_poly) and its helpers
// (:8-103), double precision, operation for operation (COMPILE WITH -ffp-contract=off: results are
// compared bit for bit with the oracle, which is pinned bit-exact to the reference's SWIG module).
//
// The algorithm sums, over the 4x4 edge pairs of the two quadrilaterals, the signed area of
// clip(triangle(o,a,b) by the three half-planes of triangle(o,c,d)) — a Sutherland-Hodgman style
// half-plane clip (polygon_cut :58-71, eps = 1e-8 sign test :8-12).  The small work polygons
// (<= 10 and <= 16 points) live in LDS as [point][thread] double2, like the rotated-IoU kernel:
// no scratch memory, no bank conflicts whatever the (divergent) point index.
// Used by the chip-merge path (ResultMerge_multi_process.py:62-123) and as config 1's companion
// of box_iou_rotated.
#include <hip/hip_runtime.h>

#include "common.hpp"

namespace s2a {
namespace {

constexpr int kPolyThreads = 128;
constexpr int kPMax = 10, kTmpMax = 16;

struct D2 {
  double x, y;
};
__device__ __forceinline__ int sgn(double d) { return (d > 1e-8) - (d < -1e-8); }
__device__ __forceinline__ double tri_cross(D2 o, D2 a, D2 b) {
  return (a.x - o.x) * (b.y - o.y) - (b.x - o.x) * (a.y - o.y);
}
__device__ __forceinline__ bool same_pt(D2 a, D2 b) { return sgn(a.x - b.x) == 0 && sgn(a.y - b.y) == 0; }

#define P(i) p[(i) * kPolyThreads]
#define T(i) tmp[(i) * kPolyThreads]

// area() :23-30 on the LDS polygon (writes p[n] = p[0] like the reference)
__device__ __forceinline__ double shoelace_lds(D2* p, int n) {
  P(n) = P(0);
  double r = 0;
  for (int i = 0; i < n; i++) {
    D2 a = P(i), b = P(i + 1);
    r += a.x * b.y - a.y * b.x;
  }
  return r / 2.0;
}

// polygon_cut :58-71
__device__ __forceinline__ void half_plane_cut(D2* p, int& n, D2 a, D2 b, D2* tmp) {
  int m = 0;
  P(n) = P(0);
  for (int i = 0; i < n; i++) {
    D2 pi = P(i), pn = P(i + 1);
    int si = sgn(tri_cross(a, b, pi));
    if (si > 0) {
      T(m) = pi;
      m++;
    }
    if (si != sgn(tri_cross(a, b, pn))) {
      // lineCross :31-41 (its return value is ignored by polygon_cut; the point is appended regardless)
      double s1 = tri_cross(a, b, pi), s2 = tri_cross(a, b, pn);
      D2 hit = T(m);   // stale slot content when lineCross bails out before writing (as the reference)
      if (!(sgn(s1) == 0 && sgn(s2) == 0) && sgn(s2 - s1) != 0) {
        hit.x = (pi.x * s2 - pn.x * s1) / (s2 - s1);
        hit.y = (pi.y * s2 - pn.y * s1) / (s2 - s1);
      }
      T(m) = hit;
      m++;
    }
  }
  n = 0;
  for (int i = 0; i < m; i++) {
    D2 ti = T(i);
    if (!i || !same_pt(ti, T(i - 1))) {
      P(n) = ti;
      n++;
    }
  }
  while (n > 1 && same_pt(P(n - 1), P(0))) n--;
}

// intersectArea(a,b,c,d) :74-90
__device__ __forceinline__ double fan_overlap(D2 a, D2 b, D2 c, D2 d, D2* p, D2* tmp) {
  D2 o{0, 0};
  int s1 = sgn(tri_cross(o, a, b)), s2 = sgn(tri_cross(o, c, d));
  if (s1 == 0 || s2 == 0) return 0.0;
  if (s1 == -1) {
    D2 t = a;
    a = b;
    b = t;
  }
  if (s2 == -1) {
    D2 t = c;
    c = d;
    d = t;
  }
  P(0) = o;
  P(1) = a;
  P(2) = b;
  int n = 3;
  half_plane_cut(p, n, o, c, tmp);
  half_plane_cut(p, n, c, d, tmp);
  half_plane_cut(p, n, d, o, tmp);
  double r = fabs(shoelace_lds(p, n));
  if (s1 * s2 == -1) r = -r;
  return r;
}

__device__ __forceinline__ double quad_area(const D2 (&q)[4]) {
  double r = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) r += q[i].x * q[(i + 1) & 3].y - q[i].y * q[(i + 1) & 3].x;
  return r / 2.0;
}

// iou_poly :108-128 + intersectArea(ps1,n1,ps2,n2) :92-103
__device__ double poly_iou(const double* pa, const double* pb, D2* p, D2* tmp) {
  D2 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    a[i] = {pa[2 * i], pa[2 * i + 1]};
    b[i] = {pb[2 * i], pb[2 * i + 1]};
  }
  if (quad_area(a) < 0) {  // std::reverse of 4 points
    D2 t = a[0]; a[0] = a[3]; a[3] = t;
    t = a[1]; a[1] = a[2]; a[2] = t;
  }
  if (quad_area(b) < 0) {
    D2 t = b[0]; b[0] = b[3]; b[3] = t;
    t = b[1]; b[1] = b[2]; b[2] = t;
  }
  double inter = 0;
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) inter += fan_overlap(a[i], a[(i + 1) & 3], b[j], b[(j + 1) & 3], p, tmp);
  double uni = fabs(quad_area(a)) + fabs(quad_area(b)) - inter;
  return inter / uni;
}

__global__ __launch_bounds__(kPolyThreads) void k_polyiou_pairs(const double* __restrict__ p8,
                                                                const double* __restrict__ q8, int64_t n,
                                                                double* __restrict__ out) {
  __shared__ D2 s_p[kPMax * kPolyThreads];
  __shared__ D2 s_t[kTmpMax * kPolyThreads];
  int64_t i = (int64_t)blockIdx.x * kPolyThreads + threadIdx.x;
  if (i >= n) return;
  out[i] = poly_iou(p8 + 8 * i, q8 + 8 * i, s_p + threadIdx.x, s_t + threadIdx.x);
}

#undef P
#undef T

}  // namespace
}  // namespace s2a

using namespace s2a;

extern "C" int s2a_polyiou_pairs(const double* polys1, const double* polys2, int64_t n, double* ious,
                                 s2a_stream_t stream) {
  S2A_CHECK_ARG(n >= 0, "polyiou_pairs: negative size");
  if (n == 0) return S2A_OK;
  S2A_CHECK_ARG(polys1 && polys2 && ious, "polyiou_pairs: NULL tensor");
  k_polyiou_pairs<<<(unsigned)((n + kPolyThreads - 1) / kPolyThreads), kPolyThreads, 0, as_stream(stream)>>>(polys1, polys2, n, ious);
  S2A_LAUNCH_CHECK();
  return S2A_OK;
}
