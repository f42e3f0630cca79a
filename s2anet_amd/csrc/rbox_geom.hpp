// Rotated-box IoU geometry for gfx950 — device code.
//
// Restates, operation for operation, the __CUDACC__ branch of the reference's
// box_iou_rotated_utils.h (utils/box_iou_rotated/src/box_iou_rotated_utils.h:56-375;
// the nms_rotated / ml_nms_rotated copies are arithmetically identical):
//   * vertices from DOUBLE cos/sin of the float angle (:62-64)
//   * 16 edge-edge tests with the |det| <= 1e-14 skip (:94-118), 4+4 vertex-in-rect
//     tests (:121-164), appended in the reference's order
//   * Graham scan with the O(n^2) swap sort of the GPU branch (:209-226), the
//     dist > 1e-8 de-dup (:243-253) and the cross >= 0 pop rule (:263-268)
//   * fan area (:285-296), iou = I / (A1 + A2 - I) (:354-362)
// Translation units including this header MUST be compiled with -ffp-contract=off:
// the NMS keep decision `iou > thr` has to be bit-identical to the reference's
// un-fused float evaluation (the oracle pins that against the reference's CPU build).
//
// MI355X mapping: the <=24 candidate points live in LDS, laid out [point][thread] as
// float2 so that a wave's 64 lanes always hit 64 distinct bank pairs whatever their
// (divergent) point index — no scratch memory, no bank conflicts.  cos/sin (double
// precision, ~200 instructions) are hoisted out of the pair loop into a per-box
// pre-pass (PreBox), which is exact because they depend on one box only.
#pragma once
#include <hip/hip_runtime.h>

namespace s2a {

struct alignas(16) PreBox {
  float x, y, w, h;  // centre, size (pixels)
  float c2, s2;      // (float)cos(a)*0.5f, (float)sin(a)*0.5f  (:63-64)
  float r;           // circumscribed-circle radius, cull only
  float label;       // ml-NMS label / segment (unused by the geometry)
};

__device__ __forceinline__ PreBox make_prebox(float x, float y, float w, float h, float a,
                                              float label) {
  PreBox b;
  b.x = x;
  b.y = y;
  b.w = w;
  b.h = h;
  double th = (double)a;
  b.c2 = (float)cos(th) * 0.5f;
  b.s2 = (float)sin(th) * 0.5f;
  b.r = 0.5f * sqrtf(w * w + h * h);
  b.label = label;
  return b;
}

// Exact shortcut: circumscribed circles more than 0.2 % (+1e-3 px) apart => the boxes
// share no point => the reference finds num == 0 and returns exactly 0.0f (:316-318).
// NaN coordinates compare false and fall through to the full evaluation.
__device__ __forceinline__ bool surely_disjoint(float ax, float ay, float ar, float bx, float by,
                                                float br) {
  float dx = ax - bx, dy = ay - by;
  float R = (ar + br) * 1.002f + 1e-3f;
  return dx * dx + dy * dy > R * R;
}

// Second-stage exact shortcut: separating-axis test on the two rectangles with the same 0.2 % +
// 1e-3 px safety margin (a gap that large cannot be bridged by the reference's float rounding, so
// it finds no intersection point and no contained vertex: num == 0 -> exactly 0.0f).  Removes the
// elongated / rotated near-misses the circle test lets through (about half of its survivors).
__device__ __forceinline__ bool sat_disjoint(const PreBox& A, const PreBox& B) {
  const float dx = B.x - A.x, dy = B.y - A.y;
  // half-extent vectors u (along w) and v (along h); unit axes are (2*c2, 2*s2), (-2*s2, 2*c2)
  const float aux = A.c2 * A.w, auy = A.s2 * A.w, avx = -A.s2 * A.h, avy = A.c2 * A.h;
  const float bux = B.c2 * B.w, buy = B.s2 * B.w, bvx = -B.s2 * B.h, bvy = B.c2 * B.h;
  const float acx = 2.f * A.c2, asx = 2.f * A.s2, bcx = 2.f * B.c2, bsx = 2.f * B.s2;
  const float ahw = 0.5f * fabsf(A.w), ahh = 0.5f * fabsf(A.h), bhw = 0.5f * fabsf(B.w), bhh = 0.5f * fabsf(B.h);
  float d, r;
  d = fabsf(dx * acx + dy * asx);   // A's w axis
  r = ahw + fabsf(bux * acx + buy * asx) + fabsf(bvx * acx + bvy * asx);
  if (d > r * 1.002f + 1e-3f) return true;
  d = fabsf(-dx * asx + dy * acx);  // A's h axis
  r = ahh + fabsf(-bux * asx + buy * acx) + fabsf(-bvx * asx + bvy * acx);
  if (d > r * 1.002f + 1e-3f) return true;
  d = fabsf(dx * bcx + dy * bsx);   // B's w axis
  r = bhw + fabsf(aux * bcx + auy * bsx) + fabsf(avx * bcx + avy * bsx);
  if (d > r * 1.002f + 1e-3f) return true;
  d = fabsf(-dx * bsx + dy * bcx);  // B's h axis
  r = bhh + fabsf(-aux * bsx + auy * bcx) + fabsf(-avx * bsx + avy * bcx);
  return d > r * 1.002f + 1e-3f;
}

// NMS pre-filter for a pair that passed the circle test: true when the pair can never set a suppression bit at
// threshold thr, so its (expensive) IoU need not be evaluated.  Two reasons:
//  (a) the separating-axis test above (exact zero), same expressions;
//  (b) an upper bound on the IoU: the intersection lies inside A, between the extreme projections of B on A's two
//      axes -- a rectangle of ov_w x ov_h in A's frame -- and likewise in B's frame, so
//      I <= Iub = min(ovA_w * ovA_h, ovB_w * ovB_h) and IoU = I / (a1 + a2 - I) <= Iub / (a1 + a2 - Iub).
//      Dropped only when that bound is below 0.99 thr, and only where the reference's own float evaluation is
//      trustworthy to well under that 1 % margin: both boxes with positive area and not thinner than 1:20 (its polygon
//      area loses ~64 eps D^2 absolute on needles), thr > 0.05.  Contains the area-ratio bound IoU <= min / max.
__device__ __forceinline__ bool nms_pair_skippable(const PreBox& A, const PreBox& B, float thr) {
  const float dx = B.x - A.x, dy = B.y - A.y;
  const float aux = A.c2 * A.w, auy = A.s2 * A.w, avx = -A.s2 * A.h, avy = A.c2 * A.h;
  const float bux = B.c2 * B.w, buy = B.s2 * B.w, bvx = -B.s2 * B.h, bvy = B.c2 * B.h;
  const float acx = 2.f * A.c2, asx = 2.f * A.s2, bcx = 2.f * B.c2, bsx = 2.f * B.s2;
  const float ahw = 0.5f * fabsf(A.w), ahh = 0.5f * fabsf(A.h), bhw = 0.5f * fabsf(B.w), bhh = 0.5f * fabsf(B.h);
  const float d1 = fabsf(dx * acx + dy * asx);
  const float p1a = fabsf(bux * acx + buy * asx), p1b = fabsf(bvx * acx + bvy * asx);
  if (d1 > (ahw + p1a + p1b) * 1.002f + 1e-3f) return true;
  const float d2 = fabsf(-dx * asx + dy * acx);
  const float p2a = fabsf(-bux * asx + buy * acx), p2b = fabsf(-bvx * asx + bvy * acx);
  if (d2 > (ahh + p2a + p2b) * 1.002f + 1e-3f) return true;
  const float d3 = fabsf(dx * bcx + dy * bsx);
  const float p3a = fabsf(aux * bcx + auy * bsx), p3b = fabsf(avx * bcx + avy * bsx);
  if (d3 > (bhw + p3a + p3b) * 1.002f + 1e-3f) return true;
  const float d4 = fabsf(-dx * bsx + dy * bcx);
  const float p4a = fabsf(-aux * bsx + auy * bcx), p4b = fabsf(-avx * bsx + avy * bcx);
  if (d4 > (bhh + p4a + p4b) * 1.002f + 1e-3f) return true;
  const float aa = A.w * A.h, ab = B.w * B.h;
  const bool sane = thr > 0.05f && fminf(aa, ab) > 0.f &&
                    fminf(fabsf(A.w), fabsf(A.h)) >= 0.05f * fmaxf(fabsf(A.w), fabsf(A.h)) &&
                    fminf(fabsf(B.w), fabsf(B.h)) >= 0.05f * fmaxf(fabsf(B.w), fabsf(B.h));
  if (!sane) return false;
  auto ov = [](float d, float ra, float rb) { return fmaxf(fminf(fminf(ra + rb - d, 2.f * ra), 2.f * rb), 0.f); };
  const float ia = ov(d1, ahw, p1a + p1b) * ov(d2, ahh, p2a + p2b);
  const float ib = ov(d3, bhw, p3a + p3b) * ov(d4, bhh, p4a + p4b);
  const float iub = fminf(ia, ib);
  const float t = 0.99f * thr;
  return iub * (1.f + t) < t * (aa + ab);      // Iub / (a1 + a2 - Iub) < 0.99 thr
}

__device__ __forceinline__ float cross2(float ax, float ay, float bx, float by) {
  return ax * by - bx * ay;  // cross_2d (:51-53)
}
__device__ __forceinline__ float dot2(float ax, float ay, float bx, float by) {
  return ax * bx + ay * by;  // dot_2d (:46-48)
}

__device__ __forceinline__ void prebox_vertices(float xc, float yc, const PreBox& b, float (&vx)[4],
                                                float (&vy)[4]) {
  // get_rotated_vertices (:66-74)
  vx[0] = xc - b.s2 * b.h - b.c2 * b.w;
  vy[0] = yc + b.c2 * b.h - b.s2 * b.w;
  vx[1] = xc + b.s2 * b.h - b.c2 * b.w;
  vy[1] = yc - b.c2 * b.h - b.s2 * b.w;
  vx[2] = 2 * xc - vx[0];
  vy[2] = 2 * yc - vy[0];
  vx[3] = 2 * xc - vx[1];
  vy[3] = 2 * yc - vy[1];
}

// Full IoU of two pre-processed boxes.  `pts` points at THIS thread's slot of an LDS
// array float2[CAP][NT]; element i of the thread is pts[i * NT].
// CAP = 24 is the reference's worst case (16 edge crossings + 8 contained vertices).  Two rectangles in general position
// produce at most 8 candidate points (the vertices of their intersection polygon); a pass that gives every lane only 8
// slots keeps three times as many waves resident (the algorithm is a chain of dependent LDS round trips: latency-bound
// at 3 waves per SIMD).  With CAP < 24 a pair with more candidates sets *overflow and returns 0: the caller redoes it
// with CAP = 24.  Same arithmetic in the same order either way.
template <int NT, int CAP = 24>
__device__ float rbox_iou(const PreBox& A, const PreBox& B, float2* pts, bool* overflow = nullptr) {
  // single_box_iou_rotated (:339-362): centre shift in double, areas in float
  float sumx = A.x + B.x, sumy = A.y + B.y;
  double shx = (double)sumx / 2.0, shy = (double)sumy / 2.0;
  float x1 = (float)((double)A.x - shx), y1 = (float)((double)A.y - shy);
  float x2 = (float)((double)B.x - shx), y2 = (float)((double)B.y - shy);
  float area1 = A.w * A.h, area2 = B.w * B.h;
  if ((double)area1 < 1e-14 || (double)area2 < 1e-14) return 0.f;

  float ax[4], ay[4], bx[4], by[4];
  prebox_vertices(x1, y1, A, ax, ay);
  prebox_vertices(x2, y2, B, bx, by);
  float eax[4], eay[4], ebx[4], eby[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    eax[i] = ax[(i + 1) & 3] - ax[i];
    eay[i] = ay[(i + 1) & 3] - ay[i];
    ebx[i] = bx[(i + 1) & 3] - bx[i];
    eby[i] = by[(i + 1) & 3] - by[i];
  }

  int n = 0;
  // get_intersection_points, edge x edge (:94-118)
#pragma unroll
  for (int i = 0; i < 4; i++) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      float det = cross2(ebx[j], eby[j], eax[i], eay[i]);
      bool parallel = fabs((double)det) <= 1e-14;
      float dx = bx[j] - ax[i], dy = by[j] - ay[i];
      float t1 = cross2(ebx[j], eby[j], dx, dy) / det;
      float t2 = cross2(eax[i], eay[i], dx, dy) / det;
      if (!parallel && t1 >= 0.0f && t1 <= 1.0f && t2 >= 0.0f && t2 <= 1.0f) {
        if (CAP >= 24 || n < CAP) pts[n * NT] = make_float2(ax[i] + eax[i] * t1, ay[i] + eay[i] * t1);
        n++;
      }
    }
  }
  // vertices of A inside B (:121-145)
  {
    float abab = dot2(ebx[0], eby[0], ebx[0], eby[0]);
    float adad = dot2(ebx[3], eby[3], ebx[3], eby[3]);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      float apx = ax[i] - bx[0], apy = ay[i] - by[0];
      float apab = dot2(apx, apy, ebx[0], eby[0]);
      float apad = -dot2(apx, apy, ebx[3], eby[3]);
      if (apab >= 0 && apad >= 0 && apab <= abab && apad <= adad) {
        if (CAP >= 24 || n < CAP) pts[n * NT] = make_float2(ax[i], ay[i]);
        n++;
      }
    }
  }
  // vertices of B inside A (:148-164)
  {
    float abab = dot2(eax[0], eay[0], eax[0], eay[0]);
    float adad = dot2(eax[3], eay[3], eax[3], eay[3]);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      float apx = bx[i] - ax[0], apy = by[i] - ay[0];
      float apab = dot2(apx, apy, eax[0], eay[0]);
      float apad = -dot2(apx, apy, eax[3], eay[3]);
      if (apab >= 0 && apad >= 0 && apab <= abab && apad <= adad) {
        if (CAP >= 24 || n < CAP) pts[n * NT] = make_float2(bx[i], by[i]);
        n++;
      }
    }
  }
  if (CAP < 24 && n > CAP) {
    *overflow = true;
    return 0.f;
  }

  float inter = 0.0f;
  if (n > 2) {  // rotated_boxes_intersection (:316-323)
    // convex_hull_graham step 1 (:181-187): lowest-y, then lowest-x point
    int t = 0;
    float2 start = pts[0];
    for (int i = 1; i < n; i++) {
      float2 p = pts[i * NT];
      if (p.y < start.y || (p.y == start.y && p.x < start.x)) {
        t = i;
        start = p;
      }
    }
    // step 2 (:191-198): shift to the start point (in place; shift_to_zero=true so the
    // original coordinates are never needed again), swap it to slot 0
    for (int i = 0; i < n; i++) {
      float2 p = pts[i * NT];
      pts[i * NT] = make_float2(p.x - start.x, p.y - start.y);
    }
    {
      float2 q0 = pts[0], qt = pts[t * NT];
      pts[0] = qt;
      pts[t * NT] = q0;
    }
    // step 3, GPU branch (:213-226).  dist[] is swapped together with q[] there, so it is
    // always dot(q[i],q[i]) of the current q[i]: recomputed instead of stored.
    for (int i = 1; i < n - 1; i++) {
      float2 qi = pts[i * NT];
      float di = dot2(qi.x, qi.y, qi.x, qi.y);
      for (int j = i + 1; j < n; j++) {
        float2 qj = pts[j * NT];
        float cp = cross2(qi.x, qi.y, qj.x, qj.y);
        float dj = dot2(qj.x, qj.y, qj.x, qj.y);
        if (((double)cp < -1e-6) || (fabs((double)cp) < 1e-6 && di > dj)) {
          pts[j * NT] = qi;
          qi = qj;
          di = dj;
        }
      }
      pts[i * NT] = qi;
    }
    // step 4 (:243-254)
    int k = 1;
    for (; k < n; k++) {
      float2 q = pts[k * NT];
      if ((double)dot2(q.x, q.y, q.x, q.y) > 1e-8) break;
    }
    if (k < n) {
      pts[1 * NT] = pts[k * NT];
      int m = 2;
      // step 5 (:263-268)
      for (int i = k + 1; i < n; i++) {
        float2 qi = pts[i * NT];
        while (m > 1) {
          float2 qa = pts[(m - 2) * NT], qb = pts[(m - 1) * NT];
          if (cross2(qi.x - qa.x, qi.y - qa.y, qb.x - qa.x, qb.y - qa.y) >= 0)
            m--;
          else
            break;
        }
        pts[m * NT] = qi;
        m++;
      }
      // polygon_area (:285-296)
      if (m > 2) {
        float2 q0 = pts[0];
        float area = 0;
        for (int i = 1; i < m - 1; i++) {
          float2 qa = pts[i * NT], qb = pts[(i + 1) * NT];
          double c = fabs((double)cross2(qa.x - q0.x, qa.y - q0.y, qb.x - q0.x, qb.y - q0.y));
          area = (float)((double)area + c);
        }
        inter = (float)((double)area / 2.0);
      }
    }
  }
  return inter / (area1 + area2 - inter);
}

}  // namespace s2a
