// ============================================================================
// s2a_oracle — CPU restatement of the reference's dense-inference hot path.
//
// TEST INFRASTRUCTURE ONLY.  Nothing in the product path (s2anet_amd/) may
// import, link or call this file; only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg use it, and only as the checker / CPU baseline.
//
// Every function cites the reference file:line whose arithmetic it restates.
// The restatement keeps the reference's operation ORDER and its mixed
// float/double promotions (they decide `iou > thr` in NMS), but is written
// independently (flat coordinate arrays, no Point/RotatedBox types).
//
// Pinning (see tests/test_oracle_pinned.py, tests/golden/):
//   * rotated IoU, CPU branch  : vs the reference's box_iou_rotated_cpu built
//                                unmodified into oracle/_ref (bit-exact)
//   * rotated IoU, GPU branch  : vs the reference header host-compiled with
//                                __CUDACC__ (oracle/ref_geom_gpubranch.cpp)
//   * NMS / ml-NMS             : vs reference nms_rotated_cpu (>= rule)
//   * polyiou                  : vs reference polyiou (SWIG wrapper), 1/7 case
//   * ARF                      : vs reference ARF_forward_cpu at small shapes
//   * deformable conv forward  : NO runnable reference (CUDA only).  Pinned by
//                                an independent torch formulation and by the
//                                zero-offset == conv2d identity: "parity
//                                unpinned" beyond that.
//   * deformable conv backward : same situation: pinned to autograd of the
//                                independent torch formulation (parity unpinned
//                                against the reference itself)
//   * RIE forward / backward (numpy, oracle/__init__.py) : vs the reference's CPU op (rie_small.npz)
//   * ARF backward, polygon NMS (py_cpu_nms_poly_fast), assign_labels, voc_eval,
//     mergesingle              : vs the reference's own CPU op / Python scripts
//                                run here (golden fixtures, identical results)
//
// Build: make -C oracle   (g++ -O2 -ffp-contract=off; x86-64 baseline has no
// FMA so this matches how the reference's CPU extension is compiled).
// ============================================================================
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

// ---------------------------------------------------------------------------
// Rotated-box geometry.  Reference: utils/box_iou_rotated/src/box_iou_rotated_utils.h
// (identical arithmetic in the nms_rotated and ml_nms_rotated copies).
// ---------------------------------------------------------------------------
constexpr int kMaxPts = 24;  // 16 edge-edge + 4 + 4 vertex hits (:305)

inline float cross2(float ax, float ay, float bx, float by) {
  // cross_2d (:51-53): A.x*B.y - B.x*A.y
  return ax * by - bx * ay;
}
inline float dot2(float ax, float ay, float bx, float by) {
  // dot_2d (:46-48)
  return ax * bx + ay * by;
}

// get_rotated_vertices (:56-75).  cos/sin are evaluated in DOUBLE on the float
// angle (radians; the degrees factor is commented out at :60) and cast to float.
inline void box_vertices(float xc, float yc, float w, float h, float a,
                         float* vx, float* vy) {
  double theta = a;
  float c2 = (float)std::cos(theta) * 0.5f;
  float s2 = (float)std::sin(theta) * 0.5f;
  vx[0] = xc - s2 * h - c2 * w;
  vy[0] = yc + c2 * h - s2 * w;
  vx[1] = xc + s2 * h - c2 * w;
  vy[1] = yc - c2 * h - s2 * w;
  vx[2] = 2 * xc - vx[0];
  vy[2] = 2 * yc - vy[0];
  vx[3] = 2 * xc - vx[1];
  vy[3] = 2 * yc - vy[1];
}

// get_intersection_points (:78-167)
inline int collect_points(const float* ax, const float* ay, const float* bx,
                          const float* by, float* ox, float* oy) {
  float eax[4], eay[4], ebx[4], eby[4];
  for (int i = 0; i < 4; i++) {
    eax[i] = ax[(i + 1) & 3] - ax[i];
    eay[i] = ay[(i + 1) & 3] - ay[i];
    ebx[i] = bx[(i + 1) & 3] - bx[i];
    eby[i] = by[(i + 1) & 3] - by[i];
  }
  int n = 0;
  // edge x edge (:94-118)
  for (int i = 0; i < 4; i++) {
    for (int j = 0; j < 4; j++) {
      float det = cross2(ebx[j], eby[j], eax[i], eay[i]);
      if (std::fabs((double)det) <= 1e-14) continue;  // parallel (:103)
      float dx = bx[j] - ax[i], dy = by[j] - ay[i];
      float t1 = cross2(ebx[j], eby[j], dx, dy) / det;
      float t2 = cross2(eax[i], eay[i], dx, dy) / det;
      if (t1 >= 0.0f && t1 <= 1.0f && t2 >= 0.0f && t2 <= 1.0f) {
        ox[n] = ax[i] + eax[i] * t1;
        oy[n] = ay[i] + eay[i] * t1;
        n++;
      }
    }
  }
  // vertices of A inside B (:121-145): AB = edge0 of B, DA = edge3 of B
  {
    float abx = ebx[0], aby = eby[0], dax = ebx[3], day = eby[3];
    float abab = dot2(abx, aby, abx, aby);
    float adad = dot2(dax, day, dax, day);
    for (int i = 0; i < 4; i++) {
      float apx = ax[i] - bx[0], apy = ay[i] - by[0];
      float apab = dot2(apx, apy, abx, aby);
      float apad = -dot2(apx, apy, dax, day);
      if (apab >= 0 && apad >= 0 && apab <= abab && apad <= adad) {
        ox[n] = ax[i];
        oy[n] = ay[i];
        n++;
      }
    }
  }
  // vertices of B inside A (:148-164)
  {
    float abx = eax[0], aby = eay[0], dax = eax[3], day = eay[3];
    float abab = dot2(abx, aby, abx, aby);
    float adad = dot2(dax, day, dax, day);
    for (int i = 0; i < 4; i++) {
      float apx = bx[i] - ax[0], apy = by[i] - ay[0];
      float apab = dot2(apx, apy, abx, aby);
      float apad = -dot2(apx, apy, dax, day);
      if (apab >= 0 && apad >= 0 && apab <= abab && apad <= adad) {
        ox[n] = bx[i];
        oy[n] = by[i];
        n++;
      }
    }
  }
  return n;
}

struct P2 {
  float x, y;
};

// convex_hull_graham with shift_to_zero=true (:170-282) followed by
// polygon_area (:285-296).  sort_mode 0 = host branch (std::sort, dist[] left
// STALE, :229-237); sort_mode 1 = __CUDACC__ branch (O(n^2) swap sort that
// permutes dist[] with q[], :209-226).
inline float hull_area(const float* px, const float* py, int n, int sort_mode) {
  int t = 0;
  for (int i = 1; i < n; i++)
    if (py[i] < py[t] || (py[i] == py[t] && px[i] < px[t])) t = i;
  P2 q[kMaxPts];
  float dist[kMaxPts];
  for (int i = 0; i < n; i++) {
    q[i].x = px[i] - px[t];
    q[i].y = py[i] - py[t];
  }
  std::swap(q[0], q[t]);
  for (int i = 0; i < n; i++) dist[i] = dot2(q[i].x, q[i].y, q[i].x, q[i].y);

  if (sort_mode == 1) {
    for (int i = 1; i < n - 1; i++)
      for (int j = i + 1; j < n; j++) {
        float cp = cross2(q[i].x, q[i].y, q[j].x, q[j].y);
        if (((double)cp < -1e-6) ||
            (std::fabs((double)cp) < 1e-6 && dist[i] > dist[j])) {
          std::swap(q[i], q[j]);
          std::swap(dist[i], dist[j]);
        }
      }
  } else {
    std::sort(q + 1, q + n, [](const P2& A, const P2& B) -> bool {
      float c = cross2(A.x, A.y, B.x, B.y);
      if (std::fabs((double)c) < 1e-6)
        return dot2(A.x, A.y, A.x, A.y) < dot2(B.x, B.y, B.x, B.y);
      return c > 0;
    });
  }
  int k = 1;
  for (; k < n; k++)
    if ((double)dist[k] > 1e-8) break;  // (:244-248), dist possibly stale
  if (k == n) return 0.0f;              // single point -> area 0 (:249-253, :286)
  q[1] = q[k];
  int m = 2;
  for (int i = k + 1; i < n; i++) {
    while (m > 1 && cross2(q[i].x - q[m - 2].x, q[i].y - q[m - 2].y,
                           q[m - 1].x - q[m - 2].x, q[m - 1].y - q[m - 2].y) >= 0)
      m--;
    q[m++] = q[i];
  }
  if (m <= 2) return 0.0f;
  float area = 0;
  for (int i = 1; i < m - 1; i++) {
    // area += fabs(cross) with fabs(double) (:292): the float sum goes through double once
    double c = std::fabs((double)cross2(q[i].x - q[0].x, q[i].y - q[0].y,
                                        q[i + 1].x - q[0].x, q[i + 1].y - q[0].y));
    area = (float)((double)area + c);
  }
  return (float)((double)area / 2.0);
}

// single_box_iou_rotated (:333-375; ml copy :314-344 adds the label test).
// b = x,y,w,h,a[,label].
inline float iou_pair(const float* b1, const float* b2, int sort_mode, bool with_label) {
  if (with_label && b1[5] != b2[5]) return 0.0f;  // ml_nms..._utils.h:319-322
  double sx = (double)(b1[0] + b2[0]) / 2.0;
  double sy = (double)(b1[1] + b2[1]) / 2.0;
  float x1 = (float)((double)b1[0] - sx), y1 = (float)((double)b1[1] - sy);
  float x2 = (float)((double)b2[0] - sx), y2 = (float)((double)b2[1] - sy);
  float area1 = b1[2] * b1[3];
  float area2 = b2[2] * b2[3];
  if ((double)area1 < 1e-14 || (double)area2 < 1e-14) return 0.f;
  float ax[4], ay[4], bx[4], by[4];
  box_vertices(x1, y1, b1[2], b1[3], b1[4], ax, ay);
  box_vertices(x2, y2, b2[2], b2[3], b2[4], bx, by);
  float px[kMaxPts], py[kMaxPts];
  int n = collect_points(ax, ay, bx, by, px, py);
  float inter = 0.0f;
  if (n > 2) inter = hull_area(px, py, n, sort_mode);
  return inter / (area1 + area2 - inter);
}

// Provably-exact shortcut used only to make the 200k-row NMS check finish in
// seconds: boxes whose circumscribed circles are >0.2 % apart share no point,
// the reference then finds num == 0 and returns exactly 0.0f
// (box_iou_rotated_utils.h:316-318).  Validated against the plain path in tests.
inline bool surely_disjoint(const float* b1, const float* b2) {
  float dx = b1[0] - b2[0], dy = b1[1] - b2[1];
  float r1 = 0.5f * std::sqrt(b1[2] * b1[2] + b1[3] * b1[3]);
  float r2 = 0.5f * std::sqrt(b2[2] * b2[2] + b2[3] * b2[3]);
  float R = (r1 + r2) * 1.002f + 1e-3f;
  return dx * dx + dy * dy > R * R;
}

// ---------------------------------------------------------------------------
// polyiou.  Reference: DOTA_devkit/polyiou/csrc/polyiou.cpp
// ---------------------------------------------------------------------------
struct D2 {
  double x, y;
};
inline int sgn(double d) { return (d > 1e-8) - (d < -1e-8); }  // sig() :8-12
inline double tri_cross(D2 o, D2 a, D2 b) {                     // cross() :20-22
  return (a.x - o.x) * (b.y - o.y) - (b.x - o.x) * (a.y - o.y);
}
inline bool same_pt(D2 a, D2 b) { return sgn(a.x - b.x) == 0 && sgn(a.y - b.y) == 0; }
inline double shoelace(D2* ps, int n) {  // area() :23-30 (writes ps[n])
  ps[n] = ps[0];
  double r = 0;
  for (int i = 0; i < n; i++) r += ps[i].x * ps[i + 1].y - ps[i].y * ps[i + 1].x;
  return r / 2.0;
}
inline int line_hit(D2 a, D2 b, D2 c, D2 d, D2& p) {  // lineCross() :31-41
  double s1 = tri_cross(a, b, c), s2 = tri_cross(a, b, d);
  if (sgn(s1) == 0 && sgn(s2) == 0) return 2;
  if (sgn(s2 - s1) == 0) return 0;
  p.x = (c.x * s2 - d.x * s1) / (s2 - s1);
  p.y = (c.y * s2 - d.y * s1) / (s2 - s1);
  return 1;
}
inline void half_plane_cut(D2* p, int& n, D2 a, D2 b, D2* tmp) {  // polygon_cut() :58-71
  int m = 0;
  p[n] = p[0];
  for (int i = 0; i < n; i++) {
    if (sgn(tri_cross(a, b, p[i])) > 0) tmp[m++] = p[i];
    if (sgn(tri_cross(a, b, p[i])) != sgn(tri_cross(a, b, p[i + 1])))
      line_hit(a, b, p[i], p[i + 1], tmp[m++]);
  }
  n = 0;
  for (int i = 0; i < m; i++)
    if (!i || !same_pt(tmp[i], tmp[i - 1])) p[n++] = tmp[i];
  while (n > 1 && same_pt(p[n - 1], p[0])) n--;
}
inline double fan_overlap(D2 a, D2 b, D2 c, D2 d) {  // intersectArea(a,b,c,d) :74-90
  D2 o{0, 0};
  int s1 = sgn(tri_cross(o, a, b)), s2 = sgn(tri_cross(o, c, d));
  if (s1 == 0 || s2 == 0) return 0.0;
  if (s1 == -1) std::swap(a, b);
  if (s2 == -1) std::swap(c, d);
  D2 p[10] = {o, a, b};
  D2 tmp[51];
  int n = 3;
  half_plane_cut(p, n, o, c, tmp);
  half_plane_cut(p, n, c, d, tmp);
  half_plane_cut(p, n, d, o, tmp);
  double r = std::fabs(shoelace(p, n));
  if (s1 * s2 == -1) r = -r;
  return r;
}
inline double quad_iou(const double* p, const double* q) {  // iou_poly :108-128 + :92-103
  D2 a[51], b[51];
  for (int i = 0; i < 4; i++) {
    a[i] = {p[2 * i], p[2 * i + 1]};
    b[i] = {q[2 * i], q[2 * i + 1]};
  }
  if (shoelace(a, 4) < 0) std::reverse(a, a + 4);
  if (shoelace(b, 4) < 0) std::reverse(b, b + 4);
  a[4] = a[0];
  b[4] = b[0];
  double inter = 0;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) inter += fan_overlap(a[i], a[i + 1], b[j], b[j + 1]);
  double uni = std::fabs(shoelace(a, 4)) + std::fabs(shoelace(b, 4)) - inter;
  return inter / uni;
}

// ---------------------------------------------------------------------------
// The same geometry instantiated for T = double (the NMS ops dispatch on the dtype of `dets`:
// AT_DISPATCH_FLOATING_TYPES, utils/nms_rotated/src/nms_rotated_cpu.cpp:66, utils/ml_nms_rotated/src/nms_rotated_cpu.cpp:67;
// AT_DISPATCH_FLOATING_TYPES_AND_HALF in the CUDA files :95-100 / :100-105).  With T = double every intermediate of
// box_iou_rotated_utils.h is a double and the float-only promotions above disappear; the 0.5f / 1e-14 / 1e-6 / 1e-8
// constants keep their values.  Written out separately so that the float path above stays byte for byte what the
// fixtures pinned.
// ---------------------------------------------------------------------------
namespace d64 {
struct Q2 {
  double x, y;
};
inline double cr(double ax, double ay, double bx, double by) { return ax * by - bx * ay; }
inline double dt(double ax, double ay, double bx, double by) { return ax * bx + ay * by; }
inline void verts(double xc, double yc, double w, double h, double a, double* vx, double* vy) {
  double c2 = std::cos(a) * 0.5f, s2 = std::sin(a) * 0.5f;     // (:62-64) theta is the double angle itself
  vx[0] = xc - s2 * h - c2 * w;
  vy[0] = yc + c2 * h - s2 * w;
  vx[1] = xc + s2 * h - c2 * w;
  vy[1] = yc - c2 * h - s2 * w;
  vx[2] = 2 * xc - vx[0];
  vy[2] = 2 * yc - vy[0];
  vx[3] = 2 * xc - vx[1];
  vy[3] = 2 * yc - vy[1];
}
inline int points(const double* ax, const double* ay, const double* bx, const double* by, double* ox, double* oy) {
  double eax[4], eay[4], ebx[4], eby[4];
  for (int i = 0; i < 4; i++) {
    eax[i] = ax[(i + 1) & 3] - ax[i];
    eay[i] = ay[(i + 1) & 3] - ay[i];
    ebx[i] = bx[(i + 1) & 3] - bx[i];
    eby[i] = by[(i + 1) & 3] - by[i];
  }
  int n = 0;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      double det = cr(ebx[j], eby[j], eax[i], eay[i]);
      if (std::fabs(det) <= 1e-14) continue;
      double dx = bx[j] - ax[i], dy = by[j] - ay[i];
      double t1 = cr(ebx[j], eby[j], dx, dy) / det;
      double t2 = cr(eax[i], eay[i], dx, dy) / det;
      if (t1 >= 0.0f && t1 <= 1.0f && t2 >= 0.0f && t2 <= 1.0f) {
        ox[n] = ax[i] + eax[i] * t1;
        oy[n] = ay[i] + eay[i] * t1;
        n++;
      }
    }
  for (int side = 0; side < 2; side++) {          // vertices of A inside B, then of B inside A
    const double* px = side ? bx : ax; const double* py = side ? by : ay;
    const double* rx = side ? ax : bx; const double* ry = side ? ay : by;
    const double* ex = side ? eax : ebx; const double* ey = side ? eay : eby;
    double abx = ex[0], aby = ey[0], dax = ex[3], day = ey[3];
    double abab = dt(abx, aby, abx, aby), adad = dt(dax, day, dax, day);
    for (int i = 0; i < 4; i++) {
      double apx = px[i] - rx[0], apy = py[i] - ry[0];
      double apab = dt(apx, apy, abx, aby), apad = -dt(apx, apy, dax, day);
      if (apab >= 0 && apad >= 0 && apab <= abab && apad <= adad) {
        ox[n] = px[i];
        oy[n] = py[i];
        n++;
      }
    }
  }
  return n;
}
inline double hull(const double* px, const double* py, int n, int sort_mode) {
  int t = 0;
  for (int i = 1; i < n; i++)
    if (py[i] < py[t] || (py[i] == py[t] && px[i] < px[t])) t = i;
  Q2 q[kMaxPts];
  double dist[kMaxPts];
  for (int i = 0; i < n; i++) q[i] = {px[i] - px[t], py[i] - py[t]};
  std::swap(q[0], q[t]);
  for (int i = 0; i < n; i++) dist[i] = dt(q[i].x, q[i].y, q[i].x, q[i].y);
  if (sort_mode == 1) {
    for (int i = 1; i < n - 1; i++)
      for (int j = i + 1; j < n; j++) {
        double cp = cr(q[i].x, q[i].y, q[j].x, q[j].y);
        if ((cp < -1e-6) || (std::fabs(cp) < 1e-6 && dist[i] > dist[j])) {
          std::swap(q[i], q[j]);
          std::swap(dist[i], dist[j]);
        }
      }
  } else {
    std::sort(q + 1, q + n, [](const Q2& A, const Q2& B) -> bool {
      double c = cr(A.x, A.y, B.x, B.y);
      if (std::fabs(c) < 1e-6) return dt(A.x, A.y, A.x, A.y) < dt(B.x, B.y, B.x, B.y);
      return c > 0;
    });
  }
  int k = 1;
  for (; k < n; k++)
    if (dist[k] > 1e-8) break;
  if (k == n) return 0.0;
  q[1] = q[k];
  int m = 2;
  for (int i = k + 1; i < n; i++) {
    while (m > 1 && cr(q[i].x - q[m - 2].x, q[i].y - q[m - 2].y, q[m - 1].x - q[m - 2].x, q[m - 1].y - q[m - 2].y) >= 0) m--;
    q[m++] = q[i];
  }
  if (m <= 2) return 0.0;
  double area = 0;
  for (int i = 1; i < m - 1; i++)
    area += std::fabs(cr(q[i].x - q[0].x, q[i].y - q[0].y, q[i + 1].x - q[0].x, q[i + 1].y - q[0].y));
  return area / 2.0;
}
inline double iou(const double* b1, const double* b2, int sort_mode, bool with_label) {
  if (with_label && b1[5] != b2[5]) return 0.0;
  double sx = (b1[0] + b2[0]) / 2.0, sy = (b1[1] + b2[1]) / 2.0;
  double area1 = b1[2] * b1[3], area2 = b2[2] * b2[3];
  if (area1 < 1e-14 || area2 < 1e-14) return 0.0;
  double ax[4], ay[4], bx[4], by[4];
  verts(b1[0] - sx, b1[1] - sy, b1[2], b1[3], b1[4], ax, ay);
  verts(b2[0] - sx, b2[1] - sy, b2[2], b2[3], b2[4], bx, by);
  double px[kMaxPts], py[kMaxPts];
  int n = points(ax, ay, bx, by, px, py);
  double inter = n > 2 ? hull(px, py, n, sort_mode) : 0.0;
  return inter / (area1 + area2 - inter);
}
}  // namespace d64

}  // namespace

extern "C" {

// ----- rotated IoU / NMS on double boxes (the dtype dispatch of the NMS ops) -------------------
void orc_iou_pairs_f64(const double* b1, const double* b2, int64_t n, int stride, double* out, int sort_mode) {
  for (int64_t i = 0; i < n; i++) out[i] = d64::iou(b1 + stride * i, b2 + stride * i, sort_mode, stride == 6);
}
// as orc_nms_rotated; the threshold stays a float (`const float iou_threshold` in both reference files) and is promoted
int64_t orc_nms_rotated_f64(const double* dets5, const double* scores, const double* labels, int64_t n, float thr,
                            int rule, int sort_mode, int64_t* keep) {
  if (n == 0) return 0;
  std::vector<int64_t> order(n);
  for (int64_t i = 0; i < n; i++) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return scores[a] > scores[b]; });
  std::vector<double> box(6 * n);
  for (int64_t i = 0; i < n; i++) {
    std::memcpy(&box[6 * i], dets5 + 5 * i, 5 * sizeof(double));
    box[6 * i + 5] = labels ? labels[i] : 0.0;
  }
  std::vector<uint8_t> dead(n, 0);
  int64_t k = 0;
  for (int64_t oi = 0; oi < n; oi++) {
    int64_t i = order[oi];
    if (dead[i]) continue;
    keep[k++] = i;
    for (int64_t oj = oi + 1; oj < n; oj++) {
      int64_t j = order[oj];
      if (dead[j]) continue;
      double v = d64::iou(&box[6 * i], &box[6 * j], sort_mode, true);
      if (rule == 0 ? (v >= thr) : (v > thr)) dead[j] = 1;
    }
  }
  return k;
}

// ----- rotated IoU ---------------------------------------------------------
float orc_iou_single(const float* b1, const float* b2, int sort_mode) {
  return iou_pair(b1, b2, sort_mode, false);
}

// box_iou_rotated_cpu (utils/box_iou_rotated/src/box_iou_rotated_cpu.cpp:7-45):
// out[i*M+j] = iou(boxes1[i], boxes2[j]).  cull!=0 enables the exact shortcut.
void orc_box_iou_rotated(const float* boxes1, int64_t n, const float* boxes2, int64_t m,
                         float* out, int sort_mode, int cull) {
  for (int64_t i = 0; i < n; i++)
    for (int64_t j = 0; j < m; j++) {
      const float* a = boxes1 + 5 * i;
      const float* b = boxes2 + 5 * j;
      out[i * m + j] = (cull && surely_disjoint(a, b)) ? 0.0f : iou_pair(a, b, sort_mode, false);
    }
}

// element-wise pairs, stride = floats per box (5, or 6 with label test)
void orc_iou_pairs(const float* b1, const float* b2, int64_t n, int stride, float* out,
                   int sort_mode) {
  for (int64_t i = 0; i < n; i++)
    out[i] = iou_pair(b1 + stride * i, b2 + stride * i, sort_mode, stride == 6);
}

// ----- NMS -----------------------------------------------------------------
// nms_rotated_cpu_kernel (utils/ml_nms_rotated/src/nms_rotated_cpu.cpp:7-58,
// utils/nms_rotated/src/nms_rotated_cpu.cpp:7-56) and the GPU op's semantics
// (utils/ml_nms_rotated/src/nms_rotated_cuda.cu:14-137).
//   labels == nullptr : single-class nms_rotated
//   rule 0: suppress when iou >= thr (reference CPU, cpu.cpp:52)
//   rule 1: suppress when iou >  thr (reference GPU, cuda.cu:63-64)
// Order: scores descending; ties by ascending index (a stable sort — the
// reference's torch sort leaves ties unspecified).  keep[] receives indices into
// the ORIGINAL order, in descending-score order; returns their count.
int64_t orc_nms_rotated(const float* dets5, const float* scores, const float* labels,
                        int64_t n, float thr, int rule, int sort_mode, int cull,
                        int64_t* keep) {
  if (n == 0) return 0;
  std::vector<int64_t> order(n);
  for (int64_t i = 0; i < n; i++) order[i] = i;
  std::stable_sort(order.begin(), order.end(),
                   [&](int64_t a, int64_t b) { return scores[a] > scores[b]; });
  std::vector<float> box(6 * n);
  for (int64_t i = 0; i < n; i++) {
    std::memcpy(&box[6 * i], dets5 + 5 * i, 5 * sizeof(float));
    box[6 * i + 5] = labels ? labels[i] : 0.0f;
  }
  std::vector<uint8_t> dead(n, 0);
  int64_t k = 0;
  for (int64_t oi = 0; oi < n; oi++) {
    int64_t i = order[oi];
    if (dead[i]) continue;
    keep[k++] = i;
    const float* bi = &box[6 * i];
    for (int64_t oj = oi + 1; oj < n; oj++) {
      int64_t j = order[oj];
      if (dead[j]) continue;
      const float* bj = &box[6 * j];
      float v;
      if (bi[5] != bj[5]) v = 0.0f;
      else if (cull && surely_disjoint(bi, bj)) v = 0.0f;
      else v = iou_pair(bi, bj, sort_mode, true);
      if (rule == 0 ? (v >= thr) : (v > thr)) dead[j] = 1;
    }
  }
  return k;
}

// smallest |iou - thr| over all same-label pairs that the greedy scan evaluates;
// fixtures record it so a `>` vs `>=` or a 1-ulp difference can be ruled out.
double orc_nms_margin(const float* dets5, const float* labels, int64_t n, float thr,
                      int sort_mode) {
  double best = 1e30;
  for (int64_t i = 0; i < n; i++)
    for (int64_t j = i + 1; j < n; j++) {
      if (labels && labels[i] != labels[j]) continue;
      if (surely_disjoint(dets5 + 5 * i, dets5 + 5 * j)) continue;
      float v = iou_pair(dets5 + 5 * i, dets5 + 5 * j, sort_mode, false);
      float w = iou_pair(dets5 + 5 * j, dets5 + 5 * i, sort_mode, false);
      best = std::min(best, std::fabs((double)v - (double)thr));
      best = std::min(best, std::fabs((double)w - (double)thr));
    }
  return best;
}

// ----- polyiou --------------------------------------------------------------
double orc_polyiou(const double* p8, const double* q8) { return quad_iou(p8, q8); }
void orc_polyiou_pairs(const double* p8, const double* q8, int64_t n, double* out) {
  for (int64_t i = 0; i < n; i++) out[i] = quad_iou(p8 + 8 * i, q8 + 8 * i);
}

// ----- polygon NMS of the chip-merge step -------------------------------------
// py_cpu_nms_poly_fast (DOTA_devkit/ResultMerge_multi_process.py:62-123).  dets[n,9] = x1,y1..x4,y4,score
// (float64).  Order: score descending; ties by ascending index (the reference's
// `scores.argsort()[::-1]` leaves ties to numpy's unstable sort).  A remaining j survives a kept i
// iff hbb_ovr <= thresh, where hbb_ovr is replaced by polyiou(i, j) when the axis-aligned boxes
// overlap (hbb_ovr > 0) (:87-115).  keep[] = original indices in descending-score order.
int64_t orc_nms_poly(const double* dets, int64_t n, double thresh, int64_t* keep) {
  std::vector<int64_t> order(n);
  for (int64_t i = 0; i < n; i++) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return dets[9 * a + 8] > dets[9 * b + 8]; });
  std::vector<double> x1(n), y1(n), x2(n), y2(n), area(n);
  for (int64_t i = 0; i < n; i++) {
    const double* d = dets + 9 * i;
    x1[i] = std::min(std::min(d[0], d[2]), std::min(d[4], d[6]));
    y1[i] = std::min(std::min(d[1], d[3]), std::min(d[5], d[7]));
    x2[i] = std::max(std::max(d[0], d[2]), std::max(d[4], d[6]));
    y2[i] = std::max(std::max(d[1], d[3]), std::max(d[5], d[7]));
    area[i] = (x2[i] - x1[i] + 1) * (y2[i] - y1[i] + 1);
  }
  std::vector<uint8_t> dead(n, 0);
  int64_t k = 0;
  for (int64_t oi = 0; oi < n; oi++) {
    int64_t i = order[oi];
    if (dead[i]) continue;
    keep[k++] = i;
    for (int64_t oj = oi + 1; oj < n; oj++) {
      int64_t j = order[oj];
      if (dead[j]) continue;
      double w = std::max(0.0, std::min(x2[i], x2[j]) - std::max(x1[i], x1[j]));
      double h = std::max(0.0, std::min(y2[i], y2[j]) - std::max(y1[i], y1[j]));
      double inter = w * h;
      double ovr = inter / (area[i] + area[j] - inter);
      if (ovr > 0) ovr = quad_iou(dets + 9 * i, dets + 9 * j);
      if (!(ovr <= thresh)) dead[j] = 1;
    }
  }
  return k;
}

// ----- deformable convolution, backward (training side) ----------------------------
// Restates deform_conv_backward_input_cuda / _parameters_cuda (models/dcn/src/deform_conv_cuda.cpp:262-489)
// with the device functions of deform_conv_cuda_kernel.cu: columns = W^T x gradOutput; gradOffset by
// get_coordinate_weight (:144-187, :372-429); gradInput by get_gradient_weight (:116-142, :278-330);
// gradWeight = gradOutput x im2col^T.  NCHW, groups = 1, deformable_groups = dg.  Coordinates and
// bilinear terms in float as the kernels; the two reductions accumulate in double.
static inline float orc_grad_weight(float ah, float aw, int h, int w, int H, int W) {
  if (ah <= -1 || ah >= H || aw <= -1 || aw >= W) return 0.f;
  int hl = (int)std::floor(ah), wl = (int)std::floor(aw), hh = hl + 1, wh = wl + 1;
  float weight = 0.f;
  if (h == hl && w == wl) weight = (h + 1 - ah) * (w + 1 - aw);
  if (h == hl && w == wh) weight = (h + 1 - ah) * (aw + 1 - w);
  if (h == hh && w == wl) weight = (ah + 1 - h) * (w + 1 - aw);
  if (h == hh && w == wh) weight = (ah + 1 - h) * (aw + 1 - w);
  return weight;
}
static inline float orc_coord_weight(float ah, float aw, int H, int W, const float* im, int dir) {
  if (ah <= -1 || ah >= H || aw <= -1 || aw >= W) return 0.f;
  int hl = (int)std::floor(ah), wl = (int)std::floor(aw), hh = hl + 1, wh = wl + 1;
  float weight = 0.f;
  if (dir == 0) {
    if (hl >= 0 && wl >= 0) weight += -1 * (wl + 1 - aw) * im[hl * W + wl];
    if (hl >= 0 && wh <= W - 1) weight += -1 * (aw - wl) * im[hl * W + wh];
    if (hh <= H - 1 && wl >= 0) weight += (wl + 1 - aw) * im[hh * W + wl];
    if (hh <= H - 1 && wh <= W - 1) weight += (aw - wl) * im[hh * W + wh];
  } else {
    if (hl >= 0 && wl >= 0) weight += -1 * (hl + 1 - ah) * im[hl * W + wl];
    if (hl >= 0 && wh <= W - 1) weight += (hl + 1 - ah) * im[hl * W + wh];
    if (hh <= H - 1 && wl >= 0) weight += -1 * (ah - hl) * im[hh * W + wl];
    if (hh <= H - 1 && wh <= W - 1) weight += (ah - hl) * im[hh * W + wh];
  }
  return weight;
}
static inline float orc_bilinear(const float* im, int H, int W, float h, float w) {
  int hl = (int)std::floor(h), wl = (int)std::floor(w), hh_ = hl + 1, wh = wl + 1;
  float lh = h - hl, lw = w - wl, hh = 1 - lh, hw = 1 - lw;
  float v1 = (hl >= 0 && wl >= 0) ? im[hl * W + wl] : 0.f;
  float v2 = (hl >= 0 && wh <= W - 1) ? im[hl * W + wh] : 0.f;
  float v3 = (hh_ <= H - 1 && wl >= 0) ? im[hh_ * W + wl] : 0.f;
  float v4 = (hh_ <= H - 1 && wh <= W - 1) ? im[hh_ * W + wh] : 0.f;
  return hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4;
}
void orc_deform_conv_backward(const float* x, const float* offset, const float* weight, const float* gout,
                              int64_t B, int64_t C, int64_t H, int64_t W, int64_t O, int kh, int kw, int sh,
                              int sw, int ph, int pw, int dh, int dw, int dg, float* gx, float* goff,
                              float* gw) {
  const int64_t Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1, Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;
  const int64_t K = kh * kw, HoWo = Ho * Wo, cpg = C / dg;
  std::vector<double> gxd((size_t)(B * C * H * W), 0.0), gwd((size_t)(O * C * K), 0.0);
  for (int64_t b = 0; b < B; b++)
    for (int64_t ho = 0; ho < Ho; ho++)
      for (int64_t wo = 0; wo < Wo; wo++) {
        const int64_t p = ho * Wo + wo;
        for (int64_t c = 0; c < C; c++) {
          const int64_t dgi = c / cpg;
          const float* im = x + (b * C + c) * H * W;
          for (int i = 0; i < kh; i++)
            for (int j = 0; j < kw; j++) {
              const int t = i * kw + j;
              const float* offp = offset + ((b * dg + dgi) * 2 * K) * HoWo + p;
              const float oh = offp[(2 * t) * HoWo], ow = offp[(2 * t + 1) * HoWo];
              const float h_im = (float)(ho * sh - ph + i * dh) + oh, w_im = (float)(wo * sw - pw + j * dw) + ow;
              // column value of the backward-input GEMM (float, as addmm_ stores it)
              double cd = 0.0;
              for (int64_t o = 0; o < O; o++) cd += (double)weight[(o * C + c) * K + t] * (double)gout[(b * O + o) * HoWo + p];
              const float col = (float)cd;
              // gradOffset (col2im_coord)
              float ih = h_im, iw = w_im;
              if (ih <= -1 || iw <= -1 || ih >= H || iw >= W) ih = iw = -2;
              goff[((b * dg + dgi) * 2 * K + 2 * t) * HoWo + p] += orc_coord_weight(ih, iw, (int)H, (int)W, im, 0) * col;
              goff[((b * dg + dgi) * 2 * K + 2 * t + 1) * HoWo + p] += orc_coord_weight(ih, iw, (int)H, (int)W, im, 1) * col;
              // gradInput (col2im)
              const int ch = (int)h_im, cw = (int)w_im;
              for (int dy = -2; dy <= 2; dy++)
                for (int dx = -2; dx <= 2; dx++) {
                  const int y = ch + dy, xx = cw + dx;
                  if (y >= 0 && y < H && xx >= 0 && xx < W && std::fabs(h_im - y) < 1 && std::fabs(w_im - xx) < 1)
                    gxd[(size_t)(((b * C + c) * H + y) * W + xx)] += (double)(orc_grad_weight(h_im, w_im, y, xx, (int)H, (int)W) * col);
                }
              // gradWeight (im2col x gradOutput)
              float val = 0.f;
              if (h_im > -1 && w_im > -1 && h_im < H && w_im < W) val = orc_bilinear(im, (int)H, (int)W, h_im, w_im);
              if (val != 0.f)
                for (int64_t o = 0; o < O; o++) gwd[(size_t)((o * C + c) * K + t)] += (double)val * (double)gout[(b * O + o) * HoWo + p];
            }
        }
      }
  for (size_t i = 0; i < gxd.size(); i++) gx[i] = (float)gxd[i];
  for (size_t i = 0; i < gwd.size(); i++) gw[i] = (float)gwd[i];
}

// ----- ORN: active rotating filter ------------------------------------------
// ARF_forward (models/orn/src/cuda/ActiveRotatingFilter_cuda.cu:20-46, the
// int-indexed GPU kernel; the CPU file's uint16 weightIndex wraps at 65536 and
// is wrong at [32,256,1,3,3] — SURVEY.md a7 — so this is the plain definition):
//   out[(i*nRot + k), j*nEntry + idx[l,k]-1] = w[i, j, l],  nEntry = nOri*kH*kW
void orc_arf_forward(const float* w, const uint8_t* idx, int64_t nOut, int64_t nIn,
                     int nOri, int kH, int kW, int nRot, float* out) {
  const int64_t nEntry = (int64_t)nOri * kH * kW;
  for (int64_t i = 0; i < nOut; i++)
    for (int64_t j = 0; j < nIn; j++)
      for (int64_t l = 0; l < nEntry; l++) {
        float v = w[(i * nIn + j) * nEntry + l];
        for (int k = 0; k < nRot; k++) {
          int64_t t = (int64_t)idx[l * nRot + k] - 1;
          out[((i * nRot + k) * nIn + j) * nEntry + t] = v;
        }
      }
}

// ARF_backward (models/orn/src/cuda/ActiveRotatingFilter_cuda.cu:49-76; CPU :41-79):
//   gradInput[i, j, l] = sum_{k=0..nRot-1} gradOutput[(i*nRot + k), j*nEntry + idx[l,k]-1]
// accumulated in ascending k from 0, as the reference does (float order matters for bit parity).
void orc_arf_backward(const float* gout, const uint8_t* idx, int64_t nOut, int64_t nIn, int nOri, int kH,
                      int kW, int nRot, float* gin) {
  const int64_t nEntry = (int64_t)nOri * kH * kW;
  for (int64_t i = 0; i < nOut; i++)
    for (int64_t j = 0; j < nIn; j++)
      for (int64_t l = 0; l < nEntry; l++) {
        float v = 0;
        for (int k = 0; k < nRot; k++) {
          int64_t t = (int64_t)idx[l * nRot + k] - 1;
          v = v + gout[((i * nRot + k) * nIn + j) * nEntry + t];
        }
        gin[(i * nIn + j) * nEntry + l] = v;
      }
}

// RotationInvariantPooling (models/orn/functions/rotation_invariant_pooling.py:19-27)
// x[B, C, HW] -> out[B, C/nOri, HW], max over groups of nOri consecutive channels
void orc_rot_inv_pool(const float* x, int64_t B, int64_t C, int64_t HW, int nOri, float* out) {
  int64_t G = C / nOri;
  for (int64_t b = 0; b < B; b++)
    for (int64_t g = 0; g < G; g++)
      for (int64_t p = 0; p < HW; p++) {
        float m = x[(b * C + g * nOri) * HW + p];
        for (int o = 1; o < nOri; o++) m = std::max(m, x[(b * C + g * nOri + o) * HW + p]);
        out[(b * G + g) * HW + p] = m;
      }
}

// ----- deformable convolution forward (DCN v1) -------------------------------
// deformable_im2col_bilinear (models/dcn/src/deform_conv_cuda_kernel.cu:83-114)
static inline float bilinear_at(const float* plane, int H, int W, float h, float w) {
  int h_low = (int)std::floor(h), w_low = (int)std::floor(w);
  int h_high = h_low + 1, w_high = w_low + 1;
  float lh = h - h_low, lw = w - w_low;
  float hh = 1 - lh, hw = 1 - lw;
  float v1 = 0, v2 = 0, v3 = 0, v4 = 0;
  if (h_low >= 0 && w_low >= 0) v1 = plane[h_low * W + w_low];
  if (h_low >= 0 && w_high <= W - 1) v2 = plane[h_low * W + w_high];
  if (h_high <= H - 1 && w_low >= 0) v3 = plane[h_high * W + w_low];
  if (h_high <= H - 1 && w_high <= W - 1) v4 = plane[h_high * W + w_high];
  float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
  return (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);
}

// deform_conv_forward_cuda (models/dcn/src/deform_conv_cuda.cpp:152-260) =
// deformable_im2col (deform_conv_cuda_kernel.cu:189-242) + addmm per group.
//   x[B,C,H,W], offset[B, dg*2*kH*kW, Ho, Wo], weight[O, C/groups, kH, kW] -> out[B,O,Ho,Wo]
// The column value is formed in float exactly as the kernel does; the channel
// contraction accumulates in double (the reference's cuBLAS order is
// unspecified; double is the neutral choice for a 1e-4 tolerance).
// round_cols_f16 != 0 rounds every sampled column value to IEEE half first — the
// oracle of the f16 path (fp16 columns, fp32+ accumulate).
extern float orc_round_f16(float v);
void orc_deform_conv_forward(const float* x, const float* offset, const float* weight,
                             int64_t B, int64_t C, int64_t H, int64_t W, int64_t O,
                             int kH, int kW, int sH, int sW, int pH, int pW, int dH, int dW,
                             int groups, int dgroups, int round_cols_f16, int relu,
                             float* out) {
  const int64_t Ho = (H + 2 * pH - (dH * (kH - 1) + 1)) / sH + 1;
  const int64_t Wo = (W + 2 * pW - (dW * (kW - 1) + 1)) / sW + 1;
  const int64_t Cg = C / groups, Og = O / groups, cpdg = C / dgroups;
  const int64_t K = Cg * kH * kW;
  // rows of the output are independent: OpenMP over (b, ho) — only used to make the CPU
  // baseline of bench.py finish in seconds; the arithmetic per output is unchanged
#pragma omp parallel for collapse(2) schedule(dynamic)
  for (int64_t b = 0; b < B; b++)
    for (int64_t ho = 0; ho < Ho; ho++) {
      std::vector<float> col(K);
      for (int64_t wo = 0; wo < Wo; wo++)
        for (int g = 0; g < groups; g++) {
          for (int64_t cl = 0; cl < Cg; cl++) {
            int64_t c = g * Cg + cl;
            int64_t dg = c / cpdg;
            const float* plane = x + (b * C + c) * H * W;
            const float* offp = offset + (b * dgroups + dg) * 2 * kH * kW * Ho * Wo;
            for (int i = 0; i < kH; i++)
              for (int j = 0; j < kW; j++) {
                float oh = offp[((2 * (i * kW + j)) * Ho + ho) * Wo + wo];
                float ow = offp[((2 * (i * kW + j) + 1) * Ho + ho) * Wo + wo];
                float him = (float)(ho * sH - pH + i * dH) + oh;
                float wim = (float)(wo * sW - pW + j * dW) + ow;
                float v = 0;
                if (him > -1 && wim > -1 && him < H && wim < W)
                  v = bilinear_at(plane, (int)H, (int)W, him, wim);
                if (round_cols_f16) v = orc_round_f16(v);
                col[(cl * kH + i) * kW + j] = v;
              }
          }
          for (int64_t ol = 0; ol < Og; ol++) {
            const float* wr = weight + (g * Og + ol) * K;
            double acc = 0;
            for (int64_t k = 0; k < K; k++) acc += (double)wr[k] * (double)col[k];
            float r = (float)acc;
            if (relu && r < 0) r = 0;
            out[((b * O + g * Og + ol) * Ho + ho) * Wo + wo] = r;
          }
        }
    }
}

// ----- the reference's HALF instantiation of the same forward (val.py:126 `.half()`) ------------------
// scalar_t = c10::Half in deformable_im2col_gpu_kernel (deform_conv_cuda_kernel.cu:189-242) and
// deformable_im2col_bilinear (:83-114); the offsets arrive already cast to the input's dtype
// (models/dcn/deform_conv.py:45 `offset.type_as(input)`).  c10::Half arithmetic = convert both operands
// to float, operate, round the result to binary16 (an int operand is converted to Half first, exact for
// the magnitudes here) — so EVERY operation of the sampling is rounded to half:
//   h_im = rh(float(h_in + i*dil) + off_h);  lh = rh(h - h_low);  hh = rh(1 - lh);  w1 = rh(hh*hw) ...
//   val  = rh(rh(rh(rh(w1*v1) + rh(w2*v2)) + rh(w3*v3)) + rh(w4*v4))            (left to right, no FMA)
// The bounds tests and floor() run on the half value widened to float.  The contraction is cuBLAS
// half-in / float-accumulate (order unspecified): accumulated in double here, result rounded to half.
// x / weight must hold binary16-representable values.  Used only to BOUND the product's f16 kernel
// (which keeps coordinates in f32) against the reference's half semantics — tests/test_gpu_e2e.py.
static inline float rh(float v) { return orc_round_f16(v); }
static inline float bilinear_at_half(const float* plane, int H, int W, float h, float w) {
  int h_low = (int)std::floor(h), w_low = (int)std::floor(w);
  int h_high = h_low + 1, w_high = w_low + 1;
  float lh = rh(h - rh((float)h_low)), lw = rh(w - rh((float)w_low));
  float hh = rh(1.0f - lh), hw = rh(1.0f - lw);
  float v1 = 0, v2 = 0, v3 = 0, v4 = 0;
  if (h_low >= 0 && w_low >= 0) v1 = plane[h_low * W + w_low];
  if (h_low >= 0 && w_high <= W - 1) v2 = plane[h_low * W + w_high];
  if (h_high <= H - 1 && w_low >= 0) v3 = plane[h_high * W + w_low];
  if (h_high <= H - 1 && w_high <= W - 1) v4 = plane[h_high * W + w_high];
  float w1 = rh(hh * hw), w2 = rh(hh * lw), w3 = rh(lh * hw), w4 = rh(lh * lw);
  float acc = rh(rh(w1 * v1) + rh(w2 * v2));
  acc = rh(acc + rh(w3 * v3));
  acc = rh(acc + rh(w4 * v4));
  return acc;
}

void orc_deform_conv_forward_half(const float* x, const float* offset, const float* weight,
                                  int64_t B, int64_t C, int64_t H, int64_t W, int64_t O,
                                  int kH, int kW, int sH, int sW, int pH, int pW, int dH, int dW,
                                  int relu, float* out) {
  const int64_t Ho = (H + 2 * pH - (dH * (kH - 1) + 1)) / sH + 1;
  const int64_t Wo = (W + 2 * pW - (dW * (kW - 1) + 1)) / sW + 1;
  const int64_t K = C * kH * kW;
#pragma omp parallel for collapse(2) schedule(dynamic)
  for (int64_t b = 0; b < B; b++)
    for (int64_t ho = 0; ho < Ho; ho++) {
      std::vector<float> col(K);
      for (int64_t wo = 0; wo < Wo; wo++) {
        const float* offp = offset + b * 2 * kH * kW * Ho * Wo;
        for (int i = 0; i < kH; i++)
          for (int j = 0; j < kW; j++) {
            float oh = rh(offp[((2 * (i * kW + j)) * Ho + ho) * Wo + wo]);
            float ow = rh(offp[((2 * (i * kW + j) + 1) * Ho + ho) * Wo + wo]);
            float him = rh(rh((float)(ho * sH - pH + i * dH)) + oh);
            float wim = rh(rh((float)(wo * sW - pW + j * dW)) + ow);
            bool in = him > -1 && wim > -1 && him < H && wim < W;
            for (int64_t c = 0; c < C; c++)
              col[(c * kH + i) * kW + j] =
                  in ? bilinear_at_half(x + (b * C + c) * H * W, (int)H, (int)W, him, wim) : 0.0f;
          }
        for (int64_t o = 0; o < O; o++) {
          const float* wr = weight + o * K;
          double acc = 0;
          for (int64_t k = 0; k < K; k++) acc += (double)wr[k] * (double)col[k];
          float r = rh((float)acc);
          if (relu && r < 0) r = 0;
          out[((b * O + o) * Ho + ho) * Wo + wo] = r;
        }
      }
    }
}

// IEEE binary16 round-to-nearest-even of a float, returned as float.
float orc_round_f16(float v) {
  uint32_t u;
  std::memcpy(&u, &v, 4);
  uint32_t sign = u & 0x80000000u;
  uint32_t mag = u & 0x7fffffffu;
  float a;
  std::memcpy(&a, &mag, 4);
  if (mag >= 0x7f800000u) return v;              // inf / nan
  if (a >= 65520.0f) {                            // overflows to inf
    uint32_t inf = sign | 0x7f800000u;
    float r;
    std::memcpy(&r, &inf, 4);
    return r;
  }
  float r;
  if (a < 6.103515625e-05f) {                     // half subnormal range: quantum 2^-24
    r = std::nearbyint(a * 16777216.0f) / 16777216.0f;
  } else {
    int e;
    std::frexp(a, &e);                            // a = f * 2^e, f in [0.5,1)
    float q = std::ldexp(1.0f, e - 11);           // 10 fraction bits
    r = std::nearbyint(a / q) * q;
  }
  return sign ? -r : r;
}

}  // extern "C"
