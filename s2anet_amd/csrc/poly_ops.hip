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

#include <algorithm>
#include <cstring>

#include <rocprim/rocprim.hpp>

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

// ---------------------------------------------------------------- polygon NMS (chip merge)
// py_cpu_nms_poly_fast (DOTA_devkit/ResultMerge_multi_process.py:62-123): dets[n,9] = 8 polygon
// coordinates + score; greedy over descending score; a lower-scored j is dropped by a kept i unless
// NOT (iou <= thresh), where iou = polyiou when the axis-aligned boxes overlap (hbb_ovr > 0) and 0
// otherwise (:87-115).  Same device pipeline as the rotated NMS: radix sort, 64x64 upper-triangle
// tiles -> bitmask in HBM, on-device greedy scan, compaction — no host round trip of the mask.
struct PolyBox {
  double c[8];
  double x1, y1, x2, y2;   // hbb (:64-67); area = (x2-x1+1)*(y2-y1+1) (:69)
};

__device__ __forceinline__ unsigned long long dbl_sortable(double d) {
  unsigned long long u = (unsigned long long)__double_as_longlong(d);
  return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}

__global__ void k_poly_keys(const double* __restrict__ dets9, int64_t n, unsigned long long* __restrict__ key,
                            int32_t* __restrict__ idx) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  key[i] = ~dbl_sortable(dets9[9 * i + 8]);   // ascending radix sort == descending score
  idx[i] = (int32_t)i;
}

__global__ void k_poly_prep(const double* __restrict__ dets9, const int32_t* __restrict__ order, int64_t n,
                            PolyBox* __restrict__ sorted, uint32_t* __restrict__ seg_start,
                            uint32_t* __restrict__ num_seg, unsigned long long* __restrict__ mask_off,
                            uint32_t* __restrict__ nblk) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p == 0) {
    const unsigned long long nb = (unsigned long long)(n + 63) / 64;
    seg_start[0] = 0;
    seg_start[1] = (uint32_t)n;
    *num_seg = 1;
    mask_off[0] = 0;
    mask_off[1] = (unsigned long long)n * nb;
    nblk[0] = (uint32_t)nb;
  }
  if (p >= n) return;
  const double* d = dets9 + 9 * (int64_t)order[p];
  PolyBox b;
#pragma unroll
  for (int k = 0; k < 8; k++) b.c[k] = d[k];
  b.x1 = fmin(fmin(d[0], d[2]), fmin(d[4], d[6]));
  b.y1 = fmin(fmin(d[1], d[3]), fmin(d[5], d[7]));
  b.x2 = fmax(fmax(d[0], d[2]), fmax(d[4], d[6]));
  b.y2 = fmax(fmax(d[1], d[3]), fmax(d[5], d[7]));
  sorted[p] = b;
}

// Fast path (thresh >= 0): only pairs whose axis-aligned boxes overlap can suppress (hbb_ovr == 0 <= thresh
// otherwise, :87-115), and on DOTA-like data they are a tiny fraction of n^2/2.  k_poly_cull walks the upper
// triangle with the HBB test alone (inter > 0  <=>  hbb_ovr > 0: the union term is >= 1) and compacts the hits
// into a dense pair list (wave ballot + one atomic per hit group); k_poly_heavy then runs polyiou with every
// lane busy and ORs the suppression bits into the (zeroed) mask.  A full pair list falls back to k_poly_mask.
// a listed pair whose polygon IoU cannot exceed the threshold is not evaluated: for two strictly convex quadrilaterals the
// intersection lies inside the intersection of their axis-aligned boxes and the union holds the larger polygon, so
// iou <= hbb_inter / max(area); skipped only with a margin (1e-9 of the two areas, absolute and relative) far above the
// rounding of the double evaluation.  Anything else (non-convex or degenerate input, NaN) is evaluated.
__device__ __forceinline__ bool poly_pair_below(const PolyBox& A, const PolyBox& B, double thresh) {
  auto convex_area = [](const double* c, double& area) -> bool {
    double cr[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const double ax = c[2 * ((k + 1) & 3)] - c[2 * k], ay = c[2 * ((k + 1) & 3) + 1] - c[2 * k + 1];
      const double bx = c[2 * ((k + 2) & 3)] - c[2 * ((k + 1) & 3)], by = c[2 * ((k + 2) & 3) + 1] - c[2 * ((k + 1) & 3) + 1];
      cr[k] = ax * by - ay * bx;
    }
    area = 0.5 * fabs((c[0] * c[3] - c[2] * c[1]) + (c[2] * c[5] - c[4] * c[3]) + (c[4] * c[7] - c[6] * c[5]) + (c[6] * c[1] - c[0] * c[7]));
    return (cr[0] > 0 && cr[1] > 0 && cr[2] > 0 && cr[3] > 0) || (cr[0] < 0 && cr[1] < 0 && cr[2] < 0 && cr[3] < 0);
  };
  double aa, ab;
  if (!convex_area(A.c, aa) || !convex_area(B.c, ab)) return false;
  const double w = fmax(0.0, fmin(A.x2, B.x2) - fmax(A.x1, B.x1)), h = fmax(0.0, fmin(A.y2, B.y2) - fmax(A.y1, B.y1));
  const double eps = 1e-9 * (aa + ab), big = fmax(aa, ab);
  return w * h + eps < thresh * (big - eps) * (1.0 - 1e-9);
}

constexpr int kCullCols = 1024;       // columns (boxes j) a workgroup of k_poly_cull walks
__global__ __launch_bounds__(256) void k_poly_cull(const PolyBox* __restrict__ sorted, int64_t n,
                                                   uint2* __restrict__ pairs, unsigned long long* __restrict__ count,
                                                   unsigned long long cap, double thresh) {
  // 256 rows (one per thread) x 1 024 columns per workgroup.  (Round 5: one 64-thread workgroup per 64 x 64 tile and one atomic
  // per column with a hit -- half a million adds to ONE address at 20 000 boxes: 8.8 of the call's 12 ms -- and every test in
  // double arithmetic.)  Now: the boxes' axis-aligned bounds as FLOAT intervals rounded outwards (a conservative test on
  // one ds_read_b128 per column); a pair that passes it takes the exact double test (hbb_inter > 0, :87-93) and, in the list
  // form (thresh >= 0), the IoU upper bound poly_pair_below -- the exact pass then only sees pairs that can suppress; ONE
  // reservation in the pair list per workgroup.
  const int64_t r0 = (int64_t)blockIdx.y * 256, c0 = (int64_t)blockIdx.x * kCullCols;
  if (c0 + kCullCols - 1 <= r0) return;          // every column of the chunk is at or in front of every row: no j > i
  __shared__ float4 s_hbb[kCullCols];            // x1 (down), y1 (down), x2 (up), y2 (up)
  for (int k = threadIdx.x; k < kCullCols; k += 256) {
    const float qn = __builtin_nanf("");
    float4 v = make_float4(qn, qn, qn, qn);      // (beyond the last box: every comparison below is false)
    if (c0 + k < n) {
      const PolyBox& b = sorted[c0 + k];
      v = make_float4(__double2float_rd(b.x1), __double2float_rd(b.y1), __double2float_ru(b.x2), __double2float_ru(b.y2));
    }
    s_hbb[k] = v;
  }
  __syncthreads();
  const int64_t i = r0 + threadIdx.x;
  float fx1 = __builtin_nanf(""), fy1 = fx1, fx2 = fx1, fy2 = fx1;
  if (i < n) {
    const PolyBox& a = sorted[i];
    fx1 = __double2float_rd(a.x1); fy1 = __double2float_rd(a.y1); fx2 = __double2float_ru(a.x2); fy2 = __double2float_ru(a.y2);
  }
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t wave_row0 = r0 + (threadIdx.x & ~63);
  // pass 1: this row's surviving columns as bit masks (registers); pass 2 writes them behind ONE reservation per workgroup
  // (a reservation per wave and column block was still ~49 k adds to one address at 20 000 boxes: 0.65 ms of queueing)
  unsigned long long masks[kCullCols / 64];
  unsigned cnt = 0;
#pragma unroll
  for (int cb = 0; cb < kCullCols / 64; cb++) {
    const int64_t j0 = c0 + cb * 64;
    unsigned long long mine = 0;
    if (j0 < n && j0 + 63 > wave_row0) {         // (wave-uniform: inside the list, not wholly in front of this wave's rows)
#pragma unroll 8
      for (int c = 0; c < 64; c++) {
        const float4 hb = s_hbb[cb * 64 + c];
        // w * h > 0 in double needs min(x2) > max(x1) and min(y2) > max(y1): the outward-rounded floats keep every such pair
        if (fx2 > hb.x && hb.z > fx1 && fy2 > hb.y && hb.w > fy1 && j0 + c > i) mine |= 1ull << c;
      }
      if (mine) {                                // the exact test of the few float survivors (and the IoU bound of the list form)
        const PolyBox& A = sorted[i];
        unsigned long long keep = 0;
        for (unsigned long long m = mine; m; m &= m - 1) {
          const int c = __ffsll((long long)m) - 1;
          const PolyBox& B = sorted[j0 + c];
          const double w = fmax(0.0, fmin(A.x2, B.x2) - fmax(A.x1, B.x1));
          const double h = fmax(0.0, fmin(A.y2, B.y2) - fmax(A.y1, B.y1));
          if (w * h > 0 && !(thresh >= 0 && poly_pair_below(A, B, thresh))) keep |= 1ull << c;
        }
        mine = keep;
      }
    }
    masks[cb] = mine;
    cnt += (unsigned)__popcll(mine);
  }
  unsigned incl = cnt;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned t = (unsigned)__shfl_up((int)incl, o);
    if (lane >= (unsigned)o) incl += t;
  }
  __shared__ unsigned s_wtot[4];
  __shared__ unsigned long long s_base;
  if (lane == 63) s_wtot[wave] = incl;
  __syncthreads();
  const unsigned total = s_wtot[0] + s_wtot[1] + s_wtot[2] + s_wtot[3];
  if (total == 0) return;                        // (uniform)
  if (threadIdx.x == 0) s_base = atomicAdd(count, (unsigned long long)total);
  __syncthreads();
  unsigned before = incl - cnt;
  for (unsigned w2 = 0; w2 < wave; w2++) before += s_wtot[w2];
  unsigned long long slot = s_base + before;
#pragma unroll
  for (int cb = 0; cb < kCullCols / 64; cb++) {
    unsigned long long mine = masks[cb];
    const int64_t j0 = c0 + cb * 64;
    while (mine) {
      const int c = __ffsll((long long)mine) - 1;
      mine &= mine - 1;
      if (slot < cap) pairs[slot] = make_uint2((unsigned)i, (unsigned)(j0 + c));
      slot++;
    }
  }
}

// exact pass of the list form: polyiou of every listed pair; a pair that suppresses (overlap not <= thresh, :115) becomes an
// EDGE (i -> j, positions in descending-score order) for the greedy resolve by rounds (rotated_ops.hip) -- no N x N / 64
// mask, no serial scan
__global__ __launch_bounds__(kPolyThreads) void k_poly_edges(const PolyBox* __restrict__ sorted,
                                                             const uint2* __restrict__ pairs,
                                                             const unsigned long long* __restrict__ count,
                                                             unsigned long long cap, double thresh,
                                                             uint2* __restrict__ edges, unsigned long long* __restrict__ edge_count) {
  __shared__ D2 s_p[kPMax * kPolyThreads];
  __shared__ D2 s_t[kTmpMax * kPolyThreads];
  const unsigned long long total = *count < cap ? *count : cap;
  const unsigned lane = threadIdx.x & 63;
  const unsigned long long stride = (unsigned long long)gridDim.x * kPolyThreads;
  for (unsigned long long e0 = (unsigned long long)blockIdx.x * kPolyThreads; e0 < total; e0 += stride) {
    const unsigned long long e = e0 + threadIdx.x;
    bool hit = false;
    uint2 pr = make_uint2(0u, 0u);
    if (e < total) {
      pr = pairs[e];
      const double ovr = poly_iou(sorted[pr.x].c, sorted[pr.y].c, s_p + threadIdx.x, s_t + threadIdx.x);
      hit = !(ovr <= thresh);
    }
    const unsigned long long m = __ballot(hit);
    if (m == 0) continue;
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(edge_count, (unsigned long long)__popcll(m));
    base = ((unsigned long long)(uint32_t)__shfl((int)(base >> 32), 0) << 32) | (uint32_t)__shfl((int)(base & 0xffffffffu), 0);
    if (hit) edges[base + __popcll(m & ((1ull << lane) - 1ull))] = pr;      // (edges <= listed pairs <= cap: never overflows)
  }
}

__global__ __launch_bounds__(kPolyThreads) void k_poly_heavy(const PolyBox* __restrict__ sorted,
                                                             const uint2* __restrict__ pairs,
                                                             const unsigned long long* __restrict__ count,
                                                             unsigned long long cap, double thresh, uint32_t nb,
                                                             unsigned long long* __restrict__ mask) {
  __shared__ D2 s_p[kPMax * kPolyThreads];
  __shared__ D2 s_t[kTmpMax * kPolyThreads];
  const unsigned long long total = *count < cap ? *count : cap;
  for (unsigned long long e = (unsigned long long)blockIdx.x * kPolyThreads + threadIdx.x; e < total;
       e += (unsigned long long)gridDim.x * kPolyThreads) {
    const uint2 pr = pairs[e];
    const double ovr = poly_iou(sorted[pr.x].c, sorted[pr.y].c, s_p + threadIdx.x, s_t + threadIdx.x);
    if (!(ovr <= thresh))   // :115 keeps j only when the overlap is <= thresh
      atomicOr(mask + (unsigned long long)pr.x * nb + (pr.y >> 6), 1ull << (pr.y & 63));
  }
}

// voc_eval's inner search (DOTA_devkit/dota_evaluation_task1.py:204-263): for one detection, the ground-truth
// polygon of its image with the largest polyiou among those whose axis-aligned boxes overlap it (the "+ 1"
// pixel convention of :235-236); ovmax = -inf / index -1 when none does.  One thread per detection.
__global__ __launch_bounds__(kPolyThreads) void k_poly_match(const double* __restrict__ dets8,
                                                             const int32_t* __restrict__ det_img, int64_t D,
                                                             const double* __restrict__ gts8,
                                                             const int64_t* __restrict__ gt_off,
                                                             double* __restrict__ ovmax, int64_t* __restrict__ argmax) {
  __shared__ D2 s_p[kPMax * kPolyThreads];
  __shared__ D2 s_t[kTmpMax * kPolyThreads];
  const int64_t d = (int64_t)blockIdx.x * kPolyThreads + threadIdx.x;
  if (d >= D) return;
  const double* bb = dets8 + 8 * d;
  const double px1 = fmin(fmin(bb[0], bb[2]), fmin(bb[4], bb[6])), py1 = fmin(fmin(bb[1], bb[3]), fmin(bb[5], bb[7]));
  const double px2 = fmax(fmax(bb[0], bb[2]), fmax(bb[4], bb[6])), py2 = fmax(fmax(bb[1], bb[3]), fmax(bb[5], bb[7]));
  double best = -INFINITY;
  int64_t arg = -1;
  const int32_t img = det_img[d];
  for (int64_t g = gt_off[img]; g < gt_off[img + 1]; g++) {
    const double* gt = gts8 + 8 * g;
    const double gx1 = fmin(fmin(gt[0], gt[2]), fmin(gt[4], gt[6])), gy1 = fmin(fmin(gt[1], gt[3]), fmin(gt[5], gt[7]));
    const double gx2 = fmax(fmax(gt[0], gt[2]), fmax(gt[4], gt[6])), gy2 = fmax(fmax(gt[1], gt[3]), fmax(gt[5], gt[7]));
    const double iw = fmax(fmin(gx2, px2) - fmax(gx1, px1) + 1.0, 0.0);
    const double ih = fmax(fmin(gy2, py2) - fmax(gy1, py1) + 1.0, 0.0);
    const double inters = iw * ih;
    const double uni = (px2 - px1 + 1.0) * (py2 - py1 + 1.0) + (gx2 - gx1 + 1.0) * (gy2 - gy1 + 1.0) - inters;
    if (!(inters / uni > 0)) continue;                                       // :244
    const double ov = poly_iou(gt, bb, s_p + threadIdx.x, s_t + threadIdx.x);   // iou_poly(GT, bb), :250
    if (arg < 0 || ov > best) { best = ov; arg = g; }                        // np.max / first np.argmax
  }
  ovmax[d] = best;
  argmax[d] = arg;
}

#undef P
#undef T
#define P(i) p[(i) * 64]
#define T(i) tmp[(i) * 64]
// 64 threads = the 64 rows of a tile; each walks the 64 columns of the tile (direct form: every pair
// evaluated in place; used for thresh < 0 and when the pair list overflows: overflow_of != NULL makes the
// launch a no-op unless *overflow_of > cap)
__global__ __launch_bounds__(64) void k_poly_mask(const PolyBox* __restrict__ sorted, int64_t n, double thresh,
                                                  unsigned long long* __restrict__ mask,
                                                  const unsigned long long* __restrict__ overflow_of,
                                                  unsigned long long cap) {
  const uint32_t rb = blockIdx.y, cb = blockIdx.x;
  if (cb < rb) return;
  if (overflow_of && *overflow_of <= cap) return;
  __shared__ D2 s_p[kPMax * 64];
  __shared__ D2 s_t[kTmpMax * 64];
  __shared__ PolyBox s_col[64];
  const uint32_t nb = (uint32_t)((n + 63) / 64);
  const int64_t j0 = (int64_t)cb * 64;
  if (j0 + threadIdx.x < n) s_col[threadIdx.x] = sorted[j0 + threadIdx.x];
  __syncthreads();
  const int64_t i = (int64_t)rb * 64 + threadIdx.x;
  if (i >= n) return;
  const PolyBox A = sorted[i];
  const double areaA = (A.x2 - A.x1 + 1) * (A.y2 - A.y1 + 1);
  unsigned long long bits = 0;
  // the per-thread LDS polygons are strided by 64 here (block of 64 threads)
  D2* p = s_p + threadIdx.x;
  D2* tmp = s_t + threadIdx.x;
  for (int c = 0; c < 64; c++) {
    const int64_t j = j0 + c;
    if (j >= n || j <= i) continue;
    const PolyBox& B = s_col[c];
    double w = fmax(0.0, fmin(A.x2, B.x2) - fmax(A.x1, B.x1));
    double h = fmax(0.0, fmin(A.y2, B.y2) - fmax(A.y1, B.y1));
    double inter = w * h;
    double ovr = inter / (areaA + (B.x2 - B.x1 + 1) * (B.y2 - B.y1 + 1) - inter);
    if (ovr > 0) {
      // poly_iou with the 64-thread LDS stride
      D2 a[4], b[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        a[k] = {A.c[2 * k], A.c[2 * k + 1]};
        b[k] = {B.c[2 * k], B.c[2 * k + 1]};
      }
      if (quad_area(a) < 0) { D2 t = a[0]; a[0] = a[3]; a[3] = t; t = a[1]; a[1] = a[2]; a[2] = t; }
      if (quad_area(b) < 0) { D2 t = b[0]; b[0] = b[3]; b[3] = t; t = b[1]; b[1] = b[2]; b[2] = t; }
      double isum = 0;
#pragma unroll
      for (int ii = 0; ii < 4; ii++)
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
          // fan_overlap with stride-64 LDS polygons
          D2 pa = a[ii], pb = a[(ii + 1) & 3], pc = b[jj], pd = b[(jj + 1) & 3];
          D2 o{0, 0};
          int s1 = sgn(tri_cross(o, pa, pb)), s2 = sgn(tri_cross(o, pc, pd));
          if (s1 == 0 || s2 == 0) continue;
          if (s1 == -1) { D2 t = pa; pa = pb; pb = t; }
          if (s2 == -1) { D2 t = pc; pc = pd; pd = t; }
          P(0) = o; P(1) = pa; P(2) = pb;
          int np_ = 3;
          // three half-plane cuts (polygon_cut), stride 64
          D2 ca[3] = {o, pc, pd}, cbv[3] = {pc, pd, o};
          for (int cut = 0; cut < 3; cut++) {
            const D2 la = ca[cut], lb = cbv[cut];
            int m = 0;
            P(np_) = P(0);
            for (int q = 0; q < np_; q++) {
              D2 pi = P(q), pn = P(q + 1);
              int si = sgn(tri_cross(la, lb, pi));
              if (si > 0) { T(m) = pi; m++; }
              if (si != sgn(tri_cross(la, lb, pn))) {
                double c1 = tri_cross(la, lb, pi), c2 = tri_cross(la, lb, pn);
                D2 hit = T(m);
                if (!(sgn(c1) == 0 && sgn(c2) == 0) && sgn(c2 - c1) != 0) {
                  hit.x = (pi.x * c2 - pn.x * c1) / (c2 - c1);
                  hit.y = (pi.y * c2 - pn.y * c1) / (c2 - c1);
                }
                T(m) = hit; m++;
              }
            }
            np_ = 0;
            for (int q = 0; q < m; q++) {
              D2 tq = T(q);
              if (!q || !same_pt(tq, T(q - 1))) { P(np_) = tq; np_++; }
            }
            while (np_ > 1 && same_pt(P(np_ - 1), P(0))) np_--;
          }
          P(np_) = P(0);
          double r = 0;
          for (int q = 0; q < np_; q++) { D2 u = P(q), v = P(q + 1); r += u.x * v.y - u.y * v.x; }
          r = fabs(r / 2.0);
          if (s1 * s2 == -1) r = -r;
          isum += r;
        }
      double uni = fabs(quad_area(a)) + fabs(quad_area(b)) - isum;
      ovr = isum / uni;
    }
    if (!(ovr <= thresh)) bits |= 1ull << c;   // :115 keeps j only when hbb_ovr <= thresh
  }
  mask[(unsigned long long)i * nb + cb] = bits;
}
#undef P
#undef T

__global__ void k_poly_flags(const uint8_t* __restrict__ keep_orig, const int32_t* __restrict__ order, int64_t n,
                             uint8_t* __restrict__ flags) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) flags[p] = keep_orig[order[p]];
}

// ------------------------------------------------------------------ rotated NMS on DOUBLE boxes
// nms_rotated / ml_nms_rotated dispatch on the dtype of `dets` (AT_DISPATCH_FLOATING_TYPES_AND_HALF,
// utils/nms_rotated/src/nms_rotated_cuda.cu:95-100, utils/ml_nms_rotated/src/nms_rotated_cuda.cu:100-105): on float64
// boxes the reference evaluates single_box_iou_rotated<double>, and keep decisions next to the threshold differ from the
// float32 evaluation.  This is that instantiation -- box_iou_rotated_utils.h:56-375 with T = double, __CUDACC__ branch
// (swap sort), the label test of the ml copy (:319-322) -- as a plain per-thread evaluation: the f32 path's cull / list
// machinery (PreBox, tiles, pair lists) is built around 32-bit boxes, and float64 inputs are an API-completeness case,
// not a hot one.  Same structure as the reference's kernel (:14-72): 64 x 64 blocks of the upper triangle, one thread
// per row, suppression bits into an N x N/64 mask, then the greedy scan shared with the polygon NMS.
struct RBox6 {
  double x, y, w, h, a, label;
};

__device__ __forceinline__ double cr2(double ax, double ay, double bx, double by) { return ax * by - bx * ay; }
__device__ __forceinline__ double dt2(double ax, double ay, double bx, double by) { return ax * bx + ay * by; }

__device__ __forceinline__ void rbox_vertices_f64(double xc, double yc, double w, double h, double a, double* vx, double* vy) {
  const double c2 = cos(a) * 0.5f, s2 = sin(a) * 0.5f;
  vx[0] = xc - s2 * h - c2 * w;
  vy[0] = yc + c2 * h - s2 * w;
  vx[1] = xc + s2 * h - c2 * w;
  vy[1] = yc - c2 * h - s2 * w;
  vx[2] = 2 * xc - vx[0];
  vy[2] = 2 * yc - vy[0];
  vx[3] = 2 * xc - vx[1];
  vy[3] = 2 * yc - vy[1];
}

__device__ double rot_iou_f64(const RBox6& A, const RBox6& B) {
  if (A.label != B.label) return 0.0;
  const double sx = (A.x + B.x) / 2.0, sy = (A.y + B.y) / 2.0;
  const double area1 = A.w * A.h, area2 = B.w * B.h;
  if (area1 < 1e-14 || area2 < 1e-14) return 0.0;
  double ax[4], ay[4], bx[4], by[4];
  rbox_vertices_f64(A.x - sx, A.y - sy, A.w, A.h, A.a, ax, ay);
  rbox_vertices_f64(B.x - sx, B.y - sy, B.w, B.h, B.a, bx, by);
  double eax[4], eay[4], ebx[4], eby[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    eax[i] = ax[(i + 1) & 3] - ax[i];
    eay[i] = ay[(i + 1) & 3] - ay[i];
    ebx[i] = bx[(i + 1) & 3] - bx[i];
    eby[i] = by[(i + 1) & 3] - by[i];
  }
  double qx[24], qy[24], dist[24];
  int n = 0;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      const double det = cr2(ebx[j], eby[j], eax[i], eay[i]);
      if (fabs(det) <= 1e-14) continue;
      const double dx = bx[j] - ax[i], dy = by[j] - ay[i];
      const double t1 = cr2(ebx[j], eby[j], dx, dy) / det;
      const double t2 = cr2(eax[i], eay[i], dx, dy) / det;
      if (t1 >= 0.0f && t1 <= 1.0f && t2 >= 0.0f && t2 <= 1.0f) {
        qx[n] = ax[i] + eax[i] * t1;
        qy[n] = ay[i] + eay[i] * t1;
        n++;
      }
    }
  {
    const double abx = ebx[0], aby = eby[0], dax = ebx[3], day = eby[3];
    const double abab = dt2(abx, aby, abx, aby), adad = dt2(dax, day, dax, day);
    for (int i = 0; i < 4; i++) {
      const double apx = ax[i] - bx[0], apy = ay[i] - by[0];
      const double apab = dt2(apx, apy, abx, aby), apad = -dt2(apx, apy, dax, day);
      if (apab >= 0 && apad >= 0 && apab <= abab && apad <= adad) { qx[n] = ax[i]; qy[n] = ay[i]; n++; }
    }
  }
  {
    const double abx = eax[0], aby = eay[0], dax = eax[3], day = eay[3];
    const double abab = dt2(abx, aby, abx, aby), adad = dt2(dax, day, dax, day);
    for (int i = 0; i < 4; i++) {
      const double apx = bx[i] - ax[0], apy = by[i] - ay[0];
      const double apab = dt2(apx, apy, abx, aby), apad = -dt2(apx, apy, dax, day);
      if (apab >= 0 && apad >= 0 && apab <= abab && apad <= adad) { qx[n] = bx[i]; qy[n] = by[i]; n++; }
    }
  }
  double inter = 0.0;
  if (n > 2) {
    int t = 0;
    for (int i = 1; i < n; i++)
      if (qy[i] < qy[t] || (qy[i] == qy[t] && qx[i] < qx[t])) t = i;
    const double ox = qx[t], oy = qy[t];
    for (int i = 0; i < n; i++) { qx[i] -= ox; qy[i] -= oy; }
    { const double tx = qx[0], ty = qy[0]; qx[0] = qx[t]; qy[0] = qy[t]; qx[t] = tx; qy[t] = ty; }
    for (int i = 0; i < n; i++) dist[i] = dt2(qx[i], qy[i], qx[i], qy[i]);
    for (int i = 1; i < n - 1; i++)
      for (int j = i + 1; j < n; j++) {
        const double cp = cr2(qx[i], qy[i], qx[j], qy[j]);
        if ((cp < -1e-6) || (fabs(cp) < 1e-6 && dist[i] > dist[j])) {
          double s_;
          s_ = qx[i]; qx[i] = qx[j]; qx[j] = s_;
          s_ = qy[i]; qy[i] = qy[j]; qy[j] = s_;
          s_ = dist[i]; dist[i] = dist[j]; dist[j] = s_;
        }
      }
    int k = 1;
    for (; k < n; k++)
      if (dist[k] > 1e-8) break;
    if (k < n) {
      qx[1] = qx[k]; qy[1] = qy[k];
      int m = 2;
      for (int i = k + 1; i < n; i++) {
        while (m > 1 && cr2(qx[i] - qx[m - 2], qy[i] - qy[m - 2], qx[m - 1] - qx[m - 2], qy[m - 1] - qy[m - 2]) >= 0) m--;
        qx[m] = qx[i]; qy[m] = qy[i];
        m++;
      }
      if (m > 2) {
        double area = 0;
        for (int i = 1; i < m - 1; i++)
          area += fabs(cr2(qx[i] - qx[0], qy[i] - qy[0], qx[i + 1] - qx[0], qy[i + 1] - qy[0]));
        inter = area / 2.0;
      }
    }
  }
  return inter / (area1 + area2 - inter);
}

__global__ void k_rot64_keys(const double* __restrict__ scores, int64_t n, unsigned long long* __restrict__ key,
                             int32_t* __restrict__ idx) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  key[i] = ~dbl_sortable(scores[i]);          // ascending stable radix sort == descending score, ties by ascending index
  idx[i] = (int32_t)i;
}

__global__ void k_rot64_prep(const double* __restrict__ dets5, const double* __restrict__ labels,
                             const int32_t* __restrict__ order, int64_t n, RBox6* __restrict__ sorted,
                             uint32_t* __restrict__ seg_start, uint32_t* __restrict__ num_seg,
                             unsigned long long* __restrict__ mask_off, uint32_t* __restrict__ nblk) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p == 0) {
    const unsigned long long nb = (unsigned long long)(n + 63) / 64;
    seg_start[0] = 0;
    seg_start[1] = (uint32_t)n;
    *num_seg = 1;
    mask_off[0] = 0;
    mask_off[1] = (unsigned long long)n * nb;
    nblk[0] = (uint32_t)nb;
  }
  if (p >= n) return;
  const int64_t o = order[p];
  const double* d = dets5 + 5 * o;
  double l = labels ? labels[o] : 0.0;
  if (l == 0.0) l = 0.0;                       // -0 == +0 in the reference's compare
  sorted[p] = RBox6{d[0], d[1], d[2], d[3], d[4], l};
}

__global__ __launch_bounds__(64) void k_rot64_mask(const RBox6* __restrict__ sorted, int64_t n, float thr,
                                                   unsigned long long* __restrict__ mask) {
  const uint32_t rb = blockIdx.y, cb = blockIdx.x;
  if (cb < rb) return;
  __shared__ RBox6 s_col[64];
  const uint32_t nb = (uint32_t)((n + 63) / 64);
  const int64_t j0 = (int64_t)cb * 64;
  if (j0 + threadIdx.x < n) s_col[threadIdx.x] = sorted[j0 + threadIdx.x];
  __syncthreads();
  const int64_t i = (int64_t)rb * 64 + threadIdx.x;
  if (i >= n) return;
  const RBox6 A = sorted[i];
  unsigned long long bits = 0;
  for (int c = 0; c < 64; c++) {
    const int64_t j = j0 + c;
    if (j >= n || j <= i) continue;
    if (rot_iou_f64(A, s_col[c]) > thr) bits |= 1ull << c;       // (cuda.cu:63-64: double > float)
  }
  mask[(unsigned long long)i * nb + cb] = bits;
}

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


extern "C" int s2a_polyiou_match(const double* dets8, const int32_t* det_image, int64_t num_dets, const double* gts8,
                                 const int64_t* gt_offsets, int64_t num_images, double* ovmax, int64_t* argmax,
                                 s2a_stream_t stream) {
  S2A_CHECK_ARG(num_dets >= 0 && num_images >= 0, "polyiou_match: negative size");
  if (num_dets == 0) return S2A_OK;
  S2A_CHECK_ARG(dets8 && det_image && gt_offsets && ovmax && argmax && num_images > 0, "polyiou_match: NULL tensor");
  k_poly_match<<<(unsigned)((num_dets + kPolyThreads - 1) / kPolyThreads), kPolyThreads, 0, as_stream(stream)>>>(
      dets8, det_image, num_dets, gts8, gt_offsets, ovmax, argmax);
  S2A_LAUNCH_CHECK();
  return S2A_OK;
}

static size_t poly_pair_cap(int64_t n) {
  const size_t all = (size_t)n * (size_t)(n > 0 ? n - 1 : 0) / 2;
  return std::min(all, std::max<size_t>((size_t)n * 32, (size_t)1 << 20)) + 1;
}

extern "C" size_t s2a_nms_poly_workspace_bytes(int64_t n) {
  if (n <= 0) return 256;
  size_t sz = (size_t)n, nb = (sz + 63) / 64;
  return align_up(sz * 8) * 2 + align_up(sz * 4) * 2 + align_up(sz * sizeof(PolyBox)) + align_up(sz * nb * 8) +
         align_up(sz) * 2 + align_up(sz * 40 + (8u << 20)) + align_up(poly_pair_cap(n) * 8) * 3 +
         nms_edge_rounds_workspace(n) + align_up(keep_compact_scratch_words() * 4) + 8192;
}

extern "C" int s2a_nms_poly(const double* dets9, int64_t n, double thresh, int64_t* keep, int64_t* count_dev,
                            int64_t* host_count, void* workspace, size_t workspace_bytes, s2a_stream_t stream) {
  S2A_CHECK_ARG(n >= 0 && n < (1ll << 31), "nms_poly: n out of range");
  S2A_CHECK_ARG(count_dev != nullptr, "nms_poly: count_dev must not be NULL");
  hipStream_t st = as_stream(stream);
  S2A_REFUSE_CAPTURE(st, "nms_poly");
  if (n == 0) {
    S2A_HIP(hipMemsetAsync(count_dev, 0, sizeof(int64_t), st));
    if (host_count) *host_count = 0;
    return S2A_OK;
  }
  S2A_CHECK_ARG(dets9 && keep, "nms_poly: NULL tensor");
  const size_t sz = (size_t)n, nb = (sz + 63) / 64;
  S2A_CHECK_ARG(nb <= 65535, "nms_poly: more than 4.19 M boxes is not supported");
  Carver cv(workspace, workspace_bytes);
  auto* key_a = cv.take<unsigned long long>(sz);
  auto* key_b = cv.take<unsigned long long>(sz);
  auto* idx_a = cv.take<int32_t>(sz);
  auto* order = cv.take<int32_t>(sz);
  auto* sorted = cv.take<PolyBox>(sz);
  auto* mask = cv.take<unsigned long long>(sz * nb);
  auto* keep_orig = cv.take<uint8_t>(sz);
  auto* flags = cv.take<uint8_t>(sz);
  auto* small = cv.take<unsigned long long>(64);   // seg_start[2] | num_seg | nblk | mask_off[2] | status
  size_t rpb = sz * 40 + (8u << 20);
  void* rp = cv.take<char>(rpb);
  const size_t cap = poly_pair_cap(n);
  auto* pairs = cv.take<uint2>(cap);
  auto* edges = cv.take<uint2>(cap);
  auto* alive = cv.take<uint2>(cap);
  const size_t rounds_bytes = nms_edge_rounds_workspace(n);
  void* rounds_ws = cv.take<char>(rounds_bytes);
  auto* cnt_scratch = cv.take<uint32_t>(keep_compact_scratch_words());
  if (!rp || !small || !pairs || !edges || !alive || !rounds_ws || !cnt_scratch || cv.off > workspace_bytes) {
    set_error("nms_poly: workspace too small (%zu < %zu)", workspace_bytes, cv.off);
    return S2A_EWORKSPACE;
  }
  uint32_t* seg_start = reinterpret_cast<uint32_t*>(small);
  uint32_t* num_seg = seg_start + 4;
  uint32_t* nblk = seg_start + 6;
  uint32_t* status = seg_start + 8;
  unsigned long long* mask_off = small + 8;
  const unsigned g = (unsigned)((n + 255) / 256);
  S2A_HIP(hipMemsetAsync(small, 0, 64 * 8, st));
  S2A_HIP(hipMemsetAsync(keep_orig, 0, sz, st));
  k_poly_keys<<<g, 256, 0, st>>>(dets9, n, key_a, idx_a);
  size_t need = 0;
  S2A_HIP(rocprim::radix_sort_pairs(nullptr, need, key_a, key_b, idx_a, order, sz, 0, 64, st));
  S2A_CHECK_ARG(need <= rpb, "nms_poly: sort scratch too small");
  S2A_HIP(rocprim::radix_sort_pairs(rp, need, key_a, key_b, idx_a, order, sz, 0, 64, st));
  k_poly_prep<<<g, 256, 0, st>>>(dets9, order, n, sorted, seg_start, num_seg, mask_off, nblk);
  dim3 grid((unsigned)nb, (unsigned)nb);
  // LIST form (round 6; a synchronous caller, thresh >= 0): HBB pair list -> polyiou on the listed pairs -> suppression EDGES
  // -> the greedy resolve by rounds the rotated NMS uses -> compaction.  No N x N / 64 mask, no serial scan (2.1 ms at
  // 20 000 boxes), no library select.  The host reads the pair count with the keep count: a pair list that overflowed (dense
  // scenes) sends the call through the mask form below.  S2A_POLY_NMS_LIST=0: the mask form always (A/B, tests)
  const char* e_list = getenv("S2A_POLY_NMS_LIST");
  if (thresh >= 0 && host_count && !(e_list && e_list[0] == '0')) {
    unsigned long long* pair_count = small + 16;
    unsigned long long* edge_count = small + 17;
    k_poly_cull<<<dim3((unsigned)((n + kCullCols - 1) / kCullCols), (unsigned)((n + 255) / 256)), 256, 0, st>>>(sorted, n, pairs, pair_count, (unsigned long long)cap, thresh);
    const unsigned hb = (unsigned)std::min<size_t>((cap + kPolyThreads - 1) / kPolyThreads, 256 * 16);
    k_poly_edges<<<hb, kPolyThreads, 0, st>>>(sorted, pairs, pair_count, (unsigned long long)cap, thresh, edges, edge_count);
    S2A_LAUNCH_CHECK();
    int rcl = launch_nms_edge_rounds(edges, (unsigned long long)cap, edge_count, alive, (unsigned long long)cap, n, order, keep_orig,
                                     rounds_ws, rounds_bytes, st);
    if (rcl != S2A_OK) return rcl;
    rcl = launch_keep_compact(keep_orig, order, n, cnt_scratch, keep, count_dev, st);
    if (rcl != S2A_OK) return rcl;
    unsigned long long host_pairs = 0;
    S2A_HIP(hipMemcpyAsync(&host_pairs, pair_count, sizeof(host_pairs), hipMemcpyDeviceToHost, st));
    S2A_HIP(hipMemcpyAsync(host_count, count_dev, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    S2A_HIP(hipStreamSynchronize(st));
    if (host_pairs <= (unsigned long long)cap) return S2A_OK;
    // the pair list overflowed: the mask form settles the call (its own zero-fills first)
    S2A_HIP(hipMemsetAsync(small + 16, 0, 16, st));
    S2A_HIP(hipMemsetAsync(keep_orig, 0, sz, st));
  }
  if (thresh >= 0) {
    unsigned long long* pair_count = small + 16;
    S2A_HIP(hipMemsetAsync(mask, 0, sz * nb * 8, st));
    k_poly_cull<<<dim3((unsigned)((n + kCullCols - 1) / kCullCols), (unsigned)((n + 255) / 256)), 256, 0, st>>>(sorted, n, pairs, pair_count, (unsigned long long)cap, -1.0);
    const unsigned hb = (unsigned)std::min<size_t>((cap + kPolyThreads - 1) / kPolyThreads, 256 * 16);
    k_poly_heavy<<<hb, kPolyThreads, 0, st>>>(sorted, pairs, pair_count, (unsigned long long)cap, thresh, (uint32_t)nb, mask);
    k_poly_mask<<<grid, 64, 0, st>>>(sorted, n, thresh, mask, pair_count, (unsigned long long)cap);   // overflow only
  } else {
    k_poly_mask<<<grid, 64, 0, st>>>(sorted, n, thresh, mask, nullptr, 0);
  }
  S2A_LAUNCH_CHECK();
  int rc = launch_nms_scan(mask, seg_start, num_seg, mask_off, nblk, order, keep_orig, (uint32_t)nb, mask_off + 1,
                           (unsigned long long)sz * nb, status, st);
  if (rc != S2A_OK) return rc;
  k_poly_flags<<<g, 256, 0, st>>>(keep_orig, order, n, flags);
  need = 0;
  S2A_HIP(rocprim::select(nullptr, need, order, flags, keep, count_dev, sz, st));
  S2A_CHECK_ARG(need <= rpb, "nms_poly: select scratch too small");
  S2A_HIP(rocprim::select(rp, need, order, flags, keep, count_dev, sz, st));
  S2A_LAUNCH_CHECK();
  if (host_count) {
    S2A_HIP(hipMemcpyAsync(host_count, count_dev, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    S2A_HIP(hipStreamSynchronize(st));
  }
  return S2A_OK;
}

extern "C" size_t s2a_nms_rotated_f64_workspace_bytes(int64_t n) {
  if (n <= 0) return 256;
  size_t sz = (size_t)n, nb = (sz + 63) / 64;
  return align_up(sz * 8) * 2 + align_up(sz * 4) * 2 + align_up(sz * sizeof(RBox6)) + align_up(sz * nb * 8) +
         align_up(sz) * 2 + align_up(sz * 40 + (8u << 20)) + 8192;
}

extern "C" int s2a_nms_rotated_f64(const double* dets5, const double* scores, const double* labels, int64_t n,
                                   float iou_threshold, int64_t* keep, int64_t* count_dev, int64_t* host_count,
                                   void* workspace, size_t workspace_bytes, s2a_stream_t stream) {
  S2A_CHECK_ARG(n >= 0 && n < (1ll << 31), "nms_rotated_f64: n out of range");
  S2A_CHECK_ARG(count_dev != nullptr, "nms_rotated_f64: count_dev must not be NULL");
  hipStream_t st = as_stream(stream);
  S2A_REFUSE_CAPTURE(st, "nms_rotated_f64");
  if (n == 0) {
    S2A_HIP(hipMemsetAsync(count_dev, 0, sizeof(int64_t), st));
    if (host_count) *host_count = 0;
    return S2A_OK;
  }
  S2A_CHECK_ARG(dets5 && scores && keep, "nms_rotated_f64: NULL tensor");
  const size_t sz = (size_t)n, nb = (sz + 63) / 64;
  S2A_CHECK_ARG(nb <= 65535 && sz * nb * 8 < (8ull << 30), "nms_rotated_f64: more than 260 k boxes is not supported (N x N/64 mask)");
  Carver cv(workspace, workspace_bytes);
  auto* key_a = cv.take<unsigned long long>(sz);
  auto* key_b = cv.take<unsigned long long>(sz);
  auto* idx_a = cv.take<int32_t>(sz);
  auto* order = cv.take<int32_t>(sz);
  auto* sorted = cv.take<RBox6>(sz);
  auto* mask = cv.take<unsigned long long>(sz * nb);
  auto* keep_orig = cv.take<uint8_t>(sz);
  auto* flags = cv.take<uint8_t>(sz);
  auto* small = cv.take<unsigned long long>(64);
  size_t rpb = sz * 40 + (8u << 20);
  void* rp = cv.take<char>(rpb);
  if (!rp || !small || cv.off > workspace_bytes) {
    set_error("nms_rotated_f64: workspace too small (%zu < %zu)", workspace_bytes, cv.off);
    return S2A_EWORKSPACE;
  }
  uint32_t* seg_start = reinterpret_cast<uint32_t*>(small);
  uint32_t* num_seg = seg_start + 4;
  uint32_t* nblk = seg_start + 6;
  uint32_t* status = seg_start + 8;
  unsigned long long* mask_off = small + 8;
  const unsigned g = (unsigned)((n + 255) / 256);
  S2A_HIP(hipMemsetAsync(small, 0, 64 * 8, st));
  S2A_HIP(hipMemsetAsync(keep_orig, 0, sz, st));
  k_rot64_keys<<<g, 256, 0, st>>>(scores, n, key_a, idx_a);
  size_t need = 0;
  S2A_HIP(rocprim::radix_sort_pairs(nullptr, need, key_a, key_b, idx_a, order, sz, 0, 64, st));
  S2A_CHECK_ARG(need <= rpb, "nms_rotated_f64: sort scratch too small");
  S2A_HIP(rocprim::radix_sort_pairs(rp, need, key_a, key_b, idx_a, order, sz, 0, 64, st));
  k_rot64_prep<<<g, 256, 0, st>>>(dets5, labels, order, n, sorted, seg_start, num_seg, mask_off, nblk);
  k_rot64_mask<<<dim3((unsigned)nb, (unsigned)nb), 64, 0, st>>>(sorted, n, iou_threshold, mask);
  S2A_LAUNCH_CHECK();
  int rc = launch_nms_scan(mask, seg_start, num_seg, mask_off, nblk, order, keep_orig, (uint32_t)nb, mask_off + 1,
                           (unsigned long long)sz * nb, status, st);
  if (rc != S2A_OK) return rc;
  k_poly_flags<<<g, 256, 0, st>>>(keep_orig, order, n, flags);
  need = 0;
  S2A_HIP(rocprim::select(nullptr, need, order, flags, keep, count_dev, sz, st));
  S2A_CHECK_ARG(need <= rpb, "nms_rotated_f64: select scratch too small");
  S2A_HIP(rocprim::select(rp, need, order, flags, keep, count_dev, sz, st));
  S2A_LAUNCH_CHECK();
  if (host_count) {
    S2A_HIP(hipMemcpyAsync(host_count, count_dev, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    S2A_HIP(hipStreamSynchronize(st));
  }
  return S2A_OK;
}
