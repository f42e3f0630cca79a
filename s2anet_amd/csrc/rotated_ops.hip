// Rotated-box IoU and rotated (multi-label) NMS for MI355X (gfx950, wave64).
// COMPILE WITH -ffp-contract=off (see rbox_geom.hpp).
//
// Replaces the reference's three CUDA extensions:
//   utils/box_iou_rotated/src/box_iou_rotated_cuda.cu:14-101
//   utils/nms_rotated/src/nms_rotated_cuda.cu:14-133
//   utils/ml_nms_rotated/src/nms_rotated_cuda.cu:14-137
//
// MI355X-first structure (not the reference's one-thread-per-pair tiles):
//   1. per-box pre-pass: double cos/sin once per box (PreBox), not once per pair
//   2. CULL pass: every pair gets a 10-instruction circumscribed-circle test that is
//      provably exact (culled => the reference returns exactly 0.0f).  Surviving pairs
//      (~1 % of random DOTA-like boxes) are compacted — LDS queue per workgroup, one
//      global atomic per flush — into a dense pair list
//   3. HEAVY pass: the divergent, register-hungry clipping/hull code runs on the dense
//      list with all 64 lanes busy; its <=24 candidate points sit in LDS ([point][thread]
//      float2, conflict-free) instead of scratch
//   4. NMS only: segments (= label runs, or image*class for the batched detector) are
//      processed independently on upper-triangle 64x64 tiles; the suppression bitmask
//      stays in HBM, is scanned ON DEVICE (one workgroup per segment, wave-level
//      readlane resolve of each diagonal word) and the keep list is compacted on device
//      in score order.  No device->host mask copy (reference cuda.cu:109), no host scan
//      (:120-131).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "common.hpp"
#include "rbox_geom.hpp"

namespace s2a {
namespace {

constexpr int kThreads = 256;
constexpr int kLdsQueue = 1024;      // entries per workgroup queue; a dense-stage batch adds at most 256
constexpr int kFlushAt = kLdsQueue - 256;
constexpr int kPersistentGrid = 1024;
constexpr int kIouCap = 8;               // candidate-point slots per lane of the fast exact-IoU passes (rbox_iou)
constexpr uint32_t kIouRedo = 0x7fc5a5a5u;   // marker of a pair that needs all 24 slots
constexpr int kHeavyGrid = 2048;         // 8 workgroups per CU
constexpr unsigned kScanCacheWords = 6144;  // 48 KB of suppression mask cached in LDS per segment

__device__ __forceinline__ uint32_t float_sortable(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // ascending order of floats
}
__device__ __forceinline__ float sortable_float(uint32_t u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// ---------------------------------------------------------------- pre-pass
// both box sets of box_iou_rotated in one launch; the first threads also reset the per-chunk pair counters
__global__ void k_prep_boxes2(const float* __restrict__ b1, int64_t n, PreBox* __restrict__ o1,
                              const float* __restrict__ b2, int64_t m, PreBox* __restrict__ o2,
                              unsigned long long* __restrict__ counters, int ncounters) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < ncounters) counters[i] = 0ull;
  if (i < n) {
    const float* b = b1 + 5 * i;
    o1[i] = make_prebox(b[0], b[1], b[2], b[3], b[4], 0.f);
  } else if (i < n + m) {
    const float* b = b2 + 5 * (i - n);
    o2[i - n] = make_prebox(b[0], b[1], b[2], b[3], b[4], 0.f);
  }
}

__global__ void k_prep_boxes(const float* __restrict__ boxes5, int64_t n, PreBox* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* b = boxes5 + 5 * i;
  out[i] = make_prebox(b[0], b[1], b[2], b[3], b[4], 0.f);
}

// ---------------------------------------------------------------- workgroup pair queue
struct PairQueue {
  uint2* q;            // LDS, kLdsQueue entries
  unsigned* count;     // LDS
  unsigned* base;      // LDS (flush broadcast)
};

__device__ __forceinline__ void queue_push(PairQueue& Q, unsigned i, unsigned j) {
  unsigned p = atomicAdd(Q.count, 1u);
  Q.q[p] = make_uint2(i, j);
}

// all threads of the workgroup call this; copies the LDS queue to the global list
__device__ __forceinline__ void queue_flush(PairQueue& Q, uint2* __restrict__ gq,
                                            unsigned long long* __restrict__ gcount,
                                            unsigned long long cap) {
  __syncthreads();
  unsigned cnt = *Q.count;
  if (cnt == 0) return;  // uniform
  if (threadIdx.x == 0) {
    unsigned long long b = atomicAdd(gcount, (unsigned long long)cnt);
    Q.base[0] = (unsigned)(b & 0xffffffffu);
    Q.base[1] = (unsigned)(b >> 32);
  }
  __syncthreads();
  unsigned long long b = ((unsigned long long)Q.base[1] << 32) | Q.base[0];
  for (unsigned e = threadIdx.x; e < cnt; e += blockDim.x) {
    unsigned long long dst = b + e;
    if (dst < cap) gq[dst] = Q.q[e];  // beyond cap: dropped, gcount keeps the true total
  }
  __syncthreads();
  if (threadIdx.x == 0) *Q.count = 0;
  __syncthreads();
}

// ================================================================= box_iou_rotated
// CULL: workgroup = 1024 columns (4 per lane: one 16-byte store per lane per row) x up to 64 rows.
// Every element of the tile is stored here as 0.0f -- exact for culled pairs; the ~1 % that survive
// the cull are also queued and overwritten by the heavy pass (same stream, later kernel).  Row boxes
// are staged in LDS and read one iteration ahead so no row waits on a load.  HBM-write-bound, so the
// kernel keeps its LDS small (18 KB: eight workgroups = 32 waves per CU; the first version staged the column
// boxes as well, 76 KB = two workgroups per CU, and reached a quarter of the write bandwidth).
// Queue discipline: every decision that all threads must take alike is taken through __syncthreads_or -- a
// plain read of an LDS counter after a barrier is NOT uniform (a fast wave may already be pushing again).
constexpr int kIouRowsPerWg = 64;
constexpr int kIouQueue = 1024;      // LDS pair queue (survivors of BOTH tests); a drain batch adds <= 256
constexpr int kIouQ1 = 3072;         // first-stage queue: circle-test survivors, (row << 10 | column) per entry
constexpr int kIouQ1DrainAt = 1024;  // <= 2048 pushes per two rows on top of this

template <bool STORE>
__global__ __launch_bounds__(kThreads) void k_iou_cull(const PreBox* __restrict__ P1,
                                                       const PreBox* __restrict__ P2,
                                                       int64_t row0, int64_t row1, int64_t m,
                                                       float* __restrict__ out,
                                                       uint2* __restrict__ gq,
                                                       unsigned long long* __restrict__ gcount,
                                                       unsigned long long cap) {
  __shared__ uint2 s_q[kIouQueue];
  __shared__ unsigned s_count, s_base[2], s_cnt1;
  __shared__ PreBox s_rows[kIouRowsPerWg];
  __shared__ unsigned short s_q1[kIouQ1];
  PairQueue Q{s_q, &s_count, s_base};
  if (threadIdx.x == 0) { s_count = 0; s_cnt1 = 0; }

  const int64_t jw = (int64_t)blockIdx.x * kThreads * 4;          // first column of the workgroup
  const int64_t j0 = jw + threadIdx.x * 4;
  const int64_t rbeg = row0 + (int64_t)blockIdx.y * kIouRowsPerWg;
  const int nrows = (int)min((int64_t)kIouRowsPerWg, row1 - rbeg);
  if (threadIdx.x < nrows) s_rows[threadIdx.x] = P1[rbeg + threadIdx.x];
  float bx[4], by[4], br[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    PreBox b = {};
    if (j0 + k < m) b = P2[j0 + k];
    bx[k] = b.x; by[k] = b.y; br[k] = b.r;
  }
  const bool vec_ok = (j0 + 3 < m) && ((m & 3) == 0);   // 16-byte aligned full group
  __syncthreads();

  // Second stage, dense: every lane takes one circle-test survivor from the LDS queue and runs the
  // separating-axis test on it (a wave with ANY surviving lane used to run the SAT for all its lanes: 86 % of
  // the iterations at a 3 % survival rate).  Column boxes come from global memory here (L2-resident, ~3 % of the
  // pairs).  Called by all threads, after a barrier that made the first-stage pushes visible.
  auto drain_q1 = [&]() {
    const unsigned cnt1 = s_cnt1;                          // stable: nobody pushes to s_q1 during a drain
    for (unsigned e0 = 0; e0 < cnt1; e0 += kThreads) {     // uniform trip count
      if (__syncthreads_or(s_count > kIouQueue - kThreads)) queue_flush(Q, gq, gcount, cap);
      const unsigned e = e0 + threadIdx.x;
      if (e < cnt1) {
        const unsigned v = s_q1[e], r = v >> 10, jl = v & 1023u;
        const PreBox B = P2[jw + jl];
        if (!sat_disjoint(s_rows[r], B)) queue_push(Q, (unsigned)(rbeg + r - row0), (unsigned)(jw + jl));
      }
    }
    __syncthreads();                                       // all entries consumed
    if (threadIdx.x == 0) s_cnt1 = 0;
    __syncthreads();
  };

  PreBox Ai = s_rows[0];
  for (int r = 0; r < nrows; r++) {
    const float ax = Ai.x, ay = Ai.y, ar = Ai.r;
    if (r + 1 < nrows) Ai = s_rows[r + 1];   // next row's box while this one is processed
    const int64_t i = rbeg + r;
    (void)i; (void)vec_ok;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (j0 + k < m && !surely_disjoint(ax, ay, ar, bx[k], by[k], br[k])) {
        const unsigned p = atomicAdd(&s_cnt1, 1u);
        s_q1[p] = (unsigned short)((r << 10) | (threadIdx.x * 4 + k));
      }
    }
    if (STORE) {
      float* dst = out + i * m + j0;
      if (vec_ok) {
        *reinterpret_cast<float4*>(dst) = make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
#pragma unroll
        for (int k = 0; k < 4; k++)
          if (j0 + k < m) dst[k] = 0.f;
      }
    }
    if ((r & 1) == 1) {          // <= 2048 pushes per two rows: the first-stage queue cannot overflow
      if (__syncthreads_or(s_cnt1 > kIouQ1DrainAt)) drain_q1();
    }
  }
  __syncthreads();
  drain_q1();
  queue_flush(Q, gq, gcount, cap);
}

// ---- pair finding beside a forked zero-fill (no stores here), with NO LDS traffic in its first stage.  A wave owns 64
// rows (lane = row) and walks its columns 64 at a time: lane j holds column j's box of the chunk in registers, v_readlane
// hands column j's circle to all lanes as scalar operands (j is a compile-time constant in the unrolled loop), the
// verdicts collect in a 64-bit mask per lane.  Second stage without a list: every lane walks its own mask bits and fetches
// the column box from the owning lane by ds_bpermute (the crossbar, not the banks); the loop runs max-popcount times
// (5-6 at DOTA-like densities).  Survivors of both tests are staged per wave, one global atomic per workgroup at the end.
// No barrier in the loop.  Circle test: (ax-bx)^2 + (ay-by)^2 > (1.002 ar + 1e-3 + 1.002 br)^2 -- surely_disjoint's
// margin with the two factors hoisted into the row and the column; NaNs compare false and are evaluated.
// (Forms that were built, tested bit-exact, measured and removed again -- DESIGN.md section 4 has the numbers: circle
// records as LDS broadcast reads with per-wave survivor lists (59 us alone became 79), the same on half the LDS, a dense
// second stage fed from a list, a grid-binned finder, and one launch per 256 x 256 tile that also fills and evaluates.)
constexpr int kCrStage = 256;                 // per-wave staged pairs
__device__ __forceinline__ float lane_bcast(float v, int j) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), j));
}
// Hand-off of LDS data between the lanes of ONE wave (no workgroup barrier): the hardware issues a wave's LDS
// operations in order, so all that is needed is that the COMPILER keeps the order too.  Wavefront-scope fences + the wave
// barrier emit no instruction; they pin the order of the LDS stores / atomics before against the LDS loads after.
__device__ __forceinline__ void wave_lds_handoff() {
#ifdef S2A_ABL_NOFENCE
  return;
#endif
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <int J0, int J1>
__device__ __forceinline__ void circle_bits(float ax, float ay, float ar, float cx, float cy, float cr, unsigned& bits) {
  // two columns per packed-f32 instruction (v_pk_add / v_pk_mul / v_pk_fma_f32; the broadcast column pair is one scalar-pair
  // operand): 8.5 instead of 12.5 instructions per column.  The test only has to be conservative (a pair with IoU > 0 has
  // circles that overlap by the 0.2 % margin), so the fused multiply-add is fine here.
  using f2 = __attribute__((ext_vector_type(2))) float;
  const f2 ax2 = {ax, ax}, ay2 = {ay, ay}, ar2 = {ar, ar};
#pragma unroll
  for (int j = J0; j < J1; j += 2) {
    const f2 cx2 = {lane_bcast(cx, j), lane_bcast(cx, j + 1)}, cy2 = {lane_bcast(cy, j), lane_bcast(cy, j + 1)};
    const f2 cr2 = {lane_bcast(cr, j), lane_bcast(cr, j + 1)};
    const f2 dx = ax2 - cx2, dy = ay2 - cy2, R = ar2 + cr2;
    const f2 d2 = __builtin_elementwise_fma(dy, dy, dx * dx), R2 = R * R;
    bits |= (d2[0] > R2[0] ? 0u : 1u) << (j - J0);
    bits |= (d2[1] > R2[1] ? 0u : 1u) << (j + 1 - J0);
  }
}
// NMS first stage: circle test AND area-ratio test (|la - lc| <= lmax; NaN log-areas never drop a pair), see
// k_nms_cull_lanes.  The column data ROTATES through the lanes (v_mov_b32_dpp wave_ror:1, one lane per step) instead of
// being broadcast through v_readlane: the readlane form needs four SGPRs per column -- 128 live ones per half tile, which
// hipcc spilled to VGPR lanes (250 v_writelane / v_readlane per tile) and padded with s_nop for the SGPR read hazard;
// 22 vector instructions per test, 15 this way.  At step k a lane holds the column of lane (lane + k * dir) & 63, with
// dir found by rotating the lane index once (no assumption about the direction of the rotation).
__device__ __forceinline__ float dpp_ror1(float v) {
  // (bound_ctrl set: every lane has a source in a full-wave rotation, and the compiler then needs no zero-initialised
  // destination -- with `old = 0` it emitted a v_mov 0 in front of every rotation)
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x13C, 0xf, 0xf, true));
}
template <int STEPS>
__device__ __forceinline__ void nms_stage1_rot(float ax, float ay, float ar, float la, float& rx, float& ry, float& rr,
                                               float& rl, float lmax, unsigned& bits) {
#pragma unroll 4
  for (int k = 0; k < STEPS; k++) {
    const float dx = ax - rx, dy = ay - ry, R = ar + rr, dl = la - rl;
    const bool drop = (__builtin_fmaf(dy, dy, dx * dx) > R * R) || (fabsf(dl) > lmax);
    bits |= (drop ? 0u : 1u) << k;
    rx = dpp_ror1(rx); ry = dpp_ror1(ry); rr = dpp_ror1(rr); rl = dpp_ror1(rl);
  }
}
// the same test on TWO columns per step with packed-f32 arithmetic: the second column set starts 32 lanes on (lane ^ 32: a
// rotation by 32 in either direction), so 32 steps cover the tile; bits0 / bits1 = the two halves the caller walks.
// 25 vector instructions per two columns instead of 30.
__device__ __forceinline__ void nms_stage1_rot2(float ax, float ay, float ar, float la, float rx, float ry, float rr, float rl,
                                                float lmax, unsigned& bits0, unsigned& bits1) {
  using f2 = __attribute__((ext_vector_type(2))) float;
  const f2 a_x = {ax, ax}, a_y = {ay, ay}, a_r = {ar, ar}, a_l = {la, la};
  f2 cx = {rx, __shfl_xor(rx, 32)}, cy = {ry, __shfl_xor(ry, 32)}, cr = {rr, __shfl_xor(rr, 32)}, cl = {rl, __shfl_xor(rl, 32)};
#pragma unroll 4
  for (int k = 0; k < 32; k++) {
    const f2 dx = a_x - cx, dy = a_y - cy, R = a_r + cr, dl = a_l - cl;
    const f2 d2 = __builtin_elementwise_fma(dy, dy, dx * dx), R2 = R * R;
    const bool drop0 = (d2[0] > R2[0]) || (fabsf(dl[0]) > lmax);
    const bool drop1 = (d2[1] > R2[1]) || (fabsf(dl[1]) > lmax);
    bits0 |= (drop0 ? 0u : 1u) << k;
    bits1 |= (drop1 ? 0u : 1u) << k;
    cx[0] = dpp_ror1(cx[0]); cx[1] = dpp_ror1(cx[1]); cy[0] = dpp_ror1(cy[0]); cy[1] = dpp_ror1(cy[1]);
    cr[0] = dpp_ror1(cr[0]); cr[1] = dpp_ror1(cr[1]); cl[0] = dpp_ror1(cl[0]); cl[1] = dpp_ror1(cl[1]);
  }
}
__global__ __launch_bounds__(kThreads) void k_iou_cull_lanes(const PreBox* __restrict__ P1, const PreBox* __restrict__ P2,
                                                             int64_t row0, int64_t row1, int64_t m, int cols_per_wg,
                                                             uint2* __restrict__ gq,
                                                             unsigned long long* __restrict__ gcount,
                                                             unsigned long long cap) {
  __shared__ uint2 s_stage[kThreads / 64][kCrStage];
  __shared__ unsigned s_left[kThreads / 64];
  __shared__ unsigned long long s_base;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t wr0 = row0 + ((int64_t)blockIdx.y * (kThreads / 64) + wave) * 64;    // this wave's rows
  const int64_t row = wr0 + lane;
  const bool valid = row < row1;
  PreBox A = {};
  if (valid) A = P1[row];
  const float ax = A.x, ay = A.y, ar = A.r * 1.002f + 1e-3f;
  const int64_t jw = (int64_t)blockIdx.x * cols_per_wg;
  const int64_t jend = min(m, jw + cols_per_wg);
  uint2* stage = s_stage[wave];
  unsigned ns = 0;                               // wave-uniform
  auto flush = [&]() {
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(gcount, (unsigned long long)ns);
    base = ((unsigned long long)(uint32_t)__shfl((int)(base >> 32), 0) << 32) | (uint32_t)__shfl((int)(base & 0xffffffffu), 0);
    wave_lds_handoff();                          // other lanes' staged pairs
    for (unsigned k = lane; k < ns; k += 64)
      if (base + k < cap) gq[base + k] = stage[k];
    wave_lds_handoff();                          // ... are read before the stage is written again
    ns = 0;
  };
  PreBox Cn = {};                                // next chunk's column box of this lane (one chunk ahead)
  if (jw + lane < jend) Cn = P2[jw + lane];
  for (int64_t jc = jw; jc < jend; jc += 64) {
    const PreBox C = Cn;
    Cn = PreBox{};
    if (jc + 64 + lane < jend) Cn = P2[jc + 64 + lane];
    const int nc = (int)min((int64_t)64, jend - jc);
    // a column beyond the end gets a circle no row can reach (NaN rows reach everything: masked below)
    const float cx = C.x, cy = C.y, cr = lane < nc ? C.r * 1.002f : -3.0e38f;
    unsigned lo = 0, hi = 0;
    circle_bits<0, 32>(ax, ay, ar, cx, cy, cr, lo);
    circle_bits<32, 64>(ax, ay, ar, cx, cy, cr, hi);
    unsigned long long mask = ((unsigned long long)hi << 32) | lo;
    if (nc < 64) mask &= (1ull << nc) - 1ull;
    if (!valid) mask = 0;
    while (__any(mask != 0ull)) {
      const bool has = mask != 0ull;
      const int j = has ? __ffsll((long long)mask) - 1 : 0;
      if (has) mask &= mask - 1ull;
      PreBox B;
      B.x = __shfl(C.x, j); B.y = __shfl(C.y, j); B.w = __shfl(C.w, j); B.h = __shfl(C.h, j);
      B.c2 = __shfl(C.c2, j); B.s2 = __shfl(C.s2, j); B.r = 0.f; B.label = 0.f;
      const bool hit = has && !sat_disjoint(A, B);
      const unsigned long long bal = __ballot(hit);
      if (hit)
        stage[ns + __popcll(bal & ((1ull << lane) - 1ull))] = make_uint2((unsigned)(row - row0), (unsigned)(jc + j));
      ns += (unsigned)__popcll(bal);
      if (ns + 64 > kCrStage) flush();
    }
  }
  if (lane == 0) s_left[wave] = ns;
  __syncthreads();
  unsigned before = 0, all = 0;
#pragma unroll
  for (int w = 0; w < kThreads / 64; w++) {
    if (w < wave) before += s_left[w];
    all += s_left[w];
  }
  if (all == 0) return;                // uniform
  if (threadIdx.x == 0) s_base = atomicAdd(gcount, (unsigned long long)all);
  __syncthreads();
  const unsigned long long base = s_base + before;
  for (unsigned k = lane; k < ns; k += 64)
    if (base + k < cap) gq[base + k] = stage[k];
}

// HEAVY: dense list, one pair per lane, persistent grid.  Values go to a compact buffer (vals[e] for pair e): this pass
// runs while the zero-fill of the output is still in flight on the side stream, so it must not touch `out`.
// Two passes: the first gives every lane 8 candidate-point slots (16 KB of LDS per workgroup instead of 48: the exact
// IoU is a chain of dependent LDS round trips and was latency-bound at 3 waves per SIMD); a pair that produces more than
// 8 candidates (shared edges, duplicates: never in general position) leaves a marker, and k_iou_scatter redoes the
// marked pairs with the full 24 slots.  A genuine result with the marker's bits would only be recomputed to itself.
__global__ __launch_bounds__(kThreads) void k_iou_heavy(const PreBox* __restrict__ P1,
                                                        const PreBox* __restrict__ P2, int64_t row0,
                                                        const uint2* __restrict__ gq,
                                                        const unsigned long long* __restrict__ gcount,
                                                        unsigned long long cap, float* __restrict__ vals) {
  __shared__ float2 s_pts[kIouCap * kThreads];
  const unsigned long long total = *gcount;
  if (total > cap) return;       // list overflow: k_iou_scatter recomputes the chunk directly
  for (unsigned long long e = (unsigned long long)blockIdx.x * kThreads + threadIdx.x; e < total;
       e += (unsigned long long)gridDim.x * kThreads) {
    uint2 ij = gq[e];
    PreBox A = P1[row0 + ij.x];
    PreBox B = P2[ij.y];
    bool redo = false;
    const float v = rbox_iou<kThreads, kIouCap>(A, B, s_pts + threadIdx.x, &redo);
    vals[e] = redo ? __uint_as_float(kIouRedo) : v;
  }
}

// after the join with the zero-fill: out[i, j] = vals[e] for the listed pairs.  Overflow fallback: the pair list of this
// chunk did not fit (extremely dense inputs) -> recompute the whole chunk pair by pair (the divergence that the list
// avoids is irrelevant when most pairs are heavy anyway).
__global__ __launch_bounds__(kThreads) void k_iou_scatter(const PreBox* __restrict__ P1, const PreBox* __restrict__ P2,
                                                          int64_t row0, int64_t row1, int64_t m,
                                                          float* __restrict__ out, const uint2* __restrict__ gq,
                                                          const unsigned long long* __restrict__ gcount,
                                                          unsigned long long cap, const float* __restrict__ vals) {
  __shared__ float2 s_pts[24 * kThreads];
  const unsigned long long total = *gcount;
  if (total > cap) {
    const int64_t all = (row1 - row0) * m;
    for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < all; e += (int64_t)gridDim.x * kThreads) {
      int64_t i = row0 + e / m, j = e % m;
      PreBox A = P1[i], B = P2[j];
      if (!surely_disjoint(A.x, A.y, A.r, B.x, B.y, B.r) && !sat_disjoint(A, B))
        out[i * m + j] = rbox_iou<kThreads>(A, B, s_pts + threadIdx.x);
    }
    return;
  }
  for (unsigned long long e = (unsigned long long)blockIdx.x * kThreads + threadIdx.x; e < total;
       e += (unsigned long long)gridDim.x * kThreads) {
    const uint2 ij = gq[e];
    float v = vals[e];
    if (__float_as_uint(v) == kIouRedo)      // more than 8 candidate points in the first pass: all 24 slots here
      v = rbox_iou<kThreads>(P1[row0 + ij.x], P2[ij.y], s_pts + threadIdx.x);
    out[(row0 + ij.x) * m + ij.y] = v;
  }
}

// small fills by a kernel: hipMemsetAsync nodes of a captured graph were not replayed correctly (ROCm 7.2; see nms_core)
__global__ void k_fill_u32(uint32_t* __restrict__ p, uint32_t v, size_t count) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) p[i] = v;
}
inline void fill_u32(void* p, uint32_t v, size_t count, hipStream_t st) {
  if (count) k_fill_u32<<<(unsigned)((count + 255) / 256), 256, 0, st>>>(static_cast<uint32_t*>(p), v, count);
}

// 16 bytes per lane, grid-stride: 7 TB/s on MI355X (profiles/r02_nms_200k_pmc_before.txt, k_zero_words)
// PACE > 0: s_sleep between the stores of a wave -- a fill that runs flat out saturates the memory system and every
// dependent read of the kernels beside it takes ~10 us (the pair finder stretched from 59 to 110 us whatever it did)
template <int PACE>
__global__ __launch_bounds__(256) void k_fill_zero(float* __restrict__ out, unsigned long long n) {
  const unsigned long long n4 = n / 4, stride = (unsigned long long)gridDim.x * 256;
  float4* o4 = reinterpret_cast<float4*>(out);
  for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    o4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (PACE > 0) __builtin_amdgcn_s_sleep(PACE);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) out[n4 * 4 + threadIdx.x] = 0.f;
}

__global__ __launch_bounds__(kThreads) void k_iou_pairs(const float* __restrict__ b1,
                                                        const float* __restrict__ b2, int64_t n,
                                                        float* __restrict__ out) {
  __shared__ float2 s_pts[24 * kThreads];
  int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  const float* a = b1 + 5 * i;
  const float* b = b2 + 5 * i;
  PreBox A = make_prebox(a[0], a[1], a[2], a[3], a[4], 0.f);
  PreBox B = make_prebox(b[0], b[1], b[2], b[3], b[4], 0.f);
  out[i] = rbox_iou<kThreads>(A, B, s_pts + threadIdx.x);
}

constexpr unsigned long long kIouQueueCap = 32ull << 20;  // 32 Mi pairs = 256 MiB

// ================================================================= NMS
// Pipeline of one call (all on the device, no host synchronisation):
//   1. sort rows by (group, score desc) and then stably by segment (label / image x class)         [rocPRIM radix sorts]
//   2. per segment: sort its rows along a Morton curve of the box centres; 64-row spatial blocks, bounding box of the
//      (inflated) circumscribed circles of each block
//   3. TILE FILTER: upper-triangle 64 x 64 tiles of a segment whose two block bounding boxes are apart hold only
//      exact-zero pairs -> dropped; the others go to a tile list (boxes of a 1024^2 chip only ever overlap their spatial
//      neighbours: 83 % of the tiles of BASELINE config 5 go away here; with all-pairs tiles in score order the cull
//      was VALU-bound on 1.33e9 circle tests: profiles/r02_nms_200k_pmc_before.txt)
//   4. CULL over the tile list: circle test for every pair of a tile, survivors -> LDS list -> dense second stage
//      (separating axes + IoU upper bound, rbox_geom.hpp:nms_pair_skippable) -> pair list (score positions i < j)
//   5. DENSE IoU pass over the pair list, all lanes busy; pairs with IoU > thr (reference GPU rule, strict) become EDGES
//      i -> j ("i suppresses j if i is kept")
//   6. RESOLVE the greedy order by parallel rounds over the edge list instead of a serial scan of a bitmask (which ran one
//      workgroup per segment at 0.25 waves/CU): a row is KEPT once all its in-neighbours are removed, REMOVED once one
//      of them is kept; every round settles at least the first unsettled row of each chain, so the result is exactly
//      the reference's host loop (ml_nms cuda.cu:120-131); 4-6 rounds on detector-like and on random inputs; a
//      single-workgroup kernel finishes chains that are longer than the fixed number of launched rounds
//   7. keep flags -> compaction in score order.
// No N x N/64 suppression mask exists any more (the reference's is 5.0 GB at 200 k rows and goes to the host; round 1
// kept a 334 MB per-segment one on the device).  If the pair or edge list overflows (pathologically dense inputs), a
// memory-free direct greedy kernel redoes the segments (slow, exact).
// device-side scalars of one call (zeroed by k_nms_prep)
struct NmsCounters {
  unsigned long long pairs;      // cull survivors (true total, may exceed the list)
  unsigned long long edges;      // pairs with IoU > thr (true total)
  unsigned long long tiles;      // tiles that passed the filter (true total)
  unsigned long long alive_list; // edges handed to the clean-up kernel
  unsigned long long bucket_cursor;  // clean-up kernel: edges of segments beyond the LDS capacity, reserved in the dead edge buffer
  uint32_t status;               // bit 0: a list overflowed -> direct greedy fallback ran
  uint32_t alive[16];            // edges still between two unsettled rows after round r
};

// ---- PRELUDE (round 3).  Round 2 ran three rocPRIM sorts one after the other (global score order, stable regrouping by
// segment, Morton order inside the segments) with a dozen 5 us kernels between them: ~300 us of launch latency at 200 k
// rows (profiles/r02_nms_200k_timeline.txt).  Now ONE kernel builds all three 64-bit keys from the inputs,
//   A = segment | score   (the order the greedy resolve works in),
//   B = segment | Morton  (the spatial blocks of the cull; big segments only),
//   C = group   | score   (the order of the output),
// the three sorts are independent of each other and run SIDE BY SIDE (A on the caller's stream, B and C on two side
// streams), and everything between the sort and the cull is three launches: segment boundaries by block counts
// (k_nms_seg_count) + one fused pass (k_nms_pos_meta: prefix of the counts, in-block scan, seg_start, pre-processed boxes,
// initial states, inverse permutation), the spatial gather, and a tile filter that needs no per-segment scan (one wave
// per 64-row block walks that block's row of the upper triangle).
__device__ __forceinline__ uint32_t nms_segkey(const float* __restrict__ labels, const int32_t* __restrict__ seg_ids,
                                               uint32_t ignore_key, int64_t i) {
  if (seg_ids) return seg_ids[i] < 0 ? ignore_key : (uint32_t)seg_ids[i];
  if (labels) {
    float l = labels[i];
    if (l == 0.0f) l = 0.0f;  // -0 == +0 in the reference's float compare
    return float_sortable(l);
  }
  return 0u;
}

// rows at or behind *row_limit (when given: the candidate count of the producer, s2a_multiclass_candidates) are padding by
// contract and are never read: a detector batch fills a quarter of its static buffer
__device__ __forceinline__ int64_t seg_row_limit(const long long* __restrict__ row_limit, int64_t n) {
  if (!row_limit) return n;
  const long long v = *row_limit;
  return v < 0 ? 0 : (v < (long long)n ? (int64_t)v : n);
}
constexpr int kPrepBlocks = 1024;     // upper bound of k_nms_prep's grid = number of bounding-box partials
constexpr int kSegCountBlocks = 4096; // upper bound of the block-count arrays (segment heads, kept rows)

// block-wide sum of two values (256 threads); every thread gets both totals
__device__ __forceinline__ void block_sum2(unsigned& a, unsigned& b) {
  __shared__ unsigned s_a[4], s_b[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
  __syncthreads();                               // (protects s_a / s_b against the previous use)
  if ((threadIdx.x & 63) == 0) { s_a[threadIdx.x >> 6] = a; s_b[threadIdx.x >> 6] = b; }
  __syncthreads();
  a = s_a[0] + s_a[1] + s_a[2] + s_a[3];
  b = s_b[0] + s_b[1] + s_b[2] + s_b[3];
}

// ---- own order-B sort (round 4).  The spatial order only has to GROUP the rows by segment and keep them spatially coherent
// inside a segment: a counting sort into (segment, Morton cell) buckets does that in three launches where rocPRIM's pair
// sort takes nine (73 us of host-paced launches at 200 k rows).  The segment key is 32 arbitrary bits (float_sortable(label)
// for the drop-in op), so the distinct keys are first hashed into a small table and numbered in order of arrival -- the B
// order may list the segments in any order: everything behind the cull works on B positions, and the direct fallback,
// which works in score order, gets segment starts of its own (seg_start_a).  More than ~8 k distinct keys, or 65 536
// (segment, cell) buckets exceeded: status bit 1 -> that fallback settles the call (slow, exact).
constexpr uint32_t kSpbSlots = 16384, kSpbHist = 65536, kSpbGroups = kSpbHist / 64, kSpbPending = 0x7fffffffu;
constexpr unsigned long long kSpbEmpty = ~0ull;
__device__ __forceinline__ uint32_t spb_hash(uint32_t k) {
  k ^= k >> 16; k *= 0x7feb352du; k ^= k >> 15; k *= 0x846ca68bu; k ^= k >> 16;
  return k & (kSpbSlots - 1);
}
// insert key (first thread to claim an empty slot numbers it); safe against stale reads: a slot only ever goes
// EMPTY -> (key | pending) -> (key | id), and the claim itself is a device-scope compare-and-swap
__device__ __forceinline__ void spb_insert(unsigned long long* __restrict__ htab, uint32_t* __restrict__ ctr, uint32_t key) {
  uint32_t h = spb_hash(key);
  for (uint32_t probe = 0; probe < kSpbSlots; probe++) {
    unsigned long long cur = __hip_atomic_load(&htab[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == kSpbEmpty) {
      const unsigned long long want = ((unsigned long long)key << 32) | kSpbPending;
      cur = atomicCAS(&htab[h], kSpbEmpty, want);
      if (cur == kSpbEmpty) {
        const uint32_t id = atomicAdd(&ctr[0], 1u);
        atomicExch(&htab[h], ((unsigned long long)key << 32) | id);
        return;
      }
    }
    if ((uint32_t)(cur >> 32) == key) return;
    h = (h + 1) & (kSpbSlots - 1);
  }
  atomicOr(&ctr[1], 1u);                         // table full
}
// the distinct keys of a 256-thread workgroup, each inserted ONCE: 200 k rows with 15 labels are 200 k reads of 15 hot
// slots otherwise, and at the start of the launch every one of them still sees "empty" and goes for the compare-and-swap
// (same-address atomics serialise at ~11 ns: the insert took 100 us).  A 64-slot LDS set per workgroup; a key that does
// not fit goes to the global table directly.
__device__ __forceinline__ void spb_insert_block(unsigned long long* __restrict__ htab, uint32_t* __restrict__ ctr,
                                                 uint32_t key, bool valid) {
  __shared__ unsigned long long s_set[64];
  if (threadIdx.x < 64) s_set[threadIdx.x] = kSpbEmpty;
  __syncthreads();
  bool mine = false, direct = false;
  if (valid) {
    uint32_t h = (key * 0x9e3779b1u) >> 26;      // 6 bits
    direct = true;
    for (int probe = 0; probe < 8; probe++) {
      const unsigned long long prev = atomicCAS(&s_set[h], kSpbEmpty, (unsigned long long)key);
      if (prev == kSpbEmpty) { mine = true; direct = false; break; }
      if ((uint32_t)prev == key) { direct = false; break; }
      h = (h + 1) & 63;
    }
  }
  if (mine || direct) spb_insert(htab, ctr, key);
}
__device__ __forceinline__ uint32_t spb_lookup(const unsigned long long* __restrict__ htab, uint32_t key) {
  uint32_t h = spb_hash(key);
  for (uint32_t probe = 0; probe < kSpbSlots; probe++) {
    const unsigned long long cur = htab[h];
    if (cur == kSpbEmpty) break;
    if ((uint32_t)(cur >> 32) == key) return (uint32_t)cur;
    h = (h + 1) & (kSpbSlots - 1);
  }
  return 0xffffffffu;
}
// bits of the Morton code that go into the bucket index: ~4 rows per bucket, at most kSpbHist buckets for the m distinct
// segments.  (16 rows per bucket made the scan one short workgroup but cost the cull 15-25 us: rows arrive in a bucket in
// any order, and the cull's 64-row blocks are as compact as the buckets are fine.)
__device__ __forceinline__ uint32_t spb_buckets(int64_t n) {
  uint32_t b = 1024;
  while (b < kSpbHist && (int64_t)b * 4 < n) b <<= 1;
  return b;
}
__device__ __forceinline__ int spb_cell_bits(uint32_t m, int64_t n) {
  int sb = 0, tb = 0;
  while ((1u << sb) < m) sb++;                   // ceil(log2(m))
  while ((1u << tb) < spb_buckets(n)) tb++;
  const int cb = tb - sb;
  return cb < 0 ? 0 : (cb > 12 ? 12 : cb);
}

// keys of all three orders + the workspace initialisation (counters, round flags, block bounding boxes: by kernels, never
// hipMemsetAsync -- DESIGN 5) + per-block partial bounding box of the finite centres (no atomics, so no init hazard)
__global__ __launch_bounds__(256) void k_nms_prep(const float* __restrict__ dets5, const float* __restrict__ scores,
                                                  const float* __restrict__ labels, const int32_t* __restrict__ seg_ids,
                                                  const int32_t* __restrict__ groups, uint32_t num_groups,
                                                  uint32_t ignore_key, int64_t n, unsigned long long* __restrict__ keyA,
                                                  unsigned long long* __restrict__ keyC, int32_t* __restrict__ idx,
                                                  uint4* __restrict__ bbox_part, NmsCounters* __restrict__ C,
                                                  uint32_t* __restrict__ blocked32, size_t nblocked32,
                                                  uint32_t* __restrict__ seg_cnt, size_t nseg_cnt,
                                                  uint2* __restrict__ lo, uint2* __restrict__ hi, size_t slots,
                                                  unsigned long long* __restrict__ htab, uint32_t* __restrict__ hist,
                                                  uint32_t* __restrict__ spb_ctr, uint32_t hist_zero,
                                                  const long long* __restrict__ row_limit) {
  const size_t stride = (size_t)gridDim.x * 256, i0 = (size_t)blockIdx.x * 256 + threadIdx.x;
  // rows at or behind *row_limit are padding on EVERY path (the own segment sort never reads them; here they get the
  // ignore key, as rows with a negative segment id): the result does not depend on which sort a call takes
  const size_t nl = seg_ids ? (size_t)seg_row_limit(row_limit, n) : (size_t)n;
  if (htab) {                                    // own order-B sort: table, histogram and counters (filled by k_spb_insert)
    for (size_t i = i0; i < kSpbSlots; i += stride) htab[i] = kSpbEmpty;
    for (size_t i = i0; i < kSpbHist + 64; i += stride) hist[i] = 0u;
    if (i0 < 4) spb_ctr[i0] = 0u;
  } else {
    for (size_t i = i0; i < hist_zero; i += stride) hist[i] = 0u;     // own segment sort: the per-segment row counts
  }
  for (size_t i = i0; i < sizeof(NmsCounters) / 4; i += stride) reinterpret_cast<uint32_t*>(C)[i] = 0u;
  for (size_t i = i0; i < nblocked32; i += stride) blocked32[i] = 0u;
  for (size_t i = i0; i < nseg_cnt; i += stride) seg_cnt[i] = 0u;
  if (lo)
    for (size_t i = i0; i < slots; i += stride) { lo[i] = make_uint2(0xffffffffu, 0xffffffffu); hi[i] = make_uint2(0u, 0u); }
  uint32_t lx = 0xffffffffu, ly = 0xffffffffu, hx = 0u, hy = 0u;
  for (size_t i = i0; i < (size_t)n; i += stride) {
    const bool pad = i >= nl;
    const uint32_t sk = pad ? ignore_key : nms_segkey(labels, seg_ids, ignore_key, (int64_t)i);
    unsigned long long g = groups ? (unsigned long long)(uint32_t)groups[i] : 0ull;
    if (pad || (seg_ids && seg_ids[i] < 0)) g = num_groups;   // ignored row: sorts behind every real group
    const unsigned long long sc = (unsigned long long)(~float_sortable(scores[i]));
    keyA[i] = ((unsigned long long)sk << 32) | sc;
    if (keyC) keyC[i] = (g << 32) | sc;
    idx[i] = (int32_t)i;
    const float x = dets5[5 * i], y = dets5[5 * i + 1];
    if (isfinite(x) && isfinite(y)) {
      const uint32_t ux = float_sortable(x), uy = float_sortable(y);
      lx = min(lx, ux); hx = max(hx, ux); ly = min(ly, uy); hy = max(hy, uy);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    lx = min(lx, (uint32_t)__shfl_xor((int)lx, o)); ly = min(ly, (uint32_t)__shfl_xor((int)ly, o));
    hx = max(hx, (uint32_t)__shfl_xor((int)hx, o)); hy = max(hy, (uint32_t)__shfl_xor((int)hy, o));
  }
  __shared__ uint4 s_bb[4];
  if ((threadIdx.x & 63) == 0) s_bb[threadIdx.x >> 6] = make_uint4(lx, ly, hx, hy);
  __syncthreads();
  if (threadIdx.x == 0) {
    uint4 r = s_bb[0];
    for (int w = 1; w < 4; w++) {
      r.x = min(r.x, s_bb[w].x); r.y = min(r.y, s_bb[w].y); r.z = max(r.z, s_bb[w].z); r.w = max(r.w, s_bb[w].w);
    }
    bbox_part[blockIdx.x] = r;
  }
}

__device__ __forceinline__ uint32_t spread10(uint32_t v) {   // 10 bits -> every other bit of 20
  v = (v | (v << 8)) & 0x00ff00ffu; v = (v | (v << 4)) & 0x0f0f0f0fu;
  v = (v | (v << 2)) & 0x33333333u; v = (v | (v << 1)) & 0x55555555u;
  return v;
}

// spatial key of every row: (segment key, 20-bit Morton code of the centre on a 1024 x 1024 grid over the bounding box
// of all finite centres).  Same segment key in the top bits as key A: both orders list the segments alike.
__global__ __launch_bounds__(256) void k_nms_spkeys(const float* __restrict__ dets5,
                                                    const unsigned long long* __restrict__ keyA,
                                                    const uint4* __restrict__ bbox_part, int nparts, int64_t n,
                                                    unsigned long long* __restrict__ keyB,
                                                    unsigned long long* __restrict__ htab, uint32_t* __restrict__ spb_ctr) {
  __shared__ uint4 s_bb[4];
  uint4 r = make_uint4(0xffffffffu, 0xffffffffu, 0u, 0u);
  for (int k = threadIdx.x; k < nparts; k += 256) {
    const uint4 v = bbox_part[k];
    r.x = min(r.x, v.x); r.y = min(r.y, v.y); r.z = max(r.z, v.z); r.w = max(r.w, v.w);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    r.x = min(r.x, (uint32_t)__shfl_xor((int)r.x, o)); r.y = min(r.y, (uint32_t)__shfl_xor((int)r.y, o));
    r.z = max(r.z, (uint32_t)__shfl_xor((int)r.z, o)); r.w = max(r.w, (uint32_t)__shfl_xor((int)r.w, o));
  }
  if ((threadIdx.x & 63) == 0) s_bb[threadIdx.x >> 6] = r;
  __syncthreads();
  r = s_bb[0];
  for (int w = 1; w < 4; w++) {
    r.x = min(r.x, s_bb[w].x); r.y = min(r.y, s_bb[w].y); r.z = max(r.z, s_bb[w].z); r.w = max(r.w, s_bb[w].w);
  }
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (htab) spb_insert_block(htab, spb_ctr, p < n ? (uint32_t)(keyA[p] >> 32) : 0u, p < n);   // own order-B sort, pass 1 (k_spb_*)
  if (p >= n) return;
  const float x0 = sortable_float(r.x), y0 = sortable_float(r.y), x1 = sortable_float(r.z), y1 = sortable_float(r.w);
  const float sx = x1 > x0 ? 1023.f / (x1 - x0) : 0.f, sy = y1 > y0 ? 1023.f / (y1 - y0) : 0.f;
  float qx = (dets5[5 * p] - x0) * sx, qy = (dets5[5 * p + 1] - y0) * sy;
  qx = qx >= 0.f ? fminf(qx, 1023.f) : 0.f;      // NaN / nothing finite -> 0
  qy = qy >= 0.f ? fminf(qy, 1023.f) : 0.f;
  const uint32_t m = spread10((uint32_t)qx) | (spread10((uint32_t)qy) << 1);
  keyB[p] = ((keyA[p] >> 32) << 20) | m;
}

// own order-B sort (pass 1, the hash insert, rides in k_nms_spkeys)
// pass 2: bucket = (segment number, top bits of the Morton code) of every row, its rank inside the bucket
__global__ __launch_bounds__(256) void k_spb_hist(const unsigned long long* __restrict__ keyB, int64_t n,
                                                  const unsigned long long* __restrict__ htab,
                                                  uint32_t* __restrict__ ctr, uint32_t* __restrict__ hist,
                                                  uint32_t* __restrict__ bucket, uint32_t* __restrict__ rank,
                                                  NmsCounters* __restrict__ C, uint32_t ignore_key, int use_ignore) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool valid = p < n;
  uint32_t b = 0xffffffffu;
  if (valid) {
    const uint32_t m = ctr[0];
    const int cb = spb_cell_bits(m, n);
    const unsigned long long k = keyB[p];
    const uint32_t sk = (uint32_t)(k >> 20);
    uint32_t id = spb_lookup(htab, sk);
    uint32_t cell = ((uint32_t)k & 0xfffffu) >> (20 - cb);
    // padding rows of the batched detector (segment -1): never compared, never kept -- their place inside their segment is
    // irrelevant, and their coordinates are whatever the caller left there (typically all equal): spread them over the
    // segment's cells by row number, or 600 k of them queue for ONE counter (6.7 ms of same-address atomics, measured)
    if (use_ignore && sk == ignore_key) cell = (((uint32_t)p * 0x9e3779b1u) >> 12) & ((1u << cb) - 1u);
    if (id == 0xffffffffu || id >= kSpbPending || ctr[1] != 0u || ((unsigned long long)id << cb) + cell >= spb_buckets(n)) {
      id = 0;                                    // cannot be bucketed: the direct fallback settles the call
      if ((C->status & 2u) == 0u) atomicOr(&C->status, 2u);
    }
    b = (id << cb) | cell;
    bucket[p] = b;
  }
  // rank inside the bucket by the returning atomic; the lanes that share the FIRST lane's bucket go together (one atomic for
  // the group: rows with identical centres -- a degenerate input -- would otherwise serialise on one counter)
  const uint32_t b0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
  const unsigned long long same = __ballot(valid && b == b0);
  const int lane = threadIdx.x & 63;
  if (valid && b == b0) {
    const int leader = __ffsll((long long)same) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(&hist[b], (uint32_t)__popcll(same));
    base = (uint32_t)__shfl((int)base, leader);
    rank[p] = base + (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
  } else if (valid) {
    rank[p] = atomicAdd(&hist[b], 1u);
  }
}
// pass 3: exclusive scan of the histogram in two levels -- a wave per group of 64 buckets (in place, group total aside),
// then the <= 1 024 group totals, by every workgroup of pass 4 for itself.  (One workgroup over 61 k buckets took 75 us, uncoalesced; the scan as
// the last act of the LAST workgroup of pass 2 -- fence, ticket, scan -- 97 us: a device-scope release per workgroup writes
// the XCD's dirty L2 lines back 782 times.)
__global__ __launch_bounds__(256) void k_spb_scan1(uint32_t* __restrict__ hist, uint32_t* __restrict__ group_sum, int64_t n) {
  const uint32_t g = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (g * 64 >= spb_buckets(n)) return;
  const unsigned v = hist[g * 64 + lane];
  unsigned incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned t = __shfl_up(incl, o);
    if (lane >= (unsigned)o) incl += t;
  }
  hist[g * 64 + lane] = incl - v;
  if (lane == 63) group_sum[g] = incl;
}
// pass 4: rows into their places.  The second level of the scan (<= 1 024 group totals) is done by every workgroup for itself
// in LDS -- 4 KB of L2 reads and a block scan instead of a launch of its own (k_spb_scan2, 4.6 us of launch-paced time)
__global__ __launch_bounds__(256) void k_spb_scatter(const unsigned long long* __restrict__ keyB, int64_t n,
                                                     const uint32_t* __restrict__ hist, const uint32_t* __restrict__ group_sum,
                                                     const uint32_t* __restrict__ bucket, const uint32_t* __restrict__ rank,
                                                     unsigned long long* __restrict__ keyB_s, int32_t* __restrict__ perm_sp) {
  __shared__ unsigned s_base[1024];
  __shared__ unsigned s_w[4];
  {
    const uint32_t groups = spb_buckets(n) / 64;               // 16 .. 1024
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned v[4], tot = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t g = threadIdx.x * 4 + k;
      v[k] = g < groups ? group_sum[g] : 0u;
      tot += v[k];
    }
    unsigned incl = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned t = __shfl_up(incl, o);
      if (lane >= o) incl += t;
    }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    unsigned base = incl - tot;
    for (int w = 0; w < wave; w++) base += s_w[w];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      s_base[threadIdx.x * 4 + k] = base;
      base += v[k];
    }
    __syncthreads();
  }
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  const uint32_t b = bucket[p];
  const uint32_t pos = s_base[b >> 6] + hist[b] + rank[p];
  keyB_s[pos] = keyB[p];
  perm_sp[pos] = (int32_t)p;
}

// number of segment heads (rows whose segment key -- key >> shift -- differs from the row before) per block of `rows`
// sorted rows, and the position + 1 of the block's last head (0: none) for the segment-start carry of k_nms_sp_meta
__global__ __launch_bounds__(256) void k_nms_seg_count(const unsigned long long* __restrict__ keys, int shift, int64_t n,
                                                       int rows, uint32_t* __restrict__ cnt,
                                                       uint32_t* __restrict__ lasthead) {
  const int64_t r0 = (int64_t)blockIdx.x * rows, r1 = min(n, r0 + rows);
  unsigned c = 0, last = 0;
  for (int64_t p = r0 + threadIdx.x; p < r1; p += 256)
    if (p == 0 || (keys[p] >> shift) != (keys[p - 1] >> shift)) { c++; last = (unsigned)p + 1u; }
  __shared__ unsigned s_last[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) last = max(last, (unsigned)__shfl_xor((int)last, o));
  if ((threadIdx.x & 63) == 0) s_last[threadIdx.x >> 6] = last;
  unsigned dummy = 0;
  block_sum2(c, dummy);                          // (its barriers also publish s_last)
  if (threadIdx.x == 0) {
    cnt[blockIdx.x] = c;
    if (lasthead) lasthead[blockIdx.x] = max(max(s_last[0], s_last[1]), max(s_last[2], s_last[3]));
  }
}

// offset of block b in an array of per-block counts (<= kSegCountBlocks of them) and the total
__device__ __forceinline__ void count_prefix(const uint32_t* __restrict__ cnt, int nb, int b, unsigned& before,
                                             unsigned& total) {
  before = 0; total = 0;
  for (int j = threadIdx.x; j < nb; j += 256) {
    const unsigned c = cnt[j];
    total += c;
    if (j < b) before += c;
  }
  block_sum2(before, total);
}

// inclusive scan of one flag per thread over the 256 threads of the block; *block_total = number of set flags
__device__ __forceinline__ unsigned block_scan_flag(bool f, unsigned* block_total) {
  __shared__ unsigned s_w[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long bal = __ballot(f);
  __syncthreads();
  if (lane == 0) s_w[wave] = (unsigned)__popcll(bal);
  __syncthreads();
  unsigned base = 0;
#pragma unroll
  for (int w = 0; w < 4; w++) base += w < wave ? s_w[w] : 0u;
  *block_total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
  return base + (unsigned)__popcll(bal & ((2ull << lane) - 1ull));
}

struct TileRef {
  uint32_t seg_start, ns, rb, cb;
};

// Greedy scan, one workgroup per segment at a time (persistent over segments).
// Equivalent to the reference's host loop (ml_nms cuda.cu:120-131) restricted to a segment.
__global__ __launch_bounds__(kThreads) void k_nms_scan(const unsigned long long* __restrict__ mask,
                                                       const uint32_t* __restrict__ seg_start,
                                                       const uint32_t* __restrict__ num_seg,
                                                       const unsigned long long* __restrict__ mask_off,
                                                       const uint32_t* __restrict__ nblk,
                                                       const int32_t* __restrict__ perm_seg,
                                                       uint8_t* __restrict__ keep_orig,
                                                       uint32_t max_blocks,
                                                       const unsigned long long* __restrict__ words_total,
                                                       unsigned long long words_bound,
                                                       uint32_t* __restrict__ status,
                                                       const unsigned long long* __restrict__ occ) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long s_remv[];  // max_blocks + 2 | mask cache
  unsigned long long* s_keep = s_remv + max_blocks;
  unsigned long long* s_mask = s_remv + max_blocks + 10;  // kScanCacheWords words (after keep[1] + rows[8] + pad)
  const uint32_t S = *num_seg;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (*words_total > words_bound) {  // mask would not fit: keep nothing, report (status bit 0)
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(status, 1u);
    return;
  }
  for (uint32_t s = blockIdx.x; s < S; s += gridDim.x) {
    const uint32_t st = seg_start[s];
    const uint32_t ns = seg_start[s + 1] - st;
    const uint32_t B = nblk[s];  // 0 for the ignored-rows segment: nothing kept
    if (B > max_blocks) {  // segment larger than promised: its rows stay suppressed (status bit 1)
      if (threadIdx.x == 0) atomicOr(status, 2u);
      continue;
    }
    const unsigned long long* M = mask + mask_off[s];
    // small segment (the usual per-(image, class) case): pull its whole mask into LDS with one
    // coalesced sweep, so the serial block loop below never waits on HBM/L2 latency again
    const unsigned long long nwords = (unsigned long long)ns * B;
    const bool cached = nwords <= kScanCacheWords;
    __syncthreads();
    for (uint32_t c = threadIdx.x; c < B; c += kThreads) s_remv[c] = 0ull;
    // middle-sized segment (a dominant class of an image: ~2000 rows): its mask does not fit, but the two words
    // the serial chain waits for do -- every row's DIAGONAL word and its word-occupancy word go to LDS up front,
    // leaving one global latency per block (the mask words of the kept rows) instead of three
    const bool semi = !cached && occ != nullptr && B <= 64 && 2ull * ns <= kScanCacheWords;
    if (cached) {
      for (uint32_t i = threadIdx.x; i < nwords; i += kThreads) s_mask[i] = M[i];
      M = s_mask;   // (generic pointer: LDS from here on)
    } else if (semi) {
      const unsigned long long seg_off0 = mask_off[s];
      for (uint32_t i = threadIdx.x; i < ns; i += kThreads) {
        s_mask[i] = M[(unsigned long long)i * B + (i >> 6)];
        s_mask[ns + i] = occ[((seg_off0 + (unsigned long long)i * B) >> 6) + st + i];
      }
    }
    __syncthreads();
    unsigned char* s_rows = reinterpret_cast<unsigned char*>(s_keep + 1);   // kept rows of the block (64 B)
    for (uint32_t b = 0; b < B; b++) {
      if (wave == 0) {
        uint32_t rowl = b * 64 + lane;
        bool valid = rowl < ns;
        unsigned long long d = valid ? (semi ? s_mask[rowl] : M[(unsigned long long)rowl * B + b]) : 0ull;
        const unsigned long long r0 = s_remv[b];
        // the running remove word lives in SGPRs (it is wave-uniform): the 64-step greedy resolve is
        // then a chain of scalar bit tests, one v_readlane pair per step
        uint32_t rlo = __builtin_amdgcn_readfirstlane((uint32_t)r0);
        uint32_t rhi = __builtin_amdgcn_readfirstlane((uint32_t)(r0 >> 32));
        uint32_t dlo = (uint32_t)d, dhi = (uint32_t)(d >> 32);
#pragma unroll
        for (int t = 0; t < 32; t++) {
          // readlane returns a signed int: go through uint32_t or bit 31 sign-extends
          uint32_t tl = (uint32_t)__builtin_amdgcn_readlane(dlo, t), th = (uint32_t)__builtin_amdgcn_readlane(dhi, t);
          if (!((rlo >> t) & 1u)) { rlo |= tl; rhi |= th; }
        }
#pragma unroll
        for (int t = 32; t < 64; t++) {
          uint32_t th = (uint32_t)__builtin_amdgcn_readlane(dhi, t);   // bits <= t are never set in row t
          if (!((rhi >> (t - 32)) & 1u)) rhi |= th;
        }
        const unsigned long long r = ((unsigned long long)rhi << 32) | rlo;
        unsigned long long validmask = (ns - b * 64 >= 64) ? ~0ull : ((1ull << (ns - b * 64)) - 1ull);
        unsigned long long kb = ~r & validmask;
        if (valid) keep_orig[perm_seg[st + rowl]] = (uint8_t)((kb >> lane) & 1ull);
        if ((kb >> lane) & 1ull) s_rows[__popcll(kb & ((1ull << lane) - 1ull))] = (unsigned char)lane;
        if (lane == 0) *s_keep = kb;
      }
      __syncthreads();
      // rows kept in this block suppress later columns: OR their mask rows into the running remove
      // vector.  The four waves split the kept rows (every 4th each, four loads in flight per lane);
      // lanes walk consecutive words of a row (coalesced); partial ORs meet in LDS (ds_or_b64).
      const int nk = __popcll(*s_keep);
      if (!cached && occ) {
        // big segment: visit only the non-zero words of the kept rows (occupancy bitmap), columns > b
        const uint32_t OW = (B + 63) >> 6;
        const unsigned long long seg_off = mask_off[s];
        for (uint32_t item = threadIdx.x; item < (uint32_t)nk * OW; item += kThreads) {
          const uint32_t k = item / OW, q = item % OW;
          const uint32_t rowl = b * 64 + s_rows[k];
          unsigned long long o = semi ? s_mask[ns + rowl] : occ[((seg_off + (unsigned long long)rowl * B) >> 6) + st + rowl + q];
          if (q < (b >> 6)) o = 0ull;
          else if (q == (b >> 6)) o &= ~((2ull << (b & 63)) - 1ull);
          while (o) {
            const uint32_t c = q * 64 + (uint32_t)__builtin_ctzll(o);
            o &= o - 1ull;
            atomicOr(&s_remv[c], M[(unsigned long long)rowl * B + c]);
          }
        }
        __syncthreads();
        continue;
      }
      for (uint32_t c0 = b + 1; c0 < B; c0 += 64) {
        const uint32_t c = c0 + lane;
        const bool cv = c < B;
        unsigned long long acc = 0ull;
        int k = wave;
        for (; k + 12 < nk; k += 16) {
          const unsigned long long t0 = s_rows[k], t1 = s_rows[k + 4], t2 = s_rows[k + 8], t3 = s_rows[k + 12];
          if (cv) {
            unsigned long long m0 = M[(b * 64ull + t0) * B + c], m1 = M[(b * 64ull + t1) * B + c];
            unsigned long long m2 = M[(b * 64ull + t2) * B + c], m3 = M[(b * 64ull + t3) * B + c];
            acc |= (m0 | m1) | (m2 | m3);
          }
        }
        for (; k < nk; k += 4)
          if (cv) acc |= M[(b * 64ull + s_rows[k]) * B + c];
        if (cv && acc) atomicOr(&s_remv[c], acc);
      }
      __syncthreads();
    }
  }
}


// ---- keep list of the drop-in ops, in the output order (perm_glob = rows by descending score): kept rows per block,
// then prefix + in-block scan + write (two launches; rocprim::select took four and a memset node)
__global__ __launch_bounds__(256) void k_nms_keep_count(const uint8_t* __restrict__ keep_orig,
                                                        const int32_t* __restrict__ perm_glob, int64_t n, int rows,
                                                        uint32_t* __restrict__ cnt) {
  const int64_t r0 = (int64_t)blockIdx.x * rows, r1 = min(n, r0 + rows);
  unsigned c = 0, dummy = 0;
  for (int64_t p = r0 + threadIdx.x; p < r1; p += 256) c += keep_orig[perm_glob[p]] ? 1u : 0u;
  block_sum2(c, dummy);
  if (threadIdx.x == 0) cnt[blockIdx.x] = c;
}
__global__ __launch_bounds__(256) void k_nms_keep_write(const uint8_t* __restrict__ keep_orig,
                                                        const int32_t* __restrict__ perm_glob, int64_t n, int rows,
                                                        const uint32_t* __restrict__ cnt, int nb,
                                                        int64_t* __restrict__ keep, int64_t* __restrict__ count_dev,
                                                        uint32_t* __restrict__ host_count /* pinned, mapped; may be NULL */,
                                                        const NmsCounters* __restrict__ C = nullptr, unsigned long long cap = 0,
                                                        unsigned long long ecap = 0, unsigned long long tile_cap = 0) {
  unsigned before, total;
  count_prefix(cnt, nb, blockIdx.x, before, total);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    *count_dev = (int64_t)total;
    // (a synchronous caller reads the count from host-mapped memory after its stream synchronise: no copy launch)
    if (host_count) __hip_atomic_store(host_count, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // deferred score order: did a list overflow (k_nms_greedy_direct's own condition)?  The host then runs the fallback.
    if (host_count && C)
      __hip_atomic_store(host_count + 1, (C->pairs > cap || C->edges > ecap || C->tiles > tile_cap || (C->status & 2u)) ? 1u : 0u,
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  const int64_t r0 = (int64_t)blockIdx.x * rows, r1 = min(n, r0 + rows);
  unsigned running = before;
  for (int64_t base = r0; base < r1; base += 256) {
    const int64_t p = base + threadIdx.x;
    int32_t o = 0;
    bool f = false;
    if (p < r1) { o = perm_glob[p]; f = keep_orig[o] != 0; }
    unsigned tot;
    const unsigned incl = block_scan_flag(f, &tot);
    if (f) keep[running + incl - 1] = (int64_t)o;
    running += tot;
  }
}

// per-group compaction (batched detector): one workgroup per group
__global__ __launch_bounds__(1024) void k_nms_group_compact(const unsigned long long* __restrict__ key1s,
                                                            const int32_t* __restrict__ perm_glob,
                                                            const uint8_t* __restrict__ keep_orig,
                                                            int64_t n, int32_t max_per_group,
                                                            int32_t* __restrict__ keep,
                                                            int32_t* __restrict__ group_counts) {
  __shared__ int s_wave[16];
  __shared__ int s_total;
  const uint32_t g = blockIdx.x;
  // [lo, hi) = rows of group g in the (group, score) order: binary search on key1 >> 32
  auto lower = [&](unsigned long long gv) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
      int64_t mid = (lo + hi) >> 1;
      if ((key1s[mid] >> 32) < gv) lo = mid + 1; else hi = mid;
    }
    return lo;
  };
  const int64_t lo = lower(g), hi = lower((unsigned long long)g + 1);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int written = 0;
  for (int64_t base = lo; base < hi && written < max_per_group; base += 1024) {
    int64_t p = base + threadIdx.x;
    int32_t o = -1;
    bool f = false;
    if (p < hi) {
      o = perm_glob[p];
      f = keep_orig[o] != 0;
    }
    unsigned long long bal = __ballot(f);
    int rank = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave[wave] = __popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) {
      int acc = 0;
      for (int w = 0; w < 16; w++) {
        int c = s_wave[w];
        s_wave[w] = acc;
        acc += c;
      }
      s_total = acc;
    }
    __syncthreads();
    int pos = written + s_wave[wave] + rank;
    if (f && pos < max_per_group) keep[(int64_t)g * max_per_group + pos] = o;
    written += s_total;
    __syncthreads();
  }
  if (written > max_per_group) written = max_per_group;
  for (int r = written + threadIdx.x; r < max_per_group; r += 1024)
    keep[(int64_t)g * max_per_group + r] = -1;
  if (threadIdx.x == 0) group_counts[g] = written;
}

// per-group compaction + output assembly of multiclass_nms_rotated (utils/bbox_nms_rotated.py:47-64) for the batched
// detector: the r-th kept row of group g (score descending) goes to wire[g][r*7 .. r*7+6] = x, y, w, h, angle, score,
// label; rows behind the last kept one are 0,0,0,0,0,0,-1; wire[g][max_per_group*7] = the count.  One workgroup per
// group; the stock index / cat / where / clamp launches this replaces were ~15 per step.
__global__ __launch_bounds__(1024) void k_nms_group_emit(const unsigned long long* __restrict__ key1s,
                                                         const int32_t* __restrict__ perm_glob,
                                                         const uint8_t* __restrict__ keep_orig,
                                                         const float* __restrict__ dets5,
                                                         const float* __restrict__ scores,
                                                         const int32_t* __restrict__ row_labels, int64_t n,
                                                         int32_t max_per_group, float* __restrict__ wire,
                                                         int32_t* __restrict__ labels_out,
                                                         int32_t* __restrict__ counts_out,
                                                         const long long* __restrict__ cand_found,
                                                         long long* __restrict__ overflow_out,
                                                         long long* __restrict__ dropped_total) {
  __shared__ int s_wave[16];
  __shared__ int s_total;
  const uint32_t g = blockIdx.x;
  if (g == 0 && threadIdx.x == 0 && cand_found) {     // candidates the static row cap n cut (the reference drops none)
    const long long found = *cand_found, dropped = found > (long long)n ? found - (long long)n : 0;
    if (overflow_out) { overflow_out[0] = found; overflow_out[1] = dropped; }
    if (dropped_total) *dropped_total += dropped;      // stream-ordered: one writer per buffer
  }
  auto lower = [&](unsigned long long gv) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
      int64_t mid = (lo + hi) >> 1;
      if ((key1s[mid] >> 32) < gv) lo = mid + 1; else hi = mid;
    }
    return lo;
  };
  const int64_t lo = lower(g), hi = lower((unsigned long long)g + 1);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* wg = wire + (int64_t)g * ((int64_t)max_per_group * 7 + 1);
  int32_t* lg = labels_out ? labels_out + (int64_t)g * max_per_group : nullptr;
  int written = 0;
  for (int64_t base = lo; base < hi && written < max_per_group; base += 1024) {
    int64_t p = base + threadIdx.x;
    int32_t o = -1;
    bool f = false;
    if (p < hi) {
      o = perm_glob[p];
      f = keep_orig[o] != 0;
    }
    unsigned long long bal = __ballot(f);
    int rank = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave[wave] = __popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) {
      int acc = 0;
      for (int w = 0; w < 16; w++) {
        int c = s_wave[w];
        s_wave[w] = acc;
        acc += c;
      }
      s_total = acc;
    }
    __syncthreads();
    int pos = written + s_wave[wave] + rank;
    if (f && pos < max_per_group) {
      const float* d = dets5 + (int64_t)o * 5;
      float* w7 = wg + (int64_t)pos * 7;
      const int lab = row_labels[o];
      w7[0] = d[0]; w7[1] = d[1]; w7[2] = d[2]; w7[3] = d[3]; w7[4] = d[4];
      w7[5] = scores[o];
      w7[6] = (float)lab;
      if (lg) lg[pos] = lab;
    }
    written += s_total;
    __syncthreads();
  }
  if (written > max_per_group) written = max_per_group;
  for (int r = written + threadIdx.x; r < max_per_group; r += 1024) {
    float* w7 = wg + (int64_t)r * 7;
    w7[0] = 0.f; w7[1] = 0.f; w7[2] = 0.f; w7[3] = 0.f; w7[4] = 0.f; w7[5] = 0.f; w7[6] = -1.f;
    if (lg) lg[r] = -1;
  }
  if (threadIdx.x == 0) {
    wg[(int64_t)max_per_group * 7] = (float)written;
    if (counts_out) counts_out[g] = written;
  }
}

// bitonic sort of P (power of two, >= 128) 64-bit keys in LDS, ascending; NT threads; ends with a barrier.  Compare-exchange
// distances up to 64 stay inside an aligned block of 128 keys: those stages run per wave on its own blocks with the two keys
// of a lane in REGISTERS (distance 64 = the lane's own pair, shorter distances by cross-lane exchange) -- no workgroup barrier
// and no LDS round trip per stage (a stage through LDS cost ~300 cycles of latency, 91 of them for 8 192 keys).  Longer
// distances go through LDS with a barrier per stage.  first_k: the length the data is ALREADY sorted to in alternating
// directions (runs of first_k / 2 keys, even runs ascending, odd runs descending), 2 = nothing: merging pre-sorted runs
// skips the levels below it.
__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int m) {
  const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)v, m), hi = (unsigned)__shfl_xor((int)(unsigned)(v >> 32), m);
  return ((unsigned long long)hi << 32) | lo;
}
template <int NT>
__device__ __forceinline__ void lds_bitonic_sort(unsigned long long* __restrict__ s, unsigned P, unsigned first_k = 2) {
  const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  auto cmpx = [&](unsigned t, unsigned j, unsigned k) {
    const unsigned i = ((t & ~(j - 1u)) << 1) | (t & (j - 1u)), l = i | j;
    const unsigned long long a = s[i], b = s[l];
    if ((a > b) == ((i & k) == 0u)) { s[i] = b; s[l] = a; }
  };
  // the stages of level k with distance <= 64 (all of them when k <= 128) on the wave's own 128-key blocks; levels k0 .. k1
  // one level in registers; KC = min(k, 128) at compile time so that the distances >= k vanish (27 exchange stages for a
  // 128-key sort, not 42)
  auto level = [&](auto kc, unsigned k, unsigned base, unsigned long long& lo, unsigned long long& hi) {
    constexpr unsigned KC = decltype(kc)::value;
    const bool asc_lo = ((base + lane) & k) == 0u, asc_hi = ((base + 64 + lane) & k) == 0u;
    if constexpr (KC >= 128) {                                            // distance 64: the lane's own two keys
      if ((lo > hi) == asc_lo) { const unsigned long long t = lo; lo = hi; hi = t; }
    }
#pragma unroll
    for (int j = 32; j >= 1; j >>= 1) {
      if ((unsigned)j >= KC) continue;                                    // (level k starts at distance k / 2)
      const unsigned long long olo = shfl_xor_u64(lo, j), ohi = shfl_xor_u64(hi, j);
      const bool low = (lane & (unsigned)j) == 0u;                        // this lane holds the lower index of its pair
      lo = ((lo > olo) == (low == asc_lo)) ? olo : lo;
      hi = ((hi > ohi) == (low == asc_hi)) ? ohi : hi;
    }
  };
  auto local = [&](unsigned k0, unsigned k1) {
    for (unsigned blk = wave; blk < P / 128; blk += NT / 64) {            // (always the same wave for a block)
      const unsigned base = blk * 128;
      unsigned long long lo = s[base + lane], hi = s[base + 64 + lane];
      if (k0 <= 2u && 2u <= k1) level(std::integral_constant<unsigned, 2>{}, 2u, base, lo, hi);
      if (k0 <= 4u && 4u <= k1) level(std::integral_constant<unsigned, 4>{}, 4u, base, lo, hi);
      if (k0 <= 8u && 8u <= k1) level(std::integral_constant<unsigned, 8>{}, 8u, base, lo, hi);
      if (k0 <= 16u && 16u <= k1) level(std::integral_constant<unsigned, 16>{}, 16u, base, lo, hi);
      if (k0 <= 32u && 32u <= k1) level(std::integral_constant<unsigned, 32>{}, 32u, base, lo, hi);
      if (k0 <= 64u && 64u <= k1) level(std::integral_constant<unsigned, 64>{}, 64u, base, lo, hi);
      for (unsigned k = k0 > 128u ? k0 : 128u; k <= k1; k <<= 1) level(std::integral_constant<unsigned, 128>{}, k, base, lo, hi);
      s[base + lane] = lo;
      s[base + 64 + lane] = hi;
    }
  };
  if (first_k <= 128) local(first_k, 128);
  __syncthreads();
  for (unsigned k = first_k > 256 ? first_k : 256; k <= P; k <<= 1) {
    for (unsigned j = k >> 1; j >= 128; j >>= 1) {
      for (unsigned t = threadIdx.x; t < P / 2; t += NT) cmpx(t, j, k);
      __syncthreads();
    }
    local(k, k);
    __syncthreads();
  }
}

// The same output assembly WITHOUT the (group, score) sort of all rows: one workgroup per group collects the KEPT rows of
// its group by scanning group ids and keep flags (L2-resident), sorts their (~score | row) keys in LDS and writes the first
// max_per_group.  Only the best max_per_group <= kEmitCap / 2 rows can be emitted, so more kept rows than the LDS holds are
// folded in by rounds: the best kEmitCap / 2 so far stay in the lower half, the next kEmitCap / 2 candidates fill the upper
// half, sort, repeat.  Replaces ~10 library launches over the padded candidate buffer (order C) by nothing.
constexpr int kEmitCap = 8192;
__global__ __launch_bounds__(1024) void k_nms_group_emit_scan(const int32_t* __restrict__ group_ids,
                                                              const uint8_t* __restrict__ keep_orig,
                                                              const float* __restrict__ dets5,
                                                              const float* __restrict__ scores,
                                                              const int32_t* __restrict__ row_labels, int64_t n,
                                                              int32_t max_per_group, float* __restrict__ wire,
                                                              int32_t* __restrict__ labels_out,
                                                              int32_t* __restrict__ counts_out,
                                                              const long long* __restrict__ cand_found,
                                                              long long* __restrict__ overflow_out,
                                                              long long* __restrict__ dropped_total) {
  // (cand_found = the producer's candidate count: rows behind it are padding, never kept, and are not read)
  const int64_t nl = seg_row_limit(cand_found, n);
  __shared__ unsigned long long s_key[kEmitCap];
  __shared__ unsigned s_cnt;
  const int32_t g = (int32_t)blockIdx.x;
  const int lane = threadIdx.x & 63;
  if (g == 0 && threadIdx.x == 0 && cand_found) {     // candidates the static row cap n cut (the reference drops none)
    const long long found = *cand_found, dropped = found > (long long)n ? found - (long long)n : 0;
    if (overflow_out) { overflow_out[0] = found; overflow_out[1] = dropped; }
    if (dropped_total) *dropped_total += dropped;      // stream-ordered: one writer per buffer
  }
  constexpr unsigned kHalf = kEmitCap / 2;
  unsigned have = 0;                                   // sorted keys held in s_key[0, have)
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  auto fold = [&]() {                                  // sort what was collected behind `have`, keep the best kHalf
    const unsigned cnt = min(s_cnt, (unsigned)kEmitCap);
    for (unsigned r = have + threadIdx.x; r < cnt; r += 1024) {       // row indices -> (~score | row) keys
      const uint32_t i = (uint32_t)s_key[r];
      s_key[r] = ((unsigned long long)(~float_sortable(scores[i])) << 32) | i;
    }
    unsigned P = 128;
    while (P < cnt) P <<= 1;
    for (unsigned i = cnt + threadIdx.x; i < P; i += 1024) s_key[i] = ~0ull;
    __syncthreads();
    lds_bitonic_sort<1024>(s_key, P);
    have = min(cnt, kHalf);
    if (threadIdx.x == 0) s_cnt = have;
    __syncthreads();
  };
  // a sweep = 4096 rows, four per thread; the eight loads of the NEXT sweep are requested before this one is processed (one
  // sweep per L2 round trip otherwise: 40 round trips of ~2 us for a 160 k-row buffer)
  uint8_t kf[4], kfn[4];
  int32_t gid[4], gidn[4];
  auto request = [&](int64_t base) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int64_t i = base + k * 1024 + threadIdx.x;
      kfn[k] = i < nl ? keep_orig[i] : (uint8_t)0;
      gidn[k] = i < nl ? group_ids[i] : -1;
    }
  };
  request(0);
  for (int64_t base = 0; base < nl; base += 4096) {
#pragma unroll
    for (int k = 0; k < 4; k++) { kf[k] = kfn[k]; gid[k] = gidn[k]; }
    request(base + 4096);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int64_t i = base + k * 1024 + threadIdx.x;
      const bool mine = kf[k] != 0 && gid[k] == g;
      const unsigned long long bal = __ballot(mine);
      if (bal) {
        unsigned slot = 0;
        if (lane == 0) slot = atomicAdd(&s_cnt, (unsigned)__popcll(bal));
        slot = (unsigned)__shfl((int)slot, 0) + (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
        if (mine) s_key[slot] = (unsigned long long)i;      // (row index; the key is made in fold(): see k_seg_sort)
      }
    }
    // fold() leaves <= kHalf keys and a sweep adds <= 4096 = kEmitCap - kHalf: a slot never passes kEmitCap.  The count is
    // read between two barriers: a wave that is already in the next sweep must not move it under a slower wave's decision
    __syncthreads();
    const unsigned c = s_cnt;
    __syncthreads();
    if (c > kHalf) fold();                               // (uniform)
  }
  fold();
  const int written = (int)min(have, (unsigned)max_per_group);
  float* wg = wire + (int64_t)g * ((int64_t)max_per_group * 7 + 1);
  int32_t* lg = labels_out ? labels_out + (int64_t)g * max_per_group : nullptr;
  for (int r = threadIdx.x; r < max_per_group; r += 1024) {
    float* w7 = wg + (int64_t)r * 7;
    if (r < written) {
      const uint32_t o = (uint32_t)s_key[r];
      const float* d = dets5 + (int64_t)o * 5;
      const int lab = row_labels[o];
      w7[0] = d[0]; w7[1] = d[1]; w7[2] = d[2]; w7[3] = d[3]; w7[4] = d[4];
      w7[5] = scores[o];
      w7[6] = (float)lab;
      if (lg) lg[r] = lab;
    } else {
      w7[0] = 0.f; w7[1] = 0.f; w7[2] = 0.f; w7[3] = 0.f; w7[4] = 0.f; w7[5] = 0.f; w7[6] = -1.f;
      if (lg) lg[r] = -1;
    }
  }
  if (threadIdx.x == 0) {
    wg[(int64_t)max_per_group * 7] = (float)written;
    if (counts_out) counts_out[g] = written;
  }
}

__global__ void k_zero_u8(uint8_t* p, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0;
}


// rows in (segment, score) order -- ONE pass behind the sort: segment index of every row (prefix of the block counts +
// in-block scan of the head flags), seg_start[], num_seg, pre-processed boxes (label slot = segment index), initial
// states (rows of the ignored segment are never kept), inverse permutation (original row -> score position)
__global__ __launch_bounds__(256) void k_nms_pos_meta(const float* __restrict__ dets5,
                                                      const unsigned long long* __restrict__ keyA_s,
                                                      const int32_t* __restrict__ perm_seg,
                                                      const uint32_t* __restrict__ cnt, int nb, int rows, int64_t n,
                                                      uint32_t ignore_key, int use_ignore,
                                                      uint32_t* __restrict__ segidx1, uint32_t* __restrict__ seg_start,
                                                      uint32_t* __restrict__ num_seg, PreBox* __restrict__ sorted,
                                                      uint8_t* __restrict__ state, uint8_t* __restrict__ state_fb,
                                                      uint32_t* __restrict__ seg_start_a, uint32_t* __restrict__ num_seg_a) {
  unsigned before, total;
  count_prefix(cnt, nb, blockIdx.x, before, total);
  if (seg_start && blockIdx.x == 0 && threadIdx.x == 0) { *num_seg = total; seg_start[total] = (uint32_t)n; }
  // (spatial path: the B order numbers the segments in its own way; the direct fallback works in THIS order)
  if (seg_start_a && blockIdx.x == 0 && threadIdx.x == 0) { *num_seg_a = total; seg_start_a[total] = (uint32_t)n; }
  const int64_t r0 = (int64_t)blockIdx.x * rows, r1 = min(n, r0 + rows);
  unsigned running = before;
  for (int64_t base = r0; base < r1; base += 256) {
    const int64_t p = base + threadIdx.x;
    uint32_t sk = 0;
    bool head = false;
    if (p < r1) {
      sk = (uint32_t)(keyA_s[p] >> 32);
      head = p == 0 || sk != (uint32_t)(keyA_s[p - 1] >> 32);
    }
    unsigned tot;
    const unsigned incl = block_scan_flag(head, &tot);
    if (p < r1) {
      const uint32_t s = running + incl - 1;
      if (seg_start) {                             // (the spatial path takes segidx / seg_start from the B order)
        segidx1[p] = s + 1;
        if (head) seg_start[s] = (uint32_t)p;
      }
      if (seg_start_a && head) seg_start_a[s] = (uint32_t)p;
      const int32_t o = perm_seg[p];
      const float* b = dets5 + 5 * (int64_t)o;
      PreBox pb = make_prebox(b[0], b[1], b[2], b[3], b[4], __uint_as_float(s));
      sorted[p] = pb;
      const uint8_t st0 = (use_ignore && sk == ignore_key) ? 2 : 0;
      if (state) state[p] = st0;
      state_fb[p] = st0;
    }
    running += tot;
  }
}

// ---- OWN SEGMENT SORT (round 5): the (segment, score) order of a detector batch without a library sort.
// A detector batch is ~120 (image, class) segments of a few hundred rows inside a static-size candidate buffer that is
// mostly padding (160 000 rows for ~40 000 candidates): rocPRIM's 64-bit pair sort over the whole buffer took ~10 launches
// of 5-17 us for order A and as many again for the output order C (0.16 ms per step).  Here:
//   k_seg_hist  rows per segment (LDS histogram per 4096-row chunk, one global atomic per chunk and non-empty segment) and
//               the ignored rows of every chunk;
//   k_seg_sort  ONE workgroup per segment: start = prefix of the counts, collects its rows by scanning the segment ids
//               (a 160 k-row id array is 640 KB from L2: ~2 us per workgroup, all segments side by side), sorts
//               (~score | original row) keys in LDS (bitonic, <= 8192 rows; a larger segment ranks its rows by counting over
//               a global list: slow, exact) and writes everything k_nms_pos_meta wrote: sorted keys, permutation, segment
//               index, pre-processed boxes, states, seg_start.  Extra workgroups place the ignored rows (in row order)
//               behind the last real segment: every row keeps a position of its own (k_nms_finish writes keep_orig through
//               the permutation).
// Segment index = segment id (empty segments exist as empty ranges), the ignored rows form segment S.
constexpr int kSegSortMaxSeg = 2048;      // segments the LDS histogram holds
constexpr int kSegChunk = 4096;           // rows per histogram block / per ignored-row placer
constexpr int kSegCap = 8192;             // rows a segment may have for the LDS sort (64 KB of keys)

__global__ __launch_bounds__(256) void k_seg_hist(const int32_t* __restrict__ seg_ids, int64_t n,
                                                  const long long* __restrict__ row_limit, uint32_t S,
                                                  uint32_t* __restrict__ seg_n, uint32_t* __restrict__ ign_cnt) {
  __shared__ uint32_t s_h[kSegSortMaxSeg + 1];
  for (uint32_t i = threadIdx.x; i <= S; i += 256) s_h[i] = 0u;
  __syncthreads();
  const int64_t nl = seg_row_limit(row_limit, n);
  const int64_t r0 = (int64_t)blockIdx.x * kSegChunk, r1 = min(n, r0 + kSegChunk), rl = min(r1, max(nl, r0));
  for (int64_t p = r0 + threadIdx.x; p < rl; p += 256) {
    const int32_t sid = seg_ids[p];
    atomicAdd(&s_h[(sid < 0 || (uint32_t)sid >= S) ? S : (uint32_t)sid], 1u);     // (ids outside [0, S) are padding by contract)
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < S; i += 256)
    if (s_h[i]) atomicAdd(&seg_n[i], s_h[i]);
  if (threadIdx.x == 0) ign_cnt[blockIdx.x] = s_h[S] + (uint32_t)(r1 - rl);
}

// block-wide sums of two values (1024 threads); every thread gets both totals
__device__ __forceinline__ void block_sum2_1024(unsigned& a, unsigned& b) {
  __shared__ unsigned s_a[16], s_b[16];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { s_a[threadIdx.x >> 6] = a; s_b[threadIdx.x >> 6] = b; }
  __syncthreads();
  a = 0; b = 0;
#pragma unroll
  for (int w = 0; w < 16; w++) { a += s_a[w]; b += s_b[w]; }
}

__global__ __launch_bounds__(1024) void k_seg_sort(const float* __restrict__ dets5, const float* __restrict__ scores,
                                                   const int32_t* __restrict__ seg_ids, int64_t n,
                                                   const long long* __restrict__ row_limit, uint32_t S,
                                                   const uint32_t* __restrict__ seg_n, const uint32_t* __restrict__ ign_cnt,
                                                   int nchunks, uint32_t ignore_key,
                                                   unsigned long long* __restrict__ scratch /* keyA: unsorted lists of big segments */,
                                                   unsigned long long* __restrict__ keyA_s, int32_t* __restrict__ perm_seg,
                                                   uint32_t* __restrict__ segidx1, uint32_t* __restrict__ seg_start,
                                                   uint32_t* __restrict__ num_seg, PreBox* __restrict__ sorted,
                                                   uint8_t* __restrict__ state, uint8_t* __restrict__ state_fb) {
  __shared__ unsigned long long s_key[kSegCap];
  __shared__ unsigned s_cnt;
  const int lane = threadIdx.x & 63;
  const int64_t nl = seg_row_limit(row_limit, n);
  // prefix of the segment counts in front of this workgroup's segment (ignore placers: the total)
  const uint32_t w = blockIdx.x;
  unsigned before = 0, total = 0;
  for (uint32_t j = threadIdx.x; j < S; j += 1024) {
    const unsigned c = seg_n[j];
    total += c;
    if (j < w) before += c;
  }
  block_sum2_1024(before, total);
  if (w == 0 && threadIdx.x == 0) { *num_seg = S + 1u; seg_start[S] = total; seg_start[S + 1u] = (uint32_t)n; }
  auto emit_row = [&](uint32_t p, uint32_t o, uint32_t sidx, unsigned long long key, uint8_t st0) {
    keyA_s[p] = key;
    perm_seg[p] = (int32_t)o;
    segidx1[p] = sidx + 1u;
    const float* b = dets5 + 5 * (int64_t)o;
    sorted[p] = make_prebox(b[0], b[1], b[2], b[3], b[4], __uint_as_float(sidx));
    state[p] = st0;
    state_fb[p] = st0;
  };
  if (w >= S) {
    // ---- ignored rows of chunk (w - S), in row order, behind the real segments
    const int cb = (int)(w - S);
    unsigned ib = 0, dummy = 0;
    for (int j = threadIdx.x; j < cb; j += 1024) ib += ign_cnt[j];
    block_sum2_1024(ib, dummy);
    if (ign_cnt[cb] == 0u) return;                         // (uniform)
    __shared__ unsigned s_wsum[16];
    unsigned running = total + ib;
    const int64_t r0 = (int64_t)cb * kSegChunk, r1 = min(n, r0 + kSegChunk);
    for (int64_t base = r0; base < r1; base += 1024) {
      const int64_t i = base + threadIdx.x;
      bool ign = false;
      if (i < r1) {
        ign = i >= nl;
        if (!ign) { const int32_t sid = seg_ids[i]; ign = sid < 0 || (uint32_t)sid >= S; }
      }
      const unsigned long long bal = __ballot(ign);
      __syncthreads();
      if (lane == 0) s_wsum[threadIdx.x >> 6] = (unsigned)__popcll(bal);
      __syncthreads();
      unsigned wbase = 0, all = 0;
#pragma unroll
      for (int k = 0; k < 16; k++) { if (k < (int)(threadIdx.x >> 6)) wbase += s_wsum[k]; all += s_wsum[k]; }
      if (ign) {       // never compared, never kept: a position of its own and the removed state is all an ignored row needs
        const uint32_t p = running + wbase + (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
        keyA_s[p] = ((unsigned long long)ignore_key << 32) | 0xffffffffull;
        perm_seg[p] = (int32_t)i;
        segidx1[p] = S + 1u;
        PreBox z = {};
        z.label = __uint_as_float(S);
        sorted[p] = z;
        state[p] = 2;
        state_fb[p] = 2;
      }
      running += all;
    }
    return;
  }
  // ---- segment w
  const uint32_t ns = seg_n[w], start = before;
  if (threadIdx.x == 0) { seg_start[w] = start; s_cnt = 0; }
  if (ns == 0u) return;                                    // (uniform)
  const bool in_lds = ns <= (uint32_t)kSegCap;
  __syncthreads();
  // collect: the rows of this segment, any order.  Eight ids per thread are requested before the first is looked at: one id per
  // trip made the loop a chain of 157 dependent L2 round trips (160 k rows), ~150 us per workgroup
  // (and the next eight are requested before these are processed: the loop then runs at the rate of its ballots)
  constexpr int kU = 8;
  int32_t sid[kU], nxt[kU];
  auto request = [&](int64_t base, int32_t (&v)[kU]) {
#pragma unroll
    for (int k = 0; k < kU; k++) {
      const int64_t i = base + k * 1024 + threadIdx.x;
      v[k] = i < nl ? seg_ids[i] : -1;
    }
  };
  request(0, nxt);
  for (int64_t base = 0; base < nl; base += kU * 1024) {
#pragma unroll
    for (int k = 0; k < kU; k++) sid[k] = nxt[k];
    request(base + kU * 1024, nxt);
#pragma unroll
    for (int k = 0; k < kU; k++) {
      const int64_t i = base + k * 1024 + threadIdx.x;
      const bool mine = sid[k] == (int32_t)w;
      const unsigned long long bal = __ballot(mine);
      if (bal) {                                           // (wave-uniform)
        unsigned slot = 0;
        if (lane == 0) slot = atomicAdd(&s_cnt, (unsigned)__popcll(bal));
        slot = (unsigned)__shfl((int)slot, 0) + (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
        // (only the row index here: the score load behind a match would wait for the prefetched ids as well -- the keys are
        // made after the scan, all loads of a workgroup in flight at once)
        if (mine) { if (in_lds) s_key[slot] = (unsigned long long)i; else scratch[start + slot] = (unsigned long long)i; }
      }
    }
  }
  __syncthreads();
  // ascending key = descending score, ties by ascending original row (the order of the stable pair sort it replaces)
  for (uint32_t r = threadIdx.x; r < ns; r += 1024) {
    const uint32_t i = (uint32_t)(in_lds ? s_key[r] : scratch[start + r]);
    const unsigned long long key = ((unsigned long long)(~float_sortable(scores[i])) << 32) | i;
    if (in_lds) s_key[r] = key; else scratch[start + r] = key;
  }
  __syncthreads();
  if (in_lds) {
    unsigned P = 128;
    while (P < ns) P <<= 1;
    for (unsigned i = ns + threadIdx.x; i < P; i += 1024) s_key[i] = ~0ull;
    __syncthreads();
    lds_bitonic_sort<1024>(s_key, P);
    for (uint32_t r = threadIdx.x; r < ns; r += 1024) {
      const unsigned long long key = s_key[r];
      emit_row(start + r, (uint32_t)key, w, ((unsigned long long)w << 32) | (key >> 32), 0);
    }
    return;
  }
  // ---- a segment beyond the LDS capacity: rank of every row = number of smaller keys in the segment's (global) list;
  // tiles of the list go through LDS, every thread carries its rows' ranks.  O(ns^2 / 1024) per thread: slow, exact.
  __threadfence();                                          // the list was written by this workgroup: make it visible to all its waves
  __syncthreads();
  const unsigned long long* L = scratch + start;
  for (uint32_t r0 = 0; r0 < ns; r0 += 4 * 1024) {
    unsigned long long mykey[4];
    unsigned rank[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t r = r0 + k * 1024 + threadIdx.x;
      mykey[k] = r < ns ? L[r] : ~0ull;
    }
    for (uint32_t t0 = 0; t0 < ns; t0 += kSegCap) {
      const uint32_t tn = min((uint32_t)kSegCap, ns - t0);
      __syncthreads();
      for (uint32_t i = threadIdx.x; i < tn; i += 1024) s_key[i] = L[t0 + i];
      __syncthreads();
      for (uint32_t i = 0; i < tn; i++) {
        const unsigned long long v = s_key[i];               // (broadcast read)
#pragma unroll
        for (int k = 0; k < 4; k++) rank[k] += v < mykey[k] ? 1u : 0u;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t r = r0 + k * 1024 + threadIdx.x;
      if (r < ns) emit_row(start + rank[k], (uint32_t)mykey[k], w, ((unsigned long long)w << 32) | (mykey[k] >> 32), 0);
    }
  }
}

// slot of block `blk` of segment s in the block-bounding-box arrays: consecutive segments never collide because
// seg_start[s + 1] >= seg_start[s] + 64 * (nblk - 1) + 1
__device__ __forceinline__ uint32_t block_slot(uint32_t seg_start_s, uint32_t s, uint32_t blk) {
  return (seg_start_s >> 6) + s + blk;
}
__host__ __device__ inline size_t block_slots_for(size_t n) { return n / 64 + n + 2; }

// rows in SPATIAL order (sort B: segment | Morton) -- ONE pass behind that sort, on the critical path of the call: segment
// index of every row, seg_start[], num_seg (prefix of the block counts + in-block scan), the row's pre-processed box, its
// rank key and initial state, and the bounding box of every 64-row block of the inflated circumscribed circles (atomics on order-preserving integers;
// one per wave and slot in the common case).  The start of a row's segment -- needed for its block slot -- is the running
// maximum of the head positions: carried in from the blocks before (lasthead[]) and scanned inside the block.
__global__ __launch_bounds__(256) void k_nms_sp_meta(const float* __restrict__ dets5,
                                                     const unsigned long long* __restrict__ keyB_s,
                                                     const int32_t* __restrict__ perm_sp,
                                                     const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ lasthead,
                                                     int nb, int rows, int64_t n, const float* __restrict__ scores,
                                                     uint32_t ignore_key, int use_ignore,
                                                     uint32_t* __restrict__ segidxq,
                                                     uint32_t* __restrict__ seg_start, uint32_t* __restrict__ num_seg,
                                                     PreBox* __restrict__ sp_box, unsigned long long* __restrict__ rankkey,
                                                     uint8_t* __restrict__ state, uint2* __restrict__ lo,
                                                     uint2* __restrict__ hi) {
  unsigned before, total;
  count_prefix(cnt, nb, blockIdx.x, before, total);
  if (blockIdx.x == 0 && threadIdx.x == 0) { *num_seg = total; seg_start[total] = (uint32_t)n; }
  __shared__ unsigned s_m[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned carry = 0;                            // (position + 1) of the last head before this block's rows
  for (int j = threadIdx.x; j < (int)blockIdx.x; j += 256) carry = max(carry, lasthead[j]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) carry = max(carry, (unsigned)__shfl_xor((int)carry, o));
  __syncthreads();
  if (lane == 0) s_m[wave] = carry;
  __syncthreads();
  carry = max(max(s_m[0], s_m[1]), max(s_m[2], s_m[3]));
  const int64_t r0 = (int64_t)blockIdx.x * rows, r1 = min(n, r0 + rows);
  unsigned running = before;
  for (int64_t base = r0; base < r1; base += 256) {
    const int64_t q = base + threadIdx.x;
    bool head = false;
    if (q < r1) head = q == 0 || (keyB_s[q] >> 20) != (keyB_s[q - 1] >> 20);
    unsigned tot;
    const unsigned incl = block_scan_flag(head, &tot);
    // inclusive max-scan of the head positions (+ 1) over the block
    unsigned m = head ? (unsigned)q + 1u : 0u;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned t = (unsigned)__shfl_up((int)m, o);
      if (lane >= o) m = max(m, t);
    }
    __syncthreads();
    if (lane == 63) s_m[wave] = m;
    __syncthreads();
    unsigned wm = carry;
#pragma unroll
    for (int w = 0; w < 4; w++) wm = w < wave ? max(wm, s_m[w]) : wm;
    m = max(m, wm);
    carry = max(carry, max(max(s_m[0], s_m[1]), max(s_m[2], s_m[3])));
    uint32_t slot = 0xffffffffu, lx = 0xffffffffu, ly = 0xffffffffu, hx = 0, hy = 0;
    if (q < r1) {
      const uint32_t sidx = running + incl - 1, st = m - 1u;
      segidxq[q] = sidx + 1;
      if (head) seg_start[sidx] = (uint32_t)q;
      const int32_t o = perm_sp[q];
      const float* d = dets5 + 5 * (int64_t)o;
      const PreBox b = make_prebox(d[0], d[1], d[2], d[3], d[4], __uint_as_float(sidx));   // label slot = segment index
      sp_box[q] = b;
      // the greedy order of two rows = the order of these keys (descending score, then ascending original row: exactly
      // the order of sort A) -- everything behind the cull works on spatial positions and never needs that sort
      rankkey[q] = ((unsigned long long)(~float_sortable(scores[o])) << 32) | (uint32_t)o;
      state[q] = (use_ignore && (uint32_t)(keyB_s[q] >> 20) == ignore_key) ? 2 : 0;
      slot = block_slot(st, sidx, ((uint32_t)q - st) >> 6);
      // margin of surely_disjoint per box (sum of two of these >= its (ar + br) * 1.002 + 1e-3), plus 1e-3 for the
      // rounding of x -+ r at chip-sized coordinates; non-finite boxes overlap everything (evaluated, never culled)
      const float rr = b.r * 1.002f + 2e-3f;
      float x0 = b.x - rr, x1 = b.x + rr, y0 = b.y - rr, y1 = b.y + rr;
      if (!(isfinite(x0) && isfinite(x1) && isfinite(y0) && isfinite(y1))) {
        x0 = y0 = -__builtin_inff(); x1 = y1 = __builtin_inff();
      }
      lx = float_sortable(x0); ly = float_sortable(y0); hx = float_sortable(x1); hy = float_sortable(y1);
    }
    running += tot;
    // a wave covers at most two slots unless segments are tiny: reduce the lanes of the first lane's slot and of the last
    // lane's slot by shuffles, everybody else (rare) goes alone
    const uint32_t s_first = (uint32_t)__shfl((int)slot, 0), s_last = (uint32_t)__shfl((int)slot, 63);
#pragma unroll
    for (int g = 0; g < 2; g++) {
      const uint32_t sg = g == 0 ? s_first : s_last;
      if (g == 1 && s_last == s_first) break;
      if (sg == 0xffffffffu) continue;
      const bool in = slot == sg;
      uint32_t a = in ? lx : 0xffffffffu, b = in ? ly : 0xffffffffu, c = in ? hx : 0u, d = in ? hy : 0u;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        a = min(a, (uint32_t)__shfl_xor((int)a, o)); b = min(b, (uint32_t)__shfl_xor((int)b, o));
        c = max(c, (uint32_t)__shfl_xor((int)c, o)); d = max(d, (uint32_t)__shfl_xor((int)d, o));
      }
      if (lane == (g == 0 ? 0 : 63)) {
        atomicMin(&lo[sg].x, a); atomicMin(&lo[sg].y, b); atomicMax(&hi[sg].x, c); atomicMax(&hi[sg].y, d);
      }
    }
    if (slot != 0xffffffffu && slot != s_first && slot != s_last) {
      atomicMin(&lo[slot].x, lx); atomicMin(&lo[slot].y, ly); atomicMax(&hi[slot].x, hx); atomicMax(&hi[slot].y, hy);
    }
  }
}

// TILE FILTER: a wave looks at 64 consecutive rows; every row that opens a 64-row block of its segment hands that block's
// row of the upper triangle to the whole wave (lane = column block): tiles whose two block bounding boxes overlap are
// staged per wave and published with one global atomic per workgroup (a flush per wave in between when a stage fills).
// No per-segment tile counts, no scan, no search: the round-2 form needed both and three launches in front of it.
constexpr int kTfStage = 512;
__global__ __launch_bounds__(kThreads) void k_nms_tile_filter(const uint32_t* __restrict__ segidx1,
                                                              const uint32_t* __restrict__ seg_start,
                                                              const unsigned long long* __restrict__ keys, int shift,
                                                              uint32_t ignore_key, int use_ignore, int64_t n,
                                                              const uint2* __restrict__ lo, const uint2* __restrict__ hi,
                                                              TileRef* __restrict__ tiles, NmsCounters* __restrict__ C,
                                                              unsigned long long tile_cap) {
  __shared__ TileRef s_stage[kThreads / 64][kTfStage];
  __shared__ unsigned s_left[kThreads / 64];
  __shared__ unsigned long long s_base;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  TileRef* stage = s_stage[wave];
  unsigned ns = 0;                               // wave-uniform: staged tiles
  auto flush = [&]() {
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(&C->tiles, (unsigned long long)ns);
    base = ((unsigned long long)(uint32_t)__shfl((int)(base >> 32), 0) << 32) | (uint32_t)__shfl((int)(base & 0xffffffffu), 0);
    wave_lds_handoff();
    for (unsigned k = lane; k < ns; k += 64)
      if (base + k < tile_cap) tiles[base + k] = stage[k];
    wave_lds_handoff();
    ns = 0;
  };
  const int64_t p = ((int64_t)blockIdx.x * (kThreads / 64) + wave) * 64 + lane;
  uint32_t s = 0, st = 0, nrows = 0;
  bool opens = false;
  if (p < n && !(use_ignore && (uint32_t)(keys[p] >> shift) == ignore_key)) {   // (the ignored segment has no tiles)
    s = segidx1[p] - 1;
    st = seg_start[s];
    nrows = seg_start[s + 1] - st;
    opens = (((uint32_t)p - st) & 63u) == 0u;
  }
  unsigned long long todo = __ballot(opens);
  while (todo) {
    const int src = __ffsll((long long)todo) - 1;
    todo &= todo - 1ull;
    const uint32_t bs = (uint32_t)__shfl((int)s, src), bst = (uint32_t)__shfl((int)st, src);
    const uint32_t bn = (uint32_t)__shfl((int)nrows, src);
    const uint32_t rb = (((uint32_t)__shfl((int)(uint32_t)p, src)) - bst) >> 6, B = (bn + 63) >> 6;
    uint2 la = make_uint2(0, 0), ha = make_uint2(0, 0);
    if (lo) { const uint32_t a = block_slot(bst, bs, rb); la = lo[a]; ha = hi[a]; }
    for (uint32_t c0 = rb; c0 < B; c0 += 64) {
      const uint32_t cb = c0 + lane;
      bool take = cb < B;
      if (take && lo && cb != rb) {
        const uint32_t b = block_slot(bst, bs, cb);
        const uint2 lb = lo[b], hb = hi[b];
        take = la.x <= hb.x && lb.x <= ha.x && la.y <= hb.y && lb.y <= ha.y;
      }
      const unsigned long long bal = __ballot(take);
      if (take) stage[ns + __popcll(bal & ((1ull << lane) - 1ull))] = TileRef{bst, bn, rb, cb};
      ns += (unsigned)__popcll(bal);
      if (ns + 64 > kTfStage) flush();
    }
  }
  if (lane == 0) s_left[wave] = ns;
  __syncthreads();
  unsigned before = 0, all = 0;
#pragma unroll
  for (int w = 0; w < kThreads / 64; w++) {
    if (w < wave) before += s_left[w];
    all += s_left[w];
  }
  if (all == 0) return;                // uniform
  if (threadIdx.x == 0) s_base = atomicAdd(&C->tiles, (unsigned long long)all);
  __syncthreads();
  const unsigned long long base = s_base + before;
  for (unsigned k = lane; k < ns; k += 64)
    if (base + k < tile_cap) tiles[base + k] = stage[k];
}

// CULL over the tile list (persistent grid, every workgroup a contiguous run of list entries).
// thread = (row = tid & 63, 16 columns).  Stage 1: circle test of every pair of the tile against the column boxes in
// LDS, survivors -> LDS list.  Stage 2, dense: every lane takes ONE survivor: separating axes + IoU upper bound
// (nms_pair_skippable).  What is left goes to the pair list as (score position of the higher-scored box, of the lower).
// The boxes of the next tile are fetched while the current one is tested (double-buffered LDS).
__global__ __launch_bounds__(kThreads) void k_nms_cull(const PreBox* __restrict__ sp_box,
                                                       const TileRef* __restrict__ tiles,
                                                       NmsCounters* __restrict__ C, unsigned long long tile_cap,
                                                       uint2* __restrict__ gq, unsigned long long cap, float thr) {
  __shared__ uint2 s_q[kLdsQueue];
  __shared__ unsigned s_count, s_base[2], s_cnt1[2];
  __shared__ PreBox s_col[2][64];
  __shared__ PreBox s_row[2][64];
  __shared__ unsigned short s_q1[4096];        // circle-test survivors of the current tile: row << 6 | column
  PairQueue Q{s_q, &s_count, s_base};
  if (threadIdx.x == 0) { s_count = 0; s_cnt1[0] = 0; s_cnt1[1] = 0; }
  unsigned long long* gcount = &C->pairs;
  const unsigned long long L = min(C->tiles, tile_cap);   // (an overflowing tile list sets the status bit: fallback)
  const unsigned long long chunk = (L + gridDim.x - 1) / gridDim.x;
  unsigned long long e = (unsigned long long)blockIdx.x * chunk;
  const unsigned long long eend = min(L, e + chunk);
  const int row = threadIdx.x & 63, quarter = threadIdx.x >> 6;
  if (e >= eend) { __syncthreads(); queue_flush(Q, gq, gcount, cap); return; }
  auto fetch = [&](const TileRef& t) {          // threads 0-63: column box, 64-127: row box of tile t
    PreBox b = {};
    if (threadIdx.x < 128) {
      const uint32_t l = (threadIdx.x < 64 ? t.cb : t.rb) * 64 + (threadIdx.x & 63);
      if (l < t.ns) b = sp_box[t.seg_start + l];
    }
    return b;
  };
  TileRef t = tiles[e];
  {
    const PreBox b = fetch(t);
    if (threadIdx.x < 64) s_col[0][threadIdx.x] = b; else if (threadIdx.x < 128) s_row[0][threadIdx.x - 64] = b;
  }
  __syncthreads();
  int cur = 0;
  for (; e < eend; e++) {
    const bool has_next = e + 1 < eend;
    TileRef tn = t;
    PreBox nb = {};
    if (has_next) { tn = tiles[e + 1]; nb = fetch(tn); }     // in flight under this tile's tests
    const uint32_t il = t.rb * 64 + row;
    const int par = (int)(e & 1);
    if (il < t.ns) {
      const PreBox a = s_row[cur][row];
#pragma unroll
      for (int c = 0; c < 16; c++) {
        const int cc = quarter * 16 + c;
        const uint32_t jl = t.cb * 64 + cc;
        if (jl < t.ns && (t.cb != t.rb || jl > il) &&
            !surely_disjoint(a.x, a.y, a.r, s_col[cur][cc].x, s_col[cur][cc].y, s_col[cur][cc].r)) {
          const unsigned p = atomicAdd(&s_cnt1[par], 1u);
          s_q1[p] = (unsigned short)((row << 6) | cc);
        }
      }
    }
    if (has_next) {
      if (threadIdx.x < 64) s_col[cur ^ 1][threadIdx.x] = nb; else if (threadIdx.x < 128) s_row[cur ^ 1][threadIdx.x - 64] = nb;
    }
    __syncthreads();                                    // survivor list and the next tile's boxes are in place
    if (threadIdx.x == 0) s_cnt1[par ^ 1] = 0;          // the next tile's counter (nobody touches it in this phase)
    const unsigned cnt1 = s_cnt1[par];                  // stable: no pushes to the list during the dense stage
    for (unsigned e0 = 0; e0 < cnt1; e0 += kThreads) {  // uniform trip count
      // (uniform decision through the barrier's OR: a plain read of s_count after a barrier can differ between waves)
      if (__syncthreads_or(s_count > kFlushAt)) queue_flush(Q, gq, gcount, cap);
      const unsigned x = e0 + threadIdx.x;
      if (x < cnt1) {
        const unsigned v = s_q1[x], r = v >> 6, cc = v & 63u;
        const PreBox& A = s_row[cur][r];
        const PreBox& B = s_col[cur][cc];
        if (!nms_pair_skippable(A, B, thr))          // pair = the two POSITIONS in the cull's row order (row block first)
          queue_push(Q, t.seg_start + t.rb * 64 + r, t.seg_start + t.cb * 64 + cc);
      }
    }
    __syncthreads();                                    // dense stage done: lists and the current buffers may be reused
    cur ^= 1;
    t = tn;
  }
  queue_flush(Q, gq, gcount, cap);
}

// The same cull with wave-private tile runs and no workgroup barrier (the structure of k_iou_cull_lanes): lane = row of
// the tile's row block AND column of its column block, both boxes in registers (the next tile's are fetched one tile
// ahead); column j's circle reaches all lanes through v_readlane; verdicts of 32 columns in a per-lane mask; every lane
// reserves its list space with ONE LDS atomic per half tile and writes its survivors; the dense stage takes 64
// survivors at a time, both boxes through ds_bpermute from the lanes that own them, nms_pair_skippable, ballot-compacted
// into a per-wave stage of pairs; one global atomic per flush and one per workgroup at the end.
#ifdef S2A_MEASURE
__device__ unsigned long long g_cull_dbg[8];   // measurement builds: cycles per phase summed over waves (+ tile / wave counts)
#define CULL_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define CULL_ACC(acc, t1, t0) (acc) += (t1) - (t0)
#else
#define CULL_T(var) do {} while (0)
#define CULL_ACC(acc, t1, t0) do {} while (0)
#endif
constexpr int kNlList = 64 * 32;              // circle-test survivors of half a tile, u16 = row << 6 | column
constexpr int kNlStage = 512;                 // per-wave staged pairs
constexpr int64_t kNmsLanesRows = 49152;      // rows from which the one-wave-per-tile cull is the default
__global__ __launch_bounds__(kThreads) void k_nms_cull_lanes(const PreBox* __restrict__ sp_box,
                                                             const TileRef* __restrict__ tiles,
                                                             NmsCounters* __restrict__ C, unsigned long long tile_cap,
                                                             uint2* __restrict__ gq, unsigned long long cap, float thr) {
  __shared__ unsigned short s_list[kThreads / 64][kNlList];
  __shared__ uint2 s_stage[kThreads / 64][kNlStage];
  __shared__ unsigned s_n[kThreads / 64], s_left[kThreads / 64];
  __shared__ unsigned long long s_base;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long* gcount = &C->pairs;
  const unsigned long long L = min(C->tiles, tile_cap);   // (an overflowing tile list sets the status bit: fallback)
  unsigned short* list = s_list[wave];
  uint2* stage = s_stage[wave];
  unsigned ns = 0;                               // wave-uniform: staged pairs
  auto flush = [&]() {
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(gcount, (unsigned long long)ns);
    base = ((unsigned long long)(uint32_t)__shfl((int)(base >> 32), 0) << 32) | (uint32_t)__shfl((int)(base & 0xffffffffu), 0);
    wave_lds_handoff();                          // other lanes' staged pairs
    for (unsigned k = lane; k < ns; k += 64)
      if (base + k < cap) gq[base + k] = stage[k];
    wave_lds_handoff();                          // ... are read before the stage is written again
    ns = 0;
  };
  auto load = [&](const TileRef& t, PreBox& R, PreBox& Cc) {
    const uint32_t il = t.rb * 64 + lane, jl = t.cb * 64 + lane;
    R = PreBox{};
    Cc = PreBox{};
    if (il < t.ns) R = sp_box[t.seg_start + il];
    if (jl < t.ns) Cc = sp_box[t.seg_start + jl];
  };
  // First stage = circumscribed circles AND the area ratio: IoU <= min(area) / max(area), so a pair whose areas differ by
  // more than 1 / (0.99 thr) can never suppress (the same 1 % margin and the same sanity conditions as the bound in
  // nms_pair_skippable, which contains this one) -- |log2 area_a - log2 area_c| > -log2(0.99 thr), one subtraction and one
  // compare per pair with the column's log-area rotating along; a box outside the sanity conditions carries NaN and
  // is never dropped.  On BASELINE config 5 it drops 58 % of the circle-test survivors before the 130-instruction second stage.
  const float t99 = 0.99f * thr;
  const float Lmax = thr > 0.05f ? -__log2f(t99) : __builtin_inff();
  auto log_area = [](const PreBox& b) {
    const float aw = fabsf(b.w), ah = fabsf(b.h), area = b.w * b.h;
    const bool sane = area > 0.f && fminf(aw, ah) >= 0.05f * fmaxf(aw, ah);
    return sane ? __log2f(area) : __builtin_nanf("");
  };
  // direction of the lane rotation: the lane whose value arrives here after one step, relative to this lane (1 or 63)
  const unsigned dir = ((unsigned)__builtin_amdgcn_mov_dpp(lane, 0x13C, 0xf, 0xf, true) - (unsigned)lane) & 63u;
  unsigned long long a_claim = 0, a_s1 = 0, a_list = 0, a_s2 = 0, a_tiles = 0;
  (void)a_claim; (void)a_s1; (void)a_list; (void)a_s2; (void)a_tiles;
  CULL_T(t_begin);
  // ---- the wave's stream of tiles: wave w takes tiles w, w + W, w + 2 W ... of a grid with MANY more workgroups than fit
  // on the chip (8 k workgroups: one or two tiles per wave at 200 k rows, at most one in a detector batch).  Tiles differ
  // by an order of magnitude in cost; the hardware dispatcher starts the next workgroup wherever one retires, which
  // balances the load without any cursor (a persistent grid claiming batches from atomic cursors spent half of its wave
  // time waiting for claims and for the loads chained behind them: 32 k probing atomics at the end of every launch).
  // Tile references are fetched two tiles ahead, boxes one tile ahead: an in-order wave blocks at the FIRST use of a load.
  const unsigned long long nwaves = (unsigned long long)gridDim.x * (kThreads / 64);
  unsigned long long tnext = (unsigned long long)blockIdx.x * (kThreads / 64) + wave;
  constexpr unsigned long long kDone = ~0ull;
  auto next_tile = [&]() -> unsigned long long {
    const unsigned long long r = tnext < L ? tnext : kDone;
    tnext += nwaves;
    return r;
  };
  unsigned long long i0 = next_tile();
  TileRef t = {}, t1 = {};
  PreBox A = {}, Cb = {};
  if (i0 != kDone) { t = tiles[i0]; load(t, A, Cb); }
  unsigned long long i1 = i0 != kDone ? next_tile() : kDone;
  if (i1 != kDone) t1 = tiles[i1];
  while (i0 != kDone) {
    {
#ifdef S2A_MEASURE
      unsigned long long tc0 = __builtin_amdgcn_s_memtime();
#endif
      const unsigned long long i2 = i1 != kDone ? next_tile() : kDone;
      TileRef t2 = {};
      if (i2 != kDone) t2 = tiles[i2];           // two tiles ahead: not waited for in this iteration
      PreBox An = {}, Cn = {};
      if (i1 != kDone) load(t1, An, Cn);         // one tile ahead (t1 arrived during the previous tile)
      const TileRef tn = t1;
      const unsigned long long in1 = i1;
      i1 = i2;
      t1 = t2;
#ifdef S2A_MEASURE
      a_tiles++;
#endif
      CULL_T(tc1);
      CULL_ACC(a_claim, tc1, tc0);
      const bool rowvalid = t.rb * 64 + lane < t.ns;
      const unsigned ncol = min(64u, t.ns - t.cb * 64);        // valid columns of the tile (the tile exists: >= 1)
      const bool diag = t.cb == t.rb;                          // upper triangle of a diagonal tile: column > row lane
      const float ax = A.x, ay = A.y, ar = A.r * 1.002f + 1e-3f;
      const float rx = Cb.x, ry = Cb.y, rr = Cb.r * 1.002f, rl = log_area(Cb);   // rotate through the lanes, a step per test
      const float la = log_area(A);
      unsigned bits_h[2] = {0u, 0u};
      {
        CULL_T(ts0);
        nms_stage1_rot2(ax, ay, ar, la, rx, ry, rr, rl, Lmax, bits_h[0], bits_h[1]);   // both halves of the tile at once
        CULL_T(ts1);
        CULL_ACC(a_s1, ts1, ts0);
      }
#pragma unroll
      for (int half = 0; half < 2; half++) {
        unsigned bits = rowvalid ? bits_h[half] : 0u;
        CULL_T(ts1);
        if (lane == 0) s_n[wave] = 0;
        wave_lds_handoff();                                      // the reset, then the reservations
        // bit b of this half = column (lane + (32 half + b) dir) & 63; columns beyond the segment and the lower triangle
        // of a diagonal tile are dropped here, where only the survivors are looked at
        unsigned keepbits = 0;
        for (unsigned w = bits; w;) {
          const unsigned b = (unsigned)__ffs((int)w) - 1u;
          w &= w - 1u;
          const unsigned c = ((unsigned)lane + (32u * half + b) * dir) & 63u;
          if (c < ncol && (!diag || c > (unsigned)lane)) keepbits |= 1u << b;
        }
        bits = keepbits;
        const unsigned cnt = (unsigned)__popc(bits);
        unsigned off = 0;
        if (cnt) off = atomicAdd(&s_n[wave], cnt);             // one reservation per lane (in order behind the reset)
        while (bits) {
          const unsigned b = (unsigned)__ffs((int)bits) - 1u;
          bits &= bits - 1u;
          const unsigned c = ((unsigned)lane + (32u * half + b) * dir) & 63u;
          list[off++] = (unsigned short)(((unsigned)lane << 6) | c);
        }
        wave_lds_handoff();                                      // every lane's list entries and the final count
        unsigned n1 = (unsigned)__builtin_amdgcn_readfirstlane((int)s_n[wave]);
        CULL_T(ts2);
        CULL_ACC(a_list, ts2, ts1);
        while (n1 > 0u) {                                        // the whole half tile: the boxes change with the tile
          const unsigned g = min(n1, 64u), base = n1 - g;
          const bool mine = (unsigned)lane < g;
          const unsigned v = mine ? (unsigned)list[base + lane] : 0u;
          const int r = (int)(v >> 6), c = (int)(v & 63u);
          PreBox R, B;
          R.x = __shfl(A.x, r); R.y = __shfl(A.y, r); R.w = __shfl(A.w, r); R.h = __shfl(A.h, r);
          R.c2 = __shfl(A.c2, r); R.s2 = __shfl(A.s2, r); R.r = 0.f; R.label = 0.f;
          B.x = __shfl(Cb.x, c); B.y = __shfl(Cb.y, c); B.w = __shfl(Cb.w, c); B.h = __shfl(Cb.h, c);
          B.c2 = __shfl(Cb.c2, c); B.s2 = __shfl(Cb.s2, c); B.r = 0.f; B.label = 0.f;
          const bool keep = mine && !nms_pair_skippable(R, B, thr);
          const unsigned long long bal = __ballot(keep);
          if (keep)                              // pair = the two POSITIONS in the cull's row order
            stage[ns + __popcll(bal & ((1ull << lane) - 1ull))] =
                make_uint2(t.seg_start + t.rb * 64 + (unsigned)r, t.seg_start + t.cb * 64 + (unsigned)c);
          ns += (unsigned)__popcll(bal);
          n1 = base;
          if (ns + 64 > kNlStage) flush();
        }
        wave_lds_handoff();                                      // list read out before the next half tile rewrites it
        CULL_T(ts3);
        CULL_ACC(a_s2, ts3, ts2);
      }
      t = tn;
      A = An;
      Cb = Cn;
      i0 = in1;
    }
  }
#ifdef S2A_MEASURE
  if (lane == 0) {
    atomicAdd(&g_cull_dbg[0], __builtin_amdgcn_s_memtime() - t_begin);
    atomicAdd(&g_cull_dbg[1], a_claim); atomicAdd(&g_cull_dbg[2], a_s1); atomicAdd(&g_cull_dbg[3], a_list);
    atomicAdd(&g_cull_dbg[4], a_s2); atomicAdd(&g_cull_dbg[5], a_tiles); atomicAdd(&g_cull_dbg[6], 1ull);
  }
#endif
  if (lane == 0) s_left[wave] = ns;
  __syncthreads();
  unsigned before = 0, all = 0;
#pragma unroll
  for (int w = 0; w < kThreads / 64; w++) {
    if (w < wave) before += s_left[w];
    all += s_left[w];
  }
  if (all == 0) return;                // uniform
  if (threadIdx.x == 0) s_base = atomicAdd(gcount, (unsigned long long)all);
  __syncthreads();
  const unsigned long long base = s_base + before;
  for (unsigned k = lane; k < ns; k += 64)
    if (base + k < cap) gq[base + k] = stage[k];
}

// DENSE IoU pass: one pair per lane; pairs above the threshold become edges.  Every wave stages its edges in a private
// LDS buffer and publishes them with ONE global atomic per flush (an atomic per wave and sweep -- ~20 k on one address at
// 200 k rows -- made the pass atomic-bound); a wave is synchronous, so the staging needs no barrier.
constexpr int kEdgeStage = 128;      // staged edges per wave
// REDO = false: every lane has 8 candidate-point slots (see k_iou_heavy); a pair that needs more is marked in place (top
// bit of its first index -- positions are < 2^31) and REDO = true, the second launch, evaluates the marked pairs with 24.
constexpr uint32_t kPairRedo = 0x80000000u;
template <bool REDO>
__global__ __launch_bounds__(kThreads) void k_nms_heavy(const PreBox* __restrict__ boxes,
                                                        const unsigned long long* __restrict__ rankkey, float thr,
                                                        uint2* __restrict__ gq, NmsCounters* __restrict__ C,
                                                        unsigned long long cap, uint2* __restrict__ edges,
                                                        unsigned long long ecap) {
  constexpr int CAP = REDO ? 24 : kIouCap;
  __shared__ float2 s_pts[CAP * kThreads];
  __shared__ uint2 s_stage[kThreads / 64][kEdgeStage];
  const unsigned long long total = min(C->pairs, cap);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint2* stage = s_stage[wave];
  unsigned staged = 0;                 // wave-uniform
  auto flush = [&]() {
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(&C->edges, (unsigned long long)staged);
    base = ((unsigned long long)(uint32_t)__shfl((int)(base >> 32), 0) << 32) | (uint32_t)__shfl((int)(base & 0xffffffffu), 0);
    for (unsigned k = lane; k < staged; k += 64)
      if (base + k < ecap) edges[base + k] = stage[k];
    staged = 0;
  };
  const unsigned long long stride = (unsigned long long)gridDim.x * kThreads;
  for (unsigned long long e0 = (unsigned long long)blockIdx.x * kThreads; e0 < total; e0 += stride) {
    const unsigned long long e = e0 + threadIdx.x;
    bool hit = false;
    uint2 ij = make_uint2(0, 0);
    if (e < total) {
      // the pair list holds positions in the cull's row order (boxes[]); which of the two comes first in the greedy order
      // is decided by the rank keys (spatial order) or by the positions themselves (blocks in score order)
      uint2 q = gq[e];
      const bool marked = (q.x & kPairRedo) != 0u;
      q.x &= ~kPairRedo;
      if (!REDO || marked) {
        const bool swap = rankkey ? rankkey[q.y] < rankkey[q.x] : q.y < q.x;
        ij = swap ? make_uint2(q.y, q.x) : q;
        const PreBox A = boxes[ij.x];   // higher score first: same argument order as the reference
        const PreBox B = boxes[ij.y];
        if (REDO) {
          hit = rbox_iou<kThreads>(A, B, s_pts + threadIdx.x) > thr;
        } else {
          bool redo = false;
          hit = rbox_iou<kThreads, CAP>(A, B, s_pts + threadIdx.x, &redo) > thr;  // reference GPU rule: strict (ml_nms cuda.cu:63-64)
          if (redo) gq[e].x |= kPairRedo;
        }
      }
    }
    if (REDO && !__any(hit)) continue;    // (wave-uniform; the common case of the second launch)
    const unsigned long long bal = __ballot(hit);
    if (hit) stage[staged + __popcll(bal & ((1ull << lane) - 1ull))] = ij;
    staged += (unsigned)__popcll(bal);
    if (staged + 64 > kEdgeStage) flush();
  }
  // what is left in the four stages goes out with ONE atomic per workgroup (a final flush per wave was 8 k atomics on
  // one address at 2048 workgroups: +25 us)
  __shared__ unsigned s_left[kThreads / 64];
  __shared__ unsigned long long s_base;
  if (lane == 0) s_left[wave] = staged;
  __syncthreads();
  unsigned before = 0, all = 0;
#pragma unroll
  for (int w = 0; w < kThreads / 64; w++) {
    if (w < wave) before += s_left[w];
    all += s_left[w];
  }
  if (all == 0) return;                // uniform
  if (threadIdx.x == 0) s_base = atomicAdd(&C->edges, (unsigned long long)all);
  __syncthreads();
  const unsigned long long base = s_base + before;
  for (unsigned k = lane; k < staged; k += 64)
    if (base + k < ecap) edges[base + k] = stage[k];
}

// ---------------------------------------------------------------- greedy order by rounds over the edge list
// state[p]: 0 open, 1 kept, 2 removed (sticky).  A row that is still open is UNSETTLED exactly when an edge from another
// unsettled row blocked it in the latest round (blocked[r & 1][p] == r); open and not blocked == kept (all its
// in-neighbours are removed).  Round r reads the round r-1 view and writes the round r one; the two blocked arrays
// alternate so that a round never overwrites the marks it reads.  Rows that nobody blocks any more stay kept without
// ever being touched again, so nothing has to be materialised between rounds.
constexpr int kNmsRounds = 4;       // launched rounds over ALL edges (each leaves ~10 x fewer edges alive); what is still alive
                                    // then is bucketed by segment and finished inside one workgroup per segment
enum : uint32_t { kOpen = 0, kKept = 1, kRemoved = 2 };

__device__ __forceinline__ uint32_t nms_view(const uint8_t* __restrict__ state, const uint8_t* __restrict__ blocked,
                                             int64_t n, int r, uint32_t p) {
  const uint32_t st = state[p];
  if (st != kOpen) return st;
  if (r == 0) return kOpen;                                   // before the first round every row is unsettled
  return blocked[(int64_t)(r & 1) * n + p] == (uint8_t)r ? kOpen : kKept;
}

// one round over all edges; r = 1 .. kNmsRounds.  alive[r] = edges between two rows that are unsettled after this round.
// The last launched round also lists those edges for the clean-up kernel.
__global__ __launch_bounds__(kThreads) void k_nms_round(const uint2* __restrict__ edges, NmsCounters* __restrict__ C,
                                                        unsigned long long ecap, uint8_t* __restrict__ state,
                                                        uint8_t* __restrict__ blocked, int64_t n, int r,
                                                        uint2* __restrict__ alive_list, unsigned long long alive_cap) {
  if (r > 1 && C->alive[r - 1] == 0) return;                  // settled already (uniform: before any barrier)
  const unsigned long long E = min(C->edges, ecap);
  unsigned cnt = 0;
  const int lane = threadIdx.x & 63;
  const unsigned long long stride = (unsigned long long)gridDim.x * kThreads;
  for (unsigned long long e0 = (unsigned long long)blockIdx.x * kThreads; e0 < E; e0 += stride) {
    const unsigned long long e = e0 + threadIdx.x;
    bool live = false;
    uint2 ij = make_uint2(0, 0);
    if (e < E) {
      ij = edges[e];
      const uint32_t sj = nms_view(state, blocked, n, r - 1, ij.y);
      if (sj == kOpen) {
        const uint32_t si = nms_view(state, blocked, n, r - 1, ij.x);
        if (si == kKept) state[ij.y] = (uint8_t)kRemoved;
        else if (si == kOpen) { blocked[(int64_t)(r & 1) * n + ij.y] = (uint8_t)r; live = true; }
      }
    }
    cnt += live ? 1u : 0u;
    if (alive_list) {
      const unsigned long long bal = __ballot(live);
      if (bal) {
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(&C->alive_list, (unsigned long long)__popcll(bal));
        base = ((unsigned long long)(uint32_t)__shfl((int)(base >> 32), 0) << 32) | (uint32_t)__shfl((int)(base & 0xffffffffu), 0);
        if (live) {
          const unsigned long long dst = base + __popcll(bal & ((1ull << lane) - 1ull));
          if (dst < alive_cap) alive_list[dst] = ij;
        }
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
  __shared__ unsigned s_cnt[kThreads / 64];
  if (lane == 0) s_cnt[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {               // one atomic per workgroup
    unsigned tot = 0;
    for (int w = 0; w < kThreads / 64; w++) tot += s_cnt[w];
    if (tot) atomicAdd(&C->alive[r], tot);
  }
}

// (the implicit "open and not blocked == kept" view after the last launched round is written out as explicit states by
// k_nms_finish_segments, only when edges are still alive; the two `blocked` arrays become plain "blocked this round" flags)
static_assert(kNmsRounds % 2 == 0 && kNmsRounds < 15, "the materialising pass of k_nms_finish_segments / NmsCounters::alive assume this");

// ---- FINISH per segment: chains longer than the launched rounds (dense detector outputs: tens of dependent rounds).
// Every segment that still has alive edges (picked out of the alive list by its workgroup, see below) is
// finished by ONE workgroup with the rows' states, the round flags and the edge list in LDS, so that a round costs two
// short passes and two barriers instead of a kernel launch.  Explicit states (the segment's workgroup wrote them out first): a round is
//   pass 1: kept source -> target removed; open source and open target -> target flagged;
//   pass 2: open and not flagged -> kept; edges whose target is still open stay (also when their source was removed
//           in pass 1: the target is then decided next round), the others are dropped; the flags of the other array are
//           cleared for the rows that stay open.
// Every open row always has an edge in its segment's list, so it is reached; the loop ends when the list is empty.
// Segments too large for LDS (> kSegRows rows or > kSegEdges alive edges) run the same loop on the global arrays.
// (Round 5: the bucketing of the alive edges by segment -- count, scan, scatter: three launches, 25 us of a detector step
// and of the drop-in op's launch-paced tail -- is gone: positions of a segment are contiguous, so the segment's workgroup picks
// its edges out of the alive list by a range test on the source position while it scans the list once, and writes out the
// implicit "open and not blocked == kept" view of its own rows first.)
constexpr int kSegRows = 8192, kSegEdges = 4096;     // 8 + 16 KB of states / flags, 2 x 32 KB of edges
#ifdef S2A_MEASURE
__device__ unsigned long long g_fin_dbg[16];   // clean-up kernel: max iterations, max edges, sum iterations, segments with work, max rows
#endif

// the rounds of one segment.  LDS: states / flags / both edge lists are the workgroup's shared arrays (the pointers arrive
// here as plain pointers, but after inlining their origin is unambiguous and they compile to ds_* instructions -- picked by a
// run-time `lds ? shared : global` they were FLAT accesses, three dependent ones per edge at ~3x the latency: 1.4 us a round);
// global: the same loop on the call's arrays, one edge list (every round reads all edges again).
#ifndef S2A_FIN_THREADS
#define S2A_FIN_THREADS 1024
#endif
constexpr int kFinThreads = S2A_FIN_THREADS;
template <bool LDS>
__device__ __forceinline__ unsigned finish_rounds(uint8_t* __restrict__ St, uint8_t* __restrict__ Fl, int64_t fstride,
                                                  uint2* __restrict__ Ed, int estride, uint32_t cnt0,
                                                  unsigned* __restrict__ s_n) {
  uint32_t cnt = cnt0;
  int cur = 0, f = 0;
  unsigned iters = 0;
  bool done = false;
#ifdef S2A_MEASURE
  unsigned long long t0 = 0, tp1 = 0, tb1 = 0, tp2 = 0, tb2 = 0;
#define FIN_TIC() t0 = __builtin_amdgcn_s_memtime()
#define FIN_TOC(acc) do { const unsigned long long t1 = __builtin_amdgcn_s_memtime(); acc += t1 - t0; t0 = t1; } while (0)
#else
#define FIN_TIC() do {} while (0)
#define FIN_TOC(acc) do {} while (0)
#endif
  // ---- (A) list rounds: two lists in turn, the surviving edges compacted every round (that is what retires waves early)
  for (;;) {
    iters++;
    FIN_TIC();
    const uint2* Ein = Ed + (LDS ? cur * estride : 0);
    uint8_t* Ff = Fl + f * fstride;
    uint8_t* Fo = Fl + (f ^ 1) * fstride;
    for (uint32_t e = threadIdx.x; e < cnt; e += kFinThreads) {        // pass 1
      const uint2 ij = Ein[e];
      const uint32_t sj = St[ij.y], si = St[ij.x];             // (both requested at once: one LDS round trip, not two)
      if (sj == kOpen) {
        if (si == kKept) St[ij.y] = (uint8_t)kRemoved;
        else if (si == kOpen) Ff[ij.y] = 1;
      }
    }
    if (threadIdx.x == 0) s_n[(iters + 1) % 3] = 0;             // the counter of the NEXT round: last read at the end of round
                                                                // iters - 2, two barriers ago (three counters in rotation)
    FIN_TOC(tp1);
    __threadfence_block();
    __syncthreads();
    FIN_TOC(tb1);
    uint2* Eout = LDS ? Ed + (cur ^ 1) * estride : nullptr;
    unsigned* ctr = s_n + iters % 3;
    for (uint32_t e = threadIdx.x; e < cnt; e += kFinThreads) {        // pass 2
      const uint2 ij = Ein[e];
      const uint32_t sj = St[ij.y], fj = Ff[ij.y];
      if (sj == kOpen) {
        if (fj == 0) {
          St[ij.y] = (uint8_t)kKept;
        } else {
          Fo[ij.y] = 0;
          // (one LDS atomic per surviving edge: a ballot + one atomic per wave measured SLOWER, 57 against 46 us in the detector)
          const unsigned pos = atomicAdd(ctr, 1u);
          if (LDS) Eout[pos] = ij;
        }
      }
    }
    FIN_TOC(tp2);
    __threadfence_block();
    __syncthreads();
    const uint32_t live = *ctr;
    FIN_TOC(tb2);
    if (live == 0) { done = true; break; }
    if (LDS) { cnt = live; cur ^= 1; }                        // (global path: no second list, all edges again)
    f ^= 1;
#ifdef S2A_MEASURE
    if (threadIdx.x == 0 && cnt0 > 1500) atomicAdd(&g_fin_dbg[11], 1ull);
#endif
    if (LDS && cnt <= (uint32_t)kFinThreads) break;           // (uniform) one edge per thread from here on
  }
  // ---- (B) the list fits one edge per thread (the list shrinks ~15 % a round): every thread keeps ITS edge in registers until
  // the edge's target is decided -- no list, no compaction, one LDS round trip per pass; waves whose edges are all dead only
  // take the barriers.  (The same form with four edges per thread from the start measured SLOWER than the lists: 88 us --
  // sixteen waves issuing the full pass every round.)
  if (LDS && !done) {
    bool lv = threadIdx.x < cnt;
    uint2 ij = make_uint2(0, 0);
    if (lv) ij = Ed[cur * estride + threadIdx.x];
    for (;;) {
      iters++;
      FIN_TIC();
      uint8_t* Ff = Fl + f * fstride;
      uint8_t* Fo = Fl + (f ^ 1) * fstride;
      if (lv) {                                                         // pass 1
        const uint32_t sj = St[ij.y], si = St[ij.x];
        if (sj == kOpen) {
          if (si == kKept) St[ij.y] = (uint8_t)kRemoved;
          else if (si == kOpen) Ff[ij.y] = 1;
        } else {
          lv = false;
        }
      }
      if (threadIdx.x == 0) s_n[(iters + 1) % 3] = 0;
      FIN_TOC(tp1);
      __threadfence_block();
      __syncthreads();
      FIN_TOC(tb1);
      bool stays = false;
      if (lv) {                                                         // pass 2
        const uint32_t sj = St[ij.y], fj = Ff[ij.y];
        if (sj == kOpen && fj != 0) {
          Fo[ij.y] = 0;
          stays = true;
        } else {
          if (sj == kOpen) St[ij.y] = (uint8_t)kKept;
          lv = false;
        }
      }
      if (stays) atomicAdd(&s_n[iters % 3], 1u);
      FIN_TOC(tp2);
      __threadfence_block();
      __syncthreads();
      const uint32_t live = s_n[iters % 3];
      FIN_TOC(tb2);
#ifdef S2A_MEASURE
      if (threadIdx.x == 0 && cnt0 > 1500) atomicAdd(&g_fin_dbg[live > 256 ? 12 : live > 64 ? 13 : live > 16 ? 14 : 15], 1ull);
#endif
      if (live == 0) break;
      f ^= 1;
    }
  }
  // (a third phase -- the last <= 64 / 256 / 512 edges handed to wave 0 alone, in registers, without workgroup barriers --
  // measured SLOWER: 44 / 68 / 147 us against 41: a lone wave pays every instruction's latency itself, and the branchy
  // predicated passes are a few hundred instructions)
#ifdef S2A_MEASURE
  if (threadIdx.x == 0 && LDS && cnt0 > 1500) {           // the big segments: cycles of wave 0 per phase, summed (phases A, B)
    atomicAdd(&g_fin_dbg[6], (unsigned long long)iters);
    atomicAdd(&g_fin_dbg[7], tp1); atomicAdd(&g_fin_dbg[8], tb1); atomicAdd(&g_fin_dbg[9], tp2); atomicAdd(&g_fin_dbg[10], tb2);
  }
#endif
  return iters;
}

__global__ __launch_bounds__(kFinThreads) void k_nms_finish_segments(NmsCounters* __restrict__ C,
                                                                     const uint32_t* __restrict__ seg_start,
                                                                     const uint32_t* __restrict__ num_seg,
                                                                     const uint2* __restrict__ alive, unsigned long long alive_cap,
                                                                     uint2* __restrict__ bucketed, unsigned long long bucket_cap,
                                                                     uint8_t* __restrict__ state,
                                                                     uint8_t* __restrict__ blocked, int64_t n,
                                                                     const unsigned long long* __restrict__ keys, int shift,
                                                                     uint32_t ignore_key, int use_ignore, int force_global) {
  if (C->alive[kNmsRounds] == 0) return;
  __shared__ uint8_t s_state[kSegRows];
  __shared__ uint8_t s_flag[2 * kSegRows];
  __shared__ uint2 s_edges[2 * kSegEdges];
  __shared__ unsigned s_n[3];
  __shared__ unsigned s_ecnt;
  __shared__ unsigned long long s_gbase;
  const uint32_t S = *num_seg;
  const unsigned long long A = min(C->alive_list, alive_cap);
  const int lane = threadIdx.x & 63;
  for (uint32_t s = blockIdx.x; s < S; s += gridDim.x) {
    const uint32_t st = seg_start[s], ns = seg_start[s + 1] - st;
    if (ns == 0u) continue;                                   // (uniform)
    // the ignored rows (the padding of a detector batch: three quarters of the buffer) are removed from the start, have no
    // edges and nothing to write out
    if (use_ignore && (uint32_t)(keys[st] >> shift) == ignore_key) continue;
    __syncthreads();
    if (threadIdx.x == 0) { s_ecnt = 0; s_n[1] = 0; }         // round r counts in s_n[r % 3]
    // the implicit view after the last launched round, written out for this segment's rows (k_nms_finish reads explicit
    // states once edges are alive); the two `blocked` arrays become plain "blocked this round" flags (kNmsRounds is even:
    // the view reads array 0 at this very index, before the write)
    for (uint32_t i = threadIdx.x; i < ns; i += kFinThreads) {
      const uint32_t p = st + i;
      const uint32_t v = nms_view(state, blocked, n, kNmsRounds, p);
      blocked[p] = 0;
      blocked[n + p] = 0;
      if (v == kKept) state[p] = (uint8_t)kKept;
    }
    __syncthreads();
    // this segment's edges: one scan of the alive list, four entries per thread in flight; the first kSegEdges go to LDS
    for (unsigned long long e0 = 0; e0 < A; e0 += 4ull * kFinThreads) {
      uint2 ij[4];
      bool mine[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const unsigned long long e = e0 + (unsigned long long)k * kFinThreads + threadIdx.x;
        ij[k] = e < A ? alive[e] : make_uint2(0xffffffffu, 0u);
      }
#pragma unroll
      for (int k = 0; k < 4; k++) {
        mine[k] = ij[k].x - st < ns;                          // (unsigned: also false for the 0xffffffff filler)
        const unsigned long long bal = __ballot(mine[k]);
        if (bal) {
          unsigned slot = 0;
          if (lane == 0) slot = atomicAdd(&s_ecnt, (unsigned)__popcll(bal));
          slot = (unsigned)__shfl((int)slot, 0) + (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
          if (mine[k] && slot < (unsigned)kSegEdges) s_edges[slot] = make_uint2(ij[k].x - st, ij[k].y - st);
        }
      }
    }
    __syncthreads();
    const uint32_t cnt0 = s_ecnt;
    if (cnt0 == 0) continue;                                  // (uniform)
    const bool lds = !force_global && ns <= (uint32_t)kSegRows && cnt0 <= (uint32_t)kSegEdges;
    unsigned iters;
    if (lds) {
      for (uint32_t i = threadIdx.x; i < ns; i += kFinThreads) { s_state[i] = state[st + i]; s_flag[i] = 0; s_flag[kSegRows + i] = 0; }
      __threadfence_block();
      __syncthreads();
      iters = finish_rounds<true>(s_state, s_flag, kSegRows, s_edges, kSegEdges, cnt0, s_n);
      for (uint32_t i = threadIdx.x; i < ns; i += kFinThreads) state[st + i] = s_state[i];
    } else {
      // a segment beyond the LDS arrays: its edges (local indices) into a region of the dead edge buffer, reserved with one
      // atomic; a second scan of the alive list fills it.  (No room left there -- cannot happen while the alive list fits
      // the pair buffer it lives in -- would leave the rows to the direct kernel: status bit 2.)
      if (threadIdx.x == 0) { s_gbase = atomicAdd(&C->bucket_cursor, (unsigned long long)cnt0); s_ecnt = 0; }
      __syncthreads();
      const unsigned long long gb = s_gbase;
      if (gb + cnt0 > bucket_cap) {                           // (uniform)
        if (threadIdx.x == 0) atomicOr(&C->status, 2u);
        continue;
      }
      for (unsigned long long e0 = 0; e0 < A; e0 += kFinThreads) {
        const unsigned long long e = e0 + threadIdx.x;
        uint2 ij = e < A ? alive[e] : make_uint2(0xffffffffu, 0u);
        const bool mine = ij.x - st < ns;
        const unsigned long long bal = __ballot(mine);
        if (bal) {
          unsigned slot = 0;
          if (lane == 0) slot = atomicAdd(&s_ecnt, (unsigned)__popcll(bal));
          slot = (unsigned)__shfl((int)slot, 0) + (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
          if (mine) bucketed[gb + slot] = make_uint2(ij.x - st, ij.y - st);
        }
      }
      __threadfence_block();
      __syncthreads();
      iters = finish_rounds<false>(state + st, blocked + st, n, bucketed + gb, 0, cnt0, s_n);
    }
    (void)iters;
#ifdef S2A_MEASURE
    if (threadIdx.x == 0) {
      atomicMax(&g_fin_dbg[0], (unsigned long long)iters); atomicMax(&g_fin_dbg[1], (unsigned long long)cnt0);
      atomicAdd(&g_fin_dbg[2], (unsigned long long)iters); atomicAdd(&g_fin_dbg[3], 1ull);
      atomicMax(&g_fin_dbg[4], (unsigned long long)ns); atomicAdd(&g_fin_dbg[5], (unsigned long long)cnt0);
    }
#endif
  }
}

// FALLBACK (a list overflowed: pathologically dense input): the plain definition of greedy NMS, one workgroup per
// segment, no lists -- for every row in score order that is still open: keep it and test all later open rows against it.
// Serial over the kept rows; only ever reached when nearly everything overlaps, i.e. when few rows are kept.
__global__ __launch_bounds__(kThreads) void k_nms_greedy_direct(const PreBox* __restrict__ sorted,
                                                                const uint32_t* __restrict__ seg_start,
                                                                const uint32_t* __restrict__ num_seg,
                                                                const unsigned long long* __restrict__ keyA_s,
                                                                uint32_t ignore_key, int use_ignore,
                                                                NmsCounters* __restrict__ C, unsigned long long cap,
                                                                unsigned long long ecap, unsigned long long tile_cap,
                                                                uint8_t* __restrict__ state /* score order: state_fb */,
                                                                float thr) {
  if (!(C->pairs > cap || C->edges > ecap || C->tiles > tile_cap || (C->status & 2u))) return;
  __shared__ float2 s_pts[24 * kThreads];
  __shared__ unsigned s_next;
  if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(&C->status, 1u);
  const uint32_t S = *num_seg;
  for (uint32_t s = blockIdx.x; s < S; s += gridDim.x) {
    const uint32_t st = seg_start[s], ns = seg_start[s + 1] - st;
    if (ns == 0u) continue;                                                   // (the own segment sort lists empty segments)
    if (use_ignore && (uint32_t)(keyA_s[st] >> 32) == ignore_key) continue;   // ignored rows: stay removed
    for (uint32_t i = threadIdx.x; i < ns; i += kThreads) state[st + i] = (uint8_t)kOpen;
    __syncthreads();
    uint32_t from = 0;
    for (;;) {
      if (threadIdx.x == 0) s_next = 0xffffffffu;
      __syncthreads();
      // first open row at or after `from` (rows are only ever closed behind the scan front)
      for (uint32_t i = from + threadIdx.x; i < ns; i += kThreads) {
        if (state[st + i] == kOpen) { atomicMin(&s_next, i); break; }
      }
      __syncthreads();
      const uint32_t i = s_next;
      __syncthreads();
      if (i == 0xffffffffu) break;
      const PreBox A = sorted[st + i];
      if (threadIdx.x == 0) state[st + i] = (uint8_t)kKept;
      for (uint32_t j = i + 1 + threadIdx.x; j < ns; j += kThreads) {
        if (state[st + j] != kOpen) continue;
        const PreBox B = sorted[st + j];
        if (!surely_disjoint(A.x, A.y, A.r, B.x, B.y, B.r) && !sat_disjoint(A, B) &&
            rbox_iou<kThreads>(A, B, s_pts + threadIdx.x) > thr)
          state[st + j] = (uint8_t)kRemoved;
      }
      from = i + 1;
      __syncthreads();
    }
  }
}

// keep flag of every row in the caller's order.  state / blocked live in the cull's row order (perm = that order's
// original rows); the direct fallback works in score order on its own array (state_fb, perm_seg).
__global__ void k_nms_finish(const uint8_t* __restrict__ state, const uint8_t* __restrict__ blocked,
                             const NmsCounters* __restrict__ C, const int32_t* __restrict__ perm,
                             const uint8_t* __restrict__ state_fb, const int32_t* __restrict__ perm_seg, int64_t n,
                             uint8_t* __restrict__ keep_orig) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  uint32_t v;
  if (C->status & 1u) {              // a list overflowed: k_nms_greedy_direct settled every row (score order)
    keep_orig[perm_seg[p]] = state_fb[p] == kKept ? 1 : 0;
    return;
  }
  if (C->alive[kNmsRounds] != 0) {
    v = state[p];                    // the clean-up kernel settled every row explicitly
  } else {
    int r = kNmsRounds;              // the view after the first launched round that left no edge alive
    for (int k = 1; k <= kNmsRounds; k++)
      if (C->alive[k] == 0) { r = k; break; }
    v = nms_view(state, blocked, n, r, (uint32_t)p);
  }
  keep_orig[perm[p]] = v == kKept ? 1 : 0;
}

// ---------------------------------------------------------------- SMALL inputs in ONE launch (round 5)
// The drop-in ops are synchronous (the host reads the count), so a call costs its chain of launches: ~40 of them at
// ~5 us each whatever the size -- 0.20 ms for 5 000 rows, the size one image's post-processing really has
// (utils/bbox_nms_rotated.py:47).  For n <= kSmallN rows whose labels split them into segments of <= kSmallSeg rows:
//   one workgroup per distinct label (every workgroup finds the distinct labels itself: <= 16 label loads per thread, an
//   LDS hash set, a rank sort of <= 64 entries) -- collects its rows, sorts (~score | row) keys in LDS, pre-processes the
//   boxes, culls all pairs (circle, separating axes, IoU upper bound: the big path's tests) into an LDS candidate list,
//   evaluates the candidates exactly (8-slot lanes, the rare 24-slot redo) into a suppression BIT MASK in LDS, resolves the
//   greedy order with one wave walking the mask rows, and appends its kept keys to a global list; the LAST workgroup to
//   finish (ticket) sorts the kept keys of all labels and writes keep[] in descending score order.
// Anything outside the limits (more labels, a bigger segment, more candidates than the list holds) raises a status bit and
// the host runs the general path: never a wrong or partial answer.
constexpr int kSmallN = 16384, kSmallSeg = 640, kSmallLabels = 64, kSmallCand = 4096, kSmallGrid = kSmallLabels;
constexpr int kSmallW = kSmallSeg / 64;                   // mask words per row
constexpr int kSmOffBox = kSmallSeg * kSmallW * 8;        // 51 200
constexpr int kSmOffCand = kSmOffBox + kSmallSeg * 32;    // 71 680
constexpr int kSmOffPts = kSmOffCand + kSmallCand * 4;    // 88 064
constexpr int kSmallExact = 768;                          // threads of the exact pass (8 candidate-point slots each)
constexpr int kSmOffKey = kSmOffPts + 8 * kSmallExact * 8;
constexpr int kSmOffIdx = kSmOffKey + 1024 * 8;
constexpr int kSmOffMisc = kSmOffIdx + kSmallSeg * 4;     // 147 968
constexpr int kSmallLds = kSmOffMisc + 1024;              // 148 992 (the merge holds <= 16 384 kept keys in [0, 131 072))
static_assert(kSmallN * 8 <= kSmOffMisc, "the merge buffer must not reach the counters");
struct SmallCtl {                 // one slot per call in flight (zero between calls: the last workgroup resets the first four)
  uint32_t kept, ticket, status, done;
  uint32_t result_count, result_status, pad2[2];
};
__device__ SmallCtl g_small_ctl[16];
#ifdef S2A_MEASURE
__device__ unsigned long long g_small_stamps[64 * 16];     // measurement builds: phase stamps of k_nms_small per workgroup
#define SMALL_STAMP(k) do { if (threadIdx.x == 0) g_small_stamps[blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SMALL_STAMP(k) do {} while (0)
#endif

__global__ __launch_bounds__(1024) void k_nms_small(const float* __restrict__ dets5, const float* __restrict__ scores,
                                                    const float* __restrict__ labels, int n, float thr,
                                                    unsigned long long* __restrict__ kept_keys, SmallCtl* __restrict__ ctl,
                                                    uint32_t* __restrict__ host_result /* pinned, mapped: count, status */,
                                                    int64_t* __restrict__ keep, int64_t* __restrict__ count_dev) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned long long* s_mask = reinterpret_cast<unsigned long long*>(smem);
  PreBox* s_box = reinterpret_cast<PreBox*>(smem + kSmOffBox);
  uint32_t* s_cand = reinterpret_cast<uint32_t*>(smem + kSmOffCand);
  float2* s_pts = reinterpret_cast<float2*>(smem + kSmOffPts);
  unsigned long long* s_key = reinterpret_cast<unsigned long long*>(smem + kSmOffKey);
  uint32_t* s_idx = reinterpret_cast<uint32_t*>(smem + kSmOffIdx);
  uint32_t* s_hash = reinterpret_cast<uint32_t*>(smem + kSmOffMisc);          // [128]
  uint32_t* s_list = s_hash + 128;                                             // [64] distinct label keys, then sorted
  uint32_t* s_ctr = s_list + 64;                                               // [8]: 0 distinct, 1 rows, 2 candidates, 3 redo, 4 last flag, 5 kept
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr uint32_t kEmptyKey = 0xffffffffu;       // (float_sortable never yields all ones for a label that is not a -NaN)
  if (tid < 128) s_hash[tid] = kEmptyKey;
  if (tid < 8) s_ctr[tid] = 0u;
  SMALL_STAMP(0);
  __syncthreads();
  // ---- distinct labels (every workgroup, the same set)
  uint32_t lab[kSmallN / 1024], sck[kSmallN / 1024];    // label keys and (~score) keys of this thread's rows: one round trip
#pragma unroll
  for (int k = 0; k < kSmallN / 1024; k++) {
    const int i = tid + 1024 * k;
    lab[k] = i < n ? nms_segkey(labels, nullptr, 0u, i) : kEmptyKey;
    sck[k] = i < n ? ~float_sortable(scores[i]) : 0u;
  }
  bool too_many = false;
#pragma unroll
  for (int k = 0; k < kSmallN / 1024; k++) {
    const uint32_t key = lab[k];
    if (tid + 1024 * k >= n) continue;
    uint32_t h = (key * 0x9e3779b1u) >> 25;          // 7 bits
    bool placed = false;
    for (int probe = 0; probe < 128; probe++) {
      // (look first: 5 000 rows are 5 000 reads of 15 hot slots, and same-address LDS atomics serialise lane by lane)
      uint32_t prev = __hip_atomic_load(&s_hash[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (prev == kEmptyKey) prev = atomicCAS(&s_hash[h], kEmptyKey, key);
      if (prev == kEmptyKey || prev == key) { placed = true; break; }
      h = (h + 1) & 127;
    }
    too_many |= !placed;
  }
  if (__syncthreads_or(too_many)) {                // > 128 distinct labels: every workgroup decides the same
    if (blockIdx.x == 0 && tid == 0) atomicOr(&ctl->status, 1u);
    if (tid == 0) s_ctr[6] = 1u;
    goto finish;
  }
  if (tid < 128 && s_hash[tid] != kEmptyKey) {
    const unsigned pos = atomicAdd(&s_ctr[0], 1u);
    if (pos < (unsigned)kSmallLabels) s_list[pos] = s_hash[tid];
  }
  __syncthreads();
  {
    const unsigned nd = s_ctr[0];
    if (nd > (unsigned)kSmallLabels) {                // (uniform, and the same in every workgroup)
      if (blockIdx.x == 0 && tid == 0) atomicOr(&ctl->status, 1u);
      if (tid == 0) s_ctr[6] = 1u;
      goto finish;
    }
    uint32_t mine = 0, rank = 0;
    if (tid < (int)nd) {
      mine = s_list[tid];
      for (unsigned j = 0; j < nd; j++) rank += s_list[j] < mine ? 1u : 0u;
    }
    __syncthreads();
    if (tid < (int)nd) s_list[rank] = mine;         // ascending label order (any fixed order would do)
    __syncthreads();
  }
  {
  const unsigned nd = s_ctr[0];
  if (blockIdx.x >= nd) return;                      // (no label for this workgroup: it takes no ticket)
  const uint32_t my_label = s_list[blockIdx.x];
  SMALL_STAMP(1);
  // ---- rows of this label
#pragma unroll
  for (int k = 0; k < kSmallN / 1024; k++) {
    const bool mine = tid + 1024 * k < n && lab[k] == my_label;
    const unsigned long long bal = __ballot(mine);
    if (bal) {
      unsigned slot = 0;
      if (lane == 0) slot = atomicAdd(&s_ctr[1], (unsigned)__popcll(bal));
      slot = (unsigned)__shfl((int)slot, 0) + (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
      // (the key goes straight into the sort buffer: ascending key = descending score, ties by ascending row)
      if (mine && slot < (unsigned)kSmallSeg) s_key[slot] = ((unsigned long long)sck[k] << 32) | (uint32_t)(tid + 1024 * k);
    }
  }
  __syncthreads();
  const unsigned ns = s_ctr[1];
  if (ns > (unsigned)kSmallSeg) {                    // (uniform) a segment beyond the LDS mask: general path
    if (tid == 0) atomicOr(&ctl->status, 2u);
  }
  const unsigned nsc = min(ns, (unsigned)kSmallSeg);
  const unsigned W = (nsc + 63) >> 6;
  unsigned P = 128;
  while (P < nsc) P <<= 1;
  for (unsigned r = nsc + tid; r < P; r += 1024) s_key[r] = ~0ull;
  for (unsigned i = tid; i < nsc * (unsigned)kSmallW; i += 1024) s_mask[i] = 0ull;
  __syncthreads();
  SMALL_STAMP(2);
  lds_bitonic_sort<1024>(s_key, P);
  SMALL_STAMP(3);
  for (unsigned r = tid; r < nsc; r += 1024) {
    const uint32_t o = (uint32_t)s_key[r];
    const float* d = dets5 + 5 * (int64_t)o;
    s_box[r] = make_prebox(d[0], d[1], d[2], d[3], d[4], 0.f);
  }
  __syncthreads();
  SMALL_STAMP(4);
  // ---- cull: 64 x 64 tiles of the upper triangle, a wave per tile, lane = row j, the 64 rows i of the other block broadcast
  // Stage 1, by tiles (half tiles: 32 rows of block ib against the 64 of block jb, a wave per item, lane = row j, the rows i a
  // broadcast read each): the circle test; survivors go to an LDS list (the exact pass's point slots are free until then).
  // Stage 2, dense: every thread takes ONE survivor through the separating-axis / IoU-bound test (in place in the tile loop
  // it ran for whole waves with a handful of live lanes: 25 k of the cull's 33-43 k cycles).
  uint32_t* s_pairs = reinterpret_cast<uint32_t*>(s_pts);          // <= kSmallPairs circle survivors, (i << 16) | j
  constexpr unsigned kSmallPairs = 8u * kSmallExact * 8u / 4u;     // 12 288
  const unsigned nb = W;
  for (unsigned t2 = wave; t2 < nb * (nb + 1); t2 += 16) {
    const unsigned t = t2 >> 1, half = t2 & 1u;
    unsigned ib = 0, rem = t;
    while (rem >= nb - ib) { rem -= nb - ib; ib++; }              // tile t = (ib, jb = ib + rem)
    const unsigned jb = ib + rem, j = jb * 64 + lane;
    const bool jv = j < nsc;
    PreBox Bj = {};
    if (jv) Bj = s_box[j];
    const unsigned ibeg = half * 32u, iend = min(ibeg + 32u, nsc - ib * 64 > ibeg ? nsc - ib * 64 : ibeg);
    unsigned pass = 0u;                                            // verdicts of this lane's row against the 32 rows i
    if (ibeg < iend) {                                             // (uniform)
      // a fixed 32 trips, eight rows requested before the first is used (a row per LDS round trip otherwise); rows behind
      // the segment's end are inside the array and masked out by i < j, j < nsc
#pragma unroll 8
      for (unsigned ii = 0; ii < 32u; ii++) {
        const unsigned i = ib * 64 + ibeg + ii;
        const float2 c = *reinterpret_cast<const float2*>(&s_box[i]);        // x, y  (same address in every lane)
        const float ar = s_box[i].r;
        const bool ok = jv && i < j && !surely_disjoint(c.x, c.y, ar, Bj.x, Bj.y, Bj.r);
        pass |= ok ? 1u << ii : 0u;
      }
    }
    // one reservation per wave and item (a prefix sum over the lanes), not one per row i: appends inside the loop put an
    // LDS atomic's round trip on every second trip
    unsigned cnt = (unsigned)__popc(pass), incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned up = (unsigned)__shfl_up((int)incl, o);
      if (lane >= o) incl += up;
    }
    const unsigned all = (unsigned)__shfl((int)incl, 63);
    if (all) {                                                     // (wave-uniform)
      unsigned slot = 0;
      if (lane == 0) slot = atomicAdd(&s_ctr[3], all);
      slot = (unsigned)__shfl((int)slot, 0) + incl - cnt;
      while (pass) {
        const unsigned ii = (unsigned)__builtin_ctz(pass);
        pass &= pass - 1u;
        if (slot < kSmallPairs) s_pairs[slot] = ((ib * 64 + ibeg + ii) << 16) | j;
        slot++;
      }
    }
  }
  __syncthreads();
  {
    const unsigned npairs = s_ctr[3];
    if (npairs > kSmallPairs && tid == 0) atomicOr(&ctl->status, 4u);
    const unsigned np = min(npairs, kSmallPairs);
    for (unsigned e = tid; e < np; e += 1024) {
      const uint32_t ij = s_pairs[e];
      const bool cand = !nms_pair_skippable(s_box[ij >> 16], s_box[ij & 0xffffu], thr);
      const unsigned long long bal = __ballot(cand);
      if (bal) {
        unsigned slot = 0;
        if (lane == 0) slot = atomicAdd(&s_ctr[2], (unsigned)__popcll(bal));
        slot = (unsigned)__shfl((int)slot, 0) + (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
        if (cand && slot < (unsigned)kSmallCand) s_cand[slot] = ij;
      }
    }
    __syncthreads();
    if (tid == 0) s_ctr[3] = 0u;                                  // (the redo counter of the exact pass)
  }
  __syncthreads();
  SMALL_STAMP(5);
  const unsigned ncand = s_ctr[2];
  if (ncand > (unsigned)kSmallCand) {
    if (tid == 0) atomicOr(&ctl->status, 4u);
  }
  const unsigned nc = min(ncand, (unsigned)kSmallCand);
  // ---- exact IoU of the candidates (higher-scored box first, as the general path); pairs with more than 8 candidate points
  // are marked (top bit) and redone with the reference's 24 slots by the first 128 threads
  if (tid < kSmallExact) {
    for (unsigned c = tid; c < nc; c += kSmallExact) {
      const uint32_t ij = s_cand[c];
      const unsigned i = ij >> 16, j = ij & 0xffffu;
      bool redo = false;
      const float v = rbox_iou<kSmallExact, kIouCap>(s_box[i], s_box[j], s_pts + tid, &redo);
      if (redo) { const unsigned q = atomicAdd(&s_ctr[3], 1u); s_idx[q % kSmallSeg] = c; if (q >= (unsigned)kSmallSeg) atomicOr(&ctl->status, 4u); }
      else if (v > thr) atomicOr(&s_mask[i * kSmallW + (j >> 6)], 1ull << (j & 63));
    }
  }
  __syncthreads();
  {
    const unsigned nredo = min(s_ctr[3], (unsigned)kSmallSeg);
    if (tid < 128) {
      for (unsigned q = tid; q < nredo; q += 128) {
        const uint32_t ij = s_cand[s_idx[q]];
        const unsigned i = ij >> 16, j = ij & 0xffffu;
        const float v = rbox_iou<128>(s_box[i], s_box[j], s_pts + tid);
        if (v > thr) atomicOr(&s_mask[i * kSmallW + (j >> 6)], 1ull << (j & 63));
      }
    }
  }
  __syncthreads();
  SMALL_STAMP(6);
  // ---- greedy order, one wave, 64 rows at a time: the block's own 64 x 64 corner of the mask sits in registers (lane r = row r)
  // and only the rows that HAVE an edge inside the block are walked (scalar work: a bit test, two v_readlane per such row; a
  // row without one cannot remove anybody here); then the kept rows of the block are OR-ed into the removed words of the
  // later blocks, lane = (later word, a sixth of the rows).  (One row per trip with the removed set spread over the lanes
  // cost ~300 cycles a row: a cross-lane read and an LDS round trip on the dependency chain.)
  if (wave == 0) {
    unsigned long long* s_removed = reinterpret_cast<unsigned long long*>(s_idx);   // [W] (s_idx is dead: the keys are made)
    if (lane < (int)kSmallW) s_removed[lane] = 0ull;
    wave_lds_handoff();
    unsigned long long keptw = 0ull;                  // lane w: kept bits of block w
    unsigned nkept = 0;
    const unsigned parts = 64u / W;                   // row slices of the transposed OR (W <= 10: >= 6)
    for (unsigned blk = 0; blk < W; blk++) {
      const unsigned row = blk * 64 + lane;
      const bool rv = row < nsc;
      const unsigned long long diag = rv ? s_mask[row * kSmallW + blk] : 0ull;
      unsigned long long rem = s_removed[blk];                              // (same address in every lane)
      rem = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(rem >> 32)) << 32) |
            (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)rem);
      unsigned long long src = __ballot(diag != 0ull);                      // rows with an edge into this block
      while (src) {
        const int r = __builtin_ctzll(src);
        src &= src - 1ull;
        if (!((rem >> r) & 1ull)) {
          const unsigned dlo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)diag, r);
          const unsigned dhi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(diag >> 32), r);
          rem |= ((unsigned long long)dhi << 32) | dlo;
        }
      }
      const unsigned rows_here = min(64u, nsc - blk * 64);
      const unsigned long long valid = rows_here == 64u ? ~0ull : ((1ull << rows_here) - 1ull);
      const unsigned long long kept = ~rem & valid;
      nkept += (unsigned)__popcll(kept);
      if (lane == (int)blk) keptw = kept;
      // this block's kept rows -> the removed words of the later blocks
      if (blk + 1 < W) {
        const unsigned w = blk + 1 + (unsigned)lane % (W - blk - 1 ? W - blk - 1 : 1), part = (unsigned)lane / (W - blk - 1);
        unsigned long long v = 0ull;
        if (part < parts) {
          unsigned long long m = kept;
          for (unsigned r = part; r < 64u; r += parts)
            if ((m >> r) & 1ull) v |= s_mask[(blk * 64 + r) * kSmallW + w];
        }
        if (v) atomicOr(&s_removed[w], v);
        wave_lds_handoff();
      }
    }
    // kept keys -> the global list (one reservation per workgroup): a run in ascending key order, its place and length
    // published for the merge
    unsigned base = 0;
    if (lane == 0) {
      base = atomicAdd(&ctl->kept, nkept);
      uint32_t* run_info = reinterpret_cast<uint32_t*>(kept_keys + n);
      run_info[2 * blockIdx.x] = base;
      run_info[2 * blockIdx.x + 1] = nkept;
      s_ctr[5] = base;
      s_ctr[7] = nkept;
    }
    base = (unsigned)__shfl((int)base, 0);
    unsigned before = 0;                              // kept rows in the words in front of this lane's word
    for (int w = 0; w < (int)W; w++) {
      const unsigned c = (unsigned)__popcll(__shfl(keptw, w));
      if (w < lane) before += c;
    }
    if (lane < (int)W) {
      unsigned long long m = keptw;
      unsigned k = 0;
      while (m) {
        const int b = __builtin_ctzll(m);
        m &= m - 1ull;
        kept_keys[base + before + k++] = s_key[lane * 64 + b];
      }
    }
  }
  }
finish:
  SMALL_STAMP(7);
  // ---- merge, by ALL labelled workgroups: once every run is published (ticket), each workgroup loads all kept keys into LDS
  // and ranks ITS keys among the other runs by binary search (all runs are sorted): keep[rank] = row.  The wait on the
  // ticket is safe -- at most 64 workgroups of one CU each, all resident -- and bounded anyway: a timeout raises a status bit
  // and the host takes the general path.  (One workgroup sorting all kept keys took 60 us for 5 000 rows.)
  __threadfence();                                   // this workgroup's kept keys are visible device-wide before its ticket
  __syncthreads();
  {
    uint32_t* s_ctr2 = reinterpret_cast<uint32_t*>(smem + kSmOffMisc) + 128 + 64;
    uint32_t* s_run = reinterpret_cast<uint32_t*>(smem + kSmOffMisc);               // [64][2] start, length (the label set is dead)
    const unsigned nd = min(s_ctr2[0], (unsigned)kSmallLabels);
    // too many labels: decided identically by every workgroup from its own LDS set (s_ctr[6]), before any ticket -- then
    // workgroup 0 alone reports (count 0, the status for the host) and nobody takes a ticket
    if (s_ctr2[6] != 0u) {
      if (blockIdx.x == 0 && tid == 0) {
        *count_dev = 0;
        __hip_atomic_store(&host_result[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&host_result[1], __hip_atomic_load(&ctl->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) | 1u,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __threadfence_system();
        ctl->kept = 0; ctl->ticket = 0; ctl->status = 0; ctl->done = 0;
      }
      return;
    }
    if (tid == 0) {
      atomicAdd(&ctl->ticket, 1u);
      unsigned polls = 0;
      while (__hip_atomic_load(&ctl->ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nd) {
        if (++polls > (1u << 16)) { atomicOr(&ctl->status, 8u); break; }       // ~10 ms: never a hang
        __builtin_amdgcn_s_sleep(8);
      }
    }
    __syncthreads();
    __threadfence();
    const unsigned status = __hip_atomic_load(&ctl->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned total = __hip_atomic_load(&ctl->kept, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (status) total = 0;
    unsigned long long* s_all = reinterpret_cast<unsigned long long*>(smem);
    const uint32_t* run_info = reinterpret_cast<const uint32_t*>(kept_keys + n);
    const unsigned my_start = s_ctr2[5], my_len = status ? 0u : s_ctr2[7];
    __syncthreads();                                 // (s_ctr2 / the label set are read: the run table may overwrite it)
    for (unsigned r = tid; r < total; r += 1024) s_all[r] = __builtin_nontemporal_load(&kept_keys[r]);
    if (tid < (int)nd * 2) s_run[tid] = __builtin_nontemporal_load(&run_info[tid]);
    __syncthreads();
    SMALL_STAMP(8);
    for (unsigned q = tid; q < my_len; q += 1024) {
      const unsigned long long key = s_all[my_start + q];
      unsigned rank = q;
      for (unsigned b0 = 0; b0 < nd; b0 += 8) {        // eight runs searched in lockstep: their probes are independent reads
        unsigned st8[8], len8[8], pos[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
          const unsigned b = b0 + k;
          const bool use = b < nd && b != blockIdx.x;
          st8[k] = use ? s_run[2 * b] : 0u;
          len8[k] = use ? s_run[2 * b + 1] : 0u;
          pos[k] = 0u;
        }
        for (unsigned step = 512; step; step >>= 1) {
#pragma unroll
          for (int k = 0; k < 8; k++) {
            const unsigned p = pos[k] + step;
            if (p <= len8[k] && s_all[st8[k] + p - 1u] < key) pos[k] = p;
          }
        }
#pragma unroll
        for (int k = 0; k < 8; k++) rank += pos[k];
      }
      keep[rank] = (int64_t)(uint32_t)key;
    }
    SMALL_STAMP(9);
    __syncthreads();
    if (tid == 0) {                                  // the workgroup that finishes last reports and clears the slot
      if (atomicAdd(&ctl->done, 1u) == nd - 1u) {
        // the status is read AGAIN here: a workgroup that timed out in its poll raised bit 8 (and wrote none of its rows)
        // possibly after this one took its copy above -- the reporter is the last to arrive, so every such atomicOr is
        // ordered in front of this load; a non-zero value reports count 0 and the host takes the general path
        const unsigned status_now = status | __hip_atomic_load(&ctl->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned total_now = status_now ? 0u : total;
        *count_dev = (int64_t)total_now;
        // (the host reads these two words after it has synchronised the stream: no copy launch behind the kernel)
        __hip_atomic_store(&host_result[0], total_now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&host_result[1], status_now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __threadfence_system();
        ctl->kept = 0; ctl->ticket = 0; ctl->status = 0; ctl->done = 0;     // ready for the next call on this slot
      }
    }
  }
}

// ---------------------------------------------------------------- side streams
// Two side streams + fork / join events per (device, caller stream), created on first use.  box_iou_rotated runs the
// zero-fill of a large output beside its pair finding; the NMS prelude runs its three independent sorts side by side.
// Fork / join is the event pattern that stream capture understands.  Keyed by the caller's stream so that calls on
// different streams (bench.py keeps three batches in flight) never queue behind each other's side work.
struct SideSet {
  int dev = -1;
  hipStream_t caller = nullptr;
  hipStream_t s[2] = {nullptr, nullptr};
  hipEvent_t fork = nullptr, join[2] = {nullptr, nullptr};
};
std::mutex g_side_mutex;
std::vector<SideSet*> g_sides;

int side_set(hipStream_t caller, SideSet** out) {
  int dev = 0;
  S2A_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(g_side_mutex);
  for (SideSet* e : g_sides)
    if (e->dev == dev && e->caller == caller) { *out = e; return S2A_OK; }
  SideSet* e = new SideSet();
  e->dev = dev;
  e->caller = caller;
  // (default priority: a low-priority fill is starved by the persistent grids of the chain and then runs alone at the
  // end, in front of the scatter -- 210 us instead of ~150)
  for (int k = 0; k < 2; k++) {
    S2A_HIP(hipStreamCreateWithFlags(&e->s[k], hipStreamNonBlocking));
    S2A_HIP(hipEventCreateWithFlags(&e->join[k], hipEventDisableTiming));
  }
  S2A_HIP(hipEventCreateWithFlags(&e->fork, hipEventDisableTiming));
  g_sides.push_back(e);
  *out = e;
  return S2A_OK;
}

inline bool stream_capturing(hipStream_t st) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  return hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
}

// ---------------------------------------------------------------- NMS workspace plan
struct NmsPlan {
  size_t rocprim_bytes;           // scratch of ONE 64-bit-key pair sort (three of them: the sorts run side by side)
  unsigned long long queue_cap;   // pair list (cull survivors)
  unsigned long long edge_cap;    // edges (pairs above the threshold)
  unsigned long long tile_cap;    // tiles that pass the bounding-box filter
};

struct NmsBuffers {
  unsigned long long *keyA, *keyA_s, *keyB, *keyB_s, *keyC, *keyC_s;
  int32_t *idx, *perm_glob, *perm_seg, *perm_sp;
  uint32_t *segidx1, *seg_start, *num_seg, *cnt, *cnt2, *cnt3, *lasthead;
  unsigned long long* rankkey;
  uint4* bbox_part;
  NmsCounters* C;
  PreBox *sorted, *sp_box;
  uint2 *lo, *hi;
  TileRef* tiles;
  uint2 *gq, *edges;
  uint8_t *keep_orig, *state, *state_fb, *blocked;
  uint32_t *seg_cnt, *seg_cur;
  // own order-B sort (k_spb_*): hash table of the distinct segment keys, (segment, cell) histogram, per-row bucket / rank,
  // [0] distinct keys, [1] table-full flag; and the A-order segment starts of the direct fallback
  unsigned long long* htab;
  uint32_t *hist, *spb_ctr, *rank, *bucket, *seg_start_a, *num_seg_a;
  void* rp_temp[3];
};

int nms_plan(int64_t n, int64_t max_seg_rows, NmsPlan* plan) {
  if (max_seg_rows <= 0 || max_seg_rows > n) max_seg_rows = n;
  size_t a = 0;
  unsigned long long* k64 = nullptr;
  int32_t* i32 = nullptr;
  size_t sz = (size_t)n;
  hipStream_t s0 = nullptr;  // size query only: nothing is launched
  bool ok = rocprim::radix_sort_pairs(nullptr, a, k64, k64, i32, i32, sz, 0, 64, s0) == hipSuccess;
  if (!ok) {
    // no device visible (size query on a CPU-only host): generous closed-form bound —
    // double-buffered keys+values plus histograms
    (void)hipGetLastError();
    a = sz * 32 + (8u << 20);
  }
  plan->rocprim_bytes = a;
  // pair list: all same-segment pairs when small, else 16 Mi entries + 256/row
  unsigned long long all_pairs = (unsigned long long)n * (unsigned long long)max_seg_rows / 2 + 64;
  unsigned long long want = (16ull << 20) + 256ull * (unsigned long long)n;
  plan->queue_cap = std::min(all_pairs, want);
  plan->edge_cap = std::min(plan->queue_cap, (8ull << 20) + 64ull * (unsigned long long)n);
  // tiles: single-block segments give one tile each (<= n); the others hold at most n/32 + 2 blocks together
  unsigned long long nb = ((unsigned long long)max_seg_rows + 63) / 64;
  plan->tile_cap = std::min<unsigned long long>((unsigned long long)n + ((unsigned long long)n / 32 + 2) * (nb + 1) / 2, 32ull << 20);
  return 0;
}

// everything except the three lists (tiles, pairs, edges)
void nms_carve_fixed(Carver& cv, int64_t n, const NmsPlan& pl, NmsBuffers* B) {
  size_t sz = (size_t)n;
  B->keyA = cv.take<unsigned long long>(sz);
  B->keyA_s = cv.take<unsigned long long>(sz);
  B->keyB = cv.take<unsigned long long>(sz);
  B->keyB_s = cv.take<unsigned long long>(sz);
  B->keyC = cv.take<unsigned long long>(sz);
  B->keyC_s = cv.take<unsigned long long>(sz);
  B->idx = cv.take<int32_t>(sz);
  B->perm_glob = cv.take<int32_t>(sz);
  B->perm_seg = cv.take<int32_t>(sz);
  B->perm_sp = cv.take<int32_t>(sz);
  B->rankkey = cv.take<unsigned long long>(sz);
  B->segidx1 = cv.take<uint32_t>(sz);
  B->seg_start = cv.take<uint32_t>(sz + 1);
  B->num_seg = cv.take<uint32_t>(4);
  B->cnt = cv.take<uint32_t>(kSegCountBlocks);
  B->cnt2 = cv.take<uint32_t>(kSegCountBlocks);
  B->cnt3 = cv.take<uint32_t>(kSegCountBlocks);
  B->lasthead = cv.take<uint32_t>(kSegCountBlocks);
  B->bbox_part = cv.take<uint4>(kPrepBlocks);
  B->C = cv.take<NmsCounters>(1);
  B->sorted = cv.take<PreBox>(sz);
  B->sp_box = cv.take<PreBox>(sz);
  B->lo = cv.take<uint2>(block_slots_for(sz));
  B->hi = cv.take<uint2>(block_slots_for(sz));
  B->keep_orig = cv.take<uint8_t>(sz);
  B->state = cv.take<uint8_t>(sz);
  B->state_fb = cv.take<uint8_t>(sz);
  B->blocked = cv.take<uint8_t>(2 * sz + 4);
  B->seg_cnt = cv.take<uint32_t>(sz + 2);
  B->seg_cur = cv.take<uint32_t>(sz + 2);
  B->htab = cv.take<unsigned long long>(kSpbSlots);
  B->hist = cv.take<uint32_t>(kSpbHist + 64 + kSpbGroups);        // histogram | group totals
  B->spb_ctr = cv.take<uint32_t>(4);
  B->rank = cv.take<uint32_t>(sz);
  B->bucket = cv.take<uint32_t>(sz);
  B->seg_start_a = cv.take<uint32_t>(sz + 1);
  B->num_seg_a = cv.take<uint32_t>(4);
  for (int k = 0; k < 3; k++) B->rp_temp[k] = cv.take<char>(pl.rocprim_bytes);
}

void nms_carve(Carver& cv, int64_t n, const NmsPlan& pl, NmsBuffers* B) {
  nms_carve_fixed(cv, n, pl, B);
  B->tiles = cv.take<TileRef>(pl.tile_cap);
  B->edges = cv.take<uint2>(pl.edge_cap);
  B->gq = cv.take<uint2>(pl.queue_cap);
}

inline unsigned grid_for(int64_t n, int threads = 256) { return (unsigned)((n + threads - 1) / threads); }

// rows per block of the block-count kernels: 256 up to 1 Mi rows, then whatever keeps the block count <= kSegCountBlocks
inline int count_rows(int64_t n) { return 256 * (int)((n + 256ll * kSegCountBlocks - 1) / (256ll * kSegCountBlocks)); }

// score order A: rows by (segment key, descending score) -> sorted[], state_fb[], perm_seg (+ segment starts / state when it
// is the MAIN order of the call)
int nms_order_a(const float* dets, int64_t n, uint32_t ignore_key, int use_ignore, const NmsPlan& pl, NmsBuffers& B,
                hipStream_t q, bool main_order) {
  const int rows = count_rows(n);
  const int nb = (int)((n + rows - 1) / rows);
  size_t rpb = pl.rocprim_bytes;
  S2A_HIP(rocprim::radix_sort_pairs(B.rp_temp[0], rpb, B.keyA, B.keyA_s, B.idx, B.perm_seg, (size_t)n, 0, 64, q));
  k_nms_seg_count<<<nb, 256, 0, q>>>(B.keyA_s, 32, n, rows, B.cnt, nullptr);
  k_nms_pos_meta<<<nb, 256, 0, q>>>(dets, B.keyA_s, B.perm_seg, B.cnt, nb, rows, n, ignore_key, use_ignore, B.segidx1,
                                    main_order ? B.seg_start : nullptr, B.num_seg, B.sorted,
                                    main_order ? B.state : nullptr, B.state_fb,
                                    main_order ? nullptr : B.seg_start_a, B.num_seg_a);
  S2A_LAUNCH_CHECK();
  return S2A_OK;
}

// the overflow fallback of a call whose score order was deferred (nms_core, defer_a): order A now, then the plain
// definition of greedy NMS per segment and the keep flags from ITS states.  Only ever reached on pathologically dense input.
int nms_overflow_fallback(const float* dets, int64_t n, float thr, const NmsPlan& pl, NmsBuffers& B, hipStream_t st) {
  int rc = nms_order_a(dets, n, 0u, 0, pl, B, st, false);
  if (rc != S2A_OK) return rc;
  k_nms_greedy_direct<<<512, kThreads, 0, st>>>(B.sorted, B.seg_start_a, B.num_seg_a, B.keyA_s, 0u, 0, B.C, pl.queue_cap, pl.edge_cap,
                                                pl.tile_cap, B.state_fb, thr);
  k_nms_finish<<<grid_for(n), 256, 0, st>>>(B.state, B.blocked, B.C, B.perm_sp, B.state_fb, B.perm_seg, n, B.keep_orig);
  S2A_LAUNCH_CHECK();
  return S2A_OK;
}

// shared core: everything up to keep_orig[].  On return the caller's stream has the (group, score) order of the OUTPUT
// in B.perm_glob / B.keyC_s (joined from its side stream).
// defer_a (in / out, may be NULL): a SYNCHRONOUS caller (it reads the keep count anyway) asks for the score order A of the
// spatial path -- needed by the overflow fallback alone -- not to be built; *defer_a comes back true when it was left out:
// the caller checks the overflow word behind its synchronisation and runs nms_overflow_fallback if it is set.  (The two
// library sorts of order A and order C ran beside the cull on side streams: 25 launches, cull 186 -> 224 us at 200 k rows.)
int nms_core(const float* dets, const float* scores, const float* labels, const int32_t* seg_ids,
             const int32_t* group_ids, int64_t n, uint32_t num_segments_hint, uint32_t num_groups,
             float thr, const NmsPlan& pl, NmsBuffers& B, hipStream_t st, bool side_streams, bool want_order_c = true,
             const long long* row_limit = nullptr, bool* defer_a = nullptr) {
  size_t sz = (size_t)n;
  const unsigned g = grid_for(n);
  const uint32_t ignore_key = num_segments_hint;           // one past the last real segment
  const int use_ignore = seg_ids != nullptr;
  // The Morton order + bounding-box tile filter pays when segments are big (thousands of rows: most tiles of a segment
  // are far apart); for many small segments (a detector batch: ~300 rows per image and class) it is a third sort and
  // two passes of pure overhead, every tile is tested anyway -> blocks in score order, no filter
  bool spatial = (seg_ids ? (uint64_t)n / std::max<uint32_t>(num_segments_hint, 1u) : (uint64_t)n) > 4096;
  if (const char* e = std::getenv("S2A_NMS_SPATIAL")) spatial = e[0] == '1';          // A/B and test switch
  const char* e_lds = std::getenv("S2A_NMS_FINISH_GLOBAL");                          // test switch: finish on the global arrays
  const int force_global = e_lds && e_lds[0] == '1';
  // one segment and no groups (plain nms_rotated): order A is the output order
  const bool need_c = want_order_c && (labels != nullptr || seg_ids != nullptr || group_ids != nullptr);
  // order B by the own counting sort (four short launches behind the key kernel) instead of rocPRIM's pair sort (nine);
  // S2A_NMS_SORTB=0: A/B, tests
  bool own_sort = spatial;
  if (const char* e = std::getenv("S2A_NMS_SORTB")) own_sort = spatial && e[0] != '0';
  // (only when the output order is an order of its own: a single-class call without groups emits in order A itself)
  const bool deferred = defer_a && *defer_a && spatial && need_c;
  if (defer_a) *defer_a = deferred;
  uint2* lo = spatial ? B.lo : nullptr;
  const size_t slots = block_slots_for(sz);
  const unsigned gp = (unsigned)std::min<size_t>(kPrepBlocks, std::max<size_t>(grid_for(n), 1));
  const int rows = count_rows(n);
  const int nb = (int)((n + rows - 1) / rows);
  // order A of a segmented call with small segments by the own segment sort (k_seg_hist + k_seg_sort) instead of the library's
  // pair sort over the whole (mostly padding) buffer; S2A_NMS_SEGSORT=0: A/B, tests
  const int64_t nchunks = (n + kSegChunk - 1) / kSegChunk;
  bool seg_sort = !spatial && seg_ids != nullptr && num_segments_hint <= (uint32_t)kSegSortMaxSeg &&
                  n >= (int64_t)num_segments_hint + 2 && nchunks <= (int64_t)kSpbHist - kSegSortMaxSeg - 8 &&
                  (int64_t)num_segments_hint * n <= (1ll << 28);
  if (const char* e = std::getenv("S2A_NMS_SEGSORT")) seg_sort = seg_sort && e[0] != '0';
  k_nms_prep<<<gp, 256, 0, st>>>(dets, scores, labels, seg_ids, group_ids, num_groups, ignore_key, n, B.keyA,
                                 need_c ? B.keyC : nullptr, B.idx, B.bbox_part, B.C,
                                 reinterpret_cast<uint32_t*>(B.blocked), (2 * sz + 3) / 4, B.seg_cnt, sz + 2, lo, B.hi, slots,
                                 own_sort ? B.htab : nullptr, B.hist, B.spb_ctr, seg_sort ? (uint32_t)kSegSortMaxSeg + 8u : 0u,
                                 row_limit);
  // A call is as long as its chain of LAUNCHES while the kernels are short (the host needs ~5 us per launch, a rocPRIM
  // sort is nine of them).  For big segments only the spatial order B is enqueued in front of the cull, and everything
  // behind the cull works on SPATIAL positions (greedy direction from the rank keys): the score order A is needed by the
  // overflow fallback alone and the output order C by the final compaction, so both go to side streams behind the cull's
  // launch and are joined late:
  //   spatial:      prep, B-keys, sort B, seg_count, sp_meta, filter, cull, exact pass, rounds ... | side 0: sort A, pos_meta
  //                                                                                                 | side 1: sort C
  //   score blocks: prep, sort A, seg_count, pos_meta, filter, cull, ...                            | side 1: sort C
  // Under stream capture everything stays on the caller's stream (S2A_NMS_FORK=0 / 1 forces either form: A/B and tests).
  // side_streams: the drop-in ops (one call, the host waits for its count: latency matters) use them; the detector's
  // segmented call does not -- with three batches in flight the extra queues cost 1.6 % of the end-to-end rate and on one
  // stream they gain nothing there (the output order is needed right behind the short cull of a detector batch).
  bool fork = side_streams && !stream_capturing(st);
  if (const char* e = std::getenv("S2A_NMS_FORK")) fork = e[0] == '1' && !stream_capturing(st);
  SideSet* ss = nullptr;
  if (fork && (spatial || need_c)) {
    int rc = side_set(st, &ss);
    if (rc != S2A_OK) return rc;
    S2A_HIP(hipEventRecord(ss->fork, st));
  }
  auto chain_a = [&](hipStream_t q, bool main_order) -> int {   // score order: sorted[], state_fb[] (+ segments, state)
    return nms_order_a(dets, n, ignore_key, use_ignore, pl, B, q, main_order);
  };
  auto sort_c = [&](hipStream_t q) -> int {
    size_t rpb = pl.rocprim_bytes;
    // no groups: the key is the 32-bit score word alone (the group word is 0 for every row) -- half the radix passes
    S2A_HIP(rocprim::radix_sort_pairs(B.rp_temp[2], rpb, B.keyC, B.keyC_s, B.idx, B.perm_glob, sz, 0, group_ids ? 64 : 32, q));
    return S2A_OK;
  };
  const PreBox* boxes = B.sorted;                // blocks in score order: the cull walks sorted[] itself
  const unsigned long long* rankkey = nullptr;   // ... and a lower position IS the higher score
  const int32_t* perm = B.perm_seg;              // original row of a position
  if (spatial) {
    k_nms_spkeys<<<g, 256, 0, st>>>(dets, B.keyA, B.bbox_part, (int)gp, n, B.keyB, own_sort ? B.htab : nullptr, B.spb_ctr);
    if (own_sort) {
      k_spb_hist<<<g, 256, 0, st>>>(B.keyB, n, B.htab, B.spb_ctr, B.hist, B.bucket, B.rank, B.C, ignore_key, use_ignore);
      k_spb_scan1<<<kSpbGroups / 4, 256, 0, st>>>(B.hist, B.hist + kSpbHist + 64, n);
      k_spb_scatter<<<g, 256, 0, st>>>(B.keyB, n, B.hist, B.hist + kSpbHist + 64, B.bucket, B.rank, B.keyB_s, B.perm_sp);
    } else {
      size_t rpb = pl.rocprim_bytes;
      S2A_HIP(rocprim::radix_sort_pairs(B.rp_temp[1], rpb, B.keyB, B.keyB_s, B.idx, B.perm_sp, sz, 0, 52, st));
    }
    k_nms_seg_count<<<nb, 256, 0, st>>>(B.keyB_s, 20, n, rows, B.cnt2, B.lasthead);
    k_nms_sp_meta<<<nb, 256, 0, st>>>(dets, B.keyB_s, B.perm_sp, B.cnt2, B.lasthead, nb, rows, n, scores, ignore_key,
                                      use_ignore, B.segidx1, B.seg_start, B.num_seg, B.sp_box, B.rankkey, B.state, B.lo, B.hi);
    k_nms_tile_filter<<<g, kThreads, 0, st>>>(B.segidx1, B.seg_start, B.keyB_s, 20, ignore_key, use_ignore, n, lo, B.hi,
                                              B.tiles, B.C, pl.tile_cap);
    boxes = B.sp_box;
    rankkey = B.rankkey;
    perm = B.perm_sp;
  } else {
    if (seg_sort) {
      uint32_t* seg_n = B.hist;
      uint32_t* ign_cnt = B.hist + kSegSortMaxSeg + 8;
      k_seg_hist<<<(unsigned)nchunks, 256, 0, st>>>(seg_ids, n, row_limit, num_segments_hint, seg_n, ign_cnt);
      k_seg_sort<<<(unsigned)(num_segments_hint + nchunks), 1024, 0, st>>>(
          dets, scores, seg_ids, n, row_limit, num_segments_hint, seg_n, ign_cnt, (int)nchunks, ignore_key, B.keyA, B.keyA_s, B.perm_seg,
          B.segidx1, B.seg_start, B.num_seg, B.sorted, B.state, B.state_fb);
    } else {
      int rc = chain_a(st, true);
      if (rc != S2A_OK) return rc;
    }
    k_nms_tile_filter<<<g, kThreads, 0, st>>>(B.segidx1, B.seg_start, B.keyA_s, 32, ignore_key, use_ignore, n, nullptr, B.hi,
                                              B.tiles, B.C, pl.tile_cap);
  }
  uint2* pair_list = B.gq;                       // what the exact pass reads
  unsigned long long pair_cap = pl.queue_cap;
  {
    // one wave per tile pays off once there are tiles for every wave (200 k rows: 392 -> 335 us); with few tiles the four
    // waves per tile of k_nms_cull finish a tile sooner (5 k / 20 k rows: 35 us less).  S2A_NMS_CULL_LANES=0|1 forces.
    const char* nl = getenv("S2A_NMS_CULL_LANES");
    const bool lanes = nl && (nl[0] == '0' || nl[0] == '1') ? nl[0] == '1' : n >= kNmsLanesRows;
    if (!lanes)
      k_nms_cull<<<kPersistentGrid, kThreads, 0, st>>>(boxes, B.tiles, B.C, pl.tile_cap, B.gq, pl.queue_cap, thr);
    else
      k_nms_cull_lanes<<<8192, kThreads, 0, st>>>(boxes, B.tiles, B.C, pl.tile_cap, B.gq, pl.queue_cap, thr);
  }
  // behind the cull's launch: the score order (spatial path: fallback only) and the output order
  if (spatial && !deferred) {
    hipStream_t q = ss ? ss->s[0] : st;
    if (ss) S2A_HIP(hipStreamWaitEvent(q, ss->fork, 0));
    int rc = chain_a(q, false);
    if (rc != S2A_OK) return rc;
    if (ss) S2A_HIP(hipEventRecord(ss->join[0], q));
  }
  if (need_c) {
    hipStream_t q = ss ? ss->s[1] : st;
    if (ss) S2A_HIP(hipStreamWaitEvent(q, ss->fork, 0));
    int rc = sort_c(q);
    if (rc != S2A_OK) return rc;
    // ONE join for the main stream: the output-order queue also waits for the score-order queue, so its event covers both
    // (every cross-stream wait is a ~6 us bubble in the waiting queue; the two side queues finish long before they are needed)
    if (ss && spatial && !deferred) S2A_HIP(hipStreamWaitEvent(q, ss->join[0], 0));
    if (ss) S2A_HIP(hipEventRecord(ss->join[1], q));
  } else {
    B.perm_glob = B.perm_seg;
    B.keyC_s = B.keyA_s;
  }
  k_nms_heavy<false><<<kHeavyGrid, kThreads, 0, st>>>(boxes, rankkey, thr, pair_list, B.C, pair_cap, B.edges, pl.edge_cap);
  k_nms_heavy<true><<<kPersistentGrid, kThreads, 0, st>>>(boxes, rankkey, thr, pair_list, B.C, pair_cap, B.edges, pl.edge_cap);
  // the pair list is dead after the dense pass: its memory takes the alive-edge list of the last launched round
  for (int r = 1; r <= kNmsRounds; r++)
    k_nms_round<<<256, kThreads, 0, st>>>(B.edges, B.C, pl.edge_cap, B.state, B.blocked, n, r,
                                          r == kNmsRounds ? B.gq : nullptr, pl.queue_cap);
  // still-alive edges (listed by the last launched round in the dead pair list): by segment into the edge buffer (the full
  // edge list is dead now), then one workgroup per segment; all five kernels return at once when nothing is alive
  k_nms_finish_segments<<<512, kFinThreads, 0, st>>>(B.C, B.seg_start, B.num_seg, B.gq, pl.queue_cap, B.edges, pl.edge_cap,
                                                     B.state, B.blocked, n, spatial ? B.keyB_s : B.keyA_s, spatial ? 20 : 32,
                                                     ignore_key, use_ignore, force_global);
  if (ss && need_c) S2A_HIP(hipStreamWaitEvent(st, ss->join[1], 0));
  else if (ss && spatial && !deferred) S2A_HIP(hipStreamWaitEvent(st, ss->join[0], 0));
  if (!deferred)
    k_nms_greedy_direct<<<512, kThreads, 0, st>>>(B.sorted, spatial ? B.seg_start_a : B.seg_start, spatial ? B.num_seg_a : B.num_seg,
                                                  B.keyA_s, ignore_key, use_ignore, B.C,
                                                  pair_cap, pl.edge_cap, pl.tile_cap, B.state_fb, thr);
  k_nms_finish<<<g, 256, 0, st>>>(B.state, B.blocked, B.C, perm, B.state_fb, B.perm_seg, n, B.keep_orig);
  S2A_LAUNCH_CHECK();
  return S2A_OK;
}

size_t nms_workspace_bytes(int64_t n, int64_t max_seg_rows) {
  if (n <= 0) return 256;
  NmsPlan pl;
  if (nms_plan(n, max_seg_rows, &pl) != 0) return 0;
  Carver cv(nullptr, 0);
  NmsBuffers B;
  nms_carve(cv, n, pl, &B);
  // (+ 64 KB: the segmented entry points want 48 KB of lists behind the fixed part, which the plan of a tiny n does not reach)
  return cv.off + 256 + (64u << 10);
}

#ifdef S2A_MEASURE
static int nms_debug_dump(const NmsBuffers& B, int64_t n, hipStream_t st) {
  if (!getenv("S2A_NMS_DEBUG")) return S2A_OK;   // measurement builds only: the device-side totals of this call
  NmsCounters h;
  S2A_HIP(hipStreamSynchronize(st));
  S2A_HIP(hipMemcpy(&h, B.C, sizeof(h), hipMemcpyDeviceToHost));
  unsigned long long dbg[8] = {};
  S2A_HIP(hipMemcpyFromSymbol(dbg, HIP_SYMBOL(g_cull_dbg), sizeof(dbg)));
  unsigned long long zero[8] = {};
  S2A_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_cull_dbg), zero, sizeof(zero)));
  if (dbg[6])
    fprintf(stderr, "[cull] waves %llu tiles %llu  cycles per wave: total %.0f  wait(claim+loads) %.0f  stage1 %.0f  list %.0f  stage2 %.0f\n",
            dbg[6], dbg[5], (double)dbg[0] / dbg[6], (double)dbg[1] / dbg[6], (double)dbg[2] / dbg[6], (double)dbg[3] / dbg[6],
            (double)dbg[4] / dbg[6]);
  unsigned long long fd[16] = {}, zero16[16] = {};
  S2A_HIP(hipMemcpyFromSymbol(fd, HIP_SYMBOL(g_fin_dbg), sizeof(fd)));
  S2A_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_fin_dbg), zero16, sizeof(zero16)));
  if (fd[6])
    fprintf(stderr, "[finish] segments > 1500 edges: %llu rounds; cycles per round (wave 0): pass 1 %.0f  barrier %.0f  pass 2 %.0f  barrier + counter %.0f\n",
            fd[6], (double)fd[7] / fd[6], (double)fd[8] / fd[6], (double)fd[9] / fd[6], (double)fd[10] / fd[6]);
  if (fd[6])
    fprintf(stderr, "[finish] of those rounds: %llu with lists (> %d edges), then edges left after the round: > 256: %llu, 65..256: %llu, 17..64: %llu, <= 16: %llu\n",
            fd[11], kFinThreads, fd[12], fd[13], fd[14], fd[15]);
  fprintf(stderr, "[finish] segments with work %llu: iterations max %llu sum %llu, edges max %llu sum %llu, rows max %llu\n", fd[3], fd[0], fd[2],
          fd[1], fd[5], fd[4]);
  fprintf(stderr, "[nms] n %lld tiles %llu pairs %llu edges %llu alive_list %llu status %u alive %u %u %u %u\n", (long long)n,
          h.tiles, h.pairs, h.edges, h.alive_list, h.status, h.alive[1], h.alive[2], h.alive[3], h.alive[4]);
  return S2A_OK;
}
#endif
// control slots of k_nms_small: one per call in flight (the kernel leaves its slot zeroed; a slot is handed out again only
// after the call that used it has been synchronised)
constexpr int kSmallDevices = 16;
std::mutex g_small_mutex;
uint32_t g_small_busy[kSmallDevices] = {};          // per device, bit s: slot s is in use
SmallCtl* g_small_dev[kSmallDevices] = {};
uint32_t* g_small_host[kSmallDevices] = {};          // pinned + mapped: 16 slots x {count, status, pad, pad}
uint32_t* g_small_host_dev[kSmallDevices] = {};      // the same memory as the device sees it
long long g_small_taken = 0, g_small_fallback = 0;   // calls settled by k_nms_small / handed on to the general path

int small_slot_acquire(int* slot, int* device, SmallCtl** dev) {
  int d = 0;
  S2A_HIP(hipGetDevice(&d));
  *slot = -1;
  *device = d;
  if (d < 0 || d >= kSmallDevices) return S2A_OK;
  std::lock_guard<std::mutex> lock(g_small_mutex);
  if (!g_small_dev[d]) {
    S2A_HIP(hipGetSymbolAddress(reinterpret_cast<void**>(&g_small_dev[d]), HIP_SYMBOL(g_small_ctl)));
    S2A_HIP(hipHostMalloc(reinterpret_cast<void**>(&g_small_host[d]), 16 * 4 * sizeof(uint32_t), hipHostMallocMapped));
    S2A_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&g_small_host_dev[d]), g_small_host[d], 0));
  }
  for (int s = 0; s < 16; s++)
    if (!(g_small_busy[d] & (1u << s))) { g_small_busy[d] |= 1u << s; *slot = s; *dev = g_small_dev[d] + s; return S2A_OK; }
  return S2A_OK;                     // (all sixteen in flight: the caller takes the general path)
}
void small_slot_release(int device, int slot, bool taken, bool count = true) {
  std::lock_guard<std::mutex> lock(g_small_mutex);
  g_small_busy[device] &= ~(1u << slot);
  if (count) (taken ? g_small_taken : g_small_fallback)++;
}

int nms_dropin(const float* dets, const float* scores, const float* labels, int64_t n, float thr,
               int64_t* keep, int64_t* count_dev, int64_t* host_count, void* ws, size_t ws_bytes,
               hipStream_t st) {
  S2A_CHECK_ARG(n >= 0 && n < (1ll << 31), "nms_rotated: n out of range");
  S2A_CHECK_ARG(count_dev != nullptr, "nms_rotated: count_dev must not be NULL");
  if (n == 0) {
    fill_u32(reinterpret_cast<uint32_t*>(count_dev), 0u, 2, st);
    if (host_count) *host_count = 0;
    return S2A_OK;
  }
  S2A_CHECK_ARG(dets && scores && keep, "nms_rotated: NULL tensor");
  // small synchronous calls: ONE launch (k_nms_small); anything outside its limits reports a status and falls through to
  // the general path below.  S2A_NMS_SMALL=0: A/B, tests
  {
    const char* e = std::getenv("S2A_NMS_SMALL");
    // (single class, labels == NULL: the rows are ONE segment -- beyond kSmallSeg rows the kernel could only report
    // "segment too long"; skipped on the host instead of paying its launch + synchronisation first.  A multi-label call
    // whose largest label exceeds kSmallSeg is only known on the device: it still pays that detour.)
    const bool small = host_count != nullptr && n <= kSmallN && !(labels == nullptr && n > kSmallSeg) &&
                       !(e && e[0] == '0') && !stream_capturing(st) &&
                       ws != nullptr && ws_bytes >= (size_t)n * 8 + 1024;
    if (small) {
      int slot = -1, device = 0;
      SmallCtl* ctl = nullptr;
      int rc = small_slot_acquire(&slot, &device, &ctl);
      if (rc != S2A_OK) return rc;
      if (slot >= 0) {
        auto kern = k_nms_small;
        static bool attr_set[kSmallDevices] = {};
        hipError_t he = hipSuccess;
        if (!attr_set[device]) {         // (once per device: the call is not free)
          he = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kSmallLds);
          attr_set[device] = he == hipSuccess;
        }
        uint32_t res[2] = {0u, 0xffffffffu};
        volatile uint32_t* hres = g_small_host[device] + 4 * slot;
        hres[0] = 0u;
        hres[1] = 0xffffffffu;
        if (he == hipSuccess) {
          kern<<<kSmallGrid, 1024, kSmallLds, st>>>(dets, scores, labels, (int)n, thr, static_cast<unsigned long long*>(ws), ctl,
                                                    g_small_host_dev[device] + 4 * slot, keep, count_dev);
          he = hipGetLastError();
        }
        if (he == hipSuccess) he = hipStreamSynchronize(st);
        res[0] = hres[0];
        res[1] = hres[1];
        if (he != hipSuccess) {
          // the slot may be dirty: clear it before anyone else gets it
          (void)hipMemset(ctl, 0, sizeof(SmallCtl));
          small_slot_release(device, slot, false);
          set_error("nms_rotated (small path) failed: %s", hipGetErrorString(he));
          return S2A_EHIP;
        }
        small_slot_release(device, slot, res[1] == 0u);
        if (res[1] == 0u) {
          *host_count = (int64_t)res[0];
          return S2A_OK;
        }
      }
    }
  }
  NmsPlan pl;
  S2A_CHECK_ARG(nms_plan(n, n, &pl) == 0, "nms_rotated: rocprim size query failed");
  Carver cv(ws, ws_bytes);
  NmsBuffers B;
  nms_carve(cv, n, pl, &B);
  if (cv.off > ws_bytes || !B.gq) {
    set_error("nms_rotated: workspace too small (%zu < %zu)", ws_bytes, cv.off);
    return S2A_EWORKSPACE;
  }
  // the count for a synchronous caller through a host-mapped word (a control slot of the small path's pool) when one is
  // free: the D2H copy behind the last kernel was one more launch (4 us) in the launch-paced tail.  With such a word the
  // score order of the spatial path (overflow fallback only) is deferred too: S2A_NMS_DEFER_A=0 builds it up front (A/B, tests)
  int slot = -1, device = 0;
  SmallCtl* ctl = nullptr;
  if (host_count && !stream_capturing(st)) {
    int rc2 = small_slot_acquire(&slot, &device, &ctl);
    if (rc2 != S2A_OK) return rc2;
  }
  bool defer_a = slot >= 0;
  if (const char* e = std::getenv("S2A_NMS_DEFER_A")) defer_a = defer_a && e[0] != '0';
  int rc = nms_core(dets, scores, labels, nullptr, nullptr, n, 0, 1, thr, pl, B, st, true, true, nullptr, &defer_a);
  if (rc != S2A_OK) {
    if (slot >= 0) small_slot_release(device, slot, false, false);
    return rc;
  }
  {
    const int rows = count_rows(n);
    const int nb = (int)((n + rows - 1) / rows);
    k_nms_keep_count<<<nb, 256, 0, st>>>(B.keep_orig, B.perm_glob, n, rows, B.cnt3);
    volatile uint32_t* hres = slot >= 0 ? g_small_host[device] + 4 * slot : nullptr;
    if (hres) { hres[0] = 0xffffffffu; hres[1] = 0u; }
    k_nms_keep_write<<<nb, 256, 0, st>>>(B.keep_orig, B.perm_glob, n, rows, B.cnt3, nb, keep, count_dev,
                                         slot >= 0 ? g_small_host_dev[device] + 4 * slot : nullptr,
                                         defer_a ? B.C : nullptr, pl.queue_cap, pl.edge_cap, pl.tile_cap);
    hipError_t he = hipGetLastError();
    if (he == hipSuccess && defer_a) {
      he = hipStreamSynchronize(st);
      if (he == hipSuccess && hres[1] != 0u) {      // a list overflowed: the direct form settles the call (score order now)
        int rcf = nms_overflow_fallback(dets, n, thr, pl, B, st);
        if (rcf != S2A_OK) { small_slot_release(device, slot, false, false); return rcf; }
        k_nms_keep_count<<<nb, 256, 0, st>>>(B.keep_orig, B.perm_glob, n, rows, B.cnt3);
        hres[0] = 0xffffffffu;
        k_nms_keep_write<<<nb, 256, 0, st>>>(B.keep_orig, B.perm_glob, n, rows, B.cnt3, nb, keep, count_dev,
                                             g_small_host_dev[device] + 4 * slot);
        he = hipGetLastError();
      }
    }
#ifdef S2A_MEASURE
    if (he == hipSuccess) { int rc_ = nms_debug_dump(B, n, st); if (rc_ != S2A_OK) { if (slot >= 0) small_slot_release(device, slot, false, false); return rc_; } }
#endif
    if (he == hipSuccess && host_count) {
      if (slot < 0) he = hipMemcpyAsync(host_count, count_dev, sizeof(int64_t), hipMemcpyDeviceToHost, st);
      if (he == hipSuccess) he = hipStreamSynchronize(st);
      if (he == hipSuccess && slot >= 0) *host_count = (int64_t)hres[0];
    }
    if (slot >= 0) small_slot_release(device, slot, false, false);
    if (he != hipSuccess) {
      set_error("nms_rotated failed: %s", hipGetErrorString(he));
      return S2A_EHIP;
    }
  }
  return S2A_OK;
}

}  // namespace
}  // namespace s2a

namespace s2a {
int build_flags_rotated() {
  int f = 0;
#ifdef S2A_MEASURE
  f |= 1;
#endif
#ifdef S2A_ABL_NOFENCE
  f |= 2;
#endif
  return f;
}
// ---- the greedy resolve by rounds over an EDGE list is metric-agnostic too: poly_ops.hip hands over the suppression edges
// of its polygon IoU (i = the higher-scored row, j = the row it suppresses; positions in descending-score order, ONE segment)
// and gets the keep flags of the original rows back -- four launched rounds, the per-segment clean-up, the view (the same
// kernels as the rotated NMS behind its exact pass).
namespace {
struct EdgeRoundsWs {
  NmsCounters C;
  uint32_t seg_start[4];
  uint32_t num_seg;
  uint32_t cnt[kSegCountBlocks];
};
__global__ void k_edge_rounds_init(EdgeRoundsWs* __restrict__ W, uint8_t* __restrict__ state, uint8_t* __restrict__ blocked,
                                   int64_t n, const unsigned long long* __restrict__ edge_count) {
  const size_t stride = (size_t)gridDim.x * blockDim.x, i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (size_t i = i0; i < (size_t)n; i += stride) { state[i] = (uint8_t)kOpen; blocked[i] = 0; blocked[n + i] = 0; }
  if (i0 == 0) {
    W->C = NmsCounters{};
    W->C.edges = *edge_count;
    W->seg_start[0] = 0u;
    W->seg_start[1] = (uint32_t)n;
    W->num_seg = 1u;
  }
}
}  // namespace
size_t nms_edge_rounds_workspace(int64_t n) {
  return align_up(sizeof(EdgeRoundsWs)) + align_up((size_t)n) + align_up(2 * (size_t)n) + 256;
}
int launch_nms_edge_rounds(uint2* edges, unsigned long long edge_cap, const unsigned long long* edge_count_dev, uint2* alive_list,
                           unsigned long long alive_cap, int64_t n, const int32_t* order, uint8_t* keep_orig, void* workspace,
                           size_t workspace_bytes, hipStream_t st) {
  S2A_CHECK_ARG(n > 0 && n < (1ll << 31), "nms edge rounds: n out of range");
  Carver cv(workspace, workspace_bytes);
  auto* W = cv.take<EdgeRoundsWs>(1);
  auto* state = cv.take<uint8_t>((size_t)n);
  auto* blocked = cv.take<uint8_t>(2 * (size_t)n);
  S2A_CHECK_ARG(W && state && blocked, "nms edge rounds: workspace too small");
  k_edge_rounds_init<<<grid_for(n), 256, 0, st>>>(W, state, blocked, n, edge_count_dev);
  for (int r = 1; r <= kNmsRounds; r++)
    k_nms_round<<<256, kThreads, 0, st>>>(edges, &W->C, edge_cap, state, blocked, n, r, r == kNmsRounds ? alive_list : nullptr, alive_cap);
  k_nms_finish_segments<<<1, kFinThreads, 0, st>>>(&W->C, W->seg_start, &W->num_seg, alive_list, alive_cap, edges, edge_cap, state,
                                                   blocked, n, nullptr, 0, 0u, 0, 0);
  k_nms_finish<<<grid_for(n), 256, 0, st>>>(state, blocked, &W->C, order, nullptr, nullptr, n, keep_orig);
  S2A_LAUNCH_CHECK();
  return S2A_OK;
}
// kept rows of `order` (rows by descending score) -> keep[], count (device + optional host-mapped word): two launches
int launch_keep_compact(const uint8_t* keep_orig, const int32_t* order, int64_t n, uint32_t* cnt_scratch /* kSegCountBlocks words */,
                        int64_t* keep, int64_t* count_dev, hipStream_t st) {
  const int rows = count_rows(n);
  const int nb = (int)((n + rows - 1) / rows);
  k_nms_keep_count<<<nb, 256, 0, st>>>(keep_orig, order, n, rows, cnt_scratch);
  k_nms_keep_write<<<nb, 256, 0, st>>>(keep_orig, order, n, rows, cnt_scratch, nb, keep, count_dev, nullptr);
  S2A_LAUNCH_CHECK();
  return S2A_OK;
}
size_t keep_compact_scratch_words() { return (size_t)kSegCountBlocks; }

// the greedy scan is metric-agnostic: poly_ops.hip (chip-merge NMS on polygon IoU) reuses it
int launch_nms_scan(const unsigned long long* mask, const uint32_t* seg_start, const uint32_t* num_seg,
                    const unsigned long long* mask_off, const uint32_t* nblk, const int32_t* perm_seg,
                    uint8_t* keep_orig, uint32_t max_blocks, const unsigned long long* words_total,
                    unsigned long long words_bound, uint32_t* status, hipStream_t st) {
  S2A_CHECK_ARG(((size_t)max_blocks + 10 + kScanCacheWords) * 8 <= 112 * 1024, "nms scan: segment too large");
  size_t lds = ((size_t)max_blocks + 10 + kScanCacheWords) * sizeof(unsigned long long);
  S2A_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_nms_scan), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  k_nms_scan<<<512, kThreads, lds, st>>>(mask, seg_start, num_seg, mask_off, nblk, perm_seg, keep_orig, max_blocks,
                                         words_total, words_bound, status, nullptr);
  S2A_LAUNCH_CHECK();
  return S2A_OK;
}
}  // namespace s2a

using namespace s2a;

extern "C" size_t s2a_box_iou_rotated_workspace_bytes(int64_t n, int64_t m) {
  if (n <= 0 || m <= 0) return 256;
  unsigned long long pairs = (unsigned long long)n * (unsigned long long)m;
  unsigned long long cap = std::max<unsigned long long>(std::min<unsigned long long>(pairs, kIouQueueCap), (unsigned long long)m * 256);
  return align_up((size_t)(n + m) * sizeof(PreBox)) * 2 + align_up(cap * (sizeof(uint2) + sizeof(float))) + 8192;
}

namespace s2a {
namespace {
constexpr unsigned long long kIouForkBytes = 8ull << 20;   // smaller outputs: one stream, zero-fill fused into the cull
}  // namespace
}  // namespace s2a

extern "C" int s2a_box_iou_rotated(const float* boxes1, int64_t n, const float* boxes2, int64_t m,
                                   float* ious, void* workspace, size_t workspace_bytes,
                                   s2a_stream_t stream) {
  S2A_CHECK_ARG(n >= 0 && m >= 0, "box_iou_rotated: negative size");
  if (n == 0 || m == 0) return S2A_OK;  // reference: empty [N,M] (cuda.cu:79)
  S2A_CHECK_ARG(boxes1 && boxes2 && ious, "box_iou_rotated: NULL tensor");
  S2A_CHECK_ARG(n < (1ll << 31) && m < (1ll << 31), "box_iou_rotated: size out of range");
  hipStream_t st = as_stream(stream);
  Carver cv(workspace, workspace_bytes);
  PreBox* P1 = cv.take<PreBox>((size_t)n);
  PreBox* P2 = cv.take<PreBox>((size_t)m);
  unsigned long long* counters = cv.take<unsigned long long>(512);
  if (!P1 || !P2 || !counters ||
      cv.off + (size_t)m * 256 * (sizeof(uint2) + sizeof(float)) + 512 > workspace_bytes) {
    set_error("box_iou_rotated: workspace too small (%zu bytes)", workspace_bytes);
    return S2A_EWORKSPACE;
  }
  unsigned long long cap = (workspace_bytes - cv.off - 256) / (sizeof(uint2) + sizeof(float));
  uint2* gq = reinterpret_cast<uint2*>(static_cast<char*>(workspace) + cv.off);
  float* vals = reinterpret_cast<float*>(static_cast<char*>(workspace) + align_up(cv.off + cap * sizeof(uint2)) );
  cap = std::min<unsigned long long>(cap, (workspace_bytes - (size_t)(reinterpret_cast<char*>(vals) - static_cast<char*>(workspace))) / sizeof(float));
  // rows per chunk: sized for up to 1/4 of the pairs surviving the cull (DOTA-like inputs: ~1 %);
  // a denser chunk overflows the list and is recomputed pair by pair in k_iou_scatter
  int64_t rows_per_chunk = (int64_t)std::min<unsigned long long>((unsigned long long)n, 4 * (cap / (unsigned long long)m));
  if (rows_per_chunk < n) rows_per_chunk = std::max<int64_t>(256, rows_per_chunk / 256 * 256);   // (else: one chunk)
  int64_t chunks = (n + rows_per_chunk - 1) / rows_per_chunk;
  S2A_CHECK_ARG(chunks <= 512, "box_iou_rotated: workspace too small for %lld x %lld", (long long)n, (long long)m);
  // Large outputs are write-bound (4 B per pair, ~1 % of DOTA-like pairs overlap): the zero-fill of the whole matrix
  // runs on a side stream at the full store rate while this stream finds the overlapping pairs and evaluates them into a
  // compact buffer; after the join a small kernel drops the values into place.  (Round 1 stored the zeros from the cull
  // kernel -- 4 TB/s beside its circle tests -- and ran the dense pass after it: 212 us at 10 k x 10 k.)
  const bool big = (unsigned long long)n * (unsigned long long)m * 4ull >= kIouForkBytes;
  const char* ef = getenv("S2A_IOU_FORK");                 // A/B: 0 = never fork, 1 = always
  const bool fork = ef && (ef[0] == '0' || ef[0] == '1') ? ef[0] == '1' : big;
  SideSet* ss = nullptr;
  hipStream_t side = nullptr;
  if (fork) {
    int rc = side_set(st, &ss);
    if (rc != S2A_OK) return rc;
    side = ss->s[0];
    S2A_HIP(hipEventRecord(ss->fork, st));
    S2A_HIP(hipStreamWaitEvent(side, ss->fork, 0));
    // The fill is PACED (s_sleep between the stores of a wave): flat out it finishes 400 MB in 50-100 us but saturates the
    // memory system, and every dependent read of the kernels beside it then takes ~10 us -- the pair finder stretched
    // from 59 to 105-125 us whatever its design and whatever the fill's workgroup count (docs/HISTORY.md, round 3).  At 512
    // workgroups and s_sleep 12 it moves ~3.5 TB/s (113 us at 10 k x 10 k), ends before the chain beside it does (pair
    // finding 62 + exact pass 60-65 us, both at their stand-alone speed) and the call takes ~157 us instead of ~195.
    int fill_wgs = 512, pace = 12;
#ifdef S2A_MEASURE
    if (const char* fw = getenv("S2A_IOU_FILL_WGS")) fill_wgs = atoi(fw);     // measurement builds only; 0 = no fill (WRONG results)
    if (const char* fp = getenv("S2A_IOU_FILL_PACE")) pace = atoi(fp);
#endif
    const unsigned long long nm = (unsigned long long)n * (unsigned long long)m;
    if (fill_wgs > 0) {
      if (pace >= 12) k_fill_zero<12><<<fill_wgs, 256, 0, side>>>(ious, nm);
      else if (pace >= 8) k_fill_zero<8><<<fill_wgs, 256, 0, side>>>(ious, nm);
      else if (pace >= 6) k_fill_zero<6><<<fill_wgs, 256, 0, side>>>(ious, nm);
      else if (pace >= 4) k_fill_zero<4><<<fill_wgs, 256, 0, side>>>(ious, nm);
      else if (pace >= 2) k_fill_zero<2><<<fill_wgs, 256, 0, side>>>(ious, nm);
      else k_fill_zero<0><<<fill_wgs, 256, 0, side>>>(ious, nm);
    }
    S2A_HIP(hipEventRecord(ss->join[0], side));
  }
  k_prep_boxes2<<<(unsigned)((std::max<int64_t>(n + m, 512) + 255) / 256), 256, 0, st>>>(boxes1, n, P1, boxes2, m, P2, counters, 512);
  const char* ec = getenv("S2A_IOU_CULL_COLS");            // A/B: round 1's column-major cull beside the forked fill
  const bool cull_cols = ec && ec[0] == '1';
  for (int64_t c = 0; c < chunks; c++) {
    int64_t r0 = c * rows_per_chunk, r1 = std::min(n, r0 + rows_per_chunk);
    dim3 grid_c((unsigned)((m + kThreads * 4 - 1) / (kThreads * 4)), (unsigned)((r1 - r0 + kIouRowsPerWg - 1) / kIouRowsPerWg));
    if (fork && !cull_cols) {
      // columns per workgroup: enough workgroups for ~8 per CU, at least 256 columns each
      const int64_t rwg = (r1 - r0 + kThreads - 1) / kThreads;
      int64_t cw = 256;
      while (cw < m && rwg * ((m + cw - 1) / cw) > 4096) cw *= 2;
      k_iou_cull_lanes<<<dim3((unsigned)((m + cw - 1) / cw), (unsigned)rwg), kThreads, 0, st>>>(P1, P2, r0, r1, m, (int)cw, gq,
                                                                                            counters + c, cap);
    }
    else if (fork)
      k_iou_cull<false><<<grid_c, kThreads, 0, st>>>(P1, P2, r0, r1, m, ious, gq, counters + c, cap);
    else
      k_iou_cull<true><<<grid_c, kThreads, 0, st>>>(P1, P2, r0, r1, m, ious, gq, counters + c, cap);
    k_iou_heavy<<<kHeavyGrid, kThreads, 0, st>>>(P1, P2, r0, gq, counters + c, cap, vals);
    if (fork && c == 0) S2A_HIP(hipStreamWaitEvent(st, ss->join[0], 0));
    k_iou_scatter<<<kPersistentGrid, kThreads, 0, st>>>(P1, P2, r0, r1, m, ious, gq, counters + c, cap, vals);
  }
  S2A_LAUNCH_CHECK();
  return S2A_OK;
}

// ================================================================= label assignment (training side)
// assign_labels (models/utils.py:33-147): the real consumer of box_iou_rotated in the reference.  The [M,N]
// matrix comes from the cull -> pair list -> dense pass pipeline above (a per-anchor loop over the gts was 10x
// slower: divergence); three streaming passes over it replace the reference's dozen tensor ops and its
// Python loop over the gts: (1) one wave per anchor row: validity / range filters in place, row max and
// first arg-max, rules 1 and 2(1); (2) column maxima (tiled, atomicMax on order-preserving keys);
// (3) one wave per row: the LAST gt whose column maximum the anchor attains (the reference's ascending loop
// lets later gts overwrite earlier ones, :131-145).
namespace s2a {
namespace {
__device__ __forceinline__ int iou_key(float v) { return v < 0.f ? 0 : __float_as_int(v) + 1; }
__device__ __forceinline__ float key_iou(int k) { return k == 0 ? -0.5f : __int_as_float(k - 1); }

__device__ __forceinline__ bool anchor_valid(const float* __restrict__ a, float img_h, float img_w) {
  return a[0] >= 0 && a[1] >= 0 && a[0] <= img_w && a[1] <= img_h && a[2] < img_w && a[3] < img_h;   // :63-69
}

__global__ __launch_bounds__(256) void k_assign_rows(const float* __restrict__ anchors, float* __restrict__ ious,
                                                     int64_t M, int64_t N, float img_h, float img_w, float pos_thr,
                                                     float neg_thr, int filt_anchor, int filt_iou,
                                                     int64_t* __restrict__ assign) {
  const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (m >= M) return;
  const bool valid = !filt_anchor || anchor_valid(anchors + 5 * m, img_h, img_w);
  float* row = ious + m * N;
  float best = -2.f;
  int64_t arg = 0;
  for (int64_t j = lane; j < N; j += 64) {
    float v = row[j];
    if (filt_iou && !(v >= 0.f && v <= 1.f)) v = -0.5f;        // :86-93
    if (!valid) v = -0.5f;                                     // :97-98
    v += 0.f;                                                  // -0.0 -> +0.0
    row[j] = v;
    if (v > best) { best = v; arg = j; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o);
    const int64_t oa = __shfl_xor(arg, o);
    if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }   // first index of the maximum
  }
  if (lane == 0) {
    int64_t a = -2;
    if (best >= 0.f && best < neg_thr) a = -1;                 // :108
    if (best >= pos_thr) a = arg;                              // :114-115
    assign[m] = a;
  }
}

// grid (ceil(N/64), ceil(M/256)): thread = one column over 256 rows; rows are read 64 columns wide (coalesced)
__global__ __launch_bounds__(64) void k_assign_colmax(const float* __restrict__ ious, int64_t M, int64_t N,
                                                      int* __restrict__ gt_key) {
  const int64_t j = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (j >= N) return;
  const int64_t m0 = (int64_t)blockIdx.y * 256, m1 = min(M, m0 + 256);
  int k = 0;
  for (int64_t m = m0; m < m1; m++) k = max(k, iou_key(ious[m * N + j]));
  if (k > 0) atomicMax(gt_key + j, k);
}

__global__ __launch_bounds__(64) void k_assign_colarg(const float* __restrict__ ious, int64_t M, int64_t N,
                                                      const int* __restrict__ gt_key, int* __restrict__ gt_arg) {
  const int64_t j = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (j >= N) return;
  const int64_t m0 = (int64_t)blockIdx.y * 256, m1 = min(M, m0 + 256);
  const int k = gt_key[j];
  for (int64_t m = m0; m < m1; m++)
    if (iou_key(ious[m * N + j]) == k) { atomicMin(gt_arg + j, (int)m); break; }   // first row of this chunk
}

__global__ __launch_bounds__(256) void k_assign_cols(const float* __restrict__ ious, int64_t M, int64_t N,
                                                     float min_pos_thr, const int* __restrict__ gt_key,
                                                     const int* __restrict__ gt_arg, int64_t* __restrict__ assign) {
  const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (m >= M) return;
  const float* row = ious + m * N;
  int64_t best = -1;
  for (int64_t j = lane; j < N; j += 64) {
    const int k = gt_key[j];
    if (!(key_iou(k) > min_pos_thr)) continue;                 // :126
    const bool hit = gt_arg ? gt_arg[j] == (int)m : iou_key(row[j]) == k;
    if (hit) best = j;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) best = max(best, __shfl_xor(best, o));
  if (lane == 0 && best >= 0) assign[m] = best;
}

// ---- LIST form (round 6): no [M,N] matrix.  The matrix form above is eleven launches (the IoU pipeline alone five) and a
// strided column-maximum pass: 146 us at 32 gts, 211 us at 300 -- slower than the unfused pair at the sizes models/utils.py:33
// sees.  Here, for N <= kAtMaxN gts and M x N pairs that fit the list:
//   k_assign_cull   a workgroup owns 16 anchor rows with ALL gts in LDS: circle + separating-axis test of its 16 x N pairs,
//                   the survivors into an LDS list, ONE global reservation per workgroup, (row, gt) pairs out;
//   k_assign_exact  exact IoU of the listed pairs, every lane busy and the whole chip balanced (the 64 P7 anchors of a chip
//                   overlap most gts: owned by one workgroup they took 0.5 ms): value kept in the list; row maximum / first
//                   arg-max, the count of filtered entries and the column maxima by atomics;
//   k_assign_rule3  with the final column maxima: the pairs that attain one -> rule 3 (all anchors), or the first such row;
//   k_assign_rows   rules 1, 2 (and 3) per anchor; k_assign_last when one anchor per gt is taken.
// Exact zeros never enter the list (a disjoint pair is 0.0 by the reference's own num == 0 return), so the row rules start
// from "all zero" and only count filtered (negative) entries.  Needs min_pos_iou_thr >= 0 and pos_iou_thr > 0.
constexpr int kAtRows = 16, kAtMaxN = 1024;
constexpr unsigned long long kAtMaxPairs = 8ull << 20;       // list capacity = M x N (never overflows), at most this
struct AtWs {                                                // carved from the workspace, in this order
  PreBox* gt_pre; int* gt_key; int* gt_arg; unsigned long long* rowbest; int* nbad; int* r3; unsigned long long* count;
  uint2* pair; float* val;
};
__global__ void k_assign_list_init(const float* __restrict__ gt, int N, int64_t M, AtWs w) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) {
    const float* g = gt + 5 * i;
    w.gt_pre[i] = make_prebox(g[0], g[1], g[2], g[3], g[4], 0.f);
    w.gt_key[i] = 0;
    w.gt_arg[i] = 0x7fffffff;
  }
  if (i < M) { w.rowbest[i] = 0ull; w.nbad[i] = 0; w.r3[i] = -1; }
  if (i == 0) *w.count = 0ull;
}

__global__ __launch_bounds__(256) void k_assign_cull(const float* __restrict__ anchors, int64_t M, int N, float img_h, float img_w,
                                                     int filt_anchor, AtWs w) {
  extern __shared__ __attribute__((aligned(16))) char smem_at[];
  PreBox* s_gt = reinterpret_cast<PreBox*>(smem_at);                                       // [N]
  uint32_t* s_list = reinterpret_cast<uint32_t*>(smem_at + (size_t)N * sizeof(PreBox));    // [kAtRows * N]
  __shared__ PreBox s_anc[kAtRows];
  __shared__ uint8_t s_valid[kAtRows];
  __shared__ unsigned s_cnt;
  __shared__ unsigned long long s_base;
  const int tid = threadIdx.x, lane = tid & 63;
  const int64_t m0 = (int64_t)blockIdx.x * kAtRows;
  for (int j = tid; j < N; j += 256) s_gt[j] = w.gt_pre[j];
  if (tid < kAtRows) {
    const int64_t m = m0 + tid;
    bool valid = false;
    PreBox A = {};
    if (m < M) {
      const float* a = anchors + 5 * m;
      valid = !filt_anchor || anchor_valid(a, img_h, img_w);     // invalid anchors: every overlap is -0.5 (:97-98), nothing to list
      A = make_prebox(a[0], a[1], a[2], a[3], a[4], 0.f);
    }
    s_anc[tid] = A;
    s_valid[tid] = valid ? 1 : 0;
  }
  if (tid == 0) s_cnt = 0;
  __syncthreads();
  const int total = kAtRows * N;
  for (int i0 = 0; i0 < total; i0 += 256) {
    const int idx = i0 + tid;
    bool hit = false;
    int r = 0, j = 0;
    if (idx < total) {
      r = idx / N; j = idx - r * N;
      if (s_valid[r]) {
        const PreBox& A = s_anc[r];
        const PreBox& B = s_gt[j];
        hit = !surely_disjoint(A.x, A.y, A.r, B.x, B.y, B.r) && !sat_disjoint(A, B);
      }
    }
    const unsigned long long bal = __ballot(hit);
    if (bal) {
      unsigned base = 0;
      if (lane == 0) base = atomicAdd(&s_cnt, (unsigned)__popcll(bal));
      base = (unsigned)__shfl((int)base, 0);
      if (hit) s_list[base + (unsigned)__popcll(bal & ((1ull << lane) - 1ull))] = ((uint32_t)r << 16) | (uint32_t)j;
    }
  }
  __syncthreads();
  const unsigned cnt = s_cnt;
  if (cnt == 0) return;                                          // (uniform)
  if (tid == 0) s_base = atomicAdd(w.count, (unsigned long long)cnt);
  __syncthreads();
  const unsigned long long base = s_base;
  for (unsigned e = tid; e < cnt; e += 256) {
    const uint32_t rj = s_list[e];
    w.pair[base + e] = make_uint2((unsigned)(m0 + (rj >> 16)), rj & 0xffffu);
  }
}

__global__ __launch_bounds__(256) void k_assign_exact(const float* __restrict__ anchors, int filt_iou, AtWs w) {
  __shared__ float2 s_pts[24 * 256];
  const unsigned long long total = *w.count;
  for (unsigned long long e = (unsigned long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (unsigned long long)gridDim.x * 256) {
    const uint2 mj = w.pair[e];
    const float* a = anchors + 5 * (int64_t)mj.x;
    const PreBox A = make_prebox(a[0], a[1], a[2], a[3], a[4], 0.f);
    float v = rbox_iou<256>(A, w.gt_pre[mj.y], s_pts + threadIdx.x);
    if (filt_iou && !(v >= 0.f && v <= 1.f)) v = -0.5f;          // :86-93
    v += 0.f;                                                    // -0.0 -> +0.0
    w.val[e] = v;
    if (v < 0.f) atomicAdd(w.nbad + mj.x, 1);
    else if (v > 0.f) {
      const int k = iou_key(v);
      atomicMax(w.rowbest + mj.x, ((unsigned long long)(uint32_t)k << 32) | (unsigned long long)(0xffffffffu - mj.y));
      atomicMax(w.gt_key + mj.y, k);
    }
  }
}

__global__ __launch_bounds__(256) void k_assign_rule3(float min_pos_thr, int assign_all, AtWs w) {
  const unsigned long long total = *w.count;
  for (unsigned long long e = (unsigned long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (unsigned long long)gridDim.x * 256) {
    const float v = w.val[e];
    if (!(v > 0.f)) continue;
    const uint2 mj = w.pair[e];
    const int kc = w.gt_key[mj.y];
    if (key_iou(kc) > min_pos_thr && iou_key(v) == kc) {         // :126: this anchor attains the gt's maximum
      if (assign_all) atomicMax(w.r3 + mj.x, (int)mj.y);
      else atomicMin(w.gt_arg + mj.y, (int)mj.x);
    }
  }
}

__global__ void k_assign_rows_list(const float* __restrict__ anchors, int64_t M, int N, float img_h, float img_w, float pos_thr,
                                   float neg_thr, int filt_anchor, int assign_all, AtWs w, int64_t* __restrict__ assign) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const bool valid = !filt_anchor || anchor_valid(anchors + 5 * m, img_h, img_w);
  const unsigned long long b = w.rowbest[m];
  float best = 0.f;                                              // every entry that was not listed is an exact zero
  int64_t arg = 0;
  if (!valid || w.nbad[m] == N) best = -0.5f;                    // :97-98 / every overlap filtered
  else if (b != 0ull) { best = key_iou((int)(b >> 32)); arg = (int64_t)(0xffffffffu - (uint32_t)(b & 0xffffffffull)); }
  int64_t a = -2;
  if (best >= 0.f && best < neg_thr) a = -1;                     // :108
  if (best >= pos_thr) a = arg;                                  // :114-115
  if (assign_all && w.r3[m] >= 0) a = w.r3[m];                   // :131-145: the last gt whose maximum the anchor attains
  assign[m] = a;
}

// one anchor per gt (gt_max_assign_all = False): gt j's anchor is the first row attaining its maximum (gt_arg); the
// reference's ascending loop lets a later gt overwrite an earlier one on the same anchor
__global__ __launch_bounds__(1024) void k_assign_last(const int* __restrict__ gt_key, const int* __restrict__ gt_arg, int N,
                                                      float min_pos_thr, int64_t* __restrict__ assign) {
  __shared__ int s_m[kAtMaxN];
  const int j = threadIdx.x;
  int mine = -1;
  if (j < N && key_iou(gt_key[j]) > min_pos_thr && gt_arg[j] != 0x7fffffff) mine = gt_arg[j];
  if (j < kAtMaxN) s_m[j] = mine;
  __syncthreads();
  if (mine < 0) return;
  for (int k = j + 1; k < N; k++)
    if (s_m[k] == mine) return;
  assign[mine] = j;
}

// no gt boxes (:72-80): valid anchors are negatives, the others stay ignored
__global__ void k_assign_empty(const float* __restrict__ anchors, int64_t M, float img_h, float img_w, int filt_anchor,
                               int64_t* __restrict__ assign) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  assign[m] = (!filt_anchor || anchor_valid(anchors + 5 * m, img_h, img_w)) ? -1 : -2;
}
}  // namespace
}  // namespace s2a

extern "C" size_t s2a_assign_labels_workspace_bytes(int64_t num_anchors, int64_t num_gts) {
  const int64_t M = std::max<int64_t>(num_anchors, 1), N = std::max<int64_t>(num_gts, 1);
  const size_t matrix_form = align_up((size_t)M * N * 4) + 2 * align_up((size_t)N * 4) + s2a_box_iou_rotated_workspace_bytes(M, N) + 1024;
  const size_t list_form = align_up((size_t)N * 32) + 2 * align_up((size_t)N * 4) + align_up((size_t)M * 8) + 2 * align_up((size_t)M * 4) +
                           256 + align_up((size_t)M * N * 8) + align_up((size_t)M * N * 4) + 1024;
  return std::max(matrix_form, (unsigned long long)M * (unsigned long long)N <= (8ull << 20) ? list_form : (size_t)0);
}

extern "C" int s2a_assign_labels(const float* anchors, int64_t num_anchors, const float* gt_boxes, int64_t num_gts,
                                 float img_h, float img_w, float pos_iou_thr, float neg_iou_thr, float min_pos_iou_thr,
                                 int gt_max_assign_all, int filter_invalid_anchors, int filter_invalid_ious,
                                 int64_t* assign_gt_ids, void* workspace, size_t workspace_bytes, s2a_stream_t stream) {
  S2A_CHECK_ARG(num_anchors >= 0 && num_gts >= 0 && num_anchors < (1ll << 31) && num_gts < (1ll << 31),
                "assign_labels: sizes out of range");
  if (num_anchors == 0) return S2A_OK;
  S2A_CHECK_ARG(anchors && assign_gt_ids && (gt_boxes || num_gts == 0), "assign_labels: NULL tensor");
  hipStream_t st = as_stream(stream);
  const int64_t M = num_anchors, N = num_gts;
  if (N == 0) {
    k_assign_empty<<<(unsigned)((M + 255) / 256), 256, 0, st>>>(anchors, M, img_h, img_w, filter_invalid_anchors, assign_gt_ids);
    S2A_LAUNCH_CHECK();
    return S2A_OK;
  }
  // list form: no IoU matrix, five short launches (see k_assign_cull).  S2A_ASSIGN_LIST=0: the matrix form (A/B, tests)
  {
    const char* e = std::getenv("S2A_ASSIGN_LIST");
    const unsigned long long mn = (unsigned long long)M * (unsigned long long)N;
    if (N <= kAtMaxN && mn <= kAtMaxPairs && min_pos_iou_thr >= 0.f && pos_iou_thr > 0.f && !(e && e[0] == '0')) {
      Carver cvt(workspace, workspace_bytes);
      AtWs w;
      w.gt_pre = cvt.take<PreBox>((size_t)N);
      w.gt_key = cvt.take<int>((size_t)N);
      w.gt_arg = cvt.take<int>((size_t)N);
      w.rowbest = cvt.take<unsigned long long>((size_t)M);
      w.nbad = cvt.take<int>((size_t)M);
      w.r3 = cvt.take<int>((size_t)M);
      w.count = cvt.take<unsigned long long>(1);
      w.pair = cvt.take<uint2>((size_t)mn);
      w.val = cvt.take<float>((size_t)mn);
      if (!w.val || !w.pair || !w.count || !w.r3 || !w.nbad || !w.rowbest || !w.gt_arg || !w.gt_key || !w.gt_pre) {
        set_error("assign_labels: workspace too small (%zu < %zu)", workspace_bytes, cvt.off);
        return S2A_EWORKSPACE;
      }
      const size_t lds = (size_t)N * sizeof(PreBox) + (size_t)kAtRows * N * 4;
      static bool attr_set = false;
      if (!attr_set) {
        S2A_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_assign_cull), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(kAtMaxN * sizeof(PreBox) + (size_t)kAtRows * kAtMaxN * 4)));
        attr_set = true;
      }
      const int64_t mx = std::max<int64_t>(M, N);
      k_assign_list_init<<<(unsigned)((mx + 255) / 256), 256, 0, st>>>(gt_boxes, (int)N, M, w);
      k_assign_cull<<<(unsigned)((M + kAtRows - 1) / kAtRows), 256, lds, st>>>(anchors, M, (int)N, img_h, img_w, filter_invalid_anchors, w);
      k_assign_exact<<<1024, 256, 0, st>>>(anchors, filter_invalid_ious, w);
      k_assign_rule3<<<1024, 256, 0, st>>>(min_pos_iou_thr, gt_max_assign_all, w);
      k_assign_rows_list<<<(unsigned)((M + 255) / 256), 256, 0, st>>>(anchors, M, (int)N, img_h, img_w, pos_iou_thr, neg_iou_thr,
                                                                      filter_invalid_anchors, gt_max_assign_all, w, assign_gt_ids);
      if (!gt_max_assign_all)
        k_assign_last<<<1, 1024, 0, st>>>(w.gt_key, w.gt_arg, (int)N, min_pos_iou_thr, assign_gt_ids);
      S2A_LAUNCH_CHECK();
      return S2A_OK;
    }
  }
  Carver cv(workspace, workspace_bytes);
  float* ious = cv.take<float>((size_t)M * N);
  int* gt_key = cv.take<int>((size_t)N);
  int* gt_arg = cv.take<int>((size_t)N);
  const size_t iou_ws = s2a_box_iou_rotated_workspace_bytes(M, N);
  char* iw = cv.take<char>(iou_ws);
  if (!ious || !gt_key || !gt_arg || !iw) {
    set_error("assign_labels: workspace too small (%zu < %zu)", workspace_bytes, cv.off);
    return S2A_EWORKSPACE;
  }
  int rc = s2a_box_iou_rotated(anchors, M, gt_boxes, N, ious, iw, iou_ws, stream);
  if (rc != S2A_OK) return rc;
  fill_u32(gt_key, 0u, (size_t)N, st);
  const unsigned gr = (unsigned)((M + 3) / 4);
  k_assign_rows<<<gr, 256, 0, st>>>(anchors, ious, M, N, img_h, img_w, pos_iou_thr, neg_iou_thr, filter_invalid_anchors,
                                    filter_invalid_ious, assign_gt_ids);
  dim3 gc((unsigned)((N + 63) / 64), (unsigned)((M + 255) / 256));
  k_assign_colmax<<<gc, 64, 0, st>>>(ious, M, N, gt_key);
  if (!gt_max_assign_all) {
    fill_u32(gt_arg, 0x7f7f7f7fu, (size_t)N, st);
    k_assign_colarg<<<gc, 64, 0, st>>>(ious, M, N, gt_key, gt_arg);
  }
  k_assign_cols<<<gr, 256, 0, st>>>(ious, M, N, min_pos_iou_thr, gt_key, gt_max_assign_all ? nullptr : gt_arg, assign_gt_ids);
  S2A_LAUNCH_CHECK();
  return S2A_OK;
}

extern "C" int s2a_box_iou_rotated_pairs(const float* boxes1, const float* boxes2, int64_t n,
                                         float* ious, s2a_stream_t stream) {
  S2A_CHECK_ARG(n >= 0, "box_iou_rotated_pairs: negative size");
  if (n == 0) return S2A_OK;
  S2A_CHECK_ARG(boxes1 && boxes2 && ious, "box_iou_rotated_pairs: NULL tensor");
  k_iou_pairs<<<(unsigned)((n + kThreads - 1) / kThreads), kThreads, 0, as_stream(stream)>>>(boxes1, boxes2, n, ious);
  S2A_LAUNCH_CHECK();
  return S2A_OK;
}

// ---------------------------------------------------------------- multiclass candidates
// utils/bbox_nms_rotated.py:29-42 for a whole batch, static shapes, no host round trip:
// (box, class) pairs with score > thr, in row-major (image, box, class) order, compacted into
// `cap` slots; unused slots are padding rows (segment id -1, ignored by the segmented NMS).
namespace s2a {
namespace {
// Ordered compaction of the flat indices with score > thr (row-major order == the reference's boolean-mask order,
// utils/bbox_nms_rotated.py:29-40), three plain kernels: per-block counts, scan of the block counts by one workgroup,
// ordered scatter.  (rocprim::select did this in round 1; its partition kernel takes a by-value closure argument that
// the ROCm 7.2 runtime mishandles when a captured HIP graph containing it is replayed a second time inside a larger
// graph -- HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION, gone with DEBUG_CLR_GRAPH_PACKET_CAPTURE=0; detect() has to
// replay from a graph, so the capturable path does not go through it.)
constexpr int kCandItems = 8, kCandBlock = 256 * kCandItems;

__device__ __forceinline__ unsigned cand_flags(const float* __restrict__ scores, float thr, int64_t base, int64_t total) {
  unsigned f = 0;
#pragma unroll
  for (int k = 0; k < kCandItems; k++)
    if (base + k < total && scores[base + k] > thr) f |= 1u << k;
  return f;
}

__global__ __launch_bounds__(256) void k_cand_count(const float* __restrict__ scores, float thr, int64_t total,
                                                    uint32_t* __restrict__ block_cnt) {
  __shared__ unsigned s_w[4];
  const int64_t base = (int64_t)blockIdx.x * kCandBlock + (int64_t)threadIdx.x * kCandItems;
  unsigned c = __popc(cand_flags(scores, thr, base, total));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) block_cnt[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// exclusive scan of nb block counts in place (one workgroup, 1024 per sweep with a running carry); total -> count
__global__ __launch_bounds__(1024) void k_cand_scan(uint32_t* __restrict__ block_cnt, int64_t nb,
                                                    unsigned long long* __restrict__ count) {
  __shared__ unsigned s_w[16];
  __shared__ unsigned long long s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t b0 = 0; b0 < nb; b0 += 1024) {
    const int64_t b = b0 + threadIdx.x;
    const unsigned v = b < nb ? block_cnt[b] : 0u;
    unsigned incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned t = __shfl_up(incl, o);
      if (lane >= o) incl += t;
    }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    unsigned wbase = 0;
    for (int w = 0; w < wave; w++) wbase += s_w[w];
    const unsigned long long carry = s_carry;
    if (b < nb) block_cnt[b] = (uint32_t)(carry + wbase + incl - v);     // < 2^31 (total is)
    __syncthreads();
    if (threadIdx.x == 1023) s_carry = carry + wbase + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) *count = s_carry;
}

__global__ __launch_bounds__(256) void k_cand_scatter(const float* __restrict__ scores, float thr, int64_t total,
                                                      const uint32_t* __restrict__ block_off,
                                                      int32_t* __restrict__ sel) {
  __shared__ unsigned s_w[4];
  const int64_t base = (int64_t)blockIdx.x * kCandBlock + (int64_t)threadIdx.x * kCandItems;
  const unsigned f = cand_flags(scores, thr, base, total);
  const unsigned c = __popc(f);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned incl = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned t = __shfl_up(incl, o);
    if (lane >= o) incl += t;
  }
  if (lane == 63) s_w[wave] = incl;
  __syncthreads();
  unsigned off = block_off[blockIdx.x] + incl - c;
  for (int w = 0; w < wave; w++) off += s_w[w];
#pragma unroll
  for (int k = 0; k < kCandItems; k++)
    if (f & (1u << k)) sel[off++] = (int32_t)(base + k);
}

__global__ void k_gather_candidates(const float* __restrict__ boxes5, const float* __restrict__ scores,
                                    const int32_t* __restrict__ sel, const unsigned long long* __restrict__ count,
                                    int64_t cap, int n, int C, float* __restrict__ obox,
                                    float* __restrict__ oscore, int32_t* __restrict__ oseg,
                                    int32_t* __restrict__ ogrp, int32_t* __restrict__ ocls) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cap) return;
  const bool valid = (unsigned long long)i < *count;
  int32_t flat = valid ? sel[i] : 0;
  int32_t cls = flat % C;
  int32_t row = flat / C;          // image * n + box
  int32_t img = row / n;
#pragma unroll
  for (int k = 0; k < 5; k++) obox[i * 5 + k] = valid ? boxes5[(int64_t)row * 5 + k] : 0.f;
  oscore[i] = valid ? scores[flat] : -1.f;
  oseg[i] = valid ? img * C + cls : -1;
  ogrp[i] = valid ? img : -1;
  ocls[i] = valid ? cls : -1;
}
}  // namespace
}  // namespace s2a

extern "C" size_t s2a_multiclass_candidates_workspace_bytes(int64_t total) {
  const size_t t = (size_t)std::max<int64_t>(total, 1);
  return align_up(t * 4) + align_up(((t + kCandBlock - 1) / kCandBlock + 1) * 4) + 1024;
}

extern "C" int s2a_multiclass_candidates(const float* boxes, const float* scores, int64_t batch,
                                         int64_t n, int64_t num_classes, float score_thr, int64_t cap,
                                         float* out_boxes, float* out_scores, int32_t* out_seg,
                                         int32_t* out_grp, int32_t* out_cls, int64_t* count_dev,
                                         void* workspace, size_t workspace_bytes, s2a_stream_t stream) {
  const int64_t total = batch * n * num_classes;
  S2A_CHECK_ARG(batch >= 0 && n >= 0 && num_classes > 0 && cap > 0, "multiclass_candidates: bad shape");
  S2A_CHECK_ARG(total < (1ll << 31), "multiclass_candidates: too many scores");
  S2A_CHECK_ARG(out_boxes && out_scores && out_seg && out_grp && out_cls && count_dev, "multiclass_candidates: NULL output");
  hipStream_t st = as_stream(stream);
  if (total == 0) {
    fill_u32(count_dev, 0u, 2, st);
    fill_u32(out_seg, 0xffffffffu, (size_t)cap, st);
    fill_u32(out_grp, 0xffffffffu, (size_t)cap, st);
    return S2A_OK;
  }
  S2A_CHECK_ARG(boxes && scores, "multiclass_candidates: NULL input");
  Carver cv(workspace, workspace_bytes);
  const int64_t nb = (total + kCandBlock - 1) / kCandBlock;
  int32_t* sel = cv.take<int32_t>((size_t)total);
  uint32_t* blk = cv.take<uint32_t>((size_t)nb + 1);
  if (!sel || !blk) {
    set_error("multiclass_candidates: workspace too small (%zu bytes)", workspace_bytes);
    return S2A_EWORKSPACE;
  }
  k_cand_count<<<(unsigned)nb, 256, 0, st>>>(scores, score_thr, total, blk);
  k_cand_scan<<<1, 1024, 0, st>>>(blk, nb, reinterpret_cast<unsigned long long*>(count_dev));
  k_cand_scatter<<<(unsigned)nb, 256, 0, st>>>(scores, score_thr, total, blk, sel);
  k_gather_candidates<<<(unsigned)((cap + 255) / 256), 256, 0, st>>>(
      boxes, scores, sel, reinterpret_cast<const unsigned long long*>(count_dev), cap, (int)n,
      (int)num_classes, out_boxes, out_scores, out_seg, out_grp, out_cls);
  S2A_LAUNCH_CHECK();
  return S2A_OK;
}

#ifdef S2A_MEASURE
extern "C" int s2a_debug_small_stamps(unsigned long long* host_dst) {
  S2A_HIP(hipDeviceSynchronize());
  S2A_HIP(hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_small_stamps), 64 * 16 * 8));
  return S2A_OK;
}
#endif
extern "C" int s2a_nms_small_stats(int64_t* taken, int64_t* fell_back) {
  std::lock_guard<std::mutex> lock(g_small_mutex);
  if (taken) *taken = g_small_taken;
  if (fell_back) *fell_back = g_small_fallback;
  return S2A_OK;
}

extern "C" size_t s2a_nms_rotated_workspace_bytes(int64_t n, int64_t max_segment_rows) {
  return nms_workspace_bytes(n, max_segment_rows);
}

extern "C" int s2a_ml_nms_rotated(const float* dets, const float* scores, const float* labels,
                                  int64_t n, float iou_threshold, int64_t* keep, int64_t* count_dev,
                                  int64_t* host_count, void* workspace, size_t workspace_bytes,
                                  s2a_stream_t stream) {
  return nms_dropin(dets, scores, labels, n, iou_threshold, keep, count_dev, host_count, workspace,
                    workspace_bytes, as_stream(stream));
}

extern "C" int s2a_nms_rotated(const float* dets, const float* scores, int64_t n, float iou_threshold,
                               int64_t* keep, int64_t* count_dev, int64_t* host_count,
                               void* workspace, size_t workspace_bytes, s2a_stream_t stream) {
  return nms_dropin(dets, scores, nullptr, n, iou_threshold, keep, count_dev, host_count, workspace,
                    workspace_bytes, as_stream(stream));
}

namespace {
// what the batched detector wants written behind the NMS (s2a_nms_rotated_segmented_dets); all NULL for the plain form
struct NmsEmit {
  const int32_t* row_labels;
  float* wire;
  int32_t* labels_out;
  int32_t* counts_out;
  const int64_t* cand_found;
  int64_t* overflow_out;
  int64_t* dropped_total;
};

__global__ __launch_bounds__(256) void k_nms_emit_empty(int32_t num_groups, int32_t max_per_group, float* __restrict__ wire,
                                                        int32_t* __restrict__ labels_out, int32_t* __restrict__ counts_out,
                                                        const long long* __restrict__ cand_found,
                                                        long long* __restrict__ overflow_out,
                                                        long long* __restrict__ dropped_total) {
  // a row cap of ZERO drops every candidate that was found: the same accounting as k_nms_group_emit with n = 0
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const long long found = cand_found ? *cand_found : 0;
    if (overflow_out) { overflow_out[0] = found; overflow_out[1] = found; }
    if (dropped_total && cand_found) *dropped_total += found;
  }
  const int64_t row = (int64_t)max_per_group * 7 + 1, total = (int64_t)num_groups * row;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t c = i % row;
    wire[i] = (c < (int64_t)max_per_group * 7 && c % 7 == 6) ? -1.f : 0.f;
  }
  if (labels_out)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)num_groups * max_per_group; i += (int64_t)gridDim.x * 256)
      labels_out[i] = -1;
  if (counts_out && blockIdx.x == 0)
    for (int i = threadIdx.x; i < num_groups; i += 256) counts_out[i] = 0;
}

int nms_segmented_impl(const float* dets, const float* scores, const int32_t* segment_ids, const int32_t* group_ids,
                       int64_t n, int32_t num_segments, int32_t num_groups, float iou_threshold, uint8_t* keep_flags,
                       int32_t* keep, int32_t* group_counts, int32_t max_per_group, const NmsEmit& em, void* workspace,
                       size_t workspace_bytes, hipStream_t st) {
  S2A_CHECK_ARG(n >= 0 && n < (1ll << 31), "nms_rotated_segmented: n out of range");
  S2A_CHECK_ARG(num_segments > 0 && num_groups > 0, "nms_rotated_segmented: bad segment/group count");
  S2A_CHECK_ARG(keep == nullptr || (group_counts != nullptr && max_per_group > 0),
                "nms_rotated_segmented: keep needs group_counts and max_per_group");
  S2A_CHECK_ARG(em.wire == nullptr || max_per_group > 0, "nms_rotated_segmented_dets: the detection rows need max_per_group");
  if (n == 0) {      // (in front of the NULL checks: the tensors of an empty candidate set have no storage)
    if (keep) {
      fill_u32(keep, 0xffffffffu, (size_t)num_groups * max_per_group, st);
      fill_u32(group_counts, 0u, (size_t)num_groups, st);
    }
    if (em.wire) {
      k_nms_emit_empty<<<64, 256, 0, st>>>(num_groups, max_per_group, em.wire, em.labels_out, em.counts_out,
                                           reinterpret_cast<const long long*>(em.cand_found),
                                           reinterpret_cast<long long*>(em.overflow_out),
                                           reinterpret_cast<long long*>(em.dropped_total));
      S2A_LAUNCH_CHECK();
    }
    return S2A_OK;
  }
  S2A_CHECK_ARG(dets && scores && segment_ids, "nms_rotated_segmented: NULL tensor");
  S2A_CHECK_ARG(em.wire == nullptr || em.row_labels != nullptr, "nms_rotated_segmented_dets: the detection rows need row_labels");
  NmsPlan pl;
  S2A_CHECK_ARG(nms_plan(n, n, &pl) == 0, "nms_rotated_segmented: rocprim size query failed");
  // the caller may have sized the workspace with a tighter per-segment bound (s2a_nms_rotated_workspace_bytes(n,
  // max_segment_rows)): accept any workspace that holds the fixed part and split the rest over the three lists --
  // tiles, edges, pairs, each capped at what n rows in ONE segment could need.  A list that turns out too small is
  // not an error: the direct greedy kernel then redoes the segments (slow, exact).
  Carver cv(workspace, workspace_bytes);
  NmsBuffers B;
  nms_carve_fixed(cv, n, pl, &B);
  if (cv.off + (48u << 10) > workspace_bytes || !B.rp_temp[2]) {
    set_error("nms_rotated_segmented: workspace too small (%zu bytes)", workspace_bytes);
    return S2A_EWORKSPACE;
  }
  size_t rest = (workspace_bytes - cv.off) / 256 * 256;
  char* base = static_cast<char*>(workspace) + cv.off;
  const size_t tbytes = std::min((size_t)pl.tile_cap * sizeof(TileRef), rest / 6) / 256 * 256;
  const size_t ebytes = std::min((size_t)pl.edge_cap * sizeof(uint2), (rest - tbytes) / 3) / 256 * 256;
  const size_t qbytes = std::min((size_t)pl.queue_cap * sizeof(uint2), rest - tbytes - ebytes) / 256 * 256;
  pl.tile_cap = tbytes / sizeof(TileRef);
  pl.edge_cap = ebytes / sizeof(uint2);
  pl.queue_cap = qbytes / sizeof(uint2);
  S2A_CHECK_ARG(pl.tile_cap >= 64 && pl.edge_cap >= 1024 && pl.queue_cap >= 1024,
                "nms_rotated_segmented: workspace too small for the pair / edge / tile lists");
  B.tiles = reinterpret_cast<TileRef*>(base);
  B.edges = reinterpret_cast<uint2*>(base + tbytes);
  B.gq = reinterpret_cast<uint2*>(base + tbytes + ebytes);
  // the detection rows without the (group, score) sort of the whole buffer: k_nms_group_emit_scan.  S2A_NMS_SEGSORT=0 keeps
  // the library sorts (A/B, tests)
  bool scan_emit = em.wire != nullptr && keep == nullptr && group_ids != nullptr && max_per_group <= kEmitCap / 2;
  if (const char* e = std::getenv("S2A_NMS_SEGSORT")) scan_emit = scan_emit && e[0] != '0';
  int rc = nms_core(dets, scores, nullptr, segment_ids, group_ids, n, (uint32_t)num_segments,
                    (uint32_t)num_groups, iou_threshold, pl, B, st, false, !scan_emit,
                    reinterpret_cast<const long long*>(em.cand_found));
  if (rc != S2A_OK) return rc;
#ifdef S2A_MEASURE
  { int rc_ = nms_debug_dump(B, n, st); if (rc_ != S2A_OK) return rc_; }
#endif
  if (keep_flags)
    S2A_HIP(hipMemcpyAsync(keep_flags, B.keep_orig, (size_t)n, hipMemcpyDeviceToDevice, st));
  if (keep) {
    k_nms_group_compact<<<(unsigned)num_groups, 1024, 0, st>>>(B.keyC_s, B.perm_glob, B.keep_orig, n,
                                                               max_per_group, keep, group_counts);
  }
  if (em.wire && scan_emit) {
    k_nms_group_emit_scan<<<(unsigned)num_groups, 1024, 0, st>>>(group_ids, B.keep_orig, dets, scores, em.row_labels, n,
                                                                 max_per_group, em.wire, em.labels_out, em.counts_out,
                                                                 reinterpret_cast<const long long*>(em.cand_found),
                                                                 reinterpret_cast<long long*>(em.overflow_out),
                                                                 reinterpret_cast<long long*>(em.dropped_total));
  } else if (em.wire) {
    k_nms_group_emit<<<(unsigned)num_groups, 1024, 0, st>>>(B.keyC_s, B.perm_glob, B.keep_orig, dets, scores,
                                                            em.row_labels, n, max_per_group, em.wire, em.labels_out,
                                                            em.counts_out, reinterpret_cast<const long long*>(em.cand_found),
                                                            reinterpret_cast<long long*>(em.overflow_out),
                                                            reinterpret_cast<long long*>(em.dropped_total));
  }
  S2A_LAUNCH_CHECK();
  return S2A_OK;
}
}  // namespace

extern "C" int s2a_nms_rotated_segmented(const float* dets, const float* scores,
                                         const int32_t* segment_ids, const int32_t* group_ids,
                                         int64_t n, int32_t num_segments, int32_t num_groups,
                                         float iou_threshold, uint8_t* keep_flags, int32_t* keep,
                                         int32_t* group_counts, int32_t max_per_group,
                                         void* workspace, size_t workspace_bytes,
                                         s2a_stream_t stream) {
  return nms_segmented_impl(dets, scores, segment_ids, group_ids, n, num_segments, num_groups, iou_threshold, keep_flags,
                            keep, group_counts, max_per_group, NmsEmit{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}, workspace,
                            workspace_bytes, as_stream(stream));
}

extern "C" int s2a_nms_rotated_segmented_dets(const float* dets, const float* scores, const int32_t* segment_ids,
                                              const int32_t* group_ids, const int32_t* row_labels, int64_t n,
                                              int32_t num_segments, int32_t num_groups, float iou_threshold,
                                              int32_t max_per_group, float* wire, int32_t* labels_out,
                                              int32_t* counts_out, const int64_t* cand_found, int64_t* overflow_out,
                                              int64_t* dropped_total, void* workspace, size_t workspace_bytes,
                                              s2a_stream_t stream) {
  S2A_CHECK_ARG(wire != nullptr, "nms_rotated_segmented_dets: NULL output");
  return nms_segmented_impl(dets, scores, segment_ids, group_ids, n, num_segments, num_groups, iou_threshold, nullptr,
                            nullptr, nullptr, max_per_group, NmsEmit{row_labels, wire, labels_out, counts_out, cand_found, overflow_out, dropped_total}, workspace,
                            workspace_bytes, as_stream(stream));
}
