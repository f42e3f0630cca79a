// Shared host-side helpers for libs2anet_hip.so (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/s2anet_hip.h"

namespace s2a {

void set_error(const char* fmt, ...);

inline hipStream_t as_stream(s2a_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

// carve a sub-buffer out of a workspace (256-B aligned); returns nullptr when exhausted
struct Carver {
  char* base;
  size_t size, off;
  Carver(void* p, size_t n) : base(static_cast<char*>(p)), size(n), off(0) {}
  template <typename T>
  T* take(size_t count) {
    size_t bytes = align_up(count * sizeof(T));
    if (base == nullptr || off + bytes > size) {
      off += bytes;  // keep counting so callers can report the need
      return nullptr;
    }
    T* r = reinterpret_cast<T*>(base + off);
    off += bytes;
    return r;
  }
};

// greedy NMS scan over a suppression bitmask (rotated_ops.hip), shared with the polygon NMS
int launch_nms_scan(const unsigned long long* mask, const uint32_t* seg_start, const uint32_t* num_seg,
                    const unsigned long long* mask_off, const uint32_t* nblk, const int32_t* perm_seg,
                    uint8_t* keep_orig, uint32_t max_blocks, const unsigned long long* words_total,
                    unsigned long long words_bound, uint32_t* status, hipStream_t st);

// greedy NMS by rounds over a list of suppression edges (rotated_ops.hip), shared with the polygon NMS: edges[e] = (i, j),
// positions in descending-score order (i < j), one segment; *edge_count_dev of them (<= edge_cap); alive_list: scratch of
// alive_cap >= edge_cap entries; order[p] = original row of position p; keep_orig[row] = 1 kept / 0 removed
size_t nms_edge_rounds_workspace(int64_t n);
int launch_nms_edge_rounds(uint2* edges, unsigned long long edge_cap, const unsigned long long* edge_count_dev, uint2* alive_list,
                           unsigned long long alive_cap, int64_t n, const int32_t* order, uint8_t* keep_orig, void* workspace,
                           size_t workspace_bytes, hipStream_t st);
// kept rows of `order` compacted into keep[] (descending score), *count_dev = how many; cnt_scratch: keep_compact_scratch_words()
size_t keep_compact_scratch_words();
int launch_keep_compact(const uint8_t* keep_orig, const int32_t* order, int64_t n, uint32_t* cnt_scratch, int64_t* keep,
                        int64_t* count_dev, hipStream_t st);

}  // namespace s2a

#define S2A_CHECK_ARG(cond, ...)          \
  do {                                    \
    if (!(cond)) {                        \
      s2a::set_error(__VA_ARGS__);        \
      return S2A_EINVAL;                  \
    }                                     \
  } while (0)

#define S2A_HIP(expr)                                                              \
  do {                                                                             \
    hipError_t e_ = (expr);                                                        \
    if (e_ != hipSuccess) {                                                        \
      s2a::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                     __LINE__);                                                    \
      return S2A_EHIP;                                                             \
    }                                                                              \
  } while (0)

#define S2A_LAUNCH_CHECK() S2A_HIP(hipGetLastError())

#ifdef __HIPCC__
// f32 tensors on the 16-bit matrix instruction (k_dcn_x3, k_dcn_bwd_weight_x3): x == hi + mid + lo EXACTLY for finite x away
// from the underflow range -- three round-to-nearest-even bf16 values, 8 + 8 + 8 significand bits (tests/test_lane_maps_cpu.py
// restates this on the CPU); a product of two such sums without its three smallest cross terms is six bf16 products
namespace s2a {
__device__ __forceinline__ void split3(float x, __bf16& hi, __bf16& mid, __bf16& lo) {
  hi = (__bf16)x;
  const float r1 = x - (float)hi;
  mid = (__bf16)r1;
  lo = (__bf16)(r1 - (float)mid);
}
}  // namespace s2a
#endif

// Entry points that still issue hipMemsetAsync / rocprim::select (the drop-in nms_rotated / ml_nms_rotated / nms_poly
// forms: they return a device count the host usually reads anyway) are NOT graph-safe on ROCm 7.2: they
// refuse a capturing stream instead of producing a graph that misbehaves on its second replay.  The capturable
// post-processing is s2a_nms_rotated_segmented / s2a_multiclass_candidates (own fill / count / scan / scatter kernels).
#define S2A_REFUSE_CAPTURE(st, what)                                                                   \
  do {                                                                                                 \
    hipStreamCaptureStatus cs_ = hipStreamCaptureStatusNone;                                           \
    if (hipStreamIsCapturing((st), &cs_) == hipSuccess && cs_ != hipStreamCaptureStatusNone) {         \
      s2a::set_error(what ": this entry point is not HIP-graph capturable (memset nodes / rocprim::select); "  \
                          "use s2a_nms_rotated_segmented on a capturing stream");                      \
      return S2A_ENOTIMPL;                                                                             \
    }                                                                                                  \
  } while (0)
