// 3x3 / stride 1 / pad 1 convolution, f16 in / f32 accumulate, with the Winograd F(2,3) minimal-filtering form along x.
//
// Serves the regular 256 -> 256 convolutions of S2ANetHead (models/head.py:163-222: fam_reg_ls, fam_cls_ls, odm_reg_ls,
// odm_cls_ls, or_conv; forward_single :296-348), where the reference calls cuDNN (which picks its own algorithm; the
// arithmetic type stays f16 in / f32 accumulate).  The direct kernel (dcn_ops.hip: k_conv_f16) spends 9 C O multiply-adds
// per output; this one 6 C O:
//     d0..d3 = four neighbouring pixels of an input row (columns 2t-1 .. 2t+2)
//     V0 = d0 - d2, V1 = d1 + d2, V2 = d2 - d1, V3 = d1 - d3                               (input transform, f16)
//     U0 = g0, U1 = (g0 + g1 + g2) / 2, U2 = (g0 - g1 + g2) / 2, U3 = g2                   (filter transform, offline, f32 -> f16 once)
//     M_xi[o, y, t] = sum_ky sum_c U_xi[ky, o, c] V_xi[c, y + ky - 1, t]                   (MFMA, f32 accumulate)
//     out[y, 2t] = M0 + M1 + M2, out[y, 2t + 1] = M1 - M2 - M3                             (output transform, f32, in registers)
// Only ONE dimension is transformed: the full F(2x2, 3x3) form needs 16 accumulators per 4 outputs (4x the direct form's),
// which at 256 KB of accumulators per CU leaves 64 tiles x 64 out channels per workgroup -- the transformed filter (2 MB)
// would be streamed L2 -> CU at 64 B/clk and the transformed input written to LDS at 64 B/clk, both above what a CU moves
// (DESIGN 4, round 6).  The 1-D form doubles the accumulators only: 16 x 32 outputs x 64 out channels per workgroup.
//
// Workgroup = 512 threads, one per CU: a 16-row x 32-column output tile of one image x 64 out channels.  Per 32-channel
// chunk the raw 18 x 34-pixel patch comes in by LDS-DMA (80-byte pixels: 64 B of channels + 16 B pad, conflict-free
// ds_read_b128 at a two-pixel lane stride) and the transformed filter of the chunk in two halves (xi pairs {0,2} and {1,3}:
// each pair needs only three of the four pixels), fragment order, also by LDS-DMA.  A wave owns 4 output rows x 16 x-tiles x
// 32 out channels x all four xi (128 accumulator registers); it builds its B fragments from the raw patch in registers
// (one v_pk_add_f16 per MFMA) -- the transformed input never exists in memory.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "common.hpp"

// S2A_WABL: compile-time timing ablations (never set in a shipped build; outputs are wrong): 1 = the patch is DMA-ed for
// chunk 0 only, 2 = the filter for stage 0 only, 4 = no MFMAs (fragment reads and transforms stay), 8 = no barriers in the loop
#ifndef S2A_WABL
#define S2A_WABL 0
#endif

namespace s2a {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using h4 = __attribute__((ext_vector_type(4))) _Float16;
using h2 = __attribute__((ext_vector_type(2))) _Float16;

constexpr int kWRows = 16, kWCols = 32;                 // output tile
constexpr int kWPR = kWRows + 2, kWPC = kWCols + 2;     // raw patch: one-pixel halo
constexpr int kWPix = kWPR * kWPC;                      // 612
constexpr int kWPitch = 80;                             // bytes per patch pixel (32 channels + 16 B pad)
constexpr int kWRawPieces = (kWPix * 5 + 63) / 64;      // 1 KB LDS-DMA pieces per chunk: 48
constexpr int kWRawBytes = kWRawPieces * 1024;          // 49152
constexpr int kWUStage = 24 * 1024;                     // 2 xi x 3 ky x 4 out-channel tiles x 1 KB
constexpr int kWOutRow = 144;                           // staged output row: 64 channels + 16 B pad
constexpr int kWLds = 2 * kWRawBytes + 2 * kWUStage;    // 147456 (>= 512 * 144 for the epilogue)
constexpr int kWThreads = 512;
constexpr int kWMaxLevels = 8;

struct WLevels {
  int n, batch;
  int H[kWMaxLevels], W[kWMaxLevels], tile0[kWMaxLevels], pix0[kWMaxLevels];
};

// blockIdx round-robins over the 8 XCDs: give every XCD a contiguous run (neighbouring tiles and the out-channel blocks
// of one tile share an L2).  Bijective for any count.
__device__ __forceinline__ unsigned wg_remap(unsigned bid, unsigned n) {
  const unsigned q = n / 8, r = n % 8, x = bid % 8;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + bid / 8;
}

// filter [O][C][3][3] f16 -> transformed, fragment order:
//   [O/64][C/32][pair 2][x 2][ky 3][a 4][lane 64][8]   xi = pair + 2x, out channel = 64 ob + 16 a + (lane & 15),
//   channel = 32 cc + 8 (lane >> 4) + j -- one (chunk, pair) stage of a 64-channel block is 24 KB contiguous
__global__ void k_wino_pack_weight(const _Float16* __restrict__ w, int O, int C, _Float16* __restrict__ up) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)O * C * 12;
  if (e >= total) return;
  const int j = (int)(e & 7), lane = (int)((e >> 3) & 63);
  int64_t r = e >> 9;
  const int a = (int)(r & 3); r >>= 2;
  const int ky = (int)(r % 3); r /= 3;
  const int x = (int)(r & 1); r >>= 1;
  const int pair = (int)(r & 1); r >>= 1;
  const int CC = C / 32;
  const int cc = (int)(r % CC), ob = (int)(r / CC);
  const int o = ob * 64 + a * 16 + (lane & 15), c = cc * 32 + (lane >> 4) * 8 + j;
  const _Float16* g = w + ((int64_t)o * C + c) * 9 + ky * 3;
  const float g0 = (float)g[0], g1 = (float)g[1], g2 = (float)g[2];
  const int xi = pair + 2 * x;
  const float u = xi == 0 ? g0 : (xi == 1 ? 0.5f * (g0 + g1 + g2) : (xi == 2 ? 0.5f * (g0 - g1 + g2) : g2));
  up[e] = (_Float16)u;
}

// a - b on eight halves as four v_pk_add_f16 with the negate modifier (the compiler scalarises a vector fsub into
// v_sub_f16 + v_sub_f16_sdwa + v_pack_b32_f16: three instructions per two values)
__device__ __forceinline__ f16x8 pk_sub(const f16x8& a, const f16x8& b) {
  using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
  const u32x4 ua = __builtin_bit_cast(u32x4, a), ub = __builtin_bit_cast(u32x4, b);
  u32x4 r;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    unsigned v;
    asm("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(v) : "v"(ua[i]), "v"(ub[i]));
    r[i] = v;
  }
  return __builtin_bit_cast(f16x8, r);
}

struct WinoArgs {
  const _Float16* x;        // [P, C] pyramid-packed NHWC
  const _Float16* u;        // k_wino_pack_weight
  const _Float16* bias;     // [O] or null
  _Float16* out;            // [P, O]
  _Float16* pool_out;       // [P, O/8] or null: max over runs of 8 channels of the finished tile
  int C, O, relu;
  WLevels lt;
};

__global__ __launch_bounds__(kWThreads, 1) void k_conv_wino_f16(WinoArgs a) {
  using V = f16x8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int og = wave & 1, rg = wave >> 1;          // 32-out-channel half, group of four output rows
  const int nob = a.O / 64;
  const unsigned wg = wg_remap(blockIdx.x, gridDim.x);
  int tile = (int)(wg / nob);
  const int ob = (int)(wg % nob);
  int H = a.lt.H[0], W = a.lt.W[0], p0 = 0, t0 = 0;
#pragma unroll
  for (int i = 1; i < kWMaxLevels; i++)
    if (i < a.lt.n && tile >= a.lt.tile0[i]) { t0 = a.lt.tile0[i]; p0 = a.lt.pix0[i]; H = a.lt.H[i]; W = a.lt.W[i]; }
  tile -= t0;
  const int C = a.C, O = a.O;
  const int64_t HW = (int64_t)H * W;
  const int txn = (W + kWCols - 1) / kWCols, tyn = (H + kWRows - 1) / kWRows;
  const int bimg = tile / (txn * tyn), trem = tile % (txn * tyn);
  const int ty0 = (trem / txn) * kWRows, tx0 = (trem % txn) * kWCols;
  const _Float16* x = a.x + (int64_t)p0 * C;
  _Float16* out = a.out + (int64_t)p0 * O;
  const unsigned x_bytes = (unsigned)((int64_t)a.lt.batch * HW * C * 2);
  const unsigned row_bytes = (unsigned)C * 2;
  const int CC = C / 32;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(x), 0, (int)x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<_Float16*>(a.u), 0, (int)((uint64_t)O * (uint64_t)C * 12 * 2), 0x00020000);

  // raw patch: slot v = pixel * 5 + q (q < 4: the q-th 8-channel group of the chunk, q = 4: pad); a wave instruction fills
  // 64 consecutive slots; pad slots and pixels outside the image read beyond the buffer (-> zeros)
  constexpr int kJ = kWRawPieces / 8;   // 6 pieces per wave
  unsigned pvoff[kJ];
#pragma unroll
  for (int j = 0; j < kJ; j++) {
    const int v = (wave + 8 * j) * 64 + lane, p = v / 5, q = v % 5;
    const int yy = ty0 - 1 + p / kWPC, xx = tx0 - 1 + p % kWPC;
    const bool in = q < 4 && p < kWPix && yy >= 0 && yy < H && xx >= 0 && xx < W;
    pvoff[j] = in ? (unsigned)(((int64_t)bimg * HW + (int64_t)yy * W + xx) * row_bytes + q * 16) : 0x80000000u;
  }
  char* raw0 = smem;
  char* ub0 = smem + 2 * kWRawBytes;
  auto raw_issue = [&](int cc) {
    char* P = raw0 + (cc & 1) * kWRawBytes;
#pragma unroll
    for (int j = 0; j < kJ; j++)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(P + (wave + 8 * j) * 1024), 16,
                                               (int)pvoff[j], cc * 64, 0, 0);
  };
  const int ubase = (ob * CC) * 2 * kWUStage;       // this out-channel block's stages: (cc * 2 + pair) * 24 KB
  auto u_issue = [&](int s) {                       // 24 pieces of 1 KB over the eight waves
    char* Ub = ub0 + (s & 1) * kWUStage;
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const int piece = wave * 3 + j;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ru, (__attribute__((address_space(3))) void*)(Ub + piece * 1024), 16,
                                               piece * 1024 + lane * 16, ubase + s * kWUStage, 0, 0);
    }
  };

  const int t = lane & 15, kg = lane >> 4;          // x-tile (B column / D column), 8-channel group (k)
  const int rbase = ((4 * rg) * kWPC + 2 * t) * kWPitch + kg * 16;
  const bool wave_active = ty0 + 4 * rg < H;        // rows beyond the image: no contraction (DMA and barriers still)
  f32x4 acc[4][2][4];
#pragma unroll
  for (int xi = 0; xi < 4; xi++)
#pragma unroll
    for (int aa = 0; aa < 2; aa++)
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int e = 0; e < 4; e++) acc[xi][aa][r][e] = 0.f;

  // ---- one (chunk, pair) stage = 2 xi-steps x 6 iterations of (filter fragment, 4 MFMAs over the wave's rows).  Software
  // pipeline: the B fragments of the NEXT xi-step are built while this one multiplies (two raw reads per row an iteration,
  // combined one iteration later), the filter fragments are read two iterations ahead -- with everything issued at the
  // head of a stage the LDS round trips were the stage (timing ablation, DESIGN 4 round 6: 114 of 142 us without a single MFMA).
  //   xi order inside a chunk: 0, 2 (pair 0: pixels d0 d2 / d2 d1), 1, 3 (pair 1: d1 d2 / d1 d3); bA / bB alternate.
  V bA[6], bB[6];
  auto raw2 = [&](auto xi_c, const char* pr, V& u, V& v) {     // the two pixels xi combines, in (minuend, subtrahend) order
    constexpr int xi = decltype(xi_c)::value;
    constexpr int o0 = xi == 0 ? 0 : (xi == 2 ? 2 : 1), o1 = xi == 0 ? 2 : (xi == 2 ? 1 : (xi == 1 ? 2 : 3));
    u = *reinterpret_cast<const V*>(pr + o0 * kWPitch);
    v = *reinterpret_cast<const V*>(pr + o1 * kWPitch);
  };
  auto combine = [&](auto xi_c, const V& u, const V& v) -> V {
    if constexpr (decltype(xi_c)::value == 1) return u + v;
    else return pk_sub(u, v);
  };
  auto stage = [&](auto pair_c, const char* P, const char* Pn, const char* Ub) {
    constexpr int pair = decltype(pair_c)::value;
    if (!wave_active) return;
    const char* ua = Ub + (2 * og) * 1024 + lane * 16;
    V af[3];
    af[0] = *reinterpret_cast<const V*>(ua);
    af[1] = *reinterpret_cast<const V*>(ua + 1024);
    V ru[2], rv[2];
#pragma unroll
    for (int i = 0; i < 12; i++) {
      const int xx = i / 6, k = i % 6, ky = k >> 1, aa = k & 1;
      if (i + 2 < 12) {
        const int i2 = i + 2, x2 = i2 / 6, k2 = i2 % 6;
        af[i2 % 3] = *reinterpret_cast<const V*>(ua + ((x2 * 3 + (k2 >> 1)) * 4 + (k2 & 1)) * 1024);
      }
      // next xi-step's B fragments: x = 0 builds bB for xi = pair + 2 from this chunk; x = 1 builds bA for the next stage's
      // first xi (pair 0 -> xi 1 of this chunk, pair 1 -> xi 0 of the next chunk: Pn)
      {
        const char* src = (xx == 0 || pair == 0) ? P : Pn;
        const char* pr = src + rbase + k * (kWPC * kWPitch);
        if (xx == 0) raw2(std::integral_constant<int, pair + 2>{}, pr, ru[k & 1], rv[k & 1]);
        else raw2(std::integral_constant<int, 1 - pair>{}, pr, ru[k & 1], rv[k & 1]);
        if (k >= 1) {
          if (xx == 0) bB[k - 1] = combine(std::integral_constant<int, pair + 2>{}, ru[(k - 1) & 1], rv[(k - 1) & 1]);
          else bA[k - 1] = combine(std::integral_constant<int, 1 - pair>{}, ru[(k - 1) & 1], rv[(k - 1) & 1]);
        }
      }
#if S2A_WABL & 4
      asm volatile("" : : "v"(af[i % 3]));
#pragma unroll
      for (int r = 0; r < 4; r++) asm volatile("" : : "v"(xx == 0 ? bA[r + ky] : bB[r + ky]));
#else
#pragma unroll
      for (int r = 0; r < 4; r++)
        acc[pair + 2 * xx][aa][r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i % 3], xx == 0 ? bA[r + ky] : bB[r + ky],
                                                                          acc[pair + 2 * xx][aa][r], 0, 0, 0);
#endif
      if (k == 5) {
        if (xx == 0) bB[5] = combine(std::integral_constant<int, pair + 2>{}, ru[1], rv[1]);
        else bA[5] = combine(std::integral_constant<int, 1 - pair>{}, ru[1], rv[1]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // bias of this wave's 2 x 16 out channels: lane holds channels 4 kg .. 4 kg + 3 of each tile (the D rows)
  h4 bq[2];
#pragma unroll
  for (int aa = 0; aa < 2; aa++) {
    bq[aa] = h4{(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
    if (a.bias) bq[aa] = *reinterpret_cast<const h4*>(a.bias + ob * 64 + og * 32 + aa * 16 + 4 * kg);
  }

  raw_issue(0);
  u_issue(0);
  __syncthreads();                                  // (vmcnt(0) in front of the barrier: the DMAs have landed)
  if (wave_active) {                                // B fragments of the first xi-step (exposed once per tile)
#pragma unroll
    for (int r = 0; r < 6; r++) {
      V u, v;
      raw2(std::integral_constant<int, 0>{}, raw0 + rbase + r * (kWPC * kWPitch), u, v);
      bA[r] = combine(std::integral_constant<int, 0>{}, u, v);
    }
  }
  for (int cc = 0; cc < CC; cc++) {
    const char* P = raw0 + (cc & 1) * kWRawBytes;
    const char* Pn = raw0 + ((cc + 1) & 1) * kWRawBytes;      // (last chunk: stale data, the fragments built from it are unused)
#if S2A_WABL
    // timing ablations only: 1 = no patch DMA, 2 = no filter DMA, 8 = no barriers
    if (!(S2A_WABL & 2)) u_issue(2 * cc + 1);
    if (!(S2A_WABL & 1) && cc + 1 < CC) raw_issue(cc + 1);
    stage(std::integral_constant<int, 0>{}, P, Pn, ub0);
    if (!(S2A_WABL & 8)) __syncthreads();
    if (!(S2A_WABL & 2) && cc + 1 < CC) u_issue(2 * cc + 2);
    stage(std::integral_constant<int, 1>{}, P, Pn, (S2A_WABL & 2) ? ub0 : ub0 + kWUStage);
    if (!(S2A_WABL & 8)) __syncthreads();
#else
    u_issue(2 * cc + 1);
    if (cc + 1 < CC) raw_issue(cc + 1);
    stage(std::integral_constant<int, 0>{}, P, Pn, ub0);
    __syncthreads();                                // chunk cc + 1's patch has landed (read by the next stage's prefetch)
    if (cc + 1 < CC) u_issue(2 * cc + 2);
    stage(std::integral_constant<int, 1>{}, P, Pn, ub0 + kWUStage);
    __syncthreads();
#endif
  }

  // ---- epilogue: output transform in registers, bias, ReLU, tile staged through LDS (row slot = 32 row + 16 (col & 1) +
  // (col >> 1): even columns first -- a lane's two outputs land 16 slots apart, 2-way instead of 4-way store conflicts)
  char* s_out = smem;
  const bool relu = a.relu != 0;
  if (wave_active) {
#pragma unroll
    for (int aa = 0; aa < 2; aa++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const f32x4 m0 = acc[0][aa][r], m1 = acc[1][aa][r], m2 = acc[2][aa][r], m3 = acc[3][aa][r];
        h4 y0, y1;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          y0[e] = (_Float16)((m0[e] + m1[e] + m2[e]) + (float)bq[aa][e]);
          y1[e] = (_Float16)((m1[e] - m2[e] - m3[e]) + (float)bq[aa][e]);
        }
        if (relu) {
          const h4 z = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
          y0 = __builtin_elementwise_max(y0, z);
          y1 = __builtin_elementwise_max(y1, z);
        }
        const int slot = (4 * rg + r) * 32 + t;
        const int chb = (og * 32 + aa * 16 + 4 * kg) * 2;
        *reinterpret_cast<h4*>(s_out + slot * kWOutRow + chb) = y0;
        *reinterpret_cast<h4*>(s_out + (slot + 16) * kWOutRow + chb) = y1;
      }
  }
  __syncthreads();
  const int64_t img0 = (int64_t)bimg * HW;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int idx = tid + kWThreads * i, slot = idx >> 3, col8 = idx & 7;
    const int y = ty0 + (slot >> 5), xq = tx0 + 2 * (slot & 15) + ((slot >> 4) & 1);
    if (y < H && xq < W)
      *reinterpret_cast<V*>(out + (img0 + (int64_t)y * W + xq) * O + ob * 64 + col8 * 8) =
          *reinterpret_cast<const V*>(s_out + slot * kWOutRow + col8 * 16);
  }
  if (a.pool_out) {
    // rotation-invariant pooling of the tile just produced (models/orn/functions/rotation_invariant_pooling.py:19-27): one
    // thread = one position x this block's 8 pooled channels
    const int slot = tid;
    const int y = ty0 + (slot >> 5), xq = tx0 + 2 * (slot & 15) + ((slot >> 4) & 1);
    if (y < H && xq < W) {
      V res;
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const V v = *reinterpret_cast<const V*>(s_out + slot * kWOutRow + e * 16);
        _Float16 mx = v[0];
#pragma unroll
        for (int k = 1; k < 8; k++) mx = v[k] > mx ? v[k] : mx;
        res[e] = mx;
      }
      *reinterpret_cast<V*>(a.pool_out + ((int64_t)p0 + img0 + (int64_t)y * W + xq) * (O / 8) + ob * 8) = res;
    }
  }
}

// 16 x 32 tiles per level, pixel offsets; returns the total tile count or -1
int64_t wino_levels(const s2a_pyramid* pyr, int64_t batch, WLevels* lt, int64_t* total_pix) {
  if (!pyr || pyr->n_levels < 1 || pyr->n_levels > kWMaxLevels) return -1;
  *lt = WLevels{};
  lt->n = pyr->n_levels;
  lt->batch = (int)batch;
  int64_t tiles = 0, pix = 0;
  for (int i = 0; i < pyr->n_levels; i++) {
    const int64_t H = pyr->height[i], W = pyr->width[i];
    if (H < 1 || W < 1 || H >= 32000 || W >= 32000) return -1;
    lt->H[i] = (int)H; lt->W[i] = (int)W;
    lt->tile0[i] = (int)tiles; lt->pix0[i] = (int)pix;
    tiles += batch * ((W + kWCols - 1) / kWCols) * ((H + kWRows - 1) / kWRows);
    pix += batch * H * W;
    if (tiles >= (1ll << 29) || pix >= (1ll << 31)) return -1;
  }
  *total_pix = pix;
  return tiles;
}

}  // namespace
}  // namespace s2a

namespace s2a {
int build_flags_wino() { return S2A_WABL ? 1 : 0; }      // a timing-ablation object: reported by s2a_build_flags, refused by the loader
}  // namespace s2a

extern "C" int64_t s2a_conv_wino_packed_elems(int64_t out_channels, int64_t channels) {
  if (out_channels <= 0 || channels <= 0 || out_channels % 64 != 0 || channels % 32 != 0) return -1;
  return out_channels * channels * 12;
}

extern "C" int s2a_conv_wino_pack_weight_f16(const void* weight, int64_t out_channels, int64_t channels, void* packed,
                                             s2a_stream_t stream) {
  S2A_CHECK_ARG(out_channels > 0 && channels > 0 && out_channels % 64 == 0 && channels % 32 == 0,
                "conv_wino_pack_weight: out_channels must be a multiple of 64, channels a multiple of 32");
  S2A_CHECK_ARG(out_channels * channels * 12 * 2 < (1ll << 31), "conv_wino_pack_weight: filter too large for 32-bit offsets");
  S2A_CHECK_ARG(weight && packed, "conv_wino_pack_weight: NULL tensor");
  const int64_t total = out_channels * channels * 12;
  s2a::k_wino_pack_weight<<<(unsigned)((total + 255) / 256), 256, 0, s2a::as_stream(stream)>>>(
      (const _Float16*)weight, (int)out_channels, (int)channels, (_Float16*)packed);
  S2A_LAUNCH_CHECK();
  return S2A_OK;
}

extern "C" int s2a_conv3x3_wino_pyramid_f16(const void* x, const void* weight_wino, const void* bias, void* out, void* pooled,
                                            int64_t batch, int64_t channels, int64_t out_channels, int relu,
                                            const s2a_pyramid* pyr, s2a_stream_t stream) {
  using namespace s2a;
  S2A_CHECK_ARG(batch >= 0 && channels > 0 && out_channels > 0, "conv_wino_pyramid: bad shape");
  S2A_CHECK_ARG(channels % 32 == 0 && out_channels % 64 == 0,
                "conv_wino_pyramid: channels must be a multiple of 32, out_channels a multiple of 64");
  S2A_CHECK_ARG(out_channels * channels * 12 * 2 < (1ll << 31), "conv_wino_pyramid: filter too large for 32-bit offsets");
  WLevels lt;
  int64_t pix = 0;
  const int64_t tiles = wino_levels(pyr, batch, &lt, &pix);
  S2A_CHECK_ARG(tiles >= 0, "conv_wino_pyramid: bad level table (1..8 levels, positive sizes)");
  for (int i = 0; i < lt.n; i++)
    S2A_CHECK_ARG((uint64_t)batch * lt.H[i] * lt.W[i] * channels * 2 < (1ull << 31),
                  "conv_wino_pyramid: a level is too large for 32-bit offsets");
  S2A_CHECK_ARG(tiles * (out_channels / 64) < (1ll << 31), "conv_wino_pyramid: too many workgroups");
  if (batch == 0) return S2A_OK;
  S2A_CHECK_ARG(x && weight_wino && out, "conv_wino_pyramid: NULL tensor");
  S2A_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)out % 16) == 0 && ((uintptr_t)weight_wino % 16) == 0 &&
                ((uintptr_t)bias % 8) == 0 && ((uintptr_t)pooled % 16) == 0, "conv_wino_pyramid: tensors must be 16-byte aligned");
  WinoArgs a{(const _Float16*)x, (const _Float16*)weight_wino, (const _Float16*)bias, (_Float16*)out, (_Float16*)pooled,
             (int)channels, (int)out_channels, relu, lt};
  S2A_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv_wino_f16), hipFuncAttributeMaxDynamicSharedMemorySize, kWLds));
  k_conv_wino_f16<<<(unsigned)(tiles * (out_channels / 64)), kWThreads, kWLds, as_stream(stream)>>>(a);
  S2A_LAUNCH_CHECK();
  return S2A_OK;
}
