// Fused ResNet stem for the end-to-end inference config:
//   uint8 image / 255 (val.py:246-247) -> conv 7x7 / stride 2 / pad 3, 3 -> 64 maps with the folded
//   BatchNorm bias -> ReLU -> max-pool 3x3 / stride 2 / pad 1   (models/backbone.py:112-117, :172-175, :307-308)
// as ONE kernel: the [B,64,H/2,W/2] activation (268 MB at batch 8 of 1024^2) and the normalised f16
// image never exist in HBM; traffic is the uint8 image in and the pooled [B,H/4,W/4,64] tile out.
//
// One workgroup = an 8 x 8 tile of pooled outputs = 17 x 17 conv outputs = a 39 x 39 x 3 input
// patch.  The filter lives in registers (fragment order, 88 VGPRs).  The patch is converted once into LDS as f16 (a 256-entry table holds f16(u8 / 255), the
// exact value of the stock f16 division) and the convolution runs on the matrix cores as an
// implicit GEMM: A = filter [64 x K], B = patch columns [K x 289 pixels], K ordered (ky, kx, c) with
// every filter row ky padded from 21 to 24 entries, so that 8 consecutive K entries are 16
// contiguous bytes of one LDS patch row (4-byte aligned: four ds_read_b32 build a B fragment, no
// packing).  K = 7 * 24 = 168 -> 11 MFMA steps of 16 (the tail multiplies zero filter entries).
// Conv tile -> bias + ReLU -> LDS (over the dead patch) -> 3x3 max -> 16-byte stores.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "common.hpp"

namespace s2a {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using h4 = __attribute__((ext_vector_type(4))) _Float16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

constexpr int kKSteps = 11;                 // 176 = 22 groups of 8 K entries; group q: ky = q / 3, g = q % 3
constexpr int kInRows = 40;                 // 39 patch rows + one zero row for the padded K tail (ky = 7)
constexpr int kInPitch = 136;               // halfs per patch row: 117 used, reads reach index 119
constexpr int kInDw = 31;                   // aligned dwords per patch row (124 bytes from byte -1)
constexpr int kConvPitch = 144;             // bytes per staged conv pixel: 64 halfs + 16 B pad
constexpr int kInBytes = kInRows * kInPitch * 2;                    // 10880
constexpr int kConvBytes = 289 * kConvPitch;                        // 41616
constexpr int kStemLds = 640 + 2 * kKSteps * 64 * 16 + (kInBytes > kConvBytes ? kInBytes : kConvBytes);   // table + bias + filter + patch / conv tile
constexpr int kInIters = (kInRows * kInDw + 255) / 256;             // 5 dwords per thread

// filter [64][3][7][7] f16 -> [m 2][step 11][lane 64][8]: lane l, element j =
// W[m*32 + (l & 31)][k = 16*step + 8*(l >> 5) + j], k = ky*24 + kx*3 + c (zero where kx*3+c > 20 or ky > 6)
__global__ void k_stem_pack(const _Float16* __restrict__ w, _Float16* __restrict__ wp) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= 2 * kKSteps * 64 * 8) return;
  int j = e & 7, lane = (e >> 3) & 63, s = (e >> 9) % kKSteps, m = (e >> 9) / kKSteps;
  int och = m * 32 + (lane & 31);
  int k = 16 * s + 8 * (lane >> 5) + j;
  int ky = k / 24, r = k % 24;
  _Float16 v = (_Float16)0.f;
  if (ky < 7 && r < 21) {
    int kx = r / 3, c = r % 3;
    v = w[((och * 3 + c) * 7 + ky) * 7 + kx];
  }
  wp[e] = v;
}

// Persistent: a workgroup walks tiles blockIdx.x, + gridDim.x, ...  The 22.5 KB of filter fragments go into LDS once
// per workgroup (fetched per tile into every wave's registers they cost 90 KB per tile through the CU's load path and
// 88 VGPRs), and the NEXT tile's patch dwords are requested right after the current patch has been converted, so their
// round trip runs under the MFMA / epilogue / pooling phases of the current tile.
constexpr int kWBytes = 2 * kKSteps * 64 * 16;                          // 22528
__global__ __launch_bounds__(256, 2) void k_stem(const uint8_t* __restrict__ img,      // [B,H,W,3]
                                                 const _Float16* __restrict__ wp,     // k_stem_pack
                                                 const _Float16* __restrict__ bias,   // [64] or null
                                                 _Float16* __restrict__ out,          // [B,Hp,Wp,64]
                                                 int H, int W, int Hc, int Wc, int Hp, int Wp,
                                                 int tiles_x, int tiles_y, float divisor, int tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  _Float16* s_lut = reinterpret_cast<_Float16*>(smem);                 // 256 halfs
  _Float16* s_bias = reinterpret_cast<_Float16*>(smem + 512);          // 64 halfs
  const char* s_w = smem + 640;                                        // filter fragments [m 2][step 11][lane 64] x 16 B
  _Float16* s_in = reinterpret_cast<_Float16*>(smem + 640 + kWBytes);
  char* s_conv = smem + 640 + kWBytes;                                 // aliases the patch (dead by then)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t row_bytes = (int64_t)W * 3;

  // patch dwords of tile t: unconditional loads from a clamped address, so none waits for the previous one
  unsigned pv[kInIters];
  bool pok[kInIters];
  auto patch_issue = [&](int t) {
    const int tx = t % tiles_x;
    const int ty = (t / tiles_x) % tiles_y, b = t / (tiles_x * tiles_y);
    const int iy0 = 2 * (16 * ty - 1) - 3;               // input origin: row 32*ty - 5
    const int64_t a0 = 96 * (int64_t)tx - 16;            // aligned byte offset inside an image row (x origin = 32*tx - 5 -> byte -15)
#pragma unroll
    for (int it = 0; it < kInIters; it++) {
      const int item = tid + 256 * it, r = item / kInDw, d = item % kInDw;
      const int iy = iy0 + r;
      const int64_t bo = a0 + 4 * d;
      pok[it] = r < 39 && iy >= 0 && iy < H && bo >= 0 && bo < row_bytes;
      pv[it] = *reinterpret_cast<const unsigned*>(img + (pok[it] ? ((int64_t)b * H + iy) * row_bytes + bo : 0));
    }
  };
  int t = blockIdx.x;
  patch_issue(t);
  for (int i = tid; i < kWBytes / 16; i += 256)
    *reinterpret_cast<f16x8*>(smem + 640 + i * 16) = *reinterpret_cast<const f16x8*>(wp + i * 8);
  s_lut[tid] = (_Float16)((float)tid / divisor);
  if (tid < 64) s_bias[tid] = bias ? bias[tid] : (_Float16)0.f;
  const int hsel = lane >> 5;

  for (; t < tiles; t += gridDim.x) {
    const int tx = t % tiles_x;
    const int ty = (t / tiles_x) % tiles_y, b = t / (tiles_x * tiles_y);
    const int cy0 = 16 * ty - 1, cx0 = 16 * tx - 1;      // conv-output origin of the tile
    __syncthreads();      // table / bias / filter visible (first tile); the previous tile's pooling has read the conv tile
#pragma unroll
    for (int it = 0; it < kInIters; it++) {
      const int item = tid + 256 * it, r = item / kInDw, d = item % kInDw;
      if (item < kInRows * kInDw) {
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int j = 4 * d + e - 1;    // half index inside the patch row (byte -15 of the row = index 0)
          if (j >= 0) s_in[r * kInPitch + j] = pok[it] ? s_lut[(pv[it] >> (8 * e)) & 255u] : (_Float16)0.f;
        }
      }
    }
    // columns 123..135 of every row are read by nobody (max index 119); nothing to clear
    if (t + (int)gridDim.x < tiles) patch_issue(t + gridDim.x);          // next tile's patch: in flight from here on
    __syncthreads();

    // ---- implicit GEMM on the matrix cores: wave w owns pixel tiles w, w+4, w+8 (10 tiles of 32 = 320 >= 289)
    f32x16 acc[2][3];
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
      for (int q = 0; q < 3; q++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[m][q][r] = 0.f;
    int pbase[3];
#pragma unroll
    for (int q = 0; q < 3; q++) {
      int n = (wave + 4 * q) * 32 + (lane & 31);
      n = n < 289 ? n : 0;
      pbase[q] = (2 * (n / 17)) * kInPitch + 6 * (n % 17);     // half index of tap (0,0), channel 0
    }
#pragma unroll
    for (int s = 0; s < kKSteps; s++) {
      const int kq = 2 * s + hsel, ky = kq / 3, g = kq % 3;
      const int koff = ky * kInPitch + 8 * g;
      f16x8 wa[2];
#pragma unroll
      for (int m = 0; m < 2; m++) wa[m] = *reinterpret_cast<const f16x8*>(s_w + ((m * kKSteps + s) * 64 + lane) * 16);
#pragma unroll
      for (int q = 0; q < 3; q++) {
        if ((wave + 4 * q) * 32 >= 289) continue;            // wave-uniform
        const unsigned* src = reinterpret_cast<const unsigned*>(s_in + pbase[q] + koff);
        u32x4 raw = {src[0], src[1], src[2], src[3]};
        const f16x8 pf = __builtin_bit_cast(f16x8, raw);
#pragma unroll
        for (int m = 0; m < 2; m++)
          acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[m], pf, acc[m][q], 0, 0, 0);
      }
    }
    __syncthreads();                                          // patch dead -> conv staging may overwrite it

    // ---- bias + ReLU -> LDS, zero where the conv pixel lies outside the conv output (the pool's padding:
    // after ReLU every real value is >= 0 and the window always holds its in-range centre, so 0 == -inf here)
#pragma unroll
    for (int q = 0; q < 3; q++) {
      const int n = (wave + 4 * q) * 32 + (lane & 31);
      if (n >= 289) continue;
      const int cy = cy0 + n / 17, cx = cx0 + n % 17;
      const bool in = cy >= 0 && cy < Hc && cx >= 0 && cx < Wc;
#pragma unroll
      for (int m = 0; m < 2; m++)
#pragma unroll
        for (int rq = 0; rq < 4; rq++) {
          const int ch = m * 32 + 8 * rq + 4 * hsel;
          const h4 bq = *reinterpret_cast<const h4*>(s_bias + ch);
          h4 v4;
#pragma unroll
          for (int e = 0; e < 4; e++) {
            float v = fmaxf(acc[m][q][rq * 4 + e] + (float)bq[e], 0.f);
            v4[e] = in ? (_Float16)v : (_Float16)0.f;
          }
          *reinterpret_cast<h4*>(s_conv + n * kConvPitch + ch * 2) = v4;
        }
    }
    __syncthreads();

    // ---- 3x3 / stride 2 max over the staged conv tile, 8 channels (16 B) per item
#pragma unroll
    for (int it = 0; it < 2; it++) {
      const int item = tid + 256 * it, pp = item >> 3, cg = item & 7;
      const int py = pp >> 3, px = pp & 7;
      const int oy = 8 * ty + py, ox = 8 * tx + px;
      f16x8 mx = *reinterpret_cast<const f16x8*>(s_conv + ((2 * py) * 17 + 2 * px) * kConvPitch + cg * 16);
#pragma unroll
      for (int dy = 0; dy < 3; dy++)
#pragma unroll
        for (int dx = 0; dx < 3; dx++) {
          if (dy == 0 && dx == 0) continue;
          const f16x8 v = *reinterpret_cast<const f16x8*>(s_conv + ((2 * py + dy) * 17 + 2 * px + dx) * kConvPitch + cg * 16);
#pragma unroll
          for (int e = 0; e < 8; e++) mx[e] = v[e] > mx[e] ? v[e] : mx[e];
        }
      if (oy < Hp && ox < Wp)
        *reinterpret_cast<f16x8*>(out + (((int64_t)b * Hp + oy) * Wp + ox) * 64 + cg * 8) = mx;
    }
  }
}

}  // namespace
}  // namespace s2a

using namespace s2a;

extern "C" int s2a_stem_pack_weight_f16(const void* weight, void* packed, s2a_stream_t stream) {
  S2A_CHECK_ARG(weight && packed, "stem_pack_weight: NULL tensor");
  const int total = 2 * kKSteps * 64 * 8;
  k_stem_pack<<<(total + 255) / 256, 256, 0, as_stream(stream)>>>((const _Float16*)weight, (_Float16*)packed);
  S2A_LAUNCH_CHECK();
  return S2A_OK;
}

extern "C" int64_t s2a_stem_packed_elems(void) { return 2 * kKSteps * 64 * 8; }

extern "C" int s2a_stem_u8_f16(const void* image_u8, const void* weight_packed, const void* bias, void* out,
                               int64_t batch, int64_t height, int64_t width, float divisor,
                               s2a_stream_t stream) {
  S2A_CHECK_ARG(batch >= 0 && height >= 7 && width >= 7, "stem: bad shape");
  S2A_CHECK_ARG(width % 4 == 0, "stem: image width must be a multiple of 4 (aligned 32-bit loads of RGB rows)");
  S2A_CHECK_ARG(divisor > 0, "stem: divisor must be positive");
  S2A_CHECK_ARG(batch * height * width * 3 < (1ll << 40), "stem: image too large");
  if (batch == 0) return S2A_OK;
  S2A_CHECK_ARG(image_u8 && weight_packed && out, "stem: NULL tensor");
  S2A_CHECK_ARG(((uintptr_t)image_u8 % 4) == 0 && ((uintptr_t)weight_packed % 16) == 0 && ((uintptr_t)out % 16) == 0 &&
                ((uintptr_t)bias % 2) == 0, "stem: misaligned tensor");
  const int H = (int)height, W = (int)width;
  const int Hc = (H - 1) / 2 + 1, Wc = (W - 1) / 2 + 1;        // 7x7 / 2 / pad 3
  const int Hp = (Hc - 1) / 2 + 1, Wp = (Wc - 1) / 2 + 1;      // 3x3 / 2 / pad 1
  const int tiles_x = (Wp + 7) / 8, tiles_y = (Hp + 7) / 8;
  const int64_t tiles = batch * tiles_x * tiles_y;
  S2A_CHECK_ARG(tiles < (1ll << 31), "stem: too many tiles");
  auto kern = k_stem;
  S2A_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kStemLds));
  // persistent grid: two workgroups fit a CU (LDS)
  const unsigned grid = (unsigned)std::min<int64_t>(tiles, 256 * 2);
  kern<<<grid, 256, kStemLds, as_stream(stream)>>>((const uint8_t*)image_u8, (const _Float16*)weight_packed,
                                                    (const _Float16*)bias, (_Float16*)out, H, W, Hc, Wc, Hp,
                                                    Wp, tiles_x, tiles_y, divisor, (int)tiles);
  S2A_LAUNCH_CHECK();
  return S2A_OK;
}
