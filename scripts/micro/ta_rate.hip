// How many bytes per clock does ONE CU's vector-memory path deliver for the AlignConv filter stream?
// 256 workgroups (one per CU) of W waves; every wave issues back-to-back 1 KB loads (16 B per lane, contiguous per wave) that
// walk a 1 MB L2-resident buffer the way the matrix waves walk the packed filter; results are xor-ed so nothing is dropped.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 scripts/micro/ta_rate.hip -o /tmp/ta_rate && /tmp/ta_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
template <int INFLIGHT>
__global__ __launch_bounds__(512) void k_stream(const u32x4* __restrict__ buf, size_t nvec, int iters, unsigned* out,
                                                unsigned long long* cyc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  u32x4 acc = {0, 0, 0, 0};
  unsigned pos = ((blockIdx.x * 8u + wave) * 64u * 97u) & (unsigned)(nvec - 1);      // (nvec is a power of two)
  for (int i = 0; i < iters; i++) {
    u32x4 v[INFLIGHT];
#pragma unroll
    for (int k = 0; k < INFLIGHT; k++) {
      v[k] = buf[(pos + lane) & (unsigned)(nvec - 1)];
      pos = (pos + 64u * 131u) & (unsigned)(nvec - 1);
    }
#pragma unroll
    for (int k = 0; k < INFLIGHT; k++) acc ^= v[k];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (acc[0] == 0x12345678u && acc[1] == 1u) out[0] = acc[2] ^ acc[3];
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
int main() {
  const size_t nvec = (1024 << 10) / 16;                 // 1 MB (the packed filter is 1.2 MB): L2-resident
  std::vector<unsigned> h(nvec * 4, 1u);
  u32x4* d; unsigned* o; unsigned long long* c;
  hipMalloc(&d, nvec * 16); hipMalloc(&o, 64); hipMalloc(&c, 256 * 8 * 8);
  hipMemcpy(d, h.data(), nvec * 16, hipMemcpyHostToDevice);
  const int iters = 2000;
  for (int waves : {1, 2, 4, 8}) {   // (4 = the matrix waves of k_dcn_patch)
    for (int rep = 0; rep < 2; rep++) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      k_stream<8><<<256, 64 * waves>>>(d, nvec, iters, o, c);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      std::vector<unsigned long long> hc(256 * 8);
      hipMemcpy(hc.data(), c, hc.size() * 8, hipMemcpyDeviceToHost);
      double cy = 0; for (int b = 0; b < 256; b++) cy += hc[b * 8]; cy /= 256;
      const double bytes_cu = (double)waves * iters * 8 * 1024;
      if (rep) printf("%d wave(s) per CU, 8 loads of 1 KB in flight per wave: %.1f B/clk per CU (%.0f cycles per 1 KB load and wave), %.2f TB/s chip-wide\n",
                      waves, bytes_cu / cy, cy / (iters * 8.0), bytes_cu * 256 / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
