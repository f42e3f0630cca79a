// minimal reproduction: rocprim::select (counting iterator + predicate flags, as s2a_multiclass_candidates uses it)
// captured into a HIP graph and replayed several times.   usage: select_graph <total> <fraction selected in %> <mode>
// mode 0: the library entry point s2a_multiclass_candidates; 1: bare rocprim::select with the same iterator types;
// 2: bare rocprim::select with a plain uint8 flag array
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../include/s2anet_hip.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
struct Above { const float* s; float thr; __device__ uint8_t operator()(int i) const { return s[i] > thr ? 1 : 0; } };
int main(int argc, char** argv) {
  const long total = argc > 1 ? atol(argv[1]) : 76770;
  const int pct = argc > 2 ? atoi(argv[2]) : 100, mode = argc > 3 ? atoi(argv[3]) : 0;
  const long B = 2, C = 15, n = total / (B * C);
  const long tot = B * n * C;
  std::vector<float> h(tot);
  for (long i = 0; i < tot; i++) h[i] = (i * 7919 % 100) < pct ? 0.9f : 0.01f;
  float *scores, *boxes, *ob, *os; int *oseg, *ogrp, *ocls, *sel; long long* cnt; void* ws; uint8_t* flags8;
  CK(hipMalloc(&scores, tot * 4)); CK(hipMalloc(&boxes, B * n * 5 * 4)); CK(hipMemset(boxes, 0, B * n * 5 * 4));
  CK(hipMemcpy(scores, h.data(), tot * 4, hipMemcpyHostToDevice));
  const long cap = tot;
  CK(hipMalloc(&ob, cap * 20)); CK(hipMalloc(&os, cap * 4)); CK(hipMalloc(&oseg, cap * 4)); CK(hipMalloc(&ogrp, cap * 4));
  CK(hipMalloc(&ocls, cap * 4)); CK(hipMalloc(&cnt, 16)); CK(hipMalloc(&sel, tot * 4)); CK(hipMalloc(&flags8, tot));
  size_t wsb = s2a_multiclass_candidates_workspace_bytes(tot) + (1 << 20);
  CK(hipMalloc(&ws, wsb));
  hipStream_t st; CK(hipStreamCreate(&st));
  auto call = [&]() -> int {
    if (mode == 0)
      return s2a_multiclass_candidates(boxes, scores, B, n, C, 0.05f, cap, ob, os, oseg, ogrp, ocls, (int64_t*)cnt, ws, wsb, st);
    size_t tb = wsb;
    if (mode == 1) {
      rocprim::counting_iterator<int32_t> ids(0);
      auto fl = rocprim::make_transform_iterator(ids, Above{scores, 0.05f});
      return (int)rocprim::select(ws, tb, ids, fl, sel, (unsigned long long*)cnt, (size_t)tot, st);
    }
    return (int)rocprim::select(ws, tb, scores, flags8, os, (unsigned long long*)cnt, (size_t)tot, st);
  };
  CK(hipMemset(flags8, pct >= 100 ? 1 : 0, tot));
  if (call() != 0) { printf("eager call failed: %s\n", mode == 0 ? s2a_last_error() : "rocprim"); return 3; }
  CK(hipStreamSynchronize(st));
  long long hc = 0; CK(hipMemcpy(&hc, cnt, 8, hipMemcpyDeviceToHost));
  printf("mode %d total %ld eager selected %lld\n", mode, tot, hc); fflush(stdout);
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  if (call() != 0) { printf("captured call failed\n"); return 4; }
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  size_t nn = 0; CK(hipGraphGetNodes(g, nullptr, &nn)); printf("graph nodes %zu\n", nn); fflush(stdout);
  for (int r = 0; r < 4; r++) {
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    CK(hipMemcpy(&hc, cnt, 8, hipMemcpyDeviceToHost));
    printf("replay %d selected %lld\n", r, hc); fflush(stdout);
  }
  printf("DONE\n");
  return 0;
}
