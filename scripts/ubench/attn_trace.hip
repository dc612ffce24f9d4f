// Builder-side experiment: the S = 256 attention kernel (bert_attn.h: attention_s256_kernel) on its own, with per-phase s_memtime stamps.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Icapreolus_amd/csrc -Iinclude scripts/ubench/attn_trace.hip -o scripts/ubench/attn_trace
#define CAPAMD_ATTN_TRACE 1
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "bert_attn.h"
using namespace capamd;

__global__ void fill(_Float16* p, size_t n, unsigned seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[i] = (_Float16)(((int)(x & 1023) - 512) * (1.f / 1024.f));
  }
}
__global__ void fill_mask(int64_t* m, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) m[i] = 1;
}

int main(int argc, char** argv) {
  const int npsg = argc > 1 ? atoi(argv[1]) : 256, H = 768, heads = 12, S = 256;
  const int grid = argc > 2 ? atoi(argv[2]) : 512;
  const size_t M = (size_t)npsg * S;
  _Float16 *q, *k, *vt, *ctx;
  int64_t* mask;
  hipMalloc(&q, M * H * 2); hipMalloc(&k, M * H * 2); hipMalloc(&vt, M * H * 2); hipMalloc(&ctx, M * H * 2); hipMalloc(&mask, M * 8);
  fill<<<2048, 256>>>(q, M * H, 1); fill<<<2048, 256>>>(k, M * H, 2); fill<<<2048, 256>>>(vt, M * H, 3); fill_mask<<<2048, 256>>>(mask, M);
  unsigned long long* trace;
  hipMalloc(&trace, (size_t)grid * 8 * 16 * 8);
  hipMemset(trace, 0, (size_t)grid * 8 * 16 * 8);
  hipMemcpyToSymbol(HIP_SYMBOL(g_attn_trace), &trace, sizeof(trace));
  AttnArgs a{q, k, vt, mask, ctx, H, heads, 1, 1};
  const int n_items = npsg * heads;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((attention_s256_kernel<_Float16, true, true>), dim3(grid), dim3(256), 0, 0, a, n_items);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("grid %d items %d: %.1f us per launch\n", grid, n_items, ms * 100);
  }
  std::vector<unsigned long long> h((size_t)grid * 8 * 16);
  hipMemcpy(h.data(), trace, h.size() * 8, hipMemcpyDeviceToHost);
  const char* names[9] = {"scores A (+V DMA) + softmax A", "wait V + B1", "P V A + stores", "scores B + softmax B", "B2 + madd", "P V B (+K DMA, Q loads) + stores",
                          "wait K/Q + B3", "-", "-"};
  for (int it = 0; it < 6; ++it) {
    double sum[9] = {0};
    int n = 0;
    for (int b = 0; b < grid; ++b) {
      const unsigned long long* t = &h[((size_t)b * 8 + it) * 16];
      if (!t[7]) continue;
      for (int j = 0; j < 7; ++j) sum[j] += (double)(t[j + 1] - t[j]);
      ++n;
    }
    if (!n) continue;
    printf("item %d of a workgroup (%d workgroups):", it, n);
    double tot = 0;
    for (int j = 0; j < 7; ++j) { printf(" %s %.0f |", names[j], sum[j] / n); tot += sum[j] / n; }
    printf(" total %.0f cycles\n", tot);
  }
  return 0;
}
