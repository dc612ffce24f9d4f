// Experiment: what the memory system sustains for reads on this part, as the yardstick for roofline.frac of the gather kernels
// (DESIGN.md section 4): (a) a streaming read of an 8 GiB buffer, (b) whole 1280-byte rows (KNRM's packed embedding rows) in random
// order from a 5.1 GB table - 16 B per lane, several loads in flight per lane, results folded into one value per thread.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/ubench/hbm_read.hip -o scripts/ubench/hbm_read
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include <math.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int U>
__global__ __launch_bounds__(256) void stream_read(const uint4* __restrict__ p, size_t n16, unsigned* out) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned acc = 0;
  for (; i + (U - 1) * stride < n16; i += U * stride) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + i + u * stride));
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// one 16-lane group per row (80 x 16 B = 1280 B: 5 loads per lane), rows taken from a shuffled index list
template <int ROWS_IN_FLIGHT>
__global__ __launch_bounds__(256) void row_gather(const uint4* __restrict__ table, const int* __restrict__ rows, size_t n_rows, unsigned* out) {
  const int lane16 = threadIdx.x & 15;
  const size_t group = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4, n_groups = ((size_t)gridDim.x * blockDim.x) >> 4;
  unsigned acc = 0;
  for (size_t r = group * ROWS_IN_FLIGHT; r + ROWS_IN_FLIGHT <= n_rows; r += n_groups * ROWS_IN_FLIGHT) {
    uint4 v[ROWS_IN_FLIGHT][5];
#pragma unroll
    for (int j = 0; j < ROWS_IN_FLIGHT; ++j) {
      const uint4* src = table + (size_t)rows[r + j] * 80 + lane16;
#pragma unroll
      for (int c = 0; c < 5; ++c) v[j][c] = src[c * 16];
    }
#pragma unroll
    for (int j = 0; j < ROWS_IN_FLIGHT; ++j)
#pragma unroll
      for (int c = 0; c < 5; ++c) acc += v[j][c].x ^ v[j][c].y ^ v[j][c].z ^ v[j][c].w;
  }
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <typename F>
static float time_ms(F&& launch, int reps) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  launch();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a, 0);
  for (int i = 0; i < reps; ++i) launch();
  (void)hipEventRecord(b, 0);
  (void)hipEventSynchronize(b);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

int main() {
  const size_t bytes = (size_t)8 << 30, n16 = bytes / 16;
  uint4* buf; unsigned* out;
  CK(hipMalloc(&buf, bytes));
  CK(hipMalloc(&out, (size_t)256 * 64 * 256 * 4));
  CK(hipMemset(buf, 1, bytes));
  for (int wgs_per_cu : {4, 8, 16, 32}) {
    const int grid = 256 * wgs_per_cu;
    float ms = time_ms([&] { hipLaunchKernelGGL(stream_read<8>, dim3(grid), dim3(256), 0, 0, buf, n16, out); }, 5);
    printf("streaming read of 8 GiB, %2d workgroups per CU, 8 x 16 B in flight per lane: %.2f TB/s\n", wgs_per_cu, bytes / ms / 1e9);
  }
  const size_t V = 4000001, n_rows = (size_t)16 << 20;     // 16 Mi row reads of 1280 B = 21.5 GB per launch
  std::vector<int> rows(n_rows);
  unsigned long long s = 88172645463325252ull;
  for (auto& r : rows) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; r = (int)(s % V); }
  int* d_rows;
  CK(hipMalloc(&d_rows, n_rows * 4));
  CK(hipMemcpy(d_rows, rows.data(), n_rows * 4, hipMemcpyHostToDevice));
  // (the table is the first 5.12 GB of the same buffer: V * 1280 B)
  for (int wgs_per_cu : {8, 16, 32}) {
    const int grid = 256 * wgs_per_cu;
    float m1 = time_ms([&] { hipLaunchKernelGGL(row_gather<1>, dim3(grid), dim3(256), 0, 0, buf, d_rows, n_rows, out); }, 3);
    float m2 = time_ms([&] { hipLaunchKernelGGL(row_gather<2>, dim3(grid), dim3(256), 0, 0, buf, d_rows, n_rows, out); }, 3);
    float m4 = time_ms([&] { hipLaunchKernelGGL(row_gather<4>, dim3(grid), dim3(256), 0, 0, buf, d_rows, n_rows, out); }, 3);
    printf("random 1280-byte rows of a 5.1 GB table, %2d workgroups per CU: %.2f / %.2f / %.2f TB/s with 1 / 2 / 4 rows in flight per 16-lane group\n",
           wgs_per_cu, n_rows * 1280.0 / m1 / 1e9, n_rows * 1280.0 / m2 / 1e9, n_rows * 1280.0 / m4 / 1e9);
  }
  // (c) the headline leg's request stream: Zipf(1.1) term ids over a 400,001-row table (512 MB - twice the Infinity Cache), documents of 303
  //     draws with repeats inside a document removed (what the distinct-term pass requests), documents handed to consecutive groups
  {
    const size_t Vz = 400001;
    std::vector<double> cdf(Vz);
    double z = 0;
    for (size_t i = 1; i < Vz; ++i) { z += pow((double)i, -1.1); cdf[i] = z; }
    std::vector<int> zr;
    zr.reserve(n_rows + 512);
    std::vector<int> doc;
    while (zr.size() < n_rows) {
      doc.clear();
      for (int t = 0; t < 303; ++t) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double u = (double)(s >> 11) / 9007199254740992.0 * z;
        doc.push_back((int)(std::lower_bound(cdf.begin() + 1, cdf.end(), u) - cdf.begin()));
      }
      std::sort(doc.begin(), doc.end());
      doc.erase(std::unique(doc.begin(), doc.end()), doc.end());
      // (original order does not matter to the memory system; shuffle lightly so that neighbours are not sorted by rank)
      for (size_t i = doc.size(); i > 1; --i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; std::swap(doc[i - 1], doc[s % i]); }
      zr.insert(zr.end(), doc.begin(), doc.end());
    }
    zr.resize(n_rows);
    CK(hipMemcpy(d_rows, zr.data(), n_rows * 4, hipMemcpyHostToDevice));
    for (int wgs_per_cu : {8, 16, 32}) {
      const int grid = 256 * wgs_per_cu;
      float m1 = time_ms([&] { hipLaunchKernelGGL(row_gather<1>, dim3(grid), dim3(256), 0, 0, buf, d_rows, n_rows, out); }, 3);
      float m2 = time_ms([&] { hipLaunchKernelGGL(row_gather<2>, dim3(grid), dim3(256), 0, 0, buf, d_rows, n_rows, out); }, 3);
      float m4 = time_ms([&] { hipLaunchKernelGGL(row_gather<4>, dim3(grid), dim3(256), 0, 0, buf, d_rows, n_rows, out); }, 3);
      printf("Zipf(1.1) rows of a 512 MB table, distinct within 303-term documents, %2d workgroups per CU: %.2f / %.2f / %.2f TB/s with 1 / 2 / 4 rows in flight\n",
             wgs_per_cu, n_rows * 1280.0 / m1 / 1e9, n_rows * 1280.0 / m2 / 1e9, n_rows * 1280.0 / m4 / 1e9);
    }
  }
  return 0;
}
