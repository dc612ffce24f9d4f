// Experiment: sustained rate and shader clock of back-to-back MFMAs (registers only, one wave per SIMD, every CU) for the two fp16 tile
// shapes - what the power management leaves of the 2.5 PFLOP/s peak, independent of any memory system effect.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/ubench/mfma_power.hip -o scripts/ubench/mfma_power
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int SHAPE, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void burn(const _Float16* __restrict__ src, float* out, int iters, unsigned long long* stamps) {
  const int lane = threadIdx.x & 63;
  h8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = *reinterpret_cast<const h8*>(src + (size_t)(blockIdx.x * 512 + threadIdx.x) * 64 + i * 8);
    b[i] = *reinterpret_cast<const h8*>(src + (size_t)(blockIdx.x * 512 + threadIdx.x) * 64 + 32 + i * 8);
  }
  const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  if (SHAPE == 32) {
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x16{0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) s += acc[i][j][lane & 15];
  } else {
    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j & 3], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) s += acc[i][j][lane & 3];
  }
  const unsigned long long dc = __builtin_readcyclecounter() - c0, dr = __builtin_amdgcn_s_memrealtime() - r0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = dc; stamps[2 * blockIdx.x + 1] = dr; }
}

__global__ void init(_Float16* p, size_t n, int mode) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + 12345u;
    x ^= x >> 13; x *= 0x5bd1e995u; x ^= x >> 15;
    float v = mode == 0 ? 0.f : mode == 1 ? ((int)(x & 255) - 128) / 256.0f : (float)((int)(x & 0xffff) - 32768) / 16384.0f;
    p[i] = (_Float16)v;
  }
}

template <int SHAPE, int WAVES>
void run(const _Float16* src, float* out, unsigned long long* st, const char* tag, int mode) {
  const int iters = 20000, grid = 256;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  burn<SHAPE, WAVES><<<grid, 64 * WAVES>>>(src, out, 2000, st);
  hipEventRecord(e0);
  burn<SHAPE, WAVES><<<grid, 64 * WAVES>>>(src, out, iters, st);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[512]; hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
  double mhz = 0; for (int i = 0; i < 256; ++i) mhz += 100.0 * h[2 * i] / h[2 * i + 1]; mhz /= 256;
  const double flops = (double)grid * 4 * WAVES / 4 * iters * (SHAPE == 32 ? 16 * 32768.0 : 32 * 16384.0) * 1.0;
  printf("%-28s data mode %d: %.1f TFLOP/s at %.0f MHz (%.1f%% of the MFMA issue rate at that clock)\n", tag, mode, flops * (WAVES >= 4 ? 1 : 1) / (ms * 1e-3) / 1e12, mhz,
         100.0 * (flops / (ms * 1e-3)) / (2.5e15 * mhz / 2400.0));
}

int main() {
  _Float16* src; float* out; unsigned long long* st;
  hipMalloc(&src, (size_t)256 * 512 * 64 * 2); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&st, 512 * 8);
  for (int mode = 0; mode < 3; ++mode) {
    init<<<256, 256>>>(src, (size_t)256 * 512 * 64, mode);
    run<32, 4>(src, out, st, "32x32x16, 1 wave/SIMD", mode);
    run<16, 4>(src, out, st, "16x16x32, 1 wave/SIMD", mode);
    run<32, 8>(src, out, st, "32x32x16, 2 waves/SIMD", mode);
  }
  return 0;
}
