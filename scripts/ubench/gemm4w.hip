// Experiment (not part of the product path): K-loop rate of a 4-wave 256x256x64 tile (one wave per SIMD, 128x128 per wave,
// 256 accumulator registers, LDS-DMA fill, double-buffered) next to the 8-wave ping-pong kernel of bert_gemm.h.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/ubench/gemm4w.hip -o scripts/ubench/gemm4w && scripts/ubench/gemm4w
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void lds_void_t;

constexpr int kStage = 64 * 1024;   // A 32 KB | B 32 KB
#ifndef ABL
#define ABL 0      // bit 0: no LDS-DMA fill in the loop, bit 1: fragments read once (no LDS reads in the loop), bit 2: one fill pair per 4 MFMAs,
                   // bit 3: fill through registers (global_load_dwordx4 at the top of the K step, ds_write_b128 at its end) instead of LDS-DMA
#endif

__global__ __launch_bounds__(256, 1) void gemm4w(const _Float16* __restrict__ A, const _Float16* __restrict__ W, _Float16* __restrict__ C, int M,
                                                 int N, int K, unsigned long long* stamps) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int tn = N / 256, m0 = (blockIdx.x / tn) * 256, n0 = (blockIdx.x % tn) * 256;
  const int KT = K / 64;
  // fill: instruction i of this wave covers rows wave*64 + i*8 + lane/8 of the operand tile, slot lane%8; the source chunk is swizzled
  int a_off[8], b_off[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = wave * 64 + i * 8 + (lane >> 3), chunk = (lane & 7) ^ (row & 7);
    a_off[i] = ((m0 + row) * K + chunk * 8) * 2;
    b_off[i] = ((n0 + row) * K + chunk * 8) * 2;
  }
  const auto ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(A), 0, (int)((size_t)M * K * 2), 0x00020000);
  const auto rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(W), 0, (int)((size_t)N * K * 2), 0x00020000);
  auto fill = [&](int stage, int kt, int lo, int hi) {
    char* da = lds + stage * kStage + wave * 8192;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (i >= lo && i < hi) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void_t*)(da + i * 1024), 16, a_off[i], kt * 128, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void_t*)(da + 32768 + i * 1024), 16, b_off[i], kt * 128, 0, 0);
      }
  };
  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x16{0};
  int arow[4], brow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    arow[i] = (wm * 128 + i * 32 + l31) * 128;
    brow[i] = 32768 + (wn * 128 + i * 32 + l31) * 128;
  }
  const int sw = l31 & 7;   // (row & 7) of every fragment row of this lane
  fill(0, 0, 0, 8);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  unsigned long long t0 = 0;
  if (stamps && tid == 0) t0 = __builtin_readcyclecounter();
  typedef __attribute__((ext_vector_type(4))) unsigned u4;
  // register-staged fill: lane loads chunk (lane & 7) of row wave*64 + i*8 + lane/8 and writes it to the swizzled slot
  int g_off[8], l_off[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = wave * 64 + i * 8 + (lane >> 3), chunk = lane & 7;
    g_off[i] = (row * K + chunk * 8) * 2;
    l_off[i] = row * 128 + ((chunk ^ (row & 7)) << 4);
  }
  const char* Ab = reinterpret_cast<const char*>(A) + (size_t)m0 * K * 2;
  const char* Wb = reinterpret_cast<const char*>(W) + (size_t)n0 * K * 2;
  for (int kt = 0; kt < KT; ++kt) {
    const char* st = lds + (kt & 1) * kStage;
    const bool more = kt + 1 < KT;
    u4 ra4[8], rb4[8];
    if ((ABL & 8) && more) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        ra4[i] = *reinterpret_cast<const u4*>(Ab + g_off[i] + (kt + 1) * 128);
        rb4[i] = *reinterpret_cast<const u4*>(Wb + g_off[i] + (kt + 1) * 128);
      }
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if (more && !(ABL & 1) && !(ABL & 4) && !(ABL & 8)) fill((kt + 1) & 1, kt + 1, kk * 2, kk * 2 + 2);   // 4 of the 16 DMA instructions of the next step per k-slice
      h8 fa[4], fb[4];
      const int ko = (ABL & 2) ? 0 : ((2 * kk + half) ^ sw) * 16;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa[i] = *reinterpret_cast<const h8*>(((ABL & 2) ? lds : st) + arow[i] + ko);
        fb[i] = *reinterpret_cast<const h8*>(((ABL & 2) ? lds : st) + brow[i] + ko);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (more && !(ABL & 1) && (ABL & 4) && !(ABL & 8) && (i & 1) == 0) fill((kt + 1) & 1, kt + 1, kk * 2 + (i >> 1), kk * 2 + (i >> 1) + 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
    }
    if ((ABL & 8) && more) {
      char* dn = lds + ((kt + 1) & 1) * kStage;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        *reinterpret_cast<u4*>(dn + l_off[i]) = ra4[i];
        *reinterpret_cast<u4*>(dn + 32768 + l_off[i]) = rb4[i];
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  if (stamps && tid == 0) {
    stamps[2 * blockIdx.x] = __builtin_readcyclecounter() - t0;
  }
  // plain epilogue: fp16 row-major, lane writes 4 consecutive columns? (accumulator register r of tile (i, j): row 8*(r/4) + 4*half + r%4, col l31)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 128 + i * 32 + (r >> 2) * 8 + half * 4 + (r & 3), col = n0 + wn * 128 + j * 32 + l31;
        C[(size_t)row * N + col] = (_Float16)acc[i][j][r];
      }
  if (stamps && tid == 0) stamps[2 * blockIdx.x + 1] = __builtin_readcyclecounter() - t0;
#endif
}

__global__ void init(_Float16* p, size_t n, unsigned seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed;
    x ^= x >> 13; x *= 0x5bd1e995u; x ^= x >> 15;
    p[i] = (_Float16)(((int)(x & 255) - 128) / 256.0f);
  }
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 65536, N = argc > 2 ? atoi(argv[2]) : 3072, K = argc > 3 ? atoi(argv[3]) : 768;
  _Float16 *A, *W, *C;
  unsigned long long* st;
  hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&W, (size_t)N * K * 2); hipMalloc(&C, (size_t)M * N * 2);
  const int tiles = (M / 256) * (N / 256);
  hipMalloc(&st, (size_t)tiles * 16);
  init<<<1024, 256>>>(A, (size_t)M * K, 1); init<<<1024, 256>>>(W, (size_t)N * K, 2);
  hipFuncSetAttribute(reinterpret_cast<const void*>(gemm4w), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kStage);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) gemm4w<<<tiles, 256, 2 * kStage>>>(A, W, C, M, N, K, st);
  hipEventRecord(e0);
  const int reps = 10;
  for (int it = 0; it < reps; ++it) gemm4w<<<tiles, 256, 2 * kStage>>>(A, W, C, M, N, K, st);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  std::vector<unsigned long long> h(2 * tiles);
  hipMemcpy(h.data(), st, (size_t)tiles * 16, hipMemcpyDeviceToHost);
  double kl = 0, tot = 0;
  for (int i = 0; i < tiles; ++i) { kl += h[2 * i]; tot += h[2 * i + 1]; }
  // spot check
  std::vector<_Float16> ha((size_t)K), hw((size_t)K); _Float16 c;
  double maxerr = 0;
  for (int t = 0; t < 8; ++t) {
    const int r = (t * 7919) % M, cc = (t * 104729) % N;
    hipMemcpy(ha.data(), A + (size_t)r * K, K * 2, hipMemcpyDeviceToHost); hipMemcpy(hw.data(), W + (size_t)cc * K, K * 2, hipMemcpyDeviceToHost);
    hipMemcpy(&c, C + (size_t)r * N + cc, 2, hipMemcpyDeviceToHost);
    double s = 0; for (int k = 0; k < K; ++k) s += (double)(float)ha[k] * (float)hw[k];
    const double e = fabs(s - (double)(float)c) / (fabs(s) + 1e-3); if (e > maxerr) maxerr = e;
  }
  printf("M=%d N=%d K=%d: %.1f us  %.1f TFLOP/s  | per tile: K loop %.0f ticks (%.0f per K step), with epilogue %.0f ticks (100 MHz counter)  | spot err %.2e\n", M, N, K,
         ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12, kl / tiles, kl / tiles / (K / 64), tot / tiles, maxerr);
  return 0;
}
