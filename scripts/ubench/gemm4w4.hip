// Variant of gemm4w2.hip with 32-k ring slots (two 16-k sub-slices per slot, 4 slots, ONE barrier per 32 MFMAs instead of per 16).
// Experiment (not part of the product path): K loop of a 4-wave 256x256 tile (one wave per SIMD, 128x128 per wave, 256 accumulator
// registers) fed by a k-slice ring: both operands chunk-major ([rows/32][K/8][32][8], so a 32-row x 16-k slice is 1 KiB contiguous in
// memory AND in LDS: one LDS-DMA instruction, conflict-free ds_read_b128 fragments, no swizzle), 8 slots of 16 KiB (two K steps), one
// barrier per 16-MFMA k-slice, slice t+8 issued while slice t is consumed.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/ubench/gemm4w2.hip -o scripts/ubench/gemm4w2 && scripts/ubench/gemm4w2
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include <vector>

typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void lds_void_t;

constexpr int kSlot = 32 * 1024;    // one k-slice (32 k): two sub-slices of [A 8 row-blocks x 1 KiB | B 8 row-blocks x 1 KiB]
constexpr int kSlots = 4;
#ifndef ABL
#define ABL 0      // bit 0: no LDS-DMA in the loop, bit 1: no LDS reads in the loop, bit 2: no sched_group_barrier pattern
#endif
#ifndef PERSIST
#define PERSIST 1
#endif

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__global__ __launch_bounds__(256, 1) void gemm4w4(const _Float16* __restrict__ A, const _Float16* __restrict__ W, _Float16* __restrict__ C, int M,
                                                  int N, int K, unsigned long long* stamps, int ntiles) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int tn = N / 256;
  const int S = K / 32;                      // k-slices per tile
  const auto ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(A), 0, (int)((size_t)M * K * 2), 0x00020000);
  const auto rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(W), 0, (int)((size_t)N * K * 2), 0x00020000);
  const int voff = lane * 16;
  const int a_base = wm * 4 * 1024 + half * 512 + l31 * 16;
  const int b_base = 8192 + wn * 4 * 1024 + half * 512 + l31 * 16;

  const int step = PERSIST ? (int)gridDim.x : ntiles;
  // XCD-aware start: hardware block b runs on XCD b % 8; consecutive logical ids (= consecutive tiles, n fastest: one A panel shared by
  // the tn column tiles) go to the same XCD, i.e. the same L2
  int tile = (gridDim.x % 8 == 0) ? (int)((blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8) : (int)blockIdx.x;
  int m0 = (tile / tn) * 256, n0 = (tile % tn) * 256;
  // scalar byte offsets of this wave's pieces (row-blocks 2w, 2w+1 of the A and of the B panel) of k-slice 0 of a tile; a k-slice
  // further is +1024 bytes.
  auto bases = [&](int t, unsigned (&sa)[2], unsigned (&sb)[2]) {
    if (t >= ntiles) t = 0;     // nothing follows: re-read tile 0's slices into slots nobody consumes (the scalar offset is not range-checked)
    const int tm = (t / tn) * 256, tn0 = (t % tn) * 256;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      sa[p] = (unsigned)(((tm >> 5) + wave * 2 + p) * (K >> 3)) * 512u;
      sb[p] = (unsigned)(((tn0 >> 5) + wave * 2 + p) * (K >> 3)) * 512u;
    }
  };
  auto issue = [&](const unsigned (&sa)[2], const unsigned (&sb)[2], int s, int slot_off) {
    char* d = lds + slot_off;
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int rb = wave * 2 + p;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void_t*)(d + h2 * 16384 + rb * 1024), 16, voff, (int)(sa[p] + (unsigned)(2 * s + h2) * 1024u), 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void_t*)(d + h2 * 16384 + 8192 + rb * 1024), 16, voff, (int)(sb[p] + (unsigned)(2 * s + h2) * 1024u), 0, 0);
      }
  };
  unsigned ca[2], cb[2], na[2], nb[2];
  bases(tile, ca, cb);
#pragma unroll 1
  for (int i = 0; i < kSlots; ++i) issue(ca, cb, i, i * kSlot);       // (S >= 4)
  unsigned long long t0 = 0, kacc = 0;
  const unsigned long long c_begin = __builtin_readcyclecounter(), r_begin = __builtin_amdgcn_s_memrealtime();
  h8 fa[2][2][4], fb[2][2][4];
  // first slice's fragments (later tiles find theirs loaded by the last iteration of the tile before)
  wait_vmcnt<24>();
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      fa[0][h2][i] = *reinterpret_cast<const h8*>(lds + h2 * 16384 + a_base + i * 1024);
      fb[0][h2][i] = *reinterpret_cast<const h8*>(lds + h2 * 16384 + b_base + i * 1024);
    }
  int slot_off = 0;   // byte offset of the slot of the slice being consumed
  for (; tile < ntiles; tile += step) {
    m0 = (tile / tn) * 256; n0 = (tile % tn) * 256;
    bases(tile + step, na, nb);
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x16{0};
    if (stamps && tid == 0) t0 = __builtin_readcyclecounter();
    // one k-slice: its fragments are in registers once lgkmcnt(0); slice +1 has landed once vmcnt(24) on every wave + barrier; then the
    // fragment reads of slice +1, the LDS-DMA of slice +8 into the slot every wave has just finished reading, and 16 MFMAs
    auto kslice = [&](auto u_c, const unsigned (&sa)[2], const unsigned (&sb)[2], int fs) {
      constexpr int cur = decltype(u_c)::value, nxt = cur ^ 1;
      wait_vmcnt<16>();      // slice +1 has landed: two younger slices x 8 pieces stay in flight
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const char* stn = lds + ((slot_off + kSlot) & (kSlots * kSlot - 1));
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          fa[nxt][h2][i] = *reinterpret_cast<const h8*>(stn + h2 * 16384 + a_base + i * 1024);
          fb[nxt][h2][i] = *reinterpret_cast<const h8*>(stn + h2 * 16384 + b_base + i * 1024);
        }
      issue(sa, sb, fs, slot_off);
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[cur][h2][i], fa[cur][h2][j], acc[i][j], 0, 0, 0);
      // 32 MFMA | 16 DS read | 8 VMEM: M R M V M R M  x8
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
      slot_off = (slot_off + kSlot) & (kSlots * kSlot - 1);
    };
    using U0 = std::integral_constant<int, 0>;
    using U1 = std::integral_constant<int, 1>;
    // slices 0 .. S-9 refill with slices 8 .. S-1 of this tile, the last eight with slices 0 .. 7 of the block's next tile
#pragma unroll 1
    for (int s = 0; s < S - 4; s += 2) { kslice(U0{}, ca, cb, s + 4); kslice(U1{}, ca, cb, s + 5); }
#pragma unroll 1
    for (int s = 0; s < 4; s += 2) { kslice(U0{}, na, nb, s); kslice(U1{}, na, nb, s + 1); }
    ca[0] = na[0]; ca[1] = na[1]; cb[0] = nb[0]; cb[1] = nb[1];
    if (stamps && tid == 0) kacc += __builtin_readcyclecounter() - t0;
    // plain epilogue: fp16 row-major (accumulator register r of tile (i = n tile, j = m tile): n = 8*(r/4) + 4*half + r%4, m = l31)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; r += 4) {
          const int row = m0 + wm * 128 + j * 32 + l31, col = n0 + wn * 128 + i * 32 + (r >> 2) * 8 + half * 4;
          typedef __attribute__((ext_vector_type(4))) _Float16 h4;
          const h4 o = {(_Float16)acc[i][j][r], (_Float16)acc[i][j][r + 1], (_Float16)acc[i][j][r + 2], (_Float16)acc[i][j][r + 3]};
          *reinterpret_cast<h4*>(C + (size_t)row * N + col) = o;
        }
  }
  if (stamps && tid == 0) {
    stamps[2 * blockIdx.x] = kacc;
    // shader clock in kHz: cycle counter ticks per 100 MHz real-time tick
    const unsigned long long dc = __builtin_readcyclecounter() - c_begin, dr = __builtin_amdgcn_s_memrealtime() - r_begin;
    stamps[2 * blockIdx.x + 1] = dr ? dc * 100000ull / dr : 0;
  }
#endif
}

__global__ void init(_Float16* p, size_t n, unsigned seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed;
    x ^= x >> 13; x *= 0x5bd1e995u; x ^= x >> 15;
    p[i] = (_Float16)(((int)(x & 255) - 128) / 256.0f);
  }
}

static size_t cm(size_t row, size_t k, size_t K) { return ((row >> 5) * (K >> 3) + (k >> 3)) * 256 + (row & 31) * 8 + (k & 7); }

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 65536, N = argc > 2 ? atoi(argv[2]) : 3072, K = argc > 3 ? atoi(argv[3]) : 768;
  _Float16 *A, *W, *C;
  unsigned long long* st;
  hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&W, (size_t)N * K * 2); hipMalloc(&C, (size_t)M * N * 2);
  const int tiles = (M / 256) * (N / 256);
  int cus = 256;
  hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, 0) == hipSuccess) cus = prop.multiProcessorCount;
  const int grid = PERSIST ? (tiles < cus ? tiles : cus) : tiles;
  hipMalloc(&st, (size_t)grid * 16);
  init<<<1024, 256>>>(A, (size_t)M * K, 1); init<<<1024, 256>>>(W, (size_t)N * K, 2);
  const int smem = kSlots * kSlot;
  hipFuncSetAttribute(reinterpret_cast<const void*>(gemm4w4), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) gemm4w4<<<grid, 256, smem>>>(A, W, C, M, N, K, st, tiles);
  hipEventRecord(e0);
  const int reps = 10;
  for (int it = 0; it < reps; ++it) gemm4w4<<<grid, 256, smem>>>(A, W, C, M, N, K, st, tiles);
  hipEventRecord(e1); hipEventSynchronize(e1);
  if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); return 1; }
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  std::vector<unsigned long long> h(2 * grid);
  hipMemcpy(h.data(), st, (size_t)grid * 16, hipMemcpyDeviceToHost);
  double kl = 0, khz = 0;
  for (int i = 0; i < grid; ++i) { kl += h[2 * i]; khz += h[2 * i + 1]; }
  khz /= grid;
  // spot check against the chunk-major operands
  std::vector<_Float16> ha((size_t)M * K > (1u << 26) ? 0 : 0);
  double maxerr = 0;
  for (int t = 0; t < 16; ++t) {
    const int r = (int)(((long long)t * 7919 + 13) % M), cc = (int)(((long long)t * 104729 + 7) % N);
    double s = 0;
    for (int k = 0; k < K; ++k) {
      _Float16 a, w;
      hipMemcpy(&a, A + cm(r, k, K), 2, hipMemcpyDeviceToHost);
      hipMemcpy(&w, W + cm(cc, k, K), 2, hipMemcpyDeviceToHost);
      s += (double)(float)a * (float)w;
    }
    _Float16 c;
    hipMemcpy(&c, C + (size_t)r * N + cc, 2, hipMemcpyDeviceToHost);
    const double e = fabs(s - (double)(float)c) / (fabs(s) + 1e-2); if (e > maxerr) maxerr = e;
  }
  printf("32-k slots: M=%d N=%d K=%d grid=%d: %.1f us  %.1f TFLOP/s  | K loop %.0f cycles per K step (64 k) at %.0f MHz = %.2f us | spot err %.2e\n", M, N, K, grid,
         ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12, kl / tiles / (K / 64), khz / 1e3, kl / tiles / (K / 64) / (khz / 1e3), maxerr);
  return 0;
}
