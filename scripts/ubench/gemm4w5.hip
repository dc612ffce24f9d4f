// gemm4w5: gemm4w5's ring consumed by v_mfma_f32_16x16x32_f16 (k = 32 per MFMA: two ring slots per step, one barrier per 64 MFMAs; per
// flop half the accumulator-register traffic of 32x32x16 and twice its operand-register traffic) - the K loop the vendor's kernel runs.
// Experiment (not part of the product path): K loop of a 4-wave 256x256 tile (one wave per SIMD, 128x128 per wave, 256 accumulator
// registers) fed by a k-slice ring: both operands chunk-major ([rows/32][K/8][32][8], so a 32-row x 16-k slice is 1 KiB contiguous in
// memory AND in LDS: one LDS-DMA instruction, conflict-free ds_read_b128 fragments, no swizzle), 8 slots of 16 KiB (two K steps), one
// barrier per 16-MFMA k-slice, slice t+8 issued while slice t is consumed.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/ubench/gemm4w5.hip -o scripts/ubench/gemm4w5 && scripts/ubench/gemm4w5
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include <vector>

typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void lds_void_t;

constexpr int kSlot = 16 * 1024;    // one k-slice (16 k): A 8 row-blocks x 1 KiB | B 8 row-blocks x 1 KiB
constexpr int kSlots = 8;
#ifndef ABL
#define ABL 0      // bit 0: no LDS-DMA in the loop, bit 1: no LDS reads in the loop, bit 2: no sched_group_barrier pattern
#endif
#ifndef PERSIST
#define PERSIST 1
#endif

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__global__ __launch_bounds__(256, 1) void gemm4w5(const _Float16* __restrict__ A, const _Float16* __restrict__ W, _Float16* __restrict__ C, int M,
                                                  int N, int K, unsigned long long* stamps, int ntiles) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int tn = N / 256;
  const int S = K / 16;                      // k-slices per tile
  const auto ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(A), 0, (int)((size_t)M * K * 2), 0x00020000);
  const auto rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(W), 0, (int)((size_t)N * K * 2), 0x00020000);
  const int voff = lane * 16;
  // fragment of 16-row tile i of a panel (rows 16 i .. 16 i + 15 of the wave's 128): lane l reads row l % 16, k-chunk q = l / 16 of the
  // step's four (chunks 0, 1 in the step's first slot, 2, 3 in its second): + (i >> 1) * 1024 + (i & 1) * 256
  const int l15 = lane & 15, q = lane >> 4;
  const int a_base = wm * 4 * 1024 + (q & 1) * 512 + l15 * 16 + (q >> 1) * kSlot;
  const int b_base = 8192 + wn * 4 * 1024 + (q & 1) * 512 + l15 * 16 + (q >> 1) * kSlot;

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
    for (int p = 0; p < 2; ++p) {
      const int rb = wave * 2 + p;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void_t*)(d + rb * 1024), 16, voff, (int)(sa[p] + (unsigned)s * 1024u), 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void_t*)(d + 8192 + rb * 1024), 16, voff, (int)(sb[p] + (unsigned)s * 1024u), 0, 0);
    }
  };
  unsigned ca[2], cb[2], na[2], nb[2];
  bases(tile, ca, cb);
#pragma unroll 1
  for (int i = 0; i < kSlots; ++i) issue(ca, cb, i, i * kSlot);       // (S >= 8)
  unsigned long long t0 = 0, kacc = 0;
  const unsigned long long c_begin = __builtin_readcyclecounter(), r_begin = __builtin_amdgcn_s_memrealtime();
  typedef __attribute__((ext_vector_type(4))) float f32x4;
  h8 fa[2][8], fb[2][8];
  // first step's fragments (later tiles find theirs loaded by the last iteration of the tile before)
  wait_vmcnt<24>();
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    fa[0][i] = *reinterpret_cast<const h8*>(lds + a_base + (i >> 1) * 1024 + (i & 1) * 256);
    fb[0][i] = *reinterpret_cast<const h8*>(lds + b_base + (i >> 1) * 1024 + (i & 1) * 256);
  }
  int slot_off = 0;   // byte offset of the first slot of the step being consumed
  for (; tile < ntiles; tile += step) {
    m0 = (tile / tn) * 256; n0 = (tile % tn) * 256;
    bases(tile + step, na, nb);
    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (stamps && tid == 0) t0 = __builtin_readcyclecounter();
    // one k-step (32 k = two slots): its fragments are in registers once lgkmcnt(0); step +1 has landed once vmcnt(16) on every wave +
    // barrier; then the fragment reads of step +1, the LDS-DMA of step +4 into the two slots every wave has just finished reading, 64 MFMAs
    auto kstep = [&](auto u_c, const unsigned (&sa)[2], const unsigned (&sb)[2], int fs) {
      constexpr int cur = decltype(u_c)::value, nxt = cur ^ 1;
      wait_vmcnt<16>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const char* stn = lds + ((slot_off + 2 * kSlot) & (kSlots * kSlot - 1));
      // issue order pinned by hand (inline-asm MFMAs accumulating in place in AGPRs: left to the register allocator the 64 four-register
      // accumulators get copied around): M M R M M | M M D M M ... 64 MFMAs, 16 fragment reads, 8 LDS-DMAs
#define MF(idx) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[(idx) >> 3][(idx) & 7]) : "v"(fb[cur][(idx) >> 3]), "v"(fa[cur][(idx) & 7]))
      auto RD = [&](int r) {
        if (ABL & 2) return;
        const int i = r & 7;
        if (r < 8) fa[nxt][i] = *reinterpret_cast<const h8*>(stn + a_base + (i >> 1) * 1024 + (i & 1) * 256);
        else fb[nxt][i] = *reinterpret_cast<const h8*>(stn + b_base + (i >> 1) * 1024 + (i & 1) * 256);
      };
      auto DM = [&](int p) {       // piece p of the step's eight: slice fs + (p >> 2), row block pair member (p >> 1) & 1, operand p & 1
        if (ABL & 1) return;
        const int sl = p >> 2, pp = (p >> 1) & 1, rb = wave * 2 + pp;
        char* d = lds + slot_off + sl * kSlot;
        if (p & 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void_t*)(d + 8192 + rb * 1024), 16, voff, (int)(sb[pp] + (unsigned)(fs + sl) * 1024u), 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void_t*)(d + rb * 1024), 16, voff, (int)(sa[pp] + (unsigned)(fs + sl) * 1024u), 0, 0);
      };
#define SB __builtin_amdgcn_sched_barrier(0)
      SB;
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        MF(4 * g); SB; MF(4 * g + 1); SB; RD(g); SB; MF(4 * g + 2); SB; MF(4 * g + 3); SB;
        if (g & 1) { DM(g >> 1); SB; }
      }
#undef SB
#undef MF
      slot_off = (slot_off + 2 * kSlot) & (kSlots * kSlot - 1);
    };
    using U0 = std::integral_constant<int, 0>;
    using U1 = std::integral_constant<int, 1>;
    // steps 0 .. S/2-5 refill with slices 8 .. S-1 of this tile, the last four with slices 0 .. 7 of the block's next tile
#pragma unroll 1
    for (int s = 0; s < S - 8; s += 4) { kstep(U0{}, ca, cb, s + 8); kstep(U1{}, ca, cb, s + 10); }
#pragma unroll 1
    for (int s = 0; s < 8; s += 4) { kstep(U0{}, na, nb, s); kstep(U1{}, na, nb, s + 2); }
    ca[0] = na[0]; ca[1] = na[1]; cb[0] = nb[0]; cb[1] = nb[1];
    if (stamps && tid == 0) kacc += __builtin_readcyclecounter() - t0;
    // plain epilogue: fp16 row-major (accumulator register r of tile (i = n tile, j = m tile): n = 4 q + r, m = l15)
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int row = m0 + wm * 128 + j * 16 + l15, col = n0 + wn * 128 + i * 16 + 4 * q;
        typedef __attribute__((ext_vector_type(4))) _Float16 h4;
        const h4 o = {(_Float16)acc[i][j][0], (_Float16)acc[i][j][1], (_Float16)acc[i][j][2], (_Float16)acc[i][j][3]};
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
  hipFuncSetAttribute(reinterpret_cast<const void*>(gemm4w5), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) gemm4w5<<<grid, 256, smem>>>(A, W, C, M, N, K, st, tiles);
  hipEventRecord(e0);
  const int reps = 10;
  for (int it = 0; it < reps; ++it) gemm4w5<<<grid, 256, smem>>>(A, W, C, M, N, K, st, tiles);
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
  printf("M=%d N=%d K=%d grid=%d: %.1f us  %.1f TFLOP/s  | K loop %.0f cycles per K step (64 k) at %.0f MHz = %.2f us | spot err %.2e\n", M, N, K, grid,
         ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12, kl / tiles / (K / 64), khz / 1e3, kl / tiles / (K / 64) / (khz / 1e3), maxerr);
  return 0;
}
