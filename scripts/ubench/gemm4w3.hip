// Experiment (not part of the product path): gemm4w2's K loop with the A operand taken STRAIGHT FROM GLOBAL MEMORY INTO REGISTERS - in the
// chunk-major layout a 32-row x 16-k fragment is 1 KiB contiguous, i.e. exactly one buffer_load_dwordx4 per wave - and only the B (weight)
// operand staged through the LDS ring.  Per k-slice and CU the LDS then moves 24 KiB (16 read + 8 written) instead of 48; the price is
// that the two waves of a row pair fetch the same A fragments (the second one from L1 / L2).  Question: does the K loop get faster /
// does the clock rise (the ring kernel is power limited)?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/ubench/gemm4w3.hip -o scripts/ubench/gemm4w3 && scripts/ubench/gemm4w3
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include <vector>

typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void lds_void_t;

constexpr int kSlot = 8 * 1024;     // one k-slice (16 k) of the B panel: 8 row-blocks x 1 KiB
constexpr int kSlots = 8;
constexpr int PF = 6;               // A fragments are requested this many k-slices ahead (4 registers x 4 fragments each)

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__global__ __launch_bounds__(256, 1) void gemm4w3(const _Float16* __restrict__ A, const _Float16* __restrict__ W, _Float16* __restrict__ C, int M,
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
  const int b_base = wn * 4 * 1024 + half * 512 + l31 * 16;
  const int step = (int)gridDim.x;
  int tile = (gridDim.x % 8 == 0) ? (int)((blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8) : (int)blockIdx.x;
  // scalar byte offsets of k-slice 0: the 4 A row-blocks this wave multiplies (wm * 4 + j), the 2 B row-blocks it stages (wave * 2 + p)
  auto bases = [&](int t, unsigned (&sa)[4], unsigned (&sb)[2]) {
    if (t >= ntiles) t = 0;
    const int tm = (t / tn) * 256, tn0 = (t % tn) * 256;
#pragma unroll
    for (int j = 0; j < 4; ++j) sa[j] = (unsigned)(((tm >> 5) + wm * 4 + j) * (K >> 3)) * 512u;
#pragma unroll
    for (int p = 0; p < 2; ++p) sb[p] = (unsigned)(((tn0 >> 5) + wave * 2 + p) * (K >> 3)) * 512u;
  };
  auto issue_b = [&](const unsigned (&sb)[2], int s, int slot_off) {
#pragma unroll
    for (int p = 0; p < 2; ++p)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void_t*)(lds + slot_off + (wave * 2 + p) * 1024), 16, voff, (int)(sb[p] + (unsigned)s * 1024u), 0, 0);
  };
  auto load_a = [&](const unsigned (&sa)[4], int s, h8 (&dst)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      dst[j] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(ra, voff, (int)(sa[j] + (unsigned)s * 1024u), 0));
  };
  unsigned ca[4], cb[2], na[4], nb[2];
  bases(tile, ca, cb);
  h8 fa[PF][4], fb[2][4];
#pragma unroll
  for (int i = 0; i < PF; ++i) load_a(ca, i, fa[i]);
#pragma unroll 1
  for (int i = 0; i < kSlots; ++i) issue_b(cb, i, i * kSlot);
  unsigned long long t0 = 0, kacc = 0;
  const unsigned long long c_begin = __builtin_readcyclecounter(), r_begin = __builtin_amdgcn_s_memrealtime();
  wait_vmcnt<14>();      // B slice 0 of every wave (7 younger slices x 2 stay in flight)
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int i = 0; i < 4; ++i) fb[0][i] = *reinterpret_cast<const h8*>(lds + b_base + i * 1024);
  int slot_off = 0;
  for (; tile < ntiles; tile += step) {
    const int m0 = (tile / tn) * 256, n0 = (tile % tn) * 256;
    bases(tile + step, na, nb);
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x16{0};
    if (stamps && tid == 0) t0 = __builtin_readcyclecounter();
    // k-slice s (register slot R = s % PF of the A ring, B fragments fb[s & 1]): A(s) and B(s + 1) are the oldest requests in flight;
    // everything issued after A(s) may stay outstanding: B(s - PF + 8) x 2 and (4 + 2) per slice since = 2 + 6 (PF - 1)
    auto kslice = [&](auto r_c, int s) {
      constexpr int R = decltype(r_c)::value, cur = R & 1, nxt = cur ^ 1;
      wait_vmcnt<2 + 6 * (PF - 1)>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const char* stn = lds + ((slot_off + kSlot) & (kSlots * kSlot - 1));
#pragma unroll
      for (int i = 0; i < 4; ++i) fb[nxt][i] = *reinterpret_cast<const h8*>(stn + b_base + i * 1024);
      // the MFMAs of this slice consume fa[R]; its registers are then refilled with slice s + PF (this tile's, or the next tile's first ones)
      h8 a_now[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) a_now[j] = fa[R][j];
      const bool own_a = s + PF < S, own_b = s + 8 < S;
      {
        unsigned sa[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) sa[j] = own_a ? ca[j] : na[j];
        load_a(sa, own_a ? s + PF : s + PF - S, fa[R]);
        unsigned sb[2] = {own_b ? cb[0] : nb[0], own_b ? cb[1] : nb[1]};
        issue_b(sb, own_b ? s + 8 : s + 8 - S, slot_off);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[cur][i], a_now[j], acc[i][j], 0, 0, 0);
      // 16 MFMA | 4 DS read | 6 VMEM
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      }
      slot_off = (slot_off + kSlot) & (kSlots * kSlot - 1);
    };
#pragma unroll 1
    for (int s = 0; s < S; s += PF) {     // (S % PF == 0: 48, 192)
      kslice(std::integral_constant<int, 0>{}, s);
      kslice(std::integral_constant<int, 1>{}, s + 1);
      kslice(std::integral_constant<int, 2>{}, s + 2);
      kslice(std::integral_constant<int, 3>{}, s + 3);
      kslice(std::integral_constant<int, 4>{}, s + 4);
      kslice(std::integral_constant<int, 5>{}, s + 5);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) ca[j] = na[j];
    cb[0] = nb[0]; cb[1] = nb[1];
    if (stamps && tid == 0) kacc += __builtin_readcyclecounter() - t0;
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
  if ((K / 16) % PF) { printf("K / 16 must be a multiple of %d\n", PF); return 1; }
  _Float16 *A, *W, *C;
  unsigned long long* st;
  hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&W, (size_t)N * K * 2); hipMalloc(&C, (size_t)M * N * 2);
  const int tiles = (M / 256) * (N / 256);
  int cus = 256;
  hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, 0) == hipSuccess) cus = prop.multiProcessorCount;
  const int grid = tiles < cus ? tiles : cus;
  hipMalloc(&st, (size_t)grid * 16);
  init<<<1024, 256>>>(A, (size_t)M * K, 1); init<<<1024, 256>>>(W, (size_t)N * K, 2);
  const int smem = kSlots * kSlot;
  hipFuncSetAttribute(reinterpret_cast<const void*>(gemm4w3), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) gemm4w3<<<grid, 256, smem>>>(A, W, C, M, N, K, st, tiles);
  hipEventRecord(e0);
  const int reps = 10;
  for (int it = 0; it < reps; ++it) gemm4w3<<<grid, 256, smem>>>(A, W, C, M, N, K, st, tiles);
  hipEventRecord(e1); hipEventSynchronize(e1);
  if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); return 1; }
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  std::vector<unsigned long long> h(2 * grid);
  hipMemcpy(h.data(), st, (size_t)grid * 16, hipMemcpyDeviceToHost);
  double kl = 0, khz = 0;
  for (int i = 0; i < grid; ++i) { kl += h[2 * i]; khz += h[2 * i + 1]; }
  khz /= grid;
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
  printf("A direct: M=%d N=%d K=%d grid=%d: %.1f us  %.1f TFLOP/s  | K loop %.0f cycles per K step (64 k) at %.0f MHz = %.2f us | spot err %.2e\n", M, N, K, grid,
         ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12, kl / tiles / (K / 64), khz / 1e3, kl / tiles / (K / 64) / (khz / 1e3), maxerr);
  return 0;
}
