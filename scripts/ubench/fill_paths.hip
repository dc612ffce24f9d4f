// Micro-benchmark (profiling only, not part of the library): per-CU rate of the ways a workgroup can pull an
// L2-resident operand panel -- LDS-DMA (global_load_lds), plain global loads to VGPRs, buffer loads to VGPRs, and
// VGPR loads followed by ds_write_b128 -- in the access shape of the GEMM fill (8 rows x 128 B per wave instruction).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/fill_paths scripts/ubench/fill_paths.hip && /tmp/fill_paths
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int kRows = 1024, kRowBytes = 1536, kIters = 64;   // panel: 1024 rows x 768 bf16 (1.5 MiB), walked in 64-k steps

template <int MODE>
__global__ __launch_bounds__(512) void fill_kernel(const char* __restrict__ panel, unsigned* sink, unsigned long long* cycles, int reps) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r8 = lane >> 3, p = lane & 7;
  u32x4 accv = {0, 0, 0, 0};
  // buffer resource over the panel: base, stride 0, num_records = bytes, raw dword format
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(panel), 0, kRows * kRowBytes, 0x00020000);
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int rep = 0; rep < reps; ++rep) {
    for (int kt = 0; kt < 12; ++kt) {
      // one "K step": 512 rows x 128 B = 64 KiB per workgroup = 8 instructions per wave
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int row = ((wave * 8 + t) * 8 + r8 + rep * 37) & (kRows - 1);
        const char* src = panel + (size_t)row * kRowBytes + kt * 128 + p * 16;
        if (MODE == 0) {
          __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(lds + ((kt & 1) * 64 + wave * 8 + t) * 1024), 16, 0, 0);
        } else if (MODE == 3) {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)(lds + ((kt & 1) * 64 + wave * 8 + t) * 1024), 16,
                                               (int)((size_t)row * kRowBytes + p * 16), kt * 128, 0, 0);
        } else if (MODE == 1) {
          const u32x4 v = *reinterpret_cast<const u32x4*>(src);
          accv ^= v;
        } else if (MODE == 2) {
          const u32x4 v = *reinterpret_cast<const u32x4*>(src);
          *reinterpret_cast<u32x4*>(lds + ((kt & 1) * 64 + wave * 8 + t) * 1024 + lane * 16) = v;
        }
      }
      if (MODE == 0 || MODE == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (MODE != 1) accv[0] ^= *reinterpret_cast<const unsigned*>(lds + tid * 4);
  if (accv[0] == 0x12345u) sink[0] = accv[1] ^ accv[2] ^ accv[3];
  if (tid == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, const char* panel, unsigned* sink, unsigned long long* cyc, int blocks, int threads = 512) {
  const int reps = 40;
  hipFuncSetAttribute(reinterpret_cast<const void*>(fill_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipLaunchKernelGGL(fill_kernel<MODE>, dim3(blocks), dim3(threads), 131072, 0, panel, sink, cyc, 2);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(fill_kernel<MODE>, dim3(blocks), dim3(threads), 131072, 0, panel, sink, cyc, reps);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
  double mean = 0; for (auto c : h) mean += c; mean /= blocks;
  const double bytes = (double)reps * 12 * 65536 * threads / 512;
  printf("%-34s blocks %3d  %6.1f B/clk/CU   (%.0f cycles per 8 instr/wave; %.1f GB/s/CU, %.2f TB/s total)\n", name, blocks, bytes / mean,
         mean / (reps * 12), bytes / (ms * 1e-3) / 1e9, bytes * blocks / (ms * 1e-3) / 1e12);
}

int main() {
  char* panel; unsigned* sink; unsigned long long* cyc;
  hipMalloc(&panel, (size_t)kRows * kRowBytes); hipMemset(panel, 1, (size_t)kRows * kRowBytes);
  hipMalloc(&sink, 64); hipMalloc(&cyc, 256 * 8);
  for (int blocks : {8, 256}) {
    run<0>("global_load_lds dwordx4, 8 waves", panel, sink, cyc, blocks);
    run<3>("buffer_load_lds dwordx4, 8 waves", panel, sink, cyc, blocks);
    run<0>("global_load_lds dwordx4, 4 waves", panel, sink, cyc, blocks, 256);
    run<3>("buffer_load_lds dwordx4, 4 waves", panel, sink, cyc, blocks, 256);
    run<0>("global_load_lds dwordx4, 1 wave", panel, sink, cyc, blocks, 64);
    run<3>("buffer_load_lds dwordx4, 1 wave", panel, sink, cyc, blocks, 64);
    run<1>("global_load_dwordx4 -> VGPR, 8 waves", panel, sink, cyc, blocks);
  }
  return 0;
}
