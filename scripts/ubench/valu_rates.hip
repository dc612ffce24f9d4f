// Micro-benchmarks behind bench.py's per-pass peaks of the whole-list route - MI355X (gfx950):
//   1  v_fma_f32 / v_pk_fma_f32 / v_exp_f32 issue rates, every SIMD busy (independent chains, no memory)
//   2  the RBF kernel evaluation of the KNRM pooling pass as a VALU-only loop: K_k(s) = 2^(c_k (s - mu_k)^2) summed, in the
//      5-instruction form (sub, mul, mul, exp, add) and the 4-instruction form (fma, mul-neg, exp, add)   -> KERNEL_EVAL_PEAK_G
//   3  v_mfma_f32_4x4x1_16b_f32: cycles per instruction, the operand / result lane map the sims pass relies on, and that its
//      k-chain is bit for bit an fmaf chain (MI355X_MICROARCH.md: "exact f32 (== fmaf chain, bitwise)")
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip ; run: ./valu_rates > profiles/r04/valu_rates.txt
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kIters = 4096;

template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(float* out, float seed) {
  float a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = seed + 0.001f * (float)(threadIdx.x + i);
  const float m = 0.999f + seed * 1e-6f, b = 1e-3f;
  for (int it = 0; it < kIters; ++it) {
    if (MODE == 0) {        // 8 independent v_fma_f32
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = __builtin_fmaf(a[i], m, b);
    } else if (MODE == 1) { // 4 independent v_pk_fma_f32 (8 fmas)
#pragma unroll
      for (int i = 0; i < 8; i += 2) {
        f32x2 v = {a[i], a[i + 1]};
        v = __builtin_elementwise_fma(v, (f32x2){m, m}, (f32x2){b, b});
        a[i] = v.x; a[i + 1] = v.y;
      }
    } else if (MODE == 2) { // 8 independent v_exp_f32
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = __builtin_amdgcn_exp2f(a[i]);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(a[i]));
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// the pooling pass's evaluation: 11 kernels per similarity
template <int FORM>
__global__ __launch_bounds__(256) void eval_kernel(float* out, const float* consts, float seed) {
  float mu[11], ck[11], A[11], B[11], acc[11];
#pragma unroll
  for (int k = 0; k < 11; ++k) {
    mu[k] = consts[k]; ck[k] = consts[16 + k]; A[k] = consts[32 + k]; B[k] = consts[48 + k];
    acc[k] = 0.f;
  }
  float s = seed + 1e-4f * (float)threadIdx.x;
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int k = 0; k < 11; ++k) {
      if (FORM == 0) {
        const float adj = s - mu[k];
        acc[k] += __builtin_amdgcn_exp2f(adj * adj * ck[k]);
      } else {
        const float t = __builtin_fmaf(s, A[k], B[k]);
        acc[k] += __builtin_amdgcn_exp2f(-t * t);
      }
    }
    s += 1e-5f;
    asm volatile("" : "+v"(s));
  }
  float r = 0.f;
#pragma unroll
  for (int k = 0; k < 11; ++k) r += acc[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

__global__ __launch_bounds__(256) void mfma_rate_kernel(float* out, float seed) {
  f32x4 c[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) c[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = seed + 0.001f * (float)threadIdx.x, b = 1.f - seed;
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += c[i].x + c[i].y + c[i].z + c[i].w;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// one wave: K steps of D_b[i][j] += A_b[i] * B_b[j]; a[l][k], b[l][k] per lane and step; d[l][4]
__global__ void mfma_map_kernel(const float* a, const float* b, float* d, int K) {
  const int l = threadIdx.x;
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < K; ++k) c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l * K + k], b[l * K + k], c, 0, 0, 0);
  d[l * 4 + 0] = c.x; d[l * 4 + 1] = c.y; d[l * 4 + 2] = c.z; d[l * 4 + 3] = c.w;
}

template <class F>
static double time_ms(F launch, int reps = 5) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) launch();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main() {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  printf("device %s, %d CUs, clock %d kHz\n", p.name, cus, p.clockRate);
  float* out;
  CK(hipMalloc(&out, sizeof(float) * 256 * 4096));
  for (int wpc : {4, 8, 16}) {          // waves per CU = blocks per CU * 4
    const int blocks = cus * (wpc / 4);
    const double lanes = (double)blocks * 256;
    double ms;
    ms = time_ms([&] { hipLaunchKernelGGL(rate_kernel<0>, dim3(blocks), dim3(256), 0, 0, out, 0.5f); });
    printf("waves/CU %2d  v_fma_f32      %8.1f G fma/s  = %6.1f TFLOP/s\n", wpc, lanes * kIters * 8 / ms / 1e6, 2 * lanes * kIters * 8 / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(rate_kernel<1>, dim3(blocks), dim3(256), 0, 0, out, 0.5f); });
    printf("waves/CU %2d  v_pk_fma_f32   %8.1f G fma/s  = %6.1f TFLOP/s\n", wpc, lanes * kIters * 8 / ms / 1e6, 2 * lanes * kIters * 8 / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(rate_kernel<2>, dim3(blocks), dim3(256), 0, 0, out, 0.5f); });
    printf("waves/CU %2d  v_exp_f32      %8.1f G exp/s\n", wpc, lanes * kIters * 8 / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL(mfma_rate_kernel, dim3(blocks), dim3(256), 0, 0, out, 0.5f); });
    printf("waves/CU %2d  v_mfma_f32_4x4x1_16b  %8.2f cycles/instr/SIMD at 2.4 GHz nominal (%6.1f TFLOP/s)\n", wpc,
           ms * 1e-3 * 2.4e9 / ((double)kIters * 4 * (wpc / 4.0)), 2.0 * 256 * (double)blocks * 4 * kIters * 4 / ms / 1e9);
  }
  // RBF evaluation loops
  float hc[64] = {0};
  for (int k = 0; k < 11; ++k) {
    const float mu = k < 10 ? -0.9f + 0.2f * k : 1.0f, sg = k < 10 ? 0.1f : 0.001f;
    const float c = (-0.5f * 1.4426950408889634f) / (sg * sg);
    hc[k] = mu; hc[16 + k] = c; hc[32 + k] = sqrtf(-c); hc[48 + k] = -sqrtf(-c) * mu;
  }
  float* dc;
  CK(hipMalloc(&dc, sizeof(hc)));
  CK(hipMemcpy(dc, hc, sizeof(hc), hipMemcpyHostToDevice));
  for (int wpc : {4, 8, 16, 32}) {
    const int blocks = cus * (wpc / 4);
    const double evals = (double)blocks * 256 * kIters * 11;
    double ms = time_ms([&] { hipLaunchKernelGGL(eval_kernel<0>, dim3(blocks), dim3(256), 0, 0, out, dc, 0.1f); });
    printf("waves/CU %2d  kernel evaluation, 5-instruction form (sub mul mul exp add)  %8.1f G evaluations/s\n", wpc, evals / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL(eval_kernel<1>, dim3(blocks), dim3(256), 0, 0, out, dc, 0.1f); });
    printf("waves/CU %2d  kernel evaluation, 4-instruction form (fma mul exp add)      %8.1f G evaluations/s\n", wpc, evals / ms / 1e6);
  }
  // MFMA 4x4x1 lane map + exactness
  {
    const int K = 320;
    std::vector<float> a(64 * K), b(64 * K), d(256);
    srand(7);
    for (auto& v : a) v = (float)rand() / RAND_MAX - 0.5f;
    for (auto& v : b) v = (float)rand() / RAND_MAX - 0.5f;
    float *da, *db, *dd;
    CK(hipMalloc(&da, a.size() * 4)); CK(hipMalloc(&db, b.size() * 4)); CK(hipMalloc(&dd, 1024));
    CK(hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(mfma_map_kernel, dim3(1), dim3(64), 0, 0, da, db, dd, K);
    CK(hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost));
    // hypothesis: lane l = 4 * blk + j holds D_blk[i][j] in register i, with A_blk[i] from lane 4 * blk + i and B_blk[j] from lane 4 * blk + j
    int bad = 0;
    for (int l = 0; l < 64; ++l)
      for (int i = 0; i < 4; ++i) {
        const int blk = l >> 2, j = l & 3;
        float c = 0.f;
        for (int k = 0; k < K; ++k) c = fmaf(a[(4 * blk + i) * K + k], b[(4 * blk + j) * K + k], c);
        if (memcmp(&c, &d[l * 4 + i], 4) != 0) ++bad;
      }
    printf("v_mfma_f32_4x4x1_16b lane map (lane 4b+j, reg i) = sum_k A[lane 4b+i][k] B[lane 4b+j][k] as a k-ordered fmaf chain over %d steps: %s (%d of 256 differ)\n",
           K, bad ? "NOT bit-identical" : "bit-identical", bad);
    if (bad) {   // the transposed hypothesis
      int bad2 = 0;
      for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 4; ++i) {
          const int blk = l >> 2, j = l & 3;
          float c = 0.f;
          for (int k = 0; k < K; ++k) c = fmaf(a[(4 * blk + j) * K + k], b[(4 * blk + i) * K + k], c);
          if (memcmp(&c, &d[l * 4 + i], 4) != 0) ++bad2;
        }
      printf("transposed hypothesis (A index = lane, B index = register): %d of 256 differ\n", bad2);
    }
  }
  return 0;
}
