// bf16 MFMA GEMM for the BERT passage encoder on gfx950:  C[M,N] = A[M,K] · W[N,K]^T (+ fused epilogue)
//
// Both operands are K-contiguous (activations row-major, nn.Linear weights [out,in] row-major), so an
// MFMA fragment of either is one 16-byte load per lane.  v_mfma_f32_32x32x16_bf16, fp32 accumulate.
//
// Block tile BM x BN (activation rows x weight rows), BK = 64; WAVES_M x WAVES_N waves, each owning
// (BM/WAVES_M) x (BN/WAVES_N) as 32x32 MFMA tiles.  The weight fragment is the MFMA "A" operand and
// the activation fragment the "B" operand ("swapped" GEMM): D[i = n][j = m], so a lane ends up with
// 4 *consecutive n* for one m -> 8-byte (bf16) / 16-byte (fp32) pieces of a C row.  Blocks that produce
// V for attention exchange the operand roles (same registers) and get 4 consecutive m for one n, i.e.
// pieces of a V^T row, for free.
//
// Staging: global_load_lds (16 B/lane, 1 KiB per wave-instruction) straight into a double-buffered LDS
// image [rows][64 bf16]; bank conflicts are removed by permuting the 16-byte chunks of each 128-byte
// row on the SOURCE side (chunk c of row r lands in slot c ^ ((r>>1)&7)) and applying the same XOR on
// the ds_read_b128 side (guide rule: linear destination + swizzled source + swizzled read).
// One barrier per K step: loads of step k+1 are issued before the MFMAs of step k.
//
// Epilogue: accumulators (+bias, x0.125 for Q, GELU, ...) are staged through the now idle LDS as a
// row-major tile and written out with 16-byte stores, whole 512-byte row segments per wave-instruction
// (an MFMA-layout store would issue 2-8 byte pieces scattered over 32 rows).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace capamd {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

enum GemmEpilogue {
  kEpiBiasBf16 = 0,      // out_bf16[m][n] = acc + bias[n]
  kEpiBiasGeluBf16 = 1,  // out_bf16[m][n] = gelu(acc + bias[n])                      (erf GELU)
  kEpiBiasResidF32 = 2,  // out_f32[m][n]  = acc + bias[n] + resid[m][n]              (pre-LayerNorm sum)
  kEpiQkv = 3,           // n < H: Q[m][n] = (acc+bias)/8 ; n < 2H: K[m][n-H] ; else V^T[(psg,head)][d][key]
};

struct GemmArgs {
  const __bf16* A;      // [M, K] activations
  const __bf16* W;      // [N, K] weights
  const float* bias;    // [N]
  int M, N, K;
  __bf16* out_bf16;     // kEpiBias*: [M, N];  kEpiQkv: Q [M, H]
  __bf16* out_k;        // kEpiQkv: K [M, H]
  __bf16* out_vt;       // kEpiQkv: V^T [M/S * heads][64][S]
  const float* resid;   // kEpiBiasResidF32: [M, N]
  float* out_f32;       // kEpiBiasResidF32: [M, N]
  int H, S, heads;      // kEpiQkv geometry (head_dim = 64)
};

__device__ __forceinline__ float gelu_erf(float x) {
  // 0.5 x (1 + erf(x / sqrt 2)); erf by Abramowitz-Stegun 7.1.26 (|abs err| < 1.5e-7, far below the
  // bf16 rounding applied to the result)
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = 1.f / (1.f + 0.3275911f * z);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float e = 1.f - poly * __expf(-z * z);
  return 0.5f * x * (1.f + (x < 0.f ? -e : e));
}

__device__ __forceinline__ int swz_chunk(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

template <int BM, int BN, int WAVES_M, int WAVES_N, int EPI>
struct GemmKernel {
  static constexpr int kWaves = WAVES_M * WAVES_N;
  static constexpr int kThreads = 64 * kWaves;
  static constexpr int BK = 64;
  static constexpr int WMT = BM / WAVES_M, WNT = BN / WAVES_N;  // per-wave extent in m and n
  static constexpr int TM = WMT / 32, TN = WNT / 32;            // 32x32 MFMA tiles per wave
  static constexpr int kStageBytes = (BM + BN) * BK * 2;        // one K step of both operands
  static constexpr int kEpiElem = (EPI == kEpiBiasResidF32) ? 4 : 2;
  // epilogue staging: rows of BN (or BM when transposed; BM == BN required for kEpiQkv) elements + 16 B pad
  static constexpr int kEpiRowBytes = BN * kEpiElem + 16;
  static constexpr int kEpiPasses = (EPI == kEpiBiasResidF32 && BM * kEpiRowBytes > 140000) ? 2 : 1;
  static constexpr int kEpiBytes = (BM / kEpiPasses) * kEpiRowBytes;
  static constexpr int kLdsBytes = (2 * kStageBytes > kEpiBytes) ? 2 * kStageBytes : kEpiBytes;
  static_assert(WMT % 32 == 0 && WNT % 32 == 0, "wave tile must be a multiple of 32x32");
  static_assert(EPI != kEpiQkv || BM == BN, "QKV epilogue stages a transposed tile: needs BM == BN");
  static_assert((BM * 8) % (64 * kWaves) == 0 && (BN * 8) % (64 * kWaves) == 0, "stage loop must divide evenly");

  // issue the global->LDS copies of K step `kt` into stage buffer `buf`
  static __device__ __forceinline__ void stage(const GemmArgs& a, char* lds, int buf, int kt, int m0, int n0, int wave, int lane) {
    char* base = lds + buf * kStageBytes;
    const int r8 = lane >> 3, p = lane & 7;
    constexpr int A_INSTR = BM * 8 / 64 / kWaves;  // wave-instructions per wave for the activation tile
    constexpr int W_INSTR = BN * 8 / 64 / kWaves;
#pragma unroll
    for (int t = 0; t < A_INSTR; ++t) {
      const int row = (wave * A_INSTR + t) * 8 + r8;
      const __bf16* src = a.A + (int64_t)(m0 + row) * a.K + kt * BK + swz_chunk(row, p) * 8;
      __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(base + (wave * A_INSTR + t) * 1024), 16, 0, 0);
    }
    char* wbase = base + BM * BK * 2;
#pragma unroll
    for (int t = 0; t < W_INSTR; ++t) {
      const int row = (wave * W_INSTR + t) * 8 + r8;
      const __bf16* src = a.W + (int64_t)(n0 + row) * a.K + kt * BK + swz_chunk(row, p) * 8;
      __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(wbase + (wave * W_INSTR + t) * 1024), 16, 0, 0);
    }
  }

  static __device__ __forceinline__ bf16x8 frag(const char* tile, int row, int kslice, int half) {
    return *reinterpret_cast<const bf16x8*>(tile + row * 128 + swz_chunk(row, 2 * kslice + half) * 16);
  }

  template <bool TRANS>
  static __device__ __forceinline__ void run(const GemmArgs& a, char* lds, int m0, int n0) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WAVES_M, wn = wave / WAVES_M;
    const int l31 = lane & 31, half = lane >> 5;
    f32x16 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int KT = a.K / BK;
    stage(a, lds, 0, 0, m0, n0, wave, lane);
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
      if (kt + 1 < KT) stage(a, lds, (kt + 1) & 1, kt + 1, m0, n0, wave, lane);
      const char* at = lds + (kt & 1) * kStageBytes;
      const char* wt = at + BM * BK * 2;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf16x8 fa[TM], fw[TN];
#pragma unroll
        for (int j = 0; j < TM; ++j) fa[j] = frag(at, wm * WMT + j * 32 + l31, ks, half);
#pragma unroll
        for (int i = 0; i < TN; ++i) fw[i] = frag(wt, wn * WNT + i * 32 + l31, ks, half);
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int j = 0; j < TM; ++j)
            acc[i][j] = TRANS ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[j], fw[i], acc[i][j], 0, 0, 0)
                              : __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[i], fa[j], acc[i][j], 0, 0, 0);
      }
      __syncthreads();  // (drains the in-flight global_load_lds of step kt+1 as well)
    }

    // ---------------- epilogue: registers -> LDS tile -> 16-byte global stores ----------------
    // !TRANS: lane holds m = mrow(j) = wm*WMT + j*32 + l31 and n = wn*WNT + i*32 + 8*(r>>2) + 4*half + (r&3)
    //  TRANS: lane holds n = wn*WNT + i*32 + l31       and m = wm*WMT + j*32 + 8*(r>>2) + 4*half + (r&3)
    // staged tile rows = m (!TRANS) or n (TRANS), BN (== BM when TRANS) elements per row.
    constexpr int ROWS = BM / kEpiPasses;
#pragma unroll
    for (int pass = 0; pass < kEpiPasses; ++pass) {
      if (pass) __syncthreads();
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          const int row_f = TRANS ? wn * WNT + i * 32 + l31 : wm * WMT + j * 32 + l31;  // fixed index of this lane
          if (row_f / ROWS != pass) continue;
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const int col = (TRANS ? wm * WMT + j * 32 : wn * WNT + i * 32) + 8 * g4 + 4 * half;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int n = TRANS ? row_f : col + e;
              float x = acc[i][j][g4 * 4 + e] + a.bias[n0 + n];
              if (EPI == kEpiBiasGeluBf16) x = gelu_erf(x);
              if (EPI == kEpiQkv && n0 < a.H) x *= 0.125f;  // 1/sqrt(head_dim = 64) folded into Q (exact in bf16)
              v[e] = x;
            }
            char* dst = lds + (row_f - pass * ROWS) * kEpiRowBytes + col * kEpiElem;
            if (kEpiElem == 4) {
              *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
              bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
              *reinterpret_cast<bf16x4*>(dst) = o;
            }
          }
        }
      __syncthreads();
      constexpr int CH_PER_ROW = BN * kEpiElem / 16;
      for (int c = tid; c < ROWS * CH_PER_ROW; c += kThreads) {
        const int row = c / CH_PER_ROW, ch = c % CH_PER_ROW;
        const char* src = lds + row * kEpiRowBytes + ch * 16;
        const int grow = pass * ROWS + row;
        if (EPI == kEpiBiasResidF32) {
          const int64_t off = (int64_t)(m0 + grow) * a.N + n0 + ch * 4;
          float4 x = *reinterpret_cast<const float4*>(src);
          const float4 rsd = *reinterpret_cast<const float4*>(a.resid + off);
          x.x += rsd.x; x.y += rsd.y; x.z += rsd.z; x.w += rsd.w;
          *reinterpret_cast<float4*>(a.out_f32 + off) = x;
        } else if (EPI == kEpiQkv) {
          const uint4 x = *reinterpret_cast<const uint4*>(src);
          if (!TRANS) {
            __bf16* dst = (n0 < a.H) ? a.out_bf16 + (int64_t)(m0 + grow) * a.H + n0 + ch * 8
                                     : a.out_k + (int64_t)(m0 + grow) * a.H + (n0 - a.H) + ch * 8;
            *reinterpret_cast<uint4*>(dst) = x;
          } else {
            const int nv = n0 + grow - 2 * a.H, head = nv >> 6, d = nv & 63;
            const int mg = m0 + ch * 8, psg = mg / a.S, key = mg % a.S;
            *reinterpret_cast<uint4*>(a.out_vt + ((int64_t)(psg * a.heads + head) * 64 + d) * a.S + key) = x;
          }
        } else {
          *reinterpret_cast<uint4*>(a.out_bf16 + (int64_t)(m0 + grow) * a.N + n0 + ch * 8) = *reinterpret_cast<const uint4*>(src);
        }
      }
    }
  }
};

// grid: x = N/BN (fast, so that consecutive blocks share the activation panel), y = M/BM
template <int BM, int BN, int WAVES_M, int WAVES_N, int EPI>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void gemm_bf16_kernel(GemmArgs a) {
  using G = GemmKernel<BM, BN, WAVES_M, WAVES_N, EPI>;
  extern __shared__ __attribute__((aligned(16))) char gemm_lds[];
  const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
  if (EPI == kEpiQkv && n0 >= 2 * a.H)
    G::template run<true>(a, gemm_lds, m0, n0);
  else
    G::template run<false>(a, gemm_lds, m0, n0);
}

}  // namespace capamd
