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
// Staging: global_load_lds (16 B/lane, 1 KiB per wave-instruction) straight into LDS.  Measured on MI355X the
// L2->LDS fill of one 64 KiB K step takes ~3.3k cycles when issued as one burst that must land before the next
// barrier (~20 B/clk/CU) while its MFMAs need 2.05k: the loop is fill-latency bound, so the fill is kept
// CONTINUOUSLY in flight: a K step (BK = 64) is staged as two k-halves of 32 KiB, four half regions
// [2 steps][2 halves] live in LDS, a half is re-staged as soon as every wave has read it (two barriers per K
// step), and waits are counted (`s_waitcnt vmcnt(2 bursts)`) so that three half-bursts (96 KiB) stay in flight
// across the barriers.  Region image: [rows][32 bf16] (64-byte rows); the four 16-byte chunks of a row are
// permuted on the SOURCE side (chunk c of row r lands in slot c ^ ((r>>2)&3)) and the same XOR is applied
// on the ds_read_b128 side: conflict-free for the b128 lane groups (linear DMA destination + swizzled
// source + swizzled read).  MFMA fragments are double-buffered in registers one 16-deep slice ahead.
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
  kEpiBiasResidBf16 = 4, // out_bf16[m][n] = acc + bias[n] + resid_bf16[m][n]         (pre-LayerNorm sum, bf16 stream)
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
  const __bf16* resid_bf16;  // kEpiBiasResidBf16: [M, N]
  float* out_f32;       // kEpiBiasResidF32: [M, N]
  int H, S, heads;      // kEpiQkv geometry (head_dim = 64)
  unsigned long long* dbg;  // optional per-block cycle stamps [blocks][32] (profiling builds of the benches only)
};

typedef __attribute__((ext_vector_type(2))) float f32x2;

// erf-GELU 0.5 x (1 + erf(x/sqrt2)) on two values at once (v_pk_* math; one v_rcp + one v_exp per value).
// erf by Abramowitz-Stegun 7.1.26: |abs err| < 1.5e-7 (+ ~1e-7 from the approximate rcp/exp2), far below the
// bf16 rounding (2^-9 relative) applied to the result.
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
  const f32x2 ax = {__builtin_fabsf(x.x), __builtin_fabsf(x.y)};
  const f32x2 z = ax * 0.70710678118654752f;
  const f32x2 den = z * 0.3275911f + 1.f;
  const f32x2 t = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
  f32x2 poly = t * 1.061405429f + -1.453152027f;
  poly = poly * t + 1.421413741f;
  poly = poly * t + -0.284496736f;
  poly = poly * t + 0.254829592f;
  poly = poly * t;
  const f32x2 arg = z * z * -1.4426950408889634f;
  const f32x2 ex = {__builtin_amdgcn_exp2f(arg.x), __builtin_amdgcn_exp2f(arg.y)};
  const f32x2 e = 1.f - poly * ex;                       // erf(|x|/sqrt2)
  const f32x2 hx = x * 0.5f, hax = ax * 0.5f;
  return hx + hax * e;                                   // 0.5x + 0.5|x| erf(|x|/sqrt2) == 0.5x(1 + erf(x/sqrt2))
}

__device__ __forceinline__ int swz_chunk(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }   // 128-byte rows (attention K tile)
__device__ __forceinline__ int swz_chunk4(int row, int chunk) { return chunk ^ ((row >> 2) & 3); }  // 64-byte rows (GEMM half regions)

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int EPI>
struct GemmKernel {
  static constexpr int kWaves = WAVES_M * WAVES_N;
  static constexpr int kThreads = 64 * kWaves;
  static constexpr int BK = 64;
  static constexpr int WMT = BM / WAVES_M, WNT = BN / WAVES_N;  // per-wave extent in m and n
  static constexpr int TM = WMT / 32, TN = WNT / 32;            // 32x32 MFMA tiles per wave
  static constexpr int kStageBytes = (BM + BN) * BK * 2;        // one K step of both operands
  static constexpr int kHalfBytes = kStageBytes / 2;            // one k-half (32 of the 64 k) of both operands
  static constexpr int kBurst = (BM + BN) * 64 / 1024 / kWaves; // global_load_lds instructions per wave per half-burst
  static constexpr int kEpiElem = (EPI == kEpiBiasResidF32 || EPI == kEpiBiasResidBf16) ? 4 : 2;  // staged element size
  // epilogue staging: rows of BN (or BM when transposed; BM == BN required for kEpiQkv) elements + 16 B pad
  static constexpr int kEpiRowBytes = BN * kEpiElem + 16;
  static constexpr int kEpiPasses = (kEpiElem == 4 && BM * kEpiRowBytes > 140000) ? 2 : 1;
  static constexpr int kEpiBytes = (BM / kEpiPasses) * kEpiRowBytes;
  static constexpr int kLdsBytes = (2 * kStageBytes > kEpiBytes) ? 2 * kStageBytes : kEpiBytes;
  static_assert(WMT % 32 == 0 && WNT % 32 == 0, "wave tile must be a multiple of 32x32");
  static_assert(EPI != kEpiQkv || BM == BN, "QKV epilogue stages a transposed tile: needs BM == BN");
  static_assert((BM * 4) % (64 * kWaves) == 0 && (BN * 4) % (64 * kWaves) == 0, "stage loop must divide evenly");

  // issue the global->LDS copies of k-half `h` of K step `kt` into its region (kt & 1, h)
  static __device__ __forceinline__ void stage_half(const GemmArgs& a, char* lds, int kt, int h, int m0, int n0, int wave, int lane) {
    char* base = lds + ((kt & 1) * 2 + h) * kHalfBytes;
    const int r16 = lane >> 2, p = lane & 3;
    constexpr int A_INSTR = BM * 4 / 64 / kWaves;  // wave-instructions per wave for the activation half tile (16 rows each)
    constexpr int W_INSTR = BN * 4 / 64 / kWaves;
    const int kcol = kt * BK + h * 32;
#pragma unroll
    for (int t = 0; t < A_INSTR; ++t) {
      const int row = (wave * A_INSTR + t) * 16 + r16;
      const __bf16* src = a.A + (int64_t)(m0 + row) * a.K + kcol + swz_chunk4(row, p) * 8;
      __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(base + (wave * A_INSTR + t) * 1024), 16, 0, 0);
    }
    char* wbase = base + BM * 64;
#pragma unroll
    for (int t = 0; t < W_INSTR; ++t) {
      const int row = (wave * W_INSTR + t) * 16 + r16;
      const __bf16* src = a.W + (int64_t)(n0 + row) * a.K + kcol + swz_chunk4(row, p) * 8;
      __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(wbase + (wave * W_INSTR + t) * 1024), 16, 0, 0);
    }
  }

  // fragment of 16-deep slice `s2` (0/1) of a half region: row, k = s2*16 + half*8 .. +7
  static __device__ __forceinline__ bf16x8 frag(const char* tile, int row, int s2, int half) {
    return *reinterpret_cast<const bf16x8*>(tile + row * 64 + swz_chunk4(row, 2 * s2 + half) * 16);
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
    unsigned long long* dbg = a.dbg ? a.dbg + (size_t)blockIdx.x * 32 : nullptr;
    int dbg_i = 0;
#define CAPAMD_STAMP() do { if (dbg && tid == 0 && dbg_i < 32) dbg[dbg_i++] = __builtin_readcyclecounter(); } while (0)
    CAPAMD_STAMP();
    // ---- main loop (see the header comment for the schedule) --------------------------------------------
    // slice ks of step kt lives in region (kt&1, ks>>1).  Per step, two barriers:
    //   B1 (entering slice 1): half 0 of this step has been read by everyone -> re-stage it with step kt+2;
    //                          half 1 of this step must have landed (its reads start now).
    //   B3 (entering slice 3): half 1 has been read by everyone -> re-stage; half 0 of step kt+1 must have landed.
    // Before each barrier a wave waits until the burst it is about to publish has landed; the bursts issued
    // after that one (two in steady state) stay in flight: vmcnt(2 * kBurst).
    bf16x8 fa[2][TM], fw[2][TN];
    auto load_frags = [&](int kt, int ks, int slot) {
      const char* at = lds + ((kt & 1) * 2 + (ks >> 1)) * kHalfBytes;
      const char* wt = at + BM * 64;
#pragma unroll
      for (int j = 0; j < TM; ++j) fa[slot][j] = frag(at, wm * WMT + j * 32 + l31, ks & 1, half);
#pragma unroll
      for (int i = 0; i < TN; ++i) fw[slot][i] = frag(wt, wn * WNT + i * 32 + l31, ks & 1, half);
    };
    auto publish = [&](int newer) {  // wait for the oldest outstanding burst (newer = bursts issued after it), then barrier
      if (newer >= 2) wait_vmcnt<2 * kBurst>();
      else if (newer == 1) wait_vmcnt<kBurst>();
      else wait_vmcnt<0>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // my fragment reads of the region about to be re-staged are done
      __builtin_amdgcn_s_barrier();
    };
    stage_half(a, lds, 0, 0, m0, n0, wave, lane);
    stage_half(a, lds, 0, 1, m0, n0, wave, lane);
    if (KT > 1) {
      stage_half(a, lds, 1, 0, m0, n0, wave, lane);
      stage_half(a, lds, 1, 1, m0, n0, wave, lane);
      wait_vmcnt<3 * kBurst>();
    } else {
      wait_vmcnt<kBurst>();
    }
    __builtin_amdgcn_s_barrier();
    CAPAMD_STAMP();
    load_frags(0, 0, 0);
    for (int kt = 0; kt < KT; ++kt) {
      if (kt < 13) CAPAMD_STAMP();
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int cur = ks & 1, nxt = cur ^ 1;
        if (ks == 1) {
          // need (kt, half 1); newer bursts: (kt+1, 0), (kt+1, 1) when step kt+1 exists
          publish(kt + 1 < KT ? 2 : 0);
          if (kt + 2 < KT) stage_half(a, lds, kt + 2, 0, m0, n0, wave, lane);
        } else if (ks == 3) {
          // need (kt+1, half 0); newer: (kt+1, 1), (kt+2, 0) when step kt+2 exists
          publish(kt + 2 < KT ? 2 : 1);
          if (kt + 2 < KT) stage_half(a, lds, kt + 2, 1, m0, n0, wave, lane);
        }
        if (ks < 3) load_frags(kt, ks + 1, nxt);
        else if (kt + 1 < KT) load_frags(kt + 1, 0, nxt);
        __builtin_amdgcn_sched_barrier(0);  // keep the prefetch reads ahead of this slice's MFMAs (their latency hides under them)
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int j = 0; j < TM; ++j)
            acc[i][j] = TRANS ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][j], fw[cur][i], acc[i][j], 0, 0, 0)
                              : __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[cur][i], fa[cur][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
    CAPAMD_STAMP();

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
              if (EPI == kEpiQkv && n0 < a.H) x *= 0.125f;  // 1/sqrt(head_dim = 64) folded into Q (exact in bf16)
              v[e] = x;
            }
            if (EPI == kEpiBiasGeluBf16) {
              const f32x2 g0 = gelu_erf2(f32x2{v[0], v[1]}), g1 = gelu_erf2(f32x2{v[2], v[3]});
              v[0] = g0.x; v[1] = g0.y; v[2] = g1.x; v[3] = g1.y;
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
      CAPAMD_STAMP();
      constexpr int CH_PER_ROW = BN * kEpiElem / 16;
      constexpr int ITERS = ROWS * CH_PER_ROW / kThreads;
      static_assert(ROWS * CH_PER_ROW % kThreads == 0, "write-out loop must divide evenly");
      // residual loads first, all in flight together (the compiler cannot hoist them over the stores itself:
      // resid and out may alias as far as it knows, which serialises load -> store -> load ...)
      float4 rs_f32[(EPI == kEpiBiasResidF32) ? ITERS : 1];
      bf16x4 rs_b16[(EPI == kEpiBiasResidBf16) ? ITERS : 1];
      if (EPI == kEpiBiasResidF32 || EPI == kEpiBiasResidBf16) {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
          const int c = it * kThreads + tid;
          const int64_t off = (int64_t)(m0 + pass * ROWS + c / CH_PER_ROW) * a.N + n0 + (c % CH_PER_ROW) * 4;
          if (EPI == kEpiBiasResidF32) rs_f32[it] = *reinterpret_cast<const float4*>(a.resid + off);
          else rs_b16[it] = *reinterpret_cast<const bf16x4*>(a.resid_bf16 + off);
        }
      }
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        const int c = it * kThreads + tid;
        const int row = c / CH_PER_ROW, ch = c % CH_PER_ROW;
        const char* src = lds + row * kEpiRowBytes + ch * 16;
        const int grow = pass * ROWS + row;
        if (EPI == kEpiBiasResidF32) {
          const int64_t off = (int64_t)(m0 + grow) * a.N + n0 + ch * 4;
          float4 x = *reinterpret_cast<const float4*>(src);
          x.x += rs_f32[it].x; x.y += rs_f32[it].y; x.z += rs_f32[it].z; x.w += rs_f32[it].w;
          *reinterpret_cast<float4*>(a.out_f32 + off) = x;
        } else if (EPI == kEpiBiasResidBf16) {  // fp32 staged sum + bf16 residual -> ONE rounding to bf16
          const int64_t off = (int64_t)(m0 + grow) * a.N + n0 + ch * 4;
          const float4 x = *reinterpret_cast<const float4*>(src);
          const bf16x4 rsd = rs_b16[it];
          const bf16x4 o = {(__bf16)(x.x + (float)rsd[0]), (__bf16)(x.y + (float)rsd[1]), (__bf16)(x.z + (float)rsd[2]),
                            (__bf16)(x.w + (float)rsd[3])};
          *reinterpret_cast<bf16x4*>(a.out_bf16 + off) = o;
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
      CAPAMD_STAMP();
    }
#undef CAPAMD_STAMP
  }
};

// 1-D grid of (M/BM)*(N/BN) blocks.  Hardware block b runs on XCD b % 8 (observed, used for speed only): the
// remap hands every XCD one contiguous range of tiles, n fastest, so the blocks that share an activation panel
// run on the same XCD and hit its L2 instead of fetching the panel once per XCD over the fabric.
template <int BM, int BN, int WAVES_M, int WAVES_N, int EPI>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void gemm_bf16_kernel(GemmArgs a) {
  using G = GemmKernel<BM, BN, WAVES_M, WAVES_N, EPI>;
  extern __shared__ __attribute__((aligned(16))) char gemm_lds[];
  const int tn = a.N / BN;
  const int nblk = gridDim.x, b = blockIdx.x;
  const int q = nblk >> 3, r = nblk & 7, xcd = b & 7, p = b >> 3;
  const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + p;  // bijective for any nblk
  const int n0 = (tile % tn) * BN, m0 = (tile / tn) * BM;
  if (EPI == kEpiQkv && n0 >= 2 * a.H)
    G::template run<true>(a, gemm_lds, m0, n0);
  else
    G::template run<false>(a, gemm_lds, m0, n0);
}

}  // namespace capamd
