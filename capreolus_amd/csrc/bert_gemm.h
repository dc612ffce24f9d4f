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
// Persistent blocks: one workgroup per CU walks its tiles (XCD-aware order); the four half-bursts of the NEXT
// tile are issued before the epilogue of the current one, so the ~7k-cycle fill latency of a tile start is
// hidden behind the epilogue instead of being paid per tile.
//
// Epilogue: no block barrier and none of the operand regions (it must not disturb the prefetch): every wave
// transposes its own sub-tile through a private 4 KiB LDS buffer, 32 rows x 128 bytes at a time (16-byte chunks
// XOR-swizzled by row), and writes/reads it back as whole 128-byte row segments -> 16-byte stores, 8 rows x 128
// contiguous bytes per wave-instruction.  (Storing the MFMA layout directly, 32 rows x 32 bytes per instruction,
// measured 15.8k cycles per tile against ~6k.)  Bias, x1/8 for Q, erf-GELU and the residual add are fused; the
// residual is fetched in the copy-out shape before the prefetch is issued and added in fp32 (one rounding).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace capamd {

typedef __attribute__((ext_vector_type(16))) float f32x16;

// 16-bit operand type of the encoder: __bf16 (default; 8 exponent / 7 mantissa bits) or _Float16 (5 / 10 bits: what
// the reference's `amp=pred` autocast uses on CUDA, trainer/pytorch.py:323-326, and ~8x smaller rounding error; same
// MFMA rate).  Selected per model (capamd_bert_model.compute_dtype).
template <typename T>
struct Half;
template <>
struct Half<__bf16> {
  typedef __attribute__((ext_vector_type(8))) __bf16 x8;
  typedef __attribute__((ext_vector_type(4))) __bf16 x4;
  static __device__ __forceinline__ f32x16 mfma(x8 a, x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
  // the two values of one packed register as fp32 (a shift and a mask: bf16 is the upper half of an fp32)
  static __device__ __forceinline__ __attribute__((ext_vector_type(2))) float unpack2(unsigned u) {
    return {__builtin_bit_cast(float, u << 16), __builtin_bit_cast(float, u & 0xffff0000u)};
  }
};
template <>
struct Half<_Float16> {
  typedef __attribute__((ext_vector_type(8))) _Float16 x8;
  typedef __attribute__((ext_vector_type(4))) _Float16 x4;
  static __device__ __forceinline__ f32x16 mfma(x8 a, x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ __attribute__((ext_vector_type(2))) float unpack2(unsigned u) {
    typedef __attribute__((ext_vector_type(2))) _Float16 h2;
    const h2 h = __builtin_bit_cast(h2, u);
    return {(float)h.x, (float)h.y};
  }
};
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

enum GemmEpilogue {
  kEpiBiasBf16 = 0,      // out_bf16[m][n] = acc + bias[n]
  kEpiBiasGeluBf16 = 1,  // out_bf16[m][n] = gelu(acc + bias[n])                      (erf GELU)
  kEpiQkv = 3,           // n < H: Q[m][n] = (acc+bias)/8 ; n < 2H: K[m][n-H] ; else V^T[(psg,head)][d][key]
  kEpiBiasResidBf16 = 4, // out_bf16[m][n] = acc + bias[n] + resid_bf16[m][n]         (pre-LayerNorm sum, bf16 stream)
  kEpiResidStats = 5,    // chunk-major pre-LayerNorm sum with the residual re-normalised on the fly + row statistics (fused LayerNorm)
};

struct GemmArgs {
  const void* A;        // [M, K] activations (16-bit type T)
  const void* W;        // [N, K] weights
  const float* bias;    // [N]
  int M, N, K;
  void* out_bf16;       // kEpiBias*: [M, N];  kEpiQkv: Q [M, H]   (16-bit type T, whatever the name says)
  void* out_k;          // kEpiQkv: K [M, H]
  void* out_vt;         // kEpiQkv: V^T [M/S * heads][64][S]
  const void* resid_bf16;    // kEpiBiasResidBf16: [M, N] (type T)
  int H, S, heads;      // kEpiQkv geometry (head_dim = 64)
  int ngroup;           // column tiles per scheduling group (divides N / 256; 0 = all of them)
  int a_cm, out_cm;     // A operand / output in the chunk-major activation layout (see cm_offset) instead of row-major
  int w_cm;             // W chunk-major as well: the 4-wave ring kernel (bert_gemm_ring.h; needs a_cm too)
  // ---- fused LayerNorm (see "LayerNorm folded into the GEMMs" below) ----
  // consumer side: A holds the UN-normalised pre-LayerNorm sums P; with W' = W . gamma packed as the weight matrix,
  //   LN(P) W^T + b  ==  rstd_m (acc - mu_m cs_n) + c_n ,   cs_n = sum_k W'[n][k],  c_n = b_n + sum_k beta_k W[n][k]  (passed as `bias`)
  const float* ln_mu;   // [M] row means of A            (NULL: A is already normalised, plain bias epilogue)
  const float* ln_rstd; // [M] 1 / sqrt(var + eps)
  const float2* ln_mr;  // [M] the same two, interleaved (mu, rstd): one 8-byte load per row where a lane owns a row
  const float* ln_cs;   // [N]
  // producer side (kEpiResidStats): out = acc + bias' + resid,  resid = (R - rmu_m) rrstd_m rgamma_n   (+ beta folded into bias')
  const void* res_src;  // [M, N] chunk-major: the un-normalised tensor the residual is the LayerNorm of (or an already
                        // normalised tensor with rmu = 0, rrstd = 1, rgamma = 1 - the embedding output of layer 0)
  const float2* res_mr;                        // [M] (mu, rstd) of the rows of R
  const float* res_gamma;                      // [N]
  float* stat_part;     // [M][N / 64][2]: (sum, sum of squares) of the 64 output columns each wave column writes
  int ring_rows;        // ring kernel: 256 = one workgroup per CU (128 x 128 wave tiles), 128 = two per CU; 0 = the library default
  int ring_mfma16;      // 128-row ring tile: on 16x16x32 MFMAs (bert_gemm_ring16.h, two workgroups per CU) instead of 32x32x16
  int ring_mfma32;      // 256-row ring tile: stay on 32x32x16 MFMAs (bert_gemm_ring.h) instead of 16x16x32 (bert_gemm_ring16.h)
  int res_touch;        // 16x16x32 ring, kEpiResidStats: touch the tile's residual lines in the first steps of its K loop (A/B switch; default off)
  int ring_stagger;     // ring kernel, two workgroups per CU: blocks of the grid's second half start this many x 64 cycles late
  unsigned long long* dbg;  // optional per-block cycle stamps [blocks][32] (profiling builds of the benches only)
};

typedef __attribute__((ext_vector_type(2))) float f32x2;

#ifndef CAPAMD_PP_GLDS_POS
#define CAPAMD_PP_GLDS_POS 2   // where a phase issues its LDS-DMA pair: 0 before its LDS reads, 1 after them, 2 after its first MFMA pair, 3 one after the first and one after the third pair
#endif
#ifndef CAPAMD_PP_LOADS_HALF
#define CAPAMD_PP_LOADS_HALF 0     // four decimal digits (phases P1..P4): pieces of the phase's LDS-DMA pair issued in its loads half, before the LDS reads
#endif
#ifndef CAPAMD_PP_MFMA_SLOT
#define CAPAMD_PP_MFMA_SLOT 0      // after which MFMA pair (0..3) of the MFMA half the remaining pieces are issued
#endif
#ifndef CAPAMD_PP_A_NT
#define CAPAMD_PP_A_NT 0   // 1: activation-panel LDS-DMA with the nt (streaming) cache policy - A/B builds only
#endif
#ifndef CAPAMD_GEMM_ABLATE
#define CAPAMD_GEMM_ABLATE 0   // profiling builds only (ping-pong kernel): 1 skip the MFMAs, 2 skip the LDS-DMA fill
#endif

// erf-GELU 0.5 x (1 + erf(x/sqrt2)) on two values at once, one v_exp_f32 per value and no division:
//   erf(z) = 1 - 2^(z (c0 + c1 z + ... + c4 z^4)) on z >= 0 (minimax fit, |error| < 6.2e-7 on [0, 4.2], decaying beyond),
//   gelu(x) = max(x, 0) - 0.5 |x| 2^P(|x|)         (erf odd; the polynomial below is in |x| = z sqrt2 directly)
// fp32 evaluation error of the whole expression < 1.2e-6 absolute for any x: three orders below the 16-bit rounding
// applied to the result (2^-11 relative for fp16, 2^-8 for bf16).  tests/test_gpu_bert.py::test_gemm_gelu pins it.
//
// The argument is h = x / 2 (the epilogues fold the 1/2 into the affine form that produces x: a power of two, exact):
//   max(x, 0) = h + |h| exactly, 0.5 |x| = |h|, and P(|x|) evaluated by Horner in |h| with the coefficients scaled by 2, 4, ... 32
//   has every intermediate an exact power-of-two multiple of the Horner chain in |x| - the same bits as the form in x, with
//   per PAIR of values 2 v_and (|h|), 4 v_pk_fma + 1 v_pk_mul, 2 v_exp, 1 v_pk_add, 1 v_pk_fma: 11 issue slots where the form in x needed
//   15 (its two v_max, its 0.5 |x| product, and a clamp of |x| the polynomial does not need: its leading coefficient is
//   negative, P(|x|) < -147 beyond |x| = 12 and falls monotonically from there, so 2^P underflows to 0 by itself).
__device__ __forceinline__ f32x2 gelu_erf_half2(f32x2 h) {
  const f32x2 ah = {__builtin_fabsf(h.x), __builtin_fabsf(h.y)};
  f32x2 p = ah * (32.f * -5.1971009666e-04f) + (16.f * 7.3937495955e-03f);
  p = p * ah + (8.f * -5.2555056596e-02f);
  p = p * ah + (4.f * -4.5925845106e-01f);
  p = p * ah + (2.f * -1.1510906938e+00f);
  p = p * ah;
  const f32x2 e = {__builtin_amdgcn_exp2f(p.x), __builtin_amdgcn_exp2f(p.y)};
  const f32x2 r = h + ah;
  return r - ah * e;
}
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) { return gelu_erf_half2(x * 0.5f); }

__device__ __forceinline__ int swz_chunk(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }   // 128-byte rows (attention K tile)
__device__ __forceinline__ int swz_chunk4(int row, int chunk) { return chunk ^ ((row >> 2) & 3); }  // 64-byte rows (GEMM half regions)

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}

// Chunk-major activation layout of a [M][C] 16-bit tensor (M % 32 == 0, C % 8 == 0): blocks of 32 rows, and inside a
// block the 16-byte chunks (8 elements) of one column position of all 32 rows are contiguous:
//     element (m, c) -> (((m >> 5) * (C / 8) + (c >> 3)) * 32 + (m & 31)) * 8 + (c & 7)
// A GEMM epilogue whose lanes own rows (one lane = one m, 8 consecutive n after a lane-pair exchange) then stores
// 1 KiB contiguous per instruction straight from registers (no LDS regrouping), and a consumer's LDS-DMA still reads
// full 128-byte lines (8 consecutive rows of one chunk).
__device__ __host__ __forceinline__ int64_t cm_offset(int64_t m, int c, int C) {
  return (((m >> 5) * (C >> 3) + (c >> 3)) * 32 + (m & 31)) * 8 + (c & 7);
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int EPI, typename T = __bf16>
struct GemmKernel {
  using bf16x8 = typename Half<T>::x8;  // (names kept from the bf16-only version: "the 16-bit vector types")
  using bf16x4 = typename Half<T>::x4;
  static constexpr int kWaves = WAVES_M * WAVES_N;
  static constexpr int kThreads = 64 * kWaves;
  static constexpr int BK = 64;
  static constexpr int WMT = BM / WAVES_M, WNT = BN / WAVES_N;  // per-wave extent in m and n
  static constexpr int TM = WMT / 32, TN = WNT / 32;            // 32x32 MFMA tiles per wave
  static constexpr int kStageBytes = (BM + BN) * BK * 2;        // one K step of both operands
  static constexpr int kHalfBytes = kStageBytes / 2;            // one k-half (32 of the 64 k) of both operands
  static constexpr int kBurst = (BM + BN) * 64 / 1024 / kWaves; // global_load_lds instructions per wave per half-burst
  static constexpr bool kResid = (EPI == kEpiBiasResidBf16);
  static constexpr bool kAccInAgprs = TN * TM * 16 > 128;        // one wave per SIMD, 256 accumulators: they live in AGPRs
  // epilogue rounds: one 32-row x 128-byte block per round = two 32x32 tiles in bf16, one in fp32 (residual sum)
  static constexpr int kStores = kResid ? TN * TM * 4 : TN * TM * 2;  // global store instructions per wave per tile
  static constexpr int kEpiLds = 4096;                                // wave-private staging bytes
  static constexpr int kLdsBytes = 2 * kStageBytes + kWaves * kEpiLds;
  static_assert(WMT % 32 == 0 && WNT % 32 == 0, "wave tile must be a multiple of 32x32");
  static_assert((BM * 4) % (64 * kWaves) == 0 && (BN * 4) % (64 * kWaves) == 0, "stage loop must divide evenly");
  static_assert(3 * kBurst + kStores <= 63, "vmcnt immediate");

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
      const T* src = static_cast<const T*>(a.A) + (int64_t)(m0 + row) * a.K + kcol + swz_chunk4(row, p) * 8;
      __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(base + (wave * A_INSTR + t) * 1024), 16, 0, 0);
    }
    char* wbase = base + BM * 64;
#pragma unroll
    for (int t = 0; t < W_INSTR; ++t) {
      const int row = (wave * W_INSTR + t) * 16 + r16;
      const T* src = static_cast<const T*>(a.W) + (int64_t)(n0 + row) * a.K + kcol + swz_chunk4(row, p) * 8;
      __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(wbase + (wave * W_INSTR + t) * 1024), 16, 0, 0);
    }
  }

  static __device__ __forceinline__ void issue_prologue(const GemmArgs& a, char* lds, int m0, int n0, int wave, int lane) {
    stage_half(a, lds, 0, 0, m0, n0, wave, lane);
    stage_half(a, lds, 0, 1, m0, n0, wave, lane);
    if (a.K > BK) {
      stage_half(a, lds, 1, 0, m0, n0, wave, lane);
      stage_half(a, lds, 1, 1, m0, n0, wave, lane);
    }
  }

  // fragment of 16-deep slice `s2` (0/1) of a half region: row, k = s2*16 + half*8 .. +7
  static __device__ __forceinline__ bf16x8 frag(const char* tile, int row, int s2, int half) {
    return *reinterpret_cast<const bf16x8*>(tile + row * 64 + swz_chunk4(row, 2 * s2 + half) * 16);
  }

  struct Lane {
    int tid, lane, wave, wm, wn, l31, half;
  };

  // ---- K loop of one tile; the tile's prologue bursts have been issued (pending_stores 16-byte stores of the
  // previous tile's epilogue were issued after them) ----------------------------------------------------------
  template <bool TRANS>
  static __device__ __forceinline__ void k_loop(const GemmArgs& a, char* lds, int m0, int n0, const Lane& L, bool pending_stores,
                                                bool chain, int m1, int n1, f32x16 (&acc)[TN][TM]) {
    const int KT = a.K / BK;
    const int wave = L.wave, lane = L.lane, wm = L.wm, wn = L.wn, l31 = L.l31, half = L.half;
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
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
    // wait for the oldest outstanding burst (newer = bursts issued after it; with_stores: the previous tile's epilogue
    // stores were also issued after it and need not have drained yet), then barrier
    auto publish = [&](int newer, bool with_stores) {
      if (with_stores && newer >= 2) wait_vmcnt<2 * kBurst + kStores>();
      else if (newer >= 2) wait_vmcnt<2 * kBurst>();
      else if (newer == 1) wait_vmcnt<kBurst>();
      else wait_vmcnt<0>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // my fragment reads of the region about to be re-staged are done
      __builtin_amdgcn_s_barrier();
    };
    // first half-burst landed?  newer ops: the other prologue bursts (+ the previous tile's stores, issued later)
    if (KT > 1) {
      if (pending_stores) wait_vmcnt<3 * kBurst + kStores>();
      else wait_vmcnt<3 * kBurst>();
    } else {
      if (pending_stores) wait_vmcnt<kBurst + kStores>();
      else wait_vmcnt<kBurst>();
    }
    __builtin_amdgcn_s_barrier();
    load_frags(0, 0, 0);
    for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int cur = ks & 1, nxt = cur ^ 1;
        if (ks == 1) {
          // need (kt, half 1); newer bursts: (kt+1, 0), (kt+1, 1) when step kt+1 exists
          publish((kt + 1 < KT || chain) ? 2 : 0, pending_stores && kt <= 1 && KT > 2);
          if (kt + 2 < KT) stage_half(a, lds, kt + 2, 0, m0, n0, wave, lane);
          else if (chain) stage_half(a, lds, kt + 2 - KT, 0, m1, n1, wave, lane);  // the fill stream runs on into the next tile
        } else if (ks == 3) {
          // need (kt+1, half 0); newer: (kt+1, 1), (kt+2, 0) when step kt+2 exists
          publish((kt + 2 < KT || chain) ? 2 : 1, pending_stores && kt == 0 && KT > 2);
          if (kt + 2 < KT) stage_half(a, lds, kt + 2, 1, m0, n0, wave, lane);
          else if (chain) stage_half(a, lds, kt + 2 - KT, 1, m1, n1, wave, lane);
        }
        if (ks < 3) load_frags(kt, ks + 1, nxt);
        else if (kt + 1 < KT) load_frags(kt + 1, 0, nxt);
        __builtin_amdgcn_sched_barrier(0);  // keep the prefetch reads ahead of this slice's MFMAs (their latency hides under them)
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int j = 0; j < TM; ++j)
            acc[i][j] = TRANS ? Half<T>::mfma(fa[cur][j], fw[cur][i], acc[i][j]) : Half<T>::mfma(fw[cur][i], fa[cur][j], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // every wave is past its last fragment read before anyone re-stages the regions for the next tile
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  // Global element offset of (row r of the wave's 32-row block rb, element column c of its column block cb).
  // !TRANS: rows are m (block j), columns n.   TRANS (V^T): rows are n (block i), columns m (keys).
  template <bool TRANS>
  static __device__ __forceinline__ int64_t out_offset(const GemmArgs& a, int m0, int n0, const Lane& L, int rb, int r, int ccol) {
    if (!TRANS) {
      const int m = m0 + L.wm * WMT + rb * 32 + r;
      const int n = n0 + L.wn * WNT + ccol;
      if (EPI == kEpiQkv) return (int64_t)m * a.H + (n < a.H ? n : n - a.H);
      return (int64_t)m * a.N + n;
    } else {
      const int nv = n0 + L.wn * WNT + rb * 32 + r - 2 * a.H, head = nv >> 6, d = nv & 63;
      const int mg = m0 + L.wm * WMT + ccol, psg = mg / a.S, key = mg % a.S;
      return ((int64_t)(psg * a.heads + head) * 64 + d) * a.S + key;
    }
  }

  // residual in the copy-out shape of epilogue round (i, j): lane reads 4 x (row = c >> 3, 4 elements at chunk c & 7)
  template <int R0, int R1>
  static __device__ __forceinline__ void load_resid(const GemmArgs& a, int m0, int n0, const Lane& L, bf16x4 (&rs)[R0][R1][4]) {
    static_assert(R0 == TN && R1 == TM, "residual registers only exist for the residual epilogue");
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int c = L.lane + 64 * k;
          rs[i][j][k] = *reinterpret_cast<const bf16x4*>(static_cast<const T*>(a.resid_bf16) + out_offset<false>(a, m0, n0, L, j, c >> 3, i * 32 + (c & 7) * 4));
        }
  }

  static __device__ __forceinline__ void epilogue_resid(const GemmArgs& a, char* wl, int m0, int n0, const Lane& L,
                                                        f32x16 (&acc)[TN][TM], bf16x4 (&rs)[TN][TM][4]) {
    epilogue_impl<false>(a, wl, m0, n0, L, acc, rs);
  }
  template <bool TRANS>
  static __device__ __forceinline__ void epilogue(const GemmArgs& a, char* wl, int m0, int n0, const Lane& L, f32x16 (&acc)[TN][TM],
                                                  bf16x4 (&rs)[1][1][4]) {
    epilogue_impl<TRANS>(a, wl, m0, n0, L, acc, rs);
  }
  template <bool TRANS, int R0, int R1>
  static __device__ __forceinline__ void epilogue_impl(const GemmArgs& a, char* wl, int m0, int n0, const Lane& L,
                                                       f32x16 (&acc)[TN][TM], bf16x4 (&rs)[R0][R1][4]) {
    // folded LayerNorm, lane <-> row orientation: (mu, rstd) of this lane's TM rows, one 8-byte load each
    float2 mr_row[TM];
#pragma unroll
    for (int j = 0; j < TM; ++j) mr_row[j] = (!TRANS && a.ln_mu) ? a.ln_mr[m0 + L.wm * WMT + j * 32 + L.l31] : make_float2(0.f, 1.f);
    // value of accumulator element (i, j, g4, e) after bias / scale / GELU
    // TRANS: this lane's column n = l31 of each of the TN column tiles (hoisted: one load per tile, not per call)
    float bias_t[TN], cs_t[TN];
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      bias_t[i] = TRANS ? a.bias[n0 + L.wn * WNT + i * 32 + L.l31] : 0.f;
      cs_t[i] = (TRANS && a.ln_mu) ? a.ln_cs[n0 + L.wn * WNT + i * 32 + L.l31] : 0.f;
    }
    // (one straight-line form: without folded LayerNorm mu = 0, rstd = 1 and the cs vector is never dereferenced - the row statistics
    // are then the neutral pair, and rstd (acc - 0 cs) + b is acc + b exactly; a run-time branch per 4 values cost more than it saved)
    const bool ln = a.ln_mu != nullptr;
    const bool q_tile = (EPI == kEpiQkv) && !TRANS && n0 < a.H;   // 1/sqrt(head_dim = 64) folded into Q (exact in 16 bits)
    const float es = q_tile ? 0.125f : 1.f;
    auto finish = [&](int i, int j, int g4, float (&v)[4]) {
      if (TRANS) {    // registers <-> 4 consecutive rows m, lane <-> n
        const float b = bias_t[i], cs = cs_t[i];
        float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), rs = make_float4(1.f, 1.f, 1.f, 1.f);
        if (ln) {
          const int mrow = m0 + L.wm * WMT + j * 32 + 8 * g4 + 4 * L.half;
          mu = *reinterpret_cast<const float4*>(a.ln_mu + mrow);
          rs = *reinterpret_cast<const float4*>(a.ln_rstd + mrow);
        }
        v[0] = rs.x * (acc[i][j][g4 * 4 + 0] - mu.x * cs) + b;
        v[1] = rs.y * (acc[i][j][g4 * 4 + 1] - mu.y * cs) + b;
        v[2] = rs.z * (acc[i][j][g4 * 4 + 2] - mu.z * cs) + b;
        v[3] = rs.w * (acc[i][j][g4 * 4 + 3] - mu.w * cs) + b;
      } else {        // lane <-> row m, registers <-> 4 consecutive n
        const int n = n0 + L.wn * WNT + i * 32 + 8 * g4 + 4 * L.half;
        const float4 b4 = *reinterpret_cast<const float4*>(a.bias + n);
        const float4 cs = *reinterpret_cast<const float4*>((ln ? a.ln_cs : a.bias) + n);
        const float mu = mr_row[j].x, rs = mr_row[j].y * es;
        v[0] = rs * (acc[i][j][g4 * 4 + 0] - mu * cs.x) + b4.x * es;
        v[1] = rs * (acc[i][j][g4 * 4 + 1] - mu * cs.y) + b4.y * es;
        v[2] = rs * (acc[i][j][g4 * 4 + 2] - mu * cs.z) + b4.z * es;
        v[3] = rs * (acc[i][j][g4 * 4 + 3] - mu * cs.w) + b4.w * es;
      }
      if (EPI == kEpiBiasGeluBf16) {
        const f32x2 g0 = gelu_erf2(f32x2{v[0], v[1]}), g1 = gelu_erf2(f32x2{v[2], v[3]});
        v[0] = g0.x; v[1] = g0.y; v[2] = g1.x; v[3] = g1.y;
      }
    };
    const int swz = (L.l31 & 7);
    if (kResid) {
      // fp32 staging, one 32x32 tile per round: lane writes 4 x float4 at row l31, chunk 2*g4 + half
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) {
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            float v[4];
            finish(i, j, g4, v);
            *reinterpret_cast<float4*>(wl + L.l31 * 128 + (((2 * g4 + L.half) ^ swz) << 4)) = make_float4(v[0], v[1], v[2], v[3]);
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int c = L.lane + 64 * k, row = c >> 3, ch = c & 7;
            const float4 x = *reinterpret_cast<const float4*>(wl + row * 128 + ((ch ^ (row & 7)) << 4));
            const bf16x4 r4 = rs[kResid ? i : 0][kResid ? j : 0][k];
            const bf16x4 o = {(T)(x.x + (float)r4[0]), (T)(x.y + (float)r4[1]), (T)(x.z + (float)r4[2]),
                              (T)(x.w + (float)r4[3])};  // fp32 sum, ONE rounding
            *reinterpret_cast<bf16x4*>(static_cast<T*>(a.out_bf16) + out_offset<false>(a, m0, n0, L, j, row, i * 32 + ch * 4)) = o;
          }
          asm volatile("" ::: "memory");  // next round's writes stay behind these reads (DS ops of a wave execute in order)
        }
    } else {
      // bf16 staging, two 32x32 tiles (64 columns) per round when the wave has them
      constexpr int RB = TRANS ? TN : TM;            // 32-row blocks of the staged orientation
      constexpr int CB = TRANS ? TM : TN;            // 32-column blocks
      constexpr int CP = (CB % 2 == 0) ? 2 : 1;      // column blocks per round
      T* base = static_cast<T*>(a.out_bf16);
      if (EPI == kEpiQkv) base = static_cast<T*>(TRANS ? a.out_vt : (n0 < a.H ? a.out_bf16 : a.out_k));
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int cp = 0; cp < CB / CP; ++cp) {
#pragma unroll
          for (int cc = 0; cc < CP; ++cc) {
            const int cb = cp * CP + cc;
            const int i = TRANS ? rb : cb, j = TRANS ? cb : rb;
            if (kAccInAgprs) {   // (see CmEpilogue::pin_acc: keep the tile in AGPRs until its round)
              __builtin_amdgcn_sched_barrier(0);
              asm volatile("" : "+a"(acc[i][j]));
              __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
              float v[4];
              finish(i, j, g4, v);
              const bf16x4 o = {(T)v[0], (T)v[1], (T)v[2], (T)v[3]};
              *reinterpret_cast<bf16x4*>(wl + L.l31 * 128 + (((4 * cc + g4) ^ swz) << 4) + L.half * 8) = o;
            }
          }
          // no wait between a wave's own LDS writes and reads (its DS operations execute in issue order); the reads of
          // this round are only waited for by the global stores, with the next round's LDS traffic already issued
#pragma unroll
          for (int k = 0; k < 2 * CP; ++k) {
            const int c = L.lane + 64 * k;
            const int row = CP == 2 ? c >> 3 : c >> 2, ch = CP == 2 ? c & 7 : c & 3;
            const uint4 x = *reinterpret_cast<const uint4*>(wl + row * 128 + ((ch ^ (row & 7)) << 4));
            *reinterpret_cast<uint4*>(base + out_offset<TRANS>(a, m0, n0, L, rb, row, cp * CP * 32 + ch * 8)) = x;
          }
        }
    }
  }

  // tile -> (m0, n0) of the persistent schedule; returns false when the block has no it-th tile
  static __device__ __forceinline__ bool tile_of(const GemmArgs& a, int it, int& m0, int& n0) {
    const int tn = a.N / BN, nblk = (a.M / BM) * tn;
    const int G = gridDim.x, b = blockIdx.x, xcd = b & 7, p = b >> 3;
    const int nb_x = (G - xcd + 7) >> 3;                       // blocks resident on this XCD
    const int q = nblk >> 3, r = nblk & 7;
    const int cnt = q + (xcd < r ? 1 : 0), start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const int t = p + it * nb_x;
    if (t >= cnt) return false;
    // tile id -> (m, n): n runs fastest inside a group of a.ngroup column tiles, then m, then the group - so the
    // workgroups of an XCD that run at the same time share ngroup weight panels (kept L2-resident) instead of all tn
    const int id = start + t, grp = a.ngroup > 0 ? a.ngroup : tn, per = (a.M / BM) * grp;
    const int ng = id / per, rem = id - ng * per;
    m0 = (rem / grp) * BM;
    n0 = (ng * grp + rem % grp) * BN;
    return true;
  }

  static __device__ __forceinline__ void run(const GemmArgs& a, char* lds) {
    Lane L;
    L.tid = threadIdx.x; L.lane = L.tid & 63; L.wave = L.tid >> 6;
    L.wm = L.wave % WAVES_M; L.wn = L.wave / WAVES_M; L.l31 = L.lane & 31; L.half = L.lane >> 5;
    unsigned long long* dbg = a.dbg ? a.dbg + (size_t)blockIdx.x * 32 : nullptr;
    int dbg_i = 0;
#define CAPAMD_STAMP() do { if (dbg && L.tid == 0 && dbg_i < 32) dbg[dbg_i++] = __builtin_readcyclecounter(); } while (0)
    int m0, n0;
    if (!tile_of(a, 0, m0, n0)) return;
    CAPAMD_STAMP();
    issue_prologue(a, lds, m0, n0, L.wave, L.lane);
    bool pending = false;
    for (int it = 0;; ++it) {
      int m1 = 0, n1 = 0;
      const bool more = tile_of(a, it + 1, m1, n1);
      f32x16 acc[TN][TM];
      bf16x4 rs[kResid ? TN : 1][kResid ? TM : 1][4];
      const bool trans = (EPI == kEpiQkv) && n0 >= 2 * a.H;
      // with an even number of K steps the region parity carries over, so the next tile's first two K steps are
      // staged by this tile's last two (the L2->LDS stream never stops); otherwise they are issued after the loop
      const bool chain = more && ((a.K / BK) & 1) == 0;
      if (trans) k_loop<true>(a, lds, m0, n0, L, pending, chain, m1, n1, acc);
      else k_loop<false>(a, lds, m0, n0, L, pending, chain, m1, n1, acc);
      CAPAMD_STAMP();
      if constexpr (kResid) load_resid(a, m0, n0, L, rs);  // older than the prefetch: waiting for it does not wait for the bursts
      if (more && !chain) issue_prologue(a, lds, m1, n1, L.wave, L.lane);
      char* wl = lds + 2 * kStageBytes + L.wave * kEpiLds;
      if constexpr (kResid) {
        epilogue_resid(a, wl, m0, n0, L, acc, rs);
      } else {
        if (trans) epilogue<true>(a, wl, m0, n0, L, acc, rs);
        else epilogue<false>(a, wl, m0, n0, L, acc, rs);
      }
      CAPAMD_STAMP();
      if (!more) break;
      pending = true;
      m0 = m1; n0 = n1;
    }
#undef CAPAMD_STAMP
  }
};

// Persistent kernel: grid = min(#tiles, #CUs) blocks of one workgroup per CU.  Hardware block b runs on XCD b % 8
// (observed, used for speed only): every XCD gets one contiguous range of tiles (n fastest) which its resident
// blocks walk in lock-step order, so blocks that share an activation panel / weight tile hit the same L2.
template <int BM, int BN, int WAVES_M, int WAVES_N, int EPI, typename T>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void gemm_bf16_kernel(GemmArgs a) {
  using G = GemmKernel<BM, BN, WAVES_M, WAVES_N, EPI, T>;
  extern __shared__ __attribute__((aligned(16))) char gemm_lds[];
  G::run(a, gemm_lds);
}

// =====================================================================================================================
// 256x256 tile, "ping-pong" K loop (the kernel the BERT-base shapes run on).
//
// Eight waves as 2 (m) x 4 (n): a wave owns 128 x 64 of the tile = 4 x 2 MFMA tiles of 32x32 (128 accumulator
// registers).  One K step (64 k) of the two operands is staged as FOUR half-tiles of 128 rows x 128 bytes (full cache
// lines, 16 KiB each; 2 LDS-DMA instructions per wave per half-tile):
//     A0 / A1 : rows  wr*128 + h*64 + (0..63)  of the activation panel (both wave rows wr)      -> LDS row wr*64 + r
//     B0 / B1 : rows  wc*64  + h*32 + (0..31)  of the weight panel     (all four wave columns)  -> LDS row wc*32 + c
// and consumed in four phases, each one quadrant (64 m x 32 n) of the wave tile over the whole K step (8 MFMAs):
//     P1 (A0,B0)   P2 (A0,B1)   P3 (A1,B1)   P4 (A1,B0)
// so that every phase needs at most one new A half (8 ds_read_b128) and one new B half (4); B0 stays in registers from
// P1 to P4.  Each half-tile is therefore read in exactly one phase (A0,B0: P1; B1: P2; A1: P3) and its slot is handed
// back to the fill stream two phases later:  P1 stages B1 of K step g+1, P2 A1 of g+1, P3 A0 of g+2, P4 B0 of g+2 -
// one half-tile per phase, four to five phases (~1.2 K steps, 64-80 KiB per CU) ahead of its use, across tile
// boundaries of the persistent schedule (the next tile's first K steps are fetched during this tile's last ones).
//
// The two wave rows run half a phase apart: a phase is  [loads] barrier [8 MFMAs] barrier,  and wave row 1 starts with
// one extra barrier, so while the four waves of one row (one per SIMD) issue MFMAs, the other four do their LDS reads,
// their LDS-DMA issue and their waits.  Hazards, with that stagger:
//   * read after fill: a half-tile is read in phase p by both rows only after every wave has waited (counted vmcnt,
//     never 0 in steady state: 8 = the four half-tiles issued after the needed one) for its own two pieces in its
//     loads of phase p-1, and a barrier has followed;
//   * fill after read: the LDS reads of phase p are retired (lgkmcnt(0)) right after the phase's first barrier, at the
//     head of its MFMA half - the loads half never waits for LDS latency - so the slot is re-staged from the loads of
//     phase p+2 on (row 1 retires its reads of phase p one barrier before row 0 starts the loads of phase p+2).
// =====================================================================================================================
// Register-direct epilogues into the chunk-major layout, shared by the 8-wave ping-pong kernel (wave tile 128 x 64) and the 4-wave ring
// kernel (128 x 128): G = the GemmKernel<> geometry (TN 32-column tiles per wave, WMT x WNT wave tile, Lane).
template <typename G, int EPI, typename T>
struct CmEpilogue {
  using Lane = typename G::Lane;
  using bf16x8 = typename Half<T>::x8;
  using bf16x4 = typename Half<T>::x4;
  static constexpr int TN = G::TN, TM = G::TM, WNT = G::WNT, WMT = G::WMT;
  // one wave per SIMD (the 256-row ring kernel: registers to spare, nobody to hide a wait): fetch a column group's operands one group
  // ahead; with two waves per SIMD (the 8-wave kernel, the 128-row ring kernel) they stay inside the group
  static constexpr bool kPipe = G::kAccInAgprs;
  static_assert(TN % 2 == 0, "a wave owns whole 64-column statistic slots");

  // The ring kernel's 256 accumulators live in AGPRs; VALU instructions cannot read those, and left alone the register allocator
  // copies ALL of them to VGPRs at the K loop's exit (and spills what does not fit - scratch accesses are VMEM operations, each
  // reload a full `vmcnt(0)` drain).  An empty asm with an AGPR constraint at the head of a column group keeps that group's four
  // tiles in AGPRs up to that point, so the copies happen group by group.
  static __device__ __forceinline__ void pin_acc(f32x16 (&row)[TM]) {
#pragma unroll
    for (int j = 0; j < TM; ++j) asm volatile("" : "+a"(row[j]));
  }

  // lanes 32..63 of x <-> lanes 0..31 of y
  // (inline asm: the compiler's hazard recogniser cannot see the cross-lane read, so the wait states a VALU-written
  // operand needs before a lane-crossing instruction are inserted by hand)
  static __device__ __forceinline__ void swap32(unsigned& x, unsigned& y) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
  }

  // Epilogue into a chunk-major output (cm_offset): bias (or the folded-LayerNorm form) (+ Q/8, + GELU) in registers, the
  // two lanes that share a row exchange their 8-byte halves (v_permlane32_swap) so each ends up with one whole 16-byte
  // chunk, and every store instruction writes 1 KiB contiguous (two adjacent chunks x 32 rows).  No LDS, no waits.
  static __device__ __forceinline__ void epilogue_cm(const GemmArgs& a, int m0, int n0, const Lane& L, f32x16 (&acc)[TN][TM]) {
    T* base = static_cast<T*>(a.out_bf16);
    int ncols = a.N, nloc = n0 + L.wn * WNT;
    float scale = 1.f;
    if (EPI == kEpiQkv) {
      ncols = a.H;
      if (n0 >= a.H) { base = static_cast<T*>(a.out_k); nloc -= a.H; }
      else scale = 0.125f;  // 1/sqrt(head_dim = 64) folded into Q (exact in 16-bit)
    }
    // a power of two applied to the result - Q / 8, and the 1/2 of the GELU's half-argument form (gelu_erf_half2) - is folded into the
    // affine form's row scale and column constant: es (rstd (acc - mu cs) + c) = (es rstd)(acc - mu cs) + es c, bit for bit
    const float es = EPI == kEpiBiasGeluBf16 ? 0.5f : scale;
    const bool ln = a.ln_mu != nullptr;
    float2 mr[TM];
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      mr[j] = ln ? a.ln_mr[m0 + L.wm * WMT + j * 32 + L.l31] : make_float2(0.f, 1.f);
      mr[j].y *= es;
    }
    // The column vectors of 32-column group i+1 are fetched BEFORE the stores of group i are issued: VMEM operations retire in order,
    // so a load issued behind 32 stores would wait for all of them (with one wave per SIMD - the ring kernel - nothing hides that).
    float4 b4s[2][4], cs4s[2][4];
    // (pipelined form: no branch on `ln` - without folded LayerNorm mu = 0, so any finite vector serves as cs, e.g. the bias itself;
    // one straight-line block per column group, fenced by sched_barrier(0) so that the scheduler neither hoists all 256 accumulator
    // reads to the top nor sinks the prefetch below the stores - both end in scratch spills, which are VMEM operations themselves)
    const float* csp = ln ? a.ln_cs : a.bias;
    auto load_cols = [&](int i, int buf) {
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int n = n0 + L.wn * WNT + i * 32 + 8 * g4 + 4 * L.half;
        b4s[buf][g4] = *reinterpret_cast<const float4*>(a.bias + n);
        if (kPipe) cs4s[buf][g4] = *reinterpret_cast<const float4*>(csp + n);
        else cs4s[buf][g4] = ln ? *reinterpret_cast<const float4*>(a.ln_cs + n) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    };
    if (kPipe) load_cols(0, 0);
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      if (!kPipe) load_cols(i, i & 1);
      else {
        __builtin_amdgcn_sched_barrier(0);
        if (i + 1 < TN) load_cols(i + 1, (i + 1) & 1);
        pin_acc(acc[i]);
        __builtin_amdgcn_sched_barrier(0);
      }
      float4 b4[4];   // (scaled at the point of use, once per column group: the prefetched vectors are not touched before they are needed)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        b4[g4] = b4s[i & 1][g4];
        if (EPI == kEpiBiasGeluBf16 || EPI == kEpiQkv) { b4[g4].x *= es; b4[g4].y *= es; b4[g4].z *= es; b4[g4].w *= es; }
      }
      const float4 (&cs4)[4] = cs4s[i & 1];
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const int64_t mblk = (m0 + L.wm * WMT + j * 32) >> 5;
        const float mu = mr[j].x, rs = mr[j].y;
        unsigned pk[4][2];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          // es (rstd_m (acc - mu_m cs_n) + c_n) ; with mu = 0, rstd = 1, es = 1 this is acc + bias exactly
          float v[4] = {rs * (acc[i][j][g4 * 4 + 0] - mu * cs4[g4].x) + b4[g4].x, rs * (acc[i][j][g4 * 4 + 1] - mu * cs4[g4].y) + b4[g4].y,
                        rs * (acc[i][j][g4 * 4 + 2] - mu * cs4[g4].z) + b4[g4].z, rs * (acc[i][j][g4 * 4 + 3] - mu * cs4[g4].w) + b4[g4].w};
          if (EPI == kEpiBiasGeluBf16) {   // v = x / 2
            const f32x2 g0 = gelu_erf_half2(f32x2{v[0], v[1]}), g1 = gelu_erf_half2(f32x2{v[2], v[3]});
            v[0] = g0.x; v[1] = g0.y; v[2] = g1.x; v[3] = g1.y;
          }
          const bf16x4 o = {(T)v[0], (T)v[1], (T)v[2], (T)v[3]};
          const uint2 u = __builtin_bit_cast(uint2, o);
          pk[g4][0] = u.x; pk[g4][1] = u.y;
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          // the lower lane collects chunk 2p (its own 4 values + the upper lane's), the upper lane chunk 2p+1
          swap32(pk[2 * p][0], pk[2 * p + 1][0]);
          swap32(pk[2 * p][1], pk[2 * p + 1][1]);
          const int chunk = ((nloc + i * 32) >> 3) + 2 * p + L.half;
          const int64_t off = ((mblk * (ncols >> 3) + chunk) * 32 + L.l31) * 8;
          *reinterpret_cast<uint4*>(base + off) = make_uint4(pk[2 * p][0], pk[2 * p][1], pk[2 * p + 1][0], pk[2 * p + 1][1]);
        }
      }
    }
  }

  // kEpiResidStats: the pre-LayerNorm sum of a residual block, chunk-major, with the LayerNorm of the block's input
  // re-computed on the fly and the row statistics of the result collected for the next LayerNorm:
  //     P[m][n] = acc + bias'[n] + (R[m][n] rrstd_m - rmu_m rrstd_m) rgamma_n       (bias' = b + rbeta, folded when packed)
  //     stat_part[m][n0/64 + wn] = (sum_n P, sum_n P^2) over this wave's 64 columns, of the ROUNDED values the consumers
  //     will read.  fp32 sum, one rounding.
  // Everything up to the rounding happens where the accumulator layout put the value: a lane owns, of each 8-column chunk of its
  // row, the 4 columns 4 half .. 4 half + 3, so it reads the residual as 8-byte pieces at those positions (the lane pair of a row
  // reads 16 adjacent bytes) and the column vectors as float4 at the same columns.  Only the ROUNDED result changes lanes - two
  // v_permlane32_swap per 8-column chunk pair on packed registers, as in epilogue_cm (round 4 exchanged the fp32 accumulators
  // first: one swap, with its two hazard no-ops, per value pair).  The arithmetic per PAIR of values is packed throughout:
  // 1 v_pk_fma (r rrstd - mu rrstd), 1 v_pk_add (acc + bias'), 1 v_pk_fma (. gamma +), 1 v_cvt_pk, 1 v_pk_add + 1 v_pk_fma (statistics)
  // beside the two 16-bit -> fp32 expansions of the residual and of the rounded result.
  static __device__ __forceinline__ void epilogue_cm_resid(const GemmArgs& a, int m0, int n0, const Lane& L, f32x16 (&acc)[TN][TM]) {
    T* base = static_cast<T*>(a.out_bf16);
    const T* rsrc = static_cast<const T*>(a.res_src);
    const int nchunks = a.N >> 3, nloc = n0 + L.wn * WNT, nslot = a.N >> 6;
    float rrs[TM], rc[TM];                      // per row: rrstd and -rmu rrstd
    f32x2 s1[TN / 2][TM], s2[TN / 2][TM];       // per 64-column slot of the wave's WNT columns: (even, odd) element partial sums
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const float2 mr = a.res_mr[m0 + L.wm * WMT + j * 32 + L.l31];
      rrs[j] = mr.y;
      rc[j] = -mr.x * mr.y;
#pragma unroll
      for (int sl = 0; sl < TN / 2; ++sl) s1[sl][j] = s2[sl][j] = f32x2{0.f, 0.f};
    }
    // Operands are fetched one step AHEAD of their use and before the stores of the step in between are issued (in-order retirement of
    // VMEM operations, see epilogue_cm): the column vectors (bias', gamma) of 32-column group i+1 during group i, the residual pieces
    // of row block (i, j+1) during (i, j).  kPipe = false (two waves per SIMD): everything inside its own step.
    float4 bbs[2][4], ggs[2][4];
    bf16x4 r4s[2][4];
    auto load_cols = [&](int i, int buf) {
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int n = n0 + L.wn * WNT + i * 32 + 8 * g4 + 4 * L.half;
        bbs[buf][g4] = *reinterpret_cast<const float4*>(a.bias + n);
        ggs[buf][g4] = *reinterpret_cast<const float4*>(a.res_gamma + n);
      }
    };
    auto load_res = [&](int t, int buf) {   // t = i * TM + j
      const int i = t / TM, j = t % TM;
      const int64_t mblk = (m0 + L.wm * WMT + j * 32) >> 5;
      // (one address per step: the four chunks of a 32-column group are 512 bytes apart - immediate offsets)
      const T* rp = rsrc + ((mblk * nchunks + ((nloc + i * 32) >> 3)) * 32 + L.l31) * 8 + 4 * L.half;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) r4s[buf][g4] = *reinterpret_cast<const bf16x4*>(rp + g4 * 256);
    };
    if (kPipe) { load_cols(0, 0); load_res(0, 0); }
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      if (!kPipe) load_cols(i, i & 1);
      else {
        __builtin_amdgcn_sched_barrier(0);
        if (i + 1 < TN) load_cols(i + 1, (i + 1) & 1);
        pin_acc(acc[i]);
        __builtin_amdgcn_sched_barrier(0);
      }
      const float4 (&bb)[4] = bbs[i & 1];
      const float4 (&gg)[4] = ggs[i & 1];
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const int t = i * TM + j;
        if (!kPipe) load_res(t, t & 1);
        else {
          __builtin_amdgcn_sched_barrier(0);
          if (t + 1 < TN * TM) load_res(t + 1, (t + 1) & 1);
          __builtin_amdgcn_sched_barrier(0);
        }
        const int64_t mblk = (m0 + L.wm * WMT + j * 32) >> 5;
        unsigned pk[4][2];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const uint2 ru = __builtin_bit_cast(uint2, r4s[t & 1][g4]);
          const f32x2 rr = {rrs[j], rrs[j]}, cc = {rc[j], rc[j]};
          const f32x2 b01 = {bb[g4].x, bb[g4].y}, b23 = {bb[g4].z, bb[g4].w}, g01 = {gg[g4].x, gg[g4].y}, g23 = {gg[g4].z, gg[g4].w};
          const f32x2 v01 = f32x2{acc[i][j][g4 * 4 + 0], acc[i][j][g4 * 4 + 1]} + b01, v23 = f32x2{acc[i][j][g4 * 4 + 2], acc[i][j][g4 * 4 + 3]} + b23;
          const f32x2 t01 = Half<T>::unpack2(ru.x) * rr + cc, t23 = Half<T>::unpack2(ru.y) * rr + cc;
          const f32x2 o01 = t01 * g01 + v01, o23 = t23 * g23 + v23;             // fp32 sum, ONE rounding
          const bf16x4 o = {(T)o01.x, (T)o01.y, (T)o23.x, (T)o23.y};
          const uint2 u = __builtin_bit_cast(uint2, o);
          pk[g4][0] = u.x; pk[g4][1] = u.y;
        }
        T* op = base + ((mblk * nchunks + ((nloc + i * 32) >> 3) + L.half) * 32 + L.l31) * 8;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          // the lower lane collects chunk 2p (its own 4 values + the upper lane's), the upper lane chunk 2p+1
          swap32(pk[2 * p][0], pk[2 * p + 1][0]);
          swap32(pk[2 * p][1], pk[2 * p + 1][1]);
          *reinterpret_cast<uint4*>(op + p * 512) = make_uint4(pk[2 * p][0], pk[2 * p][1], pk[2 * p + 1][0], pk[2 * p + 1][1]);
          // statistics of what the consumers will read, from the registers as stored (the exchange only moved values between the two
          // lanes of a row, whose sums are added below: reading them before it would need a copy of every register it overwrites)
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const f32x2 q = Half<T>::unpack2(pk[2 * p + (k >> 1)][k & 1]);
            s1[i >> 1][j] += q;
            s2[i >> 1][j] = q * q + s2[i >> 1][j];
          }
        }
      }
    }
#pragma unroll
    for (int sl = 0; sl < TN / 2; ++sl)
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        // the two lanes of a row hold its 2 x 32 columns: add them and let the lower lane write the wave's partial
        const float p1 = s1[sl][j].x + s1[sl][j].y, p2 = s2[sl][j].x + s2[sl][j].y;
        const float t1 = p1 + __shfl_xor(p1, 32, 64), t2 = p2 + __shfl_xor(p2, 32, 64);
        const int mrow = m0 + L.wm * WMT + j * 32 + L.l31;
        if (L.half == 0)
          *reinterpret_cast<float2*>(a.stat_part + ((int64_t)mrow * nslot + ((n0 + L.wn * WNT) >> 6) + sl) * 2) = make_float2(t1, t2);
      }
  }

};

template <int EPI, typename T>
struct GemmPingPong {
  using G = GemmKernel<256, 256, 2, 4, EPI, T>;  // epilogues and wave-tile geometry (WMT 128, WNT 64, TM 4, TN 2)
  using CE = CmEpilogue<G, EPI, T>;
  using Lane = typename G::Lane;
  using bf16x8 = typename Half<T>::x8;
  using bf16x4 = typename Half<T>::x4;
  static constexpr int kHalfTile = 128 * 128;     // bytes
  static constexpr int kBuf = 4 * kHalfTile;      // one K step
  static constexpr int kLdsBytes = 2 * kBuf + 8 * G::kEpiLds;
  static constexpr int kThreads = 512;
  enum { kA0 = 0, kA1 = 1, kB0 = 2, kB1 = 3 };

  struct Ctx {
    const T* A; const T* W;   // operand bases
    int K, KT, a_cm;
    int a_off[2], b_off[2];    // per-lane BYTE offsets of this wave's two pieces of an A / B half-tile (half 0, K step 0)
    unsigned a_bytes, w_bytes; // extents of the A / W tensors: the LDS-DMA goes through buffer addressing (resource over
                               // the whole tensor + 32-bit lane offset + scalar tile/K-step offset)
    int koff[4];               // per-lane byte offset of k-slice ks inside a 128-byte LDS row (XOR-swizzled chunk)
    int a_row, b_row;          // per-lane LDS byte offset of row (wr*64 + l31) / (wc*32 + l31)
    int piece;                 // wave * 2048 (scalar): this wave's pieces inside a half-tile
  };

  static __device__ __forceinline__ void make_ctx(const GemmArgs& a, const Lane& L, Ctx& c) {
    c.A = static_cast<const T*>(a.A); c.W = static_cast<const T*>(a.W);
    c.K = a.K; c.KT = a.K / 64; c.a_cm = a.a_cm;
    const int r8 = L.lane >> 3, p = L.lane & 7;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int row = (L.wave * 2 + t) * 8 + r8;            // LDS row of the half-tile this lane fills
      const int chunk = p ^ ((row >> 1) & 7);               // which 16 bytes of the source line land in slot p
      const int rr = (row >> 6) * 128 + (row & 63);         // row inside the 256-row tile (half 0)
      c.a_off[t] = a.a_cm ? (((rr >> 5) * (a.K >> 3) + chunk) * 32 + (rr & 31)) * 16 : (rr * a.K + chunk * 8) * 2;
      c.b_off[t] = (((row >> 5) * 64 + (row & 31)) * a.K + chunk * 8) * 2;
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) c.koff[ks] = ((2 * ks + L.half) ^ ((L.l31 >> 1) & 7)) * 16;
    c.a_row = (L.wm * 64 + L.l31) * 128;
    c.b_row = (L.wn * 32 + L.l31) * 128;
    c.piece = __builtin_amdgcn_readfirstlane(L.wave) * 2048;
    c.a_bytes = (unsigned)((size_t)a.M * a.K * 2);   // the dispatcher guarantees both tensors are below 4 GiB
    c.w_bytes = (unsigned)((size_t)a.N * a.K * 2);
  }

  struct Tiles {   // where the fill stream is: this tile, the next one (if any), and the running K-step parity
    int m0, n0, m1, n1;
    bool more;
    int gk;        // K steps of all earlier tiles of this block (buffer of step g of this tile = (gk + g) & 1)
  };
  struct Src {     // scalar byte offsets of the operand panels of one K step: (m0*K + kt*64)*2 and (n0*K + kt*64)*2
    unsigned a, w;
  };
  // K step g relative to the current tile; g >= KT runs on into the next tile (caller guarantees there is one)
  static __device__ __forceinline__ Src src_of(const Ctx& c, const Tiles& t, int g) {
    const bool in = g < c.KT;
    const int kt = in ? g : g - c.KT, m = in ? t.m0 : t.m1, n = in ? t.n0 : t.n1;
    // (chunk-major A: a 32-row block is 32 * K * 2 bytes and a K step advances 8 chunks of 512 bytes; the half-tile
    // offset of 64 rows is 64 * K * 2 bytes in both layouts)
    const unsigned a_byte = c.a_cm ? (unsigned)((int64_t)(m >> 5) * c.K * 64 + kt * 4096) : (unsigned)(((int64_t)m * c.K + kt * 64) * 2);
    return Src{a_byte, (unsigned)(((int64_t)n * c.K + kt * 64) * 2)};
  }

  // LDS-DMA of half-tile KIND of the K step at `s` into buffer `buf`: two 1-KiB pieces per wave
  template <int KIND, int PIECES = 3>
  static __device__ __forceinline__ void stage(const Ctx& c, char* lds, int buf, const Src& s) {
#if CAPAMD_GEMM_ABLATE & 2
    return;
#endif
#if defined(__HIP_DEVICE_COMPILE__)  // (the buffer-resource builtins do not exist in the host pass of hipcc)
    char* dst = lds + buf * kBuf + KIND * kHalfTile + c.piece;
    const unsigned soff = KIND < 2 ? s.a + (unsigned)((KIND & 1) * 64 * c.K * 2) : s.w + (unsigned)((KIND & 1) * 32 * c.K * 2);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (!((PIECES >> t) & 1)) continue;
      // raw buffer (stride 0, dword format) over the operand; loop-invariant scalar registers
      if constexpr (KIND < 2) {
#if CAPAMD_PP_A_NT
        __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(c.A), 0, (int)c.a_bytes, 0x00020000),
                                                 (lds_void_t*)(dst + t * 1024), 16, c.a_off[t], (int)soff, 0, 2);
#else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(c.A), 0, (int)c.a_bytes, 0x00020000),
                                                 (lds_void_t*)(dst + t * 1024), 16, c.a_off[t], (int)soff, 0, 0);
#endif
      } else {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(c.W), 0, (int)c.w_bytes, 0x00020000),
                                                 (lds_void_t*)(dst + t * 1024), 16, c.b_off[t], (int)soff, 0, 0);
      }
    }
#endif
  }

  static __device__ __forceinline__ bf16x8 rd(const char* half_tile, int row_off, int koff) {
    return *reinterpret_cast<const bf16x8*>(half_tile + row_off + koff);
  }

  // ---- one K step = four phases.  FILL: 2 = the fill stream is running (every phase stages one half-tile and the
  // waits are counted), 1 = only P1 stages (second-to-last K step of the block's last tile), 0 = nothing left to stage.
  template <bool TRANS, int FILL>
  static __device__ __forceinline__ void k_step(const Ctx& c, char* lds, int bcur, const Src& s1, const Src& s2, bool last,
                                                f32x16 (&acc)[2][4], bf16x8 (&fa)[2][4], bf16x8 (&fb0)[4], bf16x8 (&fb1)[4]) {
    const char* buf = lds + bcur * kBuf;
    // One phase = [fragment reads + one half-tile staged] barrier [8 MFMAs] barrier.  At the end of the loads half
    // the half-tile(s) read in the NEXT phase must have landed as far as this wave's pieces go: vmcnt(8) = the four
    // half-tiles issued after the needed one stay in flight.  The phase's own LDS reads are only waited for after the
    // barrier, at the head of the MFMA half.
    // LH = pieces (0, 1, 2) of the phase's half-tile issued in the LOADS half, before its LDS reads; the rest go between the MFMAs.
    // (CAPAMD_PP_GLDS_POS 0 / 1 / 3: the older all-or-nothing placements, kept for A/B builds.)
    auto phase = [&](auto lh_c, auto&& stage_fn, auto&& reads_fn, bool needed, const bf16x8 (&b)[4], int i, int j0) {
      constexpr int LH = decltype(lh_c)::value;
#if CAPAMD_PP_GLDS_POS == 0
      stage_fn(3);
      reads_fn();
#elif CAPAMD_PP_GLDS_POS == 1
      reads_fn();
      stage_fn(3);
#elif CAPAMD_PP_GLDS_POS == 2
      if (LH == 1) stage_fn(1);
      if (LH == 2) stage_fn(3);
      reads_fn();
#else
      reads_fn();
#endif
      if (needed) {
        // the half-tile read in the next phase was issued four phases ago: the six pieces of the three phases in between and this
        // phase's LH pieces may stay in flight
        if (FILL == 2) wait_vmcnt<(CAPAMD_PP_GLDS_POS == 2 ? 6 + LH : CAPAMD_PP_GLDS_POS == 3 ? 6 : 8)>();
        else wait_vmcnt<0>();
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#if CAPAMD_GEMM_ABLATE & 1
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) asm volatile("" ::"v"(b[ks]), "v"(fa[0][ks]), "v"(fa[1][ks]));
#else
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
          acc[i][j0 + jj] = TRANS ? Half<T>::mfma(fa[jj][ks], b[ks], acc[i][j0 + jj]) : Half<T>::mfma(b[ks], fa[jj][ks], acc[i][j0 + jj]);
#if CAPAMD_PP_GLDS_POS == 2
        if (ks == CAPAMD_PP_MFMA_SLOT && LH < 2) { __builtin_amdgcn_sched_barrier(0); stage_fn(LH == 1 ? 2 : 3); __builtin_amdgcn_sched_barrier(0); }
#elif CAPAMD_PP_GLDS_POS == 3
        if (ks == 0) { __builtin_amdgcn_sched_barrier(0); stage_fn(1); __builtin_amdgcn_sched_barrier(0); }
        if (ks == 2) { __builtin_amdgcn_sched_barrier(0); stage_fn(2); __builtin_amdgcn_sched_barrier(0); }
#endif
      }
#endif
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
    };
    using LH1 = std::integral_constant<int, (CAPAMD_PP_LOADS_HALF / 1000) % 10>;
    using LH2 = std::integral_constant<int, (CAPAMD_PP_LOADS_HALF / 100) % 10>;
    using LH3 = std::integral_constant<int, (CAPAMD_PP_LOADS_HALF / 10) % 10>;
    using LH4 = std::integral_constant<int, CAPAMD_PP_LOADS_HALF % 10>;
    auto read_a = [&](int kind) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fa[jj][ks] = rd(buf + kind * kHalfTile, c.a_row + jj * 4096, c.koff[ks]);
    };
    auto read_b = [&](int kind, bf16x8 (&fb)[4]) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) fb[ks] = rd(buf + kind * kHalfTile, c.b_row, c.koff[ks]);
    };
    // P1 (A0, B0): stages B1 of step g+1; needs B1 of this step next
    phase(LH1{}, [&](int pc) { if (FILL >= 1) { if (pc == 3) stage<kB1, 3>(c, lds, bcur ^ 1, s1); else if (pc == 1) stage<kB1, 1>(c, lds, bcur ^ 1, s1); else stage<kB1, 2>(c, lds, bcur ^ 1, s1); } },
          [&] { read_b(kB0, fb0); read_a(kA0); }, true, fb0, 0, 0);
    // P2 (A0, B1): stages A1 of step g+1; needs A1 of this step next
    phase(LH2{}, [&](int pc) { if (FILL >= 1) { if (pc == 3) stage<kA1, 3>(c, lds, bcur ^ 1, s1); else if (pc == 1) stage<kA1, 1>(c, lds, bcur ^ 1, s1); else stage<kA1, 2>(c, lds, bcur ^ 1, s1); } },
          [&] { read_b(kB1, fb1); }, true, fb1, 1, 0);
    // P3 (A1, B1): stages A0 of step g+2
    phase(LH3{}, [&](int pc) { if (FILL == 2) { if (pc == 3) stage<kA0, 3>(c, lds, bcur, s2); else if (pc == 1) stage<kA0, 1>(c, lds, bcur, s2); else stage<kA0, 2>(c, lds, bcur, s2); } },
          [&] { read_a(kA1); }, false, fb1, 1, 2);
    // P4 (A1, B0): stages B0 of step g+2; needs A0, B0 of the next step next
    phase(LH4{}, [&](int pc) { if (FILL == 2) { if (pc == 3) stage<kB0, 3>(c, lds, bcur, s2); else if (pc == 1) stage<kB0, 1>(c, lds, bcur, s2); else stage<kB0, 2>(c, lds, bcur, s2); } },
          [&] {}, !last, fb0, 0, 2);
  }

  // ---- one tile's K loop (the first six half-tiles of the tile are already in flight / landed) ---------------------
  template <bool TRANS>
  static __device__ __forceinline__ void k_loop(const Ctx& c, char* lds, const Tiles& t, const Lane& L, f32x16 (&acc)[2][4]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8 fa[2][4], fb0[4], fb1[4];
    if (L.wm == 1) __builtin_amdgcn_s_barrier();   // wave row 1 runs half a phase behind
    int g = 0;
    const int steady = t.more ? c.KT : c.KT - 2;   // K steps whose stages (g+1, g+2) all exist
    for (; g < steady; ++g) k_step<TRANS, 2>(c, lds, (t.gk + g) & 1, src_of(c, t, g + 1), src_of(c, t, g + 2), false, acc, fa, fb0, fb1);
    if (!t.more) {
      k_step<TRANS, 1>(c, lds, (t.gk + g) & 1, src_of(c, t, g + 1), Src{0u, 0u}, false, acc, fa, fb0, fb1);
      ++g;
      k_step<TRANS, 0>(c, lds, (t.gk + g) & 1, Src{0u, 0u}, Src{0u, 0u}, true, acc, fa, fb0, fb1);
    }
    if (L.wm == 0) __builtin_amdgcn_s_barrier();   // re-align the two wave rows: both run the epilogue at once
  }

  static __device__ __forceinline__ void run(const GemmArgs& a, char* lds) {
    Lane L;
    L.tid = threadIdx.x; L.lane = L.tid & 63; L.wave = L.tid >> 6;
    L.wm = L.wave >> 2; L.wn = L.wave & 3; L.l31 = L.lane & 31; L.half = L.lane >> 5;
    unsigned long long* dbg = a.dbg ? a.dbg + (size_t)blockIdx.x * 32 : nullptr;
    int dbg_i = 0;
#define CAPAMD_STAMP() do { if (dbg && L.tid == 0 && dbg_i < 32) dbg[dbg_i++] = __builtin_readcyclecounter(); } while (0)
    Tiles t;
    t.gk = 0;
    if (!G::tile_of(a, 0, t.m0, t.n0)) return;
    Ctx c;
    make_ctx(a, L, c);
    CAPAMD_STAMP();
    t.m1 = t.n1 = 0;
    t.more = G::tile_of(a, 1, t.m1, t.n1);
    // prologue, in the order the steady state would have issued them: A0 B0 B1 A1 of step 0, A0 B0 of step 1
    {
      const Src s0 = src_of(c, t, 0), s1 = src_of(c, t, 1);
      stage<kA0, 3>(c, lds, 0, s0); stage<kB0, 3>(c, lds, 0, s0); stage<kB1, 3>(c, lds, 0, s0); stage<kA1, 3>(c, lds, 0, s0);
      stage<kA0, 3>(c, lds, 1, s1); stage<kB0, 3>(c, lds, 1, s1);
      wait_vmcnt<8>();
      __builtin_amdgcn_s_barrier();
    }
    for (int it = 0;; ++it) {
      f32x16 acc[2][4];
      bf16x4 rs[G::kResid ? 2 : 1][G::kResid ? 4 : 1][4];
      const bool trans = (EPI == kEpiQkv) && t.n0 >= 2 * a.H;
      if (trans) k_loop<true>(c, lds, t, L, acc);
      else k_loop<false>(c, lds, t, L, acc);
      CAPAMD_STAMP();
      if constexpr (G::kResid) G::load_resid(a, t.m0, t.n0, L, rs);
      char* wl = lds + 2 * kBuf + L.wave * G::kEpiLds;
      if constexpr (G::kResid) {
        G::epilogue_resid(a, wl, t.m0, t.n0, L, acc, rs);
      } else {
        if constexpr (EPI == kEpiResidStats) CE::epilogue_cm_resid(a, t.m0, t.n0, L, acc);
        else if (trans) G::template epilogue<true>(a, wl, t.m0, t.n0, L, acc, rs);
        else if (a.out_cm) CE::epilogue_cm(a, t.m0, t.n0, L, acc);
        else G::template epilogue<false>(a, wl, t.m0, t.n0, L, acc, rs);
      }
      CAPAMD_STAMP();
      if (!t.more) break;
      t.gk += c.KT;
      t.m0 = t.m1; t.n0 = t.n1;
      t.more = G::tile_of(a, it + 2, t.m1, t.n1);
    }
#undef CAPAMD_STAMP
  }
};

template <int EPI, typename T>
__global__ __launch_bounds__(512) void gemm_pingpong_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char gemm_lds[];
  GemmPingPong<EPI, T>::run(a, gemm_lds);
}

}  // namespace capamd
