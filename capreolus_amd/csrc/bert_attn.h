// Fused multi-head self-attention for the BERT passage encoder on gfx950 (head_dim 64, S in {64,128,256}).
//
// One workgroup per (passage, head); S/32 waves, each owning 32 query rows and ALL S keys, so the
// softmax is exact (no online rescaling): the S/32 score tiles of a wave (16 fp32 registers per
// 32x32 tile) stay in registers from QK^T to PV.
//
//   scores^T = K · Q^T     (v_mfma_f32_32x32x16_bf16, K rows as the A operand, so a lane holds
//                           ONE query (lane&31) and S/2 keys -> row max/sum are in-lane reductions
//                           plus one exchange with lane^32)
//   P = softmax(scores + additive pad mask)      (Q was pre-scaled by 1/8 in the QKV epilogue)
//   ctx^T = V^T · P^T      (A operand = V^T fragment read from a [64 d][S keys] LDS image, B operand =
//                           the P registers as they are: the MFMA k index is only a summation index,
//                           so the key order the QK^T layout leaves in a lane is used for V as well)
// K is staged with global_load_lds into the same swizzled [rows][64] image the GEMM uses; V^T comes
// from the QKV GEMM already transposed per head and is staged through registers into rows padded
// by 8 bytes (conflict-free ds_read_b64 for 32 lanes reading 32 different d rows).
#pragma once
#include "bert_gemm.h"

namespace capamd {

struct AttnArgs {
  const void* Q;         // [M, H] (pre-scaled by 1/8), 16-bit type T
  const void* K;         // [M, H]
  const void* Vt;        // [M/S * heads][64][S]
  const int64_t* mask;   // [M/S, S] attention mask (1 = attend), rows of the current micro-batch
  void* ctx;             // [M, H]
  int H, heads;
  int qk_cm;             // Q and K in the chunk-major activation layout (bert_gemm.h cm_offset) instead of row-major
  int ctx_cm;            // ctx written chunk-major (the A operand of the ring GEMM, bert_gemm_ring.h) instead of row-major
};

// where lane (query row `tok`, half) stores its 4 consecutive output dimensions d = 32 dt + 8 g4 + 4 half .. + 3 of head `head`:
// chunk-major, the 32 rows of a wave x one 16-byte chunk are 512 contiguous bytes (both halves of a lane pair fill one chunk)
template <typename T>
__device__ __forceinline__ T* ctx_slot(const AttnArgs& a, int64_t tok, int head, int dt, int g4, int half) {
  T* c = static_cast<T*>(a.ctx);
  if (a.ctx_cm) return c + (((tok >> 5) * (a.H >> 3) + head * 8 + dt * 4 + g4) * 32 + (tok & 31)) * 8 + 4 * half;
  return c + tok * a.H + head * 64 + dt * 32 + 8 * g4 + 4 * half;
}

// element offset of the 16-byte chunk `chunk` (0..H/8) of token row `tok` in a [M, H] activation, either layout
__device__ __forceinline__ int64_t qk_offset(const AttnArgs& a, int64_t tok, int chunk) {
  return a.qk_cm ? (((tok >> 5) * (a.H >> 3) + chunk) * 32 + (tok & 31)) * 8 : tok * a.H + chunk * 8;
}

// The softmax of one query row lives in one lane pair (l31, l31+32): NT tiles x 16 registers.  It is VALU work that
// competes with the MFMAs of the other wave on the SIMD, so it is written for few instructions per score:
//  * the additive mask is the accumulator's initial value (no add, no zero-fill),
//  * exp2((s - m) log2e) is one packed fma per two scores (s * log2e - m * log2e) followed by v_exp_f32,
//  * the row sum is accumulated with packed adds into a 16-wide partial and folded once at the end.
__device__ inline f32x16 scores_init_from_mask(const float* madd_t) {
  f32x16 v;
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    const float4 ma = *reinterpret_cast<const float4*>(madd_t + 8 * g4);
    v[g4 * 4 + 0] = ma.x;
    v[g4 * 4 + 1] = ma.y;
    v[g4 * 4 + 2] = ma.z;
    v[g4 * 4 + 3] = ma.w;
  }
  return v;
}

template <int NT>
__device__ inline float softmax_inplace(f32x16 (&sc)[NT]) {
  float mx = -3.4028234663852886e38f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[t][r]);
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  constexpr float kLog2e = 1.4426950408889634f;
  const float nb = -mx * kLog2e;
  f32x16 part;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    f32x16 x = sc[t] * kLog2e + nb;
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = __builtin_amdgcn_exp2f(x[r]);
    sc[t] = x;
    part = t == 0 ? x : part + x;
  }
  float sum = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) sum += part[r];
  sum += __shfl_xor(sum, 32, 64);
  return 1.f / sum;
}

// out * (1 / sum) rounded to the 16-bit type in two steps (fp32 product, then the conversion), in every kernel: under hipcc's default
// -ffp-contract=fast the compiler may fuse the two into one v_fma_mixlo_f16 (a single rounding) in one kernel and not in another,
// and the same passage would then score a 16-bit ulp differently depending on which kernel its length bucket selects.
template <typename T>
__device__ __forceinline__ T scale_round(float v, float inv) {
#pragma clang fp contract(off)
  float m = v * inv;
  asm volatile("" : "+v"(m));
  return (T)m;
}

// ---- the arithmetic every S <= 256 kernel below shares (so that a passage scores bit-identically whichever kernel its length
//      bucket selects) ------------------------------------------------------------------------------------------------------
// Key order: the A-operand row i of a 32-key score tile holds key i with bits 2 and 3 swapped (`key_of_row`), so that register r of
// the tile is key 32 t + 16 (r >> 3) + 8 half + (r & 7): the 8 probabilities a lane feeds to one P V MFMA are 8 CONSECUTIVE keys
// and the matching V^T fragment is 16 contiguous bytes.
__device__ __forceinline__ int key_of_row(int l31) { return (l31 & 19) | ((l31 & 4) << 1) | ((l31 & 8) >> 1); }

template <int NT>
__device__ inline f32x16 scores_init_permuted(const float* madd_t) {   // madd_t = madd + 32 t + 8 half; register r <- key 32t + 16(r>>3) + 8 half + (r&7)
  f32x16 v;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 ma = *reinterpret_cast<const float4*>(madd_t + 16 * (g >> 1) + 4 * (g & 1));
    v[g * 4 + 0] = ma.x;
    v[g * 4 + 1] = ma.y;
    v[g * 4 + 2] = ma.z;
    v[g * 4 + 3] = ma.w;
  }
  return v;
}

// The value the partner lane (lane ^ 32) holds: v_permlane32_swap on two copies (x.hi <-> y.lo leaves x = [lo | lo], y = [hi | hi]) and a
// select - a VALU exchange instead of the ds_bpermute round trip through the LDS pipe `__shfl_xor(v, 32)` compiles to; the softmax has two
// of them in its serial chain (row maximum, row sum).  (Inline asm with the wait states a VALU-written operand needs before a
// lane-crossing instruction, as in bert_gemm.h: swap32.)
__device__ __forceinline__ float partner32(float v, int half) {
  unsigned x = __builtin_bit_cast(unsigned, v), y = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
  return __builtin_bit_cast(float, half ? x : y);   // lower half wants the upper value (now everywhere in y), upper half the lower (in x)
}

// exact softmax of one query row (a lane pair) over NT x 16 scores, probabilities rounded to the 16-bit type as they are produced
// (half a tile at a time: 8 fp32 scores leave as 4 packed registers); returns 1 / sum
struct NoHook {
  __device__ __forceinline__ void operator()(int) const {}
};
// hook(0 .. 2 NT - 1) runs after each half tile (the S = 256 kernel issues its refills there)
template <int NT, typename T, typename Hook = NoHook>
__device__ __forceinline__ float softmax_pack(const f32x16 (&sc)[NT], typename Half<T>::x8 (&p)[NT][2], Hook&& hook = Hook()) {
  float mx = -3.4028234663852886e38f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[t][r]);
  const int half = (int)(threadIdx.x & 32) >> 5;
  mx = fmaxf(mx, partner32(mx, half));
  constexpr float kLog2e = 1.4426950408889634f;
  const float nb = -mx * kLog2e;
  float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      float x[8];
#pragma unroll
      for (int e = 0; e < 8; e += 2) {   // (s log2e - m log2e) for two scores per v_pk_fma_f32: the same fma per element, half the issue slots
        const f32x2 y = f32x2{sc[t][8 * s2 + e], sc[t][8 * s2 + e + 1]} * f32x2{kLog2e, kLog2e} + f32x2{nb, nb};
        x[e] = __builtin_amdgcn_exp2f(y.x);
        x[e + 1] = __builtin_amdgcn_exp2f(y.y);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        part[e & 3] += x[e];
        p[t][s2][e] = (T)x[e];
      }
      hook(2 * t + s2);
      __builtin_amdgcn_sched_barrier(0);
    }
  float sum = (part[0] + part[1]) + (part[2] + part[3]);
  sum += partner32(sum, half);
  return 1.f / sum;
}

// NW = waves per workgroup, QPW = 32-query blocks per wave (handled one after the other); ceil(S/32 / (NW QPW)) workgroups share
// one (passage, head) and each stages the whole K / V^T of it.  Measured at S = 256: NW = 8 (one workgroup per (passage, head), K/V
// staged once) 131 us per 3072 blocks; NW = 4 (two resident workgroups per CU, staging overlapped but doubled) 147 us: the kernel is
// bound by the L2->LDS fill, so the doubled staging costs more than the overlap buys.
// These kernels move 4 x S x 128 bytes per (passage, head) - Q, K, V^T in, the context out - for work that grows with S^2: below
// S = 256 they are bound by that traffic (per 256 passages x 12 heads: 12.5 / 23.4 / 35.6 / 49.8 us at S = 32 .. 128 = 4.0-4.2 TB/s;
// without the softmax S = 128 still takes 46.9 us: profiling builds -DCAPAMD_ATTN1_ABLATE=1|2); S = 160 / 192 / 224 with their
// 5 / 6 / 7 waves take 70 / 94 / 146 us, 1.4-2.1 x that bound.  QPW = 2 (four waves of two blocks) brings S = 224 to 126 us; on
// three waves S = 160 / 192 get slower (75 / 137 us - the second block's latency chain is exposed), so they keep QPW = 1.
template <int S, int NW, typename T, int QPW = 1>
__global__ __launch_bounds__(64 * NW, QPW == 1 ? 1 : 2) void attention_kernel(AttnArgs a) {   // (QPW = 2: without a bound hipcc spends 330 registers on it - one wave per SIMD)
  using bf16x8 = typename Half<T>::x8;
  using bf16x4 = typename Half<T>::x4;
  constexpr int NT = S / 32;                              // key tiles = 32-query blocks
  constexpr int QB = (NT + NW * QPW - 1) / (NW * QPW);    // workgroups per (passage, head)
  constexpr int NTHR = 64 * NW;
  constexpr int VROW = S * 2 + 8;       // bytes per V^T row in LDS
  __shared__ __attribute__((aligned(16))) char lds[S * 128 + 64 * VROW + S * 4];
  char* Ks = lds;
  char* Vs = lds + S * 128;
  float* madd = reinterpret_cast<float*>(lds + S * 128 + 64 * VROW);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int ph = blockIdx.x / QB, qb = blockIdx.x % QB;   // (passage, head) index, query block
  const int psg = ph / a.heads, head = ph % a.heads;
  const int64_t tok0 = (int64_t)psg * S;
  const int qwave0 = (qb * NW + wave) * QPW;              // the first 32-query slice of the passage this wave owns

  // ---- stage K (swizzled, via LDS-DMA), V^T (padded rows, via registers), additive mask ----
  {
    const int r8 = lane >> 3, p = lane & 7;
#pragma unroll
    for (int c = wave; c < S / 8; c += NW) {              // 8 key rows (1 KiB) per instruction and wave
      const int row = c * 8 + r8;
      const T* src = static_cast<const T*>(a.K) + qk_offset(a, tok0 + row, head * 8 + swz_chunk(row, p));
      __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(Ks + c * 1024), 16, 0, 0);
    }
    const T* vsrc = static_cast<const T*>(a.Vt) + (int64_t)ph * 64 * S;
    for (int c = tid; c < 64 * S / 8; c += NTHR) {
      const int d = c / (S / 8), k8 = c % (S / 8);
      const uint4 x = *reinterpret_cast<const uint4*>(vsrc + d * S + k8 * 8);
      *reinterpret_cast<uint2*>(Vs + d * VROW + k8 * 16) = make_uint2(x.x, x.y);
      *reinterpret_cast<uint2*>(Vs + d * VROW + k8 * 16 + 8) = make_uint2(x.z, x.w);
    }
    for (int k = tid; k < S; k += NTHR)
      madd[k] = a.mask[(int64_t)psg * S + k] != 0 ? 0.f : -3.4028234663852886e38f;  // HF: (1-mask) * finfo.min
  }
  // this wave's Q fragments (B operand): query = qwave*32 + l31, d = (2*ks+half)*8 ..+7
  auto load_q = [&](bf16x8 (&q)[4], int qw) {
    qw = qw < NT ? qw : NT - 1;                           // (a block beyond the passage: a valid row, never computed)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      q[ks] = *reinterpret_cast<const bf16x8*>(static_cast<const T*>(a.Q) + qk_offset(a, tok0 + qw * 32 + l31, head * 8 + 2 * ks + half));
  };
  bf16x8 qf[4];
  load_q(qf, qwave0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // LDS-DMA of K: hipcc does not wait for it at the barrier by itself
  __syncthreads();

  const int lperm = key_of_row(l31);
  // (one block after the other, not unrolled: interleaving two blocks doubles the live registers - 298 at S = 224 - and leaves one wave per SIMD)
#pragma unroll 1
  for (int j = 0; j < QPW; ++j) {
    const int qwave = qwave0 + j;
    if (qwave >= NT) break;                               // (wave-uniform; no barrier below)
    bf16x8 qn[4];
    if (QPW > 1) load_q(qn, qwave + 1);                   // the next block's fragments arrive under this block's work
    // ---- scores^T tiles: lane <- query l31, keys 32t + 16*(r>>3) + 8*half + (r&7) (key_of_row) ----
    bf16x8 pf[NT][2];
    float inv;
    {
      f32x16 sc[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        sc[t] = scores_init_permuted<NT>(madd + t * 32 + 8 * half);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int row = t * 32 + lperm;
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + row * 128 + swz_chunk(row, 2 * ks + half) * 16);
          sc[t] = Half<T>::mfma(kf, qf[ks], sc[t]);
        }
      }
      // ---- exact softmax over the S keys of this lane's query ----
#if defined(CAPAMD_ATTN1_ABLATE) && (CAPAMD_ATTN1_ABLATE & 2)   // profiling build: no softmax (plain conversion)
      inv = 1.f;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) pf[t][e >> 3][e & 7] = (T)sc[t][e];
#else
      inv = softmax_pack<NT, T>(sc, pf);
#endif
    }

    // ---- ctx^T = V^T · P^T : out[dt] lane <- query l31, d = 32dt + 8*(r>>2) + 4*half + (r&3) ----
    f32x16 out[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) out[dt][r] = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int kb = (32 * t + 16 * s2 + 8 * half) * 2;  // byte offset of keys 32t + 16 s2 + 8 half .. + 7 in a V^T row (8-byte aligned rows)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const char* vr = Vs + (dt * 32 + l31) * VROW + kb;
          const uint2 lo = *reinterpret_cast<const uint2*>(vr);
          const uint2 hi = *reinterpret_cast<const uint2*>(vr + 8);
          const uint4 raw = make_uint4(lo.x, lo.y, hi.x, hi.y);
          const bf16x8 vf = __builtin_bit_cast(bf16x8, raw);
          out[dt] = Half<T>::mfma(vf, pf[t][s2], out[dt]);
        }
      }
    const int64_t ctok = tok0 + qwave * 32 + l31;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        bf16x4 o = {scale_round<T>(out[dt][g4 * 4 + 0], inv), scale_round<T>(out[dt][g4 * 4 + 1], inv),
                    scale_round<T>(out[dt][g4 * 4 + 2], inv), scale_round<T>(out[dt][g4 * 4 + 3], inv)};
        *reinterpret_cast<bf16x4*>(ctx_slot<T>(a, ctok, head, dt, g4, half)) = o;
      }
    if (QPW > 1) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) qf[ks] = qn[ks];
    }
  }
}

// ---- persistent, double-buffered variant for S = 256 ------------------------------------------------------------
// One workgroup per CU walks the (passage, head) items; K and V^T of item i+1 are fetched by LDS-DMA into the second
// LDS buffer while item i is computed, so the ~4.8k-cycle fill of an item (96 KiB at ~20 B/clk/CU) hides behind the
// ~6k cycles of MFMA + softmax instead of preceding them.  V^T is staged by DMA as well (linear rows of 512 B, the
// 16-byte chunks XOR-swizzled by (d & 31) on the source side; 2-way bank conflict on the ds_read_b64 pairs instead
// of the padded rows of the one-shot kernel), the additive mask through registers.
template <typename T>
__global__ __launch_bounds__(512) void attention_persistent_kernel(AttnArgs a, int n_items) {
  using bf16x8 = typename Half<T>::x8;
  using bf16x4 = typename Half<T>::x4;
  constexpr int S = 256, NT = 8, KBYTES = S * 128, VBYTES = 64 * S * 2, BUF = KBYTES + VBYTES + S * 4;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;

  auto stage = [&](int item, int buf) {
    char* Ks = lds + buf * BUF;
    char* Vs = Ks + KBYTES;
    const int psg = item / a.heads, head = item % a.heads;
    const int64_t tok0 = (int64_t)psg * S;
    {
      const int r8 = lane >> 3, p = lane & 7;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int row = (wave * 4 + t) * 8 + r8;
        const T* src = static_cast<const T*>(a.K) + qk_offset(a, tok0 + row, head * 8 + swz_chunk(row, p));
        __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(Ks + (wave * 4 + t) * 1024), 16, 0, 0);
      }
    }
    {
      const T* vsrc = static_cast<const T*>(a.Vt) + (int64_t)item * 64 * S;
      const int r2 = lane >> 5, pc = lane & 31;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int d = (wave * 4 + t) * 2 + r2;
        const T* src = vsrc + d * S + ((pc ^ (d & 31)) * 8);
        __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(Vs + (wave * 4 + t) * 1024), 16, 0, 0);
      }
    }
  };
  auto load_q = [&](int item, bf16x8 (&qf)[4]) {
    const int psg = item / a.heads, head = item % a.heads;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      qf[ks] = *reinterpret_cast<const bf16x8*>(static_cast<const T*>(a.Q) + qk_offset(a, (int64_t)psg * S + wave * 32 + l31, head * 8 + 2 * ks + half));
  };
  auto write_mask = [&](int item, int buf) {
    float* madd = reinterpret_cast<float*>(lds + buf * BUF + KBYTES + VBYTES);
    const int psg = item / a.heads;
    if (tid < S) madd[tid] = a.mask[(int64_t)psg * S + tid] != 0 ? 0.f : -3.4028234663852886e38f;
  };

  int item = blockIdx.x;
  if (item >= n_items) return;
  bf16x8 qf[4];
  stage(item, 0);
  load_q(item, qf);
  write_mask(item, 0);
  int buf = 0;
  bool stores_pending = false;
  for (;; item += gridDim.x, buf ^= 1) {
    // the DMA of this item (issued one iteration ago) is older than the previous item's 8 ctx stores per wave
    if (stores_pending) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int nxt = item + gridDim.x;
    const bool more = nxt < n_items;
    if (more) {
      stage(nxt, buf ^ 1);
      write_mask(nxt, buf ^ 1);
    }
    const char* Ks = lds + buf * BUF;
    const char* Vs = Ks + KBYTES;
    const float* madd = reinterpret_cast<const float*>(Vs + VBYTES);

    f32x16 sc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      sc[t] = scores_init_from_mask(madd + t * 32 + 4 * half);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int row = t * 32 + l31;
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + row * 128 + swz_chunk(row, 2 * ks + half) * 16);
        sc[t] = Half<T>::mfma(kf, qf[ks], sc[t]);
      }
    }
    if (more) load_q(nxt, qf);  // the Q fragments are dead after the score MFMAs: refill them for the next item now
    const float inv = softmax_inplace<NT>(sc);

    f32x16 out[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) out[dt][r] = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8 pf;
#pragma unroll
        for (int e = 0; e < 8; ++e) pf[e] = (T)sc[t][8 * s2 + e];
        const int c0 = 4 * t + 2 * s2;  // 16-byte chunk holding keys 32t + 16 s2 + {0..7}; this lane needs the half * 8 B part
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const int d = dt * 32 + l31;
          const char* vr = Vs + d * 512 + half * 8;
          const uint2 lo = *reinterpret_cast<const uint2*>(vr + ((c0 ^ (d & 31)) << 4));
          const uint2 hi = *reinterpret_cast<const uint2*>(vr + (((c0 + 1) ^ (d & 31)) << 4));
          const bf16x8 vf = __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
          out[dt] = Half<T>::mfma(vf, pf, out[dt]);
        }
      }
    {
      const int psg = item / a.heads, head = item % a.heads;
      const int64_t ctok = (int64_t)psg * S + wave * 32 + l31;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          bf16x4 o = {scale_round<T>(out[dt][g4 * 4 + 0], inv), scale_round<T>(out[dt][g4 * 4 + 1], inv), scale_round<T>(out[dt][g4 * 4 + 2], inv),
                      scale_round<T>(out[dt][g4 * 4 + 3], inv)};
          *reinterpret_cast<bf16x4*>(ctx_slot<T>(a, ctok, head, dt, g4, half)) = o;
        }
    }
    if (!more) break;
    stores_pending = true;
  }
}

// ---- S = 256, two independent 4-wave workgroups per CU (the default) ----------------------------------------------------------------
// What bounds the 8-wave kernel above is not one resource but their SUM: its waves run in lockstep (one barrier per item), so
// all of them read K fragments, then all of them do softmax VALU work, then all of them read V fragments - per item about 2 k
// cycles of LDS reads for QK^T, 4.8 k of VALU, 4 k of LDS reads for P V (the b64 V reads are 2-way bank conflicted), MFMA pipe busy
// 20 %.  This kernel is laid out so that the phases of different waves overlap instead:
//   * 4 waves per workgroup, 2 workgroups per CU (64 KiB of LDS each, 256 registers per wave): the two workgroups share nothing
//     and drift apart, so one's softmax sits beside the other's MFMAs / LDS reads;
//   * a wave owns 64 queries as two 32-query blocks handled one after the other (scores -> softmax -> P V, twice): the registers of
//     one block at a time (128 score registers) fit the 256-register budget of two waves per SIMD;
//   * the key order inside a 32-key tile is permuted on the K side (A-operand row i holds key i with bits 2 and 3 swapped), so the
//     8 probabilities a lane contributes to one P V MFMA are 8 CONSECUTIVE keys: the matching V^T fragment is one 16-byte chunk
//     (ds_read_b128, conflict-free under the chunk swizzle) instead of two conflicted b64 halves;
//   * K and V^T have one buffer each, refilled by LDS-DMA as soon as the workgroup is done with it: K(i+1) streams in under
//     P V of block B of item i, V^T(i+1) under the scores of block A of item i+1.  Three barriers per item.
// 16 bytes per lane from global memory straight into LDS (64 lanes -> 1 KiB at `lds_addr`, a wave-uniform LDS byte address), issued
// from inline assembly on purpose: hipcc puts an `s_waitcnt vmcnt(0)` in front of the first LDS read after any LDS-DMA it has seen
// issued (it cannot tell which LDS bytes are in flight), which turns a refill that should run under the next phase into a stall at
// the top of it.  The kernel below waits for its DMAs explicitly where it needs them.  M0 is saved and restored around the issue.
__device__ __forceinline__ void lds_dma16(const void* base, uint32_t voff, uint32_t lds_addr) {   // base: wave-uniform; voff: this lane's byte offset
  uint32_t saved;
  lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);   // (uniform already; keeps the value in an SGPR whatever the compiler did with the arithmetic before)
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(saved)
               : "s"(lds_addr), "v"(voff), "s"(base)
               : "memory");
}

// builder-side phase timing (scripts/ubench/attn_trace.hip defines CAPAMD_ATTN_TRACE): thread 0 of every workgroup stamps s_memtime
#ifdef CAPAMD_ATTN_TRACE
__device__ unsigned long long* g_attn_trace;
#define ATTN_STAMP_DECL int stamp_it = 0
#define ATTN_STAMP(k) do { if (tid == 0 && stamp_it < 8) g_attn_trace[(blockIdx.x * 8 + stamp_it) * 16 + (k)] = __builtin_readcyclecounter(); } while (0)
#define ATTN_STAMP_NEXT ++stamp_it
#else
#define ATTN_STAMP_DECL
#define ATTN_STAMP(k)
#define ATTN_STAMP_NEXT
#endif

// QKCM / CTXCM: the activation layouts (AttnArgs::qk_cm, ctx_cm) as compile-time constants - as run-time flags every address
// computation exists twice and the spare copies cost registers this kernel does not have
template <typename T, bool QKCM, bool CTXCM>
__global__ __launch_bounds__(256, 2) void attention_s256_kernel(AttnArgs a, int n_items) {
  using bf16x8 = typename Half<T>::x8;
  using bf16x4 = typename Half<T>::x4;
  constexpr int S = 256, NT = 8, KBYTES = S * 128, VBYTES = 64 * S * 2;
  __shared__ __attribute__((aligned(16))) char Ks[KBYTES];
  __shared__ __attribute__((aligned(16))) char Vs[VBYTES];
  __shared__ __attribute__((aligned(16))) float madd[S];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int lperm = key_of_row(l31);
  const int Hc = a.H >> 3;                                              // 16-byte chunks per activation row
  const uint32_t ks_addr = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)Ks), vs_addr = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)Vs);

  // byte offset of chunk `chunk` of local token row `row` (0..255) from the passage's first row, either activation layout
  // (a passage starts on a multiple of 32 rows, so its chunk-major image starts at the same element offset tok0 * H)
  auto row_off = [&](int row, int chunk) -> uint32_t {
    return (uint32_t)(QKCM ? (((row >> 5) * Hc + chunk) * 32 + (row & 31)) * 16 : (row * Hc + chunk) * 16);
  };
  // The refills are issued ONE instruction at a time between the MFMAs of the phase they hide behind (dma_k / dma_v / load_q1, t = 0..7
  // per wave), not as a burst: a wave that issues 8 KiB of LDS-DMA back to back sits in the issue until the CU's ~20-25 B/clk fill
  // path has taken it (measured: 1.5-2.5 k cycles per burst, 5.8 k of a 21.6 k-cycle item).
  auto dma_k = [&](int psg, int head, int t) {   // 8 rows x 128 B of the K image: rows (wave * 8 + t) * 8 .. + 7
    const char* kb = static_cast<const char*>(a.K) + (int64_t)psg * S * a.H * 2;
    const int row = (wave * 8 + t) * 8 + (lane >> 3);
    lds_dma16(kb, row_off(row, head * 8 + swz_chunk(row, lane & 7)), ks_addr + (wave * 8 + t) * 1024);
  };
  auto dma_v = [&](int item, int t) {            // 2 rows x 512 B of the V^T image: d = (wave * 8 + t) * 2 + {0, 1}
    const char* vb = static_cast<const char*>(a.Vt) + (int64_t)item * 64 * S * 2;
    const int d = (wave * 8 + t) * 2 + half;
    lds_dma16(vb, (uint32_t)(d * (S * 2) + ((l31 ^ (d & 31)) << 4)), vs_addr + (wave * 8 + t) * 1024);
  };
  auto load_q1 = [&](int psg, int head, bf16x8 (&qf)[2][4], int j) {   // Q fragment j = 4 x + ks of this wave's two query blocks
    const char* qb = static_cast<const char*>(a.Q) + (int64_t)psg * S * a.H * 2;
    const int x = j >> 2, ks = j & 3;
    qf[x][ks] = *reinterpret_cast<const bf16x8*>(qb + row_off(wave * 64 + x * 32 + l31, head * 8 + 2 * ks + half));
  };
  // scores^T of one 32-query block against all 256 keys; hook(0..3) runs between the tile pairs
  auto qk_block = [&](const bf16x8 (&q)[4], f32x16 (&sc)[NT], auto&& hook) {
    int kbase = lperm * 128 + half * 16, ksw = ((lperm >> 1) & 7) << 4;   // K image: row * 128 + ((2 ks + half) ^ ((row >> 1) & 7)) * 16
    asm volatile("" : "+v"(kbase), "+v"(ksw));                          // (re-derived per block: not hoisted out of the item loop)
    // The K fragments are requested a whole tile pair AHEAD of their MFMA: asked for right before it - what hipcc makes of the plain
    // loop: two fragment registers, an `s_waitcnt lgkmcnt` in front of every MFMA - a pair of MFMAs (64 cycles of the matrix pipe) waits
    // for an LDS round trip of twice that.  ONE set of eight fragment registers: fragment i of the next pair is requested right behind the
    // MFMA that consumed fragment i of this one (eight MFMAs = 256 cycles before its use), and the mask rows that initialise the next
    // pair's accumulators travel with them, into the accumulators themselves; the issue order (M R R ..) is pinned with sched_group_barrier.
    bf16x8 kf[8];
    auto load_frag = [&](int t, int i) {   // fragment i = 2 ks + u of the tile pair (t, t + 1)
      kf[i] = *reinterpret_cast<const bf16x8*>(Ks + (t + (i & 1)) * 4096 + ((kbase + (i >> 1) * 32) ^ ksw));
    };
#pragma unroll
    for (int i = 0; i < 8; ++i) load_frag(0, i);
    sc[0] = scores_init_permuted<NT>(madd + 8 * half);
    sc[1] = scores_init_permuted<NT>(madd + 32 + 8 * half);
    __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);
#pragma unroll
    for (int t = 0; t < NT; t += 2) {
      if (t + 2 < NT) {
        sc[t + 2] = scores_init_permuted<NT>(madd + (t + 2) * 32 + 8 * half);
        sc[t + 3] = scores_init_permuted<NT>(madd + (t + 3) * 32 + 8 * half);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        sc[t + (i & 1)] = Half<T>::mfma(kf[i], q[i >> 1], sc[t + (i & 1)]);
        if (t + 2 < NT) load_frag(t + 2, i);
      }
      if (t + 2 < NT) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // MFMA i
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);    // the next pair's fragment i + one of its eight mask-row reads
        }
      }
      hook(t >> 1);
    }
  };
  // ctx^T = V^T . P^T of one query block: out[dt] lane <- query l31, d = 32 dt + 8 (r >> 2) + 4 half + (r & 3); scaled, rounded, stored;
  // hook(0..15) runs between the (t, s2) steps
  auto pv_block = [&](const bf16x8 (&p)[NT][2], float inv, int64_t ctok, int head, auto&& hook) {
    f32x16 out[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) out[dt][r] = 0.f;
    int vbase = l31 * 512 + half * 16, vsw = l31 << 4;   // V^T image: d * 512 + ((4 t + 2 s2 + half) ^ (d & 31)) * 16
    asm volatile("" : "+v"(vbase), "+v"(vsw));
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int o = (vbase + (4 * t + 2 * s2) * 16) ^ vsw;
        const bf16x8 v0 = *reinterpret_cast<const bf16x8*>(Vs + o);
        const bf16x8 v1 = *reinterpret_cast<const bf16x8*>(Vs + o + 32 * 512);
        out[0] = Half<T>::mfma(v0, p[t][s2], out[0]);
        out[1] = Half<T>::mfma(v1, p[t][s2], out[1]);
        hook(2 * t + s2);
      }
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        bf16x4 o = {scale_round<T>(out[dt][g4 * 4 + 0], inv), scale_round<T>(out[dt][g4 * 4 + 1], inv), scale_round<T>(out[dt][g4 * 4 + 2], inv),
                    scale_round<T>(out[dt][g4 * 4 + 3], inv)};
        T* c = static_cast<T*>(a.ctx);
        if (CTXCM) c += (((ctok >> 5) * Hc + head * 8 + dt * 4 + g4) * 32 + (ctok & 31)) * 8 + 4 * half;   // ctx_slot, chunk-major
        else c += ctok * a.H + head * 64 + dt * 32 + 8 * g4 + 4 * half;
        *reinterpret_cast<bf16x4*>(c) = o;
      }
  };
  // The compiler's own wait for the Q loads would sit at their first use - the top of the next item, AFTER the V^T DMAs have been
  // issued behind them, where (the counter being in order) it would wait for those as well.  Using the registers right after the
  // explicit wait pins the compiler's wait to that point, where it is free.
  auto touch_q = [&](bf16x8 (&qf)[2][4]) {
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(qf[x][ks]));
  };
  auto no_hook = [](int) {};

  int item = blockIdx.x;
  if (item >= n_items) return;
  bf16x8 qf[2][4];
  {
    const int psg = item / a.heads, head = item % a.heads;
    const int64_t mval = a.mask[(int64_t)psg * S + tid];
#pragma unroll
    for (int t = 0; t < 8; ++t) dma_k(psg, head, t);
#pragma unroll
    for (int j = 0; j < 4; ++j) load_q1(psg, head, qf, j);
    madd[tid] = mval != 0 ? 0.f : -3.4028234663852886e38f;   // HF: (1 - mask) * finfo.min
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(qf[0][ks]));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  ATTN_STAMP_DECL;
  // One item; MORE (compile time): another item follows and its operands are fetched under this one's phases.  Where what streams in
  // (each refill one instruction at a time between the instructions of the phase that covers it; the Q fragments, which occupy
  // registers from the moment they are requested, only in the two P V passes, where 100 registers are free):
  //     scores A + softmax A : V^T(item)            P V A : Q block B (item)
  //     softmax B            : K(next)              P V B : Q block A (next)
  auto do_item = [&](auto more_tag) {
    constexpr bool MORE = decltype(more_tag)::value;
    // here: K(item), Q block A (item), madd(item) are in place for every wave; the V^T buffer is free
    ATTN_STAMP(0);
    const int nxt = MORE ? item + (int)gridDim.x : item;
    const int psg = item / a.heads, head = item % a.heads;
    const int npsg = nxt / a.heads, nhead = nxt % a.heads;
    const int64_t ctok = (int64_t)psg * S + wave * 64 + l31;
    int64_t mnext = 1;   // the next item's mask element, kept raw until the madd buffer is free
    if (MORE) mnext = a.mask[(int64_t)npsg * S + tid];
    bf16x8 p[NT][2];
    float inv;
    {
      f32x16 sc[NT];
      qk_block(qf[0], sc, [&](int i) { dma_v(item, i); });
      inv = softmax_pack<NT, T>(sc, p, [&](int i) {
        if ((i & 3) == 0) dma_v(item, 4 + (i >> 2));
      });
    }
    ATTN_STAMP(1);
    // B1: every wave's V^T(item) slices have landed
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    ATTN_STAMP(2);
    pv_block(p, inv, ctok, head, [&](int i) {
      if ((i & 3) == 1) load_q1(psg, head, qf, 4 + (i >> 2));
    });
    __builtin_amdgcn_sched_barrier(0);
    ATTN_STAMP(3);
    {
      f32x16 sc[NT];
      qk_block(qf[1], sc, no_hook);
      ATTN_STAMP(4);
      // B2: every wave is done with K and madd -> the next item's mask row goes in, K(next) streams in under the softmax of block B
      __builtin_amdgcn_s_barrier();
      if (MORE) madd[tid] = mnext != 0 ? 0.f : -3.4028234663852886e38f;
      inv = softmax_pack<NT, T>(sc, p, [&](int i) {
        if (MORE && (i & 1) == 0) dma_k(npsg, nhead, i >> 1);
      });
    }
    ATTN_STAMP(5);
    pv_block(p, inv, ctok + 32, head, [&](int i) {
      if (MORE && (i & 3) == 1) load_q1(npsg, nhead, qf, i >> 2);
    });
    ATTN_STAMP(6);
    if (MORE) {
      // B3: every wave is done with V^T; K(next) and Q block A (next) have landed (everything but this block's 8 ctx stores); madd(next) is written
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(qf[0][ks]));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    ATTN_STAMP(7);
    ATTN_STAMP_NEXT;
  };
  for (; item + (int)gridDim.x < n_items; item += gridDim.x) do_item(std::true_type{});
  do_item(std::false_type{});
}

// ---- last layer: attention for the [CLS] query only -----------------------------------------------------------------
// After the last encoder layer only row 0 of every passage is read (the pooler, ptBERTMaxP.py:82 -> [:, 1] of the
// classifier on the pooled [CLS] state), and every operation after the attention is row-wise: the last layer's attention
// output is needed for ONE query per (passage, head), and its output projection / FFN only on those rows (bert.hip).
// One wave per (passage, head): lane j scores keys j, j+64, ...; exact softmax by wave reductions; lane d accumulates
// output dimension d over the keys.  Same arithmetic as the full kernels: fp32 dot products of the 16-bit Q (pre-scaled by
// 1/8) and K, additive -FLT_MAX mask, probabilities rounded to the 16-bit type before the P V product, fp32 accumulation.
template <typename T>
__global__ __launch_bounds__(64) void cls_attention_kernel(AttnArgs a, int S, T* __restrict__ ctx_cls) {
  using bf16x8 = typename Half<T>::x8;
  __shared__ float prob[512];
  const int lane = threadIdx.x, ph = blockIdx.x;
  const int psg = ph / a.heads, head = ph % a.heads;
  const int64_t tok0 = (int64_t)psg * S;
  float q[64];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(static_cast<const T*>(a.Q) + qk_offset(a, tok0, head * 8 + c));
#pragma unroll
    for (int e = 0; e < 8; ++e) q[c * 8 + e] = (float)v[e];
  }
  float sc[8];
  float mx = -3.4028234663852886e38f;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int j = lane + 64 * r;
    float s = -3.4028234663852886e38f;
    if (j < S) {
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const bf16x8 k = *reinterpret_cast<const bf16x8*>(static_cast<const T*>(a.K) + qk_offset(a, tok0 + j, head * 8 + c));
#pragma unroll
        for (int e = 0; e < 8; ++e) acc = __builtin_fmaf(q[c * 8 + e], (float)k[e], acc);
      }
      s = acc + (a.mask[tok0 + j] != 0 ? 0.f : -3.4028234663852886e38f);
    }
    sc[r] = s;
    mx = fmaxf(mx, s);
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float sum = 0.f;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int j = lane + 64 * r;
    const float p = j < S ? __builtin_amdgcn_exp2f((sc[r] - mx) * 1.4426950408889634f) : 0.f;
    sum += p;
    if (j < S) prob[j] = (float)(T)p;   // the P V product consumes 16-bit probabilities, as in the MFMA kernels
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) sum += __shfl_xor(sum, o, 64);
  __syncthreads();
  const T* vrow = static_cast<const T*>(a.Vt) + ((int64_t)ph * 64 + lane) * S;   // V^T row d = lane
  float out = 0.f;
  for (int j0 = 0; j0 < S; j0 += 8) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(vrow + j0);
#pragma unroll
    for (int e = 0; e < 8; ++e) out = __builtin_fmaf(prob[j0 + e], (float)v[e], out);
  }
  ctx_cls[(int64_t)psg * a.H + head * 64 + lane] = (T)(out / sum);
}

// ---- long passages: S = 384 and 512 (BERT's position table ends at 512) ----------------------------------------------------
// The kernels above keep the scores of one query against ALL keys in registers (S / 2 VGPRs per lane), which stops at
// S = 256.  Here the keys are walked in chunks of 64 with the running-maximum recurrence - the same exact softmax,
// associated differently:
//     m' = max(m, max_chunk);  alpha = 2^((m - m') log2e);  l = l alpha + sum 2^((s - m') log2e);  out = out alpha + V^T P
// One workgroup per (passage, head), one wave per 32 queries (12 / 16 waves), K and V^T of the whole passage in LDS
// (132 KiB at S = 512).
template <int S, typename T>
__global__ __launch_bounds__(S * 2) void attention_long_kernel(AttnArgs a) {
  using bf16x8 = typename Half<T>::x8;
  using bf16x4 = typename Half<T>::x4;
  constexpr int NW = S / 32, NTHR = 64 * NW, NC = 2, NCHUNK = S / (32 * NC);
  constexpr int VROW = S * 2 + 8;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* Ks = lds;
  char* Vs = lds + S * 128;
  float* madd = reinterpret_cast<float*>(lds + S * 128 + 64 * VROW);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int ph = blockIdx.x;
  const int psg = ph / a.heads, head = ph % a.heads;
  const int64_t tok0 = (int64_t)psg * S;
  {
    const int r8 = lane >> 3, p = lane & 7;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int row = (wave * 4 + t) * 8 + r8;
      const T* src = static_cast<const T*>(a.K) + qk_offset(a, tok0 + row, head * 8 + swz_chunk(row, p));
      __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(Ks + (wave * 4 + t) * 1024), 16, 0, 0);
    }
    const T* vsrc = static_cast<const T*>(a.Vt) + (int64_t)ph * 64 * S;
    for (int c = tid; c < 64 * S / 8; c += NTHR) {
      const int d = c / (S / 8), k8 = c % (S / 8);
      const uint4 x = *reinterpret_cast<const uint4*>(vsrc + d * S + k8 * 8);
      *reinterpret_cast<uint2*>(Vs + d * VROW + k8 * 16) = make_uint2(x.x, x.y);
      *reinterpret_cast<uint2*>(Vs + d * VROW + k8 * 16 + 8) = make_uint2(x.z, x.w);
    }
    for (int k = tid; k < S; k += NTHR) madd[k] = a.mask[tok0 + k] != 0 ? 0.f : -3.4028234663852886e38f;  // HF: (1-mask) * finfo.min
  }
  bf16x8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
    qf[ks] = *reinterpret_cast<const bf16x8*>(static_cast<const T*>(a.Q) + qk_offset(a, tok0 + wave * 32 + l31, head * 8 + 2 * ks + half));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // LDS-DMA of K: hipcc does not wait for it at the barrier by itself
  __syncthreads();

  constexpr float kLog2e = 1.4426950408889634f;
  f32x16 out[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[dt][r] = 0.f;
  float m_run = -3.4028234663852886e38f, l_run = 0.f;
#pragma unroll 1
  for (int c = 0; c < NCHUNK; ++c) {
    f32x16 sc[NC];
#pragma unroll
    for (int t = 0; t < NC; ++t) {
      const int tt = c * NC + t;
      sc[t] = scores_init_from_mask(madd + tt * 32 + 4 * half);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int row = tt * 32 + l31;
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + row * 128 + swz_chunk(row, 2 * ks + half) * 16);
        sc[t] = Half<T>::mfma(kf, qf[ks], sc[t]);
      }
    }
    float mx = m_run;
#pragma unroll
    for (int t = 0; t < NC; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[t][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    // a fully masked prefix keeps m at -FLT_MAX: (m - m') = 0 there, alpha = 1 on zeros - still exact
    const float alpha = __builtin_amdgcn_exp2f((m_run - mx) * kLog2e);
    const float nb = -mx * kLog2e;
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < NC; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[t][r], kLog2e, nb));
        sc[t][r] = p;
        sum += p;
      }
    sum += __shfl_xor(sum, 32, 64);
    l_run = l_run * alpha + sum;
    m_run = mx;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) out[dt] = out[dt] * alpha;
#pragma unroll
    for (int t = 0; t < NC; ++t)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8 pf;
#pragma unroll
        for (int e = 0; e < 8; ++e) pf[e] = (T)sc[t][8 * s2 + e];
        const int kb = (32 * (c * NC + t) + 16 * s2 + 4 * half) * 2;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const char* vr = Vs + (dt * 32 + l31) * VROW + kb;
          const uint2 lo = *reinterpret_cast<const uint2*>(vr);
          const uint2 hi = *reinterpret_cast<const uint2*>(vr + 16);
          const bf16x8 vf = __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
          out[dt] = Half<T>::mfma(vf, pf, out[dt]);
        }
      }
  }
  const float inv = 1.f / l_run;
  const int64_t ctok = tok0 + wave * 32 + l31;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      bf16x4 o = {scale_round<T>(out[dt][g4 * 4 + 0], inv), scale_round<T>(out[dt][g4 * 4 + 1], inv), scale_round<T>(out[dt][g4 * 4 + 2], inv),
                  scale_round<T>(out[dt][g4 * 4 + 3], inv)};
      *reinterpret_cast<bf16x4*>(ctx_slot<T>(a, ctok, head, dt, g4, half)) = o;
    }
}

}  // namespace capamd
