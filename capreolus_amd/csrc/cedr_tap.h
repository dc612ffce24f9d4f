// CEDR-KNRM's reading of the encoder: for one hidden state of a micro-batch of passages, the masked cosine
// similarity matrix between the query tokens and the document tokens of each passage and its kernel pooling over the document
// axis.  Reference: CEDRKNRM_Class.masked_simmats / _cos_simmat / knrm, capreolus/reranker/CEDRKNRM.py:83-136.
//
//   sequence positions 1 .. A (A = maxqlen + 1; [CLS] is dropped, :114)  = query rows   (mask & segment 0, :98-100)
//   sequence positions 1 .. S-1                                          = document columns (mask & segment 1, :102-104)
//   sim[a][b] = x_a . x_b / ((|x_a| + 1e-9)(|x_b| + 1e-9)) * qmask[a] * dmask[b]                           (:86-93)
//   pk[k][a]  = sum_b exp(-0.5 (sim - mu_k)^2 / sigma_k^2) * dmask[b] * qmask0[a]   (:123-130; qmask0 = the query mask of the
//               document's FIRST passage, handed in per passage so that passages can be regrouped by length)
// The per-passage sums are added over a document's passages, clamped, logged and summed over the query by cedr.hip.
//
// One workgroup per passage: the A x S dot products on v_mfma_f32_32x32x16 straight from the 16-bit hidden state in global
// memory (operands are the stored activations: one product, fp32 accumulate; the row norms are accumulated from the same operand
// loads), similarities into LDS, then one thread per (query row, slice of columns) for the K exponentials.
#pragma once
#include "bert_gemm.h"

namespace capamd {

constexpr int kCedrMaxA = 32;
constexpr int kCedrMaxK = 11;
constexpr int kCedrMaxTpr = 32;

struct CedrTap {          // what a CEDR-KNRM call asks of encode_passages
  int A, K, n_sel;
  const int* layers;      // host array [n_sel]: hidden states to pool (0 = embedding output .. L), in the caller's order (slot i of pk)
  const float* mu;        // device [K]
  const float* sigma;     // device [K]
  const float* qmask0;    // device [NP][A]: the query mask of the first passage of each passage's document (CEDRKNRM.py:123)
  float* pk;              // device [n_sel][NP][K][A]
  float* cls;             // device [NP][H]: the last hidden state's [CLS] row, fp32
  int64_t NP;
};

// kernel pooling of one passage's similarities (LDS) over the document axis: thread -> (query row, slice of the document columns)
__device__ __forceinline__ void cedr_pool_phase(const float* sims, const float* kc, const float* qm0, const float* dm, float* partial, int S, int A,
                                                int K, int64_t pg, float* __restrict__ pk) {
  const int tid = threadIdx.x;
  int tpr = 256 / A;
  if (tpr > kCedrMaxTpr) tpr = kCedrMaxTpr;
  const int a = tid / tpr, sub = tid - a * tpr;
  float acc[kCedrMaxK];
#pragma unroll
  for (int k = 0; k < kCedrMaxK; ++k) acc[k] = 0.f;
  if (a < A && qm0[a] != 0.f) {
    for (int pos = 1 + sub; pos < S; pos += tpr)
      if (dm[pos] != 0.f) {
        const float s = sims[a * S + pos];
        const float4* kc4 = reinterpret_cast<const float4*>(kc);
#pragma unroll
        for (int k2 = 0; k2 < (kCedrMaxK + 1) / 2; ++k2) {
          const float4 c = kc4[k2];
          const float a0 = s - c.x, a1 = s - c.z;
          acc[2 * k2] += __builtin_amdgcn_exp2f(a0 * a0 * c.y);
          if (2 * k2 + 1 < kCedrMaxK) acc[2 * k2 + 1] += __builtin_amdgcn_exp2f(a1 * a1 * c.w);
        }
      }
  }
  if (a < A) {
#pragma unroll
    for (int k = 0; k < kCedrMaxK; ++k) partial[(a * tpr + sub) * (kCedrMaxK + 1) + k] = acc[k];
  }
  __syncthreads();
  for (int i = tid; i < A * K; i += 256) {
    const int k = i / A, aa = i - k * A;
    float s = 0.f;
    for (int u = 0; u < tpr; ++u) s += partial[(aa * tpr + u) * (kCedrMaxK + 1) + k];
    pk[(pg * K + k) * A + aa] = s;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void cedr_pool_kernel(const T* __restrict__ x, const int64_t* __restrict__ mask,
                                                        const int64_t* __restrict__ seg, const float* __restrict__ qmask0, int64_t p0, int S,
                                                        int H, int A, int K,
                                                        const float* __restrict__ mu, const float* __restrict__ sigma,
                                                        float* __restrict__ pk) {
  typedef typename Half<T>::x8 x8;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* rn = reinterpret_cast<float*>(smem_raw);                 // [4][32] query-row norms, one copy per wave (+ padding up to S floats)
  float* sims = rn + (S < 4 * kCedrMaxA ? 4 * kCedrMaxA : S);     // [A][S]
  float* kc = sims + A * S;                                       // [12] (mu, coefficient) pairs
  float* qm = kc + 2 * (kCedrMaxK + 1);                           // [A] this passage's query mask
  float* qm0 = qm + kCedrMaxA;                                    // [A] the query mask of the document's first passage
  float* dm = qm0 + kCedrMaxA;                                    // [S] document mask
  float* partial = dm + S;                                        // [256][K + 1]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int p = blockIdx.x;
  const int64_t pg = p0 + p;
  const T* xp = x + (int64_t)p * S * H;
  const int64_t* mk = mask + (int64_t)p * S;
  const int64_t* sg = seg + (int64_t)p * S;

  for (int i = tid; i < S; i += 256) dm[i] = (mk[i] != 0 && sg[i] == 1) ? 1.f : 0.f;   // ([CLS] is segment 0: never a document column)
  if (tid < A) {
    qm[tid] = (1 + tid < S && mk[1 + tid] != 0 && sg[1 + tid] == 0) ? 1.f : 0.f;
    qm0[tid] = qmask0[pg * A + tid] != 0.f ? 1.f : 0.f;
  }
  if (tid < kCedrMaxK + 1) {
    const float m = tid < K ? mu[tid] : 0.f, s = tid < K ? sigma[tid] : 1.f;
    kc[2 * tid] = m;
    kc[2 * tid + 1] = tid < K ? (-0.5f * 1.4426950408889634f) / (s * s) : 0.f;
  }
  __syncthreads();
  // dot products: query rows (sequence positions 1 + m) x document columns (sequence positions 32 t + n).  The squared row norms
  // come from the matrix pipe too: the diagonal of B B^T (and of A A^T, once per wave) accumulated next to A B^T from the same
  // operand registers - no second pass over the hidden state, no VALU work in the K loop.
  {
    const int m = lane & 31, half = lane >> 5;
    const T* ap = xp + (int64_t)((m < A && 1 + m < S) ? 1 + m : 0) * H + 8 * half;   // rows beyond A: a valid row, output never read
    float* qn = rn + wave * kCedrMaxA;                                                // this wave's copy of the query-row norms
    // accumulator register of this lane that holds the diagonal element (n, n) of a 32x32 product, if its half-wave holds it at all
    const int di = ((m >> 3) << 2) | (m & 3);
    const bool dmine = half == ((m >> 2) & 1);
    auto diagonal = [&](const f32x16& g) {
      float v = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) v = (dmine && i == di) ? g[i] : v;
      return v + __shfl_xor(v, 32, 64);
    };
    bool first = true;
    for (int t = wave; t * 32 < S; t += 4) {
      const int pos = t * 32 + m;
      const T* bp = xp + (int64_t)(pos < S ? pos : 0) * H + 8 * half;
      f32x16 c = {0}, gb = {0}, ga = {0};
      if (first) {
        for (int k0 = 0; k0 < H; k0 += 64) {          // H is a multiple of 64: four K steps per trip, their eight operand loads first
          x8 av[4], bv[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            av[u] = *reinterpret_cast<const x8*>(ap + k0 + 16 * u);
            bv[u] = *reinterpret_cast<const x8*>(bp + k0 + 16 * u);
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            c = Half<T>::mfma(av[u], bv[u], c);
            gb = Half<T>::mfma(bv[u], bv[u], gb);
            ga = Half<T>::mfma(av[u], av[u], ga);
          }
        }
        const float na = diagonal(ga);
        if (lane < 32) qn[lane] = __builtin_sqrtf(na);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        first = false;
      } else {
        for (int k0 = 0; k0 < H; k0 += 64) {
          x8 av[4], bv[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            av[u] = *reinterpret_cast<const x8*>(ap + k0 + 16 * u);
            bv[u] = *reinterpret_cast<const x8*>(bp + k0 + 16 * u);
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            c = Half<T>::mfma(av[u], bv[u], c);
            gb = Half<T>::mfma(bv[u], bv[u], gb);
          }
        }
      }
      const float nb = diagonal(gb);
      if (pos < S) {
        const float bden = __builtin_sqrtf(nb) + 1e-9f, bmask = dm[pos];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int r = (i >> 2) * 8 + half * 4 + (i & 3);
          if (r < A) sims[r * S + pos] = (1 + r < S) ? c[i] / ((qn[r] + 1e-9f) * bden) * qm[r] * bmask : 0.f;
        }
      }
    }
  }
  __syncthreads();
  cedr_pool_phase(sims, kc, qm0, dm, partial, S, A, K, pg, pk);
}

inline size_t cedr_pool_smem(int S, int A) {
  return (size_t)((S < 4 * kCedrMaxA ? 4 * kCedrMaxA : S) + A * S + 2 * (kCedrMaxK + 1) + 2 * kCedrMaxA + S + 256 * (kCedrMaxK + 1)) * sizeof(float);
}

// LayerNorm of 8 consecutive columns of one row of the fused encoder's activation stream (un-normalised pre-LayerNorm sums P with
// their row statistics): ((P - mu) rstd) gamma + beta in fp32, rounded once to the 16-bit type - the value the encoder's hidden
// state has at these positions (the fused path itself never materialises it, bert.hip "LayerNorm folded into the GEMMs").
template <typename T>
__device__ __forceinline__ typename Half<T>::x8 cedr_ln8(typename Half<T>::x8 v, float2 st, const float* g, const float* b) {
  const float4 g0 = *reinterpret_cast<const float4*>(g), g1 = *reinterpret_cast<const float4*>(g + 4);
  const float4 b0 = *reinterpret_cast<const float4*>(b), b1 = *reinterpret_cast<const float4*>(b + 4);
  const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
  typename Half<T>::x8 o;
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = (T)__builtin_fmaf(((float)v[i] - st.x) * st.y, gg[i], bb[i]);
  return o;
}

// The same tap on the CHUNK-MAJOR activation stream of the fused encoder (cm_offset, bert_gemm.h): a 32-row x 16-column MFMA operand
// fragment is 1 KiB contiguous there, so every operand load of a wave is one fully coalesced instruction (the row-major kernel below
// touches 64 separate 32-byte sectors per load).  x holds either a normalised hidden state (mr == NULL: the embedding output) or the
// pre-LayerNorm sums of a layer with the rows' (mu, rstd) in mr and the layer's gamma / beta: the tap then applies the LayerNorm to
// the operand fragments in registers (cedr_ln8).  The query rows (the A operand of every tile) are normalised once into LDS, in the
// same chunk-major order, and shared by the four waves.  S % 32 == 0, H % 256 == 0.
template <typename T, bool LN>
__global__ __launch_bounds__(256) void cedr_pool_cm_kernel(const T* __restrict__ x, const float2* __restrict__ mr, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const int64_t* __restrict__ mask,
                                                           const int64_t* __restrict__ seg, const float* __restrict__ qmask0, int64_t p0, int S,
                                                           int H, int A, int K, const float* __restrict__ mu, const float* __restrict__ sigma,
                                                           float* __restrict__ pk) {
  typedef typename Half<T>::x8 x8;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* rn = reinterpret_cast<float*>(smem_raw);                 // [4][32] query-row norms, one copy per wave
  float* sims = rn + 4 * kCedrMaxA;                               // [A][S]
  float* kc = sims + A * S;                                       // [12] (mu, coefficient) pairs
  float* qm = kc + 2 * (kCedrMaxK + 1);                           // [A] this passage's query mask
  float* qm0 = qm + kCedrMaxA;                                    // [A] the query mask of the document's first passage
  float* dm = qm0 + kCedrMaxA;                                    // [S] document mask
  float* partial = dm + S;                                        // [256][K + 1]
  float* gam = partial + 256 * (kCedrMaxK + 1);                   // [H]
  float* bet = gam + H;                                           // [H]
  T* As = reinterpret_cast<T*>(bet + H);                          // [H / 8][32][8]: the query rows, normalised, chunk-major
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int p = blockIdx.x, nch = H >> 3;
  const int64_t pg = p0 + p, row0 = (int64_t)p * S;
  const int64_t* mk = mask + (int64_t)p * S;
  const int64_t* sg = seg + (int64_t)p * S;
  constexpr bool ln = LN;

  for (int i = tid; i < S; i += 256) dm[i] = (mk[i] != 0 && sg[i] == 1) ? 1.f : 0.f;
  if (tid < A) {
    qm[tid] = (1 + tid < S && mk[1 + tid] != 0 && sg[1 + tid] == 0) ? 1.f : 0.f;
    qm0[tid] = qmask0[pg * A + tid] != 0.f ? 1.f : 0.f;
  }
  if (tid < kCedrMaxK + 1) {
    const float m = tid < K ? mu[tid] : 0.f, sd = tid < K ? sigma[tid] : 1.f;
    kc[2 * tid] = m;
    kc[2 * tid + 1] = tid < K ? (-0.5f * 1.4426950408889634f) / (sd * sd) : 0.f;
  }
  if (ln)
    for (int i = tid; i < H; i += 256) { gam[i] = gamma[i]; bet[i] = beta[i]; }
  __syncthreads();
  // query rows (sequence positions 1 + m; rows beyond A: a valid row, output never read) -> LDS; four loads in flight per thread
#if defined(CAPAMD_CEDR_ABLATE) && (CAPAMD_CEDR_ABLATE & 4)   // profiling build: no query-row staging
  for (int i0 = tid; i0 < (H < 0 ? 32 * nch : 0); i0 += 4 * 256) {
#else
  for (int i0 = tid; i0 < 32 * nch; i0 += 4 * 256) {
#endif
    x8 v[4];
    float2 stq[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int idx = i0 + j * 256, m = idx & 31, ch = idx >> 5;
      const int64_t row = row0 + ((m < A && 1 + m < S) ? 1 + m : 0);
      if (idx < 32 * nch) {
        v[j] = *reinterpret_cast<const x8*>(x + (((row >> 5) * nch + ch) * 32 + (row & 31)) * 8);
        if (LN) stq[j] = mr[row];
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int idx = i0 + j * 256, m = idx & 31, ch = idx >> 5;
      if (idx < 32 * nch) {
        if (ln) v[j] = cedr_ln8<T>(v[j], stq[j], gam + ch * 8, bet + ch * 8);
        *reinterpret_cast<x8*>(As + (ch * 32 + m) * 8) = v[j];
      }
    }
  }
  __syncthreads();
  {
    const int m = lane & 31, half = lane >> 5;
    const T* ap = As + (half * 32 + m) * 8;
    float* qn = rn + wave * kCedrMaxA;
    const int di = ((m >> 3) << 2) | (m & 3);
    const bool dmine = half == ((m >> 2) & 1);
    auto diagonal = [&](const f32x16& g) {
      float v = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) v = (dmine && i == di) ? g[i] : v;
      return v + __shfl_xor(v, 32, 64);
    };
    // A wave walks its tiles (wave, wave + 4, ...) in trips of 128 columns (8 k-slices of 16: 8 KiB of operand per trip and wave).  One
    // wave per SIMD and nothing else on the CU: the loads of trip i + 1 are issued before trip i is computed (two register buffers),
    // otherwise every trip would wait out a full memory round trip.
    constexpr int U = 8;
    const int tpt = H / (16 * U);                                       // trips per tile (H % 128 == 0)
    const int my_tiles = (S / 32 - wave + 3) / 4;
    const int n = my_tiles * tpt;
    const float* gl = gam + 8 * half;
    const float* bl = bet + 8 * half;
    // trip i of this wave = (tile wave + 4 (i / tpt), columns 128 (i % tpt) ..): the two counters below walk it without divisions
    int lt = wave, lk = 0;                                              // next trip to LOAD
    auto load_b = [&](x8 (&dst)[U], float2& st) {
      // (the trip after a wave's last one is still requested - loads under a condition would make every destination register a merge
      // point the compiler waits at; it re-reads tile 0 and is never computed)
      const int64_t row = row0 + (lt * 32 < S ? lt : 0) * 32 + m;       // (S % 32 == 0: always a row of this passage)
      const T* bp = x + (((row >> 5) * nch + half) * 32 + m) * 8 + lk * 32;   // chunk `half` of the tile's 32-row block; + 512 elements per k-slice
#pragma unroll
      for (int u = 0; u < U; ++u) dst[u] = *reinterpret_cast<const x8*>(bp + u * 512);
      if (LN) st = mr[row];
      lk += 16 * U;
      if (lk == H) { lk = 0; lt += 4; }
    };
    f32x16 c = {0}, gb = {0}, ga = {0};
    bool first = true;
    int pt = wave, pk0 = 0;                                             // next trip to COMPUTE
    auto process = [&](x8 (&bv)[U], const float2 st) {
      const int t = pt, k0 = pk0;
      pk0 += 16 * U;
      if (pk0 == H) { pk0 = 0; pt += 4; }
      if (k0 == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { c[r] = 0.f; gb[r] = 0.f; }
      }
      if (ln) {
#pragma unroll
        for (int u = 0; u < U; ++u) bv[u] = cedr_ln8<T>(bv[u], st, gl + k0 + 16 * u, bl + k0 + 16 * u);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const x8 av = *reinterpret_cast<const x8*>(ap + (k0 + 16 * u) * 32);
        c = Half<T>::mfma(av, bv[u], c);
        gb = Half<T>::mfma(bv[u], bv[u], gb);
        if (first) ga = Half<T>::mfma(av, av, ga);
      }
      if (k0 + 16 * U < H) return;
      // the tile is complete: norms from the diagonals, similarities into LDS
      if (first) {
        const float na = diagonal(ga);
        if (lane < 32) qn[lane] = __builtin_sqrtf(na);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        first = false;
      }
      const int pos = t * 32 + m;
      const float nb = diagonal(gb);
      const float bden = __builtin_sqrtf(nb) + 1e-9f, bmask = dm[pos];
#pragma unroll
      for (int r16 = 0; r16 < 16; ++r16) {
        const int r = (r16 >> 2) * 8 + half * 4 + (r16 & 3);
        if (r < A) sims[r * S + pos] = (1 + r < S) ? c[r16] / ((qn[r] + 1e-9f) * bden) * qm[r] * bmask : 0.f;
      }
    };
    x8 b0[U], b1[U];
    float2 s0 = make_float2(0.f, 1.f), s1 = s0;
#if defined(CAPAMD_CEDR_ABLATE) && (CAPAMD_CEDR_ABLATE & 2)   // profiling build: no tile loop
    if (H < 0)
#endif
    load_b(b0, s0);
#if defined(CAPAMD_CEDR_ABLATE) && (CAPAMD_CEDR_ABLATE & 2)
    for (int i = 0; i < (H < 0 ? n : 0); i += 2) {
#else
    for (int i = 0; i < n; i += 2) {        // (n is even: H % 256 == 0)
#endif
      load_b(b1, s1);
      process(b0, s0);
      load_b(b0, s0);
      process(b1, s1);
    }
  }
  __syncthreads();
#if defined(CAPAMD_CEDR_ABLATE) && (CAPAMD_CEDR_ABLATE & 1)   // profiling build: no pooling
  if (H < 0)
#endif
  cedr_pool_phase(sims, kc, qm0, dm, partial, S, A, K, pg, pk);
}

inline size_t cedr_pool_cm_smem(int S, int H, int A) {
  return (size_t)(4 * kCedrMaxA + A * S + 2 * (kCedrMaxK + 1) + 2 * kCedrMaxA + S + 256 * (kCedrMaxK + 1) + 2 * H) * sizeof(float) + (size_t)H * 32 * 2;
}

template <typename T>
__global__ void cedr_cls_rows_kernel(const T* __restrict__ x, int S, int H, float* __restrict__ cls) {
  const T* row = x + (int64_t)blockIdx.x * S * H;
  for (int j = threadIdx.x; j < H; j += blockDim.x) cls[(int64_t)blockIdx.x * H + j] = (float)row[j];
}
// ... and from the fused encoder's chunk-major stream: LayerNorm applied on the way, rounded to the 16-bit type as the hidden state is
template <typename T>
__global__ void cedr_cls_rows_cm_kernel(const T* __restrict__ x, const float2* __restrict__ mr, const float* __restrict__ gamma,
                                        const float* __restrict__ beta, int S, int H, float* __restrict__ cls) {
  const int64_t row = (int64_t)blockIdx.x * S;
  const float2 st = mr[row];
  for (int j = threadIdx.x; j < H; j += blockDim.x)
    cls[(int64_t)blockIdx.x * H + j] = (float)(T)__builtin_fmaf(((float)x[cm_offset(row, j, H)] - st.x) * st.y, gamma[j], beta[j]);
}

// hidden state `index` (0 = embedding output) of the micro-batch starting at passage p0 is in x: pool it if it is selected.
// mr != NULL: x is the fused encoder's chunk-major stream of pre-LayerNorm sums (gamma / beta of the LayerNorm that makes the hidden
// state); cm with mr == NULL: an already normalised chunk-major tensor (the embedding output).
template <typename T>
void cedr_tap_layer(const CedrTap& tap, int index, const T* x, const int64_t* mask_mb, const int64_t* seg_mb, int64_t p0, int64_t np, int S,
                    int H, hipStream_t s, bool cm = false, const float2* mr = nullptr, const float* gamma = nullptr, const float* beta = nullptr) {
  for (int i = 0; i < tap.n_sel; ++i)
    if (tap.layers[i] == index) {
      float* pk = tap.pk + (int64_t)i * tap.NP * tap.K * tap.A;
      if (cm) {
        auto k = mr ? cedr_pool_cm_kernel<T, true> : cedr_pool_cm_kernel<T, false>;
        const size_t smem = cedr_pool_cm_smem(S, H, tap.A);
        if (smem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(k, dim3((unsigned)np), dim3(256), smem, s, x, mr, gamma, beta, mask_mb, seg_mb, tap.qmask0, p0, S, H, tap.A, tap.K,
                           tap.mu, tap.sigma, pk);
        continue;
      }
      auto k = cedr_pool_kernel<T>;
      const size_t smem = cedr_pool_smem(S, tap.A);
      if (smem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      hipLaunchKernelGGL(k, dim3((unsigned)np), dim3(256), smem, s, x, mask_mb, seg_mb, tap.qmask0, p0, S, H, tap.A, tap.K,
                         tap.mu, tap.sigma, pk);
    }
}

}  // namespace capamd
