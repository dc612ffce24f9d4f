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
#include "bert_gemm.cuh"

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
  // kernel pooling: thread -> (query row, slice of the document columns)
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

inline size_t cedr_pool_smem(int S, int A) {
  return (size_t)((S < 4 * kCedrMaxA ? 4 * kCedrMaxA : S) + A * S + 2 * (kCedrMaxK + 1) + 2 * kCedrMaxA + S + 256 * (kCedrMaxK + 1)) * sizeof(float);
}

template <typename T>
__global__ void cedr_cls_rows_kernel(const T* __restrict__ x, int S, int H, float* __restrict__ cls) {
  const T* row = x + (int64_t)blockIdx.x * S * H;
  for (int j = threadIdx.x; j < H; j += blockDim.x) cls[(int64_t)blockIdx.x * H + j] = (float)row[j];
}

// hidden state `index` (0 = embedding output) of the micro-batch starting at passage p0 is in x: pool it if it is selected
template <typename T>
void cedr_tap_layer(const CedrTap& tap, int index, const T* x, const int64_t* mask_mb, const int64_t* seg_mb, int64_t p0, int64_t np, int S,
                    int H, hipStream_t s) {
  for (int i = 0; i < tap.n_sel; ++i)
    if (tap.layers[i] == index) {
      auto k = cedr_pool_kernel<T>;
      const size_t smem = cedr_pool_smem(S, tap.A);
      if (smem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      hipLaunchKernelGGL(k, dim3((unsigned)np), dim3(256), smem, s, x, mask_mb, seg_mb, tap.qmask0, p0, S, H, tap.A, tap.K,
                         tap.mu, tap.sigma, tap.pk + (int64_t)i * tap.NP * tap.K * tap.A);
    }
}

}  // namespace capamd
