// On-device ranking of scored candidate lists for gfx950: the fp16 rounding, the per-query
// stable sort and the nDCG@k that follow the scoring path in the reference, without bringing the scores to the host.
//
// Reference semantics restated:
//  * PytorchTrainer.predict stores score.astype(np.float16) (capreolus/trainer/pytorch.py:346-348): round-to-nearest-even
//    fp32 -> fp16, overflow to inf;
//  * Searcher.write_trec_run ranks a query's documents with sorted(items, key=score, reverse=True)
//    (capreolus/searcher/__init__.py:48-58): descending rounded score, ties in insertion order (Python's sort is stable,
//    also with reverse=True); -0.0 and +0.0 compare equal;
//  * evaluator.py:55-85 hands the run to pytrec_eval (trec_eval): ranked by score descending with ties broken by docid
//    descending (the run's rank column is ignored), ndcg_cut_k = DCG_k / IDCG_k with gain = relevance level (negative
//    levels count 0), discount log2(rank + 1), the ideal ranking taken from ALL judged documents of the query.
//
// One workgroup per query.  Every candidate becomes one 32-bit key = (descending-sortable fp16 score << 16) | tie-break,
// tie-break = the candidate's position (run writer) or its docid-descending rank (nDCG), both < 65536 and unique within
// the query, so the keys are unique and a bitonic sort in LDS reproduces the stable order exactly.  Integer work, LDS-
// resident; a list of 1000 candidates is 4 KB of keys.
#include "capreolus_amd.h"
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

constexpr int kRankThreads = 256;
constexpr int kMaxCandidates = 16384;

// fp32 -> rounded fp16 bits, and the ascending-sortable key of "larger score first"
__device__ __forceinline__ uint16_t f16_bits(float x) {
  const _Float16 h = (_Float16)x;  // v_cvt_f16_f32: round to nearest even, overflow -> inf (== numpy astype(float16))
  return __builtin_bit_cast(uint16_t, h);
}
__device__ __forceinline__ uint32_t desc_key(uint16_t bits, bool& is_nan) {
  is_nan = (bits & 0x7fffu) > 0x7c00u;
  if ((bits & 0x7fffu) == 0) bits = 0;                        // -0.0 == +0.0 for the comparison
  const uint16_t asc = (bits & 0x8000u) ? (uint16_t)~bits : (uint16_t)(bits | 0x8000u);  // larger float -> larger integer
  return is_nan ? 0xffffu : (uint32_t)(0xffffu - asc);        // larger float -> smaller key; NaN last
}

// bitonic sort of keys[0 .. n2) (n2 a power of two) ascending, by the whole workgroup
__device__ void bitonic_sort(uint32_t* keys, int n2) {
  for (int size = 2; size <= n2; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();
      for (int t = threadIdx.x; t < (n2 >> 1); t += blockDim.x) {
        const int lo = 2 * t - (t & (stride - 1));  // index with bit `stride` clear
        const int hi = lo + stride;
        const bool up = (lo & size) == 0;
        const uint32_t a = keys[lo], b = keys[hi];
        if ((a > b) == up) { keys[lo] = b; keys[hi] = a; }
      }
    }
  }
  __syncthreads();
}

__device__ __forceinline__ int pow2_at_least(int n) {
  int p = 1;
  while (p < n) p <<= 1;
  return p;
}

__global__ __launch_bounds__(kRankThreads) void rank_kernel(const float* __restrict__ scores, const int64_t* __restrict__ offsets, int k,
                                                            int32_t* __restrict__ out_idx, uint16_t* __restrict__ out_f16, int* status) {
  extern __shared__ uint32_t keys[];
  const int q = blockIdx.x;
  const int64_t base = offsets[q];
  const int n = (int)(offsets[q + 1] - base), n2 = pow2_at_least(n < 1 ? 1 : n);
  for (int i = threadIdx.x; i < n2; i += blockDim.x) {
    uint32_t key = 0xffffffffu;
    if (i < n) {
      bool nan;
      key = (desc_key(f16_bits(scores[base + i]), nan) << 16) | (uint32_t)i;
      if (nan) atomicOr(status, CAPAMD_STATUS_SCORE_NAN);
    }
    keys[i] = key;
  }
  bitonic_sort(keys, n2);
  for (int r = threadIdx.x; r < k; r += blockDim.x) {
    int idx = -1;
    uint16_t bits = 0;
    if (r < n) {
      idx = (int)(keys[r] & 0xffffu);
      bits = f16_bits(scores[base + idx]);
    }
    out_idx[(int64_t)q * k + r] = idx;
    out_f16[(int64_t)q * k + r] = bits;
  }
}

__global__ __launch_bounds__(kRankThreads) void ndcg_kernel(const float* __restrict__ scores, const int64_t* __restrict__ offsets,
                                                            const int32_t* __restrict__ rel, const int32_t* __restrict__ tie,
                                                            const double* __restrict__ idcg, int k, double* __restrict__ out, int* status) {
  extern __shared__ uint32_t keys[];
  __shared__ double part[kRankThreads];
  const int q = blockIdx.x;
  const int64_t base = offsets[q];
  const int n = (int)(offsets[q + 1] - base), n2 = pow2_at_least(n < 1 ? 1 : n);
  // the tie-break must order the candidates of the query by itself: carry the position in a second array
  uint32_t* pos = keys + n2;
  for (int i = threadIdx.x; i < n2; i += blockDim.x) pos[i] = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < n2; i += blockDim.x) {
    uint32_t key = 0xffffffffu;
    if (i < n) {
      bool nan;
      int t = tie[base + i];
      if (t < 0 || t >= n) {
        atomicOr(status, CAPAMD_STATUS_TIE_RANGE);  // flagged; the result of this query is meaningless but stays in bounds
        t &= n2 - 1;
      }
      key = (desc_key(f16_bits(scores[base + i]), nan) << 16) | (uint32_t)t;
      if (nan) atomicOr(status, CAPAMD_STATUS_SCORE_NAN);
      pos[t] = (uint32_t)i;   // tie ranks are a permutation of 0..n-1: rank -> position
    }
    keys[i] = key;
  }
  bitonic_sort(keys, n2);
  double g = 0.0;
  for (int r = threadIdx.x; r < k && r < n; r += blockDim.x) {
    int i = (int)pos[keys[r] & 0xffffu];
    if (i >= n) i = 0;  // only reachable with a flagged (non-permutation) tie array
    const int lvl = rel[base + i];
    if (lvl > 0) g += (double)lvl / log2((double)(r + 2));
  }
  part[threadIdx.x] = g;
  __syncthreads();
  if (threadIdx.x == 0) {
    // fixed-order sum: rank 0 first (the order a sequential evaluator adds the gains in)
    double dcg = 0.0;
    const int top = k < n ? k : n;
    for (int r = 0; r < top && r < kRankThreads; ++r) dcg += part[r];
    const double ideal = idcg[q];
    out[q] = ideal > 0.0 ? dcg / ideal : 0.0;
  }
}

}  // namespace

extern "C" {

int capamd_rank_candidates(const float* scores, const int64_t* offsets, int n_queries, int max_candidates, int k, int32_t* out_idx,
                           uint16_t* out_f16, int* status, void* stream) {
  if (n_queries == 0) return CAPAMD_OK;
  if (!scores || !offsets || !out_idx || !out_f16 || !status || n_queries < 0 || k < 1 || max_candidates < 0) return CAPAMD_ERR_ARG;
  if (max_candidates > kMaxCandidates) return CAPAMD_ERR_ARG;
  int n2 = 1;
  while (n2 < max_candidates) n2 <<= 1;
  (void)hipGetLastError();
  hipLaunchKernelGGL(rank_kernel, dim3((unsigned)n_queries), dim3(kRankThreads), (size_t)n2 * 4, (hipStream_t)stream, scores, offsets, k, out_idx,
                     out_f16, status);
  return hipGetLastError() == hipSuccess ? CAPAMD_OK : CAPAMD_ERR_LAUNCH;
}

int capamd_ndcg_cut(const float* scores, const int64_t* offsets, const int32_t* rel, const int32_t* tie, const double* idcg, int n_queries,
                    int max_candidates, int k, double* out, int* status, void* stream) {
  if (n_queries == 0) return CAPAMD_OK;
  if (!scores || !offsets || !rel || !tie || !idcg || !out || !status || n_queries < 0 || k < 1 || k > kRankThreads || max_candidates < 0)
    return CAPAMD_ERR_ARG;
  if (max_candidates > kMaxCandidates) return CAPAMD_ERR_ARG;
  int n2 = 1;
  while (n2 < max_candidates) n2 <<= 1;
  (void)hipGetLastError();
  hipLaunchKernelGGL(ndcg_kernel, dim3((unsigned)n_queries), dim3(kRankThreads), (size_t)n2 * 8, (hipStream_t)stream, scores, offsets, rel, tie,
                     idcg, k, out, status);
  return hipGetLastError() == hipSuccess ? CAPAMD_OK : CAPAMD_ERR_LAUNCH;
}

}  // extern "C"
