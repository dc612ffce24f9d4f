// CEDR-KNRM's document-level head (reference CEDRKNRM_Class.knrm / forward, capreolus/reranker/CEDRKNRM.py:117-185)
// on top of what capamd_cedr_passage_features leaves behind: per-passage kernel sums pk[layer][passage][kernel][query row] and the
// [CLS] rows of the last hidden state.
//   knrm feature (layer, k) = sum_a 0.01 * log(max(sum over the document's passages of pk, 1e-10))     (:120-134)
//   cls feature             = mean or max over the passages of the [CLS] row                             (:160-165)
//   features = [cls | layer 0 kernels | layer 1 kernels | ...]  ->  Linear, or Linear -> Linear           (:171-184, 62-74)
// One workgroup per document.
#include "capreolus_amd.h"
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

constexpr int kMaxIn = 2048, kMaxHidden = 2048;

__global__ __launch_bounds__(256) void cedr_score_kernel(const float* __restrict__ pk, const float* __restrict__ cls, int64_t NP, int P, int A,
                                                         int n_layers, int K, int Hd, int cls_mode, const float* __restrict__ w1,
                                                         const float* __restrict__ b1, int H1, const float* __restrict__ w2,
                                                         const float* __restrict__ b2, float* __restrict__ out, float* __restrict__ feat_out) {
  __shared__ float feat[kMaxIn];
  __shared__ float hid[kMaxHidden];
  __shared__ float red[4];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int64_t b = blockIdx.x;
  const int ncls = cls_mode ? Hd : 0, n_in = ncls + n_layers * K;
  for (int j = tid; j < ncls; j += 256) {
    float v = cls[(b * P) * Hd + j];
    for (int p = 1; p < P; ++p) {
      const float u = cls[(b * P + p) * Hd + j];
      v = cls_mode == 2 ? fmaxf(v, u) : v + u;
    }
    feat[j] = cls_mode == 2 ? v : v / (float)P;
  }
  for (int i = tid; i < n_layers * K; i += 256) {
    const int l = i / K, k = i - l * K;
    float f = 0.f;
    for (int a = 0; a < A; ++a) {
      float s = 0.f;
      for (int p = 0; p < P; ++p) s += pk[(((int64_t)l * NP + b * P + p) * K + k) * A + a];
      f += logf(fmaxf(s, 1e-10f)) * 0.01f;
    }
    feat[ncls + i] = f;
  }
  __syncthreads();
  if (feat_out)
    for (int i = tid; i < n_in; i += 256) feat_out[b * n_in + i] = feat[i];
  if (H1 == 0) {
    float s = 0.f;
    for (int i = tid; i < n_in; i += 256) s = __builtin_fmaf(w1[i], feat[i], s);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (tid == 0) out[b] = ((red[0] + red[1]) + (red[2] + red[3])) + b1[0];
    return;
  }
  for (int h = wave; h < H1; h += 4) {     // one wave per hidden unit: coalesced reads of its weight row
    const float* w = w1 + (int64_t)h * n_in;
    float s = 0.f;
    for (int i = lane; i < n_in; i += 64) s = __builtin_fmaf(w[i], feat[i], s);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) hid[h] = s + b1[h];
  }
  __syncthreads();
  float s = 0.f;
  for (int h = tid; h < H1; h += 256) s = __builtin_fmaf(w2[h], hid[h], s);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  if (tid == 0) out[b] = ((red[0] + red[1]) + (red[2] + red[3])) + b2[0];
}

}  // namespace

extern "C" int capamd_cedr_score(const float* passage_kernel_sums, const float* cls_rows, int B, int P, int maxqlen, int n_layers, int K,
                                 int hidden, int cls_mode, const float* w1, const float* b1, int combine_hidden, const float* w2,
                                 const float* b2, float* out, float* features_out, void* stream) {
  if (B == 0) return CAPAMD_OK;
  if (B < 0 || P < 1 || maxqlen < 1 || n_layers < 0 || K < 1 || hidden < 1 || cls_mode < 0 || cls_mode > 2 || combine_hidden < 0 || !w1 || !b1 || !out)
    return CAPAMD_ERR_ARG;
  if ((n_layers > 0 && !passage_kernel_sums) || (cls_mode && !cls_rows) || (combine_hidden && (!w2 || !b2))) return CAPAMD_ERR_ARG;
  const int n_in = (cls_mode ? hidden : 0) + n_layers * K;
  if (n_in < 1 || n_in > kMaxIn || combine_hidden > kMaxHidden) return CAPAMD_ERR_ARG;
  (void)hipGetLastError();
  hipLaunchKernelGGL(cedr_score_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, passage_kernel_sums, cls_rows, (int64_t)B * P, P, maxqlen + 1,
                     n_layers, K, hidden, cls_mode, w1, b1, combine_hidden, w2, b2, out, features_out);
  return hipGetLastError() == hipSuccess ? CAPAMD_OK : CAPAMD_ERR_LAUNCH;
}
