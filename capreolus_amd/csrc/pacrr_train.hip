// PACRR's trainable front: n-gram Conv2d over the query-document similarity matrix -> ReLU -> max over filters -> k-max over the
// document, forward WITH the winners' coordinates and backward into the convolution weights, for gfx950.
//
// Reference: PACRRConvMax2dModule.forward, capreolus/reranker/PACRR.py:68-78 - ConstantPad2d((0, ng - 1, 0, ng - 1)), Conv2d(1 -> nf,
// ng x ng), ReLU, max over the filter axis, topk(k) over the document axis - under the reference trainer's loss.backward()
// (trainer/pytorch.py:96-107).  The similarity matrix comes from the HIP front end (capamd_similarity_matrix; the embedding table is
// frozen, PACRR.py:26, so no gradient flows into it) and scoring runs the fused kernel of pacrr.hip; this file is the TRAINING step's
// convolution, which needs what scoring does not: which (filter, position) produced each of the k values, so that
//     d bias[f*]          += g                      for every selected value v = conv[f*][q][j*] > 0
//     d weight[f*][dq][dj] += g sim_pad[q + dq][j* + dj]
// Forward: one workgroup per (pair, n-gram size); a thread owns (q, j) positions, walks the filters with the weights in LDS, keeps the
// maximum and its filter; then per query row k rounds of a workgroup-wide arg-max (ties: the smaller position).
// Backward: one workgroup per (n-gram size, filter) sums its selected entries in a FIXED order (pair, q, rank): deterministic, no atomics.
#include "capreolus_amd.h"
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

constexpr int kTrMaxQ = 8, kTrMaxL = 1024, kTrMaxNg = 3, kTrMaxK = 4, kTrMaxF = 256;

struct ConvMaxArgs {
  const float* sim;     // [B, Q, L]
  int B, Q, L, mingram, maxgram, nf, kmax;
  const float* conv_w;  // per n-gram size back to back: [nf][ng][ng]
  const float* conv_b;  // [n_ng][nf]
  float* top;           // [B, Q, n_ng * kmax]   (the reference's cat over the n-gram modules, PACRR.py:45-53)
  int32_t* pos;         // [B, Q, n_ng * kmax]   document position of each value
  int32_t* filt;        // [B, Q, n_ng * kmax]   its filter (-1: the value is a ReLU zero: no gradient)
  const float* gtop;    // [B, Q, n_ng * kmax]
  float* dw;            // like conv_w
  float* db;            // like conv_b
};

__device__ __forceinline__ int w_offset(int mingram, int ng, int nf) {   // floats before n-gram size ng's weights
  int o = 0;
  for (int g = mingram; g < ng; ++g) o += nf * g * g;
  return o;
}

__global__ __launch_bounds__(256) void convmax_forward_kernel(ConvMaxArgs a) {
  extern __shared__ __attribute__((aligned(16))) char tr_lds[];
  const int n_ng = a.maxgram - a.mingram + 1;
  const int b = blockIdx.x / n_ng, gi = blockIdx.x % n_ng, ng = a.mingram + gi;
  const int Lp = a.L + kTrMaxNg;                                  // padded row: zeros right of the document
  float* S = reinterpret_cast<float*>(tr_lds);                    // [Q + kTrMaxNg][Lp] zero-padded similarity matrix
  float* W = S + (kTrMaxQ + kTrMaxNg) * Lp;                       // [nf][ng * ng] + bias [nf]
  float* val = W + kTrMaxF * (kTrMaxNg * kTrMaxNg + 1);           // [Q][L] max over the filters after ReLU
  int* arg = reinterpret_cast<int*>(val + kTrMaxQ * a.L);         // [Q][L] its filter
  float* rv = reinterpret_cast<float*>(arg + kTrMaxQ * a.L);      // [256] reduction scratch
  int* ri = reinterpret_cast<int*>(rv + 256);                     // [256]
  const int tid = threadIdx.x;
  for (int i = tid; i < (a.Q + kTrMaxNg) * Lp; i += 256) {
    const int q = i / Lp, j = i - q * Lp;
    S[i] = (q < a.Q && j < a.L) ? a.sim[((int64_t)b * a.Q + q) * a.L + j] : 0.f;
  }
  const float* wsrc = a.conv_w + w_offset(a.mingram, ng, a.nf);
  for (int i = tid; i < a.nf * ng * ng; i += 256) W[i] = wsrc[i];
  for (int i = tid; i < a.nf; i += 256) W[a.nf * ng * ng + i] = a.conv_b[gi * a.nf + i];
  __syncthreads();
  for (int i = tid; i < a.Q * a.L; i += 256) {
    const int q = i / a.L, j = i - q * a.L;
    float win[kTrMaxNg * kTrMaxNg];
    for (int dq = 0; dq < ng; ++dq)
      for (int dj = 0; dj < ng; ++dj) win[dq * ng + dj] = S[(q + dq) * Lp + j + dj];
    float best = 0.f;      // ReLU: nothing below zero survives
    int bf = -1;
    for (int f = 0; f < a.nf; ++f) {
      float v = W[a.nf * ng * ng + f];
      for (int t = 0; t < ng * ng; ++t) v = __builtin_fmaf(W[f * ng * ng + t], win[t], v);
      if (v > best) { best = v; bf = f; }
    }
    val[i] = best;
    arg[i] = bf;
  }
  __syncthreads();
  // k-max per query row: k rounds of a workgroup arg-max (value descending, then position ascending)
  for (int q = 0; q < a.Q; ++q)
    for (int r = 0; r < a.kmax; ++r) {
      float bv = -1.f;
      int bj = 0x7fffffff;
      for (int j = tid; j < a.L; j += 256) {
        const float v = val[q * a.L + j];
        if (v > bv) { bv = v; bj = j; }      // (j ascending per thread: the first maximum)
      }
      rv[tid] = bv;
      ri[tid] = bj;
      __syncthreads();
      for (int sft = 128; sft > 0; sft >>= 1) {
        if (tid < sft) {
          const float ov = rv[tid + sft];
          const int oj = ri[tid + sft];
          if (ov > rv[tid] || (ov == rv[tid] && oj < ri[tid])) { rv[tid] = ov; ri[tid] = oj; }
        }
        __syncthreads();
      }
      if (tid == 0) {
        const int j = ri[0];
        const int64_t o = ((int64_t)b * a.Q + q) * (n_ng * a.kmax) + gi * a.kmax + r;
        const bool any = j < a.L;           // (L < kmax: fewer positions than ranks - the reference's topk would raise; zeros here)
        a.top[o] = any ? rv[0] : 0.f;
        a.pos[o] = any ? j : 0;
        a.filt[o] = any ? arg[q * a.L + j] : -1;
        if (any) val[q * a.L + j] = -2.f;   // taken
      }
      __syncthreads();
    }
}

__global__ __launch_bounds__(64) void convmax_backward_kernel(ConvMaxArgs a) {
  const int n_ng = a.maxgram - a.mingram + 1;
  const int gi = blockIdx.x / a.nf, f = blockIdx.x % a.nf, ng = a.mingram + gi;
  const int tap = threadIdx.x;                    // 0 .. ng*ng-1: a weight, ng*ng: the bias
  if (tap > ng * ng) return;
  const int dq = tap / ng, dj = tap - dq * ng;
  float acc = 0.f;
  for (int b = 0; b < a.B; ++b)
    for (int q = 0; q < a.Q; ++q)
      for (int r = 0; r < a.kmax; ++r) {
        const int64_t o = ((int64_t)b * a.Q + q) * (n_ng * a.kmax) + gi * a.kmax + r;
        if (a.filt[o] != f) continue;
        const float g = a.gtop[o];
        if (tap == ng * ng) { acc += g; continue; }
        const int qq = q + dq, jj = a.pos[o] + dj;
        const float s = (qq < a.Q && jj < a.L) ? a.sim[((int64_t)b * a.Q + qq) * a.L + jj] : 0.f;
        acc = __builtin_fmaf(g, s, acc);
      }
  if (tap == ng * ng) a.db[gi * a.nf + f] = acc;
  else a.dw[w_offset(a.mingram, ng, a.nf) + f * ng * ng + tap] = acc;
}

int convmax_check(const ConvMaxArgs& a) {
  if (!a.sim || !a.conv_w || !a.conv_b || !a.pos || !a.filt) return CAPAMD_ERR_ARG;
  if (a.B < 0 || a.Q < 1 || a.Q > kTrMaxQ || a.L < 1 || a.L > kTrMaxL || a.mingram < 1 || a.maxgram < a.mingram || a.maxgram > kTrMaxNg) return CAPAMD_ERR_ARG;
  if (a.nf < 1 || a.nf > kTrMaxF || a.kmax < 1 || a.kmax > kTrMaxK) return CAPAMD_ERR_ARG;
  return CAPAMD_OK;
}

}  // namespace

extern "C" int capamd_pacrr_convmax_forward(const float* sim, int B, int Q, int L, int mingram, int maxgram, int nfilters, int kmax,
                                            const float* conv_w, const float* conv_b, float* top, int32_t* pos, int32_t* filt, void* stream) {
  ConvMaxArgs a{sim, B, Q, L, mingram, maxgram, nfilters, kmax, conv_w, conv_b, top, pos, filt, nullptr, nullptr, nullptr};
  if (!top) return CAPAMD_ERR_ARG;
  const int rc = convmax_check(a);
  if (rc != CAPAMD_OK || B == 0) return rc;
  const size_t lds = ((size_t)(kTrMaxQ + kTrMaxNg) * (L + kTrMaxNg) + kTrMaxF * (kTrMaxNg * kTrMaxNg + 1) + 2 * (size_t)kTrMaxQ * L + 512) * 4;
  (void)hipGetLastError();
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(convmax_forward_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return CAPAMD_ERR_LAUNCH;
  hipLaunchKernelGGL(convmax_forward_kernel, dim3(B * (maxgram - mingram + 1)), dim3(256), lds, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? CAPAMD_OK : CAPAMD_ERR_LAUNCH;
}

extern "C" int capamd_pacrr_convmax_backward(const float* sim, int B, int Q, int L, int mingram, int maxgram, int nfilters, int kmax,
                                             const float* gtop, const int32_t* pos, const int32_t* filt, float* dconv_w, float* dconv_b, void* stream) {
  const float dummy = 0.f;
  ConvMaxArgs a{sim, B, Q, L, mingram, maxgram, nfilters, kmax, &dummy, &dummy, nullptr, const_cast<int32_t*>(pos), const_cast<int32_t*>(filt), gtop, dconv_w, dconv_b};
  if (!gtop || !dconv_w || !dconv_b) return CAPAMD_ERR_ARG;
  const int rc = convmax_check(a);
  if (rc != CAPAMD_OK) return rc;
  (void)hipGetLastError();
  hipLaunchKernelGGL(convmax_backward_kernel, dim3((maxgram - mingram + 1) * nfilters), dim3(64), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? CAPAMD_OK : CAPAMD_ERR_LAUNCH;
}
