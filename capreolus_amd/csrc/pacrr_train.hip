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
// maximum and its filter (the window in registers: the n-gram size is a template parameter - indexed by a run-time size it lived in
// scratch memory and the kernel took 187 us per batch of 64 documents instead of 20); then a WAVE per query row: k rounds of a wave-wide
// arg-max over the row (ties: the smaller position), no barriers.
// Backward: one workgroup per (n-gram size, filter): its threads share the (pair, q, rank) entries, every thread sums its entries in
// order, the threads' partial sums are added in a fixed tree: deterministic, no atomics.
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

template <int NG>
__device__ __forceinline__ void convmax_forward_body(const ConvMaxArgs& a, char* tr_lds) {
  const int n_ng = a.maxgram - a.mingram + 1;
  const int b = blockIdx.x, gi = NG - a.mingram;
  const int Lp = a.L + kTrMaxNg;                                  // padded row: zeros right of the document
  float* S = reinterpret_cast<float*>(tr_lds);                    // [Q + kTrMaxNg][Lp] zero-padded similarity matrix
  float* W = S + (kTrMaxQ + kTrMaxNg) * Lp;                       // [nf][NG * NG] + bias [nf]
  float* val = W + kTrMaxF * (kTrMaxNg * kTrMaxNg + 1);           // [Q][L] max over the filters after ReLU
  int* arg = reinterpret_cast<int*>(val + kTrMaxQ * a.L);         // [Q][L] its filter
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < (a.Q + kTrMaxNg) * Lp; i += 256) {
    const int q = i / Lp, j = i - q * Lp;
    S[i] = (q < a.Q && j < a.L) ? a.sim[((int64_t)b * a.Q + q) * a.L + j] : 0.f;
  }
  const float* wsrc = a.conv_w + w_offset(a.mingram, NG, a.nf);
  for (int i = tid; i < a.nf * NG * NG; i += 256) W[i] = wsrc[i];
  for (int i = tid; i < a.nf; i += 256) W[a.nf * NG * NG + i] = a.conv_b[gi * a.nf + i];
  __syncthreads();
  for (int i = tid; i < a.Q * a.L; i += 256) {
    const int q = i / a.L, j = i - q * a.L;
    float win[NG * NG];
#pragma unroll
    for (int dq = 0; dq < NG; ++dq)
#pragma unroll
      for (int dj = 0; dj < NG; ++dj) win[dq * NG + dj] = S[(q + dq) * Lp + j + dj];
    float best = 0.f;      // ReLU: nothing below zero survives
    int bf = -1;
    for (int f = 0; f < a.nf; ++f) {
      float v = W[a.nf * NG * NG + f];
#pragma unroll
      for (int t = 0; t < NG * NG; ++t) v = __builtin_fmaf(W[f * NG * NG + t], win[t], v);
      if (v > best) { best = v; bf = f; }
    }
    val[i] = best;
    arg[i] = bf;
  }
  __syncthreads();
  // k-max per query row, a wave per row: k rounds of a wave arg-max (value descending, then position ascending)
  for (int q = wave; q < a.Q; q += 4)
    for (int r = 0; r < a.kmax; ++r) {
      float bv = -1.f;
      int bj = 0x7fffffff;
      for (int j = lane; j < a.L; j += 64) {
        const float v = val[q * a.L + j];
        if (v > bv) { bv = v; bj = j; }      // (j ascending per lane: the first maximum)
      }
#pragma unroll
      for (int sft = 32; sft > 0; sft >>= 1) {
        const float ov = __shfl_xor(bv, sft, 64);
        const int oj = __shfl_xor(bj, sft, 64);
        if (ov > bv || (ov == bv && oj < bj)) { bv = ov; bj = oj; }
      }
      const bool any = bj < a.L;             // (L < kmax: fewer positions than ranks - the reference's topk would raise; zeros here)
      if (lane == 0) {
        const int64_t o = ((int64_t)b * a.Q + q) * (n_ng * a.kmax) + gi * a.kmax + r;
        a.top[o] = any ? bv : 0.f;
        a.pos[o] = any ? bj : 0;
        a.filt[o] = any ? arg[q * a.L + bj] : -1;
        if (any) val[q * a.L + bj] = -2.f;   // taken (the wave's next round reads it: same wave, program order)
      }
    }
}

__global__ __launch_bounds__(256) void convmax_forward_kernel(ConvMaxArgs a) {
  extern __shared__ __attribute__((aligned(16))) char tr_lds[];
  switch (a.maxgram - (int)blockIdx.y) {          // large n-gram sizes first
    case 1: convmax_forward_body<1>(a, tr_lds); break;
    case 2: convmax_forward_body<2>(a, tr_lds); break;
    default: convmax_forward_body<3>(a, tr_lds); break;
  }
}

constexpr int kBwdThreads = 64;

__global__ __launch_bounds__(kBwdThreads) void convmax_backward_kernel(ConvMaxArgs a) {
  __shared__ float part[kBwdThreads][kTrMaxNg * kTrMaxNg + 1];
  const int n_ng = a.maxgram - a.mingram + 1;
  const int gi = blockIdx.x / a.nf, f = blockIdx.x % a.nf, ng = a.mingram + gi;
  const int tid = threadIdx.x, entries = a.B * a.Q * a.kmax;
  float acc[kTrMaxNg * kTrMaxNg + 1];          // the ng x ng weights' sums (row-major in a kTrMaxNg-wide window), then the bias's
#pragma unroll
  for (int t = 0; t <= kTrMaxNg * kTrMaxNg; ++t) acc[t] = 0.f;
  for (int e = tid; e < entries; e += kBwdThreads) {
    const int r = e % a.kmax, bq = e / a.kmax, q = bq % a.Q, b = bq / a.Q;
    const int64_t o = (int64_t)bq * (n_ng * a.kmax) + gi * a.kmax + r;
    if (a.filt[o] != f) continue;
    const float g = a.gtop[o];
    const int p = a.pos[o];
    acc[kTrMaxNg * kTrMaxNg] += g;
#pragma unroll
    for (int dq = 0; dq < kTrMaxNg; ++dq)
#pragma unroll
      for (int dj = 0; dj < kTrMaxNg; ++dj) {
        const int qq = q + dq, jj = p + dj;
        const float s = (dq < ng && dj < ng && qq < a.Q && jj < a.L) ? a.sim[((int64_t)b * a.Q + qq) * a.L + jj] : 0.f;
        acc[dq * kTrMaxNg + dj] = __builtin_fmaf(g, s, acc[dq * kTrMaxNg + dj]);
      }
  }
#pragma unroll
  for (int t = 0; t <= kTrMaxNg * kTrMaxNg; ++t) part[tid][t] = acc[t];
  __syncthreads();
  for (int sft = kBwdThreads / 2; sft > 0; sft >>= 1) {          // a fixed tree
    if (tid < sft)
      for (int t = 0; t <= kTrMaxNg * kTrMaxNg; ++t) part[tid][t] += part[tid + sft][t];
    __syncthreads();
  }
  if (tid < ng * ng) a.dw[w_offset(a.mingram, ng, a.nf) + f * ng * ng + tid] = part[0][(tid / ng) * kTrMaxNg + tid % ng];
  if (tid == ng * ng) a.db[gi * a.nf + f] = part[0][kTrMaxNg * kTrMaxNg];
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
  const size_t lds = ((size_t)(kTrMaxQ + kTrMaxNg) * (L + kTrMaxNg) + kTrMaxF * (kTrMaxNg * kTrMaxNg + 1) + 2 * (size_t)kTrMaxQ * L) * 4;
  (void)hipGetLastError();
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(convmax_forward_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return CAPAMD_ERR_LAUNCH;
  hipLaunchKernelGGL(convmax_forward_kernel, dim3(B, maxgram - mingram + 1), dim3(256), lds, (hipStream_t)stream, a);
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
  hipLaunchKernelGGL(convmax_backward_kernel, dim3((maxgram - mingram + 1) * nfilters), dim3(kBwdThreads), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? CAPAMD_OK : CAPAMD_ERR_LAUNCH;
}
