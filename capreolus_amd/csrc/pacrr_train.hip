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

// ---- one PACRR training step on the device ---------------------------------------------------------------------------------------------
// Reference: the loop body of PytorchTrainer.single_train_iteration, capreolus/trainer/pytorch.py:93-108 - PACRR.score (PACRR.py:42-55 on
// the positive and the negative documents), pair_hinge_loss / pair_softmax_loss (reranker/common.py:96-103), loss.backward(),
// torch.optim.Adam.step().  Six launches: the convolution parameters gathered from their nn.Conv2d tensors, the similarity matrices
// (capamd_similarity_matrix), convmax forward, ONE workgroup for everything between the k-max values and the loss and back
// (features + idf softmax, the three Linear layers with their nonlinearity, the loss, the backward through them, their gradients summed
// in document order, Adam on them), convmax backward, Adam on the convolutions.
namespace {

constexpr int kStepMaxNg = kTrMaxNg, kStepMaxH = 128, kStepMaxX = kTrMaxQ * (kTrMaxNg * kTrMaxK + 1), kStepLdsFloats = 36 * 1024;

struct StepAdam {
  float step_size, one_minus_beta1, beta2, eps, bc2_sqrt;
};

__device__ __forceinline__ void step_adam(float* p, float* m, float* v, float g, const StepAdam& s) {
  float mm = *m, vv = *v;      // torch.optim.Adam, single-tensor path: lerp_, mul_ / addcmul_, sqrt / bias_correction2_sqrt + eps, addcdiv_
  mm = mm + (g - mm) * s.one_minus_beta1;
  vv = vv * s.beta2 + (1.f - s.beta2) * (g * g);
  *m = mm;
  *v = vv;
  *p = *p - s.step_size * (mm / (sqrtf(vv) / s.bc2_sqrt + s.eps));
}

struct ConvParams {
  float *w[kStepMaxNg], *wm[kStepMaxNg], *wv[kStepMaxNg];      // ngrams.{i}.conv.weight [nf, 1, ng, ng] and its moments
  float *b[kStepMaxNg], *bm[kStepMaxNg], *bv[kStepMaxNg];      // ngrams.{i}.conv.bias [nf]
  int mingram, n_ng, nf;
};

// mode 0: parameters -> conv_w / conv_b (the layout of the convmax kernels); mode 1: Adam from dw / db in that layout
__global__ __launch_bounds__(256) void pacrr_conv_params_kernel(ConvParams c, float* conv_w, float* conv_b, const float* dw, const float* db, int mode,
                                                                StepAdam adam) {
  const int gi = blockIdx.y, ng = c.mingram + gi, nw = c.nf * ng * ng;
  const int off = w_offset(c.mingram, ng, c.nf);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < nw + c.nf; i += gridDim.x * 256) {
    const bool isw = i < nw;
    const int e = isw ? i : i - nw;
    if (mode == 0) {
      if (isw) conv_w[off + e] = c.w[gi][e];
      else conv_b[gi * c.nf + e] = c.b[gi][e];
    } else if (isw) {
      if (c.wm[gi]) step_adam(c.w[gi] + e, c.wm[gi] + e, c.wv[gi] + e, dw[off + e], adam);
    } else if (c.bm[gi]) {
      step_adam(c.b[gi] + e, c.bm[gi] + e, c.bv[gi] + e, db[gi * c.nf + e], adam);
    }
  }
}

struct MlpArgs {
  const float* top;     // [N, Q, nt] k-max values (nt = n_ng * kmax), documents 0..B-1 positive, B..2B-1 negative
  const float* idf;     // [N, Q] raw query idf (use_idf) or null
  int B, Q, nt, use_idf, H, act, loss_type;
  float *w1, *w1m, *w1v, *b1, *b1m, *b1v;      // linear1 [H][X], X = Q (nt + use_idf)
  float *w2, *w2m, *w2v, *b2, *b2m, *b2v;      // linear2 [H][H]
  float *w3, *w3m, *w3v, *b3, *b3m, *b3v;      // linear3 [1][H]
  StepAdam adam;
  float* gtop;          // [N, Q, nt]
  float* loss_out;
};

__device__ __forceinline__ float act_fwd(int act, float z) { return act == 1 ? fmaxf(z, 0.f) : act == 2 ? tanhf(z) : z; }
__device__ __forceinline__ float act_bwd(int act, float h) { return act == 1 ? (h > 0.f ? 1.f : 0.f) : act == 2 ? 1.f - h * h : 1.f; }   // from the OUTPUT h

__global__ __launch_bounds__(256) void pacrr_mlp_step_kernel(MlpArgs a) {
  extern __shared__ __attribute__((aligned(16))) float ms[];
  const int tid = threadIdx.x, N = 2 * a.B, H = a.H, qt = a.nt + a.use_idf, X = a.Q * qt;
  float* W1 = ms;                 // [H][X]
  float* W2 = W1 + H * X;         // [H][H]
  float* W3 = W2 + H * H;         // [H]
  float* Xs = W3 + H;             // [N][X]
  float* H1 = Xs + N * X;         // [N][H]   (outputs of the first nonlinearity)
  float* H2 = H1 + N * H;         // [N][H]
  float* D1 = H2 + N * H;         // [N][H]   d loss / d (pre-activation 1)
  float* D2 = D1 + N * H;         // [N][H]
  float* gs = D2 + N * H;         // [N] d loss / d score
  float* ls = gs + N;             // [B] the pairs' losses
  float* sc = ls + a.B;           // [N] scores
  for (int i = tid; i < H * X; i += 256) W1[i] = a.w1[i];
  for (int i = tid; i < H * H; i += 256) W2[i] = a.w2[i];
  for (int i = tid; i < H; i += 256) W3[i] = a.w3[i];
  // features: per query term its k-max values, then (PACRR.py:46-50) the softmax of the query's raw idf values
  for (int i = tid; i < N * a.Q; i += 256) {
    const int n = i / a.Q, q = i - n * a.Q;
    for (int c = 0; c < a.nt; ++c) Xs[n * X + q * qt + c] = a.top[(int64_t)i * a.nt + c];
    if (a.use_idf) {
      const float* v = a.idf + (int64_t)n * a.Q;
      float mx = v[0];
      for (int t = 1; t < a.Q; ++t) mx = fmaxf(mx, v[t]);
      float den = 0.f;
      for (int t = 0; t < a.Q; ++t) den += expf(v[t] - mx);
      Xs[n * X + q * qt + a.nt] = expf(v[q] - mx) / den;
    }
  }
  __syncthreads();
  for (int i = tid; i < N * H; i += 256) {
    const int n = i / H, h = i - n * H;
    float z = a.b1[h];
    for (int x = 0; x < X; ++x) z = __builtin_fmaf(W1[h * X + x], Xs[n * X + x], z);
    H1[i] = act_fwd(a.act, z);
  }
  __syncthreads();
  for (int i = tid; i < N * H; i += 256) {
    const int n = i / H, h = i - n * H;
    float z = a.b2[h];
    for (int j = 0; j < H; ++j) z = __builtin_fmaf(W2[h * H + j], H1[n * H + j], z);
    H2[i] = act_fwd(a.act, z);
  }
  __syncthreads();
  for (int n = tid; n < N; n += 256) {
    float z = a.b3[0];
    for (int j = 0; j < H; ++j) z = __builtin_fmaf(W3[j], H2[n * H + j], z);
    sc[n] = z;
  }
  __syncthreads();
  const float inv_b = 1.f / (float)a.B;
  for (int i = tid; i < a.B; i += 256) {
    const float sp = sc[i], sn = sc[a.B + i];
    float li, gp, gn;
    if (a.loss_type == 0) {
      const float mrg = 1.f - (sp - sn);
      li = fmaxf(mrg, 0.f);
      const float on = mrg >= 0.f ? inv_b : 0.f;      // (torch.clamp's backward passes the gradient at the boundary)
      gp = -on;
      gn = on;
    } else {
      const float mx = fmaxf(sp, sn), e0 = expf(sp - mx), e1 = expf(sn - mx), p0 = e0 / (e0 + e1), p1 = e1 / (e0 + e1);
      li = 1.f - p0;
      gp = -p0 * p1 * inv_b;
      gn = p0 * p1 * inv_b;
    }
    gs[i] = gp;
    gs[a.B + i] = gn;
    ls[i] = li;
  }
  __syncthreads();
  // backward through the layers, with the weights as they were before this step's update (the LDS copies)
  for (int i = tid; i < N * H; i += 256) {
    const int n = i / H, h = i - n * H;
    D2[i] = gs[n] * W3[h] * act_bwd(a.act, H2[i]);
  }
  __syncthreads();
  for (int i = tid; i < N * H; i += 256) {
    const int n = i / H, h = i - n * H;
    float g = 0.f;
    for (int j = 0; j < H; ++j) g = __builtin_fmaf(D2[n * H + j], W2[j * H + h], g);
    D1[i] = g * act_bwd(a.act, H1[i]);
  }
  __syncthreads();
  for (int i = tid; i < N * a.Q * a.nt; i += 256) {
    const int c = i % a.nt, nq = i / a.nt, q = nq % a.Q, n = nq / a.Q, x = q * qt + c;
    float g = 0.f;
    for (int h = 0; h < H; ++h) g = __builtin_fmaf(D1[n * H + h], W1[h * X + x], g);
    a.gtop[i] = g;
  }
  // one thread per parameter element (and one for the loss): its gradient summed over the documents in order, then Adam
  const int n1 = H * X, n2 = H * H, total = n1 + H + n2 + H + H + 1;
  for (int j = tid; j <= total; j += 256) {
    if (j == total) {
      float l = 0.f;
      for (int i = 0; i < a.B; ++i) l += ls[i];
      a.loss_out[0] = l * inv_b;
      continue;
    }
    float g = 0.f;
    float *p, *m, *v;
    int e = j;
    if (e < n1) {
      const int h = e / X, x = e - h * X;
      for (int n = 0; n < N; ++n) g = __builtin_fmaf(D1[n * H + h], Xs[n * X + x], g);
      p = a.w1; m = a.w1m; v = a.w1v;
    } else if ((e -= n1) < H) {
      for (int n = 0; n < N; ++n) g += D1[n * H + e];
      p = a.b1; m = a.b1m; v = a.b1v;
    } else if ((e -= H) < n2) {
      const int h = e / H, jj = e - h * H;
      for (int n = 0; n < N; ++n) g = __builtin_fmaf(D2[n * H + h], H1[n * H + jj], g);
      p = a.w2; m = a.w2m; v = a.w2v;
    } else if ((e -= n2) < H) {
      for (int n = 0; n < N; ++n) g += D2[n * H + e];
      p = a.b2; m = a.b2m; v = a.b2v;
    } else if ((e -= H) < H) {
      for (int n = 0; n < N; ++n) g = __builtin_fmaf(gs[n], H2[n * H + e], g);
      p = a.w3; m = a.w3m; v = a.w3v;
    } else {
      e = 0;
      for (int i = 0; i < a.B; ++i) g += gs[i] + gs[a.B + i];      // (pair by pair: exactly zero under a pairwise loss, as the other models' step kernels)
      p = a.b3; m = a.b3m; v = a.b3v;
    }
    if (m) step_adam(p + e, m + e, v + e, g, a.adam);
  }
}

size_t mlp_lds_floats(int B, int Q, int nt, int use_idf, int H) {
  const size_t N = 2 * (size_t)B, X = (size_t)Q * (nt + use_idf);
  return (size_t)H * X + (size_t)H * H + H + N * X + 4 * N * H + N + B + N;
}

struct PacrrStepLayout {
  size_t sim, conv_w, conv_b, top, pos, filt, gtop, dw, db, total;
};

PacrrStepLayout pacrr_step_layout(int B, int Q, int L, int mingram, int maxgram, int nf, int kmax) {
  PacrrStepLayout o{};
  const size_t N = 2 * (size_t)B, n_ng = maxgram - mingram + 1, nt = n_ng * kmax;
  size_t nw = 0;
  for (int g = mingram; g <= maxgram; ++g) nw += (size_t)nf * g * g;
  size_t at = 0;
  auto take = [&](size_t n) { const size_t here = at; at += (n + 3) & ~(size_t)3; return here; };
  o.sim = take(N * Q * L); o.conv_w = take(nw); o.conv_b = take(n_ng * nf); o.top = take(N * Q * nt); o.pos = take(N * Q * nt); o.filt = take(N * Q * nt);
  o.gtop = take(N * Q * nt); o.dw = take(nw); o.db = take(n_ng * nf);
  o.total = at;
  return o;
}

}  // namespace

extern "C" size_t capamd_pacrr_train_step_workspace_floats(int B, int Q, int L, int mingram, int maxgram, int nfilters, int kmax) {
  if (B < 1 || Q < 1 || L < 1 || mingram < 1 || maxgram < mingram || maxgram > kTrMaxNg || nfilters < 1 || kmax < 1) return 0;
  return pacrr_step_layout(B, Q, L, mingram, maxgram, nfilters, kmax).total;
}

extern "C" int capamd_pacrr_train_step(const int64_t* q_ids, const int64_t* d_ids, const float* idf, int B, int Q, int L, const float* packed, int64_t V,
                                       int D, int mingram, int maxgram, int nfilters, int kmax, int use_idf, int combine, int nonlinearity,
                                       float* const* ptrs, int loss_type, float step_size, float one_minus_beta1, float beta2, float eps,
                                       float bc2_sqrt, float* loss_out, float* workspace, size_t workspace_floats, int* status, void* stream) {
  if (!q_ids || !d_ids || !packed || !ptrs || !loss_out || !workspace || !status || (use_idf && !idf)) return CAPAMD_ERR_ARG;
  if (B < 1 || mingram < 1 || maxgram < mingram || maxgram > kTrMaxNg || kmax < 1 || kmax > kTrMaxK || Q < 1 || Q > kTrMaxQ || L < kmax || L > kTrMaxL ||
      nfilters < 1 || nfilters > kTrMaxF || combine < 1 || combine > kStepMaxH || nonlinearity < 0 || nonlinearity > 2 || loss_type < 0 || loss_type > 1 ||
      !(bc2_sqrt > 0.f))
    return CAPAMD_ERR_ARG;
  const int n_ng = maxgram - mingram + 1, nt = n_ng * kmax, N = 2 * B, P = 2 * n_ng + 6;
  const size_t lds_floats = mlp_lds_floats(B, Q, nt, use_idf ? 1 : 0, combine);
  if (lds_floats > kStepLdsFloats) return CAPAMD_ERR_ARG;          // (the one-workgroup stage keeps the batch's activations in LDS)
  if (reinterpret_cast<uintptr_t>(workspace) & 15) return CAPAMD_ERR_ALIGN;
  const PacrrStepLayout o = pacrr_step_layout(B, Q, L, mingram, maxgram, nfilters, kmax);
  if (workspace_floats < o.total) return CAPAMD_ERR_WORKSPACE;
  for (int i = 0; i < P; ++i)
    if (!ptrs[i]) return CAPAMD_ERR_ARG;
  ConvParams cp{};
  cp.mingram = mingram; cp.n_ng = n_ng; cp.nf = nfilters;
  for (int g = 0; g < n_ng; ++g) {
    cp.w[g] = ptrs[2 * g]; cp.wm[g] = ptrs[P + 2 * g]; cp.wv[g] = ptrs[2 * P + 2 * g];
    cp.b[g] = ptrs[2 * g + 1]; cp.bm[g] = ptrs[P + 2 * g + 1]; cp.bv[g] = ptrs[2 * P + 2 * g + 1];
  }
  const StepAdam adam{step_size, one_minus_beta1, beta2, eps, bc2_sqrt};
  hipStream_t s = (hipStream_t)stream;
  float* w = workspace;
  int32_t* pos = reinterpret_cast<int32_t*>(w + o.pos);
  int32_t* filt = reinterpret_cast<int32_t*>(w + o.filt);
  (void)hipGetLastError();
  const int nw_max = nfilters * maxgram * maxgram + nfilters;
  hipLaunchKernelGGL(pacrr_conv_params_kernel, dim3((nw_max + 255) / 256, n_ng), dim3(256), 0, s, cp, w + o.conv_w, w + o.conv_b, nullptr, nullptr, 0, adam);
  int rc = capamd_similarity_matrix(q_ids, d_ids, N, Q, L, packed, V, D, w + o.sim, status, stream);
  if (rc != CAPAMD_OK) return rc;
  rc = capamd_pacrr_convmax_forward(w + o.sim, N, Q, L, mingram, maxgram, nfilters, kmax, w + o.conv_w, w + o.conv_b, w + o.top, pos, filt, stream);
  if (rc != CAPAMD_OK) return rc;
  const int b0 = 2 * n_ng;
  MlpArgs ma{w + o.top, use_idf ? idf : nullptr, B, Q, nt, use_idf ? 1 : 0, combine, nonlinearity, loss_type,
             ptrs[b0], ptrs[P + b0], ptrs[2 * P + b0], ptrs[b0 + 1], ptrs[P + b0 + 1], ptrs[2 * P + b0 + 1],
             ptrs[b0 + 2], ptrs[P + b0 + 2], ptrs[2 * P + b0 + 2], ptrs[b0 + 3], ptrs[P + b0 + 3], ptrs[2 * P + b0 + 3],
             ptrs[b0 + 4], ptrs[P + b0 + 4], ptrs[2 * P + b0 + 4], ptrs[b0 + 5], ptrs[P + b0 + 5], ptrs[2 * P + b0 + 5],
             adam, w + o.gtop, loss_out};
  const size_t lds = lds_floats * 4;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(pacrr_mlp_step_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return CAPAMD_ERR_LAUNCH;
  hipLaunchKernelGGL(pacrr_mlp_step_kernel, dim3(1), dim3(256), lds, s, ma);
  rc = capamd_pacrr_convmax_backward(w + o.sim, N, Q, L, mingram, maxgram, nfilters, kmax, w + o.gtop, pos, filt, w + o.dw, w + o.db, stream);
  if (rc != CAPAMD_OK) return rc;
  hipLaunchKernelGGL(pacrr_conv_params_kernel, dim3((nw_max + 255) / 256, n_ng), dim3(256), 0, s, cp, nullptr, nullptr, w + o.dw, w + o.db, 1, adam);
  return hipGetLastError() == hipSuccess ? CAPAMD_OK : CAPAMD_ERR_LAUNCH;
}
