// One ConvKNRM training step on the device, for gfx950.
//
// Reference: the loop body of PytorchTrainer.single_train_iteration, capreolus/trainer/pytorch.py:93-108 - reranker.score(batch) on the
// positive and the negative documents (ConvKNRM.score -> ConvKNRM_class.forward, capreolus/reranker/ConvKNRM.py:42-77), self.loss(...)
// (pair_hinge_loss / pair_softmax_loss, reranker/common.py:96-103), loss.backward(), self.optimizer.step() (torch.optim.Adam, plain).
//
// The heavy stages are the kernels of ngram_conv.hip (the n-gram convolutions over the frozen table, forward and weight gradient) and
// kernel_pool.hip (cosine similarity of every view pair + RBF kernel pooling, forward and backward), called through their C entry points.
// What this file adds is everything that the autograd route leaves to ~60 ATen nodes per step (stack / unstack of the 2 K kernel
// parameters, the combine layer and its backward, the loss, the reductions of the pooling backward's partial results, Adam's
// multi-tensor kernels):
//   convknrm_kernels_kernel   the K (mu, sigma) scalars - one nn.Parameter each in the reference's state_dict, common.py:229-230 - into two
//                             arrays for the pooling kernels
//   convknrm_tail_kernel      ONE workgroup: scores of the 2 B documents from their features (single Linear, optional tanh), the pairwise
//                             loss, d loss / d features for the pooling backward, the Linear's gradient summed in pair order, Adam on it
//   convknrm_gather_kernel    the pooling backward's per-(view, chunk) partial d a_q summed in their order into d qrep; its per-block
//                             d mu / d sigma partials summed in order, Adam on the kernel parameters
//   convknrm_adam_kernel      Adam over the convolution weights and biases from the gradients ngram_conv's backward wrote
// Adam as torch.optim.Adam's non-capturable single-tensor path (torch/optim/adam.py): exp_avg.lerp_(g, 1 - beta1);
// exp_avg_sq.mul_(beta2).addcmul_(g, g, 1 - beta2); denom = sqrt(exp_avg_sq) / sqrt(bias_correction2) + eps;
// param.addcdiv_(exp_avg, denom, value = -lr / bias_correction1) - the step scalars come from the host, computed in double.
#include "capreolus_amd.h"
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

constexpr int kCkMaxK = 16, kCkMaxG = 4, kCkMaxBatch = 512, kCkMaxFeat = kCkMaxK * kCkMaxG * kCkMaxG;

struct AdamScalars {
  float step_size, one_minus_beta1, beta2, eps, bc2_sqrt;
};

__device__ __forceinline__ void adam_update(float* p, float* m, float* v, float g, const AdamScalars& s) {
  float mm = *m, vv = *v;
  mm = mm + (g - mm) * s.one_minus_beta1;
  vv = vv * s.beta2 + (1.f - s.beta2) * (g * g);
  *m = mm;
  *v = vv;
  const float denom = sqrtf(vv) / s.bc2_sqrt + s.eps;
  *p = *p - s.step_size * (mm / denom);
}

// parameter slots of the pointer table (each slot: parameter, exp_avg, exp_avg_sq; a null exp_avg = not trained)
struct ParamTable {
  float* p[2 * kCkMaxK + 2 * kCkMaxG + 2];
  float* m[2 * kCkMaxK + 2 * kCkMaxG + 2];
  float* v[2 * kCkMaxK + 2 * kCkMaxG + 2];
};

__global__ void convknrm_kernels_kernel(ParamTable t, int K, float* mu, float* sigma) {
  const int k = threadIdx.x;
  if (k < K) {
    mu[k] = *t.p[k];
    sigma[k] = *t.p[K + k];
  }
}

struct TailArgs {
  const float* feat;     // [2 B, NF]: the positive documents' rows, then the negative ones
  int B, NF;
  float *w, *wm, *wv;    // combine.0.weight [NF] and its moments
  float *b, *bm, *bv;    // combine.0.bias [1]
  int scoretanh, loss_type;
  AdamScalars adam;
  float* gfeat;          // [2 B, NF]
  float* loss_out;
};

__global__ __launch_bounds__(256) void convknrm_tail_kernel(TailArgs a) {
  __shared__ float W[kCkMaxFeat + 1], gsc[2][kCkMaxBatch], lsum[kCkMaxBatch];
  const int tid = threadIdx.x, NF = a.NF;
  for (int i = tid; i < NF; i += 256) W[i] = a.w[i];
  if (tid == 0) W[NF] = a.b[0];
  __syncthreads();
  const float inv_b = 1.f / (float)a.B;
  for (int i = tid; i < a.B; i += 256) {
    float sc[2], dt[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float* f = a.feat + ((int64_t)h * a.B + i) * NF;
      float v = W[NF];
      for (int k = 0; k < NF; ++k) v = __builtin_fmaf(W[k], f[k], v);
      dt[h] = 1.f;
      if (a.scoretanh) {
        v = tanhf(v);
        dt[h] = 1.f - v * v;
      }
      sc[h] = v;
    }
    float li, gp, gn;      // loss of the pair, d loss / d score of its positive / negative document
    if (a.loss_type == 0) {
      const float mrg = 1.f - (sc[0] - sc[1]);
      li = fmaxf(mrg, 0.f);
      const float on = mrg >= 0.f ? inv_b : 0.f;      // (torch.clamp's backward passes the gradient at the boundary)
      gp = -on;
      gn = on;
    } else {
      const float mx = fmaxf(sc[0], sc[1]), e0 = expf(sc[0] - mx), e1 = expf(sc[1] - mx), p0 = e0 / (e0 + e1), p1 = e1 / (e0 + e1);
      li = 1.f - p0;
      gp = -p0 * p1 * inv_b;
      gn = p0 * p1 * inv_b;
    }
    gsc[0][i] = gp * dt[0];
    gsc[1][i] = gn * dt[1];
    lsum[i] = li;
  }
  __syncthreads();
  // d loss / d features, with the weights as they were BEFORE this step's update (W is the copy)
  for (int i = tid; i < 2 * a.B * NF; i += 256) {
    const int row = i / NF, k = i - row * NF;
    a.gfeat[i] = gsc[row >= a.B][row >= a.B ? row - a.B : row] * W[k];
  }
  // one thread per element of the Linear (and one for the loss): its gradient summed over the batch in pair order, then Adam
  for (int j = tid; j < NF + 2; j += 256) {
    if (j == NF + 1) {
      float l = 0.f;
      for (int i = 0; i < a.B; ++i) l += lsum[i];
      a.loss_out[0] = l * inv_b;
      continue;
    }
    float g = 0.f;
    if (j < NF)
      for (int i = 0; i < a.B; ++i) g += gsc[0][i] * a.feat[(int64_t)i * NF + j] + gsc[1][i] * a.feat[((int64_t)a.B + i) * NF + j];
    else
      for (int i = 0; i < a.B; ++i) g += gsc[0][i] + gsc[1][i];
    if (j < NF) {
      if (a.wm) adam_update(a.w + j, a.wm + j, a.wv + j, g, a.adam);
    } else if (a.bm) {
      adam_update(a.b, a.bm, a.bv, g, a.adam);
    }
  }
}

struct GatherArgs {
  const float* dq_part;   // [N, GD, C, T, F]
  float* dqrep;           // [N, GQ, Q, F]
  int N, GQ, GD, C, Q, F, cross;
  const float* dmu_part;  // [N GD C, K]
  const float* dsg_part;
  int K;
  ParamTable t;
  AdamScalars adam;
};

__global__ __launch_bounds__(256) void convknrm_gather_kernel(GatherArgs a) {
  const int tid = threadIdx.x;
  if (blockIdx.x == gridDim.x - 1) {
    // the kernels' parameters: column j of the [rows, 2 K] partials (mu columns, then sigma columns), rows in eight slices whose sums are
    // added in order
    __shared__ float part[8][2 * kCkMaxK];
    const int rows = a.N * a.GD * a.C, j = tid & 31, sl = tid >> 5;
    if (j < 2 * a.K) {
      const float* src = j < a.K ? a.dmu_part + j : a.dsg_part + (j - a.K);
      const int lo = (int)((int64_t)rows * sl / 8), hi = (int)((int64_t)rows * (sl + 1) / 8);
      float g = 0.f;
      for (int r = lo; r < hi; ++r) g += src[(int64_t)r * a.K];
      part[sl][j] = g;
    }
    __syncthreads();
    if (tid < 2 * a.K && a.t.m[tid]) {
      float g = 0.f;
      for (int s = 0; s < 8; ++s) g += part[s][tid];
      adam_update(a.t.p[tid], a.t.m[tid], a.t.v[tid], g, a.adam);
    }
    return;
  }
  const int64_t total = (int64_t)a.N * a.GQ * a.Q * a.F;
  const int T = (a.cross ? a.GQ : 1) * a.Q;
  for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < total; i += (int64_t)(gridDim.x - 1) * 256) {
    const int f = (int)(i % a.F);
    int64_t r = i / a.F;
    const int q = (int)(r % a.Q);
    r /= a.Q;
    const int gq = (int)(r % a.GQ);
    const int64_t n = r / a.GQ;
    float g = 0.f;
    if (a.cross) {
      for (int gd = 0; gd < a.GD; ++gd)
        for (int c = 0; c < a.C; ++c) g += a.dq_part[((((n * a.GD + gd) * a.C + c) * T) + gq * a.Q + q) * a.F + f];
    } else {
      for (int c = 0; c < a.C; ++c) g += a.dq_part[((((n * a.GD + gq) * a.C + c) * T) + q) * a.F + f];
    }
    a.dqrep[i] = g;
  }
}

struct AdamSegments {
  float *p[2 * kCkMaxG], *m[2 * kCkMaxG], *v[2 * kCkMaxG];
  const float* g[2 * kCkMaxG];
  int n[2 * kCkMaxG];
  AdamScalars adam;
};

__global__ __launch_bounds__(256) void convknrm_adam_kernel(AdamSegments a) {
  const int s = blockIdx.y;
  if (!a.m[s]) return;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < a.n[s]; i += gridDim.x * 256) adam_update(a.p[s] + i, a.m[s] + i, a.v[s] + i, a.g[s][i], a.adam);
}

size_t align4(size_t n) { return (n + 3) & ~(size_t)3; }

struct StepLayout {
  size_t mu, sigma, qrep, drep, feat, ksum, rowsum, chunk_sums, gfeat, dq_part, dd, dmu_part, dsg_part, dqrep, dw[kCkMaxG], db[kCkMaxG], conv, total, conv_floats;
};

StepLayout step_layout(int B, int Q, int L, int D, int G, int F, int K, int crossmatch) {
  StepLayout o{};
  const size_t N = 2 * (size_t)B;
  const int V = crossmatch ? G * G : G, T = (crossmatch ? G : 1) * Q, C = capamd_kernel_pool_chunks(L);
  size_t at = 0;
  auto take = [&](size_t n) { const size_t here = at; at += align4(n); return here; };
  o.mu = take(K); o.sigma = take(K);
  o.qrep = take(N * G * Q * F); o.drep = take(N * G * L * F);
  o.feat = take(N * K * V); o.ksum = take(N * G * T * K); o.rowsum = take(N * G * T); o.chunk_sums = take(N * G * C * T * (K + 1));
  o.gfeat = take(N * K * V);
  o.dq_part = take(N * G * C * T * F); o.dd = take(N * G * L * F); o.dmu_part = take(N * G * C * K); o.dsg_part = take(N * G * C * K);
  o.dqrep = take(N * G * Q * F);
  for (int g = 0; g < G; ++g) { o.dw[g] = take((size_t)F * D * (g + 1)); o.db[g] = take(F); }
  const size_t cf = capamd_ngram_conv_workspace_floats(D, G, F, 0), cb = capamd_ngram_conv_workspace_floats(D, G, F, 1);
  o.conv_floats = cf > cb ? cf : cb;
  o.conv = take(o.conv_floats);
  o.total = at;
  return o;
}

}  // namespace

extern "C" size_t capamd_convknrm_train_step_workspace_floats(int B, int Q, int L, int D, int G, int F, int K, int crossmatch) {
  if (B < 1 || Q < 1 || L < 1 || D < 1 || G < 1 || G > kCkMaxG || F < 1 || K < 1 || K > kCkMaxK) return 0;
  return step_layout(B, Q, L, D, G, F, K, crossmatch).total;
}

extern "C" int capamd_convknrm_train_step(const int64_t* q_ids, const int64_t* d_ids, int B, int Q, int L, const float* emb, int64_t V, int D, int G,
                                          int F, int K, int crossmatch, float* const* ptrs, int scoretanh, int loss_type, float step_size,
                                          float one_minus_beta1, float beta2, float eps, float bc2_sqrt, float* loss_out, float* workspace,
                                          size_t workspace_floats, int* status, void* stream) {
  if (!q_ids || !d_ids || !emb || !ptrs || !loss_out || !workspace || !status) return CAPAMD_ERR_ARG;
  if (B < 1 || B > kCkMaxBatch || G < 1 || G > kCkMaxG || K < 1 || K > kCkMaxK || loss_type < 0 || loss_type > 1 || !(bc2_sqrt > 0.f)) return CAPAMD_ERR_ARG;
  if (reinterpret_cast<uintptr_t>(workspace) & 15) return CAPAMD_ERR_ALIGN;
  const StepLayout o = step_layout(B, Q, L, D, G, F, K, crossmatch);
  if (workspace_floats < o.total) return CAPAMD_ERR_WORKSPACE;
  const int P = 2 * K + 2 * G + 2, N = 2 * B, NV = crossmatch ? G * G : G, T = (crossmatch ? G : 1) * Q, C = capamd_kernel_pool_chunks(L);
  ParamTable t{};
  for (int i = 0; i < P; ++i) {
    t.p[i] = ptrs[i];
    t.m[i] = ptrs[P + i];
    t.v[i] = ptrs[2 * P + i];
    if (!t.p[i]) return CAPAMD_ERR_ARG;
  }
  const AdamScalars adam{step_size, one_minus_beta1, beta2, eps, bc2_sqrt};
  hipStream_t s = (hipStream_t)stream;
  float* w = workspace;
  (void)hipGetLastError();
  hipLaunchKernelGGL(convknrm_kernels_kernel, dim3(1), dim3(64), 0, s, t, K, w + o.mu, w + o.sigma);
  const float* cw[kCkMaxG];
  const float* cb[kCkMaxG];
  float* dcw[kCkMaxG];
  float* dcb[kCkMaxG];
  for (int g = 0; g < G; ++g) {
    cw[g] = t.p[2 * K + 2 * g];
    cb[g] = t.p[2 * K + 2 * g + 1];
    dcw[g] = w + o.dw[g];
    dcb[g] = w + o.db[g];
  }
  int rc = capamd_ngram_conv_forward(q_ids, d_ids, N, Q, L, emb, V, D, cw, cb, G, F, w + o.qrep, w + o.drep, w + o.conv, o.conv_floats, status, stream);
  if (rc != CAPAMD_OK) return rc;
  rc = capamd_kernel_pool_forward(w + o.qrep, w + o.drep, q_ids, d_ids, N, G, G, Q, L, F, crossmatch, w + o.mu, w + o.sigma, K, w + o.feat, w + o.ksum,
                                  w + o.rowsum, w + o.chunk_sums, stream);
  if (rc != CAPAMD_OK) return rc;
  TailArgs ta{w + o.feat, B, K * NV, t.p[P - 2], t.m[P - 2], t.v[P - 2], t.p[P - 1], t.m[P - 1], t.v[P - 1], scoretanh, loss_type, adam, w + o.gfeat, loss_out};
  hipLaunchKernelGGL(convknrm_tail_kernel, dim3(1), dim3(256), 0, s, ta);
  rc = capamd_kernel_pool_backward(w + o.qrep, w + o.drep, q_ids, d_ids, N, G, G, Q, L, F, crossmatch, w + o.mu, w + o.sigma, K, w + o.gfeat, w + o.ksum,
                                   w + o.rowsum, w + o.dq_part, w + o.dd, w + o.dmu_part, w + o.dsg_part, stream);
  if (rc != CAPAMD_OK) return rc;
  GatherArgs ga{w + o.dq_part, w + o.dqrep, N, G, G, C, Q, F, crossmatch ? 1 : 0, w + o.dmu_part, w + o.dsg_part, K, t, adam};
  const int64_t elems = (int64_t)N * G * Q * F;
  hipLaunchKernelGGL(convknrm_gather_kernel, dim3((unsigned)((elems + 255) / 256 < 1024 ? (elems + 255) / 256 : 1024) + 1), dim3(256), 0, s, ga);
  rc = capamd_ngram_conv_backward(q_ids, d_ids, N, Q, L, emb, V, D, G, F, w + o.dqrep, w + o.dd, dcw, dcb, w + o.conv, o.conv_floats, status, stream);
  if (rc != CAPAMD_OK) return rc;
  AdamSegments sg{};
  int longest = 1;
  for (int g = 0; g < G; ++g) {
    const int iw = 2 * K + 2 * g, ib = iw + 1;
    sg.p[2 * g] = t.p[iw]; sg.m[2 * g] = t.m[iw]; sg.v[2 * g] = t.v[iw]; sg.g[2 * g] = dcw[g]; sg.n[2 * g] = F * D * (g + 1);
    sg.p[2 * g + 1] = t.p[ib]; sg.m[2 * g + 1] = t.m[ib]; sg.v[2 * g + 1] = t.v[ib]; sg.g[2 * g + 1] = dcb[g]; sg.n[2 * g + 1] = F;
    longest = sg.n[2 * g] > longest ? sg.n[2 * g] : longest;
  }
  sg.adam = adam;
  hipLaunchKernelGGL(convknrm_adam_kernel, dim3((longest + 255) / 256 < 256 ? (longest + 255) / 256 : 256, 2 * G), dim3(256), 0, s, sg);
  return hipGetLastError() == hipSuccess ? CAPAMD_OK : CAPAMD_ERR_LAUNCH;
}
