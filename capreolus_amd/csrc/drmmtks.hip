// Fused DRMM-TKS forward for gfx950 (a sibling model on the same fused front end).
//
// Reference semantics: DRMMTKS_class.forward (capreolus/reranker/DRMMTKS.py:50-64): SimilarityMatrix -> per query term
// the top-k similarities over ALL document positions (pads contribute their 0) -> Linear(k, 1) + tanh -> IDF term gate
// (softmax over query terms, pads masked with -1e7) -> output layer.
//
// Same work layout as knrm.hip / drmm.hip (one workgroup per pair, 16 lanes per document term, real terms compacted in
// LDS, query rows in an LDS copy).  Back end: every lane keeps a sorted top-k of the similarities of the query term it
// owns (sorted_insert in registers: one v_med3_f32 per element); the 16 groups' lists are merged by one wave per query term (k rounds of
// a wave-wide arg-max over the list heads), together with the closed-form candidates of the terms that were never
// gathered: n1 ones (OOV exact matches) and n0 zeros (pads and other OOV terms).
#include "capreolus_amd.h"
#include "interaction.h"

using namespace capamd;

namespace {

constexpr int kMaxTopK = 16;
constexpr int kMaxQ = 32;

struct TksArgs {
  IdSource ids;
  const float* idf;
  int B, Q, L;
  const float* packed;
  int64_t V;
  int topk;
  const float* gate_w;   // [1]  (IDF gate)
  const float* ffw_w;    // [topk]
  const float* ffw_b;    // [1]
  const float* out_w;    // [1]
  const float* out_b;    // [1]
  float* out;
  int* status;
  float* feat;           // training-step mode (capamd_drmmtks_features): [B][Q][topk] sorted top-k similarities instead of scores
  const int64_t* d64_b;  // training step: a second block of documents - pairs split .. B - 1 take query row (b - split) and row (b - split) of d64_b
  int split;
};

#ifndef CAPAMD_TKS_U
#define CAPAMD_TKS_U 1
#endif
#ifndef CAPAMD_TKS_WAVES
#define CAPAMD_TKS_WAVES 6   // measured per 64,000 pairs: 5 -> 2.20 ms, 6 -> 2.12 ms; two rows in flight per group (CAPAMD_TKS_U 2) spill and lose 2-5x
#endif

// KT = length of the per-lane sorted lists (>= topk, a multiple of 4): an insertion is KT independent v_med3_f32
template <int NV, int KT>
__global__ __launch_bounds__(kThreads, CAPAMD_TKS_WAVES) void drmmtks_forward_kernel(TksArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  int* tok = reinterpret_cast<int*>(smem_raw);
  const int tok_cap = (a.L + 3) & ~3;
  float* lists = reinterpret_cast<float*>(tok + tok_cap);            // [16 groups][kQT][kMaxTopK]
  float* zlds = lists + kGroupsPerWG * kQT * kMaxTopK;               // [kMaxQ]
  float* glds = zlds + kMaxQ;                                        // [kMaxQ]
  int* wave_cnt = reinterpret_cast<int*>(glds + kMaxQ);              // [48]: distinct_terms' per-wave counts
  int* n_one = wave_cnt + 48;                                        // [kQT] (+4 spare)
  float4* qlds = reinterpret_cast<float4*>(n_one + 8);               // [kQT][NV*16] float4
  int* mult = reinterpret_cast<int*>(qlds + kQT * kMaxNV * 16);      // [tok_cap] multiplicity of tok[k]
  int* hkey = mult + tok_cap;                                        // [kHashSlots] phase 1 only
  int* hfirst = hkey + kHashSlots;                                   // [kHashSlots] phase 1 only

  const int tid = threadIdx.x, lane16 = tid & 15, g = tid >> 4, wave = tid >> 6, lane = tid & 63;
  const int b = blockIdx.x, K = a.topk;
  PairIds ids = pair_ids(a.ids, a.d64_b && b >= a.split ? b - a.split : b, a.Q, a.L);
  if (a.d64_b && b >= a.split) ids.d64 = a.d64_b + (int64_t)(b - a.split) * a.L;

  // the document's distinct real terms with their multiplicities (interaction.h: distinct_terms): a repeated term is gathered once
  // and its similarity enters the top-k lists as many times as the document repeats it (at most k copies can matter)
  const TermList tl = distinct_terms(ids, a.L, a.V, a.status, tok, mult, hkey, hfirst, wave_cnt);
  const int n_real = tl.n_unique;
  const int n_nonreal = a.L - tl.n_real;

  for (int q0 = 0; q0 < a.Q; q0 += kQT) {
    QueryPass<NV> qp;
    load_query_pass_lds<NV>(a.packed, ids, a.Q, q0, a.V, tid, kThreads, lane16, qlds, qp, a.status);
    if (tid < kQT) n_one[tid] = 0;
    __syncthreads();
    {
      bool any_oov_q = false;
#pragma unroll
      for (int t = 0; t < kQT; ++t) any_oov_q |= qp.id[t] < 0;
      if (any_oov_q)
        for (int j = tid; j < a.L; j += kThreads) {
          const int64_t did = ids.d(j);
          if (did < 0) {
#pragma unroll
            for (int t = 0; t < kQT; ++t)
              if (qp.id[t] == (int)did && did > -2147483648LL) atomicAdd(&n_one[t], 1);
          }
        }
    }
    // sorted (descending) top-k of this lane's query term over the terms its group visits
    float top[KT];
#pragma unroll
    for (int i = 0; i < KT; ++i) top[i] = -INFINITY;
    for (int t0 = g; t0 < n_real; t0 += CAPAMD_TKS_U * kGroupsPerWG) {   // CAPAMD_TKS_U rows in flight per 16-lane group
      RowRegs<NV> d[CAPAMD_TKS_U];
      bool has[CAPAMD_TKS_U];
#pragma unroll
      for (int u = 0; u < CAPAMD_TKS_U; ++u) {
        const int tu = t0 + u * kGroupsPerWG;
        has[u] = tu < n_real;
        load_row<NV>(a.packed, has[u] ? tok[tu] : 0, lane16, d[u]);
      }
      float x[CAPAMD_TKS_U];
      int qoff = 0;
      asm volatile("" : "+v"(qoff));
      rows_sim_my<NV, CAPAMD_TKS_U, true>(d, qp, qlds + qoff, lane16, x);
#pragma unroll
      for (int u = 0; u < CAPAMD_TKS_U; ++u) {
        const int copies = has[u] ? min(mult[t0 + u * kGroupsPerWG], K) : 0;   // uniform over the 16 lanes of the group
        for (int c = 0; c < copies; ++c) {
          sorted_insert<KT>(top, x[u]);
        }
      }
    }
    if (lane16 < kQT) {
#pragma unroll
      for (int i = 0; i < KT; ++i) lists[(g * kQT + lane16) * kMaxTopK + i] = top[i];
    }
    __syncthreads();

    // ---- merge: wave w owns query term q0 + w.  lanes 0..15: the groups' lists; lane 16: n1 ones; lane 17: n0 zeros
    const int q = q0 + wave;
    if (q < a.Q) {
      const int no = n_one[wave], nz = n_nonreal - no;
      int head = 0;
      // (what the merge rounds and the gate need from global memory, requested together in front of them: the rounds used to load
      // ffw_w[r] one by one - K dependent memory round trips per query term - and the gate its three operands after the last round)
      const float ffw_l = (!a.feat && lane < K) ? a.ffw_w[lane] : 0.f;
      const float gate_w0 = a.feat ? 0.f : a.gate_w[0], idf_q = a.feat ? 0.f : a.idf[(int64_t)ids.qrow * a.Q + q];
      const bool q_pad = ids.q(q) == 0;
      float acc = a.feat ? 0.f : a.ffw_b[0];
      for (int r = 0; r < K; ++r) {
        float cand = -INFINITY;
        if (lane < kGroupsPerWG) cand = head < KT ? lists[(lane * kQT + wave) * kMaxTopK + head] : -INFINITY;
        else if (lane == 16) cand = head < no ? 1.f : -INFINITY;
        else if (lane == 17) cand = head < nz ? 0.f : -INFINITY;
        const float best = wave_allreduce_max(cand);
        const unsigned long long who = __ballot(cand == best && best > -INFINITY);
        if (who == 0) break;  // fewer than k candidates (L < k): torch.topk would raise; the host checks L >= k
        const int winner = __ffsll((long long)who) - 1;
        if (lane == winner) ++head;
        if (a.feat) {
          if (lane == 0) a.feat[((int64_t)b * a.Q + q) * K + r] = best;
        } else {
          acc = __builtin_fmaf(__shfl(ffw_l, r, 64), best, acc);   // DRMMTKS.py:22: Linear(topk, 1) on the sorted values
        }
      }
      if (lane == 0 && !a.feat) {
        zlds[q] = tanhf(acc);
        float gl = gate_w0 * idf_q;
        if (q_pad) gl += -1e7f;   // DRMMTKS.py:38
        glds[q] = gl;
      }
    }
    __syncthreads();
  }

  if (tid == 0 && !a.feat) {  // softmax gate + output layer (DRMMTKS.py:47-48, :60-62)
    float m = glds[0];
    for (int q = 1; q < a.Q; ++q) m = fmaxf(m, glds[q]);
    float den = 0.f, num = 0.f;
    for (int q = 0; q < a.Q; ++q) {
      const float e = expf(glds[q] - m);
      den += e;
      num = __builtin_fmaf(e, zlds[q], num);
    }
    a.out[b] = __builtin_fmaf(a.out_w[0], num / den, a.out_b[0]);
  }
}

}  // namespace

extern "C" int capamd_drmmtks_forward(const int64_t* q_ids, const int64_t* d_ids, const float* idf, int B, int Q, int L,
                                      const float* packed, int64_t V, int D, int topk, const float* gate_w, const float* ffw_w,
                                      const float* ffw_b, const float* out_w, const float* out_b, float* out, int* status,
                                      void* stream) {
  if (B == 0) return CAPAMD_OK;
  if (!q_ids || !d_ids || !idf || !packed || !gate_w || !ffw_w || !ffw_b || !out_w || !out_b || !out || !status) return CAPAMD_ERR_ARG;
  if (B < 0 || Q < 1 || Q > kMaxQ || L < 1 || L > 32768 || V < 1 || V > 0x7fffffffLL) return CAPAMD_ERR_ARG;
  if (topk < 1 || topk > kMaxTopK || topk > L || capamd_packed_row_stride(D) < 0) return CAPAMD_ERR_ARG;
  const IdSource ids{q_ids, d_ids, nullptr, nullptr, nullptr, nullptr};
  TksArgs a{ids, idf, B, Q, L, packed, V, topk, gate_w, ffw_w, ffw_b, out_w, out_b, out, status, nullptr};
  const size_t smem = (size_t)((L + 3) & ~3) * 8 + (size_t)(kGroupsPerWG * kQT * kMaxTopK + 2 * kMaxQ + 56 + 2 * kHashSlots) * 4 + (size_t)kQT * kMaxNV * 16 * 16;
  hipStream_t s = (hipStream_t)stream;
  (void)hipGetLastError();
#define LAUNCH(NV_)                                                                                                                   \
  do {                                                                                                                                \
    if (topk <= 4) hipLaunchKernelGGL((drmmtks_forward_kernel<NV_, 4>), dim3(B), dim3(kThreads), smem, s, a);                          \
    else if (topk <= 8) hipLaunchKernelGGL((drmmtks_forward_kernel<NV_, 8>), dim3(B), dim3(kThreads), smem, s, a);                     \
    else if (topk <= 12) hipLaunchKernelGGL((drmmtks_forward_kernel<NV_, 12>), dim3(B), dim3(kThreads), smem, s, a);                   \
    else hipLaunchKernelGGL((drmmtks_forward_kernel<NV_, 16>), dim3(B), dim3(kThreads), smem, s, a);                                   \
  } while (0)
  switch (nv_for_dim(D)) {
    case 1: LAUNCH(1); break;
    case 2: LAUNCH(2); break;
    case 3: LAUNCH(3); break;
    case 4: LAUNCH(4); break;
    default: LAUNCH(5); break;
  }
#undef LAUNCH
  return hipGetLastError() == hipSuccess ? CAPAMD_OK : CAPAMD_ERR_LAUNCH;
}

static int drmmtks_features_launch(TksArgs a, int D, void* stream) {
  const int B = a.B, L = a.L, topk = a.topk;
  const size_t smem = (size_t)((L + 3) & ~3) * 8 + (size_t)(kGroupsPerWG * kQT * kMaxTopK + 2 * kMaxQ + 56 + 2 * kHashSlots) * 4 + (size_t)kQT * kMaxNV * 16 * 16;
  hipStream_t s = (hipStream_t)stream;
  (void)hipGetLastError();
#define LAUNCH(NV_)                                                                                                                   \
  do {                                                                                                                                \
    if (topk <= 4) hipLaunchKernelGGL((drmmtks_forward_kernel<NV_, 4>), dim3(B), dim3(kThreads), smem, s, a);                          \
    else if (topk <= 8) hipLaunchKernelGGL((drmmtks_forward_kernel<NV_, 8>), dim3(B), dim3(kThreads), smem, s, a);                     \
    else if (topk <= 12) hipLaunchKernelGGL((drmmtks_forward_kernel<NV_, 12>), dim3(B), dim3(kThreads), smem, s, a);                   \
    else hipLaunchKernelGGL((drmmtks_forward_kernel<NV_, 16>), dim3(B), dim3(kThreads), smem, s, a);                                   \
  } while (0)
  switch (nv_for_dim(D)) {
    case 1: LAUNCH(1); break;
    case 2: LAUNCH(2); break;
    case 3: LAUNCH(3); break;
    case 4: LAUNCH(4); break;
    default: LAUNCH(5); break;
  }
#undef LAUNCH
  return hipGetLastError() == hipSuccess ? CAPAMD_OK : CAPAMD_ERR_LAUNCH;
}

extern "C" int capamd_drmmtks_features(const int64_t* q_ids, const int64_t* d_ids, int B, int Q, int L, const float* packed, int64_t V,
                                       int D, int topk, float* features, int* status, void* stream) {
  if (B == 0) return CAPAMD_OK;
  if (!q_ids || !d_ids || !packed || !features || !status) return CAPAMD_ERR_ARG;
  if (B < 0 || Q < 1 || Q > kMaxQ || L < 1 || L > 32768 || V < 1 || V > 0x7fffffffLL) return CAPAMD_ERR_ARG;
  if (topk < 1 || topk > kMaxTopK || topk > L || capamd_packed_row_stride(D) < 0) return CAPAMD_ERR_ARG;
  const IdSource ids{q_ids, d_ids, nullptr, nullptr, nullptr, nullptr};
  TksArgs a{ids, nullptr, B, Q, L, packed, V, topk, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, status, features};
  return drmmtks_features_launch(a, D, stream);
}

// ---- one DRMM-TKS training step without a host round trip (reference trainer/pytorch.py:93-108) -------------------
// score() on the positive and the negative documents - top-k features (the kernel above over the 2 B documents), Linear(topk, 1) / tanh per
// query term, softmax idf gate, output layer (DRMMTKS.py:57-62) - the trainer's pairwise loss, backward through all of it, and
// torch.optim.Adam's update of the five parameter tensors in place, in two launches.  ptrs: DEVICE array of 3 x 5 device pointers - the
// parameters ffw.0.weight [topk], ffw.0.bias [1], gates.weight [1], output_layer.weight [1], output_layer.bias [1], then their exp_avg, then
// their exp_avg_sq.  The caller owns the step count: step_size = lr / (1 - beta1^t), bc2_sqrt = sqrt(1 - beta2^t), computed in double.
constexpr int kMaxStepBatch = 1024;

struct TksStepArgs {
  const float* feat;      // [2 B, Q, topk]: the positive documents, then the negative ones
  const int64_t* q_ids;   // [B, Q]
  const float* idf;       // [B, Q]
  int B, Q, topk;
  float* const* ptrs;
  int loss_type;
  float step_size, one_minus_beta1, beta2, eps, bc2_sqrt;
  float* loss_out;
  float* grads;           // [B][topk + 4] per-pair gradient contributions (workspace)
};

__global__ __launch_bounds__(256) void drmmtks_step_kernel(TksStepArgs a) {
  __shared__ float wf[kMaxTopK], sc_par[4], lsum[kMaxStepBatch];
  const int tid = threadIdx.x, K = a.topk, Q = a.Q, NP = K + 4;
  if (tid < K) wf[tid] = a.ptrs[0][tid];
  if (tid == 0) { sc_par[0] = a.ptrs[1][0]; sc_par[1] = a.ptrs[2][0]; sc_par[2] = a.ptrs[3][0]; sc_par[3] = a.ptrs[4][0]; }
  __syncthreads();
  const float bf = sc_par[0], wg = sc_par[1], wo = sc_par[2], bo = sc_par[3], inv_b = 1.f / (float)a.B;
  for (int i = tid; i < a.B; i += 256) {
    float z[2][kMaxQ], g[kMaxQ], sagg[2], score[2];
    // the gate: one per PAIR (the query is the same for both documents)
    float gl[kMaxQ], mx = -INFINITY;
    for (int q = 0; q < Q; ++q) {
      gl[q] = wg * a.idf[(int64_t)i * Q + q] + (a.q_ids[(int64_t)i * Q + q] == 0 ? -1e7f : 0.f);     // DRMMTKS.py:38
      mx = fmaxf(mx, gl[q]);
    }
    float den = 0.f;
    for (int q = 0; q < Q; ++q) { g[q] = expf(gl[q] - mx); den += g[q]; }
    for (int q = 0; q < Q; ++q) g[q] /= den;
    for (int h = 0; h < 2; ++h) {
      const float* T = a.feat + ((int64_t)(h * a.B + i) * Q) * K;
      float acc = 0.f;
      for (int q = 0; q < Q; ++q) {
        float v = bf;
        for (int j = 0; j < K; ++j) v = __builtin_fmaf(wf[j], T[q * K + j], v);
        z[h][q] = tanhf(v);
        acc = __builtin_fmaf(g[q], z[h][q], acc);
      }
      sagg[h] = acc;
      score[h] = __builtin_fmaf(wo, acc, bo);
    }
    float li, ds[2];
    if (a.loss_type == 0) {
      const float mrg = 1.f - (score[0] - score[1]);
      li = fmaxf(mrg, 0.f);
      const float on = mrg >= 0.f ? inv_b : 0.f;
      ds[0] = -on; ds[1] = on;
    } else {
      const float m2 = fmaxf(score[0], score[1]), e0 = expf(score[0] - m2), e1 = expf(score[1] - m2), p0 = e0 / (e0 + e1), p1 = e1 / (e0 + e1);
      li = 1.f - p0;
      ds[0] = -p0 * p1 * inv_b; ds[1] = p0 * p1 * inv_b;
    }
    lsum[i] = li;
    // backward: this pair's contribution to every parameter element
    float* G = a.grads + (int64_t)i * NP;
    float dwf[kMaxTopK], dbf = 0.f, dwg = 0.f, dwo = 0.f, dbo = 0.f;
    for (int j = 0; j < K; ++j) dwf[j] = 0.f;
    for (int h = 0; h < 2; ++h) {
      const float* T = a.feat + ((int64_t)(h * a.B + i) * Q) * K;
      dbo += ds[h];
      dwo = __builtin_fmaf(ds[h], sagg[h], dwo);
      const float dagg = ds[h] * wo;
      float dot = 0.f;                       // sum_r g_r dg_r with dg_r = dagg z_r
      for (int q = 0; q < Q; ++q) dot = __builtin_fmaf(g[q], dagg * z[h][q], dot);
      for (int q = 0; q < Q; ++q) {
        const float dgl = g[q] * (dagg * z[h][q] - dot);      // softmax backward
        dwg = __builtin_fmaf(dgl, a.idf[(int64_t)i * Q + q], dwg);
        const float da = dagg * g[q] * (1.f - z[h][q] * z[h][q]);
        dbf += da;
        for (int j = 0; j < K; ++j) dwf[j] = __builtin_fmaf(da, T[q * K + j], dwf[j]);
      }
    }
    for (int j = 0; j < K; ++j) G[j] = dwf[j];
    G[K] = dbf; G[K + 1] = dwg; G[K + 2] = dwo; G[K + 3] = dbo;
  }
  __syncthreads();      // (one workgroup: the barrier also orders its global writes for its own reads below)
  __threadfence_block();
  const int j = tid;
  if (j > NP) return;
  if (j == NP) {
    float l = 0.f;
    for (int i = 0; i < a.B; ++i) l += lsum[i];
    a.loss_out[0] = l * inv_b;
    return;
  }
  float gsum = 0.f;
  for (int i = 0; i < a.B; ++i) gsum += a.grads[(int64_t)i * NP + j];
  const int slot = j < K ? 0 : j - K + 1, el = j < K ? j : 0;
  float* pp = a.ptrs[slot] + el;
  float* pm = a.ptrs[5 + slot] + el;
  float* pv = a.ptrs[10 + slot] + el;
  float m = *pm, v = *pv;
  m = m + (gsum - m) * a.one_minus_beta1;                    // exp_avg.lerp_(grad, 1 - beta1)
  v = v * a.beta2 + (1.f - a.beta2) * (gsum * gsum);         // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
  *pm = m;
  *pv = v;
  const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
  *pp = *pp - a.step_size * (m / denom);                     // param.addcdiv_(exp_avg, denom, value = -lr / bias_correction1)
}

extern "C" size_t capamd_drmmtks_train_step_workspace_floats(int B, int Q, int topk) {
  return B > 0 && Q > 0 && topk > 0 ? (size_t)2 * B * Q * topk + (size_t)B * (topk + 4) : 0;
}

extern "C" int capamd_drmmtks_train_step(const int64_t* q_ids, const int64_t* pos_ids, const int64_t* neg_ids, const float* idf, int B, int Q, int L,
                                         const float* packed, int64_t V, int D, int topk, float* const* ptrs, int loss_type, float step_size,
                                         float one_minus_beta1, float beta2, float eps, float bc2_sqrt, float* loss_out, float* workspace,
                                         size_t workspace_floats, int* status, void* stream) {
  if (!q_ids || !pos_ids || !neg_ids || !idf || !packed || !ptrs || !loss_out || !workspace || !status) return CAPAMD_ERR_ARG;
  if (B < 1 || B > kMaxStepBatch || Q < 1 || Q > kMaxQ || L < 1 || L > 32768 || V < 1 || V > 0x7fffffffLL) return CAPAMD_ERR_ARG;
  if (topk < 1 || topk > kMaxTopK || topk > L || capamd_packed_row_stride(D) < 0 || loss_type < 0 || loss_type > 1 || !(bc2_sqrt > 0.f)) return CAPAMD_ERR_ARG;
  if (workspace_floats < capamd_drmmtks_train_step_workspace_floats(B, Q, topk)) return CAPAMD_ERR_WORKSPACE;
  float* feat = workspace;
  float* grads = workspace + (size_t)2 * B * Q * topk;
  const IdSource ids{q_ids, pos_ids, nullptr, nullptr, nullptr, nullptr};
  TksArgs fa{ids, nullptr, 2 * B, Q, L, packed, V, topk, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, status, feat, neg_ids, B};
  const int rc = drmmtks_features_launch(fa, D, stream);
  if (rc != CAPAMD_OK) return rc;
  TksStepArgs a{feat, q_ids, idf, B, Q, topk, ptrs, loss_type, step_size, one_minus_beta1, beta2, eps, bc2_sqrt, loss_out, grads};
  hipLaunchKernelGGL(drmmtks_step_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? CAPAMD_OK : CAPAMD_ERR_LAUNCH;
}
