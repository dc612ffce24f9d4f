// Fused KNRM forward for gfx950: gather -> cosine/exact-match interaction -> RBF kernel bank ->
// sum over document -> masked log -> sum over query -> combine MLP, one workgroup per
// (query, document) pair, nothing but the score written back.
//
// Reference semantics: KNRM_class.forward (capreolus/reranker/KNRM.py:39-55) on top of
// SimilarityMatrix (common.py:143-182) and RbfKernelBank (common.py:224-250).
//
// Layout of the work
//   workgroup  = 256 threads = 4 waves = 16 groups of 16 lanes; one pair per workgroup.
//   phase 1    = the 800 document ids are read once (coalesced int64); the DISTINCT real terms (id > 0)
//                are listed in LDS in order of first occurrence, each with its multiplicity (a document
//                repeats its frequent terms: 59 % distinct at 300 Zipf-distributed terms; a repeated
//                term's row is gathered once, its kernel values multiplied by the count -
//                interaction.h: distinct_terms); pads (id == 0) and OOV terms (id < 0)
//                are only counted: their similarity is exactly 0 (or exactly 1 for an OOV exact
//                match, common.py:155-158) so their kernel-pooling contribution is added in
//                closed form  n0[q]*K_k(0) + n1[q]*K_k(1)   (KNRM.py:50 sums over ALL positions).
//   phase 2    = group g walks compacted terms g, g+16, ...; each lane fetches its 16-byte
//                pieces of the packed row (5 x float4 for D=300), 4 query rows stay in registers;
//                16-lane DPP all-reduce gives the 4 dot products; lane l then owns query term
//                l&3 and kernels (l>>2), (l>>2)+4, (l>>2)+8 -> 3 exp per lane per term.
//   phase 3    = cross-group reduction through LDS in fixed order, log/mask/sum, combine.
// HBM/L2 traffic per pair: L*8 B of ids + one packed row per real term; the [B,Q,L] similarity
// and [B,K,Q,L] kernel tensors of the reference are never materialised.
#include "capreolus_amd.h"
#include "interaction.h"
#include "interaction_stream.h"
#include <stdlib.h>

using namespace capamd;

#ifndef CAPAMD_KNRM_ABLATE
#define CAPAMD_KNRM_ABLATE 0   // profiling builds only: 1 = no reduction / log / combine tail, 2 = no gather (phase 1 + tail only)
#endif

namespace {

constexpr int kMaxK = 12;      // 3 kernel slots x 4 lane-rows
constexpr int kMaxHidden = 64;
constexpr float kLog2e = 1.4426950408889634f;

struct KnrmArgs {
  IdSource ids;
  int B, Q, L;
  const float* packed;
  int64_t V;
  const float* mu;
  const float* sigma;
  int K;
  const float* w1;
  const float* b1;
  int hidden;
  const float* w2;
  const float* b2;
  int scoretanh;
  float* out;       // [B] scores (may be NULL for the feature/gradient call)
  int* status;
  float* feat;      // GRAD: [B, K] kernel-pooling features f_k (the input of `combine`)
  float* dfdmu;     // GRAD: [B, K] d f_k / d mu_k      (may be NULL)
  float* dfdsigma;  // GRAD: [B, K] d f_k / d sigma_k   (may be NULL)
  // GRAD, training step: the kernel parameters as the reference keeps them - one scalar tensor each (ptrs[k] = mu_k, ptrs[K + k] = sigma_k)
  // - and a second block of documents: pairs split .. B - 1 take query row (b - split) and document row (b - split) of d64_b
  float* const* kptrs;
  const int64_t* d64_b;
  int split;
};

// The per-pair tail both scoring kernels share (knrm_forward_kernel for Q <= kQT, the streaming kernel's list wave): ONE wave, lane =
// (query term q = lane & 3, kernel kk = lane >> 2); s / rs = the lane's kernel sum and its term's similarity sum over the document's real
// terms -> closed-form pad / OOV terms, masked log (KNRM.py:50-53), the sum over the query terms, `combine` (KNRM.py:27-34, :54).
// One piece of code, so that a pair's score does not depend on which of the two kernels a launch's size selected (the trainer's
// predictions are rounded to fp16: one ulp of fp32 can move a rank).
// C: LDS [64] mu[16] | sigma[16] | w1[16] (single Linear) | b1;  F: LDS [16] scratch of the hidden-layer variant.
__device__ __forceinline__ void knrm_pair_tail(const KnrmArgs& a, const float* C, float* F, int lane, float s, float rs, int n_one_q, int n_nonreal, int b) {
  const int q = lane & 3, kk = lane >> 2;
  float R = 0.f;
  if (kk < a.K) {
    const float mk = C[kk], sg = C[16 + kk];
    const float ck = (-0.5f * kLog2e) / (sg * sg);
    const int no = n_one_q;
    const int nz = n_nonreal - no;
    const float k0 = __builtin_amdgcn_exp2f(mk * mk * ck), k1 = __builtin_amdgcn_exp2f((1.f - mk) * (1.f - mk) * ck);
    s = __builtin_fmaf((float)nz, k0, s);
    s = __builtin_fmaf((float)no, k1, s);
    rs += (float)no;
    R = rs != 0.f ? logf(s + 1e-6f) : 0.f;            // KNRM.py:51-52
  }
  // f_k = sum over the query terms: the four lanes of a quad
  R += dpp_mov<0xB1>(R);
  R += dpp_mov<0x4E>(R);
  if (a.hidden > 0) {
    if (q == 0 && kk < kMaxK) F[kk] = R;
    wave_fence();
    float h = 0.f;
    if (lane < a.hidden) {
      h = a.b1[lane];
      for (int k = 0; k < a.K; ++k) h = __builtin_fmaf(a.w1[lane * a.K + k], F[k], h);
      h = a.w2[lane] * tanhf(h);
    }
    float sc = wave_allreduce_sum(h) + a.b2[0];
    if (a.scoretanh) sc = tanhf(sc);
    if (lane == 0) a.out[b] = sc;
    wave_fence();
  } else {
    const float v = (q == 0 && kk < a.K) ? C[32 + kk] * R : 0.f;
    float sc = wave_allreduce_sum(v) + C[48];
    if (a.scoretanh) sc = tanhf(sc);
    if (lane == 0) a.out[b] = sc;
  }
}

// GRAD additionally accumulates sum_j K (s - mu) and sum_j K (s - mu)^2, which give d f_k / d mu_k and d f_k / d sigma_k
// (RbfKernel parameters are trainable when `gradkernels`, reference common.py:229-230 / KNRM.py:22): the forward half
// of the training step; `combine` then runs under autograd on the [B, K] features.
template <int NV, int U, bool QLDS, int MINW, bool GRAD = false>
__global__ __launch_bounds__(kThreads, MINW) void knrm_forward_kernel(KnrmArgs a) {
  constexpr int PS = GRAD ? 12 : 4;  // floats per lane in the cross-group reduction buffer
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  // carve: tok[L] | partial[16][16][4] | S/aux
  int* tok = reinterpret_cast<int*>(smem_raw);
  const int tok_cap = (a.L + 3) & ~3;
  float* partial = reinterpret_cast<float*>(tok + tok_cap);      // 256 lanes x PS floats
  float* Rlds = partial + kGroupsPerWG * kGroup * PS;            // 48 (x3 with GRAD)
  float* Flds = Rlds + 48 * 3;                                   // kMaxK (+pad to 16) x3: f, df/dmu, df/dsigma
  float* Hlds = Flds + 48;                                       // kMaxHidden
  int* wave_cnt = reinterpret_cast<int*>(Hlds + kMaxHidden);     // [48]: distinct_terms' per-wave counts
  int* n_one = wave_cnt + 48;                                    // kQT per pass (+4 spare)
  float* Clds = reinterpret_cast<float*>(n_one + 8);             // [64]: mu[16] | sigma[16] | w1[16] (single Linear) | b1[0]: the tail's constants
  float4* qlds = reinterpret_cast<float4*>(Clds + 64);           // QLDS: [kQT][NV*16] float4
  int* mult = reinterpret_cast<int*>(qlds + kQT * kMaxNV * 16);  // [tok_cap] multiplicity of tok[k]
  int* hkey = mult + tok_cap;                                    // [kHashSlots] phase 1 only
  int* hfirst = hkey + kHashSlots;                               // [kHashSlots] phase 1 only

  const int tid = threadIdx.x;
  const int lane16 = tid & 15;
  const int g = tid >> 4;
  const int wave = tid >> 6;
  const int lane = tid & 63;
  const int b = blockIdx.x;
  PairIds ids = pair_ids(a.ids, GRAD && a.d64_b && b >= a.split ? b - a.split : b, a.Q, a.L);
  if (GRAD && a.d64_b && b >= a.split) ids.d64 = a.d64_b + (int64_t)(b - a.split) * a.L;

  // ---- phase 1: the document's distinct real terms with their multiplicities (interaction.h) ---------
  const TermList tl = distinct_terms(ids, a.L, a.V, a.status, tok, mult, hkey, hfirst, wave_cnt);
  const int n_real = CAPAMD_KNRM_ABLATE == 2 ? 0 : tl.n_unique;   // rows to gather (profiling builds: 2 = none)
  const int n_nonreal = a.L - tl.n_real;       // pads + OOV positions (closed form below)

  // per-lane kernel constants: lane owns query term (lane16 & 3), kernels krow + 4*s
  const int krow = lane16 >> 2;
  float mu_s[3], c_s[3];
  {
    float sg_l[3], mu_l[3];   // (clamped index, not a predicated load: the six loads go out together - one memory round trip, not three)
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int k = krow + 4 * s, kc = k < a.K ? k : a.K - 1;
      sg_l[s] = (GRAD && a.kptrs) ? *a.kptrs[a.K + kc] : a.sigma[kc];
      mu_l[s] = (GRAD && a.kptrs) ? *a.kptrs[kc] : a.mu[kc];
    }
    // ... and with them what the per-pair tail needs (kernel k's mu / sigma, the single Linear's weights): from LDS there instead of
    // from global memory - thread 0's 11-term combine was four to five dependent memory round trips at the very end of every pair
    if (tid < 16) {
      const int kc = tid < a.K ? tid : a.K - 1;
      // (the feature / gradient call has no combine layer: w1 and b1 are NULL there)
      const float m = (GRAD && a.kptrs) ? *a.kptrs[kc] : a.mu[kc], sg = (GRAD && a.kptrs) ? *a.kptrs[a.K + kc] : a.sigma[kc],
                  w = (a.out && a.hidden == 0) ? a.w1[kc] : 0.f, bb = a.out ? a.b1[0] : 0.f;
      Clds[tid] = m;
      Clds[16 + tid] = sg;
      Clds[32 + tid] = w;
      if (tid == 0) Clds[48] = bb;
    }
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int k = krow + 4 * s;
      mu_s[s] = k < a.K ? mu_l[s] : 0.f;
      c_s[s] = k < a.K ? (-0.5f * kLog2e) / (sg_l[s] * sg_l[s]) : 0.f;
    }
  }
  if (tid < 48) Flds[tid] = 0.f;

  for (int q0 = 0; q0 < a.Q; q0 += kQT) {
    QueryPass<NV> qp;
    if (QLDS)
      load_query_pass_lds<NV>(a.packed, ids, a.Q, q0, a.V, tid, kThreads, lane16, qlds, qp, a.status);
    else
      load_query_pass<NV>(a.packed, ids, a.Q, q0, a.V, lane16, qp, a.status);
    if (tid < kQT) n_one[tid] = 0;
    __syncthreads();
    // OOV exact matches (negative ids equal): rare, counted from the raw id row
    {
      bool any_oov_q = false;
#pragma unroll
      for (int t = 0; t < kQT; ++t) any_oov_q |= qp.id[t] < 0;
      if (any_oov_q) {
        for (int j = tid; j < a.L; j += kThreads) {
          const int64_t did = ids.d(j);
          if (did < 0) {
#pragma unroll
            for (int t = 0; t < kQT; ++t)
              if (qp.id[t] == (int)did && did > -2147483648LL) atomicAdd(&n_one[t], 1);
          }
        }
      }
    }

    float acc[3] = {0.f, 0.f, 0.f};
    float acc1[3] = {0.f, 0.f, 0.f}, acc2[3] = {0.f, 0.f, 0.f};  // GRAD: sum K*adj, sum K*adj^2
    float rowsum = 0.f;
    // (issuing a group's NEXT row before the current one is used - two rows in flight for one row's arithmetic registers - is slower
    // at every occupancy: 45.6 / 44.7 / 43.4 M pairs/s at 5 / 4 / 6 waves per SIMD against 49.2 M; the loop is not waiting for memory)
    for (int t0 = g; t0 < n_real; t0 += U * kGroupsPerWG) {
      RowRegs<NV> d[U];
      bool has[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int tu = t0 + u * kGroupsPerWG;
        has[u] = tu < n_real;
        load_row<NV>(a.packed, has[u] ? tok[tu] : 0, lane16, d[u]);
      }
      float x[U];
      int qoff = 0;
      if (QLDS) asm volatile("" : "+v"(qoff));  // opaque per iteration: keeps the LDS query reads inside the loop (no LICM into 80 VGPRs)
      rows_sim_my<NV, U, QLDS>(d, qp, qlds + qoff, lane16, x);
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (has[u]) {
          const float m = (float)mult[t0 + u * kGroupsPerWG];     // how often the document repeats this term
          rowsum = __builtin_fmaf(m, x[u], rowsum);
#pragma unroll
          for (int s = 0; s < 3; ++s) {
            const float adj = x[u] - mu_s[s];
            const float ev = __builtin_amdgcn_exp2f(adj * adj * c_s[s]);
            if (GRAD) {
              const float kv = m * ev;
              acc[s] += kv;
              acc1[s] = __builtin_fmaf(kv, adj, acc1[s]);
              acc2[s] = __builtin_fmaf(kv * adj, adj, acc2[s]);
            } else {
              acc[s] = __builtin_fmaf(m, ev, acc[s]);      // (spelled out: the streaming kernel's row() must round the same way)
            }
          }
        }
    }

    if (CAPAMD_KNRM_ABLATE == 1) {
      if (acc[0] + acc[1] + acc[2] + rowsum == 123.456f) a.out[b] = 1.f;
      continue;
    }
    // ---- phase 3: fixed-order cross-group reduction ---------------------------------------
    {
      float* pl = partial + (g * kGroup + lane16) * PS;
      *reinterpret_cast<float4*>(pl) = make_float4(acc[0], acc[1], acc[2], rowsum);
      if (GRAD) {
        *reinterpret_cast<float4*>(pl + 4) = make_float4(acc1[0], acc1[1], acc1[2], 0.f);
        *reinterpret_cast<float4*>(pl + 8) = make_float4(acc2[0], acc2[1], acc2[2], 0.f);
      }
    }
    __syncthreads();
    if (!GRAD && a.Q <= kQT) {
      // (the streaming kernel's order: a wave's four groups folded (g0 + g1) + (g2 + g3), the four waves' partials added in turn)
      if (tid < 64) {
        const int q = tid & 3, kk = tid >> 2;
        const int src_lane = (kk & 3) * 4 + q, slot = (kk >> 2) < 3 ? (kk >> 2) : 0;
        float s = 0.f, rs = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const float* p0 = partial + ((4 * w) * kGroup + src_lane) * PS + slot;
          const float* r0 = partial + ((4 * w) * kGroup + q) * PS + 3;
          s += (p0[0] + p0[kGroup * PS]) + (p0[2 * kGroup * PS] + p0[3 * kGroup * PS]);
          rs += (r0[0] + r0[kGroup * PS]) + (r0[2 * kGroup * PS] + r0[3 * kGroup * PS]);
        }
        knrm_pair_tail(a, Clds, Flds + 16, tid, s, rs, n_one[q], n_nonreal, b);
      }
      return;
    }
    if (tid < 48) {
      const int q = tid & 3, kk = tid >> 2;
      const int src_lane = (kk & 3) * 4 + q, slot = kk >> 2;
      float s = 0.f, rs = 0.f, s1 = 0.f, s2 = 0.f;
      for (int gg = 0; gg < kGroupsPerWG; ++gg) {
        s += partial[(gg * kGroup + src_lane) * PS + slot];
        rs += partial[(gg * kGroup + q) * PS + 3];
        if (GRAD) {
          s1 += partial[(gg * kGroup + src_lane) * PS + 4 + slot];
          s2 += partial[(gg * kGroup + src_lane) * PS + 8 + slot];
        }
      }
      float R = 0.f, Rmu = 0.f, Rsg = 0.f;
      if (kk < a.K) {
        const float mk = Clds[kk], sg = Clds[16 + kk];
        const float ck = (-0.5f * kLog2e) / (sg * sg);
        const int no = n_one[q];
        const int nz = n_nonreal - no;
        const float k0 = __builtin_amdgcn_exp2f(mk * mk * ck), k1 = __builtin_amdgcn_exp2f((1.f - mk) * (1.f - mk) * ck);
        s += (float)nz * k0;
        s += (float)no * k1;
        rs += (float)no;
        // KNRM.py:51-52: mask = (sum_j sim != 0); where(mask, log(result + 1e-6), 0)
        const bool on = rs != 0.f;
        R = on ? logf(s + 1e-6f) : 0.f;
        if (GRAD && on) {
          s1 += (float)nz * k0 * (-mk) + (float)no * k1 * (1.f - mk);
          s2 += (float)nz * k0 * mk * mk + (float)no * k1 * (1.f - mk) * (1.f - mk);
          const float inv = 1.f / (s + 1e-6f);
          Rmu = s1 / (sg * sg) * inv;        // d/dmu    exp(-(s-mu)^2 / (2 sigma^2)) = K (s-mu) / sigma^2
          Rsg = s2 / (sg * sg * sg) * inv;   // d/dsigma                              = K (s-mu)^2 / sigma^3
        }
      }
      Rlds[tid] = R;
      if (GRAD) {
        Rlds[48 + tid] = Rmu;
        Rlds[96 + tid] = Rsg;
      }
    }
    __syncthreads();
    if (tid < kMaxK) {
      Flds[tid] += ((Rlds[tid * 4 + 0] + Rlds[tid * 4 + 1]) + Rlds[tid * 4 + 2]) + Rlds[tid * 4 + 3];
      if (GRAD) {
        Flds[16 + tid] += ((Rlds[48 + tid * 4 + 0] + Rlds[48 + tid * 4 + 1]) + Rlds[48 + tid * 4 + 2]) + Rlds[48 + tid * 4 + 3];
        Flds[32 + tid] += ((Rlds[96 + tid * 4 + 0] + Rlds[96 + tid * 4 + 1]) + Rlds[96 + tid * 4 + 2]) + Rlds[96 + tid * 4 + 3];
      }
    }
    __syncthreads();
  }

  if (GRAD) {
    if (tid < a.K) {
      a.feat[(int64_t)b * a.K + tid] = Flds[tid];
      if (a.dfdmu) a.dfdmu[(int64_t)b * a.K + tid] = Flds[16 + tid];
      if (a.dfdsigma) a.dfdsigma[(int64_t)b * a.K + tid] = Flds[32 + tid];
    }
    if (!a.out) return;
  }

  // ---- combine (KNRM.py:27-34, :54) ----------------------------------------------------------
  if (a.hidden > 0) {
    if (tid < a.hidden) {
      float h = a.b1[tid];
      for (int k = 0; k < a.K; ++k) h = __builtin_fmaf(a.w1[tid * a.K + k], Flds[k], h);
      Hlds[tid] = tanhf(h);
    }
    __syncthreads();
  }
  if (tid == 0) {
    float sc;
    if (a.hidden > 0) {
      sc = a.b2[0];
      for (int j = 0; j < a.hidden; ++j) sc = __builtin_fmaf(a.w2[j], Hlds[j], sc);
    } else {
      sc = Clds[48];
      for (int k = 0; k < a.K; ++k) sc = __builtin_fmaf(Clds[32 + k], Flds[k], sc);
    }
    if (a.scoretanh) sc = tanhf(sc);
    a.out[b] = sc;
  }
}


// =====================================================================================================================================
// Streaming form (interaction_stream.h): persistent five-wave workgroups, the model's part as a policy.
//   gathering waves: kernel sums of a pair in registers, the wave's four groups folded ((g0 + g1) + (g2 + g3)) into partial[buf][wave][16 lanes]
//   list wave:       the four waves' partials + closed-form pad / OOV terms -> log -> sum over the query terms -> combine (KNRM.py:50-54)
struct KnrmStream {
  using Args = KnrmArgs;
  struct Gather {
    float mu_s[3], c_s[3];   // the lane's three kernels (krow + 4 s)
    float acc[3], rowsum;
  };
  // partial[2][4 waves][16 lanes][4] | Clds[64]: mu[16] | sigma[16] | w1[16] | b1 | Flds[16]
  __host__ __device__ static size_t lds_bytes(const Args&) { return (2 * 256 + 64 + 16) * sizeof(float); }
  __device__ static float* partial(char* lds, int buf) { return reinterpret_cast<float*>(lds) + buf * 256; }
  __device__ static float* consts(char* lds) { return reinterpret_cast<float*>(lds) + 512; }

  __device__ static void list_init(const Args& a, char* lds, int lane) {
    float* C = consts(lds);
    if (lane < 16) {
      const int kc = lane < a.K ? lane : a.K - 1;
      C[lane] = a.mu[kc];
      C[16 + lane] = a.sigma[kc];
      C[32 + lane] = a.hidden == 0 ? a.w1[kc] : 0.f;
      if (lane == 0) C[48] = a.b1[0];
    }
  }
  __device__ static void prepare(const Args&, char*, int, int) {}

  __device__ static void finish(const Args& a, const StreamSrc&, char* lds, int buf, const StreamMeta* meta, int lane) {
    const float* P = partial(lds, buf);
    const int q = lane & 3, kk = lane >> 2;             // lanes 0..47: (query term, kernel)
    const int src_lane = (kk & 3) * 4 + q, slot = (kk >> 2) < 3 ? (kk >> 2) : 0;
    float s = 0.f, rs = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      s += P[(w * kGroup + src_lane) * 4 + slot];
      rs += P[(w * kGroup + q) * 4 + 3];
    }
    knrm_pair_tail(a, consts(lds), consts(lds) + 64, lane, s, rs, meta->n_one[q], meta->n_nonreal, meta->pair);
  }

  __device__ static void gather_init(const Args& a, Gather& gs, int lane16) {
    const int krow = lane16 >> 2;
    float sg_l[3], mu_l[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int k = krow + 4 * s, kc = k < a.K ? k : a.K - 1;
      sg_l[s] = a.sigma[kc];
      mu_l[s] = a.mu[kc];
    }
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int k = krow + 4 * s;
      gs.mu_s[s] = k < a.K ? mu_l[s] : 0.f;
      gs.c_s[s] = k < a.K ? (-0.5f * kLog2e) / (sg_l[s] * sg_l[s]) : 0.f;
    }
  }
  template <int NV>
  __device__ static void pair_begin(const Args&, Gather& gs, QueryPass<NV>&, const StreamMeta*, int) {
    gs.acc[0] = gs.acc[1] = gs.acc[2] = 0.f;
    gs.rowsum = 0.f;
  }
  __device__ static void row(const Args&, Gather& gs, float x, unsigned entry, char*, int, int) {
    const float mlt = (float)(entry >> kIdBits);          // how often the document repeats this term
    gs.rowsum = __builtin_fmaf(mlt, x, gs.rowsum);
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const float adj = x - gs.mu_s[s];
      gs.acc[s] = __builtin_fmaf(mlt, __builtin_amdgcn_exp2f(adj * adj * gs.c_s[s]), gs.acc[s]);      // (as knrm_forward_kernel)
    }
  }
  __device__ static void pair_end(const Args&, Gather& gs, char* lds, int buf, int wave, int lane) {
    float v[4] = {gs.acc[0], gs.acc[1], gs.acc[2], gs.rowsum};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      v[c] += __shfl_xor(v[c], 16, 64);
      v[c] += __shfl_xor(v[c], 32, 64);
    }
    if (lane < 16) *reinterpret_cast<float4*>(partial(lds, buf) + (wave * kGroup + lane) * 4) = make_float4(v[0], v[1], v[2], v[3]);
  }
};

}  // namespace

namespace {

int knrm_launch(const IdSource& ids, int B, int Q, int L, const float* packed, int64_t V, int D, const float* mu,
                const float* sigma, int K, const float* w1, const float* b1, int hidden, const float* w2, const float* b2,
                int scoretanh, float* out, int* status, void* workspace, size_t workspace_bytes, unsigned flags, void* stream) {
  if (!packed || !mu || !sigma || !w1 || !b1 || !out || !status) return CAPAMD_ERR_ARG;
  if (B < 0 || Q < 1 || L < 1 || V < 1 || K < 1 || K > kMaxK || hidden < 0 || hidden > kMaxHidden) return CAPAMD_ERR_ARG;
  if (hidden > 0 && (!w2 || !b2)) return CAPAMD_ERR_ARG;
  if (capamd_packed_row_stride(D) < 0 || L > 32768 || V > 0x7fffffffLL) return CAPAMD_ERR_ARG;
  KnrmArgs a{ids, B, Q, L, packed, V, mu, sigma, K, w1, b1, hidden, w2, b2, scoretanh, out, status, nullptr, nullptr, nullptr};
  const size_t smem = (size_t)((L + 3) & ~3) * 8 + (1024 + 144 + 48 + kMaxHidden + 48 + 8 + 64) * 4 + dedup_hash_bytes(L) + (size_t)kQT * kMaxNV * 16 * 16;
  hipStream_t s = (hipStream_t)stream;
  (void)hipGetLastError();
  // Variant = how many rows each 16-lane group keeps in flight (U), where the query rows live (registers or a
  // shared LDS copy) and the occupancy target.  Measured on MI355X: U=1 + LDS query rows + 6
  // waves/SIMD (80 VGPRs) is fastest (36.7 M pairs/s vs 29.5 M for U=2/registers/3 waves); deeper unrolls at lower
  // occupancy and any variant that spills are slower.  CAPAMD_KNRM_VARIANT selects the others for profiling.
  static const int variant = [] {
    const char* e = getenv("CAPAMD_KNRM_VARIANT");
    return e ? atoi(e) : 0;
  }();
  // Launches that outnumber the workgroups the chip holds run the streaming kernel (persistent workgroups, a list wave beside four
  // gathering waves): it needs the caller's workspace word for its ticket counter, <= kQT query terms, L <= kDedupMaxL and ids < 2^22.
  static const int stream_mode = [] {
    const char* e = getenv("CAPAMD_KNRM_STREAM");   // profiling: 0 = never, 2 = whenever the geometry allows
    return e ? atoi(e) : 1;
  }();
  // ... and a table the cache hierarchy can help with: over a table several times the 256 MB Infinity Cache (measured: 5.1 GB, uniform
  // ids) every row comes from HBM, the one-pair-per-workgroup kernel already sits at the read ceiling (0.81-0.82 of 8 TB/s) and the
  // streaming kernel's barrier per pair costs 2-4 % there (0.78-0.805; profiles/r03/stream_ab.txt)
  const bool cacheable = (int64_t)V * row_stride_for_dim(D) * 4 <= (1LL << 30);
  if (stream_mode && !variant && ((B > 3072 && cacheable) || stream_mode == 2)) {
    const StreamSrc src{ids, B, Q, L, packed, V, status};
    int rc = CAPAMD_OK;
    if (stream_launch<KnrmStream>(src, a, D, workspace, workspace_bytes, s, &rc)) return rc;
  }
#define LAUNCH(NV_, U_, QL_, W_)                                                     \
  do {                                                                               \
    auto kern = knrm_forward_kernel<NV_, U_, QL_, W_>;                               \
    if (const int bad = lds_budget(kern, smem)) return bad;                          \
    hipLaunchKernelGGL(kern, dim3(B), dim3(kThreads), smem, s, a);                   \
  } while (0)
  // One candidate list per launch (B <= the 1536 workgroups the chip holds at once): the launch is as long as its longest
  // document, so four rows in flight per 16-lane group (U = 4) beat occupancy - 62 vs 77 us at B = 1000; from B = 2000 on
  // the 6-waves-per-SIMD variant wins again (96 vs 100 us; 33.0 vs 25.6 M pairs/s at B = 16000).  Same summation order.
  // ... unless the caller keeps several launches in flight on different streams (capamd_set_concurrent_launches): the small
  // launches then share the chip and occupancy wins again - 64 lists of 1000 pairs over 4 streams: 34.3 M pairs/s with the
  // 6-wave variant, 26.2 M with U = 4 (and 16.9 M strictly serial).
  const int chosen = variant ? variant : ((B <= 1536 && !(flags & CAPAMD_LAUNCH_CONCURRENT)) ? 1 : 0);
#define LAUNCH_V(NV_)                          \
  switch (chosen) {                            \
    case 1: LAUNCH(NV_, 4, false, 2); break;   \
    case 6: LAUNCH(NV_, 6, true, 2); break;    \
    case 4: LAUNCH(NV_, 1, true, 8); break;    \
    case 14: LAUNCH(NV_, 1, true, 5); break;   \
    case 15: LAUNCH(NV_, 2, false, 3); break;  \
    default: LAUNCH(NV_, 1, true, 6); break;   \
  }
  switch (nv_for_dim(D)) {
    case 1: LAUNCH(1, 2, false, 3); break;
    case 2: LAUNCH(2, 2, false, 3); break;
    case 3: LAUNCH(3, 2, false, 3); break;
    case 4: LAUNCH(4, 2, false, 3); break;
    default: LAUNCH_V(5); break;
  }
#undef LAUNCH_V
#undef LAUNCH
  return hipGetLastError() == hipSuccess ? CAPAMD_OK : CAPAMD_ERR_LAUNCH;
}

}  // namespace

extern "C" int capamd_knrm_forward(const int64_t* q_ids, const int64_t* d_ids, int B, int Q, int L, const float* packed,
                                   int64_t V, int D, const float* mu, const float* sigma, int K, const float* w1,
                                   const float* b1, int hidden, const float* w2, const float* b2, int scoretanh,
                                   float* out, int* status, void* workspace, size_t workspace_bytes, unsigned flags, void* stream) {
  if (B == 0) return CAPAMD_OK;
  if (!q_ids || !d_ids) return CAPAMD_ERR_ARG;
  const IdSource ids{q_ids, d_ids, nullptr, nullptr, nullptr, nullptr};
  return knrm_launch(ids, B, Q, L, packed, V, D, mu, sigma, K, w1, b1, hidden, w2, b2, scoretanh, out, status, workspace, workspace_bytes, flags,
                     stream);
}

static int knrm_features_launch(KnrmArgs a, int D, void* stream) {
  const size_t smem = (size_t)((a.L + 3) & ~3) * 8 + (3072 + 144 + 48 + kMaxHidden + 48 + 8 + 64) * 4 + dedup_hash_bytes(a.L) + (size_t)kQT * kMaxNV * 16 * 16;
  hipStream_t s = (hipStream_t)stream;
  (void)hipGetLastError();
#define LAUNCH(NV_)                                                                  \
  do {                                                                               \
    auto kern = knrm_forward_kernel<NV_, 1, true, 4, true>;                          \
    if (const int bad = lds_budget(kern, smem)) return bad;                          \
    hipLaunchKernelGGL(kern, dim3(a.B), dim3(kThreads), smem, s, a);                 \
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

extern "C" int capamd_knrm_features(const int64_t* q_ids, const int64_t* d_ids, int B, int Q, int L, const float* packed, int64_t V,
                                    int D, const float* mu, const float* sigma, int K, float* feat_out, float* dfdmu_out,
                                    float* dfdsigma_out, int* status, void* stream) {
  if (B == 0) return CAPAMD_OK;
  if (!q_ids || !d_ids || !packed || !mu || !sigma || !feat_out || !status) return CAPAMD_ERR_ARG;
  if (B < 0 || Q < 1 || L < 1 || L > 32768 || V < 1 || V > 0x7fffffffLL || K < 1 || K > kMaxK) return CAPAMD_ERR_ARG;
  if (capamd_packed_row_stride(D) < 0) return CAPAMD_ERR_ARG;
  const IdSource ids{q_ids, d_ids, nullptr, nullptr, nullptr, nullptr};
  KnrmArgs a{ids, B, Q, L, packed, V, mu, sigma, K, nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr, status, feat_out, dfdmu_out,
             dfdsigma_out};
  return knrm_features_launch(a, D, stream);
}

// ---- one KNRM training step without a host round trip (reference trainer/pytorch.py:93-108) -----------------------
// score() on the positive and the negative documents -> pairwise hinge (reranker/common.py:101-103) or softmax (:96-98) loss -> backward
// through `combine` (a single Linear, KNRM.py:27-34, optionally under tanh) and the RBF kernels' mu / sigma -> torch.optim.Adam's update
// (no weight decay, no amsgrad; bias corrections computed by the caller in double, as the plain Adam of the reference does), all on the
// device, in TWO launches: the feature kernel above over the 2 B documents (values + d f / d mu, d f / d sigma per document; it reads
// the 2 K scalar kernel parameters through the pointer table) and ONE workgroup for everything that is per batch.  Parameters and Adam moments are updated IN PLACE through a table of device pointers
// (the reference's state_dict names one scalar nn.Parameter per kernel and quantity, common.py:229-230):
//   ptrs[0 .. P)  the parameters, [P .. 2 P) their exp_avg, [2 P .. 3 P) their exp_avg_sq;  P = 2 K + 2:  mu_0 .. mu_{K-1}, sigma_0 .. sigma_{K-1},
//   the Linear's weight [K], its bias [1]
constexpr int kMaxStepBatch = 1024;

struct KnrmStepArgs {
  const float* f[2];      // [B, K] features of the positive / negative documents
  const float* dmu[2];    // [B, K] d f_k / d mu_k
  const float* dsg[2];    // [B, K] d f_k / d sigma_k
  int B, K;
  float* const* ptrs;
  int train_kernels, scoretanh, loss_type;
  float step_size, one_minus_beta1, beta2, eps, bc2_sqrt;
  float* loss_out;
};

__global__ __launch_bounds__(256) void knrm_step_kernel(KnrmStepArgs a) {
  __shared__ float W[kMaxK + 4], gsc[2][kMaxStepBatch], lsum[kMaxStepBatch];
  const int tid = threadIdx.x, K = a.K, P = 2 * K + 2;
  if (tid < K) W[tid] = a.ptrs[2 * K][tid];
  if (tid == K) W[K] = a.ptrs[2 * K + 1][0];
  __syncthreads();
  const float inv_b = 1.f / (float)a.B;
  for (int i = tid; i < a.B; i += 256) {
    float sc[2], dt[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float v = W[K];
      for (int k = 0; k < K; ++k) v = __builtin_fmaf(W[k], a.f[h][(int64_t)i * K + k], v);     // (the scoring kernels' order)
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
  // one thread per parameter element (and one for the loss): its gradient summed over the batch in pair order, then Adam's update
  const int j = tid;
  if (j > 3 * K + 1) return;
  if (j == 3 * K + 1) {
    float l = 0.f;
    for (int i = 0; i < a.B; ++i) l += lsum[i];
    a.loss_out[0] = l * inv_b;
    return;
  }
  if (j < 2 * K && !a.train_kernels) return;
  float g = 0.f;
  if (j < 2 * K) {
    const int k = j < K ? j : j - K;
    const float* const* d = j < K ? a.dmu : a.dsg;
    for (int i = 0; i < a.B; ++i) g += gsc[0][i] * d[0][(int64_t)i * K + k] + gsc[1][i] * d[1][(int64_t)i * K + k];
    g *= W[k];
  } else if (j < 3 * K) {
    const int k = j - 2 * K;
    for (int i = 0; i < a.B; ++i) g += gsc[0][i] * a.f[0][(int64_t)i * K + k] + gsc[1][i] * a.f[1][(int64_t)i * K + k];
  } else {
    for (int i = 0; i < a.B; ++i) g += gsc[0][i] + gsc[1][i];
  }
  const int slot = j < 2 * K ? j : (j < 3 * K ? 2 * K : 2 * K + 1), el = (j >= 2 * K && j < 3 * K) ? j - 2 * K : 0;
  float* pp = a.ptrs[slot] + el;
  float* pm = a.ptrs[P + slot] + el;
  float* pv = a.ptrs[2 * P + slot] + el;
  float m = *pm, v = *pv;
  m = m + (g - m) * a.one_minus_beta1;                       // exp_avg.lerp_(grad, 1 - beta1)
  v = v * a.beta2 + (1.f - a.beta2) * (g * g);               // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
  *pm = m;
  *pv = v;
  const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
  *pp = *pp - a.step_size * (m / denom);                     // param.addcdiv_(exp_avg, denom, value = -lr / bias_correction1)
}

extern "C" size_t capamd_knrm_train_step_workspace_floats(int B, int K) { return B > 0 && K > 0 ? (size_t)6 * B * K : 0; }

extern "C" int capamd_knrm_train_step(const int64_t* q_ids, const int64_t* pos_ids, const int64_t* neg_ids, int B, int Q, int L, const float* packed,
                                      int64_t V, int D, int K, float* const* ptrs, int train_kernels, int scoretanh, int loss_type, float step_size,
                                      float one_minus_beta1, float beta2, float eps, float bc2_sqrt, float* loss_out, float* workspace,
                                      size_t workspace_floats, int* status, void* stream) {
  if (!q_ids || !pos_ids || !neg_ids || !packed || !ptrs || !loss_out || !workspace || !status) return CAPAMD_ERR_ARG;
  if (B < 1 || B > kMaxStepBatch || Q < 1 || L < 1 || L > 32768 || V < 1 || V > 0x7fffffffLL || K < 1 || K > kMaxK) return CAPAMD_ERR_ARG;
  if (loss_type < 0 || loss_type > 1 || !(bc2_sqrt > 0.f) || capamd_packed_row_stride(D) < 0) return CAPAMD_ERR_ARG;
  if (workspace_floats < capamd_knrm_train_step_workspace_floats(B, K)) return CAPAMD_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const size_t n = (size_t)B * K;
  float *feat = workspace, *dmu = workspace + 2 * n, *dsg = workspace + 4 * n;       // [2 B, K] each: the positive documents, then the negative ones
  const IdSource ids{q_ids, pos_ids, nullptr, nullptr, nullptr, nullptr};
  KnrmArgs fa{ids, 2 * B, Q, L, packed, V, nullptr, nullptr, K, nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr, status, feat, dmu, dsg, ptrs, neg_ids, B};
  const int rc = knrm_features_launch(fa, D, stream);
  if (rc != CAPAMD_OK) return rc;
  KnrmStepArgs a{{feat, feat + n}, {dmu, dmu + n}, {dsg, dsg + n}, B, K, ptrs, train_kernels, scoretanh, loss_type, step_size,
                 one_minus_beta1, beta2, eps, bc2_sqrt, loss_out};
  hipLaunchKernelGGL(knrm_step_kernel, dim3(1), dim3(256), 0, s, a);
  return hipGetLastError() == hipSuccess ? CAPAMD_OK : CAPAMD_ERR_LAUNCH;
}

extern "C" int capamd_knrm_forward_indexed(const int32_t* q_table, const int32_t* d_table, const int32_t* pair_q,
                                           const int32_t* pair_d, int B, int Q, int L, const float* packed, int64_t V, int D,
                                           const float* mu, const float* sigma, int K, const float* w1, const float* b1,
                                           int hidden, const float* w2, const float* b2, int scoretanh, float* out,
                                           int* status, void* workspace, size_t workspace_bytes, unsigned flags, void* stream) {
  if (B == 0) return CAPAMD_OK;
  if (!q_table || !d_table || !pair_q || !pair_d) return CAPAMD_ERR_ARG;
  const IdSource ids{nullptr, nullptr, q_table, d_table, pair_q, pair_d};
  return knrm_launch(ids, B, Q, L, packed, V, D, mu, sigma, K, w1, b1, hidden, w2, b2, scoretanh, out, status, workspace, workspace_bytes, flags,
                     stream);
}
