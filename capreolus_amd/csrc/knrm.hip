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
//                interaction.cuh: distinct_terms); pads (id == 0) and OOV terms (id < 0)
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
#include "interaction.cuh"
#include <stdlib.h>
#include <type_traits>

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
};

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
  const PairIds ids = pair_ids(a.ids, b, a.Q, a.L);

  // ---- phase 1: the document's distinct real terms with their multiplicities (interaction.cuh) ---------
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
      sg_l[s] = a.sigma[kc];
      mu_l[s] = a.mu[kc];
    }
    // ... and with them what the per-pair tail needs (kernel k's mu / sigma, the single Linear's weights): from LDS there instead of
    // from global memory - thread 0's 11-term combine was four to five dependent memory round trips at the very end of every pair
    if (tid < 16) {
      const int kc = tid < a.K ? tid : a.K - 1;
      // (the feature / gradient call has no combine layer: w1 and b1 are NULL there)
      const float m = a.mu[kc], sg = a.sigma[kc], w = (a.out && a.hidden == 0) ? a.w1[kc] : 0.f, bb = a.out ? a.b1[0] : 0.f;
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
            const float kv = m * __builtin_amdgcn_exp2f(adj * adj * c_s[s]);
            acc[s] += kv;
            if (GRAD) {
              acc1[s] = __builtin_fmaf(kv, adj, acc1[s]);
              acc2[s] = __builtin_fmaf(kv * adj, adj, acc2[s]);
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
// Streaming variant: persistent workgroups of FIVE waves - four gathering waves that never leave the gather loop, and one list wave
// that (a) turns the NEXT pair's id row into its distinct-term list and stages its query rows, and (b) finishes the PREVIOUS pair
// (cross-wave reduction, closed-form pad / OOV terms, log, combine).  In the one-pair-per-workgroup kernel above a workgroup requests
// no rows while it lists its terms or reduces its sums - 23 % of its life on the benchmark's lists, which is exactly the distance
// between its 12.3 TB/s of rows requested and the 15.5 TB/s a gather-only kernel gets from the same request stream.
//   * pairs are handed out by a ticket counter (the caller's 4-byte workspace word, zeroed by the launch code): documents differ 40x
//     in length, a static partition of 64,000 pairs over 1,536 workgroups would leave the tail of the launch to the unluckiest one;
//   * everything the two roles exchange is double-buffered in LDS and ONE s_barrier per pair separates the generations:
//       between barriers i-1 and i   gatherers: pair i from list[i&1], qrows[i&1] -> partial[i&1]
//                                    list wave: finish pair i-1 from partial[(i-1)&1]; build pair i+1 into list / qrows / meta[(i+1)&1]
//   * the list wave is a single wave, so the distinct-term pass needs no barrier at all (LDS operations of one wave complete in order);
//     the hash is ONE word per slot, (id << 10 | first position): a compare-and-swap claims an empty slot, an atomic minimum on a slot
//     that already belongs to the id keeps the first position - for equal ids the order of the words is the order of the positions.
//     That needs id < 2^22 (and L <= 896 as above); tables beyond 4.19 M rows take the one-pair-per-workgroup kernel;
//   * list entries are (id | multiplicity << 22): one LDS read per gathered row;
//   * the four query rows go from global memory straight into LDS (global_load_lds_dwordx4) while the list wave hashes: no registers
//     held across the pass, no round trip on anybody's critical path.
// The list comes out in the same order (first occurrence) as interaction.cuh: distinct_terms, the gather arithmetic is the same code.
constexpr int kStreamThreads = 320;
constexpr int kStreamBlocks = (kDedupMaxL + 63) / 64;   // position blocks of 64 the list wave walks
constexpr unsigned kIdBits = 22, kIdMask = (1u << kIdBits) - 1u;
constexpr unsigned kHashEmpty = 0xffffffffu;

struct StreamMeta {     // what the list wave tells the gatherers (and its later self) about a pair
  int pair;             // -1: no more pairs
  int n_unique;
  int n_nonreal;        // L - real positions: pads + OOV terms (closed form)
  int pad;
  int n_one[kQT];       // OOV exact matches per query term
  int qid[kQT];
  float qden[kQT];
};

__device__ __forceinline__ void wave_fence() {   // orders the list wave's LDS traffic across lanes (one wave: no instruction needed)
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// 16 bytes per lane from global memory straight into LDS: 64 lanes -> 1 KiB at the wave-uniform LDS byte address `lds_addr`.
__device__ __forceinline__ void stream_dma16(const void* base, uint32_t voff, uint32_t lds_addr) {
  uint32_t saved;
  lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(saved)
               : "s"(lds_addr), "v"(voff), "s"(base)
               : "memory");
}

// The list wave's half of a pair, part 1: list[] / qrows[] / meta of pair b.
template <int NV, bool ID32>
__device__ __forceinline__ void stream_build(const KnrmArgs& a, int b, unsigned* list, float4* qrows, StreamMeta* meta, unsigned* hash,
                                             int lane) {
  const PairIds ids = pair_ids(a.ids, b, a.Q, a.L);
  const int L = a.L;
  asm volatile("" : "+v"(lane));   // opaque per call: nothing lane-derived is hoisted out of the pair loop (and then spilled)
  // hash slots empty (16 per lane)
#pragma unroll
  for (int i = 0; i < kHashSlots / 256; ++i)
    reinterpret_cast<uint4*>(hash)[i * 64 + lane] = make_uint4(kHashEmpty, kHashEmpty, kHashEmpty, kHashEmpty);
  // the document's id row: position r * 64 + lane (all requested together; one branch on the layout around the loads)
  // (unconditional loads at clamped indices: a load under a branch is waited for at the join - fourteen round trips in series)
  typename std::conditional<ID32, int, int64_t>::type dv[kStreamBlocks];
#pragma unroll
  for (int r = 0; r < kStreamBlocks; ++r) {
    const unsigned j = min((unsigned)(r * 64 + lane), (unsigned)(L - 1));   // (unsigned: scalar base + 32-bit lane offset, one register per address)
    if (ID32) dv[r] = ids.d32[j];
    else dv[r] = ids.d64[j];
  }
  // the query's ids (wave-uniform addresses) and rows: straight into LDS, under the hash work below
  int64_t qid64[kQT];
#pragma unroll
  for (int t = 0; t < kQT; ++t) {
    if (ID32) qid64[t] = ids.q32[t < a.Q ? t : a.Q - 1];
    else qid64[t] = ids.q64[t < a.Q ? t : a.Q - 1];
  }
  bool bad_q = false;
  int qid[kQT];
#pragma unroll
  for (int t = 0; t < kQT; ++t) {
    if (t >= a.Q) qid64[t] = 0;
    if (qid64[t] >= a.V) { bad_q = true; qid64[t] = 0; }
    qid[t] = (int)qid64[t];
    const float* row = a.packed + (qid64[t] > 0 ? qid64[t] : 0) * (int64_t)(64 * NV);
    const uint32_t dst = (uint32_t)(size_t)(qrows + t * NV * 16);
#pragma unroll
    for (int c0 = 0; c0 < NV * 16; c0 += 64)
      if (c0 + lane < NV * 16) stream_dma16(row, (uint32_t)(c0 + lane) * 16u, dst + c0 * 16);
  }
  if (bad_q && lane == 0) atomicOr(a.status, kErrQueryIdRange);
  const bool any_oov_q = (qid[0] | qid[1] | qid[2] | qid[3]) < 0;
  wave_fence();
  // A: every real position claims / joins its term's slot and leaves the minimum of (id << 10 | position) there
  int slot[kStreamBlocks];
  int n_real = 0;
  int n_one[kQT] = {0, 0, 0, 0};
  bool bad_d = false;
#pragma unroll
  for (int r = 0; r < kStreamBlocks; ++r) {
    slot[r] = -1;
    const int j = r * 64 + lane;
    int id;
    if (ID32) {
      id = dv[r];
    } else {
      const int64_t d = dv[r];
      id = d >= a.V ? 0x7fffffff : d < 0 ? (d > -2147483648LL ? (int)d : (int)0x80000000) : (int)d;
    }
    if (id >= a.V) { bad_d = true; id = 0; }
    if (j >= L) id = 0;
    if (any_oov_q) {   // OOV exact matches (negative query id == negative document id): rare
#pragma unroll
      for (int t = 0; t < kQT; ++t) n_one[t] += __popcll(__ballot(id < 0 && id != (int)0x80000000 && id == qid[t]));
    }
    const bool real = id > 0;
    n_real += __popcll(__ballot(real));
    if (real) {
      const unsigned word = ((unsigned)id << 10) | (unsigned)j;
      unsigned h = ((unsigned)id * 2654435761u) >> 22;
      for (;;) {
        const unsigned old = atomicCAS(&hash[h], kHashEmpty, word);
        if (old == kHashEmpty) break;
        if ((old >> 10) == (unsigned)id) { atomicMin(&hash[h], word); break; }
        h = (h + 1) & (kHashSlots - 1);
      }
      slot[r] = (int)h;
    }
  }
  if (bad_d) atomicOr(a.status, kErrDocIdRange);
  wave_fence();
  // B: the position a slot's word names owns the term ...
  unsigned own_bits = 0;
#pragma unroll
  for (int r = 0; r < kStreamBlocks; ++r)
    if (slot[r] >= 0 && (hash[slot[r]] & 1023u) == (unsigned)(r * 64 + lane)) own_bits |= 1u << r;
  wave_fence();   // every comparison is done before any slot is overwritten
  // ... owners in document order get the dense index; the slot keeps it for the counting pass
  int n_unique = 0;
#pragma unroll
  for (int r = 0; r < kStreamBlocks; ++r) {
    const bool own = (own_bits >> r) & 1u;
    const unsigned long long m = __ballot(own);
    if (own) {
      const int k = n_unique + __popcll(m & ((1ull << lane) - 1ull));
      list[k] = hash[slot[r]] >> 10;         // the id, multiplicity 0 so far
      hash[slot[r]] = (unsigned)k;
    }
    n_unique += __popcll(m);
  }
  wave_fence();
  // C: occurrences per term
#pragma unroll
  for (int r = 0; r < kStreamBlocks; ++r)
    if (slot[r] >= 0) atomicAdd(&list[hash[slot[r]]], 1u << kIdBits);
  // the query rows have landed: norm out of the last float of each row (and a 0 there for the dot products)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  wave_fence();
  if (lane < kQT) {
    float* den_slot = reinterpret_cast<float*>(qrows + lane * NV * 16) + (64 * NV - 1);
    const float den = *den_slot;
    *den_slot = 0.f;
    meta->qden[lane] = den;
    meta->qid[lane] = lane == 0 ? qid[0] : lane == 1 ? qid[1] : lane == 2 ? qid[2] : qid[3];
    meta->n_one[lane] = lane == 0 ? n_one[0] : lane == 1 ? n_one[1] : lane == 2 ? n_one[2] : n_one[3];
  }
  if (lane == 0) {
    meta->pair = b;
    meta->n_unique = n_unique;
    meta->n_nonreal = L - n_real;
  }
}

// The list wave's half of a pair, part 2: from the four gathering waves' partial sums to the score (KNRM.py:50-54).
// partial: [4 waves][16 lanes] float4 = (kernel slots 0..2 of the lane's (query term, kernel row), similarity row sum)
__device__ __forceinline__ void stream_finish(const KnrmArgs& a, const float* partial, const StreamMeta* meta, const float* Clds, float* Flds,
                                              int lane) {
  const int q = lane & 3, kk = lane >> 2;             // lanes 0..47: (query term, kernel)
  const int src_lane = (kk & 3) * 4 + q, slot = kk >> 2;
  float R = 0.f;
  if (kk < a.K) {
    float s = 0.f, rs = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      s += partial[(w * kGroup + src_lane) * 4 + slot];
      rs += partial[(w * kGroup + q) * 4 + 3];
    }
    const float mk = Clds[kk], sg = Clds[16 + kk];
    const float ck = (-0.5f * kLog2e) / (sg * sg);
    const int no = meta->n_one[q];
    const int nz = meta->n_nonreal - no;
    const float k0 = __builtin_amdgcn_exp2f(mk * mk * ck), k1 = __builtin_amdgcn_exp2f((1.f - mk) * (1.f - mk) * ck);
    s += (float)nz * k0;
    s += (float)no * k1;
    rs += (float)no;
    R = rs != 0.f ? logf(s + 1e-6f) : 0.f;            // KNRM.py:51-52
  }
  // f_k = sum over the query terms: the four lanes of a quad
  R += dpp_mov<0xB1>(R);
  R += dpp_mov<0x4E>(R);
  const int b = meta->pair;
  if (a.hidden > 0) {
    if (q == 0 && kk < kMaxK) Flds[kk] = R;
    wave_fence();
    float h = 0.f;
    if (lane < a.hidden) {
      h = a.b1[lane];
      for (int k = 0; k < a.K; ++k) h = __builtin_fmaf(a.w1[lane * a.K + k], Flds[k], h);
      h = a.w2[lane] * tanhf(h);
    }
    float sc = wave_allreduce_sum(h) + a.b2[0];
    if (a.scoretanh) sc = tanhf(sc);
    if (lane == 0) a.out[b] = sc;
    wave_fence();
  } else {
    const float v = (q == 0 && kk < a.K) ? Clds[32 + kk] * R : 0.f;
    float sc = wave_allreduce_sum(v) + Clds[48];
    if (a.scoretanh) sc = tanhf(sc);
    if (lane == 0) a.out[b] = sc;
  }
}

template <int NV, bool ID32>
__global__ __launch_bounds__(kStreamThreads, 8) void knrm_stream_kernel(KnrmArgs a, int* ticket) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int cap = (a.L + 3) & ~3;
  // carve: qrows[2] | list[2] | partial[2] | hash | meta[2] | constants | F
  float4* qrows = reinterpret_cast<float4*>(smem_raw);                    // [2][kQT * NV * 16]
  unsigned* list = reinterpret_cast<unsigned*>(qrows + 2 * kQT * NV * 16);  // [2][cap]
  float* partial = reinterpret_cast<float*>(list + 2 * cap);              // [2][4 waves][16 lanes][4]
  unsigned* hash = reinterpret_cast<unsigned*>(partial + 2 * 256);        // [kHashSlots]
  StreamMeta* meta = reinterpret_cast<StreamMeta*>(hash + kHashSlots);    // [2]
  float* Clds = reinterpret_cast<float*>(meta + 2);                       // [64]: mu[16] | sigma[16] | w1[16] | b1
  float* Flds = Clds + 64;                                                // [16]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  if (wave == 4) {
    // ------------------------------------------------------------------ the list wave --------------------------------------
    if (lane < 16) {
      const int kc = lane < a.K ? lane : a.K - 1;
      Clds[lane] = a.mu[kc];
      Clds[16 + lane] = a.sigma[kc];
      Clds[32 + lane] = a.hidden == 0 ? a.w1[kc] : 0.f;
      if (lane == 0) Clds[48] = a.b1[0];
    }
    int t_next;
    {
      int t = 0;
      if (lane == 0) t = atomicAdd(ticket, 2);          // two tickets: this pair and the next (one atomic round trip ahead from here on)
      t = __builtin_amdgcn_readfirstlane(t);
      t_next = t + 1;
      if (t < a.B) stream_build<NV, ID32>(a, t, list, qrows, meta, hash, lane);
      else if (lane == 0) meta[0].pair = -1;
    }
    __syncthreads();
    for (int i = 0;; ++i) {
      const int cur = i & 1, nxt = cur ^ 1;
      if (i > 0) stream_finish(a, partial + nxt * 256, meta + nxt, Clds, Flds, lane);   // pair i-1
      if (meta[cur].pair < 0) break;
      // ticket for the build after this one is asked for now, used next iteration
      int t_after = 0;
      if (lane == 0) t_after = atomicAdd(ticket, 1);
      const int t = t_next;
      if (t < a.B) stream_build<NV, ID32>(a, t, list + nxt * cap, qrows + nxt * kQT * NV * 16, meta + nxt, hash, lane);
      else if (lane == 0) meta[nxt].pair = -1;
      t_next = __builtin_amdgcn_readfirstlane(t_after);
      __syncthreads();
    }
    return;
  }

  // -------------------------------------------------------------------- the gathering waves --------------------------------
  const int lane16 = tid & 15;
  const int g = tid >> 4;                      // 0..15
  const int krow = lane16 >> 2, myq = lane16 & 3;
  float mu_s[3], c_s[3];
  {
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
      mu_s[s] = k < a.K ? mu_l[s] : 0.f;
      c_s[s] = k < a.K ? (-0.5f * kLog2e) / (sg_l[s] * sg_l[s]) : 0.f;
    }
  }
  __syncthreads();
  for (int i = 0;; ++i) {
    const int cur = i & 1;
    const StreamMeta* m = meta + cur;
    if (m->pair < 0) break;
    const int n = m->n_unique;
    QueryPass<NV> qp;
    qp.den_my = m->qden[myq];
    qp.id_my = m->qid[myq];
    const unsigned* lst = list + cur * cap;
    const float4* ql = qrows + cur * kQT * NV * 16;
    float acc[3] = {0.f, 0.f, 0.f};
    float rowsum = 0.f;
    for (int t0 = g; t0 < n; t0 += kGroupsPerWG) {
      const unsigned e = lst[t0];
      RowRegs<NV> d[1];
      load_row<NV>(a.packed, (int64_t)(e & kIdMask), lane16, d[0]);
      float x[1];
      int qoff = 0;
      asm volatile("" : "+v"(qoff));
      rows_sim_my<NV, 1, true>(d, qp, ql + qoff, lane16, x);
      const float mlt = (float)(e >> kIdBits);
      rowsum = __builtin_fmaf(mlt, x[0], rowsum);
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const float adj = x[0] - mu_s[s];
        acc[s] += mlt * __builtin_amdgcn_exp2f(adj * adj * c_s[s]);
      }
    }
    // the wave's four groups: (g0 + g1) + (g2 + g3), lanes 0..15 of the wave store
    float v[4] = {acc[0], acc[1], acc[2], rowsum};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      v[c] += __shfl_xor(v[c], 16, 64);
      v[c] += __shfl_xor(v[c], 32, 64);
    }
    if (lane < 16) *reinterpret_cast<float4*>(partial + cur * 256 + (wave * kGroup + lane) * 4) = make_float4(v[0], v[1], v[2], v[3]);
    __syncthreads();
  }
}

}  // namespace

namespace {

// persistent grid of the streaming kernel: workgroups the device holds at once (occupancy x CUs), looked up once per (device, NV, LDS size)
template <int NV, bool ID32>
int stream_grid(size_t smem) {
  struct Entry { int dev; size_t smem; int grid; };
  static thread_local Entry cache{-1, 0, 0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (cache.dev == dev && cache.smem == smem) return cache.grid;
  int cus = 0, per_cu = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, knrm_stream_kernel<NV, ID32>, kStreamThreads, smem) != hipSuccess) return 0;
  cache = Entry{dev, smem, cus * per_cu};
  return cache.grid;
}

int knrm_launch(const IdSource& ids, int B, int Q, int L, const float* packed, int64_t V, int D, const float* mu,
                const float* sigma, int K, const float* w1, const float* b1, int hidden, const float* w2, const float* b2,
                int scoretanh, float* out, int* status, void* workspace, size_t workspace_bytes, unsigned flags, void* stream) {
  if (!packed || !mu || !sigma || !w1 || !b1 || !out || !status) return CAPAMD_ERR_ARG;
  if (B < 0 || Q < 1 || L < 1 || V < 1 || K < 1 || K > kMaxK || hidden < 0 || hidden > kMaxHidden) return CAPAMD_ERR_ARG;
  if (hidden > 0 && (!w2 || !b2)) return CAPAMD_ERR_ARG;
  if (capamd_packed_row_stride(D) < 0 || L > 32768 || V > 0x7fffffffLL) return CAPAMD_ERR_ARG;
  KnrmArgs a{ids, B, Q, L, packed, V, mu, sigma, K, w1, b1, hidden, w2, b2, scoretanh, out, status, nullptr, nullptr, nullptr};
  const size_t smem = (size_t)((L + 3) & ~3) * 8 + (1024 + 144 + 48 + kMaxHidden + 48 + 8 + 64 + 2 * kHashSlots) * 4 + (size_t)kQT * kMaxNV * 16 * 16;
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
  if (stream_mode && !variant && workspace && workspace_bytes >= sizeof(int) && Q <= kQT && L <= kDedupMaxL && V <= (int64_t)(1u << kIdBits) &&
      (B > 3072 || stream_mode == 2)) {
    const int nv = nv_for_dim(D);
    const size_t ssm = (size_t)2 * kQT * nv * 16 * 16 + (size_t)2 * ((L + 3) & ~3) * 4 + 2 * 256 * 4 + kHashSlots * 4 + 2 * sizeof(StreamMeta) + 80 * 4;
    const bool id32 = ids.q32 != nullptr;
    int grid = 0;
#define GRID_S(NV_) grid = id32 ? stream_grid<NV_, true>(ssm) : stream_grid<NV_, false>(ssm)
    switch (nv) {
      case 1: GRID_S(1); break;
      case 2: GRID_S(2); break;
      case 3: GRID_S(3); break;
      case 4: GRID_S(4); break;
      default: GRID_S(5); break;
    }
#undef GRID_S
    if (grid > 0) {
      if (grid > B) grid = B;
      if (hipMemsetAsync(workspace, 0, sizeof(int), s) != hipSuccess) return CAPAMD_ERR_LAUNCH;
      int* ticket = static_cast<int*>(workspace);
#define LAUNCH_S(NV_)                                                                                                        \
  if (id32) hipLaunchKernelGGL((knrm_stream_kernel<NV_, true>), dim3(grid), dim3(kStreamThreads), ssm, s, a, ticket);        \
  else hipLaunchKernelGGL((knrm_stream_kernel<NV_, false>), dim3(grid), dim3(kStreamThreads), ssm, s, a, ticket)
      switch (nv) {
        case 1: LAUNCH_S(1); break;
        case 2: LAUNCH_S(2); break;
        case 3: LAUNCH_S(3); break;
        case 4: LAUNCH_S(4); break;
        default: LAUNCH_S(5); break;
      }
#undef LAUNCH_S
      return hipGetLastError() == hipSuccess ? CAPAMD_OK : CAPAMD_ERR_LAUNCH;
    }
  }
#define LAUNCH(NV_, U_, QL_, W_) hipLaunchKernelGGL((knrm_forward_kernel<NV_, U_, QL_, W_>), dim3(B), dim3(kThreads), smem, s, a)
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
  const size_t smem = (size_t)((L + 3) & ~3) * 8 + (3072 + 144 + 48 + kMaxHidden + 48 + 8 + 64 + 2 * kHashSlots) * 4 + (size_t)kQT * kMaxNV * 16 * 16;
  hipStream_t s = (hipStream_t)stream;
  (void)hipGetLastError();
#define LAUNCH(NV_) hipLaunchKernelGGL((knrm_forward_kernel<NV_, 1, true, 4, true>), dim3(B), dim3(kThreads), smem, s, a)
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
