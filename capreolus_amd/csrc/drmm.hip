// Fused DRMM forward for gfx950: gather -> cosine interaction -> matching histogram (integer
// bin counts in LDS) -> CH/NH/LCH -> per-term feed-forward net -> IDF/TV term gate -> score.
//
// Reference semantics: DRMM_class._hist_map / _term_gate / forward
// (capreolus/reranker/DRMM.py:41-81, :83-99, :101-116) on SimilarityMatrix (common.py:143-182).
//
// Same work layout as knrm.hip (one workgroup per pair, 16 lanes per document term, real terms
// compacted in LDS).  What differs is the back end:
//   * bin(sim) = first i with sim < edge_i, else none (DRMM.py:62-69 builds this as differences of
//     cumulative counts); an arithmetic guess is corrected against the fp32 edge table so the
//     comparisons are exactly the reference's `sim < edge`; the exact-match bin counts
//     0.999 < sim < 1.001 (DRMM.py:66).  Counts are integers accumulated with LDS atomics, so the
//     result does not depend on the order terms are visited.
//   * pad document positions carry +1e7 in the reference (DRMM.py:57) -> they fall in no bin and
//     are skipped; OOV document terms (id < 0) have sim exactly 0 -> added to bin(0) in closed form.
#include "capreolus_amd.h"
#include "interaction.h"
#include "interaction_stream.h"
#include <stdlib.h>

using namespace capamd;

#ifndef CAPAMD_DRMM_ABLATE
#define CAPAMD_DRMM_ABLATE 0   // profiling builds only: 1 = no binning (the histogram stays empty), 2 = no feed-forward net / gate, 3 = both
#endif

namespace {

constexpr int kMaxBins = 64;   // nbins + 1 <= 64
constexpr int kMaxNodes = 64;
constexpr int kMaxQ = 32;

struct DrmmArgs {
  IdSource ids;
  const float* idf;  // [B, Q], or [NQ, Q] indexed by the pair's query row in indexed mode
  int B, Q, L;
  const float* packed;
  int64_t V;
  int D;
  const float* edges;
  int nbins, hist_type, gate_type;
  const float* gate_w;
  const float* emb_raw;
  int64_t ld;
  const float* w1;
  const float* b1;
  int nodes;
  const float* w2;
  const float* b2;
  const float* out_w;
  const float* out_b;
  float* out;
  int32_t* counts_out;
  int* status;
  float* feat_out;  // optional [B, Q, nbins+1]: the histogram features after CH/NH/LCH (input of the feed-forward net)
  const int64_t* d64_b;  // training step: a second block of documents - pairs split .. B - 1 take query row (b - split) and row (b - split) of d64_b
  int split;
};

__device__ __forceinline__ float wave_sum(float v) { return wave_allreduce_sum(v); }   // (interaction.h: DPP + readlane, no LDS-pipe permutes)

__device__ __forceinline__ int bin_of(float x, const float* edges, int nbins) {
  int bi = (int)floorf((x + 1.f) * (0.5f * (float)nbins));
  bi = bi < 0 ? 0 : (bi > nbins ? nbins : bi);
  // the arithmetic guess is almost always right: both neighbouring edges are read at once (one LDS latency), the walks below run only
  // when the guess is off; the result is the same exact `x < edge` classification either way
  const float e_lo = edges[bi > 0 ? bi - 1 : 0], e_hi = edges[bi < nbins ? bi : nbins - 1];
  if (bi > 0 && x < e_lo) {
    --bi;
    while (bi > 0 && x < edges[bi - 1]) --bi;
  } else if (bi < nbins && !(x < e_hi)) {
    ++bi;
    while (bi < nbins && !(x < edges[bi])) ++bi;
  }
  return bi;  // == nbins: at or above the last edge (1.0): no regular bin
}

template <int NV, int U, bool QLDS, int MINW>
__global__ __launch_bounds__(kThreads, MINW) void drmm_forward_kernel(DrmmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  int* tok = reinterpret_cast<int*>(smem_raw);
  const int tok_cap = (a.L + 3) & ~3;
  int* hist = tok + tok_cap;                                   // [kQT][kMaxBins]
  float* edges = reinterpret_cast<float*>(hist + kQT * kMaxBins);  // [kMaxBins]
  float* zlds = edges + kMaxBins;                              // [kMaxQ]
  float* glds = zlds + kMaxQ;                                  // [kMaxQ]
  int* wave_cnt = reinterpret_cast<int*>(glds + kMaxQ);        // [48]: distinct_terms' per-wave counts
  float4* qlds = reinterpret_cast<float4*>(wave_cnt + 48);     // QLDS: [kQT][NV*16] float4
  int* mult = reinterpret_cast<int*>(qlds + kQT * kMaxNV * 16);  // [tok_cap] multiplicity of tok[k]
  int* hkey = mult + tok_cap;                                    // [kHashSlots] phase 1 only
  int* hfirst = hkey + kHashSlots;                               // [kHashSlots] phase 1 only

  const int tid = threadIdx.x;
  const int lane16 = tid & 15;
  const int g = tid >> 4;
  const int wave = tid >> 6;
  const int lane = tid & 63;
  const int b = blockIdx.x;
  const int NB = a.nbins + 1;
  PairIds ids = pair_ids(a.ids, a.d64_b && b >= a.split ? b - a.split : b, a.Q, a.L);
  if (a.d64_b && b >= a.split) ids.d64 = a.d64_b + (int64_t)(b - a.split) * a.L;

  if (tid < a.nbins) edges[tid] = a.edges[tid];
  // the first feed-forward layer's weights go where the hash of the distinct-term pass was (dead after it), when they fit
  float* w1lds = reinterpret_cast<float*>(hkey);
  const bool w1_in_lds = a.out && a.L <= kDedupMaxL && a.nodes * NB <= 2 * kHashSlots;   // (longer documents: no hash region in the carve-out)

  // ---- the document's distinct real terms with their multiplicities; OOV count (interaction.h) --------
  const TermList tl = distinct_terms(ids, a.L, a.V, a.status, tok, mult, hkey, hfirst, wave_cnt);
  const int n_real = tl.n_unique, n_oov = tl.n_oov;
  if (w1_in_lds)
    for (int i = tid; i < a.nodes * NB; i += kThreads) w1lds[i] = a.w1[i];   // (visible after the barrier that follows the histogram loop)

  for (int q0 = 0; q0 < a.Q; q0 += kQT) {
    // DRMM cannot score an OOV query term: the reference indexes the embedding un-clamped (DRMM.py:109)
    if (tid < kQT && q0 + tid < a.Q && ids.q(q0 + tid) < 0) atomicOr(a.status, kErrQueryOOV);
    QueryPass<NV> qp;
    if (QLDS)
      load_query_pass_lds<NV>(a.packed, ids, a.Q, q0, a.V, tid, kThreads, lane16, qlds, qp, a.status);
    else
      load_query_pass<NV>(a.packed, ids, a.Q, q0, a.V, lane16, qp, a.status);
    if (qp.id_my < 0) qp.id_my = 0;
    for (int i = tid; i < kQT * kMaxBins; i += kThreads) hist[i] = 0;
    __syncthreads();

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
      if (QLDS) asm volatile("" : "+v"(qoff));
      rows_sim_my<NV, U, QLDS>(d, qp, qlds + qoff, lane16, x);
      if (CAPAMD_DRMM_ABLATE & 1) {
        float keep = 0.f;
#pragma unroll
        for (int u = 0; u < U; ++u) keep += x[u];
        asm volatile("" ::"v"(keep));
      } else if (lane16 < kQT) {
        int* h = hist + lane16 * kMaxBins;
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (has[u]) {
            const int m = mult[t0 + u * kGroupsPerWG];     // how often the document repeats this term
            const int bu = bin_of(x[u], edges, a.nbins);
            if (bu < a.nbins) atomicAdd(&h[bu], m);
            if (x[u] > 0.999f && x[u] < 1.001f) atomicAdd(&h[a.nbins], m);
          }
      }
    }
    __syncthreads();
    if (tid < kQT && n_oov > 0) {  // OOV document terms: sim == 0 exactly
      const int bz = bin_of(0.f, edges, a.nbins);
      if (bz < a.nbins) hist[tid * kMaxBins + bz] += n_oov;
    }
    __syncthreads();

    // ---- per query term (one wave each): histogram transform + feed-forward net ------------
    const int q = q0 + wave;
    if (q < a.Q) {
      const int* h = hist + wave * kMaxBins;
      if (a.counts_out && lane < NB) a.counts_out[((int64_t)b * a.Q + q) * NB + lane] = h[lane];
      float hv = lane < NB ? (float)(h[lane] + 1) : 0.f;  // DRMM.py:71
      if (a.hist_type == 1) {                              // NH (DRMM.py:72-74)
        const float hs = wave_sum(hv);
        hv = hv / hs;
      } else if (a.hist_type == 2) {                       // LCH (DRMM.py:75-76)
        hv = lane < NB ? logf(hv) : 0.f;
      }
      if (a.feat_out && lane < NB) a.feat_out[((int64_t)b * a.Q + q) * NB + lane] = hv;
      if (CAPAMD_DRMM_ABLATE & 2) {   // profiling builds only: no feed-forward net / gate
        if (lane == 0) { zlds[q] = hv; glds[q] = 0.f; }
      } else
      if (a.out) {  // (feature call: the net / gate run under autograd on the host side -- training step, row N3)
      // ffw (DRMM.py:25): lane i holds histogram entry i; node n = one wave-wide dot product, kept by lane n.  (An earlier version
      // broadcast the entries one by one - 30 dependent ds_bpermute round trips per query term - and the per-pair tail cost 23 %
      // of the kernel.)
      // (every small operand of the tail is requested up front, so that the serial chain below waits for memory once)
      const float b1v = lane < a.nodes ? a.b1[lane] : 0.f, w2v = lane < a.nodes ? a.w2[lane] : 0.f, b2v = a.b2[0];
      const int64_t qid = ids.q(q);
      const float gate0 = a.gate_type == 0 ? a.gate_w[0] * a.idf[(int64_t)ids.qrow * a.Q + q] : 0.f;
      float acc = 0.f;
      for (int n = 0; n < a.nodes; ++n) {
        const float wv = lane < NB ? (w1_in_lds ? w1lds[n * NB + lane] : a.w1[n * NB + lane]) : 0.f;
        const float sn = wave_sum(wv * hv);
        if (lane == n) acc = sn;
      }
      acc += b1v;
      const float contrib = lane < a.nodes ? w2v * tanhf(acc) : 0.f;
      const float o = wave_sum(contrib) + b2v;
      // term gate logit (DRMM.py:83-95)
      float gl;
      if (a.gate_type == 0) {
        gl = gate0;
      } else {
        const float* e = a.emb_raw + (qid > 0 && qid < a.V ? qid : 0) * a.ld;
        float p = 0.f;
        for (int c = lane; c < a.D; c += 64) p = __builtin_fmaf(a.gate_w[c], e[c], p);
        gl = wave_sum(p);
      }
      if (qid == 0) gl += -1e7f;
      if (lane == 0) {
        zlds[q] = tanhf(o);
        glds[q] = gl;
      }
      }
    }
    __syncthreads();
  }

  if (tid == 0 && a.out) {  // softmax gate + output layer (DRMM.py:97-98, :112-114)
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


// =====================================================================================================================================
// Streaming form (interaction_stream.h): persistent five-wave workgroups, the model's part as a policy.
//   gathering waves: bin every gathered similarity into hist[buf][query term][bin] (LDS atomics, integers: order independent)
//   list wave:       OOV document terms in closed form, then per query term CH / NH / LCH -> feed-forward net -> gate logit, the softmax
//                    gate and the output layer (DRMM.py:71-116) - the tail that cost the one-pair-per-workgroup kernel 23 % of its
//                    time (a workgroup in its tail requests no rows) now runs beside the next pair's gather.
struct DrmmStream {
  using Args = DrmmArgs;
  struct Gather {};
  static constexpr int kW1Lds = 512;   // floats of the first layer kept in LDS (nodes x (nbins + 1); the reference's 5 x 30 = 150)
  // hist[2][kQT][kMaxBins] | edges[kMaxBins] | w1[kW1Lds]
  __host__ __device__ static size_t lds_bytes(const Args&) { return (2 * kQT * kMaxBins + kMaxBins + kW1Lds) * 4; }
  __device__ static int* hist(char* lds, int buf) { return reinterpret_cast<int*>(lds) + buf * kQT * kMaxBins; }
  __device__ static float* edges(char* lds) { return reinterpret_cast<float*>(lds) + 2 * kQT * kMaxBins; }
  __device__ static float* w1(char* lds) { return edges(lds) + kMaxBins; }

  __device__ static void list_init(const Args& a, char* lds, int lane) {
    if (lane < a.nbins) edges(lds)[lane] = a.edges[lane];
    const int NB = a.nbins + 1;
    if (a.nodes * NB <= kW1Lds)
      for (int i = lane; i < a.nodes * NB; i += 64) w1(lds)[i] = a.w1[i];
    wave_fence();
  }
  __device__ static void prepare(const Args&, char* lds, int buf, int lane) {
    reinterpret_cast<int4*>(hist(lds, buf))[lane] = make_int4(0, 0, 0, 0);     // kQT * kMaxBins = 256 ints
  }

  __device__ static void finish(const Args& a, const StreamSrc& src, char* lds, int buf, const StreamMeta* meta, int lane) {
    int* H = hist(lds, buf);
    const float* E = edges(lds);
    const int NB = a.nbins + 1;
    const int b = meta->pair;
    const int n_oov = meta->n_oov;
    if (lane < kQT && n_oov > 0) {  // OOV document terms: sim == 0 exactly
      const int bz = bin_of(0.f, E, a.nbins);
      if (bz < a.nbins) H[lane * kMaxBins + bz] += n_oov;
    }
    // DRMM cannot score an OOV query term: the reference indexes the embedding un-clamped (DRMM.py:109)
    if (lane < src.Q && lane < kQT && meta->qid[lane] < 0) atomicOr(src.status, kErrQueryOOV);
    wave_fence();
    const bool w1_in_lds = a.nodes * NB <= kW1Lds;
    const int qrow = src.ids.q32 ? src.ids.pair_q[b] : b;
    // (every small operand of the tail is requested up front, so that the serial chain below waits for memory once)
    const float b1v = lane < a.nodes ? a.b1[lane] : 0.f, w2v = lane < a.nodes ? a.w2[lane] : 0.f, b2v = a.b2[0];
    const float gw0 = a.gate_type == 0 ? a.gate_w[0] : 0.f;
    const float idfv = (a.gate_type == 0 && lane < src.Q) ? a.idf[(int64_t)qrow * src.Q + lane] : 0.f;
    const float ow = a.out_w[0], ob = a.out_b[0];
    float z[kQT], gl[kQT];
#pragma unroll
    for (int q = 0; q < kQT; ++q) {
      z[q] = 0.f;
      gl[q] = -3.0e38f;
      if (q < src.Q) {
        const int* h = H + q * kMaxBins;
        if (a.counts_out && lane < NB) a.counts_out[((int64_t)b * src.Q + q) * NB + lane] = h[lane];
        float hv = lane < NB ? (float)(h[lane] + 1) : 0.f;  // DRMM.py:71
        if (a.hist_type == 1) {                              // NH (DRMM.py:72-74)
          const float hs = wave_sum(hv);
          hv = hv / hs;
        } else if (a.hist_type == 2) {                       // LCH (DRMM.py:75-76)
          hv = lane < NB ? logf(hv) : 0.f;
        }
        // ffw (DRMM.py:25): lane i holds histogram entry i; node n = one wave-wide dot product, kept by lane n
        float acc = 0.f;
        for (int n = 0; n < a.nodes; ++n) {
          const float wv = lane < NB ? (w1_in_lds ? w1(lds)[n * NB + lane] : a.w1[n * NB + lane]) : 0.f;
          const float sn = wave_sum(wv * hv);
          if (lane == n) acc = sn;
        }
        acc += b1v;
        const float contrib = lane < a.nodes ? w2v * tanhf(acc) : 0.f;
        const float o = wave_sum(contrib) + b2v;
        // term gate logit (DRMM.py:83-95)
        const int qid = meta->qid[q];
        float g;
        if (a.gate_type == 0) {
          g = gw0 * __shfl(idfv, q, 64);
        } else {
          const float* e = a.emb_raw + (qid > 0 && qid < src.V ? qid : 0) * a.ld;
          float p = 0.f;
          for (int c = lane; c < a.D; c += 64) p = __builtin_fmaf(a.gate_w[c], e[c], p);
          g = wave_sum(p);
        }
        if (qid == 0) g += -1e7f;
        z[q] = tanhf(o);
        gl[q] = g;
      }
    }
    // softmax gate + output layer (DRMM.py:97-98, :112-114): the same fixed order as the one-pair-per-workgroup kernel
    float m = gl[0];
#pragma unroll
    for (int q = 1; q < kQT; ++q)
      if (q < src.Q) m = fmaxf(m, gl[q]);
    float den = 0.f, num = 0.f;
#pragma unroll
    for (int q = 0; q < kQT; ++q)
      if (q < src.Q) {
        const float e = expf(gl[q] - m);
        den += e;
        num = __builtin_fmaf(e, z[q], num);
      }
    if (lane == 0) a.out[b] = __builtin_fmaf(ow, num / den, ob);
    wave_fence();
  }

  __device__ static void gather_init(const Args&, Gather&, int) {}
  template <int NV>
  __device__ static void pair_begin(const Args&, Gather&, QueryPass<NV>& qp, const StreamMeta*, int) {
    if (qp.id_my < 0) qp.id_my = 0;      // an OOV query term matches nothing here (and is reported through the status word)
  }
  __device__ static void row(const Args& a, Gather&, float x, unsigned entry, char* lds, int buf, int lane16) {
    if (lane16 < kQT) {
      int* h = hist(lds, buf) + lane16 * kMaxBins;
      const int m = (int)(entry >> kIdBits);           // how often the document repeats this term (0: a filler beyond the list)
      const int bu = bin_of(x, edges(lds), a.nbins);
      if (m > 0 && bu < a.nbins) atomicAdd(&h[bu], m);
      if (m > 0 && x > 0.999f && x < 1.001f) atomicAdd(&h[a.nbins], m);
    }
  }
  __device__ static void pair_end(const Args&, Gather&, char*, int, int, int) {}
};

}  // namespace

namespace {

int drmm_launch(const IdSource& ids, const float* idf, int B, int Q, int L, const float* packed, int64_t V, int D,
                const float* edges, int nbins, int hist_type, int gate_type, const float* gate_w, const float* emb_raw, int64_t ld,
                const float* w1, const float* b1, int nodes, const float* w2, const float* b2, const float* out_w,
                const float* out_b, float* out, int32_t* counts_out, int* status, void* workspace, size_t workspace_bytes, unsigned flags,
                void* stream) {
  (void)flags;   // (CAPAMD_LAUNCH_CONCURRENT: this kernel uses its occupancy-oriented variant at every batch size already)
  if (!idf || !packed || !edges || !gate_w || !w1 || !b1 || !w2 || !b2 || !out_w || !out_b || !out || !status) return CAPAMD_ERR_ARG;
  if (B < 0 || Q < 1 || Q > kMaxQ || L < 1 || L > 32768 || V < 1 || V > 0x7fffffffLL) return CAPAMD_ERR_ARG;
  if (nbins < 1 || nbins + 1 > kMaxBins || nodes < 1 || nodes > kMaxNodes) return CAPAMD_ERR_ARG;
  if (hist_type < 0 || hist_type > 2 || gate_type < 0 || gate_type > 1) return CAPAMD_ERR_ARG;
  if (gate_type == 1 && (!emb_raw || ld < D)) return CAPAMD_ERR_ARG;
  if (capamd_packed_row_stride(D) < 0) return CAPAMD_ERR_ARG;
  DrmmArgs a{ids, idf, B, Q, L, packed, V, D, edges, nbins, hist_type, gate_type, gate_w, emb_raw, ld,
             w1, b1, nodes, w2, b2, out_w, out_b, out, counts_out, status, nullptr};
  const size_t smem = (size_t)((L + 3) & ~3) * 8 + (size_t)(kQT * kMaxBins + kMaxBins + 2 * kMaxQ + 48) * 4 + dedup_hash_bytes(L) + (size_t)kQT * kMaxNV * 16 * 16;
  hipStream_t s = (hipStream_t)stream;
  (void)hipGetLastError();
  // Launches that outnumber the workgroups the chip holds run the streaming kernel (interaction_stream.h; needs the caller's workspace
  // word, <= kQT query terms, L <= kDedupMaxL, ids < 2^22).  CAPAMD_DRMM_STREAM (profiling): 0 = never, 2 = whenever the geometry allows.
  static const int stream_mode = [] {
    const char* e = getenv("CAPAMD_DRMM_STREAM");
    return e ? atoi(e) : 1;
  }();
  const bool cacheable = (int64_t)V * row_stride_for_dim(D) * 4 <= (1LL << 30);   // (see knrm.hip: HBM-bound tables keep the one-pair-per-workgroup kernel)
  if (stream_mode && ((B > 3072 && cacheable) || stream_mode == 2)) {
    const StreamSrc src{ids, B, Q, L, packed, V, status};
    int rc = CAPAMD_OK;
    if (stream_launch<DrmmStream>(src, a, D, workspace, workspace_bytes, s, &rc)) return rc;
  }
#define LAUNCH(NV_, U_, QL_, W_)                                                     \
  do {                                                                               \
    auto kern = drmm_forward_kernel<NV_, U_, QL_, W_>;                               \
    if (const int bad = lds_budget(kern, smem)) return bad;                          \
    hipLaunchKernelGGL(kern, dim3(B), dim3(kThreads), smem, s, a);                   \
  } while (0)
  switch (nv_for_dim(D)) {
    case 1: LAUNCH(1, 2, false, 3); break;
    case 2: LAUNCH(2, 2, false, 3); break;
    case 3: LAUNCH(3, 2, false, 3); break;
    case 4: LAUNCH(4, 2, false, 3); break;
    default: LAUNCH(5, 1, true, 6); break;  // same choice as knrm.hip (measured there)
  }
#undef LAUNCH
  return hipGetLastError() == hipSuccess ? CAPAMD_OK : CAPAMD_ERR_LAUNCH;
}

}  // namespace

extern "C" int capamd_drmm_forward(const int64_t* q_ids, const int64_t* d_ids, const float* idf, int B, int Q, int L,
                                   const float* packed, int64_t V, int D, const float* edges, int nbins, int hist_type,
                                   int gate_type, const float* gate_w, const float* emb_raw, int64_t ld, const float* w1,
                                   const float* b1, int nodes, const float* w2, const float* b2, const float* out_w,
                                   const float* out_b, float* out, int32_t* counts_out, int* status, void* workspace,
                                   size_t workspace_bytes, unsigned flags, void* stream) {
  if (B == 0) return CAPAMD_OK;
  if (!q_ids || !d_ids) return CAPAMD_ERR_ARG;
  const IdSource ids{q_ids, d_ids, nullptr, nullptr, nullptr, nullptr};
  return drmm_launch(ids, idf, B, Q, L, packed, V, D, edges, nbins, hist_type, gate_type, gate_w, emb_raw, ld, w1, b1, nodes, w2,
                     b2, out_w, out_b, out, counts_out, status, workspace, workspace_bytes, flags, stream);
}

static int drmm_features_launch(DrmmArgs a, void* stream) {
  const int B = a.B, L = a.L;
  const size_t smem = (size_t)((L + 3) & ~3) * 8 + (size_t)(kQT * kMaxBins + kMaxBins + 2 * kMaxQ + 48) * 4 + dedup_hash_bytes(L) + (size_t)kQT * kMaxNV * 16 * 16;
  hipStream_t s = (hipStream_t)stream;
  (void)hipGetLastError();
#define LAUNCH(NV_, U_, QL_, W_)                                                     \
  do {                                                                               \
    auto kern = drmm_forward_kernel<NV_, U_, QL_, W_>;                               \
    if (const int bad = lds_budget(kern, smem)) return bad;                          \
    hipLaunchKernelGGL(kern, dim3(B), dim3(kThreads), smem, s, a);                   \
  } while (0)
  switch (nv_for_dim(a.D)) {
    case 1: LAUNCH(1, 2, false, 3); break;
    case 2: LAUNCH(2, 2, false, 3); break;
    case 3: LAUNCH(3, 2, false, 3); break;
    case 4: LAUNCH(4, 2, false, 3); break;
    default: LAUNCH(5, 1, true, 6); break;
  }
#undef LAUNCH
  return hipGetLastError() == hipSuccess ? CAPAMD_OK : CAPAMD_ERR_LAUNCH;
}

extern "C" int capamd_drmm_features(const int64_t* q_ids, const int64_t* d_ids, int B, int Q, int L, const float* packed, int64_t V,
                                    int D, const float* edges, int nbins, int hist_type, float* feat_out, int* status, void* stream) {
  if (B == 0) return CAPAMD_OK;
  if (!q_ids || !d_ids || !packed || !edges || !feat_out || !status) return CAPAMD_ERR_ARG;
  if (B < 0 || Q < 1 || Q > kMaxQ || L < 1 || L > 32768 || V < 1 || V > 0x7fffffffLL) return CAPAMD_ERR_ARG;
  if (nbins < 1 || nbins + 1 > kMaxBins || hist_type < 0 || hist_type > 2 || capamd_packed_row_stride(D) < 0) return CAPAMD_ERR_ARG;
  const IdSource ids{q_ids, d_ids, nullptr, nullptr, nullptr, nullptr};
  DrmmArgs a{ids, nullptr, B, Q, L, packed, V, D, edges, nbins, hist_type, 0, nullptr, nullptr, 0, nullptr, nullptr, 1, nullptr, nullptr,
             nullptr, nullptr, nullptr, nullptr, status, feat_out};
  return drmm_features_launch(a, stream);
}

// ---- one DRMM training step without a host round trip (reference trainer/pytorch.py:93-108) -----------------------
// score() on the positive and the negative documents - matching histograms (the kernel above over the 2 B documents), the 30 -> nodes -> 1
// tanh net per query term (DRMM.py:25), the softmax idf gate (:83-99, gateType = IDF), the output layer (:114) - the trainer's pairwise loss,
// backward through all of it and torch.optim.Adam's update of the seven parameter tensors in place, in two launches.  ptrs: DEVICE array of
// 3 x 7 device pointers - ffw.0.weight [nodes, NB], ffw.0.bias [nodes], ffw.2.weight [nodes], ffw.2.bias [1], gates.weight [1],
// output_layer.weight [1], output_layer.bias [1] (NB = nbins + 1), then their exp_avg, then their exp_avg_sq.  The caller owns the step
// count: step_size = lr / (1 - beta1^t), bc2_sqrt = sqrt(1 - beta2^t), computed in double.
constexpr int kMaxStepBatch = 1024, kMaxStepNodes = 16;

struct DrmmStepArgs {
  const float* feat;      // [2 B, Q, NB]: the positive documents, then the negative ones
  const int64_t* q_ids;   // [B, Q]
  const float* idf;       // [B, Q]
  int B, Q, NB, nodes;
  float* const* ptrs;
  int loss_type;
  float step_size, one_minus_beta1, beta2, eps, bc2_sqrt;
  float* loss_out;
  float* inter;           // workspace [B][Q]: d loss / d gate logit
};

// Phases of the one workgroup (every sum in a fixed order):  A1 a thread per (document, query term): the net's hidden activations and z;
// A2 a thread per pair: gate, scores, loss, the gradients at the net's output and at the gate logits;  B a thread per parameter element:
// its gradient over the batch from the LDS copies, then Adam's update.
constexpr int kMaxStepDQ = 1024;       // 2 B Q (batches beyond it keep the graph route)

__global__ __launch_bounds__(256) void drmm_step_kernel(DrmmStepArgs a) {
  __shared__ float W1[kMaxStepNodes * kMaxBins], B1[kMaxStepNodes], W2[kMaxStepNodes], sc_par[4];
  __shared__ float T1[kMaxStepDQ][kMaxStepNodes + 1], Z[kMaxStepDQ], DO[kMaxStepDQ];       // hidden activations | z | d loss / d o
  __shared__ float lsum[kMaxStepBatch], dso[kMaxStepBatch], dbo[kMaxStepBatch];
  const int tid = threadIdx.x, Q = a.Q, NB = a.NB, N = a.nodes, DQ = 2 * a.B * Q;
  for (int i = tid; i < N * NB; i += 256) W1[i] = a.ptrs[0][i];
  if (tid < N) { B1[tid] = a.ptrs[1][tid]; W2[tid] = a.ptrs[2][tid]; }
  if (tid == 0) { sc_par[0] = a.ptrs[3][0]; sc_par[1] = a.ptrs[4][0]; sc_par[2] = a.ptrs[5][0]; sc_par[3] = a.ptrs[6][0]; }
  __syncthreads();
  const float b2 = sc_par[0], wg = sc_par[1], wo = sc_par[2], bo = sc_par[3], inv_b = 1.f / (float)a.B;
  float* dgl = a.inter;                  // [B][Q] d loss / d gate logit (workspace)
  for (int dq = tid; dq < DQ; dq += 256) {          // A1
    const float* H = a.feat + (int64_t)dq * NB;
    float o = b2;
    for (int n = 0; n < N; ++n) {
      float v = B1[n];
      for (int j = 0; j < NB; ++j) v = __builtin_fmaf(W1[n * NB + j], H[j], v);
      const float t1 = tanhf(v);
      T1[dq][n] = t1;
      o = __builtin_fmaf(W2[n], t1, o);
    }
    Z[dq] = tanhf(o);
  }
  __syncthreads();
  for (int i = tid; i < a.B; i += 256) {            // A2
    float g[kMaxQ], sagg[2], score[2];
    float mx = -INFINITY;
    for (int q = 0; q < Q; ++q) {
      g[q] = wg * a.idf[(int64_t)i * Q + q] + (a.q_ids[(int64_t)i * Q + q] == 0 ? -1e7f : 0.f);
      mx = fmaxf(mx, g[q]);
    }
    float den = 0.f;
    for (int q = 0; q < Q; ++q) { g[q] = expf(g[q] - mx); den += g[q]; }
    for (int q = 0; q < Q; ++q) g[q] /= den;
    for (int h = 0; h < 2; ++h) {
      float acc = 0.f;
      for (int q = 0; q < Q; ++q) acc = __builtin_fmaf(g[q], Z[(h * a.B + i) * Q + q], acc);
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
    dso[i] = ds[0] * sagg[0] + ds[1] * sagg[1];       // d loss / d output_layer.weight of this pair
    dbo[i] = ds[0] + ds[1];
    float dg[kMaxQ];
    for (int q = 0; q < Q; ++q) dg[q] = 0.f;
    for (int h = 0; h < 2; ++h) {
      const float dagg = ds[h] * wo;
      float dot = 0.f;
      for (int q = 0; q < Q; ++q) dot = __builtin_fmaf(g[q], dagg * Z[(h * a.B + i) * Q + q], dot);
      for (int q = 0; q < Q; ++q) {
        const float z = Z[(h * a.B + i) * Q + q];
        dg[q] += g[q] * (dagg * z - dot);                               // softmax backward
        DO[(h * a.B + i) * Q + q] = dagg * g[q] * (1.f - z * z);       // d loss / d o_q (the second layer's pre-activation)
      }
    }
    for (int q = 0; q < Q; ++q) dgl[(int64_t)i * Q + q] = dg[q];
  }
  __syncthreads();
  __threadfence_block();
  const int n_el = N * NB + N + N + 4;
  for (int j = tid; j <= n_el; j += 256) {          // B
    if (j == n_el) {
      float l = 0.f;
      for (int i = 0; i < a.B; ++i) l += lsum[i];
      a.loss_out[0] = l * inv_b;
      continue;
    }
    float gsum = 0.f;
    int slot, el;
    if (j < N * NB) {                       // ffw.0.weight[n][jb] = sum d_o W2[n] (1 - t1^2) H[jb]
      slot = 0; el = j;
      const int n = j / NB, jb = j % NB;
      const float w2n = W2[n];
#pragma unroll 8
      for (int dq = 0; dq < DQ; ++dq) gsum = __builtin_fmaf(DO[dq] * w2n * (1.f - T1[dq][n] * T1[dq][n]), a.feat[(int64_t)dq * NB + jb], gsum);
    } else if (j < N * NB + N) {            // ffw.0.bias[n]
      slot = 1; el = j - N * NB;
      for (int dq = 0; dq < DQ; ++dq) gsum += DO[dq] * W2[el] * (1.f - T1[dq][el] * T1[dq][el]);
    } else if (j < N * NB + 2 * N) {        // ffw.2.weight[n] = sum d_o t1[n]
      slot = 2; el = j - N * NB - N;
      for (int dq = 0; dq < DQ; ++dq) gsum = __builtin_fmaf(DO[dq], T1[dq][el], gsum);
    } else {
      const int w = j - N * NB - 2 * N;     // 0: ffw.2.bias, 1: gates.weight, 2: output_layer.weight, 3: output_layer.bias
      slot = 3 + w; el = 0;
      if (w == 0) {
        for (int dq = 0; dq < DQ; ++dq) gsum += DO[dq];
      } else if (w == 1) {
        for (int i = 0; i < a.B * Q; ++i) gsum = __builtin_fmaf(dgl[i], a.idf[i], gsum);
      } else if (w == 2) {
        for (int i = 0; i < a.B; ++i) gsum += dso[i];
      } else {
        for (int i = 0; i < a.B; ++i) gsum += dbo[i];
      }
    }
    float* pp = a.ptrs[slot] + el;
    float* pm = a.ptrs[7 + slot] + el;
    float* pv = a.ptrs[14 + slot] + el;
    float m = *pm, v = *pv;
    m = m + (gsum - m) * a.one_minus_beta1;                    // exp_avg.lerp_(grad, 1 - beta1)
    v = v * a.beta2 + (1.f - a.beta2) * (gsum * gsum);         // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
    *pm = m;
    *pv = v;
    const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
    *pp = *pp - a.step_size * (m / denom);                     // param.addcdiv_(exp_avg, denom, value = -lr / bias_correction1)
  }
}

extern "C" size_t capamd_drmm_train_step_workspace_floats(int B, int Q, int nbins, int nodes) {
  return B > 0 && Q > 0 && nbins > 0 && nodes > 0 ? (size_t)2 * B * Q * (nbins + 1) + (size_t)B * Q : 0;
}

extern "C" int capamd_drmm_train_step(const int64_t* q_ids, const int64_t* pos_ids, const int64_t* neg_ids, const float* idf, int B, int Q, int L,
                                      const float* packed, int64_t V, int D, const float* edges, int nbins, int hist_type, int nodes, float* const* ptrs,
                                      int loss_type, float step_size, float one_minus_beta1, float beta2, float eps, float bc2_sqrt, float* loss_out,
                                      float* workspace, size_t workspace_floats, int* status, void* stream) {
  if (!q_ids || !pos_ids || !neg_ids || !idf || !packed || !edges || !ptrs || !loss_out || !workspace || !status) return CAPAMD_ERR_ARG;
  if (B < 1 || B > kMaxStepBatch || Q < 1 || Q > kMaxQ || L < 1 || L > 32768 || V < 1 || V > 0x7fffffffLL) return CAPAMD_ERR_ARG;
  if (nbins < 1 || nbins + 1 > kMaxBins || hist_type < 0 || hist_type > 2 || nodes < 1 || nodes > kMaxStepNodes || capamd_packed_row_stride(D) < 0) return CAPAMD_ERR_ARG;
  if (loss_type < 0 || loss_type > 1 || !(bc2_sqrt > 0.f) || 2 * B * Q > kMaxStepDQ) return CAPAMD_ERR_ARG;
  if (workspace_floats < capamd_drmm_train_step_workspace_floats(B, Q, nbins, nodes)) return CAPAMD_ERR_WORKSPACE;
  const int NB = nbins + 1;
  float* feat = workspace;
  float* inter = workspace + (size_t)2 * B * Q * NB;
  const IdSource ids{q_ids, pos_ids, nullptr, nullptr, nullptr, nullptr};
  DrmmArgs fa{ids, nullptr, 2 * B, Q, L, packed, V, D, edges, nbins, hist_type, 0, nullptr, nullptr, 0, nullptr, nullptr, 1, nullptr, nullptr,
              nullptr, nullptr, nullptr, nullptr, status, feat, neg_ids, B};
  const int rc = drmm_features_launch(fa, stream);
  if (rc != CAPAMD_OK) return rc;
  DrmmStepArgs a{feat, q_ids, idf, B, Q, NB, nodes, ptrs, loss_type, step_size, one_minus_beta1, beta2, eps, bc2_sqrt, loss_out, inter};
  hipLaunchKernelGGL(drmm_step_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? CAPAMD_OK : CAPAMD_ERR_LAUNCH;
}

extern "C" int capamd_drmm_forward_indexed(const int32_t* q_table, const int32_t* d_table, const float* idf_table,
                                           const int32_t* pair_q, const int32_t* pair_d, int B, int Q, int L, const float* packed,
                                           int64_t V, int D, const float* edges, int nbins, int hist_type, int gate_type,
                                           const float* gate_w, const float* emb_raw, int64_t ld, const float* w1, const float* b1,
                                           int nodes, const float* w2, const float* b2, const float* out_w, const float* out_b,
                                           float* out, int32_t* counts_out, int* status, void* workspace, size_t workspace_bytes,
                                           unsigned flags, void* stream) {
  if (B == 0) return CAPAMD_OK;
  if (!q_table || !d_table || !pair_q || !pair_d) return CAPAMD_ERR_ARG;
  const IdSource ids{nullptr, nullptr, q_table, d_table, pair_q, pair_d};
  return drmm_launch(ids, idf_table, B, Q, L, packed, V, D, edges, nbins, hist_type, gate_type, gate_w, emb_raw, ld, w1, b1, nodes,
                     w2, b2, out_w, out_b, out, counts_out, status, workspace, workspace_bytes, flags, stream);
}
