// KNRM / DRMM / DRMM-TKS over whole CANDIDATE LISTS (one query, its first-stage documents) for gfx950 - what PytorchTrainer.predict
// scores (reference capreolus/trainer/pytorch.py:310-353 over PredSampler's per-query lists, sampler/__init__.py:222-233; RerankTask
// hands it up to 1000 documents per query, task/rerank.py:22-23).  PACRR's list entry is in pacrr.hip; the shared passes in lists.h.
//
// The per-pair kernels (knrm.hip, drmm.hip, interaction_stream.h) gather one packed row per distinct term of every DOCUMENT: 179 rows
// = 229 KB per pair on the benchmark's lists, 14.7 GB per 64,000 pairs - and that gather is what they are bound by.  But the similarity
// of a document term to the query depends on (query, term) only, and the 1000 documents of a list share their vocabulary: a list's
// 303,000 tokens are ~50,000 distinct terms.  So per list:
//   1  mark    every document of the list flags its real term ids in a byte map over the vocabulary (plain byte stores: racing writers
//              write the same value)                                                                                    [lists.h]
//   2  sims    per (list, block of 1024 vocabulary ids): collect the flagged ids and for every one of them gather its packed row
//              ONCE, the four similarities to the list's query by the SAME arithmetic as the per-pair kernels (rows_dot2_pk /
//              sim_from_dots: bit-identical values) -> table[id]: the four floats, or DRMM's four histogram bins (a byte each: bin |
//              exact-match bit) - the binning is done once per distinct term, not once per position.  Workgroups are numbered so
//              that an XCD works on ONE id block of ALL lists at a time: the lists share most of a block's rows, so the rows come
//              from that XCD's L2.                                                                                       [lists.h]
//   3  pool    every document, a WAVE each: its ids are requested together, then their table entries (two memory round trips per
//              document - the per-pair kernels' chain entry -> row -> tail is one per ROW).  KNRM: a lane per (position, query term), K
//              exponentials into per-lane sums; DRMM: a lane per position, four integer bin counts; DRMM-TKS: a lane per (position,
//              query term) keeping a sorted top-k.  Passes of padding only (a document's tail) stop after the id load.  Then the
//              models' per-pair tails, in the same wave.  An XCD works on one list at a time: the hot part of the list's table
//              sits in its L2 (DRMM's 1.6 MB table all of it).                                                          [this file]
// Rows gathered: 4.1 GB instead of 14.7, most of them L2 hits.
// DRMM's bin counts are integers of bit-identical similarities and DRMM-TKS's top-k are selections of them: scores bit-exact with the
// per-pair kernels.  KNRM sums the same kernel values in another order (per lane over its positions, then over the lanes): equal to
// fp32 rounding of the sums (1e-6 relative).
// Workspace (caller-owned): per list in flight a table (16 B per id and block of four query terms) and a byte map over the vocabulary,
// 17 B x V (6.8 MB at V = 400,001; 33 B x V for queries of five to eight terms) + 5 KB per block for its query; lists are processed in
// groups of as many as the workspace holds (<= 256).  Q <= 8 (two blocks of kQT = 4 terms: the reference's `maxqlen` is a free option,
// extractor/embedtext.py:28-31, and its forwards take any Q: reranker/KNRM.py:39-55, DRMM.py:101-116); other limits as the per-pair entries.
#include "lists.h"
#include "capamd_profiling.h"
#include <vector>

// capamd_debug_lists_timing: an event after every pass of a launch group while enabled.  Exists only in the -DCAPAMD_PROFILING build
// (libcapreolus_amd_prof.so, bench.py / scripts; not thread-safe): the product library carries no mutable global state.
#ifdef CAPAMD_PROFILING
namespace {
bool g_lists_timing = false;
std::vector<hipEvent_t> g_lists_events;
}  // namespace
void capamd::lists_stamp(hipStream_t s) {
  if (!g_lists_timing) return;
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return;
  (void)hipEventRecord(e, s);
  g_lists_events.push_back(e);
}
#endif
namespace {

// ---- 3a: KNRM pooling ----------------------------------------------------------------------------------------------------------
struct KnrmPoolArgs {
  const float* mu;
  const float* sigma;
  int K;
  const float* w1;
  const float* b1;
  int hidden;
  const float* w2;
  const float* b2;
  int scoretanh;
  float* out;
};

// A WAVE per document (four documents per workgroup), no LDS and no barrier: the row reductions are DPP, the per-kernel logs run in
// lanes (query term, kernel), the read-out takes the kernels' features by readlane.  A document is walked in passes of kWaveTrips trips:
// every id of the pass is requested first, then every table entry, then the arithmetic; a pass without a real term stops after the ids.
// (A workgroup per document - four waves sharing its positions, LDS reduction - measured 465 us per 64,000 documents against 352.)
// The lanes go to the query terms that are REAL: a term that is not (a pad of the reference's fixed-length query row, or an OOV term)
// has similarity exactly 0 with every table entry, so its row's sums are known without a lookup - sum_j K_k(0) over the document's real
// positions, K_k(0) = 2^-(B_k^2) exactly as the evaluation produces it for s = 0 - and its feature is 0 anyway unless it is an OOV term
// the document matches (KNRM.py:51-53: rows whose similarities sum to 0 are masked).  Per block of four query terms:
//     real terms 1: lane = position slot (64 per trip);  2: lanes 0-31 the first, 32-63 the second real term (32 per trip);
//     3-4: row t of the wave (lanes 16 t .. 16 t + 15) holds term t of 16 positions;  0 (an empty second block): nothing to walk.
// Queries of five to eight terms (QP = 2) walk the document once per block; the blocks' feature sums are added in block order, as the
// per-pair kernel adds its passes (knrm.hip).
// The sums add the same kernel values as the per-pair kernels in another order: equal to fp32 rounding of the sums (1e-6 relative).
// Measured and not kept (profiles/r04/lists_pool_variants.txt, r05/lists_pool_*_ab.txt): the frequent terms' entries staged in LDS
// (bank conflicts of 64 random ds_reads: 227 -> 281 us), eight documents per wave (261), a prefetch of the next pass's ids, 4 / 12 / 16
// trips per pass.
#ifndef CAPAMD_POOL_TRIPS
#define CAPAMD_POOL_TRIPS 6       // 8 until round 6: 163 -> 158 us (5 the same, 7: 163, 3-4: 163-167, 12: 190; profiles/r06/lists_pool_ablation.txt)
#endif
constexpr int kWaveTrips = CAPAMD_POOL_TRIPS;      // KNRM: 96 positions per pass (queries of three or four real terms; 192 / 384 with two / one)
constexpr int kTksTrips = 8;                       // DRMM-TKS's pooling kernel: 128 positions per pass

__device__ __forceinline__ float lane_bcast(float v, int src) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src)); }

// one document's positions in trips of W: acc[k] += K_k(s), rs += s for this lane's positions; table entries ESTRIDE floats apart
template <int KK, int W, int ESTRIDE>
__device__ __forceinline__ void knrm_pool_walk(const DocWalk& dw, int n, int lane, const float* tabsel, const float (&ka)[KK], const float (&kb)[KK],
                                               const f32x2 (&kbv)[KK / 2 + 1], float (&acc)[KK], float& rs) {
  const int ps = lane & (W - 1);
  for (int j0 = 0; j0 < n; j0 += W * kWaveTrips) {
    int id[kWaveTrips];
    load_pass<kWaveTrips, W>(dw, j0, ps, id);
    float s[kWaveTrips];
#pragma unroll
#ifdef CAPAMD_POOL_ABL_FOLD      // ablation: every lookup lands in the table's first 16 KB (what the pooling costs without its cache misses)
    for (int u = 0; u < kWaveTrips; ++u) s[u] = tabsel[(int64_t)(id[u] & 1023) * ESTRIDE];
#else
    for (int u = 0; u < kWaveTrips; ++u) s[u] = tabsel[(int64_t)id[u] * ESTRIDE];     // (entry 0 is never written and never used)
#endif
    // (pinned here: left alone, hipcc sinks each load into the branch that uses it, where it is issued and waited for one trip at a time)
#pragma unroll
    for (int u = 0; u < kWaveTrips; ++u) asm volatile("" : "+v"(s[u]));
#pragma unroll
    for (int u = 0; u < kWaveTrips; ++u) {
      if (j0 + u * W >= n) continue;          // (wave-uniform)
      if (id[u] > 0) {
        rs += s[u];
        // two kernels per packed instruction (v_pk_fma / v_pk_mul / v_pk_add: the same fma, product and sum per element, half the
        // issue slots - 2.5 instead of 4 VALU instructions per evaluation beside its v_exp_f32); an odd last kernel on its own
#pragma unroll
        for (int k = 0; k + 1 < KK; k += 2) {
          const f32x2 tk = f32x2{s[u], s[u]} * f32x2{ka[k], ka[k + 1]} + kbv[k / 2];
          const f32x2 nq = -tk * tk;
          const f32x2 e = {__builtin_amdgcn_exp2f(nq.x), __builtin_amdgcn_exp2f(nq.y)};
          f32x2 ac = {acc[k], acc[k + 1]};
          ac += e;
          acc[k] = ac.x; acc[k + 1] = ac.y;
        }
        if (KK & 1) {
          const float tk = __builtin_fmaf(s[u], ka[KK - 1], kb[KK - 1]);
          acc[KK - 1] += __builtin_amdgcn_exp2f(-tk * tk);
        }
      }
    }
  }
}

// KK: kernels the loops run over (11 - the model's default bank - or kMaxK, slots beyond K repeating the last kernel)
#ifndef CAPAMD_POOL_WAVES
#define CAPAMD_POOL_WAVES 1      // waves per SIMD the KNRM pooling kernel's register allocation aims at (1: whatever it needs - A/B builds)
#endif
template <int KK, int QP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(CAPAMD_POOL_WAVES, 8))) void lists_knrm_pool_kernel(ListsArgs a, ListGeom g, KnrmPoolArgs m) {
  int l, dq;
  if (!list_doc_of(a, l, dq)) return;       // (a.longest counts groups of 4 documents here)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, t = lane >> 4, k = lane & 15;
  const int doc = dq * 4 + wave;
  if (doc >= g.len[l]) return;
  float ka[KK], kb[KK];      // K_k(s) = 2^-(ka s + kb)^2
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) {
    ka[kk] = a.kn_consts[4 * kMaxK + kk];
    kb[kk] = a.kn_consts[5 * kMaxK + kk];
  }
  // (B as register PAIRS in VGPRs: uniform values live in SGPRs, a VALU instruction reads one SGPR operand - with A and B both there
  //  every packed fma is preceded by a v_mov_b64 of its B pair)
  f32x2 kbv[KK / 2 + 1];
#pragma unroll
  for (int kk = 0; kk + 1 < KK; kk += 2) {
    kbv[kk / 2] = f32x2{kb[kk], kb[kk + 1]};
    asm volatile("" : "+v"(kbv[kk / 2]));
  }
  const int b = g.start[l] + doc;
  // what the mark pass left: the document's real terms, dense (int32), and its counts
  const int32_t* dm = a.meta + (int64_t)b * kDocMeta;
  const DocWalk dw = doc_walk(a, b, dm);
  const int n = dw.n, n_real_doc = dm[0];
  const float* tabl = reinterpret_cast<const float*>(a.table + (int64_t)l * a.Vp * QP);
  float F = 0.f;
#pragma unroll
  for (int h = 0; h < QP; ++h) {
    // the block's real query terms (wave-uniform): how many, and the first two
    int nq = 0, r0 = 0, r1 = 0;
#pragma unroll
    for (int tt = 0; tt < kQT; ++tt) {
      const bool real = kQT * h + tt < a.Q && a.qmeta[(int64_t)l * QP + h].id[tt] > 0;
      r1 = (real && nq == 1) ? tt : r1;
      r0 = (real && nq == 0) ? tt : r0;
      nq += real ? 1 : 0;
    }
    nq = __builtin_amdgcn_readfirstlane(nq); r0 = __builtin_amdgcn_readfirstlane(r0); r1 = __builtin_amdgcn_readfirstlane(r1);
    const int n_one_t = doc_n_one(dm, kQT * h + t);
    const float* tab0 = tabl + 4 * h;
    float acc[KK], rs = 0.f;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) acc[kk] = 0.f;
    float S = 0.f, R0 = 0.f;
    if (nq == 1) {
      knrm_pool_walk<KK, 64, 4 * QP>(dw, n, lane, tab0 + r0, ka, kb, kbv, acc, rs);
      const bool mine = t == r0;
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        const float v = group_allreduce(acc[kk]);
        const float sum = (lane_bcast(v, 0) + lane_bcast(v, 16)) + (lane_bcast(v, 32) + lane_bcast(v, 48));
        S = (k == kk && mine) ? sum : S;
      }
      const float v = group_allreduce(rs);
      R0 = mine ? (lane_bcast(v, 0) + lane_bcast(v, 16)) + (lane_bcast(v, 32) + lane_bcast(v, 48)) : 0.f;
    } else if (nq == 2) {
      knrm_pool_walk<KK, 32, 4 * QP>(dw, n, lane, tab0 + (lane < 32 ? r0 : r1), ka, kb, kbv, acc, rs);
      const bool mine0 = t == r0, mine1 = t == r1;
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        const float v = group_allreduce(acc[kk]);
        const float s0 = lane_bcast(v, 0) + lane_bcast(v, 16), s1 = lane_bcast(v, 32) + lane_bcast(v, 48);
        S = (k == kk && mine0) ? s0 : (k == kk && mine1) ? s1 : S;
      }
      const float v = group_allreduce(rs);
      const float s0 = lane_bcast(v, 0) + lane_bcast(v, 16), s1 = lane_bcast(v, 32) + lane_bcast(v, 48);
      R0 = mine0 ? s0 : mine1 ? s1 : 0.f;
    } else if (nq > 2) {
      knrm_pool_walk<KK, 16, 4 * QP>(dw, n, lane, tab0 + t, ka, kb, kbv, acc, rs);
      // the 16 lanes of a row (one query term): every lane gets the row's sums; lane (t, k) keeps kernel k
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        const float v = group_allreduce(acc[kk]);
        S = k == kk ? v : S;
      }
      R0 = group_allreduce(rs);
    }
    const float k0c = (k < m.K) ? a.kn_consts[2 * kMaxK + k] : 0.f, k1c = (k < m.K) ? a.kn_consts[3 * kMaxK + k] : 0.f;
    if (nq == 0 || (nq == 1 && t != r0) || (nq == 2 && t != r0 && t != r1)) {
      // a row that was not evaluated: K_k(0) = 2^-(B_k^2) at every real position (what the evaluation adds for s = 0)
      const float bq = a.kn_consts[5 * kMaxK + (k < m.K ? k : 0)];
      S = (float)n_real_doc * __builtin_amdgcn_exp2f(-(bq * bq));
    }
    float f = 0.f;
    if (k < m.K && kQT * h + t < a.Q) {
      const int nz = a.L - n_real_doc - n_one_t;        // pads and OOV terms without a match: similarity 0 (KNRM.py:50 sums over ALL positions)
      S += (float)nz * k0c;
      S += (float)n_one_t * k1c;
      const float R = R0 + (float)n_one_t;
      f = R != 0.f ? logf(S + 1e-6f) : 0.f;   // KNRM.py:51-53
    }
    // over the block's query terms, in their order; over the blocks in theirs
    const float Fh = ((__shfl(f, k, 64) + __shfl(f, 16 + k, 64)) + __shfl(f, 32 + k, 64)) + __shfl(f, 48 + k, 64);
    F = h == 0 ? Fh : F + Fh;
  }
  // the read-out's weights requested only now: 14 registers fewer across the position loop
  const int hn = lane < m.hidden ? lane : 0;
  float w1v[kMaxK];
#pragma unroll
  for (int kk = 0; kk < kMaxK; ++kk) w1v[kk] = m.w1[hn * m.K + (kk < m.K ? kk : m.K - 1)];
  if (m.hidden > 0) {
    float h = m.b1[hn];
#pragma unroll
    for (int kk = 0; kk < kMaxK; ++kk)
      if (kk < m.K) h = __builtin_fmaf(w1v[kk], lane_bcast(F, kk), h);
    h = lane < m.hidden ? m.w2[hn] * tanhf(h) : 0.f;
    float sc = wave_allreduce_sum(h) + m.b2[0];
    if (m.scoretanh) sc = tanhf(sc);
    if (lane == 0) m.out[b] = sc;
  } else {
    float sc = m.b1[0];
#pragma unroll
    for (int kk = 0; kk < kMaxK; ++kk)
      if (kk < m.K) sc = __builtin_fmaf(w1v[kk], lane_bcast(F, kk), sc);
    if (m.scoretanh) sc = tanhf(sc);
    if (lane == 0) m.out[b] = sc;
  }
}

// ---- 3b: DRMM pooling ----------------------------------------------------------------------------------------------------------
struct DrmmPoolArgs {
  const float* idf;   // [B, Q] or the query table's [NQ, Q] in indexed mode
  const float* edges;
  int nbins, hist_type, gate_type, D;
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
};

// the softmax gate over the query's terms and the output layer (DRMM.py:97-98, :112-114), in the per-pair kernels' fixed order: term t's
// gate logit and net output live in lane 16 (t & 3) of register pair [t >> 2]
template <int QP>
__device__ __forceinline__ float drmm_gate_sum(const float (&gl)[QP], const float (&z)[QP], int Q) {
  float mx = lane_bcast(gl[0], 0);
#pragma unroll
  for (int t = 1; t < kQT * QP; ++t)
    if (t < Q) mx = fmaxf(mx, lane_bcast(gl[t >> 2], 16 * (t & 3)));
  float den = 0.f, num = 0.f;
#pragma unroll
  for (int t = 0; t < kQT * QP; ++t)
    if (t < Q) {
      const float e = expf(lane_bcast(gl[t >> 2], 16 * (t & 3)) - mx);
      den += e;
      num = __builtin_fmaf(e, lane_bcast(z[t >> 2], 16 * (t & 3)), num);
    }
  return num / den;
}

// The general form (more than 32 bins or 16 nodes; otherwise lists_drmm_pool_wave_kernel below): a workgroup per document, a lane per
// position: 256 consecutive positions per trip, kDrmmTrips trips per pass (ids first, then the 4-byte entries, then the counts).
// Counts go to one of 16 copies of the histograms (by lane & 15: at most 4 lanes of a wave meet on an address).
constexpr int kDrmmTrips = 4, kHistCopies = 16;

template <int QP>
__global__ __launch_bounds__(256) void lists_drmm_pool_kernel(ListsArgs a, ListGeom g, DrmmPoolArgs m) {
  __shared__ int hist[kQT][kMaxBins];
  __shared__ int hrep[kHistCopies][kQT * kMaxBins + 1];
  __shared__ float zs[kListMaxQ], gs[kListMaxQ];
  int l, doc;
  if (!list_doc_of(a, l, doc) || doc >= g.len[l]) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = g.start[l] + doc, NB = m.nbins + 1;
  const PairIds qids = pair_ids(a.ids, g.start[l], a.Q, a.L);
  const int32_t* dm = a.meta + (int64_t)b * kDocMeta;
  const DocWalk dw = doc_walk(a, b, dm);
  const int n = dw.n, n_oov = dm[1];
  int* myh = hrep[lane & (kHistCopies - 1)];
  const uint32_t* tab = reinterpret_cast<const uint32_t*>(a.table) + (int64_t)l * a.Vp * QP;
#pragma unroll 1
  for (int h = 0; h < QP; ++h) {
    for (int i = tid; i < kHistCopies * (kQT * kMaxBins + 1); i += 256) (&hrep[0][0])[i] = 0;
    __syncthreads();
    for (int j0 = 0; j0 < n; j0 += 256 * kDrmmTrips) {
      int id[kDrmmTrips];
      load_pass<kDrmmTrips, 256>(dw, j0, tid, id);
      uint32_t e[kDrmmTrips];
#pragma unroll
      for (int u = 0; u < kDrmmTrips; ++u) e[u] = tab[(int64_t)id[u] * QP + h];        // (entry 0 is never written and never used)
#pragma unroll
      for (int u = 0; u < kDrmmTrips; ++u) asm volatile("" : "+v"(e[u]));      // (pinned: see the KNRM pooling)
#pragma unroll
      for (int u = 0; u < kDrmmTrips; ++u) {
        if (id[u] > 0) {
#pragma unroll
          for (int q = 0; q < kQT; ++q) {
            if (kQT * h + q < a.Q) {
              const unsigned by = (e[u] >> (8 * q)) & 0xffu, bin = by & 0x7fu;
              if ((int)bin < m.nbins) atomicAdd(&myh[q * kMaxBins + bin], 1);
              if (by & kBinExact) atomicAdd(&myh[q * kMaxBins + m.nbins], 1);
            }
          }
        }
      }
    }
    if (tid == 0 && n_oov > 0) {        // an OOV document term: similarity exactly 0 (DRMM cannot take OOV query terms: no exact match to find)
      const int bz = list_bin_of(0.f, m.edges, m.nbins);
      if (bz < m.nbins)
        for (int q = 0; q < kQT && kQT * h + q < a.Q; ++q) atomicAdd(&hrep[0][q * kMaxBins + bz], n_oov);
    }
    __syncthreads();
    {
      int hs = 0;
#pragma unroll
      for (int c = 0; c < kHistCopies; ++c) hs += hrep[c][tid];
      hist[tid >> 6][tid & 63] = hs;
    }
    __syncthreads();
    // per query term: histogram transform + feed-forward net + gate logit - the tail of drmm.hip; a wave per query term, as the per-pair kernels
    const int q = kQT * h + wave;
    if (q < a.Q) {
      const int64_t qid = qids.q(q);
      if (lane == 0 && qid < 0) atomicOr(a.status, kErrQueryOOV);
      const int* hh = hist[wave];
      if (m.counts_out && lane < NB) m.counts_out[((int64_t)b * a.Q + q) * NB + lane] = hh[lane];
      float hv = lane < NB ? (float)(hh[lane] + 1) : 0.f;
      if (m.hist_type == 1) hv = hv / wave_allreduce_sum(hv);
      else if (m.hist_type == 2) hv = lane < NB ? logf(hv) : 0.f;
      const float b1v = lane < m.nodes ? m.b1[lane] : 0.f, w2v = lane < m.nodes ? m.w2[lane] : 0.f, b2v = m.b2[0];
      const float gate0 = m.gate_type == 0 ? m.gate_w[0] * m.idf[(int64_t)qids.qrow * a.Q + q] : 0.f;
      float acc = 0.f;
      for (int nn = 0; nn < m.nodes; ++nn) {
        const float wv = lane < NB ? m.w1[nn * NB + lane] : 0.f;
        const float sn = wave_allreduce_sum(wv * hv);
        if (lane == nn) acc = sn;
      }
      acc += b1v;
      const float o = wave_allreduce_sum(lane < m.nodes ? w2v * tanhf(acc) : 0.f) + b2v;
      float gl;
      if (m.gate_type == 0) {
        gl = gate0;
      } else {
        const float* e = m.emb_raw + (qid > 0 && qid < a.V ? qid : 0) * m.ld;
        float p = 0.f;
        for (int c = lane; c < m.D; c += 64) p = __builtin_fmaf(m.gate_w[c], e[c], p);
        gl = wave_allreduce_sum(p);
      }
      if (qid == 0) gl += -1e7f;
      if (lane == 0) { zs[q] = tanhf(o); gs[q] = gl; }
    }
    __syncthreads();
  }
  if (tid == 0) {
    float mx = gs[0];
    for (int t = 1; t < a.Q; ++t) mx = fmaxf(mx, gs[t]);
    float den = 0.f, num = 0.f;
    for (int t = 0; t < a.Q; ++t) {
      const float e = expf(gs[t] - mx);
      den += e;
      num = __builtin_fmaf(e, zs[t], num);
    }
    m.out[b] = __builtin_fmaf(m.out_w[0], num / den, m.out_b[0]);
  }
}

// Up to 32 bins and 16 nodes (the model's defaults: 30 and 5): a WAVE per document, four documents per workgroup, no barrier.  A lane
// is a position (64 per trip); the counts go to one of 8 copies of the wave's own histograms; then ONE wave does the block's four
// terms' tails: row q of the wave (lanes 16 q .. 16 q + 15) stands for the 64 lanes the per-pair tail gives term q - lane j for its lanes
// j and j + 16 - and reduces as that tail does, (row 0 + row 1) + (row 2 + row 3) with the rows beyond the bins / nodes all zero: the same
// sums in the same order, bit-identical scores.  (Workgroup per document with a wave per term for the tail: 277 us per 64,000
// documents, of which the tails 104 and the counting 144; with the single-wave tail 245.)  Queries of five to eight terms: the document
// is walked once per block of four terms, the gate runs over all of them at the end.
constexpr int kWaveCopies = 8, kWaveBins = 32, kWaveStride = kQT * kWaveBins + 1;

template <int QP>
__global__ __launch_bounds__(256) void lists_drmm_pool_wave_kernel(ListsArgs a, ListGeom g, DrmmPoolArgs m) {
  __shared__ int hrep[4][kWaveCopies * kWaveStride];
  int l, dq;
  if (!list_doc_of(a, l, dq)) return;       // (a.longest counts groups of 4 documents here)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int doc = dq * 4 + wave;
  if (doc >= g.len[l]) return;
  const uint32_t* tab = reinterpret_cast<const uint32_t*>(a.table) + (int64_t)l * a.Vp * QP;
  const int NB = m.nbins + 1;
  const PairIds qids = pair_ids(a.ids, g.start[l], a.Q, a.L);
  const int b = g.start[l] + doc;
  const int32_t* dm = a.meta + (int64_t)b * kDocMeta;
  const DocWalk dw = doc_walk(a, b, dm);
  const int n = dw.n, n_oov = dm[1];
  int* H = hrep[wave];
  int* myh = H + (lane & (kWaveCopies - 1)) * kWaveStride;
  float zv[QP], glv[QP];
#pragma unroll
  for (int h = 0; h < QP; ++h) {
    for (int i = lane; i < kWaveCopies * kWaveStride; i += 64) H[i] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // (one wave: LDS program order is the synchronisation)
    for (int j0 = 0; j0 < n; j0 += 64 * kDrmmTrips) {
      int id[kDrmmTrips];
      load_pass<kDrmmTrips, 64>(dw, j0, lane, id);
      uint32_t e[kDrmmTrips];
#pragma unroll
      for (int u = 0; u < kDrmmTrips; ++u) e[u] = tab[(int64_t)id[u] * QP + h];        // (entry 0 is never written and never used)
#pragma unroll
      for (int u = 0; u < kDrmmTrips; ++u) asm volatile("" : "+v"(e[u]));      // (pinned: see the KNRM pooling)
#pragma unroll
      for (int u = 0; u < kDrmmTrips; ++u) {
        if (id[u] > 0) {
#pragma unroll
          for (int q = 0; q < kQT; ++q) {
            if (kQT * h + q < a.Q) {
              const unsigned by = (e[u] >> (8 * q)) & 0xffu, bin = by & 0x7fu;
              if ((int)bin < m.nbins) atomicAdd(&myh[q * kWaveBins + bin], 1);
              if (by & kBinExact) atomicAdd(&myh[q * kWaveBins + m.nbins], 1);
            }
          }
        }
      }
    }
    if (lane == 0 && n_oov > 0) {       // an OOV document term: similarity exactly 0 (DRMM cannot take OOV query terms: no exact match to find)
      const int bz = list_bin_of(0.f, m.edges, m.nbins);
      if (bz < m.nbins)
        for (int q = 0; q < kQT && kQT * h + q < a.Q; ++q) atomicAdd(&H[q * kWaveBins + bz], n_oov);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    {
      const int q = lane >> 4, j = lane & 15, qq = kQT * h + q;
      const bool on = qq < a.Q;
      const int64_t qid = on ? qids.q(qq) : 0;
      if (on && j == 0 && qid < 0) atomicOr(a.status, kErrQueryOOV);
      const bool ina = j < NB, inb = j + 16 < NB;
      int ha = 0, hb = 0;
#pragma unroll
      for (int c = 0; c < kWaveCopies; ++c) {
        ha += H[c * kWaveStride + q * kWaveBins + j];
        hb += H[c * kWaveStride + q * kWaveBins + j + 16];
      }
      if (m.counts_out && on) {
        int32_t* co = m.counts_out + ((int64_t)b * a.Q + qq) * NB;
        if (ina) co[j] = ha;
        if (inb) co[j + 16] = hb;
      }
      float va = ina ? (float)(ha + 1) : 0.f, vb = inb ? (float)(hb + 1) : 0.f;
      if (m.hist_type == 1) {
        const float tot = group_allreduce(unfused(va)) + group_allreduce(unfused(vb));
        va = va / tot;
        vb = vb / tot;
      } else if (m.hist_type == 2) {
        va = ina ? logf(va) : 0.f;
        vb = inb ? logf(vb) : 0.f;
      }
      float acc = 0.f;
      for (int nn = 0; nn < m.nodes; ++nn) {
        const float wa = ina ? m.w1[nn * NB + j] : 0.f, wb = inb ? m.w1[nn * NB + j + 16] : 0.f;
        const float sn = group_allreduce(unfused(wa * va)) + group_allreduce(unfused(wb * vb));
        if (j == nn) acc = sn;
      }
      acc += j < m.nodes ? m.b1[j] : 0.f;
      const float o = group_allreduce(unfused(j < m.nodes ? m.w2[j] * tanhf(acc) : 0.f)) + m.b2[0];
      float gl;
      if (m.gate_type == 0) {
        gl = on ? m.gate_w[0] * m.idf[(int64_t)qids.qrow * a.Q + qq] : 0.f;
      } else {
        const float* e = m.emb_raw + (qid > 0 && qid < a.V ? qid : 0) * m.ld;
        float p[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          p[r] = 0.f;
          for (int c = j + 16 * r; c < m.D; c += 64) p[r] = __builtin_fmaf(m.gate_w[c], e[c], p[r]);
          p[r] = group_allreduce(unfused(p[r]));
        }
        gl = (p[0] + p[1]) + (p[2] + p[3]);
      }
      if (qid == 0) gl += -1e7f;
      zv[h] = tanhf(o);
      glv[h] = gl;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // (the wave's histograms are cleared for the next block)
  }
  const float agg = drmm_gate_sum<QP>(glv, zv, a.Q);
  if (lane == 0) m.out[b] = __builtin_fmaf(m.out_w[0], agg, m.out_b[0]);
}

// ---- 3c: DRMM-TKS pooling -------------------------------------------------------------------------------------------------------
//   The KNRM form of the sims pass (four floats per term and block), then per document the top-k
// similarities of every query term over ALL positions (reference DRMMTKS.py:55-56; pads and unmatched OOV terms contribute their 0, OOV
// exact matches their 1) -> Linear(k, 1) + tanh -> idf gate -> output layer, as drmmtks.hip.  A wave per document, a lane per (position
// slot, query term) as in the KNRM pooling's four-term form: every lane keeps a sorted top-KT of the similarities it meets (sorted_insert:
// one v_med3_f32 per element), the 16 lists of a row are merged in k rounds of a row-wide maximum over the list heads (LDS) and the
// closed-form candidates.  The values are selections of bit-identical similarities and enter the Linear in the same order: the scores
// equal capamd_drmmtks_forward's.
struct TksPoolArgs {
  const float* idf;      // [B, Q] or the query table's [NQ, Q] in indexed mode
  int topk;
  const float* gate_w;   // [1]
  const float* ffw_w;    // [topk]
  const float* ffw_b;    // [1]
  const float* out_w;
  const float* out_b;
  float* out;
};
constexpr int kMaxTopK = 16;

__device__ __forceinline__ float group_allreduce_max(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v));
  v = fmaxf(v, dpp_mov<0x4E>(v));
  v = fmaxf(v, dpp_mov<0x141>(v));
  v = fmaxf(v, dpp_mov<0x140>(v));
  return v;
}

template <int KT, int QP>
__global__ __launch_bounds__(256) void lists_tks_pool_kernel(ListsArgs a, ListGeom g, TksPoolArgs m) {
  __shared__ float heads[4][KT][64];        // [wave][list entry][lane]
  int l, dq;
  if (!list_doc_of(a, l, dq)) return;       // (a.longest counts groups of 4 documents here)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, t = lane >> 4, ps = lane & 15;
  const int doc = dq * 4 + wave;
  if (doc >= g.len[l]) return;
  const int K = m.topk;
  const PairIds qids = pair_ids(a.ids, g.start[l], a.Q, a.L);   // (the list's query: its first pair's row)
  const float ffw_l = lane < K ? m.ffw_w[lane] : 0.f;
  const float* tabl = reinterpret_cast<const float*>(a.table + (int64_t)l * a.Vp * QP);
  const int b = g.start[l] + doc;
  const int32_t* dm = a.meta + (int64_t)b * kDocMeta;
  const DocWalk dw = doc_walk(a, b, dm);
  const int n = dw.n, n_real_doc = dm[0];
  float zv[QP], glv[QP];
#pragma unroll
  for (int h = 0; h < QP; ++h) {
    const int tq = kQT * h + t;
    int64_t qid = tq < a.Q ? qids.q(tq) : 0;
    if (qid >= a.V) qid = 0;                // (flagged by the sims pass)
    const float gl0 = tq < a.Q ? m.gate_w[0] * m.idf[(int64_t)qids.qrow * a.Q + tq] : 0.f;
    const float* tab = tabl + 4 * h + t;
    const int n_one_t = doc_n_one(dm, tq);
    float top[KT];
#pragma unroll
    for (int i = 0; i < KT; ++i) top[i] = -INFINITY;
    for (int j0 = 0; j0 < n; j0 += 16 * kTksTrips) {
      int id[kTksTrips];
      load_pass<kTksTrips, 16>(dw, j0, ps, id);
      float s[kTksTrips];
#pragma unroll
#ifdef CAPAMD_POOL_ABL_FOLD      // ablation: every lookup lands in the table's first 16 KB (what the pooling costs without its cache misses)
      for (int u = 0; u < kTksTrips; ++u) s[u] = tab[(int64_t)(id[u] & 1023) * (4 * QP)];
#else
      for (int u = 0; u < kTksTrips; ++u) s[u] = tab[(int64_t)id[u] * (4 * QP)];     // (entry 0 is never written and never used)
#endif
      // (pinned here: left alone, hipcc sinks each load into the branch that uses it, where it is issued and waited for one trip at a time)
#pragma unroll
      for (int u = 0; u < kTksTrips; ++u) asm volatile("" : "+v"(s[u]));
#pragma unroll
      for (int u = 0; u < kTksTrips; ++u) {
        if (j0 + u * 16 >= n) continue;          // (wave-uniform)
        if (id[u] > 0) sorted_insert<KT>(top, s[u]);
      }
    }
#pragma unroll
    for (int i = 0; i < KT; ++i) heads[wave][i][lane] = top[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // (one wave: LDS program order is the synchronisation)
    const int no = n_one_t, nz = a.L - n_real_doc - no;         // nz: pads and OOV terms without a match: similarity 0
    int head = 0, used1 = 0, used0 = 0;
    float acc = m.ffw_b[0];
    for (int r = 0; r < K; ++r) {
      const float cand = head < KT ? heads[wave][head][lane] : -INFINITY;
      const float from_lists = group_allreduce_max(cand);
      const float c1 = used1 < no ? 1.f : -INFINITY, c0 = used0 < nz ? 0.f : -INFINITY;
      const float best = fmaxf(from_lists, fmaxf(c1, c0));
      const unsigned row = (unsigned)(__ballot(cand == best && best > -INFINITY) >> (16 * t)) & 0xffffu;
      if (row) {
        if (ps == __ffs((int)row) - 1) ++head;          // one list of the row advances
      } else if (c1 == best) {
        ++used1;
      } else {
        ++used0;
      }
      if (best > -INFINITY) acc = __builtin_fmaf(lane_bcast(ffw_l, r), best, acc);   // DRMMTKS.py:22: Linear(topk, 1) on the sorted values
    }
    float gl = gl0;
    if (qid == 0) gl += -1e7f;   // DRMMTKS.py:38
    zv[h] = tanhf(acc);
    glv[h] = gl;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // (the wave's list heads are rewritten by the next block)
  }
  const float agg = drmm_gate_sum<QP>(glv, zv, a.Q);
  if (lane == 0) m.out[b] = __builtin_fmaf(m.out_w[0], agg, m.out_b[0]);
}

}  // namespace

#ifdef CAPAMD_PROFILING
extern "C" void capamd_debug_lists_timing(int enable) {
  g_lists_timing = enable != 0;
  for (hipEvent_t e : g_lists_events) (void)hipEventDestroy(e);
  g_lists_events.clear();
}

// adds the durations of the passes (kListPasses of them per launch group) to ms[0 .. kListPasses), returns the launch groups seen
extern "C" int capamd_debug_lists_timing_read(double* ms) {
  constexpr int per = capamd::kListStamps;
  const size_t groups = g_lists_events.size() / per;
  for (size_t i = 0; i < groups; ++i) {
    (void)hipEventSynchronize(g_lists_events[per * i + per - 1]);
    for (int k = 0; k + 1 < per; ++k) {
      float t = 0.f;
      if (ms && hipEventElapsedTime(&t, g_lists_events[per * i + k], g_lists_events[per * i + k + 1]) == hipSuccess) ms[k] += t;
    }
  }
  for (hipEvent_t e : g_lists_events) (void)hipEventDestroy(e);
  g_lists_events.clear();
  return (int)groups;
}
#endif

extern "C" size_t capamd_lists_workspace_bytes_q(int n_lists, int64_t V, int64_t n_pairs, int L, int Q) {
  if (n_lists < 1 || V < 1 || n_pairs < 0 || L < 1 || Q < 1 || Q > kListMaxQ) return 0;
  const int n = n_lists < kListChunk ? n_lists : kListChunk;
  return lists_pair_bytes(n_pairs, L) + (size_t)n * lists_per_list_bytes(lists_vp(V), (Q + kQT - 1) / kQT) + kListConstBytes;
}

extern "C" size_t capamd_lists_workspace_bytes(int n_lists, int64_t V, int64_t n_pairs, int L) {
  return capamd_lists_workspace_bytes_q(n_lists, V, n_pairs, L, kQT);
}

extern "C" int capamd_knrm_forward_lists(const int64_t* q_ids, const int64_t* d_ids, const int32_t* q_table, const int32_t* d_table,
                                         const int32_t* pair_q, const int32_t* pair_d, const int64_t* list_offsets_host, int n_lists, int Q,
                                         int L, const float* packed, int64_t V, int D, const float* mu, const float* sigma, int K,
                                         const float* w1, const float* b1, int hidden, const float* w2, const float* b2, int scoretanh,
                                         float* out, int* status, void* workspace, size_t workspace_bytes, void* stream) {
  if (n_lists == 0) return CAPAMD_OK;
  const bool indexed = q_table != nullptr;
  if (indexed ? (!d_table || !pair_q || !pair_d) : (!q_ids || !d_ids)) return CAPAMD_ERR_ARG;
  if (!mu || !sigma || !w1 || !b1 || !out || K < 1 || K > kMaxK || hidden < 0 || hidden > kMaxHidden || (hidden > 0 && (!w2 || !b2))) return CAPAMD_ERR_ARG;
  const IdSource ids = indexed ? IdSource{nullptr, nullptr, q_table, d_table, pair_q, pair_d} : IdSource{q_ids, d_ids, nullptr, nullptr, nullptr, nullptr};
  const KnrmPoolArgs m{mu, sigma, K, w1, b1, hidden, w2, b2, scoretanh, out};
  hipStream_t s = (hipStream_t)stream;
  return lists_run(ids, list_offsets_host, n_lists, Q, L, packed, V, D, status, workspace, workspace_bytes, s, nullptr, 0, mu, sigma, K, true, nullptr, kListMaxQ,
                   [&](const ListsArgs& a, const ListGeom& g, int nl, int longest) {
                     ListsArgs aq = a;
                     aq.longest = (longest + 3) / 4;       // four documents per workgroup
                     const dim3 grid = list_doc_grid(nl, aq.longest);
                     if (a.QP == 1) {
                       if (K == 11) hipLaunchKernelGGL((lists_knrm_pool_kernel<11, 1>), grid, dim3(256), 0, s, aq, g, m);
                       else hipLaunchKernelGGL((lists_knrm_pool_kernel<kMaxK, 1>), grid, dim3(256), 0, s, aq, g, m);
                     } else {
                       if (K == 11) hipLaunchKernelGGL((lists_knrm_pool_kernel<11, 2>), grid, dim3(256), 0, s, aq, g, m);
                       else hipLaunchKernelGGL((lists_knrm_pool_kernel<kMaxK, 2>), grid, dim3(256), 0, s, aq, g, m);
                     }
                   });
}

extern "C" int capamd_drmm_forward_lists(const int64_t* q_ids, const int64_t* d_ids, const int32_t* q_table, const int32_t* d_table,
                                         const int32_t* pair_q, const int32_t* pair_d, const float* idf, const int64_t* list_offsets_host,
                                         int n_lists, int Q, int L, const float* packed, int64_t V, int D, const float* edges, int nbins,
                                         int hist_type, int gate_type, const float* gate_w, const float* emb_raw, int64_t ld, const float* w1,
                                         const float* b1, int nodes, const float* w2, const float* b2, const float* out_w, const float* out_b,
                                         float* out, int32_t* counts_out, int* status, void* workspace, size_t workspace_bytes, void* stream) {
  if (n_lists == 0) return CAPAMD_OK;
  const bool indexed = q_table != nullptr;
  if (indexed ? (!d_table || !pair_q || !pair_d) : (!q_ids || !d_ids)) return CAPAMD_ERR_ARG;
  if (!idf || !edges || !gate_w || !w1 || !b1 || !w2 || !b2 || !out_w || !out_b || !out) return CAPAMD_ERR_ARG;
  if (nbins < 1 || nbins + 1 > kMaxBins || nodes < 1 || nodes > kMaxNodes || hist_type < 0 || hist_type > 2 || gate_type < 0 || gate_type > 1) return CAPAMD_ERR_ARG;
  if (gate_type == 1 && (!emb_raw || ld < D)) return CAPAMD_ERR_ARG;
  const IdSource ids = indexed ? IdSource{nullptr, nullptr, q_table, d_table, pair_q, pair_d} : IdSource{q_ids, d_ids, nullptr, nullptr, nullptr, nullptr};
  const DrmmPoolArgs m{idf, edges, nbins, hist_type, gate_type, D, gate_w, emb_raw, ld, w1, b1, nodes, w2, b2, out_w, out_b, out, counts_out};
  hipStream_t s = (hipStream_t)stream;
  return lists_run(ids, list_offsets_host, n_lists, Q, L, packed, V, D, status, workspace, workspace_bytes, s, edges, nbins, nullptr, nullptr, 0, true,
                   gate_type == 0 ? idf : nullptr,      // (the term-vector gate never reads an idf row)
                   kListMaxQ,
                   [&](const ListsArgs& a, const ListGeom& g, int nl, int longest) {
                     if (nbins + 1 <= kWaveBins && nodes <= 16) {
                       ListsArgs aq = a;
                       aq.longest = (longest + 3) / 4;       // four documents per workgroup
                       if (a.QP == 1) hipLaunchKernelGGL(lists_drmm_pool_wave_kernel<1>, list_doc_grid(nl, aq.longest), dim3(256), 0, s, aq, g, m);
                       else hipLaunchKernelGGL(lists_drmm_pool_wave_kernel<2>, list_doc_grid(nl, aq.longest), dim3(256), 0, s, aq, g, m);
                     } else if (a.QP == 1) {
                       hipLaunchKernelGGL(lists_drmm_pool_kernel<1>, list_doc_grid(nl, longest), dim3(256), 0, s, a, g, m);
                     } else {
                       hipLaunchKernelGGL(lists_drmm_pool_kernel<2>, list_doc_grid(nl, longest), dim3(256), 0, s, a, g, m);
                     }
                   });
}

extern "C" int capamd_drmmtks_forward_lists(const int64_t* q_ids, const int64_t* d_ids, const int32_t* q_table, const int32_t* d_table,
                                            const int32_t* pair_q, const int32_t* pair_d, const float* idf, const int64_t* list_offsets_host,
                                            int n_lists, int Q, int L, const float* packed, int64_t V, int D, int topk, const float* gate_w,
                                            const float* ffw_w, const float* ffw_b, const float* out_w, const float* out_b, float* out, int* status,
                                            void* workspace, size_t workspace_bytes, void* stream) {
  if (n_lists == 0) return CAPAMD_OK;
  const bool indexed = q_table != nullptr;
  if (indexed ? (!d_table || !pair_q || !pair_d) : (!q_ids || !d_ids)) return CAPAMD_ERR_ARG;
  if (!idf || !gate_w || !ffw_w || !ffw_b || !out_w || !out_b || !out || topk < 1 || topk > kMaxTopK || topk > L) return CAPAMD_ERR_ARG;
  const IdSource ids = indexed ? IdSource{nullptr, nullptr, q_table, d_table, pair_q, pair_d} : IdSource{q_ids, d_ids, nullptr, nullptr, nullptr, nullptr};
  const TksPoolArgs m{idf, topk, gate_w, ffw_w, ffw_b, out_w, out_b, out};
  hipStream_t s = (hipStream_t)stream;
  return lists_run(ids, list_offsets_host, n_lists, Q, L, packed, V, D, status, workspace, workspace_bytes, s, nullptr, 0, nullptr, nullptr, 0, true, idf, kListMaxQ,
                   [&](const ListsArgs& a, const ListGeom& g, int nl, int longest) {
                     ListsArgs aq = a;
                     aq.longest = (longest + 3) / 4;       // four documents per workgroup
                     const dim3 grid = list_doc_grid(nl, aq.longest);
#define CAPAMD_TKS(QP_)                                                                                         \
  if (topk <= 4) hipLaunchKernelGGL((lists_tks_pool_kernel<4, QP_>), grid, dim3(256), 0, s, aq, g, m);          \
  else if (topk <= 8) hipLaunchKernelGGL((lists_tks_pool_kernel<8, QP_>), grid, dim3(256), 0, s, aq, g, m);     \
  else if (topk <= 12) hipLaunchKernelGGL((lists_tks_pool_kernel<12, QP_>), grid, dim3(256), 0, s, aq, g, m);   \
  else hipLaunchKernelGGL((lists_tks_pool_kernel<16, QP_>), grid, dim3(256), 0, s, aq, g, m)
                     if (a.QP == 1) { CAPAMD_TKS(1); } else { CAPAMD_TKS(2); }
#undef CAPAMD_TKS
                   });
}
