// The passes every whole-candidate-list entry shares (lists.hip: KNRM, DRMM, DRMM-TKS; pacrr.hip: PACRR): list geometry, the mark pass,
// the per-list query image, the sims pass (table of four similarities - or DRMM's four bins - per distinct term of a list and per BLOCK
// of four query terms: queries of up to kListMaxQ = 8 terms are two blocks) and the host loop over launch groups.  See lists.hip for the
// design.  Everything here is TU-local (anonymous namespace): each including file gets its own copy of the kernels.
// (The experiments of rounds 4-5 that lived here behind macros - the sims pass on v_mfma_f32_4x4x1, the dense head of the vocabulary on
//  v_mfma_f32_16x16x4, the query rows in registers, three rows per trip - were measured slower or equal, are described with their numbers
//  in DESIGN.md section 3.5 / docs/history.md, and were removed in round 6; git history has the code.)
#pragma once
#include "capreolus_amd.h"
#include "interaction.h"
#include <stdlib.h>
#include <type_traits>

namespace capamd {
constexpr int kListStamps = 6;        // stamps per launch group: before the clear and after each of clear, mark, query, sims, pool
#ifdef CAPAMD_PROFILING
void lists_stamp(hipStream_t s);      // lists.hip: records an event between passes while capamd_debug_lists_timing is on
#else
inline void lists_stamp(hipStream_t) {}
#endif
}

using namespace capamd;

namespace {

#ifndef CAPAMD_LIST_CHUNK
#define CAPAMD_LIST_CHUNK 256     // 64 until round 5: DRMM at configs[2] (250 lists per call) 129.8 -> 140.4 (128) -> 142.3 M pairs/s (256): fewer, longer passes, fewer tails
#endif
constexpr int kListChunk = CAPAMD_LIST_CHUNK;            // lists per launch group (their start / length travel as kernel arguments)
#ifndef CAPAMD_LISTS_SIMS_IDS
#define CAPAMD_LISTS_SIMS_IDS 1024
#endif
constexpr int kSimsIds = CAPAMD_LISTS_SIMS_IDS;             // vocabulary ids one workgroup of the sims pass scans
constexpr int kMaxK = 12, kMaxHidden = 64, kMaxBins = 64, kMaxNodes = 64;
constexpr int kListMaxQP = 2, kListMaxQ = kListMaxQP * kQT;      // blocks of kQT query terms a list query may have (reference `maxqlen` is a free option, embedtext.py:28-31)
constexpr float kLog2e = 1.4426950408889634f;

struct ListGeom {
  int start[kListChunk];   // first pair of the list
  int len[kListChunk];     // its documents
};

struct ListQuery;
struct ListsArgs {
  IdSource ids;
  int Q, L;
  const float* packed;
  int64_t V, Vp;           // Vp: V rounded up to kSimsIds (row stride of flags / table / idlist)
  uint8_t* flags;          // [lists][Vp]
  float4* table;           // [lists][Vp][QP] entry of a flagged id (written for flagged ids only) per block of four query terms: KNRM 4 floats,
                           // DRMM 4 bytes (at 4 B stride: [lists][Vp][QP] uint32 in the same memory)
  int* status;
  int nl, longest;         // lists in this launch group, documents of the longest
  const float* edges;      // DRMM: histogram bin edges
  int nbins;
  float4* qimg;            // [lists][QP][kQueryImage] the list's query rows as the sims pass wants them in LDS
  struct ListQuery* qmeta; // [lists][QP]
  const float* kn_mu;      // KNRM: the kernels' parameters (null for DRMM) ...
  const float* kn_sigma;
  int kn_K;
  int32_t* cid;            // [pairs][cid_stride] every document's REAL terms (0 < id < V), dense, in document order - written by the mark pass
  int32_t* meta;           // [pairs][kDocMeta] n_real, n_oov, n_one[0..3] (OOV terms equal to the list's OOV query term t), entries, -, n_one[4..7]
  int cid_stride;
  float* kn_consts;        // ... and what the pooling pass needs of them, computed once per call: [6][kMaxK] mu, c = -log2(e) / (2 sigma^2),
                           //     K(0) = 2^(c mu^2), K(1) = 2^(c (1 - mu)^2), A = sqrt(-c), B = -A mu   (slots beyond K repeat the last kernel)
  int preflag;             // ids below this count as flagged in every list (lists_clear_kernel): the mark pass does not store their bytes
  const float* list_idf;   // [B, Q] / [NQ, Q]: the idf rows of models that read the LIST's (its first pair's) row - DRMM, DRMM-TKS; else null
  int QP;                  // blocks of four query terms: (Q + 3) / 4
};

constexpr int kQueryImage = kQT * kMaxNV * 16;      // float4s
constexpr int kDocMeta = 12;                         // ints of per-document metadata the mark pass leaves for the pooling pass
__device__ __forceinline__ int doc_n_one(const int32_t* dm, int t) { return dm[t < kQT ? 2 + t : 4 + t]; }      // OOV terms of the document equal to the list's OOV query term t
// What the pooling pass reads of a document: CAPAMD_LISTS_COMPACT = 1 (default) its compact row (the mark pass appends every real term
// id, int32, to a per-pair workspace row: 4 L bytes per pair); 0 the document's own id row up to its LAST real position, which the mark
// pass leaves in the metadata - no compaction in the mark pass, no per-pair rows in the workspace.  Measured (profiles/r04/
// lists_compact_ab.txt): 0 takes 16 us off the mark pass and puts 15-30 us on every pooling kernel (8-byte ids, pads inside the row) -
// a wash for DRMM, a loss for KNRM and DRMM-TKS - so the compact rows stay.
#ifndef CAPAMD_LISTS_COMPACT
#define CAPAMD_LISTS_COMPACT 1
#endif
constexpr bool kCompactRows = CAPAMD_LISTS_COMPACT != 0;
struct ListQuery {
  int id[kQT];             // query term ids (0: pad or beyond Q)
  float den[kQT];          // their rows' norms
};

// workgroup -> (list, document) of the mark / pool launches.  With 8 or more lists, XCD x (workgroups whose linear index is x mod 8:
// blockIdx.x = 8 * document + x, the grid's x extent a multiple of 8) takes the lists 8 k + x, one at a time: what a list's documents
// share (flags; table) stays in that XCD's L2.
__device__ __forceinline__ bool list_doc_of(const ListsArgs& a, int& l, int& doc) {
  if (a.nl >= 8) {
    doc = blockIdx.x >> 3;
    l = blockIdx.y * 8 + (blockIdx.x & 7);
  } else {
    doc = blockIdx.x;
    l = blockIdx.y;
  }
  return l < a.nl;
}
dim3 list_doc_grid(int nl, int longest) { return nl >= 8 ? dim3((unsigned)longest * 8, (unsigned)(nl + 7) / 8) : dim3((unsigned)longest, (unsigned)nl); }

__device__ __forceinline__ int64_t doc_id_at(const PairIds& ids, int j) { return ids.d32 ? (int64_t)ids.d32[j] : ids.d64[j]; }

// ---- 0: clear ------------------------------------------------------------------------------------------------------------------
// The byte maps of the launch group's lists: zero, except the first `preflag` ids, which count as flagged in EVERY list - vocabularies
// are frequency-ordered (GloVe's is; the benchmark's ids are Zipf ranks), the first thousand ids are two thirds of all document
// positions and in (almost) every 1000-candidate list anyway, so the mark pass skips their byte stores (134 -> 1xx us) and the sims pass
// computes at most `preflag` rows per list that nobody looks up (+2-5 % of its rows on the benchmark's lists).  A flag is only ever a
// licence to compute an entry: a superset of the list's terms changes no result.  (Replaces the hipMemsetAsync of rounds 3-4.)
// How many: a term of rank r is in a list that holds a few times r positions; the call's positions per list (pads included) / 256, at
// most kPreflagMax - 3,120 on the benchmark's 1000 x 800 lists (mark 135 -> 104 us, sims +5), 300 on 100-candidate lists.
// (profiles/r04/lists_preflag_ab.txt)
#ifndef CAPAMD_LISTS_PREFLAG
#define CAPAMD_LISTS_PREFLAG 4096
#endif
constexpr int kPreflagMax = CAPAMD_LISTS_PREFLAG;
inline int lists_preflag(int64_t n_pairs, int L, int n_lists) {
  if (kPreflagMax <= 0 || n_lists < 1) return 0;
  const int64_t p = n_pairs * L / n_lists / 256;
  return (int)((p < kPreflagMax ? p : kPreflagMax) / 16 * 16);
}
__global__ __launch_bounds__(256) void lists_clear_kernel(ListsArgs a) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 16;          // (Vp is a multiple of kSimsIds, itself a multiple of 16)
  if (i >= a.Vp) return;
  const int64_t lim = a.V < a.preflag ? a.V : a.preflag;                      // (only rows of the table)
  unsigned w[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    w[k] = 0u;
#pragma unroll
    for (int b = 0; b < 4; ++b) w[k] |= (i + 4 * k + b < lim) ? (1u << (8 * b)) : 0u;
  }
  *reinterpret_cast<uint4*>(a.flags + (int64_t)blockIdx.y * a.Vp + i) = make_uint4(w[0], w[1], w[2], w[3]);
}

// ---- 1: mark -------------------------------------------------------------------------------------------------------------------
// A wave per document (four per workgroup): every real term id (0 < id < V) flags its byte of the list's map AND is appended to the
// document's compact id row - int32, pads / OOV terms dropped - which is what the pooling pass reads instead of the [L] int64 row (303
// real terms of 800 positions on the benchmark's lists: 78 MB written here, 330 MB not read there); the OOV terms are counted, those
// equal to one of the list's OOV query terms per term (their similarity is exactly 1, common.py:155-158; everything else without a row
// is exactly 0: the pooling pass adds both in closed form).  EMIT = false (PACRR: its kernel needs positions) only flags.
#ifndef CAPAMD_MARK_ABL
#define CAPAMD_MARK_ABL 0        // measurement builds: 1 = no flag stores, 2 = no compact-row stores
#endif
#ifndef CAPAMD_MARK_TRIPS
#define CAPAMD_MARK_TRIPS 13
#endif
constexpr int kMarkTrips = CAPAMD_MARK_TRIPS;     // 832 positions per pass, their ids requested together (the reference's documents are 800 positions)

template <bool EMIT>
__global__ __launch_bounds__(256) void lists_mark_kernel(ListsArgs a, ListGeom g) {
  int l, dq;
  if (!list_doc_of(a, l, dq)) return;       // (a.longest counts groups of four documents here)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int doc = dq * 4 + wave;
  if (doc >= g.len[l]) return;
  const int b = g.start[l] + doc;
  const PairIds ids = pair_ids(a.ids, b, a.Q, a.L);
  const PairIds qids = pair_ids(a.ids, g.start[l], a.Q, a.L);   // (the list's query: its first pair's row)
  int qo[kListMaxQ];          // the list's OOV query terms (0: not an OOV term)
  bool any_qo = false;
#pragma unroll
  for (int t = 0; t < kListMaxQ; ++t) {
    const int64_t q = t < a.Q ? qids.q(t) : 0;
    qo[t] = (q < 0 && q > -2147483648LL) ? (int)q : 0;
    any_qo |= qo[t] != 0;
  }
  // A list is ONE query against its documents: the passes below use the first pair's query row (and, where the model gates by idf, its
  // idf row) for all of them.  A pair that brings another row is a caller error - said through the status word instead of scoring the
  // pair against a query it does not have.  (Wave-uniform loads of 2 Q ids per document next to its L.)
  if (doc > 0 && ids.qrow != qids.qrow) {
    bool differs = false;
#pragma unroll
    for (int t = 0; t < kListMaxQ; ++t)
      if (t < a.Q) {
        differs |= ids.q(t) != qids.q(t);
        if (a.list_idf)
          differs |= __float_as_uint(a.list_idf[(int64_t)ids.qrow * a.Q + t]) != __float_as_uint(a.list_idf[(int64_t)qids.qrow * a.Q + t]);
      }
    if (differs && lane == 0) atomicOr(a.status, kErrListQuery);
  }
  uint8_t* f = a.flags + (int64_t)l * a.Vp;
  int32_t* out = (EMIT && kCompactRows) ? a.cid + (int64_t)b * a.cid_stride : nullptr;
  int n_real = 0, c_oov = 0, c_one[kListMaxQ] = {0, 0, 0, 0, 0, 0, 0, 0}, used = 0;      // used: positions up to the last one that is not a pad
  bool bad = false;
  for (int j0 = 0; j0 < a.L; j0 += 64 * kMarkTrips) {
    int64_t id[kMarkTrips];
    const bool full = j0 + 64 * kMarkTrips <= a.L;
    if (ids.d32) {
      int v[kMarkTrips];
      if (full) {
        const int* p = ids.d32 + j0 + lane;
#pragma unroll
        for (int u = 0; u < kMarkTrips; ++u) v[u] = p[u * 64];
      } else {
#pragma unroll
        for (int u = 0; u < kMarkTrips; ++u) {
          const int j = j0 + u * 64 + lane;
          v[u] = ids.d32[j < a.L ? j : a.L - 1];
        }
      }
#pragma unroll
      for (int u = 0; u < kMarkTrips; ++u) id[u] = (j0 + u * 64 + lane < a.L) ? (int64_t)v[u] : 0;
    } else if (full) {
      const int64_t* p = ids.d64 + j0 + lane;
#pragma unroll
      for (int u = 0; u < kMarkTrips; ++u) id[u] = __builtin_nontemporal_load(p + u * 64);
    } else {
#pragma unroll
      for (int u = 0; u < kMarkTrips; ++u) {
        const int j = j0 + u * 64 + lane;
        id[u] = __builtin_nontemporal_load(ids.d64 + (j < a.L ? j : a.L - 1));
      }
#pragma unroll
      for (int u = 0; u < kMarkTrips; ++u) id[u] = (j0 + u * 64 + lane < a.L) ? id[u] : 0;
    }
#pragma unroll
    for (int u = 0; u < kMarkTrips; ++u) {
      const int64_t v = id[u];
      if (!__any(v != 0)) continue;          // padding only (wave-uniform): the tail of most documents
      const bool real = v > 0 && v < a.V;
      if (v >= a.V) bad = true;
      // (unconditional: a check of the flag first puts a load in front of every store and measures the same.  Ids below a.preflag are
      //  flagged by lists_clear_kernel for every list - on frequency-ordered vocabularies two thirds of all positions' stores)
      if (!(CAPAMD_MARK_ABL & 1) && real && v >= a.preflag) f[v] = 1;
      if (EMIT) {
        const uint64_t set = __ballot(real);
        if (kCompactRows) {
          if (!(CAPAMD_MARK_ABL & 2) && real) out[n_real + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(set >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)set, 0u))] = (int)v;
        } else if (set) {
          used = j0 + u * 64 + 64 - __builtin_clzll(set);
        }
        n_real += __builtin_popcountll(set);
        const uint64_t neg = __ballot(v < 0);       // (counts as wave-uniform scalars: no cross-lane reduction at the end of the document)
        if (neg) {
          c_oov += __builtin_popcountll(neg);
          const int vi = (v < 0 && v > -2147483648LL) ? (int)v : 0;
          if (any_qo) {
#pragma unroll
            for (int t = 0; t < kListMaxQ; ++t)
              if (qo[t] != 0) c_one[t] += __builtin_popcountll(__ballot(vi == qo[t]));
          }
        }
      }
    }
  }
  if (bad) atomicOr(a.status, kErrDocIdRange);
  if (EMIT) {
    if (lane == 0) {
      int32_t* m = a.meta + (int64_t)b * kDocMeta;
      m[0] = n_real;
      m[1] = c_oov;
      m[6] = kCompactRows ? n_real : used;     // entries the pooling pass walks
#pragma unroll
      for (int t = 0; t < kListMaxQ; ++t)
        if (t < kQT || a.QP > 1) m[t < kQT ? 2 + t : 4 + t] = c_one[t];
    }
  }
}

// The ids of one pass of a pooling kernel over a document's OWN id row (kCompactRows = false) - TRIPS trips of STRIDE consecutive
// positions below `used`, this lane's slot `ps` - as int: > 0 a real term of the table, 0 anything else (pads, OOV terms and ids beyond the
// table: the mark pass counted them).
template <int TRIPS, int STRIDE>
__device__ __forceinline__ void load_pass_rows(const PairIds& ids, int j0, int ps, int used, int64_t V, int (&id)[TRIPS]) {
  const bool full = j0 + TRIPS * STRIDE <= used;
  if (ids.d32) {
    if (full) {
      const int* p = ids.d32 + j0 + ps;
#pragma unroll
      for (int u = 0; u < TRIPS; ++u) id[u] = p[u * STRIDE];
    } else {
#pragma unroll
      for (int u = 0; u < TRIPS; ++u) {
        const int j = j0 + u * STRIDE + ps;
        id[u] = ids.d32[j < used ? j : 0];
      }
#pragma unroll
      for (int u = 0; u < TRIPS; ++u) id[u] = (j0 + u * STRIDE + ps < used) ? id[u] : 0;
    }
#pragma unroll
    for (int u = 0; u < TRIPS; ++u) id[u] = (id[u] > 0 && id[u] < V) ? id[u] : 0;
  } else {
    int64_t w[TRIPS];
    if (full) {
      const int64_t* p = ids.d64 + j0 + ps;
#pragma unroll
      for (int u = 0; u < TRIPS; ++u) w[u] = __builtin_nontemporal_load(p + u * STRIDE);
    } else {
#pragma unroll
      for (int u = 0; u < TRIPS; ++u) {
        const int j = j0 + u * STRIDE + ps;
        w[u] = __builtin_nontemporal_load(ids.d64 + (j < used ? j : 0));
      }
#pragma unroll
      for (int u = 0; u < TRIPS; ++u) w[u] = (j0 + u * STRIDE + ps < used) ? w[u] : 0;
    }
#pragma unroll
    for (int u = 0; u < TRIPS; ++u) id[u] = (w[u] > 0 && w[u] < V) ? (int)w[u] : 0;
  }
}

// the compact ids of one pass of a pooling kernel - TRIPS trips of STRIDE consecutive entries, this lane's slot `ps`; 0 beyond the row's n
#ifndef CAPAMD_POOL_CID_NT
#define CAPAMD_POOL_CID_NT 1      // the compact id rows are read once: nontemporal loads (0: plain loads - A/B builds)
#endif
__device__ __forceinline__ int cid_load(const int32_t* p) { return CAPAMD_POOL_CID_NT ? __builtin_nontemporal_load(p) : *p; }
template <int TRIPS, int STRIDE>
__device__ __forceinline__ void load_pass_cids(const int32_t* row, int j0, int ps, int n, int (&id)[TRIPS]) {
  if (j0 + TRIPS * STRIDE <= n) {
    const int32_t* p = row + j0 + ps;
#pragma unroll
    for (int u = 0; u < TRIPS; ++u) id[u] = cid_load(p + u * STRIDE);
  } else {
#pragma unroll
    for (int u = 0; u < TRIPS; ++u) {
      const int j = j0 + u * STRIDE + ps;
      id[u] = cid_load(row + (j < n ? j : 0));
    }
#pragma unroll
    for (int u = 0; u < TRIPS; ++u) id[u] = (j0 + u * STRIDE + ps < n) ? id[u] : 0;
  }
}

// what a pooling kernel walks of document b: (entries, loader of a pass)
struct DocWalk {
  PairIds ids;
  const int32_t* row;
  int n;          // entries to walk: the compact row's length, or the positions up to the document's last real one
  int64_t V;
};
__device__ __forceinline__ DocWalk doc_walk(const ListsArgs& a, int b, const int32_t* dm) {
  DocWalk w;
  w.ids = pair_ids(a.ids, b, a.Q, a.L);
#ifdef CAPAMD_POOL_ABL_HOTIDS      // ablation: every document reads one of eight id rows (what the pooling costs when its id rows are cache-hot)
  w.row = kCompactRows ? a.cid + (int64_t)(b & 7) * a.cid_stride : nullptr;
#else
  w.row = kCompactRows ? a.cid + (int64_t)b * a.cid_stride : nullptr;
#endif
  w.n = dm[6];
  w.V = a.V;
  return w;
}
template <int TRIPS, int STRIDE>
__device__ __forceinline__ void load_pass(const DocWalk& w, int j0, int ps, int (&id)[TRIPS]) {
#ifdef CAPAMD_POOL_ABL_IDS      // ablation: the pass's ids without a load (what the pooling costs without its id round trips)
#pragma unroll
  for (int u = 0; u < TRIPS; ++u) id[u] = (j0 + u * STRIDE + ps < w.n) ? 1 + ((j0 + u * STRIDE + ps) * 37 + (int)(reinterpret_cast<uintptr_t>(w.row) >> 6)) % 40000 : 0;
  return;
#endif
  if (kCompactRows) load_pass_cids<TRIPS, STRIDE>(w.row, j0, ps, w.n, id);
  else load_pass_rows<TRIPS, STRIDE>(w.ids, j0, ps, w.n, w.V, id);
}

// ---- 2: sims -------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int list_bin_of(float x, const float* edges, int nbins) {   // as drmm.hip: exactly the reference's `x < edge`
  int bi = (int)floorf((x + 1.f) * (0.5f * (float)nbins));
  bi = bi < 0 ? 0 : (bi > nbins ? nbins : bi);
  const float e_lo = edges[bi > 0 ? bi - 1 : 0], e_hi = edges[bi < nbins ? bi : nbins - 1];
  if (bi > 0 && x < e_lo) {
    --bi;
    while (bi > 0 && x < edges[bi - 1]) --bi;
  } else if (bi < nbins && !(x < e_hi)) {
    ++bi;
    while (bi < nbins && !(x < edges[bi])) ++bi;
  }
  return bi;
}
constexpr unsigned kBinExact = 0x80;      // entry byte: bin (nbins = above the last edge) | kBinExact when 0.999 < s < 1.001 (drmm.hip's exact-match bin)

// the query of every list once, per block of four terms (blockIdx.y): its packed rows in the PAIRED LDS layout of rows_dot_pk, its ids
// and norms; and (list 0's first workgroup) the KNRM kernel constants of the call
template <int NV>
__global__ __launch_bounds__(128) void lists_query_kernel(ListsArgs a, ListGeom g) {
  __shared__ __attribute__((aligned(16))) float4 qlds[kQueryImage];
  const int l = blockIdx.x, h = blockIdx.y, tid = threadIdx.x, lane16 = tid & 15;
  const PairIds ids = pair_ids(a.ids, g.start[l], a.Q, a.L);
  QueryPass<NV> qp;
  load_query_pass_lds<NV, true>(a.packed, ids, a.Q, kQT * h, a.V, tid, 128, lane16, qlds, qp, a.status);
  __syncthreads();
  float4* img = a.qimg + ((int64_t)l * a.QP + h) * kQueryImage;
  for (int i = tid; i < kQT * NV * 16; i += 128) img[i] = qlds[i];
  if (tid < kQT) {           // lane16 = tid: the term this lane "owns" in QueryPass
    a.qmeta[(int64_t)l * a.QP + h].id[tid] = qp.id_my;
    a.qmeta[(int64_t)l * a.QP + h].den[tid] = qp.den_my;
  }
  if (l == 0 && h == 0 && a.kn_consts && tid < kMaxK) {
    const int kc = tid < a.kn_K ? tid : a.kn_K - 1;
    const float sg = a.kn_sigma[kc], mk = a.kn_mu[kc], c = (-0.5f * kLog2e) / (sg * sg);
    a.kn_consts[tid] = mk;
    a.kn_consts[kMaxK + tid] = c;
    a.kn_consts[2 * kMaxK + tid] = __builtin_amdgcn_exp2f(mk * mk * c);
    a.kn_consts[3 * kMaxK + tid] = __builtin_amdgcn_exp2f((1.f - mk) * (1.f - mk) * c);
    // the evaluation as K(s) = 2^-(A s + B)^2 with A = sqrt(-c), B = -A mu: fma, mul, exp, add - one VALU instruction fewer per value than
    // (s - mu)^2 c (scripts/ubench/valu_rates.hip: 8.7 against 7.2 T evaluations/s)
    const float A = sqrtf(-c);
    a.kn_consts[4 * kMaxK + tid] = A;
    a.kn_consts[5 * kMaxK + tid] = -A * mk;
  }
}

#ifndef CAPAMD_LISTS_SIMS_HALF
#define CAPAMD_LISTS_SIMS_HALF 1      // 0: every block's four dot products per row (A/B builds)
#endif
#ifndef CAPAMD_LISTS_SIMS_WAVES
#define CAPAMD_LISTS_SIMS_WAVES 1
#endif
// A workgroup per (list, block of kSimsIds vocabulary ids): the block's flagged ids compacted into LDS (per flag byte one ballot, a
// lane's slot = the set lanes below it), then a 16-lane group per row, two rows per group and trip: five global_load_dwordx4, the dot
// products against the list's query rows in LDS (read once for both rows), DPP reduce, divide - rows_dot2_pk / sim_from_dots: per term
// and row exactly the fma chain, the reduction tree and the divide of the per-pair kernels - bit-identical similarities.  QP = 2
// (queries of five to eight terms): the same two rows against the second block's image too, their entries side by side in the table.
// XCD x (workgroups whose linear index is x mod 8: blockIdx.x = 8 * list + x) takes the id blocks 8 k + x, each for all lists back to
// back: the lists share most of a block's rows, so the rows come from that XCD's L2.
// What bounds it: the L2 -> CU gather of the rows (13.9 TB/s; profiles/r06/lists_sims_steps.txt has the round-6 rebuilds around
// precomputed work lists and persistent workgroups, all slower).
#ifndef CAPAMD_LISTS_SIMS_BALLAST
#define CAPAMD_LISTS_SIMS_BALLAST 0     // KiB of dynamic LDS a workgroup of the sims pass is launched with and never uses: caps the pass's workgroups per
                                        // CU, leaving wave slots and registers to the other step stream's kernels (A/B builds)
#endif
template <int NV, bool BINS, int QP>
__global__ __launch_bounds__(256, CAPAMD_LISTS_SIMS_WAVES) void lists_sims_kernel(ListsArgs a, ListGeom g) {
  __shared__ __attribute__((aligned(16))) float4 qlds[QP][kQT * kMaxNV * 16];
  __shared__ int lst[kSimsIds];
  __shared__ int wave_cnt[4];
  __shared__ float edges[kMaxBins];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lane16 = tid & 15, grp = tid >> 4;
  const int l = blockIdx.x >> 3, blk = blockIdx.y * 8 + (blockIdx.x & 7);
  if ((int64_t)blk * kSimsIds >= a.Vp) return;
  const int id0 = blk * kSimsIds;
  constexpr int kPer = kSimsIds / 256;      // ids per thread: their flag bytes in one load
  static_assert(kPer == 2 || kPer == 4 || kPer == 8, "kSimsIds is 512, 1024 or 2048");
  const uint8_t* fp = a.flags + (int64_t)l * a.Vp + id0 + tid * kPer;
  const uint64_t fw = kPer == 2 ? (uint64_t)*reinterpret_cast<const uint16_t*>(fp) : kPer == 4 ? (uint64_t)*reinterpret_cast<const uint32_t*>(fp)
                                                                                               : *reinterpret_cast<const uint64_t*>(fp);
  // the list's query rows: the LDS image lists_query_kernel left (built here, from the ids, it is five dependent loads per workgroup)
  QueryPass<NV> qp[QP];
#pragma unroll
  for (int h = 0; h < QP; ++h) {
    const float4* img = a.qimg + ((int64_t)l * QP + h) * kQueryImage;
    for (int i = tid; i < kQT * NV * 16; i += 256) qlds[h][i] = img[i];
    qp[h].den_my = a.qmeta[(int64_t)l * QP + h].den[lane16 & 3];
    qp[h].id_my = a.qmeta[(int64_t)l * QP + h].id[lane16 & 3];
  }
  if (BINS && tid < a.nbins) edges[tid] = a.edges[tid];
  // the flagged ids, dense, in LDS (any order): per flag byte one ballot, the lane's slot = the set lanes below it
  int slot[kPer], mine = 0;
#pragma unroll
  for (int c = 0; c < kPer; ++c) {
    const uint64_t set = __ballot(((fw >> (8 * c)) & 0xffu) != 0);
    slot[c] = mine + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(set >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)set, 0u));
    mine += __builtin_popcountll(set);       // (wave-uniform from here: the wave's count so far)
  }
  if (lane == 0) wave_cnt[wave] = mine;
  __syncthreads();
  const int c0 = wave_cnt[0], c1 = wave_cnt[1], c2 = wave_cnt[2], c3 = wave_cnt[3];
  const int total = __builtin_amdgcn_readfirstlane(c0 + c1 + c2 + c3);      // (workgroup-uniform: an SGPR)
  if (total == 0) return;
  const int base = wave == 0 ? 0 : wave == 1 ? c0 : wave == 2 ? c0 + c1 : c0 + c1 + c2;
#pragma unroll
  for (int c = 0; c < kPer; ++c)
    if ((fw >> (8 * c)) & 0xffu) lst[base + slot[c]] = tid * kPer + c;
  __syncthreads();
  float* tab = reinterpret_cast<float*>(a.table + (int64_t)l * a.Vp * QP);
  uint8_t* tabb = reinterpret_cast<uint8_t*>(reinterpret_cast<uint32_t*>(a.table) + (int64_t)l * a.Vp * QP);
  auto put = [&](int id, int h, float sm) {        // lane t < 4 of the group: the similarity of query term 4 h + t
    if (lane16 < kQT) {
      if (BINS) {
        const unsigned bin = (unsigned)list_bin_of(sm, edges, a.nbins) | ((sm > 0.999f && sm < 1.001f) ? kBinExact : 0u);
        tabb[((int64_t)id * QP + h) * 4 + lane16] = (uint8_t)bin;
      } else {
        tab[((int64_t)id * QP + h) * 4 + lane16] = sm;
      }
    }
  };
  // A block whose query terms 2 and 3 are not real (a query of one or two terms in the reference's fixed-length row: half of the
  // benchmark's queries; the second block of a six-term query) needs the first pair's dot products only: the other two similarities are 0
  // by definition (sim_from_dots) - half the fmas, LDS query reads and row reductions; the same table entries, bit for bit.
  bool half[QP];
#pragma unroll
  for (int h = 0; h < QP; ++h) half[h] = CAPAMD_LISTS_SIMS_HALF && a.qmeta[(int64_t)l * QP + h].id[2] <= 0 && a.qmeta[(int64_t)l * QP + h].id[3] <= 0;
  auto trips = [&](auto H0, auto H1) {             // H0 / H1: std::integral_constant<bool, ...> - block 0 / 1 on the two-term form
#pragma clang loop unroll(disable)
    for (int e = grp; e < total; e += 2 * kGroupsPerWG) {
      // (an odd last row is done twice: a branch here makes hipcc sink row b's fma chain into it and keep the whole query copy in registers)
      const int ida = id0 + lst[e], idb = id0 + lst[e + kGroupsPerWG < total ? e + kGroupsPerWG : e];
      RowRegs<NV> da, db;
#ifdef CAPAMD_LISTS_ABL_HOTROWS      // ablation: every row load hits one of 16 rows (what the pass costs without its gather)
      load_row<NV>(a.packed, 1 + (ida & 15), lane16, da);
      load_row<NV>(a.packed, 1 + (idb & 15), lane16, db);
#else
      load_row<NV>(a.packed, ida, lane16, da);
      load_row<NV>(a.packed, idb, lane16, db);
#endif
      const float dena = row_den<NV>(da), denb = row_den<NV>(db);
#pragma unroll
      for (int h = 0; h < QP; ++h) {
        constexpr bool kH0 = decltype(H0)::value, kH1 = decltype(H1)::value;
        float pa[kQT], pb[kQT];
        int qoff = 0;
        asm volatile("" : "+v"(qoff));
        if (h > 0) {
          // The rows are handed to the second block's pass as NEW values, as whole 128-bit tuples: hipcc otherwise keeps the {x, x} operand
          // pairs it built for the first pass's packed fmas alive for the second (170 registers, three waves per SIMD; laundered float by
          // float every value is copied to an even register for the operand pair: 132) - 86 registers and five waves this way (round 6:
          // queries of eight terms 68 -> 90 M pairs/s)
#pragma unroll
          for (int i = 0; i < NV; ++i) {
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            f32x4 ta = {da.v[i].x, da.v[i].y, da.v[i].z, da.v[i].w}, tb = {db.v[i].x, db.v[i].y, db.v[i].z, db.v[i].w};
            asm volatile("" : "+v"(ta), "+v"(tb));
            da.v[i] = make_float4(ta.x, ta.y, ta.z, ta.w);
            db.v[i] = make_float4(tb.x, tb.y, tb.z, tb.w);
          }
        }
        if ((h == 0 && kH0) || (h == 1 && kH1)) {
          rows_dot2_pk<NV, 1>(da, db, &qlds[h][0] + qoff, lane16, pa, pb);
          put(ida, h, sim_from_dots<NV, 2>(pa, dena, qp[h], lane16));
          put(idb, h, sim_from_dots<NV, 2>(pb, denb, qp[h], lane16));
        } else {
          rows_dot2_pk<NV>(da, db, &qlds[h][0] + qoff, lane16, pa, pb);
          put(ida, h, sim_from_dots<NV>(pa, dena, qp[h], lane16));
          put(idb, h, sim_from_dots<NV>(pb, denb, qp[h], lane16));
        }
      }
    }
  };
  using T = std::true_type;
  using F = std::false_type;
  if (QP == 1) {
    if (half[0]) trips(T{}, F{});
    else trips(F{}, F{});
  } else {
    // (the extractor fills the query row from the left: a short query under a long `maxqlen` has a short block 0 and an empty block 1)
    if (half[0] && half[QP - 1]) trips(T{}, T{});
    else if (half[QP - 1]) trips(F{}, T{});
    else if (half[0]) trips(T{}, F{});
    else trips(F{}, F{});
  }
  // (measured and not kept, round 6 - profiles/r06/lists_sims_{pairs,lead,ballast}_ab.txt: TWO lists per workgroup, a row both flag loaded once
  //  (18 % fewer rows through the L1s: the pass 2 % shorter, its fabric reads doubled - twice as many id blocks in flight per XCD); the first
  //  lists running 2-24 grid rows ahead so that shared rows' first touches are made early (equal); the pass capped at 4 / 3 / 2 workgroups per
  //  CU under two step streams so that the other stream's passes co-reside (the cap costs what it costs alone))
  // (measured and not kept, rounds 3-5: one row per trip with the NEXT row requested before the current one is used - a software pipeline -
  //  4-5 % slower end to end; FOUR rows per trip 10 % slower; three rows per trip equal; 512 / 2048 ids per workgroup slower; the query
  //  rows in registers 30-70 % slower; the dot products on v_mfma_f32_4x4x1 through an LDS turn 40-60 % slower)
}

// ---- host side -----------------------------------------------------------------------------------------------------------------
constexpr size_t kListQueryBytes = kQueryImage * sizeof(float4) + sizeof(ListQuery);   // per block of four query terms: the image + ids / norms
constexpr size_t kListConstBytes = 6 * kMaxK * sizeof(float);                          // per call: the KNRM kernel constants
int64_t lists_vp(int64_t V) { return (V + kSimsIds - 1) / kSimsIds * kSimsIds; }
int lists_cid_stride(int L) { return (L + 3) & ~3; }
size_t lists_pair_bytes(int64_t n_pairs, int L) { return (size_t)n_pairs * ((kCompactRows ? (size_t)lists_cid_stride(L) * 4 : 0) + kDocMeta * 4); }
// per list in flight: table (16 B per id and block of four query terms), byte map (1 B per id), the query images
size_t lists_per_list_bytes(int64_t Vp, int QP) { return (size_t)Vp * (16 * QP + 1) + (size_t)QP * kListQueryBytes; }

// runs `pool(geometry, lists in the chunk, longest list)` for chunks of lists that fit the workspace, after marking and the sims pass
template <class Pool>
int lists_run(const IdSource& ids, const int64_t* offsets_host, int n_lists, int Q, int L, const float* packed, int64_t V, int D, int* status,
              void* workspace, size_t workspace_bytes, hipStream_t s, const float* edges, int nbins, const float* kn_mu, const float* kn_sigma, int kn_K,
              bool emit, const float* list_idf, int max_q, Pool pool) {
  if (!offsets_host || !packed || !status || !workspace) return CAPAMD_ERR_ARG;
  if (n_lists < 0 || Q < 1 || Q > max_q || Q > kListMaxQ || L < 1 || L > 32768 || V < 1 || V > 0x7fffffffLL || capamd_packed_row_stride(D) < 0) return CAPAMD_ERR_ARG;
  if ((reinterpret_cast<uintptr_t>(workspace) & 15) != 0) return CAPAMD_ERR_ALIGN;
  const int64_t Vp = lists_vp(V);
  const int QP = (Q + kQT - 1) / kQT;
  const size_t per_list = lists_per_list_bytes(Vp, QP);
  // workspace: [compact id rows + document metadata of ALL the call's pairs] [per list in flight: table, byte map, query images] [KNRM constants]
  const int64_t n_pairs = n_lists > 0 ? offsets_host[n_lists] : 0;
  if (n_pairs < 0 || n_pairs > 0x7fffffffLL) return CAPAMD_ERR_ARG;
  const size_t pair_bytes = emit ? lists_pair_bytes(n_pairs, L) : 0;
  if (workspace_bytes < kListConstBytes + pair_bytes) return CAPAMD_ERR_WORKSPACE;
  const size_t fit = (workspace_bytes - kListConstBytes - pair_bytes) / per_list;
  int cap = (int)(fit < (size_t)kListChunk ? fit : (size_t)kListChunk);
  if (cap < 1) return CAPAMD_ERR_WORKSPACE;
  for (int l = 0; l < n_lists; ++l)
    if (offsets_host[l + 1] < offsets_host[l] || offsets_host[l + 1] > 0x7fffffffLL) return CAPAMD_ERR_ARG;
  (void)hipGetLastError();
  for (int l0 = 0; l0 < n_lists; l0 += cap) {
    const int nl = n_lists - l0 < cap ? n_lists - l0 : cap;
    ListGeom g{};
    int longest = 0;
    for (int i = 0; i < nl; ++i) {
      g.start[i] = (int)offsets_host[l0 + i];
      g.len[i] = (int)(offsets_host[l0 + i + 1] - offsets_host[l0 + i]);
      if (g.len[i] > longest) longest = g.len[i];
    }
    if (longest == 0) continue;
    // per-list part: table [cap][Vp][QP] x 16 B | byte maps [cap][Vp] | query images [cap][QP] | query ids and norms [cap][QP] | KNRM kernel constants
    char* ws = static_cast<char*>(workspace) + pair_bytes;
    int32_t* cid = emit ? reinterpret_cast<int32_t*>(workspace) : nullptr;
    int32_t* meta = emit ? cid + (kCompactRows ? (size_t)n_pairs * lists_cid_stride(L) : 0) : nullptr;
    float4* table = reinterpret_cast<float4*>(ws);
    uint8_t* flags = reinterpret_cast<uint8_t*>(ws + (size_t)cap * Vp * 16 * QP);
    float4* qimg = reinterpret_cast<float4*>(ws + (size_t)cap * Vp * (16 * QP + 1));
    ListQuery* qmeta = reinterpret_cast<ListQuery*>(qimg + (size_t)cap * QP * kQueryImage);
    float* kn_consts = reinterpret_cast<float*>(qmeta + (size_t)cap * QP);
    if (!kn_mu) kn_consts = nullptr;
    ListsArgs a{ids, Q, L, packed, V, Vp, flags, table, status, nl, longest, edges, nbins, qimg, qmeta, kn_mu, kn_sigma, kn_K, cid, meta, lists_cid_stride(L),
                kn_consts, lists_preflag(n_pairs, L, n_lists), list_idf, QP};
    lists_stamp(s);
    hipLaunchKernelGGL(lists_clear_kernel, dim3((unsigned)((Vp + 256 * 16 - 1) / (256 * 16)), (unsigned)nl), dim3(256), 0, s, a);
    lists_stamp(s);
    {
      ListsArgs am = a;
      am.longest = (longest + 3) / 4;       // four documents per workgroup
      if (emit) hipLaunchKernelGGL(lists_mark_kernel<true>, list_doc_grid(nl, am.longest), dim3(256), 0, s, am, g);
      else hipLaunchKernelGGL(lists_mark_kernel<false>, list_doc_grid(nl, am.longest), dim3(256), 0, s, am, g);
    }
    lists_stamp(s);
    const dim3 sg((unsigned)nl * 8, (unsigned)((Vp / kSimsIds + 7) / 8));
#define CAPAMD_SIMS_Q(NV, QP_)                                                                                  \
  if (edges) hipLaunchKernelGGL((lists_sims_kernel<NV, true, QP_>), sg, dim3(256), CAPAMD_LISTS_SIMS_BALLAST * 1024, s, a, g);   \
  else hipLaunchKernelGGL((lists_sims_kernel<NV, false, QP_>), sg, dim3(256), CAPAMD_LISTS_SIMS_BALLAST * 1024, s, a, g)
#define CAPAMD_SIMS(NV)                                                                                         \
  hipLaunchKernelGGL(lists_query_kernel<NV>, dim3(nl, QP), dim3(128), 0, s, a, g);                              \
  lists_stamp(s);                                                                                               \
  if (QP == 1) { CAPAMD_SIMS_Q(NV, 1); } else { CAPAMD_SIMS_Q(NV, 2); }
    switch (nv_for_dim(D)) {
      case 1: CAPAMD_SIMS(1); break;
      case 2: CAPAMD_SIMS(2); break;
      case 3: CAPAMD_SIMS(3); break;
      case 4: CAPAMD_SIMS(4); break;
      default: CAPAMD_SIMS(5); break;
    }
#undef CAPAMD_SIMS
#undef CAPAMD_SIMS_Q
    lists_stamp(s);
    pool(a, g, nl, longest);
    lists_stamp(s);
    if (hipGetLastError() != hipSuccess) return CAPAMD_ERR_LAUNCH;
  }
  return CAPAMD_OK;
}

}  // namespace
