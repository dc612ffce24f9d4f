// The passes every whole-candidate-list entry shares (lists.hip: KNRM, DRMM, DRMM-TKS; pacrr.hip: PACRR): list geometry, the mark pass,
// the per-list query image, the sims pass (table of four similarities - or DRMM's four bins - per distinct term of a list) and the host
// loop over launch groups.  See lists.hip for the design.  Everything here is TU-local (anonymous namespace): each including file gets
// its own copy of the kernels.
#pragma once
#include "capreolus_amd.h"
#include "interaction.cuh"
#include <stdlib.h>

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
  float4* table;           // [lists][Vp] entry of a flagged id (written for flagged ids only): KNRM 4 floats, DRMM 4 bytes (at 4 B stride)
  int* status;
  int nl, longest;         // lists in this launch group, documents of the longest
  const float* edges;      // DRMM: histogram bin edges
  int nbins;
  float4* qimg;            // [lists][kQueryImage] the list's query rows as the sims pass wants them in LDS
  struct ListQuery* qmeta; // [lists]
  const float* kn_mu;      // KNRM: the kernels' parameters (null for DRMM) ...
  const float* kn_sigma;
  int kn_K;
  int32_t* cid;            // [pairs][cid_stride] every document's REAL terms (0 < id < V), dense, in document order - written by the mark pass
  int32_t* meta;           // [pairs][kDocMeta] n_real, n_oov, n_one[kQT] (OOV terms equal to the list's OOV query term t)
  int cid_stride;
  float* kn_consts;        // ... and what the pooling pass needs of them, computed once per call: [6][kMaxK] mu, c = -log2(e) / (2 sigma^2),
                           //     K(0) = 2^(c mu^2), K(1) = 2^(c (1 - mu)^2), A = sqrt(-c), B = -A mu   (slots beyond K repeat the last kernel)
  float4* qplain;          // [lists][64 NV] the list's query rows once more, as [column][term]: what the dense head pass stages in LDS
  float* head;             // [H / 16][64 NV][16] the first H table rows, sixteen by sixteen, column-major inside a block (lists_head_pack_kernel)
  int H;                   // rows 0 .. H - 1 get their similarities from the dense matrix-pipe pass (lists_head_sims_kernel), for EVERY list
  int preflag;             // ids below this count as flagged in every list (lists_clear_kernel): the mark pass does not store their bytes
  const float* list_idf;   // [B, Q] / [NQ, Q]: the idf rows of models that read the LIST's (its first pair's) row - DRMM, DRMM-TKS; else null
};

constexpr int kQueryImage = kQT * kMaxNV * 16;      // float4s
constexpr int kDocMeta = 8;                          // ints of per-document metadata the mark pass leaves for the pooling pass
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
  int qo[kQT];                // the list's OOV query terms (0: not an OOV term)
#pragma unroll
  for (int t = 0; t < kQT; ++t) {
    const int64_t q = t < a.Q ? qids.q(t) : 0;
    qo[t] = (q < 0 && q > -2147483648LL) ? (int)q : 0;
  }
  // A list is ONE query against its documents: the passes below use the first pair's query row (and, where the model gates by idf, its
  // idf row) for all of them.  A pair that brings another row is a caller error - said through the status word instead of scoring the
  // pair against a query it does not have.  (Wave-uniform loads of 2 Q ids per document next to its L.)
  if (doc > 0 && ids.qrow != qids.qrow) {
    bool differs = false;
#pragma unroll
    for (int t = 0; t < kQT; ++t)
      if (t < a.Q) {
        differs |= ids.q(t) != qids.q(t);
        if (a.list_idf)
          differs |= __float_as_uint(a.list_idf[(int64_t)ids.qrow * a.Q + t]) != __float_as_uint(a.list_idf[(int64_t)qids.qrow * a.Q + t]);
      }
    if (differs && lane == 0) atomicOr(a.status, kErrListQuery);
  }
  uint8_t* f = a.flags + (int64_t)l * a.Vp;
  int32_t* out = (EMIT && kCompactRows) ? a.cid + (int64_t)b * a.cid_stride : nullptr;
  int n_real = 0, c_oov = 0, c_one[kQT] = {0, 0, 0, 0}, used = 0;      // used: positions up to the last one that is not a pad
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
#pragma unroll
          for (int t = 0; t < kQT; ++t)
            if (qo[t] != 0) c_one[t] += __builtin_popcountll(__ballot(vi == qo[t]));
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
      for (int t = 0; t < kQT; ++t) m[2 + t] = c_one[t];
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
template <int TRIPS, int STRIDE>
__device__ __forceinline__ void load_pass_cids(const int32_t* row, int j0, int ps, int n, int (&id)[TRIPS]) {
  if (j0 + TRIPS * STRIDE <= n) {
    const int32_t* p = row + j0 + ps;
#pragma unroll
    for (int u = 0; u < TRIPS; ++u) id[u] = __builtin_nontemporal_load(p + u * STRIDE);
  } else {
#pragma unroll
    for (int u = 0; u < TRIPS; ++u) {
      const int j = j0 + u * STRIDE + ps;
      id[u] = __builtin_nontemporal_load(row + (j < n ? j : 0));
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
  w.row = kCompactRows ? a.cid + (int64_t)b * a.cid_stride : nullptr;
  w.n = dm[6];
  w.V = a.V;
  return w;
}
template <int TRIPS, int STRIDE>
__device__ __forceinline__ void load_pass(const DocWalk& w, int j0, int ps, int (&id)[TRIPS]) {
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

// the query of every list once: its packed rows in the PAIRED LDS layout of rows_dot_pk, its ids and norms
// The sims pass runs on the fp32 VALU (lists_sims_kernel).  -DCAPAMD_LISTS_SIMS_MFMA builds the round-4 experiment instead - the same
// partial chains on v_mfma_f32_4x4x1_16b_f32, bit-identical, measured SLOWER (see lists_sims_mfma_kernel) - with its plain query image.
#ifdef CAPAMD_LISTS_SIMS_MFMA
constexpr bool kSimsOnMfma = true;
#define CAPAMD_SIMS_KERNEL lists_sims_mfma_kernel
#else
constexpr bool kSimsOnMfma = false;
#ifndef CAPAMD_LISTS_SIMS_QREG
#define CAPAMD_LISTS_SIMS_QREG 0       // 1: lists_sims_qreg_kernel - the list's query rows in REGISTERS instead of LDS (see there)
#endif
#if CAPAMD_LISTS_SIMS_QREG
#define CAPAMD_SIMS_KERNEL lists_sims_qreg_kernel
#else
#define CAPAMD_SIMS_KERNEL lists_sims_kernel
#endif
#endif
#ifndef CAPAMD_SIMS_QREG_BLOCKS
#define CAPAMD_SIMS_QREG_BLOCKS 4      // id blocks (of kSimsIds ids) one workgroup of lists_sims_qreg_kernel walks with one copy of the query
#endif
constexpr int kSimsBlocksPerWG = (CAPAMD_LISTS_SIMS_QREG && !kSimsOnMfma) ? CAPAMD_SIMS_QREG_BLOCKS : 1;
template <int NV>
__global__ __launch_bounds__(128) void lists_query_kernel(ListsArgs a, ListGeom g) {
  __shared__ __attribute__((aligned(16))) float4 qlds[kQueryImage];
  const int l = blockIdx.x, tid = threadIdx.x, lane16 = tid & 15;
  const PairIds ids = pair_ids(a.ids, g.start[l], a.Q, a.L);
  QueryPass<NV> qp;
  // (the MFMA sims pass reads the plain layout [(term * NV + chunk) * 16 + piece]; the VALU one the paired layout of rows_dot_pk)
  load_query_pass_lds<NV, !kSimsOnMfma>(a.packed, ids, a.Q, 0, a.V, tid, 128, lane16, qlds, qp, a.status);
  __syncthreads();
  float4* img = a.qimg + (int64_t)l * kQueryImage;
  for (int i = tid; i < kQT * NV * 16; i += 128) img[i] = qlds[i];
  if (a.H > 0 && !kSimsOnMfma) {
    // [column][term] for the dense head pass.  In the paired image float (term 2 P + s, column 64 i + 4 p + e) is component 2 (e & 1) + s
    // of float4 ((P NV + i) 2 + (e >> 1)) 16 + p (rows_dot2_pk)
    const float* qf = reinterpret_cast<const float*>(qlds);
    float* out = reinterpret_cast<float*>(a.qplain + (int64_t)l * (64 * kMaxNV));
    for (int idx = tid; idx < 64 * NV * 4; idx += 128) {
      const int col = idx >> 2, t = idx & 3, i = col >> 6, p = (col >> 2) & 15, e = col & 3, P = t >> 1;
      out[idx] = qf[((((P * NV + i) * 2 + (e >> 1)) * 16 + p) << 2) + 2 * (e & 1) + (t & 1)];
    }
  }
  if (tid < kQT) {           // lane16 = tid: the term this lane "owns" in QueryPass
    a.qmeta[l].id[tid] = qp.id_my;
    a.qmeta[l].den[tid] = qp.den_my;
  }
  if (l == 0 && a.kn_consts && tid < kMaxK) {
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

#ifndef CAPAMD_LISTS_SIMS_MFMA
#ifndef CAPAMD_LISTS_SIMS_HALF
#define CAPAMD_LISTS_SIMS_HALF 1      // 0: every list's four dot products per row (A/B builds)
#endif
#ifndef CAPAMD_LISTS_SIMS_ROWS
#define CAPAMD_LISTS_SIMS_ROWS 2      // rows per 16-lane group and trip (3: A/B builds)
#endif
#ifndef CAPAMD_LISTS_SIMS_WAVES
#define CAPAMD_LISTS_SIMS_WAVES 1
#endif
template <int NV, bool BINS>
__global__ __launch_bounds__(256, CAPAMD_LISTS_SIMS_WAVES) void lists_sims_kernel(ListsArgs a, ListGeom g) {
  __shared__ __attribute__((aligned(16))) float4 qlds[kQT * kMaxNV * 16];
  __shared__ int lst[kSimsIds];
  __shared__ int wave_cnt[4];
  __shared__ float edges[kMaxBins];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lane16 = tid & 15, grp = tid >> 4;
  // XCD x (workgroups whose linear index is x mod 8: blockIdx.x = 8 * list + x) takes the id blocks 8 k + x, each for all lists back to back
  const int l = blockIdx.x >> 3, blk = blockIdx.y * 8 + (blockIdx.x & 7);
  if ((int64_t)blk * kSimsIds >= a.Vp) return;
  const int id0 = blk * kSimsIds;
  constexpr int kPer = kSimsIds / 256;      // ids per thread: their flag bytes in one load
  static_assert(kPer == 2 || kPer == 4 || kPer == 8, "kSimsIds is 512, 1024 or 2048");
  const uint8_t* fp = a.flags + (int64_t)l * a.Vp + id0 + tid * kPer;
  if (id0 + kSimsIds <= a.H) return;          // the dense head pass has these rows
  uint64_t fw = kPer == 2 ? (uint64_t)*reinterpret_cast<const uint16_t*>(fp) : kPer == 4 ? (uint64_t)*reinterpret_cast<const uint32_t*>(fp)
                                                                                         : *reinterpret_cast<const uint64_t*>(fp);
  {
    const int below = a.H - (id0 + tid * kPer);        // this thread's first `below` ids belong to the head pass
    if (below > 0) fw = below >= kPer ? 0ull : fw & (~0ull << (8 * below));
  }
  // the list's query rows: the LDS image lists_query_kernel left (built here, from the ids, it is five dependent loads per workgroup)
  {
    const float4* img = a.qimg + (int64_t)l * kQueryImage;
    for (int i = tid; i < kQT * NV * 16; i += 256) qlds[i] = img[i];
  }
  QueryPass<NV> qp;
  qp.den_my = a.qmeta[l].den[lane16 & 3];
  qp.id_my = a.qmeta[l].id[lane16 & 3];
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
  const int total = c0 + c1 + c2 + c3;
  if (total == 0) return;
  const int base = wave == 0 ? 0 : wave == 1 ? c0 : wave == 2 ? c0 + c1 : c0 + c1 + c2;
#pragma unroll
  for (int c = 0; c < kPer; ++c)
    if ((fw >> (8 * c)) & 0xffu) lst[base + slot[c]] = tid * kPer + c;
  __syncthreads();
  float* tab = reinterpret_cast<float*>(a.table + (int64_t)l * a.Vp);
  uint8_t* tabb = reinterpret_cast<uint8_t*>(reinterpret_cast<uint32_t*>(a.table) + (int64_t)l * a.Vp);
  auto put = [&](int id, float sm) {        // lane l: the similarity of query term l & 3
    if (lane16 < kQT) {
      if (BINS) {
        const unsigned bin = (unsigned)list_bin_of(sm, edges, a.nbins) | ((sm > 0.999f && sm < 1.001f) ? kBinExact : 0u);
        tabb[(int64_t)id * 4 + lane16] = (uint8_t)bin;
      } else {
        tab[(int64_t)id * 4 + lane16] = sm;
      }
    }
  };
  // two rows per group and trip: the LDS query copy is read once for both; packed fmas (rows_dot2_pk: per row and term the fma order of rows_dot)
#if CAPAMD_LISTS_SIMS_ROWS == 3
#pragma clang loop unroll(disable)
  for (int e = grp; e < total; e += 3 * kGroupsPerWG) {
    const int ida = id0 + lst[e], idb = id0 + lst[e + kGroupsPerWG < total ? e + kGroupsPerWG : e],
              idc = id0 + lst[e + 2 * kGroupsPerWG < total ? e + 2 * kGroupsPerWG : e];
    RowRegs<NV> da, db, dc;
    load_row<NV>(a.packed, ida, lane16, da);
    load_row<NV>(a.packed, idb, lane16, db);
    load_row<NV>(a.packed, idc, lane16, dc);
    float pa[kQT], pb[kQT], pc[kQT];
    int qoff = 0;
    asm volatile("" : "+v"(qoff));
    rows_dot3_pk<NV>(da, db, dc, qlds + qoff, lane16, pa, pb, pc);
    put(ida, sim_from_dots<NV>(pa, row_den<NV>(da), qp, lane16));
    put(idb, sim_from_dots<NV>(pb, row_den<NV>(db), qp, lane16));
    put(idc, sim_from_dots<NV>(pc, row_den<NV>(dc), qp, lane16));
  }
  return;
#endif
#if CAPAMD_LISTS_SIMS_HALF
  // A list whose query terms 2 and 3 are not real (a query of one or two terms in the reference's fixed-length row: half of the
  // benchmark's queries) needs the first pair's dot products only: the other two similarities are 0 by definition (sim_from_dots) -
  // half the fmas, LDS query reads and row reductions of a trip; the same table entries, bit for bit.
  if (a.qmeta[l].id[2] <= 0 && a.qmeta[l].id[3] <= 0) {
#pragma clang loop unroll(disable)
    for (int e = grp; e < total; e += 2 * kGroupsPerWG) {
      const int ida = id0 + lst[e], idb = id0 + lst[e + kGroupsPerWG < total ? e + kGroupsPerWG : e];
      RowRegs<NV> da, db;
      load_row<NV>(a.packed, ida, lane16, da);
      load_row<NV>(a.packed, idb, lane16, db);
      float pa[kQT], pb[kQT];
      int qoff = 0;
      asm volatile("" : "+v"(qoff));
      rows_dot2_pk<NV, 1>(da, db, qlds + qoff, lane16, pa, pb);
      put(ida, sim_from_dots<NV, 2>(pa, row_den<NV>(da), qp, lane16));
      put(idb, sim_from_dots<NV, 2>(pb, row_den<NV>(db), qp, lane16));
    }
    return;
  }
#endif
#pragma clang loop unroll(disable)
  for (int e = grp; e < total; e += 2 * kGroupsPerWG) {
    const int ida = id0 + lst[e], idb = id0 + lst[e + kGroupsPerWG < total ? e + kGroupsPerWG : e];
    RowRegs<NV> da, db;
#ifdef CAPAMD_LISTS_ABL_HOTROWS      // ablation: every row load hits one of 16 rows (what the pass costs without its gather)
    load_row<NV>(a.packed, 1 + (ida & 15), lane16, da);
    load_row<NV>(a.packed, 1 + (idb & 15), lane16, db);
#else
    load_row<NV>(a.packed, ida, lane16, da);
    load_row<NV>(a.packed, idb, lane16, db);
#endif
    float pa[kQT], pb[kQT];
    int qoff = 0;
    asm volatile("" : "+v"(qoff));
    rows_dot2_pk<NV>(da, db, qlds + qoff, lane16, pa, pb);
    put(ida, sim_from_dots<NV>(pa, row_den<NV>(da), qp, lane16));
    put(idb, sim_from_dots<NV>(pb, row_den<NV>(db), qp, lane16));    // (an odd last row is done twice: a branch here makes hipcc sink row b's
                                                                     //  fma chain into it and keep the whole query copy in registers for that)
  }
  // (one row per trip with the NEXT row requested before the current one is used - a software pipeline - is 4-5 % slower end to end;
  //  FOUR rows per trip - half the LDS query reads per row, twice the loads in flight, 126 registers - 10 % slower: 317-327 against 291-293 us)
}

// The same pass with the list's query rows in REGISTERS (round 5; -DCAPAMD_LISTS_SIMS_QREG=1).  lists_sims_kernel reads the query copy
// from LDS again for every pair of rows: 20 ds_read_b128 per lane and trip = 20 KiB per wave and trip, 8 GB per call through a 128 B/clk
// LDS pipe - 117 us of LDS time beside 108 us of VALU time in a 289 us pass, and a dependent wait per read.  A lane only ever needs ITS 20
// float4 of the image (2 term pairs x NV chunks x 2 halves at its lane16): 80 registers, loaded once per workgroup, which then walks
// kSimsBlocksPerWG id blocks with them.  Same fma order per (row, term) as rows_dot2_pk: the same bits.
template <int NV>
__device__ __forceinline__ void rows_dot2_pk_reg(const RowRegs<NV>& d0, const RowRegs<NV>& d1, const float4 (&qr)[2 * NV * 2], float (&p0)[kQT],
                                                 float (&p1)[kQT]) {
  f32x2 acc0[2] = {{0.f, 0.f}, {0.f, 0.f}}, acc1[2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
  for (int i = 0; i < NV; ++i) {
#pragma unroll
    for (int P = 0; P < 2; ++P) {
      const float4 qa = qr[(P * NV + i) * 2 + 0], qb = qr[(P * NV + i) * 2 + 1];
      f32x2 a = acc0[P], b = acc1[P];
      a = __builtin_elementwise_fma((f32x2){d0.v[i].x, d0.v[i].x}, (f32x2){qa.x, qa.y}, a);
      b = __builtin_elementwise_fma((f32x2){d1.v[i].x, d1.v[i].x}, (f32x2){qa.x, qa.y}, b);
      a = __builtin_elementwise_fma((f32x2){d0.v[i].y, d0.v[i].y}, (f32x2){qa.z, qa.w}, a);
      b = __builtin_elementwise_fma((f32x2){d1.v[i].y, d1.v[i].y}, (f32x2){qa.z, qa.w}, b);
      a = __builtin_elementwise_fma((f32x2){d0.v[i].z, d0.v[i].z}, (f32x2){qb.x, qb.y}, a);
      b = __builtin_elementwise_fma((f32x2){d1.v[i].z, d1.v[i].z}, (f32x2){qb.x, qb.y}, b);
      a = __builtin_elementwise_fma((f32x2){d0.v[i].w, d0.v[i].w}, (f32x2){qb.z, qb.w}, a);
      b = __builtin_elementwise_fma((f32x2){d1.v[i].w, d1.v[i].w}, (f32x2){qb.z, qb.w}, b);
      acc0[P] = a;
      acc1[P] = b;
    }
  }
  p0[0] = acc0[0].x; p0[1] = acc0[0].y; p0[2] = acc0[1].x; p0[3] = acc0[1].y;
  p1[0] = acc1[0].x; p1[1] = acc1[0].y; p1[2] = acc1[1].x; p1[3] = acc1[1].y;
}

// CAPAMD_LISTS_SIMS_QREG = 2: the multi-block walk with the query copy in LDS (lists_sims_kernel's arithmetic, its per-workgroup
// prologue - query copy, norms, edges - once per kSimsBlocksPerWG id blocks)
constexpr bool kSimsQueryInLds = CAPAMD_LISTS_SIMS_QREG == 2;
template <int NV, bool BINS>
__global__ __launch_bounds__(256, 2) void lists_sims_qreg_kernel(ListsArgs a, ListGeom g) {
  __shared__ __attribute__((aligned(16))) float4 qlds[kSimsQueryInLds ? kQT * kMaxNV * 16 : 1];
  __shared__ int lst[kSimsIds];
  __shared__ int wave_cnt[4];
  __shared__ float edges[kMaxBins];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lane16 = tid & 15, grp = tid >> 4;
  const int l = blockIdx.x >> 3, blk0 = (blockIdx.y * 8 + (blockIdx.x & 7)) * kSimsBlocksPerWG;
  if ((int64_t)blk0 * kSimsIds >= a.Vp) return;
  constexpr int kPer = kSimsIds / 256;
  float4 qr[kSimsQueryInLds ? 1 : 2 * NV * 2];
  {
    const float4* img = a.qimg + (int64_t)l * kQueryImage;
    if (kSimsQueryInLds) {
      for (int i = tid; i < kQT * NV * 16; i += 256) qlds[i] = img[i];
    } else {
#pragma unroll
      for (int k = 0; k < (kSimsQueryInLds ? 1 : 2 * NV * 2); ++k) qr[k] = img[k * 16 + lane16];
    }
  }
  QueryPass<NV> qp;
  qp.den_my = a.qmeta[l].den[lane16 & 3];
  qp.id_my = a.qmeta[l].id[lane16 & 3];
  if (BINS && tid < a.nbins) edges[tid] = a.edges[tid];
  float* tab = reinterpret_cast<float*>(a.table + (int64_t)l * a.Vp);
  uint8_t* tabb = reinterpret_cast<uint8_t*>(reinterpret_cast<uint32_t*>(a.table) + (int64_t)l * a.Vp);
  auto put = [&](int id, float sm) {
    if (lane16 < kQT) {
      if (BINS) {
        const unsigned bin = (unsigned)list_bin_of(sm, edges, a.nbins) | ((sm > 0.999f && sm < 1.001f) ? kBinExact : 0u);
        tabb[(int64_t)id * 4 + lane16] = (uint8_t)bin;
      } else {
        tab[(int64_t)id * 4 + lane16] = sm;
      }
    }
  };
#pragma unroll 1
  for (int bi = 0; bi < kSimsBlocksPerWG; ++bi) {
    const int id0 = (blk0 + bi) * kSimsIds;
    if (id0 >= a.Vp) break;
    if (id0 + kSimsIds <= a.H) continue;
    const uint8_t* fp = a.flags + (int64_t)l * a.Vp + id0 + tid * kPer;
    uint64_t fw = kPer == 2 ? (uint64_t)*reinterpret_cast<const uint16_t*>(fp) : kPer == 4 ? (uint64_t)*reinterpret_cast<const uint32_t*>(fp)
                                                                                           : *reinterpret_cast<const uint64_t*>(fp);
    {
      const int below = a.H - (id0 + tid * kPer);
      if (below > 0) fw = below >= kPer ? 0ull : fw & (~0ull << (8 * below));
    }
    int slot[kPer], mine = 0;
#pragma unroll
    for (int c = 0; c < kPer; ++c) {
      const uint64_t set = __ballot(((fw >> (8 * c)) & 0xffu) != 0);
      slot[c] = mine + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(set >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)set, 0u));
      mine += __builtin_popcountll(set);
    }
    __syncthreads();             // (the previous block's list and counts have been read by everyone)
    if (lane == 0) wave_cnt[wave] = mine;
    __syncthreads();
    const int c0 = wave_cnt[0], c1 = wave_cnt[1], c2 = wave_cnt[2], c3 = wave_cnt[3];
    const int total = c0 + c1 + c2 + c3;
    if (total == 0) continue;
    const int base = wave == 0 ? 0 : wave == 1 ? c0 : wave == 2 ? c0 + c1 : c0 + c1 + c2;
#pragma unroll
    for (int c = 0; c < kPer; ++c)
      if ((fw >> (8 * c)) & 0xffu) lst[base + slot[c]] = tid * kPer + c;
    __syncthreads();
#pragma clang loop unroll(disable)
    for (int e = grp; e < total; e += 2 * kGroupsPerWG) {
      const int ida = id0 + lst[e], idb = id0 + lst[e + kGroupsPerWG < total ? e + kGroupsPerWG : e];
      RowRegs<NV> da, db;
      load_row<NV>(a.packed, ida, lane16, da);
      load_row<NV>(a.packed, idb, lane16, db);
      float pa[kQT], pb[kQT];
      if constexpr (kSimsQueryInLds) {
        int qoff = 0;
        asm volatile("" : "+v"(qoff));
        rows_dot2_pk<NV>(da, db, qlds + qoff, lane16, pa, pb);
      } else {
        rows_dot2_pk_reg<NV>(da, db, qr, pa, pb);
      }
      put(ida, sim_from_dots<NV>(pa, row_den<NV>(da), qp, lane16));
      put(idb, sim_from_dots<NV>(pb, row_den<NV>(db), qp, lane16));
    }
  }
}

#endif

// ---- 2h: the dense head of the vocabulary on the matrix pipe ---------------------------------------------------------------------
// Vocabularies are frequency-ordered (GloVe's is; the benchmark's ids are Zipf ranks): on 1000-candidate lists the first ~16,000 ids are
// flagged in almost every list - 30 % of all (list, term) rows.  For those a gather buys nothing, and a dense product has the shape the
// matrix pipe wants: rows 0 .. H - 1 against the 4 x 4 query terms of FOUR lists = a [16 rows] x [16 columns] tile per
// v_mfma_f32_16x16x4_f32, for every list, flagged or not (an entry nobody looks up costs nothing).  BIT-IDENTICAL to lists_sims_kernel -
// the pooling kernels and the per-pair kernels must not see which pass produced a similarity: the VALU form sums, per (row, term), 16
// lane-partial fma chains (lane p: floats 64 c + 4 p + e, c ascending, e = x, y, z, w) and then a balanced tree over the partials
// (group_allreduce: p ^ 1, p ^ 2, the other quad, the other half).  An fp32 MFMA accumulates as a k-ordered fmaf chain, so partial p IS a
// chain of NV MFMAs over k = (c, e) into its own accumulator: 16 accumulators of 4 registers, 16 NV MFMAs per tile (the flops of one long
// chain, arranged as sixteen short ones), the tree as 15 register adds per result.  The operands arrive in MFMA layout without any turn:
// the A side from a copy of the head rows laid out [block of 16 rows][column][row] (lists_head_pack_kernel, 21 MB per call for H = 16,384:
// lane (row i, e) reads float (column 64 c + 4 p + e, row i) = 64 consecutive floats per instruction), the B side from an LDS image of the
// four lists' query rows laid out [column][16 query columns] (conflict-free: 64 consecutive floats per read).
// MEASURED AND NOT THE DEFAULT (profiles/r04/lists_head_mfma.txt; -DCAPAMD_LISTS_HEAD_ROWS=16384 builds it, every list-route parity test
// passes with it - bit-identical tables): the gather pass loses the 58 us of its head rows (285 -> 227 us), this pass costs 48 us
// (21 of MFMAs - an fp32 MFMA is the vector rate, not sixteen times it -, 13 of row loads that four blocks per wave cannot hide, 11 of
// launch / query staging / tree, 3 of stores) and the per-call copy of the head rows 9: no gain, for 21 MB more workspace.
#ifndef CAPAMD_LISTS_HEAD_ROWS
#define CAPAMD_LISTS_HEAD_ROWS 0          // 0: no dense head pass (every row through lists_sims_kernel)
#endif
constexpr int kHeadMax = CAPAMD_LISTS_HEAD_ROWS;
#ifndef CAPAMD_HEAD_ABL
#define CAPAMD_HEAD_ABL 0      // profiling builds of lists_head_sims_kernel: 1 = no stores, 2 = no MFMAs, 4 = no row loads
#endif
typedef float f32x4m __attribute__((ext_vector_type(4)));

// rows of the dense head for a call: a term of rank r is in a list when the list holds about r positions, so H follows the list length
// (positions per list / 32, pads included: ~1/12 of the real positions), capped by the table and by kHeadMax
inline int lists_head_rows(int64_t V, int64_t n_pairs, int L, int n_lists) {
  if (kHeadMax <= 0 || n_lists < 1) return 0;
  int64_t h = n_pairs * L / n_lists / 32;
  if (h > kHeadMax) h = kHeadMax;
  if (h > V) h = V;
  return (int)(h / 16 * 16);
}

template <int NV>
__global__ __launch_bounds__(256) void lists_head_pack_kernel(ListsArgs a) {
  __shared__ float t[64 * NV * 17];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* src = a.packed + (int64_t)b * 16 * (64 * NV);
  for (int i = tid; i < 16 * 64 * NV; i += 256) {
    const int r = i / (64 * NV), c = i - r * (64 * NV);
    t[c * 17 + r] = src[i];
  }
  __syncthreads();
  float* dst = a.head + (int64_t)b * 16 * (64 * NV);
  for (int i = tid; i < 16 * 64 * NV; i += 256) dst[i] = t[(i >> 4) * 17 + (i & 15)];
}

template <int NV, bool BINS>
__global__ __launch_bounds__(256, 4) void lists_head_sims_kernel(ListsArgs a, ListGeom g) {
  __shared__ __attribute__((aligned(16))) float QT[64 * NV * 16];        // [column][query column j = 4 (list of the group) + term]
  __shared__ float qden[16];
  __shared__ int qid[16];
  __shared__ float edges[kMaxBins];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l0 = blockIdx.y * 4;            // the group's lists l0 .. l0 + 3
  // the four lists' [column][term] images side by side: one float4 per (column, list) - unconditional loads, a clamped list index
  for (int idx = tid; idx < 64 * NV * 4; idx += 256) {
    const int col = idx >> 2, lg = idx & 3, ll = l0 + lg;
    const float4 v = a.qplain[(int64_t)(ll < a.nl ? ll : a.nl - 1) * (64 * kMaxNV) + col];
    *reinterpret_cast<float4*>(QT + col * 16 + 4 * lg) = ll < a.nl ? v : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (tid < 16) {
    const int ll = l0 + (tid >> 2);
    qden[tid] = ll < a.nl ? a.qmeta[ll].den[tid & 3] : 1.f;
    qid[tid] = ll < a.nl ? a.qmeta[ll].id[tid & 3] : 0;
  }
  if (BINS && tid < a.nbins) edges[tid] = a.edges[tid];
  __syncthreads();
  const int j = lane & 15, ll = l0 + (j >> 2), t = j & 3, r0 = 4 * (lane >> 4);
  const float qd = qden[j];
  const bool live = qid[j] > 0;
  float* tab = reinterpret_cast<float*>(a.table + (int64_t)(ll < a.nl ? ll : 0) * a.Vp);
  uint8_t* tabb = reinterpret_cast<uint8_t*>(reinterpret_cast<uint32_t*>(a.table) + (int64_t)(ll < a.nl ? ll : 0) * a.Vp);
  // a wave walks row blocks (the query image above is built once per workgroup: sixteen-row blocks are 2.5 k MFMA cycles each)
  for (int blk = blockIdx.x * 4 + wave; blk * 16 < a.H; blk += gridDim.x * 4) {          // (no barrier below)
  const float* hp = a.head + (int64_t)blk * 16 * (64 * NV) + lane;
  // (the query operands are re-read from LDS for every block: hoisted out of this loop - they do not depend on the block - they occupy
  //  16 NV registers per lane, and what hides a block's 16 NV row loads is the number of resident waves, four per SIMD at <= 128 registers)
  int qoff = lane;
  asm volatile("" : "+v"(qoff));
  f32x4m acc[16];
#pragma unroll
  for (int p = 0; p < 16; ++p) acc[p] = f32x4m{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    float av[16], bv[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) av[p] = (CAPAMD_HEAD_ABL & 4) ? (float)(p + c) : hp[(64 * c + 4 * p) * 16];
#pragma unroll
    for (int p = 0; p < 16; ++p) bv[p] = QT[(64 * c + 4 * p) * 16 + qoff];
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      if (CAPAMD_HEAD_ABL & 2) acc[p][0] += av[p] * bv[p];
      else acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[p], bv[p], acc[p], 0, 0, 0);
    }
  }
  // C/D map: column j = lane & 15, rows 4 (lane >> 4) + r.  The tree of group_allreduce over the partials, in registers.
  f32x4m lv[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) lv[p] = acc[2 * p] + acc[2 * p + 1];
  const f32x4m s = ((lv[0] + lv[1]) + (lv[2] + lv[3])) + ((lv[4] + lv[5]) + (lv[6] + lv[7]));
  if (ll >= a.nl) continue;
  const float4 dd = *reinterpret_cast<const float4*>(a.head + (int64_t)blk * 16 * (64 * NV) + (64 * NV - 1) * 16 + r0);   // the rows' norms: their last float
  const float dden[4] = {dd.x, dd.y, dd.z, dd.w};
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int id = blk * 16 + r0 + r;
    if (id == 0) continue;                  // (the pad row: never looked up)
    if ((CAPAMD_HEAD_ABL & 1) && s[r] != 12345.f) continue;
    const float q = s[r] / (qd * dden[r]);
    const float sm = live ? q : 0.f;
    if (BINS) {
      const unsigned bin = (unsigned)list_bin_of(sm, edges, a.nbins) | ((sm > 0.999f && sm < 1.001f) ? kBinExact : 0u);
      tabb[(int64_t)id * 4 + t] = (uint8_t)bin;
    } else {
      tab[(int64_t)id * 4 + t] = sm;
    }
  }
  }
}

// ---- 2': sims on the matrix pipe (round-4 experiment, not the default: -DCAPAMD_LISTS_SIMS_MFMA) --------------------------------------
#ifdef CAPAMD_LISTS_SIMS_MFMA
// The same dot products - per (row, query term) the 16 lane-partial fma chains of rows_dot and their balanced tree, bit for bit - with
// the chains on v_mfma_f32_4x4x1_16b_f32 instead of the fp32 VALU.  That instruction is 16 independent 4 x 4 outer products, K = 1:
// D_b[i][j] += A_b[i] * B_b[j] for blocks b = 0..15, lane 4 b + i supplying A_b[i], lane 4 b + j supplying B_b[j] and keeping column j of
// D_b in its four result registers; and an fp32 MFMA accumulates as a k-ordered fmaf chain, bit for bit (MI355X_MICROARCH.md, checked by
// scripts/ubench/valu_rates.hip).  So: block b IS lane-partial b of the VALU form (the chain over floats 64 c + 4 b + e, c ascending, e =
// x, y, z, w), i = one of FOUR table rows, j = one of the four query terms: 20 instructions (NV = 5) give four rows' 16 x 4 partial
// chains at twice the VALU's fma rate, the VALU left to the reductions - 3 DPP adds per result register for the in-row levels of the tree
// (partials b ^ 1, b ^ 2) and a reduce-scatter over the wave's four rows for the last two (b ^ 4, b ^ 8), after which row R of the wave
// holds the finished dot products of the R-th four-row group of a 16-row batch: 12 shuffles per 16 rows where the VALU form spends 64
// DPP adds per ROW.  What the MFMA wants - lane = (partial, row), i.e. four different rows in every quad of lanes - is not what a gather
// can deliver (a quad of lanes on four rows is four cache lines per request): rows are fetched as before, a 16-lane group per row, and
// turned through a wave-private LDS stage (ds_write_b128 at slot 4 p + i, then one LINEAR ds_read_b128 per chunk: conflict-free).
typedef float f32x4v __attribute__((ext_vector_type(4)));
#ifndef CAPAMD_SIMS_MFMA_BLOCKS
#define CAPAMD_SIMS_MFMA_BLOCKS 5      // workgroups per CU the register budget is set for (<= 96 registers)
#endif

__device__ __forceinline__ float lane_xor4(float v) {        // the value of lane ^ 4: quad reversed, then the 8-lane half mirrored
  return dpp_mov<0x141>(dpp_mov<0x1B>(v));
}

template <int NV, bool BINS>
__global__ __launch_bounds__(256, CAPAMD_SIMS_MFMA_BLOCKS) void lists_sims_mfma_kernel(ListsArgs a, ListGeom g) {
  __shared__ int lst[kSimsIds];
  static_assert(kSimsIds <= 4096, "16 flag bytes per thread at most");
  __shared__ int wave_cnt[4];
  __shared__ float edges[kMaxBins];
  __shared__ __attribute__((aligned(16))) f32x4v stage[4][NV * 64];        // [wave][chunk * 64 + 4 * piece + row of the group]
  __shared__ __attribute__((aligned(16))) float dens[4][8];               // [wave][row of the batch] the rows' norms
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, pl = tid & 15, grp = lane >> 4;
  // XCD x (workgroups whose linear index is x mod 8: blockIdx.x = 8 * list + x) takes the id blocks 8 k + x, each for all lists back to back
  const int l = blockIdx.x >> 3, blk = blockIdx.y * 8 + (blockIdx.x & 7);
  if ((int64_t)blk * kSimsIds >= a.Vp) return;
  const int id0 = blk * kSimsIds;
  constexpr int kPer = kSimsIds / 256;      // ids per thread: their flag bytes in one load
  static_assert(kPer == 2 || kPer == 4 || kPer == 8 || kPer == 16, "kSimsIds is 512, 1024, 2048 or 4096");
  const uint8_t* fp = a.flags + (int64_t)l * a.Vp + id0 + tid * kPer;
  uint64_t fw, fw2 = 0;
  if (kPer == 16) {
    const uint4 w = *reinterpret_cast<const uint4*>(fp);
    fw = (uint64_t)w.x | ((uint64_t)w.y << 32);
    fw2 = (uint64_t)w.z | ((uint64_t)w.w << 32);
  } else {
    fw = kPer == 2 ? (uint64_t)*reinterpret_cast<const uint16_t*>(fp) : kPer == 4 ? (uint64_t)*reinterpret_cast<const uint32_t*>(fp)
                                                                                   : *reinterpret_cast<const uint64_t*>(fp);
  }
  auto flag_of = [&](int c) { return (unsigned)(((c < 8 ? fw : fw2) >> (8 * (c & 7))) & 0xffu); };
  // this lane's B operands: query term j = lane & 3, the pieces of partial chain b = lane >> 2 (requested now, used after the compaction)
  const int j = lane & 3, b = lane >> 2;
  f32x4v qreg[NV];
  {
    const f32x4v* img = reinterpret_cast<const f32x4v*>(a.qimg + (int64_t)l * kQueryImage);
#pragma unroll
    for (int c = 0; c < NV; ++c) qreg[c] = img[(j * NV + c) * 16 + b];
  }
  const float qden = a.qmeta[l].den[j];
  const int qid = a.qmeta[l].id[j];
  if (BINS && tid < a.nbins) edges[tid] = a.edges[tid];
  // the flagged ids, dense, in LDS (any order): per flag byte one ballot, the lane's slot = the set lanes below it
  int slot[kPer], mine = 0;
#pragma unroll
  for (int c = 0; c < kPer; ++c) {
    const uint64_t set = __ballot(flag_of(c) != 0);
    slot[c] = mine + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(set >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)set, 0u));
    mine += __builtin_popcountll(set);       // (wave-uniform from here: the wave's count so far)
  }
  if (lane == 0) wave_cnt[wave] = mine;
  __syncthreads();
  const int c0 = wave_cnt[0], c1 = wave_cnt[1], c2 = wave_cnt[2], c3 = wave_cnt[3];
  const int total = c0 + c1 + c2 + c3;
  if (total == 0) return;
  const int base = wave == 0 ? 0 : wave == 1 ? c0 : wave == 2 ? c0 + c1 : c0 + c1 + c2;
#pragma unroll
  for (int c = 0; c < kPer; ++c)
    if (flag_of(c)) lst[base + slot[c]] = tid * kPer + c;
  __syncthreads();
  float* tab = reinterpret_cast<float*>(a.table + (int64_t)l * a.Vp);
  uint8_t* tabb = reinterpret_cast<uint8_t*>(reinterpret_cast<uint32_t*>(a.table) + (int64_t)l * a.Vp);
#ifndef CAPAMD_SIMS_ABL
#define CAPAMD_SIMS_ABL 0      // profiling builds: 1 = every row load hits one of 16 rows, 2 = no MFMAs
#endif
  // (native vectors, not float4 structs: a struct copied whole from memory to LDS stays a memcpy through a stack slot)
  auto fetch = [&](int id, f32x4v (&r)[NV]) {
    const f32x4v* p = reinterpret_cast<const f32x4v*>(a.packed + (int64_t)id * (64 * NV)) + pl;
#pragma unroll
    for (int c = 0; c < NV; ++c) r[c] = p[c * 16];
  };
  // Batches of EIGHT rows (two groups of four) per wave: what bounds this pass is how many row requests a CU keeps in flight while each
  // wave walks its chain request -> LDS turn -> MFMA chain -> reduction, i.e. the number of resident waves: 16-row batches with a second
  // register set for the next batch (200 registers, 8 waves per CU) measured 470 us per call, the fp32-VALU form 285.  A group's registers
  // are re-requested (the same group of the wave's next batch) right after the LDS store that frees them.
  auto fetch2 = [&](int batch, int G, f32x4v (&r)[NV]) {
    const int e = batch * 8 + G * 4 + grp;
    const int id = id0 + lst[e < total ? e : total - 1];
    fetch((CAPAMD_SIMS_ABL & 1) ? 1 + (id & 15) : id, r);
  };
  const int nb8 = (total + 7) >> 3;
  f32x4v* st = stage[wave];
  f32x4v rr[2][NV];
  int bt = wave;
  if (bt >= nb8) return;                      // (no barrier below)
  fetch2(bt, 0, rr[0]);
  fetch2(bt, 1, rr[1]);
  for (; bt < nb8; bt += 4) {
    const int next = bt + 4 < nb8 ? bt + 4 : bt;      // (the last batch is requested twice: a load under a condition makes its destination a merge
                                                      //  point, which hipcc resolves through scratch memory)
    f32x4v acc[2];
#pragma unroll
    for (int G = 0; G < 2; ++G) {
#pragma unroll
      for (int c = 0; c < NV; ++c) st[c * 64 + 4 * pl + grp] = rr[G][c];
      if (pl == 15) dens[wave][G * 4 + grp] = rr[G][NV - 1].w;
      fetch2(next, G, rr[G]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");              // (one wave: LDS program order is the synchronisation)
      f32x4v d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < NV; ++c) {
        const f32x4v av = st[c * 64 + lane];
        if (CAPAMD_SIMS_ABL & 2) {
          d += av;
          continue;
        }
        d = __builtin_amdgcn_mfma_f32_4x4x1f32(av.x, qreg[c].x, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_4x4x1f32(av.y, qreg[c].y, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_4x4x1f32(av.z, qreg[c].z, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_4x4x1f32(av.w, qreg[c].w, d, 0, 0, 0);
      }
      acc[G] = d;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    // levels 1, 2 of the tree (partials b ^ 1, b ^ 2: inside the 16-lane row): every lane of a row ends with the row's sum for its term
#pragma unroll
    for (int G = 0; G < 2; ++G)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v = acc[G][i];
        v += lane_xor4(v);
        v += dpp_mov<0x128>(v);            // row_ror:8 = lane ^ 8
        acc[G][i] = v;
      }
    // levels 3, 4 (b ^ 4, b ^ 8: the wave's rows R = lane >> 4): rows R and R ^ 1 exchange the group they do not keep, then R and R ^ 2
    // add what they hold of the same group - (R0 + R1) + (R2 + R3): rows 0 and 2 end with group 0, rows 1 and 3 with group 1
    const bool odd = (grp & 1) != 0;
    float fin[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float keep = odd ? acc[1][i] : acc[0][i], give = odd ? acc[0][i] : acc[1][i];
      const float t = keep + __shfl_xor(give, 16, 64);
      fin[i] = t + __shfl_xor(t, 32, 64);
    }
    // lane (R < 2, r, j): row r of group R against query term j
    const int r = (lane >> 2) & 3, e = bt * 8 + (grp & 1) * 4 + r;
    const float dot = r == 0 ? fin[0] : r == 1 ? fin[1] : r == 2 ? fin[2] : fin[3];
    const float dden = dens[wave][(grp & 1) * 4 + r];
    const float sdiv = dot / (qden * dden);
    const float sm = qid > 0 ? sdiv : 0.f;
    if (e < total && grp < 2) {
      const int id = id0 + lst[e];
      if (BINS) {
        const unsigned bin = (unsigned)list_bin_of(sm, edges, a.nbins) | ((sm > 0.999f && sm < 1.001f) ? kBinExact : 0u);
        tabb[(int64_t)id * 4 + j] = (uint8_t)bin;
      } else {
        tab[(int64_t)id * 4 + j] = sm;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // (dens is rewritten by the next batch)
  }
}
#endif   // CAPAMD_LISTS_SIMS_MFMA

// ---- host side -----------------------------------------------------------------------------------------------------------------
constexpr size_t kListQueryBytes = kQueryImage * sizeof(float4) + sizeof(ListQuery) + 64 * kMaxNV * sizeof(float4);   // both query images + ids / norms
// per call: the KNRM kernel constants, then the dense head's copy of the first rows (sized for the widest rows)
constexpr size_t kListHeadBytes = (size_t)(kHeadMax > 0 ? kHeadMax : 0) * 64 * kMaxNV * sizeof(float), kListKnBytes = 6 * kMaxK * sizeof(float);
constexpr size_t kListConstBytes = kListKnBytes + kListHeadBytes;
int64_t lists_vp(int64_t V) { return (V + kSimsIds - 1) / kSimsIds * kSimsIds; }
int lists_cid_stride(int L) { return (L + 3) & ~3; }
size_t lists_pair_bytes(int64_t n_pairs, int L) { return (size_t)n_pairs * ((kCompactRows ? (size_t)lists_cid_stride(L) * 4 : 0) + kDocMeta * 4); }

// runs `pool(geometry, lists in the chunk, longest list)` for chunks of lists that fit the workspace, after marking and the sims pass
template <class Pool>
int lists_run(const IdSource& ids, const int64_t* offsets_host, int n_lists, int Q, int L, const float* packed, int64_t V, int D, int* status,
              void* workspace, size_t workspace_bytes, hipStream_t s, const float* edges, int nbins, const float* kn_mu, const float* kn_sigma, int kn_K,
              bool emit, const float* list_idf, Pool pool) {
  if (!offsets_host || !packed || !status || !workspace) return CAPAMD_ERR_ARG;
  if (n_lists < 0 || Q < 1 || Q > kQT || L < 1 || L > 32768 || V < 1 || V > 0x7fffffffLL || capamd_packed_row_stride(D) < 0) return CAPAMD_ERR_ARG;
  if ((reinterpret_cast<uintptr_t>(workspace) & 15) != 0) return CAPAMD_ERR_ALIGN;
  const int64_t Vp = lists_vp(V);
  const size_t per_list = (size_t)Vp * 17 + kListQueryBytes;
  // workspace: [compact id rows + document metadata of ALL the call's pairs] [per list in flight: table, byte map, query image] [KNRM constants]
  const int64_t n_pairs = n_lists > 0 ? offsets_host[n_lists] : 0;
  if (n_pairs < 0 || n_pairs > 0x7fffffffLL) return CAPAMD_ERR_ARG;
  const size_t pair_bytes = emit ? lists_pair_bytes(n_pairs, L) : 0;
  if (workspace_bytes < kListConstBytes + pair_bytes) return CAPAMD_ERR_WORKSPACE;
  const size_t fit = (workspace_bytes - kListConstBytes - pair_bytes) / per_list;
  int cap = (int)(fit < (size_t)kListChunk ? fit : (size_t)kListChunk);
  if (cap < 1) return CAPAMD_ERR_WORKSPACE;
  if (cap > kListChunk) cap = kListChunk;
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
    // per-list part: table [cap][Vp] x 16 B | byte maps [cap][Vp] | query images [cap] | query ids and norms [cap] | KNRM kernel constants
    char* ws = static_cast<char*>(workspace) + pair_bytes;
    int32_t* cid = emit ? reinterpret_cast<int32_t*>(workspace) : nullptr;
    int32_t* meta = emit ? cid + (kCompactRows ? (size_t)n_pairs * lists_cid_stride(L) : 0) : nullptr;
    float4* table = reinterpret_cast<float4*>(ws);
    uint8_t* flags = reinterpret_cast<uint8_t*>(ws + (size_t)cap * Vp * 16);
    float4* qimg = reinterpret_cast<float4*>(ws + (size_t)cap * Vp * 17);
    ListQuery* qmeta = reinterpret_cast<ListQuery*>(qimg + (size_t)cap * kQueryImage);
    float4* qplain = reinterpret_cast<float4*>(qmeta + cap);
    float* kn_consts = reinterpret_cast<float*>(qplain + (size_t)cap * 64 * kMaxNV);
    float* head = reinterpret_cast<float*>(reinterpret_cast<char*>(kn_consts) + kListKnBytes);        // (16-byte aligned: every part before it is)
    if (!kn_mu) kn_consts = nullptr;
    const int H = lists_head_rows(V, n_pairs, L, n_lists);
    ListsArgs a{ids, Q, L, packed, V, Vp, flags, table, status, nl, longest, edges, nbins, qimg, qmeta, kn_mu, kn_sigma, kn_K, cid, meta, lists_cid_stride(L),
                kn_consts, qplain, head, H, lists_preflag(n_pairs, L, n_lists), list_idf};
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
    const dim3 sg((unsigned)nl * 8, (unsigned)((Vp / kSimsIds + 8 * kSimsBlocksPerWG - 1) / (8 * kSimsBlocksPerWG)));
#define CAPAMD_SIMS(NV)                                                                                         \
  hipLaunchKernelGGL(lists_query_kernel<NV>, dim3(nl), dim3(128), 0, s, a, g);                                  \
  lists_stamp(s);                                                                                               \
  if (H > 0) {                                                                                                  \
    const unsigned hgroups = (unsigned)(nl + 3) / 4, hwg = (unsigned)(H / 16 + 3) / 4;                          \
    const dim3 hg(hwg * hgroups <= 1024 ? hwg : (1024 / hgroups > 0 ? 1024 / hgroups : 1), hgroups);            \
    if (l0 == 0) hipLaunchKernelGGL(lists_head_pack_kernel<NV>, dim3(H / 16), dim3(256), 0, s, a);              \
    if (edges) hipLaunchKernelGGL((lists_head_sims_kernel<NV, true>), hg, dim3(256), 0, s, a, g);               \
    else hipLaunchKernelGGL((lists_head_sims_kernel<NV, false>), hg, dim3(256), 0, s, a, g);                    \
  }                                                                                                             \
  if (edges) hipLaunchKernelGGL((CAPAMD_SIMS_KERNEL<NV, true>), sg, dim3(256), 0, s, a, g);                     \
  else hipLaunchKernelGGL((CAPAMD_SIMS_KERNEL<NV, false>), sg, dim3(256), 0, s, a, g)
    switch (nv_for_dim(D)) {
      case 1: CAPAMD_SIMS(1); break;
      case 2: CAPAMD_SIMS(2); break;
      case 3: CAPAMD_SIMS(3); break;
      case 4: CAPAMD_SIMS(4); break;
      default: CAPAMD_SIMS(5); break;
    }
#undef CAPAMD_SIMS
    lists_stamp(s);
    pool(a, g, nl, longest);
    lists_stamp(s);
    if (hipGetLastError() != hipSuccess) return CAPAMD_ERR_LAUNCH;
  }
  return CAPAMD_OK;
}

}  // namespace
