// The passes every whole-candidate-list entry shares (lists.hip: KNRM, DRMM, DRMM-TKS; pacrr.hip: PACRR): list geometry, the mark pass,
// the per-list query image, the sims pass (table of four similarities - or DRMM's four bins - per distinct term of a list) and the host
// loop over launch groups.  See lists.hip for the design.  Everything here is TU-local (anonymous namespace): each including file gets
// its own copy of the kernels.
#pragma once
#include "capreolus_amd.h"
#include "interaction.cuh"
#include <stdlib.h>

namespace capamd {
constexpr int kListStamps = 6;        // stamps per launch group: before the memset and after each of memset, mark, query, sims, pool
#ifdef CAPAMD_PROFILING
void lists_stamp(hipStream_t s);      // lists.hip: records an event between passes while capamd_debug_lists_timing is on
#else
inline void lists_stamp(hipStream_t) {}
#endif
}

using namespace capamd;

namespace {

constexpr int kListChunk = 64;            // lists per launch group (their start / length travel as kernel arguments)
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
  float* kn_consts;        // ... and what the pooling pass needs of them, computed once per call: [4][kMaxK] mu, c = -log2(e) / (2 sigma^2),
                           //     K(0) = 2^(c mu^2), K(1) = 2^(c (1 - mu)^2)   (slots beyond K repeat the last kernel)
};

constexpr int kQueryImage = kQT * kMaxNV * 16;      // float4s
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

// ---- 1: mark -------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lists_mark_kernel(ListsArgs a, ListGeom g) {
  int l, doc;
  if (!list_doc_of(a, l, doc) || doc >= g.len[l]) return;
  const PairIds ids = pair_ids(a.ids, g.start[l] + doc, a.Q, a.L);
  uint8_t* f = a.flags + (int64_t)l * a.Vp;
  bool bad = false;
  for (int j0 = 0; j0 < a.L; j0 += 256 * 4) {
    int64_t id[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = j0 + u * 256 + (int)threadIdx.x;
      id[u] = j < a.L ? doc_id_at(ids, j) : 0;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (id[u] >= a.V) bad = true;
      else if (id[u] > 0) f[id[u]] = 1;        // (unconditional: a check of the flag first puts a load in front of every store and measures the same;
                                               //  so does an LDS bit map of the terms a workgroup of 16 documents has stored already: 123-142 us for 114)
    }
  }
  if (bad) atomicOr(a.status, kErrDocIdRange);
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
template <int NV>
__global__ __launch_bounds__(128) void lists_query_kernel(ListsArgs a, ListGeom g) {
  __shared__ __attribute__((aligned(16))) float4 qlds[kQueryImage];
  const int l = blockIdx.x, tid = threadIdx.x, lane16 = tid & 15;
  const PairIds ids = pair_ids(a.ids, g.start[l], a.Q, a.L);
  QueryPass<NV> qp;
  load_query_pass_lds<NV, true>(a.packed, ids, a.Q, 0, a.V, tid, 128, lane16, qlds, qp, a.status);
  __syncthreads();
  float4* img = a.qimg + (int64_t)l * kQueryImage;
  for (int i = tid; i < kQT * NV * 16; i += 128) img[i] = qlds[i];
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
  }
}

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
  const uint64_t fw = kPer == 2 ? (uint64_t)*reinterpret_cast<const uint16_t*>(fp) : kPer == 4 ? (uint64_t)*reinterpret_cast<const uint32_t*>(fp)
                                                                                               : *reinterpret_cast<const uint64_t*>(fp);
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
  // (one row per trip with the NEXT row requested before the current one is used - a software pipeline - is 4-5 % slower end to end)
}

// ---- host side -----------------------------------------------------------------------------------------------------------------
constexpr size_t kListQueryBytes = kQueryImage * sizeof(float4) + sizeof(ListQuery), kListConstBytes = 4 * kMaxK * sizeof(float);
int64_t lists_vp(int64_t V) { return (V + kSimsIds - 1) / kSimsIds * kSimsIds; }

// runs `pool(geometry, lists in the chunk, longest list)` for chunks of lists that fit the workspace, after marking and the sims pass
template <class Pool>
int lists_run(const IdSource& ids, const int64_t* offsets_host, int n_lists, int Q, int L, const float* packed, int64_t V, int D, int* status,
              void* workspace, size_t workspace_bytes, hipStream_t s, const float* edges, int nbins, const float* kn_mu, const float* kn_sigma, int kn_K,
              Pool pool) {
  if (!offsets_host || !packed || !status || !workspace) return CAPAMD_ERR_ARG;
  if (n_lists < 0 || Q < 1 || Q > kQT || L < 1 || L > 32768 || V < 1 || V > 0x7fffffffLL || capamd_packed_row_stride(D) < 0) return CAPAMD_ERR_ARG;
  if ((reinterpret_cast<uintptr_t>(workspace) & 15) != 0) return CAPAMD_ERR_ALIGN;
  const int64_t Vp = lists_vp(V);
  const size_t per_list = (size_t)Vp * 17 + kListQueryBytes;
  if (workspace_bytes < kListConstBytes) return CAPAMD_ERR_WORKSPACE;
  const size_t fit = (workspace_bytes - kListConstBytes) / per_list;
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
    // workspace: table [cap][Vp] x 16 B | byte maps [cap][Vp] | query images [cap] | query ids and norms [cap] | KNRM kernel constants
    char* ws = static_cast<char*>(workspace);
    float4* table = reinterpret_cast<float4*>(ws);
    uint8_t* flags = reinterpret_cast<uint8_t*>(ws + (size_t)cap * Vp * 16);
    float4* qimg = reinterpret_cast<float4*>(ws + (size_t)cap * Vp * 17);
    ListQuery* qmeta = reinterpret_cast<ListQuery*>(qimg + (size_t)cap * kQueryImage);
    float* kn_consts = kn_mu ? reinterpret_cast<float*>(qmeta + cap) : nullptr;
    ListsArgs a{ids, Q, L, packed, V, Vp, flags, table, status, nl, longest, edges, nbins, qimg, qmeta, kn_mu, kn_sigma, kn_K, kn_consts};
    lists_stamp(s);
    if (hipMemsetAsync(flags, 0, (size_t)nl * Vp, s) != hipSuccess) return CAPAMD_ERR_LAUNCH;
    lists_stamp(s);
    hipLaunchKernelGGL(lists_mark_kernel, list_doc_grid(nl, longest), dim3(256), 0, s, a, g);
    lists_stamp(s);
    const dim3 sg((unsigned)nl * 8, (unsigned)((Vp / kSimsIds + 7) / 8));
#define CAPAMD_SIMS(NV)                                                                                         \
  hipLaunchKernelGGL(lists_query_kernel<NV>, dim3(nl), dim3(128), 0, s, a, g);                                  \
  lists_stamp(s);                                                                                               \
  if (edges) hipLaunchKernelGGL((lists_sims_kernel<NV, true>), sg, dim3(256), 0, s, a, g);                      \
  else hipLaunchKernelGGL((lists_sims_kernel<NV, false>), sg, dim3(256), 0, s, a, g)
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
