// Shared front end of the KNRM / DRMM hot path for gfx950 (MI355X, wave64).
//
// What it computes is the reference's SimilarityMatrix (capreolus/reranker/common.py:143-182):
//   sim[q][j] = exact(q,j) + cos(q,j)
//   cos  = <E[q], E[d_j]> / ((|E[q]|+1e-9) * (|E[d_j]|+1e-9)),   0 where q<=0 or d_j<=0   (common.py:160-167)
//   exact= 1 where q == d_j < 0 (OOV ids are negative),           0 elsewhere               (common.py:155-158,179)
//
// How it computes it (the documented arithmetic order; oracle/interaction_oracle.c restates it
// bit for bit):
//   * the embedding table is re-laid out once ("packed"): row stride RS = 64*NV floats with
//     NV = ceil((D+1)/64); floats [D, RS-1) are 0 and float RS-1 holds den = |row|_2 + 1e-9f.
//     A row is then NV*4 aligned 64-byte pieces -> one 16-lane group fetches it with NV float4
//     loads per lane, 256 contiguous bytes per group per load (row start is 256-B aligned).
//   * a doc term is owned by one 16-lane group (a DPP "row").  Lane l holds floats
//     {(i*16+l)*4 .. +3 : i < NV} of the row and of each of the (up to) 4 query rows.
//     per-lane partial:  p = fma(d.x,q.x,p); p = fma(d.y,q.y,p); p = fma(d.z,q.z,p); p = fma(d.w,q.w,p)
//                        for i = 0..NV-1 in that order, starting from p = 0
//     group all-reduce:  p += p[l^1]; p += p[l^2]; p += p[half-mirror]; p += p[mirror]
//                        (== the balanced tree ((p0+p1)+(p2+p3)) + ... over the 16 lane partials;
//                         every lane ends with the same bits because fp add is commutative)
//     sim = p / (qden * dden)     IEEE fp32 multiply then correctly-rounded divide
//   * pads / OOV doc terms are never gathered: their sim is exactly 0 (or 1 for an OOV exact
//     match) and callers add their contribution in closed form.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace capamd {

constexpr int kGroup = 16;        // lanes per doc term (one DPP row)
constexpr int kThreads = 256;     // 4 waves per workgroup
constexpr int kGroupsPerWG = kThreads / kGroup;
constexpr int kQT = 4;            // query terms held in registers per pass
constexpr int kMaxNV = 5;         // D <= 319

// status word bits (device int32 the caller zeroes and reads back)
constexpr int kErrDocIdRange = 1;    // doc id >= V
constexpr int kErrQueryIdRange = 2;  // query id >= V
constexpr int kErrQueryOOV = 4;      // negative query id where the reference model cannot take one (DRMM.py:109)
constexpr int kErrListQuery = 32;    // whole-list entries: a pair's query (idf) row differs from its list's first pair's

__host__ __device__ inline int nv_for_dim(int D) { return (D + 1 + 63) / 64; }
__host__ __device__ inline int row_stride_for_dim(int D) { return 64 * nv_for_dim(D); }

// Where a batch's term ids come from: the extractor layout (int64 [B,Q] / [B,L], embedtext.py:146-147) or a
// device-resident candidate store (int32 tables uploaded once + per-pair row indices).
struct IdSource {
  const int64_t* q64;     // [B, Q]
  const int64_t* d64;     // [B, L]
  const int32_t* q32;     // [NQ, Q] query table   (indexed mode)
  const int32_t* d32;     // [ND, L] document table
  const int32_t* pair_q;  // [B] row of q32 for each pair
  const int32_t* pair_d;  // [B] row of d32 for each pair
};

struct PairIds {  // the id rows of one pair
  const int64_t* q64;
  const int64_t* d64;
  const int32_t* q32;
  const int32_t* d32;
  int qrow;  // row of the query in q32 / idf table (indexed mode), else the pair index
  __device__ __forceinline__ int64_t q(int t) const { return q32 ? (int64_t)q32[t] : q64[t]; }
  __device__ __forceinline__ int64_t d(int j) const { return d32 ? (int64_t)d32[j] : d64[j]; }
};

__device__ __forceinline__ PairIds pair_ids(const IdSource& s, int b, int Q, int L) {
  PairIds p;
  if (s.q32) {
    p.qrow = s.pair_q[b];
    p.q32 = s.q32 + (int64_t)p.qrow * Q;
    p.d32 = s.d32 + (int64_t)s.pair_d[b] * L;
    p.q64 = nullptr;
    p.d64 = nullptr;
  } else {
    p.qrow = b;
    p.q64 = s.q64 + (int64_t)b * Q;
    p.d64 = s.d64 + (int64_t)b * L;
    p.q32 = nullptr;
    p.d32 = nullptr;
  }
  return p;
}

// ---- phase 1 of the scoring kernels: the document's real terms, each DISTINCT id once, with its multiplicity ------------------------
// A document repeats its frequent terms (Zipf-distributed ids, 300 real terms: 59 % distinct; natural text is no different), and a
// repeated term has the same similarity to every query term - so its packed row is gathered ONCE and its contribution multiplied by
// its count (kernel pooling: count x K(sim); matching histogram: bin += count, still exact integers).  The distinct ids come out in
// the order of their FIRST occurrence in the document, which makes the list - and with it every floating-point sum over it - a
// function of the id row alone, whatever order the hash insertions happened to race in:
//   A  every real position inserts its id into an LDS hash set (linear probing, ds_cmpst) and atomic-mins its position into the slot
//   B  the position that holds a slot's minimum is the id's owner: owners are compacted in document order (ballot + popcount, as
//      before) into tok[]; the slot remembers the owner's dense index
//   C  every real position adds 1 to mult[dense index of its slot]
// Documents longer than kDedupMaxL positions (the API allows 32768) skip the hash and list every real term with multiplicity 1.
constexpr int kHashSlots = 1024;
constexpr int kDedupMaxL = 896;               // <= 4 positions per thread, load factor <= 0.875
constexpr int kDedupPos = (kDedupMaxL + 255) / 256;

struct TermList {
  int n_unique;   // entries of tok[] / mult[]
  int n_real;     // real (id > 0) positions, repeats included
  int n_oov;      // negative ids
};

// The hash part of the distinct-term pass: tok[k] = the k-th distinct real term in order of first occurrence, first[slot] = -1 - k for
// every used hash slot, and per thread the (term, slot) of its up to kDedupPos positions (position j = it * 256 + tid).  L <= kDedupMaxL.
// key, first: [kHashSlots] ints; wave_cnt: [48] ints.  Five barriers in all for a caller that adds one counting pass: the ids of a
// thread's positions are requested together, and the owners of all four position blocks are compacted behind ONE pair of barriers.
struct DistinctCore {
  int id_r[kDedupPos], slot_r[kDedupPos];
  TermList r;
};
__device__ __forceinline__ void distinct_core(const PairIds& ids, int L, int64_t V, int* status, int* tok, int* key, int* first, int* wave_cnt,
                                              DistinctCore& c) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < kHashSlots; i += kThreads) { key[i] = 0; first[i] = 0x7fffffff; }
  int64_t did[kDedupPos];
  // (one branch on the id layout around all of a thread's loads, not one per load: inside `ids.d(j)` the int32 side's sign extension
  // made hipcc wait for each load before it issued the next - four memory round trips per pair on the resident-store path)
  if (ids.d32) {
    int v[kDedupPos];
#pragma unroll
    for (int it = 0; it < kDedupPos; ++it) {
      const int j = it * kThreads + tid;
      v[it] = ids.d32[j < L ? j : L - 1];          // (unconditional: a load under its own exec mask is waited for at the join)
    }
#pragma unroll
    for (int it = 0; it < kDedupPos; ++it) did[it] = (it * kThreads + tid < L) ? (int64_t)v[it] : 0;
  } else {
#pragma unroll
    for (int it = 0; it < kDedupPos; ++it) {
      const int j = it * kThreads + tid;
      did[it] = (j < L) ? ids.d64[j] : 0;
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < kDedupPos; ++it) {            // A: hash insert, first position per term
    const int j = it * kThreads + tid;
    if (did[it] >= V) { atomicOr(status, kErrDocIdRange); did[it] = 0; }
    c.id_r[it] = did[it] < 0 ? -1 : (int)did[it];
    c.slot_r[it] = -1;
    if (did[it] > 0) {
      unsigned h = ((unsigned)did[it] * 2654435761u) >> 22;
      for (;;) {
        const int old = atomicCAS(&key[h], 0, (int)did[it]);
        if (old == 0 || old == (int)did[it]) break;
        h = (h + 1) & (kHashSlots - 1);
      }
      c.slot_r[it] = (int)h;
      atomicMin(&first[h], j);
    }
  }
  __syncthreads();
  unsigned long long m_own[kDedupPos];
  bool owner[kDedupPos];
#pragma unroll
  for (int it = 0; it < kDedupPos; ++it) {            // B1: who owns a term (its first position), per-wave counts
    const int j = it * kThreads + tid;
    const bool real = c.slot_r[it] >= 0;
    owner[it] = real && first[c.slot_r[it]] == j;
    m_own[it] = __ballot(owner[it]);
    const unsigned long long mr = __ballot(real), mo = __ballot(c.id_r[it] < 0);
    if (lane == 0) {
      wave_cnt[(it * 3 + 0) * 4 + wave] = __popcll(m_own[it]);
      wave_cnt[(it * 3 + 1) * 4 + wave] = __popcll(mo);
      wave_cnt[(it * 3 + 2) * 4 + wave] = __popcll(mr);
    }
  }
  __syncthreads();   // (every comparison against first[] is done before any slot is overwritten below)
  c.r = TermList{0, 0, 0};
#pragma unroll
  for (int it = 0; it < kDedupPos; ++it) {            // B2: owners compacted in document order (position block major, then thread)
    const int* w0 = wave_cnt + (it * 3) * 4;
    int off = c.r.n_unique;
    for (int w = 0; w < wave; ++w) off += w0[w];
    if (owner[it]) {
      const int k = off + __popcll(m_own[it] & ((1ull << lane) - 1ull));
      tok[k] = c.id_r[it];
      first[c.slot_r[it]] = -1 - k;       // negative: never equal to a position
    }
    c.r.n_unique += w0[0] + w0[1] + w0[2] + w0[3];
    c.r.n_oov += w0[4] + w0[5] + w0[6] + w0[7];
    c.r.n_real += w0[8] + w0[9] + w0[10] + w0[11];
  }
  __syncthreads();
}

// tok, mult: [L rounded up to 4] ints; key, first: [kHashSlots] ints; wave_cnt: [48] ints.  All LDS.  256 threads.
__device__ __forceinline__ TermList distinct_terms(const PairIds& ids, int L, int64_t V, int* status, int* tok, int* mult, int* key, int* first,
                                                   int* wave_cnt) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  TermList r{0, 0, 0};
  if (L > kDedupMaxL) {   // no hash: every real term, multiplicity 1
    for (int base = 0; base < L; base += kThreads) {
      const int j = base + tid;
      int64_t did = (j < L) ? ids.d(j) : 0;
      if (did >= V) { atomicOr(status, kErrDocIdRange); did = 0; }
      const bool real = did > 0;
      const unsigned long long m = __ballot(real), mo = __ballot(did < 0);
      if (lane == 0) { wave_cnt[wave] = __popcll(m); wave_cnt[4 + wave] = __popcll(mo); }
      __syncthreads();
      int off = r.n_unique;
      for (int w = 0; w < wave; ++w) off += wave_cnt[w];
      if (real) { const int k = off + __popcll(m & ((1ull << lane) - 1ull)); tok[k] = (int)did; mult[k] = 1; }
      r.n_unique += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
      r.n_oov += wave_cnt[4] + wave_cnt[5] + wave_cnt[6] + wave_cnt[7];
      __syncthreads();
    }
    r.n_real = r.n_unique;
    return r;
  }
  for (int i = tid; i < L; i += kThreads) mult[i] = 0;
  DistinctCore c;
  distinct_core(ids, L, V, status, tok, key, first, wave_cnt, c);
#pragma unroll
  for (int it = 0; it < kDedupPos; ++it)              // occurrences per term
    if (c.slot_r[it] >= 0) atomicAdd(&mult[-1 - first[c.slot_r[it]]], 1);
  __syncthreads();
  return c.r;
}

// distinct_terms for kernels that need the POSITIONS of a term rather than its count (PACRR writes a similarity to every position of the
// pair's matrix): tok[k] = the k-th distinct real term, plist[start[k] .. start[k + 1]) = its positions (any order).
// tok: [L rounded up to 4] ints; start: [L + 1] and plist: [L] uint16 (L <= 65535); wave_cnt: [48] ints; scratch: `scratch_ints` ints that
// may alias anything dead during the call (hash keys | first positions | per-term counters).  Without room for the hash, or beyond
// kDedupMaxL positions, every real position is listed as its own term.  256 threads; ends on a barrier.
__device__ __forceinline__ TermList distinct_terms_positions(const PairIds& ids, int L, int64_t V, int* status, int* tok, unsigned short* start,
                                                             unsigned short* plist, int* scratch, int scratch_ints, int* wave_cnt) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  TermList r{0, 0, 0};
  const int tok_cap = (L + 3) & ~3;
  if (L > kDedupMaxL || scratch_ints < 2 * kHashSlots + tok_cap) {
    for (int base = 0; base < L; base += kThreads) {
      const int j = base + tid;
      int64_t did = (j < L) ? ids.d(j) : 0;
      if (did >= V) { atomicOr(status, kErrDocIdRange); did = 0; }
      const bool real = did > 0;
      const unsigned long long m = __ballot(real), mo = __ballot(did < 0);
      if (lane == 0) { wave_cnt[wave] = __popcll(m); wave_cnt[4 + wave] = __popcll(mo); }
      __syncthreads();
      int off = r.n_unique;
      for (int w = 0; w < wave; ++w) off += wave_cnt[w];
      if (real) {
        const int k = off + __popcll(m & ((1ull << lane) - 1ull));
        tok[k] = (int)did;
        start[k] = (unsigned short)k;
        plist[k] = (unsigned short)j;
      }
      r.n_unique += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
      r.n_oov += wave_cnt[4] + wave_cnt[5] + wave_cnt[6] + wave_cnt[7];
      __syncthreads();
    }
    if (tid == 0) start[r.n_unique] = (unsigned short)r.n_unique;
    r.n_real = r.n_unique;
    __syncthreads();
    return r;
  }
  int *key = scratch, *first = scratch + kHashSlots, *cnt = scratch + 2 * kHashSlots;
  for (int i = tid; i < L; i += kThreads) cnt[i] = 0;
  DistinctCore c;
  distinct_core(ids, L, V, status, tok, key, first, wave_cnt, c);
  r = c.r;
#pragma unroll
  for (int it = 0; it < kDedupPos; ++it)              // occurrences per term
    if (c.slot_r[it] >= 0) atomicAdd(&cnt[-1 - first[c.slot_r[it]]], 1);
  __syncthreads();
  if (wave == 0) {                                    // exclusive prefix sum of the counts: one wave, a run of terms per lane
    const int n = r.n_unique, chunk = (n + 63) >> 6, k0 = lane * chunk, k1 = min(n, k0 + chunk);
    int sum = 0;
    for (int k = k0; k < k1; ++k) sum += cnt[k];
    int incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(incl, o, 64);
      if (lane >= o) incl += up;
    }
    int run = incl - sum;
    for (int k = k0; k < k1; ++k) {
      start[k] = (unsigned short)run;
      run += cnt[k];
    }
    if (lane == 63) start[n] = (unsigned short)incl;
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < kDedupPos; ++it)              // D: positions grouped by term
    if (c.slot_r[it] >= 0) {
      const int k = -1 - first[c.slot_r[it]];
      plist[start[k] + atomicSub(&cnt[k], 1) - 1] = (unsigned short)(it * kThreads + tid);
    }
  __syncthreads();
  return r;
}

// Dynamic LDS of a one-pair-per-workgroup kernel: refused (CAPAMD_ERR_ARG = 1) beyond the 160 KiB a gfx950 workgroup can have, and the
// kernel's limit raised where it is above the 64 KiB default.  (The term list and its multiplicities are 8 bytes per document
// position: with the kernels' fixed parts that puts the longest document at ~19,000 positions; the distinct-term hash - 8 KiB -
// exists only up to kDedupMaxL positions.)
constexpr size_t kMaxLds = 160 * 1024;
__host__ inline size_t dedup_hash_bytes(int L) { return L <= kDedupMaxL ? (size_t)2 * kHashSlots * 4 : 0; }
template <class K>
__host__ inline int lds_budget(K kernel, size_t smem) {
  if (smem > kMaxLds) return 1;
  if (smem > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return 3;
  return 0;
}

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// all-reduce over the 16 lanes of a DPP row; order documented in the header comment.
__device__ __forceinline__ float group_allreduce(float v) {
  v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]  : lane ^ 1
  v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]  : lane ^ 2
  v += dpp_mov<0x141>(v);  // row_half_mirror      : other quad of the 8-lane half
  v += dpp_mov<0x140>(v);  // row_mirror           : other half of the row
  return v;
}

// Reductions over the 64 lanes of a wave, every lane gets the result: DPP all-reduce inside each 16-lane row, then the four row values by
// v_readlane.  No LDS-pipe permutes: a `__shfl_xor` butterfly is six ds_bpermute round trips (~100 cycles each), and the per-pair tails
// of these kernels (feed-forward nets, top-k merges) are serial chains of such reductions during which the workgroup requests no rows.
// Insertion into a descending sorted list in registers: new top[i] = median(top[i-1], top[i], v) - the element above if v passed it, v if
// it lands here, the old one otherwise - one v_med3_f32 per element and no carried value (a compare-exchange chain is a v_max and a v_min
// per element, each waiting for the one before).
template <int K>
__device__ __forceinline__ void sorted_insert(float (&top)[K], float v) {
#pragma unroll
  for (int i = K - 1; i >= 1; --i) top[i] = __builtin_amdgcn_fmed3f(top[i - 1], top[i], v);
  top[0] = fmaxf(top[0], v);
}

// `unfused`: the value as computed - a product handed to a reduction is not contracted into the reduction's first add (hipcc's default
// -ffp-contract=fast would, or would not, depending on the code around it: two spellings of one tail must round alike).
__device__ __forceinline__ float unfused(float v) {
  asm("" : "+v"(v));
  return v;
}
__device__ __forceinline__ float wave_allreduce_sum(float v) {
  v = group_allreduce(unfused(v));
  const int bits = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(bits, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(bits, 16)),
              r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(bits, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(bits, 48));
  return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float wave_allreduce_max(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v));
  v = fmaxf(v, dpp_mov<0x4E>(v));
  v = fmaxf(v, dpp_mov<0x141>(v));
  v = fmaxf(v, dpp_mov<0x140>(v));
  const int bits = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(bits, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(bits, 16)),
              r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(bits, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(bits, 48));
  return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

template <int NV>
struct RowRegs {
  float4 v[NV];
};

// Loads this lane's slice of packed row `row` (row must be in [0, V)).
template <int NV>
__device__ __forceinline__ void load_row(const float* __restrict__ packed, int64_t row, int lane16, RowRegs<NV>& r) {
  const float4* p = reinterpret_cast<const float4*>(packed + row * (int64_t)(64 * NV)) + lane16;
#pragma unroll
  for (int i = 0; i < NV; ++i) r.v[i] = p[i * 16];
}

// den of a row lives in the last float of the row = lane 15's last float4 .w
template <int NV>
__device__ __forceinline__ float row_den(const RowRegs<NV>& r) {
  return __shfl(r.v[NV - 1].w, 15, 16);
}

template <int NV>
__device__ __forceinline__ float lane_dot(const RowRegs<NV>& d, const RowRegs<NV>& q) {
  float p = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    p = __builtin_fmaf(d.v[i].x, q.v[i].x, p);
    p = __builtin_fmaf(d.v[i].y, q.v[i].y, p);
    p = __builtin_fmaf(d.v[i].z, q.v[i].z, p);
    p = __builtin_fmaf(d.v[i].w, q.v[i].w, p);
  }
  return p;
}

// Query side of one pass: up to kQT query terms, each lane holding its slice of every row.
// Lane l of a group "owns" query term (l & 3) of the pass: it finishes that term's similarity.
template <int NV>
struct QueryPass {
  RowRegs<NV> row[kQT];
  int id[kQT];     // original ids (0 = pad, <0 = OOV); terms beyond Q are 0
  float den_my;    // |E[q]|+1e-9 of the term this lane owns
  int id_my;       // id of the term this lane owns
};

template <int NV>
__device__ __forceinline__ void load_query_pass(const float* __restrict__ packed, const PairIds& ids,
                                                int Q, int q0, int64_t V, int lane16, QueryPass<NV>& qp, int* status) {
  const int myq = lane16 & 3;
  qp.den_my = 1e-9f;
  qp.id_my = 0;
#pragma unroll
  for (int t = 0; t < kQT; ++t) {
    int64_t id = (q0 + t < Q) ? ids.q(q0 + t) : 0;
    if (id >= V) {
      if (status) atomicOr(status, kErrQueryIdRange);
      id = 0;
    }
    qp.id[t] = (int)id;
    load_row<NV>(packed, id > 0 ? id : 0, lane16, qp.row[t]);
    const float den = row_den<NV>(qp.row[t]);
    if (myq == t) {
      qp.den_my = den;
      qp.id_my = (int)id;
    }
    if (lane16 == 15) qp.row[t].v[NV - 1].w = 0.f;  // keep the den slot out of the dot product
  }
}

// Similarity of one gathered doc row (id > 0) against the query term this lane owns.  Lanes
// l, l+4, l+8, l+12 of a group return the same bits.
template <int NV>
__device__ __forceinline__ float row_sim_my(const RowRegs<NV>& d, const QueryPass<NV>& qp, int lane16) {
  const float dden = row_den<NV>(d);
  float p[kQT];
#pragma unroll
  for (int t = 0; t < kQT; ++t) p[t] = group_allreduce(lane_dot<NV>(d, qp.row[t]));
  const int myq = lane16 & 3;
  const float pm = myq == 0 ? p[0] : myq == 1 ? p[1] : myq == 2 ? p[2] : p[3];
  const float s = pm / (qp.den_my * dden);
  return qp.id_my > 0 ? s : 0.f;
}

// ---- multi-term form used by the scoring kernels -------------------------------------------------
// U document rows are gathered per group per iteration (U*NV float4 loads in flight per lane).  The
// query rows come either from registers (QueryPass) or, when QLDS, from an LDS copy shared by the
// workgroup ([kQT][NV*16] float4, den slot zeroed) which frees 16*NV VGPRs per wave for more rows in
// flight; each query chunk read from LDS is reused for all U rows.  The fma order per (row, query
// term) is unchanged (i ascending; x, y, z, w), so results are bit-identical across variants.
// (Tried in round 2: the LDS copy interleaved in term PAIRS so that two terms' partial sums advance with one v_pk_fma_f32 - 40
// instead of 80 FMA instructions per 4 rows, bit-identical sums.  Same-box A/B: KNRM headline 49.9 = 49.9 M pairs/s, its HBM-bound
// leg 0.797 against 0.817; DRMM +2 %, DRMM-TKS -2 %, PACRR -2 %: the kernels are not bound by VALU issue.  Not kept.)
template <int NV, int U, bool QLDS>
__device__ __forceinline__ void rows_sim_my(const RowRegs<NV> (&d)[U], const QueryPass<NV>& qp, const float4* qlds, int lane16,
                                            float (&sim)[U]) {
  float p[U][kQT];
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int t = 0; t < kQT; ++t) p[u][t] = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int t = 0; t < kQT; ++t) {
      const float4 q = QLDS ? qlds[(t * NV + i) * 16 + lane16] : qp.row[t].v[i];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float a = p[u][t];
        a = __builtin_fmaf(d[u].v[i].x, q.x, a);
        a = __builtin_fmaf(d[u].v[i].y, q.y, a);
        a = __builtin_fmaf(d[u].v[i].z, q.z, a);
        a = __builtin_fmaf(d[u].v[i].w, q.w, a);
        p[u][t] = a;
      }
      // keep at most one query chunk in flight ahead of its use: otherwise the scheduler front-loads all
      // kQT*NV LDS reads and the 16*NV VGPRs the LDS copy was meant to free are back
      if (QLDS && (t & 1)) __builtin_amdgcn_sched_barrier(0);
    }
  const int myq = lane16 & 3;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const float dden = row_den<NV>(d[u]);
    float r[kQT];
#pragma unroll
    for (int t = 0; t < kQT; ++t) r[t] = group_allreduce(p[u][t]);
    const float pm = myq == 0 ? r[0] : myq == 1 ? r[1] : myq == 2 ? r[2] : r[3];
    const float s = pm / (qp.den_my * dden);
    sim[u] = qp.id_my > 0 ? s : 0.f;
  }
}

// The two halves of rows_sim_my<NV, 1, true> for callers that put something between them (the streaming kernels request the group's next
// row as soon as the dot products are formed): same fma order, same reduction, same divide - bit-identical similarities.
template <int NV>
__device__ __forceinline__ void rows_dot(const RowRegs<NV>& d, const float4* qlds, int lane16, float (&p)[kQT]) {
#pragma unroll
  for (int t = 0; t < kQT; ++t) p[t] = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int t = 0; t < kQT; ++t) {
      const float4 q = qlds[(t * NV + i) * 16 + lane16];
      float a = p[t];
      a = __builtin_fmaf(d.v[i].x, q.x, a);
      a = __builtin_fmaf(d.v[i].y, q.y, a);
      a = __builtin_fmaf(d.v[i].z, q.z, a);
      a = __builtin_fmaf(d.v[i].w, q.w, a);
      p[t] = a;
      if (t & 1) __builtin_amdgcn_sched_barrier(0);   // (as in rows_sim_my: at most one query chunk ahead of its use)
    }
}
// two rows against one pass over the LDS query copy (each query chunk read once for both rows; per row the same fma order as rows_dot)
template <int NV>
__device__ __forceinline__ void rows_dot2(const RowRegs<NV>& d0, const RowRegs<NV>& d1, const float4* qlds, int lane16, float (&p0)[kQT], float (&p1)[kQT]) {
#pragma unroll
  for (int t = 0; t < kQT; ++t) p0[t] = p1[t] = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int t = 0; t < kQT; ++t) {
      const float4 q = qlds[(t * NV + i) * 16 + lane16];
      float a = p0[t], b = p1[t];
      a = __builtin_fmaf(d0.v[i].x, q.x, a);
      b = __builtin_fmaf(d1.v[i].x, q.x, b);
      a = __builtin_fmaf(d0.v[i].y, q.y, a);
      b = __builtin_fmaf(d1.v[i].y, q.y, b);
      a = __builtin_fmaf(d0.v[i].z, q.z, a);
      b = __builtin_fmaf(d1.v[i].z, q.z, b);
      a = __builtin_fmaf(d0.v[i].w, q.w, a);
      b = __builtin_fmaf(d1.v[i].w, q.w, b);
      p0[t] = a;
      p1[t] = b;
      if (t & 1) {
        // both rows' chains are tied to this point: left alone, hipcc runs row 0's chain through all the chunks first and keeps the
        // whole query copy (NV x kQT float4) in registers for row 1's
        asm volatile("" : "+v"(p0[t - 1]), "+v"(p0[t]), "+v"(p1[t - 1]), "+v"(p1[t]));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
}
// rows_dot / rows_dot2 on packed fp32 (v_pk_fma_f32: two fmas per lane and instruction): the two lanes of an instruction are two QUERY
// TERMS' chains, each with exactly the operands and the order of rows_dot - bit-identical dot products at half the fma instructions.
// The LDS copy is the PAIRED layout of load_query_pass_lds: per pair of terms (2P, 2P + 1), chunk i and half h a float4
// {q[2P].c, q[2P+1].c, q[2P].c', q[2P+1].c'} with (c, c') = (x, y) for h = 0 and (z, w) for h = 1, at ((P * NV + i) * 2 + h) * 16 + lane16.
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int NV>
__device__ __forceinline__ void rows_dot_pk(const RowRegs<NV>& d, const float4* qlds, int lane16, float (&p)[kQT]) {
  f32x2 acc[2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
  for (int i = 0; i < NV; ++i) {
#pragma unroll
    for (int P = 0; P < 2; ++P) {
      const float4 qa = qlds[((P * NV + i) * 2 + 0) * 16 + lane16], qb = qlds[((P * NV + i) * 2 + 1) * 16 + lane16];
      f32x2 a = acc[P];
      a = __builtin_elementwise_fma((f32x2){d.v[i].x, d.v[i].x}, (f32x2){qa.x, qa.y}, a);
      a = __builtin_elementwise_fma((f32x2){d.v[i].y, d.v[i].y}, (f32x2){qa.z, qa.w}, a);
      a = __builtin_elementwise_fma((f32x2){d.v[i].z, d.v[i].z}, (f32x2){qb.x, qb.y}, a);
      a = __builtin_elementwise_fma((f32x2){d.v[i].w, d.v[i].w}, (f32x2){qb.z, qb.w}, a);
      acc[P] = a;
    }
    asm volatile("" : "+v"(acc[0]), "+v"(acc[1]));      // (one chunk at a time: see rows_dot2)
    __builtin_amdgcn_sched_barrier(0);
  }
  p[0] = acc[0].x; p[1] = acc[0].y; p[2] = acc[1].x; p[3] = acc[1].y;
}
// NP = 1: only the first pair of query terms (terms 2 and 3 are not real - pads of the fixed-length query row: their similarities are 0 whatever
// their dot products are, sim_from_dots) - half the fmas and half the LDS reads; the partials of the second pair are returned as 0
template <int NV, int NP = 2>
__device__ __forceinline__ void rows_dot2_pk(const RowRegs<NV>& d0, const RowRegs<NV>& d1, const float4* qlds, int lane16, float (&p0)[kQT],
                                             float (&p1)[kQT]) {
  f32x2 acc0[2] = {{0.f, 0.f}, {0.f, 0.f}}, acc1[2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
  for (int i = 0; i < NV; ++i) {
#pragma unroll
    for (int P = 0; P < NP; ++P) {
      const float4 qa = qlds[((P * NV + i) * 2 + 0) * 16 + lane16], qb = qlds[((P * NV + i) * 2 + 1) * 16 + lane16];
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
    if (NP == 2) asm volatile("" : "+v"(acc0[0]), "+v"(acc0[1]), "+v"(acc1[0]), "+v"(acc1[1]));
    else asm volatile("" : "+v"(acc0[0]), "+v"(acc1[0]));
    __builtin_amdgcn_sched_barrier(0);
  }
  p0[0] = acc0[0].x; p0[1] = acc0[0].y; p0[2] = acc0[1].x; p0[3] = acc0[1].y;
  p1[0] = acc1[0].x; p1[1] = acc1[0].y; p1[2] = acc1[1].x; p1[3] = acc1[1].y;
}
// three rows per trip: the LDS query copy read once for three rows (A/B builds: -DCAPAMD_LISTS_SIMS_ROWS=3)
template <int NV>
__device__ __forceinline__ void rows_dot3_pk(const RowRegs<NV>& d0, const RowRegs<NV>& d1, const RowRegs<NV>& d2, const float4* qlds, int lane16,
                                             float (&p0)[kQT], float (&p1)[kQT], float (&p2)[kQT]) {
  f32x2 acc0[2] = {{0.f, 0.f}, {0.f, 0.f}}, acc1[2] = {{0.f, 0.f}, {0.f, 0.f}}, acc2[2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
  for (int i = 0; i < NV; ++i) {
#pragma unroll
    for (int P = 0; P < 2; ++P) {
      const float4 qa = qlds[((P * NV + i) * 2 + 0) * 16 + lane16], qb = qlds[((P * NV + i) * 2 + 1) * 16 + lane16];
      f32x2 a = acc0[P], b = acc1[P], c = acc2[P];
      a = __builtin_elementwise_fma((f32x2){d0.v[i].x, d0.v[i].x}, (f32x2){qa.x, qa.y}, a);
      b = __builtin_elementwise_fma((f32x2){d1.v[i].x, d1.v[i].x}, (f32x2){qa.x, qa.y}, b);
      c = __builtin_elementwise_fma((f32x2){d2.v[i].x, d2.v[i].x}, (f32x2){qa.x, qa.y}, c);
      a = __builtin_elementwise_fma((f32x2){d0.v[i].y, d0.v[i].y}, (f32x2){qa.z, qa.w}, a);
      b = __builtin_elementwise_fma((f32x2){d1.v[i].y, d1.v[i].y}, (f32x2){qa.z, qa.w}, b);
      c = __builtin_elementwise_fma((f32x2){d2.v[i].y, d2.v[i].y}, (f32x2){qa.z, qa.w}, c);
      a = __builtin_elementwise_fma((f32x2){d0.v[i].z, d0.v[i].z}, (f32x2){qb.x, qb.y}, a);
      b = __builtin_elementwise_fma((f32x2){d1.v[i].z, d1.v[i].z}, (f32x2){qb.x, qb.y}, b);
      c = __builtin_elementwise_fma((f32x2){d2.v[i].z, d2.v[i].z}, (f32x2){qb.x, qb.y}, c);
      a = __builtin_elementwise_fma((f32x2){d0.v[i].w, d0.v[i].w}, (f32x2){qb.z, qb.w}, a);
      b = __builtin_elementwise_fma((f32x2){d1.v[i].w, d1.v[i].w}, (f32x2){qb.z, qb.w}, b);
      c = __builtin_elementwise_fma((f32x2){d2.v[i].w, d2.v[i].w}, (f32x2){qb.z, qb.w}, c);
      acc0[P] = a;
      acc1[P] = b;
      acc2[P] = c;
    }
    asm volatile("" : "+v"(acc0[0]), "+v"(acc0[1]), "+v"(acc1[0]), "+v"(acc1[1]), "+v"(acc2[0]), "+v"(acc2[1]));
    __builtin_amdgcn_sched_barrier(0);
  }
  p0[0] = acc0[0].x; p0[1] = acc0[0].y; p0[2] = acc0[1].x; p0[3] = acc0[1].y;
  p1[0] = acc1[0].x; p1[1] = acc1[0].y; p1[2] = acc1[1].x; p1[3] = acc1[1].y;
  p2[0] = acc2[0].x; p2[1] = acc2[0].y; p2[2] = acc2[1].x; p2[3] = acc2[1].y;
}
// NT: the terms whose dot products were computed (the others' similarities are 0: their ids are not real)
template <int NV, int NT = kQT>
__device__ __forceinline__ float sim_from_dots(const float (&p)[kQT], float dden, const QueryPass<NV>& qp, int lane16) {
  float r[kQT];
#pragma unroll
  for (int t = 0; t < kQT; ++t) r[t] = t < NT ? group_allreduce(p[t]) : 0.f;
  const int myq = lane16 & 3;
  const float pm = myq == 0 ? r[0] : myq == 1 ? r[1] : myq == 2 ? r[2] : r[3];
  const float s = pm / (qp.den_my * dden);
  return qp.id_my > 0 ? s : 0.f;
}

// query pass whose rows live in LDS: only ids / den of the owned term stay in registers.  The loads are batched - the kQT ids together
// (one branch on the id layout around them, indices clamped instead of predicated), then the kQT rows' chunks and norms together, then
// the LDS writes: written term by term, hipcc waited for each id before it asked for that term's row and for each row before the next
// id - eight memory round trips per pair in series where two do.  nthreads >= NV * 16 (one chunk per thread and row).
// PAIRED: the LDS copy interleaves the query terms in pairs for rows_dot_pk (see there) instead of term after term.
template <int NV, bool PAIRED = false>
__device__ __forceinline__ void load_query_pass_lds(const float* __restrict__ packed, const PairIds& ids, int Q,
                                                    int q0, int64_t V, int tid, int nthreads, int lane16, float4* qlds,
                                                    QueryPass<NV>& qp, int* status) {
  (void)nthreads;
  const int myq = lane16 & 3;
  int64_t id[kQT];
  if (ids.q32) {
    int v[kQT];
#pragma unroll
    for (int t = 0; t < kQT; ++t) v[t] = ids.q32[q0 + t < Q ? q0 + t : Q - 1];
#pragma unroll
    for (int t = 0; t < kQT; ++t) id[t] = (q0 + t < Q) ? (int64_t)v[t] : 0;
  } else {
#pragma unroll
    for (int t = 0; t < kQT; ++t) id[t] = ids.q64[q0 + t < Q ? q0 + t : Q - 1];
#pragma unroll
    for (int t = 0; t < kQT; ++t) id[t] = (q0 + t < Q) ? id[t] : 0;
  }
  bool bad = false;
#pragma unroll
  for (int t = 0; t < kQT; ++t)
    if (id[t] >= V) { bad = true; id[t] = 0; }
  if (bad && status) atomicOr(status, kErrQueryIdRange);
  float den[kQT];
  float4 v[kQT];
  const int c = tid < NV * 16 ? tid : 0;
#pragma unroll
  for (int t = 0; t < kQT; ++t) {
    const float* row = packed + (id[t] > 0 ? id[t] : 0) * (int64_t)(64 * NV);
    den[t] = row[64 * NV - 1];
    v[t] = reinterpret_cast<const float4*>(row)[c];
  }
  qp.den_my = 1e-9f;
  qp.id_my = 0;
#pragma unroll
  for (int t = 0; t < kQT; ++t) {
    qp.id[t] = (int)id[t];
    if (myq == t) {
      qp.den_my = den[t];
      qp.id_my = (int)id[t];
    }
    if (tid == NV * 16 - 1) v[t].w = 0.f;  // keep the den slot out of the dot product
    if (!PAIRED && tid < NV * 16) qlds[t * NV * 16 + tid] = v[t];
  }
  if (PAIRED && tid < NV * 16) {
    static_assert(kQT == 4, "two pairs of query terms");
    const int i = tid >> 4, ln = tid & 15;
#pragma unroll
    for (int P = 0; P < 2; ++P) {
      const float4 a = v[2 * P], b = v[2 * P + 1];
      qlds[((P * NV + i) * 2 + 0) * 16 + ln] = make_float4(a.x, b.x, a.y, b.y);
      qlds[((P * NV + i) * 2 + 1) * 16 + ln] = make_float4(a.z, b.z, a.w, b.w);
    }
  }
}

}  // namespace capamd
