// Streaming form of the KNRM / DRMM scoring kernels for gfx950: persistent workgroups of FIVE waves - four gathering waves that never
// leave the gather loop, and one list wave that (a) turns the NEXT pair's id row into its distinct-term list and stages its query rows,
// and (b) finishes the PREVIOUS pair (cross-wave reduction, closed-form pad / OOV terms, the model's per-pair tail).
//
// Why: in the one-pair-per-workgroup kernels (knrm.hip: knrm_forward_kernel, drmm.hip: drmm_forward_kernel) a workgroup requests no rows
// while it lists its terms or runs its tail - 23 % of its life on the benchmark's candidate lists, which is exactly the distance between
// the 12.3 TB/s of rows those kernels request and the 15.5 TB/s a gather-only kernel gets from the same request stream.
//   * pairs are handed out by a ticket counter (the caller's 4-byte workspace word, zeroed by the launch code): documents differ 40x in
//     length, a static partition of 64,000 pairs over 1,536 workgroups would leave the tail of the launch to the unluckiest one;
//   * everything the two roles exchange is double-buffered in LDS and ONE s_barrier per pair separates the generations:
//       between barriers i-1 and i   gatherers: pair i from list[i&1], qrows[i&1] -> the model's partial results [i&1]
//                                    list wave: finish pair i-1 from partial[(i-1)&1]; build pair i+1 into list / qrows / meta[(i+1)&1]
//   * the list wave is a single wave, so the distinct-term pass needs no barrier at all (LDS operations of one wave complete in order);
//     the hash is ONE word per slot, (id << 10 | first position): a compare-and-swap claims an empty slot, an atomic minimum on a slot
//     that already belongs to the id keeps the first position - for equal ids the order of the words is the order of the positions.
//     That needs id < 2^22 (and L <= kDedupMaxL); bigger tables take the one-pair-per-workgroup kernels;
//   * list entries are (id | multiplicity << 22): one LDS read per gathered row;
//   * the four query rows go from global memory straight into LDS (global_load_lds_dwordx4) while the list wave hashes: no registers
//     held across the pass, no memory round trip on anybody's critical path;
//   * a gathering wave that is done with pair i asks for the first row of pair i+1 BEFORE the barrier when the list wave has already
//     published it (a generation word next to the list): the request then travels under the barrier and the other waves' last rows
//     instead of starting a cold pipeline after it.  Purely a hint: nothing is consumed before the barrier.
// The list comes out in the same order (first occurrence) as interaction.h: distinct_terms, the gather arithmetic is the same code
// (rows_dot / sim_from_dots = rows_sim_my): similarities are bit-identical to the one-pair-per-workgroup kernels.
//
// A model plugs in as a policy struct M (knrm.hip: KnrmStream, drmm.hip: DrmmStream):
//   typename M::Args                          kernel argument block (by value)
//   static size_t M::lds_bytes(const Args&)   LDS the model needs after the common part (16-byte aligned start)
//   M::list_init(a, lds, lane)                list wave, once: constants into LDS
//   M::prepare(a, lds, buf, lane)             list wave, before pair's gather starts: clear the model's buffer `buf`
//   M::finish(a, src, lds, buf, meta, lane)   list wave: pair meta->pair from buffer `buf` -> outputs
//   typename M::Gather                        per-lane state of a gathering wave
//   M::gather_init(a, gs, lane16)             once
//   M::pair_begin(a, gs, meta, lane16)        per pair (returns the lane's QueryPass den / id through gs.qp)
//   M::row(a, gs, x, entry, lds, buf, lane16) one gathered row: x = similarity of the lane's query term, entry = id | mult << 22
//   M::pair_end(a, gs, lds, buf, wave, lane)  per pair: the wave's results into buffer `buf`
#pragma once
#include "interaction.h"
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

#ifndef CAPAMD_STREAM_PREFETCH
#define CAPAMD_STREAM_PREFETCH 1   // A/B builds: 0 = no request for the next pair's first row before the barrier
#endif
#ifndef CAPAMD_STREAM_NT_IDS
#define CAPAMD_STREAM_NT_IDS 1     // the id rows are read with the non-temporal hint (streamed once, 410 MB per 64,000 pairs: +0.7-1.4 % on the headline leg); A/B builds: 0
#endif
#ifndef CAPAMD_STREAM_NT_ROWS
#define CAPAMD_STREAM_NT_ROWS 0    // A/B builds: T > 0 = table rows with id >= T are requested with the non-temporal hint
#endif

namespace capamd {

constexpr int kStreamThreads = 320;
constexpr int kStreamBlocks = (kDedupMaxL + 63) / 64;   // position blocks of 64 the list wave walks
constexpr unsigned kIdBits = 22, kIdMask = (1u << kIdBits) - 1u;
constexpr unsigned kHashEmpty = 0xffffffffu;

struct StreamSrc {      // where the pairs come from (what every model shares)
  IdSource ids;
  int B, Q, L;
  const float* packed;
  int64_t V;
  int* status;
};

struct StreamMeta {     // what the list wave tells the gatherers (and its later self) about a pair; 64 bytes
  int pair;             // -1: no more pairs
  int n_unique;
  int n_nonreal;        // L - real positions: pads + OOV terms (closed form)
  int n_oov;            // negative document ids
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
__device__ __forceinline__ void stream_build(const StreamSrc& a, int b, unsigned* list, float4* qrows, StreamMeta* meta, unsigned* hash, int lane) {
  const PairIds ids = pair_ids(a.ids, b, a.Q, a.L);
  const int L = a.L;
  asm volatile("" : "+v"(lane));   // opaque per call: nothing lane-derived is hoisted out of the pair loop (and then spilled)
  // hash slots empty (16 per lane)
#pragma unroll
  for (int i = 0; i < kHashSlots / 256; ++i)
    reinterpret_cast<uint4*>(hash)[i * 64 + lane] = make_uint4(kHashEmpty, kHashEmpty, kHashEmpty, kHashEmpty);
  // the document's id row: position r * 64 + lane (all requested together)
  // (unconditional loads at clamped indices: a load under a branch is waited for at the join - fourteen round trips in series)
  typename std::conditional<ID32, int, int64_t>::type dv[kStreamBlocks];
#pragma unroll
  for (int r = 0; r < kStreamBlocks; ++r) {
    const unsigned j = min((unsigned)(r * 64 + lane), (unsigned)(L - 1));   // (unsigned: scalar base + 32-bit lane offset, one register per address)
#if CAPAMD_STREAM_NT_IDS
    if (ID32) dv[r] = __builtin_nontemporal_load(&ids.d32[j]);
    else dv[r] = __builtin_nontemporal_load(&ids.d64[j]);
#else
    if (ID32) dv[r] = ids.d32[j];
    else dv[r] = ids.d64[j];
#endif
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
  int n_real = 0, n_oov = 0;
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
    n_oov += __popcll(__ballot(id < 0));
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
    meta->n_oov = n_oov;
  }
}

// bytes of the common LDS part: qrows[2] | list[2] | hash | meta[2] | gen[2] (+ pad to 16)
__host__ __device__ inline size_t stream_common_lds(int nv, int L) {
  return (size_t)2 * kQT * nv * 16 * 16 + (size_t)2 * ((L + 3) & ~3) * 4 + (size_t)kHashSlots * 4 + 2 * sizeof(StreamMeta) + 16;
}

// A gathering group's row request (NT_ROWS builds: rows of rare terms - large ids in a frequency-ordered vocabulary - with the non-temporal hint)
template <int NV>
__device__ __forceinline__ void stream_load_row(const float* __restrict__ packed, int64_t row, int lane16, RowRegs<NV>& r) {
#if CAPAMD_STREAM_NT_ROWS > 0
  const float4* p = reinterpret_cast<const float4*>(packed + row * (int64_t)(64 * NV)) + lane16;
  if (row >= CAPAMD_STREAM_NT_ROWS) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      r.v[i].x = __builtin_nontemporal_load(&p[i * 16].x);
      r.v[i].y = __builtin_nontemporal_load(&p[i * 16].y);
      r.v[i].z = __builtin_nontemporal_load(&p[i * 16].z);
      r.v[i].w = __builtin_nontemporal_load(&p[i * 16].w);
    }
  } else {
#pragma unroll
    for (int i = 0; i < NV; ++i) r.v[i] = p[i * 16];
  }
#else
  load_row<NV>(packed, row, lane16, r);
#endif
}

#ifndef CAPAMD_STREAM_U
#define CAPAMD_STREAM_U 1          // rows a 16-lane group keeps in flight: 1 (64 registers, 6 workgroups per CU) or 2 (96 registers, 4 per CU; each LDS query chunk read once for both rows)
#endif
constexpr int kStreamU = CAPAMD_STREAM_U;

template <int NV, bool ID32, class M>
__global__ __launch_bounds__(kStreamThreads, kStreamU == 1 ? 8 : 5) void stream_kernel(StreamSrc src, typename M::Args a, int* ticket) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int cap = (src.L + 3) & ~3;
  float4* qrows = reinterpret_cast<float4*>(smem_raw);                      // [2][kQT * NV * 16]
  unsigned* list = reinterpret_cast<unsigned*>(qrows + 2 * kQT * NV * 16);  // [2][cap]
  unsigned* hash = list + 2 * cap;                                          // [kHashSlots]
  StreamMeta* meta = reinterpret_cast<StreamMeta*>(hash + kHashSlots);      // [2]
  volatile int* gen = reinterpret_cast<volatile int*>(meta + 2);            // [2]: which pair-generation buffer b holds, published last (+2 pad)
  char* mlds = reinterpret_cast<char*>(meta + 2) + 16;                      // the model's part

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  if (wave == 4) {
    // ------------------------------------------------------------------ the list wave --------------------------------------
    M::list_init(a, mlds, lane);
    if (lane < 2) gen[lane] = -1;
    int t_next;
    {
      int t = 0;
      if (lane == 0) t = atomicAdd(ticket, 2);          // two tickets: this pair and the next (one atomic round trip ahead from here on)
      t = __builtin_amdgcn_readfirstlane(t);
      t_next = t + 1;
      M::prepare(a, mlds, 0, lane);
      if (t < src.B) stream_build<NV, ID32>(src, t, list, qrows, meta, hash, lane);
      else if (lane == 0) meta[0].pair = -1;
      wave_fence();
      if (lane == 0) gen[0] = 0;
    }
    __syncthreads();
    for (int i = 0;; ++i) {
      const int cur = i & 1, nxt = cur ^ 1;
      if (i > 0) M::finish(a, src, mlds, nxt, meta + nxt, lane);   // pair i-1
      if (meta[cur].pair < 0) break;
      int t_after = 0;
      if (lane == 0) t_after = atomicAdd(ticket, 1);    // the ticket for the build after this one: asked for now, used next trip
      const int t = t_next;
      M::prepare(a, mlds, nxt, lane);
      if (t < src.B) stream_build<NV, ID32>(src, t, list + nxt * cap, qrows + nxt * kQT * NV * 16, meta + nxt, hash, lane);
      else if (lane == 0) meta[nxt].pair = -1;
      wave_fence();
      if (lane == 0) gen[nxt] = i + 1;                  // published: buffer nxt now holds pair-generation i + 1
      t_next = __builtin_amdgcn_readfirstlane(t_after);
      __syncthreads();
    }
    return;
  }

  // -------------------------------------------------------------------- the gathering waves --------------------------------
  const int lane16 = tid & 15;
  const int g = tid >> 4;                      // 0..15
  const int myq = lane16 & 3;
  typename M::Gather gs;
  M::gather_init(a, gs, lane16);
  // Rotated loop: a row's registers are dead once its four dot products are formed, so the group's NEXT row is requested right there -
  // into the same registers - and travels under the current row's reduction / divide / tail.  Requests are unconditional - beyond the
  // list they ask for row 0, the pad row every workgroup keeps hot - because a load under a condition makes its destination a merge
  // point the compiler waits at.
  RowRegs<NV> d[kStreamU];
  unsigned e[kStreamU];
  bool have_first = false;                     // d / e already hold the group's first row(s) of the coming pair (requested before the barrier)
  __syncthreads();
  for (int i = 0;; ++i) {
    const int cur = i & 1, nxt = cur ^ 1;
    const StreamMeta* m = meta + cur;
    if (m->pair < 0) break;
    const int n = m->n_unique;
    QueryPass<NV> qp;
    qp.den_my = m->qden[myq];
    qp.id_my = m->qid[myq];
    M::pair_begin(a, gs, qp, m, lane16);
    const unsigned* lst = list + cur * cap;
    const float4* ql = qrows + cur * kQT * NV * 16;
    int t0 = g;
    // (an entry beyond the list counts as id 0 x 0 occurrences: the pad row, which adds nothing to any model's sums)
    auto request = [&](const unsigned* l, int t, int count) {
#pragma unroll
      for (int u = 0; u < kStreamU; ++u) {
        const int tu = t + u * kGroupsPerWG;
        const unsigned v = l[min(tu, cap - 1)];
        e[u] = tu < count ? v : 0u;
        stream_load_row<NV>(src.packed, (int64_t)(e[u] & kIdMask), lane16, d[u]);
      }
    };
    if (!have_first) request(lst, t0, n);
    while (t0 < n) {
      float p[kStreamU][kQT], dden[kStreamU];
      unsigned e_cur[kStreamU];
      int qoff = 0;
      asm volatile("" : "+v"(qoff));           // opaque per trip: keeps the LDS query reads inside the loop (no LICM into 80 VGPRs)
      if constexpr (kStreamU == 1) rows_dot<NV>(d[0], ql + qoff, lane16, p[0]);
      else rows_dot2<NV>(d[0], d[1], ql + qoff, lane16, p[0], p[1]);
#pragma unroll
      for (int u = 0; u < kStreamU; ++u) {
        dden[u] = row_den<NV>(d[u]);
        e_cur[u] = e[u];
      }
      t0 += kStreamU * kGroupsPerWG;
      request(lst, t0, n);
#pragma unroll
      for (int u = 0; u < kStreamU; ++u) {
        const float x = sim_from_dots<NV>(p[u], dden[u], qp, lane16);
        M::row(a, gs, x, e_cur[u], mlds, cur, lane16);
      }
    }
    // the first row(s) of the next pair, if the list wave has already published it
    {
      const bool ready = CAPAMD_STREAM_PREFETCH && gen[nxt] == i + 1 && meta[nxt].pair >= 0;
      request(list + nxt * cap, g, ready ? meta[nxt].n_unique : 0);
      have_first = ready;
    }
    M::pair_end(a, gs, mlds, cur, wave, lane);
    __syncthreads();
  }
}

// persistent grid of a streaming kernel: workgroups the device holds at once (occupancy x CUs), looked up once per (device, LDS size)
template <class K>
int stream_grid(K kernel, size_t smem) {
  struct Entry { int dev; size_t smem; int grid; };
  static thread_local Entry cache{-1, 0, 0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (cache.dev == dev && cache.smem == smem) return cache.grid;
  int cus = 0, per_cu = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kStreamThreads, smem) != hipSuccess) return 0;
  cache = Entry{dev, smem, cus * per_cu};
  return cache.grid;
}

// Launches stream_kernel<NV, ID32, M> over B pairs; returns false when the geometry / device cannot take it (the caller then runs its
// one-pair-per-workgroup kernel).  `workspace` holds the ticket counter (zeroed here, on the stream).
template <class M>
bool stream_launch(const StreamSrc& src, const typename M::Args& a, int D, void* workspace, size_t workspace_bytes, hipStream_t s, int* rc) {
  if (!workspace || workspace_bytes < sizeof(int) || src.Q > kQT || src.L > kDedupMaxL || src.V > (int64_t)(1u << kIdBits)) return false;
  const int nv = nv_for_dim(D);
  const size_t smem = stream_common_lds(nv, src.L) + M::lds_bytes(a);
  const bool id32 = src.ids.q32 != nullptr;
  int grid = 0;
#define CAPAMD_STREAM_DISPATCH(WHAT)                                  \
  switch (nv) {                                                       \
    case 1: if (id32) { WHAT(1, true); } else { WHAT(1, false); } break; \
    case 2: if (id32) { WHAT(2, true); } else { WHAT(2, false); } break; \
    case 3: if (id32) { WHAT(3, true); } else { WHAT(3, false); } break; \
    case 4: if (id32) { WHAT(4, true); } else { WHAT(4, false); } break; \
    default: if (id32) { WHAT(5, true); } else { WHAT(5, false); } break; \
  }
#define CAPAMD_STREAM_GRID(NV_, I32_) grid = stream_grid(stream_kernel<NV_, I32_, M>, smem)
  CAPAMD_STREAM_DISPATCH(CAPAMD_STREAM_GRID)
  if (grid <= 0) return false;
  static const int per_cu = [] { const char* e = getenv("CAPAMD_STREAM_WG_PER_CU"); return e ? atoi(e) : 0; }();   // profiling: override the occupancy query
  if (per_cu > 0) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) grid = per_cu * cus;
  }
  static const bool debug = getenv("CAPAMD_STREAM_DEBUG") != nullptr;
  if (debug) fprintf(stderr, "[capamd] streaming kernel: %d persistent workgroups of %d threads, %zu bytes of LDS each, %d pairs\n", grid, kStreamThreads, smem, src.B);
  if (grid > src.B) grid = src.B;
  if (hipMemsetAsync(workspace, 0, sizeof(int), s) != hipSuccess) { *rc = CAPAMD_ERR_LAUNCH; return true; }
  int* ticket = static_cast<int*>(workspace);
#define CAPAMD_STREAM_GO(NV_, I32_) hipLaunchKernelGGL((stream_kernel<NV_, I32_, M>), dim3(grid), dim3(kStreamThreads), smem, s, src, a, ticket)
  CAPAMD_STREAM_DISPATCH(CAPAMD_STREAM_GO)
#undef CAPAMD_STREAM_GO
#undef CAPAMD_STREAM_GRID
#undef CAPAMD_STREAM_DISPATCH
  *rc = hipGetLastError() == hipSuccess ? CAPAMD_OK : CAPAMD_ERR_LAUNCH;
  return true;
}

}  // namespace capamd
