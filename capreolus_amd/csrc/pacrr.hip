// Fused PACRR forward for gfx950 (a sibling model on the same fused front end).
//
// Reference semantics: PACRR_class.forward and PACRRConvMax2dModule.forward (capreolus/reranker/PACRR.py:42-78): the
// SimilarityMatrix of KNRM -> per n-gram size ng: zero-pad right/bottom by ng-1, Conv2d(1 -> nfilters, ng x ng) + bias, ReLU,
// max over the filters, the kmax largest values along the document axis (over ALL positions, pads included) -> optional idf
// channel (softmax over the query's raw idf values) -> flatten query-major -> Linear / nonlin / Linear / nonlin / Linear.
//
// Work layout: one workgroup (4 waves) per pair.
//   front end  = knrm.hip's: real document terms compacted in LDS (here with their positions), 16 lanes per gathered
//                embedding row, query rows in an LDS copy; the 4 similarities of a term land in the pair's [Q][L] similarity
//                matrix in LDS (pads and unmatched OOV terms stay 0, OOV exact matches are set to 1).
//   back end   = wave w owns query row w (+4, ...): every lane takes a run of consecutive document positions and keeps
//                the ng x (run + ng - 1) window of the matrix in registers; for every filter the ng*ng weights are read
//                once (LDS broadcast) and applied to the whole run; running max over the filters (ReLU = the 0 it starts
//                from); per-lane sorted top-k, then k rounds of a wave-wide arg-max merge.
//   head       = idf softmax and the three small linear layers by the first threads.
//                (pacrr_forward_kernel: any geometry within the limits below.)
//   MFMA back end (pacrr_mfma_kernel: Q <= 5, nfilters <= 32 - the reference defaults) = the convolutions as matrix products
//                on v_mfma_f32_32x32x16_f16.  M = the 32 filters, N = 32 document positions, K = the 8 rows of the padded
//                similarity matrix at position l + dl (lanes 0-31) and at l + dl + 1 (lanes 32-63): with the matrix kept
//                TRANSPOSED in LDS ([position][8 rows] f16) the B fragment of a lane is one aligned 16-byte LDS read and
//                needs no VALU work, and the A fragment is the filter's column dl of weights placed at rows q .. q + ng - 1
//                (a Toeplitz image, built once per query row in registers).  Row 7 of the matrix is a constant 1 that
//                carries the bias.  fp32 accuracy comes from a two-term f16 split of both operands (hi + lo, 3 products;
//                the dropped lo*lo term is 2^-22 relative), accumulated in fp32 by the matrix pipe.  ReLU + max over the
//                filters = max over the lane's 16 accumulator registers and 0, then one v_permlane32_swap joins the two
//                halves of two adjacent tiles so that all 64 lanes carry one position each into the top-k insertion.
//                (Most K slots multiply zeros - 3 of 16 carry weights - and the matrix pipe is still ~9x the VALU form.)
//   Q <= 4 (pacrr_mfma4_kernel, and every whole-list call): the two-term split of a document offset in ONE product - 6 instead of 12
//                matrix instructions per 32 positions - and no convolutions further than one 64-position step behind the document's
//                last term (see pacrr_mfma4_body).  The form above remains for Q = 5.
#include "capreolus_amd.h"
#include "interaction.h"
#include "lists.h"

using namespace capamd;

namespace {

constexpr int kPacrrMaxQ = 8;       // query rows (two passes of the front end)
constexpr int kPacrrMaxGram = 3;
constexpr int kPacrrMaxK = 4;
constexpr int kPacrrMaxC = 128;     // width of the combine layers
constexpr int kPacrrMaxFeat = kPacrrMaxQ * (kPacrrMaxGram * kPacrrMaxK + 1);

struct PacrrArgs {
  IdSource ids;
  const float* idf;
  int B, Q, L;
  const float* packed;
  int64_t V;
  int mingram, maxgram, nfilters, kmax;
  const float* conv_w;   // n-gram modules back to back, each [nfilters][ng][ng]
  const float* conv_b;   // [n_ngrams][nfilters]
  int n_conv_w;          // floats in conv_w
  int use_idf, C, nonlin;
  const float *w1, *b1, *w2, *b2, *w3, *b3;
  float* out;
  int* status;
  float* feats;          // whole-list route: != nullptr -> the pair's [Q][qts] k-max features go here (row stride kPacrrMaxFeat) and the combine
                         // layers run in pacrr_head_lists_kernel afterwards; nullptr -> the kernel ends with the pair's own head
};

__device__ __forceinline__ float pacrr_act(float x, int nonlin) { return nonlin == 1 ? fmaxf(x, 0.f) : (nonlin == 2 ? tanhf(x) : x); }

// ---- pieces shared by the two kernels ----

// Real document terms (id > 0) compacted in document order, with their positions.  Returns their number.
template <typename Pos>
__device__ __forceinline__ int pacrr_compact(const PacrrArgs& a, const PairIds& ids, int* tok, Pos* pos, int* wave_cnt, int tid) {
  const int wave = tid >> 6, lane = tid & 63;
  int n_real = 0;
  for (int base = 0; base < a.L; base += kThreads) {
    const int j = base + tid;
    int64_t did = (j < a.L) ? ids.d(j) : 0;
    if (did >= a.V) {
      atomicOr(a.status, kErrDocIdRange);
      did = 0;
    }
    const bool real = did > 0;
    const unsigned long long m = __ballot(real);
    if (lane == 0) wave_cnt[wave] = __popcll(m);
    __syncthreads();
    int off = n_real;
    for (int w = 0; w < wave; ++w) off += wave_cnt[w];
    if (real) {
      const int slot = off + __popcll(m & ((1ull << lane) - 1ull));
      tok[slot] = (int)did;
      pos[slot] = (Pos)j;
    }
    n_real += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
  }
  return n_real;
}

// The pair's similarity matrix, kQT query terms per pass; put(row, position, value) stores one entry.
template <int NV, int U, typename Pos, typename Put>
__device__ __forceinline__ void pacrr_similarities(const PacrrArgs& a, const PairIds& ids, const int* tok, const Pos* pos, int n_real,
                                                   float4* qlds, int tid, Put put) {
  const int lane16 = tid & 15, g = tid >> 4;
  for (int q0 = 0; q0 < a.Q; q0 += kQT) {
    QueryPass<NV> qp;
    load_query_pass_lds<NV>(a.packed, ids, a.Q, q0, a.V, tid, kThreads, lane16, qlds, qp, a.status);
    __syncthreads();
    {  // OOV exact matches (equal negative ids): 1.0 (common.py:155-158)
      bool any_oov_q = false;
#pragma unroll
      for (int t = 0; t < kQT; ++t) any_oov_q |= qp.id[t] < 0;
      if (any_oov_q)
        for (int j = tid; j < a.L; j += kThreads) {
          const int64_t did = ids.d(j);
          if (did < 0) {
#pragma unroll
            for (int t = 0; t < kQT; ++t)
              if (qp.id[t] == (int)did && did > -2147483648LL) put(q0 + t, j, 1.f);
          }
        }
    }
    for (int t0 = g; t0 < n_real; t0 += U * kGroupsPerWG) {   // U rows in flight per 16-lane group
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
      asm volatile("" : "+v"(qoff));
      rows_sim_my<NV, U, true>(d, qp, qlds + qoff, lane16, x);
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (has[u] && lane16 < kQT && q0 + lane16 < a.Q) put(q0 + lane16, (int)pos[t0 + u * kGroupsPerWG], x[u]);   // lane l owns query term l & 3
    }
    __syncthreads();
  }
}

// The same over the pair's DISTINCT terms (interaction.h: distinct_terms_positions): a term the document repeats is gathered once and its
// similarities are written to every position it occupies.
template <int NV, int U, typename Put>
__device__ __forceinline__ void pacrr_similarities_distinct(const PacrrArgs& a, const PairIds& ids, const int* tok, const unsigned short* start,
                                                            const unsigned short* plist, int n_unique, float4* qlds, int tid, Put put) {
  const int lane16 = tid & 15, g = tid >> 4;
  for (int q0 = 0; q0 < a.Q; q0 += kQT) {
    QueryPass<NV> qp;
    load_query_pass_lds<NV>(a.packed, ids, a.Q, q0, a.V, tid, kThreads, lane16, qlds, qp, a.status);
    __syncthreads();
    {  // OOV exact matches (equal negative ids): 1.0 (common.py:155-158)
      bool any_oov_q = false;
#pragma unroll
      for (int t = 0; t < kQT; ++t) any_oov_q |= qp.id[t] < 0;
      if (any_oov_q)
        for (int j = tid; j < a.L; j += kThreads) {
          const int64_t did = ids.d(j);
          if (did < 0) {
#pragma unroll
            for (int t = 0; t < kQT; ++t)
              if (qp.id[t] == (int)did && did > -2147483648LL) put(q0 + t, j, 1.f);
          }
        }
    }
    for (int t0 = g; t0 < n_unique; t0 += U * kGroupsPerWG) {   // U rows in flight per 16-lane group
      RowRegs<NV> d[U];
      bool has[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int tu = t0 + u * kGroupsPerWG;
        has[u] = tu < n_unique;
        load_row<NV>(a.packed, has[u] ? tok[tu] : 0, lane16, d[u]);
      }
      float x[U];
      int qoff = 0;
      asm volatile("" : "+v"(qoff));
      rows_sim_my<NV, U, true>(d, qp, qlds + qoff, lane16, x);
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (has[u] && lane16 < kQT && q0 + lane16 < a.Q) {   // lane l owns query term l & 3
          const int tu = t0 + u * kGroupsPerWG;
          for (int i = start[tu], e = start[tu + 1]; i < e; ++i) put(q0 + lane16, (int)plist[i], x[u]);
        }
    }
    __syncthreads();
  }
}

// Per-lane sorted candidate lists -> the kmax largest of the wave, written to dst[0 .. kmax).
template <int KM>
__device__ __forceinline__ void pacrr_wave_topk(float (&top)[KM], int kmax, int lane, float* dst) {
  int head = 0;
  for (int r = 0; r < kmax; ++r) {
    float cand = -INFINITY;
#pragma unroll
    for (int i = 0; i < KM; ++i)
      if (i == head) cand = top[i];
    const float bst = wave_allreduce_max(cand);
    const unsigned long long who = __ballot(cand == bst && bst > -INFINITY);
    if (who == 0) break;
    if (lane == __ffsll((long long)who) - 1) ++head;
    if (lane == 0) dst[r] = bst;
  }
}

template <int KM>
__device__ __forceinline__ void pacrr_insert(float (&top)[KM], float v) {
  sorted_insert<KM>(top, v);
}

// idf channel + the three linear layers (PACRR.py:48-55); feat = [Q][qts] in LDS
// `hw` != nullptr: the three layers' weights and biases staged in LDS by the caller (w1 [C][nin] | w2 [C][C] | w3 [C] | b1 [C] | b2 [C] |
// b3) - one parallel round trip for all of them; from global memory a thread walks its row's weights a few loads at a time, about a dozen
// dependent round trips per pair during which the workgroup does nothing else.
// `idf_pre` != nullptr: the query's raw idf values (clamped index beyond Q) the caller requested earlier, under its convolutions
__device__ __forceinline__ void pacrr_head(const PacrrArgs& a, const PairIds& ids, float* feat, float* h1, float* h2, int qts, int tid, int b,
                                           const float* hw = nullptr, const float* idf_pre = nullptr) {
  if (a.use_idf && tid == 0) {   // softmax over the raw idf values of the query (PACRR.py:48-50)
    const float* idf_g = a.idf + (int64_t)ids.qrow * a.Q;
    float idf[kPacrrMaxQ];   // (Q <= 8; requested together - clamped index - instead of one dependent load per use: three loops over Q by one thread)
#pragma unroll
    for (int q = 0; q < kPacrrMaxQ; ++q) idf[q] = idf_pre ? idf_pre[q < kQT ? q : kQT - 1] : idf_g[q < a.Q ? q : a.Q - 1];
    float m = idf[0];
#pragma unroll
    for (int q = 1; q < 8; ++q)
      if (q < a.Q) m = fmaxf(m, idf[q]);
    float den = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (q < a.Q) den += expf(idf[q] - m);
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (q < a.Q) feat[q * qts + qts - 1] = expf(idf[q] - m) / den;
  }
  __syncthreads();
  const int nin = a.Q * qts;
  const float *w1 = a.w1, *w2 = a.w2, *w3 = a.w3, *b1 = a.b1, *b2 = a.b2, *b3 = a.b3;
  if (hw) {
    w1 = hw;
    w2 = w1 + a.C * nin;
    w3 = w2 + a.C * a.C;
    b1 = w3 + a.C;
    b2 = b1 + a.C;
    b3 = b2 + a.C;
  }
  if (tid < a.C) {
    float s = b1[tid];
    for (int i = 0; i < nin; ++i) s = __builtin_fmaf(w1[tid * nin + i], feat[i], s);
    h1[tid] = pacrr_act(s, a.nonlin);
  }
  __syncthreads();
  if (tid < a.C) {
    float s = b2[tid];
    for (int i = 0; i < a.C; ++i) s = __builtin_fmaf(w2[tid * a.C + i], h1[i], s);
    h2[tid] = pacrr_act(s, a.nonlin);
  }
  __syncthreads();
  if (tid == 0) {
    float s = b3[0];
    for (int i = 0; i < a.C; ++i) s = __builtin_fmaf(w3[i], h2[i], s);
    a.out[b] = s;
  }
}

// floats of the head's weights and biases as pacrr_head wants them staged
__device__ __forceinline__ int pacrr_head_floats(const PacrrArgs& a, int nin) { return a.C * nin + a.C * a.C + 3 * a.C + 1; }
__device__ __forceinline__ void pacrr_stage_head(const PacrrArgs& a, int nin, float* hw, int tid) {
  const int n1 = a.C * nin, n2 = a.C * a.C;
  for (int i = tid; i < n1; i += kThreads) hw[i] = a.w1[i];
  for (int i = tid; i < n2; i += kThreads) hw[n1 + i] = a.w2[i];
  if (tid < a.C) {
    hw[n1 + n2 + tid] = a.w3[tid];
    hw[n1 + n2 + a.C + tid] = a.b1[tid];
    hw[n1 + n2 + 2 * a.C + tid] = a.b2[tid];
  }
  if (tid == 0) hw[n1 + n2 + 3 * a.C] = a.b3[0];
}

// ---- general kernel: fp32 VALU convolutions.  PPL = document positions per lane in the back end (64 * PPL >= L) ----
template <int NV, int PPL>
__global__ __launch_bounds__(kThreads, 4) void pacrr_forward_kernel(PacrrArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int LS = 64 * PPL + kPacrrMaxGram;                       // row stride of the similarity matrix (zero tail = right padding)
  const int QS = a.Q + kPacrrMaxGram - 1;                        // rows incl. the zero bottom padding
  int* tok = reinterpret_cast<int*>(smem_raw);
  const int tok_cap = (a.L + 3) & ~3;
  int* pos = tok + tok_cap;                                      // position of each compacted term
  float* sim = reinterpret_cast<float*>(pos + tok_cap);          // [QS][LS]
  float* wts = sim + QS * LS;                                    // conv_w | conv_b
  float* feat = wts + a.n_conv_w + (a.maxgram - a.mingram + 1) * a.nfilters;   // [Q][qts]
  float* h1 = feat + kPacrrMaxFeat;                              // [C]
  float* h2 = h1 + kPacrrMaxC;                                   // [C]
  int* wave_cnt = reinterpret_cast<int*>(h2 + kPacrrMaxC);       // [4] (+4 spare)
  float4* qlds = reinterpret_cast<float4*>(wave_cnt + 8);        // [kQT][NV*16] float4

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int b = blockIdx.x;
  const PairIds ids = pair_ids(a.ids, b, a.Q, a.L);
  const int n_ng = a.maxgram - a.mingram + 1, qts = n_ng * a.kmax + (a.use_idf ? 1 : 0);

  for (int i = tid; i < QS * LS; i += kThreads) sim[i] = 0.f;
  for (int i = tid; i < a.n_conv_w; i += kThreads) wts[i] = a.conv_w[i];
  for (int i = tid; i < n_ng * a.nfilters; i += kThreads) wts[a.n_conv_w + i] = a.conv_b[i];

  const int n_real = pacrr_compact(a, ids, tok, pos, wave_cnt, tid);
  pacrr_similarities<NV, 1>(a, ids, tok, pos, n_real, qlds, tid, [&](int row, int j, float x) { sim[row * LS + j] = x; });

  // ---- convolutions, ReLU, max over filters, k-max over the document: wave w owns query rows w, w + 4, ... ----
  for (int q = wave; q < a.Q; q += 4) {
    const int j0 = lane * PPL;
    const float* wg = wts;
    for (int gi = 0; gi < n_ng; ++gi) {
      const int ng = a.mingram + gi;
      float best[PPL];
#pragma unroll
      for (int r = 0; r < PPL; ++r) best[r] = 0.f;   // ReLU: the maximum over the filters is never below 0
      // the ng x (PPL + ng - 1) window of the matrix this lane's positions see (zero padding is part of the LDS image)
      float win[kPacrrMaxGram][PPL + kPacrrMaxGram - 1];
#pragma unroll
      for (int r0 = 0; r0 < kPacrrMaxGram; ++r0)
#pragma unroll
        for (int cidx = 0; cidx < PPL + kPacrrMaxGram - 1; ++cidx)
          win[r0][cidx] = (r0 < ng && cidx < PPL + ng - 1) ? sim[(q + r0) * LS + j0 + cidx] : 0.f;
      const float* bg = wts + a.n_conv_w + gi * a.nfilters;
      for (int f = 0; f < a.nfilters; ++f) {
        const float* wf = wg + f * ng * ng;
        float acc[PPL];
        const float bias = bg[f];
#pragma unroll
        for (int r = 0; r < PPL; ++r) acc[r] = bias;
#pragma unroll
        for (int r0 = 0; r0 < kPacrrMaxGram; ++r0)
#pragma unroll
          for (int c0 = 0; c0 < kPacrrMaxGram; ++c0)
            if (r0 < ng && c0 < ng) {
              const float wv = wf[r0 * ng + c0];   // same address in every lane: one LDS broadcast per weight and filter
#pragma unroll
              for (int r = 0; r < PPL; ++r) acc[r] = __builtin_fmaf(wv, win[r0][r + c0], acc[r]);
            }
#pragma unroll
        for (int r = 0; r < PPL; ++r) best[r] = fmaxf(best[r], acc[r]);
      }
      wg += a.nfilters * ng * ng;
      // per-lane sorted top-k over its valid positions, then k rounds of a wave-wide arg-max merge
      float top[kPacrrMaxK];
#pragma unroll
      for (int i = 0; i < kPacrrMaxK; ++i) top[i] = -INFINITY;
#pragma unroll
      for (int r = 0; r < PPL; ++r) pacrr_insert(top, (j0 + r < a.L) ? best[r] : -INFINITY);
      pacrr_wave_topk(top, a.kmax, lane, feat + q * qts + gi * a.kmax);
    }
  }
  pacrr_head(a, ids, feat, h1, h2, qts, tid, b);
}

// ---- MFMA kernel (Q <= kMfmaMaxQ, nfilters <= 32): see the header comment ----
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#ifndef CAPAMD_PACRR_ABLATE
#define CAPAMD_PACRR_ABLATE 0       // measurement builds only: 1 = no convolutions, 2 = no gather, 3 = no combine layers, 4 = no table lookups (list route)
#endif
#ifndef CAPAMD_PACRR_U
#define CAPAMD_PACRR_U 1            // embedding rows in flight per 16-lane group in the gather loop
#endif
#ifndef CAPAMD_PACRR_WAVES
#define CAPAMD_PACRR_WAVES 4        // register budget: waves per SIMD.  Measured per 64,000 pairs with the 36.7 KB LDS image (4 workgroups per CU):
                                    // U1/W4 3.47 ms (no spills), U2/W4 3.45 (20 spilled registers), U3/W3 3.52, U4/W3 3.65, U3/W4 5.64 (47 spills)
#endif
constexpr int kMfmaMaxQ = 5;        // rows 0 .. Q + 1 of the padded matrix + the bias row fit the 8 K slots of a half-wave
constexpr int kBiasRow = 7;

__device__ __forceinline__ unsigned f16_bits(float x) { return (unsigned)__builtin_bit_cast(unsigned short, (_Float16)x); }
__device__ __forceinline__ float f16_round(float x) { return (float)(_Float16)x; }

// x (three f16 in bits 0..47) placed at halfword `q` of a 128-bit fragment
__device__ __forceinline__ u32x4 place_halfwords(unsigned long long x, int q) {
  unsigned long long lo, hi;
  if (q == 0) {
    lo = x;
    hi = 0;
  } else if (q < 4) {
    lo = x << (16 * q);
    hi = x >> (64 - 16 * q);
  } else {
    lo = 0;
    hi = x << (16 * (q - 4));
  }
  u32x4 r;
  r[0] = (unsigned)lo;
  r[1] = (unsigned)(lo >> 32);
  r[2] = (unsigned)hi;
  r[3] = (unsigned)(hi >> 32);
  return r;
}

// A fragments (hi, lo) of one product: rows = this lane's filter, K slots q .. q + ng - 1 = column `dl` of its ng x ng weights,
// slot kBiasRow = its bias (lanes 0-31 of the first product of an n-gram size only).
__device__ __forceinline__ void pacrr_a_fragment(const float* w_ng, const float* b_ng, int ng, int dl, int nfilters, int q, bool with_bias,
                                                 int lane, h8& a_hi, h8& a_lo) {
  const int f = lane & 31;
  const bool live = f < nfilters && dl < ng;
  unsigned long long xh = 0, xl = 0;
#pragma unroll
  for (int dq = 0; dq < kPacrrMaxGram; ++dq)
    if (dq < ng) {
      const float w = live ? w_ng[(f * ng + dq) * ng + dl] : 0.f;
      const float h = f16_round(w);
      xh |= (unsigned long long)f16_bits(h) << (16 * dq);
      xl |= (unsigned long long)f16_bits(w - h) << (16 * dq);
    }
  u32x4 fh = place_halfwords(xh, q), fl = place_halfwords(xl, q);
  if (with_bias && lane < 32 && f < nfilters) {
    const float bv = b_ng[f], h = f16_round(bv);
    fh[3] |= f16_bits(h) << 16;
    fl[3] |= f16_bits(bv - h) << 16;
  }
  a_hi = __builtin_bit_cast(h8, fh);
  a_lo = __builtin_bit_cast(h8, fl);
}

__device__ __forceinline__ void pacrr_swap32(float& x, float& y) {
  unsigned ux = __float_as_uint(x), uy = __float_as_uint(y);
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(ux), "+v"(uy));
  x = __uint_as_float(ux);
  y = __uint_as_float(uy);
}

__device__ __forceinline__ float pacrr_relu_max(const f32x16& c) {
  float m = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) m = fmaxf(m, c[i]);
  return m;
}

// bytes of the LDS region shared by the front end's term list and the back end's weights / head vectors
__host__ __device__ inline int pacrr_mfma_region0(int L, int n_weights) {
  const int front = ((L + 7) & ~7) * 8 + 16, back = (n_weights + kPacrrMaxFeat + 2 * kPacrrMaxC) * 4;   // tok int32 | start uint16 (+1) | plist uint16
  return ((front > back ? front : back) + 15) & ~15;
}

// KM = length of the per-lane candidate lists (>= kmax)
// `table` != nullptr: the whole-list route (lists.h) - the pair's list has its terms' four similarities in table[id] already, and the
// front end is a lookup per position instead of the distinct-term pass and the gather (same values: the matrix, and with it the score,
// is bit-identical).
template <int NV, int KM>
__device__ __forceinline__ void pacrr_mfma_body(const PacrrArgs& a, const int b, const float4* table) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int tok_cap = (a.L + 7) & ~7;
  const int LP = ((a.L + 63) & ~63) + 4;                         // positions in the LDS image (zero tail = right padding)
  // region 0 is used twice: the compacted terms (front end), then the convolution weights and the head's vectors (back end)
  const int r0 = pacrr_mfma_region0(a.L, a.n_conv_w + (a.maxgram - a.mingram + 1) * a.nfilters);
  int* tok = reinterpret_cast<int*>(smem_raw);
  unsigned short* start = reinterpret_cast<unsigned short*>(tok + tok_cap);   // [tok_cap + 8]
  unsigned short* plist = start + tok_cap + 8;                                // [tok_cap]
  float* wts = reinterpret_cast<float*>(smem_raw);               // conv_w | conv_b   (after the front end)
  float* feat = wts + a.n_conv_w + (a.maxgram - a.mingram + 1) * a.nfilters;
  float* h1 = feat + kPacrrMaxFeat;
  float* h2 = h1 + kPacrrMaxC;
  _Float16* s_hi = reinterpret_cast<_Float16*>(smem_raw + r0);   // [LP][8]: f16(sim[row][position]), row kBiasRow = 1
  _Float16* s_lo = s_hi + LP * 8;                                // [LP][8]: f16(sim - hi)
  int* wave_cnt = reinterpret_cast<int*>(s_lo + LP * 8);         // [48]: distinct_terms_positions' per-wave counts
  float4* qlds = reinterpret_cast<float4*>(wave_cnt + 48);       // (192 bytes after a 16-byte aligned plane: aligned)

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const PairIds ids = pair_ids(a.ids, b, a.Q, a.L);
  const int n_ng = a.maxgram - a.mingram + 1, qts = n_ng * a.kmax + (a.use_idf ? 1 : 0);

  // distinct terms and their positions first: the hash of that pass borrows the (not yet initialised) matrix planes
  int n_real = 0;
  if (!table) {
    const TermList tl = distinct_terms_positions(ids, a.L, a.V, a.status, tok, start, plist, reinterpret_cast<int*>(s_hi), LP * 8, wave_cnt);
    n_real = tl.n_unique;
  }
  {
    u32x4 z = {0u, 0u, 0u, 0u}, one = {0u, 0u, 0u, 0x3C000000u};   // row 7 = 1.0
    for (int i = tid; i < LP; i += kThreads) {
      reinterpret_cast<u32x4*>(s_hi)[i] = one;
      reinterpret_cast<u32x4*>(s_lo)[i] = z;
    }
  }
  if (CAPAMD_PACRR_ABLATE == 2) n_real = 0;
  auto put = [&](int row, int j, float x) {
    const float h = f16_round(x);
    s_hi[j * 8 + row] = (_Float16)h;
    s_lo[j * 8 + row] = (_Float16)(x - h);
  };
  if (table) {
    int64_t qid[kQT];
#pragma unroll
    for (int t = 0; t < kQT; ++t) qid[t] = t < a.Q ? ids.q(t) : 0;
    __syncthreads();        // (the planes are initialised)
    for (int j = tid; j < a.L; j += kThreads) {
      const int64_t did = ids.d(j);
      if (did > 0 && did < a.V) {        // (an id beyond the table was flagged by the mark pass)
        const float4 x = table[did];
        put(0, j, x.x);
        if (a.Q > 1) put(1, j, x.y);
        if (a.Q > 2) put(2, j, x.z);
        if (a.Q > 3) put(3, j, x.w);
      } else if (did < 0 && did > -2147483648LL) {   // OOV exact matches (equal negative ids): 1.0 (common.py:155-158)
#pragma unroll
        for (int t = 0; t < kQT; ++t)
          if (qid[t] < 0 && (int)qid[t] == (int)did) put(t, j, 1.f);
      }
    }
    __syncthreads();
  } else {
    pacrr_similarities_distinct<NV, CAPAMD_PACRR_U>(a, ids, tok, start, plist, n_real, qlds, tid, put);
  }
  // (the front end ended on a barrier: tok / pos are dead, region 0 now takes the weights)
  for (int i = tid; i < a.n_conv_w; i += kThreads) wts[i] = a.conv_w[i];
  for (int i = tid; i < n_ng * a.nfilters; i += kThreads) wts[a.n_conv_w + i] = a.conv_b[i];
  // no convolutions further than one 64-position step behind the document's last term (see pacrr_mfma4_body: the same k largest)
  int last = -1;
  for (int j = tid; j < a.L; j += kThreads)
    if (ids.d(j) != 0) last = j;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) last = max(last, __shfl_xor(last, o));
  if (lane == 0) wave_cnt[wave] = last;
  __syncthreads();
  const int l_end = min(a.L, ((max(max(wave_cnt[0], wave_cnt[1]), max(wave_cnt[2], wave_cnt[3])) + 64) & ~63) + 64);

  // ---- convolutions on the matrix pipe; wave w owns query rows w, w + 4 ----
  for (int q = wave; q < (CAPAMD_PACRR_ABLATE == 1 ? 0 : a.Q); q += 4) {
    // products: [0] ng=1 (dl 0 | -), [1] ng=2 (dl 0 | 1), [2] ng=3 (dl 0 | 1), [3] ng=3 (dl 2 | -); lanes 32-63 take the second dl
    h8 ah[4], al[4];
    const int half = lane >> 5;
    {
      const float* w = wts;
      const float* bb = wts + a.n_conv_w;
#pragma unroll
      for (int ng = 1; ng <= kPacrrMaxGram; ++ng) {
        const bool on = ng >= a.mingram && ng <= a.maxgram;
        const int first = ng == 1 ? 0 : (ng == 2 ? 1 : 2);
        // an n-gram size that is switched off keeps all-zero fragments (nfilters = 0)
        pacrr_a_fragment(w, bb, ng, half, on ? a.nfilters : 0, q, true, lane, ah[first], al[first]);
        if (ng == 3) pacrr_a_fragment(w, bb, ng, 2 + half, on ? a.nfilters : 0, q, false, lane, ah[3], al[3]);
        if (on) {
          w += a.nfilters * ng * ng;
          bb += a.nfilters;
        }
      }
    }
    float top[kPacrrMaxGram][KM];
#pragma unroll
    for (int g = 0; g < kPacrrMaxGram; ++g)
#pragma unroll
      for (int i = 0; i < KM; ++i) top[g][i] = -INFINITY;

    for (int l0 = 0; l0 < l_end; l0 += 64) {
      float m[kPacrrMaxGram][2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int p = l0 + 32 * t + (lane & 31) + half;            // lanes 32-63 read the next position: the second dl of a product
        const h8 bh0 = *reinterpret_cast<const h8*>(s_hi + p * 8), bl0 = *reinterpret_cast<const h8*>(s_lo + p * 8);
        const h8 bh2 = *reinterpret_cast<const h8*>(s_hi + (p + 2) * 8), bl2 = *reinterpret_cast<const h8*>(s_lo + (p + 2) * 8);
        f32x16 c1 = {0}, c2 = {0}, c3 = {0};
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0], bh0, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[1], bh0, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[2], bh0, c3, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[0], bh0, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[1], bh0, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[2], bh0, c3, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0], bl0, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[1], bl0, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[2], bl0, c3, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[3], bh2, c3, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[3], bh2, c3, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[3], bl2, c3, 0, 0, 0);
        m[0][t] = pacrr_relu_max(c1);   // ReLU + max over this lane's 16 filters
        m[1][t] = pacrr_relu_max(c2);
        m[2][t] = pacrr_relu_max(c3);
      }
#pragma unroll
      for (int g = 0; g < kPacrrMaxGram; ++g) {
        // lanes 0-31 end with tile 0's position (lane), lanes 32-63 with tile 1's (lane - 32): the value of position l0 + lane
        pacrr_swap32(m[g][0], m[g][1]);
        const float v = fmaxf(m[g][0], m[g][1]);
        pacrr_insert(top[g], (l0 + lane < a.L) ? v : -INFINITY);
      }
    }
#pragma unroll
    for (int ng = 1; ng <= kPacrrMaxGram; ++ng)
      if (ng >= a.mingram && ng <= a.maxgram) pacrr_wave_topk(top[ng - 1], a.kmax, lane, feat + q * qts + (ng - a.mingram) * a.kmax);
  }
  if (CAPAMD_PACRR_ABLATE == 3) {
    __syncthreads();
    if (tid == 0) a.out[b] = feat[0];
    return;
  }
  // the matrix planes are dead once every wave has its rows' k-max values: the head's weights take their place (when they fit)
  const float* hw = nullptr;
  {
    const int nin = a.Q * qts;
    if ((size_t)pacrr_head_floats(a, nin) * 4 <= (size_t)LP * 32) {
      __syncthreads();
      pacrr_stage_head(a, nin, reinterpret_cast<float*>(s_hi), tid);
      hw = reinterpret_cast<const float*>(s_hi);
    }
  }
  pacrr_head(a, ids, feat, h1, h2, qts, tid, b, hw);
}

// ---- the Q <= 4 form of the MFMA back end: half the matrix instructions of the form above ----
// A 32x32x16 product has 16 K slots; the form above fills 3 of them per half-wave and spends one product on each of hi*hi, lo*hi, hi*lo.
// With at most four query rows the WHOLE two-term split of one document offset fits one product:
//     image (LDS, 32 bytes per position):  chunk A = [hi row 0..3 | lo row 0..3]      chunk B = [hi row 0..3 | 1 | 1 | 0 | 0]
//     lanes 0-31  (K 0-7)  read chunk A of position p + dl:  weights' hi part at slots q + dq (x hi) and 4 + q + dq (x lo)
//     lanes 32-63 (K 8-15) read chunk B of the same position: weights' lo part at slots q + dq (x hi), the bias' hi and lo parts at the
//                                                             two constant slots (first product of an n-gram size)
// i.e. w_hi * (s_hi + s_lo) + w_lo * s_hi + b in ONE instruction per (n-gram size, column dl) - 1 + 2 + 3 = 6 per 32 positions instead
// of 12 - and the same three terms as before (lo * lo dropped, 2^-22 relative).  Rows q + dq >= 4 are the bottom padding: their slots
// fall off the 64-bit shift.  The front end writes a position's image as two 16-byte stores (the lookup route) instead of eight
// halfword stores.
__device__ __forceinline__ h8 pacrr_a4_fragment(const float* w_ng, const float* b_ng, int ng, int dl, int nfilters, int q, bool with_bias, int lane) {
  const int f = lane & 31;
  const bool upper = lane >= 32, live = f < nfilters && dl < ng;
  unsigned long long x = 0;
#pragma unroll
  for (int dq = 0; dq < kPacrrMaxGram; ++dq)
    if (dq < ng) {
      const float w = live ? w_ng[(f * ng + dq) * ng + dl] : 0.f;
      const float h = f16_round(w);
      x |= (unsigned long long)f16_bits(upper ? w - h : h) << (16 * dq);
    }
  const unsigned long long rows = x << (16 * q);
  u32x4 r = {(unsigned)rows, (unsigned)(rows >> 32), upper ? 0u : (unsigned)rows, upper ? 0u : (unsigned)(rows >> 32)};
  if (with_bias && upper && f < nfilters) {
    const float bv = b_ng[f], h = f16_round(bv);
    r[2] = f16_bits(h) | (f16_bits(bv - h) << 16);
  }
  return __builtin_bit_cast(h8, r);
}

template <int NV, int KM>
__device__ __forceinline__ void pacrr_mfma4_body(const PacrrArgs& a, const int b, const float4* table) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int tok_cap = (a.L + 7) & ~7;
  const int LP = ((a.L + 63) & ~63) + 4;                         // positions in the LDS image (zero tail = right padding)
  const int r0 = pacrr_mfma_region0(a.L, a.n_conv_w + (a.maxgram - a.mingram + 1) * a.nfilters);
  int* tok = reinterpret_cast<int*>(smem_raw);
  unsigned short* start = reinterpret_cast<unsigned short*>(tok + tok_cap);
  unsigned short* plist = start + tok_cap + 8;
  float* wts = reinterpret_cast<float*>(smem_raw);               // conv_w | conv_b   (after the front end; the lookup route: from the start)
  float* feat = wts + a.n_conv_w + (a.maxgram - a.mingram + 1) * a.nfilters;
  float* h1 = feat + kPacrrMaxFeat;
  float* h2 = h1 + kPacrrMaxC;
  _Float16* img = reinterpret_cast<_Float16*>(smem_raw + r0);    // [LP][16]: chunk A | chunk B of the comment above
  int* wave_cnt = reinterpret_cast<int*>(img + LP * 16);
  float4* qlds = reinterpret_cast<float4*>(wave_cnt + 48);

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const PairIds ids = pair_ids(a.ids, b, a.Q, a.L);
  const int n_ng = a.maxgram - a.mingram + 1, qts = n_ng * a.kmax + (a.use_idf ? 1 : 0);
  const u32x4 zero4 = {0u, 0u, 0u, 0u}, ones4 = {0u, 0u, 0x3C003C00u, 0u};   // chunk B of an empty position: slots 4, 5 = 1.0
  int last = -1;         // this thread's last position that is not padding (id != 0)

  if (table) {
    // one pass: every position of the image is written once (pads and the tail as zeros), the weights go to region 0 under it
    int64_t qid[kQT];
#pragma unroll
    for (int t = 0; t < kQT; ++t) qid[t] = t < a.Q ? ids.q(t) : 0;
    for (int j = tid; j < LP; j += kThreads) {
      float x[kQT] = {0.f, 0.f, 0.f, 0.f};
      const int64_t did = j < a.L ? ids.d(j) : 0;
      if (did != 0) last = j;
      if (did > 0 && did < a.V) {        // (an id beyond the table was flagged by the mark pass)
        const float4 v = CAPAMD_PACRR_ABLATE == 4 ? make_float4(0.25f, 0.5f, 0.125f, 0.75f) : table[did];
        x[0] = v.x;
        if (a.Q > 1) x[1] = v.y;
        if (a.Q > 2) x[2] = v.z;
        if (a.Q > 3) x[3] = v.w;
      } else if (did < 0 && did > -2147483648LL) {   // OOV exact matches (equal negative ids): 1.0 (common.py:155-158)
#pragma unroll
        for (int t = 0; t < kQT; ++t)
          if (qid[t] < 0 && (int)qid[t] == (int)did) x[t] = 1.f;
      }
      unsigned hb[kQT], lb[kQT];
#pragma unroll
      for (int t = 0; t < kQT; ++t) {
        const float h = f16_round(x[t]);
        hb[t] = f16_bits(h);
        lb[t] = f16_bits(x[t] - h);
      }
      const u32x4 ca = {hb[0] | (hb[1] << 16), hb[2] | (hb[3] << 16), lb[0] | (lb[1] << 16), lb[2] | (lb[3] << 16)};
      const u32x4 cb = {ca[0], ca[1], 0x3C003C00u, 0u};
      reinterpret_cast<u32x4*>(img)[2 * j] = ca;
      reinterpret_cast<u32x4*>(img)[2 * j + 1] = cb;
    }
  } else {
    // distinct terms and their positions first: the hash of that pass borrows the (not yet initialised) image
    const TermList tl = distinct_terms_positions(ids, a.L, a.V, a.status, tok, start, plist, reinterpret_cast<int*>(img), LP * 8, wave_cnt);
    int n_real = tl.n_unique;
    for (int i = tid; i < LP; i += kThreads) {
      reinterpret_cast<u32x4*>(img)[2 * i] = zero4;
      reinterpret_cast<u32x4*>(img)[2 * i + 1] = ones4;
    }
    if (CAPAMD_PACRR_ABLATE == 2) n_real = 0;
    auto put = [&](int row, int j, float x) {
      const float h = f16_round(x);
      img[j * 16 + row] = (_Float16)h;
      img[j * 16 + 4 + row] = (_Float16)(x - h);
      img[j * 16 + 8 + row] = (_Float16)h;
    };
    pacrr_similarities_distinct<NV, CAPAMD_PACRR_U>(a, ids, tok, start, plist, n_real, qlds, tid, put);
    // (the front end ended on a barrier: tok / pos are dead, region 0 now takes the weights)
    for (int j = tid; j < a.L; j += kThreads)
      if (ids.d(j) != 0) last = j;
  }
  for (int i = tid; i < a.n_conv_w; i += kThreads) wts[i] = a.conv_w[i];
  for (int i = tid; i < n_ng * a.nfilters; i += kThreads) wts[a.n_conv_w + i] = a.conv_b[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) last = max(last, __shfl_xor(last, o));
  if (lane == 0) wave_cnt[wave] = last;
  __syncthreads();
  // Behind the document's last term every window is all padding: the same value (ReLU of the bias, max over the filters) at every such
  // position.  The reference takes its k largest over ALL positions, so those values count - but kmax <= 4 copies of them are all the
  // k-max can use.  The loop below covers the positions up to the last term in 64-position steps plus ONE more step: either that step
  // holds the rest of the document, or it holds 64 all-padding positions, more copies than the k-max can take.  Same k largest, bit for bit.
  const int l_end = min(a.L, ((max(max(wave_cnt[0], wave_cnt[1]), max(wave_cnt[2], wave_cnt[3])) + 64) & ~63) + 64);

  // The head's weights and the query's idf row are REQUESTED here, into registers, and reach LDS behind the convolutions: asked for
  // after them, the two round trips (idf row; ~8 KB of weights) sit in a pair's serial tail with nothing of this workgroup beside them
  // (the combine layers were 0.19 ms of the 1.37 ms call).  8 weights per thread cover the reference's default head (2,017 floats).
  constexpr int kHeadRegs = 8;
  const int nin_h = a.Q * qts, n1_h = a.C * nin_h, n2_h = a.C * a.C, nh = pacrr_head_floats(a, nin_h);
  // (kmax <= 2 builds only: with four candidates per lane and n-gram size the eight registers are two spills at 128)
  const bool head_in_regs = KM <= 2 && !a.feats && (size_t)nh * 4 <= (size_t)LP * 32 && nh <= kHeadRegs * kThreads;
  float hreg[kHeadRegs], idf_pre[kQT];
  if (head_in_regs) {
#pragma unroll
    for (int k = 0; k < kHeadRegs; ++k) {
      const int i = tid + k * kThreads;
      const float* src = i < n1_h ? a.w1 + i : i < n1_h + n2_h ? a.w2 + (i - n1_h) : i < n1_h + n2_h + a.C ? a.w3 + (i - n1_h - n2_h)
                         : i < n1_h + n2_h + 2 * a.C ? a.b1 + (i - n1_h - n2_h - a.C) : i < n1_h + n2_h + 3 * a.C ? a.b2 + (i - n1_h - n2_h - 2 * a.C) : a.b3;
      hreg[k] = i < nh ? *src : 0.f;
    }
  }
  if (a.use_idf && !a.feats) {
    const float* idf_g = a.idf + (int64_t)ids.qrow * a.Q;
#pragma unroll
    for (int t = 0; t < kQT; ++t) idf_pre[t] = idf_g[t < a.Q ? t : a.Q - 1];
  }

  // ---- convolutions on the matrix pipe; wave w owns query row w ----
  for (int q = wave; q < (CAPAMD_PACRR_ABLATE == 1 ? 0 : a.Q); q += 4) {
    // products: [0] ng=1 dl 0; [1], [2] ng=2 dl 0, 1; [3], [4], [5] ng=3 dl 0, 1, 2
    h8 af[6];
    {
      const float* w = wts;
      const float* bb = wts + a.n_conv_w;
#pragma unroll
      for (int ng = 1; ng <= kPacrrMaxGram; ++ng) {
        const bool on = ng >= a.mingram && ng <= a.maxgram;
        const int first = ng * (ng - 1) / 2;
        // an n-gram size that is switched off keeps all-zero fragments (nfilters = 0)
#pragma unroll
        for (int dl = 0; dl < ng; ++dl) af[first + dl] = pacrr_a4_fragment(w, bb, ng, dl, on ? a.nfilters : 0, q, dl == 0, lane);
        if (on) {
          w += a.nfilters * ng * ng;
          bb += a.nfilters;
        }
      }
    }
    float top[kPacrrMaxGram][KM];
#pragma unroll
    for (int g = 0; g < kPacrrMaxGram; ++g)
#pragma unroll
      for (int i = 0; i < KM; ++i) top[g][i] = -INFINITY;

    const _Float16* mine = img + (lane & 31) * 16 + (lane >> 5) * 8;       // lanes 32-63: chunk B
    for (int l0 = 0; l0 < l_end; l0 += 64) {
      float m[kPacrrMaxGram][2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const _Float16* at = mine + (l0 + 32 * t) * 16;
        const h8 b0 = *reinterpret_cast<const h8*>(at), b1 = *reinterpret_cast<const h8*>(at + 16), b2 = *reinterpret_cast<const h8*>(at + 32);
        f32x16 c1 = {0}, c2 = {0}, c3 = {0};
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0], b0, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[1], b0, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[3], b0, c3, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[2], b1, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[4], b1, c3, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[5], b2, c3, 0, 0, 0);
        m[0][t] = pacrr_relu_max(c1);   // ReLU + max over this lane's 16 filters
        m[1][t] = pacrr_relu_max(c2);
        m[2][t] = pacrr_relu_max(c3);
      }
#pragma unroll
      for (int g = 0; g < kPacrrMaxGram; ++g) {
        // lanes 0-31 end with tile 0's position (lane), lanes 32-63 with tile 1's (lane - 32): the value of position l0 + lane
        pacrr_swap32(m[g][0], m[g][1]);
        const float v = fmaxf(m[g][0], m[g][1]);
        pacrr_insert(top[g], (l0 + lane < a.L) ? v : -INFINITY);
      }
    }
#pragma unroll
    for (int ng = 1; ng <= kPacrrMaxGram; ++ng)
      if (ng >= a.mingram && ng <= a.maxgram) pacrr_wave_topk(top[ng - 1], a.kmax, lane, feat + q * qts + (ng - a.mingram) * a.kmax);
  }
  if (CAPAMD_PACRR_ABLATE == 3) {
    __syncthreads();
    if (tid == 0) a.out[b] = feat[0];
    return;
  }
  if (a.feats) {          // the combine layers of all the call's pairs run in one pass afterwards (pacrr_head_lists_kernel)
    __syncthreads();
    if (tid < a.Q * qts) a.feats[(int64_t)b * kPacrrMaxFeat + tid] = feat[tid];
    return;
  }
  // the image is dead once every wave has its rows' k-max values: the head's weights take its place (when they fit)
  const float* hw = nullptr;
  {
    const int nin = a.Q * qts;
    if (head_in_regs) {
      __syncthreads();
      float* dst = reinterpret_cast<float*>(img);
#pragma unroll
      for (int k = 0; k < kHeadRegs; ++k) {
        const int i = tid + k * kThreads;
        if (i < nh) dst[i] = hreg[k];
      }
      hw = dst;
    } else if ((size_t)pacrr_head_floats(a, nin) * 4 <= (size_t)LP * 32) {
      __syncthreads();
      pacrr_stage_head(a, nin, reinterpret_cast<float*>(img), tid);
      hw = reinterpret_cast<const float*>(img);
    }
  }
  pacrr_head(a, ids, feat, h1, h2, qts, tid, b, hw, a.use_idf ? idf_pre : nullptr);
}

// The combine layers of the whole-list route, for all pairs [p0, p0 + n) of a launch group in one pass: inside pacrr_mfma_lists_kernel they
// are a serial tail per pair - idf row, 8 KB of weights staged, three dependent layers on 32 of 256 threads - 0.19 ms of a 1.33 ms call.
// Here the weights are staged once per workgroup, TPP threads take a pair (TPP >= C), 256 / TPP pairs per trip.  The arithmetic per pair is
// pacrr_head's, operation for operation (sequential idf softmax by the pair's first thread, fma chains in input order): the same bits.
// The reference's default head (Q x (3 kmax + 1) <= 32 inputs, combine <= 32) with every thread's weight rows in REGISTERS across its
// workgroup's trips: thread (slot, o) keeps row o of w1 and w2 (zeros beyond nin / C: fma(0, x, s) = s exactly, so the unrolled 32-step
// chains round like pacrr_head's nin- and C-step chains), the trip's inputs are one LDS broadcast read per four values, and the next
// trip's features and idf row are requested before the current trip's layers.  (The general kernel below walks its chains one LDS
// round trip per input: 144 us per 64,000 pairs where this form needs ~15.)
__global__ __launch_bounds__(kThreads) void pacrr_head32_lists_kernel(PacrrArgs a, int p0, int n) {
  constexpr int TPP = 32, kSlots = kThreads / TPP;
  __shared__ __attribute__((aligned(16))) float feat_s[kSlots][32], h1_s[kSlots][32], h2_s[kSlots][32];
  const int tid = threadIdx.x, slot = tid / TPP, o = tid % TPP;
  const int n_ng = a.maxgram - a.mingram + 1, qts = n_ng * a.kmax + (a.use_idf ? 1 : 0), nin = a.Q * qts;
  const int oc = o < a.C ? o : a.C - 1;
  float w1r[32], w2r[32], w3r[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    w1r[i] = i < nin ? a.w1[oc * nin + i] : 0.f;
    w2r[i] = i < a.C ? a.w2[oc * a.C + i] : 0.f;
    w3r[i] = i < a.C ? a.w3[i] : 0.f;
  }
  const float b1v = a.b1[oc], b2v = a.b2[oc], b3v = a.b3[0];
  auto request = [&](int pr, float& f, float (&idf)[kQT]) {      // thread o: feature o of the pair; its first thread: the query's idf row
    const int b = p0 + (pr < n ? pr : n - 1);
    f = o < nin ? a.feats[(int64_t)b * kPacrrMaxFeat + o] : 0.f;
    if (a.use_idf && o == 0) {
      const PairIds ids = pair_ids(a.ids, b, a.Q, a.L);
      const float* idf_g = a.idf + (int64_t)ids.qrow * a.Q;
#pragma unroll
      for (int t = 0; t < kQT; ++t) idf[t] = idf_g[t < a.Q ? t : a.Q - 1];
    }
  };
  float f_next, idf_next[kQT] = {0.f, 0.f, 0.f, 0.f};
  request(blockIdx.x * kSlots + slot, f_next, idf_next);
  for (int base = blockIdx.x * kSlots; base < n; base += gridDim.x * kSlots) {
    const int pr = base + slot;
    const bool live = pr < n;
    const float f = f_next;
    float idf[kQT];
#pragma unroll
    for (int t = 0; t < kQT; ++t) idf[t] = idf_next[t];
    if (base + gridDim.x * kSlots < n) request(pr + gridDim.x * kSlots, f_next, idf_next);
    __syncthreads();      // (the previous trip's h2 has been read)
    if (!(a.use_idf && o < nin && o % qts == qts - 1)) feat_s[slot][o] = f;     // (the idf channel's slots are its first thread's, below)
    if (a.use_idf && o == 0) {   // softmax over the raw idf values of the query (PACRR.py:48-50), as pacrr_head computes it
      float m = idf[0];
#pragma unroll
      for (int q = 1; q < kQT; ++q)
        if (q < a.Q) m = fmaxf(m, idf[q]);
      float den = 0.f;
#pragma unroll
      for (int q = 0; q < kQT; ++q)
        if (q < a.Q) den += expf(idf[q] - m);
#pragma unroll
      for (int q = 0; q < kQT; ++q)
        if (q < a.Q) feat_s[slot][q * qts + qts - 1] = expf(idf[q] - m) / den;
    }
    __syncthreads();
    {
      float s = b1v;
#pragma unroll
      for (int i4 = 0; i4 < 8; ++i4) {
        const float4 x = *reinterpret_cast<const float4*>(&feat_s[slot][4 * i4]);
        s = __builtin_fmaf(w1r[4 * i4], x.x, s);
        s = __builtin_fmaf(w1r[4 * i4 + 1], x.y, s);
        s = __builtin_fmaf(w1r[4 * i4 + 2], x.z, s);
        s = __builtin_fmaf(w1r[4 * i4 + 3], x.w, s);
      }
      h1_s[slot][o] = o < a.C ? pacrr_act(s, a.nonlin) : 0.f;
    }
    __syncthreads();
    {
      float s = b2v;
#pragma unroll
      for (int i4 = 0; i4 < 8; ++i4) {
        const float4 x = *reinterpret_cast<const float4*>(&h1_s[slot][4 * i4]);
        s = __builtin_fmaf(w2r[4 * i4], x.x, s);
        s = __builtin_fmaf(w2r[4 * i4 + 1], x.y, s);
        s = __builtin_fmaf(w2r[4 * i4 + 2], x.z, s);
        s = __builtin_fmaf(w2r[4 * i4 + 3], x.w, s);
      }
      h2_s[slot][o] = o < a.C ? pacrr_act(s, a.nonlin) : 0.f;
    }
    __syncthreads();
    if (live && o == 0) {
      float s = b3v;
#pragma unroll
      for (int i4 = 0; i4 < 8; ++i4) {
        const float4 x = *reinterpret_cast<const float4*>(&h2_s[slot][4 * i4]);
        s = __builtin_fmaf(w3r[4 * i4], x.x, s);
        s = __builtin_fmaf(w3r[4 * i4 + 1], x.y, s);
        s = __builtin_fmaf(w3r[4 * i4 + 2], x.z, s);
        s = __builtin_fmaf(w3r[4 * i4 + 3], x.w, s);
      }
      a.out[p0 + pr] = s;
    }
  }
}

template <int TPP>
__global__ __launch_bounds__(kThreads) void pacrr_head_lists_kernel(PacrrArgs a, int p0, int n) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int kSlots = kThreads / TPP;
  const int tid = threadIdx.x, slot = tid / TPP, o = tid % TPP;
  const int n_ng = a.maxgram - a.mingram + 1, qts = n_ng * a.kmax + (a.use_idf ? 1 : 0), nin = a.Q * qts;
  float* hw = reinterpret_cast<float*>(smem_raw);
  float* feat = hw + ((pacrr_head_floats(a, nin) + 3) & ~3) + slot * (kPacrrMaxFeat + 2 * kPacrrMaxC);
  float* h1 = feat + kPacrrMaxFeat;
  float* h2 = h1 + kPacrrMaxC;
  pacrr_stage_head(a, nin, hw, tid);
  const float *w1 = hw, *w2 = w1 + a.C * nin, *w3 = w2 + a.C * a.C, *b1 = w3 + a.C, *b2 = b1 + a.C, *b3 = b2 + a.C;
  for (int base = blockIdx.x * kSlots; base < n; base += gridDim.x * kSlots) {
    const int pr = base + slot, b = p0 + pr;
    const bool live = pr < n;
    __syncthreads();      // (the staged weights; the previous trip's h2 has been read)
    if (live) {
      for (int i = o; i < nin; i += TPP)
        if (!(a.use_idf && i % qts == qts - 1)) feat[i] = a.feats[(int64_t)b * kPacrrMaxFeat + i];     // (the idf channel's slots: below)
      if (a.use_idf && o == 0) {   // softmax over the raw idf values of the query (PACRR.py:48-50), as pacrr_head computes it
        const PairIds ids = pair_ids(a.ids, b, a.Q, a.L);
        const float* idf_g = a.idf + (int64_t)ids.qrow * a.Q;
        float idf[kPacrrMaxQ];
#pragma unroll
        for (int q = 0; q < kPacrrMaxQ; ++q) idf[q] = idf_g[q < a.Q ? q : a.Q - 1];
        float m = idf[0];
#pragma unroll
        for (int q = 1; q < 8; ++q)
          if (q < a.Q) m = fmaxf(m, idf[q]);
        float den = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (q < a.Q) den += expf(idf[q] - m);
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (q < a.Q) feat[q * qts + qts - 1] = expf(idf[q] - m) / den;
      }
    }
    __syncthreads();
    if (live && o < a.C) {
      float s = b1[o];
      for (int i = 0; i < nin; ++i) s = __builtin_fmaf(w1[o * nin + i], feat[i], s);
      h1[o] = pacrr_act(s, a.nonlin);
    }
    __syncthreads();
    if (live && o < a.C) {
      float s = b2[o];
      for (int i = 0; i < a.C; ++i) s = __builtin_fmaf(w2[o * a.C + i], h1[i], s);
      h2[o] = pacrr_act(s, a.nonlin);
    }
    __syncthreads();
    if (live && o == 0) {
      float s = b3[0];
      for (int i = 0; i < a.C; ++i) s = __builtin_fmaf(w3[i], h2[i], s);
      a.out[b] = s;
    }
  }
}

template <int NV, int KM>
__global__ __launch_bounds__(kThreads, CAPAMD_PACRR_WAVES) void pacrr_mfma_kernel(PacrrArgs a) {
  pacrr_mfma_body<NV, KM>(a, blockIdx.x, nullptr);
}

template <int NV, int KM>
__global__ __launch_bounds__(kThreads, CAPAMD_PACRR_WAVES) void pacrr_mfma4_kernel(PacrrArgs a) {
  pacrr_mfma4_body<NV, KM>(a, blockIdx.x, nullptr);
}

// whole candidate lists: a workgroup per (list, document) in the XCD-aware numbering of lists.h
template <int NV, int KM>
__global__ __launch_bounds__(kThreads, CAPAMD_PACRR_WAVES) void pacrr_mfma_lists_kernel(PacrrArgs a, ListsArgs la, ListGeom g) {
  int l, doc;
  if (!list_doc_of(la, l, doc) || doc >= g.len[l]) return;
  pacrr_mfma4_body<NV, KM>(a, g.start[l] + doc, la.table + (int64_t)l * la.Vp);     // (Q <= kQT = 4 on this route)
}

}  // namespace

static bool pacrr_force_valu() {   // CAPAMD_PACRR_VALU=1: the general kernel for every geometry (A/B measurements, tests)
  const char* e = getenv("CAPAMD_PACRR_VALU");
  return e && e[0] == '1';
}

extern "C" int capamd_pacrr_forward(const int64_t* q_ids, const int64_t* d_ids, const float* idf, int B, int Q, int L, const float* packed,
                                    int64_t V, int D, int mingram, int maxgram, int nfilters, int kmax, const float* conv_w,
                                    const float* conv_b, int use_idf, int combine, int nonlinearity, const float* w1, const float* b1,
                                    const float* w2, const float* b2, const float* w3, const float* b3, float* out, int* status,
                                    void* stream) {
  if (B == 0) return CAPAMD_OK;
  if (!q_ids || !d_ids || !packed || !conv_w || !conv_b || !w1 || !b1 || !w2 || !b2 || !w3 || !b3 || !out || !status) return CAPAMD_ERR_ARG;
  if (use_idf && !idf) return CAPAMD_ERR_ARG;
  if (B < 0 || Q < 1 || Q > kPacrrMaxQ || L < 1 || L > 1024 || V < 1 || V > 0x7fffffffLL) return CAPAMD_ERR_ARG;
  if (mingram < 1 || maxgram < mingram || maxgram > kPacrrMaxGram || nfilters < 1 || nfilters > 256) return CAPAMD_ERR_ARG;
  if (kmax < 1 || kmax > kPacrrMaxK || kmax > L || combine < 1 || combine > kPacrrMaxC || nonlinearity < 0 || nonlinearity > 2) return CAPAMD_ERR_ARG;
  if (capamd_packed_row_stride(D) < 0) return CAPAMD_ERR_ARG;
  int ncw = 0;
  for (int ng = mingram; ng <= maxgram; ++ng) ncw += nfilters * ng * ng;
  const IdSource ids{q_ids, d_ids, nullptr, nullptr, nullptr, nullptr};
  PacrrArgs a{ids, idf, B, Q, L, packed, V, mingram, maxgram, nfilters, kmax, conv_w, conv_b, ncw, use_idf ? 1 : 0, combine, nonlinearity,
              w1, b1, w2, b2, w3, b3, out, status, nullptr};
  hipStream_t s = (hipStream_t)stream;
  (void)hipGetLastError();
  const size_t tail = (size_t)(ncw + (maxgram - mingram + 1) * nfilters) * 4 + (size_t)(kPacrrMaxFeat + 2 * kPacrrMaxC + 8) * 4 + 16 +
                      (size_t)kQT * kMaxNV * 16 * 16;
  if (Q <= kMfmaMaxQ && nfilters <= 32 && !pacrr_force_valu()) {
    const size_t smem = (size_t)pacrr_mfma_region0(L, ncw + (maxgram - mingram + 1) * nfilters) + (size_t)(((L + 63) & ~63) + 4) * 32 + 192 +
                        (size_t)kQT * kMaxNV * 16 * 16;
#define LAUNCH_M(NV_)                                                                                                           \
  do {                                                                                                                          \
    auto k = Q <= 4 ? (kmax <= 2 ? pacrr_mfma4_kernel<NV_, 2> : pacrr_mfma4_kernel<NV_, kPacrrMaxK>)                           \
                    : (kmax <= 2 ? pacrr_mfma_kernel<NV_, 2> : pacrr_mfma_kernel<NV_, kPacrrMaxK>);                            \
    if (smem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
    hipLaunchKernelGGL(k, dim3(B), dim3(kThreads), smem, s, a);                                                                 \
  } while (0)
    switch (nv_for_dim(D)) {
      case 1: LAUNCH_M(1); break;
      case 2: LAUNCH_M(2); break;
      case 3: LAUNCH_M(3); break;
      case 4: LAUNCH_M(4); break;
      default: LAUNCH_M(5); break;
    }
#undef LAUNCH_M
    return hipGetLastError() == hipSuccess ? CAPAMD_OK : CAPAMD_ERR_LAUNCH;
  }
  const int ppl = L <= 256 ? 4 : (L <= 512 ? 8 : (L <= 832 ? 13 : 16));
  const size_t smem = (size_t)((L + 3) & ~3) * 8 + (size_t)(Q + kPacrrMaxGram - 1) * (64 * ppl + kPacrrMaxGram) * 4 + tail;
  if (smem > 160 * 1024) return CAPAMD_ERR_ARG;
#define LAUNCH(NV_, PPL_)                                                                                                       \
  do {                                                                                                                          \
    auto k = pacrr_forward_kernel<NV_, PPL_>;                                                                                   \
    if (smem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
    hipLaunchKernelGGL(k, dim3(B), dim3(kThreads), smem, s, a);                                                                 \
  } while (0)
#define LAUNCH_P(NV_)                               \
  switch (ppl) {                                    \
    case 4: LAUNCH(NV_, 4); break;                  \
    case 8: LAUNCH(NV_, 8); break;                  \
    case 13: LAUNCH(NV_, 13); break;                \
    default: LAUNCH(NV_, 16); break;                \
  }
  switch (nv_for_dim(D)) {
    case 1: LAUNCH_P(1); break;
    case 2: LAUNCH_P(2); break;
    case 3: LAUNCH_P(3); break;
    case 4: LAUNCH_P(4); break;
    default: LAUNCH_P(5); break;
  }
#undef LAUNCH_P
#undef LAUNCH
  return hipGetLastError() == hipSuccess ? CAPAMD_OK : CAPAMD_ERR_LAUNCH;
}

/* PACRR over whole candidate lists (lists.h): mark -> sims (every distinct term of a list gathered once) -> the MFMA kernel with a
 * table lookup per position as its front end.  Q <= 4, nfilters <= 32 (the MFMA kernel's geometry); scores bit-identical to
 * capamd_pacrr_forward's. */
extern "C" int capamd_pacrr_forward_lists(const int64_t* q_ids, const int64_t* d_ids, const int32_t* q_table, const int32_t* d_table,
                                          const int32_t* pair_q, const int32_t* pair_d, const float* idf, const int64_t* list_offsets_host,
                                          int n_lists, int Q, int L, const float* packed, int64_t V, int D, int mingram, int maxgram, int nfilters,
                                          int kmax, const float* conv_w, const float* conv_b, int use_idf, int combine, int nonlinearity,
                                          const float* w1, const float* b1, const float* w2, const float* b2, const float* w3, const float* b3,
                                          float* out, int* status, void* workspace, size_t workspace_bytes, void* stream) {
  if (n_lists == 0) return CAPAMD_OK;
  const bool indexed = q_table != nullptr;
  if (indexed ? (!d_table || !pair_q || !pair_d) : (!q_ids || !d_ids)) return CAPAMD_ERR_ARG;
  if (!packed || !conv_w || !conv_b || !w1 || !b1 || !w2 || !b2 || !w3 || !b3 || !out || !status) return CAPAMD_ERR_ARG;
  if (use_idf && !idf) return CAPAMD_ERR_ARG;
  if (Q < 1 || Q > kQT || L < 1 || L > 1024 || nfilters < 1 || nfilters > 32) return CAPAMD_ERR_ARG;
  if (mingram < 1 || maxgram < mingram || maxgram > kPacrrMaxGram) return CAPAMD_ERR_ARG;
  if (kmax < 1 || kmax > kPacrrMaxK || kmax > L || combine < 1 || combine > kPacrrMaxC || nonlinearity < 0 || nonlinearity > 2) return CAPAMD_ERR_ARG;
  int ncw = 0;
  for (int ng = mingram; ng <= maxgram; ++ng) ncw += nfilters * ng * ng;
  const IdSource ids = indexed ? IdSource{nullptr, nullptr, q_table, d_table, pair_q, pair_d} : IdSource{q_ids, d_ids, nullptr, nullptr, nullptr, nullptr};
  PacrrArgs a{ids, idf, 0, Q, L, packed, V, mingram, maxgram, nfilters, kmax, conv_w, conv_b, ncw, use_idf ? 1 : 0, combine, nonlinearity,
              w1, b1, w2, b2, w3, b3, out, status, nullptr};
  hipStream_t s = (hipStream_t)stream;
  const size_t smem = (size_t)pacrr_mfma_region0(L, ncw + (maxgram - mingram + 1) * nfilters) + (size_t)(((L + 63) & ~63) + 4) * 32 + 192 +
                      (size_t)kQT * kMaxNV * 16 * 16;
  // The combine layers in one pass behind the convolutions (pacrr_head_lists_kernel) when the workspace has room for the pairs' features
  // BEHIND what the lists in flight need (416 B per pair; a workspace sized by capamd_lists_workspace_bytes with the call's n_pairs has
  // 4 L + 32 per pair that this entry does not otherwise use) and the head's weights fit a workgroup's LDS; else every pair's own head at
  // the end of the convolution kernel.  Same scores bit for bit either way: a matter of speed only.
  const int qts = (maxgram - mingram + 1) * kmax + (use_idf ? 1 : 0), nin = Q * qts;
  const size_t head_floats = (size_t)combine * nin + (size_t)combine * combine + 3 * (size_t)combine + 1;
  const int tpp = combine <= 32 ? 32 : combine <= 64 ? 64 : 128;
  const size_t head_smem = (((head_floats + 3) & ~(size_t)3) + (size_t)(kThreads / tpp) * (kPacrrMaxFeat + 2 * kPacrrMaxC)) * 4;
  if (list_offsets_host && workspace && head_smem <= 48 * 1024) {
    const int64_t n_pairs = list_offsets_host[n_lists];
    const size_t feat_bytes = ((size_t)(n_pairs > 0 ? n_pairs : 0) * kPacrrMaxFeat * 4 + 15) & ~(size_t)15;
    const size_t lists_need = capamd_lists_workspace_bytes(n_lists, V, 0, L);
    if (n_pairs > 0 && lists_need > 0 && workspace_bytes >= lists_need + feat_bytes + 16) {
      const size_t at = (workspace_bytes - feat_bytes) & ~(size_t)15;
      a.feats = reinterpret_cast<float*>(static_cast<char*>(workspace) + at);
      workspace_bytes = at;
    }
  }
  return lists_run(ids, list_offsets_host, n_lists, Q, L, packed, V, D, status, workspace, workspace_bytes, s, nullptr, 0, nullptr, nullptr, 0, false, nullptr, kQT,
                   [&](const ListsArgs& la, const ListGeom& g, int nl, int longest) {
#define LAUNCH_L(NV_)                                                                                                           \
  do {                                                                                                                          \
    auto k = kmax <= 2 ? pacrr_mfma_lists_kernel<NV_, 2> : pacrr_mfma_lists_kernel<NV_, kPacrrMaxK>;                           \
    if (smem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
    hipLaunchKernelGGL(k, list_doc_grid(nl, longest), dim3(kThreads), smem, s, a, la, g);                                       \
  } while (0)
                     switch (nv_for_dim(D)) {
                       case 1: LAUNCH_L(1); break;
                       case 2: LAUNCH_L(2); break;
                       case 3: LAUNCH_L(3); break;
                       case 4: LAUNCH_L(4); break;
                       default: LAUNCH_L(5); break;
                     }
#undef LAUNCH_L
                     if (a.feats) {       // the group's pairs are contiguous: lists laid out one after the other
                       const int p0 = g.start[0], n = g.start[nl - 1] + g.len[nl - 1] - p0;
                       const int slots = kThreads / tpp, trips = (n + slots - 1) / slots;
                       const dim3 hg((unsigned)(trips < 2048 ? trips : 2048));
                       if (nin <= 32 && combine <= 32) hipLaunchKernelGGL(pacrr_head32_lists_kernel, dim3((unsigned)(trips < 1024 ? trips : 1024)), dim3(kThreads), 0, s, a, p0, n);
                       else if (tpp == 32) hipLaunchKernelGGL(pacrr_head_lists_kernel<32>, hg, dim3(kThreads), head_smem, s, a, p0, n);
                       else if (tpp == 64) hipLaunchKernelGGL(pacrr_head_lists_kernel<64>, hg, dim3(kThreads), head_smem, s, a, p0, n);
                       else hipLaunchKernelGGL(pacrr_head_lists_kernel<128>, hg, dim3(kThreads), head_smem, s, a, p0, n);
                     }
                   });
}
