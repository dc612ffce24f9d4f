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
// The convolutions are ~1.4 MFLOP of fp32 VALU per pair next to ~1 MB of gathered rows: the kernel stays gather-bound.
#include "capreolus_amd.h"
#include "interaction.cuh"

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
};

__device__ __forceinline__ float pacrr_act(float x, int nonlin) { return nonlin == 1 ? fmaxf(x, 0.f) : (nonlin == 2 ? tanhf(x) : x); }

// PPL = document positions per lane in the back end (64 * PPL >= L)
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

  const int tid = threadIdx.x, lane16 = tid & 15, g = tid >> 4, wave = tid >> 6, lane = tid & 63;
  const int b = blockIdx.x;
  const PairIds ids = pair_ids(a.ids, b, a.Q, a.L);
  const int n_ng = a.maxgram - a.mingram + 1, qts = n_ng * a.kmax + (a.use_idf ? 1 : 0);

  for (int i = tid; i < QS * LS; i += kThreads) sim[i] = 0.f;
  for (int i = tid; i < a.n_conv_w; i += kThreads) wts[i] = a.conv_w[i];
  for (int i = tid; i < n_ng * a.nfilters; i += kThreads) wts[a.n_conv_w + i] = a.conv_b[i];

  // ---- compact the real document terms (id > 0) in document order, with their positions ----
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
      pos[slot] = j;
    }
    n_real += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
  }

  // ---- similarity matrix: kQT query terms per pass ----
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
              if (qp.id[t] == (int)did && did > -2147483648LL) sim[(q0 + t) * LS + j] = 1.f;
          }
        }
    }
    for (int t0 = g; t0 < n_real; t0 += kGroupsPerWG) {
      RowRegs<NV> d[1];
      load_row<NV>(a.packed, tok[t0], lane16, d[0]);
      float x[1];
      int qoff = 0;
      asm volatile("" : "+v"(qoff));
      rows_sim_my<NV, 1, true>(d, qp, qlds + qoff, lane16, x);
      if (lane16 < kQT && q0 + lane16 < a.Q) sim[(q0 + lane16) * LS + pos[t0]] = x[0];   // lane l of a group owns query term l & 3
    }
    __syncthreads();
  }

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
      for (int r = 0; r < PPL; ++r) {
        float v = (j0 + r < a.L) ? best[r] : -INFINITY;
#pragma unroll
        for (int i = 0; i < kPacrrMaxK; ++i) {
          const float hi = fmaxf(top[i], v);
          v = fminf(top[i], v);
          top[i] = hi;
        }
      }
      int head = 0;
      for (int r = 0; r < a.kmax; ++r) {
        float cand = -INFINITY;
#pragma unroll
        for (int i = 0; i < kPacrrMaxK; ++i)
          if (i == head) cand = top[i];
        float bst = cand;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) bst = fmaxf(bst, __shfl_xor(bst, o, 64));
        const unsigned long long who = __ballot(cand == bst && bst > -INFINITY);
        if (who == 0) break;
        if (lane == __ffsll((long long)who) - 1) ++head;
        if (lane == 0) feat[q * qts + gi * a.kmax + r] = bst;
      }
    }
  }
  if (a.use_idf && tid == 0) {   // softmax over the raw idf values of the query (PACRR.py:48-50)
    const float* idf = a.idf + (int64_t)ids.qrow * a.Q;
    float m = idf[0];
    for (int q = 1; q < a.Q; ++q) m = fmaxf(m, idf[q]);
    float den = 0.f;
    for (int q = 0; q < a.Q; ++q) den += expf(idf[q] - m);
    for (int q = 0; q < a.Q; ++q) feat[q * qts + qts - 1] = expf(idf[q] - m) / den;
  }
  __syncthreads();

  // ---- combine: Linear(Q*qts, C) -> nonlin -> Linear(C, C) -> nonlin -> Linear(C, 1) ----
  const int nin = a.Q * qts;
  if (tid < a.C) {
    float s = a.b1[tid];
    for (int i = 0; i < nin; ++i) s = __builtin_fmaf(a.w1[tid * nin + i], feat[i], s);
    h1[tid] = pacrr_act(s, a.nonlin);
  }
  __syncthreads();
  if (tid < a.C) {
    float s = a.b2[tid];
    for (int i = 0; i < a.C; ++i) s = __builtin_fmaf(a.w2[tid * a.C + i], h1[i], s);
    h2[tid] = pacrr_act(s, a.nonlin);
  }
  __syncthreads();
  if (tid == 0) {
    float s = a.b3[0];
    for (int i = 0; i < a.C; ++i) s = __builtin_fmaf(a.w3[i], h2[i], s);
    a.out[b] = s;
  }
}

}  // namespace

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
              w1, b1, w2, b2, w3, b3, out, status};
  const int ppl = L <= 256 ? 4 : (L <= 512 ? 8 : (L <= 832 ? 13 : 16));
  const size_t smem = (size_t)((L + 3) & ~3) * 8 + (size_t)(Q + kPacrrMaxGram - 1) * (64 * ppl + kPacrrMaxGram) * 4 +
                      (size_t)(ncw + (maxgram - mingram + 1) * nfilters) * 4 + (size_t)(kPacrrMaxFeat + 2 * kPacrrMaxC + 8) * 4 +
                      (size_t)kQT * kMaxNV * 16 * 16;
  if (smem > 160 * 1024) return CAPAMD_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  (void)hipGetLastError();
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
