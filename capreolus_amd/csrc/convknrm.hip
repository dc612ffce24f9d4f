// Fused ConvKNRM forward for gfx950.
//
// Reference semantics: ConvKNRM_class.forward (capreolus/reranker/ConvKNRM.py:42-77): embeddings of query and document ->
// per n-gram size g = 1..G a Conv1d(D -> F, g) over the right-zero-padded sequence -> cosine similarity of every query view
// with every document view (crossmatch) or of equal sizes only (StackedSimilarityMatrix, common.py:195-221; 0 where the query
// or document token at the position is pad) -> the KNRM kernel pooling over each view -> Linear / (Linear, tanh, Linear).
//
// MI355X design: the convolutions never run at scoring time.  The embedding table is frozen (ConvKNRM.py:17, non_trainable) and
// a Conv1d is linear in its taps, so rep_g[j] = sum_c W_g[:, :, c] E[tok[j + c]] + b_g is a sum of per-token projections.
// capamd_convknrm_pack_tables computes them once for the whole vocabulary - G(G+1)/2 parts of F floats per token, 1.2 GB for
// 400k tokens x 6 parts x 128 filters, a small corner of the 288 GB of HBM - and a position's three n-gram vectors become a
// gather of 3 KB (6 parts from 3 adjacent tokens) plus 3 vector adds instead of 0.46 MFLOP of convolution: 368 MFLOP per
// 800-term document drop to ~1 MB of gathered rows, the same kind of work as KNRM's own front end.
//   parts of token t, in this order so that what a position needs from each of its tokens is contiguous:
//     tap 0 of g = 1..G (bias folded in) | tap 1 of g = 2..G | tap 2 of g = 3
// One workgroup (4 waves) per pair; document positions whose own token is pad are never touched (their similarity is exactly 0
// in every view: closed-form kernel contribution, as in KNRM).  Real positions go through tiles of 16 (one per 16-lane group):
//   A  gather + add + L2-normalise (16 lanes per position), two-term f16 split, transposed into LDS [view][position][F];
//      the 12 loads of the NEXT tile's position are issued before phases B and C of the current one and stay in flight
//      across them (48 registers), so the gather latency is covered by the tile's own arithmetic
//   B  similarities on the matrix pipe: v_mfma_f32_16x16x32_f16, M = 16 of the G*Q normalised query vectors, N = 16
//      positions, K = F; hi/lo split of both operands (3 products, fp32 accumulate, ~2^-22 relative); one wave per
//      (document view, block of 16 query vectors)
//   C  kernel pooling: every thread owns one (view, query term) row and a slice of the tile's positions; K exp2 per value
// and a fixed-order reduction, log, query sum and the combine layers at the end.
#include "capreolus_amd.h"
#include "interaction.h"
#include "lists.h"      // whole candidate lists: geometry, the clear / mark passes (capamd_convknrm_forward_lists below)

using namespace capamd;

namespace {

constexpr int kCkMaxG = 3;
constexpr int kCkMaxQ = 8;
constexpr int kCkMaxK = 11;
constexpr int kCkMaxF = 128;
constexpr int kCkMaxH = 64;
constexpr int kCkTile = 16;
constexpr int kCkMaxRows = kCkMaxG * kCkMaxG * kCkMaxQ;   // (view, query term) rows
constexpr int kCkMaxTpr = 16;                             // threads per row in the pooling phase
constexpr int kCkPoolCols = 2 * kCkTile;                  // the pooling phase runs once per two tiles
#ifndef CAPAMD_CK_WAVES
#define CAPAMD_CK_WAVES 3    // waves per SIMD the register budget allows (164 registers at 128 filters); 4 measured: see DESIGN.md §6 N4
#endif
#ifndef CAPAMD_CK_ABLATE
#define CAPAMD_CK_ABLATE 0   // profiling builds only, bit mask: 1 = no gather (phase A), 2 = no MFMA (phase B), 4 = no pooling (phase C)
#endif

typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(4))) _Float16 h4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__host__ __device__ inline int ck_parts(int G) { return G * (G + 1) / 2; }
// part index of tap c of n-gram size g (1-based)
__host__ __device__ inline int ck_part(int G, int g, int c) { return (c == 0 ? 0 : (c == 1 ? G : 2 * G - 1)) + (g - 1 - c); }

// ---- table pack: tables[t][part(g, c)][f] = (c == 0 ? bias_g[f] : 0) + sum_d E[t][d] * W_g[f][d][c] ----
constexpr int kPackTokens = 16;

__global__ __launch_bounds__(256) void convknrm_pack_kernel(const float* __restrict__ emb, int64_t V, int D, int64_t ld,
                                                            const float* __restrict__ conv_w, const float* __restrict__ conv_b, int G,
                                                            int F, float* __restrict__ tables) {
  extern __shared__ __attribute__((aligned(16))) float e_t[];   // [D][16 tokens]
  const int64_t t0 = (int64_t)blockIdx.x * kPackTokens;
  for (int i = threadIdx.x; i < D * kPackTokens; i += blockDim.x) {
    const int t = i / D, d = i - t * D;
    e_t[d * kPackTokens + t] = (t0 + t < V) ? emb[(t0 + t) * ld + d] : 0.f;
  }
  __syncthreads();
  const int P = ck_parts(G);
  for (int o = threadIdx.x; o < P * F; o += blockDim.x) {
    const int p = o / F, f = o - p * F;
    const int c = p < G ? 0 : (p < 2 * G - 1 ? 1 : 2);
    const int g = p - (c == 0 ? 0 : (c == 1 ? G : 2 * G - 1)) + 1 + c;
    int64_t woff = 0;
    for (int gg = 1; gg < g; ++gg) woff += (int64_t)F * D * gg;
    const float* w = conv_w + woff + (int64_t)f * D * g + c;
    float acc[kPackTokens];
    const float b0 = c == 0 ? conv_b[(g - 1) * F + f] : 0.f;
#pragma unroll
    for (int t = 0; t < kPackTokens; ++t) acc[t] = b0;
    for (int d = 0; d < D; ++d) {
      const float wv = w[(int64_t)d * g];
      const float4* e4 = reinterpret_cast<const float4*>(e_t + d * kPackTokens);
#pragma unroll
      for (int t4 = 0; t4 < kPackTokens / 4; ++t4) {
        const float4 e = e4[t4];
        acc[t4 * 4 + 0] = __builtin_fmaf(e.x, wv, acc[t4 * 4 + 0]);
        acc[t4 * 4 + 1] = __builtin_fmaf(e.y, wv, acc[t4 * 4 + 1]);
        acc[t4 * 4 + 2] = __builtin_fmaf(e.z, wv, acc[t4 * 4 + 2]);
        acc[t4 * 4 + 3] = __builtin_fmaf(e.w, wv, acc[t4 * 4 + 3]);
      }
    }
#pragma unroll
    for (int t = 0; t < kPackTokens; ++t)
      if (t0 + t < V) tables[((t0 + t) * P + p) * F + f] = acc[t];
  }
}

// byte offsets of the workgroup's LDS regions (host: size of the launch; device: the carve-up)
struct CkLayout {
  int tok, pos, d_hi, d_lo, q_hi, q_lo, sims, kr, misc, total;
};

__host__ __device__ inline CkLayout ck_layout(int L, int F, int G, int Q, int views) {
  const int lcap = (L + 7) & ~7, RS = F + 8, R = views * Q;
  CkLayout l;
  int o = 0;
  l.tok = o, o += lcap * 4;
  l.pos = o, o += lcap * 2;
  const int planes = 2 * G * kCkTile * RS * 2, part = kThreads * (kCkMaxK + 1) * 4;   // the partial sums reuse the planes at the end
  l.d_hi = o, l.d_lo = o + planes / 2, o += ((planes > part ? planes : part) + 15) & ~15;
  l.q_hi = o, o += ((G * Q * RS * 2) + 15) & ~15;
  l.q_lo = o, o += ((G * Q * RS * 2) + 15) & ~15;
  l.sims = o, o += R * kCkPoolCols * 4;
  l.kr = o, o += ((R * kCkMaxK * 4) + 15) & ~15;
  l.misc = o, o += (3 * (kCkMaxK + 1) + kCkMaxK * kCkMaxG * kCkMaxG + 1 + kCkMaxH + kCkMaxQ + kCkMaxG * 32 + 4) * 4;
  l.total = (o + 15) & ~15;
  return l;
}

struct ConvKnrmArgs {
  const int64_t* q_ids;
  const int64_t* d_ids;
  int B, Q, L;
  const float* tables;
  int64_t V;
  int G, F, crossmatch;
  const float* mu;
  const float* sigma;
  int K;
  const float *w1, *b1;
  int H;
  const float *w2, *b2;
  int score_tanh;
  float* out;
  int* status;
  // whole candidate lists (capamd_convknrm_forward_lists): the unigram document view's similarities come from a per-list table
  const float* ltab;      // [lists][Vp][TS] sims of a token's unigram vector with the list's G * Q query vectors (ck_lists_sims_kernel)
  int64_t Vp;
  int TS;                 // floats per table entry: G * Q rounded up to 4
};

// The lane's share (NF4 float4 per part) of what one position gathers: tap 0 from its own token t0, tap 1 from t1, tap 2 from t2
// (-1: beyond the sequence, contributes nothing).  Loads only: the sums are taken when the tile is built.
template <int NF4>
struct CkGather {
  float4 c0[kCkMaxG][NF4], c1[kCkMaxG][NF4], c2[NF4];
  int taps;   // bit 0: tap 1 exists (t1 >= 0), bit 1: tap 2 exists
};

// Loads only, each from a clamped, always valid address: nothing here depends on the loaded data, so the loads stay in flight
// until ck_sum - one tile later - masks and adds them.  (A select between a global address and a zero constant would also
// turn the loads into flat loads through a scratch copy of the constant.)
// SKIP1 (list route, document positions): the unigram view (g = 1) is not gathered - its similarities come from the list's table.
template <int NF4, bool SKIP1 = false>
__device__ __forceinline__ void ck_load(const ConvKnrmArgs& a, int t0, int t1, int t2, int lane16, CkGather<NF4>& r) {
  const int F4 = a.F >> 2, P = ck_parts(a.G);
  const float4* r0 = reinterpret_cast<const float4*>(a.tables) + (int64_t)t0 * P * F4;
  const float4* r1 = reinterpret_cast<const float4*>(a.tables) + (int64_t)(t1 < 0 ? 0 : t1) * P * F4;
  const float4* r2 = reinterpret_cast<const float4*>(a.tables) + (int64_t)(t2 < 0 ? 0 : t2) * P * F4;
  r.taps = (t1 >= 0 ? 1 : 0) | (t2 >= 0 ? 2 : 0);
#pragma unroll
  for (int g = 1; g <= kCkMaxG; ++g)
#pragma unroll
    for (int i = 0; i < NF4; ++i) {
      const int x = i * 16 + lane16, xc = x < F4 ? x : F4 - 1;
      if (!(SKIP1 && g == 1)) r.c0[g - 1][i] = r0[(g <= a.G ? g - 1 : 0) * F4 + xc];
      if (g >= 2) r.c1[g - 1][i] = r1[(g <= a.G ? a.G + g - 2 : 0) * F4 + xc];
      if (g == 3) r.c2[i] = r2[(g <= a.G ? 2 * a.G - 1 : 0) * F4 + xc];
    }
}

// rep_g = (tap 0 + tap 1) + tap 2; lanes beyond F, n-gram sizes beyond G and taps beyond the sequence end contribute 0
template <int NF4, bool SKIP1 = false>
__device__ __forceinline__ void ck_sum(const ConvKnrmArgs& a, const CkGather<NF4>& r, int lane16, float4 (&rep)[kCkMaxG][NF4]) {
  const int F4 = a.F >> 2;
#pragma unroll
  for (int g = (SKIP1 ? 2 : 1); g <= kCkMaxG; ++g)
#pragma unroll
    for (int i = 0; i < NF4; ++i) {
      const bool act = g <= a.G && i * 16 + lane16 < F4;
      float4 v = r.c0[g - 1][i];
      if (g >= 2) {
        const float4 u = r.c1[g - 1][i];
        v.x += (r.taps & 1) ? u.x : 0.f;
        v.y += (r.taps & 1) ? u.y : 0.f;
        v.z += (r.taps & 1) ? u.z : 0.f;
        v.w += (r.taps & 1) ? u.w : 0.f;
      }
      if (g == 3) {
        const float4 u = r.c2[i];
        v.x += (r.taps & 2) ? u.x : 0.f;
        v.y += (r.taps & 2) ? u.y : 0.f;
        v.z += (r.taps & 2) ? u.z : 0.f;
        v.w += (r.taps & 2) ? u.w : 0.f;
      }
      rep[g - 1][i] = act ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// the query token as the kernel uses it (ids outside [0, V) count as pad; the status word reports them)
__device__ __forceinline__ int qtok_early(const int64_t* qi, int q, int64_t V) {
  const int64_t id = qi[q];
  return (id < 0 || id >= V) ? 0 : (int)id;
}

__device__ __forceinline__ void ck_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// L2-normalise (x / (|x| + 1e-9), common.py:210-213), split into f16 hi + lo and store the lane's share of row `row` of a
// [rows][RS] plane pair.
template <int NF4>
__device__ __forceinline__ void ck_store_unit(const float4 (&v)[NF4], int F4, int lane16, _Float16* hi, _Float16* lo, int row, int RS) {
  // (no contraction: e = v * inv is used twice - rounded to f16, and in e - hi - and hipcc's default -ffp-contract=fast may fold the
  //  second use into an fma in one kernel and not in another; the list route's table and the per-pair kernel must split alike)
#pragma clang fp contract(off)
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NF4; ++i) {
    ss = __builtin_fmaf(v[i].x, v[i].x, ss);
    ss = __builtin_fmaf(v[i].y, v[i].y, ss);
    ss = __builtin_fmaf(v[i].z, v[i].z, ss);
    ss = __builtin_fmaf(v[i].w, v[i].w, ss);
  }
  ss = group_allreduce(ss);
  const float inv = 1.f / (__builtin_sqrtf(ss) + 1e-9f);
#pragma unroll
  for (int i = 0; i < NF4; ++i) {
    const int x = i * 16 + lane16;
    if (x < F4) {
      const float e[4] = {unfused(v[i].x * inv), unfused(v[i].y * inv), unfused(v[i].z * inv), unfused(v[i].w * inv)};      // (a value, not an expression: see above)
      h4 h, l;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        h[j] = (_Float16)e[j];
        l[j] = (_Float16)(e[j] - (float)h[j]);
      }
      *reinterpret_cast<h4*>(hi + row * RS + x * 4) = h;
      *reinterpret_cast<h4*>(lo + row * RS + x * 4) = l;
    }
  }
}

// LISTS: a workgroup per (list, document) in the XCD-aware numbering of lists.h; the unigram document view is looked up, not computed.
template <int NF4, bool LISTS>
__device__ __forceinline__ void convknrm_forward_body(const ConvKnrmArgs& a, int b, const float* ltab_l) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lane16 = tid & 15, grp = tid >> 4;
  const int G = a.G, Q = a.Q, F = a.F, F4 = F >> 2, K = a.K;
  const int views = a.crossmatch ? G * G : G, R = views * Q, GQ = G * Q;
  const int RS = F + 8;                                  // f16 per row of an operand plane: 16 B of padding spreads the rows over the banks

  const CkLayout lay = ck_layout(a.L, F, G, Q, views);
  int* tok = reinterpret_cast<int*>(smem_raw + lay.tok);                 // [lcap] every token of the document
  unsigned short* pos = reinterpret_cast<unsigned short*>(smem_raw + lay.pos);   // [lcap] positions whose own token is real
  _Float16* d_hi = reinterpret_cast<_Float16*>(smem_raw + lay.d_hi);     // [G][16][RS]
  _Float16* d_lo = reinterpret_cast<_Float16*>(smem_raw + lay.d_lo);
  _Float16* q_hi = reinterpret_cast<_Float16*>(smem_raw + lay.q_hi);     // [G*Q][RS]
  _Float16* q_lo = reinterpret_cast<_Float16*>(smem_raw + lay.q_lo);
  float* sims = reinterpret_cast<float*>(smem_raw + lay.sims);           // [R][32]: two tiles side by side
  float* kr = reinterpret_cast<float*>(smem_raw + lay.kr);               // [R][K] log kernel sums
  float* kmu = reinterpret_cast<float*>(smem_raw + lay.misc);            // [12] (mu, coefficient) pairs, zero beyond K
  float* kz = kmu + 2 * (kCkMaxK + 1);                                   // [K] kernel value of a similarity of exactly 0
  float* feat = kz + kCkMaxK + 1;                                        // [K * views]
  float* hid = feat + kCkMaxK * kCkMaxG * kCkMaxG + 1;                   // [H]
  int* qtok = reinterpret_cast<int*>(hid + kCkMaxH);                     // [Q]
  int* rowmap = qtok + kCkMaxQ;                                          // [G][32]: sims row of query vector m for document view g, or -1
  int* wave_cnt = rowmap + kCkMaxG * 32;                                 // [4]
  float* partial = reinterpret_cast<float*>(d_hi);                       // [pooled rows][tpr][kCkMaxK + 1], after the last tile

  const int64_t* qi = a.q_ids + (int64_t)b * Q;
  const int64_t* di = a.d_ids + (int64_t)b * a.L;

  // ---- tokens, compaction of the real positions, constants ----
  if (tid < Q) {
    int64_t id = qi[tid];
    if (id < 0 || id >= a.V) {
      atomicOr(a.status, id < 0 ? kErrQueryOOV : kErrQueryIdRange);
      id = 0;
    }
    qtok[tid] = (int)id;
  }
  if (tid < kCkMaxK + 1) {
    const float m = tid < K ? a.mu[tid] : 0.f, sg = tid < K ? a.sigma[tid] : 1.f;
    const float c = tid < K ? (-0.5f * kLog2e) / (sg * sg) : 0.f;
    kmu[2 * tid] = m;
    kmu[2 * tid + 1] = c;
    kz[tid] = __builtin_amdgcn_exp2f(m * m * c);
  }
  for (int i = tid; i < G * 32; i += kThreads) {
    const int gb = i >> 5, m = i & 31;
    int r = -1;
    if (m < GQ) {
      const int ga = m / Q, q = m - ga * Q;
      if (a.crossmatch) r = (ga * G + gb) * Q + q;
      else if (ga == gb) r = gb * Q + q;
    }
    rowmap[i] = r;
  }
  int n_real = 0;
  for (int base = 0; base < a.L; base += kThreads) {
    const int j = base + tid;
    int64_t id = (j < a.L) ? di[j] : 0;
    if (id < 0 || id >= a.V) {
      atomicOr(a.status, kErrDocIdRange);
      id = 0;
    }
    if (j < a.L) tok[j] = (int)id;
    const bool real = id != 0;
    const unsigned long long m = __ballot(real);
    if (lane == 0) wave_cnt[wave] = __popcll(m);
    __syncthreads();
    int off = n_real;
    for (int w = 0; w < wave; ++w) off += wave_cnt[w];
    if (real) pos[off + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)j;
    n_real += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
  }

  // ---- query vectors: group q builds the G views of query position q ----
  if (grp < Q) {
    const int q = grp;
    CkGather<NF4> gq;
    ck_load<NF4>(a, qtok[q], q + 1 < Q ? qtok[q + 1] : -1, q + 2 < Q ? qtok[q + 2] : -1, lane16, gq);
    float4 rep[kCkMaxG][NF4];
    ck_sum<NF4>(a, gq, lane16, rep);
#pragma unroll
    for (int g = 0; g < kCkMaxG; ++g)
      if (g < G) ck_store_unit<NF4>(rep[g], F4, lane16, q_hi, q_lo, g * Q + q, RS);
  }

  // ---- pooling state: thread -> (row, slice of positions) ----
  // Only rows of real query terms are pooled (a pad term's row is all zero and masked at the end, ConvKNRM.py:72-73): with nq real
  // terms there are views * nq rows, each shared by as many threads as fit.
  int nq = 0, qreal[kCkMaxQ];
  int64_t qid_l[kCkMaxQ];   // (the Q ids requested together - clamped index, unconditional: one memory round trip, not one per term)
#pragma unroll
  for (int q = 0; q < kCkMaxQ; ++q) qid_l[q] = qi[q < Q ? q : Q - 1];
#pragma unroll
  for (int q = 0; q < kCkMaxQ; ++q) {
    qreal[q] = 0;
    if (q < Q && !(qid_l[q] <= 0 || qid_l[q] >= a.V)) {
#pragma unroll
      for (int j = 0; j < kCkMaxQ; ++j)
        if (j == nq) qreal[j] = q;
      ++nq;
    }
  }
  const int racts = views * nq;
  int tpr = racts > 0 ? kThreads / racts : 1;
  if (tpr > kCkMaxTpr) tpr = kCkMaxTpr;
  const int prow = tid / tpr, psub = tid - prow * tpr;
  const bool pool = prow < racts;
  int srow = 0;                                          // the row of `sims` this thread pools
  if (pool) {
    const int v = prow / nq, qi2 = prow - v * nq;
    int q = 0;
#pragma unroll
    for (int j = 0; j < kCkMaxQ; ++j)
      if (j == qi2) q = qreal[j];
    srow = v * Q + q;
  }
  float kacc[kCkMaxK];
#pragma unroll
  for (int k = 0; k < kCkMaxK; ++k) kacc[k] = 0.f;
  float rowsum = 0.f;
  __syncthreads();

  const int n_mb = (GQ + 15) >> 4, n_jobs = G * n_mb;    // phase B jobs: (document view, block of 16 query vectors)
  CkGather<NF4> gat;                                     // the position this group owns in the NEXT tile, in flight
  // list route: thread (position n = tid & 15, query vector m = tid >> 4 [+ 16]) looks the unigram view's similarity up - requested a
  // tile ahead, like the gather
  float tv_cur[2] = {0.f, 0.f}, tv_nxt[2] = {0.f, 0.f};
  auto issue = [&](int base) {
    if (LISTS) {
      // (requested BEFORE the tile's gather: loads return in order, and the lookup is consumed - moved to tv_cur - a phase earlier than
      //  the gathered parts; behind them, that move would wait for the whole gather)
      const int n = tid & 15, m = tid >> 4;
      const int jn = pos[base + n < n_real ? base + n : (n_real > 0 ? n_real - 1 : 0)];
      const float* e = ltab_l + (int64_t)tok[n_real > 0 ? jn : 0] * a.TS;
      tv_nxt[0] = e[m < GQ ? m : 0];
      if (GQ > 16) tv_nxt[1] = e[m + 16 < GQ ? m + 16 : 0];
    }
    if (!(CAPAMD_CK_ABLATE & 1) && base + grp < n_real) {
      const int j = pos[base + grp];
      ck_load<NF4, LISTS>(a, tok[j], j + 1 < a.L ? tok[j + 1] : -1, j + 2 < a.L ? tok[j + 2] : -1, lane16, gat);
    }
  };
  issue(0);
  tv_cur[0] = tv_nxt[0];
  tv_cur[1] = tv_nxt[1];
  for (int base = 0; base < n_real; base += kCkTile) {
    const int nv = min(kCkTile, n_real - base);
    const int col0 = base & kCkTile;                        // even tiles fill columns 0..15 of `sims`, odd tiles 16..31
    // -- A: add, normalise, split (the loads were issued one tile ago) --
    if (!(CAPAMD_CK_ABLATE & 1) && grp < nv) {
      float4 rep[kCkMaxG][NF4];
      ck_sum<NF4, LISTS>(a, gat, lane16, rep);
#pragma unroll
      for (int g = (LISTS ? 1 : 0); g < kCkMaxG; ++g)
        if (g < G) ck_store_unit<NF4>(rep[g], F4, lane16, d_hi + g * kCkTile * RS, d_lo + g * kCkTile * RS, grp, RS);
    }
    ck_lds_barrier();
    issue(base + kCkTile);
    // -- B: similarities of document view gb with query vectors mb * 16 .. +15 --
    if (!(CAPAMD_CK_ABLATE & 2))
      for (int job = wave + (LISTS ? n_mb : 0); job < n_jobs; job += 4) {      // (list route: the jobs of document view 0 are lookups, below)
        const int gb = job / n_mb, mb = job - gb * n_mb;
        const int n = lane & 15, kg = lane >> 4, m = mb * 16 + n;
        const bool arow = m < GQ;
        const _Float16* ah = q_hi + (arow ? m : 0) * RS + 8 * kg;
        const _Float16* al = q_lo + (arow ? m : 0) * RS + 8 * kg;
        const _Float16* bh = d_hi + (gb * kCkTile + n) * RS + 8 * kg;
        const _Float16* bl = d_lo + (gb * kCkTile + n) * RS + 8 * kg;
        f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};   // two chains (even / odd K steps)
        const h8 zero = {0};
#pragma unroll
        for (int pr = 0; pr < kCkMaxF / 64; ++pr)               // two K steps (one per chain) at a time: 8 fragment reads, then 6 MFMAs
          if (pr * 64 < F) {
            h8 a_hi[2], a_lo[2], b_hi[2], b_lo[2];
#pragma unroll
            for (int st = 0; st < 2; ++st) {
              const int kk = pr * 64 + st * 32;
              const bool kin = kk + 8 * kg < F;                 // F is a multiple of 16: the last step may be partly or wholly empty
              a_hi[st] = (arow && kin) ? *reinterpret_cast<const h8*>(ah + kk) : zero;
              a_lo[st] = (arow && kin) ? *reinterpret_cast<const h8*>(al + kk) : zero;
              b_hi[st] = kin ? *reinterpret_cast<const h8*>(bh + kk) : zero;
              b_lo[st] = kin ? *reinterpret_cast<const h8*>(bl + kk) : zero;
            }
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi[0], b_hi[0], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi[1], b_hi[1], c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo[0], b_hi[0], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo[1], b_hi[1], c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi[0], b_lo[0], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi[1], b_lo[1], c1, 0, 0, 0);
          }
        if (n < nv) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int row = mb * 16 + kg * 4 + i;           // query vector of accumulator register i
            const int r = row < 32 ? rowmap[gb * 32 + row] : -1;
            if (r >= 0) sims[r * kCkPoolCols + col0 + n] = c0[i] + c1[i];
          }
        }
      }
    if (LISTS) {      // the unigram document view: the similarities ck_lists_sims_kernel computed for this list's tokens (the same MFMA chain: same bits)
      const int n = tid & 15, m = tid >> 4;
      if (n < nv) {
        const int r0 = m < GQ ? rowmap[m] : -1;
        if (r0 >= 0) sims[r0 * kCkPoolCols + col0 + n] = tv_cur[0];
        if (GQ > 16) {
          const int r1 = m + 16 < GQ ? rowmap[m + 16] : -1;
          if (r1 >= 0) sims[r1 * kCkPoolCols + col0 + n] = tv_cur[1];
        }
      }
      tv_cur[0] = tv_nxt[0];
      tv_cur[1] = tv_nxt[1];
    }
    ck_lds_barrier();
    // -- C: kernel pooling --
    // once per two tiles (or at the last one): 32 columns split over the row's threads balance better than 16
    if (pool && !(CAPAMD_CK_ABLATE & 4) && (col0 != 0 || base + kCkTile >= n_real)) {
      const int ncols = col0 + nv;
      for (int n = psub; n < ncols; n += tpr) {
        const float s = sims[srow * kCkPoolCols + n];
        rowsum += s;
        // (mu, coefficient) pairs are re-read from LDS for every value - broadcast reads, two kernels per 16-byte read - instead
        // of living in 22 registers next to the 48 of the gather in flight (the opaque offset keeps the reads inside the loop:
        // hoisted, they spill, and a scratch reload would have to wait for the whole gather because loads return in order)
        int koff = 0;
        asm volatile("" : "+v"(koff));
        const float4* kc4 = reinterpret_cast<const float4*>(kmu + koff);
#pragma unroll
        for (int k2 = 0; k2 < (kCkMaxK + 1) / 2; ++k2) {        // kernels beyond K have mu = 0, coefficient 0: their sums are never read
          const float4 c = kc4[k2];
          const float a0 = s - c.x, a1 = s - c.z;
          kacc[2 * k2] += __builtin_amdgcn_exp2f(a0 * a0 * c.y);
          if (2 * k2 + 1 < kCkMaxK) kacc[2 * k2 + 1] += __builtin_amdgcn_exp2f(a1 * a1 * c.w);
        }
      }
    }
  }
  __syncthreads();

  // ---- fixed-order reduction over the slices, pads in closed form, log, sum over the query ----
  if (pool) {
    float* p = partial + (prow * tpr + psub) * (kCkMaxK + 1);
#pragma unroll
    for (int k = 0; k < kCkMaxK; ++k) p[k] = kacc[k];
    p[kCkMaxK] = rowsum;
  }
  __syncthreads();
  const float n_zero = (float)(a.L - n_real);
  for (int i = tid; i < racts * K; i += kThreads) {
    const int r = i / K, k = i - r * K;
    float s = 0.f, rs = 0.f;
    for (int u = 0; u < tpr; ++u) {
      s += partial[(r * tpr + u) * (kCkMaxK + 1) + k];
      rs += partial[(r * tpr + u) * (kCkMaxK + 1) + kCkMaxK];
    }
    s = __builtin_fmaf(n_zero, kz[k], s);
    kr[r * K + k] = rs != 0.f ? logf(s + 1e-6f) : 0.f;
  }
  __syncthreads();
  for (int i = tid; i < K * views; i += kThreads) {
    const int k = i / views, v = i - k * views;
    float s = 0.f;
    for (int q = 0; q < nq; ++q) s += kr[(v * nq + q) * K + k];
    feat[i] = s;
  }
  __syncthreads();
  const int nin = K * views;
  if (a.H == 0) {
    if (wave == 0) {
      float s = 0.f;
      for (int i = lane; i < nin; i += 64) s = __builtin_fmaf(a.w1[i], feat[i], s);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
      if (lane == 0) {
        s += a.b1[0];
        a.out[b] = a.score_tanh ? tanhf(s) : s;
      }
    }
  } else {
    if (tid < a.H) {
      float s = a.b1[tid];
      for (int i = 0; i < nin; ++i) s = __builtin_fmaf(a.w1[tid * nin + i], feat[i], s);
      hid[tid] = tanhf(s);
    }
    __syncthreads();
    if (tid == 0) {
      float s = a.b2[0];
      for (int h = 0; h < a.H; ++h) s = __builtin_fmaf(a.w2[h], hid[h], s);
      a.out[b] = a.score_tanh ? tanhf(s) : s;
    }
  }
}

template <int NF4>
__global__ __launch_bounds__(kThreads, CAPAMD_CK_WAVES) void convknrm_forward_kernel(ConvKnrmArgs a) {
  convknrm_forward_body<NF4, false>(a, blockIdx.x, nullptr);
}

// ---- whole candidate lists: the unigram document view once per distinct term of a LIST ------------------------------------------------
//   A document position's three n-gram vectors are 3 KB of gathered projections (6 parts of 512 B); the unigram
// one - part (1, 0) of the position's own token, 512 B - depends on the token alone, and so do its similarities with the list's G x Q
// query vectors.  Per list: the shared clear / mark passes of lists.h flag the list's distinct tokens; ck_lists_sims_kernel walks
// them in tiles of 16 - gather the part, L2-normalise, split into f16 hi + lo, the 16 x 16 x F products of phase B (the same
// instruction sequence on the same operands: bit-identical similarities) - and leaves G * Q floats per token in the list's table; the
// per-pair kernel then gathers 5 parts per position instead of 6, normalises two views instead of three, runs two thirds of the
// MFMAs, and looks the unigram view up (48 B per position, requested a tile ahead).  Same `sims` rows, same pooling: the scores equal
// capamd_convknrm_forward's bit for bit.
struct CkListsArgs {
  ConvKnrmArgs a;
  const uint8_t* flags;    // [lists][Vp]
  float* table;            // [lists][Vp][TS]
  _Float16* qimg;          // [lists][2][G * Q][RS]: the list's normalised query vectors, hi plane then lo plane
  int nl;
};

// the list's query vectors, as the per-pair kernel builds them for every pair
template <int NF4>
__global__ __launch_bounds__(kThreads) void ck_lists_query_kernel(CkListsArgs c, ListGeom g) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const ConvKnrmArgs& a = c.a;
  const int tid = threadIdx.x, lane16 = tid & 15, grp = tid >> 4, l = blockIdx.x;
  const int G = a.G, Q = a.Q, F4 = a.F >> 2, RS = a.F + 8, GQ = G * Q;
  _Float16* q_hi = reinterpret_cast<_Float16*>(smem_raw);
  _Float16* q_lo = q_hi + GQ * RS;
  __shared__ int qtok[kCkMaxQ];
  const int64_t* qi = a.q_ids + (int64_t)g.start[l] * Q;
  if (tid < Q) {
    int64_t id = qi[tid];
    if (id < 0 || id >= a.V) id = 0;        // (the per-pair kernel reports it through the status word)
    qtok[tid] = (int)id;
  }
  for (int i = tid; i < GQ * RS; i += kThreads) q_hi[i] = q_lo[i] = (_Float16)0.f;      // (the 8 padding halves of a row are never read; keep the image defined)
  __syncthreads();
  if (grp < Q) {
    const int q = grp;
    CkGather<NF4> gq;
    ck_load<NF4>(a, qtok[q], q + 1 < Q ? qtok[q + 1] : -1, q + 2 < Q ? qtok[q + 2] : -1, lane16, gq);
    float4 rep[kCkMaxG][NF4];
    ck_sum<NF4>(a, gq, lane16, rep);
#pragma unroll
    for (int gg = 0; gg < kCkMaxG; ++gg)
      if (gg < G) ck_store_unit<NF4>(rep[gg], F4, lane16, q_hi, q_lo, gg * Q + q, RS);
  }
  __syncthreads();
  _Float16* img = c.qimg + (int64_t)l * 2 * GQ * RS;
  for (int i = tid; i < 2 * GQ * RS; i += kThreads) img[i] = q_hi[i];      // (q_lo follows q_hi)
}

template <int NF4>
__global__ __launch_bounds__(kThreads) void ck_lists_sims_kernel(CkListsArgs c) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  __shared__ int lst[kSimsIds];
  __shared__ int wave_cnt[4];
  const ConvKnrmArgs& a = c.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lane16 = tid & 15, g4 = lane >> 4;
  const int l = blockIdx.x >> 3, blk = blockIdx.y * 8 + (blockIdx.x & 7);      // XCD x: the id blocks 8 k + x, each for all lists back to back
  if ((int64_t)blk * kSimsIds >= a.Vp) return;
  const int id0 = blk * kSimsIds;
  const int G = a.G, Q = a.Q, F = a.F, F4 = F >> 2, RS = F + 8, GQ = G * Q;
  constexpr int kPer = kSimsIds / 256;
  const uint8_t* fp = c.flags + (int64_t)l * a.Vp + id0 + tid * kPer;
  const uint64_t fw = kPer == 2 ? (uint64_t)*reinterpret_cast<const uint16_t*>(fp) : kPer == 4 ? (uint64_t)*reinterpret_cast<const uint32_t*>(fp)
                                                                                               : *reinterpret_cast<const uint64_t*>(fp);
  _Float16* q_hi = reinterpret_cast<_Float16*>(smem_raw);                 // [G * Q][RS]
  _Float16* q_lo = q_hi + GQ * RS;
  _Float16* d_hi = q_lo + GQ * RS + wave * 2 * kCkTile * RS;              // the wave's own [16][RS] planes
  _Float16* d_lo = d_hi + kCkTile * RS;
  {
    const _Float16* img = c.qimg + (int64_t)l * 2 * GQ * RS;
    for (int i = tid * 8; i < 2 * GQ * RS; i += kThreads * 8) *reinterpret_cast<h8*>(q_hi + i) = *reinterpret_cast<const h8*>(img + i);     // (RS is a multiple of 8)
  }
  int slot[kPer], mine = 0;
#pragma unroll
  for (int cc = 0; cc < kPer; ++cc) {
    const uint64_t set = __ballot(((fw >> (8 * cc)) & 0xffu) != 0);
    slot[cc] = mine + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(set >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)set, 0u));
    mine += __builtin_popcountll(set);
  }
  if (lane == 0) wave_cnt[wave] = mine;
  __syncthreads();
  const int c0n = wave_cnt[0], c1n = wave_cnt[1], c2n = wave_cnt[2], c3n = wave_cnt[3];
  const int total = c0n + c1n + c2n + c3n;
  if (total == 0) return;
  const int basei = wave == 0 ? 0 : wave == 1 ? c0n : wave == 2 ? c0n + c1n : c0n + c1n + c2n;
#pragma unroll
  for (int cc = 0; cc < kPer; ++cc)
    if ((fw >> (8 * cc)) & 0xffu) lst[basei + slot[cc]] = tid * kPer + cc;
  __syncthreads();
  float* tab = c.table + (int64_t)l * a.Vp * a.TS;
  const int n_mb = (GQ + 15) >> 4;
  const float4* parts = reinterpret_cast<const float4*>(a.tables);
  const int P = ck_parts(G);
  for (int e0 = wave * kCkTile; e0 < total; e0 += 4 * kCkTile) {      // (no barrier below: a wave and its own planes)
    const int nv = total - e0 < kCkTile ? total - e0 : kCkTile;
    // the tile's 16 tokens, four at a time (a 16-lane group each): part (1, 0) - tap 0 of the unigram convolution, bias folded in
    float4 v[4][NF4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int e = e0 + 4 * r + g4;
      const int t = id0 + lst[e < total ? e : total - 1];
      const float4* row = parts + (int64_t)(t < a.V ? t : 0) * P * F4;
#pragma unroll
      for (int i = 0; i < NF4; ++i) {
        const int x = i * 16 + lane16;
        v[r][i] = row[x < F4 ? x : F4 - 1];
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int i = 0; i < NF4; ++i)
        if (i * 16 + lane16 >= F4) v[r][i] = make_float4(0.f, 0.f, 0.f, 0.f);
      ck_store_unit<NF4>(v[r], F4, lane16, d_hi, d_lo, 4 * r + g4, RS);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // (one wave: LDS program order is the synchronisation)
    for (int mb = 0; mb < n_mb; ++mb) {
      const int n = lane & 15, kg = lane >> 4, m = mb * 16 + n;
      const bool arow = m < GQ;
      const _Float16* ah = q_hi + (arow ? m : 0) * RS + 8 * kg;
      const _Float16* al = q_lo + (arow ? m : 0) * RS + 8 * kg;
      const _Float16* bh = d_hi + n * RS + 8 * kg;
      const _Float16* bl = d_lo + n * RS + 8 * kg;
      f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};   // two chains (even / odd K steps), as phase B of the per-pair kernel
      const h8 zero = {0};
#pragma unroll
      for (int pr = 0; pr < kCkMaxF / 64; ++pr)
        if (pr * 64 < F) {
          h8 a_hi[2], a_lo[2], b_hi[2], b_lo[2];
#pragma unroll
          for (int st = 0; st < 2; ++st) {
            const int kk = pr * 64 + st * 32;
            const bool kin = kk + 8 * kg < F;
            a_hi[st] = (arow && kin) ? *reinterpret_cast<const h8*>(ah + kk) : zero;
            a_lo[st] = (arow && kin) ? *reinterpret_cast<const h8*>(al + kk) : zero;
            b_hi[st] = kin ? *reinterpret_cast<const h8*>(bh + kk) : zero;
            b_lo[st] = kin ? *reinterpret_cast<const h8*>(bl + kk) : zero;
          }
          c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi[0], b_hi[0], c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi[1], b_hi[1], c1, 0, 0, 0);
          c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo[0], b_hi[0], c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo[1], b_hi[1], c1, 0, 0, 0);
          c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi[0], b_lo[0], c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi[1], b_lo[1], c1, 0, 0, 0);
        }
      if (n < nv) {
        float* e = tab + (int64_t)(id0 + lst[e0 + n]) * a.TS;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = mb * 16 + kg * 4 + i;           // query vector of accumulator register i
          if (row < GQ) e[row] = c0[i] + c1[i];
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // (the planes are rewritten by the wave's next tile)
  }
}

template <int NF4>
__global__ __launch_bounds__(kThreads, CAPAMD_CK_WAVES) void convknrm_forward_lists_kernel(ConvKnrmArgs a, ListsArgs la, ListGeom g) {
  int l, doc;
  if (!list_doc_of(la, l, doc) || doc >= g.len[l]) return;
  convknrm_forward_body<NF4, true>(a, g.start[l] + doc, a.ltab + (int64_t)l * a.Vp * a.TS);
}

int ck_table_stride(int G, int Q) { return (G * Q + 3) & ~3; }
size_t ck_lists_per_list_bytes(int64_t Vp, int G, int Q, int F) {
  return (size_t)Vp * (1 + 4 * (size_t)ck_table_stride(G, Q)) + (((size_t)2 * G * Q * (F + 8) * 2) + 15 & ~(size_t)15);
}

}  // namespace

extern "C" size_t capamd_convknrm_lists_workspace_bytes(int n_lists, int64_t V, int Q, int maxngram, int filters) {
  if (n_lists < 1 || V < 1 || Q < 1 || Q > kCkMaxQ || capamd_convknrm_table_bytes(V, maxngram, filters) < 0) return 0;
  const int n = n_lists < kListChunk ? n_lists : kListChunk;
  return (size_t)n * ck_lists_per_list_bytes(lists_vp(V), maxngram, Q, filters) + 16;
}

extern "C" int capamd_convknrm_forward_lists(const int64_t* q_ids, const int64_t* d_ids, const int64_t* list_offsets_host, int n_lists, int Q, int L,
                                             const float* tables, int64_t V, int maxngram, int filters, int crossmatch, const float* mu,
                                             const float* sigma, int K, const float* w1, const float* b1, int H, const float* w2, const float* b2,
                                             int score_tanh, float* out, int* status, void* workspace, size_t workspace_bytes, void* stream) {
  if (n_lists == 0) return CAPAMD_OK;
  if (!q_ids || !d_ids || !list_offsets_host || !tables || !mu || !sigma || !w1 || !b1 || !out || !status || !workspace) return CAPAMD_ERR_ARG;
  if (H != 0 && (!w2 || !b2)) return CAPAMD_ERR_ARG;
  if (n_lists < 0 || Q < 1 || Q > kCkMaxQ || L < 1 || L > 4096 || V > 0x7fffffffLL || K < 1 || K > kCkMaxK || H < 0 || H > kCkMaxH) return CAPAMD_ERR_ARG;
  if (capamd_convknrm_table_bytes(V, maxngram, filters) < 0) return CAPAMD_ERR_ARG;
  if ((reinterpret_cast<uintptr_t>(tables) & 15) != 0 || (reinterpret_cast<uintptr_t>(workspace) & 15) != 0) return CAPAMD_ERR_ALIGN;
  const int G = maxngram, F = filters, views = crossmatch ? G * G : G;
  const size_t smem = (size_t)ck_layout(L, F, G, Q, views).total;
  if (smem > 160 * 1024) return CAPAMD_ERR_ARG;
  const int64_t Vp = lists_vp(V), n_pairs = list_offsets_host[n_lists];
  if (n_pairs < 0 || n_pairs > 0x7fffffffLL) return CAPAMD_ERR_ARG;
  for (int l = 0; l < n_lists; ++l)
    if (list_offsets_host[l + 1] < list_offsets_host[l]) return CAPAMD_ERR_ARG;
  const int TS = ck_table_stride(G, Q), RS = F + 8, GQ = G * Q;
  const size_t per_list = ck_lists_per_list_bytes(Vp, G, Q, F);
  if (workspace_bytes < per_list + 16) return CAPAMD_ERR_WORKSPACE;
  const size_t fit = (workspace_bytes - 16) / per_list;
  const int cap = (int)(fit < (size_t)kListChunk ? fit : (size_t)kListChunk);
  hipStream_t s = (hipStream_t)stream;
  (void)hipGetLastError();
  const IdSource ids{q_ids, d_ids, nullptr, nullptr, nullptr, nullptr};
  const size_t qimg_bytes = (((size_t)2 * GQ * RS * 2) + 15) & ~(size_t)15;
  for (int l0 = 0; l0 < n_lists; l0 += cap) {
    const int nl = n_lists - l0 < cap ? n_lists - l0 : cap;
    ListGeom g{};
    int longest = 0;
    for (int i = 0; i < nl; ++i) {
      g.start[i] = (int)list_offsets_host[l0 + i];
      g.len[i] = (int)(list_offsets_host[l0 + i + 1] - list_offsets_host[l0 + i]);
      if (g.len[i] > longest) longest = g.len[i];
    }
    if (longest == 0) continue;
    // workspace: table [cap][Vp][TS] floats | byte maps [cap][Vp] | query images [cap]
    char* ws = static_cast<char*>(workspace);
    float* table = reinterpret_cast<float*>(ws);
    uint8_t* flags = reinterpret_cast<uint8_t*>(ws + (size_t)cap * Vp * TS * 4);
    _Float16* qimg = reinterpret_cast<_Float16*>((reinterpret_cast<uintptr_t>(flags + (size_t)cap * Vp) + 15) & ~(uintptr_t)15);
    // (the shared clear / mark passes: they read these fields of ListsArgs only)
    ListsArgs la{};
    la.ids = ids; la.Q = Q; la.L = L; la.V = V; la.Vp = Vp; la.flags = flags; la.status = status; la.nl = nl; la.longest = longest;
    la.preflag = lists_preflag(n_pairs, L, n_lists); la.QP = (Q + kQT - 1) / kQT;
    hipLaunchKernelGGL(lists_clear_kernel, dim3((unsigned)((Vp + 256 * 16 - 1) / (256 * 16)), (unsigned)nl), dim3(256), 0, s, la);
    {
      ListsArgs am = la;
      am.longest = (longest + 3) / 4;       // four documents per workgroup
      hipLaunchKernelGGL(lists_mark_kernel<false>, list_doc_grid(nl, am.longest), dim3(256), 0, s, am, g);
    }
    ConvKnrmArgs a{q_ids, d_ids, (int)n_pairs, Q, L, tables, V, G, F, crossmatch ? 1 : 0, mu, sigma, K, w1, b1, H, w2, b2, score_tanh ? 1 : 0,
                   out, status, table, Vp, TS};
    CkListsArgs c{a, flags, table, qimg, nl};
    const size_t q_smem = (size_t)2 * GQ * RS * 2, s_smem = q_smem + (size_t)4 * 2 * kCkTile * RS * 2;
    const dim3 sg((unsigned)nl * 8, (unsigned)((Vp / kSimsIds + 7) / 8));
#define LAUNCH_L(NF4_)                                                                                                          \
  do {                                                                                                                          \
    hipLaunchKernelGGL(ck_lists_query_kernel<NF4_>, dim3(nl), dim3(kThreads), q_smem, s, c, g);                                 \
    hipLaunchKernelGGL(ck_lists_sims_kernel<NF4_>, sg, dim3(kThreads), s_smem, s, c);                                           \
    auto k = convknrm_forward_lists_kernel<NF4_>;                                                                               \
    if (smem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
    hipLaunchKernelGGL(k, list_doc_grid(nl, longest), dim3(kThreads), smem, s, a, la, g);                                       \
  } while (0)
    if (filters <= 64) LAUNCH_L(1);
    else LAUNCH_L(2);
#undef LAUNCH_L
    (void)qimg_bytes;
    if (hipGetLastError() != hipSuccess) return CAPAMD_ERR_LAUNCH;
  }
  return CAPAMD_OK;
}

extern "C" int64_t capamd_convknrm_table_bytes(int64_t V, int maxngram, int filters) {
  if (V < 1 || maxngram < 1 || maxngram > kCkMaxG || filters < 16 || filters > kCkMaxF || (filters & 15)) return -1;
  return V * ck_parts(maxngram) * filters * (int64_t)sizeof(float);
}

extern "C" int capamd_convknrm_pack_tables(const float* emb, int64_t V, int D, int64_t ld, const float* conv_w, const float* conv_b,
                                           int maxngram, int filters, float* tables, void* stream) {
  if (!emb || !conv_w || !conv_b || !tables || D < 1 || D > 1024 || ld < D) return CAPAMD_ERR_ARG;
  if (capamd_convknrm_table_bytes(V, maxngram, filters) < 0) return CAPAMD_ERR_ARG;
  if ((reinterpret_cast<uintptr_t>(tables) & 15) != 0) return CAPAMD_ERR_ALIGN;
  (void)hipGetLastError();
  const int64_t blocks = (V + kPackTokens - 1) / kPackTokens;
  hipLaunchKernelGGL(convknrm_pack_kernel, dim3((unsigned)blocks), dim3(256), (size_t)D * kPackTokens * sizeof(float), (hipStream_t)stream, emb,
                     V, D, ld, conv_w, conv_b, maxngram, filters, tables);
  return hipGetLastError() == hipSuccess ? CAPAMD_OK : CAPAMD_ERR_LAUNCH;
}

extern "C" int capamd_convknrm_forward(const int64_t* q_ids, const int64_t* d_ids, int B, int Q, int L, const float* tables, int64_t V,
                                       int maxngram, int filters, int crossmatch, const float* mu, const float* sigma, int K, const float* w1,
                                       const float* b1, int H, const float* w2, const float* b2, int score_tanh, float* out, int* status,
                                       void* stream) {
  if (B == 0) return CAPAMD_OK;
  if (!q_ids || !d_ids || !tables || !mu || !sigma || !w1 || !b1 || !out || !status) return CAPAMD_ERR_ARG;
  if (H != 0 && (!w2 || !b2)) return CAPAMD_ERR_ARG;
  if (B < 0 || Q < 1 || Q > kCkMaxQ || L < 1 || L > 4096 || V > 0x7fffffffLL || K < 1 || K > kCkMaxK || H < 0 || H > kCkMaxH)
    return CAPAMD_ERR_ARG;
  if (capamd_convknrm_table_bytes(V, maxngram, filters) < 0) return CAPAMD_ERR_ARG;
  if ((reinterpret_cast<uintptr_t>(tables) & 15) != 0) return CAPAMD_ERR_ALIGN;
  const size_t smem = (size_t)ck_layout(L, filters, maxngram, Q, crossmatch ? maxngram * maxngram : maxngram).total;
  if (smem > 160 * 1024) return CAPAMD_ERR_ARG;
  ConvKnrmArgs a{q_ids, d_ids, B, Q, L, tables, V, maxngram, filters, crossmatch ? 1 : 0, mu, sigma, K, w1, b1, H, w2, b2, score_tanh ? 1 : 0,
                 out, status, nullptr, 0, 0};
  hipStream_t s = (hipStream_t)stream;
  (void)hipGetLastError();
#define LAUNCH(NF4_)                                                                                                            \
  do {                                                                                                                          \
    auto k = convknrm_forward_kernel<NF4_>;                                                                                     \
    if (smem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
    hipLaunchKernelGGL(k, dim3(B), dim3(kThreads), smem, s, a);                                                                 \
  } while (0)
  if (filters <= 64) LAUNCH(1);
  else LAUNCH(2);
#undef LAUNCH
  return hipGetLastError() == hipSuccess ? CAPAMD_OK : CAPAMD_ERR_LAUNCH;
}
