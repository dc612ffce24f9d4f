// Differentiable cosine-similarity + RBF kernel pooling over DENSE n-gram representations, forward and backward, for gfx950:
// the part of ConvKNRM's TRAINING step between its (trainable) n-gram convolutions and its combine layer
// (reference capreolus/reranker/ConvKNRM.py:53-76 on StackedSimilarityMatrix, common.py:195-221, and RbfKernelBank, common.py:224-250).
//
//   sim[v][q][j] = <a_q, b_j> / ((|a_q| + 1e-9) (|b_j| + 1e-9)),  0 where query position q or document position j is a pad   (common.py:206-219)
//   S[v][k][q]   = sum over ALL L positions of exp(-(sim - mu_k)^2 / (2 sigma_k^2))                                            (ConvKNRM.py:63-71)
//   feat[k V + v] = sum_q  (sum_j sim != 0) ? log(S + 1e-6) : 0                                                                (ConvKNRM.py:72-75)
// with a = an n-gram view of the query [Q, F], b = an n-gram view of the document [L, F]; views v = every (a view, b view) pair
// (crossmatch) or the matching ones.  Scoring never runs this (convknrm.hip folds the convolutions into per-token tables); training
// needs the gradient through the cosine into both operands, which this file supplies - row N3 of SURVEY.md section 8f without ATen.
//
// One workgroup per (pair, document view): 16 groups of 16 lanes walk the document positions; a lane holds float4 chunks lane16 + 16 i
// of its position's vector and, after the 16-lane DPP all-reduce of the dot products, owns RBF kernel k = lane16.
// Backward recomputes the similarities (cheaper than storing [B, V, Q, L]) and needs from the forward only S and the row sums:
//   coef[t][k] = g[k V + v] [row sum != 0] / (S + 1e-6)
//   w          = coef K_k(sim) (sim - mu_k) / sigma_k^2        d mu_k += w      d sigma_k += w (sim - mu_k) / sigma_k      d sim = - sum_k w
//   d a_q      = sum_j d sim ( b_j / (na nb) - sim a_q / (|a_q| na) )            (na = |a_q| + 1e-9, nb = |b_j| + 1e-9; no gradient
//   d b_j      = sum_q d sim ( a_q / (na nb) - sim b_j / (|b_j| nb) )             through a masked entry, whose similarity is the constant 0)
#include "capreolus_amd.h"
#include "interaction.h"

using namespace capamd;

namespace {

constexpr int kPoolMaxT = 24;      // query vectors a block holds: (a views in its pair set) x Q
constexpr int kPoolMaxK = 16;
constexpr int kPoolMaxNC = 4;      // float4 chunks per lane: F <= 256
constexpr float kPoolLog2e = 1.4426950408889634f;

struct PoolArgs {
  const float* qrep;      // [B, GQ, Q, F]
  const float* drep;      // [B, GD, L, F]
  const int64_t* q_ids;   // [B, Q]   (pad = 0)
  const int64_t* d_ids;   // [B, L]
  int B, GQ, GD, Q, L, F, cross;
  const float* mu;
  const float* sigma;
  int K;
  float* feat;            // [B, K V]
  float* ksum;            // [B, GD, T, K]   forward -> backward
  float* rowsum;          // [B, GD, T]
  const float* gfeat;     // [B, K V]
  float* dq_part;         // [B, GD, C, T, F]   per document view and chunk: summed over them by the caller
  float* dd;              // [B, GD, L, F]
  float* dmu_part;        // [B GD C, K]
  float* dsigma_part;     // [B GD C, K]
  int C;                  // chunks a document's real positions are split into: one workgroup per (pair, document view, chunk)
  float* part;            // forward: [B, GD, C, T, K + 1] kernel sums and row sum of every chunk -> pool_finish_kernel
};

// chunk slots per document: one per 160 positions of L, at most 8 - a (pair, view) per workgroup leaves a batch of 64 documents on 192 of
// 256 CUs for as long as its LONGEST document takes.  How many of the slots a document uses depends on its real positions (below).
// (One slot per 64 positions, up to 16, so that long documents split finer in the backward: 141 -> 155 us, the partial results' sums grow.)
__host__ __device__ inline int pool_chunks(int L) {
  const int c = (L + 159) / 160;
  return c < 1 ? 1 : (c > 8 ? 8 : c);
}
// the a view of query vector t of block (b, gb), and the view index v of its pair
__device__ __forceinline__ int pool_ga(const PoolArgs& a, int t, int gb) { return a.cross ? t / a.Q : gb; }
__device__ __forceinline__ int pool_v(const PoolArgs& a, int t, int gb) { return a.cross ? (t / a.Q) * a.GD + gb : gb; }

// TT: compile-time bound on the query vectors of a block (register arrays are sized by it: 4, 12 or kPoolMaxT)
template <int NC, int TT, bool BWD>
__global__ __launch_bounds__(256, 1) void kernel_pool_kernel(PoolArgs a) {
  extern __shared__ __attribute__((aligned(16))) char pool_lds[];
  const int T = (a.cross ? a.GQ : 1) * a.Q;
  float* A = reinterpret_cast<float*>(pool_lds);            // [T][F]
  float* an = A + kPoolMaxT * a.F;                          // [T] |a| + 1e-9
  float* araw = an + kPoolMaxT;                             // [T] |a|
  int* qpad = reinterpret_cast<int*>(araw + kPoolMaxT);     // [T]
  float* coef = reinterpret_cast<float*>(qpad + kPoolMaxT); // [T][16]   (BWD)
  float* red = coef + kPoolMaxT * 16;                       // [16 groups][T][16] forward sums / [16][F] backward slices
  int* jl = reinterpret_cast<int*>(red + 16 * kPoolMaxT * 16 + 16 * kPoolMaxT + 16 * a.F + 64);   // [L] the real positions, then the pads (descending from the end)
  __shared__ int n_real_s;
  const int tid = threadIdx.x, lane16 = tid & 15, g = tid >> 4;
  const int b = blockIdx.x / a.GD, gb = blockIdx.x % a.GD, chunk = blockIdx.y;
  const int V = a.cross ? a.GQ * a.GD : a.GD;
  const int F4 = a.F >> 2;

  // the document's real positions in order (front of jl) and its pads (back of jl): pads never reach the similarity loop - a masked entry
  // is the constant 0, so what a pad adds to a kernel sum (and to d mu / d sigma) is a closed form times the number of pads
  // (all four waves, 1024 positions a round, every lane's four ids requested together: one wave walking the row 64 positions at a time
  //  was thirteen dependent round trips at L = 800 - most of the fixed cost of a workgroup, and every chunk's workgroup pays it)
  {
    __shared__ int trip_real[16], trip_pad[16], tot_real, tot_pad;
    const int64_t* di = a.d_ids + (int64_t)b * a.L;
    const int lane = tid & 63, wave = tid >> 6;
    const uint64_t below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    if (tid == 0) tot_real = tot_pad = 0;
    for (int base = 0; base < a.L; base += 1024) {
      int64_t v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = base + (wave * 4 + u) * 64 + lane;
        v[u] = di[j < a.L ? j : a.L - 1];
      }
      uint64_t mr[4], mp[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = base + (wave * 4 + u) * 64 + lane;
        const bool in = j < a.L;
        mr[u] = __ballot(in && v[u] != 0);
        mp[u] = __ballot(in && v[u] == 0);
        if (lane == 0) {
          trip_real[wave * 4 + u] = __builtin_popcountll(mr[u]);
          trip_pad[wave * 4 + u] = __builtin_popcountll(mp[u]);
        }
      }
      __syncthreads();
      int nr = tot_real, np = tot_pad, all_r = 0, all_p = 0;
      for (int k = 0; k < 16; ++k) {
        if (k < wave * 4) {
          nr += trip_real[k];
          np += trip_pad[k];
        }
        all_r += trip_real[k];
        all_p += trip_pad[k];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = base + (wave * 4 + u) * 64 + lane;
        if ((mr[u] >> lane) & 1) jl[nr + __builtin_popcountll(mr[u] & below)] = j;
        else if ((mp[u] >> lane) & 1) jl[a.L - 1 - (np + __builtin_popcountll(mp[u] & below))] = j;
        nr += __builtin_popcountll(mr[u]);
        np += __builtin_popcountll(mp[u]);
      }
      __syncthreads();
      if (tid == 0) {
        tot_real += all_r;
        tot_pad += all_p;
      }
    }
    __syncthreads();
    if (tid == 0) n_real_s = tot_real;
  }
  // the query vectors of this block's pairs, their norms, the pad flags
  for (int i = tid; i < T * F4; i += 256) {
    const int t = i / F4, c = i - t * F4;
    const float* src = a.qrep + (((int64_t)b * a.GQ + pool_ga(a, t, gb)) * a.Q + t % a.Q) * a.F;
    reinterpret_cast<float4*>(A)[t * F4 + c] = reinterpret_cast<const float4*>(src)[c];
  }
  __syncthreads();
  // The document's real positions go to its first `act` chunks; a chunk beyond them (most documents end in padding: 303 real
  // positions of 800 on the benchmark's lists = 2 chunks of 5) writes zeros for its partial results and is done - a workgroup costs its
  // prologue and epilogue whatever it holds, so five chunks of 60 positions were slower than two of 150.
  // (The backward's loop is a long dependent chain per position and wants SHORT chunks - 64 positions: 153 us at 160, 141 at 64; the
  //  forward 160: 45 us against 57.)
  constexpr int kPerChunk = BWD ? 64 : 160;
  const int want = (n_real_s + kPerChunk - 1) / kPerChunk;
  const int act = want < 1 ? 1 : (want < a.C ? want : a.C);
  if (chunk >= act) {
    const int64_t blk0 = ((int64_t)b * a.GD + gb) * a.C + chunk;
    if (!BWD) {
      for (int i = tid; i < T * (a.K + 1); i += 256) a.part[blk0 * T * (a.K + 1) + i] = 0.f;
    } else {
      for (int i = tid; i < T * a.F; i += 256) a.dq_part[blk0 * T * a.F + i] = 0.f;
      if (tid < a.K) {
        a.dmu_part[blk0 * a.K + tid] = 0.f;
        a.dsigma_part[blk0 * a.K + tid] = 0.f;
      }
    }
    return;
  }
  if (tid < T) {
    float s = 0.f;
    for (int c = 0; c < a.F; ++c) s = __builtin_fmaf(A[tid * a.F + c], A[tid * a.F + c], s);
    const float n = sqrtf(s);
    araw[tid] = n;
    an[tid] = n + 1e-9f;
    qpad[tid] = a.q_ids[(int64_t)b * a.Q + tid % a.Q] == 0;
  }
  if (BWD)
    for (int i = tid; i < T * 16; i += 256) {
      const int t = i >> 4, k = i & 15;
      float c = 0.f;
      if (k < a.K) {
        const int64_t base = ((int64_t)b * a.GD + gb) * T + t;
        const float S = a.ksum[base * a.K + k];
        c = a.rowsum[base] != 0.f ? a.gfeat[(int64_t)b * a.K * V + k * V + pool_v(a, t, gb)] / (S + 1e-6f) : 0.f;
      }
      coef[i] = c;
    }
  __syncthreads();

  const int kk = lane16 < a.K ? lane16 : a.K - 1;
  const float mu_k = a.mu[kk], sg_k = a.sigma[kk];
  const float c_k = lane16 < a.K ? (-0.5f * kPoolLog2e) / (sg_k * sg_k) : 0.f;
  float acc[TT], rs[TT];          // forward: kernel sums of kernel lane16, row sums
  float da[BWD ? TT : 1][NC * 4];        // backward: this lane's slice of d a_t (the b_j / (na nb) part)
  float ca[BWD ? TT : 1];                //           sum_j d sim sim (the norm part's coefficient)
  float dmu = 0.f, dsg = 0.f;
#pragma unroll
  for (int t = 0; t < TT; ++t) {
    acc[t] = rs[t] = 0.f;
    if (BWD) {
      ca[t] = 0.f;
#pragma unroll
      for (int e = 0; e < NC * 4; ++e) da[t][e] = 0.f;
    }
  }
  const float* drow = a.drep + ((int64_t)b * a.GD + gb) * a.L * a.F;
  const int n_real = n_real_s, n_pad = a.L - n_real;
  if (g == 0 && n_pad > 0 && chunk == 0) {          // the pads' share: similarity 0 at every one of them
    const float adj = 0.f - mu_k, k0 = __builtin_amdgcn_exp2f(adj * adj * c_k);
#pragma unroll
    for (int t = 0; t < TT; ++t)
      if (t < T) {
        if (!BWD) acc[t] += (float)n_pad * k0;
        else {
          const float w = lane16 < a.K ? coef[t * 16 + lane16] * k0 * adj / (sg_k * sg_k) : 0.f;
          dmu += (float)n_pad * w;
          dsg += (float)n_pad * (w * adj / sg_k);
        }
      }
  }
  if (BWD) {                          // no gradient into a pad position's vector
    float* dz = a.dd + ((int64_t)b * a.GD + gb) * a.L * a.F;
    for (int i = tid + 256 * chunk; i < n_pad * F4; i += 256 * act) {
      const int p = i / F4, c = i - p * F4;
      reinterpret_cast<float4*>(dz + (int64_t)jl[a.L - 1 - p] * a.F)[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  const int ji_lo = (int)((int64_t)n_real * chunk / act), ji_hi = (int)((int64_t)n_real * (chunk + 1) / act);
#ifndef CAPAMD_KPOOL_ABL
#define CAPAMD_KPOOL_ABL 0       // profiling builds: 1 = no position loop, 2 = no d a_t epilogue, 4 = one query vector in the loop's update half, 8 = no exp / reduction in its first half
#endif
  for (int ji = ji_lo + g; ji < ((CAPAMD_KPOOL_ABL & 1) ? 0 : ji_hi); ji += 16) {
    const int j = jl[ji];
    float4 x[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = lane16 + 16 * i;
      x[i] = c < F4 ? reinterpret_cast<const float4*>(drow + (int64_t)j * a.F)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float nb2 = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) nb2 += x[i].x * x[i].x + x[i].y * x[i].y + x[i].z * x[i].z + x[i].w * x[i].w;
    nb2 = group_allreduce(nb2);
    const float braw = sqrtf(nb2), nb = braw + 1e-9f;
    constexpr bool dpad = false;        // (real positions only)
    float s[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      s[t] = 0.f;
      if (t < T) {
        float p = 0.f;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
          const int c = lane16 + 16 * i;
          if (c < F4) {
            const float4 q = reinterpret_cast<const float4*>(A)[t * F4 + c];
            p += x[i].x * q.x + x[i].y * q.y + x[i].z * q.z + x[i].w * q.w;
          }
        }
        p = group_allreduce(p);
        s[t] = (dpad || qpad[t]) ? 0.f : p / (an[t] * nb);
      }
    }
    if (!BWD) {
#pragma unroll
      for (int t = 0; t < TT; ++t)
        if (t < T) {
          rs[t] += s[t];
          const float adj = s[t] - mu_k;
          acc[t] += __builtin_amdgcn_exp2f(adj * adj * c_k);
        }
    } else {
      float dsim[TT], cb = 0.f;     // cb = sum_t d sim sim: coefficient of b_j's own direction
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        dsim[t] = 0.f;
        if (t < T) {
          const float adj = s[t] - mu_k;
          const float w = (CAPAMD_KPOOL_ABL & 8) ? adj : lane16 < a.K ? coef[t * 16 + lane16] * __builtin_amdgcn_exp2f(adj * adj * c_k) * adj / (sg_k * sg_k) : 0.f;
          dmu += w;
          dsg += w * adj / sg_k;
          const float ds = (CAPAMD_KPOOL_ABL & 8) ? w : group_allreduce(-w);
          dsim[t] = (dpad || qpad[t]) ? 0.f : ds;
          cb += dsim[t] * s[t];
          ca[t] += dsim[t] * s[t];
        }
      }
      // d b_j slice; d a_t slices accumulate b_j / (na nb)
      const float binv = braw > 0.f ? cb / (braw * nb) : 0.f;
      float4 o[NC];
#pragma unroll
      for (int i = 0; i < NC; ++i) o[i] = make_float4(-binv * x[i].x, -binv * x[i].y, -binv * x[i].z, -binv * x[i].w);
#pragma unroll
      for (int t = 0; t < ((CAPAMD_KPOOL_ABL & 4) ? 1 : TT); ++t)
        if (t < T) {
          const float w = dsim[t] / (an[t] * nb);
#pragma unroll
          for (int i = 0; i < NC; ++i) {
            const int c = lane16 + 16 * i;
            if (c < F4) {
              const float4 q = reinterpret_cast<const float4*>(A)[t * F4 + c];
              o[i].x = __builtin_fmaf(w, q.x, o[i].x); o[i].y = __builtin_fmaf(w, q.y, o[i].y);
              o[i].z = __builtin_fmaf(w, q.z, o[i].z); o[i].w = __builtin_fmaf(w, q.w, o[i].w);
              da[t][i * 4 + 0] = __builtin_fmaf(w, x[i].x, da[t][i * 4 + 0]); da[t][i * 4 + 1] = __builtin_fmaf(w, x[i].y, da[t][i * 4 + 1]);
              da[t][i * 4 + 2] = __builtin_fmaf(w, x[i].z, da[t][i * 4 + 2]); da[t][i * 4 + 3] = __builtin_fmaf(w, x[i].w, da[t][i * 4 + 3]);
            }
          }
        }
      float* dst = a.dd + (((int64_t)b * a.GD + gb) * a.L + j) * a.F;
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        const int c = lane16 + 16 * i;
        if (c < F4) reinterpret_cast<float4*>(dst)[c] = o[i];
      }
    }
  }

  const int64_t blk = ((int64_t)b * a.GD + gb) * a.C + chunk;
  if (!BWD) {
    // 16 groups -> one, fixed order: this chunk's kernel sums S[t][k] and row sums (pool_finish_kernel adds the chunks up)
#pragma unroll
    for (int t = 0; t < TT; ++t)
      if (t < T) {
        red[(g * kPoolMaxT + t) * 16 + lane16] = acc[t];
        if (lane16 == 0) red[16 * kPoolMaxT * 16 + g * kPoolMaxT + t] = rs[t];
      }
    __syncthreads();
    for (int i = tid; i < T * 16; i += 256) {
      const int t = i >> 4, k = i & 15;
      float v = 0.f;
      for (int gg = 0; gg < 16; ++gg) v += red[(gg * kPoolMaxT + t) * 16 + k];
      if (k < a.K) a.part[(blk * T + t) * (a.K + 1) + k] = v;
    }
    if (tid < T) {
      float v = 0.f;
      for (int gg = 0; gg < 16; ++gg) v += red[16 * kPoolMaxT * 16 + gg * kPoolMaxT + tid];
      a.part[(blk * T + tid) * (a.K + 1) + a.K] = v;
    }
  } else {
    // d mu / d sigma partials of this block: kernel lanes over the 16 groups
    red[g * 32 + lane16] = dmu;
    red[g * 32 + 16 + lane16] = dsg;
    __syncthreads();
    if (tid < 32) {
      float v = 0.f;
      for (int gg = 0; gg < 16; ++gg) v += red[gg * 32 + tid];
      // every lane of a group carried the same w for ITS kernel; a group's 16 lanes are the 16 kernels
      if ((tid & 15) < a.K) (tid < 16 ? a.dmu_part : a.dsigma_part)[blk * a.K + (tid & 15)] = v;
    }
    __syncthreads();
    // d a_t: group slices through LDS, one query vector at a time; the norm part from the summed coefficient
    // (unrolled over the compile-time bound, barriers outside the guards: a run-time index into the register arrays would move
    // them to scratch)
#pragma unroll
    for (int t = 0; t < ((CAPAMD_KPOOL_ABL & 2) ? 0 : TT); ++t) {
      if (t < T) {
#pragma unroll
        for (int i = 0; i < NC; ++i) {
          const int c = lane16 + 16 * i;
          if (c < F4) reinterpret_cast<float4*>(red)[g * F4 + c] = make_float4(da[t][i * 4], da[t][i * 4 + 1], da[t][i * 4 + 2], da[t][i * 4 + 3]);
        }
        if (lane16 == 0) red[16 * a.F + g] = ca[t];
      }
      __syncthreads();
      if (t < T) {
        float cat = 0.f;
        for (int gg = 0; gg < 16; ++gg) cat += red[16 * a.F + gg];
        const float ainv = araw[t] > 0.f ? cat / (araw[t] * an[t]) : 0.f;
        for (int e = tid; e < a.F; e += 256) {
          float v = 0.f;
          for (int gg = 0; gg < 16; ++gg) v += red[gg * a.F + e];
          a.dq_part[(blk * T + t) * a.F + e] = v - ainv * A[t * a.F + e];
        }
      }
      __syncthreads();
    }
  }
}

// forward, second half: the chunks of a (pair, document view) summed in their order -> ksum, rowsum, and the view's features
//   feat[k V + v] = sum over the Q query positions of the pair of (row sum != 0 ? log(S + 1e-6) : 0)
__global__ __launch_bounds__(256) void pool_finish_kernel(PoolArgs a) {
  __shared__ float S[kPoolMaxT * 16];
  __shared__ float R[kPoolMaxT];
  const int T = (a.cross ? a.GQ : 1) * a.Q, V = a.cross ? a.GQ * a.GD : a.GD;
  const int tid = threadIdx.x, b = blockIdx.x / a.GD, gb = blockIdx.x % a.GD;
  const int64_t blk = (int64_t)b * a.GD + gb;
  for (int i = tid; i < T * (a.K + 1); i += 256) {
    const int t = i / (a.K + 1), k = i - t * (a.K + 1);
    float v = 0.f;
    for (int c = 0; c < a.C; ++c) v += a.part[((blk * a.C + c) * T + t) * (a.K + 1) + k];
    if (k < a.K) {
      S[t * 16 + k] = v;
      a.ksum[(blk * T + t) * a.K + k] = v;
    } else {
      R[t] = v;
      a.rowsum[blk * T + t] = v;
    }
  }
  __syncthreads();
  const int npair = a.cross ? a.GQ : 1;
  for (int i = tid; i < npair * a.K; i += 256) {
    const int pr = i / a.K, k = i - pr * a.K;
    float f = 0.f;
    for (int q = 0; q < a.Q; ++q) {
      const int t = pr * a.Q + q;
      f += R[t] != 0.f ? logf(S[t * 16 + k] + 1e-6f) : 0.f;
    }
    a.feat[(int64_t)b * a.K * V + k * V + pool_v(a, pr * a.Q, gb)] = f;
  }
}

int pool_check(const PoolArgs& a) {
  if (!a.qrep || !a.drep || !a.q_ids || !a.d_ids || !a.mu || !a.sigma || !a.ksum || !a.rowsum) return CAPAMD_ERR_ARG;
  if (a.B < 0 || a.GQ < 1 || a.GD < 1 || a.Q < 1 || a.L < 1 || a.F < 4 || (a.F & 3) || a.F > 64 * kPoolMaxNC || a.K < 1 || a.K > kPoolMaxK)
    return CAPAMD_ERR_ARG;
  if (!a.cross && a.GQ != a.GD) return CAPAMD_ERR_ARG;
  if ((a.cross ? a.GQ : 1) * a.Q > kPoolMaxT) return CAPAMD_ERR_ARG;
  return CAPAMD_OK;
}

template <bool BWD>
int pool_launch(const PoolArgs& a, void* stream) {
  if (a.B == 0) return CAPAMD_OK;
  const size_t lds = ((size_t)kPoolMaxT * a.F + 3 * kPoolMaxT + kPoolMaxT * 16 + 16 * kPoolMaxT * 16 + 16 * kPoolMaxT + 16 * (size_t)a.F + 64 + (size_t)a.L) * 4;
  if (lds > 160 * 1024) return CAPAMD_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  (void)hipGetLastError();
  const int nc = (a.F + 63) / 64;
#define GO2(NC_, TT_)                                                                                                        \
  do {                                                                                                                       \
    auto k = kernel_pool_kernel<NC_, TT_, BWD>;                                                                              \
    if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
      return CAPAMD_ERR_LAUNCH;                                                                                              \
    hipLaunchKernelGGL(k, dim3(a.B * a.GD, a.C), dim3(256), lds, s, a);                                                      \
  } while (0)
#define GO(NC_)                                 \
  do {                                          \
    if (T <= 4) GO2(NC_, 4);                    \
    else if (T <= 12) GO2(NC_, 12);             \
    else GO2(NC_, kPoolMaxT);                   \
  } while (0)
  const int T = (a.cross ? a.GQ : 1) * a.Q;
  switch (nc) {
    case 1: GO(1); break;
    case 2: GO(2); break;
    default: GO(4); break;
  }
#undef GO
#undef GO2
  if (!BWD) hipLaunchKernelGGL(pool_finish_kernel, dim3(a.B * a.GD), dim3(256), 0, s, a);
  return hipGetLastError() == hipSuccess ? CAPAMD_OK : CAPAMD_ERR_LAUNCH;
}

}  // namespace

extern "C" int capamd_kernel_pool_chunks(int L) { return pool_chunks(L); }

extern "C" int capamd_kernel_pool_forward(const float* qrep, const float* drep, const int64_t* q_ids, const int64_t* d_ids, int B, int GQ, int GD,
                                          int Q, int L, int F, int crossmatch, const float* mu, const float* sigma, int K, float* feat,
                                          float* ksum, float* rowsum, float* chunk_sums, void* stream) {
  PoolArgs a{qrep, drep, q_ids, d_ids, B, GQ, GD, Q, L, F, crossmatch ? 1 : 0, mu, sigma, K, feat, ksum, rowsum, nullptr, nullptr, nullptr, nullptr, nullptr,
             pool_chunks(L), chunk_sums};
  if (!feat || !chunk_sums) return CAPAMD_ERR_ARG;
  const int rc = pool_check(a);
  return rc != CAPAMD_OK ? rc : pool_launch<false>(a, stream);
}

extern "C" int capamd_kernel_pool_backward(const float* qrep, const float* drep, const int64_t* q_ids, const int64_t* d_ids, int B, int GQ, int GD,
                                           int Q, int L, int F, int crossmatch, const float* mu, const float* sigma, int K, const float* gfeat,
                                           const float* ksum, const float* rowsum, float* dq_part, float* dd, float* dmu_part,
                                           float* dsigma_part, void* stream) {
  PoolArgs a{qrep, drep, q_ids, d_ids, B, GQ, GD, Q, L, F, crossmatch ? 1 : 0, mu, sigma, K, nullptr, const_cast<float*>(ksum), const_cast<float*>(rowsum),
             gfeat, dq_part, dd, dmu_part, dsigma_part, pool_chunks(L), nullptr};
  if (!gfeat || !dq_part || !dd || !dmu_part || !dsigma_part) return CAPAMD_ERR_ARG;
  const int rc = pool_check(a);
  return rc != CAPAMD_OK ? rc : pool_launch<true>(a, stream);
}
