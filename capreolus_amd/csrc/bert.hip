// BERT-MaxP passage scoring for gfx950: PTBERTMaxP_Class.predict_step (reference
// capreolus/reranker/ptBERTMaxP.py:67-96) with the transformers BertForSequenceClassification it
// calls at :82 re-built as hand-written kernels: embedding sum + LayerNorm, bf16 MFMA GEMMs with
// fused bias / GELU / residual epilogues (bert_gemm.cuh), fused exact-softmax attention
// (bert_attn.cuh), LayerNorm, pooler + classifier, passage pooling.
//
// Precision: the activation stream (incl. the residual path) is bf16 end to end; every accumulation, the
// pre-LayerNorm sum (one rounding), LayerNorm statistics, softmax and the pooler/classifier are fp32.
// (A fp32 residual stream was measured to give the same 6e-3 logit error: the error is set by the bf16
// GEMM operands, not by the residual precision.)
#include "bert_attn.cuh"
#include "bert_gemm.cuh"
#include "capreolus_amd.h"
#include <stdlib.h>
#include <utility>
#include <vector>

using namespace capamd;

namespace {

constexpr float kLnEps = 1e-12f;  // BertConfig.layer_norm_eps

__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- fp32 -> bf16 weight conversion (row-major copy into the blob) ----------------------------
template <typename T>
__global__ void cvt_bf16_kernel(const float* __restrict__ src, T* __restrict__ dst, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = (T)src[i];
}
__global__ void copy_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

// One wave per token: v = LayerNorm(x) over H (H % 8 == 0, H <= 1024; 16-byte accesses on the 16-bit stream).
// MODE 0: x = word[id] + pos[s] + type[seg]  (fp32 embedding tables);  MODE 1: x = pre[token] (bf16 pre-LN sum).
// Output: bf16 (the activation stream is bf16 end to end; statistics and the affine are fp32).
template <int MODE, typename T>
__global__ __launch_bounds__(256) void ln_kernel(const T* __restrict__ pre, const int64_t* __restrict__ ids,
                                                 const int64_t* __restrict__ seg, const float* __restrict__ word,
                                                 const float* __restrict__ pos, const float* __restrict__ type, int vocab,
                                                 int type_vocab, int S, const float* __restrict__ gamma,
                                                 const float* __restrict__ beta, int64_t M, int H,
                                                 T* xb, int* status) {
  // one wave per token row, 8 elements (16 bytes of the 16-bit stream) per lane and step; H <= 1024
  using bf16x8 = typename Half<T>::x8;
  const int lane = threadIdx.x & 63;
  const int64_t tok = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tok >= M) return;
  const int nchunk = H >> 3;
  float v[2][8];
  const float* r0 = nullptr;
  const float* r1 = nullptr;
  const float* r2 = nullptr;
  if (MODE == 0) {
    int64_t id = ids[tok], sg = seg[tok];
    if (id < 0 || id >= vocab || sg < 0 || sg >= type_vocab) {
      if (lane == 0) atomicOr(status, 1);
      id = 0;
      sg = 0;
    }
    r0 = word + id * H;
    r1 = pos + (tok % S) * H;
    r2 = type + sg * H;
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
      if (MODE == 0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const float4 x = reinterpret_cast<const float4*>(r0)[2 * c + h], y = reinterpret_cast<const float4*>(r1)[2 * c + h],
                       z = reinterpret_cast<const float4*>(r2)[2 * c + h];
          v[i][4 * h + 0] = x.x + (y.x + z.x); v[i][4 * h + 1] = x.y + (y.y + z.y);
          v[i][4 * h + 2] = x.z + (y.z + z.z); v[i][4 * h + 3] = x.w + (y.w + z.w);
        }
      } else {
        const bf16x8 b = reinterpret_cast<const bf16x8*>(pre + tok * H)[c];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[i][e] = (float)b[e];
        if (MODE == 2) {  // residual sum in fp32: projection output + the stream row it will replace
          const bf16x8 r = reinterpret_cast<const bf16x8*>(xb + tok * H)[c];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[i][e] += (float)r[e];
        }
      }
#pragma unroll
      for (int e = 0; e < 8; e += 4) s += (v[i][e] + v[i][e + 1]) + (v[i][e + 2] + v[i][e + 3]);
    }
  }
  const float mean = wave_sum64(s) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
    if (lane + 64 * i < nchunk) {
#pragma unroll
      for (int e = 0; e < 8; e += 4) {
        const float a = v[i][e] - mean, b = v[i][e + 1] - mean, c = v[i][e + 2] - mean, d = v[i][e + 3] - mean;
        q += (a * a + b * b) + (c * c + d * d);
      }
    }
  const float rstd = rsqrtf(wave_sum64(q) / (float)H + kLnEps);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
      bf16x8 ob;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float4 g = reinterpret_cast<const float4*>(gamma)[2 * c + h], b = reinterpret_cast<const float4*>(beta)[2 * c + h];
        ob[4 * h + 0] = (T)((v[i][4 * h + 0] - mean) * rstd * g.x + b.x);
        ob[4 * h + 1] = (T)((v[i][4 * h + 1] - mean) * rstd * g.y + b.y);
        ob[4 * h + 2] = (T)((v[i][4 * h + 2] - mean) * rstd * g.z + b.z);
        ob[4 * h + 3] = (T)((v[i][4 * h + 3] - mean) * rstd * g.w + b.w);
      }
      reinterpret_cast<bf16x8*>(xb + tok * H)[c] = ob;
    }
  }
}

// pooler tanh(Wp h_CLS + bp) and classifier logit 1 (ptBERTMaxP.py:82 takes [:, 1]).
// grid (ceil(n_psg / kHeadPsg), H / 64): a block owns 64 pooler rows and kHeadPsg passages, so every pooler row is
// fetched once per kHeadPsg passages; its partial sum over those 64 rows of cls_w[1][j] * tanh(pooler_j) goes to
// part[psg][slice]; head_reduce_kernel adds the slices in fixed order (deterministic, no atomics).
constexpr int kHeadPsg = 8;
template <typename T>
__global__ __launch_bounds__(256) void head_kernel(const T* __restrict__ xf, int64_t n_psg, int S, int H,
                                                   const float* __restrict__ pw, const float* __restrict__ pb,
                                                   const float* __restrict__ cw, float* __restrict__ part) {
  __shared__ float cls[kHeadPsg][1024];
  __shared__ float wsum[4][kHeadPsg];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t p0 = (int64_t)blockIdx.x * kHeadPsg;
  const int nslice = gridDim.y, j0 = blockIdx.y * 64;
#pragma unroll
  for (int q = 0; q < kHeadPsg; ++q) {
    const int64_t psg = p0 + q < n_psg ? p0 + q : n_psg - 1;
    const T* h = xf + psg * S * H;  // token 0 ([CLS]) of the passage
    for (int i = tid; i < H; i += 256) cls[q][i] = (float)h[i];
  }
  __syncthreads();
  float acc[kHeadPsg];
#pragma unroll
  for (int q = 0; q < kHeadPsg; ++q) acc[q] = 0.f;
  const int nc = (H + 63) >> 6;  // <= 16
  for (int jj = wave; jj < 64; jj += 4) {
    const int j = j0 + jj;
    const float* w = pw + (int64_t)j * H;
    float wr[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) wr[c] = (c < nc && lane + 64 * c < H) ? w[lane + 64 * c] : 0.f;
    const float bj = pb[j], cj = cw[H + j];
#pragma unroll
    for (int q = 0; q < kHeadPsg; ++q) {
      float p = 0.f;
#pragma unroll
      for (int c = 0; c < 16; ++c)
        if (c < nc) p = __builtin_fmaf(wr[c], cls[q][(lane + 64 * c) & 1023], p);
      p = wave_sum64(p);
      acc[q] = __builtin_fmaf(cj, tanhf(p + bj), acc[q]);
    }
  }
  if (lane == 0)
#pragma unroll
    for (int q = 0; q < kHeadPsg; ++q) wsum[wave][q] = acc[q];
  __syncthreads();
  if (tid < kHeadPsg && p0 + tid < n_psg)
    part[(p0 + tid) * nslice + blockIdx.y] = (wsum[0][tid] + wsum[1][tid]) + (wsum[2][tid] + wsum[3][tid]);
}
__global__ void head_reduce_kernel(const float* __restrict__ part, int64_t n_psg, int nslice, const float* __restrict__ cb,
                                   float* __restrict__ logits) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_psg) return;
  float s = 0.f;
  for (int k = 0; k < nslice; ++k) s += part[p * nslice + k];
  logits[p] = s + cb[1];
}

// passage pooling (ptBERTMaxP.py:75-94); one wave per document.  agg: 0 max, 1 first, 2 sum, 3 avg
__global__ __launch_bounds__(64) void pool_kernel(const float* __restrict__ logits, const int64_t* __restrict__ mask,
                                                  const int64_t* __restrict__ seg, int P, int S, int agg, float* __restrict__ out,
                                                  int* __restrict__ total_cnt) {
  const int b = blockIdx.x, lane = threadIdx.x;
  float best = -INFINITY, sum = 0.f;
  int cnt = 0;
  for (int p = 0; p < P; ++p) {
    const float s = logits[(int64_t)b * P + p];
    best = fmaxf(best, s);
    if (agg >= 2) {
      const int64_t* m = mask + ((int64_t)b * P + p) * S;
      const int64_t* g = seg + ((int64_t)b * P + p) * S;
      float pos = 0.f;
      for (int i = lane; i < S; i += 64) pos += (float)(m[i] * g[i]);
      pos = wave_sum64(pos);
      if (pos > 5.f) {  // passage_mask = (sum(mask*seg) > 5)
        sum += s;
        cnt += 1;
      }
    }
  }
  if (lane == 0) {
    out[b] = agg == 0 ? best : agg == 1 ? logits[(int64_t)b * P] : sum;
    if (agg == 3) atomicAdd(total_cnt, cnt);
  }
}
__global__ void avg_div_kernel(float* out, int B, const int* total_cnt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) out[i] = out[i] / (float)(*total_cnt);  // batch-wide denominator, as the reference (:92)
}

struct Dims {
  int H, layers, heads, F, vocab, max_pos, type_vocab;
};

bool dims_ok(const capamd_bert_model* m) {
  return m && m->hidden >= 64 && m->hidden <= 1024 && m->hidden % 64 == 0 && m->heads * 64 == m->hidden && m->layers >= 1 &&
         m->ffn >= 64 && m->ffn % 64 == 0 && m->vocab >= 1 && m->max_pos >= 1 && m->type_vocab >= 1 &&
         (m->compute_dtype == 0 || m->compute_dtype == 1);
}

int64_t layer_blob_elems(int H, int F) { return (int64_t)3 * H * H + (int64_t)H * H + (int64_t)2 * F * H; }
int64_t layer_f32_floats(int H, int F) { return (int64_t)3 * H + H + H + H + F + H + H + H; }

int num_cus() {
  static int n = [] {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
      hipDeviceProp_t p;
      if (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) cus = p.multiProcessorCount;
    }
    return cus;
  }();
  return n;
}

template <typename T>
void launch_attention(const AttnArgs& at, int S, unsigned nblk, hipStream_t s) {
  if (S == 256) {
    // persistent + double-buffered (bert_attn.cuh); CAPAMD_ATTN_ONESHOT=1 selects the one-shot kernel for A/B runs
    static const bool oneshot = [] { const char* e = getenv("CAPAMD_ATTN_ONESHOT"); return e && e[0] == '1'; }();
    if (oneshot) {
      hipLaunchKernelGGL((attention_kernel<256, 8, T>), dim3(nblk), dim3(512), 0, s, at);
    } else {
      constexpr int kAttnLds = 2 * (256 * 128 + 64 * 256 * 2 + 256 * 4);
      static bool attr_set = false;
      if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_persistent_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, kAttnLds);
        attr_set = true;
      }
      const unsigned grid = nblk < (unsigned)num_cus() ? nblk : (unsigned)num_cus();
      hipLaunchKernelGGL((attention_persistent_kernel<T>), dim3(grid), dim3(512), kAttnLds, s, at, (int)nblk);
    }
  } else if (S == 128) hipLaunchKernelGGL((attention_kernel<128, 4, T>), dim3(nblk), dim3(256), 0, s, at);
  else hipLaunchKernelGGL((attention_kernel<64, 2, T>), dim3(nblk), dim3(128), 0, s, at);
}

// Column tiles scheduled together.  All of them when the whole weight matrix fits an XCD's 4 MB L2 next to the activation
// panels in flight (<= 3.6 MB) or there are few (every group re-reads the activations once: tn / group passes over A);
// otherwise the largest divisor of tn whose weight panels (256 x K, 16-bit) stay within ~1.6 MB.  Measured on MI355X,
// FFN1 of BERT-base (tn = 12, 4.7 MB of weights): 381 us with 12, 358 with 4 or 3, 391 with 1.  CAPAMD_GEMM_NGROUP overrides.
int column_group(int tn, int K) {
  static const int forced = [] { const char* e = getenv("CAPAMD_GEMM_NGROUP"); return e ? atoi(e) : 0; }();
  if (forced > 0 && tn % forced == 0) return forced;
  const long panel = 256L * K * 2;
  if (tn <= 4 || tn * panel <= 3774873L) return tn;
  int best = 1;
  for (int g = 1; g <= tn; ++g)
    if (tn % g == 0 && g * panel <= 1677721L) best = g;
  return best;
}

template <int EPI, typename T>
hipError_t launch_gemm(const GemmArgs& g, hipStream_t s) {
  // CAPAMD_GEMM_KLOOP=halves selects the older 256x256 kernel (k-half regions, 4x2 waves) for A/B runs
  static const bool pingpong = [] { const char* e = getenv("CAPAMD_GEMM_KLOOP"); return !(e && e[0] == 'h'); }();
  if (pingpong && EPI != kEpiBiasResidBf16 && g.M % 256 == 0 && g.N % 256 == 0 && g.K >= 128 && (size_t)g.M * g.K < (1ull << 31) &&
      (size_t)g.N * g.K < (1ull << 31)) {  // (buffer addressing: operands below 4 GiB)
    using P = GemmPingPong<EPI, T>;
    auto k = gemm_pingpong_kernel<EPI, T>;
    static bool attr_set = false;
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, P::kLdsBytes);
      if (e != hipSuccess) return e;
      attr_set = true;
    }
    const int tiles = (g.N / 256) * (g.M / 256), grid = tiles < num_cus() ? tiles : num_cus();  // one persistent workgroup per CU
    GemmArgs gg = g;
    gg.ngroup = column_group(g.N / 256, g.K);
    hipLaunchKernelGGL(k, dim3(grid), dim3(P::kThreads), P::kLdsBytes, s, gg);
  } else if (g.M % 256 == 0 && g.N % 256 == 0) {
    using G = GemmKernel<256, 256, 4, 2, EPI, T>;
    auto k = gemm_bf16_kernel<256, 256, 4, 2, EPI, T>;
    static bool attr_set = false;
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, G::kLdsBytes);
      if (e != hipSuccess) return e;
      attr_set = true;
    }
    const int tiles = (g.N / 256) * (g.M / 256), grid = tiles < num_cus() ? tiles : num_cus();  // one persistent workgroup per CU
    hipLaunchKernelGGL(k, dim3(grid), dim3(G::kThreads), G::kLdsBytes, s, g);
  } else {
    using G = GemmKernel<64, 64, 2, 2, EPI, T>;
    const int tiles = (g.N / 64) * (g.M / 64), cap = 4 * num_cus(), grid = tiles < cap ? tiles : cap;
    hipLaunchKernelGGL((gemm_bf16_kernel<64, 64, 2, 2, EPI, T>), dim3(grid), dim3(G::kThreads), G::kLdsBytes, s, g);
  }
  return hipGetLastError();
}

// profiling hook (capamd_debug_ffn1_timing): HIP events around the FFN1 launches of the timed forward passes
struct Ffn1Timing {
  static inline bool on = false;
  static inline std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
  static inline int64_t rows = 0;
  static void begin(hipStream_t s) {
    if (!on) return;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    ev.emplace_back(a, b);
    hipEventRecord(a, s);
  }
  static void end(hipStream_t s, int64_t m) {
    if (!on) return;
    hipEventRecord(ev.back().second, s);
    rows += m;
  }
};

struct Workspace {
  uint16_t* xb;   // [M, H] activation / residual stream      (16-bit type T of the model: bf16 or fp16)
  uint16_t* q;    // [M, H]
  uint16_t* k;    // [M, H]
  uint16_t* vt;   // [M/S*heads, 64, S]
  uint16_t* ctx;  // [M, H]
  uint16_t* pre;  // [M, H] pre-LayerNorm sums
  uint16_t* mid;  // [M, F]
  float* logits;  // [B*P] (whole call)
  int* cnt;       // avg denominator
};

size_t ws_bytes_for(int H, int F, int S, int64_t n_psg_mb, int64_t n_psg_total) {
  const int64_t M = n_psg_mb * S;
  size_t b = 0;
  auto add = [&](size_t x) { b += (x + 255) & ~(size_t)255; };
  add((size_t)M * H * 2); add((size_t)M * H * 2); add((size_t)M * H * 2); add((size_t)M * H * 2);
  add((size_t)M * H * 2); add((size_t)M * H * 2); add((size_t)M * F * 2); add((size_t)n_psg_total * 4); add(256);
  return b;
}

Workspace carve(char* p, int H, int F, int S, int64_t n_psg_mb, int64_t n_psg_total) {
  const int64_t M = n_psg_mb * S;
  Workspace w;
  auto take = [&](size_t x) { char* r = p; p += (x + 255) & ~(size_t)255; return r; };
  w.xb = (uint16_t*)take((size_t)M * H * 2);
  w.q = (uint16_t*)take((size_t)M * H * 2);
  w.k = (uint16_t*)take((size_t)M * H * 2);
  w.vt = (uint16_t*)take((size_t)M * H * 2);
  w.ctx = (uint16_t*)take((size_t)M * H * 2);
  w.pre = (uint16_t*)take((size_t)M * H * 2);
  w.mid = (uint16_t*)take((size_t)M * F * 2);
  w.logits = (float*)take((size_t)n_psg_total * 4);
  w.cnt = (int*)take(256);
  return w;
}

template <typename T>
hipError_t encode_passages(const int64_t* ids, const int64_t* mask, const int64_t* seg, int64_t NP, int64_t mb, int S,
                           const capamd_bert_model* m, const Workspace& w, int* status, hipStream_t s) {
  const int H = m->hidden, F = m->ffn;
  const T* blob = (const T*)m->blob;
  hipError_t e = hipSuccess;

  for (int64_t p0 = 0; p0 < NP && e == hipSuccess; p0 += mb) {
    const int64_t np = (NP - p0 < mb) ? NP - p0 : mb;
    const int64_t M = np * S;
    const int64_t* ids_mb = ids + p0 * S;
    const int64_t* mask_mb = mask + p0 * S;
    const int64_t* seg_mb = seg + p0 * S;
    hipLaunchKernelGGL((ln_kernel<0, T>), dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, (const T*)nullptr, ids_mb, seg_mb, m->word_emb, m->pos_emb,
                       m->type_emb, m->vocab, m->type_vocab, S, m->emb_ln_g, m->emb_ln_b, M, H, (T*)w.xb, status);
    for (int l = 0; l < m->layers && e == hipSuccess; ++l) {
      const T* wl = blob + (int64_t)l * layer_blob_elems(H, F);
      const float* fl = m->layer_f32 + (int64_t)l * layer_f32_floats(H, F);
      const T *wqkv = wl, *wo = wl + (int64_t)3 * H * H, *w1 = wl + (int64_t)4 * H * H, *w2 = w1 + (int64_t)F * H;
      const float *bqkv = fl, *bo = fl + 3 * H, *ln1g = fl + 4 * H, *ln1b = fl + 5 * H, *b1 = fl + 6 * H, *b2 = fl + 6 * H + F,
                  *ln2g = fl + 7 * H + F, *ln2b = fl + 8 * H + F;
      GemmArgs g{};
      g.H = H; g.S = S; g.heads = m->heads;
      // QKV projection (+bias, Q/8, V transposed per head)
      g.A = w.xb; g.W = wqkv; g.bias = bqkv; g.M = (int)M; g.N = 3 * H; g.K = H; g.out_bf16 = w.q; g.out_k = w.k; g.out_vt = w.vt;
      e = launch_gemm<kEpiQkv, T>(g, s);
      if (e != hipSuccess) break;
      AttnArgs at{w.q, w.k, w.vt, mask_mb, w.ctx, H, m->heads};
      const unsigned nblk = (unsigned)(np * m->heads);
      launch_attention<T>(at, S, nblk, s);
      // attention output projection; the residual is added in fp32 inside the LayerNorm pass (one rounding, and the
      // GEMM keeps its cheap 16-bit epilogue)
      g.A = w.ctx; g.W = wo; g.bias = bo; g.N = H; g.K = H; g.out_bf16 = w.pre;
      e = launch_gemm<kEpiBiasBf16, T>(g, s);
      if (e != hipSuccess) break;
      hipLaunchKernelGGL((ln_kernel<2, T>), dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, (const T*)w.pre, nullptr, nullptr, nullptr, nullptr, nullptr,
                         0, 0, S, ln1g, ln1b, M, H, (T*)w.xb, status);
      // feed-forward: 768 -> 3072 (GELU) -> 768, residual + LayerNorm
      g.A = w.xb; g.W = w1; g.bias = b1; g.N = F; g.K = H; g.out_bf16 = w.mid;
      Ffn1Timing::begin(s);
      e = launch_gemm<kEpiBiasGeluBf16, T>(g, s);
      Ffn1Timing::end(s, M);
      if (e != hipSuccess) break;
      g.A = w.mid; g.W = w2; g.bias = b2; g.N = H; g.K = F; g.out_bf16 = w.pre;
      e = launch_gemm<kEpiBiasBf16, T>(g, s);
      if (e != hipSuccess) break;
      hipLaunchKernelGGL((ln_kernel<2, T>), dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, (const T*)w.pre, nullptr, nullptr, nullptr, nullptr, nullptr,
                         0, 0, S, ln2g, ln2b, M, H, (T*)w.xb, status);
    }
    if (e != hipSuccess) break;
    // (the partial sums reuse the pre-LayerNorm buffer, which is dead after the last layer)
    float* hpart = reinterpret_cast<float*>(w.pre);
    hipLaunchKernelGGL(head_kernel<T>, dim3((unsigned)((np + kHeadPsg - 1) / kHeadPsg), (unsigned)(H / 64)), dim3(256), 0, s, (const T*)w.xb, np, S, H,
                       m->pooler_w, m->pooler_b, m->cls_w, hpart);
    hipLaunchKernelGGL(head_reduce_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, s, hpart, np, H / 64, m->cls_b, w.logits + p0);
    e = hipGetLastError();
  }
  return e;
}

template <typename T>
int gemm_dispatch(GemmArgs& g, int epilogue, const void* resid, void* out, hipStream_t s) {
  hipError_t e;
  g.out_bf16 = out;
  if (epilogue == kEpiBiasBf16) e = launch_gemm<kEpiBiasBf16, T>(g, s);
  else if (epilogue == kEpiBiasGeluBf16) e = launch_gemm<kEpiBiasGeluBf16, T>(g, s);
  else if (epilogue == kEpiBiasResidBf16) {
    if (!resid) return CAPAMD_ERR_ARG;
    g.resid_bf16 = resid;
    e = launch_gemm<kEpiBiasResidBf16, T>(g, s);
  } else return CAPAMD_ERR_ARG;
  return e == hipSuccess ? CAPAMD_OK : CAPAMD_ERR_LAUNCH;
}

}  // namespace

extern "C" {

int64_t capamd_bert_blob_bytes(const capamd_bert_model* m) {
  return dims_ok(m) ? (int64_t)m->layers * layer_blob_elems(m->hidden, m->ffn) * 2 : -1;
}
int64_t capamd_bert_layer_f32_floats(const capamd_bert_model* m) { return dims_ok(m) ? layer_f32_floats(m->hidden, m->ffn) : -1; }

int capamd_bert_pack_layer(const capamd_bert_model* m, int layer, const float* const* t /* 16 device pointers, host array */,
                           void* blob, float* layer_f32, void* stream) {
  if (!dims_ok(m) || !t || !blob || !layer_f32 || layer < 0 || layer >= m->layers) return CAPAMD_ERR_ARG;
  for (int i = 0; i < 16; ++i)
    if (!t[i]) return CAPAMD_ERR_ARG;
  const int64_t H = m->hidden, F = m->ffn;
  hipStream_t s = (hipStream_t)stream;
  (void)hipGetLastError();
  uint16_t* wb = (uint16_t*)blob + (int64_t)layer * layer_blob_elems(H, F);
  float* fb = layer_f32 + (int64_t)layer * layer_f32_floats(H, F);
  // order of t: q.w q.b k.w k.b v.w v.b o.w o.b ln1.g ln1.b ffn1.w ffn1.b ffn2.w ffn2.b ln2.g ln2.b
  struct { int src; int64_t off, n; } wcp[] = {{0, 0, H * H}, {2, H * H, H * H}, {4, 2 * H * H, H * H}, {6, 3 * H * H, H * H},
                                               {10, 4 * H * H, F * H}, {12, 4 * H * H + F * H, H * F}};
  for (auto& c : wcp) {
    if (m->compute_dtype == 1) hipLaunchKernelGGL(cvt_bf16_kernel<_Float16>, dim3(512), dim3(256), 0, s, t[c.src], (_Float16*)(wb + c.off), c.n);
    else hipLaunchKernelGGL(cvt_bf16_kernel<__bf16>, dim3(512), dim3(256), 0, s, t[c.src], (__bf16*)(wb + c.off), c.n);
  }
  struct { int src; int64_t off, n; } fcp[] = {{1, 0, H}, {3, H, H}, {5, 2 * H, H}, {7, 3 * H, H}, {8, 4 * H, H}, {9, 5 * H, H},
                                               {11, 6 * H, F}, {13, 6 * H + F, H}, {14, 7 * H + F, H}, {15, 8 * H + F, H}};
  for (auto& c : fcp) hipLaunchKernelGGL(copy_f32_kernel, dim3(8), dim3(256), 0, s, t[c.src], fb + c.off, c.n);
  return hipGetLastError() == hipSuccess ? CAPAMD_OK : CAPAMD_ERR_LAUNCH;
}

int64_t capamd_bert_workspace_bytes(const capamd_bert_model* m, int S, int64_t passages_per_microbatch, int64_t total_passages) {
  if (!dims_ok(m) || S < 1 || passages_per_microbatch < 1 || total_passages < 1) return -1;
  return (int64_t)ws_bytes_for(m->hidden, m->ffn, S, passages_per_microbatch, total_passages);
}

int capamd_bert_maxp_forward(const int64_t* ids, const int64_t* mask, const int64_t* seg, int B, int P, int S,
                             const capamd_bert_model* m, int aggregation, int64_t passages_per_microbatch, void* workspace,
                             int64_t workspace_bytes, float* out, float* passage_logits_out, int* status, void* stream) {
  if (B == 0) return CAPAMD_OK;
  if (!ids || !mask || !seg || !dims_ok(m) || !workspace || !out || !status || B < 0 || P < 1) return CAPAMD_ERR_ARG;
  if (!(S == 64 || S == 128 || S == 256) || S > m->max_pos || aggregation < 0 || aggregation > 3) return CAPAMD_ERR_ARG;
  if (!m->word_emb || !m->pos_emb || !m->type_emb || !m->emb_ln_g || !m->emb_ln_b || !m->pooler_w || !m->pooler_b || !m->cls_w ||
      !m->cls_b || !m->blob || !m->layer_f32)
    return CAPAMD_ERR_ARG;
  const int H = m->hidden, F = m->ffn;
  const int64_t NP = (int64_t)B * P;
  if (passages_per_microbatch < 1) return CAPAMD_ERR_ARG;
  // equal-sized micro-batches (4000 passages at a cap of 256 -> 16 x 250, not 15 x 256 + 160: no ragged last pass)
  const int64_t n_mb = (NP + passages_per_microbatch - 1) / passages_per_microbatch;
  const int64_t mb = (NP + n_mb - 1) / n_mb;
  if ((int64_t)ws_bytes_for(H, F, S, mb, NP) > workspace_bytes) return CAPAMD_ERR_WORKSPACE;
  if ((reinterpret_cast<uintptr_t>(workspace) & 255) != 0) return CAPAMD_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  (void)hipGetLastError();
  Workspace w = carve((char*)workspace, H, F, S, mb, NP);
  const hipError_t e = (m->compute_dtype == 1)
                           ? encode_passages<_Float16>(ids, mask, seg, NP, mb, S, m, w, status, s)
                           : encode_passages<__bf16>(ids, mask, seg, NP, mb, S, m, w, status, s);
  if (e != hipSuccess) return CAPAMD_ERR_LAUNCH;
  if (aggregation == 3) (void)hipMemsetAsync(w.cnt, 0, 4, s);
  hipLaunchKernelGGL(pool_kernel, dim3(B), dim3(64), 0, s, w.logits, mask, seg, P, S, aggregation, out, w.cnt);
  if (aggregation == 3) hipLaunchKernelGGL(avg_div_kernel, dim3((B + 255) / 256), dim3(256), 0, s, out, B, w.cnt);
  if (passage_logits_out) (void)hipMemcpyAsync(passage_logits_out, w.logits, (size_t)NP * 4, hipMemcpyDeviceToDevice, s);
  return hipGetLastError() == hipSuccess ? CAPAMD_OK : CAPAMD_ERR_LAUNCH;
}

/* building blocks, exported for unit tests and for callers that want the encoder pieces */
void capamd_debug_ffn1_timing(int enable) { Ffn1Timing::on = enable != 0; }
int capamd_debug_ffn1_timing_read(double* total_ms, int64_t* launches, int64_t* rows) {
  if (!total_ms || !launches || !rows) return CAPAMD_ERR_ARG;
  double t = 0;
  for (auto& p : Ffn1Timing::ev) {
    float ms = 0.f;
    if (hipEventSynchronize(p.second) != hipSuccess || hipEventElapsedTime(&ms, p.first, p.second) != hipSuccess) return CAPAMD_ERR_LAUNCH;
    t += ms;
    hipEventDestroy(p.first); hipEventDestroy(p.second);
  }
  *total_ms = t; *launches = (int64_t)Ffn1Timing::ev.size(); *rows = Ffn1Timing::rows;
  Ffn1Timing::ev.clear(); Ffn1Timing::rows = 0;
  return CAPAMD_OK;
}
static unsigned long long* g_gemm_dbg = nullptr;
void capamd_debug_set_gemm_stamps(void* p) { g_gemm_dbg = (unsigned long long*)p; }  /* profiling hook (scripts/gemm_timeline.py) */

int capamd_bert_gemm(const void* A, const void* W, const float* bias, int M, int N, int K, int epilogue, const void* resid,
                     void* out, int dtype, void* stream) {
  if (!A || !W || !bias || !out || M < 64 || N < 64 || K < 64 || M % 64 || N % 64 || K % 64) return CAPAMD_ERR_ARG;
  (void)hipGetLastError();
  GemmArgs g{};
  g.A = A; g.W = W; g.bias = bias; g.M = M; g.N = N; g.K = K;
  g.dbg = g_gemm_dbg;
  if (dtype == 1) return gemm_dispatch<_Float16>(g, epilogue, resid, out, (hipStream_t)stream);
  if (dtype == 0) return gemm_dispatch<__bf16>(g, epilogue, resid, out, (hipStream_t)stream);
  return CAPAMD_ERR_ARG;
}

int capamd_bert_qkv_attention(const void* x, const void* wqkv, const float* bqkv, const int64_t* mask, int n_passages, int S,
                              int hidden, int heads, void* q, void* k, void* vt, void* ctx, int dtype, void* stream) {
  if (!x || !wqkv || !bqkv || !mask || !q || !k || !vt || !ctx || n_passages < 1 || heads * 64 != hidden) return CAPAMD_ERR_ARG;
  if (!(S == 64 || S == 128 || S == 256)) return CAPAMD_ERR_ARG;
  (void)hipGetLastError();
  hipStream_t s = (hipStream_t)stream;
  GemmArgs g{};
  g.H = hidden; g.S = S; g.heads = heads;
  g.A = x; g.W = wqkv; g.bias = bqkv; g.M = n_passages * S; g.N = 3 * hidden; g.K = hidden;
  g.out_bf16 = q; g.out_k = k; g.out_vt = vt;
  if (dtype != 0 && dtype != 1) return CAPAMD_ERR_ARG;
  const hipError_t e = dtype == 1 ? launch_gemm<kEpiQkv, _Float16>(g, s) : launch_gemm<kEpiQkv, __bf16>(g, s);
  if (e != hipSuccess) return CAPAMD_ERR_LAUNCH;
  AttnArgs at{q, k, vt, mask, ctx, hidden, heads};
  const unsigned nblk = (unsigned)(n_passages * heads);
  if (dtype == 1) launch_attention<_Float16>(at, S, nblk, s);
  else launch_attention<__bf16>(at, S, nblk, s);
  return hipGetLastError() == hipSuccess ? CAPAMD_OK : CAPAMD_ERR_LAUNCH;
}

}  // extern "C"
