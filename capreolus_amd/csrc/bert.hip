// BERT-MaxP passage scoring for gfx950: PTBERTMaxP_Class.predict_step (reference
// capreolus/reranker/ptBERTMaxP.py:67-96) with the transformers BertForSequenceClassification it
// calls at :82 re-built as hand-written kernels: embedding sum + LayerNorm, 16-bit MFMA GEMMs with
// fused bias / GELU / QKV epilogues (bert_gemm.h), fused exact-softmax attention (bert_attn.h),
// residual + LayerNorm, pooler + classifier, passage pooling.
//
// Precision: the activation stream is 16-bit end to end (fp16 by default - the reference's autocast type - or
// bf16, `compute_dtype` of the model); every accumulation, the residual sums (added in fp32 inside the LayerNorm
// pass, one rounding), LayerNorm statistics, softmax and the pooler/classifier are fp32.  (A fp32 residual stream
// was measured to give the same logit error: the error is set by the 16-bit GEMM operands.)
#include "bert_attn.h"
#include "bert_gemm.h"
#include "bert_gemm_ring16.h"
#include "capreolus_amd.h"
#include "capamd_profiling.h"
#include "cedr_tap.h"
#include <stdlib.h>
#include <string.h>
#include <utility>
#include <vector>

using namespace capamd;

namespace {

constexpr float kLnEps = 1e-12f;  // BertConfig.layer_norm_eps (capamd_bert_model.ln_eps = 0)

// sum over the 64 lanes, every lane gets it: DPP all-reduce inside each 16-lane row, then the four row sums by v_readlane (a
// __shfl_xor butterfly is six ds_bpermute round trips through the LDS pipe, ~100 cycles each, and a LayerNorm row needs two in series)
template <int CTRL>
__device__ __forceinline__ float dpp_mov_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum64(float v) {
  v += dpp_mov_f<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_mov_f<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_mov_f<0x141>(v);   // row_half_mirror
  v += dpp_mov_f<0x140>(v);   // row_mirror
  const int bits = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(bits, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(bits, 16)),
              r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(bits, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(bits, 48));
  return (r0 + r1) + (r2 + r3);
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
                                                 T* xb, int* status, int out_cm, float eps, int pos_pad_id, int max_pos) {
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
    int64_t pidx = tok % S;
    if (pos_pad_id >= 0) {
      // RoBERTa: positions count the non-pad tokens of the passage up to and including this one (HF create_position_ids_from_input_ids)
      const int64_t i = tok % S;
      const int64_t* row = ids + (tok - i);
      int cnt = 0;
      for (int64_t j = lane; j <= i; j += 64) cnt += row[j] != pos_pad_id;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
      pidx = id != pos_pad_id ? pos_pad_id + cnt : pos_pad_id;
    }
    if (pidx >= max_pos) {
      if (lane == 0) atomicOr(status, 1);
      pidx = 0;
    }
    r0 = word + id * H;
    r1 = pos + pidx * H;
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
  const float rstd = rsqrtf(wave_sum64(q) / (float)H + eps);
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
      if (out_cm) *reinterpret_cast<bf16x8*>(xb + cm_offset(tok, c * 8, H)) = ob;   // chunk-major stream (bert_gemm.h)
      else reinterpret_cast<bf16x8*>(xb + tok * H)[c] = ob;
    }
  }
}

// ln_kernel<0> for the chunk-major stream: a block takes one 32-token row group, whose chunk-major image is ONE contiguous run of
// 32 H elements, builds it in LDS (eight waves, four tokens each, all four in flight at once) and copies it out linearly - a wave of ln_kernel writing
// its token straight to the chunk-major layout touches 64 different 128-byte lines with 16 bytes each.  Same arithmetic as ln_kernel<0>.
template <typename T>
__global__ __launch_bounds__(512) void embed_ln_cm_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ seg, const float* __restrict__ word,
                                                          const float* __restrict__ pos, const float* __restrict__ type, int vocab, int type_vocab, int S,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta, int H, T* xb, int* status,
                                                          float eps, int pos_pad_id, int max_pos) {
  using bf16x8 = typename Half<T>::x8;
  extern __shared__ __attribute__((aligned(16))) char stage_raw[];
  bf16x8* stage = reinterpret_cast<bf16x8*>(stage_raw);   // [32 tokens][H / 8 + 1] 16-byte pieces (token-major, rows padded by one piece: conflict-free
                                                          // writes along a token, 2-way conflicts on the transposed read)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nchunk = H >> 3;
  const int pos0 = (int)(((int64_t)blockIdx.x * 32) % S);   // the block's 32 tokens lie inside one passage (S % 32 == 0): one division per block
  constexpr int TB = 4;   // tokens a wave works on at once: their gathers are all in flight before the first reduction (the kernel is latency bound)
  for (int t0 = wave * 4; t0 < wave * 4 + 4; t0 += TB) {
    const float4 *r0[TB], *r1[TB], *r2[TB];
#pragma unroll
    for (int u = 0; u < TB; ++u) {
      const int64_t tok = (int64_t)blockIdx.x * 32 + t0 + u;
      int64_t id = ids[tok], sg = seg[tok];
      if (id < 0 || id >= vocab || sg < 0 || sg >= type_vocab) {
        if (lane == 0) atomicOr(status, 1);
        id = 0;
        sg = 0;
      }
      int64_t pidx = pos0 + t0 + u;
      if (pos_pad_id >= 0) {   // RoBERTa positions (see ln_kernel)
        const int64_t i = pos0 + t0 + u;
        const int64_t* row = ids + (tok - i);
        int cnt = 0;
        for (int64_t j = lane; j <= i; j += 64) cnt += row[j] != pos_pad_id;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
        pidx = id != pos_pad_id ? pos_pad_id + cnt : pos_pad_id;
      }
      if (pidx >= max_pos) {
        if (lane == 0) atomicOr(status, 1);
        pidx = 0;
      }
      r0[u] = reinterpret_cast<const float4*>(word + id * H);
      r1[u] = reinterpret_cast<const float4*>(pos + pidx * H);
      r2[u] = reinterpret_cast<const float4*>(type + sg * H);
    }
    float v[TB][2][8];
    float s[TB];
#pragma unroll
    for (int u = 0; u < TB; ++u) {
      s[u] = 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const float4 x = r0[u][2 * c + h], y = r1[u][2 * c + h], z = r2[u][2 * c + h];
            v[u][i][4 * h + 0] = x.x + (y.x + z.x); v[u][i][4 * h + 1] = x.y + (y.y + z.y);
            v[u][i][4 * h + 2] = x.z + (y.z + z.z); v[u][i][4 * h + 3] = x.w + (y.w + z.w);
          }
#pragma unroll
          for (int e = 0; e < 8; e += 4) s[u] += (v[u][i][e] + v[u][i][e + 1]) + (v[u][i][e + 2] + v[u][i][e + 3]);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < TB; ++u) {
      const int tl = t0 + u;
      const float mean = wave_sum64(s[u]) / (float)H;
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i)
        if (lane + 64 * i < nchunk) {
#pragma unroll
          for (int e = 0; e < 8; e += 4) {
            const float a = v[u][i][e] - mean, b = v[u][i][e + 1] - mean, c = v[u][i][e + 2] - mean, d = v[u][i][e + 3] - mean;
            q += (a * a + b * b) + (c * c + d * d);
          }
        }
      const float rstd = rsqrtf(wave_sum64(q) / (float)H + eps);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) {
          bf16x8 ob;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const float4 g = reinterpret_cast<const float4*>(gamma)[2 * c + h], b = reinterpret_cast<const float4*>(beta)[2 * c + h];
            ob[4 * h + 0] = (T)((v[u][i][4 * h + 0] - mean) * rstd * g.x + b.x);
            ob[4 * h + 1] = (T)((v[u][i][4 * h + 1] - mean) * rstd * g.y + b.y);
            ob[4 * h + 2] = (T)((v[u][i][4 * h + 2] - mean) * rstd * g.z + b.z);
            ob[4 * h + 3] = (T)((v[u][i][4 * h + 3] - mean) * rstd * g.w + b.w);
          }
          stage[tl * (nchunk + 1) + c] = ob;
        }
      }
    }
  }
  __syncthreads();
  bf16x8* dst = reinterpret_cast<bf16x8*>(xb + (int64_t)blockIdx.x * 32 * H);
  for (int i = threadIdx.x; i < nchunk * 32; i += 512) dst[i] = stage[(i & 31) * (nchunk + 1) + (i >> 5)];   // chunk-major: piece (chunk i >> 5, token i & 31)
}

// pooler tanh(Wp h_CLS + bp) and classifier logit 1 (ptBERTMaxP.py:82 takes [:, 1]).
// grid (ceil(n_psg / kHeadPsg), H / kHeadRows): a block owns kHeadRows pooler rows (8 per wave) and kHeadPsg passages.  A lane holds
// elements lane + 64 c of its wave's 8 weight rows (coalesced fp32 reads of the live parameters) and accumulates the 8 x 8 partial
// dot products against the [CLS] rows staged in LDS; the 64 partials are then summed over the 64 lanes by ONE transposing butterfly
// (step k keeps the half of the values the lane's bit 5-k selects and adds the partner's copy of them: 32+16+..+1 = 63 exchanges
// instead of 64 x 6, and lane l ends up with the complete dot product of (row l >> 3, passage l & 7)), so tanh runs once per output
// and not once per lane.  The partial sum over the block's rows of cls_w[1][j] * tanh(pooler_j) goes to part[psg][slice];
// head_reduce_kernel adds the slices in fixed order (deterministic, no atomics; a passage's logit does not depend on its neighbours).
constexpr int kHeadPsg = 8, kHeadRows = 32;
template <typename T>
__global__ __launch_bounds__(256) void head_kernel(const T* __restrict__ xf, int64_t n_psg, int S, int H,
                                                   const float* __restrict__ pw, const float* __restrict__ pb,
                                                   const float* __restrict__ cw, float* __restrict__ part,
                                                   const float* __restrict__ ln_mu, const float* __restrict__ ln_rstd,
                                                   const float* __restrict__ ln_g, const float* __restrict__ ln_b) {
  __shared__ float cls[kHeadPsg][1024];
  __shared__ float wsum[4][kHeadPsg];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t p0 = (int64_t)blockIdx.x * kHeadPsg;
  const int nslice = gridDim.y, j0 = blockIdx.y * kHeadRows + wave * 8;
#pragma unroll
  for (int q = 0; q < kHeadPsg; ++q) {
    const int64_t psg = p0 + q < n_psg ? p0 + q : n_psg - 1;
    const int64_t tok = psg * S;    // token 0 ([CLS]) of the passage
    if (ln_mu) {  // fused-LayerNorm path: xf holds the chunk-major pre-LayerNorm sums of the last layer
      const float mu = ln_mu[tok], rs = ln_rstd[tok];
      for (int i = tid; i < H; i += 256) cls[q][i] = ((float)xf[cm_offset(tok, i, H)] - mu) * rs * ln_g[i] + ln_b[i];
    } else {
      const T* h = xf + tok * H;
      for (int i = tid; i < H; i += 256) cls[q][i] = (float)h[i];
    }
  }
  __syncthreads();
  float p[64];   // p[8 r + q]: this lane's share of <pooler row j0 + r, [CLS] of passage q>
#pragma unroll
  for (int i = 0; i < 64; ++i) p[i] = 0.f;
  const int nc = H >> 6;  // <= 16
  for (int c = 0; c < nc; ++c) {
    float wr[8], xq[kHeadPsg];
#pragma unroll
    for (int r = 0; r < 8; ++r) wr[r] = pw[(int64_t)(j0 + r) * H + lane + 64 * c];
#pragma unroll
    for (int q = 0; q < kHeadPsg; ++q) xq[q] = cls[q][lane + 64 * c];
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int q = 0; q < kHeadPsg; ++q) p[8 * r + q] = __builtin_fmaf(wr[r], xq[q], p[8 * r + q]);
  }
  // transposing butterfly: 64 values x 64 lanes -> lane l holds the sum over all lanes of value l
#pragma unroll
  for (int half = 32; half >= 1; half >>= 1) {
    const bool up = (lane & half) != 0;
#pragma unroll
    for (int i = 0; i < half; ++i) {
      const float keep = up ? p[half + i] : p[i], send = up ? p[i] : p[half + i];
      p[i] = keep + __shfl_xor(send, half, 64);
    }
  }
  const int r = lane >> 3, q = lane & 7, j = j0 + r;
  float contrib = cw[H + j] * tanhf(p[0] + pb[j]);
  contrib += __shfl_xor(contrib, 8, 64);     // over the wave's 8 rows, fixed order
  contrib += __shfl_xor(contrib, 16, 64);
  contrib += __shfl_xor(contrib, 32, 64);
  if (lane < kHeadPsg) wsum[wave][q] = contrib;
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
         (m->compute_dtype == 0 || m->compute_dtype == 1) && m->ln_eps >= 0.f && m->pos_pad_id < m->vocab;
}

// LayerNorm folded into the GEMMs (bert_gemm.h) needs every encoder GEMM on the ping-pong kernel: N and K multiples of 256
bool fused_capable(int H, int F) { return H % 256 == 0 && F % 256 == 0; }
// per layer, 16-bit: wqkv [3H,H] | wo [H,H] | w1 [F,H] | w2 [H,F]   (+ wqkv' = wqkv . gamma_in | w1' = w1 . ln1_gamma when fused_capable)
//   (+ when fused_capable: a chunk-major copy of all six, same order, for the ring kernel)
int64_t layer_rowmajor_elems(int H, int F) {
  return (int64_t)3 * H * H + (int64_t)H * H + (int64_t)2 * F * H + (fused_capable(H, F) ? (int64_t)3 * H * H + (int64_t)F * H : 0);
}
int64_t layer_blob_elems(int H, int F) { return layer_rowmajor_elems(H, F) * (fused_capable(H, F) ? 2 : 1); }
// per layer, fp32: bqkv 3H | bo H | ln1g H | ln1b H | b1 F | b2 H | ln2g H | ln2b H
//                | cs_qkv 3H | c_qkv 3H | cs_1 F | c_1 F | g_in H | bo + beta_in H | b2 + ln1b H      (folded-LayerNorm vectors)
int64_t layer_f32_floats(int H, int F) { return (int64_t)9 * H + F + (int64_t)9 * H + 2 * F; }

int num_cus();
// Workgroup slots a persistent ring GEMM asks for: all CUs, or - CAPAMD_GEMM_CU_SHARE=n - 1/n of them, so that the kernels of n streams run
// side by side on disjoint CUs with their phases (K loop / HBM-bound epilogue) out of step instead of one after the other.
int gemm_cus() {
  static const int share = [] { const char* e = getenv("CAPAMD_GEMM_CU_SHARE"); const int v = e ? atoi(e) : 1; return v >= 1 && v <= 8 ? v : 1; }();
  return num_cus() / share;
}
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
    // two 4-wave workgroups per CU, 64 queries per wave (bert_attn.h: attention_s256_kernel); CAPAMD_ATTN=oneshot | persistent
    // select the earlier kernels for A/B runs
    static const int which = [] {
      const char* e = getenv("CAPAMD_ATTN");
      if (e && e[0] == 'o') return 1;
      if (e && e[0] == 'p') return 2;
      return 0;
    }();
    if (which == 0) {
      const unsigned slots = 2u * (unsigned)num_cus();
      const dim3 grid(nblk < slots ? nblk : slots), block(256);
      if (at.qk_cm && at.ctx_cm) hipLaunchKernelGGL((attention_s256_kernel<T, true, true>), grid, block, 0, s, at, (int)nblk);
      else if (at.qk_cm) hipLaunchKernelGGL((attention_s256_kernel<T, true, false>), grid, block, 0, s, at, (int)nblk);
      else if (at.ctx_cm) hipLaunchKernelGGL((attention_s256_kernel<T, false, true>), grid, block, 0, s, at, (int)nblk);
      else hipLaunchKernelGGL((attention_s256_kernel<T, false, false>), grid, block, 0, s, at, (int)nblk);
    } else if (which == 1) {
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
  } else if (S > 256) {
    // 384 / 512: keys walked in chunks with the running-maximum softmax (bert_attn.h), K and V^T of the passage in LDS
    constexpr int kLds384 = 384 * 128 + 64 * (384 * 2 + 8) + 384 * 4, kLds512 = 512 * 128 + 64 * (512 * 2 + 8) + 512 * 4;
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_long_kernel<384, T>), hipFuncAttributeMaxDynamicSharedMemorySize, kLds384);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_long_kernel<512, T>), hipFuncAttributeMaxDynamicSharedMemorySize, kLds512);
      attr_set = true;
    }
    if (S == 384) hipLaunchKernelGGL((attention_long_kernel<384, T>), dim3(nblk), dim3(768), kLds384, s, at);
    else hipLaunchKernelGGL((attention_long_kernel<512, T>), dim3(nblk), dim3(1024), kLds512, s, at);
  } else {
    // every other multiple of 32: the one-shot kernel, one wave per 32 queries (the length buckets of BertEngine)
    switch (S) {
      case 32: hipLaunchKernelGGL((attention_kernel<32, 1, T>), dim3(nblk), dim3(64), 0, s, at); break;
      case 64: hipLaunchKernelGGL((attention_kernel<64, 2, T>), dim3(nblk), dim3(128), 0, s, at); break;
      case 96: hipLaunchKernelGGL((attention_kernel<96, 3, T>), dim3(nblk), dim3(192), 0, s, at); break;
      case 128: hipLaunchKernelGGL((attention_kernel<128, 4, T>), dim3(nblk), dim3(256), 0, s, at); break;
      case 160: hipLaunchKernelGGL((attention_kernel<160, 5, T>), dim3(nblk), dim3(320), 0, s, at); break;
      case 192: hipLaunchKernelGGL((attention_kernel<192, 6, T>), dim3(nblk), dim3(384), 0, s, at); break;
      // (seven waves spread badly over four SIMDs: four waves of two query blocks each, 146 -> 126 us per 3072 blocks; the same form is
      // slower at S = 160 / 192 - three waves, 70 -> 75 and 94 -> 137 us)
      default: hipLaunchKernelGGL((attention_kernel<224, 4, T, 2>), dim3(nblk), dim3(256), 0, s, at); break;
    }
  }
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

// Shapes the ping-pong kernel takes (buffer addressing: operands below 4 GiB).  CAPAMD_GEMM_KLOOP=halves selects the
// older 256x256 kernel (k-half regions, 4x2 waves) for A/B runs; CAPAMD_GEMM_CM=0 keeps every activation row-major.
bool pingpong_shape(int64_t M, int N, int K) {
  static const bool pingpong = [] { const char* e = getenv("CAPAMD_GEMM_KLOOP"); return !(e && e[0] == 'h'); }();
  return pingpong && M % 256 == 0 && N % 256 == 0 && K >= 128 && K % 64 == 0 && (size_t)M * K < (1ull << 31) && (size_t)N * K < (1ull << 31);
}
// Shapes the 4-wave ring kernel takes (bert_gemm_ring.h): whole 256 x 256 tiles, at least 16 k-slices (the ring holds 8 and the
// tile loop has a head and a tail of 8), an even number of them.  CAPAMD_GEMM_RING=0 keeps the 8-wave ping-pong kernel (A/B runs).
bool ring_shape(int64_t M, int N, int K) {
  return M % 256 == 0 && N % 256 == 0 && K % 32 == 0 && K >= 256 && (size_t)M * K < (1ull << 31) && (size_t)N * K < (1ull << 31);
}
int ring_rows() {   // CAPAMD_RING_BM=256: one workgroup per CU with 128 x 128 wave tiles; default 128: two workgroups per CU (see bert_gemm_ring.h)
  static const int bm = [] { const char* e = getenv("CAPAMD_RING_BM"); return (e && atoi(e) == 256) ? 256 : 128; }();
  return bm;
}
// Which ring kernel each of the encoder's four GEMMs runs on: tile rows (256: one workgroup per CU, 128: two) and, for 256, the MFMA
// shape (16x16x32, bert_gemm_ring16.h, or 32x32x16).  Defaults = what measured fastest inside the encoder at M = 64,000
// (profiles/r05/bert_gemm_pick_ab.txt); CAPAMD_GEMM_PICK="qkv=256x16,ffn1=128,oproj=256x32,ffn2=256x32" overrides any of them (A/B runs),
// CAPAMD_RING_BM every one alike.
struct GemmPick { int rows, mfma32, mfma16; };   // mfma32: 256 rows on 32x32x16; mfma16: 128 rows on 16x16x32
GemmPick gemm_pick(int kind /* 0 QKV, 1 FFN1, 2 O-proj, 3 FFN2 */) {
  static const struct Picks { GemmPick p[4]; } picks = [] {
    Picks k{{{256, 0, 0}, {128, 0, 0}, {256, 0, 0}, {256, 0, 0}}};
    const char* names[4] = {"qkv=", "ffn1=", "oproj=", "ffn2="};
    if (const char* e = getenv("CAPAMD_GEMM_PICK"))
      for (int i = 0; i < 4; ++i)
        if (const char* f = strstr(e, names[i])) {
          const char* v = f + strlen(names[i]);
          k.p[i].rows = atoi(v) == 256 ? 256 : 128;
          k.p[i].mfma32 = strncmp(v, "256x32", 6) == 0 ? 1 : 0;
          k.p[i].mfma16 = strncmp(v, "128x16", 6) == 0 ? 1 : 0;
        }
    if (const char* e = getenv("CAPAMD_RING_BM"))
      for (int i = 0; i < 4; ++i) k.p[i].rows = atoi(e) == 256 ? 256 : 128;
    return k;
  }();
  return picks.p[kind];
}
int ring_stagger_override() {
  static const int v = [] { const char* e = getenv("CAPAMD_RING_STAGGER"); return e ? atoi(e) : -1; }();
  return v;
}
bool ring16_enabled() {   // CAPAMD_RING16=0: the 256-row ring kernel on 32x32x16 MFMAs (round 4's; A/B runs)
  static const bool on = [] { const char* e = getenv("CAPAMD_RING16"); return !(e && e[0] == '0'); }();
  return on;
}
bool ring_enabled() {
  static const bool on = [] { const char* e = getenv("CAPAMD_GEMM_RING"); return !(e && e[0] == '0'); }();
  return on;
}
// passage lengths the attention kernels exist for: every multiple of 32 up to 256, then 384 and 512
bool supported_length(int S) { return (S >= 32 && S <= 256 && S % 32 == 0) || S == 384 || S == 512; }
// passages of S tokens (a multiple of 32) that make whole 256-row GEMM tiles: 256 / gcd(S, 256)
int64_t tile_passages(int S) { return (S % 256 == 0) ? 1 : (S % 128 == 0) ? 2 : (S % 64 == 0) ? 4 : 8; }

bool fused_ln_enabled() {  // CAPAMD_BERT_FUSED_LN=0 keeps the separate residual + LayerNorm passes (A/B runs)
  static const bool on = [] { const char* e = getenv("CAPAMD_BERT_FUSED_LN"); return !(e && e[0] == '0'); }();
  return on;
}
bool cedr_fused_enabled() {  // CAPAMD_CEDR_FUSED=0: the tapped encoder on the separate-LayerNorm path (A/B runs)
  static const bool on = [] { const char* e = getenv("CAPAMD_CEDR_FUSED"); return !(e && e[0] == '0'); }();
  return on;
}
bool small_tiles_enabled() {  // CAPAMD_GEMM_SMALL_TILES=0: A/B switch
  static const bool on = [] { const char* e = getenv("CAPAMD_GEMM_SMALL_TILES"); return !(e && e[0] == '0'); }();
  return on;
}
bool cls_tail_enabled() {  // CAPAMD_BERT_CLS_TAIL=0: compute the last layer for every token (A/B runs)
  static const bool on = [] { const char* e = getenv("CAPAMD_BERT_CLS_TAIL"); return !(e && e[0] == '0'); }();
  return on;
}
bool chunk_major_enabled() {
  static const bool on = [] { const char* e = getenv("CAPAMD_GEMM_CM"); return !(e && e[0] == '0'); }();
  return on;
}

template <int EPI, typename T>
hipError_t launch_gemm(const GemmArgs& g, hipStream_t s) {
  if (g.w_cm) {   // both operands chunk-major: the ring kernel or nothing
    if constexpr (EPI != kEpiBiasResidBf16) {
      if (!g.a_cm || !ring_shape(g.M, g.N, g.K) || (EPI == kEpiResidStats && !g.out_cm)) return hipErrorInvalidValue;
      GemmArgs gg = g;
      gg.ngroup = column_group(g.N / 256, g.K);
      const int rows = g.ring_rows ? g.ring_rows : ring_rows();
      const bool r16 = ring16_enabled() && !g.ring_mfma32 && g.out_cm && g.K % 64 == 0;
      if (rows == 256 && r16) {
        // one workgroup per CU on 16x16x32 MFMAs (bert_gemm_ring16.h)
        using R = GemmRing16<EPI, T, 256>;
        auto k = gemm_ring16_kernel<EPI, T, 256>;
        static bool attr_set = false;
        if (!attr_set) {
          hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, R::kLdsBytes);
          if (e != hipSuccess) return e;
          attr_set = true;
        }
        const int tiles = (g.N / 256) * (g.M / 256), grid = tiles < gemm_cus() ? tiles : gemm_cus();
        static const int touch = [] { const char* e = getenv("CAPAMD_R16_TOUCH"); return (e && e[0] == '1') ? 1 : 0; }();
        gg.res_touch = touch;
        hipLaunchKernelGGL(k, dim3(grid), dim3(R::kThreads), R::kLdsBytes, s, gg);
      } else if (rows == 128 && r16 && g.ring_mfma16 && EPI != kEpiQkv) {
        // two workgroups per CU on 16x16x32 MFMAs (the 128-row form of bert_gemm_ring16.h; no V^T tiles)
        if constexpr (EPI != kEpiQkv) {
          using R = GemmRing16<EPI, T, 128>;
          auto k = gemm_ring16_kernel<EPI, T, 128>;
          static bool attr_set = false;
          if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, R::kLdsBytes);
            if (e != hipSuccess) return e;
            attr_set = true;
          }
          const int tiles = (g.N / 256) * (g.M / 128), cap = 2 * gemm_cus(), grid = tiles < cap ? tiles : cap;
          hipLaunchKernelGGL(k, dim3(grid), dim3(R::kThreads), R::kLdsBytes, s, gg);
        }
      } else if ((g.ring_rows ? g.ring_rows : ring_rows()) == 256) {
        using R = GemmRing<EPI, T, 256>;
        auto k = gemm_ring_kernel<EPI, T, 256>;
        static bool attr_set = false;
        if (!attr_set) {
          hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, R::kLdsBytes);
          if (e != hipSuccess) return e;
          attr_set = true;
        }
        const int tiles = (g.N / 256) * (g.M / 256), grid = tiles < gemm_cus() ? tiles : gemm_cus();  // one persistent workgroup per CU
        hipLaunchKernelGGL(k, dim3(grid), dim3(R::kThreads), R::kLdsBytes, s, gg);
      } else {
        using R = GemmRing<EPI, T, 128>;
        auto k = gemm_ring_kernel<EPI, T, 128>;
        static bool attr_set = false;
        if (!attr_set) {
          hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, R::kLdsBytes);
          if (e != hipSuccess) return e;
          attr_set = true;
        }
        const int tiles = (g.N / 256) * (g.M / 128), cap = 2 * gemm_cus(), grid = tiles < cap ? tiles : cap;  // two persistent workgroups per CU
        // (a deliberate half-tile start offset of the grid's second half - CAPAMD_RING_STAGGER, in units of 64 cycles - measured 0.7 %
        // SLOWER end to end than letting the two workgroups of a CU drift apart by themselves: default 0)
        gg.ring_stagger = ring_stagger_override() >= 0 ? ring_stagger_override() : 0;
        hipLaunchKernelGGL(k, dim3(grid), dim3(R::kThreads), R::kLdsBytes, s, gg);
      }
      return hipGetLastError();
    } else {
      return hipErrorInvalidValue;
    }
  }
  if constexpr (EPI != kEpiBiasResidBf16) {
    // a handful of 256x256 tiles would leave most of the chip idle (the last layer's [CLS]-row tail: M = 256): plain
    // row-major GEMMs with fewer than 64 such tiles go to the 64x64-tile kernel instead (16x the workgroups)
    const bool few_tiles = !g.a_cm && !g.out_cm && !g.ln_mu && EPI != kEpiResidStats && EPI != kEpiQkv &&
                           (int64_t)(g.M / 256) * (g.N / 256) < 64 && small_tiles_enabled();
    if (pingpong_shape(g.M, g.N, g.K) && !few_tiles) {
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
      return hipGetLastError();
    }
  }
  if constexpr (EPI == kEpiResidStats) {
    return hipErrorInvalidValue;  // the fused-LayerNorm producer exists on the ping-pong kernel only
  } else {
    if (g.a_cm || g.out_cm || g.ln_mu) return hipErrorInvalidValue;  // layouts / folded LayerNorm: ping-pong kernel only
    const bool few = (int64_t)(g.M / 256) * (g.N / 256) < 64 && small_tiles_enabled();
    if (g.M % 256 == 0 && g.N % 256 == 0 && !few) {
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
}

// ---- folded-LayerNorm packing (once per model) ---------------------------------------------------------------
// one wave per output row n of a LayerNorm-consuming weight matrix:
//   W'[n][k] = (T)(gamma[k] W[n][k] - mean_k(gamma W[n]))    the row is CENTRED: the normalised activations sum to zero over
//              k, so a constant added to a weight row changes nothing - but it removes the common-mode term mu_m * cs_n
//              that the folded form would otherwise have to cancel in fp32 (cs_n shrinks to the rounding residue)
//   cs[n] = sum_k (float)W'[n][k]   (of the ROUNDED operand the MFMA sees),   c[n] = bias[n] + sum_k beta[k] W[n][k]
template <typename T>
__global__ __launch_bounds__(256) void fold_rows_kernel(const float* __restrict__ W, const float* __restrict__ gamma, const float* __restrict__ bias,
                                                        const float* __restrict__ beta, T* __restrict__ Wp, float* __restrict__ cs,
                                                        float* __restrict__ c, int rows, int K) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (n >= rows) return;
  const float* w = W + (int64_t)n * K;
  float m = 0.f, b = 0.f;
  for (int k = lane; k < K; k += 64) {
    m = __builtin_fmaf(gamma[k], w[k], m);
    b = __builtin_fmaf(beta[k], w[k], b);
  }
  m = wave_sum64(m) / (float)K;
  b = wave_sum64(b);
  float a = 0.f;
  for (int k = lane; k < K; k += 64) {
    const T r = (T)(gamma[k] * w[k] - m);
    Wp[(int64_t)n * K + k] = r;
    a += (float)r;
  }
  a = wave_sum64(a);
  if (lane == 0) { cs[n] = a; c[n] = bias[n] + b; }
}
__global__ void vec_fill_kernel(float* __restrict__ out, float v, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = v;
}
__global__ void vec_add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = a[i] + b[i];
}

// row statistics of a pre-LayerNorm sum from the partials its producing GEMM wrote (kEpiResidStats), fixed order
__global__ void ln_stats_kernel(const float* __restrict__ part, int nslot, int H, int64_t M, float* __restrict__ mu, float* __restrict__ rstd,
                                float2* __restrict__ mr, float eps) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  // every slot covers 64 columns: per-slot (mean, M2) merged pairwise-style (equal counts), so that a row whose mean is large
  // against its spread does not lose its variance in E[x^2] - mean^2 over all H columns
  float msum = 0.f, m2 = 0.f;
  for (int k = 0; k < nslot; ++k) {
    const float2 p = *reinterpret_cast<const float2*>(part + (m * nslot + k) * 2);
    const float ms = p.x * (1.f / 64.f);
    msum += ms;
    m2 += fmaxf(p.y - p.x * ms, 0.f);
  }
  const float mean = msum / (float)nslot;
  float between = 0.f;
  for (int k = 0; k < nslot; ++k) {
    const float d = part[(m * nslot + k) * 2] * (1.f / 64.f) - mean;
    between = __builtin_fmaf(d, d, between);
  }
  const float var = (m2 + 64.f * between) / (float)H;
  const float r = rsqrtf(var + eps);
  mu[m] = mean;    // separate arrays: float4 loads of 4 consecutive rows (transposed V^T epilogue)
  rstd[m] = r;
  mr[m] = make_float2(mean, r);  // interleaved: one load per row where a lane owns a row
}
// x_cls[psg][:] = LayerNorm_in(P[psg * S][:]) for the [CLS] row of every passage: chunk-major un-normalised stream -> compact
// row-major rows (the input of the last layer's row-wise tail).  beta may be NULL (0).
template <typename T>
__global__ void cls_rows_kernel(const T* __restrict__ P, const float2* __restrict__ mr, const float* __restrict__ gamma,
                                const float* __restrict__ beta, int64_t n_psg, int S, int H, T* __restrict__ out) {
  const int64_t psg = blockIdx.x;
  if (psg >= n_psg) return;
  const int64_t tok = psg * S;
  const float2 st = mr[tok];
  for (int i = threadIdx.x; i < H; i += blockDim.x)
    out[psg * H + i] = (T)(((float)P[cm_offset(tok, i, H)] - st.x) * st.y * gamma[i] + (beta ? beta[i] : 0.f));
}
__global__ void neutral_stats_kernel(int64_t M, float* __restrict__ mu, float* __restrict__ rstd, float2* __restrict__ mr) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (int64_t)gridDim.x * blockDim.x) {
    mu[i] = 0.f; rstd[i] = 1.f; mr[i] = make_float2(0.f, 1.f);
  }
}

// profiling hook (capamd_debug_ffn1_timing): HIP events around the FFN1 launches of the timed forward passes.  Exists only in the
// -DCAPAMD_PROFILING build (libcapreolus_amd_prof.so, bench.py / scripts): the product library carries no mutable global state.
#ifdef CAPAMD_PROFILING
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
#else
struct Ffn1Timing {
  static void begin(hipStream_t) {}
  static void end(hipStream_t, int64_t) {}
};
#endif

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
  // fused LayerNorm: row-statistic partials of the GEMM that wrote a pre-LayerNorm sum, and (mu, rstd) of the tensors in xb / pre
  float* part;    // [M][H/64][2]
  float* mu_x; float* rstd_x; float* mu_p; float* rstd_p;  // [M] each
  float2* mr_x; float2* mr_p;                              // [M] the same, interleaved
};

size_t ws_bytes_for(int H, int F, int S, int64_t n_psg_mb, int64_t n_psg_total) {
  const int64_t M = n_psg_mb * S;
  size_t b = 0;
  auto add = [&](size_t x) { b += (x + 255) & ~(size_t)255; };
  add((size_t)M * H * 2); add((size_t)M * H * 2); add((size_t)M * H * 2); add((size_t)M * H * 2);
  add((size_t)M * H * 2); add((size_t)M * H * 2); add((size_t)M * F * 2); add((size_t)n_psg_total * 4); add(256);
  add((size_t)M * (H / 64) * 8); add((size_t)M * 4); add((size_t)M * 4); add((size_t)M * 4); add((size_t)M * 4);
  add((size_t)M * 8); add((size_t)M * 8);
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
  w.part = (float*)take((size_t)M * (H / 64) * 8);
  w.mu_x = (float*)take((size_t)M * 4); w.rstd_x = (float*)take((size_t)M * 4);
  w.mu_p = (float*)take((size_t)M * 4); w.rstd_p = (float*)take((size_t)M * 4);
  w.mr_x = (float2*)take((size_t)M * 8); w.mr_p = (float2*)take((size_t)M * 8);
  return w;
}

// Passages per micro-batch of a call over NP passages under the caller's cap (both in passages of S tokens; the loop in encode_passages
// runs full micro-batches and one remainder).  Two candidates: EQUAL parts (4000 passages under a cap of 256 -> 16 x 250) and FULL parts
// plus a tail (15 x 256 + 160).  The persistent GEMM kernels hand out whole tiles to num_cus() workgroups (2 num_cus() for the 128-row
// tile), so a launch takes ceil(tiles / slots) rounds whatever the last round holds: 250 row tiles x 3 column tiles = 750 tiles on 256
// CUs is 3 rounds for 2.93 rounds of work, in every GEMM of every layer, while 256 row tiles fill every round exactly and only the
// tail pays.  The cheaper plan by that count wins (rounds weighted by the K of the GEMM); the workspace is sized for the cap either way.
int64_t plan_microbatch(int64_t NP, int64_t cap, int S, int H, int F) {
  const int64_t q = tile_passages(S);
  auto up = [&](int64_t x) { return (x + q - 1) / q * q; };
  const int64_t n_mb = (NP + cap - 1) / cap;
  const int64_t equal = up((NP + n_mb - 1) / n_mb), full = up(cap);
  if (NP <= cap || equal == full) return equal;
  static const int forced = [] { const char* e = getenv("CAPAMD_BERT_MB_PLAN"); return e ? (e[0] == 'e' ? 1 : e[0] == 'f' ? 2 : 0) : 0; }();   // A/B runs: equal / full
  if (forced) return forced == 1 ? equal : full;
  const int64_t cus = num_cus();
  auto rounds = [&](int64_t rows, int64_t cols, int64_t slots) { return ((rows / 256) * (cols / 256) + slots - 1) / slots; };
  auto cost_mb = [&](int64_t np) {     // one layer of one micro-batch, in (rounds x K) units
    const int64_t rows = (up(np) * S + 255) / 256 * 256;
    return rounds(rows, 3 * H, cus) * H + rounds(rows, H, cus) * (H + F) + rounds(rows, F, cus) * H;
  };
  auto cost = [&](int64_t mbp) {
    const int64_t whole = NP / mbp, rest = NP - whole * mbp;
    return whole * cost_mb(mbp) + (rest ? cost_mb(rest) : 0);
  };
  return cost(full) < cost(equal) ? full : equal;
}

template <typename T>
hipError_t encode_passages(const int64_t* ids, const int64_t* mask, const int64_t* seg, int64_t NP, int64_t mb, int S,
                           const capamd_bert_model* m, const Workspace& w, int* status, hipStream_t s, const CedrTap* tap = nullptr) {
  const int H = m->hidden, F = m->ffn;
  const T* blob = (const T*)m->blob;
  const float eps = m->ln_eps > 0.f ? m->ln_eps : kLnEps;
  hipError_t e = hipSuccess;

  for (int64_t p0 = 0; p0 < NP && e == hipSuccess; p0 += mb) {
    const int64_t np = (NP - p0 < mb) ? NP - p0 : mb;
    // The GEMM kernels schedule whole row tiles only: a ragged micro-batch (np * S not a multiple of 256 - an odd passage count
    // at S = 32, 96, 160, 224, ...) runs on np_pad passages; the extra rows are zero embeddings / zero context (rows are
    // independent in every GEMM and LayerNorm, attention and the head run on the np real passages), so nothing they hold
    // reaches a real row.  The workspace is sized for whole tiles (capamd_bert_workspace_bytes).
    const int64_t tq = tile_passages(S), np_pad = (np + tq - 1) / tq * tq;
    const int64_t M = np_pad * S, M_real = np * S;
    const int64_t* ids_mb = ids + p0 * S;
    const int64_t* mask_mb = mask + p0 * S;
    const int64_t* seg_mb = seg + p0 * S;
    // LayerNorm folded into the GEMMs: every encoder GEMM of this microbatch must be a ping-pong shape
    // (a CEDR-KNRM call reads every layer's normalised output: its tap applies the LayerNorm to the operand fragments it loads from
    // the un-normalised stream, cedr_tap.h; CAPAMD_CEDR_FUSED=0 runs it on the path that materialises them instead)
    const bool fused = (!tap || (cedr_fused_enabled() && S % 32 == 0 && cedr_pool_cm_smem(S, H, tap->A) <= 160 * 1024)) && fused_ln_enabled() &&
                       fused_capable(H, F) && pingpong_shape(M, 3 * H, H) && pingpong_shape(M, H, H) && pingpong_shape(M, F, H) &&
                       pingpong_shape(M, H, F);
    if (fused) {  // chunk-major stream: whole 32-token groups through LDS (S % 32 == 0, so M_real % 32 == 0)
      const size_t lds = (size_t)32 * (H / 8 + 1) * 16;
      if (lds > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(embed_ln_cm_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL((embed_ln_cm_kernel<T>), dim3((unsigned)(M_real / 32)), dim3(512), lds, s, ids_mb, seg_mb, m->word_emb, m->pos_emb,
                         m->type_emb, m->vocab, m->type_vocab, S, m->emb_ln_g, m->emb_ln_b, H, (T*)w.xb, status, eps, m->pos_pad_id, m->max_pos);
    } else
      hipLaunchKernelGGL((ln_kernel<0, T>), dim3((unsigned)((M_real + 3) / 4)), dim3(256), 0, s, (const T*)nullptr, ids_mb, seg_mb, m->word_emb, m->pos_emb,
                         m->type_emb, m->vocab, m->type_vocab, S, m->emb_ln_g, m->emb_ln_b, M_real, H, (T*)w.xb, status, 0, eps, m->pos_pad_id, m->max_pos);
    if (M > M_real) {   // (S % 32 == 0: the pad rows are whole 32-row groups, contiguous in the row-major and the chunk-major layout alike)
      (void)hipMemsetAsync(w.xb + M_real * H, 0, (size_t)(M - M_real) * H * 2, s);
      (void)hipMemsetAsync(w.ctx + M_real * H, 0, (size_t)(M - M_real) * H * 2, s);
    }
    if (tap) cedr_tap_layer<T>(*tap, 0, (const T*)w.xb, mask_mb, seg_mb, p0, np, S, H, s, fused);   // (normalised in either layout)
    if (fused) {
      // Activation stream: xb and pre hold UN-normalised pre-LayerNorm sums in the chunk-major layout, (mu, rstd) of
      // their rows next to them; no LayerNorm pass exists.  Consumers fold the normalisation into their epilogue
      // (weights pre-scaled by gamma when packed), producers rebuild the normalised residual on the fly and emit the row
      // statistics of what they write.  Layer 0 reads the (normalised) embedding output: neutral statistics.
      hipLaunchKernelGGL(neutral_stats_kernel, dim3(256), dim3(256), 0, s, M, w.mu_x, w.rstd_x, w.mr_x);
      const float *last_g = nullptr, *last_b = nullptr;
      bool cls_done = false;
      const T* cls_x = nullptr;
      for (int l = 0; l < m->layers && e == hipSuccess; ++l) {
        const T* wl = blob + (int64_t)l * layer_blob_elems(H, F);
        const float* fl = m->layer_f32 + (int64_t)l * layer_f32_floats(H, F);
        const T *wqkv = wl, *wo = wl + (int64_t)3 * H * H, *w1 = wl + (int64_t)4 * H * H, *w2 = w1 + (int64_t)F * H;
        const T *wqkv_s = w2 + (int64_t)H * F, *w1_s = wqkv_s + (int64_t)3 * H * H;
        // both operands chunk-major -> the 4-wave ring kernel (bert_gemm_ring.h); its weights are the copies behind the row-major ones
        const bool ring = ring_enabled() && ring_shape(M, 3 * H, H) && ring_shape(M, H, H) && ring_shape(M, F, H) && ring_shape(M, H, F);
        const int64_t cmo = ring ? layer_rowmajor_elems(H, F) : 0;
        const float *bqkv = fl, *ln1g = fl + 4 * H, *ln2g = fl + 7 * H + F, *ln2b = fl + 8 * H + F;
        const float* fx = fl + 9 * H + F;
        const float *cs_qkv = fx, *c_qkv = fx + 3 * H, *cs_1 = fx + 6 * H, *c_1 = fx + 6 * H + F, *g_in = fx + 6 * H + 2 * F, *bo_f = g_in + H,
                    *b2_f = bo_f + H;
        GemmArgs g{};
        g.H = H; g.S = S; g.heads = m->heads; g.M = (int)M;
        // QKV projection of LN_in(xb): layer 0 reads the normalised embeddings with the plain weights
        g.A = w.xb; g.a_cm = 1; g.N = 3 * H; g.K = H; g.out_bf16 = w.q; g.out_k = w.k; g.out_vt = w.vt; g.out_cm = 1;  // Q, K chunk-major
        if (l == 0) { g.W = wqkv + cmo; g.bias = bqkv; }
        else { g.W = wqkv_s + cmo; g.bias = c_qkv; g.ln_cs = cs_qkv; g.ln_mu = w.mu_x; g.ln_rstd = w.rstd_x; g.ln_mr = w.mr_x; }
        g.w_cm = ring;
        g.ring_rows = gemm_pick(0).rows; g.ring_mfma32 = gemm_pick(0).mfma32; g.ring_mfma16 = gemm_pick(0).mfma16;
        e = launch_gemm<kEpiQkv, T>(g, s);
        if (e != hipSuccess) break;
        AttnArgs at{w.q, w.k, w.vt, mask_mb, w.ctx, H, m->heads, 1, ring ? 1 : 0};
        if (l == m->layers - 1 && cls_tail_enabled() && !tap) {   // (CEDR-KNRM pools every row of the last hidden state too)
          // Last layer: only the [CLS] row of every passage is read afterwards and everything after the attention is
          // row-wise - attention for that one query, then the output projection / LayerNorm / FFN / LayerNorm on n_psg
          // rows (padded to whole 256-row tiles) instead of n_psg * S, through the plain row-major kernels.
          const int64_t npad = (np + 255) / 256 * 256;
          T *ctx_c = (T*)w.ctx, *x_c = (T*)w.q, *pre_c = (T*)w.pre, *mid_c = (T*)w.mid;   // (x_c reuses the Q buffer once Q is dead)
          const float *bo = fl + 3 * H, *ln1b = fl + 5 * H, *b1 = fl + 6 * H, *b2 = fl + 6 * H + F;
          const float* beta_in = l > 0 ? (fl - layer_f32_floats(H, F)) + 8 * H + F : nullptr;
          at.ctx_cm = 0;   // (the [CLS]-row tail runs on compact row-major rows)
          (void)hipMemsetAsync(ctx_c + np * H, 0, (size_t)(npad - np) * H * 2, s);
          hipLaunchKernelGGL(cls_attention_kernel<T>, dim3((unsigned)(np * m->heads)), dim3(64), 0, s, at, S, ctx_c);
          (void)hipMemsetAsync(x_c + np * H, 0, (size_t)(npad - np) * H * 2, s);   // (inside the Q buffer: only after its last reader)
          hipLaunchKernelGGL(cls_rows_kernel<T>, dim3((unsigned)np), dim3(256), 0, s, (const T*)w.xb, (const float2*)w.mr_x, g_in, beta_in, np, S, H, x_c);
          g = GemmArgs{};
          g.M = (int)npad; g.N = H; g.K = H; g.A = ctx_c; g.W = wo; g.bias = bo; g.out_bf16 = pre_c;
          e = launch_gemm<kEpiBiasBf16, T>(g, s);
          if (e != hipSuccess) break;
          hipLaunchKernelGGL((ln_kernel<2, T>), dim3((unsigned)((npad + 3) / 4)), dim3(256), 0, s, (const T*)pre_c, nullptr, nullptr, nullptr, nullptr, nullptr,
                             0, 0, 1, ln1g, ln1b, npad, H, x_c, status, 0, eps, -1, 0);
          g.N = F; g.K = H; g.A = x_c; g.W = w1; g.bias = b1; g.out_bf16 = mid_c;
          e = launch_gemm<kEpiBiasGeluBf16, T>(g, s);
          if (e != hipSuccess) break;
          g.N = H; g.K = F; g.A = mid_c; g.W = w2; g.bias = b2; g.out_bf16 = pre_c;
          e = launch_gemm<kEpiBiasBf16, T>(g, s);
          if (e != hipSuccess) break;
          hipLaunchKernelGGL((ln_kernel<2, T>), dim3((unsigned)((npad + 3) / 4)), dim3(256), 0, s, (const T*)pre_c, nullptr, nullptr, nullptr, nullptr, nullptr,
                             0, 0, 1, ln2g, ln2b, npad, H, x_c, status, 0, eps, -1, 0);
          cls_done = true;
          cls_x = x_c;
          break;
        }
        launch_attention<T>(at, S, (unsigned)(np * m->heads), s);
        // pre = ctx Wo^T + bo + LN_in(xb)   (+ row statistics of pre)
        g = GemmArgs{};
        // (ring kernel, measured per GEMM inside the encoder at M = 64,000: the residual + statistics producers run faster on its 256-row
        // tile - 181 us against 200 on the 128-row tile and 193 on the ping-pong kernel - QKV and FFN1, with their heavier epilogues,
        // on the 128-row tile whose two workgroups per CU overlap epilogue and K loop: 248 / 341 us against 253 / 356)
        g.M = (int)M; g.N = H; g.K = H; g.A = w.ctx; g.a_cm = ring; g.W = wo + cmo; g.w_cm = ring; g.ring_rows = gemm_pick(2).rows; g.ring_mfma32 = gemm_pick(2).mfma32; g.ring_mfma16 = gemm_pick(2).mfma16;
        g.bias = bo_f; g.out_bf16 = w.pre; g.out_cm = 1;
        g.res_src = w.xb; g.res_mr = w.mr_x; g.res_gamma = g_in; g.stat_part = w.part;
        e = launch_gemm<kEpiResidStats, T>(g, s);
        if (e != hipSuccess) break;
        hipLaunchKernelGGL(ln_stats_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, w.part, H / 64, H, M, w.mu_p, w.rstd_p, w.mr_p, eps);
        // mid = gelu(LN1(pre) W1^T + b1)
        g = GemmArgs{};
        g.M = (int)M; g.N = F; g.K = H; g.A = w.pre; g.a_cm = 1; g.W = w1_s + cmo; g.w_cm = ring; g.bias = c_1; g.ln_cs = cs_1; g.ln_mu = w.mu_p; g.ln_rstd = w.rstd_p; g.ln_mr = w.mr_p;
        g.out_bf16 = w.mid; g.out_cm = 1;
        g.ring_rows = gemm_pick(1).rows; g.ring_mfma32 = gemm_pick(1).mfma32; g.ring_mfma16 = gemm_pick(1).mfma16;
        Ffn1Timing::begin(s);
        e = launch_gemm<kEpiBiasGeluBf16, T>(g, s);
        Ffn1Timing::end(s, M);
        if (e != hipSuccess) break;
        // xb = mid W2^T + b2 + LN1(pre)   (+ row statistics of xb)
        g = GemmArgs{};
        g.M = (int)M; g.N = H; g.K = F; g.A = w.mid; g.a_cm = 1; g.W = w2 + cmo; g.w_cm = ring; g.ring_rows = gemm_pick(3).rows; g.ring_mfma32 = gemm_pick(3).mfma32; g.ring_mfma16 = gemm_pick(3).mfma16;
        g.bias = b2_f; g.out_bf16 = w.xb; g.out_cm = 1;
        g.res_src = w.pre; g.res_mr = w.mr_p; g.res_gamma = ln1g; g.stat_part = w.part;
        e = launch_gemm<kEpiResidStats, T>(g, s);
        if (e != hipSuccess) break;
        hipLaunchKernelGGL(ln_stats_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, w.part, H / 64, H, M, w.mu_x, w.rstd_x, w.mr_x, eps);
        last_g = ln2g; last_b = ln2b;
        if (tap) cedr_tap_layer<T>(*tap, l + 1, (const T*)w.xb, mask_mb, seg_mb, p0, np, S, H, s, true, (const float2*)w.mr_x, ln2g, ln2b);
      }
      if (e != hipSuccess) break;
      if (tap) {   // the [CLS] rows of the last hidden state (CEDRKNRM.py:160), LayerNorm applied on the way
        hipLaunchKernelGGL(cedr_cls_rows_cm_kernel<T>, dim3((unsigned)np), dim3(256), 0, s, (const T*)w.xb, (const float2*)w.mr_x, last_g, last_b, S, H,
                           tap->cls + p0 * H);
        e = hipGetLastError();
        continue;
      }
      float* hpart = reinterpret_cast<float*>(cls_done ? w.ctx : w.pre);
      if (cls_done)   // compact, already normalised [CLS] rows: one row per passage
        hipLaunchKernelGGL(head_kernel<T>, dim3((unsigned)((np + kHeadPsg - 1) / kHeadPsg), (unsigned)(H / kHeadRows)), dim3(256), 0, s, cls_x, np, 1, H,
                           m->pooler_w, m->pooler_b, m->cls_w, hpart, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                           (const float*)nullptr);
      else
      hipLaunchKernelGGL(head_kernel<T>, dim3((unsigned)((np + kHeadPsg - 1) / kHeadPsg), (unsigned)(H / kHeadRows)), dim3(256), 0, s, (const T*)w.xb, np, S, H,
                         m->pooler_w, m->pooler_b, m->cls_w, hpart, (const float*)w.mu_x, (const float*)w.rstd_x, last_g, last_b);
      hipLaunchKernelGGL(head_reduce_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, s, hpart, np, H / kHeadRows, m->cls_b, w.logits + p0);
      e = hipGetLastError();
      continue;
    }
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
      AttnArgs at{w.q, w.k, w.vt, mask_mb, w.ctx, H, m->heads, 0, 0};
      const unsigned nblk = (unsigned)(np * m->heads);
      launch_attention<T>(at, S, nblk, s);
      // attention output projection; the residual is added in fp32 inside the LayerNorm pass (one rounding, and the
      // GEMM keeps its cheap 16-bit epilogue)
      g.A = w.ctx; g.W = wo; g.bias = bo; g.N = H; g.K = H; g.out_bf16 = w.pre;
      e = launch_gemm<kEpiBiasBf16, T>(g, s);
      if (e != hipSuccess) break;
      hipLaunchKernelGGL((ln_kernel<2, T>), dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, (const T*)w.pre, nullptr, nullptr, nullptr, nullptr, nullptr,
                         0, 0, S, ln1g, ln1b, M, H, (T*)w.xb, status, 0, eps, -1, 0);
      // feed-forward: 768 -> 3072 (GELU) -> 768, residual + LayerNorm
      const bool cm = chunk_major_enabled() && pingpong_shape(M, F, H) && pingpong_shape(M, H, F);  // mid in the chunk-major layout (bert_gemm.h)
      g.A = w.xb; g.W = w1; g.bias = b1; g.N = F; g.K = H; g.out_bf16 = w.mid; g.out_cm = cm;
      Ffn1Timing::begin(s);
      e = launch_gemm<kEpiBiasGeluBf16, T>(g, s);
      Ffn1Timing::end(s, M);
      if (e != hipSuccess) break;
      g.A = w.mid; g.W = w2; g.bias = b2; g.N = H; g.K = F; g.out_bf16 = w.pre; g.out_cm = 0; g.a_cm = cm;
      e = launch_gemm<kEpiBiasBf16, T>(g, s);
      g.a_cm = 0;
      if (e != hipSuccess) break;
      hipLaunchKernelGGL((ln_kernel<2, T>), dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, (const T*)w.pre, nullptr, nullptr, nullptr, nullptr, nullptr,
                         0, 0, S, ln2g, ln2b, M, H, (T*)w.xb, status, 0, eps, -1, 0);
      if (tap) cedr_tap_layer<T>(*tap, l + 1, (const T*)w.xb, mask_mb, seg_mb, p0, np, S, H, s);
    }
    if (e != hipSuccess) break;
    if (tap) {   // CEDR-KNRM reads the [CLS] rows of the last hidden state itself (CEDRKNRM.py:160); no pooler / classifier
      hipLaunchKernelGGL(cedr_cls_rows_kernel<T>, dim3((unsigned)np), dim3(256), 0, s, (const T*)w.xb, S, H, tap->cls + p0 * H);
      e = hipGetLastError();
      continue;
    }
    // (the partial sums reuse the pre-LayerNorm buffer, which is dead after the last layer)
    float* hpart = reinterpret_cast<float*>(w.pre);
    hipLaunchKernelGGL(head_kernel<T>, dim3((unsigned)((np + kHeadPsg - 1) / kHeadPsg), (unsigned)(H / kHeadRows)), dim3(256), 0, s, (const T*)w.xb, np, S, H,
                       m->pooler_w, m->pooler_b, m->cls_w, hpart, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr);
    hipLaunchKernelGGL(head_reduce_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, s, hpart, np, H / kHeadRows, m->cls_b, w.logits + p0);
    e = hipGetLastError();
  }
  return e;
}

template <typename T>
void pack_folded(const float* const* t, int64_t H, int64_t F, uint16_t* wb, float* fb, hipStream_t s) {
  // gamma_in / beta_in: the LayerNorm whose output feeds this layer (t[16], t[17]; NULL for layer 0 = identity)
  float* fx = fb + 9 * H + F;
  float *cs_qkv = fx, *c_qkv = fx + 3 * H, *cs_1 = fx + 6 * H, *c_1 = fx + 6 * H + F, *g_in = fx + 6 * H + 2 * F, *bo_f = g_in + H, *b2_f = bo_f + H;
  float* zeros = b2_f;  // scratch for beta_in = 0 until b2_f is written at the end
  if (t[16]) {
    hipLaunchKernelGGL(copy_f32_kernel, dim3(8), dim3(256), 0, s, t[16], g_in, H);
    hipLaunchKernelGGL(vec_add_kernel, dim3(8), dim3(256), 0, s, t[7], t[17], bo_f, H);
  } else {
    hipLaunchKernelGGL(vec_fill_kernel, dim3(8), dim3(256), 0, s, g_in, 1.f, H);
    hipLaunchKernelGGL(copy_f32_kernel, dim3(8), dim3(256), 0, s, t[7], bo_f, H);
    hipLaunchKernelGGL(vec_fill_kernel, dim3(8), dim3(256), 0, s, zeros, 0.f, H);
  }
  const float* beta_in = t[17] ? t[17] : zeros;
  T* wqkv_s = (T*)(wb + 4 * H * H + 2 * F * H);
  T* w1_s = wqkv_s + 3 * H * H;
  for (int part = 0; part < 3; ++part) {  // q, k, v
    hipLaunchKernelGGL(fold_rows_kernel<T>, dim3((unsigned)((H + 3) / 4)), dim3(256), 0, s, t[2 * part], (const float*)g_in, t[2 * part + 1], beta_in,
                       wqkv_s + part * H * H, cs_qkv + part * H, c_qkv + part * H, (int)H, (int)H);
  }
  hipLaunchKernelGGL(fold_rows_kernel<T>, dim3((unsigned)((F + 3) / 4)), dim3(256), 0, s, t[10], t[8], t[11], t[9], w1_s, cs_1, c_1, (int)F, (int)H);
  hipLaunchKernelGGL(vec_add_kernel, dim3(8), dim3(256), 0, s, t[13], t[9], b2_f, H);  // b2 + ln1.beta (last: b2_f doubled as the zero vector)
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

int capamd_bert_pack_layer(const capamd_bert_model* m, int layer, const float* const* t /* 18 device pointers, host array */,
                           void* blob, float* layer_f32, void* stream) {
  if (!dims_ok(m) || !t || !blob || !layer_f32 || layer < 0 || layer >= m->layers) return CAPAMD_ERR_ARG;
  for (int i = 0; i < 16; ++i)
    if (!t[i]) return CAPAMD_ERR_ARG;
  if ((t[16] == nullptr) != (t[17] == nullptr)) return CAPAMD_ERR_ARG;
  const int64_t H = m->hidden, F = m->ffn;
  hipStream_t s = (hipStream_t)stream;
  (void)hipGetLastError();
  uint16_t* wb = (uint16_t*)blob + (int64_t)layer * layer_blob_elems(H, F);
  float* fb = layer_f32 + (int64_t)layer * layer_f32_floats(H, F);
  // order of t: q.w q.b k.w k.b v.w v.b o.w o.b ln1.g ln1.b ffn1.w ffn1.b ffn2.w ffn2.b ln2.g ln2.b  [in.g in.b]
  struct { int src; int64_t off, n; } wcp[] = {{0, 0, H * H}, {2, H * H, H * H}, {4, 2 * H * H, H * H}, {6, 3 * H * H, H * H},
                                               {10, 4 * H * H, F * H}, {12, 4 * H * H + F * H, H * F}};
  for (auto& c : wcp) {
    if (m->compute_dtype == 1) hipLaunchKernelGGL(cvt_bf16_kernel<_Float16>, dim3(512), dim3(256), 0, s, t[c.src], (_Float16*)(wb + c.off), c.n);
    else hipLaunchKernelGGL(cvt_bf16_kernel<__bf16>, dim3(512), dim3(256), 0, s, t[c.src], (__bf16*)(wb + c.off), c.n);
  }
  struct { int src; int64_t off, n; } fcp[] = {{1, 0, H}, {3, H, H}, {5, 2 * H, H}, {7, 3 * H, H}, {8, 4 * H, H}, {9, 5 * H, H},
                                               {11, 6 * H, F}, {13, 6 * H + F, H}, {14, 7 * H + F, H}, {15, 8 * H + F, H}};
  for (auto& c : fcp) hipLaunchKernelGGL(copy_f32_kernel, dim3(8), dim3(256), 0, s, t[c.src], fb + c.off, c.n);
  if (fused_capable((int)H, (int)F)) {
    if (m->compute_dtype == 1) pack_folded<_Float16>(t, H, F, wb, fb, s);
    else pack_folded<__bf16>(t, H, F, wb, fb, s);
    // chunk-major copies of the six matrices (same order, after the row-major ones): the ring kernel's weight operand
    struct { int64_t off, rows, K; } mats[] = {{0, 3 * H, H}, {3 * H * H, H, H}, {4 * H * H, F, H}, {4 * H * H + F * H, H, F},
                                               {4 * H * H + 2 * F * H, 3 * H, H}, {7 * H * H + 2 * F * H, F, H}};
    uint16_t* cmb = wb + layer_rowmajor_elems((int)H, (int)F);
    for (auto& mt : mats)   // (a 16-bit copy: the element type does not matter)
      hipLaunchKernelGGL(to_chunk_major_kernel<uint16_t>, dim3(512), dim3(256), 0, s, (const uint16_t*)(wb + mt.off), cmb + mt.off, (int)mt.rows, (int)mt.K);
  }
  return hipGetLastError() == hipSuccess ? CAPAMD_OK : CAPAMD_ERR_LAUNCH;
}

int64_t capamd_bert_workspace_bytes(const capamd_bert_model* m, int S, int64_t passages_per_microbatch, int64_t total_passages) {
  if (!dims_ok(m) || S < 1 || passages_per_microbatch < 1 || total_passages < 1) return -1;
  const int64_t q = tile_passages(S);   // (micro-batches are whole 256-row tiles, see capamd_bert_maxp_forward)
  return (int64_t)ws_bytes_for(m->hidden, m->ffn, S, (passages_per_microbatch + q - 1) / q * q, total_passages);
}

int capamd_maxp_pool(const float* passage_logits, const int64_t* mask, const int64_t* seg, int B, int P, int S, int aggregation,
                     float* out, int* count_scratch, void* stream) {
  if (B == 0) return CAPAMD_OK;
  if (!passage_logits || !mask || !seg || !out || !count_scratch || B < 0 || P < 1 || S < 1 || aggregation < 0 || aggregation > 3)
    return CAPAMD_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  (void)hipGetLastError();
  if (aggregation == 3) (void)hipMemsetAsync(count_scratch, 0, 4, s);
  hipLaunchKernelGGL(pool_kernel, dim3(B), dim3(64), 0, s, passage_logits, mask, seg, P, S, aggregation, out, count_scratch);
  if (aggregation == 3) hipLaunchKernelGGL(avg_div_kernel, dim3((B + 255) / 256), dim3(256), 0, s, out, B, count_scratch);
  return hipGetLastError() == hipSuccess ? CAPAMD_OK : CAPAMD_ERR_LAUNCH;
}

int capamd_bert_maxp_forward(const int64_t* ids, const int64_t* mask, const int64_t* seg, int B, int P, int S,
                             const capamd_bert_model* m, int aggregation, int64_t passages_per_microbatch, void* workspace,
                             int64_t workspace_bytes, float* out, float* passage_logits_out, int* status, void* stream) {
  if (B == 0) return CAPAMD_OK;
  if (!ids || !mask || !seg || !dims_ok(m) || !workspace || !out || !status || B < 0 || P < 1) return CAPAMD_ERR_ARG;
  if (!supported_length(S) || S > m->max_pos || aggregation < 0 || aggregation > 3) return CAPAMD_ERR_ARG;
  if (!m->word_emb || !m->pos_emb || !m->type_emb || !m->emb_ln_g || !m->emb_ln_b || !m->pooler_w || !m->pooler_b || !m->cls_w ||
      !m->cls_b || !m->blob || !m->layer_f32)
    return CAPAMD_ERR_ARG;
  const int H = m->hidden, F = m->ffn;
  const int64_t NP = (int64_t)B * P;
  if (passages_per_microbatch < 1) return CAPAMD_ERR_ARG;
  const int64_t mb = plan_microbatch(NP, passages_per_microbatch, S, H, F);
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

int capamd_cedr_passage_features(const int64_t* ids, const int64_t* mask, const int64_t* seg, int B, int P, int S,
                                 const capamd_bert_model* m, int64_t passages_per_microbatch, void* workspace, int64_t workspace_bytes,
                                 int maxqlen, const float* query_mask0, const int* simmat_layers /* host */, int n_layers, const float* mu,
                                 const float* sigma, int K, float* passage_kernel_sums, float* cls_rows, int* status, void* stream) {
  if (B == 0) return CAPAMD_OK;
  if (!ids || !mask || !seg || !dims_ok(m) || !workspace || !status || B < 0 || P < 1 || !cls_rows) return CAPAMD_ERR_ARG;
  if (!supported_length(S) || S > m->max_pos || passages_per_microbatch < 1) return CAPAMD_ERR_ARG;
  if (!m->word_emb || !m->pos_emb || !m->type_emb || !m->emb_ln_g || !m->emb_ln_b || !m->blob || !m->layer_f32) return CAPAMD_ERR_ARG;
  if (n_layers < 0 || n_layers > m->layers + 1 || maxqlen < 1 || maxqlen + 1 > kCedrMaxA) return CAPAMD_ERR_ARG;
  if (n_layers > 0 && (!simmat_layers || !mu || !sigma || !passage_kernel_sums || !query_mask0 || K < 1 || K > kCedrMaxK)) return CAPAMD_ERR_ARG;
  for (int i = 0; i < n_layers; ++i)
    if (simmat_layers[i] < 0 || simmat_layers[i] > m->layers) return CAPAMD_ERR_ARG;
  const int H = m->hidden, F = m->ffn;
  const int64_t NP = (int64_t)B * P;
  const int64_t mb = plan_microbatch(NP, passages_per_microbatch, S, H, F);
  if ((int64_t)ws_bytes_for(H, F, S, mb, NP) > workspace_bytes) return CAPAMD_ERR_WORKSPACE;
  if ((reinterpret_cast<uintptr_t>(workspace) & 255) != 0) return CAPAMD_ERR_ALIGN;
  if (cedr_pool_smem(S, maxqlen + 1) > 160 * 1024) return CAPAMD_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  (void)hipGetLastError();
  Workspace w = carve((char*)workspace, H, F, S, mb, NP);
  CedrTap tap{maxqlen + 1, K, n_layers, simmat_layers, mu, sigma, query_mask0, passage_kernel_sums, cls_rows, NP};
  const hipError_t e = (m->compute_dtype == 1) ? encode_passages<_Float16>(ids, mask, seg, NP, mb, S, m, w, status, s, &tap)
                                                : encode_passages<__bf16>(ids, mask, seg, NP, mb, S, m, w, status, s, &tap);
  if (e != hipSuccess) return CAPAMD_ERR_LAUNCH;
  return hipGetLastError() == hipSuccess ? CAPAMD_OK : CAPAMD_ERR_LAUNCH;
}

/* building blocks, exported for unit tests and for callers that want the encoder pieces */
#ifdef CAPAMD_PROFILING
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
#else
static constexpr unsigned long long* g_gemm_dbg = nullptr;
#endif

int capamd_bert_gemm(const void* A, const void* W, const float* bias, int M, int N, int K, int epilogue, const void* resid,
                     void* out, int dtype, void* stream) {
  if (!A || !W || !bias || !out || M < 64 || N < 64 || K < 64 || M % 64 || N % 64 || K % 64) return CAPAMD_ERR_ARG;
  (void)hipGetLastError();
  GemmArgs g{};
  g.A = A; g.W = W; g.bias = bias; g.M = M; g.N = N; g.K = K;
  g.dbg = g_gemm_dbg;
  // layout bits of `epilogue`: the chunk-major activation layout exists only on the ping-pong kernel's shapes
  g.a_cm = (epilogue & CAPAMD_GEMM_A_CHUNK_MAJOR) ? 1 : 0;
  g.out_cm = (epilogue & CAPAMD_GEMM_OUT_CHUNK_MAJOR) ? 1 : 0;
  g.w_cm = (epilogue & CAPAMD_GEMM_W_CHUNK_MAJOR) ? 1 : 0;
  g.ring_rows = (epilogue & CAPAMD_GEMM_RING_256) ? 256 : 0;
  g.ring_mfma32 = (epilogue & CAPAMD_GEMM_RING_MFMA32) ? 1 : 0;
  g.ring_mfma16 = (epilogue & CAPAMD_GEMM_RING_MFMA16) ? 1 : 0;
  epilogue &= 0xff;
  if ((g.a_cm || g.out_cm) && (!pingpong_shape(M, N, K) || epilogue == kEpiBiasResidBf16)) return CAPAMD_ERR_ARG;
  if (g.w_cm && (!g.a_cm || !ring_shape(M, N, K) || epilogue == kEpiBiasResidBf16)) return CAPAMD_ERR_ARG;
  if (dtype == 1) return gemm_dispatch<_Float16>(g, epilogue, resid, out, (hipStream_t)stream);
  if (dtype == 0) return gemm_dispatch<__bf16>(g, epilogue, resid, out, (hipStream_t)stream);
  return CAPAMD_ERR_ARG;
}

int capamd_bert_gemm_ln(const void* A, const void* W, const float* bias, int M, int N, int K, int epilogue, const float* ln_mu,
                        const float* ln_rstd, const float* ln_mr, const float* ln_cs, const void* res_src, const float* res_mr,
                        const float* res_gamma, float* stat_part, void* out, int dtype, void* stream) {
  if (!A || !W || !bias || !out || !pingpong_shape(M, N, K) || (dtype != 0 && dtype != 1)) return CAPAMD_ERR_ARG;
  (void)hipGetLastError();
  GemmArgs g{};
  g.A = A; g.W = W; g.bias = bias; g.M = M; g.N = N; g.K = K; g.out_bf16 = out;
  g.dbg = g_gemm_dbg;
  g.a_cm = (epilogue & CAPAMD_GEMM_A_CHUNK_MAJOR) ? 1 : 0;
  g.out_cm = (epilogue & CAPAMD_GEMM_OUT_CHUNK_MAJOR) ? 1 : 0;
  g.w_cm = (epilogue & CAPAMD_GEMM_W_CHUNK_MAJOR) ? 1 : 0;
  g.ring_rows = (epilogue & CAPAMD_GEMM_RING_256) ? 256 : 0;
  g.ring_mfma32 = (epilogue & CAPAMD_GEMM_RING_MFMA32) ? 1 : 0;
  g.ring_mfma16 = (epilogue & CAPAMD_GEMM_RING_MFMA16) ? 1 : 0;
  epilogue &= 0xff;
  if (g.w_cm && (!g.a_cm || !ring_shape(M, N, K))) return CAPAMD_ERR_ARG;
  if (ln_mu) {
    if (!ln_rstd || !ln_mr || !ln_cs) return CAPAMD_ERR_ARG;
    g.ln_mu = ln_mu; g.ln_rstd = ln_rstd; g.ln_mr = (const float2*)ln_mr; g.ln_cs = ln_cs;
  }
  hipStream_t s = (hipStream_t)stream;
  hipError_t e;
  if (epilogue == kEpiResidStats) {
    if (!res_src || !res_mr || !res_gamma || !stat_part || !g.out_cm) return CAPAMD_ERR_ARG;
    g.res_src = res_src; g.res_mr = (const float2*)res_mr; g.res_gamma = res_gamma; g.stat_part = stat_part;
    e = dtype == 1 ? launch_gemm<kEpiResidStats, _Float16>(g, s) : launch_gemm<kEpiResidStats, __bf16>(g, s);
  } else if (epilogue == kEpiBiasGeluBf16) {
    e = dtype == 1 ? launch_gemm<kEpiBiasGeluBf16, _Float16>(g, s) : launch_gemm<kEpiBiasGeluBf16, __bf16>(g, s);
  } else if (epilogue == kEpiBiasBf16) {
    e = dtype == 1 ? launch_gemm<kEpiBiasBf16, _Float16>(g, s) : launch_gemm<kEpiBiasBf16, __bf16>(g, s);
  } else {
    return CAPAMD_ERR_ARG;
  }
  return e == hipSuccess ? CAPAMD_OK : CAPAMD_ERR_LAUNCH;
}

int capamd_bert_qkv_attention(const void* x, const void* wqkv, const float* bqkv, const int64_t* mask, int n_passages, int S,
                              int hidden, int heads, void* q, void* k, void* vt, void* ctx, int dtype, void* stream) {
  if (!x || !wqkv || !bqkv || !mask || !q || !k || !vt || !ctx || n_passages < 1 || heads * 64 != hidden) return CAPAMD_ERR_ARG;
  if (!supported_length(S)) return CAPAMD_ERR_ARG;
  (void)hipGetLastError();
  hipStream_t s = (hipStream_t)stream;
  GemmArgs g{};
  g.H = hidden; g.S = S; g.heads = heads;
  g.A = x; g.W = wqkv; g.bias = bqkv; g.M = n_passages * S; g.N = 3 * hidden; g.K = hidden;
  g.out_bf16 = q; g.out_k = k; g.out_vt = vt;
  if (dtype != 0 && dtype != 1) return CAPAMD_ERR_ARG;
  const hipError_t e = dtype == 1 ? launch_gemm<kEpiQkv, _Float16>(g, s) : launch_gemm<kEpiQkv, __bf16>(g, s);
  if (e != hipSuccess) return CAPAMD_ERR_LAUNCH;
  AttnArgs at{q, k, vt, mask, ctx, hidden, heads, 0, 0};
  const unsigned nblk = (unsigned)(n_passages * heads);
  if (dtype == 1) launch_attention<_Float16>(at, S, nblk, s);
  else launch_attention<__bf16>(at, S, nblk, s);
  return hipGetLastError() == hipSuccess ? CAPAMD_OK : CAPAMD_ERR_LAUNCH;
}

}  // extern "C"
