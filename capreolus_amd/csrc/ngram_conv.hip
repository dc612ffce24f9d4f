// ConvKNRM's trainable n-gram convolutions, forward and backward, as fp32-exact matrix-pipe GEMMs for gfx950.
//
// Reference: ConvKNRM_class.forward, capreolus/reranker/ConvKNRM.py:42-51 - for every n-gram size g = 1..G
//     rep_g = Conv1d(D -> F, kernel g)(ConstantPad1d((0, g - 1), 0)(embeddings(ids).permute(0, 2, 1))).permute(0, 2, 1)
// on the query and on the document, under the reference trainer's loss.backward() (trainer/pytorch.py:96-107).  The embedding table is
// frozen (ConvKNRM.py:17), so the step needs the convolutions' outputs and the gradients of their weights and biases, nothing else.
//
// A Conv1d over gathered embedding rows is a sum of g matrix products, one per tap c, whose left operand is the SAME gathered matrix
// shifted by c positions:
//     rep_g[m][f] = b_g[f] + sum_{c < g} sum_d E[ids[m + c]][d] W_g[f][d][c]           (zero rows beyond the sequence end: the ConstantPad1d)
//     dW_g[f][d][c] = sum_m E[ids[m + c]][d] dRep_g[m][f]        db_g[f] = sum_m dRep_g[m][f]
// so nothing is materialised: no [B, D, L] embedding tensor, no permutes, no padded copies, no im2col - the kernels gather the table's rows.
// The backward runs on v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: bit for bit a k-ordered fmaf chain, 157 TFLOP/s peak = the fp32
// vector peak, but one operand register per lane and the VALU left free); the forward on v_mfma_f32_32x32x16_bf16 with both operands
// split EXACTLY into three bf16 values each (six products per fp32 product, at sixteen times the fp32 MFMA rate: ngram_forward3_kernel;
// the fp32 form, ngram_forward_kernel, is kept behind -DCAPAMD_NGRAM_FP32=1).  The same split in the backward measured no faster - its
// tiles have to be transposed at the LDS stage, and that costs what the MFMAs save (profiles/r04/ngram_conv.txt).  The bias rides along as
// column D of the left operand (a constant 1 in tap 0) and row D of the weights, forward and backward.
//
//   forward:  persistent workgroups over a device queue of (128-position tile, n-gram size, 128-filter panel) units, heaviest n-gram size
//             first; the weights of a unit's taps as three bf16 planes [f][d] (ngram_pack3_kernel splits the Conv1d layout [F][D][g] once
//             per call), K loop over (tap, 32 embedding dimensions), both operands staged in LDS - gathered rows by float4, split at the
//             stage, fragments by 16-byte reads (pitch 80 B: conflict-free).  Tiles without a single real token (the padding behind a
//             training document) are written as zeros and skipped: the kernel pooling behind the convolutions masks pad positions.
//   backward: C[d][f] = sum_m X[m + c][d] dRep[m][f] - both operands are position-major, which is what the 32x32x2 MFMA wants of a "TN"
//             product (lane = column, the two k of an instruction = two consecutive positions): no transposes.  One workgroup per
//             (tap panel, slice of the positions), wave w owns filters 32 w .. 32 w + 31 and all of D; the slices' partial panels are
//             summed in a FIXED order by ngram_reduce_kernel, which also writes the Conv1d layout: deterministic, no atomics.
//             Blocks of 32 positions without a real token are skipped (their dRep rows are zero: masked positions).
#include "capreolus_amd.h"
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef CAPAMD_NGRAM_FP32
#define CAPAMD_NGRAM_FP32 0        // 1: the forward K loop on v_mfma_f32_32x32x2_f32 too (the first round-4 version) instead of bf16 x 3
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kNcMaxG = 4;             // n-gram sizes 1..G
constexpr int kNcMaxParts = kNcMaxG * (kNcMaxG + 1) / 2;
constexpr int kNcMaxDp = 320;          // padded embedding width (D + 1 rounded up to 32): D <= 319
constexpr int kFwdRows = 64;           // positions per forward tile
constexpr int kPA = 33;                // LDS pitch of the gathered-row tile (floats): column reads conflict-free
constexpr int kPB = 160;               // LDS pitch of a 128-wide panel: the two k rows of an MFMA land 32 banks apart
constexpr int kBwdBlock = 32;          // positions per backward K step

struct ConvArgs {
  const int64_t* ids[2];     // segment 0: queries [N, len[0]], segment 1: documents [N, len[1]]   (pad = 0)
  int N, len[2];
  const float* emb;          // [V, D] fp32 row-major (nn.Embedding.weight)
  int64_t V;
  int D, Dp, G, F;
  const float* w[kNcMaxG];   // Conv1d weights [F][D][g]
  const float* b[kNcMaxG];   // [F]
  float* wt;                 // [parts][Dp][F]: tap panels, part(g, c) = g (g - 1) / 2 + c for n-gram size g (1-based), bias in row D of tap 0
  float* out[2];             // forward: [N, G, len, F]
  const float* dout[2];      // backward: gradients of out
  float* partial;            // backward: [S][parts][Dp][F]
  int S;
  float* dw[kNcMaxG];        // [F][D][g]
  float* db[kNcMaxG];        // [F]
  int* status;
};

__device__ __forceinline__ int part_of(int g1, int c) { return g1 * (g1 - 1) / 2 + c; }   // g1 = n-gram size, 1-based
__device__ __forceinline__ int n_parts_dev(int G) { return G * (G + 1) / 2; }

// ---- weights: Conv1d layout -> tap panels ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ngram_pack_kernel(ConvArgs a, unsigned* queue) {
  const int part = blockIdx.y;
  int g1 = 1;
  while (part_of(g1 + 1, 0) <= part) ++g1;
  const int c = part - part_of(g1, 0);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i == 0 && part == 0) *queue = 0u;          // the forward kernel's work queue
  if (i >= a.Dp * a.F) return;
  const int d = i / a.F, f = i - d * a.F;
  float v = 0.f;
  if (d < a.D) v = a.w[g1 - 1][((int64_t)f * a.D + d) * g1 + c];
  else if (d == a.D && c == 0) v = a.b[g1 - 1][f];
  a.wt[((int64_t)part * a.Dp + d) * a.F + f] = v;
}

// ---- three-way bf16 split -------------------------------------------------------------------------------------------------------------
// An fp32 value is EXACTLY hi + mid + lo with three bf16 values (8 significant bits each, truncation; bf16 has fp32's exponent range, so no
// scaling and no range to watch): x = hi + r, hi = x & 0xffff0000; r = mid + lo likewise.  A product of two fp32 values is then
// hi hi + (hi mid + mid hi) + (mid mid + hi lo + lo hi) + terms below 2^-24 of it: six bf16 MFMAs at sixteen times the fp32 MFMA rate.
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8v;
struct Split3 {
  unsigned hi, mid, lo;      // the values' upper 16 bits (the bf16 patterns), in the low half-words
};
__device__ __forceinline__ Split3 split3(float x) {
  const unsigned xb = __float_as_uint(x), hb = xb & 0xffff0000u;
  const float r = x - __uint_as_float(hb);
  const unsigned rb = __float_as_uint(r), mb = rb & 0xffff0000u;
  const float r2 = r - __uint_as_float(mb);
  return Split3{hb >> 16, mb >> 16, __float_as_uint(r2) >> 16};
}

// weights: Conv1d layout -> [plane][part][f][Dp] bf16 (d contiguous: the MFMA's B fragment is eight consecutive d of one filter)
__global__ __launch_bounds__(256) void ngram_pack3_kernel(ConvArgs a, unsigned short* wt3, unsigned* queue) {
  const int part = blockIdx.y;
  int g1 = 1;
  while (part_of(g1 + 1, 0) <= part) ++g1;
  const int c = part - part_of(g1, 0);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i == 0 && part == 0) *queue = 0u;          // the forward kernel's work queue
  if (i >= a.Dp * a.F) return;
  const int f = i / a.Dp, d = i - f * a.Dp;
  float v = 0.f;
  if (d < a.D) v = a.w[g1 - 1][((int64_t)f * a.D + d) * g1 + c];
  else if (d == a.D && c == 0) v = a.b[g1 - 1][f];
  const Split3 sp = split3(v);
  const int64_t plane = (int64_t)gridDim.y * a.F * a.Dp, at = ((int64_t)part * a.F + f) * a.Dp + d;
  wt3[at] = (unsigned short)sp.hi;
  wt3[plane + at] = (unsigned short)sp.mid;
  wt3[2 * plane + at] = (unsigned short)sp.lo;
}

// the table row behind position (segment, row m) shifted by tap c: -1 = a zero row (beyond the sequence end, or an id outside the table)
__device__ __forceinline__ int64_t tap_row(const ConvArgs& a, int seg, int64_t m, int c, bool& bad) {
  const int len = a.len[seg];
  const int64_t n = m / len;
  const int j = (int)(m - n * len) + c;
  if (n >= a.N || j >= len) return -1;
  const int64_t id = a.ids[seg][n * len + j];
  if (id < 0 || id >= a.V) {
    bad = true;
    return -1;
  }
  return id;
}

// ---- forward --------------------------------------------------------------------------------------------------------------------------
// Persistent workgroups over a queue of (64-position tile, n-gram size, 128-filter panel) units, heaviest n-gram size first: a training
// batch's documents end in padding of very different lengths, so which tiles hold work is known only on the device, and a g = 3 unit
// costs three times a g = 1 unit - a fixed grid left CUs idle for 40 % of the launch (profiles/r04/ngram_conv.txt).
__global__ __launch_bounds__(256, 3) void ngram_forward_kernel(ConvArgs a, int tiles0, int tiles, int panels, unsigned* queue) {
  __shared__ float As[kFwdRows * kPA];
  __shared__ __attribute__((aligned(16))) float Bs[32 * kPB];
  __shared__ int64_t rid[kNcMaxG][kFwdRows];
  __shared__ int64_t orow[kFwdRows];
  __shared__ int any_real;
  __shared__ unsigned unit_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned units = (unsigned)tiles * a.G * panels;
  const int ksteps = a.Dp / 32;
  const int wr = wave >> 1, wc = wave & 1;               // a wave: 32 positions x 64 filters
  for (;;) {
    __syncthreads();                                     // (the previous unit's readers of unit_s, rid, any_real are done)
    if (tid == 0) {
      unit_s = atomicAdd(queue, 1u);
      any_real = 0;
    }
    __syncthreads();
    const unsigned unit = unit_s;
    if (unit >= units) return;
    const int tile = unit % tiles, rest = unit / tiles;
    const int f0 = (rest % panels) * 128, g1 = a.G - rest / panels;          // heavy n-gram sizes first
    const int seg = tile < tiles0 ? 0 : 1;
    const int64_t m0 = (int64_t)(seg ? tile - tiles0 : tile) * kFwdRows;
    const int len = a.len[seg];
    const int64_t M = (int64_t)a.N * len;
    // one thread per position of the tile: its (sequence, offset) once - the only division of the unit -, the table rows of its taps,
    // where its output row starts
    if (tid < kFwdRows) {
      bool bad = false, real = false;
      const int64_t m = m0 + tid;
      const unsigned n = m < M ? (unsigned)m / (unsigned)len : 0u;             // (N * len < 2^31: checked by the host)
      const int j = m < M ? (int)((unsigned)m - n * (unsigned)len) : 0;
      orow[tid] = m < M ? ((int64_t)n * a.G + (g1 - 1)) * len + j : -1;
      for (int c = 0; c < g1; ++c) {
        int64_t id = -1;
        if (m < M && j + c < len) {
          id = a.ids[seg][(int64_t)n * len + j + c];
          if (id < 0 || id >= a.V) {
            bad = true;
            id = -1;
          }
        }
        rid[c][tid] = id;
        if (c == 0 && id > 0) real = true;
      }
      if (bad) atomicOr(a.status, CAPAMD_STATUS_DOC_ID_RANGE);
      if (real) any_real = 1;
    }
    __syncthreads();
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    if (any_real) {
      const int steps = g1 * ksteps;
      const float* wt = a.wt + (int64_t)part_of(g1, 0) * a.Dp * a.F;
      float4 ra[2], rb[4];
      // (every load unconditional - a clamped address, the value masked afterwards: a load under a condition becomes a branch with a
      //  wait for everything in flight behind it, and the loads of a step then go out one round trip at a time)
      auto fetch = [&](int s) {
        const int c = s / ksteps, k0 = (s - c * ksteps) * 32;
        const int d = k0 + 4 * (tid & 7);
        const int dcl = d < a.D ? d : 0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int64_t id = rid[c][(tid >> 3) + 32 * i];
          ra[i] = *reinterpret_cast<const float4*>(a.emb + (id >= 0 ? id : 0) * a.D + dcl);        // (D % 4 == 0)
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int kk = (tid >> 5) + 8 * i, f = f0 + 4 * (tid & 31);
          rb[i] = *reinterpret_cast<const float4*>(wt + ((int64_t)c * a.Dp + k0 + kk) * a.F + (f < a.F ? f : 0));
        }
      };
      // (masked HERE, a whole MFMA loop after the loads were issued: touching a loaded value earlier makes the wave wait for it there)
      auto stage = [&](int s) {
        const int c = s / ksteps, d = (s - c * ksteps) * 32 + 4 * (tid & 7);
        const float one = (c == 0 && d == a.D) ? 1.f : 0.f;                                         // (d == D: the bias column)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const bool live = rid[c][(tid >> 3) + 32 * i] >= 0 && d < a.D;
          float* p = As + ((tid >> 3) + 32 * i) * kPA + 4 * (tid & 7);
          p[0] = live ? ra[i].x : one; p[1] = live ? ra[i].y : 0.f; p[2] = live ? ra[i].z : 0.f; p[3] = live ? ra[i].w : 0.f;
        }
        const bool cols = f0 + 4 * (tid & 31) < a.F;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          *reinterpret_cast<float4*>(Bs + ((tid >> 5) + 8 * i) * kPB + 4 * (tid & 31)) = cols ? rb[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      };
#ifndef CAPAMD_NC_ABL
#define CAPAMD_NC_ABL 0        // profiling builds: 1 = the K loop fetches nothing after its first step, 2 = nor stages anything (MFMAs on stale tiles)
#endif
      fetch(0);
      for (int s = 0; s < steps; ++s) {
        if (!(CAPAMD_NC_ABL & 2) || s == 0) {
          __syncthreads();          // the previous step's reads are done
          stage(s);
          __syncthreads();
        }
        if (s + 1 < steps && !(CAPAMD_NC_ABL & 1)) fetch(s + 1);
        const float* ap = As + (wr * 32 + (lane & 31)) * kPA + (lane >> 5);
        const float* bp = Bs + (lane >> 5) * kPB + wc * 64 + (lane & 31);
#pragma unroll
        for (int kp = 0; kp < 16; ++kp) {
          const float a0 = ap[2 * kp];
          const float b0 = bp[2 * kp * kPB], b1 = bp[2 * kp * kPB + 32];
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
        }
      }
    }
    // C/D map of a 32 x 32 tile: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).  (A tile without work: zeros.)
    float* out = a.out[seg];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int64_t o = orow[wr * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)];
      if (o < 0) continue;
      float* dst = out + o * (int64_t)a.F;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int f = f0 + wc * 64 + jj * 32 + (lane & 31);
        if (f < a.F) dst[f] = acc[jj][e];
      }
    }
  }
}

// ---- forward on bf16 x 3 ----------------------------------------------------------------------------------------------------------------
// The same units, queue and epilogue as ngram_forward_kernel, the K loop on v_mfma_f32_32x32x16_bf16 with both operands split three ways:
// the weights once per call (ngram_pack3_kernel), the gathered rows at the LDS stage (4 VALU instructions per value + packing).  Per
// 32-dimension step a wave issues 24 MFMAs of 32 cycles where the fp32 form issues 32 of 64.
constexpr int kRows3 = 128;            // positions per tile of the bf16 x 3 forward (the weight planes are re-staged per tile and step: 64-position tiles made the LDS, not the MFMAs, the bound)
constexpr int kP3 = 80;                // LDS pitch (bytes) of a 32-k row of one plane: 64 B of bf16 + 16: ds_read_b128 of 32 rows conflict-free
__global__ __launch_bounds__(256, 2) void ngram_forward3_kernel(ConvArgs a, const unsigned short* wt3, int tiles0, int tiles, int panels, unsigned* queue) {
  __shared__ __attribute__((aligned(16))) unsigned char A3[3 * kRows3 * kP3];
  __shared__ __attribute__((aligned(16))) unsigned char B3[3 * 128 * kP3];
  __shared__ int64_t rid[kNcMaxG][kRows3];
  __shared__ int64_t orow[kRows3];
  __shared__ int any_real;
  __shared__ unsigned unit_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned units = (unsigned)tiles * a.G * panels;
  const int ksteps = a.Dp / 32;
  const int wr = wave >> 1, wc = wave & 1;               // a wave: 64 positions x 64 filters
  const int64_t plane = (int64_t)n_parts_dev(a.G) * a.F * a.Dp;
  for (;;) {
    __syncthreads();
    if (tid == 0) {
      unit_s = atomicAdd(queue, 1u);
      any_real = 0;
    }
    __syncthreads();
    const unsigned unit = unit_s;
    if (unit >= units) return;
    const int tile = unit % tiles, rest = unit / tiles;
    const int f0 = (rest % panels) * 128, g1 = a.G - rest / panels;          // heavy n-gram sizes first
    const int seg = tile < tiles0 ? 0 : 1;
    const int64_t m0 = (int64_t)(seg ? tile - tiles0 : tile) * kRows3;
    const int len = a.len[seg];
    const int64_t M = (int64_t)a.N * len;
    if (tid < kRows3) {
      bool bad = false, real = false;
      const int64_t m = m0 + tid;
      const unsigned n = m < M ? (unsigned)m / (unsigned)len : 0u;
      const int j = m < M ? (int)((unsigned)m - n * (unsigned)len) : 0;
      orow[tid] = m < M ? ((int64_t)n * a.G + (g1 - 1)) * len + j : -1;
      for (int c = 0; c < g1; ++c) {
        int64_t id = -1;
        if (m < M && j + c < len) {
          id = a.ids[seg][(int64_t)n * len + j + c];
          if (id < 0 || id >= a.V) {
            bad = true;
            id = -1;
          }
        }
        rid[c][tid] = id;
        if (c == 0 && id > 0) real = true;
      }
      if (bad) atomicOr(a.status, CAPAMD_STATUS_DOC_ID_RANGE);
      if (real) any_real = 1;
    }
    __syncthreads();
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    if (any_real) {
      const int steps = g1 * ksteps;
      const unsigned short* wt = wt3 + (int64_t)part_of(g1, 0) * a.F * a.Dp;
      float4 ra[4];
      uint4 rb[6];
      auto fetch = [&](int s) {
        const int c = s / ksteps, k0 = (s - c * ksteps) * 32;
        const int d = k0 + 4 * (tid & 7);
        const int dcl = d < a.D ? d : 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int64_t id = rid[c][(tid >> 3) + 32 * i];
          ra[i] = *reinterpret_cast<const float4*>(a.emb + (id >= 0 ? id : 0) * a.D + dcl);        // (D % 4 == 0)
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const int idx = tid + 256 * j, p = idx >> 9, f = (idx >> 2) & 127, ch = idx & 3;
          const int fg = f0 + f < a.F ? f0 + f : 0;
          rb[j] = *reinterpret_cast<const uint4*>(wt + p * plane + ((int64_t)c * a.F + fg) * a.Dp + k0 + 8 * ch);
        }
      };
      auto stage = [&](int s) {
        const int c = s / ksteps, d = (s - c * ksteps) * 32 + 4 * (tid & 7);
        const float one = (c == 0 && d == a.D) ? 1.f : 0.f;                                         // (d == D: the bias column)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = (tid >> 3) + 32 * i;
          const bool live = rid[c][r] >= 0 && d < a.D;
          const Split3 s0 = split3(live ? ra[i].x : one), s1 = split3(live ? ra[i].y : 0.f), s2 = split3(live ? ra[i].z : 0.f), s3 = split3(live ? ra[i].w : 0.f);
          unsigned char* at = A3 + r * kP3 + 8 * (tid & 7);
          *reinterpret_cast<uint2*>(at) = make_uint2(s0.hi | (s1.hi << 16), s2.hi | (s3.hi << 16));
          *reinterpret_cast<uint2*>(at + kRows3 * kP3) = make_uint2(s0.mid | (s1.mid << 16), s2.mid | (s3.mid << 16));
          *reinterpret_cast<uint2*>(at + 2 * kRows3 * kP3) = make_uint2(s0.lo | (s1.lo << 16), s2.lo | (s3.lo << 16));
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const int idx = tid + 256 * j, p = idx >> 9, f = (idx >> 2) & 127, ch = idx & 3;
          *reinterpret_cast<uint4*>(B3 + (p * 128 + f) * kP3 + 16 * ch) = f0 + f < a.F ? rb[j] : make_uint4(0u, 0u, 0u, 0u);
        }
      };
      fetch(0);
      for (int s = 0; s < steps; ++s) {
        __syncthreads();          // the previous step's reads are done
        stage(s);
        __syncthreads();
        if (s + 1 < steps) fetch(s + 1);
        const unsigned char* ap = A3 + (wr * 64 + (lane & 31)) * kP3 + 16 * (lane >> 5);
        const unsigned char* bp = B3 + (wc * 64 + (lane & 31)) * kP3 + 16 * (lane >> 5);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          bf16x8v av[2][3], bv[2][3];
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            av[0][p] = *reinterpret_cast<const bf16x8v*>(ap + p * kRows3 * kP3 + 32 * kb);
            av[1][p] = *reinterpret_cast<const bf16x8v*>(ap + (p * kRows3 + 32) * kP3 + 32 * kb);
            bv[0][p] = *reinterpret_cast<const bf16x8v*>(bp + p * 128 * kP3 + 32 * kb);
            bv[1][p] = *reinterpret_cast<const bf16x8v*>(bp + (p * 128 + 32) * kP3 + 32 * kb);
          }
#pragma unroll
          for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {      // the small products first
              acc[ii][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[ii][2], bv[jj][0], acc[ii][jj], 0, 0, 0);
              acc[ii][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[ii][0], bv[jj][2], acc[ii][jj], 0, 0, 0);
              acc[ii][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[ii][1], bv[jj][1], acc[ii][jj], 0, 0, 0);
              acc[ii][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[ii][1], bv[jj][0], acc[ii][jj], 0, 0, 0);
              acc[ii][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[ii][0], bv[jj][1], acc[ii][jj], 0, 0, 0);
              acc[ii][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[ii][0], bv[jj][0], acc[ii][jj], 0, 0, 0);
            }
        }
      }
    }
    float* out = a.out[seg];
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t o = orow[wr * 64 + ii * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)];
        if (o < 0) continue;
        float* dst = out + o * (int64_t)a.F;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int f = f0 + wc * 64 + jj * 32 + (lane & 31);
          if (f < a.F) dst[f] = acc[ii][jj][e];
        }
      }
  }
}

// ---- backward -------------------------------------------------------------------------------------------------------------------------
constexpr int kBwdList = 2048;         // 32-position blocks a slice can hold (the batch limit: S * 2048 * 32 positions)

// DT = 32-row tiles of the padded embedding width a wave keeps accumulators for (Dp = 32 DT): 16 DT accumulator registers
template <int DT>
__global__ __launch_bounds__(256, 1) void ngram_backward_kernel(ConvArgs a, int blocks0, int blocks) {
  extern __shared__ __attribute__((aligned(16))) float bw_lds[];
  __shared__ int list[kBwdList];
  __shared__ int wave_cnt[4];
  constexpr int PX = 32 * DT + 32;               // pitch of a position's row: the two k rows of an MFMA land 32 banks apart
  constexpr int XV = DT;                         // float4 loads per thread and block: 8 DT per position, 8 threads per position
  float* Xs = bw_lds;                            // [32][PX]
  float* Rs = bw_lds + kBwdBlock * PX;           // [32][kPB]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int part = blockIdx.y, split = blockIdx.x, f0 = blockIdx.z * 128;
  int g1 = 1;
  while (part_of(g1 + 1, 0) <= part) ++g1;
  const int c = part - part_of(g1, 0);
  // this slice's blocks (split, split + S, ...) that hold a real token, in order: a position's gradient row is zero unless the position
  // itself is a real token (the kernel pooling masks pads)
  int total = 0;
  {
    const int mine = (blocks - split + a.S - 1) / a.S;           // (<= kBwdList: checked by the host)
    for (int base = 0; base < mine; base += 256) {
      const int k = base + tid, blk = split + k * a.S;
      bool real = false;
      if (k < mine) {
        const int seg = blk < blocks0 ? 0 : 1;
        const int64_t m0 = (int64_t)(seg ? blk - blocks0 : blk) * kBwdBlock, M = (int64_t)a.N * a.len[seg];
        const int64_t* p = a.ids[seg] + m0;
        const int n = M - m0 < kBwdBlock ? (int)(M - m0) : kBwdBlock;
        for (int r = 0; r < n; ++r) real |= p[r] != 0;
      }
      const uint64_t set = __ballot(real);
      if (lane == 0) wave_cnt[wave] = __builtin_popcountll(set);
      __syncthreads();
      int before = total;
      for (int w = 0; w < wave; ++w) before += wave_cnt[w];
      if (real) list[before + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(set >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)set, 0u))] = blk;
      total += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
      __syncthreads();
    }
  }
  f32x16 acc[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
  // a thread stages ONE position of a block (8 threads per position): its tap-c table row, its gradient row
  const int r = tid >> 3, sub = tid & 7;
  float4 rx[XV], rr[4];
  bool bad = false, live_row = false, in_range = false;
  // the table row of this thread's position of block blk: one unconditional load (the id, clamped address) whose value is looked at one
  // block LATER - the id -> row address -> row loads chain costs a round trip per link, so the id travels a block ahead of its rows
  // (positions as 32-bit numbers - N * len < 2^31, checked by the host: one 32-bit division per thread and block)
  auto tap_id = [&](int blk) -> int64_t {
    const int seg = blk < blocks0 ? 0 : 1;
    const unsigned m = (unsigned)(seg ? blk - blocks0 : blk) * kBwdBlock + r, len = (unsigned)a.len[seg];
    const unsigned n = m / len, j = m - n * len + c;
    return a.ids[seg][(n < (unsigned)a.N && j < len) ? (int64_t)n * len + j : 0];           // (the raw value: fetch() decides whether it counts)
  };
  auto fetch = [&](int blk, int64_t id) {
    const int seg = blk < blocks0 ? 0 : 1;
    const unsigned m = (unsigned)(seg ? blk - blocks0 : blk) * kBwdBlock + r, len = (unsigned)a.len[seg];
    const unsigned M = (unsigned)a.N * len;
    const unsigned n = m < M ? m / len : 0u, j = m < M ? m - n * len : 0u;
    if (!(m < M && j + c < len)) id = -1;                                  // beyond the sequence end: the ConstantPad1d's zero row
    else if (id < 0 || id >= a.V) {
      bad = true;
      id = -1;
    }
    const float* row = a.emb + (id >= 0 ? id : 0) * a.D;
    const float* src = a.dout[seg] + ((((int64_t)n * a.G + (g1 - 1)) * len + j) * (int64_t)a.F);
    // (unconditional loads from clamped addresses, masked afterwards: see the forward kernel)
#pragma unroll
    for (int k = 0; k < XV; ++k) {
      const int d = 4 * (sub + 8 * k);
      rx[k] = *reinterpret_cast<const float4*>(row + (d < a.D ? d : 0));           // (D % 4 == 0)
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = f0 + 4 * (sub + 8 * i);
      rr[i] = *reinterpret_cast<const float4*>(src + (f < a.F ? f : 0));
    }
    live_row = id >= 0;
    in_range = m < M;
  };
  // (masked a whole MFMA loop after the loads were issued)
  auto stage = [&]() {
#pragma unroll
    for (int k = 0; k < XV; ++k) {
      const int d = 4 * (sub + 8 * k);
      const bool live = live_row && d < a.D;
      float4 v;
      v.x = live ? rx[k].x : (c == 0 && d == a.D && in_range ? 1.f : 0.f);        // (d == D: the bias column)
      v.y = live ? rx[k].y : 0.f;
      v.z = live ? rx[k].z : 0.f;
      v.w = live ? rx[k].w : 0.f;
      *reinterpret_cast<float4*>(Xs + r * PX + d) = v;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int fo = 4 * (sub + 8 * i);
      *reinterpret_cast<float4*>(Rs + r * kPB + fo) = (in_range && f0 + fo < a.F) ? rr[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  int64_t id_next = -1;
  if (total > 0) fetch(list[0], tap_id(list[0]));
  if (total > 1) id_next = tap_id(list[1]);
  for (int i = 0; i < total; ++i) {
    __syncthreads();            // the previous block's reads are done
    stage();
    __syncthreads();
    if (i + 1 < total) fetch(list[i + 1], id_next);
    if (i + 2 < total) id_next = tap_id(list[i + 2]);
    const float* xp = Xs + (lane >> 5) * PX + (lane & 31);
    const float* rp = Rs + (lane >> 5) * kPB + wave * 32 + (lane & 31);
#pragma unroll 2
    for (int kp = 0; kp < kBwdBlock / 2; ++kp) {
      const float bv = rp[2 * kp * kPB];
      float av[DT];
#pragma unroll
      for (int t = 0; t < DT; ++t) av[t] = xp[2 * kp * PX + 32 * t];
#pragma unroll
      for (int t = 0; t < DT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv, acc[t], 0, 0, 0);
    }
  }
  if (bad) atomicOr(a.status, CAPAMD_STATUS_DOC_ID_RANGE);
  float* dst = a.partial + ((int64_t)split * gridDim.y + part) * a.Dp * a.F;
  const int f = f0 + wave * 32 + (lane & 31);
  if (f < a.F) {
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int d = 32 * t + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        dst[(int64_t)d * a.F + f] = acc[t][e];
      }
  }
}

// partial panels -> the Conv1d layout, slices summed in their order
__global__ __launch_bounds__(256) void ngram_reduce_kernel(ConvArgs a, int parts) {
  const int part = blockIdx.y;
  int g1 = 1;
  while (part_of(g1 + 1, 0) <= part) ++g1;
  const int c = part - part_of(g1, 0);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= (a.D + 1) * a.F) return;
  const int d = i / a.F, f = i - d * a.F;
  if (d == a.D && c != 0) return;
  float v = 0.f;
  for (int s = 0; s < a.S; ++s) v += a.partial[(((int64_t)s * parts + part) * a.Dp + d) * a.F + f];
  if (d < a.D) a.dw[g1 - 1][((int64_t)f * a.D + d) * g1 + c] = v;
  else a.db[g1 - 1][f] = v;
}

int conv_check(const ConvArgs& a) {
  if (!a.ids[0] || !a.ids[1] || !a.emb || !a.wt || !a.status) return CAPAMD_ERR_ARG;
  if (a.N < 0 || a.len[0] < 1 || a.len[1] < 1 || a.V < 1 || a.D < 4 || (a.D & 3) || a.D + 1 > kNcMaxDp || a.G < 1 || a.G > kNcMaxG || a.F < 4 || (a.F & 3) ||
      a.F > 256)
    return CAPAMD_ERR_ARG;
  for (int g = 0; g < a.G; ++g)
    if (!a.w[g] || !a.b[g]) return CAPAMD_ERR_ARG;
  if ((reinterpret_cast<uintptr_t>(a.emb) | reinterpret_cast<uintptr_t>(a.wt)) & 15) return CAPAMD_ERR_ALIGN;
  return CAPAMD_OK;
}

int padded_width(int D) { return (D + 1 + 31) / 32 * 32; }
int n_parts(int G) { return G * (G + 1) / 2; }
int n_splits(int G, int F) { return 256 / (n_parts(G) * ((F + 127) / 128)) > 0 ? 256 / (n_parts(G) * ((F + 127) / 128)) : 1; }

}  // namespace

extern "C" size_t capamd_ngram_conv_workspace_floats(int D, int G, int F, int backward) {
  if (D < 1 || G < 1 || G > kNcMaxG || F < 1) return 0;
  const size_t panel = (size_t)n_parts(G) * padded_width(D) * F;
  // (forward: the tap panels - fp32 [d][f], or three bf16 planes [f][d] = 1.5 x the bytes - and the work queue's counter)
  return backward ? panel * n_splits(G, F) : panel + panel / 2 + 8;
}

extern "C" int capamd_ngram_conv_forward(const int64_t* q_ids, const int64_t* d_ids, int N, int Q, int L, const float* emb, int64_t V, int D,
                                         const float* const* conv_w, const float* const* conv_b, int G, int F, float* qrep, float* drep,
                                         float* workspace, size_t workspace_floats, int* status, void* stream) {
  if (!conv_w || !conv_b || !qrep || !drep) return CAPAMD_ERR_ARG;
  ConvArgs a{};
  a.ids[0] = q_ids; a.ids[1] = d_ids; a.N = N; a.len[0] = Q; a.len[1] = L; a.emb = emb; a.V = V; a.D = D; a.Dp = padded_width(D); a.G = G; a.F = F;
  for (int g = 0; g < G && g < kNcMaxG; ++g) { a.w[g] = conv_w[g]; a.b[g] = conv_b[g]; }
  a.wt = workspace; a.out[0] = qrep; a.out[1] = drep; a.status = status;
  const int rc = conv_check(a);
  if (rc != CAPAMD_OK) return rc;
  if (workspace_floats < capamd_ngram_conv_workspace_floats(D, G, F, 0)) return CAPAMD_ERR_WORKSPACE;
  if ((reinterpret_cast<uintptr_t>(qrep) | reinterpret_cast<uintptr_t>(drep)) & 3) return CAPAMD_ERR_ALIGN;
  if (N == 0) return CAPAMD_OK;
  if ((int64_t)N * L >= (1ll << 31) || (int64_t)N * Q >= (1ll << 31)) return CAPAMD_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  (void)hipGetLastError();
  const size_t panel = (size_t)n_parts(G) * a.Dp * F;
  unsigned* queue = reinterpret_cast<unsigned*>(workspace + panel + panel / 2 + 4);
  unsigned short* wt3 = reinterpret_cast<unsigned short*>(workspace);
  if (CAPAMD_NGRAM_FP32) hipLaunchKernelGGL(ngram_pack_kernel, dim3((a.Dp * F + 255) / 256, n_parts(G)), dim3(256), 0, s, a, queue);
  else hipLaunchKernelGGL(ngram_pack3_kernel, dim3((a.Dp * F + 255) / 256, n_parts(G)), dim3(256), 0, s, a, wt3, queue);
  const int rows = CAPAMD_NGRAM_FP32 ? kFwdRows : kRows3;
  const int64_t tiles0 = ((int64_t)N * Q + rows - 1) / rows, tiles1 = ((int64_t)N * L + rows - 1) / rows;
  const int panels = (F + 127) / 128;
  if ((tiles0 + tiles1) * G * panels >= (1ll << 31)) return CAPAMD_ERR_ARG;
  const int64_t units = (tiles0 + tiles1) * G * panels;
  if (CAPAMD_NGRAM_FP32)
    hipLaunchKernelGGL(ngram_forward_kernel, dim3((unsigned)(units < 768 ? units : 768)), dim3(256), 0, s, a, (int)tiles0, (int)(tiles0 + tiles1), panels, queue);
  else
    hipLaunchKernelGGL(ngram_forward3_kernel, dim3((unsigned)(units < 512 ? units : 512)), dim3(256), 0, s, a, wt3, (int)tiles0, (int)(tiles0 + tiles1), panels,
                       queue);
  return hipGetLastError() == hipSuccess ? CAPAMD_OK : CAPAMD_ERR_LAUNCH;
}

extern "C" int capamd_ngram_conv_backward(const int64_t* q_ids, const int64_t* d_ids, int N, int Q, int L, const float* emb, int64_t V, int D,
                                          int G, int F, const float* dqrep, const float* ddrep, float* const* dconv_w, float* const* dconv_b,
                                          float* workspace, size_t workspace_floats, int* status, void* stream) {
  if (!dconv_w || !dconv_b || !dqrep || !ddrep) return CAPAMD_ERR_ARG;
  ConvArgs a{};
  a.ids[0] = q_ids; a.ids[1] = d_ids; a.N = N; a.len[0] = Q; a.len[1] = L; a.emb = emb; a.V = V; a.D = D; a.Dp = padded_width(D); a.G = G; a.F = F;
  for (int g = 0; g < G && g < kNcMaxG; ++g) {
    a.dw[g] = dconv_w[g]; a.db[g] = dconv_b[g];
    a.w[g] = dconv_w[g]; a.b[g] = dconv_b[g];          // (only checked for null)
  }
  a.wt = workspace; a.partial = workspace; a.dout[0] = dqrep; a.dout[1] = ddrep; a.status = status;
  const int rc = conv_check(a);
  if (rc != CAPAMD_OK) return rc;
  if (workspace_floats < capamd_ngram_conv_workspace_floats(D, G, F, 1)) return CAPAMD_ERR_WORKSPACE;
  if ((reinterpret_cast<uintptr_t>(dqrep) | reinterpret_cast<uintptr_t>(ddrep)) & 15) return CAPAMD_ERR_ALIGN;
  if ((int64_t)N * L >= (1ll << 31) || (int64_t)N * Q >= (1ll << 31)) return CAPAMD_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  (void)hipGetLastError();
  a.S = n_splits(G, F);
  const int parts = n_parts(G);
  const int blocks0 = (int)(((int64_t)N * Q + kBwdBlock - 1) / kBwdBlock), blocks1 = (int)(((int64_t)N * L + kBwdBlock - 1) / kBwdBlock);
  const dim3 grid(a.S, parts, (F + 127) / 128);
  if ((blocks0 + blocks1 + a.S - 1) / a.S > kBwdList) return CAPAMD_ERR_ARG;          // (S * 65,536 positions: far beyond a training batch)
  const size_t lds = (size_t)kBwdBlock * (a.Dp + 32 + kPB) * 4;
  switch (a.Dp / 32) {
#define CASE(DT_)                                                                                                                          \
  case DT_: {                                                                                                                              \
    auto k = ngram_backward_kernel<DT_>;                                                                                                   \
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)         \
      return CAPAMD_ERR_LAUNCH;                                                                                                            \
    hipLaunchKernelGGL(k, grid, dim3(256), lds, s, a, blocks0, blocks0 + blocks1);                                                         \
  } break;
    CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10)
#undef CASE
    default: return CAPAMD_ERR_ARG;
  }
  hipLaunchKernelGGL(ngram_reduce_kernel, dim3(((D + 1) * F + 255) / 256, parts), dim3(256), 0, s, a, parts);
  return hipGetLastError() == hipSuccess ? CAPAMD_OK : CAPAMD_ERR_LAUNCH;
}
