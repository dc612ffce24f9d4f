// Embedding-table packing and the stand-alone similarity matrix (reference
// capreolus/reranker/common.py:143-182; create_emb_layer common.py:279-288) for gfx950.
#include "capreolus_amd.h"
#include "interaction.h"

using namespace capamd;

namespace {

// One 16-lane group per table row.  Writes the packed row and its den = |row|_2 + 1e-9f.
// Sum-of-squares order: lane partial s = fma(v,v,s) over the lane's floats in increasing index,
// then the same 16-lane tree as the dot product (interaction.h).
__global__ __launch_bounds__(kThreads) void pack_rows_kernel(const float* __restrict__ emb, int64_t V, int D, int64_t ld,
                                                              float* __restrict__ packed) {
  const int RS = row_stride_for_dim(D);
  const int NV = RS / 64;
  const int lane16 = threadIdx.x & 15;
  const int64_t row = (int64_t)blockIdx.x * kGroupsPerWG + (threadIdx.x >> 4);
  if (row >= V) return;  // whole group exits together
  const float* src = emb + row * ld;
  float* dst = packed + row * (int64_t)RS;
  float s = 0.f;
  for (int i = 0; i < NV; ++i) {
    const int f0 = (i * 16 + lane16) * 4;
    float v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      v[c] = (f0 + c < D) ? src[f0 + c] : 0.f;
      s = __builtin_fmaf(v[c], v[c], s);
    }
    *reinterpret_cast<float4*>(dst + f0) = make_float4(v[0], v[1], v[2], v[3]);
  }
  s = group_allreduce(s);
  if (lane16 == 15) dst[RS - 1] = __builtin_sqrtf(s) + 1e-9f;
}

// sim[b][q][j] for every position (no compaction): the K2 row of SURVEY.md §8(a) on its own.
template <int NV>
__global__ __launch_bounds__(kThreads) void simmat_kernel(const int64_t* __restrict__ q_ids, const int64_t* __restrict__ d_ids,
                                                           int Q, int L, const float* __restrict__ packed, int64_t V,
                                                           float* __restrict__ sim_out, int* status) {
  const int b = blockIdx.x;
  const int lane16 = threadIdx.x & 15;
  const int g = threadIdx.x >> 4;
  const IdSource src{q_ids, d_ids, nullptr, nullptr, nullptr, nullptr};
  const PairIds ids = pair_ids(src, b, Q, L);
  float* out = sim_out + (int64_t)b * Q * L;
  for (int q0 = 0; q0 < Q; q0 += kQT) {
    QueryPass<NV> qp;
    load_query_pass<NV>(packed, ids, Q, q0, V, lane16, qp, status);
    for (int j = g; j < L; j += kGroupsPerWG) {
      int64_t did = ids.d(j);
      if (did >= V) {
        atomicOr(status, kErrDocIdRange);
        did = 0;
      }
      float sim;
      if (did > 0) {
        RowRegs<NV> d;
        load_row<NV>(packed, did, lane16, d);
        sim = row_sim_my<NV>(d, qp, lane16);
      } else {
        sim = (did < 0 && qp.id_my == (int)did) ? 1.f : 0.f;
      }
      if (lane16 < kQT && q0 + lane16 < Q) out[(int64_t)(q0 + lane16) * L + j] = sim;
    }
  }
}

}  // namespace

extern "C" {

int capamd_version(void) { return CAPAMD_VERSION; }
size_t capamd_interaction_workspace_bytes(void) { return 64; }
const char* capamd_arch(void) { return "gfx950"; }

int64_t capamd_packed_row_stride(int D) { return (D >= 1 && D <= 64 * kMaxNV - 1) ? row_stride_for_dim(D) : -1; }

int64_t capamd_packed_table_bytes(int64_t V, int D) {
  const int64_t rs = capamd_packed_row_stride(D);
  return rs < 0 || V < 1 ? -1 : V * rs * (int64_t)sizeof(float);
}

int capamd_pack_embeddings(const float* emb, int64_t V, int D, int64_t ld, float* packed, void* stream) {
  if (!emb || !packed || V < 1 || ld < D || capamd_packed_row_stride(D) < 0) return CAPAMD_ERR_ARG;
  if ((reinterpret_cast<uintptr_t>(packed) & 255) != 0) return CAPAMD_ERR_ALIGN;
  const int64_t blocks = (V + kGroupsPerWG - 1) / kGroupsPerWG;
  if (blocks > 0x7fffffff) return CAPAMD_ERR_ARG;
  (void)hipGetLastError();
  hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream, emb, V, D, ld, packed);
  return hipGetLastError() == hipSuccess ? CAPAMD_OK : CAPAMD_ERR_LAUNCH;
}

int capamd_similarity_matrix(const int64_t* q_ids, const int64_t* d_ids, int B, int Q, int L, const float* packed,
                             int64_t V, int D, float* sim_out, int* status, void* stream) {
  if (B == 0) return CAPAMD_OK;
  if (!q_ids || !d_ids || !packed || !sim_out || !status || B < 0 || Q < 1 || L < 1 || V < 1) return CAPAMD_ERR_ARG;
  if (capamd_packed_row_stride(D) < 0) return CAPAMD_ERR_ARG;
  if (B == 0) return CAPAMD_OK;
  hipStream_t s = (hipStream_t)stream;
  (void)hipGetLastError();
#define LAUNCH(NV_)                                                                                              \
  hipLaunchKernelGGL(simmat_kernel<NV_>, dim3(B), dim3(kThreads), 0, s, q_ids, d_ids, Q, L, packed, V, sim_out, \
                     status)
  switch (nv_for_dim(D)) {
    case 1: LAUNCH(1); break;
    case 2: LAUNCH(2); break;
    case 3: LAUNCH(3); break;
    case 4: LAUNCH(4); break;
    default: LAUNCH(5); break;
  }
#undef LAUNCH
  return hipGetLastError() == hipSuccess ? CAPAMD_OK : CAPAMD_ERR_LAUNCH;
}

}  // extern "C"
