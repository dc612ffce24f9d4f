/*
 * capreolus_amd.h — C ABI of the MI355X (gfx950) reranker scoring engine.
 *
 * Drop-in boundary for the inference forward pass of three Capreolus rerankers.  The reference
 * has no FFI (it is pure Python); each entry point below names the reference
 * Python interface it replaces (file:line under the reference tree) and is what a ctypes stub
 * on the reference side binds (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host; the library allocates
 *     nothing and frees nothing; the caller (the PyTorch caching allocator) owns inputs, outputs
 *     and workspaces.
 *   - `stream` is a hipStream_t (pass torch.cuda.current_stream().cuda_stream); kernels are
 *     enqueued and the call returns without synchronising (reference sync point:
 *     capreolus/trainer/pytorch.py:345).
 *   - return value: CAPAMD_OK or a CAPAMD_ERR_* (argument / launch errors, never an exception,
 *     never abort()).  Data-dependent errors (an id >= V, ...) are reported asynchronously by
 *     OR-ing CAPAMD_STATUS_* bits into the caller-zeroed int32 `status` word in device memory;
 *     the offending term is scored as a pad so the launch itself never faults.
 *   - ids are int64 exactly as the extractors emit them (embedtext.py:146-147,
 *     bertpassage.py:308-310): 0 = pad, negative = OOV term.
 */
#ifndef CAPREOLUS_AMD_H
#define CAPREOLUS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CAPAMD_VERSION 100

#define CAPAMD_OK 0
#define CAPAMD_ERR_ARG 1       /* null pointer / bad size / unsupported configuration */
#define CAPAMD_ERR_ALIGN 2     /* pointer not aligned as documented */
#define CAPAMD_ERR_LAUNCH 3    /* hipLaunchKernel reported an error */
#define CAPAMD_ERR_WORKSPACE 4 /* workspace too small */

#define CAPAMD_STATUS_DOC_ID_RANGE 1   /* a document term id >= V        (torch raises IndexError) */
#define CAPAMD_STATUS_QUERY_ID_RANGE 2 /* a query term id >= V */
#define CAPAMD_STATUS_QUERY_OOV 4      /* negative query id in DRMM      (reference DRMM.py:109 raises IndexError) */

/* library identity */
int capamd_version(void);
const char* capamd_arch(void); /* "gfx950" */

/* ---- embedding table ------------------------------------------------------------------------
 * Replaces create_emb_layer + nn.Embedding lookup (capreolus/reranker/common.py:279-288, :161).
 * The fp32 [V, D] table (row 0 = zeros, extractor/common.py:38-40) is re-laid out once into
 * rows of capamd_packed_row_stride(D) floats (256-byte aligned rows, |row|+1e-9 in the last
 * float); all forward calls take the packed table.  D <= 319. */
int64_t capamd_packed_row_stride(int D);               /* floats per packed row, -1 if unsupported */
int64_t capamd_packed_table_bytes(int64_t V, int D);   /* bytes the caller must allocate (256-B aligned) */
int capamd_pack_embeddings(const float* emb, int64_t V, int D, int64_t ld /* floats between rows of emb */,
                           float* packed, void* stream);

/* ---- SimilarityMatrix.forward (capreolus/reranker/common.py:170-182) ------------------------
 * sim_out fp32 [B, Q, L] = exact-match(OOV) + cosine(in-vocab), pads zeroed. */
int capamd_similarity_matrix(const int64_t* q_ids /*[B,Q]*/, const int64_t* d_ids /*[B,L]*/, int B, int Q, int L,
                             const float* packed, int64_t V, int D, float* sim_out, int* status, void* stream);

/* ---- KNRM_class.forward (capreolus/reranker/KNRM.py:39-55) behind KNRM.test (KNRM.py:96-101) --
 * mu, sigma: fp32 [K] (RbfKernel parameters, common.py:229-230; K <= 12, reference K = 11).
 * combine (KNRM.py:27-34): hidden == 0 -> score = w1[0,:K]·f + b1[0]            ("singlefc")
 *                          hidden  > 0 -> score = w2[0,:hidden]·tanh(w1·f + b1) + b2[0]
 *                          scoretanh != 0 applies a final tanh.
 * out fp32 [B].  No workspace. */
int capamd_knrm_forward(const int64_t* q_ids /*[B,Q]*/, const int64_t* d_ids /*[B,L]*/, int B, int Q, int L,
                        const float* packed, int64_t V, int D, const float* mu, const float* sigma, int K,
                        const float* w1, const float* b1, int hidden, const float* w2, const float* b2, int scoretanh,
                        float* out, int* status, void* stream);

/* ---- DRMM_class.forward (capreolus/reranker/DRMM.py:101-116) behind DRMM.test (DRMM.py:150-155)
 * idf fp32 [B,Q]; edges fp32 [nbins] = torch.linspace(-1,1,nbins+1)[1:] (DRMM.py:63);
 * hist_type 0 = CH, 1 = NH, 2 = LCH (DRMM.py:72-79); gate_type 0 = IDF (gate_w fp32 [1]),
 * 1 = TV (gate_w fp32 [D], emb_raw = the un-packed fp32 [V, D] table with leading dim ld);
 * ffw: w1 [nodes, nbins+1], b1 [nodes], w2 [nodes], b2 [1]; out_w [1], out_b [1].
 * nbins <= 63, nodes <= 64, Q <= 32.  counts_out: optional int32 [B, Q, nbins+1] raw bin counts
 * (may be NULL).  out fp32 [B]. */
int capamd_drmm_forward(const int64_t* q_ids, const int64_t* d_ids, const float* idf, int B, int Q, int L,
                        const float* packed, int64_t V, int D, const float* edges, int nbins, int hist_type,
                        int gate_type, const float* gate_w, const float* emb_raw, int64_t ld, const float* w1,
                        const float* b1, int nodes, const float* w2, const float* b2, const float* out_w,
                        const float* out_b, float* out, int32_t* counts_out, int* status, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CAPREOLUS_AMD_H */
