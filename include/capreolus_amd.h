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

#define CAPAMD_VERSION 400

#define CAPAMD_OK 0
#define CAPAMD_ERR_ARG 1       /* null pointer / bad size / unsupported configuration */
#define CAPAMD_ERR_ALIGN 2     /* pointer not aligned as documented */
#define CAPAMD_ERR_LAUNCH 3    /* hipLaunchKernel reported an error */
#define CAPAMD_ERR_WORKSPACE 4 /* workspace too small */

#define CAPAMD_STATUS_DOC_ID_RANGE 1   /* a document term id >= V        (torch raises IndexError) */
#define CAPAMD_STATUS_QUERY_ID_RANGE 2 /* a query term id >= V */
#define CAPAMD_STATUS_QUERY_OOV 4      /* negative query id in DRMM      (reference DRMM.py:109 raises IndexError) */
#define CAPAMD_STATUS_SCORE_NAN 8      /* a NaN score reached the ranking kernels (ranked last) */
#define CAPAMD_STATUS_TIE_RANGE 16     /* capamd_ndcg_cut: a tie-break rank outside 0..n-1 */
#define CAPAMD_STATUS_LIST_QUERY 32    /* *_forward_lists: a pair of a list brings another query row (DRMM, DRMM-TKS: or another idf row) than
                                          the list's first pair - a list is ONE query against its documents */

/* library identity */
int capamd_version(void);
const char* capamd_arch(void); /* "gfx950" */

/* Per-call launch flags of the interaction scoring entries (`flags` argument; no process-wide state).
 * CAPAMD_LAUNCH_CONCURRENT: the caller keeps several scoring calls in flight on different streams (one candidate list per launch,
 * the reference's PytorchTrainer.predict loop at small evalbatch, trainer/pytorch.py:334-348), so a small launch no longer has the
 * chip to itself: the kernels then use their occupancy-oriented variant at every batch size instead of the latency-oriented one
 * they pick for a lone launch of <= 1536 pairs.  Scores are bit-identical either way. */
#define CAPAMD_LAUNCH_CONCURRENT 1u

/* Workspace of the interaction scoring entries (`workspace`, `workspace_bytes`): a few bytes of device memory the caller owns (any
 * contents; 4-byte aligned; one per call in flight).  Launches that outnumber the workgroups the chip holds run persistent
 * workgroups that draw their pairs from a ticket counter kept there (zeroed by the call, on `stream`).  NULL / too small: allowed,
 * every launch then runs one workgroup per pair.  Scores do not depend on it beyond the last bits of fp32 rounding. */
size_t capamd_interaction_workspace_bytes(void);

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
 * out fp32 [B].  workspace / flags: see CAPAMD_LAUNCH_* and capamd_interaction_workspace_bytes above.
 * L: the document's term list lives in LDS (8 bytes per position beside ~11 KiB of fixed parts): documents beyond ~19,000 positions
 * do not fit the 160 KiB of a workgroup and are refused with CAPAMD_ERR_ARG (DRMM alike). */
int capamd_knrm_forward(const int64_t* q_ids /*[B,Q]*/, const int64_t* d_ids /*[B,L]*/, int B, int Q, int L,
                        const float* packed, int64_t V, int D, const float* mu, const float* sigma, int K,
                        const float* w1, const float* b1, int hidden, const float* w2, const float* b2, int scoretanh,
                        float* out, int* status, void* workspace, size_t workspace_bytes, unsigned flags, void* stream);

/* Forward half of the training step (reference trainer/pytorch.py:96-99 -> KNRM.score):
 * the kernel-pooling features f[b][k] = sum_q mask_q log(sum_j K_k(sim_qj) + 1e-6) (KNRM.py:50-53) that feed `combine`,
 * and their derivatives w.r.t. the RbfKernel parameters (trainable when `gradkernels`).  The embedding is frozen
 * (`finetune = False`, KNRM.py:23); `combine` and the loss run under autograd on these [B, K] tensors.
 * dfdmu_out / dfdsigma_out may be NULL. */
int capamd_knrm_features(const int64_t* q_ids, const int64_t* d_ids, int B, int Q, int L, const float* packed, int64_t V,
                         int D, const float* mu, const float* sigma, int K, float* feat_out, float* dfdmu_out,
                         float* dfdsigma_out, int* status, void* stream);

/* One KNRM training step on the device, no host round trip (reference capreolus/trainer/pytorch.py:93-108):
 * `reranker.score()` on B (query, positive document, negative document) triples -> the trainer's pairwise loss (loss_type 0: hinge,
 * reranker/common.py:101-103; 1: softmax, :96-98) -> backward through `combine` (a single Linear, KNRM.py:27-34: `singlefc`; under tanh when
 * `scoretanh`) and the RBF kernels' mu / sigma (when train_kernels: `gradkernels`) -> torch.optim.Adam's in-place update (betas / eps as
 * given, no weight decay, no amsgrad).  The caller owns the step count and hands in what depends on it, computed in double as the plain
 * Adam of the reference does: step_size = lr / (1 - beta1^t), bc2_sqrt = sqrt(1 - beta2^t).
 * ptrs: DEVICE array of 3 P device pointers, P = 2 K + 2 - the parameters (mu_0 .. mu_{K-1}, sigma_0 .. sigma_{K-1}: one scalar each, as the
 * reference's state_dict names them, common.py:229-230; the Linear's weight [K]; its bias [1]), then their exp_avg, then their exp_avg_sq
 * (the mu / sigma moments may be NULL when !train_kernels).  loss_out: [1] the batch's mean loss.  workspace:
 * capamd_knrm_train_step_workspace_floats(B, K) floats.  B <= 1024.  Two launches on `stream`: capamd_knrm_features' kernel over the 2 B
 * documents (it reads the scalar kernel parameters through `ptrs`), then one workgroup for everything that is per batch. */
size_t capamd_knrm_train_step_workspace_floats(int B, int K);
int capamd_knrm_train_step(const int64_t* q_ids, const int64_t* pos_ids, const int64_t* neg_ids, int B, int Q, int L, const float* packed,
                           int64_t V, int D, int K, float* const* ptrs, int train_kernels, int scoretanh, int loss_type, float step_size,
                           float one_minus_beta1, float beta2, float eps, float bc2_sqrt, float* loss_out, float* workspace,
                           size_t workspace_floats, int* status, void* stream);

/* Same scoring from a device-resident candidate store (replaces the per-sample
 * PredSampler -> DataLoader collate -> .to(device) of capreolus/sampler/__init__.py:207-264 and
 * trainer/pytorch.py:334-342): the run's query / document id rows are uploaded ONCE as int32 tables
 * q_table [NQ, Q], d_table [ND, L]; a batch is B (query row, document row) index pairs. */
int capamd_knrm_forward_indexed(const int32_t* q_table, const int32_t* d_table, const int32_t* pair_q,
                                const int32_t* pair_d, int B, int Q, int L, const float* packed, int64_t V, int D,
                                const float* mu, const float* sigma, int K, const float* w1, const float* b1, int hidden,
                                const float* w2, const float* b2, int scoretanh, float* out, int* status, void* workspace,
                                size_t workspace_bytes, unsigned flags, void* stream);

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
                        const float* out_b, float* out, int32_t* counts_out, int* status, void* workspace, size_t workspace_bytes,
                        unsigned flags, void* stream);

/* Training-step forward half for DRMM: feat_out fp32 [B, Q, nbins+1] = the matching histogram after CH/NH/LCH
 * (DRMM._hist_map, DRMM.py:41-81; it has no trainable inputs, the embedding is frozen at DRMM.py:22); the
 * feed-forward net, gate and output layer run under autograd on it. */
int capamd_drmm_features(const int64_t* q_ids, const int64_t* d_ids, int B, int Q, int L, const float* packed, int64_t V,
                         int D, const float* edges, int nbins, int hist_type, float* feat_out, int* status, void* stream);

/* One DRMM training step on the device (see capamd_knrm_train_step; reference DRMM.py:101-116 under trainer/pytorch.py:93-108; gateType = IDF):
 * matching histograms of the 2 B documents, the 30 -> nodes -> 1 tanh net per query term, softmax idf gate, output layer, the pairwise loss
 * (0 hinge / 1 softmax), backward and torch.optim.Adam's in-place update, in two launches.  ptrs: DEVICE array of 3 x 7 device pointers -
 * ffw.0.weight [nodes, nbins + 1], ffw.0.bias [nodes], ffw.2.weight [nodes], ffw.2.bias [1], gates.weight [1], output_layer.weight [1],
 * output_layer.bias [1], then their exp_avg, then their exp_avg_sq.  idf: [B, Q].  workspace:
 * capamd_drmm_train_step_workspace_floats(B, Q, nbins, nodes) floats.  B <= 1024, nodes <= 16. */
size_t capamd_drmm_train_step_workspace_floats(int B, int Q, int nbins, int nodes);
int capamd_drmm_train_step(const int64_t* q_ids, const int64_t* pos_ids, const int64_t* neg_ids, const float* idf, int B, int Q, int L,
                           const float* packed, int64_t V, int D, const float* edges, int nbins, int hist_type, int nodes, float* const* ptrs,
                           int loss_type, float step_size, float one_minus_beta1, float beta2, float eps, float bc2_sqrt, float* loss_out,
                           float* workspace, size_t workspace_floats, int* status, void* stream);

/* indexed variant (see capamd_knrm_forward_indexed); idf_table fp32 [NQ, Q] is indexed by the pair's query row */
int capamd_drmm_forward_indexed(const int32_t* q_table, const int32_t* d_table, const float* idf_table,
                                const int32_t* pair_q, const int32_t* pair_d, int B, int Q, int L, const float* packed,
                                int64_t V, int D, const float* edges, int nbins, int hist_type, int gate_type,
                                const float* gate_w, const float* emb_raw, int64_t ld, const float* w1, const float* b1,
                                int nodes, const float* w2, const float* b2, const float* out_w, const float* out_b,
                                float* out, int32_t* counts_out, int* status, void* workspace, size_t workspace_bytes, unsigned flags,
                                void* stream);

/* ---- KNRM / DRMM over whole candidate lists (what PytorchTrainer.predict scores: one query, its first-stage documents;
 * capreolus/trainer/pytorch.py:310-353 over PredSampler's per-query lists, sampler/__init__.py:222-233) ------------------------
 * The similarity of a document term to the query depends on (query, term) only, and the documents of a list share their vocabulary:
 * per list every distinct term's packed row is gathered ONCE (same arithmetic as the per-pair entries: bit-identical similarities)
 * into a float4 table over the vocabulary, and every document is pooled from 16-byte lookups.  DRMM's scores are bit-identical to
 * capamd_drmm_forward's, KNRM's equal to fp32 rounding of the pooling sums (another summation order).
 * Pairs are laid out list after list: list l owns pairs list_offsets_host[l] .. list_offsets_host[l+1] (a HOST array of n_lists + 1
 * entries) and is scored against the query of its FIRST pair (and, DRMM, that pair's idf row); a pair whose own query (idf) row differs
 * sets CAPAMD_STATUS_LIST_QUERY - the scores of such a call are those of the first pair's query, not the reference's.  Ids either as [B,Q] / [B,L] int64
 * (q_ids, d_ids; the table arguments NULL) or through a candidate store (q_table, d_table, pair_q, pair_d; q_ids / d_ids NULL).
 * Q <= 8 (two blocks of four query terms; the reference's `maxqlen` is a free ConfigOption, extractor/embedtext.py:28-31, its forwards
 * take any Q: reranker/KNRM.py:39-55, DRMM.py:101-116; capamd_pacrr_forward_lists: Q <= 4); the other limits as the per-pair entries.
 * workspace: capamd_lists_workspace_bytes_q(n_lists, V, n_pairs, L, Q) bytes (16-byte aligned; any contents;
 * capamd_lists_workspace_bytes(...) = the same for Q <= 4), in two parts:
 *   per PAIR of the call   4 x L + 48 bytes: the document's real term ids, compacted to int32 by the first pass (what the pooling pass reads
 *                          instead of the [L] id row), and its pad / OOV counts - 3.2 KB per pair at L = 800 (capamd_pacrr_forward_lists keeps
 *                          416 B of features per pair there and runs its combine layers in one pass behind the convolutions; with n_pairs = 0
 *                          - no such room - every pair's combine layers run inside its convolution workgroup: same scores, slower)
 *   per LIST in flight     17 B x V + 5 KB (a 16-byte table entry and a flag byte per vocabulary id): 6.8 MB at V = 400,001, 68 MB at V = 4 M
 *                          (queries of five to eight terms: a second 16-byte entry per id and a second 5 KB: 33 B x V + 10 KB);
 *                          at most 256 lists are in flight at a time - FEWER when the buffer is smaller (a caller bounds the workspace by
 *                          handing in less: the lists are then processed in more, smaller groups; CAPAMD_ERR_WORKSPACE below one list). */
size_t capamd_lists_workspace_bytes(int n_lists, int64_t V, int64_t n_pairs, int L);
size_t capamd_lists_workspace_bytes_q(int n_lists, int64_t V, int64_t n_pairs, int L, int Q);
int capamd_knrm_forward_lists(const int64_t* q_ids, const int64_t* d_ids, const int32_t* q_table, const int32_t* d_table, const int32_t* pair_q,
                              const int32_t* pair_d, const int64_t* list_offsets_host, int n_lists, int Q, int L, const float* packed, int64_t V,
                              int D, const float* mu, const float* sigma, int K, const float* w1, const float* b1, int hidden, const float* w2,
                              const float* b2, int scoretanh, float* out, int* status, void* workspace, size_t workspace_bytes, void* stream);
int capamd_drmm_forward_lists(const int64_t* q_ids, const int64_t* d_ids, const int32_t* q_table, const int32_t* d_table, const int32_t* pair_q,
                              const int32_t* pair_d, const float* idf, const int64_t* list_offsets_host, int n_lists, int Q, int L,
                              const float* packed, int64_t V, int D, const float* edges, int nbins, int hist_type, int gate_type,
                              const float* gate_w, const float* emb_raw, int64_t ld, const float* w1, const float* b1, int nodes, const float* w2,
                              const float* b2, const float* out_w, const float* out_b, float* out, int32_t* counts_out, int* status,
                              void* workspace, size_t workspace_bytes, void* stream);
/* DRMM-TKS the same way (the float form of the table, then per document the top-k of every query term's lookups): scores bit-identical
 * to capamd_drmmtks_forward's (selections of bit-identical similarities, fed to the Linear in the same order).  IDF gate; Q <= 4. */
int capamd_drmmtks_forward_lists(const int64_t* q_ids, const int64_t* d_ids, const int32_t* q_table, const int32_t* d_table, const int32_t* pair_q,
                                 const int32_t* pair_d, const float* idf, const int64_t* list_offsets_host, int n_lists, int Q, int L,
                                 const float* packed, int64_t V, int D, int topk, const float* gate_w, const float* ffw_w, const float* ffw_b,
                                 const float* out_w, const float* out_b, float* out, int* status, void* workspace, size_t workspace_bytes,
                                 void* stream);

/* PACRR the same way (the float form of the table, then capamd_pacrr_forward's MFMA kernel with a table lookup per position as its front
 * end): scores bit-identical to capamd_pacrr_forward's.  Q <= 4, nfilters <= 32, maxgram <= 3, L <= 1024. */
int capamd_pacrr_forward_lists(const int64_t* q_ids, const int64_t* d_ids, const int32_t* q_table, const int32_t* d_table, const int32_t* pair_q,
                               const int32_t* pair_d, const float* idf, const int64_t* list_offsets_host, int n_lists, int Q, int L,
                               const float* packed, int64_t V, int D, int mingram, int maxgram, int nfilters, int kmax, const float* conv_w,
                               const float* conv_b, int use_idf, int combine, int nonlinearity, const float* w1, const float* b1, const float* w2,
                               const float* b2, const float* w3, const float* b3, float* out, int* status, void* workspace, size_t workspace_bytes,
                               void* stream);

/* ---- DRMMTKS_class.forward (capreolus/reranker/DRMMTKS.py:50-64) behind DRMMTKS.test (:105-110) ------------------
 * A sibling of DRMM on the same fused front end: per query term the top-k similarities over all
 * L positions -> Linear(topk, 1) + tanh (ffw_w fp32 [topk], ffw_b [1]) -> IDF gate (gate_w [1]) -> output layer.
 * topk <= 16, topk <= L, Q <= 32.  Only gateType = IDF (the reference's TV branch feeds integer ids to nn.Linear). */
int capamd_drmmtks_forward(const int64_t* q_ids, const int64_t* d_ids, const float* idf, int B, int Q, int L,
                           const float* packed, int64_t V, int D, int topk, const float* gate_w, const float* ffw_w,
                           const float* ffw_b, const float* out_w, const float* out_b, float* out, int* status, void* stream);
/* Training-step half of DRMM-TKS (row N3; reference trainer/pytorch.py:96-99 -> DRMMTKS.score): features[B][Q][topk] = the sorted
 * top-k similarities of every query term (DRMMTKS.py:55-56).  The embedding table is frozen (freezeemb), so no gradient flows
 * through them; the Linear(topk,1)/tanh, the idf gate and the output layer (a few dozen flops per pair) run under autograd. */
int capamd_drmmtks_features(const int64_t* q_ids, const int64_t* d_ids, int B, int Q, int L, const float* packed, int64_t V, int D,
                            int topk, float* features, int* status, void* stream);

/* One DRMM-TKS training step on the device (see capamd_knrm_train_step; reference DRMMTKS.py:50-64 under trainer/pytorch.py:93-108): top-k
 * features of the 2 B documents, Linear(topk, 1) / tanh, softmax idf gate, output layer, the pairwise loss (0 hinge / 1 softmax), backward and
 * torch.optim.Adam's in-place update, in two launches.  ptrs: DEVICE array of 3 x 5 device pointers - ffw.0.weight [topk], ffw.0.bias [1],
 * gates.weight [1], output_layer.weight [1], output_layer.bias [1], then their exp_avg, then their exp_avg_sq.  idf: [B, Q] of the queries.
 * workspace: capamd_drmmtks_train_step_workspace_floats(B, Q, topk) floats.  B <= 1024. */
size_t capamd_drmmtks_train_step_workspace_floats(int B, int Q, int topk);
int capamd_drmmtks_train_step(const int64_t* q_ids, const int64_t* pos_ids, const int64_t* neg_ids, const float* idf, int B, int Q, int L,
                              const float* packed, int64_t V, int D, int topk, float* const* ptrs, int loss_type, float step_size,
                              float one_minus_beta1, float beta2, float eps, float bc2_sqrt, float* loss_out, float* workspace,
                              size_t workspace_floats, int* status, void* stream);

/* ---- PTBERTMaxP_Class.predict_step (capreolus/reranker/ptBERTMaxP.py:67-96) behind PTBERTMaxP.test
 * (ptBERTMaxP.py:134-135), including the transformers.BertForSequenceClassification forward it calls
 * at :82 (embeddings, 12 post-LN encoder layers, pooler, classifier, logit 1).
 *
 * The model is described by plain pointers.  fp32 tensors are read in place from the live
 * nn.Module parameters; the GEMM weights are converted once to bf16 into a caller-owned `blob`
 * (capamd_bert_pack_layer; re-run it after load_weights / an optimizer step), the per-layer
 * biases and LayerNorm vectors are gathered into `layer_f32`.
 * hidden = 64*heads, hidden % 64 == 0, hidden <= 1024, ffn % 64 == 0; S a multiple of 32 up to 256, or 384, or 512. */
typedef struct capamd_bert_model {
  int hidden, layers, heads, ffn, vocab, max_pos, type_vocab;
  int compute_dtype;      /* 16-bit operand/activation type: 0 = bf16 (default), 1 = fp16 (the reference's amp autocast type;
                             ~8x smaller rounding error, same MFMA rate, narrower range) */
  const float* word_emb;  /* [vocab, hidden]      bert.embeddings.word_embeddings.weight */
  const float* pos_emb;   /* [max_pos, hidden]    bert.embeddings.position_embeddings.weight */
  const float* type_emb;  /* [type_vocab, hidden] bert.embeddings.token_type_embeddings.weight */
  const float* emb_ln_g;  /* [hidden]             bert.embeddings.LayerNorm.weight */
  const float* emb_ln_b;  /* [hidden]             bert.embeddings.LayerNorm.bias */
  const float* pooler_w;  /* [hidden, hidden]     bert.pooler.dense.weight */
  const float* pooler_b;  /* [hidden] */
  const float* cls_w;     /* [2, hidden]          classifier.weight */
  const float* cls_b;     /* [2] */
  const void* blob;       /* capamd_bert_blob_bytes(): per layer Wqkv[3H,H] | Wo[H,H] | W1[F,H] | W2[H,F] (+ gamma-scaled Wqkv, W1), 16-bit */
  const float* layer_f32; /* layers * capamd_bert_layer_f32_floats(): bqkv | bo | ln1.g | ln1.b | b1 | b2 | ln2.g | ln2.b | folded-LayerNorm vectors */
  /* RoBERTa bodies (transformers.RobertaForSequenceClassification behind ptBERTMaxP.py:46-48, 57-58): the same encoder with
   *   - position ids counted over the non-pad tokens: pad_id + #(ids[0..i] != pad_id) for a non-pad token, pad_id for a pad
   *     (pos_pad_id >= 0; -1 = BERT: position i),
   *   - its own LayerNorm epsilon (1e-5; 0 = BERT's 1e-12),
   *   - a `dense -> tanh -> out_proj` head on the first token: the pooler / classifier arithmetic under other names
   *     (pooler_w/b = classifier.dense, cls_w/b = classifier.out_proj) and a one-row token-type table. */
  float ln_eps;
  int pos_pad_id;
} capamd_bert_model;

int64_t capamd_bert_blob_bytes(const capamd_bert_model* m);        /* only the int fields are read */
int64_t capamd_bert_layer_f32_floats(const capamd_bert_model* m);  /* floats per layer */
/* tensors_host: HOST array of 18 DEVICE pointers for encoder layer `layer`, in this order:
 * attention.self.query.{weight,bias}, .key.{weight,bias}, .value.{weight,bias}, attention.output.dense.{weight,bias},
 * attention.output.LayerNorm.{weight,bias}, intermediate.dense.{weight,bias}, output.dense.{weight,bias},
 * output.LayerNorm.{weight,bias}, and [16], [17] = {weight, bias} of the LayerNorm whose output is this layer's input
 * (the previous layer's output.LayerNorm; both NULL for layer 0, which reads the normalised embeddings).  The last
 * two are folded into this layer's QKV and attention-output epilogues (LayerNorm fused into the GEMMs: weights
 * pre-scaled by gamma, per-column sums and folded biases stored behind the plain vectors of `layer_f32`). */
int capamd_bert_pack_layer(const capamd_bert_model* m, int layer, const float* const* tensors_host, void* blob,
                           float* layer_f32, void* stream);
/* workspace for scoring `total_passages` = B*P passages in micro-batches of `passages_per_microbatch` */
int64_t capamd_bert_workspace_bytes(const capamd_bert_model* m, int S, int64_t passages_per_microbatch,
                                    int64_t total_passages);
/* ids/mask/seg int64 [B, P, S] (bertpassage.py:313-325); aggregation 0 max, 1 first, 2 sum, 3 avg
 * (ptBERTMaxP.py:85-94; avg divides by the batch-wide passage count as the reference does);
 * out fp32 [B]; passage_logits_out optional fp32 [B*P]; workspace 256-byte aligned. */
int capamd_bert_maxp_forward(const int64_t* ids, const int64_t* mask, const int64_t* seg, int B, int P, int S,
                             const capamd_bert_model* m, int aggregation, int64_t passages_per_microbatch,
                             void* workspace, int64_t workspace_bytes, float* out, float* passage_logits_out,
                             int* status, void* stream);

/* The passage pooling of predict_step alone (capreolus/reranker/ptBERTMaxP.py:75-94) over fp32 passage logits [B*P]:
 * aggregation 0 max, 1 first, 2 sum, 3 avg, with passage_mask = (sum(mask*seg) > 5) from the FULL [B,P,S] mask / seg
 * arrays.  Used when the passages of a call are encoded in length buckets (capreolus_amd.engine.BertEngine,
 * skip_padding): passages are independent and padded positions never reach a real token, so a passage whose tokens end
 * before position 32k can be encoded at S = 32k (any supported length) with bit-identical logits.  count_scratch: 4 bytes. */
int capamd_maxp_pool(const float* passage_logits, const int64_t* mask, const int64_t* seg, int B, int P, int S, int aggregation,
                     float* out, int* count_scratch, void* stream);

/* Encoder building blocks (what BertSelfAttention / nn.Linear + activation compute inside the HF model
 * the reference calls at ptBERTMaxP.py:82).  16-bit (bf16 or fp16) operands, fp32 accumulation.
 * capamd_bert_gemm: out[M,N] = A[M,K] · W[N,K]^T + bias, epilogue 0: bf16 out; 1: erf-GELU, bf16 out;
 * 4: + resid[M,N] (bf16), bf16 out (fp32 sum, one rounding).  M, N, K multiples of 64.
 * Layout bits OR-ed into `epilogue` (M, N multiples of 256, K >= 128; not with epilogue 4): the engine's internal
 * "chunk-major" activation layout of a [R][C] 16-bit tensor - blocks of 32 rows, inside a block the 16-byte chunks
 * of one column position of all 32 rows contiguous:  (r, c) -> (((r/32)*(C/8) + c/8)*32 + r%32)*8 + c%8.
 * Its producers store straight from MFMA registers (no LDS regrouping), its consumers still fetch full 128-byte lines. */
#define CAPAMD_GEMM_A_CHUNK_MAJOR 0x200   /* A is chunk-major */
#define CAPAMD_GEMM_OUT_CHUNK_MAJOR 0x100 /* out is chunk-major */
#define CAPAMD_GEMM_RING_256 0x800         /* with W chunk-major: the ring kernel's 256-row tile (one workgroup per CU) instead of its default 128-row tile (two per CU);
                                            with a chunk-major output and K % 64 == 0 that tile runs on 16x16x32 MFMAs (k accumulated in steps of 32: not the
                                            bits of the other kernels, which accumulate in steps of 16) */
#define CAPAMD_GEMM_RING_MFMA16 0x2000     /* without CAPAMD_GEMM_RING_256: the 128-row tile (two workgroups per CU) on 16x16x32 MFMAs too (chunk-major output, K % 64 == 0;
                                            not for the QKV epilogue) - the bits of the 256-row 16x16x32 tile */
#define CAPAMD_GEMM_RING_MFMA32 0x1000     /* with CAPAMD_GEMM_RING_256: keep the 256-row tile on 32x32x16 MFMAs (the bits of the 128-row tile and the 8-wave kernel) */
#define CAPAMD_GEMM_W_CHUNK_MAJOR 0x400   /* W is chunk-major too (with A chunk-major, M, N % 256 == 0, K % 32 == 0, K >= 256: the 4-wave ring kernel) */
int capamd_bert_gemm(const void* A, const void* W, const float* bias, int M, int N, int K, int epilogue,
                     const void* resid, void* out, int dtype /* 0 bf16, 1 fp16 */, void* stream);
/* The same GEMM with LayerNorm folded in (what capamd_bert_maxp_forward runs on BERT-base shapes; exported for the unit
 * tests).  M, N multiples of 256, K >= 128 and a multiple of 64.  `epilogue`: 0 / 1 as above, or 5, plus the layout bits.
 * Consumer side (ln_mu != NULL): A holds UN-normalised rows P, W = W0 . gamma (column-scaled), ln_cs[n] = sum_k W[n][k],
 *   bias[n] = b[n] + sum_k beta[k] W0[n][k]; the epilogue computes rstd_m (acc - mu_m cs_n) + bias_n == LN(P) W0^T + b.
 *   ln_mu, ln_rstd fp32 [M]; ln_mr fp32 [M][2] the same two interleaved.
 * Producer side (epilogue 5, chunk-major out): out = acc + bias + (R - mu_m) rstd_m gamma_n with R = res_src [M, N] chunk-major,
 *   res_mr [M][2] its (mu, rstd), res_gamma [N]; stat_part fp32 [M][N/64][2] receives (sum, sum of squares) of the rounded
 *   output over each 64-column slice (fixed slots, no atomics). */
int capamd_bert_gemm_ln(const void* A, const void* W, const float* bias, int M, int N, int K, int epilogue, const float* ln_mu,
                        const float* ln_rstd, const float* ln_mr, const float* ln_cs, const void* res_src, const float* res_mr,
                        const float* res_gamma, float* stat_part, void* out, int dtype, void* stream);
/* x bf16 [n_passages*S, hidden] -> fused QKV projection (+bias, Q/8) -> softmax(QK^T + pad mask) V.
 * q, k: bf16 [n_passages*S, hidden]; vt: bf16 [n_passages*heads, 64, S]; ctx: bf16 [n_passages*S, hidden];
 * mask int64 [n_passages, S]. */
int capamd_bert_qkv_attention(const void* x, const void* wqkv, const float* bqkv, const int64_t* mask, int n_passages,
                              int S, int hidden, int heads, void* q, void* k, void* vt, void* ctx, int dtype, void* stream);

/* ---- CEDR-KNRM -----------------------------------------------------------------------
 * Replaces CEDRKNRM_Class.forward, capreolus/reranker/CEDRKNRM.py:151-185 (called from CEDRKNRM.test :216-217), BERT-architecture
 * encoders only.  Two calls:
 *  capamd_cedr_passage_features: runs the encoder (same model / workspace as capamd_bert_maxp_forward; the pooler and classifier
 *   pointers of the model may be NULL) over all B*P passages and, for every hidden state listed in simmat_layers (host array,
 *   0 = embedding output .. layers), leaves passage_kernel_sums[layer][passage][kernel][query row] =
 *   sum over the passage's document tokens of the RBF kernels of the masked cosine similarity matrix (masked_simmats /
 *   _cos_simmat / knrm, CEDRKNRM.py:83-130; query rows = sequence positions 1..maxqlen+1), plus cls_rows[passage][hidden] = the
 *   last hidden state's [CLS] row in fp32 (:160).  query_mask0 [B*P][maxqlen+1] (fp32 0/1): for every passage, the query mask
 *   (attention mask & segment 0 at sequence positions 1..maxqlen+1) of the FIRST passage of its document - the mask the
 *   reference applies to all of a document's passages (:123); handing it in per passage lets a caller regroup passages (e.g. by
 *   length, with P = 1) without losing their document.  maxqlen + 1 <= 32, K <= 11.
 *  capamd_cedr_score: the document level (:117-136, 160-185): sums over a document's P passages, clamp / log / 0.01, sum over
 *   the query rows; cls feature (cls_mode 0 none, 1 avg, 2 max); combine = Linear(n_in, 1) (combine_hidden = 0) or
 *   Linear(n_in, combine_hidden) -> Linear(combine_hidden, 1).  features_out (optional) [B][n_in]. */
int capamd_cedr_passage_features(const int64_t* ids, const int64_t* mask, const int64_t* seg, int B, int P, int S,
                                 const capamd_bert_model* m, int64_t passages_per_microbatch, void* workspace, int64_t workspace_bytes,
                                 int maxqlen, const float* query_mask0, const int* simmat_layers, int n_layers, const float* mu,
                                 const float* sigma, int K, float* passage_kernel_sums, float* cls_rows, int* status, void* stream);
int capamd_cedr_score(const float* passage_kernel_sums, const float* cls_rows, int B, int P, int maxqlen, int n_layers, int K, int hidden,
                      int cls_mode, const float* w1, const float* b1, int combine_hidden, const float* w2, const float* b2, float* out,
                      float* features_out, void* stream);

/* ---- ConvKNRM ------------------------------------------------------------------------
 * Replaces ConvKNRM_class.forward, capreolus/reranker/ConvKNRM.py:42-77 (called from ConvKNRM.test :112-116).
 * The Conv1d layers (ConvKNRM.py:24-32) over the frozen embedding table (:17) are folded, once per model, into a table of
 * per-token projections: tables[t][part][f], maxngram*(maxngram+1)/2 parts per token in the order
 *     tap 0 of g = 1..G (bias added) | tap 1 of g = 2..G | tap 2 of g = 3,
 * so that rep_g[j] = part(g,0)[tok j] + part(g,1)[tok j+1] + part(g,2)[tok j+2] (terms beyond the sequence end dropped).
 * conv_w: Conv1d weights of g = 1..maxngram back to back, each [filters][D][g]; conv_b [maxngram][filters].
 * Limits: maxngram <= 3; filters a multiple of 16, 16..128; Q <= 8; K <= 11 kernels; H (hidden width of the two-layer
 * combine, 0 = single Linear) <= 64; L <= 4096.  Ids must lie in [0, V): others set the status bits (nn.Embedding raises).
 * w1 [K*views] (H = 0) or [H][K*views]; feature index = kernel * views + view, view = query_g * G + doc_g (crossmatch) or g. */
int64_t capamd_convknrm_table_bytes(int64_t V, int maxngram, int filters);   /* -1: unsupported geometry */
int capamd_convknrm_pack_tables(const float* emb, int64_t V, int D, int64_t ld, const float* conv_w, const float* conv_b,
                                int maxngram, int filters, float* tables, void* stream);
int capamd_convknrm_forward(const int64_t* q_ids, const int64_t* d_ids, int B, int Q, int L, const float* tables, int64_t V,
                            int maxngram, int filters, int crossmatch, const float* mu, const float* sigma, int K, const float* w1,
                            const float* b1, int H, const float* w2, const float* b2, int score_tanh, float* out, int* status,
                            void* stream);

/* ConvKNRM over whole candidate lists (what PytorchTrainer.predict scores: capreolus/trainer/pytorch.py:310-353 over PredSampler's per-query
 * lists, sampler/__init__.py:222-233): pairs laid out list after list, list l = pairs list_offsets_host[l] .. list_offsets_host[l + 1] (a
 * HOST array of n_lists + 1 entries), every list scored against its FIRST pair's query row (a pair with another row sets
 * CAPAMD_STATUS_LIST_QUERY).  The unigram document view (ConvKNRM.py:46-49 with kernel size 1: a per-token projection) is computed once per
 * DISTINCT token of a list - normalised, multiplied with the list's maxngram x Q query vectors by the same matrix instructions as the
 * per-pair kernel - and looked up per position; the n-gram views of sizes 2 .. maxngram stay per position.  Scores equal
 * capamd_convknrm_forward's bit for bit.  workspace: capamd_convknrm_lists_workspace_bytes(n_lists, V, Q, maxngram, filters) caller-owned
 * bytes (16-byte aligned; per list in flight a flag byte and maxngram x Q floats - rounded up to four - per vocabulary id: 19.6 MB at V =
 * 400,001, Q = 4, maxngram = 3; fewer lists are kept in flight when the buffer is smaller). */
size_t capamd_convknrm_lists_workspace_bytes(int n_lists, int64_t V, int Q, int maxngram, int filters);
int capamd_convknrm_forward_lists(const int64_t* q_ids, const int64_t* d_ids, const int64_t* list_offsets_host, int n_lists, int Q, int L,
                                  const float* tables, int64_t V, int maxngram, int filters, int crossmatch, const float* mu, const float* sigma,
                                  int K, const float* w1, const float* b1, int H, const float* w2, const float* b2, int score_tanh, float* out,
                                  int* status, void* workspace, size_t workspace_bytes, void* stream);

/* ---- PACRR ---------------------------------------------------------------------------
 * Replaces PACRR_class.forward + PACRRConvMax2dModule.forward, capreolus/reranker/PACRR.py:42-78 (called from PACRR.test
 * :114-118): similarity matrix as in KNRM -> per n-gram size Conv2d(1 -> nfilters, ng x ng) on the zero-padded matrix,
 * ReLU, max over filters, kmax largest values over ALL document positions -> optional idf channel (softmax over the
 * query's raw idf values) -> three linear layers with `nonlinearity` (0 none, 1 relu, 2 tanh) in between.
 * conv_w: Conv2d weights of the n-gram modules (mingram..maxgram) back to back, each [nfilters][ng][ng]; conv_b
 * [n_ngrams][nfilters]; w1 [combine][Q * (n_ngrams * kmax + use_idf)], w2 [combine][combine], w3 [combine].
 * Limits: Q <= 8, L <= 1024, maxgram <= 3, kmax <= 4, nfilters <= 256, combine <= 128 (CAPAMD_ERR_ARG beyond them).  Q <= 5
 * with nfilters <= 32 (the reference defaults) runs the convolutions on the matrix pipe (f16 hi/lo split, fp32 accumulate,
 * ~1e-6 of the fp32 form); other geometries an fp32 VALU kernel.  Status bits as for capamd_knrm_forward. */
int capamd_pacrr_forward(const int64_t* q_ids, const int64_t* d_ids, const float* idf, int B, int Q, int L, const float* packed,
                         int64_t V, int D, int mingram, int maxgram, int nfilters, int kmax, const float* conv_w,
                         const float* conv_b, int use_idf, int combine, int nonlinearity, const float* w1, const float* b1,
                         const float* w2, const float* b2, const float* w3, const float* b3, float* out, int* status,
                         void* stream);

/* One PACRR training step on the device (reference PACRR.py:42-78 under trainer/pytorch.py:93-108, the loss
 * reranker/common.py:96-103, torch.optim.Adam): similarity matrices -> n-gram Conv2d / ReLU / max over filters / k-max with the winners'
 * coordinates (capamd_pacrr_convmax_forward) -> ONE workgroup: features + idf softmax, linear1 / linear2 / linear3 with `nonlinearity`
 * (0 none, 1 relu, 2 tanh), the pairwise loss (0 hinge / 1 softmax), the backward through the layers, their gradients summed in document
 * order, Adam on them -> convolution gradients (capamd_pacrr_convmax_backward) -> Adam on the convolutions.  Six launches, no autograd.
 * q_ids int64 [2 B, Q] (the batch's queries twice), d_ids int64 [2 B, L] (positive documents, then negative), idf fp32 [2 B, Q] (use_idf);
 * ptrs: HOST array of 3 P device pointers, P = 2 n_ngrams + 6 - ngrams.{i}.conv.weight / .bias (mingram..maxgram), linear1.weight / .bias,
 * linear2.weight / .bias, linear3.weight / .bias, then their exp_avg, then their exp_avg_sq (null moments: not trained).
 * Limits: those of capamd_pacrr_convmax_*, combine <= 128, and the batch's activations must fit one workgroup's LDS:
 * H X + H^2 + H + 2 B X + 8 B H + 5 B <= 36,864 floats (H = combine, X = Q (n_ngrams kmax + use_idf)); CAPAMD_ERR_ARG beyond.
 * workspace: capamd_pacrr_train_step_workspace_floats(...) floats, 16-byte aligned. */
size_t capamd_pacrr_train_step_workspace_floats(int B, int Q, int L, int mingram, int maxgram, int nfilters, int kmax);
int capamd_pacrr_train_step(const int64_t* q_ids, const int64_t* d_ids, const float* idf, int B, int Q, int L, const float* packed, int64_t V, int D,
                            int mingram, int maxgram, int nfilters, int kmax, int use_idf, int combine, int nonlinearity, float* const* ptrs,
                            int loss_type, float step_size, float one_minus_beta1, float beta2, float eps, float bc2_sqrt, float* loss_out,
                            float* workspace, size_t workspace_floats, int* status, void* stream);

/* ---- ConvKNRM's trainable n-gram convolutions, forward and backward (ConvKNRM's training step) --------------
 * Replaces, for training, the convolution stack of ConvKNRM_class.forward, capreolus/reranker/ConvKNRM.py:42-51 - embeddings(ids) ->
 * permute -> ConstantPad1d((0, g - 1), 0) -> Conv1d(D -> F, kernel g) -> permute, for g = 1..G, on the query and on the document -
 * and its gradient under the reference trainer's loss.backward() (trainer/pytorch.py:96-107): no [B, D, L] embedding tensor, no padded
 * copies, no library convolution - the kernels gather the table's rows and run the g taps as fp32 matrix products on
 * v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate).  The table is frozen (ConvKNRM.py:17): gradients go to the weights and biases only.
 * q_ids int64 [N, Q], d_ids int64 [N, L] (pad = 0; ids outside [0, V) set CAPAMD_STATUS_DOC_ID_RANGE and count as zero rows);
 * emb fp32 [V, D] row-major (nn.Embedding.weight), 16-byte aligned, D % 4 == 0, D <= 316; conv_w / conv_b: HOST arrays of G device
 * pointers, Conv1d weight [F][D][g] and bias [F] of n-gram size g = index + 1; G <= 4; F % 4 == 0, F <= 256.
 * forward:  qrep fp32 [N, G, Q, F], drep fp32 [N, G, L, F].  A 128-position tile without a single real token is written as ZEROS (the
 *           reference's value there is the bias plus the pad row's projection): capamd_kernel_pool_* masks those positions.
 * backward: dqrep / ddrep (16-byte aligned) -> dconv_w / dconv_b (HOST arrays of device pointers, the weights' layouts), overwritten.
 *           Rows of ddrep at pad positions are taken to be zero (what capamd_kernel_pool_backward writes there).  Deterministic: the
 *           position slices are summed in a fixed order.
 * workspace: capamd_ngram_conv_workspace_floats(D, G, F, backward) fp32 values, 16-byte aligned (CAPAMD_ERR_WORKSPACE below that). */
size_t capamd_ngram_conv_workspace_floats(int D, int G, int F, int backward);
int capamd_ngram_conv_forward(const int64_t* q_ids, const int64_t* d_ids, int N, int Q, int L, const float* emb, int64_t V, int D,
                              const float* const* conv_w, const float* const* conv_b, int G, int F, float* qrep, float* drep,
                              float* workspace, size_t workspace_floats, int* status, void* stream);
int capamd_ngram_conv_backward(const int64_t* q_ids, const int64_t* d_ids, int N, int Q, int L, const float* emb, int64_t V, int D, int G,
                               int F, const float* dqrep, const float* ddrep, float* const* dconv_w, float* const* dconv_b,
                               float* workspace, size_t workspace_floats, int* status, void* stream);

/* One ConvKNRM training step on the device (reference ConvKNRM.py:42-77 under trainer/pytorch.py:93-108, the loss
 * reranker/common.py:96-103, torch.optim.Adam): convolutions (capamd_ngram_conv_forward) -> kernel pooling (capamd_kernel_pool_forward) ->
 * single-Linear combine (+ tanh), pairwise loss (0 hinge / 1 softmax), its backward and Adam on the Linear in ONE workgroup -> pooling
 * backward -> the partial results' reductions + Adam on the kernels' mu / sigma -> convolution weight gradients
 * (capamd_ngram_conv_backward) -> Adam on the convolutions.  Eleven launches, no autograd, no ATen node.
 * q_ids int64 [2 B, Q] (the batch's queries twice), d_ids int64 [2 B, L] (the positive documents, then the negative ones);
 * ptrs: HOST array of 3 P device pointers, P = 2 K + 2 G + 2 - kernels.kernels.{k}.mu (K), .sigma (K), convs.{g}.0.weight / .bias (G pairs),
 * combine.0.weight [K V], combine.0.bias [1], then their exp_avg, then their exp_avg_sq (null moments: a parameter that is not trained).
 * step_size = lr / (1 - beta1^t), bc2_sqrt = sqrt(1 - beta2^t) from the host (double).  loss_out: fp32 [1].  B <= 512; other limits as
 * the two kernel families'.  workspace: capamd_convknrm_train_step_workspace_floats(...) floats, 16-byte aligned. */
size_t capamd_convknrm_train_step_workspace_floats(int B, int Q, int L, int D, int G, int F, int K, int crossmatch);
int capamd_convknrm_train_step(const int64_t* q_ids, const int64_t* d_ids, int B, int Q, int L, const float* emb, int64_t V, int D, int G, int F,
                               int K, int crossmatch, float* const* ptrs, int scoretanh, int loss_type, float step_size, float one_minus_beta1,
                               float beta2, float eps, float bc2_sqrt, float* loss_out, float* workspace, size_t workspace_floats, int* status,
                               void* stream);

/* ---- differentiable kernel pooling over dense n-gram representations (ConvKNRM's training step) ---------
 * The part of ConvKNRM_class.forward between its trainable n-gram convolutions and `combine`, capreolus/reranker/ConvKNRM.py:53-76
 * (StackedSimilarityMatrix common.py:195-221 + RbfKernelBank common.py:224-250), forward and backward - so that the reference
 * trainer's loss.backward() (trainer/pytorch.py:96-107) reaches the convolutions through HIP kernels, not ATen ops.
 * qrep fp32 [B, GQ, Q, F] / drep fp32 [B, GD, L, F]: the GQ / GD n-gram views of query and document (F % 4 == 0, F <= 256);
 * q_ids / d_ids: the token ids (pad = 0: masked positions); crossmatch != 0: every (query view, document view) pair, V = GQ GD
 * views (v = gq GD + gd), else the matching ones (GQ == GD, V = GD).  (crossmatch ? GQ : 1) * Q <= 24, K <= 16.
 * A document's real positions are split into at most C = capamd_kernel_pool_chunks(L) chunks (1..8 slots; 160 real positions per chunk forward, 64 backward; unused slots hold zeros), one workgroup per
 * (pair, document view, chunk); pad positions never reach the similarity loop (a masked entry is the constant 0: a closed form).
 * forward:  feat fp32 [B, K V] (feature k V + v, the reference's kernels.reshape(B, K V, Q, L) order); ksum fp32 [B, GD, T, K] and
 *           rowsum fp32 [B, GD, T] (T = (crossmatch ? GQ : 1) Q) are what the backward needs of it; chunk_sums: B GD C T (K + 1) fp32
 *           values of scratch (the chunks' partial sums, added up in their order by a second launch).
 * backward: gfeat fp32 [B, K V] -> dq_part fp32 [B, GD, C, T, F] (the caller sums over C, and over GD the blocks that share a query
 *           view: with crossmatch view gq = t / Q of every gd; without, block gd holds query view gd), dd fp32 [B, GD, L, F] (zero
 *           rows at pad positions), dmu_part / dsigma_part fp32 [B GD C, K] (summed over their first axis by the caller). */
int capamd_kernel_pool_chunks(int L);
int capamd_kernel_pool_forward(const float* qrep, const float* drep, const int64_t* q_ids, const int64_t* d_ids, int B, int GQ, int GD, int Q,
                               int L, int F, int crossmatch, const float* mu, const float* sigma, int K, float* feat, float* ksum,
                               float* rowsum, float* chunk_sums, void* stream);
int capamd_kernel_pool_backward(const float* qrep, const float* drep, const int64_t* q_ids, const int64_t* d_ids, int B, int GQ, int GD, int Q,
                                int L, int F, int crossmatch, const float* mu, const float* sigma, int K, const float* gfeat,
                                const float* ksum, const float* rowsum, float* dq_part, float* dd, float* dmu_part, float* dsigma_part,
                                void* stream);

/* ---- PACRR's trainable convolution stage with its gradient (PACRR's training step) --------------------------
 * PACRRConvMax2dModule.forward, capreolus/reranker/PACRR.py:68-78, for every n-gram size mingram..maxgram at once: zero padding,
 * Conv2d(1 -> nfilters, ng x ng) over the similarity matrix sim fp32 [B, Q, L] (capamd_similarity_matrix), ReLU, max over the
 * filters, the kmax largest values over the document - and, for the reference trainer's loss.backward() (trainer/pytorch.py:96-107),
 * the coordinates of every value.  conv_w / conv_b as in capamd_pacrr_forward.  Q <= 8, L <= 1024, maxgram <= 3, kmax <= 4.
 * forward:  top fp32 / pos int32 / filt int32 [B, Q, n_ngrams * kmax]: value, document position, filter (-1: a ReLU zero).
 * backward: gtop fp32 [B, Q, n_ngrams * kmax] -> dconv_w, dconv_b (laid out like conv_w / conv_b; summed in a fixed order). */
int capamd_pacrr_convmax_forward(const float* sim, int B, int Q, int L, int mingram, int maxgram, int nfilters, int kmax, const float* conv_w,
                                 const float* conv_b, float* top, int32_t* pos, int32_t* filt, void* stream);
int capamd_pacrr_convmax_backward(const float* sim, int B, int Q, int L, int mingram, int maxgram, int nfilters, int kmax, const float* gtop,
                                  const int32_t* pos, const int32_t* filt, float* dconv_w, float* dconv_b, void* stream);

/* ---- ranking of scored candidate lists ---------------------------------------------
 * What follows the scoring call in the reference, kept on the device:
 *   - score.astype(np.float16)                         capreolus/trainer/pytorch.py:346-348
 *   - Searcher.write_trec_run's per-query stable sort  capreolus/searcher/__init__.py:48-58
 *   - trec_eval's ndcg_cut_k behind evaluator.py:55-85 (score desc, ties by docid desc, gain = level, log2 discount)
 * scores fp32 [total]; offsets int64 [n_queries + 1] (CSR: query q owns scores[offsets[q] .. offsets[q+1])),
 * max_candidates >= the longest list (<= 16384).
 * capamd_rank_candidates: out_idx int32 [n_queries, k] = positions inside the query's list in run-file order
 *   (rounded score descending, equal scores in list order; -1 beyond the list), out_f16 uint16 [n_queries, k] = the
 *   rounded scores' fp16 bits.
 * capamd_ndcg_cut (k <= 256): rel int32 [total] relevance level of each candidate (0 = unjudged), tie int32 [total] =
 *   the candidate's rank in docid-descending order inside its query (a permutation of 0..n-1), idcg fp64 [n_queries] =
 *   the ideal DCG@k from the query's qrels (host side); out fp64 [n_queries] = nDCG@k (0 when idcg is 0).
 * Status bits: CAPAMD_STATUS_SCORE_NAN, CAPAMD_STATUS_TIE_RANGE. */
int capamd_rank_candidates(const float* scores, const int64_t* offsets, int n_queries, int max_candidates, int k,
                           int32_t* out_idx, uint16_t* out_f16, int* status, void* stream);
int capamd_ndcg_cut(const float* scores, const int64_t* offsets, const int32_t* rel, const int32_t* tie, const double* idcg,
                    int n_queries, int max_candidates, int k, double* out, int* status, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CAPREOLUS_AMD_H */
