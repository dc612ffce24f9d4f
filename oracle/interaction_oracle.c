/*
 * oracle/interaction_oracle.c — CPU restatement of the KNRM / DRMM scoring path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under capreolus_amd/ imports, links or executes this file;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, and only as the
 * checker / the timed CPU baseline.  The product path is the HIP library and fails loudly
 * without it.
 *
 * Parity pinning: this restatement is checked against golden vectors produced by running the
 * reference nn.Modules themselves (tests/golden/make_golden.py imports them from the reference
 * tree in the build container); see tests/test_oracle_golden.py.  The reference's own test
 * suite holds no known-answer vectors for this path.
 *
 * Each function cites the reference lines (under capreolus/) it restates.
 *
 * Arithmetic order.  The similarity front end (pack, dot, cosine) reproduces, operation for
 * operation, the order documented in capreolus_amd/csrc/interaction.h — 16 "lane" partial
 * fma chains combined by a balanced tree — so that GPU and oracle agree BIT FOR BIT on every
 * similarity and therefore exactly on DRMM's integer bin counts.  Everything downstream of the
 * similarities (kernel pooling sums, logs, MLPs) is done here in double precision with libm,
 * i.e. more accurately than either the reference or the GPU; those are compared with a
 * tolerance.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define GROUP 16

static int nv_for_dim(int D) { return (D + 1 + 63) / 64; }
int64_t oracle_row_stride(int D) { return 64 * nv_for_dim(D); }

/* balanced tree over 16 lane partials: ((p0+p1)+(p2+p3)) + ((p4+p5)+(p6+p7)) ... */
static float tree16(const float* p) {
  float q[4];
  for (int j = 0; j < 4; ++j) q[j] = (p[4 * j] + p[4 * j + 1]) + (p[4 * j + 2] + p[4 * j + 3]);
  return (q[0] + q[1]) + (q[2] + q[3]);
}

/* create_emb_layer + the norm half of cosine_similarity_matrix (reranker/common.py:279-288, :162-163):
 * packed row = [row, 0..., |row|_2 + 1e-9]. */
void oracle_pack(const float* emb, int64_t V, int D, int64_t ld, float* packed) {
  const int64_t RS = oracle_row_stride(D);
  const int NV = (int)(RS / 64);
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < V; ++r) {
    const float* src = emb + r * ld;
    float* dst = packed + r * RS;
    memset(dst, 0, sizeof(float) * (size_t)RS);
    memcpy(dst, src, sizeof(float) * (size_t)D);
    float part[GROUP];
    for (int l = 0; l < GROUP; ++l) {
      float s = 0.f;
      for (int i = 0; i < NV; ++i)
        for (int c = 0; c < 4; ++c) {
          const int f = (i * 16 + l) * 4 + c;
          const float v = f < D ? src[f] : 0.f;
          s = fmaf(v, v, s);
        }
      part[l] = s;
    }
    dst[RS - 1] = sqrtf(tree16(part)) + 1e-9f;
  }
}

/* <E[q], E[d]> in the documented lane order; the den slot (last float) is excluded. */
static float dot_rows(const float* q, const float* d, int NV) {
  float part[GROUP];
  const int last = 64 * NV - 1;
  for (int l = 0; l < GROUP; ++l) {
    float p = 0.f;
    for (int i = 0; i < NV; ++i)
      for (int c = 0; c < 4; ++c) {
        const int f = (i * 16 + l) * 4 + c;
        const float qv = f == last ? 0.f : q[f];
        p = fmaf(d[f], qv, p);
      }
    part[l] = p;
  }
  return tree16(part);
}

/* SimilarityMatrix.forward (reranker/common.py:170-182): exact_match_matrix (:155-158) on
 * clamp(max=0) ids + cosine_similarity_matrix (:160-167) on clamp(min=0) ids, padding removed
 * (:149-153).  Returns 0 or a bitmask of {1: doc id >= V, 2: query id >= V}. */
static float sim_one(int64_t qid, int64_t did, const float* packed, int64_t RS, int NV) {
  if (did > 0) {
    if (qid <= 0) return 0.f;
    const float* qr = packed + qid * RS;
    const float* dr = packed + did * RS;
    return dot_rows(qr, dr, NV) / (qr[RS - 1] * dr[RS - 1]);
  }
  return (did < 0 && qid == did) ? 1.f : 0.f;
}

int oracle_simmat(const int64_t* q_ids, const int64_t* d_ids, int B, int Q, int L, const float* packed, int64_t V, int D,
                  float* sim_out) {
  const int64_t RS = oracle_row_stride(D);
  const int NV = (int)(RS / 64);
  int err = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(| : err)
  for (int b = 0; b < B; ++b)
    for (int q = 0; q < Q; ++q) {
      int64_t qid = q_ids[(int64_t)b * Q + q];
      if (qid >= V) { err |= 2; qid = 0; }
      for (int j = 0; j < L; ++j) {
        int64_t did = d_ids[(int64_t)b * L + j];
        if (did >= V) { err |= 1; did = 0; }
        sim_out[((int64_t)b * Q + q) * L + j] = sim_one(qid, did, packed, RS, NV);
      }
    }
  return err;
}

/* KNRM_class.forward (reranker/KNRM.py:39-55) with RbfKernel.forward (reranker/common.py:232-234):
 *   K_k = exp(-0.5*(sim-mu_k)*(sim-mu_k)/sigma_k/sigma_k); sum over ALL L positions (KNRM.py:50);
 *   mask = (sum_j sim != 0) (KNRM.py:51); where(mask, log(sum + 1e-6), 0) summed over q (:52-53);
 *   combine (KNRM.py:27-34): hidden == 0 -> Linear(K,1); else Linear(K,hidden),Tanh,Linear(hidden,1); optional Tanh. */
int oracle_knrm(const int64_t* q_ids, const int64_t* d_ids, int B, int Q, int L, const float* packed, int64_t V, int D,
                const float* mu, const float* sigma, int K, const float* w1, const float* b1, int hidden, const float* w2,
                const float* b2, int scoretanh, float* out) {
  const int64_t RS = oracle_row_stride(D);
  const int NV = (int)(RS / 64);
  int err = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(| : err)
  for (int b = 0; b < B; ++b) {
    double F[64];
    for (int k = 0; k < K; ++k) F[k] = 0.0;
    for (int q = 0; q < Q; ++q) {
      int64_t qid = q_ids[(int64_t)b * Q + q];
      if (qid >= V) { err |= 2; qid = 0; }
      double S[64], rowsum = 0.0;
      for (int k = 0; k < K; ++k) S[k] = 0.0;
      for (int j = 0; j < L; ++j) {
        int64_t did = d_ids[(int64_t)b * L + j];
        if (did >= V) { err |= 1; did = 0; }
        const float sim = sim_one(qid, did, packed, RS, NV);
        rowsum += sim;
        for (int k = 0; k < K; ++k) {
          const double adj = (double)sim - (double)mu[k];
          S[k] += exp(-0.5 * adj * adj / (double)sigma[k] / (double)sigma[k]);
        }
      }
      if (rowsum != 0.0)
        for (int k = 0; k < K; ++k) F[k] += log(S[k] + 1e-6);
    }
    double sc;
    if (hidden > 0) {
      sc = b2[0];
      for (int h = 0; h < hidden; ++h) {
        double a = b1[h];
        for (int k = 0; k < K; ++k) a += (double)w1[h * K + k] * F[k];
        sc += (double)w2[h] * tanh(a);
      }
    } else {
      sc = b1[0];
      for (int k = 0; k < K; ++k) sc += (double)w1[k] * F[k];
    }
    if (scoretanh) sc = tanh(sc);
    out[b] = (float)sc;
  }
  return err;
}

/* DRMM_class.forward (reranker/DRMM.py:101-116):
 *   _hist_map (:41-81): sim += 1e7 on pad doc positions (:57); cum_i = #(sim < edge_i) (:62-65);
 *     hist_nbins = #(0.999 < sim < 1.001) (:66); adjacent differences (:68-69); +1 (:71);
 *     NH / LCH / CH (:72-79).
 *   ffw (:25): tanh(w2 . tanh(W1 h + b1) + b2) per query term.
 *   _term_gate (:83-99): softmax_q( gate(q) + (1-qmask)*-1e7 ), gate = w*idf (IDF) or w.E[q] (TV).
 *   output_layer (:34, :114).
 * hist_type 0 CH, 1 NH, 2 LCH; gate_type 0 IDF, 1 TV.  A negative query id returns error bit 4
 * (the reference raises IndexError at DRMM.py:109). */
/* back end of one query term from its raw bin counts (DRMM.py:71-79 histogram types, :25 ffw): z_q */
static double drmm_term_z(const int32_t* cnt, int NB, int hist_type, const float* w1, const float* b1, int nodes, const float* w2,
                          const float* b2) {
  double h[128], hs = 0.0;
  for (int i = 0; i < NB; ++i) { h[i] = (double)cnt[i] + 1.0; hs += h[i]; }
  if (hist_type == 1) for (int i = 0; i < NB; ++i) h[i] /= hs;
  else if (hist_type == 2) for (int i = 0; i < NB; ++i) h[i] = log(h[i]);
  double o = b2[0];
  for (int n = 0; n < nodes; ++n) {
    double a = b1[n];
    for (int i = 0; i < NB; ++i) a += (double)w1[n * NB + i] * h[i];
    o += (double)w2[n] * tanh(a);
  }
  return tanh(o);
}
/* gate logit of one query term (DRMM.py:83-99), rounded through fp32 as the reference computes it */
static double drmm_gate_logit(int64_t qid, float idf, int gate_type, const float* gate_w, const float* emb_raw, int64_t ld, int D) {
  double gl;
  if (gate_type == 0) gl = (double)gate_w[0] * (double)idf;
  else {
    gl = 0.0;
    const float* e = emb_raw + qid * ld; /* un-normalised query embedding (DRMM.py:109) */
    for (int c = 0; c < D; ++c) gl += (double)gate_w[c] * (double)e[c];
  }
  if (qid == 0) gl += -1e7;
  /* the reference adds in fp32: w*idf + (-1e7) rounds to exactly -1e7 for |w*idf| < 0.5 */
  return (double)(float)gl;
}
/* softmax gate over the query terms and the output layer (DRMM.py:97, 112-114) */
static float drmm_combine(const double* z, const double* glogit, int Q, const float* out_w, const float* out_b) {
  double m = glogit[0];
  for (int q = 1; q < Q; ++q) if (glogit[q] > m) m = glogit[q];
  double den = 0.0, num = 0.0;
  for (int q = 0; q < Q; ++q) { const double e = exp(glogit[q] - m); den += e; num += e * z[q]; }
  return (float)((double)out_w[0] * (num / den) + (double)out_b[0]);
}

int oracle_drmm(const int64_t* q_ids, const int64_t* d_ids, const float* idf, int B, int Q, int L, const float* packed,
                int64_t V, int D, const float* edges, int nbins, int hist_type, int gate_type, const float* gate_w,
                const float* emb_raw, int64_t ld, const float* w1, const float* b1, int nodes, const float* w2,
                const float* b2, const float* out_w, const float* out_b, float* out, int32_t* counts_out) {
  const int64_t RS = oracle_row_stride(D);
  const int NV = (int)(RS / 64);
  const int NB = nbins + 1;
  int err = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(| : err)
  for (int b = 0; b < B; ++b) {
    double z[64], glogit[64];
    for (int q = 0; q < Q; ++q) {
      int64_t qid = q_ids[(int64_t)b * Q + q];
      if (qid >= V) { err |= 2; qid = 0; }
      if (qid < 0) { err |= 4; qid = 0; }
      int32_t cnt[128];
      for (int i = 0; i < NB; ++i) cnt[i] = 0;
      for (int j = 0; j < L; ++j) {
        int64_t did = d_ids[(int64_t)b * L + j];
        if (did >= V) { err |= 1; did = 0; }
        if (did == 0) continue; /* +1e7: below no edge, outside (0.999, 1.001) */
        const float sim = sim_one(qid, did, packed, RS, NV);
        int bin = 0;
        while (bin < nbins && !(sim < edges[bin])) ++bin; /* first edge with sim < edge */
        if (bin < nbins) cnt[bin] += 1;
        if (sim > 0.999f && sim < 1.001f) cnt[nbins] += 1;
      }
      if (counts_out)
        for (int i = 0; i < NB; ++i) counts_out[((int64_t)b * Q + q) * NB + i] = cnt[i];
      z[q] = drmm_term_z(cnt, NB, hist_type, w1, b1, nodes, w2, b2);
      glogit[q] = drmm_gate_logit(qid, idf[(int64_t)b * Q + q], gate_type, gate_w, emb_raw, ld, D);
    }
    out[b] = drmm_combine(z, glogit, Q, out_w, out_b);
  }
  return err;
}

/* The back end of oracle_drmm on GIVEN raw bin counts [B][Q][nbins+1] (what `_hist_map` counts before the +1, DRMM.py:62-70):
 * histogram type -> ffw -> term gate -> output layer.  Fed with the reference's own counts it must reproduce the reference's
 * scores on EVERY pair, including those where the front end's cos(a, a) < 1.0 coin flips differently. */
int oracle_drmm_from_counts(const int32_t* counts, const int64_t* q_ids, const float* idf, int B, int Q, int64_t V, int D, int nbins,
                            int hist_type, int gate_type, const float* gate_w, const float* emb_raw, int64_t ld, const float* w1,
                            const float* b1, int nodes, const float* w2, const float* b2, const float* out_w, const float* out_b,
                            float* out) {
  const int NB = nbins + 1;
  int err = 0;
  for (int b = 0; b < B; ++b) {
    double z[64], glogit[64];
    for (int q = 0; q < Q; ++q) {
      int64_t qid = q_ids[(int64_t)b * Q + q];
      if (qid >= V) { err |= 2; qid = 0; }
      if (qid < 0) { err |= 4; qid = 0; }
      z[q] = drmm_term_z(counts + ((int64_t)b * Q + q) * NB, NB, hist_type, w1, b1, nodes, w2, b2);
      glogit[q] = drmm_gate_logit(qid, idf[(int64_t)b * Q + q], gate_type, gate_w, emb_raw, ld, D);
    }
    out[b] = drmm_combine(z, glogit, Q, out_w, out_b);
  }
  return err;
}

/* DRMMTKS_class.forward (reranker/DRMMTKS.py:50-64): cos_mat = SimilarityMatrix(query, doc) (:55); topk over the last
 * dimension, pads included (:56); ffw = tanh(Linear(topk, 1)) per query term (:22, :57); IDF term gate (:31-48);
 * output layer (:61). */
int oracle_drmmtks(const int64_t* q_ids, const int64_t* d_ids, const float* idf, int B, int Q, int L, const float* packed, int64_t V,
                   int D, int topk, const float* gate_w, const float* ffw_w, const float* ffw_b, const float* out_w, const float* out_b,
                   float* out) {
  const int64_t RS = oracle_row_stride(D);
  const int NV = (int)(RS / 64);
  int err = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(| : err)
  for (int b = 0; b < B; ++b) {
    double z[64], gl[64];
    float* sims = (float*)malloc(sizeof(float) * (size_t)L);
    for (int q = 0; q < Q; ++q) {
      int64_t qid = q_ids[(int64_t)b * Q + q];
      if (qid >= V) { err |= 2; qid = 0; }
      for (int j = 0; j < L; ++j) {
        int64_t did = d_ids[(int64_t)b * L + j];
        if (did >= V) { err |= 1; did = 0; }
        sims[j] = sim_one(qid, did, packed, RS, NV);
      }
      double acc = ffw_b[0];
      for (int r = 0; r < topk; ++r) { /* selection of the r-th largest */
        int best = r;
        for (int j = r + 1; j < L; ++j) if (sims[j] > sims[best]) best = j;
        const float t = sims[r]; sims[r] = sims[best]; sims[best] = t;
        acc += (double)ffw_w[r] * (double)sims[r];
      }
      z[q] = tanh(acc);
      double g = (double)gate_w[0] * (double)idf[(int64_t)b * Q + q];
      if (qid == 0) g += -1e7;
      gl[q] = (double)(float)g;
    }
    free(sims);
    double m = gl[0];
    for (int q = 1; q < Q; ++q) if (gl[q] > m) m = gl[q];
    double den = 0.0, num = 0.0;
    for (int q = 0; q < Q; ++q) { const double e = exp(gl[q] - m); den += e; num += e * z[q]; }
    out[b] = (float)((double)out_w[0] * (num / den) + (double)out_b[0]);
  }
  return err;
}

/* PACRR forward (row N4), restating PACRR_class.forward / PACRRConvMax2dModule.forward of the reference
 * (capreolus/reranker/PACRR.py:42-78) on top of the same SimilarityMatrix as KNRM:
 *   simmat [Q, L] (cosine + OOV exact match, pads zeroed, common.py:170-182)
 *   for ng = mingram..maxgram: zero-pad right/bottom by ng-1 (:61), Conv2d(1 -> nfilters, ng x ng) + bias (:64), ReLU (:72), max over
 *     the filters (:73), the kmax largest values along the document axis (:74), over ALL L positions (pads included)
 *   optional idf channel: softmax over the Q raw idf values (:48-50)
 *   per query term the channels [ng1 top-1..k, ng2 top-1..k, ..., idf], flattened query-major (:51-52), then
 *   Linear -> nonlin -> Linear -> nonlin -> Linear (:30-40).  nonlin: 0 none, 1 relu, 2 tanh.
 * conv_w: the Conv2d weights of the n-gram modules back to back, each [nfilters][ng][ng]; conv_b [n_ngrams][nfilters]. */
static double pacrr_act(double x, int nonlin) { return nonlin == 1 ? (x > 0 ? x : 0) : (nonlin == 2 ? tanh(x) : x); }

int oracle_pacrr(const int64_t* q_ids, const int64_t* d_ids, const float* idf, int B, int Q, int L, const float* packed, int64_t V, int D,
                 int mingram, int maxgram, int nfilters, int kmax, const float* conv_w, const float* conv_b, int use_idf, int C,
                 const float* w1, const float* b1, const float* w2, const float* b2, const float* w3, const float* b3, int nonlin,
                 float* out) {
  const int64_t RS = oracle_row_stride(D);
  const int NV = (int)(RS / 64);
  const int n_ng = maxgram - mingram + 1, qts = n_ng * kmax + (use_idf ? 1 : 0);
  int err = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(| : err)
  for (int b = 0; b < B; ++b) {
    const int LP = L + maxgram, QP = Q + maxgram;
    float* sim = (float*)calloc((size_t)QP * LP, sizeof(float));   /* zero-padded on the right / bottom */
    float* top = (float*)malloc(sizeof(float) * (size_t)L);
    double* feat = (double*)calloc((size_t)Q * qts, sizeof(double));
    double h1[256], h2[256];
    for (int q = 0; q < Q; ++q) {
      int64_t qid = q_ids[(int64_t)b * Q + q];
      if (qid >= V) { err |= 2; qid = 0; }
      for (int j = 0; j < L; ++j) {
        int64_t did = d_ids[(int64_t)b * L + j];
        if (did >= V) { err |= 1; did = 0; }
        sim[(size_t)q * LP + j] = sim_one(qid, did, packed, RS, NV);
      }
    }
    const float* w = conv_w;
    for (int gi = 0; gi < n_ng; ++gi) {
      const int ng = mingram + gi;
      for (int q = 0; q < Q; ++q) {
        for (int j = 0; j < L; ++j) {
          float best = 0.f; /* ReLU output is >= 0 */
          for (int f = 0; f < nfilters; ++f) {
            float acc = conv_b[gi * nfilters + f];
            for (int a = 0; a < ng; ++a)
              for (int c = 0; c < ng; ++c) acc = fmaf(w[(f * ng + a) * ng + c], sim[(size_t)(q + a) * LP + j + c], acc);
            if (acc > best) best = acc;
          }
          top[j] = best;
        }
        for (int r = 0; r < kmax; ++r) { /* the r-th largest */
          int bi = r;
          for (int j = r + 1; j < L; ++j) if (top[j] > top[bi]) bi = j;
          const float t = top[r]; top[r] = top[bi]; top[bi] = t;
          feat[q * qts + gi * kmax + r] = top[r];
        }
      }
      w += (size_t)nfilters * ng * ng;
    }
    if (use_idf) {
      double m = idf[(int64_t)b * Q];
      for (int q = 1; q < Q; ++q) if (idf[(int64_t)b * Q + q] > m) m = idf[(int64_t)b * Q + q];
      double den = 0.0;
      for (int q = 0; q < Q; ++q) den += exp((double)idf[(int64_t)b * Q + q] - m);
      for (int q = 0; q < Q; ++q) feat[q * qts + qts - 1] = exp((double)idf[(int64_t)b * Q + q] - m) / den;
    }
    const int nin = Q * qts;
    for (int c = 0; c < C; ++c) {
      double s = b1[c];
      for (int i = 0; i < nin; ++i) s += (double)w1[c * nin + i] * feat[i];
      h1[c] = pacrr_act(s, nonlin);
    }
    for (int c = 0; c < C; ++c) {
      double s = b2[c];
      for (int i = 0; i < C; ++i) s += (double)w2[c * C + i] * h1[i];
      h2[c] = pacrr_act(s, nonlin);
    }
    double s = b3[0];
    for (int i = 0; i < C; ++i) s += (double)w3[i] * h2[i];
    out[b] = (float)s;
    free(sim); free(top); free(feat);
  }
  return err;
}

/* ---- ConvKNRM: ConvKNRM_class.forward, capreolus/reranker/ConvKNRM.py:42-77 ---------------------------
 *   embeddings (common.py:279-288; ids outside [0, V) make nn.Embedding raise -> error bits) of the query and the document;
 *   per n-gram size g = 1..maxngram: right zero padding by g-1 and Conv1d(D -> F, g) over the sequence (:46-49, built at :24-32):
 *       rep_g[j][f] = bias_g[f] + sum_c sum_d W_g[f][d][c] * E[tok[j + c]][d]       (terms with j + c beyond the sequence are 0)
 *   StackedSimilarityMatrix (common.py:195-221) of every query view with every document view (crossmatch, :52-55; view = a * G + b)
 *   or of equal sizes only (:57-59): cos = a.b / ((|a| + 1e-9)(|b| + 1e-9)), 0 where the query or document token AT the position
 *   is the pad id 0; RbfKernelBank (common.py:232-234, 249-250): exp(-0.5 (s - mu)^2 / sigma / sigma); sum over the document
 *   (:71), log(. + 1e-6) where the row's similarities do not sum to 0, else 0 (:72-73); sum over the query (:74); feature index
 *   = kernel * VIEWS + view (:63-64); combine (:35-41): Linear(K * VIEWS, 1), or Linear(., H) -> tanh -> Linear(H, 1); optional tanh.
 * conv_w: Conv1d weights of g = 1..G back to back, each [F][D][g]; conv_b [G][F].  H = 0: single layer (w1 [K*VIEWS], b1 [1]).
 * Accumulations are in double (this is the checker, not the measured path). */
int oracle_convknrm(const int64_t* q_ids, const int64_t* d_ids, int B, int Q, int L, const float* emb, int64_t V, int D, int G, int F,
                    const float* conv_w, const float* conv_b, int crossmatch, const float* mu, const float* sigma, int K,
                    const float* w1, const float* b1, int H, const float* w2, const float* b2, int score_tanh, float* out) {
  const int views = crossmatch ? G * G : G;
  int err = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(| : err)
  for (int b = 0; b < B; ++b) {
    const int64_t* qi = q_ids + (int64_t)b * Q;
    const int64_t* di = d_ids + (int64_t)b * L;
    float* qrep = (float*)malloc(sizeof(float) * (size_t)G * Q * F);
    float* drep = (float*)malloc(sizeof(float) * (size_t)G * L * F);
    float* qn = (float*)malloc(sizeof(float) * (size_t)G * Q);
    float* dn = (float*)malloc(sizeof(float) * (size_t)G * L);
    double* feat = (double*)calloc((size_t)K * views, sizeof(double));
    for (int side = 0; side < 2; ++side) {
      const int64_t* ids = side ? di : qi;
      const int n = side ? L : Q;
      float* rep = side ? drep : qrep;
      float* nrm = side ? dn : qn;
      const float* w = conv_w;
      for (int g = 1; g <= G; ++g) {
        for (int j = 0; j < n; ++j) {
          double ss = 0.0;
          for (int f = 0; f < F; ++f) {
            double s = conv_b[(g - 1) * F + f];
            for (int c = 0; c < g && j + c < n; ++c) {
              int64_t id = ids[j + c];
              if (id < 0 || id >= V) { err |= side ? 1 : 2; id = 0; }
              const float* e = emb + id * D;
              for (int d = 0; d < D; ++d) s += (double)w[((size_t)f * D + d) * g + c] * e[d];
            }
            const float r = (float)s;
            rep[((size_t)(g - 1) * n + j) * F + f] = r;
            ss += (double)r * r;
          }
          nrm[(g - 1) * n + j] = (float)sqrt(ss);
        }
        w += (size_t)F * D * g;
      }
    }
    for (int v = 0; v < views; ++v) {
      const int ga = crossmatch ? v / G : v, gb = crossmatch ? v % G : v;
      for (int q = 0; q < Q; ++q) {
        double rowsum = 0.0;
        double* ks = (double*)calloc((size_t)K, sizeof(double));
        const float* a = qrep + ((size_t)ga * Q + q) * F;
        const float an = qn[ga * Q + q] + 1e-9f;
        for (int j = 0; j < L; ++j) {
          float s = 0.f;
          if (qi[q] != 0 && di[j] != 0) {
            const float* bb = drep + ((size_t)gb * L + j) * F;
            double dot = 0.0;
            for (int f = 0; f < F; ++f) dot += (double)a[f] * bb[f];
            s = (float)dot / (an * (dn[gb * L + j] + 1e-9f));
          }
          rowsum += s;
          for (int k = 0; k < K; ++k) {
            const double adj = (double)s - mu[k];
            ks[k] += exp(-0.5 * adj * adj / sigma[k] / sigma[k]);
          }
        }
        if (rowsum != 0.0)
          for (int k = 0; k < K; ++k) feat[k * views + v] += log((double)(float)ks[k] + 1e-6);
        free(ks);
      }
    }
    const int nin = K * views;
    double s;
    if (H == 0) {
      s = b1[0];
      for (int i = 0; i < nin; ++i) s += (double)w1[i] * (float)feat[i];
    } else {
      s = b2[0];
      for (int h = 0; h < H; ++h) {
        double t = b1[h];
        for (int i = 0; i < nin; ++i) t += (double)w1[(size_t)h * nin + i] * (float)feat[i];
        s += (double)w2[h] * tanh(t);
      }
    }
    if (score_tanh) s = tanh(s);
    out[b] = (float)s;
    free(qrep); free(drep); free(qn); free(dn); free(feat);
  }
  return err;
}
