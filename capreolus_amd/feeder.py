"""Device-resident candidate lists.

The reference feeds prediction one sample at a time: `PredSampler.generate_samples` calls
`extractor.id2vec(qid, docid)` per pair, the DataLoader collates, `predict` copies every batch to the device
(capreolus/sampler/__init__.py:222-233, trainer/pytorch.py:334-342).  With the kernels at tens of millions
of pairs per second that host path, not the scoring, is the bottleneck (410 MB of int64 ids per 64,000 pairs).

A `CandidateStore` tokenises every distinct query and document of a run ONCE (through the same
`id2vec`-style callables), uploads two int32 tables, and a batch becomes a pair of row-index vectors; the
kernels read ids through the indirection (`capamd_*_forward_indexed`).
"""
import numpy as np
import torch


class CandidateStore:
    def __init__(self, device):
        self.device = torch.device(device)
        self.qrow, self.drow = {}, {}
        self._q, self._idf, self._d = [], [], []
        self.q_table = self.idf_table = self.d_table = None

    def add_query(self, qid, query_ids, query_idf=None):
        if qid not in self.qrow:
            self.qrow[qid] = len(self._q)
            self._q.append(np.asarray(query_ids))
            self._idf.append(np.zeros(len(query_ids), np.float32) if query_idf is None else np.asarray(query_idf, np.float32))
        return self.qrow[qid]

    def add_doc(self, docid, doc_ids):
        if docid not in self.drow:
            self.drow[docid] = len(self._d)
            self._d.append(np.asarray(doc_ids))
        return self.drow[docid]

    @classmethod
    def from_id2vec(cls, device, qid_to_docids, id2vec):
        """`id2vec(qid, docid)` -> dict with "query", "posdoc", "query_idf" (the EmbedText contract, embedtext.py:128-162)."""
        st = cls(device)
        for qid, docids in qid_to_docids.items():
            for docid in docids:
                if qid in st.qrow and docid in st.drow:
                    continue
                v = id2vec(qid, docid)
                st.add_query(qid, v["query"], v.get("query_idf"))
                st.add_doc(docid, v["posdoc"])
        return st.finalize()

    @staticmethod
    def _int32_block(rows, name):
        """[n, L] int32 from a list of equally long id rows (any integer dtype)"""
        block = np.stack(rows)
        if block.dtype != np.int32:
            if block.min() < -2 ** 31 or block.max() >= 2 ** 31:
                raise ValueError(f"{name} term ids do not fit int32")
            block = block.astype(np.int32)
        return block

    CHUNK = 512       # rows stacked and narrowed at a time: the int64 rows of a 64,000-document run are 410 MB - stacked whole and then
                      # narrowed they were read and written three times over (0.3 s of a 0.35 s first `predict`).  (Narrowing in a
                      # background thread under the caller's loop was tried: np.stack over a list holds the interpreter lock - 4x slower.)

    def finalize(self):
        self.q_table = torch.as_tensor(self._int32_block(self._q, "query")).to(self.device)
        self.idf_table = torch.as_tensor(np.stack(self._idf)).to(self.device)
        d = np.empty((len(self._d), len(self._d[0])), dtype=np.int32)
        for lo in range(0, len(self._d), self.CHUNK):
            d[lo:lo + self.CHUNK] = self._int32_block(self._d[lo:lo + self.CHUNK], "document")
        self.d_table = torch.as_tensor(d).to(self.device)
        return self

    def pairs(self, qid_to_docids):
        """(keys, pair_q, pair_d): the run's (qid, docid) pairs in PredSampler order and their table rows on the device."""
        keys = [(q, d) for q, ds in qid_to_docids.items() for d in ds]
        pq = torch.as_tensor(np.fromiter((self.qrow[q] for q, _ in keys), dtype=np.int32, count=len(keys))).to(self.device)
        pd = torch.as_tensor(np.fromiter((self.drow[d] for _, d in keys), dtype=np.int32, count=len(keys))).to(self.device)
        return keys, pq, pd
