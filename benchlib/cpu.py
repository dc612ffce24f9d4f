"""cpu_baseline legs (rank 0, N = 1): the C oracle (OpenMP) and the reference's ATen op sequence on the host cores, the BASELINE
configs[0] stand-in.  The oracle is the checker and the CPU baseline here - never the thing measured as `value`."""
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch


def cpu_baseline(args, model, oracle_sample, leg):
    """The CPU oracle (oracle/interaction_oracle.c, OpenMP over pairs) timed on this box's host cores on a bounded sample of the
    same workload (and checked against the scores the GPU just produced for that sample); the reference's ATen op sequence
    (oracle/torch_port.py) swept over thread counts and batch sizes; for KNRM the BASELINE configs[0] stand-in."""
    from oracle import torch_port

    cores = os.cpu_count() or 1
    n = args.cpu_pairs or min(leg.n_pairs, 2000 * max(1, cores // 4))
    run, (q, d, idf, emb_h, sd), err = oracle_sample(n)
    D = leg.D
    run()  # warm
    t0 = time.perf_counter()
    reps = 0
    while True:
        run()
        reps += 1
        if time.perf_counter() - t0 > 6.0 or reps >= 5:
            break
    c_rate = n * reps / (time.perf_counter() - t0)

    # ATen port (what the reference executes on CPU), swept: the reference leaves the thread count to torch's default (= cores),
    # which is far from the best on a many-core host
    te = torch.as_tensor(emb_h)
    tq, td, tidf = torch.as_tensor(q), torch.as_tensor(d), torch.as_tensor(idf)
    nb = min(n, 4000)
    if model == "knrm":
        mu, sigma = (x.cpu() for x in leg.m.kernels.stacked())
        tw, tb = torch.as_tensor(sd["combine.0.weight"]), torch.as_tensor(sd["combine.0.bias"])

        def trun(lo, hi):
            return torch_port.knrm(te, tq[lo:hi], td[lo:hi], mu, sigma, tw, tb)
    else:
        ts = {k: torch.as_tensor(v) for k, v in sd.items()}

        def trun(lo, hi):
            return torch_port.drmm(te, tq[lo:hi], td[lo:hi], tidf[lo:hi], 29, "LCH", "IDF", ts["gates.weight"],
                                   ts["ffw.0.weight"], ts["ffw.0.bias"], ts["ffw.2.weight"], ts["ffw.2.bias"],
                                   ts["output_layer.weight"], ts["output_layer.bias"])
    sweep = []
    threads = sorted({t for t in (1, 8, 16, 32, 64, 128, cores) if t <= cores})
    with torch.no_grad():
        for bs in (32, 256, 1000):          # 32 = the reference's default evalbatch (trainer/pytorch.py:24-25, 334)
            for t in threads:
                torch.set_num_threads(t)
                trun(0, min(bs, nb))
                t0 = time.perf_counter()
                done = 0
                while time.perf_counter() - t0 < 0.7:
                    for lo in range(0, nb, bs):
                        trun(lo, min(lo + bs, nb))
                        done += min(lo + bs, nb) - lo
                        if time.perf_counter() - t0 > 0.7:
                            break
                sweep.append({"threads": t, "batch": bs, "pairs_per_s": done / (time.perf_counter() - t0)})
    best = max(sweep, key=lambda r: r["pairs_per_s"])
    default32 = next(r for r in sweep if r["threads"] == cores and r["batch"] == 32)
    res = {
        "value": c_rate,
        "unit": "pairs/s",
        "cores": cores,
        "kind": "port",
        "sample": f"first {n} pairs of the last timed batch, oracle/interaction_oracle.c with OpenMP over pairs ({reps} repetitions); the timed GPU "
                  f"scores of these pairs agree with it to {err:.1e}",
        "aten_port_value": best["pairs_per_s"],
        "aten_port_threads": best["threads"],
        "aten_port_batch": best["batch"],
        "aten_port_note": "oracle/torch_port.py (the reference's ATen op sequence) at the best of the swept (threads, batch) settings; "
                          "aten_port_reference_default = torch's default thread count at the reference's default evalbatch 32",
        "aten_port_reference_default": default32["pairs_per_s"],
        "aten_port_sweep": sweep,
    }
    if model == "knrm":
        torch.set_num_threads(best["threads"])
        res.update(config0_standin(te, tq, td, mu, sigma, tw, tb, best["threads"]))
        res.update(config0_gpu(leg))
    torch.set_num_threads(cores)
    return res


def config0_standin(te, tq, td, mu, sigma, w, b, threads):
    """BASELINE.json configs[0] ("KNRM on NFCorpus, niters=1, CUDA_VISIBLE_DEVICES=''") cannot run offline; SURVEY.md §8d's stand-in: the
    reference trainer's defaults on synthetic data of the same shapes - 16 training steps (itersize 512 / batch 32: score() on a
    positive and a negative document, pairwise hinge loss, Adam on mu, sigma and the combine layer; trainer/pytorch.py:76-122) and a
    predict pass over 325 queries x 100 documents at evalbatch 32 (:310-353; dev threshold 100, task/rerank.py:22) - through the
    reference's ATen op sequence on the host cores."""
    from oracle import torch_port

    n = tq.shape[0]
    mu_p, sg_p = torch.nn.Parameter(mu.clone()), torch.nn.Parameter(sigma.clone())
    w_p, b_p = torch.nn.Parameter(w.clone()), torch.nn.Parameter(b.clone())
    opt = torch.optim.Adam([mu_p, sg_p, w_p, b_p], lr=1e-3)
    t0 = time.perf_counter()
    for s in range(16):
        lo = (s * 64) % max(1, n - 64)
        pos = torch_port.knrm(te, tq[lo:lo + 32], td[lo:lo + 32], mu_p, sg_p, w_p, b_p)
        neg = torch_port.knrm(te, tq[lo:lo + 32], td[lo + 32:lo + 64], mu_p, sg_p, w_p, b_p)
        loss = torch.clamp(1.0 - (pos - neg), min=0).mean()
        loss.backward()
        opt.step()
        opt.zero_grad()
    train_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    total, done = 325 * 100, 0
    with torch.no_grad():
        while done < total:
            lo = done % max(1, n - 32)
            torch_port.knrm(te, tq[lo:lo + 32], td[lo:lo + 32], mu, sigma, w, b)
            done += 32
    pred_s = time.perf_counter() - t0
    return {"config0_s": train_s + pred_s, "config0_train_s": train_s, "config0_predict_s": pred_s,
            "config0_note": f"BASELINE configs[0] stand-in on {threads} threads: 16 training steps of batch 32 (pos + neg forward, hinge loss, backward, Adam) "
                            f"+ predict over 325 x 100 pairs at evalbatch 32, reference ATen op sequence (oracle/torch_port.py); "
                            f"predict alone = {total / pred_s:.0f} pairs/s"}


def config0_gpu(leg):
    """The same BASELINE configs[0] stand-in through this engine on the GPU: 16 training steps of batch 32 with the reranker's own
    `score()` (capamd_knrm_features: pooled features + their mu / sigma derivatives in one kernel, the combine layer under autograd),
    hinge loss, Adam; then the 325 x 100 predict pass - once as 1,016 `test()` calls of 32 pairs (what the reference trainer issues at
    evalbatch 32) and once as the single coalesced call capreolus_amd.trainer.PytorchTrainer.predict makes of them."""
    rr, m = leg.rr, leg.m
    b = leg.batches[0]
    q, d, idf = b["query"], b["posdoc"], b["query_idf"]
    n = q.shape[0]
    saved = {k: v.clone() for k, v in m.state_dict().items() if "embedding" not in k}
    params = [p for p in m.parameters() if p.requires_grad]
    opt = torch.optim.Adam(params, lr=1e-3)

    from capreolus_amd import engine

    def train16():
        # what PytorchTrainer.single_train_iteration does by default: the reranker's fused step (capamd_knrm_train_step: score(pos),
        # score(neg), hinge loss, backward and Adam in two launches, the kernels' status read once at the end)
        m.train()
        with engine.deferred_status(leg.ctx.dev):
            for s in range(16):
                lo = (s * 64) % max(1, n - 64)
                batch = {"query": q[lo:lo + 32], "posdoc": d[lo:lo + 32], "negdoc": d[lo + 32:lo + 64], "query_idf": idf[lo:lo + 32]}
                if rr.fused_train_step(batch, opt) is None:
                    pos, neg = rr.score(batch)
                    loss = torch.clamp(1.0 - (pos - neg), min=0).mean()
                    loss.backward()
                    opt.step()
                    opt.zero_grad()
        m.eval()
        torch.cuda.synchronize()

    train16()                      # first call: module load, allocator
    t0 = time.perf_counter()
    train16()
    train_s = time.perf_counter() - t0
    total = 325 * 100
    with torch.no_grad():
        rr.test({"query": q[:32], "posdoc": d[:32], "query_idf": idf[:32]})
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        done = 0
        while done < total:
            lo = done % max(1, n - 32)
            rr.test({"query": q[lo:lo + 32], "posdoc": d[lo:lo + 32], "query_idf": idf[lo:lo + 32]})
            done += 32
        torch.cuda.synchronize()
        pred32_s = time.perf_counter() - t0
        t0 = time.perf_counter()
        rr.test({"query": q[:total], "posdoc": d[:total], "query_idf": idf[:total]})
        torch.cuda.synchronize()
        pred1_s = time.perf_counter() - t0
    m.load_state_dict(saved, strict=False)
    return {"config0_gpu_s": train_s + pred1_s, "config0_gpu_train_s": train_s, "config0_gpu_predict_s": pred1_s, "config0_gpu_predict_evalbatch32_s": pred32_s,
            "config0_gpu_note": "the same stand-in through this engine on the GPU (batches already in HBM): 16 training steps of batch 32 via reranker.fused_train_step() "
                                "(the trainer's default: features of positives and negatives, hinge loss, backward, Adam in two launches per step) + the 325 x 100 predict as "
                                "ONE scoring call (what this engine's trainer makes of the evalbatch-32 loader; config0_gpu_predict_evalbatch32_s = the same "
                                "pairs as 1,016 separate test() calls of 32, each checking the status word)"}

