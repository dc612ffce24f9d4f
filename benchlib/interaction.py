"""The KNRM / DRMM legs: candidate lists resident in HBM, the timed steps (list route or per-pair kernels), the per-pass figures and
rooflines, the PMC traffic passes (child runs of bench.py under rocprofv3 --pmc)."""
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))      # the repository root (bench.py lives there)
from benchlib.common import Ctx1, F32_PEAK_TFLOPS, GATHER_CEILING_GBS, L2_PEAK_GBS, HBM_PEAK_GBS, KERNEL_EVAL_PEAK_G, KERNEL_EVAL_PEAK_SOURCE, KERNEL_VARIANT, SIMS_LIMITER, SIMS_PIPE, _tables, algorithmic_bytes_per_pair, collective_info, default_queries, kernel_of, repeated_timed_loop, table
from benchlib.cpu import cpu_baseline


class InteractionLeg:
    """KNRM / DRMM over `nb` distinct batches of `n_queries` x `docs` candidate lists on one GPU."""

    def __init__(self, args, ctx, model, V, uniform, n_queries, nb, seed0, zero_idf=False):
        from types import SimpleNamespace

        from capreolus_amd import engine, synthetic
        from capreolus_amd.reranker import DRMM, KNRM

        self.args, self.ctx, self.model, self.V, self.uniform = args, ctx, model, V, uniform
        dev = ctx.dev
        self.Q, self.L, self.D = getattr(args, "qlen", 4), 800, args.dim
        self.n_pairs = n_queries * args.docs
        self.emb = table(dev, V, self.D)
        self.batches = []
        for b in range(nb):
            batch = synthetic.make_candidate_list_torch(n_queries, args.docs, V, dev, seed=seed0 + 1000 * b, maxqlen=self.Q, maxdoclen=self.L,
                                                        uniform_ids=uniform)
            if model == "drmm":
                batch["query"] = batch["query"].clamp(min=0)
            if zero_idf:        # EmbedText's default behaviour (no idf computed: all zeros) - SURVEY 8(d), configs[2]'s second run
                batch["query_idf"] = torch.zeros_like(batch["query_idf"])
            self.batches.append(batch)
        torch.manual_seed(0)
        stub = SimpleNamespace(embeddings=np.zeros((2, self.D), dtype=np.float32))
        self.rr = rr = (KNRM if model == "knrm" else DRMM)({}, stub)
        self.m = m = rr.build_model().to(dev).eval()
        m.embedding = torch.nn.Embedding.from_pretrained(self.emb, freeze=True)
        w = m.embedding.weight
        self.packed = packed = m._packed.get(w)
        self.row_stride = packed.numel() // V
        self.out = torch.empty(self.n_pairs, dtype=torch.float32, device=dev)
        self.cur_out = self.out        # where launch_one writes (a per-stream buffer when consecutive steps alternate over streams)
        launch = args.launch_docs or self.n_pairs
        self.slices = [(i, min(i + launch, self.n_pairs)) for i in range(0, self.n_pairs, launch)]
        # whole candidate lists (csrc/lists.hip) unless asked otherwise; launches of single lists (38.6 M pairs/s as a list against 48.6 M
        # pair by pair; two lists per launch: 54.3 against 49.3) and the HBM-bound leg (uniform ids: a list's documents share almost no
        # vocabulary) stay on the per-pair kernels
        self.lists = not args.per_pair and (not uniform or getattr(args, "force_lists", False)) and launch % args.docs == 0 and launch >= 2 * args.docs
        D = self.D
        if model == "knrm":
            mu, sigma = m.kernels.stacked()
            w1, b1 = m.combine[0].weight.detach().contiguous(), m.combine[0].bias.detach()
            if args.resident:  # one query row per query, one document row per candidate, int32
                tabs = [(b["query"][:: args.docs].to(torch.int32).contiguous(), b["posdoc"].to(torch.int32).contiguous()) for b in self.batches]
                pq = torch.arange(self.n_pairs, device=dev, dtype=torch.int32) // args.docs
                pd = torch.arange(self.n_pairs, device=dev, dtype=torch.int32)

                stores = [SimpleNamespace(q_table=t[0], d_table=t[1]) for t in tabs]

                def launch_one(bi, lo, hi):
                    if self.lists:      # the store's lists as lists (index pairs into the int32 tables)
                        engine.knrm_forward_lists(np.arange(0, hi - lo + 1, args.docs), packed, V, D, mu, sigma, w1, b1, store=stores[bi], pair_q=pq[lo:hi],
                                                  pair_d=pd[lo:hi], out=self.cur_out[lo:hi], check=False)
                    else:
                        engine.knrm_forward_indexed(tabs[bi][0], tabs[bi][1], pq[lo:hi], pd[lo:hi], packed, V, D, mu, sigma, w1, b1, out=self.cur_out[lo:hi], check=False)
            elif self.lists:
                def launch_one(bi, lo, hi):      # the step's candidate lists (args.docs documents per query) as lists
                    b = self.batches[bi]
                    engine.knrm_forward_lists(np.arange(0, hi - lo + 1, args.docs), packed, V, D, mu, sigma, w1, b1, query=b["query"][lo:hi],
                                              doc=b["posdoc"][lo:hi], out=self.cur_out[lo:hi], check=False)
            else:
                def launch_one(bi, lo, hi):
                    b = self.batches[bi]
                    engine.knrm_forward(b["query"][lo:hi], b["posdoc"][lo:hi], packed, V, D, mu, sigma, w1, b1, out=self.cur_out[lo:hi], check=False)
        else:
            edges = m._bin_edges(dev)
            gw = m.gates.weight.detach().contiguous().view(-1)
            f0w, f0b = m.ffw[0].weight.detach().contiguous(), m.ffw[0].bias.detach()
            f2w, f2b = m.ffw[2].weight.detach().contiguous().view(-1), m.ffw[2].bias.detach()
            ow, ob = m.output_layer.weight.detach().view(-1), m.output_layer.bias.detach()

            def launch_one(bi, lo, hi):
                b = self.batches[bi]
                if self.lists:
                    engine.drmm_forward_lists(np.arange(0, hi - lo + 1, args.docs), b["query_idf"][lo:hi], packed, V, D, edges, "LCH", "IDF", gw, w, f0w, f0b,
                                              f2w, f2b, ow, ob, query=b["query"][lo:hi], doc=b["posdoc"][lo:hi], out=self.cur_out[lo:hi], check=False)
                else:
                    engine.drmm_forward(b["query"][lo:hi], b["posdoc"][lo:hi], b["query_idf"][lo:hi], packed, V, D, edges, "LCH", "IDF", gw, w, f0w, f0b,
                                        f2w, f2b, ow, ob, out=self.cur_out[lo:hi], check=False)
        self.launch_one = launch_one
        n_side = min(args.launch_streams, len(self.slices)) if len(self.slices) > 1 else 1
        self.side = [torch.cuda.Stream(device=dev) for _ in range(n_side)] if n_side > 1 else []
        # multi-GPU: the step's all-gather runs asynchronously on RCCL's stream from a snapshot of the scores, under the NEXT step's
        # scoring (two snapshots / destinations in rotation); every gather is waited for before its buffers are reused and before the
        # timed region closes
        self.gathered = [torch.empty(self.n_pairs * ctx.world, dtype=torch.float32, device=dev) for _ in range(2)] if ctx.use_dist else None
        self.snap = [torch.empty(self.n_pairs, dtype=torch.float32, device=dev) for _ in range(2)] if ctx.use_dist else None
        self.pending = [None, None]
        self.last_batch = 0
        # --step-streams S: consecutive STEPS (independent batches of candidate lists) go round-robin over S HIP streams, each with its
        # own score buffer and its own whole-list workspace (engine._lists_workspace is per stream): a step's HBM-bound mark pass runs
        # under the previous step's VALU-bound pooling pass, and every pass's tail is filled by the other stream's workgroups
        ns = getattr(args, "step_streams", 1)
        if ns <= 0:      # auto: calls of up to 128 lists leave tails and fixed passes a second stream fills (64 lists: +9 %, 128: +3-5 %; 250: -1 %)
            ns = 2 if n_queries <= 128 else 1
        self.step_side = [torch.cuda.Stream(device=dev) for _ in range(ns)] if ns > 1 and len(self.slices) == 1 and self.lists else []
        self.step_out = [torch.empty(self.n_pairs, dtype=torch.float32, device=dev) for _ in self.step_side]
        self.snap_done = [None] * len(self.step_side)      # (multi-GPU: the main stream's snapshot of a buffer, awaited before the buffer is rewritten)

    def capture(self):
        """one HIP graph per batch: the step's launches (fork over the side streams, join) replayed with a single host call"""
        self.graphs = []
        if len(self.slices) == 1 or self.args.no_graph:
            return
        for bi in range(len(self.batches)):
            self._launch_all(bi)                       # eager once: module load, workspace
        torch.cuda.synchronize()
        for bi in range(len(self.batches)):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._launch_all(bi)
            self.graphs.append(g)

    def step(self, i):
        bi = i % len(self.batches)
        if self.step_side:
            k = i % len(self.step_side)
            st = self.step_side[k]
            if self.snap_done[k] is not None:
                st.wait_event(self.snap_done[k])
            with torch.cuda.stream(st):
                self.cur_out = self.step_out[k]
                self.launch_one(bi, 0, self.n_pairs)
                self.cur_out = self.out
            self.last_side = k
            if self.ctx.use_dist:      # the step's gather from a snapshot on the main stream; only THIS buffer's next step waits for it
                main = torch.cuda.current_stream()
                done = torch.cuda.Event()
                done.record(st)
                main.wait_event(done)
                j = i & 1
                if self.pending[j] is not None:
                    self.pending[j].wait()
                self.snap[j].copy_(self.step_out[k])
                self.snap_done[k] = torch.cuda.Event()
                self.snap_done[k].record(main)
                self.pending[j] = self.ctx.dist.all_gather_into_tensor(self.gathered[j], self.snap[j], async_op=True)
                self.last_gather = j
            self.last_batch = bi
            return
        if getattr(self, "graphs", None):
            self.graphs[bi].replay()
        else:
            self._launch_all(bi)
        if self.ctx.use_dist:
            k = i & 1
            if self.pending[k] is not None:
                self.pending[k].wait()
            self.snap[k].copy_(self.out)
            self.pending[k] = self.ctx.dist.all_gather_into_tensor(self.gathered[k], self.snap[k], async_op=True)
            self.last_gather = k
        self.last_batch = bi

    def drain(self):
        if self.step_side:
            main = torch.cuda.current_stream()
            for st in self.step_side:
                main.wait_stream(st)
        for k in range(2):
            if self.pending[k] is not None:
                self.pending[k].wait()
                self.pending[k] = None

    def _launch_all(self, bi):
        if self.side:      # independent candidate lists: round-robin over side streams, joined before the step ends
            main = torch.cuda.current_stream()
            for st in self.side:
                st.wait_stream(main)
            from capreolus_amd import engine

            with engine.concurrent_launches():             # per-call flag: small launches share the chip, use the occupancy variant
                for k, (lo, hi) in enumerate(self.slices):
                    with torch.cuda.stream(self.side[k % len(self.side)]):
                        self.launch_one(bi, lo, hi)
            for st in self.side:
                main.wait_stream(st)
        else:
            for lo, hi in self.slices:
                self.launch_one(bi, lo, hi)

    def run(self, warmup, steps, repeats=1):
        from capreolus_amd import engine

        self.capture()
        if self.step_side:
            for st in self.step_side:          # module load / workspace allocation of every stream outside the timed region
                with torch.cuda.stream(st):
                    self.launch_one(0, 0, self.n_pairs)
            torch.cuda.synchronize()
        elapsed, dev_s, self.repeats = repeated_timed_loop(self.ctx, self.step, warmup, steps, self.drain if (self.ctx.use_dist or self.step_side) else None, repeats)
        if self.step_side:
            self.out.copy_(self.step_out[self.last_side])      # (the last step's scores: what the oracle check and the gather check compare)
            torch.cuda.synchronize()
        engine.status_word(self.ctx.dev).raise_if_set()
        assert torch.isfinite(self.out).all()
        if self.ctx.use_dist:
            r = self.ctx.rank
            assert torch.equal(self.gathered[self.last_gather][r * self.n_pairs:(r + 1) * self.n_pairs], self.out)
        return elapsed, dev_s

    def lists_pass_times(self, steps=5):
        """The passes of the whole-list route one by one: HIP events on the stream the kernels run on, recorded by the library between the
        passes of `steps` more steps after the timed loop (csrc/capamd_profiling.h: capamd_debug_lists_timing; the events sit between
        launches, so a pass's figure includes its launch gap - the five add up to the step).  Returns ms per step of
        (clear, mark, query, sims, pool)."""
        from capreolus_amd import _lib

        with _lib.profiling_build() as lib:      # the -DCAPAMD_PROFILING build of the same kernels: the product library has no hooks
            for lo, hi in self.slices:
                self.launch_one(0, lo, hi)       # (module load of the second library)
            torch.cuda.synchronize()
            lib.capamd_debug_lists_timing(1)
            try:
                for i in range(steps):
                    for lo, hi in self.slices:
                        self.launch_one(i % len(self.batches), lo, hi)
                    self.last_batch = i % len(self.batches)       # (`out` now holds this batch's scores: what check_against_oracle compares)
                ms = (ctypes.c_double * 8)()
                groups = lib.capamd_debug_lists_timing_read(ms)
            finally:
                lib.capamd_debug_lists_timing(0)
            torch.cuda.synchronize()
        return [m / steps for m in ms][:5] if groups else None

    def bytes_requested_per_pair(self):
        """What the kernel asks the memory system for: the id rows (int64), one packed table row (row_stride floats: the embedding,
        its norm, padding to whole 128-byte lines) per DISTINCT in-vocabulary document term and per query term, the score.  A term the
        document repeats is gathered once and weighted by its count; pads and OOV terms are scored in closed form without a gather
        (DESIGN.md §3.1).  Returns (bytes per pair, mean non-pad terms per document, mean distinct terms per document)."""
        nonpad = distinct = 0
        for b in self.batches:
            srt = torch.sort(b["posdoc"], dim=1).values
            nonpad += int((srt > 0).sum().item())
            distinct += int((srt[:, 0] > 0).sum().item()) + int(((srt[:, 1:] != srt[:, :-1]) & (srt[:, 1:] > 0)).sum().item())
        n = len(self.batches) * self.n_pairs
        nonpad, distinct = nonpad / n, distinct / n
        return (self.L * 8 + self.Q * 8 + (distinct + self.Q) * self.row_stride * 4 + 4 + (4 * self.Q if self.model == "drmm" else 0), nonpad,
                distinct)

    def bytes_requested_per_pair_lists(self):
        """The whole-list route's requests per pair (csrc/lists.hip): the id row twice (mark pass, pooling pass), one byte-map store and one
        table lookup per real position (KNRM: the four similarities, 16 B; DRMM: the four bins, 4 B), and the list's distinct terms' packed
        rows (gathered once per LIST) spread over its documents."""
        docs = self.args.docs
        nonpad = rows = 0
        for b in self.batches:
            d = b["posdoc"].view(-1, docs * self.L)
            nonpad += int((d > 0).sum().item())
            for i in range(d.shape[0]):
                u = torch.unique(d[i])
                rows += int((u > 0).sum().item())
        n = len(self.batches) * self.n_pairs
        entry = 16 if self.model == "knrm" else 4
        return 2 * self.L * 8 + self.Q * 8 + (nonpad / n) * (1 + entry) + (rows / n) * (self.row_stride * 4 + entry) + 4, rows / (n / docs)

    def check_against_oracle(self, n):
        """The scores the timed loop left in `out` (its last step's batch) against the C oracle on the first n pairs; returns what the
        CPU baseline needs to time the same sample."""
        from oracle import cpu as oracle

        assert self.V <= 400001, "the oracle sample is drawn on the BASELINE table"
        b = self.batches[self.last_batch]
        q, d, idf = (b[k][:n].cpu().numpy() for k in ("query", "posdoc", "query_idf"))
        emb_h = self.emb.cpu().numpy()
        packed = oracle.pack(emb_h)
        sd = {k: v.detach().cpu().numpy() for k, v in self.m.state_dict().items() if "embedding" not in k}
        if self.model == "knrm":
            mu, sigma = (x.cpu().numpy() for x in self.m.kernels.stacked())

            def run():
                return oracle.knrm(q, d, packed, self.D, mu, sigma, sd["combine.0.weight"], sd["combine.0.bias"])[0]
        else:
            edges = torch.linspace(-1, 1, 30)[1:].numpy()

            def run():
                return oracle.drmm(q, d, idf, packed, self.D, edges, "LCH", "IDF", sd["gates.weight"], emb_h, sd["ffw.0.weight"],
                                   sd["ffw.0.bias"], sd["ffw.2.weight"], sd["ffw.2.bias"], sd["output_layer.weight"],
                                   sd["output_layer.bias"])[0]
        want = run()
        got = self.out[:n].cpu().numpy()
        err = float(np.abs(got - want).max() / max(1e-6, np.abs(want).max()))
        # (CAPAMD_BENCH_NO_CHECK: ablation builds of the library, scripts/build_variant_obj.sh - their scores are wrong on purpose)
        assert err <= 2e-5 or os.environ.get("CAPAMD_BENCH_NO_CHECK"), f"{self.model}: the timed scores differ from the oracle's by {err}"
        return run, (q, d, idf, emb_h, sd), err


def pmc_traffic(args, model, route="per_pair_hbm"):
    """HBM-side bytes from the PMC counters, measured inside this invocation and collected as MI355X_MICROARCH.md (section HBM) prescribes -
    FETCH_SIZE and WRITE_SIZE in separate `rocprofv3 --pmc` passes (own child runs of this script, counters only, no trace domains),
    bytes = (FETCH_SIZE x 2 + WRITE_SIZE) x 1024: gfx950 tallies the 128-byte requests of wide (16 B/lane) coalesced reads at 64 B.
      route "per_pair_hbm": per launch of the per-pair kernel on the HBM-bound leg (uniform ids over the --roofline-vocab table)
      route "lists":        per CALL of the whole-list route on the headline configuration: every kernel of the call summed (the byte-map
                            lists_clear, lists_mark, lists_query, lists_sims, the pooling kernel)
    Returns (bytes or None, how / why not)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    if any(k.startswith(("ROCPROFILER", "ROCP_", "ROCTRACER")) for k in os.environ):
        return None, "not measured: this run is itself under a profiler"
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "not measured: rocprofv3 not found"
    child = [sys.executable, os.path.join(ROOT, "bench.py"), "--model", model, "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-also", "--no-roofline-leg",
             "--no-pmc-traffic", "--batches", "2", "--dim", str(args.dim), "--step-streams", "1"]
    if route == "per_pair_hbm":
        child += ["--uniform-ids", "--vocab", str(args.roofline_vocab)]

        def mine(name):
            return "forward_kernel" in name or "stream_kernel" in name

        def unit(name):
            return mine(name)
    else:
        child += ["--vocab", str(args.vocab), "--queries", str(args.queries or default_queries(model)), "--docs", str(args.docs), "--no-pass-times"]

        def mine(name):
            return "lists_" in name or "fillBuffer" in name

        def unit(name):
            return "lists_mark_kernel" in name
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "CAPAMD_FORCE_DIST")}
    env["TMPDIR"] = "/tmp"
    kb = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        with tempfile.TemporaryDirectory(dir="/tmp") as td:
            try:
                subprocess.run([exe, "--pmc", counter, "--output-format", "csv", "-d", td, "-o", "c", "--"] + child, cwd="/tmp", env=env, timeout=300,
                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
            except (OSError, subprocess.TimeoutExpired) as e:
                return None, f"not measured: rocprofv3 --pmc {counter} failed ({type(e).__name__})"
            total, units = 0.0, 0
            for f in glob.glob(os.path.join(td, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r["Counter_Name"] != counter:
                        continue
                    if mine(r["Kernel_Name"]):
                        total += float(r["Counter_Value"])
                    if unit(r["Kernel_Name"]):
                        units += 1
            if not units:
                return None, f"not measured: the rocprofv3 --pmc {counter} pass returned no rows for the kernel"
            kb[counter] = (total / units, units)
    what = "launches of the per-pair kernel" if route == "per_pair_hbm" else "calls of the list route (all its kernels summed)"
    return (kb["FETCH_SIZE"][0] * 2 + kb["WRITE_SIZE"][0]) * 1024, (
        f"measured in this invocation: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in two child runs ({kb['FETCH_SIZE'][1]} / {kb['WRITE_SIZE'][1]} {what} "
        f"sampled, {kb['FETCH_SIZE'][0]:.0f} / {kb['WRITE_SIZE'][0]:.0f} KB each); bytes = (FETCH_SIZE x 2 + WRITE_SIZE) x 1024 per MI355X_MICROARCH.md section HBM "
        "(gfx950 tallies 128-B requests of wide coalesced reads at 64 B); memory-side requests of the L2s, Infinity-Cache hits included: an upper bound on HBM bytes")


def lists_roofline(model, headline, hbm_leg, n_pairs, dev_s, compulsory, traffic, traffic_src):
    """`roofline` of a line whose timed steps run the whole-list route: one entry per pass (what binds it, its rate against that peak), the
    top-level keys = the longest pass, the call's PMC traffic against its compulsory bytes, and the per-pair kernel's HBM-bound leg kept as
    a clearly labelled secondary."""
    rows = []
    for q in headline.get("passes") or []:
        ms = q["ms"]
        e = {"kernel": q["pass"], "ms": ms}
        if "bytes_cleared" in q:
            e.update(bound="hbm", achieved=q["bytes_cleared"] / (ms * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s")
        elif "fp32_fma" in q:
            # the rows it gathers over its duration against the L2's peak (what binds it: common.SIMS_LIMITER); its arithmetic against the
            # fp32 vector peak beside it
            gbps = q["row_bytes"] / (ms * 1e-3) / 1e9
            e.update(bound="l2", achieved=gbps, peak=L2_PEAK_GBS, unit="GB/s", frac_of_gather_only_ceiling=gbps / GATHER_CEILING_GBS,
                     gather_only_ceiling_GBps=GATHER_CEILING_GBS, fp32_TFLOPs=2 * q["fp32_fma"] / (ms * 1e-3) / 1e12,
                     fp32_frac_of_vector_peak=2 * q["fp32_fma"] / (ms * 1e-3) / 1e12 / F32_PEAK_TFLOPS, limiter=SIMS_LIMITER)
        elif "exponentials" in q:
            e.update(bound="valu", achieved=q["exponentials"] / (ms * 1e-3) / 1e9, peak=KERNEL_EVAL_PEAK_G, unit="G kernel evaluations/s",
                     peak_source=KERNEL_EVAL_PEAK_SOURCE)
        elif "id_row_bytes" in q:
            e.update(bound="hbm", achieved=(q["id_row_bytes"] + q.get("bytes_written", 0)) / (ms * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s")
        if "achieved" in e:
            e["frac"] = e["achieved"] / e["peak"]
        rows.append(e)
    top = max((r for r in rows if "frac" in r), key=lambda r: r["ms"], default=None)
    out = {"bound": top["bound"] if top else "valu", "kernel": (top["kernel"] if top else headline.get("kernel")) + " (the longest pass of the timed call; all passes below)",
           "achieved": top["achieved"] if top else None, "peak": top["peak"] if top else None, "unit": top["unit"] if top else None,
           "frac": top["frac"] if top else None, "kernel_ms": top["ms"] if top else None,
           "traffic": traffic, "traffic_source": traffic_src, "compulsory_bytes": compulsory,
           "traffic_over_compulsory": (traffic / compulsory) if traffic else None, "call_ms": dev_s * 1e3,
           "call_compulsory_GBps": compulsory / dev_s / 1e9, "call_hbm_frac_on_compulsory_bytes": compulsory / dev_s / 1e9 / HBM_PEAK_GBS,
           "passes": rows,
           "note": "the timed steps run the whole-list route (csrc/lists.hip): its passes bind on different resources, so every pass is priced against "
                   "its own peak (bound = hbm: bytes / 8 TB/s; l2: packed rows gathered / 34.5 TB/s; valu: RBF kernel evaluations / the rate of a VALU-only "
                   "loop of the same evaluation) and the top-level keys repeat the longest pass; compulsory_bytes = the id rows once + one packed row per "
                   "distinct term of the STEP (rows that lists share are compulsory once) + the scores; traffic = PMC bytes of ALL the call's kernels.  SURVEY 8(d)'s algorithmic bytes (every "
                   "position x a fp32 row) do not describe this route: it gathers a term once per LIST (roofline.headline_leg.algorithmic_GBps is kept "
                   "for reference and exceeds the HBM peak).  kernel_ms / passes[].ms are the passes of ONE call running alone (HIP events between them); with the default two "
                   "step streams the kernels of consecutive steps overlap and stretch, so the rocprofv3 average that agrees with kernel_ms is the one of "
                   "`bench.py --step-streams 1` (profiles/r06/knrm_bench_kernel_stats.csv; profiles/r06/README.md)",
           "headline_leg": headline}
    if hbm_leg is not None:
        out["per_pair_hbm_leg"] = {k: v for k, v in hbm_leg.items() if k != "headline_leg"}
        out["per_pair_hbm_leg"]["what"] = ("SECONDARY, not what the timed steps launch: the one-pair-per-workgroup kernel on uniform ids over a table 20x the "
                                            "Infinity Cache, where lists share nothing and HBM binds (launches that are not whole lists, training batches and huge "
                                            "tables run this kernel)")
    return out


def interaction_record(args, ctx, model, steps, warmup, n_queries, with_cpu):
    """One KNRM / DRMM measurement: headline leg + HBM roofline leg (+ CPU baseline on rank 0 at N = 1)."""
    Q, L, D = getattr(args, "qlen", 4), 800, args.dim
    world = ctx.world
    strong = args.scaling == "strong"
    per_rank_q = n_queries // world if strong else n_queries
    if strong and n_queries % world:
        raise SystemExit("--scaling strong needs --queries divisible by the number of GPUs")
    nb = max(1, args.batches if model == "knrm" else min(args.batches, 2))      # (a DRMM batch is 250,000 pairs = 1.6 GB of id rows)
    leg = InteractionLeg(args, ctx, model, args.vocab, args.uniform_ids, per_rank_q, nb, 1 + ctx.rank)
    elapsed, dev_s = leg.run(warmup, steps, args.repeats)
    serial = None
    if leg.step_side and world == 1:
        # the same steps strictly one after the other on ONE stream (what the line reported before round 6's step streams): the call's
        # own duration - the per-pass figures of `roofline` belong to this form
        sides, leg.step_side = leg.step_side, []
        s_elapsed, s_dev, s_rep = repeated_timed_loop(ctx, leg.step, 2, steps, None, 3)
        leg.step_side = sides
        serial = {"value": leg.n_pairs * steps / s_elapsed, "unit": "pairs/s", "ms_per_step": 1e3 * s_elapsed / steps, "call_ms": s_dev * 1e3,
                  "ms_per_step_min": s_rep["ms_per_step_min"], "ms_per_step_max": s_rep["ms_per_step_max"],
                  "what": "bench.py --step-streams 1: every step's call waits for the previous one (3 repetitions of the K steps, median)"}
    zero_idf = None
    if model == "drmm" and world == 1 and not args.uniform_ids:
        # configs[2]'s second run: the same lists with the all-zero idf rows EmbedText produces by default
        z = InteractionLeg(args, ctx, model, args.vocab, args.uniform_ids, per_rank_q, 1, 1 + ctx.rank, zero_idf=True)
        z_elapsed, _ = z.run(2, max(3, steps // 2))
        z_err = z.check_against_oracle(min(64, z.n_pairs))[2]       # (the timed scores of its last step against the C oracle)
        zero_idf = {"value": z.n_pairs * max(3, steps // 2) / z_elapsed, "unit": "pairs/s", "ms_per_step": 1e3 * z_elapsed / max(3, steps // 2),
                    "steps": max(3, steps // 2), "query_idf": "all zeros (EmbedText without an idf table, extractor/embedtext.py:86-96)",
                    "oracle_check": z_err}
        del z
        torch.cuda.empty_cache()
    n_pairs = leg.n_pairs
    req_b, nonpad, distinct = leg.bytes_requested_per_pair()
    abytes = algorithmic_bytes_per_pair(model, Q, L, D)
    launches = len(leg.slices)
    if leg.lists:
        req_lists, distinct_per_list = leg.bytes_requested_per_pair_lists()
    headline = {
        "ids": "uniform" if args.uniform_ids else "Zipf(1.1)", "vocab": args.vocab, "kernel": kernel_of(model, n_pairs // launches, args.vocab, leg.row_stride, args.resident),
        "kernel_ms": dev_s * 1e3 / launches,
        "pairs_per_launch": n_pairs / launches, "mean_nonpad_terms_per_doc": nonpad, "mean_distinct_terms_per_doc": distinct, "requested_bytes_per_pair": req_b,
        "requested_GBps": n_pairs * req_b / dev_s / 1e9,
        "algorithmic_bytes_per_pair": abytes, "algorithmic_GBps": n_pairs * abytes / dev_s / 1e9,
        "note": "cache-level rates of the headline leg: bytes the kernel requests (ids + one packed row per distinct in-vocabulary term of a document) and the "
                "SURVEY §8(d) algorithmic bytes (all L positions x fp32 row - pads and OOV terms are scored in closed form, never gathered) "
                "over the per-step device time (one HIP event pair around the timed steps; in a multi-GPU run it includes the all_gather). "
                "Zipf ids hit L2 / Infinity Cache, so neither is an HBM rate",
    }
    if leg.lists:
        passes = leg.lists_pass_times() if (len(leg.slices) == 1 and not args.no_pass_times) else None
        if passes:
            # what each pass does per step (the figures DESIGN.md section 3.5 prices the passes with) over its own duration
            rows = distinct_per_list * (n_pairs / args.docs)
            tokens = nonpad * n_pairs
            K = 11
            # query-term rows the pooling kernel walks per position: 1 / 2 for queries of one / two real terms, 4 from three on (a row per term slot)
            nq = torch.cat([(b["query"][:: args.docs] > 0).sum(dim=1) for b in leg.batches]).float()
            real_q = float(torch.where(nq <= 2, nq, torch.full_like(nq, 4.0) * ((Q + 3) // 4)).mean().item()) if Q <= 4 else float(Q)
            idb = 4 if (args.resident and model == "knrm") else 8       # bytes per id: the candidate store's tables are int32
            names = ["lists_clear_kernel (byte maps)", "lists_mark_kernel", "lists_query_kernel<5>", f"lists_sims_kernel<5, {'false' if model == 'knrm' else 'true'}>",
                     "lists_knrm_pool_kernel" if model == "knrm" else "lists_drmm_pool_wave_kernel"]
            work = [
                {"bytes_cleared": (n_pairs / args.docs) * ((args.vocab + 1023) // 1024 * 1024)},
                {"id_row_bytes": n_pairs * L * idb, "byte_stores": tokens, "GBps_of_id_rows": n_pairs * L * idb / (passes[1] * 1e-3) / 1e9},
                {"lists": n_pairs / args.docs},
                {"rows_gathered": rows, "row_bytes": rows * leg.row_stride * 4, "row_GBps": rows * leg.row_stride * 4 / (passes[3] * 1e-3) / 1e9,
                 "fp32_fma": rows * Q * leg.row_stride, "fp32_TFLOPs": 2 * rows * Q * leg.row_stride / (passes[3] * 1e-3) / 1e12, "pipe": SIMS_PIPE},
                # (the pooling pass evaluates the kernels of REAL query terms only - a pad of the fixed-length query row has similarity 0 at every
                #  position, its sums are closed forms: lists.hip - so the pass is priced on the evaluations it executes; the nominal count, every
                #  position x Q terms x K kernels, is kept beside it)
                ({"id_row_bytes": n_pairs * L * idb, "table_lookups": tokens, "exponentials": tokens * real_q * K, "exponentials_nominal": tokens * Q * K,
                  "mean_query_term_rows_walked": real_q, "Gexp_per_s": tokens * real_q * K / (passes[4] * 1e-3) / 1e9}
                 if model == "knrm" else {"id_row_bytes": n_pairs * L * idb, "table_lookups": tokens, "lds_increments": tokens * Q}),
            ]
            headline["passes"] = [{"pass": nm, "ms": ms, **w} for nm, ms, w in zip(names, passes, work)]
            headline["passes_note"] = ("HIP events recorded by the library on the launch stream between the passes of 5 more steps after the timed loop "
                                       "(csrc/capamd_profiling.h: capamd_debug_lists_timing); a pass's ms includes its launch gap, the five add up to the step")
        headline.update({
            "route": f"whole candidate lists (capamd_{model}_forward_lists): per list every distinct term's row gathered once "
                     + ("(its four similarities kept), documents pooled from 16-byte lookups" if model == "knrm" else "(its four histogram bins kept), documents pooled from 4-byte lookups"),
            "kernel": f"lists_mark_kernel + lists_query_kernel<5> + lists_sims_kernel<5, {'false' if model == 'knrm' else 'true'}> + lists_{model}_pool_kernel",
            "mean_distinct_terms_per_list": distinct_per_list,
            "requested_bytes_per_pair": req_lists, "requested_GBps": n_pairs * req_lists / dev_s / 1e9,
            "per_pair_kernel_requested_bytes_per_pair": req_b,
            "note": headline["note"] + "; on this route the rows of a LIST's distinct terms are gathered once (requested_bytes_per_pair counts them spread over the "
                                       "list's documents; per_pair_kernel_requested_bytes_per_pair is what the per-pair kernels - bench.py --per-pair - ask for)"})
    roof = None
    if not args.no_roofline_leg and world == 1:       # (N > 1: every rank does the same work; the roofline leg is an N = 1 measurement)
        # HBM-bound leg: uniform ids over a table 20x the Infinity Cache -> (almost) every gathered row comes from HBM
        big = InteractionLeg(args, Ctx1(ctx), model, args.roofline_vocab, True, 64, 2, 77)
        _, big_s = big.run(2, max(5, min(steps, 10)))
        big_req, big_nonpad, big_distinct = big.bytes_requested_per_pair()
        ach = big.n_pairs * big_req / big_s / 1e9
        roof = {
            "bound": "hbm", "kernel": kernel_of(model, big.n_pairs, args.roofline_vocab, big.row_stride), "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": ach / HBM_PEAK_GBS,
            "traffic": None,
            "traffic_source": f"not measured in this run (--no-pmc-traffic): profiles/r03/{model}_hbm_traffic.json holds the builder-run "
                              "FETCH_SIZE x2 + WRITE_SIZE per launch of this leg and of the headline leg",
            "leg": f"uniform term ids over a {args.roofline_vocab}-row table ({args.roofline_vocab * big.row_stride * 4 / 1e9:.1f} GB packed, 20x the 256 MB "
                   "Infinity Cache), 64 x 1000 pairs per launch, 2 alternating batches: HBM is the binding resource",
            "kernel_ms": big_s * 1e3, "pairs_per_launch": big.n_pairs, "requested_bytes_per_pair": big_req, "mean_nonpad_terms_per_doc": big_nonpad,
            "mean_distinct_terms_per_doc": big_distinct,
            "definition": "achieved = bytes the kernel requests (int64 id rows + one packed 1280-byte table row per distinct in-vocabulary document term "
                          "and per query term + score) / per-launch device time (one HIP event pair around the timed launches); every requested row is a "
                          "distinct random row, so requested bytes = HBM bytes up to the <= 5 % the Infinity Cache can hold",
            "headline_leg": headline,
            "read_ceiling_note": "a kernel that only reads sustains 6.0-6.6 TB/s streaming 8 GiB and 6.3-6.4 TB/s on random 1280-byte rows of the same "
                                 "5.1 GB table on this part (scripts/ubench/hbm_read.hip, profiles/r02/hbm_read.txt; builder-run, not measured in this invocation)",
        }
        del big
        _tables.pop((ctx.dev.index, args.roofline_vocab, args.dim), None)
        torch.cuda.empty_cache()
        if not args.no_pmc_traffic and ctx.rank == 0 and args.vocab <= 400001 and not args.uniform_ids:
            roof["traffic"], roof["traffic_source"] = pmc_traffic(args, model)
            if roof["traffic"] is not None:
                roof["traffic_over_requested"] = roof["traffic"] / (roof["pairs_per_launch"] * roof["requested_bytes_per_pair"])
    if leg.lists and world == 1:
        # the line's roofline describes what its timed steps launch: the list route's passes (the per-pair HBM-bound leg stays as a secondary)
        # compulsory HBM bytes of a call: the id rows once, every table row the step's lists touch once (shared rows come from cache), the scores
        union_rows = float(np.mean([int((torch.unique(b["posdoc"]) > 0).sum().item()) for b in leg.batches]))
        compulsory = n_pairs * (L * 8 + 4) + (n_pairs / args.docs) * Q * (8 + leg.row_stride * 4) + union_rows * leg.row_stride * 4
        headline["distinct_terms_per_step"] = union_rows
        traffic, traffic_src = None, "not measured in this run (--no-pmc-traffic)"
        if not args.no_pmc_traffic and ctx.rank == 0 and not args.uniform_ids:
            traffic, traffic_src = pmc_traffic(args, model, "lists")
        roof = lists_roofline(model, headline, roof, n_pairs, serial["call_ms"] * 1e-3 if serial else dev_s, compulsory, traffic, traffic_src)
    total_pairs = n_pairs * world
    rec = {
        "metric": "query-doc pairs scored/sec",
        "value": total_pairs * steps / elapsed,
        "unit": "pairs/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": 1e3 * elapsed / steps,
        "repeats": dict(leg.repeats, value_min=total_pairs * steps / (leg.repeats["ms_per_step_max"] * 1e-3 * steps),
                        value_max=total_pairs * steps / (leg.repeats["ms_per_step_min"] * 1e-3 * steps)),
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"{model.upper()} inference (BASELINE.json configs[{1 if model == 'knrm' else 2}]): qlen={Q} dlen={L} "
                        f"embed={D} vocab={args.vocab}, {args.docs} docs/query x {per_rank_q} queries per step per GPU, "
                        f"{'uniform' if args.uniform_ids else 'Zipf(1.1)'} term ids, lognormal doc lengths, "
                        + (f"scored as whole candidate lists ({distinct_per_list:.0f} distinct terms per list), " if leg.lists else "") +
                        f"{launches} launch(es) per step" + (f" round-robin over {len(leg.side)} HIP streams" if leg.side else "") + (", replayed as one captured HIP graph" if leg.graphs else "") +
                        (f", consecutive steps round-robin over {len(leg.step_side)} HIP streams (step i + 1's passes overlap step i's)" if leg.step_side else "") +
                        f", {len(leg.batches)} distinct batches in rotation"
                        + (", query_idf ~ U(0.5, 8) per query (a second run with all-zero idf rows: zero_idf_run)" if model == "drmm" else ""),
            "pairs_per_step_per_gpu": n_pairs,
            "parallelism": f"query-sharded x{world}, one all_gather of scores per step (asynchronous, under the next step's scoring)" if world > 1 else "single GPU",
        },
        "roofline": roof if roof is not None else {"bound": "hbm", "kernel": KERNEL_VARIANT[model], "achieved": None, "peak": HBM_PEAK_GBS,
                                                   "unit": "GB/s", "frac": None, "traffic": None, "headline_leg": headline},
    }
    if serial is not None:
        rec["step_streams"] = {"streams": len(leg.step_side), "serial_steps": serial,
                               "what": "consecutive steps (independent batches of candidate lists) are issued round-robin over this many HIP streams, each "
                                       "with its own workspace and score buffer: a step's HBM-bound mark pass and the tails of its passes run under the other "
                                       "stream's sims / pooling passes.  `value` / `ms_per_step` are the K steps' throughput in this form; `roofline` prices "
                                       "the passes of ONE call as they run alone (serial_steps.call_ms)"}
    if zero_idf is not None:
        rec["zero_idf_run"] = zero_idf
    if ctx.use_dist:
        rec["collective"] = collective_info(ctx, n_pairs)
    if with_cpu and world == 1:
        rec["cpu_baseline"] = cpu_baseline(args, model, leg.check_against_oracle, leg)
    return rec

