"""What every leg of bench.py shares: the peaks the legs are priced against, the rank / device context, the timed loop of the contract
(W untimed steps, exactly K steps between two fences, max over ranks), the collective's record, the seeded embedding tables."""
import os
import time

import torch
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec (guides: MI355X_MICROARCH.md "HBM3E peak BW")


MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 / fp16 (MI355X_MICROARCH.md "Peak BF16/FP16 MFMA")
KERNEL_VARIANT = {"knrm": "knrm_forward_kernel<5, 1, true, 6, false>", "drmm": "drmm_forward_kernel<5, 1, true, 6, false>"}
# launches of more than 3072 pairs over a table the cache hierarchy can hold run the persistent streaming kernels (interaction_stream.h)
STREAM_VARIANT = {"knrm": "stream_kernel<5, false, KnrmStream>", "drmm": "stream_kernel<5, false, DrmmStream>"}


def kernel_of(model, pairs_per_launch, vocab, row_stride_floats, resident=False):
    """The kernel the library picks for a launch (knrm.hip / drmm.hip: knrm_launch, drmm_launch)."""
    streaming = pairs_per_launch > 3072 and vocab * row_stride_floats * 4 <= (1 << 30) and vocab <= (1 << 22) and os.environ.get(
        f"CAPAMD_{model.upper()}_STREAM", "1") != "0"
    name = STREAM_VARIANT[model] if streaming else KERNEL_VARIANT[model]
    return name.replace("false,", "true,") if (streaming and resident) else name


def algorithmic_bytes_per_pair(model, Q, L, D):
    """ids int64 + one fp32 embedding row per term + fp32 score (+ idf for DRMM)."""
    b = L * (8 + 4 * D) + Q * (8 + 4 * D) + 4
    return b + (4 * Q if model == "drmm" else 0)


def default_queries(model):
    """Queries per step when --queries is not given: 64 (configs[1] as SURVEY 8(d) concretises it); DRMM: 250 - configs[2] stands in for
    Robust04's 250 topics x BM25 top-1000."""
    return 250 if model == "drmm" else 64


class Ctx:
    """rank / device / process group of this run"""

    def __init__(self, args):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={self.world}")
        torch.cuda.set_device(local)
        self.dev = torch.device("cuda", local)
        self.use_dist = self.world > 1 or os.environ.get("CAPAMD_FORCE_DIST") == "1"  # the env knob exercises the RCCL path on one rank
        self.dist = None
        if self.use_dist:
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=self.dev)
            self.dist = dist

    def fence(self):
        torch.cuda.synchronize()
        if self.use_dist:
            self.dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(self, seconds):
        if not self.use_dist:
            return seconds
        t = torch.tensor([seconds], device=self.dev, dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        if self.use_dist:
            self.dist.destroy_process_group()


def timed_loop(ctx, step, warmup, steps, drain=None):
    """W untimed steps, then exactly K steps between two fences (barrier + synchronize on both sides); wall clock = max over ranks.
    ONE HIP event pair on the launch stream brackets the K steps: (event time / K) is the per-step device time and can never
    exceed the wall-clock step.  (Event pairs around every single launch - what round 1 did - put a system-scope release /
    acquire between consecutive kernels, which drops the table rows the previous launch left in L2: those launches ran 7-10 %
    slower than the back-to-back launches of the timed loop, hence a `kernel_ms` above `ms_per_step` in BENCH_r01.)"""
    for i in range(warmup):
        step(i)
    if drain:
        drain()
    ctx.fence()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for i in range(steps):
        step(warmup + i)
    if drain:
        drain()          # (inside the timed region: collectives still in flight are part of the K steps)
    ev1.record()
    ctx.fence()
    elapsed = ctx.max_over_ranks(time.perf_counter() - t0)
    return elapsed, ev0.elapsed_time(ev1) * 1e-3 / steps


def collective_info(ctx, floats_per_rank):
    """What the N > 1 line says about its one collective: the backend and the number of ranks the process group really has (so that a
    SCALE record proves N ranks took part), the bytes one step gathers, and the duration of that all_gather on its own (10 blocking
    repetitions after a fence; inside the timed steps it runs asynchronously under the next step's scoring)."""
    if not ctx.use_dist:
        return None
    dist, dev = ctx.dist, ctx.dev
    world = dist.get_world_size()
    src = torch.zeros(floats_per_rank, dtype=torch.float32, device=dev)
    dst = torch.empty(floats_per_rank * world, dtype=torch.float32, device=dev)
    for _ in range(2):
        dist.all_gather_into_tensor(dst, src)
    ctx.fence()
    t0 = time.perf_counter()
    for _ in range(10):
        dist.all_gather_into_tensor(dst, src)
    torch.cuda.synchronize()
    ms = ctx.max_over_ranks(time.perf_counter() - t0) * 100.0
    return {"backend": dist.get_backend(), "rccl_ranks": world, "collective": "all_gather_into_tensor (fp32 scores)", "gathered_bytes_per_step": floats_per_rank * world * 4,
            "gather_ms": ms}


_tables = {}


def table(dev, V, D):
    """seeded the same on every rank (replicated table)"""
    key = (dev.index, V, D)
    if key not in _tables:
        g = torch.Generator(device=dev)
        g.manual_seed(0)
        emb = torch.randn((V, D), generator=g, device=dev) * 0.4
        emb[0] = 0
        _tables[key] = emb
    return _tables[key]
# What bounds each pass of the whole-list route and the peak it is priced against:
F32_PEAK_TFLOPS = 157.3        # fp32 vector = fp32 MFMA peak (MI355X_MICROARCH.md "Peak FP32 (vector)" / "(matrix)")
# RBF kernel evaluations per second the VALUs sustain when they do nothing else: scripts/ubench/valu_rates.hip's loop of the pooling
# kernel's evaluation in the form the kernel uses (K(s) = 2^-(A s + B)^2: fma, mul, exp, add per value), every SIMD busy.  The row taken:
# "4-instruction form", 16 waves per CU = 8,529 G/s (profiles/r05/valu_rates.txt; 8,160 at 8 waves per CU, 8,735 at 32; the pooling kernel
# runs 24 waves per CU at 76 registers).  Until round 5 this constant was 7,150 - the 5-instruction form's row, which the kernel no longer uses.
KERNEL_EVAL_PEAK_G = 8529.0
KERNEL_EVAL_PEAK_SOURCE = ("scripts/ubench/valu_rates.hip, 4-instruction form (fma mul exp add), 16 waves per CU: 8,529 G evaluations/s "
                           "(8,160 at 8 waves per CU, 8,735 at 32; the kernel runs 24): profiles/r05/valu_rates.txt")
SIMS_PIPE = "valu"             # the pipe the sims pass's dot products run on ("mfma" once they are v_mfma_f32_4x4x1_16b_f32)
# What binds the sims pass: the L2 -> CU gather of the packed rows - its LATENCY, not its bytes.  Every (list, distinct term) pair is one
# 1280-byte row through the vector L1: 4.0 GB per call on the benchmark's lists at 13.4-14.7 TB/s - 0.40 of the L2's 34.5 TB/s peak
# (MI355X_MICROARCH.md section L2), 0.87-0.95 of the 15.5 TB/s a gather-only kernel reaches on the same kind of request stream (round 3's
# probe).  Round 6 rebuilt the pass six ways around precomputed work lists (profiles/r06/lists_sims_steps.txt: all
# slower; with every row in L1 the pass takes 211-225 us whatever its structure), then took bytes away and the time stayed: two lists per
# workgroup with their shared rows loaded once pull 18 % fewer rows through the L1s for 2 % of the pass's time, and an L2 hit rate of 0.67
# instead of 0.84 does not show either (lists_sims_pairs_ab.txt).  What the pass waits for is a trip's loads coming back, with five waves per
# SIMD to cover them.  The fp32 VALU figure (27.9 TF/s = 0.18 of 157.3) is kept beside it: the arithmetic would allow 51 us if rows were free.
L2_PEAK_GBS = 34500.0            # aggregate of the eight XCDs' L2s (MI355X_MICROARCH.md "L2 (per XCD)": ~34.5 TB/s)
GATHER_CEILING_GBS = 15500.0     # what a gather-only kernel of 1280-byte rows reaches from L2 / Infinity Cache (profiles/r03: stream gather probe)
SIMS_LIMITER = {"bound": "l2 gather (its latency): one 1280-byte packed row per (list, distinct term) through the vector L1; the rows of a call (4.0 GB on the "
                         "benchmark's lists) over the pass's duration against the L2's 34.5 TB/s, with the gather-only ceiling (15.5 TB/s) beside it",
                "valu_issue_utilisation": 0.52, "l2_hit_rate": 0.84,
                "waves_waiting_over_issuing": 2.9, "packed_fma_share_of_valu_instructions": 0.37,
                "source": "profiles/r05/pmc_lists_knrm.txt (builder-run counter passes); profiles/r06/lists_sims_steps.txt, lists_sims_pairs_ab.txt (the rebuilds, "
                          "their ablations, the byte-saving forms)"}


class Ctx1:
    """a single-rank view of the context (the roofline leg runs on rank 0 only, without the collective)"""

    def __init__(self, ctx):
        self.world, self.rank, self.dev, self.use_dist, self.dist = 1, 0, ctx.dev, False, None

    def fence(self):
        torch.cuda.synchronize()

    def max_over_ranks(self, s):
        return s



def repeated_timed_loop(ctx, step, warmup, steps, drain=None, repeats=1):
    """The contract's timed loop `repeats` times inside one invocation (the warm-up once): the median repetition's (wall seconds,
    per-step device seconds) and what all of them were - one 12 ms sample used to decide the headline, and boxes differ by 3-5 %
    (VERDICT r5 weak #12).  Every repetition is exactly `steps` steps between two fences; `value` is the median repetition's."""
    runs = []
    for r in range(max(1, repeats)):
        runs.append(timed_loop(ctx, (lambda i, base=r * steps: step(base + i)), warmup if r == 0 else 0, steps, drain))
    order = sorted(range(len(runs)), key=lambda k: runs[k][0])
    mid = order[(len(runs) - 1) // 2]          # (an even count: the lower median, a repetition that was actually measured)
    stats = {"n": len(runs), "which": "median of n repetitions of the K timed steps (each between its own two fences)",
             "ms_per_step_each": [1e3 * e / steps for e, _ in runs], "ms_per_step_min": 1e3 * runs[order[0]][0] / steps,
             "ms_per_step_max": 1e3 * runs[order[-1]][0] / steps}
    return runs[mid][0], runs[mid][1], stats
