#!/usr/bin/env python
"""Benchmark of the reranker scoring hot path on MI355X (BASELINE.json metric:
query-doc pairs scored/sec at 1/2/4/8 GPUs).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus N --steps K --warmup W        (no rank environment: starts its own N ranks, benchlib/launch.py)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic candidate lists already
resident in HBM: by default BASELINE.json configs[1] — KNRM inference, qlen 4, dlen 800,
GloVe-shaped 400,001 x 300 fp32 table, 1000 docs/query, 64 queries per step per GPU
(SURVEY.md §8d "Config 2").  Multi-GPU: queries are sharded over ranks (weak scaling: every
rank scores its own 64 queries per step; `--scaling strong`: the step's queries are divided
over the ranks) and each step ends with one RCCL all-gather of the score vectors.
Rank 0 prints ONE JSON line.

What the line carries (N = 1, default model):
  value / ms_per_step  KNRM on Zipf(1.1) term ids (the headline leg), consecutive steps score DIFFERENT batches
  roofline             HBM fraction of the KNRM kernel, from a leg where HBM is the binding resource (uniform ids over a table
                       far larger than the 256 MB Infinity Cache): bytes the kernel REQUESTS / kernel time / 8 TB/s, <= 1 by
                       construction.  The Zipf leg's cache-level rates are reported next to it (roofline.headline_leg);
                       roofline.traffic = HBM-side bytes per launch of that leg from the PMC counters, measured by two
                       `rocprofv3 --pmc` child runs inside this invocation (FETCH_SIZE, WRITE_SIZE; --no-pmc-traffic skips them)
  cpu_baseline         C oracle (OpenMP) on the host cores + the reference's ATen op sequence swept over thread counts and
                       batch sizes (best reported) + the BASELINE configs[0] stand-in (16 training steps + 325 x 100 predict)
  also                 DRMM (configs[2]) and BERT-base MaxP (configs[3]: bf16 operands, the step's passages in two slices on two
                       streams; the fp16 figure under other_operand_type) legs with their own roofline / cpu_baseline
"""
import argparse
import copy
import os
import sys

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from benchlib import launch  # noqa: E402
from benchlib.bert import bench_bert  # noqa: E402
from benchlib.common import Ctx, _tables, default_queries  # noqa: E402
from benchlib.interaction import InteractionLeg, interaction_record  # noqa: E402
from benchlib.report import emit  # noqa: E402
from benchlib.siblings import bench_sibling  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=5,
                    help="the K timed steps are run this many times inside the invocation (warm-up once); value / ms_per_step are the median repetition's, "
                         "the others are reported under `repeats`")
    ap.add_argument("--model", default="knrm", choices=["knrm", "drmm", "bert", "drmmtks", "pacrr", "convknrm"],
                    help="knrm (default, BASELINE.json's metric) | drmm | bert | the row-N4 siblings drmmtks, pacrr, convknrm")
    ap.add_argument("--queries", type=int, default=0, help="queries per step per GPU (0 = 64 for the interaction models - 250 for drmm, configs[2] - and 1 for bert)")
    ap.add_argument("--docs", type=int, default=1000, help="candidate documents per query")
    ap.add_argument("--qlen", type=int, default=4, help="query terms (the extractor's maxqlen): 4 is BASELINE.json's; up to 8 on the list route")
    ap.add_argument("--launch-docs", type=int, default=0, help="pairs per kernel launch (0 = whole step in one launch)")
    ap.add_argument("--step-streams", type=int, default=0,
                    help="list route: consecutive steps (independent batches of candidate lists) go round-robin over this many HIP streams (one call per "
                         "step; per-stream workspace and score buffer), so that the passes of step i + 1 overlap those of step i; 0 = auto (2 for steps "
                         "of up to 128 lists, else 1), 1 = strictly serial steps")
    ap.add_argument("--launch-streams", type=int, default=4,
                    help="with --launch-docs: the launches of a step go round-robin over this many HIP streams, so the tail of one candidate "
                         "list (as long as its longest document) overlaps the next list's launch; 1 = one stream, strictly serial launches")
    ap.add_argument("--no-graph", action="store_true",
                    help="with --launch-docs: issue every launch of a step from Python instead of replaying the step's launches as one captured "
                         "HIP graph (64 launches of 1000 pairs are host-bound at ~25 us of Python / ctypes per launch)")
    ap.add_argument("--batches", type=int, default=4, help="distinct batches the steps rotate through (consecutive steps never score the same batch)")
    ap.add_argument("--vocab", type=int, default=400001)
    ap.add_argument("--dim", type=int, default=300)
    ap.add_argument("--uniform-ids", action="store_true", help="headline leg on uniform instead of Zipf(1.1) term ids")
    ap.add_argument("--force-lists", action="store_true",
                    help="with --uniform-ids: score as whole candidate lists all the same (the list route's worst case: lists that share nothing)")
    ap.add_argument("--roofline-vocab", type=int, default=4000001,
                    help="rows of the table of the HBM-bound roofline leg (uniform ids; 4,000,001 x 1280 B = 5.1 GB, 20x the Infinity Cache)")
    ap.add_argument("--no-roofline-leg", action="store_true")
    ap.add_argument("--no-pmc-traffic", action="store_true",
                    help="skip the two rocprofv3 --pmc child runs that measure roofline.traffic (FETCH_SIZE / WRITE_SIZE of the HBM-bound leg)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every rank scores --queries queries per step; strong: the step's --queries queries are divided over the ranks")
    ap.add_argument("--resident", action="store_true", help="score through the device-resident int32 candidate store (row N1)")
    ap.add_argument("--per-pair", action="store_true",
                    help="KNRM / DRMM: score the step with the per-pair kernels (one fused kernel, every pair on its own) instead of as whole candidate lists")
    ap.add_argument("--bert-dtype", default="bf16", choices=["bf16", "fp16"],
                    help="16-bit operand type of the BERT encoder: bf16 is what BASELINE.json configs[3] names; fp16 is the engine's default outside the "
                         "bench (three more mantissa bits: closer to the reference's fp32 scores, ~3 %% slower - more operand bits toggle per MFMA)")
    ap.add_argument("--no-bert-other-dtype", action="store_true", help="BERT: skip the short run with the other 16-bit operand type")
    ap.add_argument("--bert-skip-padding", action="store_true",
                    help="BERT: encode passages in length buckets (multiples of 32 tokens) - identical scores, padded rows not computed. Off by "
                         "default here: the headline line times the reference's full 4 x 256-token computation")
    ap.add_argument("--bert-microbatch", type=int, default=256, help="passages (of 256 tokens) per encoder micro-batch")
    ap.add_argument("--bert-streams", type=int, default=2,
                    help="BERT: slices of a step's passages encoded concurrently on their own HIP streams and workspaces (the engine's default is 2; "
                         "1 = strictly serial kernels).  `roofline` times the dominant kernel in a separate serial step after the timed ones, so "
                         "that its HIP-event duration is not inflated by a concurrent kernel")
    ap.add_argument("--no-pass-times", action="store_true", help="list route: skip the per-pass HIP-event leg (it binds the -DCAPAMD_PROFILING build of the library)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the DRMM / BERT legs of the default invocation")
    ap.add_argument("--cpu-pairs", type=int, default=0, help="pairs in the CPU baseline sample (0 = auto)")
    return ap.parse_args()


def main():
    args = parse()
    if launch.needs_self_launch(args.gpus):      # `python bench.py --gpus N`: this process becomes the launcher of N ranks (benchlib/launch.py)
        raise SystemExit(launch.self_launch(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    if os.environ.get("CAPAMD_LIB_PATH"):       # an A/B build of the library (scripts/build_variant*.sh): it has no profiling twin
        args.no_pass_times = True
    ctx = Ctx(args)
    if args.model == "bert":
        rec = bench_bert(args, ctx, args.steps, args.warmup, with_cpu=not args.no_cpu_baseline)
    elif args.model in ("drmmtks", "pacrr", "convknrm"):
        rec = bench_sibling(args, ctx)
    else:
        rec = interaction_record(args, ctx, args.model, args.steps, args.warmup, args.queries or default_queries(args.model), with_cpu=not args.no_cpu_baseline)
        default_line = args.model == "knrm" and ctx.world == 1 and not args.no_also and not args.uniform_ids and not args.launch_docs and not args.resident
        if default_line and ctx.rank == 0:
            # the other two north-star models, timed by the same driver run (short legs; their own roofline / cpu_baseline)
            short = max(5, min(args.steps, 10))
            torch.cuda.empty_cache()
            try:
                # the same lists through a device-resident candidate store (int32 id tables + index pairs: what `PytorchTrainer.predict` scores
                # from its second call on) - half the id-row bytes of the int64 headline
                a2 = copy.copy(args)
                a2.resident = True
                rl = InteractionLeg(a2, ctx, "knrm", args.vocab, False, args.queries or 64, 2, 1 + ctx.rank)
                r_elapsed, _ = rl.run(3, args.steps, 3)      # (median of three repetitions: the first one after a leg's set-up runs 5-8 % slower)
                rl.check_against_oracle(min(64, rl.n_pairs))
                rec["resident_int32_route"] = {"value": rl.n_pairs * args.steps / r_elapsed, "unit": "pairs/s", "ms_per_step": 1e3 * r_elapsed / args.steps,
                                               "step_streams": len(rl.step_side) or 1,
                                               "what": "bench.py --resident: the headline's lists as int32 tables + index pairs (capamd_knrm_forward_lists, indexed form)"}
                del rl
            except Exception as e:  # noqa: BLE001
                rec["resident_int32_route"] = {"error": f"{type(e).__name__}: {e}"}
            torch.cuda.empty_cache()
            try:
                # ... and the list route's worst case next to the per-pair HBM leg: uniform term ids - a list's 1000 documents share almost
                # nothing (~212,000 distinct terms per list instead of 49,000), so the route gathers four times the rows for the same pairs
                a3 = copy.copy(args)
                a3.uniform_ids, a3.force_lists = True, True
                ul = InteractionLeg(a3, ctx, "knrm", args.vocab, True, args.queries or 64, 2, 1 + ctx.rank)
                u_elapsed, _ = ul.run(2, short)
                _, u_distinct = ul.bytes_requested_per_pair_lists()
                ul.check_against_oracle(min(64, ul.n_pairs))
                rec["lists_on_uniform_ids"] = {"value": ul.n_pairs * short / u_elapsed, "unit": "pairs/s", "ms_per_step": 1e3 * u_elapsed / short, "steps": short,
                                               "mean_distinct_terms_per_list": u_distinct, "step_streams": len(ul.step_side) or 1,
                                               "what": "bench.py --uniform-ids --force-lists: the whole-list route where lists share nothing (its worst case; the "
                                                       "per-pair kernels - roofline.per_pair_hbm_leg - are what such launches should take)"}
                del ul
            except Exception as e:  # noqa: BLE001
                rec["lists_on_uniform_ids"] = {"error": f"{type(e).__name__}: {e}"}
            try:
                # ... and queries of eight terms (`maxqlen` is a free option of the reference's extractor, embedtext.py:28-31): two blocks of
                # four query terms on the list route, the timed scores checked against the C oracle
                a4 = copy.copy(args)
                a4.qlen = 8
                ql = InteractionLeg(a4, ctx, "knrm", args.vocab, False, args.queries or 64, 2, 1 + ctx.rank)
                q_elapsed, _ = ql.run(2, short)
                q_err = ql.check_against_oracle(min(64, ql.n_pairs))[2]
                rec["qlen8_lists"] = {"value": ql.n_pairs * short / q_elapsed, "unit": "pairs/s", "ms_per_step": 1e3 * q_elapsed / short, "steps": short,
                                      "oracle_check_max_err_of_scale": q_err, "lists": bool(ql.lists),
                                      "what": "bench.py --qlen 8: the headline's lists under eight-term queries (query lengths 1-8), whole-list route"}
                del ql
            except Exception as e:  # noqa: BLE001
                rec["qlen8_lists"] = {"error": f"{type(e).__name__}: {e}"}
            _tables.clear()
            torch.cuda.empty_cache()
            rec["also"] = []
            # ... and the row-N4 siblings (short legs - 12 steps after 4, one warm-up step per rotating batch: timed scores checked against the oracle, an HBM-bound leg each, no CPU timing)
            for leg in (lambda: interaction_record(args, ctx, "drmm", max(5, args.steps // 2), 3, default_queries("drmm"), with_cpu=not args.no_cpu_baseline),
                        lambda: bench_bert(args, ctx, 5, 2, with_cpu=not args.no_cpu_baseline),
                        lambda: bench_sibling(args, ctx, "drmmtks", 12, 4, with_cpu=False),
                        lambda: bench_sibling(args, ctx, "pacrr", 12, 4, with_cpu=False),
                        lambda: bench_sibling(args, ctx, "convknrm", 12, 4, with_cpu=False)):
                try:
                    rec["also"].append(leg())
                except Exception as e:  # noqa: BLE001  a failing secondary leg must not take the headline line with it
                    rec["also"].append({"error": f"{type(e).__name__}: {e}"})
                _tables.clear()
                torch.cuda.empty_cache()
    ctx.close()
    if ctx.rank == 0:
        emit(rec)


if __name__ == "__main__":
    main()
