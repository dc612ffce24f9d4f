"""Multi-GPU readiness on whatever the box has: the distributed legs of bench.py run through RCCL with the ranks that
exist - one, under CAPAMD_FORCE_DIST=1, on a single-GPU box; two when two GPUs are visible - and their JSON line is validated, so that
the first 8-GPU run of the driver does not meet these code paths for the first time."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _run_bench(extra, nproc=1, timeout=600):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    if nproc == 1:
        env["CAPAMD_FORCE_DIST"] = "1"        # a process group of ONE rank still takes the collective path
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + extra
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
               "--master-port", env["MASTER_PORT"], os.path.join(ROOT, "bench.py"), "--gpus", str(nproc)] + extra
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


def _check_line(rec, n_gpus, scaling):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in rec, k
    assert rec["n_gpus"] == n_gpus and rec["scaling"] == scaling and rec["value"] > 0 and rec["higher_is_better"] is True
    c = rec["collective"]
    assert c["backend"] == "nccl" and c["rccl_ranks"] == n_gpus and c["gathered_bytes_per_step"] > 0 and c["gather_ms"] > 0


def test_bench_knrm_weak_leg_over_rccl():
    rec = _run_bench(["--steps", "3", "--warmup", "1", "--queries", "8", "--no-cpu-baseline", "--no-also", "--no-roofline-leg", "--no-pmc-traffic"])
    _check_line(rec, 1, "weak")
    assert rec["collective"]["gathered_bytes_per_step"] == 8 * 1000 * 4


def test_bench_bert_strong_leg_over_rccl():
    """BASELINE configs[4]'s shape (queries in contiguous blocks over the ranks, one gather of the document scores) on one rank."""
    rec = _run_bench(["--model", "bert", "--scaling", "strong", "--queries", "1", "--docs", "64", "--steps", "1", "--warmup", "1",
                      "--no-cpu-baseline", "--no-bert-other-dtype"])
    _check_line(rec, 1, "strong")
    assert "configs[4]" in rec["config"]["workload"]


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (the single-GPU box covers the same path with one rank)")
def test_bench_two_ranks_over_rccl():
    rec = _run_bench(["--steps", "3", "--warmup", "1", "--queries", "8", "--no-cpu-baseline", "--no-also", "--no-roofline-leg", "--no-pmc-traffic"], nproc=2)
    _check_line(rec, 2, "weak")
    rec = _run_bench(["--model", "bert", "--scaling", "strong", "--queries", "2", "--docs", "32", "--steps", "1", "--warmup", "1",
                      "--no-cpu-baseline", "--no-bert-other-dtype"], nproc=2)
    _check_line(rec, 2, "strong")


def _predict_worker(rank, world, port, out_dir):
    import numpy as np
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    from tests.test_gpu_parity import _knrm_model
    from tests.helpers import load_case
    from capreolus_amd.trainer import PytorchTrainer

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    c = load_case("knrm", "ranklist")
    r = _knrm_model(c)
    r.model.to(torch.device("cuda", rank))
    B = c["query"].shape[0]
    q2d = {"1": [f"d{i}" for i in range(0, 70)], "2": [f"d{i}" for i in range(70, 120)], "3": [f"d{i}" for i in range(120, B)]}

    class Sampler(torch.utils.data.IterableDataset):
        qid_to_docids = q2d

        def __iter__(self):
            for qid, docs in self.qid_to_docids.items():
                for d in docs:
                    i = int(d[1:])
                    yield {"qid": qid, "posdocid": d, "query": c["query"][0], "posdoc": c["posdoc"][i], "query_idf": c["query_idf"][0]}

        def __len__(self):
            return sum(len(v) for v in self.qid_to_docids.values())

        def get_qid_docid_pairs(self):
            return ((q, d) for q, docs in self.qid_to_docids.items() for d in docs)

    preds = PytorchTrainer({"batch": 16}).predict(r, Sampler())
    flat = np.array([preds[q][d] for q, docs in q2d.items() for d in docs], dtype=np.float16)
    np.save(os.path.join(out_dir, f"preds{rank}.npy"), flat)
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (test_predict_over_rccl_single_rank covers one rank)")
def test_predict_two_ranks_over_rccl(tmp_path):
    """`PytorchTrainer.predict` sharded by query over two GPUs, scores joined by ONE RCCL all_gather: every rank ends with the
    predictions of the whole run, equal to the reference fixture's fp16 scores."""
    import numpy as np
    import torch.multiprocessing as mp

    from tests.helpers import load_case

    mp.spawn(_predict_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    want = load_case("knrm", "ranklist")["ref_scores_f16"]
    for rank in range(2):
        assert np.array_equal(np.load(tmp_path / f"preds{rank}.npy"), want)
