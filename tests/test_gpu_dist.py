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
    from tests.test_gpu_parity import _knrm_model, _multiquery_sampler
    from tests.helpers import load_case
    from capreolus_amd.trainer import PytorchTrainer

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    c = load_case("knrm", "multiquery")
    r = _knrm_model(c)
    r.model.to(torch.device("cuda", rank))
    s = _multiquery_sampler(c)
    preds = PytorchTrainer({"batch": 16}).predict(r, s)
    flat = np.array([preds[q][d] for q, docs in s.qid_to_docids.items() for d in docs], dtype=np.float16)
    np.save(os.path.join(out_dir, f"preds{rank}.npy"), flat)
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (test_predict_over_rccl_single_rank covers one rank)")
def test_predict_two_ranks_over_rccl(tmp_path):
    """`PytorchTrainer.predict` sharded by query over two GPUs, scores joined by ONE RCCL all_gather: every rank ends with the
    predictions of the whole run - the SAME fp16 bits as one process scoring the unsharded run (the reference's 8-query KNRM fixture;
    tests/test_gpu_parity.py: test_knrm_predictions_do_not_depend_on_the_sharding is the single-GPU form of this assertion)."""
    import numpy as np
    import torch.multiprocessing as mp

    sys.path.insert(0, ROOT)
    from tests.helpers import load_case
    from tests.test_gpu_parity import _knrm_model, _multiquery_sampler
    from capreolus_amd.trainer import PytorchTrainer

    mp.spawn(_predict_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    c = load_case("knrm", "multiquery")
    s = _multiquery_sampler(c)
    unsharded = PytorchTrainer({"batch": 16}).predict(_knrm_model(c), s)
    want = np.array([unsharded[q][d] for q, docs in s.qid_to_docids.items() for d in docs], dtype=np.float16)
    assert np.abs(want.astype(np.float64) - c["ref_scores"]).max() <= 2e-3 * np.abs(c["ref_scores"]).max()
    for rank in range(2):
        assert np.array_equal(np.load(tmp_path / f"preds{rank}.npy"), want)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no rank environment (how the driver's record of round 5 invoked it): bench.py re-executes itself under
    torch.distributed.run (benchlib/launch.py) and rank 0's JSON line says two RCCL ranks took part."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--queries", "8"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    rec = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    _check_line(rec, 2, "weak")
