"""`python bench.py --gpus N` with no rank environment starts its own N ranks.

The driver may run the multi-GPU bench either as `python -m torch.distributed.run ... bench.py --gpus N` (RANK / WORLD_SIZE are
set: bench.py is a rank and never comes here) or as plain `python bench.py --gpus N`.  In the second form this module re-executes
bench.py under torch.distributed.run - one process per GPU, rendezvous on 127.0.0.1 (the container hostname may not resolve),
HSA_ENABLE_IPC_MODE_LEGACY=0 kept (the host driver only supports dmabuf IPC) - and passes the ranks' stdout through, so that the
last line of THIS process's stdout is rank 0's one JSON line.
"""
import os
import socket
import subprocess
import sys


def needs_self_launch(gpus, environ=None):
    """True when --gpus N > 1 was asked for and nobody has made this process a rank"""
    environ = os.environ if environ is None else environ
    return gpus > 1 and "WORLD_SIZE" not in environ and "RANK" not in environ


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_command(script, argv, gpus, port, python=None):
    """The command line of the N-rank run: the driver's own form (see bench.py's docstring) with the caller's arguments unchanged."""
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
            "--master-port", str(port), script] + list(argv)


def launch_env(environ=None):
    env = dict(os.environ if environ is None else environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")           # torch.distributed.run would set 1 and say so on stderr; the CPU legs are rank-0-at-N=1 only
    env["CAPAMD_SELF_LAUNCHED"] = "1"
    return env


def self_launch(script, argv, gpus):
    """Run the N ranks, return their exit code.  stdout / stderr are inherited: rank 0's JSON line is the last line of stdout."""
    cmd = launch_command(script, argv, gpus, free_port())
    print("bench.py: --gpus %d without a rank environment: starting %s" % (gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=launch_env())
