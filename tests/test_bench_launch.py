"""`python bench.py --gpus N` without a rank environment starts its own N ranks (benchlib/launch.py): the command it builds is the
driver's documented form, and the launcher really brings up N ranks (gloo, on CPU, with a stand-in script) whose rank 0 prints the last
line of stdout."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from benchlib import launch  # noqa: E402


def test_needs_self_launch_only_without_a_rank_environment():
    assert launch.needs_self_launch(8, environ={})
    assert launch.needs_self_launch(2, environ={"PATH": "/bin"})
    assert not launch.needs_self_launch(1, environ={})
    assert not launch.needs_self_launch(8, environ={"WORLD_SIZE": "8", "RANK": "3"})
    assert not launch.needs_self_launch(8, environ={"RANK": "0"})


def test_launch_command_is_the_drivers_form():
    cmd = launch.launch_command("/x/bench.py", ["--gpus", "4", "--steps", "20", "--warmup", "3"], 4, 29617, python="python3")
    assert cmd == ["python3", "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1", "--master-port", "29617",
                   "/x/bench.py", "--gpus", "4", "--steps", "20", "--warmup", "3"]
    env = launch.launch_env({"A": "b"})
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and env["A"] == "b" and env["CAPAMD_SELF_LAUNCHED"] == "1"
    assert launch.launch_env({"HSA_ENABLE_IPC_MODE_LEGACY": "1"})["HSA_ENABLE_IPC_MODE_LEGACY"] == "1"   # an explicit setting is kept


def test_bench_py_takes_the_launcher_before_touching_a_gpu():
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main():"):]
    assert main.index("launch.needs_self_launch(args.gpus)") < main.index("Ctx(args)")
    assert "launch with torch.distributed.run" not in src           # (the round-5 exit)


def test_self_launch_brings_up_n_ranks(tmp_path):
    script = tmp_path / "standin.py"
    script.write_text(textwrap.dedent("""
        import json, os, sys
        import torch, torch.distributed as dist
        dist.init_process_group("gloo")
        t = torch.ones(1)
        dist.all_reduce(t)
        if dist.get_rank() == 0:
            print(json.dumps({"ranks": int(t.item()), "argv": sys.argv[1:], "self": os.environ.get("CAPAMD_SELF_LAUNCHED")}), flush=True)
        dist.destroy_process_group()
    """))
    driver = textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        from benchlib import launch
        raise SystemExit(launch.self_launch({str(script)!r}, ["--gpus", "2", "--steps", "5"], 2))
    """)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, "-c", driver], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    last = [l for l in p.stdout.splitlines() if l.strip()][-1]
    assert json.loads(last) == {"ranks": 2, "argv": ["--gpus", "2", "--steps", "5"], "self": "1"}
