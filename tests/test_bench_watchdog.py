"""bench.py's process handling for N > 1: `python bench.py --gpus N` starts its own ranks (a child process, before
anything touches the GPU), the shutdown watchdog, and the pieces of the fresh-child protocol the N > 1 extras run under
that need no GPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(code, timeout=60):
    return subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=timeout)


def test_watchdog_ends_a_hung_shutdown_with_the_given_status():
    r = _run("import time, bench\n"
             "dog = bench.Watchdog(0.2, 'handshake', 0)\n"
             "time.sleep(30)\n"
             "print('never')\n")
    assert r.returncode == 0 and "never" not in r.stdout
    assert "handshake not finished" in r.stderr
    r = _run("import time, bench\n"
             "dog = bench.Watchdog(0.2, 'handshake', 4)\n"
             "time.sleep(30)\n")
    assert r.returncode == 4


def test_watchdog_stands_down_when_the_handshake_returns():
    r = _run("import time, bench\n"
             "dog = bench.Watchdog(0.3, 'handshake', 7)\n"
             "dog.finish()\n"
             "time.sleep(0.8)\n"
             "print('done')\n")
    assert r.returncode == 0 and r.stdout.strip() == "done", r.stderr


def test_bench_with_gpus_2_and_no_rank_environment_launches_its_own_ranks():
    """The driver's N = 1 command is `python3 bench.py --gpus 1 ...`; the same shape with --gpus 2 must not die in
    argument checking.  Without a GPU both ranks get as far as the device check (and say so): the launch path itself
    needs none."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["HIP_VISIBLE_DEVICES"] = "-1"  # (the same outcome on a box that has a GPU)
    env["CUDA_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--backend", "gloo", "--no-extra", "--no-cpu",
                        "--steps", "2", "--warmup", "1"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert "launch with torch.distributed.run" not in r.stderr
    assert r.stderr.count("bench.py needs an MI355X") >= 2, r.stderr[-2000:]
    assert r.returncode != 0  # the ranks' failure is the launcher's status


def test_free_port_is_bindable():
    r = _run("import bench, socket\n"
             "p = bench.free_port()\n"
             "s = socket.socket(); s.bind(('127.0.0.1', p)); print(p)\n")
    assert r.returncode == 0 and 1024 < int(r.stdout) < 65536


@pytest.mark.gpu
def test_direct_launch_runs_two_ranks_on_one_gpu_over_gloo():
    """`python bench.py --gpus 2` exactly as typed, ranks sharing GPU 0 over gloo (RCCL refuses two ranks on one
    device): one line, n_gpus 2, headline checked against the CPU checker."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--backend", "gloo", "--no-extra", "--no-cpu",
                        "--steps", "5", "--warmup", "2", "--prewarm-ms", "10"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["extras_failed"] is False and line["check_failed"] is False
    assert line["checked"]["ok"] is True and len(line["checked"]["seqs"]) == 2
