"""bench.py's watchdog for the N>1 extras: if they do not come back in time, rank 0 prints the headline line with what
the extras have so far and the process ends with status 0 (the driver's scaling record must not be lost to them)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(code):
    return subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=60)


def test_watchdog_prints_the_headline_and_exits_cleanly_when_the_extras_hang():
    r = _run("import time, bench\n"
             "line = {'metric': 'm', 'value': 1.0}\n"
             "partial = {'5a': {'GBps': 2.0}}\n"
             "dog = bench.Watchdog(0.2, 0, line, partial)\n"
             "time.sleep(30)\n"  # the 'hung' extras
             "print('never')\n")
    assert r.returncode == 0, r.stderr
    out = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(out) == 1 and "never" not in r.stdout
    line = json.loads(out[0])
    assert line["value"] == 1.0 and line["multi_gpu_workloads"]["5a"] == {"GBps": 2.0}
    assert "not finished" in line["multi_gpu_workloads"]["error"]


def test_watchdog_on_other_ranks_only_ends_the_process():
    r = _run("import time, bench\n"
             "dog = bench.Watchdog(0.2, 3, None, {})\n"
             "time.sleep(30)\n")
    assert r.returncode == 0 and r.stdout.strip() == ""


def test_watchdog_stands_down_when_the_extras_finish():
    r = _run("import time, bench\n"
             "dog = bench.Watchdog(5.0, 0, {'value': 1.0}, {})\n"
             "assert dog.finish() is True\n"
             "assert dog.finish() is False\n"  # (only the first caller prints)
             "print('done')\n")
    assert r.returncode == 0 and r.stdout.strip() == "done", r.stderr
