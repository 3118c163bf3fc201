"""`bench.py --gpus N` must start its N ranks itself: launcher plumbing on CPU, no GPU touched.
The ranks run bench.py's --selftest-launcher leg (gloo group, one all-reduce) instead of the tracker."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                          env=env, timeout=300)


def test_gpus2_spawns_two_ranks_and_relays_one_json_line():
    r = _run(["--gpus", "2", "--rehearse-gloo", "--selftest-launcher", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["gpus_arg"] == 2
    assert out["value"] == 3.0          # 1 + 2: both ranks took part in the collective
    assert out["steps"] == 3 and out["warmup"] == 1   # the arguments reached the ranks


def test_launcher_reports_rank_failure():
    # one rank dies before the collective: the launcher must exit non-zero and print no result line
    r = _run(["--gpus", "2", "--selftest-launcher"], {"ICTR_SELFTEST_FAIL_RANK": "1"})
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
