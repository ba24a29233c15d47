"""bench.py --gpus N without a launcher starts the N ranks itself (child processes, before any GPU call), relays rank 0's ONE
JSON line and fails when a rank fails.  Dry run on CPU: gloo rendezvous on 127.0.0.1, stand-in renderer, the real sharding
(job j -> rank j mod G) and the real single gather of openwurli_amd.distributed."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args, env_extra=None, timeout=300):
    env = dict(os.environ, OW_BENCH_DRYRUN_BACKEND="gloo")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True, text=True, env=env, timeout=timeout)


def test_gpus_flag_launches_that_many_ranks_and_prints_one_line():
    r = _bench("--gpus", "2", "--workload", "batch", "--steps", "1", "--warmup", "0")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 1 and d["warmup"] == 0 and d["dry_run"] is True
    for key in ("metric", "value", "unit", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d
    b = d["batch"]
    assert b["literal"]["jobs"] == 512 and b["literal"]["ranks_seen_by_collective"] == 2
    assert b["scaled"]["jobs"] == 2 * 64 and b["scaled"]["scaling"] == "weak" and b["literal"]["scaling"] == "strong"
    assert b["literal"]["gather_ms"] >= 0.0 and b["literal"]["render_ms"] > 0.0


def test_single_rank_needs_no_launcher_and_no_process_group():
    r = _bench("--gpus", "1", "--workload", "batch", "--steps", "1", "--warmup", "0")
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 1 and d["batch"]["literal"]["ranks_seen_by_collective"] == 1


def test_a_failing_rank_fails_the_launcher():
    r = _bench("--gpus", "2", "--workload", "engines", "--steps", "1", "--warmup", "0")      # the dry run refuses the engine workload
    assert r.returncode != 0 and "ranks failed" in r.stderr
