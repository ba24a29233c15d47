"""N>1 path on CPU: world_size-2 gloo run of the job sharding + single gather.  The render itself is
replaced by a deterministic stand-in (no GPU here); what is under test is shard_indices / padding / gather / unshard."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fake_render(n):
    def f(jobs):
        return np.stack([np.full(n, j["note"] * 1000 + j["velocity"], dtype=np.float32) + np.arange(n, dtype=np.float32) / n for j in jobs])
    return f


def _worker(rank, world, port, n_jobs, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from openwurli_amd import distributed as owd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    jobs = owd.model_notes_job_list()[:n_jobs]
    sr, dur = 1000.0, 0.016
    out = owd.batch_render_sharded(jobs, sr, dur, render_fn=_fake_render(int(sr * dur)))
    if rank == 0:
        ref = _fake_render(int(sr * dur))(jobs)
        q.put(bool(out.shape == ref.shape and np.array_equal(out, ref)))
    else:
        q.put(out is None)
    dist.destroy_process_group()


def _fake_midi(lists):
    # a deterministic "render" whose length and content depend on the event list: length = 10 * events + 3
    return [np.arange(10 * len(l) + 3, dtype=np.float32) + 1000.0 * len(l) for l in lists]


def _worker_midi(rank, world, port, n_jobs, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from openwurli_amd import distributed as owd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lists = [[(0.1 * k, 0, 60, 100)] * (j % 5) for j, k in enumerate(range(n_jobs))]      # ragged, incl. empty jobs
    out = owd.render_midi_sharded(lists, render_fn=_fake_midi)
    if rank == 0:
        ref = _fake_midi(lists)
        q.put(bool(len(out) == n_jobs and all(a.shape == b.shape and np.array_equal(a, b) for a, b in zip(out, ref))))
    else:
        q.put(out is None)
    dist.destroy_process_group()


def _run_midi(world, n_jobs):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_midi, args=(r, world, port, n_jobs, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(res), res


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, n_jobs):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_jobs, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(res), res


def test_shard_helpers():
    sys.path.insert(0, ROOT)
    from openwurli_amd import distributed as owd
    assert owd.shard_indices(10, 1, 4) == [1, 5, 9]
    assert sorted(sum((owd.shard_indices(513, r, 8) for r in range(8)), [])) == list(range(513))
    assert len(owd.model_notes_job_list()) == 512


def test_gather_world2_even():
    _run(2, 16)


def test_gather_world2_ragged():
    _run(2, 13)            # unequal shards: padding rows must not leak into the result


def test_midi_jobs_world2_ragged_lengths():
    _run_midi(2, 11)       # renders of different lengths, shards of different sizes, empty event lists
