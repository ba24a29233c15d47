"""N>1 path on CPU: world_size-2 and world_size-8 gloo runs of the job sharding + gather.  The render itself is
replaced by a deterministic stand-in (no GPU here); what is under test is shard_indices / padding / gather (one step, or bounded
row chunks into rank 0's one pre-sized receive buffer) / reassembly."""
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


def _worker(rank, world, port, n_jobs, q, max_bytes=2 << 30, want_steps=1, to_host=True):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from openwurli_amd import distributed as owd
    torch.set_num_threads(1)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    jobs = owd.model_notes_job_list()[:n_jobs]
    sr, dur = 1000.0, 0.016
    tm = {}
    out = owd.batch_render_sharded(jobs, sr, dur, render_fn=_fake_render(int(sr * dur)), max_gather_bytes=max_bytes, timings=tm, to_host=to_host)
    ok = tm["world_seen"] == world and tm["gather_steps"] == want_steps
    if rank == 0:
        ref = _fake_render(int(sr * dur))(jobs)
        if not to_host:                     # the per-rank slabs, still views of ONE [world, n_pad, n] buffer
            ok = ok and len(out) == world and all(o.data_ptr() == out[0].data_ptr() + r * out[0].numel() * 4 for r, o in enumerate(out))
            out = owd.unshard([o.numpy() for o in out], n_jobs, world)
        q.put(bool(ok and out.shape == ref.shape and np.array_equal(out, ref)))
    else:
        q.put(bool(ok and out is None))
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


def _run(world, n_jobs, **kw):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_jobs, q), kwargs=kw) for r in range(world)]
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


def test_gather_chunks_helper():
    sys.path.insert(0, ROOT)
    from openwurli_amd import distributed as owd
    assert owd.gather_chunks(64, 16, 8, 2 << 30) == [(0, 64)]
    assert owd.gather_chunks(65, 16, 8, 8 * 16 * 4 * 10) == [(0, 10), (10, 20), (20, 30), (30, 40), (40, 50), (50, 60), (60, 65)]
    assert owd.gather_chunks(3, 1000, 8, 1) == [(0, 1), (1, 2), (2, 3)]          # never less than a row
    # the scaled batch grid at 8 GPUs (8 192 jobs x 1 s per GPU): 12.6 GB of slabs in six bounded steps
    ch = owd.gather_chunks(8192, 48000, 8, 2 << 30)
    assert len(ch) == 6 and all((k1 - k0) * 8 * 48000 * 4 <= 2 << 30 for k0, k1 in ch) and ch[-1][1] == 8192


def test_gather_world8_even_one_step():
    _run(8, 512)           # configs[3] literally: 64 jobs per rank, one gather


def test_gather_world8_ragged():
    _run(8, 509)           # shards of 64 and 63 jobs: padding rows must not leak, every job lands in its place


def test_gather_world8_fewer_jobs_than_ranks():
    _run(8, 5)             # three ranks render nothing and still take part in the collective


def test_gather_world8_in_bounded_chunks():
    # receive buffer limited to 8 ranks x 10 rows x 16 samples: the 64-row slabs arrive in seven steps and are reassembled as they come
    _run(8, 509, max_bytes=8 * 16 * 4 * 10, want_steps=7)


def test_gather_world8_left_on_device_is_one_buffer():
    _run(8, 509, to_host=False)


def test_midi_jobs_world8_ragged_lengths():
    _run_midi(8, 43)
