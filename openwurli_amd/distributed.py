"""Multi-GPU driver for the batch-render path: independent note x velocity jobs are dealt
round-robin over the ranks (job j -> rank j mod G, which balances bass/treble cost), every rank renders its
shard with the lane = job kernels, and ONE gather collective brings the f32 result slabs to rank 0
(``torch.distributed`` backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).
There is no collective anywhere in the data path.

The realtime engine path does not shard (one engine = one wavefront of voices + one serial chain):
many instances are independent replicas, see bench.py.
"""
import numpy as np


def shard_indices(n_jobs, rank, world):
    """Indices of the jobs rank ``rank`` renders: j = rank, rank + world, ..."""
    return list(range(rank, n_jobs, world))


def unshard(slabs, n_jobs, world):
    """Inverse of ``shard_indices``: slabs[r] is [len(shard_indices(n_jobs, r, world)) (padded), n] -> [n_jobs, n]."""
    n = slabs[0].shape[1]
    out = np.zeros((n_jobs, n), dtype=slabs[0].dtype)
    for r in range(world):
        idx = shard_indices(n_jobs, r, world)
        out[idx] = slabs[r][:len(idx)]
    return out


def gather_chunks(n_pad, n, world, max_gather_bytes):
    """Row ranges [k0, k1) of a rank's slab such that rank 0's receive buffer (world x rows x n f32) stays within max_gather_bytes."""
    rows = max(1, int(max_gather_bytes // max(1, world * n * 4)))
    return [(k0, min(k0 + rows, n_pad)) for k0 in range(0, max(n_pad, 1), rows)] if n_pad else []


def batch_render_sharded(jobs, sample_rate, duration_s, render_fn=None, device=None, group=None, timings=None, to_host=True,
                         max_gather_bytes=2 << 30):
    """Render ``jobs`` across the ranks of the default process group; rank 0 returns float32 [n_jobs, n], others None.

    ``render_fn(local_jobs) -> array [n_local, n]`` defaults to the HIP batch renderer; tests inject a CPU stand-in
    to exercise the sharding + gather logic under gloo.  Without an initialised process group this is a world of one
    (no collective).  ``timings`` (dict) receives ``render_s`` and ``gather_s`` of this rank (device-synchronised);
    ``to_host=False`` leaves the gathered slabs on rank 0's device and returns the list of per-rank slabs instead
    (benchmarks: the host-side reassembly of a multi-GB result is not part of the render).

    Rank 0's receive side is ONE pre-sized buffer, not ``world`` separate slabs, and with ``to_host=True`` it is bounded: the slabs come
    in row chunks of at most ``max_gather_bytes`` (world x rows x n f32) that are reassembled into the result as they arrive -- 8 ranks x
    1.57 GB (the scaled grid at 8 GPUs) never stand on rank 0's device at once.  ``to_host=False`` needs the whole set on the device and
    gathers it in one step into a [world, n_pad, n] tensor (returned as its ``world`` views).  ``timings['gather_steps']`` = collectives used.
    """
    import time
    import torch
    import torch.distributed as dist
    distributed = dist.is_available() and dist.is_initialized()
    rank, world = (dist.get_rank(group), dist.get_world_size(group)) if distributed else (0, 1)
    jobs = list(jobs)
    n_jobs = len(jobs)
    n = int(duration_s * sample_rate)
    mine = shard_indices(n_jobs, rank, world)
    n_pad = (n_jobs + world - 1) // world          # equal-sized slabs so one gather suffices
    on_gpu = (dist.get_backend(group) == "nccl") if distributed else (render_fn is None)
    dev = torch.device("cuda", device if device is not None else torch.cuda.current_device()) if on_gpu else torch.device("cpu")

    def sync():
        if on_gpu:
            torch.cuda.synchronize(dev)
    slab = torch.zeros((n_pad, n), dtype=torch.float32, device=dev)
    sync()
    t0 = time.perf_counter()
    if mine:
        if render_fn is None:
            from . import engine
            if on_gpu:
                # render straight into a torch-owned HBM buffer (f64), then narrow to the f32 slab: no host round trip.
                # The library writes on its own stream: drain torch's stream first (a recycled block of the caching allocator may
                # still have torch work pending on it), and synchronise afterwards before torch reads the result.
                tmp = torch.empty((len(mine), n), dtype=torch.float64, device=dev)
                torch.cuda.current_stream(dev).synchronize()
                engine.batch_render([jobs[i] for i in mine], sample_rate, duration_s, device=dev.index, out_device_ptr=tmp.data_ptr(), stride=n)
                torch.cuda.synchronize(dev)
                slab[:len(mine)] = tmp.to(torch.float32)
                del tmp
            else:
                raise RuntimeError("the HIP renderer needs the nccl backend (a GPU per rank); pass render_fn for CPU tests")
        else:
            local = np.asarray(render_fn([jobs[i] for i in mine]), dtype=np.float32)
            slab[:len(mine)] = torch.from_numpy(local).to(dev)
    sync()
    t1 = time.perf_counter()
    out = None
    steps = 0
    if distributed and not to_host:
        recv = torch.zeros((world, n_pad, n), dtype=torch.float32, device=dev) if rank == 0 else None
        gathered = list(recv.unbind(0)) if rank == 0 else None
        sync()                                      # receive buffer allocated and zeroed before the gather clock starts
        t2 = time.perf_counter()
        dist.gather(slab, gathered, dst=0, group=group)     # the one exchange step
        steps = 1
        sync()
    elif distributed:
        chunks = gather_chunks(n_pad, n, world, max_gather_bytes)
        rows = max((k1 - k0 for k0, k1 in chunks), default=0)
        recv = torch.zeros((world, rows, n), dtype=torch.float32, device=dev) if rank == 0 else None
        if rank == 0:
            out = np.zeros((n_jobs, n), dtype=np.float32)
        sync()
        t2 = time.perf_counter()
        for k0, k1 in chunks:                       # the exchange step, in bounded pieces (one piece unless the result exceeds max_gather_bytes)
            piece = slab[k0:k1].contiguous()
            dist.gather(piece, [recv[r, :k1 - k0] for r in range(world)] if rank == 0 else None, dst=0, group=group)
            steps += 1
            if rank == 0:
                sync()
                host = recv[:, :k1 - k0].cpu().numpy()
                for r in range(world):              # rows k0..k1 of rank r's slab are jobs r + world * k
                    idx = [r + world * k for k in range(k0, k1) if r + world * k < n_jobs]
                    out[idx] = host[r, :len(idx)]
        sync()
        gathered = None
    else:
        gathered = [slab]
        t2 = time.perf_counter()
    t3 = time.perf_counter()
    if timings is not None:
        timings["render_s"] = t1 - t0
        timings["gather_s"] = t3 - t2
        timings["world_seen"] = world
        timings["gather_steps"] = steps
    if rank != 0:
        return None
    if not to_host:
        return gathered
    if out is not None:
        return out
    return unshard([g.cpu().numpy() for g in gathered], n_jobs, world)


def render_midi_sharded(event_lists, render_fn=None, device=None, group=None, **render_kw):
    """``midi_render.render_midi`` across the ranks: event list j -> rank j mod G, every rank renders its shard, the rows are padded
    to the longest render of the whole job set (one MAX all-reduce of a scalar) and ONE gather brings the f32 slabs to rank 0, which
    returns the list of renders at their own lengths (others None).  ``render_fn(local_lists) -> list of 1-D arrays`` defaults to
    the HIP renderer; the CPU tests inject a stand-in."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    event_lists = list(event_lists)
    n_jobs = len(event_lists)
    mine = shard_indices(n_jobs, rank, world)
    on_gpu = dist.get_backend(group) == "nccl"
    dev = torch.device("cuda", device if device is not None else torch.cuda.current_device()) if on_gpu else torch.device("cpu")
    if render_fn is None:
        if not on_gpu:
            raise RuntimeError("the HIP renderer needs the nccl backend (a GPU per rank); pass render_fn for CPU tests")
        from . import midi_render

        def render_fn(lists):
            return midi_render.render_midi(lists, device=dev.index, **render_kw)
    local = [np.asarray(x, dtype=np.float32) for x in render_fn([event_lists[i] for i in mine])] if mine else []
    lengths = torch.zeros(max(n_jobs, 1), dtype=torch.int64, device=dev)
    for i, x in zip(mine, local):
        lengths[i] = x.size
    dist.all_reduce(lengths, op=dist.ReduceOp.MAX, group=group)          # every rank learns every job's length (and the longest)
    longest = int(lengths.max().item())
    n_pad = (n_jobs + world - 1) // world
    slab = torch.zeros((n_pad, max(longest, 1)), dtype=torch.float32, device=dev)
    for k, x in enumerate(local):
        slab[k, :x.size] = torch.from_numpy(x).to(dev)
    recv = torch.zeros((world,) + tuple(slab.shape), dtype=torch.float32, device=dev) if rank == 0 else None     # one pre-sized buffer
    gathered = list(recv.unbind(0)) if rank == 0 else None
    dist.gather(slab, gathered, dst=0, group=group)                      # the one exchange step of the data path
    if rank != 0:
        return None
    full = unshard([g.cpu().numpy() for g in gathered], n_jobs, world)
    return [full[j, :int(lengths[j].item())].copy() for j in range(n_jobs)]


def model_notes_job_list(notes=range(33, 97), velocities=(20, 35, 50, 65, 80, 95, 110, 127)):
    """The job grid of ml/render_model_notes.py:26,106-114 (64 notes x 8 velocity buckets)."""
    return [{"note": n, "velocity": v, "mlp": False, "poweramp": False, "volume": 1.0, "speaker": 0.0, "r_ldr": 1e6}
            for n in notes for v in velocities]
