#!/usr/bin/env python3
"""Throughput bench of the OpenWurli DSP hot path on MI355X.

Default workload (BASELINE.json configs[1], SURVEY.md 8d config 2, scaled out per configs[4]):
  I independent engine instances per GPU, each 64 voices (all keys 33..96 struck at t=0,
  note_off+note_on re-strike of all 64 every 1.0 s so that 64 voices stay alive), 48 kHz host
  rate (2x oversampled chain at 96 kHz), full chain: tremolo + legacy DK preamp + behavioural
  power amp + speaker, volume 0.5, tremolo depth 0.5, MLP on, buffers of 512 samples, events
  split sample-accurately like the plugin does (plugin/src/lib.rs:128-149).
  Instance k plays velocity (40 + (37 k mod 88))/127 (config 5 decorrelation).

One "step" = one 512-sample buffer rendered by every instance of every rank.  `value` =
output samples per second over all instances of all ranks (weak scaling: I per GPU fixed).
The rendered audio stays in HBM (inputs resident, no PCIe in the timed region except the
per-step event list and the 40-byte-per-engine status block the host state machine needs);
`pcie_inclusive` repeats the run with every block copied to pinned host memory.

`--workload batch` makes BASELINE configs[3] (ml/render_model_notes.py: 64 notes x 8 velocities x 5 s,
`preamp-bench render` semantics) the headline instead: jobs dealt round-robin over the ranks, ONE
RCCL gather of the f32 slabs to rank 0, timed separately.  Every run (any workload, any N) also
carries a `batch` object with that literal grid and a grid scaled to 8 192 jobs per GPU, so the
driver's 1/2/4/8-GPU runs show the batch path at both sizes.

`--gpus N` without a launcher (WORLD_SIZE unset) starts the N ranks itself, before this process
touches the GPU; under torchrun it reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as usual.

After the timed region the line verifies itself (`verified`): every row of the last block finite and inside the level band, eight
engines spread over the pool with 64 active voices and zero NaN-guard / reset counters; a failed check exits non-zero.

Prints ONE JSON line (rank 0).  See DESIGN.md "Measurement" for the roofline arithmetic.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SR = 48000.0
BUF = 512
EPOCH = 48000          # re-strike period in samples
NOTES = list(range(33, 97))

# Algorithmic f64 flops per OUTPUT sample of one 64-voice engine (restated in DESIGN.md):
# 7 modes x 15 (the reference's 16 minus the amplitude x envelope multiply that the steady kernel carries as one recurrence, DESIGN
# deviation 9) + jitter 3 + pickup 13 + gain/sum 2 per voice-sample: the algorithm AS BUILT.  (SURVEY 8d counts 130 for the reference.)
FLOPS_VOICE_SAMPLE = 123
FLOPS_VOICES = 64 * FLOPS_VOICE_SAMPLE
# f64 flops the steady voice kernel EXECUTES per voice-sample by its PMC instruction mix (3.8 add + 20.3 mul + 2 x 35.9 fma + 1.1
# transcendental, profiles/r05_pmc_per_sample.md; 74.2 VALU instructions in all -- 8.4 / 27.5 / 35.5 / 1.2 and 83.8 until the pickup's soft
# limit stopped calling the library's tanh and the envelope moved onto the quadrature pair's radius, round 5): below the algorithmic
# count because the jitter-corrected rotation coefficients (4 flops per mode, reed.rs:281-283 evaluates them every sample) only change
# with the drift, every 16th sample, and are hoisted there, and because the folded form has no `envelope *= decay` per mode and sample
FLOPS_VOICE_SAMPLE_EXECUTED = 3.8 + 20.3 + 2 * 35.9 + 1.1
FLOPS_TREMOLO = 2 * (1000 + 25)    # Twin-T NR step + LDR law per OS sample, 2 OS samples -- per tremolo PHASE GROUP, not per engine
FLOPS_PREAMP = 2 * 1400 + 24       # main+shadow dk_step per OS sample + half-band up
# melange 12-node preamp as the kernel EXECUTES it (rank-one update of the inverse, no per-sample LU): per state and chain-rate
# sample build_rhs ~120 + S.rhs 288 + Sherman-Morrison correction ~50 + 3 Newton sweeps x ~150 + S_NI.i 72 + damp net ~50 = ~1050
FLOPS_PREAMP_MELANGE = 2 * 2 * 1050 + 24          # rank-one kernel (OW_MEL_RANK1=1): the Newton solve only, matrices updated by Sherman-Morrison
# literal kernel (default, k_preamp_mel_col): + the per-sample rebuild as the DEVICE does it, per solver state (the two states of an
# engine no longer share it): trailing 6x6 factorisation 13 quotients + 26 multiply-subtracts, twelve sparse unit-column solves 309
# multiply-subtracts + 123 quotients, folding the columns into v_pred / S N_i 123 + 84 multiply-adds: ~1 150 flops (quotient = 1; the
# reference's dense 12x12 inverse would be ~6 100)
FLOPS_PREAMP_MELANGE_LIT = 2 * 2 * (1050 + 1150) + 24
# melange power amp, per chain-rate sample with ONE Newton iteration (the count is data dependent: 1 on silence, 2..3 mean on a chord,
# up to 70): device models 8 x ~300, Jacobian 1 024, 16x16 LU 2 730, substitution 256, K products 2 x 512, S rhs 800, S_NI i 640
FLOPS_PA_MELANGE_PER_CHAIN_SAMPLE = 8 * 300 + 1024 + 2730 + 256 + 1024 + 800 + 640
FLOPS_POST = 2 * 90 + 24 + 45      # power amp x2 + half-band down + speaker/gain
KERNEL_OF = {"ops": "k_apply_ops", "voices": "k_voice_steady", "tremolo": "k_tremolo", "preamp": "k_preamp", "post": "k_post"}
PEAK_FP64_VALU_TFLOPS = 78.6       # 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz (MI355X FP64 vector)
FLOPS_BATCH_JOB = 3000             # config-4 job: 1 voice 129 + half-band 48 + 2 x 1400 legacy preamp + speaker 30


def instance_velocity(k):
    return (40 + (37 * k) % 88) / 127.0


def build_events(n_inst, kind):
    """kind 'strike' = note_on all keys; 'restrike' = note_off then note_on per key (config 2)."""
    from openwurli_amd import binding
    notes = np.array(NOTES, dtype=np.uint8)
    vel = np.array([instance_velocity(k) for k in range(n_inst)], dtype=np.float32)
    per = len(NOTES) * (2 if kind == "restrike" else 1)
    ev = np.zeros((n_inst, per), dtype=np.dtype(binding.MIDI_DTYPE))
    ev["engine"] = np.arange(n_inst, dtype=np.uint32)[:, None]
    if kind == "restrike":
        ev["type"][:, 0::2] = 1
        ev["note"][:, 0::2] = notes[None, :]
        ev["type"][:, 1::2] = 0
        ev["note"][:, 1::2] = notes[None, :]
        ev["value"][:, 1::2] = vel[:, None]
    else:
        ev["type"] = 0
        ev["note"] = notes[None, :]
        ev["value"] = vel[:, None]
    return ev.reshape(-1)


class Script:
    """Sample-accurate event script: renders one buffer per step, splitting at epoch boundaries."""

    def __init__(self, pool, n_inst, buf=BUF, host_out=None, epoch=None):
        self.pool = pool
        self.pos = 0
        self.buf = buf
        self.epoch = epoch or EPOCH
        self.host_out = host_out          # (pointer, stride) of a pinned host block, or None = audio stays in HBM
        self.ev_strike = build_events(n_inst, "strike")
        self.ev_restrike = build_events(n_inst, "restrike")
        # a host that batches the events of many instances keeps the list in a pinned block (ow_host_alloc): a burst that a big pool applies
        # on the device (ow_vm.h) is then uploaded straight from it
        if hasattr(pool, "alloc_host_events") and n_inst >= 8192:
            cache = getattr(pool, "_bench_events", None)
            if cache is None:
                cache = {}
                for name, ev in (("strike", self.ev_strike), ("restrike", self.ev_restrike)):
                    pinned = pool.alloc_host_events(ev.size)
                    pinned[:] = ev
                    cache[name] = pinned
                pool._bench_events = cache
            self.ev_strike, self.ev_restrike = cache["strike"], cache["restrike"]
        self.kernel_ms = np.zeros(5)
        self.kernel_launches = 0
        self.t_midi = 0.0
        self.t_render = 0.0

    def step(self, profile=False):
        done = 0
        while done < self.buf:
            t0 = time.perf_counter()
            if self.pos % self.epoch == 0:
                self.pool.midi(self.ev_strike if self.pos == 0 else self.ev_restrike)
            t1 = time.perf_counter()
            nxt = min(self.buf - done, self.epoch - (self.pos % self.epoch))
            if self.host_out is None:
                self.pool.render(nxt, to_host=False)
            else:
                self.pool.render_into(self.host_out[0], self.host_out[1], nxt)
            self.t_midi += t1 - t0
            self.t_render += time.perf_counter() - t1
            if profile:
                ms = self.pool.last_kernel_ms()
                self.kernel_ms += np.array([ms["ops"], ms["voices"], ms["tremolo"], ms["preamp"], ms["post"]]) * (nxt / self.buf)
            self.pos += nxt
            done += nxt
        if profile:
            self.kernel_launches += 1


def effective_cpus():
    """CPUs this process may actually use: affinity mask capped by the cgroup CPU quota (cpu.max).  The GPU boxes expose 256
    hardware threads but run the job under a 16-CPU quota; starting 256 threads there only adds throttling."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max" and int(period) > 0:
            n = max(1, min(n, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(seconds_audio=1.0, preamp_kind=0):
    """Oracle (CPU restatement, kind 'port') on the host cores: one cfg-2 instance per thread."""
    import oracle_binding as ob
    ob.lib()
    cores = effective_cpus()
    n = int(SR * seconds_audio)

    def work(k, out):
        e = ob.OracleEngine(SR, preamp_kind=preamp_kind)
        e.set_volume(0.5); e.set_tremolo_depth(0.5); e.set_speaker_character(0.0); e.set_mlp_enabled(True)
        for nn in NOTES:
            e.note_on(nn, instance_velocity(k))
        t0 = time.perf_counter()
        done = 0
        while done < n:
            e.render(BUF)
            done += BUF
        out[k] = (done, time.perf_counter() - t0)
        e.close()

    # single thread first (the reference's own single-thread design)
    r1 = {}
    work(0, r1)
    single = r1[0][0] / r1[0][1]
    res = {}
    th = [threading.Thread(target=work, args=(k, res)) for k in range(cores)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    wall = time.perf_counter() - t0
    total = sum(v[0] for v in res.values())
    return {
        "value": total / wall, "unit": "samples/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
        "sample": f"{cores} threads (= usable CPUs: affinity capped by the cgroup quota) x 1 cfg-2 instance (64 voices, full chain) x {seconds_audio:.1f} s audio, buffers of {BUF}; "
                  f"oracle = C++ f64 restatement (reference Rust is not buildable in this image)",
        "single_thread_value": single,
    }


# ---------------------------------------------------------------------------------------------------------------- launcher
def visible_gpus():
    """GPUs of this node counted WITHOUT touching HIP (the parent must stay GPU-free: it starts the ranks as children): the KFD
    topology lists one node per agent, GPU agents have simd_count > 0; HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES narrow it further.
    None when the topology is not readable (then the ranks find out themselves)."""
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for node in os.listdir(base):
            props = dict(ln.split()[:2] for ln in open(os.path.join(base, node, "properties")) if len(ln.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
    except (OSError, ValueError):
        return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def spawn_ranks(n, argv, poll_s=0.2):
    """`python bench.py --gpus N` without a launcher: start N ranks as CHILD processes (one per GPU) before anything in this process
    touches the GPU (no torch / HIP call here at all), relay rank 0's line.  All children are polled: the first one that exits non-zero
    takes the others down with it (a rank that dies before the rendezvous would otherwise leave the rest waiting for the store / RCCL
    timeout), and the launcher exits non-zero.  Never re-execs."""
    if os.environ.get("OW_BENCH_DRYRUN_BACKEND") is None:
        have = visible_gpus()
        if have is not None and have < n:
            print(f"bench.py: --gpus {n} but only {have} GPU(s) visible", file=sys.stderr)
            return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0 = []
    reader = threading.Thread(target=lambda: out0.extend(procs[0].stdout.readlines()), daemon=True)   # drain rank 0 while polling
    reader.start()
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            rc = p.poll()
            if rc not in (None, 0):
                failed = (r, rc)
                break
        time.sleep(poll_s)
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        deadline = time.time() + 10.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
    rcs = [p.wait() for p in procs]
    reader.join(timeout=5.0)
    for ln in out0:                                # rank 0's JSON line goes to stdout; library chatter (gloo / RCCL banners) to stderr
        (sys.stdout if ln.startswith("{") else sys.stderr).write(ln if ln.endswith("\n") else ln + "\n")
    sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if failed is not None or bad:
        print(f"bench.py: rank {failed[0] if failed else bad[0][0]} failed first; exit codes {rcs}", file=sys.stderr)
        return 1
    return 0


# ---------------------------------------------------------------------------------------------------------------- dry run stand-in
class DryRunPool:
    """OW_BENCH_DRYRUN_BACKEND=gloo, --workload engines: stands where an EnginePool would, so that the launcher, the barriers, the
    max-over-ranks timing, the aggregation over ranks and the JSON contract of the engines workload run on CPU
    (tests/test_bench_launcher.py).  A render sleeps a fixed 2 ms; nothing is measured and the line says so."""

    class _Eng:
        class _Diag:
            active_voices = 64; nan_guard_fires = 0; output_nan_resets = 0; preamp_nan_resets = 0; tremolo_be_fallbacks = 0

        def diag(self):
            return self._Diag()

        def set_volume(self, v): pass
        def set_tremolo_depth(self, v): pass
        def set_speaker_character(self, v): pass
        def set_mlp_enabled(self, v): pass

    def __init__(self, n):
        self.n = n
        self._block = np.zeros((n, BUF), dtype=np.float32)

    def __getitem__(self, k): return self._Eng()
    def midi(self, ev): pass
    def render(self, length, to_host=True):
        time.sleep(0.002)
        bad = os.environ.get("OW_BENCH_TEST_BAD_BLOCK_RANK") == os.environ.get("RANK", "0")      # launcher test: this rank renders a NaN
        self._block = np.full((self.n, int(length)), np.nan if bad else 0.1, dtype=np.float32)
        return self._block if to_host else None
    def render_into(self, ptr, stride, length): self.render(length, to_host=False)
    def set_profiling(self, on): pass
    def last_kernel_ms(self): return {"ops": 0.0, "voices": 1.0, "tremolo": 0.0, "preamp": 0.5, "post": 0.25}
    def power_amp_passes(self): return np.ones(self.n, dtype=np.uint32)
    def last_block(self): return self._block
    def trajectory_info(self): return (self.n, 0, 0)
    def close(self): pass


# ---------------------------------------------------------------------------------------------------------------- batch path
def batch_pass(jobs, sr, dur, dist, world, render_fn=None):
    """One pass of the sharded batch render: (elapsed_s over render + gather, max over ranks; per-phase maxima; ranks the collective saw)."""
    import torch
    from openwurli_amd import distributed as owd
    tm = {}
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    res = owd.batch_render_sharded(jobs, sr, dur, render_fn=render_fn, timings=tm, to_host=False)
    el = time.perf_counter() - t0
    vals = [el, tm["render_s"], tm["gather_s"]]
    if dist is not None:
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor(vals, dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        vals = [float(x) for x in t.tolist()]
    del res
    return vals[0], vals[1], vals[2], tm["world_seen"]


def batch_bench(dist, world, steps, warmup, scaled_jobs_per_gpu=8192, scaled_dur=1.0, render_fn=None, literal_dur=5.0):
    """configs[3] literal (512 jobs x 5 s: 8 + 64 wavefronts, latency-bound, cannot scale) and a grid scaled until every GPU holds
    `scaled_jobs_per_gpu` jobs (the regime in which >= 6x at 8 GPUs is reachable)."""
    from openwurli_amd.distributed import model_notes_job_list
    out = {}
    base = model_notes_job_list()
    grids = {"literal": (base, literal_dur),
             "scaled": ([base[i % len(base)] for i in range(scaled_jobs_per_gpu * world)], scaled_dur)}
    for name, (jobs, dur) in grids.items():
        n = int(dur * SR)
        for _ in range(warmup):
            batch_pass(jobs, SR, dur, dist, world, render_fn)
        el = rs = gs = 0.0
        seen = world
        for _ in range(steps):
            e, r, g, seen = batch_pass(jobs, SR, dur, dist, world, render_fn)
            el += e; rs += r; gs += g
        samples = len(jobs) * n * steps
        out[name] = {"jobs": len(jobs), "seconds_per_job": dur, "samples_per_s": samples / el, "x_realtime": samples / el / SR,
                     "render_ms": 1e3 * rs / steps, "gather_ms": 1e3 * gs / steps, "pass_ms": 1e3 * el / steps, "ranks_seen_by_collective": seen,
                     "scaling": "strong" if name == "literal" else "weak",
                     "valu_frac": FLOPS_BATCH_JOB * samples / el / world / 1e12 / PEAK_FP64_VALU_TFLOPS}
    return out


# ---------------------------------------------------------------------------------------------------------------- main
def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", choices=["engines", "batch"], default="engines",
                    help="engines = configs[1] replicated per configs[4] (the metric's config); batch = configs[3] sharded over the ranks")
    ap.add_argument("--instances", type=int, default=int(os.environ.get("OW_BENCH_INSTANCES", "131072")), help="engine instances per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--deliver", choices=["hbm", "host"], default="hbm",
                    help="hbm (default, the contract's `value`): every block stays in HBM; host: every block of the timed region is delivered into a "
                         "pinned host block, as render(&mut [f32]) hands it back (engine.rs:425-462) -- the 10 s run in this mode IS the API-faithful figure")
    ap.add_argument("--no-extras", action="store_true", help="skip pcie_inclusive / config4_literal / single_instance / batch objects")
    ap.add_argument("--preamp", choices=["legacy", "melange"], default="legacy",
                    help="legacy = the 8-node DK solver of the reference's default build (the metric's config); melange = the generated "
                         "12-node solver of its `--features melange-preamp` build")
    ap.add_argument("--power-amp", choices=["behavioral", "melange"], default="behavioral",
                    help="behavioral = the closed-loop Newton amp of the reference's default build (the metric's config); melange = the generated "
                         "7-BJT Class-AB solver + rail dynamics of its `--no-default-features` build")
    ap.add_argument("--host-rate", type=float, default=48000.0,
                    help="host sample rate: 48000 = BASELINE configs[1] (the metric's config, default); 96000 = configs[2] (no oversampling)")
    ap.add_argument("--tremolo-groups", type=int, default=0,
                    help="tremolo phase groups of the pool.  0 (default) = one per instance: every instance runs its own Twin-T oscillator, as "
                         "independently created plugin instances do (the pool is staggered with the test hook after the warm-up); 1 = a fresh pool "
                         "as it is: every instance was built at the same sample, all share one phase and ONE oscillator is computed per pool (the "
                         "best case, reported as the `shared_tremolo_phase` extra of the default run); G = that many groups")
    args = ap.parse_args(argv)
    if args.steps is None:
        args.steps = 100 if args.workload == "engines" else 3
    if args.warmup is None:
        args.warmup = 5 if args.workload == "engines" else 1

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return spawn_ranks(args.gpus, argv)

    global SR, EPOCH, FLOPS_TREMOLO, FLOPS_PREAMP, FLOPS_POST
    if args.host_rate != SR:
        SR = float(args.host_rate)
        EPOCH = int(SR)                              # 1.0 s re-strike epochs at any rate
        osr = 2 if SR < 88200.0 else 1               # engine.rs:195
        FLOPS_TREMOLO = osr * (1000 + 25)
        FLOPS_PREAMP = osr * 1400 + (24 if osr == 2 else 0)
        FLOPS_POST = osr * 90 + (24 if osr == 2 else 0) + 45

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("OW_BENCH_DRYRUN_BACKEND") and os.environ.get("OW_BENCH_TEST_FAIL_RANK") == str(rank):
        return 3                                     # launcher test: this rank dies before the rendezvous, the others would wait for it
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the launcher decides; reporting n_gpus={world}", file=sys.stderr)
    import datetime
    import torch
    RENDEZVOUS_TIMEOUT = datetime.timedelta(seconds=float(os.environ.get("OW_BENCH_RENDEZVOUS_TIMEOUT_S", "300")))
    dist = None
    # OW_BENCH_DRYRUN_BACKEND=gloo: launcher / sharding / gather / JSON-contract dry run on CPU (tests/test_bench_launcher.py): the ranks
    # rendezvous over gloo and the batch render is a deterministic stand-in.  Nothing is measured in that mode and the line says so.
    dryrun = os.environ.get("OW_BENCH_DRYRUN_BACKEND")
    if dryrun:
        if world > 1:
            import torch.distributed as dist
            dist.init_process_group(dryrun, rank=rank, world_size=world, timeout=RENDEZVOUS_TIMEOUT)
    else:
        torch.cuda.set_device(local_rank)
        if world > 1:
            import torch.distributed as dist
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=RENDEZVOUS_TIMEOUT)

    import openwurli_amd as ow

    def barrier():
        if dist is not None:
            dist.barrier()
        if not dryrun:
            torch.cuda.synchronize()

    preamp_kind = 1 if args.preamp == "melange" else 0
    flops_preamp = ((FLOPS_PREAMP_MELANGE if os.environ.get("OW_MEL_RANK1", "0") not in ("", "0") else FLOPS_PREAMP_MELANGE_LIT)
                    if preamp_kind else FLOPS_PREAMP)
    n_inst = args.instances
    line = None

    if args.workload == "batch":
        barrier()
        fake = None
        if dryrun:
            def fake(jobs, n=int(0.01 * SR)):
                return np.stack([np.full(n, j["note"] * 1000 + j["velocity"], dtype=np.float32) for j in jobs])
        b = batch_bench(dist, world, args.steps, args.warmup, render_fn=fake, **({"scaled_jobs_per_gpu": 64, "scaled_dur": 0.01, "literal_dur": 0.01} if dryrun else {}))
        if rank == 0:
            lit = b["literal"]
            line = {
                "metric": "audio samples/s, batch-render path (ml/render_model_notes.py grid, preamp-bench render semantics)",
                "value": lit["samples_per_s"], "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": lit["pass_ms"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
                "data": "synthetic" if not dryrun else "DRY RUN (gloo, stand-in renderer): launcher and gather plumbing only, nothing measured",
                "config": {"workload": "cfg4: 512 jobs (64 notes x 8 velocities) x 5 s at 48 kHz, job j -> rank j mod G, one RCCL gather of f32 slabs to rank 0",
                           "parallelism": f"{world} rank(s), static job deal, no data-path collective besides the final gather"},
                # what scales and what cannot: the literal grid is STRONG scaling of a latency-bound job -- 512 jobs leave 7/8 of one chip idle and
                # a serial stream does not get shorter on more chips, so its N-GPU figure cannot exceed ~1 x the 1-GPU one; the scaled grid
                # (the same jobs repeated until every GPU holds 8 192) is the weak-scaling figure of this path
                "scaling_figure": {"grid": "scaled", "scaling": "weak", "samples_per_s": b["scaled"]["samples_per_s"], "jobs": b["scaled"]["jobs"],
                                   "pass_ms": b["scaled"]["pass_ms"], "gather_ms": b["scaled"]["gather_ms"]},
                "scaling_note": ("`value` is the LITERAL grid (512 jobs x 5 s): strong scaling of a latency-bound job, by construction <= ~1 x at any N; "
                                 "`scaling_figure` (the scaled grid, 8 192 jobs per GPU) is the number to compare across N"),
                "batch": b,
                "dry_run": bool(dryrun),
                "roofline": {"bound": "valu_f64", "kernel": "k_job_chain_row", "achieved": lit["valu_frac"] * PEAK_FP64_VALU_TFLOPS, "peak": PEAK_FP64_VALU_TFLOPS,
                             "unit": "TFLOP/s", "frac": lit["valu_frac"], "traffic": None,
                             "note": "512 jobs = 64 workgroups of five wavefronts (four row-of-sixteen preamp wavefronts | output stage) on 1 024 SIMDs, the voices rendered beside them: bounded by the serial instruction stream of one preamp state, not by chip-wide issue or HBM"},
                "cpu_baseline": None,
            }
    else:
        pa_kind = 1 if args.power_amp == "melange" else 0
        phases = n_inst if args.tremolo_groups <= 0 else max(1, min(args.tremolo_groups, n_inst))
        red_dev = "cpu" if dryrun else "cuda"
        ranks_seen = 1
        if dist is not None:
            t_ = torch.ones(1, dtype=torch.float64, device=red_dev)
            dist.all_reduce(t_, op=dist.ReduceOp.SUM)
            ranks_seen = int(round(float(t_.item())))

        def make_pool(n, sr=None, preamp=preamp_kind, pa=pa_kind, n_phases=1, trajectory=True):
            """the plugin's initialize() for n instances: chain build + 0.6 s warm-up (not timed), then the tremolo phases.
            trajectory=False: a pool created under OW_TREM_TRAJ=0 -- one Twin-T oscillator per phase group instead of the shared
            trajectory (rounds 1-3); the variable is read when the pool is created and restored right after."""
            if dryrun:
                return DryRunPool(n)
            sr = SR if sr is None else sr
            old = os.environ.get("OW_TREM_TRAJ")
            try:
                if not trajectory:
                    os.environ["OW_TREM_TRAJ"] = "0"
                p = ow.EnginePool(sr, n, device=local_rank, preamp_kind=preamp, power_amp_kind=pa)
            finally:
                if not trajectory:
                    if old is None:
                        os.environ.pop("OW_TREM_TRAJ", None)
                    else:
                        os.environ["OW_TREM_TRAJ"] = old
            p.set_sample_rate(sr)
            p.ensure_buffer_capacity(BUF)
            if n_phases > 1:
                p.stagger_tremolo(min(n_phases, n))
            # volume 0.5 / tremolo depth 0.5 / speaker character 0.0 / MLP on are the engine defaults (engine.rs:221-224)
            for k in range(min(n, 4096)):
                e = p[k]
                e.set_volume(0.5); e.set_tremolo_depth(0.5); e.set_speaker_character(0.0); e.set_mlp_enabled(True)
            return p

        def timed_steps(sc, k_steps, profile=False):
            """k_steps of the script between barriers; max over ranks"""
            barrier()
            ta = time.perf_counter()
            trace = [] if os.environ.get("OW_BENCH_STEP_MS") else None      # (diagnostic: wall time of every step to stderr)
            for _ in range(k_steps):
                sc.step(profile=profile)
                if trace is not None:
                    trace.append(time.perf_counter())
            tb = time.perf_counter()
            barrier()
            el_ = time.perf_counter() - ta
            if trace is not None:
                ms_ = np.diff(np.array([ta] + trace)) * 1e3
                print(f"[step ms] {k_steps} steps: " + " ".join(f"{x:.2f}" for x in ms_) + f" | closing barrier {1e3 * (time.perf_counter() - tb):.2f} ms", file=sys.stderr)
            if dist is not None:
                t_ = torch.tensor([el_], dtype=torch.float64, device=red_dev)
                dist.all_reduce(t_, op=dist.ReduceOp.MAX)
                el_ = float(t_.item())
            return el_

        def side_run(p, n, k_warm, k_steps, epoch=None, buf=BUF, pos=None):
            """a short profiled run of another configuration (extras): value, ms per step and per kernel"""
            sc = Script(p, n, buf=buf, epoch=epoch)
            if pos is not None:
                sc.pos = pos
            for _ in range(k_warm):
                sc.step()
            p.set_profiling(True)
            el_ = timed_steps(sc, k_steps, profile=True)
            p.set_profiling(False)
            kms_ = sc.kernel_ms / max(sc.kernel_launches, 1)
            return {"value": k_steps * buf * n * world / el_, "unit": "samples/s", "ms_per_step": 1e3 * el_ / k_steps, "steps": k_steps, "warmup": k_warm,
                    "instances_per_gpu": n, "kernel_ms_per_step": {nm: float(x) for nm, x in zip(("ops", "voices", "tremolo", "preamp", "post"), kms_)}}

        def verify(p, n):
            """What the timed region rendered, looked at: every row of the last block finite and inside the level band of a 64-voice
            chord at volume 0.5, eight engines spread over the pool with all 64 voices alive (the 1.0 s re-strike keeps them: the shortest
            note reaches -80 dB after 1.4 s) and no NaN-guard / reset activity.  Returns (ok, details)."""
            blk = p.last_block()
            finite = bool(np.all(np.isfinite(blk)))
            peaks = np.max(np.abs(blk), axis=1) if blk.size else np.zeros(n)
            lo, hi = float(peaks.min()), float(peaks.max())
            picks = sorted({k for k in (0, 1, 31, 32, 4095, 4096, n // 2, n - 1) if 0 <= k < n})
            diags = [p[k].diag() for k in picks]
            voices = [int(d.active_voices) for d in diags]
            counters = int(sum(d.nan_guard_fires + d.output_nan_resets + d.preamp_nan_resets for d in diags))
            ok = finite and lo > 1e-3 and hi < 4.0 and all(v == 64 for v in voices) and counters == 0
            return ok, {"rows_checked": int(blk.shape[0]), "all_finite": finite, "row_peak_min": lo, "row_peak_max": hi, "engines_checked": picks,
                        "active_voices": voices, "nan_guard_and_reset_counters": counters}

        pool = make_pool(n_inst, n_phases=phases)
        main_host = pool.alloc_host_block(BUF) if (args.deliver == "host" and not dryrun) else None
        script = Script(pool, n_inst, host_out=main_host)
        for _ in range(args.warmup):
            script.step()
        pool.set_profiling(True)
        elapsed = timed_steps(script, args.steps, profile=True)
        # melange power amp: Newton passes per chain-rate sample, mean over the engines, in the LAST block of the timed region
        pa_passes = float(pool.power_amp_passes().mean()) / (BUF * (2 if SR < 88200.0 else 1)) if pa_kind else None
        pool.set_profiling(False)
        verified, verify_details = verify(pool, n_inst)
        on_traj = pool.trajectory_info()[0]
        failed_ranks = [] if verified else [rank]
        if dist is not None:
            t_ = torch.zeros(world, dtype=torch.float64, device=red_dev)     # one slot per rank: which ranks failed, not just whether one did
            t_[rank] = 0.0 if verified else 1.0
            dist.all_reduce(t_, op=dist.ReduceOp.SUM)
            failed_ranks = [r for r, x in enumerate(t_.tolist()) if x > 0.5]
            verified = not failed_ranks

        extras = {}
        if not args.no_extras and not dryrun:
            # (a) PCIe-inclusive: the same steps with every block delivered into a pinned host buffer (what render(&mut [f32]) hands back,
            # engine.rs:425-462): big pools store the rows from the chain kernel itself (k_chain_stream), nothing is copied afterwards
            host = pool.alloc_host_block(BUF)
            sp = Script(pool, n_inst, host_out=host)
            sp.pos = script.pos
            k_steps = max(5, min(args.steps, 20))
            for _ in range(2):
                sp.step()
            el = timed_steps(sp, k_steps)
            extras["pcie_inclusive"] = {"value": k_steps * BUF * n_inst * world / el, "unit": "samples/s", "ms_per_step": 1e3 * el / k_steps, "steps": k_steps,
                                        "bytes_per_step_per_gpu": 4 * BUF * n_inst, "host_memory": "pinned (ow_host_alloc), mapped: rows stored by the output stage itself",
                                        "restrikes_in_timed_region": int((sp.pos // EPOCH) - ((sp.pos - k_steps * BUF) // EPOCH))}
            # (a1) ten steps that straddle a re-strike epoch (note_off + note_on of all 64 keys of every instance: 5 ms steal crossfades,
            # onset ramps and attack noise in the general voice kernel, host MIDI + op upload) -- the default timed region may hold none;
            # once with the audio left in HBM, once delivered to the host block
            r = side_run(pool, n_inst, 0, 10, pos=EPOCH - 3 * BUF)
            r["restrikes_in_timed_region"] = 1
            r["note"] = "steps [epoch - 3 buffers, epoch + 7 buffers): one whole-keyboard re-strike of every instance inside"
            extras["with_restrike"] = r
            sh = Script(pool, n_inst, host_out=host)
            sh.pos = EPOCH - 3 * BUF
            el = timed_steps(sh, 10)
            extras["with_restrike_host"] = {"value": 10 * BUF * n_inst * world / el, "unit": "samples/s", "ms_per_step": 1e3 * el / 10, "steps": 10,
                                            "restrikes_in_timed_region": 1, "note": "the same ten steps delivered to the pinned host block"}
            pool.free_host_block(host)
        if main_host is not None:
            pool.free_host_block(main_host)
        pool.close()

        if not args.no_extras and not dryrun and phases * 4 >= n_inst:
            # (a2) the per-instance oscillators of rounds 1-3 on the same workload: a pool created without the shared trajectory
            # (OW_TREM_TRAJ=0), staggered to one oscillator per instance; then the same with the block-ahead oscillators serialised in front
            # of the voices (every kernel's HIP-event interval is then its own time)
            pt = make_pool(n_inst, n_phases=phases, trajectory=False)
            r = side_run(pt, n_inst, 8, 10)
            r["tremolo"] = "per_instance_oscillator"
            r["tremolo_oscillators"] = phases
            r["note"] = "OW_TREM_TRAJ=0 at pool creation: one Twin-T oscillator per instance (k_tremolo, lane = oscillator, own stream beside the voices)"
            extras["tremolo_per_instance"] = r
            pt.set_switch("trem_serial", 1)
            r = side_run(pt, n_inst, 2, 10, pos=12 * BUF)
            pt.set_switch("trem_serial", 0)
            ks_ = r["kernel_ms_per_step"]
            r["voices_frac_of_fp64_peak"] = FLOPS_VOICES * BUF * n_inst / (ks_["voices"] * 1e-3) / 1e12 / PEAK_FP64_VALU_TFLOPS if ks_["voices"] > 0 else None
            r["tremolo_frac_of_fp64_peak"] = FLOPS_TREMOLO * BUF * phases / (ks_["tremolo"] * 1e-3) / 1e12 / PEAK_FP64_VALU_TFLOPS if ks_["tremolo"] > 0 else None
            r["note"] = "the same pool with the `trem_serial` switch (test hook): the voices wait for the block-ahead tremolo kernel instead of overlapping it"
            extras["tremolo_per_instance_serialised"] = r
            pt.close()

        if not args.no_extras and not dryrun and SR == 48000.0 and not preamp_kind and not pa_kind:
            # (a3) the neighbouring BASELINE configs on the same pool size, short runs: configs[2] (96 kHz host, no oversampling) and the
            # melange 12-node preamp the north star names (literal per-sample rebuild)
            p3 = make_pool(n_inst, sr=96000.0, n_phases=phases)
            r = side_run(p3, n_inst, 8, 10, epoch=96000)      # 8 warm-up blocks: past the onset ramps and the 15 ms attack noise of the strike
            r["x_realtime_at_96k"] = r["value"] / 96000.0
            r["workload"] = "cfg3: 64-voice all-keys, 96 kHz host (no oversampling), full chain, MLP on, buffers of 512"
            extras["config3"] = r
            p3.close()
            n_mel = min(n_inst, 65536)
            pm = make_pool(n_mel, preamp=1, n_phases=min(phases, n_mel))
            r = side_run(pm, n_mel, 8, 6)
            r["workload"] = "cfg2 with the melange 12-node preamp (k_preamp_mel_col: the reference's rebuild_matrices + invert_n per chain-rate sample, column-streamed)"
            r["preamp_frac_of_fp64_peak"] = FLOPS_PREAMP_MELANGE_LIT * BUF * n_mel / (r["kernel_ms_per_step"]["preamp"] * 1e-3) / 1e12 / PEAK_FP64_VALU_TFLOPS
            extras["preamp_melange"] = r
            pm.close()
            n_pa = min(n_inst, 131072)
            pp = make_pool(n_pa, pa=1, n_phases=min(phases, n_pa))
            r = side_run(pp, n_pa, 3, 3)
            r["workload"] = ("cfg2 with the melange 7-BJT power amp + rail sag (k_post_mpa: eight lanes per engine, every engine on its own sample counter, "
                             "engines dispatched by falling Newton demand of their last block)")
            r["x_realtime_aggregate"] = r["value"] / SR
            r["newton_passes_per_chain_sample"] = float(pp.power_amp_passes().mean()) / (BUF * 2)
            r["post_frac_of_fp64_peak"] = ((2 * ((8 * 300 + 1024 + 2730 + 256 + 1024) * r["newton_passes_per_chain_sample"] + 800 + 640) + 24 + 45) * BUF * n_pa
                                           / (r["kernel_ms_per_step"]["post"] * 1e-3) / 1e12 / PEAK_FP64_VALU_TFLOPS)
            extras["power_amp_melange"] = r
            pp.close()

        single = None
        cpu = None
        if rank == 0 and not args.no_extras and not dryrun:
            lib = ow.load_library()

            def run256(cold):
                """configs[4] taken literally: ONE pool of 256 instances on this GPU with 256 tremolo phases.  cold: the process-wide
                trajectory store is dropped first, so the pool's oldest instance has to extend it as it goes (one serial oscillator:
                the block time); otherwise the store already reaches past the run (any earlier instance of this process got there,
                or ow_tremolo_prefetch did)."""
                if cold:
                    lib.ow_test_clear_settle_caches()
                p256 = make_pool(256, n_phases=min(phases, 256))
                s256 = Script(p256, 256)
                for _ in range(3):
                    s256.step()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(30):
                    s256.step()
                el_ = time.perf_counter() - t1
                held = p256.trajectory_info()[1]
                p256.close()
                return {"value": 30 * BUF * 256 / el_, "unit": "samples/s", "x_realtime_aggregate": 30 * BUF * 256 / el_ / SR, "ms_per_step": 1e3 * el_ / 30,
                        "trajectory_samples_in_hbm_after": held}

            # (b) the warm figure first: the store of this chain rate reaches far past the run (the big pools above extended it)
            c4 = run256(cold=False)
            c4.update({"instances": 256, "tremolo_phases": min(phases, 256), "preamp": args.preamp, "power_amp": args.power_amp,
                       "trajectory": "already in HBM for the whole run (extended by earlier instances of the process)"})
            extras["config4_literal"] = c4
            # (c) one instance (what one plugin instance sees): per-buffer latency at the usual host buffer sizes, audio copied to the host
            table = []
            for buf in (64, 128, 256, 512):
                one = make_pool(1)
                one.ensure_buffer_capacity(buf)
                s1 = Script(one, 1, buf=buf)
                host1 = np.zeros((1, buf), dtype=np.float32)
                lat = []
                for i in range(8 + 60):
                    ta = time.perf_counter()
                    if s1.pos % EPOCH == 0:
                        one.midi(s1.ev_strike if s1.pos == 0 else s1.ev_restrike)
                    one.render_into(host1.ctypes.data, buf, buf)
                    s1.pos += buf
                    if i >= 8:
                        lat.append(time.perf_counter() - ta)
                lat = np.array(lat) * 1e6
                # ... and PACED like a real-time host: one call per buffer period
                period = buf / SR
                paced = []
                t_next = time.perf_counter() + period
                for i in range(8 + 40):
                    while time.perf_counter() < t_next:
                        pass
                    ta = time.perf_counter()
                    if s1.pos % EPOCH == 0:
                        one.midi(s1.ev_strike if s1.pos == 0 else s1.ev_restrike)
                    one.render_into(host1.ctypes.data, buf, buf)
                    s1.pos += buf
                    if i >= 8:
                        paced.append(time.perf_counter() - ta)
                    t_next += period
                paced = np.array(paced) * 1e6
                table.append({"buffer": buf, "latency_us_mean": float(lat.mean()), "latency_us_p50": float(np.median(lat)), "latency_us_max": float(lat.max()),
                              "samples_per_s": buf / (lat.mean() * 1e-6), "x_realtime": buf / (lat.mean() * 1e-6) / SR,
                              "buffer_period_us": 1e6 * buf / SR,
                              "paced_latency_us_mean": float(paced.mean()), "paced_latency_us_max": float(paced.max()),
                              "paced_load": float(paced.mean() / (1e6 * period))})
                one.close()
            extras["single_instance"] = table
            extras["single_instance_config"] = {"preamp": args.preamp, "power_amp": args.power_amp,
                                                "trajectory": "already in HBM (back to back a lone instance outruns the one oscillator that extends it; paced at real time it never does)"}
            single = table[-1]["samples_per_s"]
            # (d) instantiation and config 1: a second engine of the process (settled states cached), and Voice::render_note for 60 s
            # (tools/reed-renderer; the reference publishes 0.08 s for it, CHANGELOG.md:185) next to the oracle on one host thread
            t1 = time.perf_counter()
            e2 = ow.WurliEngine(SR, device=local_rank, preamp_kind=preamp_kind, power_amp_kind=pa_kind)
            t2 = time.perf_counter()
            e2.set_sample_rate(SR)
            t3 = time.perf_counter()
            e2.close()
            extras["instantiate_ms"] = {"engine_new": 1e3 * (t2 - t1), "set_sample_rate_incl_0.6s_warm_up": 1e3 * (t3 - t2)}
            t1 = time.perf_counter()
            rn = ow.render_note(60, 100 / 127.0, 60.0, SR, device=local_rank)
            t2 = time.perf_counter()
            rn_cpu_ms = None
            if not args.no_cpu_baseline:
                import oracle_binding as ob
                t3 = time.perf_counter()
                ob.render_note(60, 100 / 127.0, 60.0, SR)
                rn_cpu_ms = 1e3 * (time.perf_counter() - t3)
            extras["render_note_60s_ms"] = {"gpu": 1e3 * (t2 - t1), "cpu_oracle_one_thread": rn_cpu_ms, "samples": int(rn.size),
                                            "note": "one voice = one lane: a serial recurrence, latency-bound on a GPU; voices only (no chain launches, no settle)"}
            # (e) the cold counterparts, last (they drop the process-wide store): the 256-instance pool and a lone instance whose own
            # oldest engine has to extend the trajectory while it renders -- the serial rate of ONE Twin-T oscillator on a quad of lanes
            c4c = run256(cold=True)
            c4c["trajectory"] = "cold: extended by the pool's oldest instance while it renders"
            extras["config4_literal"]["cold"] = c4c
            lib.ow_test_clear_settle_caches()
            one = make_pool(1)
            s1 = Script(one, 1, buf=64)
            host1 = np.zeros((1, 64), dtype=np.float32)
            lat = []
            for i in range(8 + 60):
                ta = time.perf_counter()
                if s1.pos % EPOCH == 0:
                    one.midi(s1.ev_strike if s1.pos == 0 else s1.ev_restrike)
                one.render_into(host1.ctypes.data, 64, 64)
                s1.pos += 64
                if i >= 8:
                    lat.append(time.perf_counter() - ta)
            one.close()
            extras["single_instance_cold_64"] = {"latency_us_mean": float(np.mean(lat) * 1e6), "x_realtime": 64 / float(np.mean(lat)) / SR,
                                                 "note": "64-sample buffers back to back on a fresh store: the instance waits for its own oscillator"}
        if not args.no_extras and not dryrun:
            barrier()
            extras["batch"] = batch_bench(dist, world, steps=2, warmup=1)
        if rank == 0 and world == 1 and not args.no_cpu_baseline and not dryrun:
            cpu = cpu_baseline(preamp_kind=preamp_kind)

        if rank == 0:
            total_samples = args.steps * BUF * n_inst * world
            value = total_samples / elapsed
            kms = script.kernel_ms / max(script.kernel_launches, 1)     # average ms per step, per kernel
            names = ["ops", "voices", "tremolo", "preamp", "post"]
            shared = on_traj == n_inst
            # shared trajectory: ONE oscillator per (device, chain rate) whatever the pool size -- not a per-engine cost, not in the numerator
            trem_per_engine = 0.0 if shared else FLOPS_TREMOLO * phases / n_inst
            osr = 2 if SR < 88200.0 else 1
            flops_pa_sample = (8 * 300 + 1024 + 2730 + 256 + 1024) * (pa_passes or 1.0) + 800 + 640        # per pass: devices, Jacobian, LU, substitution, K products
            flops_post = (FLOPS_POST - osr * 90 + osr * flops_pa_sample) if pa_kind else FLOPS_POST
            flops = {"ops": 0.0, "voices": FLOPS_VOICES, "tremolo": trem_per_engine, "preamp": flops_preamp, "post": flops_post}
            per_sample = FLOPS_VOICES + trem_per_engine + flops_preamp + flops_post
            audio = ["voices", "preamp", "post"] + (["tremolo"] if (not shared and phases * 4 >= n_inst) else [])
            dom = max(audio, key=lambda k: kms[names.index(k)])
            dom_ms = float(kms[names.index(dom)])
            achieved = flops[dom] * BUF * n_inst / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
            traffic = None
            tp = os.path.join(ROOT, "profiles", "hbm_traffic.json")
            if os.path.exists(tp):
                try:
                    tj = json.load(open(tp))      # PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE) measured per engine
                    per_engine = tj.get("kernels", {}).get(dom, {}).get("hbm_bytes_per_engine_launch")
                    traffic = per_engine * n_inst if per_engine is not None else None
                except Exception:
                    traffic = None
            solver = "melange 12-node DK" if preamp_kind else "legacy DK"
            amp = "melange 7-BJT power amp + rail sag" if pa_kind else "behavioural power amp"
            pcie = extras.get("pcie_inclusive", {}).get("value")

            def epoch_weighted(window_ms, restrike_ms, steps=None, restrikes=0):
                """samples/s over one config-2 epoch (48 000 samples = 93.75 buffers): 83.75 steady buffers + the 10 around the re-strike.
                A timed window that itself held `restrikes` re-strikes (the 100-step default does) is first taken apart:
                steps x window = (steps - 10 R) x steady + 10 R x restrike."""
                if not window_ms or not restrike_ms:
                    return None
                steady_ms = window_ms
                if restrikes and steps and steps > 10 * restrikes:
                    steady_ms = (steps * window_ms - 10 * restrikes * restrike_ms) / (steps - 10 * restrikes)
                per_epoch_ms = (EPOCH / BUF - 10) * steady_ms + 10 * restrike_ms
                return {"value": EPOCH * n_inst * world / (per_epoch_ms * 1e-3), "unit": "samples/s", "ms_per_step": per_epoch_ms / (EPOCH / BUF),
                        "steady_ms_per_step": steady_ms, "restrike_window_ms_per_step": restrike_ms}
            line = {
                "metric": f"audio samples/s, 64-voice full chain (x real-time @{SR / 1000:.0f} kHz = value / {SR:.0f})",
                "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f64",
                "data": "synthetic" if not dryrun else "DRY RUN (gloo, stand-in pool): launcher, barriers and aggregation only, nothing measured",
                "verified": verified, "verify": verify_details, "verify_failed_ranks": failed_ranks, "ranks_seen": ranks_seen, "dry_run": bool(dryrun),
                "config": {
                    "workload": (f"cfg2: 64-voice all-keys-sustained (1.0 s re-strike), 48 kHz host / 96 kHz chain, full chain "
                                 f"(tremolo+{solver} preamp+{amp}+speaker), MLP on, buffers of 512") if SR == 48000.0 else
                                (f"cfg3: 64-voice all-keys-sustained (1.0 s re-strike), {SR:.0f} Hz host"
                                 f"{' (no oversampling)' if SR >= 88200.0 else ' / 2x chain'}, full chain, MLP on, buffers of 512"),
                    "instances_per_gpu": n_inst, "buffer": BUF, "parallelism": f"{world} x independent pools (no data-path collective)",
                    "tremolo": "shared_trajectory" if shared else "per_group_oscillator",
                    "tremolo_phases": phases,
                    "tremolo_note": ((f"every instance stands at its own t ({phases} decorrelated phases) of ONE Twin-T / CdS trajectory per (device, chain rate) kept in HBM "
                                      "and extended by a single oscillator; bit-identical to one oscillator per instance (tests/test_gpu_trajectory.py), whose "
                                      "figure is the `tremolo_per_instance` extra; the oscillator's flops are not in any numerator") if shared else
                                     f"{phases} tremolo phase groups, one Twin-T oscillator each (its flops are in whole_chain_frac)"),
                    "restrikes_in_timed_region": int((script.pos // EPOCH) - ((script.pos - args.steps * BUF) // EPOCH)),
                    "audio_left_in_hbm": args.deliver != "host",
                    "pcie_inclusive_samples_per_s": pcie,
                    "pcie_note": "`value` leaves every block in HBM; pcie_inclusive delivers each block into a pinned host buffer -- the render(&mut [f32]) equivalent",
                    # SURVEY 8d config 2 as defined: all 64 keys re-struck every 48 000 samples.  One epoch = 93.75 buffers of 512; the ten
                    # buffers around the re-strike are the `with_restrike` window, the other 83.75 run at the steady rate of the timed region
                    "config2_epoch_weighted": epoch_weighted(1e3 * elapsed / args.steps, extras.get("with_restrike", {}).get("ms_per_step"), args.steps,
                                                             int((script.pos // EPOCH) - ((script.pos - args.steps * BUF) // EPOCH))),
                    # ... and with every block delivered to the caller's host buffer (engine.rs:425-462): what a caller of render(&mut [f32]) gets
                    "api_faithful": epoch_weighted(extras.get("pcie_inclusive", {}).get("ms_per_step"), extras.get("with_restrike_host", {}).get("ms_per_step"),
                                                   extras.get("pcie_inclusive", {}).get("steps"), extras.get("pcie_inclusive", {}).get("restrikes_in_timed_region", 0)),
                },
                "x_realtime_aggregate": value / SR,
                "host_midi_s": script.t_midi, "render_calls_s": script.t_render, "elapsed_s": elapsed,
                "single_instance_samples_per_s": single,
                "roofline": {
                    "bound": "valu_f64", "kernel": KERNEL_OF[dom] + (("_mel" if flops_preamp == FLOPS_PREAMP_MELANGE else "_mel_lit") if preamp_kind and dom == "preamp" else "") + ("_mpa" if pa_kind and dom == "post" else ""),
                    "achieved": achieved,
                    "peak": PEAK_FP64_VALU_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP64_VALU_TFLOPS, "traffic": traffic,
                    "traffic_source": ("profiles/hbm_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes on a 4 096-engine pool, corrected per the guide, "
                                       "scaled by instances_per_gpu -- not measured in this run"),
                    "kernel_ms_per_step": {n: float(k) for n, k in zip(names, kms)},
                    "flops_per_output_sample": {"voices": FLOPS_VOICES, "tremolo": trem_per_engine, "preamp": flops_preamp, "post": flops_post},
                    "flops_per_voice_sample": FLOPS_VOICE_SAMPLE, "flops_executed_per_voice_sample": FLOPS_VOICE_SAMPLE_EXECUTED,
                    "frac_executed": (64 * FLOPS_VOICE_SAMPLE_EXECUTED * BUF * n_inst / (float(kms[1]) * 1e-3) / 1e12 / PEAK_FP64_VALU_TFLOPS
                                      if kms[1] > 0 else None),
                    "whole_chain_frac": per_sample * value / world / 1e12 / PEAK_FP64_VALU_TFLOPS,
                    # the contract's own vocabulary, for reference: PMC HBM bytes of the dominant kernel / its duration against 8 TB/s
                    "hbm": ({"achieved": traffic / (dom_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                             "frac": traffic / (dom_ms * 1e-3) / 1e9 / 8000.0} if traffic and dom_ms > 0 else None),
                    "note": "path is FP64-VALU/latency bound (not HBM, not MFMA); achieved = algorithmic f64 flops of the dominant "
                            "kernel per launch / its HIP-event duration on the pool stream" +
                            (f"; melange power amp: the flop count is per-pass work x the {pa_passes:.2f} Newton passes per chain-rate sample the engines "
                             "spent on the last block (mean; data dependent)" if pa_kind else ""),
                    "power_amp_newton_passes_per_chain_sample": pa_passes,
                },
                "cpu_baseline": cpu,
            }
            line.update(extras)
            # the contract-faithful figures where a reader of the first keys finds them (they also stay inside `config` / the extras)
            front = {"config2_epoch_weighted": line["config"].get("config2_epoch_weighted"), "api_faithful": line["config"].get("api_faithful"),
                     "configs4_fresh": ({"instances": 256, "x_realtime_aggregate": extras["config4_literal"]["cold"]["x_realtime_aggregate"],
                                         "ms_per_step": extras["config4_literal"]["cold"]["ms_per_step"],
                                         "warm_x_realtime_aggregate": extras["config4_literal"]["x_realtime_aggregate"],
                                         "note": "BASELINE configs[4] literally: ONE pool of 256 instances, 30 steps right after every process-wide store was dropped "
                                                 "(no import, no prefetch: the pool's oldest instance extends the trajectory as it renders); warm = the store already in HBM"}
                                        if "cold" in extras.get("config4_literal", {}) else None)}
            head = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
            line = {**{k: line[k] for k in head}, **front, **{k: v for k, v in line.items() if k not in head}}
    rc = 0
    if line is not None and line.get("verified") is False:
        print("bench.py: the rendered block failed verification: " + json.dumps(line.get("verify")), file=sys.stderr)
        rc = 4
    if rank == 0 and line is not None:
        print(json.dumps(line))
        sys.stdout.flush()
    if dist is not None:
        dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main())
