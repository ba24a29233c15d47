#!/usr/bin/env python3
"""Throughput bench of the OpenWurli DSP hot path on MI355X.

Workload (BASELINE.json configs[1], SURVEY.md 8d config 2, scaled out per configs[4]):
  I independent engine instances per GPU, each 64 voices (all keys 33..96 struck at t=0,
  note_off+note_on re-strike of all 64 every 1.0 s so that 64 voices stay alive), 48 kHz host
  rate (2x oversampled chain at 96 kHz), full chain: tremolo + legacy DK preamp + behavioural
  power amp + speaker, volume 0.5, tremolo depth 0.5, MLP on, buffers of 512 samples, events
  split sample-accurately like the plugin does (plugin/src/lib.rs:128-149).
  Instance k plays velocity (40 + (37 k mod 88))/127 (config 5 decorrelation).

One "step" = one 512-sample buffer rendered by every instance of every rank.  `value` =
output samples per second over all instances of all ranks (weak scaling: I per GPU fixed).
The rendered audio stays in HBM (inputs resident, no PCIe in the timed region except the
per-step event list and the 32-byte-per-engine status block the host state machine needs).

Prints ONE JSON line (rank 0).  See DESIGN.md "Measurement" for the roofline arithmetic.
"""
import argparse
import json
import os
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
FLOPS_VOICES = 64 * 130            # 7 modes x 16 + jitter 3 + pickup 13 + gain/sum 2 per voice-sample
FLOPS_TREMOLO = 2 * (1000 + 25)    # Twin-T NR step + LDR law per OS sample, 2 OS samples
FLOPS_PREAMP = 2 * 1400 + 24       # main+shadow dk_step per OS sample + half-band up
FLOPS_POST = 2 * 90 + 24 + 45      # power amp x2 + half-band down + speaker/gain
FLOPS_PER_SAMPLE = FLOPS_VOICES + FLOPS_TREMOLO + FLOPS_PREAMP + FLOPS_POST
KERNEL_OF = {"ops": "k_apply_ops", "voices": "k_voice_steady", "tremolo": "k_tremolo", "preamp": "k_preamp", "post": "k_post"}
PEAK_FP64_VALU_TFLOPS = 78.6       # 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz (MI355X FP64 vector)


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
    """Sample-accurate event script: renders one 512-sample step, splitting at epoch boundaries."""

    def __init__(self, pool, n_inst):
        self.pool = pool
        self.pos = 0
        self.ev_strike = build_events(n_inst, "strike")
        self.ev_restrike = build_events(n_inst, "restrike")
        self.kernel_ms = np.zeros(5)
        self.kernel_launches = 0
        self.t_midi = 0.0
        self.t_render = 0.0

    def step(self, profile=False):
        done = 0
        while done < BUF:
            t0 = time.perf_counter()
            if self.pos % EPOCH == 0:
                self.pool.midi(self.ev_strike if self.pos == 0 else self.ev_restrike)
            t1 = time.perf_counter()
            nxt = min(BUF - done, EPOCH - (self.pos % EPOCH))
            self.pool.render(nxt, to_host=False)
            self.t_midi += t1 - t0
            self.t_render += time.perf_counter() - t1
            if profile:
                ms = self.pool.last_kernel_ms()
                self.kernel_ms += np.array([ms["ops"], ms["voices"], ms["tremolo"], ms["preamp"], ms["post"]]) * (nxt / BUF)
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
        "value": total / wall, "unit": "samples/s", "cores": cores, "kind": "port",
        "sample": f"{cores} threads (= usable CPUs: affinity capped by the cgroup quota) x 1 cfg-2 instance (64 voices, full chain) x {seconds_audio:.1f} s audio, buffers of {BUF}; "
                  f"oracle = C++ f64 restatement (reference Rust is not buildable in this image)",
        "single_thread_value": single,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--instances", type=int, default=int(os.environ.get("OW_BENCH_INSTANCES", "65536")), help="engine instances per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--preamp", choices=["legacy", "melange"], default="legacy",
                    help="legacy = the 8-node DK solver of the reference's default build (the metric's config); melange = the generated "
                         "12-node solver of its `--features melange-preamp` build")
    ap.add_argument("--host-rate", type=float, default=48000.0,
                    help="host sample rate: 48000 = BASELINE configs[1] (the metric's config, default); 96000 = configs[2] (no oversampling)")
    args = ap.parse_args()
    global SR, EPOCH, FLOPS_TREMOLO, FLOPS_PREAMP, FLOPS_POST, FLOPS_PER_SAMPLE
    if args.host_rate != SR:
        SR = float(args.host_rate)
        EPOCH = int(SR)                              # 1.0 s re-strike epochs at any rate
        osr = 2 if SR < 88200.0 else 1               # engine.rs:195
        FLOPS_TREMOLO = osr * (1000 + 25)
        FLOPS_PREAMP = osr * 1400 + (24 if osr == 2 else 0)
        FLOPS_POST = osr * 90 + (24 if osr == 2 else 0) + 45
        FLOPS_PER_SAMPLE = FLOPS_VOICES + FLOPS_TREMOLO + FLOPS_PREAMP + FLOPS_POST

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)

    import openwurli_amd as ow
    n_inst = args.instances
    preamp_kind = 1 if args.preamp == "melange" else 0
    if preamp_kind:                   # +~12 kflop per output sample (two states x 2 OS samples x 12-node solve)
        FLOPS_PREAMP += 12000
        FLOPS_PER_SAMPLE += 12000
    pool = ow.EnginePool(SR, n_inst, device=local_rank, preamp_kind=preamp_kind)
    pool.set_sample_rate(SR)          # the plugin's initialize(): chain build + 0.6 s warm-up (not timed)
    pool.ensure_buffer_capacity(BUF)
    # volume 0.5 / tremolo depth 0.5 / speaker character 0.0 / MLP on are the engine defaults (engine.rs:221-224)
    for k in range(min(n_inst, 4096)):
        e = pool[k]
        e.set_volume(0.5); e.set_tremolo_depth(0.5); e.set_speaker_character(0.0); e.set_mlp_enabled(True)
    script = Script(pool, n_inst)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        script.step()
    pool.set_profiling(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        script.step(profile=True)
    barrier()
    elapsed = time.perf_counter() - t0
    pool.set_profiling(False)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # single-instance latency figure (what one plugin instance would see), rank 0 only, not part of `value`
    single = None
    cpu = None
    if rank == 0:
        one = ow.EnginePool(SR, 1, device=local_rank, preamp_kind=preamp_kind)
        one.set_sample_rate(SR)
        s1 = Script(one, 1)
        for _ in range(3):
            s1.step()
        t1 = time.perf_counter()
        for _ in range(20):
            s1.step()
        single = 20 * BUF / (time.perf_counter() - t1)
        one.close()
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(preamp_kind=preamp_kind)

    if rank == 0:
        total_samples = args.steps * BUF * n_inst * world
        value = total_samples / elapsed
        kms = script.kernel_ms / max(script.kernel_launches, 1)     # average ms per step, per kernel
        names = ["ops", "voices", "tremolo", "preamp", "post"]
        flops = {"ops": 0.0, "voices": FLOPS_VOICES, "tremolo": FLOPS_TREMOLO, "preamp": FLOPS_PREAMP, "post": FLOPS_POST}
        dom = names[int(np.argmax(kms))]
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
        line = {
            "metric": f"audio samples/s, 64-voice full chain (x real-time @{SR / 1000:.0f} kHz = value / {SR:.0f})",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": ("cfg2: 64-voice all-keys-sustained (1.0 s re-strike), 48 kHz host / 96 kHz chain, full chain "
                             "(tremolo+legacy DK preamp+behavioural power amp+speaker), MLP on, buffers of 512").replace("legacy DK", "melange 12-node DK" if preamp_kind else "legacy DK") if SR == 48000.0 else
                            (f"cfg3: 64-voice all-keys-sustained (1.0 s re-strike), {SR:.0f} Hz host"
                             f"{' (no oversampling)' if SR >= 88200.0 else ' / 2x chain'}, full chain, MLP on, buffers of 512"),
                "instances_per_gpu": n_inst, "buffer": BUF, "parallelism": f"{world} x independent pools (no data-path collective)",
            },
            "x_realtime_aggregate": value / SR,
            "host_midi_s": script.t_midi, "render_calls_s": script.t_render, "elapsed_s": elapsed,
            "single_instance_samples_per_s": single,
            "roofline": {
                "bound": "valu_f64", "kernel": KERNEL_OF[dom] + ("_mel" if preamp_kind and dom == "preamp" else ""), "achieved": achieved,
                "peak": PEAK_FP64_VALU_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP64_VALU_TFLOPS, "traffic": traffic,
                "kernel_ms_per_step": {n: float(k) for n, k in zip(names, kms)},
                "whole_chain_frac": FLOPS_PER_SAMPLE * value / world / 1e12 / PEAK_FP64_VALU_TFLOPS,
                # k_tremolo (block ahead, own stream) runs INSIDE the voice kernel's interval and shares its SIMDs, so the interval's
                # arithmetic is voices + tremolo; `frac` above charges the whole interval to the voice kernel alone
                "voices_plus_tremolo_frac": ((FLOPS_VOICES + FLOPS_TREMOLO) * BUF * n_inst / (float(kms[1]) * 1e-3) / 1e12
                                             / PEAK_FP64_VALU_TFLOPS if kms[1] > 0 else None),
                # the contract's own vocabulary, for reference: PMC HBM bytes of the dominant kernel / its duration against 8 TB/s
                "hbm": ({"achieved": traffic / (dom_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                         "frac": traffic / (dom_ms * 1e-3) / 1e9 / 8000.0} if traffic and dom_ms > 0 else None),
                "note": "path is FP64-VALU/latency bound (not HBM, not MFMA); achieved = algorithmic f64 flops of the dominant "
                        "kernel per launch / its HIP-event duration",
            },
            "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    pool.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
