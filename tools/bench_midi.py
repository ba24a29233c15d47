#!/usr/bin/env python3
"""Informational companion to bench.py: the same pool under a PLAYED workload instead of the all-keys-sustained script.

Every engine plays its own random part: note-ons arrive as a Poisson stream (default 8 per second per engine, uniform keys
33..96, velocities 0.3..1.0), each note is held 0.1-1.5 s, the sustain pedal toggles now and then.  Events are applied at
512-sample block boundaries through ow_pool_midi.  Polyphony settles around rate x (hold + decay), i.e. a few dozen sounding
voices per engine; what differs from bench.py is that a sizeable fraction of the engines always has a voice inside its onset
ramp / attack noise / damper ramp and therefore takes the general voice kernel instead of the steady-state one.
Prints one JSON line (not the driver's contract line)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--instances", type=int, default=65536)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=60)
    ap.add_argument("--notes-per-second", type=float, default=8.0)
    args = ap.parse_args()
    import openwurli_amd as ow
    from openwurli_amd import binding
    sr, buf, n = 48000.0, 512, args.instances
    pool = ow.EnginePool(sr, n)
    pool.set_sample_rate(sr)
    rng = np.random.default_rng(7)
    dt = buf / sr
    offs = {}          # block index -> list of (engine, note) note-offs falling due
    dtype = np.dtype(binding.MIDI_DTYPE)

    def events_for(block):
        k = rng.poisson(args.notes_per_second * dt * n)
        eng = np.sort(rng.integers(0, n, k)).astype(np.uint32)
        note = rng.integers(33, 97, k).astype(np.uint8)
        vel = rng.uniform(0.3, 1.0, k).astype(np.float32)
        hold = rng.uniform(0.1, 1.5, k)
        for e, nn, h in zip(eng, note, hold):
            offs.setdefault(block + max(1, int(h / dt)), []).append((int(e), int(nn)))
        due = offs.pop(block, [])
        ped = rng.integers(0, n, max(1, n // 2000)).astype(np.uint32)
        ev = np.zeros(k + len(due) + ped.size, dtype=dtype)
        ev["engine"][:k] = eng; ev["type"][:k] = 0; ev["note"][:k] = note; ev["value"][:k] = vel
        if due:
            d = np.array(due, dtype=np.int64)
            ev["engine"][k:k + len(due)] = d[:, 0]; ev["type"][k:k + len(due)] = 1; ev["note"][k:k + len(due)] = d[:, 1]
        ev["engine"][k + len(due):] = ped; ev["type"][k + len(due):] = 2; ev["value"][k + len(due):] = rng.integers(0, 2, ped.size)
        return ev[np.argsort(ev["engine"], kind="stable")], k

    t_midi = t_render = 0.0
    notes = 0
    ms = np.zeros(5)
    for b in range(args.warmup + args.steps):
        timed = b >= args.warmup
        if b == args.warmup:
            pool.set_profiling(True)
        ev, k = events_for(b)                      # building the script is not part of the measurement
        t0 = time.perf_counter()
        pool.midi(ev)
        t1 = time.perf_counter()
        pool.render(buf, to_host=False)
        t2 = time.perf_counter()
        if timed:
            t_midi += t1 - t0; t_render += t2 - t1; notes += k
            m = pool.last_kernel_ms()
            ms += np.array([m["ops"], m["voices"], m["tremolo"], m["preamp"], m["post"]])
    active = np.mean([pool[i].active_voice_count() for i in range(0, n, max(1, n // 256))])
    elapsed = t_midi + t_render
    print(json.dumps({"workload": f"played: {args.notes_per_second} note-ons/s per engine, holds 0.1-1.5 s, pedal toggles; blocks of {buf}",
                      "instances": n, "steps": args.steps, "samples_per_s": args.steps * buf * n / elapsed,
                      "x_realtime": args.steps * buf * n / elapsed / sr, "ms_per_step": 1e3 * elapsed / args.steps,
                      "host_midi_ms_per_step": 1e3 * t_midi / args.steps, "note_ons_per_step": notes / args.steps,
                      "mean_sounding_voices_per_engine": float(active),
                      "kernel_ms_per_step": dict(zip(["ops", "voices", "tremolo", "preamp", "post"], (ms / args.steps).round(3).tolist()))}))
    pool.close()


if __name__ == "__main__":
    main()
