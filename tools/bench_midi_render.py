"""Throughput of ow_render_midi (preamp-bench render-midi semantics) on many event lists at once, next to the CPU oracle on a
sample of the same jobs (the oracle is only the timed baseline here).  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def phrase(seed, n_notes, span_s):
    rng = np.random.default_rng(seed)
    t = np.sort(rng.uniform(0.0, span_s, n_notes))
    keys = rng.integers(33, 97, n_notes)
    vel = rng.integers(20, 128, n_notes)
    dur = rng.uniform(0.05, 1.5, n_notes)
    items = []
    for k in range(n_notes):
        items.append((float(t[k]), 0, int(keys[k]), int(vel[k])))
        items.append((float(t[k] + dur[k]), 1, int(keys[k]), 0))
    for k in range(int(span_s / 2.0)):
        items.append((2.0 * k + 0.5, 2, 0, 1)); items.append((2.0 * k + 1.9, 2, 0, 0))
    return items


def main():
    from openwurli_amd import midi_render as mr
    n_jobs = int(os.environ.get("OW_MIDI_JOBS", "1024"))
    span = float(os.environ.get("OW_MIDI_SPAN", "20"))
    notes = int(os.environ.get("OW_MIDI_NOTES", "200"))
    jobs = [phrase(s, notes, span) for s in range(n_jobs)]
    evs = [mr.events(j) for j in jobs]
    mr.render_midi(evs[:2])                                   # warm-up (module load)
    t0 = time.perf_counter()
    out, stats = mr.render_midi(evs, return_stats=True)
    dt = time.perf_counter() - t0
    total = sum(o.size for o in out)
    from test_midi_render_host import _oracle_render
    import oracle_binding as ob
    k = 4
    t1 = time.perf_counter()
    cs = [_oracle_render(ob, jobs[i])[0] for i in range(k)]
    dc = time.perf_counter() - t1
    csamp = sum(c.size for c in cs)
    worst = max(ob.parity_report(out[i], cs[i], abs_floor=ob.ABS_FLOOR_BATCH)["worst_ratio"] for i in range(k))
    print(json.dumps({"workload": f"{n_jobs} MIDI jobs x {span:.0f} s, {notes} notes each, pedal every 2 s, tail 2 s", "gpu_s": dt,
                      "samples": total, "gpu_samples_per_s": total / dt, "x_realtime": total / dt / 44100.0,
                      "mean_peak_polyphony": float(np.mean([s[1] for s in stats])),
                      "cpu_oracle_1thread_samples_per_s": csamp / dc, "parity_worst_ratio_on_sample": worst}))


if __name__ == "__main__":
    main()
