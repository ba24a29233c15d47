#!/usr/bin/env python3
"""BASELINE configs[3]: the ml/render_model_notes.py batch (64 notes x 8 velocities x 5 s, `preamp-bench render` semantics) through
ow_batch_render on one GPU, plus the feature stage on the renders.  Prints one JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import openwurli_amd as ow
    from openwurli_amd import features
    sr, dur = 48000.0, 5.0
    pairs = [(m, v) for m in range(33, 97) for v in (20, 35, 50, 65, 80, 95, 110, 127)]
    jobs = [{"note": m, "velocity": v} for m, v in pairs]
    ow.batch_render(jobs[:8], sample_rate=sr, duration_s=0.1)                      # warm-up
    t0 = time.perf_counter()
    audio = ow.batch_render(jobs, sample_rate=sr, duration_s=dur)
    t1 = time.perf_counter()
    feats = features.extract_model_features(audio, sr, pairs)
    t2 = time.perf_counter()
    one = features.render_and_extract(pairs, sr, dur)
    t3 = time.perf_counter()
    n = audio.shape[0] * audio.shape[1]
    print(json.dumps({"workload": "cfg4: 512 jobs (64 notes x 8 velocities) x 5 s at 48 kHz, preamp-bench render semantics", "jobs": len(jobs),
                      "samples": n, "render_s": t1 - t0, "render_samples_per_s": n / (t1 - t0), "features_s": t2 - t1,
                      "render_and_extract_on_device_s": t3 - t2, "same_features": one == feats}))


if __name__ == "__main__":
    main()
