#!/usr/bin/env python3
"""Measurement of the feature stage at config-4 size: 512 notes x 5 s at 44.1 kHz, 11 segments per note
(3 windows x 8 harmonics, 6 decay points, 2 RMS spans).  Prints one JSON line: GPU wall time of
openwurli_amd.features.extract_model_features (host audio, so the 0.9 GB upload is included), the same stage with the numpy
oracle on a bounded sample of notes (single thread), and the arithmetic of k_feat_peaks for the roofline."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    import features_oracle as fo
    from openwurli_amd import features
    sr, dur = 44100.0, 5.0
    n = int(sr * dur)
    pairs = [(m, v) for m in range(33, 97) for v in (20, 35, 50, 65, 80, 95, 110, 127)]
    rng = np.random.default_rng(1)
    t = np.arange(n) / sr
    audio = np.empty((len(pairs), n))
    for j, (m, v) in enumerate(pairs):
        f0 = fo.midi_to_freq(m) * 2.0 ** (rng.uniform(-5, 5) / 1200.0)
        x = np.zeros(n)
        for h in range(1, 9):
            if f0 * h < sr / 2:
                x += (v / 127.0) / h ** 1.5 * np.exp(-t * (1.0 + 0.3 * h)) * np.sin(2 * np.pi * f0 * h * t + 0.3 * h)
        audio[j] = 0.2 * x
    features.extract_model_features(audio[:8], sr, pairs[:8])           # warm-up (library load, first launches)
    t0 = time.perf_counter()
    got = features.extract_model_features(audio, sr, pairs)
    gpu_s = time.perf_counter() - t0
    sample = list(range(0, len(pairs), 16))                             # 32 notes
    t0 = time.perf_counter()
    for j in sample:
        fo.model_features(audio[j], sr, *pairs[j])
    cpu_s = (time.perf_counter() - t0) * len(pairs) / len(sample)
    # arithmetic of the dominant kernel: candidate bins x segment length, 2 FMA (accumulate) + 4 MUL + 2 ADD (twiddle rotation)
    cmacs = 0
    for (m, _) in pairs:
        for seg in features._note_segments(0, n, sr, features.midi_to_freq(m)):
            if seg is None or seg[3] == 0:
                continue
            bins, _ = fo.harmonic_bins(seg[2] - seg[1], sr, seg[4], seg[3])
            cmacs += sum((b[1] - b[0] + 1) * (seg[2] - seg[1]) for b in bins if b is not None)
    print(json.dumps({"stage": "extract_model_features", "notes": len(pairs), "seconds_of_audio_per_note": dur, "sample_rate": sr,
                      "gpu_wall_s": gpu_s, "gpu_notes_per_s": len(pairs) / gpu_s, "cpu_oracle_s_extrapolated": cpu_s, "cpu_threads": 1,
                      "cpu_sample": f"{len(sample)} of {len(pairs)} notes", "speedup": cpu_s / gpu_s,
                      "k_feat_peaks_complex_macs": cmacs, "k_feat_peaks_f64_flops": 10 * cmacs,
                      "check": got[pairs[0]]["windows"]["sustain"]["freqs_hz"][:2]}))


if __name__ == "__main__":
    main()
