#!/usr/bin/env python3
"""Generate tests/golden/harmonics_golden.npz by running the REFERENCE's own ml/goertzel_utils.py here (build container only:
/root/reference does not exist on the GPU box, so the outputs travel as a fixture).  `soundfile` is not installed; the module
only needs it for load_audio, so an empty stub is registered before the import.  Inputs are small seeded synthetic notes
(decaying harmonic series + noise) stored in the fixture together with the reference's outputs."""
import os
import sys
import types

import numpy as np

sys.modules.setdefault("soundfile", types.ModuleType("soundfile"))
sys.path.insert(0, "/root/reference/ml")
import goertzel_utils as ref  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def synth(seed, sr, n, f0, inharm):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / sr
    x = np.zeros(n)
    for h in range(1, 11):
        fh = f0 * h * (1.0 + inharm * h * h)
        if fh < sr / 2:
            x += rng.uniform(0.05, 1.0) / h * np.exp(-t * rng.uniform(1.0, 12.0)) * np.sin(2 * np.pi * fh * t + rng.uniform(0, 6.28))
    return 0.3 * x + 1e-4 * rng.standard_normal(n)


def main():
    cases = []
    # (seed, sr, n, midi, detune cents, inharmonicity, n_harmonics): short/long windows, bass/treble (upper harmonics past the
    # Nyquist-100 Hz cut), odd lengths, a 1-harmonic decay-style call
    spec = [(1, 44100.0, 2205, 60, 3.0, 1e-4, 8), (2, 44100.0, 6615, 33, -4.0, 3e-4, 8), (3, 44100.0, 26460, 72, 0.0, 0.0, 8),
            (4, 48000.0, 4800, 96, 7.0, 2e-4, 8), (5, 48000.0, 2400, 91, -2.0, 0.0, 8), (6, 44100.0, 4410, 48, 1.0, 5e-5, 1),
            (7, 44100.0, 129, 84, 0.0, 0.0, 8), (8, 48000.0, 7201, 40, 9.0, 1e-4, 8)]
    out = {}
    for i, (seed, sr, n, midi, cents, inh, nh) in enumerate(spec):
        f0 = float(ref.midi_to_freq(midi))
        x = synth(seed, sr, n, f0 * 2.0 ** (cents / 1200.0), inh)
        amps, freqs = ref.extract_harmonics_fft(x, sr, f0, nh)
        out[f"x{i}"] = x; out[f"amps{i}"] = amps; out[f"freqs{i}"] = freqs; out[f"db{i}"] = ref.amps_to_dB(amps)
        out[f"meta{i}"] = np.array([sr, n, midi, f0, nh], dtype=np.float64)
    out["n_cases"] = np.array([len(spec)])
    out["midi_freqs"] = np.array([ref.midi_to_freq(m) for m in range(21, 109)])
    np.savez_compressed(os.path.join(HERE, "harmonics_golden.npz"), **out)
    print("wrote", len(spec), "cases")


if __name__ == "__main__":
    main()
