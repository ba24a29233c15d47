"""GPU parity of the feature stage: ow_extract_harmonics through the C-ABI against
(a) the golden outputs of the reference's own Python (tests/golden/harmonics_golden.npz) and
(b) the numpy oracle on real batch renders, plus the end-to-end `extract_model_features` dictionary.

Bar: peak frequency (= peak bin) identical; amplitudes within 1e-10 relative, with a floor of 1e-13 of the strongest harmonic of
the segment (a harmonic 1e-8 below the fundamental is mostly spectral leakage of the fundamental: both the GPU's direct f64
summation and numpy's pocketfft carry an error proportional to the whole signal, not to that bin); RMS within 1e-12 relative."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import features_oracle as fo  # noqa: E402

AMP_REL = 1e-10


def _close_amps(g, c):
    skip = c == 1e-20
    assert np.array_equal(g[skip], c[skip])
    tol = np.maximum(AMP_REL * np.abs(c[~skip]), 1e-13 * np.max(np.abs(c)))
    assert np.all(np.abs(g[~skip] - c[~skip]) <= tol), (g, c)


def test_against_reference_golden(hiplib):
    from openwurli_amd import features
    g = np.load(os.path.join(ROOT, "tests", "golden", "harmonics_golden.npz"))
    for i in range(int(g["n_cases"][0])):
        sr, n, midi, f0, nh = g[f"meta{i}"]
        amps, freqs = features.extract_harmonics_fft(g[f"x{i}"], sr, f0, int(nh))
        assert np.array_equal(freqs, g[f"freqs{i}"]), (i, freqs, g[f"freqs{i}"])
        _close_amps(amps, g[f"amps{i}"])
        assert np.allclose(features.amps_to_dB(amps), g[f"db{i}"], rtol=0, atol=1e-8)


def test_segments_batch_vs_oracle_on_renders(hiplib):
    """Many segments of different rows / offsets / lengths / harmonic counts in one call, RMS-only segments, empty call."""
    import openwurli_amd as ow
    from openwurli_amd import features
    sr = 44100.0
    jobs = [{"note": n, "velocity": v} for n, v in ((33, 127), (60, 80), (84, 50), (96, 110))]
    audio = ow.batch_render(jobs, sample_rate=sr, duration_s=1.0)
    segs = []
    for row, j in enumerate(jobs):
        f0 = fo.midi_to_freq(j["note"])
        segs += [(row, 0, 2205, 8, f0), (row, 2205, 8820, 8, f0), (row, 8820, 35280, 8, f0), (row, 13230, 17640, 1, f0),
                 (row, 1000, 1129, 3, f0), (row, 0, 441, 0, f0), (row, 4410, 8820, 0, f0)]
    amps, freqs, rms = features.extract_segments(audio, sr, segs)
    for k, (row, a, b, nh, f0) in enumerate(segs):
        x = audio[row, a:b]
        assert abs(rms[k] - fo.rms(x)) <= 1e-12 * fo.rms(x)
        if nh:
            ca, cf = fo.extract_harmonics(x, sr, f0, nh)
            assert np.array_equal(freqs[k, :nh], cf), (k, freqs[k], cf)
            _close_amps(amps[k, :nh], ca)
        assert np.all(amps[k, nh:] == 0.0) and np.all(freqs[k, nh:] == 0.0)
    e = features.extract_segments(audio, sr, [])
    assert e[0].shape == (0, 8)
    with pytest.raises(ow.OwError):
        features.extract_segments(audio, sr, [(9, 0, 100, 8, 440.0)])          # row out of range
    with pytest.raises(ow.OwError):
        features.extract_segments(audio, sr, [(0, 100, 100, 8, 440.0)])        # empty segment


def test_extract_model_features_matches_oracle(hiplib):
    """render_model_notes.py stage 4 end to end: batch render on the GPU -> features on the GPU, vs the numpy oracle applied
    to the same renders.  Reported numbers are rounded (8 / 2 / 2 / 1 decimals), so compare with half-a-step slack."""
    import openwurli_amd as ow
    from openwurli_amd import features
    sr = 44100.0
    pairs = [(36, 35), (48, 127), (60, 80), (72, 65), (91, 110), (96, 20)]
    audio = ow.batch_render([{"note": n, "velocity": v} for n, v in pairs], sample_rate=sr, duration_s=2.0)
    got = features.extract_model_features(audio, sr, pairs)          # default: through the 24-bit WAV quantiser, like the script
    raw = features.extract_model_features(audio, sr, pairs, wav24=None)
    assert list(got) == pairs
    assert any(got[p]["windows"]["sustain"]["amps_linear"] != raw[p]["windows"]["sustain"]["amps_linear"] for p in pairs)
    for j, (midi, vel) in enumerate(pairs):
        r0 = fo.model_features(audio[j], sr, midi, vel)
        assert np.allclose(raw[(midi, vel)]["windows"]["sustain"]["amps_linear"], r0["windows"]["sustain"]["amps_linear"], rtol=0, atol=1.01e-8)
        ref = fo.model_features(fo.quantize_round(audio[j]) / 8388608.0, sr, midi, vel)   # what soundfile reads back from the WAV
        f = got[(midi, vel)]
        assert f["midi_note"] == midi and f["velocity_midi"] == vel and f["f0"] == ref["f0"] and f["duration_s"] == ref["duration_s"]
        for name in ("attack", "early_sustain", "sustain"):
            a, b = f["windows"][name], ref["windows"][name]
            assert (a is None) == (b is None)
            if a is not None:
                assert a["freqs_hz"] == b["freqs_hz"]
                assert np.allclose(a["amps_linear"], b["amps_linear"], rtol=0, atol=1.01e-8)
                assert np.allclose(a["amps_dB_rel_H1"], b["amps_dB_rel_H1"], rtol=0, atol=0.0101)
        assert [x is None for x in f["decay"]["h1_amps"]] == [x is None for x in ref["decay"]["h1_amps"]]
        assert np.allclose([x for x in f["decay"]["h1_amps"] if x is not None], [x for x in ref["decay"]["h1_amps"] if x is not None],
                           rtol=0, atol=1.01e-8)
        assert (f["decay"]["decay_rate_dB_s"] is None) == (ref["decay"]["decay_rate_dB_s"] is None)
        if ref["decay"]["decay_rate_dB_s"] is not None:
            assert abs(f["decay"]["decay_rate_dB_s"] - ref["decay"]["decay_rate_dB_s"]) <= 0.0101
        assert abs(f["overshoot_dB"] - ref["overshoot_dB"]) <= 0.0101
        for name in ("attack", "sustain"):
            assert abs(f[f"centroid_{name}"] - ref[f"centroid_{name}"]) <= 0.101


def test_render_and_extract_stays_on_the_device(hiplib):
    """ow_batch_render(out_is_device) -> ow_extract_harmonics(audio_is_device): same dictionary as the two host-visible steps."""
    import openwurli_amd as ow
    from openwurli_amd import features
    sr = 44100.0
    pairs = [(40, 50), (60, 127), (79, 95), (93, 35)]
    one = features.render_and_extract(pairs, sample_rate=sr, duration_s=1.0)
    audio = ow.batch_render([{"note": n, "velocity": v} for n, v in pairs], sample_rate=sr, duration_s=1.0)
    two = features.extract_model_features(audio, sr, pairs)
    assert one == two and list(one) == pairs


def test_features_full_size_properties(hiplib):
    """Config-4 sized call (512 notes x 11 segments) on synthetic audio: linearity in amplitude (amps scale, peak bins do not
    move) and agreement of a sample of segments with the oracle."""
    from openwurli_amd import features
    sr = 44100.0
    n = int(2.0 * sr)
    rng = np.random.default_rng(3)
    t = np.arange(n) / sr
    pairs = [(m, v) for m in range(33, 97) for v in (20, 35, 50, 65, 80, 95, 110, 127)]
    audio = np.empty((len(pairs), n))
    for j, (m, v) in enumerate(pairs):
        f0 = fo.midi_to_freq(m) * 2.0 ** (rng.uniform(-5, 5) / 1200.0)
        x = np.zeros(n)
        for h in range(1, 9):
            if f0 * h < sr / 2:
                x += (v / 127.0) / h ** 1.5 * np.exp(-t * (2.0 + 0.5 * h)) * np.sin(2 * np.pi * f0 * h * t + 0.3 * h)
        audio[j] = 0.2 * x
    a1 = features.extract_model_features(audio, sr, pairs, wav24=None)
    a2 = features.extract_model_features(0.5 * audio, sr, pairs, wav24=None)
    for key in (pairs[0], pairs[100], pairs[300], pairs[-1]):
        w1, w2 = a1[key]["windows"]["sustain"], a2[key]["windows"]["sustain"]
        assert w1["freqs_hz"] == w2["freqs_hz"]
        assert np.allclose(np.array(w1["amps_linear"]), 2.0 * np.array(w2["amps_linear"]), rtol=0, atol=3e-8)
        ref = fo.model_features(audio[pairs.index(key)], sr, *key)
        assert ref["windows"]["sustain"]["freqs_hz"] == w1["freqs_hz"]
        assert np.allclose(ref["windows"]["sustain"]["amps_linear"], w1["amps_linear"], rtol=0, atol=1.01e-8)
