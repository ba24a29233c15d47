"""CPU tests of the feature-stage oracle (oracle/features_oracle.py) and of the host-only WAV entry points of the C-ABI.

The oracle is pinned: tests/golden/harmonics_golden.npz holds the outputs of the reference's own
ml/goertzel_utils.extract_harmonics_fft / amps_to_dB / midi_to_freq (tests/golden/make_harmonics_golden.py ran them
in the build container).  ow_wav24_quantize / ow_wav24_write are host functions (no device), so they run here."""
import os
import struct
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import features_oracle as fo  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden", "harmonics_golden.npz")


def test_oracle_matches_reference_outputs_bit_for_bit():
    g = np.load(GOLD)
    n = int(g["n_cases"][0])
    assert n >= 8
    for i in range(n):
        sr, ns, midi, f0, nh = g[f"meta{i}"]
        amps, freqs = fo.extract_harmonics(g[f"x{i}"], sr, f0, int(nh))
        assert np.array_equal(amps, g[f"amps{i}"]), i
        assert np.array_equal(freqs, g[f"freqs{i}"]), i
        assert np.array_equal(fo.amps_to_db(amps), g[f"db{i}"]), i
        assert f0 == fo.midi_to_freq(int(midi))
    assert np.array_equal(np.array([fo.midi_to_freq(m) for m in range(21, 109)]), g["midi_freqs"])
    # treble cases really exercise the Nyquist - 100 Hz cut (amplitude 1e-20 at h * f0)
    assert any(np.any(g[f"amps{i}"] == 1e-20) for i in range(n))


def test_candidate_bins_are_the_numpy_mask():
    for n, sr, f0 in [(2205, 44100.0, 261.6255653005986), (129, 44100.0, 1046.5), (7201, 48000.0, 82.4), (4800, 48000.0, 2093.0)]:
        axis = np.fft.rfftfreq(4 * n, d=1.0 / sr)
        bins, val = fo.harmonic_bins(n, sr, f0)
        for h, b in enumerate(bins):
            fh = f0 * (h + 1)
            idx = np.where((axis >= fh * 0.99) & (axis <= fh * 1.01))[0]
            if fh >= sr / 2 - 100 or idx.size == 0:
                assert b is None
            else:
                assert b == (idx[0], idx[-1]) and axis[idx[0]] == idx[0] * val


def test_quantisers_known_answers(hiplib_host):
    from openwurli_amd import features
    mx = 2 ** 23 - 1
    x = np.array([0.0, 1.0, -1.0, 2.0, -3.0, 0.5 / mx, 1.5 / mx, 2.5 / mx, -0.5 / mx, -2.5 / mx, 0.49999 / mx, np.nan, 1e-30, 0.123456789])
    r = features.quantize_24bit(x, 1.0, "round")
    # Rust f64::round: half away from zero; `as i32`: NaN -> 0; then clamp to +-(2^23 - 1)
    assert list(r[:12]) == [0, mx, -mx, mx, -mx, 1, 2, 3, -1, -3, 0, 0]
    assert np.array_equal(r, fo.quantize_round(x))
    t = features.quantize_24bit(x, 1.0, "truncate")
    assert list(t[:12]) == [0, mx, -mx, mx, -mx, 0, 1, 2, 0, -2, 0, 0]
    assert np.array_equal(t, fo.quantize_truncate(x))
    rng = np.random.default_rng(5)
    y = rng.uniform(-1.2, 1.2, 20000)
    assert np.array_equal(features.quantize_24bit(y, 0.7, "round"), fo.quantize_round(y, 0.7))
    assert np.array_equal(features.quantize_24bit(y, 1.0, "truncate"), fo.quantize_truncate(y))


def _parse_wav(path):
    b = open(path, "rb").read()
    assert b[:4] == b"RIFF" and b[8:12] == b"WAVE" and struct.unpack("<I", b[4:8])[0] == len(b) - 8
    pos, fmt, data = 12, None, None
    while pos < len(b):
        tag, size = b[pos:pos + 4], struct.unpack("<I", b[pos + 4:pos + 8])[0]
        body = b[pos + 8:pos + 8 + size]
        if tag == b"fmt ":
            fmt = body
        elif tag == b"data":
            data = body
        pos += 8 + size + (size & 1)
    return fmt, data


def test_wav_writer_round_trip(hiplib_host, tmp_path):
    """reed-renderer integration.rs:23-60 properties (frames, mono, rate, 24 bit) + sample payload == quantiser output."""
    from openwurli_amd import features
    sr = 44100
    t = np.arange(22051) / sr
    x = 0.8 * np.sin(2 * np.pi * 261.63 * t) * np.exp(-3 * t)
    for mode, q in (("round", fo.quantize_round(x)), ("truncate", fo.quantize_truncate(x))):
        path = tmp_path / f"{mode}.wav"
        features.write_wav_24bit(path, x, sr, 1.0, mode)
        fmt, data = _parse_wav(path)
        tag, ch, rate, bps, align, bits, cb, valid, mask = struct.unpack("<HHIIHHHHI", fmt[:24])
        assert (tag, ch, rate, bps, align, bits, cb, valid, mask) == (0xFFFE, 1, sr, sr * 3, 3, 24, 22, 24, 4)
        assert fmt[24:40] == bytes.fromhex("0100000000001000800000aa00389b71")       # KSDATAFORMAT_SUBTYPE_PCM
        assert len(data) == 3 * x.size
        raw = np.frombuffer(data, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        val = raw[:, 0] | (raw[:, 1] << 8) | (raw[:, 2] << 16)
        val = np.where(val >= 1 << 23, val - (1 << 24), val)
        assert np.array_equal(val, q)
    with pytest.raises(Exception):
        features.write_wav_24bit(tmp_path / "no_such_dir" / "x.wav", x, sr)


def test_model_features_of_an_oracle_render(oracle):
    """The whole feature dict on a real (CPU-rendered) note: shape and plausibility of what the GPU tests compare against."""
    sr = 44100.0
    x = oracle.batch_render_job(60, 100, 2.0, sr)
    f = fo.model_features(x, sr, 60, 100)
    assert set(f["windows"]) == {"attack", "early_sustain", "sustain"} and all(v is not None for v in f["windows"].values())
    assert abs(f["windows"]["sustain"]["freqs_hz"][0] - 261.63) < 2.7 and f["windows"]["sustain"]["amps_dB_rel_H1"][0] == 0.0
    assert f["decay"]["h1_amps"][-1] is not None and 1.0 < f["decay"]["decay_rate_dB_s"] < 20.0
    assert f["overshoot_dB"] is not None and f["centroid_attack"] > 200.0
    short = fo.model_features(x[:2100], sr, 60, 100)          # 47.6 ms: only the (clipped) attack window exists
    assert short["windows"]["attack"] is not None and short["windows"]["early_sustain"] is None and short["windows"]["sustain"] is None
    assert short["decay"]["decay_rate_dB_s"] is None and short["overshoot_dB"] is None
