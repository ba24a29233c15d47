"""GPU parity of `preamp-bench render-midi` (tools/preamp-bench/src/main.rs:1603-1923) through the C-ABI against the CPU oracle.

Bar: the batch-job bar (1e-5 relative; absolute floor ABS_FLOOR_BATCH, the same static-LDR chain) on every output sample of every
job, note-on count and peak polyphony identical."""
import numpy as np
import pytest

import midi_util as mu
from test_midi_render_host import _oracle_render

pytestmark = pytest.mark.gpu


def _phrase(seed, n_notes, span_s, pedal=True):
    """Random two-hand playing: chords, repeated keys, pedal changes, a few out-of-range keys, events off the chunk grid."""
    rng = np.random.default_rng(seed)
    items = []
    t = 0.0
    for _ in range(n_notes):
        t += float(rng.choice([0.0, 0.0, 0.013, 0.11, 0.25])) * span_s / max(n_notes * 0.1, 1.0)
        key = int(rng.choice([28, 36, 40, 45, 48, 52, 55, 57, 60, 62, 64, 67, 69, 72, 76, 79, 84, 88, 91, 96, 101]))
        vel = int(rng.integers(1, 128))
        dur = float(rng.uniform(0.02, 0.6))
        items.append((t, 0, key, vel))
        items.append((t + dur, 1, key, 0))
    if pedal:
        for k in range(int(span_s / 0.7)):
            items.append((0.35 + 0.7 * k, 2, 0, 1))
            items.append((0.35 + 0.7 * k + 0.45, 2, 0, 0))
    return items


def _check(oracle, g, gs, items, **kw):
    c, cs = _oracle_render(oracle, items, **kw)
    assert g.size == c.size and gs == cs, (g.size, c.size, gs, cs)
    if c.size:
        rep = oracle.parity_report(g, c, abs_floor=oracle.ABS_FLOOR_BATCH)
        assert rep["n_bad"] == 0, rep
    return c


def test_phrases_with_pedal_chords_and_repeats(hiplib, oracle):
    from openwurli_amd import midi_render as mr
    jobs = [_phrase(1, 40, 3.0), _phrase(2, 12, 1.0, pedal=False), [], _phrase(3, 70, 2.0)]
    got, stats = mr.render_midi([mr.events(j) for j in jobs], return_stats=True)
    assert got[2].size == 0 and stats[2] == (0, 0)
    peaks = []
    for j, g, s in zip(jobs, got, stats):
        if j:
            peaks.append(np.max(np.abs(_check(oracle, g, s, j))))
    assert min(peaks) > 1e-3                                                   # the renders are not silence


def test_more_than_64_voices_replaces_the_oldest(hiplib, oracle):
    from openwurli_amd import midi_render as mr
    items = [(0.002 * k, 0, 33 + (k * 7) % 64, 40 + (k * 13) % 80) for k in range(150)]      # nothing is released: 150 note-ons
    items += [(0.35, 1, 33 + (k * 7) % 64, 0) for k in range(0, 150, 3)]
    g, s = mr.render_midi([mr.events(items)], tail=0.4, return_stats=True)
    assert s[0] == (150, 64)
    _check(oracle, g[0], s[0], items, tail=0.4)


def test_options_no_poweramp_speaker_volume_tail(hiplib, oracle):
    from openwurli_amd import midi_render as mr
    items = _phrase(5, 16, 1.2)
    for kw in (dict(volume=1.0, speaker=0.0, no_pa=True, tail=0.25), dict(volume=0.3, speaker=0.5, no_pa=False, tail=0.0)):
        g, s = mr.render_midi([mr.events(items)], volume=kw["volume"], speaker=kw["speaker"], no_poweramp=kw["no_pa"], tail=kw["tail"], return_stats=True)
        _check(oracle, g[0], s[0], items, **kw)


def test_file_to_wav_end_to_end(hiplib, oracle, tmp_path):
    """SMF bytes -> parse -> render -> 24-bit WAV, as the command does; the samples read back equal the oracle's render
    through the same quantiser within one LSB (f64 differences of 1e-9 can flip a rounding)."""
    from openwurli_amd import midi_render as mr
    tracks = [[(0, "tempo", 400000, 0)],
              [(0, "on", 60, 100), (0, "on", 64, 90), (0, "on", 67, 80), (240, "cc", 64, 127), (240, "off", 60, 0), (0, "off", 64, 0), (0, "off", 67, 0),
               (480, "cc", 64, 0), (0, "on", 72, 110), (480, "on", 72, 0)]]
    p = tmp_path / "phrase.mid"
    p.write_bytes(mu.smf_bytes(tracks, 480, running_status=True))
    w = tmp_path / "phrase.wav"
    res = mr.render_midi_files([str(p)], [str(w)], tail=0.5)
    items = mu.expected_events(tracks, 480)
    c = _check(oracle, res[0], (4, 4), items, tail=0.5)
    raw = w.read_bytes()
    data = raw[raw.index(b"data") + 8:]
    q = np.frombuffer(data[: 3 * c.size], dtype=np.uint8).reshape(-1, 3).astype(np.int32)
    pcm = q[:, 0] | (q[:, 1] << 8) | (q[:, 2] << 16)
    pcm = np.where(pcm >= 1 << 23, pcm - (1 << 24), pcm)
    want = np.clip(np.sign(c) * np.floor(np.abs(c) * 8388607.0 + 0.5), -8388607, 8388607).astype(np.int64)   # round half away from zero
    assert pcm.size == c.size and np.max(np.abs(pcm - want)) <= 1
    # --track 0 holds no notes: "No note events found", nothing is written
    w2 = tmp_path / "none.wav"
    assert mr.render_midi_files([str(p)], [str(w2)], track=0)[0].size == 0 and not w2.exists()
