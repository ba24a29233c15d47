"""render-midi (tools/preamp-bench/src/main.rs:1603-1923), CPU side: the SMF reader of the product (host code of the C-ABI library)
and of the oracle against hand-built files with an independent timing model, the length/validation rules of ow_render_midi
(stats-only calls need no device), and the oracle's voice-manager semantics (pedal deferral, 65th note replaces the oldest)."""
import ctypes as C

import numpy as np
import pytest

import midi_util as mu

TRACKS = [
    [(0, "tempo", 500000, 0), (0, "text", b"conductor", 0), (960, "tempo", 250000, 0), (480, "on", 60, 100), (240, "off", 60, 64)],
    [(0, "pc", 4, 0), (0, "on", 48, 90), (0, "on", 55, 70), (120, "cc", 64, 127), (120, "on", 55, 0), (0, "bend", 0, 64),
     (240, "on", 20, 127), (10, "sysex", b"\x7e\x7f\x09\x01\xf7", 0), (230, "cc", 64, 10), (0, "cc", 7, 100), (480, "off", 48, 0),
     (0, "on", 110, 1), (960, "off", 110, 0), (0, "off", 20, 0)],
    [(100, "tempo", 1000000, 0), (480, "on", 72, 64), (480, "on", 72, 0)],
]


def _oracle_parse(oracle, data, track_filter):
    L = oracle.lib()
    L.owo_smf_parse.restype = C.c_longlong
    buf = np.frombuffer(data, dtype=np.uint8)
    cap = 256
    t = np.zeros(cap); ty = np.zeros(cap, np.uint8); no = np.zeros(cap, np.uint8); va = np.zeros(cap, np.uint8)
    n = L.owo_smf_parse(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size), int(track_filter), t.ctypes.data_as(C.c_void_p),
                        ty.ctypes.data_as(C.c_void_p), no.ctypes.data_as(C.c_void_p), va.ctypes.data_as(C.c_void_p), C.c_size_t(cap))
    return None if n < 0 else [(t[i], int(ty[i]), int(no[i]), int(va[i])) for i in range(n)]


def _product_parse(data, track):
    from openwurli_amd import midi_render
    ev = midi_render.parse_smf(data, track)
    return [(float(e["time_s"]), int(e["type"]), int(e["note"]), int(e["value"])) for e in ev]


@pytest.mark.parametrize("running", [False, True])
@pytest.mark.parametrize("track", [None, 0, 1, 2, 7])
def test_smf_reader_matches_the_timing_model(hiplib_host, oracle, running, track):
    data = mu.smf_bytes(TRACKS, 480, running_status=running)
    want = mu.expected_events(TRACKS, 480, track)
    got = _product_parse(data, track)
    assert got == want                      # bit-identical times: same (delta / tpb) * (tempo / 1e6) accumulation
    assert _oracle_parse(oracle, data, -1 if track is None else track) == want
    # track 1 starts again at 120 BPM although track 0 changed the tempo (per-track tempo state, main.rs:1653)
    if track == 1:
        assert want[0][0] == 0.0 and abs(want[2][0] - 0.125) < 1e-15


def test_smf_reader_rejects_what_the_command_rejects(hiplib_host, oracle):
    from openwurli_amd import midi_render, OwError
    good = mu.smf_bytes(TRACKS, 480)
    smpte = good[:12] + bytes([0xE7, 0x28]) + good[14:]                     # SMPTE timing: "Only metrical ..." (main.rs:1630-1636)
    for bad in (b"RIFFxxxxWAVE", smpte, good[:40], good[:14] + b"MTrk\x00\x00\x00\x04\x00\x40\x40\x40"):
        with pytest.raises(OwError):
            midi_render.parse_smf(bad)
        assert _oracle_parse(oracle, bad, -1) is None
    with pytest.raises(OwError, match="metrical"):
        midi_render.parse_smf(smpte)
    # unknown chunks are skipped, an empty file body yields no events
    alien = good[:14] + b"XFIH\x00\x00\x00\x02ab" + good[14:]
    assert _product_parse(alien, None) == mu.expected_events(TRACKS, 480)
    assert _product_parse(good[:14], None) == []


def test_render_length_rules_need_no_device(hiplib_host):
    from openwurli_amd import binding, midi_render, OwError
    L = hiplib_host
    jobs = [midi_render.events([(0.0, 0, 60, 100), (1.25, 1, 60, 0)]), midi_render.events([]),
            midi_render.events([(3.0, 0, 64, 80), (0.5, 2, 0, 1)])]          # unsorted on purpose
    offs = np.zeros(4, dtype=np.uint64); offs[1:] = np.cumsum([j.size for j in jobs])
    allev = np.concatenate(jobs)
    stats = (binding.OwMidiRenderStats * 3)()

    def call(cfg, out=None, stride=0):
        return L.ow_render_midi(allev.ctypes.data_as(C.c_void_p), offs.ctypes.data_as(C.c_void_p), 3, C.byref(cfg), out, stride, C.cast(stats, C.c_void_p))
    cfg = binding.OwMidiRenderCfg(0.6, 1.0, 2.0, 0, 0, 0, 0, 0, 0)
    assert call(cfg) == int((3.0 + 2.0) * 44100.0)                            # (last_event_time + tail) * BASE_SR as usize (main.rs:1719-1721)
    assert [s.n_samples for s in stats] == [int(3.25 * 44100.0), 0, int(5.0 * 44100.0)]
    cfg.tail_s = 0.0
    assert call(cfg) == int(3.0 * 44100.0)
    cfg.preamp_kind = 7
    assert call(cfg) < 0                                                      # unknown preamp kind
    cfg.preamp_kind = 0
    allev[0]["time_s"] = np.nan
    assert call(cfg) < 0                                                      # partial_cmp().unwrap() panics in the reference
    with pytest.raises(OwError):
        midi_render.render_midi([allev])


def _oracle_render(oracle, items, volume=0.6, speaker=1.0, no_pa=False, tail=2.0):
    L = oracle.lib()
    L.owo_render_midi.restype = C.c_size_t
    t = np.array([x[0] for x in items], dtype=np.float64); ty = np.array([x[1] for x in items], dtype=np.uint8)
    no = np.array([x[2] for x in items], dtype=np.uint8); va = np.array([x[3] for x in items], dtype=np.uint8)
    cap = int((max(x[0] for x in items) + tail) * 44100.0) + 8 if items else 8
    out = np.zeros(cap)
    st = (C.c_ulonglong * 2)()
    n = L.owo_render_midi(t.ctypes.data_as(C.c_void_p), ty.ctypes.data_as(C.c_void_p), no.ctypes.data_as(C.c_void_p), va.ctypes.data_as(C.c_void_p),
                          C.c_size_t(len(items)), C.c_double(volume), C.c_double(speaker), 1 if no_pa else 0, C.c_double(tail), out.ctypes.data_as(C.c_void_p),
                          C.c_size_t(cap), st)
    return out[:n], (int(st[0]), int(st[1]))


def test_oracle_voice_manager_semantics(oracle):
    # pedal defers the note-off to the pedal-up event (main.rs:1812-1847): same render as a note-off at the pedal-up time
    a, sa = _oracle_render(oracle, [(0.0, 0, 60, 100), (0.1, 2, 0, 1), (0.3, 1, 60, 0), (0.6, 2, 0, 0)], tail=0.5)
    b, sb = _oracle_render(oracle, [(0.0, 0, 60, 100), (0.6, 1, 60, 0)], tail=0.5)
    assert a.size == b.size == int(1.1 * 44100.0) and np.array_equal(a, b) and sa == sb == (1, 1)
    c, _ = _oracle_render(oracle, [(0.0, 0, 60, 100), (0.3, 1, 60, 0)], tail=0.8)
    assert c.size == a.size and not np.array_equal(a, c) and np.array_equal(a[: int(0.29 * 44100)], c[: int(0.29 * 44100)])
    # events fire at 64-sample chunk starts: a note at t = 1 sample sounds from sample 64 on
    d, _ = _oracle_render(oracle, [(1.0 / 44100.0, 0, 72, 127)], tail=0.05)
    assert np.all(d[:64] == 0.0) and np.any(d[64:128] != 0.0)
    # the 65th simultaneous note replaces the oldest voice outright; polyphony never exceeds 64
    many = [(0.0, 0, 33 + (k % 64), 90) for k in range(65)]
    e, se = _oracle_render(oracle, many, tail=0.1)
    assert se == (65, 64) and np.all(np.isfinite(e))
    # out-of-range keys clamp to 33..96 (main.rs:1784): same voice as the clamped key
    f, _ = _oracle_render(oracle, [(0.0, 0, 10, 100)], tail=0.1)
    g, _ = _oracle_render(oracle, [(0.0, 0, 33, 100)], tail=0.1)
    assert np.array_equal(f, g)
    assert _oracle_render(oracle, [])[0].size == 0
