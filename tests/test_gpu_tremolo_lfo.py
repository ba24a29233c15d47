"""`--features legacy-tremolo` (tremolo.rs:8,53-57,80-90,170-178): the behavioural 5.63 Hz sine LFO in place of the Twin-T circuit, as a
construction-time kind (OW_TREMOLO_LEGACY_LFO).  GPU against the oracle through the engine: R stream, preamp tap, output; reset and
rate change restart the LFO at phase 0; phase groups keep working (the LFO, too, is identical across engines built together)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
LFO = 1


def _compare(oracle, g, cs, blocks, length, tag, osr=2):
    for b in range(blocks):
        go = g.render(length)
        gr = g.tremolo_r(length * osr)
        gp = g.preamp_out(length * osr)
        for k, c in enumerate(cs):
            co, _, cp, cr = c.render_taps(length, osr=osr)
            # the oracle's r tap is the shunt impedance at the engine's depth; the device streams the CdS cell resistance and applies the
            # divider in the preamp kernel -- compare through the preamp tap and the output, and the cell through its envelope below
            # (preamp floor: the device's sin() and glibc's differ in the last place, which is the one-ulp experiment of DESIGN section 2 --
            # ABS_FLOOR_DENSE is what that experiment moves the oracle itself by; measured here 2.07e-9 on one sample)
            rp = oracle.parity_report(gp[k], cp, abs_floor=oracle.ABS_FLOOR_DENSE)
            ro = oracle.parity_report(go[k], co, abs_floor=oracle.ABS_FLOOR_OUTPUT)
            assert rp["n_bad"] == 0 and ro["n_bad"] == 0, (tag, b, k, rp, ro)
    return gr


@pytest.mark.parametrize("sr", [48000.0, 96000.0])
def test_legacy_lfo_engine_parity(hiplib, oracle, sr):
    import openwurli_amd as ow
    n, osr = 3, (2 if sr < 88200.0 else 1)
    g = ow.EnginePool(sr, n, tremolo_kind=LFO)
    cs = [oracle.OracleEngine(sr, tremolo_kind=LFO) for _ in range(n)]
    g.set_sample_rate(sr)
    for c in cs:
        c.set_sample_rate(sr)
    for k in range(n):
        for e in (g[k], cs[k]):
            e.set_tremolo_depth(1.0 - 0.4 * k); e.set_volume(0.6)
            for note in (45 + 3 * k, 60, 64 + k, 79):
                e.note_on(note, 0.75)
    r = _compare(oracle, g, cs, 12, 512, ("chords", sr), osr)
    assert g.tremolo_groups() == 1                                     # one LFO for the pool
    g[1].reset(); cs[1].reset()                                        # LFO of engine 1 restarts at phase 0: its own group
    for e in (g[1], cs[1]):
        e.note_on(52, 0.9)
    _compare(oracle, g, cs, 8, 333, ("after reset", sr), osr)
    assert g.tremolo_groups() == 2
    new_sr = 44100.0 if sr == 48000.0 else 88200.0
    g.set_sample_rate(new_sr)
    for c in cs:
        c.set_sample_rate(new_sr)
    for k in range(n):
        for e in (g[k], cs[k]):
            e.note_on(57 + k, 0.8)
    _compare(oracle, g, cs, 6, 256, ("re-rated", new_sr), 2 if new_sr < 88200.0 else 1)
    g.close()


def test_legacy_lfo_stream_is_the_rectified_sine(hiplib):
    """Shape of the R stream itself: 5.63 Hz, dark (1 MOhm) during the negative half wave after the 35 ms release, ~9.15 kOhm at the crest."""
    import openwurli_amd as ow
    sr = 48000.0
    g = ow.EnginePool(sr, 1, tremolo_kind=LFO)
    g.set_sample_rate(sr)
    rs = []
    for _ in range(48):                                                # 0.5 s at the 96 kHz chain rate
        g.render(512)
        rs.append(g.tremolo_r(1024)[0].copy())
    r = np.concatenate(rs)
    assert 9000.0 <= r.min() < 9300.0 and r.max() > 5e5, (r.min(), r.max())       # crest 9.15 kOhm (2.5 ms attack lag), 0.72 MOhm before the next rise
    bright = r < 2e4
    edges = np.flatnonzero(bright[1:] & ~bright[:-1])
    assert len(edges) >= 2
    period = np.diff(edges).mean() / 96000.0
    assert abs(1.0 / period - 5.63) < 0.02, 1.0 / period
    g.close()


def test_every_non_default_kind_together(hiplib, oracle):
    """A `--no-default-features --features melange-preamp,legacy-tremolo` build of the crate: melange 12-node preamp (literal rebuild),
    melange 7-BJT power amp with rail sag, legacy LFO tremolo -- one pool, against the oracle with the same three kinds."""
    import openwurli_amd as ow
    sr, n = 48000.0, 2
    g = ow.EnginePool(sr, n, preamp_kind=1, power_amp_kind=1, tremolo_kind=LFO)
    cs = [oracle.OracleEngine(sr, preamp_kind=1, power_amp_kind=1, tremolo_kind=LFO) for _ in range(n)]
    g.set_sample_rate(sr)
    for c in cs:
        c.set_sample_rate(sr)
    for k in range(n):
        for e in (g[k], cs[k]):
            e.set_tremolo_depth(0.9 - 0.5 * k); e.set_volume(0.4)
            for note in (50 + 2 * k, 62, 69):
                e.note_on(note, 0.7)
    for b in range(8):
        go = g.render(256)
        for k, c in enumerate(cs):
            rep = oracle.parity_report(go[k], c.render(256), abs_floor=oracle.ABS_FLOOR_MELANGE_LIT_OUTPUT)
            assert rep["n_bad"] == 0, (b, k, rep)
    d = g[0].power_amp_diag()
    assert d.guard_resets == cs[0].power_amp_diag()[3] and d.nan_resets == 0
    g.close()


def test_legacy_lfo_stagger_hook(hiplib):
    """ow_test_pool_stagger_tremolo on an LFO pool: four phase groups a quarter period apart, each engine reads its group's stream."""
    import openwurli_amd as ow
    sr = 48000.0
    g = ow.EnginePool(sr, 8, tremolo_kind=LFO)
    g.set_sample_rate(sr)
    g.stagger_tremolo(4)
    assert g.tremolo_groups() == 4
    g.render(512)
    r = g.tremolo_r(1024)
    for k in range(8):
        assert np.array_equal(r[k], r[k % 4])                    # engine k follows group k mod 4
    assert not np.array_equal(r[0], r[1]) and not np.array_equal(r[1], r[2])
    g.reset()                                                    # whole-pool reset: one group again, phase 0
    assert g.tremolo_groups() == 1
    g.close()
