"""GPU parity tests proper: the HIP path (through the C-ABI) against the CPU oracle on identical MIDI input.

Bar (BASELINE.json north_star): output within +-1e-5 relative (f32).  Metric:
  |gpu - cpu| <= max(1e-5 * max(|cpu|, 1e-3 * peak|cpu|), ABS_FLOOR)
where ABS_FLOOR = 2e-9 is the reference algorithm's own indeterminacy measured by
tests/test_oracle_sensitivity.py (Newton stop criterion 1e-9 V in the legacy preamp).  The f64 voice-sum tap is
held to 1e-12 of peak (same arithmetic, different libm only in transient phases).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

VEL = 100 / 127.0


def _check(rep, what):
    assert rep["n_bad"] == 0, (what, rep)


def _both(ow, ob, sr, warm=True, n=1):
    g = ow.EnginePool(sr, n)
    cs = [ob.OracleEngine(sr) for _ in range(n)]
    if warm:
        g.set_sample_rate(sr)
        for c in cs:
            c.set_sample_rate(sr)
    return g, cs


def _render_compare(ob, g, cs, blocks, length, what, check_taps=True, tap_rel=1e-12, tap_abs=0.0):
    for b in range(blocks):
        go = g.render(length)
        gv = g.voice_sum(length) if check_taps else None
        for i, c in enumerate(cs):
            co, cv, _, _ = c.render_taps(length)
            _check(ob.parity_report(go[i], co, abs_floor=ob.ABS_FLOOR_OUTPUT), (what, "out", b, i))
            if check_taps:
                rep = ob.parity_report(gv[i], cv, rel=tap_rel, floor_frac=1.0, abs_floor=tap_abs)
                _check(rep, (what, "voice_sum", b, i))


# ------------------------------------------------------------------ config 1: Voice::render_note
@pytest.mark.parametrize("midi,vel,sr", [(60, VEL, 48000.0), (60, VEL, 44100.0), (33, 1.0, 48000.0), (96, 0.3, 48000.0),
                                          (48, 0.5, 48000.0), (84, 0.85, 44100.0), (72, 20 / 127.0, 48000.0)])
def test_render_note(hiplib, oracle, midi, vel, sr):
    import openwurli_amd as ow
    g = ow.render_note(midi, vel, 0.5, sr)
    c = oracle.render_note(midi, vel, 0.5, sr)
    assert g.size == c.size == int(0.5 * sr)
    _check(oracle.parity_report(g, c, rel=1e-10, floor_frac=1.0), ("render_note", midi, vel, sr))


def test_render_note_config1_full_length(hiplib, oracle):
    """BASELINE configs[0]: single C4 note, 1 voice, 48 kHz, 2 s."""
    import openwurli_amd as ow
    g = ow.render_note(60, VEL, 2.0, 48000.0)
    c = oracle.render_note(60, VEL, 2.0, 48000.0)
    assert g.size == 96000
    _check(oracle.parity_report(g, c, rel=1e-10, floor_frac=1.0), "config1")
    # reed-renderer properties (tools/reed-renderer/tests/integration.rs): deterministic, velocity ordering
    assert np.array_equal(g, ow.render_note(60, VEL, 2.0, 48000.0))
    assert np.max(np.abs(ow.render_note(60, 1.0, 0.5, 48000.0))) > np.max(np.abs(g[:24000])) > np.max(np.abs(ow.render_note(60, 0.25, 0.5, 48000.0)))


# ------------------------------------------------------------------ engine scenarios
def test_engine_chord_with_tremolo(hiplib, oracle):
    import openwurli_amd as ow
    g, cs = _both(ow, oracle, 48000.0)
    for e in (g[0], cs[0]):
        e.set_volume(0.5); e.set_tremolo_depth(0.5); e.set_speaker_character(0.0); e.set_mlp_enabled(True)
        for n in (48, 60, 64, 67, 84):
            e.note_on(n, VEL)
    _render_compare(oracle, g, cs, 24, 512, "chord")
    assert g[0].active_voice_count() == cs[0].active_voice_count() == 5
    g.close()


def test_engine_fresh_without_warmup_and_speaker_ramp(hiplib, oracle):
    """WurliEngine::new without set_sample_rate (unit-test style), speaker character ramping through the 0.002 hysteresis."""
    import openwurli_amd as ow
    g, cs = _both(ow, oracle, 48000.0, warm=False)
    for e in (g[0], cs[0]):
        e.set_volume(0.8); e.set_tremolo_depth(1.0); e.set_speaker_character(0.7)
        for n in (45, 60, 64, 79):
            e.note_on(n, 0.8)
    _render_compare(oracle, g, cs, 8, 256, "fresh")
    for e in (g[0], cs[0]):
        e.set_speaker_character(1.0); e.set_volume(0.3); e.set_tremolo_depth(0.2)
    _render_compare(oracle, g, cs, 30, 333, "ramp")         # ragged block length
    g.close()


def test_engine_note_off_sustain_and_damper(hiplib, oracle):
    import openwurli_amd as ow
    g, cs = _both(ow, oracle, 48000.0)
    es = (g[0], cs[0])
    for e in es:
        e.set_speaker_character(0.2)
        for n in (40, 52, 59, 71, 90, 95):
            e.note_on(n, 0.9)
    _render_compare(oracle, g, cs, 6, 512, "held")
    for e in es:
        e.set_sustain(True); e.note_off(52); e.note_off(95)
    _render_compare(oracle, g, cs, 4, 512, "pedal")
    assert g[0].sustained_voice_count() == cs[0].count_voices_in_state(2) == 2
    for e in es:
        e.note_off(40); e.set_sustain(False); e.note_off(71)     # pedal release damps the sustained ones
    _render_compare(oracle, g, cs, 40, 512, "release")            # spans all three damper ramp classes (8/25/50 ms)
    for s in range(64):
        assert hiplib.ow_engine_slot_state(g[0]._h, s) == cs[0].slot_state(s)
    g.close()


def test_engine_voice_stealing_and_restrike(hiplib, oracle):
    """>64 notes: oldest-held stealing with the 5 ms crossfade; re-strike of releasing notes (config 2 event pattern)."""
    import openwurli_amd as ow
    g, cs = _both(ow, oracle, 48000.0)
    es = (g[0], cs[0])
    for e in es:
        for n in range(33, 97):
            e.note_on(n, 0.6)
    _render_compare(oracle, g, cs, 3, 512, "full", check_taps=True)
    for e in es:
        for n in (60, 61, 62):
            e.note_on(n, 0.9)                                       # steals three oldest Held voices
    assert g[0].diag().steal_voices == cs[0].steal_voice_count() == 3
    # during the crossfade the steal voices are summed as a separate pass (documented order deviation ~1e-16)
    _render_compare(oracle, g, cs, 2, 100, "steal-fade")
    _render_compare(oracle, g, cs, 2, 512, "after-fade")
    assert g[0].diag().steal_voices == cs[0].steal_voice_count() == 0
    for e in es:
        for n in range(33, 97):
            e.note_off(n); e.note_on(n, 0.7)                         # re-strike epoch
    _render_compare(oracle, g, cs, 4, 512, "restrike")
    for s in range(64):
        assert hiplib.ow_engine_slot_state(g[0]._h, s) == cs[0].slot_state(s)
        assert hiplib.ow_engine_slot_note(g[0]._h, s) == cs[0].slot_note(s)
    g.close()


def test_engine_voices_are_freed_when_silent(hiplib, oracle):
    import openwurli_amd as ow
    g, cs = _both(ow, oracle, 48000.0)
    for e in (g[0], cs[0]):
        e.note_on(96, 0.4); e.note_on(90, 0.4); e.note_off(90)
    counts_g, counts_c = [], []
    for _ in range(40):
        g.render(2048); cs[0].render(2048)
        counts_g.append(g[0].active_voice_count()); counts_c.append(cs[0].active_voice_count())
    assert counts_g == counts_c and counts_g[0] == 2 and counts_g[-1] == 0
    g.close()


def test_engine_reset(hiplib, oracle):
    import openwurli_amd as ow
    g, cs = _both(ow, oracle, 48000.0)
    for e in (g[0], cs[0]):
        e.set_tremolo_depth(0.8)
        e.note_on(55, 0.9)
    _render_compare(oracle, g, cs, 3, 512, "pre-reset")
    g[0].reset(); cs[0].reset()
    assert g[0].active_voice_count() == 0
    for e in (g[0], cs[0]):
        e.note_on(67, 0.7)
    _render_compare(oracle, g, cs, 6, 512, "post-reset")
    g.close()


@pytest.mark.parametrize("sr", [44100.0, 96000.0])
def test_engine_other_rates(hiplib, oracle, sr):
    """44.1 kHz (reference default; tremolo matrices rebuilt for 88.2 kHz) and config 3: 96 kHz host, no oversampling."""
    import openwurli_amd as ow
    g, cs = _both(ow, oracle, sr)
    for e in (g[0], cs[0]):
        e.set_speaker_character(0.5); e.set_tremolo_depth(0.6)
        for n in (36, 57, 64, 88):
            e.note_on(n, 0.75)
    _render_compare(oracle, g, cs, 10, 512, ("rate", sr))
    g.close()


def test_engine_ragged_block_lengths(hiplib, oracle):
    """Block lengths around every internal chunk size (voice tile 24, chain tiles 64, the one-sample software pipeline of the
    steady voice loop) and up to MAX_BLOCK_SIZE (engine.rs:25), with note events in between: transient -> steady hand-over
    between k_voice and k_voice_steady happens at arbitrary block boundaries.  The block-ahead tremolo mis-speculates on every
    length change, so its rollback is exercised throughout."""
    import openwurli_amd as ow
    sr = 48000.0
    g, cs = _both(ow, oracle, sr, n=2)
    for k in range(2):
        for e in (g[k], cs[k]):
            e.set_tremolo_depth(0.8); e.set_volume(0.6)
            for n in (36 + k, 55, 60, 64, 67 + k, 90):
                e.note_on(n, 0.4 + 0.3 * k)
    lengths = [1, 1, 2, 23, 24, 25, 47, 48, 49, 63, 64, 65, 100, 1, 511, 513, 1000, 3, 8192, 17]
    for i, length in enumerate(lengths):
        if i == 7:
            for k in range(2):
                for e in (g[k], cs[k]):
                    e.note_off(60); e.note_on(72, 0.9)
        if i == 13:
            for k in range(2):
                for e in (g[k], cs[k]):
                    e.note_on(60, 0.7); e.set_sustain(True); e.note_off(55)
        go = g.render(length)
        gv = g.voice_sum(length)
        for k, c in enumerate(cs):
            co, cv, _, _ = c.render_taps(length)
            assert go[k].size == length
            _check(oracle.parity_report(go[k], co, abs_floor=oracle.ABS_FLOOR_OUTPUT), ("ragged out", i, length, k))
            # f64 voice sum: 1e-12 of the block peak, with an absolute floor of 1e-15 (signal scale 1e-2; a 1-sample block
            # inside the onset ramp peaks at 4e-8, where one ulp of the OCML-vs-glibc cos() is already 1e-10 of "peak")
            _check(oracle.parity_report(gv[k], cv, rel=1e-12, floor_frac=1.0, abs_floor=1e-15), ("ragged voice_sum", i, length, k))
    assert g.render(0).shape == (2, 0)
    g.close()


def test_voices_keep_their_rate_constants_across_set_sample_rate(hiplib, oracle):
    """WurliEngine::set_sample_rate (engine.rs:272-286) rebuilds the chain but keeps the Voice objects: a voice struck at 48 kHz
    goes on with its 48 kHz phase increments, decay multipliers, pickup beta and jitter constants while the engine now runs at
    44.1 kHz (it sounds flat -- that is the reference's behaviour), and warm_up() renders 0.6 s of it.  New notes use the new rate."""
    import openwurli_amd as ow
    g, cs = _both(ow, oracle, 48000.0)
    es = (g[0], cs[0])
    for e in es:
        e.set_tremolo_depth(0.3)
        for n in (45, 64, 88):
            e.note_on(n, 0.7)
    _render_compare(oracle, g, cs, 3, 512, "before rate change")
    g.set_sample_rate(44100.0); cs[0].set_sample_rate(44100.0)
    _render_compare(oracle, g, cs, 3, 441, "old voices at the new rate")
    for e in es:
        e.note_off(64); e.note_on(52, 0.9)
    _render_compare(oracle, g, cs, 4, 441, "old and new voices")
    g.set_sample_rate(96000.0); cs[0].set_sample_rate(96000.0)           # no oversampling from here
    _render_compare(oracle, g, cs, 2, 960, "second change")
    g.close()


def test_tremolo_wide_is_bit_identical(hiplib, oracle):
    """Small pools run the Twin-T oscillator with four lanes per engine (ow_trem_wide.h: rows / ports / elimination rows spread over a
    quad); large pools with one lane per engine.  Both must produce the SAME bits: the settle of 50 + 2 sr steps at pool creation,
    then R[n] over several blocks incl. a block-length change (speculation rollback), at two rates; and both match the oracle."""
    import openwurli_amd as ow
    for sr in (48000.0, 44100.0):
        streams = {}
        for wide in ("0", "1"):
            os.environ["OW_TREM_WIDE"] = wide
            os.environ["OW_TREM_TRAJ"] = "0"        # per-group oscillators: on the shared trajectory (the default) neither kernel runs per pool
            try:
                p = ow.EnginePool(sr, 3)
                assert p.get_switch("trem_traj") == 0 and p.get_switch("trem_wide") == int(wide)
                p.set_sample_rate(sr)
                rs = []
                for length in (512, 512, 100, 512, 37):
                    p.render(length)
                    rs.append(p.tremolo_r(2 * length)[1].copy())
                p.close()
            finally:
                del os.environ["OW_TREM_WIDE"]
                del os.environ["OW_TREM_TRAJ"]
            streams[wide] = np.concatenate(rs)
        assert np.array_equal(streams["0"], streams["1"]), (sr, np.max(np.abs(streams["0"] - streams["1"])))
        assert 1e3 < streams["1"].min() < streams["1"].max() <= 1e6
        c = oracle.OracleEngine(sr)
        c.set_sample_rate(sr)
        ro = np.concatenate([c.render_taps(n)[3] for n in (512, 512, 100, 512, 37)])   # oracle tap = shunt impedance at depth 0.5
        r = streams["1"]
        shunt = 25000.0 * 18000.0 / (25000.0 + 18000.0) + 25000.0 * (680.0 + r) / (25000.0 + (680.0 + r))   # tremolo.rs:152-167
        assert np.max(np.abs(shunt - ro) / ro) < 1e-9


@pytest.mark.parametrize("sr", [64000.0, 88199.0, 88200.0, 176400.0, 192000.0])
def test_unusual_host_rates(hiplib, oracle, sr):
    """Host rates around the oversampling switch (engine.rs:195: 2x chain below 88.2 kHz, none from 88.2 kHz on) and far from the
    usual ones: rate-dependent constants (matrices, ramps, decay multipliers, steal fade length) all come from the pool's rate.
    (Below 40 kHz the speaker's 20 kHz low-pass lies above Nyquist and the reference algorithm itself blows up -- 32 kHz reaches
    1e38 on both sides -- so there is nothing to compare there.)"""
    import openwurli_amd as ow
    g, cs = _both(ow, oracle, sr)
    for e in (g[0], cs[0]):
        e.set_tremolo_depth(0.7)
        for n, v in ((40, 0.9), (64, 0.5), (88, 1.0)):
            e.note_on(n, v)
    _render_compare(oracle, g, cs, 3, 400, ("rate", sr))
    for e in (g[0], cs[0]):
        e.note_off(64); e.note_on(64, 0.8); e.set_sustain(True); e.note_off(40)
    _render_compare(oracle, g, cs, 3, 333, ("rate-2", sr))
    g.close()


def test_chain_wide_is_bit_identical(hiplib, oracle):
    """Job paths run the legacy preamp with a quad of lanes per solver state while jobs are few (ow_chain_wide.h: the rows of S and the
    two junction exponentials spread over the quad); with many jobs a lane pair per job.  Both must produce the SAME bits, at
    both rates (with and without oversampling), with and without power amp / speaker character, and both match the oracle.
    (Static LDR values stay inside the cell's range, >= 2e4 ohm: at a few kilohm the preamp leaves its operating region, output peaks
    of 1e4, and neither side is a reference for the other any more.)"""
    import openwurli_amd as ow
    jobs = [dict(note=n, velocity=v, mlp=bool(k & 1), poweramp=bool(k & 2), volume=1.0 - 0.1 * (k % 4), speaker=0.25 * (k % 5), r_ldr=r)
            for k, (n, v, r) in enumerate([(33, 127, 1e6), (48, 50, 1e6), (60, 100, 2.5e4), (72, 20, 1e6), (84, 127, 4e4), (91, 64, 1e6), (96, 110, 1e5),
                                           (40, 90, 1e6), (55, 35, 7e5), (67, 80, 1e6), (79, 127, 1e6)])]
    for sr in (44100.0, 96000.0):
        outs = {}
        # "1": the quad chain as preamp | output stage on two wavefronts (k_job_chain_fused) with the voices rendered BESIDE it on a second
        # stream (k_job_voice publishes its progress, the chain waits chunk by chunk); "1-after-voices": the same chain behind the voices
        # ("1" with few jobs is k_job_chain_row -- one solver state per row of sixteen lanes, four preamp wavefronts + the output-stage one per
        # eight jobs, ow_chain_row.h; "1-quad" (OW_JOB_ROW=0) is the quad-lane k_job_chain_fused)
        for wide in ("0", "1", "1-quad", "1-one-wavefront", "1-after-voices"):
            os.environ["OW_CHAIN_WIDE"] = wide[0]
            if wide in ("0", "1-one-wavefront"):
                os.environ["OW_JOB_FUSED"] = "0"
            if wide == "1-after-voices":
                os.environ["OW_JOB_OVERLAP"] = "0"
            if wide == "1-quad":
                os.environ["OW_JOB_ROW"] = "0"
            try:
                outs[wide] = ow.batch_render(jobs, sr, 0.35)
            finally:
                del os.environ["OW_CHAIN_WIDE"]
                os.environ.pop("OW_JOB_FUSED", None); os.environ.pop("OW_JOB_OVERLAP", None); os.environ.pop("OW_JOB_ROW", None)
        for other in ("0", "1-quad", "1-one-wavefront", "1-after-voices"):
            assert np.array_equal(outs[other], outs["1"]), (sr, other, np.max(np.abs(outs[other] - outs["1"])))
        for k in (0, 2, 4, 10):
            j = jobs[k]
            c = oracle.batch_render_job(j["note"], j["velocity"], 0.35, sr, volume=j["volume"], speaker=j["speaker"], r_ldr=j["r_ldr"],
                                        mlp=j["mlp"], poweramp=j["poweramp"])
            rep = oracle.parity_report(outs["1"][k][:c.size], c, abs_floor=oracle.ABS_FLOOR_BATCH)
            assert rep["n_bad"] == 0, (sr, k, rep)


def test_preamp_wide_is_bit_identical(hiplib, oracle):
    """Small pools run the legacy preamp with a quad of lanes per solver state (k_preamp_wide), large ones with a lane pair per engine
    (k_preamp).  Same bits at the preamp tap and at the output, with tremolo, depth ramps, reset and the steal pass in play, at both
    rates; 11 engines = one full wavefront of the wide kernel and a ragged second one."""
    import openwurli_amd as ow
    for sr in (48000.0, 96000.0):
        osr = 2 if sr < 88200.0 else 1
        res = {}
        for wide in ("0", "1"):
            os.environ["OW_PREAMP_WIDE"] = wide
            try:
                g = ow.EnginePool(sr, 11)
                g.set_sample_rate(sr)
                for k in range(11):
                    g[k].set_tremolo_depth(0.1 * k); g[k].set_volume(0.3 + 0.05 * k)
                    for note in (40 + 3 * k, 60 + k, 72):
                        g[k].note_on(note, 0.5 + 0.04 * k)
                outs, pres = [], []
                for b in range(10):
                    if b == 3:
                        g[4].set_tremolo_depth(1.0); g[7].note_on(60 + 7, 1.0)              # depth ramp; re-strike of a sounding key (steal pass)
                    if b == 6:
                        g[2].reset(); g[2].note_on(55, 0.9)
                    outs.append(g.render(300 if b % 2 else 512).copy())
                    pres.append(g.preamp_out((300 if b % 2 else 512) * osr).copy())
                res[wide] = (outs, pres)
                g.close()
            finally:
                del os.environ["OW_PREAMP_WIDE"]
        for b in range(10):
            assert np.array_equal(res["0"][1][b], res["1"][1][b]), (sr, b, "preamp tap")
            assert np.array_equal(res["0"][0][b], res["1"][0][b]), (sr, b, "output")
        assert np.max(np.abs(res["1"][0][5])) > 1e-3


def test_pool_of_independent_engines(hiplib, oracle):
    """Lane = engine kernels: 5 engines with different scripts in one pool vs 5 separate oracle engines."""
    import openwurli_amd as ow
    n = 5
    g, cs = _both(ow, oracle, 48000.0, n=n)
    for k in range(n):
        for e in (g[k], cs[k]):
            e.set_volume(0.3 + 0.1 * k); e.set_tremolo_depth(0.2 * k); e.set_speaker_character(0.25 * (k % 3)); e.set_mlp_enabled(k % 2 == 0)
            for j in range(k + 1):
                e.note_on(40 + 7 * j + k, (40 + (37 * k) % 88) / 127.0)
    _render_compare(oracle, g, cs, 8, 512, "pool")
    for k in range(n):
        for e in (g[k], cs[k]):
            e.note_off(40 + k)
    _render_compare(oracle, g, cs, 8, 512, "pool-release")
    g.close()


def test_packed_dispatch_many_sparse_engines(hiplib, oracle):
    """Packed voice dispatch: 14 engines sounding 1..9 voices each share wavefronts (lane = sounding voice); the voices of one
    engine are never split across blocks and its ordered sum is taken over its own lanes.  Engines move between the steady and
    the general list as their onset / noise / damper phases begin and end, at different times per engine."""
    import openwurli_amd as ow
    sr, n = 48000.0, 14
    g, cs = _both(ow, oracle, sr, n=n)
    keys = [33, 38, 45, 52, 57, 60, 64, 69, 76, 84, 91, 96]
    for k in range(n):
        for e in (g[k], cs[k]):
            e.set_tremolo_depth(0.1 * (k % 5))
            for j in range(k % 9 + 1):
                e.note_on(keys[(k + 2 * j) % len(keys)], 0.35 + 0.05 * ((k + j) % 10))
    _render_compare(oracle, g, cs, 5, 512, "sparse: strike")          # all general at first, steady once onsets and noise are over
    for k in range(n):
        for e in (g[k], cs[k]):
            if k % 3 == 0:
                e.note_off(keys[k % len(keys)])                       # damper phase: back to the general list
            elif k % 3 == 1:
                e.note_on(keys[(k + 5) % len(keys)], 0.8)             # new onset in an otherwise steady engine
    # released voices: exp() of the damper ramp is OCML on the device and glibc in the oracle (1 ulp), visible at 2e-12 of a quiet block
    _render_compare(oracle, g, cs, 6, 512, "sparse: mixed phases", tap_rel=1e-11, tap_abs=1e-15)
    for k in range(0, n, 2):
        for e in (g[k], cs[k]):
            e.set_sustain(True)
            e.note_off(keys[(k + 2) % len(keys)])
    _render_compare(oracle, g, cs, 3, 300, "sparse: sustained", tap_rel=1e-11, tap_abs=1e-15)   # damped voices ring out: block peaks fall to 1e-6 (signal scale 1e-2)
    assert [g[k].active_voice_count() for k in range(n)] == [cs[k].active_voice_count() for k in range(n)]
    from openwurli_amd import binding
    assert "voice dispatch" not in binding.last_error(hiplib)
    g.close()


def test_played_random_script_against_oracle(hiplib, oracle):
    """A played part per engine (random note-ons / note-offs / pedal at block boundaries, like tools/bench_midi.py) against one oracle
    engine each: engines keep hopping between the steady and the general voice list, voices are stolen, freed and restruck, and
    several engines share each wavefront.  Slot states and voice counts are compared after every block."""
    import openwurli_amd as ow
    from openwurli_amd import binding
    sr, n, blocks, length = 48000.0, 7, 36, 256
    g, cs = _both(ow, oracle, sr, n=n)
    rng = np.random.default_rng(2024)
    for k in range(n):
        for e in (g[k], cs[k]):
            e.set_tremolo_depth(0.15 * k); e.set_volume(0.4 + 0.05 * k)
    held = [[] for _ in range(n)]
    for b in range(blocks):
        for k in range(n):
            r = rng.random()
            if r < 0.45 + 0.05 * k:                                   # busier engines get more notes; engine 6 fills up and steals
                note, vel = int(rng.integers(33, 97)), float(rng.uniform(0.3, 1.0))
                for e in (g[k], cs[k]):
                    e.note_on(note, vel)
                held[k].append(note)
            if held[k] and rng.random() < 0.35:
                note = held[k].pop(int(rng.integers(0, len(held[k]))))
                for e in (g[k], cs[k]):
                    e.note_off(note)
            if rng.random() < 0.08:
                on = bool(rng.integers(0, 2))
                for e in (g[k], cs[k]):
                    e.set_sustain(on)
        go = g.render(length)
        gv = g.voice_sum(length)
        for k, c in enumerate(cs):
            co, cv, _, _ = c.render_taps(length)
            _check(oracle.parity_report(go[k], co, abs_floor=oracle.ABS_FLOOR_OUTPUT), ("played out", b, k))
            _check(oracle.parity_report(gv[k], cv, rel=1e-11, floor_frac=1.0, abs_floor=1e-15), ("played voice_sum", b, k))
            assert g[k].active_voice_count() == c.active_voice_count(), (b, k)
            assert [g[k].slot_state(i) for i in range(64)] == [c.slot_state(i) for i in range(64)], (b, k)
    assert "voice dispatch" not in binding.last_error(hiplib)
    g.close()


def test_config2_all_keys_restrike_against_oracle(hiplib, oracle):
    """BASELINE configs[1] event script: all 64 keys, 1.0 s re-strike, buffers of 512, 1.5 s."""
    import openwurli_amd as ow
    sr = 48000.0
    g, cs = _both(ow, oracle, sr)
    es = (g[0], cs[0])
    for e in es:
        e.set_volume(0.5); e.set_tremolo_depth(0.5); e.set_speaker_character(0.0); e.set_mlp_enabled(True)
    pos, total = 0, int(1.5 * sr)
    worst = 0.0
    while pos < total:
        if pos % 48000 == 0:
            for e in es:
                for n in range(33, 97):
                    if pos:
                        e.note_off(n)
                    e.note_on(n, VEL)
        length = min(512, 48000 - pos % 48000, total - pos)
        go = g.render(length)[0]
        co = cs[0].render(length)
        rep = oracle.parity_report(go, co, abs_floor=oracle.ABS_FLOOR_OUTPUT)
        worst = max(worst, rep["worst_ratio"])
        _check(rep, ("config2", pos))
        pos += length
    assert g[0].active_voice_count() == cs[0].active_voice_count() == 64
    g.close()


def test_pool_midi_threaded_paths_agree(hiplib):
    """ow_pool_midi applies large event lists on several host threads: a list grouped by engine (sliced at engine
    boundaries), the same events interleaved across engines (every thread filters the list), and per-engine calls must
    leave the pool in the same state -- per-engine event order is all that matters."""
    import openwurli_amd as ow
    from openwurli_amd import binding
    n_eng, sr = 160, 48000.0
    rng = np.random.default_rng(11)
    per = 40
    ev = np.zeros((n_eng, per), dtype=np.dtype(binding.MIDI_DTYPE))
    ev["engine"] = np.arange(n_eng, dtype=np.uint32)[:, None]
    ev["note"] = rng.integers(33, 97, size=(n_eng, per))
    ev["type"] = rng.choice([0, 0, 0, 1, 2], size=(n_eng, per))
    ev["value"] = rng.uniform(0.2, 1.0, size=(n_eng, per)).astype(np.float32)
    grouped = ev.reshape(-1)
    interleaved = np.ascontiguousarray(ev.T).reshape(-1)      # engine index cycles: ungrouped, same per-engine order
    assert grouped.size >= 4096

    def run(mode):
        p = ow.EnginePool(sr, n_eng)
        if mode == "grouped":
            p.midi(grouped)
        elif mode == "interleaved":
            p.midi(interleaved)
        else:
            for k in range(n_eng):
                for x in ev[k]:
                    if x["type"] == 0:
                        p[k].note_on(int(x["note"]), float(x["value"]))
                    elif x["type"] == 1:
                        p[k].note_off(int(x["note"]))
                    else:
                        p[k].set_sustain(bool(x["value"] >= 0.5))
        out = np.concatenate([p.render(256) for _ in range(3)], axis=1)
        counts = [p[k].active_voice_count() for k in range(n_eng)]
        p.close()
        return out, counts
    a, ca = run("grouped")
    b, cb = run("interleaved")
    c, cc = run("calls")
    assert ca == cb == cc
    assert np.array_equal(a, b) and np.array_equal(a, c)
    assert np.max(np.abs(a)) > 1e-3
    # the steady kernel flags a voice it finds inside a transient phase (host classification of the packed dispatch went wrong)
    assert "voice dispatch" not in binding.last_error(hiplib)


@pytest.mark.parametrize("sr", [48000.0, 96000.0])
def test_configs_2_and_3_in_full(hiplib, oracle, sr):
    """configs 2 and 3 IN FULL -- 10 s, all 64 keys, re-strike every 1.0 s (ten epochs of release-steal + 5 ms crossfades),
    buffers of 512, 48 kHz host / 96 kHz chain and 96 kHz host without oversampling; a second instance runs the speaker at
    character 1.0 (polynomial + tanh + thermal path) with the per-instance velocity of config 5."""
    import openwurli_amd as ow
    g, cs = _both(ow, oracle, sr, n=2)
    for k in range(2):
        for e in (g[k], cs[k]):
            e.set_volume(0.5); e.set_tremolo_depth(0.5); e.set_speaker_character(1.0 * k); e.set_mlp_enabled(True)
    epoch, total, pos, worst = int(sr), int(10.0 * sr), 0, 0.0
    while pos < total:
        if pos % epoch == 0:
            for k in range(2):
                for e in (g[k], cs[k]):
                    for n in range(33, 97):
                        if pos:
                            e.note_off(n)
                        e.note_on(n, (40 + 37 * k % 88) / 127.0 if k else VEL)
        length = min(512, epoch - pos % epoch, total - pos)
        go = g.render(length)
        for k in range(2):
            rep = oracle.parity_report(go[k], cs[k].render(length), abs_floor=oracle.ABS_FLOOR_OUTPUT)
            worst = max(worst, rep["worst_ratio"])
            _check(rep, ("full config", sr, pos, k))
        pos += length
    assert worst < 1.0 and all(g[k].active_voice_count() == cs[k].active_voice_count() for k in range(2))
    g.close()


def test_steady_kernel_voice_sum_over_twelve_seconds(hiplib, oracle):
    """the steady voice kernel carries amplitude x envelope as ONE recurrence (deviation 9: one rounding per sample instead of
    two, envelope recovered as ae / amplitude at every block end).  Held bass and mid notes, no events for 12 s, so every block after the
    onset runs k_voice_steady.  Its fused multiply-adds and the single amp x env rounding make the oscillator phases a random walk
    against the reference's: measured on MI355X the f64 voice-sum tap stays within 1e-12 of the block's peak for the first 9 s and
    reaches 1.1e-12 at 9.4 s (the block peak has decayed to 1.4e-4 by then).  Asserted: 1e-12 for 5 s, 5e-12 for the rest -- seven
    orders of magnitude inside the 1e-5 output bar, which the output itself is held to on every block (absolute floor: the dense-play
    one, 5e-9 -- seven voices at 0.9 under a moving tremolo gain; one sample in the 12 s reaches 3.2e-9 = 1.5e-6 of its block's peak
    where the preamp's Newton loop stops one iteration apart, tests/test_oracle_sensitivity.py::test_dense_play_floor)."""
    import openwurli_amd as ow
    g, cs = _both(ow, oracle, 48000.0)
    for e in (g[0], cs[0]):
        e.set_tremolo_depth(0.3)
        for n in (33, 36, 40, 45, 52, 57, 60):
            e.note_on(n, 0.9)
    worst_tap = 0.0
    for b in range(int(12.0 * 48000 / 2048)):
        go = g.render(2048)
        gv = g.voice_sum(2048)
        co, cv, _, _ = cs[0].render_taps(2048)
        _check(oracle.parity_report(go[0], co, abs_floor=oracle.ABS_FLOOR_DENSE), ("steady 12 s", "out", b))
        rep = oracle.parity_report(gv[0], cv, rel=1e-12 if (b + 1) * 2048 <= 5 * 48000 else 5e-12, floor_frac=1.0)
        _check(rep, ("steady 12 s", "voice_sum", b))
        worst_tap = max(worst_tap, rep["max_err_rel_peak"])
    assert g[0].active_voice_count() == cs[0].active_voice_count() >= 3       # the bass notes are still sounding
    assert worst_tap < 5e-12
    g.close()


# ------------------------------------------------------------------ size-independent properties at full size
def test_properties_full_size(hiplib):
    """64-voice instances at the bench size: determinism (two pools, same script -> bit-identical), volume linearity
    (engine.rs:839-882), bounded output and no NaN-guard activity."""
    import openwurli_amd as ow
    sr = 48000.0

    def run(vol):
        p = ow.EnginePool(sr, 8)
        p.set_sample_rate(sr)
        for k in range(8):
            p[k].set_volume(vol); p[k].set_speaker_character(0.0)
            for n in range(33, 97):
                p[k].note_on(n, (40 + (37 * k) % 88) / 127.0)
        out = np.concatenate([p.render(512) for _ in range(20)], axis=1)
        d = [p[k].diag() for k in range(8)]
        p.close()
        return out, d
    a, da = run(0.5)
    b, _ = run(0.5)
    assert np.array_equal(a, b)
    assert np.all(np.isfinite(a)) and np.max(np.abs(a)) < 4.0
    assert all(x.nan_guard_fires == 0 and x.preamp_nan_resets == 0 and x.output_nan_resets == 0 and x.active_voices == 64 for x in da)
    c, _ = run(0.25)
    ratio = np.max(np.abs(a), axis=1) / np.max(np.abs(c), axis=1)
    assert np.all(np.abs(ratio - 2.0) < 0.04)
    assert not np.array_equal(a[0], a[1])                   # instances with different velocities differ


# ------------------------------------------------------------------ batch path (config 4)
@pytest.mark.parametrize("sr", [44100.0, 48000.0])
def test_batch_render_jobs(hiplib, oracle, sr):
    """16-job subset of config 4: notes {33,48,60,72,84,91,96,40} x velocities {50,127}, render_model_notes flags."""
    import openwurli_amd as ow
    jobs = [{"note": n, "velocity": v} for n in (33, 48, 60, 72, 84, 91, 96, 40) for v in (50, 127)]
    dur = 5.0 if sr == 48000.0 else 0.75        # config 4 length (5 s) at 48 kHz; the reference's default rate gets a short run
    g = ow.batch_render(jobs, sample_rate=sr, duration_s=dur)
    assert g.shape == (16, int(dur * sr))
    for i, j in enumerate(jobs):
        c = oracle.batch_render_job(j["note"], j["velocity"], dur, sr)
        _check(oracle.parity_report(g[i], c, abs_floor=oracle.ABS_FLOOR_BATCH), ("job", j, sr))


def test_batch_render_variants(hiplib, oracle):
    """Non-default job flags: MLP on, power amp on (base rate, drive = volume^2), speaker character, low static LDR,
    and a job count that is not a multiple of the wavefront (ragged last block)."""
    import openwurli_amd as ow
    jobs = [{"note": 40 + 3 * k, "velocity": 30 + 5 * k, "mlp": k % 2 == 0, "poweramp": k % 3 == 0, "volume": 0.4 + 0.03 * k,
             "speaker": (k % 4) / 4.0, "r_ldr": 19000.0 if k % 5 == 0 else 1e6} for k in range(19)]
    dur, sr = 0.4, 48000.0
    g = ow.batch_render(jobs, sample_rate=sr, duration_s=dur)
    for i, j in enumerate(jobs):
        c = oracle.batch_render_job(j["note"], j["velocity"], dur, sr, volume=j["volume"], speaker=j["speaker"], r_ldr=j["r_ldr"],
                                    mlp=j["mlp"], poweramp=j["poweramp"])
        _check(oracle.parity_report(g[i], c, abs_floor=oracle.ABS_FLOOR_BATCH), ("variant", i, j))


def test_batch_wav_quantiser_round_trip(hiplib, oracle):
    """24-bit WAV quantiser of preamp-bench (round, clamp: main.rs:941-957) applied to GPU and oracle renders agrees to 1 LSB."""
    import openwurli_amd as ow
    g = ow.batch_render([{"note": 60, "velocity": 127}], sample_rate=44100.0, duration_s=0.5)[0]
    c = oracle.batch_render_job(60, 127, 0.5, 44100.0)
    mx = 2 ** 23 - 1
    qg = np.clip(np.round(g * mx), -mx, mx).astype(np.int64)
    qc = np.clip(np.round(c * mx), -mx, mx).astype(np.int64)
    assert np.max(np.abs(qg - qc)) <= 1


# ------------------------------------------------------------------ note-on MLP on the f64 matrix cores
def test_mlp_mfma(hiplib, oracle):
    """v_mfma_f64_16x16x4_f64 batch MLP (what k_apply_ops runs) vs the scalar lane path vs the oracle's clamped outputs."""
    import ctypes as C
    rng = np.random.default_rng(7)
    n = 200                                                   # not a multiple of 64: ragged last wavefront
    notes = rng.integers(21, 109, size=n).astype(np.uint8)
    vels = rng.random(n)
    raw_m = np.zeros((n, 11)); raw_s = np.zeros((n, 11))
    assert hiplib.ow_debug_mlp_raw(notes.ctypes.data_as(C.c_void_p), vels.ctypes.data_as(C.c_void_p), n, raw_m.ctypes.data_as(C.c_void_p), 1, 0) == 0
    assert hiplib.ow_debug_mlp_raw(notes.ctypes.data_as(C.c_void_p), vels.ctypes.data_as(C.c_void_p), n, raw_s.ctypes.data_as(C.c_void_p), 0, 0) == 0
    assert np.all(np.isfinite(raw_m))
    scale = np.maximum(np.abs(raw_s), 1.0)
    assert np.max(np.abs(raw_m - raw_s) / scale) < 1e-13       # same network, different summation order / fusion
    # oracle's finished corrections (fade + clamps, mlp_correction.rs:118-133) from the MFMA raw outputs
    L = oracle.lib()
    out = np.zeros(11)
    for i in range(n):
        midi = int(notes[i])
        L.owo_mlp_infer(midi, C.c_double(vels[i]), out.ctypes.data_as(C.c_void_p))
        if midi < 65:
            fade = min(max((midi - 53) / 12.0, 0.0), 1.0)
        elif midi > 97:
            fade = min(max((109 - midi) / 12.0, 0.0), 1.0)
        else:
            fade = 1.0
        if fade <= 0:
            continue
        cents = np.clip(raw_m[i, :5] * fade, -100, 100)
        decay = 1.0 + (np.clip(raw_m[i, 5:10], 0.3, 3.0) - 1.0) * fade
        ds = 1.0 + (np.clip(raw_m[i, 10], 0.7, 1.2) - 1.0) * fade
        assert np.max(np.abs(cents - out[:5])) < 1e-10 and np.max(np.abs(decay - out[5:10])) < 1e-12 and abs(ds - out[10]) < 1e-12


# ------------------------------------------------------------------ melange 12-node preamp (second solver)
@pytest.mark.parametrize("kernel", ["literal", "rank1"])
def test_melange_engine_parity(hiplib, oracle, kernel, monkeypatch):
    """The default kernel re-factors the 12x12 system per sample like the reference (ow_melange_lit.h) and is held to the LEGACY preamp's
    floors (2e-9 V at the node; measured 3.5e-10 steady, 5.1e-10 under depth-knob ramps).  OW_MEL_RANK1=1 selects the rank-one kernel: its
    floors are the loose ones of deviation 6 (1.8e-7 V while R_ldr moves), asserted relative to peak."""
    import openwurli_amd as ow
    if kernel == "rank1":
        monkeypatch.setenv("OW_MEL_RANK1", "1")
    else:
        monkeypatch.delenv("OW_MEL_RANK1", raising=False)
    fl_p = oracle.ABS_FLOOR_MELANGE_PREAMP if kernel == "rank1" else oracle.ABS_FLOOR_PREAMP
    fl_o = oracle.ABS_FLOOR_MELANGE_OUTPUT if kernel == "rank1" else oracle.ABS_FLOOR_OUTPUT
    sr = 48000.0
    g = ow.EnginePool(sr, 2, preamp_kind=1)
    cs = [oracle.OracleEngine(sr, preamp_kind=1) for _ in range(2)]
    g.set_sample_rate(sr)
    for c in cs:
        c.set_sample_rate(sr)
    for k in range(2):
        for e in (g[k], cs[k]):
            e.set_volume(0.5); e.set_tremolo_depth(0.5 + 0.5 * k); e.set_speaker_character(0.0)
            for n in (48, 60, 67) if k == 0 else (40, 72, 76, 91):
                e.note_on(n, 0.8)
    for b in range(12):
        if b in (4, 8):                       # depth-knob ramps: R_ldr moves fast -- where the rank-one kernel leaves the reference's LU noise
            for k in range(2):
                for e in (g[k], cs[k]):
                    e.set_tremolo_depth(0.1 if b == 4 else 1.0)
        go = g.render(512)
        gp = g.preamp_out(1024)
        for k in range(2):
            co, _, cp, _ = cs[k].render_taps(512)
            rp = oracle.parity_report(gp[k], cp, abs_floor=fl_p)
            ro = oracle.parity_report(go[k], co, abs_floor=fl_o)
            _check(rp, ("melange preamp", b, k)); _check(ro, ("melange out", b, k))
            assert rp["max_err_rel_peak"] < 1e-5 and ro["max_err_rel_peak"] < 1e-5      # the north-star bar, relative to peak
    for k in range(2):
        d = g[k].diag()
        assert d.preamp_nan_resets == 0
    g.close()


def test_melange_literal_fast_path_is_the_generic_rebuild_bit_for_bit(hiplib, monkeypatch):
    """ow_melange_lit.h: the precomputed-leading-block rebuild (default) and the plain LU of the whole 12x12 system (OW_MEL_GENERIC=1)
    perform the same operations on the R-dependent entries: preamp and output streams must be bit-identical, at two rates, under a
    tremolo at full depth, depth-knob ramps and a static shunt."""
    import openwurli_amd as ow
    for sr in (48000.0, 44100.0):
        res = {}
        for mode in ("fast", "generic"):
            if mode == "generic":
                monkeypatch.setenv("OW_MEL_GENERIC", "1")
            else:
                monkeypatch.delenv("OW_MEL_GENERIC", raising=False)
            g = ow.EnginePool(sr, 3, preamp_kind=1)
            g.set_sample_rate(sr)
            for k in range(3):
                g[k].set_tremolo_depth((1.0, 0.5, 0.0)[k])
                for n in (45 + 7 * k, 60, 72):
                    g[k].note_on(n, 0.9)
            outs, pres = [], []
            for b in range(8):
                if b == 3:
                    g[1].set_tremolo_depth(1.0); g[2].set_tremolo_depth(0.7)
                outs.append(g.render(256)); pres.append(g.preamp_out(512))
            g.close()
            res[mode] = (np.concatenate(outs, axis=1), np.concatenate(pres, axis=1))
        assert np.array_equal(res["fast"][1], res["generic"][1]) and np.array_equal(res["fast"][0], res["generic"][0]), sr
        assert np.max(np.abs(res["fast"][1])) > 1e-3


def test_melange_lane_engine_kernel_is_bit_identical(hiplib, monkeypatch):
    """ow_melange_eng.h (one lane per engine: factorisation, unit columns and S N_i once for both solver states, the per-state parts as
    a rolled loop) against k_preamp_mel_col (one lane per state): the same operations on the same operands per state, so preamp and
    output streams are bit-identical -- 70 engines (two workgroups of the lane = engine kernel, a ragged tail in both), thermal noise on
    some, depth ramps, a reset and a re-rate mid-run, the generic rebuild (OW_MEL_GENERIC=1) through both kernels as well."""
    import openwurli_amd as ow
    sr, n = 48000.0, 70
    res = {}
    for mode in ("col", "eng", "eng_generic"):
        monkeypatch.setenv("OW_MEL_ENG", "0" if mode == "col" else "1")
        if mode == "eng_generic":
            monkeypatch.setenv("OW_MEL_GENERIC", "1")
        else:
            monkeypatch.delenv("OW_MEL_GENERIC", raising=False)
        g = ow.EnginePool(sr, n, preamp_kind=1)
        assert g.get_switch("mel_eng") == (0 if mode == "col" else 1)
        g.set_sample_rate(sr)
        g.stagger_tremolo(23)
        for k in range(n):
            g[k].set_tremolo_depth((1.0, 0.5, 0.0, 0.8)[k % 4]); g[k].set_volume(0.3 + 0.01 * (k % 7))
            if k % 5 == 0:
                g[k].set_noise_seed(1000 + k); g[k].set_noise_gain(3.0); g[k].set_noise_enabled(True)
            for m in (40 + (3 * k) % 40, 60 + k % 12, 84):
                g[k].note_on(m, 0.5 + 0.05 * (k % 9))
        outs, pres = [], []
        for b in range(7):
            if b == 2:
                for k in range(0, n, 3):
                    g[k].set_tremolo_depth(0.1 + 0.9 * ((k // 3) % 2))
            if b == 4:
                g[5].reset(); g[5].note_on(64, 0.9); g[69].reset(); g[69].note_on(50, 0.7)
            length = (256, 100, 256, 1, 256, 333, 64)[b]
            outs.append(g.render(length)); pres.append(g.preamp_out(2 * length))
        d = [g[k].diag().preamp_nan_resets for k in (0, 5, 69)]
        g.close()
        res[mode] = (np.concatenate(outs, axis=1), np.concatenate(pres, axis=1), d)
    for mode in ("eng", "eng_generic"):
        assert np.array_equal(res[mode][1], res["col"][1]), mode
        assert np.array_equal(res[mode][0], res["col"][0]), mode
        assert res[mode][2] == res["col"][2]
    assert np.max(np.abs(res["col"][1])) > 1e-3


def test_melange_thermal_noise_parity(hiplib, oracle):
    """set_noise_enabled / set_noise_gain on the melange preamp (engine.rs:394-400; gen_preamp.rs:3433-3461): the 11 resistor
    noise currents are integer-exact xoshiro256++ streams shaped by Marsaglia polar (log, sqrt), so with a fixed seed
    (ow_engine_set_noise_seed = gen_preamp::set_seed, which WurliEngine itself never calls) the GPU must follow the oracle
    sample for sample: same floors as the noise-free melange test.  Covers idle hiss, notes over hiss, gain changes at block
    rate, switching off and on (lag kept), reset (streams restart, settings kept) and set_sample_rate (settings dropped)."""
    import openwurli_amd as ow
    sr = 48000.0
    g = ow.EnginePool(sr, 3, preamp_kind=1)
    cs = [oracle.OracleEngine(sr, preamp_kind=1) for _ in range(3)]
    g.set_sample_rate(sr)
    for c in cs:
        c.set_sample_rate(sr)
    seeds = (12345, 0xDEADBEEFCAFE, 777)
    for k in range(3):
        for e in (g[k], cs[k]):
            e.set_noise_seed(seeds[k])
            e.set_tremolo_depth(0.5 * k)
            e.set_noise_gain((1.0, 0.05, 30.0)[k])
            if k != 1:
                e.set_noise_enabled(True)          # engine 1 stays silent until block 6

    def compare(tag, blocks, length=512):
        idle_peak = 0.0
        for b in range(blocks):
            go = g.render(length)
            gp = g.preamp_out(2 * length)
            for k in range(3):
                co, _, cp, _ = cs[k].render_taps(length)
                rp = oracle.parity_report(gp[k], cp, abs_floor=oracle.ABS_FLOOR_MELANGE_LIT_PREAMP)
                ro = oracle.parity_report(go[k], co, abs_floor=oracle.ABS_FLOOR_MELANGE_LIT_OUTPUT)
                _check(rp, (tag, "preamp", b, k)); _check(ro, (tag, "out", b, k))
            idle_peak = max(idle_peak, float(np.max(np.abs(go[0]))))
        return go, idle_peak

    out, idle = compare("idle hiss", 4)
    assert 1e-5 < idle < 1e-3 and np.max(np.abs(out[1])) == 0.0        # ~-91 dBFS RMS at gain 1 (CHANGELOG: -86 dBFS before the 0.6.0 gain change)
    assert np.max(np.abs(out[2])) > 10 * np.max(np.abs(out[0]))        # gain 30x
    for k in range(3):
        for e in (g[k], cs[k]):
            e.note_on(60 + 4 * k, 0.7)
    compare("notes over hiss", 2, 333)
    for e in (g[1], cs[1]):
        e.set_noise_enabled(True)
    for e in (g[0], cs[0]):
        e.set_noise_enabled(False)
    for e in (g[2], cs[2]):
        e.set_noise_gain(0.0)                                           # scale_half == 0: no draws, streams frozen
    compare("switches", 3)
    for e in (g[0], cs[0]):
        e.set_noise_enabled(True)
    for e in (g[2], cs[2]):
        e.set_noise_gain(2.0)
    compare("back on", 2)
    for k in range(3):
        g[k].reset(); cs[k].reset()                                     # streams restart from the engine's seed, settings survive
    out, _ = compare("after reset", 2)
    assert np.max(np.abs(out[0])) > 1e-6
    g.set_sample_rate(44100.0)
    for c in cs:
        c.set_sample_rate(44100.0)                                      # DkPreamp::new: noise off again
    out, _ = compare("after rate change", 1, 441)
    assert all(np.max(np.abs(out[k])) < 1e-7 for k in range(3))         # no hiss: the new DkPreamp starts with noise off
    g.close()


def test_melange_noise_streams_are_deterministic_and_independent(hiplib):
    """Same seed -> bit-identical render; different seeds -> different hiss; default = one process-wide seed for every engine
    (the reference clones one cached state, melange_adapter.rs:12-29)."""
    import openwurli_amd as ow
    sr = 48000.0

    def run(seeds, init=True):
        p = ow.EnginePool(sr, len(seeds), preamp_kind=1)
        if init:
            p.set_sample_rate(sr)
        for k, s in enumerate(seeds):
            if s is not None:
                p[k].set_noise_seed(s)
            p[k].set_noise_enabled(True)
        out = np.concatenate([p.render(256) for _ in range(4)], axis=1)
        p.close()
        return out
    a = run([5, 5, 6, None, None])
    b = run([5, 5, 6, None, None])
    assert np.array_equal(a, b)
    assert np.array_equal(a[0], a[1]) and not np.array_equal(a[0], a[2])
    assert np.array_equal(a[3], a[4]) and np.max(np.abs(a[3])) > 1e-6
    c = run([None] * 70, init=False)                    # a fresh pool (engines replicated from engine 0, no set_sample_rate call)
    assert np.max(np.abs(c[69])) > 1e-6 and np.array_equal(c[0], c[69])


def test_melange_static_ldr_and_reset(hiplib, oracle):
    import openwurli_amd as ow
    sr = 44100.0
    g = ow.EnginePool(sr, 1, preamp_kind=1)
    c = oracle.OracleEngine(sr, preamp_kind=1)
    for e in (g[0], c):
        e.set_tremolo_depth(0.0)             # static shunt: matrices rebuilt only while the depth smoother ramps
        e.note_on(57, 0.9)
    for b in range(6):
        go = g.render(512)[0]
        co = c.render(512)
        _check(oracle.parity_report(go, co, abs_floor=oracle.ABS_FLOOR_MELANGE_LIT_OUTPUT), ("melange static", b))
    g[0].reset(); c.reset()
    for e in (g[0], c):
        e.note_on(64, 0.7)
    for b in range(6):
        go = g.render(512)[0]
        co = c.render(512)
        _check(oracle.parity_report(go, co, abs_floor=oracle.ABS_FLOOR_MELANGE_LIT_OUTPUT), ("melange reset", b))
    g.close()


def test_batch_render_melange(hiplib, oracle):
    """preamp-bench render built with `--features melange-preamp`: static LDR at several values incl. the 100 kOhm nominal."""
    import openwurli_amd as ow
    jobs = [{"note": n, "velocity": v, "r_ldr": r} for (n, v, r) in
            ((48, 100, 1e6), (60, 127, 1e6), (72, 50, 19000.0), (84, 90, 1e5), (91, 127, 47000.0))]
    dur, sr = 0.4, 44100.0
    g = ow.batch_render(jobs, sample_rate=sr, duration_s=dur, preamp_kind=1)
    for i, j in enumerate(jobs):
        c = oracle.batch_render_job(j["note"], j["velocity"], dur, sr, r_ldr=j["r_ldr"], preamp_kind=1)
        # static LDR: the job's matrices come from ONE literal rebuild (ow_melange_lit.h), like the reference's set_pot -- the batch
        # floor of the legacy path holds (it was 6e-6 with the rank-one matrices)
        rep = oracle.parity_report(g[i], c, abs_floor=oracle.ABS_FLOOR_MELANGE_LIT_OUTPUT)
        print("melange job", j, "max abs err", rep.get("max_abs_err"), "rel peak", rep["max_err_rel_peak"])
        _check(rep, ("melange job", j))
        assert rep["max_err_rel_peak"] < 1e-5


def test_chain_fused_is_bit_identical(hiplib, oracle):
    """Small pools run the preamp and the output stage as two wavefronts of ONE launch (k_chain_fused: quad-lane preamp on chunk c, power
    amp / half-band / speaker / gain on chunk c - 1, handed over through LDS) instead of k_preamp_wide then k_post.  Same bits at the
    preamp tap and at the output -- tremolo, depth / volume / speaker-character ramps through the 0.002 hysteresis, reset of one engine,
    the steal pass, ragged block lengths incl. 1 and lengths that are no multiple of the 16-sample chunk, an output NaN guard event and
    the blocks after it, at both rates; 11 engines = one full workgroup and a ragged second one.  And it is the oracle's engine."""
    import openwurli_amd as ow
    lengths = (512, 300, 1, 17, 64, 16, 15, 777, 512, 33)
    for sr in (48000.0, 96000.0):
        osr = 2 if sr < 88200.0 else 1
        res = {}
        for fused in (0, 1):
            g = ow.EnginePool(sr, 11)
            g.set_sample_rate(sr)
            g.set_switch("chain_fused", fused)
            assert g.get_switch("chain_fused") == fused
            for k in range(11):
                g[k].set_tremolo_depth(0.1 * k); g[k].set_volume(0.3 + 0.05 * k); g[k].set_speaker_character(0.08 * k)
                for note in (40 + 3 * k, 60 + k, 72):
                    g[k].note_on(note, 0.5 + 0.04 * k)
            outs, pres, diags = [], [], []
            for b, length in enumerate(lengths):
                if b == 3:
                    g[4].set_tremolo_depth(1.0); g[7].note_on(60 + 7, 1.0); g[5].set_speaker_character(1.0)
                if b == 5:
                    g[9].set_volume(1e308)                     # unbounded set_volume (the reference does not clamp): a non-finite output -> NaN guard
                if b == 6:
                    g[2].reset(); g[2].note_on(55, 0.9); g[9].set_volume(0.5)
                outs.append(g.render(length).copy())
                pres.append(g.preamp_out(length * osr).copy())
                diags.append([(g[k].diag().output_nan_resets, g[k].diag().preamp_nan_resets) for k in (2, 9)])
            res[fused] = (outs, pres, diags)
            g.close()
        for b in range(len(lengths)):
            assert np.array_equal(res[0][1][b], res[1][1][b]), (sr, b, "preamp tap")
            assert np.array_equal(res[0][0][b], res[1][0][b]), (sr, b, "output")
            assert res[0][2][b] == res[1][2][b], (sr, b, "diag")
        assert res[1][2][-1][1][0] >= 1                                # the guard did fire on engine 9
    # the fused path against the oracle (the default for a pool of one)
    g, c = ow.WurliEngine(48000.0), oracle.OracleEngine(48000.0)
    g.set_sample_rate(48000.0); c.set_sample_rate(48000.0)
    for e in (g, c):
        e.set_tremolo_depth(0.8); e.set_speaker_character(0.4)
        for n in (45, 60, 64, 79):
            e.note_on(n, 0.8)
    for length in (64, 64, 128, 7, 512):
        rep = oracle.parity_report(g.render(length), c.render(length), abs_floor=oracle.ABS_FLOOR_OUTPUT)
        assert rep["n_bad"] == 0, (length, rep)
    g.close(); c.close()


def test_chain_row_is_bit_identical(hiplib, oracle):
    """The smallest pools run the fused chain launch with ONE preamp state per row of sixteen lanes (k_chain_row, ow_chain_row.h: lane r =
    matrix row r, one v_mov_b64_dpp per value that crosses lanes, the Newton sweeps of a wavefront's four states in one uniform loop, four
    preamp wavefronts + the output-stage wavefront per eight engines) instead of the quad-lane k_chain_fused.  Same bits at the preamp
    tap and at the output as k_chain_fused AND as the two-launch path -- tremolo at every depth incl. 0 (the LDR hysteresis never fires)
    and 1, depth / volume / speaker-character ramps, reset of one engine, the steal pass, ragged block lengths incl. 1 and lengths that
    are no multiple of the 16-sample chunk, an output NaN guard event and the blocks after it (the deferred preamp reset), at both rates;
    pools of 11 (one full workgroup and a ragged one with an odd engine count: a preamp wavefront with one engine) and of 1."""
    import openwurli_amd as ow
    lengths = (512, 300, 1, 17, 64, 16, 15, 777, 512, 33)
    for sr, n_eng in ((48000.0, 11), (96000.0, 11), (48000.0, 1)):
        osr = 2 if sr < 88200.0 else 1
        res = {}
        for mode in ("two_launches", "quad", "row"):
            g = ow.EnginePool(sr, n_eng)
            g.set_sample_rate(sr)
            g.set_switch("chain_fused", 0 if mode == "two_launches" else 1)
            g.set_switch("chain_row", 1 if mode == "row" else 0)
            for k in range(n_eng):
                g[k].set_tremolo_depth(0.1 * k); g[k].set_volume(0.3 + 0.05 * k); g[k].set_speaker_character(0.08 * k)
                for note in (40 + 3 * k, 60 + k, 72):
                    g[k].note_on(note, 0.5 + 0.04 * k)
            outs, pres, diags = [], [], []
            hot = min(9, n_eng - 1)
            for b, length in enumerate(lengths):
                if b == 3:
                    g[min(4, n_eng - 1)].set_tremolo_depth(1.0); g[min(7, n_eng - 1)].note_on(60 + 7, 1.0); g[min(5, n_eng - 1)].set_speaker_character(1.0)
                if b == 5:
                    g[hot].set_volume(1e308)                   # a non-finite output -> NaN guard -> deferred preamp / oversampler reset
                if b == 6:
                    g[min(2, n_eng - 1)].reset(); g[min(2, n_eng - 1)].note_on(55, 0.9); g[hot].set_volume(0.5)
                outs.append(g.render(length).copy())
                pres.append(g.preamp_out(length * osr).copy())
                diags.append([(g[k].diag().output_nan_resets, g[k].diag().preamp_nan_resets) for k in sorted({min(2, n_eng - 1), hot})])
            res[mode] = (outs, pres, diags)
            g.close()
        for mode in ("quad", "row"):
            for b in range(len(lengths)):
                assert np.array_equal(res["two_launches"][1][b], res[mode][1][b]), (sr, n_eng, mode, b, "preamp tap")
                assert np.array_equal(res["two_launches"][0][b], res[mode][0][b]), (sr, n_eng, mode, b, "output")
                assert res["two_launches"][2][b] == res[mode][2][b], (sr, n_eng, mode, b, "diag")
        assert res["row"][2][-1][-1][0] >= 1                           # the guard did fire
        assert max(float(np.abs(o).max()) for o in res["row"][0]) > 1e-3
    # the default for a pool of one is the row kernel: against the oracle
    g, c = ow.WurliEngine(48000.0), oracle.OracleEngine(48000.0)
    for e in (g, c):
        e.set_tremolo_depth(0.8); e.set_speaker_character(0.4)
        for n in (45, 60, 64, 79):
            e.note_on(n, 0.8)
    for length in (64, 64, 128, 7, 512):
        rep = oracle.parity_report(g.render(length), c.render(length), abs_floor=oracle.ABS_FLOOR_OUTPUT)
        assert rep["n_bad"] == 0, (length, rep)
    g.close(); c.close()


def test_preamp_nan_reset_in_every_chain_kernel(hiplib, oracle):
    """The preamp's OWN NaN reset (dk_preamp_legacy.rs:610-615: `main - shadow` non-finite -> both solver states back to the DC solution at the
    current R_ldr, this sample's output 0.0, the block goes on) cannot be reached through the API; a test hook overwrites a node voltage
    of a solver state with NaN (main of engine 1, shadow of engine 3) or infinity (engine 4, a block later) before a block.  Every chain kernel -- two launches
    with the lane-pair and the quad-lane preamp, the fused quad chain, the row chain, the stream chain -- must count the same resets and
    produce the same bits at the preamp tap and the output, in that block and after it; and they are the oracle's (same poke, same block)."""
    import openwurli_amd as ow
    sr, n_eng, length = 48000.0, 5, 96
    modes = {"lane pairs": {"chain_fused": 0, "preamp_wide": 0, "chain_stream": 0}, "quad, two launches": {"chain_fused": 0, "preamp_wide": 1},
             "quad, fused": {"chain_fused": 1, "chain_row": 0}, "row": {"chain_fused": 1, "chain_row": 1},
             "stream": {"chain_fused": 0, "preamp_wide": 0, "chain_stream": 1}}
    res = {}
    cs = [oracle.OracleEngine(sr) for _ in range(n_eng)]
    for name in list(modes) + ["oracle"]:
        g = None
        if name != "oracle":
            g = ow.EnginePool(sr, n_eng); g.set_sample_rate(sr)
            for k, v in modes[name].items():
                g.set_switch(k, v)
        es = [g[k] for k in range(n_eng)] if g is not None else cs
        for k, e in enumerate(es):
            if g is None:
                e.set_sample_rate(sr)
            e.set_tremolo_depth(0.2 * k); e.set_volume(0.5)
            for note in (45 + k, 60, 67 + k):
                e.note_on(note, 0.8)
        outs, pres = [], []
        for b in range(5):
            if b == 2:
                es[1].poke_preamp_node(6, float("nan")); es[3].poke_preamp_node(2, float("nan"), shadow=True)
            if b == 3:
                es[4].poke_preamp_node(0, float("inf"))                    # an infinite node voltage: inf - inf inside the step
            if g is not None:
                outs.append(g.render(length).copy()); pres.append(g.preamp_out(2 * length).copy())
            else:
                taps = [e.render_taps(length) for e in es]
                outs.append(np.stack([t[0] for t in taps])); pres.append(np.stack([t[2] for t in taps]))
        resets = [es[k].diag().preamp_nan_resets for k in range(n_eng)] if g is not None else None      # (the reference keeps no counter of them)
        res[name] = (outs, pres, resets)
        if g is not None:
            g.close()
    ref = res["lane pairs"]
    assert ref[2][1] >= 1 and ref[2][3] >= 1 and ref[2][4] >= 1 and ref[2][0] == 0 and ref[2][2] == 0, ref[2]
    for name in modes:
        assert res[name][2] == ref[2], (name, res[name][2], ref[2])
        for b in range(5):
            assert np.array_equal(res[name][1][b], ref[1][b]), (name, b, "preamp tap")
            assert np.array_equal(res[name][0][b], ref[0][b]), (name, b, "output")
            assert np.all(np.isfinite(res[name][0][b]))
    for b in range(5):
        for k in range(n_eng):
            rep = oracle.parity_report(ref[0][b][k], res["oracle"][0][b][k], abs_floor=oracle.ABS_FLOOR_OUTPUT)
            assert rep["n_bad"] == 0, (b, k, rep)
    for c in cs:
        c.close()


def test_preamp_state_rows_are_the_references_fields_in_every_chain_kernel(hiplib, oracle):
    """dk_step carries the junction evaluation from one step into the next (i_nl beside a register-only gm: DESIGN.md section 4.1) -- the
    state rows in HBM stay the reference's DkState (dk_preamp_legacy.rs:231-239).  After every block of a scenario with notes, a depth
    ramp, a release and a reset, under every chain kernel: the fourteen fields of main and shadow are the same bits in every kernel, and
    the oracle's within the preamp bar (volts: 1e-5 relative + the preamp floor; currents and the capacitor's companion source, which
    are ~1e-4 A and ~1e-6 A: the same relative bar + 1e-12)."""
    import openwurli_amd as ow
    sr, n_eng = 48000.0, 3
    modes = {"lane pairs": {"chain_fused": 0, "preamp_wide": 0}, "quad, two launches": {"chain_fused": 0, "preamp_wide": 1},
             "quad, fused": {"chain_fused": 1, "chain_row": 0}, "row": {"chain_fused": 1, "chain_row": 1}}
    lens = [64, 97, 512, 1, 33, 256, 128, 512]
    states = {}
    for name in list(modes) + ["oracle"]:
        g = None
        if name != "oracle":
            g = ow.EnginePool(sr, n_eng); g.set_sample_rate(sr)
            for k, v in modes[name].items():
                g.set_switch(k, v)
            es = [g[k] for k in range(n_eng)]
        else:
            es = [oracle.OracleEngine(sr) for _ in range(n_eng)]
            for e in es:
                e.set_sample_rate(sr)
        for k, e in enumerate(es):
            e.set_tremolo_depth(0.4 * k); e.set_volume(0.5)
            for note in (43 + 2 * k, 60, 72 + k):
                e.note_on(note, 0.85)
        rows = []
        for b, n in enumerate(lens):
            if b == 2:
                es[1].set_tremolo_depth(1.0)
            if b == 4:
                es[0].note_off(60)
            if b == 6:
                es[2].reset()
            if g is not None:
                g.render(n)
                rows.append(np.stack([np.stack([e.read_preamp_state(False), e.read_preamp_state(True)]) for e in es]))
            else:
                for e in es:
                    e.render(n)
                rows.append(np.stack([np.stack([e.preamp_state(False)[:14], e.preamp_state(True)[:14]]) for e in es]))
        states[name] = rows
        if g is not None:
            g.close()
        else:
            for e in es:
                e.close()
    ref = states["lane pairs"]
    for name in modes:
        for b in range(len(lens)):
            assert states[name][b].tobytes() == ref[b].tobytes(), (name, b)
    for b in range(len(lens)):
        a, o = ref[b], states["oracle"][b]
        volts = np.abs(a[..., 2:10] - o[..., 2:10]) <= 1e-5 * np.abs(o[..., 2:10]) + oracle.ABS_FLOOR_PREAMP
        vnl = np.abs(a[..., 12:14] - o[..., 12:14]) <= 1e-5 * np.abs(o[..., 12:14]) + oracle.ABS_FLOOR_PREAMP
        amps = np.abs(a[..., [0, 1, 10, 11]] - o[..., [0, 1, 10, 11]]) <= 1e-5 * np.abs(o[..., [0, 1, 10, 11]]) + 1e-12
        assert volts.all() and vnl.all() and amps.all(), (b, a - o)


def test_post_pair_is_bit_identical(hiplib):
    """The oversampled output stage with lane = engine (k_post<false, true>: both chain samples of an output sample solved in one lane,
    the half-band / speaker part run once; the default for ranges of >= 131 072 engines) against the lane-pair form (k_post<true>): the
    same bits at the output through notes, speaker-character and volume ramps, a ragged pool (not a multiple of 64 engines) and blocks
    that are not multiples of the 64-sample store chunk."""
    import openwurli_amd as ow
    sr, n_eng = 48000.0, 97
    outs = {}
    for pair in (0, 1):
        g = ow.EnginePool(sr, n_eng); g.set_sample_rate(sr)
        g.set_switch("chain_fused", 0); g.set_switch("preamp_wide", 0); g.set_switch("post_pair", pair)
        assert g.get_switch("post_pair") == pair
        for k in range(n_eng):
            e = g[k]
            e.set_tremolo_depth((k % 5) * 0.25); e.set_volume(0.3 + 0.05 * (k % 8)); e.set_speaker_character((k % 3) * 0.5)
            for note in (40 + k % 30, 60, 65 + k % 20):
                e.note_on(note, 0.5 + 0.005 * k)
        blocks = []
        for b, n in enumerate([512, 97, 64, 1, 300, 512]):
            if b == 2:
                for k in range(0, n_eng, 7):
                    g[k].set_speaker_character(1.0); g[k].set_volume(0.9)
            blocks.append(g.render(n).copy())
        outs[pair] = blocks
        g.close()
    for b in range(len(outs[0])):
        assert np.any(outs[0][b] != 0.0)
        assert outs[0][b].tobytes() == outs[1][b].tobytes(), b


def test_chain_stream_is_bit_identical(hiplib, oracle):
    """Big oversampled pools whose block goes to a pinned host block run preamp and output stage as ONE launch (k_chain_stream,
    ow_chain_stream.h: one wavefront per 32 engines alternates between the two per 64-sample chunk and stores the f32 rows straight into
    the caller's mapped block) instead of k_preamp, k_post and a device-to-host copy behind them.  Same bits at the preamp tap and at the
    output as the two launches -- tremolo, depth / volume / speaker-character ramps, a reset, the steal pass, ragged block lengths incl.
    1 and lengths that are no multiple of the chunk, an output NaN guard event and the blocks after it; 70 engines = two full workgroups
    and a ragged third.  The host block (row stride larger than the block) equals the block left in HBM."""
    import ctypes
    import openwurli_amd as ow
    lengths = (512, 300, 1, 17, 64, 65, 15, 777, 512, 33)
    sr, n = 48000.0, 70
    res = {}
    stride = 800
    for streamed in (0, 1):
        g = ow.EnginePool(sr, n)
        g.set_sample_rate(sr)
        g.ensure_buffer_capacity(1024)
        g.set_switch("preamp_wide", 0); g.set_switch("chain_fused", 0)
        g.set_switch("chain_stream", streamed); g.set_switch("out_direct", streamed)
        assert g.get_switch("chain_stream") == streamed
        host = g.alloc_host_block(stride)
        hview = np.ctypeslib.as_array((ctypes.c_float * (n * stride)).from_address(host[0])).reshape(n, stride)
        for k in range(n):
            g[k].set_tremolo_depth(0.013 * k); g[k].set_volume(0.3 + 0.005 * k); g[k].set_speaker_character(0.014 * k)
            for note in (40 + k % 30, 60 + k % 11, 72):
                g[k].note_on(note, 0.5 + 0.006 * k)
        outs, pres, diags = [], [], []
        for b, length in enumerate(lengths):
            if b == 3:
                g[4].set_tremolo_depth(1.0); g[37].note_on(60 + 7, 1.0); g[65].set_speaker_character(1.0)
                for note in range(33, 97):
                    g[33].note_on(note, 0.6)
            if b == 4:
                for note in range(33, 97):                  # the steal pass
                    g[33].note_off(note); g[33].note_on(note, 0.7)
            if b == 5:
                g[69].set_volume(1e308)                     # a non-finite output -> NaN guard
            if b == 6:
                g[2].reset(); g[2].note_on(55, 0.9); g[69].set_volume(0.5)
            hview[:] = -7.0
            g.render_into(host[0], stride, length)
            blk = g.last_block()[:, :length].copy()
            assert np.array_equal(hview[:, :length], blk), (streamed, b, "host block")
            assert np.all(hview[:, length:] == -7.0)
            outs.append(blk)
            pres.append(g.preamp_out(length * 2).copy())
            diags.append([(g[k].diag().output_nan_resets, g[k].diag().preamp_nan_resets) for k in (2, 69)])
        res[streamed] = (outs, pres, diags)
        g.free_host_block(host)
        g.close()
    for b in range(len(lengths)):
        assert np.array_equal(res[0][1][b], res[1][1][b]), (b, "preamp tap")
        assert np.array_equal(res[0][0][b], res[1][0][b]), (b, "output")
        assert res[0][2][b] == res[1][2][b], (b, "diag")
    assert res[1][2][-1][1][0] >= 1                                # the guard did fire on engine 69
    assert all(np.all(np.isfinite(o)) for o in res[1][0])
    assert max(float(np.max(np.abs(o))) for o in res[1][0]) > 1e-3


def test_skewed_voice_clocks_are_bit_identical(hiplib, oracle):
    """k_voice_steady<true> (ow_kernels.h): voices struck at different samples update their jitter on different 16-sample grids; the skewed
    variant delays each lane by 0..15 loop trips so that all updates of a wavefront share trips.  Every voice performs the operations it
    performed before, so voice sums and output are bit-identical to the plain loop (OW_VOICE_SKEW=0) -- keys struck one sample apart, a
    sparse engine and a dense one sharing wavefronts, ragged block lengths incl. some below the 32-sample limit of the skewed loop --
    and both follow the oracle."""
    import openwurli_amd as ow
    sr, n = 48000.0, 5
    res = {}
    for skew in (1, 0):
        p = ow.EnginePool(sr, n)
        p.set_switch("voice_skew", skew)
        p.set_sample_rate(sr)
        cs = [oracle.OracleEngine(sr) for _ in range(n)] if skew else None
        if cs:
            for c in cs:
                c.set_sample_rate(sr)
        outs, sums = [], []
        def both(f):
            for k in range(n):
                f(k, p[k])
                if cs:
                    f(k, cs[k])
        def render(length):
            o = p.render(length); outs.append(o.copy()); sums.append(p.voice_sum(length).copy())
            if cs:
                for k in range(n):
                    co, cv, _, _ = cs[k].render_taps(length)
                    rep = oracle.parity_report(o[k], co, abs_floor=oracle.ABS_FLOOR_OUTPUT)
                    assert rep["n_bad"] == 0, (k, length, rep)
        keys = {0: list(range(40, 90)), 1: [45, 52], 2: list(range(33, 97)), 3: [60], 4: [36, 48, 60, 72, 84, 96]}
        for step in range(64):                                   # one key per sample where the engine still has keys left
            both(lambda k, e: e.note_on(keys[k][step], 0.7) if step < len(keys[k]) else None)
            render(1)
        for b in range(30):                                      # past onset ramps and attack noise, then long enough for several renormalisations
            render((512, 300, 31, 512, 33, 777, 64, 512)[b % 8])
        for k in range(n):
            assert p[k].active_voice_count() == len(keys[k])
        assert p.get_switch("voice_skew_active") == 1          # the plain loop reported more than one grid per wavefront; skew=0 only ignores it
        p.close()
        if cs:
            for c in cs:
                c.close()
        res[skew] = (np.concatenate(outs, axis=1), np.concatenate(sums, axis=1))
    assert np.array_equal(res[1][1], res[0][1])
    assert np.array_equal(res[1][0], res[0][0])
    assert np.max(np.abs(res[1][0])) > 1e-3


def test_attack_variant_follows_the_oracle_and_the_general_kernel(hiplib, oracle):
    """k_voice_steady<false, true> (ow_kernels.h): engines inside onset ramps / attack noise whose slot voices are not damping render on the
    steady loop with the onset gain tabulated per chunk and the noise burst beside it, instead of on the general kernel.  Scenarios: one
    note; the whole keyboard at once; a key per block over ragged block lengths (incl. lengths below the chunk); a key released while
    others are still in their onset (-> the engine moves to the general kernel and stays there while the voice damps); pedal down + note
    off (Sustained: not damping, stays on the attack variant) and pedal up (-> general); a re-struck key (steal pass beside the attack).
    Every block follows the oracle; against OW_VOICE_ATTACK=0 (general kernel for every phase) voice sums differ by rounding only; the
    lists of the two runs show where the engines went; no misdispatch is reported (the render would fail)."""
    import openwurli_amd as ow
    sr, n = 48000.0, 6
    lengths = (512, 64, 7, 300, 1, 33, 777, 512, 24, 25, 512, 512, 512, 512, 512, 512)
    res, blocks = {}, {}
    for attack in (1, 0):
        p = ow.EnginePool(sr, n)
        p.set_switch("voice_attack", attack)
        p.set_sample_rate(sr)
        cs = [oracle.OracleEngine(sr) for _ in range(n)] if attack else None
        if cs:
            for c in cs:
                c.set_sample_rate(sr)
        def both(k, f):
            f(p[k])
            if cs:
                f(cs[k])
        sums, outs, blk = [], [], []
        both(0, lambda e: e.note_on(60, 0.8))
        for note in range(33, 97):
            both(1, lambda e: e.note_on(note, 0.3 + 0.01 * (note - 33)))
        both(3, lambda e: (e.note_on(48, 0.9), e.note_on(55, 0.6)))
        both(4, lambda e: (e.set_sustain(True), e.note_on(50, 0.7), e.note_on(62, 0.7)))
        both(5, lambda e: e.note_on(70, 1.0))
        for b, length in enumerate(lengths):
            if b < 12:
                both(2, lambda e: e.note_on(36 + 5 * b, 0.4 + 0.05 * b))
            if b == 1:
                both(3, lambda e: (e.note_off(48), e.note_on(67, 0.5)))      # a damper phase beside a fresh onset
            if b == 2:
                both(4, lambda e: e.note_off(50))                            # pedal down: Sustained, no damper
            if b == 5:
                both(4, lambda e: e.set_sustain(False))                      # the damper starts now
            if b in (1, 3):
                both(5, lambda e: (e.note_off(70), e.note_on(70, 0.9)))      # re-struck: the old voice damps in its slot
            o = p.render(length)
            outs.append(o.copy()); sums.append(p.voice_sum(length).copy())
            blk.append((p.get_switch("blocks_attack"), p.get_switch("blocks_general"), p.get_switch("blocks_steady")))
            if cs:
                for k in range(n):
                    co, cv, _, _ = cs[k].render_taps(length)
                    floor = oracle.ABS_FLOOR_DENSE if k == 1 else oracle.ABS_FLOOR_OUTPUT
                    rep = oracle.parity_report(o[k], co, abs_floor=floor)
                    assert rep["n_bad"] == 0, (k, b, length, rep)
                    vs = sums[-1][k]
                    assert np.max(np.abs(vs - cv)) <= 2e-9 * max(1.0, float(np.max(np.abs(cv)))), (k, b, float(np.max(np.abs(vs - cv))))
        p.close()
        if cs:
            for c in cs:
                c.close()
        res[attack] = (np.concatenate(outs, axis=1), np.concatenate(sums, axis=1))
        blocks[attack] = blk
    assert all(a == 0 for a, _, _ in blocks[0])
    assert blocks[1][0][0] >= 2 and blocks[1][0][1] == 0            # first block: every engine on the attack variant (the keyboard fills a block of its own)
    assert blocks[1][1][1] >= 1 and blocks[1][1][0] >= 1            # engines 3 and 5 moved to the general kernel, the others did not
    assert blocks[1][-1][0] == 0 and blocks[1][-1][2] >= 1          # phases over: steady (damping voices of 3 / 4 / 5 may still sound)
    scale = max(1.0, float(np.max(np.abs(res[0][1]))))
    assert np.max(np.abs(res[1][1] - res[0][1])) <= 1e-10 * scale, float(np.max(np.abs(res[1][1] - res[0][1])))
    assert np.max(np.abs(res[1][0] - res[0][0])) <= 1e-6
    assert np.max(np.abs(res[1][0])) > 1e-3


def test_steal_variant_follows_the_oracle_and_the_general_kernel(hiplib, oracle):
    """k_voice_steady<false, 2> (ow_kernels.h): the steal voices of an engine during their 5 ms crossfade render on the steady loop with the
    damper's factors and the crossfade gain beside it, instead of on the general kernel; each block decides by itself (voice_steal_takes).
    Engines: 0 the whole keyboard re-struck (every steal voice inside its damper ramp); 1 three more keys on a full keyboard (held voices
    stolen: no damper); 2 two more keys in the block of the first strike (voices stolen inside their onset ramp: the general kernel's);
    3 the top keys (no damper at all) released and re-struck with the rest; 4 released 64 ms before the re-strike (past every damper ramp:
    the multipliers); 5 released 42 ms before (bass ramps end inside the crossfade).  Ragged block lengths incl. some below the chunk.
    Every block follows the oracle; against OW_VOICE_STEAL=0 voice sums differ by rounding only."""
    import openwurli_amd as ow
    sr, n = 48000.0, 6
    lengths = (512, 512, 512, 512, 512, 100, 7, 133, 24, 25, 512, 1, 512, 512, 512, 300, 512)
    res = {}
    keys = list(range(33, 97))
    for steal in (1, 0):
        p = ow.EnginePool(sr, n)
        p.set_switch("voice_steal", steal)
        assert p.get_switch("voice_steal") == steal
        p.set_sample_rate(sr)
        cs = [oracle.OracleEngine(sr) for _ in range(n)] if steal else None
        if cs:
            for c in cs:
                c.set_sample_rate(sr)
        def both(k, f):
            f(p[k])
            if cs:
                f(cs[k])
        def chord(vel):
            return lambda e: [e.note_on(note, vel + 0.004 * (note - 33)) for note in keys]
        def release(e):
            for note in keys:
                e.note_off(note)
        sums, outs = [], []
        for k in range(n):
            both(k, chord(0.45 + 0.05 * k))
        both(2, lambda e: (e.note_on(50, 0.9), e.note_on(51, 0.8)))            # 65th and 66th voice in the block of the strike
        for b, length in enumerate(lengths):
            if b == 5:
                both(0, release); both(0, chord(0.7))                          # the benchmark's re-strike
                both(1, lambda e: [e.note_on(note, 0.8) for note in (40, 60, 80)])
                both(3, release); both(3, chord(0.6))
                both(4, release)
            if b == 7:
                both(5, release)
            if b == 11:
                both(4, chord(0.65))                                           # 64.6 ms after the release
                both(5, chord(0.75))                                           # 42.5 ms after the release
            o = p.render(length)
            outs.append(o.copy()); sums.append(p.voice_sum(length).copy())
            if cs:
                for k in range(n):
                    co, cv, _, _ = cs[k].render_taps(length)
                    rep = oracle.parity_report(o[k], co, abs_floor=oracle.ABS_FLOOR_DENSE)
                    assert rep["n_bad"] == 0, (k, b, length, rep)
                    vs = sums[-1][k]
                    assert np.max(np.abs(vs - cv)) <= 2e-9 * max(1.0, float(np.max(np.abs(cv)))), (k, b, float(np.max(np.abs(vs - cv))))
        for k in range(n):
            assert p[k].active_voice_count() == 64
        p.close()
        if cs:
            for c in cs:
                c.close()
        res[steal] = (np.concatenate(outs, axis=1), np.concatenate(sums, axis=1))
    scale = max(1.0, float(np.max(np.abs(res[0][1]))))
    d = np.abs(res[1][1] - res[0][1])
    assert np.max(d) <= 1e-10 * scale, float(np.max(d))
    assert np.array_equal(res[1][1][2, :512], res[0][1][2, :512])              # engine 2's crossfade stayed on the general kernel
    assert np.max(d[0]) > 0.0                                                  # ... and engine 0's did not
    assert np.max(np.abs(res[1][0] - res[0][0])) <= 1e-6
    assert np.max(np.abs(res[1][0])) > 1e-3


def test_release_variant_follows_the_oracle_and_the_general_kernel(hiplib, oracle):
    """k_voice_steady<false, 3> (ow_kernels.h): the slot voices of engines with a released key that still sounds render on the steady loop
    with the damper step beside it (ramp, then multipliers) instead of on the general kernel; each packed block decides by itself
    (voice_steal_takes: a voice inside an onset ramp or noise burst anywhere in it leaves the block to k_voice).  Engines: 0 a chord, some
    keys released; 1 the whole keyboard released at once (bass ramps of 50 ms, treble of 8 ms, the top keys without a damper); 2 pedal
    down, keys up (Sustained: nothing damps), pedal up (everything damps at once); 3 released and re-struck at once (damping steal voices
    beside fresh onsets: general, then the variants take over); 4 a key released inside its own onset ramp; 5 sparse: one held note, one
    released (shares its packed block with its neighbours).  Ragged block lengths incl. some below the chunk; long enough for voices to
    be freed at -80 dB (slot states follow the oracle).  Every block follows the oracle; against OW_VOICE_RELEASE=0 voice sums differ by
    rounding only, and the two runs free their voices in the same blocks."""
    import openwurli_amd as ow
    sr, n = 48000.0, 6
    lengths = (512, 512, 512, 512, 512, 512, 100, 7, 133, 24, 25, 512, 1, 512, 300) + (512,) * 30
    res = {}
    keys = list(range(33, 97))
    for release in (1, 0):
        p = ow.EnginePool(sr, n)
        p.set_switch("voice_release", release)
        assert p.get_switch("voice_release") == release
        p.set_sample_rate(sr)
        cs = [oracle.OracleEngine(sr) for _ in range(n)] if release else None
        if cs:
            for c in cs:
                c.set_sample_rate(sr)
        def both(k, f):
            f(p[k])
            if cs:
                f(cs[k])
        sums, outs, counts = [], [], []
        both(0, lambda e: [e.note_on(k, 0.7) for k in (40, 47, 52, 60, 64, 67, 72, 88)])
        both(1, lambda e: [e.note_on(k, 0.5 + 0.004 * (k - 33)) for k in keys])
        both(2, lambda e: (e.set_sustain(True), [e.note_on(k, 0.6) for k in (36, 48, 55, 62, 70, 81)]))
        both(3, lambda e: [e.note_on(k, 0.65) for k in keys[::2]])
        both(5, lambda e: (e.note_on(45, 0.8), e.note_on(69, 0.8)))
        for b, length in enumerate(lengths):
            if b == 6:
                both(0, lambda e: [e.note_off(k) for k in (40, 52, 64, 88)])
                both(1, lambda e: [e.note_off(k) for k in keys])
                both(2, lambda e: [e.note_off(k) for k in (36, 48, 55, 62, 70, 81)])
                both(3, lambda e: [(e.note_off(k), e.note_on(k, 0.8)) for k in keys[::2]])
                both(4, lambda e: e.note_on(38, 0.3))                           # a slow onset (bass, soft) ...
                both(5, lambda e: e.note_off(69))
            if b == 7:
                both(4, lambda e: e.note_off(38))                               # ... released 100 samples into it
            if b == 12:
                both(2, lambda e: e.set_sustain(False))
            o = p.render(length)
            outs.append(o.copy()); sums.append(p.voice_sum(length).copy())
            counts.append([p[k].active_voice_count() for k in range(n)])
            if cs:
                for k in range(n):
                    co, cv, _, _ = cs[k].render_taps(length)
                    rep = oracle.parity_report(o[k], co, abs_floor=oracle.ABS_FLOOR_DENSE)
                    assert rep["n_bad"] == 0, (k, b, length, rep)
                    vs = sums[-1][k]
                    assert np.max(np.abs(vs - cv)) <= 2e-9 * max(1.0, float(np.max(np.abs(cv)))), (k, b, float(np.max(np.abs(vs - cv))))
                    assert counts[-1][k] == cs[k].active_voice_count(), (k, b, counts[-1][k], cs[k].active_voice_count())
        p.close()
        if cs:
            for c in cs:
                c.close()
        res[release] = (np.concatenate(outs, axis=1), np.concatenate(sums, axis=1), counts)
    assert res[1][2] == res[0][2]
    assert res[1][2][-1][1] <= 8 and res[1][2][6][1] == 64                      # the released keyboard has been freed (all but the undamped top keys and the slowest bass)
    scale = max(1.0, float(np.max(np.abs(res[0][1]))))
    d = np.abs(res[1][1] - res[0][1])
    assert np.max(d) <= 1e-10 * scale, float(np.max(d))
    assert np.max(d[1]) > 0.0                                                  # engine 1's release did run on the variant
    assert np.max(np.abs(res[1][0] - res[0][0])) <= 1e-6
    assert np.max(np.abs(res[1][0])) > 1e-3
