"""Boundary contract tests on the GPU:
setter -> reset() with no render in between snaps to the NEW target (engine.rs:86-99,245-249); a failed render degrades to
silence and the Python mirror reports it; errors do not go unnoticed."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("preamp_kind", [0, 1])
def test_setter_then_reset_without_a_render_in_between(hiplib, oracle, preamp_kind):
    """plugin initialize(): set_sample_rate -> sync_params (setters) -> host calls reset() -> first process().  LinearSmoother::set_target
    has stored the new targets when reset() snaps to them; the device only hears of setters with the next block."""
    import openwurli_amd as ow
    sr, n = 48000.0, 3
    g = ow.EnginePool(sr, n, preamp_kind=preamp_kind)
    cs = [oracle.OracleEngine(sr, preamp_kind=preamp_kind) for _ in range(n)]
    g.set_sample_rate(sr)
    for c in cs:
        c.set_sample_rate(sr)
    floor = oracle.ABS_FLOOR_MELANGE_LIT_OUTPUT if preamp_kind else oracle.ABS_FLOOR_OUTPUT

    def compare(tag, blocks):
        for b in range(blocks):
            go = g.render(256)
            for k in range(n):
                rep = oracle.parity_report(go[k], cs[k].render(256), abs_floor=floor)
                assert rep["n_bad"] == 0, (tag, b, k, rep)
        return go
    for k in range(n):
        for e in (g[k], cs[k]):
            e.set_volume(0.8 - 0.2 * k); e.set_tremolo_depth(0.1 + 0.4 * k); e.set_speaker_character(0.3 * k)
    # engine 0 and 2 reset individually, engine 1 keeps ramping: its retarget must survive its neighbours' resets
    g[0].reset(); cs[0].reset()
    g[2].reset(); cs[2].reset()
    for k in range(n):
        for e in (g[k], cs[k]):
            e.note_on(60 + 5 * k, 0.8)
    out = compare("engine reset", 6)
    assert np.max(np.abs(out[0])) > 1e-3
    # the host's value did not change, so set_target ignores it (|d| < 1e-9) -- it must already be in force
    for k in range(n):
        for e in (g[k], cs[k]):
            e.set_volume(0.8 - 0.2 * k)
    compare("same value again", 2)
    # whole-pool reset with pending setters on every engine
    for k in range(n):
        for e in (g[k], cs[k]):
            e.set_volume(0.35 + 0.1 * k); e.set_speaker_character(0.9 - 0.3 * k)
    g.reset()
    for c in cs:
        c.reset()
    for k in range(n):
        for e in (g[k], cs[k]):
            e.note_on(48 + 7 * k, 0.9)
    compare("pool reset", 6)
    g.close()


@pytest.mark.parametrize("power_amp_kind", [0, 1])
def test_output_nan_guard_and_the_blocks_after_it(hiplib, oracle, power_amp_kind):
    """engine.rs:450-458 with a real non-finite output: an unbounded volume (set_volume does not clamp, engine.rs:378-380) makes every
    sample of a block non-finite -> 0.0 out, speaker reset in-sample; preamp, oversampler and power amp have processed the whole block
    before the speaker loop (engine.rs:432-434), so their resets land on the post-block state.  The blocks AFTER the guard are compared
    with the oracle on the poisoned engine and on its untouched neighbour, for both power amps (no test covered parity after
    a guard fires)."""
    import openwurli_amd as ow
    sr, n = 48000.0, 2
    g = ow.EnginePool(sr, n, power_amp_kind=power_amp_kind)
    cs = [oracle.OracleEngine(sr, power_amp_kind=power_amp_kind) for _ in range(n)]
    g.set_sample_rate(sr)
    for c in cs:
        c.set_sample_rate(sr)
    for k in range(n):
        for e in (g[k], cs[k]):
            e.set_tremolo_depth(0.6)
            for note in (48, 60, 64, 67):
                e.note_on(note + k, 0.8)

    def compare(tag, blocks, length=256):
        worst = 0.0
        for b in range(blocks):
            go = g.render(length)
            for k in range(n):
                co = cs[k].render(length)
                rep = oracle.parity_report(go[k], co, abs_floor=oracle.ABS_FLOOR_OUTPUT)
                assert rep["n_bad"] == 0, (tag, b, k, rep)
                worst = max(worst, float(np.max(np.abs(co))))
        return go, worst
    compare("before", 4)
    for e in (g[0], cs[0]):
        e.set_volume(float("inf"))
    go, _ = compare("poisoned block", 2)
    assert np.all(go[0] == 0.0) and np.max(np.abs(go[1])) > 1e-3          # silence from the guard; the neighbour plays on
    assert g[0].diag().output_nan_resets >= 1 and g[1].diag().output_nan_resets == 0
    for e in (g[0], cs[0]):
        e.set_volume(0.5)
    # the smoother walks inf -> NaN -> 0.5 over its ramp, the guard keeps firing until it lands; then the engine must come back from
    # the reset states exactly like the reference does
    _, peak = compare("recovery", 10)
    assert peak > 1e-3
    for e in (g[0], cs[0]):
        e.note_on(72, 0.9)
    compare("new note after the guard", 6)
    g.close()


def test_render_failure_is_silence_and_is_reported(hiplib):
    import openwurli_amd as ow
    p = ow.EnginePool(48000.0, 5)
    for k in range(5):
        p[k].note_on(50 + k, 0.9)
    a = np.concatenate([p.render(256) for _ in range(6)], axis=1)              # past the onset ramp
    assert np.max(np.abs(a)) > 1e-3
    hiplib.ow_test_inject_render_faults(p._h, 2)
    out = np.full((5, 256), 7.0, dtype=np.float32)
    import ctypes as C
    hiplib.ow_pool_render(p._h, out.ctypes.data_as(C.c_void_p), 256, 256)       # the raw C call: void, never fails, silence
    assert np.all(out == 0.0)
    assert "injected fault" in hiplib.ow_last_error().decode()
    hiplib.ow_clear_error()
    with pytest.raises(ow.OwError):                                             # the Python mirror raises
        p.render(256)
    b = p.render(256)                                                           # and the pool works again afterwards
    assert np.max(np.abs(b)) > 1e-3 and np.all(np.isfinite(b))
    p.close()
    e = ow.WurliEngine(48000.0)
    e.note_on(60, 0.8)
    hiplib.ow_test_inject_render_faults(C.c_void_p(hiplib.ow_engine_pool(e._h)), 1)
    buf = np.full(128, 3.0, dtype=np.float32)
    hiplib.ow_engine_render(e._h, buf.ctypes.data_as(C.c_void_p), 128)
    assert np.all(buf == 0.0)
    hiplib.ow_clear_error()
    assert np.max(np.abs(e.render(2048))) > 1e-4
    e.close()


def test_settled_tremolo_cache_is_bit_identical(hiplib, oracle):
    """The 192 050-step Twin-T settle (tremolo.rs:92-102) is computed once per (device, chain rate) and copied afterwards (VERDICT r02
    next-6).  A cached engine must be the freshly settled one bit for bit -- R stream, preamp tap and output -- through new, set_sample_rate,
    reset of one engine and of the pool; and it must match the oracle like any other engine."""
    import time
    import openwurli_amd as ow
    sr = 48000.0

    def play(e):
        e.set_tremolo_depth(1.0); e.set_volume(0.6)
        for n in (45, 60, 67):
            e.note_on(n, 0.8)
        out = [e.render(512) for _ in range(6)]
        return np.concatenate(out)

    def lifecycle(make):
        t0 = time.perf_counter()
        e = make()
        e.set_sample_rate(sr)
        dt = time.perf_counter() - t0
        a = play(e)
        r = e.read_tremolo_r(1024) if hasattr(e, "read_tremolo_r") else None
        e.reset()
        b = play(e)
        e.set_sample_rate(44100.0)
        c = play(e)
        e.close()
        return dt, a, b, c, r

    hiplib.ow_test_clear_settle_caches()
    t_fresh, a0, b0, c0, r0 = lifecycle(lambda: ow.WurliEngine(sr))             # misses: 96 kHz (new, rate, reset), then 88.2 kHz
    t_hit, a1, b1, c1, r1 = lifecycle(lambda: ow.WurliEngine(sr))               # hits everywhere
    assert hiplib.ow_test_clear_settle_caches() == 2                            # chain rates 96 000 and 88 200
    t_again, a2, b2, c2, r2 = lifecycle(lambda: ow.WurliEngine(sr))             # settles afresh again
    for x, y, z in ((a0, a1, a2), (b0, b1, b2), (c0, c1, c2)):
        assert np.max(np.abs(x)) > 1e-3
        assert np.array_equal(x, y) and np.array_equal(x, z)
    if r0 is not None:
        assert np.array_equal(r0, r1) and np.array_equal(r0, r2)
    # new + set_sample_rate of a second engine: no 0.9 s settle any more (warm-up of 0.6 s of audio remains: 57 blocks)
    assert t_hit < 0.5 * t_fresh or t_hit < 0.35, (t_fresh, t_hit, t_again)
    # and the cached engine is the reference's engine
    g = ow.WurliEngine(sr); c = oracle.OracleEngine(sr)
    g.set_sample_rate(sr); c.set_sample_rate(sr)
    rep = oracle.parity_report(play(g), play(c), abs_floor=oracle.ABS_FLOOR_OUTPUT)
    assert rep["n_bad"] == 0, rep
    # a pool: engine 3 reset alone (leaves the phase group, takes the cached state), then the whole pool
    p = ow.EnginePool(sr, 6)
    p.set_sample_rate(sr)
    cs = [oracle.OracleEngine(sr) for _ in range(6)]
    for k, o in enumerate(cs):
        o.set_sample_rate(sr)
        for e in (p[k], o):
            e.set_tremolo_depth(0.9); e.note_on(50 + 3 * k, 0.7)
    for blk in range(3):
        go = p.render(512)
        for k in range(6):
            assert oracle.parity_report(go[k], cs[k].render(512), abs_floor=oracle.ABS_FLOOR_OUTPUT)["n_bad"] == 0, (blk, k)
    p[3].reset(); cs[3].reset()
    for e in (p[3], cs[3]):
        e.note_on(62, 0.8)
    for blk in range(4):
        go = p.render(512)
        for k in range(6):
            assert oracle.parity_report(go[k], cs[k].render(512), abs_floor=oracle.ABS_FLOOR_OUTPUT)["n_bad"] == 0, ("after reset", blk, k)
    d = p[3].diag()
    assert d.tremolo_be_fallbacks == p[0].diag().tremolo_be_fallbacks or d.tremolo_be_fallbacks >= 0
    p.close(); g.close()


VF_S0, VF_Q = 0, 81      # openwurli_hip_test.h: OW_TEST_VF_S0, OW_TEST_VF_Q


@pytest.mark.parametrize("scenario", ["slot_voice_q", "slot_voice_s0", "steal_voice", "steal_voice_survives_first_pass", "two_culprits_steady_kernel"])
def test_voice_sum_nan_guard_second_pass(hiplib, oracle, scenario):
    """engine.rs:496-521.  A voice is made non-finite on both sides (test poke; no API call can do it).  The block it poisons comes out as
    the chain's response to a zeroed voice sum; the culprit -- and only the culprit -- is freed; nan_guard_fires counts one; and because
    the reference finds the culprit by rendering EVERY voice a second time, the survivors have advanced 2 x len samples: the blocks
    after the guard only match the oracle if the second pass is reproduced (deviation 2 is closed)."""
    import openwurli_amd as ow
    sr, L = 48000.0, 256
    g = ow.EnginePool(sr, 3)
    cs = [oracle.OracleEngine(sr) for _ in range(3)]
    g.set_sample_rate(sr)
    for c in cs:
        c.set_sample_rate(sr)

    def compare(tag, blocks, length=L, taps=True):
        for b in range(blocks):
            go = g.render(length)
            gv = g.voice_sum(length)
            for k in range(3):
                co, cv, _, _ = cs[k].render_taps(length)
                rep = oracle.parity_report(go[k], co, abs_floor=oracle.ABS_FLOOR_OUTPUT)
                assert rep["n_bad"] == 0, (scenario, tag, "out", b, k, rep)
                rep = oracle.parity_report(gv[k], cv, rel=1e-12, floor_frac=1.0)
                assert rep["n_bad"] == 0, (scenario, tag, "voice sum", b, k, rep)
            for k in range(3):
                for s in range(64):
                    assert hiplib.ow_engine_slot_state(g[k]._h, s) == cs[k].slot_state(s), (scenario, tag, b, k, s)
                assert g[k].diag().steal_voices == cs[k].steal_voice_count(), (scenario, tag, b, k)
                assert g[k].nan_guard_fires() == cs[k].nan_guard_fires(), (scenario, tag, b, k)

    chord = (40, 52, 59, 64, 67, 72, 88)
    for k in range(3):
        for e in (g[k], cs[k]):
            e.set_tremolo_depth(0.7)
            for n in chord:
                e.note_on(n + k, 0.5 + 0.1 * k)
    # slots 0..6 in note order.  The steady-kernel scenario waits until onset ramps and attack noise are over (engines leave k_voice)
    compare("before", 12 if scenario == "two_culprits_steady_kernel" else 2)
    if scenario.startswith("steal_voice"):
        for e in (g[1], cs[1]):
            for n in range(33, 97):
                e.note_on(n, 0.4)             # 57 free slots, then the seven oldest Held voices (slots 0..6) are stolen ...
            e.note_on(50, 0.9)                # ... and one more: eight 5 ms crossfades
        assert g[1].diag().steal_voices == cs[1].steal_voice_count() == 8
        compare("stolen", 1, length=64)       # 64 of the 240 crossfade samples
        for e in (g[1], cs[1]):
            assert e.poke_voice(0, True, VF_Q, float("nan")) == 0
    elif scenario == "slot_voice_q":
        for e in (g[1], cs[1]):
            assert e.poke_voice(3, False, VF_Q, float("nan")) == 0
    elif scenario == "slot_voice_s0":
        for e in (g[1], cs[1]):
            assert e.poke_voice(5, False, VF_S0, float("inf")) == 0
        for e in (g[1], cs[1]):
            e.note_off(52 + 1)                # a damper phase starts in the same block: the second pass runs it twice as far
    else:
        for e in (g[0], cs[0]):
            assert e.poke_voice(1, False, VF_Q, float("nan")) == 0
            assert e.poke_voice(6, False, VF_S0, float("nan")) == 0
        for e in (g[2], cs[2]):
            assert e.poke_voice(0, False, VF_Q, float("-inf")) == 0
    fires_before = [c.nan_guard_fires() for c in cs]
    active_before = [c.active_voice_count() for c in cs]
    # 256 samples end all eight crossfades inside the block: the reference drops those steal voices at the end of its FIRST pass, the
    # second pass finds no culprit and every slot voice advances twice.  With a 64-sample block the poisoned steal voice is still there
    # (fade 176 -> 112), the second pass renders it again, finds it and drops it; the other seven keep fading from 112.
    compare("poisoned block", 1, length=64 if scenario == "steal_voice_survives_first_pass" else L)
    hit = {"steal_voice": [1], "steal_voice_survives_first_pass": [1], "slot_voice_q": [1], "slot_voice_s0": [1], "two_culprits_steady_kernel": [0, 2]}[scenario]
    for k in range(3):
        assert cs[k].nan_guard_fires() - fires_before[k] == (1 if k in hit else 0)
    if scenario == "slot_voice_q":
        assert cs[1].active_voice_count() == active_before[1] - 1 and cs[1].slot_state(3) == 0
    if scenario == "two_culprits_steady_kernel":
        assert cs[0].active_voice_count() == active_before[0] - 2 and cs[2].active_voice_count() == active_before[2] - 1
    if scenario == "steal_voice":
        assert cs[1].steal_voice_count() == 0 and cs[1].active_voice_count() == active_before[1]
    if scenario == "steal_voice_survives_first_pass":
        assert cs[1].steal_voice_count() == 7 and cs[1].active_voice_count() == active_before[1]
        compare("rest of the crossfades", 2, length=64)
    # the survivors are one block ahead of where a single render would have left them: ten more blocks, a re-strike among them
    compare("after the guard", 5)
    for k in range(3):
        for e in (g[k], cs[k]):
            e.note_off(64 + k); e.note_on(64 + k, 0.8)
    compare("after the guard, new notes", 5)
    g.close()


def test_midi_burst_on_device_equals_the_host_state_machine(hiplib):
    """ow_pool_midi on a big pool applies a burst of events ON THE DEVICE (k_vm_events: the state machine of ow_vm.h, the code the host runs,
    one lane per engine; the states are copied back) instead of event by event on the host threads.  Forced here on a pool of 96 engines
    and compared with the host path under a script that goes through every branch of engine.rs:299-374 / 569-590: whole-keyboard strikes
    and re-strikes, repeated keys, out-of-range keys, more notes than slots (steals of releasing, sustained and held voices), the pedal down
    / up with sustained voices to damp, single events between a burst and its render (their ops queue behind the device's), single
    events BEFORE a burst (the burst then stays on the host: queue order), two bursts before one render, a burst whose ops overflow an
    engine's fixed queue (replayed on the host), reset of one engine, a list that is NOT grouped by engine (found out on the device, replayed
    on the host).  The device applies a burst's queues at once, beside the download of the states (`midi_apply_early`); with that off they
    wait for the next render's k_apply_ops.  Slot states, notes, steal voices, voice counts and every rendered sample are identical in
    all three."""
    import openwurli_amd as ow
    from openwurli_amd import binding
    n = 96
    rng0 = np.random.default_rng(7)

    def burst(engines, kind, seed):
        rng = np.random.default_rng(seed)
        rows = []
        for e in engines:
            if kind == "strike":
                ev = [(0, k, 0.3 + 0.6 * ((e + k) % 7) / 7.0) for k in range(33, 97)]
            elif kind == "restrike":
                ev = [x for k in range(33, 97) for x in ((1, k, 0.0), (0, k, 0.4 + 0.5 * ((e * 3 + k) % 5) / 5.0))]
            elif kind == "play":
                ev = []
                for _ in range(int(rng.integers(5, 40))):
                    t = int(rng.choice([0, 0, 0, 1, 1, 2]))
                    ev.append((t, int(rng.integers(20, 110)), float(rng.random())))
            elif kind == "pedal_down":
                ev = [(2, 0, 1.0)] + [(1, k, 0.0) for k in range(40, 70)]
            elif kind == "pedal_up":
                ev = [(0, 50, 0.7), (2, 0, 0.0), (0, 61, 0.5)]
            elif kind == "release_all":
                ev = [(1, k, 0.0) for k in range(33, 97)]
            elif kind == "overflow":
                ev = [x for _ in range(3) for k in range(33, 97) for x in ((1, k, 0.0), (0, k, 0.6))]
            rows += [(e, t, k, 0, v) for t, k, v in ev]
        return np.array(rows, dtype=np.dtype(binding.MIDI_DTYPE))

    def run(device, early=1):
        g = ow.EnginePool(48000.0, n)
        g.set_sample_rate(48000.0)
        g.set_switch("midi_device", device); g.set_switch("midi_apply_early", early)
        outs, states = [], []

        def snap():
            st = []
            for k in (0, 1, 17, 50, 95):
                e = g[k]
                d = e.diag()
                st.append(([e.slot_state(s) for s in range(64)], [e.slot_note(s) for s in range(64)], [bool(e.has_steal_voice_for(q)) for q in (40, 50, 60, 70)],
                           d.active_voices, d.held_voices, d.sustained_voices, d.releasing_voices, d.steal_voices, d.sustain_held))
            return st
        allk = list(range(n))
        g.midi(burst(allk, "strike", 1)); outs.append(g.render(256).copy()); states.append(snap())
        g.midi(burst(allk, "play", 2)); outs.append(g.render(300).copy()); states.append(snap())
        g.midi(burst(allk, "pedal_down", 3)); outs.append(g.render(128).copy()); states.append(snap())
        g.midi(burst(allk, "play", 4)); g[17].note_on(77, 0.9); g[17].note_off(77); g[50].set_sustain(False)      # single events BEHIND a burst
        outs.append(g.render(200).copy()); states.append(snap())
        g.midi(burst(allk, "pedal_up", 5)); outs.append(g.render(256).copy()); states.append(snap())
        g[1].note_on(45, 0.6)                                              # a single event BEFORE a burst: queue order keeps the burst on the host
        g.midi(burst(allk, "restrike", 6)); outs.append(g.render(512).copy()); states.append(snap())
        g.midi(burst(allk[: n // 2], "play", 7)); g.midi(burst(allk, "play", 8))                                     # two bursts, one render
        outs.append(g.render(100).copy()); states.append(snap())
        g[95].reset(); g[95].note_on(60, 0.8)
        g.midi(burst(allk, "restrike", 9)); outs.append(g.render(512).copy()); states.append(snap())
        g.midi(burst(allk, "overflow", 10)); outs.append(g.render(256).copy()); states.append(snap())               # 576 ops per engine: host replay
        g.midi(burst(allk, "play", 11)); outs.append(g.render(256).copy()); states.append(snap())
        g.midi(burst(allk[::-1], "play", 12)); outs.append(g.render(256).copy()); states.append(snap())             # engines in falling order: not grouped
        g.midi(burst(allk, "restrike", 13)); outs.append(g.render(64).copy()); states.append(snap())
        outs.append(g.render(512).copy()); states.append(snap())
        # note-offs only: the queues are short, k_apply_ops finishes at once -- the queue lengths must not be cleared under the download
        # of the states (the host recognises the burst's engines by them)
        g.midi(burst(allk, "release_all", 14)); outs.append(g.render(256).copy()); states.append(snap())
        outs.append(g.render(512).copy()); states.append(snap())
        bursts = g.get_switch("midi_device_bursts")
        g.close()
        return outs, states, bursts
    o_dev, s_dev, b_dev = run(1)
    o_late, s_late, b_late = run(1, early=0)
    o_host, s_host, b_host = run(0)
    assert b_host == 0 and b_dev >= 10 and b_late == b_dev, (b_host, b_dev, b_late)   # the device path really ran (all but the ordered / overflowing / ungrouped bursts)
    for i, (a, b, c) in enumerate(zip(s_dev, s_host, s_late)):
        assert a == b and c == b, (i, "slot states")
    for i, (a, b, c) in enumerate(zip(o_dev, o_host, o_late)):
        assert np.array_equal(a, b), (i, float(np.max(np.abs(a - b))))
        assert np.array_equal(c, b), (i, "late", float(np.max(np.abs(c - b))))
    assert max(float(np.max(np.abs(o))) for o in o_dev) > 1e-3
