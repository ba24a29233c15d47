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
