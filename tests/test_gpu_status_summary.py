"""Status summary of a block (openwurli_hip.hip `k_eout_attention`): big ranges copy one bit per engine to the host instead of every
engine's status block, and fetch the blocks only when a bit is set.  That must be invisible: the same script -- strikes, releases, steals
of sounding keys, top-octave voices that fall silent and are freed, a reset, a re-rate, one engine rendered on its own, an output-NaN
guard -- through the summary path (forced on a small pool) and through the status-block path gives the same samples, the same voice
bookkeeping and the same diagnostics; and a steady block of an untouched pool skips the per-engine host scans without changing anything."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _script(ow, sr, n, attn):
    p = ow.EnginePool(sr, n)
    p.set_switch("eout_attn", attn)
    assert p.get_switch("eout_attn") == attn
    p.set_sample_rate(sr)
    outs, diag = [], []
    for k in range(n):
        p[k].set_volume(0.3 + 0.01 * (k % 11)); p[k].set_tremolo_depth(0.1 * (k % 10))
        for m in (40 + k % 30, 60 + k % 20, 96 - k % 5):
            p[k].note_on(m, 0.4 + 0.05 * (k % 10))
    for b in range(40):
        if b == 3:
            for k in range(0, n, 2):
                p[k].note_off(60 + k % 20)                 # damper phases
        if b == 5:
            for k in range(0, n, 3):
                p[k].note_on(40 + k % 30, 0.9)             # re-strike of a sounding key: a steal fade
        if b == 9:
            p[7].reset(); p[7].note_on(72, 0.8)
        if b == 12:
            p[5].warm_up()                                 # one engine on its own: the status-block path inside a summary pool
        if b == 15:
            p[11].set_volume(1e300); p[11].note_on(50, 1.0)   # unbounded gain: output NaN guard (engine.rs:450-458)
        if b == 17:
            p[11].set_volume(0.4)
        if b == 20:
            for k in range(n):
                p[k].note_off(96 - k % 5); p[k].note_off(40 + k % 30)
        length = (512, 300, 64, 512, 1, 777)[b % 6]
        outs.append(p.render(length).copy())
        if b % 4 == 0:
            diag.append([(p[k].active_voice_count(), p[k].diag().output_nan_resets, p[k].diag().nan_guard_fires) for k in (0, 5, 7, 11, n - 1)])
    p.close()
    return np.concatenate(outs, axis=1), diag


@pytest.mark.parametrize("sr", [48000.0])
def test_status_summary_is_invisible(hiplib, sr):
    import openwurli_amd as ow
    n = 130                                                # three summary words, a ragged last one
    a, da = _script(ow, sr, n, 1)
    b, db = _script(ow, sr, n, 0)
    assert np.array_equal(a, b)
    assert da == db
    assert np.max(np.abs(a)) > 1e-3
    assert any(x[3][1] > 0 for x in da)                    # the output guard did fire on engine 11
    assert da[-1][2][0] == 1 and da[0][2][0] == 3          # engine 7 was reset and re-struck with one key


def test_steady_blocks_of_an_untouched_pool(hiplib, oracle):
    """No API call between blocks: the render skips its per-engine scans (dirty_any) and the summary has no bit set.  Same samples as an
    oracle engine, and a touch of ONE engine afterwards is seen."""
    import openwurli_amd as ow
    sr, n = 48000.0, 192
    p = ow.EnginePool(sr, n)
    p.set_switch("eout_attn", 1)
    p.set_sample_rate(sr)
    c = oracle.OracleEngine(sr); c.set_sample_rate(sr)
    for e in [p[k] for k in range(n)] + [c]:
        e.note_on(57, 0.8); e.note_on(64, 0.7)
    for b in range(12):
        if b == 8:
            p[100].note_on(72, 0.9); c.note_on(72, 0.9)
        out = p.render(512)
        co = c.render(512)
        rep = oracle.parity_report(out[100], co, abs_floor=oracle.ABS_FLOOR_OUTPUT)
        assert rep["n_bad"] == 0, (b, rep)
        if b < 8:
            assert np.array_equal(out[0], out[100])
        elif b > 8:
            assert not np.array_equal(out[0], out[100])
    assert p[100].active_voice_count() == 3 and p[0].active_voice_count() == 2
    p.close(); c.close()
