"""The reference's own engine tests (crates/openwurli-dsp/src/engine.rs:682-1179), restated against the HIP engine through
the C-ABI mirror -- same names, same stimuli, same thresholds.  State-machine-only tests also run on the GPU box because
every engine owns device state."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
MAX_VOICES = 64


@pytest.fixture
def engine(hiplib):
    import openwurli_amd as ow
    e = ow.WurliEngine(44100.0)
    yield e
    e.close()


def _state(e, st):
    return e.count_voices_in_state(st)


def test_note_on_allocates_voice(engine):
    engine.note_on(60, 0.8)
    assert engine.held_voice_count() == 1


def test_note_off_releases_voice(engine):
    engine.note_on(60, 0.8); engine.note_off(60)
    assert engine.held_voice_count() == 0


def test_polyphony_up_to_max_voices(engine):
    for n in range(MAX_VOICES):
        engine.note_on(36 + n, 0.8)
    assert engine.held_voice_count() == MAX_VOICES


def test_voice_stealing_when_full(engine):
    for n in range(MAX_VOICES):
        engine.note_on(36 + n, 0.8)
    engine.note_on(96, 0.8)
    assert engine.held_voice_count() == MAX_VOICES
    assert engine.has_steal_voice_for(96)


def test_render_produces_output(engine):
    engine.note_on(60, 0.8)
    buf = engine.render(256)
    assert float(np.sum(buf.astype(np.float64) ** 2)) > 0.0


def test_render_no_notes_is_near_silent(engine):
    assert float(np.max(np.abs(engine.render(512)))) < 0.05


def test_reset_clears_voices(engine):
    engine.note_on(60, 0.8); engine.note_on(72, 0.8)
    engine.reset()
    assert engine.active_voice_count() == 0


def test_sustain_defers_note_off(engine):
    engine.set_sustain(True); engine.note_on(60, 0.8); engine.note_off(60)
    assert engine.sustained_voice_count() == 1 and engine.held_voice_count() == 0
    engine.set_sustain(False)
    assert engine.sustained_voice_count() == 0


def test_volume_smoother_ramps(engine):
    """engine.rs:776-785 reads volume.current; through the ABI the ramp is observed on the output instead: the block after
    set_volume(1.0) must end at twice the level of a block at the default 0.5 (5 ms ramp = 220 samples at 44.1 kHz)."""
    import openwurli_amd as ow
    a, b = engine, ow.WurliEngine(44100.0)
    for e in (a, b):
        e.set_tremolo_depth(0.0)
        e.render(4096)
        e.note_on(60, 0.9)
    a.set_volume(1.0)
    xa, xb = a.render(1024).astype(np.float64), b.render(1024).astype(np.float64)
    b.close()
    assert abs(xa[1] / xb[1] - (0.5 + 2 * 0.5 / 220) / 0.5) < 1e-3 or abs(xb[1]) < 1e-9      # 2 steps into the ramp
    assert np.allclose(xa[300:], 2.0 * xb[300:], rtol=1e-5, atol=1e-9)


def _chord_render(vol, trem, notes, vel, seconds, sr=44100.0):
    import openwurli_amd as ow
    e = ow.WurliEngine(sr)
    e.ensure_buffer_capacity(1024)
    e.set_volume(vol); e.set_tremolo_depth(trem); e.set_speaker_character(0.0); e.set_mlp_enabled(True); e.set_noise_enabled(False)
    for _ in range(6):
        e.render(1024)
    for n in notes:
        e.note_on(n, vel)
    total = int(sr * seconds)
    out = np.concatenate([e.render(min(1024, total - p)) for p in range(0, total, 1024)])
    e.close()
    return out


def test_engine_peak_below_unity_at_vol_1(hiplib):
    out = _chord_render(1.0, 1.0, (48, 55, 60, 63, 67, 70), 0.95, 1.0)
    assert float(np.max(np.abs(out))) <= 1.02


def test_user_volume_scales_output_linearly(hiplib):
    p05 = float(np.max(np.abs(_chord_render(0.5, 0.0, (60,), 0.95, 0.5))))
    p10 = float(np.max(np.abs(_chord_render(1.0, 0.0, (60,), 0.95, 0.5))))
    assert 1.96 <= p10 / p05 <= 2.04


def test_higher_velocity_louder(engine):
    engine.set_volume(0.5)
    engine.note_on(60, 0.2)
    soft = engine.render(4096).astype(np.float64)
    engine.reset()
    engine.note_on(60, 1.0)
    loud = engine.render(4096).astype(np.float64)
    assert np.sqrt(np.mean(loud ** 2)) > np.sqrt(np.mean(soft ** 2))


def test_note_clamps_to_valid_range(engine):
    engine.note_on(0, 0.8); engine.note_on(127, 0.8)
    assert engine.held_voice_count() == 2


def test_sustain_pedal_release_triggers_damping(engine):
    engine.set_sustain(True); engine.note_on(60, 0.8); engine.note_off(60)
    assert engine.sustained_voice_count() == 1
    engine.set_sustain(False)
    assert engine.sustained_voice_count() == 0 and _state(engine, 3) == 1


def test_sustain_held_voices_still_render(engine):
    engine.set_sustain(True); engine.note_on(60, 0.8)
    engine.render(1024)
    engine.note_off(60)
    engine.render(1024)
    engine.set_sustain(False)
    buf = engine.render(1024)
    assert float(np.sum(buf.astype(np.float64) ** 2)) > 0.0


def test_no_sustain_normal_note_off(engine):
    engine.note_on(60, 0.8); engine.note_off(60)
    assert engine.held_voice_count() == 0 and _state(engine, 3) == 1


def test_voice_stealing_prefers_sustained_over_held(engine):
    engine.set_sustain(True)
    for n in range(MAX_VOICES // 2):
        engine.note_on(36 + n, 0.8); engine.note_off(36 + n)
    for n in range(MAX_VOICES // 2, MAX_VOICES):
        engine.note_on(36 + n, 0.8)
    s0, h0 = engine.sustained_voice_count(), engine.held_voice_count()
    assert s0 + h0 == MAX_VOICES
    engine.note_on(127, 0.8)
    assert engine.held_voice_count() == h0 + 1 and engine.sustained_voice_count() == s0 - 1


def test_reattack_releases_sustained_same_note(engine):
    engine.set_sustain(True); engine.note_on(60, 0.8); engine.note_off(60); engine.note_on(60, 0.8)
    assert engine.count_voices_with_note_in_state(60, 2) == 0
    assert engine.count_voices_with_note_in_state(60, 1) == 1


def test_pedal_up_only_releases_sustained_not_held(engine):
    engine.set_sustain(True); engine.note_on(60, 0.8); engine.note_off(60); engine.note_on(64, 0.8)
    assert engine.sustained_voice_count() == 1 and engine.held_voice_count() == 1
    engine.set_sustain(False)
    assert engine.sustained_voice_count() == 0 and engine.held_voice_count() == 1


def test_reset_clears_sustain_state(engine):
    engine.set_sustain(True); engine.note_on(60, 0.8); engine.note_off(60)
    engine.reset()
    assert not engine.is_sustain_held() and engine.active_voice_count() == 0


def test_note_off_for_nonexistent_note_is_noop(engine):
    engine.note_on(60, 0.8); engine.note_off(72)
    assert engine.held_voice_count() == 1


def test_volume_zero_and_back_no_nan(engine):
    engine.note_on(60, 0.8)
    for _ in range(4):
        engine.set_volume(0.0); engine.render(512)
        engine.set_volume(0.5); buf = engine.render(512)
    assert np.all(np.isfinite(buf))


def test_no_catastrophic_output_spikes_under_continuous_play(engine):
    chords = [(60, 64, 67), (62, 65, 69), (64, 67, 71), (65, 69, 72)]
    peak = 0.0
    for i in range(8):
        chord = chords[i % 4]
        for n in chord:
            engine.note_on(n, 1.0)
        for _ in range(86):
            peak = max(peak, float(np.max(np.abs(engine.render(256)))))
        for n in chord:
            engine.note_off(n)
        if i % 2 == 1:
            for _ in range(5):
                engine.render(256)
    assert 20 * np.log10(max(peak, 1e-12)) < 14.0
    assert engine.nan_guard_fires() == 0


def test_sound_after_sample_rate_change(engine):
    engine.set_sample_rate(48000.0)
    engine.note_on(60, 0.8)
    assert float(np.sum(engine.render(1024).astype(np.float64) ** 2)) > 0.0


def test_buffer_capacity_grows_to_render(engine):
    engine.note_on(60, 0.8)
    big = engine.render(16384)                         # > MAX_BLOCK_SIZE default
    assert np.all(np.isfinite(big)) and float(np.max(np.abs(big))) > 0.0


def test_tremolo_smoother_does_not_pin_depth_to_zero(engine):
    engine.note_on(60, 0.9)
    sr = 44100
    x = np.concatenate([engine.render(256) for _ in range(4 * sr // 256)]).astype(np.float64)
    win = sr // 50
    env = [20 * np.log10(np.sqrt(np.mean(x[i * win:(i + 1) * win] ** 2)) + 1e-12) for i in range(25, x.size // win)]
    assert max(env) - min(env) > 3.0
