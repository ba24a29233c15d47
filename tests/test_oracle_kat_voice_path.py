"""The voice path's reference tests, ONE FOR ONE: every `#[test]` of tables.rs, reed.rs, pickup.rs,
mlp_correction.rs, hammer.rs, voice.rs and variation.rs restated against the CPU oracle with the literals and tolerances the reference
test itself states.  (tests/test_oracle_kat.py holds condensed versions of some of them, often tighter; this file is the audit trail:
test name = reference test name, citation = file:line under /root/reference/crates/openwurli-dsp/src/.)  The three intermod-risk tests
of tables.rs (:898-978) belong to the report block :675-801, which is outside the hot path.  CPU only.

What these pin and what they do not: they are the reference's OWN acceptance bands for the voice path (most are inequalities, a few are
literals: reed lengths, blank dimensions, ds(C4), onset times, the jitter / variation ranges, exact determinism).  They cannot pin
samples -- the reference holds no sample vectors for this path -- so Voice::note_on's parameter plumbing, ModalReed::render
and Pickup::process stay pinned by these bands, by f0 of the golden spectral JSON (detune + MLP frequency, 0.05 Hz) and by line-by-line
correspondence; DESIGN.md section 2 lists which functions are band-pinned only."""
import ctypes as C

import numpy as np

RATIOS = np.array([1.0, 6.267, 17.547, 34.386, 56.842, 85.1, 119.3])      # the literal the reed.rs / hammer.rs tests use


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def d(x):
    return C.c_double(x)


def _reed(L, f0, amps, decays, onset, vel, sr, seed, n):
    out = np.zeros(n)
    L.owo_reed_render(d(f0), _p(RATIOS.copy()), _p(np.asarray(amps, dtype=float)), _p(np.asarray(decays, dtype=float)), d(onset), d(vel), d(sr),
                      C.c_uint(seed), _p(out), C.c_size_t(n))
    return out


def _amps(*nz):
    a = np.zeros(7)
    for i, v in nz:
        a[i] = v
    return a


# ---------------------------------------------------------------------------------------------------- tables.rs
def test_tables_midi_to_freq(oracle):                       # tables.rs:837-841
    L = oracle.lib()
    assert abs(L.owo_midi_to_freq(69) - 440.0) < 0.01
    assert abs(L.owo_midi_to_freq(60) - 261.63) < 0.1
    assert abs(L.owo_midi_to_freq(33) - 55.0) < 0.1


def test_tables_mode_ratios_bare_beam_and_tip_mass(oracle):  # :844-856
    L = oracle.lib()
    r = np.zeros(7)
    L.owo_mode_ratios(d(0.0), _p(r))
    assert abs(r[0] - 1.0) < 1e-6 and abs(r[1] - 6.267) < 0.01 and abs(r[2] - 17.547) < 0.02
    L.owo_mode_ratios(d(0.10), _p(r))
    assert abs(r[1] - 7.13) < 0.05


def test_tables_tip_mass_ratio_range(oracle):                # :858-861
    L = oracle.lib()
    assert L.owo_tip_mass_ratio(33) > 0.05 and L.owo_tip_mass_ratio(57) < 0.02


def test_tables_decay_rate(oracle):                          # :864-895
    L = oracle.lib()
    f = L.owo_fundamental_decay_rate
    assert f(60) > f(48) and f(84) > f(72)
    assert abs(f(36) - 3.0) < 0.5
    assert 3.5 < f(60) < 7.0 and 7.0 < f(72) < 16.0 and 17.0 < f(84) < 35.0


def test_tables_reed_length_known_values(oracle):            # :981-991
    L = oracle.lib()
    assert abs(L.owo_reed_length_mm(33) - 74.93) < 0.1
    assert abs(L.owo_reed_length_mm(96) - 25.4) < 0.1
    assert abs(L.owo_reed_length_mm(52) - 50.8) < 0.1


def test_tables_mode_shape_tip_nonzero_and_clamp_zero(oracle):   # :994-1022
    L = oracle.lib()
    b = np.zeros(7)
    for mu in (0.0, 0.05, 0.10, 0.20, 0.50):
        L.owo_eigenvalues(d(mu), _p(b))
        for beta in b:
            assert abs(L.owo_mode_shape(d(beta), d(1.0))) > 0.1
    for mu in (0.0, 0.10, 0.50):
        L.owo_eigenvalues(d(mu), _p(b))
        for beta in b:
            assert abs(L.owo_mode_shape(d(beta), d(0.0))) < 1e-10


def test_tables_coupling(oracle):                            # :1025-1093 (mode 1 unity, decreasing, register variation)
    L = oracle.lib()
    k = np.zeros(7)
    for midi in range(33, 97, 4):
        L.owo_spatial_coupling(d(L.owo_tip_mass_ratio(midi)), d(L.owo_reed_length_mm(midi)), _p(k))
        assert abs(k[0] - 1.0) < 1e-10
        assert np.all(k[1:] <= k[0] + 1e-6) and k[1] < k[0]
    kb, kt = np.zeros(7), np.zeros(7)
    L.owo_spatial_coupling(d(L.owo_tip_mass_ratio(33)), d(L.owo_reed_length_mm(33)), _p(kb))
    L.owo_spatial_coupling(d(L.owo_tip_mass_ratio(96)), d(L.owo_reed_length_mm(96)), _p(kt))
    assert np.all(kt[2:] < kb[2:])


def test_tables_eigenvalues_match_mode_ratios(oracle):       # :1096-1113
    L = oracle.lib()
    b, r = np.zeros(7), np.zeros(7)
    for mu in (0.0, 0.01, 0.05, 0.10, 0.15, 0.20, 0.30, 0.50):
        L.owo_eigenvalues(d(mu), _p(b)); L.owo_mode_ratios(d(mu), _p(r))
        assert np.max(np.abs(b * b / (b[0] * b[0]) - r)) < 1e-10


def test_tables_blank_dims(oracle):                          # :1116-1154
    L = oracle.lib()
    wt = np.zeros(2)
    for midi, w, t in ((33, 0.151, 0.026), (96, 0.098, 0.034), (74, 0.121, 0.034)):
        L.owo_reed_blank_dims(midi, _p(wt))
        assert abs(wt[0] - w * 25.4) < 0.01 and abs(wt[1] - t * 25.4) < 0.01
    th = {}
    for m in (48, 53, 58):
        L.owo_reed_blank_dims(m, _p(wt)); th[m] = wt[1]
    assert abs(th[48] - 0.026 * 25.4) < 0.01 and abs(th[58] - 0.034 * 25.4) < 0.01
    assert th[48] + 0.02 < th[53] < th[58] - 0.02


def test_tables_compliance_and_displacement_scale(oracle):   # :1157-1222
    L = oracle.lib()
    cb, cm, ct = (L.owo_reed_compliance(m) for m in (33, 60, 96))
    assert cb > 5.0 * cm and cm > 2.0 * ct
    ds = L.owo_pickup_displacement_scale
    assert ds(33) >= ds(60) > ds(96)
    assert abs(ds(60) - 0.85) < 0.001
    assert ds(33) > 0.50 and ds(96) < 0.35 and ds(33) / ds(96) > 2.5


# ---------------------------------------------------------------------------------------------------- reed.rs
def test_reed_single_mode_sine_and_decay(oracle):            # reed.rs:336-381
    L = oracle.lib()
    x = _reed(L, 440.0, _amps((0, 1.0)), np.zeros(7), 0.0, 1.0, 44100.0, 12345, 44100)
    zc = int(np.sum((x[:-1] < 0) & (x[1:] >= 0)))
    assert abs(zc - 440) < 3
    n = int(44100 * 0.5)
    y = _reed(L, 440.0, _amps((0, 1.0)), np.full(7, 60.0), 0.0, 1.0, 44100.0, 12345, n + 441)
    late = float(np.max(np.abs(y[n:])))
    assert 0.01 < late < 0.1                                   # 60 dB/s -> -30 dB at 0.5 s


def test_reed_onset_ramp_shapes_attack(oracle):              # :383-422
    L = oracle.lib()
    sr, ramp = 44100.0, 0.020
    buf = _reed(L, 440.0, _amps((0, 1.0)), np.zeros(7), ramp, 1.0, sr, 42, int(sr * 0.050))
    assert abs(buf[0]) < 0.01
    mid = int(ramp * 0.5 * sr)
    assert np.max(np.abs(buf[max(mid - 5, 0):mid + 5])) < 0.8
    late = int(sr * 0.030)
    assert np.max(np.abs(buf[late:late + 200])) > 0.85


def test_reed_onset_ramp_ff_vs_pp(oracle):                   # :425-448
    L = oracle.lib()
    sr = 44100.0
    n = int(sr * 0.010)
    ff = _reed(L, 440.0, _amps((0, 1.0)), np.zeros(7), 0.001, 1.0, sr, 42, n)
    pp = _reed(L, 440.0, _amps((0, 1.0)), np.zeros(7), 0.005, 0.0, sr, 42, n)
    t2 = int(sr * 0.002)
    assert np.sum(ff[:t2] ** 2) > 1.5 * np.sum(pp[:t2] ** 2)


def test_reed_onset_zero_dwell_is_instant(oracle):           # :451-467
    L = oracle.lib()
    buf = _reed(L, 440.0, _amps((0, 1.0)), np.zeros(7), 0.0, 1.0, 44100.0, 42, 100)
    assert np.max(np.abs(buf[:10])) > 0.05


def test_reed_jitter_breaks_phase_coherence(oracle):         # :470-506
    L = oracle.lib()
    sr = 44100.0
    n = int(sr * 0.5)
    a = _reed(L, 440.0, _amps((0, 1.0), (1, 0.3)), np.zeros(7), 0.0, 1.0, sr, 100, n)
    b = _reed(L, 440.0, _amps((0, 1.0), (1, 0.3)), np.zeros(7), 0.0, 1.0, sr, 200, n)
    s0 = int(sr * 0.2)
    rel = np.sqrt(np.mean((a[s0:] - b[s0:]) ** 2)) / max(np.sqrt(np.mean(a[s0:] ** 2)), 1e-10)
    assert 0.001 < rel < 0.5


def test_reed_jitter_deterministic_and_preserves_frequency(oracle):   # :509-551
    L = oracle.lib()
    sr = 44100.0
    a = _reed(L, 440.0, _amps((0, 1.0)), np.zeros(7), 0.0, 1.0, sr, 42, int(sr * 0.2))
    b = _reed(L, 440.0, _amps((0, 1.0)), np.zeros(7), 0.0, 1.0, sr, 42, int(sr * 0.2))
    assert np.array_equal(a, b)
    x = _reed(L, 440.0, _amps((0, 1.0)), np.zeros(7), 0.0, 1.0, sr, 77, int(sr))
    zc = int(np.sum((x[:-1] < 0) & (x[1:] >= 0)))
    assert abs(zc - 440) < 3


# ---------------------------------------------------------------------------------------------------- pickup.rs
KNEE, MAXY, SENS = 0.94, 0.98, 1.8375                         # PICKUP_KNEE_Y, PICKUP_MAX_Y, PICKUP_SENSITIVITY (pickup.rs:62-86)


def test_pickup_soft_saturate(oracle):                       # pickup.rs:164-254 (six tests)
    f = lambda y: oracle.lib().owo_pickup_soft_saturate(d(y))
    for y in (0.0, 0.1, 0.5, 0.85, 0.9, 0.93, -0.5, -0.93):                        # identity below the knee
        assert abs(f(y) - y) < 1e-15
    assert abs(f(KNEE - 1e-9) - f(KNEE + 1e-9)) < 1e-7                            # continuous at the knee
    for y in (0.95, 0.96, 0.98, 1.0, 2.0, 100.0, -100.0):                         # bounded, never undershoots the knee
        assert KNEE <= abs(f(y)) <= MAXY + 1e-15
    for y in (0.95, 0.96, 0.97, 0.98, 1.0, 1.5):                                  # strictly inside (knee, limit)
        assert KNEE < f(y) < MAXY
    prev = f(-1.5)                                                                # monotone
    for i in range(1, 601):
        cur = f(-1.5 + i * 0.005)
        assert cur >= prev - 1e-12
        prev = cur
    for y in (0.86, 0.9, 0.95, 0.98, 1.5, 5.0):                                   # odd-symmetric
        assert abs(f(y) + f(-y)) < 1e-12


def _pickup(L, x, sr=44100.0, ds=0.85):
    buf = np.ascontiguousarray(x, dtype=float).copy()
    L.owo_pickup_process(d(sr), d(ds), _p(buf), C.c_size_t(buf.size))
    return buf


def test_pickup_dc_equilibrium_and_rc_response(oracle):      # :256-303
    L = oracle.lib()
    sr = 44100.0
    assert np.max(np.abs(_pickup(L, np.zeros(int(sr * 0.05))))) < 1e-10
    fc = 1.0 / (2.0 * np.pi * 68.88e-6)                       # TAU (pickup.rs:30-60): 2312 Hz
    assert abs(fc - 2312.0) < 2.0
    amplitude = 0.01
    for freq in (100.0, 500.0, 1000.0, 2312.0, 5000.0, 10000.0):
        n = int(sr * 0.1)
        y = _pickup(L, amplitude * np.sin(2 * np.pi * freq * np.arange(n) / sr))
        measured = np.max(np.abs(y[n // 2:]))
        expected = amplitude * 0.85 * SENS * freq / np.sqrt(freq * freq + fc * fc)
        assert abs(20 * np.log10(measured / expected)) < 2.0, freq


def test_pickup_hpf_and_nonlinearity(oracle):                # :306-405
    L = oracle.lib()
    sr = 44100.0
    n = int(sr * 0.05)
    pk = np.max(np.abs(_pickup(L, np.sin(2 * np.pi * 10000.0 * np.arange(n) / sr))[n // 2:]))
    assert 0.5 < pk < 12.0                                     # passes 10 kHz
    n = int(sr * 0.1)
    pk = np.max(np.abs(_pickup(L, np.sin(2 * np.pi * 100.0 * np.arange(n) / sr))[n // 2:]))
    assert pk < 0.65                                           # attenuates 100 Hz
    n = int(sr * 0.2)
    y = _pickup(L, np.sin(2 * np.pi * 2000.0 * np.arange(n) / sr))[n * 3 // 4:]
    t = np.arange(y.size) / sr
    h1, h2, h3 = (abs(np.sum(y * np.exp(-2j * np.pi * k * 2000.0 * t))) for k in (1, 2, 3))
    assert h2 > h3 and h2 / h1 > 0.05                          # even harmonics from the capacitance modulation
    y = _pickup(L, 0.5 * np.sin(2 * np.pi * 500.0 * np.arange(n) / sr))[n // 2:]
    assert np.max(y) > 1.05 * abs(np.min(y))                   # asymmetry below the RC corner


# ---------------------------------------------------------------------------------------------------- mlp_correction.rs
def test_mlp(oracle):                                        # mlp_correction.rs:148-202
    L = oracle.lib()
    out = np.zeros(11)
    L.owo_mlp_infer(33, d(0.8), _p(out))                       # below the fade range = identity(): exact zeros and ones
    assert np.all(out[:5] == 0.0) and np.all(out[5:10] == 1.0) and out[10] == 1.0
    L.owo_mlp_infer(60, d(0.8), _p(out))                       # infer produces corrections
    assert np.any(np.abs(out[:5]) > 0.01) or np.any(np.abs(out[5:10] - 1.0) > 0.01) or abs(out[10] - 1.0) > 0.01
    a, b = np.zeros(11), np.zeros(11)
    L.owo_mlp_infer(40, d(0.8), _p(a)); L.owo_mlp_infer(80, d(0.8), _p(b))
    assert np.any(np.abs(a[:5] - b[:5]) > 0.001) or np.any(np.abs(a[5:10] - b[5:10]) > 0.001)
    for midi in (33, 48, 60, 72, 84, 96):
        for vel in (0.2, 0.5, 0.8, 1.0):
            L.owo_mlp_infer(midi, d(vel), _p(out))
            assert np.all(np.abs(out[:5]) <= 100.0) and np.all((out[5:10] >= 0.3) & (out[5:10] <= 3.0)) and 0.7 <= out[10] <= 1.2


# ---------------------------------------------------------------------------------------------------- hammer.rs
def test_hammer_dwell(oracle):                               # hammer.rs:205-221
    L = oracle.lib()
    ff, pp, at = np.zeros(7), np.zeros(7), np.zeros(7)
    L.owo_dwell_attenuation(d(1.0), d(262.0), _p(RATIOS.copy()), _p(ff))
    L.owo_dwell_attenuation(d(0.1), d(262.0), _p(RATIOS.copy()), _p(pp))
    assert np.all(ff[1:] >= pp[1:])
    L.owo_dwell_attenuation(d(0.5), d(440.0), _p(RATIOS.copy()), _p(at))
    assert abs(at[0] - 1.0) < 1e-10


def test_hammer_attack_noise(oracle):                        # :223-239
    L = oracle.lib()
    buf = np.zeros(700)
    L.owo_attack_noise_render(d(1.0), d(440.0), d(44100.0), C.c_uint(0x12345678), _p(buf), C.c_size_t(700))
    assert np.sum(buf[:100] ** 2) > 5.0 * np.sum(buf[600:] ** 2)
    buf = np.zeros(1000)
    assert L.owo_attack_noise_render(d(1.0), d(440.0), d(44100.0), C.c_uint(0x12345678), _p(buf), C.c_size_t(1000)) == 1


def test_hammer_onset_ramp(oracle):                          # :242-286
    L = oracle.lib()
    o = lambda v, f: L.owo_onset_ramp_time(d(v), d(f))
    bass, mid, treble = o(1.0, 65.0), o(1.0, 262.0), o(1.0, 1047.0)
    assert bass > mid > treble
    assert abs(bass - 1.0 / 65.0) < 0.001 and abs(treble - 0.002) < 0.0001 and abs(mid - 1.0 / 262.0) < 0.001
    ff, pp = o(1.0, 262.0), o(0.0, 262.0)
    assert pp > ff and abs(ff - 1.0 / 262.0) < 0.001 and abs(pp - 2.0 / 262.0) < 0.001


# ---------------------------------------------------------------------------------------------------- voice.rs, variation.rs
def test_voice_render_note(oracle):                          # voice.rs:229-262
    assert np.max(np.abs(oracle.render_note(60, 0.8, 0.5, 44100.0))) > 0.0
    soft, loud = oracle.render_note(60, 0.3, 0.1, 44100.0), oracle.render_note(60, 1.0, 0.1, 44100.0)
    assert np.max(np.abs(loud)) > np.max(np.abs(soft))
    a, b = oracle.render_note(60, 0.8, 0.1, 44100.0), oracle.render_note(60, 0.8, 0.1, 44100.0)
    assert np.array_equal(a, b)
    assert not np.array_equal(a, oracle.render_note(72, 0.8, 0.1, 44100.0))


def test_variation(oracle):                                  # variation.rs:45-78
    L = oracle.lib()
    assert L.owo_freq_detune(60) == L.owo_freq_detune(60)
    assert L.owo_freq_detune(60) != L.owo_freq_detune(61)
    off = np.zeros(7)
    for midi in range(33, 97):
        assert 0.99 < L.owo_freq_detune(midi) < 1.01                               # the reference's band
        assert abs(L.owo_freq_detune(midi) - 1.0) < 0.002                          # what the hash actually spans: +-3 cents, 2^(3/1200) - 1 = 0.00173
        L.owo_mode_amplitude_offsets(midi, _p(off))
        assert np.all((off > 0.9) & (off < 1.1))                                   # +-8 %
