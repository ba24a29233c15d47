"""The voice path pinned by a SECOND restatement.

The reference holds acceptance bands for `Voice::note_on -> ModalReed::render -> Pickup::process`, no sample vectors, so
the C++ oracle's fidelity there rested on line-by-line correspondence alone.  oracle/voice_path_numpy.py restates the same path a second
time, independently, from the Rust (voice.rs:28-221, reed.rs:90-306, pickup.rs:30-149, hammer.rs:26-198, tables.rs:32-830,
variation.rs:10-38) in numpy scalars; two restatements that agree sample for sample catch the transcription slips bands cannot.
CPU only: neither side is the product."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import oracle_binding as ob
import voice_path_numpy as vp

SR = 48000.0
CASES = [(n, v) for n in (33, 60, 96) for v in (0.3, 1.0)]


@pytest.mark.parametrize("midi,vel", CASES)
def test_render_note_agrees_with_the_cpp_oracle_over_one_second(midi, vel):
    """1 s of Voice::render_note: within 1e-13 of the render's peak on every sample (measured: bit-identical -- both sides are IEEE f64
    without contraction over the same glibc functions, so any difference at all would be a difference in the statements)."""
    ob.lib()
    a = vp.render_note(midi, vel, 1.0, SR)
    b = ob.render_note(midi, vel, 1.0, SR)
    assert a.shape == b.shape == (48000,)
    peak = float(np.max(np.abs(b)))
    assert peak > 1e-5 and np.all(np.isfinite(a))
    assert float(np.max(np.abs(a - b))) <= 1e-13 * peak, (midi, vel, float(np.max(np.abs(a - b))) / peak)


def test_note_on_parameters_agree_field_by_field():
    """Voice::note_on's outputs (phase increments, amplitudes after spatial coupling / dwell / +-8 % / velocity curve, decay multipliers,
    onset ramp, displacement scale, post-pickup gain) against the oracle's packed block (owo_voice_params), to the last bit."""
    import ctypes as C
    L = ob.lib()
    L.owo_voice_params.argtypes = [C.c_int, C.c_double, C.c_double, C.c_uint, C.c_int, C.c_void_p]
    for midi, vel in CASES + [(45, 0.62), (72, 0.05), (84, 0.999)]:
        out = np.zeros(64)
        L.owo_voice_params(midi, vel, SR, (midi * 2654435761) & 0xFFFFFFFF, 0, out.ctypes.data)
        taps = {}
        vp.render_note(midi, vel, 0.001, SR, taps=taps)
        p = taps["params"]
        tau = np.float64(6.283185307179586)
        for i in range(7):
            assert out[i] == tau * (np.float64(p["f0"]) * np.float64(p["ratios"][i])) / np.float64(SR), (midi, vel, i, "phase_inc")
            assert out[7 + i] == p["amps"][i], (midi, vel, i, "amplitude")
        assert out[42] == p["onset_samples"] and out[44] == p["onset_exp"]
        assert out[49] == p["ds"] and out[50] == p["gain"], (midi, vel, "ds / gain")


def test_extended_precision_run_bounds_the_f64_rounding_noise():
    """The same statements in 80-bit arithmetic: what f64 rounding (incl. the rounded rotation coefficients and the eight cancelled digits
    of mode_shape) does to a note is far inside the 1e-5 output bar -- and far above 1e-13, which is why the two f64 restatements above
    could only agree that closely by executing the same operations."""
    if np.finfo(np.longdouble).nmant < 63:
        pytest.skip("no extended precision on this platform")
    ob.lib()
    a = vp.render_note(60, 1.0, 0.5, SR, T=np.longdouble)
    b = ob.render_note(60, 1.0, 0.5, SR)
    d = float(np.max(np.abs(a - b))) / float(np.max(np.abs(b)))
    assert 1e-13 < d < 1e-4, d
