"""ow_div (the kernels' f64 division: the compiler's IEEE sequence without the v_div_scale pre-scaling) against the compiler's own
`a / b` on the device and against numpy, bit for bit, over the operand ranges the DSP produces and far beyond them."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _div(hiplib, a, b):
    a = np.ascontiguousarray(a, dtype=np.float64); b = np.ascontiguousarray(b, dtype=np.float64)
    f = np.zeros_like(a); i = np.zeros_like(a)
    assert hiplib.ow_debug_div(a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), a.size, f.ctypes.data_as(C.c_void_p),
                               i.ctypes.data_as(C.c_void_p), 0) == 0
    return f, i


def _rand(rng, n, emin, emax):
    m = rng.uniform(1.0, 2.0, n) * rng.choice([-1.0, 1.0], n)
    return np.ldexp(m, rng.integers(emin, emax + 1, n))


def _same_bits(x, y):
    return np.array_equal(x.view(np.uint64), y.view(np.uint64))


@pytest.mark.parametrize("emin,emax", [(-40, 40), (-200, 200), (-480, 480)])
def test_bit_identical_in_the_normal_range(hiplib, emin, emax):
    rng = np.random.default_rng(emax)
    n = 1 << 24
    a, b = _rand(rng, n, emin, emax), _rand(rng, n, emin, emax)
    fast, ieee = _div(hiplib, a, b)
    assert _same_bits(fast, ieee)
    with np.errstate(all="ignore"):
        assert _same_bits(fast, a / b)            # and both are the correctly rounded quotient


def test_hard_cases_and_specials(hiplib):
    rng = np.random.default_rng(1)
    # quotients next to rounding boundaries: a = b * (1 + k ulp) and neighbours, small-integer ratios, powers of two, constants of the DSP
    b = _rand(rng, 1 << 20, -30, 30)
    k = rng.integers(-4, 5, b.size)
    a = b * (1.0 + k * 2.0 ** -52)
    a2 = np.nextafter(a, np.inf)
    cases_a = np.concatenate([a, a2, rng.integers(-1000, 1000, 1 << 18).astype(np.float64), np.full(64, 1.0), _rand(rng, 1 << 18, -30, 30)])
    cases_b = np.concatenate([b, b, rng.integers(1, 1000, 1 << 18).astype(np.float64), np.ldexp(1.0, np.arange(-32, 32)), np.full(1 << 18, 0.026)])
    fast, ieee = _div(hiplib, cases_a, cases_b)
    assert _same_bits(fast, ieee)
    with np.errstate(all="ignore"):
        assert _same_bits(fast, cases_a / cases_b)
    # zero, signed zero, infinities, NaN, division by zero: v_div_fixup is kept, so these are IEEE too
    sp = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-300, -3.5e10])
    A, B = [x.ravel() for x in np.meshgrid(sp, sp)]
    fast, ieee = _div(hiplib, A, B)
    with np.errstate(all="ignore"):
        want = A / B
    nan = np.isnan(want)
    assert np.array_equal(np.isnan(fast), nan) and np.array_equal(np.isnan(ieee), nan)
    assert _same_bits(fast[~nan], want[~nan]) and _same_bits(ieee[~nan], want[~nan])


def test_outside_the_promise_is_reported_not_hidden(hiplib):
    """Denormal operands / quotients and exponent differences near the format's range are where v_div_scale matters.  The DSP never
    gets there (voices are freed at -80 dB, node voltages are volts); this pins how far off the unscaled sequence can be: at most
    one unit in the last place of the (denormal) result, never a wrong order of magnitude."""
    rng = np.random.default_rng(2)
    n = 1 << 20
    a, b = _rand(rng, n, -1022, -900), _rand(rng, n, 100, 120)        # quotient deep in the denormal range or zero
    fast, ieee = _div(hiplib, a, b)
    ulp = 2.0 ** -1074
    assert np.max(np.abs(fast - ieee)) <= ulp


CONSTANTS = [2147483647.5, 1.0 * 2.58519910000000012e-2, 10.95 - 0.70, 0.026, 0.013 * 0.013, 22.0, 2147483647.0, 0.98 - 0.94]


@pytest.mark.parametrize("which", range(8))
def test_constant_divisors_bit_identical(hiplib, which):
    """OW_DIV_C(a, B): reciprocal folded at compile time, product, exact residual, one correction.  Numerators: what the call site
    can produce and several decades around it, rounding-boundary neighbours of multiples of B, zeros, infinities, NaN."""
    rng = np.random.default_rng(100 + which)
    B = CONSTANTS[which]
    n = 1 << 23
    a = np.concatenate([_rand(rng, n, -60, 40), rng.uniform(-40.0, 40.0, n), B * rng.integers(-1 << 20, 1 << 20, n).astype(np.float64),
                        np.nextafter(B * rng.integers(1, 1 << 20, n).astype(np.float64), np.inf),
                        np.array([0.0, -0.0, np.inf, -np.inf, np.nan, B, -B, 1.0, 2.0 ** -1060])])
    f = np.zeros_like(a); i = np.zeros_like(a)
    assert hiplib.ow_debug_div_const(which, a.ctypes.data_as(C.c_void_p), a.size, f.ctypes.data_as(C.c_void_p), i.ctypes.data_as(C.c_void_p), None, 0) == 0
    with np.errstate(all="ignore"):
        want = a / B
    nan = np.isnan(want)
    assert np.array_equal(np.isnan(f), nan) and np.array_equal(np.isnan(i), nan)
    assert _same_bits(f[~nan], want[~nan]) and _same_bits(i[~nan], want[~nan])


def test_jitter_draw_exhaustive(hiplib):
    """u = (jitter_state >> 1) / 2147483647.5 (reed.rs:267-272): every one of the 2^31 possible numerators, on the device."""
    bad = C.c_uint64(123)
    assert hiplib.ow_debug_div_const(0, None, 0, None, None, C.byref(bad), 0) == 0
    assert bad.value == 0


def test_noise_draw_exhaustive(hiplib):
    """noise = (rng as i32) / 2147483647.0 (hammer.rs:192-195, the attack-noise burst): every one of the 2^32 numerators, on the device."""
    bad = C.c_uint64(123)
    assert hiplib.ow_debug_div_const(6, None, 0, None, None, C.byref(bad), 0) == 0
    assert bad.value == 0


def test_bounded_exp_is_the_library_exp(hiplib):
    """exp_bounded (the junction law's exponential: library algorithm without its overflow / underflow selects) on the whole clamp
    range [-1 V, 0.85 V] / 0.026 V and a margin around it: bit-identical to the device library's exp(), and within one unit in the
    last place of numpy's."""
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(-38.5, 32.7, 1 << 24), rng.uniform(-60.0, 60.0, 1 << 22), np.linspace(-38.5, 32.7, 100001),
                        np.array([0.0, -0.0, -1.0 / 0.026, 0.85 / 0.026, 1e-300, -1e-300, 1e-17])])
    f = np.zeros_like(x); l = np.zeros_like(x)
    assert hiplib.ow_debug_unary(0, x.ctypes.data_as(C.c_void_p), x.size, f.ctypes.data_as(C.c_void_p), l.ctypes.data_as(C.c_void_p), 0) == 0
    assert _same_bits(f, l)
    ref = np.exp(x)
    assert np.max(np.abs(f - ref) / np.spacing(ref)) <= 1.0


def test_tanh_fast_accuracy(hiplib):
    """tanh_fast (power amp, speaker): expm1-based, against numpy's tanh over the whole domain -- tiny arguments (the power amp's
    usual regime, where 1 - 2/(e^2x + 1) would cancel), the reduction boundaries, saturation, signs, specials."""
    rng = np.random.default_rng(6)
    x = np.concatenate([rng.uniform(-1.0, 1.0, 1 << 23), rng.uniform(-25.0, 25.0, 1 << 22), _rand(rng, 1 << 22, -60, 2),
                        np.ldexp(1.0, np.arange(-1074, 12)), 0.5 * np.log(2.0) * (np.arange(-60, 61) + 0.5) / 2.0,
                        np.array([0.0, -0.0, 19.0, 20.0, 20.0000001, 700.0, -700.0, 1e300, np.inf, -np.inf])])
    f = np.zeros_like(x); l = np.zeros_like(x)
    assert hiplib.ow_debug_unary(1, x.ctypes.data_as(C.c_void_p), x.size, f.ctypes.data_as(C.c_void_p), l.ctypes.data_as(C.c_void_p), 0) == 0
    with np.errstate(all="ignore"):
        ref = np.tanh(x.astype(np.longdouble))                                             # x87 extended precision: 64-bit significand
        sp = np.spacing(np.abs(ref.astype(np.float64)) + (ref == 0))

        def ulps(y):
            return np.max(np.abs(y.astype(np.longdouble) - ref).astype(np.float64) / sp)
        assert ulps(f) <= 2.0, ulps(f)
        assert ulps(l) <= 1.0 and ulps(np.tanh(x)) <= 2.0                                  # the device library and glibc on the same yardstick
    assert np.array_equal(np.signbit(f), np.signbit(x)) and np.all(np.abs(f) <= 1.0)
    nan = np.array([np.nan]); fn = np.zeros(1); ln = np.zeros(1)
    assert hiplib.ow_debug_unary(1, nan.ctypes.data_as(C.c_void_p), 1, fn.ctypes.data_as(C.c_void_p), ln.ctypes.data_as(C.c_void_p), 0) == 0
    assert np.isnan(fn[0]) and np.isnan(ln[0])


def _forms(hiplib, mode, a, b, y=None):
    a = np.ascontiguousarray(a, dtype=np.float64); b = np.ascontiguousarray(b, dtype=np.float64)
    yy = np.ascontiguousarray(y, dtype=np.float64) if y is not None else None
    f = np.zeros_like(a); i = np.zeros_like(a)
    assert hiplib.ow_debug_div_forms(mode, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), yy.ctypes.data_as(C.c_void_p) if yy is not None else None,
                                     a.size, f.ctypes.data_as(C.c_void_p), i.ctypes.data_as(C.c_void_p), 0) == 0
    return f, i


def _power_amp_divisors():
    """The per-device constants bjt_evaluate divides by (gen_power_amp.rs:7870-8017): NF VT, NR VT, NE VT, NC VT, VAR, VAF, IKF, IKR of
    the eight transistors, formed like ow_consts_host.hpp forms them."""
    import os
    import re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data", "ow_gen_data.h")).read()

    def tab(name):
        m = re.search(r"double PA_DEV_" + name + r"\[8\] = \{(.*?)\};", src, re.S)
        return np.array([float(x) if "INF" not in x else np.inf for x in m.group(1).replace(" ", "").split(",") if x])
    vt = tab("VT")
    out = [tab("NF") * vt, tab("NR") * vt, tab("NE") * vt, tab("NC") * vt, tab("VAR"), tab("VAF"), tab("IKF"), tab("IKR")]
    d = np.unique(np.concatenate(out))
    return d[np.isfinite(d) & (d != 0.0)]


def test_short_division_forms_are_the_ieee_quotient(hiplib):
    """Round 3 added three shorter forms of the same division: a host-computed reciprocal for divisors that are per-device constants
    (the power amp's device law), one refined reciprocal shared by the quotients over a pivot (the 3x3 / 4x4 / 6x6 / 16x16 eliminations),
    and that form without v_div_fixup where operands are finite and pivots checked (the melange column solves).  Each must give the
    compiler's `a / b` bit for bit on the ranges it is used on."""
    rng = np.random.default_rng(33)
    n = 1 << 22
    # (0) every power-amp divisor x junction-voltage / current sized numerators
    for b0 in _power_amp_divisors():
        a = np.concatenate([_rand(rng, 1 << 16, -60, 10), rng.uniform(-30.0, 30.0, 1 << 16), np.array([0.0, -0.0])])
        b = np.full(a.size, b0)
        fast, ieee = _forms(hiplib, 0, a, b, 1.0 / b)
        assert _same_bits(fast, ieee), b0
    # (1) shared refined reciprocal == ow_div for random operands, incl. the special values (the fixup is kept there)
    a, b = _rand(rng, n, -200, 200), _rand(rng, n, -200, 200)
    fast, ieee = _forms(hiplib, 1, a, b)
    assert _same_bits(fast, ieee)
    # (2) without the fixup: finite operands with moderate exponents (pivots and sums of the preamp's LU: 1e-30 < |b|, |a / b| far from
    # the ends of the range), zero numerators of either sign included
    a = _rand(rng, n, -80, 80)
    b = _rand(rng, a.size, -90, 90)
    fast, ieee = _forms(hiplib, 2, a, b)
    assert _same_bits(fast, ieee)
    # a zero numerator gives a zero quotient whose SIGN the fixup would have set (fma(r, y, q) adds +0 to -0): equal as numbers, and a
    # signed zero cannot turn into anything else in the substitutions that follow (x - 0 * y, sums, products; no division by it)
    a0 = np.concatenate([np.zeros(1024), -np.zeros(1024)])
    fast, ieee = _forms(hiplib, 2, a0, _rand(rng, a0.size, -90, 90))
    assert np.array_equal(fast, ieee) and np.all(fast == 0.0)


def test_onset_gain_accuracy(hiplib):
    """The reed's onset gain (reed.rs:251-264) for mid velocities is cosine^p with p in (1.001, 1.999); the kernels form it as
    exp(p ln cosine) instead of the library's pow (which carries the logarithm in double-double: ~250 instructions per lane and sample of
    a re-struck wavefront).  The result is a gain in [0, 1): its absolute error stays below 2.5e-16 -- about one ulp of the gains near 1
    that carry the signal -- over the whole ramp incl. its first samples (cosine ~ 1e-7) and the end points, and equals the library's
    pow to 4 ulp relative where the gain is not tiny."""
    rng = np.random.default_rng(11)
    x = np.concatenate([rng.uniform(0.0, np.pi, 1 << 22), np.pi * np.arange(1, 2049) / 2049.0, np.ldexp(np.pi, -np.arange(1, 40)),
                        np.array([0.0, np.pi, np.pi / 2, 1e-9])])
    for which, p in ((2, 1.25), (3, 1.5), (4, 1.9)):
        f = np.zeros_like(x); l = np.zeros_like(x)
        assert hiplib.ow_debug_unary(which, x.ctypes.data_as(C.c_void_p), x.size, f.ctypes.data_as(C.c_void_p), l.ctypes.data_as(C.c_void_p), 0) == 0
        xl = x.astype(np.longdouble)
        ref = (0.5 * (1.0 - np.cos(xl))) ** np.longdouble(p)
        # the cosine itself is formed in f64 on the device (1 - cos x cancels for small x exactly as in the reference): compare on the
        # device's own cosine, i.e. against pow of the f64 value
        c64 = 0.5 * (1.0 - np.cos(x))
        ref64 = (c64.astype(np.longdouble)) ** np.longdouble(p)
        err = np.abs(f.astype(np.longdouble) - ref64).astype(np.float64)
        lib_err = np.abs(l.astype(np.longdouble) - ref64).astype(np.float64)
        assert np.max(err) <= 2.5e-16 + 4 * np.max(lib_err), (p, np.max(err), np.max(lib_err))
        big = l > 1e-3
        # relative to the library's pow on the same phase: |p ln c| eps from the logarithm, the exponential's ulp, and the two sides' own
        # cosines (<= 1.5 ulp of a number near 1 each, i.e. 2.5e-16 absolute on c = (1 - cos x) / 2, which the power turns into p / c relative)
        allowed = l[big] * (32 * 2.2204e-16 + p * 2.5e-16 / c64[big])
        assert np.all(np.abs(f[big] - l[big]) <= allowed), (p, np.max(np.abs(f[big] - l[big]) / allowed))
        assert f[x == 0.0][0] == 0.0 and np.all(f >= 0.0) and np.all(f <= 1.0)
        assert np.all(np.isfinite(f))
        del ref


@pytest.mark.gpu
def test_damper_ramp_exp_accuracy(hiplib):
    """Damper ramp (reed.rs:227-247): every released voice multiplies its seven mode envelopes by exp(-damper_rate * t / ramp) per sample,
    with damper_rate <= 2000 / sr (0.0454 at 44.1 kHz).  The kernels evaluate that exponential on [0, 1/8] by a degree-11 polynomial
    without range reduction (exp_neg_small, ow_voice_dev.h): within 1 ulp of the device library's exp and within 0.6 ulp of the true value
    (measured: 0.562)
    (the reference's f64::exp is glibc's, itself specified to 1 ulp); larger arguments take the library."""
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(0.0, 0.125, 1 << 22), rng.uniform(0.0, 0.0454, 1 << 22), np.ldexp(1.0, -np.arange(3, 60)),
                        np.array([0.0, 0.125, 2000.0 / 44100.0, 2000.0 / 48000.0, 2000.0 / 96000.0]), rng.uniform(0.125, 3.0, 4096)])
    f = np.zeros_like(x); l = np.zeros_like(x)
    assert hiplib.ow_debug_unary(5, x.ctypes.data_as(C.c_void_p), x.size, f.ctypes.data_as(C.c_void_p), l.ctypes.data_as(C.c_void_p), 0) == 0
    ref = np.exp(-x.astype(np.longdouble))
    ulp = np.spacing(ref.astype(np.float64))
    err = np.abs(f.astype(np.longdouble) - ref).astype(np.float64) / ulp
    lib_err = np.abs(l.astype(np.longdouble) - ref).astype(np.float64) / ulp
    small = x <= 0.125
    assert np.max(err[small]) <= 0.6, np.max(err[small])
    assert np.max(np.abs(f - l) / ulp) <= 1.0
    assert np.array_equal(f[~small], l[~small])              # the library beyond 1/8
    assert f[x == 0.0][0] == 1.0
    assert np.max(lib_err) <= 1.0


def test_folded_envelope_is_the_power_of_the_decay(hiplib):
    """env_after (ow_kernels.h): the envelope k samples into a block of the steady voice kernel, which carries it on the radius of its
    quadrature pairs (DESIGN deviation 15), is env0 * decay^k by the device library's pow: within an ulp of numpy's, where the
    reference's k-fold product carries ~sqrt(k) ulp of rounding of its own (checked here: the two differ by less than k / 4 ulp)."""
    rng = np.random.default_rng(15)
    d = np.concatenate([1.0 - 10.0 ** rng.uniform(-6.0, -1.5, 1 << 16), np.array([1.0, 0.5, 0.999999999])])
    f = np.zeros_like(d); l = np.zeros_like(d)
    assert hiplib.ow_debug_unary(6, d.ctypes.data_as(C.c_void_p), d.size, f.ctypes.data_as(C.c_void_p), l.ctypes.data_as(C.c_void_p), 0) == 0
    k = 512 + (np.arange(d.size) & 1023)
    want = np.power(d, k.astype(np.float64))
    ok = want > 1e-300
    assert _same_bits(f, l)
    assert np.max(np.abs(f[ok] - want[ok]) / np.spacing(want[ok])) <= 1.0
    # the reference's recurrence: k multiplications
    i = int(np.argmin(np.abs(d - 0.9995)))
    e = 1.0
    for _ in range(int(k[i])):
        e *= float(d[i])
    assert abs(e - f[i]) / np.spacing(f[i]) < k[i] / 4
