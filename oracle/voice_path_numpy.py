"""TEST INFRASTRUCTURE (oracle/): a SECOND, independent restatement of the voice path, in numpy scalars.

    Voice::note_on -> ModalReed::render (+ AttackNoise::render) -> Pickup::process -> x post_pickup_gain   (= Voice::render_note)

written from the reference's Rust alone (file:line below, /root/reference/crates/openwurli-dsp/src/), without looking at the C++
oracle (oracle/ow_voice.hpp, ow_tables.hpp).  The reference holds no sample vectors for this path, only acceptance bands
(SURVEY.md 8c): two restatements that agree sample for sample catch the transcription slips the bands cannot.
tests/test_oracle_voice_path_numpy.py compares the two.

`T` is the scalar type every quantity is carried in.  T = numpy.float64 repeats the reference's arithmetic (IEEE f64, one rounding
per operation, no contraction; the C library's elementary functions) and is what the C++ oracle is held to at 1e-13 of peak; T = numpy.longdouble runs
the same statements in 80-bit arithmetic and bounds what f64 rounding does to one second of a note.

Only pure-Python loops: one second of one voice takes a few seconds.  Not imported by the product (tests/test_cabi.py checks).
"""
import math

import numpy as np

NUM_MODES = 7
U32 = 0xFFFFFFFF


class _Fn:
    """Elementary functions for scalar type T.  float64: the C library's (math.* = glibc here, what Rust's f64 methods and the C++
    oracle call -- numpy's own scalar loops differ from it in the last bit now and then, and mode_shape's cosh - cos - sigma (sinh - sin)
    cancels eight digits, so a last-bit difference in cosh shows up at 1e-8 in the upper modes' amplitudes); longdouble: numpy's."""

    def __init__(self, T):
        self.T = T
        if T is np.float64:
            w = lambda f: (lambda *a: np.float64(f(*[float(x) for x in a])))
            self.exp, self.log, self.log10, self.cos, self.sin = w(math.exp), w(math.log), w(math.log10), w(math.cos), w(math.sin)
            self.cosh, self.sinh, self.tanh, self.sqrt, self.pow = w(math.cosh), w(math.sinh), w(math.tanh), w(math.sqrt), w(math.pow)
        else:
            self.exp, self.log, self.log10, self.cos, self.sin = np.exp, np.log, np.log10, np.cos, np.sin
            self.cosh, self.sinh, self.tanh, self.sqrt, self.pow = np.cosh, np.sinh, np.tanh, np.sqrt, np.power


_FN = {}


def F(T):
    if T not in _FN:
        _FN[T] = _Fn(T)
    return _FN[T]


# ------------------------------------------------------------------------------------------------ tables.rs
def midi_to_freq(midi, T):                                     # tables.rs:36-38
    return T(440.0) * F(T).pow(T(2.0), (T(midi) - T(69.0)) / T(12.0))


def tip_mass_ratio(midi, T):                                   # tables.rs:50-76
    m = T(midi)
    anchors = [(33.0, 0.10), (52.0, 0.00), (62.0, 0.00), (74.0, 0.02), (96.0, 0.01)]
    if m <= anchors[0][0]:
        return T(anchors[0][1])
    if m >= anchors[-1][0]:
        return T(anchors[-1][1])
    for (x0, y0), (x1, y1) in zip(anchors[:-1], anchors[1:]):
        if m <= x1:
            t = (m - T(x0)) / (T(x1) - T(x0))
            return T(y0) + t * (T(y1) - T(y0))
    return T(0.0)


EIG_TABLE = [                                                  # tables.rs:90-123
    (0.00, [1.8751, 4.6941, 7.8548, 10.9955, 14.1372, 17.2788, 20.4204]),
    (0.01, [1.8584, 4.6849, 7.8504, 10.9930, 14.1356, 17.2776, 20.4195]),
    (0.05, [1.7920, 4.6477, 7.8316, 10.9830, 14.1288, 17.2726, 20.4158]),
    (0.10, [1.7227, 4.6024, 7.8077, 10.9700, 14.1198, 17.2660, 20.4110]),
    (0.15, [1.6625, 4.5618, 7.7859, 10.9580, 14.1114, 17.2598, 20.4065]),
    (0.20, [1.6097, 4.5254, 7.7659, 10.9470, 14.1036, 17.2540, 20.4023]),
    (0.30, [1.5201, 4.4620, 7.7310, 10.9280, 14.0894, 17.2434, 20.3946]),
    (0.50, [1.3853, 4.3601, 7.6745, 10.8970, 14.0650, 17.2252, 20.3814]),
]


def eigenvalues(mu, T):                                        # tables.rs:84-141
    mu_c = min(max(mu, T(0.0)), T(0.50))
    lo = 0
    for i, (m, _) in enumerate(EIG_TABLE):                     # rposition(row.mu <= mu_clamped)
        if T(m) <= mu_c:
            lo = i
    hi = min(lo + 1, len(EIG_TABLE) - 1)
    mlo, mhi = T(EIG_TABLE[lo][0]), T(EIG_TABLE[hi][0])
    t = (mu_c - mlo) / (mhi - mlo) if mhi > mlo else T(0.0)
    return [T(EIG_TABLE[lo][1][i]) + t * (T(EIG_TABLE[hi][1][i]) - T(EIG_TABLE[lo][1][i])) for i in range(NUM_MODES)]


def mode_ratios(mu, T):                                        # tables.rs:147-151
    b = eigenvalues(mu, T)
    b1 = b[0] * b[0]
    return [(b[i] * b[i]) / b1 for i in range(NUM_MODES)]


def reed_length_mm(midi, T):                                   # tables.rs:159-167
    n = min(max(T(midi) - T(32.0), T(1.0)), T(64.0))
    inches = T(3.0) - n / T(20.0) if n <= 20.0 else T(2.0) - (n - T(20.0)) / T(44.0)
    return inches * T(25.4)


def reed_blank_dims(midi, T):                                  # tables.rs:181-211
    reed = min(max(int(midi) - 32, 1), 64)
    if reed <= 14:
        w = 0.151
    elif reed <= 20:
        w = 0.127
    elif reed <= 42:
        w = 0.121
    elif reed <= 50:
        w = 0.111
    else:
        w = 0.098
    if reed <= 16:
        th = T(0.026)
    elif reed <= 26:
        t = (T(reed) - T(16.0)) / T(10.0)
        th = T(0.026) + t * (T(0.034) - T(0.026))
    else:
        th = T(0.034)
    return T(w) * T(25.4), th * T(25.4)


def reed_compliance(midi, T):                                  # tables.rs:217-221
    l = reed_length_mm(midi, T)
    w, t = reed_blank_dims(midi, T)
    return (l * l * l) / (w * t * t * t)


def pickup_displacement_scale(midi, T):                        # tables.rs:250-252, 279-288
    c, c_ref = reed_compliance(midi, T), reed_compliance(60, T)
    ds = T(0.85) * F(T).pow(c / c_ref, T(0.75))
    return min(max(ds, T(0.02)), T(0.95))


def mode_shape(beta, xi, T):                                   # tables.rs:295-299
    sigma = (F(T).cosh(beta) + F(T).cos(beta)) / (F(T).sinh(beta) + F(T).sin(beta))
    bx = beta * xi
    return F(T).cosh(bx) - F(T).cos(bx) - sigma * (F(T).sinh(bx) - F(T).sin(bx))


def spatial_coupling_coefficients(mu, reed_len_mm, T):         # tables.rs:306, 324-370
    betas = eigenvalues(mu, T)
    ell = min(max(T(6.0) / reed_len_mm, T(0.0)), T(1.0))
    xi_start = T(1.0) - ell
    raw = []
    for beta in betas:
        tip = mode_shape(beta, T(1.0), T)
        if abs(tip) < 1e-30 or ell < 1e-12:
            raw.append(T(1.0))
            continue
        h = ell / T(32.0)
        s = mode_shape(beta, xi_start, T) + mode_shape(beta, T(1.0), T)
        for j in range(1, 32):
            xi = xi_start + T(j) * h
            s = s + (T(4.0) if j % 2 == 1 else T(2.0)) * mode_shape(beta, xi, T)
        integral = s * h / T(3.0)
        k = abs(integral / (ell * tip))
        raw.append(min(max(k, T(0.0)), T(1.0)))
    k1 = raw[0]
    if k1 > 1e-30:
        return [min(max(r / k1, T(0.0)), T(1.0)) for r in raw]
    return [T(1.0)] * NUM_MODES


def fundamental_decay_rate(midi, T):                           # tables.rs:390-395
    f = midi_to_freq(midi, T)
    return max(T(0.005) * F(T).pow(f, T(1.22)), T(3.0))


def mode_decay_rates(midi, ratios, T):                         # tables.rs:418-422
    base = fundamental_decay_rate(midi, T)
    return [base * r * r for r in ratios]


def pickup_rms_proxy(ds, f0, fc, T):                           # tables.rs:438-455
    if ds < 1e-10:
        return T(0.0)
    r = (T(1.0) - F(T).sqrt(T(1.0) - ds * ds)) / ds
    inv_sqrt = T(1.0) / F(T).sqrt(T(1.0) - ds * ds)
    sum_sq = T(0.0)
    r_n = r
    for n in range(1, 9):
        cn = T(2.0) * r_n * inv_sqrt
        nf = T(n) * f0
        hpf = nf / F(T).sqrt(nf * nf + fc * fc)
        sum_sq = sum_sq + (cn * hpf) * (cn * hpf)
        r_n = r_n * r
    return F(T).sqrt(sum_sq)


TRIM_ANCHORS = [(36.0, -1.3), (40.0, 0.0), (44.0, -1.3), (48.0, 0.7), (52.0, 0.2), (56.0, -1.0), (60.0, 0.0), (64.0, 0.9), (68.0, 1.2),
                (72.0, 0.0), (76.0, 1.8), (80.0, 2.4), (84.0, 3.6)]                   # tables.rs:471-485


def register_trim_db(midi, T):                                 # tables.rs:465-503
    m = T(midi)
    if m <= TRIM_ANCHORS[0][0]:
        return T(TRIM_ANCHORS[0][1])
    if m >= TRIM_ANCHORS[-1][0]:
        return T(TRIM_ANCHORS[-1][1])
    for (x0, y0), (x1, y1) in zip(TRIM_ANCHORS[:-1], TRIM_ANCHORS[1:]):
        if m <= x1:
            t = (m - T(x0)) / (T(x1) - T(x0))
            return T(y0) + t * (T(y1) - T(y0))
    return T(0.0)


def velocity_exponent(midi, T):                                # tables.rs:632-653
    m = T(midi)
    d = (m - T(62.0)) / T(15.0)
    t = F(T).exp(T(-0.5) * (d * d))
    mn = T(0.55) if m < 62.0 else T(1.3)
    return mn + t * (T(1.7) - mn)


def velocity_scurve(v, T):                                     # tables.rs:659-665
    k = T(1.5)
    s = T(1.0) / (T(1.0) + F(T).exp(-k * (v - T(0.5))))
    s0 = T(1.0) / (T(1.0) + F(T).exp(k * T(0.5)))
    s1 = T(1.0) / (T(1.0) + F(T).exp(-k * T(0.5)))
    return (s - s0) / (s1 - s0)


def output_scale(midi, v, T):                                  # tables.rs:574-616 (default CalibrationConfig :267-276)
    fc = T(2312.0)
    ds = pickup_displacement_scale(midi, T)
    f0 = midi_to_freq(midi, T)
    sc = velocity_scurve(v, T)
    vel_scale = F(T).pow(sc, velocity_exponent(midi, T))
    vel_scale_c4 = F(T).pow(sc, velocity_exponent(60, T))
    eff = max(ds * vel_scale, T(1e-6))
    eff_ref = max(T(0.85) * vel_scale_c4, T(1e-6))
    rms = pickup_rms_proxy(eff, f0, fc, T)
    rms_ref = pickup_rms_proxy(eff_ref, midi_to_freq(60, T), fc, T)
    flat_db = T(-20.0) * F(T).log10(rms / rms_ref)
    voicing_db = T(-0.04) * max(T(midi) - T(60.0), T(0.0))
    trim = register_trim_db(midi, T)
    blend = F(T).pow(v, T(1.3))
    return F(T).pow(T(10.0), (T(-35.0) + flat_db + voicing_db + trim * blend) / T(20.0))


def note_params(midi, T):                                      # tables.rs:804-830
    mu = tip_mass_ratio(midi, T)
    ratios = mode_ratios(mu, T)
    base = [1.0, 0.005, 0.0035, 0.0018, 0.0011, 0.0007, 0.0005]                       # tables.rs:32-33
    coupling = spatial_coupling_coefficients(mu, reed_length_mm(midi, T), T)
    return midi_to_freq(midi, T), ratios, [T(b) * k for b, k in zip(base, coupling)], mode_decay_rates(midi, ratios, T)


# ------------------------------------------------------------------------------------------------ variation.rs
def hash_f64(midi, seed, T):                                   # variation.rs:10-19
    h = 2166136261
    h ^= midi
    h = (h * 16777619) & U32
    h ^= seed
    h = (h * 16777619) & U32
    h ^= h >> 16
    h = (h * 2654435769) & U32
    return T(h & 0x00FFFFFF) / T(16777216.0)


def freq_detune(midi, T):                                      # variation.rs:26-29
    return T(1.0) + (hash_f64(midi, 0xDEAD, T) * T(2.0) - T(1.0)) * T(0.00173)


def mode_amplitude_offsets(midi, T):                           # variation.rs:33-38
    return [T(1.0) + (hash_f64(midi, (0xBEEF + i) & U32, T) * T(2.0) - T(1.0)) * T(0.08) for i in range(NUM_MODES)]


# ------------------------------------------------------------------------------------------------ hammer.rs
def dwell_attenuation(v, f0, ratios, T):                       # hammer.rs:26-29 (dwell_time), 67-90
    cycles = T(0.75) + T(0.25) * (T(1.0) - v)
    t_dwell = min(max(cycles / f0, T(0.0003)), T(0.020))
    sigma_sq = T(8.0) * T(8.0)
    att = []
    for r in ratios:
        ft = f0 * r * t_dwell
        att.append(F(T).exp(-ft * ft / (T(2.0) * sigma_sq)))
    a0 = att[0]
    if a0 > 1e-30:
        att = [a / a0 for a in att]
    return att


def onset_ramp_time(v, f0, T):                                 # hammer.rs:53-57
    periods = T(1.0) + T(1.0) * (T(1.0) - v)
    return max(periods * (T(1.0) / f0), T(0.002))


def rust_round(x):                                             # f64::round: half away from zero
    return np.floor(x + 0.5) if x >= 0 else -np.floor(-x + 0.5)


def lcg(s):
    return (s * 1664525 + 1013904223) & U32


# ------------------------------------------------------------------------------------------------ Voice::render_note
def render_note(midi, velocity, duration_s, sample_rate, T=np.float64, taps=None):
    """Voice::render_note (voice.rs:191-221): seed note * 2654435761, MLP corrections off, chunks of 1024 (immaterial: nothing in
    Voice::render depends on the chunking).  Returns a float64 array of floor(duration * sr) samples."""
    old = np.seterr(all="ignore")
    try:
        return _render_note(int(midi), T(velocity), T(duration_s), T(sample_rate), T, taps)
    finally:
        np.seterr(**old)


def _render_note(midi, v, dur, sr, T, taps):
    seed = (midi * 2654435761) & U32
    # ---- Voice::note_on (voice.rs:28-142), MlpCorrections::identity()
    f0_nom, ratios, mode_amps, decay_db = note_params(midi, T)
    f0 = f0_nom * freq_detune(midi, T)
    dwell = dwell_attenuation(v, f0, ratios, T)
    onset_time = onset_ramp_time(v, f0, T)
    offs = mode_amplitude_offsets(midi, T)
    amps = [mode_amps[i] * dwell[i] * offs[i] for i in range(NUM_MODES)]
    vel_scale = F(T).pow(velocity_scurve(v, T), velocity_exponent(midi, T))
    amps = [a * vel_scale for a in amps]
    ds = pickup_displacement_scale(midi, T) * T(1.0)
    gain = output_scale(midi, v, T) * T(1.0)
    # ---- ModalReed::new (reed.rs:108-182)
    dt = T(1.0) / sr
    revert = F(T).exp(-dt / T(0.020))
    diffusion = T(0.0004) * F(T).sqrt(T(1.0) - revert * revert)
    js = max(seed, 1)
    half = T(4294967295.0) / T(2.0)
    drift = []
    tau = T(6.283185307179586476925286766559)                  # std::f64::consts::TAU rounded into T
    pi = T(3.14159265358979323846264338327950288)
    for _ in range(NUM_MODES):
        js = lcg(js)
        u1 = T(js >> 1) / half
        js = lcg(js)
        u2 = T(js >> 1) / half
        r = F(T).sqrt(T(-2.0) * F(T).log(max(u1, T(1e-30))))
        drift.append(T(0.0004) * r * F(T).cos(tau * u2))
    s = [T(0.0)] * NUM_MODES
    c = [T(1.0)] * NUM_MODES
    phase_inc, cos_inc, sin_inc, decay_mult = [], [], [], []
    for i in range(NUM_MODES):
        freq = f0 * ratios[i]
        pinc = tau * freq / sr
        phase_inc.append(pinc)
        cos_inc.append(F(T).cos(pinc))
        sin_inc.append(F(T).sin(pinc))
        decay_mult.append(F(T).exp(-((decay_db[i] / T(8.686)) / sr)))
    env = [T(1.0)] * NUM_MODES
    ramp_samps = int(rust_round(onset_time * sr))
    ramp_inc = pi / T(ramp_samps) if ramp_samps > 0 else T(0.0)
    shape = T(1.0) + (T(1.0) - v)
    # ---- AttackNoise::new (hammer.rs:122-143) + Biquad::bandpass (filters.rs:15-21; melange-primitives, restated from its contract:
    # RBJ cookbook, constant skirt gain, coefficients normalised by a0, Direct Form II Transposed -- SURVEY.md 8c)
    namp = T(0.025) * v * v
    ndecay = F(T).exp(T(-1.0) / (T(0.003) * sr))
    nrem = int(T(0.015) * sr)
    nfade = 16
    center = min(max(f0 * T(5.0), T(200.0)), T(2000.0))
    w0 = tau * center / sr
    alpha = F(T).sin(w0) / (T(2.0) * T(0.7))
    a0 = T(1.0) + alpha
    b0, b1, b2 = (F(T).sin(w0) / T(2.0)) / a0, T(0.0) / a0, (-F(T).sin(w0) / T(2.0)) / a0
    a1, a2 = (T(-2.0) * F(T).cos(w0)) / a0, (T(1.0) - alpha) / a0
    z1 = z2 = T(0.0)
    nrng = seed
    # ---- Pickup::new (pickup.rs:30-33, 108-116)
    beta = dt / (T(2.0) * (T(287.0e3) * T(240.0e-12)))
    q = T(1.0)
    n_out = int(dur * sr)
    out = np.zeros(n_out, dtype=np.float64)
    reed_tap = np.zeros(n_out, dtype=np.float64) if taps is not None else None
    sample = 0
    for k in range(n_out):
        # ModalReed::render, one sample (reed.rs:223-305; no damper: render_note never releases)
        if sample < ramp_samps:
            cosine = T(0.5) * (T(1.0) - F(T).cos(T(sample) * ramp_inc))
            if shape <= 1.001:
                onset = cosine
            elif shape >= 1.999:
                onset = cosine * cosine
            else:
                onset = F(T).pow(cosine, shape)
        else:
            onset = T(1.0)
        if sample & 15 == 0:
            for m in range(NUM_MODES):
                js = lcg(js)
                u = T(js >> 1) / half
                noise = (u * T(2.0) - T(1.0)) * T(1.7320508080)
                drift[m] = revert * drift[m] + diffusion * noise
        acc = T(0.0)
        for m in range(NUM_MODES):
            acc = acc + amps[m] * s[m] * onset * env[m]
            dp = drift[m] * phase_inc[m]
            ci = cos_inc[m] - dp * sin_inc[m]
            si = sin_inc[m] + dp * cos_inc[m]
            s_new = s[m] * ci + c[m] * si
            c_new = c[m] * ci - s[m] * si
            s[m], c[m] = s_new, c_new
            env[m] = env[m] * decay_mult[m]
        if sample & 1023 == 0 and sample > 0:
            for m in range(NUM_MODES):
                r_inv = T(1.0) / F(T).sqrt(s[m] * s[m] + c[m] * c[m])
                s[m] = s[m] * r_inv
                c[m] = c[m] * r_inv
        x = T(0.0) + acc                                        # `*sample += sum` into the cleared buffer (voice.rs:163)
        sample += 1
        if reed_tap is not None:
            reed_tap[k] = np.float64(x)
        # AttackNoise::render (hammer.rs:150-179)
        if nrem > 0:
            if nfade > 0:
                t = T(16 - nfade) / T(16.0)
                nfade -= 1
                e = T(0.5) * (T(1.0) - F(T).cos(pi * t))
            else:
                e = T(1.0)
            nrng = lcg(nrng)
            signed = nrng - (1 << 32) if nrng & 0x80000000 else nrng
            nz = T(signed) / T(2147483647.0)
            y = b0 * nz + z1
            z1 = b1 * nz - a1 * y + z2
            z2 = b2 * nz - a2 * y
            x = x + namp * e * y
            namp = namp * ndecay
            nrem -= 1
        # Pickup::process (pickup.rs:130-149) with pickup_soft_saturate (:72-80)
        y = x * ds
        ay = abs(y)
        if not (ay < 0.94):
            rng_ = T(0.98) - T(0.94)
            y = np.copysign(T(0.94) + rng_ * F(T).tanh((ay - T(0.94)) / rng_), y)
        omy = T(1.0) - y
        al = beta * omy
        q = (q * (T(1.0) - al) + T(2.0) * beta) / (T(1.0) + al)
        out[k] = np.float64(((q * omy - T(1.0)) * T(1.8375)) * gain)
    if taps is not None:
        taps["reed"] = reed_tap
        taps["params"] = {"f0": float(f0), "ratios": [float(r) for r in ratios], "amps": [float(a) for a in amps],
                          "decay_db": [float(d) for d in decay_db], "ds": float(ds), "gain": float(gain), "onset_samples": ramp_samps,
                          "onset_exp": float(shape)}
    return out
