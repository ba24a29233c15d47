// ORACLE (test infrastructure, CPU, f64) -- NOT part of the shipped product path.
//
// Restates the voice layer of the reference (citations into
// /root/reference/crates/openwurli-dsp/src/):
//   filters.rs:13-59   Biquad newtype over melange-primitives 0.1.0 (git hal0zer0/melange
//                      rev de9dc81d..., Cargo.lock:312-314) -- NOT vendored in the reference
//                      tree.  Restated from its documented contract ("Audio EQ Cookbook
//                      coefficients, Direct Form II Transposed", bandpass = constant skirt
//                      gain).  PARITY UNPINNED at this boundary: no reference test pins the
//                      coefficients numerically.
//   reed.rs:16-329     ModalReed
//   pickup.rs:30-154   Pickup
//   hammer.rs:108-198  AttackNoise
//   voice.rs:28-221    Voice
#pragma once
#include "ow_tables.hpp"
#include <vector>

#ifdef OW_ORACLE_VOICE_PERTURB
#define OW_VOICE_LIBM(x) ((x) * (1.0 + 2.2e-16))
#else
#define OW_VOICE_LIBM(x) (x)
#endif
namespace owo {

constexpr double PI_ = 3.14159265358979323846;   // std::f64::consts::PI
constexpr double TAU_ = 6.28318530717958647692;  // std::f64::consts::TAU

// ---------------------------------------------------------------- filters.rs
struct Biquad {
    double b0 = 1, b1 = 0, b2 = 0, a1 = 0, a2 = 0;
    double s1 = 0, s2 = 0;
    enum Kind { LOWPASS, HIGHPASS, BANDPASS };
    void set(Kind kind, double fc, double q, double sr) {  // RBJ cookbook; keeps state (filters.rs:39-49)
        const double w0 = 2.0 * PI_ * fc / sr;
        const double cw = std::cos(w0), sw = std::sin(w0);
        const double alpha = sw / (2.0 * q);
        const double a0 = 1.0 + alpha;
        double nb0, nb1, nb2;
        switch (kind) {
            case LOWPASS:  nb0 = (1.0 - cw) / 2.0; nb1 = 1.0 - cw;    nb2 = (1.0 - cw) / 2.0; break;
            case HIGHPASS: nb0 = (1.0 + cw) / 2.0; nb1 = -(1.0 + cw); nb2 = (1.0 + cw) / 2.0; break;
            default:       nb0 = sw / 2.0;         nb1 = 0.0;         nb2 = -sw / 2.0;        break;
        }
        b0 = nb0 / a0; b1 = nb1 / a0; b2 = nb2 / a0;
        a1 = (-2.0 * cw) / a0; a2 = (1.0 - alpha) / a0;
    }
    static Biquad make(Kind kind, double fc, double q, double sr) { Biquad b; b.set(kind, fc, q, sr); return b; }
    inline double process(double x) {  // DF-II transposed
        const double y = b0 * x + s1;
        s1 = b1 * x - a1 * y + s2;
        s2 = b2 * x - a2 * y;
        return y;
    }
    void reset() { s1 = 0; s2 = 0; }
};

// ------------------------------------------------------------------- reed.rs
constexpr double JITTER_SIGMA = 0.0004;   // reed.rs:21
constexpr double JITTER_TAU = 0.020;      // reed.rs:26
constexpr double SQRT_3 = 1.7320508080;   // reed.rs:30

struct Mode {  // reed.rs:44-67
    double s, c, cos_inc, sin_inc, phase_inc, amplitude, decay_mult, envelope, jitter_drift, damper_rate, damper_mult;
};

struct ModalReed {
    Mode modes[NUM_MODES];
    uint64_t sample = 0;
    uint64_t onset_ramp_samples = 0;
    double onset_ramp_inc = 0, onset_shape_exp = 1;
    bool damper_active = false;
    double damper_ramp_samples = 0, damper_release_count = 0;
    bool damper_ramp_done = false;
    uint32_t jitter_state = 1;
    double jitter_revert = 0, jitter_diffusion = 0;

    // reed.rs:108-182
    void init(double f0, const double ratios[NUM_MODES], const double amps[NUM_MODES], const double decay_db[NUM_MODES],
              double onset_time_s, double velocity, double sr, uint32_t seed) {
        const double dt = 1.0 / sr;
        jitter_revert = std::exp(-dt / JITTER_TAU);
        jitter_diffusion = JITTER_SIGMA * std::sqrt(1.0 - jitter_revert * jitter_revert);
        jitter_state = seed > 1u ? seed : 1u;
        double drift[NUM_MODES];
        for (int i = 0; i < NUM_MODES; ++i) {
            jitter_state = jitter_state * 1664525u + 1013904223u;
            const double u1 = (double)(jitter_state >> 1) / (4294967295.0 / 2.0);
            jitter_state = jitter_state * 1664525u + 1013904223u;
            const double u2 = (double)(jitter_state >> 1) / (4294967295.0 / 2.0);
            const double r = std::sqrt(-2.0 * std::log(std::fmax(u1, 1e-30)));
            drift[i] = JITTER_SIGMA * r * std::cos(TAU_ * u2);
        }
        for (int i = 0; i < NUM_MODES; ++i) {
            const double freq = f0 * ratios[i];
            const double phase_inc = TAU_ * freq / sr;
            const double alpha_nepers = decay_db[i] / 8.686;
            const double decay_per_sample = alpha_nepers / sr;
            Mode& m = modes[i];
            m.s = 0.0; m.c = 1.0;
            // OW_ORACLE_VOICE_PERTURB builds a second sensitivity variant: the three library calls behind a mode's rotation and decay return
            // their neighbour in the last place -- the reference under a different libm, seen from the voice path (tools/soak_parity.py --ulp-voice)
            m.cos_inc = OW_VOICE_LIBM(std::cos(phase_inc)); m.sin_inc = OW_VOICE_LIBM(std::sin(phase_inc));
            m.phase_inc = phase_inc; m.amplitude = amps[i];
            m.decay_mult = OW_VOICE_LIBM(std::exp(-decay_per_sample));
            m.envelope = 1.0; m.jitter_drift = drift[i];
            m.damper_rate = 0.0; m.damper_mult = 1.0;
        }
        const uint64_t ramp = as_u64(std::round(onset_time_s * sr));
        onset_ramp_samples = ramp;
        onset_ramp_inc = ramp > 0 ? PI_ / (double)ramp : 0.0;
        onset_shape_exp = 1.0 + (1.0 - velocity);
        sample = 0;
        damper_active = false; damper_ramp_samples = 0; damper_release_count = 0; damper_ramp_done = false;
    }

    // reed.rs:191-216
    void start_damper(int midi, double sr) {
        if (midi >= 92) return;
        const double base_rate = std::fmax(55.0 * std::pow(2.0, ((double)midi - 60.0) / 24.0), 0.5);
        double p3 = 1.0;  // 3^m, exact for m <= 6 (powi)
        for (int m = 0; m < NUM_MODES; ++m) {
            const double factor = std::fmin(base_rate * p3, 2000.0);
            modes[m].damper_rate = factor / sr;
            modes[m].damper_mult = std::exp(-modes[m].damper_rate);
            p3 *= 3.0;
        }
        const double ramp_time = midi < 48 ? 0.050 : (midi < 72 ? 0.025 : 0.008);
        damper_ramp_samples = ramp_time * sr;
        damper_active = true;
        damper_release_count = 0.0;
        damper_ramp_done = false;
    }

    // reed.rs:219-306 (additive)
    void render(double* out, size_t n) {
        const double revert = jitter_revert, diffusion = jitter_diffusion;
        for (size_t k = 0; k < n; ++k) {
            double sum = 0.0;
            if (damper_active) {
                damper_release_count += 1.0;
                const double t = damper_release_count, ramp = damper_ramp_samples;
                if (!damper_ramp_done) {
                    if (t > ramp) damper_ramp_done = true;
                    else {
                        for (int m = 0; m < NUM_MODES; ++m) {
                            const double inst_rate = modes[m].damper_rate * t / ramp;
                            modes[m].envelope *= std::exp(-inst_rate);
                        }
                    }
                }
                if (damper_ramp_done) {
                    for (int m = 0; m < NUM_MODES; ++m) modes[m].envelope *= modes[m].damper_mult;
                }
            }
            double onset;
            if (sample < onset_ramp_samples) {
                const double nn = (double)sample;
                const double cosine = 0.5 * (1.0 - std::cos(nn * onset_ramp_inc));
                if (onset_shape_exp <= 1.001) onset = cosine;
                else if (onset_shape_exp >= 1.999) onset = cosine * cosine;
                else onset = std::pow(cosine, onset_shape_exp);
            } else onset = 1.0;

            if ((sample & 15u) == 0) {
                for (int m = 0; m < NUM_MODES; ++m) {
                    jitter_state = jitter_state * 1664525u + 1013904223u;
                    const double u = (double)(jitter_state >> 1) / (4294967295.0 / 2.0);
                    const double noise = (u * 2.0 - 1.0) * SQRT_3;
                    modes[m].jitter_drift = revert * modes[m].jitter_drift + diffusion * noise;
                }
            }
            for (int m = 0; m < NUM_MODES; ++m) {
                Mode& md = modes[m];
                sum += md.amplitude * md.s * onset * md.envelope;
                const double delta_phase = md.jitter_drift * md.phase_inc;
                const double ci = md.cos_inc - delta_phase * md.sin_inc;
                const double si = md.sin_inc + delta_phase * md.cos_inc;
                const double s_new = md.s * ci + md.c * si;
                const double c_new = md.c * ci - md.s * si;
                md.s = s_new; md.c = c_new;
                md.envelope *= md.decay_mult;
            }
            if ((sample & 1023u) == 0 && sample > 0) {
                for (int m = 0; m < NUM_MODES; ++m) {
                    Mode& md = modes[m];
                    const double r_sq = md.s * md.s + md.c * md.c;
                    const double r_inv = 1.0 / std::sqrt(r_sq);
                    md.s *= r_inv; md.c *= r_inv;
                }
            }
            out[k] += sum;
            sample += 1;
        }
    }

    // reed.rs:309-314
    bool is_silent(double threshold_db) const {
        const double thr = std::pow(10.0, threshold_db / 20.0);
        for (int m = 0; m < NUM_MODES; ++m)
            if (!(std::fabs(modes[m].amplitude * modes[m].envelope) <= thr)) return false;
        return true;
    }
    double release_seconds(double sr) const { return damper_active ? damper_release_count / sr : 0.0; }
};

// ----------------------------------------------------------------- pickup.rs
constexpr double PICKUP_TAU = 287.0e3 * 240.0e-12;  // pickup.rs:34
constexpr double PICKUP_SENSITIVITY = 1.8375;       // pickup.rs:38
constexpr double PICKUP_MAX_Y = 0.98;               // pickup.rs:49
constexpr double PICKUP_KNEE_Y = 0.94;              // pickup.rs:56

// pickup.rs:72-80
inline double pickup_soft_saturate(double y) {
    const double ay = std::fabs(y);
    if (ay < PICKUP_KNEE_Y) return y;
    const double range = PICKUP_MAX_Y - PICKUP_KNEE_Y;
    const double sat = PICKUP_KNEE_Y + range * std::tanh((ay - PICKUP_KNEE_Y) / range);
    return std::copysign(sat, y);
}

struct Pickup {  // pickup.rs:88-154
    double q = 1.0, beta = 0, displacement_scale = 0.85;
    void init(double sr) {
        const double dt = 1.0 / sr;
        beta = dt / (2.0 * PICKUP_TAU);
        q = 1.0;
        displacement_scale = 0.85;
    }
    void process(double* buf, size_t n) {
        const double scale = displacement_scale, b = beta;
        for (size_t i = 0; i < n; ++i) {
            const double y = pickup_soft_saturate(buf[i] * scale);
            const double omy = 1.0 - y;
            const double alpha = b * omy;
            const double q_next = (q * (1.0 - alpha) + 2.0 * b) / (1.0 + alpha);
            q = q_next;
            buf[i] = (q_next * omy - 1.0) * PICKUP_SENSITIVITY;
        }
    }
};

// ---------------------------------------------------------- hammer.rs (noise)
struct AttackNoise {  // hammer.rs:108-198
    double amplitude = 0, decay_per_sample = 0;
    uint32_t remaining = 0, fade_in_remaining = 0;
    Biquad bpf;
    uint32_t rng_state = 0;
    void init(double velocity, double f0, double sr, uint32_t seed) {
        amplitude = 0.025 * velocity * velocity;
        const double tau = 0.003;
        decay_per_sample = std::exp(-1.0 / (tau * sr));
        remaining = as_u32(0.015 * sr);
        fade_in_remaining = 16;
        const double center = rclamp(f0 * 5.0, 200.0, 2000.0);
        bpf = Biquad::make(Biquad::BANDPASS, center, 0.7, sr);
        rng_state = seed;
    }
    bool is_done() const { return remaining == 0; }
    void render(double* out, size_t n) {
        const size_t count = std::min((size_t)remaining, n);
        double amp = amplitude;
        uint32_t fade_in = fade_in_remaining;
        for (size_t i = 0; i < count; ++i) {
            double env;
            if (fade_in > 0) {
                const uint32_t pos = 16u - fade_in;
                const double t = (double)pos / 16.0;
                fade_in -= 1;
                env = 0.5 * (1.0 - std::cos(PI_ * t));
            } else env = 1.0;
            rng_state = rng_state * 1664525u + 1013904223u;
            const double noise = (double)(int32_t)rng_state / 2147483647.0;
            const double filtered = bpf.process(noise);
            out[i] += amp * env * filtered;
            amp *= decay_per_sample;
        }
        amplitude = amp;
        fade_in_remaining = fade_in;
        remaining -= (uint32_t)count;
    }
};

// ------------------------------------------------------------------ voice.rs
struct Voice {
    ModalReed reed;
    Pickup pickup;
    AttackNoise noise;
    double post_pickup_gain = 0, sample_rate = 0;
    int midi_note = 0;

    // voice.rs:28-142
    void note_on(int midi, double velocity, double sr, uint32_t seed, bool mlp_enabled) {
        const NoteParams params = note_params(midi);
        const double f0d = params.fundamental_hz * freq_detune((uint8_t)midi);
        double dwell[NUM_MODES], offs[NUM_MODES], amps[NUM_MODES];
        dwell_attenuation(velocity, f0d, params.mode_ratios, dwell);
        const double onset_time = onset_ramp_time(velocity, f0d);
        mode_amplitude_offsets((uint8_t)midi, offs);
        for (int i = 0; i < NUM_MODES; ++i) amps[i] = params.mode_amplitudes[i] * dwell[i] * offs[i];
        const double vel_exp = velocity_exponent(midi);
        const double vel_scale = std::pow(velocity_scurve(velocity), vel_exp);
        for (int i = 0; i < NUM_MODES; ++i) amps[i] *= vel_scale;

        const MlpCorrections corr = mlp_enabled ? mlp_infer(midi, velocity) : mlp_identity();
        double ratios[NUM_MODES], decay[NUM_MODES];
        for (int i = 0; i < NUM_MODES; ++i) { ratios[i] = params.mode_ratios[i]; decay[i] = params.mode_decay_rates[i]; }
        for (int h = 0; h < 5; ++h) ratios[1 + h] *= std::pow(2.0, corr.freq_offsets_cents[h] / 1200.0);
        for (int h = 0; h < 5; ++h) decay[1 + h] /= corr.decay_offsets[h];
        const double corrected_ds = pickup_displacement_scale(midi) * corr.ds_correction;

        reed.init(f0d, ratios, amps, decay, onset_time, velocity, sr, seed);
        pickup.init(sr);
        pickup.displacement_scale = corrected_ds;
        noise.init(velocity, f0d, sr, seed);

        const double base_output_scale = output_scale(midi, velocity);
        const double base_ds = pickup_displacement_scale(midi);
        double comp = 1.0;
        if (std::fabs(corr.ds_correction - 1.0) > 1e-6) {
            const double f0 = midi_to_freq(midi);
            const double HPF_FC = 2312.0;
            const double pb = pickup_rms_proxy(base_ds, f0, HPF_FC);
            const double pc = pickup_rms_proxy(corrected_ds, f0, HPF_FC);
            comp = (pc > 1e-10) ? std::sqrt(pb / pc) : 1.0;
        }
        post_pickup_gain = base_output_scale * comp;
        sample_rate = sr;
        midi_note = midi;
    }
    void note_off() { reed.start_damper(midi_note, sample_rate); }  // voice.rs:156-158

    // voice.rs:162-179
    void render(double* out, size_t n) {
        for (size_t i = 0; i < n; ++i) out[i] = 0.0;
        reed.render(out, n);
        if (!noise.is_done()) noise.render(out, n);
        pickup.process(out, n);
        const double g = post_pickup_gain;
        for (size_t i = 0; i < n; ++i) out[i] *= g;
    }
    // voice.rs:183-188
    bool is_silent() const {
        if (reed.damper_active && reed.release_seconds(sample_rate) > 10.0) return true;
        return reed.is_silent(-80.0);
    }
};

// voice.rs:191-221
inline std::vector<double> render_note(int midi, double velocity, double dur_s, double sr) {
    const uint32_t seed = (uint32_t)midi * 2654435761u;
    Voice v;
    v.note_on(midi, velocity, sr, seed, false);
    const size_t n = (size_t)as_u64(dur_s * sr);
    std::vector<double> out(n, 0.0);
    for (size_t off = 0; off < n; off += 1024) v.render(out.data() + off, std::min((size_t)1024, n - off));
    return out;
}

}  // namespace owo
