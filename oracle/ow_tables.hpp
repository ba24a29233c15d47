// ORACLE (test infrastructure, CPU, f64) -- NOT part of the shipped product path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
//
// Restates the note-on parameter path of the reference (all citations into
// /root/reference/crates/openwurli-dsp/src/):
//   tables.rs:5-830      per-note physics tables
//   variation.rs:10-38   per-note detune / mode amplitude hash
//   hammer.rs:26-90      dwell filter, onset ramp time
//   mlp_correction.rs:21-140 (+ mlp_weights.rs as data)  note-on MLP
//
// Build with -O2 -ffp-contract=off -fno-fast-math: rustc never contracts a*b+c and
// never reassociates, and its f64 math methods call the platform libm (glibc here).
#pragma once
#include <cmath>
#include <cstdint>
#include <algorithm>
#include "../data/ow_gen_data.h"

namespace owo {

constexpr int NUM_MODES = 7;     // tables.rs:5
constexpr int MIDI_LO = 33;      // tables.rs:6
constexpr int MIDI_HI = 96;      // tables.rs:7

// Rust f64::clamp: NaN propagates, otherwise saturate (core::f64::clamp).
inline double rclamp(double x, double lo, double hi) {
    if (x < lo) return lo;
    if (x > hi) return hi;
    return x;
}
// Rust `as u32` / `as usize` / `as u64`: truncate toward zero, saturate, NaN -> 0.
inline uint32_t as_u32(double x) {
    if (!(x == x)) return 0;
    if (x <= 0.0) return 0;
    if (x >= 4294967295.0) return 4294967295u;
    return (uint32_t)x;
}
inline uint64_t as_u64(double x) {
    if (!(x == x)) return 0;
    if (x <= 0.0) return 0;
    if (x >= 18446744073709551615.0) return UINT64_MAX;
    return (uint64_t)x;
}

// tables.rs:33-34
static const double BASE_MODE_AMPLITUDES[NUM_MODES] = {1.0, 0.005, 0.0035, 0.0018, 0.0011, 0.0007, 0.0005};

// tables.rs:37-39
inline double midi_to_freq(int midi) { return 440.0 * std::pow(2.0, ((double)midi - 69.0) / 12.0); }

// tables.rs:51-77
inline double tip_mass_ratio(int midi) {
    const double m = (double)midi;
    static const double ax[5] = {33.0, 52.0, 62.0, 74.0, 96.0};
    static const double ay[5] = {0.10, 0.00, 0.00, 0.02, 0.01};
    if (m <= ax[0]) return ay[0];
    if (m >= ax[4]) return ay[4];
    for (int i = 0; i < 4; ++i) {
        if (m <= ax[i + 1]) {
            double t = (m - ax[i]) / (ax[i + 1] - ax[i]);
            return ay[i] + t * (ay[i + 1] - ay[i]);
        }
    }
    return 0.0;
}

// tables.rs:85-143
inline void eigenvalues(double mu, double out[NUM_MODES]) {
    static const double mus[8] = {0.00, 0.01, 0.05, 0.10, 0.15, 0.20, 0.30, 0.50};
    static const double betas[8][NUM_MODES] = {
        {1.8751, 4.6941, 7.8548, 10.9955, 14.1372, 17.2788, 20.4204},
        {1.8584, 4.6849, 7.8504, 10.9930, 14.1356, 17.2776, 20.4195},
        {1.7920, 4.6477, 7.8316, 10.9830, 14.1288, 17.2726, 20.4158},
        {1.7227, 4.6024, 7.8077, 10.9700, 14.1198, 17.2660, 20.4110},
        {1.6625, 4.5618, 7.7859, 10.9580, 14.1114, 17.2598, 20.4065},
        {1.6097, 4.5254, 7.7659, 10.9470, 14.1036, 17.2540, 20.4023},
        {1.5201, 4.4620, 7.7310, 10.9280, 14.0894, 17.2434, 20.3946},
        {1.3853, 4.3601, 7.6745, 10.8970, 14.0650, 17.2252, 20.3814},
    };
    const double mc = rclamp(mu, 0.0, 0.50);
    int lo = 0;  // rposition(row.mu <= mc).unwrap_or(0)
    for (int i = 7; i >= 0; --i) {
        if (mus[i] <= mc) { lo = i; break; }
    }
    const int hi = std::min(lo + 1, 7);
    const double t = (mus[hi] > mus[lo]) ? (mc - mus[lo]) / (mus[hi] - mus[lo]) : 0.0;
    for (int i = 0; i < NUM_MODES; ++i) out[i] = betas[lo][i] + t * (betas[hi][i] - betas[lo][i]);
}

// tables.rs:149-153
inline void mode_ratios(double mu, double out[NUM_MODES]) {
    double b[NUM_MODES];
    eigenvalues(mu, b);
    const double b1sq = b[0] * b[0];
    for (int i = 0; i < NUM_MODES; ++i) out[i] = (b[i] * b[i]) / b1sq;
}

// tables.rs:161-169
inline double reed_length_mm(int midi) {
    const double n = rclamp((double)midi - 32.0, 1.0, 64.0);
    const double inches = (n <= 20.0) ? 3.0 - n / 20.0 : 2.0 - (n - 20.0) / 44.0;
    return inches * 25.4;
}

// tables.rs:183-213
inline void reed_blank_dims(int midi, double& w_mm, double& t_mm) {
    int reed = midi - 32;
    reed = std::max(1, std::min(64, reed));
    double width_inch;
    if (reed <= 14) width_inch = 0.151;
    else if (reed <= 20) width_inch = 0.127;
    else if (reed <= 42) width_inch = 0.121;
    else if (reed <= 50) width_inch = 0.111;
    else width_inch = 0.098;
    double thick_inch;
    if (reed <= 16) thick_inch = 0.026;
    else if (reed <= 26) {
        double t = ((double)reed - 16.0) / 10.0;
        thick_inch = 0.026 + t * (0.034 - 0.026);
    } else thick_inch = 0.034;
    w_mm = width_inch * 25.4;
    t_mm = thick_inch * 25.4;
}

// tables.rs:221-225
inline double reed_compliance(int midi) {
    const double l = reed_length_mm(midi);
    double w, t;
    reed_blank_dims(midi, w, t);
    return (l * l * l) / (w * t * t * t);
}

// tables.rs:252-288 (CalibrationConfig::default())
constexpr double DS_AT_C4 = 0.85;
constexpr double DS_EXPONENT = 0.75;
constexpr double DS_CLAMP_LO = 0.02, DS_CLAMP_HI = 0.95;
constexpr double TARGET_DB = -35.0;
constexpr double VOICING_SLOPE = -0.04;

inline double pickup_displacement_scale(int midi) {
    const double c = reed_compliance(midi);
    const double c_ref = reed_compliance(60);
    const double ds = DS_AT_C4 * std::pow(c / c_ref, DS_EXPONENT);
    return rclamp(ds, DS_CLAMP_LO, DS_CLAMP_HI);
}

// tables.rs:295-299
inline double mode_shape(double beta, double xi) {
    const double sigma = (std::cosh(beta) + std::cos(beta)) / (std::sinh(beta) + std::sin(beta));
    const double bx = beta * xi;
    return std::cosh(bx) - std::cos(bx) - sigma * (std::sinh(bx) - std::sin(bx));
}

constexpr double PLATE_ACTIVE_LENGTH_MM = 6.0;  // tables.rs:306

// tables.rs:324-370
inline void spatial_coupling_coefficients(double mu, double reed_len_mm, double out[NUM_MODES]) {
    double betas[NUM_MODES];
    eigenvalues(mu, betas);
    const double ell = rclamp(PLATE_ACTIVE_LENGTH_MM / reed_len_mm, 0.0, 1.0);
    double kraw[NUM_MODES];
    const int NS = 32;
    const double xi_start = 1.0 - ell;
    for (int mode = 0; mode < NUM_MODES; ++mode) {
        const double beta = betas[mode];
        const double tip = mode_shape(beta, 1.0);
        if (std::fabs(tip) < 1e-30 || ell < 1e-12) { kraw[mode] = 1.0; continue; }
        const double h = ell / (double)NS;
        double sum = mode_shape(beta, xi_start) + mode_shape(beta, 1.0);
        for (int j = 1; j < NS; ++j) {
            const double xi = xi_start + (double)j * h;
            const double coeff = (j % 2 == 1) ? 4.0 : 2.0;
            sum += coeff * mode_shape(beta, xi);
        }
        const double integral = sum * h / 3.0;
        const double k = std::fabs(integral / (ell * tip));
        kraw[mode] = rclamp(k, 0.0, 1.0);
    }
    const double k1 = kraw[0];
    if (k1 > 1e-30) {
        for (int i = 0; i < NUM_MODES; ++i) out[i] = rclamp(kraw[i] / k1, 0.0, 1.0);
    } else {
        for (int i = 0; i < NUM_MODES; ++i) out[i] = 1.0;
    }
}

// tables.rs:391-396
inline double fundamental_decay_rate(int midi) {
    const double f = midi_to_freq(midi);
    return std::fmax(0.005 * std::pow(f, 1.22), 3.0);
}
// tables.rs:418-422
inline void mode_decay_rates(int midi, const double ratios[NUM_MODES], double out[NUM_MODES]) {
    const double base = fundamental_decay_rate(midi);
    for (int i = 0; i < NUM_MODES; ++i) out[i] = base * ratios[i] * ratios[i];
}

// tables.rs:424-441
inline double pickup_rms_proxy(double ds, double f0, double fc) {
    if (ds < 1e-10) return 0.0;
    const double r = (1.0 - std::sqrt(1.0 - ds * ds)) / ds;
    const double inv_sqrt = 1.0 / std::sqrt(1.0 - ds * ds);
    double sum_sq = 0.0;
    double r_n = r;
    for (int n = 1; n <= 8; ++n) {
        const double cn = 2.0 * r_n * inv_sqrt;
        const double nf = (double)n * f0;
        const double hpf_n = nf / std::sqrt(nf * nf + fc * fc);
        sum_sq += (cn * hpf_n) * (cn * hpf_n);
        r_n *= r;
    }
    return std::sqrt(sum_sq);
}

// tables.rs:443-481
inline double register_trim_db(int midi) {
    static const double ax[13] = {36, 40, 44, 48, 52, 56, 60, 64, 68, 72, 76, 80, 84};
    static const double ay[13] = {-1.3, 0.0, -1.3, 0.7, 0.2, -1.0, 0.0, 0.9, 1.2, 0.0, 1.8, 2.4, 3.6};
    const double m = (double)midi;
    if (m <= ax[0]) return ay[0];
    if (m >= ax[12]) return ay[12];
    for (int i = 0; i < 12; ++i) {
        if (m <= ax[i + 1]) {
            const double t = (m - ax[i]) / (ax[i + 1] - ax[i]);
            return ay[i] + t * (ay[i + 1] - ay[i]);
        }
    }
    return 0.0;
}

constexpr double POST_SPEAKER_GAIN = 7.498942093324558;  // tables.rs:485
constexpr double FIXED_CIRCUIT_DRIVE = 0.25;             // tables.rs:487

// tables.rs:535-554
inline double velocity_exponent(int midi) {
    const double m = (double)midi;
    const double center = 62.0, sigma = 15.0, max_exp = 1.7, treble_min = 1.3, bass_min = 0.55;
    const double z = (m - center) / sigma;
    const double t = std::exp(-0.5 * (z * z));
    const double min_exp = (m < center) ? bass_min : treble_min;
    return min_exp + t * (max_exp - min_exp);
}
// tables.rs:556-562
inline double velocity_scurve(double velocity) {
    const double k = 1.5;
    const double s = 1.0 / (1.0 + std::exp(-k * (velocity - 0.5)));
    const double s0 = 1.0 / (1.0 + std::exp(k * 0.5));
    const double s1 = 1.0 / (1.0 + std::exp(-k * 0.5));
    return (s - s0) / (s1 - s0);
}

// tables.rs:489-533 (output_scale_with_config, default config)
inline double output_scale(int midi, double velocity_norm) {
    const double HPF_FC = 2312.0;
    const double ds = pickup_displacement_scale(midi);
    const double f0 = midi_to_freq(midi);
    const double scurve_v = velocity_scurve(velocity_norm);
    const double vel_scale = std::pow(scurve_v, velocity_exponent(midi));
    const double vel_scale_c4 = std::pow(scurve_v, velocity_exponent(60));
    const double effective_ds = std::fmax(ds * vel_scale, 1e-6);
    const double effective_ds_ref = std::fmax(DS_AT_C4 * vel_scale_c4, 1e-6);
    const double rms = pickup_rms_proxy(effective_ds, f0, HPF_FC);
    const double rms_ref = pickup_rms_proxy(effective_ds_ref, midi_to_freq(60), HPF_FC);
    const double flat_db = -20.0 * std::log10(rms / rms_ref);
    const double voicing_db = VOICING_SLOPE * std::fmax((double)midi - 60.0, 0.0);
    const double trim = register_trim_db(midi);
    const double vel_blend = std::pow(velocity_norm, 1.3);
    const double effective_trim = trim * vel_blend;
    return std::pow(10.0, (TARGET_DB + flat_db + voicing_db + effective_trim) / 20.0);
}

struct NoteParams {
    double fundamental_hz;
    double mode_ratios[NUM_MODES];
    double mode_amplitudes[NUM_MODES];
    double mode_decay_rates[NUM_MODES];
};

// tables.rs:804-830
inline NoteParams note_params(int midi) {
    NoteParams p;
    p.fundamental_hz = midi_to_freq(midi);
    const double mu = tip_mass_ratio(midi);
    mode_ratios(mu, p.mode_ratios);
    mode_decay_rates(midi, p.mode_ratios, p.mode_decay_rates);
    double coupling[NUM_MODES];
    spatial_coupling_coefficients(mu, reed_length_mm(midi), coupling);
    for (int i = 0; i < NUM_MODES; ++i) p.mode_amplitudes[i] = BASE_MODE_AMPLITUDES[i] * coupling[i];
    return p;
}

// ---------------------------------------------------------------- variation.rs
// variation.rs:10-19
inline double hash_f64(uint8_t midi, uint32_t seed) {
    uint32_t h = 2166136261u;
    h ^= (uint32_t)midi;
    h *= 16777619u;
    h ^= seed;
    h *= 16777619u;
    h ^= h >> 16;
    h *= 2654435769u;
    return (double)(h & 0x00FFFFFFu) / 16777216.0;
}
// variation.rs:26-29
inline double freq_detune(uint8_t midi) {
    const double r = hash_f64(midi, 0xDEADu) * 2.0 - 1.0;
    return 1.0 + r * 0.00173;
}
// variation.rs:33-38
inline void mode_amplitude_offsets(uint8_t midi, double out[NUM_MODES]) {
    for (int i = 0; i < NUM_MODES; ++i) {
        const double r = hash_f64(midi, 0xBEEFu + (uint32_t)i) * 2.0 - 1.0;
        out[i] = 1.0 + r * 0.08;
    }
}

// ------------------------------------------------------------------- hammer.rs
// hammer.rs:26-29
inline double dwell_time(double velocity, double f0) {
    const double cycles = 0.75 + 0.25 * (1.0 - velocity);
    return rclamp(cycles / f0, 0.0003, 0.020);
}
// hammer.rs:53-57
inline double onset_ramp_time(double velocity, double f0) {
    const double period_s = 1.0 / f0;
    const double periods = 1.0 + 1.0 * (1.0 - velocity);
    return std::fmax(periods * period_s, 0.002);
}
// hammer.rs:69-90
inline void dwell_attenuation(double velocity, double f0, const double ratios[NUM_MODES], double atten[NUM_MODES]) {
    const double t_dwell = dwell_time(velocity, f0);
    const double sigma_sq = 8.0 * 8.0;
    for (int i = 0; i < NUM_MODES; ++i) {
        const double ft = f0 * ratios[i] * t_dwell;
        atten[i] = std::exp(-ft * ft / (2.0 * sigma_sq));
    }
    const double a0 = atten[0];
    if (a0 > 1e-30) {
        for (int i = 0; i < NUM_MODES; ++i) atten[i] /= a0;
    }
}

// ----------------------------------------------------------- mlp_correction.rs
struct MlpCorrections {
    double freq_offsets_cents[5];
    double decay_offsets[5];
    double ds_correction;
};
// mlp_correction.rs:49-55
inline MlpCorrections mlp_identity() {
    MlpCorrections c;
    for (int i = 0; i < 5; ++i) { c.freq_offsets_cents[i] = 0.0; c.decay_offsets[i] = 1.0; }
    c.ds_correction = 1.0;
    return c;
}
// mlp_correction.rs:61-140
inline MlpCorrections mlp_infer(int midi_note, double velocity) {
    const double MIDI_MIN = 21.0, MIDI_MAX = 108.0;
    const double TRAIN_LO = 65.0, TRAIN_HI = 97.0, FADE = 12.0;
    const double midi = (double)midi_note;
    double fade;
    if (midi < TRAIN_LO) fade = rclamp((midi - (TRAIN_LO - FADE)) / FADE, 0.0, 1.0);
    else if (midi > TRAIN_HI) fade = rclamp(((TRAIN_HI + FADE) - midi) / FADE, 0.0, 1.0);
    else fade = 1.0;
    if (fade <= 0.0) return mlp_identity();

    const double in0 = rclamp((midi - MIDI_MIN) / (MIDI_MAX - MIDI_MIN), 0.0, 1.0);
    const double in1 = rclamp(velocity, 0.0, 1.0);
    const double input[2] = {in0, in1};
    double h1[16], h2[16], raw[11];
    for (int i = 0; i < 16; ++i) {
        double sum = MLP_B1[i];
        for (int j = 0; j < 2; ++j) sum += MLP_W1[i][j] * input[j];
        h1[i] = sum > 0.0 ? sum : 0.0;
    }
    for (int i = 0; i < 16; ++i) {
        double sum = MLP_B2[i];
        for (int j = 0; j < 16; ++j) sum += MLP_W2[i][j] * h1[j];
        h2[i] = sum > 0.0 ? sum : 0.0;
    }
    for (int i = 0; i < 11; ++i) {
        double sum = MLP_B3[i];
        for (int j = 0; j < 16; ++j) sum += MLP_W3[i][j] * h2[j];
        raw[i] = sum * MLP_TARGET_STDS[i] + MLP_TARGET_MEANS[i];
    }
    MlpCorrections c;
    for (int h = 0; h < 5; ++h) c.freq_offsets_cents[h] = rclamp(raw[h] * fade, -100.0, 100.0);
    for (int h = 0; h < 5; ++h) {
        const double rd = rclamp(raw[5 + h], 0.3, 3.0);
        c.decay_offsets[h] = 1.0 + (rd - 1.0) * fade;
    }
    const double raw_ds = rclamp(raw[10], 0.7, 1.2);
    c.ds_correction = 1.0 + (raw_ds - 1.0) * fade;
    return c;
}

}  // namespace owo
