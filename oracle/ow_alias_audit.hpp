// ORACLE (test infrastructure, CPU): restatement of the click-band alias detector,
//   crates/openwurli-dsp/src/alias_audit.rs:25-57    stimulus constants
//   crates/openwurli-dsp/src/alias_audit.rs:95-160   run_with_note / run_sweep / render_stimulus
//   crates/openwurli-dsp/src/alias_audit.rs:163-282  analyze, plateau_metric, dft_magnitude, refine_f0, bandpass_rms
// Pinned by the reference's own unit tests (alias_audit.rs:293-361) and its regression gate against
// tests/baselines/alias_audit_v0_5_1.json (tests/alias_audit_regression.rs:29-30, 59-127); see tests/test_oracle_kat.py.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <vector>
#include "ow_engine.hpp"

namespace owo {

constexpr uint8_t AUDIT_STIMULUS_NOTE = 84;            // alias_audit.rs:28
constexpr uint8_t AUDIT_STIMULUS_VELOCITY = 120;       // :30
constexpr double AUDIT_STIMULUS_VOLUME = 0.5;          // :32
constexpr double AUDIT_SAMPLE_RATE = 44100.0;          // :47
constexpr double AUDIT_RENDER_SECONDS = 1.5;           // :50
constexpr double AUDIT_ANALYZE_SECONDS = 0.5;          // :53
constexpr int AUDIT_NUM_HARMONICS = 12;                // :56
constexpr int AUDIT_PLATEAU_FIRST = 6, AUDIT_PLATEAU_LAST = 11;   // :58-60
constexpr double AUDIT_HF_LO_HZ = 5000.0, AUDIT_HF_HI_HZ = 18000.0;  // :62-64

struct AliasAuditResult {   // alias_audit.rs:68-93
    double f0_hz, h1_dbfs;
    double harmonic_db[AUDIT_NUM_HARMONICS], harmonic_dbc[AUDIT_NUM_HARMONICS];
    double max_step_up_db;
    uint32_t max_step_up_from_harmonic, pad;
    double hf_band_dbc;
};

inline double audit_dft_magnitude(const double* s, size_t len, double freq, double sr) {   // :229-240
    const double n = (double)len;
    double re = 0.0, im = 0.0;
    const double omega = 2.0 * PI_ * freq / sr;
    for (size_t i = 0; i < len; ++i) {
        const double phase = omega * (double)i;
        re += s[i] * std::cos(phase);
        im -= s[i] * std::sin(phase);
    }
    const double a = re / n, b = im / n;
    return 2.0 * std::sqrt(a * a + b * b);   // powi(2) is x*x
}

inline double audit_mag_to_db(double mag) { return mag > 0.0 ? 20.0 * std::log10(mag) : -200.0; }   // :242-248

inline double audit_refine_f0(const double* s, size_t len, double sr, double nominal) {   // :252-265
    double best_f = nominal, best_mag = audit_dft_magnitude(s, len, nominal, sr);
    double f = nominal - 5.0;
    while (f <= nominal + 5.0) {
        const double mag = audit_dft_magnitude(s, len, f, sr);
        if (mag > best_mag) { best_mag = mag; best_f = f; }
        f += 0.1;
    }
    return best_f;
}

inline double audit_bandpass_rms(const double* s, size_t len, double sr, double lo, double hi) {   // :270-282
    const double q = 0.70710678118654752440;   // FRAC_1_SQRT_2
    Biquad hp1 = Biquad::make(Biquad::HIGHPASS, lo, q, sr), hp2 = Biquad::make(Biquad::HIGHPASS, lo, q, sr);
    Biquad lp1 = Biquad::make(Biquad::LOWPASS, hi, q, sr), lp2 = Biquad::make(Biquad::LOWPASS, hi, q, sr);
    double sum_sq = 0.0;
    for (size_t i = 0; i < len; ++i) {
        const double y = lp2.process(lp1.process(hp2.process(hp1.process(s[i]))));
        sum_sq += y * y;
    }
    return std::sqrt(sum_sq / (double)len);
}

inline void audit_plateau_metric(const double* dbc, double* worst_out, uint32_t* from_out) {   // :213-227
    double worst = -INFINITY;
    uint32_t worst_from = AUDIT_PLATEAU_FIRST;
    for (int i = AUDIT_PLATEAU_FIRST - 1; i < AUDIT_PLATEAU_LAST - 1; ++i) {
        const double delta = dbc[i + 1] - dbc[i];
        if (delta > worst) { worst = delta; worst_from = (uint32_t)(i + 1); }
    }
    *worst_out = worst; *from_out = worst_from;
}

// :163-211.  Returns false where the reference asserts (signal shorter than the analysis window).
inline bool audit_analyze(const double* signal, size_t len, double sr, double nominal_f0, AliasAuditResult* r) {
    const size_t analyze_n = (size_t)(sr * AUDIT_ANALYZE_SECONDS);
    if (len < analyze_n) return false;
    const double* tail = signal + (len - analyze_n);
    const double f0 = audit_refine_f0(tail, analyze_n, sr, nominal_f0);
    const double h1 = audit_dft_magnitude(tail, analyze_n, f0, sr);
    for (int k = 0; k < AUDIT_NUM_HARMONICS; ++k) {
        const double mag = audit_dft_magnitude(tail, analyze_n, (double)(k + 1) * f0, sr);
        r->harmonic_db[k] = audit_mag_to_db(mag);
        r->harmonic_dbc[k] = h1 > 0.0 ? 20.0 * std::log10(mag / h1) : -200.0;
    }
    r->harmonic_dbc[0] = 0.0;
    audit_plateau_metric(r->harmonic_dbc, &r->max_step_up_db, &r->max_step_up_from_harmonic);
    const double hf = audit_bandpass_rms(tail, analyze_n, sr, AUDIT_HF_LO_HZ, AUDIT_HF_HI_HZ);
    r->f0_hz = f0;
    r->h1_dbfs = audit_mag_to_db(h1);
    r->hf_band_dbc = h1 > 0.0 ? 20.0 * std::log10(hf / h1) : -200.0;
    r->pad = 0;
    return true;
}

inline double audit_midi_note_hz(int note) { return 440.0 * std::pow(2.0, ((double)note - 69.0) / 12.0); }   // :284-287

inline std::vector<double> audit_render_stimulus(int note, int velocity, int preamp_kind = 0) {   // :127-160
    const double sr = AUDIT_SAMPLE_RATE;
    WurliEngine eng(sr, preamp_kind);
    eng.ensure_buffer_capacity(1024);
    eng.set_volume(AUDIT_STIMULUS_VOLUME);
    eng.set_tremolo_depth(0.0);
    eng.set_speaker_character(0.0);
    eng.set_mlp_enabled(true);
    eng.set_noise_enabled(false);
    std::vector<float> buf(1024, 0.0f);
    for (int k = 0; k < 6; ++k) eng.render(buf.data(), 1024);
    eng.note_on(note, (float)velocity / 127.0f);
    const size_t total = (size_t)(sr * AUDIT_RENDER_SECONDS);
    std::vector<double> signal;
    signal.reserve(total);
    size_t pos = 0;
    while (pos < total) {
        const size_t len = std::min<size_t>(1024, total - pos);
        eng.render(buf.data(), len);
        for (size_t i = 0; i < len; ++i) signal.push_back((double)buf[i]);
        pos += len;
    }
    return signal;
}

}  // namespace owo
