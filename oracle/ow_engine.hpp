// ORACLE (test infrastructure, CPU, f64) -- NOT part of the shipped product path.
//
// Restates the engine layer of the reference (citations into /root/reference/):
//   crates/openwurli-dsp/src/engine.rs:24-130,194-680   WurliEngine (voice pool, smoothers, chain)
//   tools/preamp-bench/src/main.rs:371-549,941-957      `render` batch job semantics
#pragma once
#include "ow_chain.hpp"
#include "ow_tremolo.hpp"
#include "ow_melange.hpp"
#include "ow_power_amp.hpp"
#include <memory>

namespace owo {

constexpr int MAX_VOICES = 64;        // engine.rs:24
constexpr int MAX_BLOCK_SIZE = 8192;  // engine.rs:25

enum VoiceState { VS_FREE = 0, VS_HELD = 1, VS_SUSTAINED = 2, VS_RELEASING = 3 };  // engine.rs:30-37

struct VoiceSlot {  // engine.rs:39-62
    std::unique_ptr<Voice> voice, steal_voice;
    int state = VS_FREE;
    int midi_note = 0;
    uint64_t age = 0;
    uint32_t steal_fade = 0, steal_fade_len = 0;
};

struct LinearSmoother {  // engine.rs:67-130
    double current, target, step;
    uint32_t samples_remaining, ramp_samples;
    void init(double initial, uint32_t ramp) { current = target = initial; step = 0; samples_remaining = 0; ramp_samples = ramp; }
    void set_target(double t) {
        if (std::fabs(t - target) < 1e-9) return;
        target = t;
        const double delta = t - current;
        if (ramp_samples == 0) { current = t; samples_remaining = 0; return; }
        step = delta / (double)ramp_samples;
        samples_remaining = ramp_samples;
    }
    void snap_to(double v) { current = target = v; step = 0; samples_remaining = 0; }
    void set_ramp_samples(uint32_t r) {
        ramp_samples = r;
        if (samples_remaining > 0) {
            step = (target - current) / (double)std::max(r, 1u);
            samples_remaining = r;
        }
    }
    inline double next() {
        if (samples_remaining > 0) {
            current += step;
            samples_remaining -= 1;
            if (samples_remaining == 0) current = target;
        }
        return current;
    }
};

inline uint32_t ramp_samples_for_rate(double sr) { return std::max(as_u32(sr * 0.005), 1u); }  // engine.rs:677-680

struct WurliEngine {
    VoiceSlot voices[MAX_VOICES];
    uint64_t age_counter = 0;
    DkPreamp preamp;          // default solver (dk_preamp/mod.rs:14-15)
    MelangePreamp mel;        // `--features melange-preamp` solver (dk_preamp/melange_adapter.rs)
    int preamp_kind = 0;      // 0 = legacy 8-node, 1 = melange 12-node
    void pre_init(double sr) { if (preamp_kind) mel.init(sr); else preamp.init(sr); }
    void pre_reset() { if (preamp_kind) mel.reset(); else preamp.reset(); }
    void pre_set_ldr(double r) { if (preamp_kind) mel.set_ldr_resistance(r); else preamp.set_ldr_resistance(r); }
    double pre_process(double x) { return preamp_kind ? mel.process_sample(x) : preamp.process_sample(x); }
    Tremolo tremolo;
    Oversampler oversampler;
    PowerAmp power_amp;       // behavioural amp of the default build (`legacy-power-amp`)
    MelangePowerAmp mel_pa;   // melange 7-BJT amp of a `--no-default-features` build (power_amp.rs:279-465)
    int power_amp_kind = 0;   // 0 = behavioural, 1 = melange
    double* pa_tap = nullptr; // optional: power-amp output per chain-rate sample
    double pa_process(double x) { return power_amp_kind ? mel_pa.process(x) : power_amp.process(x); }
    void set_rail_sag(bool on) { if (power_amp_kind) mel_pa.set_rail_sag(on); }          // engine.rs:406-408
    bool rail_sag_enabled() const { return power_amp_kind ? mel_pa.rail_sag_on : false; } // :410-412
    Speaker speaker;
    std::vector<double> voice_buf, sum_buf, up_buf, out_buf;
    double sample_rate = 0, os_sample_rate = 0;
    bool oversample = true, sustain_held = false, mlp_enabled = true;
    LinearSmoother volume, tremolo_depth, speaker_character;
    uint64_t nan_guard_fires = 0;

    // engine.rs:194-229
    explicit WurliEngine(double sr, int kind = 0, int pa_kind = 0, int trem_kind = 0) {
        preamp_kind = kind;
        power_amp_kind = pa_kind;
        tremolo.kind = trem_kind;                      // `--features legacy-tremolo` build
        oversample = sr < 88200.0;
        const double os_sr = oversample ? sr * 2.0 : sr;
        const uint32_t ramp = ramp_samples_for_rate(sr);
        pre_init(os_sr);
        tremolo.init(0.5, os_sr);
        if (power_amp_kind) mel_pa.init(os_sr);        // PowerAmp::new_at_sample_rate(os_sr), engine.rs:213
        speaker.init(sr);
        voice_buf.assign(MAX_BLOCK_SIZE, 0.0);
        sum_buf.assign(MAX_BLOCK_SIZE, 0.0);
        up_buf.assign(MAX_BLOCK_SIZE * 2, 0.0);
        out_buf.assign(MAX_BLOCK_SIZE, 0.0);
        sample_rate = sr;
        os_sample_rate = os_sr;
        volume.init(0.5, ramp);
        tremolo_depth.init(0.5, ramp);
        speaker_character.init(0.0, ramp);
    }

    // engine.rs:231-251
    void reset() {
        for (auto& s : voices) { s.state = VS_FREE; s.voice.reset(); s.steal_voice.reset(); s.steal_fade = 0; }
        pre_reset();
        tremolo.reset();
        oversampler.reset();
        if (power_amp_kind) mel_pa.reset();            // power_amp.reset() (a no-op on the behavioural amp)
        speaker.reset();
        age_counter = 0;
        sustain_held = false;
        volume.snap_to(volume.target);
        tremolo_depth.snap_to(tremolo_depth.target);
        speaker_character.snap_to(speaker_character.target);
        warm_up();
    }
    // engine.rs:261-270
    void warm_up() {
        float scratch[512];
        const size_t total = (size_t)as_u64(sample_rate * 0.6);
        size_t done = 0;
        while (done < total) {
            const size_t len = std::min((size_t)512, total - done);
            render(scratch, len);
            done += len;
        }
    }
    // engine.rs:272-286
    void set_sample_rate(double sr) {
        sample_rate = sr;
        oversample = sr < 88200.0;
        os_sample_rate = oversample ? sr * 2.0 : sr;
        pre_init(os_sample_rate);
        tremolo.init(tremolo_depth.target, os_sample_rate);
        oversampler.reset();
        if (power_amp_kind) mel_pa.init(os_sample_rate);   // PowerAmp::new_at_sample_rate (engine.rs:279): rail sag back to its default (on), last_good 0
        speaker.init(sr);
        const uint32_t ramp = ramp_samples_for_rate(sr);
        volume.set_ramp_samples(ramp);
        tremolo_depth.set_ramp_samples(ramp);
        speaker_character.set_ramp_samples(ramp);
        warm_up();
    }
    // engine.rs:288-295
    void ensure_buffer_capacity(size_t n) {
        if (sum_buf.size() < n) {
            voice_buf.resize(n, 0.0); sum_buf.resize(n, 0.0); up_buf.resize(n * 2, 0.0); out_buf.resize(n, 0.0);
        }
    }

    // engine.rs:569-590
    int allocate_voice() const {
        int best_idx = 0;
        uint64_t best = UINT64_MAX;
        for (int i = 0; i < MAX_VOICES; ++i) {
            const VoiceSlot& s = voices[i];
            uint64_t pr;
            switch (s.state) {
                case VS_FREE: return i;
                case VS_RELEASING: pr = s.age; break;
                case VS_SUSTAINED: pr = s.age + UINT64_MAX / 4; break;
                default: pr = s.age + UINT64_MAX / 2; break;
            }
            if (pr < best) { best = pr; best_idx = i; }
        }
        return best_idx;
    }

    // engine.rs:299-338
    void note_on(int note_in, float velocity) {
        const int note = std::max(MIDI_LO, std::min(MIDI_HI, note_in));
        for (auto& s : voices) {
            if (s.state == VS_SUSTAINED && s.midi_note == note) {
                s.state = VS_RELEASING;
                if (s.voice) s.voice->note_off();
            }
        }
        const int idx = allocate_voice();
        VoiceSlot& slot = voices[idx];
        if (slot.state != VS_FREE) {
            const uint32_t fade = as_u32(sample_rate * 0.005);
            slot.steal_voice = std::move(slot.voice);
            slot.steal_fade = fade;
            slot.steal_fade_len = fade;
        }
        age_counter += 1;
        const uint32_t seed = (uint32_t)note * 2654435761u + (uint32_t)age_counter;
        slot.voice.reset(new Voice());
        slot.voice->note_on(note, (double)velocity, sample_rate, seed, mlp_enabled);
        slot.state = VS_HELD;
        slot.midi_note = note;
        slot.age = age_counter;
    }
    // engine.rs:340-359
    void note_off(int note_in) {
        const int note = std::max(MIDI_LO, std::min(MIDI_HI, note_in));
        int oldest = -1;
        for (int i = 0; i < MAX_VOICES; ++i) {
            if (voices[i].state == VS_HELD && voices[i].midi_note == note) {
                if (oldest < 0 || voices[i].age < voices[oldest].age) oldest = i;
            }
        }
        if (oldest >= 0) {
            if (sustain_held) voices[oldest].state = VS_SUSTAINED;
            else {
                voices[oldest].state = VS_RELEASING;
                if (voices[oldest].voice) voices[oldest].voice->note_off();
            }
        }
    }
    // engine.rs:361-374
    void set_sustain(bool held) {
        if (sustain_held && !held) {
            for (auto& s : voices) {
                if (s.state == VS_SUSTAINED) {
                    s.state = VS_RELEASING;
                    if (s.voice) s.voice->note_off();
                }
            }
        }
        sustain_held = held;
    }
    void set_volume(double v) { volume.set_target(v); }
    void set_tremolo_depth(double d) { tremolo_depth.set_target(d); }
    void set_speaker_character(double c) { speaker_character.set_target(c); }
    void set_mlp_enabled(bool on) { mlp_enabled = on; }
    // engine.rs:394-400: thermal noise of the melange preamp (no-ops on the legacy solver)
    void set_noise_enabled(bool on) { if (preamp_kind) mel.set_noise_enabled(on); }
    void set_noise_gain(double g) { if (preamp_kind) mel.set_thermal_gain(g); }
    void set_noise_seed(uint64_t seed) { if (preamp_kind) mel.set_noise_seed(seed); }   // extension (gen_preamp::set_seed, not reachable through WurliEngine)

    // engine.rs:425-462
    void render(float* out, size_t len) {
        if (len == 0) return;
        ensure_buffer_capacity(len);
        render_voices_to_preamp_out(0, len);
        for (size_t i = 0; i < len; ++i) {
            const double sc = speaker_character.next();
            speaker.set_character(sc);
            const double shaped = speaker.process(out_buf[i]);
            const double user_vol = volume.next();
            const double post_gain = shaped * POST_SPEAKER_GAIN * user_vol;
            const float sample = (float)post_gain;
            if (std::isfinite(sample)) out[i] = sample;
            else {
                pre_reset();
                oversampler.reset();
                if (power_amp_kind) mel_pa.reset();
                speaker.reset();
                out[i] = 0.0f;
            }
        }
        cleanup_voices();
    }

    // engine.rs:466-567.  `voice_sum_tap` (optional) receives sum_buf (pre-chain) for parity tests.
    double* voice_sum_tap = nullptr;
    double* preamp_tap = nullptr;   // optional: preamp output per chain-rate sample (before the power amp)
    double* r_tap = nullptr;        // optional: tremolo shunt R per chain-rate sample
    void render_voices_to_preamp_out(size_t offset, size_t len) {
        for (size_t i = 0; i < len; ++i) sum_buf[i] = 0.0;
        for (auto& slot : voices) {
            if (slot.state == VS_FREE && !slot.steal_voice) continue;
            if (slot.voice) {
                slot.voice->render(voice_buf.data(), len);
                for (size_t i = 0; i < len; ++i) sum_buf[i] += voice_buf[i];
            }
            if (slot.steal_voice) {
                slot.steal_voice->render(voice_buf.data(), len);
                const double fade_len = (double)slot.steal_fade_len;
                for (size_t i = 0; i < len; ++i) {
                    const uint32_t ii = (uint32_t)i;
                    const uint32_t remaining = slot.steal_fade > ii ? slot.steal_fade - ii : 0u;
                    const double gain = (double)remaining / fade_len;
                    sum_buf[i] += voice_buf[i] * gain;
                }
                const uint32_t l32 = (uint32_t)len;
                slot.steal_fade = slot.steal_fade > l32 ? slot.steal_fade - l32 : 0u;
                if (slot.steal_fade == 0) slot.steal_voice.reset();
            }
        }
        bool any_nan = false;
        for (size_t i = 0; i < len; ++i) if (!std::isfinite(sum_buf[i])) { any_nan = true; break; }
        if (any_nan) {
            nan_guard_fires += 1;
            for (size_t i = 0; i < len; ++i) sum_buf[i] = 0.0;
            for (auto& slot : voices) {
                if (slot.state == VS_FREE && !slot.steal_voice) continue;
                if (slot.voice) {
                    slot.voice->render(voice_buf.data(), len);
                    bool bad = false;
                    for (size_t i = 0; i < len; ++i) if (!std::isfinite(voice_buf[i])) bad = true;
                    if (bad) { slot.state = VS_FREE; slot.voice.reset(); }
                }
                if (slot.steal_voice) {
                    slot.steal_voice->render(voice_buf.data(), len);
                    bool bad = false;
                    for (size_t i = 0; i < len; ++i) if (!std::isfinite(voice_buf[i])) bad = true;
                    if (bad) { slot.steal_voice.reset(); slot.steal_fade = 0; }
                }
            }
        }
        if (voice_sum_tap) for (size_t i = 0; i < len; ++i) voice_sum_tap[i] = sum_buf[i];

        if (oversample) {
            oversampler.upsample_2x(sum_buf.data(), len, up_buf.data());
            for (size_t i = 0; i < len; ++i) {
                const double depth = tremolo_depth.next();
                tremolo.set_depth(depth);
                for (int j = 0; j < 2; ++j) {
                    const size_t idx = i * 2 + j;
                    const double r = tremolo.process();
                    pre_set_ldr(r);
                    const double pre = pre_process(up_buf[idx]);
                    if (preamp_tap) preamp_tap[idx] = pre;
                    if (r_tap) r_tap[idx] = r;
                    up_buf[idx] = pa_process(pre * FIXED_CIRCUIT_DRIVE);
                    if (pa_tap) pa_tap[idx] = up_buf[idx];
                }
            }
            oversampler.downsample_2x(up_buf.data(), out_buf.data() + offset, len);
        } else {
            for (size_t i = 0; i < len; ++i) {
                const double depth = tremolo_depth.next();
                tremolo.set_depth(depth);
                const double r = tremolo.process();
                pre_set_ldr(r);
                const double pre = pre_process(sum_buf[i]);
                if (preamp_tap) preamp_tap[i] = pre;
                if (r_tap) r_tap[i] = r;
                out_buf[offset + i] = pa_process(pre * FIXED_CIRCUIT_DRIVE);
                if (pa_tap) pa_tap[i] = out_buf[offset + i];
            }
        }
    }
    // engine.rs:592-602
    void cleanup_voices() {
        for (auto& s : voices) {
            if (s.state != VS_FREE && s.voice && s.voice->is_silent()) { s.state = VS_FREE; s.voice.reset(); }
        }
    }
    int count_state(int st) const { int c = 0; for (auto& s : voices) c += (s.state == st); return c; }
    int active_voice_count() const { return MAX_VOICES - count_state(VS_FREE); }
    int steal_voice_count() const { int c = 0; for (auto& s : voices) c += (s.steal_voice != nullptr); return c; }
};

// tools/preamp-bench/src/main.rs:371-549, every flag of `preamp-bench render` that changes samples.  Output: final_output f64
// (--normalize only scales what write_wav_24bit quantises, main.rs:512-523: see batch_normalize_scale).
struct BatchJobOpts {
    double volume = 0.60, speaker_char = 1.0, r_ldr = 1000000.0, tremolo_depth = 0.0;     // main.rs:375-378 defaults
    bool mlp = true, poweramp = true, no_preamp = false, no_attack_noise = false, no_rail_sag = false;
    bool has_displacement_scale = false;
    double displacement_scale = 0.30;
    int preamp_kind = 0;        // 0 legacy DkPreamp, 1 melange (cargo feature melange-preamp)
    int power_amp_kind = 0;     // 0 behavioural (default feature legacy-power-amp), 1 melange 7-BJT: PowerAmp::new() = 44.1 kHz whatever --sample-rate says
};
inline std::vector<double> batch_render_job_ex(int note, int velocity_u8, double duration, double sr, const BatchJobOpts& o) {
    const bool do_os = sr < 88200.0;
    const double preamp_sr = do_os ? sr * 2.0 : sr;
    const double vel_norm = (double)velocity_u8 / 127.0;
    const uint32_t seed = (uint32_t)note * 2654435761u;
    Voice voice;
    voice.note_on(note, vel_norm, sr, seed, o.mlp);
    if (o.has_displacement_scale) voice.pickup.displacement_scale = o.displacement_scale;     // main.rs:406-408, voice.rs:145-147
    if (o.no_attack_noise) voice.noise.remaining = 0;                                         // main.rs:409-411, hammer.rs:187-189
    const size_t n = (size_t)as_u64(duration * sr);
    std::vector<double> reed(n, 0.0);
    for (size_t off = 0; off < n; off += 1024) voice.render(reed.data() + off, std::min((size_t)1024, n - off));

    std::vector<double> pre(n, 0.0);
    if (o.no_preamp) {                                                                        // main.rs:425-427
        pre = reed;
    } else {
        DkPreamp legacy;
        MelangePreamp mel;
        if (o.preamp_kind) mel.init(preamp_sr); else legacy.init(preamp_sr);
        Tremolo trem;
        const bool use_trem = o.tremolo_depth > 0.0;                                          // main.rs:430-441
        if (use_trem) trem.init(o.tremolo_depth, preamp_sr);
        else if (o.preamp_kind) { mel.reset(); mel.set_ldr_resistance(o.r_ldr); }
        else { legacy.reset(); legacy.set_ldr_resistance(o.r_ldr); }
        auto step = [&](double x) -> double {
            if (use_trem) { const double r = trem.process(); if (o.preamp_kind) mel.set_ldr_resistance(r); else legacy.set_ldr_resistance(r); }
            return o.preamp_kind ? mel.process_sample(x) : legacy.process_sample(x);
        };
        if (do_os) {
            Oversampler os;
            for (size_t i = 0; i < n; ++i) {
                double up[2], proc[2], down[1];
                os.upsample_2x(&reed[i], 1, up);
                proc[0] = step(up[0]);
                proc[1] = step(up[1]);
                os.downsample_2x(proc, down, 1);
                pre[i] = down[0];
            }
        } else {
            for (size_t i = 0; i < n; ++i) pre[i] = step(reed[i]);
        }
    }
    PowerAmp pa;
    MelangePowerAmp mpa;
    if (o.power_amp_kind) { mpa.init(44100.0); if (o.no_rail_sag) mpa.set_rail_sag(false); }  // main.rs:480-483; power_amp.rs:321-323
    Speaker spk;
    spk.init(sr);
    spk.set_character(o.speaker_char);
    std::vector<double> out(n, 0.0);
    for (size_t i = 0; i < n; ++i) {
        const double att = pre[i] * o.volume * o.volume;
        const double amp = o.poweramp ? (o.power_amp_kind ? mpa.process(att) : pa.process(att)) : att;
        out[i] = spk.process(amp) * POST_SPEAKER_GAIN;
    }
    return out;
}
// --normalize (main.rs:505-511): the factor write_wav_24bit applies
inline double batch_normalize_scale(const double* x, size_t n) {
    double peak = 0.0;
    for (size_t i = 0; i < n; ++i) peak = std::fmax(peak, std::fabs(x[i]));
    return peak > 0.7 ? 0.7 / peak : 1.0;
}
inline std::vector<double> batch_render_job(int note, int velocity_u8, double duration, double sr, double volume,
                                            double speaker_char, double r_ldr, bool mlp, bool poweramp, int preamp_kind = 0) {
    BatchJobOpts o;
    o.volume = volume; o.speaker_char = speaker_char; o.r_ldr = r_ldr; o.mlp = mlp; o.poweramp = poweramp; o.preamp_kind = preamp_kind;
    return batch_render_job_ex(note, velocity_u8, duration, sr, o);
}

}  // namespace owo
