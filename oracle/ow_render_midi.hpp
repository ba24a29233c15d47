// ORACLE (test infrastructure, CPU): restatement of `preamp-bench render-midi`,
//   tools/preamp-bench/src/main.rs:1603-1923   cmd_render_midi: SMF -> timed events -> 64-sample chunks with its own 64-slot voice
//                                              manager (first free slot, else oldest; no steal crossfade; pedal defers note-offs)
//                                              -> oversampled preamp at a static 1 Mohm LDR -> volume^2 -> power amp at BASE
//                                              rate -> speaker -> POST_SPEAKER_GAIN
// The SMF reader restates what the reference uses of midly 0.5.3 (Cargo.lock; not vendored): header timing, per-track delta times,
// running status, tempo meta events, note on/off and controller 64.  No golden vectors exist for this command: the DSP blocks it
// chains are the pinned ones of ow_voice.hpp / ow_chain.hpp, the voice manager and the parser are checked on hand-built cases
// (tests/test_oracle_kat.py) -- "parity unpinned" for the container format.
#pragma once
#include <algorithm>
#include <cstdint>
#include <stdexcept>
#include <vector>
#include "ow_engine.hpp"

namespace owo {

constexpr double MIDI_BASE_SR = 44100.0;   // main.rs:27

struct TimedEvent {        // main.rs:1639-1649
    double time_s;
    uint8_t type;          // 0 NoteOn, 1 NoteOff, 2 Pedal
    uint8_t note;          // key (NoteOn/NoteOff)
    uint8_t value;         // velocity (NoteOn) / 1 = pedal down (Pedal)
};

// main.rs:1627-1708.  track_filter < 0: all tracks.  Throws on malformed data / SMPTE timing.
inline std::vector<TimedEvent> smf_events(const uint8_t* d, size_t n, int track_filter) {
    auto need = [&](size_t pos, size_t k) { if (pos + k > n) throw std::runtime_error("truncated MIDI file"); };
    auto be32 = [&](size_t p) { return ((uint32_t)d[p] << 24) | ((uint32_t)d[p + 1] << 16) | ((uint32_t)d[p + 2] << 8) | d[p + 3]; };
    need(0, 14);
    if (!(d[0] == 'M' && d[1] == 'T' && d[2] == 'h' && d[3] == 'd')) throw std::runtime_error("not a MIDI file");
    const uint32_t hlen = be32(4);
    if (hlen < 6) throw std::runtime_error("bad header");
    const uint16_t division = (uint16_t)((d[12] << 8) | d[13]);
    if (division & 0x8000) throw std::runtime_error("Only metrical (ticks per beat) MIDI timing is supported");   // main.rs:1630-1636
    const double tpb = (double)division;
    size_t pos = 8 + hlen;
    std::vector<TimedEvent> ev;
    int track_idx = 0;
    while (pos + 8 <= n) {
        const bool is_track = d[pos] == 'M' && d[pos + 1] == 'T' && d[pos + 2] == 'r' && d[pos + 3] == 'k';
        const uint32_t len = be32(pos + 4);
        pos += 8;
        need(pos, len);
        if (!is_track) { pos += len; continue; }
        const size_t end = pos + len;
        double tempo = 500000.0, time_s = 0.0;              // per track, main.rs:1653-1654
        const bool emit = track_filter < 0 || track_filter == track_idx;
        uint8_t running = 0;
        size_t p = pos;
        auto vlq = [&]() {
            uint32_t v = 0;
            for (int k = 0; k < 4; ++k) {
                if (p >= end) throw std::runtime_error("truncated track");
                const uint8_t b = d[p++];
                v = (v << 7) | (b & 0x7F);
                if (!(b & 0x80)) return v;
            }
            throw std::runtime_error("bad variable-length quantity");
        };
        while (p < end) {
            const uint32_t delta = vlq();
            time_s += ((double)(uint64_t)delta / tpb) * (tempo / 1000000.0);   // main.rs:1661-1663
            if (p >= end) throw std::runtime_error("truncated track");
            uint8_t status = d[p];
            if (status & 0x80) { ++p; } else { if (!running) throw std::runtime_error("data byte without running status"); status = running; }
            if (status == 0xFF) {                           // meta
                if (p >= end) throw std::runtime_error("truncated track");
                const uint8_t type = d[p++];
                const uint32_t l = vlq();
                if (p + l > end) throw std::runtime_error("truncated track");
                if (type == 0x51 && l == 3) tempo = (double)(((uint32_t)d[p] << 16) | ((uint32_t)d[p + 1] << 8) | d[p + 2]);   // :1666-1668
                p += l;
                running = 0;
            } else if (status == 0xF0 || status == 0xF7) {  // sysex / escape
                const uint32_t l = vlq();
                if (p + l > end) throw std::runtime_error("truncated track");
                p += l;
                running = 0;
            } else if (status >= 0xF0) {
                throw std::runtime_error("unexpected system message in track");
            } else {
                running = status;
                const uint8_t hi = status & 0xF0;
                const int nd = (hi == 0xC0 || hi == 0xD0) ? 1 : 2;
                if (p + nd > end) throw std::runtime_error("truncated track");
                const uint8_t a = d[p] & 0x7F, b = nd == 2 ? (d[p + 1] & 0x7F) : 0;
                p += nd;
                if (!emit) continue;
                if (hi == 0x90) ev.push_back({time_s, (uint8_t)(b == 0 ? 1 : 0), a, b});          // :1670-1686
                else if (hi == 0x80) ev.push_back({time_s, 1, a, 0});                              // :1687-1692
                else if (hi == 0xB0 && a == 64) ev.push_back({time_s, 2, 0, (uint8_t)(b >= 64)});  // :1693-1703
            }
        }
        pos = end;
        ++track_idx;
    }
    return ev;
}

struct MidiRenderStats { uint64_t note_ons, peak_polyphony; };

// main.rs:1711-1891.  Returns the f64 output (empty when there are no events, where the reference prints and returns).
// preamp_kind / power_amp_kind: which DkPreamp / PowerAmp the command was built with (cargo features melange-preamp, legacy-power-amp)
inline std::vector<double> render_midi(std::vector<TimedEvent> events, double volume, double speaker_char, bool no_poweramp,
                                       double tail_seconds, MidiRenderStats* stats = nullptr, int preamp_kind = 0, int power_amp_kind = 0,
                                       bool no_rail_sag = false) {
    std::stable_sort(events.begin(), events.end(), [](const TimedEvent& a, const TimedEvent& b) { return a.time_s < b.time_s; });   // :1712
    if (stats) { stats->note_ons = 0; stats->peak_polyphony = 0; }
    if (events.empty()) return {};
    const double last_event_time = events.back().time_s;
    const double total_duration = last_event_time + tail_seconds;
    const size_t total_samples = (size_t)as_u64(total_duration * MIDI_BASE_SR);
    const int MAXV = 64;
    struct Slot { Voice voice; bool has = false, active = false; uint8_t midi_note = 0; uint64_t age = 0; };
    std::vector<Slot> voices(MAXV);
    uint64_t age_counter = 0;
    // main.rs:1752-1754: set_ldr_resistance(1 Mohm) and THEN reset(): the legacy solver keeps its resistance across reset(), the melange
    // adapter's reset() goes back to the settled clone at the nominal 100 kOhm (melange_adapter.rs:88-93) and stays there
    DkPreamp legacy;
    MelangePreamp mel;
    if (preamp_kind) { mel.init(MIDI_BASE_SR * 2.0); mel.set_ldr_resistance(1000000.0); mel.reset(); }
    else { legacy.init(MIDI_BASE_SR * 2.0); legacy.set_ldr_resistance(1000000.0); legacy.reset(); }
    struct { DkPreamp* l; MelangePreamp* m; double process_sample(double x) { return m ? m->process_sample(x) : l->process_sample(x); } }
        preamp{preamp_kind ? nullptr : &legacy, preamp_kind ? &mel : nullptr};
    Oversampler os;
    PowerAmp pa_behavioural;
    MelangePowerAmp pa_melange;
    if (power_amp_kind) { pa_melange.init(44100.0); if (no_rail_sag) pa_melange.set_rail_sag(false); }     // PowerAmp::new(), main.rs:1756
    struct { PowerAmp* b; MelangePowerAmp* m; double process(double x) { return m ? m->process(x) : b->process(x); } }
        power_amp{power_amp_kind ? nullptr : &pa_behavioural, power_amp_kind ? &pa_melange : nullptr};
    Speaker speaker;
    speaker.init(MIDI_BASE_SR);
    speaker.set_character(speaker_char);
    std::vector<double> output(total_samples, 0.0);
    size_t event_idx = 0;
    const size_t chunk_size = 64;
    double voice_buf[64], sum_buf[64], up_buf[128], down_buf[64];
    size_t sample_pos = 0;
    uint64_t peak_poly = 0, note_on_count = 0;
    bool pedal_down = false;
    std::vector<uint8_t> pedal_held;
    auto release = [&](uint8_t note) {   // oldest active slot playing `note`
        Slot* best = nullptr;
        for (auto& s : voices) if (s.active && s.midi_note == note && (!best || s.age < best->age)) best = &s;
        if (best && best->has) best->voice.note_off();
    };
    while (sample_pos < total_samples) {
        const size_t chunk_end = std::min(sample_pos + chunk_size, total_samples);
        const size_t len = chunk_end - sample_pos;
        const double chunk_time = (double)sample_pos / MIDI_BASE_SR;
        while (event_idx < events.size() && events[event_idx].time_s <= chunk_time) {
            const TimedEvent e = events[event_idx];
            if (e.type == 0) {
                const uint8_t note = std::min<uint8_t>(std::max<uint8_t>(e.note, MIDI_LO), MIDI_HI);
                const double vel = (double)e.value / 127.0;
                age_counter += 1;
                note_on_count += 1;
                int slot_idx = -1;
                for (int i = 0; i < MAXV; ++i) if (!voices[i].active) { slot_idx = i; break; }
                if (slot_idx < 0) {
                    slot_idx = 0;
                    for (int i = 1; i < MAXV; ++i) if (voices[i].age < voices[slot_idx].age) slot_idx = i;
                }
                const uint32_t seed = (uint32_t)note * 2654435761u + (uint32_t)age_counter;
                Slot& s = voices[slot_idx];
                s.voice = Voice();
                s.voice.note_on(note, vel, MIDI_BASE_SR, seed, true);
                s.has = true; s.active = true; s.midi_note = note; s.age = age_counter;
                uint64_t act = 0;
                for (auto& v : voices) act += v.active;
                peak_poly = std::max(peak_poly, act);
            } else if (e.type == 1) {
                const uint8_t note = std::min<uint8_t>(std::max<uint8_t>(e.note, MIDI_LO), MIDI_HI);
                if (pedal_down) pedal_held.push_back(note);
                else release(note);
            } else {
                pedal_down = e.value != 0;
                if (!pedal_down) {
                    for (uint8_t h : pedal_held) release(h);
                    pedal_held.clear();
                }
            }
            event_idx += 1;
        }
        for (auto& s : voices)
            if (s.active && s.has && s.voice.is_silent()) { s.active = false; s.has = false; }
        for (size_t i = 0; i < len; ++i) sum_buf[i] = 0.0;
        for (auto& s : voices) {
            if (!s.active || !s.has) continue;
            s.voice.render(voice_buf, len);
            for (size_t i = 0; i < len; ++i) sum_buf[i] += voice_buf[i];
        }
        os.upsample_2x(sum_buf, len, up_buf);
        for (size_t i = 0; i < 2 * len; ++i) up_buf[i] = preamp.process_sample(up_buf[i]);
        os.downsample_2x(up_buf, down_buf, len);
        for (size_t i = 0; i < len; ++i) {
            const double attenuated = down_buf[i] * volume * volume;
            const double amplified = no_poweramp ? attenuated : power_amp.process(attenuated);
            output[sample_pos + i] = speaker.process(amplified) * POST_SPEAKER_GAIN;
        }
        sample_pos = chunk_end;
    }
    if (stats) { stats->note_ons = note_on_count; stats->peak_polyphony = peak_poly; }
    return output;
}

}  // namespace owo
