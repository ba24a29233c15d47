// openwurli-hip: voice stage of `preamp-bench render-midi` (tools/preamp-bench/src/main.rs:1711-1852), block = MIDI job.
//
// The command has its own voice manager (not WurliEngine's): 64 slots, a note-on takes the first inactive slot or else replaces the
// oldest voice outright (no steal crossfade), a note-off starts the damper of the oldest active voice of that key, the pedal defers
// note-offs, silent voices are dropped at the top of every 64-sample chunk, events fire at the first chunk whose start time has
// reached them.  Lane = slot; the events of a job are wave-uniform, so the manager runs on ballots and wave reductions with no
// divergence except the note-on itself (one lane builds the voice: MLP, tables, hammer).  The job's ordered voice sum
// (sum_buf[i] += voice_buf[i] in slot order, :1813-1826) goes to HBM; the chain that follows it is the batch render's
// (k_job_chain: static 1 Mohm LDR, volume^2, power amp at base rate, speaker, POST_SPEAKER_GAIN).
#pragma once
#include "ow_job_kernels.h"

namespace owdev {

struct OwMidiEvDev {
    uint32_t chunk;                 // first 64-sample chunk whose start time is >= the event time (host: exact f64 comparison, :1778-1781)
    uint8_t type, note, value, pad; // 0 NoteOn(note, velocity) / 1 NoteOff(note) / 2 Pedal(value != 0 = down)
};
struct OwMidiJobDev {
    uint64_t ev_begin;              // into the event array and the pedal scratch
    uint32_t n_events;
    uint32_t pad;
    uint64_t total_samples;
};
struct OwMidiStatsDev { uint64_t note_ons, peak_polyphony; };

#define OW_MIDI_CHUNK 64

__global__ __launch_bounds__(64) void k_midi_voices(const OwConsts* __restrict__ K, const double* __restrict__ nt, double* __restrict__ vrec,
                                                    const OwMidiJobDev* __restrict__ jobs, const OwMidiEvDev* __restrict__ events,
                                                    uint32_t* __restrict__ pedal_scratch, double* __restrict__ sum, long long stride,
                                                    OwMidiStatsDev* __restrict__ stats) {
    __shared__ double tile[64 * (OW_MIDI_CHUNK + 1)];
    __shared__ double lcoef[OW_LCOEF_ROWS * 64];
    const int lane = threadIdx.x;
    const OwMidiJobDev jb = jobs[blockIdx.x];
    const OwMidiEvDev* __restrict__ ev = events + jb.ev_begin;
    uint32_t* held = pedal_scratch + jb.ev_begin;       // pedal_held (:1774): at most one entry per note-off event of the job
    double* rec = vrec + (size_t)blockIdx.x * OW_VREC_DOUBLES + lane;
    double* out = sum + (size_t)blockIdx.x * stride;
    VoiceRegs v;
    // lane state as a word, not a bool: as a bool (an SGPR lane mask) the value written under `lane == idx` inside the uniform event
    // loop only became visible one chunk later with hipcc 7.2 -- the first chunk of every note was lost
    uint32_t active = 0u;
    uint32_t midi_note = 0, age = 0;
    uint32_t age_counter = 0, n_held = 0, ei = 0, peak = 0, note_ons = 0;
    bool pedal_down = false;

    // `voices.iter().filter(active && midi_note == note).min_by_key(age)` -> v.note_off()  (:1798-1806, :1815-1823)
    auto release = [&](uint32_t note) {
        uint32_t a = (active != 0u && midi_note == note) ? age : 0xFFFFFFFFu;
        uint32_t m = a;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)m, off); m = o < m ? o : m; }
        if (m != 0xFFFFFFFFu && a == m) {                // ages are unique among live voices
            v.store(rec);
            start_damper_lane(rec, K);
            v.load(rec);
            lcoef_load(lcoef + lane, rec);
        }
    };

    const uint64_t total = jb.total_samples;
    uint32_t c = 0;
    for (uint64_t pos = 0; pos < total; pos += OW_MIDI_CHUNK, ++c) {
        const int len = (int)((total - pos) < OW_MIDI_CHUNK ? (total - pos) : OW_MIDI_CHUNK);
        while (ei < jb.n_events && ev[ei].chunk <= c) {
            const OwMidiEvDev e = ev[ei];
            const uint32_t note = e.note < OW_MIDI_LO ? OW_MIDI_LO : (e.note > OW_MIDI_HI ? OW_MIDI_HI : e.note);   // clamp(MIDI_LO, MIDI_HI)
            if (e.type == 0) {                           // :1783-1811
                age_counter += 1;
                note_ons += 1;
                const uint64_t act = __ballot(active != 0u);
                int idx;
                if (~act) idx = __builtin_ctzll(~act);   // position(|s| !s.active)
                else {                                   // min_by_key(age): the oldest voice is replaced outright
                    uint32_t m = age;
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)m, off); m = o < m ? o : m; }
                    idx = __builtin_ctzll(__ballot(age == m));
                }
                if (lane == idx) {
                    const double vel = (double)e.value / 127.0;
                    double raw[11];
                    mlp_raw_scalar(clampd(((double)note - 21.0) / (108.0 - 21.0), 0.0, 1.0), clampd(vel, 0.0, 1.0), raw);
                    const MlpOut corr = mlp_finish((int)note, raw, true);
                    note_on_lane(rec, nt, K, (int)note, vel, note * 2654435761u + age_counter, corr);
                    v.load(rec);
                    lcoef_load(lcoef + lane, rec);
                    active = 1u; midi_note = note; age = age_counter;
                }
                const uint32_t now = (uint32_t)__builtin_popcountll(__ballot(active != 0u));
                peak = now > peak ? now : peak;
            } else if (e.type == 1) {                    // :1812-1829
                if (pedal_down) {
                    if (lane == 0) __hip_atomic_store(held + n_held, note, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    n_held += 1;
                } else release(note);
            } else {                                     // :1830-1847
                pedal_down = e.value != 0;
                if (!pedal_down) {
                    __threadfence_block();
                    for (uint32_t h = 0; h < n_held; ++h) release(__hip_atomic_load(held + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    n_held = 0;
                }
            }
            ++ei;
        }
        if (active != 0u && v.is_silent(rec)) active = 0u;  // :1853-1862
        const uint64_t act = __ballot(active != 0u);
        if (act) {
            for (int s = 0; s < len; ++s) tile[lane * (OW_MIDI_CHUNK + 1) + s] = active != 0u ? v.step<false>(lcoef + lane) : 0.0;
        }
        __syncthreads();
        if (lane < len) {
            double acc = 0.0;
            for (uint64_t m = act; m; m &= m - 1) acc += tile[__builtin_ctzll(m) * (OW_MIDI_CHUNK + 1) + lane];   // slot order, :1865-1878
            out[pos + lane] = acc;
        }
        __syncthreads();
    }
    if (lane == 0) { stats[blockIdx.x].note_ons = note_ons; stats[blockIdx.x].peak_polyphony = peak; }
}

}  // namespace owdev
