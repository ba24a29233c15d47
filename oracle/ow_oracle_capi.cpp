// ORACLE (test infrastructure, CPU, f64) -- NOT part of the shipped product path.
// C entry points (ctypes) over the CPU restatement.  Only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may load this library; it is the checker, never the
// thing measured or shipped.
//
// Pinning status: the reference is 100% Rust and no Rust toolchain exists in the build
// image, so the reference cannot be compiled or run here (oracle/_ref is not buildable).
// The restatement is pinned against every known-answer value the reference's own tests
// hold for this path (tests/test_oracle_kat.py) and against the
// reference's only golden numeric fixture (tests/golden/alias_audit_v0_5_1.json).
// Since round 3 the three generated solvers (Twin-T tremolo, 12-node preamp, 7-BJT power amp) are also pinned to the
// reference's BAKED matrices: tools/extract_constants.py lifts S / K / S_NI / A_neg (and the backward-Euler sets, DC operating
// points, device tables) out of the generated files as data, and tests/test_oracle_baked_matrices.py requires the oracle's own
// rebuild at the codegen rate to reproduce them (1e-15 / 1e-11 of the largest entry), re-derives them in numpy from G and C,
// and checks the DC operating points against independent device laws.
// The melange-primitives Biquad boundary stays "parity unpinned" (see ow_voice.hpp).
#include "ow_engine.hpp"
#include "ow_alias_audit.hpp"
#include "ow_render_midi.hpp"
#include <cstdio>

using namespace owo;

extern "C" {

// ---- engine mirror (engine.rs public API) ----
void* owo_engine_new(double sr) { return new WurliEngine(sr); }
void* owo_engine_new_kind(double sr, int preamp_kind) { return new WurliEngine(sr, preamp_kind); }
void* owo_engine_new_kinds(double sr, int preamp_kind, int power_amp_kind) { return new WurliEngine(sr, preamp_kind, power_amp_kind); }
void* owo_engine_new_kinds3(double sr, int preamp_kind, int power_amp_kind, int tremolo_kind) { return new WurliEngine(sr, preamp_kind, power_amp_kind, tremolo_kind); }
void owo_engine_free(void* e) { delete (WurliEngine*)e; }
void owo_engine_melange_stats(void* e, unsigned long long* out4) { const MelangePreamp& m = ((WurliEngine*)e)->mel; out4[0] = m.stat_samples; out4[1] = m.stat_main; out4[2] = m.stat_shadow; out4[3] = m.stat_max; }
void owo_engine_melange_be(void* e, unsigned long long* out2) { const MelangePreamp& m = ((WurliEngine*)e)->mel; out2[0] = m.main.diag_be_fallback_count; out2[1] = m.shadow.diag_be_fallback_count; }
void owo_engine_set_rail_sag(void* e, int on) { ((WurliEngine*)e)->set_rail_sag(on != 0); }
int owo_engine_rail_sag_enabled(void* e) { return ((WurliEngine*)e)->rail_sag_enabled() ? 1 : 0; }
// power_amp_diag (engine.rs:418-420): clamp_count, nr_max_iter_count, peak_output_volts; + oracle-only guard reset count
void owo_engine_power_amp_diag(void* e, unsigned long long* clamp, unsigned long long* nr_max, double* peak, unsigned long long* guard_resets) {
    const WurliEngine* w = (const WurliEngine*)e;
    *clamp = w->power_amp_kind ? w->mel_pa.state.diag_clamp_count : 0ull;
    *nr_max = w->power_amp_kind ? w->mel_pa.state.diag_nr_max_iter_count : 0ull;
    *peak = w->power_amp_kind ? w->mel_pa.state.diag_peak_output : 0.0;
    *guard_resets = w->power_amp_kind ? w->mel_pa.guard_resets : 0ull;
}
// test poke (mirror of the product's ow_test_engine_poke_voice): field 81 = pickup charge q, 0 = mode 0's sine state.  Returns 0, -1 if
// the slot has no such voice.
int owo_engine_poke_voice(void* e, int slot, int steal, int field, double value) {
    WurliEngine* w = (WurliEngine*)e;
    if (slot < 0 || slot >= MAX_VOICES) return -1;
    Voice* v = steal ? w->voices[slot].steal_voice.get() : w->voices[slot].voice.get();
    if (!v) return -1;
    if (field == 81) v->pickup.q = value;
    else if (field == 0) v->reed.modes[0].s = value;
    else return -1;
    return 0;
}
// n steps of the tremolo cell alone (Tremolo::process: oscillator + LED + CdS envelope), results discarded: what the product's
// ow_test_pool_stagger_tremolo does to a phase group, so that staggered engines can be compared with the oracle
void owo_engine_advance_tremolo(void* e, size_t n) {
    WurliEngine* w = (WurliEngine*)e;
    for (size_t i = 0; i < n; ++i) (void)w->tremolo.process();
}
// test poke (mirror of the product's ow_test_engine_poke_preamp_node): node voltage of the legacy preamp's main / shadow solver state
void owo_engine_poke_preamp_node(void* e, int shadow, int node, double v) {
    WurliEngine* w = (WurliEngine*)e;
    (shadow ? w->preamp.shadow : w->preamp.main).v[node] = v;
}
// the legacy preamp's solver state (main / shadow) as stored: j_cin, cin_rhs_prev, v[8], i_nl[2], v_nl[2], then bjt_ic(v_nl[0..1]) evaluated
// NOW -- tests/test_oracle_sensitivity.py pins the invariant i_nl == bjt_ic(v_nl) the product's carried evaluation rests on
void owo_engine_preamp_state(void* e, int shadow, double* out16) {
    WurliEngine* w = (WurliEngine*)e;
    const DkState& s = shadow ? w->preamp.shadow : w->preamp.main;
    out16[0] = s.j_cin; out16[1] = s.cin_rhs_prev;
    for (int i = 0; i < 8; ++i) out16[2 + i] = s.v[i];
    out16[10] = s.i_nl[0]; out16[11] = s.i_nl[1]; out16[12] = s.v_nl[0]; out16[13] = s.v_nl[1];
    out16[14] = dk::bjt_ic(s.v_nl[0]); out16[15] = dk::bjt_ic(s.v_nl[1]);
}
// test instrumentation: every r_ldr the engine's tremolo produces from now on is moved by `ulps` doubles (tests/test_oracle_sensitivity.py:
// what a libm whose pow / exp / sin differ in the last places does to the reference itself)
void owo_engine_set_r_ulp(void* e, int ulps) { ((WurliEngine*)e)->tremolo.r_ulp = ulps; }
void owo_engine_poke_pa_node(void* e, int node, double v) { ((WurliEngine*)e)->mel_pa.state.v_prev[node] = v; }
// render with the power-amp tap (chain rate) besides the output
void owo_engine_render_pa_tap(void* e, float* out, double* pa, size_t n) {
    WurliEngine* w = (WurliEngine*)e;
    w->pa_tap = pa;
    w->render(out, n);
    w->pa_tap = nullptr;
}

// ---- melange power amp alone (power_amp.rs melange_adapter::PowerAmp), for the known-answer tests and solver-internal taps ----
void* owo_mpa_new(double sr) { MelangePowerAmp* p = new MelangePowerAmp(); p->init(sr); return p; }
void owo_mpa_free(void* p) { delete (MelangePowerAmp*)p; }
void owo_mpa_reset(void* p) { ((MelangePowerAmp*)p)->reset(); }
void owo_mpa_set_rail_sag(void* p, int on) { ((MelangePowerAmp*)p)->set_rail_sag(on != 0); }
// out[n]; taps (optional, [n][4]): outer Newton iterations (70 = exhausted), inner iterations summed, BE retry used, guard resets so far
void owo_mpa_process(void* p, const double* in, double* out, double* taps, size_t n) {
    MelangePowerAmp* a = (MelangePowerAmp*)p;
    for (size_t i = 0; i < n; ++i) {
        out[i] = a->process(in[i]);
        if (taps) {
            taps[4 * i + 0] = (double)a->state.tap_outer_iters; taps[4 * i + 1] = (double)a->state.tap_inner_iters;
            taps[4 * i + 2] = (double)a->state.tap_be_used; taps[4 * i + 3] = (double)a->guard_resets;
        }
    }
}
// the same interface as the product's debug hook ow_debug_power_amp (one row): taps [n][3] = outer iterations, guard resets, positive rail
void owo_mpa_run(double sr, const double* in, size_t n, int rail_sag, long long poke_at, int poke_node, double poke_val, double* out, double* taps) {
    MelangePowerAmp a; a.init(sr); a.set_rail_sag(rail_sag != 0);
    for (size_t i = 0; i < n; ++i) {
        if (poke_at == (long long)i) a.state.v_prev[poke_node] = poke_val;
        out[i] = a.process(in[i]);
        if (taps) { taps[3 * i] = (double)a.state.last_nr_iterations; taps[3 * i + 1] = (double)a.guard_resets; taps[3 * i + 2] = a.rails.v_rail_pos; }
    }
}
void owo_mpa_stats(unsigned long long* out21, int reset) {
    PaStats& st = pa_stats();
    if (out21) for (int i = 0; i < 21; ++i) out21[i] = st.v[i];
    if (reset) st = PaStats{};
}
// inner-loop trips of bjt_with_parasitics by device since the last reset: out[8][16] (what the device-phase mapping of k_post_mpa is sized from)
void owo_mpa_inner_hist(unsigned long long* out128, int reset) {
    PaStats& st = pa_stats();
    if (out128) for (int d = 0; d < 8; ++d) for (int t = 0; t < 16; ++t) out128[d * 16 + t] = st.inner_hist[d][t];
    if (reset) for (int d = 0; d < 8; ++d) for (int t = 0; t < 16; ++t) st.inner_hist[d][t] = 0;
}
void owo_mpa_deepest_hist(unsigned long long* out16, int reset) {    // passes by the trip count of their deepest device loop
    PaStats& st = pa_stats();
    if (out16) for (int t = 0; t < 16; ++t) out16[t] = st.deepest_hist[t];
    if (reset) for (int t = 0; t < 16; ++t) st.deepest_hist[t] = 0;
}
void owo_mpa_rails(void* p, double* pos, double* neg) {
    const MelangePowerAmp* a = (const MelangePowerAmp*)p;
    *pos = a->rail_sag_on ? a->rails.v_rail_pos : 22.5; *neg = a->rail_sag_on ? a->rails.v_rail_neg : 22.5;
}
void owo_mpa_state(void* p, double* v20) { for (int i = 0; i < 20; ++i) v20[i] = ((MelangePowerAmp*)p)->state.v_prev[i]; }
void owo_mpa_poke_node(void* p, int node, double v) { ((MelangePowerAmp*)p)->state.v_prev[node] = v; }   // forced-divergence tests
// RailDynamics alone (power_amp.rs:65-165)
void owo_rail_run(double sr, const double* v_out, size_t n, double* pos, double* neg) {
    RailDynamics r; r.init(sr);
    for (size_t i = 0; i < n; ++i) { r.step(v_out[i]); pos[i] = r.v_rail_pos; neg[i] = r.v_rail_neg; }
}
void owo_engine_set_sample_rate(void* e, double sr) { ((WurliEngine*)e)->set_sample_rate(sr); }
void owo_engine_reset(void* e) { ((WurliEngine*)e)->reset(); }
void owo_engine_warm_up(void* e) { ((WurliEngine*)e)->warm_up(); }
void owo_engine_note_on(void* e, int note, float vel) { ((WurliEngine*)e)->note_on(note, vel); }
void owo_engine_note_off(void* e, int note) { ((WurliEngine*)e)->note_off(note); }
void owo_engine_set_sustain(void* e, int held) { ((WurliEngine*)e)->set_sustain(held != 0); }
void owo_engine_set_volume(void* e, double v) { ((WurliEngine*)e)->set_volume(v); }
void owo_engine_set_tremolo_depth(void* e, double v) { ((WurliEngine*)e)->set_tremolo_depth(v); }
void owo_engine_set_speaker_character(void* e, double v) { ((WurliEngine*)e)->set_speaker_character(v); }
void owo_engine_set_mlp_enabled(void* e, int on) { ((WurliEngine*)e)->set_mlp_enabled(on != 0); }
void owo_engine_set_noise_enabled(void* e, int on) { ((WurliEngine*)e)->set_noise_enabled(on != 0); }
void owo_engine_set_noise_gain(void* e, double g) { ((WurliEngine*)e)->set_noise_gain(g); }
void owo_engine_set_noise_seed(void* e, unsigned long long seed) { ((WurliEngine*)e)->set_noise_seed((uint64_t)seed); }
// KAT hook: first n u64 outputs of thermal stream k after seeding with `master`, and n gaussians of that stream
void owo_noise_stream(unsigned long long master, int k, int n, unsigned long long* u64_out, double* gauss_out) {
    owo::MelState st; st.init_default(); st.set_seed((uint64_t)master);
    owo::MelState g = st;
    for (int i = 0; i < n; ++i) u64_out[i] = st.noise_rng[k].next_u64();
    for (int i = 0; i < n; ++i) gauss_out[i] = g.gaussian(k);
}
void owo_engine_render(void* e, float* out, size_t len) { ((WurliEngine*)e)->render(out, len); }
// render + tap of the pre-chain voice sum (f64), for stage-wise parity tests
void owo_engine_render_tap(void* e, float* out, double* voice_sum, size_t len) {
    WurliEngine* en = (WurliEngine*)e;
    en->voice_sum_tap = voice_sum;
    en->render(out, len);
    en->voice_sum_tap = nullptr;
}
// render + taps: voice sum [len], preamp out [len*osr], tremolo R [len*osr] (any may be null)
void owo_engine_render_taps(void* e, float* out, double* voice_sum, double* preamp_out, double* r_out, size_t len) {
    WurliEngine* en = (WurliEngine*)e;
    en->voice_sum_tap = voice_sum; en->preamp_tap = preamp_out; en->r_tap = r_out;
    en->render(out, len);
    en->voice_sum_tap = nullptr; en->preamp_tap = nullptr; en->r_tap = nullptr;
}
int owo_engine_count_state(void* e, int st) { return ((WurliEngine*)e)->count_state(st); }
int owo_engine_active_voice_count(void* e) { return ((WurliEngine*)e)->active_voice_count(); }
int owo_engine_steal_voice_count(void* e) { return ((WurliEngine*)e)->steal_voice_count(); }
unsigned long long owo_engine_nan_guard_fires(void* e) { return ((WurliEngine*)e)->nan_guard_fires; }
int owo_engine_slot_state(void* e, int slot) { return ((WurliEngine*)e)->voices[slot].state; }
int owo_engine_slot_note(void* e, int slot) { return ((WurliEngine*)e)->voices[slot].midi_note; }

// ---- offline / batch ----
// Voice::render_note (voice.rs:191-221); returns number of samples written (<= cap)
size_t owo_render_note(int midi, double vel, double dur_s, double sr, double* out, size_t cap) {
    std::vector<double> v = render_note(midi, vel, dur_s, sr);
    const size_t n = std::min(cap, v.size());
    for (size_t i = 0; i < n; ++i) out[i] = v[i];
    return v.size();
}
size_t owo_batch_render_job(int note, int vel_u8, double dur_s, double sr, double volume, double speaker_char, double r_ldr,
                            int mlp, int poweramp, double* out, size_t cap) {
    std::vector<double> v = batch_render_job(note, vel_u8, dur_s, sr, volume, speaker_char, r_ldr, mlp != 0, poweramp != 0);
    const size_t n = std::min(cap, v.size());
    for (size_t i = 0; i < n; ++i) out[i] = v[i];
    return v.size();
}

size_t owo_batch_render_job_kind(int note, int vel_u8, double dur_s, double sr, double volume, double speaker_char, double r_ldr,
                                 int mlp, int poweramp, int preamp_kind, double* out, size_t cap) {
    std::vector<double> v = batch_render_job(note, vel_u8, dur_s, sr, volume, speaker_char, r_ldr, mlp != 0, poweramp != 0, preamp_kind);
    const size_t n = std::min(cap, v.size());
    for (size_t i = 0; i < n; ++i) out[i] = v[i];
    return v.size();
}

// every sample-changing flag of `preamp-bench render`; opts: [volume, speaker, r_ldr, tremolo_depth, displacement_scale (NaN = none)];
// flags bit0 mlp, 1 poweramp, 2 no_preamp, 3 no_attack_noise, 4 no_rail_sag
size_t owo_batch_render_job_ex(int note, int vel_u8, double dur_s, double sr, const double* opts, unsigned flags, int preamp_kind, int power_amp_kind,
                               double* out, size_t cap) {
    BatchJobOpts o;
    o.volume = opts[0]; o.speaker_char = opts[1]; o.r_ldr = opts[2]; o.tremolo_depth = opts[3];
    o.has_displacement_scale = opts[4] == opts[4]; o.displacement_scale = opts[4];
    o.mlp = flags & 1u; o.poweramp = flags & 2u; o.no_preamp = flags & 4u; o.no_attack_noise = flags & 8u; o.no_rail_sag = flags & 16u;
    o.preamp_kind = preamp_kind; o.power_amp_kind = power_amp_kind;
    std::vector<double> v = batch_render_job_ex(note, vel_u8, dur_s, sr, o);
    const size_t n = std::min(cap, v.size());
    for (size_t i = 0; i < n; ++i) out[i] = v[i];
    return v.size();
}
double owo_batch_normalize_scale(const double* x, size_t n) { return batch_normalize_scale(x, n); }
// Voice::render_note_with_scale (voice.rs:201-221)
size_t owo_render_note_scaled(int midi, double vel, double dur, double sr, double scale, double* out, size_t cap) {
    const uint32_t seed = (uint32_t)midi * 2654435761u;
    Voice voice;
    voice.note_on(midi, vel, sr, seed, false);
    voice.pickup.displacement_scale = scale;
    const size_t n = (size_t)as_u64(dur * sr);
    std::vector<double> v(n, 0.0);
    for (size_t off = 0; off < n; off += 1024) voice.render(v.data() + off, std::min((size_t)1024, n - off));
    for (size_t i = 0; i < std::min(cap, n); ++i) out[i] = v[i];
    return n;
}

// ---- unit-level hooks for the known-answer tests ----
double owo_midi_to_freq(int m) { return midi_to_freq(m); }
double owo_tip_mass_ratio(int m) { return tip_mass_ratio(m); }
void owo_eigenvalues(double mu, double* out) { eigenvalues(mu, out); }
double owo_mode_shape(double beta, double xi) { return mode_shape(beta, xi); }
double owo_reed_compliance(int m) { return reed_compliance(m); }
void owo_mode_ratios(double mu, double* out) { mode_ratios(mu, out); }
double owo_reed_length_mm(int m) { return reed_length_mm(m); }
void owo_reed_blank_dims(int m, double* wt) { reed_blank_dims(m, wt[0], wt[1]); }
double owo_pickup_displacement_scale(int m) { return pickup_displacement_scale(m); }
double owo_fundamental_decay_rate(int m) { return fundamental_decay_rate(m); }
void owo_spatial_coupling(double mu, double len_mm, double* out) { spatial_coupling_coefficients(mu, len_mm, out); }
double owo_output_scale(int m, double v) { return output_scale(m, v); }
double owo_velocity_exponent(int m) { return velocity_exponent(m); }
double owo_velocity_scurve(double v) { return velocity_scurve(v); }
double owo_register_trim_db(int m) { return register_trim_db(m); }
double owo_pickup_rms_proxy(double ds, double f0, double fc) { return pickup_rms_proxy(ds, f0, fc); }
double owo_freq_detune(int m) { return freq_detune((uint8_t)m); }
void owo_mode_amplitude_offsets(int m, double* out) { mode_amplitude_offsets((uint8_t)m, out); }
double owo_dwell_time(double v, double f) { return dwell_time(v, f); }
double owo_onset_ramp_time(double v, double f) { return onset_ramp_time(v, f); }
void owo_dwell_attenuation(double v, double f, const double* ratios, double* out) { dwell_attenuation(v, f, ratios, out); }
// out: 5 freq cents, 5 decay ratios, 1 ds
void owo_mlp_infer(int midi, double vel, double* out) {
    MlpCorrections c = mlp_infer(midi, vel);
    for (int i = 0; i < 5; ++i) { out[i] = c.freq_offsets_cents[i]; out[5 + i] = c.decay_offsets[i]; }
    out[10] = c.ds_correction;
}
// Packed voice parameter block after Voice::note_on -- compared field-by-field with the device note-on.
// layout: [0..6] phase_inc, [7..13] amplitude, [14..20] decay_mult, [21..27] jitter_drift, [28..34] cos_inc, [35..41] sin_inc,
//         42 onset_ramp_samples, 43 onset_ramp_inc, 44 onset_shape_exp, 45 jitter_state, 46 jitter_revert, 47 jitter_diffusion,
//         48 pickup.beta, 49 pickup.displacement_scale, 50 post_pickup_gain, 51 noise.amplitude, 52 noise.decay, 53 noise.remaining,
//         54..58 bpf b0,b1,b2,a1,a2
void owo_voice_params(int midi, double vel, double sr, unsigned seed, int mlp, double* out) {
    Voice v;
    v.note_on(midi, vel, sr, seed, mlp != 0);
    for (int m = 0; m < NUM_MODES; ++m) {
        out[m] = v.reed.modes[m].phase_inc;
        out[7 + m] = v.reed.modes[m].amplitude;
        out[14 + m] = v.reed.modes[m].decay_mult;
        out[21 + m] = v.reed.modes[m].jitter_drift;
        out[28 + m] = v.reed.modes[m].cos_inc;
        out[35 + m] = v.reed.modes[m].sin_inc;
    }
    out[42] = (double)v.reed.onset_ramp_samples;
    out[43] = v.reed.onset_ramp_inc;
    out[44] = v.reed.onset_shape_exp;
    out[45] = (double)v.reed.jitter_state;
    out[46] = v.reed.jitter_revert;
    out[47] = v.reed.jitter_diffusion;
    out[48] = v.pickup.beta;
    out[49] = v.pickup.displacement_scale;
    out[50] = v.post_pickup_gain;
    out[51] = v.noise.amplitude;
    out[52] = v.noise.decay_per_sample;
    out[53] = (double)v.noise.remaining;
    out[54] = v.noise.bpf.b0; out[55] = v.noise.bpf.b1; out[56] = v.noise.bpf.b2; out[57] = v.noise.bpf.a1; out[58] = v.noise.bpf.a2;
}

// pickup alone (pickup.rs tests): process buffer in place
void owo_pickup_process(double sr, double ds, double* buf, size_t n) {
    Pickup p;
    p.init(sr);
    p.displacement_scale = ds;
    p.process(buf, n);
}
double owo_pickup_soft_saturate(double y) { return pickup_soft_saturate(y); }

// reed alone (reed.rs tests)
void owo_reed_render(double f0, const double* ratios, const double* amps, const double* decay, double onset, double vel, double sr,
                     unsigned seed, double* out, size_t n) {
    ModalReed r;
    r.init(f0, ratios, amps, decay, onset, vel, sr, seed);
    for (size_t i = 0; i < n; ++i) out[i] = 0.0;
    r.render(out, n);
}

// attack noise alone (hammer.rs:223-239): AttackNoise::new(velocity, f0, sr, seed).render(out[0..n)) into zeros; returns is_done()
int owo_attack_noise_render(double velocity, double f0, double sr, unsigned seed, double* out, size_t n) {
    AttackNoise a;
    a.init(velocity, f0, sr, seed);
    for (size_t i = 0; i < n; ++i) out[i] = 0.0;
    a.render(out, n);
    return a.is_done() ? 1 : 0;
}

// biquad: kind 0=LP 1=HP 2=BP; filters x in place
void owo_biquad_process(int kind, double fc, double q, double sr, double* x, size_t n) {
    Biquad b = Biquad::make((Biquad::Kind)kind, fc, q, sr);
    for (size_t i = 0; i < n; ++i) x[i] = b.process(x[i]);
}

// legacy preamp: DC operating point (8 node voltages + 2 Vbe) at R_ldr = 1 Mohm
void owo_preamp_dc(double sr, double* v8_vnl2) {
    DkPreamp p;
    p.init(sr);
    for (int i = 0; i < 8; ++i) v8_vnl2[i] = p.v_dc[i];
    v8_vnl2[8] = p.main.v_nl[0];
    v8_vnl2[9] = p.main.v_nl[1];
}
// preamp matrices for identity tests: s_base (64), a_neg_base (64), k (4), s_fb_fb
void owo_preamp_matrices(double sr, double* s, double* aneg, double* k4, double* sfbfb) {
    DkPreamp p;
    p.init(sr);
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) { s[i * 8 + j] = p.s_base[i][j]; aneg[i * 8 + j] = p.a_neg_base[i][j]; }
    k4[0] = p.k[0][0]; k4[1] = p.k[0][1]; k4[2] = p.k[1][0]; k4[3] = p.k[1][1];
    *sfbfb = p.s_fb_fb;
}
// preamp run: per-sample input x[n] and R_ldr r[n] (r may be null -> static r_static after reset()) -> y[n]
void owo_preamp_run(double sr, const double* x, const double* r, double r_static, double* y, size_t n) {
    DkPreamp p;
    p.init(sr);
    if (!r) { p.reset(); p.set_ldr_resistance(r_static); }
    for (size_t i = 0; i < n; ++i) {
        if (r) p.set_ldr_resistance(r[i]);
        y[i] = p.process_sample(x[i]);
    }
}

// tremolo: n shunt-impedance samples at depth d after Tremolo::new(d, sr); optional osc voltage tap
void owo_tremolo_stats(unsigned long long* out5) { for (int i = 0; i < 5; ++i) out5[i] = trem_stats()[i]; }
void owo_tremolo_run(double depth, double sr, double* r_out, size_t n) {
    Tremolo* t = new Tremolo();
    t->init(depth, sr);
    for (size_t i = 0; i < n; ++i) r_out[i] = t->process();
    delete t;
}
// the same for a given oscillator kind (1 = the `legacy-tremolo` LFO)
void owo_tremolo_run_kind(int kind, double depth, double sr, double* r_out, size_t n) {
    Tremolo* t = new Tremolo();
    t->kind = kind;
    t->init(depth, sr);
    for (size_t i = 0; i < n; ++i) r_out[i] = t->process();
    delete t;
}
// raw oscillator voltage: state as after Tremolo::new(sr) settle, then n samples of v[OUT]
void owo_tremolo_osc(double sr, double* v_out, size_t n) {
    Tremolo* t = new Tremolo();
    t->init(1.0, sr);
    for (size_t i = 0; i < n; ++i) v_out[i] = t->osc.process_sample(0.0);
    delete t;
}
// tremolo matrices after set_sample_rate(sr): s(49) k(16) s_ni(28) a_neg(49)
void owo_tremolo_matrices(double sr, double* s, double* k, double* sni, double* aneg) {
    TremCircuit c;
    c.init_default();
    if (std::fabs(sr - TREM_SAMPLE_RATE) > 0.5) c.set_sample_rate(sr);
    for (int i = 0; i < 7; ++i) for (int j = 0; j < 7; ++j) { s[i * 7 + j] = c.s[i][j]; aneg[i * 7 + j] = c.a_neg[i][j]; }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) k[i * 4 + j] = c.k[i][j];
    for (int i = 0; i < 7; ++i) for (int j = 0; j < 4; ++j) sni[i * 4 + j] = c.s_ni[i][j];
}
double owo_fast_exp(double x) { return fast_exp(x); }

// melange 12-node preamp (main - shadow) run: per-sample input x[n], R_ldr r[n] (null -> untouched 100 kOhm nominal)
void owo_melange_run(double sr, const double* x, const double* r, double* y, size_t n) {
    MelangePreamp* p = new MelangePreamp();
    p->init(sr);
    for (size_t i = 0; i < n; ++i) {
        if (r) p->set_ldr_resistance(r[i]);
        y[i] = p->process_sample(x[i]);
    }
    delete p;
}
// same with the thermal noise of the main state on (seed != 0: deterministic), gain = thermal_gain
void owo_melange_run_noise(double sr, const double* x, const double* r, double* y, size_t n, unsigned long long seed, double gain) {
    MelangePreamp* p = new MelangePreamp();
    p->init(sr);
    p->set_noise_seed((uint64_t)seed); p->set_noise_enabled(true); p->set_thermal_gain(gain);
    for (size_t i = 0; i < n; ++i) {
        if (r) p->set_ldr_resistance(r[i]);
        y[i] = p->process_sample(x ? x[i] : 0.0);
    }
    delete p;
}
// single CircuitState from CircuitState::default(): n zero-input steps, returns v_prev (12) + i_nl_prev (3) + diag counters (3)
void owo_melange_default_steps(size_t n, double* out18) {
    MelState* s = new MelState();
    s->init_default();
    for (size_t i = 0; i < n; ++i) s->process_sample(0.0);
    for (int i = 0; i < 12; ++i) out18[i] = s->v_prev[i];
    for (int i = 0; i < 3; ++i) out18[12 + i] = s->i_nl_prev[i];
    out18[15] = (double)s->diag_be_fallback_count; out18[16] = (double)s->diag_nan_reset_count; out18[17] = (double)s->diag_voltage_damp_count;
    delete s;
}
// rebuilt trapezoidal matrices at (sr, r): s(144) k(9) s_ni(36) a_neg(144)
void owo_melange_matrices(double sr, double r, double* s, double* k, double* sni, double* aneg) {
    MelState* st = new MelState();
    st->init_default();
    st->current_sample_rate = sr;
    st->pot_0_resistance = r;
    st->rebuild_matrices();
    for (int i = 0; i < 12; ++i) for (int j = 0; j < 12; ++j) { s[i * 12 + j] = st->s[i][j]; aneg[i * 12 + j] = st->a_neg[i][j]; }
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) k[i * 3 + j] = st->k[i][j];
    for (int i = 0; i < 12; ++i) for (int j = 0; j < 3; ++j) sni[i * 3 + j] = st->s_ni[i][j];
    delete st;
}

double owo_power_amp(double x) { PowerAmp p; return p.process(x); }
void owo_speaker_run(double sr, double character, double* x, size_t n) {
    Speaker s;
    s.init(sr);
    s.set_character(character);
    for (size_t i = 0; i < n; ++i) x[i] = s.process(x[i]);
}
void owo_oversampler_down(const double* up, double* y, size_t n_out) { Oversampler os; os.downsample_2x(up, y, n_out); }
// G (without R_ldr), 2w and v at the DC point of the legacy preamp, for the layer-1 / layer-4 tests of the reference's pyramid
void owo_preamp_gw(double sr, double* g64, double* two_w8) {
    DkPreamp p;
    p.init(sr);
    for (int i = 0; i < 8; ++i) { for (int j = 0; j < 8; ++j) g64[i * 8 + j] = p.g_dc_base[i][j]; two_w8[i] = p.two_w[i]; }
}
// legacy preamp with R_ldr set BEFORE reset() (the pyramid's time-domain tests), then n_pre samples of x_pre, an R step, n samples of x
void owo_preamp_step(double sr, double r0, size_t n_pre, double r1, const double* x, double* y, size_t n, double* v8_end) {
    DkPreamp p;
    p.init(sr);
    p.set_ldr_resistance(r0); p.reset();
    for (size_t i = 0; i < n_pre; ++i) p.process_sample(0.0);
    p.set_ldr_resistance(r1);
    for (size_t i = 0; i < n; ++i) y[i] = p.process_sample(x ? x[i] : 0.0);
    if (v8_end) for (int i = 0; i < 8; ++i) v8_end[i] = p.main.v[i];
}
void owo_oversampler_roundtrip(const double* x, double* up, double* y, size_t n) {
    Oversampler os;
    os.upsample_2x(x, n, up);
    os.downsample_2x(up, y, n);
}

// ---- alias_audit.rs (click-band alias detector) ----
size_t owo_alias_audit_result_size() { return sizeof(AliasAuditResult); }
int owo_alias_audit_analyze(const double* signal, size_t len, double sr, double nominal_f0, void* result) {
    return audit_analyze(signal, len, sr, nominal_f0, (AliasAuditResult*)result) ? 0 : -1;
}
// render_stimulus (alias_audit.rs:127-160): returns the sample count, writes min(cap, count) samples
size_t owo_alias_audit_render_stimulus(int note, int velocity, int preamp_kind, double* out, size_t cap) {
    std::vector<double> v = audit_render_stimulus(note, velocity, preamp_kind);
    const size_t n = std::min(cap, v.size());
    for (size_t i = 0; i < n; ++i) out[i] = v[i];
    return v.size();
}
int owo_alias_audit_run(int note, int velocity, int preamp_kind, void* result) {   // run_with_note, alias_audit.rs:104-108
    std::vector<double> v = audit_render_stimulus(note, velocity, preamp_kind);
    return audit_analyze(v.data(), v.size(), AUDIT_SAMPLE_RATE, audit_midi_note_hz(note), (AliasAuditResult*)result) ? 0 : -1;
}
double owo_alias_dft_magnitude(const double* s, size_t len, double freq, double sr) { return audit_dft_magnitude(s, len, freq, sr); }
double owo_alias_bandpass_rms(const double* s, size_t len, double sr, double lo, double hi) { return audit_bandpass_rms(s, len, sr, lo, hi); }
double owo_alias_plateau_metric(const double* dbc12, unsigned* from) { double w; uint32_t f; audit_plateau_metric(dbc12, &w, &f); *from = f; return w; }

// ---- preamp-bench render-midi (tools/preamp-bench/src/main.rs:1603-1923) ----
// events as parallel arrays; returns the event count (writes min(cap, count)), -1 on a parse error
long long owo_smf_parse(const uint8_t* data, size_t len, int track_filter, double* time_s, uint8_t* type, uint8_t* note, uint8_t* value, size_t cap) {
    try {
        std::vector<TimedEvent> ev = smf_events(data, len, track_filter);
        for (size_t i = 0; i < std::min(cap, ev.size()); ++i) { time_s[i] = ev[i].time_s; type[i] = ev[i].type; note[i] = ev[i].note; value[i] = ev[i].value; }
        return (long long)ev.size();
    } catch (const std::exception&) { return -1; }
}
// returns the sample count (writes min(cap, count)); stats2 = {note-ons, peak polyphony}
size_t owo_render_midi(const double* time_s, const uint8_t* type, const uint8_t* note, const uint8_t* value, size_t n, double volume,
                       double speaker_char, int no_poweramp, double tail_s, double* out, size_t cap, unsigned long long* stats2) {
    std::vector<TimedEvent> ev(n);
    for (size_t i = 0; i < n; ++i) ev[i] = TimedEvent{time_s[i], type[i], note[i], value[i]};
    MidiRenderStats st;
    std::vector<double> v = render_midi(ev, volume, speaker_char, no_poweramp != 0, tail_s, &st);
    for (size_t i = 0; i < std::min(cap, v.size()); ++i) out[i] = v[i];
    if (stats2) { stats2[0] = st.note_ons; stats2[1] = st.peak_polyphony; }
    return v.size();
}

// ---- baked matrices of the three generated solvers next to what the restated rebuild_matrices / invert_n produce AT THE CODEGEN RATE
// with the "within 0.5 Hz -> copy the defaults" shortcut bypassed (gen_tremolo.rs:29-1132,2139-2342; gen_preamp.rs consts,1990-2219;
// gen_power_amp.rs:965-7204,8624-8831).  solver: 0 tremolo (N 7, M 4), 1 preamp (N 12, M 3), 2 power amp (N 20, M 16).
// Each output is row-major: s[N][N] k[M][M] sni[N][M] aneg[N][N], then the backward-Euler set.  The preamp's rebuild leaves its BE set
// alone (gen_preamp.rs:2058-2061), so its rebuilt BE outputs ARE the baked ones.  Returns N * 100 + M, -1 for an unknown solver.
static void owo_flat(const double* src, size_t n, double* dst) { if (dst) std::memcpy(dst, src, sizeof(double) * n); }
int owo_baked_matrices(int solver, double* s, double* k, double* sni, double* aneg, double* s_be, double* k_be, double* sni_be, double* aneg_be) {
    switch (solver) {
        case 0: owo_flat(&TREM_S_DEFAULT[0][0], 49, s); owo_flat(&TREM_K_DEFAULT[0][0], 16, k); owo_flat(&TREM_S_NI_DEFAULT[0][0], 28, sni); owo_flat(&TREM_A_NEG_DEFAULT[0][0], 49, aneg);
                owo_flat(&TREM_S_BE_DEFAULT[0][0], 49, s_be); owo_flat(&TREM_K_BE_DEFAULT[0][0], 16, k_be); owo_flat(&TREM_S_NI_BE_DEFAULT[0][0], 28, sni_be); owo_flat(&TREM_A_NEG_BE_DEFAULT[0][0], 49, aneg_be);
                return 704;
        case 1: owo_flat(&PRE_S_DEFAULT[0][0], 144, s); owo_flat(&PRE_K_DEFAULT[0][0], 9, k); owo_flat(&PRE_S_NI_DEFAULT[0][0], 36, sni); owo_flat(&PRE_A_NEG_DEFAULT[0][0], 144, aneg);
                owo_flat(&PRE_S_BE_DEFAULT[0][0], 144, s_be); owo_flat(&PRE_K_BE_DEFAULT[0][0], 9, k_be); owo_flat(&PRE_S_NI_BE_DEFAULT[0][0], 36, sni_be); owo_flat(&PRE_A_NEG_BE_DEFAULT[0][0], 144, aneg_be);
                return 1203;
        case 2: owo_flat(&PA_S_DEFAULT[0][0], 400, s); owo_flat(&PA_K_DEFAULT[0][0], 256, k); owo_flat(&PA_S_NI_DEFAULT[0][0], 320, sni); owo_flat(&PA_A_NEG_DEFAULT[0][0], 400, aneg);
                owo_flat(&PA_S_BE_DEFAULT[0][0], 400, s_be); owo_flat(&PA_K_BE_DEFAULT[0][0], 256, k_be); owo_flat(&PA_S_NI_BE_DEFAULT[0][0], 320, sni_be); owo_flat(&PA_A_NEG_BE_DEFAULT[0][0], 400, aneg_be);
                return 2016;
    }
    return -1;
}
int owo_rebuilt_matrices_at_codegen_rate(int solver, double* s, double* k, double* sni, double* aneg, double* s_be, double* k_be, double* sni_be, double* aneg_be) {
    switch (solver) {
        case 0: {
            TremCircuit* c = new TremCircuit();
            c->init_default();
            std::memset(c->s, 0, sizeof c->s); std::memset(c->k, 0, sizeof c->k); std::memset(c->s_ni, 0, sizeof c->s_ni); std::memset(c->a_neg, 0, sizeof c->a_neg);
            std::memset(c->s_be, 0, sizeof c->s_be); std::memset(c->k_be, 0, sizeof c->k_be); std::memset(c->s_ni_be, 0, sizeof c->s_ni_be); std::memset(c->a_neg_be, 0, sizeof c->a_neg_be);
            c->rebuild_matrices(TREM_SAMPLE_RATE * 1.0);
            owo_flat(&c->s[0][0], 49, s); owo_flat(&c->k[0][0], 16, k); owo_flat(&c->s_ni[0][0], 28, sni); owo_flat(&c->a_neg[0][0], 49, aneg);
            owo_flat(&c->s_be[0][0], 49, s_be); owo_flat(&c->k_be[0][0], 16, k_be); owo_flat(&c->s_ni_be[0][0], 28, sni_be); owo_flat(&c->a_neg_be[0][0], 49, aneg_be);
            delete c;
            return 704;
        }
        case 1: {
            MelState* st = new MelState();
            st->init_default();
            std::memset(st->s, 0, sizeof st->s); std::memset(st->k, 0, sizeof st->k); std::memset(st->s_ni, 0, sizeof st->s_ni); std::memset(st->a_neg, 0, sizeof st->a_neg);
            st->current_sample_rate = PRE_SAMPLE_RATE;          // pot at its nominal 100 kOhm, as the code generator had it
            st->rebuild_matrices();
            owo_flat(&st->s[0][0], 144, s); owo_flat(&st->k[0][0], 9, k); owo_flat(&st->s_ni[0][0], 36, sni); owo_flat(&st->a_neg[0][0], 144, aneg);
            owo_flat(&st->s_be[0][0], 144, s_be); owo_flat(&st->k_be[0][0], 9, k_be); owo_flat(&st->s_ni_be[0][0], 36, sni_be); owo_flat(&st->a_neg_be[0][0], 144, aneg_be);
            const int singular = (int)st->diag_singular_matrix_count;
            delete st;
            return singular ? -2 : 1203;
        }
        case 2: {
            PaCircuit* c = new PaCircuit();
            c->init_default();
            std::memset(c->s, 0, sizeof c->s); std::memset(c->k, 0, sizeof c->k); std::memset(c->s_ni, 0, sizeof c->s_ni); std::memset(c->a_neg, 0, sizeof c->a_neg);
            std::memset(c->s_be, 0, sizeof c->s_be); std::memset(c->k_be, 0, sizeof c->k_be); std::memset(c->s_ni_be, 0, sizeof c->s_ni_be); std::memset(c->a_neg_be, 0, sizeof c->a_neg_be);
            c->rebuild_matrices(PA_SAMPLE_RATE * 1.0);
            owo_flat(&c->s[0][0], 400, s); owo_flat(&c->k[0][0], 256, k); owo_flat(&c->s_ni[0][0], 320, sni); owo_flat(&c->a_neg[0][0], 400, aneg);
            owo_flat(&c->s_be[0][0], 400, s_be); owo_flat(&c->k_be[0][0], 256, k_be); owo_flat(&c->s_ni_be[0][0], 320, sni_be); owo_flat(&c->a_neg_be[0][0], 400, aneg_be);
            delete c;
            return 2016;
        }
    }
    return -1;
}
// G, C (N x N), N_v (M x N), N_i (N x M) and the codegen rate of a solver, for an independent numpy re-derivation in the tests.
// (gen_preamp.rs stores N_I transposed, [M][N]; it is handed out as N x M like the others.)
int owo_baked_circuit(int solver, double* g, double* c, double* nv, double* ni, double* rate) {
    switch (solver) {
        case 0: owo_flat(&TREM_G[0][0], 49, g); owo_flat(&TREM_C[0][0], 49, c); owo_flat(&TREM_N_V[0][0], 28, nv); owo_flat(&TREM_N_I[0][0], 28, ni); *rate = TREM_SAMPLE_RATE; return 704;
        case 1: owo_flat(&PRE_G[0][0], 144, g); owo_flat(&PRE_C[0][0], 144, c); owo_flat(&PRE_N_V[0][0], 36, nv);
                for (int i = 0; i < 12; ++i) for (int j = 0; j < 3; ++j) ni[i * 3 + j] = PRE_N_I[j][i];
                *rate = PRE_SAMPLE_RATE; return 1203;
        case 2: owo_flat(&PA_G[0][0], 400, g); owo_flat(&PA_C[0][0], 400, c); owo_flat(&PA_N_V[0][0], 320, nv); owo_flat(&PA_N_I[0][0], 320, ni); *rate = PA_SAMPLE_RATE; return 2016;
    }
    return -1;
}
// DC operating point the code generator baked (DC_OP, DC_NL_I) of a solver: v[N], i_nl[M]
int owo_baked_dc(int solver, double* v, double* inl) {
    switch (solver) {
        case 0: owo_flat(TREM_DC_OP, 7, v); owo_flat(TREM_DC_NL_I, 4, inl); return 704;
        case 1: owo_flat(PRE_DC_OP, 12, v); owo_flat(PRE_DC_NL_I, 3, inl); return 1203;
        case 2: owo_flat(PA_DC_OP, 20, v); owo_flat(PA_DC_NL_I, 16, inl); return 2016;
    }
    return -1;
}

size_t owo_render_midi_ex(const double* time_s, const uint8_t* type, const uint8_t* note, const uint8_t* value, size_t n, double volume,
                          double speaker_char, int no_poweramp, double tail_s, int preamp_kind, int power_amp_kind, int no_rail_sag, double* out, size_t cap) {
    std::vector<TimedEvent> ev(n);
    for (size_t i = 0; i < n; ++i) ev[i] = TimedEvent{time_s[i], type[i], note[i], value[i]};
    std::vector<double> v = render_midi(ev, volume, speaker_char, no_poweramp != 0, tail_s, nullptr, preamp_kind, power_amp_kind, no_rail_sag != 0);
    for (size_t i = 0; i < std::min(cap, v.size()); ++i) out[i] = v[i];
    return v.size();
}

}  // extern "C"
