/* openwurli-hip: C-ABI of the MI355X (gfx950) render core for the OpenWurli DSP hot path.
 *
 * This is the drop-in boundary: plain C, opaque handles, raw pointers and sizes.  Each entry
 * point replaces one item of the `openwurli-dsp` Rust public API that the nih-plug shell,
 * tools/reed-renderer and tools/preamp-bench call today (citations into /root/reference/).
 * A Rust facade with the reference's method names forwards 1:1.
 *
 * Conventions kept from the reference: one thread drives an engine; realtime
 * calls never fail (bad input is clamped, numeric failure degrades to silence + counters);
 * render() does not allocate once ensure_buffer_capacity() has been called; output is mono f32.
 * Only constructors and the offline entry points report errors (NULL / negative return).
 *
 * A *pool* is the MI355X-native unit: I independent engines at one sample rate that render in
 * lock-step: one lane per sounding voice (packed across engines) in the voice kernels, one lane
 * per engine in the chain kernels.  An `ow_engine*` is one engine of a pool;
 * `ow_engine_new` creates a pool of one.
 */
#ifndef OPENWURLI_HIP_H
#define OPENWURLI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ow_pool ow_pool;
typedef struct ow_engine ow_engine;

/* Version of this header's struct layouts and signatures.  ow_abi_version() returns the value the library was built with; a binding
 * checks it once after loading.  The by-pointer configuration structs (ow_batch_cfg, ow_midi_render_cfg) additionally carry their own
 * size in their first field, and ow_batch_cfg the size of one ow_job: a caller built against another header is refused ("ABI mismatch",
 * negative return) instead of having fields read past the end of what it passed. */
#define OW_ABI_VERSION 4
int ow_abi_version(void);

/* VoiceState, crates/openwurli-dsp/src/engine.rs:30-37 */
enum { OW_VOICE_FREE = 0, OW_VOICE_HELD = 1, OW_VOICE_SUSTAINED = 2, OW_VOICE_RELEASING = 3 };
/* preamp solver selection: cargo features of crates/openwurli-dsp/Cargo.toml:9-17 become a runtime enum */
enum { OW_PREAMP_LEGACY8 = 0, OW_PREAMP_MELANGE12 = 1 };
/* power amp selection (crates/openwurli-dsp/Cargo.toml:9-17: `legacy-power-amp` is a default feature; a `--no-default-features` build
 * gets the melange-generated 7-BJT Class-AB solver with rail dynamics, power_amp.rs:279-465 + gen_power_amp.rs) */
enum { OW_POWER_AMP_BEHAVIORAL = 0, OW_POWER_AMP_MELANGE = 1 };
/* tremolo oscillator selection (crates/openwurli-dsp/Cargo.toml:18 `legacy-tremolo`, tremolo.rs:8,53-57,80-90,170-178): the default is the
 * melange-generated Twin-T circuit; the legacy build replaces it by a half-wave rectified 5.63 Hz sine LFO in front of the same CdS model */
enum { OW_TREMOLO_TWIN_T = 0, OW_TREMOLO_LEGACY_LFO = 1 };

/* Introspection block (engine.rs:606-670 test/inspection helpers + diag counters of the solvers). */
typedef struct ow_diag {
    uint32_t active_voices, held_voices, sustained_voices, releasing_voices, steal_voices;
    uint32_t sustain_held;
    uint64_t nan_guard_fires;        /* engine.rs:662-664 */
    uint64_t tremolo_be_fallbacks;   /* gen_tremolo.rs diag_be_fallback_count */
    uint64_t preamp_nan_resets;      /* dk_preamp_legacy.rs:610-615 */
    uint64_t output_nan_resets;      /* engine.rs:450-458 */
} ow_diag;

/* Last error message of the calling thread ("" if none).  The realtime calls return void and never fail (the reference's contract,
 * SURVEY.md 8b): when one of them had to degrade (silence, dropped request) the reason is left here; it stays until the next
 * error replaces it or ow_clear_error() is called. */
const char* ow_last_error(void);
void ow_clear_error(void);

/* ---- pools ------------------------------------------------------------------------------ */
/* n_engines >= 1 engines at `sample_rate` on HIP device `device`.  Like WurliEngine::new
 * (engine.rs:194-229) the engines are NOT warmed up; call ow_pool_set_sample_rate or
 * ow_engine_set_sample_rate (what the plugin's initialize() does, plugin/src/lib.rs:96-97). */
ow_pool* ow_pool_new(double sample_rate, size_t n_engines, int device, int preamp_kind);       /* behavioural power amp */
ow_pool* ow_pool_new_with(double sample_rate, size_t n_engines, int device, int preamp_kind, int power_amp_kind);
ow_pool* ow_pool_new_kinds(double sample_rate, size_t n_engines, int device, int preamp_kind, int power_amp_kind, int tremolo_kind);
void ow_pool_free(ow_pool*);
size_t ow_pool_size(const ow_pool*);
ow_engine* ow_pool_engine(ow_pool*, size_t index);
ow_pool* ow_engine_pool(ow_engine*);              /* the pool an engine belongs to (a pool of one for ow_engine_new) */
/* WurliEngine::set_sample_rate for every engine of the pool (engine.rs:272-286): rebuilds the chain, 0.6 s warm-up. */
int ow_pool_set_sample_rate(ow_pool*, double sample_rate);
/* WurliEngine::reset for every engine (engine.rs:231-251). */
void ow_pool_reset(ow_pool*);
void ow_pool_ensure_buffer_capacity(ow_pool*, size_t max_samples);
/* Render `len` samples on every engine.  out_host: [n_engines][out_stride] f32 (out_stride >= len) or NULL to
 * leave the block in HBM (see ow_pool_device_output).  Blocking; never fails. */
void ow_pool_render(ow_pool*, float* out_host, size_t out_stride, size_t len);
/* Sample-accurate MIDI for many engines in one call: the plugin's handle_event (plugin/src/lib.rs:49-62)
 * applied in array order.  type 0 = NoteOn(note, value=velocity 0..1), 1 = NoteOff(note), 2 = sustain (value >= 0.5 = held).
 * Only the order of the events of one engine matters.  Large lists are applied by several host threads; a list grouped by engine
 * (non-decreasing `engine`) is cut into per-thread slices, any other order makes every thread scan the whole list.
 * A grouped list of >= 65 536 events on a pool of >= 8 192 engines is applied ON THE DEVICE (the voice-pool state machine of
 * engine.rs:299-374 / 569-590 one lane per engine; a list in a block of ow_host_alloc is read where it lies), and its note-ons build
 * their voices before the call returns, as Voice::note_on does (voice.rs:28-110); every other list queues its slot ops for the next
 * render.  Same states, same samples either way (OW_MIDI_DEVICE=0 / OW_MIDI_APPLY_EARLY=0 switch the two steps off). */
typedef struct ow_midi_event {
    uint32_t engine;
    uint8_t type;
    uint8_t note;
    uint16_t reserved;
    float value;
} ow_midi_event;
void ow_pool_midi(ow_pool*, const ow_midi_event* events, size_t n_events);
/* Device pointer of the last rendered block, f32 [n_engines][*stride]; *stride = the length of that block (rows are packed). */
const float* ow_pool_device_output(const ow_pool*, size_t* stride);
/* Stage-wise taps of the last rendered block, copied to host (parity tests): voice sum f64 [n_engines][len]. */
int ow_pool_read_voice_sum(ow_pool*, double* out_host, size_t out_stride, size_t len);
/* Preamp output (main - shadow, before the power amp) of the last block at the chain rate: f64 [n_engines][n_os],
 * n_os = len * (oversampled ? 2 : 1). */
int ow_pool_read_preamp_out(ow_pool*, double* out_host, size_t out_stride, size_t n_os);
/* CdS-cell resistance R[n] the tremolo produced for the last block (before the depth divider), chain rate: f64 [n_engines][n_os]. */
int ow_pool_read_tremolo_r(ow_pool*, double* out_host, size_t out_stride, size_t n_os);
/* The Twin-T / CdS tremolo cell takes no audio and no depth (tremolo.rs:121-146) and Tremolo::new / reset always leave the same settled
 * state (:83-102,:192-216), so its resistance r_ldr[t] is one sequence per chain rate.  The library computes it once per (device, chain
 * rate) into HBM -- the engine with the largest t extends it with a single oscillator, every engine of every pool of the process reads it
 * at its own t -- instead of one oscillator per engine.  Bit-identical to per-engine oscillators (tests/test_gpu_trajectory.py).
 * The store runs ahead of its oldest reader in the background from the moment it exists (Tremolo::new settles inside the constructor,
 * tremolo.rs:83-102: nothing for the host to call), a lead of 60 s of audio by default, on its own stream, one wavefront.  Its buffers
 * start at 150 s of audio (115 MB at 96 kHz; pools of >= 4 096 engines take the whole capacity at once) and double on a helper thread
 * long before a reader gets to their end; an engine older than the capacity (default 1 800 s since its new / reset / set_sample_rate)
 * continues on an oscillator of its own -- same samples, one oscillator per such engine.  OW_TREM_TRAJ=0 when a pool is created: no
 * trajectory for that pool.
 * ow_tremolo_configure: capacity and lead, in seconds of audio, of the stores of `device` (<= 0 / < 0: back to the defaults, which
 * OW_TREM_TRAJ_SECONDS / OW_TREM_TRAJ_LEAD_SECONDS override).  Applies to stores created afterwards and re-limits the existing ones
 * (never below what they hold).  Nothing is reserved by this call.  0 on success.
 * ow_tremolo_prefetch: make the first `seconds` of the trajectory for host rate `sample_rate` exist now (blocking; optional -- a host
 * that wants its first block after instantiation to find everything in place).  Returns the samples known complete, <0 on error. */
int ow_tremolo_configure(int device, double capacity_seconds, double lead_seconds);
long long ow_tremolo_prefetch(double sample_rate, int device, double seconds);
/* Keeping the trajectory ACROSS PROCESSES -- the counterpart of the reference's in-process start-up caches (the OnceLock settles of
 * dk_preamp/melange_adapter.rs:12-29; Tremolo::new's own settle, tremolo.rs:92-102), which a new process pays again.  Optional.
 * ow_tremolo_export: write what the store of (device, chain rate of host rate `sample_rate`) holds, cut to a 4 096-sample checkpoint
 * boundary, to `path` (8 bytes per chain sample).  Returns the samples written, <0 on error.
 * ow_tremolo_import: load such a file into the store (created if it does not exist yet; its 2 s settle is taken from the file too).  The
 * file is only accepted when THIS library would have produced it: same build id, same chain rate, same tremolo constants, payload
 * checksum, and its first and last checkpoint segments regenerated on the device by the product kernel and compared bit for bit.
 * Returns the samples the store took from the file (0: it already held more), <0 when the file is rejected (ow_last_error says why;
 * the store is untouched and extends itself as usual).  Call it where allocating is allowed (instantiation). */
long long ow_tremolo_export(double sample_rate, int device, const char* path);
long long ow_tremolo_import(double sample_rate, int device, const char* path);
/* HIP stream the pool launches on (hipStream_t as void*), for event timing by the caller. */
void* ow_pool_stream(ow_pool*);
/* Time (ms, HIP events on the pool stream) each kernel of the last ow_pool_render took:
 * [0] ops  [1] voices  [2] tremolo  [3] preamp  [4] post.  Enabled by ow_pool_set_profiling(pool,1). */
void ow_pool_set_profiling(ow_pool*, int on);
void ow_pool_last_kernel_ms(const ow_pool*, float ms[5]);

/* ---- engines: the WurliEngine API (engine.rs) -------------------------------------------- */
ow_engine* ow_engine_new(double sample_rate, int device, int preamp_kind);        /* WurliEngine::new        :194 */
ow_engine* ow_engine_new_with(double sample_rate, int device, int preamp_kind, int power_amp_kind);
ow_engine* ow_engine_new_kinds(double sample_rate, int device, int preamp_kind, int power_amp_kind, int tremolo_kind);   /* every cargo feature of the crate as a runtime kind */
void ow_engine_free(ow_engine*);                                                  /* Drop (pool-of-one only)      */
void ow_engine_set_sample_rate(ow_engine*, double sample_rate);                   /* set_sample_rate         :272 */
void ow_engine_reset(ow_engine*);                                                 /* reset                   :231 */
void ow_engine_warm_up(ow_engine*);                                               /* warm_up                 :261 */
void ow_engine_ensure_buffer_capacity(ow_engine*, size_t max_samples);            /* ensure_buffer_capacity  :288 */
void ow_engine_note_on(ow_engine*, uint8_t note, float velocity);                 /* note_on                 :299 */
void ow_engine_note_off(ow_engine*, uint8_t note);                                /* note_off                :340 */
void ow_engine_set_sustain(ow_engine*, int held);                                 /* set_sustain             :361 */
void ow_engine_set_volume(ow_engine*, double v);                                  /* set_volume              :378 */
void ow_engine_set_tremolo_depth(ow_engine*, double depth);                       /* set_tremolo_depth       :382 */
void ow_engine_set_speaker_character(ow_engine*, double c);                       /* set_speaker_character   :386 */
void ow_engine_set_mlp_enabled(ow_engine*, int on);                               /* set_mlp_enabled         :390 */
void ow_engine_set_noise_enabled(ow_engine*, int on);                             /* set_noise_enabled       :394 (melange preamp; no-op on legacy) */
void ow_engine_set_noise_gain(ow_engine*, double gain);                           /* set_noise_gain          :398 (-> set_thermal_gain; no-op on legacy) */
void ow_engine_set_rail_sag(ow_engine*, int on);                                  /* set_rail_sag            :406 (melange power amp; no-op otherwise) */
int ow_engine_rail_sag_enabled(const ow_engine*);                                 /* rail_sag_enabled        :410 */
/* power_amp_diag (engine.rs:418-420: clamp_count, nr_max_iter_count, peak_output_volts of the solver state) plus what the adapter
 * keeps besides: NaN resets of the solver, divergence-guard resets (power_amp.rs:410-421; the reference does not count them) and
 * the rail magnitudes (PowerAmp::rail_voltages, :359-365).  All zero / 22.5 V on the behavioural amp. */
typedef struct ow_power_amp_diag {
    uint64_t clamp_count, nr_max_iter_count;
    double peak_output_volts;
    uint64_t nan_resets, guard_resets;
    double rail_pos_volts, rail_neg_volts;
} ow_power_amp_diag;
void ow_engine_power_amp_diag(const ow_engine*, ow_power_amp_diag* out);
/* gen_preamp::CircuitState::set_seed (gen_preamp.rs:2094-2100) of this engine's main preamp state: restarts its 11 thermal-noise
 * streams from `seed` (0 = the process-wide clock entropy every engine starts from, like the reference) and clears the lag.  Not
 * reachable through WurliEngine in the reference (whose noise is therefore never reproducible); exported so parity can be tested.
 * Survives reset / set_sample_rate.  No-op on the legacy preamp. */
void ow_engine_set_noise_seed(ow_engine*, uint64_t seed);
void ow_engine_render(ow_engine*, float* out, size_t len);                        /* render                  :425 (pool-of-one only) */
void ow_engine_get_diag(const ow_engine*, ow_diag* out);
int ow_engine_slot_state(const ow_engine*, int slot);                             /* VoiceSlot.state              */
int ow_engine_slot_note(const ow_engine*, int slot);                              /* VoiceSlot.midi_note          */
int ow_engine_has_steal_voice_for(const ow_engine*, uint8_t note);                /* has_steal_voice_for     :627 */

/* ---- offline / batch ---------------------------------------------------------------------- */
/* Voice::render_note (voice.rs:191-221): one voice, no chain, f64.  Returns the number of samples
 * of the note ((dur_s*sr) as usize); writes min(n, cap).  Negative on device error. */
long long ow_render_note(uint8_t midi, double velocity, double dur_s, double sample_rate, int device, double* out, size_t cap);
/* Voice::render_note_with_scale(.., Some(displacement_scale)) (voice.rs:201-221): the same with the pickup's displacement scale overridden. */
long long ow_render_note_with_scale(uint8_t midi, double velocity, double dur_s, double sample_rate, double displacement_scale, int device,
                                    double* out, size_t cap);

/* `preamp-bench render` job (tools/preamp-bench/src/main.rs:371-549) as driven by ml/render_model_notes.py:49-116.  One field per flag of
 * the command that changes samples; zero-initialise and set the first seven for what the ML pipeline passes. */
typedef struct ow_job {
    uint8_t note;        /* --note, 33..96 */
    uint8_t velocity;    /* --velocity, 0..127 (vel_norm = velocity/127) */
    uint8_t mlp;         /* !--no-mlp */
    uint8_t poweramp;    /* !--no-poweramp */
    uint8_t no_preamp;               /* --no-preamp: the reed / pickup signal goes straight to the output stage (main.rs:425-427) */
    uint8_t no_attack_noise;         /* --no-attack-noise: Voice::disable_attack_noise (main.rs:409-411, voice.rs:150-152) */
    uint8_t has_displacement_scale;  /* --displacement-scale given (main.rs:388-392) */
    uint8_t reserved;
    double volume;       /* --volume (audio taper: x volume^2) */
    double speaker;      /* --speaker character */
    double r_ldr;        /* --ldr static resistance (used while tremolo_depth <= 0) */
    double tremolo_depth;        /* --tremolo-depth: > 0 puts Tremolo::new(depth, preamp rate) in front of the preamp instead of the static --ldr
                                  * (main.rs:430-441,447-463) */
    double displacement_scale;   /* Voice::set_displacement_scale when has_displacement_scale (main.rs:406-408) */
} ow_job;
typedef struct ow_batch_cfg {
    uint32_t struct_size;  /* = sizeof(ow_batch_cfg) of the caller's header */
    uint32_t job_size;     /* = sizeof(ow_job) of the caller's header (the stride of `jobs`) */
    double sample_rate;  /* --sample-rate (44100 in the reference, main.rs:27) */
    double duration_s;   /* --duration */
    int device;
    int preamp_kind;
    int power_amp_kind;  /* which PowerAmp the command was built with (cargo feature legacy-power-amp, main.rs:480): OW_POWER_AMP_BEHAVIORAL, or
                          * OW_POWER_AMP_MELANGE = PowerAmp::new() = the 7-BJT solver at 44.1 kHz whatever --sample-rate says (power_amp.rs:321-323) */
    int no_rail_sag;     /* --no-rail-sag (main.rs:381,481-483; melange power amp only) */
} ow_batch_cfg;
/* --normalize (main.rs:505-511): the factor write_wav_24bit applies to a finished render, 0.7 / peak when the peak exceeds 0.7, else 1.0;
 * pass it as `scale` to ow_wav24_write / ow_wav24_quantize.  (The samples ow_batch_render returns are never scaled, like final_output.) */
double ow_normalize_scale(const double* samples, size_t n);
/* Renders n_jobs independent jobs lane-parallel (lane = job).  out: f64 [n_jobs][stride], stride >= (duration*sr) as usize.
 * If out_is_device != 0, `out` is a device pointer and nothing is copied to the host.  Returns samples per job, <0 on error. */
long long ow_batch_render(const ow_job* jobs, size_t n_jobs, const ow_batch_cfg* cfg, double* out, size_t stride, int out_is_device);

/* Plain device buffers for chaining the offline entry points without a host round trip (out_is_device / audio_is_device). */
void* ow_device_alloc(size_t bytes, int device);   /* NULL on failure */
void ow_device_free(void* ptr, int device);
/* Page-locked host memory for `out_host` of ow_pool_render: the block copy of a big pool then runs at PCIe rate instead of through
 * the runtime's staging of pageable memory.  (A pool of one copies 2 KB per block; it does not need this.) */
void* ow_host_alloc(size_t bytes, int device);     /* NULL on failure */
void ow_host_free(void* ptr, int device);

/* ---- ML-pipeline stage after the batch render --------------------------------- */
/* 24-bit PCM quantisers of the reference's two WAV writers.  OW_WAV_ROUND: preamp-bench write_wav_24bit
 * (tools/preamp-bench/src/main.rs:941-957): (sample * scale * (2^23-1)).round() as i32, clamped to +-(2^23-1); Rust round() is
 * half-away-from-zero and the cast saturates (NaN -> 0).  OW_WAV_TRUNCATE: reed-renderer write_wav
 * (tools/reed-renderer/src/main.rs:110-126): (s.clamp(-1,1) * (2^23-1)) as i32, truncation toward zero (scale ignored). */
enum { OW_WAV_ROUND = 0, OW_WAV_TRUNCATE = 1 };
int ow_wav24_quantize(const double* samples, size_t n, double scale, int mode, int32_t* out);
/* Mono 24-bit PCM WAV file with that quantiser.  Container as hound 3.5.1 (Cargo.lock) writes it for 24-bit data:
 * RIFF/WAVE, 40-byte WAVE_FORMAT_EXTENSIBLE "fmt " chunk (PCM sub-format, 24 valid bits, channel mask 0x4), "data" chunk of
 * little-endian 3-byte samples.  Returns 0, <0 on I/O error. */
int ow_wav24_write(const char* path, const double* samples, size_t n, uint32_t sample_rate, double scale, int mode);

/* extract_harmonics_fft (ml/goertzel_utils.py:60-107) on many segments at once, as extract_model_features
 * (ml/render_model_notes.py:118-237) applies it to each rendered note: amplitude and frequency of the peak bin of the
 * Hann-windowed, 4x zero-padded spectrum inside +-search_pct of each harmonic h*f0, h = 1..n_harmonics (<= OW_MAX_HARMONICS);
 * harmonics at or above sr/2 - 100 Hz, or without a bin in the band, report amplitude 1e-20 at h*f0.  Also the RMS of every
 * segment (max(sqrt(mean(x^2)), 1e-20); n_harmonics = 0 asks for the RMS only). */
#define OW_MAX_HARMONICS 8
typedef struct ow_segment {
    uint32_t row;          /* audio row (job) */
    uint32_t start, end;   /* sample range [start, end) inside the row */
    uint32_t n_harmonics;  /* 0..OW_MAX_HARMONICS */
    double f0;             /* fundamental the harmonics are searched around */
} ow_segment;
/* audio: f64 [n_rows][stride] (device pointer if audio_is_device != 0, e.g. the output of ow_batch_render).
 * wav24_mode: OW_WAV_ROUND / OW_WAV_TRUNCATE analyse what a 24-bit WAV written with that quantiser and read back as float
 * (int / 2^23, libsndfile's PCM_24 normalisation used by the script's load_audio, goertzel_utils.py:11-17) would contain;
 * OW_WAV_NONE analyses the samples as they are.
 * amps, freqs: [n_segs][OW_MAX_HARMONICS] (entries past n_harmonics are 0); rms: [n_segs] or NULL.  Returns 0, <0 on error. */
#define OW_WAV_NONE (-1)
int ow_extract_harmonics(const double* audio, size_t n_rows, size_t stride, double sample_rate, const ow_segment* segs, size_t n_segs,
                         double search_pct, int wav24_mode, int device, int audio_is_device, double* amps, double* freqs, double* rms);

/* ---- click-band alias audit (crates/openwurli-dsp/src/alias_audit.rs) ----------- */
/* AliasAuditResult (alias_audit.rs:68-93), same fields in the same order. */
#define OW_AUDIT_HARMONICS 12
typedef struct ow_alias_audit_result {
    double f0_hz;                               /* peak of the +-5 Hz / 0.1 Hz search around the nominal pitch */
    double h1_dbfs;                             /* H1 magnitude, dB FS */
    double harmonic_db[OW_AUDIT_HARMONICS];     /* H(i+1), dB FS */
    double harmonic_dbc[OW_AUDIT_HARMONICS];    /* H(i+1) relative to H1, dB; [0] is 0.0 */
    double max_step_up_db;                      /* largest harmonic_dbc[n+1] - harmonic_dbc[n], n over H6..H10 */
    uint32_t max_step_up_from_harmonic;         /* 1-based harmonic the worst rise starts from */
    uint32_t reserved;
    double hf_band_dbc;                         /* RMS of the 5-18 kHz band relative to H1, dB */
} ow_alias_audit_result;
/* analyze (alias_audit.rs:163-211) of n_signals rows at once.  signals: f64 [n_signals][stride], the first `len` samples of a
 * row are the render (device pointer if signals_is_device != 0); the last floor(sample_rate * 0.5) of them are analysed.
 * nominal_f0: [n_signals].  Returns 0; <0 if len is shorter than the analysis window (the reference asserts) or on a device error. */
int ow_alias_audit_analyze(const double* signals, size_t n_signals, size_t stride, size_t len, double sample_rate,
                           const double* nominal_f0, int device, int signals_is_device, ow_alias_audit_result* out);
/* run_with_note for n (note, velocity 0..127) pairs at once (alias_audit.rs:104-108; run_sweep :123-133 is the three
 * STIMULUS_NOTES at velocity 120): one pool engine per pair renders the canonical stimulus (render_stimulus :135-160: 44.1 kHz,
 * volume 0.5, tremolo depth 0, speaker 0, MLP on, noise off, six settling blocks of 1024, note-on, 1.5 s in blocks of 1024), the
 * blocks stay in HBM and are analysed there.  signals_out: NULL or host f64 [n][signals_stride >= 66150] receiving the renders. */
int ow_alias_audit_run(const uint8_t* notes, const uint8_t* velocities, size_t n, int device, int preamp_kind,
                       ow_alias_audit_result* out, double* signals_out, size_t signals_stride);

/* ---- MIDI-file render (`preamp-bench render-midi`, tools/preamp-bench/src/main.rs:1603-1923) ---- */
/* One timed event of the command's internal list (main.rs:1639-1649). */
typedef struct ow_timed_event {
    double time_s;
    uint8_t type;        /* 0 NoteOn(note, value = velocity 1..127)  1 NoteOff(note)  2 Pedal(value != 0 = down) */
    uint8_t note;
    uint8_t value;
    uint8_t reserved[5];
} ow_timed_event;
/* The part of midly 0.5.3 (Cargo.lock) the command uses: Standard MIDI File -> events with absolute times in seconds, in file
 * order (main.rs:1627-1708): metrical timing only, tempo meta events applied per track (every track starts at 500 000 us per
 * beat), note-on with velocity 0 = note-off, controller 64 >= 64 = pedal down.  track_filter < 0: all tracks (`--track N`
 * otherwise).  Returns the event count (writes min(cap, count) events; out may be NULL with cap 0), <0 on malformed data. */
long long ow_smf_parse(const uint8_t* data, size_t len, int track_filter, ow_timed_event* out, size_t cap);
typedef struct ow_midi_render_cfg {
    uint32_t struct_size;  /* = sizeof(ow_midi_render_cfg) of the caller's header */
    uint32_t reserved0;
    double volume;       /* --volume, default 0.60 (applied squared)       */
    double speaker;      /* --speaker, default 1.0                         */
    double tail_s;       /* --tail, default 2.0                            */
    int no_poweramp;     /* --no-poweramp                                  */
    int device;
    int preamp_kind;     /* OW_PREAMP_LEGACY8 (`--model dk` of the default build) or OW_PREAMP_MELANGE12 (melange-preamp build) */
    int power_amp_kind;  /* the build's PowerAmp::new() (main.rs:1756): OW_POWER_AMP_BEHAVIORAL or OW_POWER_AMP_MELANGE (44.1 kHz, rail sag on) */
    int no_rail_sag;     /* PowerAmp::set_rail_sag(false) (not a flag of render-midi; for parity with `render`) */
    int reserved;
} ow_midi_render_cfg;
typedef struct ow_midi_render_stats { uint64_t n_samples, note_ons, peak_polyphony; } ow_midi_render_stats;
/* cmd_render_midi's render loop (main.rs:1711-1891) for n_jobs event lists at once, 44.1 kHz (BASE_SR, main.rs:27).
 * Job j owns events[job_offsets[j] .. job_offsets[j+1]); they are stably sorted by time like the command does.  Its output is
 * floor((last event time + tail_s) * 44100) samples (0 for an empty list, where the command prints and returns).
 * out: host f64 [n_jobs][stride], rows zero-padded behind their job's samples; NULL only fills stats (to size the buffer).
 * Returns the longest job's sample count, <0 on error (e.g. stride too small, NaN times). */
long long ow_render_midi(const ow_timed_event* events, const size_t* job_offsets, size_t n_jobs, const ow_midi_render_cfg* cfg,
                         double* out, size_t stride, ow_midi_render_stats* stats);

#ifdef __cplusplus
}
#endif
#endif /* OPENWURLI_HIP_H */
