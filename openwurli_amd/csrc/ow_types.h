// openwurli-hip: data layout shared by host code and gfx950 kernels.
//
// HBM layout (one pool = I independent engines at one sample rate):
//   voice records   double  vrec[I][2][VF_COUNT][64]   pass 0 = slot voices, pass 1 = steal voices;
//                                                      field-major so the 64 lanes (= 64 voice slots,
//                                                      engine.rs:24 MAX_VOICES) of a wave load 512 B rows
//   chain state     double  cs[CS_COUNT][I]            engine-minor: lane = engine in the chain kernels
//   per-note table  double  note_tab[NT_COUNT][64]     note-only part of Voice::note_on (tables.rs)
//   stream buffers  double  sum[2][I][Lcap]            voice sums (slot pass, steal pass), engine-major
//                   double  rbuf[2][2*Lcap][I]         CdS cell resistance per OS sample, sample-major; two halves: the tremolo
//                                                      oscillator has no audio input and runs one block ahead
//                   double  pre[2*Lcap][I]             preamp out (main - shadow), sample-major
//                   float   out[I][Lcap]               final mono f32 (engine.rs:448)
#pragma once
#include <stdint.h>

#define OW_NUM_MODES 7      /* tables.rs:5 */
#define OW_MAX_VOICES 64    /* engine.rs:24 */
#define OW_MIDI_LO 33       /* tables.rs:6 */
#define OW_MIDI_HI 96       /* tables.rs:7 */
#define OW_MAX_BLOCK 8192   /* engine.rs:25 */

// ---- voice record fields (doubles; integer fields are stored as raw 64-bit patterns) ----
enum {
    VF_S = 0,            // [7] quadrature sine state            (reed.rs:46)
    VF_C = 7,            // [7] quadrature cosine state
    VF_ENV = 14,         // [7] envelope
    VF_DRIFT = 21,       // [7] OU jitter drift
    VF_COS_INC = 28,     // [7]
    VF_SIN_INC = 35,     // [7]
    VF_PHASE_INC = 42,   // [7]
    VF_AMP = 49,         // [7]
    VF_DECAY = 56,       // [7] decay_mult
    VF_DRATE = 63,       // [7] damper_rate
    VF_DMULT = 70,       // [7] damper_mult
    VF_ONSET_INC = 77,   // onset_ramp_inc
    VF_ONSET_EXP = 78,   // onset_shape_exp
    VF_DRAMP = 79,       // damper_ramp_samples
    VF_DCOUNT = 80,      // damper_release_count
    VF_Q = 81,           // pickup charge
    VF_DS = 82,          // pickup displacement_scale
    VF_GAIN = 83,        // post_pickup_gain
    VF_NAMP = 84,        // attack-noise amplitude
    VF_NB0 = 85, VF_NB1 = 86, VF_NB2 = 87, VF_NA1 = 88, VF_NA2 = 89,  // attack-noise BPF coefficients
    VF_NS1 = 90, VF_NS2 = 91,                                        // attack-noise BPF state
    VF_SAMPLE = 92,      // u64 sample counter                      (reed.rs:71)
    VF_ONSET_N = 93,     // u64 onset_ramp_samples
    VF_RNG = 94,         // lo32 jitter_state, hi32 noise rng_state
    VF_NCNT = 95,        // lo32 noise remaining, hi32 noise fade_in_remaining
    VF_FLAGS = 96,       // lo32 flags (bit0 damper_active, bit1 damper_ramp_done), hi32 midi note
    VF_STEAL = 97,       // steal records only: lo32 steal_fade, hi32 steal_fade_len (engine.rs:45-46)
    // constants of the sample rate the voice was struck at: a Voice keeps them for life, also across WurliEngine::set_sample_rate
    VF_BETA = 98,        // pickup beta = 1/(2 sr tau)             (pickup.rs:30-60)
    VF_JREV = 99,        // OU jitter mean reversion                 (reed.rs:150-160)
    VF_JDIFF = 100,      // OU jitter diffusion
    VF_NDECAY = 101,     // attack-noise decay per sample            (hammer.rs:129-133)
    VF_VSR = 102,        // that sample rate (10 s damper safety timeout, voice.rs:183-188)
    VF_COUNT = 103
};
#define OW_VREC_DOUBLES (VF_COUNT * 64)   /* per (engine, pass) */

// ---- per-note table fields (note index = midi - 33) ----
enum {
    NT_F0D = 0,          // detuned fundamental (tables.rs:37 x variation.rs:26)
    NT_RATIO = 1,        // [7] mode ratios
    NT_AMP = 8,          // [7] base amp x spatial coupling (tables.rs:804-830)
    NT_DECAY = 15,       // [7] decay rates dB/s
    NT_AOFF = 22,        // [7] per-note mode amplitude offsets (variation.rs:33)
    NT_DS = 29,          // pickup_displacement_scale (tables.rs:279)
    NT_VEL_EXP = 30,     // velocity_exponent
    NT_F0 = 31,          // undetuned midi_to_freq
    NT_TRIM = 32,        // register_trim_db
    NT_VOICING = 33,     // voicing_slope * max(midi-60,0)
    NT_COUNT = 34
};

// ---- chain state fields (per engine) ----
enum {
    CS_T_V = 0,          // [7] tremolo osc v_prev
    CS_T_I = 7,          // [4] i_nl_prev
    CS_T_IP = 11,        // [4] i_nl_prev_prev
    CS_T_ENV = 15,       // ldr_envelope
    CS_T_RLDR = 16,      // r_ldr (CdS cell)
    CS_T_BE = 17,        // u64 tremolo BE-fallback counter (gen_tremolo.rs diag_be_fallback_count); rows 0..17 = all state k_tremolo touches
    CS_SM_DEPTH = 18,    // [4] tremolo-depth smoother current,target,step,remaining(u64) (engine.rs:67-130)
    CS_SM_SPK = 22,      // [4] speaker-character smoother
    CS_SM_VOL = 26,      // [4] volume smoother
    CS_P_MAIN = 30,      // [14] j_cin, cin_rhs_prev, v[8], i_nl[2], v_nl[2]
    CS_P_SHADOW = 44,    // [14]
    CS_P_RLDR = 58, CS_P_GLDR = 59, CS_P_GPREV = 60,
    CS_OS_UA = 61,       // [3] upsampler branch A state
    CS_OS_UB = 64,       // [3]
    CS_OS_DA = 67,       // [3] downsampler
    CS_OS_DB = 70,       // [3]
    CS_OS_DD = 73,       // down_delay
    CS_SPK_HPF = 74,     // [7] b0 b1 b2 a1 a2 s1 s2
    CS_SPK_LPF = 81,     // [7]
    CS_SPK_CHAR = 88, CS_SPK_A2 = 89, CS_SPK_A3 = 90, CS_SPK_TC = 91, CS_SPK_TS = 92,
    CS_FLAGS = 93,       // u64 bit0: preamp+oversampler reset pending (engine.rs:450-457)
    CS_DIAG = 94,        // u64 counters: hi32 preamp NaN resets
    CS_M_MAIN = 95,      // [21] melange 12-node preamp, main state: v_prev[12] i_nl_prev[3] i_nl_prev_prev[3] input_prev pot word
    CS_M_SHADOW = 116,   // [21] shadow state
    CS_COUNT = 137
};

// ---- slot ops applied before a render (host voice-pool state machine -> device) ----
enum { OP_NOTE_ON = 1, OP_DAMPER = 2, OP_MOVE_STEAL = 3,
       OP_SET_DS = 4 /* Voice::set_displacement_scale(velocity field) on the slot voice (voice.rs:145-147; render_note_with_scale) */ };
struct OwOp {              // 16 bytes
    uint8_t type, slot, note, mlp;
    uint32_t seed;
    double velocity;
};

// ---- pool constants (uniform over engines; computed on the host at pool creation) ----
struct OwConsts {
    double sr, os_sr;
    int oversample, preamp_kind;
    // voice-level rate constants
    double jitter_revert, jitter_diffusion;   // reed.rs:120-122
    double pickup_beta;                        // pickup.rs:105-106
    double noise_decay;                        // hammer.rs:128-129
    uint32_t noise_len;                        // hammer.rs:130
    uint32_t ramp_samples;                     // engine.rs:677-680
    // tremolo oscillator matrices at os_sr (gen_tremolo.rs:2139-2260)
    double t_a_neg[7][7], t_s[7][7], t_k[4][4], t_s_ni[7][4];
    double t_a_neg_be[7][7], t_s_be[7][7], t_k_be[4][4], t_s_ni_be[7][4];
    double ldr_attack, ldr_release, ln_r_max, ln_min_minus_max;  // tremolo.rs:104-112
    double lfo_phase_inc;                      // legacy-tremolo build: 2 pi 5.63 / os_sr (tremolo.rs:76,86)
    uint32_t tremolo_kind, pad_tk;             // OW_TREMOLO_TWIN_T | OW_TREMOLO_LEGACY_LFO
    // legacy DK preamp at os_sr (dk_preamp_legacy.rs:269-366)
    double p_s[8][8], p_a_neg[8][8], p_k[2][2], p_two_w[8], p_s_fb_col[8], p_s_fb_fb, p_nv_sfb[2], p_sfb_ni[2];
    // (s_base[i][EMIT1] - s_base[i][COLL1]), (s_base[i][EMIT2] - s_base[i][COLL2]) of dk_step's last loop (:528-531): differences of
    // constants, formed once on the host by the same f64 subtraction the reference performs per sample
    double p_sni_d1[8], p_sni_d2[8];
    double p_g_cin, p_c_cin, p_gc_1pc;
    double p_g_dc_base[8][8];
    // speaker
    double spk_thermal_alpha;                  // speaker.rs:74
    // melange 12-node preamp at os_sr and the nominal 100 kOhm pot (gen_preamp.rs:1990-2062) + Sherman-Morrison vectors for R_ldr.
    // Field order = struct MelMats (ow_melange_dev.h), copied to LDS as one block.
    double m_s0[12][12], m_aneg0[12][12], m_k0[3][3], m_sni0[12][3];
    double m_u[12], m_w[12], m_wn[3], m_nvu[3], m_s66, m_g_nom;
    double m_noise_scale;    // sqrt(8 k_B T fs_chain), T = 290 K (gen_preamp.rs:1752,1936)
    // Literal per-sample rebuild of the melange preamp (ow_melange_lit.h): the part of invert_n's work that does not depend on R_ldr.
    // R_ldr enters A = G_eff + alpha C in [6][6] only; elimination steps 0..5 never read that entry (row 6 is not a pivot row there, the
    // host checks it), so the factors they produce, the forward substitutions through them and the upper rows are the same for every R.
    // All tables are in POSITION order (after the row exchanges of steps 0..5); positions 6..11 are the trailing block.
    int ml_ok, ml_t6;            // fast path usable; trailing index of original row 6 (whose column-6 entry carries R)
    double ml_chain_m[6], ml_chain_u[6];   // e66 = a66; e66 -= chain_m[k] * chain_u[k], k = 0..5: the eliminations that touch the R entry
    double ml_t0[6][6];          // trailing block after step 5 (entry [ml_t6][0] is replaced by e66)
    double ml_utop[6][12];       // rows 0..5 of U (row i: columns i..11)
    double ml_btop[12][6];       // per unit column: forward-substituted b at positions 0..5
    double ml_part[12][6];       // per unit column: b at trailing position 6+t after the terms j < 6 of its forward substitution
    // column-streamed kernel (ow_melange_col.h): correctly rounded reciprocals of U's R-independent pivots, and whether the factors have
    // the sparsity pattern that kernel compiles in (checked on the host at three resistances; 0 -> the LDS-matrix kernel is used)
    double ml_utop_rcp[6];
    int ml_sparse_ok, ml_pad;
};

#define PA_N 20   /* gen_power_amp.rs:29 */
#define PA_M 16   /* gen_power_amp.rs:38 */
// Pool-uniform constants of the amp at the chain rate (host: ow_consts_host.hpp build_pa_consts).
struct OwPaConsts {
    double a_neg[PA_N][PA_N], s[PA_N][PA_N], k[PA_M][PA_M], s_ni[PA_N][PA_M];
    double a_neg_be[PA_N][PA_N], s_be[PA_N][PA_N], k_be[PA_M][PA_M], s_ni_be[PA_N][PA_M];
    double dc_block_r;
    double alpha_attack, alpha_release, alpha_i_avg;      // RailDynamics::set_sample_rate
    int rate_is_codegen;                                   // |rate - 88 200| <= 0.5: init_state skips set_sample_rate (power_amp.rs:299-301)
    int pad;
    // per-device constants and the constant quotients the device model forms on every call (IEEE divisions done once on the host:
    // the same bits as forming them per call)
    struct Dev {
        double is, vt, sign, vcrit, rb, rc, re;
        double nf_vt, nr_vt, ne_vt, nc_vt;        // n * VT
        double var, vaf, ikf, ikr;
        double is_bf, is_br;                      // is / beta_f, is / beta_r
        double c_dib_fwd, c_dib_rev;              // is / (beta_f * nf_vt), is / (beta_r * nr_vt)
        double ise, isc, c_leak_be, c_leak_bc;    // ise / (ne * vt), isc / (nc * vt)
        double c_dq2_be, c_dq2_bc;                // is / (nf_vt * ikf), is / (nr_vt * ikr)
        double c_dicc_be, c_dicc_bc;              // is / nf_vt, -is / nr_vt
        double max_step;                          // 4 * vt
        // correctly rounded reciprocals of the divisors bjt_evaluate divides by on every call (ow_div_const: same quotient bits)
        double r_nf_vt, r_nr_vt, r_ne_vt, r_nc_vt, r_var, r_vaf, r_ikf, r_ikr;
    } dev[8];
};


// per-engine per-render parameters (host -> device)
// Thermal-noise state of the melange preamp's main solver state, one column per engine in a separate [NZ_COUNT][I] buffer
// (gen_preamp.rs:1708-1745): 11 xoshiro256++ streams, Marsaglia-polar second values, two-draw lag, BE-replay cache.
enum OwNoiseRow {
    NZ_RNG = 0,        // [11][4] u64 bit patterns
    NZ_CACHE = 44,     // [11] cached second gaussian
    NZ_WPREV = 55,     // [11] previous draw (Nyquist-zeroing two-draw stamp)
    NZ_LAST = 66,      // [11] last stamped i_n (replayed by the BE fallback)
    NZ_VALID = 77,     // bit k: NZ_CACHE[k] holds a value
    NZ_SEED = 78,      // u64 master seed of this engine (already resolved: never 0)
    NZ_COUNT = 79
};

struct OwEngineArgs {
    uint64_t main_mask;      // slots with a voice
    uint64_t steal_mask;     // slots with a steal voice
    uint32_t op_begin, op_count;
    // setter targets accepted since the last render (LinearSmoother::set_target, engine.rs:86-99)
    double depth_target, spk_target, vol_target;
    uint32_t set_flags;      // bit0 depth, bit1 spk, bit2 vol
    uint32_t noise_on;       // melange preamp thermal noise enabled (engine.rs:394; block-rate, persists)
    double thermal_gain;     // set_noise_gain -> set_thermal_gain (engine.rs:398-399)
    uint32_t pa_flags;       // melange power amp: bit0 rail sag on (engine.rs:406-408; block-rate, persists)
    uint32_t pad;
};
