// openwurli-hip: device code for the voice layer (gfx950, wave64, f64).
//
//   A voice record = one of the 64 slots of an engine (engine.rs:24), field-major [field][64 slots]; pass 0 = slot voices, pass 1 =
//   steal voices fading out (engine.rs:481-493).  The voice kernels (ow_kernels.h) run one lane per SOUNDING voice, packed across engines.
//
// Mirrors (citations into /root/reference/crates/openwurli-dsp/src/):
//   voice.rs:28-142   Voice::note_on            -> note_on_lane()
//   reed.rs:108-216   ModalReed::new/start_damper
//   reed.rs:219-306   ModalReed::render         -> VoiceRegs::step()
//   hammer.rs:150-198 AttackNoise::render
//   pickup.rs:72-149  pickup_soft_saturate / Pickup::process
//   tables.rs / variation.rs / hammer.rs:26-90 note-only maths -> k_note_table
//   mlp_correction.rs:61-140 MLP forward (scalar lane version here; the MFMA batch version is in ow_mlp_mfma.h)
#pragma once
#include <hip/hip_runtime.h>
#include "ow_types.h"

namespace owdev {

#define OW_DEV __device__ __forceinline__

// IEEE-754 f64 division for the hot loops.  hipcc expands a / b into v_div_scale x2, v_rcp, 2 Newton steps, q = a*y, residual,
// v_div_fmas, v_div_fixup (11 instructions).  The two v_div_scale only pre-scale operands whose exponents are extreme (denormal
// inputs, |exponent difference| near the format's range) and v_div_fmas undoes that; everything the DSP divides lies far inside
// the normal range, where the scale factors are exactly 1 and the sequence below is the same arithmetic, bit for bit, without them.
// v_div_fixup is kept: zero / infinite / NaN operands and the sign of a zero quotient come out as IEEE requires.
// tests/test_gpu_division.py compares it with `a / b` on the device and with numpy for 2^24 operand pairs per exponent range.
// OW_IEEE_DIV restores the compiler's expansion.
OW_DEV double ow_div(double a, double b) {
#ifdef OW_IEEE_DIV
    return a / b;
#else
    double y = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    const double q = a * y;
    const double r = __builtin_fma(-b, q, a);
    return __builtin_amdgcn_div_fixup(__builtin_fma(r, y, q), b, a);
#endif
}
// The two halves of ow_div, for several quotients over ONE divisor (pivot rows of the solvers): the refined reciprocal depends on the
// divisor alone, so y = ow_rcp_refined(b) once and ow_div_y(a, b, y) per numerator is ow_div(a, b) instruction for instruction.
OW_DEV double ow_rcp_refined(double b) {
#ifdef OW_IEEE_DIV
    return b;
#else
    double y = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    return y;
#endif
}
OW_DEV double ow_div_y(double a, double b, double y) {
#ifdef OW_IEEE_DIV
    return a / b;
#else
    const double q = a * y;
    const double r = __builtin_fma(-b, q, a);
    return __builtin_amdgcn_div_fixup(__builtin_fma(r, y, q), b, a);
#endif
}
// Division by a compile-time constant: y = 1.0 / B is folded by the compiler (correctly rounded), which leaves q = a*y, the exact
// residual and the final correction of the sequence above -- the quotient is again the correctly rounded a / B (Markstein's
// correction step with a correctly rounded reciprocal); v_div_fixup keeps zero / infinite / NaN numerators IEEE.  Checked on the
// device against `a / B` for every constant the kernels use (tests/test_gpu_division.py; all 2^31 numerators of the jitter draw).
OW_DEV double ow_div_const(double a, const double b, const double y) {
#ifdef OW_IEEE_DIV
    return a / b;
#else
    const double q = a * y;
    const double r = __builtin_fma(-b, q, a);
    return __builtin_amdgcn_div_fixup(__builtin_fma(r, y, q), b, a);
#endif
}
#define OW_DIV_C(a, B) ow_div_const((a), (B), 1.0 / (B))
#define OW_JITTER_DIV 2147483647.5            // reed.rs:267-272
OW_DEV double clampd(double x, double lo, double hi) {  // Rust f64::clamp (NaN propagates)
    return x < lo ? lo : (x > hi ? hi : x);
}
OW_DEV uint64_t dbits(double x) { return (uint64_t)__double_as_longlong(x); }
OW_DEV double bitsd(uint64_t b) { return __longlong_as_double((long long)b); }
OW_DEV uint32_t lcg(uint32_t s) { return s * 1664525u + 1013904223u; }  // reed.rs:91, hammer.rs:192-195
OW_DEV uint64_t sat_u64(double x) {  // Rust `as u64`
    if (!(x == x) || x <= 0.0) return 0ull;
    if (x >= 18446744073709551615.0) return ~0ull;
    return (uint64_t)x;
}

// ------------------------------------------------------------------ per-note table
// One thread per MIDI note 33..96.  tables.rs:804-830 + variation.rs:26-38 + the note-only
// scalars Voice::note_on needs (ds, velocity exponent, trim, voicing).
OW_DEV double tip_mass_ratio(double m) {  // tables.rs:51-77
    const double ax[5] = {33.0, 52.0, 62.0, 74.0, 96.0};
    const double ay[5] = {0.10, 0.00, 0.00, 0.02, 0.01};
    if (m <= ax[0]) return ay[0];
    if (m >= ax[4]) return ay[4];
    double r = 0.0;
    bool done = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (!done && m <= ax[i + 1]) {
            const double t = (m - ax[i]) / (ax[i + 1] - ax[i]);
            r = ay[i] + t * (ay[i + 1] - ay[i]);
            done = true;
        }
    }
    return r;
}

__device__ const double EIG_MU[8] = {0.00, 0.01, 0.05, 0.10, 0.15, 0.20, 0.30, 0.50};
__device__ const double EIG_BETA[8][7] = {  // tables.rs:91-124
    {1.8751, 4.6941, 7.8548, 10.9955, 14.1372, 17.2788, 20.4204}, {1.8584, 4.6849, 7.8504, 10.9930, 14.1356, 17.2776, 20.4195},
    {1.7920, 4.6477, 7.8316, 10.9830, 14.1288, 17.2726, 20.4158}, {1.7227, 4.6024, 7.8077, 10.9700, 14.1198, 17.2660, 20.4110},
    {1.6625, 4.5618, 7.7859, 10.9580, 14.1114, 17.2598, 20.4065}, {1.6097, 4.5254, 7.7659, 10.9470, 14.1036, 17.2540, 20.4023},
    {1.5201, 4.4620, 7.7310, 10.9280, 14.0894, 17.2434, 20.3946}, {1.3853, 4.3601, 7.6745, 10.8970, 14.0650, 17.2252, 20.3814},
};
__device__ inline void eigenvalues(double mu, double out[7]) {  // tables.rs:85-143
    const double mc = clampd(mu, 0.0, 0.50);
    int lo = 0;
    for (int i = 7; i >= 0; --i)
        if (EIG_MU[i] <= mc) { lo = i; break; }
    const int hi = lo + 1 < 7 ? lo + 1 : 7;
    const double t = (EIG_MU[hi] > EIG_MU[lo]) ? (mc - EIG_MU[lo]) / (EIG_MU[hi] - EIG_MU[lo]) : 0.0;
    for (int i = 0; i < 7; ++i) out[i] = EIG_BETA[lo][i] + t * (EIG_BETA[hi][i] - EIG_BETA[lo][i]);
}
__device__ inline double mode_shape(double beta, double xi) {  // tables.rs:295-299
    const double sigma = (cosh(beta) + cos(beta)) / (sinh(beta) + sin(beta));
    const double bx = beta * xi;
    return cosh(bx) - cos(bx) - sigma * (sinh(bx) - sin(bx));
}
__device__ inline double reed_length_mm(double midi) {  // tables.rs:161-169
    const double n = clampd(midi - 32.0, 1.0, 64.0);
    const double inches = (n <= 20.0) ? 3.0 - n / 20.0 : 2.0 - (n - 20.0) / 44.0;
    return inches * 25.4;
}
__device__ inline double reed_compliance(int midi) {  // tables.rs:183-225
    int reed = midi - 32;
    reed = reed < 1 ? 1 : (reed > 64 ? 64 : reed);
    const double width_inch = reed <= 14 ? 0.151 : reed <= 20 ? 0.127 : reed <= 42 ? 0.121 : reed <= 50 ? 0.111 : 0.098;
    double thick_inch;
    if (reed <= 16) thick_inch = 0.026;
    else if (reed <= 26) thick_inch = 0.026 + (((double)reed - 16.0) / 10.0) * (0.034 - 0.026);
    else thick_inch = 0.034;
    const double l = reed_length_mm((double)midi), w = width_inch * 25.4, t = thick_inch * 25.4;
    return (l * l * l) / (w * t * t * t);
}
__device__ inline double hash01(uint32_t midi, uint32_t seed) {  // variation.rs:10-19
    uint32_t h = 2166136261u;
    h ^= midi; h *= 16777619u;
    h ^= seed; h *= 16777619u;
    h ^= h >> 16; h *= 2654435769u;
    return (double)(h & 0x00FFFFFFu) / 16777216.0;
}
__device__ inline double register_trim_db(double m) {  // tables.rs:443-481
    const double ax[13] = {36, 40, 44, 48, 52, 56, 60, 64, 68, 72, 76, 80, 84};
    const double ay[13] = {-1.3, 0.0, -1.3, 0.7, 0.2, -1.0, 0.0, 0.9, 1.2, 0.0, 1.8, 2.4, 3.6};
    if (m <= ax[0]) return ay[0];
    if (m >= ax[12]) return ay[12];
    for (int i = 0; i < 12; ++i)
        if (m <= ax[i + 1]) return ay[i] + ((m - ax[i]) / (ax[i + 1] - ax[i])) * (ay[i + 1] - ay[i]);
    return 0.0;
}

__global__ void k_note_table(double* __restrict__ nt) {
    const int ni = threadIdx.x;
    if (ni >= 64) return;
    const int midi = OW_MIDI_LO + ni;
    const double m = (double)midi;
    const double f0 = 440.0 * pow(2.0, (m - 69.0) / 12.0);                        // tables.rs:37-39
    const double detune = 1.0 + (hash01((uint32_t)midi, 0xDEADu) * 2.0 - 1.0) * 0.00173;  // variation.rs:26-29
    nt[NT_F0 * 64 + ni] = f0;
    nt[NT_F0D * 64 + ni] = f0 * detune;
    const double mu = tip_mass_ratio(m);
    double betas[7], ratios[7];
    eigenvalues(mu, betas);
    const double b1sq = betas[0] * betas[0];
    for (int i = 0; i < 7; ++i) ratios[i] = (betas[i] * betas[i]) / b1sq;          // tables.rs:149-153
    const double base_decay = fmax(0.005 * pow(f0, 1.22), 3.0);                    // tables.rs:391-396
    // spatial pickup coupling, 32-interval Simpson (tables.rs:324-370)
    const double ell = clampd(6.0 / reed_length_mm(m), 0.0, 1.0);
    const double xi_start = 1.0 - ell;
    double kraw[7];
    for (int mode = 0; mode < 7; ++mode) {
        const double beta = betas[mode];
        const double tip = mode_shape(beta, 1.0);
        if (fabs(tip) < 1e-30 || ell < 1e-12) { kraw[mode] = 1.0; continue; }
        const double h = ell / 32.0;
        double sum = mode_shape(beta, xi_start) + mode_shape(beta, 1.0);
        for (int j = 1; j < 32; ++j) {
            const double xi = xi_start + (double)j * h;
            sum += ((j & 1) ? 4.0 : 2.0) * mode_shape(beta, xi);
        }
        const double integral = sum * h / 3.0;
        kraw[mode] = clampd(fabs(integral / (ell * tip)), 0.0, 1.0);
    }
    const double base_amp[7] = {1.0, 0.005, 0.0035, 0.0018, 0.0011, 0.0007, 0.0005};  // tables.rs:33-34
    const double k1 = kraw[0];
    for (int i = 0; i < 7; ++i) {
        const double kap = (k1 > 1e-30) ? clampd(kraw[i] / k1, 0.0, 1.0) : 1.0;
        nt[(NT_RATIO + i) * 64 + ni] = ratios[i];
        nt[(NT_AMP + i) * 64 + ni] = base_amp[i] * kap;
        nt[(NT_DECAY + i) * 64 + ni] = base_decay * ratios[i] * ratios[i];           // tables.rs:418-422
        nt[(NT_AOFF + i) * 64 + ni] = 1.0 + (hash01((uint32_t)midi, 0xBEEFu + (uint32_t)i) * 2.0 - 1.0) * 0.08;
    }
    const double ds = 0.85 * pow(reed_compliance(midi) / reed_compliance(60), 0.75);  // tables.rs:279-288
    nt[NT_DS * 64 + ni] = clampd(ds, 0.02, 0.95);
    {   // velocity_exponent, tables.rs:535-554
        const double z = (m - 62.0) / 15.0;
        const double t = exp(-0.5 * (z * z));
        const double mn = (m < 62.0) ? 0.55 : 1.3;
        nt[NT_VEL_EXP * 64 + ni] = mn + t * (1.7 - mn);
    }
    nt[NT_TRIM * 64 + ni] = register_trim_db(m);
    nt[NT_VOICING * 64 + ni] = -0.04 * fmax(m - 60.0, 0.0);                        // tables.rs:522
}

// ------------------------------------------------------------------ note-on (one lane = one slot)
__device__ inline double pickup_rms_proxy(double ds, double f0, double fc) {  // tables.rs:424-441
    if (ds < 1e-10) return 0.0;
    const double r = (1.0 - sqrt(1.0 - ds * ds)) / ds;
    const double inv_sqrt = 1.0 / sqrt(1.0 - ds * ds);
    double sum_sq = 0.0, r_n = r;
    for (int n = 1; n <= 8; ++n) {
        const double cn = 2.0 * r_n * inv_sqrt;
        const double nf = (double)n * f0;
        const double hpf_n = nf / sqrt(nf * nf + fc * fc);
        sum_sq += (cn * hpf_n) * (cn * hpf_n);
        r_n *= r;
    }
    return sqrt(sum_sq);
}
__device__ inline double velocity_scurve(double v) {  // tables.rs:556-562
    const double k = 1.5;
    const double s = 1.0 / (1.0 + exp(-k * (v - 0.5)));
    const double s0 = 1.0 / (1.0 + exp(k * 0.5));
    const double s1 = 1.0 / (1.0 + exp(-k * 0.5));
    return (s - s0) / (s1 - s0);
}

struct MlpOut { double cents[5], decay[5], ds; };

// Scalar forward pass, same accumulation order as mlp_correction.rs:86-116.
__device__ inline void mlp_raw_scalar(double in0, double in1, double raw[11]);

__device__ inline MlpOut mlp_finish(int midi_note, const double raw[11], bool enabled) {  // mlp_correction.rs:61-140
    MlpOut o;
    for (int i = 0; i < 5; ++i) { o.cents[i] = 0.0; o.decay[i] = 1.0; }
    o.ds = 1.0;
    if (!enabled) return o;
    const double midi = (double)midi_note;
    double fade;
    if (midi < 65.0) fade = clampd((midi - (65.0 - 12.0)) / 12.0, 0.0, 1.0);
    else if (midi > 97.0) fade = clampd(((97.0 + 12.0) - midi) / 12.0, 0.0, 1.0);
    else fade = 1.0;
    if (fade <= 0.0) return o;
    for (int h = 0; h < 5; ++h) o.cents[h] = clampd(raw[h] * fade, -100.0, 100.0);
    for (int h = 0; h < 5; ++h) o.decay[h] = 1.0 + (clampd(raw[5 + h], 0.3, 3.0) - 1.0) * fade;
    o.ds = 1.0 + (clampd(raw[10], 0.7, 1.2) - 1.0) * fade;
    return o;
}

// Writes a fresh voice into record `rec` (pointer already offset by slot; field stride 64).
__device__ inline void note_on_lane(double* __restrict__ rec, const double* __restrict__ nt, const OwConsts* __restrict__ K, int note,
                                    double vel, uint32_t seed, const MlpOut& corr) {
    const int ni = note - OW_MIDI_LO;
    const double sr = K->sr;
    const double f0d = nt[NT_F0D * 64 + ni];
    // dwell filter (hammer.rs:26-29, 69-90) on the uncorrected ratios
    const double t_dwell = clampd((0.75 + 0.25 * (1.0 - vel)) / f0d, 0.0003, 0.020);
    double a0;
    {
        const double ft = f0d * nt[NT_RATIO * 64 + ni] * t_dwell;
        a0 = exp(-ft * ft / (2.0 * (8.0 * 8.0)));
    }
    const double onset_time = fmax((1.0 + 1.0 * (1.0 - vel)) * (1.0 / f0d), 0.002);  // hammer.rs:53-57
    const double scurve = velocity_scurve(vel);
    const double vel_scale = pow(scurve, nt[NT_VEL_EXP * 64 + ni]);
    const double base_ds = nt[NT_DS * 64 + ni];
    const double corrected_ds = base_ds * corr.ds;
    // One ROLLED loop over the seven modes (dwell attenuation, MLP corrections voice.rs:68-84, ModalReed::new reed.rs:108-182): every
    // value is the expression it was when the steps were seven-wide loops of their own, but the exponentials, powers, logarithms and
    // sines of a mode are in the code once instead of seven times side by side -- unrolled, k_apply_ops wanted > 512 registers, spilled
    // 430 of them at its cap of 256 and a wavefront spent 0.55 ms waiting for its own scratch memory (a whole-pool re-strike: 18 ms).
    uint32_t js = seed > 1u ? seed : 1u;
#pragma unroll 1
    for (int i = 0; i < 7; ++i) {
        double ratio = nt[(NT_RATIO + i) * 64 + ni], dec = nt[(NT_DECAY + i) * 64 + ni];
        const double ft = f0d * ratio * t_dwell;
        double att = exp(-ft * ft / (2.0 * (8.0 * 8.0)));
        if (a0 > 1e-30) att /= a0;
        double amp = nt[(NT_AMP + i) * 64 + ni] * att * nt[(NT_AOFF + i) * 64 + ni];
        amp *= vel_scale;
        if (i >= 1 && i <= 5) {                                                     // voice.rs:68-84
            ratio *= pow(2.0, corr.cents[i - 1] / 1200.0);
            dec /= corr.decay[i - 1];
        }
        js = lcg(js);
        const double u1 = (double)(js >> 1) / 2147483647.5;
        js = lcg(js);
        const double u2 = (double)(js >> 1) / 2147483647.5;
        const double r = sqrt(-2.0 * log(fmax(u1, 1e-30)));
        const double drift = 0.0004 * r * cos(6.28318530717958647692 * u2);
        const double freq = f0d * ratio;
        const double phase_inc = 6.28318530717958647692 * freq / sr;
        const double decay_per_sample = (dec / 8.686) / sr;
        rec[(VF_S + i) * 64] = 0.0;
        rec[(VF_C + i) * 64] = 1.0;
        rec[(VF_ENV + i) * 64] = 1.0;
        rec[(VF_DRIFT + i) * 64] = drift;
        rec[(VF_COS_INC + i) * 64] = cos(phase_inc);
        rec[(VF_SIN_INC + i) * 64] = sin(phase_inc);
        rec[(VF_PHASE_INC + i) * 64] = phase_inc;
        rec[(VF_AMP + i) * 64] = amp;
        rec[(VF_DECAY + i) * 64] = exp(-decay_per_sample);
        rec[(VF_DRATE + i) * 64] = 0.0;
        rec[(VF_DMULT + i) * 64] = 1.0;
    }
    const uint64_t ramp = sat_u64(round(onset_time * sr));
    rec[VF_ONSET_N * 64] = bitsd(ramp);
    rec[VF_ONSET_INC * 64] = ramp > 0 ? 3.14159265358979323846 / (double)ramp : 0.0;
    rec[VF_ONSET_EXP * 64] = 1.0 + (1.0 - vel);
    rec[VF_DRAMP * 64] = 0.0;
    rec[VF_DCOUNT * 64] = 0.0;
    rec[VF_SAMPLE * 64] = bitsd(0ull);
    rec[VF_Q * 64] = 1.0;                                                            // pickup.rs:103-113
    rec[VF_DS * 64] = corrected_ds;

    // AttackNoise::new (hammer.rs:126-147) + RBJ constant-skirt band-pass (filters.rs:15-21)
    rec[VF_NAMP * 64] = 0.025 * vel * vel;
    rec[VF_BETA * 64] = K->pickup_beta; rec[VF_JREV * 64] = K->jitter_revert; rec[VF_JDIFF * 64] = K->jitter_diffusion;
    rec[VF_NDECAY * 64] = K->noise_decay; rec[VF_VSR * 64] = sr;
    {
        const double center = clampd(f0d * 5.0, 200.0, 2000.0);
        const double w0 = 2.0 * 3.14159265358979323846 * center / sr;
        const double cw = cos(w0), sw = sin(w0);
        const double alpha = sw / (2.0 * 0.7);
        const double a0b = 1.0 + alpha;
        rec[VF_NB0 * 64] = (sw / 2.0) / a0b;
        rec[VF_NB1 * 64] = 0.0 / a0b;
        rec[VF_NB2 * 64] = (-sw / 2.0) / a0b;
        rec[VF_NA1 * 64] = (-2.0 * cw) / a0b;
        rec[VF_NA2 * 64] = (1.0 - alpha) / a0b;
        rec[VF_NS1 * 64] = 0.0;
        rec[VF_NS2 * 64] = 0.0;
    }
    rec[VF_RNG * 64] = bitsd((uint64_t)js | ((uint64_t)seed << 32));
    rec[VF_NCNT * 64] = bitsd((uint64_t)K->noise_len | (16ull << 32));
    rec[VF_FLAGS * 64] = bitsd((uint64_t)0u | ((uint64_t)(uint32_t)note << 32));

    // post-pickup gain: output_scale (tables.rs:489-533) x MLP level compensation (voice.rs:108-127)
    const double HPF_FC = 2312.0;
    const double f0 = nt[NT_F0 * 64 + ni];
    const double f0_c4 = nt[NT_F0 * 64 + (60 - OW_MIDI_LO)];
    const double vel_scale_c4 = pow(scurve, nt[NT_VEL_EXP * 64 + (60 - OW_MIDI_LO)]);
    const double eff_ds = fmax(base_ds * vel_scale, 1e-6);
    const double eff_ds_ref = fmax(0.85 * vel_scale_c4, 1e-6);
    const double rms = pickup_rms_proxy(eff_ds, f0, HPF_FC);
    const double rms_ref = pickup_rms_proxy(eff_ds_ref, f0_c4, HPF_FC);
    const double flat_db = -20.0 * log10(rms / rms_ref);
    const double eff_trim = nt[NT_TRIM * 64 + ni] * pow(vel, 1.3);
    const double out_scale = pow(10.0, (-35.0 + flat_db + nt[NT_VOICING * 64 + ni] + eff_trim) / 20.0);
    double comp = 1.0;
    if (fabs(corr.ds - 1.0) > 1e-6) {
        const double pb = pickup_rms_proxy(base_ds, f0, HPF_FC);
        const double pc = pickup_rms_proxy(corrected_ds, f0, HPF_FC);
        comp = (pc > 1e-10) ? sqrt(pb / pc) : 1.0;
    }
    rec[VF_GAIN * 64] = out_scale * comp;
    rec[VF_STEAL * 64] = bitsd(0ull);
}

// reed.rs:191-216 on the record in memory
__device__ inline void start_damper_lane(double* __restrict__ rec, const OwConsts* __restrict__ K) {
    const uint64_t fl = dbits(rec[VF_FLAGS * 64]);
    const int midi = (int)(fl >> 32);
    if (midi >= 92) return;
    const double sr = rec[VF_VSR * 64];   // Voice::note_off passes the voice's own rate (voice.rs:157), not the engine's current one
    const double base_rate = fmax(55.0 * pow(2.0, ((double)midi - 60.0) / 24.0), 0.5);
    double p3 = 1.0;
    for (int m = 0; m < 7; ++m) {
        const double rate = fmin(base_rate * p3, 2000.0) / sr;
        rec[(VF_DRATE + m) * 64] = rate;
        rec[(VF_DMULT + m) * 64] = exp(-rate);
        p3 *= 3.0;
    }
    const double ramp_time = midi < 48 ? 0.050 : (midi < 72 ? 0.025 : 0.008);
    rec[VF_DRAMP * 64] = ramp_time * sr;
    rec[VF_DCOUNT * 64] = 0.0;
    rec[VF_FLAGS * 64] = bitsd(((fl & 0xFFFFFFFF00000000ull) | 1ull));  // damper_active=1, ramp_done=0
}

// Rare-phase transcendentals are kept out of line so their temporaries do not inflate the register
// footprint of the steady-state loop (onset ramp: first ~1-2 periods; damper ramp: 8-50 ms after note-off;
// attack noise fade-in: 16 samples; pickup saturation: only |y| >= 0.94).
// cos on [0, pi] (the onset ramp's phase n * pi / N, n < N): quadrant by Cody-Waite with the first 33 bits of pi/2 (k <= 2: the product is
// exact) and the fdlibm kernels (k_cos.c / k_sin.c polynomials) with explicit fused steps; <= 1.5 ulp.  The library's cos carries a
// Payne-Hanek path for huge arguments and costs ~150 instructions per lane and sample of an onset ramp; this is ~35.
// fma(a, b, k) with the CONSTANT k read from a scalar register pair (v_fma_f64's third operand).  Left to itself the compiler turns a
// Horner step with a literal coefficient into two v_mov_b32 (the literal into the destination) + v_fmac_f64: three vector issue slots per
// step instead of one -- a third of onset_gain's 150 vector instructions were such moves.  The s_mov pair that fills the scalar
// register issues beside the other wavefront's vector work.  Same operation, same bits.
OW_DEV double fma_k(double a, double b, double k) {
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(k));
    return r;
}
OW_DEV double cos_0_pi(double x) {
    const double kf = floor(x * 6.36619772367581382433e-01 + 0.5);                      // 0, 1 or 2
    const double r = __builtin_fma(-kf, 6.07710050650619224932e-11, __builtin_fma(-kf, 1.57079632673412561417e+00, x));
    const double z = r * r;
    // cos(r)
    double pc = __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma_k(z, pc, -2.75573143513906633035e-07);
    pc = fma_k(z, pc, 2.48015872894767294178e-05);
    pc = fma_k(z, pc, -1.38888888888741095749e-03);
    pc = fma_k(z, pc, 4.16666666666666019037e-02);
    const double hz = 0.5 * z;
    const double w = 1.0 - hz;
    const double cr = w + (((1.0 - w) - hz) + z * (z * pc));                              // 1 - z/2 + z^2 pc with the rounding error of 1 - z/2 fed back
    // sin(r)
    double ps = __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma_k(z, ps, 2.75573137070700676789e-06);
    ps = fma_k(z, ps, -1.98412698298579493134e-04);
    ps = fma_k(z, ps, 8.33333333332248946124e-03);
    const double sr = __builtin_fma(z * r, fma_k(z, ps, -1.66666666666666324348e-01), r);
    return kf == 0.0 ? cr : (kf == 1.0 ? -sr : -cr);
}
// ln on (0, 1] (fdlibm e_log.c: x = 2^k (1 + f), s = f / (2 + f), the Lg1..Lg7 series, k ln2 split in hi / lo); <= 1 ulp.  Normal
// arguments only (the onset cosine is 0 -- handled by the caller -- or >= ~1e-8).
OW_DEV double log_unit(double x) {
    int k = __builtin_amdgcn_frexp_exp(x);                                               // x = m 2^k, m in [0.5, 1)
    double m = __builtin_amdgcn_frexp_mant(x);
    if (m < 7.07106781186547524401e-01) { m += m; k -= 1; }                              // m in [sqrt(1/2), sqrt(2))
    const double f = m - 1.0;
    const double s = ow_div(f, 2.0 + f);
    const double z = s * s, w = z * z;
    const double t1 = w * fma_k(w, __builtin_fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * fma_k(w, fma_k(w, __builtin_fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01),
                                6.666666666666735130e-01);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)k;
    return dk * 6.93147180369123816490e-01 - ((hfsq - (s * (hfsq + R) + dk * 1.90821492927058770002e-10)) - f);
}
// The library's exp (ocml: n = rint(x log2 e), two-step reduction by ln 2, degree-11 polynomial, ldexp -- the constants and operations of
// exp_bounded, ow_chain_dev.h) for -700 < x <= 0 -- the onset ramp's p ln(cosine) >= -75 -- without the overflow / underflow
// selects, which cannot fire there, and with the coefficients in scalar registers (fma_k).  Bit-identical to exp() on that range.
OW_DEV double exp_nonpos(double x) {
    const double n = rint(x * __longlong_as_double(0x3ff71547652b82feLL));
    double r = __builtin_fma(__longlong_as_double((long long)0xbfe62e42fefa39efULL), n, x);
    r = __builtin_fma(__longlong_as_double((long long)0xbc7abc9e3b39803fULL), n, r);
    double p = __builtin_fma(__longlong_as_double(0x3e5ade156a5dcb37LL), r, __longlong_as_double(0x3e928af3fca7ab0cLL));
    p = fma_k(r, p, __longlong_as_double(0x3ec71dee623fde64LL));
    p = fma_k(r, p, __longlong_as_double(0x3efa01997c89e6b0LL));
    p = fma_k(r, p, __longlong_as_double(0x3f2a01a014761f6eLL));
    p = fma_k(r, p, __longlong_as_double(0x3f56c16c1852b7b0LL));
    p = fma_k(r, p, __longlong_as_double(0x3f81111111122322LL));
    p = fma_k(r, p, __longlong_as_double(0x3fa55555555502a1LL));
    p = fma_k(r, p, __longlong_as_double(0x3fc5555555555511LL));
    p = fma_k(r, p, __longlong_as_double(0x3fe000000000000bLL));
    p = __builtin_fma(r, p, 1.0);
    p = __builtin_fma(r, p, 1.0);
    return ldexp(p, (int)n);
}
__device__ __noinline__ __attribute__((const)) double onset_gain(double n, double onset_inc, double onset_exp) {  // reed.rs:251-264
#ifdef OW_LIB_POW
    const double cosine = 0.5 * (1.0 - cos(n * onset_inc));
    if (onset_exp <= 1.001) return cosine;
    if (onset_exp >= 1.999) return cosine * cosine;
    return pow(cosine, onset_exp);
#else
    // The ramp's phase n * inc lies in [0, pi): a cosine for that interval alone (cos_0_pi).
    const double cosine = 0.5 * (1.0 - cos_0_pi(n * onset_inc));
    if (onset_exp <= 1.001) return cosine;
    if (onset_exp >= 1.999) return cosine * cosine;
    // cosine^p as exp(p ln cosine), cosine in [0, 1), p in (1.001, 1.999).  The library's pow carries ln in double-double to stay below
    // 1 ulp for every argument (~215 instructions; every lane of a re-struck wavefront pays it on every sample of its onset ramp).
    // Here the result is a GAIN in [0, 1): the error of p ln c is |p ln c| eps relative on the result, i.e. c^p |p ln c| eps absolute,
    // whose maximum over c is eps / e = 4e-17 -- below half an ulp of the gains near 1 that carry the signal, and the reference's own
    // f64::powf (glibc) is only specified to 1 ulp.  pow(0, p) = 0.  (tests/test_gpu_division.py::test_onset_gain_accuracy)
    if (!(cosine > 0.0)) return 0.0;
    return exp_nonpos(onset_exp * log_unit(cosine));   // cosine = (1 - cos) / 2 is 0 or >= 2^-54: p ln(cosine) >= -75
#endif
}
__device__ __noinline__ __attribute__((const)) double exp_neg(double x) { return exp(-x); }                         // reed.rs:238
// exp(-x) for the damper ramp's per-sample factors (reed.rs:236-239): x = damper_rate * t / ramp <= damper_rate = min(55 * 2^((n-60)/24)
// * 3^m, 2000) / sr (reed.rs:198-201), i.e. 0 <= x <= 0.0454 at 44.1 kHz.  On [0, 1/8] no range reduction is needed: the degree-11 Taylor
// polynomial in Horner form (fused steps) has a truncation error of x^12 / 12! <= 3e-20 and ends in fma(y, p, 1.0), one rounding of a
// value in (0.88, 1]: within 0.57 ulp of exp(-x) (measured, 2^23 arguments), where the library's exp (the same kind of polynomial behind a reduction by ln 2, an
// ldexp and overflow / underflow selects: ~45 instructions behind a call, seven times per sample of every released voice for the 8-50 ms
// of its ramp -- the cost of a whole-keyboard re-strike's steal pass) is specified to 1 ulp, like the reference's f64::exp (glibc).
// Larger arguments (host rates below 16 kHz) take the library.  tests/test_gpu_division.py::test_damper_ramp_exp_accuracy.
OW_DEV double exp_neg_poly(double x) {          // 0 <= x <= 1/8
    const double y = -x;
    double p = __builtin_fma(y, 2.50521083854417187751e-08, 2.75573192239858906526e-07);       // 1/11!, 1/10!
    p = __builtin_fma(y, p, 2.75573192239858906526e-06);                                        // 1/9!
    p = __builtin_fma(y, p, 2.48015873015873015873e-05);                                        // 1/8!
    p = __builtin_fma(y, p, 1.98412698412698412698e-04);                                        // 1/7!
    p = __builtin_fma(y, p, 1.38888888888888888889e-03);                                        // 1/6!
    p = __builtin_fma(y, p, 8.33333333333333333333e-03);                                        // 1/5!
    p = __builtin_fma(y, p, 4.16666666666666666667e-02);                                        // 1/4!
    p = __builtin_fma(y, p, 1.66666666666666666667e-01);                                        // 1/3!
    p = __builtin_fma(y, p, 0.5);
    p = __builtin_fma(y, p, 1.0);
    return __builtin_fma(y, p, 1.0);
}
OW_DEV double exp_neg_small(double x) { return __builtin_expect(!(x <= 0.125), 0) ? exp_neg(x) : exp_neg_poly(x); }
// The instantaneous damper rate of reed.rs:237, `damper_rate * t / ramp`, as damper_rate * (t / ramp): ONE division per sample for the
// seven modes instead of seven.  The two roundings swap places (<= 1 ulp of an x <= 0.045: 6e-18 relative on exp(-x), a twentieth of
// that function's own half-ulp).
OW_DEV double damper_ramp_pos(double t, double ramp) { return ow_div(t, ramp); }
__device__ __noinline__ __attribute__((const)) double noise_fade_env(double t) { return 0.5 * (1.0 - cos(3.14159265358979323846 * t)); }  // hammer.rs:165
// tanh for the power amp's Newton loop, the speaker and the pickup's soft limit.  The device library's f64 tanh is a < 1 ulp routine built on
// extended-precision (double-double) exponentials: ~130 instructions, 100 of them adds, twice or three times per output sample
// and lane.  The reference calls libm tanh, itself good to 1-2 ulp; this one is tanh(x) = z / (z + 2), z = expm1(2|x|) from the
// same ln 2 reduction as exp_bounded and a degree-13 series: ~45 instructions, <= 2 ulp (measured on the device against an 80-bit
// reference over the whole domain, tests/test_gpu_division.py::test_tanh_fast_accuracy; glibc's own tanh measures 1.2 there).  OW_LIB_TANH restores the library call.
OW_DEV double tanh_fast(double x) {
#ifdef OW_LIB_TANH
    return tanh(x);
#else
    double ax = fabs(x);
    ax = ax > 20.0 ? 20.0 : ax;                       // tanh(20) rounds to 1.0; NaN stays NaN
    const double y = ax + ax;
    const double n = rint(y * __longlong_as_double(0x3ff71547652b82feLL));
    double r = __builtin_fma(__longlong_as_double((long long)0xbfe62e42fefa39efULL), n, y);
    r = __builtin_fma(__longlong_as_double((long long)0xbc7abc9e3b39803fULL), n, r);
    double q = 1.0 / 6227020800.0;                     // 1/13!
    q = __builtin_fma(q, r, 1.0 / 479001600.0);
    q = __builtin_fma(q, r, 1.0 / 39916800.0);
    q = __builtin_fma(q, r, 1.0 / 3628800.0);
    q = __builtin_fma(q, r, 1.0 / 362880.0);
    q = __builtin_fma(q, r, 1.0 / 40320.0);
    q = __builtin_fma(q, r, 1.0 / 5040.0);
    q = __builtin_fma(q, r, 1.0 / 720.0);
    q = __builtin_fma(q, r, 1.0 / 120.0);
    q = __builtin_fma(q, r, 1.0 / 24.0);
    q = __builtin_fma(q, r, 1.0 / 6.0);
    q = __builtin_fma(q, r, 0.5);
    const double em1 = __builtin_fma(r * r, q, r);     // expm1(r), |r| <= ln2 / 2
    const double s = __longlong_as_double((long long)(1023 + (int)n) << 52);   // 2^n, 0 <= n <= 58
    const double z = __builtin_fma(s, em1, s - 1.0);   // expm1(2|x|) = 2^n expm1(r) + (2^n - 1)
    // z / (z + 2) with the rounding error of the sum fed back into the quotient's residual (Fast2Sum is exact while z <= 2, which is
    // where it matters: for larger z the quotient is insensitive to it)
    const double d = z + 2.0;
    const double d_lo = z <= 2.0 ? z - (d - 2.0) : 0.0;
    double rc = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, rc, 1.0);
    rc = __builtin_fma(rc, e, rc);
    e = __builtin_fma(-d, rc, 1.0);
    rc = __builtin_fma(rc, e, rc);
    const double q0 = z * rc;
    double res = __builtin_fma(-d, q0, z);
    res = __builtin_fma(-d_lo, q0, res);
    const double t = __builtin_fma(res, rc, q0);        // d >= 2: no special cases (NaN propagates)
    return copysign(t, x);
#endif
}
// (one argument: with |y| passed in, the caller materialises it -- two instructions -- on every sample, ahead of the branch that is
// taken once in a blue moon)
// Round 5: tanh_fast and the constant-divisor form instead of the library's tanh behind a full division (180 -> 61 vector
// instructions per call).  "Once in a blue moon" holds for quiet play; the benchmark's chord at ff does not: for the first ~0.3 s after
// a whole-keyboard strike some lane of a wavefront is above 0.94 on 4-5 % of the samples, and the call was 9 % of the steady kernel's
// instructions in the driver's window (87 against 79 per voice-sample a second later, tools/pmc_restrike.sh).  tanh_fast is <= 2 ulp
// (as the library's and glibc's are 1-2): 0.04 * 2 ulp = 9e-18 on a value near 0.95.
__device__ __noinline__ __attribute__((const)) double pickup_saturate_hi(double y) {                                 // pickup.rs:76-79
    const double ay = fabs(y);
    const double range = 0.98 - 0.94;
    return copysign(0.94 + range * tanh_fast(OW_DIV_C(ay - 0.94, 0.98 - 0.94)), y);
}

// ------------------------------------------------------------------ per-sample voice state in registers
#define OW_LCOEF_ROWS 19
// fills the lane's column of the phase-only coefficient table (see VoiceRegs::step)
OW_DEV void lcoef_load(double* __restrict__ lcoef, const double* __restrict__ rec) {
    for (int i = 0; i < 5; ++i) lcoef[i * 64] = rec[(VF_NB0 + i) * 64];
    for (int i = 0; i < 7; ++i) { lcoef[(5 + i) * 64] = rec[(VF_DRATE + i) * 64]; lcoef[(12 + i) * 64] = rec[(VF_DMULT + i) * 64]; }
}

struct VoiceRegs {
    double s[7], c[7], env[7], drift[7], cos_inc[7], sin_inc[7], phase_inc[7], amp[7], decay[7];
    double ci[7], si[7];                // jitter-corrected rotation (reed.rs:281-283): a function of the drift alone, which moves every 16th sample
    double onset_inc, onset_exp, dramp, dcount, q, ds, gain;
    double namp, ns1, ns2;              // attack noise: amplitude and BPF state; the five BPF coefficients stay in the record / LDS
    double beta, jrev, jdiff, ndecay;   // per-voice rate constants (fixed at note-on; loaded once per kernel, never re-read in the sample loop)
    uint64_t sample, onset_n;
    uint32_t jitter_state, noise_rng, noise_rem, noise_fade, flags, midi;

    OW_DEV void load(const double* __restrict__ rec) {
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            s[i] = rec[(VF_S + i) * 64]; c[i] = rec[(VF_C + i) * 64]; env[i] = rec[(VF_ENV + i) * 64];
            drift[i] = rec[(VF_DRIFT + i) * 64]; cos_inc[i] = rec[(VF_COS_INC + i) * 64]; sin_inc[i] = rec[(VF_SIN_INC + i) * 64];
            phase_inc[i] = rec[(VF_PHASE_INC + i) * 64]; amp[i] = rec[(VF_AMP + i) * 64]; decay[i] = rec[(VF_DECAY + i) * 64];
        }
        onset_inc = rec[VF_ONSET_INC * 64]; onset_exp = rec[VF_ONSET_EXP * 64];
        dramp = rec[VF_DRAMP * 64]; dcount = rec[VF_DCOUNT * 64];
        q = rec[VF_Q * 64]; ds = rec[VF_DS * 64]; gain = rec[VF_GAIN * 64];
        beta = rec[VF_BETA * 64]; jrev = rec[VF_JREV * 64]; jdiff = rec[VF_JDIFF * 64]; ndecay = rec[VF_NDECAY * 64];
        namp = rec[VF_NAMP * 64];
        ns1 = rec[VF_NS1 * 64]; ns2 = rec[VF_NS2 * 64];
        sample = dbits(rec[VF_SAMPLE * 64]); onset_n = dbits(rec[VF_ONSET_N * 64]);
        const uint64_t r = dbits(rec[VF_RNG * 64]); jitter_state = (uint32_t)r; noise_rng = (uint32_t)(r >> 32);
        const uint64_t n = dbits(rec[VF_NCNT * 64]); noise_rem = (uint32_t)n; noise_fade = (uint32_t)(n >> 32);
        const uint64_t f = dbits(rec[VF_FLAGS * 64]); flags = (uint32_t)f; midi = (uint32_t)(f >> 32);
        update_rotation();
    }
    // ci / si of reed.rs:281-283.  The reference forms them on every sample from the same four numbers; they only change when the drift
    // does (every 16th sample, below), so they are formed there -- the same expressions on the same operands, the same bits.
    OW_DEV void update_rotation() {
#ifndef OW_STRICT_FP
#pragma clang fp contract(fast)
#endif
#pragma unroll
        for (int m = 0; m < 7; ++m) {
            const double delta_phase = drift[m] * phase_inc[m];
            ci[m] = cos_inc[m] - delta_phase * sin_inc[m];
            si[m] = sin_inc[m] + delta_phase * cos_inc[m];
        }
    }
    OW_DEV void store(double* __restrict__ rec) const {
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            rec[(VF_S + i) * 64] = s[i]; rec[(VF_C + i) * 64] = c[i]; rec[(VF_ENV + i) * 64] = env[i];
            rec[(VF_DRIFT + i) * 64] = drift[i];
        }
        rec[VF_DCOUNT * 64] = dcount; rec[VF_Q * 64] = q;
        rec[VF_NAMP * 64] = namp; rec[VF_NS1 * 64] = ns1; rec[VF_NS2 * 64] = ns2;
        rec[VF_SAMPLE * 64] = bitsd(sample);
        rec[VF_RNG * 64] = bitsd((uint64_t)jitter_state | ((uint64_t)noise_rng << 32));
        rec[VF_NCNT * 64] = bitsd((uint64_t)noise_rem | ((uint64_t)noise_fade << 32));
        rec[VF_FLAGS * 64] = bitsd((uint64_t)flags | ((uint64_t)midi << 32));
    }

    // One sample of Voice::render (voice.rs:162-179): reed (reed.rs:223-305) + attack noise
    // (hammer.rs:150-179) -> pickup (pickup.rs:130-149) -> x post_pickup_gain.
    // `rec` is only touched for the damper tables (rare path: released voices).
    //
    // STEADY = true is the wave-uniform fast path: no lane of the wave is inside a damper phase, an onset ramp
    // or an attack-noise burst for this chunk, so those three blocks (and their per-sample tests) are compiled out.
    // The arithmetic of the remaining blocks is identical.
    //
    // FMA contraction is enabled for this function (OW_STRICT_FP restores the reference's unfused a*b+c):
    // the kernel is VALU-issue bound and the rotation / pickup are mul-add chains.  Fused results differ from the
    // unfused reference by <= 1e-12 of peak over the parity renders (tests/test_gpu_parity.py voice-sum tap).
    template <bool STEADY, bool ONSET_FROM_TABLE = false>
    // lcoef: LDS copy of the lane's phase-only coefficients, lcoef[i * 64]: i = 0..4 attack-noise BPF b0, b1, b2, a1, a2
    // (VF_NB0..VF_NA2, first 15 ms of a note), i = 5..11 damper_rate, i = 12..18 damper_mult (VF_DRATE / VF_DMULT, while the key is
    // released).  In registers they would cost 38 VGPRs for the whole kernel; read from the voice record in HBM they cost a
    // dependent global load per mode per sample for as long as a released voice rings (measured: the played workload's bottleneck).
    // onset_tab / damp_tab: this sample's onset gain / the seven damper-ramp factors exp(-rate t / ramp), evaluated ahead for the
    // whole chunk by k_voice (lane-parallel over the samples), or nullptr to evaluate them here.
    OW_DEV double step(const double* __restrict__ lcoef, const double* __restrict__ onset_tab = nullptr,
                       const double* __restrict__ damp_tab = nullptr) {
#ifndef OW_STRICT_FP
#pragma clang fp contract(fast)
#endif
        double onset = 1.0;
        if (!STEADY) {
            if (flags & 1u) {  // damper_active
                dcount += 1.0;
                const double t = dcount;
                if (!(flags & 2u)) {
                    if (t > dramp) flags |= 2u;
                    else if (damp_tab) {
#pragma unroll
                        for (int m = 0; m < 7; ++m) env[m] *= damp_tab[m];
                    } else {
                        const double tr = damper_ramp_pos(t, dramp);
                        // damper_rate rises with the mode number (reed.rs:198-201) and t <= ramp: mode 6 decides for all seven
                        if (__builtin_expect(!(lcoef[11 * 64] <= 0.125), 0)) {
                            for (int m = 0; m < 7; ++m) env[m] *= exp_neg(lcoef[(5 + m) * 64] * tr);
                        } else {
#pragma unroll
                            for (int m = 0; m < 7; ++m) env[m] *= exp_neg_poly(lcoef[(5 + m) * 64] * tr);
                        }
                    }
                }
                if (flags & 2u) {
#pragma unroll
                    for (int m = 0; m < 7; ++m) env[m] *= lcoef[(12 + m) * 64];
                }
            }
            // (ONSET_FROM_TABLE: k_voice tabulates the gains of every lane inside its ramp ahead of the chunk -- no call in its sample loop)
            if (sample < onset_n) onset = (ONSET_FROM_TABLE || onset_tab) ? *onset_tab : onset_gain((double)sample, onset_inc, onset_exp);
        }
        const uint32_t lo = (uint32_t)sample;
        if ((lo & 15u) == 0u) {
            const double revert = jrev, diffusion = jdiff;
#pragma unroll
            for (int m = 0; m < 7; ++m) {
                jitter_state = lcg(jitter_state);
                const double u = OW_DIV_C((double)(jitter_state >> 1), OW_JITTER_DIV);
                const double noise = (u * 2.0 - 1.0) * 1.7320508080;
                drift[m] = revert * drift[m] + diffusion * noise;
            }
            update_rotation();
        }
        double sum = 0.0;
#pragma unroll
        for (int m = 0; m < 7; ++m) {
            if (STEADY) sum += amp[m] * s[m] * env[m];          // onset == 1.0: x * 1.0 == x exactly
            else sum += amp[m] * s[m] * onset * env[m];
            const double s_new = s[m] * ci[m] + c[m] * si[m];
            const double c_new = c[m] * ci[m] - s[m] * si[m];
            s[m] = s_new;
            c[m] = c_new;
            env[m] *= decay[m];
        }
        // (a real branch: written as a plain `if` the compiler predicates the seven square roots and divisions -- ~130 instructions -- into
        // every sample of the general loop; one lane in 1 024 is due)
        const bool renorm_due = (lo & 1023u) == 0u && sample > 0ull;
        if (__builtin_amdgcn_ballot_w64(renorm_due) != 0ull) {
            asm volatile("" ::: "memory");       // not speculated
            if (renorm_due) {
#pragma unroll
                for (int m = 0; m < 7; ++m) {
                    const double r_sq = s[m] * s[m] + c[m] * c[m];
                    const double r_inv = 1.0 / sqrt(r_sq);
                    s[m] *= r_inv;
                    c[m] *= r_inv;
                }
            }
        }
        sample += 1ull;
        double x = 0.0 + sum;
        if (!STEADY && noise_rem > 0u) {
            double e = 1.0;
            if (noise_fade > 0u) {
                const uint32_t pos = 16u - noise_fade;
                noise_fade -= 1u;
                e = noise_fade_env((double)pos / 16.0);
            }
            noise_rng = lcg(noise_rng);
            const double nz = OW_DIV_C((double)(int32_t)noise_rng, 2147483647.0);     // (= the IEEE quotient for all 2^32 draws, tests/test_gpu_division.py)
            const double y = lcoef[0] * nz + ns1;
            ns1 = lcoef[64] * nz - lcoef[192] * y + ns2;
            ns2 = lcoef[128] * nz - lcoef[256] * y;
            x += namp * e * y;
            namp *= ndecay;
            noise_rem -= 1u;
        }
        // pickup
        double y = x * ds;
        const double ay = fabs(y);
        if (!(ay < 0.94)) y = pickup_saturate_hi(y);
        const double omy = 1.0 - y;
        const double alpha = beta * omy;
        const double q_next = ow_div(q * (1.0 - alpha) + 2.0 * beta, 1.0 + alpha);
        q = q_next;
        return ((q_next * omy - 1.0) * 1.8375) * gain;
    }
    // true while the lane needs the general (non-steady) step: damper running, onset ramp or attack noise not finished
    OW_DEV bool in_transient() const { return (flags & 1u) || sample < onset_n || noise_rem > 0u; }
    // a non-finite output sample always leaves a non-finite trace in the recurrences (q, s, c, env never recover)
    OW_DEV bool state_finite() const {
        bool ok = isfinite(q);
#pragma unroll
        for (int m = 0; m < 7; ++m) ok = ok && isfinite(s[m]) && isfinite(c[m]) && isfinite(env[m]);
        return ok;
    }

    // Voice::is_silent (voice.rs:183-188, reed.rs:309-314); threshold 10^(-80/20)
    OW_DEV bool is_silent(const double* __restrict__ rec) const {
        if ((flags & 1u) && (dcount / rec[VF_VSR * 64]) > 10.0) return true;
        bool all = true;
#pragma unroll
        for (int m = 0; m < 7; ++m) all = all && (fabs(amp[m] * env[m]) <= 1e-4);
        return all;
    }
};

}  // namespace owdev
