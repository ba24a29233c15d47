// ORACLE (test infrastructure, CPU, f64) -- NOT part of the shipped product path.
//
// Restates the opt-in melange 12-node DK preamp of the reference (cargo feature `melange-preamp`; citations into
// /root/reference/crates/openwurli-dsp/src/):
//   gen_preamp.rs:1568-1588,1748-1876  DC_OP / CircuitState default / reset
//   gen_preamp.rs:1930-2062            set_sample_rate, set_runtime_R_r_ldr (lazy dirty flag), rebuild_matrices
//   gen_preamp.rs:2117-2219            invert_n (LU, partial pivoting, identity fallback)
//   gen_preamp.rs:2416-2428            diode model; :3148-3157 forward-active 1-D BJTs
//   gen_preamp.rs:3041-3095            build_rhs (emitted sparsity, grouped adds)
//   gen_preamp.rs:3122-3357            solve_nonlinear (MAX_ITER 265)
//   gen_preamp.rs:3399-3663            process_sample (BE fallback + 64-sample cooldown, voltage-damp net with the one
//                                      explicit mul_add of the whole path, NaN reset)
//   dk_preamp/melange_adapter.rs:12-94 settled-state cache (176 400 samples), main/shadow, reset
// Quirks reproduced: only the trapezoidal S/A_neg/K/S_NI are rebuilt on a rate or R change -- the BE matrices stay at
// their 48 kHz / 100 kOhm codegen values for the life of the state; the thermal-noise path is not restated (off by default,
// SURVEY.md 8f row 4).
#pragma once
#include "ow_tremolo.hpp"   // fast_exp, pnjlim (identical text in both generated files)
#include <cmath>

#include <chrono>
namespace owo {

constexpr int PN = 12, PM = 3;

struct MelState {
    double v_prev[PN], i_nl_prev[PM], i_nl_prev_prev[PM], input_prev;
    uint32_t last_nr_iterations, be_cooldown;
    uint64_t diag_nr_max_iter_count, diag_be_fallback_count, diag_nan_reset_count, diag_voltage_damp_count, diag_singular_matrix_count;
    double s[PN][PN], a_neg[PN][PN], k[PM][PM], s_ni[PN][PM];
    double s_be[PN][PN], k_be[PM][PM], s_ni_be[PN][PM], a_neg_be[PN][PN];
    double pot_0_resistance, current_sample_rate;
    bool matrices_dirty;
    // ---- authentic circuit noise, phase 1 = thermal (gen_preamp.rs:1434-1561, state :1708-1745) ----
    struct Xoshiro256pp {   // :1465-1490
        uint64_t s[4];
        static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
        uint64_t next_u64() {
            const uint64_t result = rotl(s[0] + s[3], 23) + s[0];
            const uint64_t t = s[1] << 17;
            s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
            s[2] ^= t;
            s[3] = rotl(s[3], 45);
            return result;
        }
        double next_f64() { return (double)(next_u64() >> 11) * (1.0 / (double)(1ull << 53)); }
    };
    static constexpr int NT = 11;                       // NOISE_THERMAL_N
    Xoshiro256pp noise_rng[NT];
    bool noise_cache_valid[NT];
    double noise_cache[NT];
    bool noise_enabled;
    double noise_gain, thermal_gain, temperature_k;
    uint64_t noise_master_seed;
    double noise_thermal_scale, noise_fs;
    double noise_sqrt_inv_r[NT], noise_w_prev[NT], noise_last_i_n[NT];

    static uint64_t splitmix64(uint64_t& st) {          // :1493-1499
        st += 0x9E3779B97F4A7C15ull;
        uint64_t z = st;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    // master == 0 draws entropy from the system clock in the reference (:1512-1521); the oracle does the same once per process
    static uint64_t entropy_seed() {
        static const uint64_t e = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(
                                      std::chrono::system_clock::now().time_since_epoch()).count();
        return e ? e : 0x0123456789ABCDEFull;
    }
    void seed_noise_rngs(uint64_t master) {             // seed_noise_rngs_salted(master, 0), :1511-1536
        uint64_t sm = master == 0 ? entropy_seed() : master;
        (void)splitmix64(sm);
        for (int k = 0; k < NT; ++k) {
            for (int q = 0; q < 4; ++q) noise_rng[k].s[q] = splitmix64(sm);
            if (!(noise_rng[k].s[0] | noise_rng[k].s[1] | noise_rng[k].s[2] | noise_rng[k].s[3])) noise_rng[k].s[0] = 1;
        }
    }
    double gaussian(int k) {                            // Marsaglia polar, second value cached (:1547-1561)
        if (noise_cache_valid[k]) { noise_cache_valid[k] = false; return noise_cache[k]; }
        for (;;) {
            const double u = 2.0 * noise_rng[k].next_f64() - 1.0;
            const double v = 2.0 * noise_rng[k].next_f64() - 1.0;
            const double ss = u * u + v * v;
            if (ss > 0.0 && ss < 1.0) {
                const double factor = std::sqrt(-2.0 * std::log(ss) / ss);
                noise_cache[k] = v * factor; noise_cache_valid[k] = true;
                return u * factor;
            }
        }
    }
    void noise_clear_lag() { for (int k = 0; k < NT; ++k) { noise_w_prev[k] = 0.0; noise_last_i_n[k] = 0.0; } }
    void set_seed(uint64_t master) {                    // :2094-2100
        noise_master_seed = master;
        seed_noise_rngs(master);
        for (int k = 0; k < NT; ++k) noise_cache_valid[k] = false;
        noise_clear_lag();
    }
    static void noise_stamp(double rhs[PN], int k, double i_n) {   // :3443-3451
        const int ni = (int)PRE_NOISE_THERMAL_NODE_I[k], nj = (int)PRE_NOISE_THERMAL_NODE_J[k];
        if (ni > 0) rhs[ni - 1] += i_n;
        if (nj > 0) rhs[nj - 1] -= i_n;
    }

    void init_default() {  // gen_preamp.rs:1748-1821
        for (int i = 0; i < PN; ++i) v_prev[i] = PRE_DC_OP[i];
        for (int i = 0; i < PM; ++i) { i_nl_prev[i] = PRE_DC_NL_I[i]; i_nl_prev_prev[i] = PRE_DC_NL_I[i]; }
        input_prev = 0.0;
        last_nr_iterations = 0; be_cooldown = 0;
        diag_nr_max_iter_count = diag_be_fallback_count = diag_nan_reset_count = diag_voltage_damp_count = diag_singular_matrix_count = 0;
        std::memcpy(s, PRE_S_DEFAULT, sizeof s); std::memcpy(a_neg, PRE_A_NEG_DEFAULT, sizeof a_neg);
        std::memcpy(k, PRE_K_DEFAULT, sizeof k); std::memcpy(s_ni, PRE_S_NI_DEFAULT, sizeof s_ni);
        std::memcpy(s_be, PRE_S_BE_DEFAULT, sizeof s_be); std::memcpy(k_be, PRE_K_BE_DEFAULT, sizeof k_be);
        std::memcpy(s_ni_be, PRE_S_NI_BE_DEFAULT, sizeof s_ni_be); std::memcpy(a_neg_be, PRE_A_NEG_BE_DEFAULT, sizeof a_neg_be);
        pot_0_resistance = 9.99999999999999854e4;
        current_sample_rate = PRE_SAMPLE_RATE;
        matrices_dirty = false;
        noise_fs = PRE_SAMPLE_RATE * 1.0;                                   // SAMPLE_RATE * OVERSAMPLING_FACTOR
        temperature_k = 290.0;
        noise_thermal_scale = std::sqrt(8.0 * 1.380649e-23 * 290.0 * noise_fs);
        noise_master_seed = 0;
        seed_noise_rngs(0);
        for (int k = 0; k < NT; ++k) { noise_cache_valid[k] = false; noise_cache[k] = 0.0; noise_sqrt_inv_r[k] = PRE_NOISE_THERMAL_SQRT_INV_R_DEFAULT[k]; }
        noise_enabled = false; noise_gain = 1.0; thermal_gain = 1.0;
        noise_clear_lag();
    }

    void set_sample_rate(double sr) {  // :1930-1961
        if (!(sr > 0.0 && std::isfinite(sr))) return;
        current_sample_rate = sr;
        noise_fs = sr * 1.0;                                                // :1935-1936
        noise_thermal_scale = std::sqrt(8.0 * 1.380649e-23 * temperature_k * noise_fs);
        if (std::fabs(sr - PRE_SAMPLE_RATE) < 0.5) {
            std::memcpy(s, PRE_S_DEFAULT, sizeof s); std::memcpy(a_neg, PRE_A_NEG_DEFAULT, sizeof a_neg);
            std::memcpy(k, PRE_K_DEFAULT, sizeof k); std::memcpy(s_ni, PRE_S_NI_DEFAULT, sizeof s_ni);
            std::memcpy(s_be, PRE_S_BE_DEFAULT, sizeof s_be); std::memcpy(k_be, PRE_K_BE_DEFAULT, sizeof k_be);
            std::memcpy(s_ni_be, PRE_S_NI_BE_DEFAULT, sizeof s_ni_be); std::memcpy(a_neg_be, PRE_A_NEG_BE_DEFAULT, sizeof a_neg_be);
            return;
        }
        rebuild_matrices();
    }
    void set_runtime_r_ldr(double r_in) {  // :1973-1984
        if (!std::isfinite(r_in)) return;
        const double r = rclamp(r_in, 1000.0, 1000000.0);
        if (std::fabs(r - pot_0_resistance) < 1e-12) return;
        pot_0_resistance = r;
        matrices_dirty = true;
        noise_sqrt_inv_r[10] = std::sqrt(1.0 / r);                          // :1983
    }

    static bool invert_n(const double a[PN][PN], double result[PN][PN]) {  // :2117-2219; returns singular flag
        double lu[PN][PN];
        int perm[PN];
        for (int i = 0; i < PN; ++i) { perm[i] = i; for (int j = 0; j < PN; ++j) lu[i][j] = a[i][j]; }
        auto identity = [&]() { for (int i = 0; i < PN; ++i) for (int j = 0; j < PN; ++j) result[i][j] = (i == j) ? 1.0 : 0.0; };
        for (int kk = 0; kk < PN; ++kk) {
            int max_row = kk;
            double max_val = std::fabs(lu[kk][kk]);
            for (int i = kk + 1; i < PN; ++i) {
                const double v = std::fabs(lu[i][kk]);
                if (v > max_val) { max_val = v; max_row = i; }
            }
            if (max_val < 1e-30) { identity(); return true; }
            if (max_row != kk) {
                for (int j = 0; j < PN; ++j) std::swap(lu[kk][j], lu[max_row][j]);
                std::swap(perm[kk], perm[max_row]);
            }
            const double pivot = lu[kk][kk];
            for (int i = kk + 1; i < PN; ++i) {
                const double m = lu[i][kk] / pivot;
                lu[i][kk] = m;
                for (int j = kk + 1; j < PN; ++j) lu[i][j] -= m * lu[kk][j];
            }
        }
        for (int i = 0; i < PN; ++i) for (int j = 0; j < PN; ++j) result[i][j] = 0.0;
        for (int col = 0; col < PN; ++col) {
            double b[PN];
            for (int i = 0; i < PN; ++i) b[i] = 0.0;
            int start = PN;
            for (int i = 0; i < PN; ++i)
                if (perm[i] == col) { b[i] = 1.0; start = i; break; }
            for (int i = start + 1; i < PN; ++i) {
                double sum = b[i];
                for (int j = start; j < i; ++j) sum -= lu[i][j] * b[j];
                b[i] = sum;
            }
            for (int i = PN - 1; i >= 0; --i) {
                double sum = b[i];
                for (int j = i + 1; j < PN; ++j) sum -= lu[i][j] * b[j];
                const double pivot = lu[i][i];
                if (std::fabs(pivot) < 1e-30) { identity(); return true; }
                b[i] = sum / pivot;
            }
            for (int i = 0; i < PN; ++i) result[i][col] = b[i];
        }
        return false;
    }

    void rebuild_matrices() {  // :1990-2062 (BE matrices intentionally untouched)
        const double alpha = 2.0 * (current_sample_rate * 1.0);
        double g_eff[PN][PN];
        std::memcpy(g_eff, PRE_G, sizeof g_eff);
        g_eff[6][6] += 1.0 / pot_0_resistance - PRE_POT_0_G_NOM;
        double a[PN][PN], an[PN][PN];
        for (int i = 0; i < PN; ++i)
            for (int j = 0; j < PN; ++j) {
                a[i][j] = g_eff[i][j] + alpha * PRE_C[i][j];
                an[i][j] = alpha * PRE_C[i][j] - g_eff[i][j];
            }
        for (int j = 0; j < PN; ++j) an[11][j] = 0.0;
        double sn[PN][PN];
        if (invert_n(a, sn)) diag_singular_matrix_count += 1;
        double sni[PN][PM];
        for (int i = 0; i < PN; ++i)
            for (int j = 0; j < PM; ++j) {
                double sum = 0.0;
                for (int kk = 0; kk < PN; ++kk) sum += sn[i][kk] * PRE_N_I[j][kk];
                sni[i][j] = sum;
            }
        double kn[PM][PM];
        for (int i = 0; i < PM; ++i)
            for (int j = 0; j < PM; ++j) {
                double sum = 0.0;
                for (int n = 0; n < PN; ++n) sum += PRE_N_V[i][n] * sni[n][j];
                kn[i][j] = sum;
            }
        std::memcpy(s, sn, sizeof s); std::memcpy(a_neg, an, sizeof a_neg); std::memcpy(k, kn, sizeof k); std::memcpy(s_ni, sni, sizeof s_ni);
    }

    // :3122-3357; kk/ = active kernel (K or K_be)
    void solve_nonlinear(const double p[PM], const double kk[PM][PM], double i_nl[PM]) {
        for (int i = 0; i < PM; ++i) i_nl[i] = 2.0 * i_nl_prev[i] - i_nl_prev_prev[i];
        for (int iter = 0; iter < 265; ++iter) {
            const double v_d0 = p[0] + kk[0][0] * i_nl[0] + kk[0][1] * i_nl[1] + kk[0][2] * i_nl[2];
            const double v_d1 = p[1] + kk[1][0] * i_nl[0] + kk[1][1] * i_nl[1] + kk[1][2] * i_nl[2];
            const double v_d2 = p[2] + kk[2][0] * i_nl[0] + kk[2][1] * i_nl[1] + kk[2][2] * i_nl[2];
            const double is0 = PRE_DEVICE_0_IS, nvt0 = PRE_DEVICE_0_N_VT;
            const double vcl = rclamp(v_d0, -40.0 * nvt0, 40.0 * nvt0);
            const double i_dev0 = is0 * (fast_exp(vcl / nvt0) - 1.0);
            const double jdev00 = (is0 / nvt0) * fast_exp(vcl / nvt0);
            const double vbe_1 = v_d1 * 1.0;
            const double exp_be_1 = fast_exp(vbe_1 / (PRE_DEVICE_1_NF * PRE_DEVICE_1_VT));
            const double i_dev1 = PRE_DEVICE_1_IS * (exp_be_1 - 1.0) * 1.0;
            const double jdev11 = PRE_DEVICE_1_IS / (PRE_DEVICE_1_NF * PRE_DEVICE_1_VT) * exp_be_1;
            const double vbe_2 = v_d2 * 1.0;
            const double exp_be_2 = fast_exp(vbe_2 / (PRE_DEVICE_2_NF * PRE_DEVICE_2_VT));
            const double i_dev2 = PRE_DEVICE_2_IS * (exp_be_2 - 1.0) * 1.0;
            const double jdev22 = PRE_DEVICE_2_IS / (PRE_DEVICE_2_NF * PRE_DEVICE_2_VT) * exp_be_2;
            const double f0 = i_nl[0] - i_dev0, f1 = i_nl[1] - i_dev1, f2 = i_nl[2] - i_dev2;
            double a[3][3] = {
                {1.0 - jdev00 * kk[0][0], 0.0 - jdev00 * kk[0][1], 0.0 - jdev00 * kk[0][2]},
                {0.0 - jdev11 * kk[1][0], 1.0 - jdev11 * kk[1][1], 0.0 - jdev11 * kk[1][2]},
                {0.0 - jdev22 * kk[2][0], 0.0 - jdev22 * kk[2][1], 1.0 - jdev22 * kk[2][2]},
            };
            double b[3] = {f0, f1, f2};
            bool singular = false;
            for (int col = 0; col < 3; ++col) {
                int max_row = col;
                double max_val = std::fabs(a[col][col]);
                for (int row = col + 1; row < 3; ++row)
                    if (std::fabs(a[row][col]) > max_val) { max_val = std::fabs(a[row][col]); max_row = row; }
                if (max_val < 1e-15) { singular = true; break; }
                if (max_row != col) {
                    for (int j = 0; j < 3; ++j) std::swap(a[col][j], a[max_row][j]);
                    std::swap(b[col], b[max_row]);
                }
                const double pivot = a[col][col];
                for (int row = col + 1; row < 3; ++row) {
                    const double factor = a[row][col] / pivot;
                    for (int j = col + 1; j < 3; ++j) a[row][j] -= factor * a[col][j];
                    b[row] -= factor * b[col];
                }
            }
            if (!singular) {
                for (int i = 2; i >= 0; --i) {
                    double sum = b[i];
                    for (int j = i + 1; j < 3; ++j) sum -= a[i][j] * b[j];
                    if (std::fabs(a[i][i]) < 1e-15) { singular = true; break; }
                    b[i] = sum / a[i][i];
                }
            }
            if (!singular) {
                const double delta0 = b[0], delta1 = b[1], delta2 = b[2];
                const double dv[3] = {-(kk[0][0] * delta0 + kk[0][1] * delta1 + kk[0][2] * delta2),
                                      -(kk[1][0] * delta0 + kk[1][1] * delta1 + kk[1][2] * delta2),
                                      -(kk[2][0] * delta0 + kk[2][1] * delta1 + kk[2][2] * delta2)};
                const double vd[3] = {v_d0, v_d1, v_d2};
                const double vts[3] = {PRE_DEVICE_0_N_VT, PRE_DEVICE_1_VT, PRE_DEVICE_2_VT};
                const double vcr[3] = {PRE_DEVICE_0_VCRIT, PRE_DEVICE_1_VCRIT, PRE_DEVICE_2_VCRIT};
                double alpha[3] = {1.0, 1.0, 1.0};
                bool any_limited = false;
                for (int q = 0; q < 3; ++q) {
                    if (std::fabs(dv[q]) > 1e-4) {
                        const double v_lim = pnjlim(vd[q] + dv[q], vd[q], vts[q], vcr[q]);
                        const double ratio = std::fmax((v_lim - vd[q]) / dv[q], 0.01);
                        if (ratio < alpha[q]) { alpha[q] = ratio; if (ratio < 1.0) any_limited = true; }
                    }
                }
                double alpha_scalar = std::fmin(alpha[0], std::fmin(alpha[1], alpha[2]));
                if (alpha_scalar < 1.0) any_limited = true;
                const double max_di = std::fmax(std::fmax(std::fabs(delta0), std::fabs(delta1)), std::fabs(delta2));
                if (max_di * alpha_scalar > 0.1) alpha_scalar = std::fmin(std::fmax(0.1 / max_di, 0.01), alpha_scalar);
                i_nl[0] -= alpha_scalar * delta0;
                i_nl[1] -= alpha_scalar * delta1;
                i_nl[2] -= alpha_scalar * delta2;
                bool conv = true;
                if (!any_limited) {
                    for (int q = 0; q < 3; ++q) {
                        const double step = dv[q] * alpha_scalar;
                        const double v_new = vd[q] + step;
                        const double thr = 1e-3 * std::fmax(std::fabs(vd[q]), std::fabs(v_new)) + 1e-6;
                        if (std::fabs(step) > thr) conv = false;
                    }
                }
                const double idev[3] = {i_dev0, i_dev1, i_dev2}, ff[3] = {f0, f1, f2};
                for (int q = 0; q < 3; ++q) {
                    const double i_thr = 1e-3 * std::fmax(std::fmax(std::fabs(i_nl[q]), std::fabs(idev[q])), 1e-9) + 1e-12;
                    if (std::fabs(ff[q]) > i_thr) conv = false;
                }
                if (conv) { last_nr_iterations = (uint32_t)iter; return; }
            } else {
                const double ff[3] = {f0, f1, f2};
                for (int q = 0; q < 3; ++q) {
                    const double cl = std::fmax(std::fabs(i_nl[q]) * 0.1, 0.01);
                    i_nl[q] -= rclamp(ff[q] * 0.5, -cl, cl);
                }
            }
        }
        last_nr_iterations = 265;
        for (int q = 0; q < 3; ++q) if (!std::isfinite(i_nl[q])) i_nl[q] = i_nl_prev[q];
    }

    double process_sample(double input_in) {  // :3399-3663
        const double input = std::isfinite(input_in) ? rclamp(input_in, -100.0, 100.0) : 0.0;
        if (matrices_dirty) { rebuild_matrices(); matrices_dirty = false; }
        for (int i = 0; i < PN; ++i) v_prev[i] = v_prev[i] + 1e-25 - 1e-25;
        for (int i = 0; i < PM; ++i) i_nl_prev[i] = i_nl_prev[i] + 1e-25 - 1e-25;
        const bool force_be = be_cooldown > 0;
        if (be_cooldown > 0) be_cooldown -= 1;

        double rhs[PN];
        for (int i = 0; i < PN; ++i) rhs[i] = PRE_RHS_CONST[i];
        const double (*an)[PN] = a_neg;
        const double* v = v_prev;
        rhs[0] += an[0][0] * v[0] + an[0][1] * v[1];
        rhs[1] += an[1][0] * v[0] + an[1][1] * v[1] + an[1][2] * v[2];
        rhs[2] += an[2][1] * v[1] + an[2][2] * v[2] + an[2][3] * v[3] + an[2][4] * v[4] + an[2][5] * v[5];
        rhs[3] += an[3][2] * v[2] + an[3][3] * v[3] + an[3][4] * v[4] + an[3][7] * v[7] + an[3][11] * v[11];
        rhs[4] += an[4][2] * v[2] + an[4][3] * v[3] + an[4][4] * v[4] + an[4][7] * v[7] + an[4][8] * v[8];
        rhs[5] += an[5][2] * v[2] + an[5][5] * v[5] + an[5][6] * v[6];
        rhs[6] += an[6][5] * v[5] + an[6][6] * v[6] + an[6][10] * v[10];
        rhs[7] += an[7][3] * v[3] + an[7][4] * v[4] + an[7][7] * v[7] + an[7][10] * v[10];
        rhs[8] += an[8][4] * v[4] + an[8][8] * v[8] + an[8][9] * v[9];
        rhs[9] += an[9][8] * v[8] + an[9][9] * v[9];
        rhs[10] += an[10][6] * v[6] + an[10][7] * v[7] + an[10][10] * v[10];
        rhs[2] += PRE_N_I[0][2] * i_nl_prev[0];
        rhs[2] += PRE_N_I[1][2] * i_nl_prev[1];
        rhs[4] += PRE_N_I[1][4] * i_nl_prev[1];
        rhs[4] += PRE_N_I[2][4] * i_nl_prev[2];
        rhs[5] += PRE_N_I[1][5] * i_nl_prev[1];
        rhs[7] += PRE_N_I[2][7] * i_nl_prev[2];
        rhs[8] += PRE_N_I[2][8] * i_nl_prev[2];
        rhs[0] += (input + input_prev) / PRE_INPUT_RESISTANCE;

        if (noise_enabled) {   // two-draw thermal stamp w_new + w_prev (:3433-3461)
            const double scale_half = noise_thermal_scale * noise_gain * thermal_gain * 0.5;
            if (scale_half != 0.0) {
                for (int k = 0; k < NT; ++k) {
                    const double g = gaussian(k);
                    const double w_new = scale_half * noise_sqrt_inv_r[k] * g;
                    const double i_n = w_new + noise_w_prev[k];
                    noise_w_prev[k] = w_new;
                    noise_last_i_n[k] = i_n;
                    noise_stamp(rhs, k, i_n);
                }
            } else {
                for (int k = 0; k < NT; ++k) noise_last_i_n[k] = 0.0;
            }
        } else {
            for (int k = 0; k < NT; ++k) noise_last_i_n[k] = 0.0;
        }

        double v_pred[PN];
        for (int i = 0; i < PN; ++i) {
            double sum = 0.0;
            for (int j = 0; j < PN; ++j) sum += s[i][j] * rhs[j];
            v_pred[i] = sum;
        }
        const double p[PM] = {-v_pred[2], v_pred[2] - v_pred[5], v_pred[4] - v_pred[8]};
        double i_nl[PM];
        solve_nonlinear(p, k, i_nl);
        double vn[PN];
        for (int i = 0; i < PN; ++i) {
            vn[i] = v_pred[i];
            for (int j = 0; j < PM; ++j) vn[i] += s_ni[i][j] * i_nl[j];
        }
        const bool nr_failed = last_nr_iterations >= 265u;
        bool ringing = false;
        for (int i = 0; i < 11; ++i) if (std::fabs(vn[i]) > 55.0) ringing = true;
        if (nr_failed || ringing || force_be) {
            if (nr_failed) diag_nr_max_iter_count += 1;
            if (ringing || nr_failed) be_cooldown = 64;
            diag_be_fallback_count += 1;
            double rhs_be[PN];
            for (int i = 0; i < PN; ++i) {
                double sum = PRE_RHS_CONST_BE[i];
                for (int j = 0; j < PN; ++j) sum += a_neg_be[i][j] * v_prev[j];
                for (int j = 0; j < PM; ++j) sum += PRE_N_I[j][i] * i_nl_prev[j];
                rhs_be[i] = sum;
            }
            rhs_be[0] += input / PRE_INPUT_RESISTANCE;
            if (noise_enabled)                                              // BE replay of the trap stamp (:3522-3535)
                for (int k = 0; k < NT; ++k) noise_stamp(rhs_be, k, noise_last_i_n[k]);
            double v_pred_be[PN];
            for (int i = 0; i < PN; ++i) {
                double sum = 0.0;
                for (int j = 0; j < PN; ++j) sum += s_be[i][j] * rhs_be[j];
                v_pred_be[i] = sum;
            }
            double p_be[PM];
            for (int i = 0; i < PM; ++i) {
                double sum = 0.0;
                for (int j = 0; j < PN; ++j) sum += PRE_N_V[i][j] * v_pred_be[j];
                p_be[i] = sum;
            }
            solve_nonlinear(p_be, k_be, i_nl);
            for (int i = 0; i < PN; ++i) {
                vn[i] = v_pred_be[i];
                for (int j = 0; j < PM; ++j) vn[i] += s_ni_be[i][j] * i_nl[j];
            }
        }
        {   // voltage-damp safety net (:3576-3613)
            double max_delta = 0.0;
            for (int i = 0; i < 11; ++i) { const double d = std::fabs(vn[i] - v_prev[i]); if (d > max_delta) max_delta = d; }
            double max_dc = 0.0;
            for (int i = 0; i < 11; ++i) { const double a = std::fabs(PRE_DC_OP[i]); if (a > max_dc) max_dc = a; }
            const double damp_thresh = std::fma(max_dc, 0.05, 2.0);      // the path's one explicit mul_add (:3592)
            if (max_delta > damp_thresh) {
                diag_voltage_damp_count += 1;
                const double damp = std::fmax(damp_thresh / max_delta, 0.01);
                for (int i = 0; i < PN; ++i) vn[i] = v_prev[i] + damp * (vn[i] - v_prev[i]);
                for (int i = 0; i < PM; ++i) i_nl[i] = i_nl_prev[i] + damp * (i_nl[i] - i_nl_prev[i]);
            }
        }
        bool finite = true;
        for (int i = 0; i < PN; ++i) finite = finite && std::isfinite(vn[i]);
        if (!finite) {
            for (int i = 0; i < PN; ++i) v_prev[i] = PRE_DC_OP[i];
            for (int i = 0; i < PM; ++i) { i_nl_prev[i] = PRE_DC_NL_I[i]; i_nl_prev_prev[i] = PRE_DC_NL_I[i]; }
            input_prev = 0.0;
            pot_0_resistance = 9.99999999999999854e4;
            be_cooldown = 0;
            noise_clear_lag();                                              // :3625-3627 (RNG state preserved)
            diag_nan_reset_count += 1;
            return rclamp(PRE_DC_OP[10] * 1.0, -10.0, 10.0);
        }
        for (int i = 0; i < PN; ++i) v_prev[i] = vn[i];
        for (int i = 0; i < PM; ++i) { i_nl_prev_prev[i] = i_nl_prev[i]; i_nl_prev[i] = i_nl[i]; }
        input_prev = input;
        if (last_nr_iterations >= 265u) diag_nr_max_iter_count += 1;
        const double raw = std::isfinite(vn[10]) ? vn[10] : 0.0;
        return raw * 1.0;
    }
};

// dk_preamp/melange_adapter.rs:12-94
struct MelangePreamp {
    // Newton sweeps spent (statistics for the GPU mapping, read by owo_engine_melange_stats): samples, main, shadow, max of the two
    unsigned long long stat_samples = 0, stat_main = 0, stat_shadow = 0, stat_max = 0;
    MelState main, shadow;
    double sample_rate = 0;
    bool noise_enabled = false;          // melange_adapter.rs:35-36,45-46
    double thermal_gain = 1.0;
    bool have_seed = false;              // extension mirrored by the HIP library: an explicit per-engine seed (gen_preamp::set_seed)
    uint64_t seed = 0;
    void set_noise_enabled(bool on) { noise_enabled = on; main.noise_enabled = on; }             // :54-57 (main only)
    void set_thermal_gain(double g) { thermal_gain = g; main.thermal_gain = g; }                 // :65-68
    void set_noise_seed(uint64_t s) { have_seed = true; seed = s; main.set_seed(s); }
    void reapply_noise() {                                                                       // :91-92
        main.noise_enabled = noise_enabled; main.thermal_gain = thermal_gain;
        if (have_seed) main.set_seed(seed);
    }
    static const MelState& settled() {
        static MelState st;
        static bool done = false;
        if (!done) {
            st.init_default();
            for (int i = 0; i < 176400; ++i) st.process_sample(0.0);
            done = true;
        }
        return st;
    }
    static MelState init_state(double sr) {
        MelState s = settled();
        if (std::fabs(sr - PRE_SAMPLE_RATE) > 0.5) s.set_sample_rate(sr);
        return s;
    }
    // DkPreamp::new (:40-48): noise off, gain 1.0 -- WurliEngine::set_sample_rate builds a new preamp and so drops both settings
    void init(double sr) { sample_rate = sr; noise_enabled = false; thermal_gain = 1.0; main = init_state(sr); shadow = init_state(sr); reapply_noise(); }
    double process_sample(double input) {
        const double m = main.process_sample(input);
        const double pump = shadow.process_sample(0.0);
        stat_samples += 1; stat_main += main.last_nr_iterations + 1u; stat_shadow += shadow.last_nr_iterations + 1u;
        stat_max += (main.last_nr_iterations > shadow.last_nr_iterations ? main.last_nr_iterations : shadow.last_nr_iterations) + 1u;
        const double result = m - pump;
        if (!std::isfinite(result)) { reset(); return 0.0; }
        return result;
    }
    void set_ldr_resistance(double r) {
#ifdef OW_ORACLE_EXP_PERTURB       // sensitivity variant: R_ldr off by one ulp (a libm whose pow/exp in the LDR law differs in the last bit)
        r = r * (1.0 + 2.2e-16);
#endif
        main.set_runtime_r_ldr(r); shadow.set_runtime_r_ldr(r);
    }
    void reset() { main = init_state(sample_rate); shadow = init_state(sample_rate); reapply_noise(); }
};

}  // namespace owo
