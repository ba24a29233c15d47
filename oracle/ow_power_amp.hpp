// ORACLE (test infrastructure, CPU, f64) -- NOT part of the shipped product path.
//
// Restates the melange-generated 7-BJT Class-AB power amp of the reference and its adapter (the amp a `--no-default-features`
// build of openwurli-dsp uses; citations into /root/reference/crates/openwurli-dsp/src/):
//   gen_power_amp.rs:29-74            N = 20 unknowns (18 nodes + 2 rail source rows), M = 16 ports (8 BJTs x {Ic, Ib}), MAX_ITER 70
//   gen_power_amp.rs:7464-7487        fast_exp (same text as gen_tremolo.rs; shared restatement)
//   gen_power_amp.rs:7527-7545        pnjlim (libm ln)
//   gen_power_amp.rs:7870-8017        bjt_evaluate: Gummel-Poon transport current (Early effect q1, high-injection q2), Ebers-Moll base
//                                     current with ISE / ISC leakage, 2x2 Jacobian by the quotient rule
//   gen_power_amp.rs:8032-8145        bjt_with_parasitics: inner 2-D Newton (<= 15 iterations, +-4 VT step clamp) for RB / RC / RE,
//                                     external Jacobian J_dev * J_F^-1
//   gen_power_amp.rs:8150-8198        DC_OP / DC_NL_I
//   gen_power_amp.rs:8373-8541        CircuitState::default (+ 50-sample warmup), reset
//   gen_power_amp.rs:8588-8755        set_sample_rate / rebuild_matrices (backward-Euler companion: A = G + C/T, A_neg = C/T)
//   gen_power_amp.rs:8758-8831        invert_n
//   gen_power_amp.rs:8838-12337       process_sample: sparse build_rhs, S*rhs, 16-dim Schur Newton with pivoted 16x16 elimination,
//                                     pnjlim + global step scale, BE-matrix retry, NaN reset, DC blocker, +-30 V clamp
//   power_amp.rs:65-165               RailDynamics (current envelope -> load line -> asymmetric rail one-pole)
//   power_amp.rs:279-465              melange_adapter::PowerAmp: settled-state cache (44 100 silent samples at the codegen rate),
//                                     rail offsets pushed before each sample, divergence guard (non-finite / NR exhausted /
//                                     |node| > 100 V -> reset + hold last good), clamp, rail update from the raw output
// The chord-method fields and the *_sub matrices of the state are never read by the emitted process_sample and are not restated.
#pragma once
#include "ow_tremolo.hpp"   // fast_exp (identical text in every generated file)
#include <cmath>
#include <cstring>

namespace owo {

constexpr int AN = 20, AM = 16;

inline double pa_pnjlim(double vnew, double vold, double vt, double vcrit) {   // gen_power_amp.rs:7527-7545
    if (vnew > vcrit && std::fabs(vnew - vold) > vt + vt) {
        if (vold >= 0.0) {
            const double arg = 1.0 + (vnew - vold) / vt;
            return arg > 0.0 ? vold + vt * std::log(arg) : vcrit;
        }
        return vt * std::log(vnew / vt);
    }
    return vnew;
}

struct PaBjt { double ic, ib, jac[4]; };

// gen_power_amp.rs:7870-8017 with the device's constant parameters (IS / VT / BF / BR are copied into the state by the reference but
// never changed from their DEVICE_* defaults: WurliEngine exposes no setter for them)
inline PaBjt pa_bjt_evaluate(double vbe, double vbc, int d) {
    const double is = PA_DEV_IS[d], vt = PA_DEV_VT[d], nf = PA_DEV_NF[d], nr = PA_DEV_NR[d], beta_f = PA_DEV_BETA_F[d], beta_r = PA_DEV_BETA_R[d];
    const double sign = PA_DEV_SIGN[d], vaf = PA_DEV_VAF[d], var = PA_DEV_VAR[d], ikf = PA_DEV_IKF[d], ikr = PA_DEV_IKR[d];
    const double ise = PA_DEV_ISE[d], ne = PA_DEV_NE[d], isc = PA_DEV_ISC[d], nc = PA_DEV_NC[d];
    const bool use_gp = PA_DEV_USE_GP[d] != 0.0;
    const double vbe_eff = sign * vbe, vbc_eff = sign * vbc;
    const double nf_vt = nf * vt, nr_vt = nr * vt;
    const double exp_be = fast_exp(vbe_eff / nf_vt);
    const double exp_bc = fast_exp(vbc_eff / nr_vt);
    const double exp_be_leak = ise > 0.0 ? fast_exp(vbe_eff / (ne * vt)) : 0.0;
    const double exp_bc_leak = isc > 0.0 ? fast_exp(vbc_eff / (nc * vt)) : 0.0;
    const double i_cc = is * (exp_be - exp_bc);
    const double ib_fwd = is / beta_f * (exp_be - 1.0);
    const double ib_rev = is / beta_r * (exp_bc - 1.0);
    const double ib_leak_be = ise > 0.0 ? ise * (exp_be_leak - 1.0) : 0.0;
    const double ib_leak_bc = isc > 0.0 ? isc * (exp_bc_leak - 1.0) : 0.0;
    const double dib_fwd_dvbe = (is / (beta_f * nf_vt)) * exp_be;
    const double dib_rev_dvbc = (is / (beta_r * nr_vt)) * exp_bc;
    const double dib_leak_dvbe = ise > 0.0 ? (ise / (ne * vt)) * exp_be_leak : 0.0;
    const double dib_leak_dvbc = isc > 0.0 ? (isc / (nc * vt)) * exp_bc_leak : 0.0;
    PaBjt r;
    if (!use_gp) {
        r.ic = sign * (i_cc - is / beta_r * (exp_bc - 1.0));
        r.ib = sign * (ib_fwd + ib_rev + ib_leak_be + ib_leak_bc);
        r.jac[0] = is / nf_vt * exp_be;
        r.jac[1] = -(is / nr_vt) * exp_bc - (is / (beta_r * nr_vt)) * exp_bc;
        r.jac[2] = dib_fwd_dvbe + dib_leak_dvbe;
        r.jac[3] = dib_rev_dvbc + dib_leak_dvbc;
        return r;
    }
    const double q1_denom = 1.0 - vbe_eff / var - vbc_eff / vaf;
    double q1, dq1_dvbe, dq1_dvbc;
    if (q1_denom <= 0.0 || std::fabs(q1_denom) < 1e-30) { q1 = 1.0; dq1_dvbe = 0.0; dq1_dvbc = 0.0; }
    else { q1 = 1.0 / q1_denom; dq1_dvbe = q1 * q1 / var; dq1_dvbc = q1 * q1 / vaf; }
    const double cbe = is * (exp_be - 1.0);
    const double cbc = is * (exp_bc - 1.0);
    const double q2 = cbe / ikf + cbc / ikr;
    const double dq2_dvbe = (is / (nf_vt * ikf)) * exp_be;
    const double dq2_dvbc = (is / (nr_vt * ikr)) * exp_bc;
    const double disc = std::fmax(1.0 + 4.0 * q2, 0.0);
    const double dd = std::sqrt(disc);
    const double dd_dvbe = dd > 1e-15 ? 2.0 * dq2_dvbe / dd : 0.0;
    const double dd_dvbc = dd > 1e-15 ? 2.0 * dq2_dvbc / dd : 0.0;
    const double qb = q1 * (1.0 + dd) / 2.0;
    const double dqb_dvbe = dq1_dvbe * (1.0 + dd) / 2.0 + q1 * dd_dvbe / 2.0;
    const double dqb_dvbc = dq1_dvbc * (1.0 + dd) / 2.0 + q1 * dd_dvbc / 2.0;
    r.ic = sign * (i_cc / qb - is / beta_r * (exp_bc - 1.0));
    r.ib = sign * (ib_fwd + ib_rev + ib_leak_be + ib_leak_bc);
    const double dicc_dvbe = is / nf_vt * exp_be;
    const double dicc_dvbc = -is / nr_vt * exp_bc;
    const double qb2 = std::fmax(qb * qb, 1e-30);
    const double quotient_dvbe = (dicc_dvbe * qb - i_cc * dqb_dvbe) / qb2;
    const double quotient_dvbc = (dicc_dvbc * qb - i_cc * dqb_dvbc) / qb2;
    const double d_bc_term_dvbc = is / (beta_r * nr_vt) * exp_bc;
    r.jac[0] = quotient_dvbe;
    r.jac[1] = quotient_dvbc - d_bc_term_dvbc;
    r.jac[2] = dib_fwd_dvbe + dib_leak_dvbe;
    r.jac[3] = dib_rev_dvbc + dib_leak_dvbc;
    return r;
}

inline double pa_clamp(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }   // f64::clamp (NaN propagates)

// gen_power_amp.rs:8032-8145.  inner_iters (diagnostic tap of the oracle): inner Newton iterations spent.
// Statistics of the Newton passes since the last owo_mpa_stats(reset) -- numbers the GPU mapping was designed from:
// [0] passes, [1] passes with at least one row exchange, [2] row exchanges, [3] sum over passes of the deepest inner loop of the pass,
// [4] sum over passes and devices of the inner-loop trips, [5..20] row exchanges by column
struct PaStats { unsigned long long v[21]; unsigned long long inner_hist[8][16]; unsigned long long deepest_hist[16]; };   // inner_hist[d][t]: calls of device d whose inner loop made t trips; deepest_hist[t]: passes whose deepest device made t trips
inline PaStats& pa_stats() { static thread_local PaStats s = {}; return s; }
inline PaBjt pa_bjt_with_parasitics(double vbe_ext, double vbc_ext, int d, int* inner_iters = nullptr) {
    const double rb = PA_DEV_RB[d], rc = PA_DEV_RC[d], re = PA_DEV_RE[d], vt = PA_DEV_VT[d];
    double vbe_int = vbe_ext, vbc_int = vbc_ext;
    int it = 0;
    for (; it < 15; ++it) {
        const PaBjt e = pa_bjt_evaluate(vbe_int, vbc_int, d);
        const double dic_dvbe = e.jac[0], dic_dvbc = e.jac[1], dib_dvbe = e.jac[2], dib_dvbc = e.jac[3];
        const double f1 = vbe_int - vbe_ext + e.ib * rb + (e.ic + e.ib) * re;
        const double f2 = vbc_int - vbc_ext + e.ib * rb - e.ic * rc;
        if (std::fabs(f1) < 1e-10 && std::fabs(f2) < 1e-10) break;
        const double j11 = 1.0 + dib_dvbe * rb + (dic_dvbe + dib_dvbe) * re;
        const double j12 = dib_dvbc * rb + (dic_dvbc + dib_dvbc) * re;
        const double j21 = dib_dvbe * rb - dic_dvbe * rc;
        const double j22 = 1.0 + dib_dvbc * rb - dic_dvbc * rc;
        const double det = j11 * j22 - j12 * j21;
        if (std::fabs(det) < 1e-30) break;
        const double inv_det = 1.0 / det;
        double dvbe = (j22 * f1 - j12 * f2) * inv_det;
        double dvbc = (j11 * f2 - j21 * f1) * inv_det;
        const double max_step = 4.0 * vt;
        dvbe = pa_clamp(dvbe, -max_step, max_step);
        dvbc = pa_clamp(dvbc, -max_step, max_step);
        vbe_int -= dvbe;
        vbc_int -= dvbc;
    }
    if (inner_iters) { *inner_iters += it; pa_stats().inner_hist[d][it < 15 ? it : 15] += 1; }
    const PaBjt e = pa_bjt_evaluate(vbe_int, vbc_int, d);
    const double dic_dvbe = e.jac[0], dic_dvbc = e.jac[1], dib_dvbe = e.jac[2], dib_dvbc = e.jac[3];
    const double j11 = 1.0 + dib_dvbe * rb + (dic_dvbe + dib_dvbe) * re;
    const double j12 = dib_dvbc * rb + (dic_dvbc + dib_dvbc) * re;
    const double j21 = dib_dvbe * rb - dic_dvbe * rc;
    const double j22 = 1.0 + dib_dvbc * rb - dic_dvbc * rc;
    const double det = j11 * j22 - j12 * j21;
    if (std::fabs(det) < 1e-30) return e;
    const double inv_det = 1.0 / det;
    const double fi11 = j22 * inv_det, fi12 = -j12 * inv_det, fi21 = -j21 * inv_det, fi22 = j11 * inv_det;
    PaBjt r;
    r.ic = e.ic; r.ib = e.ib;
    r.jac[0] = dic_dvbe * fi11 + dic_dvbc * fi21;
    r.jac[1] = dic_dvbe * fi12 + dic_dvbc * fi22;
    r.jac[2] = dib_dvbe * fi11 + dib_dvbc * fi21;
    r.jac[3] = dib_dvbe * fi12 + dib_dvbc * fi22;
    return r;
}

// invert_n, gen_power_amp.rs:8758-8831 (None -> false)
inline bool pa_invert_n(const double a[AN][AN], double result[AN][AN]) {
    double lu[AN][AN];
    int perm[AN];
    std::memcpy(lu, a, sizeof lu);
    for (int i = 0; i < AN; ++i) perm[i] = i;
    for (int k = 0; k < AN; ++k) {
        int max_row = k;
        double max_val = std::fabs(lu[k][k]);
        for (int i = k + 1; i < AN; ++i) {
            const double v = std::fabs(lu[i][k]);
            if (v > max_val) { max_val = v; max_row = i; }
        }
        if (max_val < 1e-30) return false;
        if (max_row != k) {
            for (int j = 0; j < AN; ++j) std::swap(lu[k][j], lu[max_row][j]);
            std::swap(perm[k], perm[max_row]);
        }
        const double pivot = lu[k][k];
        for (int i = k + 1; i < AN; ++i) {
            const double m = lu[i][k] / pivot;
            lu[i][k] = m;
            for (int j = k + 1; j < AN; ++j) lu[i][j] -= m * lu[k][j];
        }
    }
    for (int col = 0; col < AN; ++col) {
        double b[AN] = {0.0};
        int start = AN;
        for (int i = 0; i < AN; ++i)
            if (perm[i] == col) { b[i] = 1.0; start = i; break; }
        for (int i = start + 1; i < AN; ++i) {
            double sum = b[i];
            for (int j = start; j < i; ++j) sum -= lu[i][j] * b[j];
            b[i] = sum;
        }
        for (int i = AN - 1; i >= 0; --i) {
            double sum = b[i];
            for (int j = i + 1; j < AN; ++j) sum -= lu[i][j] * b[j];
            const double pivot = lu[i][i];
            if (std::fabs(pivot) < 1e-30) return false;
            b[i] = sum / pivot;
        }
        for (int i = 0; i < AN; ++i) result[i][col] = b[i];
    }
    return true;
}

struct PaCircuit {
    double v_prev[AN], i_nl_prev[AM], i_nl_prev_prev[AM], dc_operating_point[AN], input_prev;
    uint32_t last_nr_iterations;
    double dc_block_x_prev, dc_block_y_prev, dc_block_r;
    double diag_peak_output;
    uint64_t diag_clamp_count, diag_nr_max_iter_count, diag_be_fallback_count, diag_nan_reset_count;
    double a_neg[AN][AN], a_neg_be[AN][AN], s[AN][AN], k[AM][AM], s_ni[AN][AM], s_be[AN][AN], k_be[AM][AM], s_ni_be[AN][AM];
    double v_rail_pos_offset, v_rail_neg_offset;
    // oracle-only taps of the last process_sample call (solver internals for GPU-vs-oracle comparison)
    uint32_t tap_outer_iters, tap_inner_iters, tap_be_used, tap_singular;

    void load_default_matrices() {
        std::memcpy(a_neg, PA_A_NEG_DEFAULT, sizeof a_neg); std::memcpy(a_neg_be, PA_A_NEG_BE_DEFAULT, sizeof a_neg_be);
        std::memcpy(s, PA_S_DEFAULT, sizeof s); std::memcpy(k, PA_K_DEFAULT, sizeof k); std::memcpy(s_ni, PA_S_NI_DEFAULT, sizeof s_ni);
        std::memcpy(s_be, PA_S_BE_DEFAULT, sizeof s_be); std::memcpy(k_be, PA_K_BE_DEFAULT, sizeof k_be); std::memcpy(s_ni_be, PA_S_NI_BE_DEFAULT, sizeof s_ni_be);
    }
    void init_default() {   // impl Default, gen_power_amp.rs:8373-8461
        for (int i = 0; i < AN; ++i) { v_prev[i] = PA_DC_OP[i]; dc_operating_point[i] = PA_DC_OP[i]; }
        for (int i = 0; i < AM; ++i) { i_nl_prev[i] = PA_DC_NL_I[i]; i_nl_prev_prev[i] = PA_DC_NL_I[i]; }
        input_prev = 0.0; last_nr_iterations = 0;
        dc_block_x_prev = PA_DC_BLOCK_X0; dc_block_y_prev = 0.0; dc_block_r = PA_DC_BLOCK_R;
        diag_peak_output = 0.0; diag_clamp_count = diag_nr_max_iter_count = diag_be_fallback_count = diag_nan_reset_count = 0;
        load_default_matrices();
        v_rail_pos_offset = 0.0; v_rail_neg_offset = 0.0;
        tap_outer_iters = tap_inner_iters = tap_be_used = tap_singular = 0;
        for (int i = 0; i < 50; ++i) process_sample(0.0);   // warmup()
    }
    // gen_power_amp.rs:8588-8622
    void set_sample_rate(double sample_rate) {
        if (!(sample_rate > 0.0 && std::isfinite(sample_rate))) return;
        if (std::fabs(sample_rate - PA_SAMPLE_RATE) < 0.5) {
            load_default_matrices();
            dc_block_r = PA_DC_BLOCK_R; dc_block_x_prev = 0.0; dc_block_y_prev = 0.0;
            return;
        }
        const double internal_rate = sample_rate * 1.0;
        rebuild_matrices(internal_rate);
        dc_block_r = 1.0 - 2.0 * 3.14159265358979323846 * 5.0 / internal_rate;
        dc_block_x_prev = 0.0; dc_block_y_prev = 0.0;
    }
    // gen_power_amp.rs:8624-8755 (the *_sub set is built there too and never read)
    void rebuild_matrices(double internal_rate) {
        const double alpha = internal_rate, alpha_be = internal_rate;
        static thread_local double a[AN][AN], a_be[AN][AN], inv[AN][AN];
        for (int i = 0; i < AN; ++i)
            for (int j = 0; j < AN; ++j) {
                a[i][j] = PA_G[i][j] + alpha * PA_C[i][j];
                a_neg[i][j] = alpha * PA_C[i][j];
                a_be[i][j] = PA_G[i][j] + alpha_be * PA_C[i][j];
                a_neg_be[i][j] = alpha_be * PA_C[i][j];
            }
        for (int i = 18; i < 20; ++i)
            for (int j = 0; j < AN; ++j) { a_neg[i][j] = 0.0; a_neg_be[i][j] = 0.0; }
        auto derive = [&](double S[AN][AN], double K[AM][AM], double SNI[AN][AM]) {
            for (int i = 0; i < AM; ++i)
                for (int j = 0; j < AM; ++j) {
                    double sum = 0.0;
                    for (int aa = 0; aa < AN; ++aa) {
                        double s_ni_aj = 0.0;
                        for (int b = 0; b < AN; ++b) s_ni_aj += S[aa][b] * PA_N_I[b][j];
                        sum += PA_N_V[i][aa] * s_ni_aj;
                    }
                    K[i][j] = sum;
                }
            for (int i = 0; i < AN; ++i)
                for (int j = 0; j < AM; ++j) {
                    double sum = 0.0;
                    for (int aa = 0; aa < AN; ++aa) sum += S[i][aa] * PA_N_I[aa][j];
                    SNI[i][j] = sum;
                }
        };
        if (pa_invert_n(a, inv)) { std::memcpy(s, inv, sizeof s); derive(s, k, s_ni); }
        if (pa_invert_n(a_be, inv)) { std::memcpy(s_be, inv, sizeof s_be); derive(s_be, k_be, s_ni_be); }
    }

    // One Newton solve: the trapezoid-slot sweep (:8956-10680) or the BE retry (:10745-12245).  Returns through last_nr_iterations.
    void newton(const double p[AM], const double kk[AM][AM], double i_nl[AM], bool be) {
        for (int iter = 0; iter < PA_MAX_ITER; ++iter) {
            double vd[AM];
            for (int i = 0; i < AM; ++i) {
                double acc = p[i];
                for (int j = 0; j < AM; ++j) acc = acc + kk[i][j] * i_nl[j];
                vd[i] = acc;
            }
            double idev[AM], jdev[AM][2];   // jdev[i] = row i of its device's 2x2 block (columns 2d, 2d+1)
            int inner_max = 0, swaps = 0;
            for (int d = 0; d < 8; ++d) {
                int inner = 0;
                const PaBjt e = pa_bjt_with_parasitics(vd[2 * d], vd[2 * d + 1], d, &inner);
                tap_inner_iters += (uint32_t)inner;
                inner_max = inner > inner_max ? inner : inner_max;
                pa_stats().v[4] += (unsigned long long)inner;
                idev[2 * d] = e.ic; idev[2 * d + 1] = e.ib;
                jdev[2 * d][0] = e.jac[0]; jdev[2 * d][1] = e.jac[1]; jdev[2 * d + 1][0] = e.jac[2]; jdev[2 * d + 1][1] = e.jac[3];
            }
            double f[AM], a[AM][AM], b[AM];
            for (int i = 0; i < AM; ++i) f[i] = i_nl[i] - idev[i];
            for (int i = 0; i < AM; ++i) {
                const int d2 = i & ~1;
                for (int j = 0; j < AM; ++j) a[i][j] = (i == j ? 1.0 : 0.0) - jdev[i][0] * kk[d2][j] - jdev[i][1] * kk[d2 + 1][j];
                b[i] = f[i];
            }
            bool singular = false;
            for (int col = 0; col < AM; ++col) {
                int max_row = col;
                double max_val = std::fabs(a[col][col]);
                for (int row = col + 1; row < AM; ++row)
                    if (std::fabs(a[row][col]) > max_val) { max_val = std::fabs(a[row][col]); max_row = row; }
                if (max_val < 1e-15) { singular = true; break; }
                if (max_row != col) {
                    for (int j = 0; j < AM; ++j) std::swap(a[col][j], a[max_row][j]);
                    std::swap(b[col], b[max_row]);
                    ++swaps; pa_stats().v[5 + col] += 1;
                }
                const double pivot = a[col][col];
                for (int row = col + 1; row < AM; ++row) {
                    const double factor = a[row][col] / pivot;
                    for (int j = col + 1; j < AM; ++j) a[row][j] -= factor * a[col][j];
                    b[row] -= factor * b[col];
                }
            }
            { PaStats& st = pa_stats(); st.deepest_hist[inner_max < 15 ? inner_max : 15] += 1; st.v[0] += 1; st.v[1] += swaps ? 1 : 0; st.v[2] += (unsigned long long)swaps; st.v[3] += (unsigned long long)inner_max; }
            if (!singular) {
                for (int i = AM - 1; i >= 0; --i) {
                    double sum = b[i];
                    for (int j = i + 1; j < AM; ++j) sum -= a[i][j] * b[j];
                    if (std::fabs(a[i][i]) < 1e-15) { singular = true; break; }
                    b[i] = sum / a[i][i];
                }
            }
            if (singular) {
                tap_singular += 1;
                for (int i = 0; i < AM; ++i) {
                    const double cl = be ? 0.01 : std::fmax(std::fabs(i_nl[i]) * 0.1, 0.01);
                    i_nl[i] -= pa_clamp(f[i] * 0.5, -cl, cl);
                }
                continue;
            }
            const double* delta = b;
            if (!be) {   // :9849-10622
                double dv_trial[AM], v_lim[AM];
                double i_trial[AM];
                for (int i = 0; i < AM; ++i) i_trial[i] = i_nl[i] - delta[i];
                for (int i = 0; i < AM; ++i) {
                    double acc = p[i];
                    for (int j = 0; j < AM; ++j) acc = acc + kk[i][j] * i_trial[j];
                    dv_trial[i] = acc - vd[i];
                    v_lim[i] = std::fabs(dv_trial[i]) > 1e-4 ? pa_pnjlim(acc, vd[i], PA_DEV_VT[i >> 1], PA_DEV_VCRIT[i >> 1]) : acc;
                }
                bool any_limited = false;
                double global_alpha = 1.0;
                for (int i = 0; i < AM; ++i) {
                    const double dv_lim = v_lim[i] - vd[i];
                    if (std::fabs(dv_trial[i]) > 1e-15) {
                        const double r = dv_trial[i] * dv_lim < 0.0 ? 0.0 : pa_clamp(dv_lim / dv_trial[i], 0.0, 1.0);
                        if (r < global_alpha) { global_alpha = r; any_limited = true; }
                    }
                }
                {
                    double max_dv = std::fabs(dv_trial[0] * global_alpha);
                    for (int i = 1; i < AM; ++i) max_dv = std::fmax(max_dv, std::fabs(dv_trial[i] * global_alpha));
                    if (max_dv > 3.5) { global_alpha *= std::fmax(3.5 / max_dv, 0.1); any_limited = true; }
                }
                for (int i = 0; i < AM; ++i) i_nl[i] -= global_alpha * delta[i];
                if (!any_limited) {
                    bool conv = true;
                    for (int i = 0; i < AM; ++i) {
                        const double dv = dv_trial[i] * global_alpha;
                        const double thr = 1e-3 * std::fmax(std::fabs(vd[i]), std::fabs(vd[i] + dv)) + 1e-6;
                        if (std::fabs(dv) > thr) conv = false;
                    }
                    if (conv) { last_nr_iterations = (uint32_t)iter; return; }
                }
            } else {     // :11634-12225
                double dv[AM], alpha[AM];
                for (int i = 0; i < AM; ++i) {
                    double acc = kk[i][0] * delta[0];
                    for (int j = 1; j < AM; ++j) acc = acc + kk[i][j] * delta[j];
                    dv[i] = -acc;
                    alpha[i] = 1.0;
                }
                bool any_limited = false;
                for (int i = 0; i < AM; ++i) {
                    if (std::fabs(dv[i]) > 1e-4) {
                        const double vl = pa_pnjlim(vd[i] + dv[i], vd[i], PA_DEV_VT[i >> 1], PA_DEV_VCRIT[i >> 1]);
                        const double ratio = std::fmax((vl - vd[i]) / dv[i], 0.01);
                        if (ratio < alpha[i]) { alpha[i] = ratio; if (ratio < 1.0) any_limited = true; }
                    }
                }
                for (int d = 0; d < 8; ++d) { const double m = std::fmin(alpha[2 * d], alpha[2 * d + 1]); alpha[2 * d] = m; alpha[2 * d + 1] = m; }
                double max_dv = std::fabs(dv[0] * alpha[0]);
                for (int i = 1; i < AM; ++i) max_dv = std::fmax(max_dv, std::fabs(dv[i] * alpha[i]));
                if (max_dv > 3.5) {
                    const double factor = std::fmax(3.5 / max_dv, 0.1);
                    for (int i = 0; i < AM; ++i) alpha[i] *= factor;
                }
                for (int i = 0; i < AM; ++i) i_nl[i] -= alpha[i] * delta[i];
                if (!any_limited) {
                    bool conv = true;
                    for (int i = 0; i < AM; ++i) {
                        const double step = dv[i] * alpha[i];
                        const double v_new = vd[i] + step;
                        const double thr = 1e-3 * std::fmax(std::fabs(vd[i]), std::fabs(v_new)) + 1e-6;
                        if (std::fabs(step) > thr) conv = false;
                    }
                    if (conv) { last_nr_iterations = (uint32_t)iter; return; }
                }
            }
        }
    }

    // gen_power_amp.rs:8838-12337
    double process_sample(double input_in) {
        const double input = std::isfinite(input_in) ? pa_clamp(input_in, -100.0, 100.0) : 0.0;
        for (int i = 0; i < AN; ++i) v_prev[i] = v_prev[i] + 1e-25 - 1e-25;
        for (int i = 0; i < AM; ++i) i_nl_prev[i] = i_nl_prev[i] + 1e-25 - 1e-25;
        double rhs[AN];
        for (int i = 0; i < AN; ++i) rhs[i] = PA_RHS_CONST[i];
        for (int q = 0; q < PA_RHS_NNZ; ++q) {
            const int i = (int)PA_RHS_NZ_ROW[q], j = (int)PA_RHS_NZ_COL[q];
            rhs[i] += a_neg[i][j] * v_prev[j];
        }
        const double input_conductance = 1.0 / PA_INPUT_RESISTANCE;
        rhs[0] += input * input_conductance;
        input_prev = input;
        rhs[18] += v_rail_pos_offset;
        rhs[19] += v_rail_neg_offset;
        double v_pred[AN];
        for (int i = 0; i < AN; ++i) {
            double sum = 0.0;
            for (int j = 0; j < AN; ++j) sum += s[i][j] * rhs[j];
            v_pred[i] = sum;
        }
        double p[AM];
        for (int i = 0; i < AM; ++i) {
            const int na = (int)PA_P_NODE_A[i], nb = (int)PA_P_NODE_B[i];
            p[i] = PA_N_V[i][na] * v_pred[na] + PA_N_V[i][nb] * v_pred[nb];
        }
        double i_nl[AM];
        for (int i = 0; i < AM; ++i) i_nl[i] = 2.0 * i_nl_prev[i] - i_nl_prev_prev[i];
        last_nr_iterations = (uint32_t)PA_MAX_ITER;
        tap_inner_iters = 0; tap_be_used = 0; tap_singular = 0;
        newton(p, k, i_nl, false);
        tap_outer_iters = last_nr_iterations;
        double v[AN];
        for (int i = 0; i < AN; ++i) {
            v[i] = v_pred[i];
            for (int j = 0; j < AM; ++j) v[i] += s_ni[i][j] * i_nl[j];
        }
        const bool converged = last_nr_iterations < (uint32_t)PA_MAX_ITER;
        if (!converged) {
            diag_nr_max_iter_count += 1;
            diag_be_fallback_count += 1;
            tap_be_used = 1;
            double rhs_be[AN], v_pred_be[AN], p_be[AM];
            for (int i = 0; i < AN; ++i) {
                double sum = PA_RHS_CONST_BE[i];
                for (int j = 0; j < AN; ++j) sum += a_neg_be[i][j] * v_prev[j];
                for (int j = 0; j < AM; ++j) sum += PA_N_I[i][j] * i_nl_prev[j];
                rhs_be[i] = sum;
            }
            rhs_be[0] += input * input_conductance;
            for (int i = 0; i < AN; ++i) {
                double sum = 0.0;
                for (int j = 0; j < AN; ++j) sum += s_be[i][j] * rhs_be[j];
                v_pred_be[i] = sum;
            }
            for (int i = 0; i < AM; ++i) {
                double sum = 0.0;
                for (int j = 0; j < AN; ++j) sum += PA_N_V[i][j] * v_pred_be[j];
                p_be[i] = sum;
            }
            for (int i = 0; i < AM; ++i) i_nl[i] = 2.0 * i_nl_prev[i] - i_nl_prev_prev[i];
            newton(p_be, k_be, i_nl, true);     // leaves last_nr_iterations at MAX_ITER when it does not converge either
            for (int i = 0; i < AN; ++i) {
                v[i] = v_pred_be[i];
                for (int j = 0; j < AM; ++j) v[i] += s_ni_be[i][j] * i_nl[j];
            }
        }
        bool finite = true;
        for (int i = 0; i < AN; ++i) finite = finite && std::isfinite(v[i]);
        if (!finite) {
            for (int i = 0; i < AN; ++i) v_prev[i] = dc_operating_point[i];
            for (int i = 0; i < AM; ++i) { i_nl_prev[i] = PA_DC_NL_I[i]; i_nl_prev_prev[i] = PA_DC_NL_I[i]; }
            input_prev = 0.0; dc_block_x_prev = 0.0; dc_block_y_prev = 0.0;
            diag_nan_reset_count += 1;
            return PA_DC_BLOCK_X0;
        }
        for (int i = 0; i < AN; ++i) v_prev[i] = v[i];
        for (int i = 0; i < AM; ++i) { i_nl_prev_prev[i] = i_nl_prev[i]; i_nl_prev[i] = i_nl[i]; }
        double raw_out = v[8];
        if (!std::isfinite(raw_out)) raw_out = 0.0;
        const double dc_blocked = raw_out - dc_block_x_prev + dc_block_r * dc_block_y_prev;
        dc_block_x_prev = raw_out;
        dc_block_y_prev = dc_blocked;
        const double scaled = dc_blocked * 1.0;
        const double abs_out = std::fabs(scaled);
        if (abs_out > diag_peak_output) diag_peak_output = abs_out;
        if (abs_out > 3e1) diag_clamp_count += 1;
        return pa_clamp(scaled, -3e1, 3e1);
    }
};

// power_amp.rs:65-165
struct RailDynamics {
    double v_rail_pos, v_rail_neg, i_avg_pos, i_avg_neg, alpha_attack, alpha_release, alpha_i_avg;
    void init(double sample_rate) {
        v_rail_pos = 22.5; v_rail_neg = 22.5; i_avg_pos = 0.0; i_avg_neg = 0.0;
        set_sample_rate(sample_rate);
    }
    void set_sample_rate(double sample_rate) {
        const double dt = 1.0 / sample_rate;
        alpha_attack = 1.0 - std::exp(-dt / 0.008);
        alpha_release = 1.0 - std::exp(-dt / 0.015);
        alpha_i_avg = 1.0 - std::exp(-dt / 0.030);
    }
    void reset() { v_rail_pos = 22.5; v_rail_neg = 22.5; i_avg_pos = 0.0; i_avg_neg = 0.0; }
    void step(double v_out) {
        const double i_pos = std::fmax(v_out / 8.0, 0.0);
        const double i_neg = std::fmax(-v_out / 8.0, 0.0);
        i_avg_pos += alpha_i_avg * (i_pos - i_avg_pos);
        i_avg_neg += alpha_i_avg * (i_neg - i_avg_neg);
        const double target_pos = 24.5 - i_avg_pos * 3.5;
        const double target_neg = 24.5 - i_avg_neg * 3.5;
        const double alpha_p = target_pos < v_rail_pos ? alpha_attack : alpha_release;
        const double alpha_n = target_neg < v_rail_neg ? alpha_attack : alpha_release;
        v_rail_pos += alpha_p * (target_pos - v_rail_pos);
        v_rail_neg += alpha_n * (target_neg - v_rail_neg);
    }
};

// power_amp.rs:279-465
struct MelangePowerAmp {
    PaCircuit state;
    double sample_rate, last_good;
    RailDynamics rails;
    bool rail_sag_on;
    uint64_t guard_resets;     // oracle-only tap: how often the divergence guard fired

    static const PaCircuit& settled() {     // SETTLED_STATE: default state + 44 100 silent samples at the codegen matrices
        static const PaCircuit st = [] {
            PaCircuit c;
            c.init_default();
            for (int i = 0; i < 44100; ++i) c.process_sample(0.0);
            return c;
        }();
        return st;
    }
    void init_state() {
        state = settled();
        if (std::fabs(sample_rate - PA_SAMPLE_RATE) > 0.5) state.set_sample_rate(sample_rate);
    }
    void init(double sr) {   // new_at_sample_rate
        sample_rate = sr; init_state(); last_good = 0.0; rails.init(sr); rail_sag_on = true; guard_resets = 0;
    }
    void set_rail_sag(bool on) {
        rail_sag_on = on;
        if (!on) { state.v_rail_pos_offset = 0.0; state.v_rail_neg_offset = 0.0; }
    }
    void reset() { init_state(); rails.reset(); }
    double process(double input) {
        if (rail_sag_on) {
            state.v_rail_pos_offset = rails.v_rail_pos - 22.5;
            state.v_rail_neg_offset = rails.v_rail_neg - 22.5;
        }
        const double raw = state.process_sample(input);
        const double result = raw / 22.0;
        const bool nr_failed = state.last_nr_iterations >= (uint32_t)PA_MAX_ITER - 1u;
        bool state_insane = false;
        for (int i = 0; i < AN; ++i) state_insane = state_insane || !std::isfinite(state.v_prev[i]) || std::fabs(state.v_prev[i]) > 100.0;
        if (!std::isfinite(result) || nr_failed || state_insane) {
            reset();
            guard_resets += 1;
            return last_good;
        }
        const double clamped = pa_clamp(result, -1.0, 1.0);
        last_good = clamped;
        if (rail_sag_on) rails.step(raw);
        return clamped;
    }
};

}  // namespace owo
