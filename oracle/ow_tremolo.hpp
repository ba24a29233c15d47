// ORACLE (test infrastructure, CPU, f64) -- NOT part of the shipped product path.
//
// Restates the default tremolo of the reference (citations into
// /root/reference/crates/openwurli-dsp/src/):
//   gen_tremolo.rs:1140-1218   fast_exp / fast_ln / pnjlim
//   gen_tremolo.rs:1546-1633   bjt_evaluate (Ebers-Moll branch; USE_GP=false, ISE=ISC=0 for both devices)
//   gen_tremolo.rs:1853-2075   CircuitState default / warmup
//   gen_tremolo.rs:2111-2342   set_sample_rate / rebuild_matrices / invert_n
//   gen_tremolo.rs:2353-3116   process_sample (Schur-complement NR, BE fallback, NaN reset)
//   tremolo.rs:16-216          LED drive -> CdS envelope -> power-law R -> vibrato-pot divider
// Circuit matrices are data (data/ow_gen_data.h, TREM_*).  The emitted sparsity of the
// generated code (which matrix entries each formula touches) is followed literally.
#pragma once
#include "ow_tables.hpp"
#include <cstring>

namespace owo {
// Newton sweeps and row exchanges per column of the Twin-T solver since process start (statistics for the GPU mapping, test infrastructure)
inline unsigned long long* trem_stats() { static thread_local unsigned long long v[5] = {0, 0, 0, 0, 0}; return v; }


// gen_tremolo.rs:1140-1166
inline double fast_exp(double x0) {
    const double x = rclamp(x0, -40.0, 40.0);
    const double LN2_INV = 1.4426950408889634;  // std::f64::consts::LOG2_E
    const double LN2_HI = 0.6931471803691238;
    const double LN2_LO = 1.9082149292705877e-10;
    const double SHIFT = 6755399441055744.0;
    const double z = x * LN2_INV + SHIFT;
    int64_t zb, sb;
    std::memcpy(&zb, &z, 8);
    std::memcpy(&sb, &SHIFT, 8);
    const int64_t n_i64 = zb - sb;
    const double n = (double)n_i64;
    const double f = (x - n * LN2_HI) - n * LN2_LO;
    const double p = 1.0 + f * (1.0 + f * (0.5 + f * (0.16666666666666607 + f * (0.04166666666665876 + f * 0.008333333333492337))));
    const uint64_t pb = ((uint64_t)(1023 + n_i64)) << 52;
    double pow2n;
    std::memcpy(&pow2n, &pb, 8);
    return p * pow2n;
}

// gen_tremolo.rs:1203-1218
inline double pnjlim(double vnew, double vold, double vt, double vcrit) {
    if (vnew > vcrit && std::fabs(vnew - vold) > vt + vt) {
        if (vold >= 0.0) {
            const double arg = 1.0 + (vnew - vold) / vt;
            if (arg > 0.0) return vold + vt * std::log(arg);
            return vcrit;
        }
        return vt * std::log(vnew / vt);
    }
    return vnew;
}

struct BjtEval { double ic, ib, jac[4]; };
// gen_tremolo.rs:1546-1633, Ebers-Moll path with sign=+1, ISE=ISC=0 (DEVICE_*_ constants :1098-1132)
inline BjtEval bjt_evaluate_em(double vbe, double vbc, double is, double vt, double nf, double nr, double beta_f, double beta_r) {
    const double sign = 1.0;
    const double vbe_eff = sign * vbe, vbc_eff = sign * vbc;
    const double nf_vt = nf * vt, nr_vt = nr * vt;
    const double exp_be = fast_exp(vbe_eff / nf_vt);
    const double exp_bc = fast_exp(vbc_eff / nr_vt);
    const double i_cc = is * (exp_be - exp_bc);
    const double ib_fwd = is / beta_f * (exp_be - 1.0);
    const double ib_rev = is / beta_r * (exp_bc - 1.0);
    const double ib_leak_be = 0.0, ib_leak_bc = 0.0;
    const double dib_fwd_dvbe = (is / (beta_f * nf_vt)) * exp_be;
    const double dib_rev_dvbc = (is / (beta_r * nr_vt)) * exp_bc;
    const double dib_leak_dvbe = 0.0, dib_leak_dvbc = 0.0;
    BjtEval r;
    r.ic = sign * (i_cc - is / beta_r * (exp_bc - 1.0));
    r.ib = sign * (ib_fwd + ib_rev + ib_leak_be + ib_leak_bc);
    r.jac[0] = is / nf_vt * exp_be;
    r.jac[1] = -(is / nr_vt) * exp_bc - (is / (beta_r * nr_vt)) * exp_bc;
    r.jac[2] = dib_fwd_dvbe + dib_leak_dvbe;
    r.jac[3] = dib_rev_dvbc + dib_leak_dvbc;
    return r;
}

constexpr int TN = 7, TM = 4;

// gen_tremolo.rs:2273-2342 (LU with partial pivoting, factor-then-solve per column)
inline bool trem_invert_n(const double a[TN][TN], double result[TN][TN]) {
    double lu[TN][TN];
    int perm[TN];
    for (int i = 0; i < TN; ++i) { perm[i] = i; for (int j = 0; j < TN; ++j) lu[i][j] = a[i][j]; }
    for (int k = 0; k < TN; ++k) {
        int max_row = k;
        double max_val = std::fabs(lu[k][k]);
        for (int i = k + 1; i < TN; ++i) {
            const double v = std::fabs(lu[i][k]);
            if (v > max_val) { max_val = v; max_row = i; }
        }
        if (max_val < 1e-30) return false;
        if (max_row != k) {
            for (int j = 0; j < TN; ++j) std::swap(lu[k][j], lu[max_row][j]);
            std::swap(perm[k], perm[max_row]);
        }
        const double pivot = lu[k][k];
        for (int i = k + 1; i < TN; ++i) {
            const double m = lu[i][k] / pivot;
            lu[i][k] = m;
            for (int j = k + 1; j < TN; ++j) lu[i][j] -= m * lu[k][j];
        }
    }
    for (int col = 0; col < TN; ++col) {
        double b[TN] = {0, 0, 0, 0, 0, 0, 0};
        int start = TN;
        for (int i = 0; i < TN; ++i) {
            if (perm[i] == col) { b[i] = 1.0; start = i; break; }
        }
        for (int i = start + 1; i < TN; ++i) {
            double sum = b[i];
            for (int j = start; j < i; ++j) sum -= lu[i][j] * b[j];
            b[i] = sum;
        }
        for (int i = TN - 1; i >= 0; --i) {
            double sum = b[i];
            for (int j = i + 1; j < TN; ++j) sum -= lu[i][j] * b[j];
            const double pivot = lu[i][i];
            if (std::fabs(pivot) < 1e-30) return false;
            b[i] = sum / pivot;
        }
        for (int i = 0; i < TN; ++i) result[i][col] = b[i];
    }
    return true;
}

struct TremCircuit {
    double v_prev[TN], i_nl_prev[TM], i_nl_prev_prev[TM], input_prev;
    uint32_t last_nr_iterations;
    uint64_t diag_nr_max_iter_count, diag_be_fallback_count, diag_nan_reset_count;
    double a_neg[TN][TN], a_neg_be[TN][TN];
    double s[TN][TN], k[TM][TM], s_ni[TN][TM];
    double s_be[TN][TN], k_be[TM][TM], s_ni_be[TN][TM];

    // gen_tremolo.rs:1965-2024 (Default) -- includes the 50-sample warmup at the codegen matrices
    void init_default() {
        for (int i = 0; i < TN; ++i) v_prev[i] = TREM_DC_OP[i];
        for (int i = 0; i < TM; ++i) { i_nl_prev[i] = TREM_DC_NL_I[i]; i_nl_prev_prev[i] = TREM_DC_NL_I[i]; }
        input_prev = 0.0;
        last_nr_iterations = 0;
        diag_nr_max_iter_count = diag_be_fallback_count = diag_nan_reset_count = 0;
        std::memcpy(a_neg, TREM_A_NEG_DEFAULT, sizeof a_neg);
        std::memcpy(a_neg_be, TREM_A_NEG_BE_DEFAULT, sizeof a_neg_be);
        std::memcpy(s, TREM_S_DEFAULT, sizeof s);
        std::memcpy(k, TREM_K_DEFAULT, sizeof k);
        std::memcpy(s_ni, TREM_S_NI_DEFAULT, sizeof s_ni);
        std::memcpy(s_be, TREM_S_BE_DEFAULT, sizeof s_be);
        std::memcpy(k_be, TREM_K_BE_DEFAULT, sizeof k_be);
        std::memcpy(s_ni_be, TREM_S_NI_BE_DEFAULT, sizeof s_ni_be);
        for (int i = 0; i < 50; ++i) process_sample(0.0);  // warmup(), gen_tremolo.rs:2071-2075
    }

    // gen_tremolo.rs:2111-2137
    void set_sample_rate(double sr) {
        if (!(sr > 0.0 && std::isfinite(sr))) return;
        if (std::fabs(sr - TREM_SAMPLE_RATE) < 0.5) {
            std::memcpy(a_neg, TREM_A_NEG_DEFAULT, sizeof a_neg);
            std::memcpy(a_neg_be, TREM_A_NEG_BE_DEFAULT, sizeof a_neg_be);
            std::memcpy(s, TREM_S_DEFAULT, sizeof s);
            std::memcpy(s_be, TREM_S_BE_DEFAULT, sizeof s_be);
            std::memcpy(k, TREM_K_DEFAULT, sizeof k);
            std::memcpy(s_ni, TREM_S_NI_DEFAULT, sizeof s_ni);
            std::memcpy(k_be, TREM_K_BE_DEFAULT, sizeof k_be);
            std::memcpy(s_ni_be, TREM_S_NI_BE_DEFAULT, sizeof s_ni_be);
            return;
        }
        rebuild_matrices(sr * 1.0);
    }

    // gen_tremolo.rs:2139-2260 (the *_sub matrices are rebuilt there too but never read by process_sample)
    void rebuild_matrices(double rate) {
        const double alpha = 2.0 * rate, alpha_be = rate;
        double a[TN][TN], a_be[TN][TN];
        for (int i = 0; i < TN; ++i)
            for (int j = 0; j < TN; ++j) {
                a[i][j] = TREM_G[i][j] + alpha * TREM_C[i][j];
                a_neg[i][j] = alpha * TREM_C[i][j] - TREM_G[i][j];
                a_be[i][j] = TREM_G[i][j] + alpha_be * TREM_C[i][j];
                a_neg_be[i][j] = alpha_be * TREM_C[i][j];
            }
        for (int j = 0; j < TN; ++j) { a_neg[6][j] = 0.0; a_neg_be[6][j] = 0.0; }
        double inv[TN][TN];
        if (trem_invert_n(a, inv)) {
            std::memcpy(s, inv, sizeof s);
            for (int i = 0; i < TM; ++i)
                for (int j = 0; j < TM; ++j) {
                    double sum = 0.0;
                    for (int aa = 0; aa < TN; ++aa) {
                        double s_ni_aj = 0.0;
                        for (int b = 0; b < TN; ++b) s_ni_aj += s[aa][b] * TREM_N_I[b][j];
                        sum += TREM_N_V[i][aa] * s_ni_aj;
                    }
                    k[i][j] = sum;
                }
            for (int i = 0; i < TN; ++i)
                for (int j = 0; j < TM; ++j) {
                    double sum = 0.0;
                    for (int aa = 0; aa < TN; ++aa) sum += s[i][aa] * TREM_N_I[aa][j];
                    s_ni[i][j] = sum;
                }
        }
        if (trem_invert_n(a_be, inv)) {
            std::memcpy(s_be, inv, sizeof s_be);
            for (int i = 0; i < TM; ++i)
                for (int j = 0; j < TM; ++j) {
                    double sum = 0.0;
                    for (int aa = 0; aa < TN; ++aa) {
                        double s_ni_aj = 0.0;
                        for (int b = 0; b < TN; ++b) s_ni_aj += s_be[aa][b] * TREM_N_I[b][j];
                        sum += TREM_N_V[i][aa] * s_ni_aj;
                    }
                    k_be[i][j] = sum;
                }
            for (int i = 0; i < TN; ++i)
                for (int j = 0; j < TM; ++j) {
                    double sum = 0.0;
                    for (int aa = 0; aa < TN; ++aa) sum += s_be[i][aa] * TREM_N_I[aa][j];
                    s_ni_be[i][j] = sum;
                }
        }
    }

    // One NR sweep shared by the trapezoidal (sparse v_d, gen_tremolo.rs:2423-2438) and the
    // BE-fallback (dense v_d, :2798-2817) solves.  Returns true on convergence at `iter`.
    // kk = K or K_be; be = which limiter/convergence variant.
    bool nr_solve(const double p[TM], const double kk[TM][TM], double i_nl[TM], bool be, uint32_t& iters_out) {
        for (int iter = 0; iter < TREM_MAX_ITER; ++iter) {
            double v_d0, v_d1, v_d2, v_d3;
            if (!be) {
                v_d0 = p[0] + kk[0][0] * i_nl[0] + kk[0][1] * i_nl[1] + kk[0][2] * i_nl[2] + kk[0][3] * i_nl[3];
                v_d1 = p[1] + kk[1][0] * i_nl[0] + kk[1][1] * i_nl[1] + kk[1][2] * i_nl[2];
                v_d2 = p[2] + kk[2][0] * i_nl[0] + kk[2][1] * i_nl[1] + kk[2][3] * i_nl[3];
                v_d3 = p[3] + kk[3][0] * i_nl[0] + kk[3][1] * i_nl[1] + kk[3][2] * i_nl[2] + kk[3][3] * i_nl[3];
            } else {
                v_d0 = p[0] + kk[0][0] * i_nl[0] + kk[0][1] * i_nl[1] + kk[0][2] * i_nl[2] + kk[0][3] * i_nl[3];
                v_d1 = p[1] + kk[1][0] * i_nl[0] + kk[1][1] * i_nl[1] + kk[1][2] * i_nl[2] + kk[1][3] * i_nl[3];
                v_d2 = p[2] + kk[2][0] * i_nl[0] + kk[2][1] * i_nl[1] + kk[2][2] * i_nl[2] + kk[2][3] * i_nl[3];
                v_d3 = p[3] + kk[3][0] * i_nl[0] + kk[3][1] * i_nl[1] + kk[3][2] * i_nl[2] + kk[3][3] * i_nl[3];
            }
            const BjtEval d0 = bjt_evaluate_em(v_d0, v_d1, TREM_DEVICE_0_IS, TREM_DEVICE_0_VT, TREM_DEVICE_0_NF, TREM_DEVICE_0_NR,
                                               TREM_DEVICE_0_BETA_F, TREM_DEVICE_0_BETA_R);
            const BjtEval d1 = bjt_evaluate_em(v_d2, v_d3, TREM_DEVICE_1_IS, TREM_DEVICE_1_VT, TREM_DEVICE_1_NF, TREM_DEVICE_1_NR,
                                               TREM_DEVICE_1_BETA_F, TREM_DEVICE_1_BETA_R);
            const double jd00 = d0.jac[0], jd01 = d0.jac[1], jd10 = d0.jac[2], jd11 = d0.jac[3];
            const double jd22 = d1.jac[0], jd23 = d1.jac[1], jd32 = d1.jac[2], jd33 = d1.jac[3];
            const double f0 = i_nl[0] - d0.ic, f1 = i_nl[1] - d0.ib, f2 = i_nl[2] - d1.ic, f3 = i_nl[3] - d1.ib;
            double a[4][4] = {
                {1.0 - jd00 * kk[0][0] - jd01 * kk[1][0], 0.0 - jd00 * kk[0][1] - jd01 * kk[1][1],
                 0.0 - jd00 * kk[0][2] - jd01 * kk[1][2], 0.0 - jd00 * kk[0][3] - jd01 * kk[1][3]},
                {0.0 - jd10 * kk[0][0] - jd11 * kk[1][0], 1.0 - jd10 * kk[0][1] - jd11 * kk[1][1],
                 0.0 - jd10 * kk[0][2] - jd11 * kk[1][2], 0.0 - jd10 * kk[0][3] - jd11 * kk[1][3]},
                {0.0 - jd22 * kk[2][0] - jd23 * kk[3][0], 0.0 - jd22 * kk[2][1] - jd23 * kk[3][1],
                 1.0 - jd22 * kk[2][2] - jd23 * kk[3][2], 0.0 - jd22 * kk[2][3] - jd23 * kk[3][3]},
                {0.0 - jd32 * kk[2][0] - jd33 * kk[3][0], 0.0 - jd32 * kk[2][1] - jd33 * kk[3][1],
                 0.0 - jd32 * kk[2][2] - jd33 * kk[3][2], 1.0 - jd32 * kk[2][3] - jd33 * kk[3][3]},
            };
            double b[4] = {f0, f1, f2, f3};
            bool singular = false;
            trem_stats()[0] += 1;
            for (int col = 0; col < 4; ++col) {
                int max_row = col;
                double max_val = std::fabs(a[col][col]);
                for (int row = col + 1; row < 4; ++row)
                    if (std::fabs(a[row][col]) > max_val) { max_val = std::fabs(a[row][col]); max_row = row; }
                if (max_val < 1e-15) { singular = true; break; }
                if (max_row != col) {
                    trem_stats()[1 + col] += 1;
                    for (int j = 0; j < 4; ++j) std::swap(a[col][j], a[max_row][j]);
                    std::swap(b[col], b[max_row]);
                }
                const double pivot = a[col][col];
                for (int row = col + 1; row < 4; ++row) {
                    const double factor = a[row][col] / pivot;
                    for (int j = col + 1; j < 4; ++j) a[row][j] -= factor * a[col][j];
                    b[row] -= factor * b[col];
                }
            }
            if (!singular) {
                for (int i = 3; i >= 0; --i) {
                    double sum = b[i];
                    for (int j = i + 1; j < 4; ++j) sum -= a[i][j] * b[j];
                    if (std::fabs(a[i][i]) < 1e-15) { singular = true; break; }
                    b[i] = sum / a[i][i];
                }
            }
            const double vd[4] = {v_d0, v_d1, v_d2, v_d3};
            const double vts[4] = {TREM_DEVICE_0_VT, TREM_DEVICE_0_VT, TREM_DEVICE_1_VT, TREM_DEVICE_1_VT};
            const double vcr[4] = {TREM_DEVICE_0_VCRIT, TREM_DEVICE_0_VCRIT, TREM_DEVICE_1_VCRIT, TREM_DEVICE_1_VCRIT};
            if (!singular && !be) {
                // gen_tremolo.rs:2547-2713
                const double delta[4] = {b[0], b[1], b[2], b[3]};
                double i_trial[4], dv_trial[4], v_lim[4];
                for (int q = 0; q < 4; ++q) i_trial[q] = i_nl[q] - delta[q];
                for (int q = 0; q < 4; ++q) {
                    const double v_trial = p[q] + kk[q][0] * i_trial[0] + kk[q][1] * i_trial[1] + kk[q][2] * i_trial[2] + kk[q][3] * i_trial[3];
                    dv_trial[q] = v_trial - vd[q];
                    v_lim[q] = (std::fabs(dv_trial[q]) > 1e-4) ? pnjlim(v_trial, vd[q], vts[q], vcr[q]) : v_trial;
                }
                bool any_limited = false;
                double global_alpha = 1.0;
                for (int q = 0; q < 4; ++q) {
                    const double dv_lim = v_lim[q] - vd[q];
                    if (std::fabs(dv_trial[q]) > 1e-15) {
                        const double r = (dv_trial[q] * dv_lim < 0.0) ? 0.0 : rclamp(dv_lim / dv_trial[q], 0.0, 1.0);
                        if (r < global_alpha) { global_alpha = r; any_limited = true; }
                    }
                }
                {
                    const double max_dv = std::fmax(std::fmax(std::fmax(std::fabs(dv_trial[0] * global_alpha), std::fabs(dv_trial[1] * global_alpha)),
                                                              std::fabs(dv_trial[2] * global_alpha)), std::fabs(dv_trial[3] * global_alpha));
                    if (max_dv > 3.5) { global_alpha *= std::fmax(3.5 / max_dv, 0.1); any_limited = true; }
                }
                for (int q = 0; q < 4; ++q) i_nl[q] -= global_alpha * delta[q];
                if (!any_limited) {
                    bool conv = true;
                    for (int q = 0; q < 4; ++q) {
                        const double dv = dv_trial[q] * global_alpha;
                        const double thr = 1e-3 * std::fmax(std::fabs(vd[q]), std::fabs(vd[q] + dv)) + 1e-6;
                        if (std::fabs(dv) > thr) conv = false;
                    }
                    if (conv) { iters_out = (uint32_t)iter; return true; }
                }
            } else if (!singular && be) {
                // gen_tremolo.rs:2932-3056
                const double delta[4] = {b[0], b[1], b[2], b[3]};
                double dv[4], alpha[4] = {1.0, 1.0, 1.0, 1.0};
                for (int q = 0; q < 4; ++q)
                    dv[q] = -(kk[q][0] * delta[0] + kk[q][1] * delta[1] + kk[q][2] * delta[2] + kk[q][3] * delta[3]);
                bool any_limited = false;
                for (int q = 0; q < 4; ++q) {
                    if (std::fabs(dv[q]) > 1e-4) {
                        const double vl = pnjlim(vd[q] + dv[q], vd[q], vts[q], vcr[q]);
                        const double ratio = std::fmax((vl - vd[q]) / dv[q], 0.01);
                        if (ratio < alpha[q]) { alpha[q] = ratio; if (ratio < 1.0) any_limited = true; }
                    }
                }
                { const double da = std::fmin(alpha[0], alpha[1]); alpha[0] = da; alpha[1] = da; }
                { const double da = std::fmin(alpha[2], alpha[3]); alpha[2] = da; alpha[3] = da; }
                const double max_dv = std::fmax(std::fmax(std::fmax(std::fabs(dv[0] * alpha[0]), std::fabs(dv[1] * alpha[1])),
                                                          std::fabs(dv[2] * alpha[2])), std::fabs(dv[3] * alpha[3]));
                if (max_dv > 3.5) {
                    const double factor = std::fmax(3.5 / max_dv, 0.1);
                    for (int q = 0; q < 4; ++q) alpha[q] *= factor;
                }
                for (int q = 0; q < 4; ++q) i_nl[q] -= alpha[q] * delta[q];
                if (!any_limited) {
                    bool conv = true;
                    for (int q = 0; q < 4; ++q) {
                        const double step = dv[q] * alpha[q];
                        const double v_new = vd[q] + step;
                        const double thr = 1e-3 * std::fmax(std::fabs(vd[q]), std::fabs(v_new)) + 1e-6;
                        if (std::fabs(step) > thr) conv = false;
                    }
                    if (conv) { iters_out = (uint32_t)iter; return true; }
                }
            } else {
                const double f[4] = {f0, f1, f2, f3};
                for (int q = 0; q < 4; ++q) {
                    const double cl = be ? 0.01 : std::fmax(std::fabs(i_nl[q]) * 0.1, 0.01);
                    i_nl[q] -= rclamp(f[q] * 0.5, -cl, cl);
                }
            }
        }
        return false;
    }

    // gen_tremolo.rs:2353-3116.  Returns v[OUT]=v[0].
    double process_sample(double input_in) {
        const double input = std::isfinite(input_in) ? rclamp(input_in, -100.0, 100.0) : 0.0;
        for (int i = 0; i < TN; ++i) v_prev[i] = v_prev[i] + 1e-25 - 1e-25;
        for (int i = 0; i < TM; ++i) i_nl_prev[i] = i_nl_prev[i] + 1e-25 - 1e-25;

        double rhs[TN];
        for (int i = 0; i < TN; ++i) rhs[i] = TREM_RHS_CONST[i];
        rhs[0] += a_neg[0][0] * v_prev[0];
        rhs[0] += a_neg[0][1] * v_prev[1];
        rhs[0] += a_neg[0][3] * v_prev[3];
        rhs[0] += a_neg[0][5] * v_prev[5];
        rhs[1] += a_neg[1][0] * v_prev[0];
        rhs[1] += a_neg[1][1] * v_prev[1];
        rhs[1] += a_neg[1][2] * v_prev[2];
        rhs[2] += a_neg[2][1] * v_prev[1];
        rhs[2] += a_neg[2][2] * v_prev[2];
        rhs[2] += a_neg[2][3] * v_prev[3];
        rhs[3] += a_neg[3][0] * v_prev[0];
        rhs[3] += a_neg[3][2] * v_prev[2];
        rhs[3] += a_neg[3][3] * v_prev[3];
        rhs[4] += a_neg[4][4] * v_prev[4];
        rhs[5] += a_neg[5][0] * v_prev[0];
        rhs[5] += a_neg[5][5] * v_prev[5];
        rhs[5] += a_neg[5][6] * v_prev[6];
        rhs[0] += TREM_N_I[0][0] * i_nl_prev[0];
        rhs[0] += TREM_N_I[0][2] * i_nl_prev[2];
        rhs[2] += TREM_N_I[2][1] * i_nl_prev[1];
        rhs[4] += TREM_N_I[4][0] * i_nl_prev[0];
        rhs[4] += TREM_N_I[4][1] * i_nl_prev[1];
        rhs[4] += TREM_N_I[4][3] * i_nl_prev[3];

        const double input_conductance = 1.0 / TREM_INPUT_RESISTANCE;
        rhs[0] += (input + input_prev) * input_conductance;
        input_prev = input;

        double v_pred[TN];
        for (int i = 0; i < TN; ++i) {
            double sum = 0.0;
            for (int j = 0; j < TN; ++j) sum += s[i][j] * rhs[j];
            v_pred[i] = sum;
        }
        double p[TM];
        p[0] = TREM_N_V[0][2] * v_pred[2] + TREM_N_V[0][4] * v_pred[4];
        p[1] = TREM_N_V[1][0] * v_pred[0] + TREM_N_V[1][2] * v_pred[2];
        p[2] = TREM_N_V[2][4] * v_pred[4];
        p[3] = TREM_N_V[3][0] * v_pred[0] + TREM_N_V[3][4] * v_pred[4];

        double i_nl[TM];
        for (int i = 0; i < TM; ++i) i_nl[i] = 2.0 * i_nl_prev[i] - i_nl_prev_prev[i];
        last_nr_iterations = TREM_MAX_ITER;
        uint32_t it = 0;
        if (nr_solve(p, k, i_nl, false, it)) last_nr_iterations = it;

        double v[TN];
        for (int i = 0; i < TN; ++i) {
            v[i] = v_pred[i];
            for (int j = 0; j < TM; ++j) v[i] += s_ni[i][j] * i_nl[j];
        }
        const bool converged = last_nr_iterations < (uint32_t)TREM_MAX_ITER;
        if (!converged) {
            diag_nr_max_iter_count += 1;
            diag_be_fallback_count += 1;
            double rhs_be[TN];
            for (int i = 0; i < TN; ++i) {
                double sum = TREM_RHS_CONST_BE[i];
                for (int j = 0; j < TN; ++j) sum += a_neg_be[i][j] * v_prev[j];
                for (int j = 0; j < TM; ++j) sum += TREM_N_I[i][j] * i_nl_prev[j];
                rhs_be[i] = sum;
            }
            rhs_be[0] += input * input_conductance;
            double v_pred_be[TN];
            for (int i = 0; i < TN; ++i) {
                double sum = 0.0;
                for (int j = 0; j < TN; ++j) sum += s_be[i][j] * rhs_be[j];
                v_pred_be[i] = sum;
            }
            double p_be[TM];
            for (int i = 0; i < TM; ++i) {
                double sum = 0.0;
                for (int j = 0; j < TN; ++j) sum += TREM_N_V[i][j] * v_pred_be[j];
                p_be[i] = sum;
            }
            for (int i = 0; i < TM; ++i) i_nl[i] = 2.0 * i_nl_prev[i] - i_nl_prev_prev[i];
            if (nr_solve(p_be, k_be, i_nl, true, it)) last_nr_iterations = it;
            for (int i = 0; i < TN; ++i) {
                v[i] = v_pred_be[i];
                for (int j = 0; j < TM; ++j) v[i] += s_ni_be[i][j] * i_nl[j];
            }
        }
        bool finite = true;
        for (int i = 0; i < TN; ++i) finite = finite && std::isfinite(v[i]);
        if (!finite) {
            for (int i = 0; i < TN; ++i) v_prev[i] = TREM_DC_OP[i];
            for (int i = 0; i < TM; ++i) { i_nl_prev[i] = TREM_DC_NL_I[i]; i_nl_prev_prev[i] = TREM_DC_NL_I[i]; }
            input_prev = 0.0;
            diag_nan_reset_count += 1;
            return 4.26480458363572357e0;
        }
        for (int i = 0; i < TN; ++i) v_prev[i] = v[i];
        for (int i = 0; i < TM; ++i) { i_nl_prev_prev[i] = i_nl_prev[i]; i_nl_prev[i] = i_nl[i]; }
        return v[0];
    }
};

// tremolo.rs:16-216
struct Tremolo {
    TremCircuit osc;
    double sample_rate = 0, depth = 0, r_ldr = 0, ldr_envelope = 0, ldr_attack = 0, ldr_release = 0;
    double r_ldr_max = 1000000.0, gamma = 0.9, ln_r_max = 0, ln_min_minus_max = 0;
    // `--features legacy-tremolo` (tremolo.rs:8, 53-57, 80-90, 170-178, 195-198): behavioural sine LFO instead of the Twin-T circuit.
    // A cargo feature in the reference, a construction-time kind here (set before init()).
    int kind = 0;                      // 0 = Twin-T circuit (default), 1 = legacy LFO
    int r_ulp = 0;                     // test instrumentation: the CdS law's result moved by this many doubles (what another libm's pow / exp does to it)
    double phase = 0, phase_inc = 0;

    void settle_osc() {  // tremolo.rs:92-102 / 215-222
        if (kind == 1) { phase = 0.0; return; }                  // legacy: new() and reset() both start the LFO at phase 0
        osc.init_default();
        if (std::fabs(sample_rate - TREM_SAMPLE_RATE) > 0.5) osc.set_sample_rate(sample_rate);
        const size_t n = (size_t)as_u64(sample_rate * 2.0);
        for (size_t i = 0; i < n; ++i) osc.process_sample(0.0);
    }
    void init(double depth_, double sr) {
        sample_rate = sr;
        phase_inc = 2.0 * 3.14159265358979323846264338327950288 * 5.63 / sr;      // tremolo.rs:86, LEGACY_RATE_HZ :76
        settle_osc();
        depth = depth_;
        r_ldr = 1000000.0;
        ldr_envelope = 0.0;
        ldr_attack = std::exp(-1.0 / (0.0025 * sr));
        ldr_release = std::exp(-1.0 / (0.035 * sr));
        r_ldr_max = 1000000.0;
        gamma = 0.9;
        ln_r_max = std::log(1000000.0);
        ln_min_minus_max = std::log(9000.0) - std::log(1000000.0);
    }
    void set_depth(double d) { depth = rclamp(d, 0.0, 1.0); }
    double shunt_impedance() const {  // tremolo.rs:152-167
        const double r_upper = 50000.0 * (1.0 - depth);
        const double r_lower = 50000.0 * depth;
        const double top = r_upper > 0.0 ? r_upper * 18000.0 / (r_upper + 18000.0) : 0.0;
        const double branch = 680.0 + r_ldr;
        const double low = r_lower > 0.0 ? r_lower * branch / (r_lower + branch) : 0.0;
        return top + low;
    }
    double oscillator_drive() {  // tremolo.rs:170-186
        if (kind == 1) {
            const double lfo = std::sin(phase);
            phase += phase_inc;
            if (phase >= 2.0 * 3.14159265358979323846264338327950288) phase -= 2.0 * 3.14159265358979323846264338327950288;
            return lfo > 0.0 ? lfo : 0.0;          // f64::max(lfo, 0.0): half-wave rectify (NaN -> 0.0, like max)
        }
        const double v_out = osc.process_sample(0.0);
        return rclamp((10.95 - v_out) / (10.95 - 0.70), 0.0, 1.0);
    }
    double process() {  // tremolo.rs:121-146
        const double led_drive = oscillator_drive();
        const double coeff = led_drive > ldr_envelope ? ldr_attack : ldr_release;
        ldr_envelope = led_drive + coeff * (ldr_envelope - led_drive);
        const double drive = rclamp(ldr_envelope, 0.0, 1.0);
        if (drive < 1e-6) r_ldr = r_ldr_max;
        else {
            const double log_r = ln_r_max + ln_min_minus_max * std::pow(drive, gamma);
            r_ldr = std::exp(log_r);
            for (int k = 0; k < r_ulp; ++k) r_ldr = std::nextafter(r_ldr, 1e300);
            for (int k = 0; k > r_ulp; --k) r_ldr = std::nextafter(r_ldr, 0.0);
        }
        return shunt_impedance();
    }
    void reset() {  // tremolo.rs:199-227
        settle_osc();
        ldr_envelope = 0.0;
        r_ldr = r_ldr_max;
    }
};

}  // namespace owo
