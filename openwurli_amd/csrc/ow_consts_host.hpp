// openwurli-hip: host-side construction of the pool constants (OwConsts).
// Init-time only (pool creation / set_sample_rate); nothing here runs per sample.
//
// Mirrors, for one sample rate:
//   gen_tremolo.rs:2111-2342   CircuitState::set_sample_rate / rebuild_matrices / invert_n
//   dk_preamp_legacy.rs:269-366 DkPreamp::new matrix construction (S, A_neg, K, Sherman-Morrison vectors)
//   reed.rs:118-122, pickup.rs:103-106, hammer.rs:126-130, tremolo.rs:104-112, speaker.rs:74
#pragma once
#include <cmath>
#include <cstring>
#include <stdexcept>
#include "ow_types.h"
#include "../../data/ow_gen_data.h"

namespace owhip {

// Rust `as u32`: truncate toward zero, saturating, NaN -> 0.
static inline uint32_t sat_u32(double x) {
    if (!(x == x) || x <= 0.0) return 0u;
    if (x >= 4294967295.0) return 4294967295u;
    return (uint32_t)x;
}

// LU with partial pivoting, then one forward/back substitution per unit column.  The pivot
// search, elimination order and the "start at the permuted unit row" shortcut follow
// gen_tremolo.rs:2273-2342 so the rebuilt S matches the reference's rounding.
template <int NN>
static bool lu_invert(const double* a, double* inv) {
    double lu[NN * NN];
    int perm[NN];
    std::memcpy(lu, a, sizeof lu);
    for (int i = 0; i < NN; ++i) perm[i] = i;
    for (int k = 0; k < NN; ++k) {
        int piv = k;
        double best = std::fabs(lu[k * NN + k]);
        for (int i = k + 1; i < NN; ++i) {
            const double v = std::fabs(lu[i * NN + k]);
            if (v > best) { best = v; piv = i; }
        }
        if (best < 1e-30) return false;
        if (piv != k) {
            for (int j = 0; j < NN; ++j) { const double t = lu[k * NN + j]; lu[k * NN + j] = lu[piv * NN + j]; lu[piv * NN + j] = t; }
            const int t = perm[k]; perm[k] = perm[piv]; perm[piv] = t;
        }
        const double d = lu[k * NN + k];
        for (int i = k + 1; i < NN; ++i) {
            const double m = lu[i * NN + k] / d;
            lu[i * NN + k] = m;
            for (int j = k + 1; j < NN; ++j) lu[i * NN + j] -= m * lu[k * NN + j];
        }
    }
    for (int col = 0; col < NN; ++col) {
        double b[NN];
        for (int i = 0; i < NN; ++i) b[i] = 0.0;
        int start = NN;
        for (int i = 0; i < NN; ++i)
            if (perm[i] == col) { b[i] = 1.0; start = i; break; }
        for (int i = start + 1; i < NN; ++i) {
            double acc = b[i];
            for (int j = start; j < i; ++j) acc -= lu[i * NN + j] * b[j];
            b[i] = acc;
        }
        for (int i = NN - 1; i >= 0; --i) {
            double acc = b[i];
            for (int j = i + 1; j < NN; ++j) acc -= lu[i * NN + j] * b[j];
            const double d = lu[i * NN + i];
            if (std::fabs(d) < 1e-30) return false;
            b[i] = acc / d;
        }
        for (int i = 0; i < NN; ++i) inv[i * NN + col] = b[i];
    }
    return true;
}

// K = N_v S N_i and S_NI = S N_i with the loop nest of gen_tremolo.rs:2172-2194.
// Test switch (ow_test_host_matrices): rebuild even at a solver's codegen rate, where the reference copies its baked tables, so that
// the rebuild itself can be compared with those tables (tests/test_oracle_baked_matrices.py).
static thread_local bool g_force_rebuild = false;

static void trem_kernel_mats(const double s[7][7], double k[4][4], double s_ni[7][4]) {
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double acc = 0.0;
            for (int a = 0; a < 7; ++a) {
                double inner = 0.0;
                for (int b = 0; b < 7; ++b) inner += s[a][b] * TREM_N_I[b][j];
                acc += TREM_N_V[i][a] * inner;
            }
            k[i][j] = acc;
        }
    for (int i = 0; i < 7; ++i)
        for (int j = 0; j < 4; ++j) {
            double acc = 0.0;
            for (int a = 0; a < 7; ++a) acc += s[i][a] * TREM_N_I[a][j];
            s_ni[i][j] = acc;
        }
}

static void build_tremolo_consts(OwConsts& c) {
    const double rate = c.os_sr;
    if (std::fabs(rate - TREM_SAMPLE_RATE) < 0.5 && !g_force_rebuild) {  // gen_tremolo.rs:2117-2130 (codegen-rate defaults)
        std::memcpy(c.t_a_neg, TREM_A_NEG_DEFAULT, sizeof c.t_a_neg);
        std::memcpy(c.t_a_neg_be, TREM_A_NEG_BE_DEFAULT, sizeof c.t_a_neg_be);
        std::memcpy(c.t_s, TREM_S_DEFAULT, sizeof c.t_s);
        std::memcpy(c.t_k, TREM_K_DEFAULT, sizeof c.t_k);
        std::memcpy(c.t_s_ni, TREM_S_NI_DEFAULT, sizeof c.t_s_ni);
        std::memcpy(c.t_s_be, TREM_S_BE_DEFAULT, sizeof c.t_s_be);
        std::memcpy(c.t_k_be, TREM_K_BE_DEFAULT, sizeof c.t_k_be);
        std::memcpy(c.t_s_ni_be, TREM_S_NI_BE_DEFAULT, sizeof c.t_s_ni_be);
    } else {
        const double alpha = 2.0 * rate, alpha_be = rate;
        double a[7][7], a_be[7][7];
        for (int i = 0; i < 7; ++i)
            for (int j = 0; j < 7; ++j) {
                a[i][j] = TREM_G[i][j] + alpha * TREM_C[i][j];
                c.t_a_neg[i][j] = alpha * TREM_C[i][j] - TREM_G[i][j];
                a_be[i][j] = TREM_G[i][j] + alpha_be * TREM_C[i][j];
                c.t_a_neg_be[i][j] = alpha_be * TREM_C[i][j];
            }
        for (int j = 0; j < 7; ++j) { c.t_a_neg[6][j] = 0.0; c.t_a_neg_be[6][j] = 0.0; }  // voltage-source row
        // a failed inversion keeps the codegen-rate matrices (gen_tremolo.rs:2170 `if let Some`)
        std::memcpy(c.t_s, TREM_S_DEFAULT, sizeof c.t_s);
        std::memcpy(c.t_k, TREM_K_DEFAULT, sizeof c.t_k);
        std::memcpy(c.t_s_ni, TREM_S_NI_DEFAULT, sizeof c.t_s_ni);
        std::memcpy(c.t_s_be, TREM_S_BE_DEFAULT, sizeof c.t_s_be);
        std::memcpy(c.t_k_be, TREM_K_BE_DEFAULT, sizeof c.t_k_be);
        std::memcpy(c.t_s_ni_be, TREM_S_NI_BE_DEFAULT, sizeof c.t_s_ni_be);
        double inv[7][7];
        if (lu_invert<7>(&a[0][0], &inv[0][0])) {
            std::memcpy(c.t_s, inv, sizeof inv);
            trem_kernel_mats(c.t_s, c.t_k, c.t_s_ni);
        }
        if (lu_invert<7>(&a_be[0][0], &inv[0][0])) {
            std::memcpy(c.t_s_be, inv, sizeof inv);
            trem_kernel_mats(c.t_s_be, c.t_k_be, c.t_s_ni_be);
        }
    }
    c.ldr_attack = std::exp(-1.0 / (0.0025 * rate));
    c.ldr_release = std::exp(-1.0 / (0.035 * rate));
    c.ln_r_max = std::log(1000000.0);
    c.ln_min_minus_max = std::log(9000.0) - std::log(1000000.0);
    c.lfo_phase_inc = 2.0 * 3.14159265358979323846264338327950288 * 5.63 / rate;      // tremolo.rs:86
}

// Gauss-Jordan on [A | I] with partial pivoting (dk_preamp_legacy.rs:122-168).
static void gauss_jordan8(const double m[8][8], double inv[8][8]) {
    double w[8][16];
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) { w[i][j] = m[i][j]; w[i][8 + j] = (i == j) ? 1.0 : 0.0; }
    for (int col = 0; col < 8; ++col) {
        int piv = col;
        double best = std::fabs(w[col][col]);
        for (int r = col + 1; r < 8; ++r)
            if (std::fabs(w[r][col]) > best) { best = std::fabs(w[r][col]); piv = r; }
        if (!(best > 1e-30)) throw std::runtime_error("openwurli-hip: singular preamp matrix");
        if (piv != col)
            for (int j = 0; j < 16; ++j) { const double t = w[col][j]; w[col][j] = w[piv][j]; w[piv][j] = t; }
        const double d = w[col][col];
        for (int j = 0; j < 16; ++j) w[col][j] /= d;
        for (int r = 0; r < 8; ++r) {
            if (r == col) continue;
            const double f = w[r][col];
            for (int j = 0; j < 16; ++j) w[r][j] -= f * w[col][j];
        }
    }
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) inv[i][j] = w[i][8 + j];
}

static void build_preamp_consts(OwConsts& c) {
    enum { BASE1 = 0, EMIT1, COLL1, EMIT2, EMIT2B, COLL2, OUT, FB };
    const double VCC = 15.0, R1 = 22000.0, R2 = 2000000.0, R3 = 470000.0, RE1 = 33000.0, RC1 = 150000.0, RE2A = 270.0, RE2B = 820.0,
                 RC2 = 1800.0, R9 = 6800.0, R10 = 56000.0;
    const double CIN = 0.022e-6, C3 = 100.0e-12, C4 = 100.0e-12, CE1 = 4.7e-6, CE2 = 22.0e-6;
    const double sr = c.os_sr;
    const double t = 1.0 / sr, two_over_t = 2.0 / t;
    const double alpha_cin = 2.0 * R1 * CIN * sr;
    c.p_g_cin = (2.0 * CIN * sr) / (1.0 + alpha_cin);
    c.p_c_cin = (1.0 - alpha_cin) / (1.0 + alpha_cin);
    c.p_gc_1pc = c.p_g_cin * (1.0 + c.p_c_cin);

    double g[8][8] = {}, cap[8][8] = {}, w[8] = {};
    auto res = [&](int i, int j, double r) { const double y = 1.0 / r; g[i][i] += y; g[j][j] += y; g[i][j] -= y; g[j][i] -= y; };
    auto cp = [&](int i, int j, double v) { cap[i][i] += v; cap[j][j] += v; cap[i][j] -= v; cap[j][i] -= v; };
    g[BASE1][BASE1] += 1.0 / R2;  w[BASE1] += VCC / R2;
    g[BASE1][BASE1] += 1.0 / R3;
    g[EMIT1][EMIT1] += 1.0 / RE1;
    g[COLL1][COLL1] += 1.0 / RC1; w[COLL1] += VCC / RC1;
    res(EMIT2, EMIT2B, RE2A);
    g[EMIT2B][EMIT2B] += 1.0 / RE2B;
    g[COLL2][COLL2] += 1.0 / RC2; w[COLL2] += VCC / RC2;
    res(COLL2, OUT, R9);
    res(OUT, FB, R10);
    std::memcpy(c.p_g_dc_base, g, sizeof g);
    g[BASE1][BASE1] += c.p_g_cin;
    cp(COLL1, BASE1, C3);
    cp(COLL2, COLL1, C4);
    cp(EMIT1, FB, CE1);
    cp(EMIT2, EMIT2B, CE2);
    double a[8][8];
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) {
            const double tc = two_over_t * cap[i][j];
            a[i][j] = tc + g[i][j];
            c.p_a_neg[i][j] = tc - g[i][j];
        }
    for (int i = 0; i < 8; ++i) c.p_two_w[i] = 2.0 * w[i] + 0.0;      // (never -0.0: dk_step's rhs rows rely on it, see there)
    gauss_jordan8(a, c.p_s);
    const double (*s)[8] = c.p_s;
    c.p_k[0][0] = s[BASE1][EMIT1] - s[BASE1][COLL1] - s[EMIT1][EMIT1] + s[EMIT1][COLL1];
    c.p_k[0][1] = s[BASE1][EMIT2] - s[BASE1][COLL2] - s[EMIT1][EMIT2] + s[EMIT1][COLL2];
    c.p_k[1][0] = s[COLL1][EMIT1] - s[COLL1][COLL1] - s[EMIT2][EMIT1] + s[EMIT2][COLL1];
    c.p_k[1][1] = s[COLL1][EMIT2] - s[COLL1][COLL2] - s[EMIT2][EMIT2] + s[EMIT2][COLL2];
    double row[8];
    for (int i = 0; i < 8; ++i) { c.p_s_fb_col[i] = s[i][FB]; row[i] = s[FB][i]; }
    c.p_s_fb_fb = s[FB][FB];
    c.p_nv_sfb[0] = c.p_s_fb_col[BASE1] - c.p_s_fb_col[EMIT1];
    c.p_nv_sfb[1] = c.p_s_fb_col[COLL1] - c.p_s_fb_col[EMIT2];
    c.p_sfb_ni[0] = row[EMIT1] - row[COLL1];
    c.p_sfb_ni[1] = row[EMIT2] - row[COLL2];
    for (int i = 0; i < 8; ++i) { c.p_sni_d1[i] = s[i][EMIT1] - s[i][COLL1]; c.p_sni_d2[i] = s[i][EMIT2] - s[i][COLL2]; }
}

// melange 12-node preamp: S0 = A^-1 at (chain rate, nominal 100 kOhm pot) with the reference's LU (gen_preamp.rs:1990-2062,
// 2117-2219; codegen tables within 0.5 Hz of 48 kHz, :1941-1957), plus the rank-one vectors of the R_ldr entry A[6][6].
static void build_melange_consts(OwConsts& c) {
    const double rate = c.os_sr;
    if (std::fabs(rate - PRE_SAMPLE_RATE) < 0.5 && !g_force_rebuild) {
        std::memcpy(c.m_s0, PRE_S_DEFAULT, sizeof c.m_s0);
        std::memcpy(c.m_aneg0, PRE_A_NEG_DEFAULT, sizeof c.m_aneg0);
        std::memcpy(c.m_k0, PRE_K_DEFAULT, sizeof c.m_k0);
        std::memcpy(c.m_sni0, PRE_S_NI_DEFAULT, sizeof c.m_sni0);
    } else {
        const double alpha = 2.0 * (rate * 1.0);
        double a[12][12];
        for (int i = 0; i < 12; ++i)
            for (int j = 0; j < 12; ++j) {
                a[i][j] = PRE_G[i][j] + alpha * PRE_C[i][j];
                c.m_aneg0[i][j] = (i == 11) ? 0.0 : alpha * PRE_C[i][j] - PRE_G[i][j];
            }
        if (!lu_invert<12>(&a[0][0], &c.m_s0[0][0]))
            for (int i = 0; i < 12; ++i) for (int j = 0; j < 12; ++j) c.m_s0[i][j] = (i == j) ? 1.0 : 0.0;   // identity fallback (:2142-2148)
        for (int i = 0; i < 12; ++i)
            for (int j = 0; j < 3; ++j) {
                double sum = 0.0;
                for (int k = 0; k < 12; ++k) sum += c.m_s0[i][k] * PRE_N_I[j][k];
                c.m_sni0[i][j] = sum;
            }
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double sum = 0.0;
                for (int n = 0; n < 12; ++n) sum += PRE_N_V[i][n] * c.m_sni0[n][j];
                c.m_k0[i][j] = sum;
            }
    }
    {   // R-independent part of invert_n(A) at this rate (gen_preamp.rs:2117-2219 followed step by step through column 5)
        const double alpha = 2.0 * (rate * 1.0);
        double lu[12][12];
        int perm[12];
        for (int i = 0; i < 12; ++i) { perm[i] = i; for (int j = 0; j < 12; ++j) lu[i][j] = PRE_G[i][j] + alpha * PRE_C[i][j]; }
        bool ok = true;
        for (int k = 0; k < 6 && ok; ++k) {
            int max_row = k;
            double max_val = std::fabs(lu[k][k]);
            for (int i = k + 1; i < 12; ++i) {
                const double v = std::fabs(lu[i][k]);
                if (v > max_val) { max_val = v; max_row = i; }
            }
            if (max_val < 1e-30) { ok = false; break; }
            if (max_row != k) {
                for (int j = 0; j < 12; ++j) { const double t = lu[k][j]; lu[k][j] = lu[max_row][j]; lu[max_row][j] = t; }
                const int t = perm[k]; perm[k] = perm[max_row]; perm[max_row] = t;
            }
            if (perm[k] == 6) { ok = false; break; }        // the R-carrying row became a pivot row: its entry would spread
            const double pivot = lu[k][k];
            for (int i = k + 1; i < 12; ++i) {
                const double m = lu[i][k] / pivot;
                lu[i][k] = m;
                for (int j = k + 1; j < 12; ++j) {
                    if (perm[i] == 6 && j == 6) { c.ml_chain_m[k] = m; c.ml_chain_u[k] = lu[k][6]; continue; }   // replayed per sample on the device
                    lu[i][j] -= m * lu[k][j];
                }
            }
        }
        c.ml_ok = 0;
        if (ok) {
            int t6 = -1;
            for (int t = 0; t < 6; ++t) if (perm[6 + t] == 6) t6 = t;
            if (t6 >= 0) {
                c.ml_t6 = t6;
                for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) c.ml_t0[a][b] = lu[6 + a][6 + b];
                for (int i = 0; i < 6; ++i) for (int j = 0; j < 12; ++j) c.ml_utop[i][j] = j >= i ? lu[i][j] : 0.0;
                for (int col = 0; col < 12; ++col) {
                    double b[12];
                    for (int i = 0; i < 12; ++i) b[i] = perm[i] == col ? 1.0 : 0.0;
                    for (int i = 1; i < 6; ++i) {
                        double sum = b[i];
                        for (int j = 0; j < i; ++j) sum -= lu[i][j] * b[j];
                        b[i] = sum;
                    }
                    for (int i = 0; i < 6; ++i) c.ml_btop[col][i] = b[i];
                    for (int t = 0; t < 6; ++t) {
                        double sum = b[6 + t];
                        for (int j = 0; j < 6; ++j) sum -= lu[6 + t][j] * b[j];
                        c.ml_part[col][t] = sum;
                    }
                }
                c.ml_ok = 1;
            }
        }
    }
    {   // pattern the column-streamed kernel compiles in (ow_melange_col.h): U rows 0..5, trailing factors, position of the R row
        static const unsigned U_TOP[6] = {0x003u, 0x006u, 0x03Cu, 0x008u, 0x1B0u, 0x1E0u};            // bit j: U[i][j] may be non-zero
        static const unsigned T_PAT[6] = {0x17u, 0x17u, 0x1Fu, 0x1Cu, 0x1Fu, 0x3Fu};                 // bit b: trailing factor [a][b] may be non-zero
        bool ok = c.ml_ok != 0 && c.ml_t6 == 0;
        for (int i = 0; i < 6 && ok; ++i)
            for (int j = i; j < 12; ++j)
                if (!((U_TOP[i] >> j) & 1u) && c.ml_utop[i][j] != 0.0) ok = false;
        for (int i = 0; i < 6; ++i) c.ml_utop_rcp[i] = 1.0 / c.ml_utop[i][i];
        // zeros of the forward-substituted unit columns (MCOL_PART / MCOL_BTOP of ow_melange_col.h)
        static const unsigned PART[12] = {0x3F, 0x3F, 0x3F, 0x20, 0x3F, 0x3F, 0x3F, 0x3E, 0x3C, 0x38, 0x30, 0x3F};
        static const unsigned BTOP[12] = {0x37, 0x36, 0x34, 0x00, 0x30, 0x20, 0x00, 0x00, 0x00, 0x00, 0x00, 0x38};
        for (int col = 0; col < 12 && ok; ++col)
            for (int t = 0; t < 6; ++t) {
                if (!((PART[col] >> t) & 1u) && c.ml_part[col][t] != 0.0) ok = false;
                if (!((BTOP[col] >> t) & 1u) && c.ml_btop[col][t] != 0.0) ok = false;
            }
        const double rs[3] = {1000.0, 9.99999999999999854e4, 1000000.0};
        for (int q = 0; q < 3 && ok; ++q) {          // the trailing elimination at three resistances: nothing outside the pattern
            const double alpha = 2.0 * (rate * 1.0);
            double T[6][6];
            for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) T[a][b] = c.ml_t0[a][b];
            double e = (PRE_G[6][6] + (1.0 / rs[q] - PRE_POT_0_G_NOM)) + alpha * PRE_C[6][6];
            for (int k = 0; k < 6; ++k) e -= c.ml_chain_m[k] * c.ml_chain_u[k];
            T[0][0] = e;
            for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) if (!((T_PAT[a] >> b) & 1u) && T[a][b] != 0.0) ok = false;
            for (int k = 0; k < 6 && ok; ++k) {
                for (int i = k + 1; i < 6; ++i) {
                    const double m = T[i][k] / T[k][k];
                    T[i][k] = m;
                    for (int j = k + 1; j < 6; ++j) T[i][j] -= m * T[k][j];
                }
                for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) if (!((T_PAT[a] >> b) & 1u) && T[a][b] != 0.0) ok = false;
            }
        }
        c.ml_sparse_ok = ok ? 1 : 0;
    }
    for (int i = 0; i < 12; ++i) { c.m_u[i] = c.m_s0[i][6]; c.m_w[i] = c.m_s0[6][i]; }
    c.m_s66 = c.m_s0[6][6];
    c.m_g_nom = PRE_POT_0_G_NOM;
    c.m_noise_scale = std::sqrt(8.0 * 1.380649e-23 * 290.0 * (rate * 1.0));   // noise_fs = sample_rate * OVERSAMPLING_FACTOR
    for (int j = 0; j < 3; ++j) {
        double sum = 0.0;
        for (int k = 0; k < 12; ++k) sum += c.m_w[k] * PRE_N_I[j][k];
        c.m_wn[j] = sum;
        double s2 = 0.0;
        for (int n = 0; n < 12; ++n) s2 += PRE_N_V[j][n] * c.m_u[n];
        c.m_nvu[j] = s2;
    }
}

// Melange 7-BJT power amp at the chain rate `rate` (gen_power_amp.rs:8588-8755: set_sample_rate / rebuild_matrices with the
// backward-Euler companion A = G + C/T, A_neg = C/T, source rows of A_neg zeroed; the codegen tables when the rate is the codegen
// rate) + RailDynamics::set_sample_rate (power_amp.rs:108-113) + the per-device constant quotients of bjt_evaluate.
static void build_pa_consts(OwPaConsts& c, double rate) {
    std::memset(&c, 0, sizeof c);
    c.rate_is_codegen = std::fabs(rate - PA_SAMPLE_RATE) <= 0.5 ? 1 : 0;
    if (std::fabs(rate - PA_SAMPLE_RATE) < 0.5 && !g_force_rebuild) {
        std::memcpy(c.a_neg, PA_A_NEG_DEFAULT, sizeof c.a_neg); std::memcpy(c.a_neg_be, PA_A_NEG_BE_DEFAULT, sizeof c.a_neg_be);
        std::memcpy(c.s, PA_S_DEFAULT, sizeof c.s); std::memcpy(c.k, PA_K_DEFAULT, sizeof c.k); std::memcpy(c.s_ni, PA_S_NI_DEFAULT, sizeof c.s_ni);
        std::memcpy(c.s_be, PA_S_BE_DEFAULT, sizeof c.s_be); std::memcpy(c.k_be, PA_K_BE_DEFAULT, sizeof c.k_be); std::memcpy(c.s_ni_be, PA_S_NI_BE_DEFAULT, sizeof c.s_ni_be);
        c.dc_block_r = PA_DC_BLOCK_R;
    } else {
        const double alpha = rate * 1.0, alpha_be = rate * 1.0;
        static thread_local double a[PA_N][PA_N], a_be[PA_N][PA_N], inv[PA_N][PA_N];
        for (int i = 0; i < PA_N; ++i)
            for (int j = 0; j < PA_N; ++j) {
                a[i][j] = PA_G[i][j] + alpha * PA_C[i][j];
                c.a_neg[i][j] = alpha * PA_C[i][j];
                a_be[i][j] = PA_G[i][j] + alpha_be * PA_C[i][j];
                c.a_neg_be[i][j] = alpha_be * PA_C[i][j];
            }
        for (int i = 18; i < 20; ++i)
            for (int j = 0; j < PA_N; ++j) { c.a_neg[i][j] = 0.0; c.a_neg_be[i][j] = 0.0; }
        auto derive = [](const double S[PA_N][PA_N], double K[PA_M][PA_M], double SNI[PA_N][PA_M]) {
            for (int i = 0; i < PA_M; ++i)
                for (int j = 0; j < PA_M; ++j) {
                    double sum = 0.0;
                    for (int aa = 0; aa < PA_N; ++aa) {
                        double s_ni_aj = 0.0;
                        for (int b = 0; b < PA_N; ++b) s_ni_aj += S[aa][b] * PA_N_I[b][j];
                        sum += PA_N_V[i][aa] * s_ni_aj;
                    }
                    K[i][j] = sum;
                }
            for (int i = 0; i < PA_N; ++i)
                for (int j = 0; j < PA_M; ++j) {
                    double sum = 0.0;
                    for (int aa = 0; aa < PA_N; ++aa) sum += S[i][aa] * PA_N_I[aa][j];
                    SNI[i][j] = sum;
                }
        };
        // a failed inversion keeps the previous (codegen) matrices in the reference; it cannot fail for a positive rate, so say so
        if (!lu_invert<PA_N>(&a[0][0], &inv[0][0])) throw std::runtime_error("power amp: singular system matrix");
        std::memcpy(c.s, inv, sizeof c.s); derive(c.s, c.k, c.s_ni);
        if (!lu_invert<PA_N>(&a_be[0][0], &inv[0][0])) throw std::runtime_error("power amp: singular BE system matrix");
        std::memcpy(c.s_be, inv, sizeof c.s_be); derive(c.s_be, c.k_be, c.s_ni_be);
        c.dc_block_r = 1.0 - 2.0 * 3.14159265358979323846 * 5.0 / rate;
    }
    const double dt = 1.0 / rate;
    c.alpha_attack = 1.0 - std::exp(-dt / 0.008);
    c.alpha_release = 1.0 - std::exp(-dt / 0.015);
    c.alpha_i_avg = 1.0 - std::exp(-dt / 0.030);
    for (int i = 0; i < PA_N; ++i)      // the kernel writes the constant part of the right-hand side as "22.5 V on the two supply rows"
        if (PA_RHS_CONST[i] != (i >= 18 ? 22.5 : 0.0)) throw std::runtime_error("power amp: RHS_CONST is not the two 22.5 V supply rows the kernel assumes");
    for (int d = 0; d < 8; ++d) {
        if (PA_DEV_USE_GP[d] == 0.0) throw std::runtime_error("power amp: a device without Gummel-Poon terms (only the GP branch is built)");
        OwPaConsts::Dev& D = c.dev[d];
        const double is = PA_DEV_IS[d], vt = PA_DEV_VT[d], nf = PA_DEV_NF[d], nr = PA_DEV_NR[d], bf = PA_DEV_BETA_F[d], br = PA_DEV_BETA_R[d];
        const double ise = PA_DEV_ISE[d], ne = PA_DEV_NE[d], isc = PA_DEV_ISC[d], nc = PA_DEV_NC[d];
        D.is = is; D.vt = vt; D.sign = PA_DEV_SIGN[d]; D.vcrit = PA_DEV_VCRIT[d]; D.rb = PA_DEV_RB[d]; D.rc = PA_DEV_RC[d]; D.re = PA_DEV_RE[d];
        D.nf_vt = nf * vt; D.nr_vt = nr * vt; D.ne_vt = ne * vt; D.nc_vt = nc * vt;
        D.var = PA_DEV_VAR[d]; D.vaf = PA_DEV_VAF[d]; D.ikf = PA_DEV_IKF[d]; D.ikr = PA_DEV_IKR[d];
        D.is_bf = is / bf; D.is_br = is / br;
        D.c_dib_fwd = is / (bf * D.nf_vt); D.c_dib_rev = is / (br * D.nr_vt);
        D.ise = ise; D.isc = isc; D.c_leak_be = ise / (ne * vt); D.c_leak_bc = isc / (nc * vt);
        D.c_dq2_be = is / (D.nf_vt * D.ikf); D.c_dq2_bc = is / (D.nr_vt * D.ikr);
        D.c_dicc_be = is / D.nf_vt; D.c_dicc_bc = -is / D.nr_vt;
        D.max_step = 4.0 * vt;
        D.r_nf_vt = 1.0 / D.nf_vt; D.r_nr_vt = 1.0 / D.nr_vt; D.r_ne_vt = 1.0 / D.ne_vt; D.r_nc_vt = 1.0 / D.nc_vt;
        D.r_var = 1.0 / D.var; D.r_vaf = 1.0 / D.vaf; D.r_ikf = 1.0 / D.ikf; D.r_ikr = 1.0 / D.ikr;
    }
}

static void build_consts(OwConsts& c, double sample_rate, int preamp_kind) {
    std::memset(&c, 0, sizeof c);
    c.sr = sample_rate;
    c.oversample = sample_rate < 88200.0 ? 1 : 0;                 // engine.rs:195
    c.os_sr = c.oversample ? sample_rate * 2.0 : sample_rate;
    c.preamp_kind = preamp_kind;
    const double dt = 1.0 / sample_rate;
    c.jitter_revert = std::exp(-dt / 0.020);
    c.jitter_diffusion = 0.0004 * std::sqrt(1.0 - c.jitter_revert * c.jitter_revert);
    c.pickup_beta = dt / (2.0 * (287.0e3 * 240.0e-12));
    c.noise_decay = std::exp(-1.0 / (0.003 * sample_rate));
    c.noise_len = sat_u32(0.015 * sample_rate);
    const uint32_t ramp = sat_u32(sample_rate * 0.005);
    c.ramp_samples = ramp > 1u ? ramp : 1u;
    c.spk_thermal_alpha = 1.0 / (5.0 * sample_rate);
    build_tremolo_consts(c);
    build_preamp_consts(c);
    build_melange_consts(c);
}

}  // namespace owhip
