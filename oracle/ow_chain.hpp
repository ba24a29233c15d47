// ORACLE (test infrastructure, CPU, f64) -- NOT part of the shipped product path.
//
// Restates the shared mono chain of the reference (citations into
// /root/reference/crates/openwurli-dsp/src/):
//   oversampler.rs:17-148        2x polyphase half-band (allpass branches)
//   dk_preamp_legacy.rs:21-690   default 8-node DK preamp (main + shadow, Sherman-Morrison R_ldr)
//   power_amp.rs:168-273         default behavioural power amp (stateless NR)
//   speaker.rs:33-139            Hammerstein speaker + HPF/LPF
#pragma once
#include "ow_voice.hpp"

namespace owo {

// ------------------------------------------------------------- oversampler.rs
static const double OS_BRANCH_A[3] = {0.036681502163648, 0.248030921580110, 0.643184620136480};  // :17-21
static const double OS_BRANCH_B[3] = {0.110377634768680, 0.420399304190880, 0.854640112701920};  // :23-27

struct Oversampler {
    double ua[3], ub[3], da[3], db[3], down_delay;
    Oversampler() { reset(); }
    void reset() {
        for (int i = 0; i < 3; ++i) ua[i] = ub[i] = da[i] = db[i] = 0.0;
        down_delay = 0.0;
    }
    static inline double branch(const double coeff[3], double st[3], double x) {  // :41-45, :63-69
        double y = x;
        for (int i = 0; i < 3; ++i) {
            const double yy = coeff[i] * y + st[i];
            st[i] = y - coeff[i] * yy;
            y = yy;
        }
        return y;
    }
    void upsample_2x(const double* in, size_t n, double* out) {  // :108-121
        for (size_t i = 0; i < n; ++i) {
            out[2 * i] = branch(OS_BRANCH_A, ua, in[i]);
            out[2 * i + 1] = branch(OS_BRANCH_B, ub, in[i]);
        }
    }
    void downsample_2x(const double* in, double* out, size_t n) {  // :126-139
        for (size_t i = 0; i < n; ++i) {
            const double a = branch(OS_BRANCH_A, da, in[2 * i]);
            const double b = branch(OS_BRANCH_B, db, in[2 * i + 1]);
            out[i] = (a + down_delay) * 0.5;
            down_delay = b;
        }
    }
};

// -------------------------------------------------------- dk_preamp_legacy.rs
namespace dk {
constexpr double VCC = 15.0, R1 = 22000.0, R2 = 2000000.0, R3 = 470000.0, RE1 = 33000.0, RC1 = 150000.0, RE2A = 270.0,
                 RE2B = 820.0, RC2 = 1800.0, R9 = 6800.0, R10 = 56000.0;               // :21-33
constexpr double CIN = 0.022e-6, C3 = 100.0e-12, C4 = 100.0e-12, CE1 = 4.7e-6, CE2 = 22.0e-6;  // :36-40
constexpr double IS = 3.03e-14, VT = 0.026, IS_OVER_VT = IS / VT, VBE_MAX = 0.85;       // :43-49
enum { BASE1 = 0, EMIT1, COLL1, EMIT2, EMIT2B, COLL2, OUT, FB, N = 8 };                 // :52-61

typedef double Mat8[N][N];
typedef double Vec8[N];

inline void mat_vec_mul(const Mat8 a, const Vec8 x, Vec8 y) {  // :76-86
    for (int i = 0; i < N; ++i) {
        double sum = 0.0;
        for (int j = 0; j < N; ++j) sum += a[i][j] * x[j];
        y[i] = sum;
    }
}
inline void mat_inverse(const Mat8 m, Mat8 inv) {  // :122-168 (Gauss-Jordan, partial pivoting)
    double aug[N][2 * N];
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) { aug[i][j] = m[i][j]; aug[i][N + j] = (i == j) ? 1.0 : 0.0; }
    for (int col = 0; col < N; ++col) {
        double max_val = std::fabs(aug[col][col]);
        int max_row = col;
        for (int row = col + 1; row < N; ++row)
            if (std::fabs(aug[row][col]) > max_val) { max_val = std::fabs(aug[row][col]); max_row = row; }
        if (max_row != col)
            for (int j = 0; j < 2 * N; ++j) std::swap(aug[col][j], aug[max_row][j]);
        const double pivot = aug[col][col];
        for (int j = 0; j < 2 * N; ++j) aug[col][j] /= pivot;
        for (int row = 0; row < N; ++row) {
            if (row != col) {
                const double factor = aug[row][col];
                for (int j = 0; j < 2 * N; ++j) aug[row][j] -= factor * aug[col][j];
            }
        }
    }
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) inv[i][j] = aug[i][N + j];
}
inline void stamp_resistor(Mat8 g, int i, int j, double r) {  // :641-647
    const double c = 1.0 / r;
    g[i][i] += c; g[j][j] += c; g[i][j] -= c; g[j][i] -= c;
}
inline void stamp_capacitor(Mat8 c, int i, int j, double cap) {  // :649-654
    c[i][i] += cap; c[j][j] += cap; c[i][j] -= cap; c[j][i] -= cap;
}
// OW_ORACLE_EXP_PERTURB builds a sensitivity variant whose BJT exp() is off by one ulp: it measures how far the
// reference algorithm's own output moves under a libm that differs in the last bit (tests/test_oracle_sensitivity.py).
#ifdef OW_ORACLE_EXP_PERTURB
inline double bjt_exp(double x) { return std::exp(x) * (1.0 + 2.2e-16); }
#else
inline double bjt_exp(double x) { return std::exp(x); }
#endif
inline double bjt_ic(double vbe) {  // :663-666
    const double v = rclamp(vbe, -1.0, VBE_MAX);
    return IS * (bjt_exp(v / VT) - 1.0);
}
inline void bjt_ic_gm(double vbe, double& ic, double& gm) {  // :686-690
    const double v = rclamp(vbe, -1.0, VBE_MAX);
    const double e = bjt_exp(v / VT);
    ic = IS * (e - 1.0);
    gm = IS_OVER_VT * e;
}
inline void compute_k(const Mat8 s, double k[2][2]) {  // :414-425
    k[0][0] = s[BASE1][EMIT1] - s[BASE1][COLL1] - s[EMIT1][EMIT1] + s[EMIT1][COLL1];
    k[0][1] = s[BASE1][EMIT2] - s[BASE1][COLL2] - s[EMIT1][EMIT2] + s[EMIT1][COLL2];
    k[1][0] = s[COLL1][EMIT1] - s[COLL1][COLL1] - s[EMIT2][EMIT1] + s[EMIT2][COLL1];
    k[1][1] = s[COLL1][EMIT2] - s[COLL1][COLL2] - s[EMIT2][EMIT2] + s[EMIT2][COLL2];
}
}  // namespace dk

struct DkState {  // dk_preamp_legacy.rs:231-251
    double j_cin, cin_rhs_prev, v[8], i_nl[2], v_nl[2];
};

struct DkPreamp {
    dk::Mat8 s_base, a_neg_base, g_dc_base;
    double k[2][2];
    dk::Vec8 two_w, s_fb_col, s_fb_row, v_dc;
    double s_fb_fb, nv_sfb[2], sfb_ni[2];
    double g_cin, c_cin, gc_1pc;
    DkState main, shadow;
    double r_ldr, g_ldr, g_ldr_prev;

    static DkState at_dc(double g_cin, const double v_nl_dc[2], const dk::Vec8 v_dc) {  // :241-250
        DkState s;
        s.j_cin = g_cin * v_dc[dk::BASE1];
        s.cin_rhs_prev = g_cin * v_dc[dk::BASE1];
        for (int i = 0; i < 8; ++i) s.v[i] = v_dc[i];
        s.i_nl[0] = dk::bjt_ic(v_nl_dc[0]); s.i_nl[1] = dk::bjt_ic(v_nl_dc[1]);
        s.v_nl[0] = v_nl_dc[0]; s.v_nl[1] = v_nl_dc[1];
        return s;
    }

    // :369-412
    static void full_dc_solve(const dk::Mat8 g_dc_base, const dk::Vec8 w, double r_ldr, double v_nl[2], dk::Vec8 v_dc) {
        using namespace dk;
        Mat8 g_full, s_dc;
        for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) g_full[i][j] = g_dc_base[i][j];
        g_full[FB][FB] += 1.0 / r_ldr;
        mat_inverse(g_full, s_dc);
        double k_dc[2][2];
        compute_k(s_dc, k_dc);
        Vec8 sv;
        mat_vec_mul(s_dc, w, sv);
        const double p_dc[2] = {sv[BASE1] - sv[EMIT1], sv[COLL1] - sv[EMIT2]};
        v_nl[0] = 0.56; v_nl[1] = 0.66;
        for (int iter = 0; iter < 100; ++iter) {
            double ic0, gm0, ic1, gm1;
            bjt_ic_gm(v_nl[0], ic0, gm0);
            bjt_ic_gm(v_nl[1], ic1, gm1);
            const double f0 = v_nl[0] - p_dc[0] - k_dc[0][0] * ic0 - k_dc[0][1] * ic1;
            const double f1 = v_nl[1] - p_dc[1] - k_dc[1][0] * ic0 - k_dc[1][1] * ic1;
            if (std::fabs(f0) < 1e-12 && std::fabs(f1) < 1e-12) break;
            const double j00 = 1.0 - k_dc[0][0] * gm0, j01 = -k_dc[0][1] * gm1;
            const double j10 = -k_dc[1][0] * gm0, j11 = 1.0 - k_dc[1][1] * gm1;
            const double det = j00 * j11 - j01 * j10;
            const double inv_det = 1.0 / det;
            const double dv0 = inv_det * (j11 * f0 - j01 * f1);
            const double dv1 = inv_det * (j00 * f1 - j10 * f0);
            const double max_step = 2.0 * VT;
            v_nl[0] -= rclamp(dv0, -max_step, max_step);
            v_nl[1] -= rclamp(dv1, -max_step, max_step);
        }
        const double ic[2] = {bjt_ic(v_nl[0]), bjt_ic(v_nl[1])};
        Vec8 dc_rhs;
        for (int i = 0; i < N; ++i) dc_rhs[i] = w[i];
        dc_rhs[EMIT1] += ic[0]; dc_rhs[COLL1] -= ic[0];
        dc_rhs[EMIT2] += ic[1]; dc_rhs[COLL2] -= ic[1];
        mat_vec_mul(s_dc, dc_rhs, v_dc);
    }

    // :269-366
    void init(double sample_rate) {
        using namespace dk;
        const double t = 1.0 / sample_rate;
        const double two_over_t = 2.0 / t;
        const double alpha_cin = 2.0 * R1 * CIN * sample_rate;
        g_cin = (2.0 * CIN * sample_rate) / (1.0 + alpha_cin);
        c_cin = (1.0 - alpha_cin) / (1.0 + alpha_cin);
        gc_1pc = g_cin * (1.0 + c_cin);

        Mat8 g_base = {};
        Vec8 w = {};
        g_base[BASE1][BASE1] += 1.0 / R2;
        w[BASE1] += VCC / R2;
        g_base[BASE1][BASE1] += 1.0 / R3;
        g_base[EMIT1][EMIT1] += 1.0 / RE1;
        g_base[COLL1][COLL1] += 1.0 / RC1;
        w[COLL1] += VCC / RC1;
        stamp_resistor(g_base, EMIT2, EMIT2B, RE2A);
        g_base[EMIT2B][EMIT2B] += 1.0 / RE2B;
        g_base[COLL2][COLL2] += 1.0 / RC2;
        w[COLL2] += VCC / RC2;
        stamp_resistor(g_base, COLL2, OUT, R9);
        stamp_resistor(g_base, OUT, FB, R10);
        for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) g_dc_base[i][j] = g_base[i][j];
        g_base[BASE1][BASE1] += g_cin;

        Mat8 c = {};
        stamp_capacitor(c, COLL1, BASE1, C3);
        stamp_capacitor(c, COLL2, COLL1, C4);
        stamp_capacitor(c, EMIT1, FB, CE1);
        stamp_capacitor(c, EMIT2, EMIT2B, CE2);
        Mat8 a_base;
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < N; ++j) {
                const double tc = two_over_t * c[i][j];
                a_base[i][j] = tc + g_base[i][j];
                a_neg_base[i][j] = tc - g_base[i][j];
            }
        for (int i = 0; i < N; ++i) two_w[i] = 2.0 * w[i];
        mat_inverse(a_base, s_base);
        compute_k(s_base, k);
        for (int i = 0; i < N; ++i) { s_fb_col[i] = s_base[i][FB]; s_fb_row[i] = s_base[FB][i]; }
        s_fb_fb = s_base[FB][FB];
        nv_sfb[0] = s_fb_col[BASE1] - s_fb_col[EMIT1];
        nv_sfb[1] = s_fb_col[COLL1] - s_fb_col[EMIT2];
        sfb_ni[0] = s_fb_row[EMIT1] - s_fb_row[COLL1];
        sfb_ni[1] = s_fb_row[EMIT2] - s_fb_row[COLL2];

        const double r_init = 1000000.0;
        double v_nl_dc[2];
        full_dc_solve(g_dc_base, w, r_init, v_nl_dc, v_dc);
        main = at_dc(g_cin, v_nl_dc, v_dc);
        shadow = main;
        r_ldr = r_init;
        g_ldr = 1.0 / r_init;
        g_ldr_prev = 1.0 / r_init;
    }

    // :447-554
    double dk_step(DkState& st, double input) const {
        using namespace dk;
        Vec8 rhs;
        mat_vec_mul(a_neg_base, st.v, rhs);
        rhs[FB] -= g_ldr_prev * st.v[FB];
        const double cin_rhs_now = g_cin * input + st.j_cin;
        rhs[BASE1] += cin_rhs_now + st.cin_rhs_prev;
        rhs[EMIT1] += st.i_nl[0];
        rhs[COLL1] -= st.i_nl[0];
        rhs[EMIT2] += st.i_nl[1];
        rhs[COLL2] -= st.i_nl[1];
        for (int i = 0; i < N; ++i) rhs[i] += two_w[i];
        Vec8 v_pred_base;
        mat_vec_mul(s_base, rhs, v_pred_base);
        const double sm_k = g_ldr / (1.0 + s_fb_fb * g_ldr);
        const double sm_vpred = sm_k * v_pred_base[FB];
        Vec8 v_pred;
        for (int i = 0; i < N; ++i) v_pred[i] = v_pred_base[i] - sm_vpred * s_fb_col[i];
        const double p[2] = {v_pred[BASE1] - v_pred[EMIT1], v_pred[COLL1] - v_pred[EMIT2]};
        const double k00 = k[0][0] - sm_k * nv_sfb[0] * sfb_ni[0];
        const double k01 = k[0][1] - sm_k * nv_sfb[0] * sfb_ni[1];
        const double k10 = k[1][0] - sm_k * nv_sfb[1] * sfb_ni[0];
        const double k11 = k[1][1] - sm_k * nv_sfb[1] * sfb_ni[1];
        double v_nl[2] = {st.v_nl[0], st.v_nl[1]};
        for (int iter = 0; iter < 6; ++iter) {
            double ic0, gm0, ic1, gm1;
            bjt_ic_gm(v_nl[0], ic0, gm0);
            bjt_ic_gm(v_nl[1], ic1, gm1);
            const double f0 = v_nl[0] - p[0] - k00 * ic0 - k01 * ic1;
            const double f1 = v_nl[1] - p[1] - k10 * ic0 - k11 * ic1;
            if (std::fabs(f0) < 1e-9 && std::fabs(f1) < 1e-9) break;
            const double j00 = 1.0 - k00 * gm0, j01 = -k01 * gm1, j10 = -k10 * gm0, j11 = 1.0 - k11 * gm1;
            const double det = j00 * j11 - j01 * j10;
            if (std::fabs(det) < 1e-30) break;
            const double inv_det = 1.0 / det;
            v_nl[0] -= inv_det * (j11 * f0 - j01 * f1);
            v_nl[1] -= inv_det * (j00 * f1 - j10 * f0);
        }
        const double ic_new[2] = {bjt_ic(v_nl[0]), bjt_ic(v_nl[1])};
        const double sfb_ni_dot_ic = sfb_ni[0] * ic_new[0] + sfb_ni[1] * ic_new[1];
        for (int i = 0; i < N; ++i) {
            const double s_ni_i = ic_new[0] * (s_base[i][EMIT1] - s_base[i][COLL1]) + ic_new[1] * (s_base[i][EMIT2] - s_base[i][COLL2]);
            st.v[i] = v_pred[i] + s_ni_i - sm_k * sfb_ni_dot_ic * s_fb_col[i];
        }
        st.cin_rhs_prev = cin_rhs_now;
        const double dv_cin = input - st.v[BASE1];
        st.j_cin = -gc_1pc * dv_cin - c_cin * st.j_cin;
        st.i_nl[0] = ic_new[0]; st.i_nl[1] = ic_new[1];
        st.v_nl[0] = v_nl[0]; st.v_nl[1] = v_nl[1];
        return st.v[OUT];
    }

    // :557-618
    double process_sample(double input) {
        const double main_out = dk_step(main, input);
        const double pump = dk_step(shadow, 0.0);
        g_ldr_prev = g_ldr;
        const double result = main_out - pump;
        if (!std::isfinite(result)) { reset(); return 0.0; }
        return result;
    }
    // :620-626
    void set_ldr_resistance(double r) {
        const double new_r = std::fmax(r, 1000.0);
        if (std::fabs(new_r - r_ldr) > 0.01) { r_ldr = new_r; g_ldr = 1.0 / new_r; }
    }
    // :628-640
    void reset() {
        dk::Vec8 w;
        for (int i = 0; i < 8; ++i) w[i] = two_w[i] * 0.5;
        double v_nl_dc[2];
        full_dc_solve(g_dc_base, w, r_ldr, v_nl_dc, v_dc);
        g_ldr = 1.0 / r_ldr;
        g_ldr_prev = g_ldr;
        main = at_dc(g_cin, v_nl_dc, v_dc);
        shadow = main;
    }
};

// --------------------------------------------------- power_amp.rs (behavioural)
struct PowerAmp {  // power_amp.rs:168-240
    static constexpr double OPEN_LOOP_GAIN = 19000.0;
    static constexpr double HEADROOM = 22.0, CROSSOVER_VT = 0.013, QUIESCENT_GAIN = 0.1, NR_TOL = 1e-6;
    double feedback_beta, closed_loop_gain;
    PowerAmp() {
        feedback_beta = 220.0 / (220.0 + 15000.0);
        closed_loop_gain = OPEN_LOOP_GAIN / (1.0 + OPEN_LOOP_GAIN * feedback_beta);
    }
    inline void forward_path(double v, double& f_val, double& f_deriv) const {
        const double v_sq = v * v;
        const double vt_sq = CROSSOVER_VT * CROSSOVER_VT;
        const double exp_term = std::exp(-v_sq / vt_sq);
        const double q = QUIESCENT_GAIN;
        const double cross_gain = q + (1.0 - q) * (1.0 - exp_term);
        const double v_cross = v * cross_gain;
        const double dcross_dv = cross_gain + v * (1.0 - q) * (2.0 * v / vt_sq) * exp_term;
        const double tanh_arg = v_cross / HEADROOM;
        const double tanh_val = std::tanh(tanh_arg);
        f_val = HEADROOM * tanh_val;
        f_deriv = (1.0 - tanh_val * tanh_val) * dcross_dv;
    }
    double process(double input) const {
        double y = rclamp(input * closed_loop_gain, -HEADROOM + NR_TOL, HEADROOM - NR_TOL);
        for (int it = 0; it < 8; ++it) {
            const double error = input - feedback_beta * y;
            const double v = OPEN_LOOP_GAIN * error;
            double f_val, f_deriv;
            forward_path(v, f_val, f_deriv);
            const double residual = y - f_val;
            const double jacobian = 1.0 + OPEN_LOOP_GAIN * feedback_beta * f_deriv;
            const double delta = residual / jacobian;
            y -= delta;
            if (std::fabs(delta) < NR_TOL) break;
        }
        return y / HEADROOM;
    }
};

// ----------------------------------------------------------------- speaker.rs
struct Speaker {  // speaker.rs:50-139
    Biquad hpf, lpf;
    double character, sample_rate, a2, a3, thermal_coeff, thermal_alpha, thermal_state;
    void init(double sr) {
        hpf = Biquad::make(Biquad::HIGHPASS, 30.0, 0.75, sr);
        lpf = Biquad::make(Biquad::LOWPASS, 5500.0, 0.707, sr);
        character = 1.0;
        sample_rate = sr;
        a2 = a3 = thermal_coeff = 0.0;
        thermal_alpha = 1.0 / (5.0 * sr);
        thermal_state = 0.0;
        update_coefficients();
    }
    void set_character(double ch) {
        const double c = rclamp(ch, 0.0, 1.0);
        if (std::fabs(c - character) > 0.002) { character = c; update_coefficients(); }
    }
    void update_coefficients() {
        const double c = character;
        const double hpf_hz = 20.0 * std::pow(30.0 / 20.0, c);
        const double lpf_hz = 20000.0 * std::pow(5500.0 / 20000.0, c);
        hpf.set(Biquad::HIGHPASS, hpf_hz, 0.75, sample_rate);
        lpf.set(Biquad::LOWPASS, lpf_hz, 0.707, sample_rate);
        a2 = 0.2 * c;
        a3 = 0.6 * c;
        thermal_coeff = 2.0 * c;
    }
    double process(double input) {
        const double x2 = input * input;
        const double x3 = x2 * input;
        const double shaped = (input + a2 * x2 + a3 * x3) / (1.0 + a2 + a3);
        const double limited = character < 0.001 ? shaped : std::tanh(shaped);
        const double power = x2;
        thermal_state += (power - thermal_state) * thermal_alpha;
        const double thermal_gain = 1.0 / (1.0 + thermal_coeff * std::sqrt(thermal_state));
        const double filtered = hpf.process(limited * thermal_gain);
        return lpf.process(filtered);
    }
    void reset() { hpf.reset(); lpf.reset(); thermal_state = 0.0; }
};

}  // namespace owo
