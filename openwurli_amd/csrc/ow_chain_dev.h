// openwurli-hip: device code for the shared mono chain (gfx950, wave64, f64).
//
//   lane = engine.  Every function below is a per-lane serial recurrence; all pool constants
//   (OwConsts) are wave-uniform, so the compiler keeps the MNA matrices in SGPRs / scalar loads.
//
// Mirrors (citations into /root/reference/crates/openwurli-dsp/src/):
//   gen_tremolo.rs:1140-1218,1546-1633,2353-3116  Twin-T oscillator step (Schur NR, BE fallback)
//   tremolo.rs:121-167                             LED drive, CdS envelope, power law, depth divider
//   dk_preamp_legacy.rs:369-412,447-554,620-640    DC solve, dk_step, set_ldr_resistance, reset
//   oversampler.rs:41-45,108-139                   half-band allpass branches
//   power_amp.rs:206-240                           behavioural power amp
//   speaker.rs:81-132, filters.rs:13-59            speaker + RBJ biquads
#pragma once
#include "ow_voice_dev.h"

namespace owdev {

// ------------------------------------------------------------------ tremolo oscillator
OW_DEV double fast_exp(double x0) {  // gen_tremolo.rs:1140-1166 (pure arithmetic; bit-reproducible without contraction)
    const double x = clampd(x0, -40.0, 40.0);
    const double SHIFT = 6755399441055744.0;
    const double z = x * 1.4426950408889634 + SHIFT;
    const long long n_i64 = __double_as_longlong(z) - __double_as_longlong(SHIFT);
    const double n = z - SHIFT;       // == (double)n_i64 exactly (z is SHIFT plus an integer of at most 58): one add instead of a 64-bit int -> f64 conversion
    const double f = (x - n * 0.6931471803691238) - n * 1.9082149292705877e-10;
    const double p = 1.0 + f * (1.0 + f * (0.5 + f * (0.16666666666666607 + f * (0.04166666666665876 + f * 0.008333333333492337))));
    const double pow2n = __longlong_as_double((long long)(((unsigned long long)(1023 + n_i64)) << 52));
    return p * pow2n;
}

// The logarithmic branch only fires on large forward steps above vcrit (start-up transients); out of line so the
// two inlined log() bodies per call site do not bloat the Newton loop.
__device__ __noinline__ __attribute__((const)) double pnjlim_limited(double vnew, double vold, double vt, double vcrit) {
    if (vold >= 0.0) {
        const double arg = 1.0 + (vnew - vold) / vt;
        return arg > 0.0 ? vold + vt * log(arg) : vcrit;
    }
    return vt * log(vnew / vt);
}
OW_DEV double pnjlim(double vnew, double vold, double vt, double vcrit) {  // gen_tremolo.rs:1203-1218
    if (vnew > vcrit && fabs(vnew - vold) > vt + vt) return pnjlim_limited(vnew, vold, vt, vcrit);
    return vnew;
}

// Both Twin-T BJTs share one parameter set (gen_tremolo.rs:1098-1132): Ebers-Moll, sign=+1, ISE=ISC=0.
#define OW_T_IS 1.40000000000000003e-14
#define OW_T_VT 2.58519910000000012e-2
#define OW_T_BF 2.0e2
#define OW_T_BR 3.0e0
#define OW_T_VCRIT 7.21213101001093038e-1

struct Bjt { double ic, ib, j0, j1, j2, j3; };
OW_DEV Bjt bjt_eval(double vbe, double vbc) {  // gen_tremolo.rs:1546-1633 (Ebers-Moll return)
    const double is = OW_T_IS, vt = OW_T_VT, nf = 1.0, nr = 1.0, beta_f = OW_T_BF, beta_r = OW_T_BR, sign = 1.0;
    const double vbe_eff = sign * vbe, vbc_eff = sign * vbc;
    const double nf_vt = nf * vt, nr_vt = nr * vt;
    const double exp_be = fast_exp(OW_DIV_C(vbe_eff, 1.0 * OW_T_VT));   // nf_vt, nr_vt are this constant
    const double exp_bc = fast_exp(OW_DIV_C(vbc_eff, 1.0 * OW_T_VT));
    const double i_cc = is * (exp_be - exp_bc);
    const double ib_fwd = is / beta_f * (exp_be - 1.0);
    const double ib_rev = is / beta_r * (exp_bc - 1.0);
    Bjt r;
    r.ic = sign * (i_cc - is / beta_r * (exp_bc - 1.0));
    r.ib = sign * (ib_fwd + ib_rev + 0.0 + 0.0);
    r.j0 = is / nf_vt * exp_be;
    r.j1 = -(is / nr_vt) * exp_bc - (is / (beta_r * nr_vt)) * exp_bc;
    r.j2 = (is / (beta_f * nf_vt)) * exp_be + 0.0;
    r.j3 = (is / (beta_r * nr_vt)) * exp_bc + 0.0;
    return r;
}

// 4x4 Gaussian elimination with partial pivoting, register-resident (all indices static,
// row exchanges done with selects) -- gen_tremolo.rs:2515-2561.
OW_DEV bool solve4(double a[4][4], double b[4]) {
    bool singular = false;
    double yp[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int col = 0; col < 4; ++col) {
        // Pivot choice: the scan (first strict maximum of |a[row][col]|, rows col..3 in order) ends at row 2 for column 0 and at the diagonal
        // for the others on every sweep the oracle has seen; it does exactly when that entry beats the rows before it strictly and no row
        // after it beats it -- three / two / one compares.  The scan itself only runs when some lane of the wavefront disagrees.
        const int ex = col == 0 ? 2 : col;
        int max_row = ex;
        double max_val = fabs(a[ex][col]);
        bool ok = true;
#pragma unroll
        for (int row = col; row < 4; ++row) {
            if (row < ex) ok = ok && (max_val > fabs(a[row][col]));
            if (row > ex) ok = ok && !(fabs(a[row][col]) > max_val);
        }
        if (__builtin_amdgcn_ballot_w64(!singular && !ok) != 0ull) {
            max_row = col;
            max_val = fabs(a[col][col]);
#pragma unroll
            for (int row = col + 1; row < 4; ++row) {
                const double v = fabs(a[row][col]);
                if (v > max_val) { max_val = v; max_row = row; }
            }
        }
        if (!singular && max_val < 1e-15) singular = true;
        if (!singular) {
            // Row exchange.  The Twin-T Jacobian's column 0 has its largest entry in row 2 on every sweep the oracle has seen (815 515 of
            // 815 515 over 4 s, `owo_tremolo_stats`), and columns 1..3 never exchange: when every lane of the wavefront agrees on exactly
            // that, rows 0 and 2 trade places by name (no instruction, or a register move); any other choice anywhere takes the general
            // selects below, which only run when some lane picked an off-diagonal pivot (wave-uniform branches both).
            if (col == 0 && __builtin_amdgcn_ballot_w64(max_row != 2) == 0ull) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { const double x = a[0][j]; a[0][j] = a[2][j]; a[2][j] = x; }
                const double x = b[0]; b[0] = b[2]; b[2] = x;
            } else
            if (__builtin_amdgcn_ballot_w64(max_row != col) != 0ull) {
#pragma unroll
            for (int row = col + 1; row < 4; ++row) {
                const bool sw = (max_row == row);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const double x = a[col][j], y = a[row][j];
                    a[col][j] = sw ? y : x;
                    a[row][j] = sw ? x : y;
                }
                const double x = b[col], y = b[row];
                b[col] = sw ? y : x;
                b[row] = sw ? x : y;
            }
            }
            const double pivot = a[col][col];
            yp[col] = ow_rcp_refined(pivot);   // the quotients over this pivot (up to three factors + the back substitution) share its reciprocal:
                                               // ow_div_y(x, pivot, y) is ow_div(x, pivot) instruction for instruction
#pragma unroll
            for (int row = col + 1; row < 4; ++row) {
                const double factor = ow_div_y(a[row][col], pivot, yp[col]);
#pragma unroll
                for (int j = col + 1; j < 4; ++j) a[row][j] -= factor * a[col][j];
                b[row] -= factor * b[col];
            }
        }
    }
    if (!singular) {
#pragma unroll
        for (int i = 3; i >= 0; --i) {
            double sum = b[i];
#pragma unroll
            for (int j = i + 1; j < 4; ++j) sum -= a[i][j] * b[j];
            // (the reference tests |a[i][i]| < 1e-15 again here: a[i][i] is column i's pivot, final since that column, and it passed
            // this very test above -- or is NaN, which passes both -- so the second test can never fire)
            b[i] = ow_div_y(sum, a[i][i], yp[i]);
        }
    }
    return !singular;
}

// Trapezoidal-step matrices of the Twin-T solver staged in LDS: 110 wave-uniform f64 constants do not fit the 102 SGPRs
// (they spilled through v_readlane/v_writelane); a uniform-address ds_read broadcasts one value to the whole wavefront.
struct TremMats {
    double a_neg[7][7], s[7][7], k[4][4], s_ni[7][4];
};
OW_DEV void trem_mats_load(TremMats* __restrict__ m, const OwConsts* __restrict__ K, int tid, int nthreads) {
    double* dst = (double*)m;
    for (int i = tid; i < 49; i += nthreads) { dst[i] = (&K->t_a_neg[0][0])[i]; dst[49 + i] = (&K->t_s[0][0])[i]; }
    for (int i = tid; i < 16; i += nthreads) dst[98 + i] = (&K->t_k[0][0])[i];
    for (int i = tid; i < 28; i += nthreads) dst[114 + i] = (&K->t_s_ni[0][0])[i];
}

// DC operating point of the Twin-T (gen_tremolo.rs DC_OP / DC_NL_I): v[0..6], i_nl[0..3]
__device__ const double OW_TREM_DC[11] = {4.26480458363572357e0, 0.0, 1.24642300965575981e0, 2.75561285973736503e0, 6.66518981651571640e-1, 1.5e1, -2.28408414614134341e-3,
                                          7.72841164985201955e-5, 3.86420577732601037e-7, 2.20372764986731876e-3, 1.10186382445765932e-5};
struct TremState {   // register-resident part; v / i_prev / i_pp live in TremPark
    double env, r_ldr;
    uint32_t be_fallbacks;
};

// NR sweep shared by the trapezoidal solve (sparse v_d as emitted, gen_tremolo.rs:2423-2438) and the
// BE fallback (dense v_d, :2798-2817).  Returns true when converged within MAX_ITER (=50).
// SK: kk0 points into the constant block (wave-uniform): the kernel is re-read per sweep through an opaque SCALAR zero, i.e. by two
// s_load_dwordx16 into SGPRs, instead of 48 LDS reads per sweep through an opaque vector zero.
// One sweep: i_nl is advanced, true = this sweep met the convergence test.
template <bool BE, bool SK = false>
OW_DEV bool trem_nr_sweep(const double p[4], const double (*__restrict__ kk0)[4], double i_nl[4]) {
    {
        int z = 0, zs = 0;
        asm volatile("" : "+v"(z));   // opaque zero: K is re-read from LDS every sweep instead of held in 32 VGPRs
        asm volatile("" : "+s"(zs));
        const double (*__restrict__ kk)[4] = SK ? kk0 + zs : kk0 + z;
        double vd[4];
        vd[0] = p[0] + kk[0][0] * i_nl[0] + kk[0][1] * i_nl[1] + kk[0][2] * i_nl[2] + kk[0][3] * i_nl[3];
        if (BE) {
            vd[1] = p[1] + kk[1][0] * i_nl[0] + kk[1][1] * i_nl[1] + kk[1][2] * i_nl[2] + kk[1][3] * i_nl[3];
            vd[2] = p[2] + kk[2][0] * i_nl[0] + kk[2][1] * i_nl[1] + kk[2][2] * i_nl[2] + kk[2][3] * i_nl[3];
        } else {
            vd[1] = p[1] + kk[1][0] * i_nl[0] + kk[1][1] * i_nl[1] + kk[1][2] * i_nl[2];
            vd[2] = p[2] + kk[2][0] * i_nl[0] + kk[2][1] * i_nl[1] + kk[2][3] * i_nl[3];
        }
        vd[3] = p[3] + kk[3][0] * i_nl[0] + kk[3][1] * i_nl[1] + kk[3][2] * i_nl[2] + kk[3][3] * i_nl[3];
        const Bjt d0 = bjt_eval(vd[0], vd[1]);
        const Bjt d1 = bjt_eval(vd[2], vd[3]);
        const double f[4] = {i_nl[0] - d0.ic, i_nl[1] - d0.ib, i_nl[2] - d1.ic, i_nl[3] - d1.ib};
        double a[4][4];
        int z1 = 0;
        if (!SK) asm volatile("" : "+v"(z1));
        const double (*__restrict__ kj)[4] = SK ? kk : kk0 + z1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a[0][j] = (j == 0 ? 1.0 : 0.0) - d0.j0 * kj[0][j] - d0.j1 * kj[1][j];
            a[1][j] = (j == 1 ? 1.0 : 0.0) - d0.j2 * kj[0][j] - d0.j3 * kj[1][j];
            a[2][j] = (j == 2 ? 1.0 : 0.0) - d1.j0 * kj[2][j] - d1.j1 * kj[3][j];
            a[3][j] = (j == 3 ? 1.0 : 0.0) - d1.j2 * kj[2][j] - d1.j3 * kj[3][j];
        }
        double b[4] = {f[0], f[1], f[2], f[3]};
        const bool ok = solve4(a, b);
        if (ok) {
            if (!BE) {  // gen_tremolo.rs:2562-2713
                double i_trial[4], dv_trial[4], v_lim[4];
                bool lim_any = false;
#pragma unroll
                for (int q = 0; q < 4; ++q) i_trial[q] = i_nl[q] - b[q];
                int z2 = 0;
                if (!SK) asm volatile("" : "+v"(z2));
                const double (*__restrict__ kt)[4] = SK ? kk : kk0 + z2;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const double v_trial = p[q] + kt[q][0] * i_trial[0] + kt[q][1] * i_trial[1] + kt[q][2] * i_trial[2] + kt[q][3] * i_trial[3];
                    dv_trial[q] = v_trial - vd[q];
                    v_lim[q] = (fabs(dv_trial[q]) > 1e-4) ? pnjlim(v_trial, vd[q], OW_T_VT, OW_T_VCRIT) : v_trial;
                    lim_any = lim_any || !(v_lim[q] == v_trial);      // (a NaN trial counts as limited: the full path handles it)
                }
                bool any_limited = false;
                double ga = 1.0;
                // A junction the limiter left alone has v_lim == v_trial, so dv_lim is dv_trial bit for bit and the ratio below is x / x == 1:
                // never below ga.  The four divisions only run when some lane of the wavefront was limited (start-up transients).
                if (__builtin_amdgcn_ballot_w64(lim_any) != 0ull) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const double dv_lim = v_lim[q] - vd[q];
                    if (fabs(dv_trial[q]) > 1e-15) {
                        const double r = (dv_trial[q] * dv_lim < 0.0) ? 0.0 : clampd(ow_div(dv_lim, dv_trial[q]), 0.0, 1.0);
                        if (r < ga) { ga = r; any_limited = true; }
                    }
                }
                }
                const double max_dv = fmax(fmax(fmax(fabs(dv_trial[0] * ga), fabs(dv_trial[1] * ga)), fabs(dv_trial[2] * ga)), fabs(dv_trial[3] * ga));
                if (max_dv > 3.5) { ga *= fmax(ow_div(3.5, max_dv), 0.1); any_limited = true; }
#pragma unroll
                for (int q = 0; q < 4; ++q) i_nl[q] -= ga * b[q];
                if (!any_limited) {
                    bool conv = true;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const double dv = dv_trial[q] * ga;
                        const double thr = 1e-3 * fmax(fabs(vd[q]), fabs(vd[q] + dv)) + 1e-6;
                        if (fabs(dv) > thr) conv = false;
                    }
                    if (conv) return true;
                }
            } else {    // gen_tremolo.rs:2932-3056
                double dv[4], al[4] = {1.0, 1.0, 1.0, 1.0};
#pragma unroll
                for (int q = 0; q < 4; ++q) dv[q] = -(kk[q][0] * b[0] + kk[q][1] * b[1] + kk[q][2] * b[2] + kk[q][3] * b[3]);
                bool any_limited = false;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (fabs(dv[q]) > 1e-4) {
                        const double vl = pnjlim(vd[q] + dv[q], vd[q], OW_T_VT, OW_T_VCRIT);
                        const double ratio = fmax(ow_div(vl - vd[q], dv[q]), 0.01);
                        if (ratio < al[q]) { al[q] = ratio; if (ratio < 1.0) any_limited = true; }
                    }
                }
                { const double m = fmin(al[0], al[1]); al[0] = m; al[1] = m; }
                { const double m = fmin(al[2], al[3]); al[2] = m; al[3] = m; }
                const double max_dv = fmax(fmax(fmax(fabs(dv[0] * al[0]), fabs(dv[1] * al[1])), fabs(dv[2] * al[2])), fabs(dv[3] * al[3]));
                if (max_dv > 3.5) {
                    const double fct = fmax(ow_div(3.5, max_dv), 0.1);
#pragma unroll
                    for (int q = 0; q < 4; ++q) al[q] *= fct;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) i_nl[q] -= al[q] * b[q];
                if (!any_limited) {
                    bool conv = true;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const double stp = dv[q] * al[q];
                        const double thr = 1e-3 * fmax(fabs(vd[q]), fabs(vd[q] + stp)) + 1e-6;
                        if (fabs(stp) > thr) conv = false;
                    }
                    if (conv) return true;
                }
            }
        } else {  // singular Jacobian: damped fallback (gen_tremolo.rs:2715-2733 / 3058-3063)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double cl = BE ? 0.01 : fmax(fabs(i_nl[q]) * 0.1, 0.01);
                i_nl[q] -= clampd(f[q] * 0.5, -cl, cl);
            }
        }
    }
    return false;
}
template <bool BE, bool SK = false>
__device__ inline bool trem_nr(const double p[4], const double (*__restrict__ kk0)[4], double i_nl[4]) {
    for (int iter = 0; iter < 50; ++iter)
        if (trem_nr_sweep<BE, SK>(p, kk0, i_nl)) return true;
    return false;
}

// Backward-Euler retry (gen_tremolo.rs:2755-3080).  Never taken in normal operation; inlined behind __builtin_expect so the
// register allocator places its spills inside this cold block (a noinline callee cannot be register-capped).
__device__ inline void trem_be_fallback(const OwConsts* __restrict__ K, const double v_prev[7], const double i_prev[4], const double i_pp[4],
                                              double v_out[7], double i_nl[4]) {
    double rhs[7], vp[7], p[4];
    const double rc_be[7] = {0, 0, 0, 0, 0, 0, 15.0};
    const double ni[7][4] = {{-1, 0, -1, 0}, {0, 0, 0, 0}, {0, -1, 0, 0}, {0, 0, 0, 0}, {1, 1, 0, -1}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    const double nv[4][7] = {{0, 0, 1, 0, -1, 0, 0}, {-1, 0, 1, 0, 0, 0, 0}, {0, 0, 0, 0, 1, 0, 0}, {-1, 0, 0, 0, 1, 0, 0}};
    for (int i = 0; i < 7; ++i) {
        double sum = rc_be[i];
        for (int j = 0; j < 7; ++j) sum += K->t_a_neg_be[i][j] * v_prev[j];
        for (int j = 0; j < 4; ++j) sum += ni[i][j] * i_prev[j];
        rhs[i] = sum;
    }
    rhs[0] += 0.0 * (1.0 / 1.0e7);
    for (int i = 0; i < 7; ++i) {
        double sum = 0.0;
        for (int j = 0; j < 7; ++j) sum += K->t_s_be[i][j] * rhs[j];
        vp[i] = sum;
    }
    for (int i = 0; i < 4; ++i) {
        double sum = 0.0;
        for (int j = 0; j < 7; ++j) sum += nv[i][j] * vp[j];
        p[i] = sum;
    }
    for (int i = 0; i < 4; ++i) i_nl[i] = 2.0 * i_prev[i] - i_pp[i];
    trem_nr<true>(p, K->t_k_be, i_nl);
    for (int i = 0; i < 7; ++i) {
        double x = vp[i];
        for (int j = 0; j < 4; ++j) x += K->t_s_ni_be[i][j] * i_nl[j];
        v_out[i] = x;
    }
}

// gen_tremolo.rs:2353-3116 with input == 0.0 (Tremolo always drives the oscillator with silence,
// tremolo.rs:183).  Returns v[OUT].
// LDS home of the oscillator state of one wavefront (lane-minor): the 15 state doubles and v_pred live here, not in VGPRs,
// so the Newton sweep runs with ~40 fewer live registers and the kernel fits beside two voice wavefronts on a SIMD.
struct TremPark {
    double v[7][64], ip[4][64], ipp[4][64], vp[7][64];
};
__device__ inline TremPark* park_opaque(TremPark* P) {
    int z = 0;
    asm volatile("" : "+v"(z));   // opaque zero: no store-to-load forwarding through registers
    return P + z;
}

__device__ inline double trem_osc_step(TremState& st, TremPark* __restrict__ P0, const OwConsts* __restrict__ K, const TremMats* __restrict__ M) {
    const int ln = threadIdx.x & 63;
    double p[4], i_nl[4];
    {
        TremPark* P = park_opaque(P0);
        double sv[7], ip[4];
#pragma unroll
        for (int i = 0; i < 7; ++i) { sv[i] = P->v[i][ln] + 1e-25 - 1e-25; P->v[i][ln] = sv[i]; }
#pragma unroll
        for (int i = 0; i < 4; ++i) { ip[i] = P->ip[i][ln] + 1e-25 - 1e-25; P->ip[i][ln] = ip[i]; }
        const double (*__restrict__ an)[7] = M->a_neg;
        double rhs[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 15.0};  // RHS_CONST (gen_tremolo.rs:1055-1063)
        rhs[0] += an[0][0] * sv[0];
        rhs[0] += an[0][1] * sv[1];
        rhs[0] += an[0][3] * sv[3];
        rhs[0] += an[0][5] * sv[5];
        rhs[1] += an[1][0] * sv[0];
        rhs[1] += an[1][1] * sv[1];
        rhs[1] += an[1][2] * sv[2];
        rhs[2] += an[2][1] * sv[1];
        rhs[2] += an[2][2] * sv[2];
        rhs[2] += an[2][3] * sv[3];
        rhs[3] += an[3][0] * sv[0];
        rhs[3] += an[3][2] * sv[2];
        rhs[3] += an[3][3] * sv[3];
        rhs[4] += an[4][4] * sv[4];
        rhs[5] += an[5][0] * sv[0];
        rhs[5] += an[5][5] * sv[5];
        rhs[5] += an[5][6] * sv[6];
        // N_I entries are exactly +-1 (gen_tremolo.rs:519-562)
        rhs[0] += -1.0 * ip[0];
        rhs[0] += -1.0 * ip[2];
        rhs[2] += -1.0 * ip[1];
        rhs[4] += 1.0 * ip[0];
        rhs[4] += 1.0 * ip[1];
        rhs[4] += -1.0 * ip[3];
        rhs[0] += (0.0 + 0.0) * (1.0 / 1.0e7);  // input source, input == input_prev == 0
        double v_pred[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            double sum = 0.0;
#pragma unroll
            for (int j = 0; j < 7; ++j) sum += M->s[i][j] * rhs[j];
            v_pred[i] = sum;
            P->vp[i][ln] = sum;
        }
        // N_V entries are exactly +-1 (gen_tremolo.rs:479-516)
        p[0] = 1.0 * v_pred[2] + -1.0 * v_pred[4];
        p[1] = -1.0 * v_pred[0] + 1.0 * v_pred[2];
        p[2] = 1.0 * v_pred[4];
        p[3] = -1.0 * v_pred[0] + 1.0 * v_pred[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) i_nl[i] = 2.0 * ip[i] - P->ipp[i][ln];
    }
    // the 4x4 kernel K is wave-uniform: the lane = group kernels read it per sweep by two scalar loads (SGPRs) instead of 48 LDS reads
    // (k_tremolo 9.08 -> 8.64 ms per 131 072-oscillator block); OW_TREM_SK=0 restores the LDS path
#ifndef OW_TREM_SK
#define OW_TREM_SK 1
#endif
    const bool converged = OW_TREM_SK ? trem_nr<false, true>(p, K->t_k, i_nl) : trem_nr<false, false>(p, M->k, i_nl);
    TremPark* P = park_opaque(P0);
    double v[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        double x = P->vp[i][ln];
#pragma unroll
        for (int j = 0; j < 4; ++j) x += M->s_ni[i][j] * i_nl[j];
        v[i] = x;
    }
    if (__builtin_expect(!converged, 0)) {
        st.be_fallbacks += 1u;
        double v_prev[7], i_prev[4], i_pp[4];
        for (int i = 0; i < 7; ++i) v_prev[i] = P->v[i][ln];
        for (int i = 0; i < 4; ++i) { i_prev[i] = P->ip[i][ln]; i_pp[i] = P->ipp[i][ln]; }
        trem_be_fallback(K, v_prev, i_prev, i_pp, v, i_nl);
    }
    bool finite = true;
#pragma unroll
    for (int i = 0; i < 7; ++i) finite = finite && isfinite(v[i]);
    if (__builtin_expect(!finite, 0)) {  // NaN reset to the baked DC operating point (gen_tremolo.rs:3083-3093); constants read from memory on this cold path
        int z = 0;
        asm volatile("" : "+v"(z));
        const double* dc = OW_TREM_DC + z;
#pragma unroll
        for (int i = 0; i < 7; ++i) P->v[i][ln] = dc[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) { P->ip[i][ln] = dc[7 + i]; P->ipp[i][ln] = dc[7 + i]; }
        return dc[0];
    }
#pragma unroll
    for (int i = 0; i < 7; ++i) P->v[i][ln] = v[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) { P->ipp[i][ln] = P->ip[i][ln]; P->ip[i][ln] = i_nl[i]; }
    return v[0];
}

// Tremolo::process, oscillator half (tremolo.rs:121-146): LED drive -> CdS envelope -> power-law cell resistance r_ldr.
// It has no audio input and no dependence on the depth knob, so it is produced a block ahead (k_tremolo).
__device__ inline double trem_cell_r(TremState& st, TremPark* __restrict__ P, const OwConsts* __restrict__ K, const TremMats* __restrict__ M) {
    const double v_out = trem_osc_step(st, P, K, M);
    const double led = clampd(OW_DIV_C(10.95 - v_out, 10.95 - 0.70), 0.0, 1.0);
    const double coeff = led > st.env ? K->ldr_attack : K->ldr_release;
    st.env = led + coeff * (st.env - led);
    const double drive = clampd(st.env, 0.0, 1.0);
    if (drive < 1e-6) st.r_ldr = 1000000.0;
    else st.r_ldr = exp(K->ln_r_max + K->ln_min_minus_max * pow(drive, 0.9));
    return st.r_ldr;
}
// Tremolo::shunt_impedance (tremolo.rs:152-167): the vibrato-pot divider seen by fb_junction, applied where the preamp consumes R.
OW_DEV double trem_shunt(double depth, double r_ldr) {
    const double r_upper = 50000.0 * (1.0 - depth);
    const double r_lower = 50000.0 * depth;
    const double top = r_upper > 0.0 ? ow_div(r_upper * 18000.0, r_upper + 18000.0) : 0.0;
    const double branch = 680.0 + r_ldr;
    const double low = r_lower > 0.0 ? ow_div(r_lower * branch, r_lower + branch) : 0.0;
    return top + low;
}

// ------------------------------------------------------------------ LinearSmoother (engine.rs:67-130)
struct Smoother {
    double cur, target, step;
    uint32_t rem;
    OW_DEV double next() {
        if (rem > 0u) {
            cur += step;
            rem -= 1u;
            if (rem == 0u) cur = target;
        }
        return cur;
    }
    OW_DEV void retarget(double t, uint32_t ramp) {  // set_target after the host accepted it (|t - target| >= 1e-9)
        target = t;
        const double delta = t - cur;
        if (ramp == 0u) { cur = t; rem = 0u; return; }
        step = delta / (double)ramp;
        rem = ramp;
    }
};

// Value of lane ^ 32 (main <-> shadow) without the LDS crossbar: v_permlane32_swap exchanges the upper half of its first operand
// with the lower half of its second; with both = x the first result holds x[lane - 32] in lanes 32-63 and the second x[lane + 32]
// in lanes 0-31.
// xor32_t: the same exchange for the pool-sized kernels (two wavefronts per SIMD, vector-issue bound): OW_XOR32_SWAP picks the swap form
// there too, the default is the LDS crossbar's ds_bpermute, which costs them no vector issue slots
OW_DEV double xor32(double x);
OW_DEV double xor32_t(double x) {
#ifdef OW_XOR32_SWAP
    return xor32(x);
#else
    return __shfl_xor(x, 32);
#endif
}
OW_DEV double xor32(double x) {
    const int lo = __double2loint(x), hi = __double2hiint(x);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    const bool low_half = (threadIdx.x & 63) < 32;      // (lane of the wavefront: k_chain_row's preamp wavefronts are not wavefront 0 of their block)
    return __hiloint2double(low_half ? b[1] : b[0], low_half ? a[1] : a[0]);
}

// ------------------------------------------------------------------ legacy DK preamp
#define OW_P_IS 3.03e-14
#define OW_P_VT 0.026
// gm[]: the junctions' transconductances AT v_nl, beside i_nl (their currents at v_nl): the evaluation the reference's Newton loop opens
// every step with (bjt_ic_gm(state.v_nl), dk_preamp_legacy.rs:508-509) is a pure function of state.v_nl, and state.i_nl is by
// construction bjt_ic(state.v_nl) (at_dc :247, step 9 :548-549) -- so the step's last evaluation is carried into the next one instead of
// being repeated: one exponential pair per Newton UPDATE, none for the opening residual.  gm lives in registers only: dk_load derives it
// from v_nl once per block, the state rows in HBM are the reference's fields.
struct DkSt { double j_cin, cin_prev, v[8], i_nl[2], v_nl[2], gm[2]; };
#ifdef OW_DBG_COUNTERS
extern __device__ unsigned long long g_ow_dbg[8];      // (ow_melange_dev.h)
#endif

// exp() for the junction laws below, whose argument is clamped to [-1 V, 0.85 V] / V_T = [-38.5, 32.7]: the device library's f64 exp
// (x * log2(e) rounded to n, two-step reduction by ln 2, degree-11 polynomial, ldexp) without its overflow / underflow selects
// (two compares, three v_cndmask), which cannot fire here.  Same constants, same operations: bit-identical to exp() on the whole
// clamp range (tests/test_gpu_division.py::test_bounded_exp_is_the_library_exp).
OW_DEV double exp_bounded(double x) {
#ifdef OW_LIB_EXP
    return exp(x);
#else
    const double n = rint(x * __longlong_as_double(0x3ff71547652b82feLL));
    double r = __builtin_fma(__longlong_as_double((long long)0xbfe62e42fefa39efULL), n, x);
    r = __builtin_fma(__longlong_as_double((long long)0xbc7abc9e3b39803fULL), n, r);
    double p = __builtin_fma(__longlong_as_double(0x3e5ade156a5dcb37LL), r, __longlong_as_double(0x3e928af3fca7ab0cLL));
    p = __builtin_fma(r, p, __longlong_as_double(0x3ec71dee623fde64LL));
    p = __builtin_fma(r, p, __longlong_as_double(0x3efa01997c89e6b0LL));
    p = __builtin_fma(r, p, __longlong_as_double(0x3f2a01a014761f6eLL));
    p = __builtin_fma(r, p, __longlong_as_double(0x3f56c16c1852b7b0LL));
    p = __builtin_fma(r, p, __longlong_as_double(0x3f81111111122322LL));
    p = __builtin_fma(r, p, __longlong_as_double(0x3fa55555555502a1LL));
    p = __builtin_fma(r, p, __longlong_as_double(0x3fc5555555555511LL));
    p = __builtin_fma(r, p, __longlong_as_double(0x3fe000000000000bLL));
    p = __builtin_fma(r, p, 1.0);
    p = __builtin_fma(r, p, 1.0);
    return ldexp(p, (int)n);
#endif
}
// vbe.clamp(-1.0, VBE_MAX) as v_max_f64 + v_min_f64 (the compare-and-select form is two compares and four v_cndmask per junction).  The
// two differ for a NaN only: f64::clamp hands it on, this gives -1 V.  A junction voltage can only turn NaN in a step whose v_pred is
// already non-finite (the residuals and the 2x2 are built from p = N_v v_pred and sm_k, and sm_k is in every v_pred too; |det| < 1e-30
// leaves before 1 / det), every node voltage v = v_pred + ... of that step is then non-finite, the caller's guard (:610-615) sees a
// non-finite output and replaces the whole state by the DC solve: the currents of such a step are never used
// (tests/test_gpu_parity.py::test_preamp_nan_reset_in_every_chain_kernel).
OW_DEV double dk_clamp_vbe(double vbe) { return __builtin_fmin(__builtin_fmax(vbe, -1.0), 0.85); }
OW_DEV double dk_ic(double vbe) {  // dk_preamp_legacy.rs:663-666
    return OW_P_IS * (exp_bounded(OW_DIV_C(dk_clamp_vbe(vbe), OW_P_VT)) - 1.0);
}
OW_DEV void dk_ic_gm(double vbe, double& ic, double& gm) {  // :686-690
    const double e = exp_bounded(OW_DIV_C(dk_clamp_vbe(vbe), OW_P_VT));
    ic = OW_P_IS * (e - 1.0);
    gm = (OW_P_IS / OW_P_VT) * e;
}
// gm at the state's v_nl (block start, DC solve): see DkSt
OW_DEV void dk_refresh_gm(DkSt& st) {
    double ic;
    dk_ic_gm(st.v_nl[0], ic, st.gm[0]);
    dk_ic_gm(st.v_nl[1], ic, st.gm[1]);
}

// dk_step, dk_preamp_legacy.rs:447-554.  Node order BASE1,EMIT1,COLL1,EMIT2,EMIT2B,COLL2,OUT,FB.
// (Staging the 114 wave-uniform constants in LDS instead of spilled SGPRs was measured slower: the ds_read latency is exposed
// with two wavefronts per SIMD, the v_readlane of a spilled SGPR is not.)
// k_reload(): the same pointer behind an opaque scalar zero.  dk_step reads its 114 wave-uniform constants through four of
// these, so each group is fetched by a few s_load_dwordx16 from the scalar cache where it is used, instead of all 228 SGPRs
// being live across the sample loop, spilled to VGPR lanes and read back with ~290 v_readlane per sample.
__device__ inline const OwConsts* k_reload(const OwConsts* K) {
    int z = 0;
    asm volatile("" : "+s"(z));
    return K + z;
}
__device__ inline double dk_step(DkSt& st, double input, double g_ldr, double g_ldr_prev, const OwConsts* __restrict__ K0) {
    const OwConsts* __restrict__ K = k_reload(K0);
    // rhs = A_neg v (dk_preamp_legacy.rs:466).  A_neg = 2C/T - G has 20 structural non-zeros (resistor/capacitor stamps,
    // :283-309); the reference multiplies the zeros too, which adds exact +-0.0 terms, so skipping them is bit-identical
    // for finite v (a non-finite v still propagates through its node's own diagonal entry).  The sums also start at their first product
    // instead of the reference's 0.0: `0.0 + x` is x except for x = -0.0, and a row that is +-0.0 at this point gets 2w[i] added below,
    // which is non-zero or +0.0 (ow_consts_host.hpp) -- the row ends as the same bits either way.
    const double (*__restrict__ an)[8] = K->p_a_neg;
    const double* v = st.v;
    double rhs[8];
    rhs[0] = an[0][0] * v[0] + an[0][2] * v[2];
    rhs[1] = an[1][1] * v[1] + an[1][7] * v[7];
    rhs[2] = an[2][0] * v[0] + an[2][2] * v[2] + an[2][5] * v[5];
    rhs[3] = an[3][3] * v[3] + an[3][4] * v[4];
    rhs[4] = an[4][3] * v[3] + an[4][4] * v[4];
    rhs[5] = an[5][2] * v[2] + an[5][5] * v[5] + an[5][6] * v[6];
    rhs[6] = an[6][5] * v[5] + an[6][6] * v[6] + an[6][7] * v[7];
    rhs[7] = an[7][1] * v[1] + an[7][6] * v[6] + an[7][7] * v[7];
    rhs[7] -= g_ldr_prev * st.v[7];
    const double cin_now = K->p_g_cin * input + st.j_cin;
    rhs[0] += cin_now + st.cin_prev;
    rhs[1] += st.i_nl[0];
    rhs[2] -= st.i_nl[0];
    rhs[3] += st.i_nl[1];
    rhs[5] -= st.i_nl[1];
#pragma unroll
    for (int i = 0; i < 8; ++i) rhs[i] += K->p_two_w[i];
    double vpb[8];
    {
        const OwConsts* __restrict__ Kb = k_reload(K0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double sum = 0.0;
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += Kb->p_s[i][j] * rhs[j];
            vpb[i] = sum;
        }
        const OwConsts* __restrict__ Kc = k_reload(K0);
#pragma unroll
        for (int i = 4; i < 8; ++i) {
            double sum = 0.0;
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += Kc->p_s[i][j] * rhs[j];
            vpb[i] = sum;
        }
    }
    K = k_reload(K0);
    const double sm_k = ow_div(g_ldr, 1.0 + K->p_s_fb_fb * g_ldr);
    const double sm_vpred = sm_k * vpb[7];
    double v_pred[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v_pred[i] = vpb[i] - sm_vpred * K->p_s_fb_col[i];
    const double p0 = v_pred[0] - v_pred[1], p1 = v_pred[2] - v_pred[3];
    const double k00 = K->p_k[0][0] - sm_k * K->p_nv_sfb[0] * K->p_sfb_ni[0];
    const double k01 = K->p_k[0][1] - sm_k * K->p_nv_sfb[0] * K->p_sfb_ni[1];
    const double k10 = K->p_k[1][0] - sm_k * K->p_nv_sfb[1] * K->p_sfb_ni[0];
    const double k11 = K->p_k[1][1] - sm_k * K->p_nv_sfb[1] * K->p_sfb_ni[1];
    double vn0 = st.v_nl[0], vn1 = st.v_nl[1];
    // Newton on the two junctions (:505-532).  (ic, gm) always hold the evaluation AT (vn0, vn1): the state's own on entry (DkSt), a fresh
    // one after every update -- so the residual of a sweep needs no exponential, and the evaluation the reference makes after its loop
    // (:535, bjt_ic is the ic half of bjt_ic_gm) is the one already held, whichever way a lane left: converged, singular 2x2, six updates.
    // The sweeps of the wavefront's lanes run in ONE wave-uniform loop: a lane that has left the reference's loop at one of its two
    // `break`s (`done`) is no longer moved -- the evaluations it still takes part in repeat its last one bit for bit -- and the loop ends
    // when every lane has, or after the six updates: the same values lane by lane as a per-lane loop (which a SIMD runs for as long as
    // its slowest lane anyway) without a divergent loop's execution-mask bookkeeping.
    double ic0 = st.i_nl[0], ic1 = st.i_nl[1], gm0 = st.gm[0], gm1 = st.gm[1];
    bool done = false;
#ifdef OW_DBG_COUNTERS
    // development counters (tools/probe_dk_updates.py): Newton updates a lane needs itself, by half of the wavefront (k_preamp: lanes 0-31
    // main, 32-63 shadow), against the updates its wavefront executes
    unsigned own_updates = 0u, wave_updates = 0u;
#endif
    for (int iter = 0; iter < 6; ++iter) {
        const double f0 = vn0 - p0 - k00 * ic0 - k01 * ic1;
        const double f1 = vn1 - p1 - k10 * ic0 - k11 * ic1;
        done = done || (fabs(f0) < 1e-9 && fabs(f1) < 1e-9);
        if (__builtin_amdgcn_ballot_w64(!done) == 0ull) break;
#ifdef OW_DBG_COUNTERS
        own_updates += done ? 0u : 1u; wave_updates += 1u;
#endif
        const double j00 = 1.0 - k00 * gm0, j01 = -k01 * gm1, j10 = -k10 * gm0, j11 = 1.0 - k11 * gm1;
        const double det = j00 * j11 - j01 * j10;
        done = done || fabs(det) < 1e-30;
        const double inv_det = ow_div(1.0, det);
        const double n0 = vn0 - inv_det * (j11 * f0 - j01 * f1);
        const double n1 = vn1 - inv_det * (j00 * f1 - j10 * f0);
        vn0 = done ? vn0 : n0;
        vn1 = done ? vn1 : n1;
        dk_ic_gm(vn0, ic0, gm0);
        dk_ic_gm(vn1, ic1, gm1);
    }
#ifdef OW_DBG_COUNTERS
    {
        const int ln = threadIdx.x & 63;
        atomicAdd(&g_ow_dbg[ln < 32 ? 0 : 1], (unsigned long long)own_updates);
        if (ln == 0) { atomicAdd(&g_ow_dbg[2], (unsigned long long)wave_updates); atomicAdd(&g_ow_dbg[3], 1ull); }
        // (slots 4, 5: the slowest lane of each half)
        unsigned m = own_updates;
        for (int o = 16; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
        if (ln == 0) atomicAdd(&g_ow_dbg[4], (unsigned long long)m);
        if (ln == 32) atomicAdd(&g_ow_dbg[5], (unsigned long long)m);
    }
#endif
    K = k_reload(K0);
    const double dot = K->p_sfb_ni[0] * ic0 + K->p_sfb_ni[1] * ic1;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const double s_ni_i = ic0 * K->p_sni_d1[i] + ic1 * K->p_sni_d2[i];      // the two column differences of S come from the host
        st.v[i] = v_pred[i] + s_ni_i - sm_k * dot * K->p_s_fb_col[i];
    }
    st.cin_prev = cin_now;
    const double dv_cin = input - st.v[0];
    st.j_cin = -K->p_gc_1pc * dv_cin - K->p_c_cin * st.j_cin;
    st.i_nl[0] = ic0; st.i_nl[1] = ic1;
    st.v_nl[0] = vn0; st.v_nl[1] = vn1;
    st.gm[0] = gm0; st.gm[1] = gm1;
    return st.v[6];
}

// DkPreamp::reset -> full_dc_solve at the current R_ldr (dk_preamp_legacy.rs:369-412,628-640).
// Rare path (engine reset, NaN guards); dynamic indexing is fine here.
__device__ __noinline__ void dk_dc_state(const OwConsts* __restrict__ K, double r_ldr, DkSt* out) {
    double w[8][16];
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) { w[i][j] = K->p_g_dc_base[i][j]; w[i][8 + j] = (i == j) ? 1.0 : 0.0; }
    w[7][7] += 1.0 / r_ldr;
    for (int col = 0; col < 8; ++col) {
        int piv = col;
        double best = fabs(w[col][col]);
        for (int r = col + 1; r < 8; ++r)
            if (fabs(w[r][col]) > best) { best = fabs(w[r][col]); piv = r; }
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
#define SDC(i, j) w[i][8 + (j)]
    const double kd00 = SDC(0, 1) - SDC(0, 2) - SDC(1, 1) + SDC(1, 2);
    const double kd01 = SDC(0, 3) - SDC(0, 5) - SDC(1, 3) + SDC(1, 5);
    const double kd10 = SDC(2, 1) - SDC(2, 2) - SDC(3, 1) + SDC(3, 2);
    const double kd11 = SDC(2, 3) - SDC(2, 5) - SDC(3, 3) + SDC(3, 5);
    double wv[8], sv[8];
    for (int i = 0; i < 8; ++i) wv[i] = K->p_two_w[i] * 0.5;
    for (int i = 0; i < 8; ++i) {
        double sum = 0.0;
        for (int j = 0; j < 8; ++j) sum += SDC(i, j) * wv[j];
        sv[i] = sum;
    }
    const double pd0 = sv[0] - sv[1], pd1 = sv[2] - sv[3];
    double vn0 = 0.56, vn1 = 0.66;
    for (int iter = 0; iter < 100; ++iter) {
        double ic0, gm0, ic1, gm1;
        dk_ic_gm(vn0, ic0, gm0);
        dk_ic_gm(vn1, ic1, gm1);
        const double f0 = vn0 - pd0 - kd00 * ic0 - kd01 * ic1;
        const double f1 = vn1 - pd1 - kd10 * ic0 - kd11 * ic1;
        if (fabs(f0) < 1e-12 && fabs(f1) < 1e-12) break;
        const double j00 = 1.0 - kd00 * gm0, j01 = -kd01 * gm1, j10 = -kd10 * gm0, j11 = 1.0 - kd11 * gm1;
        const double det = j00 * j11 - j01 * j10;
        const double inv_det = 1.0 / det;
        const double dv0 = inv_det * (j11 * f0 - j01 * f1);
        const double dv1 = inv_det * (j00 * f1 - j10 * f0);
        const double ms = 2.0 * OW_P_VT;
        vn0 -= clampd(dv0, -ms, ms);
        vn1 -= clampd(dv1, -ms, ms);
    }
    const double ic0 = dk_ic(vn0), ic1 = dk_ic(vn1);
    double rhs[8];
    for (int i = 0; i < 8; ++i) rhs[i] = wv[i];
    rhs[1] += ic0; rhs[2] -= ic0; rhs[3] += ic1; rhs[5] -= ic1;
    for (int i = 0; i < 8; ++i) {
        double sum = 0.0;
        for (int j = 0; j < 8; ++j) sum += SDC(i, j) * rhs[j];
        out->v[i] = sum;
    }
#undef SDC
    out->j_cin = K->p_g_cin * out->v[0];     // DkState::at_dc (:241-250)
    out->cin_prev = K->p_g_cin * out->v[0];
    out->i_nl[0] = dk_ic(vn0); out->i_nl[1] = dk_ic(vn1);
    out->v_nl[0] = vn0; out->v_nl[1] = vn1;
}
// st = DC state.  The out-of-line solver writes a temporary: handing it &st would pin the hot loop's state to scratch memory
// (8 scratch loads + 8 stores per oversampled sample in k_preamp before this wrapper existed).
__device__ inline void dk_dc_reset(const OwConsts* __restrict__ K, double r_ldr, DkSt& st) {
    DkSt tmp;
    dk_dc_state(K, r_ldr, &tmp);
    st = tmp;
    dk_refresh_gm(st);
}

// ------------------------------------------------------------------ oversampler (oversampler.rs:17-45)
OW_DEV double allpass3(const double c0, const double c1, const double c2, double st[3], double x) {
    double y = c0 * x + st[0];
    st[0] = x - c0 * y;
    double y2 = c1 * y + st[1];
    st[1] = y - c1 * y2;
    double y3 = c2 * y2 + st[2];
    st[2] = y2 - c2 * y3;
    return y3;
}
#define OW_OS_A0 0.036681502163648
#define OW_OS_A1 0.248030921580110
#define OW_OS_A2 0.643184620136480
#define OW_OS_B0 0.110377634768680
#define OW_OS_B1 0.420399304190880
#define OW_OS_B2 0.854640112701920

// (tanh_fast: ow_voice_dev.h)

// ------------------------------------------------------------------ power amp (power_amp.rs:206-240)
__device__ inline double power_amp(double input) {
    const double A = 19000.0, H = 22.0, TOL = 1e-6;
    const double beta = 220.0 / (220.0 + 15000.0);
    const double clg = A / (1.0 + A * beta);
    double y = clampd(input * clg, -H + TOL, H - TOL);
#ifndef OW_DK_DIVERGENT_NEWTON
    bool done = false;     // (wave-uniform loop: see dk_step)
#endif
    for (int it = 0; it < 8; ++it) {
        const double error = input - beta * y;
        const double v = A * error;
        const double v_sq = v * v;
        const double vt_sq = 0.013 * 0.013;
        const double exp_term = exp(OW_DIV_C(-v_sq, 0.013 * 0.013));
        const double q = 0.1;
        const double cross_gain = q + (1.0 - q) * (1.0 - exp_term);
        const double v_cross = v * cross_gain;
        const double dcross_dv = cross_gain + v * (1.0 - q) * OW_DIV_C(2.0 * v, 0.013 * 0.013) * exp_term;
        const double tanh_val = tanh_fast(OW_DIV_C(v_cross, 22.0));
        const double f_val = H * tanh_val;
        const double f_deriv = (1.0 - tanh_val * tanh_val) * dcross_dv;
        const double residual = y - f_val;
        const double jac = 1.0 + A * beta * f_deriv;
        const double delta = ow_div(residual, jac);
#ifndef OW_DK_DIVERGENT_NEWTON
        y = done ? y : y - delta;
        done = done || fabs(delta) < TOL;
        if (__builtin_amdgcn_ballot_w64(!done) == 0ull) break;
#else
        y -= delta;
        if (fabs(delta) < TOL) break;
#endif
    }
    return OW_DIV_C(y, 22.0);
}

// ------------------------------------------------------------------ speaker (speaker.rs:81-132)
struct Bq { double b0, b1, b2, a1, a2, s1, s2; };
OW_DEV double bq_process(Bq& f, double x) {
    const double y = f.b0 * x + f.s1;
    f.s1 = f.b1 * x - f.a1 * y + f.s2;
    f.s2 = f.b2 * x - f.a2 * y;
    return y;
}
__device__ inline void bq_design(Bq& f, bool highpass, double fc, double q, double sr) {  // RBJ, state kept (filters.rs:39-49)
    const double w0 = 2.0 * 3.14159265358979323846 * fc / sr;
    const double cw = cos(w0), sw = sin(w0);
    const double alpha = sw / (2.0 * q);
    const double a0 = 1.0 + alpha;
    if (highpass) { f.b0 = ((1.0 + cw) / 2.0) / a0; f.b1 = (-(1.0 + cw)) / a0; f.b2 = ((1.0 + cw) / 2.0) / a0; }
    else          { f.b0 = ((1.0 - cw) / 2.0) / a0; f.b1 = (1.0 - cw) / a0;    f.b2 = ((1.0 - cw) / 2.0) / a0; }
    f.a1 = (-2.0 * cw) / a0;
    f.a2 = (1.0 - alpha) / a0;
}
struct SpeakerSt {
    Bq hpf, lpf;
    double character, a2, a3, tc, ts;
};
__device__ inline void speaker_update(SpeakerSt& s, double sr) {  // speaker.rs:89-103
    const double c = s.character;
    bq_design(s.hpf, true, 20.0 * pow(30.0 / 20.0, c), 0.75, sr);
    bq_design(s.lpf, false, 20000.0 * pow(5500.0 / 20000.0, c), 0.707, sr);
    s.a2 = 0.2 * c;
    s.a3 = 0.6 * c;
    s.tc = 2.0 * c;
}
OW_DEV void speaker_set_character(SpeakerSt& s, double ch, double sr) {  // speaker.rs:81-87
    const double c = clampd(ch, 0.0, 1.0);
    if (fabs(c - s.character) > 0.002) { s.character = c; speaker_update(s, sr); }
}
OW_DEV double speaker_process(SpeakerSt& s, double input, double thermal_alpha) {  // speaker.rs:105-132
    const double x2 = input * input;
    const double x3 = x2 * input;
    const double shaped = ow_div(input + s.a2 * x2 + s.a3 * x3, 1.0 + s.a2 + s.a3);
    const double limited = s.character < 0.001 ? shaped : tanh_fast(shaped);
    s.ts += (x2 - s.ts) * thermal_alpha;
    const double tg = ow_div(1.0, 1.0 + s.tc * sqrt(s.ts));
    const double filtered = bq_process(s.hpf, limited * tg);
    return bq_process(s.lpf, filtered);
}

}  // namespace owdev
