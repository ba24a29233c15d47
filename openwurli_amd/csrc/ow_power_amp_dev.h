// openwurli-hip: melange 7-BJT Class-AB power amp + rail dynamics on the device.
//
// Mirrors (citations into /root/reference/crates/openwurli-dsp/src/):
//   gen_power_amp.rs:7870-8017   bjt_evaluate (Gummel-Poon transport current, Ebers-Moll base current + ISE / ISC leakage)
//   gen_power_amp.rs:8032-8145   bjt_with_parasitics (inner 2-D Newton for RB / RC / RE, external Jacobian)
//   gen_power_amp.rs:8838-12337  process_sample: sparse build_rhs, S*rhs, 16-dim Schur Newton, pivoted 16x16 elimination, pnjlim +
//                                global step scale, BE-matrix retry, NaN reset, DC blocker, +-30 V clamp
//   power_amp.rs:65-165          RailDynamics
//   power_amp.rs:279-465         adapter: rail offsets, divergence guard (reset + hold last good), clamp
//
// Mapping: lane = engine, ONE wavefront per workgroup.  The 16x16 Newton Jacobian of every lane lives in LDS (lane-minor, 128 KB:
// no bank conflicts, and partial pivoting is an address: rows are reached through a per-lane permutation kept as 16 nibbles of one
// 64-bit register, so a row exchange moves no data).  The wave-uniform circuit matrices (S 20x20, K 16x16, S_NI 20x16, sparse A_neg)
// come from the constant block through scalar loads; the per-lane vectors are private arrays.  Every sum keeps the reference's
// operand order and the library is built without FMA contraction: fast_exp is pure arithmetic, divisions and square roots are IEEE,
// so the solver follows the CPU restatement bit for bit (only pnjlim's logarithm -- large forward steps -- can differ in the last
// place), which is what keeps the divergence guard firing on the same sample on both sides.
#pragma once
#include "ow_chain_dev.h"

namespace owdev {


// per-engine state rows of the power-amp buffer pa[PAS_COUNT][I]
enum {
    PAS_V = 0,        // [20] v_prev
    PAS_IP = 20,      // [16] i_nl_prev
    PAS_IPP = 36,     // [16] i_nl_prev_prev
    PAS_DCX = 52, PAS_DCY = 53,          // DC blocker memory
    PAS_PEAK = 54,    // diag_peak_output
    PAS_CLAMP = 55,   // u64 diag_clamp_count
    PAS_NRMAX = 56,   // u64 diag_nr_max_iter_count (== BE retries)
    PAS_NAN = 57,     // u64 diag_nan_reset_count
    PAS_CIRCUIT_END = 58,                // rows [0, 58) = gen_power_amp::CircuitState as the adapter clones it (settled-state blob)
    PAS_LASTGOOD = 58,
    PAS_RAILP = 59, PAS_RAILN = 60, PAS_IAVGP = 61, PAS_IAVGN = 62,
    PAS_GUARD = 63,   // u64 divergence-guard resets (diagnostic; the reference has no counter for it)
    PAS_COUNT = 64
};

struct PaBjt { double ic, ib, j0, j1, j2, j3; };

// gen_power_amp.rs:7870-8017, Gummel-Poon branch (all eight devices of the netlist set USE_GP; the host refuses anything else)
__device__ inline PaBjt pa_bjt_evaluate(double vbe, double vbc, const OwPaConsts::Dev& D) {
    const double vbe_eff = D.sign * vbe, vbc_eff = D.sign * vbc;
    const double exp_be = fast_exp(ow_div(vbe_eff, D.nf_vt));
    const double exp_bc = fast_exp(ow_div(vbc_eff, D.nr_vt));
    const bool has_ise = D.ise > 0.0, has_isc = D.isc > 0.0;
    const double exp_be_leak = has_ise ? fast_exp(ow_div(vbe_eff, D.ne_vt)) : 0.0;
    const double exp_bc_leak = has_isc ? fast_exp(ow_div(vbc_eff, D.nc_vt)) : 0.0;
    const double i_cc = D.is * (exp_be - exp_bc);
    const double ib_fwd = D.is_bf * (exp_be - 1.0);
    const double ib_rev = D.is_br * (exp_bc - 1.0);
    const double ib_leak_be = has_ise ? D.ise * (exp_be_leak - 1.0) : 0.0;
    const double ib_leak_bc = has_isc ? D.isc * (exp_bc_leak - 1.0) : 0.0;
    const double dib_fwd_dvbe = D.c_dib_fwd * exp_be;
    const double dib_rev_dvbc = D.c_dib_rev * exp_bc;
    const double dib_leak_dvbe = has_ise ? D.c_leak_be * exp_be_leak : 0.0;
    const double dib_leak_dvbc = has_isc ? D.c_leak_bc * exp_bc_leak : 0.0;
    const double q1_denom = 1.0 - ow_div(vbe_eff, D.var) - ow_div(vbc_eff, D.vaf);
    double q1 = 1.0, dq1_dvbe = 0.0, dq1_dvbc = 0.0;
    if (!(q1_denom <= 0.0 || fabs(q1_denom) < 1e-30)) {
        q1 = ow_div(1.0, q1_denom);
        dq1_dvbe = ow_div(q1 * q1, D.var);
        dq1_dvbc = ow_div(q1 * q1, D.vaf);
    }
    const double cbe = D.is * (exp_be - 1.0);
    const double cbc = D.is * (exp_bc - 1.0);
    const double q2 = ow_div(cbe, D.ikf) + ow_div(cbc, D.ikr);
    const double dq2_dvbe = D.c_dq2_be * exp_be;
    const double dq2_dvbc = D.c_dq2_bc * exp_bc;
    const double disc = fmax(1.0 + 4.0 * q2, 0.0);
    const double dd = sqrt(disc);
    const double dd_dvbe = dd > 1e-15 ? ow_div(2.0 * dq2_dvbe, dd) : 0.0;
    const double dd_dvbc = dd > 1e-15 ? ow_div(2.0 * dq2_dvbc, dd) : 0.0;
    const double qb = q1 * (1.0 + dd) * 0.5;                       // x / 2.0 == x * 0.5 exactly
    const double dqb_dvbe = dq1_dvbe * (1.0 + dd) * 0.5 + q1 * dd_dvbe * 0.5;
    const double dqb_dvbc = dq1_dvbc * (1.0 + dd) * 0.5 + q1 * dd_dvbc * 0.5;
    PaBjt r;
    r.ic = D.sign * (ow_div(i_cc, qb) - D.is_br * (exp_bc - 1.0));
    r.ib = D.sign * (ib_fwd + ib_rev + ib_leak_be + ib_leak_bc);
    const double dicc_dvbe = D.c_dicc_be * exp_be;
    const double dicc_dvbc = D.c_dicc_bc * exp_bc;
    const double qb2 = fmax(qb * qb, 1e-30);
    const double quotient_dvbe = ow_div(dicc_dvbe * qb - i_cc * dqb_dvbe, qb2);
    const double quotient_dvbc = ow_div(dicc_dvbc * qb - i_cc * dqb_dvbc, qb2);
    const double d_bc_term_dvbc = D.c_dib_rev * exp_bc;
    r.j0 = quotient_dvbe;
    r.j1 = quotient_dvbc - d_bc_term_dvbc;
    r.j2 = dib_fwd_dvbe + dib_leak_dvbe;
    r.j3 = dib_rev_dvbc + dib_leak_dvbc;
    return r;
}

// gen_power_amp.rs:8032-8145
__device__ __noinline__ PaBjt pa_bjt_with_parasitics(double vbe_ext, double vbc_ext, const OwPaConsts::Dev* __restrict__ Dp) {
    const OwPaConsts::Dev D = *Dp;
    double vbe_int = vbe_ext, vbc_int = vbc_ext;
    for (int it = 0; it < 15; ++it) {
        const PaBjt e = pa_bjt_evaluate(vbe_int, vbc_int, D);
        const double f1 = vbe_int - vbe_ext + e.ib * D.rb + (e.ic + e.ib) * D.re;
        const double f2 = vbc_int - vbc_ext + e.ib * D.rb - e.ic * D.rc;
        if (fabs(f1) < 1e-10 && fabs(f2) < 1e-10) break;
        const double j11 = 1.0 + e.j2 * D.rb + (e.j0 + e.j2) * D.re;
        const double j12 = e.j3 * D.rb + (e.j1 + e.j3) * D.re;
        const double j21 = e.j2 * D.rb - e.j0 * D.rc;
        const double j22 = 1.0 + e.j3 * D.rb - e.j1 * D.rc;
        const double det = j11 * j22 - j12 * j21;
        if (fabs(det) < 1e-30) break;
        const double inv_det = ow_div(1.0, det);
        double dvbe = (j22 * f1 - j12 * f2) * inv_det;
        double dvbc = (j11 * f2 - j21 * f1) * inv_det;
        dvbe = clampd(dvbe, -D.max_step, D.max_step);
        dvbc = clampd(dvbc, -D.max_step, D.max_step);
        vbe_int -= dvbe;
        vbc_int -= dvbc;
    }
    const PaBjt e = pa_bjt_evaluate(vbe_int, vbc_int, D);
    const double j11 = 1.0 + e.j2 * D.rb + (e.j0 + e.j2) * D.re;
    const double j12 = e.j3 * D.rb + (e.j1 + e.j3) * D.re;
    const double j21 = e.j2 * D.rb - e.j0 * D.rc;
    const double j22 = 1.0 + e.j3 * D.rb - e.j1 * D.rc;
    const double det = j11 * j22 - j12 * j21;
    if (fabs(det) < 1e-30) return e;
    const double inv_det = ow_div(1.0, det);
    const double fi11 = j22 * inv_det, fi12 = -j12 * inv_det, fi21 = -j21 * inv_det, fi22 = j11 * inv_det;
    PaBjt r;
    r.ic = e.ic; r.ib = e.ib;
    r.j0 = e.j0 * fi11 + e.j1 * fi21;
    r.j1 = e.j0 * fi12 + e.j1 * fi22;
    r.j2 = e.j2 * fi11 + e.j3 * fi21;
    r.j3 = e.j2 * fi12 + e.j3 * fi22;
    return r;
}

__device__ __noinline__ __attribute__((const)) double pa_pnjlim_limited(double vnew, double vold, double vt, double vcrit) {
    if (vold >= 0.0) {
        const double arg = 1.0 + (vnew - vold) / vt;
        return arg > 0.0 ? vold + vt * log(arg) : vcrit;
    }
    return vt * log(vnew / vt);
}
OW_DEV double pa_pnjlim(double vnew, double vold, double vt, double vcrit) {   // gen_power_amp.rs:7527-7545
    if (vnew > vcrit && fabs(vnew - vold) > vt + vt) return pa_pnjlim_limited(vnew, vold, vt, vcrit);
    return vnew;
}

// The solver state of one lane.  Private arrays (dynamic indices): the compiler keeps them in scratch, which the 64 lanes of the
// wavefront touch coalesced; the O(16^2) Jacobian traffic goes to LDS instead.
struct PaState {
    double v[PA_N], ip[PA_M], ipp[PA_M];
    double dcx, dcy, peak, last_good, rail_p, rail_n, iavg_p, iavg_n;
    unsigned long long clamp_cnt, nrmax_cnt, nan_cnt, guard_cnt;
    uint32_t last_nr;
};

#define PA_J(r, c) JA[(((r) * PA_M) + (c)) * 64]     // JA already points at this lane's column

// One Newton solve (the main sweep with K, or the BE retry with K_be): gen_power_amp.rs:8956-10680 / 10745-12245.
template <bool BE>
__device__ __noinline__ void pa_newton(const OwPaConsts* __restrict__ C, const double* __restrict__ p, double* __restrict__ i_nl, double* __restrict__ JA,
                                       uint32_t* __restrict__ last_nr) {
    const double (*__restrict__ kk)[PA_M] = BE ? C->k_be : C->k;
    for (int iter = 0; iter < 70; ++iter) {
        double vd[PA_M], f[PA_M], b[PA_M];
        for (int i = 0; i < PA_M; ++i) {
            double acc = p[i];
            for (int j = 0; j < PA_M; ++j) acc = acc + kk[i][j] * i_nl[j];
            vd[i] = acc;
        }
        for (int d = 0; d < 8; ++d) {
            const PaBjt e = pa_bjt_with_parasitics(vd[2 * d], vd[2 * d + 1], &C->dev[d]);
            f[2 * d] = i_nl[2 * d] - e.ic;
            f[2 * d + 1] = i_nl[2 * d + 1] - e.ib;
            for (int j = 0; j < PA_M; ++j) {                     // J[i][j] = delta_ij - jdev[i][2d] K[2d][j] - jdev[i][2d+1] K[2d+1][j]
                const double k0 = kk[2 * d][j], k1 = kk[2 * d + 1][j];
                PA_J(2 * d, j) = (j == 2 * d ? 1.0 : 0.0) - e.j0 * k0 - e.j1 * k1;
                PA_J(2 * d + 1, j) = (j == 2 * d + 1 ? 1.0 : 0.0) - e.j2 * k0 - e.j3 * k1;
            }
        }
        for (int i = 0; i < PA_M; ++i) b[i] = f[i];
        // pivoted elimination; logical row r lives in physical row (perm >> 4r) & 15
        unsigned long long perm = 0xFEDCBA9876543210ull;
        bool singular = false;
        for (int col = 0; col < PA_M && !singular; ++col) {
            int max_row = col;
            double max_val = fabs(PA_J((int)((perm >> (4 * col)) & 15ull), col));
            for (int row = col + 1; row < PA_M; ++row) {
                const double v = fabs(PA_J((int)((perm >> (4 * row)) & 15ull), col));
                if (v > max_val) { max_val = v; max_row = row; }
            }
            if (max_val < 1e-15) { singular = true; break; }
            if (max_row != col) {
                const unsigned long long pc = (perm >> (4 * col)) & 15ull, pm = (perm >> (4 * max_row)) & 15ull;
                perm = (perm & ~(15ull << (4 * col)) & ~(15ull << (4 * max_row))) | (pm << (4 * col)) | (pc << (4 * max_row));
                const double t = b[col]; b[col] = b[max_row]; b[max_row] = t;
            }
            const int pr = (int)((perm >> (4 * col)) & 15ull);
            const double pivot = PA_J(pr, col);
            const double bcol = b[col];
            for (int row = col + 1; row < PA_M; ++row) {
                const int rr = (int)((perm >> (4 * row)) & 15ull);
                const double factor = ow_div(PA_J(rr, col), pivot);
                for (int j = col + 1; j < PA_M; ++j) PA_J(rr, j) -= factor * PA_J(pr, j);
                b[row] -= factor * bcol;
            }
        }
        if (!singular) {
            for (int i = PA_M - 1; i >= 0; --i) {
                const int ri = (int)((perm >> (4 * i)) & 15ull);
                double sum = b[i];
                for (int j = i + 1; j < PA_M; ++j) sum -= PA_J(ri, j) * b[j];
                const double aii = PA_J(ri, i);
                if (fabs(aii) < 1e-15) { singular = true; break; }
                b[i] = ow_div(sum, aii);
            }
        }
        if (singular) {
            for (int i = 0; i < PA_M; ++i) {
                const double cl = BE ? 0.01 : fmax(fabs(i_nl[i]) * 0.1, 0.01);
                i_nl[i] -= clampd(f[i] * 0.5, -cl, cl);
            }
            continue;
        }
        if (!BE) {
            double dv_trial[PA_M], v_lim[PA_M], i_trial[PA_M];
            for (int i = 0; i < PA_M; ++i) i_trial[i] = i_nl[i] - b[i];
            for (int i = 0; i < PA_M; ++i) {
                double acc = p[i];
                for (int j = 0; j < PA_M; ++j) acc = acc + kk[i][j] * i_trial[j];
                dv_trial[i] = acc - vd[i];
                v_lim[i] = fabs(dv_trial[i]) > 1e-4 ? pa_pnjlim(acc, vd[i], C->dev[i >> 1].vt, C->dev[i >> 1].vcrit) : acc;
            }
            bool any_limited = false;
            double ga = 1.0;
            for (int i = 0; i < PA_M; ++i) {
                const double dv_lim = v_lim[i] - vd[i];
                if (fabs(dv_trial[i]) > 1e-15) {
                    const double r = dv_trial[i] * dv_lim < 0.0 ? 0.0 : clampd(ow_div(dv_lim, dv_trial[i]), 0.0, 1.0);
                    if (r < ga) { ga = r; any_limited = true; }
                }
            }
            double max_dv = fabs(dv_trial[0] * ga);
            for (int i = 1; i < PA_M; ++i) max_dv = fmax(max_dv, fabs(dv_trial[i] * ga));
            if (max_dv > 3.5) { ga *= fmax(ow_div(3.5, max_dv), 0.1); any_limited = true; }
            for (int i = 0; i < PA_M; ++i) i_nl[i] -= ga * b[i];
            if (!any_limited) {
                bool conv = true;
                for (int i = 0; i < PA_M; ++i) {
                    const double dv = dv_trial[i] * ga;
                    const double thr = 1e-3 * fmax(fabs(vd[i]), fabs(vd[i] + dv)) + 1e-6;
                    if (fabs(dv) > thr) conv = false;
                }
                if (conv) { *last_nr = (uint32_t)iter; return; }
            }
        } else {
            double dv[PA_M], al[PA_M];
            for (int i = 0; i < PA_M; ++i) {
                double acc = kk[i][0] * b[0];
                for (int j = 1; j < PA_M; ++j) acc = acc + kk[i][j] * b[j];
                dv[i] = -acc;
                al[i] = 1.0;
            }
            bool any_limited = false;
            for (int i = 0; i < PA_M; ++i) {
                if (fabs(dv[i]) > 1e-4) {
                    const double vl = pa_pnjlim(vd[i] + dv[i], vd[i], C->dev[i >> 1].vt, C->dev[i >> 1].vcrit);
                    const double ratio = fmax(ow_div(vl - vd[i], dv[i]), 0.01);
                    if (ratio < al[i]) { al[i] = ratio; if (ratio < 1.0) any_limited = true; }
                }
            }
            for (int d = 0; d < 8; ++d) { const double m = fmin(al[2 * d], al[2 * d + 1]); al[2 * d] = m; al[2 * d + 1] = m; }
            double max_dv = fabs(dv[0] * al[0]);
            for (int i = 1; i < PA_M; ++i) max_dv = fmax(max_dv, fabs(dv[i] * al[i]));
            if (max_dv > 3.5) {
                const double factor = fmax(ow_div(3.5, max_dv), 0.1);
                for (int i = 0; i < PA_M; ++i) al[i] *= factor;
            }
            for (int i = 0; i < PA_M; ++i) i_nl[i] -= al[i] * b[i];
            if (!any_limited) {
                bool conv = true;
                for (int i = 0; i < PA_M; ++i) {
                    const double stp = dv[i] * al[i];
                    const double thr = 1e-3 * fmax(fabs(vd[i]), fabs(vd[i] + stp)) + 1e-6;
                    if (fabs(stp) > thr) conv = false;
                }
                if (conv) { *last_nr = (uint32_t)iter; return; }
            }
        }
    }
}

// gen_power_amp.rs:8838-12337.  off_p / off_n: the runtime rail offsets (v_rail_pos_offset / v_rail_neg_offset of the state).
__device__ __noinline__ double pa_process_sample(PaState* __restrict__ st, const OwPaConsts* __restrict__ C, double input_in, double off_p, double off_n,
                                                 double* __restrict__ JA) {
    const double input = isfinite(input_in) ? clampd(input_in, -100.0, 100.0) : 0.0;
    for (int i = 0; i < PA_N; ++i) st->v[i] = st->v[i] + 1e-25 - 1e-25;
    for (int i = 0; i < PA_M; ++i) st->ip[i] = st->ip[i] + 1e-25 - 1e-25;
    double rhs[PA_N], v_pred[PA_N], p[PA_M], i_nl[PA_M], v[PA_N];
    for (int i = 0; i < PA_N; ++i) rhs[i] = PA_RHS_CONST[i];
    for (int q = 0; q < PA_RHS_NNZ; ++q) {
        const int i = (int)PA_RHS_NZ_ROW[q], j = (int)PA_RHS_NZ_COL[q];
        rhs[i] += C->a_neg[i][j] * st->v[j];
    }
    rhs[0] += input * (1.0 / PA_INPUT_RESISTANCE);
    rhs[18] += off_p;
    rhs[19] += off_n;
    for (int i = 0; i < PA_N; ++i) {
        double sum = 0.0;
        for (int j = 0; j < PA_N; ++j) sum += C->s[i][j] * rhs[j];
        v_pred[i] = sum;
    }
    for (int i = 0; i < PA_M; ++i) {
        const int na = (int)PA_P_NODE_A[i], nb = (int)PA_P_NODE_B[i];
        p[i] = PA_N_V[i][na] * v_pred[na] + PA_N_V[i][nb] * v_pred[nb];
    }
    for (int i = 0; i < PA_M; ++i) i_nl[i] = 2.0 * st->ip[i] - st->ipp[i];
    st->last_nr = 70u;
    pa_newton<false>(C, p, i_nl, JA, &st->last_nr);
    for (int i = 0; i < PA_N; ++i) {
        double x = v_pred[i];
        for (int j = 0; j < PA_M; ++j) x += C->s_ni[i][j] * i_nl[j];
        v[i] = x;
    }
    if (__builtin_expect(!(st->last_nr < 70u), 0)) {       // backward-Euler-matrix retry
        st->nrmax_cnt += 1ull;
        double rhs_be[PA_N], v_pred_be[PA_N], p_be[PA_M];
        for (int i = 0; i < PA_N; ++i) {
            double sum = PA_RHS_CONST_BE[i];
            for (int j = 0; j < PA_N; ++j) sum += C->a_neg_be[i][j] * st->v[j];
            for (int j = 0; j < PA_M; ++j) sum += PA_N_I[i][j] * st->ip[j];
            rhs_be[i] = sum;
        }
        rhs_be[0] += input * (1.0 / PA_INPUT_RESISTANCE);
        for (int i = 0; i < PA_N; ++i) {
            double sum = 0.0;
            for (int j = 0; j < PA_N; ++j) sum += C->s_be[i][j] * rhs_be[j];
            v_pred_be[i] = sum;
        }
        for (int i = 0; i < PA_M; ++i) {
            double sum = 0.0;
            for (int j = 0; j < PA_N; ++j) sum += PA_N_V[i][j] * v_pred_be[j];
            p_be[i] = sum;
        }
        for (int i = 0; i < PA_M; ++i) i_nl[i] = 2.0 * st->ip[i] - st->ipp[i];
        pa_newton<true>(C, p_be, i_nl, JA, &st->last_nr);
        for (int i = 0; i < PA_N; ++i) {
            double x = v_pred_be[i];
            for (int j = 0; j < PA_M; ++j) x += C->s_ni_be[i][j] * i_nl[j];
            v[i] = x;
        }
    }
    bool finite = true;
    for (int i = 0; i < PA_N; ++i) finite = finite && isfinite(v[i]);
    if (__builtin_expect(!finite, 0)) {
        for (int i = 0; i < PA_N; ++i) st->v[i] = PA_DC_OP[i];            // dc_operating_point == DC_OP (never re-set by the adapter)
        for (int i = 0; i < PA_M; ++i) { st->ip[i] = PA_DC_NL_I[i]; st->ipp[i] = PA_DC_NL_I[i]; }
        st->dcx = 0.0; st->dcy = 0.0;
        st->nan_cnt += 1ull;
        return PA_DC_BLOCK_X0;
    }
    for (int i = 0; i < PA_N; ++i) st->v[i] = v[i];
    for (int i = 0; i < PA_M; ++i) { st->ipp[i] = st->ip[i]; st->ip[i] = i_nl[i]; }
    const double raw_out = v[8];
    const double dc_blocked = raw_out - st->dcx + C->dc_block_r * st->dcy;
    st->dcx = raw_out;
    st->dcy = dc_blocked;
    const double scaled = dc_blocked * 1.0;
    const double abs_out = fabs(scaled);
    if (abs_out > st->peak) st->peak = abs_out;
    if (abs_out > 3e1) st->clamp_cnt += 1ull;
    return clampd(scaled, -3e1, 3e1);
}

// state <- settled blob (+ the per-state part of set_sample_rate when the chain does not run at the codegen rate): init_state,
// power_amp.rs:294-302
__device__ inline void pa_init_state(PaState* __restrict__ st, const double* __restrict__ settled, const OwPaConsts* __restrict__ C) {
    for (int i = 0; i < PA_N; ++i) st->v[i] = settled[PAS_V + i];
    for (int i = 0; i < PA_M; ++i) { st->ip[i] = settled[PAS_IP + i]; st->ipp[i] = settled[PAS_IPP + i]; }
    st->dcx = settled[PAS_DCX]; st->dcy = settled[PAS_DCY]; st->peak = settled[PAS_PEAK];
    st->clamp_cnt = dbits(settled[PAS_CLAMP]); st->nrmax_cnt = dbits(settled[PAS_NRMAX]); st->nan_cnt = dbits(settled[PAS_NAN]);
    if (!C->rate_is_codegen) { st->dcx = 0.0; st->dcy = 0.0; }
    st->last_nr = 0u;
}
__device__ inline void pa_rails_reset(PaState* __restrict__ st) { st->rail_p = 22.5; st->rail_n = 22.5; st->iavg_p = 0.0; st->iavg_n = 0.0; }

// melange_adapter::PowerAmp::process, power_amp.rs:373-431
__device__ inline double pa_process(PaState* __restrict__ st, const OwPaConsts* __restrict__ C, const double* __restrict__ settled, double input, bool rail_sag,
                                    double* __restrict__ JA) {
    const double off_p = rail_sag ? st->rail_p - 22.5 : 0.0, off_n = rail_sag ? st->rail_n - 22.5 : 0.0;
    const double raw = pa_process_sample(st, C, input, off_p, off_n, JA);
    const double result = OW_DIV_C(raw, 22.0);
    const bool nr_failed = st->last_nr >= 69u;
    bool insane = false;
    for (int i = 0; i < PA_N; ++i) insane = insane || !isfinite(st->v[i]) || fabs(st->v[i]) > 100.0;
    if (__builtin_expect(!isfinite(result) || nr_failed || insane, 0)) {
        pa_init_state(st, settled, C);
        pa_rails_reset(st);
        st->guard_cnt += 1ull;
        return st->last_good;
    }
    const double clamped = clampd(result, -1.0, 1.0);
    st->last_good = clamped;
    if (rail_sag) {        // RailDynamics::step(raw), power_amp.rs:131-156
        const double i_pos = fmax(OW_DIV_C(raw, 8.0), 0.0);
        const double i_neg = fmax(OW_DIV_C(-raw, 8.0), 0.0);
        st->iavg_p += C->alpha_i_avg * (i_pos - st->iavg_p);
        st->iavg_n += C->alpha_i_avg * (i_neg - st->iavg_n);
        const double target_pos = 24.5 - st->iavg_p * 3.5;
        const double target_neg = 24.5 - st->iavg_n * 3.5;
        const double alpha_p = target_pos < st->rail_p ? C->alpha_attack : C->alpha_release;
        const double alpha_n = target_neg < st->rail_n ? C->alpha_attack : C->alpha_release;
        st->rail_p += alpha_p * (target_pos - st->rail_p);
        st->rail_n += alpha_n * (target_neg - st->rail_n);
    }
    return clamped;
}

OW_DEV void pa_load(PaState* __restrict__ st, const double* __restrict__ pa, int I, int e) {
    for (int i = 0; i < PA_N; ++i) st->v[i] = pa[(size_t)(PAS_V + i) * I + e];
    for (int i = 0; i < PA_M; ++i) { st->ip[i] = pa[(size_t)(PAS_IP + i) * I + e]; st->ipp[i] = pa[(size_t)(PAS_IPP + i) * I + e]; }
    st->dcx = pa[(size_t)PAS_DCX * I + e]; st->dcy = pa[(size_t)PAS_DCY * I + e]; st->peak = pa[(size_t)PAS_PEAK * I + e];
    st->clamp_cnt = dbits(pa[(size_t)PAS_CLAMP * I + e]); st->nrmax_cnt = dbits(pa[(size_t)PAS_NRMAX * I + e]); st->nan_cnt = dbits(pa[(size_t)PAS_NAN * I + e]);
    st->last_good = pa[(size_t)PAS_LASTGOOD * I + e];
    st->rail_p = pa[(size_t)PAS_RAILP * I + e]; st->rail_n = pa[(size_t)PAS_RAILN * I + e];
    st->iavg_p = pa[(size_t)PAS_IAVGP * I + e]; st->iavg_n = pa[(size_t)PAS_IAVGN * I + e];
    st->guard_cnt = dbits(pa[(size_t)PAS_GUARD * I + e]);
    st->last_nr = 0u;
}
OW_DEV void pa_store(const PaState* __restrict__ st, double* __restrict__ pa, int I, int e) {
    for (int i = 0; i < PA_N; ++i) pa[(size_t)(PAS_V + i) * I + e] = st->v[i];
    for (int i = 0; i < PA_M; ++i) { pa[(size_t)(PAS_IP + i) * I + e] = st->ip[i]; pa[(size_t)(PAS_IPP + i) * I + e] = st->ipp[i]; }
    pa[(size_t)PAS_DCX * I + e] = st->dcx; pa[(size_t)PAS_DCY * I + e] = st->dcy; pa[(size_t)PAS_PEAK * I + e] = st->peak;
    pa[(size_t)PAS_CLAMP * I + e] = bitsd(st->clamp_cnt); pa[(size_t)PAS_NRMAX * I + e] = bitsd(st->nrmax_cnt); pa[(size_t)PAS_NAN * I + e] = bitsd(st->nan_cnt);
    pa[(size_t)PAS_LASTGOOD * I + e] = st->last_good;
    pa[(size_t)PAS_RAILP * I + e] = st->rail_p; pa[(size_t)PAS_RAILN * I + e] = st->rail_n;
    pa[(size_t)PAS_IAVGP * I + e] = st->iavg_p; pa[(size_t)PAS_IAVGN * I + e] = st->iavg_n;
    pa[(size_t)PAS_GUARD * I + e] = bitsd(st->guard_cnt);
}

// Settled state of the amp (compute_settled_state, power_amp.rs:290-296): CircuitState::default() (DC_OP + 50 warm-up samples) and
// 44 100 silent samples, all with the codegen-rate matrices.  One lane; cached per device by the host like the reference's OnceLock.
__global__ __launch_bounds__(64) void k_mpa_settle(const OwPaConsts* __restrict__ C88, double* __restrict__ settled) {
    __shared__ double JA_all[PA_M * PA_M * 64];
    if (threadIdx.x != 0) return;
    PaState st;
    for (int i = 0; i < PA_N; ++i) st.v[i] = PA_DC_OP[i];
    for (int i = 0; i < PA_M; ++i) { st.ip[i] = PA_DC_NL_I[i]; st.ipp[i] = PA_DC_NL_I[i]; }
    st.dcx = PA_DC_BLOCK_X0; st.dcy = 0.0; st.peak = 0.0; st.clamp_cnt = st.nrmax_cnt = st.nan_cnt = st.guard_cnt = 0ull;
    st.last_good = 0.0; st.last_nr = 0u;
    pa_rails_reset(&st);
    for (int n = 0; n < 50 + 44100; ++n) pa_process_sample(&st, C88, 0.0, 0.0, 0.0, JA_all);
    for (int i = 0; i < PA_N; ++i) settled[PAS_V + i] = st.v[i];
    for (int i = 0; i < PA_M; ++i) { settled[PAS_IP + i] = st.ip[i]; settled[PAS_IPP + i] = st.ipp[i]; }
    settled[PAS_DCX] = st.dcx; settled[PAS_DCY] = st.dcy; settled[PAS_PEAK] = st.peak;
    settled[PAS_CLAMP] = bitsd(st.clamp_cnt); settled[PAS_NRMAX] = bitsd(st.nrmax_cnt); settled[PAS_NAN] = bitsd(st.nan_cnt);
}

// The amp alone on given input (debug hook ow_debug_power_amp): lane b of block b processes row b of `in`.  taps [row][n][3]: outer Newton
// iterations of the sample (70 = exhausted), guard resets so far, positive rail after the sample.
__global__ __launch_bounds__(64) void k_mpa_debug(const OwPaConsts* __restrict__ C, const double* __restrict__ settled, const double* __restrict__ in,
                                                  double* __restrict__ out, double* __restrict__ taps, long long n, int rail_sag, const long long* __restrict__ poke_at,
                                                  const int* __restrict__ poke_node, const double* __restrict__ poke_val) {
    __shared__ double JA_all[PA_M * PA_M * 64];
    if (threadIdx.x != 0) return;
    const size_t row = blockIdx.x;
    PaState st;
    pa_init_state(&st, settled, C);
    pa_rails_reset(&st);
    st.last_good = 0.0; st.guard_cnt = 0ull;
    for (long long i = 0; i < n; ++i) {
        if (poke_at && poke_at[row] == i) st.v[poke_node[row]] = poke_val[row];
        out[row * n + i] = pa_process(&st, C, settled, in[row * n + i], rail_sag != 0, JA_all);
        if (taps) {
            double* t = taps + (row * n + i) * 3;
            t[0] = (double)st.last_nr; t[1] = (double)st.guard_cnt; t[2] = st.rail_p;
        }
    }
}

// PowerAmp::new_at_sample_rate (mode 1: fresh object -- last_good 0, rails at the DC bias) / PowerAmp::reset (mode 0: state and
// rails only, last_good survives) for engines [e0, e0+ne), power_amp.rs:335-347,453-458
__global__ void k_mpa_init(const OwPaConsts* __restrict__ C, const double* __restrict__ settled, double* __restrict__ pa, int I, int e0, int ne, int fresh) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ne) return;
    const int e = e0 + t;
    for (int r = 0; r < PAS_CIRCUIT_END; ++r) pa[(size_t)r * I + e] = settled[r];
    if (!C->rate_is_codegen) { pa[(size_t)PAS_DCX * I + e] = 0.0; pa[(size_t)PAS_DCY * I + e] = 0.0; }
    pa[(size_t)PAS_RAILP * I + e] = 22.5; pa[(size_t)PAS_RAILN * I + e] = 22.5; pa[(size_t)PAS_IAVGP * I + e] = 0.0; pa[(size_t)PAS_IAVGN * I + e] = 0.0;
    if (fresh) { pa[(size_t)PAS_LASTGOOD * I + e] = 0.0; pa[(size_t)PAS_GUARD * I + e] = bitsd(0ull); }
}

// Output stage with the melange power amp: lane = engine (the amp is a stateful recurrence at the chain rate, so the two chain-rate
// samples of an output sample are solved one after the other by the same lane), then half-band down, speaker, gain, f32 as k_post.
// One wavefront per workgroup (144 KB of the CU's 160 KB LDS).
__global__ __launch_bounds__(64) void k_post_mpa(const OwConsts* __restrict__ K, const OwPaConsts* __restrict__ C, const double* __restrict__ settled,
                                                 double* __restrict__ cs, double* __restrict__ pa, const OwEngineArgs* __restrict__ args,
                                                 OwEngineOut* __restrict__ eout, const double* __restrict__ pre, float* __restrict__ out,
                                                 double* __restrict__ pa_tap, int I, int L, int Lout, int e0, int ne) {
    __shared__ double JA_all[PA_M * PA_M * 64];          // 128 KB: the Jacobians of the 64 lanes
    __shared__ float tile[64 * (OW_OCHUNK + 1)];
    const int lane = threadIdx.x;
    double* JA = JA_all + lane;
    const int eb = e0 + blockIdx.x * 64;
    const int e_raw = eb + lane;
    const bool valid = e_raw < e0 + ne;
    const int e = valid ? e_raw : (e0 + ne - 1);
    const int osr = K->oversample ? 2 : 1;
    const double sr = K->sr;
    const double thermal_alpha = K->spk_thermal_alpha;
    const bool rail_sag = (args[e].pa_flags & 1u) != 0u;

    double da[3], db[3], dd;
    for (int i = 0; i < 3; ++i) { da[i] = CSF(CS_OS_DA + i); db[i] = CSF(CS_OS_DB + i); }
    dd = CSF(CS_OS_DD);
    SpeakerSt sp;
    {
        double* hp = &sp.hpf.b0; double* lp = &sp.lpf.b0;
        for (int i = 0; i < 7; ++i) { hp[i] = CSF(CS_SPK_HPF + i); lp[i] = CSF(CS_SPK_LPF + i); }
        sp.character = CSF(CS_SPK_CHAR); sp.a2 = CSF(CS_SPK_A2); sp.a3 = CSF(CS_SPK_A3); sp.tc = CSF(CS_SPK_TC); sp.ts = CSF(CS_SPK_TS);
    }
    Smoother ss, sv;
    smoother_load(ss, cs, I, e, CS_SM_SPK);
    smoother_load(sv, cs, I, e, CS_SM_VOL);
    if (args[e].set_flags & 2u) ss.retarget(args[e].spk_target, K->ramp_samples);
    if (args[e].set_flags & 4u) sv.retarget(args[e].vol_target, K->ramp_samples);
    PaState st;
    pa_load(&st, pa, I, e);
    bool nan_fired = false;
    for (int base = 0; base < L; base += OW_OCHUNK) {
        const int cn = min(OW_OCHUNK, L - base);
        for (int n = 0; n < cn; ++n) {
            double y[2] = {0.0, 0.0};
            for (int j = 0; j < osr; ++j) {
                const size_t idx = (size_t)(base + n) * osr + j;
                y[j] = pa_process(&st, C, settled, pre[idx * I + e] * 0.25, rail_sag, JA);     // x FIXED_CIRCUIT_DRIVE, engine.rs:544-546
                if (pa_tap && valid) pa_tap[idx * I + e] = y[j];
            }
            double o;
            if (osr == 2) {
                const double a = allpass3(OW_OS_A0, OW_OS_A1, OW_OS_A2, da, y[0]);
                const double b = allpass3(OW_OS_B0, OW_OS_B1, OW_OS_B2, db, y[1]);
                o = (a + dd) * 0.5;
                dd = b;
            } else {
                o = y[0];
            }
            speaker_set_character(sp, ss.next(), sr);
            const double shaped = speaker_process(sp, o, thermal_alpha);
            const double post = shaped * 7.498942093324558 * sv.next();
            float f = (float)post;
            if (!isfinite(f)) {                                                 // engine.rs:450-458 (power_amp.reset() included)
                f = 0.0f;
                sp.hpf.s1 = sp.hpf.s2 = sp.lpf.s1 = sp.lpf.s2 = 0.0;
                sp.ts = 0.0;
                pa_init_state(&st, settled, C);
                pa_rails_reset(&st);
                nan_fired = true;
            }
            tile[lane * (OW_OCHUNK + 1) + n] = f;
        }
        __syncthreads();
        for (int r = 0; r < 64; ++r) {
            const int er = eb + r;
            if (er < e0 + ne && lane < cn) out[(size_t)er * Lout + base + lane] = tile[r * (OW_OCHUNK + 1) + lane];
        }
        __syncthreads();
    }
    if (!valid) return;
    if (nan_fired) {
        for (int i = 0; i < 3; ++i) { da[i] = 0.0; db[i] = 0.0; }
        dd = 0.0;
        CSF(CS_FLAGS) = bitsd(dbits(CSF(CS_FLAGS)) | 1ull);
        eout[e].out_nonfinite = 1u;
    }
    for (int i = 0; i < 3; ++i) { CSF(CS_OS_DA + i) = da[i]; CSF(CS_OS_DB + i) = db[i]; }
    CSF(CS_OS_DD) = dd;
    {
        const double* hp = &sp.hpf.b0; const double* lp = &sp.lpf.b0;
        for (int i = 0; i < 7; ++i) { CSF(CS_SPK_HPF + i) = hp[i]; CSF(CS_SPK_LPF + i) = lp[i]; }
        CSF(CS_SPK_CHAR) = sp.character; CSF(CS_SPK_A2) = sp.a2; CSF(CS_SPK_A3) = sp.a3; CSF(CS_SPK_TC) = sp.tc; CSF(CS_SPK_TS) = sp.ts;
    }
    smoother_store(ss, cs, I, e, CS_SM_SPK);
    smoother_store(sv, cs, I, e, CS_SM_VOL);
    pa_store(&st, pa, I, e);
}

}  // namespace owdev
