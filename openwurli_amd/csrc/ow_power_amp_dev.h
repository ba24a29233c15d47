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
// Mapping and LDS layout: see the block above pa_newton (a lane pair per engine, 32 engines per wavefront, solver vectors in LDS).
// The wave-uniform circuit matrices (S 20x20, K 16x16, S_NI 20x16, sparse A_neg) come from the constant block through scalar loads.
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

// ---------------------------------------------------------------------------------------------------------------------------
// Mapping.  32 engines per wavefront, a LANE PAIR per engine (lane el = role 0, lane el + 32 = role 1), one wavefront per workgroup.
// Everything the solver indexes dynamically lives in LDS, engine-minor (row r of engine el at W[r * 32], W = base + el): the 16x16 Newton
// Jacobian, the vectors of the Newton sweep, the node / port state and the vectors of process_sample -- 512 rows = 128 KB.  (The first
// version kept the vectors in private arrays: 8.4 KB of scratch per lane, every access a trip to L2 / HBM, 6 ms per chain-rate sample.)
// The two lanes of a pair run the same control flow on the same values and share the O(16^2) work: four of the eight transistors each
// (device model + their Jacobian rows), every other row of an elimination step / matrix-vector product.  O(16) vector work is done by
// both lanes redundantly (same values written twice).  Partial pivoting moves no data: logical row r of the Jacobian AND of its
// right-hand side is physical row (perm >> 4r) & 15, perm a 64-bit register that both lanes keep.  PA_SYNC() orders the LDS traffic
// where one lane reads what its partner wrote (single-wavefront workgroup: the barrier itself is free).
// Every sum keeps the reference's operand order; the library is built without FMA contraction: the solver follows the CPU
// restatement bit for bit (pnjlim's logarithm excepted), which keeps the divergence guard firing on the same sample on both sides.
#define PA_EPW 32
enum {
    PL_J = 0,         // [16][16] Jacobian
    PL_B = 256,       // [16] right-hand side, by PHYSICAL row
    PL_X = 272,       // [16] solution of the linear system (delta)
    PL_VD = 288, PL_F = 304, PL_INL = 320, PL_P = 336,
    PL_T0 = 352,      // dv_trial / dv
    PL_T1 = 368,      // v_lim / alpha
    PL_T2 = 384,      // i_trial
    PL_V = 400,       // [20] v_prev
    PL_IP = 420, PL_IPP = 436,
    PL_RHS = 452, PL_VPRED = 472, PL_VNEW = 492,
    PL_ROWS = 512
};
#define PL(r) W[(r) * PA_EPW]
#define PA_JE(r, c) W[(PL_J + (r) * PA_M + (c)) * PA_EPW]
#define PA_SYNC() __syncthreads()
#define PA_PERM(p, r) ((int)(((p) >> (4 * (r))) & 15ull))

struct PaScal {      // per-engine scalars, identical in both lanes of the pair
    double dcx, dcy, peak, last_good, rail_p, rail_n, iavg_p, iavg_n;
    unsigned long long clamp_cnt, nrmax_cnt, nan_cnt, guard_cnt;
    uint32_t last_nr;
};

// One Newton solve (main sweep with K, or the BE retry with K_be): gen_power_amp.rs:8956-10680 / 10745-12245.  p in PL_P, i_nl in PL_INL.
template <bool BE>
__device__ __noinline__ uint32_t pa_newton(const OwPaConsts* __restrict__ C, double* __restrict__ W, int role) {
    const double (*__restrict__ kk)[PA_M] = BE ? C->k_be : C->k;
    for (int iter = 0; iter < 70; ++iter) {
        for (int d = role; d < 8; d += 2) {                       // four transistors per lane
            double vd2[2];
            for (int q = 0; q < 2; ++q) {
                const int r = 2 * d + q;
                double acc = PL(PL_P + r);
                for (int j = 0; j < PA_M; ++j) acc = acc + kk[r][j] * PL(PL_INL + j);
                vd2[q] = acc;
                PL(PL_VD + r) = acc;
            }
            const PaBjt e = pa_bjt_with_parasitics(vd2[0], vd2[1], &C->dev[d]);
            const double f0 = PL(PL_INL + 2 * d) - e.ic, f1 = PL(PL_INL + 2 * d + 1) - e.ib;
            PL(PL_F + 2 * d) = f0; PL(PL_F + 2 * d + 1) = f1;
            PL(PL_B + 2 * d) = f0; PL(PL_B + 2 * d + 1) = f1;
            for (int j = 0; j < PA_M; ++j) {                      // J[i][j] = delta_ij - jdev[i][2d] K[2d][j] - jdev[i][2d+1] K[2d+1][j]
                const double k0 = kk[2 * d][j], k1 = kk[2 * d + 1][j];
                PA_JE(2 * d, j) = (j == 2 * d ? 1.0 : 0.0) - e.j0 * k0 - e.j1 * k1;
                PA_JE(2 * d + 1, j) = (j == 2 * d + 1 ? 1.0 : 0.0) - e.j2 * k0 - e.j3 * k1;
            }
        }
        PA_SYNC();
        unsigned long long perm = 0xFEDCBA9876543210ull;
        bool singular = false;
        for (int col = 0; col < PA_M; ++col) {
            int max_row = col;
            double max_val = fabs(PA_JE(PA_PERM(perm, col), col));
            for (int row = col + 1; row < PA_M; ++row) {
                const double v = fabs(PA_JE(PA_PERM(perm, row), col));
                if (v > max_val) { max_val = v; max_row = row; }
            }
            if (max_val < 1e-15) { singular = true; break; }
            if (max_row != col) {
                const unsigned long long pc = (perm >> (4 * col)) & 15ull, pm = (perm >> (4 * max_row)) & 15ull;
                perm = (perm & ~(15ull << (4 * col)) & ~(15ull << (4 * max_row))) | (pm << (4 * col)) | (pc << (4 * max_row));
            }
            const int pr = PA_PERM(perm, col);
            const double pivot = PA_JE(pr, col);
            const double bcol = PL(PL_B + pr);
            for (int row = col + 1 + role; row < PA_M; row += 2) {     // the pair shares the rows below the pivot
                const int rr = PA_PERM(perm, row);
                const double factor = ow_div(PA_JE(rr, col), pivot);
                for (int j = col + 1; j < PA_M; ++j) PA_JE(rr, j) -= factor * PA_JE(pr, j);
                PL(PL_B + rr) -= factor * bcol;
            }
            PA_SYNC();
        }
        if (!singular) {
            for (int i = PA_M - 1; i >= 0; --i) {                 // both lanes: same values, each reads back what it wrote itself
                const int ri = PA_PERM(perm, i);
                double sum = PL(PL_B + ri);
                for (int j = i + 1; j < PA_M; ++j) sum -= PA_JE(ri, j) * PL(PL_X + j);
                const double aii = PA_JE(ri, i);
                if (fabs(aii) < 1e-15) { singular = true; break; }
                PL(PL_X + i) = ow_div(sum, aii);
            }
        }
        PA_SYNC();
        if (singular) {
            for (int i = 0; i < PA_M; ++i) {
                const double inl = PL(PL_INL + i);
                const double cl = BE ? 0.01 : fmax(fabs(inl) * 0.1, 0.01);
                PL(PL_INL + i) = inl - clampd(PL(PL_F + i) * 0.5, -cl, cl);
            }
            PA_SYNC();
            continue;
        }
        bool converged = false;
        if (!BE) {
            for (int i = 0; i < PA_M; ++i) PL(PL_T2 + i) = PL(PL_INL + i) - PL(PL_X + i);       // i_trial (both lanes, same values)
            for (int i = role; i < PA_M; i += 2) {                                            // v_trial rows shared
                double acc = PL(PL_P + i);
                for (int j = 0; j < PA_M; ++j) acc = acc + kk[i][j] * PL(PL_T2 + j);
                const double vdi = PL(PL_VD + i);
                const double dvt = acc - vdi;
                PL(PL_T0 + i) = dvt;
                PL(PL_T1 + i) = fabs(dvt) > 1e-4 ? pa_pnjlim(acc, vdi, C->dev[i >> 1].vt, C->dev[i >> 1].vcrit) : acc;
            }
            PA_SYNC();
            bool any_limited = false;
            double ga = 1.0;
            for (int i = 0; i < PA_M; ++i) {
                const double dvt = PL(PL_T0 + i);
                const double dv_lim = PL(PL_T1 + i) - PL(PL_VD + i);
                if (fabs(dvt) > 1e-15) {
                    const double r = dvt * dv_lim < 0.0 ? 0.0 : clampd(ow_div(dv_lim, dvt), 0.0, 1.0);
                    if (r < ga) { ga = r; any_limited = true; }
                }
            }
            double max_dv = fabs(PL(PL_T0) * ga);
            for (int i = 1; i < PA_M; ++i) max_dv = fmax(max_dv, fabs(PL(PL_T0 + i) * ga));
            if (max_dv > 3.5) { ga *= fmax(ow_div(3.5, max_dv), 0.1); any_limited = true; }
            for (int i = 0; i < PA_M; ++i) PL(PL_INL + i) = PL(PL_INL + i) - ga * PL(PL_X + i);
            if (!any_limited) {
                bool conv = true;
                for (int i = 0; i < PA_M; ++i) {
                    const double dv = PL(PL_T0 + i) * ga, vdi = PL(PL_VD + i);
                    const double thr = 1e-3 * fmax(fabs(vdi), fabs(vdi + dv)) + 1e-6;
                    if (fabs(dv) > thr) conv = false;
                }
                converged = conv;
            }
        } else {
            for (int i = role; i < PA_M; i += 2) {                                            // dv rows shared
                double acc = kk[i][0] * PL(PL_X);
                for (int j = 1; j < PA_M; ++j) acc = acc + kk[i][j] * PL(PL_X + j);
                PL(PL_T0 + i) = -acc;
            }
            PA_SYNC();
            bool any_limited = false;
            for (int i = 0; i < PA_M; ++i) {
                const double dvi = PL(PL_T0 + i), vdi = PL(PL_VD + i);
                double al = 1.0;
                if (fabs(dvi) > 1e-4) {
                    const double vl = pa_pnjlim(vdi + dvi, vdi, C->dev[i >> 1].vt, C->dev[i >> 1].vcrit);
                    const double ratio = fmax(ow_div(vl - vdi, dvi), 0.01);
                    if (ratio < al) { al = ratio; if (ratio < 1.0) any_limited = true; }
                }
                PL(PL_T1 + i) = al;
            }
            for (int d = 0; d < 8; ++d) { const double m = fmin(PL(PL_T1 + 2 * d), PL(PL_T1 + 2 * d + 1)); PL(PL_T1 + 2 * d) = m; PL(PL_T1 + 2 * d + 1) = m; }
            double max_dv = fabs(PL(PL_T0) * PL(PL_T1));
            for (int i = 1; i < PA_M; ++i) max_dv = fmax(max_dv, fabs(PL(PL_T0 + i) * PL(PL_T1 + i)));
            if (max_dv > 3.5) {
                const double factor = fmax(ow_div(3.5, max_dv), 0.1);
                for (int i = 0; i < PA_M; ++i) PL(PL_T1 + i) = PL(PL_T1 + i) * factor;
            }
            for (int i = 0; i < PA_M; ++i) PL(PL_INL + i) = PL(PL_INL + i) - PL(PL_T1 + i) * PL(PL_X + i);
            if (!any_limited) {
                bool conv = true;
                for (int i = 0; i < PA_M; ++i) {
                    const double stp = PL(PL_T0 + i) * PL(PL_T1 + i), vdi = PL(PL_VD + i);
                    const double thr = 1e-3 * fmax(fabs(vdi), fabs(vdi + stp)) + 1e-6;
                    if (fabs(stp) > thr) conv = false;
                }
                converged = conv;
            }
        }
        PA_SYNC();
        if (converged) return (uint32_t)iter;
    }
    return 70u;
}

// gen_power_amp.rs:8838-12337.  off_p / off_n: the runtime rail offsets (v_rail_pos_offset / v_rail_neg_offset of the state).
__device__ __noinline__ double pa_process_sample(PaScal* __restrict__ sc, const OwPaConsts* __restrict__ C, double* __restrict__ W, int role, double input_in,
                                                 double off_p, double off_n) {
    const double input = isfinite(input_in) ? clampd(input_in, -100.0, 100.0) : 0.0;
    for (int i = 0; i < PA_N; ++i) PL(PL_V + i) = PL(PL_V + i) + 1e-25 - 1e-25;
    for (int i = 0; i < PA_M; ++i) PL(PL_IP + i) = PL(PL_IP + i) + 1e-25 - 1e-25;
    for (int i = 0; i < PA_N; ++i) PL(PL_RHS + i) = PA_RHS_CONST[i];
    for (int q = 0; q < PA_RHS_NNZ; ++q) {
        const int i = (int)PA_RHS_NZ_ROW[q], j = (int)PA_RHS_NZ_COL[q];
        PL(PL_RHS + i) = PL(PL_RHS + i) + C->a_neg[i][j] * PL(PL_V + j);
    }
    PL(PL_RHS + 0) = PL(PL_RHS + 0) + input * (1.0 / PA_INPUT_RESISTANCE);
    PL(PL_RHS + 18) = PL(PL_RHS + 18) + off_p;
    PL(PL_RHS + 19) = PL(PL_RHS + 19) + off_n;
    for (int i = role; i < PA_N; i += 2) {                        // v_pred = S rhs, rows shared
        double sum = 0.0;
        for (int j = 0; j < PA_N; ++j) sum += C->s[i][j] * PL(PL_RHS + j);
        PL(PL_VPRED + i) = sum;
    }
    PA_SYNC();
    for (int i = 0; i < PA_M; ++i) {
        const int na = (int)PA_P_NODE_A[i], nb = (int)PA_P_NODE_B[i];
        PL(PL_P + i) = PA_N_V[i][na] * PL(PL_VPRED + na) + PA_N_V[i][nb] * PL(PL_VPRED + nb);
        PL(PL_INL + i) = 2.0 * PL(PL_IP + i) - PL(PL_IPP + i);
    }
    PA_SYNC();
    sc->last_nr = pa_newton<false>(C, W, role);
    for (int i = role; i < PA_N; i += 2) {
        double x = PL(PL_VPRED + i);
        for (int j = 0; j < PA_M; ++j) x += C->s_ni[i][j] * PL(PL_INL + j);
        PL(PL_VNEW + i) = x;
    }
    PA_SYNC();
    if (__builtin_expect(!(sc->last_nr < 70u), 0)) {              // backward-Euler-matrix retry
        sc->nrmax_cnt += 1ull;
        for (int i = role; i < PA_N; i += 2) {
            double sum = PA_RHS_CONST_BE[i];
            for (int j = 0; j < PA_N; ++j) sum += C->a_neg_be[i][j] * PL(PL_V + j);
            for (int j = 0; j < PA_M; ++j) sum += PA_N_I[i][j] * PL(PL_IP + j);
            PL(PL_RHS + i) = sum;
        }
        PA_SYNC();
        PL(PL_RHS + 0) = PL(PL_RHS + 0) + input * (1.0 / PA_INPUT_RESISTANCE) * (role == 0 ? 1.0 : 0.0);   // one lane adds the input term
        PA_SYNC();
        for (int i = role; i < PA_N; i += 2) {
            double sum = 0.0;
            for (int j = 0; j < PA_N; ++j) sum += C->s_be[i][j] * PL(PL_RHS + j);
            PL(PL_VPRED + i) = sum;
        }
        PA_SYNC();
        for (int i = 0; i < PA_M; ++i) {
            double sum = 0.0;
            for (int j = 0; j < PA_N; ++j) sum += PA_N_V[i][j] * PL(PL_VPRED + j);
            PL(PL_P + i) = sum;
            PL(PL_INL + i) = 2.0 * PL(PL_IP + i) - PL(PL_IPP + i);
        }
        PA_SYNC();
        sc->last_nr = pa_newton<true>(C, W, role);
        for (int i = role; i < PA_N; i += 2) {
            double x = PL(PL_VPRED + i);
            for (int j = 0; j < PA_M; ++j) x += C->s_ni_be[i][j] * PL(PL_INL + j);
            PL(PL_VNEW + i) = x;
        }
        PA_SYNC();
    }
    bool finite = true;
    for (int i = 0; i < PA_N; ++i) finite = finite && isfinite(PL(PL_VNEW + i));
    PA_SYNC();
    if (__builtin_expect(!finite, 0)) {
        for (int i = 0; i < PA_N; ++i) PL(PL_V + i) = PA_DC_OP[i];            // dc_operating_point == DC_OP (never re-set by the adapter)
        for (int i = 0; i < PA_M; ++i) { PL(PL_IP + i) = PA_DC_NL_I[i]; PL(PL_IPP + i) = PA_DC_NL_I[i]; }
        sc->dcx = 0.0; sc->dcy = 0.0;
        sc->nan_cnt += 1ull;
        PA_SYNC();
        return PA_DC_BLOCK_X0;
    }
    for (int i = 0; i < PA_N; ++i) PL(PL_V + i) = PL(PL_VNEW + i);
    for (int i = 0; i < PA_M; ++i) { PL(PL_IPP + i) = PL(PL_IP + i); PL(PL_IP + i) = PL(PL_INL + i); }
    const double raw_out = PL(PL_VNEW + 8);
    PA_SYNC();
    const double dc_blocked = raw_out - sc->dcx + C->dc_block_r * sc->dcy;
    sc->dcx = raw_out;
    sc->dcy = dc_blocked;
    const double scaled = dc_blocked * 1.0;
    const double abs_out = fabs(scaled);
    if (abs_out > sc->peak) sc->peak = abs_out;
    if (abs_out > 3e1) sc->clamp_cnt += 1ull;
    return clampd(scaled, -3e1, 3e1);
}

// state <- settled blob (+ the per-state part of set_sample_rate when the chain does not run at the codegen rate): init_state,
// power_amp.rs:294-302.  Both lanes write the same values.
__device__ inline void pa_init_state(PaScal* __restrict__ sc, double* __restrict__ W, const double* __restrict__ settled, const OwPaConsts* __restrict__ C) {
    for (int i = 0; i < PA_N; ++i) PL(PL_V + i) = settled[PAS_V + i];
    for (int i = 0; i < PA_M; ++i) { PL(PL_IP + i) = settled[PAS_IP + i]; PL(PL_IPP + i) = settled[PAS_IPP + i]; }
    sc->dcx = settled[PAS_DCX]; sc->dcy = settled[PAS_DCY]; sc->peak = settled[PAS_PEAK];
    sc->clamp_cnt = dbits(settled[PAS_CLAMP]); sc->nrmax_cnt = dbits(settled[PAS_NRMAX]); sc->nan_cnt = dbits(settled[PAS_NAN]);
    if (!C->rate_is_codegen) { sc->dcx = 0.0; sc->dcy = 0.0; }
    sc->last_nr = 0u;
    PA_SYNC();
}
__device__ inline void pa_rails_reset(PaScal* __restrict__ sc) { sc->rail_p = 22.5; sc->rail_n = 22.5; sc->iavg_p = 0.0; sc->iavg_n = 0.0; }

// melange_adapter::PowerAmp::process, power_amp.rs:373-431
__device__ inline double pa_process(PaScal* __restrict__ sc, const OwPaConsts* __restrict__ C, double* __restrict__ W, int role, const double* __restrict__ settled,
                                    double input, bool rail_sag) {
    const double off_p = rail_sag ? sc->rail_p - 22.5 : 0.0, off_n = rail_sag ? sc->rail_n - 22.5 : 0.0;
    const double raw = pa_process_sample(sc, C, W, role, input, off_p, off_n);
    const double result = OW_DIV_C(raw, 22.0);
    const bool nr_failed = sc->last_nr >= 69u;
    bool insane = false;
    for (int i = 0; i < PA_N; ++i) { const double v = PL(PL_V + i); insane = insane || !isfinite(v) || fabs(v) > 100.0; }
    PA_SYNC();
    if (__builtin_expect(!isfinite(result) || nr_failed || insane, 0)) {
        pa_init_state(sc, W, settled, C);
        pa_rails_reset(sc);
        sc->guard_cnt += 1ull;
        return sc->last_good;
    }
    const double clamped = clampd(result, -1.0, 1.0);
    sc->last_good = clamped;
    if (rail_sag) {        // RailDynamics::step(raw), power_amp.rs:131-156
        const double i_pos = fmax(OW_DIV_C(raw, 8.0), 0.0);
        const double i_neg = fmax(OW_DIV_C(-raw, 8.0), 0.0);
        sc->iavg_p += C->alpha_i_avg * (i_pos - sc->iavg_p);
        sc->iavg_n += C->alpha_i_avg * (i_neg - sc->iavg_n);
        const double target_pos = 24.5 - sc->iavg_p * 3.5;
        const double target_neg = 24.5 - sc->iavg_n * 3.5;
        const double alpha_p = target_pos < sc->rail_p ? C->alpha_attack : C->alpha_release;
        const double alpha_n = target_neg < sc->rail_n ? C->alpha_attack : C->alpha_release;
        sc->rail_p += alpha_p * (target_pos - sc->rail_p);
        sc->rail_n += alpha_n * (target_neg - sc->rail_n);
    }
    return clamped;
}

// engine e's state rows <-> LDS (both lanes load the same values; one lane stores)
OW_DEV void pa_load(PaScal* __restrict__ sc, double* __restrict__ W, const double* __restrict__ pa, int I, int e) {
    for (int i = 0; i < PA_N; ++i) PL(PL_V + i) = pa[(size_t)(PAS_V + i) * I + e];
    for (int i = 0; i < PA_M; ++i) { PL(PL_IP + i) = pa[(size_t)(PAS_IP + i) * I + e]; PL(PL_IPP + i) = pa[(size_t)(PAS_IPP + i) * I + e]; }
    sc->dcx = pa[(size_t)PAS_DCX * I + e]; sc->dcy = pa[(size_t)PAS_DCY * I + e]; sc->peak = pa[(size_t)PAS_PEAK * I + e];
    sc->clamp_cnt = dbits(pa[(size_t)PAS_CLAMP * I + e]); sc->nrmax_cnt = dbits(pa[(size_t)PAS_NRMAX * I + e]); sc->nan_cnt = dbits(pa[(size_t)PAS_NAN * I + e]);
    sc->last_good = pa[(size_t)PAS_LASTGOOD * I + e];
    sc->rail_p = pa[(size_t)PAS_RAILP * I + e]; sc->rail_n = pa[(size_t)PAS_RAILN * I + e];
    sc->iavg_p = pa[(size_t)PAS_IAVGP * I + e]; sc->iavg_n = pa[(size_t)PAS_IAVGN * I + e];
    sc->guard_cnt = dbits(pa[(size_t)PAS_GUARD * I + e]);
    sc->last_nr = 0u;
    PA_SYNC();
}
OW_DEV void pa_store(const PaScal* __restrict__ sc, const double* __restrict__ W, double* __restrict__ pa, int I, int e) {
    for (int i = 0; i < PA_N; ++i) pa[(size_t)(PAS_V + i) * I + e] = PL(PL_V + i);
    for (int i = 0; i < PA_M; ++i) { pa[(size_t)(PAS_IP + i) * I + e] = PL(PL_IP + i); pa[(size_t)(PAS_IPP + i) * I + e] = PL(PL_IPP + i); }
    pa[(size_t)PAS_DCX * I + e] = sc->dcx; pa[(size_t)PAS_DCY * I + e] = sc->dcy; pa[(size_t)PAS_PEAK * I + e] = sc->peak;
    pa[(size_t)PAS_CLAMP * I + e] = bitsd(sc->clamp_cnt); pa[(size_t)PAS_NRMAX * I + e] = bitsd(sc->nrmax_cnt); pa[(size_t)PAS_NAN * I + e] = bitsd(sc->nan_cnt);
    pa[(size_t)PAS_LASTGOOD * I + e] = sc->last_good;
    pa[(size_t)PAS_RAILP * I + e] = sc->rail_p; pa[(size_t)PAS_RAILN * I + e] = sc->rail_n;
    pa[(size_t)PAS_IAVGP * I + e] = sc->iavg_p; pa[(size_t)PAS_IAVGN * I + e] = sc->iavg_n;
    pa[(size_t)PAS_GUARD * I + e] = bitsd(sc->guard_cnt);
}

// Settled state of the amp (compute_settled_state, power_amp.rs:290-296): CircuitState::default() (DC_OP + 50 warm-up samples) and
// 44 100 silent samples, all with the codegen-rate matrices.  One lane pair; cached per device by the host like the reference's OnceLock.
__global__ __launch_bounds__(64) void k_mpa_settle(const OwPaConsts* __restrict__ C88, double* __restrict__ settled) {
    __shared__ double WS[PL_ROWS * PA_EPW];
    const int el = threadIdx.x & 31, role = threadIdx.x >> 5;
    double* W = WS + el;                                   // all 32 pairs run the same silent settle (no divergence); pair 0 reports
    PaScal sc;
    for (int i = 0; i < PA_N; ++i) PL(PL_V + i) = PA_DC_OP[i];
    for (int i = 0; i < PA_M; ++i) { PL(PL_IP + i) = PA_DC_NL_I[i]; PL(PL_IPP + i) = PA_DC_NL_I[i]; }
    sc.dcx = PA_DC_BLOCK_X0; sc.dcy = 0.0; sc.peak = 0.0; sc.clamp_cnt = sc.nrmax_cnt = sc.nan_cnt = sc.guard_cnt = 0ull;
    sc.last_good = 0.0; sc.last_nr = 0u;
    pa_rails_reset(&sc);
    PA_SYNC();
    for (int n = 0; n < 50 + 44100; ++n) pa_process_sample(&sc, C88, W, role, 0.0, 0.0, 0.0);
    if (threadIdx.x != 0) return;
    for (int i = 0; i < PA_N; ++i) settled[PAS_V + i] = PL(PL_V + i);
    for (int i = 0; i < PA_M; ++i) { settled[PAS_IP + i] = PL(PL_IP + i); settled[PAS_IPP + i] = PL(PL_IPP + i); }
    settled[PAS_DCX] = sc.dcx; settled[PAS_DCY] = sc.dcy; settled[PAS_PEAK] = sc.peak;
    settled[PAS_CLAMP] = bitsd(sc.clamp_cnt); settled[PAS_NRMAX] = bitsd(sc.nrmax_cnt); settled[PAS_NAN] = bitsd(sc.nan_cnt);
}

// The amp alone on given input (debug hook ow_debug_power_amp): pair p of block b processes row b * 32 + p of `in`.  taps [row][n][3]: outer
// Newton iterations of the sample (70 = exhausted), guard resets so far, positive rail after the sample.
__global__ __launch_bounds__(64) void k_mpa_debug(const OwPaConsts* __restrict__ C, const double* __restrict__ settled, const double* __restrict__ in,
                                                  double* __restrict__ out, double* __restrict__ taps, long long n, int n_rows, int rail_sag,
                                                  const long long* __restrict__ poke_at, const int* __restrict__ poke_node, const double* __restrict__ poke_val) {
    __shared__ double WS[PL_ROWS * PA_EPW];
    const int el = threadIdx.x & 31, role = threadIdx.x >> 5;
    double* W = WS + el;
    const int row_raw = blockIdx.x * PA_EPW + el;
    const bool valid = row_raw < n_rows;
    const size_t row = valid ? row_raw : n_rows - 1;
    PaScal sc;
    pa_init_state(&sc, W, settled, C);
    pa_rails_reset(&sc);
    sc.last_good = 0.0; sc.guard_cnt = 0ull;
    for (long long i = 0; i < n; ++i) {
        if (poke_at && poke_at[row] == i) { PL(PL_V + poke_node[row]) = poke_val[row]; }
        PA_SYNC();
        const double y = pa_process(&sc, C, W, role, settled, in[row * n + i], rail_sag != 0);
        if (valid && role == 0) {
            out[row * n + i] = y;
            if (taps) {
                double* t = taps + (row * n + i) * 3;
                t[0] = (double)sc.last_nr; t[1] = (double)sc.guard_cnt; t[2] = sc.rail_p;
            }
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

// Output stage with the melange power amp: a lane pair per engine (the amp is a stateful recurrence at the chain rate, so the two
// chain-rate samples of an output sample are solved one after the other), then half-band down, speaker, gain, f32 as k_post -- that
// cheap tail is computed by both lanes of the pair, lane 0 of the pair owns state and output.  One wavefront per workgroup.
__global__ __launch_bounds__(64) void k_post_mpa(const OwConsts* __restrict__ K, const OwPaConsts* __restrict__ C, const double* __restrict__ settled,
                                                 double* __restrict__ cs, double* __restrict__ pa, const OwEngineArgs* __restrict__ args,
                                                 OwEngineOut* __restrict__ eout, const double* __restrict__ pre, float* __restrict__ out,
                                                 double* __restrict__ pa_tap, int I, int L, int Lout, int e0, int ne) {
    __shared__ double WS[PL_ROWS * PA_EPW];                      // 128 KB
    __shared__ float tile[PA_EPW * (OW_OCHUNK + 1)];
    const int lane = threadIdx.x;
    const int el = lane & 31, role = lane >> 5;
    double* W = WS + el;
    const int eb = e0 + blockIdx.x * PA_EPW;
    const int e_raw = eb + el;
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
    PaScal sc;
    pa_load(&sc, W, pa, I, e);
    bool nan_fired = false;
    for (int base = 0; base < L; base += OW_OCHUNK) {
        const int cn = min(OW_OCHUNK, L - base);
        for (int n = 0; n < cn; ++n) {
            double y[2] = {0.0, 0.0};
            for (int j = 0; j < osr; ++j) {
                const size_t idx = (size_t)(base + n) * osr + j;
                y[j] = pa_process(&sc, C, W, role, settled, pre[idx * I + e] * 0.25, rail_sag);     // x FIXED_CIRCUIT_DRIVE, engine.rs:544-546
                if (pa_tap && valid && role == 0) pa_tap[idx * I + e] = y[j];
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
                pa_init_state(&sc, W, settled, C);
                pa_rails_reset(&sc);
                nan_fired = true;
            }
            if (role == 0) tile[el * (OW_OCHUNK + 1) + n] = f;
        }
        __syncthreads();
        for (int r = 0; r < PA_EPW; ++r) {
            const int er = eb + r;
            if (er < e0 + ne && lane < cn) out[(size_t)er * Lout + base + lane] = tile[r * (OW_OCHUNK + 1) + lane];
        }
        __syncthreads();
    }
    if (!valid || role != 0) return;
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
    pa_store(&sc, W, pa, I, e);
}

}  // namespace owdev
