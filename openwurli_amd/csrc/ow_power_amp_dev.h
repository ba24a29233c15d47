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
// development counters (-DOW_DBG_COUNTERS, tools/probe_power_amp_waves.py): how often a WAVEFRONT executes the pieces of the solver
// [2] trips of k_post_mpa's main loop  [3] device evaluations  [4] passes of the backward-Euler retry  [5] limited pnjlim calls (lanes)
// [6] sample completions  [7] passes of any kind (pa_newton_pass bodies)
#ifdef OW_DBG_COUNTERS
#define OW_DBG_WAVE(i) do { const unsigned long long act_ = __builtin_amdgcn_ballot_w64(true); \
                            if ((int)(threadIdx.x & 63) == __builtin_ctzll(act_)) atomicAdd(&g_ow_dbg[i], 1ull); } while (0)
#else
#define OW_DBG_WAVE(i) ((void)0)
#endif

// gen_power_amp.rs:7870-8017, Gummel-Poon branch (all eight devices of the netlist set USE_GP; the host refuses anything else)
__device__ inline PaBjt pa_bjt_evaluate(double vbe, double vbc, const OwPaConsts::Dev& D) {
    const double vbe_eff = D.sign * vbe, vbc_eff = D.sign * vbc;
    const double exp_be = fast_exp(ow_div_const(vbe_eff, D.nf_vt, D.r_nf_vt));
    const double exp_bc = fast_exp(ow_div_const(vbc_eff, D.nr_vt, D.r_nr_vt));
    const bool has_ise = D.ise > 0.0, has_isc = D.isc > 0.0;
    const double exp_be_leak = has_ise ? fast_exp(ow_div_const(vbe_eff, D.ne_vt, D.r_ne_vt)) : 0.0;
    const double exp_bc_leak = has_isc ? fast_exp(ow_div_const(vbc_eff, D.nc_vt, D.r_nc_vt)) : 0.0;
    const double i_cc = D.is * (exp_be - exp_bc);
    const double ib_fwd = D.is_bf * (exp_be - 1.0);
    const double ib_rev = D.is_br * (exp_bc - 1.0);
    const double ib_leak_be = has_ise ? D.ise * (exp_be_leak - 1.0) : 0.0;
    const double ib_leak_bc = has_isc ? D.isc * (exp_bc_leak - 1.0) : 0.0;
    const double dib_fwd_dvbe = D.c_dib_fwd * exp_be;
    const double dib_rev_dvbc = D.c_dib_rev * exp_bc;
    const double dib_leak_dvbe = has_ise ? D.c_leak_be * exp_be_leak : 0.0;
    const double dib_leak_dvbc = has_isc ? D.c_leak_bc * exp_bc_leak : 0.0;
    const double q1_denom = 1.0 - ow_div_const(vbe_eff, D.var, D.r_var) - ow_div_const(vbc_eff, D.vaf, D.r_vaf);
    double q1 = 1.0, dq1_dvbe = 0.0, dq1_dvbc = 0.0;
    if (!(q1_denom <= 0.0 || fabs(q1_denom) < 1e-30)) {
        q1 = ow_div(1.0, q1_denom);
        dq1_dvbe = ow_div_const(q1 * q1, D.var, D.r_var);
        dq1_dvbc = ow_div_const(q1 * q1, D.vaf, D.r_vaf);
    }
    const double cbe = D.is * (exp_be - 1.0);
    const double cbc = D.is * (exp_bc - 1.0);
    const double q2 = ow_div_const(cbe, D.ikf, D.r_ikf) + ow_div_const(cbc, D.ikr, D.r_ikr);
    const double dq2_dvbe = D.c_dq2_be * exp_be;
    const double dq2_dvbc = D.c_dq2_bc * exp_bc;
    const double disc = fmax(1.0 + 4.0 * q2, 0.0);
    const double dd = sqrt(disc);
    const double y_dd = ow_rcp_refined(dd);                        // two quotients over dd, two over qb2: one refined reciprocal each
    const double dd_dvbe = dd > 1e-15 ? ow_div_y(2.0 * dq2_dvbe, dd, y_dd) : 0.0;
    const double dd_dvbc = dd > 1e-15 ? ow_div_y(2.0 * dq2_dvbc, dd, y_dd) : 0.0;
    const double qb = q1 * (1.0 + dd) * 0.5;                       // x / 2.0 == x * 0.5 exactly
    const double dqb_dvbe = dq1_dvbe * (1.0 + dd) * 0.5 + q1 * dd_dvbe * 0.5;
    const double dqb_dvbc = dq1_dvbc * (1.0 + dd) * 0.5 + q1 * dd_dvbc * 0.5;
    PaBjt r;
    r.ic = D.sign * (ow_div(i_cc, qb) - D.is_br * (exp_bc - 1.0));
    r.ib = D.sign * (ib_fwd + ib_rev + ib_leak_be + ib_leak_bc);
    const double dicc_dvbe = D.c_dicc_be * exp_be;
    const double dicc_dvbc = D.c_dicc_bc * exp_bc;
    const double qb2 = fmax(qb * qb, 1e-30);
    const double y_qb2 = ow_rcp_refined(qb2);
    const double quotient_dvbe = ow_div_y(dicc_dvbe * qb - i_cc * dqb_dvbe, qb2, y_qb2);
    const double quotient_dvbc = ow_div_y(dicc_dvbc * qb - i_cc * dqb_dvbc, qb2, y_qb2);
    const double d_bc_term_dvbc = D.c_dib_rev * exp_bc;
    r.j0 = quotient_dvbe;
    r.j1 = quotient_dvbc - d_bc_term_dvbc;
    r.j2 = dib_fwd_dvbe + dib_leak_dvbe;
    r.j3 = dib_rev_dvbc + dib_leak_dvbc;
    return r;
}

// gen_power_amp.rs:8032-8145
__device__ inline PaBjt pa_bjt_with_parasitics(double vbe_ext, double vbc_ext, const OwPaConsts::Dev* __restrict__ Dp) {
    const OwPaConsts::Dev& D = *Dp;
    double vbe_int = vbe_ext, vbc_int = vbc_ext;
    // The reference evaluates the device once more after its inner loop (:8113); when the loop ended at the convergence test that is the
    // evaluation the test itself just made, at the same voltages: it is kept (`held`) and only a lane that left the loop another way --
    // iteration limit, singular 2x2 -- evaluates again.  (All lanes of the wavefront converging is the normal case.)
    PaBjt held;
    bool fresh = false;
    for (int it = 0; it < 15; ++it) {
        OW_DBG_WAVE(3);
        const PaBjt e = pa_bjt_evaluate(vbe_int, vbc_int, D);
        held = e;
        const double f1 = vbe_int - vbe_ext + e.ib * D.rb + (e.ic + e.ib) * D.re;
        const double f2 = vbc_int - vbc_ext + e.ib * D.rb - e.ic * D.rc;
#ifdef OW_PA_UNIFORM_INNER
        // experiment (profiles/r06_mpa_experiments.md): the lanes' loops as one wave-uniform loop, a finished lane holds its voltages
        // (its re-evaluation is at the same voltages: the same `held`, the same f1/f2)
        fresh = fresh || (fabs(f1) < 1e-10 && fabs(f2) < 1e-10);
        if (__builtin_amdgcn_ballot_w64(!fresh) == 0ull) break;
        const double j11 = 1.0 + held.j2 * D.rb + (held.j0 + held.j2) * D.re;
        const double j12 = held.j3 * D.rb + (held.j1 + held.j3) * D.re;
        const double j21 = held.j2 * D.rb - held.j0 * D.rc;
        const double j22 = 1.0 + held.j3 * D.rb - held.j1 * D.rc;
        const double det = j11 * j22 - j12 * j21;
        fresh = fresh || fabs(det) < 1e-30;
        const double inv_det = ow_div(1.0, det);
        double dvbe = (j22 * f1 - j12 * f2) * inv_det;
        double dvbc = (j11 * f2 - j21 * f1) * inv_det;
        dvbe = clampd(dvbe, -D.max_step, D.max_step);
        dvbc = clampd(dvbc, -D.max_step, D.max_step);
        vbe_int = fresh ? vbe_int : vbe_int - dvbe;
        vbc_int = fresh ? vbc_int : vbc_int - dvbc;
    }
#else
        if (fabs(f1) < 1e-10 && fabs(f2) < 1e-10) { fresh = true; break; }
        const double j11 = 1.0 + e.j2 * D.rb + (e.j0 + e.j2) * D.re;
        const double j12 = e.j3 * D.rb + (e.j1 + e.j3) * D.re;
        const double j21 = e.j2 * D.rb - e.j0 * D.rc;
        const double j22 = 1.0 + e.j3 * D.rb - e.j1 * D.rc;
        const double det = j11 * j22 - j12 * j21;
        if (fabs(det) < 1e-30) { fresh = true; break; }            // voltages unchanged since `held` was evaluated
        const double inv_det = ow_div(1.0, det);
        double dvbe = (j22 * f1 - j12 * f2) * inv_det;
        double dvbc = (j11 * f2 - j21 * f1) * inv_det;
        dvbe = clampd(dvbe, -D.max_step, D.max_step);
        dvbc = clampd(dvbc, -D.max_step, D.max_step);
        vbe_int -= dvbe;
        vbc_int -= dvbc;
    }
#endif
    PaBjt e = held;
    if (__builtin_amdgcn_ballot_w64(!fresh) != 0ull) {
        OW_DBG_WAVE(3);
        const PaBjt e2 = pa_bjt_evaluate(vbe_int, vbc_int, D);
        if (!fresh) e = e2;
    }
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
// Mapping.  EIGHT LANES PER ENGINE -- one per transistor -- and eight engines per wavefront (lane = role * 8 + engine slot); four
// wavefronts per workgroup, two workgroups per CU (LDS), so every SIMD holds two wavefronts.  The lanes of an engine sit in ONE
// wavefront: they run in lock step, "synchronisation" between them is a compiler fence, not a barrier.  The eight ENGINES of a
// wavefront are not in lock step: k_post_mpa / k_mpa_debug run one Newton pass per loop trip for every engine, each engine on its own
// sample counter, so a sample that takes 60 passes on one engine does not make the other seven wait for it (see k_post_mpa).
//   lane `role` owns transistor `role`: its two controlling voltages, the device model (the expensive part: the exponentials and the
//   inner 2x2 parasitic solve), and ROWS 2 role, 2 role + 1 of the 16x16 Newton Jacobian -- in REGISTERS, with their right-hand sides;
//   Gaussian elimination with partial pivoting, column by column: every lane posts |J[row][col]| of its un-pivoted rows to LDS at
//   their logical positions, all eight pick the pivot (the reference's scan order), the owner posts the pivot row, every lane
//   eliminates its own rows.  Back substitution: the owner of row i computes x_i (ascending-j sum, as the
//   reference) and posts it.  No Jacobian in LDS at all; the K rows a lane needs come from a 2 KB LDS copy of K.
//   Matrix-vector products (S rhs, S_NI i_nl, K i_trial) are split by rows over the eight lanes; O(16) vector passes (step limiting,
//   convergence test, finiteness) are done by all eight on the same values.
// LDS per engine: 217 doubles (vectors of the sweep + node / port state), engine-minor with a row stride of 9 doubles inside the
// wavefront's slab (8 engines + 1 pad: the role-split accesses of consecutive rows fall in different banks).
// (History: private arrays -- 8.4 KB scratch per lane -- 6.2 s per 512-sample block at 16 384 engines; lane pair per engine with the
// Jacobian in LDS, one wavefront per CU: 3.3 s.)
// Every sum keeps the reference's operand order and the library is built without FMA contraction: the solver follows the CPU
// restatement bit for bit (pnjlim's logarithm excepted), which keeps the divergence guard firing on the same sample on both sides.
#define PA_EPW 8              // engines per wavefront
#ifndef PA_WPB
#define PA_WPB 4              // wavefronts per workgroup
#endif
#ifndef PA_MIN_BLOCKS
#define PA_MIN_BLOCKS 2       // workgroups per CU the register budget is set for (__launch_bounds__)
#endif
#define PA_EPB (PA_EPW * PA_WPB)
#define PA_LS 9               // LDS row stride in doubles
enum {
    // vectors of one Newton iteration
    PL_CAND = 0,              // [16] |J[row][col]| of the rows not yet used as pivots, by physical row
    PL_PROW = 16,             // [17] the pivot row (entries col..15) and its right-hand side at [16]
    PL_X = 33,                // [16] solution of the linear system
    PL_VD = 49, PL_T0 = 65 /* dv_trial / dv */, PL_T1 = 81 /* v_lim / alpha */, PL_T2 = 97 /* i_trial */,
    PL_NT_END = 113,
    PL_RHS = 0,               // [20] lives before the sweep only
    PL_VNEW = 49,             // [20] lives after the sweep only
    // carried through the sample / between samples
    PL_INL = 113, PL_P = 129, PL_VPRED = 145, PL_V = 165, PL_IP = 185, PL_IPP = 201,
    PL_ROWS = 217
};
#define PL(r) W[(r) * PA_LS]
#define PA_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#define PA_LDS_DOUBLES (PA_WPB * PL_ROWS * PA_LS)

// The workgroup's LDS copy of the tables its lanes index PER LANE (row = f(role)): from the constant block those would be vector loads
// the compiler hoists out of the sample loop (loop-invariant, ~400 registers: spills); LDS reads stay where they are used.
struct PaTab {
    double k[PA_M * PA_M];
    double s[PA_N * PA_N];
    double s_ni[PA_N * PA_M];
    OwPaConsts::Dev dev[8];
    double nz_coef[PA_RHS_NNZ];          // A_neg's non-zeros in the reference's order (sorted by row), as CSR
    uint8_t nz_col[PA_RHS_NNZ];          // (bytes: with the two tables as ints the workgroup's LDS passed 80 KB and only one fit a CU)
    uint8_t row_ptr[PA_N + 1];
};
__device__ __forceinline__ void pa_stage_tables(PaTab* __restrict__ T, const OwPaConsts* __restrict__ C) {
    for (int i = threadIdx.x; i < PA_M * PA_M; i += blockDim.x) T->k[i] = (&C->k[0][0])[i];
    for (int i = threadIdx.x; i < PA_N * PA_N; i += blockDim.x) T->s[i] = (&C->s[0][0])[i];
    for (int i = threadIdx.x; i < PA_N * PA_M; i += blockDim.x) T->s_ni[i] = (&C->s_ni[0][0])[i];
    double* d = reinterpret_cast<double*>(&T->dev[0]);
    const double* g = reinterpret_cast<const double*>(&C->dev[0]);
    for (int i = threadIdx.x; i < (int)(8 * sizeof(OwPaConsts::Dev) / sizeof(double)); i += blockDim.x) d[i] = g[i];
    for (int q = threadIdx.x; q < PA_RHS_NNZ; q += blockDim.x) {
        const int i = (int)PA_RHS_NZ_ROW[q], j = (int)PA_RHS_NZ_COL[q];
        T->nz_coef[q] = C->a_neg[i][j];
        T->nz_col[q] = (uint8_t)j;
    }
    for (int r = threadIdx.x; r <= PA_N; r += blockDim.x) {
        int c = 0;
        for (int q = 0; q < PA_RHS_NNZ; ++q) c += ((int)PA_RHS_NZ_ROW[q] < r) ? 1 : 0;
        T->row_ptr[r] = (uint8_t)c;
    }
    __syncthreads();
}

struct PaScal {      // per-engine scalars, identical in the eight lanes of the engine
    double dcx, dcy, peak, last_good, rail_p, rail_n, iavg_p, iavg_n;
    unsigned long long clamp_cnt, nrmax_cnt, nan_cnt, guard_cnt;
    uint32_t last_nr;
};

// ONE pass of the Newton solve (main sweep with K, or the BE retry with K_be): gen_power_amp.rs:8956-10680 / 10745-12245, the body of
// the `for iter in 0..MAX_ITER` loop.  p in PL_P, i_nl in PL_INL (updated).  KM: the 16x16 K of this sweep (the LDS copy for the main
// sweep).  Returns true when the reference's loop would `break` converged after this pass.  Nothing is carried from one pass to the
// next outside LDS, so the engines of a wavefront need not be in the same pass, nor in the same sample (k_post_mpa below).
template <bool BE>
__device__ __forceinline__ bool pa_newton_pass(double* __restrict__ W, const double* KM, const OwPaConsts::Dev* D, int role) {
    const int r0 = 2 * role, r1 = 2 * role + 1;
    const double vt = D->vt, vcrit = D->vcrit;
    OW_DBG_WAVE(7);
    if (BE) OW_DBG_WAVE(4);
    {
        double vd0 = PL(PL_P + r0), vd1 = PL(PL_P + r1);
#pragma unroll
        for (int j = 0; j < PA_M; ++j) {
            const double inl = PL(PL_INL + j);
            vd0 = vd0 + KM[r0 * PA_M + j] * inl;
            vd1 = vd1 + KM[r1 * PA_M + j] * inl;
        }
        const PaBjt e = pa_bjt_with_parasitics(vd0, vd1, D);
        const double f0 = PL(PL_INL + r0) - e.ic, f1 = PL(PL_INL + r1) - e.ib;
        PL(PL_VD + r0) = vd0; PL(PL_VD + r1) = vd1;
        double A[PA_M], B[PA_M], bA = f0, bB = f1;                 // my two rows: J[i][j] = delta_ij - jdev[i][2d] K[2d][j] - jdev[i][2d+1] K[2d+1][j]
#pragma unroll
        for (int j = 0; j < PA_M; ++j) {
            const double k0 = KM[r0 * PA_M + j], k1 = KM[r1 * PA_M + j];
            A[j] = (j == r0 ? 1.0 : 0.0) - e.j0 * k0 - e.j1 * k1;
            B[j] = (j == r1 ? 1.0 : 0.0) - e.j2 * k0 - e.j3 * k1;
        }
        // Gaussian elimination, one ROLLED loop over the columns (the unrolled form is 7000 instructions: with four wavefronts at
        // different places in it the 64 KB instruction cache two CUs share thrashes, 5x slower).  The register rows ROTATE: after
        // column c a row that is still being eliminated holds J[row][c+1 + j] at index j, so every column works on index 0 and the
        // shift rides on the update itself (A[j-1] = A[j] - f prow[j]).  A row chosen as pivot at column c is frozen from then on:
        // it keeps U[c][c + j] at index j for the back substitution.  Entries past the live width are stale and never read.
        // Row exchanges move no data either: every lane keeps the LOGICAL positions of its two rows (identity at the start), a pivot
        // choice swaps two positions, candidates are posted at their logical position (so the scan below reads fixed LDS offsets in
        // the reference's order), and the owner of logical row i is the lane that holds position i.
        int posA = r0, posB = r1;
        bool singular = false, usedA = false, usedB = false;
        // Four rolled loops of four columns: after column c only entries 1 .. 15 - c of a live row matter, so the tier that starts at
        // column c0 updates (and fetches from the pivot row) WIDTH = 15 - c0 entries instead of all 15 -- the same operations on the
        // same values, minus the ones on stale entries past the live width.  The two quotients over a column's pivot share its refined
        // reciprocal (ow_div_y == ow_div instruction for instruction).
#define PA_ELIM_TIER(C0, C1, WIDTH)                                                                                          \
        _Pragma("unroll 1") for (int col = (C0); col < (C1) && !singular; ++col) {                                              \
            /* the candidates |J[row][col]| were posted at the end of the previous column (before its closing fence) */       \
            /* the reference's scan: first strict maximum over logical rows col..15; all candidates are fetched at once */     \
            int max_row = col;                                                                                                  \
            double max_val = PL(PL_CAND + col);                                                                                 \
            double cand[PA_M];                                                                                                  \
            _Pragma("unroll") for (int row = (C0) + 1; row < PA_M; ++row) cand[row] = PL(PL_CAND + row);                         \
            _Pragma("unroll") for (int row = (C0) + 1; row < PA_M; ++row) {                                                      \
                if (row > col) {                    /* wave-uniform: col is the loop counter */                                \
                    const bool take = cand[row] > max_val;                                                                      \
                    max_val = take ? cand[row] : max_val;                                                                       \
                    max_row = take ? row : max_row;                                                                             \
                }                                                                                                               \
            }                                                                                                                   \
            if (max_val < 1e-15) { singular = true; break; }                                                                    \
            /* exchange logical positions col <-> max_row (a no-op when they are equal) */                                     \
            posA = posA == col ? max_row : (posA == max_row ? col : posA);                                                      \
            posB = posB == col ? max_row : (posB == max_row ? col : posB);                                                      \
            if (posA == col) {                                                                                                  \
                _Pragma("unroll") for (int j = 0; j <= (WIDTH); ++j) PL(PL_PROW + j) = A[j];                                      \
                PL(PL_PROW + 16) = bA;                                                                                          \
                usedA = true;                                                                                                   \
            }                                                                                                                   \
            if (posB == col) {                                                                                                  \
                _Pragma("unroll") for (int j = 0; j <= (WIDTH); ++j) PL(PL_PROW + j) = B[j];                                      \
                PL(PL_PROW + 16) = bB;                                                                                          \
                usedB = true;                                                                                                   \
            }                                                                                                                   \
            PA_SYNC();                                                                                                          \
            const double pivot = PL(PL_PROW), bcol = PL(PL_PROW + 16);                                                          \
            const double ypiv = ow_rcp_refined(pivot);                                                                          \
            double prow[PA_M];                                                                                                  \
            _Pragma("unroll") for (int j = 1; j <= (WIDTH); ++j) prow[j] = PL(PL_PROW + j);                                       \
            if (!usedA) {                                                                                                       \
                const double factor = ow_div_y(A[0], pivot, ypiv);                                                              \
                _Pragma("unroll") for (int j = 1; j <= (WIDTH); ++j) A[j - 1] = A[j] - factor * prow[j];                          \
                bA -= factor * bcol;                                                                                            \
            }                                                                                                                   \
            if (!usedB) {                                                                                                       \
                const double factor = ow_div_y(B[0], pivot, ypiv);                                                              \
                _Pragma("unroll") for (int j = 1; j <= (WIDTH); ++j) B[j - 1] = B[j] - factor * prow[j];                          \
                bB -= factor * bcol;                                                                                            \
            }                                                                                                                   \
            /* candidates of the next column, by logical position (rows used as pivots sit at positions <= col: never looked at again) */ \
            PL(PL_CAND + posA) = fabs(A[0]);                                                                                    \
            PL(PL_CAND + posB) = fabs(B[0]);                                                                                    \
            PA_SYNC();                                                                                                          \
        }
        PL(PL_CAND + posA) = fabs(A[0]);
        PL(PL_CAND + posB) = fabs(B[0]);
        PA_SYNC();
#ifdef OW_PA_TIERS8      // experiment (round 6, profiles/r06_mpa_experiments.md): eight tiers of two columns -- 7 % fewer update operations, twice the code
        PA_ELIM_TIER(0, 2, 15)
        PA_ELIM_TIER(2, 4, 13)
        PA_ELIM_TIER(4, 6, 11)
        PA_ELIM_TIER(6, 8, 9)
        PA_ELIM_TIER(8, 10, 7)
        PA_ELIM_TIER(10, 12, 5)
        PA_ELIM_TIER(12, 14, 3)
        PA_ELIM_TIER(14, 16, 1)
#else
        PA_ELIM_TIER(0, 4, 15)
        PA_ELIM_TIER(4, 8, 11)
        PA_ELIM_TIER(8, 12, 7)
        PA_ELIM_TIER(12, 16, 3)
#endif
#undef PA_ELIM_TIER
        if (singular) {
            const double i0 = PL(PL_INL + r0), i1 = PL(PL_INL + r1);
            const double c0 = BE ? 0.01 : fmax(fabs(i0) * 0.1, 0.01), c1 = BE ? 0.01 : fmax(fabs(i1) * 0.1, 0.01);
            PA_SYNC();
            PL(PL_INL + r0) = i0 - clampd(f0 * 0.5, -c0, c0);
            PL(PL_INL + r1) = i1 - clampd(f1 * 0.5, -c1, c1);
            PA_SYNC();
            return false;
        }
        // back substitution: x_i = (b_i - sum_{j>i} U_ij x_j) / U_ii, j ascending, by the owner of logical row i (its frozen register
        // row holds U_ij at index j - i); U_ii is the pivot of column i, so it passed the 1e-15 test above (the reference's second
        // test on it can never fire)
#pragma unroll
        for (int i = PA_M - 1; i >= 0; --i) {
            const bool slot = posB == i;
            double sum = slot ? bB : bA;
#pragma unroll
            for (int j = i + 1; j < PA_M; ++j) sum -= (slot ? B[j - i] : A[j - i]) * PL(PL_X + j);
            const double aii = slot ? B[0] : A[0];
            if (slot || posA == i) PL(PL_X + i) = ow_div(sum, aii);
            PA_SYNC();
        }
        bool converged = false;
        const double x0 = PL(PL_X + r0), x1 = PL(PL_X + r1);
        if (!BE) {
            PL(PL_T2 + r0) = PL(PL_INL + r0) - x0;                                              // i_trial
            PL(PL_T2 + r1) = PL(PL_INL + r1) - x1;
            PA_SYNC();
            double a0 = PL(PL_P + r0), a1 = PL(PL_P + r1);
#pragma unroll
            for (int j = 0; j < PA_M; ++j) {
                const double it = PL(PL_T2 + j);
                a0 = a0 + KM[r0 * PA_M + j] * it;
                a1 = a1 + KM[r1 * PA_M + j] * it;
            }
            const double dvt0 = a0 - vd0, dvt1 = a1 - vd1;
            PL(PL_T0 + r0) = dvt0; PL(PL_T0 + r1) = dvt1;
            const double vl0 = fabs(dvt0) > 1e-4 ? pa_pnjlim(a0, vd0, vt, vcrit) : a0;
            const double vl1 = fabs(dvt1) > 1e-4 ? pa_pnjlim(a1, vd1, vt, vcrit) : a1;
            // Step scale ga = min over the ports, in port order, of the limiter ratios dv_lim / dv_trial below ga (gen_power_amp.rs:9849-9960).
            // A port the limiter left alone has v_lim == v_trial: dv_lim is dv_trial bit for bit, the ratio is x / x == 1 and never below ga.
            // So when no port of the wavefront was limited (every sample but start-up and hard clipping) ga stays 1 without a division;
            // otherwise each lane forms the ratios of its own two ports and all of them take the running minimum in port order.
            bool any_limited = false;
            double ga = 1.0;
            if (__builtin_amdgcn_ballot_w64(!(vl0 == a0) || !(vl1 == a1)) != 0ull) {
                double ra = 1.0, rb = 1.0;                                                       // 1.0: "takes no part" (1 < ga is never true)
                {
                    const double dl0 = vl0 - vd0, dl1 = vl1 - vd1;
                    if (fabs(dvt0) > 1e-15) ra = dvt0 * dl0 < 0.0 ? 0.0 : clampd(ow_div(dl0, dvt0), 0.0, 1.0);
                    if (fabs(dvt1) > 1e-15) rb = dvt1 * dl1 < 0.0 ? 0.0 : clampd(ow_div(dl1, dvt1), 0.0, 1.0);
                }
                PA_SYNC();                                                                       // everybody has read i_trial from PL_T2
                PL(PL_T2 + r0) = ra; PL(PL_T2 + r1) = rb;
                PA_SYNC();
                for (int i = 0; i < PA_M; ++i) {
                    const double r = PL(PL_T2 + i);
                    if (r < ga) { ga = r; any_limited = true; }
                }
            }
            PA_SYNC();
            double max_dv = fabs(PL(PL_T0) * ga);
            for (int i = 1; i < PA_M; ++i) max_dv = fmax(max_dv, fabs(PL(PL_T0 + i) * ga));
            if (max_dv > 3.5) { ga *= fmax(ow_div(3.5, max_dv), 0.1); any_limited = true; }
            if (!any_limited) {
                bool conv = true;
                for (int i = 0; i < PA_M; ++i) {
                    const double dv = PL(PL_T0 + i) * ga, vdi = PL(PL_VD + i);
                    const double thr = 1e-3 * fmax(fabs(vdi), fabs(vdi + dv)) + 1e-6;
                    if (fabs(dv) > thr) conv = false;
                }
                converged = conv;
            }
            const double n0 = PL(PL_INL + r0) - ga * x0, n1 = PL(PL_INL + r1) - ga * x1;
            PA_SYNC();
            PL(PL_INL + r0) = n0; PL(PL_INL + r1) = n1;
        } else {
            double a0 = KM[r0 * PA_M] * PL(PL_X), a1 = KM[r1 * PA_M] * PL(PL_X);
#pragma unroll
            for (int j = 1; j < PA_M; ++j) {
                const double xj = PL(PL_X + j);
                a0 = a0 + KM[r0 * PA_M + j] * xj;
                a1 = a1 + KM[r1 * PA_M + j] * xj;
            }
            const double dv0 = -a0, dv1 = -a1;
            double al0 = 1.0, al1 = 1.0;
            bool lim = false;
            if (fabs(dv0) > 1e-4) {
                const double vl = pa_pnjlim(vd0 + dv0, vd0, vt, vcrit);
                const double ratio = fmax(ow_div(vl - vd0, dv0), 0.01);
                if (ratio < al0) { al0 = ratio; if (ratio < 1.0) lim = true; }
            }
            if (fabs(dv1) > 1e-4) {
                const double vl = pa_pnjlim(vd1 + dv1, vd1, vt, vcrit);
                const double ratio = fmax(ow_div(vl - vd1, dv1), 0.01);
                if (ratio < al1) { al1 = ratio; if (ratio < 1.0) lim = true; }
            }
            const double am = fmin(al0, al1);                                                    // one step length per transistor
            PL(PL_T0 + r0) = dv0; PL(PL_T0 + r1) = dv1;
            PL(PL_T1 + r0) = am; PL(PL_T1 + r1) = am;
            PL(PL_T2 + r0) = lim ? 1.0 : 0.0;
            PA_SYNC();
            bool any_limited = false;
            for (int d = 0; d < 8; ++d) any_limited = any_limited || PL(PL_T2 + 2 * d) != 0.0;
            double max_dv = fabs(PL(PL_T0) * PL(PL_T1));
            for (int i = 1; i < PA_M; ++i) max_dv = fmax(max_dv, fabs(PL(PL_T0 + i) * PL(PL_T1 + i)));
            double fac = 1.0;
            const bool scale = max_dv > 3.5;
            if (scale) fac = fmax(ow_div(3.5, max_dv), 0.1);
            if (!any_limited) {
                bool conv = true;
                for (int i = 0; i < PA_M; ++i) {
                    const double al = scale ? PL(PL_T1 + i) * fac : PL(PL_T1 + i);
                    const double stp = PL(PL_T0 + i) * al, vdi = PL(PL_VD + i);
                    const double thr = 1e-3 * fmax(fabs(vdi), fabs(vdi + stp)) + 1e-6;
                    if (fabs(stp) > thr) conv = false;
                }
                converged = conv;
            }
            const double amf = scale ? am * fac : am;
            const double n0 = PL(PL_INL + r0) - amf * x0, n1 = PL(PL_INL + r1) - amf * x1;
            PA_SYNC();
            PL(PL_INL + r0) = n0; PL(PL_INL + r1) = n1;
        }
        PA_SYNC();
        return converged;
    }
}
// The whole sweep in lock step (the BE retry, the settle and debug kernels): iterations spent, 70 = exhausted
template <bool BE>
__device__ __forceinline__ uint32_t pa_newton_body(double* __restrict__ W, const double* KM, const OwPaConsts::Dev* D, int role) {
    _Pragma("unroll 1") for (int iter = 0; iter < 70; ++iter)
        if (pa_newton_pass<BE>(W, KM, D, role)) return (uint32_t)iter;
    return 70u;
}
// The rare paths.  Inlined naively, their table loads are loop-invariant, get hoisted out of the sample loop and sit in hundreds of
// registers for nothing; a real function for the backward-Euler retry (it contains a whole Newton sweep) would set the register
// count of every kernel that can call it.  So: the retry is inlined behind opaque table addresses, the two small resets are real
// functions working on the LDS state and returning scalars by value.
// (pa_opaque: the address comes out of an empty asm, so the loads behind it cannot be speculated out of the rare branch they are in)
template <typename T> __device__ __forceinline__ T* pa_opaque(T* p) { asm volatile("" : "+s"(p)); return p; }
__device__ __forceinline__ uint32_t pa_be_retry(const OwPaConsts* __restrict__ C_in, double* __restrict__ W, int role, double input) {
    const OwPaConsts* C = pa_opaque(C_in);
    const double* rhs_be = pa_opaque(&PA_RHS_CONST_BE[0]);
    const double* n_i = pa_opaque(&PA_N_I[0][0]);
    const double* n_v = pa_opaque(&PA_N_V[0][0]);
    for (int i = role; i < PA_N; i += 8) {
        double sum = rhs_be[i];
        for (int j = 0; j < PA_N; ++j) sum += C->a_neg_be[i][j] * PL(PL_V + j);
        for (int j = 0; j < PA_M; ++j) sum += n_i[i * PA_M + j] * PL(PL_IP + j);
        if (i == 0) sum += input * (1.0 / PA_INPUT_RESISTANCE);
        PL(PL_RHS + i) = sum;
    }
    PA_SYNC();
    for (int i = role; i < PA_N; i += 8) {
        double sum = 0.0;
        for (int j = 0; j < PA_N; ++j) sum += C->s_be[i][j] * PL(PL_RHS + j);
        PL(PL_VPRED + i) = sum;
    }
    PA_SYNC();
    for (int i = role; i < PA_M; i += 8) {
        double sum = 0.0;
        for (int j = 0; j < PA_N; ++j) sum += n_v[i * PA_N + j] * PL(PL_VPRED + j);
        PL(PL_P + i) = sum;
        PL(PL_INL + i) = 2.0 * PL(PL_IP + i) - PL(PL_IPP + i);
    }
    PA_SYNC();
    const uint32_t nr = pa_newton_body<true>(W, &C->k_be[0][0], &C->dev[role], role);
    for (int i = role; i < PA_N; i += 8) {
        double x = PL(PL_VPRED + i);
        for (int j = 0; j < PA_M; ++j) x += C->s_ni_be[i][j] * PL(PL_INL + j);
        PL(PL_VNEW + i) = x;
    }
    PA_SYNC();
    return nr;
}
__device__ __noinline__ void pa_state_to_dc(double* __restrict__ W, int role) {     // the NaN reset of process_sample
    PA_SYNC();
    for (int i = role; i < PA_N; i += 8) PL(PL_V + i) = PA_DC_OP[i];           // dc_operating_point == DC_OP (never re-set by the adapter)
    for (int i = role; i < PA_M; i += 8) { PL(PL_IP + i) = PA_DC_NL_I[i]; PL(PL_IPP + i) = PA_DC_NL_I[i]; }
    PA_SYNC();
}
struct PaSettledScal { double dcx, dcy, peak; unsigned long long clamp_cnt, nrmax_cnt, nan_cnt; };
__device__ __noinline__ PaSettledScal pa_state_from_settled(double* __restrict__ W, int role, const double* __restrict__ settled, int rate_is_codegen) {
    PA_SYNC();
    for (int i = role; i < PA_N; i += 8) PL(PL_V + i) = settled[PAS_V + i];
    for (int i = role; i < PA_M; i += 8) { PL(PL_IP + i) = settled[PAS_IP + i]; PL(PL_IPP + i) = settled[PAS_IPP + i]; }
    PaSettledScal r;
    r.dcx = settled[PAS_DCX]; r.dcy = settled[PAS_DCY]; r.peak = settled[PAS_PEAK];
    r.clamp_cnt = dbits(settled[PAS_CLAMP]); r.nrmax_cnt = dbits(settled[PAS_NRMAX]); r.nan_cnt = dbits(settled[PAS_NAN]);
    if (!rate_is_codegen) { r.dcx = 0.0; r.dcy = 0.0; }
    PA_SYNC();
    return r;
}

// gen_power_amp.rs:8838-12337.  off_p / off_n: the runtime rail offsets (v_rail_pos_offset / v_rail_neg_offset of the state).
// T: the LDS tables.  Rows of the matrix-vector products and of the state vectors are dealt to the lanes as i = role, role + 8, ...
// The sample in three pieces, so that a kernel may run the passes of the main sweep one at a time (k_post_mpa):
//   pa_sample_begin   :8838-8955  right-hand side, prediction, p, the extrapolated i_nl -- returns the sanitised input
//   pa_newton_pass    the main sweep's passes, until one converges or 70 are spent (nr = passes before the converged one, or 70)
//   pa_sample_end     :10681-12337 v_new, BE retry, NaN reset, state update, DC blocker, clamp
__device__ __forceinline__ double pa_sample_begin(double* __restrict__ W, const PaTab* T, int role, double input_in, double off_p, double off_n) {
    const double input = isfinite(input_in) ? clampd(input_in, -100.0, 100.0) : 0.0;
    for (int i = role; i < PA_N; i += 8) PL(PL_V + i) = PL(PL_V + i) + 1e-25 - 1e-25;
    for (int i = role; i < PA_M; i += 8) PL(PL_IP + i) = PL(PL_IP + i) + 1e-25 - 1e-25;
    PA_SYNC();
    for (int i = role; i < PA_N; i += 8) {                        // rhs = RHS_CONST + A_neg v_prev (sparse, the reference's term order) + sources
        double acc = i >= 18 ? 22.5 : 0.0;                           // RHS_CONST (gen_power_amp.rs): the two supply rows
        for (int q = T->row_ptr[i]; q < T->row_ptr[i + 1]; ++q) acc = acc + T->nz_coef[q] * PL(PL_V + T->nz_col[q]);
        if (i == 0) acc = acc + input * (1.0 / PA_INPUT_RESISTANCE);
        if (i == 18) acc = acc + off_p;
        if (i == 19) acc = acc + off_n;
        PL(PL_RHS + i) = acc;
    }
    PA_SYNC();
    for (int i = role; i < PA_N; i += 8) {                        // v_pred = S rhs
        double sum = 0.0;
        for (int j = 0; j < PA_N; ++j) sum += T->s[i * PA_N + j] * PL(PL_RHS + j);
        PL(PL_VPRED + i) = sum;
    }
    PA_SYNC();
    for (int i = role; i < PA_M; i += 8) {
        const int na = (int)PA_P_NODE_A[i], nb = (int)PA_P_NODE_B[i];
        PL(PL_P + i) = PA_N_V[i][na] * PL(PL_VPRED + na) + PA_N_V[i][nb] * PL(PL_VPRED + nb);
        PL(PL_INL + i) = 2.0 * PL(PL_IP + i) - PL(PL_IPP + i);
    }
    PA_SYNC();
    return input;
}
__device__ __forceinline__ double pa_sample_end(PaScal& sc, const OwPaConsts* __restrict__ C, double* __restrict__ W, const PaTab* T, int role, double input,
                                                uint32_t nr) {
    sc.last_nr = nr;
    for (int i = role; i < PA_N; i += 8) {
        double x = PL(PL_VPRED + i);
        for (int j = 0; j < PA_M; ++j) x += T->s_ni[i * PA_M + j] * PL(PL_INL + j);
        PL(PL_VNEW + i) = x;
    }
    PA_SYNC();
    if (__builtin_expect(!(sc.last_nr < 70u), 0)) {               // backward-Euler-matrix retry
        sc.nrmax_cnt += 1ull;
        sc.last_nr = pa_be_retry(C, W, role, input);
    }
    bool finite = true;
    for (int i = 0; i < PA_N; ++i) finite = finite && isfinite(PL(PL_VNEW + i));
    const double raw_out = PL(PL_VNEW + 8);
    PA_SYNC();
    if (__builtin_expect(!finite, 0)) {
        pa_state_to_dc(W, role);
        sc.dcx = 0.0; sc.dcy = 0.0;
        sc.nan_cnt += 1ull;
        return PA_DC_BLOCK_X0;
    }
    for (int i = role; i < PA_N; i += 8) PL(PL_V + i) = PL(PL_VNEW + i);
    for (int i = role; i < PA_M; i += 8) { PL(PL_IPP + i) = PL(PL_IP + i); PL(PL_IP + i) = PL(PL_INL + i); }
    PA_SYNC();
    const double dc_blocked = raw_out - sc.dcx + C->dc_block_r * sc.dcy;
    sc.dcx = raw_out;
    sc.dcy = dc_blocked;
    const double scaled = dc_blocked * 1.0;
    const double abs_out = fabs(scaled);
    if (abs_out > sc.peak) sc.peak = abs_out;
    if (abs_out > 3e1) sc.clamp_cnt += 1ull;
    return clampd(scaled, -3e1, 3e1);
}
// the sample in lock step (settle and debug kernels)
__device__ __forceinline__ double pa_process_sample(PaScal& sc, const OwPaConsts* __restrict__ C, double* __restrict__ W, const PaTab* T, int role,
                                                    double input_in, double off_p, double off_n) {
    const double input = pa_sample_begin(W, T, role, input_in, off_p, off_n);
    const uint32_t nr = pa_newton_body<false>(W, T->k, &T->dev[role], role);
    return pa_sample_end(sc, C, W, T, role, input, nr);
}

// state <- settled blob (+ the per-state part of set_sample_rate when the chain does not run at the codegen rate): init_state,
// power_amp.rs:294-302
__device__ __forceinline__ void pa_init_state(PaScal& sc, double* __restrict__ W, int role, const double* __restrict__ settled, const OwPaConsts* __restrict__ C) {
    const PaSettledScal r = pa_state_from_settled(W, role, settled, C->rate_is_codegen);
    sc.dcx = r.dcx; sc.dcy = r.dcy; sc.peak = r.peak;
    sc.clamp_cnt = r.clamp_cnt; sc.nrmax_cnt = r.nrmax_cnt; sc.nan_cnt = r.nan_cnt;
    sc.last_nr = 0u;
}
__device__ __forceinline__ void pa_rails_reset(PaScal& sc) { sc.rail_p = 22.5; sc.rail_n = 22.5; sc.iavg_p = 0.0; sc.iavg_n = 0.0; }

// melange_adapter::PowerAmp::process, power_amp.rs:373-431
// (pa_process_end: everything after process_sample returns `raw`)
__device__ __forceinline__ double pa_process_end(PaScal& sc, const OwPaConsts* __restrict__ C, double* __restrict__ W, int role,
                                                 const double* __restrict__ settled, double raw, bool rail_sag) {
    const double result = OW_DIV_C(raw, 22.0);
    const bool nr_failed = sc.last_nr >= 69u;
    bool insane = false;
    for (int i = 0; i < PA_N; ++i) { const double v = PL(PL_V + i); insane = insane || !isfinite(v) || fabs(v) > 100.0; }
    if (__builtin_expect(!isfinite(result) || nr_failed || insane, 0)) {
        pa_init_state(sc, W, role, settled, C);
        pa_rails_reset(sc);
        sc.guard_cnt += 1ull;
        return sc.last_good;
    }
    const double clamped = clampd(result, -1.0, 1.0);
    sc.last_good = clamped;
    if (rail_sag) {        // RailDynamics::step(raw), power_amp.rs:131-156
        const double i_pos = fmax(OW_DIV_C(raw, 8.0), 0.0);
        const double i_neg = fmax(OW_DIV_C(-raw, 8.0), 0.0);
        sc.iavg_p += C->alpha_i_avg * (i_pos - sc.iavg_p);
        sc.iavg_n += C->alpha_i_avg * (i_neg - sc.iavg_n);
        const double target_pos = 24.5 - sc.iavg_p * 3.5;
        const double target_neg = 24.5 - sc.iavg_n * 3.5;
        const double alpha_p = target_pos < sc.rail_p ? C->alpha_attack : C->alpha_release;
        const double alpha_n = target_neg < sc.rail_n ? C->alpha_attack : C->alpha_release;
        sc.rail_p += alpha_p * (target_pos - sc.rail_p);
        sc.rail_n += alpha_n * (target_neg - sc.rail_n);
    }
    return clamped;
}
__device__ __forceinline__ double pa_process(PaScal& sc, const OwPaConsts* __restrict__ C, double* __restrict__ W, const PaTab* T, int role,
                                             const double* __restrict__ settled, double input, bool rail_sag) {
    const double off_p = rail_sag ? sc.rail_p - 22.5 : 0.0, off_n = rail_sag ? sc.rail_n - 22.5 : 0.0;
    const double raw = pa_process_sample(sc, C, W, T, role, input, off_p, off_n);
    return pa_process_end(sc, C, W, role, settled, raw, rail_sag);
}

// engine e's state rows <-> LDS (rows dealt to the eight lanes; the scalars are read by all, stored by role 0)
__device__ __forceinline__ void pa_load(PaScal& sc, double* __restrict__ W, int role, const double* __restrict__ pa, int I, int e) {
    for (int i = role; i < PA_N; i += 8) PL(PL_V + i) = pa[(size_t)(PAS_V + i) * I + e];
    for (int i = role; i < PA_M; i += 8) { PL(PL_IP + i) = pa[(size_t)(PAS_IP + i) * I + e]; PL(PL_IPP + i) = pa[(size_t)(PAS_IPP + i) * I + e]; }
    sc.dcx = pa[(size_t)PAS_DCX * I + e]; sc.dcy = pa[(size_t)PAS_DCY * I + e]; sc.peak = pa[(size_t)PAS_PEAK * I + e];
    sc.clamp_cnt = dbits(pa[(size_t)PAS_CLAMP * I + e]); sc.nrmax_cnt = dbits(pa[(size_t)PAS_NRMAX * I + e]); sc.nan_cnt = dbits(pa[(size_t)PAS_NAN * I + e]);
    sc.last_good = pa[(size_t)PAS_LASTGOOD * I + e];
    sc.rail_p = pa[(size_t)PAS_RAILP * I + e]; sc.rail_n = pa[(size_t)PAS_RAILN * I + e];
    sc.iavg_p = pa[(size_t)PAS_IAVGP * I + e]; sc.iavg_n = pa[(size_t)PAS_IAVGN * I + e];
    sc.guard_cnt = dbits(pa[(size_t)PAS_GUARD * I + e]);
    sc.last_nr = 0u;
    PA_SYNC();
}
__device__ __forceinline__ void pa_store(const PaScal& sc, const double* __restrict__ W, int role, double* __restrict__ pa, int I, int e) {
    PA_SYNC();
    for (int i = role; i < PA_N; i += 8) pa[(size_t)(PAS_V + i) * I + e] = PL(PL_V + i);
    for (int i = role; i < PA_M; i += 8) { pa[(size_t)(PAS_IP + i) * I + e] = PL(PL_IP + i); pa[(size_t)(PAS_IPP + i) * I + e] = PL(PL_IPP + i); }
    if (role != 0) return;
    pa[(size_t)PAS_DCX * I + e] = sc.dcx; pa[(size_t)PAS_DCY * I + e] = sc.dcy; pa[(size_t)PAS_PEAK * I + e] = sc.peak;
    pa[(size_t)PAS_CLAMP * I + e] = bitsd(sc.clamp_cnt); pa[(size_t)PAS_NRMAX * I + e] = bitsd(sc.nrmax_cnt); pa[(size_t)PAS_NAN * I + e] = bitsd(sc.nan_cnt);
    pa[(size_t)PAS_LASTGOOD * I + e] = sc.last_good;
    pa[(size_t)PAS_RAILP * I + e] = sc.rail_p; pa[(size_t)PAS_RAILN * I + e] = sc.rail_n;
    pa[(size_t)PAS_IAVGP * I + e] = sc.iavg_p; pa[(size_t)PAS_IAVGN * I + e] = sc.iavg_n;
    pa[(size_t)PAS_GUARD * I + e] = bitsd(sc.guard_cnt);
}
// Settled state of the amp (compute_settled_state, power_amp.rs:290-296): CircuitState::default() (DC_OP + 50 warm-up samples) and
// 44 100 silent samples, all with the codegen-rate matrices.  One wavefront (its eight engine slots run the same silent settle);
// cached per device by the host like the reference's OnceLock.
__global__ __launch_bounds__(64) void k_mpa_settle(const OwPaConsts* __restrict__ C88, double* __restrict__ settled) {
    __shared__ double WS[PL_ROWS * PA_LS];
    __shared__ PaTab TS;
    pa_stage_tables(&TS, C88);
    const PaTab* T = &TS;
    const int role = threadIdx.x >> 3;
    double* W = WS + (threadIdx.x & 7);
    PaScal sc;
    for (int i = role; i < PA_N; i += 8) PL(PL_V + i) = PA_DC_OP[i];
    for (int i = role; i < PA_M; i += 8) { PL(PL_IP + i) = PA_DC_NL_I[i]; PL(PL_IPP + i) = PA_DC_NL_I[i]; }
    sc.dcx = PA_DC_BLOCK_X0; sc.dcy = 0.0; sc.peak = 0.0; sc.clamp_cnt = sc.nrmax_cnt = sc.nan_cnt = sc.guard_cnt = 0ull;
    sc.last_good = 0.0; sc.last_nr = 0u;
    pa_rails_reset(sc);
    PA_SYNC();
    for (int n = 0; n < 50 + 44100; ++n) pa_process_sample(sc, C88, W, T, role, 0.0, 0.0, 0.0);
    PA_SYNC();
    if (threadIdx.x != 0) return;
    for (int i = 0; i < PA_N; ++i) settled[PAS_V + i] = PL(PL_V + i);
    for (int i = 0; i < PA_M; ++i) { settled[PAS_IP + i] = PL(PL_IP + i); settled[PAS_IPP + i] = PL(PL_IPP + i); }
    settled[PAS_DCX] = sc.dcx; settled[PAS_DCY] = sc.dcy; settled[PAS_PEAK] = sc.peak;
    settled[PAS_CLAMP] = bitsd(sc.clamp_cnt); settled[PAS_NRMAX] = bitsd(sc.nrmax_cnt); settled[PAS_NAN] = bitsd(sc.nan_cnt);
}

// The amp alone on given input (debug hook ow_debug_power_amp): engine slot s of block b processes row b * 32 + s of `in`.  taps
// [row][n][3]: outer Newton iterations of the sample (70 = exhausted), guard resets so far, positive rail after the sample.
__global__ __launch_bounds__(PA_WPB * 64, PA_MIN_BLOCKS) void k_mpa_debug(const OwPaConsts* __restrict__ C, const double* __restrict__ settled, const double* __restrict__ in,
                                                           double* __restrict__ out, double* __restrict__ taps, long long n, int n_rows, int rail_sag,
                                                           const long long* __restrict__ poke_at, const int* __restrict__ poke_node, const double* __restrict__ poke_val,
                                                           long long ld = 0) {
    if (ld == 0) ld = n;                     // row stride of in / out (the batch path hands over rows of a wider buffer)
    __shared__ double WS[PA_LDS_DOUBLES];
    __shared__ PaTab TS;
    pa_stage_tables(&TS, C);
    const PaTab* T = &TS;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int role = lane >> 3, slot = wv * PA_EPW + (lane & 7);
    double* W = WS + wv * (PL_ROWS * PA_LS) + (lane & 7);
    const int row_raw = blockIdx.x * PA_EPB + slot;
    const bool valid = row_raw < n_rows;
    const size_t row = valid ? row_raw : n_rows - 1;
    PaScal sc;
    pa_init_state(sc, W, role, settled, C);
    pa_rails_reset(sc);
    sc.last_good = 0.0; sc.guard_cnt = 0ull;
    // one Newton pass per trip for every row that still has samples, each row on its own sample counter (see k_post_mpa)
    long long i = 0;
    uint32_t iter = 0u;
    bool begun = false;
    double input = 0.0;
    while (__builtin_amdgcn_ballot_w64(i < n) != 0ull) {
        if (i < n) {
            if (!begun) {
                if (poke_at && poke_at[row] == i) { PA_SYNC(); PL(PL_V + poke_node[row]) = poke_val[row]; }
                PA_SYNC();
                const double off_p = rail_sag ? sc.rail_p - 22.5 : 0.0, off_n = rail_sag ? sc.rail_n - 22.5 : 0.0;
                input = pa_sample_begin(W, T, role, in[row * ld + i], off_p, off_n);
                begun = true;
                iter = 0u;
            }
            const bool conv = pa_newton_pass<false>(W, T->k, &T->dev[role], role);
            if (conv || iter == 69u) {
                const double raw = pa_sample_end(sc, C, W, T, role, input, conv ? iter : 70u);
                const double y = pa_process_end(sc, C, W, role, settled, raw, rail_sag != 0);
                if (valid && role == 0) {
                    out[row * ld + i] = y;
                    if (taps) {
                        double* t = taps + (row * n + i) * 3;
                        t[0] = (double)sc.last_nr; t[1] = (double)sc.guard_cnt; t[2] = sc.rail_p;
                    }
                }
                ++i;
                begun = false;
            } else {
                ++iter;
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

// Engines [e0, e0 + ne) by falling demand (Newton passes of their last block, `demand`, in units of 1/16 pass per chain-rate sample over
// the minimum of one; 256 classes): counting sort in three launches.  The order INSIDE a class is whatever the atomics make it -- it
// decides which engines share a wavefront, not what any engine computes.
#define PA_ORDER_CLASSES 256
__device__ __forceinline__ int pa_demand_class(uint32_t d, uint32_t total) {
    if (d <= total) return 0;
    const unsigned long long q = (unsigned long long)(d - total) * 16ull / (total ? total : 1u);
    return q > (unsigned long long)(PA_ORDER_CLASSES - 1) ? PA_ORDER_CLASSES - 1 : (int)q;
}
__global__ void k_pa_order_hist(const uint32_t* __restrict__ demand, int e0, int ne, uint32_t total, uint32_t* __restrict__ hist) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < ne) atomicAdd(&hist[pa_demand_class(demand[e0 + t], total)], 1u);
}
__global__ __launch_bounds__(PA_ORDER_CLASSES) void k_pa_order_scan(uint32_t* __restrict__ hist) {   // hist -> first position of the class (heaviest class first)
    __shared__ uint32_t h[PA_ORDER_CLASSES];
    h[threadIdx.x] = hist[threadIdx.x];
    __syncthreads();
    uint32_t base = 0;
    for (int b = PA_ORDER_CLASSES - 1; b > (int)threadIdx.x; --b) base += h[b];
    hist[threadIdx.x] = base;
}
__global__ void k_pa_order_scatter(const uint32_t* __restrict__ demand, int e0, int ne, uint32_t total, uint32_t* __restrict__ cursor, uint32_t* __restrict__ order) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ne) return;
    const uint32_t pos = atomicAdd(&cursor[pa_demand_class(demand[e0 + t], total)], 1u);
    order[e0 + pos] = (uint32_t)(e0 + t);
}

// Output stage with the melange power amp: eight lanes per engine, 32 engines per workgroup (the amp is a stateful recurrence at the
// chain rate, so the two chain-rate samples of an output sample are solved one after the other), then half-band down, speaker, gain,
// f32 as k_post.  That cheap tail runs on the role-0 lane of the engine with its state (filters, smoothers) in LDS between samples:
// in registers it would sit on top of the solver's ~430 and spill to scratch.
struct PaTail { SpeakerSt sp; Smoother ss, sv; double da[3], db[3], dd; };
__global__ __launch_bounds__(PA_WPB * 64, PA_MIN_BLOCKS) void k_post_mpa(const OwConsts* __restrict__ K, const OwPaConsts* __restrict__ C, const double* __restrict__ settled,
                                                          double* __restrict__ cs, double* __restrict__ pa, const OwEngineArgs* __restrict__ args,
                                                          OwEngineOut* __restrict__ eout, const double* __restrict__ pre, float* __restrict__ out,
                                                          double* __restrict__ pa_tap, int I, int L, int Lout, int e0, int ne,
                                                          const uint32_t* __restrict__ order, uint32_t* __restrict__ demand) {
    __shared__ double WS[PA_LDS_DOUBLES];                        // 61 KB
    __shared__ PaTab TS;
    __shared__ PaTail TL[PA_EPB];
    pa_stage_tables(&TS, C);
    const PaTab* T = &TS;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int role = lane >> 3, slot = wv * PA_EPW + (lane & 7);
    double* W = WS + wv * (PL_ROWS * PA_LS) + (lane & 7);
    // position g of the range -> engine: in index order, or through `order` (the range's engines sorted by the Newton passes their
    // last block took, heaviest first -- k_pa_order_*): a wavefront lasts as long as its slowest engine, a workgroup as its slowest
    // wavefront, so engines of like demand share them and the long ones are dispatched first
    const int g_raw = blockIdx.x * PA_EPB + slot;
    const bool valid = g_raw < ne;
    const int g = valid ? g_raw : ne - 1;
    const int e = order ? (int)order[e0 + g] : e0 + g;
    const int osr = K->oversample ? 2 : 1;
    const bool rail_sag = (args[e].pa_flags & 1u) != 0u;
    PaTail& t = TL[slot];
    if (role == 0) {
        for (int i = 0; i < 3; ++i) { t.da[i] = CSF(CS_OS_DA + i); t.db[i] = CSF(CS_OS_DB + i); }
        t.dd = CSF(CS_OS_DD);
        double* hp = &t.sp.hpf.b0; double* lp = &t.sp.lpf.b0;
        for (int i = 0; i < 7; ++i) { hp[i] = CSF(CS_SPK_HPF + i); lp[i] = CSF(CS_SPK_LPF + i); }
        t.sp.character = CSF(CS_SPK_CHAR); t.sp.a2 = CSF(CS_SPK_A2); t.sp.a3 = CSF(CS_SPK_A3); t.sp.tc = CSF(CS_SPK_TC); t.sp.ts = CSF(CS_SPK_TS);
        smoother_load(t.ss, cs, I, e, CS_SM_SPK);
        smoother_load(t.sv, cs, I, e, CS_SM_VOL);
        if (args[e].set_flags & 2u) t.ss.retarget(args[e].spk_target, K->ramp_samples);
        if (args[e].set_flags & 4u) t.sv.retarget(args[e].vol_target, K->ramp_samples);
    }
    PaScal sc;
    pa_load(sc, W, role, pa, I, e);
    bool bad_any = false;                                                 // output NaN guard fired in this block (role-0 lane)
    // The engines of the wavefront are NOT kept on the same sample.  Newton pass counts are spiky (a sample in fifty takes 40-70 passes
    // where its neighbours take two or three; eight engines with different material: mean 2.8 passes per sample and engine, mean of
    // the slowest of eight 7), and a wavefront that solved sample n for all its engines before going to n + 1 would pay the slowest one
    // every time.  Instead every trip of the loop below is ONE pass for every engine that still has samples, preceded by the sample's
    // set-up for the engines that begin one and followed by its completion for those whose pass converged: each engine (its eight
    // lanes: they share c / iter / begun) walks its own sample counter, and the wavefront takes max over engines of the SUM of
    // passes, not the sum of the maxima.  Nothing but the schedule changes: an engine's operations and their order are those of
    // pa_process in a loop over its samples.
    const int total = L * osr;
    int c = 0;                                                            // chain-rate sample this engine is in
    uint32_t iter = 0u;                                                   // passes of the main sweep already spent on it
    uint32_t trips = 0u;                                                  // passes of this block (the engine's demand figure)
    bool begun = false;
    double input = 0.0, y0 = 0.0;
    while (__builtin_amdgcn_ballot_w64(c < total) != 0ull) {
        OW_DBG_WAVE(2);
        if (c < total) {
            ++trips;
            if (!begun) {
                const double off_p = rail_sag ? sc.rail_p - 22.5 : 0.0, off_n = rail_sag ? sc.rail_n - 22.5 : 0.0;
                input = pa_sample_begin(W, T, role, pre[(size_t)c * I + e] * 0.25, off_p, off_n);   // x FIXED_CIRCUIT_DRIVE, engine.rs:544-546
                begun = true;
                iter = 0u;
            }
            const bool conv = pa_newton_pass<false>(W, T->k, &T->dev[role], role);
            if (conv || iter == 69u) {
                OW_DBG_WAVE(6);
                const double raw = pa_sample_end(sc, C, W, T, role, input, conv ? iter : 70u);
                const double y = pa_process_end(sc, C, W, role, settled, raw, rail_sag);
                if (pa_tap && valid && role == 0) pa_tap[(size_t)c * I + e] = y;
                const bool second = osr == 2 && (c & 1);
                if (role == 0 && (osr == 1 || second)) {
                    const int n = osr == 2 ? (c >> 1) : c;
                    double o;
                    if (osr == 2) {
                        const double a = allpass3(OW_OS_A0, OW_OS_A1, OW_OS_A2, t.da, y0);
                        const double b = allpass3(OW_OS_B0, OW_OS_B1, OW_OS_B2, t.db, y);
                        o = (a + t.dd) * 0.5;
                        t.dd = b;
                    } else {
                        o = y;
                    }
                    speaker_set_character(t.sp, t.ss.next(), K->sr);
                    const double shaped = speaker_process(t.sp, o, K->spk_thermal_alpha);
                    const double post = shaped * 7.498942093324558 * t.sv.next();
                    float f = (float)post;
                    const bool bad = !isfinite(f);
                    if (bad) {                                                          // engine.rs:450-458
                        f = 0.0f;
                        t.sp.hpf.s1 = t.sp.hpf.s2 = t.sp.lpf.s1 = t.sp.lpf.s2 = 0.0;
                        t.sp.ts = 0.0;
                    }
                    // 4 bytes per engine per output sample, straight to its row: 33 MB per block at 16 384 engines against >= 100 ms of
                    // solver time -- not worth 8 KB of LDS for a transposing tile
                    if (valid) out[(size_t)e * Lout + n] = f;
                    bad_any = bad_any || bad;
                }
                y0 = y;
                ++c;
                begun = false;
            } else {
                ++iter;
            }
        }
    }
    // engine.rs:450-458 resets preamp, oversampler and POWER AMP at the faulty output sample -- but render_voices_to_preamp_out has
    // run all three over the whole block before the speaker loop starts (engine.rs:432-434), so those resets act on the post-block
    // state: here, at the block's end.  Only the speaker is reset in-sample (above).
    const bool nan_fired = __shfl((int)bad_any, lane & 7) != 0;          // from the engine's role-0 lane (same wavefront)
    if (__builtin_expect(nan_fired, 0)) {
        pa_init_state(sc, W, role, settled, C);
        pa_rails_reset(sc);
    }
    if (valid) pa_store(sc, W, role, pa, I, e);
    if (!valid || role != 0) return;
    if (demand) demand[e] = trips;
    if (nan_fired) {
        for (int i = 0; i < 3; ++i) { t.da[i] = 0.0; t.db[i] = 0.0; }
        t.dd = 0.0;
        CSF(CS_FLAGS) = bitsd(dbits(CSF(CS_FLAGS)) | 1ull);
        eout[e].out_nonfinite = 1u;
    }
    for (int i = 0; i < 3; ++i) { CSF(CS_OS_DA + i) = t.da[i]; CSF(CS_OS_DB + i) = t.db[i]; }
    CSF(CS_OS_DD) = t.dd;
    {
        const double* hp = &t.sp.hpf.b0; const double* lp = &t.sp.lpf.b0;
        for (int i = 0; i < 7; ++i) { CSF(CS_SPK_HPF + i) = hp[i]; CSF(CS_SPK_LPF + i) = lp[i]; }
        CSF(CS_SPK_CHAR) = t.sp.character; CSF(CS_SPK_A2) = t.sp.a2; CSF(CS_SPK_A3) = t.sp.a3; CSF(CS_SPK_TC) = t.sp.tc; CSF(CS_SPK_TS) = t.sp.ts;
    }
    smoother_store(t.ss, cs, I, e, CS_SM_SPK);
    smoother_store(t.sv, cs, I, e, CS_SM_VOL);
}

}  // namespace owdev
