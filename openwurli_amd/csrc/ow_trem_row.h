// openwurli-hip: the Twin-T oscillator as ONE system per wavefront, written for the instruction count of its serial step.
//
// The shared trajectory (k_trem_traj_extend) and the settle of Tremolo::new are a single recurrence that nothing can run beside: what
// every small pool waits for is (instructions per step) x (issue interval of a lone wavefront: 5.8 cycles per independent f64
// instruction, 8.8 per dependent one -- profiles/r04_issue_cost.txt).  The quad-lane step (ow_trem_wide.h) spends 1 385 vector + 470
// scalar instructions per step, half of them moves, selects and branch bookkeeping.  This step keeps the same arithmetic -- every
// number is produced by the same operations on the same operands as in trem_osc_step / trem_nr<false> / solve4 (ow_chain_dev.h), so
// the R stream is bit-identical (tests/test_gpu_trajectory.py, test_gpu_parity.py::test_tremolo_wide_is_bit_identical) -- and
// removes the rest:
//   * lanes of a ROW of sixteen: lane r = matrix row r of the three matrix-vector products (A_neg v + N_i i, S rhs, S_NI i_nl) with the
//     operand vector wave-uniform and the emitted sparsity as zero coefficients (x + 0*y is x: partial sums here are never -0), lane
//     q = r & 3 = port q of the Newton sweep (one junction exponential, one Jacobian row, one row of the pivoted elimination); the four
//     rows of the wavefront and the lanes above the matrix sizes repeat the same work.  Values cross lanes with ONE v_mov_b64_dpp
//     row_newbcast per double (two 32-bit quad_perm moves before);
//   * the elimination never moves a row: the usual pivot order (row 2 for column 0, the diagonal otherwise: solve4) is a fixed
//     assignment lane -> logical row (2, 1, 0, 3); every pivot test, the singularity tests, the junction limiter's and the 3.5 V cap's
//     triggers are collected in ONE mask per sweep, and a sweep in which any of them fires is redone from its unchanged input by the
//     generic sweep (trem_nr_sweep<false>: the statement-for-statement one) -- one branch per sweep instead of nine;
//   * the pivots are wave-uniform after their broadcast, so their refined reciprocals are shared by the elimination factors and the back
//     substitution (ow_div_y: ow_div instruction for instruction); an unlimited sweep has ga == 1.0, i.e. i_nl - ga*b IS i_trial;
//   * the per-port device laws are written once with per-lane coefficients (collector rows / base rows) instead of both + selects:
//     a - c*u == a + (-c)*u and x - 0*y == x exactly.
// gen_tremolo.rs:2353-3116 (input == 0), tremolo.rs:121-146.
#pragma once
#include "ow_trem_wide.h"

namespace owdev {

template <int L> OW_DEV double rowb(double x) {     // lane L of this row of sixteen lanes, to all sixteen
    long long v = __double_as_longlong(x);
    v = __builtin_amdgcn_update_dpp(__builtin_nondeterministic_value(v), v, 0x150 + L, 0xF, 0xF, true);      // every lane is written: no `old` to set up
    return __longlong_as_double(v);
}
OW_DEV double qswap02(double x) { return qperm<0xC6>(x); }      // lanes 0 <-> 2 of every quad

struct TremRowK {      // per-lane coefficients (registers), r = lane & 15, q = lane & 3
    double an[7], ni[4], rhs0;      // row r of A_neg (emitted entries only), N_i, RHS_CONST; zero rows for r >= 7
    double s[7], sni[4];            // row r of S, S_NI
    double kvd[4], kq[4];           // row q of K: as v_d reads it (no [1][3], no [2][2]: gen_tremolo.rs:2423-2438) / in full
    double k0[4], k1[4];            // rows (q & 2), (q & 2) + 1 of K: the two junctions of port q's transistor
    double dq[4];                   // row q of the identity
    double cA, cB1, cB2, g1, g2;    // device-law coefficients of a collector row (even q) / a base row (odd q)
};
OW_DEV void trem_row_consts(TremRowK& c, const OwConsts* __restrict__ K, int lane) {
    const int r = lane & 15, q = lane & 3;
    const int rr = r < 7 ? r : 0;
    const unsigned emitted[7] = {0x2Bu, 0x07u, 0x0Eu, 0x0Du, 0x10u, 0x61u, 0x00u};      // gen_tremolo.rs:2360-2400: the 17 A_neg terms
    const double nimat[7][4] = {{-1, 0, -1, 0}, {0, 0, 0, 0}, {0, -1, 0, 0}, {0, 0, 0, 0}, {1, 1, 0, -1}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        c.an[j] = (r < 7 && ((emitted[rr] >> j) & 1u)) ? K->t_a_neg[rr][j] : 0.0;
        c.s[j] = r < 7 ? K->t_s[rr][j] : 0.0;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        c.ni[j] = r < 7 ? nimat[rr][j] : 0.0;
        c.sni[j] = r < 7 ? K->t_s_ni[rr][j] : 0.0;
        c.kq[j] = K->t_k[q][j];
        c.kvd[j] = ((q == 1 && j == 3) || (q == 2 && j == 2)) ? 0.0 : K->t_k[q][j];
        c.k0[j] = K->t_k[q & 2][j];
        c.k1[j] = K->t_k[(q & 2) + 1][j];
        c.dq[j] = j == q ? 1.0 : 0.0;
    }
    c.rhs0 = r == 6 ? 15.0 : 0.0;
    const double is = OW_T_IS, vt = OW_T_VT, beta_f = OW_T_BF, beta_r = OW_T_BR;
    const bool base = (q & 1) != 0;
    c.cA = base ? (is / (beta_f * (1.0 * vt))) : (is / (1.0 * vt));                      // j2 | j0
    c.cB1 = base ? (is / (beta_r * (1.0 * vt))) : -(is / (1.0 * vt));                   // j3 | j1's first product
    c.cB2 = base ? 0.0 : (is / (beta_r * (1.0 * vt)));                                   //    | j1's second product
    c.g1 = base ? (is / beta_f) : is;                                                    // ib_fwd | i_cc
    c.g2 = base ? (is / beta_r) : -(is / beta_r);                                        // + ib_rev | - ib_rev
}

// sweeps the row step handed to the generic sweep / steps that took the backward-Euler retry, since the library was loaded (one atomic in
// a cold block; read by ow_debug_trem_trajectory: the bit-identity test wants to know that its scenario exercises those paths)
__device__ unsigned long long g_trem_row_cold[2] = {0ull, 0ull};
struct TremRow {
    double vf[7], ipf[4], ipp[4];   // wave-uniform: v and i_prev after the denormal flush (what the step reads), i_pp
    double ip[4];                   // wave-uniform: i_prev as the state holds it (before the flush)
    double v_me;                    // lane r: v[r] as the state holds it (0 for r >= 7)
    double env;
    uint32_t be_fallbacks;
};
OW_DEV double trem_flush(double x) { return x + 1e-25 - 1e-25; }

// The generic sweep and the cold tails, out of line: the hot loop stays small and the cold code's registers are its own.
__device__ inline bool trem_row_sweep_generic(const double* __restrict__ vp, const OwConsts* __restrict__ K, double* __restrict__ i_nl) {
    double p[4], x[4];
    p[0] = 1.0 * vp[2] + -1.0 * vp[4];
    p[1] = -1.0 * vp[0] + 1.0 * vp[2];
    p[2] = 1.0 * vp[4];
    p[3] = -1.0 * vp[0] + 1.0 * vp[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = i_nl[j];
    const bool conv = trem_nr_sweep<false, true>(p, K->t_k, x);
#pragma unroll
    for (int j = 0; j < 4; ++j) i_nl[j] = x[j];
    return conv;
}
__device__ inline void trem_row_be_fallback(const OwConsts* __restrict__ K, const double* __restrict__ vf, const double* __restrict__ ipf,
                                                  const double* __restrict__ ipp, double* __restrict__ v, double* __restrict__ i_nl) {
    double a[7], b[4], c[4], vo[7], io[4];
#pragma unroll
    for (int i = 0; i < 7; ++i) a[i] = vf[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) { b[i] = ipf[i]; c[i] = ipp[i]; io[i] = i_nl[i]; }
    trem_be_fallback(K, a, b, c, vo, io);
#pragma unroll
    for (int i = 0; i < 7; ++i) v[i] = vo[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) i_nl[i] = io[i];
}
OW_DEV double pick4(const double x[4], int q) {      // x[q], as three selects
    const bool lo = (q & 1) != 0, hi = (q & 2) != 0;
    const double a = lo ? x[1] : x[0], b = lo ? x[3] : x[2];
    return hi ? b : a;
}
OW_DEV double pick7(const double x[7], int r) {
    return r == 0 ? x[0] : (r == 1 ? x[1] : (r == 2 ? x[2] : (r == 3 ? x[3] : (r == 4 ? x[4] : (r == 5 ? x[5] : (r == 6 ? x[6] : 0.0))))));
}

// One oscillator step.  Returns v[OUT] in lane 0 of every row (lane r holds v[r]).
__device__ __forceinline__ double trem_osc_step_row(TremRow& st, const TremRowK& c, const OwConsts* __restrict__ K, int lane) {
    const int q = lane & 3;
    // rhs = RHS_CONST + A_neg v + N_i i_prev (+ the input source: + 0.0), row r
    double acc = c.rhs0;
#pragma unroll
    for (int j = 0; j < 7; ++j) acc += c.an[j] * st.vf[j];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc += c.ni[j] * st.ipf[j];
    double rhs[7];
    static_for<0, 7>([&](auto I) { rhs[I] = rowb<I>(acc); });
    // v_pred = S rhs, row r
    double va = 0.0;
#pragma unroll
    for (int j = 0; j < 7; ++j) va += c.s[j] * rhs[j];
    double vp[7];
    static_for<0, 7>([&](auto I) { vp[I] = rowb<I>(va); });
    // p[q] = N_v v_pred: (vp2 - vp4, vp2 - vp0, vp4, vp4 - vp0): -1.0 * x + 1.0 * y is y - x, 1.0 * x is x - 0
    const double pa = (q & 2) ? vp[4] : vp[2];
    const double pb1 = (q & 1) ? vp[0] : vp[4];
    const double pb0 = q == 2 ? 0.0 : pb1;
    const double pq = pa - pb0;
    // i_nl = 2 i_prev - i_pp (2x is exact, so the fused form rounds once like the product-then-difference)
    double i_nl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) i_nl[j] = __builtin_fma(2.0, st.ipf[j], -st.ipp[j]);
    double i_me = pick4(i_nl, q);

    // ---- Newton-Raphson, trapezoidal (trem_nr<false>)
    bool converged = false;
    for (int iter = 0; iter < 50; ++iter) {
        const double vd = pq + c.kvd[0] * i_nl[0] + c.kvd[1] * i_nl[1] + c.kvd[2] * i_nl[2] + c.kvd[3] * i_nl[3];
        const double e_me = fast_exp(OW_DIV_C(1.0 * vd, 1.0 * OW_T_VT));
        const double e_ot = qswap1(e_me);
        const double exp_be = (q & 1) ? e_ot : e_me, exp_bc = (q & 1) ? e_me : e_ot;
        const double u2 = exp_bc - 1.0;
        const double sel = exp_be - ((q & 1) ? 1.0 : exp_bc);
        const double cur = c.g1 * sel + c.g2 * u2;                 // ic (even q) | ib (odd q)
        const double jA = c.cA * exp_be;
        const double jB = c.cB1 * exp_bc - c.cB2 * exp_bc;
        double ar[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) ar[j] = c.dq[j] - jA * c.k0[j] - jB * c.k1[j];
        double br = i_me - cur;
        // ---- solve4 with the usual pivot order: logical rows (0, 1, 2, 3) live in lanes (2, 1, 0, 3)
        uint64_t bad = 0ull;
        // column 0: pivot row in lane 2, rows below it in lanes 1, 0, 3
        const double pe0 = rowb<2>(ar[0]);
        bad |= __builtin_amdgcn_ballot_w64(!(fabs(pe0) > fabs(ar[0]))) & 0xBBBBBBBBBBBBBBBBull;
        bad |= __builtin_amdgcn_ballot_w64(fabs(pe0) < 1e-15);
        const double y0 = ow_rcp_refined(pe0);
        {
            const double p1 = rowb<2>(ar[1]), p2 = rowb<2>(ar[2]), p3 = rowb<2>(ar[3]), pb = rowb<2>(br);
            if (q != 2) {
                const double f = ow_div_y(ar[0], pe0, y0);
                ar[1] -= f * p1; ar[2] -= f * p2; ar[3] -= f * p3; br -= f * pb;
            }
        }
        // column 1: pivot row in lane 1, rows below it in lanes 0, 3
        const double pe1 = rowb<1>(ar[1]);
        bad |= __builtin_amdgcn_ballot_w64(!(fabs(pe1) > fabs(ar[1]))) & 0x9999999999999999ull;
        bad |= __builtin_amdgcn_ballot_w64(fabs(pe1) < 1e-15);
        const double y1 = ow_rcp_refined(pe1);
        {
            const double p2 = rowb<1>(ar[2]), p3 = rowb<1>(ar[3]), pb = rowb<1>(br);
            if (q == 0 || q == 3) {
                const double f = ow_div_y(ar[1], pe1, y1);
                ar[2] -= f * p2; ar[3] -= f * p3; br -= f * pb;
            }
        }
        // column 2: pivot row in lane 0, the row below it in lane 3
        const double pe2 = rowb<0>(ar[2]);
        bad |= __builtin_amdgcn_ballot_w64(!(fabs(pe2) > fabs(ar[2]))) & 0x8888888888888888ull;
        bad |= __builtin_amdgcn_ballot_w64(fabs(pe2) < 1e-15);
        const double y2 = ow_rcp_refined(pe2);
        {
            const double p3 = rowb<0>(ar[3]), pb = rowb<0>(br);
            if (q == 3) {
                const double f = ow_div_y(ar[2], pe2, y2);
                ar[3] -= f * p3; br -= f * pb;
            }
        }
        const double pe3 = rowb<3>(ar[3]);
        bad |= __builtin_amdgcn_ballot_w64(fabs(pe3) < 1e-15);
        const double y3 = ow_rcp_refined(pe3);
        // back substitution: unknown i is solved in the lane of logical row i
        double b[4];
        b[3] = rowb<3>(ow_div_y(br, pe3, y3));
        b[2] = rowb<0>(ow_div_y(br - ar[3] * b[3], pe2, y2));
        b[1] = rowb<1>(ow_div_y(br - ar[2] * b[2] - ar[3] * b[3], pe1, y1));
        const double x0 = ow_div_y(br - ar[1] * b[1] - ar[2] * b[2] - ar[3] * b[3], pe0, y0);
        b[0] = rowb<2>(x0);
        // gen_tremolo.rs:2562-2713 for a sweep no limiter touches: ga == 1.0, i_nl - ga * b == i_trial
        double i_trial[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) i_trial[j] = i_nl[j] - b[j];
        const double v_trial = pq + c.kq[0] * i_trial[0] + c.kq[1] * i_trial[1] + c.kq[2] * i_trial[2] + c.kq[3] * i_trial[3];
        const double dv = v_trial - vd;
        // pnjlim acts when v_trial > vcrit and |dv| > 2 vt (which is > 1e-4); the step cap when some |dv| > 3.5
        bad |= __builtin_amdgcn_ballot_w64((v_trial > OW_T_VCRIT && fabs(dv) > OW_T_VT + OW_T_VT) || fabs(dv) > 3.5);
        if (__builtin_expect(bad != 0ull, 0)) {
            if (lane == 0) atomicAdd(&g_trem_row_cold[0], 1ull);
            double tv[7], ti[4];      // copies: the callee takes addresses, and only this cold block may put anything in memory
#pragma unroll
            for (int j = 0; j < 7; ++j) tv[j] = vp[j];
#pragma unroll
            for (int j = 0; j < 4; ++j) ti[j] = i_nl[j];
            const bool cv = __builtin_amdgcn_readfirstlane((int)trem_row_sweep_generic(tv, K, ti)) != 0;      // (every lane computes the same: keep the loop uniform)
#pragma unroll
            for (int j = 0; j < 4; ++j) i_nl[j] = ti[j];
            i_me = pick4(i_nl, q);
            if (cv) { converged = true; break; }
            continue;
        }
        i_me -= pick4(b, q);
#pragma unroll
        for (int j = 0; j < 4; ++j) i_nl[j] = i_trial[j];
        const double thr = 1e-3 * fmax(fabs(vd), fabs(vd + dv)) + 1e-6;
        if (__builtin_amdgcn_ballot_w64(fabs(dv) > thr) == 0ull) { converged = true; break; }
    }
    // v = v_pred + S_NI i_nl, row r
    double xa = va;
#pragma unroll
    for (int j = 0; j < 4; ++j) xa += c.sni[j] * i_nl[j];
    if (__builtin_expect(!converged, 0)) {
        st.be_fallbacks += 1u;
        if (lane == 0) atomicAdd(&g_trem_row_cold[1], 1ull);
        double v[7], tv[7], ta[4], tb[4], ti[4];
#pragma unroll
        for (int j = 0; j < 7; ++j) tv[j] = st.vf[j];
#pragma unroll
        for (int j = 0; j < 4; ++j) { ta[j] = st.ipf[j]; tb[j] = st.ipp[j]; ti[j] = i_nl[j]; }
        trem_row_be_fallback(K, tv, ta, tb, v, ti);
#pragma unroll
        for (int j = 0; j < 4; ++j) i_nl[j] = ti[j];
        xa = pick7(v, lane & 15);
    }
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!isfinite(xa)) != 0ull, 0)) {      // NaN reset to DC_OP (gen_tremolo.rs:3083-3093)
        const int r = lane & 15;
        xa = r < 7 ? OW_TREM_DC[r] : 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) { st.ipf[i] = OW_TREM_DC[7 + i]; i_nl[i] = OW_TREM_DC[7 + i]; }
    }
    // state: v = xa, i_pp = i_prev (as flushed), i_prev = i_nl; the flush of the next step is applied here, on the row form
    st.v_me = xa;
    const double xf = trem_flush(xa);
    static_for<0, 7>([&](auto I) { st.vf[I] = rowb<I>(xf); });
#pragma unroll
    for (int i = 0; i < 4; ++i) { st.ipp[i] = st.ipf[i]; st.ip[i] = i_nl[i]; st.ipf[i] = trem_flush(i_nl[i]); }
    return xa;
}

OW_DEV void trem_row_load(TremRow& t, const double* __restrict__ state, int lane) {      // I = 1 layout of the chain-state tremolo rows
    const int r = lane & 15;
    double v[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) { v[i] = state[CS_T_V + i]; t.vf[i] = trem_flush(v[i]); }
    t.v_me = pick7(v, r);
#pragma unroll
    for (int i = 0; i < 4; ++i) { t.ip[i] = state[CS_T_I + i]; t.ipf[i] = trem_flush(t.ip[i]); t.ipp[i] = state[CS_T_IP + i]; }
    t.env = state[CS_T_ENV];
    t.be_fallbacks = 0;
}
// state rows as the quad-lane kernels leave them (v, i_prev unflushed)
OW_DEV void trem_row_store_circuit(const TremRow& t, double* __restrict__ v_dst, double* __restrict__ ip_dst, double* __restrict__ ipp_dst, int lane) {
    if (lane < 7) v_dst[lane] = t.v_me;
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { ip_dst[i] = t.ip[i]; ipp_dst[i] = t.ipp[i]; }
    }
}
// LED drive and CdS envelope of one step (tremolo.rs:121-127), lane 0 of a row; the other lanes run it on their own row's voltage (bounded
// by the clamps, never read)
OW_DEV double trem_cell_drive_row(TremRow& st, const TremRowK& c, const OwConsts* __restrict__ K, int lane) {
    const double v_out = trem_osc_step_row(st, c, K, lane);
    const double led = clampd(OW_DIV_C(10.95 - v_out, 10.95 - 0.70), 0.0, 1.0);
    const double coeff = led > st.env ? K->ldr_attack : K->ldr_release;
    st.env = led + coeff * (st.env - led);
    return clampd(st.env, 0.0, 1.0);
}

// n oscillator steps without the cell (Tremolo::new's settle, CircuitState::default's warm-up): state rows in, state rows out
__global__ __launch_bounds__(64) void k_trem_settle_row(const OwConsts* __restrict__ K, double* __restrict__ state, long long n) {
    const int lane = threadIdx.x;
    TremRowK c;
    trem_row_consts(c, K, lane);
    TremRow t;
    trem_row_load(t, state, lane);
    for (long long i = 0; i < n; ++i) trem_osc_step_row(t, c, K, lane);
    trem_row_store_circuit(t, state + CS_T_V, state + CS_T_I, state + CS_T_IP, lane);
    if (lane == 0 && t.be_fallbacks) state[CS_T_BE] = bitsd(dbits(state[CS_T_BE]) + (uint64_t)t.be_fallbacks);
}

// k_trem_traj_extend (ow_trem_wide.h) on the row step: same arguments, same results
__global__ __launch_bounds__(64) void k_trem_traj_extend_row(const OwConsts* __restrict__ K, double* __restrict__ state, double* __restrict__ r, long long t0,
                                                             long long n, double* __restrict__ ckpt, unsigned long long* __restrict__ be) {
    const int lane = threadIdx.x;
    TremRowK c;
    trem_row_consts(c, K, lane);
    TremRow t;
    trem_row_load(t, state, lane);
    double drive = 0.0;
    for (long long i = 0; i < n; ++i) {
        if (((t0 + i) & (long long)(OW_TRAJ_CK - 1)) == 0) {
            double* ck = ckpt + (size_t)((t0 + i) / OW_TRAJ_CK) * OW_TRAJ_CKD;
            trem_row_store_circuit(t, ck, ck + 7, ck + 11, lane);
            if (lane == 0) ck[15] = t.env;
        }
        const uint32_t be0 = t.be_fallbacks;
        drive = trem_cell_drive_row(t, c, K, lane);
        if (lane == 0) {
            r[i] = drive;                                                  // the envelope for now; turned into R below
            if (__builtin_expect(t.be_fallbacks != be0, 0)) {
                const unsigned long long k = be[0];
                if (k < OW_TRAJ_BE_CAP) be[1 + k] = (unsigned long long)(t0 + i);
                be[0] = k + 1ull;
            }
        }
    }
    if (((t0 + n) & (long long)(OW_TRAJ_CK - 1)) == 0) {                     // the state AT a checkpoint boundary the store ends on
        double* ck = ckpt + (size_t)((t0 + n) / OW_TRAJ_CK) * OW_TRAJ_CKD;
        trem_row_store_circuit(t, ck, ck + 7, ck + 11, lane);
        if (lane == 0) ck[15] = t.env;
    }
    double r_ldr = state[CS_T_RLDR];
    if (n > 0) r_ldr = trem_cell_law(__shfl(drive, 0), K);
    __threadfence_block();                                                 // lane 0's envelope stores before the wavefront reads them back
    for (long long i = lane; i < n; i += 64) r[i] = trem_cell_law(r[i], K);
    trem_row_store_circuit(t, state + CS_T_V, state + CS_T_I, state + CS_T_IP, lane);
    if (lane == 0) { state[CS_T_ENV] = t.env; state[CS_T_RLDR] = r_ldr; }
}

}  // namespace owdev
