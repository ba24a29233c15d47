// openwurli-hip: Twin-T tremolo oscillator with FOUR LANES PER ENGINE, for pools too small to fill the chip.
//
// k_tremolo (lane = engine) is one long dependent chain: ~2 000 f64 instructions per 96 kHz sample, ~7 us for a lone wavefront.
// In a small pool that latency IS the block time (the oscillator is serial in time and nothing else can hide it).  Here the four
// lanes of a quad share one engine: lane q owns matrix row q (and q + 4), nonlinear port q (one of the four junction exponentials,
// one Jacobian row, one limiter port) and row q of the pivoted 4x4 elimination; values cross lanes with quad shuffles.  Every
// number is produced by the same operations in the same order as in trem_osc_step / trem_nr<false> / solve4 (ow_chain_dev.h), so
// the R stream is bit-identical to the scalar kernel's (tests/test_gpu_parity.py::test_tremolo_wide_is_bit_identical); the dependent
// chain is ~4x shorter.  State is replicated in the four lanes; the LDR law, the BE retry and the NaN reset run redundantly in all four.
#pragma once
#include <type_traits>
#include "ow_kernels.h"

namespace owdev {

// Quad cross-lane moves on the DPP path (one v_mov_b32 quad_perm per dword, a few cycles) instead of ds_bpermute (LDS crossbar,
// ~100 cycles): the whole point of this kernel is a short dependent chain.
template <int CTRL> OW_DEV double qperm(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
template <int SRC> OW_DEV double qget(double x) { return qperm<SRC * 0x55>(x); }   // value of lane SRC of this quad
// lane-dependent source: DPP must not run under a lane-divergent branch (it would read inactive lanes), so take all four and select
OW_DEV double qget_dyn(double x, int src) {
    const double a = qget<0>(x), b = qget<1>(x), c = qget<2>(x), d = qget<3>(x);
    return src == 0 ? a : (src == 1 ? b : (src == 2 ? c : d));
}
template <int B, int E, class F> OW_DEV void static_for(F&& f) {   // f(std::integral_constant<int, i>) for i = B .. E-1
    if constexpr (B < E) { f(std::integral_constant<int, B>{}); static_for<B + 1, E>(f); }
}
OW_DEV double qswap1(double x) { return qperm<0xB1>(x); }   // lanes 0<->1, 2<->3

// (k_tremolo pins its LDS matrix reads inside the loops with an opaque zero so that they do not turn into 200 live registers; here the
// compiler may hoist what it likes -- the register count does not move and the lone wavefront waits less: 4.93 -> 4.72 ms per block)
#define OW_TW_PIN(z) ((void)(z))
struct TremWide {   // replicated per quad lane
    double v[7], ip[4], ipp[4], env, r_ldr;
    uint32_t be_fallbacks;
};

// gen_tremolo.rs:2353-3116 (input == 0), quad-parallel.  Returns v[OUT].
__device__ inline double trem_osc_step_wide(TremWide& st, const OwConsts* __restrict__ K, const TremMats* __restrict__ M) {
    const int q = threadIdx.x & 3;
#pragma unroll
    for (int i = 0; i < 7; ++i) st.v[i] = st.v[i] + 1e-25 - 1e-25;
#pragma unroll
    for (int i = 0; i < 4; ++i) st.ip[i] = st.ip[i] + 1e-25 - 1e-25;
    const double (*__restrict__ an)[7] = M->a_neg;
    double rhs[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 15.0};   // replicated: 23 MACs (same statements as trem_osc_step)
    rhs[0] += an[0][0] * st.v[0];
    rhs[0] += an[0][1] * st.v[1];
    rhs[0] += an[0][3] * st.v[3];
    rhs[0] += an[0][5] * st.v[5];
    rhs[1] += an[1][0] * st.v[0];
    rhs[1] += an[1][1] * st.v[1];
    rhs[1] += an[1][2] * st.v[2];
    rhs[2] += an[2][1] * st.v[1];
    rhs[2] += an[2][2] * st.v[2];
    rhs[2] += an[2][3] * st.v[3];
    rhs[3] += an[3][0] * st.v[0];
    rhs[3] += an[3][2] * st.v[2];
    rhs[3] += an[3][3] * st.v[3];
    rhs[4] += an[4][4] * st.v[4];
    rhs[5] += an[5][0] * st.v[0];
    rhs[5] += an[5][5] * st.v[5];
    rhs[5] += an[5][6] * st.v[6];
    rhs[0] += -1.0 * st.ip[0];
    rhs[0] += -1.0 * st.ip[2];
    rhs[2] += -1.0 * st.ip[1];
    rhs[4] += 1.0 * st.ip[0];
    rhs[4] += 1.0 * st.ip[1];
    rhs[4] += -1.0 * st.ip[3];
    rhs[0] += (0.0 + 0.0) * (1.0 / 1.0e7);
    // v_pred = S rhs: lane q computes rows q and q + 4 (row 7 does not exist: lane 3 repeats row 3)
    const int rb = q + 4 < 7 ? q + 4 : 3;
    double va = 0.0, vb = 0.0;
#pragma unroll
    for (int j = 0; j < 7; ++j) { va += M->s[q][j] * rhs[j]; vb += M->s[rb][j] * rhs[j]; }
    double v_pred[7];
    static_for<0, 4>([&](auto I) { v_pred[I] = qget<I>(va); });
    static_for<0, 3>([&](auto I) { v_pred[4 + I] = qget<I>(vb); });
    double p[4];
    p[0] = 1.0 * v_pred[2] + -1.0 * v_pred[4];
    p[1] = -1.0 * v_pred[0] + 1.0 * v_pred[2];
    p[2] = 1.0 * v_pred[4];
    p[3] = -1.0 * v_pred[0] + 1.0 * v_pred[4];
    double i_nl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) i_nl[i] = 2.0 * st.ip[i] - st.ipp[i];
    const double pq = q == 0 ? p[0] : (q == 1 ? p[1] : (q == 2 ? p[2] : p[3]));

    // ---- Newton-Raphson, trapezoidal (trem_nr<false>): lane q = port q
    bool converged = false;
    for (int iter = 0; iter < 50; ++iter) {
        int z = 0;
        OW_TW_PIN(z);
        const double (*__restrict__ kk)[4] = M->k + z;
        const double kq0 = kk[q][0], kq1 = kk[q][1], kq2 = kk[q][2], kq3 = kk[q][3];
        // v_d as emitted (gen_tremolo.rs:2423-2438): v_d1 has no k[1][3] term, v_d2 no k[2][2] term
        double vd = pq + kq0 * i_nl[0] + kq1 * i_nl[1];
        { const double t = vd + kq2 * i_nl[2]; vd = (q != 2) ? t : vd; }      // selects, not branches: the lanes of a quad differ here
        { const double t = vd + kq3 * i_nl[3]; vd = (q != 1) ? t : vd; }
        // junction exponentials: lane 0 exp_be(Q1), 1 exp_bc(Q1), 2 exp_be(Q2), 3 exp_bc(Q2)   (bjt_eval: sign = nf = nr = 1)
        const double e_me = fast_exp(OW_DIV_C(1.0 * vd, 1.0 * OW_T_VT));
        const double e_ot = qswap1(e_me);
        const double exp_be = (q & 1) ? e_ot : e_me, exp_bc = (q & 1) ? e_me : e_ot;
        const double is = OW_T_IS, vt = OW_T_VT, beta_f = OW_T_BF, beta_r = OW_T_BR;
        const double i_cc = is * (exp_be - exp_bc);
        const double ib_fwd = is / beta_f * (exp_be - 1.0);
        const double ib_rev = is / beta_r * (exp_bc - 1.0);
        const double ic = 1.0 * (i_cc - is / beta_r * (exp_bc - 1.0));
        const double ib = 1.0 * (ib_fwd + ib_rev + 0.0 + 0.0);
        const double j0 = is / (1.0 * vt) * exp_be;
        const double j1 = -(is / (1.0 * vt)) * exp_bc - (is / (beta_r * (1.0 * vt))) * exp_bc;
        const double j2 = (is / (beta_f * (1.0 * vt))) * exp_be + 0.0;
        const double j3 = (is / (beta_r * (1.0 * vt))) * exp_bc + 0.0;
        const double inq = q == 0 ? i_nl[0] : (q == 1 ? i_nl[1] : (q == 2 ? i_nl[2] : i_nl[3]));
        const double f_me = inq - ((q & 1) ? ib : ic);
        const double jA = (q & 1) ? j2 : j0, jB = (q & 1) ? j3 : j1;
        const int r0 = q & 2;
        double ar[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) ar[j] = (j == q ? 1.0 : 0.0) - jA * kk[r0][j] - jB * kk[r0 + 1][j];
        double br = f_me;
        // ---- solve4, one row per lane (same pivot choice, same row exchange, same updates per element)
        bool singular = false;
        static_for<0, 4>([&](auto COL) {
            constexpr int col = COL;
            // Pivot choice.  The scan (first strict maximum of |J[row][col]|, rows col..3 in order) ends at the row it has ended at on
            // every sweep the oracle has seen -- row 2 for column 0, the diagonal otherwise (see solve4) -- exactly when that row's entry
            // beats the rows before it strictly and no row after it beats it.  Each lane tests that for its own row against the
            // expected row's entry (one broadcast, one compare); only if some lane of the wavefront disagrees is the scan itself run on
            // the gathered column.
            constexpr int ex = col == 0 ? 2 : col;
            const double pe = qget<ex>(ar[col]);
            const double ape = fabs(pe), amine = fabs(ar[col]);
            const bool ok_me = singular || q < col || q == ex || (q < ex ? (ape > amine) : !(amine > ape));
            int max_row = ex;
            double max_val = ape;
            const bool usual = __builtin_amdgcn_ballot_w64(!ok_me) == 0ull;
            if (!usual) {
                double cv[4];
                static_for<0, 4>([&](auto R) { cv[R] = qget<R>(ar[col]); });
                max_row = col;
                max_val = fabs(cv[col]);
#pragma unroll
                for (int row = col + 1; row < 4; ++row) {
                    const double v = fabs(cv[row]);
                    if (v > max_val) { max_val = v; max_row = row; }
                }
            }
            if (!singular && max_val < 1e-15) singular = true;
            if (!singular) {
                if (col == 0 && usual) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) ar[j] = qperm<0xC6>(ar[j]);      // rows 0 <-> 2: quad_perm [2, 1, 0, 3], one move per dword
                    br = qperm<0xC6>(br);
                } else
                if (max_row != col) {   // row exchange col <-> max_row
                    const int src = q == col ? max_row : (q == max_row ? col : q);
#pragma unroll
                    for (int j = 0; j < 4; ++j) ar[j] = qget_dyn(ar[j], src);
                    br = qget_dyn(br, src);
                }
                const double pivot = usual ? pe : qget<col>(ar[col]);      // the usual pivot row's entry is already here (before or after its move to lane col)
                const double pb = qget<col>(br);
                double prow[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) prow[j] = qget<col>(ar[j]);
                if (q > col) {
                    const double factor = ow_div(ar[col], pivot);
#pragma unroll
                    for (int j = col + 1; j < 4; ++j) ar[j] -= factor * prow[j];
                    br -= factor * pb;
                }
            }
        });
        double b[4] = {0.0, 0.0, 0.0, 0.0};
        if (!singular) {
            // the four back-substitution quotients form a serial chain; their divisors (lane i's own diagonal) are known now, so every lane
            // refines its reciprocal once, off that chain: ow_div_y(sum, d, y) is ow_div(sum, d) instruction for instruction
            const double my_diag = q == 0 ? ar[0] : (q == 1 ? ar[1] : (q == 2 ? ar[2] : ar[3]));
            const double y_diag = ow_rcp_refined(my_diag);
            static_for<0, 4>([&](auto KI) {
                constexpr int i = 3 - KI;
                double sum = br;
#pragma unroll
                for (int j = i + 1; j < 4; ++j) sum -= ar[j] * b[j];
                b[i] = qget<i>(ow_div_y(sum, ar[i], y_diag));      // (the reference's second |diagonal| < 1e-15 test cannot fire: see solve4)
            });
        }
        if (!singular) {   // gen_tremolo.rs:2562-2713
            double i_trial[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) i_trial[j] = i_nl[j] - b[j];
            int z2 = 0;
            OW_TW_PIN(z2);
            const double (*__restrict__ k2)[4] = M->k + z2;
            const double v_trial = pq + k2[q][0] * i_trial[0] + k2[q][1] * i_trial[1] + k2[q][2] * i_trial[2] + k2[q][3] * i_trial[3];
            const double dv_trial = v_trial - vd;
            const double v_lim = (fabs(dv_trial) > 1e-4) ? pnjlim(v_trial, vd, OW_T_VT, OW_T_VCRIT) : v_trial;
            const double dv_lim = v_lim - vd;
            double r_me = 1.0;           // ports that do not take part leave ga alone
            // an unlimited port has v_lim == v_trial, i.e. dv_lim is dv_trial bit for bit and the ratio is x / x == 1: the division only runs
            // when some port of the wavefront was limited (start-up transients)
            if (__builtin_amdgcn_ballot_w64(!(v_lim == v_trial)) != 0ull) {
                if (fabs(dv_trial) > 1e-15) r_me = (dv_trial * dv_lim < 0.0) ? 0.0 : clampd(ow_div(dv_lim, dv_trial), 0.0, 1.0);
            }
            // ga = min over the ports in port order with a strict "<" (a NaN ratio never wins, as in the scalar loop)
            double ga = 1.0;
            bool any_limited = false;
            if (__builtin_amdgcn_ballot_w64(r_me < 1.0) != 0ull) {               // otherwise every ratio is 1 (or NaN): nothing is below ga
                static_for<0, 4>([&](auto J) {
                    const double rj = qget<J>(r_me);
                    if (rj < ga) { ga = rj; any_limited = true; }
                });
            }
            // the 3.5 V step cap (gen_tremolo.rs:2655-2668): max over the ports only matters when some port is above it
            const double adv = fabs(dv_trial * ga);
            if (__builtin_amdgcn_ballot_w64(adv > 3.5) != 0ull) {
                double max_dv = qget<0>(adv);
                max_dv = fmax(max_dv, qget<1>(adv)); max_dv = fmax(max_dv, qget<2>(adv)); max_dv = fmax(max_dv, qget<3>(adv));
                if (max_dv > 3.5) { ga *= fmax(ow_div(3.5, max_dv), 0.1); any_limited = true; }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) i_nl[j] -= ga * b[j];
            if (!any_limited) {
                const double dv = dv_trial * ga;
                const double thr = 1e-3 * fmax(fabs(vd), fabs(vd + dv)) + 1e-6;
                const bool ok_me = !(fabs(dv) > thr);
                const uint64_t bal = __ballot(ok_me);                                    // all four ports of this quad within tolerance?
                if (((bal >> (threadIdx.x & 60)) & 0xFull) == 0xFull) { converged = true; break; }
            }
        } else {   // singular Jacobian: damped fallback (gen_tremolo.rs:2715-2733); `singular` is uniform over the quad, so the gather may sit here
            double f[4];
            static_for<0, 4>([&](auto J) { f[J] = qget<J>(f_me); });
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double cl = fmax(fabs(i_nl[j]) * 0.1, 0.01);
                i_nl[j] -= clampd(f[j] * 0.5, -cl, cl);
            }
        }
    }
    // v = v_pred + S_NI i_nl: rows q and q + 4 per lane, then replicated
    double xa = (q == 0 ? v_pred[0] : (q == 1 ? v_pred[1] : (q == 2 ? v_pred[2] : v_pred[3])));
    double xb = (rb == 4 ? v_pred[4] : (rb == 5 ? v_pred[5] : (rb == 6 ? v_pred[6] : v_pred[3])));
#pragma unroll
    for (int j = 0; j < 4; ++j) { xa += M->s_ni[q][j] * i_nl[j]; xb += M->s_ni[rb][j] * i_nl[j]; }
    double v[7];
    static_for<0, 4>([&](auto I) { v[I] = qget<I>(xa); });
    static_for<0, 3>([&](auto I) { v[4 + I] = qget<I>(xb); });
    if (__builtin_expect(!converged, 0)) {
        st.be_fallbacks += 1u;
        trem_be_fallback(K, st.v, st.ip, st.ipp, v, i_nl);      // redundantly in the four lanes
    }
    bool finite = true;
#pragma unroll
    for (int i = 0; i < 7; ++i) finite = finite && isfinite(v[i]);
    if (__builtin_expect(!finite, 0)) {
#pragma unroll
        for (int i = 0; i < 7; ++i) st.v[i] = OW_TREM_DC[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) { st.ip[i] = OW_TREM_DC[7 + i]; st.ipp[i] = OW_TREM_DC[7 + i]; }
        return OW_TREM_DC[0];
    }
#pragma unroll
    for (int i = 0; i < 7; ++i) st.v[i] = v[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) { st.ipp[i] = st.ip[i]; st.ip[i] = i_nl[i]; }
    return v[0];
}

// Tremolo::process, oscillator half (tremolo.rs:121-146), split in two: the LED drive / CdS envelope recurrence (serial), and the
// power-law cell resistance (tremolo.rs:128-146), which is a pure function of the envelope and is NOT fed back into the oscillator,
// so the kernel applies it afterwards with the quad's four lanes working on four different samples.
__device__ inline double trem_cell_drive_wide(TremWide& st, const OwConsts* __restrict__ K, const TremMats* __restrict__ M) {
    const double v_out = trem_osc_step_wide(st, K, M);
    const double led = clampd(OW_DIV_C(10.95 - v_out, 10.95 - 0.70), 0.0, 1.0);
    const double coeff = led > st.env ? K->ldr_attack : K->ldr_release;
    st.env = led + coeff * (st.env - led);
    return clampd(st.env, 0.0, 1.0);
}
OW_DEV double trem_cell_law(double drive, const OwConsts* __restrict__ K) {
    if (drive < 1e-6) return 1000000.0;
    return exp(K->ln_r_max + K->ln_min_minus_max * pow(drive, 0.9));
}

OW_DEV void trem_wide_load(TremWide& t, const double* __restrict__ cs, int I, int e) {
#pragma unroll
    for (int i = 0; i < 7; ++i) t.v[i] = CSF(CS_T_V + i);
#pragma unroll
    for (int i = 0; i < 4; ++i) { t.ip[i] = CSF(CS_T_I + i); t.ipp[i] = CSF(CS_T_IP + i); }
    t.env = CSF(CS_T_ENV); t.r_ldr = CSF(CS_T_RLDR);
    t.be_fallbacks = 0;
}
OW_DEV void trem_wide_store(const TremWide& t, double* __restrict__ cs, int I, int e) {
#pragma unroll
    for (int i = 0; i < 7; ++i) CSF(CS_T_V + i) = t.v[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) { CSF(CS_T_I + i) = t.ip[i]; CSF(CS_T_IP + i) = t.ipp[i]; }
    CSF(CS_T_ENV) = t.env; CSF(CS_T_RLDR) = t.r_ldr;
    if (t.be_fallbacks) CSF(CS_T_BE) = bitsd(dbits(CSF(CS_T_BE)) + (uint64_t)t.be_fallbacks);
}

// 16 engines per wavefront.  settle_only: n oscillator steps without the LDR law (k_trem_settle); otherwise R[n] for n_os samples.
template <bool SETTLE>
// leaders[0..n_lead): the engines that own a tremolo phase group (see "Tremolo phase groups", openwurli_hip.hip); column e of rbuf is
// written for those engines only, every other engine of a group reads its leader's column.
__global__ __launch_bounds__(64) void k_tremolo_wide(const OwConsts* __restrict__ K, double* __restrict__ cs, double* __restrict__ rbuf, int I,
                                                     long long n, const uint32_t* __restrict__ leaders, int n_lead) {
    __shared__ TremMats M;
    trem_mats_load(&M, K, threadIdx.x, 64);
    __syncthreads();
    const int el = threadIdx.x >> 2, q = threadIdx.x & 3;
    const int idx = blockIdx.x * 16 + el;
    const bool valid = idx < n_lead;
    const int e = (int)leaders[valid ? idx : n_lead - 1];      // idle quads shadow the last engine (no divergence), and store nothing
    TremWide t;
    trem_wide_load(t, cs, I, e);
    double drive = 0.0;
    for (long long i = 0; i < n; ++i) {
        int z = 0;
        OW_TW_PIN(z);
        if (SETTLE) trem_osc_step_wide(t, K, &M + z);
        else {
            drive = trem_cell_drive_wide(t, K, &M + z);
            if (valid && q == 0) rbuf[(size_t)i * I + e] = drive;      // the envelope for now; turned into R[n] below
        }
    }
    if (!SETTLE) {
        if (n > 0) t.r_ldr = trem_cell_law(drive, K);                 // Tremolo.r_ldr after the block = R of its last sample
        __threadfence_block();                                        // lane 0's envelope stores before the quad reads them back
        if (valid)
            for (long long i = q; i < n; i += 4) rbuf[(size_t)i * I + e] = trem_cell_law(rbuf[(size_t)i * I + e], K);
    }
    if (valid && q == 0) trem_wide_store(t, cs, I, e);
}


// ------------------------------------------------------------------ shared trajectory (openwurli_hip.hip `TremTraj`)
// r_ldr[t] of the Twin-T / CdS cell for t = 0, 1, 2, ... calls of Tremolo::process after Tremolo::new (tremolo.rs:83-146), computed ONCE per
// (device, chain rate) and read by every engine at its own t (ow_kernels.h, OwTremSrc).  One wavefront extends it: its sixteen quads run
// the same oscillator (the quad-lane step above; idle quads would diverge, shadows do not), lane 0 keeps the LED envelope of every step,
// and the power law -- which is not fed back -- is applied afterwards by all 64 lanes on 64 samples at a time.
//   state   18 rows (I = 1 layout of the chain-state tremolo rows) at sample t0; advanced to t0 + n
//   r       &trajectory[t0]
//   ckpt    oscillator state in front of every OW_TRAJ_CK-th sample ([t / CK][OW_TRAJ_CKD]: v[7] ip[4] ipp[4] env): an engine that
//           outlives the store's cap continues from the checkpoint below its t (k_trem_from_ckpt)
//   be      [0] = backward-Euler fallbacks so far, [1 + k] = sample index of the k-th (k < OW_TRAJ_BE_CAP): ow_diag's counter of an
//           engine at t is the number of entries below t
#define OW_TRAJ_CK 4096
#define OW_TRAJ_CKD 16
#define OW_TRAJ_BE_CAP 1023
__global__ void k_trem_state_dc(double* __restrict__ state) {     // CircuitState at DC_OP, cell at rest (k_chain_init's tremolo rows)
    const int f = threadIdx.x;
    if (f < 7) state[CS_T_V + f] = OW_TREM_DC[f];
    else if (f < 11) { state[CS_T_I + f - 7] = OW_TREM_DC[f]; state[CS_T_IP + f - 7] = OW_TREM_DC[f]; }
    else if (f == 11) { state[CS_T_ENV] = 0.0; state[CS_T_RLDR] = 1000000.0; state[CS_T_BE] = bitsd(0ull); }
}
__global__ __launch_bounds__(64) void k_trem_traj_extend(const OwConsts* __restrict__ K, double* __restrict__ state, double* __restrict__ r, long long t0,
                                                         long long n, double* __restrict__ ckpt, unsigned long long* __restrict__ be) {
    __shared__ TremMats M;
    trem_mats_load(&M, K, threadIdx.x, 64);
    __syncthreads();
    const int lane = threadIdx.x;
    TremWide t;
    trem_wide_load(t, state, 1, 0);
    double drive = 0.0;
    for (long long i = 0; i < n; ++i) {
        if ((((t0 + i) & (long long)(OW_TRAJ_CK - 1)) == 0) && lane == 0) {
            double* c = ckpt + (size_t)((t0 + i) / OW_TRAJ_CK) * OW_TRAJ_CKD;
#pragma unroll
            for (int k = 0; k < 7; ++k) c[k] = t.v[k];
#pragma unroll
            for (int k = 0; k < 4; ++k) { c[7 + k] = t.ip[k]; c[11 + k] = t.ipp[k]; }
            c[15] = t.env;
        }
        const uint32_t be0 = t.be_fallbacks;
        drive = trem_cell_drive_wide(t, K, &M);
        if (lane == 0) {
            r[i] = drive;                                                  // the envelope for now; turned into R below
            if (__builtin_expect(t.be_fallbacks != be0, 0)) {
                const unsigned long long k = be[0];
                if (k < OW_TRAJ_BE_CAP) be[1 + k] = (unsigned long long)(t0 + i);
                be[0] = k + 1ull;
            }
        }
    }
    if ((((t0 + n) & (long long)(OW_TRAJ_CK - 1)) == 0) && lane == 0) {      // the state AT a checkpoint boundary the store ends on (an engine may leave from there)
        double* c = ckpt + (size_t)((t0 + n) / OW_TRAJ_CK) * OW_TRAJ_CKD;
#pragma unroll
        for (int k = 0; k < 7; ++k) c[k] = t.v[k];
#pragma unroll
        for (int k = 0; k < 4; ++k) { c[7 + k] = t.ip[k]; c[11 + k] = t.ipp[k]; }
        c[15] = t.env;
    }
    if (n > 0) t.r_ldr = trem_cell_law(drive, K);
    __threadfence_block();                                                 // lane 0's envelope stores before the wavefront reads them back
    for (long long i = lane; i < n; i += 64) r[i] = trem_cell_law(r[i], K);
    if (lane == 0) { t.be_fallbacks = 0; trem_wide_store(t, state, 1, 0); }
}

// Engines leaving the trajectory (lane = engine): engine engines[i] stands at sample tpos[i]; its own oscillator rows are rebuilt from the
// checkpoint below tpos[i] and stepped up to it with the lane = engine cell (bit-identical to the quad-lane one), after which the engine
// is a phase group of one.  be_count[i]: value of its fallback counter.
__global__ __launch_bounds__(64) void k_trem_from_ckpt(const OwConsts* __restrict__ K, const double* __restrict__ traj_r, const double* __restrict__ ckpt,
                                                       double* __restrict__ cs, int I, const uint32_t* __restrict__ engines, const long long* __restrict__ tpos,
                                                       const unsigned long long* __restrict__ be_count, int n) {
    __shared__ TremMats M;
    __shared__ TremPark P;
    trem_mats_load(&M, K, threadIdx.x, blockDim.x);
    __syncthreads();
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int e = (int)engines[idx];
    const int ln = threadIdx.x & 63;
    const long long tp = tpos[idx];
    const double* c = ckpt + (size_t)(tp / OW_TRAJ_CK) * OW_TRAJ_CKD;
    TremState t;
#pragma unroll
    for (int k = 0; k < 7; ++k) P.v[k][ln] = c[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) { P.ip[k][ln] = c[7 + k]; P.ipp[k][ln] = c[11 + k]; }
    t.env = c[15];
    t.r_ldr = tp > 0 ? traj_r[tp - 1] : 1000000.0;
    t.be_fallbacks = 0;
    const long long steps = tp % OW_TRAJ_CK;
    for (long long i = 0; i < steps; ++i) {
        int z = 0;
        asm volatile("" : "+v"(z));
        trem_cell_r(t, &P, K, &M + z);
    }
    t.be_fallbacks = 0;                                                    // already part of be_count
    CSF(CS_T_BE) = bitsd((uint64_t)be_count[idx]);
    trem_store(t, &P, cs, I, e);
}

// rows of R the last block consumed, for engines on the trajectory (ow_pool_read_tremolo_r): out[e][i] = traj[t_end(e) - n_os + i]
__global__ void k_trem_traj_gather(const double* __restrict__ traj_at_clock, const long long* __restrict__ birth, int I, long long n_os, double* __restrict__ out) {
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int e = (int)(k / n_os);
    const long long i = k - (long long)e * n_os;
    if (e >= I) return;
    const long long b = birth[e];
    out[(size_t)e * n_os + i] = b == OW_OFF_TRAJ ? 0.0 : traj_at_clock[i - n_os - b];
}
// birth[e] += delta for e in [e0, e0 + ne) on the trajectory (a sub-range rendered on its own grows older than the pool clock says)
__global__ void k_trem_birth_shift(long long* __restrict__ birth, int e0, int ne, long long delta) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ne) return;
    const long long b = birth[e0 + k];
    if (b != OW_OFF_TRAJ) birth[e0 + k] = b + delta;
}

}  // namespace owdev
