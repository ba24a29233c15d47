// openwurli-hip: legacy DK preamp step with FOUR LANES PER SOLVER STATE, for the job paths (batch render, render-midi) whose
// few hundred jobs leave the chip idle and whose run time is the serial latency of dk_step (ow_chain_dev.h): ~840 dependent-ish
// instructions per 96 kHz sample for a lone wavefront.
//
// The four lanes of a quad hold the same state.  What is split is the dense part of the step:
//   * v_pred_base = S * rhs (8x8): lane q forms rows q and q + 4 (its two rows of S live in its own registers) and the quad
//     all-gathers the eight sums with DPP quad moves;
//   * the two junction exponentials of every Newton iteration (and of the final current evaluation): even lanes take Q1, odd
//     lanes Q2, one exchange.
// Everything else (the sparse A_neg product, Sherman-Morrison scalars, the 2x2 Newton update, the state update) is cheap and stays
// replicated, statement for statement as in dk_step.  Every number is produced by the same operations in the same order, so the
// result is bit-identical to dk_step's (tests/test_gpu_parity.py::test_chain_wide_is_bit_identical).
#pragma once
#include "ow_trem_wide.h"

namespace owdev {

// rows q and q + 4 (lane q of the quad) of S, of the two S N_i column differences and of S's feedback column, loop-invariant
// ... and the step's wave-uniform constants, one copy per lane IN VECTOR REGISTERS: these kernels run one wavefront per SIMD and their
// time is the dependent latency of one sample, so the twelve s_load / s_waitcnt round trips per sample with which the pool-sized kernel
// keeps its constants out of the SGPR file (k_reload) sit right on the serial path here; 44 doubles of 512 available registers do not.
struct DkWideRows {
    double s_lo[8], s_hi[8], c1_lo, c1_hi, c2_lo, c2_hi, fb_lo, fb_hi;
    double an[20];          // A_neg's structural non-zeros in dk_step's order of use
    double two_w[8], fb_col[4], k[4], nv_sfb[2], sfb_ni[2], g_cin, s_fb_fb, gc_1pc, c_cin;
};
OW_DEV void vgpr_pin(double& x) { asm volatile("" : "+v"(x)); }      // the value stays in a vector register: no rematerialised scalar load

__device__ inline void dk_wide_rows_load(DkWideRows& R, const OwConsts* __restrict__ K, int q) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { R.s_lo[j] = K->p_s[q][j]; R.s_hi[j] = K->p_s[q + 4][j]; }
    // (s_base[i][EMIT1] - s_base[i][COLL1]) and (s_base[i][EMIT2] - s_base[i][COLL2]) of dk_step's last loop: differences of constants
    R.c1_lo = K->p_sni_d1[q];                      R.c1_hi = K->p_sni_d1[q + 4];
    R.c2_lo = K->p_sni_d2[q];                      R.c2_hi = K->p_sni_d2[q + 4];
    R.fb_lo = K->p_s_fb_col[q];                    R.fb_hi = K->p_s_fb_col[q + 4];
    const double (*__restrict__ an)[8] = K->p_a_neg;
    const double a20[20] = {an[0][0], an[0][2], an[1][1], an[1][7], an[2][0], an[2][2], an[2][5], an[3][3], an[3][4], an[4][3],
                            an[4][4], an[5][2], an[5][5], an[5][6], an[6][5], an[6][6], an[6][7], an[7][1], an[7][6], an[7][7]};
#pragma unroll
    for (int i = 0; i < 20; ++i) { R.an[i] = a20[i]; vgpr_pin(R.an[i]); }
#pragma unroll
    for (int i = 0; i < 8; ++i) { R.two_w[i] = K->p_two_w[i]; vgpr_pin(R.two_w[i]); }
#pragma unroll
    for (int i = 0; i < 4; ++i) { R.fb_col[i] = K->p_s_fb_col[i]; vgpr_pin(R.fb_col[i]); }
    R.k[0] = K->p_k[0][0]; R.k[1] = K->p_k[0][1]; R.k[2] = K->p_k[1][0]; R.k[3] = K->p_k[1][1];
    R.nv_sfb[0] = K->p_nv_sfb[0]; R.nv_sfb[1] = K->p_nv_sfb[1]; R.sfb_ni[0] = K->p_sfb_ni[0]; R.sfb_ni[1] = K->p_sfb_ni[1];
    R.g_cin = K->p_g_cin; R.s_fb_fb = K->p_s_fb_fb; R.gc_1pc = K->p_gc_1pc; R.c_cin = K->p_c_cin;
#pragma unroll
    for (int i = 0; i < 4; ++i) vgpr_pin(R.k[i]);
    vgpr_pin(R.nv_sfb[0]); vgpr_pin(R.nv_sfb[1]); vgpr_pin(R.sfb_ni[0]); vgpr_pin(R.sfb_ni[1]);
    vgpr_pin(R.g_cin); vgpr_pin(R.s_fb_fb); vgpr_pin(R.gc_1pc); vgpr_pin(R.c_cin);
#pragma unroll
    for (int j = 0; j < 8; ++j) { vgpr_pin(R.s_lo[j]); vgpr_pin(R.s_hi[j]); }
    vgpr_pin(R.c1_lo); vgpr_pin(R.c1_hi); vgpr_pin(R.c2_lo); vgpr_pin(R.c2_hi); vgpr_pin(R.fb_lo); vgpr_pin(R.fb_hi);
}

// ic, gm of both junctions with one exponential per lane: even lanes of the quad evaluate vn0, odd lanes vn1 (dk_ic_gm, :686-690)
OW_DEV void dk_ic_gm_pair(int q, double vn0, double vn1, double& ic0, double& gm0, double& ic1, double& gm1) {
    double ic, gm;
    dk_ic_gm((q & 1) ? vn1 : vn0, ic, gm);
    ic0 = qperm<0xA0>(ic); gm0 = qperm<0xA0>(gm);   // lanes (0,1,2,3) read lanes (0,0,2,2)
    ic1 = qperm<0xF5>(ic); gm1 = qperm<0xF5>(gm);   // lanes (0,1,2,3) read lanes (1,1,3,3)
}

// dk_step (dk_preamp_legacy.rs:447-554), quad-parallel.  `st` is replicated in the four lanes.
__device__ inline double dk_step_wide(DkSt& st, const DkWideRows& R, int q, double input, double g_ldr, double g_ldr_prev,
                                      const OwConsts* __restrict__ /*K0: every constant is in R*/) {
    const double* an = R.an;
    const double* v = st.v;
    double rhs[8];
    rhs[0] = an[0] * v[0] + an[1] * v[2];
    rhs[1] = an[2] * v[1] + an[3] * v[7];
    rhs[2] = an[4] * v[0] + an[5] * v[2] + an[6] * v[5];
    rhs[3] = an[7] * v[3] + an[8] * v[4];
    rhs[4] = an[9] * v[3] + an[10] * v[4];
    rhs[5] = an[11] * v[2] + an[12] * v[5] + an[13] * v[6];
    rhs[6] = an[14] * v[5] + an[15] * v[6] + an[16] * v[7];
    rhs[7] = an[17] * v[1] + an[18] * v[6] + an[19] * v[7];
    rhs[7] -= g_ldr_prev * st.v[7];
    const double cin_now = R.g_cin * input + st.j_cin;
    rhs[0] += cin_now + st.cin_prev;
    rhs[1] += st.i_nl[0];
    rhs[2] -= st.i_nl[0];
    rhs[3] += st.i_nl[1];
    rhs[5] -= st.i_nl[1];
#pragma unroll
    for (int i = 0; i < 8; ++i) rhs[i] += R.two_w[i];
    // v_pred_base = S rhs: this lane's two rows, then the quad's eight
    double lo = 0.0, hi = 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { lo += R.s_lo[j] * rhs[j]; hi += R.s_hi[j] * rhs[j]; }
    // rows 0..3 (they feed p) and row 7 (the Sherman-Morrison scalar) go to every lane; rows 4..6 are only needed by their owners
    double vpb[4];
    static_for<0, 4>([&](auto i) { vpb[i] = qget<i>(lo); });
    const double vpb7 = qget<3>(hi);
    const double sm_k = ow_div(g_ldr, 1.0 + R.s_fb_fb * g_ldr);
    const double sm_vpred = sm_k * vpb7;
    double v_pred[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v_pred[i] = vpb[i] - sm_vpred * R.fb_col[i];
    const double vp_lo = lo - sm_vpred * R.fb_lo, vp_hi = hi - sm_vpred * R.fb_hi;      // v_pred of this lane's rows q, q + 4
    const double p0 = v_pred[0] - v_pred[1], p1 = v_pred[2] - v_pred[3];
    const double k00 = R.k[0] - sm_k * R.nv_sfb[0] * R.sfb_ni[0];
    const double k01 = R.k[1] - sm_k * R.nv_sfb[0] * R.sfb_ni[1];
    const double k10 = R.k[2] - sm_k * R.nv_sfb[1] * R.sfb_ni[0];
    const double k11 = R.k[3] - sm_k * R.nv_sfb[1] * R.sfb_ni[1];
    double vn0 = st.v_nl[0], vn1 = st.v_nl[1];
    // dk_step's loop (ow_chain_dev.h): (ic, gm) hold the evaluation at (vn0, vn1) -- the state's own on entry, a fresh one after every
    // update -- and the quads of the wavefront sweep in one wave-uniform loop (a finished quad is no longer moved), so the quad moves
    // inside dk_ic_gm_pair always find their source lanes active.  `done` is the same in the four lanes of a quad (replicated values).
    double ic0 = st.i_nl[0], ic1 = st.i_nl[1], gm0 = st.gm[0], gm1 = st.gm[1];
    bool done = false;
    for (int iter = 0; iter < 6; ++iter) {
        const double f0 = vn0 - p0 - k00 * ic0 - k01 * ic1;
        const double f1 = vn1 - p1 - k10 * ic0 - k11 * ic1;
        done = done || (fabs(f0) < 1e-9 && fabs(f1) < 1e-9);
        if (__builtin_amdgcn_ballot_w64(!done) == 0ull) break;
        const double j00 = 1.0 - k00 * gm0, j01 = -k01 * gm1, j10 = -k10 * gm0, j11 = 1.0 - k11 * gm1;
        const double det = j00 * j11 - j01 * j10;
        done = done || fabs(det) < 1e-30;
        const double inv_det = ow_div(1.0, det);
        const double n0 = vn0 - inv_det * (j11 * f0 - j01 * f1);
        const double n1 = vn1 - inv_det * (j00 * f1 - j10 * f0);
        vn0 = done ? vn0 : n0;
        vn1 = done ? vn1 : n1;
        dk_ic_gm_pair(q, vn0, vn1, ic0, gm0, ic1, gm1);
    }
    const double dot = R.sfb_ni[0] * ic0 + R.sfb_ni[1] * ic1;
    // v = v_pred + S N_i i_c - sm_k (s_fb N_i . i_c) s_fb_col: the same expression for every row with the row's constants -- each
    // lane forms its rows q and q + 4, the quad gathers the eight
    const double v_lo = vp_lo + (ic0 * R.c1_lo + ic1 * R.c2_lo) - sm_k * dot * R.fb_lo;
    const double v_hi = vp_hi + (ic0 * R.c1_hi + ic1 * R.c2_hi) - sm_k * dot * R.fb_hi;
    static_for<0, 4>([&](auto i) { st.v[i] = qget<i>(v_lo); st.v[i + 4] = qget<i>(v_hi); });
    st.cin_prev = cin_now;
    const double dv_cin = input - st.v[0];
    st.j_cin = -R.gc_1pc * dv_cin - R.c_cin * st.j_cin;
    st.i_nl[0] = ic0; st.i_nl[1] = ic1;
    st.v_nl[0] = vn0; st.v_nl[1] = vn1;
    st.gm[0] = gm0; st.gm[1] = gm1;
    return st.v[6];
}

// k_job_chain<false> with a quad per solver state: 8 jobs per wavefront (main quads in lanes 0-31, shadow quads in lanes 32-63).
#define OW_WCHUNK 64
__global__ __launch_bounds__(64) void k_job_chain_wide(const OwConsts* __restrict__ K, const OwJobDev* __restrict__ jobs, const double* __restrict__ reed,
                                                       double* __restrict__ out, int n_jobs, long long n, long long stride) {
    __shared__ double tin[8 * (OW_WCHUNK + 1)];
    __shared__ double tout[8 * (OW_WCHUNK + 1)];
    const int lane = threadIdx.x;
    const int q = lane & 3, jl = (lane & 31) >> 2, role = lane >> 5;
    const int jb = blockIdx.x * 8;
    const int j = jb + jl;
    const bool valid = j < n_jobs;
    const OwJobDev jd = jobs[valid ? j : n_jobs - 1];
    const int osr = K->oversample ? 2 : 1;
    const double sr = K->sr;
    DkWideRows R;
    dk_wide_rows_load(R, K, q);

    // DkPreamp::new(preamp_sr); preamp.reset(); preamp.set_ldr_resistance(r_ldr)  (main.rs:432-441)
    DkSt st;
    double r_ldr = 1000000.0;
    double g_ldr = 1.0 / r_ldr, g_prev = g_ldr;
    dk_dc_reset(K, r_ldr, st);
    {
        const double r_new = fmax(jd.r_ldr, 1000.0);
        if (fabs(r_new - r_ldr) > 0.01) { r_ldr = r_new; g_ldr = 1.0 / r_new; }
    }
    auto preamp_step = [&](double x) -> double {
        const double o = dk_step_wide(st, R, q, x, g_ldr, g_prev, K);
        g_prev = g_ldr;
        const double other = xor32(o);
        double res = role ? (other - o) : (o - other);
        if (!isfinite(res)) {
            dk_dc_reset(K, r_ldr, st); g_ldr = 1.0 / r_ldr; g_prev = g_ldr;
            res = 0.0;
        }
        return res;
    };
    double ua[3] = {0, 0, 0}, ub[3] = {0, 0, 0}, da[3] = {0, 0, 0}, db[3] = {0, 0, 0}, dd = 0.0;
    SpeakerSt sp;                                      // Speaker::new(sr); set_character(c)  (main.rs:483-484)
    sp.character = 1.0; sp.ts = 0.0;
    sp.hpf.s1 = sp.hpf.s2 = sp.lpf.s1 = sp.lpf.s2 = 0.0;
    speaker_update(sp, sr);
    speaker_set_character(sp, jd.speaker, sr);
    const double vol2_a = jd.volume;

    for (long long base = 0; base < n; base += OW_WCHUNK) {
        const int cn = (int)((n - base) < OW_WCHUNK ? (n - base) : OW_WCHUNK);
        for (int r = 0; r < 8; ++r) {
            double x = 0.0;
            if (jb + r < n_jobs && lane < cn) x = reed[(size_t)(jb + r) * stride + base + lane];
            tin[r * (OW_WCHUNK + 1) + lane] = x;
        }
        __syncthreads();
        for (int s = 0; s < cn; ++s) {
            const double x = tin[jl * (OW_WCHUNK + 1) + s];
            double pre;
            if (osr == 2) {                            // main.rs:445-466: per-sample up(1) -> 2x process -> down(1)
                const double a = allpass3(OW_OS_A0, OW_OS_A1, OW_OS_A2, ua, x);
                const double b = allpass3(OW_OS_B0, OW_OS_B1, OW_OS_B2, ub, x);
                double p[2];
                const double in[2] = {role ? 0.0 : a, role ? 0.0 : b};
                for (int k = 0; k < 2; ++k) p[k] = preamp_step(in[k]);
                const double fa = allpass3(OW_OS_A0, OW_OS_A1, OW_OS_A2, da, p[0]);
                const double fb = allpass3(OW_OS_B0, OW_OS_B1, OW_OS_B2, db, p[1]);
                pre = (fa + dd) * 0.5;
                dd = fb;
            } else {
                pre = preamp_step(role ? 0.0 : x);
            }
            // main.rs:487-496: volume^2 (audio taper) -> optional power amp at base rate -> speaker -> PSG
            const double att = pre * vol2_a * vol2_a;
            const double amp = jd.poweramp ? power_amp(att) : att;
            const double y = speaker_process(sp, amp, K->spk_thermal_alpha) * 7.498942093324558;
            if (role == 0 && q == 0) tout[jl * (OW_WCHUNK + 1) + s] = y;
        }
        __syncthreads();
        for (int r = 0; r < 8; ++r)
            if (jb + r < n_jobs && lane < cn) out[(size_t)(jb + r) * stride + base + lane] = tout[r * (OW_WCHUNK + 1) + lane];
        __syncthreads();
    }
}

// k_preamp (ow_kernels.h) with a quad per solver state, for pools too small to fill the chip: 8 engines per wavefront (main quads in
// lanes 0-31, shadow quads in lanes 32-63).  Same interface, same arithmetic statement for statement (dk_step_wide is bit-identical to
// dk_step): tests/test_gpu_parity.py::test_preamp_wide_is_bit_identical.  In a pool of a few hundred engines the preamp's serial
// latency is the block time once the tremolo is out of the way (paced single instance: 314 of the 425 us of a 64-sample buffer).
__global__ __launch_bounds__(64) void k_preamp_wide(const OwConsts* __restrict__ K, double* __restrict__ cs, const OwEngineArgs* __restrict__ args,
                                                    const OwEngineOut* __restrict__ eout, const double* __restrict__ sum, const OwTremSrc tsrc,
                                                    double* __restrict__ pre, int I, int L, int Lcap, int e0, int ne) {
    __shared__ double tile[8 * (OW_WCHUNK + 1)];
    const int lane = threadIdx.x;
    const int q = lane & 3, el = (lane & 31) >> 2, role = lane >> 5;
    const int eb = e0 + blockIdx.x * 8;
    const int e = eb + el;
    const bool valid = e < e0 + ne;
    const int ec = valid ? e : (e0 + ne - 1);
    const int osr = K->oversample ? 2 : 1;
    DkWideRows R;
    dk_wide_rows_load(R, K, q);

    DkSt st;
    double ua[3], ub[3];
    double r_ldr, g_ldr, g_prev;
    Smoother sd;
    {
        const int e = ec;
        smoother_load(sd, cs, I, e, CS_SM_DEPTH);
        if (args[e].set_flags & 1u) sd.retarget(args[e].depth_target, K->ramp_samples);
        dk_load(st, cs, I, e, role ? CS_P_SHADOW : CS_P_MAIN);
        for (int i = 0; i < 3; ++i) { ua[i] = CSF(CS_OS_UA + i); ub[i] = CSF(CS_OS_UB + i); }
        r_ldr = CSF(CS_P_RLDR); g_ldr = CSF(CS_P_GLDR); g_prev = CSF(CS_P_GPREV);
        const uint64_t fl = dbits(CSF(CS_FLAGS));
        if (fl & 1ull) {  // deferred preamp.reset() + oversampler.reset() from the output NaN guard (engine.rs:450-457)
            dk_dc_reset(K, r_ldr, st);
            g_ldr = 1.0 / r_ldr; g_prev = g_ldr;
            for (int i = 0; i < 3; ++i) { ua[i] = 0.0; ub[i] = 0.0; }
        }
    }
    uint32_t nan_resets = 0;
    const TremCol rc = trem_col(tsrc, I, ec);
    double rnx[2];                                               // R one host sample ahead (registers): the global load is off the serial path
    rnx[0] = trem_col_at(rc, 0u);
    rnx[1] = osr == 2 ? trem_col_at(rc, 1u) : 0.0;
    for (int base = 0; base < L; base += OW_WCHUNK) {
        const int cn = min(OW_WCHUNK, L - base);
        for (int r = 0; r < 8; ++r) {                 // stage 8 engine rows x 64 samples of the voice sum (slot pass + steal pass)
            const int er = eb + r;
            double x = 0.0;
            if (er < e0 + ne && lane < cn && !eout[er].sum_nonfinite) {
                if (args[er].main_mask) x = sum[((size_t)0 * I + er) * Lcap + base + lane];
                if (args[er].steal_mask) x += sum[((size_t)1 * I + er) * Lcap + base + lane];
            }
            tile[r * (OW_WCHUNK + 1) + lane] = x;
        }
        __syncthreads();
        for (int n = 0; n < cn; ++n) {
            const double x = tile[el * (OW_WCHUNK + 1) + n];
            const double rcur[2] = {rnx[0], rnx[1]};
            {
                const uint32_t nx = (uint32_t)(min(base + n + 1, L - 1) * osr);
                rnx[0] = trem_col_at(rc, nx);
                if (osr == 2) rnx[1] = trem_col_at(rc, nx + 1u);
            }
            const double depth = clampd(sd.next(), 0.0, 1.0);   // engine.rs:533-534, tremolo.rs:117-119
            double in[2];
            if (osr == 2) {
                const double a = allpass3(OW_OS_A0, OW_OS_A1, OW_OS_A2, ua, x);
                const double b = allpass3(OW_OS_B0, OW_OS_B1, OW_OS_B2, ub, x);
                in[0] = role ? 0.0 : a;
                in[1] = role ? 0.0 : b;
            } else {
                in[0] = role ? 0.0 : x;
                in[1] = 0.0;
            }
            for (int j = 0; j < osr; ++j) {
                const size_t idx = (size_t)((base + n) * osr + j);
                const double r_new = fmax(trem_shunt(depth, rcur[j]), 1000.0);   // tremolo.rs:152-167; set_ldr_resistance
                if (fabs(r_new - r_ldr) > 0.01) { r_ldr = r_new; g_ldr = ow_div(1.0, r_new); }
                const double o = dk_step_wide(st, R, q, in[j], g_ldr, g_prev, K);
                g_prev = g_ldr;
                const double other = xor32(o);
                double result = role ? (other - o) : (o - other);                 // main - pump
                if (!isfinite(result)) {
                    dk_dc_reset(K, r_ldr, st);
                    g_ldr = 1.0 / r_ldr; g_prev = g_ldr;
                    result = 0.0;
                    nan_resets += 1u;
                }
                if (valid && role == 0 && q == 0) pre[idx * I + e] = result;
            }
        }
        __syncthreads();
    }
    if (valid && q == 0) {
        dk_store(st, cs, I, e, role ? CS_P_SHADOW : CS_P_MAIN);
        if (role == 0) {
            for (int i = 0; i < 3; ++i) { CSF(CS_OS_UA + i) = ua[i]; CSF(CS_OS_UB + i) = ub[i]; }
            CSF(CS_P_RLDR) = r_ldr; CSF(CS_P_GLDR) = g_ldr; CSF(CS_P_GPREV) = g_prev;
            smoother_store(sd, cs, I, e, CS_SM_DEPTH);
            const uint64_t fl = dbits(CSF(CS_FLAGS));
            if (fl & 1ull) CSF(CS_FLAGS) = bitsd(fl & ~1ull);
            if (nan_resets) {
                const uint64_t d = dbits(CSF(CS_DIAG));
                CSF(CS_DIAG) = bitsd((d & 0xFFFFFFFFull) | ((uint64_t)((uint32_t)(d >> 32) + nan_resets) << 32));
            }
        }
    }
}


// ------------------------------------------------------------------ preamp + output stage in ONE launch, for small pools
// k_preamp_wide and k_post are two serial recurrences in a row: in a pool that leaves the chip empty the block time is the SUM of their
// latencies (paced single instance, 64 samples: 213 + 64 us) plus a launch gap.  Here they are two wavefronts of one workgroup, eight
// engines each: wavefront 0 runs the quad-lane preamp of k_preamp_wide on chunk c of the block while wavefront 1 runs k_post's power amp /
// half-band / speaker / gain on chunk c - 1, handed over through a two-slot LDS ring (OW_FCHUNK host samples per slot), one
// __syncthreads per chunk.  The output stage disappears behind the preamp: the block lasts as long as the preamp alone.  Every value is
// produced by the same statements as in the two kernels (their parity tests run on this kernel whenever the pool is small:
// tests/test_gpu_parity.py::test_chain_fused_is_bit_identical compares it with the two-launch path bit for bit).
// North star: "one persistent kernel per audio buffer ... LDS hand-off" -- the voice kernels stay a launch of their own (a wavefront of
// voices per engine, packed lists), so a pool-of-one block is two launches: voices, chain.
#define OW_FCHUNK 16
#define OW_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
template <bool SPLIT>
__global__ __launch_bounds__(128) void k_chain_fused(const OwConsts* __restrict__ K, double* __restrict__ cs, const OwEngineArgs* __restrict__ args,
                                                     OwEngineOut* __restrict__ eout, const double* __restrict__ sum, const OwTremSrc tsrc,
                                                     double* __restrict__ pre, float* __restrict__ out, int I, int L, int Lcap, int Lout, int e0, int ne) {
    constexpr int OSR = SPLIT ? 2 : 1;
    __shared__ double tin[8 * (OW_FCHUNK + 1)];                 // voice sums of the chunk (wavefront 0 only)
    __shared__ double ring[2][OW_FCHUNK * OSR][8];              // preamp out, chain rate: [slot][sample][engine of the block]
    __shared__ float tout[8 * (OW_FCHUNK + 1)];                 // finished samples of the chunk (wavefront 1 only)
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int eb = e0 + blockIdx.x * 8;
    const int n_chunks = (L + OW_FCHUNK - 1) / OW_FCHUNK;

    // ---- wavefront 0: k_preamp_wide's lanes and state
    const int q = lane & 3, el = (lane & 31) >> 2, role = lane >> 5;
    // ---- wavefront 1: k_post<SPLIT>'s lanes (lane = (engine, oversample phase)); only eight engines per block here
    const int pel = SPLIT ? (lane & 31) : lane, phase = SPLIT ? (lane >> 5) : 0;
    const int e_raw = eb + (wv == 0 ? el : pel);
    const bool in_block = (wv == 0) || pel < 8;
    const bool valid = in_block && e_raw < e0 + ne;
    const int e = valid ? e_raw : (e0 + ne - 1);

    DkWideRows R;
    DkSt st;
    double ua[3] = {0, 0, 0}, ub[3] = {0, 0, 0};
    double r_ldr = 1000000.0, g_ldr = 1e-6, g_prev = 1e-6;
    Smoother sd;
    uint32_t nan_resets = 0;
    TremCol rc;
    rc.p = nullptr; rc.stride8 = 0u;
    double rnx[2] = {1000000.0, 1000000.0};                     // R one host sample ahead (registers): the global load is off the serial path
    double da[3] = {0, 0, 0}, db[3] = {0, 0, 0}, dd = 0.0;
    SpeakerSt sp;
    Smoother ss, sv;
    bool nan_fired = false;
    if (wv == 0) {
        dk_wide_rows_load(R, K, q);
        smoother_load(sd, cs, I, e, CS_SM_DEPTH);
        if (args[e].set_flags & 1u) sd.retarget(args[e].depth_target, K->ramp_samples);
        dk_load(st, cs, I, e, role ? CS_P_SHADOW : CS_P_MAIN);
        for (int i = 0; i < 3; ++i) { ua[i] = CSF(CS_OS_UA + i); ub[i] = CSF(CS_OS_UB + i); }
        r_ldr = CSF(CS_P_RLDR); g_ldr = CSF(CS_P_GLDR); g_prev = CSF(CS_P_GPREV);
        const uint64_t fl = dbits(CSF(CS_FLAGS));
        if (fl & 1ull) {  // deferred preamp.reset() + oversampler.reset() from the output NaN guard (engine.rs:450-457)
            dk_dc_reset(K, r_ldr, st);
            g_ldr = 1.0 / r_ldr; g_prev = g_ldr;
            for (int i = 0; i < 3; ++i) { ua[i] = 0.0; ub[i] = 0.0; }
        }
        rc = trem_col(tsrc, I, e);
        rnx[0] = trem_col_at(rc, 0u);
        if (OSR == 2) rnx[1] = trem_col_at(rc, 1u);
    } else {
        for (int i = 0; i < 3; ++i) { da[i] = CSF(CS_OS_DA + i); db[i] = CSF(CS_OS_DB + i); }
        dd = CSF(CS_OS_DD);
        double* hp = &sp.hpf.b0; double* lp = &sp.lpf.b0;
        for (int i = 0; i < 7; ++i) { hp[i] = CSF(CS_SPK_HPF + i); lp[i] = CSF(CS_SPK_LPF + i); }
        sp.character = CSF(CS_SPK_CHAR); sp.a2 = CSF(CS_SPK_A2); sp.a3 = CSF(CS_SPK_A3); sp.tc = CSF(CS_SPK_TC); sp.ts = CSF(CS_SPK_TS);
        smoother_load(ss, cs, I, e, CS_SM_SPK);
        smoother_load(sv, cs, I, e, CS_SM_VOL);
        if (args[e].set_flags & 2u) ss.retarget(args[e].spk_target, K->ramp_samples);
        if (args[e].set_flags & 4u) sv.retarget(args[e].vol_target, K->ramp_samples);
    }
    const double sr = K->sr, thermal_alpha = K->spk_thermal_alpha;
    double nxt[8 * OW_FCHUNK / 64];
    auto fetch_sums = [&](int chunk) {                           // wavefront 0: voice sums of `chunk` into registers
        const int b0 = chunk * OW_FCHUNK;
#pragma unroll
        for (int u = 0; u < 8 * OW_FCHUNK / 64; ++u) {
            const int k = lane + 64 * u;
            const int r = k / OW_FCHUNK, n = k % OW_FCHUNK;
            const int er = eb + r;
            double x = 0.0;
            if (er < e0 + ne && b0 + n < L && !eout[er].sum_nonfinite) {
                if (args[er].main_mask) x = sum[((size_t)0 * I + er) * Lcap + b0 + n];
                if (args[er].steal_mask) x += sum[((size_t)1 * I + er) * Lcap + b0 + n];
            }
            nxt[u] = x;
        }
    };

    for (int c = 0; c <= n_chunks; ++c) {
        if (wv == 0 && c < n_chunks) {
            const int base = c * OW_FCHUNK;
            const int cn = min(OW_FCHUNK, L - base);
            {   // 8 engine rows x OW_FCHUNK samples of the voice sum (slot pass + steal pass), lanes = (row, sample): the values were
                // fetched while the previous chunk was being solved (registers), so the serial loop never waits for HBM
                if (c == 0) fetch_sums(0);
#pragma unroll
                for (int u = 0; u < 8 * OW_FCHUNK / 64; ++u) {
                    const int k = lane + 64 * u;
                    tin[(k / OW_FCHUNK) * (OW_FCHUNK + 1) + (k % OW_FCHUNK)] = nxt[u];
                }
                OW_WAVE_SYNC();
                if (c + 1 < n_chunks) fetch_sums(c + 1);
            }
            double (*slot)[8] = ring[c & 1];
            for (int n = 0; n < cn; ++n) {
                const double x = tin[el * (OW_FCHUNK + 1) + n];
                const double rcur[2] = {rnx[0], rnx[1]};            // R of this sample, fetched one host sample ago
                {
                    const uint32_t nx = (uint32_t)(min(base + n + 1, L - 1) * OSR);
                    rnx[0] = trem_col_at(rc, nx);
                    if (OSR == 2) rnx[1] = trem_col_at(rc, nx + 1u);
                }
                const double depth = clampd(sd.next(), 0.0, 1.0);   // engine.rs:533-534, tremolo.rs:117-119
                double in[2];
                if (OSR == 2) {
                    const double a = allpass3(OW_OS_A0, OW_OS_A1, OW_OS_A2, ua, x);
                    const double b = allpass3(OW_OS_B0, OW_OS_B1, OW_OS_B2, ub, x);
                    in[0] = role ? 0.0 : a;
                    in[1] = role ? 0.0 : b;
                } else {
                    in[0] = role ? 0.0 : x;
                    in[1] = 0.0;
                }
                for (int j = 0; j < OSR; ++j) {
                    const size_t idx = (size_t)((base + n) * OSR + j);
                    const double r_new = fmax(trem_shunt(depth, rcur[j]), 1000.0);   // tremolo.rs:152-167; set_ldr_resistance
                    if (fabs(r_new - r_ldr) > 0.01) { r_ldr = r_new; g_ldr = ow_div(1.0, r_new); }
                    const double o = dk_step_wide(st, R, q, in[j], g_ldr, g_prev, K);
                    g_prev = g_ldr;
                    const double other = xor32(o);
                    double result = role ? (other - o) : (o - other);                 // main - pump
                    if (!isfinite(result)) {
                        dk_dc_reset(K, r_ldr, st);
                        g_ldr = 1.0 / r_ldr; g_prev = g_ldr;
                        result = 0.0;
                        nan_resets += 1u;
                    }
                    if (role == 0 && q == 0) {
                        slot[n * OSR + j][el] = result;
                        if (valid) pre[idx * I + e] = result;                        // the preamp tap (ow_pool_read_preamp_out)
                    }
                }
            }
        }
        if (wv == 1 && c >= 1) {
            const int base = (c - 1) * OW_FCHUNK;
            const int cn = min(OW_FCHUNK, L - base);
            const double (*slot)[8] = ring[(c - 1) & 1];
            const int pe = pel < 8 ? pel : 7;
            for (int n = 0; n < cn; ++n) {
                const double pc = slot[n * OSR + phase][pe];
                const double y = power_amp(pc * 0.25);
                double o;
                if (SPLIT) {  // engine.rs:536-553
                    const double yo = __shfl_xor(y, 32);
                    const double y0 = phase ? yo : y, y1 = phase ? y : yo;
                    const double a = allpass3(OW_OS_A0, OW_OS_A1, OW_OS_A2, da, y0);
                    const double b = allpass3(OW_OS_B0, OW_OS_B1, OW_OS_B2, db, y1);
                    o = (a + dd) * 0.5;
                    dd = b;
                } else {
                    o = y;
                }
                speaker_set_character(sp, ss.next(), sr);                           // engine.rs:437-438
                const double shaped = speaker_process(sp, o, thermal_alpha);
                const double post = shaped * 7.498942093324558 * sv.next();         // POST_SPEAKER_GAIN x user volume
                float f = (float)post;
                if (!isfinite(f)) {                                                 // engine.rs:450-458
                    f = 0.0f;
                    sp.hpf.s1 = sp.hpf.s2 = sp.lpf.s1 = sp.lpf.s2 = 0.0;
                    sp.ts = 0.0;
                    nan_fired = true;
                }
                if (phase == 0 && pel < 8) tout[pel * (OW_FCHUNK + 1) + n] = f;
            }
            OW_WAVE_SYNC();
            for (int k = lane; k < 8 * OW_FCHUNK; k += 64) {
                const int r = k / OW_FCHUNK, n = k - r * OW_FCHUNK;
                const int er = eb + r;
                if (er < e0 + ne && n < cn) out[(size_t)er * Lout + base + n] = tout[r * (OW_FCHUNK + 1) + n];
            }
            OW_WAVE_SYNC();
        }
        __syncthreads();
    }
    // ---- state back: the preamp's half first (it clears the deferred-reset flag), then the output stage's (it may set it)
    if (wv == 0 && valid && q == 0) {
        dk_store(st, cs, I, e, role ? CS_P_SHADOW : CS_P_MAIN);
        if (role == 0) {
            for (int i = 0; i < 3; ++i) { CSF(CS_OS_UA + i) = ua[i]; CSF(CS_OS_UB + i) = ub[i]; }
            CSF(CS_P_RLDR) = r_ldr; CSF(CS_P_GLDR) = g_ldr; CSF(CS_P_GPREV) = g_prev;
            smoother_store(sd, cs, I, e, CS_SM_DEPTH);
            const uint64_t fl = dbits(CSF(CS_FLAGS));
            if (fl & 1ull) CSF(CS_FLAGS) = bitsd(fl & ~1ull);
            if (nan_resets) {
                const uint64_t d = dbits(CSF(CS_DIAG));
                CSF(CS_DIAG) = bitsd((d & 0xFFFFFFFFull) | ((uint64_t)((uint32_t)(d >> 32) + nan_resets) << 32));
            }
        }
    }
    __threadfence_block();
    __syncthreads();
    if (wv != 1 || !valid || phase != 0) return;
    if (nan_fired) {  // preamp.reset()/oversampler.reset() act on post-block state: defer the preamp/up half to the next block's preamp
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
}


// k_job_chain_wide as two wavefronts (round 4): a job's run time is the chain's serial latency, and half-band decimation, volume taper,
// behavioural power amp and speaker (main.rs:445-496) hang behind the preamp in the same lane.  Here wavefront 0 runs the up-sampler and
// the two preamp steps of a sample (k_job_chain_wide's lanes and state) and hands the pair of chain-rate outputs over through a
// two-slot LDS ring; wavefront 1 -- one lane per job -- follows a chunk behind with the rest.  Same statements per job in the same
// order: bit-identical to k_job_chain_wide (tests/test_gpu_render_flags.py, OW_JOB_FUSED=0/1).
// (`reed` and `voice_prog` are NOT __restrict__ / read-only to the compiler: in the overlapped form k_job_voice writes both while this
// kernel runs, and the acquire load of the progress word is what orders the chunk's loads behind the producer's stores.)
__global__ __launch_bounds__(128) void k_job_chain_fused(const OwConsts* __restrict__ K, const OwJobDev* __restrict__ jobs, const volatile double* reed,
                                                         double* __restrict__ out, int n_jobs, long long n, long long stride, const int* voice_prog,
                                                         int* voice_err) {
    __shared__ double tin[8 * (OW_FCHUNK + 1)];
    __shared__ double ring[2][OW_FCHUNK * 2][8];               // preamp out at the chain rate: [slot][sample x phase][job of the block]
    __shared__ double tout[8 * (OW_FCHUNK + 1)];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int q = lane & 3, role = lane >> 5;
    const int jl = wv == 0 ? (lane & 31) >> 2 : (lane & 7);      // wavefront 1: lanes 0..7 own a job each, the others shadow them
    const int jb = blockIdx.x * 8;
    const int j = jb + jl;
    const bool valid = j < n_jobs;
    const OwJobDev jd = jobs[valid ? j : n_jobs - 1];
    const int osr = K->oversample ? 2 : 1;
    const double sr = K->sr;
    const long long n_chunks = (n + OW_FCHUNK - 1) / OW_FCHUNK;
    bool gave_up = false;

    DkWideRows R;
    DkSt st;
    double r_ldr = 1000000.0, g_ldr = 1.0 / 1000000.0, g_prev = g_ldr;
    double ua[3] = {0, 0, 0}, ub[3] = {0, 0, 0}, da[3] = {0, 0, 0}, db[3] = {0, 0, 0}, dd = 0.0;
    SpeakerSt sp;
    if (wv == 0) {
        dk_wide_rows_load(R, K, q);
        dk_dc_reset(K, r_ldr, st);                               // DkPreamp::new(preamp_sr); reset(); set_ldr_resistance(r_ldr)  (main.rs:432-441)
        const double r_new = fmax(jd.r_ldr, 1000.0);
        if (fabs(r_new - r_ldr) > 0.01) { r_ldr = r_new; g_ldr = 1.0 / r_new; }
    } else {
        sp.character = 1.0; sp.ts = 0.0;                         // Speaker::new(sr); set_character(c)  (main.rs:483-484)
        sp.hpf.s1 = sp.hpf.s2 = sp.lpf.s1 = sp.lpf.s2 = 0.0;
        speaker_update(sp, sr);
        speaker_set_character(sp, jd.speaker, sr);
    }
    const double vol2_a = jd.volume;
    auto preamp_step = [&](double x) -> double {
        const double o = dk_step_wide(st, R, q, x, g_ldr, g_prev, K);
        g_prev = g_ldr;
        const double other = xor32(o);
        double res = role ? (other - o) : (o - other);
        if (!isfinite(res)) {
            dk_dc_reset(K, r_ldr, st); g_ldr = 1.0 / r_ldr; g_prev = g_ldr;
            res = 0.0;
        }
        return res;
    };
    for (long long c = 0; c <= n_chunks; ++c) {
        if (wv == 0 && c < n_chunks) {
            const long long base = c * OW_FCHUNK;
            const int cn = (int)((n - base) < OW_FCHUNK ? (n - base) : OW_FCHUNK);
            if (voice_prog && !gave_up) {   // the voices of these eight jobs (one block of k_job_voice) are being rendered beside this kernel: wait for the chunk
                const int need = (int)(base + cn);
                // bounded: a producer that never gets scheduled (a chip filled by other work) must not hang the device -- after ~5 s the
                // kernel raises the flag behind the progress words and runs on without waiting; the host then repeats the chain alone
                long spins = 0;
                while (__hip_atomic_load(&voice_prog[jb >> 6], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < need) {
                    __builtin_amdgcn_s_sleep(32);
                    if (++spins > 5000000L) { gave_up = true; if (lane == 0) atomicOr(voice_err, 1); break; }
                }
            }
#pragma unroll
            for (int u = 0; u < 8 * OW_FCHUNK / 64; ++u) {
                const int k = lane + 64 * u;
                const int r = k / OW_FCHUNK, sm = k % OW_FCHUNK;
                double x = 0.0;
                if (jb + r < n_jobs && sm < cn) x = reed[(size_t)(jb + r) * stride + base + sm];
                tin[r * (OW_FCHUNK + 1) + sm] = x;
            }
            OW_WAVE_SYNC();
            double (*slot)[8] = ring[c & 1];
            for (int sidx = 0; sidx < cn; ++sidx) {
                const double x = tin[jl * (OW_FCHUNK + 1) + sidx];
                if (osr == 2) {                                  // main.rs:445-466: per-sample up(1) -> 2x process (-> down(1) on wavefront 1)
                    const double a = allpass3(OW_OS_A0, OW_OS_A1, OW_OS_A2, ua, x);
                    const double b = allpass3(OW_OS_B0, OW_OS_B1, OW_OS_B2, ub, x);
                    const double in[2] = {role ? 0.0 : a, role ? 0.0 : b};
                    for (int k = 0; k < 2; ++k) {
                        const double pk = preamp_step(in[k]);
                        if (role == 0 && q == 0) slot[sidx * 2 + k][jl] = pk;
                    }
                } else {
                    const double pk = preamp_step(role ? 0.0 : x);
                    if (role == 0 && q == 0) slot[sidx * 2][jl] = pk;
                }
            }
        }
        if (wv == 1 && c >= 1) {
            const long long base = (c - 1) * OW_FCHUNK;
            const int cn = (int)((n - base) < OW_FCHUNK ? (n - base) : OW_FCHUNK);
            const double (*slot)[8] = ring[(c - 1) & 1];
            for (int sidx = 0; sidx < cn; ++sidx) {
                double pre;
                if (osr == 2) {
                    const double fa = allpass3(OW_OS_A0, OW_OS_A1, OW_OS_A2, da, slot[sidx * 2][jl]);
                    const double fb = allpass3(OW_OS_B0, OW_OS_B1, OW_OS_B2, db, slot[sidx * 2 + 1][jl]);
                    pre = (fa + dd) * 0.5;
                    dd = fb;
                } else {
                    pre = slot[sidx * 2][jl];
                }
                // main.rs:487-496: volume^2 (audio taper) -> optional power amp at base rate -> speaker -> PSG
                const double att = pre * vol2_a * vol2_a;
                const double amp = jd.poweramp ? power_amp(att) : att;
                const double y = speaker_process(sp, amp, K->spk_thermal_alpha) * 7.498942093324558;
                if (lane < 8) tout[jl * (OW_FCHUNK + 1) + sidx] = y;
            }
            OW_WAVE_SYNC();
            for (int k = lane; k < 8 * OW_FCHUNK; k += 64) {
                const int r = k / OW_FCHUNK, sm = k - r * OW_FCHUNK;
                if (jb + r < n_jobs && sm < cn) out[(size_t)(jb + r) * stride + base + sm] = tout[r * (OW_FCHUNK + 1) + sm];
            }
            OW_WAVE_SYNC();
        }
        __syncthreads();
    }
}

}  // namespace owdev
