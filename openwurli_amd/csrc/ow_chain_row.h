// openwurli-hip: the legacy DK preamp with ONE SOLVER STATE PER ROW OF SIXTEEN LANES, for pools that leave the chip empty.
//
// A small pool's block time is the serial time of one chain sample on one wavefront, and a lone wavefront issues one f64 instruction
// per 5.8 cycles whatever its lanes do (profiles/r04_issue_cost.txt): what counts is the number of instructions per chain sample.  The
// quad-lane step (ow_chain_wide.h, four lanes per state) executes ~600 vector + ~150 scalar instructions per chain sample inside
// k_chain_fused -- a third of them accumulator-register copies: that kernel is two roles in one loop and needs both roles' registers at
// once.  This step is dk_step (dk_preamp_legacy.rs:447-554) with
//   * lane r (r = lane & 7) = row r of the three matrix products: rhs = A_neg v in the reference's own dense form (it multiplies the
//     structural zeros too), the per-sample source terms as per-row coefficients (x + 0*y is x; 1*y is y; x + (-1)*y is x - y),
//     v_pred = S rhs, v = v_pred + S N_i i_c - ...: 8 + 8 multiply-adds for ALL rows where the quad-lane step spends 2 x 16 + 40, and ONE
//     v_mov_b64_dpp row_newbcast per value that crosses lanes (two 32-bit quad moves before);
//   * the two junction exponentials of a Newton sweep on alternate lanes, the 2x2 update replicated (as in dk_step_wide), the sweeps of the
//     wavefront's four states in one wave-uniform loop;
//   * four states per wavefront: main preamps of two engines in rows 0-1, their shadow preamps in rows 2-3 (main - shadow is one
//     v_permlane32_swap pair), so a workgroup is four preamp wavefronts (eight engines) + the output-stage wavefront of
//     k_chain_fused, each role in a loop of its own (its registers are its own), one barrier per 16-sample chunk.
// Every number is produced by the same operations on the same operands as in dk_step: bit-identical to k_chain_fused and to the
// two-launch path (tests/test_gpu_parity.py::test_chain_row_is_bit_identical).
#pragma once
#include "ow_chain_wide.h"
#include "ow_trem_row.h"

namespace owdev {

struct DkRowK {      // r = lane & 7
    double an[8], s[8], fb, c1, c2, two_w;      // row r of A_neg, S, S's feedback column, the two S N_i column differences, 2w
    double m7, e0, ci0, ci1;                    // rows that take -g_prev v7 | the input source | +-i_nl[0] | +-i_nl[1]
    double g_cin, s_fb_fb, k[4], nv_sfb[2], sfb_ni[2], gc_1pc, c_cin;
    double oc0, oc1, oc2;                       // half-band branch A (even lanes) / B (odd lanes)
};
OW_DEV void dk_row_consts(DkRowK& R, const OwConsts* __restrict__ K, int lane) {
    const int r = lane & 7;
#pragma unroll
    for (int j = 0; j < 8; ++j) { R.an[j] = K->p_a_neg[r][j]; R.s[j] = K->p_s[r][j]; }
    R.fb = K->p_s_fb_col[r]; R.c1 = K->p_sni_d1[r]; R.c2 = K->p_sni_d2[r]; R.two_w = K->p_two_w[r];
    R.m7 = r == 7 ? 1.0 : 0.0;
    R.e0 = r == 0 ? 1.0 : 0.0;
    R.ci0 = r == 1 ? 1.0 : (r == 2 ? -1.0 : 0.0);
    R.ci1 = r == 3 ? 1.0 : (r == 5 ? -1.0 : 0.0);
    R.g_cin = K->p_g_cin; R.s_fb_fb = K->p_s_fb_fb; R.gc_1pc = K->p_gc_1pc; R.c_cin = K->p_c_cin;
    R.k[0] = K->p_k[0][0]; R.k[1] = K->p_k[0][1]; R.k[2] = K->p_k[1][0]; R.k[3] = K->p_k[1][1];
    R.nv_sfb[0] = K->p_nv_sfb[0]; R.nv_sfb[1] = K->p_nv_sfb[1]; R.sfb_ni[0] = K->p_sfb_ni[0]; R.sfb_ni[1] = K->p_sfb_ni[1];
    const bool odd = (lane & 1) != 0;
    R.oc0 = odd ? OW_OS_B0 : OW_OS_A0; R.oc1 = odd ? OW_OS_B1 : OW_OS_A1; R.oc2 = odd ? OW_OS_B2 : OW_OS_A2;
    // the wave-uniform ones in vector registers too (512 of them for one wavefront per SIMD): the scalar registers are left to the junction
    // exponential's polynomial, whose Horner steps then read their coefficients as the one scalar operand of a v_fma_f64 -- kept in
    // vector registers each step costs a v_mov_b64 in front of a v_fmac_f64
    vgpr_pin(R.g_cin); vgpr_pin(R.s_fb_fb); vgpr_pin(R.gc_1pc); vgpr_pin(R.c_cin);
#pragma unroll
    for (int i = 0; i < 4; ++i) vgpr_pin(R.k[i]);
    vgpr_pin(R.nv_sfb[0]); vgpr_pin(R.nv_sfb[1]); vgpr_pin(R.sfb_ni[0]); vgpr_pin(R.sfb_ni[1]);
}

// exp_bounded (ow_chain_dev.h) with every Horner step as a three-address v_fma_f64: the compiler, with the coefficients in vector registers
// (a wavefront of this kernel has no scalar registers to spare), emits v_mov_b64 + v_fmac_f64 per step -- 9 moves at 8.4 cycles each per
// evaluation.  Same operations, same constants.
OW_DEV double fma_vvv(double a, double b, double c) {
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
OW_DEV double exp_bounded_row(double x) {
#ifdef OW_LIB_EXP
    return exp(x);
#else
    const double n = rint(x * __longlong_as_double(0x3ff71547652b82feLL));
    double r = __builtin_fma(__longlong_as_double((long long)0xbfe62e42fefa39efULL), n, x);
    r = __builtin_fma(__longlong_as_double((long long)0xbc7abc9e3b39803fULL), n, r);
    double p = fma_vvv(__longlong_as_double(0x3e5ade156a5dcb37LL), r, __longlong_as_double(0x3e928af3fca7ab0cLL));
    p = fma_vvv(r, p, __longlong_as_double(0x3ec71dee623fde64LL));
    p = fma_vvv(r, p, __longlong_as_double(0x3efa01997c89e6b0LL));
    p = fma_vvv(r, p, __longlong_as_double(0x3f2a01a014761f6eLL));
    p = fma_vvv(r, p, __longlong_as_double(0x3f56c16c1852b7b0LL));
    p = fma_vvv(r, p, __longlong_as_double(0x3f81111111122322LL));
    p = fma_vvv(r, p, __longlong_as_double(0x3fa55555555502a1LL));
    p = fma_vvv(r, p, __longlong_as_double(0x3fc5555555555511LL));
    p = fma_vvv(r, p, __longlong_as_double(0x3fe000000000000bLL));
    p = __builtin_fma(r, p, 1.0);
    p = __builtin_fma(r, p, 1.0);
    return ldexp(p, (int)n);
#endif
}
OW_DEV void dk_ic_gm_row(double vbe, double& ic, double& gm) {  // dk_ic_gm
    const double e = exp_bounded_row(OW_DIV_C(dk_clamp_vbe(vbe), OW_P_VT));
    ic = OW_P_IS * (e - 1.0);
    gm = (OW_P_IS / OW_P_VT) * e;
}

// dk_step for the state of this row.  st.v[] etc. are uniform over the row; returns v[OUT].
__device__ __forceinline__ double dk_step_row(DkSt& st, const DkRowK& R, int lane, double input, double g_ldr, double g_ldr_prev) {
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += R.an[j] * st.v[j];
    acc -= (R.m7 * g_ldr_prev) * st.v[7];
    const double cin_now = R.g_cin * input + st.j_cin;
    acc += R.e0 * (cin_now + st.cin_prev);
    acc += R.ci0 * st.i_nl[0];
    acc += R.ci1 * st.i_nl[1];
    acc += R.two_w;
    double rhs[8];
    static_for<0, 8>([&](auto I) { rhs[I] = rowb<I>(acc); });
    double lo = 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) lo += R.s[j] * rhs[j];                 // v_pred_base[r]
    const double vpb7 = rowb<7>(lo);
    const double sm_k = ow_div(g_ldr, 1.0 + R.s_fb_fb * g_ldr);
    const double sm_vpred = sm_k * vpb7;
    const double vp = lo - sm_vpred * R.fb;                            // v_pred[r]
    const double p0 = rowb<0>(vp) - rowb<1>(vp), p1 = rowb<2>(vp) - rowb<3>(vp);
    const double k00 = R.k[0] - sm_k * R.nv_sfb[0] * R.sfb_ni[0];
    const double k01 = R.k[1] - sm_k * R.nv_sfb[0] * R.sfb_ni[1];
    const double k10 = R.k[2] - sm_k * R.nv_sfb[1] * R.sfb_ni[0];
    const double k11 = R.k[3] - sm_k * R.nv_sfb[1] * R.sfb_ni[1];
    double vn0 = st.v_nl[0], vn1 = st.v_nl[1];
    // dk_step's Newton loop (ow_chain_dev.h): (ic, gm) hold the evaluation at (vn0, vn1) -- the state's own on entry, a fresh one after
    // every update (even lanes evaluate junction 0, odd lanes junction 1) -- and the four states of the wavefront sweep in ONE wave-uniform
    // loop: a state that has left the reference's loop at one of its two `break`s (`done`) stays where it is until every state of the
    // wavefront has; the row moves always run with every lane active and the loop costs two scalar branches per sweep.
    double ic0 = st.i_nl[0], ic1 = st.i_nl[1], gm0 = st.gm[0], gm1 = st.gm[1];
    bool done = false;
    const bool odd = (lane & 1) != 0;
    for (int iter = 0; iter < 6; ++iter) {
        const double f0 = vn0 - p0 - k00 * ic0 - k01 * ic1;
        const double f1 = vn1 - p1 - k10 * ic0 - k11 * ic1;
        const double j00 = 1.0 - k00 * gm0, j01 = -k01 * gm1, j10 = -k10 * gm0, j11 = 1.0 - k11 * gm1;
        const double det = j00 * j11 - j01 * j10;
        done = done || (fabs(f0) < 1e-9 && fabs(f1) < 1e-9) || fabs(det) < 1e-30;
        if (__builtin_amdgcn_ballot_w64(!done) == 0ull) break;
        const double inv_det = ow_div(1.0, det);
        const double n0 = vn0 - inv_det * (j11 * f0 - j01 * f1);
        const double n1 = vn1 - inv_det * (j00 * f1 - j10 * f0);
        vn0 = done ? vn0 : n0;
        vn1 = done ? vn1 : n1;
        double ic, gm;
        dk_ic_gm_row(odd ? vn1 : vn0, ic, gm);
        ic0 = rowb<0>(ic); ic1 = rowb<1>(ic);
        gm0 = rowb<0>(gm); gm1 = rowb<1>(gm);
    }
    const double dot = R.sfb_ni[0] * ic0 + R.sfb_ni[1] * ic1;
    const double v_me = vp + (ic0 * R.c1 + ic1 * R.c2) - sm_k * dot * R.fb;
    static_for<0, 8>([&](auto I) { st.v[I] = rowb<I>(v_me); });
    st.cin_prev = cin_now;
    const double dv_cin = input - st.v[0];
    st.j_cin = -R.gc_1pc * dv_cin - R.c_cin * st.j_cin;
    st.i_nl[0] = ic0; st.i_nl[1] = ic1;
    st.v_nl[0] = vn0; st.v_nl[1] = vn1;
    st.gm[0] = gm0; st.gm[1] = gm1;
    return st.v[6];
}

// k_chain_fused (ow_chain_wide.h) with the row step: eight engines per workgroup, wavefronts 0-3 = the preamps of two engines each
// (rows 0-1 main, rows 2-3 shadow), wavefront 4 = k_post<SPLIT>'s lanes for the eight.  Same arguments, same results.
// CH: host samples per hand-over chunk.  The output stage follows the preamps one chunk behind, so a block ends with one chunk of
// output-stage time (1.2 us per sample) that nothing overlaps: 16 for long blocks (a barrier per 16 samples), 8 for blocks of <= 128
// samples, where those 10 us are 4 % of a lone instance's 64-sample buffer.
template <bool SPLIT, int CH>
__global__ __launch_bounds__(320) void k_chain_row(const OwConsts* __restrict__ K, double* __restrict__ cs, const OwEngineArgs* __restrict__ args,
                                                   OwEngineOut* __restrict__ eout, const double* __restrict__ sum, const OwTremSrc tsrc,
                                                   double* __restrict__ pre, float* __restrict__ out, int I, int L, int Lcap, int Lout, int e0, int ne) {
    constexpr int OSR = SPLIT ? 2 : 1;
    __shared__ double tin[8 * (CH + 1)];                 // voice sums of the chunk: [engine of the block][sample]
    __shared__ double ring[2][CH * OSR][8];              // preamp out, chain rate: [slot][sample][engine of the block]
    __shared__ float tout[8 * (CH + 1)];                 // finished samples of the chunk (output wavefront only)
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int eb = e0 + blockIdx.x * 8;
    const int n_chunks = (L + CH - 1) / CH;

    if (wv < 4) {
        // ---- a preamp wavefront: rows (0, 1) = main of engines el0, el0 + 1; rows (2, 3) = their shadows
        const int row = lane >> 4, role = row >> 1;
        const int el = 2 * wv + (row & 1);
        const int e_raw = eb + el;
        const bool valid = e_raw < e0 + ne;
        const int e = valid ? e_raw : (e0 + ne - 1);
        DkRowK R;
        dk_row_consts(R, K, lane);
        DkSt st;
        Smoother sd;
        smoother_load(sd, cs, I, e, CS_SM_DEPTH);
        if (args[e].set_flags & 1u) sd.retarget(args[e].depth_target, K->ramp_samples);
        dk_load(st, cs, I, e, role ? CS_P_SHADOW : CS_P_MAIN);
        const bool odd = (lane & 1) != 0;
        double us[3];                                            // this lane's half-band branch (A on even lanes, B on odd ones)
        for (int i = 0; i < 3; ++i) us[i] = CSF((odd ? CS_OS_UB : CS_OS_UA) + i);
        double r_ldr = CSF(CS_P_RLDR), g_ldr = CSF(CS_P_GLDR), g_prev = CSF(CS_P_GPREV);
        {
            const uint64_t fl = dbits(CSF(CS_FLAGS));
            if (fl & 1ull) {  // deferred preamp.reset() + oversampler.reset() from the output NaN guard (engine.rs:450-457)
                dk_dc_reset(K, r_ldr, st);
                g_ldr = 1.0 / r_ldr; g_prev = g_ldr;
                for (int i = 0; i < 3; ++i) us[i] = 0.0;
            }
        }
        uint32_t nan_resets = 0;
        double sh_depth = __longlong_as_double(0x7FF8000000000000LL), sh_top = 0.0, sh_lower = 0.0;      // (NaN: nothing formed yet)
        const TremCol rc = trem_col(tsrc, I, e);
        double rnx[2];                                           // R one host sample ahead (registers): the global load is off the serial path
        rnx[0] = trem_col_at(rc, 0u);
        rnx[1] = OSR == 2 ? trem_col_at(rc, 1u) : 0.0;
        // voice sums of the next chunk, fetched while this one is solved: lanes 0-31 = (engine of this wavefront, sample)
        const int f_el = 2 * wv + (lane >= CH ? 1 : 0), f_n = lane >= CH ? lane - CH : lane;      // lanes 0 .. 2 CH - 1 = (engine of this wavefront, sample of the chunk)
        auto fetch_sum = [&](int chunk) -> double {
            const int er = eb + f_el, b0 = chunk * CH;
            double x = 0.0;
            if (lane < 2 * CH && er < e0 + ne && b0 + f_n < L && !eout[er].sum_nonfinite) {
                if (args[er].main_mask) x = sum[((size_t)0 * I + er) * Lcap + b0 + f_n];
                if (args[er].steal_mask) x += sum[((size_t)1 * I + er) * Lcap + b0 + f_n];
            }
            return x;
        };
        double nxt = fetch_sum(0);
        const double sgn = role ? -1.0 : 1.0;
        for (int c = 0; c <= n_chunks; ++c) {
            if (c < n_chunks) {
                const int base = c * CH;
                const int cn = min(CH, L - base);
                if (lane < 2 * CH) tin[f_el * (CH + 1) + f_n] = nxt;
                OW_WAVE_SYNC();
                if (c + 1 < n_chunks) nxt = fetch_sum(c + 1);
                double (*slot)[8] = ring[c & 1];
                for (int n = 0; n < cn; ++n) {
                    const double x = tin[el * (CH + 1) + n];
                    const double rcur[2] = {rnx[0], rnx[1]};            // R of this sample, fetched one host sample ago
                    {
                        const uint32_t nx = (uint32_t)(min(base + n + 1, L - 1) * OSR);
                        rnx[0] = trem_col_at(rc, nx);
                        if (OSR == 2) rnx[1] = trem_col_at(rc, nx + 1u);
                    }
                    const double depth = clampd(sd.next(), 0.0, 1.0);   // engine.rs:533-534, tremolo.rs:117-119
                    // Tremolo::shunt_impedance (tremolo.rs:152-167; trem_shunt): the pot's upper leg and r_lower depend on the depth alone, and
                    // the depth only moves while its smoother ramps -- they are formed again (the same operations: the same bits) when some
                    // state of the wavefront sees another depth than the one they were formed from, i.e. on a few blocks per knob movement
                    if (__builtin_amdgcn_ballot_w64(!(depth == sh_depth)) != 0ull) {
                        sh_depth = depth;
                        const double r_upper = 50000.0 * (1.0 - depth);
                        sh_lower = 50000.0 * depth;
                        sh_top = r_upper > 0.0 ? ow_div(r_upper * 18000.0, r_upper + 18000.0) : 0.0;
                    }
                    double in[2];
                    if (OSR == 2) {
                        const double y = allpass3(R.oc0, R.oc1, R.oc2, us, x);      // branch A on even lanes, B on odd ones
                        in[0] = role ? 0.0 : rowb<0>(y);
                        in[1] = role ? 0.0 : rowb<1>(y);
                    } else {
                        in[0] = role ? 0.0 : x;
                        in[1] = 0.0;
                    }
#pragma unroll
                    for (int j = 0; j < OSR; ++j) {
                        const size_t idx = (size_t)((base + n) * OSR + j);
                        const double branch = 680.0 + rcur[j];
                        const double low = sh_lower > 0.0 ? ow_div(sh_lower * branch, sh_lower + branch) : 0.0;
                        const double r_new = fmax(sh_top + low, 1000.0);                 // set_ldr_resistance
                        if (fabs(r_new - r_ldr) > 0.01) { r_ldr = r_new; g_ldr = ow_div(1.0, r_new); }
                        const double o = dk_step_row(st, R, lane, in[j], g_ldr, g_prev);
                        g_prev = g_ldr;
                        const double other = xor32(o);
                        double result = (o - other) * sgn;                                // main - pump, in both roles (x * -1.0 is -x)
                        if (!isfinite(result)) {
                            dk_dc_reset(K, r_ldr, st);
                            g_ldr = 1.0 / r_ldr; g_prev = g_ldr;
                            result = 0.0;
                            nan_resets += 1u;
                        }
                        if (role == 0 && (lane & 15) == 0) {
                            slot[n * OSR + j][el] = result;
                            if (valid) pre[idx * I + e] = result;                        // the preamp tap (ow_pool_read_preamp_out)
                        }
                    }
                }
            }
            __syncthreads();
        }
        // ---- state back (the deferred-reset flag is cleared here; the output stage may set it again after the barrier below)
        if (valid && (lane & 15) < 2) {
            if ((lane & 15) == 0) dk_store(st, cs, I, e, role ? CS_P_SHADOW : CS_P_MAIN);
            if (role == 0) {
                for (int i = 0; i < 3; ++i) CSF((odd ? CS_OS_UB : CS_OS_UA) + i) = us[i];
                if ((lane & 15) == 0) {
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
        __threadfence_block();
        __syncthreads();
        return;
    }

    // ---- the output-stage wavefront: k_post<SPLIT>'s lanes (lane = (engine, oversample phase)) for the eight engines of the block
    const int pel = SPLIT ? (lane & 31) : lane, phase = SPLIT ? (lane >> 5) : 0;
    const int e_raw = eb + pel;
    const bool valid = pel < 8 && e_raw < e0 + ne;
    const int e = valid ? e_raw : (e0 + ne - 1);
    double da[3], db[3], dd;
    SpeakerSt sp;
    Smoother ss, sv;
    bool nan_fired = false;
    for (int i = 0; i < 3; ++i) { da[i] = CSF(CS_OS_DA + i); db[i] = CSF(CS_OS_DB + i); }
    dd = CSF(CS_OS_DD);
    {
        double* hp = &sp.hpf.b0; double* lp = &sp.lpf.b0;
        for (int i = 0; i < 7; ++i) { hp[i] = CSF(CS_SPK_HPF + i); lp[i] = CSF(CS_SPK_LPF + i); }
        sp.character = CSF(CS_SPK_CHAR); sp.a2 = CSF(CS_SPK_A2); sp.a3 = CSF(CS_SPK_A3); sp.tc = CSF(CS_SPK_TC); sp.ts = CSF(CS_SPK_TS);
    }
    smoother_load(ss, cs, I, e, CS_SM_SPK);
    smoother_load(sv, cs, I, e, CS_SM_VOL);
    if (args[e].set_flags & 2u) ss.retarget(args[e].spk_target, K->ramp_samples);
    if (args[e].set_flags & 4u) sv.retarget(args[e].vol_target, K->ramp_samples);
    const double sr = K->sr, thermal_alpha = K->spk_thermal_alpha;
    for (int c = 0; c <= n_chunks; ++c) {
        if (c >= 1) {
            const int base = (c - 1) * CH;
            const int cn = min(CH, L - base);
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
                if (phase == 0 && pel < 8) tout[pel * (CH + 1) + n] = f;
            }
            OW_WAVE_SYNC();
            for (int k = lane; k < 8 * CH; k += 64) {
                const int r = k / CH, n = k - r * CH;
                const int er = eb + r;
                if (er < e0 + ne && n < cn) out[(size_t)er * Lout + base + n] = tout[r * (CH + 1) + n];
            }
            OW_WAVE_SYNC();
        }
        __syncthreads();
    }
    __threadfence_block();
    __syncthreads();                                             // the preamp wavefronts have cleared the deferred-reset flag
    if (!valid || phase != 0) return;
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

}  // namespace owdev

namespace owdev {

// k_job_chain_fused (ow_chain_wide.h) with the row step: eight jobs per workgroup, wavefronts 0-3 = up-sampler + the two preamp steps of
// a sample for two jobs each (rows 0-1 main, rows 2-3 shadow), wavefront 4 = the rest of main.rs:445-496 one lane per job, a chunk behind.
// Same arguments, same statements per job in the same order: bit-identical (tests/test_gpu_render_flags.py, OW_JOB_ROW=0/1).
__global__ __launch_bounds__(320) void k_job_chain_row(const OwConsts* __restrict__ K, const OwJobDev* __restrict__ jobs, const volatile double* reed,
                                                       double* __restrict__ out, int n_jobs, long long n, long long stride, const int* voice_prog,
                                                       int* voice_err) {
    __shared__ double tin[8 * (OW_FCHUNK + 1)];
    __shared__ double ring[2][OW_FCHUNK * 2][8];               // preamp out at the chain rate: [slot][sample x phase][job of the block]
    __shared__ double tout[8 * (OW_FCHUNK + 1)];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int jb = blockIdx.x * 8;
    const int osr = K->oversample ? 2 : 1;
    const long long n_chunks = (n + OW_FCHUNK - 1) / OW_FCHUNK;

    if (wv < 4) {
        const int row = lane >> 4, role = row >> 1;
        const int jl = 2 * wv + (row & 1);
        const int j = jb + jl;
        const bool valid = j < n_jobs;
        const OwJobDev jd = jobs[valid ? j : n_jobs - 1];
        DkRowK R;
        dk_row_consts(R, K, lane);
        DkSt st;
        double r_ldr = 1000000.0, g_ldr = 1.0 / 1000000.0, g_prev = g_ldr;
        double us[3] = {0, 0, 0};
        dk_dc_reset(K, r_ldr, st);                               // DkPreamp::new(preamp_sr); reset(); set_ldr_resistance(r_ldr)  (main.rs:432-441)
        {
            const double r_new = fmax(jd.r_ldr, 1000.0);
            if (fabs(r_new - r_ldr) > 0.01) { r_ldr = r_new; g_ldr = 1.0 / r_new; }
        }
        const double sgn = role ? -1.0 : 1.0;
        bool gave_up = false;
        auto preamp_step = [&](double x) -> double {
            const double o = dk_step_row(st, R, lane, x, g_ldr, g_prev);
            g_prev = g_ldr;
            const double other = xor32(o);
            double res = (o - other) * sgn;
            if (!isfinite(res)) {
                dk_dc_reset(K, r_ldr, st); g_ldr = 1.0 / r_ldr; g_prev = g_ldr;
                res = 0.0;
            }
            return res;
        };
        const int f_jl = 2 * wv + ((lane >> 4) & 1), f_n = lane & 15;
        for (long long c = 0; c <= n_chunks; ++c) {
            if (c < n_chunks) {
                const long long base = c * OW_FCHUNK;
                const int cn = (int)((n - base) < OW_FCHUNK ? (n - base) : OW_FCHUNK);
                if (voice_prog && !gave_up) {   // the voices of these jobs are being rendered beside this kernel: wait for the chunk (see k_job_chain_fused)
                    const int need = (int)(base + cn);
                    long spins = 0;
                    while (__hip_atomic_load(&voice_prog[jb >> 6], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < need) {
                        __builtin_amdgcn_s_sleep(32);
                        if (++spins > 5000000L) { gave_up = true; if (lane == 0) atomicOr(voice_err, 1); break; }
                    }
                }
                if (lane < 32) {
                    double x = 0.0;
                    if (jb + f_jl < n_jobs && f_n < cn) x = reed[(size_t)(jb + f_jl) * stride + base + f_n];
                    tin[f_jl * (OW_FCHUNK + 1) + f_n] = x;
                }
                OW_WAVE_SYNC();
                double (*slot)[8] = ring[c & 1];
                for (int sidx = 0; sidx < cn; ++sidx) {
                    const double x = tin[jl * (OW_FCHUNK + 1) + sidx];
                    if (osr == 2) {                                  // main.rs:445-466: per-sample up(1) -> 2x process (-> down(1) on the output wavefront)
                        const double y = allpass3(R.oc0, R.oc1, R.oc2, us, x);      // branch A on even lanes, B on odd ones
                        const double in[2] = {role ? 0.0 : rowb<0>(y), role ? 0.0 : rowb<1>(y)};
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            const double pk = preamp_step(in[k]);
                            if (role == 0 && (lane & 15) == 0) slot[sidx * 2 + k][jl] = pk;
                        }
                    } else {
                        const double pk = preamp_step(role ? 0.0 : x);
                        if (role == 0 && (lane & 15) == 0) slot[sidx * 2][jl] = pk;
                    }
                }
            }
            __syncthreads();
        }
        return;
    }

    // ---- the output-stage wavefront: lanes 0..7 own a job each, the others shadow them
    const int jl = lane & 7;
    const int j = jb + jl;
    const bool valid = j < n_jobs;
    const OwJobDev jd = jobs[valid ? j : n_jobs - 1];
    const double sr = K->sr;
    double da[3] = {0, 0, 0}, db[3] = {0, 0, 0}, dd = 0.0;
    SpeakerSt sp;
    sp.character = 1.0; sp.ts = 0.0;                         // Speaker::new(sr); set_character(c)  (main.rs:483-484)
    sp.hpf.s1 = sp.hpf.s2 = sp.lpf.s1 = sp.lpf.s2 = 0.0;
    speaker_update(sp, sr);
    speaker_set_character(sp, jd.speaker, sr);
    const double vol2_a = jd.volume;
    for (long long c = 0; c <= n_chunks; ++c) {
        if (c >= 1) {
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
