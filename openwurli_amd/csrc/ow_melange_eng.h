// openwurli-hip: melange 12-node preamp, literal per-sample rebuild, column-streamed, LANE = ENGINE (round 4).
//
// k_preamp_mel_col (ow_melange_col.h) gives each solver state of an engine its own lane, and both lanes repeat the rebuild of the
// matrices -- which depends on R_ldr only and is the same for main and shadow (melange_adapter.rs:72-86 sets one resistance on both).
// Here one lane owns the engine: the factorisation, the twelve unit-column solves, the S N_i sums and K are computed ONCE, each column
// is folded into the v_pred sums of BOTH states (the same products in the same order as before, per state), and the per-state parts
// (build_rhs; Newton, update, guards) run once per state.
// Bit-identical to k_preamp_mel_col by construction: every state sees the same operations on the same operands
// (tests/test_gpu_parity.py::test_melange_lane_engine_kernel_is_bit_identical).
// Measured (round 4, one MI355X): 64 engines per wavefront instead of 32 needs >= 131 072 engines to put two wavefronts on every SIMD;
// there 39.2 ms per block against 40.4 for k_preamp_mel_col (-3 %, where the shared rebuild is 14.5 % of the instructions: two solver
// states per lane are 40 more doubles than 256 registers hold, and the compiler's spill traffic takes most of the saving back); with the
// per-state parts as a rolled loop over states in private memory 57 ms; at 65 536 engines 45 against 21 ms.  Opt-in: OW_MEL_ENG=1.
#pragma once
#include "ow_melange_col.h"

namespace owdev {

template <int COL>
__device__ inline void mel_eng_step(const OwConsts* __restrict__ K, const MelColT& T, const double ra, const double rb, double acc_a[12], double acc_b[12],
                                    double* __restrict__ sni) {
    using Z = MelColNz<COL>;
    double b[12];
    mel_col_solve<COL>(K, T, b);
#pragma unroll
    for (int i = 0; i < 12; ++i)
        if (Z::nz(i)) { acc_a[i] += b[i] * ra; acc_b[i] += b[i] * rb; }
    mel_col_fold_sni<COL>(b, sni);
}

__global__ __launch_bounds__(64, 2) void k_preamp_mel_eng(const OwConsts* __restrict__ K, double* __restrict__ cs,
                                                          const double* __restrict__ settled, const OwEngineArgs* __restrict__ args,
                                                          const OwEngineOut* __restrict__ eout, const double* __restrict__ sum, const OwTremSrc tsrc,
                                                          double* __restrict__ pre, double* __restrict__ noise, int I, int L,
                                                          int Lcap, int e0, int ne, int generic_only, double* __restrict__ lu_scratch, size_t lu_ld) {
    __shared__ double sni_all[36 * 64];
    const int lane = threadIdx.x;
    const int e = e0 + blockIdx.x * 64 + lane;
    const bool valid = e < e0 + ne;
    const int ec = valid ? e : (e0 + ne - 1);
    const int osr = K->oversample ? 2 : 1;
    const TremCol rc = trem_col(tsrc, I, ec);
    const double alpha = 2.0 * (K->os_sr * 1.0);                    // gen_preamp.rs:1991-1992
    double* sni = sni_all + lane;
    double* lu = lu_scratch + (size_t)2 * (valid ? e : I + (lane & 31));   // the main state's workspace column of k_preamp_mel_col

    MelSt st[2];                                                     // [0] main, [1] shadow
    double ua[3], ub[3];
    Smoother sd;
    {
        const int e = ec;
        smoother_load(sd, cs, I, e, CS_SM_DEPTH);
        if (args[e].set_flags & 1u) sd.retarget(args[e].depth_target, K->ramp_samples);
        mel_load(st[0], cs, I, e, CS_M_MAIN);
        mel_load(st[1], cs, I, e, CS_M_SHADOW);
        for (int i = 0; i < 3; ++i) { ua[i] = CSF(CS_OS_UA + i); ub[i] = CSF(CS_OS_UB + i); }
        const uint64_t fl = dbits(CSF(CS_FLAGS));
        if (fl & 1ull) {
            mel_init_state(st[0], settled);
            mel_init_state(st[1], settled);
            for (int i = 0; i < 3; ++i) { ua[i] = 0.0; ub[i] = 0.0; }
        }
    }
    uint32_t adapter_resets = 0;
    const bool nz_mine = noise != nullptr && valid;
    const bool nz_on = nz_mine && args[ec].noise_on != 0u;
    const double scale_half = K->m_noise_scale * 1.0 * args[ec].thermal_gain * 0.5;
    double* nzcol = noise ? noise + ec : nullptr;
    if (nz_mine && (dbits(CSF(CS_FLAGS)) & 1ull)) nz_reseed(nzcol, I);
    const bool force_generic = generic_only != 0;
    const bool has_in = valid && !eout[ec].sum_nonfinite;
    const double* row0 = (has_in && args[ec].main_mask) ? sum + ((size_t)0 * I + ec) * Lcap : nullptr;
    const double* row1 = (has_in && args[ec].steal_mask) ? sum + ((size_t)1 * I + ec) * Lcap : nullptr;
    auto voice_in = [&](int n) -> double {
        double x = 0.0;
        if (row0) x = row0[n];
        if (row1) x += row1[n];
        return x;
    };
    double x_next = voice_in(0);
    double r_next = trem_col_at(rc, 0u);
    const int n_os = L * osr;
    for (int n = 0; n < L; ++n) {
        const double x = x_next;
        if (n + 1 < L) x_next = voice_in(n + 1);
        const double depth = clampd(sd.next(), 0.0, 1.0);
        double in[2];
        if (osr == 2) {
            in[0] = allpass3(OW_OS_A0, OW_OS_A1, OW_OS_A2, ua, x);
            in[1] = allpass3(OW_OS_B0, OW_OS_B1, OW_OS_B2, ub, x);
        } else {
            in[0] = x;
            in[1] = 0.0;
        }
        for (int j = 0; j < osr; ++j) {
            const int s_i = n * osr + j;
            const size_t s_idx = (size_t)s_i;
            const double r_now = r_next;
            if (s_i + 1 < n_os) r_next = trem_col_at(rc, (uint32_t)(s_i + 1));
            const double r_sh = trem_shunt(depth, r_now);
            mel_set_r(st[0], r_sh);
            mel_set_r(st[1], r_sh);
            const double pot = st[0].pot;        // the matrices follow the main state's resistance (ow_melange_lit.h)
            const double* nzp = nullptr;
            if (nz_on && scale_half != 0.0) {
                const double sir10 = st[0].pot == 9.99999999999999854e4 ? PRE_NOISE_THERMAL_SQRT_INV_R_DEFAULT[10] : sqrt(1.0 / st[0].pot);
                nz_draw(nzcol, I, scale_half, sir10);
                nzp = nzcol;
            }
            const uint32_t nan_before = st[0].nan_resets;
            // ---- (1) per state: clamp, flush, cooldown, build_rhs
            double rhs[2][12], input[2];
            bool force_be[2];
#pragma unroll
            for (int s = 0; s < 2; ++s)
                mel_col_pre(st[s], s ? 0.0 : in[j], pot, alpha, K, s ? nullptr : nzp, I, rhs[s], input[s], force_be[s]);
            // ---- (2) once per engine: factor, unit columns, fold into both states' sums
            double vp[2][12];
            {
                double acc_a[12], acc_b[12];
#pragma unroll
                for (int i = 0; i < 12; ++i) { acc_a[i] = 0.0; acc_b[i] = 0.0; }
                bool fast = !force_generic;
                if (fast) {
                    MelColT T;
                    fast = mel_col_factor(K, pot, alpha, T);
                    if (__builtin_expect(fast, 1)) {
                        mel_eng_step<0>(K, T, rhs[0][0], rhs[1][0], acc_a, acc_b, sni);
                        mel_eng_step<1>(K, T, rhs[0][1], rhs[1][1], acc_a, acc_b, sni);
                        mel_eng_step<2>(K, T, rhs[0][2], rhs[1][2], acc_a, acc_b, sni);
                        mel_eng_step<3>(K, T, rhs[0][3], rhs[1][3], acc_a, acc_b, sni);
                        mel_eng_step<4>(K, T, rhs[0][4], rhs[1][4], acc_a, acc_b, sni);
                        mel_eng_step<5>(K, T, rhs[0][5], rhs[1][5], acc_a, acc_b, sni);
                        mel_eng_step<6>(K, T, rhs[0][6], rhs[1][6], acc_a, acc_b, sni);
                        mel_eng_step<7>(K, T, rhs[0][7], rhs[1][7], acc_a, acc_b, sni);
                        mel_eng_step<8>(K, T, rhs[0][8], rhs[1][8], acc_a, acc_b, sni);
                        mel_eng_step<9>(K, T, rhs[0][9], rhs[1][9], acc_a, acc_b, sni);
                        mel_eng_step<10>(K, T, rhs[0][10], rhs[1][10], acc_a, acc_b, sni);
                        mel_eng_step<11>(K, T, rhs[0][11], rhs[1][11], acc_a, acc_b, sni);
                    }
                }
                if (__builtin_expect(!fast, 0)) {
                    MelColGen g;
                    for (int i = 0; i < 12; ++i) g.rhs[i] = rhs[1][i];
                    mel_col_generic(pot, alpha, lu, lu_ld, &g);
                    for (int i = 0; i < 12; ++i) acc_b[i] = g.acc[i];
                    for (int i = 0; i < 12; ++i) g.rhs[i] = rhs[0][i];
                    mel_col_generic(pot, alpha, lu, lu_ld, &g);
                    for (int i = 0; i < 12; ++i) acc_a[i] = g.acc[i];
                    for (int k = 0; k < 3; ++k) for (int i = 0; i < 12; ++i) MCOL_SNI(k, i) = g.sni[k][i];
                }
#pragma unroll
                for (int i = 0; i < 12; ++i) { vp[0][i] = acc_a[i]; vp[1][i] = acc_b[i]; }
            }
            double kk[3][3];
            mel_col_kernel(sni, kk);
            // ---- (3) per state: Newton, update, guards
            double o[2];
#pragma unroll
            for (int s = 0; s < 2; ++s)
                o[s] = mel_col_post(st[s], input[s], force_be[s], vp[s], kk, sni, s ? nullptr : nzp, I);
            if (nz_mine && st[0].nan_resets != nan_before) nz_clear_lag(nzcol, I);
            double result = o[0] - o[1];
            if (!isfinite(result)) {
                mel_init_state(st[0], settled);
                mel_init_state(st[1], settled);
                if (nz_mine) nz_reseed(nzcol, I);
                result = 0.0;
                adapter_resets += 1u;
            }
            if (valid) pre[s_idx * I + e] = result;
        }
    }
    if (valid) {
        mel_store(st[0], cs, I, e, CS_M_MAIN);
        mel_store(st[1], cs, I, e, CS_M_SHADOW);
        for (int i = 0; i < 3; ++i) { CSF(CS_OS_UA + i) = ua[i]; CSF(CS_OS_UB + i) = ub[i]; }
        smoother_store(sd, cs, I, e, CS_SM_DEPTH);
        const uint64_t fl = dbits(CSF(CS_FLAGS));
        if (fl & 1ull) CSF(CS_FLAGS) = bitsd(fl & ~1ull);
        const uint32_t nr = adapter_resets + st[0].nan_resets;
        if (nr) {
            const uint64_t d = dbits(CSF(CS_DIAG));
            CSF(CS_DIAG) = bitsd((d & 0xFFFFFFFFull) | ((uint64_t)((uint32_t)(d >> 32) + nr) << 32));
        }
    }
}

}  // namespace owdev
