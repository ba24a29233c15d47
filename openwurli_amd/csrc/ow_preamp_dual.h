// openwurli-hip: k_preamp with lane = ENGINE -- the main and the shadow solver state of an engine in ONE lane (round 6).
//
// k_preamp (ow_kernels.h) gives the two DK states of an engine (DkPreamp::process_sample, dk_preamp_legacy.rs:586-618: main with the audio,
// shadow with 0.0, output = main - shadow) to the lanes l and l + 32 of a wavefront.  Both lanes then form what the two states SHARE -- the
// depth smoother, the shunt divider and 1 / R (tremolo.rs:152-167, :620-626), the Sherman-Morrison scalar and the four K entries of dk_step
// (functions of g_ldr alone), the up-sampler's two all-pass chains -- and the wave-uniform Newton loop of dk_step runs for as many sweeps
// as the slowest of the 64 lanes needs: a shadow state, which only follows the slow movement of R, sits through the sweeps of the main
// states beside it.  Here a lane forms the shared part once and steps its two states one after the other, each in a loop over states of
// ITS kind (64 mains, then 64 shadows); main - shadow needs no cross-lane exchange.  Same operations on the same operands for every value
// either way: bit-identical to k_preamp at the preamp tap (tests/test_gpu_parity.py::test_preamp_dual_is_bit_identical).
// MEASURED SLOWER and therefore off unless `preamp_dual` / OW_PREAMP_DUAL=1 asks for it: 9.1 against 5.9 ms per 131 072-engine block on one
// box (tools/ab_preamp_dual.sh).  Two inlined dk_steps per chain sample keep two sets of the step's 114 scalar constants in flight: the
// compiler issues the s_load groups of the second step early and parks them in vector-register lanes -- 140 v_writelane + 148 v_readlane
// per sample beside ~1 180 instructions of arithmetic (scheduling fences between the steps do not move them: the loads hang on nothing
// but the opaque pointer) -- at 255 vector registers.  Kept as the experiment it is; the shared part alone (dk_shared) costs nothing.
#pragma once
#include "ow_kernels.h"

namespace owdev {

#define OW_DCHUNK 32          // 64 engine rows x 32 samples of the voice sum: the 16.9 KB of k_preamp's 32 x 64 tile
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_preamp_dual(const OwConsts* __restrict__ K, double* __restrict__ cs, const OwEngineArgs* __restrict__ args,
                                                    const OwEngineOut* __restrict__ eout, const double* __restrict__ sum, const OwTremSrc tsrc,
                                                    double* __restrict__ pre, int I, int L, int Lcap, int e0, int ne) {
    __shared__ double tile[64 * (OW_DCHUNK + 1)];
    const int lane = threadIdx.x;
    const int eb = e0 + blockIdx.x * 64;
    const int e_raw = eb + lane;
    const bool valid = e_raw < e0 + ne;
    const int e = valid ? e_raw : (e0 + ne - 1);   // clamp so every lane runs the same (harmless) work
    const int osr = K->oversample ? 2 : 1;

    DkSt sm, ss;
    double ua[3], ub[3];
    double r_ldr, g_ldr, g_prev;
    Smoother sd;
    smoother_load(sd, cs, I, e, CS_SM_DEPTH);
    if (args[e].set_flags & 1u) sd.retarget(args[e].depth_target, K->ramp_samples);
    dk_load(sm, cs, I, e, CS_P_MAIN);
    dk_load(ss, cs, I, e, CS_P_SHADOW);
    for (int i = 0; i < 3; ++i) { ua[i] = CSF(CS_OS_UA + i); ub[i] = CSF(CS_OS_UB + i); }
    r_ldr = CSF(CS_P_RLDR); g_ldr = CSF(CS_P_GLDR); g_prev = CSF(CS_P_GPREV);
    {
        const uint64_t fl = dbits(CSF(CS_FLAGS));
        if (fl & 1ull) {  // deferred preamp.reset() + oversampler.reset() from the output NaN guard (engine.rs:450-457)
            dk_dc_reset(K, r_ldr, sm);
            ss = sm;                                 // (both states restart from the same DC solve, dk_preamp_legacy.rs:628-640)
            g_ldr = 1.0 / r_ldr; g_prev = g_ldr;
            for (int i = 0; i < 3; ++i) { ua[i] = 0.0; ub[i] = 0.0; }
        }
    }
    uint32_t nan_resets = 0;
    double sh_depth = __longlong_as_double(0x7FF8000000000000LL), sh_top = 0.0, sh_lower = 0.0;      // trem_shunt's depth-only part (NaN: nothing formed yet)
    // staging flags of engine row `lane`: bit0 = slot pass present, bit1 = steal pass present; 0 when the row is past the range or its
    // block is non-finite (engine.rs:499-501 zeroes it)
    int rowflag = 0;
    if (e_raw < e0 + ne && !eout[e_raw].sum_nonfinite) rowflag = (args[e_raw].main_mask ? 1 : 0) | (args[e_raw].steal_mask ? 2 : 0);
    const int e_last = e0 + ne - 1;
    const TremCol tcol = trem_col(tsrc, I, e);   // this engine's place on the shared trajectory, or the column of its phase group
    double rn[2];
    rn[0] = trem_col_at(tcol, 0u);
    rn[1] = osr == 2 ? trem_col_at(tcol, 1u) : 0.0;
    const int half = lane >> 5, cl = lane & 31;
    for (int base = 0; base < L; base += OW_DCHUNK) {
        const int cn = min(OW_DCHUNK, L - base);
        // stage 64 engine rows x 32 samples of the voice sum (slot pass + steal pass) through LDS, two rows per pass
        const int col = base + min(cl, cn - 1);
#pragma unroll 8
        for (int r2 = 0; r2 < 32; ++r2) {
            const int fa = __builtin_amdgcn_readlane(rowflag, 2 * r2), fb = __builtin_amdgcn_readlane(rowflag, 2 * r2 + 1);
            const int r = 2 * r2 + half;
            const int fl = half ? fb : fa;
            const int er = min(eb + r, e_last);
            const double a = sum[((size_t)0 * I + er) * Lcap + col];
            const double b = sum[((size_t)1 * I + er) * Lcap + col];
            double x = (fl & 1) ? a : 0.0;
            x = (fl & 2) ? x + b : x;
            tile[r * (OW_DCHUNK + 1) + cl] = x;
        }
        __syncthreads();
        for (int n = 0; n < cn; ++n) {
            const double x = tile[lane * (OW_DCHUNK + 1) + n];
            const double rc[2] = {rn[0], rn[1]};
            {
                const uint32_t nx = (uint32_t)(min(base + n + 1, L - 1) * osr);
                rn[0] = trem_col_at(tcol, nx);
                if (osr == 2) rn[1] = trem_col_at(tcol, nx + 1u);
            }
            const double depth = clampd(sd.next(), 0.0, 1.0);   // engine.rs:533-534, tremolo.rs:117-119
            if (__builtin_amdgcn_ballot_w64(!(depth == sh_depth)) != 0ull) {      // (k_preamp's form)
                sh_depth = depth;
                const double r_upper = 50000.0 * (1.0 - depth);
                sh_lower = 50000.0 * depth;
                sh_top = r_upper > 0.0 ? ow_div(r_upper * 18000.0, r_upper + 18000.0) : 0.0;
            }
            double in[2];
            if (osr == 2) {  // Oversampler::upsample_2x (oversampler.rs:108-121)
                in[0] = allpass3(OW_OS_A0, OW_OS_A1, OW_OS_A2, ua, x);
                in[1] = allpass3(OW_OS_B0, OW_OS_B1, OW_OS_B2, ub, x);
            } else {
                in[0] = x;
                in[1] = 0.0;
            }
            for (int j = 0; j < osr; ++j) {
                const size_t idx = (size_t)((base + n) * osr + j);
                const double branch = 680.0 + rc[j];
                const double low = sh_lower > 0.0 ? ow_div(sh_lower * branch, sh_lower + branch) : 0.0;
                const double r_new = fmax(sh_top + low, 1000.0);               // tremolo.rs:152-167; set_ldr_resistance, :620-626
                if (fabs(r_new - r_ldr) > 0.01) { r_ldr = r_new; g_ldr = ow_div(1.0, r_new); }
                const DkShared sh = dk_shared(g_ldr, K);
                const double om = dk_step(sm, in[j], sh, g_prev, K);             // :598
                const double os = dk_step(ss, 0.0, sh, g_prev, K);               // :599
                g_prev = g_ldr;                                                   // :604
                double result = om - os;                                          // main - pump, :608
                if (!isfinite(result)) {                                          // :610-615
                    dk_dc_reset(K, r_ldr, sm);
                    ss = sm;
                    g_ldr = 1.0 / r_ldr; g_prev = g_ldr;
                    result = 0.0;
                    nan_resets += 1u;
                }
                if (valid) pre[idx * I + e] = result;
            }
        }
        __syncthreads();
    }
    if (valid) {
        dk_store(sm, cs, I, e, CS_P_MAIN);
        dk_store(ss, cs, I, e, CS_P_SHADOW);
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

}  // namespace owdev
