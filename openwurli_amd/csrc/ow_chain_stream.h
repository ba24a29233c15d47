// openwurli-hip: preamp and output stage of a BIG pool as one launch (k_chain_stream).
//
// k_preamp and k_post (ow_kernels.h) both give a wavefront 32 engines: lanes = (engine, main | shadow solver state) in the preamp,
// lanes = (engine, oversample phase) in the output stage.  Here one wavefront alternates between the two per 64-sample chunk: preamp of
// chunk c (voice sums staged through LDS, results to `pre`), then the output stage of chunk c (power amp, half-band down, speaker, gain,
// f32), whose state is fetched from / returned to the chain-state rows per chunk so that it is not live during the preamp phase.
//
// Why: a block that goes to the host (render(&mut [f32]), engine.rs:425-462) used to leave the GPU idle behind the last kernel for the
// whole device-to-host copy (268 MB per 131 072-engine block = 4.9 ms of a 26.7 ms step).  With the output stage interleaved into the
// preamp's 6.8 ms, the f32 rows leave the chip all through the chain phase: the kernel stores them straight into the caller's pinned
// block (`out2`: mapped host memory, ow_host_alloc) 256 B at a time -- 30 GB/s on average against ~55 GB/s of PCIe -- and nothing
// trails the launch.  One launch instead of two, `pre` written and read back by the same wavefront 64 samples later.
//
// Every statement of the two phases is the statement of k_preamp / k_post<true>: bit-identical at the preamp tap and at the output
// (tests/test_gpu_parity.py::test_chain_stream_is_bit_identical).  Oversampled chains only (host rates below 88.2 kHz); the others keep
// the two launches.
#pragma once
#include "ow_kernels.h"

namespace owdev {

#define OW_SCHUNK 64
// Output stage of one chunk (the body of k_post<true>), out of line: its own register allocation -- inlined, the scalar values of the
// preamp phase stayed live across it and the compiler spilled ~120 SGPRs per sample to vector lanes inside this loop (+12 % on the
// launch).  State comes from and goes back to the chain-state rows; the f32 samples of the chunk go to the tile (`otile`, LDS).
typedef __attribute__((address_space(3))) float* OwLdsFloatPtr;
__device__ __noinline__ bool chain_stream_post_chunk(const OwConsts* __restrict__ K, double* __restrict__ cs, const OwEngineArgs* __restrict__ args,
                                                     const double* __restrict__ pre, OwLdsFloatPtr otile, int I, int e, int el, int phase, bool valid,
                                                     int base, int cn, uint32_t set_flags) {
    const double sr = K->sr;
    const double thermal_alpha = K->spk_thermal_alpha;
    bool nan_fired = false;
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
    if (base == 0) {
        if (set_flags & 2u) ss.retarget(args[e].spk_target, K->ramp_samples);
        if (set_flags & 4u) sv.retarget(args[e].vol_target, K->ramp_samples);
    }
    double pn = pre[((size_t)base * 2 + phase) * I + e];
    for (int n = 0; n < cn; ++n) {
        const double pc = pn;
        pn = pre[((size_t)(base + min(n + 1, cn - 1)) * 2 + phase) * I + e];     // one sample ahead, inside the chunk
        const double y = power_amp(pc * 0.25);
        const double yo = xor32_t(y);                                // engine.rs:536-553
        const double y0 = phase ? yo : y, y1 = phase ? y : yo;
        const double a = allpass3(OW_OS_A0, OW_OS_A1, OW_OS_A2, da, y0);
        const double b = allpass3(OW_OS_B0, OW_OS_B1, OW_OS_B2, db, y1);
        const double o = (a + dd) * 0.5;
        dd = b;
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
        if (phase == 0) otile[el * (OW_SCHUNK + 1) + n] = f;
    }
    if (valid && phase == 0) {
        for (int i = 0; i < 3; ++i) { CSF(CS_OS_DA + i) = da[i]; CSF(CS_OS_DB + i) = db[i]; }
        CSF(CS_OS_DD) = dd;
        const double* hp = &sp.hpf.b0; const double* lp = &sp.lpf.b0;
        for (int i = 0; i < 7; ++i) { CSF(CS_SPK_HPF + i) = hp[i]; CSF(CS_SPK_LPF + i) = lp[i]; }
        CSF(CS_SPK_CHAR) = sp.character; CSF(CS_SPK_A2) = sp.a2; CSF(CS_SPK_A3) = sp.a3; CSF(CS_SPK_TC) = sp.tc; CSF(CS_SPK_TS) = sp.ts;
        smoother_store(ss, cs, I, e, CS_SM_SPK);
        smoother_store(sv, cs, I, e, CS_SM_VOL);
    }
    return nan_fired;
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_chain_stream(const OwConsts* __restrict__ K, double* __restrict__ cs, const OwEngineArgs* __restrict__ args, OwEngineOut* __restrict__ eout,
                    const double* __restrict__ sum, const OwTremSrc tsrc, double* __restrict__ pre, float* __restrict__ out, int I, int L, int Lcap, int Lout,
                    int e0, int ne, float* __restrict__ out2, size_t ld2) {
    __shared__ double tile[32 * (OW_SCHUNK + 1)];            // preamp phase: voice sums of the chunk; output phase: its f32 rows
    float* otile = reinterpret_cast<float*>(tile);
    const int lane = threadIdx.x;
    const int el = lane & 31, role = lane >> 5;              // role: solver state in the preamp phase, oversample phase in the output phase
    const int eb = e0 + blockIdx.x * 32;
    const int e_raw = eb + el;
    const bool valid = e_raw < e0 + ne;
    const int e = valid ? e_raw : (e0 + ne - 1);             // clamp so every lane runs the same (harmless) work
    const int e_last = e0 + ne - 1;

    // ---- preamp state (k_preamp), in registers for the whole launch
    DkSt st;
    double ua[3], ub[3];
    double r_ldr, g_ldr, g_prev;
    Smoother sd;
    smoother_load(sd, cs, I, e, CS_SM_DEPTH);
    if (args[e].set_flags & 1u) sd.retarget(args[e].depth_target, K->ramp_samples);
    dk_load(st, cs, I, e, role ? CS_P_SHADOW : CS_P_MAIN);
    for (int i = 0; i < 3; ++i) { ua[i] = CSF(CS_OS_UA + i); ub[i] = CSF(CS_OS_UB + i); }
    r_ldr = CSF(CS_P_RLDR); g_ldr = CSF(CS_P_GLDR); g_prev = CSF(CS_P_GPREV);
    {
        const uint64_t fl = dbits(CSF(CS_FLAGS));
        if (fl & 1ull) {  // deferred preamp.reset() + oversampler.reset() from the output NaN guard (engine.rs:450-457)
            dk_dc_reset(K, r_ldr, st);
            g_ldr = 1.0 / r_ldr; g_prev = g_ldr;
            for (int i = 0; i < 3; ++i) { ua[i] = 0.0; ub[i] = 0.0; }
        }
    }
    uint32_t nan_resets = 0;
    int rowflag = 0;
    {
        const int er = eb + el;
        if (er < e0 + ne && !eout[er].sum_nonfinite) rowflag = (args[er].main_mask ? 1 : 0) | (args[er].steal_mask ? 2 : 0);
    }
    const TremCol tcol = trem_col(tsrc, I, e);
    double rn[2];
    rn[0] = trem_col_at(tcol, 0u);
    rn[1] = trem_col_at(tcol, 1u);
    bool nan_fired = false;
    const uint32_t set_flags = args[e].set_flags;

    for (int base = 0; base < L; base += OW_SCHUNK) {
        const int cn = min(OW_SCHUNK, L - base);
        // ================================================================ preamp of the chunk (k_preamp)
        const int col = base + min(lane, cn - 1);
#pragma unroll 8
        for (int r = 0; r < 32; ++r) {
            const int er = min(eb + r, e_last);
            const int fl = __builtin_amdgcn_readlane(rowflag, r);
            const double a = sum[((size_t)0 * I + er) * Lcap + col];
            const double b = sum[((size_t)1 * I + er) * Lcap + col];
            double x = (fl & 1) ? a : 0.0;
            x = (fl & 2) ? x + b : x;
            tile[r * (OW_SCHUNK + 1) + lane] = x;
        }
        __syncthreads();
        for (int n = 0; n < cn; ++n) {
            const double x = tile[el * (OW_SCHUNK + 1) + n];
            const double rc[2] = {rn[0], rn[1]};
            {
                const uint32_t nx = (uint32_t)(min(base + n + 1, L - 1) * 2);
                rn[0] = trem_col_at(tcol, nx);
                rn[1] = trem_col_at(tcol, nx + 1u);
            }
            const double depth = clampd(sd.next(), 0.0, 1.0);   // engine.rs:533-534, tremolo.rs:117-119
            double in[2];
            {   // Oversampler::upsample_2x (oversampler.rs:108-121); shadow input is 0.0 (dk_preamp_legacy.rs:599)
                const double a = allpass3(OW_OS_A0, OW_OS_A1, OW_OS_A2, ua, x);
                const double b = allpass3(OW_OS_B0, OW_OS_B1, OW_OS_B2, ub, x);
                in[0] = role ? 0.0 : a;
                in[1] = role ? 0.0 : b;
            }
            for (int j = 0; j < 2; ++j) {
                const size_t idx = (size_t)((base + n) * 2 + j);
                const double r_new = fmax(trem_shunt(depth, rc[j]), 1000.0);   // tremolo.rs:152-167; set_ldr_resistance, :620-626
                if (fabs(r_new - r_ldr) > 0.01) { r_ldr = r_new; g_ldr = ow_div(1.0, r_new); }
                const double o = dk_step(st, in[j], g_ldr, g_prev, K);
                g_prev = g_ldr;                                                   // :604
                const double other = xor32_t(o);
                double result = role ? (other - o) : (o - other);                 // main - pump, :608
                if (!isfinite(result)) {                                          // :610-615
                    dk_dc_reset(K, r_ldr, st);
                    g_ldr = 1.0 / r_ldr; g_prev = g_ldr;
                    result = 0.0;
                    nan_resets += 1u;
                }
                if (valid && role == 0) pre[idx * I + e] = result;
            }
        }
        __syncthreads();      // (a workgroup-scope release / acquire: the chunk of `pre` written above is read by other lanes below)
        // ================================================================ output stage of the chunk (k_post<true>)
        if (chain_stream_post_chunk(K, cs, args, pre, (OwLdsFloatPtr)otile, I, e, el, role, valid, base, cn, set_flags)) nan_fired = true;
        __syncthreads();
        for (int r = 0; r < 32; ++r) {
            const int er = eb + r;
            if (er < e0 + ne && lane < cn) {
                const float f = otile[r * (OW_SCHUNK + 1) + lane];
                out[(size_t)er * Lout + base + lane] = f;
                if (out2) out2[(size_t)(er - e0) * ld2 + base + lane] = f;
            }
        }
        __syncthreads();      // the rows above were written by lanes of the (engine, phase 0) half, read by all; and the state rows by phase 0 for both
    }
    if (!valid) return;
    dk_store(st, cs, I, e, role ? CS_P_SHADOW : CS_P_MAIN);
    if (role != 0) return;
    for (int i = 0; i < 3; ++i) { CSF(CS_OS_UA + i) = ua[i]; CSF(CS_OS_UB + i) = ub[i]; }
    CSF(CS_P_RLDR) = r_ldr; CSF(CS_P_GLDR) = g_ldr; CSF(CS_P_GPREV) = g_prev;
    smoother_store(sd, cs, I, e, CS_SM_DEPTH);
    uint64_t fl = dbits(CSF(CS_FLAGS));
    fl &= ~1ull;
    if (nan_resets) {
        const uint64_t d = dbits(CSF(CS_DIAG));
        CSF(CS_DIAG) = bitsd((d & 0xFFFFFFFFull) | ((uint64_t)((uint32_t)(d >> 32) + nan_resets) << 32));
    }
    if (nan_fired) {  // preamp.reset()/oversampler.reset() act on post-block state: the down-sampler here, the preamp/up half at the next block
        for (int i = 0; i < 3; ++i) { CSF(CS_OS_DA + i) = 0.0; CSF(CS_OS_DB + i) = 0.0; }
        CSF(CS_OS_DD) = 0.0;
        fl |= 1ull;
        eout[e].out_nonfinite = 1u;
    }
    CSF(CS_FLAGS) = bitsd(fl);
}

}  // namespace owdev
