// openwurli-hip: batch-render kernels (lane = job).
//
// One job = `preamp-bench render` (tools/preamp-bench/src/main.rs:371-549) as driven by
// ml/render_model_notes.py:49-116: one voice (seed note*2654435761) -> fresh legacy preamp after
// reset() + set_ldr_resistance(r) (static LDR, tremolo depth 0) with per-sample 2x oversampling ->
// x volume^2 -> optional power amp at BASE rate -> Speaker(character) -> x POST_SPEAKER_GAIN, f64.
// Jobs share no state, so a wavefront renders 64 voices (k_job_voice) or 32 main/shadow preamp
// lane pairs (k_job_chain) of different jobs in lock-step.
#pragma once
#include "ow_kernels.h"
#include "ow_melange_dev.h"
#include "ow_melange_lit.h"

namespace owdev {

struct OwJobDev {           // device copy of ow_job
    uint8_t note, velocity, mlp, poweramp;
    uint8_t no_preamp, no_attack_noise, has_ds, pad8;
    double volume, speaker, r_ldr;
    double tremolo_depth;        // > 0: Tremolo::new(depth, preamp_sr) drives the LDR (main.rs:430-431), the static --ldr is ignored
    double displacement_scale;   // has_ds: Voice::set_displacement_scale (main.rs:406-408)
};
// what the chain kernel writes: the finished sample, or the power amp's input (volume^2 taper applied) when the melange power amp
// runs as its own launch between the preamp and the speaker (ow_batch_render with OW_POWER_AMP_MELANGE)
enum { JOB_OUT_FINAL = 0, JOB_OUT_PA_INPUT = 1 };

// Voice::note_on + Voice::render for n samples, lane = job.  reed[job][n] (row stride `stride`).
__global__ __launch_bounds__(64) void k_job_voice(const OwConsts* __restrict__ K, const double* __restrict__ nt, double* __restrict__ vrec,
                                                  const OwJobDev* __restrict__ jobs, double* __restrict__ reed, int n_jobs, long long n,
                                                  long long stride, int* __restrict__ prog = nullptr) {
    __shared__ double tile[64 * (OW_VCHUNK + 1)];
    const int lane = threadIdx.x;
    const int jb = blockIdx.x * 64;
    const int j = jb + lane;
    const bool active = j < n_jobs;
    double* rec = vrec + (size_t)blockIdx.x * OW_VREC_DOUBLES + lane;
    __shared__ double lcoef[OW_LCOEF_ROWS * 64];
    VoiceRegs v;
    if (active) {
        const OwJobDev jd = jobs[j];
        const int note = jd.note < OW_MIDI_LO ? OW_MIDI_LO : (jd.note > OW_MIDI_HI ? OW_MIDI_HI : jd.note);
        const double vel = (double)jd.velocity / 127.0;                    // main.rs:403
        double raw[11];
        mlp_raw_scalar(clampd(((double)note - 21.0) / (108.0 - 21.0), 0.0, 1.0), clampd(vel, 0.0, 1.0), raw);
        const MlpOut corr = mlp_finish(note, raw, jd.mlp != 0);
        note_on_lane(rec, nt, K, note, vel, (uint32_t)jd.note * 2654435761u, corr);   // main.rs:404-405
        if (jd.has_ds) rec[VF_DS * 64] = jd.displacement_scale;                        // --displacement-scale (pickup.rs:114-116)
        if (jd.no_attack_noise) rec[VF_NCNT * 64] = bitsd(dbits(rec[VF_NCNT * 64]) & 0xFFFFFFFF00000000ull);   // --no-attack-noise: remaining = 0 (hammer.rs:187-189)
        v.load(rec);
        lcoef_load(lcoef + lane, rec);
    }
    for (long long base = 0; base < n; base += OW_VCHUNK) {
        const int cn = (int)((n - base) < OW_VCHUNK ? (n - base) : OW_VCHUNK);
        for (int s = 0; s < cn; ++s) tile[lane * (OW_VCHUNK + 1) + s] = active ? v.step<false>(lcoef + lane) : 0.0;
        __syncthreads();
        // transposed, coalesced store: 2 job rows per pass (32 samples each)
        for (int r = (lane >> 5); r < 64; r += 2) {
            const int s = lane & 31;
            if (jb + r < n_jobs && s < cn) reed[(size_t)(jb + r) * stride + base + s] = tile[r * (OW_VCHUNK + 1) + s];
        }
        __syncthreads();
        // prog != nullptr: the chain kernel of these jobs runs BESIDE this one (k_job_chain_fused on another stream) and waits, chunk by
        // chunk, for the samples to exist: publish how far the block's 64 rows are written
        if (prog) {
            __threadfence();
            if (lane == 0) __hip_atomic_store(&prog[blockIdx.x], (int)(base + cn), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// Preamp + output stage, lane pair (job, main|shadow): 32 jobs per wavefront.  MEL selects the melange 12-node solver
// (`--features melange-preamp` build of preamp-bench) instead of the legacy 8-node one.  A job's LDR is static, so its matrices
// are built ONCE, by the reference's literal rebuild (ow_melange_lit.h: A -> LU -> S -> K in the reference's operation order), and
// live in LDS, job-minor -- the per-lane rank-one state this kernel used to carry spilled 707 registers.  They are rebuilt lazily
// when the adapter's NaN reset puts a state (and with it its resistance) back to the settled clone, as the reference's own
// matrices go back with the state.
template <bool MEL>
__global__ __launch_bounds__(64) void k_job_chain(const OwConsts* __restrict__ K, const OwJobDev* __restrict__ jobs, const double* __restrict__ reed,
                                                  double* __restrict__ out, const double* __restrict__ settled, int n_jobs, long long n, long long stride,
                                                  const double* __restrict__ trem_r = nullptr, int out_mode = JOB_OUT_FINAL) {
    __shared__ double tin[32 * (OW_PCHUNK + 1)];
    __shared__ double tout[32 * (OW_PCHUNK + 1)];
    __shared__ double LU_all[MEL ? 12 * 12 * 32 : 1];
    __shared__ double S_all[MEL ? 12 * 12 * 32 : 1];
    const int lane = threadIdx.x;
    const int jl = lane & 31, role = lane >> 5;
    const int jb = blockIdx.x * 32;
    const int j = jb + jl;
    const bool valid = j < n_jobs;
    const OwJobDev jd = jobs[valid ? j : n_jobs - 1];
    const int osr = K->oversample ? 2 : 1;
    const double sr = K->sr;

    // DkPreamp::new(preamp_sr); preamp.reset(); preamp.set_ldr_resistance(r_ldr)  (main.rs:432-441)
    DkSt st;
    MelSt ms;
    double r_ldr = 1000000.0;
    double g_ldr = 1.0 / r_ldr, g_prev = g_ldr;
    double* lu = LU_all + (MEL ? jl : 0);
    double* S = S_all + (MEL ? jl : 0);
    const double alpha = 2.0 * (K->os_sr * 1.0);       // gen_preamp.rs:1991-1992
    double s_pot = __longlong_as_double(0x7ff8000000000000LL);    // resistance the LDS matrices were built for (NaN: none yet)
    double kk[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    double an66 = 0.0;
    // --tremolo-depth > 0: the oscillator's CdS stream (one per call: Tremolo::new settles every job's copy to the same state) through
    // this job's vibrato-pot divider, set before every chain-rate sample (main.rs:447-463); else the static --ldr after reset()
    const bool use_trem = trem_r != nullptr && jd.tremolo_depth > 0.0;
    const double tdepth = clampd(jd.tremolo_depth, 0.0, 1.0);
    long long os_idx = 0;
    if (MEL) {
        mel_init_state(ms, settled);                  // new() and reset() both clone the settled state (melange_adapter.rs:22-29,88-93)
        ms.nan_resets = 0; ms.be_fallbacks = 0;
        if (!use_trem) mel_set_r(ms, jd.r_ldr);
    } else {
        dk_dc_reset(K, r_ldr, st);                   // new() and reset() both solve DC at the initial 1 Mohm
        if (!use_trem) {
            const double r_new = fmax(jd.r_ldr, 1000.0);
            if (fabs(r_new - r_ldr) > 0.01) { r_ldr = r_new; g_ldr = 1.0 / r_new; }
        }
    }
    // one preamp sample for this lane's state (main: audio, shadow: 0.0), returns main - pump with the adapter's NaN reset
    auto preamp_step = [&](double x) -> double {
        double o;
        if (trem_r != nullptr) {                      // wave-uniform: some job of the call has a tremolo
            const double r = trem_shunt(tdepth, trem_r[os_idx]);
            os_idx += 1;
            if (use_trem) {
                if (MEL) mel_set_r(ms, r);
                else {
                    const double r_new = fmax(r, 1000.0);                                  // set_ldr_resistance, dk_preamp_legacy.rs:620-626
                    if (fabs(r_new - r_ldr) > 0.01) { r_ldr = r_new; g_ldr = ow_div(1.0, r_new); }
                }
            }
        }
        if (MEL) {
            const double pot_main = __shfl(ms.pot, jl);
            if (__any(!(pot_main == s_pot))) {        // lazy rebuild (gen_preamp.rs:3408-3411); every lane takes part (barriers inside)
                // the generic rebuild (a real function: it runs once per job, its registers must not weigh on the sample loop);
                // bit-identical to the fast path the pool kernel prefers (tests/test_gpu_parity.py)
                double kt[3][3];
                mel_lit_rebuild(pot_main, role, alpha, lu, S, kt);
                for (int a = 0; a < 3; ++a)
                    for (int b = 0; b < 3; ++b) kk[a][b] = kt[a][b];
                s_pot = pot_main;
                const double g66 = PRE_G[6][6] + (ow_div(1.0, pot_main) - PRE_POT_0_G_NOM);
                an66 = alpha * PRE_C[6][6] - g66;
            }
            o = mel_process_lit(ms, x, K->m_aneg0, an66, S, kk, nullptr, 0);
        } else {
            o = dk_step(st, x, g_ldr, g_prev, K);
            g_prev = g_ldr;
        }
        const double other = xor32_t(o);
        double res = role ? (other - o) : (o - other);
        if (!isfinite(res)) {
            if (MEL) { mel_init_state(ms, settled); }
            else { dk_dc_reset(K, r_ldr, st); g_ldr = 1.0 / r_ldr; g_prev = g_ldr; }
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

    for (long long base = 0; base < n; base += OW_PCHUNK) {
        const int cn = (int)((n - base) < OW_PCHUNK ? (n - base) : OW_PCHUNK);
        for (int r = 0; r < 32; ++r) {
            double x = 0.0;
            if (jb + r < n_jobs && lane < cn) x = reed[(size_t)(jb + r) * stride + base + lane];
            tin[r * (OW_PCHUNK + 1) + lane] = x;
        }
        __syncthreads();
        for (int s = 0; s < cn; ++s) {
            const double x = tin[jl * (OW_PCHUNK + 1) + s];
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
            if (jd.no_preamp) pre = x;                  // --no-preamp: the reed signal itself (main.rs:425-427)
            // main.rs:487-496: volume^2 (audio taper) -> optional power amp at base rate -> speaker -> PSG
            const double att = pre * vol2_a * vol2_a;
            double y;
            if (out_mode == JOB_OUT_PA_INPUT) y = att;
            else {
                const double amp = jd.poweramp ? power_amp(att) : att;
                y = speaker_process(sp, amp, K->spk_thermal_alpha) * 7.498942093324558;
            }
            if (role == 0) tout[jl * (OW_PCHUNK + 1) + s] = y;
        }
        __syncthreads();
        for (int r = 0; r < 32; ++r)
            if (jb + r < n_jobs && lane < cn) out[(size_t)(jb + r) * stride + base + lane] = tout[r * (OW_PCHUNK + 1) + lane];
        __syncthreads();
    }
}

// Output stage behind a power amp that ran as its own launch: per job, amp[n] (normalised amp output) or att[n] (--no-poweramp) ->
// Speaker(character) -> x POST_SPEAKER_GAIN (main.rs:487-496).  lane = job; rows are read and written in place.
__global__ __launch_bounds__(64) void k_job_speaker(const OwConsts* __restrict__ K, const OwJobDev* __restrict__ jobs, const double* __restrict__ att,
                                                    const double* __restrict__ amp, double* __restrict__ out, int n_jobs, long long n, long long stride) {
    const int j = blockIdx.x * 64 + threadIdx.x;
    if (j >= n_jobs) return;
    const OwJobDev jd = jobs[j];
    SpeakerSt sp;
    sp.character = 1.0; sp.ts = 0.0;
    sp.hpf.s1 = sp.hpf.s2 = sp.lpf.s1 = sp.lpf.s2 = 0.0;
    speaker_update(sp, K->sr);
    speaker_set_character(sp, jd.speaker, K->sr);
    const double* src = (jd.poweramp ? amp : att) + (size_t)j * stride;      // all three buffers share the row stride
    double* dst = out + (size_t)j * stride;
    for (long long i = 0; i < n; ++i) dst[i] = speaker_process(sp, src[i], K->spk_thermal_alpha) * 7.498942093324558;
}

}  // namespace owdev
