// openwurli-hip: gfx950 kernels of the render path.
//
//   k_apply_ops   block = engine, lane = slot     note-on / damper / move-to-steal on voice records
//   k_voice(_steady) lane = sounding voice (packed across engines by the host), ordered per-engine LDS reduction
//   k_tremolo     lane = engine                    Twin-T oscillator + LDR -> R[n]
//   k_preamp      lane = (engine, main|shadow)     half-band up + DK preamp (main - shadow)
//   k_post        lane = engine                    power amp x2 -> half-band down -> speaker -> gain -> f32
//
// All global traffic is coalesced: voice records are field-major [field][64 slots], chain state and the
// R / preamp streams are engine-minor [..][I]; engine-major audio rows are transposed through LDS tiles.
#pragma once
#include "ow_vm.h"
#include "ow_chain_dev.h"

namespace owdev {

#define OW_DATA_DECL __device__ const
#define OW_GEN_DATA_REINCLUDE
#include "../../data/ow_gen_data.h"
#undef OW_GEN_DATA_REINCLUDE
#undef OW_DATA_DECL

// per-engine result block written by k_voice / k_post, read back by the host after every render
struct OwEngineOut {
    uint64_t silent_mask;     // slot voices that satisfy Voice::is_silent after this block
    uint64_t bad_main;        // slot voices that produced a non-finite sample
    uint64_t bad_steal;       // steal voices that produced a non-finite sample
    uint32_t sum_nonfinite;   // engine.rs:499 NaN guard condition (either pass)
    uint32_t out_nonfinite;   // engine.rs:450 output NaN guard fired this block
    uint32_t transient;       // after this block some slot voice is inside a damper / onset / attack-noise phase (host: general kernel
                              // next block); 2 = a voice in such a phase was found by the steady kernel (host classification bug)
    uint32_t pad;
};

// mlp_correction.rs:86-116, scalar lane version (same accumulation order as the reference)
__device__ inline void mlp_raw_scalar(double in0, double in1, double raw[11]) {
    double h1[16], h2[16];
    for (int i = 0; i < 16; ++i) {
        double sum = MLP_B1[i];
        sum += MLP_W1[i][0] * in0;
        sum += MLP_W1[i][1] * in1;
        h1[i] = sum > 0.0 ? sum : 0.0;
    }
    for (int i = 0; i < 16; ++i) {
        double sum = MLP_B2[i];
        for (int j = 0; j < 16; ++j) sum += MLP_W2[i][j] * h1[j];
        h2[i] = sum > 0.0 ? sum : 0.0;
    }
    for (int i = 0; i < 11; ++i) {
        double sum = MLP_B3[i];
        for (int j = 0; j < 16; ++j) sum += MLP_W3[i][j] * h2[j];
        raw[i] = sum * MLP_TARGET_STDS[i] + MLP_TARGET_MEANS[i];
    }
}

// ------------------------------------------------------------------ slot ops
#ifndef OW_APPLY_OPS_ATTR
#define OW_APPLY_OPS_ATTR __attribute__((amdgpu_waves_per_eu(2, 2)))
#endif
__device__ inline void mlp_raw_mfma(double* __restrict__ h, double in0, double in1, double raw[11]);   // ow_mlp_mfma.h

// Ops of one engine are applied in queue order per slot.  The wavefront advances in rounds: every lane takes its next
// pending op; the note-ons of a round share ONE batched MLP evaluation on the f64 matrix cores (ow_mlp_mfma.h).
// Capped at 256 registers (it would take ~500 to keep the MLP weights resident): a wavefront of this kernel then fits beside a
// tremolo wavefront on a SIMD, so the block-ahead oscillator can be launched before the host has prepared the ops.
__global__ __launch_bounds__(64) OW_APPLY_OPS_ATTR void k_apply_ops(const OwConsts* __restrict__ K, const double* __restrict__ nt, double* __restrict__ vrec,
                                                  const OwEngineArgs* __restrict__ args, const OwOp* __restrict__ ops_packed,
                                                  const uint32_t* __restrict__ engines, const OwOp* __restrict__ ops_fixed = nullptr,
                                                  const OwVm* __restrict__ vm = nullptr, int e_base = 0) {
    __shared__ double h[64 * 17];
    // vm != nullptr: the launch of ow_pool_midi's device burst itself -- block b is engine e_base + b, its queue is the one k_vm_events
    // just wrote at the engine's fixed place (vm[e].n_dev_ops entries); no args, no engine list
    const int e = vm ? e_base + (int)blockIdx.x : (int)engines[blockIdx.x];
    const int lane = threadIdx.x;
    OwEngineArgs a;
    if (vm) { a.op_count = vm[e].n_dev_ops; a.op_begin = 0x80000000u | (uint32_t)(e * OW_VM_OPS_MAX); }
    else a = args[e];
    if (a.op_count == 0) return;
    // bit 31 of op_begin: the queue was written on the device (k_vm_events) at the engine's fixed place, not packed and uploaded by the host
    const OwOp* __restrict__ ops = (a.op_begin & 0x80000000u) ? ops_fixed : ops_packed;
    a.op_begin &= 0x7FFFFFFFu;
    double* main_rec = vrec + ((size_t)e * 2 + 0) * OW_VREC_DOUBLES + lane;
    double* steal_rec = vrec + ((size_t)e * 2 + 1) * OW_VREC_DOUBLES + lane;
    // Which ops are addressed to this lane's slot: the queue is read once, 64 entries at a time (lane l holds entry 64 t + l), and one
    // ballot per slot value hands every lane the positions of ITS ops in the tile as a bit mask -- instead of every lane walking the
    // queue entry by entry behind a dependent global load (a whole-keyboard re-strike queues 192 ops per engine: 0.3 ms of load latency
    // per wavefront, 22 ms for 131 072 engines, against 8 ms for the note-ons themselves).  Longer queues than eight tiles keep the walk.
    constexpr int OPS_TILES = 8;
    uint64_t mine[OPS_TILES];
    const bool tiled = a.op_count <= 64u * OPS_TILES;
    if (tiled) {
#pragma unroll
        for (int t = 0; t < OPS_TILES; ++t) {
            mine[t] = 0ull;
            if ((uint32_t)t * 64u < a.op_count) {                  // wave-uniform
                const uint32_t idx = (uint32_t)t * 64u + (uint32_t)lane;
                const int sl = idx < a.op_count ? (int)ops[a.op_begin + idx].slot : -1;
                for (int v = 0; v < 64; ++v) {
                    const uint64_t m = __ballot(sl == v);
                    if (lane == v) mine[t] = m;
                }
            }
        }
    }
    uint32_t cursor = 0;
    for (;;) {
        // The record pointers pass through an opaque move once per round: left visible, the address of every field the round may touch
        // (rec + 512 f, beyond the 4 KB a load's immediate offset spans) is a loop-invariant 64-bit value the compiler forms AHEAD of the
        // loop -- ~200 register pairs, 430 registers spilled at this kernel's cap of 256, and a wavefront spent 0.55 ms of a
        // whole-keyboard re-strike waiting for its own scratch memory (131 072 engines: 18 ms).
        asm volatile("" : "+v"(main_rec), "+v"(steal_rec));
        OwOp op;
        op.type = 0;
        if (tiled) {                                               // next op addressed to this slot: lowest set bit of the lowest non-empty tile
            bool got = false;
#pragma unroll
            for (int t = 0; t < OPS_TILES; ++t) {
                if (!got && mine[t]) {
                    const int b = __builtin_ctzll(mine[t]);
                    mine[t] &= mine[t] - 1ull;
                    op = ops[a.op_begin + (uint32_t)t * 64u + (uint32_t)b];
                    got = true;
                }
            }
        } else {
            while (cursor < a.op_count) {
                const OwOp cand = ops[a.op_begin + cursor];
                ++cursor;
                if (cand.slot == lane) { op = cand; break; }
            }
        }
        if (!__any(op.type != 0)) break;
        const bool is_on = op.type == OP_NOTE_ON;
        if (__any(is_on)) {                      // wave-uniform: one MFMA batch for all note-ons of this round
            const double midi = is_on ? (double)op.note : 60.0;
            const double in0 = clampd((midi - 21.0) / (108.0 - 21.0), 0.0, 1.0);
            const double in1 = is_on ? clampd(op.velocity, 0.0, 1.0) : 0.0;
            double raw[11];
            mlp_raw_mfma(h, in0, in1, raw);
            if (is_on) {
                const MlpOut corr = mlp_finish((int)op.note, raw, op.mlp != 0);
                note_on_lane(main_rec, nt, K, (int)op.note, op.velocity, op.seed, corr);
            }
        }
        if (op.type == OP_DAMPER) {
            start_damper_lane(main_rec, K);
        } else if (op.type == OP_SET_DS) {
            main_rec[VF_DS * 64] = op.velocity;
        } else if (op.type == OP_MOVE_STEAL) {  // slot.steal_voice = slot.voice.take() (engine.rs:316-321)
            // sixteen loads in flight, then their stores (field by field the copy is 103 dependent load -> store round trips per
            // wavefront: the two records may alias as far as the compiler knows)
            constexpr int CP = 16;
            for (int f0 = 0; f0 < VF_COUNT; f0 += CP) {
                double t[CP];
#pragma unroll
                for (int k = 0; k < CP; ++k) t[k] = f0 + k < VF_COUNT ? main_rec[(f0 + k) * 64] : 0.0;
#pragma unroll
                for (int k = 0; k < CP; ++k) if (f0 + k < VF_COUNT) steal_rec[(f0 + k) * 64] = t[k];
            }
            steal_rec[VF_STEAL * 64] = bitsd((uint64_t)op.seed | ((uint64_t)op.seed << 32));
        }
    }
}

// ------------------------------------------------------------------ voices
// Packed dispatch: lane = SOUNDING VOICE, not voice slot.  The host deals the sounding voices of consecutive engines into blocks of
// up to 64 (an engine is never split), in (engine, slot) order; `entries[block * 64 + lane]` = (engine << 6) | slot or OW_NO_VOICE
// behind the last one.  An engine that sounds 8 voices then costs an eighth of a wavefront instead of a whole one; with all 64 keys
// down a block is one engine as before.  The ordered voice sum (engine.rs:469-479) is taken per engine over its lanes, which are in
// slot order, so it is still the reference's sequential sum -- without the +0.0 terms of the silent slots.
// k_voice: capped at 256 registers (two wavefronts per SIMD).  Uncapped the compiler takes ~350 (one wavefront per SIMD, 42-110 of them
// accumulator registers it shuffles through): a re-struck pool's three general blocks ran 17 % slower.
#ifndef OW_VOICE_GENERAL_ATTR
#define OW_VOICE_GENERAL_ATTR __attribute__((amdgpu_waves_per_eu(2, 2)))
#endif
#define OW_VCHUNK 24   // 64 voices x 24 samples x f64 = 12.8 KB tile: 8 voice blocks + 4 tremolo blocks fit the 160 KB LDS of a CU
#define OW_NO_VOICE 0xFFFFFFFFu

struct VoiceLanes {           // per-lane view of one packed block
    int e, slot;
    bool active;
    uint64_t seg_end;         // bit l: lane l is the last voice of its engine in this block (wave-uniform)
    int nvalid;               // entries are contiguous from lane 0
};
OW_DEV VoiceLanes voice_lanes(const uint32_t* __restrict__ entries, int* __restrict__ eng_l) {
    const int lane = threadIdx.x;
    const uint32_t ent = entries[(size_t)blockIdx.x * 64 + lane];
    VoiceLanes w;
    w.active = ent != OW_NO_VOICE;
    w.e = w.active ? (int)(ent >> 6) : -1;
    w.slot = (int)(ent & 63u);
    eng_l[lane] = w.e;
    const int e_next = __shfl_down(w.e, 1);
    w.seg_end = __ballot(w.active && (lane == 63 || e_next != w.e));
    w.nvalid = __popcll(__ballot(w.active));
    return w;
}
// Sum the voices of each engine of the block in slot order for sample (base + lane) and write its row of sum[pass][engine][.].
// HALVES (the steady kernel and its variants): a block that is one engine with all 64 voices is summed as (v0 + ... + v31) + (v32 + ... +
// v63), both halves of the wavefront at work on 32 voices each, instead of 64 dependent additions on half of its lanes -- the chain
// the wavefront waits for at every chunk edge is half as long.  One rounding of the engine.rs:469-479 order moves (the class of
// deviation 9; k_voice keeps the strict order).
template <int CH = OW_VCHUNK, int RS = CH + 1, bool HALVES = false>
OW_DEV void voice_reduce(const double* __restrict__ tile, const int* __restrict__ eng_l, const VoiceLanes& w, int cn, int base, int pass,
                         double* __restrict__ sum, OwEngineOut* __restrict__ eout, int I, int Lcap) {
    const int lane = threadIdx.x;
    if (HALVES && CH <= 32 && w.nvalid == 64 && w.seg_end == (1ull << 63)) {      // (wave-uniform)
        const int j = lane & 31, h = lane >> 5;
        double acc = 0.0;
        if (j < cn) {
#pragma unroll 16
            for (int l = 0; l < 32; ++l) acc += tile[(32 * h + l) * RS + j];
        }
        const double hi = __shfl(acc, j + 32);
        if (h == 0 && j < cn) {
            acc += hi;
            const int e = eng_l[0];
            if (!isfinite(acc)) atomicOr(&eout[e].sum_nonfinite, 1u);
            sum[((size_t)pass * I + e) * Lcap + base + j] = acc;
        }
        return;
    }
    if (lane < cn) {
        if (w.seg_end == (1ull << (w.nvalid - 1))) {   // one engine in the block (every block when all keys are down): no segment tests
            double acc = 0.0;
            if (w.nvalid == 64) {
#pragma unroll 16
                for (int l = 0; l < 64; ++l) acc += tile[l * RS + lane];
            } else {
#pragma unroll 4
                for (int l = 0; l < w.nvalid; ++l) acc += tile[l * RS + lane];
            }
            const int e = eng_l[0];
            if (!isfinite(acc)) atomicOr(&eout[e].sum_nonfinite, 1u);
            sum[((size_t)pass * I + e) * Lcap + base + lane] = acc;
            return;
        }
        double acc = 0.0;
        for (int l = 0; l < w.nvalid; ++l) {
            acc += tile[l * RS + lane];
            if ((w.seg_end >> l) & 1ull) {
                const int e = eng_l[l];
                if (!isfinite(acc)) atomicOr(&eout[e].sum_nonfinite, 1u);
                sum[((size_t)pass * I + e) * Lcap + base + lane] = acc;
                acc = 0.0;
            }
        }
    }
}

// Does the steal variant of the steady kernel (k_voice_steady<false, 2>, below) render this block's engine?  One test for both kernels.
OW_DEV bool voice_steal_takes(bool active, uint64_t sample, uint64_t onset_n, uint32_t noise_rem, uint32_t flags, double rate6) {
    const bool other_phase = active && (sample < onset_n || noise_rem > 0u || ((flags & 1u) && !(rate6 <= 0.125)));
    return __ballot(other_phase) == 0ull;
}

// General step (any phase).  pass 0 = slot voices of the engines the host classified as "in a transient phase", pass 1 = steal voices
// (one engine per block there, so the crossfade early-out below is per engine).
// pass | 2 = the voice-sum NaN guard's second render (engine.rs:496-521): after a block whose voice sum was non-finite the reference
// renders every remaining voice of the engine AGAIN to find the culprit, which advances the survivors by another `L` samples.  The same
// stepping runs here with nothing summed or written but the voice records and the status bits; a steal voice's crossfade counter is
// not touched (the reference decrements it in the first pass only).
//
// Onset gains (reed.rs:251-264: a cosine, and for mid velocities a power -- ~200 instructions) are never evaluated inside the sample
// loop: evaluated there they cost EVERY sample of the wavefront that price for as long as one lane is inside its ramp (the bass keys of
// a re-struck engine: 1 200 samples with a dozen lanes left in it).  They depend on the voice's own sample counter only, so each chunk
// starts with a tabulation pass that is lane-parallel over (voice, sample): four voices x 16 samples per pass, ceil(k / 4) passes for k
// lanes in their ramp -- the in-loop price when all 64 are (the first 2 ms after a whole-keyboard strike), a sixteenth of it for the
// last few.  The table needs no LDS of its own: the gains of lane l's chunk are parked in lane l's row of the voice-sum tile, which that
// lane reads (gain of sample n) just before it overwrites the slot (output of sample n).  Same function, same arguments: same bits.
#ifndef OW_KCHUNK
#define OW_KCHUNK 24      // 64 voices x 24 samples: 12.8 KB tile, seven workgroups per CU with the 9.7 KB coefficient table (16: +28 % with no phase active -- the chunk edge costs more than the tabulation saves)
#endif
__global__ __launch_bounds__(64) OW_VOICE_GENERAL_ATTR void k_voice(const OwConsts* __restrict__ K, double* __restrict__ vrec, const uint32_t* __restrict__ entries,
                                              double* __restrict__ sum, OwEngineOut* __restrict__ eout, int I, int L, int Lcap, int pass) {
    constexpr int CH = OW_KCHUNK;
    __shared__ double tile[64 * (CH + 1)];
    __shared__ double lcoef[OW_LCOEF_ROWS * 64];
    __shared__ int eng_l[64];
    const int lane = threadIdx.x;
    const VoiceLanes w = voice_lanes(entries, eng_l);
    const bool active = w.active;
    const bool guard = (pass & 2) != 0;
    const bool skip_taken = (pass & 4) != 0;                   // pass | 4: engines the steal variant of the steady kernel renders are skipped
    pass &= 1;
    double* rec = vrec + ((size_t)(active ? w.e : 0) * 2 + pass) * OW_VREC_DOUBLES + w.slot;

    if (skip_taken) {                    // (ahead of the record: the blocks that leave here are most of a re-strike's launch)
        uint64_t smp = 0ull, on_n = 0ull;
        uint32_t nrem = 0u, fl = 0u;
        double rate6 = 0.0;
        if (active) {
            smp = dbits(rec[VF_SAMPLE * 64]); on_n = dbits(rec[VF_ONSET_N * 64]); nrem = (uint32_t)dbits(rec[VF_NCNT * 64]);
            fl = (uint32_t)dbits(rec[VF_FLAGS * 64]); rate6 = rec[(VF_DRATE + 6) * 64];
        }
        if (voice_steal_takes(active, smp, on_n, nrem, fl, rate6)) return;
    }
    VoiceRegs v;
    uint32_t steal_fade = 0, steal_len = 1;
    if (active) {
        v.load(rec);
        lcoef_load(lcoef + lane, rec);   // read back by this lane only
        if (pass) {
            const uint64_t sf = dbits(rec[VF_STEAL * 64]);
            steal_fade = (uint32_t)sf;
            steal_len = (uint32_t)(sf >> 32);
        }
    }
    __syncthreads();                     // eng_l
    bool bad_voice = false;
    for (int base = 0; base < L; base += CH) {
        const int cn = min(CH, L - base);
        // Steal pass: once every crossfade of this engine has run out (gain (fade - i)/len == 0 from here on, engine.rs:483-489)
        // the rest of the block only adds voice x 0.0, and the voices are dropped after the block (steal_fade reaches 0): stop
        // stepping them.  (The reference keeps rendering them; only a voice turning non-finite inside its last 5 ms would differ.)
        if (pass && !guard && __all(!active || steal_fade <= (uint32_t)base)) {
            double* row = sum + ((size_t)pass * I + eng_l[0]) * Lcap;
            for (int i = base + lane; i < L; i += 64) row[i] = 0.0;
            break;
        }
        // ---- onset gains of the chunk into the tile rows of the lanes inside their ramp, four voices per pass
        const uint64_t m_on = __ballot(active && v.sample < v.onset_n);
        {
            constexpr int SLOT = CH <= 16 ? 16 : 32, VP = 64 / SLOT;              // lanes per voice of a pass, voices per pass
            const int sub = lane / SLOT, j = lane % SLOT;
            for (uint64_t m = m_on; m; ) {
                // the sub-th of the next (up to) VP set bits of m for this lane's part of the wavefront
                int l = -1;
                uint64_t mm = m;
                for (int k = 0; k < VP; ++k) {
                    const int b = mm ? __builtin_ctzll(mm) : -1;
                    if (k == sub) l = b;
                    mm &= mm - 1ull;
                }
                m = mm;                                                          // wave-uniform: the VP lowest bits are consumed
                const int src = l < 0 ? 0 : l;
                const unsigned long long s0 = __shfl((unsigned long long)v.sample, src), on = __shfl((unsigned long long)v.onset_n, src);
                const double inc = __shfl(v.onset_inc, src), ex = __shfl(v.onset_exp, src);
                if (l >= 0 && j < cn) tile[l * (CH + 1) + j] = (s0 + (unsigned long long)j < on) ? onset_gain((double)(s0 + (unsigned long long)j), inc, ex) : 1.0;
            }
        }
        __syncthreads();
        const double* my_gt = tile + lane * (CH + 1);
        for (int n = 0; n < cn; ++n) {
            double o = 0.0;
            if (active) {
                o = v.step<false, true>(lcoef + lane, my_gt + n, nullptr);      // (my_gt is only dereferenced inside the ramp)
                if (pass) {  // 5 ms linear crossfade, engine.rs:483-489
                    const uint32_t i = (uint32_t)(base + n);
                    const uint32_t remaining = steal_fade > i ? steal_fade - i : 0u;
                    o = o * ow_div((double)remaining, (double)steal_len);       // (= the IEEE quotient, tests/test_gpu_division.py)
                }
            }
            tile[lane * (CH + 1) + n] = o;
        }
        if (active && !v.state_finite()) bad_voice = true;
        __syncthreads();
        if (!guard) voice_reduce<CH>(tile, eng_l, w, cn, base, pass, sum, eout, I, Lcap);
        __syncthreads();
    }
    if (active) {
        if (pass && !guard) {  // slot.steal_fade.saturating_sub(len) (engine.rs:490)
            const uint32_t l32 = (uint32_t)L;
            steal_fade = steal_fade > l32 ? steal_fade - l32 : 0u;
            rec[VF_STEAL * 64] = bitsd((uint64_t)steal_fade | ((uint64_t)steal_len << 32));
        }
        v.store(rec);
        OwEngineOut* o = eout + w.e;
        const unsigned long long bit = 1ull << w.slot;
        if (pass == 0) {
            if (v.is_silent(rec)) atomicOr((unsigned long long*)&o->silent_mask, bit);
            if (bad_voice) atomicOr((unsigned long long*)&o->bad_main, bit);
            if (v.in_transient()) o->transient = 1u;
        } else if (bad_voice) {
            atomicOr((unsigned long long*)&o->bad_steal, bit);
        }
    }
}


// ------------------------------------------------------------------ voices, steady-state fast path
// Slot voices of engines in which NO voice is inside a damper phase, an onset ramp or an attack-noise burst: the
// per-sample work is then exactly {OU jitter every 16 samples, 7-mode rotation, renorm every 1024, pickup}.  Those
// conditions are monotone between events, so one test at block start covers the whole block.  The kernel carries
// only the fields that path needs (fewer registers) and has no phase tests in its sample loop; engines that fail the
// test are left to k_voice.  Arithmetic per block is the same as VoiceRegs::step<true>.
// (the envelope k samples into a block in the folded form: env0 * d^k by the library's pow -- correctly rounded to an ulp where the
// reference's k-fold product carries ~sqrt(k) ulp of its own; binary powers of d would carry k / 2.  Out of line: once per 1 024 samples.)
__device__ __noinline__ __attribute__((const)) double env_after(double env0, double d, uint32_t k) {
    return env0 * pow(d, (double)k);
}
// FOLD (the steady kernel proper, round 5): the envelope is not a recurrence of its own but the RADIUS of the quadrature pair -- the
// rotation coefficients carry decay_mult (folded in where they are formed, every 16th sample), `ae` holds the constant amplitude, and
// the seven `ae *= decay` of every sample are gone (78 -> 72 vector instructions per voice-sample).  (s, c) enter the block multiplied
// by the record's envelope and leave it divided by env0 * d^L; the renormalisation every 1 024 samples (reed.rs:292-299) resets the
// radius to env0 * d^k instead of 1.  Same mathematics; roundings move by ~1e-16 per step, as with deviation 9.
template <bool FOLD>
struct VoiceSteadyT {
    // ae = amplitude * envelope carried as ONE recurrence (ae *= decay_mult): the modal sum is then one FMA per mode, where
    // reed.rs:276 multiplies amplitude * s * onset * envelope (two multiplies + add with onset == 1).  The product is reassociated and
    // rounded once per sample instead of twice: a relative random walk of ~1e-16 per sample against the reference's envelope, 3e-14
    // after 10 s of a voice's life, inside the 1e-12 bar of the voice-sum tap (the general kernel keeps the reference's order).  The
    // record keeps `envelope`: it is recovered as ae / amplitude when the block ends.
    double s[7], c[7], ae[7], drift[7], cos_inc[7], sin_inc[7], phase_inc[7], decay[7];
    double ci[7], si[7];   // jitter-corrected rotation (reed.rs:281-283): depends on drift only, which changes every 16 samples
    double q, ds, gain;
    double beta, revert, diffusion;   // the voice's own pickup beta / jitter constants (VF_BETA..), read once per kernel: a load inside the
                                      // sample loop is re-issued every sample (the noinline saturate call may write memory) and stalls the wave
    // The voice's sample counter only matters at every 16th sample (jitter, reed.rs:262) and every 1024th (renormalisation): the loop
    // carries a countdown to the next multiple of 16 (compare + decrement per sample) instead of a 64-bit counter with two mask tests
    // (7 VALU per voice-sample); the counter itself is rebuilt at those events and at the block's end.
    uint64_t next_evt;        // sample index of the next jitter update
    uint32_t cd;              // samples until it (0: this sample)
    uint32_t renorm;          // this sample is also a renormalisation point
    uint32_t jitter_state;
    // ATTACK variant only (k_voice_steady<false, true>: engines inside onset ramps / attack-noise bursts, no voice damping): samples left of
    // the onset ramp (reed.rs:251-264; the gains themselves are tabulated per chunk into the lane's tile row) and the noise burst's state
    // (hammer.rs:150-179)
    uint32_t on_rem, noise_rem, noise_fade, noise_rng;
    double namp, ns1, ns2, ndecay;
    // STEAL variant only (k_voice_steady<false, 2>: the steal voices of an engine during their 5 ms crossfade, usually damping): the damper's
    // clock and ramp length (reed.rs:228-247; the seven rates -- the seven multipliers once the ramp is over -- sit in the lane's LDS column)
    double dt, dramp, dramp_y;   // (dramp_y = ow_rcp_refined(dramp): the per-sample quotient t / ramp is ow_div instruction for instruction)
    uint32_t dflags;          // bit 0 damper active, bit 1 ramp done (VF_FLAGS)

    OW_DEV void set_sample(uint64_t sample) {
        cd = (16u - ((uint32_t)sample & 15u)) & 15u;
        next_evt = sample + (uint64_t)cd;
        renorm = 0u;
    }
    // MEM (the STEAL variant): cos_inc / sin_inc / phase_inc are read from the voice record `rc` at every update (once in 16 samples)
    // instead of living in 42 registers -- the registers the seven damper polynomials need; with them resident that loop spilled on
    // every sample.
    template <bool MEM = false>
    OW_DEV void update_rotation(const double* __restrict__ rc = nullptr, const volatile int* zero = nullptr) {
#ifndef OW_STRICT_FP
#pragma clang fp contract(fast)
#endif
        if constexpr (MEM) {
            // (the record's address behind a zero the compiler cannot see through -- a volatile LDS word -- so that the loads stay here:
            // left alone they are hoisted out of the sample loop, back into registers; volatile global reads bypass the caches; an opaque
            // zero from inline asm counts as a convergent operation and the loop is then no longer unrolled by two)
            rc += *zero;
#pragma unroll
            for (int m = 0; m < 7; ++m) {
                const double cinc = rc[(VF_COS_INC + m) * 64], sinc = rc[(VF_SIN_INC + m) * 64], pinc = rc[(VF_PHASE_INC + m) * 64];
                const double delta_phase = drift[m] * pinc;
                ci[m] = cinc - delta_phase * sinc;
                si[m] = sinc + delta_phase * cinc;
            }
        } else {
#pragma unroll
            for (int m = 0; m < 7; ++m) {
                const double delta_phase = drift[m] * phase_inc[m];
                ci[m] = cos_inc[m] - delta_phase * sin_inc[m];
                si[m] = sin_inc[m] + delta_phase * cos_inc[m];
                if (FOLD) { ci[m] *= decay[m]; si[m] *= decay[m]; }
            }
        }
    }

    // One reed sample is split in three so the kernel can software-pipeline it: the pickup division chain of sample n-1
    // (pickup(), ~25 dependent f64 ops) is emitted in the same basic block as the 7 independent rotations of sample n
    // (advance()), with the rare branches (jitter / renormalise / saturate) at the block edges.  Same arithmetic, same order
    // per value as VoiceRegs::step<true>; only the instruction interleaving changes.
    template <bool MEM = false>
    OW_DEV void jitter(const double* __restrict__ rc = nullptr, const volatile int* zero = nullptr) {   // reed.rs:262-283, every 16th sample of this voice
#ifndef OW_STRICT_FP
#pragma clang fp contract(fast)
#endif
        if (cd == 0u) jitter_due<MEM>(rc, zero);
    }
    template <bool MEM = false>
    OW_DEV void jitter_due(const double* __restrict__ rc = nullptr, const volatile int* zero = nullptr) {   // the update itself (the caller knows this voice's sample counter is at a multiple of 16)
#ifndef OW_STRICT_FP
#pragma clang fp contract(fast)
#endif
        {
            cd = 16u;
            const uint64_t sample = next_evt;
            next_evt += 16ull;
            renorm = (((uint32_t)sample & 1023u) == 0u && sample > 0ull) ? 1u : 0u;      // reed.rs:292 (tested after the rotation, below)
#pragma unroll
            for (int m = 0; m < 7; ++m) {
                jitter_state = lcg(jitter_state);
                const double u = OW_DIV_C((double)(jitter_state >> 1), OW_JITTER_DIV);
                const double noise = (u * 2.0 - 1.0) * 1.7320508080;
                drift[m] = revert * drift[m] + diffusion * noise;
            }
            update_rotation<MEM>(rc, zero);
        }
    }
    // gain: this sample's slot of the lane's tile row (ATTACK: the onset gain sits there while the ramp lasts); nco: the lane's column of
    // the attack-noise BPF coefficients b0, b1, b2, a1, a2 in LDS (nco[i * 64]); fade16: the sixteen fade-in values of hammer.rs:165
    // DAMP: dtab = the lane's column of damper rates (dtab[m * 64]; the multipliers once the ramp is done), rec = its voice record
    template <bool ATTACK = false, bool DAMP = false>
    OW_DEV double advance(const double* __restrict__ gain = nullptr, const double* __restrict__ nco = nullptr, const double* __restrict__ fade16 = nullptr,
                          double* __restrict__ dtab = nullptr, const double* __restrict__ rec = nullptr) {
#ifndef OW_STRICT_FP
#pragma clang fp contract(fast)
#endif
        if (DAMP) {                                          // reed.rs:228-247, ahead of the modal sum; the factors land on ae = amplitude * envelope
            // Branch-free but for the (wave-uniform, once per voice) end of a ramp: every lane evaluates the seven polynomials and a
            // select picks polynomial / multiplier / 1 -- lane-dependent branches here cut the loop's scheduling region in pieces.
            const bool damp = (dflags & 1u) != 0u;
            dt += damp ? 1.0 : 0.0;
            if (__builtin_amdgcn_ballot_w64(damp && !(dflags & 2u) && dt > dramp) != 0ull) {
                if (damp && !(dflags & 2u) && dt > dramp) {
                    dflags |= 2u;
#pragma unroll
                    for (int m = 0; m < 7; ++m) dtab[m * 64] = rec[(VF_DMULT + m) * 64];
                }
            }
            const bool in_ramp = damp && !(dflags & 2u);
            const double tr = ow_div_y(dt, dramp, dramp_y);   // damper_ramp_pos: ow_div(t, ramp) with the ramp's reciprocal refined once (unused garbage outside a ramp)
#pragma unroll
            for (int m = 0; m < 7; ++m) {
                const double k = dtab[m * 64];               // rate inside the ramp (<= 1/8: the kernel's entry test), multiplier after it
                const double pf = exp_neg_poly(k * tr);
                ae[m] *= in_ramp ? pf : (damp ? k : 1.0);    // (x * 1.0 == x)
            }
        }
        double sum = 0.0;
#pragma unroll
        for (int m = 0; m < 7; ++m) {
            sum += s[m] * ae[m];                           // onset == 1.0 (x * 1.0 == x)
            const double s_new = s[m] * ci[m] + c[m] * si[m];
            const double c_new = c[m] * ci[m] - s[m] * si[m];
            s[m] = s_new;
            c[m] = c_new;
            if (!FOLD) ae[m] *= decay[m];
        }
        if (ATTACK && on_rem > 0u) {                         // reed.rs:276: every term carries the onset gain -- here the sum does (one
            sum *= *gain;                                    // multiply instead of seven: the products round differently, <= 1 ulp of the sum)
            on_rem -= 1u;
        }
        // (reed.rs adds the modal sum to a buffer of zeros, and the noise burst on top: `0.0 + sum` only turns a -0.0 sum into +0.0, which
        // nothing downstream of a noiseless sample can tell -- 1 - y, |y| and the products are the same)
        double x = ATTACK ? 0.0 + sum : sum;
        // (both additions as lane-dependent branches.  Written branch-free -- selects, one loop copy per phase set picked per chunk, as
        // the steal variant's damper step is -- this loop spills: the block after a re-strike took 38.7 instead of 27.3 ms, and 68 ms
        // with the rotation constants read from the record as well)
        if (ATTACK && noise_rem > 0u) {                      // hammer.rs:150-179, as VoiceRegs::step
            double e = 1.0;
            if (noise_fade > 0u) { e = fade16[16u - noise_fade]; noise_fade -= 1u; }
            noise_rng = lcg(noise_rng);
            const double nz = OW_DIV_C((double)(int32_t)noise_rng, 2147483647.0);   // (= the IEEE quotient for all 2^32 draws, tests/test_gpu_division.py)
            const double yb = nco[0] * nz + ns1;
            ns1 = nco[64] * nz - nco[192] * yb + ns2;
            ns2 = nco[128] * nz - nco[256] * yb;
            x += namp * e * yb;
            namp *= ndecay;
            noise_rem -= 1u;
        }
        double y = x * ds;
        const double ay = fabs(y);
        if (renorm) {
            renorm = 0u;
            // FOLD: the radius this pair should have now = the envelope after this sample: env0 * d^k, k = samples into the block
            // including this one (jitter_due has just moved next_evt 16 past this sample's index)
            const uint32_t k = FOLD ? (uint32_t)(next_evt - 15ull - dbits(rec[VF_SAMPLE * 64])) : 0u;
#pragma unroll
            for (int m = 0; m < 7; ++m) {
                if (FOLD) {
                    // (on the pair divided by its target radius: the squares of a fast-decaying upper mode's 1e-160 would underflow.
                    // A mode whose envelope has left the normal range is left alone: it adds nothing to any sum ever again.)
                    const double env_t = env_after(rec[(VF_ENV + m) * 64], decay[m], k);
                    if (env_t > 1e-290) {
                        const double u = ow_div(s[m], env_t), w = ow_div(c[m], env_t);
                        const double r_inv = 1.0 / sqrt(u * u + w * w);
                        s[m] = (u * r_inv) * env_t;
                        c[m] = (w * r_inv) * env_t;
                    }
                } else {
                    const double r_sq = s[m] * s[m] + c[m] * c[m];
                    const double r_inv = 1.0 / sqrt(r_sq);
                    s[m] *= r_inv;
                    c[m] *= r_inv;
                }
            }
        }
        cd -= 1u;
        if (!(ay < 0.94)) y = pickup_saturate_hi(y);
        return y;
    }
    OW_DEV double pickup(double y) {                         // pickup.rs 1/(1-y) bilinear HPF + voice gain
#ifndef OW_STRICT_FP
#pragma clang fp contract(fast)
#endif
        const double omy = 1.0 - y;
        const double alpha = beta * omy;
        const double q_next = ow_div(q * (1.0 - alpha) + 2.0 * beta, 1.0 + alpha);
        q = q_next;
        return (q_next * omy - 1.0) * gain;                    // gain = 1.8375 * post_pickup_gain, formed once per block (deviation 9's class: <= 1 ulp of the sample)
    }
};

// Skewed clocks (round 4).  Every voice updates its jitter when ITS sample counter is a multiple of 16 (reed.rs:262).  Voices struck at
// one sample share that grid, and the wavefront takes the update branch once in 16 samples; the voices of played input do not, some lane
// is due at almost every sample, and the whole wavefront walks through the ~85-instruction update almost every sample (measured: 1.7 x the
// kernel time).  The update needs nothing but the lane's own state, so the lanes may run on shifted clocks: lane l works on its sample
// n = i - d_l in loop trip i, with d_l in 0..15 chosen so that all updates of the wavefront fall on the trips with i % 16 == g -- a
// uniform branch taken once in 16 trips again.  The block then takes L + max d trips (the first and last max d of them with part of the
// lanes masked), the voice-sum tile becomes a ring of two 16-sample chunks whose reduction runs one chunk behind, and every voice sees
// exactly the operations it saw before: bit-identical (tests/test_gpu_parity.py::test_skewed_voice_clocks_are_bit_identical).
// A wavefront whose voices already share a grid (d = 0 for all: the all-keys chord of the benchmark) keeps the plain loop.
#define OW_SKEW_CH 16
#define OW_SKEW_RING 32
// Row stride of the ring.  A lane writes column (i - d_l) & 31 of its row: the LDS bank is 2 ((RS l + i - d_l) mod 32), so lanes l, l' of a
// half-wavefront collide when RS (l - l') = d_l - d_l' (mod 32).  With RS = 33 a chord rolled one sample per key (d_l = l mod 16) puts
// sixteen lanes on one bank (measured: 1.4 x the kernel time); with RS = 35 that pattern is a two-way conflict, uniform columns (the plain
// loop, the reduction's reads) stay conflict-free (3 is odd), and 64 x 35 doubles x 8 wavefronts still fit the CU's LDS.
#define OW_SKEW_RS 35
// Two instantiations, so that each loop gets its own register allocation (one kernel holding both took 322 registers, or spilled inside
// the plain loop when capped): SKEW = false is the plain loop, which also REPORTS whether some wavefront of the launch held more than one
// jitter grid (skew_seen); the host launches the skewed variant for the next block then -- both give the same samples, so a stale choice
// only costs time, and the phases of sounding voices relative to each other never change between note events.
// ATTACK (round 5): the plain loop for engines whose slot voices are inside onset ramps and attack-noise bursts but NOT damping (what a
// note-on leaves behind; a release puts the engine in k_voice).  Same pipelined loop with two per-lane additions -- the onset gain,
// tabulated ahead of every 32-sample chunk into the lane's tile row (two voices per pass, lane-parallel over samples; read just before the
// slot is overwritten), and the noise burst -- at 1.2-1.7 x the steady price instead of the general kernel's 2.6 x: the four blocks
// after a whole-keyboard re-strike were 19 + 47 + 37 + 17 ms there.  Values differ from k_voice's in the last bit (fused steps, the
// onset gain on the sum): the same class as deviation 9, inside the voice-sum bar.
// STEAL (PHASE = 2, round 5): the steal voices of one engine per block (the host's steal list) for the 5 ms of their crossfade
// (engine.rs:483-490) -- released voices inside or past their damper ramp, as a re-struck key leaves them behind.  The loop adds the
// damper's factors (seven exp_neg_poly per sample during the ramp, rates in the lane's LDS column) and the crossfade gain; 24-sample
// chunks, so that the rate table fits beside the tile at eight workgroups per CU.  A block decides by itself whether it may: an engine
// with a steal voice still inside its onset ramp or noise burst (stolen within 40 ms of its strike), or with a damper rate above 1/8
// (host rates below 16 kHz), returns at once and is rendered by k_voice, which skips the others (voice_steal_takes: one test, both
// kernels).  The general kernel spent 14.0 + 13.0 ms on the two sub-blocks of a whole-pool re-strike's crossfade.
// RELEASE (PHASE = 3): the same damper step for the SLOT voices of the general list -- engines with a released key that still sounds
// (engine.rs:340-374: what note-off and pedal-up leave behind for the ~170 ms until the voice is freed) -- packed like every slot list,
// several engines per block; the same test decides per BLOCK (any voice of it inside an onset ramp or noise burst: k_voice's).
template <bool SKEW, int PHASE = 0>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_voice_steady(const OwConsts* __restrict__ K, double* __restrict__ vrec, const uint32_t* __restrict__ entries,
                                                     double* __restrict__ sum, OwEngineOut* __restrict__ eout, int I, int L, int Lcap, uint32_t* __restrict__ skew_seen) {
    constexpr bool ATTACK = PHASE == 1, STEAL = PHASE == 2, RELEASE = PHASE == 3;
    constexpr bool DAMPV = STEAL || RELEASE;                  // the variants with the damper step
    constexpr int PCH = DAMPV ? 24 : 32;                      // chunk of the plain loop (24 -> 32: a 512-sample block is sixteen whole chunks; 12.9 against 13.05 ms)
    constexpr int TILE_DOUBLES = SKEW ? 64 * OW_SKEW_RS : 64 * (PCH + 1);   // (the skewed variant's short blocks and aligned wavefronts use the plain loop in the ring's tile)
    constexpr int PRS = SKEW ? OW_SKEW_RS : PCH + 1;          // ... and its row stride
    constexpr int PASS = STEAL ? 1 : 0;                       // which record of the slot and which row of the sums
    static_assert(!(SKEW && PHASE != 0), "the attack / steal / release variants are the plain loop");
    __shared__ double tile[TILE_DOUBLES];
    __shared__ double nco[ATTACK ? 5 * 64 + 16 : (DAMPV ? 7 * 64 : 1)];   // attack-noise BPF coefficients per lane, then the sixteen fade-in values / damper rates per lane
    __shared__ int eng_l[64];
    __shared__ int zero_l[1];                                 // (DAMPV: update_rotation<true>)
    const int lane = threadIdx.x;
    if (DAMPV && lane == 0) zero_l[0] = 0;
    const VoiceLanes w = voice_lanes(entries, eng_l);
    const bool active = w.active;
    double* rec = vrec + ((size_t)(active ? w.e : 0) * 2 + PASS) * OW_VREC_DOUBLES + w.slot;
    VoiceSteadyT<!DAMPV> v;                                   // (a damper multiplies the envelope every sample: those variants keep the recurrence)
    constexpr bool FOLD = !DAMPV;
    uint32_t noise_rng = 0;
    v.on_rem = 0u; v.noise_rem = 0u; v.noise_fade = 0u; v.noise_rng = 0u; v.namp = 0.0; v.ns1 = 0.0; v.ns2 = 0.0; v.ndecay = 0.0;
    v.dt = 0.0; v.dramp = 0.0; v.dramp_y = 0.0; v.dflags = 0u;
    uint32_t steal_fade = 0u, steal_len = 1u;
    double steal_len_d = 1.0, steal_len_y = 1.0;
    if (DAMPV) {
        uint64_t smp = 0ull, on_n = 0ull;
        uint32_t nrem = 0u, fl = 0u;
        double rate6 = 0.0;
        if (active) {
            smp = dbits(rec[VF_SAMPLE * 64]); on_n = dbits(rec[VF_ONSET_N * 64]); nrem = (uint32_t)dbits(rec[VF_NCNT * 64]);
            fl = (uint32_t)dbits(rec[VF_FLAGS * 64]); rate6 = rec[(VF_DRATE + 6) * 64];
        }
        if (!voice_steal_takes(active, smp, on_n, nrem, fl, rate6)) return;      // k_voice renders this block
    }
    if (ATTACK && lane < 16) nco[5 * 64 + lane] = noise_fade_env((double)lane / 16.0);      // the values k_voice forms per sample (hammer.rs:165)
    if (active) {
        // The host sends an engine here only if its status after the previous block said "no transient phase" and no note event
        // arrived since; those phases never start by themselves.  A voice found inside one means that bookkeeping is wrong.
        const uint32_t flags = (uint32_t)dbits(rec[VF_FLAGS * 64]);
        const uint64_t smp = dbits(rec[VF_SAMPLE * 64]), onset_n = dbits(rec[VF_ONSET_N * 64]);
        const uint32_t noise_rem = (uint32_t)dbits(rec[VF_NCNT * 64]);
        if (ATTACK) {
            if (flags & 1u) eout[w.e].transient = 2u;      // a damping voice: the host's classification is wrong
            v.on_rem = smp < onset_n ? (uint32_t)min((unsigned long long)(onset_n - smp), 0xFFFFFFFFull) : 0u;
            const uint64_t nc = dbits(rec[VF_NCNT * 64]);
            v.noise_rem = (uint32_t)nc; v.noise_fade = (uint32_t)(nc >> 32);
            v.namp = rec[VF_NAMP * 64]; v.ns1 = rec[VF_NS1 * 64]; v.ns2 = rec[VF_NS2 * 64]; v.ndecay = rec[VF_NDECAY * 64];
#pragma unroll
            for (int i = 0; i < 5; ++i) nco[i * 64 + lane] = rec[(VF_NB0 + i) * 64];
        } else if (DAMPV) {
            v.dflags = flags & 3u; v.dt = rec[VF_DCOUNT * 64]; v.dramp = rec[VF_DRAMP * 64]; v.dramp_y = ow_rcp_refined(v.dramp);
#pragma unroll
            for (int i = 0; i < 7; ++i) nco[i * 64 + lane] = rec[(((flags & 2u) ? VF_DMULT : VF_DRATE) + i) * 64];
            if (STEAL) {
                const uint64_t sf = dbits(rec[VF_STEAL * 64]);
                steal_fade = (uint32_t)sf; steal_len = (uint32_t)(sf >> 32);
                steal_len_d = (double)steal_len; steal_len_y = ow_rcp_refined(steal_len_d);
            }
        } else if ((flags & 1u) || smp < onset_n || noise_rem > 0u) eout[w.e].transient = 2u;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            if (FOLD) { const double env0 = rec[(VF_ENV + i) * 64]; v.s[i] = rec[(VF_S + i) * 64] * env0; v.c[i] = rec[(VF_C + i) * 64] * env0; v.ae[i] = rec[(VF_AMP + i) * 64]; }
            else { v.s[i] = rec[(VF_S + i) * 64]; v.c[i] = rec[(VF_C + i) * 64]; v.ae[i] = rec[(VF_AMP + i) * 64] * rec[(VF_ENV + i) * 64]; }
            v.drift[i] = rec[(VF_DRIFT + i) * 64]; v.decay[i] = rec[(VF_DECAY + i) * 64];
            if (!DAMPV) { v.cos_inc[i] = rec[(VF_COS_INC + i) * 64]; v.sin_inc[i] = rec[(VF_SIN_INC + i) * 64]; v.phase_inc[i] = rec[(VF_PHASE_INC + i) * 64]; }
        }
        v.q = rec[VF_Q * 64]; v.ds = rec[VF_DS * 64]; v.gain = 1.8375 * rec[VF_GAIN * 64];
        v.beta = rec[VF_BETA * 64]; v.revert = rec[VF_JREV * 64]; v.diffusion = rec[VF_JDIFF * 64];
        v.set_sample(dbits(rec[VF_SAMPLE * 64]));
        const uint64_t r = dbits(rec[VF_RNG * 64]);
        v.jitter_state = (uint32_t)r; noise_rng = (uint32_t)(r >> 32);
        v.noise_rng = noise_rng;
        v.template update_rotation<DAMPV>(rec, zero_l);
    }
    __syncthreads();                     // eng_l, nco
    // ---- which trips carry the jitter updates: g minimising the longest delay over the phases present in this wavefront
    int g = 0, D = 0, d = 0;
    {
        const uint32_t cd0 = v.cd;                                  // samples until this voice's next update (0 = its first sample here)
        uint32_t present = 0;                                       // bit b: some voice has cd0 == b
        for (int b = 0; b < 16; ++b) present |= (__ballot(active && cd0 == (uint32_t)b) != 0ull ? 1u : 0u) << b;
        if (!SKEW) {
            if (PHASE == 0 && lane == 0 && (present & (present - 1u)) != 0u && L >= 2 * OW_SKEW_CH) atomicOr(skew_seen, 1u);
        } else if (L >= 2 * OW_SKEW_CH) {
        int best = 16;
        for (int gg = 0; gg < 16; ++gg) {
            int mx = 0;
            for (int b = 0; b < 16; ++b) if ((present >> b) & 1u) mx = max(mx, (gg - b) & 15);
            if (mx < best) { best = mx; g = gg; }
        }
        D = best;
        d = (int)((uint32_t)(g - (int)cd0) & 15u);
        if (lane == 0 && D > 0) atomicOr(skew_seen, 1u);
        }
    }
    if (SKEW && D > 0) {
        constexpr int RS = OW_SKEW_RS;
        double* trow = tile + lane * RS;
        const int n_trips = L + D;
        int wcol = 0;                                               // ring column of this lane's next sample (n & 31)
        int red = 0;                                                // next chunk to reduce
        const int n_chunks = (L + OW_SKEW_CH - 1) / OW_SKEW_CH;
        for (int i0 = 0; i0 < n_trips; i0 += OW_SKEW_CH) {
            const int i1 = min(i0 + OW_SKEW_CH, n_trips);
            if (active) {
                // One copy of the masked step and one of the pipelined loop: segment 0 = the trips of this chunk before every lane has
                // started, then the trips in which every lane works, segment 1 = the trips after the first lanes have finished.
                int i = i0;
                int nseg = 2;
                asm volatile("" : "+s"(nseg));                      // keeps the segment loop rolled (one masked body, not two)
#pragma unroll 1
                for (int seg = 0; seg < nseg; ++seg) {
                    if (seg == 1) {
                        const int f1 = min(i1, L);
                        if (i < f1) {
                            if ((i & 15) == g) v.jitter_due();
                            double y = v.template advance<false, false>(nullptr, nullptr, nullptr, nullptr, rec);
                            ++i;
#pragma unroll 2
                            for (; i < f1; ++i) {
                                if ((i & 15) == g) v.jitter_due();
                                trow[wcol] = v.pickup(y);
                                wcol = (wcol + 1) & (OW_SKEW_RING - 1);
                                y = v.template advance<false, false>(nullptr, nullptr, nullptr, nullptr, rec);
                            }
                            trow[wcol] = v.pickup(y);
                            wcol = (wcol + 1) & (OW_SKEW_RING - 1);
                        }
                    }
                    const int m1 = seg ? i1 : min(i1, D);
#pragma unroll 1
                    for (; i < m1; ++i) {
                        if (i >= d && i - d < L) {
                            if ((i & 15) == g) v.jitter_due();
                            const double y = v.template advance<false, false>(nullptr, nullptr, nullptr, nullptr, rec);
                            trow[wcol] = v.pickup(y);
                            wcol = (wcol + 1) & (OW_SKEW_RING - 1);
                        }
                    }
                }
            }
            __syncthreads();
            // chunks every lane has finished: chunk r is complete after trip 16 r + (its length - 1) + D
            while (red < n_chunks) {
                const int cn = min(OW_SKEW_CH, L - red * OW_SKEW_CH);
                if (red * OW_SKEW_CH + cn - 1 + D > i1 - 1) break;
                voice_reduce<OW_SKEW_RING, OW_SKEW_RS, true>(tile + ((red * OW_SKEW_CH) & (OW_SKEW_RING - 1)), eng_l, w, cn, red * OW_SKEW_CH, 0, sum, eout, I, Lcap);
                ++red;
            }
            __syncthreads();
        }
    } else
    for (int base = 0; base < L; base += PCH) {
        const int cn = min(PCH, L - base);
        if (STEAL && __all(!active || steal_fade <= (uint32_t)base)) {   // every crossfade of the engine has run out (as k_voice, pass 1)
            double* row = sum + ((size_t)PASS * I + eng_l[0]) * Lcap;
            for (int i = base + lane; i < L; i += 64) row[i] = 0.0;
            break;
        }
        if (ATTACK) {
            // onset gains of the chunk into the tile rows of the lanes inside their ramp: two voices per pass, 32 samples each.  The
            // ramp's parameters travel from the owning lane by cross-lane reads (fetched from the voice record in every pass -- three
            // dependent global loads behind two shuffles -- the passes cost 5 % more)
            const bool ramp = active && v.on_rem > 0u;
            const uint64_t m_on = __ballot(ramp);
            uint32_t my_on = 0u;
            double my_inc = 0.0, my_exp = 0.0;
            if (ramp) { my_on = (uint32_t)dbits(rec[VF_ONSET_N * 64]); my_inc = rec[VF_ONSET_INC * 64]; my_exp = rec[VF_ONSET_EXP * 64]; }
            const int sub = lane >> 5, j = lane & 31;
            for (uint64_t m = m_on; m; ) {
                const int l0 = __builtin_ctzll(m);
                m &= m - 1ull;
                const int l1 = m ? __builtin_ctzll(m) : -1;
                if (l1 >= 0) m &= m - 1ull;
                const int l = sub ? l1 : l0;
                const int src = l < 0 ? 0 : l;
                const uint32_t rem_l = (uint32_t)__shfl((int)v.on_rem, src), on_l = (uint32_t)__shfl((int)my_on, src);
                const double inc_l = __shfl(my_inc, src), exp_l = __shfl(my_exp, src);
                if (l >= 0 && j < cn && (uint32_t)j < rem_l)
                    tile[l * PRS + j] = onset_gain((double)((uint64_t)on_l - (uint64_t)rem_l + (uint64_t)j), inc_l, exp_l);
            }
            __syncthreads();
        }
        if (active) {
            double* trow = tile + lane * PRS;
            // 5 ms linear crossfade of a steal voice, engine.rs:483-489 (as k_voice, pass 1)
            auto faded = [&](double o, int n) {
                if (!STEAL) return o;
                const uint32_t i = (uint32_t)(base + n);
                const uint32_t remaining = steal_fade > i ? steal_fade - i : 0u;
                return o * ow_div_y((double)remaining, steal_len_d, steal_len_y);     // ow_div(remaining, len), the divisor's reciprocal refined once
            };
            v.template jitter<DAMPV>(rec, zero_l);
            double y = v.template advance<ATTACK, DAMPV>(trow, nco + lane, nco + 5 * 64, nco + lane, rec);
            auto one = [&](int n) {
                v.template jitter<DAMPV>(rec, zero_l);
                trow[n - 1] = faded(v.pickup(y), n - 1);
                y = v.template advance<ATTACK, DAMPV>(trow + n, nco + lane, nco + 5 * 64, nco + lane, rec);
            };
            // two samples per trip: the compiler renames the pipelined state instead of copying it back (7 v_mov_b64 per sample)
            if (DAMPV) {          // (by hand: the ballot in the damper step is a convergent operation, which the unroller leaves alone)
                int n = 1;
                for (; n + 1 < cn; n += 2) { one(n); one(n + 1); }
                if (n < cn) one(n);
            } else {
#pragma unroll 2
                for (int n = 1; n < cn; ++n) one(n);
            }
            trow[cn - 1] = faded(v.pickup(y), cn - 1);
        }
        __syncthreads();
        voice_reduce<PCH, PRS, true>(tile, eng_l, w, cn, base, PASS, sum, eout, I, Lcap);
        __syncthreads();
    }
    if (active) {
        bool fin = isfinite(v.q);
        bool all_quiet = true;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            // envelope = ae / amplitude (amplitude read again here rather than held in 14 registers across the block); a mode without
            // amplitude contributes nothing whatever its envelope is: it keeps decaying by the closed form
            const double amp = rec[(VF_AMP + i) * 64];
            if (FOLD) {
                // back to the record's form: envelope env0 * d^L, (s, c) divided by it.  An envelope that has left the normal range (a
                // fast-decaying upper mode of a long-held key, 1e-290 and falling) cannot give the pair back: it keeps its direction at
                // unit length (or (0, 1) once the pair itself has underflowed) -- a mode at that level adds nothing to any sum ever again.
                const double env = env_after(rec[(VF_ENV + i) * 64], v.decay[i], (uint32_t)L);
                double so = v.s[i], co = v.c[i];
                if (env > 1e-290) { so = ow_div(so, env); co = ow_div(co, env); }
                else {
                    const double r = sqrt(so * so + co * co);
                    if (r > 0.0) { so = so / r; co = co / r; } else { so = 0.0; co = 1.0; }
                }
                fin = fin && isfinite(v.s[i]) && isfinite(v.c[i]);
                rec[(VF_S + i) * 64] = so; rec[(VF_C + i) * 64] = co; rec[(VF_ENV + i) * 64] = env; rec[(VF_DRIFT + i) * 64] = v.drift[i];
                all_quiet = all_quiet && (fabs(amp * env) <= 1e-4);
            } else {
            const double env = amp != 0.0 ? v.ae[i] / amp : rec[(VF_ENV + i) * 64] * pow(v.decay[i], (double)L);
            rec[(VF_S + i) * 64] = v.s[i]; rec[(VF_C + i) * 64] = v.c[i]; rec[(VF_ENV + i) * 64] = env; rec[(VF_DRIFT + i) * 64] = v.drift[i];
            fin = fin && isfinite(v.s[i]) && isfinite(v.c[i]) && isfinite(v.ae[i]);
            all_quiet = all_quiet && (fabs(v.ae[i]) <= 1e-4);
            }
        }
        rec[VF_Q * 64] = v.q;
        rec[VF_SAMPLE * 64] = bitsd(dbits(rec[VF_SAMPLE * 64]) + (uint64_t)L);
        if (ATTACK) {
            noise_rng = v.noise_rng;
            rec[VF_NAMP * 64] = v.namp; rec[VF_NS1 * 64] = v.ns1; rec[VF_NS2 * 64] = v.ns2;
            rec[VF_NCNT * 64] = bitsd((uint64_t)v.noise_rem | ((uint64_t)v.noise_fade << 32));
            if (v.on_rem > 0u || v.noise_rem > 0u) eout[w.e].transient = 1u;      // still inside a phase: this variant again next block
        }
        rec[VF_RNG * 64] = bitsd((uint64_t)v.jitter_state | ((uint64_t)noise_rng << 32));
        const unsigned long long bit = 1ull << w.slot;
        if (DAMPV) {
            rec[VF_DCOUNT * 64] = v.dt;
            const uint64_t fl = dbits(rec[VF_FLAGS * 64]);
            rec[VF_FLAGS * 64] = bitsd((fl & ~3ull) | (uint64_t)v.dflags);
        }
        if (STEAL) {
            const uint32_t l32 = (uint32_t)L;                                      // slot.steal_fade.saturating_sub(len) (engine.rs:490)
            steal_fade = steal_fade > l32 ? steal_fade - l32 : 0u;
            rec[VF_STEAL * 64] = bitsd((uint64_t)steal_fade | ((uint64_t)steal_len << 32));
            if (!fin) atomicOr((unsigned long long*)&eout[w.e].bad_steal, bit);
        } else if (RELEASE) {
            // Voice::is_silent (voice.rs:183-188): -80 dB on every mode, or ten seconds under the damper; a voice that still damps keeps its
            // engine in the transient class (as k_voice reports it)
            const bool timed_out = (v.dflags & 1u) && ow_div(v.dt, rec[VF_VSR * 64]) > 10.0;
            if (all_quiet || timed_out) atomicOr((unsigned long long*)&eout[w.e].silent_mask, bit);
            if (!fin) atomicOr((unsigned long long*)&eout[w.e].bad_main, bit);
            if (v.dflags & 1u) eout[w.e].transient = 1u;
        } else {
            if (all_quiet) atomicOr((unsigned long long*)&eout[w.e].silent_mask, bit);   // damper inactive here: only the -80 dB test of Voice::is_silent
            if (!fin) atomicOr((unsigned long long*)&eout[w.e].bad_main, bit);
        }
    }
}

// ------------------------------------------------------------------ chain state helpers (cs[field][I])
#define CSF(f) cs[(size_t)(f) * I + e]

OW_DEV void smoother_load(Smoother& s, const double* __restrict__ cs, int I, int e, int f) {
    s.cur = CSF(f); s.target = CSF(f + 1); s.step = CSF(f + 2); s.rem = (uint32_t)dbits(CSF(f + 3));
}
OW_DEV void smoother_store(const Smoother& s, double* __restrict__ cs, int I, int e, int f) {
    CSF(f) = s.cur; CSF(f + 1) = s.target; CSF(f + 2) = s.step; CSF(f + 3) = bitsd((uint64_t)s.rem);
}
OW_DEV void trem_load(TremState& t, TremPark* __restrict__ P, const double* __restrict__ cs, int I, int e) {
    const int ln = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < 7; ++i) P->v[i][ln] = CSF(CS_T_V + i);
#pragma unroll
    for (int i = 0; i < 4; ++i) { P->ip[i][ln] = CSF(CS_T_I + i); P->ipp[i][ln] = CSF(CS_T_IP + i); }
    t.env = CSF(CS_T_ENV); t.r_ldr = CSF(CS_T_RLDR);
    t.be_fallbacks = 0;
}
OW_DEV void trem_store(const TremState& t, TremPark* __restrict__ P0, double* __restrict__ cs, int I, int e) {
    const int ln = threadIdx.x & 63;
    const TremPark* P = park_opaque(P0);
#pragma unroll
    for (int i = 0; i < 7; ++i) CSF(CS_T_V + i) = P->v[i][ln];
#pragma unroll
    for (int i = 0; i < 4; ++i) { CSF(CS_T_I + i) = P->ip[i][ln]; CSF(CS_T_IP + i) = P->ipp[i][ln]; }
    CSF(CS_T_ENV) = t.env; CSF(CS_T_RLDR) = t.r_ldr;
    if (t.be_fallbacks) CSF(CS_T_BE) = bitsd(dbits(CSF(CS_T_BE)) + (uint64_t)t.be_fallbacks);
}
OW_DEV void dk_load(DkSt& s, const double* __restrict__ cs, int I, int e, int base) {
    s.j_cin = CSF(base); s.cin_prev = CSF(base + 1);
#pragma unroll
    for (int i = 0; i < 8; ++i) s.v[i] = CSF(base + 2 + i);
    s.i_nl[0] = CSF(base + 10); s.i_nl[1] = CSF(base + 11); s.v_nl[0] = CSF(base + 12); s.v_nl[1] = CSF(base + 13);
    dk_refresh_gm(s);
}
OW_DEV void dk_store(const DkSt& s, double* __restrict__ cs, int I, int e, int base) {
    CSF(base) = s.j_cin; CSF(base + 1) = s.cin_prev;
#pragma unroll
    for (int i = 0; i < 8; ++i) CSF(base + 2 + i) = s.v[i];
    CSF(base + 10) = s.i_nl[0]; CSF(base + 11) = s.i_nl[1]; CSF(base + 12) = s.v_nl[0]; CSF(base + 13) = s.v_nl[1];
}

// status blocks of a LIST of engines: cleared before, gathered after the second pass of the voice-sum NaN guard (one launch and one
// transfer however many engines the guard caught, openwurli_hip.hip guard_second_pass)
__global__ void k_eout_clear_list(OwEngineOut* __restrict__ eout, const uint32_t* __restrict__ engs, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    OwEngineOut z;
    z.silent_mask = 0; z.bad_main = 0; z.bad_steal = 0; z.sum_nonfinite = 0; z.out_nonfinite = 0; z.transient = 0; z.pad = 0;
    eout[engs[i]] = z;
}
__global__ void k_eout_gather_list(const OwEngineOut* __restrict__ eout, const uint32_t* __restrict__ engs, int n, OwEngineOut* __restrict__ packed) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) packed[i] = eout[engs[i]];
}

// One bit per engine of [e0, e0 + ne): does the host have to look at the engine's status block after this block?  (a voice of the
// engine fell silent, a steal fade is running, a NaN guard fired, the transient flag is not the one the host knows, a misdispatch.)
// prev_tr mirrors the host's p->transient; the kernel moves it along.  attn[k / 64] = the flags of engines e0 + 64 (k / 64) + 0..63.
__global__ __launch_bounds__(256) void k_eout_attention(const OwEngineOut* __restrict__ eout, const OwEngineArgs* __restrict__ args,
                                                        uint8_t* __restrict__ prev_tr, int e0, int ne, uint64_t* __restrict__ attn) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    bool flag = false;
    if (k < ne) {
        const int e = e0 + k;
        const OwEngineOut o = eout[e];
        const uint8_t tr = o.transient != 0u ? 1 : 0;
        flag = args[e].steal_mask != 0ull || (o.silent_mask & args[e].main_mask) != 0ull || o.bad_main != 0ull || o.bad_steal != 0ull ||
               o.sum_nonfinite != 0u || o.out_nonfinite != 0u || o.transient == 2u || tr != prev_tr[e];
        prev_tr[e] = tr;
    }
    const uint64_t b = __builtin_amdgcn_ballot_w64(flag);
    if ((threadIdx.x & 63) == 0 && k < ne) attn[k >> 6] = b;
}

// ------------------------------------------------------------------ where an engine's CdS resistance R[n] comes from
// Tremolo::process takes no audio and no depth into the oscillator, the LED envelope or r_ldr (tremolo.rs:121-146; depth only enters
// shunt_impedance, :152-167), and new() / reset() start from the same settled state (:83-102, :192-216): r_ldr[t] is ONE deterministic
// sequence per chain rate, t = calls of process() since the cell was built.  The library keeps that sequence in HBM once per (device,
// chain rate) -- the shared trajectory, openwurli_hip.hip `TremTraj` -- and an engine reads it at its own t = pool clock - birth[e].
// Engines that are not on the trajectory (legacy-LFO build, trajectory switched off, older than the store's cap) read the column of
// their phase-group leader in the pool's rbuf instead, as before.
#define OW_OFF_TRAJ (-0x7FFFFFFFFFFFFFFFLL - 1)
struct OwTremSrc {
    const double* rbuf;        // [n_os][I] R of the phase-group leaders for the block being rendered
    const uint32_t* lead;      // [I] leader of every engine
    const double* traj;        // trajectory store AT THE POOL CLOCK (traj[i - birth[e]] = R of chain sample i of this block); null = none
    const long long* birth;    // [I] pool clock at which engine e's cell was built; OW_OFF_TRAJ = the engine reads its phase group
};
struct TremCol { const char* p; uint32_t stride8; };     // R of chain sample i of the block = *(double*)(p + i * stride8): one v_mad_u64_u32
OW_DEV TremCol trem_col(const OwTremSrc& ts, int I, int e) {
    TremCol c;
    const long long b = ts.traj ? ts.birth[e] : OW_OFF_TRAJ;
    if (b != OW_OFF_TRAJ) { c.p = (const char*)(ts.traj - b); c.stride8 = 8u; }
    else { c.p = (const char*)(ts.rbuf + ts.lead[e]); c.stride8 = 8u * (uint32_t)I; }
    return c;
}
OW_DEV double trem_col_at(const TremCol& c, uint32_t i) { return *(const double*)(c.p + (uint64_t)i * c.stride8); }

// ------------------------------------------------------------------ chain init / reset / settle
// Chain state (re)initialisation, lane = engine.
//   mode 1  WurliEngine::new (engine.rs:194-229): fresh chain objects, default smoothers
//   mode 2  set_sample_rate (engine.rs:272-286): fresh chain objects at the new rate, smoothers keep their
//           values and are re-ramped (LinearSmoother::set_ramp_samples, engine.rs:109-116)
//   mode 0  reset (engine.rs:231-251): preamp.reset() at the current R_ldr, tremolo rebuilt, oversampler and
//           speaker state cleared, smoothers snapped to their targets
// The Twin-T oscillator is left at CircuitState DC_OP; k_trem_settle then runs the 50 + 2*sr settle.
// snap (mode 0): [3][I] host-side smoother targets (depth, speaker, volume): LinearSmoother::set_target stores its target when it is
// called (engine.rs:86-99), the device only learns it with the next block, and reset() must snap to the NEW target (engine.rs:245-249).
__global__ __launch_bounds__(64) void k_chain_init(const OwConsts* __restrict__ K, double* __restrict__ cs, const double* __restrict__ snap, int I, int e0, int ne,
                                                   int mode, double depth0) {
    const int e = e0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= e0 + ne) return;
    const double dc[7] = {4.26480458363572357e0, 0.0, 1.24642300965575981e0, 2.75561285973736503e0, 6.66518981651571640e-1, 1.5e1, -2.28408414614134341e-3};
    const double dci[4] = {7.72841164985201955e-5, 3.86420577732601037e-7, 2.20372764986731876e-3, 1.10186382445765932e-5};
    for (int i = 0; i < 7; ++i) CSF(CS_T_V + i) = dc[i];
    for (int i = 0; i < 4; ++i) { CSF(CS_T_I + i) = dci[i]; CSF(CS_T_IP + i) = dci[i]; }
    if (K->tremolo_kind == 1u) CSF(CS_T_V) = 0.0;     // legacy-tremolo build: row 0 is the LFO phase, new() and reset() start it at 0 (tremolo.rs:84,196)
    CSF(CS_T_ENV) = 0.0;
    CSF(CS_T_RLDR) = 1000000.0;
    if (mode == 1) CSF(CS_T_BE) = bitsd(0ull);
    double r_ldr = 1000000.0;
    if (mode != 0) {
        if (mode == 1) {
            Smoother s;
            s.step = 0.0; s.rem = 0u;
            s.cur = s.target = depth0; smoother_store(s, cs, I, e, CS_SM_DEPTH);
            s.cur = s.target = 0.0;    smoother_store(s, cs, I, e, CS_SM_SPK);
            s.cur = s.target = 0.5;    smoother_store(s, cs, I, e, CS_SM_VOL);
            CSF(CS_DIAG) = bitsd(0ull);
        } else {
            for (int f = CS_SM_DEPTH; f <= CS_SM_VOL; f += 4) {
                Smoother s;
                smoother_load(s, cs, I, e, f);
                if (s.rem > 0u) {
                    const uint32_t r = K->ramp_samples;
                    s.step = (s.target - s.cur) / (double)(r > 1u ? r : 1u);
                    s.rem = r;
                }
                smoother_store(s, cs, I, e, f);
            }
        }
        SpeakerSt sp;   // Speaker::new (speaker.rs:63-79)
        sp.character = 1.0; sp.ts = 0.0;
        sp.hpf.s1 = sp.hpf.s2 = sp.lpf.s1 = sp.lpf.s2 = 0.0;
        speaker_update(sp, K->sr);
        const double* hp = &sp.hpf.b0; const double* lp = &sp.lpf.b0;
        for (int i = 0; i < 7; ++i) { CSF(CS_SPK_HPF + i) = hp[i]; CSF(CS_SPK_LPF + i) = lp[i]; }
        CSF(CS_SPK_CHAR) = sp.character; CSF(CS_SPK_A2) = sp.a2; CSF(CS_SPK_A3) = sp.a3; CSF(CS_SPK_TC) = sp.tc; CSF(CS_SPK_TS) = 0.0;
    } else {
        r_ldr = CSF(CS_P_RLDR);
        for (int f = CS_SM_DEPTH, r = 0; f <= CS_SM_VOL; f += 4, ++r) {   // snap_to(target)
            const double t = snap[(size_t)r * I + e];
            CSF(f) = t; CSF(f + 1) = t; CSF(f + 2) = 0.0; CSF(f + 3) = bitsd(0ull);
        }
        CSF(CS_SPK_HPF + 5) = 0.0; CSF(CS_SPK_HPF + 6) = 0.0; CSF(CS_SPK_LPF + 5) = 0.0; CSF(CS_SPK_LPF + 6) = 0.0;
        CSF(CS_SPK_TS) = 0.0;
    }
    DkSt st;
    dk_dc_reset(K, r_ldr, st);
    dk_store(st, cs, I, e, CS_P_MAIN);
    dk_store(st, cs, I, e, CS_P_SHADOW);
    CSF(CS_P_RLDR) = r_ldr; CSF(CS_P_GLDR) = 1.0 / r_ldr; CSF(CS_P_GPREV) = 1.0 / r_ldr;
    for (int i = 0; i < 13; ++i) CSF(CS_OS_UA + i) = 0.0;
    CSF(CS_FLAGS) = bitsd(0ull);
}

// n oscillator steps with the matrices of `K` (Tremolo::new settle loop tremolo.rs:97-100; CircuitState::warmup
// gen_tremolo.rs:2071-2075 when K holds the 48 kHz codegen matrices).
// stagger: leader idx runs idx * stagger additional steps of the WHOLE cell (oscillator + LED + CdS envelope, Tremolo::process) -- the
// test hook that decorrelates the tremolo phases of a pool's groups; the settle proper (Tremolo::new, tremolo.rs:97-100) steps the
// oscillator only.
__global__ __launch_bounds__(64) void k_trem_settle(const OwConsts* __restrict__ K, double* __restrict__ cs, int I, const uint32_t* __restrict__ leaders,
                                                    int n_lead, long long n0, long long stagger = 0) {
    __shared__ TremMats M;
    __shared__ TremPark P;   // oscillator state of this wavefront (LDS-resident, see ow_chain_dev.h)
    trem_mats_load(&M, K, threadIdx.x, blockDim.x);
    __syncthreads();
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_lead) return;
    const int e = (int)leaders[idx];
    const long long n = n0 + stagger * idx;
    TremState t;
    trem_load(t, &P, cs, I, e);
    for (long long i = 0; i < n; ++i) {
        int z = 0;
        asm volatile("" : "+v"(z));        // opaque zero: keeps the LDS reads inside the loop (no hoist into 200 live VGPRs)
        if (stagger) trem_cell_r(t, &P, K, &M + z);
        else trem_osc_step(t, &P, K, &M + z);
    }
    trem_store(t, &P, cs, I, e);
}

// Settled Twin-T state from the process-wide cache (openwurli_hip.hip, trem_settle_cached): rows 0..16 of engine e are overwritten with
// the cached post-settle values, the BE-fallback counter (row 17) advances by the count the settle itself produced.
__global__ void k_trem_load_settled(double* __restrict__ cs, int I, int e, const double* __restrict__ settled18) {
    const int f = threadIdx.x;
    if (f < CS_T_BE) cs[(size_t)f * I + e] = settled18[f];
    else if (f == CS_T_BE) cs[(size_t)f * I + e] = bitsd(dbits(cs[(size_t)f * I + e]) + dbits(settled18[f]));
}

// copy the chain state of engine `src` to engines [e0, e0+ne) (identical by determinism at pool creation)
__global__ void k_chain_replicate(double* __restrict__ cs, int I, int src, int e0, int ne) {
    const int e = e0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= e0 + ne || e == src) return;
    for (int f = 0; f < CS_COUNT; ++f) cs[(size_t)f * I + e] = cs[(size_t)f * I + src];
}

// copy the 18 tremolo rows of engine src[i] to engine dst[i] (a phase group changes its leader / a part of it is split off)
__global__ void k_trem_copy_rows(double* __restrict__ cs, int I, const uint32_t* __restrict__ src, const uint32_t* __restrict__ dst, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || src[i] == dst[i]) return;
    for (int f = 0; f <= CS_T_BE; ++f) cs[(size_t)f * I + dst[i]] = cs[(size_t)f * I + src[i]];
}

__global__ void k_trem_copy_rows_from(double* __restrict__ cs, const double* __restrict__ from, int I, const uint32_t* __restrict__ src,
                                      const uint32_t* __restrict__ dst, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int f = 0; f <= CS_T_BE; ++f) cs[(size_t)f * I + dst[i]] = from[(size_t)f * I + src[i]];
}

// ------------------------------------------------------------------ tremolo stream
// CdS cell resistance for n_os chain-rate samples (no audio input, no depth dependence): runs a block ahead of the audio.
// lane = tremolo phase group (its leader engine); engines of one group share one R stream (column of the leader).
__device__ __forceinline__ void tremolo_body(const OwConsts* __restrict__ K, double* __restrict__ cs, double* __restrict__ rbuf, int I, int n_os,
                                             const uint32_t* __restrict__ leaders, int n_lead, TremMats* M, TremPark* P) {
    trem_mats_load(M, K, threadIdx.x, 64);
    __syncthreads();
    const int idx = blockIdx.x * 64 + threadIdx.x;
    if (idx >= n_lead) return;
    const int e = (int)leaders[idx];
    TremState t;
    trem_load(t, P, cs, I, e);
    for (int i = 0; i < n_os; ++i) {
        int z = 0;
        asm volatile("" : "+v"(z));    // opaque zero: keeps the LDS reads inside the loop
        rbuf[(size_t)i * I + e] = trem_cell_r(t, P, K, M + z);
    }
    trem_store(t, P, cs, I, e);
}
__global__ __launch_bounds__(64) void k_tremolo(const OwConsts* __restrict__ K, double* __restrict__ cs, double* __restrict__ rbuf, int I, int n_os,
                                                const uint32_t* __restrict__ leaders, int n_lead) {
    __shared__ TremMats M;
    __shared__ TremPark P;   // oscillator state of this wavefront (LDS-resident, see ow_chain_dev.h)
    tremolo_body(K, cs, rbuf, I, n_os, leaders, n_lead, &M, &P);
}
// `--features legacy-tremolo` (tremolo.rs:8, 53-57, 80-90, 170-178): the behavioural sine LFO in place of the Twin-T circuit, same CdS
// model behind it.  lane = tremolo phase group; the phase lives in row CS_T_V.  n_extra: leader idx steps idx * n_extra more samples
// (the stagger test hook); rbuf may be null (stepping only).
__global__ __launch_bounds__(64) void k_tremolo_lfo(const OwConsts* __restrict__ K, double* __restrict__ cs, double* __restrict__ rbuf, int I, long long n_os,
                                                    const uint32_t* __restrict__ leaders, int n_lead, long long n_extra) {
    const int idx = blockIdx.x * 64 + threadIdx.x;
    if (idx >= n_lead) return;
    const int e = (int)leaders[idx];
    double phase = CSF(CS_T_V), env = CSF(CS_T_ENV), r_ldr = CSF(CS_T_RLDR);
    const double inc = K->lfo_phase_inc, two_pi = 2.0 * 3.14159265358979323846264338327950288;
    const long long n = n_os + n_extra * idx;
    for (long long i = 0; i < n; ++i) {
        const double lfo = sin(phase);
        phase += inc;
        if (phase >= two_pi) phase -= two_pi;
        const double led = lfo > 0.0 ? lfo : 0.0;                               // f64::max(lfo, 0.0)
        const double coeff = led > env ? K->ldr_attack : K->ldr_release;        // tremolo.rs:126-146, as trem_cell_r
        env = led + coeff * (env - led);
        const double drive = clampd(env, 0.0, 1.0);
        if (drive < 1e-6) r_ldr = 1000000.0;
        else r_ldr = exp(K->ln_r_max + K->ln_min_minus_max * pow(drive, 0.9));
        if (rbuf) rbuf[(size_t)i * I + e] = r_ldr;
    }
    CSF(CS_T_V) = phase; CSF(CS_T_ENV) = env; CSF(CS_T_RLDR) = r_ldr;
}

// ------------------------------------------------------------------ preamp stream
#define OW_PCHUNK 64
__global__ __launch_bounds__(64) void k_preamp(const OwConsts* __restrict__ K, double* __restrict__ cs, const OwEngineArgs* __restrict__ args,
                                               const OwEngineOut* __restrict__ eout, const double* __restrict__ sum, const OwTremSrc tsrc,
                                               double* __restrict__ pre, int I, int L, int Lcap, int e0, int ne) {
    __shared__ double tile[32 * (OW_PCHUNK + 1)];
    const int lane = threadIdx.x;
    const int el = lane & 31, role = lane >> 5;
    const int eb = e0 + blockIdx.x * 32;
    const int e = eb + el;
    const bool valid = e < e0 + ne;
    const int ec = valid ? e : (e0 + ne - 1);   // clamp so every lane runs the same (harmless) work
    const int osr = K->oversample ? 2 : 1;

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
    double sh_depth = __longlong_as_double(0x7FF8000000000000LL), sh_top = 0.0, sh_lower = 0.0;      // trem_shunt's depth-only part (NaN: nothing formed yet)
    // per-lane staging flags of engine row (lane & 31): bit0 = slot pass present, bit1 = steal pass present; 0 when the row is
    // past the range or its block is non-finite (engine.rs:499-501 zeroes it).  Broadcast per row with v_readlane below, so the
    // staging loop has no scalar loads and no branches and the row loads of a group are all in flight together.
    int rowflag = 0;
    {
        const int er = eb + el;
        if (er < e0 + ne && !eout[er].sum_nonfinite) rowflag = (args[er].main_mask ? 1 : 0) | (args[er].steal_mask ? 2 : 0);
    }
    const int e_last = e0 + ne - 1;
    // R[n] is read one host sample ahead (registers), so its global-load latency is hidden behind the previous sample's solve
    const TremCol tcol = trem_col(tsrc, I, ec);   // this engine's place on the shared trajectory, or the column of its phase group
    double rn[2];
    rn[0] = trem_col_at(tcol, 0u);
    rn[1] = osr == 2 ? trem_col_at(tcol, 1u) : 0.0;
    for (int base = 0; base < L; base += OW_PCHUNK) {
        const int cn = min(OW_PCHUNK, L - base);
        // stage 32 engine rows x 64 samples of the voice sum (slot pass + steal pass) through LDS
        const int col = base + min(lane, cn - 1);
#pragma unroll 8
        for (int r = 0; r < 32; ++r) {
            const int er = min(eb + r, e_last);
            const int fl = __builtin_amdgcn_readlane(rowflag, r);
            const double a = sum[((size_t)0 * I + er) * Lcap + col];
            const double b = sum[((size_t)1 * I + er) * Lcap + col];
            double x = (fl & 1) ? a : 0.0;
            x = (fl & 2) ? x + b : x;
            tile[r * (OW_PCHUNK + 1) + lane] = x;
        }
        __syncthreads();
        for (int n = 0; n < cn; ++n) {
            const double x = tile[el * (OW_PCHUNK + 1) + n];
            const double rc[2] = {rn[0], rn[1]};
            {
                const uint32_t nx = (uint32_t)(min(base + n + 1, L - 1) * osr);
                rn[0] = trem_col_at(tcol, nx);
                if (osr == 2) rn[1] = trem_col_at(tcol, nx + 1u);
            }
            const double depth = clampd(sd.next(), 0.0, 1.0);   // engine.rs:533-534, tremolo.rs:117-119
            // Tremolo::shunt_impedance (tremolo.rs:152-167; trem_shunt): the pot's upper leg and r_lower depend on the depth alone, and the
            // depth only moves while its smoother ramps -- formed again (the same operations: the same bits) when some lane of the wavefront
            // sees another depth than the one they were formed from (k_chain_row's form)
            if (__builtin_amdgcn_ballot_w64(!(depth == sh_depth)) != 0ull) {
                sh_depth = depth;
                const double r_upper = 50000.0 * (1.0 - depth);
                sh_lower = 50000.0 * depth;
                sh_top = r_upper > 0.0 ? ow_div(r_upper * 18000.0, r_upper + 18000.0) : 0.0;
            }
            double in[2];
            if (osr == 2) {  // Oversampler::upsample_2x (oversampler.rs:108-121); shadow input is 0.0 (dk_preamp_legacy.rs:599)
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
                const double branch = 680.0 + rc[j];
                const double low = sh_lower > 0.0 ? ow_div(sh_lower * branch, sh_lower + branch) : 0.0;
                const double r_new = fmax(sh_top + low, 1000.0);               // tremolo.rs:152-167; set_ldr_resistance, :620-626
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
        __syncthreads();
    }
    if (valid) {
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

// ------------------------------------------------------------------ output stage
#define OW_OCHUNK 64
// SPLIT (2x oversampled chain): lanes = (engine, oversample phase), 32 engines per wavefront.  The behavioural power amp is
// stateless (power_amp.rs:206), so the two chain-rate samples of one output sample are solved by the two lanes of a pair and
// exchanged with one __shfl_xor; both lanes then run the identical half-band / speaker recurrence (no divergence), the phase-0
// lane owns the state and the output.  Twice the wavefronts of the lane=engine layout, each with ~60 % of the instructions.
// PAIR (round 6, with SPLIT = false): the oversampled chain with lane = engine -- the lane solves the two chain samples one after the
// other and runs the half-band / speaker part ONCE: 0.83 of SPLIT's lane-work per engine, for ranges big enough to fill the chip with
// half the wavefronts (>= 2 per SIMD: the host's choice, `post_pair`).  Same operations per value: the same bits.
template <bool SPLIT, bool PAIR = false>
__global__ __launch_bounds__(64) void k_post(const OwConsts* __restrict__ K, double* __restrict__ cs, const OwEngineArgs* __restrict__ args,
                                             OwEngineOut* __restrict__ eout, const double* __restrict__ pre, float* __restrict__ out, int I, int L,
                                             int Lcap, int e0, int ne, float* __restrict__ out2 = nullptr, size_t ld2 = 0) {
    // out2 / ld2: a second copy of the block, rows at stride ld2 with row 0 = engine e0 -- the caller's pinned host block, mapped into the
    // device's address space (render_range), so that no device-to-host copy trails the kernel
    constexpr int NROWS = SPLIT ? 32 : 64;
    __shared__ float tile[NROWS * (OW_OCHUNK + 1)];
    const int lane = threadIdx.x;
    const int el = SPLIT ? (lane & 31) : lane;
    const int phase = SPLIT ? (lane >> 5) : 0;
    const int eb = e0 + blockIdx.x * NROWS;
    const int e_raw = eb + el;
    const bool valid = e_raw < e0 + ne;
    const int e = valid ? e_raw : (e0 + ne - 1);
    static_assert(!(SPLIT && PAIR), "PAIR is the lane = engine form of the oversampled chain");
    const int osr = (SPLIT || PAIR) ? 2 : 1;
    const double sr = K->sr;
    const double thermal_alpha = K->spk_thermal_alpha;

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
    bool nan_fired = false;

    // the preamp stream is read one host sample ahead (registers): hides the global-load latency behind the previous sample
    double pn = pre[(size_t)phase * I + e];
    double pn1 = PAIR ? pre[(size_t)I + e] : 0.0;
    for (int base = 0; base < L; base += OW_OCHUNK) {
        const int cn = min(OW_OCHUNK, L - base);
        for (int n = 0; n < cn; ++n) {
            const double pc = pn, pc1 = pn1;
            pn = pre[((size_t)min(base + n + 1, L - 1) * osr + phase) * I + e];
            if (PAIR) pn1 = pre[((size_t)min(base + n + 1, L - 1) * 2 + 1) * I + e];
            const double y = power_amp(pc * 0.25);
            double o;
            if (PAIR) {   // engine.rs:536-553, both chain samples in this lane
                const double y1 = power_amp(pc1 * 0.25);
                const double a = allpass3(OW_OS_A0, OW_OS_A1, OW_OS_A2, da, y);
                const double b = allpass3(OW_OS_B0, OW_OS_B1, OW_OS_B2, db, y1);
                o = (a + dd) * 0.5;
                dd = b;
            } else if (SPLIT) {  // engine.rs:536-553
                const double yo = xor32_t(y);
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
            if (phase == 0) tile[el * (OW_OCHUNK + 1) + n] = f;
        }
        __syncthreads();
        for (int r = 0; r < NROWS; ++r) {
            const int er = eb + r;
            if (er < e0 + ne && lane < cn) {
                const float f = tile[r * (OW_OCHUNK + 1) + lane];
                out[(size_t)er * Lcap + base + lane] = f;
                if (out2) out2[(size_t)(er - e0) * ld2 + base + lane] = f;
            }
        }
        __syncthreads();
    }
    if (!valid || phase != 0) return;
    if (nan_fired) {  // preamp.reset()/oversampler.reset() act on post-block state: defer the preamp/up half to k_preamp
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
