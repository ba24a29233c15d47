/* openwurli-hip: TEST AND DEBUG hooks of the library -- NOT part of the drop-in boundary.
 *
 * A facade binds what include/openwurli_hip.h declares; nothing in this header replaces an item of the reference's API.
 * These entry points exist so that tests can reach pieces that have no public surface: the host voice-pool state machine
 * without a device, the note-on MLP's raw outputs, the kernels' own division / exp / tanh routines next to the device
 * library's, and a fault injector for the "never fails, degrades to silence" contract.
 */
#ifndef OPENWURLI_HIP_TEST_H
#define OPENWURLI_HIP_TEST_H

#include "openwurli_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- host-logic test hooks ------------------------------------------------------------------ */
/* A detached engine: the host voice-pool / MIDI state machine (engine.rs:299-374,569-602) without a pool or a device.
 * note_on / note_off / set_sustain / get_diag / slot_* work on it; the slot ops it would send to the GPU can be taken
 * out, and the post-render bookkeeping can be driven with an explicit "silent" mask.  Used by the CPU-only tests. */
ow_engine* ow_test_engine_new(double sample_rate);
void ow_test_engine_free(ow_engine*);
/* Copies up to cap pending ops (type 1 = note-on, 2 = damper, 3 = move-to-steal; seed = fade length for type 3), clears the queue,
 * returns the number that were pending. */
size_t ow_test_engine_take_ops(ow_engine*, uint8_t* type, uint8_t* slot, uint8_t* note, uint32_t* seed, double* velocity, size_t cap);
void ow_test_engine_after_render(ow_engine*, size_t len, uint64_t silent_mask);
uint64_t ow_test_engine_masks(const ow_engine*, int which /* 0 = slot voices, 1 = steal voices */);

/* ---- diagnostics ------------------------------------------------------------------------- */
/* Raw (pre-fade, pre-clamp) outputs of the note-on MLP (mlp_correction.rs:86-116) for n (note, velocity) pairs:
 * out[n][11].  use_mfma = 1 runs the wavefront-batched v_mfma_f64_16x16x4_f64 path that k_apply_ops uses,
 * 0 the scalar lane path (batch jobs).  Returns 0, <0 on device error. */
int ow_debug_mlp_raw(const uint8_t* notes, const double* velocities, size_t n, double* out, int use_mfma, int device);
/* The kernels' division routine (ow_voice_dev.h, ow_div: the compiler's IEEE f64 division sequence without the operand
 * pre-scaling that only extreme exponents need) next to the compiler's own `a / b`, element-wise on the device:
 * fast[i], ieee[i] for n operand pairs.  Returns 0, <0 on device error. */
int ow_debug_div(const double* a, const double* b, size_t n, double* fast, double* ieee, int device);
/* The same for the constant-divisor form (OW_DIV_C): which = 0 jitter draw 2147483647.5, 1 Twin-T V_T, 2 LED span 10.25,
 * 3 preamp V_T 0.026, 4 power-amp 0.013^2, 5 power-amp headroom 22, 6 attack-noise draw 2147483647, 7 pickup soft-limit range 0.04.  a == NULL (which = 0 or 6) runs
 * every numerator of that draw (all 2^31 integers / every int32) on the device and stores the number of quotients that differ from
 * `a / B` in *mismatches. */
int ow_debug_div_const(int which, const double* a, size_t n, double* fast, double* ieee, uint64_t* mismatches, int device);
/* The other short division forms of the kernels next to the compiler's `a / b`, element-wise: mode 0 = divisor with a host-computed
 * reciprocal y[i] = 1.0 / b[i] (ow_div_const: the power amp's per-device constants), 1 = refined reciprocal shared among the quotients over
 * one pivot (ow_rcp_refined + ow_div_y), 2 = the same without the final v_div_fixup (finite operands, pivots checked: the melange column
 * kernel). */
int ow_debug_div_forms(int mode, const double* a, const double* b, const double* y, size_t n, double* fast, double* ieee, int device);
/* Element-wise, the kernels' own elementary functions next to the device library's: which = 0 the preamp's junction exponential
 * (exp without the overflow / underflow selects, for arguments inside the junction clamp) and exp(); 1 the power amp's / speaker's
 * tanh (expm1-based, <= 2 ulp) and tanh(); 2 / 3 / 4 the reed's onset gain (reed.rs:251-264: (0.5 (1 - cos x))^p as exp(p ln .)) at phase x
 * with p = 1.25 / 1.5 / 1.9 and the same expression with the library's pow(). */
int ow_debug_unary(int which, const double* x, size_t n, double* fast, double* lib, int device);

/* ---- tremolo phase groups -------------------------------------------------------------------- */
/* Engines whose tremolo oscillators are bit-identical share one oscillator (a fresh pool is one group).  This hook cuts the pool into
 * n_groups groups {g, g + n_groups, ...} and runs group g ahead by g * (one oscillator period / n_groups) samples: decorrelated
 * tremolo phases, as instances that were reset at different times would have -- what the bench uses to price the per-engine path.
 * Returns 0, <0 on error. */
int ow_test_pool_stagger_tremolo(ow_pool*, size_t n_groups);
/* Number of tremolo phase groups of the pool (1 for a fresh pool, +1 for every engine reset / warmed up on its own). */
size_t ow_test_pool_tremolo_groups(const ow_pool*);

/* The melange power amp ALONE (melange_adapter::PowerAmp::new_at_sample_rate(sample_rate) + process, power_amp.rs:335-431) on n_rows
 * independent input rows of n samples each (f64 [n_rows][n]), one solver per row.  poke_at / poke_node / poke_val (each [n_rows], or all
 * NULL): before sample poke_at[r] of row r, node voltage poke_node[r] of the solver state is overwritten with poke_val[r] (forced
 * divergence).  out: [n_rows][n] normalised amp output; taps (or NULL): [n_rows][n][3] = outer Newton iterations of the sample (70 =
 * exhausted), divergence-guard resets so far, positive rail voltage after the sample.  Returns 0, <0 on error. */
int ow_debug_power_amp(double sample_rate, const double* in, size_t n_rows, size_t n, int rail_sag, const long long* poke_at, const int* poke_node,
                       const double* poke_val, double* out, double* taps, int device);

/* ---- melange power amp: solver taps ------------------------------------------------------------ */
/* Keep the amp's output of every chain-rate sample of the last block (before the half-band down-sampler): f64 [n_engines][n_os] through
 * ow_test_pool_read_power_amp_out.  Costs 16 B per engine and chain-rate sample of the block capacity; off until enabled. */
int ow_test_pool_enable_power_amp_tap(ow_pool*);
/* Melange power amp: Newton passes (main sweep) engine k spent on its LAST rendered block -- the figure the demand-ordered dispatch of
 * k_post_mpa sorts by (0 = not rendered yet).  out: one uint32 per engine of the pool; n must equal the pool size.  -1 without the amp. */
int ow_test_pool_power_amp_passes(ow_pool*, uint32_t* out, size_t n);
int ow_test_pool_read_power_amp_out(ow_pool*, double* out_host, size_t out_stride, size_t n_os);
/* Overwrite one node voltage of the amp's solver state (v_prev[node], node < 20) before the next block: the way to force the
 * divergence guard (power_amp.rs:410-421: |node| > 100 V -> reset + hold last good) at a known sample. */
int ow_test_engine_poke_power_amp_node(ow_engine*, int node, double volts);

/* ---- host constant builders ------------------------------------------------------------------ */
/* The library's own set_sample_rate / rebuild_matrices of the three generated solvers (ow_consts_host.hpp; gen_tremolo.rs:2111-2342,
 * gen_preamp.rs:1930-2219, gen_power_amp.rs:8588-8831) at chain rate `rate`, on the host, no device needed.  solver: 0 Twin-T tremolo
 * (N 7, M 4), 1 melange preamp at the nominal pot (N 12, M 3), 2 melange power amp (N 20, M 16).  Outputs are row-major s[N][N] k[M][M]
 * sni[N][M] aneg[N][N] and the backward-Euler set (any pointer may be NULL).  force_rebuild != 0 rebuilds even at the solver's codegen
 * rate, where the reference copies its baked tables -- the tests compare that rebuild with the baked tables.  Returns N * 100 + M. */
int ow_test_host_matrices(int solver, double rate, int force_rebuild, double* s, double* k, double* sni, double* aneg,
                          double* s_be, double* k_be, double* sni_be, double* aneg_be);

/* ---- settled-state caches --------------------------------------------------------------------- */
/* The library keeps the settled Twin-T oscillator state per (device, chain rate) and the settled melange preamp / power-amp states per
 * device for the life of the process (the reference's OnceLock caches: melange_adapter.rs:12-29, power_amp.rs:283-299; the Twin-T one is
 * this library's own, tremolo.rs:92-102 pays 2 s of solver steps per Tremolo::new).  This drops them, so that the next pool runs the
 * settle kernels again -- the test compares a cached engine with a freshly settled one bit for bit.  Returns the number of cached
 * Twin-T states dropped. */
int ow_test_clear_settle_caches(void);

/* ---- the trajectory's oscillator kernels on their own --------------------------------------------- */
/* CircuitState at DC_OP, n_settle oscillator steps without the cell, then n steps of the shared trajectory (r_ldr[0 .. n)) in launches of
 * `chunk` steps, with the quad-lane kernels (row = 0: ow_trem_wide.h) or the one-system-per-wavefront kernels (row = 1: ow_trem_row.h).
 * No store and no pool are touched.  r_out[n]; state_out[18] = the oscillator rows after the last step; ckpt_out[(n / 4096 + 2) * 16] and
 * be_out[1024] may be NULL; *ms_out = device time of the trajectory launches; cold_out[2] (or NULL, row kernels only) = Newton sweeps the row
 * step handed to the generic sweep / steps that took the backward-Euler retry during those launches; kick18 (or NULL): added to the
 * DC_OP state rows (v[7], i_prev[4], i_pp[4]) before anything runs -- a circuit pushed off its operating point.  The two must agree bit for bit
 * (tests/test_gpu_trajectory.py); the time per step is the figure every small pool waits for.  Returns 0, <0 on error. */
int ow_debug_trem_trajectory(double sample_rate, long long n_settle, long long n, long long chunk, int row, double* r_out, double* state_out,
                             double* ckpt_out, unsigned long long* be_out, double* ms_out, int device, unsigned long long* cold_out, const double* kick18);

/* ---- voice-sum NaN guard ----------------------------------------------------------------------- */
/* Overwrite one double of a voice record on the device before the next block (slot 0..63; steal != 0 selects the slot's steal voice;
 * field = a VF_* index of openwurli_amd/csrc/ow_types.h: OW_TEST_VF_Q is the pickup charge, OW_TEST_VF_S0 mode 0's sine state).  A
 * voice cannot turn non-finite through the API, so this is how the guard of engine.rs:496-521 (zero the block, render every voice
 * again, free the culprits; the survivors advance twice) is provoked.  Returns 0, <0 on error. */
#define OW_TEST_VF_S0 0
#define OW_TEST_VF_Q 81
int ow_test_engine_poke_voice(ow_engine*, int slot, int steal, int field, double value);

/* The same for the legacy preamp: overwrite node voltage `node` (0..7) of engine e's main (shadow = 0) or shadow solver state before the next
 * block.  A non-finite value provokes the preamp's own NaN reset (dk_preamp_legacy.rs:610-615).  Returns 0, <0 on error. */
int ow_test_engine_poke_preamp_node(ow_engine*, int shadow, int node, double volts);
/* ... and read that state back after the blocks rendered so far: out14 = j_cin, cin_rhs_prev, v[8], i_nl[2], v_nl[2] (DkState,
 * dk_preamp_legacy.rs:231-239).  Returns 0, <0 on error. */
int ow_test_engine_read_preamp_state(ow_engine*, int shadow, double* out14);

/* Plain device-to-host copy, for reading a block that ow_pool_render(pool, NULL, ...) left in HBM (ow_pool_device_output). */
int ow_test_device_read(void* dst_host, const void* src_device, size_t bytes, int device);

/* Fast paths of the melange preamp's literal rebuild the host found usable at chain rate `rate` (host only): bit 0 = leading block
 * replayed once per rate, bit 1 = the LU factors have the compiled-in sparsity pattern of the column-streamed kernel.  <0 on error. */
int ow_test_host_melange_paths(double rate);

/* ---- latched switches ------------------------------------------------------------------------- */
/* The OW_* environment switches ("Environment switches") are read once, when a pool is created, and kept in the pool: the
 * render path never calls getenv.  These change / read one on a live pool: "trem_serial", "trem_wide", "preamp_wide" (-1 = by size),
 * "mel_generic", "mel_rank1", "mel_lds", "pa_sort" (0..2), "host_profile"; get also answers "trem_traj" and "trem_cache".  set returns
 * 0, <0 for an unknown or creation-only switch; get returns the value, -2 for an unknown name. */
int ow_test_pool_set_switch(ow_pool*, const char* name, int value);
int ow_test_pool_get_switch(const ow_pool*, const char* name);
/* out[0] = engines of the pool reading the shared tremolo trajectory, out[1] = samples its store holds (produced or enqueued),
 * out[2] = the store's capacity in samples. */
int ow_test_pool_trajectory_info(const ow_pool*, uint64_t out[3]);
/* The store behind the pool, in samples: [0] produced or enqueued, [1] known complete, [2] held by its buffers now, [3] configured capacity,
 * [4] t of the pool's oldest engine on it. */
int ow_test_pool_trajectory_state(const ow_pool*, uint64_t out[5]);

/* ---- fault injection ------------------------------------------------------------------------ */
/* The next n_renders calls of ow_pool_render / ow_engine_render on this pool fail before their first launch, exactly as a HIP
 * error would (exception inside the guarded region): the caller's block must come back as silence in every row, ow_last_error
 * must name the failure, and the renders after them must work again. */
void ow_test_inject_render_faults(ow_pool*, int n_renders);

#ifdef __cplusplus
}
#endif
#endif /* OPENWURLI_HIP_TEST_H */
