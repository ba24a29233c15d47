// openwurli-hip: melange 12-node preamp with the reference's LITERAL per-sample matrix rebuild (closes DESIGN deviation 6).
//
// gen_preamp.rs:1990-2062 rebuild_matrices + :2117-2219 invert_n: whenever R_ldr moved, the reference re-forms A = G_eff + alpha C,
// factors it (LU, partial pivoting), solves the twelve unit columns for S = A^-1, and derives S N_i and K = N_v S N_i.  The
// rank-one kernel (ow_melange_dev.h) is the same mathematics with a shorter operation sequence; while R_ldr moves the audible
// signal contains the DIFFERENCE of successive inverses, and there the LU's rounding noise (eps * cond(A)) is part of the
// reference's output.  This kernel performs the reference's operations in the reference's order, so that noise is reproduced.
//
// Mapping: lane = (engine, main | shadow), 32 engines per wavefront, as k_preamp_mel.  Both states of an engine see the same
// R_ldr (melange_adapter.rs:82-85), hence the same matrices: ONE factorisation per engine, built by its two lanes together --
// each lane eliminates every other row of a column step and solves six of the twelve unit columns.  LU and S live in LDS,
// engine-minor ([12][12][32] doubles each, 72 KB together: two wavefronts per CU); the row exchanges of the pivoting are a per-engine
// permutation (twelve nibbles in a register), so no row is ever moved.  K (3x3) is kept in registers; S N_i is formed where it is
// used, from S, with the reference's sum order (its structurally zero terms add +-0 and are skipped).
//
// (A state whose solver was NaN-reset on its own keeps, like in the reference, a pot value that can differ from its partner's by
// less than the 1e-12 hysteresis of set_runtime_R: the shared matrices then follow the main state's value.)
#pragma once
#include "ow_melange_dev.h"

namespace owdev {

#define OW_LCHUNK 8     // voice-sum staging tile: 32 rows x 8 samples, so that LDS stays under 40 KB (four workgroups = one wavefront per SIMD per CU)
#define MLU(r, c) lu[((r) * 12 + (c)) * 32]
#define MS(r, c) S[((r) * 12 + (c)) * 32]
#define PERM(p, r) ((int)(((p) >> (4 * (r))) & 15ull))

// rebuild_matrices for this lane's engine at resistance `pot`; lu / S point at the engine's column.  Both lanes of the pair call it
// with identical arguments (role = 0 / 1); kk receives K.  Contains workgroup barriers: every lane of the wavefront must call it.
__device__ inline void mel_lit_kernel(const double* __restrict__ S, double kk[3][3]);
__device__ __noinline__ void mel_lit_rebuild(double pot, int role, double alpha, double* __restrict__ lu, double* __restrict__ S, double kk[3][3]) {
    // A = G_eff + alpha C (gen_preamp.rs:2001-2016); g_eff[6][6] += 1/R - G_nom
    const double g66 = PRE_G[6][6] + (ow_div(1.0, pot) - PRE_POT_0_G_NOM);
    for (int i = role; i < 12; i += 2)
        for (int j = 0; j < 12; ++j) MLU(i, j) = ((i == 6 && j == 6) ? g66 : PRE_G[i][j]) + alpha * PRE_C[i][j];
    __syncthreads();
    unsigned long long perm = 0xBA9876543210ull;
    bool singular = false;
    for (int k = 0; k < 12; ++k) {
        int max_row = k;
        double max_val = fabs(MLU(PERM(perm, k), k));
        for (int i = k + 1; i < 12; ++i) {
            const double v = fabs(MLU(PERM(perm, i), k));
            if (v > max_val) { max_val = v; max_row = i; }
        }
        if (max_val < 1e-30) singular = true;            // identity fallback below (:2142-2148); keep going so that barriers stay uniform
        if (max_row != k) {
            const unsigned long long pk = (perm >> (4 * k)) & 15ull, pm = (perm >> (4 * max_row)) & 15ull;
            perm = (perm & ~(15ull << (4 * k)) & ~(15ull << (4 * max_row))) | (pm << (4 * k)) | (pk << (4 * max_row));
        }
        const int pr = PERM(perm, k);
        const double pivot = MLU(pr, k);
        for (int i = k + 1 + role; i < 12; i += 2) {      // the pair shares the rows below the pivot
            const int ri = PERM(perm, i);
            const double m = ow_div(MLU(ri, k), pivot);
            MLU(ri, k) = m;
            for (int j = k + 1; j < 12; ++j) MLU(ri, j) -= m * MLU(pr, j);
        }
        __syncthreads();
    }
    // twelve unit columns: forward substitution from the permuted unit row, back substitution (:2184-2215); six columns per lane
    for (int col = role; col < 12; col += 2) {
        double b[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) b[i] = (PERM(perm, i) == col) ? 1.0 : 0.0;
        // rows above the unit row only subtract products with b == 0 (+-0): skipping them, as the reference does, leaves the same values
#pragma unroll
        for (int i = 1; i < 12; ++i) {
            const int ri = PERM(perm, i);
            double sum = b[i];
#pragma unroll
            for (int j = 0; j < i; ++j) sum -= MLU(ri, j) * b[j];
            b[i] = sum;
        }
#pragma unroll
        for (int i = 11; i >= 0; --i) {
            const int ri = PERM(perm, i);
            double sum = b[i];
#pragma unroll
            for (int j = i + 1; j < 12; ++j) sum -= MLU(ri, j) * b[j];
            const double pivot = MLU(ri, i);
            if (fabs(pivot) < 1e-30) singular = true;
            b[i] = ow_div(sum, pivot);
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) MS(i, col) = b[i];
    }
    __syncthreads();
    if (__builtin_expect(singular, 0)) {                   // invert_n returns the identity when the matrix is singular
        for (int i = role; i < 12; i += 2)
            for (int j = 0; j < 12; ++j) MS(i, j) = (i == j) ? 1.0 : 0.0;
    }
    __syncthreads();
    mel_lit_kernel(S, kk);
}

// K = N_v (S N_i) (:2038-2056).  N_i rows: [0] = {2}, [1] = {2, 4, 5}, [2] = {4, 7, 8}; N_v rows: [0] = {2}, [1] = {2, 5}, [2] = {4, 8}.
// s_ni[n][j] = sum_k s[n][k] N_I[j][k] in k order, then K[i][j] = sum_n N_V[i][n] s_ni[n][j] in n order: zero terms skipped (they add +-0).
#define SNI0(n) (MS(n, 2) * PRE_N_I[0][2])
#define SNI1(n) (MS(n, 2) * PRE_N_I[1][2] + MS(n, 4) * PRE_N_I[1][4] + MS(n, 5) * PRE_N_I[1][5])
#define SNI2(n) (MS(n, 4) * PRE_N_I[2][4] + MS(n, 7) * PRE_N_I[2][7] + MS(n, 8) * PRE_N_I[2][8])
__device__ inline void mel_lit_kernel(const double* __restrict__ S, double kk[3][3]) {
    kk[0][0] = PRE_N_V[0][2] * SNI0(2); kk[0][1] = PRE_N_V[0][2] * SNI1(2); kk[0][2] = PRE_N_V[0][2] * SNI2(2);
    kk[1][0] = PRE_N_V[1][2] * SNI0(2) + PRE_N_V[1][5] * SNI0(5);
    kk[1][1] = PRE_N_V[1][2] * SNI1(2) + PRE_N_V[1][5] * SNI1(5);
    kk[1][2] = PRE_N_V[1][2] * SNI2(2) + PRE_N_V[1][5] * SNI2(5);
    kk[2][0] = PRE_N_V[2][4] * SNI0(4) + PRE_N_V[2][8] * SNI0(8);
    kk[2][1] = PRE_N_V[2][4] * SNI1(4) + PRE_N_V[2][8] * SNI1(8);
    kk[2][2] = PRE_N_V[2][4] * SNI2(4) + PRE_N_V[2][8] * SNI2(8);
}

// The same rebuild when the pivot order is the expected one (it is, for every R in the clamp range at every rate tried: the only row
// exchange happens in step 3, before R plays any part).  The host has replayed elimination steps 0..5 and the forward substitutions
// through them (OwConsts::ml_*): what depends on R is the Schur-complement entry of [6][6], the trailing 6x6 factorisation and the
// substitutions through it -- all with compile-time indices, in registers, no LDS besides the result.  Every pivot choice of the
// trailing steps is CHECKED against what invert_n would choose (first maximum of the column, strict >): on any difference, or a
// pivot below 1e-30, the function returns false and the caller runs the generic rebuild above.  Same operations in the same order
// as invert_n on the R-dependent entries; the R-independent ones are the host's (IEEE, no contraction) results of the same operations.
// Both lanes of a pair compute the trailing factors (cheap, no exchange needed) and six unit columns each.
__device__ inline bool mel_lit_rebuild_fast(const OwConsts* __restrict__ K, double pot, int role, double alpha, double* __restrict__ S) {
    double T[6][6];
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = 0; b < 6; ++b) T[a][b] = K->ml_t0[a][b];
    {
        const double g66 = PRE_G[6][6] + (ow_div(1.0, pot) - PRE_POT_0_G_NOM);
        double e = g66 + alpha * PRE_C[6][6];
#pragma unroll
        for (int k = 0; k < 6; ++k) e -= K->ml_chain_m[k] * K->ml_chain_u[k];
        const int t6 = K->ml_t6;
#pragma unroll
        for (int a = 0; a < 6; ++a) if (a == t6) T[a][0] = e;
    }
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 6; ++k) {                           // steps 6..11 of the elimination, no row exchange expected
        const double pk = fabs(T[k][k]);
#pragma unroll
        for (int i = k + 1; i < 6; ++i) ok = ok && !(fabs(T[i][k]) > pk);
        ok = ok && !(pk < 1e-30);
        const double pivot = T[k][k];
#pragma unroll
        for (int i = k + 1; i < 6; ++i) {
            const double m = ow_div(T[i][k], pivot);
            T[i][k] = m;
#pragma unroll
            for (int j = k + 1; j < 6; ++j) T[i][j] -= m * T[k][j];
        }
    }
    if (!__all(ok)) return false;                           // wave-uniform: the generic path has barriers
#pragma unroll 1
    for (int col = role; col < 12; col += 2) {
        double b[12];
#pragma unroll
        for (int t = 0; t < 6; ++t) {                       // forward substitution, trailing rows (terms j < 6 are in ml_part)
            double sum = K->ml_part[col][t];
#pragma unroll
            for (int j = 0; j < t; ++j) sum -= T[t][j] * b[6 + j];
            b[6 + t] = sum;
        }
#pragma unroll
        for (int t = 5; t >= 0; --t) {                      // back substitution, trailing rows
            double sum = b[6 + t];
#pragma unroll
            for (int j = t + 1; j < 6; ++j) sum -= T[t][j] * b[6 + j];
            b[6 + t] = ow_div(sum, T[t][t]);
        }
#pragma unroll
        for (int i = 5; i >= 0; --i) {                      // back substitution, rows 5..0 (R-independent U rows)
            double sum = K->ml_btop[col][i];
#pragma unroll
            for (int j = i + 1; j < 12; ++j) sum -= K->ml_utop[i][j] * b[j];
            b[i] = ow_div(sum, K->ml_utop[i][i]);
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) MS(i, col) = b[i];
    }
    return true;
}

// gen_preamp::process_sample (gen_preamp.rs:3399-3663) with the engine's own S (LDS) and K.  an_c: alpha C - G of the pool (constant
// except [6][6]); an66 = alpha C[6][6] - g_eff[6][6] of the current R.
__device__ inline double mel_process_lit(MelSt& st, double input_in, const double (*__restrict__ an)[12], double an66, const double* __restrict__ S,
                                         const double kk[3][3], const double* nz, int nz_stride) {
    const double input = isfinite(input_in) ? clampd(input_in, -100.0, 100.0) : 0.0;
#pragma unroll
    for (int i = 0; i < 12; ++i) st.v[i] = st.v[i] + 1e-25 - 1e-25;
#pragma unroll
    for (int i = 0; i < 3; ++i) st.ip[i] = st.ip[i] + 1e-25 - 1e-25;
    const bool force_be = st.be_cooldown > 0u;
    if (st.be_cooldown > 0u) st.be_cooldown -= 1u;
    const double* v = st.v;
#define AN(i, j) an[i][j]
    double rhs[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 15.0};     // RHS_CONST (gen_preamp.rs:760-773); build_rhs :3041-3095
    rhs[0] += AN(0, 0) * v[0] + AN(0, 1) * v[1];
    rhs[1] += AN(1, 0) * v[0] + AN(1, 1) * v[1] + AN(1, 2) * v[2];
    rhs[2] += AN(2, 1) * v[1] + AN(2, 2) * v[2] + AN(2, 3) * v[3] + AN(2, 4) * v[4] + AN(2, 5) * v[5];
    rhs[3] += AN(3, 2) * v[2] + AN(3, 3) * v[3] + AN(3, 4) * v[4] + AN(3, 7) * v[7] + AN(3, 11) * v[11];
    rhs[4] += AN(4, 2) * v[2] + AN(4, 3) * v[3] + AN(4, 4) * v[4] + AN(4, 7) * v[7] + AN(4, 8) * v[8];
    rhs[5] += AN(5, 2) * v[2] + AN(5, 5) * v[5] + AN(5, 6) * v[6];
    rhs[6] += AN(6, 5) * v[5] + an66 * v[6] + AN(6, 10) * v[10];
    rhs[7] += AN(7, 3) * v[3] + AN(7, 4) * v[4] + AN(7, 7) * v[7] + AN(7, 10) * v[10];
    rhs[8] += AN(8, 4) * v[4] + AN(8, 8) * v[8] + AN(8, 9) * v[9];
    rhs[9] += AN(9, 8) * v[8] + AN(9, 9) * v[9];
    rhs[10] += AN(10, 6) * v[6] + AN(10, 7) * v[7] + AN(10, 10) * v[10];
#undef AN
    rhs[2] += PRE_N_I[0][2] * st.ip[0];
    rhs[2] += PRE_N_I[1][2] * st.ip[1];
    rhs[4] += PRE_N_I[1][4] * st.ip[1];
    rhs[4] += PRE_N_I[2][4] * st.ip[2];
    rhs[5] += PRE_N_I[1][5] * st.ip[1];
    rhs[7] += PRE_N_I[2][7] * st.ip[2];
    rhs[8] += PRE_N_I[2][8] * st.ip[2];
    rhs[0] += (input + st.input_prev) / PRE_INPUT_RESISTANCE;
    if (nz) nz_stamp(rhs, nz, nz_stride);
    double v_pred[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) {                                  // v_pred = S rhs, rows in j order
        double sum = 0.0;
#pragma unroll
        for (int j = 0; j < 12; ++j) sum += MS(i, j) * rhs[j];
        v_pred[i] = sum;
    }
    const double p[3] = {-v_pred[2], v_pred[2] - v_pred[5], v_pred[4] - v_pred[8]};
    double i_nl[3];
    uint32_t last_it = mel_solve_nl(p, kk, st.ip, st.ipp, i_nl);
    // (fence: the S entries below were all read for S rhs above; without it the compiler keeps those 60 doubles in registers across
    // the Newton solve -- and spills -- instead of reading LDS again)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double vn[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) {                                  // v = v_pred + (S N_i) i_nl
        double x = v_pred[i];
        x += SNI0(i) * i_nl[0];
        x += SNI1(i) * i_nl[1];
        x += SNI2(i) * i_nl[2];
        vn[i] = x;
    }
    const bool nr_failed = last_it >= 265u;
    bool ringing = false;
#pragma unroll
    for (int i = 0; i < 11; ++i) ringing = ringing || (fabs(vn[i]) > 55.0);
    if (__builtin_expect(nr_failed || ringing || force_be, 0)) {
        if (ringing || nr_failed) st.be_cooldown = 64u;
        st.be_fallbacks += 1u;
        MelSt tmp = st;
        double vn2[12], inl2[3];
        last_it = mel_be_fallback(tmp, input, vn2, inl2, nz, nz_stride);
#pragma unroll
        for (int i = 0; i < 12; ++i) vn[i] = vn2[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) i_nl[i] = inl2[i];
    }
    {   // voltage-damp net (gen_preamp.rs:3576-3613)
        double max_delta = 0.0;
#pragma unroll
        for (int i = 0; i < 11; ++i) { const double d = fabs(vn[i] - st.v[i]); if (d > max_delta) max_delta = d; }
        double max_dc = 0.0;
#pragma unroll
        for (int i = 0; i < 11; ++i) { const double a = fabs(PRE_DC_OP[i]); if (a > max_dc) max_dc = a; }
        const double thr = fma(max_dc, 0.05, 2.0);
        if (max_delta > thr) {
            const double damp = fmax(ow_div(thr, max_delta), 0.01);
#pragma unroll
            for (int i = 0; i < 12; ++i) vn[i] = st.v[i] + damp * (vn[i] - st.v[i]);
#pragma unroll
            for (int i = 0; i < 3; ++i) i_nl[i] = st.ip[i] + damp * (i_nl[i] - st.ip[i]);
        }
    }
    bool finite = true;
#pragma unroll
    for (int i = 0; i < 12; ++i) finite = finite && isfinite(vn[i]);
    if (!finite) {
#pragma unroll
        for (int i = 0; i < 12; ++i) st.v[i] = PRE_DC_OP[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) { st.ip[i] = PRE_DC_NL_I[i]; st.ipp[i] = PRE_DC_NL_I[i]; }
        st.input_prev = 0.0;
        st.pot = 9.99999999999999854e4;
        st.be_cooldown = 0u;
        st.nan_resets += 1u;
        return clampd(PRE_DC_OP[10] * 1.0, -10.0, 10.0);
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) st.v[i] = vn[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) { st.ipp[i] = st.ip[i]; st.ip[i] = i_nl[i]; }
    st.input_prev = input;
    const double raw = isfinite(vn[10]) ? vn[10] : 0.0;
    return raw * 1.0;
}

// Preamp stream, literal rebuild.  Same interface as k_preamp_mel.
__global__ __launch_bounds__(64) void k_preamp_mel_lit(const OwConsts* __restrict__ K, double* __restrict__ cs,
                                                       const double* __restrict__ settled, const OwEngineArgs* __restrict__ args,
                                                       const OwEngineOut* __restrict__ eout, const double* __restrict__ sum, const OwTremSrc tsrc,
                                                       double* __restrict__ pre, double* __restrict__ noise, int I, int L,
                                                       int Lcap, int e0, int ne, int generic_only, double* __restrict__ lu_scratch) {
    __shared__ double tile[32 * (OW_LCHUNK + 1)];
    __shared__ double S_all[12 * 12 * 32];
    const int lane = threadIdx.x;
    const int el = lane & 31, role = lane >> 5;
    const int eb = e0 + blockIdx.x * 32;
    const int e = eb + el;
    const bool valid = e < e0 + ne;
    const int ec = valid ? e : (e0 + ne - 1);
    const int osr = K->oversample ? 2 : 1;
    const TremCol rc = trem_col(tsrc, I, ec);
    const double alpha = 2.0 * (K->os_sr * 1.0);                    // gen_preamp.rs:1991-1992
    // The LU workspace is only touched by the generic rebuild (the fallback of the fast path): it lives in HBM, one [12][12][32] slab
    // per workgroup (slabs of disjoint engine ranges are disjoint: ceil(e0/32) + block), so that LDS holds S alone -- 39 KB, four
    // workgroups per CU, every SIMD busy (with LU in LDS: two workgroups, half the SIMDs idle; 52.8 -> see DESIGN).
    double* lu = lu_scratch + ((size_t)(e0 / 32) + blockIdx.x) * (12 * 12 * 32) + el;     // stage starts are multiples of 32 engines (build_voice_lists)
    double* S = S_all + el;

    MelSt st;
    double ua[3], ub[3];
    Smoother sd;
    {
        const int e = ec;
        smoother_load(sd, cs, I, e, CS_SM_DEPTH);
        if (args[e].set_flags & 1u) sd.retarget(args[e].depth_target, K->ramp_samples);
        mel_load(st, cs, I, e, role ? CS_M_SHADOW : CS_M_MAIN);
        for (int i = 0; i < 3; ++i) { ua[i] = CSF(CS_OS_UA + i); ub[i] = CSF(CS_OS_UB + i); }
        const uint64_t fl = dbits(CSF(CS_FLAGS));
        if (fl & 1ull) {
            mel_init_state(st, settled);
            for (int i = 0; i < 3; ++i) { ua[i] = 0.0; ub[i] = 0.0; }
        }
    }
    uint32_t adapter_resets = 0;
    const bool nz_mine = noise != nullptr && valid && role == 0;
    const bool nz_on = nz_mine && args[ec].noise_on != 0u;
    const double scale_half = K->m_noise_scale * 1.0 * args[ec].thermal_gain * 0.5;
    double* nzcol = noise ? noise + ec : nullptr;                   // the noise column stays in HBM here (LDS holds the matrices)
    if (nz_mine && (dbits(CSF(CS_FLAGS)) & 1ull)) nz_reseed(nzcol, I);
    // The matrices a state works with are those of the R its last rebuild saw.  They are a pure function of (rate, R), so they need not
    // survive the block: the first sample rebuilds them (s_pot = NaN never equals a resistance).  Exception, as in the reference: a state
    // that has never seen a set_runtime_R (pot at the nominal literal) would run on the baked / set_sample_rate matrices -- every sample
    // of the engine path sets R before it processes, so that case does not occur here.
    const bool force_generic = generic_only != 0;                  // OW_MEL_GENERIC=1: always the generic rebuild (the test compares the two bit for bit)
    double s_pot = __longlong_as_double(0x7ff8000000000000LL);
    double kk[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    double an66 = 0.0;
    for (int base = 0; base < L; base += OW_LCHUNK) {
        const int cn = min(OW_LCHUNK, L - base);
        for (int r = 0; r < 32; ++r) {
            const int er = eb + r;
            double x = 0.0;
            if (er < e0 + ne && lane < cn && !eout[er].sum_nonfinite) {
                if (args[er].main_mask) x = sum[((size_t)0 * I + er) * Lcap + base + lane];
                if (args[er].steal_mask) x += sum[((size_t)1 * I + er) * Lcap + base + lane];
            }
            if (lane < OW_LCHUNK) tile[r * (OW_LCHUNK + 1) + lane] = x;
        }
        __syncthreads();
        for (int n = 0; n < cn; ++n) {
            const double x = tile[el * (OW_LCHUNK + 1) + n];
            const double depth = clampd(sd.next(), 0.0, 1.0);
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
                const size_t s_idx = (size_t)((base + n) * osr + j);
                mel_set_r(st, trem_shunt(depth, trem_col_at(rc, (uint32_t)s_idx)));
                // lazy rebuild (gen_preamp.rs:3408-3411), once per engine, keyed on the main state's resistance
                const double pot_main = __shfl(st.pot, el);
                const bool dirty = !(pot_main == s_pot);
                if (__any(dirty)) {
                    // every lane takes part (the rebuild contains barriers); for an engine whose R did not move this recomputes the
                    // matrices it already has -- they are a pure function of R
                    bool fast = K->ml_ok != 0 && !force_generic;
                    if (fast) {
                        __syncthreads();                                          // the previous sample's reads of S are done
                        fast = mel_lit_rebuild_fast(K, pot_main, role, alpha, S);
                        __syncthreads();                                          // both lanes' columns are in place
                        if (fast) mel_lit_kernel(S, kk);
                    }
                    if (!fast) mel_lit_rebuild(pot_main, role, alpha, lu, S, kk);
                    s_pot = pot_main;
                    const double g66 = PRE_G[6][6] + (ow_div(1.0, pot_main) - PRE_POT_0_G_NOM);
                    an66 = alpha * PRE_C[6][6] - g66;
                }
                const double* nzp = nullptr;
                if (nz_on && scale_half != 0.0) {
                    const double sir10 = st.pot == 9.99999999999999854e4 ? PRE_NOISE_THERMAL_SQRT_INV_R_DEFAULT[10] : sqrt(1.0 / st.pot);
                    nz_draw(nzcol, I, scale_half, sir10);
                    nzp = nzcol;
                }
                const uint32_t nan_before = st.nan_resets;
                const double o = mel_process_lit(st, in[j], K->m_aneg0, an66, S, kk, nzp, I);
                if (nz_mine && st.nan_resets != nan_before) nz_clear_lag(nzcol, I);
                const double other = __shfl_xor(o, 32);
                double result = role ? (other - o) : (o - other);
                if (!isfinite(result)) {
                    mel_init_state(st, settled);
                    if (nz_mine) nz_reseed(nzcol, I);
                    result = 0.0;
                    adapter_resets += 1u;
                }
                if (valid && role == 0) pre[s_idx * I + e] = result;
            }
        }
        __syncthreads();
    }
    if (valid) {
        mel_store(st, cs, I, e, role ? CS_M_SHADOW : CS_M_MAIN);
        if (role == 0) {
            for (int i = 0; i < 3; ++i) { CSF(CS_OS_UA + i) = ua[i]; CSF(CS_OS_UB + i) = ub[i]; }
            smoother_store(sd, cs, I, e, CS_SM_DEPTH);
            const uint64_t fl = dbits(CSF(CS_FLAGS));
            if (fl & 1ull) CSF(CS_FLAGS) = bitsd(fl & ~1ull);
            const uint32_t nr = adapter_resets + st.nan_resets;
            if (nr) {
                const uint64_t d = dbits(CSF(CS_DIAG));
                CSF(CS_DIAG) = bitsd((d & 0xFFFFFFFFull) | ((uint64_t)((uint32_t)(d >> 32) + nr) << 32));
            }
        }
    }
}

}  // namespace owdev
