// openwurli-hip: melange 12-node preamp, literal per-sample rebuild, COLUMN-STREAMED (round 3; replaces the LDS-resident S of
// ow_melange_lit.h for every pool size).
//
// What the reference does per chain-rate sample whose R_ldr moved (gen_preamp.rs:1990-2062, 2117-2219, 3399-3663): factor
// A = G_eff + alpha C (LU, partial pivoting), solve the twelve unit columns for S = A^-1, form S N_i and K = N_v S N_i, then
// v_pred = S rhs, Newton on the three junctions, v = v_pred + (S N_i) i_nl -- in both solver states (main, shadow).
//
// S itself is never needed as a matrix.  Every consumer sums over S's COLUMNS in ascending order:
//     v_pred[i] = ((S[i][0] rhs[0] + S[i][1] rhs[1]) + ...) + S[i][11] rhs[11]          (gen_preamp.rs:3470-3476)
//     (S N_i)[i][k] = S[i][2] N_i[k][2] + S[i][4] N_i[k][4] + ...                        (:2026-2036, structural zeros skipped)
// so a lane that solves the unit columns in order 0..11 can fold column j into the twelve running sums the moment it exists, with the
// reference's products and the reference's order of additions, and drop it.  The 12x12 inverse (1 152 B per engine) no longer lives
// anywhere: no LDS matrix, no barriers, no exchange between the two states of an engine -- each lane = (engine, main | shadow) does the
// whole rebuild for itself.  LDS holds only the 36 running S N_i sums per lane (18 KB per wavefront): two wavefronts per SIMD.
//
// The factorisation exploits what ow_melange_lit.h's fast path established -- R_ldr only reaches the trailing 6x6 block, the host
// replays elimination steps 0..5 and the forward substitutions through them once per rate (OwConsts::ml_*) -- plus the SPARSITY of
// the factors: the circuit matrix has 38 structural non-zeros and its LU 61 (31 in U, 30 in L) instead of 144; the pattern is the same
// at every rate because it follows from which nodes share a component.  invert_n's dense loops multiply and subtract exact zeros
// there (x - 0 * y == x), so leaving those operations out gives the same bits.  The pattern is compiled in (MCOL_* below) and the host
// verifies it against the factors it computed (ml_sparse_ok); the device still CHECKS every pivot choice of the R-dependent steps
// against what invert_n would pick and takes the generic per-lane LU (HBM workspace) for that sample otherwise -- never seen to happen
// for R in the reference's clamp range, forced by OW_MEL_GENERIC=1 for the bit-identity test.
// Divisions by a pivot share the pivot's refined reciprocal (ow_rcp_refined / ow_div_y: the same instruction sequence as ow_div,
// evaluated once per pivot instead of once per quotient); the R-independent pivots of rows 0..5 use host reciprocals (ow_div_const).
#pragma once
#include "ow_melange_lit.h"

namespace owdev {

// sum_i -= T[t][j] * b  with the reference's rounding (product, then difference)
#define MCOL_SUB(sum, a, b) sum -= (a) * (b)

struct MelColT {
    // trailing 6x6 block (positions 6..11) after elimination: the 26 structurally non-zero factors and the reciprocals of its pivots
    double t00, t01, t02, t04;
    double t10, t11, t12, t14;
    double t20, t21, t22, t23, t24;
    double t32, t33, t34;
    double t40, t41, t42, t43, t44;
    double t50, t51, t52, t53, t54, t55;
    double y0, y1, y2, y3, y4, y5;
};

// Steps 6..11 of invert_n's elimination on the trailing block (gen_preamp.rs:2150-2181) with the R-dependent entry e66 in place.
// Returns false when a pivot choice differs from invert_n's (first maximum of the column, strict >) or a pivot is below 1e-30.
__device__ inline bool mel_col_factor(const OwConsts* __restrict__ K0, double pot, double alpha, MelColT& T) {
    const OwConsts* __restrict__ K = k_reload(K0);
    const double g66 = PRE_G[6][6] + (ow_div(1.0, pot) - PRE_POT_0_G_NOM);
    double e = g66 + alpha * PRE_C[6][6];
#pragma unroll
    for (int k = 0; k < 6; ++k) e -= K->ml_chain_m[k] * K->ml_chain_u[k];
    T.t00 = e;               T.t01 = K->ml_t0[0][1]; T.t02 = K->ml_t0[0][2]; T.t04 = K->ml_t0[0][4];
    T.t10 = K->ml_t0[1][0]; T.t11 = K->ml_t0[1][1]; T.t12 = K->ml_t0[1][2]; T.t14 = K->ml_t0[1][4];
    T.t20 = K->ml_t0[2][0]; T.t21 = K->ml_t0[2][1]; T.t22 = K->ml_t0[2][2]; T.t23 = K->ml_t0[2][3]; T.t24 = K->ml_t0[2][4];
    T.t32 = K->ml_t0[3][2]; T.t33 = K->ml_t0[3][3]; T.t34 = K->ml_t0[3][4];
    T.t40 = K->ml_t0[4][0]; T.t41 = K->ml_t0[4][1]; T.t42 = K->ml_t0[4][2]; T.t43 = K->ml_t0[4][3]; T.t44 = K->ml_t0[4][4];
    T.t50 = K->ml_t0[5][0]; T.t51 = K->ml_t0[5][1]; T.t52 = K->ml_t0[5][2]; T.t53 = K->ml_t0[5][3]; T.t54 = K->ml_t0[5][4]; T.t55 = K->ml_t0[5][5];
    bool ok = true;
    // k = 0: rows 1, 2, 4, 5 carry a column-0 entry; U row 0 = {1, 2, 4}
    {
        const double pk = fabs(T.t00);
        ok = ok && !(fabs(T.t10) > pk) && !(fabs(T.t20) > pk) && !(fabs(T.t40) > pk) && !(fabs(T.t50) > pk) && !(pk < 1e-30);
        T.y0 = ow_rcp_refined(T.t00);
        double m;
        m = ow_div_y(T.t10, T.t00, T.y0); T.t10 = m; MCOL_SUB(T.t11, m, T.t01); MCOL_SUB(T.t12, m, T.t02); MCOL_SUB(T.t14, m, T.t04);
        m = ow_div_y(T.t20, T.t00, T.y0); T.t20 = m; MCOL_SUB(T.t21, m, T.t01); MCOL_SUB(T.t22, m, T.t02); MCOL_SUB(T.t24, m, T.t04);
        m = ow_div_y(T.t40, T.t00, T.y0); T.t40 = m; MCOL_SUB(T.t41, m, T.t01); MCOL_SUB(T.t42, m, T.t02); MCOL_SUB(T.t44, m, T.t04);
        m = ow_div_y(T.t50, T.t00, T.y0); T.t50 = m; MCOL_SUB(T.t51, m, T.t01); MCOL_SUB(T.t52, m, T.t02); MCOL_SUB(T.t54, m, T.t04);
    }
    // k = 1: rows 2, 4, 5; U row 1 = {2, 4}
    {
        const double pk = fabs(T.t11);
        ok = ok && !(fabs(T.t21) > pk) && !(fabs(T.t41) > pk) && !(fabs(T.t51) > pk) && !(pk < 1e-30);
        T.y1 = ow_rcp_refined(T.t11);
        double m;
        m = ow_div_y(T.t21, T.t11, T.y1); T.t21 = m; MCOL_SUB(T.t22, m, T.t12); MCOL_SUB(T.t24, m, T.t14);
        m = ow_div_y(T.t41, T.t11, T.y1); T.t41 = m; MCOL_SUB(T.t42, m, T.t12); MCOL_SUB(T.t44, m, T.t14);
        m = ow_div_y(T.t51, T.t11, T.y1); T.t51 = m; MCOL_SUB(T.t52, m, T.t12); MCOL_SUB(T.t54, m, T.t14);
    }
    // k = 2: rows 3, 4, 5; U row 2 = {3, 4}
    {
        const double pk = fabs(T.t22);
        ok = ok && !(fabs(T.t32) > pk) && !(fabs(T.t42) > pk) && !(fabs(T.t52) > pk) && !(pk < 1e-30);
        T.y2 = ow_rcp_refined(T.t22);
        double m;
        m = ow_div_y(T.t32, T.t22, T.y2); T.t32 = m; MCOL_SUB(T.t33, m, T.t23); MCOL_SUB(T.t34, m, T.t24);
        m = ow_div_y(T.t42, T.t22, T.y2); T.t42 = m; MCOL_SUB(T.t43, m, T.t23); MCOL_SUB(T.t44, m, T.t24);
        m = ow_div_y(T.t52, T.t22, T.y2); T.t52 = m; MCOL_SUB(T.t53, m, T.t23); MCOL_SUB(T.t54, m, T.t24);
    }
    // k = 3: rows 4, 5; U row 3 = {4}
    {
        const double pk = fabs(T.t33);
        ok = ok && !(fabs(T.t43) > pk) && !(fabs(T.t53) > pk) && !(pk < 1e-30);
        T.y3 = ow_rcp_refined(T.t33);
        double m;
        m = ow_div_y(T.t43, T.t33, T.y3); T.t43 = m; MCOL_SUB(T.t44, m, T.t34);
        m = ow_div_y(T.t53, T.t33, T.y3); T.t53 = m; MCOL_SUB(T.t54, m, T.t34);
    }
    // k = 4: row 5; U row 4 is its diagonal alone
    {
        const double pk = fabs(T.t44);
        ok = ok && !(fabs(T.t54) > pk) && !(pk < 1e-30);
        T.y4 = ow_rcp_refined(T.t44);
        T.t54 = ow_div_y(T.t54, T.t44, T.y4);
    }
    ok = ok && !(fabs(T.t55) < 1e-30);
    T.y5 = ow_rcp_refined(T.t55);
    return ok;
}

// Structure of the unit columns.  Row exchanges put unknown 3's equation (the supply source row) at position 11 and equation 11 at
// position 3; everything else stays.  So the permuted unit vector of column c sits at position c (c = 3 -> 11, c = 11 -> 3), the
// forward substitution leaves zeros in front of it, and the host's tables are zero there:
//   MCOL_PART[c] bit t: ml_part[c][t] may be non-zero;  MCOL_BTOP[c] bit i: ml_btop[c][i] may be non-zero  (verified by the host).
// Column 3 is e_11 / U[11][11] (nothing else reaches column 11 of U), and unknown 3 is zero in every column but 11.  The solve below
// carries "known zero" flags through the substitutions at compile time and leaves out operations whose operand is a known zero --
// the reference subtracts 0 * x there.
__device__ constexpr unsigned MCOL_PART[12] = {0x3F, 0x3F, 0x3F, 0x20, 0x3F, 0x3F, 0x3F, 0x3E, 0x3C, 0x38, 0x30, 0x3F};
__device__ constexpr unsigned MCOL_BTOP[12] = {0x37, 0x36, 0x34, 0x00, 0x30, 0x20, 0x00, 0x00, 0x00, 0x00, 0x00, 0x38};

// a / pivot with the pivot's refined reciprocal, without v_div_fixup: numerators here are finite sums of finite products and the pivots
// were checked against 1e-30, so none of the special cases that instruction repairs (infinite / NaN operands, a zero divisor, exponents
// at the ends of the range) can occur; for everything else it returns its first operand unchanged -- except that it also sets the SIGN of
// a zero quotient: a zero numerator gives +0 here where IEEE gives -0 for unlike signs.  Equal as numbers, and a signed zero stays a zero
// through everything that follows (x - 0 * y, sums, products; never a divisor).  tests/test_gpu_division.py checks both statements.
OW_DEV double mcol_div(double a, double b, double y) {
#ifdef OW_IEEE_DIV
    return a / b;
#else
    const double q = a * y;
    const double r = __builtin_fma(-b, q, a);
    return __builtin_fma(r, y, q);
#endif
}

template <int COL>
struct MelColNz {      // which entries of the unit column are (structurally) non-zero at each stage
    static constexpr unsigned P = MCOL_PART[COL], B = MCOL_BTOP[COL];
    static constexpr bool f6 = P & 1, f7 = ((P >> 1) & 1) || f6, f8 = ((P >> 2) & 1) || f6 || f7, f9 = ((P >> 3) & 1) || f8,
                          f10 = ((P >> 4) & 1) || f6 || f7 || f8 || f9, f11 = ((P >> 5) & 1) || f6 || f7 || f8 || f9 || f10;
    static constexpr bool g11 = f11, g10 = f10, g9 = f9 || g10, g8 = f8 || g9 || g10, g7 = f7 || g8 || g10, g6 = f6 || g7 || g8 || g10;
    static constexpr bool g5 = ((B >> 5) & 1) || g6 || g7 || g8, g4 = ((B >> 4) & 1) || g5 || g7 || g8, g3 = (B >> 3) & 1,
                          g2 = ((B >> 2) & 1) || g3 || g4 || g5, g1 = ((B >> 1) & 1) || g2, g0 = (B & 1) || g1;
    static constexpr bool nz(int i) {
        return i == 0 ? g0 : i == 1 ? g1 : i == 2 ? g2 : i == 3 ? g3 : i == 4 ? g4 : i == 5 ? g5 : i == 6 ? g6 : i == 7 ? g7 : i == 8 ? g8 : i == 9 ? g9
               : i == 10 ? g10 : g11;
    }
};

// Unit column COL of S = A^-1: forward substitution through the trailing rows (the part through rows 0..5 is the host's ml_part /
// ml_btop), back substitution through the trailing block and through U's rows 5..0 (gen_preamp.rs:2184-2215).
#define MCOL_SUBZ(cond, sum, a, b) if constexpr (cond) MCOL_SUB(sum, a, b)
template <int COL>
__device__ inline void mel_col_solve(const OwConsts* __restrict__ K0, const MelColT& T, double b[12]) {
    using Z = MelColNz<COL>;
    const OwConsts* __restrict__ K = k_reload(K0);
    double b6 = 0.0, b7 = 0.0, b8 = 0.0, b9 = 0.0, b10 = 0.0, b11 = 0.0;
    if constexpr (Z::P & 1) b6 = K->ml_part[COL][0];
    if constexpr ((Z::P >> 1) & 1) b7 = K->ml_part[COL][1];
    MCOL_SUBZ(Z::f6, b7, T.t10, b6);
    if constexpr ((Z::P >> 2) & 1) b8 = K->ml_part[COL][2];
    MCOL_SUBZ(Z::f6, b8, T.t20, b6);  MCOL_SUBZ(Z::f7, b8, T.t21, b7);
    if constexpr ((Z::P >> 3) & 1) b9 = K->ml_part[COL][3];
    MCOL_SUBZ(Z::f8, b9, T.t32, b8);
    if constexpr ((Z::P >> 4) & 1) b10 = K->ml_part[COL][4];
    MCOL_SUBZ(Z::f6, b10, T.t40, b6); MCOL_SUBZ(Z::f7, b10, T.t41, b7); MCOL_SUBZ(Z::f8, b10, T.t42, b8); MCOL_SUBZ(Z::f9, b10, T.t43, b9);
    if constexpr ((Z::P >> 5) & 1) b11 = K->ml_part[COL][5];
    MCOL_SUBZ(Z::f6, b11, T.t50, b6); MCOL_SUBZ(Z::f7, b11, T.t51, b7); MCOL_SUBZ(Z::f8, b11, T.t52, b8); MCOL_SUBZ(Z::f9, b11, T.t53, b9);
    MCOL_SUBZ(Z::f10, b11, T.t54, b10);
    if constexpr (Z::g11) b11 = mcol_div(b11, T.t55, T.y5);
    if constexpr (Z::g10) b10 = mcol_div(b10, T.t44, T.y4);
    MCOL_SUBZ(Z::g10, b9, T.t34, b10);
    if constexpr (Z::g9) b9 = mcol_div(b9, T.t33, T.y3);
    MCOL_SUBZ(Z::g9, b8, T.t23, b9);  MCOL_SUBZ(Z::g10, b8, T.t24, b10);
    if constexpr (Z::g8) b8 = mcol_div(b8, T.t22, T.y2);
    MCOL_SUBZ(Z::g8, b7, T.t12, b8);  MCOL_SUBZ(Z::g10, b7, T.t14, b10);
    if constexpr (Z::g7) b7 = mcol_div(b7, T.t11, T.y1);
    MCOL_SUBZ(Z::g7, b6, T.t01, b7);  MCOL_SUBZ(Z::g8, b6, T.t02, b8);  MCOL_SUBZ(Z::g10, b6, T.t04, b10);
    if constexpr (Z::g6) b6 = mcol_div(b6, T.t00, T.y0);
    // rows 5..0 of U (R-independent): row 5 = {6, 7, 8}, row 4 = {5, 7, 8}, row 3 = {}, row 2 = {3, 4, 5}, row 1 = {2}, row 0 = {1}
    double b5 = 0.0, b4 = 0.0, b3 = 0.0, b2 = 0.0, b1 = 0.0, b0 = 0.0;
    if constexpr ((Z::B >> 5) & 1) b5 = K->ml_btop[COL][5];
    MCOL_SUBZ(Z::g6, b5, K->ml_utop[5][6], b6); MCOL_SUBZ(Z::g7, b5, K->ml_utop[5][7], b7); MCOL_SUBZ(Z::g8, b5, K->ml_utop[5][8], b8);
    if constexpr (Z::g5) b5 = mcol_div(b5, K->ml_utop[5][5], K->ml_utop_rcp[5]);
    if constexpr ((Z::B >> 4) & 1) b4 = K->ml_btop[COL][4];
    MCOL_SUBZ(Z::g5, b4, K->ml_utop[4][5], b5); MCOL_SUBZ(Z::g7, b4, K->ml_utop[4][7], b7); MCOL_SUBZ(Z::g8, b4, K->ml_utop[4][8], b8);
    if constexpr (Z::g4) b4 = mcol_div(b4, K->ml_utop[4][4], K->ml_utop_rcp[4]);
    if constexpr (Z::g3) b3 = mcol_div(K->ml_btop[COL][3], K->ml_utop[3][3], K->ml_utop_rcp[3]);
    if constexpr ((Z::B >> 2) & 1) b2 = K->ml_btop[COL][2];
    MCOL_SUBZ(Z::g3, b2, K->ml_utop[2][3], b3); MCOL_SUBZ(Z::g4, b2, K->ml_utop[2][4], b4); MCOL_SUBZ(Z::g5, b2, K->ml_utop[2][5], b5);
    if constexpr (Z::g2) b2 = mcol_div(b2, K->ml_utop[2][2], K->ml_utop_rcp[2]);
    if constexpr ((Z::B >> 1) & 1) b1 = K->ml_btop[COL][1];
    MCOL_SUBZ(Z::g2, b1, K->ml_utop[1][2], b2);
    if constexpr (Z::g1) b1 = mcol_div(b1, K->ml_utop[1][1], K->ml_utop_rcp[1]);
    if constexpr (Z::B & 1) b0 = K->ml_btop[COL][0];
    MCOL_SUBZ(Z::g1, b0, K->ml_utop[0][1], b1);
    if constexpr (Z::g0) b0 = mcol_div(b0, K->ml_utop[0][0], K->ml_utop_rcp[0]);
    b[0] = b0; b[1] = b1; b[2] = b2; b[3] = b3; b[4] = b4; b[5] = b5; b[6] = b6; b[7] = b7; b[8] = b8; b[9] = b9; b[10] = b10; b[11] = b11;
}

// running S N_i sums of this lane in LDS: sni[k][i][lane]
#define MCOL_SNI(k, i) sni[((k) * 12 + (i)) * 64]

// Fold column COL of S into v_pred and into the S N_i sums (N_i rows: [0] = {2}, [1] = {2, 4, 5}, [2] = {4, 7, 8}).  Entries that are
// structurally zero add +-0 in the reference and are left out (none of them in the S N_i columns 2, 4, 5, 7, 8 besides unknown 3).
template <int COL>
__device__ inline void mel_col_fold_sni(const double b[12], double* __restrict__ sni) {
    if (COL == 2) {
#pragma unroll
        for (int i = 0; i < 12; ++i) { MCOL_SNI(0, i) = b[i] * PRE_N_I[0][2]; MCOL_SNI(1, i) = b[i] * PRE_N_I[1][2]; }
    } else if (COL == 4) {
#pragma unroll
        for (int i = 0; i < 12; ++i) { MCOL_SNI(1, i) = MCOL_SNI(1, i) + b[i] * PRE_N_I[1][4]; MCOL_SNI(2, i) = b[i] * PRE_N_I[2][4]; }
    } else if (COL == 5) {
#pragma unroll
        for (int i = 0; i < 12; ++i) MCOL_SNI(1, i) = MCOL_SNI(1, i) + b[i] * PRE_N_I[1][5];
    } else if (COL == 7) {
#pragma unroll
        for (int i = 0; i < 12; ++i) MCOL_SNI(2, i) = MCOL_SNI(2, i) + b[i] * PRE_N_I[2][7];
    } else if (COL == 8) {
#pragma unroll
        for (int i = 0; i < 12; ++i) MCOL_SNI(2, i) = MCOL_SNI(2, i) + b[i] * PRE_N_I[2][8];
    }
}
template <int COL>
__device__ inline void mel_col_fold(const double b[12], const double rhs[12], double acc[12], double* __restrict__ sni) {
    using Z = MelColNz<COL>;
#pragma unroll
    for (int i = 0; i < 12; ++i)
        if (Z::nz(i)) acc[i] += b[i] * rhs[COL];
    mel_col_fold_sni<COL>(b, sni);
}

template <int COL>
__device__ inline void mel_col_step(const OwConsts* __restrict__ K, const MelColT& T, const double rhs[12], double acc[12], double* __restrict__ sni) {
    double b[12];
    mel_col_solve<COL>(K, T, b);
    mel_col_fold<COL>(b, rhs, acc, sni);
}

// The generic rebuild for ONE lane (cold): invert_n statement for statement on an HBM workspace lu[144] (stride ld), then the same
// column folds.  In / out through a private block so that the hot path's arrays are never address-taken.
struct MelColGen { double rhs[12], acc[12], sni[3][12]; };
__device__ __noinline__ void mel_col_generic(double pot, double alpha, double* __restrict__ lu, size_t ld, MelColGen* __restrict__ g) {
#define GLU(r, c) lu[(size_t)((r) * 12 + (c)) * ld]
    const double g66 = PRE_G[6][6] + (ow_div(1.0, pot) - PRE_POT_0_G_NOM);
    for (int i = 0; i < 12; ++i)
        for (int j = 0; j < 12; ++j) GLU(i, j) = ((i == 6 && j == 6) ? g66 : PRE_G[i][j]) + alpha * PRE_C[i][j];
    int perm[12];
    for (int i = 0; i < 12; ++i) perm[i] = i;
    bool singular = false;
    for (int k = 0; k < 12 && !singular; ++k) {
        int max_row = k;
        double max_val = fabs(GLU(perm[k], k));
        for (int i = k + 1; i < 12; ++i) {
            const double v = fabs(GLU(perm[i], k));
            if (v > max_val) { max_val = v; max_row = i; }
        }
        if (max_val < 1e-30) { singular = true; break; }
        if (max_row != k) { const int t = perm[k]; perm[k] = perm[max_row]; perm[max_row] = t; }   // rows stay where they are: perm names them
        const int pr = perm[k];
        const double pivot = GLU(pr, k);
        for (int i = k + 1; i < 12; ++i) {
            const int ri = perm[i];
            const double m = ow_div(GLU(ri, k), pivot);
            GLU(ri, k) = m;
            for (int j = k + 1; j < 12; ++j) GLU(ri, j) -= m * GLU(pr, j);
        }
    }
    for (int i = 0; i < 12; ++i) g->acc[i] = 0.0;
    for (int k = 0; k < 3; ++k) for (int i = 0; i < 12; ++i) g->sni[k][i] = 0.0;
    for (int col = 0; col < 12; ++col) {
        double b[12];
        if (!singular) {
            for (int i = 0; i < 12; ++i) b[i] = (perm[i] == col) ? 1.0 : 0.0;
            for (int i = 1; i < 12; ++i) {
                double sum = b[i];
                for (int j = 0; j < i; ++j) sum -= GLU(perm[i], j) * b[j];
                b[i] = sum;
            }
            for (int i = 11; i >= 0; --i) {
                double sum = b[i];
                for (int j = i + 1; j < 12; ++j) sum -= GLU(perm[i], j) * b[j];
                const double pivot = GLU(perm[i], i);
                if (fabs(pivot) < 1e-30) singular = true;
                b[i] = ow_div(sum, pivot);
            }
        }
        if (singular) {                                  // invert_n hands back the identity (:2142-2148, :2203-2207)
            for (int i = 0; i < 12; ++i) b[i] = (i == col) ? 1.0 : 0.0;
            if (col > 0) {                               // ... for the WHOLE matrix: start over with identity columns
                for (int i = 0; i < 12; ++i) g->acc[i] = 0.0;
                for (int k = 0; k < 3; ++k) for (int i = 0; i < 12; ++i) g->sni[k][i] = 0.0;
                for (int c2 = 0; c2 < col; ++c2)
                    for (int i = 0; i < 12; ++i) {
                        const double s = (i == c2) ? 1.0 : 0.0;
                        g->acc[i] += s * g->rhs[c2];
                        for (int k = 0; k < 3; ++k) if (PRE_N_I[k][c2] != 0.0) g->sni[k][i] = g->sni[k][i] + s * PRE_N_I[k][c2];
                    }
            }
        }
        for (int i = 0; i < 12; ++i) {
            g->acc[i] += b[i] * g->rhs[col];
            for (int k = 0; k < 3; ++k)
                if (PRE_N_I[k][col] != 0.0) {
                    const bool first = (k == 0) || (k == 1 && col == 2) || (k == 2 && col == 4);
                    g->sni[k][i] = first ? b[i] * PRE_N_I[k][col] : g->sni[k][i] + b[i] * PRE_N_I[k][col];
                }
        }
    }
#undef GLU
}

// gen_preamp::process_sample (gen_preamp.rs:3399-3663) with the column-streamed rebuild above, in three pieces so that the
// lane = engine kernel (ow_melange_eng.h) can run the middle one once for both solver states.  pot: the resistance the matrices are
// built for (the engine's main state, as in k_preamp_mel_lit).
// (1) input clamp, denormal flush, BE cooldown, build_rhs (:3399-3468, 3041-3095)
__device__ inline void mel_col_pre(MelSt& st, double input_in, double pot, double alpha, const OwConsts* __restrict__ K0, const double* nz, int nz_stride,
                                   double rhs[12], double& input_out, bool& force_be_out) {
    const double input = isfinite(input_in) ? clampd(input_in, -100.0, 100.0) : 0.0;
#pragma unroll
    for (int i = 0; i < 12; ++i) st.v[i] = st.v[i] + 1e-25 - 1e-25;
#pragma unroll
    for (int i = 0; i < 3; ++i) st.ip[i] = st.ip[i] + 1e-25 - 1e-25;
    force_be_out = st.be_cooldown > 0u;
    if (st.be_cooldown > 0u) st.be_cooldown -= 1u;
#pragma unroll
    for (int i = 0; i < 11; ++i) rhs[i] = 0.0;                     // RHS_CONST (gen_preamp.rs:760-773); build_rhs :3041-3095
    rhs[11] = 15.0;
    {
        const OwConsts* __restrict__ K = k_reload(K0);
        const double (*__restrict__ an)[12] = K->m_aneg0;
        const double g66 = PRE_G[6][6] + (ow_div(1.0, pot) - PRE_POT_0_G_NOM);
        const double an66 = alpha * PRE_C[6][6] - g66;
        const double* v = st.v;
#define AN(i, j) an[i][j]
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
    }
    rhs[2] += PRE_N_I[0][2] * st.ip[0];
    rhs[2] += PRE_N_I[1][2] * st.ip[1];
    rhs[4] += PRE_N_I[1][4] * st.ip[1];
    rhs[4] += PRE_N_I[2][4] * st.ip[2];
    rhs[5] += PRE_N_I[1][5] * st.ip[1];
    rhs[7] += PRE_N_I[2][7] * st.ip[2];
    rhs[8] += PRE_N_I[2][8] * st.ip[2];
    rhs[0] += (input + st.input_prev) / PRE_INPUT_RESISTANCE;
    if (nz) nz_stamp(rhs, nz, nz_stride);
    input_out = input;
}

// K = N_v (S N_i) from the lane's running sums (:2038-2056); N_v rows: [0] = {2}, [1] = {2, 5}, [2] = {4, 8}
__device__ inline void mel_col_kernel(const double* __restrict__ sni, double kk[3][3]) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const double s2 = MCOL_SNI(j, 2), s4 = MCOL_SNI(j, 4), s5 = MCOL_SNI(j, 5), s8 = MCOL_SNI(j, 8);
        kk[0][j] = PRE_N_V[0][2] * s2;
        kk[1][j] = PRE_N_V[1][2] * s2 + PRE_N_V[1][5] * s5;
        kk[2][j] = PRE_N_V[2][4] * s4 + PRE_N_V[2][8] * s8;
    }
}

// (3) Newton on the junctions, v = v_pred + (S N_i) i_nl, BE fallback, voltage-damp net, NaN reset, state update (:3478-3663)
__device__ inline double mel_col_post(MelSt& st, double input, bool force_be, const double v_pred[12], const double kk[3][3], const double* __restrict__ sni,
                                      const double* nz, int nz_stride) {
    const double p[3] = {-v_pred[2], v_pred[2] - v_pred[5], v_pred[4] - v_pred[8]};
    double i_nl[3];
    uint32_t last_it = mel_solve_nl(p, kk, st.ip, st.ipp, i_nl);
    double vn[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) {                                  // v = v_pred + (S N_i) i_nl
        double x = v_pred[i];
        x += MCOL_SNI(0, i) * i_nl[0];
        x += MCOL_SNI(1, i) * i_nl[1];
        x += MCOL_SNI(2, i) * i_nl[2];
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

__device__ inline double mel_process_col(MelSt& st, double input_in, double pot, double alpha, const OwConsts* __restrict__ K0, double* __restrict__ sni,
                                         bool force_generic, double* __restrict__ lu, size_t lu_ld, const double* nz, int nz_stride) {
    double rhs[12], input;
    bool force_be;
    mel_col_pre(st, input_in, pot, alpha, K0, nz, nz_stride, rhs, input, force_be);
    // ---- (2) rebuild_matrices + v_pred = S rhs + S N_i, one unit column of S at a time
    double v_pred[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) v_pred[i] = 0.0;
    bool fast = !force_generic;
    if (fast) {
        MelColT T;
        fast = mel_col_factor(K0, pot, alpha, T);
        if (__builtin_expect(fast, 1)) {
            mel_col_step<0>(K0, T, rhs, v_pred, sni);
            mel_col_step<1>(K0, T, rhs, v_pred, sni);
            mel_col_step<2>(K0, T, rhs, v_pred, sni);
            mel_col_step<3>(K0, T, rhs, v_pred, sni);
            mel_col_step<4>(K0, T, rhs, v_pred, sni);
            mel_col_step<5>(K0, T, rhs, v_pred, sni);
            mel_col_step<6>(K0, T, rhs, v_pred, sni);
            mel_col_step<7>(K0, T, rhs, v_pred, sni);
            mel_col_step<8>(K0, T, rhs, v_pred, sni);
            mel_col_step<9>(K0, T, rhs, v_pred, sni);
            mel_col_step<10>(K0, T, rhs, v_pred, sni);
            mel_col_step<11>(K0, T, rhs, v_pred, sni);
        }
    }
    if (__builtin_expect(!fast, 0)) {
        MelColGen g;
        for (int i = 0; i < 12; ++i) g.rhs[i] = rhs[i];
        mel_col_generic(pot, alpha, lu, lu_ld, &g);
        for (int i = 0; i < 12; ++i) v_pred[i] = g.acc[i];
        for (int k = 0; k < 3; ++k) for (int i = 0; i < 12; ++i) MCOL_SNI(k, i) = g.sni[k][i];
    }
    double kk[3][3];
    mel_col_kernel(sni, kk);
    return mel_col_post(st, input, force_be, v_pred, kk, sni, nz, nz_stride);
}

// Preamp stream, literal rebuild, column-streamed.  Same interface as k_preamp_mel_lit; lu_scratch: [144][2 * ceil(I / 32) * 32] doubles,
// lane-minor, touched by the generic fallback only.
#ifndef OW_MEL_COL_WAVES
#define OW_MEL_COL_WAVES 2     // wavefronts per SIMD the register budget is set for (experiment, profiles/r06_melange_experiments.md: 1 = no spills, half the occupancy)
#endif
__global__ __launch_bounds__(64, OW_MEL_COL_WAVES) void k_preamp_mel_col(const OwConsts* __restrict__ K, double* __restrict__ cs,
                                                          const double* __restrict__ settled, const OwEngineArgs* __restrict__ args,
                                                          const OwEngineOut* __restrict__ eout, const double* __restrict__ sum, const OwTremSrc tsrc,
                                                          double* __restrict__ pre, double* __restrict__ noise, int I, int L,
                                                          int Lcap, int e0, int ne, int generic_only, double* __restrict__ lu_scratch, size_t lu_ld) {
    __shared__ double sni_all[36 * 64];
    const int lane = threadIdx.x;
    const int el = lane & 31, role = lane >> 5;
    const int eb = e0 + blockIdx.x * 32;
    const int e = eb + el;
    const bool valid = e < e0 + ne;
    const int ec = valid ? e : (e0 + ne - 1);
    const int osr = K->oversample ? 2 : 1;
    const TremCol rc = trem_col(tsrc, I, ec);
    const double alpha = 2.0 * (K->os_sr * 1.0);                    // gen_preamp.rs:1991-1992
    double* sni = sni_all + lane;
    double* lu = lu_scratch + (size_t)2 * (valid ? e : I + el) + role;   // this lane's workspace column (generic fallback); masked lanes get spare ones

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
    double* nzcol = noise ? noise + ec : nullptr;
    if (nz_mine && (dbits(CSF(CS_FLAGS)) & 1ull)) nz_reseed(nzcol, I);
    const bool force_generic = generic_only != 0;
    // voice sum of this engine (main lanes), read straight from its row one sample ahead: at ~2 500 instructions per sample one
    // uncoalesced 8-byte load per lane and sample is noise, and the staging tile + its barriers are gone
    const bool has_in = valid && role == 0 && !eout[ec].sum_nonfinite;
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
            const double a = allpass3(OW_OS_A0, OW_OS_A1, OW_OS_A2, ua, x);
            const double b = allpass3(OW_OS_B0, OW_OS_B1, OW_OS_B2, ub, x);
            in[0] = role ? 0.0 : a;
            in[1] = role ? 0.0 : b;
        } else {
            in[0] = role ? 0.0 : x;
            in[1] = 0.0;
        }
        for (int j = 0; j < osr; ++j) {
            const int s_i = n * osr + j;
            const size_t s_idx = (size_t)s_i;
            const double r_now = r_next;
            if (s_i + 1 < n_os) r_next = trem_col_at(rc, (uint32_t)(s_i + 1));
            mel_set_r(st, trem_shunt(depth, r_now));
            // the matrices follow the main state's resistance (see ow_melange_lit.h: a state NaN-reset on its own may sit within the 1e-12
            // hysteresis of its partner's value)
            const double pot_main = __shfl(st.pot, el);
            const double* nzp = nullptr;
            if (nz_on && scale_half != 0.0) {
                const double sir10 = st.pot == 9.99999999999999854e4 ? PRE_NOISE_THERMAL_SQRT_INV_R_DEFAULT[10] : sqrt(1.0 / st.pot);
                nz_draw(nzcol, I, scale_half, sir10);
                nzp = nzcol;
            }
            const uint32_t nan_before = st.nan_resets;
            const double o = mel_process_col(st, in[j], pot_main, alpha, K, sni, force_generic, lu, lu_ld, nzp, I);
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
