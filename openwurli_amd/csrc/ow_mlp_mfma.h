// openwurli-hip: note-on MLP (2 -> 16 -> 16 -> 11, mlp_correction.rs:86-116) for a whole wavefront of note-ons on the
// f64 matrix cores: v_mfma_f64_16x16x4_f64, D(16x16) = A(16x4) * B(4x16) + C.
//
//   lane = voice slot.  Layer 1 (K = 2) is two scalar MACs per hidden unit.  Layers 2 and 3 are 16x16 (x16 voices)
//   tiles: A = weight block W[:, 4c..4c+3], B = activations of 16 voices, K = 16 in four MFMA steps, four voice tiles
//   per wavefront -> 32 MFMAs replace 2 x 432 scalar multiply-adds per lane.  Activations cross lanes through a
//   64 x 17 LDS tile (row = voice, padded).
//
// Fragment layout (cdna_hip_programming.md section 3, f64 form): A: lane l holds A[i = l & 15][k = l >> 4];
// B: lane l holds B[k = l >> 4][j = l & 15]; C/D: lane l, register r holds D[row = (l >> 4) + 4 r][col = l & 15].
// The bias enters as the C operand, so each output is bias + sum_k (fused multiply-adds); it differs from the scalar
// reference order by f64 rounding only (checked against the scalar path by tests/test_gpu_parity.py::test_mlp_mfma).
#pragma once
#include "ow_kernels.h"

namespace owdev {

typedef double v4f64 __attribute__((ext_vector_type(4)));

// h: LDS tile [64][17]; on entry row v holds the 16 input activations of voice v, on exit the 16 outputs.
// W: [rows][16] row-major weights (rows <= 16), bias[rows].  relu: apply max(x, 0).
__device__ inline void mlp_layer_mfma(double* __restrict__ h, const double (*__restrict__ W)[16], const double* __restrict__ bias, int rows, bool relu) {
    const int lane = threadIdx.x & 63;
    const int i = lane & 15, kq = lane >> 4;
    v4f64 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {            // voice tile t = voices 16t .. 16t+15
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = kq + 4 * r;
            acc[t][r] = row < rows ? bias[row] : 0.0;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {        // K chunk: hidden units 4c .. 4c+3
            const double a = i < rows ? W[i][4 * c + kq] : 0.0;
            const double b = h[(16 * t + i) * 17 + 4 * c + kq];
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
        }
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double x = acc[t][r];
            if (relu) x = x > 0.0 ? x : 0.0;
            h[(16 * t + i) * 17 + kq + 4 * r] = x;
        }
    __syncthreads();
}

// raw[11] for every lane (inputs in0/in1 per lane; lanes without a note-on may pass zeros).  Must be called by all
// 64 lanes of the block in uniform control flow.  h = LDS scratch of 64*17 doubles.
__device__ inline void mlp_raw_mfma(double* __restrict__ h, double in0, double in1, double raw[11]) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < 16; ++i) {           // layer 1: affine + ReLU, same order as the scalar reference
        double sum = MLP_B1[i];
        sum += MLP_W1[i][0] * in0;
        sum += MLP_W1[i][1] * in1;
        h[lane * 17 + i] = sum > 0.0 ? sum : 0.0;
    }
    __syncthreads();
    mlp_layer_mfma(h, MLP_W2, MLP_B2, 16, true);
    mlp_layer_mfma(h, MLP_W3, MLP_B3, 11, false);
#pragma unroll
    for (int i = 0; i < 11; ++i) raw[i] = h[lane * 17 + i] * MLP_TARGET_STDS[i] + MLP_TARGET_MEANS[i];
    __syncthreads();
}

// debug/test kernel: raw MLP outputs for n (note, velocity) pairs, scalar or MFMA path
__global__ __launch_bounds__(64) void k_debug_mlp(const uint8_t* __restrict__ notes, const double* __restrict__ vels, int n, double* __restrict__ out, int use_mfma) {
    __shared__ double h[64 * 17];
    const int idx = blockIdx.x * 64 + threadIdx.x;
    const bool valid = idx < n;
    const double midi = valid ? (double)notes[idx] : 60.0;
    const double vel = valid ? vels[idx] : 0.0;
    const double in0 = clampd((midi - 21.0) / (108.0 - 21.0), 0.0, 1.0), in1 = clampd(vel, 0.0, 1.0);
    double raw[11];
    if (use_mfma) mlp_raw_mfma(h, in0, in1, raw);
    else mlp_raw_scalar(in0, in1, raw);
    if (valid)
        for (int i = 0; i < 11; ++i) out[(size_t)idx * 11 + i] = raw[i];
}

// OW_DIV_C against the compiler's division, for the constants the kernels divide by (tests/test_gpu_division.py)
#define OW_DIVC_CASES(X) X(0, OW_JITTER_DIV) X(1, 1.0 * OW_T_VT) X(2, 10.95 - 0.70) X(3, OW_P_VT) X(4, 0.013 * 0.013) X(5, 22.0) X(6, 2147483647.0) X(7, 0.98 - 0.94)
__device__ inline void divc_pair(int which, double a, double& fast, double& ieee) {
    fast = ieee = 0.0;
#define OW_DIVC_ONE(i, B) if (which == i) { fast = OW_DIV_C(a, B); ieee = a / (B); }
    OW_DIVC_CASES(OW_DIVC_ONE)
#undef OW_DIVC_ONE
}
__global__ __launch_bounds__(256) void k_debug_div_const(int which, const double* __restrict__ a, size_t n, double* __restrict__ fast, double* __restrict__ ieee) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) divc_pair(which, a[i], fast[i], ieee[i]);
}
// every numerator a draw can produce, counting quotients whose bits differ.  which = 0: the jitter draw, (jitter_state >> 1) for all 2^31
// values over 2147483647.5 (reed.rs:267-272); which = 6: the attack-noise draw, every int32 over 2147483647.0 (hammer.rs:192-195)
__global__ __launch_bounds__(256) void k_debug_div_draw_all(int which, unsigned long long* __restrict__ mismatches) {
    unsigned long long bad = 0;
    const unsigned long long n = which == 0 ? (1ull << 31) : (1ull << 32);
    for (unsigned long long v = (unsigned long long)blockIdx.x * 256 + threadIdx.x; v < n; v += (unsigned long long)gridDim.x * 256) {
        double f, e;
        divc_pair(which, which == 0 ? (double)(uint32_t)v : (double)(int32_t)(uint32_t)v, f, e);
        bad += __double_as_longlong(f) != __double_as_longlong(e);
    }
    if (bad) atomicAdd(mismatches, bad);
}

// exp_bounded against the library's exp (tests/test_gpu_division.py)
__global__ __launch_bounds__(256) void k_debug_exp(int which, const double* __restrict__ x, size_t n, double* __restrict__ fast, double* __restrict__ lib) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (which == 0) { fast[i] = exp_bounded(x[i]); lib[i] = exp(x[i]); }
    else if (which == 1) { fast[i] = tanh_fast(x[i]);   lib[i] = tanh(x[i]); }
    else if (which == 5) { fast[i] = exp_neg_small(x[i]); lib[i] = exp(-x[i]); }       // damper-ramp factor (ow_voice_dev.h)
    else if (which == 6) { fast[i] = env_after(1.0, x[i], 512u + (uint32_t)(i & 1023u)); lib[i] = pow(x[i], (double)(512u + (uint32_t)(i & 1023u))); }   // the folded steady kernel's envelope (ow_kernels.h)
    else {               // 2, 3, 4: the onset gain at phase x with shape exponent 1.25 / 1.5 / 1.9, next to the library's pow
        const double p = which == 2 ? 1.25 : (which == 3 ? 1.5 : 1.9);
        fast[i] = onset_gain(x[i], 1.0, p);
        lib[i] = pow(0.5 * (1.0 - cos(x[i])), p);
    }
}

// ow_div against the compiler's division (tests/test_gpu_division.py)
__global__ __launch_bounds__(256) void k_debug_div(const double* __restrict__ a, const double* __restrict__ b, size_t n, double* __restrict__ fast,
                                                   double* __restrict__ ieee) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { fast[i] = ow_div(a[i], b[i]); ieee[i] = a[i] / b[i]; }
}

// The other short division forms next to the compiler's a / b (tests/test_gpu_division.py): mode 0 = ow_div_const(a, b, y) with a host
// reciprocal y (the per-device divisors of the power amp), 1 = ow_div_y(a, b, ow_rcp_refined(b)) (shared pivot reciprocals), 2 = the
// melange column kernel's quotient without v_div_fixup (mcol_div in ow_melange_col.h: same three operations).
__global__ __launch_bounds__(256) void k_debug_div_forms(int mode, const double* __restrict__ a, const double* __restrict__ b, const double* __restrict__ y, size_t n,
                                                         double* __restrict__ fast, double* __restrict__ ieee) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double r;
    if (mode == 0) r = ow_div_const(a[i], b[i], y[i]);
    else if (mode == 1) r = ow_div_y(a[i], b[i], ow_rcp_refined(b[i]));
    else {
        const double yy = ow_rcp_refined(b[i]);
        const double q = a[i] * yy;
        const double rr = __builtin_fma(-b[i], q, a[i]);
        r = __builtin_fma(rr, yy, q);
    }
    fast[i] = r;
    ieee[i] = a[i] / b[i];
}

}  // namespace owdev
