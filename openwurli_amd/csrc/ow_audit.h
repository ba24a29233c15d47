// openwurli-hip: click-band alias audit of rendered stimuli (gfx950, f64).
//
//   k_audit_gather    one rendered block (f32 [engine][Lcap]) appended to the f64 signal rows (`*s as f64`, alias_audit.rs:155)
//   k_audit_dft       block = (probe frequency, signal), 4 wavefronts: re = sum x[i] cos(w i), im = -sum x[i] sin(w i) over the
//                     analysis tail, phase = w * i evaluated exactly as the reference does (alias_audit.rs:229-240, no recurrence)
//   k_audit_bandpass  lane = signal: 4th-order HP + 4th-order LP (four RBJ biquads, DF-II transposed) and the sum of squares
//                     (alias_audit.rs:270-282); the coefficients come from the host (libm cos/sin, as filters.rs computes them)
//
// The frequency lists (refine_f0's 0.1 Hz walk, then (k+1)*f0) and the argmax are host work (openwurli_hip.hip,
// ow_alias_audit_analyze): ~100 numbers per signal.  FP64 VALU bound (sincos per sample per probe); HBM traffic is the tail
// once per probe and L2-resident.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace owdev {

struct OwAuditBq { double b0, b1, b2, a1, a2; };
struct OwAuditBand { OwAuditBq hp, lp; };   // hp1 == hp2 and lp1 == lp2 (same design, separate state)

__global__ __launch_bounds__(256) void k_audit_gather(const float* __restrict__ block, size_t block_stride, double* __restrict__ sig,
                                                      size_t sig_stride, size_t pos, uint32_t len) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const size_t e = blockIdx.y;
    if (i < len) sig[e * sig_stride + pos + i] = (double)block[e * block_stride + i];
}

// omega: [n_sig][omega_stride] probe angular frequencies (2 pi f / sr), n_probe[sig] of them valid; out: [n_sig][omega_stride] (re, im)
__global__ __launch_bounds__(256) void k_audit_dft(const double* __restrict__ sig, size_t sig_stride, size_t tail_off, uint32_t n,
                                                   const double* __restrict__ omega, const uint32_t* __restrict__ n_probe,
                                                   uint32_t omega_stride, double2* __restrict__ out) {
    __shared__ double red[2][4];
    const uint32_t s = blockIdx.y, f = blockIdx.x;
    if (f >= n_probe[s]) return;
    const double* x = sig + (size_t)s * sig_stride + tail_off;
    const double w = omega[(size_t)s * omega_stride + f];
    double re = 0.0, im = 0.0;
    for (uint32_t i = threadIdx.x; i < n; i += 256) {
        double sn, cs;
        sincos(w * (double)i, &sn, &cs);
        const double v = x[i];
        re += v * cs;
        im -= v * sn;
    }
    for (int off = 32; off > 0; off >>= 1) { re += __shfl_down(re, off); im += __shfl_down(im, off); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = re; red[1][threadIdx.x >> 6] = im; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double2 r;
        r.x = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        r.y = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        out[(size_t)s * omega_stride + f] = r;
    }
}

__device__ inline double audit_bq(const OwAuditBq& c, double& s1, double& s2, double x) {
    const double y = c.b0 * x + s1;
    s1 = c.b1 * x - c.a1 * y + s2;
    s2 = c.b2 * x - c.a2 * y;
    return y;
}

__global__ __launch_bounds__(64) void k_audit_bandpass(const double* __restrict__ sig, size_t sig_stride, size_t tail_off, uint32_t n,
                                                       uint32_t n_sig, OwAuditBand c, double* __restrict__ sumsq) {
    const uint32_t s = blockIdx.x * 64u + threadIdx.x;
    if (s >= n_sig) return;
    const double* x = sig + (size_t)s * sig_stride + tail_off;
    double h1a = 0, h1b = 0, h2a = 0, h2b = 0, l1a = 0, l1b = 0, l2a = 0, l2b = 0, acc = 0.0;
    for (uint32_t i = 0; i < n; ++i) {
        const double y = audit_bq(c.lp, l2a, l2b, audit_bq(c.lp, l1a, l1b, audit_bq(c.hp, h2a, h2b, audit_bq(c.hp, h1a, h1b, x[i]))));
        acc += y * y;
    }
    sumsq[s] = acc;
}

}  // namespace owdev
