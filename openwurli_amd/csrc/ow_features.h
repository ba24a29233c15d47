// openwurli-hip: harmonic feature extraction of the ML pipeline stage that consumes batch renders (gfx950, f64).
//
//   k_feat_window   block = segment          xw[n] = x[n] * hann(n, N); sum of squares (RMS)
//   k_feat_peaks    block = (segment, harmonic), 4 wavefronts
//                   |X_k| of the Hann-windowed, 4x zero-padded spectrum evaluated ONLY at the candidate bins k of that
//                   harmonic's +-1 % search band (a few dozen to a few hundred of the 2N+1 bins) by direct summation:
//                   lanes stride over n, every lane carries 8 bins per pass (xw is read once per 8 bins), twiddles advance by
//                   a complex rotation and are re-seeded from an exact integer phase every 32 steps.
//
// Mirrors ml/goertzel_utils.py:60-107 (extract_harmonics_fft): numpy computes all bins with an FFT and then looks at the
// same few; the peak bin (first maximum) and its magnitude are what the caller keeps.  Candidate bin ranges are computed on
// the host with numpy's own frequency-axis arithmetic (openwurli_hip.hip, ow_extract_harmonics).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace owdev {

struct OwSegDev {
    uint32_t row, start, n;      // audio row, first sample, length N
    uint32_t n_harm;
    uint64_t xw_off;             // offset of this segment's windowed copy in the scratch buffer
};
struct OwBinsDev {               // one per (segment, harmonic)
    uint32_t seg;
    uint32_t k_lo, k_hi;         // inclusive candidate range; k_lo > k_hi: skipped harmonic
    uint32_t pad;
};
struct OwPeakDev {
    uint32_t k;
    uint32_t pad;
    double re, im;
};

#define OW_PI 3.14159265358979323846

__global__ __launch_bounds__(256) void k_feat_window(const double* __restrict__ audio, size_t stride, const OwSegDev* __restrict__ segs,
                                                     double* __restrict__ xw, double* __restrict__ sumsq, int wav24_mode) {
    __shared__ double red[4];
    const OwSegDev s = segs[blockIdx.x];
    const double* x = audio + (size_t)s.row * stride + s.start;
    double* w = xw + s.xw_off;
    const double m1 = (double)s.n - 1.0;
    double acc = 0.0;
    for (uint32_t i = threadIdx.x; i < s.n; i += 256) {
        double v = x[i];
        if (wav24_mode >= 0) {   // what a 24-bit WAV of this sample reads back as: quantiser of the reference's writer, then int / 2^23
            const double mx = 8388607.0;
            double q = wav24_mode == 0 ? round(v * mx) : trunc(fmin(fmax(v, -1.0), 1.0) * mx);   // Rust round(): half away from zero
            if (!(q == q)) q = 0.0;
            v = fmin(fmax(q, -mx), mx) * (1.0 / 8388608.0);
        }
        acc += v * v;
        // np.hanning(M): n = arange(1-M, M, 2); 0.5 + 0.5*cos(pi*n/(M-1))   (M == 1 -> ones)
        const double nn = (double)(1 - (int64_t)s.n + 2 * (int64_t)i);
        const double h = s.n > 1 ? 0.5 + 0.5 * cos(OW_PI * nn / m1) : 1.0;
        w[i] = v * h;
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) sumsq[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

#define OW_FEAT_J 8        // bins per lane per pass over the segment
#define OW_FEAT_RESEED 32  // exact twiddle every 32 rotation steps (2048 samples)

__device__ inline void feat_twiddle(uint64_t k, uint64_t n, uint64_t nfft, double& c, double& s) {
    const uint64_t m = (k * n) % nfft;                    // exact phase index: k*n < 2^32 * 2^32 never reached (k, n < 2^22)
    sincospi(-2.0 * ((double)m / (double)nfft), &s, &c);  // e^{-2 pi i k n / nfft}
}

__global__ __launch_bounds__(256) void k_feat_peaks(const OwSegDev* __restrict__ segs, const OwBinsDev* __restrict__ bins,
                                                    const double* __restrict__ xw, OwPeakDev* __restrict__ peaks) {
    __shared__ double best_p[4], best_re[4], best_im[4];
    __shared__ uint32_t best_k[4];
    const OwBinsDev b = bins[blockIdx.x];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double bp = -1.0, bre = 0.0, bim = 0.0;
    uint32_t bk = 0xFFFFFFFFu;
    if (b.k_lo <= b.k_hi) {
        const OwSegDev s = segs[b.seg];
        const double* x = xw + s.xw_off;
        const uint64_t nfft = 4ull * s.n;
        const uint32_t nb = b.k_hi - b.k_lo + 1;
        // bins are dealt to the 4 wavefronts in groups of OW_FEAT_J
        for (uint32_t g0 = wave * OW_FEAT_J; g0 < nb; g0 += 4 * OW_FEAT_J) {
            double wr[OW_FEAT_J], wi[OW_FEAT_J], sr[OW_FEAT_J], si[OW_FEAT_J], ar[OW_FEAT_J], ai[OW_FEAT_J];
            uint64_t kk[OW_FEAT_J];
#pragma unroll
            for (int j = 0; j < OW_FEAT_J; ++j) {
                kk[j] = b.k_lo + min(g0 + j, nb - 1);      // the tail group repeats its last bin (ignored below)
                feat_twiddle(kk[j], 64, nfft, sr[j], si[j]);
                ar[j] = 0.0; ai[j] = 0.0;
            }
            uint32_t step = 0;
            for (uint32_t n = lane; n < s.n; n += 64, ++step) {
                if ((step % OW_FEAT_RESEED) == 0) {
#pragma unroll
                    for (int j = 0; j < OW_FEAT_J; ++j) feat_twiddle(kk[j], n, nfft, wr[j], wi[j]);
                }
                const double v = x[n];
#pragma unroll
                for (int j = 0; j < OW_FEAT_J; ++j) {
                    ar[j] += v * wr[j];
                    ai[j] += v * wi[j];
                    const double t = wr[j] * sr[j] - wi[j] * si[j];
                    wi[j] = wr[j] * si[j] + wi[j] * sr[j];
                    wr[j] = t;
                }
            }
#pragma unroll
            for (int j = 0; j < OW_FEAT_J; ++j) {
                double re = ar[j], im = ai[j];
                for (int o = 32; o > 0; o >>= 1) { re += __shfl_xor(re, o); im += __shfl_xor(im, o); }
                if (g0 + j < nb) {
                    const double p = re * re + im * im;
                    const uint32_t k = (uint32_t)kk[j];
                    if (p > bp || (p == bp && k < bk)) { bp = p; bre = re; bim = im; bk = k; }   // np.argmax: first maximum
                }
            }
        }
    }
    if (lane == 0) { best_p[wave] = bp; best_re[wave] = bre; best_im[wave] = bim; best_k[wave] = bk; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            if (best_p[w] > bp || (best_p[w] == bp && best_k[w] < bk)) { bp = best_p[w]; bre = best_re[w]; bim = best_im[w]; bk = best_k[w]; }
        OwPeakDev o;
        o.k = bk; o.pad = 0; o.re = bre; o.im = bim;
        peaks[blockIdx.x] = o;
    }
}

}  // namespace owdev
