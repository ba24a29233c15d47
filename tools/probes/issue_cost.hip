// Issue cost of the f64 vector instructions the voice / chain kernels are made of, one wavefront alone on a SIMD and two wavefronts
// sharing one: `hipcc --offload-arch=gfx950 -O2 tools/probes/issue_cost.hip -o /tmp/issue_cost && /tmp/issue_cost`.
// Each test runs REP x 32 copies of one instruction over eight independent register chains between two s_memtime reads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP 64
#define UNROLL8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define UNROLL32(X) UNROLL8(X) UNROLL8(X) UNROLL8(X) UNROLL8(X)

template <int WHICH>
__global__ void k_cost(double* out, long long* cyc, double seed) {
    double a[8], b = seed * 1.0000001, c = seed * 0.5;
    float f[8], g0 = (float)seed * 3.0f, g1 = (float)seed * 5.0f;
    for (int i = 0; i < 8; ++i) { a[i] = seed + i * 0.125 + threadIdx.x * 1e-3; f[i] = (float)a[i]; }
    long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < REP; ++r) {
#define FMA(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define MUL(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define ADD(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define RCP(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[i]));
#define RCPF(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[i]));
#define CVTDF(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(a[i]));
#define CVTFD(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[i]) : "v"(f[i]));
#define FIXUP(i) asm volatile("v_div_fixup_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define FMAS(i) asm volatile("v_div_fmas_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc");
#define SQRT(i) asm volatile("v_sqrt_f64 %0, %0" : "+v"(a[i]));
#define RSQ(i) asm volatile("v_rsq_f64 %0, %0" : "+v"(a[i]));
#define MOV64(i) asm volatile("v_mov_b64 %0, %1" : "=v"(a[i]) : "v"(b));
#define FMAF(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
#define LDEXP(i) asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(a[i]));
#define FREXP(i) asm volatile("v_frexp_mant_f64 %0, %0" : "+v"(a[i]));
#define CMP(i) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
#define CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
#define DPP(i) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(f[i]) : "v"(f[(i + 1) & 7]));
#define MAX64(i) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define TRIG(i) asm volatile("v_trig_preop_f64 %0, %0, 1" : "+v"(a[i]));
#define FRACT(i) asm volatile("v_fract_f64 %0, %0" : "+v"(a[i]));
#define RNDNE(i) asm volatile("v_rndne_f64 %0, %0" : "+v"(a[i]));
#define CVTI(i) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(f[i]) : "v"(a[i]));
#define MADU64(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(a[i]) : "v"(f[i]) : "vcc");
#define CND64(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(f[i]) : "v"(f[(i + 1) & 7]) : "s10", "s11");
#define CNDIND(i) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(f[i]) : "v"(g0), "v"(g1));
#define MOV32(i) asm volatile("v_mov_b32 %0, %1" : "=v"(f[i]) : "v"(g0));
#define ADDU(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(f[i]) : "v"(g0));
#define ADDF(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(g0));
#define CMPS(i) asm volatile("v_cmp_lt_f64_e64 s[10:11], %0, %1" : : "v"(a[i]), "v"(b) : "s10", "s11");
#define CMPCND(i) asm volatile("v_cmp_lt_f64 vcc, %1, %2\n v_cndmask_b32 %0, %0, %3, vcc" : "+v"(f[i]) : "v"(a[i]), "v"(b), "v"(g0) : "vcc");
#define ANDB(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(f[i]) : "v"(g0));
#define LSHL64(i) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(a[i]));
#define PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
#define CVTU(i) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(a[i]) : "v"(f[i]));
#define CMPCND2(i) asm volatile("v_cmp_lt_f64 vcc, %2, %3\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc" : "+v"(f[i]), "+v"(f[(i + 4) & 7]) : "v"(a[i]), "v"(b), "v"(g0) : "vcc");
#define CMPCND2S(i) asm volatile("v_cmp_lt_f64_e64 s[10:11], %2, %3\n v_cndmask_b32_e64 %0, %0, %4, s[10:11]\n v_cndmask_b32_e64 %1, %1, %4, s[10:11]" : "+v"(f[i]), "+v"(f[(i + 4) & 7]) : "v"(a[i]), "v"(b), "v"(g0) : "s10", "s11");
#define CND4(i) asm volatile("v_cmp_lt_f64 vcc, %2, %3\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %0, %0, %5, vcc\n v_cndmask_b32 %1, %1, %5, vcc" : "+v"(f[i]), "+v"(f[(i + 4) & 7]) : "v"(a[i]), "v"(b), "v"(g0), "v"(g1) : "vcc");
#define CNDFMA(i) asm volatile("v_cndmask_b32 %0, %0, %2, vcc\n v_fma_f64 %1, %1, %3, %3" : "+v"(f[i]), "+v"(a[i]) : "v"(g0), "v"(b));
#define DEPFMA(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));
#define DEPMUL(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[0]) : "v"(b));
#define DEPRCP(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[0]));
        if (WHICH == 0) { UNROLL32(FMA) } if (WHICH == 1) { UNROLL32(MUL) } if (WHICH == 2) { UNROLL32(ADD) }
        if (WHICH == 3) { UNROLL32(RCP) } if (WHICH == 4) { UNROLL32(RCPF) } if (WHICH == 5) { UNROLL32(CVTDF) }
        if (WHICH == 6) { UNROLL32(CVTFD) } if (WHICH == 7) { UNROLL32(FIXUP) } if (WHICH == 8) { UNROLL32(FMAS) }
        if (WHICH == 9) { UNROLL32(SQRT) } if (WHICH == 10) { UNROLL32(RSQ) } if (WHICH == 11) { UNROLL32(MOV64) }
        if (WHICH == 12) { UNROLL32(FMAF) } if (WHICH == 13) { UNROLL32(LDEXP) } if (WHICH == 14) { UNROLL32(FREXP) }
        if (WHICH == 15) { UNROLL32(CMP) } if (WHICH == 16) { UNROLL32(CNDMASK) } if (WHICH == 17) { UNROLL32(DPP) }
        if (WHICH == 18) { UNROLL32(MAX64) } if (WHICH == 19) { UNROLL32(TRIG) } if (WHICH == 20) { UNROLL32(FRACT) }
        if (WHICH == 21) { UNROLL32(RNDNE) } if (WHICH == 22) { UNROLL32(CVTI) } if (WHICH == 23) { UNROLL32(MADU64) }
        if (WHICH == 24) { UNROLL32(DEPFMA) } if (WHICH == 25) { UNROLL32(DEPMUL) } if (WHICH == 26) { UNROLL32(DEPRCP) }
        if (WHICH == 27) { UNROLL32(CND64) } if (WHICH == 28) { UNROLL32(CNDIND) } if (WHICH == 29) { UNROLL32(MOV32) }
        if (WHICH == 30) { UNROLL32(ADDU) } if (WHICH == 31) { UNROLL32(ADDF) } if (WHICH == 32) { UNROLL32(CMPS) }
        if (WHICH == 33) { UNROLL32(CMPCND) } if (WHICH == 34) { UNROLL32(ANDB) } if (WHICH == 35) { UNROLL32(LSHL64) }
        if (WHICH == 36) { UNROLL32(PKFMA) } if (WHICH == 37) { UNROLL32(CVTU) }
        if (WHICH == 38) { UNROLL32(CMPCND2) } if (WHICH == 39) { UNROLL32(CMPCND2S) } if (WHICH == 40) { UNROLL32(CND4) } if (WHICH == 41) { UNROLL32(CNDFMA) }
    }
    long long t1 = __builtin_readcyclecounter();
    double s = 0; float g = 0;
    for (int i = 0; i < 8; ++i) { s += a[i]; g += f[i]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + g;
    if (threadIdx.x % 64 == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

__global__ void k_spin(long long ticks, long long* out) {
    long long t0 = __builtin_readcyclecounter(), t;
    double a = 1.0; long long n = 0;
    do { for (int i = 0; i < 256; ++i) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(a)); n += 256; t = __builtin_readcyclecounter(); } while (t - t0 < ticks);
    out[0] = t - t0; out[1] = n; out[2] = (long long)a;
}

// whole-chip rate of independent f64 FMAs (eight chains per lane) at W wavefronts per SIMD: the ceiling a VALU-bound kernel has at that occupancy
template <int DEP>
__global__ void __launch_bounds__(256) k_flops(double* out, int reps, double seed) {
    double a[8]; double b = seed * 1.0000001, c = seed * 1e-9;
    for (int i = 0; i < 8; ++i) a[i] = seed + i * 0.125 + threadIdx.x * 1e-3;
    for (int r = 0; r < reps; ++r) {
        if (DEP == 0) { UNROLL32(FMA) } else if (DEP == 1) { UNROLL32(DEPFMA) }
        else { UNROLL8(FMA) UNROLL8(MUL) UNROLL8(ADD) UNROLL8(FMA) }
    }
    double s = 0; for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static const char* NAMES[] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_rcp_f64", "v_rcp_f32", "v_cvt_f32_f64", "v_cvt_f64_f32",
    "v_div_fixup_f64", "v_div_fmas_f64", "v_sqrt_f64", "v_rsq_f64", "v_mov_b64", "v_fma_f32", "v_ldexp_f64", "v_frexp_mant_f64",
    "v_cmp_lt_f64", "v_cndmask_b32", "v_mov_b32_dpp", "v_max_f64", "v_trig_preop_f64", "v_fract_f64", "v_rndne_f64", "v_cvt_i32_f64",
    "v_mad_u64_u32", "v_fma_f64 (dependent chain)", "v_mul_f64 (dependent chain)", "v_rcp_f64 (dependent chain)",
    "v_cndmask_b32_e64 (sgpr mask)", "v_cndmask_b32 (independent)", "v_mov_b32", "v_add_u32", "v_add_f32", "v_cmp_lt_f64_e64 (sgpr dst)",
    "v_cmp_lt_f64 + v_cndmask_b32 (pair)", "v_and_b32", "v_lshlrev_b64", "v_pk_fma_f32", "v_cvt_f64_u32",
    "v_cmp_lt_f64 + 2 v_cndmask_b32 vcc (3 instr)", "v_cmp_lt_f64_e64 + 2 v_cndmask_b32_e64 sgpr (3 instr)", "v_cmp + 4 v_cndmask vcc (5 instr)", "v_cndmask vcc + v_fma_f64 (2 instr)"};

template <int W>
static void run_one(double* d_out, long long* d_cyc, int waves_per_block, int blocks) {
    std::vector<long long> h(blocks * waves_per_block);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_cost<W>, dim3(blocks), dim3(64 * waves_per_block), 0, 0, d_out, d_cyc, 1.25);
        hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), d_cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    double m = 0; for (auto v : h) m += (double)v; m /= h.size();
    printf("%-30s %2d wave(s)/SIMD  %7.2f clock ticks per instruction and wavefront\n", NAMES[W], waves_per_block / 4, m / (REP * 32.0));
}

template <int W> struct Runner { static void go(double* o, long long* c) {
    run_one<W>(o, c, 4, 1);      // 4 wavefronts in one workgroup = one per SIMD of one CU
    run_one<W>(o, c, 8, 1);      // two per SIMD
    run_one<W>(o, c, 16, 1);     // four per SIMD
    Runner<W + 1>::go(o, c); } };
template <> struct Runner<42> { static void go(double*, long long*) {} };

int main() {
    double* d_out; long long* d_cyc;
    hipMalloc(&d_out, 1 << 20); hipMalloc(&d_cyc, 1 << 16);
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    int wclk = 0; hipDeviceGetAttribute(&wclk, hipDeviceAttributeWallClockRate, 0);
    printf("# shader clock %d kHz, wall clock %d kHz; s_memtime ticks (see the v_fma_f64 row for the scale: 4 shader cycles per f64 FMA)\n", clk, wclk);
    {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, 0, 200000000LL, d_cyc);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            long long h[3]; hipMemcpy(h, d_cyc, sizeof(h), hipMemcpyDeviceToHost);
            printf("# calibration: %lld ticks in %.3f ms = %.1f MHz tick rate; %lld dependent v_fma_f64 of one wavefront = %.2f ns each\n", h[0], ms, h[0] / (ms * 1e3), h[1], ms * 1e6 / h[1]);
        }
    }
    {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        double* big; hipMalloc(&big, 256 * 16 * 256 * sizeof(double));
        for (int dep = 0; dep < 3; ++dep)
            for (int w : {1, 2, 3, 4, 6, 8}) {
                int reps = 40000;
                float best = 1e30f;
                for (int rep = 0; rep < 3; ++rep) {
                    hipEventRecord(e0, 0);
                    if (dep == 0) hipLaunchKernelGGL(k_flops<0>, dim3(256 * w), dim3(256), 0, 0, big, reps, 1.25);
                    else if (dep == 1) hipLaunchKernelGGL(k_flops<1>, dim3(256 * w), dim3(256), 0, 0, big, reps, 1.25);
                    else hipLaunchKernelGGL(k_flops<2>, dim3(256 * w), dim3(256), 0, 0, big, reps, 1.25);
                    hipEventRecord(e1, 0); hipEventSynchronize(e1);
                    float ms = 0; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
                }
                double inst = 256.0 * w * 4 * reps * 32;       // wavefront-instructions
                printf("# whole chip, %s, %d wavefront(s) per SIMD: %.3f ms, %.2f ns per instruction and SIMD, %.1f T f64 instructions x 64 lanes /s (FMA = 2 flops: %.1f TFLOP/s)\n",
                       dep == 0 ? "independent v_fma_f64" : dep == 1 ? "one dependent v_fma_f64 chain" : "fma/mul/add/fma mix, independent", w, best,
                       best * 1e6 / (reps * 32.0 * w), inst * 64 / (best * 1e-3) / 1e12, inst * 128 / (best * 1e-3) / 1e12);
            }
    }
    Runner<0>::go(d_out, d_cyc);
    return 0;
}
