// Issue-rate microbenchmark for the f64 / integer instructions the brutus
// kernels lean on (gfx950).  Each kernel runs ITER x 8 independent copies of one
// instruction per lane, 4 waves per SIMD on every CU; the printed figure is
// "cycles per wave-instruction per SIMD" (v_fma_f64 = 4 means full rate... the
// table in HISTORY.md section 4 quotes these; round 5: tools/ubench/dpp_rate.hip).  Build + run:
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/f64_rates.hip -o /tmp/f64_rates && /tmp/f64_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITER = 4096;

#define BENCH_KERNEL(NAME, ASM, CONSTR_IO, INIT)                                               \
    __global__ void __launch_bounds__(256) NAME(double *out, double seed) {                       \
        double a0 = INIT + threadIdx.x * 1e-3, a1 = a0 + 0.1, a2 = a0 + 0.2, a3 = a0 + 0.3,      \
               a4 = a0 + 0.4, a5 = a0 + 0.5, a6 = a0 + 0.6, a7 = a0 + 0.7;                       \
        const double b = seed;                                                                   \
        for (int i = 0; i < ITER; ++i) {                                                         \
            asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                  \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6),  \
                           "+v"(a7)                                                              \
                         : "v"(b));                                                              \
        }                                                                                        \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;      \
    }

#define A_FMA(n) "v_fma_f64 %" #n ", %" #n ", %8, %8\n"
#define A_MUL(n) "v_mul_f64 %" #n ", %" #n ", %8\n"
#define A_ADD(n) "v_add_f64 %" #n ", %" #n ", %8\n"
#define A_RCP(n) "v_rcp_f64 %" #n ", %" #n "\n"
#define A_RSQ(n) "v_rsq_f64 %" #n ", %" #n "\n"
#define A_SQRT(n) "v_sqrt_f64 %" #n ", %" #n "\n"
#define A_RNDNE(n) "v_rndne_f64 %" #n ", %" #n "\n"
#define A_FREXPM(n) "v_frexp_mant_f64 %" #n ", %" #n "\n"
#define A_LDEXP(n) "v_ldexp_f64 %" #n ", %" #n ", 1\n"
#define A_DIVFIX(n) "v_div_fixup_f64 %" #n ", %" #n ", %8, %8\n"
#define A_DIVSCALE(n) "v_div_scale_f64 %" #n ", vcc, %" #n ", %8, %8\n"
#define A_DIVFMAS(n) "v_div_fmas_f64 %" #n ", %" #n ", %8, %8\n"
#define A_MAX(n) "v_max_f64 %" #n ", %" #n ", %8\n"
#define A_CNDMASK(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define A_CMP(n) "v_cmp_lt_f64 vcc, %" #n ", %8\n"
#define A_FMA32(n) "v_fma_f32 %" #n ", %" #n ", %8, %8\n"
#define A_PKFMA32(n) "v_pk_fma_f32 %" #n ", %" #n ", %8, %8\n"

BENCH_KERNEL(k_fma, A_FMA, , 1.0)
BENCH_KERNEL(k_mul, A_MUL, , 1.0)
BENCH_KERNEL(k_add, A_ADD, , 1.0)
BENCH_KERNEL(k_rcp, A_RCP, , 1.5)
BENCH_KERNEL(k_rsq, A_RSQ, , 1.5)
BENCH_KERNEL(k_sqrt, A_SQRT, , 1.5)
BENCH_KERNEL(k_rndne, A_RNDNE, , 1.5)
BENCH_KERNEL(k_frexpm, A_FREXPM, , 1.5)
BENCH_KERNEL(k_ldexp, A_LDEXP, , 1e-300)
BENCH_KERNEL(k_divfix, A_DIVFIX, , 1.5)
BENCH_KERNEL(k_divscale, A_DIVSCALE, , 1.5)
BENCH_KERNEL(k_divfmas, A_DIVFMAS, , 1.5)
BENCH_KERNEL(k_max, A_MAX, , 1.5)
BENCH_KERNEL(k_cmp, A_CMP, , 1.5)
BENCH_KERNEL(k_pkfma32, A_PKFMA32, , 1.5)

// 32-bit lanes: operate on the low dword registers of the doubles
#define BENCH_KERNEL32(NAME, ASM)                                                              \
    __global__ void __launch_bounds__(256) NAME(double *out, double seed) {                       \
        unsigned a0 = threadIdx.x + 1, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4,       \
                 a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;                                         \
        const unsigned b = (unsigned)seed | 1u;                                                  \
        for (int i = 0; i < ITER; ++i) {                                                         \
            asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                  \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6),  \
                           "+v"(a7)                                                              \
                         : "v"(b));                                                              \
        }                                                                                        \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (double)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7); \
    }
#define A_MULLO(n) "v_mul_lo_u32 %" #n ", %" #n ", %8\n"
#define A_MULHI(n) "v_mul_hi_u32 %" #n ", %" #n ", %8\n"
#define A_XOR(n) "v_xor_b32 %" #n ", %" #n ", %8\n"
#define A_ADD32(n) "v_add_u32 %" #n ", %" #n ", %8\n"
#define A_FMAF32(n) "v_fma_f32 %" #n ", %" #n ", %8, %8\n"
BENCH_KERNEL32(k_mullo, A_MULLO)
BENCH_KERNEL32(k_mulhi, A_MULHI)
BENCH_KERNEL32(k_xor, A_XOR)
BENCH_KERNEL32(k_add32, A_ADD32)
BENCH_KERNEL32(k_fmaf32, A_FMAF32)

// 64-bit results from 32-bit inputs / conversions: mixed register widths
__global__ void __launch_bounds__(256) k_mad64(double *out, double seed) {
    unsigned long long a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,
                       a6 = a0 + 6, a7 = a0 + 7;
    const unsigned b = (unsigned)seed | 1u, c = threadIdx.x | 3u;
#define A_MAD64(n) "v_mad_u64_u32 %" #n ", vcc, %8, %9, %" #n "\n"
    for (int i = 0; i < ITER; ++i) {
        asm volatile(A_MAD64(0) A_MAD64(1) A_MAD64(2) A_MAD64(3) A_MAD64(4) A_MAD64(5) A_MAD64(6) A_MAD64(7)
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c)
                     : "vcc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (double)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
}
__global__ void __launch_bounds__(256) k_cvt_f64_u32(double *out, double seed) {
    double a0, a1, a2, a3, a4, a5, a6, a7;
    const unsigned b = (unsigned)seed + threadIdx.x;
    double acc = 0.;
#define A_CVT(n) "v_cvt_f64_u32 %" #n ", %8\n"
    for (int i = 0; i < ITER; ++i) {
        asm volatile(A_CVT(0) A_CVT(1) A_CVT(2) A_CVT(3) A_CVT(4) A_CVT(5) A_CVT(6) A_CVT(7)
                     : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7)
                     : "v"(b));
    }
    acc = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
__global__ void __launch_bounds__(256) k_cvt_i32_f64(double *out, double seed) {
    int a0, a1, a2, a3, a4, a5, a6, a7;
    const double b = seed + threadIdx.x;
#define A_CVTI(n) "v_cvt_i32_f64 %" #n ", %8\n"
    for (int i = 0; i < ITER; ++i) {
        asm volatile(A_CVTI(0) A_CVTI(1) A_CVTI(2) A_CVTI(3) A_CVTI(4) A_CVTI(5) A_CVTI(6) A_CVTI(7)
                     : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7)
                     : "v"(b));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (double)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
}
__global__ void __launch_bounds__(256) k_frexp_exp(double *out, double seed) {
    int a0, a1, a2, a3, a4, a5, a6, a7;
    const double b = seed + threadIdx.x;
#define A_FREXPE(n) "v_frexp_exp_i32_f64 %" #n ", %8\n"
    for (int i = 0; i < ITER; ++i) {
        asm volatile(A_FREXPE(0) A_FREXPE(1) A_FREXPE(2) A_FREXPE(3) A_FREXPE(4) A_FREXPE(5) A_FREXPE(6) A_FREXPE(7)
                     : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7)
                     : "v"(b));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (double)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
}

typedef void (*kern_t)(double *, double);

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    const double mhz = prop.clockRate / 1e3;
    printf("device %s, %d CUs, %.0f MHz\n", prop.gcnArchName, ncu, mhz);
    const int blocks = ncu * 4;            // 4 blocks x 4 waves per CU = 4 waves / SIMD
    double *out;
    CHECK(hipMalloc(&out, sizeof(double) * blocks * 256));
    struct { const char *name; kern_t k; } tests[] = {
        {"v_fma_f64", k_fma}, {"v_mul_f64", k_mul}, {"v_add_f64", k_add}, {"v_max_f64", k_max},
        {"v_cmp_lt_f64", k_cmp}, {"v_rcp_f64", k_rcp}, {"v_rsq_f64", k_rsq}, {"v_sqrt_f64", k_sqrt},
        {"v_rndne_f64", k_rndne}, {"v_frexp_mant_f64", k_frexpm}, {"v_frexp_exp_i32_f64", k_frexp_exp},
        {"v_ldexp_f64", k_ldexp}, {"v_div_scale_f64", k_divscale}, {"v_div_fmas_f64", k_divfmas},
        {"v_div_fixup_f64", k_divfix}, {"v_cvt_f64_u32", k_cvt_f64_u32}, {"v_cvt_i32_f64", k_cvt_i32_f64},
        {"v_mad_u64_u32", k_mad64}, {"v_mul_lo_u32", k_mullo}, {"v_mul_hi_u32", k_mulhi},
        {"v_xor_b32", k_xor}, {"v_add_u32", k_add32}, {"v_fma_f32", k_fmaf32}, {"v_pk_fma_f32", k_pkfma32},
    };
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (auto &t : tests) {
        hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, out, 1.25);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, out, 1.25);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        // per SIMD: 4 waves x ITER x 8 wave-instructions
        const double instr = 4.0 * ITER * 8;
        printf("%-22s %7.3f ms  %6.2f cycles/wave-instr/SIMD (at %.0f MHz nominal)\n", t.name, ms,
               ms * 1e-3 * mhz * 1e6 / instr, mhz);
    }
    return 0;
}
