// Issue rate of DPP row_newbcast VALU operations against plain ones (gfx950).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/dpp_rate.hip -o tools/ubench/dpp_rate && tools/ubench/dpp_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int MODE>
__global__ void k(float *out, int iters) {
    float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f, a4 = 4.f, a5 = 5.f, a6 = 6.f, a7 = 7.f;
    float r = threadIdx.x * 0.5f, x = 1.0001f;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
            REP16(asm volatile("v_fmac_f32_e32 %0, %8, %9\n v_fmac_f32_e32 %1, %8, %9\n v_fmac_f32_e32 %2, %8, %9\n v_fmac_f32_e32 %3, %8, %9\n"
                               "v_fmac_f32_e32 %4, %8, %9\n v_fmac_f32_e32 %5, %8, %9\n v_fmac_f32_e32 %6, %8, %9\n v_fmac_f32_e32 %7, %8, %9\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(r), "v"(x));)
        } else if (MODE == 1) {
            REP16(asm volatile("v_fmac_f32_dpp %0, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %1, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_fmac_f32_dpp %2, %8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %3, %8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_fmac_f32_dpp %4, %8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %5, %8, %9 row_newbcast:9 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_fmac_f32_dpp %6, %8, %9 row_newbcast:11 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %7, %8, %9 row_newbcast:0 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(r), "v"(x));)
        } else if (MODE == 2) {   // transcendental
            REP16(asm volatile("v_exp_f32_e32 %0, %0\n v_exp_f32_e32 %1, %1\n v_exp_f32_e32 %2, %2\n v_exp_f32_e32 %3, %3\n"
                               "v_exp_f32_e32 %4, %4\n v_exp_f32_e32 %5, %5\n v_exp_f32_e32 %6, %6\n v_exp_f32_e32 %7, %7\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(r), "v"(x));)
        } else if (MODE == 3) {   // f64 fma
            double d0 = a0, d1 = a1, d2 = a2, d3 = a3, dr = r, dx = x;
            REP16(asm volatile("v_fmac_f64_e32 %0, %4, %5\n v_fmac_f64_e32 %1, %4, %5\n v_fmac_f64_e32 %2, %4, %5\n v_fmac_f64_e32 %3, %4, %5\n"
                               "v_fmac_f64_e32 %0, %4, %5\n v_fmac_f64_e32 %1, %4, %5\n v_fmac_f64_e32 %2, %4, %5\n v_fmac_f64_e32 %3, %4, %5\n"
                               : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(dr), "v"(dx));)
            a0 = (float)d0; a1 = (float)d1; a2 = (float)d2; a3 = (float)d3;
        } else if (MODE == 4) {   // packed f32
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pr = {r, r}, px = {x, x};
            REP16(asm volatile("v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3\n"
                               "v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3\n"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pr), "v"(px));)
            a0 = p0.x + p0.y; a1 = p1.x + p1.y; a2 = p2.x + p2.y; a3 = p3.x + p3.y;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int MODE>
void run(const char *name, int waves_per_simd) {
    float *d;
    const int blocks = 256 * waves_per_simd, iters = 2000;     // 256-thread blocks: one wave per SIMD each
    hipMalloc(&d, sizeof(float) * blocks * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, 10);
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double insts = (double)iters * 128 * waves_per_simd;      // wave-instructions per SIMD
    printf("%-28s %d waves/SIMD: %.3f ms, %.2f ns per wave-instruction per SIMD (= %.2f cycles at 2.4 GHz)\n", name,
           waves_per_simd, ms, ms * 1e6 / insts, ms * 1e6 / insts * 2.4);
    hipFree(d);
}
int main() {
    for (int w : {1, 2, 4}) {
        run<0>("v_fmac_f32", w);
        run<1>("v_fmac_f32_dpp row_newbcast", w);
        run<2>("v_exp_f32", w);
        run<3>("v_fmac_f64", w);
        run<4>("v_pk_fma_f32", w);
    }
    return 0;
}
