// Do float32 MFMAs and float32 vector instructions of one SIMD overlap (gfx950)?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu.hip -o tools/ubench/mfma_valu && tools/ubench/mfma_valu
// Per loop iteration a wave issues NV v_fmac_f32 (8 independent chains) and / or NM
// v_mfma_f32_16x16x4_f32 (4 independent accumulators); modes:
//   0  vector only              1  matrix only
//   2  both, interleaved in EVERY wave (one MFMA, then NV / NM vector instructions, ...)
//   3  both, split by wave: the even waves of a SIMD run the vector part, the odd ones the matrix part
// If the two pipes overlap, (2) and (3) take max(t0, t1); if the MFMA occupies the vector
// unit's multipliers, t0 + t1.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
#define V8 asm volatile("v_fmac_f32_e32 %0, %8, %9\n v_fmac_f32_e32 %1, %8, %9\n v_fmac_f32_e32 %2, %8, %9\n v_fmac_f32_e32 %3, %8, %9\n" \
                        "v_fmac_f32_e32 %4, %8, %9\n v_fmac_f32_e32 %5, %8, %9\n v_fmac_f32_e32 %6, %8, %9\n v_fmac_f32_e32 %7, %8, %9\n" \
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(r), "v"(x));
#define M1(c) c = __builtin_amdgcn_mfma_f32_16x16x4f32(r, x, c, 0, 0, 0);
template <int MODE, int KIND>
__global__ void __launch_bounds__(256) k(float *out, int iters) {
    float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f, a4 = 4.f, a5 = 5.f, a6 = 6.f, a7 = 7.f;
    float r = threadIdx.x * 0.5f, x = 1.0001f;
    f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    // (blocks are dealt round-robin to CUs; a 256-thread block puts one wave on each SIMD, so block
    // parity = wave parity within a SIMD when two blocks share a CU)
    const bool matrix_wave = (blockIdx.x / 256) & 1;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0 || (MODE == 3 && !matrix_wave)) {
            V8 V8 V8 V8 V8 V8 V8 V8          // 64 vector instructions
        } else if (MODE == 1 || (MODE == 3 && matrix_wave)) {
            M1(c0) M1(c1) M1(c2) M1(c3)      // KIND 0: 4 MFMAs = 128 matrix cycles = the 64 vector instructions' 128 cycles
            if (KIND == 1) { M1(c0) M1(c1) M1(c2) M1(c3) }
        } else {
            M1(c0) V8 V8 M1(c1) V8 V8 M1(c2) V8 V8 M1(c3) V8 V8
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + c0[0] + c1[1] + c2[2] + c3[3];
}
template <int MODE>
float run(const char *name, int waves_per_simd) {
    float *d;
    const int blocks = 256 * waves_per_simd, iters = 4000;
    hipMalloc(&d, sizeof(float) * blocks * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<MODE, 0><<<blocks, 256>>>(d, 10);
    hipEventRecord(e0);
    k<MODE, 0><<<blocks, 256>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %d waves/SIMD: %.3f ms  (%.1f cycles per loop iteration and wave at 2.4 GHz)\n", name, waves_per_simd, ms,
           ms * 1e-3 * 2.4e9 / iters / 1.);
    hipFree(d);
    return ms;
}
int main() {
    for (int w : {2, 4}) {
        run<0>("0 vector only (64 v_fmac_f32 per iteration)", w);
        run<1>("1 matrix only (4 v_mfma_f32_16x16x4_f32)", w);
        run<2>("2 both, interleaved in every wave", w);
        run<3>("3 both, vector waves and matrix waves", w);
    }
    return 0;
}
