// Write-pattern probe for the record planes (round 4): P float64 planes, work items of 256
// records (2 KiB per plane), persistent workgroups walking the items the way k_fflux /
// k_derive do (chunk-major numbering, every XCD owns whole chunks).  What differs is WHERE an
// item's records go:
//   star-major  : slot = star * per_star + chunk * per_seg + piece * 256   (round 3: the lists are
//                 per star, so workgroups that run together write 2 KiB pieces ~1 MiB apart)
//   chunk-major : slot = item * 256                                        (workgroups that run
//                 together write neighbouring pieces of every plane)
// and how the items are taken (persistent XCD walk / one workgroup per item in launch order).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/plane_writes.hip -o tools/ubench/plane_writes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int NCHUNK = 64;

// items numbered chunk-major: item = (c * nstar + s) * pieces + piece
template <int P, bool CHUNK_MAJOR, bool PERSIST, bool NT, int VALU>
__global__ __launch_bounds__(256) void k_planes(double *__restrict__ vals, int64_t cap, int nstar, int pieces,
                                                int64_t per_star, int64_t per_seg) {
    const int per_chunk = nstar * pieces;
    auto body = [&](int item) {
        const int c = item / per_chunk, r = item - c * per_chunk;
        const int s = r / pieces, piece = r - s * pieces;
        const int64_t slot = CHUNK_MAJOR ? (int64_t)item * 256 + threadIdx.x
                                         : (int64_t)s * per_star + (int64_t)c * per_seg + (int64_t)piece * 256 + threadIdx.x;
        double x = (double)slot;
#pragma unroll 1
        for (int k = 0; k < VALU; ++k) x = fma(x, 1.0000001, 0.5);     // stand-in for the MLE
#pragma unroll
        for (int v = 0; v < P; ++v) {
            double *p = vals + (int64_t)v * cap + slot;
            if (NT) __builtin_nontemporal_store(x + v, p); else *p = x + v;
        }
    };
    if (PERSIST) {
        const int xcd = blockIdx.x & 7, me = blockIdx.x >> 3, nw = (gridDim.x + 7 - xcd) >> 3;
        for (int c = xcd; c < NCHUNK; c += 8)
            for (int item = c * per_chunk + me; item < (c + 1) * per_chunk; item += nw) body(item);
    } else {
        body(blockIdx.x);
    }
}

template <int P, bool CM, bool PERSIST, bool NT, int VALU>
static void run(const char *tag, double *vals, int64_t cap, int nstar, int pieces, int64_t per_star, int64_t per_seg) {
    const int nitem = NCHUNK * nstar * pieces;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double best = 1e30, sum = 0;
    for (int g = 0; g < 4; ++g) {
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < 10; ++r)
            hipLaunchKernelGGL((k_planes<P, CM, PERSIST, NT, VALU>), dim3(PERSIST ? 4096 : nitem), dim3(256), 0, 0,
                               vals, cap, nstar, pieces, per_star, per_seg);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (g) { best = std::min(best, (double)ms / 10); sum += ms / 10; }
    }
    const double bytes = (double)nitem * 256 * 8 * P;
    printf("%-44s P=%2d valu=%4d  %.3f ms (mean %.3f)  %.2f TB/s\n", tag, P, VALU, best, sum / 3, bytes / (best * 1e-3) / 1e12);
}

int main() {
    const int nstar = 128, pieces = 10;            // 128 x 64 x 10 items of 256 records = 21 M records
    const int64_t per_seg = (int64_t)pieces * 256, per_star = per_seg * NCHUNK;
    const int64_t cap = (int64_t)nstar * per_star;
    double *vals; CK(hipMalloc(&vals, sizeof(double) * cap * 13));
    CK(hipMemset(vals, 0, sizeof(double) * cap * 13));
    printf("# %lld records, planes of %.1f MB\n", (long long)cap, cap * 8 / 1e6);
#define ROW(P, V) \
    run<P, false, true, false, V>("star-major slots, persistent XCD walk", vals, cap, nstar, pieces, per_star, per_seg); \
    run<P, true, true, false, V>("chunk-major slots, persistent XCD walk", vals, cap, nstar, pieces, per_star, per_seg); \
    run<P, true, true, true, V>("chunk-major slots, persistent, nt stores", vals, cap, nstar, pieces, per_star, per_seg); \
    run<P, false, false, false, V>("star-major slots, one workgroup per item", vals, cap, nstar, pieces, per_star, per_seg); \
    run<P, true, false, false, V>("chunk-major slots, one workgroup per item", vals, cap, nstar, pieces, per_star, per_seg); \
    run<P, true, false, true, V>("chunk-major slots, wg per item, nt stores", vals, cap, nstar, pieces, per_star, per_seg);
    ROW(10, 0)
    ROW(13, 0)
    ROW(10, 200)
    ROW(1, 0)
    ROW(2, 0)
    return 0;
}
