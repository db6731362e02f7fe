// Stream-rate sweep for one MI355X: what does a plain 16 B / lane stream
// reach on THIS pool, as a function of launch shape, loads in flight, cache policy and mode?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/stream_sweep.hip -o tools/ubench/stream_sweep
//   tools/ubench/stream_sweep [GiB per buffer = 2] > profiles/r04_stream_sweep.txt
// Rates are (bytes read + bytes written) / time, HIP events around `reps` back-to-back launches
// (sustained: clocks settled), best and mean of 3 such groups.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

enum Mode { COPY = 0, READ = 1, WRITE = 2 };

template <int U, int MODE, bool NT, bool GRIDSTRIDE>
__global__ __launch_bounds__(256) void k_stream(const f4 *__restrict__ in, f4 *__restrict__ out,
                                                int64_t n, f4 *sink) {
    // GRIDSTRIDE: element i of pass p is p * (grid * 256 * U) + u * (grid * 256) + tid  (every
    // load instruction of a wave is one contiguous 1 KiB, U of them in flight per lane);
    // else: one contiguous block of 256 * U elements per workgroup (exact grid).
    const int64_t nthr = (int64_t)gridDim.x * 256;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    if (GRIDSTRIDE) {
        for (int64_t base = (int64_t)blockIdx.x * 256 + threadIdx.x; base < n; base += nthr * U) {
            f4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t i = base + u * nthr;
                if (MODE != WRITE) {
                    if (i < n) v[u] = NT ? __builtin_nontemporal_load(in + i) : in[i];
                    else v[u] = acc;
                } else v[u] = f4{(float)i, 1.f, 2.f, 3.f};
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t i = base + u * nthr;
                if (MODE == READ) acc += v[u];
                else if (i < n) { if (NT) __builtin_nontemporal_store(v[u], out + i); else out[i] = v[u]; }
            }
        }
    } else {
        const int64_t b0 = (int64_t)blockIdx.x * 256 * U + threadIdx.x;
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = b0 + u * 256;
            if (MODE != WRITE) {
                if (i < n) v[u] = NT ? __builtin_nontemporal_load(in + i) : in[i];
                else v[u] = acc;
            } else v[u] = f4{(float)i, 1.f, 2.f, 3.f};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = b0 + u * 256;
            if (MODE == READ) acc += v[u];
            else if (i < n) { if (NT) __builtin_nontemporal_store(v[u], out + i); else out[i] = v[u]; }
        }
    }
    if (MODE == READ && acc.x == 1.2345e-30f) *sink = acc;
}

struct Result { const char *mode; int U; int nt; const char *shape; int64_t grid; double best, mean; };

template <int U, int MODE, bool NT, bool GS>
static Result run(const f4 *in, f4 *out, int64_t n, int64_t grid, f4 *sink, const char *shape) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 12;
    double best = 0, sum = 0;
    const double bytes = (double)n * 16 * (MODE == COPY ? 2 : 1);
    for (int g = 0; g < 4; ++g) {       // group 0 = warm-up
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; ++r)
            hipLaunchKernelGGL((k_stream<U, MODE, NT, GS>), dim3((unsigned)grid), dim3(256), 0, 0, in, out, n, sink);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double rate = bytes * reps / (ms * 1e-3) / 1e12;
        if (g) { best = std::max(best, rate); sum += rate; }
    }
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    static const char *names[] = {"copy", "read", "write"};
    return Result{names[MODE], U, NT ? 1 : 0, shape, grid, best, sum / 3};
}

template <int U, int MODE, bool NT>
static void sweep_shapes(const f4 *in, f4 *out, int64_t n, f4 *sink, std::vector<Result> &res) {
    for (int64_t grid : {1024, 2048, 4096, 8192, 16384, 65536})
        res.push_back(run<U, MODE, NT, true>(in, out, n, grid, sink, "grid-stride"));
    const int64_t exact = (n + 256 * U - 1) / (256 * U);
    res.push_back(run<U, MODE, NT, false>(in, out, n, exact, sink, "exact"));
}

template <int MODE, bool NT>
static void sweep_u(const f4 *in, f4 *out, int64_t n, f4 *sink, std::vector<Result> &res) {
    sweep_shapes<1, MODE, NT>(in, out, n, sink, res);
    sweep_shapes<2, MODE, NT>(in, out, n, sink, res);
    sweep_shapes<4, MODE, NT>(in, out, n, sink, res);
    sweep_shapes<8, MODE, NT>(in, out, n, sink, res);
    sweep_shapes<16, MODE, NT>(in, out, n, sink, res);
}

int main(int argc, char **argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 2.0;
    const int64_t nbytes = (int64_t)(gib * (1ll << 30)) & ~((int64_t)(2 << 20) - 1);
    const int64_t n = nbytes / 16;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("# device %s, %d CUs, mem clock %d kHz, bus %d bit; buffers %.2f GiB each (2 MiB aligned: %s)\n",
           prop.name, prop.multiProcessorCount, prop.memoryClockRate, prop.memoryBusWidth, gib, "hipMalloc");
    f4 *in, *out, *sink;
    CK(hipMalloc(&in, nbytes)); CK(hipMalloc(&out, nbytes)); CK(hipMalloc(&sink, 64));
    printf("# in %p out %p\n", (void *)in, (void *)out);
    CK(hipMemset(in, 1, nbytes)); CK(hipMemset(out, 0, nbytes));
    // the runtime's own device copy for reference
    {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int g = 0; g < 3; ++g) {
            CK(hipEventRecord(e0, 0));
            for (int r = 0; r < 8; ++r) CK(hipMemcpyAsync(out, in, nbytes, hipMemcpyDeviceToDevice, 0));
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (g) printf("# hipMemcpyAsync D2D: %.3f TB/s (read + write)\n", 2.0 * nbytes * 8 / (ms * 1e-3) / 1e12);
        }
    }
    std::vector<Result> res;
    sweep_u<COPY, false>(in, out, n, sink, res);
    sweep_u<COPY, true>(in, out, n, sink, res);
    sweep_u<READ, false>(in, out, n, sink, res);
    sweep_u<READ, true>(in, out, n, sink, res);
    sweep_u<WRITE, false>(in, out, n, sink, res);
    sweep_u<WRITE, true>(in, out, n, sink, res);
    printf("%-6s %3s %3s %-12s %8s %9s %9s\n", "mode", "U", "nt", "shape", "grid", "best TB/s", "mean TB/s");
    for (auto &r : res)
        printf("%-6s %3d %3d %-12s %8lld %9.3f %9.3f\n", r.mode, r.U, r.nt, r.shape, (long long)r.grid, r.best, r.mean);
    for (const char *m : {"copy", "read", "write"}) {
        const Result *b = nullptr;
        for (auto &r : res) if (r.mode == m || std::string(r.mode) == m) if (!b || r.best > b->best) b = &r;
        if (b) printf("# best %s: %.3f TB/s (U=%d nt=%d %s grid=%lld)\n", m, b->best, b->U, b->nt, b->shape, (long long)b->grid);
    }
    return 0;
}
