// Probe of gfx950's global_load_lds_dwordx4: every lane loads 16 B from its own
// global address straight into LDS (no VGPR round trip); the data of lane l of
// a wave lands at M0 base + 16 l.  Checks that layout against a plain gather.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/lds_gather.hip -o /tmp/lds_gather && /tmp/lds_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k(const float4 *__restrict__ src, const int *__restrict__ idx, float4 *__restrict__ out) {
    __shared__ float4 buf[2][256];
    const int t = threadIdx.x;
    for (int r = 0; r < 2; ++r) {
        const int i = idx[(blockIdx.x * 2 + r) * 256 + t];
        __builtin_amdgcn_global_load_lds(
            (const void __attribute__((address_space(1))) *)(src + i),
            (void __attribute__((address_space(3))) *)(&buf[r][t & ~63]), 16, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0)
    __syncthreads();
    for (int r = 0; r < 2; ++r) out[(blockIdx.x * 2 + r) * 256 + t] = buf[r][t];
}

int main() {
    const int n = 1 << 16, nb = 8;
    std::vector<float4> h(n);
    for (int i = 0; i < n; ++i) h[i] = make_float4(i, i + 0.25f, i + 0.5f, i + 0.75f);
    std::vector<int> hi(nb * 512);
    for (size_t q = 0; q < hi.size(); ++q) hi[q] = (int)((q * 2654435761u) % n);
    float4 *d, *o;
    int *di;
    hipMalloc(&d, n * sizeof(float4));
    hipMalloc(&o, hi.size() * sizeof(float4));
    hipMalloc(&di, hi.size() * sizeof(int));
    hipMemcpy(d, h.data(), n * sizeof(float4), hipMemcpyHostToDevice);
    hipMemcpy(di, hi.data(), hi.size() * sizeof(int), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, d, di, o);
    std::vector<float4> ho(hi.size());
    hipMemcpy(ho.data(), o, hi.size() * sizeof(float4), hipMemcpyDeviceToHost);
    int bad = 0;
    for (size_t q = 0; q < hi.size(); ++q)
        if (ho[q].x != h[hi[q]].x || ho[q].w != h[hi[q]].w) ++bad;
    printf("global_load_lds_dwordx4 gather: %d mismatches of %zu (%s)\n", bad, hi.size(), hipGetErrorString(hipGetLastError()));
    return bad != 0;
}
