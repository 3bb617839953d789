// Achievable HBM WRITE bandwidth on this chip: write-only streams (plain / nontemporal stores), and a copy with plain
// loads + nontemporal stores -- the access shapes of tdx_encode_fwd, the up-sampling kernel and the GroupNorm apply pass.
// Buffers of 453 MB (= level-0 64-channel bf16 activation at B = 6).   hipcc -O3 --offload-arch=gfx950 bw_write_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ntload(const uint4* p) { u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)); return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void ntstore(uint4 r, uint4* p) { u32x4 v = {r.x, r.y, r.z, r.w}; __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p)); }
template <bool NT>
__global__ void __launch_bounds__(256) k_write(uint4* __restrict__ c, size_t n, unsigned seed) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const uint4 r = make_uint4((unsigned)i, seed, (unsigned)(i >> 7), seed ^ (unsigned)i);
        if (NT) ntstore(r, c + i); else c[i] = r;
    }
}
template <int LD, int ST>  // 0 plain, 1 nontemporal
__global__ void __launch_bounds__(256) k_copy(const uint4* __restrict__ a, uint4* __restrict__ c, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    for (size_t i = (size_t)blockIdx.x * 256 * 4 + threadIdx.x; i < n; i += stride) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const size_t j = i + (size_t)u * 256; if (j < n) v[u] = LD ? ntload(a + j) : a[j]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) { const size_t j = i + (size_t)u * 256; if (j < n) { uint4 r = v[u]; r.x ^= 1u; if (ST) ntstore(r, c + j); else c[j] = r; } }
    }
}
template <typename F>
static float timeit(F f, int reps = 10) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) f();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}
int main() {
    const size_t bytes = (size_t)6 * 192 * 64 * 48 * 64 * 2, n = bytes / 16;
    uint4 *a, *c;
    (void)hipMalloc(&a, bytes); (void)hipMalloc(&c, bytes);
    (void)hipMemset(a, 1, bytes);
    for (int blocks : {2048, 8192, 32768}) {
        float w0 = timeit([&] { hipLaunchKernelGGL((k_write<false>), dim3(blocks), dim3(256), 0, 0, c, n, 7u); });
        float w1 = timeit([&] { hipLaunchKernelGGL((k_write<true>), dim3(blocks), dim3(256), 0, 0, c, n, 7u); });
        float c00 = timeit([&] { hipLaunchKernelGGL((k_copy<0, 0>), dim3(blocks), dim3(256), 0, 0, a, c, n); });
        float c10 = timeit([&] { hipLaunchKernelGGL((k_copy<1, 0>), dim3(blocks), dim3(256), 0, 0, a, c, n); });
        float c01 = timeit([&] { hipLaunchKernelGGL((k_copy<0, 1>), dim3(blocks), dim3(256), 0, 0, a, c, n); });
        float c11 = timeit([&] { hipLaunchKernelGGL((k_copy<1, 1>), dim3(blocks), dim3(256), 0, 0, a, c, n); });
        printf("blocks %5d: 1W plain %5.0f us %5.2f TB/s | 1W nt %5.0f us %5.2f | copy ld/st plain/plain %5.0f us %5.2f | nt/plain %5.0f us %5.2f | plain/nt %5.0f us %5.2f | nt/nt %5.0f us %5.2f\n",
               blocks, w0 * 1e3, bytes / w0 / 1e9, w1 * 1e3, bytes / w1 / 1e9, c00 * 1e3, 2 * bytes / c00 / 1e9, c10 * 1e3, 2 * bytes / c10 / 1e9,
               c01 * 1e3, 2 * bytes / c01 / 1e9, c11 * 1e3, 2 * bytes / c11 / 1e9);
    }
    return 0;
}
