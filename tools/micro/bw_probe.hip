// Achievable HBM streaming bandwidth on this chip for the access shapes of the normalisation kernels:
// read-only reduction, copy (1R + 1W), 2R + 1W.  Buffers of 453 MB (= level-0 64-channel bf16 activation at B = 6).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ntload(const uint4* p) { u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)); return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void ntstore(uint4 r, uint4* p) { u32x4 v = {r.x, r.y, r.z, r.w}; __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p)); }
template <int UNROLL, bool NT>
__global__ void __launch_bounds__(256) k_read(const uint4* __restrict__ a, const uint4* __restrict__ b, size_t n, float* out) {
    float s = 0.f;
    const size_t stride = (size_t)gridDim.x * 256 * UNROLL;
    for (size_t i = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x; i < n; i += stride) {
        uint4 v[UNROLL], w[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const size_t j = i + (size_t)u * 256;
            if (j < n) {
                v[u] = NT ? ntload(a + j) : a[j];
                if (b) w[u] = NT ? ntload(b + j) : b[j];
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { s += __uint_as_float(v[u].x ^ v[u].w); if (b) s += __uint_as_float(w[u].y); }
    }
    if (s == 1.2345f) out[0] = s;
}
template <int UNROLL, bool NT>
__global__ void __launch_bounds__(256) k_copy(const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* __restrict__ c, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256 * UNROLL;
    for (size_t i = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x; i < n; i += stride) {
        uint4 v[UNROLL], w[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const size_t j = i + (size_t)u * 256;
            if (j < n) { v[u] = NT ? ntload(a + j) : a[j]; if (b) w[u] = NT ? ntload(b + j) : b[j]; }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const size_t j = i + (size_t)u * 256;
            if (j < n) {
                uint4 r = v[u];
                if (b) { r.x ^= w[u].x; r.y += w[u].y; }
                if (NT) ntstore(r, c + j); else c[j] = r;
            }
        }
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
    uint4 *a, *b, *c; float* out;
    (void)hipMalloc(&a, bytes); (void)hipMalloc(&b, bytes); (void)hipMalloc(&c, bytes); (void)hipMalloc(&out, 4);
    (void)hipMemset(a, 1, bytes); (void)hipMemset(b, 2, bytes);
    for (int blocks : {1024, 2048, 4096, 8192, 32768}) {
#define RUN(U, NTV)                                                                                                      \
    {                                                                                                                    \
        float t1 = timeit([&] { hipLaunchKernelGGL((k_read<U, NTV>), dim3(blocks), dim3(256), 0, 0, a, (const uint4*)nullptr, n, out); }); \
        float t2 = timeit([&] { hipLaunchKernelGGL((k_read<U, NTV>), dim3(blocks), dim3(256), 0, 0, a, b, n, out); });    \
        float t3 = timeit([&] { hipLaunchKernelGGL((k_copy<U, NTV>), dim3(blocks), dim3(256), 0, 0, a, (const uint4*)nullptr, c, n); }); \
        float t4 = timeit([&] { hipLaunchKernelGGL((k_copy<U, NTV>), dim3(blocks), dim3(256), 0, 0, a, b, c, n); });      \
        printf("blocks %5d unroll %d nt %d: 1R %6.0f us %5.2f TB/s | 2R %6.0f us %5.2f | 1R+1W %6.0f us %5.2f | 2R+1W %6.0f us %5.2f\n", blocks, U, \
               (int)NTV, t1 * 1e3, bytes / t1 / 1e9, t2 * 1e3, 2 * bytes / t2 / 1e9, t3 * 1e3, 2 * bytes / t3 / 1e9, t4 * 1e3, 3 * bytes / t4 / 1e9); \
    }
        RUN(1, false) RUN(4, false) RUN(8, false) RUN(4, true) RUN(8, true)
    }
    return 0;
}
