// Sample metrics on the device: the turbulent-kinetic-energy spectrum of the
// reference's TurbulentKineticEnergySpectrum.forward (turbdiff/models/metrics.py:289-316) around the FFT
// (which stays rocFFT, via torch.fft.fftn):
//   tdx_tke_energy : tke = 0.5 * sum_c u_c^2                                       (metrics.py:291)
//   tdx_tke_sphere : E[b, k] = 4 pi k^2 sum_n w_n exp(interp3(log |F|^2, k p_n + c)) (metrics.py:294-313)
// The reference materialises fftshift(F), |F|^2, its log, eight (B, K, N) gathers, their weighted sum, the
// exp and a matmul; here one kernel reads the UNSHIFTED complex spectrum at the 8 corners of every query
// point (the shift is index arithmetic, the log-power is computed at the corner), interpolates in the
// reference's term order, exponentiates and reduces over the quadrature points.  The spectrum of one
// sample (46^3 complex64 = 0.8 MB for the reference's regions) stays in L2; the kernel is latency /
// transcendental-bound, not HBM-bound.
#include "tdx_common.h"

__global__ void __launch_bounds__(256)
tke_energy_kernel(const float* __restrict__ u, float* __restrict__ tke, int64_t V) {
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= V) return;
    const float* ub = u + (int64_t)blockIdx.y * 3 * V + v;
    const float x = ub[0], y = ub[V], z = ub[2 * V];
    tke[(int64_t)blockIdx.y * V + v] = 0.5f * (__fmul_rn(x, x) + __fmul_rn(y, y) + __fmul_rn(z, z));
}

struct SphereGrid { int n[3]; };

__device__ __forceinline__ float log_power(const float2* __restrict__ f, SphereGrid g, int jx, int jy, int jz) {
    // shifted index j -> unshifted index (j - n/2) mod n   (torch.fft.fftshift rolls by n // 2)
    const int ix = (jx - g.n[0] / 2 + g.n[0]) % g.n[0], iy = (jy - g.n[1] / 2 + g.n[1]) % g.n[1],
              iz = (jz - g.n[2] / 2 + g.n[2]) % g.n[2];
    const float2 c = f[((int64_t)ix * g.n[1] + iy) * g.n[2] + iz];
    const float a = hypotf(c.x, c.y);
    return logf(a * a);
}

__global__ void __launch_bounds__(256)
tke_sphere_kernel(const float2* __restrict__ fft, const float* __restrict__ p, const float* __restrict__ w,
                  const float* __restrict__ k, float* __restrict__ E, SphereGrid g, int N, int K) {
    __shared__ double red[4];
    const int ki = blockIdx.x, b = blockIdx.y;
    const float kk = k[ki];
    const float2* f = fft + (int64_t)b * g.n[0] * g.n[1] * g.n[2];
    double acc = 0.0;
    for (int n = threadIdx.x; n < N; n += 256) {
        float q[3], wt[3];
        int j0[3], j1[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            q[a] = __fadd_rn(__fmul_rn(kk, p[n * 3 + a]), (float)(g.n[a] / 2));
            const int fl = (int)floorf(q[a]);
            j0[a] = min(max(fl, 0), g.n[a] - 1);
            j1[a] = min(max(fl + 1, 0), g.n[a] - 1);
            wt[a] = q[a] - (float)j0[a];
        }
        const float wx = wt[0], wy = wt[1], wz = wt[2];
        float v = (1 - wx) * (1 - wy) * (1 - wz) * log_power(f, g, j0[0], j0[1], j0[2]);
        v += (1 - wx) * (1 - wy) * wz * log_power(f, g, j0[0], j0[1], j1[2]);
        v += (1 - wx) * wy * (1 - wz) * log_power(f, g, j0[0], j1[1], j0[2]);
        v += (1 - wx) * wy * wz * log_power(f, g, j0[0], j1[1], j1[2]);
        v += wx * (1 - wy) * (1 - wz) * log_power(f, g, j1[0], j0[1], j0[2]);
        v += wx * (1 - wy) * wz * log_power(f, g, j1[0], j0[1], j1[2]);
        v += wx * wy * (1 - wz) * log_power(f, g, j1[0], j1[1], j0[2]);
        v += wx * wy * wz * log_power(f, g, j1[0], j1[1], j1[2]);
        acc += (double)(expf(v) * w[n]);
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float s = (float)(red[0] + red[1] + red[2] + red[3]);
        E[(int64_t)b * K + ki] = s * (4.0f * 3.14159265358979323846f * kk * kk);
    }
}

extern "C" int tdx_tke_energy(const float* u, float* tke, int B, int64_t V, void* stream) {
    TDX_CHECK_ARG(u && tke && B > 0 && V > 0);
    if (B > 65535) return TDX_ESHAPE;
    hipLaunchKernelGGL(tke_energy_kernel, dim3((unsigned)ceil_div(V, (int64_t)256), B), dim3(256), 0, as_stream(stream), u, tke, V);
    return tdx_launch_status();
}

extern "C" int tdx_tke_sphere(const float* fft, const float* p, const float* w, const float* k, float* E, int B, int X, int Y,
                              int Z, int N, int K, void* stream) {
    TDX_CHECK_ARG(fft && p && w && k && E && B > 0 && X > 0 && Y > 0 && Z > 0 && N > 0 && K > 0);
    if (B > 65535 || (int64_t)X * Y * Z >= (1ll << 31)) return TDX_ESHAPE;
    SphereGrid g = {{X, Y, Z}};
    hipLaunchKernelGGL(tke_sphere_kernel, dim3(K, B), dim3(256), 0, as_stream(stream), (const float2*)fft, p, w, k, E, g, N, K);
    return tdx_launch_status();
}
