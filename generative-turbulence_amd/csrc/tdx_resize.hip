// Trilinear resize, align_corners=True, NDHWC (ddpm.py:359-361, 367-369).
// HBM-bound gather kernels: one lane owns 8 channels of one voxel (16 B bf16 / 32 B f32).
// Index arithmetic follows ATen's upsample_trilinear3d exactly (float scale, float source
// index, truncation, clamped second tap) so that tap selection matches the reference.
#include "tdx_common.h"

struct AxisMap {
    int in, out;
    float scale;
};
__host__ __device__ inline AxisMap make_axis(int in, int out) {
    AxisMap a;
    a.in = in; a.out = out;
    a.scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.0f;
    return a;
}
// taps of output index o: i0, i1, weight of i1
__device__ __forceinline__ void axis_taps(const AxisMap& a, int o, int& i0, int& i1, float& w1) {
    const float src = a.scale * (float)o;
    i0 = min((int)src, a.in - 1);
    w1 = fminf(fmaxf(src - (float)i0, 0.0f), 1.0f);
    i1 = i0 + ((i0 + 1 < a.in) ? 1 : 0);
}

template <typename T>
__global__ void __launch_bounds__(256)
resize_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, AxisMap ax, AxisMap ay, AxisMap az, int C, int64_t total) {
    const int L = C >> 3;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int lc = (int)(i % L);
    int64_t v = i / L;
    const int oz = (int)(v % az.out); v /= az.out;
    const int oy = (int)(v % ay.out); v /= ay.out;
    const int ox = (int)(v % ax.out);
    const int b = (int)(v / ax.out);
    int x0, x1, y0, y1, z0, z1;
    float wx, wy, wz;
    axis_taps(ax, ox, x0, x1, wx);
    axis_taps(ay, oy, y0, y1, wy);
    axis_taps(az, oz, z0, z1, wz);
    const T* xb = x + ((int64_t)b * ax.in * ay.in * az.in) * C + lc * 8;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    const int xs[2] = {x0, x1}, ys[2] = {y0, y1}, zs[2] = {z0, z1};
    const float wxs[2] = {1.0f - wx, wx}, wys[2] = {1.0f - wy, wy}, wzs[2] = {1.0f - wz, wz};
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int bb = 0; bb < 2; ++bb)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float w = wxs[a] * wys[bb] * wzs[c];
                Vec8<T> t;
                t.load(xb + (((int64_t)xs[a] * ay.in + ys[bb]) * az.in + zs[c]) * C);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += w * t.v[j];
            }
    Vec8<T> o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o.v[j] = acc[j];
    o.store(y + ((((int64_t)b * ax.out + ox) * ay.out + oy) * az.out + oz) * C + lc * 8);
}

// range [lo, hi] of outputs that can read input index i (a superset; weights decide)
__device__ __forceinline__ void axis_range(const AxisMap& a, int i, int& lo, int& hi) {
    if (a.scale > 0.f) {
        lo = max(0, (int)floorf((float)(i - 1) / a.scale) - 1);
        hi = min(a.out - 1, (int)ceilf((float)(i + 1) / a.scale) + 1);
    } else {
        lo = 0; hi = a.out - 1;  // out == 1 or in == 1: every output reads input 0
    }
}
// weight with which output o reads input i (adjoint of axis_taps); 0 if it does not
__device__ __forceinline__ float axis_weight(const AxisMap& a, int o, int i) {
    int i0, i1;
    float w1;
    axis_taps(a, o, i0, i1, w1);
    float w = 0.f;
    if (i0 == i) w += 1.0f - w1;
    if (i1 == i) w += w1;
    return w;
}

// adjoint gather: dx[i] = sum over outputs o of w(o, i) dy[o]; weights are recomputed on the
// fly (a handful of VALU ops) instead of being tabulated, so nothing lives in scratch
template <typename T>
__global__ void __launch_bounds__(256)
resize_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, AxisMap ax, AxisMap ay, AxisMap az, int C, int64_t total) {
    const int L = C >> 3;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int lc = (int)(i % L);
    int64_t v = i / L;
    const int iz = (int)(v % az.in); v /= az.in;
    const int iy = (int)(v % ay.in); v /= ay.in;
    const int ix = (int)(v % ax.in);
    const int b = (int)(v / ax.in);
    int x0, x1, y0, y1, z0, z1;
    axis_range(ax, ix, x0, x1);
    axis_range(ay, iy, y0, y1);
    axis_range(az, iz, z0, z1);
    const T* gb = dy + ((int64_t)b * ax.out * ay.out * az.out) * C + lc * 8;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int ox = x0; ox <= x1; ++ox) {
        const float wx = axis_weight(ax, ox, ix);
        if (wx == 0.f) continue;
        for (int oy = y0; oy <= y1; ++oy) {
            const float wxy = wx * axis_weight(ay, oy, iy);
            if (wxy == 0.f) continue;
            const T* row = gb + (((int64_t)ox * ay.out + oy) * az.out) * C;
            for (int oz = z0; oz <= z1; ++oz) {
                const float w = wxy * axis_weight(az, oz, iz);
                if (w == 0.f) continue;
                Vec8<T> t;
                t.load(row + (int64_t)oz * C);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += w * t.v[j];
            }
        }
    }
    Vec8<T> o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o.v[j] = acc[j];
    o.store(dx + ((((int64_t)b * ax.in + ix) * ay.in + iy) * az.in + iz) * C + lc * 8);
}

static int resize_args_ok(int B, int Xi, int Yi, int Zi, int Xo, int Yo, int Zo, int C) {
    return B > 0 && Xi > 0 && Yi > 0 && Zi > 0 && Xo > 0 && Yo > 0 && Zo > 0 && C > 0;
}
extern "C" int tdx_resize_fwd(const void* x, void* y, int B, int Xi, int Yi, int Zi, int Xo, int Yo, int Zo, int C,
                              int dtype, void* stream) {
    TDX_CHECK_ARG(x && y && resize_args_ok(B, Xi, Yi, Zi, Xo, Yo, Zo, C));
    if (C % 8) return TDX_ESHAPE;
    const int64_t total = (int64_t)B * Xo * Yo * Zo * (C / 8);
    TDX_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((resize_fwd_kernel<T>), dim3(ceil_div(total, 256)), dim3(256), 0,
                                                  as_stream(stream), (const T*)x, (T*)y, make_axis(Xi, Xo),
                                                  make_axis(Yi, Yo), make_axis(Zi, Zo), C, total));
    return tdx_launch_status();
}

extern "C" int tdx_resize_bwd(const void* dy, void* dx, int B, int Xi, int Yi, int Zi, int Xo, int Yo, int Zo, int C,
                              int dtype, void* stream) {
    TDX_CHECK_ARG(dy && dx && resize_args_ok(B, Xi, Yi, Zi, Xo, Yo, Zo, C));
    if (C % 8) return TDX_ESHAPE;
    const int64_t total = (int64_t)B * Xi * Yi * Zi * (C / 8);
    TDX_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((resize_bwd_kernel<T>), dim3(ceil_div(total, 256)), dim3(256), 0,
                                                  as_stream(stream), (const T*)dy, (T*)dx, make_axis(Xi, Xo),
                                                  make_axis(Yi, Yo), make_axis(Zi, Zo), C, total));
    return tdx_launch_status();
}
