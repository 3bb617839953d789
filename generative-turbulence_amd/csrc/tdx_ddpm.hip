// DDPM noise / denoise arithmetic, masked loss, layout transforms and the Philox generator.
// All HBM-bound streaming kernels: 16 B per lane, grid-stride, no LDS needed except for
// the (B,C,V) <-> (B,V,C) transposes which go through an LDS tile so that both sides are
// coalesced.
#include "tdx_common.h"
#include "tdx_conv3.h"  // tdx_deterministic

extern "C" int tdx_version(void) { return 1; }
extern "C" const char* tdx_arch(void) { return "gfx950"; }

// ------------------------------------------------------------------ layout ---------------
// (B, C, V) -> (B, V, C): tile of 64 voxels x C channels (C <= 64 per pass) through LDS.
template <typename TI, typename TO>
__global__ void __launch_bounds__(256) ncv_to_nvc_kernel(const TI* __restrict__ x, TO* __restrict__ y, int C, int64_t V) {
    __shared__ float tile[64][65];
    const int b = blockIdx.z;
    const int c0 = blockIdx.y * 64;
    const int64_t v0 = (int64_t)blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
    const int nc = min(64, C - c0);
    for (int c = ty; c < nc; c += 4) {
        int64_t v = v0 + tx;
        if (v < V) tile[c][tx] = ldf(x + ((int64_t)b * C + c0 + c) * V + v);
    }
    __syncthreads();
    // write: consecutive threads -> consecutive channels of one voxel
    for (int i = threadIdx.x; i < 64 * nc; i += 256) {
        int vv = i / nc, c = i - vv * nc;
        int64_t v = v0 + vv;
        if (v < V) stf(y + ((int64_t)b * V + v) * C + c0 + c, tile[c][vv]);
    }
}

template <typename TI, typename TO>
__global__ void __launch_bounds__(256) nvc_to_ncv_kernel(const TI* __restrict__ x, TO* __restrict__ y, int C, int64_t V) {
    __shared__ float tile[64][65];
    const int b = blockIdx.z;
    const int c0 = blockIdx.y * 64;
    const int64_t v0 = (int64_t)blockIdx.x * 64;
    const int nc = min(64, C - c0);
    for (int i = threadIdx.x; i < 64 * nc; i += 256) {
        int vv = i / nc, c = i - vv * nc;
        int64_t v = v0 + vv;
        if (v < V) tile[c][vv] = ldf(x + ((int64_t)b * V + v) * C + c0 + c);
    }
    __syncthreads();
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int c = ty; c < nc; c += 4) {
        int64_t v = v0 + tx;
        if (v < V) stf(y + ((int64_t)b * C + c0 + c) * V + v, tile[c][tx]);
    }
}

template <typename TI, typename TO>
__global__ void cast_kernel(const TI* __restrict__ x, TO* __restrict__ y, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) stf(y + i, ldf(x + i));
}

#define DISPATCH2(di, dto, CALL)                                                   \
    do {                                                                           \
        if ((di) == TDX_F32 && (dto) == TDX_F32) { typedef float TI; typedef float TO; CALL; }      \
        else if ((di) == TDX_F32 && (dto) == TDX_BF16) { typedef float TI; typedef bf16 TO; CALL; } \
        else if ((di) == TDX_BF16 && (dto) == TDX_F32) { typedef bf16 TI; typedef float TO; CALL; } \
        else if ((di) == TDX_BF16 && (dto) == TDX_BF16) { typedef bf16 TI; typedef bf16 TO; CALL; } \
        else if ((di) == TDX_F32 && (dto) == TDX_F16) { typedef float TI; typedef f16 TO; CALL; }   \
        else if ((di) == TDX_F16 && (dto) == TDX_F32) { typedef f16 TI; typedef float TO; CALL; }   \
        else if ((di) == TDX_F16 && (dto) == TDX_F16) { typedef f16 TI; typedef f16 TO; CALL; }     \
        else return TDX_EDTYPE;                                                    \
    } while (0)

extern "C" int tdx_ncv_to_nvc(const void* x, void* y, int B, int C, int64_t V, int di, int dto, void* stream) {
    TDX_CHECK_ARG(x && y && B > 0 && C > 0 && V > 0);
    dim3 grid(ceil_div(V, 64), ceil_div(C, 64), B);
    DISPATCH2(di, dto, hipLaunchKernelGGL((ncv_to_nvc_kernel<TI, TO>), grid, dim3(256), 0, as_stream(stream),
                                           (const TI*)x, (TO*)y, C, V));
    return tdx_launch_status();
}
extern "C" int tdx_nvc_to_ncv(const void* x, void* y, int B, int C, int64_t V, int di, int dto, void* stream) {
    TDX_CHECK_ARG(x && y && B > 0 && C > 0 && V > 0);
    dim3 grid(ceil_div(V, 64), ceil_div(C, 64), B);
    DISPATCH2(di, dto, hipLaunchKernelGGL((nvc_to_ncv_kernel<TI, TO>), grid, dim3(256), 0, as_stream(stream),
                                           (const TI*)x, (TO*)y, C, V));
    return tdx_launch_status();
}
extern "C" int tdx_cast(const void* x, void* y, int64_t n, int di, int dto, void* stream) {
    TDX_CHECK_ARG(x && y && n > 0);
    int grid = (int)min((int64_t)2048, (n + 255) / 256);
    DISPATCH2(di, dto, hipLaunchKernelGGL((cast_kernel<TI, TO>), dim3(grid), dim3(256), 0, as_stream(stream),
                                           (const TI*)x, (TO*)y, n));
    return tdx_launch_status();
}

// ------------------------------------------------------------------ cell mask ------------
__global__ void cell_mask_kernel(const int64_t* __restrict__ idx, int64_t n, uint8_t* __restrict__ mask, int64_t V) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        int64_t j = idx[i];
        if (j >= 0 && j < V) mask[j] = 1;
    }
}
extern "C" int tdx_cell_mask(const int64_t* cell_idx, int64_t n_cells, uint8_t* mask, int64_t V, void* stream) {
    TDX_CHECK_ARG(mask && V > 0 && n_cells >= 0);
    int e = tdx_zero_async(mask, (size_t)V, as_stream(stream));
    if (e != TDX_OK) return e;
    if (n_cells > 0) {
        TDX_CHECK_ARG(cell_idx);
        hipLaunchKernelGGL(cell_mask_kernel, dim3(ceil_div(n_cells, 256)), dim3(256), 0, as_stream(stream), cell_idx,
                           n_cells, mask, V);
    }
    return tdx_launch_status();
}

// ------------------------------------------------------------------ q_sample -------------
// One block row per (b, f) plane so the per-sample coefficients are block-uniform scalars.
// 4 elements per lane (16 B) when V % 4 == 0.
__global__ void __launch_bounds__(256) q_sample_kernel(const float* __restrict__ x0, const float* __restrict__ noise,
                                                       const float* __restrict__ sa, const float* __restrict__ sb,
                                                       const int64_t* __restrict__ t, int t_stride,
                                                       const uint8_t* __restrict__ mask, int keep_bcs,
                                                       float* __restrict__ out, int F, int64_t V) {
    const int plane = blockIdx.y;  // b * F + f
    const int b = plane / F;
    const int64_t tt = t[(int64_t)b * t_stride];
    const float a = sa[tt], s = sb[tt];
    const int64_t base = (int64_t)plane * V;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if ((V & 3) == 0) {
        const int64_t n4 = V >> 2;
        const float4* x4 = reinterpret_cast<const float4*>(x0 + base);
        const float4* z4 = reinterpret_cast<const float4*>(noise + base);
        float4* o4 = reinterpret_cast<float4*>(out + base);
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
            float4 xv = x4[i], zv = z4[i], r;
            r.x = a * xv.x + s * zv.x; r.y = a * xv.y + s * zv.y;
            r.z = a * xv.z + s * zv.z; r.w = a * xv.w + s * zv.w;
            if (keep_bcs) {
                uchar4 m = reinterpret_cast<const uchar4*>(mask)[i];
                if (!m.x) r.x = xv.x; if (!m.y) r.y = xv.y; if (!m.z) r.z = xv.z; if (!m.w) r.w = xv.w;
            }
            o4[i] = r;
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < V; i += stride) {
            float xv = x0[base + i];
            float r = a * xv + s * noise[base + i];
            if (keep_bcs && !mask[i]) r = xv;
            out[base + i] = r;
        }
    }
}

extern "C" int tdx_q_sample(const float* x0, const float* noise, const float* sqrt_ac, const float* sqrt_1mac,
                            const int64_t* t, int t_stride, const uint8_t* mask, int keep_bcs, float* out, int B, int F,
                            int64_t V, void* stream) {
    TDX_CHECK_ARG(x0 && noise && sqrt_ac && sqrt_1mac && t && out && B > 0 && F > 0 && V > 0);
    TDX_CHECK_ARG(!keep_bcs || mask);
    dim3 grid((unsigned)min((int64_t)256, (V / 4 + 255) / 256 + 1), B * F);
    hipLaunchKernelGGL(q_sample_kernel, grid, dim3(256), 0, as_stream(stream), x0, noise, sqrt_ac, sqrt_1mac, t,
                       t_stride, mask, keep_bcs, out, F, V);
    return tdx_launch_status();
}

// ------------------------------------------------------------------ fused reverse step ---
__global__ void __launch_bounds__(256)
p_sample_step_kernel(const float* __restrict__ x_t, const float* __restrict__ eps, const float* __restrict__ z,
                     const float* __restrict__ z2, const float* __restrict__ x_bcs, const uint8_t* __restrict__ mask,
                     const float* __restrict__ sched, int T, const int64_t* __restrict__ tp, int noise_bcs, int clip,
                     float* __restrict__ out, int64_t V) {
    const int64_t t = *tp;
    const float recip = sched[t], recipm1 = sched[T + t], c1 = sched[2 * T + t], c2 = sched[3 * T + t];
    const float sigma = __expf(sched[4 * T + t] * 0.5f);
    const float sa = sched[5 * T + t], sb = sched[6 * T + t];
    const bool last = (t == 0);
    const int64_t base = (int64_t)blockIdx.y * V;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < V; i += stride) {
        const bool inside = mask[i] != 0;
        const float xt = x_t[base + i];
        float x0h = recip * xt - recipm1 * eps[base + i];
        if (!noise_bcs && !inside) x0h = xt;
        if (clip) x0h = fminf(fmaxf(x0h, -1.0f), 1.0f);
        float r = c1 * x0h + c2 * xt;
        if (last) {
            if (!inside) r = x_bcs[base + i];
        } else {
            if (noise_bcs) {
                if (inside) r += sigma * z[base + i];
                else r = sa * x_bcs[base + i] + sb * z2[base + i];
            } else {
                if (inside) r += sigma * z[base + i];
            }
        }
        out[base + i] = r;
    }
}

extern "C" int tdx_p_sample_step(const float* x_t, const float* eps, const float* z, const float* z2,
                                 const float* x_bcs, const uint8_t* mask, const float* sched, int T, const int64_t* t,
                                 int noise_bcs, int clip, float* out, int B, int F, int64_t V, void* stream) {
    TDX_CHECK_ARG(x_t && eps && x_bcs && mask && sched && t && out && T > 0 && B > 0 && F > 0 && V > 0);
    dim3 grid((unsigned)min((int64_t)128, (V + 255) / 256), B * F);
    hipLaunchKernelGGL(p_sample_step_kernel, grid, dim3(256), 0, as_stream(stream), x_t, eps, z, z2, x_bcs, mask, sched,
                       T, t, noise_bcs, clip, out, V);
    return tdx_launch_status();
}

// ------------------------------------------------------------------ masked loss ----------
// pass 1: per-block partial sums in double -> atomicAdd(double) into workspace[0]
// pass 2: scale -> loss.   grad written in pass 1.
__global__ void __launch_bounds__(256)
masked_loss_kernel(const float* __restrict__ e, const float* __restrict__ n, const uint8_t* __restrict__ mask, int l1,
                   double* __restrict__ acc, float* __restrict__ grad, float gscale, int64_t V,
                   const int64_t* __restrict__ n_cells_dev, double samples, double quant) {
    if (n_cells_dev) gscale = (float)(1.0 / (samples * (double)*n_cells_dev));
    const int64_t base = (int64_t)blockIdx.y * V;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < V; i += stride) {
        const float d = e[base + i] - n[base + i];
        const bool in = mask[i] != 0;
        float g = 0.f;
        if (in) {
            if (l1) { s += fabsf(d); g = (d > 0.f) ? gscale : ((d < 0.f) ? -gscale : 0.f); }
            else { s += d * d; g = 2.0f * d * gscale; }
        }
        if (grad) grad[base + i] = g;
    }
    __shared__ double part[4];
    double ws = wave_sum((double)s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = ws;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = part[0] + part[1] + part[2] + part[3];
        if (quant != 0.0) t = rint(t * quant) / quant;  // TDX_DETERMINISTIC: exact (order-independent) f64 sums, see gn_stats_launch
        atomicAdd(acc, t);
    }
}
// the same pass with 16-B accesses, two independent trips in flight per thread (V % 4 == 0: every sample and feature
// plane starts on a 16-B boundary).  The scalar kernel above walks 36 dependent-free but ROLLED trips of 4-B loads per
// thread at 192 x 64 x 48: 67 us for 170 MB (profiles/r11_batch_scaling.txt); this one streams.
__global__ void __launch_bounds__(256)
masked_loss_vec_kernel(const float* __restrict__ e, const float* __restrict__ n, const uint8_t* __restrict__ mask, int l1,
                       double* __restrict__ acc, float* __restrict__ grad, float gscale, int64_t V,
                       const int64_t* __restrict__ n_cells_dev, double samples, double quant) {
    if (n_cells_dev) gscale = (float)(1.0 / (samples * (double)*n_cells_dev));
    const int64_t base = (int64_t)blockIdx.y * V;
    const int64_t V4 = V >> 2, stride = (int64_t)gridDim.x * blockDim.x;
    float s = 0.f;
    auto one = [&](const float4& a, const float4& b, unsigned m, int64_t i) {
        const float d[4] = {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w};
        float g[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            g[k] = 0.f;
            if ((m >> (8 * k)) & 0xffu) {
                if (l1) { s += fabsf(d[k]); g[k] = (d[k] > 0.f) ? gscale : ((d[k] < 0.f) ? -gscale : 0.f); }
                else { s += d[k] * d[k]; g[k] = 2.0f * d[k] * gscale; }
            }
        }
        if (grad) *reinterpret_cast<float4*>(grad + base + 4 * i) = make_float4(g[0], g[1], g[2], g[3]);
    };
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + stride < V4; i += 2 * stride) {
        const float4 a0 = *reinterpret_cast<const float4*>(e + base + 4 * i), b0 = *reinterpret_cast<const float4*>(n + base + 4 * i);
        const float4 a1 = *reinterpret_cast<const float4*>(e + base + 4 * (i + stride)), b1 = *reinterpret_cast<const float4*>(n + base + 4 * (i + stride));
        const unsigned m0 = *reinterpret_cast<const unsigned*>(mask + 4 * i), m1 = *reinterpret_cast<const unsigned*>(mask + 4 * (i + stride));
        one(a0, b0, m0, i);
        one(a1, b1, m1, i + stride);
    }
    for (; i < V4; i += stride)
        one(*reinterpret_cast<const float4*>(e + base + 4 * i), *reinterpret_cast<const float4*>(n + base + 4 * i),
            *reinterpret_cast<const unsigned*>(mask + 4 * i), i);
    __shared__ double part[4];
    double ws = wave_sum((double)s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = ws;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = part[0] + part[1] + part[2] + part[3];
        if (quant != 0.0) t = rint(t * quant) / quant;  // TDX_DETERMINISTIC: exact (order-independent) f64 sums, see gn_stats_launch
        atomicAdd(acc, t);
    }
}
__global__ void masked_loss_finish(const double* acc, float* loss, double inv, const int64_t* n_cells_dev, double samples) {
    if (n_cells_dev) inv = 1.0 / (samples * (double)*n_cells_dev);
    loss[0] = (float)(acc[0] * inv);
}

extern "C" size_t tdx_masked_loss_workspace_bytes(void) { return 16; }
static int masked_loss_launch(const float* eps_hat, const float* noise, const uint8_t* mask, int64_t n_cells,
                              const int64_t* n_cells_dev, int l1, float* loss, float* grad, int B, int F, int64_t V,
                              void* workspace, void* stream) {
    int err = tdx_zero_async(workspace, 16, as_stream(stream));
    if (err != TDX_OK) return err;
    const double samples = (double)B * F;
    const double inv = n_cells_dev ? 0.0 : 1.0 / (samples * (double)n_cells);
    const double quant = tdx_deterministic() ? 1048576.0 : 0.0;  // block partials on a 2^-20 grid: exact up to a total of 2^33
    const bool vec = (V % 4) == 0 && ((uintptr_t)eps_hat % 16) == 0 && ((uintptr_t)noise % 16) == 0 && ((uintptr_t)mask % 4) == 0 &&
                     (grad == nullptr || ((uintptr_t)grad % 16) == 0);
    if (vec) {
        dim3 grid((unsigned)min((int64_t)128, (V / 4 + 511) / 512), B * F);
        hipLaunchKernelGGL(masked_loss_vec_kernel, grid, dim3(256), 0, as_stream(stream), eps_hat, noise, mask, l1,
                           (double*)workspace, grad, (float)inv, V, n_cells_dev, samples, quant);
    } else {
        dim3 grid((unsigned)min((int64_t)64, (V + 255) / 256), B * F);
        hipLaunchKernelGGL(masked_loss_kernel, grid, dim3(256), 0, as_stream(stream), eps_hat, noise, mask, l1,
                           (double*)workspace, grad, (float)inv, V, n_cells_dev, samples, quant);
    }
    hipLaunchKernelGGL(masked_loss_finish, dim3(1), dim3(1), 0, as_stream(stream), (const double*)workspace, loss, inv,
                       n_cells_dev, samples);
    return tdx_launch_status();
}

extern "C" int tdx_masked_loss(const float* eps_hat, const float* noise, const uint8_t* mask, int64_t n_cells, int l1,
                               float* loss, float* grad, int B, int F, int64_t V, void* workspace, void* stream) {
    TDX_CHECK_ARG(eps_hat && noise && mask && loss && workspace && n_cells > 0 && B > 0 && F > 0 && V > 0);
    return masked_loss_launch(eps_hat, noise, mask, n_cells, nullptr, l1, loss, grad, B, F, V, workspace, stream);
}

// the same with the number of in-domain cells read from device memory at run time (int64 scalar): a captured training
// step (hipGraph) serves geometries with different cell counts through one graph
extern "C" int tdx_masked_loss_dyn(const float* eps_hat, const float* noise, const uint8_t* mask,
                                   const int64_t* n_cells_dev, int l1, float* loss, float* grad, int B, int F, int64_t V,
                                   void* workspace, void* stream) {
    TDX_CHECK_ARG(eps_hat && noise && mask && loss && workspace && n_cells_dev && B > 0 && F > 0 && V > 0);
    return masked_loss_launch(eps_hat, noise, mask, 0, n_cells_dev, l1, loss, grad, B, F, V, workspace, stream);
}

// ------------------------------------------------------------------ Philox N(0,1) --------
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
    uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
    c[0] = hi1 ^ c[1] ^ k0; c[1] = lo1; c[2] = hi0 ^ c[3] ^ k1; c[3] = lo0;
}
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
// four N(0,1) draws of counter `ctr` in Philox stream `sid`: (0,1] uniforms from 32 bits each, two Box-Muller pairs
__device__ __forceinline__ void philox_normal4(uint64_t ctr, uint64_t sid, uint64_t seed, float (&r)[4]) {
    uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)sid, (uint32_t)(sid >> 32)};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        float u1 = ((float)c[2 * k] + 1.0f) * 2.3283064365386963e-10f;
        float u2 = (float)c[2 * k + 1] * 2.3283064365386963e-10f;
        float rad = sqrtf(-2.0f * __logf(u1));
        float sn, cs;
        __sincosf(6.283185307179586f * u2, &sn, &cs);
        r[2 * k] = rad * cs; r[2 * k + 1] = rad * sn;
    }
}

__global__ void __launch_bounds__(256) randn_kernel(float* __restrict__ out, int64_t n, uint64_t seed, uint64_t sid,
                                                    const uint64_t* __restrict__ offp) {
    const uint64_t off = *offp;
    const int64_t n4 = (n + 3) >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float r[4];
        philox_normal4(off + (uint64_t)i, sid, seed, r);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (4 * i + k < n) out[4 * i + k] = r[k];
    }
}
__global__ void advance_offset(uint64_t* offp, uint64_t by) { *offp += by; }

__global__ void __launch_bounds__(256) randn_batched_kernel(float* __restrict__ out, int64_t n, uint64_t seed,
                                                            const uint64_t* __restrict__ sids,
                                                            const uint64_t* __restrict__ offp) {
    const uint64_t off = *offp;
    const uint64_t sid = sids[blockIdx.y];
    float* o = out + (int64_t)blockIdx.y * n;
    const int64_t n4 = (n + 3) >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float r[4];
        philox_normal4(off + (uint64_t)i, sid, seed, r);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (4 * i + k < n) o[4 * i + k] = r[k];
    }
}

extern "C" int tdx_randn_batched(float* out, int B, int64_t n, uint64_t seed, const uint64_t* stream_ids,
                                 uint64_t* offset_dev, void* stream) {
    TDX_CHECK_ARG(out && stream_ids && offset_dev && n > 0 && B > 0);
    const int64_t n4 = (n + 3) >> 2;
    dim3 grid((unsigned)min((int64_t)512, (n4 + 255) / 256), B);
    hipLaunchKernelGGL(randn_batched_kernel, grid, dim3(256), 0, as_stream(stream), out, n, seed, stream_ids, offset_dev);
    hipLaunchKernelGGL(advance_offset, dim3(1), dim3(1), 0, as_stream(stream), offset_dev, (uint64_t)n4);
    return tdx_launch_status();
}

// ------------------------------------------------------------------ reverse step, noise drawn in the kernel ---
// The same update as p_sample_step_kernel with z and z2 generated where they are consumed: lane i of sample b draws the
// four normals randn_batched_kernel would have written to z[b][4i..4i+3] (counter off + i) and to z2 (counter
// off + n4 + i), so a run is bit-identical to tdx_randn_batched(z); tdx_randn_batched(z2); tdx_p_sample_step(...).
// Saves two 4-byte writes and two reads per value; the Philox rounds are a few microseconds of VALU per launch.
__global__ void __launch_bounds__(256)
p_sample_step_rng_kernel(const float* __restrict__ x_t, const float* __restrict__ eps, const float* __restrict__ x_bcs,
                         const uint8_t* __restrict__ mask, const float* __restrict__ sched, int T,
                         const int64_t* __restrict__ tp, int noise_bcs, int clip, float* __restrict__ out, int64_t V,
                         int64_t n4, uint64_t seed, const uint64_t* __restrict__ sids,
                         const uint64_t* __restrict__ offp) {
    const int64_t t = *tp;
    const uint64_t off = *offp, sid = sids[blockIdx.y];
    const float recip = sched[t], recipm1 = sched[T + t], c1 = sched[2 * T + t], c2 = sched[3 * T + t];
    const float sigma = __expf(sched[4 * T + t] * 0.5f);
    const float sa = sched[5 * T + t], sb = sched[6 * T + t];
    const bool last = (t == 0);
    const int64_t base4 = (int64_t)blockIdx.y * n4;
    const int64_t v4 = V >> 2;
    const float4* xt4 = reinterpret_cast<const float4*>(x_t) + base4;
    const float4* e4 = reinterpret_cast<const float4*>(eps) + base4;
    const float4* xb4 = reinterpret_cast<const float4*>(x_bcs) + base4;
    const uchar4* m4 = reinterpret_cast<const uchar4*>(mask);
    float4* o4 = reinterpret_cast<float4*>(out) + base4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 xv = xt4[i], ev = e4[i];
        const uchar4 mv = m4[i % v4];
        const bool in[4] = {mv.x != 0, mv.y != 0, mv.z != 0, mv.w != 0};
        const bool any_in = in[0] | in[1] | in[2] | in[3], any_out = !(in[0] & in[1] & in[2] & in[3]);
        const float xt[4] = {xv.x, xv.y, xv.z, xv.w}, ee[4] = {ev.x, ev.y, ev.z, ev.w};
        float xb[4] = {0.f, 0.f, 0.f, 0.f}, z[4] = {0.f, 0.f, 0.f, 0.f}, z2[4] = {0.f, 0.f, 0.f, 0.f};
        if (any_out && (last || noise_bcs)) {
            const float4 bv = xb4[i];
            xb[0] = bv.x; xb[1] = bv.y; xb[2] = bv.z; xb[3] = bv.w;
        }
        if (!last) {
            if (any_in) philox_normal4(off + (uint64_t)i, sid, seed, z);
            if (noise_bcs && any_out) philox_normal4(off + (uint64_t)n4 + (uint64_t)i, sid, seed, z2);
        }
        float r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float x0h = recip * xt[k] - recipm1 * ee[k];
            if (!noise_bcs && !in[k]) x0h = xt[k];
            if (clip) x0h = fminf(fmaxf(x0h, -1.0f), 1.0f);
            float v = c1 * x0h + c2 * xt[k];
            if (last) {
                if (!in[k]) v = xb[k];
            } else if (in[k]) {
                v += sigma * z[k];
            } else if (noise_bcs) {
                v = sa * xb[k] + sb * z2[k];
            }
            r[k] = v;
        }
        o4[i] = make_float4(r[0], r[1], r[2], r[3]);
    }
}
// offset += by; t -= 1: the two scalar updates that close a reverse step, in one launch
__global__ void advance_step(uint64_t* offp, uint64_t by, int64_t* tp) { *offp += by; *tp -= 1; }

extern "C" int tdx_p_sample_step_rng(const float* x_t, const float* eps, const float* x_bcs, const uint8_t* mask,
                                     const float* sched, int T, int64_t* t, int noise_bcs, int clip, float* out, int B,
                                     int F, int64_t V, uint64_t seed, const uint64_t* stream_ids, uint64_t* offset_dev,
                                     void* stream) {
    TDX_CHECK_ARG(x_t && eps && x_bcs && mask && sched && t && out && stream_ids && offset_dev);
    TDX_CHECK_ARG(T > 0 && B > 0 && F > 0 && V > 0 && (V & 3) == 0);
    TDX_CHECK_ARG(((uintptr_t)x_t | (uintptr_t)eps | (uintptr_t)x_bcs | (uintptr_t)out) % 16 == 0 && (uintptr_t)mask % 4 == 0);
    const int64_t n4 = (int64_t)F * V / 4;
    dim3 grid((unsigned)min((int64_t)256, (n4 + 255) / 256), B);
    hipLaunchKernelGGL(p_sample_step_rng_kernel, grid, dim3(256), 0, as_stream(stream), x_t, eps, x_bcs, mask, sched, T,
                       (const int64_t*)t, noise_bcs, clip, out, V, n4, seed, stream_ids, (const uint64_t*)offset_dev);
    hipLaunchKernelGGL(advance_step, dim3(1), dim3(1), 0, as_stream(stream), offset_dev,
                       (uint64_t)(noise_bcs ? 2 * n4 : n4), t);
    return tdx_launch_status();
}

extern "C" int tdx_randn(float* out, int64_t n, uint64_t seed, uint64_t stream_id, uint64_t* offset_dev, void* stream) {
    TDX_CHECK_ARG(out && offset_dev && n > 0);
    const int64_t n4 = (n + 3) >> 2;
    int grid = (int)min((int64_t)2048, (n4 + 255) / 256);
    hipLaunchKernelGGL(randn_kernel, dim3(grid), dim3(256), 0, as_stream(stream), out, n, seed, stream_id, offset_dev);
    hipLaunchKernelGGL(advance_offset, dim3(1), dim3(1), 0, as_stream(stream), offset_dev, (uint64_t)n4);
    return tdx_launch_status();
}


// ------------------------------------------------------------------ host signal ----------
// One lane stores gen * TDX_SIGNAL_STRIDE + k into a word of HOST memory (pinned, device-mapped) with a system-scope release:
// everything enqueued before it on the stream -- the staging kernels of a gradient bucket inside a captured backward pass --
// has completed and is visible.  This runtime refuses event-record nodes inside a captured graph
// (hipEventRecordWithFlags(..., hipEventRecordExternal) -> hipErrorInvalidValue; torch: "External events are disallowed in
// rocm", tools/micro/external_event_probe.py), so a graph-external stream cannot wait on a point INSIDE a graph; the host
// can: it polls this word and launches the bucket's all-reduce on its communication stream while the rest of the graph runs.
__global__ void signal_host_kernel(uint32_t* __restrict__ flag, const uint32_t* __restrict__ gen, uint32_t k) {
    __hip_atomic_store(flag, gen[0] * (uint32_t)TDX_SIGNAL_STRIDE + k, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
extern "C" int tdx_signal_host(uint32_t* host_flag, const uint32_t* gen_dev, uint32_t k, void* stream) {
    TDX_CHECK_ARG(host_flag && gen_dev && k < (uint32_t)TDX_SIGNAL_STRIDE);
    hipLaunchKernelGGL(signal_host_kernel, dim3(1), dim3(1), 0, as_stream(stream), host_flag, gen_dev, k);
    return tdx_launch_status();
}
