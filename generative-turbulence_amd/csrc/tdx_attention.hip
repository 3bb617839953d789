// Bottleneck self-attention on the to_qkv output (attention.py:9-15, ddpm.py:295-308).
//
// qkv is the NDHWC output of the 1x1 to_qkv conv: [B][N][3*H*D] with channel thirds q|k|v,
// head-major inside each third, so no permute/contiguous copies are needed (the reference
// makes three, ddpm.py:298-303).
//
// Vector-ALU "row" kernels: one lane owns one query (fwd, dQ) or one key (dK/dV) row with
// its D=32 vector in registers and walks the other side; all lanes of a wave read the same
// K/V (or Q/dO) row, i.e. wave-uniform broadcast loads, and keep an online softmax.  This is
// exact fp32 and is the path used at the U-Net bottleneck (N = 144 tokens at 192x64x48,
// 0.01 GFLOP).  The MFMA flash kernel for long sequences lives in tdx_attention_mfma.hip.
#include "tdx_common.h"
#include <stdlib.h>

// The "other side" rows are walked in tiles of 64 staged in LDS as f32 (thread t stages row t with
// 16-B loads); the inner loops then read them with wave-uniform (broadcast) LDS loads instead of a
// chain of dependent global loads per row.
#define AT_TILE 64
template <typename T, int D>
__device__ __forceinline__ void attn_stage_row(float* __restrict__ dst, const T* __restrict__ src, bool ok) {
#pragma unroll
    for (int c = 0; c < D / 8; ++c) {
        Vec8<T> v;
        if (ok) v.load(src + c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) dst[c * 8 + e] = ok ? v.v[e] : 0.f;
    }
}

template <typename T, int D>
__global__ void __launch_bounds__(64)
attn_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ out, float* __restrict__ lse, int N, int H) {
    const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
    const int i = blockIdx.x * 64 + threadIdx.x;
    const int ld = 3 * H * D;
    const T* base = qkv + (int64_t)b * N * ld;
    const float scale = rsqrtf((float)D);
    float q[D], acc[D];
    const bool valid = i < N;
    const int ii = valid ? i : N - 1;
#pragma unroll
    for (int d = 0; d < D; ++d) { q[d] = ldf(base + (int64_t)ii * ld + h * D + d) * scale; acc[d] = 0.f; }
    float m = -INFINITY, l = 0.f;
    __shared__ float sK[AT_TILE][D], sV[AT_TILE][D];
    for (int j0 = 0; j0 < N; j0 += AT_TILE) {
        const int jr = j0 + (int)threadIdx.x;
        __syncthreads();
        attn_stage_row<T, D>(sK[threadIdx.x], base + (int64_t)jr * ld + H * D + h * D, jr < N);
        attn_stage_row<T, D>(sV[threadIdx.x], base + (int64_t)jr * ld + 2 * H * D + h * D, jr < N);
        __syncthreads();
        const int nj = min(AT_TILE, N - j0);
        for (int j = 0; j < nj; ++j) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) s += q[d] * sK[j][d];
            const float mn = fmaxf(m, s);
            const float alpha = __expf(m - mn);
            const float p = __expf(s - mn);
            l = l * alpha + p;
#pragma unroll
            for (int d = 0; d < D; ++d) acc[d] = acc[d] * alpha + p * sV[j][d];
            m = mn;
        }
    }
    if (valid) {
        const float inv = 1.0f / l;
        T* o = out + ((int64_t)b * N + i) * (H * D) + h * D;
#pragma unroll
        for (int d = 0; d < D; ++d) stf(o + d, acc[d] * inv);
        lse[((int64_t)b * H + h) * N + i] = m + __logf(l);
    }
}

// delta[b,h,i] = sum_d dO[i,d] O[i,d]
template <typename T, int D>
__global__ void attn_delta_kernel(const T* __restrict__ out, const T* __restrict__ dout, float* __restrict__ delta,
                                  int N, int H, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (b, i, h)
    if (idx >= total) return;
    const int h = (int)(idx % H);
    const int64_t bi = idx / H;
    const int i = (int)(bi % N);
    const int b = (int)(bi / N);
    const T* o = out + bi * (H * D) + h * D;
    const T* g = dout + bi * (H * D) + h * D;
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) s += ldf(o + d) * ldf(g + d);
    delta[((int64_t)b * H + h) * N + i] = s;
}

// lane = query i:  dQ_i = scale * sum_j dS_ij K_j,  dS_ij = P_ij (dO_i . V_j - delta_i)
template <typename T, int D>
__global__ void __launch_bounds__(64)
attn_bwd_dq_kernel(const T* __restrict__ qkv, const T* __restrict__ dout, const float* __restrict__ lse,
                   const float* __restrict__ delta, T* __restrict__ dqkv, int N, int H) {
    const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
    const int i = blockIdx.x * 64 + threadIdx.x;
    const int ld = 3 * H * D;
    const T* base = qkv + (int64_t)b * N * ld;
    const float scale = rsqrtf((float)D);
    const bool valid = i < N;
    const int ii = valid ? i : N - 1;
    float q[D], g[D], acc[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        q[d] = ldf(base + (int64_t)ii * ld + h * D + d) * scale;
        g[d] = ldf(dout + ((int64_t)b * N + ii) * (H * D) + h * D + d);
        acc[d] = 0.f;
    }
    const float L = lse[((int64_t)b * H + h) * N + ii];
    const float dl = delta[((int64_t)b * H + h) * N + ii];
    __shared__ float sK[AT_TILE][D], sV[AT_TILE][D];
    for (int j0 = 0; j0 < N; j0 += AT_TILE) {
        const int jr = j0 + (int)threadIdx.x;
        __syncthreads();
        attn_stage_row<T, D>(sK[threadIdx.x], base + (int64_t)jr * ld + H * D + h * D, jr < N);
        attn_stage_row<T, D>(sV[threadIdx.x], base + (int64_t)jr * ld + 2 * H * D + h * D, jr < N);
        __syncthreads();
        const int nj = min(AT_TILE, N - j0);
        for (int j = 0; j < nj; ++j) {
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) { s += q[d] * sK[j][d]; dp += g[d] * sV[j][d]; }
            const float ds = __expf(s - L) * (dp - dl);
#pragma unroll
            for (int d = 0; d < D; ++d) acc[d] += ds * sK[j][d];
        }
    }
    if (valid) {
        T* o = dqkv + ((int64_t)b * N + i) * ld + h * D;
#pragma unroll
        for (int d = 0; d < D; ++d) stf(o + d, acc[d] * scale);
    }
}

// lane = key j:  dV_j = sum_i P_ij dO_i,  dK_j = scale * sum_i dS_ij Q_i
template <typename T, int D>
__global__ void __launch_bounds__(64)
attn_bwd_dkv_kernel(const T* __restrict__ qkv, const T* __restrict__ dout, const float* __restrict__ lse,
                    const float* __restrict__ delta, T* __restrict__ dqkv, int N, int H) {
    const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
    const int j = blockIdx.x * 64 + threadIdx.x;
    const int ld = 3 * H * D;
    const T* base = qkv + (int64_t)b * N * ld;
    const float scale = rsqrtf((float)D);
    const bool valid = j < N;
    const int jj = valid ? j : N - 1;
    float k[D], v[D], dk[D], dv[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        k[d] = ldf(base + (int64_t)jj * ld + H * D + h * D + d) * scale;
        v[d] = ldf(base + (int64_t)jj * ld + 2 * H * D + h * D + d);
        dk[d] = dv[d] = 0.f;
    }
    __shared__ float sQ[AT_TILE][D], sG[AT_TILE][D], sL[AT_TILE], sDl[AT_TILE];
    for (int i0 = 0; i0 < N; i0 += AT_TILE) {
        const int ir = i0 + (int)threadIdx.x;
        __syncthreads();
        attn_stage_row<T, D>(sQ[threadIdx.x], base + (int64_t)ir * ld + h * D, ir < N);
        attn_stage_row<T, D>(sG[threadIdx.x], dout + ((int64_t)b * N + ir) * (H * D) + h * D, ir < N);
        sL[threadIdx.x] = ir < N ? lse[((int64_t)b * H + h) * N + ir] : 0.f;
        sDl[threadIdx.x] = ir < N ? delta[((int64_t)b * H + h) * N + ir] : 0.f;
        __syncthreads();
        const int ni = min(AT_TILE, N - i0);
        for (int i = 0; i < ni; ++i) {
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) { s += k[d] * sQ[i][d]; dp += v[d] * sG[i][d]; }
            const float p = __expf(s - sL[i]);
            const float ds = p * (dp - sDl[i]);
#pragma unroll
            for (int d = 0; d < D; ++d) { dv[d] += p * sG[i][d]; dk[d] += ds * sQ[i][d]; }
        }
    }
    if (valid) {
        T* ok = dqkv + ((int64_t)b * N + j) * ld + H * D + h * D;
        T* ov = ok + H * D;
#pragma unroll
        for (int d = 0; d < D; ++d) { stf(ok + d, dk[d] * scale); stf(ov + d, dv[d]); }
    }
}

// tdx_attention_mfma.hip; TDX_ATTN_IMPL=vector forces the kernels of this file
bool attn_mfma_supported(int N, int D);
int attn_fwd_mfma_launch(const void* qkv, void* out, float* lse, int B, int N, int H, int dtype, hipStream_t st);
// tdx_attention_bwd_mfma.hip
int attn_bwd_mfma_launch(const void* qkv, const void* dout, const float* lse, const float* delta, void* dqkv, int B, int N,
                         int H, int dtype, hipStream_t st);

extern "C" int tdx_attn_fwd(const void* qkv, void* out, float* lse, int B, int N, int H, int D, int dtype,
                            void* stream) {
    TDX_CHECK_ARG(qkv && out && lse && B > 0 && N > 0 && H > 0);
    if (D != 32) return TDX_ESHAPE;
    {
        const char* e = getenv("TDX_ATTN_IMPL");
        const bool force_vector = e && e[0] == 'v';
        if ((dtype == TDX_BF16 || dtype == TDX_F16) && !force_vector && attn_mfma_supported(N, D))
            return attn_fwd_mfma_launch(qkv, out, lse, B, N, H, dtype, as_stream(stream));
    }
    dim3 grid(ceil_div(N, 64), B * H);
    TDX_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((attn_fwd_kernel<T, 32>), grid, dim3(64), 0, as_stream(stream),
                                                  (const T*)qkv, (T*)out, lse, N, H));
    return tdx_launch_status();
}

extern "C" size_t tdx_attn_bwd_workspace_bytes(int B, int N, int H, int D) {
    (void)D;
    return (size_t)B * N * H * sizeof(float);
}

extern "C" int tdx_attn_bwd(const void* qkv, const void* out, const float* lse, const void* dout, void* dqkv, int B,
                            int N, int H, int D, int dtype, void* workspace, void* stream) {
    TDX_CHECK_ARG(qkv && out && lse && dout && dqkv && workspace && B > 0 && N > 0 && H > 0);
    if (D != 32) return TDX_ESHAPE;
    float* delta = (float*)workspace;
    const int64_t total = (int64_t)B * N * H;
    dim3 grid(ceil_div(N, 64), B * H);
    hipStream_t st = as_stream(stream);
    {
        const char* e = getenv("TDX_ATTN_IMPL");
        const bool force_vector = e && e[0] == 'v';
        if (dtype == TDX_BF16 && !force_vector && attn_mfma_supported(N, D)) {  // long sequences: MFMA flash backward
            hipLaunchKernelGGL((attn_delta_kernel<bf16, 32>), dim3(ceil_div(total, 256)), dim3(256), 0, st, (const bf16*)out,
                               (const bf16*)dout, delta, N, H, total);
            return attn_bwd_mfma_launch(qkv, dout, lse, delta, dqkv, B, N, H, dtype, st);
        }
        if (dtype == TDX_F16 && !force_vector && attn_mfma_supported(N, D)) {
            hipLaunchKernelGGL((attn_delta_kernel<f16, 32>), dim3(ceil_div(total, 256)), dim3(256), 0, st, (const f16*)out,
                               (const f16*)dout, delta, N, H, total);
            return attn_bwd_mfma_launch(qkv, dout, lse, delta, dqkv, B, N, H, dtype, st);
        }
    }
    TDX_DISPATCH_DTYPE(dtype, {
        hipLaunchKernelGGL((attn_delta_kernel<T, 32>), dim3(ceil_div(total, 256)), dim3(256), 0, st, (const T*)out,
                           (const T*)dout, delta, N, H, total);
        hipLaunchKernelGGL((attn_bwd_dq_kernel<T, 32>), grid, dim3(64), 0, st, (const T*)qkv, (const T*)dout, lse,
                           delta, (T*)dqkv, N, H);
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, 32>), grid, dim3(64), 0, st, (const T*)qkv, (const T*)dout, lse,
                           delta, (T*)dqkv, N, H);
    });
    return tdx_launch_status();
}
