// General 3-D convolution family of the reference's regression baselines, gfx950:
//   dilated 3x3x3, replicate padding          DilatedCNNBlock, dilresnet.py:22-38   (dilation 1, 2, 4, 8)
//   strided k x k x k, zero padding           tfnet.py:185-199  conv()    (stride 2)
//   transposed 4x4x4, stride 2, padding 1     tfnet.py:201-208  deconv()
// on NDHWC tensors (fp32 or bf16 storage, fp32 accumulation).  These layers are off the benchmark path.  The kernels
// in this file are vector-ALU kernels with the arithmetic laid out for coalesced 16-B accesses (a thread owns one
// output voxel and 8 output channels; the weights of a tap are read as [ci][co] rows shared by the whole wave): they
// serve fp32 tensors.  bf16 tensors take the matrix-core kernels of tdx_convg_mfma.hip through the same entry points
// (TDX_CONVG_MFMA=0 keeps them here: the A/B switch of tools/baseline_conv_bench.py).
//
// Three kernels cover forward and both gradients of all three layer types:
//   gather     out[o]  = sum_t W[t] in[src(o, t)],      src = o * stride - pad + t * dilation  (zero or clamped)
//              = conv forward; = data gradient of the transposed conv
//   scatter^T  out[i]  = sum_t W[t] in[(i + pad - t * dilation) / stride]   where divisible and in range
//              = data gradient of a zero-padded conv; = transposed-conv forward.  The replicate-padded (dilated)
//              conv takes its data gradient on the padded grid (pad = 0 there) followed by fold_clamp, which adds
//              every padded position onto the voxel it clamps to.
//   wgrad      dW[t][ci][co] = sum_{b, o} in[src(o, t)][ci] * dy[o][co]
// Weights are passed as [taps][Cin][Cout] fp32 (the host transposes the reference layouts once per call).
#include "tdx_common.h"
#include <stdlib.h>

int convg_mfma_apply(const void* in, const float* w, const float* bias, void* out, int B, const int* Ei, const int* Eo, int Cin,
                     int Cout, int k, int stride, int dil, int pad, int replicate, int transposed, hipStream_t st);
int convg_mfma_bwd_weight(const void* in, const void* dy, float* dw, float* dbias, int B, const int* Ei, const int* Eo, int Cin,
                          int Cout, int k, int stride, int dil, int pad, int replicate, hipStream_t st);
static bool convg_use_mfma(int dtype) {  // read per call: the switch is a test / benchmark knob
    const char* e = getenv("TDX_CONVG_MFMA");
    return dtype == TDX_BF16 && !(e && atoi(e) == 0);
}

struct ConvG {
    int B;
    int Ei[3], Eo[3];  // grid of `in` and of `out`
    int k, stride, dil, pad;
    int clamp;         // gather only: 1 = replicate padding (clamp the source), 0 = zero padding
    int Cin, Cout;     // channels of `in` and of `out`
};

template <typename T, bool TRANSPOSED>
__global__ void __launch_bounds__(256)
convg_kernel(const T* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias, T* __restrict__ out, ConvG g) {
    const int groups = g.Cout >> 3;  // 8 output channels per thread
    const int64_t nout = (int64_t)g.B * g.Eo[0] * g.Eo[1] * g.Eo[2];
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nout * groups) return;
    const int cg = (int)(idx % groups);
    int64_t v = idx / groups;
    const int o2 = (int)(v % g.Eo[2]); v /= g.Eo[2];
    const int o1 = (int)(v % g.Eo[1]); v /= g.Eo[1];
    const int o0 = (int)(v % g.Eo[0]);
    const int b = (int)(v / g.Eo[0]);
    const int o[3] = {o0, o1, o2};
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = bias ? bias[cg * 8 + j] : 0.f;
    const int k = g.k;
    for (int t0 = 0; t0 < k; ++t0)
        for (int t1 = 0; t1 < k; ++t1)
            for (int t2 = 0; t2 < k; ++t2) {
                const int t[3] = {t0, t1, t2};
                int s[3];
                bool ok = true;
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    if (TRANSPOSED) {
                        const int num = o[a] + g.pad - t[a] * g.dil;
                        ok = ok && num >= 0 && (num % g.stride) == 0 && num / g.stride < g.Ei[a];
                        s[a] = num / g.stride;
                    } else {
                        int q = o[a] * g.stride - g.pad + t[a] * g.dil;
                        if (g.clamp) q = min(max(q, 0), g.Ei[a] - 1);
                        ok = ok && q >= 0 && q < g.Ei[a];
                        s[a] = q;
                    }
                }
                if (!ok) continue;
                const T* src = in + ((((int64_t)b * g.Ei[0] + s[0]) * g.Ei[1] + s[1]) * g.Ei[2] + s[2]) * g.Cin;
                const float* wt = w + ((int64_t)((t0 * k + t1) * k + t2) * g.Cin) * g.Cout + cg * 8;
                for (int ci = 0; ci < g.Cin; ci += 8) {
                    Vec8<T> xv;
                    xv.load(src + ci);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float4 wa = *reinterpret_cast<const float4*>(wt + (int64_t)(ci + e) * g.Cout);
                        const float4 wb = *reinterpret_cast<const float4*>(wt + (int64_t)(ci + e) * g.Cout + 4);
                        acc[0] += xv.v[e] * wa.x; acc[1] += xv.v[e] * wa.y; acc[2] += xv.v[e] * wa.z; acc[3] += xv.v[e] * wa.w;
                        acc[4] += xv.v[e] * wb.x; acc[5] += xv.v[e] * wb.y; acc[6] += xv.v[e] * wb.z; acc[7] += xv.v[e] * wb.w;
                    }
                }
            }
    Vec8<T> r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r.v[j] = acc[j];
    r.store(out + (idx / groups) * g.Cout + cg * 8);
}

// dx[i] = sum of dpad[q] over the padded positions q (grid E + 2 pad) that clamp onto i.  Interior voxels have one.
template <typename T>
__global__ void __launch_bounds__(256)
convg_fold_clamp_kernel(const T* __restrict__ dpad, T* __restrict__ dx, int B, int E0, int E1, int E2, int pad, int C) {
    const int groups = C >> 3;
    const int64_t n = (int64_t)B * E0 * E1 * E2;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * groups) return;
    const int cg = (int)(idx % groups);
    int64_t v = idx / groups;
    const int i2 = (int)(v % E2); v /= E2;
    const int i1 = (int)(v % E1); v /= E1;
    const int i0 = (int)(v % E0);
    const int b = (int)(v / E0);
    const int E[3] = {E0, E1, E2}, i[3] = {i0, i1, i2};
    int lo[3], hi[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        lo[a] = i[a] == 0 ? 0 : i[a] + pad;                    // padded coordinates q = i + pad; q <= pad clamps to 0
        hi[a] = i[a] == E[a] - 1 ? E[a] - 1 + 2 * pad : i[a] + pad;
    }
    const int P1 = E1 + 2 * pad, P2 = E2 + 2 * pad;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int q0 = lo[0]; q0 <= hi[0]; ++q0)
        for (int q1 = lo[1]; q1 <= hi[1]; ++q1)
            for (int q2 = lo[2]; q2 <= hi[2]; ++q2) {
                Vec8<T> t;
                t.load(dpad + ((((int64_t)b * (E0 + 2 * pad) + q0) * P1 + q1) * P2 + q2) * C + cg * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += t.v[j];
            }
    Vec8<T> r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r.v[j] = acc[j];
    r.store(dx + (idx / groups) * C + cg * 8);
}

// dW[t][ci][co]: one workgroup per (tap, 8-channel ci group, chunk of output voxels); thread = (ci lane, co group)
#define CGW_VOX 4096
template <typename T>
__global__ void __launch_bounds__(256)
convg_wgrad_kernel(const T* __restrict__ in, const T* __restrict__ dy, float* __restrict__ dw, float* __restrict__ dbias, ConvG g) {
    // g.Cin = channels of `in` (the conv's input), g.Cout = channels of dy; g.Eo = dy grid
    const int tap = blockIdx.y, k = g.k;
    const int t[3] = {tap / (k * k), (tap / k) % k, tap % k};
    const int ci0 = blockIdx.z * 8;
    const int64_t nvox = (int64_t)g.B * g.Eo[0] * g.Eo[1] * g.Eo[2];
    const int64_t v0 = (int64_t)blockIdx.x * CGW_VOX, v1 = min(nvox, v0 + CGW_VOX);
    const int cgs = g.Cout >> 3;           // co groups of 8
    // threads: co group = tid % cgs_pad, voxel lane = tid / cgs_pad
    const int tid = threadIdx.x;
    const int lanes = 256 / cgs;           // voxels in flight per iteration (cgs <= 64 -> lanes >= 4)
    const int cg = tid % cgs, vl = tid / cgs;
    float acc[8][8];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[a][c] = 0.f;
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bool do_bias = dbias != nullptr && tap == 0 && blockIdx.z == 0;
    if (vl < lanes) {
        for (int64_t v = v0 + vl; v < v1; v += lanes) {
            int64_t r = v;
            const int o2 = (int)(r % g.Eo[2]); r /= g.Eo[2];
            const int o1 = (int)(r % g.Eo[1]); r /= g.Eo[1];
            const int o0 = (int)(r % g.Eo[0]);
            const int b = (int)(r / g.Eo[0]);
            const int o[3] = {o0, o1, o2};
            Vec8<T> gy;
            gy.load(dy + v * g.Cout + cg * 8);
            if (do_bias) {
#pragma unroll
                for (int c = 0; c < 8; ++c) bsum[c] += gy.v[c];
            }
            int s[3];
            bool ok = true;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                int q = o[a] * g.stride - g.pad + t[a] * g.dil;
                if (g.clamp) q = min(max(q, 0), g.Ei[a] - 1);
                ok = ok && q >= 0 && q < g.Ei[a];
                s[a] = q;
            }
            if (!ok) continue;
            Vec8<T> xv;
            xv.load(in + ((((int64_t)b * g.Ei[0] + s[0]) * g.Ei[1] + s[1]) * g.Ei[2] + s[2]) * g.Cin + ci0);
#pragma unroll
            for (int a = 0; a < 8; ++a)
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[a][c] += xv.v[a] * gy.v[c];
        }
    }
    // reduce over the voxel lanes through LDS, then one atomic per (ci, co)
    __shared__ float red[256][9];
    for (int a = 0; a < 8; ++a) {
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 8; ++c) red[tid][c] = (vl < lanes) ? acc[a][c] : 0.f;
        __syncthreads();
        if (tid < cgs * 8) {
            const int g8 = tid / 8, c = tid % 8;
            float s = 0.f;
            for (int l = 0; l < lanes; ++l) s += red[l * cgs + g8][c];
            atomicAdd(&dw[((int64_t)tap * g.Cin + ci0 + a) * g.Cout + g8 * 8 + c], s);
        }
    }
    if (do_bias) {
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 8; ++c) red[tid][c] = (vl < lanes) ? bsum[c] : 0.f;
        __syncthreads();
        if (tid < cgs * 8) {
            const int g8 = tid / 8, c = tid % 8;
            float s = 0.f;
            for (int l = 0; l < lanes; ++l) s += red[l * cgs + g8][c];
            atomicAdd(&dbias[g8 * 8 + c], s);
        }
    }
}

static int convg_check(int B, const int* Ei, const int* Eo, int k, int stride, int dil, int pad, int Cin, int Cout) {
    if (B <= 0 || k < 1 || k > 7 || stride < 1 || dil < 1 || pad < 0) return TDX_EINVAL;
    for (int a = 0; a < 3; ++a)
        if (Ei[a] <= 0 || Eo[a] <= 0) return TDX_EINVAL;
    if ((Cin % 8) || (Cout % 8) || Cin <= 0 || Cout <= 0) return TDX_ESHAPE;
    return TDX_OK;
}

// out = gather (transposed == 0) or scatter^T (transposed != 0) of `in` with w [k^3][Cin][Cout] (+ bias)
extern "C" int tdx_convg_apply(const void* in, const float* w, const float* bias, void* out, int B, int Xi, int Yi, int Zi,
                               int Cin, int Xo, int Yo, int Zo, int Cout, int k, int stride, int dilation, int pad,
                               int replicate, int transposed, int dtype, void* stream) {
    TDX_CHECK_ARG(in && w && out);
    ConvG g = {B, {Xi, Yi, Zi}, {Xo, Yo, Zo}, k, stride, dilation, pad, replicate, Cin, Cout};
    int rc = convg_check(B, g.Ei, g.Eo, k, stride, dilation, pad, Cin, Cout);
    if (rc != TDX_OK) return rc;
    if (transposed && replicate) return TDX_EINVAL;
    if (convg_use_mfma(dtype))
        return convg_mfma_apply(in, w, bias, out, B, g.Ei, g.Eo, Cin, Cout, k, stride, dilation, pad, replicate, transposed,
                                as_stream(stream));
    const int64_t total = (int64_t)B * Xo * Yo * Zo * (Cout / 8);
    if (transposed)
        TDX_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((convg_kernel<T, true>), dim3(ceil_div(total, 256)), dim3(256), 0,
                                                      as_stream(stream), (const T*)in, w, bias, (T*)out, g));
    else
        TDX_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((convg_kernel<T, false>), dim3(ceil_div(total, 256)), dim3(256), 0,
                                                      as_stream(stream), (const T*)in, w, bias, (T*)out, g));
    return tdx_launch_status();
}

// dx[i] = sum of dpad over the positions of the (E + 2 pad)^3 grid that clamp onto i
extern "C" int tdx_convg_fold_clamp(const void* dpad, void* dx, int B, int X, int Y, int Z, int pad, int C, int dtype,
                                    void* stream) {
    TDX_CHECK_ARG(dpad && dx && B > 0 && X > 0 && Y > 0 && Z > 0 && pad >= 0 && C > 0);
    if (C % 8) return TDX_ESHAPE;
    const int64_t total = (int64_t)B * X * Y * Z * (C / 8);
    TDX_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((convg_fold_clamp_kernel<T>), dim3(ceil_div(total, 256)), dim3(256), 0,
                                                  as_stream(stream), (const T*)dpad, (T*)dx, B, X, Y, Z, pad, C));
    return tdx_launch_status();
}

// dw [k^3][Cin][Cout] f32 (+ dbias [Cout]) of out = gather(in, w): in (B, Xi.., Cin), dy (B, Xo.., Cout).
// Accumulates with atomics: dw / dbias must be zero on entry.
extern "C" int tdx_convg_bwd_weight(const void* in, const void* dy, float* dw, float* dbias, int B, int Xi, int Yi, int Zi,
                                    int Cin, int Xo, int Yo, int Zo, int Cout, int k, int stride, int dilation, int pad,
                                    int replicate, int dtype, void* stream) {
    TDX_CHECK_ARG(in && dy && dw);
    ConvG g = {B, {Xi, Yi, Zi}, {Xo, Yo, Zo}, k, stride, dilation, pad, replicate, Cin, Cout};
    int rc = convg_check(B, g.Ei, g.Eo, k, stride, dilation, pad, Cin, Cout);
    if (rc != TDX_OK) return rc;
    if (convg_use_mfma(dtype))
        return convg_mfma_bwd_weight(in, dy, dw, dbias, B, g.Ei, g.Eo, Cin, Cout, k, stride, dilation, pad, replicate,
                                     as_stream(stream));
    if (Cout > 512) return TDX_ESHAPE;  // co groups of 8 must fit one workgroup
    const int64_t nvox = (int64_t)B * Xo * Yo * Zo;
    dim3 grid(ceil_div(nvox, CGW_VOX), k * k * k, Cin / 8);
    TDX_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((convg_wgrad_kernel<T>), grid, dim3(256), 0, as_stream(stream), (const T*)in,
                                                  (const T*)dy, dw, dbias, g));
    return tdx_launch_status();
}
