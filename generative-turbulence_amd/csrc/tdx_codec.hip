// Model-boundary kernels: the 4 -> dim encoders and the dim -> 4 decoder of the DenoisingModel
// fused with the NCDHW <-> NDHWC layout change and the channel concatenation
// (ddpm.py:433,436,459,495-501,505).
//
//   encode: y[b,v, 0:D ] = Wx x[b,:,v] + bx         x  (B, Fx, V) f32  NCDHW
//           y[b,v, D:2D] = Wc c[:,v]   + bc         c  (Fc, V)    f32, shared by the batch
//   decode: y[b,f,v] = sum_c W[f][c] h[b,v,c] + b[f]   h NDHWC, y (B, F, V) f32
//
// HBM-bound streaming kernels (the wide NDHWC side dominates the bytes): lane (lc, r) owns the
// 8-channel vector lc of voxel r, 16 B per lane on the NDHWC side; the few-channel NCDHW side
// is read/written as per-plane scalars.  Parameter gradients are reduced per block through LDS
// and merged with one f32 atomic per value.
#include "tdx_common.h"
#include "tdx_conv3.h"  // tdx_deterministic, ordered_sum_launch, the scratch arena

#define CD_THREADS 256
#define CD_VOX 512  // voxels per block (~4 resident blocks per CU at 192x64x48)
#define CD_VOX_DEC 2048  // decode: grid already has a batch dimension; fewer blocks -> fewer gradient atomics
#define CD_MAXF 8

template <int N>
__device__ __forceinline__ void block_reduce_lanes(float (&s)[N], int L, int lc, int r, int rows, float* __restrict__ out,
                                                   int out_stride) {
    // sum s[] over the `rows` threads that share lc; result j of lane-vector lc -> out[lc*out_stride + j]
    __shared__ float red[CD_THREADS][N + 1];
    const int tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < N; ++j) red[tid][j] = (r < rows) ? s[j] : 0.f;
    __syncthreads();
    for (int o = tid; o < L * N; o += CD_THREADS) {
        const int l = o / N, j = o - l * N;
        float t = 0.f;
        for (int q = 0; q < rows; ++q) t += red[q * L + l][j];
        atomicAdd(&out[l * out_stride + j], t);
    }
}

// ------------------------------------------------------------------ encode ---------------
template <typename T, int F>
__global__ void __launch_bounds__(CD_THREADS)
encode_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wx, const float* __restrict__ bx,
                  const float* __restrict__ c, const float* __restrict__ wc, const float* __restrict__ bc,
                  T* __restrict__ y, int64_t V, int D, int Dtot) {
    const int b = blockIdx.y;
    const int L = Dtot >> 3, rows = CD_THREADS / L;
    const int lc = threadIdx.x % L, r = threadIdx.x / L;
    if (r >= rows) return;
    const bool is_x = lc * 8 < D;
    const int ch0 = is_x ? lc * 8 : lc * 8 - D;
    const float* w = is_x ? wx : wc;
    const float* bias = is_x ? bx : bc;
    const float* src = is_x ? x + (int64_t)b * F * V : c;
    float wr[8][F], br[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        br[j] = bias[ch0 + j];
#pragma unroll
        for (int k = 0; k < F; ++k) wr[j][k] = w[(ch0 + j) * F + k];
    }
    const int64_t v0 = (int64_t)blockIdx.x * CD_VOX, v1 = min(V, v0 + CD_VOX);
    for (int64_t v = v0 + r; v < v1; v += rows) {
        float in[F];
#pragma unroll
        for (int k = 0; k < F; ++k) in[k] = src[(int64_t)k * V + v];
        Vec8<T> o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float a = br[j];
#pragma unroll
            for (int k = 0; k < F; ++k) a = __builtin_fmaf(wr[j][k], in[k], a);  // explicit: gn_apply_encoded_kernel repeats it
            o.v[j] = a;
        }
        o.store(y + ((int64_t)b * V + v) * Dtot + lc * 8);
    }
}

// dw[ch][k] += sum_v dy[b,v,ch] in[k][v]; db[ch] += sum_v dy; dc[k][v] = sum_b sum_ch wc[ch][k] dy[b,v,D+ch]
template <typename T, int F>
__global__ void __launch_bounds__(CD_THREADS)
encode_bwd_kernel(const T* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ c,
                  const float* __restrict__ wc, float* __restrict__ dwx, float* __restrict__ dbx,
                  float* __restrict__ dwc, float* __restrict__ dbc, float* __restrict__ dc, int B, int64_t V, int D,
                  int Dtot, int64_t slab_stride) {
    // slab_stride != 0 (TDX_DETERMINISTIC): the four gradient pointers address block 0's zeroed slab; block k merges into slab k
    // (one contributor per element), the host adds the slabs in order afterwards
    dwx += blockIdx.x * slab_stride; dbx += blockIdx.x * slab_stride;
    if (dwc) { dwc += blockIdx.x * slab_stride; dbc += blockIdx.x * slab_stride; }
    const int L = Dtot >> 3, rows = CD_THREADS / L;
    const int lc = threadIdx.x % L, r = threadIdx.x / L;
    const bool active = r < rows;
    const bool is_x = lc * 8 < D;
    const int ch0 = is_x ? lc * 8 : lc * 8 - D;
    float s[8 * F + 8];
#pragma unroll
    for (int j = 0; j < 8 * F + 8; ++j) s[j] = 0.f;
    float wr[8][F];
    if (!is_x) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int k = 0; k < F; ++k) wr[j][k] = wc[(ch0 + j) * F + k];
    }
    const int Lc = (Dtot - D) >> 3;  // lanes of the conditioning half (0 when there is none)
    const int64_t v0 = (int64_t)blockIdx.x * CD_VOX, v1 = min(V, v0 + CD_VOX);
    for (int64_t vb = v0; vb < v1; vb += rows) {
        const int64_t v = vb + r;
        const bool ok = active && v < v1;
        float dcl[F];
#pragma unroll
        for (int k = 0; k < F; ++k) dcl[k] = 0.f;
        // samples in groups of 3: all loads of a group are issued before the arithmetic
        for (int b0 = 0; b0 < B; b0 += 3) {
            Raw8<T> graw[3];
            float in[3][F];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int b = min(b0 + u, B - 1);
                const bool okb = ok && b0 + u < B;
                graw[u].load(dy + ((int64_t)b * V + (okb ? v : v0)) * Dtot + lc * 8);
                const float* src = is_x ? x + (int64_t)b * F * V : c;
#pragma unroll
                for (int k = 0; k < F; ++k) in[u][k] = src[(int64_t)k * V + (okb ? v : v0)];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const bool okb = ok && b0 + u < B;
                Vec8<T> g = graw[u].get();
#pragma unroll
                for (int j = 0; j < 8; ++j) g.v[j] = okb ? g.v[j] : 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
#pragma unroll
                    for (int k = 0; k < F; ++k) s[j * F + k] += g.v[j] * in[u][k];
                    s[8 * F + j] += g.v[j];
                }
                if (!is_x) {
#pragma unroll
                    for (int k = 0; k < F; ++k)
#pragma unroll
                        for (int j = 0; j < 8; ++j) dcl[k] += wr[j][k] * g.v[j];
                }
            }
        }
        if (dc != nullptr && Lc > 0) {
            // sum dcl over the Lc conditioning lanes of this voxel (lanes lc in [L - Lc, L), adjacent)
#pragma unroll
            for (int k = 0; k < F; ++k) {
                float t = is_x ? 0.f : dcl[k];
                for (int o = 1; o < L; o <<= 1) t += __shfl_xor(t, o, 64);  // L is a power of two <= 64
                if (ok && lc == 0) dc[(int64_t)k * V + v] = t;
            }
        }
    }
    // parameter gradients: [ch][F] weights then [ch] bias, per lane-vector
    __shared__ float red[CD_THREADS][8 * F + 8 + 1];
    const int tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < 8 * F + 8; ++j) red[tid][j] = active ? s[j] : 0.f;
    __syncthreads();
    constexpr int N = 8 * F + 8;
    for (int o = tid; o < L * N; o += CD_THREADS) {
        const int l = o / N, j = o - l * N;
        float t = 0.f;
        for (int q = 0; q < rows; ++q) t += red[q * L + l][j];
        const bool lx = l * 8 < D;
        const int c0 = lx ? l * 8 : l * 8 - D;
        float* dw = lx ? dwx : dwc;
        float* db = lx ? dbx : dbc;
        if (j < 8 * F) atomicAdd(&dw[(c0 + j / F) * F + (j % F)], t);
        else atomicAdd(&db[c0 + (j - 8 * F)], t);
    }
}

static bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
static hipError_t codec_zero(float* p, size_t n, hipStream_t st) {  // (a kernel, not hipMemsetAsync: tdx_common.h)
    return tdx_zero_async(p, n * sizeof(float), st) == TDX_OK ? hipSuccess : hipErrorUnknown;
}
// TDX_DETERMINISTIC: `floats` of per-block slabs in the launching stream's scratch arena (behind its zero block), or nullptr
static float* det_slabs(int64_t floats) {
    char* arena = (char*)tdx_scratch_ptr();
    if (!arena || (int64_t)tdx_scratch_bytes() < 256 + floats * (int64_t)sizeof(float)) return nullptr;
    return reinterpret_cast<float*>(arena + 256);
}

extern "C" int tdx_encode_fwd(const float* x, int Fx, const float* wx, const float* bx, const float* c, int Fc,
                              const float* wc, const float* bc, void* y, int B, int64_t V, int D, int dtype,
                              void* stream) {
    TDX_CHECK_ARG(x && wx && bx && y && B > 0 && V > 0 && D > 0);
    TDX_CHECK_ARG(c == nullptr || (wc && bc));
    const int Dtot = c ? 2 * D : D;
    if ((D % 8) || !pow2(Dtot / 8) || Dtot / 8 > 64 || Fx != 4 || (c && Fc != 4)) return TDX_ESHAPE;
    dim3 grid(ceil_div(V, CD_VOX), B);
    TDX_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((encode_fwd_kernel<T, 4>), grid, dim3(CD_THREADS), 0, as_stream(stream),
                                                 x, wx, bx, c, wc, bc, (T*)y, V, D, Dtot));
    return tdx_launch_status();
}

extern "C" int tdx_encode_bwd(const void* dy, const float* x, int Fx, const float* c, int Fc, const float* wc,
                              float* dwx, float* dbx, float* dwc, float* dbc, float* dc, int B, int64_t V, int D,
                              int dtype, void* stream) {
    TDX_CHECK_ARG(dy && x && dwx && dbx && B > 0 && V > 0 && D > 0);
    TDX_CHECK_ARG(c == nullptr || (wc && dwc && dbc));
    const int Dtot = c ? 2 * D : D;
    if ((D % 8) || !pow2(Dtot / 8) || Dtot / 8 > 64 || Fx != 4 || (c && Fc != 4)) return TDX_ESHAPE;
    hipStream_t st = as_stream(stream);
    dim3 grid(ceil_div(V, CD_VOX));
    if (tdx_deterministic()) {
        // slab = [dwx (4 D) | dbx (D) | dwc (4 D) | dbc (D)], one per block, in the scratch arena
        const int64_t slab = 10 * (int64_t)D;
        float* slabs = det_slabs((int64_t)grid.x * slab);
        if (!slabs) return TDX_EINVAL;  // no arena (TDX_SCRATCH_MB=0) or too small: refuse rather than merge in arrival order
        int e0 = tdx_zero_async(slabs, (size_t)grid.x * slab * sizeof(float), st);
        if (e0 != TDX_OK) return e0;
        TDX_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((encode_bwd_kernel<T, 4>), grid, dim3(CD_THREADS), 0, st, (const T*)dy, x, c, wc,
                                                     slabs, slabs + 4 * D, c ? slabs + 5 * D : nullptr,
                                                     c ? slabs + 9 * D : nullptr, dc, B, V, D, Dtot, slab));
        int rc = tdx_launch_status();
        if (rc == TDX_OK) rc = ordered_sum_launch(slabs, (int)grid.x, slab, dwx, 1, 4 * D, 4 * D, false, st);
        if (rc == TDX_OK) rc = ordered_sum_launch(slabs + 4 * D, (int)grid.x, slab, dbx, 1, D, D, false, st);
        if (rc == TDX_OK && c) rc = ordered_sum_launch(slabs + 5 * D, (int)grid.x, slab, dwc, 1, 4 * D, 4 * D, false, st);
        if (rc == TDX_OK && c) rc = ordered_sum_launch(slabs + 9 * D, (int)grid.x, slab, dbc, 1, D, D, false, st);
        return rc;
    }
    hipError_t e = codec_zero(dwx, (size_t)D * 4, st);
    if (e == hipSuccess) e = codec_zero(dbx, (size_t)D, st);
    if (e == hipSuccess && c) e = codec_zero(dwc, (size_t)D * 4, st);
    if (e == hipSuccess && c) e = codec_zero(dbc, (size_t)D, st);
    if (e != hipSuccess) return (int)e;
    TDX_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((encode_bwd_kernel<T, 4>), grid, dim3(CD_THREADS), 0, st, (const T*)dy,
                                                 x, c, wc, dwx, dbx, dwc, dbc, dc, B, V, D, Dtot, (int64_t)0));
    return tdx_launch_status();
}

// ------------------------------------------------------------------ decode ---------------
template <typename T, int F>
__global__ void __launch_bounds__(CD_THREADS)
decode_fwd_kernel(const T* __restrict__ h, const float* __restrict__ w, const float* __restrict__ bias,
                  float* __restrict__ y, int64_t V, int D) {
    const int b = blockIdx.y;
    const int L = D >> 3, rows = CD_THREADS / L;
    const int lc = threadIdx.x % L, r = threadIdx.x / L;
    float wr[F][8];
#pragma unroll
    for (int f = 0; f < F; ++f)
#pragma unroll
        for (int j = 0; j < 8; ++j) wr[f][j] = w[f * D + lc * 8 + j];
    const int64_t v0 = (int64_t)blockIdx.x * CD_VOX_DEC, v1 = min(V, v0 + CD_VOX_DEC);
    for (int64_t vb = v0; vb < v1; vb += rows) {
        const int64_t v = vb + r;
        const bool ok = r < rows && v < v1;
        Vec8<T> a;
#pragma unroll
        for (int j = 0; j < 8; ++j) a.v[j] = 0.f;
        if (ok) a.load(h + ((int64_t)b * V + v) * D + lc * 8);
#pragma unroll
        for (int f = 0; f < F; ++f) {
            float t = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) t = __builtin_fmaf(wr[f][j], a.v[j], t);  // explicit: gn_apply_decode_kernel repeats it
            for (int o = 1; o < L; o <<= 1) t += __shfl_xor(t, o, 64);
            if (ok && lc == 0) y[((int64_t)b * F + f) * V + v] = t + bias[f];
        }
    }
}

template <typename T, int F>
__global__ void __launch_bounds__(CD_THREADS)
decode_bwd_kernel(const float* __restrict__ dy, const T* __restrict__ h, const float* __restrict__ w,
                  T* __restrict__ dh, float* __restrict__ dw, float* __restrict__ db, int64_t V, int D, int64_t slab_stride) {
    const int b = blockIdx.y;
    dw += ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * slab_stride;  // TDX_DETERMINISTIC: one zeroed slab per block
    db += ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * slab_stride;
    const int L = D >> 3, rows = CD_THREADS / L;
    const int lc = threadIdx.x % L, r = threadIdx.x / L;
    const bool active = r < rows;
    float wr[F][8], s[F * 8 + F];
#pragma unroll
    for (int f = 0; f < F; ++f)
#pragma unroll
        for (int j = 0; j < 8; ++j) wr[f][j] = w[f * D + lc * 8 + j];
#pragma unroll
    for (int j = 0; j < F * 8 + F; ++j) s[j] = 0.f;
    const int64_t v0 = (int64_t)blockIdx.x * CD_VOX_DEC, v1 = min(V, v0 + CD_VOX_DEC);
    if (active) {
        for (int64_t v = v0 + r; v < v1; v += rows) {
            float g[F];
#pragma unroll
            for (int f = 0; f < F; ++f) g[f] = dy[((int64_t)b * F + f) * V + v];
            Vec8<T> a, o;
            a.load(h + ((int64_t)b * V + v) * D + lc * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float t = 0.f;
#pragma unroll
                for (int f = 0; f < F; ++f) { t += wr[f][j] * g[f]; s[f * 8 + j] += g[f] * a.v[j]; }
                o.v[j] = t;
            }
            if (lc == 0) {
#pragma unroll
                for (int f = 0; f < F; ++f) s[F * 8 + f] += g[f];
            }
            o.store(dh + ((int64_t)b * V + v) * D + lc * 8);
        }
    }
    __shared__ float red[CD_THREADS][F * 8 + F + 1];
    const int tid = threadIdx.x;
    constexpr int N = F * 8 + F;
#pragma unroll
    for (int j = 0; j < N; ++j) red[tid][j] = active ? s[j] : 0.f;
    __syncthreads();
    for (int o = tid; o < L * N; o += CD_THREADS) {
        const int l = o / N, j = o - l * N;
        float t = 0.f;
        for (int q = 0; q < rows; ++q) t += red[q * L + l][j];
        if (j < F * 8) atomicAdd(&dw[(j / 8) * D + l * 8 + (j % 8)], t);
        else if (l == 0) atomicAdd(&db[j - F * 8], t);
    }
}

extern "C" int tdx_decode_fwd(const void* h, const float* w, const float* bias, float* y, int B, int64_t V, int D, int F,
                              int dtype, void* stream) {
    TDX_CHECK_ARG(h && w && bias && y && B > 0 && V > 0 && D > 0);
    if ((D % 8) || !pow2(D / 8) || D / 8 > 64 || F != 4) return TDX_ESHAPE;
    dim3 grid(ceil_div(V, CD_VOX_DEC), B);
    TDX_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((decode_fwd_kernel<T, 4>), grid, dim3(CD_THREADS), 0, as_stream(stream),
                                                 (const T*)h, w, bias, y, V, D));
    return tdx_launch_status();
}

extern "C" int tdx_decode_bwd(const float* dy, const void* h, const float* w, void* dh, float* dw, float* db, int B,
                              int64_t V, int D, int F, int dtype, void* stream) {
    TDX_CHECK_ARG(dy && h && w && dh && dw && db && B > 0 && V > 0 && D > 0);
    if ((D % 8) || !pow2(D / 8) || D / 8 > 64 || F != 4) return TDX_ESHAPE;
    hipStream_t st = as_stream(stream);
    dim3 grid(ceil_div(V, CD_VOX_DEC), B);
    if (tdx_deterministic()) {
        const int64_t slab = (int64_t)F * D + F, nblk = (int64_t)grid.x * grid.y;  // [dw (F D) | db (F)] per block
        float* slabs = det_slabs(nblk * slab);
        if (!slabs) return TDX_EINVAL;
        int e0 = tdx_zero_async(slabs, (size_t)nblk * slab * sizeof(float), st);
        if (e0 != TDX_OK) return e0;
        TDX_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((decode_bwd_kernel<T, 4>), grid, dim3(CD_THREADS), 0, st, dy, (const T*)h, w,
                                                     (T*)dh, slabs, slabs + (int64_t)F * D, V, D, slab));
        int rc = tdx_launch_status();
        if (rc == TDX_OK) rc = ordered_sum_launch(slabs, (int)nblk, slab, dw, 1, F * D, F * D, false, st);
        if (rc == TDX_OK) rc = ordered_sum_launch(slabs + (int64_t)F * D, (int)nblk, slab, db, 1, F, F, false, st);
        return rc;
    }
    hipError_t e = codec_zero(dw, (size_t)F * D, st);
    if (e == hipSuccess) e = codec_zero(db, (size_t)F, st);
    if (e != hipSuccess) return (int)e;
    TDX_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((decode_bwd_kernel<T, 4>), grid, dim3(CD_THREADS), 0, st, dy,
                                                 (const T*)h, w, (T*)dh, dw, db, V, D, (int64_t)0));
    return tdx_launch_status();
}
