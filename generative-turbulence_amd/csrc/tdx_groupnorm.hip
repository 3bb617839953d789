// GroupNorm (+ FiLM + SiLU + residual) forward / backward on NDHWC activations.
//
// HBM-bound: every pass streams the activation once with 8 channels (16 B bf16 / 32 B f32)
// per lane.  A thread always owns the same 8-channel vector (tid % (C/8)), so per-channel
// coefficients live in registers.  The streaming kernels are grid-stride with four voxels per
// thread per trip: the trip's loads are issued back to back (sched_barrier keeps them ahead of
// the arithmetic), sigmoid uses the hardware exp2 / rcp, bf16 results are packed with
// v_cvt_pk_bf16_f32 -- at 9-16 VALU instructions per element the kernels stay memory-bound
// (4.6-5.4 TB/s on the level-0 tensors; a plain copy reaches 5.6-6.4 TB/s on this chip).
//   forward : statistics (normally accumulated in the conv epilogue, tdx_conv3_fwd_gn; the
//             stand-alone stats pass uses f64 atomics) -> per group mean/rstd -> apply pass
//   backward: reduce pass (P = sum dn, Q = sum dn*xhat per channel; per-block partial sums, no
//             atomics, deterministic) -> finalize -> apply pass
#define TDX_NT_LOADS 1  // activations are streamed once per pass: nontemporal 16-B loads (+0.4 % step)
#include "tdx_common.h"
#include "tdx_conv3.h"

#define GN_THREADS 256
#define GN_MAX_BLOCKS 512  // streaming blocks per sample; bounds the backward partial-sum buffer
#define GN_VOX_PER_BLOCK 1024

// per-channel partial sums of two quantities, reduced over the block and added (f64 atomics)
// into acc[(b*C + c)*2 + {0,1}]
template <int NQ>
// store (TDX_DETERMINISTIC): `acc` is this block's own table -- the partial is stored, not added; gn_stats_merge_kernel adds
// the blocks' tables in block order
__device__ __forceinline__ void block_channel_reduce(float (&s)[NQ][8], int L, int lane_c, bool active,
                                                     double* __restrict__ acc, int C, bool store = false) {
    __shared__ float red[GN_THREADS][NQ * 8 + 1];
    const int tid = threadIdx.x;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int j = 0; j < 8; ++j) red[tid][q * 8 + j] = active ? s[q][j] : 0.f;
    __syncthreads();
    const int rows = GN_THREADS / L;  // threads tid < rows*L are the active ones
    // C*NQ outputs, each summed over `rows` entries
    for (int o = tid; o < C * NQ; o += GN_THREADS) {
        const int q = o / C, c = o - q * C;
        const int lc = c >> 3, j = c & 7;
        double t = 0.0;
        for (int r = 0; r < rows; ++r) t += (double)red[r * L + lc][q * 8 + j];
        if (store) acc[(size_t)c * NQ + q] = t;
        else atomicAdd(&acc[(size_t)c * NQ + q], t);
    }
    (void)lane_c;
}

template <typename T>
__global__ void __launch_bounds__(GN_THREADS)
gn_stats_kernel(const T* __restrict__ x, double* __restrict__ acc, int64_t V, int C, int vpb, double* __restrict__ part) {
    const int b = blockIdx.y;
    const int L = C >> 3;
    const int rows = GN_THREADS / L;
    const int tid = threadIdx.x;
    const int lc = tid % L, r = tid / L;
    const bool active = r < rows;
    const int64_t v0 = (int64_t)blockIdx.x * vpb;
    const int64_t v1 = min(V, v0 + vpb);
    float s[2][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[0][j] = s[1][j] = 0.f;
    if (active) {
        const T* xb = x + ((int64_t)b * V) * C + lc * 8;
        int64_t v = v0 + r;
        // four independent loads in flight per trip (a rolled load / add loop is a chain of memory round trips: this pass
        // runs on the tiny tensors of the deep levels, where latency is all it costs)
        for (; v + 3 * rows < v1; v += 4 * rows) {
            Raw8<T> a[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u].load(xb + (v + u * rows) * C);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const Vec8<T> t = a[u].get();
#pragma unroll
                for (int j = 0; j < 8; ++j) { s[0][j] += t.v[j]; s[1][j] += t.v[j] * t.v[j]; }
            }
        }
        for (; v < v1; v += rows) {
            Vec8<T> a;
            a.load(xb + v * C);
#pragma unroll
            for (int j = 0; j < 8; ++j) { s[0][j] += a.v[j]; s[1][j] += a.v[j] * a.v[j]; }
        }
    }
    if (part) block_channel_reduce<2>(s, L, lc, active, part + ((size_t)b * gridDim.x + blockIdx.x) * C * 2, C, true);
    else block_channel_reduce<2>(s, L, lc, active, acc + (size_t)b * C * 2, C);
}

// TDX_DETERMINISTIC: acc[b][c][q] = sum over the sample's blocks of part[b][blk][c][q], in block order (f64)
__global__ void __launch_bounds__(256) gn_stats_merge_kernel(const double* __restrict__ part, double* __restrict__ acc, int nblk,
                                                            int C2, int total) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int b = e / C2, r = e - b * C2;
    const double* p = part + (size_t)b * nblk * C2 + r;
    double v = 0.0;
    int k = 0;
    for (; k + 4 <= nblk; k += 4) {  // four loads in flight, added in block order
        const double a0 = p[(size_t)k * C2], a1 = p[(size_t)(k + 1) * C2], a2 = p[(size_t)(k + 2) * C2], a3 = p[(size_t)(k + 3) * C2];
        v = (((v + a0) + a1) + a2) + a3;
    }
    for (; k < nblk; ++k) v += p[(size_t)k * C2];
    acc[e] = v;
}

// per (b, g): mean / rstd from per-channel (sum, sumsq)
// acc holds R replicas of [B][C][2] (replicas spread the atomics of the fused conv epilogue).
// One wave per (b, group): lanes split the R * cpg (replica, channel) pairs.
__global__ void __launch_bounds__(64)
gn_stats_finalize(double* __restrict__ acc, float* __restrict__ stats, int B, int C, int G, int64_t V, float eps,
                  int R) {
    const int i = blockIdx.x;  // b * G + g
    const int b = i / G, g = i - b * G;
    const int cpg = C / G;
    double s = 0.0, ss = 0.0;
    // four (replica, channel) pairs per trip, their loads issued before the first add / clearing store (the stores may
    // alias the loads as far as the compiler knows: a rolled loop is one memory round trip per pair)
    for (int k0 = threadIdx.x; k0 < R * cpg; k0 += 4 * 64) {
        double2 v[4];
        double* a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = min(k0 + u * 64, R * cpg - 1);
            const int r = k / cpg, c = g * cpg + (k - r * cpg);
            a[u] = acc + (((size_t)r * B + b) * C + c) * 2;
            v[u] = *reinterpret_cast<const double2*>(a[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (k0 + u * 64 < R * cpg) {
                s += v[u].x;
                ss += v[u].y;
                *reinterpret_cast<double2*>(a[u]) = make_double2(0.0, 0.0);  // left all-zero for the next use (TDX_WS_CLEAN)
            }
    }
    s = wave_sum(s);
    ss = wave_sum(ss);
    if (threadIdx.x == 0) {
        const double n = (double)cpg * (double)V;
        const double mean = s / n;
        double var = ss / n - mean * mean;
        if (var < 0.0) var = 0.0;
        stats[2 * i] = (float)mean;
        stats[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

#define GN_REPLICAS 32
// covers both uses: GN_REPLICAS f64 moment tables (+ stats) of the fused conv epilogue, and the
// backward's [B][C][2] f64 sums + [B][C][2] group sums + GN_MAX_BLOCKS per-block partial tables
extern "C" size_t tdx_gn_workspace_bytes(int B, int C) {
    const size_t fwd = (size_t)GN_REPLICAS * B * C * 2 * sizeof(double) + (size_t)B * C * 2 * sizeof(float) + 64;
    const size_t bwd = (size_t)B * C * 2 * sizeof(double) + (size_t)B * C * 2 * sizeof(float) +
                       (size_t)512 * B * C * 2 * sizeof(float) + 64;
    return fwd > bwd ? fwd : bwd;
}

static int gn_shape_ok(int C, int G) { return C > 0 && G > 0 && C % 8 == 0 && C % G == 0 && (C / 8) <= GN_THREADS; }

// used by tdx_conv3_fwd_gn (statistics accumulated in the conv epilogue)
int gn_finalize_launch(double* acc, float* stats, int B, int C, int G, int64_t V, float eps, int replicas,
                       hipStream_t st) {
    hipLaunchKernelGGL(gn_stats_finalize, dim3(B * G), dim3(64), 0, st, acc, stats, B, C, G, V, eps, replicas);
    return tdx_launch_status();
}

int gn_stats_launch(const void* x, float* stats, int B, int64_t V, int C, int G, float eps, int dtype, void* workspace,
                    bool clean, hipStream_t stream);  // also declared in tdx_conv3.h for tdx_conv3_fwd_gn

extern "C" int tdx_gn_stats(const void* x, float* stats, int B, int64_t V, int C, int G, float eps, int dtype,
                            void* workspace, void* stream) {
    return gn_stats_launch(x, stats, B, V, C, G, eps, dtype, workspace, false, as_stream(stream));
}

// clean: the caller vouches that the accumulators are all-zero (TDX_WS_CLEAN: a workspace zeroed once and only ever
// used by launches whose finalize pass re-zeroes what it read) -- no memset launch then
int gn_stats_launch(const void* x, float* stats, int B, int64_t V, int C, int G, float eps, int dtype, void* workspace,
                    bool clean, hipStream_t stream) {
    TDX_CHECK_ARG(x && stats && workspace && B > 0 && V > 0);
    if (!gn_shape_ok(C, G)) return TDX_ESHAPE;
    double* acc = (double*)workspace;
    if (!clean) {
        int e = tdx_zero_async(acc, (size_t)B * C * 2 * sizeof(double), stream);
        if (e != TDX_OK) return e;
    }
    // voxels per block: 1024 on big tensors; on the small ones of the deep U-Net levels (where this pass is used: the
    // small-grid conv kernels do not accumulate moments) enough blocks to put ~128 on the chip -- 12 blocks of 256 serial
    // trips took 27 us for 7 MB, profiles/r11_batch_scaling.txt -- but not so many that the f64 atomics take over
    int64_t vpb = ((int64_t)B * V + 127) / 128;
    vpb = vpb < 32 ? 32 : (vpb > GN_VOX_PER_BLOCK ? GN_VOX_PER_BLOCK : (vpb + 31) / 32 * 32);
    double* part = nullptr;
    if (tdx_deterministic()) {
        // no f64 atomics in arrival order: at most 64 blocks per sample, each stores its own table in the scratch arena (one
        // writer per element), gn_stats_merge_kernel adds them in block order
        if (ceil_div(V, vpb) > 64) vpb = (ceil_div(V, 64) + 31) / 32 * 32;
        const size_t need = (size_t)B * ceil_div(V, vpb) * C * 2 * sizeof(double);
        if (tdx_scratch_ptr() != nullptr && tdx_scratch_bytes() >= 256 + need) part = reinterpret_cast<double*>((char*)tdx_scratch_ptr() + 256);
        else vpb = (V + 31) / 32 * 32;  // no arena (TDX_SCRATCH_MB=0): ONE block per sample adds into the zeroed table -- slow, still ordered
    }
    dim3 grid(ceil_div(V, vpb), B);
    TDX_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((gn_stats_kernel<T>), grid, dim3(GN_THREADS), 0, stream,
                                                  (const T*)x, acc, V, C, (int)vpb, part));
    if (part) {
        const int total = B * C * 2;
        hipLaunchKernelGGL(gn_stats_merge_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, stream, part, acc, (int)grid.x, C * 2, total);
    }
    hipLaunchKernelGGL(gn_stats_finalize, dim3(B * G), dim3(64), 0, stream, acc, stats, B, C, G, V, eps, 1);
    return tdx_launch_status();
}

// per-thread affine coefficients of its 8 channels: n = x*a + c0  (FiLM folded in)
struct GnCoef {
    float a[8], c0[8];
};
__device__ __forceinline__ void gn_load_coef(GnCoef& k, float (&mean)[8], float (&rstd)[8], float (&gam)[8],
                                             float (&film)[8], const float* __restrict__ stats,
                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                             const float* __restrict__ scale, const float* __restrict__ shift, int b,
                                             int C, int G, int cbase) {
    const int cpg = C / G;
    // issue every load before the first use (one exposed latency per block instead of one per channel)
    float be[8], sc[8], sh[8];
    float2 st[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) st[j] = *reinterpret_cast<const float2*>(stats + ((size_t)b * G + (cbase + j) / cpg) * 2);
    {
        const float4 g0 = *reinterpret_cast<const float4*>(gamma + cbase), g1 = *reinterpret_cast<const float4*>(gamma + cbase + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(beta + cbase), b1 = *reinterpret_cast<const float4*>(beta + cbase + 4);
        gam[0] = g0.x; gam[1] = g0.y; gam[2] = g0.z; gam[3] = g0.w; gam[4] = g1.x; gam[5] = g1.y; gam[6] = g1.z; gam[7] = g1.w;
        be[0] = b0.x; be[1] = b0.y; be[2] = b0.z; be[3] = b0.w; be[4] = b1.x; be[5] = b1.y; be[6] = b1.z; be[7] = b1.w;
    }
    if (scale) {
        const float* sp = scale + (size_t)b * C + cbase;
        const float* hp = shift + (size_t)b * C + cbase;
        const float4 s0 = *reinterpret_cast<const float4*>(sp), s1 = *reinterpret_cast<const float4*>(sp + 4);
        const float4 h0 = *reinterpret_cast<const float4*>(hp), h1 = *reinterpret_cast<const float4*>(hp + 4);
        sc[0] = s0.x; sc[1] = s0.y; sc[2] = s0.z; sc[3] = s0.w; sc[4] = s1.x; sc[5] = s1.y; sc[6] = s1.z; sc[7] = s1.w;
        sh[0] = h0.x; sh[1] = h0.y; sh[2] = h0.z; sh[3] = h0.w; sh[4] = h1.x; sh[5] = h1.y; sh[6] = h1.z; sh[7] = h1.w;
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) { sc[j] = 0.f; sh[j] = 0.f; }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        mean[j] = st[j].x;
        rstd[j] = st[j].y;
        film[j] = 1.0f + sc[j];
        k.a[j] = rstd[j] * gam[j] * film[j];
        k.c0[j] = (be[j] - mean[j] * rstd[j] * gam[j]) * film[j] + sh[j];
    }
}

// Streaming skeleton of the apply / backward passes: block (bx, b) walks the voxels of sample b
// with stride gridDim.x * rows, GN_UNROLL voxels per thread per trip (independent 16-B loads in
// flight, no branches between them); the grid is sized to about 8 resident blocks per CU so the
// per-block set-up (coefficient loads) is paid once per ~MB streamed.
#define GN_UNROLL 4

static int gn_blocks_per_sample(int B, int64_t V, int C) {
    const int rows = GN_THREADS / (C >> 3);
    int64_t want = (2048 + B - 1) / B;
    const int64_t most = (V + (int64_t)rows * GN_UNROLL - 1) / ((int64_t)rows * GN_UNROLL);
    if (want > most) want = most;
    if (want > GN_MAX_BLOCKS) want = GN_MAX_BLOCKS;
    return want < 1 ? 1 : (int)want;
}

// silu(n) + r as ONE explicit fma: the three tail kernels (apply / apply_encoded / apply_decode) promise bit-identical
// results, which must not hinge on the compiler contracting n * sigmoid(n) + r the same way in each
__device__ __forceinline__ float silu_add(float n, float r) { return __builtin_fmaf(n, sigmoid_f(n), r); }

template <typename T, bool HAS_RES, bool ACT>
__global__ void __launch_bounds__(GN_THREADS)
gn_apply_kernel(const T* __restrict__ x, const float* __restrict__ stats, const float* __restrict__ gamma,
                const float* __restrict__ beta, const float* __restrict__ scale, const float* __restrict__ shift,
                const T* __restrict__ res, T* __restrict__ y, int64_t V, int C, int G) {
    const int b = blockIdx.y;
    const int L = C >> 3;
    const int rows = GN_THREADS / L;
    const int tid = threadIdx.x;
    const int lc = tid % L, r = tid / L;
    if (r >= rows) return;
    GnCoef k;
    float mean[8], rstd[8], gam[8], film[8];
    gn_load_coef(k, mean, rstd, gam, film, stats, gamma, beta, scale, shift, b, C, G, lc * 8);
    const int64_t base = ((int64_t)b * V) * C + lc * 8;
    const int64_t stride = (int64_t)gridDim.x * rows;
    auto one = [&](const Vec8<T>& a, const Vec8<T>& rr, int64_t vv) {
        Vec8<T> o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float n = __builtin_fmaf(a.v[j], k.a[j], k.c0[j]);
            if (ACT && HAS_RES) o.v[j] = silu_add(n, rr.v[j]);
            else {
                o.v[j] = ACT ? silu_f(n) : n;
                if (HAS_RES) o.v[j] += rr.v[j];
            }
        }
        o.store(y + base + vv * C);
    };
    int64_t v = (int64_t)blockIdx.x * rows + r;
    // full trips: GN_UNROLL independent loads in flight, no branches between them
    for (; v + (GN_UNROLL - 1) * stride < V; v += stride * GN_UNROLL) {
        Raw8<T> a[GN_UNROLL], rr[GN_UNROLL];
#pragma unroll
        for (int u = 0; u < GN_UNROLL; ++u) {
            a[u].load(x + base + (v + u * stride) * C);
            if (HAS_RES) rr[u].load(res + base + (v + u * stride) * C);
        }
        __builtin_amdgcn_sched_barrier(0);  // all loads of the trip are issued before any arithmetic
#pragma unroll
        for (int u = 0; u < GN_UNROLL; ++u) one(a[u].get(), HAS_RES ? rr[u].get() : Vec8<T>(), v + u * stride);
    }
    for (; v < V; v += stride) {
        Vec8<T> a, rr;
        a.load(x + base + v * C);
        if (HAS_RES) rr.load(res + base + v * C);
        one(a, rr, v);
    }
}

extern "C" int tdx_gn_apply(const void* x, const float* stats, const float* gamma, const float* beta,
                            const float* scale, const float* shift, const void* res, void* y, int B, int64_t V, int C,
                            int G, int act, int dtype, void* stream) {
    TDX_CHECK_ARG(x && stats && gamma && beta && y && B > 0 && V > 0);
    TDX_CHECK_ARG((scale == nullptr) == (shift == nullptr));
    if (!gn_shape_ok(C, G)) return TDX_ESHAPE;
    dim3 grid(gn_blocks_per_sample(B, V, C), B);
#define GN_APPLY_GO(R, A)                                                                                              \
    TDX_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((gn_apply_kernel<T, R, A>), grid, dim3(GN_THREADS), 0,                \
                                                  as_stream(stream), (const T*)x, stats, gamma, beta, scale, shift,    \
                                                  (const T*)res, (T*)y, V, C, G))
    if (res) {
        if (act) GN_APPLY_GO(true, true); else GN_APPLY_GO(true, false);
    } else {
        if (act) GN_APPLY_GO(false, true); else GN_APPLY_GO(false, false);
    }
#undef GN_APPLY_GO
    return tdx_launch_status();
}

// ---- y = silu(GN(x)) + encode(raw): the tail of the U-Net's FIRST block, whose identity skip is the encoder output
// cat(Wx x_raw + bx, Wc c_raw + bc) (ddpm.py:495-501).  The skip is recomputed here from the 4 + 4 raw f32 planes instead
// of being written by tdx_encode_fwd and read back: per voxel 2 x (C x 2 B) of HBM traffic become 8 x 4 B.  The skip value
// is rounded to T before the add, exactly what a stored encoder output would have held, so the result is bit-identical
// to tdx_encode_fwd + tdx_gn_apply(res = that tensor).
template <typename T> __device__ __forceinline__ float round_as(float a);
template <> __device__ __forceinline__ float round_as<float>(float a) { return a; }
template <> __device__ __forceinline__ float round_as<bf16>(float a) { return __uint_as_float(pack_bf16x2(a, 0.f) << 16); }
template <> __device__ __forceinline__ float round_as<f16>(float a) { return (float)(_Float16)a; }

template <typename T, int F>
__global__ void __launch_bounds__(GN_THREADS)
gn_apply_encoded_kernel(const T* __restrict__ x, const float* __restrict__ stats, const float* __restrict__ gamma,
                        const float* __restrict__ beta, const float* __restrict__ xr, const float* __restrict__ wx,
                        const float* __restrict__ bx, const float* __restrict__ cr, const float* __restrict__ wc,
                        const float* __restrict__ bc, T* __restrict__ y, int64_t V, int C, int D, int G) {
    const int b = blockIdx.y;
    const int L = C >> 3;
    const int rows = GN_THREADS / L;
    const int tid = threadIdx.x;
    const int lc = tid % L, r = tid / L;
    if (r >= rows) return;
    GnCoef k;
    float mean[8], rstd[8], gam[8], film[8];
    gn_load_coef(k, mean, rstd, gam, film, stats, gamma, beta, nullptr, nullptr, b, C, G, lc * 8);
    const bool is_x = lc * 8 < D;
    const int ch0 = is_x ? lc * 8 : lc * 8 - D;
    const float* w = is_x ? wx : wc;
    const float* bias = is_x ? bx : bc;
    const float* src = is_x ? xr + (int64_t)b * F * V : cr;
    float wr[8][F], br[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        br[j] = bias[ch0 + j];
#pragma unroll
        for (int q = 0; q < F; ++q) wr[j][q] = w[(ch0 + j) * F + q];
    }
    const int64_t base = ((int64_t)b * V) * C + lc * 8;
    const int64_t stride = (int64_t)gridDim.x * rows;
    auto one = [&](const Vec8<T>& a, const float (&in)[F], int64_t vv) {
        Vec8<T> o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float e = br[j];
#pragma unroll
            for (int q = 0; q < F; ++q) e = __builtin_fmaf(wr[j][q], in[q], e);  // encode_fwd_kernel's chain, bit for bit
            const float n = __builtin_fmaf(a.v[j], k.a[j], k.c0[j]);
            o.v[j] = silu_add(n, round_as<T>(e));
        }
        o.store(y + base + vv * C);
    };
    int64_t v = (int64_t)blockIdx.x * rows + r;
    for (; v + (GN_UNROLL - 1) * stride < V; v += stride * GN_UNROLL) {
        Raw8<T> a[GN_UNROLL];
        float in[GN_UNROLL][F];
#pragma unroll
        for (int u = 0; u < GN_UNROLL; ++u) {
            a[u].load(x + base + (v + u * stride) * C);
#pragma unroll
            for (int q = 0; q < F; ++q) in[u][q] = src[(int64_t)q * V + v + u * stride];
        }
        __builtin_amdgcn_sched_barrier(0);  // all loads of the trip are issued before any arithmetic
#pragma unroll
        for (int u = 0; u < GN_UNROLL; ++u) one(a[u].get(), in[u], v + u * stride);
    }
    for (; v < V; v += stride) {
        Vec8<T> a;
        float in[F];
        a.load(x + base + v * C);
#pragma unroll
        for (int q = 0; q < F; ++q) in[q] = src[(int64_t)q * V + v];
        one(a, in, v);
    }
}

extern "C" int tdx_gn_apply_encoded(const void* x, const float* stats, const float* gamma, const float* beta,
                                    const float* x_raw, int Fx, const float* wx, const float* bx, const float* c_raw,
                                    int Fc, const float* wc, const float* bc, void* y, int B, int64_t V, int D, int G,
                                    int dtype, void* stream) {
    TDX_CHECK_ARG(x && stats && gamma && beta && x_raw && wx && bx && y && B > 0 && V > 0 && D > 0);
    TDX_CHECK_ARG(c_raw == nullptr || (wc && bc));
    const int C = c_raw ? 2 * D : D;
    if (!gn_shape_ok(C, G) || (D % 8) || Fx != 4 || (c_raw && Fc != 4)) return TDX_ESHAPE;
    dim3 grid(gn_blocks_per_sample(B, V, C), B);
    TDX_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((gn_apply_encoded_kernel<T, 4>), grid, dim3(GN_THREADS), 0,
                                                 as_stream(stream), (const T*)x, stats, gamma, beta, x_raw, wx, bx, c_raw,
                                                 wc, bc, (T*)y, V, C, D, G));
    return tdx_launch_status();
}

// ---- out = decode(silu(GN(x)) + res): the tail of the model's LAST ResnetBlock (decode[0], ddpm.py:429) followed by the
// dim -> F 1x1 decoder and the NDHWC -> NCDHW layout change (decode[1], ddpm.py:505), inference only.  The block output
// (B, V, C) is neither written nor read back: per voxel 2 x (C x 2 B) of HBM traffic disappear.  The block output is
// rounded to T before the dot product and the dot product is decode_fwd_kernel's explicit FMA chain + the same butterfly,
// so the result is bit-identical to tdx_gn_apply(res, act = 1) + tdx_decode_fwd.
template <typename T, int F>
__global__ void __launch_bounds__(GN_THREADS)
gn_apply_decode_kernel(const T* __restrict__ x, const float* __restrict__ stats, const float* __restrict__ gamma,
                       const float* __restrict__ beta, const T* __restrict__ res, const float* __restrict__ w,
                       const float* __restrict__ bias, float* __restrict__ out, int64_t V, int C, int G) {
    const int b = blockIdx.y;
    const int L = C >> 3;  // a power of two <= 64 (checked by the host): the L lanes of a voxel are adjacent in a wave
    const int rows = GN_THREADS / L;
    const int tid = threadIdx.x;
    const int lc = tid % L, r = tid / L;
    GnCoef k;
    float mean[8], rstd[8], gam[8], film[8];
    gn_load_coef(k, mean, rstd, gam, film, stats, gamma, beta, nullptr, nullptr, b, C, G, lc * 8);
    float wr[F][8], bf[F];
#pragma unroll
    for (int f = 0; f < F; ++f) {
        bf[f] = bias[f];
#pragma unroll
        for (int j = 0; j < 8; ++j) wr[f][j] = w[f * C + lc * 8 + j];
    }
    const int64_t base = ((int64_t)b * V) * C + lc * 8;
    const int64_t stride = (int64_t)gridDim.x * rows;
    auto one = [&](const Vec8<T>& a, const Vec8<T>& rr, int64_t vv, bool ok) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float n = __builtin_fmaf(a.v[j], k.a[j], k.c0[j]);
            o[j] = round_as<T>(silu_add(n, rr.v[j]));
        }
#pragma unroll
        for (int f = 0; f < F; ++f) {
            float t = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) t = __builtin_fmaf(wr[f][j], o[j], t);
            for (int q = 1; q < L; q <<= 1) t += __shfl_xor(t, q, 64);
            if (ok && lc == 0) out[((int64_t)b * F + f) * V + vv] = t + bf[f];
        }
    };
    // every lane of a wave takes part in the butterflies: the loop bound is uniform per voxel group, and lanes past the
    // end (or the spare threads when 256 % L != 0 -- not with L a power of two) compute on zeros
    int64_t v = (int64_t)blockIdx.x * rows + r;
    for (; v + (GN_UNROLL - 1) * stride < V; v += stride * GN_UNROLL) {
        Raw8<T> a[GN_UNROLL], rr[GN_UNROLL];
#pragma unroll
        for (int u = 0; u < GN_UNROLL; ++u) {
            a[u].load(x + base + (v + u * stride) * C);
            rr[u].load(res + base + (v + u * stride) * C);
        }
        __builtin_amdgcn_sched_barrier(0);  // all loads of the trip are issued before any arithmetic
#pragma unroll
        for (int u = 0; u < GN_UNROLL; ++u) one(a[u].get(), rr[u].get(), v + u * stride, true);
    }
    for (; v < V; v += stride) {
        Vec8<T> a, rr;
        a.load(x + base + v * C);
        rr.load(res + base + v * C);
        one(a, rr, v, true);
    }
}

extern "C" int tdx_gn_apply_decode(const void* x, const float* stats, const float* gamma, const float* beta, const void* res,
                                   const float* w, const float* bias, float* out, int B, int64_t V, int C, int G, int F,
                                   int dtype, void* stream) {
    TDX_CHECK_ARG(x && stats && gamma && beta && res && w && bias && out && B > 0 && V > 0);
    const int L = C / 8;
    if (!gn_shape_ok(C, G) || F != 4 || L > 64 || (L & (L - 1)) != 0) return TDX_ESHAPE;
    dim3 grid(gn_blocks_per_sample(B, V, C), B);
    TDX_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((gn_apply_decode_kernel<T, 4>), grid, dim3(GN_THREADS), 0,
                                                 as_stream(stream), (const T*)x, stats, gamma, beta, (const T*)res, w, bias,
                                                 out, V, C, G));
    return tdx_launch_status();
}

// ------------------------------------------------------------------ backward -------------
// With n = a x + c0 (GroupNorm affine + FiLM folded), dn = dy act'(n), xhat = (x - mean) rstd:
//   reduce pass : P[b,c] = sum_v dn, Q[b,c] = sum_v dn xhat     (per-block partials, no atomics)
//   finalize    : per (b, group) A = mean_g(k P), Bq = mean_g(k Q), k = gamma (1 + scale); parameter grads
//   apply pass  : dx = rstd (k dn - A - xhat Bq)
// partial[(b * nblk + blk) * C + c][2] holds one block's sums; the finalize kernels add them in f64
// in a fixed order, so the result is deterministic.
template <typename T, bool ACT>
__global__ void __launch_bounds__(GN_THREADS)
gn_bwd_reduce_kernel(const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ stats,
                     const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ scale,
                     const float* __restrict__ shift, float* __restrict__ partial, int64_t V, int C, int G) {
    __shared__ float red[GN_THREADS][17];
    const int b = blockIdx.y;
    const int L = C >> 3;
    const int rows = GN_THREADS / L;
    const int tid = threadIdx.x;
    const int lc = tid % L, r = tid / L;
    const bool active = r < rows;
    float s[2][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[0][j] = s[1][j] = 0.f;
    if (active) {
        GnCoef k;
        float mean[8], rstd[8], gam[8], film[8], mr[8];
        gn_load_coef(k, mean, rstd, gam, film, stats, gamma, beta, scale, shift, b, C, G, lc * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) mr[j] = -mean[j] * rstd[j];
        const int64_t base = ((int64_t)b * V) * C + lc * 8;
        const int64_t stride = (int64_t)gridDim.x * rows;
        auto one = [&](const Vec8<T>& a, const Vec8<T>& g) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float n = __builtin_fmaf(a.v[j], k.a[j], k.c0[j]);
                const float dn = ACT ? g.v[j] * dsilu_f(n) : g.v[j];
                const float xh = __builtin_fmaf(a.v[j], rstd[j], mr[j]);
                s[0][j] += dn;
                s[1][j] = __builtin_fmaf(dn, xh, s[1][j]);
            }
        };
        int64_t v = (int64_t)blockIdx.x * rows + r;
        for (; v + (GN_UNROLL - 1) * stride < V; v += stride * GN_UNROLL) {
            Raw8<T> a[GN_UNROLL], g[GN_UNROLL];
#pragma unroll
            for (int u = 0; u < GN_UNROLL; ++u) {
                a[u].load(x + base + (v + u * stride) * C);
                g[u].load(dy + base + (v + u * stride) * C);
            }
        __builtin_amdgcn_sched_barrier(0);  // all loads of the trip are issued before any arithmetic
#pragma unroll
            for (int u = 0; u < GN_UNROLL; ++u) one(a[u].get(), g[u].get());
        }
        for (; v < V; v += stride) {
            Vec8<T> a, g;
            a.load(x + base + v * C);
            g.load(dy + base + v * C);
            one(a, g);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        red[tid][j] = s[0][j];
        red[tid][8 + j] = s[1][j];
    }
    __syncthreads();
    float* out = partial + ((size_t)b * gridDim.x + blockIdx.x) * C * 2;
    for (int o = tid; o < C * 2; o += GN_THREADS) {
        const int c = o >> 1, q = o & 1;
        const int lcc = c >> 3, j = c & 7;
        float t = 0.f;
        for (int rr = 0; rr < rows; ++rr) t += red[rr * L + lcc][q * 8 + j];
        out[o] = t;
    }
}

// One block per (b, group): sums the per-block partials of the group's channels (f64, fixed order) into
// acc[(b*C + c)*2 + {P, Q}] for the parameter-gradient kernel and forms the group sums
// gsum[(b*G + g)*2] = mean_g(k P), mean_g(k Q) with k = gamma (1 + scale).
__global__ void __launch_bounds__(256)
gn_bwd_group_kernel(const float* __restrict__ partial, double* __restrict__ acc, const float* __restrict__ gamma,
                    const float* __restrict__ scale, float* __restrict__ gsum, int nblk, int C, int G, int64_t V) {
    __shared__ double sA[4], sB[4];
    const int i = blockIdx.x;  // b * G + g
    const int b = i / G, g = i - b * G;
    const int cpg = C / G;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double A = 0.0, Bq = 0.0;
    for (int cc = wave; cc < cpg; cc += 4) {
        const int c = g * cpg + cc;
        double p = 0.0, q = 0.0;
        // nblk <= GN_MAX_BLOCKS = 8 x 64: all of a lane's loads go out together (fixed summation order as before)
        float2 t[GN_MAX_BLOCKS / 64];
#pragma unroll
        for (int i = 0; i < GN_MAX_BLOCKS / 64; ++i) {
            const int k = lane + 64 * i;
            t[i] = k < nblk ? *reinterpret_cast<const float2*>(partial + (((size_t)b * nblk + k) * C + c) * 2) : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < GN_MAX_BLOCKS / 64; ++i) {
            p += (double)t[i].x;
            q += (double)t[i].y;
        }
        p = wave_sum(p);
        q = wave_sum(q);
        if (lane == 0) {
            acc[((size_t)b * C + c) * 2] = p;
            acc[((size_t)b * C + c) * 2 + 1] = q;
        }
        const double kk = (double)gamma[c] * (scale ? 1.0 + (double)scale[(size_t)b * C + c] : 1.0);
        A += kk * p;
        Bq += kk * q;
    }
    if (lane == 0) { sA[wave] = A; sB[wave] = Bq; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double n = (double)cpg * (double)V;
        gsum[2 * i] = (float)((sA[0] + sA[1] + sA[2] + sA[3]) / n);
        gsum[2 * i + 1] = (float)((sB[0] + sB[1] + sB[2] + sB[3]) / n);
    }
}
// Parameter gradients from the channel sums: dgamma / dbeta summed over the samples, dscale / dshift per sample.
// Runs as the prologue of ONE workgroup of the apply pass (block (0, 0)) instead of as a launch of its own: at 22
// GroupNorm backwards per training step the launch count is what these tiny kernels cost.
__device__ __forceinline__ void gn_bwd_params(const double* __restrict__ acc, const float* __restrict__ gamma,
                                              const float* __restrict__ beta, const float* __restrict__ scale,
                                              float* __restrict__ dgamma, float* __restrict__ dbeta,
                                              float* __restrict__ dscale, float* __restrict__ dshift, int B, int C) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        double dg = 0.0, db = 0.0;
        for (int b = 0; b < B; ++b) {
            const double P = acc[((size_t)b * C + c) * 2], Q = acc[((size_t)b * C + c) * 2 + 1];
            const double f = scale ? 1.0 + (double)scale[(size_t)b * C + c] : 1.0;
            dg += f * Q;
            db += f * P;
            if (dscale) {
                dshift[(size_t)b * C + c] = (float)P;
                dscale[(size_t)b * C + c] = (float)((double)gamma[c] * Q + (double)beta[c] * P);
            }
        }
        dgamma[c] = (float)dg;
        dbeta[c] = (float)db;
    }
}

template <typename T, bool ACT>
__global__ void __launch_bounds__(GN_THREADS)
gn_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ stats,
                    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ scale,
                    const float* __restrict__ shift, const float* __restrict__ gsum, T* __restrict__ dx, int64_t V,
                    int C, int G, const double* __restrict__ acc, float* __restrict__ dgamma, float* __restrict__ dbeta,
                    float* __restrict__ dscale, float* __restrict__ dshift) {
    if (blockIdx.x == 0 && blockIdx.y == 0) gn_bwd_params(acc, gamma, beta, scale, dgamma, dbeta, dscale, dshift, gridDim.y, C);
    const int b = blockIdx.y;
    const int L = C >> 3;
    const int rows = GN_THREADS / L;
    const int tid = threadIdx.x;
    const int lc = tid % L, r = tid / L;
    if (r >= rows) return;
    GnCoef k;
    float mean[8], rstd[8], gam[8], film[8];
    gn_load_coef(k, mean, rstd, gam, film, stats, gamma, beta, scale, shift, b, C, G, lc * 8);
    // dx = k1 dn - k2 - xhat k3,  xhat = x rstd + mr
    float k1[8], k2[8], k3[8], mr[8];
    const int cpg = C / G;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int g = (lc * 8 + j) / cpg;
        k1[j] = rstd[j] * gam[j] * film[j];
        k2[j] = rstd[j] * gsum[((size_t)b * G + g) * 2];
        k3[j] = rstd[j] * gsum[((size_t)b * G + g) * 2 + 1];
        mr[j] = -mean[j] * rstd[j];
    }
    const int64_t base = ((int64_t)b * V) * C + lc * 8;
    const int64_t stride = (int64_t)gridDim.x * rows;
    auto one = [&](const Vec8<T>& a, const Vec8<T>& g, int64_t vv) {
        Vec8<T> o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float n = __builtin_fmaf(a.v[j], k.a[j], k.c0[j]);
            const float dn = ACT ? g.v[j] * dsilu_f(n) : g.v[j];
            const float xh = __builtin_fmaf(a.v[j], rstd[j], mr[j]);
            o.v[j] = __builtin_fmaf(k1[j], dn, -__builtin_fmaf(xh, k3[j], k2[j]));
        }
        o.store(dx + base + vv * C);
    };
    int64_t v = (int64_t)blockIdx.x * rows + r;
    for (; v + (GN_UNROLL - 1) * stride < V; v += stride * GN_UNROLL) {
        Raw8<T> a[GN_UNROLL], g[GN_UNROLL];
#pragma unroll
        for (int u = 0; u < GN_UNROLL; ++u) {
            a[u].load(x + base + (v + u * stride) * C);
            g[u].load(dy + base + (v + u * stride) * C);
        }
        __builtin_amdgcn_sched_barrier(0);  // all loads of the trip are issued before any arithmetic
#pragma unroll
        for (int u = 0; u < GN_UNROLL; ++u) one(a[u].get(), g[u].get(), v + u * stride);
    }
    for (; v < V; v += stride) {
        Vec8<T> a, g;
        a.load(x + base + v * C);
        g.load(dy + base + v * C);
        one(a, g, v);
    }
}

extern "C" int tdx_gn_bwd(const void* x, const void* dy, const float* stats, const float* gamma, const float* beta,
                          const float* scale, const float* shift, void* dx, float* dgamma, float* dbeta, float* dscale,
                          float* dshift, int B, int64_t V, int C, int G, int act, int dtype, void* workspace,
                          void* stream) {
    TDX_CHECK_ARG(x && dy && stats && gamma && beta && dx && dgamma && dbeta && workspace && B > 0 && V > 0);
    TDX_CHECK_ARG((scale == nullptr) == (shift == nullptr));
    TDX_CHECK_ARG((scale == nullptr) == (dscale == nullptr) && (dscale == nullptr) == (dshift == nullptr));
    if (!gn_shape_ok(C, G)) return TDX_ESHAPE;
    const int nblk = gn_blocks_per_sample(B, V, C);
    double* acc = (double*)workspace;                      // [B][C][2]
    float* gsum = (float*)(acc + (size_t)B * C * 2);       // [B][G][2] (G <= C)
    float* partial = gsum + (size_t)B * C * 2;             // [B][nblk][C][2]
    dim3 grid(nblk, B);
    hipStream_t st = as_stream(stream);
    if (act)
        TDX_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((gn_bwd_reduce_kernel<T, true>), grid, dim3(GN_THREADS), 0, st,
                                                      (const T*)x, (const T*)dy, stats, gamma, beta, scale, shift,
                                                      partial, V, C, G));
    else
        TDX_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((gn_bwd_reduce_kernel<T, false>), grid, dim3(GN_THREADS), 0, st,
                                                      (const T*)x, (const T*)dy, stats, gamma, beta, scale, shift,
                                                      partial, V, C, G));
    hipLaunchKernelGGL(gn_bwd_group_kernel, dim3(B * G), dim3(256), 0, st, partial, acc, gamma, scale, gsum, nblk, C, G, V);
    if (act)
        TDX_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((gn_bwd_apply_kernel<T, true>), grid, dim3(GN_THREADS), 0, st,
                                                      (const T*)x, (const T*)dy, stats, gamma, beta, scale, shift, gsum,
                                                      (T*)dx, V, C, G, acc, dgamma, dbeta, dscale, dshift));
    else
        TDX_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((gn_bwd_apply_kernel<T, false>), grid, dim3(GN_THREADS), 0, st,
                                                      (const T*)x, (const T*)dy, stats, gamma, beta, scale, shift, gsum,
                                                      (T*)dx, V, C, G, acc, dgamma, dbeta, dscale, dshift));
    return tdx_launch_status();
}
