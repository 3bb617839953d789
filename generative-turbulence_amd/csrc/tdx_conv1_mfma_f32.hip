// fp32 MFMA 1x1x1 convolution and its weight gradient (gfx950, v_mfma_f32_32x32x2_f32), for the fp32 modes.
//   forward : y[r, :] = bias + [x1 | x2][r, :] @ w (+ add[r, :])          w [Cin][ldw] f32
//   wgrad   : dw[ci][co] += sum_r x[r, ci] dy[r, co],  dbias[co] += sum_r dy[r, co]
// IEEE fp32 products and sums, like the vector-ALU kernels in tdx_conv1.hip that they replace where the
// channel counts allow (those stage with scalar loads and run far below both of their rooflines); at the
// U-Net's channel counts these layers are HBM-bound, and the MFMA form gets them there: tiles are staged
// with 16-B loads, the next slice's loads are in flight during the MFMAs, results leave as 128-B rows.
//
// Forward: a workgroup owns 128 rows x BN = 32 NT columns, wave w the rows 32 w .. 32 w + 31.  x is the A
// operand (lane = row, k = lane >> 5) read from an LDS tile [row][32 k + 1 pad] (conflict-free b32 reads),
// w the B operand from [k][BN + 32 pad] (the two k rows of a step fall into different bank halves), so
// D[row][col = channel = lane & 31] and a store instruction writes two full 128-B row segments.
// Weight gradient: wave w of a workgroup owns the 32 (ci) x BN (co) tile of ci block w (up to 128 input
// channels per workgroup share one dy tile); K = rows, both operands are row-major = K-major, so fragments
// are again plain b32 reads of [row][channel] tiles; partial tiles are merged with fp32 atomics.
#include "tdx_common.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;

#define F1_ROWS 128
#define F1_BK 32
#define F1_XP (F1_BK + 1)

bool conv1_mfma_f32_supported(int C1, int C2, int Cout, const float* w, int ldw) {
    return C1 > 0 && (C1 % F1_BK) == 0 && (C2 % F1_BK) == 0 && (Cout % 32) == 0 && (ldw % 4) == 0 && ((uintptr_t)w % 16) == 0;
}

template <int NT, bool HAS_ADD>
__global__ void __launch_bounds__(256, 2)
conv1_f32_mfma_fwd_kernel(const float* __restrict__ x1, int C1, const float* __restrict__ x2, int C2,
                          const float* __restrict__ w, int ldw, const float* __restrict__ bias,
                          const float* __restrict__ add, float* __restrict__ y, int64_t rows, int Cout) {
    constexpr int BN = 32 * NT;
    constexpr int WP = (BN % 64 == 0) ? BN + 32 : BN + 64;  // pitch = 32 mod 64 floats: rows k, k + 1 in different bank halves
    __shared__ float xs[F1_ROWS * F1_XP];
    __shared__ float ws[F1_BK * WP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int64_t r0 = (int64_t)blockIdx.x * F1_ROWS;
    const int n0 = blockIdx.y * BN;
    const int Cin = C1 + C2;

    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nt][i] = 0.f;

    // staging roles: x tile 128 rows x 8 float4 (4 per thread), w tile 32 k x BN/4 float4 (NT * 256 / 256 ... per thread)
    constexpr int WPT = (F1_BK * BN / 4) / 256;  // 1 (NT = 1) or 2
    float4 xr[4], wr[WPT];
    auto load_slice = [&](int k0) {
        const float* xs_g;
        int Cs, kk;
        if (k0 < C1) { xs_g = x1; Cs = C1; kk = k0; } else { xs_g = x2; Cs = C2; kk = k0 - C1; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = tid + i * 256, row = p >> 3, c4 = p & 7;
            xr[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r0 + row < rows) xr[i] = *reinterpret_cast<const float4*>(xs_g + (r0 + row) * Cs + kk + c4 * 4);
        }
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int p = tid + i * 256, k = p / (BN / 4), c4 = p % (BN / 4);
            wr[i] = *reinterpret_cast<const float4*>(w + (int64_t)(k0 + k) * ldw + n0 + c4 * 4);
        }
    };
    auto store_slice = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = tid + i * 256, row = p >> 3, c4 = p & 7;
            float* d = xs + row * F1_XP + c4 * 4;
            d[0] = xr[i].x; d[1] = xr[i].y; d[2] = xr[i].z; d[3] = xr[i].w;
        }
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int p = tid + i * 256, k = p / (BN / 4), c4 = p % (BN / 4);
            *reinterpret_cast<float4*>(ws + k * WP + c4 * 4) = wr[i];
        }
    };

    const float* xa = xs + (wave * 32 + r) * F1_XP + hh;
    const float* wb = ws + hh * WP + r;
    load_slice(0);
    for (int k0 = 0; k0 < Cin; k0 += F1_BK) {
        __syncthreads();
        store_slice();
        __syncthreads();
        if (k0 + F1_BK < Cin) load_slice(k0 + F1_BK);
#pragma unroll
        for (int i = 0; i < F1_BK / 2; ++i) {
            const float a = xa[2 * i];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wb[2 * i * WP + nt * 32], acc[nt], 0, 0, 0);
        }
    }
    // D[row = 8 j + 4 hh + t][col = r]: lane holds channel n0 + nt*32 + r of 16 rows
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int ch = n0 + nt * 32 + r;
        const float bv = bias ? bias[ch] : 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int64_t row = r0 + wave * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
            if (row < rows) {
                float v = acc[nt][i] + bv;
                if (HAS_ADD) v += add[row * Cout + ch];
                y[row * Cout + ch] = v;
            }
        }
    }
}

int conv1_mfma_f32_fwd_launch(const void* x1, int C1, const void* x2, int C2, const float* w, int ldw, const float* bias,
                              const void* add, void* y, int64_t rows, int Cout, hipStream_t st) {
    const int NT = (Cout % 64 == 0) ? 2 : 1;
    dim3 grid((unsigned)ceil_div(rows, (int64_t)F1_ROWS), Cout / (32 * NT));
#define F1_GO(NTV, ADD)                                                                                                    \
    hipLaunchKernelGGL((conv1_f32_mfma_fwd_kernel<NTV, ADD>), grid, dim3(256), 0, st, (const float*)x1, C1, (const float*)x2, \
                       C2, w, ldw, bias, (const float*)add, (float*)y, rows, Cout)
    if (NT == 2) { if (add) F1_GO(2, true); else F1_GO(2, false); }
    else { if (add) F1_GO(1, true); else F1_GO(1, false); }
#undef F1_GO
    return tdx_launch_status();
}

// ---------------------------------------------------------------------------------------------------
#define F1W_RS 32                 // rows per staged slice
#define F1W_CI 128                // input channels per workgroup (4 waves x 32)
#define F1W_XP (F1W_CI + 32)      // x tile pitch

bool conv1_wgrad_mfma_f32_supported(int Cin, int Cout) { return (Cin % 32) == 0 && (Cout % 32) == 0; }

template <int NT, bool TR>  // TR: dw[co][ci] (row stride ldw), MFMA operands swapped so that a lane owns one ci
__global__ void __launch_bounds__(256, 2)
conv1_f32_mfma_wgrad_kernel(const float* __restrict__ x, int Cin, const float* __restrict__ dy, int Cout,
                            float* __restrict__ dw, int ldw, float* __restrict__ dbias, int64_t rows, int64_t rows_per_split,
                            int64_t split_stride) {
    constexpr int BN = 32 * NT;
    constexpr int GP = (BN % 64 == 0) ? BN + 32 : BN + 64;  // pitch = 32 mod 64 floats
    __shared__ float xs[F1W_RS * F1W_XP];
    __shared__ float gs[F1W_RS * GP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int ci_blk = blockIdx.y * F1W_CI;
    const int nci = min(F1W_CI, Cin - ci_blk);       // multiple of 32
    const int co0 = blockIdx.z * BN;
    const int64_t rbeg = (int64_t)blockIdx.x * rows_per_split, rend = min(rows, rbeg + rows_per_split);
    const bool active = wave * 32 < nci;
    dw += (int64_t)blockIdx.x * split_stride;  // TDX_DETERMINISTIC: split k merges into its own zeroed slab
    if (dbias) dbias += (int64_t)blockIdx.x * split_stride;

    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nt][i] = 0.f;
    const bool do_bias = dbias != nullptr && blockIdx.y == 0;
    float bs[4] = {0.f, 0.f, 0.f, 0.f};  // piece tid % (BN / 4) of dy: always the same 4 columns

    constexpr int XPT = (F1W_RS * F1W_CI / 4) / 256;  // 4
    constexpr int GPT = (F1W_RS * BN / 4) / 256;      // 1 or 2
    float4 xr[XPT], gr[GPT];
    auto load_slice = [&](int64_t rs) {
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int p = tid + i * 256, row = p >> 5, c4 = p & 31;
            xr[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rs + row < rend && c4 * 4 < nci) xr[i] = *reinterpret_cast<const float4*>(x + (rs + row) * Cin + ci_blk + c4 * 4);
        }
#pragma unroll
        for (int i = 0; i < GPT; ++i) {
            const int p = tid + i * 256, row = p / (BN / 4), c4 = p % (BN / 4);
            gr[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rs + row < rend) gr[i] = *reinterpret_cast<const float4*>(dy + (rs + row) * Cout + co0 + c4 * 4);
        }
    };
    auto store_slice = [&]() {
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int p = tid + i * 256, row = p >> 5, c4 = p & 31;
            *reinterpret_cast<float4*>(xs + row * F1W_XP + c4 * 4) = xr[i];
        }
#pragma unroll
        for (int i = 0; i < GPT; ++i) {
            const int p = tid + i * 256, row = p / (BN / 4), c4 = p % (BN / 4);
            *reinterpret_cast<float4*>(gs + row * GP + c4 * 4) = gr[i];
            if (do_bias) { bs[0] += gr[i].x; bs[1] += gr[i].y; bs[2] += gr[i].z; bs[3] += gr[i].w; }
        }
    };

    const float* xa = xs + hh * F1W_XP + wave * 32 + r;
    const float* gb = gs + hh * GP + r;
    if (rbeg < rend) load_slice(rbeg);
    for (int64_t rs = rbeg; rs < rend; rs += F1W_RS) {
        __syncthreads();
        store_slice();
        __syncthreads();
        if (rs + F1W_RS < rend) load_slice(rs + F1W_RS);
        if (active) {
#pragma unroll
            for (int i = 0; i < F1W_RS / 2; ++i) {
                const float a = xa[2 * i * F1W_XP];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt] = TR ? __builtin_amdgcn_mfma_f32_32x32x2f32(gb[2 * i * GP + nt * 32], a, acc[nt], 0, 0, 0)
                                 : __builtin_amdgcn_mfma_f32_32x32x2f32(a, gb[2 * i * GP + nt * 32], acc[nt], 0, 0, 0);
            }
        }
    }
    // D[row = ci][col = co]
    if (active) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int q = (i & 3) + 8 * (i >> 2) + 4 * hh;
                if (TR)
                    atomicAdd(&dw[(int64_t)(co0 + nt * 32 + q) * ldw + ci_blk + wave * 32 + r], acc[nt][i]);
                else
                    atomicAdd(&dw[(int64_t)(ci_blk + wave * 32 + q) * ldw + co0 + nt * 32 + r], acc[nt][i]);
            }
    }
    if (do_bias) {
        __syncthreads();
        float* red = xs;  // [256][4]
#pragma unroll
        for (int e = 0; e < 4; ++e) red[tid * 4 + e] = bs[e];
        __syncthreads();
        if (tid < BN) {
            const int c4 = tid >> 2, e = tid & 3;
            float t = 0.f;
            for (int k = c4; k < 256; k += BN / 4) t += red[k * 4 + e];
            atomicAdd(&dbias[co0 + tid], t);
        }
    }
}

int conv1_wgrad_mfma_f32_launch(const void* x, int Cin, const void* dy, int Cout, float* dw, int ldw, float* dbias,
                                int64_t rows, bool transposed, hipStream_t st, int max_split_arg, int64_t split_stride,
                                int* nsplit_out, bool plan_only) {
    const int NT = (Cout % 64 == 0) ? 2 : 1;
    const int nblk = ceil_div(Cin, F1W_CI), nco = Cout / (32 * NT);
    // ~1024 workgroups overall (two per CU resident), each at least 8 slices long
    int64_t nsplit = ceil_div((int64_t)1024, (int64_t)nblk * nco);
    const int64_t max_split = ceil_div(rows, (int64_t)(8 * F1W_RS));
    if (nsplit > max_split) nsplit = max_split;
    if (max_split_arg > 0 && nsplit > max_split_arg) nsplit = max_split_arg;
    if (nsplit < 1) nsplit = 1;
    int64_t rps = ceil_div(rows, nsplit);
    rps = ceil_div(rps, (int64_t)F1W_RS) * F1W_RS;
    nsplit = ceil_div(rows, rps);
    if (nsplit_out) *nsplit_out = (int)nsplit;
    if (plan_only) return TDX_OK;
    dim3 grid((unsigned)nsplit, nblk, nco);
#define F1W_LAUNCH(N, T)                                                                                                  \
    hipLaunchKernelGGL((conv1_f32_mfma_wgrad_kernel<N, T>), grid, dim3(256), 0, st, (const float*)x, Cin, (const float*)dy, \
                       Cout, dw, ldw, dbias, rows, rps, split_stride)
    if (NT == 2 && transposed) F1W_LAUNCH(2, true);
    else if (NT == 2) F1W_LAUNCH(2, false);
    else if (transposed) F1W_LAUNCH(1, true);
    else F1W_LAUNCH(1, false);
#undef F1W_LAUNCH
    return tdx_launch_status();
}
