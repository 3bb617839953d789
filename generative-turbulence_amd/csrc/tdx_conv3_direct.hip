// 3x3x3 convolution on NDHWC activations: weight packing, the vector-ALU implicit-GEMM
// kernels (any dtype, exact f32 accumulation -- the f32 parity path and the cross-check
// for the MFMA kernels), the halo fold of the data gradient, and the public entry points
// that dispatch between this file and tdx_conv3_mfma.hip.
//
// Geometry shared by both implementations: an "output grid" (Xo,Yo,Zo) is produced from an
// "input grid" (Xi,Yi,Zi); output voxel o reads input voxel o + off + e for the 27 taps
// e in {-1,0,1}^3, either clamped to the grid (replicate padding, forward: off = 0, grids
// equal) or treated as zero outside (data gradient: the adjoint is evaluated on the padded
// grid Xi+2 with off = -1 and the halo is folded back onto the boundary afterwards).
#include "tdx_common.h"
#include "tdx_conv3.h"
#include <stdlib.h>

// ------------------------------------------------------------------ weight packing -------
// element index of operand (tap, k, n) for a conv with K input / N output channels
__host__ __device__ __forceinline__ int64_t wp_index(int kc, int tap, int k, int n, int K, int N) {
    return kc ? ((((int64_t)(k / kc) * 27 + tap) * N + n) * kc + (k % kc)) : (((int64_t)tap * K + k) * N + n);
}

// w (Cout, Cin, 27) f32 -> wf: forward operand (K = Cin, N = Cout);
//                           wb: data-gradient operand (K = Cout, N = Cin, taps flipped)
template <typename T>
__global__ void conv3_pack_kernel(const float* __restrict__ w, T* __restrict__ wf, T* __restrict__ wb, int Cin,
                                  int Cout, int lf, int lb) {
    const int64_t n = (int64_t)Cout * Cin * 27;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int tap = (int)(i % 27);
        const int64_t r = i / 27;
        const int ci = (int)(r % Cin), co = (int)(r / Cin);
        const float v = w[i];
        if (wf) stf(wf + wp_index(lf, tap, ci, co, Cin, Cout), v);
        if (wb) stf(wb + wp_index(lb, 26 - tap, co, ci, Cout, Cin), v);
    }
}
// Tiled pack for the case that both operands use the MFMA layout: a workgroup owns a 16 (co) x 16 (ci) tile, reads
// its 16 runs of 16*27 contiguous floats, and writes, per tap, one 512-B run of wf ([ci/16][tap][co][16 ci]) and one
// of wb ([co/16][26-tap][ci][16 co]); per tap 16 rows x 2 halves of 8 packed values (one 16-B store each): 32 threads
// per tap, 8 taps per pass.  FMT = PACK_SPLIT: fp32 tensors' split-precision operands, a hi and a lo image (hi = bf16(v),
// lo = bf16(v - hi)) 27 Cin Cout elements apart; PACK_F16: the bf16 layout with IEEE half elements.
#define PACK_BF16 0
#define PACK_SPLIT 1
#define PACK_F16 2
template <int FMT>
__device__ __forceinline__ void conv3_pack_tile(const float* __restrict__ w, bf16* __restrict__ wf, bf16* __restrict__ wb,
                                                int Cin, int Cout, int ci0, int co0, float (*t)[16 * 27 + 1]) {
    constexpr bool SPLIT = FMT == PACK_SPLIT;
    typedef H16<FMT == PACK_F16> H;
    const int tid = threadIdx.x;
    const int64_t n = (int64_t)Cout * Cin * 27;
    for (int i = tid; i < 16 * 432; i += 256) {
        const int co = i / 432, r = i - co * 432;
        t[co][r] = w[((int64_t)(co0 + co) * Cin + ci0) * 27 + r];
    }
    __syncthreads();
    const int row = (tid >> 1) & 15, half = tid & 1, tsub = tid >> 5;
    auto put = [&](bf16* dst, int64_t j, const float* v) {
        unsigned h[4], l[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            h[q] = H::pack2(v[2 * q], v[2 * q + 1]);
            if (SPLIT)
                l[q] = pack_bf16x2(v[2 * q] - __uint_as_float(h[q] << 16), v[2 * q + 1] - __uint_as_float(h[q] & 0xffff0000u));
        }
        *reinterpret_cast<uint4*>(dst + j) = make_uint4(h[0], h[1], h[2], h[3]);
        if (SPLIT) *reinterpret_cast<uint4*>(dst + n + j) = make_uint4(l[0], l[1], l[2], l[3]);
    };
    for (int tap = tsub; tap < 27; tap += 8) {
        float v[8];
        if (wf) {  // row = co, k = ci
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = t[row][(half * 8 + q) * 27 + tap];
            // bf16: [ci/16][tap][co][16 ci]; split images: [ci/8][tap][co][8 ci] (8-channel slices are what their kernels stage)
            put(wf, SPLIT ? (((int64_t)((ci0 >> 3) + half) * 27 + tap) * Cout + co0 + row) << 3
                          : ((((int64_t)(ci0 >> 4) * 27 + tap) * Cout + co0 + row) << 4) + half * 8, v);
        }
        if (wb) {  // row = ci, k = co
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = t[half * 8 + q][row * 27 + tap];
            put(wb, SPLIT ? (((int64_t)((co0 >> 3) + half) * 27 + (26 - tap)) * Cin + ci0 + row) << 3
                          : ((((int64_t)(co0 >> 4) * 27 + (26 - tap)) * Cin + ci0 + row) << 4) + half * 8, v);
        }
    }
}

template <int FMT>
__global__ void __launch_bounds__(256)
conv3_pack_tiled_kernel(const float* __restrict__ w, bf16* __restrict__ wf, bf16* __restrict__ wb, int Cin, int Cout) {
    __shared__ float t[16][16 * 27 + 1];  // [co][ci*27 + tap]
    conv3_pack_tile<FMT>(w, wf, wb, Cin, Cout, blockIdx.x * 16, blockIdx.y * 16, t);
}

// the same over several weights in one launch (tdx_conv3_pack_weights): block -> (job, tile)
#define PACK_MAX_JOBS 32
struct PackTable {
    const float* w[PACK_MAX_JOBS];
    bf16* wf[PACK_MAX_JOBS];
    bf16* wb[PACK_MAX_JOBS];
    int Cin[PACK_MAX_JOBS], Cout[PACK_MAX_JOBS];
    int first[PACK_MAX_JOBS + 1];  // first block of job i
    int n;
};
template <int FMT>
__global__ void __launch_bounds__(256)
conv3_pack_tiled_many_kernel(PackTable tab) {
    __shared__ float t[16][16 * 27 + 1];
    int j = 0;
    while (j + 1 < tab.n && (int)blockIdx.x >= tab.first[j + 1]) ++j;
    const int tile = blockIdx.x - tab.first[j], ncx = tab.Cin[j] / 16;
    conv3_pack_tile<FMT>(tab.w[j], tab.wf[j], tab.wb[j], tab.Cin[j], tab.Cout[j], (tile % ncx) * 16, (tile / ncx) * 16, t);
}

// split-precision operands (tdx_conv3_mfma_split.hip): hi = bf16(v), lo = bf16(v - hi), two images [2][K/8][27][N][8] bf16
// in a buffer of the fp32 operand's size
__global__ void conv3_pack_split_kernel(const float* __restrict__ w, bf16* __restrict__ wf, bf16* __restrict__ wb, int Cin,
                                        int Cout) {
    const int64_t n = (int64_t)Cout * Cin * 27;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int tap = (int)(i % 27);
        const int64_t r = i / 27;
        const int ci = (int)(r % Cin), co = (int)(r / Cin);
        const float v = w[i];
        const bf16 hi = __float2bfloat16(v);
        const bf16 lo = __float2bfloat16(v - __bfloat162float(hi));
        if (wf) {
            const int64_t j = wp_index(8, tap, ci, co, Cin, Cout);
            wf[j] = hi;
            wf[n + j] = lo;
        }
        if (wb) {
            const int64_t j = wp_index(8, 26 - tap, co, ci, Cout, Cin);
            wb[j] = hi;
            wb[n + j] = lo;
        }
    }
}

extern "C" int tdx_conv3_pack_weight(const float* w, void* wf, void* wb, int Cin, int Cout, int dtype, void* stream) {
    TDX_CHECK_ARG(w && (wf || wb) && Cin > 0 && Cout > 0);
    const int64_t n = (int64_t)Cout * Cin * 27;
    if (dtype == TDX_F32_SPLIT) {
        // per operand: the split images where the split kernel can run (a function of (K, N) only, like every
        // layout decision here), else the plain fp32 operand
        void* sf = (wf && conv3_mfma_split_supported(Cin, 0, Cout)) ? wf : nullptr;
        void* sb = (wb && conv3_mfma_split_supported(Cout, 0, Cin)) ? wb : nullptr;
        if (sf && sb) {
            hipLaunchKernelGGL(conv3_pack_tiled_kernel<PACK_SPLIT>, dim3(Cin / 16, Cout / 16), dim3(256), 0, as_stream(stream), w,
                               (bf16*)sf, (bf16*)sb, Cin, Cout);
        } else if (sf || sb) {
            int grid = (int)min((int64_t)1024, (n + 255) / 256);
            hipLaunchKernelGGL(conv3_pack_split_kernel, dim3(grid), dim3(256), 0, as_stream(stream), w, (bf16*)sf, (bf16*)sb,
                               Cin, Cout);
        }
        void* pf = sf ? nullptr : wf;
        void* pb = sb ? nullptr : wb;
        if (pf || pb) return tdx_conv3_pack_weight(w, pf, pb, Cin, Cout, TDX_F32, stream);
        return tdx_launch_status();
    }
    const int lf = conv3_layout_kc(dtype, Cin, Cout), lb = conv3_layout_kc(dtype, Cout, Cin);
    if (tdx_is_h16(dtype) && (lf || !wf) && (lb || !wb) && (Cin % 16) == 0 && (Cout % 16) == 0) {
        if (dtype == TDX_F16)
            hipLaunchKernelGGL(conv3_pack_tiled_kernel<PACK_F16>, dim3(Cin / 16, Cout / 16), dim3(256), 0, as_stream(stream), w,
                               (bf16*)wf, (bf16*)wb, Cin, Cout);
        else
            hipLaunchKernelGGL(conv3_pack_tiled_kernel<PACK_BF16>, dim3(Cin / 16, Cout / 16), dim3(256), 0, as_stream(stream), w,
                               (bf16*)wf, (bf16*)wb, Cin, Cout);
        return tdx_launch_status();
    }
    int grid = (int)min((int64_t)1024, (n + 255) / 256);
    TDX_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((conv3_pack_kernel<T>), dim3(grid), dim3(256), 0, as_stream(stream),
                                                  w, (T*)wf, (T*)wb, Cin, Cout, lf, lb));
    return tdx_launch_status();
}

// Several weights in one call: the jobs whose both operands take the tiled kernel (the MFMA layouts, Cin % 16 ==
// Cout % 16 == 0 -- every 3x3x3 conv of the U-Net but the first) go out in launches of up to PACK_MAX_JOBS, the rest
// one by one through tdx_conv3_pack_weight.
extern "C" int tdx_conv3_pack_weights(const TdxPackJob* jobs, int n, int dtype, void* stream) {
    TDX_CHECK_ARG(jobs && n > 0);
    PackTable tab;
    tab.n = 0;
    tab.first[0] = 0;
    const bool split = dtype == TDX_F32_SPLIT;
    auto flush = [&]() -> int {
        if (tab.n == 0) return TDX_OK;
        if (split)
            hipLaunchKernelGGL(conv3_pack_tiled_many_kernel<PACK_SPLIT>, dim3(tab.first[tab.n]), dim3(256), 0, as_stream(stream), tab);
        else if (dtype == TDX_F16)
            hipLaunchKernelGGL(conv3_pack_tiled_many_kernel<PACK_F16>, dim3(tab.first[tab.n]), dim3(256), 0, as_stream(stream), tab);
        else
            hipLaunchKernelGGL(conv3_pack_tiled_many_kernel<PACK_BF16>, dim3(tab.first[tab.n]), dim3(256), 0, as_stream(stream), tab);
        tab.n = 0;
        return tdx_launch_status();
    };
    for (int i = 0; i < n; ++i) {
        const TdxPackJob& j = jobs[i];
        TDX_CHECK_ARG(j.w && j.wf && j.wb && j.Cin > 0 && j.Cout > 0);
        bool tiled = (j.Cin % 16) == 0 && (j.Cout % 16) == 0;
        if (split)
            tiled = tiled && conv3_mfma_split_supported(j.Cin, 0, j.Cout) && conv3_mfma_split_supported(j.Cout, 0, j.Cin);
        else
            tiled = tiled && tdx_is_h16(dtype) && conv3_layout_kc(dtype, j.Cin, j.Cout) && conv3_layout_kc(dtype, j.Cout, j.Cin);
        if (!tiled) {
            int rc = tdx_conv3_pack_weight(j.w, j.wf, j.wb, j.Cin, j.Cout, dtype, stream);
            if (rc != TDX_OK) return rc;
            continue;
        }
        const int k = tab.n++;
        tab.w[k] = j.w; tab.wf[k] = (bf16*)j.wf; tab.wb[k] = (bf16*)j.wb; tab.Cin[k] = j.Cin; tab.Cout[k] = j.Cout;
        tab.first[k + 1] = tab.first[k] + (j.Cin / 16) * (j.Cout / 16);
        if (tab.n == PACK_MAX_JOBS) {
            int rc = flush();
            if (rc != TDX_OK) return rc;
        }
    }
    return flush();
}

// Transposed copies of several fp32 matrices in one launch (the 1x1 conv weights (Cout, Cin) -> the [Cin][Cout]
// operand of tdx_conv1_fwd): dst[c][r] = src[r][c].
struct TransposeTable {
    const float* src[PACK_MAX_JOBS];
    float* dst[PACK_MAX_JOBS];
    int rows[PACK_MAX_JOBS], cols[PACK_MAX_JOBS];
    int first[PACK_MAX_JOBS + 1];
    int n;
};
__global__ void __launch_bounds__(256)
transpose_many_kernel(TransposeTable tab) {
    __shared__ float t[32][33];
    int j = 0;
    while (j + 1 < tab.n && (int)blockIdx.x >= tab.first[j + 1]) ++j;
    const int R = tab.rows[j], C = tab.cols[j];
    const int tile = blockIdx.x - tab.first[j], ntc = (C + 31) / 32;
    const int r0 = (tile / ntc) * 32, c0 = (tile % ntc) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8)
        if (r0 + i < R && c0 + tx < C) t[i][tx] = tab.src[j][(int64_t)(r0 + i) * C + c0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < C && r0 + tx < R) tab.dst[j][(int64_t)(c0 + i) * R + r0 + tx] = t[tx][i];
}
extern "C" int tdx_transpose_many(const TdxTransposeJob* jobs, int n, void* stream) {
    TDX_CHECK_ARG(jobs && n > 0);
    for (int lo = 0; lo < n; lo += PACK_MAX_JOBS) {
        TransposeTable tab;
        tab.n = min(PACK_MAX_JOBS, n - lo);
        tab.first[0] = 0;
        for (int k = 0; k < tab.n; ++k) {
            const TdxTransposeJob& j = jobs[lo + k];
            TDX_CHECK_ARG(j.src && j.dst && j.rows > 0 && j.cols > 0);
            tab.src[k] = j.src; tab.dst[k] = j.dst; tab.rows[k] = j.rows; tab.cols[k] = j.cols;
            tab.first[k + 1] = tab.first[k] + ceil_div(j.rows, 32) * ceil_div(j.cols, 32);
        }
        hipLaunchKernelGGL(transpose_many_kernel, dim3(tab.first[tab.n]), dim3(256), 0, as_stream(stream), tab);
        int rc = tdx_launch_status();
        if (rc != TDX_OK) return rc;
    }
    return TDX_OK;
}

// dwp [27][Cin][Cout] f32 -> dw (Cout, Cin, 27) f32, through a 16 (ci) x 16 (co) LDS tile so that
// both sides move 64-B+ runs; the accumulators are cleared behind the read (the workspace is left
// all-zero, see TDX_WS_CLEAN), and block (0, *) also moves the bias-gradient accumulator.
__global__ void __launch_bounds__(256)
conv3_unpack_wgrad_kernel(float* __restrict__ dwp, float* __restrict__ dw, float* __restrict__ dbw,
                          float* __restrict__ dbias, int Cin, int Cout, const float* __restrict__ slabs, int nslab) {
    __shared__ float t[16][16 * 27 + 1];  // [co][ci*27 + tap]
    const int ci0 = blockIdx.x * 16, co0 = blockIdx.y * 16;
    const int tid = threadIdx.x;
    const int ci = tid >> 4, co = tid & 15;
    const bool ok = ci0 + ci < Cin && co0 + co < Cout;
    // all 27 loads first (the clearing stores below may alias them as far as the compiler knows: interleaved, every
    // tap would wait for a full memory round trip)
    float v[27];
    const int64_t stride = (int64_t)27 * Cin * Cout, tstride = (int64_t)Cin * Cout;
    const int64_t idx0 = (int64_t)(ci0 + ci) * Cout + co0 + co;
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) v[tap] = 0.f;
    if (ok) {
        if (nslab > 0) {  // the K-splits stored their partial tiles into slabs: add them up
            for (int k = 0; k < nslab; ++k)
#pragma unroll
                for (int tap = 0; tap < 27; ++tap) v[tap] += slabs[k * stride + tap * tstride + idx0];
        } else {
#pragma unroll
            for (int tap = 0; tap < 27; ++tap) v[tap] = dwp[tap * tstride + idx0];
#pragma unroll
            for (int tap = 0; tap < 27; ++tap) dwp[tap * tstride + idx0] = 0.f;
        }
    }
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) t[co][ci * 27 + tap] = v[tap];
    __syncthreads();
    const int nci = min(16, Cin - ci0);
    for (int i = tid; i < 16 * nci * 27; i += 256) {
        const int c = i / (nci * 27), r = i - c * (nci * 27);
        if (co0 + c < Cout) dw[((int64_t)(co0 + c) * Cin + ci0) * 27 + r] = t[c][r];
    }
    if (blockIdx.x == 0 && tid < 16 && co0 + tid < Cout) {
        const float b = dbw[co0 + tid];
        dbw[co0 + tid] = 0.f;
        if (dbias) dbias[co0 + tid] = b;
    }
}

// The same unpack for launches that stored MANY per-split slabs (round 4: the fine levels' weight gradients split K over
// 32-512 workgroups; their fp32 atomic merge was 56 MB per launch at the chip's 1.3 TB/s of atomic adds = 30-43 us of
// every launch, and summed in arrival order).  Workgroup = (8 ci x 32 co) of ONE tap: a thread adds its element over the
// slabs in slab order (128-B runs per slab row, eight loads in flight) and writes dw in the parameter layout -- plain
// stores in the weight-gradient kernel, a fixed summation order here: the weight gradient is bit-reproducible.
__global__ void __launch_bounds__(256)
conv3_unpack_sum_kernel(float* __restrict__ dw, float* __restrict__ dbw, float* __restrict__ dbias, int Cin, int Cout,
                        const float* __restrict__ slabs, int nslab) {
    const int tap = blockIdx.z, ci = blockIdx.x * 8 + (threadIdx.x >> 5), co = blockIdx.y * 32 + (threadIdx.x & 31);
    const int64_t stride = (int64_t)27 * Cin * Cout;
    if (ci < Cin && co < Cout) {
        const float* p = slabs + ((int64_t)tap * Cin + ci) * Cout + co;
        float v = 0.f;
        int k = 0;
        for (; k + 8 <= nslab; k += 8) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = p[(int64_t)(k + u) * stride];
#pragma unroll
            for (int u = 0; u < 8; ++u) v += t[u];
        }
        for (; k < nslab; ++k) v += p[(int64_t)k * stride];
        dw[((int64_t)co * Cin + ci) * 27 + tap] = v;
    }
    if (blockIdx.x == 0 && blockIdx.z == 0 && threadIdx.x < 32 && co < Cout) {
        const float b = dbw[co];
        dbw[co] = 0.f;
        if (dbias) dbias[co] = b;
    }
}

// ------------------------------------------------------------------ direct implicit GEMM -
#define D3_BM 64
#define D3_BN 64
#define D3_BK 32

template <typename T, bool ZERO_PAD>
__global__ void __launch_bounds__(256)
conv3_direct_kernel(const T* __restrict__ x1, int C1, const T* __restrict__ x2, int C2, const T* __restrict__ wp,
                    const float* __restrict__ bias, T* __restrict__ y, Conv3Geom g, int Cout, int mfma_layout) {
    __shared__ float xs[D3_BK][D3_BM + 4];
    __shared__ float ws[D3_BK][D3_BN + 4];
    const int Cin = C1 + C2;
    const int64_t nvox_out = (int64_t)g.B * g.Xo * g.Yo * g.Zo;
    const int64_t m0 = (int64_t)blockIdx.x * D3_BM;
    const int n0 = blockIdx.y * D3_BN;
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;

    // staging role: voxel row rr, 8-channel group kk
    const int rr = tid >> 2, kk = (tid & 3) * 8;
    const int64_t mv = m0 + rr;
    const bool row_ok = mv < nvox_out;
    int ob = 0, ox = 0, oy = 0, oz = 0;
    if (row_ok) {
        int64_t v = mv;
        oz = (int)(v % g.Zo); v /= g.Zo;
        oy = (int)(v % g.Yo); v /= g.Yo;
        ox = (int)(v % g.Xo); ob = (int)(v / g.Xo);
    }
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

    for (int tap = 0; tap < 27; ++tap) {
        const int ex = tap / 9 - 1, ey = (tap / 3) % 3 - 1, ez = tap % 3 - 1;
        int sx = ox + g.off + ex, sy = oy + g.off + ey, sz = oz + g.off + ez;
        bool src_ok = row_ok;
        if (ZERO_PAD) {
            src_ok = src_ok && sx >= 0 && sx < g.Xi && sy >= 0 && sy < g.Yi && sz >= 0 && sz < g.Zi;
        } else {
            sx = min(max(sx, 0), g.Xi - 1); sy = min(max(sy, 0), g.Yi - 1); sz = min(max(sz, 0), g.Zi - 1);
        }
        const int64_t sv = (((int64_t)ob * g.Xi + sx) * g.Yi + sy) * g.Zi + sz;
        for (int k0 = 0; k0 < Cin; k0 += D3_BK) {
            {
                const int k = k0 + kk;
                Vec8<T> a;
#pragma unroll
                for (int j = 0; j < 8; ++j) a.v[j] = 0.f;
                if (src_ok && k < Cin) {
                    if (k < C1) a.load(x1 + sv * C1 + k);
                    else a.load(x2 + sv * C2 + (k - C1));
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) xs[kk + j][rr] = a.v[j];
            }
            {
                const int kq = tid >> 3, nn = (tid & 7) * 8;
                const int k = k0 + kq;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int n = n0 + nn + j;
                    ws[kq][nn + j] = (k < Cin && n < Cout) ? ldf(wp + wp_index(mfma_layout, tap, k, n, Cin, Cout)) : 0.f;
                }
            }
            __syncthreads();
#pragma unroll 8
            for (int k = 0; k < D3_BK; ++k) {
                const float4 a = *reinterpret_cast<const float4*>(&xs[k][ty * 4]);
                const float4 b = *reinterpret_cast<const float4*>(&ws[k][tx * 4]);
                const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] += av[i] * bv[j];
            }
            __syncthreads();
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t m = m0 + ty * 4 + i;
        if (m >= nvox_out) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tx * 4 + j;
            if (n >= Cout) continue;
            float v = acc[i][j];
            if (bias) v += bias[n];
            stf(y + m * Cout + n, v);
        }
    }
}

int conv3_direct_launch(const void* x1, int C1, const void* x2, int C2, const void* wp, const float* bias, void* y,
                        const Conv3Geom& g, int Cout, int dtype, bool zero_pad, hipStream_t st) {
    if ((C1 % 8) || (C2 % 8)) return TDX_ESHAPE;
    const int64_t nvox = (int64_t)g.B * g.Xo * g.Yo * g.Zo;
    dim3 grid(ceil_div(nvox, D3_BM), ceil_div(Cout, D3_BN));
    const int ml = conv3_layout_kc(dtype, C1 + C2, Cout);
    TDX_DISPATCH_DTYPE(dtype, {
        if (zero_pad)
            hipLaunchKernelGGL((conv3_direct_kernel<T, true>), grid, dim3(256), 0, st, (const T*)x1, C1, (const T*)x2,
                               C2, (const T*)wp, bias, (T*)y, g, Cout, ml);
        else
            hipLaunchKernelGGL((conv3_direct_kernel<T, false>), grid, dim3(256), 0, st, (const T*)x1, C1,
                               (const T*)x2, C2, (const T*)wp, bias, (T*)y, g, Cout, ml);
    });
    return tdx_launch_status();
}

// ------------------------------------------------------------------ halo fold ------------
// dx[b,u,:] (+)= sum over padded positions p with clamp(p) == u of dpad[b,p,:]
// dpad grid is (X+2, Y+2, Z+2) with p' = p + 1.  Channels are split over dx1 | dx2.
__device__ __forceinline__ void fold_range(int u, int n, int& lo, int& hi) {
    lo = (u == 0) ? 0 : u + 1;
    hi = (u == n - 1) ? n + 1 : u + 1;
}
template <typename T>
__global__ void __launch_bounds__(256)
conv3_fold_kernel(const T* __restrict__ dpad, T* __restrict__ dx1, int C1, T* __restrict__ dx2, int C2,
                  const T* __restrict__ add1, const T* __restrict__ add2,
                  int B, int X, int Y, int Z, int64_t total) {
    const int C = C1 + C2;
    const int L = C >> 3;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int lc = (int)(i % L);
    int64_t v = i / L;
    const int64_t vox = v;
    const int uz = (int)(v % Z); v /= Z;
    const int uy = (int)(v % Y); v /= Y;
    const int ux = (int)(v % X);
    const int b = (int)(v / X);
    int x0, x1, y0, y1, z0, z1;
    fold_range(ux, X, x0, x1);
    fold_range(uy, Y, y0, y1);
    fold_range(uz, Z, z0, z1);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    const int Xp = X + 2, Yp = Y + 2, Zp = Z + 2;
    for (int px = x0; px <= x1; ++px)
        for (int py = y0; py <= y1; ++py)
            for (int pz = z0; pz <= z1; ++pz) {
                Vec8<T> t;
                t.load(dpad + ((((int64_t)b * Xp + px) * Yp + py) * Zp + pz) * C + lc * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += t.v[j];
            }
    const int c = lc * 8;
    T* dst = (c < C1) ? dx1 + vox * C1 + c : dx2 + vox * C2 + (c - C1);
    Vec8<T> o;
    const T* asrc = (c < C1) ? (add1 ? add1 + vox * C1 + c : nullptr) : (add2 ? add2 + vox * C2 + (c - C1) : nullptr);
    if (asrc) {
        o.load(asrc);
#pragma unroll
        for (int j = 0; j < 8; ++j) o.v[j] += acc[j];
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) o.v[j] = acc[j];
    }
    o.store(dst);
}

// ------------------------------------------------------------------ direct weight grad ---
// dwp[tap][ci][co] += sum over a chunk of voxels of x[clamp(v+tap), ci] * dy[v, co]
#define D3W_VOX 8192
template <typename T>
__global__ void __launch_bounds__(256)
conv3_wgrad_direct_kernel(const T* __restrict__ x1, int C1, const T* __restrict__ x2, int C2,
                          const T* __restrict__ dy, float* __restrict__ dwp, float* __restrict__ dbias, int B, int X,
                          int Y, int Z, int Cout, int n_ci_tiles, int64_t vpb, int64_t slab_stride) {
    // vpb: voxels per block (D3W_VOX).  slab_stride != 0 (TDX_DETERMINISTIC): voxel chunk k STORES its partial sums into
    // slab k (every element of a slab has exactly one writer); the unpack kernel adds the slabs in order
    __shared__ float xs[D3_BK][D3_BM + 4];  // [voxel slice][ci]
    __shared__ float gs[D3_BK][D3_BN + 4];  // [voxel slice][co]
    const int Cin = C1 + C2;
    const int64_t nvox = (int64_t)B * X * Y * Z;
    const int64_t vbeg = (int64_t)blockIdx.x * vpb;
    const int64_t vend = min(nvox, vbeg + vpb);
    dwp += (int64_t)blockIdx.x * slab_stride;
    const int tap = blockIdx.y;
    const int ci0 = (blockIdx.z % n_ci_tiles) * D3_BM, co0 = (blockIdx.z / n_ci_tiles) * D3_BN;
    const int ex = tap / 9 - 1, ey = (tap / 3) % 3 - 1, ez = tap % 3 - 1;
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int rr = tid >> 3, cc = (tid & 7) * 8;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    float bsum = 0.f;
    const bool do_bias = dbias && tap == 13 && ci0 == 0;

    for (int64_t vs = vbeg; vs < vend; vs += D3_BK) {
        const int64_t v = vs + rr;
        Vec8<T> a, gq;
#pragma unroll
        for (int j = 0; j < 8; ++j) a.v[j] = gq.v[j] = 0.f;
        if (v < vend) {
            int64_t t = v;
            const int oz = (int)(t % Z); t /= Z;
            const int oy = (int)(t % Y); t /= Y;
            const int ox = (int)(t % X);
            const int ob = (int)(t / X);
            const int sx = min(max(ox + ex, 0), X - 1), sy = min(max(oy + ey, 0), Y - 1),
                      sz = min(max(oz + ez, 0), Z - 1);
            const int64_t sv = (((int64_t)ob * X + sx) * Y + sy) * Z + sz;
            const int ci = ci0 + cc, co = co0 + cc;
            if (ci < Cin) {
                if (ci < C1) a.load(x1 + sv * C1 + ci);
                else a.load(x2 + sv * C2 + (ci - C1));
            }
            if (co < Cout) gq.load(dy + v * Cout + co);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { xs[rr][cc + j] = a.v[j]; gs[rr][cc + j] = gq.v[j]; }
        __syncthreads();
#pragma unroll 8
        for (int k = 0; k < D3_BK; ++k) {
            const float4 av4 = *reinterpret_cast<const float4*>(&xs[k][ty * 4]);
            const float4 bv4 = *reinterpret_cast<const float4*>(&gs[k][tx * 4]);
            const float av[4] = {av4.x, av4.y, av4.z, av4.w}, bv[4] = {bv4.x, bv4.y, bv4.z, bv4.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] += av[i] * bv[j];
        }
        if (do_bias && tid < D3_BN) {
#pragma unroll 8
            for (int k = 0; k < D3_BK; ++k) bsum += gs[k][tid];
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ci = ci0 + ty * 4 + i;
        if (ci >= Cin) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int co = co0 + tx * 4 + j;
            if (co >= Cout) continue;
            if (slab_stride) dwp[((int64_t)tap * Cin + ci) * Cout + co] = acc[i][j];
            else atomicAdd(&dwp[((int64_t)tap * Cin + ci) * Cout + co], acc[i][j]);
        }
    }
    if (do_bias && tid < D3_BN && co0 + tid < Cout) atomicAdd(&dbias[co0 + tid], bsum);
}

// ------------------------------------------------------------------ scratch arena --------
// caller-provided transient workspace of kernels whose entry points have no argument for one (the K-split slabs of
// the small-grid conv, tdx_conv3_small.hip).  The first 64 bytes must be zero and stay zero.
int tdx_persistent_cus() {
    const char* env = getenv("TDX_PERSISTENT_CUS");
    int n = env ? atoi(env) : 256;
    n = n < 8 ? 8 : (n > 256 ? 256 : n);
    return n & ~7;
}
static void* g_scratch = nullptr;
static size_t g_scratch_bytes = 0;
void* tdx_scratch_ptr() { return g_scratch; }
size_t tdx_scratch_bytes() { return g_scratch_bytes; }
extern "C" int tdx_set_scratch(void* ptr, size_t bytes) {
    if (ptr != nullptr && bytes < 64) return TDX_EINVAL;  // at least the zero block
    g_scratch = ptr;
    g_scratch_bytes = ptr ? bytes : 0;
    return TDX_OK;
}

// ------------------------------------------------------------------ entry points ---------
// the matrix-core kernels of the 16-bit formats (bf16 and fp16 tensors share them: H16<HF>, tdx_common.h)
static bool mfma_ok(int dtype, int Cin1, int Cin2, int Cout) {
    return tdx_is_h16(dtype) && conv3_mfma_supported(Cin1, Cin2, Cout);
}

extern "C" int tdx_conv3_fwd(const void* x1, int C1, const void* x2, int C2, const void* wf, const float* bias,
                             void* y, int B, int X, int Y, int Z, int Cout, int dtype, int impl, void* stream) {
    TDX_CHECK_ARG(x1 && wf && y && B > 0 && X > 0 && Y > 0 && Z > 0 && C1 > 0 && C2 >= 0 && Cout > 0);
    TDX_CHECK_ARG(C2 == 0 || x2);
    Conv3Geom g = {B, X, Y, Z, X, Y, Z, 0};
    if (dtype == TDX_F32 && impl == TDX_CONV_SPLIT && conv3_mfma_split_supported(C1 + C2, 0, Cout)) {
        // the operand was packed as split images (layout is a function of (K, N)); both inputs must be sliceable
        if (!conv3_mfma_split_supported(C1, C2, Cout)) return TDX_ESHAPE;
        int rs = conv3_small_launch(x1, C1, x2, C2, wf, bias, y, Cout, nullptr, nullptr, nullptr, B, X, Y, Z, Cout, false, true,
                                    as_stream(stream));
        if (rs != TDX_ESHAPE) return rs;
        return conv3_mfma_split_launch(x1, C1, x2, C2, wf, bias, y, g, Cout, false, as_stream(stream));
    }
    if (dtype == TDX_F32 && impl != TDX_CONV_DIRECT && conv3_mfma_f32_supported(C1, C2, Cout))
        return conv3_mfma_f32_launch(x1, C1, x2, C2, wf, bias, y, g, Cout, false, as_stream(stream));
    const bool use_mfma = impl == TDX_CONV_MFMA || (impl == TDX_CONV_AUTO && mfma_ok(dtype, C1, C2, Cout));
    if (use_mfma) {
        if (!mfma_ok(dtype, C1, C2, Cout)) return TDX_ESHAPE;
        // deep U-Net levels: the small-grid kernel (packed M tiles, split K); TDX_ESHAPE = not such a case
        const bool hf = dtype == TDX_F16;
        int rs = conv3_small_launch(x1, C1, x2, C2, wf, bias, y, Cout, nullptr, nullptr, nullptr, B, X, Y, Z, Cout, false,
                                    false, as_stream(stream), hf);
        if (rs != TDX_ESHAPE) return rs;
        // the two finest levels: persistent LDS-DMA ring kernel (same products, fp32 sums in another order: equal to the brick
        // kernel up to ~1 bf16 ulp on a few % of the elements, tdx_conv3_ring.hip)
        rs = conv3_ring_launch(x1, C1, x2, C2, wf, bias, y, B, X, Y, Z, Cout, false, as_stream(stream), nullptr, nullptr, 0,
                               nullptr, nullptr, nullptr, hf);
        if (rs != TDX_ESHAPE) return rs;
        return conv3_mfma_launch(x1, C1, x2, C2, wf, bias, y, g, Cout, false, as_stream(stream), nullptr, nullptr, 0, nullptr,
                                 nullptr, nullptr, nullptr, nullptr, hf);
    }
    return conv3_direct_launch(x1, C1, x2, C2, wf, bias, y, g, Cout, dtype, false, as_stream(stream));
}

// Which kernel family tdx_conv3_fwd / tdx_conv3_fwd_gn run for this call (mirrors the dispatch above)
extern "C" int tdx_conv3_fwd_kernel(int C1, int C2, int Cout, int B, int X, int Y, int Z, int dtype, int impl) {
    impl &= 0xff;
    if (dtype == TDX_F32 && impl == TDX_CONV_SPLIT && conv3_mfma_split_supported(C1 + C2, 0, Cout))
        return conv3_small_applies(C1, C2, B, X, Y, Z, Cout, false, true) ? TDX_KERNEL_SMALL : TDX_KERNEL_BRICK;
    if (dtype == TDX_F32 && impl != TDX_CONV_DIRECT && conv3_mfma_f32_supported(C1, C2, Cout)) return TDX_KERNEL_BRICK;
    if (impl == TDX_CONV_MFMA || (impl == TDX_CONV_AUTO && mfma_ok(dtype, C1, C2, Cout))) {
        if (!mfma_ok(dtype, C1, C2, Cout)) return TDX_KERNEL_DIRECT;
        if (conv3_small_applies(C1, C2, B, X, Y, Z, Cout, false, false)) return TDX_KERNEL_SMALL;
        return conv3_ring_supported(C1, C2, Cout, B, X, Y, Z) ? TDX_KERNEL_RING : TDX_KERNEL_BRICK;
    }
    return TDX_KERNEL_DIRECT;
}

extern "C" int tdx_conv3_fwd_gn(const void* x1, int C1, const void* x2, int C2, const void* wf, const float* bias,
                                void* y, float* stats, int G, float eps, void* gn_workspace, int B, int X, int Y, int Z,
                                int Cout, int dtype, int impl, void* stream) {
    TDX_CHECK_ARG(x1 && wf && y && stats && gn_workspace && B > 0 && X > 0 && Y > 0 && Z > 0 && C1 > 0 && C2 >= 0);
    TDX_CHECK_ARG(Cout > 0 && G > 0 && (Cout % G) == 0 && (C2 == 0 || x2));
    const bool clean = (impl & TDX_WS_CLEAN) != 0;
    if (tdx_deterministic()) {
        // the conv kernels' epilogues merge their moments with f64 atomics in arrival order; deterministic runs take the conv and
        // then the statistics pass over its result, whose block partials are exact in f64 (gn_stats_launch)
        int rc = tdx_conv3_fwd(x1, C1, x2, C2, wf, bias, y, B, X, Y, Z, Cout, dtype, impl & 0xff, stream);
        if (rc != TDX_OK) return rc;
        return gn_stats_launch(y, stats, B, (int64_t)X * Y * Z, Cout, G, eps, dtype, gn_workspace, clean, as_stream(stream));
    }
    impl &= 0xff;
    const bool use_mfma =
        tdx_is_h16(dtype) && (impl == TDX_CONV_MFMA || (impl == TDX_CONV_AUTO && mfma_ok(dtype, C1, C2, Cout)));
    const bool hf = dtype == TDX_F16;
    const bool f32_split = dtype == TDX_F32 && impl == TDX_CONV_SPLIT && conv3_mfma_split_supported(C1 + C2, 0, Cout);
    const bool f32_mfma = dtype == TDX_F32 && !f32_split && impl != TDX_CONV_DIRECT && conv3_mfma_f32_supported(C1, C2, Cout);
    if (f32_split || f32_mfma) {  // fp32 tensors: the MFMA kernels accumulate the moments in their store loop too
        if (f32_split && !conv3_mfma_split_supported(C1, C2, Cout)) return TDX_ESHAPE;
        hipStream_t st = as_stream(stream);
        if (f32_split) {  // deep U-Net levels: small-grid conv, then the statistics pass over its (tiny) result
            int rs = conv3_small_launch(x1, C1, x2, C2, wf, bias, y, Cout, nullptr, nullptr, nullptr, B, X, Y, Z, Cout, false, true, st);
            if (rs == TDX_OK) return gn_stats_launch(y, stats, B, (int64_t)X * Y * Z, Cout, G, eps, dtype, gn_workspace, clean, st);
            if (rs != TDX_ESHAPE) return rs;
        }
        double* acc = (double*)gn_workspace;
        if (!clean) {
            int e = tdx_zero_async(acc, (size_t)TDX_GN_REPLICAS * B * Cout * 2 * sizeof(double), st);
            if (e != TDX_OK) return e;
        }
        Conv3Geom g = {B, X, Y, Z, X, Y, Z, 0};
        int rc = f32_split ? conv3_mfma_split_launch(x1, C1, x2, C2, wf, bias, y, g, Cout, false, st, acc)
                           : conv3_mfma_f32_launch(x1, C1, x2, C2, wf, bias, y, g, Cout, false, st, acc);
        if (rc != TDX_OK) return rc;
        return gn_finalize_launch(acc, stats, B, Cout, G, (int64_t)X * Y * Z, eps, TDX_GN_REPLICAS, st);
    }
    if (!use_mfma) {  // unfused: conv, then the streaming statistics pass
        int rc = tdx_conv3_fwd(x1, C1, x2, C2, wf, bias, y, B, X, Y, Z, Cout, dtype, impl, stream);
        if (rc != TDX_OK) return rc;
        return tdx_gn_stats(y, stats, B, (int64_t)X * Y * Z, Cout, G, eps, dtype, gn_workspace, stream);
    }
    if (!mfma_ok(dtype, C1, C2, Cout)) return tdx_is_h16(dtype) ? TDX_ESHAPE : TDX_EDTYPE;
    hipStream_t st = as_stream(stream);
    {   // deep U-Net levels: small-grid conv, then the statistics pass over its (tiny) result
        int rs = conv3_small_launch(x1, C1, x2, C2, wf, bias, y, Cout, nullptr, nullptr, nullptr, B, X, Y, Z, Cout, false, false, st, hf);
        if (rs == TDX_OK) return gn_stats_launch(y, stats, B, (int64_t)X * Y * Z, Cout, G, eps, dtype, gn_workspace, clean, st);
        if (rs != TDX_ESHAPE) return rs;
    }
    double* acc = (double*)gn_workspace;
    if (!clean) {
        int e = tdx_zero_async(acc, (size_t)TDX_GN_REPLICAS * B * Cout * 2 * sizeof(double), st);
        if (e != TDX_OK) return e;
    }
    Conv3Geom g = {B, X, Y, Z, X, Y, Z, 0};
    int rc = conv3_ring_launch(x1, C1, x2, C2, wf, bias, y, B, X, Y, Z, Cout, false, st, acc, nullptr, 0, nullptr, nullptr, nullptr, hf);
    if (rc == TDX_ESHAPE)
        rc = conv3_mfma_launch(x1, C1, x2, C2, wf, bias, y, g, Cout, false, st, acc, nullptr, 0, nullptr, nullptr, nullptr, nullptr,
                               nullptr, hf);
    if (rc != TDX_OK) return rc;
    return gn_finalize_launch(acc, stats, B, Cout, G, (int64_t)X * Y * Z, eps, TDX_GN_REPLICAS, st);
}

// Forward with a strided first input and accumulators that start from a precomputed partial
// convolution (bf16 MFMA path only): y = conv3(x1[..., :C1] with row stride ld1, wf) + bias + init.
extern "C" int tdx_conv3_fwd_partial(const void* x1, int C1, int ld1, const void* wf, const float* bias,
                                     const void* init, int init_shared, void* y, float* stats, int G, float eps,
                                     void* gn_workspace, int B, int X, int Y, int Z, int Cout, int dtype, int impl,
                                     void* stream) {
    TDX_CHECK_ARG(x1 && wf && y && B > 0 && X > 0 && Y > 0 && Z > 0 && C1 > 0 && Cout > 0 && ld1 >= C1 && (ld1 % 8) == 0);
    TDX_CHECK_ARG(stats == nullptr || (gn_workspace && G > 0 && (Cout % G) == 0));
    const bool clean = (impl & TDX_WS_CLEAN) != 0;
    if (!mfma_ok(dtype, C1, 0, Cout)) return tdx_is_h16(dtype) ? TDX_ESHAPE : TDX_EDTYPE;
    hipStream_t st = as_stream(stream);
    const bool det = tdx_deterministic();  // then: conv, and the ordered statistics pass over its result (as tdx_conv3_fwd_gn)
    double* acc = (stats && !det) ? (double*)gn_workspace : nullptr;
    if (acc && !clean) {
        int e = tdx_zero_async(acc, (size_t)TDX_GN_REPLICAS * B * Cout * 2 * sizeof(double), st);
        if (e != TDX_OK) return e;
    }
    Conv3Geom g = {B, X, Y, Z, X, Y, Z, 0};
    Conv3Ext ext = {ld1, 0, init, init_shared != 0};
    int rc = conv3_mfma_launch(x1, C1, nullptr, 0, wf, bias, y, g, Cout, false, st, acc, nullptr, 0, nullptr, nullptr, nullptr,
                               &ext, nullptr, dtype == TDX_F16);
    if (rc != TDX_OK || !stats) return rc;
    if (det) return gn_stats_launch(y, stats, B, (int64_t)X * Y * Z, Cout, G, eps, dtype, gn_workspace, clean, st);
    return gn_finalize_launch(acc, stats, B, Cout, G, (int64_t)X * Y * Z, eps, TDX_GN_REPLICAS, st);
}

extern "C" size_t tdx_conv3_bwd_data_workspace_bytes(int B, int X, int Y, int Z, int Cin, int dtype, int impl) {
    // the padded tensor of the vector-ALU path (shapes the MFMA kernels do not cover, TDX_CONV_DIRECT); which path a
    // call takes also depends on Cout, so the size is the same for all; the MFMA paths use it only for the position buffer
    // of the deterministic halo-shell route
    (void)impl;
    const size_t padded = (size_t)B * (X + 2) * (Y + 2) * (Z + 2) * Cin * (tdx_is_h16(dtype) ? 2 : 4);
    const size_t shell = conv3_shell_buffer_bytes(B, X, Y, Z, Cin);  // TDX_SHELL_DETERMINISTIC=1: one fp32 row per shell position
    return (padded > shell ? padded : shell) + 256;
}

// dx = adjoint of the replicate-padded conv.  MFMA paths: main term = zero-padded correlation on the original grid
// (the conv kernel, epilogue writes dx incl. the fused addend), then the halo-shell term added by
// conv3_shell_launch.  Vector-ALU path: correlation on the padded grid into the workspace, then the fold.
static int conv3_bwd_data_impl(const void* dy, const void* wb, void* dx1, int C1, void* dx2, int C2, const void* add1,
                               const void* add2, int B, int X, int Y, int Z, int Cout, int dtype, int impl,
                               void* workspace, void* stream) {
    TDX_CHECK_ARG(dy && wb && dx1 && workspace && B > 0 && X > 0 && Y > 0 && Z > 0 && C1 > 0 && C2 >= 0 && Cout > 0);
    TDX_CHECK_ARG(C2 == 0 || dx2);
    const int Cin = C1 + C2;
    if ((C1 % 8) || (C2 % 8) || (Cout % 8)) return TDX_ESHAPE;
    hipStream_t st = as_stream(stream);
    const Conv3Geom g0 = {B, X, Y, Z, X, Y, Z, 0};
    const bool use_mfma = tdx_is_h16(dtype) && (impl == TDX_CONV_MFMA || (impl == TDX_CONV_AUTO && mfma_ok(dtype, Cout, 0, Cin)));
    const bool hf = dtype == TDX_F16;
    int rc;
    if (use_mfma) {
        if (!mfma_ok(dtype, Cout, 0, Cin)) return TDX_ESHAPE;
        // deep U-Net levels: adjoint on the padded grid by the small-grid kernel, halo fold in its reduce pass
        rc = conv3_small_launch(dy, Cout, nullptr, 0, wb, nullptr, dx1, C1, dx2, add1, add2, B, X, Y, Z, Cin, true, false, st, hf);
        if (rc != TDX_ESHAPE) return rc;
        rc = conv3_ring_launch(dy, Cout, nullptr, 0, wb, nullptr, nullptr, B, X, Y, Z, Cin, true, st, nullptr, dx1, C1, dx2, add1, add2,
                               hf);
        if (rc == TDX_ESHAPE)
            rc = conv3_mfma_launch(dy, Cout, nullptr, 0, wb, nullptr, nullptr, g0, Cin, true, st, nullptr, dx1, C1, dx2, add1, add2,
                                   nullptr, nullptr, hf);
        if (rc != TDX_OK) return rc;
        return conv3_shell_launch(dy, wb, dx1, C1, dx2, B, X, Y, Z, Cout, Cin, hf ? 3 : 0, st, workspace);
    } else if (dtype == TDX_F32 && impl != TDX_CONV_DIRECT &&
               ((impl == TDX_CONV_SPLIT && conv3_mfma_split_supported(Cout, 0, Cin)) || conv3_mfma_f32_supported(Cout, 0, Cin))) {
        const bool split = impl == TDX_CONV_SPLIT && conv3_mfma_split_supported(Cout, 0, Cin);
        if (split) {
            rc = conv3_small_launch(dy, Cout, nullptr, 0, wb, nullptr, dx1, C1, dx2, add1, add2, B, X, Y, Z, Cin, true, true, st);
            if (rc != TDX_ESHAPE) return rc;
        }
        rc = split ? conv3_mfma_split_launch(dy, Cout, nullptr, 0, wb, nullptr, nullptr, g0, Cin, true, st, nullptr, dx1, C1, dx2,
                                             add1, add2)
                   : conv3_mfma_f32_launch(dy, Cout, nullptr, 0, wb, nullptr, nullptr, g0, Cin, true, st, nullptr, dx1, C1, dx2,
                                           add1, add2);
        if (rc != TDX_OK) return rc;
        return conv3_shell_launch(dy, wb, dx1, C1, dx2, B, X, Y, Z, Cout, Cin, split ? 2 : 1, st, workspace);
    }
    // adjoint on the padded grid: dpad[p'] = sum_e wb[e] dy_zero[p' - 1 + e]
    const Conv3Geom g = {B, X, Y, Z, X + 2, Y + 2, Z + 2, -1};
    rc = conv3_direct_launch(dy, Cout, nullptr, 0, wb, nullptr, workspace, g, Cin, dtype, true, st);
    if (rc != TDX_OK) return rc;
    const int64_t total = (int64_t)B * X * Y * Z * (Cin / 8);
    TDX_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((conv3_fold_kernel<T>), dim3(ceil_div(total, 256)), dim3(256), 0, st,
                                                  (const T*)workspace, (T*)dx1, C1, (T*)dx2, C2, (const T*)add1, (const T*)add2,
                                                  B, X, Y, Z, total));
    return tdx_launch_status();
}

extern "C" int tdx_conv3_bwd_data(const void* dy, const void* wb, void* dx1, int C1, void* dx2, int C2,
                                  int accumulate, int B, int X, int Y, int Z, int Cout, int dtype, int impl,
                                  void* workspace, void* stream) {
    return conv3_bwd_data_impl(dy, wb, dx1, C1, dx2, C2, accumulate ? dx1 : nullptr, accumulate ? dx2 : nullptr, B, X, Y, Z,
                               Cout, dtype, impl, workspace, stream);
}

extern "C" int tdx_conv3_bwd_data_add(const void* dy, const void* wb, void* dx1, int C1, void* dx2, int C2,
                                      const void* add1, const void* add2, int B, int X, int Y, int Z, int Cout, int dtype,
                                      int impl, void* workspace, void* stream) {
    return conv3_bwd_data_impl(dy, wb, dx1, C1, dx2, C2, add1, add2, B, X, Y, Z, Cout, dtype, impl, workspace, stream);
}

#define W3_MAX_SLABS 8
// per-split slabs a workspace holds: 8 for the wide layers (few K splits), up to 512 for the narrow ones of the fine
// levels, whose launches split K over 32-512 workgroups -- about 2 MB x 27 of slabs either way
static int w3_slab_capacity(int Cin, int Cout) {
    const int64_t c = ((int64_t)1 << 19) / ((int64_t)Cin * Cout);
    return (int)(c < W3_MAX_SLABS ? W3_MAX_SLABS : (c > 512 ? 512 : c));
}
extern "C" size_t tdx_conv3_bwd_weight_workspace_bytes(int Cin, int Cout, int impl) {
    (void)impl;
    // dw + dbias accumulators (the part covered by TDX_WS_CLEAN), then the partial-sum slabs (scratch, never needs zeroing)
    return (size_t)(1 + w3_slab_capacity(Cin, Cout)) * 27 * Cin * Cout * sizeof(float) + (size_t)Cout * sizeof(float) + 512;
}

extern "C" int tdx_conv3_bwd_weight(const void* x1, int C1, const void* x2, int C2, const void* dy, float* dw,
                                    float* dbias, int B, int X, int Y, int Z, int Cout, int dtype, int impl,
                                    void* workspace, void* stream) {
    TDX_CHECK_ARG(x1 && dy && dw && workspace && B > 0 && X > 0 && Y > 0 && Z > 0 && C1 > 0 && C2 >= 0 && Cout > 0);
    TDX_CHECK_ARG(C2 == 0 || x2);
    const int Cin = C1 + C2;
    if ((C1 % 8) || (C2 % 8) || (Cout % 8)) return TDX_ESHAPE;
    hipStream_t st = as_stream(stream);
    const bool clean = (impl & TDX_WS_CLEAN) != 0;
    impl &= 0xff;
    float* dwp = (float*)workspace;
    float* dbw = dwp + (size_t)27 * Cin * Cout;  // bias-gradient accumulator
    if (!clean) {
        int e = tdx_zero_async(dwp, ((size_t)27 * Cin * Cout + Cout) * sizeof(float), st);
        if (e != TDX_OK) return e;
    }
    const bool use_mfma = impl == TDX_CONV_MFMA || (impl == TDX_CONV_AUTO && tdx_is_h16(dtype) &&
                                                     conv3_wgrad_mfma_supported(C1, C2, Cout));
    int nslab = 0;
    const float* slab_ptr = nullptr;
    // TDX_DETERMINISTIC: no bias-gradient atomics inside the weight-gradient kernels -- the bias gradient is summed from dy in a
    // fixed order afterwards (partials in the slab region, free again once the unpack kernel has read it); the launchers hold
    // their K splits to the slab capacity (per-split slabs added in order: their default for few splits)
    const bool det = tdx_deterministic();
    float* bias_acc = (dbias && !det) ? dbw : nullptr;
    float* slab_base = dbw + ((Cout + 63) / 64) * 64;
    auto ordered_bias = [&]() -> int {
        if (!det || !dbias) return TDX_OK;
        return bias_grad_ordered_launch(dy, (int64_t)B * X * Y * Z, Cout, dtype, dbias, slab_base,
                                        (size_t)W3_MAX_SLABS * 27 * Cin * Cout, st);
    };
    // the fp32-tensor kernels split K 256-fold on the fine levels and merge with atomics by default (8 slabs); deterministic runs
    // give them the slab capacity the 16-bit kernels use
    const int cap_f32 = det ? w3_slab_capacity(Cin, Cout) : W3_MAX_SLABS;
    auto many_slabs = [&]() -> int {  // more slabs than the 16 x 16 unpack kernel walks: the per-tap summing unpack
        hipLaunchKernelGGL(conv3_unpack_sum_kernel, dim3(ceil_div(Cin, 8), ceil_div(Cout, 32), 27), dim3(256), 0, st, dw, dbw, dbias,
                           Cin, Cout, slab_ptr, nslab);
        const int rc2 = tdx_launch_status();
        return rc2 != TDX_OK ? rc2 : ordered_bias();
    };
    if (dtype == TDX_F32 && impl == TDX_CONV_SPLIT && conv3_wgrad_mfma_split_supported(C1, C2, Cout)) {
        float* slabs = dbw + ((Cout + 63) / 64) * 64;
        int rc = conv3_wgrad_mfma_split_launch(x1, C1, x2, C2, dy, dwp, bias_acc, B, X, Y, Z, Cout, st, slabs,
                                               cap_f32, &nslab);
        slab_ptr = slabs;
        if (rc != TDX_OK) return rc;
        if (nslab > W3_MAX_SLABS) return many_slabs();
    } else if (dtype == TDX_F32 && impl != TDX_CONV_DIRECT && conv3_wgrad_mfma_f32_supported(C1, C2, Cout)) {
        float* slabs = dbw + ((Cout + 63) / 64) * 64;
        int rc = conv3_wgrad_mfma_f32_launch(x1, C1, x2, C2, dy, dwp, bias_acc, B, X, Y, Z, Cout, st, slabs,
                                             cap_f32, &nslab);
        slab_ptr = slabs;
        if (rc != TDX_OK) return rc;
        if (nslab > W3_MAX_SLABS) return many_slabs();
    } else if (use_mfma) {
        if (!tdx_is_h16(dtype)) return TDX_EDTYPE;
        if (!conv3_wgrad_mfma_supported(C1, C2, Cout)) return TDX_ESHAPE;
        float* slabs = dbw + ((Cout + 63) / 64) * 64;
        const int cap = w3_slab_capacity(Cin, Cout);
        int rc = conv3_wgrad_mfma_launch(x1, C1, x2, C2, dy, dwp, bias_acc, B, X, Y, Z, Cout, st, slabs, cap, &nslab,
                                         dtype == TDX_F16);
        slab_ptr = slabs;
        if (rc != TDX_OK) return rc;
        if (nslab > W3_MAX_SLABS) return many_slabs();
    } else {
        const int64_t nvox = (int64_t)B * X * Y * Z;
        const int nci = ceil_div(Cin, D3_BM), nco = ceil_div(Cout, D3_BN);
        int64_t vpb = D3W_VOX, slab_stride = 0;
        float* out = dwp;
        if (det) {  // at most min(capacity, 64) voxel chunks, one slab each
            const int chunks = std::min(w3_slab_capacity(Cin, Cout), 64);
            vpb = (ceil_div(nvox, chunks) + D3_BK - 1) / D3_BK * D3_BK;
            slab_stride = (int64_t)27 * Cin * Cout;
            out = slab_base;
            nslab = ceil_div(nvox, vpb);
            slab_ptr = slab_base;
        }
        dim3 grid(ceil_div(nvox, vpb), 27, nci * nco);
        TDX_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((conv3_wgrad_direct_kernel<T>), grid, dim3(256), 0, st,
                                                      (const T*)x1, C1, (const T*)x2, C2, (const T*)dy, out,
                                                      bias_acc, B, X, Y, Z, Cout, nci, vpb, slab_stride));
    }
    hipLaunchKernelGGL(conv3_unpack_wgrad_kernel, dim3(ceil_div(Cin, 16), ceil_div(Cout, 16)), dim3(256), 0, st, dwp, dw,
                       dbw, dbias, Cin, Cout, slab_ptr, nslab);
    const int rc_unpack = tdx_launch_status();
    return rc_unpack != TDX_OK ? rc_unpack : ordered_bias();
}
