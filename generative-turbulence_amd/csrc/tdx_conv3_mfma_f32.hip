// fp32 MFMA implicit-GEMM 3x3x3 convolution (forward and, zero-padded on the padded grid, the data
// gradient) for the fp32 parity mode.  gfx950, v_mfma_f32_32x32x2_f32 (256 FLOP/clk/CU: the same peak as
// the packed-FMA vector ALU, but fed from an LDS halo brick with 27-fold reuse instead of a tile that is
// re-fetched for every tap).
//
// Same structure as the bf16 kernel (tdx_conv3_mfma.hip), in its plainest form: a workgroup owns a
// 4 x 8 x 8 brick of output voxels and BN = 32 * NT output channels; K = 27 taps x input channels is
// walked in 8-channel slices.  Per slice the halo'd brick (6 x 10 x 10 voxels x 8 ch) and the slice's
// weights of all 27 taps go to LDS as two half-planes of 16-B entries (4 fp32 channels each; the brick's z
// stride is padded 10 -> 12 so that fragment reads are conflict-free with affine addresses); the next
// slice's global loads are in flight during the MFMAs of the current one.  A lane reads one 16-B entry
// per operand and tap and feeds its four floats to four MFMAs (lanes 0-31 carry k = 0, lanes 32-63
// k = 1 of each 32x32x2 step: channel 4*(lane >> 5) + j for the j-th MFMA, on both operands).
// The MFMA is issued transposed (weights as A), so a lane owns one voxel and 4 consecutive channels per
// accumulator quad: 16-B writes into an LDS output tile, then full-row coalesced stores.
// Products and sums are IEEE fp32 (no reduced-precision operands): the 1e-4 parity gate holds as with
// the vector-ALU kernels; only the summation order differs.
#include "tdx_common.h"
#include "tdx_conv3.h"
#include "tdx_conv3_brick.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(16))) float f32x16;

#define F3_KC 8

bool conv3_mfma_f32_supported(int C1, int C2, int Cout) {
    return C1 > 0 && (C1 % F3_KC) == 0 && (C2 % F3_KC) == 0 && (Cout % 32) == 0;
}

// output tile rows of BN floats; 16-B chunk c of row v at c ^ swizzle(v)
template <int BN>
__device__ __forceinline__ int outf_addr(int v, int c) {
    return v * (BN * 4) + ((c ^ (v & (BN / 4 - 1))) << 4);
}

template <int NT, bool ZERO_PAD, int SHAPE, bool PERM>
__global__ void __launch_bounds__(256, 2)
conv3_mfma_f32_kernel(const float* __restrict__ x1, int C1, const float* __restrict__ x2, int C2,
                      const float* __restrict__ wp, const float* __restrict__ bias, float* __restrict__ y, BrickRegions R,
                      int Cout, double* __restrict__ gn_acc, float* __restrict__ d1, int D1, float* __restrict__ d2,
                      const float* __restrict__ a1, const float* __restrict__ a2) {
    using BR = Brick<SHAPE>;
    constexpr int MT = BR::MT;
    constexpr int BN = NT * 32;
    constexpr int HY = BR::HY, HZ = BR::HZ, SZ = BR::SZ;
    constexpr int NHALO = BR::NHALO;
    constexpr int APLANE = BR::ENTRIES * 16 + 64;
    constexpr int BRICK_BYTES = 2 * APLANE;
    constexpr int B_PLANE = 27 * BN * 16 + 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sA = smem;
    unsigned char* sB = smem + BRICK_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;

    int b, o[3];
    const BrickView g = brick_decode<SHAPE>(R, xcd_contiguous((int)blockIdx.x, (int)gridDim.x), b, o);
    const int n0 = blockIdx.y * BN;
    const int Cin = C1 + C2;

    // staging plan of the halo brick: 2 * NHALO 16-B pieces (voxel, half = channel group of 4)
    constexpr int A_PIECES = NHALO * 2;
    constexpr int A_PER_THREAD = (A_PIECES + 255) / 256;
    constexpr int B_PIECES = 27 * BN * 2;
    constexpr int B_PER_THREAD = (B_PIECES + 255) / 256;
    int a_src[A_PER_THREAD], a_dst[A_PER_THREAD];
#pragma unroll
    for (int i = 0; i < A_PER_THREAD; ++i) {
        const int p = tid + i * 256;
        a_dst[i] = -1;
        a_src[i] = -1;
        if (p < A_PIECES) {
            const int hv = ((p >> 3) << 2) + (p & 3), half = (p >> 2) & 1;
            const int hx = hv / (HY * HZ), rem = hv - hx * (HY * HZ);
            const int hy = rem / HZ, hz = rem - hy * HZ;
            a_dst[i] = half * APLANE + ((hx * HY + hy) * SZ + hz) * 16;
            const int src = brick_halo_source<ZERO_PAD, PERM>(g, o, hx, hy, hz);
            if (src >= 0) a_src[i] = src * 2 + half;
        }
    }
    const int64_t batch_vox = (int64_t)b * g.Ei[0] * g.Ei[1] * g.Ei[2];

    // weight staging role: (row = b_row0 + 128 i, half); packed layout [K/8][27][Cout][8]
    const int b_half = (tid >> 2) & 1;
    const int b_row0 = ((tid >> 3) << 2) + (tid & 3);
    const int b_goff = ((b_row0 / BN) * Cout + (b_row0 % BN)) * F3_KC + b_half * 4;
    const int b_dst = b_half * B_PLANE + b_row0 * 16;

    float4 areg[A_PER_THREAD], breg[B_PER_THREAD];
    auto load_slice = [&](int c) {
        const int k0 = c * F3_KC;
        const float* xs;
        int Cs, kk;
        if (k0 < C1) { xs = x1; Cs = C1; kk = k0; } else { xs = x2; Cs = C2; kk = k0 - C1; }
        xs += batch_vox * Cs + kk;
#pragma unroll
        for (int i = 0; i < A_PER_THREAD; ++i) {
            areg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a_src[i] >= 0)
                areg[i] = *reinterpret_cast<const float4*>(xs + (int64_t)(a_src[i] >> 1) * Cs + (a_src[i] & 1) * 4);
        }
        const float* wc = wp + (int64_t)c * 27 * Cout * F3_KC + (int64_t)n0 * F3_KC + b_goff;
#pragma unroll
        for (int i = 0; i < B_PER_THREAD; ++i) {
            breg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (b_row0 + 128 * i < 27 * BN)
                breg[i] = *reinterpret_cast<const float4*>(wc + (int64_t)i * (128 / BN) * Cout * F3_KC);
        }
    };
    auto store_slice = [&]() {
#pragma unroll
        for (int i = 0; i < A_PER_THREAD; ++i)
            if (a_dst[i] >= 0) *reinterpret_cast<float4*>(sA + a_dst[i]) = areg[i];
#pragma unroll
        for (int i = 0; i < B_PER_THREAD; ++i)
            if (b_row0 + 128 * i < 27 * BN) *reinterpret_cast<float4*>(sB + b_dst + i * 2048) = breg[i];
    };

    // this lane's voxel of M tile mt (Brick<THIN>::lane_voxel)
    int a_h[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int lx, ly, lz;
        BR::lane_voxel(wave, mt, r, lx, ly, lz);
        a_h[mt] = ((lx + 1) * HY + (ly + 1)) * SZ + (lz + 1);
    }
    int tap_row[27];  // weight-image row of every local tap (immediates when the axes are not permuted)
#pragma unroll
    for (int t = 0; t < 27; ++t) tap_row[t] = (PERM ? brick_tap(g, t / 9 - 1, (t / 3) % 3 - 1, t % 3 - 1) : t) * (BN * 16);
    int b_off[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) b_off[nt] = hh * B_PLANE + (nt * 32 + r) * 16;

    f32x16 acc[NT][MT];  // D[row = channel][col = voxel]
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[nt][mt][i] = 0.f;

    const int nchunks = Cin / F3_KC;
    load_slice(0);
    for (int c = 0; c < nchunks; ++c) {
        __syncthreads();
        store_slice();
        __syncthreads();
        if (c + 1 < nchunks) load_slice(c + 1);
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) {
            const int ex = tap / 9 - 1, ey = (tap / 3) % 3 - 1, ez = tap % 3 - 1;
            const int toff = (ex * HY + ey) * SZ + ez;
            float4 xf[MT], wf[NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) xf[mt] = *reinterpret_cast<const float4*>(sA + hh * APLANE + (a_h[mt] + toff) * 16);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wf[nt] = *reinterpret_cast<const float4*>(sB + tap_row[tap] + b_off[nt]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[nt].x, xf[mt].x, acc[nt][mt], 0, 0, 0);
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[nt].y, xf[mt].y, acc[nt][mt], 0, 0, 0);
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[nt].z, xf[mt].z, acc[nt][mt], 0, 0, 0);
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[nt].w, xf[mt].w, acc[nt][mt], 0, 0, 0);
                }
        }
    }

    // epilogue: lane (r, hh) holds voxel (wave, 4 mt + (r & 3), r >> 2) and channels nt*32 + 8 j + 4 hh + (0..3)
    __syncthreads();
    unsigned char* sO = smem;  // [256 voxels][BN] fp32
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ch = nt * 32 + 8 * j + 4 * hh;
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (bias) bv = *reinterpret_cast<const float4*>(bias + n0 + ch);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                int lx, ly, lz;
                BR::lane_voxel(wave, mt, r, lx, ly, lz);
                const int v = BR::tile_index(lx, ly, lz);
                *reinterpret_cast<float4*>(sO + outf_addr<BN>(v, ch >> 2)) =
                    make_float4(acc[nt][mt][4 * j] + bv.x, acc[nt][mt][4 * j + 1] + bv.y, acc[nt][mt][4 * j + 2] + bv.z,
                                acc[nt][mt][4 * j + 3] + bv.w);
            }
        }
    __syncthreads();
    constexpr int CHUNKS = BN / 4;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};  // GroupNorm moments of this thread's 4 channels  // 16-B chunks per voxel row
#pragma unroll
    for (int i = 0; i < CHUNKS * (BR::NVOX / 256); ++i) {
        const int p = tid + i * 256;
        const int v = p / CHUNKS, cidx = p % CHUNKS;
        int lx, ly, lz, c[3];
        BR::tile_voxel(v, lx, ly, lz);
        if (brick_out_coords<PERM>(g, o, lx, ly, lz, c)) {
            const int c0 = c[0], c1 = c[1], c2 = c[2];
            const int64_t ov = (((int64_t)b * g.Eo[0] + c0) * g.Eo[1] + c1) * g.Eo[2] + c2;
            float4 val = *reinterpret_cast<const float4*>(sO + outf_addr<BN>(v, cidx));
            bool direct = false;
            if (ZERO_PAD && d1 != nullptr) {
                // data gradient: padded position = original voxel + 1.  Positions inside the original grid go
                // straight to dx (split over the two inputs of a concatenated conv, plus the optional addend);
                // only the halo shell is written to the padded workspace for the face fix-up
                const int u0 = c0 + g.off, u1 = c1 + g.off, u2 = c2 + g.off;
                if (u0 >= 0 && u0 < g.Ei[0] && u1 >= 0 && u1 < g.Ei[1] && u2 >= 0 && u2 < g.Ei[2]) {
                    const int64_t u = (((int64_t)b * g.Ei[0] + u0) * g.Ei[1] + u1) * g.Ei[2] + u2;
                    const int n = n0 + cidx * 4;
                    const bool lo = n < D1;
                    float* dst = lo ? d1 + u * D1 + n : d2 + u * (Cout - D1) + (n - D1);
                    const float* asrc = lo ? (a1 ? a1 + u * D1 + n : nullptr) : (a2 ? a2 + u * (Cout - D1) + (n - D1) : nullptr);
                    if (asrc) {
                        const float4 av = *reinterpret_cast<const float4*>(asrc);
                        val.x += av.x; val.y += av.y; val.z += av.z; val.w += av.w;
                    }
                    *reinterpret_cast<float4*>(dst) = val;
                    direct = true;
                }
            }
            if (!direct) *reinterpret_cast<float4*>(y + ov * Cout + n0 + cidx * 4) = val;
            if (gn_acc != nullptr) {
                s1[0] += val.x; s2[0] += val.x * val.x; s1[1] += val.y; s2[1] += val.y * val.y;
                s1[2] += val.z; s2[2] += val.z * val.z; s1[3] += val.w; s2[3] += val.w * val.w;
            }
        }
    }
    if (gn_acc != nullptr) {
        // per-channel moments of the tile (tdx_conv3_fwd_gn): threads with equal tid % CHUNKS hold the same 4
        // channels; LDS reduce behind the output tile, then one f64 atomic per channel and moment into one of
        // TDX_GN_REPLICAS tables (as the bf16 kernel)
        constexpr int NP = 256 / CHUNKS;
        float* red = reinterpret_cast<float*>(smem + BR::NVOX * BN * 4);  // [NP][BN][2]
        const int cidx = tid % CHUNKS, part = tid / CHUNKS;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            red[(part * BN + cidx * 4 + e) * 2] = s1[e];
            red[(part * BN + cidx * 4 + e) * 2 + 1] = s2[e];
        }
        __syncthreads();
        if (tid < BN * 2) {
            float t = 0.f;
#pragma unroll 8
            for (int pp = 0; pp < NP; ++pp) t += red[pp * BN * 2 + tid];
            const int rep = blockIdx.x & (TDX_GN_REPLICAS - 1);
            atomicAdd(&gn_acc[(((size_t)rep * g.B + b) * Cout + n0) * 2 + tid], (double)t);
        }
    }
}

int conv3_mfma_f32_launch(const void* x1, int C1, const void* x2, int C2, const void* wp, const float* bias, void* y,
                          const Conv3Geom& g, int Cout, bool zero_pad, hipStream_t st, double* gn_acc, void* d1, int D1, void* d2,
                          const void* a1, const void* a2) {
    if ((int64_t)g.Xi * g.Yi * g.Zi * 2 >= (1ll << 31) || (int64_t)g.Xo * g.Yo * g.Zo >= (1ll << 31)) return TDX_ESHAPE;
    static const bool no_thin = getenv("TDX_CONV3_THIN") && atoi(getenv("TDX_CONV3_THIN")) == 0;  // A/B switch
    const int NT = (Cout % 64 == 0) ? 2 : 1;
    // (8 x 8 x 8 bricks with four M tiles per wave, which pay in the split kernel, measured 25-35 % SLOWER here: this
    // kernel is MFMA-bound with two workgroups per CU already)
    const bool big = false;
    BrickRegions main, thin;
    brick_plan(g, zero_pad, !no_thin, main, thin, big ? BRICK_BIG : BRICK_MAIN);
#define F3_GO(NTV, ZP, TH, PM, REG)                                                                                     \
    do {                                                                                                                \
        constexpr int BNV = NTV * 32;                                                                                   \
        size_t lds = (size_t)2 * (Brick<TH>::ENTRIES * 16 + 64) + (size_t)2 * (27 * BNV * 16 + 64);                     \
        const size_t tile = (size_t)Brick<TH>::NVOX * BNV * 4 + 8192; /* epilogue: output tile + moment reduction */    \
        if (tile > lds) lds = tile;                                                                                     \
        auto kern = conv3_mfma_f32_kernel<NTV, ZP, TH, PM>;                                                             \
        static bool attr_set = false;                                                                                   \
        if (!attr_set) {                                                                                                \
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return (int)e;                                                                         \
            attr_set = true;                                                                                            \
        }                                                                                                               \
        dim3 grid((unsigned)(REG).start[(REG).n], Cout / BNV);                                                          \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, (const float*)x1, C1, (const float*)x2, C2, (const float*)wp, \
                           bias, (float*)y, REG, Cout, gn_acc, (float*)d1, D1, (float*)d2, (const float*)a1,             \
                           (const float*)a2);                                                                           \
    } while (0)
    const bool perm = main.v[0].perm[0] != 0;
    if (NT == 2) {
        if (zero_pad && perm) F3_GO(2, true, BRICK_MAIN, true, main);
        else if (zero_pad) F3_GO(2, true, BRICK_MAIN, false, main);
        else if (perm) F3_GO(2, false, BRICK_MAIN, true, main);
        else F3_GO(2, false, BRICK_MAIN, false, main);
    } else if (big) {
        if (zero_pad && perm) F3_GO(1, true, BRICK_BIG, true, main);
        else if (zero_pad) F3_GO(1, true, BRICK_BIG, false, main);
        else if (perm) F3_GO(1, false, BRICK_BIG, true, main);
        else F3_GO(1, false, BRICK_BIG, false, main);
    } else {
        if (zero_pad && perm) F3_GO(1, true, BRICK_MAIN, true, main);
        else if (zero_pad) F3_GO(1, true, BRICK_MAIN, false, main);
        else if (perm) F3_GO(1, false, BRICK_MAIN, true, main);
        else F3_GO(1, false, BRICK_MAIN, false, main);
    }
    if (thin.n > 0) {  // remainder slabs of the padded grid: 2 x 16 x 8 bricks
        if (NT == 2) F3_GO(2, true, BRICK_THIN, true, thin); else F3_GO(1, true, BRICK_THIN, true, thin);
    }
#undef F3_GO
    return tdx_launch_status();
}
