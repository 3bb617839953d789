// Kernel template of the split-precision MFMA conv (design: tdx_conv3_mfma_split.hip) and the launcher of one
// instantiation.  Every (NT, ZERO_PAD, brick shape, PERM) instantiation lives in its own translation unit
// (tdx_conv3_mfma_split_i*.hip): its fully unrolled 27-tap x 6 NT-MFMA loop costs the instruction scheduler ~30 s, and the
// thirteen of them used to make this the longest single compile of the library (7 min).
#pragma once
#include "tdx_common.h"
#include "tdx_conv3.h"
#include "tdx_conv3_brick.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define SP_KC 16



template <int BN>
__device__ __forceinline__ int outs_addr(int v, int c) {
    return v * (BN * 4) + ((c ^ (v & (BN / 4 - 1))) << 4);
}

// 8 fp32 -> 8 bf16 hi and 8 bf16 lo
__device__ __forceinline__ void split8(const float4& a, const float4& b, uint4& hi, uint4& lo) {
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    unsigned h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h[i] = pack_bf16x2(v[2 * i], v[2 * i + 1]);
        const float r0 = v[2 * i] - __uint_as_float(h[i] << 16), r1 = v[2 * i + 1] - __uint_as_float(h[i] & 0xffff0000u);
        l[i] = pack_bf16x2(r0, r1);
    }
    hi = make_uint4(h[0], h[1], h[2], h[3]);
    lo = make_uint4(l[0], l[1], l[2], l[3]);
}

// LDS image of one 8-channel slice: activations [part = hi, lo][halo voxel, z stride 12][8 ch] (16-B entries), weights
// [part][tap pair j = 0..13][n][tap 2 j, 2 j + 1][8 ch]: an MFMA K step of 16 = 8 channels x 2 taps, lanes 0-31 hold tap 2 j
// and lanes 32-63 tap 2 j + 1 of both operands (tap 27 = zero weight rows).
template <int SHAPE, int BN> struct SplitLds {
    static constexpr int APLANE = Brick<SHAPE>::ENTRIES * 16 + 64;
    static constexpr int BPLANE = 28 * BN * 16 + 64;
    static constexpr int STAGE = 2 * APLANE + 2 * BPLANE;
    static constexpr int OUT = Brick<SHAPE>::NVOX * BN * 4 + (256 / (BN / 4)) * BN * 8;  // output tile + moment partials
    static constexpr int BYTES = STAGE > OUT ? STAGE : OUT;
};

// 4 fp32 -> 4 bf16 hi and 4 bf16 lo
__device__ __forceinline__ void split4(const float4& a, uint2& hi, uint2& lo) {
    const float v[4] = {a.x, a.y, a.z, a.w};
    unsigned h[2], l[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        h[i] = pack_bf16x2(v[2 * i], v[2 * i + 1]);
        const float r0 = v[2 * i] - __uint_as_float(h[i] << 16), r1 = v[2 * i + 1] - __uint_as_float(h[i] & 0xffff0000u);
        l[i] = pack_bf16x2(r0, r1);
    }
    hi = make_uint2(h[0], h[1]);
    lo = make_uint2(l[0], l[1]);
}

template <int NT, bool ZERO_PAD, int SHAPE, bool PERM>
__global__ void __launch_bounds__(256, 2)
conv3_mfma_split_kernel(const float* __restrict__ x1, int C1, const float* __restrict__ x2, int C2,
                        const bf16* __restrict__ wp, const float* __restrict__ bias, float* __restrict__ y, BrickRegions R,
                        int Cout, int64_t lo_offset, double* __restrict__ gn_acc, float* __restrict__ d1, int D1,
                        float* __restrict__ d2, const float* __restrict__ a1, const float* __restrict__ a2) {
    using BR = Brick<SHAPE>;
    constexpr int MT = BR::MT;
    constexpr int BN = NT * 32;
    constexpr int HY = BR::HY, HZ = BR::HZ, SZ = BR::SZ;
    constexpr int NHALO = BR::NHALO;
    constexpr int APLANE = SplitLds<SHAPE, BN>::APLANE;
    constexpr int B_PLANE = SplitLds<SHAPE, BN>::BPLANE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sA = smem;
    unsigned char* sB = smem + 2 * APLANE;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;

    int b, o[3];
    const BrickView g = brick_decode<SHAPE>(R, xcd_contiguous((int)blockIdx.x, (int)gridDim.x), b, o);
    const int n0 = blockIdx.y * BN;
    const int Cin = C1 + C2;

    // ---- staging plan of the halo brick: one piece = 4 of the 8 fp32 channels of a halo voxel; a lane pair covers the voxel's
    // 32 B, so one load instruction touches 32 cache lines, not 64 (the vector-memory path, not the LDS stores, is what this
    // staging costs: profiles/r12_split_staging_ablation.txt)
    constexpr int A_PIECES = NHALO * 2;
    constexpr int A_PER_THREAD = (A_PIECES + 255) / 256;
    int a_src[A_PER_THREAD], a_dst[A_PER_THREAD];  // source byte offset / (4 Cs) * 4 + 16 q, LDS byte
#pragma unroll
    for (int i = 0; i < A_PER_THREAD; ++i) {
        const int p = tid + i * 256;
        a_dst[i] = -1;
        a_src[i] = -1;
        if (p < A_PIECES) {
            const int hv = p >> 1, q = p & 1;
            const int hx = hv / (HY * HZ), rem = hv - hx * (HY * HZ);
            const int hy = rem / HZ, hz = rem - hy * HZ;
            a_dst[i] = ((hx * HY + hy) * SZ + hz) * 16 + q * 8;
            const int src = brick_halo_source<ZERO_PAD, PERM>(g, o, hx, hy, hz);
            if (src >= 0) a_src[i] = src * 2 + q;
        }
    }
    const int64_t batch_vox = (int64_t)b * g.Ei[0] * g.Ei[1] * g.Ei[2];

    // ---- weight staging: piece p = tid + 256 i -> (m = p / (2 BN): part m & 1, tap pair m >> 1; n = (p % (2 BN)) >> 1;
    // tap parity p & 1), 16 B = 8 channels; source = the packed [part][K/8][27][Cout][8] images: a slice's tap row is 16 B x
    // Cout contiguous (the [K/16][27][N][16] layout of round 4 made every piece half of a 32-B row: 2x the lines).
    // tid / (2 BN) is wave-uniform and i moves m by an even step: the part and the thread's byte offset are fixed, the tap pair
    // of piece i is uniform -- addresses are (uniform base) + (32-bit lane offset), nothing is kept per piece.
    constexpr int B_PIECES = 56 * BN;
    constexpr int B_PER_THREAD = B_PIECES / 256;
    constexpr int B_MSTEP = 256 / (2 * BN);
    static_assert(B_PIECES % 256 == 0 && B_MSTEP % 2 == 0, "weight pieces per thread");
    const int b_m0 = __builtin_amdgcn_readfirstlane(tid / (2 * BN));
    const int b_par = tid & 1;
    const unsigned b_lane = (unsigned)((((b_m0 & 1) * lo_offset) + (int64_t)(n0 + ((tid % (2 * BN)) >> 1)) * 8) * 2);
    const unsigned b_dst = (b_m0 & 1) * B_PLANE + ((b_m0 >> 1) * 2 * BN + (tid % (2 * BN))) * 16;

    float4 areg[A_PER_THREAD];
    uint4 breg[B_PER_THREAD];
    auto load_slice = [&](int s) {
        const int k0 = s * 8;
        const float* xs;
        int Cs, kk;
        if (k0 < C1) { xs = x1; Cs = C1; kk = k0; } else { xs = x2; Cs = C2; kk = k0 - C1; }
        const char* xb = reinterpret_cast<const char*>(xs + batch_vox * Cs + kk);
#pragma unroll
        for (int i = 0; i < A_PER_THREAD; ++i) {
            areg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a_src[i] >= 0)
                areg[i] = *reinterpret_cast<const float4*>(xb + (size_t)((unsigned)(a_src[i] >> 1) * (unsigned)Cs * 4u + (a_src[i] & 1) * 16u));
        }
        const char* wc = reinterpret_cast<const char*>(wp + (int64_t)s * 27 * Cout * 8);
#pragma unroll
        for (int i = 0; i < B_PER_THREAD; ++i) {
            const int tap0 = 2 * ((b_m0 >> 1) + i * (B_MSTEP / 2));  // uniform
            const int t1 = tap0 + 1 < 27 ? tap0 + 1 : 26;
            const int g0 = PERM ? brick_tap(g, tap0 / 9 - 1, (tap0 / 3) % 3 - 1, tap0 % 3 - 1) : tap0;
            const int g1 = PERM ? brick_tap(g, t1 / 9 - 1, (t1 / 3) % 3 - 1, t1 % 3 - 1) : t1;
            const unsigned off = b_lane + (unsigned)((b_par ? g1 : g0) * Cout * 16);
            breg[i] = *reinterpret_cast<const uint4*>(wc + (size_t)off);  // tap 27 reads tap 26's row; zeroed when stored
        }
    };
    auto store_slice = [&]() {
#pragma unroll
        for (int i = 0; i < A_PER_THREAD; ++i)
            if (a_dst[i] >= 0) {
                uint2 hi, lo;
                split4(areg[i], hi, lo);
                *reinterpret_cast<uint2*>(sA + a_dst[i]) = hi;
                *reinterpret_cast<uint2*>(sA + APLANE + a_dst[i]) = lo;
            }
#pragma unroll
        for (int i = 0; i < B_PER_THREAD; ++i)
        {
            // (zeroing the register right behind its load would make the compiler wait for every load in turn)
            // only a thread's last piece can be the 28th tap (pair 13, odd lane)
            uint4 v = breg[i];
            if (i == B_PER_THREAD - 1) {
                const unsigned keep = (2 * ((b_m0 >> 1) + i * (B_MSTEP / 2)) + 1 >= 27 && b_par) ? 0u : 0xffffffffu;
                v.x &= keep; v.y &= keep; v.z &= keep; v.w &= keep;
            }
            *reinterpret_cast<uint4*>(sB + b_dst + i * (B_MSTEP / 2) * (2 * BN * 16)) = v;
        }
    };

    // this lane's voxel of M tile mt (Brick<THIN>::lane_voxel)
    int a_h[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int lx, ly, lz;
        BR::lane_voxel(wave, mt, r, lx, ly, lz);
        a_h[mt] = (((lx + 1) * HY + (ly + 1)) * SZ + (lz + 1)) * 16;
    }
    const int b_off = r * 32 + hh * 16;

    f32x16 acc[NT][MT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[nt][mt][i] = 0.f;

    struct Frags { bf16x8 xh[MT], xl[MT], wh[NT], wl[NT]; };
    // K step j: taps 2 j (lanes 0-31) and 2 j + 1 (lanes 32-63; the 28th tap reads tap 26's voxel against zero weights)
    auto tap_off = [&](int tap) {
        const int t = tap < 27 ? tap : 26;
        return (((t / 9 - 1) * HY + ((t / 3) % 3 - 1)) * SZ + (t % 3 - 1)) * 16;
    };
    auto read_frags = [&](int j, Frags& f) {
        const int toff = hh ? tap_off(2 * j + 1) : tap_off(2 * j);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            f.xh[mt] = *reinterpret_cast<const bf16x8*>(sA + a_h[mt] + toff);
            f.xl[mt] = *reinterpret_cast<const bf16x8*>(sA + APLANE + a_h[mt] + toff);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f.wh[nt] = *reinterpret_cast<const bf16x8*>(sB + b_off + j * (2 * BN * 16) + nt * 1024);
            f.wl[nt] = *reinterpret_cast<const bf16x8*>(sB + B_PLANE + b_off + j * (2 * BN * 16) + nt * 1024);
        }
    };
    // term-major order: consecutive MFMAs go to different accumulators (tools/micro/mfma_peak: back-to-back MFMAs onto one
    // accumulator stall ~12 cycles each)
    auto mfmas = [&](const Frags& f) {
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(term == 2 ? f.wl[nt] : f.wh[nt],
                                                                         term == 1 ? f.xl[mt] : f.xh[mt], acc[nt][mt], 0, 0, 0);
    };

    const int nslices = Cin / 8;
    load_slice(0);
    for (int s = 0; s < nslices; ++s) {
        __syncthreads();
        store_slice();
        __syncthreads();
        if (s + 1 < nslices) load_slice(s + 1);
        Frags f0, f1;
        read_frags(0, f0);
#pragma unroll
        for (int j = 0; j < 14; j += 2) {
            read_frags(j + 1, f1);
            mfmas(f0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * MT + 2 * NT, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 3 * MT * NT, 0);
            if (j + 2 < 14) read_frags(j + 2, f0);
            mfmas(f1);
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * MT + 2 * NT, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 3 * MT * NT, 0);
        }
    }

    // ---- epilogue: as the fp32 kernel (lane (r, hh): voxel (wave, 4 mt + (r & 3), r >> 2), channels nt*32 + 8 j + 4 hh + 0..3)
    __syncthreads();
    unsigned char* sO = smem;  // [NVOX voxels][BN] fp32
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
                *reinterpret_cast<float4*>(sO + outs_addr<BN>(v, ch >> 2)) =
                    make_float4(acc[nt][mt][4 * j] + bv.x, acc[nt][mt][4 * j + 1] + bv.y, acc[nt][mt][4 * j + 2] + bv.z,
                                acc[nt][mt][4 * j + 3] + bv.w);
            }
        }
    __syncthreads();
    constexpr int CHUNKS = BN / 4;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};  // GroupNorm moments of this thread's 4 channels
#pragma unroll
    for (int i = 0; i < CHUNKS * (BR::NVOX / 256); ++i) {
        const int p = tid + i * 256;
        const int v = p / CHUNKS, cidx = p % CHUNKS;
        int lx, ly, lz, c[3];
        BR::tile_voxel(v, lx, ly, lz);
        if (brick_out_coords<PERM>(g, o, lx, ly, lz, c)) {
            const int c0 = c[0], c1 = c[1], c2 = c[2];
            const int64_t ov = (((int64_t)b * g.Eo[0] + c0) * g.Eo[1] + c1) * g.Eo[2] + c2;
            float4 val = *reinterpret_cast<const float4*>(sO + outs_addr<BN>(v, cidx));
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

#define SPLIT_GO_ARGS const void* x1, int C1, const void* x2, int C2, const void* wp, const float* bias, void* y,          \
                      const BrickRegions& reg, int Cout, int64_t lo_offset, double* gn_acc, void* d1, int D1, void* d2,       \
                      const void* a1, const void* a2, hipStream_t st
template <int NTV, bool ZP, int TH, bool PM>
static int split_go(SPLIT_GO_ARGS) {
    constexpr int BNV = NTV * 32;
    const size_t lds = SplitLds<TH, BNV>::BYTES;
    auto kern = conv3_mfma_split_kernel<NTV, ZP, TH, PM>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    dim3 grid((unsigned)reg.start[reg.n], Cout / BNV);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, (const float*)x1, C1, (const float*)x2, C2, (const bf16*)wp, bias, (float*)y,
                       reg, Cout, lo_offset, gn_acc, (float*)d1, D1, (float*)d2, (const float*)a1, (const float*)a2);
    return tdx_launch_status();
}
#define SPLIT_INSTANCE(NTV, ZP, TH, PM, NAME) \
    int NAME(SPLIT_GO_ARGS) { return split_go<NTV, ZP, TH, PM>(x1, C1, x2, C2, wp, bias, y, reg, Cout, lo_offset, gn_acc, d1, D1, d2, a1, a2, st); }
