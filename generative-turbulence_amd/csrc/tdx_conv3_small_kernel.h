// Kernel template of the small-grid conv (see tdx_conv3_small.hip for the design) and the per-instantiation launcher.
#pragma once
#include "tdx_common.h"
#include "tdx_conv3.h"
#include "tdx_conv3.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define SM_KC 16
#define SM_BN 32
#define SM_MAX_TILES 28

struct SmallGeom {
    int B;
    int Ev[3];     // virtual grid (rows)
    int Es[3];     // source grid
    int off;       // source coordinate = virtual coordinate - off
    int clamp;     // 1: clamp sources into the grid (forward), 0: zero outside (data gradient)
    int nbg;       // samples per row group (1 when a sample is cut into x slabs)
    int xs;        // virtual x planes per row group
    int gx;        // x slabs per sample
    int Ix, Iy, Iz;  // LDS image per sample of a group: (xs + 2) x (Ev[1] + 2) x (Ev[2] + 2) entries
    int K, N;      // channels of the source tensor(s) / of the result
    int per_split; // K slices per split
    int nsplit;
};

// SPLIT = false: bf16 tensors.  SPLIT = true: fp32 tensors with split-precision products (every operand as bf16 hi + lo,
// x w ~ xh wh + xl wh + xh wl, as tdx_conv3_mfma_split.hip): the image is staged as raw fp32 (four 16-B quarter planes of
// 4 channels) and split into hi / lo when a fragment is read; the weights arrive pre-split ([2 parts][K/8][27][N][8] bf16,
// lo_offset elements apart).  fp32 images and two weight parts do not fit twice: that mode is single-buffered.
// Round 4: FOUR LOADER WAVES beside the four computing waves (512 threads; one computing + one loader wave per SIMD).  An
// LDS-DMA instruction blocks its issuing wave for ~150 cycles (profiles/r10_ring_stamps.txt); a slice is 16-23 of them per
// wave against 27 x MTW MFMAs of 32 cycles, and with ONE wave per SIMD nothing else could issue meanwhile: the matrix pipe
// was busy 24-33 % of the kernel (profiles/r11bf16_summary.md).  The loader waves issue every copy, wait for it and meet
// the computing waves at the slice barrier, as in tdx_conv3_ring.hip; the computing waves carry no vector-memory
// instruction besides their final stores.
#define SM_LOADERS 4
// HF (SPLIT = false only): the 16-bit tensors and the packed weight are IEEE half instead of bfloat16 (H16<HF>).
template <int MTW, bool SPLIT, bool HF>
__global__ void __launch_bounds__(256 + 64 * SM_LOADERS, 1)
conv3_small_kernel(const void* __restrict__ x1_, int C1, const void* __restrict__ x2_, int C2, const bf16* __restrict__ wp,
                   float* __restrict__ slab, const void* __restrict__ zero16, SmallGeom g, int64_t lo_offset) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const bool loader = tid >= 256;               // waves 4-7: issue the LDS-DMA copies; waves 0-3: compute
    const int wave = (tid >> 6) & 3;              // index among the waves of its kind
    const int r = lane & 31, hh = lane >> 5;
    constexpr int NPL = SPLIT ? 4 : 2;            // 16-B planes per image entry (8 bf16 / 4 fp32 channels each)
    constexpr int NBUF = SPLIT ? 1 : 2;           // LDS buffers of image and weights
    constexpr int WPARTS = SPLIT ? 2 : 1;         // weight images: hi, lo
    constexpr int ESZ = SPLIT ? 4 : 2;            // bytes per source element

    // ---- this workgroup: row group (samples b0 .. b0 + nb, virtual planes x0 .. x0 + xs), channel tile, K split
    // blockIdx.x = channel tile, y = row group, z = K split.  (An XCD-aware layout -- the low 3 bits of the block id
    // enumerating (K split, channel-tile class), so that one XCD's L2 sees 3.5 instead of 21 MB at 24 x 8 x 6 -- was
    // measured: no difference; the kernel is bound by its LDS fragment reads, 1.2 per MFMA at 32-wide channel tiles.)
    const int group = blockIdx.y, n0 = blockIdx.x * SM_BN, split = blockIdx.z;
    const int b0 = (group / g.gx) * g.nbg, x0 = (group % g.gx) * g.xs;
    const int nb = min(g.nbg, g.B - b0), xs = min(g.xs, g.Ev[0] - x0);
    const int per_sample = xs * g.Ev[1] * g.Ev[2];
    const int nrows = nb * per_sample;
    const int img = g.Ix * g.Iy * g.Iz;           // entries per sample image
    const int entries = nb * img;
    const int IMG_HALF = ((g.nbg * img + 63) & ~63) * 16;  // bytes of one plane: a full group, in whole 1-KiB DMA pieces
    constexpr int W_HALF = 27 * SM_BN * 16;       // 13824 B
    unsigned char* sImg = smem;                   // [NBUF][NPL planes][IMG_HALF]
    unsigned char* sW = smem + NBUF * NPL * IMG_HALF;  // [NBUF][WPARTS][2 halves][W_HALF + 512]
    const unsigned char* x1 = reinterpret_cast<const unsigned char*>(x1_);
    const unsigned char* x2 = reinterpret_cast<const unsigned char*>(x2_);

    // ---- DMA plan of the image: pieces of 64 consecutive entries of one half; lane l of piece p fills entry 64 p + l.
    // The source voxel of an entry does not depend on the slice: computed once (-1: zero fill, or beyond the image).
    // A wave owns pieces wave, wave + 4, ...: at most MAXP per wave.
    constexpr int MAXP = 8;                       // 4 waves x 8 pieces x 64 = 2048 entries
    int src_vox[MAXP];
#pragma unroll
    for (int j = 0; j < MAXP; ++j) {
        const int e = (wave + 4 * j) * 64 + lane;
        int v = -1;
        if (e < entries) {
            const int bl = e / img, rem = e - bl * img;
            const int ix = rem / (g.Iy * g.Iz), rem2 = rem - ix * (g.Iy * g.Iz);
            const int iy = rem2 / g.Iz, iz = rem2 - iy * g.Iz;
            int s0 = x0 + ix - 1 - g.off, s1 = iy - 1 - g.off, s2 = iz - 1 - g.off;
            bool ok = true;
            if (g.clamp) {
                s0 = min(max(s0, 0), g.Es[0] - 1); s1 = min(max(s1, 0), g.Es[1] - 1); s2 = min(max(s2, 0), g.Es[2] - 1);
            } else {
                ok = s0 >= 0 && s0 < g.Es[0] && s1 >= 0 && s1 < g.Es[1] && s2 >= 0 && s2 < g.Es[2];
            }
            if (ok) v = (((b0 + bl) * g.Es[0] + s0) * g.Es[1] + s1) * g.Es[2] + s2;
        }
        src_vox[j] = v;
    }
    const int npieces = (entries + 63) >> 6;
    auto dma_slice = [&](int c, int buf) {
        const int k0 = c * SM_KC;
        const unsigned char* xs_;
        int Cs, kk;
        if (k0 < C1) { xs_ = x1; Cs = C1; kk = k0; } else { xs_ = x2; Cs = C2; kk = k0 - C1; }
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
            for (int j = 0; j < MAXP; ++j) {
                const int p = wave + 4 * j;
                if (p < npieces) {
                    const void* src = src_vox[j] >= 0 ? (const void*)(xs_ + ((int64_t)src_vox[j] * Cs + kk) * ESZ + pl * 16) : zero16;
                    unsigned char* dst = sImg + (buf * NPL + pl) * IMG_HALF + p * 1024;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
                }
            }
        // weights of the slice: LDS image [part][half][row = tap * 32 + n] of 16-B entries, 864 rows per half = 13.5 DMA
        // pieces of 64 rows -> 14 pieces per half (the last one half full: its upper lanes re-read the last row into slack)
#pragma unroll
        for (int j = 0; j < 7 * WPARTS; ++j) {
            const int q = wave + 4 * j;           // piece index over parts x halves (14 each)
            if (q < 28 * WPARTS) {
                const int part = q / 28, half = (q % 28) / 14, row = (q % 14) * 64 + lane;  // row = tap * 32 + n
                const int rr = min(row, 27 * SM_BN - 1);
                const int tap = rr >> 5, n = rr & 31;
                const bf16* src = SPLIT ? wp + part * lo_offset + ((int64_t)((c * 2 + half) * 27 + tap) * g.N + n0 + n) * 8
                                        : wp + ((int64_t)(c * 27 + tap) * g.N + n0 + n) * SM_KC + half * 8;
                unsigned char* dst = sW + ((buf * WPARTS + part) * 2 + half) * (W_HALF + 512) + (q % 14) * 1024;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            }
        }
    };

    const int c_first = split * g.per_split, c_end = min(g.K / SM_KC, c_first + g.per_split);
    if (loader) {
        // ---- loader waves: the copies of slice c + 1 go out while slice c computes (double-buffered), or behind the
        // barrier that ends slice c (split precision: one buffer); every barrier below has its twin in the computing waves
        if (c_first < c_end) dma_slice(c_first, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int c = c_first; c < c_end; ++c) {
            const int buf = SPLIT ? 0 : ((c - c_first) & 1);
            if (!SPLIT && c + 1 < c_end) dma_slice(c + 1, buf ^ 1);
            if (SPLIT) {
                __syncthreads();
                if (c + 1 < c_end) dma_slice(c + 1, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        return;
    }

    // ---- rows of this lane: M tile m = wave + 4 i, row = 32 m + r -> image entry of its centre
    int a_ent[MTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
        const int row = (wave + 4 * i) * 32 + r;
        int ent = 0;
        if (row < nrows) {
            const int bl = row / per_sample, rem = row - bl * per_sample;
            const int lx = rem / (g.Ev[1] * g.Ev[2]), rem2 = rem - lx * (g.Ev[1] * g.Ev[2]);
            const int ly = rem2 / g.Ev[2], lz = rem2 - ly * g.Ev[2];
            ent = ((bl * g.Ix + lx + 1) * g.Iy + ly + 1) * g.Iz + lz + 1;
        } else {
            ent = g.Iy * g.Iz + g.Iz + 1;  // any entry with a full neighbourhood inside the image
        }
        a_ent[i] = ent * 16 + hh * (SPLIT ? 2 : 1) * IMG_HALF;  // split: this lane's 8 channels = planes 2 hh and 2 hh + 1
    }
    const int w_off = hh * (W_HALF + 512) + r * 16;

    f32x16 acc[MTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    // 8 fp32 -> 8 bf16 hi and 8 bf16 lo
    auto split8 = [](const float4& a, const float4& b, bf16x8& hi, bf16x8& lo) {
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        unsigned h[4], l[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            h[i] = pack_bf16x2(v[2 * i], v[2 * i + 1]);
            l[i] = pack_bf16x2(v[2 * i] - __uint_as_float(h[i] << 16), v[2 * i + 1] - __uint_as_float(h[i] & 0xffff0000u));
        }
        const uint4 uh = make_uint4(h[0], h[1], h[2], h[3]), ul = make_uint4(l[0], l[1], l[2], l[3]);
        hi = *reinterpret_cast<const bf16x8*>(&uh);
        lo = *reinterpret_cast<const bf16x8*>(&ul);
    };
    __syncthreads();  // the first slice has landed (the loader waves waited for it)
    for (int c = c_first; c < c_end; ++c) {
        const int buf = SPLIT ? 0 : ((c - c_first) & 1);
        const unsigned char* A = sImg + buf * NPL * IMG_HALF;
        const unsigned char* W = sW + buf * WPARTS * 2 * (W_HALF + 512) + w_off;
        // fragments of tap t + 1 are read while the MFMAs of tap t issue (two register sets, pinned with
        // sched_group_barrier): one wave per SIMD has nothing else to hide an LDS round trip behind
        struct Frags { bf16x8 w, wl, x[MTW]; float4 xa[SPLIT ? MTW : 1], xb[SPLIT ? MTW : 1]; };
        // The 27 taps run as CHUNKS groups of TPC: one fully unrolled group for the bf16 kernels; three x planes of 9 taps
        // (rolled outer loop) for the split-precision ones, whose 27 x MTW x 3 MFMAs in one basic block cost the
        // instruction scheduler 2-6 minutes of compile time per instantiation (97 % of the build of this file).
        constexpr int CHUNKS = SPLIT ? 3 : 1, TPC = 27 / CHUNKS;
        auto read_tap = [&](int ch, int k, Frags& f) {
            const int ex = CHUNKS == 3 ? ch - 1 : k / 9 - 1, kk = k % 9, ey = kk / 3 - 1, ez = kk % 3 - 1;
            const int tap = CHUNKS == 3 ? ch * 9 + k : k;
            const int toff = ((ex * g.Iy + ey) * g.Iz + ez) * 16;
            f.w = *reinterpret_cast<const bf16x8*>(W + tap * (SM_BN * 16));
            if (SPLIT) f.wl = *reinterpret_cast<const bf16x8*>(W + 2 * (W_HALF + 512) + tap * (SM_BN * 16));
#pragma unroll
            for (int i = 0; i < MTW; ++i) {
                if (SPLIT) {
                    f.xa[SPLIT ? i : 0] = *reinterpret_cast<const float4*>(A + a_ent[i] + toff);
                    f.xb[SPLIT ? i : 0] = *reinterpret_cast<const float4*>(A + a_ent[i] + toff + IMG_HALF);
                } else {
                    f.x[i] = *reinterpret_cast<const bf16x8*>(A + a_ent[i] + toff);
                }
            }
        };
        auto mfma_tap = [&](const Frags& f) {
#pragma unroll
            for (int i = 0; i < MTW; ++i) {
                if (SPLIT) {
                    bf16x8 xh, xl;
                    split8(f.xa[SPLIT ? i : 0], f.xb[SPLIT ? i : 0], xh, xl);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.w, xh, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.w, xl, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.wl, xh, acc[i], 0, 0, 0);
                } else {
                    acc[i] = H16<HF>::mfma(f.w, f.x[i], acc[i]);
                }
            }
        };
        constexpr int NRD = SPLIT ? 2 * MTW + 2 : MTW + 1;  // ds_read_b128 per tap
        constexpr int NMF = SPLIT ? 3 * MTW : MTW;          // MFMAs per tap
#pragma unroll 1
        for (int ch = 0; ch < CHUNKS; ++ch) {
            Frags f0, f1;
            read_tap(ch, 0, f0);
            __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
#pragma unroll
            for (int k = 0; k < TPC; k += 2) {
                if (k + 1 < TPC) read_tap(ch, k + 1, f1);
                mfma_tap(f0);
                if (k + 1 < TPC) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, NMF - 1, 0);
                    if (k + 2 < TPC) read_tap(ch, k + 2, f0);
                    mfma_tap(f1);
                    if (k + 2 < TPC) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, NMF - 1, 0);
                    } else {
                        __builtin_amdgcn_sched_group_barrier(0x008, NMF, 0);
                    }
                } else {
                    __builtin_amdgcn_sched_group_barrier(0x008, NMF, 0);
                }
            }
        }
        if (SPLIT) __syncthreads();  // single-buffered: the next slice is copied only after every wave is done with this one
        __syncthreads();             // the next slice has landed, everybody is done with this one
    }

    // ---- partial tile -> slab[split][b][virtual voxel][N] (fp32).  Lane (r, hh) holds, for tile i, row 32 m + r and
    // channels 8 j + 4 hh + (0..3) in accumulator registers 4 j .. 4 j + 3.
    const int64_t Vv = (int64_t)g.Ev[0] * g.Ev[1] * g.Ev[2];
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
        const int row = (wave + 4 * i) * 32 + r;
        if (row >= nrows) continue;
        const int bl = row / per_sample, rem = row - bl * per_sample;
        const int64_t vox = (int64_t)x0 * g.Ev[1] * g.Ev[2] + rem;  // x slabs are contiguous runs of virtual voxels
        float* dst = slab + (((int64_t)split * g.B + b0 + bl) * Vv + vox) * g.N + n0 + 4 * hh;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<float4*>(dst + 8 * j) = make_float4(acc[i][4 * j], acc[i][4 * j + 1], acc[i][4 * j + 2], acc[i][4 * j + 3]);
    }
}

// host-side launch of one instantiation; every (MTW, SPLIT) pair lives in its own translation unit
// (tdx_conv3_small_i*.hip: the 27-tap x MTW-tile loop is fully unrolled and takes minutes to compile, so the ten
// instantiations build in parallel), declared here for the dispatcher in tdx_conv3_small.hip
#define SMALL_GO_ARGS const void* x1, int C1, const void* x2, int C2, const void* wp, float* slab, const void* zero16, \
                      const SmallGeom& g, size_t lds, int64_t lo_offset, hipStream_t st
template <int MTW, bool SPLIT, bool HF = false>
static int small_go(SMALL_GO_ARGS) {
    static_assert(!(SPLIT && HF), "split precision has bf16 halves");
    auto kern = conv3_small_kernel<MTW, SPLIT, HF>;
    static size_t attr = 0;
    if (lds > attr) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr = lds;
    }
    const int ngroups = ceil_div(g.B, g.nbg) * g.gx;
    hipLaunchKernelGGL(kern, dim3(g.N / SM_BN, ngroups, g.nsplit), dim3(256 + 64 * SM_LOADERS), lds, st, x1, C1, x2, C2, (const bf16*)wp, slab,
                       zero16, g, lo_offset);
    return tdx_launch_status();
}
#define SMALL_INSTANCE(MTW, SPLIT, NAME) \
    int NAME(SMALL_GO_ARGS) { return small_go<MTW, SPLIT>(x1, C1, x2, C2, wp, slab, zero16, g, lds, lo_offset, st); }
#define SMALL_INSTANCE_F16(MTW, NAME) \
    int NAME(SMALL_GO_ARGS) { return small_go<MTW, false, true>(x1, C1, x2, C2, wp, slab, zero16, g, lds, lo_offset, st); }

