// bf16 MFMA 3x3x3 convolution for SMALL grids (the deep U-Net levels: 24 x 8 x 6 and 12 x 4 x 3 voxels, 256-1024
// channels), forward and data gradient.  gfx950.
//
// The brick kernel (tdx_conv3_mfma.hip) gives such a launch 96-288 workgroups, each filling 28-75 % of a 4 x 8 x 8
// brick and walking K = 27 x 512 alone on its CU, one 73-KB stage per 2.8 us: 130-600 TFLOP/s, and another 55-80 us
// per layer for the halo-shell kernel of the data gradient.  Here the GEMM is cut the other way:
//
//   M rows     = voxels of a "virtual grid", packed densely into 32-row M tiles regardless of brick shapes.  Forward:
//                the real grid, sources clamped (replicate padding).  Data gradient: the PADDED grid (X+2)(Y+2)(Z+2)
//                with zero sources outside the real grid -- the adjoint on the padded grid, whose rows the reduce pass
//                folds onto the voxels they clamp to; no separate shell launch.
//   workgroup  = (row group, 32 output channels, K split).  A row group is a few whole samples or an x slab of one,
//                at most 28 M tiles = 7 per wave; its sources live in LDS as a zero- / clamp-filled image with a
//                one-voxel rim, so a tap is a uniform offset and any row can sit in any lane.
//   K split    = contiguous ranges of 16-channel slices; every workgroup stores its fp32 partial tile to
//                slab[split][virtual voxel][channel], and conv3_small_reduce_kernel sums the splits (and, for the data
//                gradient, the up to 8 padded positions that clamp onto a voxel), adds bias / the residual addend,
//                converts and stores.  The split is what fills the chip: 12 x 4 x 3 x 6 samples is 27 M tiles in all.
//   staging    = LDS-DMA (global_load_lds_dwordx4), double-buffered: slice c + 1 lands while slice c's 27 taps run;
//                zero fill = DMA from a zeroed 16-B block.  No staging registers: 7 accumulator tiles + fragments fit.
//
// Slab traffic is the price (S x rows x N x 4 B written and read once); at these sizes it is 5-40 MB per layer.
#include "tdx_common.h"
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
// 4 channels) and split into hi / lo when a fragment is read; the weights arrive pre-split ([2 parts][K/16][27][N][16]
// bf16, lo_offset elements apart).  fp32 images and two weight parts do not fit twice: that mode is single-buffered.
template <int MTW, bool SPLIT>
__global__ void __launch_bounds__(256, 1)
conv3_small_kernel(const void* __restrict__ x1_, int C1, const void* __restrict__ x2_, int C2, const bf16* __restrict__ wp,
                   float* __restrict__ slab, const void* __restrict__ zero16, SmallGeom g, int64_t lo_offset) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
                const bf16* src = wp + part * lo_offset + ((int64_t)(c * 27 + tap) * g.N + n0 + n) * SM_KC + half * 8;
                unsigned char* dst = sW + ((buf * WPARTS + part) * 2 + half) * (W_HALF + 512) + (q % 14) * 1024;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            }
        }
    };

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

    const int c_first = split * g.per_split, c_end = min(g.K / SM_KC, c_first + g.per_split);
    auto drain_and_sync = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
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
    if (c_first < c_end) dma_slice(c_first, 0);
    drain_and_sync();
    for (int c = c_first; c < c_end; ++c) {
        const int buf = SPLIT ? 0 : ((c - c_first) & 1);
        if (!SPLIT && c + 1 < c_end) dma_slice(c + 1, buf ^ 1);  // lands while this slice's taps run (buffer last read one slice ago)
        const unsigned char* A = sImg + buf * NPL * IMG_HALF;
        const unsigned char* W = sW + buf * WPARTS * 2 * (W_HALF + 512) + w_off;
        // fragments of tap t + 1 are read while the MFMAs of tap t issue (two register sets, pinned with
        // sched_group_barrier): one wave per SIMD has nothing else to hide an LDS round trip behind
        struct Frags { bf16x8 w, wl, x[MTW]; float4 xa[SPLIT ? MTW : 1], xb[SPLIT ? MTW : 1]; };
        auto read_tap = [&](int tap, Frags& f) {
            const int ex = tap / 9 - 1, ey = (tap / 3) % 3 - 1, ez = tap % 3 - 1;
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
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.w, f.x[i], acc[i], 0, 0, 0);
                }
            }
        };
        constexpr int NRD = SPLIT ? 2 * MTW + 2 : MTW + 1;  // ds_read_b128 per tap
        constexpr int NMF = SPLIT ? 3 * MTW : MTW;          // MFMAs per tap
        Frags f0, f1;
        read_tap(0, f0);
        __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
#pragma unroll
        for (int tap = 0; tap < 27; tap += 2) {
            if (tap + 1 < 27) read_tap(tap + 1, f1);
            mfma_tap(f0);
            if (tap + 1 < 27) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, NMF - 1, 0);
                if (tap + 2 < 27) read_tap(tap + 2, f0);
                mfma_tap(f1);
                if (tap + 2 < 27) {
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
        if (SPLIT) {  // single-buffered: the next slice is copied only after every wave is done with this one
            __syncthreads();
            if (c + 1 < c_end) dma_slice(c + 1, 0);
        }
        drain_and_sync();
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

// out[b, u, :] = sum over splits and over the virtual voxels v that map onto real voxel u of slab[split][b][v][:]
//                (+ bias) (+ addend), as T.  fold == 0: v = u (forward).  fold == 1: the virtual grid is the padded
//                grid and v ranges over the positions that clamp onto u (data gradient).  The result has N channels,
//                split over out1 (channels [0, D1)) and out2.
template <typename T>
__global__ void __launch_bounds__(256)
conv3_small_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ bias, T* __restrict__ out1, int D1,
                          T* __restrict__ out2, const T* __restrict__ add1, const T* __restrict__ add2, int B, int X,
                          int Y, int Z, int N, int nsplit, int fold) {
    const int groups = N >> 3;
    const int64_t total = (int64_t)B * X * Y * Z * groups;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int cg = (int)(idx % groups);
    int64_t u = idx / groups;
    const int uz = (int)(u % Z); int64_t t = u / Z;
    const int uy = (int)(t % Y); t /= Y;
    const int ux = (int)(t % X);
    const int b = (int)(t / X);
    const int E[3] = {X + 2 * fold, Y + 2 * fold, Z + 2 * fold}, c[3] = {ux, uy, uz}, R[3] = {X, Y, Z};
    int lo[3], hi[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        lo[a] = fold ? (c[a] == 0 ? 0 : c[a] + 1) : c[a];
        hi[a] = fold ? (c[a] == R[a] - 1 ? R[a] + 1 : c[a] + 1) : c[a];
    }
    const int64_t Vv = (int64_t)E[0] * E[1] * E[2];
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = bias ? bias[cg * 8 + j] : 0.f;
    for (int s = 0; s < nsplit; ++s)
        for (int v0 = lo[0]; v0 <= hi[0]; ++v0)
            for (int v1 = lo[1]; v1 <= hi[1]; ++v1)
                for (int v2 = lo[2]; v2 <= hi[2]; ++v2) {
                    const float* p = slab + (((int64_t)s * B + b) * Vv + ((int64_t)v0 * E[1] + v1) * E[2] + v2) * N + cg * 8;
                    const float4 a = *reinterpret_cast<const float4*>(p), bq = *reinterpret_cast<const float4*>(p + 4);
                    acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w;
                    acc[4] += bq.x; acc[5] += bq.y; acc[6] += bq.z; acc[7] += bq.w;
                }
    const int n = cg * 8;
    const int64_t vox = idx / groups;
    const bool first = n < D1;
    T* dst = first ? out1 + vox * D1 + n : out2 + vox * (N - D1) + (n - D1);
    const T* asrc = first ? (add1 ? add1 + vox * D1 + n : nullptr) : (add2 ? add2 + vox * (N - D1) + (n - D1) : nullptr);
    Vec8<T> o;
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

// ------------------------------------------------------------------------------------------------ host side
// Plan a launch; false if the small-grid kernel does not apply (grid too large, shapes, no arena, arena too small).
static bool small_plan(SmallGeom& g, int B, int X, int Y, int Z, int K, int N, bool data_gradient, bool split,
                       size_t arena_bytes, size_t& lds_bytes) {
    static const int mode = getenv("TDX_CONV3_SMALL") ? atoi(getenv("TDX_CONV3_SMALL")) : 1;
    if (mode == 0) return false;
    if ((K % SM_KC) || (N % SM_BN) || K < 128) return false;
    const int pad = data_gradient ? 1 : 0;
    g.B = B;
    g.Ev[0] = X + 2 * pad; g.Ev[1] = Y + 2 * pad; g.Ev[2] = Z + 2 * pad;
    g.Es[0] = X; g.Es[1] = Y; g.Es[2] = Z;
    g.off = pad; g.clamp = data_gradient ? 0 : 1;
    g.K = K; g.N = N;
    const int64_t rows_total = (int64_t)B * g.Ev[0] * g.Ev[1] * g.Ev[2];
    // Where it pays (tools/conv_bench.py, B = 6, us per launch, against brick kernel [+ halo-shell kernel]):
    //   12 x 4 x 3, 512 -> 512:    forward 33 vs 91;   data gradient 73 vs 141 + 55
    //   24 x 8 x 6, 512 -> 512:    forward 154 vs 162; 1024 -> 256: 149 vs 181; 256 -> 512: 90 vs 85
    //   24 x 8 x 6 data gradients: 183-353 vs 166-297 (the padded grid has 1.8x the rows there) -> brick kernels
    // TDX_CONV3_SMALL_ROWS moves the row gate (tests, A/B runs; read per call).
    const char* env_rows = getenv("TDX_CONV3_SMALL_ROWS");
    const int max_total = env_rows ? atoi(env_rows) : 6000;
    const bool deep_forward = !data_gradient && K >= 512 && rows_total <= 24000;
    if (rows_total > max_total && !(deep_forward && !env_rows)) return false;
    const int plane = g.Ev[1] * g.Ev[2], sample = g.Ev[0] * plane;
    const int max_rows = SM_MAX_TILES * 32;
    if (plane > max_rows) return false;
    if (sample <= max_rows) {  // whole samples per group; prefer >= 2 groups so that two row groups share the weights' L2 lines
        g.nbg = max(1, min(B, max_rows / sample));
        if (g.nbg == B && B > 1) g.nbg = (B + 1) / 2;
        g.xs = g.Ev[0]; g.gx = 1;
    } else {
        g.nbg = 1;
        const int slabs = ceil_div(sample, max_rows);
        g.xs = ceil_div(g.Ev[0], slabs);
        g.gx = ceil_div(g.Ev[0], g.xs);
    }
    g.Ix = g.xs + 2; g.Iy = g.Ev[1] + 2; g.Iz = g.Ev[2] + 2;
    const int img = g.Ix * g.Iy * g.Iz;
    auto lds_for = [&](int nbg) { return (size_t)4 * (((size_t)nbg * img + 63) & ~(size_t)63) * 16 + (size_t)4 * (27 * SM_BN * 16 + 512); };
    while (g.nbg > 1 && ((int64_t)g.nbg * img > 2048 || lds_for(g.nbg) > 160 * 1024)) --g.nbg;  // DMA plan: 4 x 8 x 64 entries
    if ((int64_t)g.nbg * img > 2048) return false;
    lds_bytes = lds_for(g.nbg);
    if (lds_bytes > 160 * 1024) return false;
    // K splits.  One workgroup per CU (LDS), so a launch runs in rounds of 256 workgroups; a split count is judged by
    // rounds x slices per workgroup (at ~3.5 us per slice) plus the slab round trip it causes (written and read once
    // at ~5 TB/s), the smallest estimate wins: 12 x 4 x 3 forward 8 splits (1 round x 4 slices), 24 x 8 x 6 with 192
    // tile-groups 4 splits (3 full rounds x 8 slices instead of 2 rounds, the second half empty, x 16)
    const int ngroups = ceil_div(B, g.nbg) * g.gx, nslices = K / SM_KC;
    const int64_t base = (int64_t)ngroups * (N / SM_BN);
    int splits = 1;
    double best = 1e30;
    for (int sp = 1; sp <= 16 && sp * 2 <= std::max(nslices, 2); ++sp) {
        if ((size_t)sp * rows_total * N * 4 + 64 > arena_bytes) break;
        const double rounds = (double)ceil_div(base * sp, 256), per = (double)ceil_div(nslices, sp);
        const double est = rounds * per * (split ? 9e-6 : 3.5e-6) + (sp > 1 ? 2.0 * sp * rows_total * N * 4 / 5e12 : rows_total * N * 8.0 / 5e12);
        if (est < best * 0.97) { best = est; splits = sp; }
    }
    if ((size_t)splits * rows_total * N * 4 + 64 > arena_bytes) return false;
    g.per_split = ceil_div(nslices, splits);
    g.nsplit = ceil_div(nslices, g.per_split);
    return true;
}

template <int MTW, bool SPLIT>
static int small_go(const void* x1, int C1, const void* x2, int C2, const void* wp, float* slab, const void* zero16,
                    const SmallGeom& g, size_t lds, int64_t lo_offset, hipStream_t st) {
    auto kern = conv3_small_kernel<MTW, SPLIT>;
    static size_t attr = 0;
    if (lds > attr) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr = lds;
    }
    const int ngroups = ceil_div(g.B, g.nbg) * g.gx;
    hipLaunchKernelGGL(kern, dim3(g.N / SM_BN, ngroups, g.nsplit), dim3(256), lds, st, x1, C1, x2, C2, (const bf16*)wp, slab,
                       zero16, g, lo_offset);
    return tdx_launch_status();
}

template <bool SPLIT>
static int small_dispatch(int mtw, const void* x1, int C1, const void* x2, int C2, const void* wp, float* slab,
                          const void* z16, const SmallGeom& g, size_t lds, int64_t lo, hipStream_t st) {
    switch (mtw) {
        case 1: case 2: case 3: return small_go<3, SPLIT>(x1, C1, x2, C2, wp, slab, z16, g, lds, lo, st);
        case 4: return small_go<4, SPLIT>(x1, C1, x2, C2, wp, slab, z16, g, lds, lo, st);
        case 5: return small_go<5, SPLIT>(x1, C1, x2, C2, wp, slab, z16, g, lds, lo, st);
        case 6: return small_go<6, SPLIT>(x1, C1, x2, C2, wp, slab, z16, g, lds, lo, st);
        default: return small_go<7, SPLIT>(x1, C1, x2, C2, wp, slab, z16, g, lds, lo, st);
    }
}

// Forward (data_gradient == false: y = conv3([x1 | x2]) + bias, N = Cout) or data gradient (x1 = dy with K = C1
// channels, x2 unused; result N channels split over out1 [0, D1) / out2, plus addends).  split == false: bf16 tensors,
// wp = the bf16 packed weight; split == true: fp32 tensors, wp = the split-precision packed weight (hi image, lo image).
// Returns TDX_ESHAPE when the launch is not a small-grid case (the caller then takes the brick kernels).
int conv3_small_launch(const void* x1, int C1, const void* x2, int C2, const void* wp, const float* bias, void* out1, int D1,
                       void* out2, const void* add1, const void* add2, int B, int X, int Y, int Z, int N, bool data_gradient,
                       bool split, hipStream_t st) {
    char* arena = (char*)tdx_scratch_ptr();
    if (arena == nullptr) return TDX_ESHAPE;
    if ((C1 % SM_KC) || (C2 % SM_KC)) return TDX_ESHAPE;
    SmallGeom g;
    size_t lds = 0;
    if (!small_plan(g, B, X, Y, Z, C1 + C2, N, data_gradient, split, tdx_scratch_bytes(), lds)) return TDX_ESHAPE;
    float* slab = reinterpret_cast<float*>(arena + 64);
    const int rows = g.nbg * g.xs * g.Ev[1] * g.Ev[2];
    const int mtw = ceil_div(ceil_div(rows, 32), 4);
    const int64_t lo = (int64_t)27 * (C1 + C2) * N;  // elements between the hi and the lo weight image
    const int rc = split ? small_dispatch<true>(mtw, x1, C1, x2, C2, wp, slab, arena, g, lds, lo, st)
                         : small_dispatch<false>(mtw, x1, C1, x2, C2, wp, slab, arena, g, lds, 0, st);
    if (rc != TDX_OK) return rc;
    const int64_t total = (int64_t)B * X * Y * Z * (N / 8);
    if (split)
        hipLaunchKernelGGL(conv3_small_reduce_kernel<float>, dim3(ceil_div(total, 256)), dim3(256), 0, st, slab, bias, (float*)out1,
                           D1, (float*)out2, (const float*)add1, (const float*)add2, B, X, Y, Z, N, g.nsplit, data_gradient ? 1 : 0);
    else
        hipLaunchKernelGGL(conv3_small_reduce_kernel<bf16>, dim3(ceil_div(total, 256)), dim3(256), 0, st, slab, bias, (bf16*)out1,
                           D1, (bf16*)out2, (const bf16*)add1, (const bf16*)add2, B, X, Y, Z, N, g.nsplit, data_gradient ? 1 : 0);
    return tdx_launch_status();
}
