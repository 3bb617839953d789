// bf16 MFMA implicit-GEMM 3x3x3 convolution for gfx950 (forward and, with zero padding on
// the padded grid, the data gradient).
//
//   M = output voxels (a 4 x 8 x 8 brick per workgroup), N = output channels (BN = 32/64
//   per workgroup), K = 27 taps x input channels, walked in 16-channel slices.
//
// Per K slice the workgroup stages ONE halo'd input brick (6 x 10 x 10 voxels x 16 ch =
// 19 KB) and the slice's weights for all 27 taps (27 x BN x 16 = 54 KB at BN = 64) into
// LDS; every tap then re-reads the same brick at a shifted voxel offset, so the 27-fold
// input reuse of the convolution is served from LDS, not from L2/HBM.  78 KB of LDS per
// workgroup -> two workgroups per CU.
//
// Pipeline: the global loads of slice c+1 are issued into registers right after slice c has
// been written to LDS and stay in flight during the 27 x 2 x NT MFMAs of slice c (the wait
// lands at the next LDS write), so L2/HBM latency hides behind the matrix work; the second
// workgroup on the CU covers the short LDS-write window.
//
// Wave w owns the x = w slab of the brick: 8 x 8 voxels = two 32-voxel M tiles
// (r <-> y = 4*mt + (r & 3), z = r >> 2) x NT 32-channel N tiles.  The MFMA is issued as
// D^T = W^T X^T (weights as the A operand), so a lane owns ONE voxel and 4 consecutive
// channels per accumulator quad: the epilogue packs them to 8-B writes of an LDS output
// tile [256 voxels][BN] that is then stored to HBM in whole 16-B-per-lane voxel rows.
//
// LDS images (conflict-free for ds_read_b128, checked by exhaustive enumeration of the
// 16-lane read groups over all tap offsets), each split in two half-planes holding
// channels 0-7 and 8-15 of the slice so that a lane's 16-B fragment is one plane entry:
//   brick  : [half][voxel h = (hx*10 + hy)*12 + hz] (z stride padded 10 -> 12), 16 B each
//   weights: [half][tap*BN + n], 16 B each
#include "tdx_common.h"
#include "tdx_conv3.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define M3_BX 4                                // x extent of the brick at MT = 2 (2*MT in general)
#define M3_BY 8
#define M3_BZ 8
#define M3_HX (M3_BX + 2)
#define M3_HY (M3_BY + 2)
#define M3_HZ (M3_BZ + 2)
#define M3_SZ 12                               // padded z stride of the LDS brick
#define M3_NVOX_HALO (M3_HX * M3_HY * M3_HZ)   // 600
#define M3_BRICK_BYTES (M3_HX * M3_HY * M3_SZ * 32 + 128)  // 23168
#define M3_KC 16
#define M3_DEFAULT_PP false

bool conv3_mfma_supported(int C1, int C2, int Cout) {
    return C1 > 0 && (C1 % M3_KC) == 0 && (C2 % M3_KC) == 0 && (Cout % 32) == 0;
}

// two half-planes (channels 0-7 / 8-15 of the slice), 16 B per voxel: with the z stride of 12
// the 16 voxels of every ds_read_b128 lane group are distinct mod 16 -> conflict-free with
// plain affine addresses (tap offsets become instruction immediates)
#define M3_APLANE (M3_HX * M3_HY * M3_SZ * 16 + 64)  // +64 B: the two halves of a voxel land 4 slots apart
__device__ __forceinline__ int brick_addr(int h, int half) { return half * M3_APLANE + h * 16; }
// output tile rows of 64 B (BN = 32) or 128 B (BN = 64); 16-B chunk c of row v at c ^ swizzle(v)
template <int BN>
__device__ __forceinline__ int out_addr(int v, int c) {
    if (BN == 64) return v * 128 + ((c ^ (v & 7)) << 4);
    return v * 64 + ((c ^ ((v >> 1) & 3)) << 4);
}

// MT = 32-voxel M tiles per wave: 2 (4x8x8 brick, 2 workgroups/CU) or 4 (8x8x8 brick, one
// workgroup/CU with twice the register tile -> 25 % fewer LDS fragment bytes per MFMA).
// PP = ping-pong LDS: one workgroup per CU owns two stage buffers; slice c+1 is written into the
// idle buffer at the top of slice c's MFMA phase (its global loads were issued a slice earlier),
// so there is ONE barrier per slice and the staging costs only its issue slots.
template <int NT, int MT, bool ZERO_PAD, bool PP>
__global__ void __launch_bounds__(256, (MT == 2 && !PP) ? 2 : 1)
conv3_mfma_kernel(const bf16* __restrict__ x1, int C1, const bf16* __restrict__ x2, int C2,
                  const bf16* __restrict__ wp, const float* __restrict__ bias, bf16* __restrict__ y, Conv3Geom g,
                  int Cout, int nbx, int nby, int nbz, double* __restrict__ gn_acc, bf16* __restrict__ d1, int D1,
                  bf16* __restrict__ d2, const bf16* __restrict__ a1, const bf16* __restrict__ a2) {
    constexpr int BN = NT * 32;
    constexpr int BX = 2 * MT, HX = BX + 2;
    constexpr int NVOX = BX * M3_BY * M3_BZ;                 // 256 / 512 output voxels
    constexpr int NHALO = HX * M3_HY * M3_HZ;                // 600 / 1000 staged voxels
    constexpr int APLANE = HX * M3_HY * M3_SZ * 16 + 64;     // one half-plane of the brick image
    constexpr int BRICK_BYTES = 2 * APLANE;
    constexpr int B_PLANE = 27 * BN * 16 + 64;
    constexpr int STAGE_BYTES = BRICK_BYTES + 2 * B_PLANE;   // one stage buffer (brick + weights)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;

    // block -> (n tile, b, brick)
    int bid = blockIdx.x;
    const int bz = bid % nbz; bid /= nbz;
    const int by = bid % nby; bid /= nby;
    const int bx = bid % nbx; bid /= nbx;
    const int b = bid;
    const int n0 = blockIdx.y * BN;
    const int ox0 = bx * BX, oy0 = by * M3_BY, oz0 = bz * M3_BZ;
    const int Cin = C1 + C2;

    // ---- staging plan for the input brick: 1200 16-B pieces, <= 5 per thread.
    // a_src: (voxel index in the input grid) * 2 + half, or -1 for zero fill / no piece
    constexpr int A_PIECES = NHALO * 2;
    constexpr int A_PER_THREAD = (A_PIECES + 255) / 256;  // 5 / 8
    constexpr int B_PIECES = 27 * BN * 2;
    constexpr int B_PER_THREAD = (B_PIECES + 255) / 256;  // 14 (NT=2) / 7 (NT=1)
    int a_src[A_PER_THREAD];
    int a_dst[A_PER_THREAD];
#pragma unroll
    for (int i = 0; i < A_PER_THREAD; ++i) {
        const int p = tid + i * 256;
        a_dst[i] = -1;
        a_src[i] = -1;
        if (p < A_PIECES) {
            // lanes 0-3 / 4-7 of every 8-lane group: 4 consecutive voxels x the two halves
            const int hv = ((p >> 3) << 2) + (p & 3), half = (p >> 2) & 1;
            const int hx = hv / (M3_HY * M3_HZ), rem = hv - hx * (M3_HY * M3_HZ);
            const int hy = rem / M3_HZ, hz = rem - hy * M3_HZ;
            a_dst[i] = half * APLANE + ((hx * M3_HY + hy) * M3_SZ + hz) * 16;
            int sx = ox0 + hx - 1 + g.off, sy = oy0 + hy - 1 + g.off, sz = oz0 + hz - 1 + g.off;
            bool ok = true;
            if (ZERO_PAD) {
                ok = sx >= 0 && sx < g.Xi && sy >= 0 && sy < g.Yi && sz >= 0 && sz < g.Zi;
            } else {
                sx = min(max(sx, 0), g.Xi - 1); sy = min(max(sy, 0), g.Yi - 1); sz = min(max(sz, 0), g.Zi - 1);
            }
            if (ok) a_src[i] = ((sx * g.Yi + sy) * g.Zi + sz) * 2 + half;
        }
    }
    const int64_t batch_vox = (int64_t)b * g.Xi * g.Yi * g.Zi;

    // weight staging role of this thread
    const int b_half = (tid >> 2) & 1;
    const int b_row0 = ((tid >> 3) << 2) + (tid & 3);                       // 0..127
    const int b_goff = ((b_row0 / BN) * Cout + (b_row0 % BN)) * 16 + b_half * 8;  // elements
    const int b_dst = b_half * B_PLANE + b_row0 * 16;

    uint4 areg[A_PER_THREAD], breg[B_PER_THREAD];
    auto load_slice = [&](int c) {
        const int k0 = c * M3_KC;
        const bf16* xs;
        int Cs, kk;
        if (k0 < C1) { xs = x1; Cs = C1; kk = k0; } else { xs = x2; Cs = C2; kk = k0 - C1; }
        xs += batch_vox * Cs + kk;
#pragma unroll
        for (int i = 0; i < A_PER_THREAD; ++i) {
            areg[i] = make_uint4(0, 0, 0, 0);
            if (a_src[i] >= 0)
                areg[i] = *reinterpret_cast<const uint4*>(xs + (int64_t)(a_src[i] >> 1) * Cs + (a_src[i] & 1) * 8);
        }
        // weights: thread -> (row = b_row0 + 128 i, half); 128 rows = 128/BN taps per step, so both
        // the global and the LDS address advance by a constant per i
        const bf16* wc = wp + (int64_t)c * 27 * Cout * 16 + (int64_t)n0 * 16 + b_goff;
#pragma unroll
        for (int i = 0; i < B_PER_THREAD; ++i) {
            breg[i] = make_uint4(0, 0, 0, 0);
            if (b_row0 + 128 * i < 27 * BN)
                breg[i] = *reinterpret_cast<const uint4*>(wc + (int64_t)i * (128 / BN) * Cout * 16);
        }
    };
    auto store_slice = [&](int buf) {
        unsigned char* sA = smem + buf * STAGE_BYTES;
        unsigned char* sB = sA + BRICK_BYTES;
#pragma unroll
        for (int i = 0; i < A_PER_THREAD; ++i)
            if (a_dst[i] >= 0) *reinterpret_cast<uint4*>(sA + a_dst[i]) = areg[i];
#pragma unroll
        for (int i = 0; i < B_PER_THREAD; ++i)
            if (b_row0 + 128 * i < 27 * BN) *reinterpret_cast<uint4*>(sB + b_dst + i * 2048) = breg[i];
    };

    // ---- per-lane fragment bases
    // M tile mt of wave w: x = w*(MT/2) + (mt >> 1), y = 4*(mt & 1) + (r & 3), z = r >> 2
    int a_h[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
        a_h[mt] = ((wave * (MT / 2) + (mt >> 1) + 1) * M3_HY + (4 * (mt & 1) + (r & 3) + 1)) * M3_SZ + ((r >> 2) + 1);
    int b_off[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = nt * 32 + r;
        b_off[nt] = hh * B_PLANE + n * 16;
    }

    f32x16 acc[NT][MT];  // D[row = channel][col = voxel]
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[nt][mt][i] = 0.f;

    const int nchunks = Cin / M3_KC;
    // the 27 x (MT x NT) MFMAs of one slice; fragments of tap t+1 are read while the MFMAs of tap t
    // issue (two register sets)
    auto compute_slice = [&](int buf) {
        const unsigned char* sA = smem + buf * STAGE_BYTES;
        const unsigned char* sB = sA + BRICK_BYTES;
        constexpr int PD = PP ? 3 : 1;       // fragment prefetch distance in taps (ring of PD + 1 sets):
        constexpr int RING = PD + 1;         // one wave per SIMD (PP) has to cover the LDS latency alone
        bf16x8 xf[RING][MT], wf[RING][NT];
        auto read_frags = [&](int tap, int buf) {
            const int ex = tap / 9 - 1, ey = (tap / 3) % 3 - 1, ez = tap % 3 - 1;
            const int toff = (ex * M3_HY + ey) * M3_SZ + ez;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                xf[buf][mt] = *reinterpret_cast<const bf16x8*>(sA + hh * APLANE + (a_h[mt] + toff) * 16);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                wf[buf][nt] = *reinterpret_cast<const bf16x8*>(sB + tap * (BN * 16) + b_off[nt]);
        };
#pragma unroll
        for (int t = 0; t < PD; ++t) read_frags(t, t);
        __builtin_amdgcn_sched_group_barrier(0x100, PD * (MT + NT), 0);  // DS_READ: the first PD taps
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) {
            if (tap + PD < 27) read_frags(tap + PD, (tap + PD) % RING);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[tap % RING][nt], xf[tap % RING][mt], acc[nt][mt], 0, 0, 0);
            // pin the interleave: one fragment read of tap+PD behind each MFMA of tap
            if (tap + PD < 27) {
#pragma unroll
                for (int k = 0; k < MT * NT; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
                    if (k < MT + NT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS_READ
                }
            } else {
                __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 0);
            }
        }
    };

    load_slice(0);
    if constexpr (PP) {
        store_slice(0);
        __syncthreads();
        if (nchunks > 1) load_slice(1);
        for (int c = 0; c < nchunks; ++c) {
            if (c + 1 < nchunks) {
                store_slice((c + 1) & 1);                  // buffer last read during slice c-1
                if (c + 2 < nchunks) load_slice(c + 2);    // in flight for a whole slice
            }
            compute_slice(c & 1);
            __syncthreads();
        }
    } else {
        for (int c = 0; c < nchunks; ++c) {
            __syncthreads();  // previous slice's fragment reads are done
            store_slice(0);
            __syncthreads();
            if (c + 1 < nchunks) load_slice(c + 1);  // in flight during the MFMAs below
            compute_slice(0);
        }
        __syncthreads();
    }

    // ---------------- epilogue.  Lane (r, hh) of wave w holds, for M tile mt, voxel
    // (x = w, y = 4 mt + (r & 3), z = r >> 2) and channels nt*32 + 8 j + 4 hh + (0..3) in
    // accumulator registers 4 j .. 4 j + 3.
    unsigned char* sO = smem;  // [NVOX voxels][BN] bf16, voxel v = (x*8 + y)*8 + z
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ch = nt * 32 + 8 * j + 4 * hh;
            float bv[4] = {0.f, 0.f, 0.f, 0.f};
            if (bias) {
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[e] = bias[n0 + ch + e];
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int v = ((wave * (MT / 2) + (mt >> 1)) * 8 + 4 * (mt & 1) + (r & 3)) * 8 + (r >> 2);
                const unsigned lo = (unsigned)f32_to_bf16_bits(acc[nt][mt][4 * j] + bv[0]) |
                                    ((unsigned)f32_to_bf16_bits(acc[nt][mt][4 * j + 1] + bv[1]) << 16);
                const unsigned hi = (unsigned)f32_to_bf16_bits(acc[nt][mt][4 * j + 2] + bv[2]) |
                                    ((unsigned)f32_to_bf16_bits(acc[nt][mt][4 * j + 3] + bv[3]) << 16);
                *reinterpret_cast<uint2*>(sO + out_addr<BN>(v, ch >> 3) + (ch & 7) * 2) = make_uint2(lo, hi);
            }
        }
    __syncthreads();
    // store loop: thread -> 16-B chunk (tid % CHUNKS) of voxels tid / CHUNKS + (256 / CHUNKS) i.
    // The same registers feed the fused GroupNorm statistics (forward only): per-channel sum and
    // sum of squares of the (bf16-rounded) tile, reduced over the brick through LDS and merged
    // across bricks with one f64 atomic per (channel, moment) -- saves the read pass over y.
    constexpr int CHUNKS = BN / 8;
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
#pragma unroll
    for (int i = 0; i < CHUNKS * NVOX / 256; ++i) {
        const int p = tid + i * 256;
        const int v = p / CHUNKS, cidx = p % CHUNKS;
        const int ox = ox0 + (v >> 6), oy = oy0 + ((v >> 3) & 7), oz = oz0 + (v & 7);
        if (ox < g.Xo && oy < g.Yo && oz < g.Zo) {
            const uint4 val = *reinterpret_cast<const uint4*>(sO + out_addr<BN>(v, cidx));
            bool direct = false;
            if (ZERO_PAD && d1 != nullptr) {
                // data gradient: padded position (ox,oy,oz) = original voxel + 1.  Interior positions
                // go straight to dx (split over the two inputs of a concatenated conv); only the
                // halo shell is written to the padded workspace for the face fix-up.
                const int ux = ox - 1, uy = oy - 1, uz = oz - 1;
                if (ux >= 0 && ux < g.Xi && uy >= 0 && uy < g.Yi && uz >= 0 && uz < g.Zi) {
                    const int64_t u = (((int64_t)b * g.Xi + ux) * g.Yi + uy) * g.Zi + uz;
                    const int n = n0 + cidx * 8;
                    // optional fused addend (the gradient arriving over the block's residual path)
                    const bool lo = n < D1;
                    bf16* dst = lo ? d1 + u * D1 + n : d2 + u * (Cout - D1) + (n - D1);
                    const bf16* asrc = lo ? (a1 ? a1 + u * D1 + n : nullptr) : (a2 ? a2 + u * (Cout - D1) + (n - D1) : nullptr);
                    if (asrc) {
                        Vec8<bf16> va, vb;
                        va.load(reinterpret_cast<const bf16*>(&val));
                        vb.load(asrc);
#pragma unroll
                        for (int e = 0; e < 8; ++e) va.v[e] += vb.v[e];
                        va.store(dst);
                    } else {
                        *reinterpret_cast<uint4*>(dst) = val;
                    }
                    direct = true;
                }
            }
            if (!direct)
                *reinterpret_cast<uint4*>(y + ((((int64_t)b * g.Xo + ox) * g.Yo + oy) * g.Zo + oz) * Cout + n0 + cidx * 8) = val;
            if (gn_acc != nullptr) {
                const unsigned wds[4] = {val.x, val.y, val.z, val.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float lo = __uint_as_float(wds[e] << 16), hi = __uint_as_float(wds[e] & 0xffff0000u);
                    s1[2 * e] += lo; s2[2 * e] += lo * lo;
                    s1[2 * e + 1] += hi; s2[2 * e + 1] += hi * hi;
                }
            }
        }
    }
    if (gn_acc != nullptr) {
        constexpr int NP = 256 / CHUNKS;  // threads per chunk column
        float* red = reinterpret_cast<float*>(smem + NVOX * BN * 2);  // [NP][BN][2], behind the output tile
        const int cidx = tid % CHUNKS, part = tid / CHUNKS;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            red[(part * BN + cidx * 8 + e) * 2] = s1[e];
            red[(part * BN + cidx * 8 + e) * 2 + 1] = s2[e];
        }
        __syncthreads();
        if (tid < BN * 2) {
            float t = 0.f;
#pragma unroll 8
            for (int pp = 0; pp < NP; ++pp) t += red[pp * BN * 2 + tid];
            // 32 replicas of the [B][Cout][2] table, picked by brick index: all workgroups of a
            // sample would otherwise hammer the same 2*Cout addresses
            const int rep = blockIdx.x & (TDX_GN_REPLICAS - 1);
            atomicAdd(&gn_acc[(((size_t)rep * g.B + b) * Cout + n0) * 2 + tid], (double)t);
        }
    }
}

int conv3_mfma_launch(const void* x1, int C1, const void* x2, int C2, const void* wp, const float* bias, void* y,
                      const Conv3Geom& g, int Cout, bool zero_pad, hipStream_t st, double* gn_acc, void* d1, int D1,
                      void* d2, const void* a1, const void* a2) {
    const int NT = (Cout % 64 == 0) ? 2 : 1;
    const int BN = NT * 32;
    // big register tile (8x8x8 bricks) where the grid has room for it and there is enough K to amortise
    // its longer prologue/epilogue; the env switch is for A/B measurements
    static const int force_mt = getenv("TDX_CONV3_MT") ? atoi(getenv("TDX_CONV3_MT")) : 0;
    int MT = 2;  // measured: MT = 4 (one workgroup/CU) is 5-30 % slower on every layer of the U-Net
    if (force_mt == 2 || force_mt == 4) MT = force_mt;
    // ping-pong LDS variant (one workgroup per CU); TDX_CONV3_PP=0/1 overrides for A/B runs
    static const int force_pp = getenv("TDX_CONV3_PP") ? atoi(getenv("TDX_CONV3_PP")) : -1;
    const bool PPv = force_pp >= 0 ? (force_pp != 0) : M3_DEFAULT_PP;
    const int BX = 2 * MT;
    const int nbx = ceil_div(g.Xo, BX), nby = ceil_div(g.Yo, M3_BY), nbz = ceil_div(g.Zo, M3_BZ);
    if ((int64_t)g.Xi * g.Yi * g.Zi * 2 >= (1ll << 31)) return TDX_ESHAPE;  // a_src packs (voxel, half) in 31 bits
    dim3 grid((unsigned)((int64_t)g.B * nbx * nby * nbz), Cout / BN);
    const size_t stage = (size_t)2 * ((BX + 2) * M3_HY * M3_SZ * 16 + 64) + (size_t)27 * BN * 32 + 128;
    const size_t lds = PPv ? 2 * stage : stage;
#define M3_LAUNCH_PP(NTV, MTV, ZP, PPV)                                                                              \
    do {                                                                                                             \
        auto kern = conv3_mfma_kernel<NTV, MTV, ZP, PPV>;                                                            \
        static bool attr_set = false;                                                                                \
        if (!attr_set) {                                                                                             \
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return (int)e;                                                                      \
            attr_set = true;                                                                                         \
        }                                                                                                            \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, (const bf16*)x1, C1, (const bf16*)x2, C2,                 \
                           (const bf16*)wp, bias, (bf16*)y, g, Cout, nbx, nby, nbz, gn_acc, (bf16*)d1, D1,           \
                           (bf16*)d2, (const bf16*)a1, (const bf16*)a2);                                             \
    } while (0)
#define M3_LAUNCH(NTV, MTV, ZP)                                     \
    do {                                                            \
        if (PPv) M3_LAUNCH_PP(NTV, MTV, ZP, true);                  \
        else M3_LAUNCH_PP(NTV, MTV, ZP, false);                     \
    } while (0)
    if (NT == 2 && MT == 4) { if (zero_pad) M3_LAUNCH(2, 4, true); else M3_LAUNCH(2, 4, false); }
    else if (NT == 2)       { if (zero_pad) M3_LAUNCH(2, 2, true); else M3_LAUNCH(2, 2, false); }
    else if (MT == 4)       { if (zero_pad) M3_LAUNCH(1, 4, true); else M3_LAUNCH(1, 4, false); }
    else                    { if (zero_pad) M3_LAUNCH(1, 2, true); else M3_LAUNCH(1, 2, false); }
#undef M3_LAUNCH
#undef M3_LAUNCH_PP
    return tdx_launch_status();
}
