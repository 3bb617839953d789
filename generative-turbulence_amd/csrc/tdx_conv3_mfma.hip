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
// workgroup -> two workgroups per CU, one staging while the other issues MFMAs.
//
// Wave w owns the x = w slab of the brick: 8 x 8 voxels = two 32-row M tiles
// (row r <-> y = 4*mt + (r & 3), z = r >> 2) x NT 32-column N tiles, i.e. 2*NT
// v_mfma_f32_32x32x16_bf16 per tap per slice, A and B fragments by ds_read_b128.
//
// LDS images (both conflict-free for ds_read_b128, checked by exhaustive enumeration of the
// 16-lane read groups over all tap offsets):
//   brick  : voxel h = (hx*10 + hy)*12 + hz (z stride padded 10 -> 12), 32 B per voxel,
//            the two 16-B halves swapped when (h >> 3) & 1
//   weights: row (tap*BN + n), 32 B per row, halves swapped when (n >> 3) & 1
#include "tdx_common.h"
#include "tdx_conv3.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define M3_BX 4
#define M3_BY 8
#define M3_BZ 8
#define M3_HX (M3_BX + 2)
#define M3_HY (M3_BY + 2)
#define M3_HZ (M3_BZ + 2)
#define M3_SZ 12                               // padded z stride of the LDS brick
#define M3_NVOX_HALO (M3_HX * M3_HY * M3_HZ)   // 600
#define M3_BRICK_BYTES (M3_HX * M3_HY * M3_SZ * 32)  // 23040
#define M3_KC 16

bool conv3_mfma_supported(int C1, int C2, int Cout) {
    return C1 > 0 && (C1 % M3_KC) == 0 && (C2 % M3_KC) == 0 && (Cout % 32) == 0;
}

__device__ __forceinline__ int brick_addr(int h, int half) { return h * 32 + ((half ^ ((h >> 3) & 1)) << 4); }

template <int NT, bool ZERO_PAD>
__global__ void __launch_bounds__(256, 2)
conv3_mfma_kernel(const bf16* __restrict__ x1, int C1, const bf16* __restrict__ x2, int C2,
                  const bf16* __restrict__ wp, const float* __restrict__ bias, bf16* __restrict__ y, Conv3Geom g,
                  int Cout, int nbx, int nby, int nbz) {
    constexpr int BN = NT * 32;
    constexpr int W_BYTES = 27 * BN * 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sA = smem;
    unsigned char* sB = smem + M3_BRICK_BYTES;

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
    const int ox0 = bx * M3_BX, oy0 = by * M3_BY, oz0 = bz * M3_BZ;
    const int Cin = C1 + C2;

    // ---- staging plan for the input brick: 1200 16-B pieces, <= 5 per thread
    constexpr int A_PIECES = M3_NVOX_HALO * 2;
    constexpr int A_PER_THREAD = (A_PIECES + 255) / 256;  // 5
    int64_t a_src[A_PER_THREAD];  // voxel index in the input grid, -1 = zero fill
    int a_dst[A_PER_THREAD];      // LDS byte offset, -1 = no piece
#pragma unroll
    for (int i = 0; i < A_PER_THREAD; ++i) {
        const int p = tid + i * 256;
        a_dst[i] = -1;
        a_src[i] = -1;
        if (p < A_PIECES) {
            const int hv = p >> 1, half = p & 1;
            const int hx = hv / (M3_HY * M3_HZ), rem = hv - hx * (M3_HY * M3_HZ);
            const int hy = rem / M3_HZ, hz = rem - hy * M3_HZ;
            const int h = (hx * M3_HY + hy) * M3_SZ + hz;
            a_dst[i] = brick_addr(h, half);
            int sx = ox0 + hx - 1 + g.off, sy = oy0 + hy - 1 + g.off, sz = oz0 + hz - 1 + g.off;
            bool ok = true;
            if (ZERO_PAD) {
                ok = sx >= 0 && sx < g.Xi && sy >= 0 && sy < g.Yi && sz >= 0 && sz < g.Zi;
            } else {
                sx = min(max(sx, 0), g.Xi - 1); sy = min(max(sy, 0), g.Yi - 1); sz = min(max(sz, 0), g.Zi - 1);
            }
            if (ok) a_src[i] = ((((int64_t)b * g.Xi + sx) * g.Yi + sy) * g.Zi + sz) * 2 + half;  // (voxel, half)
        }
    }

    // ---- per-lane fragment bases
    // A: voxel (x = wave, y = 4*mt + (r&3), z = r>>2) at tap (0,0,0) sits at halo coords +1
    int a_h[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
        a_h[mt] = ((wave + 1) * M3_HY + (4 * mt + (r & 3) + 1)) * M3_SZ + ((r >> 2) + 1);
    // B: row n = nt*32 + r
    int b_off[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = nt * 32 + r;
        b_off[nt] = n * 32 + ((hh ^ ((n >> 3) & 1)) << 4);
    }

    f32x16 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;

    const int nchunks = Cin / M3_KC;
    for (int c = 0; c < nchunks; ++c) {
        // ---------------- stage: global -> registers -> LDS
        const int k0 = c * M3_KC;
        const bf16* xs;
        int Cs, kk;
        if (k0 < C1) { xs = x1; Cs = C1; kk = k0; } else { xs = x2; Cs = C2; kk = k0 - C1; }
        uint4 areg[A_PER_THREAD];
#pragma unroll
        for (int i = 0; i < A_PER_THREAD; ++i) {
            areg[i] = make_uint4(0, 0, 0, 0);
            if (a_src[i] >= 0) {
                const int64_t vox = a_src[i] >> 1;
                const int half = (int)(a_src[i] & 1);
                areg[i] = *reinterpret_cast<const uint4*>(xs + vox * Cs + kk + half * 8);
            }
        }
        constexpr int B_PIECES = 27 * BN * 2;
        constexpr int B_PER_THREAD = (B_PIECES + 255) / 256;
        uint4 breg[B_PER_THREAD];
        const bf16* wc = wp + (int64_t)c * 27 * Cout * 16;
#pragma unroll
        for (int i = 0; i < B_PER_THREAD; ++i) {
            const int p = tid + i * 256;
            breg[i] = make_uint4(0, 0, 0, 0);
            if (p < B_PIECES) {
                const int row = p >> 1, half = p & 1;  // row = tap*BN + n
                const int tap = row / BN, n = row - tap * BN;
                breg[i] = *reinterpret_cast<const uint4*>(wc + ((int64_t)tap * Cout + n0 + n) * 16 + half * 8);
            }
        }
        __syncthreads();  // previous slice's fragment reads are done
#pragma unroll
        for (int i = 0; i < A_PER_THREAD; ++i)
            if (a_dst[i] >= 0) *reinterpret_cast<uint4*>(sA + a_dst[i]) = areg[i];
#pragma unroll
        for (int i = 0; i < B_PER_THREAD; ++i) {
            const int p = tid + i * 256;
            if (p < B_PIECES) {
                const int row = p >> 1, half = p & 1;
                const int n = row % BN;
                *reinterpret_cast<uint4*>(sB + row * 32 + ((half ^ ((n >> 3) & 1)) << 4)) = breg[i];
            }
        }
        __syncthreads();

        // ---------------- compute: 27 taps x (2 x NT) MFMAs
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) {
            const int ex = tap / 9 - 1, ey = (tap / 3) % 3 - 1, ez = tap % 3 - 1;
            const int toff = (ex * M3_HY + ey) * M3_SZ + ez;
            bf16x8 af[2], bfr[NT];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                af[mt] = *reinterpret_cast<const bf16x8*>(sA + brick_addr(a_h[mt] + toff, hh));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                bfr[nt] = *reinterpret_cast<const bf16x8*>(sB + tap * (BN * 32) + b_off[nt]);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt], bfr[nt], acc[mt][nt], 0, 0, 0);
        }
    }

    // ---------------- epilogue: D[row = voxel][col = channel]; lane holds col r, rows
    // (i & 3) + 8 (i >> 2) + 4 hh  ->  y = 4 mt + (i & 3), z = 2 (i >> 2) + hh
    const int ox = ox0 + wave;
    if (ox < g.Xo) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = n0 + nt * 32 + r;
            const float bv = bias ? bias[n] : 0.f;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int oy = oy0 + 4 * mt + (i & 3);
                    const int oz = oz0 + 2 * (i >> 2) + hh;
                    if (oy < g.Yo && oz < g.Zo)
                        y[((((int64_t)b * g.Xo + ox) * g.Yo + oy) * g.Zo + oz) * Cout + n] =
                            __float2bfloat16(acc[mt][nt][i] + bv);
                }
        }
    }
}

int conv3_mfma_launch(const void* x1, int C1, const void* x2, int C2, const void* wp, const float* bias, void* y,
                      const Conv3Geom& g, int Cout, bool zero_pad, hipStream_t st) {
    const int nbx = ceil_div(g.Xo, M3_BX), nby = ceil_div(g.Yo, M3_BY), nbz = ceil_div(g.Zo, M3_BZ);
    const int NT = (Cout % 64 == 0) ? 2 : 1;
    const int BN = NT * 32;
    dim3 grid((unsigned)((int64_t)g.B * nbx * nby * nbz), Cout / BN);
    const size_t lds = M3_BRICK_BYTES + (size_t)27 * BN * 32;
#define M3_LAUNCH(NTV, ZP)                                                                                           \
    do {                                                                                                             \
        auto kern = conv3_mfma_kernel<NTV, ZP>;                                                                      \
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return (int)e;                                                                          \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, (const bf16*)x1, C1, (const bf16*)x2, C2,                 \
                           (const bf16*)wp, bias, (bf16*)y, g, Cout, nbx, nby, nbz);                                 \
    } while (0)
    if (NT == 2) { if (zero_pad) M3_LAUNCH(2, true); else M3_LAUNCH(2, false); }
    else         { if (zero_pad) M3_LAUNCH(1, true); else M3_LAUNCH(1, false); }
#undef M3_LAUNCH
    return tdx_launch_status();
}
