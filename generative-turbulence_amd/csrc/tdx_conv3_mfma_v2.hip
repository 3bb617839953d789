// bf16 MFMA implicit-GEMM 3x3x3 convolution, wide-tile variant (gfx950): forward and, with zero padding, the main term
// of the data gradient, for layers with 64-wide output tiles on grids that a 4 x 8 x 16 brick tiles well.
//
// Why a second kernel.  In tdx_conv3_mfma.hip a wave owns a 64-voxel x 64-channel register tile: every
// v_mfma_f32_32x32x16_bf16 needs one fresh ds_read_b128 per wave, i.e. 4 KB of LDS reads per 32 cycles per CU = the
// LDS peak itself at the MFMA peak, and its workgroups re-stream the 27-tap weight slice (54 KB) per 256 voxels.  That
// structure measured 0.8-1.0 PFLOP/s whatever else was tuned.  Here a wave owns 128 voxels x 64 channels (four M tiles x
// two N tiles): 6 fragment reads feed 8 MFMAs (0.75 per MFMA), the brick is 4 x 8 x 16 = 512 voxels (halo overhead 2.1x
// instead of 2.3x, weights re-streamed once per 512 voxels), and two workgroups still fit a CU because the slice's
// weights are staged in three 9-tap groups (18 KB each, double-buffered) instead of all 27 taps at once.  The weight
// groups are copied global -> LDS by the DMA path (global_load_lds_dwordx4: no staging registers, which is what makes
// 128 accumulator + 48 fragment registers fit in 256); the input brick stays register-staged (its LDS image is padded
// and zero-filled / clamped per voxel).
//
// Per K slice (16 input channels), workgroup = 4 waves, wave w = brick plane x = w:
//     group 0: DMA weights of group 1 -> buffer q^1; issue the next slice's brick loads -> registers;
//              9 taps x 8 MFMAs from buffer q;                                               barrier
//     group 1: DMA group 2 -> q;      9 taps x 8 MFMAs from q^1;                             barrier
//     group 2: DMA next slice's group 0 -> q^1;  9 taps x 8 MFMAs from q;                    barrier
//              brick registers -> LDS;                                                        barrier
// (q flips every slice: three groups.)  A DMA targets the buffer that was last read one group earlier, behind a
// barrier; __syncthreads() drains the DMA (vmcnt(0)), which by then has had 72 MFMAs (> 2000 cycles) to land.
//
// LDS: brick [2 halves][6 x 10 x 20 entries of 16 B] (z stride padded 18 -> 20: the 16 voxels of every ds_read_b128
// lane group -- 4 y x 4 z -- are distinct mod 16, conflict-free with affine tap offsets) = 38.5 KB; weights
// [2 buffers][2 halves][9 taps x 64 rows of 16 B] = 36.9 KB; 75.4 KB per workgroup.  Epilogue as the narrow kernel's:
// accumulators -> bf16 output tile in LDS (512 voxels x 64 ch = 64 KB) -> 16-B-per-lane voxel-row stores, with the
// GroupNorm moments (forward) or the dx split / residual addend (data gradient) in the store loop.
#include "tdx_common.h"
#include "tdx_conv3.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#ifndef V2_EXP
#define V2_EXP 0
#endif
#define V2_KC 16
#define V2_BN 64
#define V2_BX 4
#define V2_BY 8
#define V2_BZ 16
#define V2_HX (V2_BX + 2)
#define V2_HY (V2_BY + 2)
#define V2_HZ (V2_BZ + 2)
#define V2_SZ 20
#define V2_NVOX (V2_BX * V2_BY * V2_BZ)

struct V2View {
    int B;
    int Ei[3];  // grid extents along the local axes (input == output grid)
    int st[3];  // voxel strides of the local axes
    int nb[3];  // bricks per local axis
    int ws[3];  // weight-tap strides of the local axes ({9, 3, 1} permuted)
};

__device__ __forceinline__ int v2_out_addr(int v, int c) { return v * 128 + ((c ^ (v & 7)) << 4); }

template <bool ZERO_PAD, bool PERM>
__global__ void __launch_bounds__(256, 2)
conv3_mfma_v2_kernel(const bf16* __restrict__ x1, int C1, const bf16* __restrict__ x2, int C2, const bf16* __restrict__ wp,
                     const float* __restrict__ bias, bf16* __restrict__ y, V2View g, int Cout, double* __restrict__ gn_acc,
                     bf16* __restrict__ d1, int D1, bf16* __restrict__ d2, const bf16* __restrict__ a1,
                     const bf16* __restrict__ a2, const bf16* __restrict__ zero16) {
    constexpr int BN = V2_BN, HY = V2_HY, HZ = V2_HZ, SZ = V2_SZ;
    constexpr int NHALO = V2_HX * V2_HY * V2_HZ;            // 1080 staged voxels
    constexpr int APLANE = V2_HX * V2_HY * SZ * 16 + 64;    // one half-plane of the brick
    constexpr int BRICK_BYTES = 2 * APLANE;
    constexpr int GROWS = 9 * BN;                            // rows of one tap group
    constexpr int B_HALF = GROWS * 16;                       // 9216 B = nine 1-KiB DMA pieces
    constexpr int B_BUF = 2 * B_HALF;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sA = smem;
    unsigned char* sB = smem + BRICK_BYTES;                  // [2 buffers][2 halves][GROWS]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;

    int bid = xcd_contiguous((int)blockIdx.x, (int)gridDim.x);
    const int b2 = bid % g.nb[2]; bid /= g.nb[2];
    const int b1 = bid % g.nb[1]; bid /= g.nb[1];
    const int b0 = bid % g.nb[0]; bid /= g.nb[0];
    const int b = bid;
    const int n0 = blockIdx.y * BN;
    const int o0 = b0 * V2_BX, o1 = b1 * V2_BY, o2 = b2 * V2_BZ;
    const int Cin = C1 + C2;

    // ---- staging of the input brick: 2 * NHALO 16-B pieces, 8-9 per thread.  The piece -> (LDS slot, source voxel) map
    // is recomputed where it is used (a few integer ops per piece and slice) instead of being held in 18 registers:
    // the accumulators and fragment sets leave no room for it.
    constexpr int A_PIECES = NHALO * 2;
    constexpr int A_PER_THREAD = (A_PIECES + 255) / 256;
    const int64_t batch_vox = (int64_t)b * g.Ei[0] * g.Ei[1] * g.Ei[2];
    auto piece = [&](int i, int& dst, int& src) {  // dst: LDS byte offset; src: (voxel * 2 + half) or -1 (zero fill)
        const int p = tid + i * 256;
        const int hv = ((p >> 3) << 2) + (p & 3), half = (p >> 2) & 1;  // 4 consecutive voxels x the two halves per 8 lanes
        const int hx = hv / (HY * HZ), rem = hv - hx * (HY * HZ);
        const int hy = rem / HZ, hz = rem - hy * HZ;
        dst = half * APLANE + ((hx * HY + hy) * SZ + hz) * 16;
        int s0 = o0 + hx - 1, s1 = o1 + hy - 1, s2 = o2 + hz - 1;
        bool ok = p < A_PIECES;
        if (ZERO_PAD) {
            ok = ok && s0 >= 0 && s0 < g.Ei[0] && s1 >= 0 && s1 < g.Ei[1] && s2 >= 0 && s2 < g.Ei[2];
        } else {
            s0 = min(max(s0, 0), g.Ei[0] - 1); s1 = min(max(s1, 0), g.Ei[1] - 1); s2 = min(max(s2, 0), g.Ei[2] - 1);
        }
        src = ok ? (s0 * g.st[0] + s1 * g.st[1] + s2 * g.st[2]) * 2 + half : -1;
        if (p >= A_PIECES) dst = -1;
    };

    uint4 areg[A_PER_THREAD];
    auto load_brick = [&](int c) {
        const int k0 = c * V2_KC;
        const bf16* xs;
        int Cs, kk;
        if (k0 < C1) { xs = x1; Cs = C1; kk = k0; } else { xs = x2; Cs = C2; kk = k0 - C1; }
        xs += batch_vox * Cs + kk;
#pragma unroll
        for (int i = 0; i < A_PER_THREAD; ++i) {
            int dst, src;
            piece(i, dst, src);
            areg[i] = make_uint4(0, 0, 0, 0);
            if (src >= 0) areg[i] = *reinterpret_cast<const uint4*>(xs + (int64_t)(src >> 1) * Cs + (src & 1) * 8);
        }
    };
    auto store_brick = [&]() {
#pragma unroll
        for (int i = 0; i < A_PER_THREAD; ++i) {
            int dst, src;
            piece(i, dst, src);
            if (dst >= 0) *reinterpret_cast<uint4*>(sA + dst) = areg[i];
        }
    };

    // ---- weight groups by LDS-DMA.  Group (slice c, local x tap ex = grp - 1) = 9 taps x BN rows x 2 halves = 18 pieces
    // of 1 KiB; wave w copies pieces w, w + 4, ...; lane l of piece i fills [half = i / 9][row = (i % 9) * 64 + l].
    // The global row of (local tap, n) sits at ((c * 27 + tap_global) * Cout + n0 + n) * 16 + half * 8 elements.
    auto dma_group = [&](int c, int grp, int buf) {
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int i = wave + 4 * j;  // piece
            if (i < 18) {
                const int half = i / 9, t9 = i % 9;          // one piece = one tap's 64 rows of one half
                const int ey = t9 / 3, ez = t9 % 3;
                const int tapg = PERM ? grp * g.ws[0] + ey * g.ws[1] + ez * g.ws[2] : grp * 9 + t9;
                const bf16* src = wp + ((int64_t)(c * 27 + tapg) * Cout + n0 + lane) * V2_KC + half * 8;
                unsigned char* dst = sB + buf * B_BUF + half * B_HALF + t9 * (BN * 16);  // wave-uniform; lane l lands at + 16 l
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            }
        }
    };

    // ---- per-lane fragment bases.  Wave w = brick plane x = w; M tile mt: y = 4 (mt & 1) + (r & 3), z = 8 (mt >> 1) + (r >> 2)
    int a_h[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
        a_h[mt] = hh * APLANE + (((wave + 1) * HY + (4 * (mt & 1) + (r & 3) + 1)) * SZ + (8 * (mt >> 1) + (r >> 2) + 1)) * 16;
    const int b_off = hh * B_HALF + r * 16;

    f32x16 acc[2][4];  // D[row = channel][col = voxel]
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[nt][mt][i] = 0.f;

    // One tap group: 9 taps x 8 MFMAs.  MFMAs go M-tile-major (both N tiles of an M tile back to back), so an x
    // fragment is dead after its pair and the next tap's x fragment of the same M tile is read into its place right
    // behind it: 4 + 1 x fragments and 2 + 2 w fragments are live instead of two full sets (the register file holds
    // 128 accumulators, 36 brick-staging registers and these).  sched_group_barrier pins that interleave.
    auto compute_group = [&](int grp, int buf) {
        const int ex = grp - 1;
        bf16x8 x[4], w[2];
        auto x_addr = [&](int t9, int mt) { return sA + a_h[mt] + ((ex * HY + (t9 / 3 - 1)) * SZ + (t9 % 3 - 1)) * 16; };
        auto w_addr = [&](int t9, int nt) { return sB + buf * B_BUF + b_off + (t9 * BN + nt * 32) * 16; };
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) w[nt] = *reinterpret_cast<const bf16x8*>(w_addr(0, nt));
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) x[mt] = *reinterpret_cast<const bf16x8*>(x_addr(0, mt));
        __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
        for (int t9 = 0; t9 < 9; ++t9) {
            const bool nxt = t9 + 1 < 9;
            bf16x8 wn[2];
            if (nxt && (V2_EXP != 2)) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) wn[nt] = *reinterpret_cast<const bf16x8*>(w_addr(t9 + 1, nt));
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                acc[0][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[mt], acc[0][mt], 0, 0, 0);
                acc[1][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], x[mt], acc[1][mt], 0, 0, 0);
#if V2_EXP == 1  /* experiment: twice the MFMAs, everything else unchanged */
                acc[0][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[mt], acc[0][mt], 0, 0, 0);
                acc[1][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], x[mt], acc[1][mt], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
#endif
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                if (nxt && (V2_EXP != 2)) {  /* experiment 2: no fragment reads after tap 0 */
                    x[mt] = *reinterpret_cast<const bf16x8*>(x_addr(t9 + 1, mt));
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
            if (nxt && (V2_EXP != 2)) { w[0] = wn[0]; w[1] = wn[1]; }
        }
    };

    // barrier that also retires this wave's outstanding LDS-DMA (and brick loads): a DMA is ordered for other waves'
    // ds_reads only by the issuing wave's vmcnt wait followed by a barrier
    auto drain_and_sync = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    const int nchunks = Cin / V2_KC;
    // prologue: brick 0 and weight group (0, 0)
    load_brick(0);
    dma_group(0, 0, 0);
    store_brick();
    drain_and_sync();
    int q = 0;  // buffer that holds the current slice's group 0
    for (int c = 0; c < nchunks; ++c) {
        const bool more = c + 1 < nchunks;
        if (V2_EXP != 4) dma_group(c, 1, q ^ 1);
        if (more && V2_EXP != 5) load_brick(c + 1);
        compute_group(0, q);
        drain_and_sync();
        if (V2_EXP != 4) dma_group(c, 2, q);
        compute_group(1, q ^ 1);
        drain_and_sync();
        if (more && V2_EXP != 4) dma_group(c + 1, 0, q ^ 1);
        compute_group(2, q);
        drain_and_sync();
        if (more) {
            store_brick();
            __syncthreads();
        }
        q ^= 1;
    }

    // ---------------- epilogue.  Lane (r, hh) of wave w holds, for M tile mt, voxel (w, 4 (mt & 1) + (r & 3), 8 (mt >> 1) + (r >> 2))
    // and channels nt*32 + 8 j + 4 hh + (0..3) in accumulator registers 4 j .. 4 j + 3.  Tile voxel index v = (x*8 + y)*16 + z.
#if V2_EXP == 3
    {
        float keep = 0.f;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int i = 0; i < 16; ++i) keep += acc[nt][mt][i];
        if (keep == 12345.678f) y[0] = __float2bfloat16(keep);
        return;
    }
#endif
    unsigned char* sO = smem;  // [512 voxels][64] bf16 = 64 KB
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ch = nt * 32 + 8 * j + 4 * hh;
            float bv[4] = {0.f, 0.f, 0.f, 0.f};
            if (bias) {
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[e] = bias[n0 + ch + e];
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int v = (wave * V2_BY + 4 * (mt & 1) + (r & 3)) * V2_BZ + 8 * (mt >> 1) + (r >> 2);
                const unsigned lo = pack_bf16x2(acc[nt][mt][4 * j] + bv[0], acc[nt][mt][4 * j + 1] + bv[1]);
                const unsigned hi = pack_bf16x2(acc[nt][mt][4 * j + 2] + bv[2], acc[nt][mt][4 * j + 3] + bv[3]);
                *reinterpret_cast<uint2*>(sO + v2_out_addr(v, ch >> 3) + (ch & 7) * 2) = make_uint2(lo, hi);
            }
        }
    __syncthreads();
    constexpr int CHUNKS = BN / 8;
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
#pragma unroll
    for (int i = 0; i < CHUNKS * (V2_NVOX / 256); ++i) {
        const int p = tid + i * 256;
        const int v = p / CHUNKS, cidx = p % CHUNKS;
        const int c0 = o0 + v / (V2_BY * V2_BZ), c1 = o1 + (v / V2_BZ) % V2_BY, c2 = o2 + v % V2_BZ;
        if (c0 < g.Ei[0] && c1 < g.Ei[1] && c2 < g.Ei[2]) {
            uint4 val = *reinterpret_cast<const uint4*>(sO + v2_out_addr(v, cidx));
            const int64_t u = batch_vox + (int64_t)c0 * g.st[0] + (int64_t)c1 * g.st[1] + (int64_t)c2 * g.st[2];
            if (ZERO_PAD && d1 != nullptr) {
                // data gradient: straight to dx, split over the two inputs of a concatenated conv, plus the fused addend
                const int n = n0 + cidx * 8;
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
            } else {
                *reinterpret_cast<uint4*>(y + u * Cout + n0 + cidx * 8) = val;
            }
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
        constexpr int NP = 256 / CHUNKS;
        __syncthreads();  // the moment table reuses the output tile's LDS
        float* red = reinterpret_cast<float*>(smem);  // [NP][BN][2] = 16 KB
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
            const int rep = blockIdx.x & (TDX_GN_REPLICAS - 1);
            atomicAdd(&gn_acc[(((size_t)rep * g.B + b) * Cout + n0) * 2 + tid], (double)t);
        }
    }
    (void)zero16;
}

// does the wide-tile kernel take this launch?  64-wide output tiles, both inputs sliceable, and a grid that the
// 4 x 8 x 16 brick (in its best orientation) fills at least 90 %, with enough bricks to load the chip
bool conv3_mfma_v2_applies(const Conv3Geom& g, int C1, int C2, int Cout, int perm_out[3]) {
    const char* env = getenv("TDX_CONV3_V2");  // 0: off, 1 (default): where it pays, 2: wherever the shapes allow (tests)
    const int mode = env ? atoi(env) : 1;
    if (mode == 0) return false;
    if (C1 <= 0 || (C1 % V2_KC) || (C2 % V2_KC) || (Cout % V2_BN)) return false;
    if (g.Xi != g.Xo || g.Yi != g.Yo || g.Zi != g.Zo || g.off != 0) return false;
    const int E[3] = {g.Xo, g.Yo, g.Zo};
    static const int perms[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
    const int bd[3] = {V2_BX, V2_BY, V2_BZ};
    int64_t best = -1;
    for (int c = 0; c < 6; ++c) {
        int64_t n = 1;
        for (int k = 0; k < 3; ++k) n *= ceil_div(E[perms[c][k]], bd[k]);
        if (best < 0 || n < best) {
            best = n;
            for (int k = 0; k < 3; ++k) perm_out[k] = perms[c][k];
        }
    }
    const int64_t vox = (int64_t)E[0] * E[1] * E[2];
    if (mode == 2) return true;  // forced (tests)
    return best * V2_NVOX * 9 <= vox * 10 && (int64_t)g.B * best * (Cout / V2_BN) >= 512;
}

int conv3_mfma_v2_launch(const void* x1, int C1, const void* x2, int C2, const void* wp, const float* bias, void* y,
                         const Conv3Geom& g, int Cout, bool zero_pad, const int perm[3], hipStream_t st, double* gn_acc, void* d1,
                         int D1, void* d2, const void* a1, const void* a2) {
    if ((int64_t)g.Xi * g.Yi * g.Zi * 2 >= (1ll << 31)) return TDX_ESHAPE;
    const int E[3] = {g.Xo, g.Yo, g.Zo}, str[3] = {g.Yo * g.Zo, g.Zo, 1}, tapw[3] = {9, 3, 1};
    const int bd[3] = {V2_BX, V2_BY, V2_BZ};
    V2View v;
    v.B = g.B;
    int64_t n = g.B;
    for (int k = 0; k < 3; ++k) {
        v.Ei[k] = E[perm[k]]; v.st[k] = str[perm[k]]; v.ws[k] = tapw[perm[k]];
        v.nb[k] = ceil_div(v.Ei[k], bd[k]);
        n *= v.nb[k];
    }
    const bool permuted = !(perm[0] == 0 && perm[1] == 1 && perm[2] == 2);
    const size_t brick = (size_t)2 * (V2_HX * V2_HY * V2_SZ * 16 + 64), wts = (size_t)2 * 2 * 9 * V2_BN * 16;
    size_t lds = brick + wts;
    if (lds < (size_t)V2_NVOX * V2_BN * 2) lds = (size_t)V2_NVOX * V2_BN * 2;
#define V2_GO(ZP, PM)                                                                                                    \
    do {                                                                                                                 \
        auto kern = conv3_mfma_v2_kernel<ZP, PM>;                                                                        \
        static bool attr_set = false;                                                                                    \
        if (!attr_set) {                                                                                                 \
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return (int)e;                                                                          \
            attr_set = true;                                                                                             \
        }                                                                                                                \
        hipLaunchKernelGGL(kern, dim3((unsigned)n, Cout / V2_BN), dim3(256), lds, st, (const bf16*)x1, C1, (const bf16*)x2, C2, \
                           (const bf16*)wp, bias, (bf16*)y, v, Cout, gn_acc, (bf16*)d1, D1, (bf16*)d2, (const bf16*)a1,   \
                           (const bf16*)a2, (const bf16*)nullptr);                                                       \
    } while (0)
    if (zero_pad) { if (permuted) V2_GO(true, true); else V2_GO(true, false); }
    else { if (permuted) V2_GO(false, true); else V2_GO(false, false); }
#undef V2_GO
    return tdx_launch_status();
}
