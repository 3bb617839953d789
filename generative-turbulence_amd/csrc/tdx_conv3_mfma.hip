// bf16 MFMA implicit-GEMM 3x3x3 convolution for gfx950 (forward and, with zero padding, the main
// term of the data gradient; its halo-shell term is tdx_conv3_shell.hip).
//
//   M = output voxels (a 256-voxel brick per workgroup), N = output channels (BN = 32/64 per
//   workgroup), K = 27 taps x input channels, walked in 16-channel slices.
//
// Per K slice the workgroup stages ONE halo'd input brick (6 x 10 x 10 voxels x 16 ch = 19 KB)
// and the slice's weights for all 27 taps (27 x BN x 16 = 54 KB at BN = 64) into LDS; every tap
// then re-reads the same brick at a shifted voxel offset, so the 27-fold input reuse of the
// convolution is served from LDS, not from L2/HBM.  78 KB of LDS per workgroup -> two
// workgroups per CU, one staging while the other issues MFMAs.
//
// Pipeline: the global loads of slice c+1 are issued into registers right after slice c has been
// written to LDS and stay in flight during the 27 x 2 x NT MFMAs of slice c (the wait lands at
// the next LDS write); inside a slice the fragments of tap t+1 are read while the MFMAs of tap t
// issue (pinned with sched_group_barrier).
//
// Wave w owns an 8 x 8 (y, z) slab of the brick = two 32-voxel M tiles (r <-> y = 4*mt + (r & 3),
// z = r >> 2) x NT 32-channel N tiles.  The MFMA is issued as D^T = W^T X^T (weights as the A
// operand), so a lane owns ONE voxel and 4 consecutive channels per accumulator quad: the
// epilogue packs them to 8-B writes of an LDS output tile [256 voxels][BN] that is then stored to
// HBM in whole 16-B-per-lane voxel rows (+ fused GroupNorm moments, + fused residual addend for
// the data gradient).
//
// LDS images, each split in two half-planes (channels 0-7 / 8-15 of the slice) of 16-B entries:
//   brick  : [half][voxel h = (hx*HY + hy)*SZ + hz]; with the main shape's z stride padded
//            10 -> 12 the 16 voxels of every ds_read_b128 lane group are distinct mod 16, i.e.
//            conflict-free with plain affine addresses (tap offsets are instruction immediates;
//            found by exhaustive enumeration of lane groups x tap offsets)
//   weights: [half][tap*BN + n]
//
// Brick shapes.  The main shape is 4 x 8 x 8 (wave = x plane).  Grids whose extent leaves a
// remainder of 1-2 voxels along an axis -- every padded data-gradient grid (X+2 ...), and the
// reference's real 194 x 50 x 50 grids -- would waste a whole row of mostly empty bricks per such
// axis (+34 % bricks at 194 x 66 x 50), so those remainder slabs are tiled with a thin
// 2 x 16 x 8 brick (XT: wave = (x plane, y half)), with the kernel's local axes permuted so that
// its thin axis is the slab's thin axis.  All index arithmetic is therefore in "local axes" with
// explicit voxel strides.
//
// Measured alternatives that lost (kept out of the build): an 8 x 8 x 8 brick with a 128 x 64
// register tile per wave and one workgroup per CU (-5...-30 %), and a ping-pong-LDS single
// workgroup per CU with one barrier per slice (700 vs 840 TFLOP/s): with this much LDS traffic
// per MFMA, two co-resident workgroups hide more than a deeper pipeline in one.
#include "tdx_common.h"
#include "tdx_conv3.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define M3_KC 16

bool conv3_mfma_supported(int C1, int C2, int Cout) {
    return C1 > 0 && (C1 % M3_KC) == 0 && (C2 % M3_KC) == 0 && (Cout % 32) == 0;
}

// geometry of one launch, in the kernel's local axes (axis 0 = brick "x", axis 2 = brick "z")
struct ConvView {
    int B;
    int Ei[3];   // input grid extents
    int si[3];   // input voxel strides
    int so[3];   // output voxel strides
    int r0[3];   // first output coordinate of the region handled by this launch
    int r1[3];   // end (exclusive)
    int nb[3];   // bricks per axis
    int off;     // output voxel o reads input voxel o + off + e
    int ws[3];   // weight-tap strides of the local axes: tap index = sum_k (e_k + 1) * ws[k]  ({9,3,1} permuted)
    int in_batch, out_batch;  // voxels per sample
    int ld1, ld2;             // row strides (elements per voxel) of the two inputs (>= C1 / C2)
    int init_batch;           // voxels between samples of the accumulator-init tensor (0: shared by the batch)
};

// up to three regions handled by ONE launch (the three remainder slabs of a grid): blocks
// [start[r], start[r + 1]) belong to region r
struct ConvViews {
    ConvView v[3];
    int start[4];
};

// output tile rows of 64 B (BN = 32) or 128 B (BN = 64); 16-B chunk c of row v at c ^ swizzle(v)
template <int BN>
__device__ __forceinline__ int out_addr(int v, int c) {
    if (BN == 64) return v * 128 + ((c ^ (v & 7)) << 4);
    return v * 64 + ((c ^ ((v >> 1) & 3)) << 4);
}

// EXT: strided inputs (ConvView::ld1/ld2) and the `init` tensor of tdx_conv3_fwd_partial; a separate
// instantiation so that the hot kernels carry none of it (it cost the 32-channel layers 7-19 %)
// HF: the operand format (H16<HF>: bf16 or fp16 words behind the bf16-typed pointers)
template <int NT, bool XT, bool ZERO_PAD, bool PERM, bool EXT, bool HF>
__global__ void __launch_bounds__(256, 2)
conv3_mfma_kernel(const bf16* __restrict__ x1, int C1, const bf16* __restrict__ x2, int C2,
                  const bf16* __restrict__ wp, const float* __restrict__ bias, bf16* __restrict__ y, ConvViews gs,
                  int Cout, double* __restrict__ gn_acc, bf16* __restrict__ d1, int D1, bf16* __restrict__ d2,
                  const bf16* __restrict__ a1, const bf16* __restrict__ a2, const bf16* __restrict__ init) {
    typedef H16<HF> H;
    typedef typename H::T HT;
    constexpr int BN = NT * 32;
    constexpr int BX = XT ? 2 : 4, BY = XT ? 16 : 8, BZ = 8;
    constexpr int HX = BX + 2, HY = BY + 2, HZ = BZ + 2;
    constexpr int SZ = XT ? 10 : 12;                    // z stride of the LDS brick (padded for the main shape)
    constexpr int NHALO = HX * HY * HZ;                 // 600 / 720 staged voxels
    constexpr int APLANE = HX * HY * SZ * 16 + 64;      // +64 B: the two halves of a voxel land 4 slots apart
    constexpr int BRICK_BYTES = 2 * APLANE;
    constexpr int B_PLANE = 27 * BN * 16 + 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sA = smem;
    unsigned char* sB = smem + BRICK_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;

    // block -> region -> (n tile, b, brick)
    int bid = blockIdx.x;
    bid = xcd_contiguous(bid, (int)gridDim.x);  // neighbouring bricks share an XCD's L2 (halo reuse)
    const int region = bid >= gs.start[1] ? (bid >= gs.start[2] ? 2 : 1) : 0;
    const ConvView& g = gs.v[region];
    bid -= gs.start[region];
    const int b2 = bid % g.nb[2]; bid /= g.nb[2];
    const int b1 = bid % g.nb[1]; bid /= g.nb[1];
    const int b0 = bid % g.nb[0]; bid /= g.nb[0];
    const int b = bid;
    const int n0 = blockIdx.y * BN;
    const int o0 = g.r0[0] + b0 * BX, o1 = g.r0[1] + b1 * BY, o2 = g.r0[2] + b2 * BZ;
    const int Cin = C1 + C2;

    // ---- staging plan for the input brick: 2 * NHALO 16-B pieces, 5-6 per thread.
    // a_src: (voxel index in the sample's input grid) * 2 + half, or -1 for zero fill / no piece
    constexpr int A_PIECES = NHALO * 2;
    constexpr int A_PER_THREAD = (A_PIECES + 255) / 256;
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
            const int hx = hv / (HY * HZ), rem = hv - hx * (HY * HZ);
            const int hy = rem / HZ, hz = rem - hy * HZ;
            a_dst[i] = half * APLANE + ((hx * HY + hy) * SZ + hz) * 16;
            int s0 = o0 + hx - 1 + g.off, s1 = o1 + hy - 1 + g.off, s2 = o2 + hz - 1 + g.off;
            bool ok = true;
            if (ZERO_PAD) {
                ok = s0 >= 0 && s0 < g.Ei[0] && s1 >= 0 && s1 < g.Ei[1] && s2 >= 0 && s2 < g.Ei[2];
            } else {
                s0 = min(max(s0, 0), g.Ei[0] - 1); s1 = min(max(s1, 0), g.Ei[1] - 1); s2 = min(max(s2, 0), g.Ei[2] - 1);
            }
            if (ok) a_src[i] = (s0 * g.si[0] + s1 * g.si[1] + s2 * g.si[2]) * 2 + half;
        }
    }
    const int64_t batch_vox = (int64_t)b * g.in_batch;

    // weight staging role of this thread: (row = b_row0 + 128 i, half); 128 rows = 128/BN taps per
    // step, so both the global and the LDS address advance by a constant per i
    const int b_half = (tid >> 2) & 1;
    const int b_row0 = ((tid >> 3) << 2) + (tid & 3);                             // 0..127
    const int b_goff = ((b_row0 / BN) * Cout + (b_row0 % BN)) * 16 + b_half * 8;  // elements
    const int b_dst = b_half * B_PLANE + b_row0 * 16;

    uint4 areg[A_PER_THREAD], breg[B_PER_THREAD];
    auto load_slice = [&](int c) {
        const int k0 = c * M3_KC;
        const bf16* xs;
        int Cs, kk;
        if (k0 < C1) { xs = x1; Cs = EXT ? g.ld1 : C1; kk = k0; } else { xs = x2; Cs = EXT ? g.ld2 : C2; kk = k0 - C1; }
        xs += batch_vox * Cs + kk;
#pragma unroll
        for (int i = 0; i < A_PER_THREAD; ++i) {
            areg[i] = make_uint4(0, 0, 0, 0);
            if (a_src[i] >= 0)
                areg[i] = *reinterpret_cast<const uint4*>(xs + (int64_t)(a_src[i] >> 1) * Cs + (a_src[i] & 1) * 8);
        }
        const bf16* wc = wp + (int64_t)c * 27 * Cout * 16 + (int64_t)n0 * 16 + b_goff;
#pragma unroll
        for (int i = 0; i < B_PER_THREAD; ++i) {
            breg[i] = make_uint4(0, 0, 0, 0);
            if (b_row0 + 128 * i < 27 * BN)
                breg[i] = *reinterpret_cast<const uint4*>(wc + (int64_t)i * (128 / BN) * Cout * 16);
        }
    };
    auto store_slice = [&]() {
#pragma unroll
        for (int i = 0; i < A_PER_THREAD; ++i)
            if (a_dst[i] >= 0) *reinterpret_cast<uint4*>(sA + a_dst[i]) = areg[i];
#pragma unroll
        for (int i = 0; i < B_PER_THREAD; ++i)
            if (b_row0 + 128 * i < 27 * BN) *reinterpret_cast<uint4*>(sB + b_dst + i * 2048) = breg[i];
    };

    // ---- per-lane fragment bases.  Wave w owns brick plane wx and the y rows [wy0, wy0 + 8):
    // M tile mt: y = wy0 + 4*mt + (r & 3), z = r >> 2
    const int wx = XT ? (wave >> 1) : wave, wy0 = XT ? 8 * (wave & 1) : 0;
    int a_h[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
        a_h[mt] = ((wx + 1) * HY + (wy0 + 4 * mt + (r & 3) + 1)) * SZ + ((r >> 2) + 1);
    int b_off[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) b_off[nt] = hh * B_PLANE + (nt * 32 + r) * 16;

    f32x16 acc[NT][2];  // D[row = channel][col = voxel]
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[nt][mt][i] = 0.f;

    const int nchunks = Cin / M3_KC;
    load_slice(0);
    for (int c = 0; c < nchunks; ++c) {
        __syncthreads();  // previous slice's fragment reads are done
        store_slice();
        __syncthreads();
        if (c + 1 < nchunks) load_slice(c + 1);  // in flight during the MFMAs below

        // fragments of tap t+1 are read while the MFMAs of tap t issue (two register sets)
        bf16x8 xf[2][2], wf[2][NT];
        auto read_frags = [&](int tap, int buf) {
            const int ex = tap / 9 - 1, ey = (tap / 3) % 3 - 1, ez = tap % 3 - 1;
            const int toff = (ex * HY + ey) * SZ + ez;
            // the weight tap is indexed in GLOBAL axes; thin bricks run on permuted local axes
            const int wtap = (XT || PERM) ? (ex + 1) * g.ws[0] + (ey + 1) * g.ws[1] + (ez + 1) * g.ws[2] : tap;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                xf[buf][mt] = *reinterpret_cast<const bf16x8*>(sA + hh * APLANE + (a_h[mt] + toff) * 16);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                wf[buf][nt] = *reinterpret_cast<const bf16x8*>(sB + wtap * (BN * 16) + b_off[nt]);
        };
        read_frags(0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2 + NT, 0);  // DS_READ: tap 0's fragments
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) {
            if (tap + 1 < 27) read_frags(tap + 1, (tap + 1) & 1);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    acc[nt][mt] = H::mfma(wf[tap & 1][nt], xf[tap & 1][mt], acc[nt][mt]);
            // pin the interleave: one fragment read of tap+1 behind each MFMA of tap
            if (tap + 1 < 27) {
#pragma unroll
                for (int k = 0; k < 2 * NT; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
                    if (k < 2 + NT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS_READ
                }
            } else {
                __builtin_amdgcn_sched_group_barrier(0x008, 2 * NT, 0);
            }
        }
    }

    // ---------------- epilogue.  Lane (r, hh) of wave w holds, for M tile mt, voxel
    // (wx, wy0 + 4 mt + (r & 3), r >> 2) and channels nt*32 + 8 j + 4 hh + (0..3) in accumulator
    // registers 4 j .. 4 j + 3.  Tile voxel index v = (x*BY + y)*8 + z.
    __syncthreads();
    unsigned char* sO = smem;  // [256 voxels][BN] bf16
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
            for (int mt = 0; mt < 2; ++mt) {
                const int v = (wx * BY + wy0 + 4 * mt + (r & 3)) * 8 + (r >> 2);
                const unsigned lo = H::pack2(acc[nt][mt][4 * j] + bv[0], acc[nt][mt][4 * j + 1] + bv[1]);
                const unsigned hi = H::pack2(acc[nt][mt][4 * j + 2] + bv[2], acc[nt][mt][4 * j + 3] + bv[3]);
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
    for (int i = 0; i < CHUNKS; ++i) {
        const int p = tid + i * 256;
        const int v = p / CHUNKS, cidx = p % CHUNKS;
        const int c0 = o0 + (v >> 3) / BY, c1 = o1 + (v >> 3) % BY, c2 = o2 + (v & 7);
        if (c0 < g.r1[0] && c1 < g.r1[1] && c2 < g.r1[2]) {
            uint4 val = *reinterpret_cast<const uint4*>(sO + out_addr<BN>(v, cidx));
            bool direct = false;
            if (ZERO_PAD && d1 != nullptr) {
                // data gradient: output position c is voxel c + off of dx (off = 0: evaluated on the original
                // grid, the halo-shell term comes from tdx_conv3_shell.hip).  Positions inside the grid go
                // straight to dx (split over the two inputs of a concatenated conv, plus the optional
                // fused addend = the gradient arriving over the block's residual path).
                const int u0 = c0 + g.off, u1 = c1 + g.off, u2 = c2 + g.off;
                if (u0 >= 0 && u0 < g.Ei[0] && u1 >= 0 && u1 < g.Ei[1] && u2 >= 0 && u2 < g.Ei[2]) {
                    const int64_t u = (int64_t)b * g.in_batch + u0 * g.si[0] + u1 * g.si[1] + u2 * g.si[2];
                    const int n = n0 + cidx * 8;
                    const bool lo = n < D1;
                    bf16* dst = lo ? d1 + u * D1 + n : d2 + u * (Cout - D1) + (n - D1);
                    const bf16* asrc = lo ? (a1 ? a1 + u * D1 + n : nullptr) : (a2 ? a2 + u * (Cout - D1) + (n - D1) : nullptr);
                    if (asrc) {
                        Vec8<HT> va, vb;
                        va.load(reinterpret_cast<const HT*>(&val));
                        vb.load(reinterpret_cast<const HT*>(asrc));
#pragma unroll
                        for (int e = 0; e < 8; ++e) va.v[e] += vb.v[e];
                        va.store(reinterpret_cast<HT*>(dst));
                    } else {
                        *reinterpret_cast<uint4*>(dst) = val;
                    }
                    direct = true;
                }
            }
            if (!direct) {
                const int64_t ov = (int64_t)b * g.out_batch + c0 * g.so[0] + c1 * g.so[1] + c2 * g.so[2];
                if (EXT && init != nullptr) {
                    // continue from a precomputed partial convolution ([B or 1][voxels][Cout] bf16), added in
                    // the coalesced store loop; the statistics below see the sum
                    const int64_t iv = (int64_t)b * g.init_batch + c0 * g.so[0] + c1 * g.so[1] + c2 * g.so[2];
                    Vec8<HT> va, vb;
                    va.load(reinterpret_cast<const HT*>(&val));
                    vb.load(reinterpret_cast<const HT*>(init + iv * Cout + n0 + cidx * 8));
#pragma unroll
                    for (int e = 0; e < 8; ++e) va.v[e] += vb.v[e];
                    va.store(reinterpret_cast<HT*>(&val));
                }
                *reinterpret_cast<uint4*>(y + ov * Cout + n0 + cidx * 8) = val;
            }
            if (gn_acc != nullptr) {
                const unsigned wds[4] = {val.x, val.y, val.z, val.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float lo = H::lo(wds[e]), hi = H::hi(wds[e]);
                    s1[2 * e] += lo; s2[2 * e] += lo * lo;
                    s1[2 * e + 1] += hi; s2[2 * e + 1] += hi * hi;
                }
            }
        }
    }
    if (gn_acc != nullptr) {
        constexpr int NP = 256 / CHUNKS;  // threads per chunk column
        float* red = reinterpret_cast<float*>(smem + 256 * BN * 2);  // [NP][BN][2], behind the output tile
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

template <int NT, bool XT, bool ZP, bool PERM, bool EXT, bool HF>
static int launch_view_h(const void* x1, int C1, const void* x2, int C2, const void* wp, const float* bias, void* y,
                       const ConvViews& v, int Cout, hipStream_t st, double* gn_acc, void* d1, int D1, void* d2,
                       const void* a1, const void* a2, const void* init) {
    constexpr int BN = NT * 32;
    constexpr int HXv = (XT ? 2 : 4) + 2, HYv = (XT ? 16 : 8) + 2, SZv = XT ? 10 : 12;
    const size_t lds = (size_t)2 * (HXv * HYv * SZv * 16 + 64) + (size_t)27 * BN * 32 + 128;
    auto kern = conv3_mfma_kernel<NT, XT, ZP, PERM, EXT, HF>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    dim3 grid((unsigned)v.start[3], Cout / BN);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, (const bf16*)x1, C1, (const bf16*)x2, C2, (const bf16*)wp, bias,
                       (bf16*)y, v, Cout, gn_acc, (bf16*)d1, D1, (bf16*)d2, (const bf16*)a1, (const bf16*)a2, (const bf16*)init);
    return tdx_launch_status();
}
template <int NT, bool XT, bool ZP, bool PERM, bool EXT = false>
static int launch_view(const void* x1, int C1, const void* x2, int C2, const void* wp, const float* bias, void* y,
                       const ConvViews& v, int Cout, hipStream_t st, double* gn_acc, void* d1, int D1, void* d2,
                       const void* a1, const void* a2, const void* init, bool hf) {
    if (hf) return launch_view_h<NT, XT, ZP, PERM, EXT, true>(x1, C1, x2, C2, wp, bias, y, v, Cout, st, gn_acc, d1, D1, d2, a1, a2, init);
    return launch_view_h<NT, XT, ZP, PERM, EXT, false>(x1, C1, x2, C2, wp, bias, y, v, Cout, st, gn_acc, d1, D1, d2, a1, a2, init);
}

int conv3_mfma_launch(const void* x1, int C1, const void* x2, int C2, const void* wp, const float* bias, void* y,
                      const Conv3Geom& g, int Cout, bool zero_pad, hipStream_t st, double* gn_acc, void* d1, int D1,
                      void* d2, const void* a1, const void* a2, const Conv3Ext* ext, const int* slabs_beyond, bool hf) {
    const int NT = (Cout % 64 == 0) ? 2 : 1;
    const void* init = ext ? ext->init : nullptr;
    if ((int64_t)g.Xi * g.Yi * g.Zi * 2 >= (1ll << 31) || (int64_t)g.Xo * g.Yo * g.Zo >= (1ll << 31)) return TDX_ESHAPE;
    static const bool no_thin = getenv("TDX_CONV3_THIN") && atoi(getenv("TDX_CONV3_THIN")) == 0;  // A/B switch

    // global axes: 0 = x, 1 = y, 2 = z (z fastest in memory)
    const int Ei[3] = {g.Xi, g.Yi, g.Zi}, Eo[3] = {g.Xo, g.Yo, g.Zo};
    const int si[3] = {g.Yi * g.Zi, g.Zi, 1}, so[3] = {g.Yo * g.Zo, g.Zo, 1};
    const int bdim[3] = {4, 8, 8};
    // an axis whose extent leaves a remainder of 1-2 voxels gets a thin slab instead of a row of
    // mostly empty main bricks
    // (measured: pays on the two finest U-Net levels, loses to the extra launches below ~50k voxels)
    static const bool force_thin = getenv("TDX_CONV3_THIN") && atoi(getenv("TDX_CONV3_THIN")) == 2;
    const bool big = force_thin || (int64_t)g.Xo * g.Yo * g.Zo >= 60000;
    int main_end[3];
    bool thin[3];
    for (int a = 0; a < 3; ++a) {
        const int rem = Eo[a] % bdim[a];
        thin[a] = !no_thin && rem >= 1 && rem <= 2 && Eo[a] > bdim[a] && big && ext == nullptr;
        main_end[a] = thin[a] ? Eo[a] - rem : Eo[a];
        if (slabs_beyond != nullptr) {  // the main region is somebody else's (the ring kernel's): only what lies beyond it
            if (ext != nullptr || slabs_beyond[a] <= 0 || Eo[a] - slabs_beyond[a] > 2 || Eo[a] < slabs_beyond[a]) return TDX_ESHAPE;
            main_end[a] = slabs_beyond[a];
            thin[a] = main_end[a] < Eo[a];
        }
    }
    // view of region [lo, hi) (global output coordinates) with local axis k = global axis perm[k];
    // returns the number of bricks (0: empty region)
    auto make_view = [&](const int perm[3], const int lo[3], const int hi[3], bool xt, ConvView& v) -> int {
        v.B = g.B; v.off = g.off;
        v.in_batch = g.Xi * g.Yi * g.Zi; v.out_batch = g.Xo * g.Yo * g.Zo;
        v.ld1 = ext && ext->ld1 ? ext->ld1 : C1;
        v.ld2 = ext && ext->ld2 ? ext->ld2 : C2;
        v.init_batch = ext && ext->init && !ext->init_shared ? v.out_batch : 0;
        const int bl[3] = {xt ? 2 : 4, xt ? 16 : 8, 8};
        int64_t n = g.B;
        for (int k = 0; k < 3; ++k) {
            const int a = perm[k];
            v.Ei[k] = Ei[a]; v.si[k] = si[a]; v.so[k] = so[a];
            v.ws[k] = (a == 0) ? 9 : (a == 1 ? 3 : 1);
            v.r0[k] = lo[a]; v.r1[k] = hi[a];
            v.nb[k] = hi[a] > lo[a] ? ceil_div(hi[a] - lo[a], bl[k]) : 0;
            n *= v.nb[k];
        }
        return (int)n;
    };
    auto launch = [&](const ConvViews& vs, bool xt, bool permuted) -> int {
        if (vs.start[3] == 0) return TDX_OK;
#define M3_GO(NTV, XTV, PV)                                                                                             \
    (zero_pad ? launch_view<NTV, XTV, true, PV>(x1, C1, x2, C2, wp, bias, y, vs, Cout, st, gn_acc, d1, D1, d2, a1, a2,  \
                                                nullptr, hf)                                                           \
              : launch_view<NTV, XTV, false, PV>(x1, C1, x2, C2, wp, bias, y, vs, Cout, st, gn_acc, d1, D1, d2, a1, a2, \
                                                 nullptr, hf))
        if (ext != nullptr) {  // strided input / init tensor: forward main bricks only
            if (zero_pad || xt || permuted) return TDX_ESHAPE;
            if (NT == 2)
                return launch_view<2, false, false, false, true>(x1, C1, x2, C2, wp, bias, y, vs, Cout, st, gn_acc, d1, D1, d2,
                                                                 a1, a2, init, hf);
            return launch_view<1, false, false, false, true>(x1, C1, x2, C2, wp, bias, y, vs, Cout, st, gn_acc, d1, D1, d2, a1,
                                                             a2, init, hf);
        }
        if (NT == 2) return xt ? M3_GO(2, true, false) : (permuted ? M3_GO(2, false, true) : M3_GO(2, false, false));
        return xt ? M3_GO(1, true, false) : (permuted ? M3_GO(1, false, true) : M3_GO(1, false, false));
#undef M3_GO
    };
    const int lo[3] = {0, 0, 0}, hi[3] = {main_end[0], main_end[1], main_end[2]};
    // main region: the brick is 4 x 8 x 8; put its short axis on the global axis that leaves the fewest
    // bricks (e.g. 48 x 16 x 12: 36 bricks with the 4 along z instead of 48) -- ragged grids only
    static const bool no_perm = getenv("TDX_CONV3_PERM") && atoi(getenv("TDX_CONV3_PERM")) == 0;  // A/B switch
    int best[3] = {0, 1, 2};
    if (!no_perm && ext == nullptr) {
        const int cand[3][3] = {{0, 1, 2}, {1, 0, 2}, {2, 0, 1}};
        int64_t best_n = -1;
        for (int c = 0; c < 3; ++c) {
            const int64_t n = (int64_t)ceil_div(hi[cand[c][0]], 4) * ceil_div(hi[cand[c][1]], 8) * ceil_div(hi[cand[c][2]], 8);
            if (best_n < 0 || n < best_n) {
                best_n = n;
                for (int k = 0; k < 3; ++k) best[k] = cand[c][k];
            }
        }
    }
    ConvViews mainv;
    const int nmain = make_view(best, lo, hi, false, mainv.v[0]);
    mainv.v[1] = mainv.v[2] = mainv.v[0];
    mainv.start[0] = 0; mainv.start[1] = mainv.start[2] = mainv.start[3] = nmain;
    int rc = slabs_beyond ? TDX_OK : launch(mainv, false, !(best[0] == 0 && best[1] == 1 && best[2] == 2));
    if (rc != TDX_OK) return rc;
    // remainder slabs, all in ONE launch of the thin-brick kernel (each has too few workgroups to
    // fill the chip alone):
    //   x slab: [main_end_x, Xo) x all y x all z           (local axes x, y, z)
    //   y slab: main x range x [main_end_y, Yo) x all z    (local axes y, x, z)
    //   z slab: main x, y ranges x [main_end_z, Zo)         (local axes z, x, y)
    if (thin[0] || thin[1] || thin[2]) {
        const int pm[3][3] = {{0, 1, 2}, {1, 0, 2}, {2, 0, 1}};
        const int l[3][3] = {{main_end[0], 0, 0}, {0, main_end[1], 0}, {0, 0, main_end[2]}};
        const int h[3][3] = {{Eo[0], Eo[1], Eo[2]}, {main_end[0], Eo[1], Eo[2]}, {main_end[0], main_end[1], Eo[2]}};
        ConvViews tv;
        int total = 0;
        for (int r = 0; r < 3; ++r) {
            tv.start[r] = total;
            const int n = make_view(pm[r], l[r], h[r], true, tv.v[r]);
            if (thin[r]) total += n;
        }
        tv.start[3] = total;
        // regions that are not thin own an empty block range [start, start)
        for (int r = 2; r >= 0; --r)
            if (!thin[r]) tv.start[r] = tv.start[r + 1];
        if ((rc = launch(tv, true, false)) != TDX_OK) return rc;
    }
    return TDX_OK;
}
