// Halo-shell term of the 3x3x3 data gradient (adjoint of replicate padding), gfx950.
//
// y[o] = sum_e W[e] x[clamp(o + e)]  (reference ddpm.py:164, padding_mode="replicate")  has the adjoint
//     dx[i] = sum over (o, e) with clamp(o + e) = i of  W[e]^T dy[o]
//           = sum_e wb[e] dy0[i + e]                                  (main term: zero-padded correlation on the
//                                                                      ORIGINAL grid, the conv kernels with ZERO_PAD)
//           + sum over shell positions p != i with clamp(p) = i of g[p],   g[p] = sum_e wb[e] dy0[p + e]
// where wb is the flipped / transposed operand and dy0 is dy continued by zeros.  A shell position p lies one voxel
// outside the grid along 1-3 axes; along such an axis only ONE tap reaches into the grid (e = +1 at p = -1, e = -1 at
// p = E), so g[p] on a face is a 9-tap 2-D correlation of dy's boundary plane -- 1/3 of the taps on 8 % of the voxels,
// 2.9 % of the layer's FLOPs at 192 x 64 x 48.  Evaluating the adjoint on the padded (X+2)(Y+2)(Z+2) grid instead
// (round 1) cost +9 % bricks with thin remainder slabs at the fine levels and 2x the bricks at the deep ones, a
// padded workspace and a fold pass.
//
// This kernel evaluates the shell as three regions of one launch, each a pair of faces with the in-plane extent
// chosen so that every shell position is covered exactly once (edges and corners included):
//     z faces: p_z in {-1, Z},  p_x in [-1, X],  p_y in [-1, Y]
//     y faces: p_y in {-1, Y},  p_x in [-1, X],  p_z in [0, Z)
//     x faces: p_x in {-1, X},  p_y in [0, Y),   p_z in [0, Z)
// A workgroup owns a 16 x 16 patch of one face and BN = 32 NT channels of dx: per K slice the halo'd patch of the
// source plane (18 x 18 voxels) and the slice's weights of the 9 live taps go to LDS; wave w owns patch rows
// 4w .. 4w+3 = two 32-position M tiles.  Results are ADDED onto dx[clamp(p)]: a voxel on exactly one face receives
// exactly one shell position, so it is a plain 16-B read-add-write; edge and corner voxels (up to 7 positions from
// different workgroups) use hardware atomics (global_atomic_add_f32 / global_atomic_pk_add_bf16) -- in whatever order the
// workgroups finish, each add rounded in the tensor's dtype: those voxels are not bit-reproducible from run to run.
// TDX_SHELL_DETERMINISTIC=1 (read per call) takes the other route: every workgroup STORES its fp32 tile into a buffer with
// one row per shell position (each position is computed exactly once), and conv3_shell_fold_kernel adds, for every boundary
// voxel, the up to 7 positions that clamp onto it in a fixed order and rounds once.  Atomics for the
// whole shell were measured first: 44 G atomics/s made the kernel 10x slower than its MFMA work.  Three arithmetic modes mirror the three conv kernels: bf16 MFMA, IEEE fp32 MFMA, and
// split-precision (bf16 hi + lo, three MFMAs per product) on fp32 tensors.
#include "tdx_common.h"
#include "tdx_conv3.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define SH_BF16 0
#define SH_F32 1
#define SH_SPLIT 2
#define SH_F16 3    // as SH_BF16 with IEEE half tensors / operands (v_mfma_f32_32x32x16_f16, global_atomic_pk_add_f16)

// Diagnostic builds only (tools/micro/shell_stamp.hip defines SH_STAMPS): s_memtime stamps of every wave's phases, written
// straight to sh_stamps_dev; the product build carries none of it.
#ifdef SH_STAMPS
#define SH_NSTAMP 16
__device__ unsigned long long* sh_stamps_dev;
#define SH_T()                                                                                                   \
    do {                                                                                                         \
        if ((threadIdx.x & 63) == 0 && sh_stamps_dev != nullptr && nst_ < SH_NSTAMP)                             \
            sh_stamps_dev[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6)) * SH_NSTAMP + nst_] = \
                __builtin_amdgcn_s_memtime();                                                                    \
        ++nst_;                                                                                                  \
    } while (0)
#else
#define SH_T() do {} while (0)
#endif

#define SH_P 16                 // patch edge
#define SH_H (SH_P + 2)         // halo'd patch edge
#define SH_SZ 24                // row stride of the LDS patch: == 8 mod 16, so the 16 lanes of a ds_read_b128 group
                                // (2 rows x 8 columns) hit 16 distinct 16-B slots

struct ShellView {
    int E[3];    // grid extents along the local axes (local axis 0 = the clamped axis of this pair of faces)
    int st[3];   // voxel strides of the local axes
    int org[2];  // first position along local axes 1, 2 (-1 or 0)
    int ext[2];  // number of positions along local axes 1, 2
    int nb[2];   // patches along local axes 1, 2
    int ws[3];   // weight-tap strides of the local axes ({9, 3, 1} permuted)
};

struct ShellRegions {
    ShellView v[3];
    int start[4];   // first block of region r; blocks of a region: [b][face][patch1][patch2]
    int pstart[4];  // first row of region r in the position buffer; rows of a region: [b][face][position 1][position 2]
    int B;
};

// 8 fp32 -> 8 bf16 hi and 8 bf16 lo (as tdx_conv3_mfma_split.hip)
__device__ __forceinline__ void sh_split8(const float4& a, const float4& b, uint4& hi, uint4& lo) {
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

template <int MODE, int NT, int S>
__global__ void __launch_bounds__(256, 2)
conv3_shell_kernel(const void* __restrict__ dy_, const void* __restrict__ wb_, void* __restrict__ d1_, int D1,
                   void* __restrict__ d2_, ShellRegions R, int K, int N, int64_t lo_offset, float* __restrict__ sbuf) {
    constexpr int BN = NT * 32;
    constexpr int KC = MODE == SH_F32 ? 8 : 16;                // channels per K slice (one MFMA K step / four fp32 ones)
    constexpr int Q = 2 * S;                                   // S slices are staged per iteration, as Q half-slice planes
    constexpr int PARTS = MODE == SH_SPLIT ? 2 : 1;            // hi / lo images
    constexpr int APLANE = SH_H * SH_SZ * 16 + 64;             // one half-plane of the patch (16-B entries)
    constexpr int B_ROWS = 9 * BN;
    constexpr int B_PLANE = B_ROWS * 16 + 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sA = smem;                                  // [part][q][APLANE]
    unsigned char* sB = smem + PARTS * Q * APLANE;             // [part][q][B_PLANE]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
#ifdef SH_STAMPS
    int nst_ = 0;
#endif
    SH_T();  // 0: start

    // block -> region, sample, face, patch.  The view is selected by value (uniform selects).
    int bid = blockIdx.x;
    ShellView g = R.v[0];
    int first = 0, prow0 = R.pstart[0];
    if (bid >= R.start[1]) { g = R.v[1]; first = R.start[1]; prow0 = R.pstart[1]; }
    if (bid >= R.start[2]) { g = R.v[2]; first = R.start[2]; prow0 = R.pstart[2]; }
    bid -= first;
    const int q2 = bid % g.nb[1]; bid /= g.nb[1];
    const int q1 = bid % g.nb[0]; bid /= g.nb[0];
    const int face = bid & 1;
    const int b = bid >> 1;
    const int n0 = blockIdx.y * BN;
    const int s0 = face ? g.E[0] - 1 : 0;  // source (= destination) plane along the clamped axis
    const int e0 = face ? -1 : 1;          // the one tap along that axis that reaches into the grid
    const int p1 = g.org[0] + q1 * SH_P, p2 = g.org[1] + q2 * SH_P;  // first position of the patch

    // ---- staging plan of the halo'd source patch: (voxel, half) pieces
    constexpr int A_PIECES = SH_H * SH_H * Q;
    constexpr int A_PER_THREAD = (A_PIECES + 255) / 256;
    int a_src[A_PER_THREAD], a_dst[A_PER_THREAD];
#pragma unroll
    for (int i = 0; i < A_PER_THREAD; ++i) {
        const int p = tid + i * 256;
        a_src[i] = a_dst[i] = -1;
        if (p < A_PIECES) {
            const int hv = p / Q, q = p % Q;
            const int h1 = hv / SH_H, h2 = hv - h1 * SH_H;
            a_dst[i] = q * APLANE + (h1 * SH_SZ + h2) * 16;
            const int c1 = p1 + h1 - 1, c2 = p2 + h2 - 1;
            if (c1 >= 0 && c1 < g.E[1] && c2 >= 0 && c2 < g.E[2])
                a_src[i] = (s0 * g.st[0] + c1 * g.st[1] + c2 * g.st[2]) * Q + q;
        }
    }
    const int64_t batch_vox = (int64_t)b * g.E[0] * g.E[1] * g.E[2];

    // ---- weight staging: pieces (part, row = t9 * BN + n, half); packed [part][K/KC][27][N][KC]
    constexpr int B_PIECES = B_ROWS * Q * PARTS;
    constexpr int B_PER_THREAD = (B_PIECES + 255) / 256;
    int b_goff[B_PER_THREAD], b_dst[B_PER_THREAD];  // element offset inside a slice / LDS byte offset
#pragma unroll
    for (int i = 0; i < B_PER_THREAD; ++i) {
        const int p = tid + i * 256;
        b_goff[i] = b_dst[i] = -1;
        if (p < B_PIECES) {
            const int q = p % Q, row = (p / Q) % B_ROWS, part = (p / Q) / B_ROWS;
            const int t9 = row / BN, n = row - t9 * BN;
            const int tap = (e0 + 1) * g.ws[0] + (t9 / 3) * g.ws[1] + (t9 % 3) * g.ws[2];
            // bf16 / fp32: [K/KC][27][N][KC], read in halves; split images: [K/8][27][N][8], half q = slice q
            b_goff[i] = MODE == SH_SPLIT ? ((q * 27 + tap) * N + n0 + n) * (KC / 2)
                                         : (((q >> 1) * 27 + tap) * N + n0 + n) * KC + (q & 1) * (KC / 2);
            b_dst[i] = (part * Q + q) * B_PLANE + row * 16;
            if (part) b_goff[i] = -2 - b_goff[i];  // lo image: flagged by sign, offset added at load time
        }
    }

    // registers of the slice in flight: 16 B per piece (32 B of fp32 source in split mode)
    uint4 areg[A_PER_THREAD][MODE == SH_SPLIT ? 2 : 1], breg[B_PER_THREAD];
    constexpr bool H16M = MODE == SH_BF16 || MODE == SH_F16;  // 16-bit tensors
    typedef H16<MODE == SH_F16> H;
    const int esz = H16M ? 2 : 4;  // bytes per dy element
    auto load_slice = [&](int c) {
        const unsigned char* xs = reinterpret_cast<const unsigned char*>(dy_) + (batch_vox * K + (int64_t)c * KC * S) * esz;
#pragma unroll
        for (int i = 0; i < A_PER_THREAD; ++i) {
#pragma unroll
            for (int j = 0; j < (MODE == SH_SPLIT ? 2 : 1); ++j) areg[i][j] = make_uint4(0, 0, 0, 0);
            if (a_src[i] >= 0) {
                // a half = KC/2 channels = 16 B (bf16, f32) or 32 B (split: 8 fp32)
                const unsigned char* src = xs + ((int64_t)(a_src[i] / Q) * K + (a_src[i] % Q) * (KC / 2)) * esz;
                areg[i][0] = *reinterpret_cast<const uint4*>(src);
                if (MODE == SH_SPLIT) areg[i][MODE == SH_SPLIT ? 1 : 0] = *reinterpret_cast<const uint4*>(src + 16);
            }
        }
        const int wsz = MODE == SH_F32 ? 4 : 2;
        const unsigned char* wc = reinterpret_cast<const unsigned char*>(wb_) + (int64_t)c * S * 27 * N * KC * wsz;
#pragma unroll
        for (int i = 0; i < B_PER_THREAD; ++i) {
            breg[i] = make_uint4(0, 0, 0, 0);
            if (b_goff[i] != -1) {
                const int64_t off = b_goff[i] >= 0 ? (int64_t)b_goff[i] : (int64_t)(-2 - b_goff[i]) + lo_offset;
                breg[i] = *reinterpret_cast<const uint4*>(wc + off * wsz);
            }
        }
    };
    auto store_slice = [&]() {
#pragma unroll
        for (int i = 0; i < A_PER_THREAD; ++i)
            if (a_dst[i] >= 0) {
                if (MODE == SH_SPLIT) {
                    uint4 hi, lo;
                    const uint4 u0 = areg[i][0], u1 = areg[i][MODE == SH_SPLIT ? 1 : 0];
                    sh_split8(make_float4(__uint_as_float(u0.x), __uint_as_float(u0.y), __uint_as_float(u0.z), __uint_as_float(u0.w)),
                              make_float4(__uint_as_float(u1.x), __uint_as_float(u1.y), __uint_as_float(u1.z), __uint_as_float(u1.w)),
                              hi, lo);
                    *reinterpret_cast<uint4*>(sA + a_dst[i]) = hi;
                    *reinterpret_cast<uint4*>(sA + Q * APLANE + a_dst[i]) = lo;
                } else {
                    *reinterpret_cast<uint4*>(sA + a_dst[i]) = areg[i][0];
                }
            }
#pragma unroll
        for (int i = 0; i < B_PER_THREAD; ++i)
            if (b_dst[i] >= 0) *reinterpret_cast<uint4*>(sB + b_dst[i]) = breg[i];
    };

    // this lane's position of M tile mt: row l1 = 4 wave + 2 mt + (r & 1), column l2 = r >> 1
    int a_h[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
        a_h[mt] = hh * APLANE + ((4 * wave + 2 * mt + (r & 1) + 1) * SH_SZ + (r >> 1) + 1) * 16;
    const int b_off = hh * B_PLANE + r * 16;  // (+ 2 s planes for slice s of the iteration)

    f32x16 acc[NT][2];  // D[row = channel][col = position]
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[nt][mt][i] = 0.f;

    struct Frags { uint4 x[PARTS][2], w[PARTS][NT]; };
    const int nchunks = K / (KC * S);
    SH_T();  // 1: plans done
    load_slice(0);
    for (int c = 0; c < nchunks; ++c) {
        __syncthreads();
        store_slice();
        __syncthreads();
        if (c < 2) SH_T();  // 2, 4: slice c in LDS
        if (c + 1 < nchunks) load_slice(c + 1);
        // Fragment ring, prefetch distance RING - 1: the LDS reads of tap-step ts + 2 are issued before the MFMAs of step ts
        // (2-12 MFMAs per step do not cover an LDS round trip; with one or two waves per SIMD nothing else does).
        constexpr int NS = 9 * S;
        constexpr int NREAD = PARTS * (2 + NT);  // ds_read_b128 per step
        constexpr int NMFMA = 2 * NT * (MODE == SH_F32 ? 4 : (MODE == SH_SPLIT ? 3 : 1));
        constexpr int RING = (MODE == SH_SPLIT && NT == 2) ? 2 : 3;  // (the widest split tile has no registers for 3)
        Frags ring[RING];
        auto read_step = [&](int ts, Frags& f) {
            const int t9 = ts % 9, sl = ts / 9;
            const int toff = ((t9 / 3 - 1) * SH_SZ + (t9 % 3 - 1)) * 16;
#pragma unroll
            for (int pt = 0; pt < PARTS; ++pt) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    f.x[pt][mt] = *reinterpret_cast<const uint4*>(sA + (pt * Q + 2 * sl) * APLANE + a_h[mt] + toff);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    f.w[pt][nt] = *reinterpret_cast<const uint4*>(sB + (pt * Q + 2 * sl) * B_PLANE + b_off + (t9 * BN + nt * 32) * 16);
            }
        };
        auto mfma_step = [&](const Frags& f) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    if (MODE == SH_F32) {
                        const uint4 w = f.w[0][nt], x = f.x[0][mt];
                        acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w.x), __uint_as_float(x.x), acc[nt][mt], 0, 0, 0);
                        acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w.y), __uint_as_float(x.y), acc[nt][mt], 0, 0, 0);
                        acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w.z), __uint_as_float(x.z), acc[nt][mt], 0, 0, 0);
                        acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w.w), __uint_as_float(x.w), acc[nt][mt], 0, 0, 0);
                    } else {
                        const bf16x8 wh = *reinterpret_cast<const bf16x8*>(&f.w[0][nt]), xh = *reinterpret_cast<const bf16x8*>(&f.x[0][mt]);
                        acc[nt][mt] = H::mfma(wh, xh, acc[nt][mt]);
                        if (MODE == SH_SPLIT) {
                            const bf16x8 wl = *reinterpret_cast<const bf16x8*>(&f.w[PARTS - 1][nt]);
                            const bf16x8 xl = *reinterpret_cast<const bf16x8*>(&f.x[PARTS - 1][mt]);
                            acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xl, acc[nt][mt], 0, 0, 0);
                            acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, xh, acc[nt][mt], 0, 0, 0);
                        }
                    }
                }
        };
#pragma unroll
        for (int i = 0; i < RING - 1; ++i) read_step(i, ring[i]);
        __builtin_amdgcn_sched_group_barrier(0x100, (RING - 1) * NREAD, 0);
#pragma unroll
        for (int ts = 0; ts < NS; ++ts) {
            constexpr int D = RING - 1;
            if (ts + D < NS) read_step(ts + D, ring[(ts + D) % RING]);
            mfma_step(ring[ts % RING]);
            if (ts + D < NS) __builtin_amdgcn_sched_group_barrier(0x100, NREAD, 0);  // DS reads of step ts + D first,
            __builtin_amdgcn_sched_group_barrier(0x008, NMFMA, 0);                    // then the MFMAs of step ts
        }
        if (c < 2) SH_T();  // 3, 5: MFMAs of slice c issued
    }
    SH_T();  // K loop done

    // ---- fold.  Lane (r, hh) holds, for M tile mt, its position and channels nt*32 + 8 j + 4 hh + (0..3) in
    // accumulator registers 4 j .. 4 j + 3: transposed through an fp32 LDS tile [256 positions][BN] so that a thread
    // owns 16 B of one dx row.  A destination voxel on exactly ONE face receives exactly one shell position (this
    // one): plain read-add-write of whole 16-B pieces.  Edge and corner voxels receive up to 7 positions from
    // different workgroups: hardware atomics there (1-3 % of the shell).
    __syncthreads();
    unsigned char* sO = smem;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ch = nt * 32 + 8 * j + 4 * hh;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int v = (4 * wave + 2 * mt + (r & 1)) * SH_P + (r >> 1);
                *reinterpret_cast<float4*>(sO + v * (BN * 4) + (((ch >> 2) ^ (v & (BN / 4 - 1))) << 4)) =
                    make_float4(acc[nt][mt][4 * j], acc[nt][mt][4 * j + 1], acc[nt][mt][4 * j + 2], acc[nt][mt][4 * j + 3]);
            }
        }
    __syncthreads();
    SH_T();  // transposed tile in LDS
    constexpr int CH = H16M ? 8 : 4;  // channels per 16-B piece of dx
    constexpr int CHUNKS = BN / CH;
#pragma unroll 4
    for (int i = 0; i < CHUNKS; ++i) {
        const int p = tid + i * 256;
        const int v = p / CHUNKS, cidx = p % CHUNKS;
        const int l1 = v / SH_P, l2 = v % SH_P;
        if (q1 * SH_P + l1 >= g.ext[0] || q2 * SH_P + l2 >= g.ext[1]) continue;
        if (sbuf != nullptr) {  // deterministic route: this position's fp32 row, folded by conv3_shell_fold_kernel
            const int64_t row = prow0 + ((int64_t)(b * 2 + face) * g.ext[0] + q1 * SH_P + l1) * g.ext[1] + q2 * SH_P + l2;
            float* dst = sbuf + row * N + n0 + cidx * CH;
#pragma unroll
            for (int q = 0; q < CH / 4; ++q) {
                const int c4 = cidx * (CH / 4) + q;
                *reinterpret_cast<float4*>(dst + 4 * q) =
                    *reinterpret_cast<const float4*>(sO + v * (BN * 4) + ((c4 ^ (v & (BN / 4 - 1))) << 4));
            }
            continue;
        }
        const int c1 = min(max(p1 + l1, 0), g.E[1] - 1), c2 = min(max(p2 + l2, 0), g.E[2] - 1);
        // on a second face (a grid one voxel thick has both faces of the clamped axis on the same plane)
        const bool shared = c1 == 0 || c1 == g.E[1] - 1 || c2 == 0 || c2 == g.E[2] - 1 || g.E[0] == 1;
        const int64_t u = batch_vox + (int64_t)s0 * g.st[0] + (int64_t)c1 * g.st[1] + (int64_t)c2 * g.st[2];
        const int n = n0 + cidx * CH;
        const bool lo = n < D1;
        const int64_t idx = lo ? u * D1 + n : u * (N - D1) + (n - D1);
        float val[CH];
#pragma unroll
        for (int q = 0; q < CH / 4; ++q) {
            const int c4 = cidx * (CH / 4) + q;
            const float4 t = *reinterpret_cast<const float4*>(sO + v * (BN * 4) + ((c4 ^ (v & (BN / 4 - 1))) << 4));
            val[4 * q] = t.x; val[4 * q + 1] = t.y; val[4 * q + 2] = t.z; val[4 * q + 3] = t.w;
        }
        if (H16M) {
            typedef typename H::T HT;
            HT* dst = reinterpret_cast<HT*>(lo ? d1_ : d2_) + idx;
            if (shared) {
#pragma unroll
                for (int e = 0; e < CH / 2; ++e) {
                    const unsigned pk = H::pack2(val[2 * e], val[2 * e + 1]);
                    if (MODE == SH_F16)
                        __builtin_amdgcn_global_atomic_fadd_v2f16((__attribute__((address_space(1))) f16x2_t*)(dst + 2 * e),
                                                                  *reinterpret_cast<const f16x2_t*>(&pk));
                    else
                        __builtin_amdgcn_global_atomic_fadd_v2bf16((__attribute__((address_space(1))) bf16x2_t*)(dst + 2 * e),
                                                                   *reinterpret_cast<const bf16x2_t*>(&pk));
                }
            } else {
                Vec8<HT> o;
                o.load(dst);
#pragma unroll
                for (int e = 0; e < 8; ++e) o.v[e] += val[e < CH ? e : 0];
                o.store(dst);
            }
        } else {
            float* dst = reinterpret_cast<float*>(lo ? d1_ : d2_) + idx;
            if (shared) {
#pragma unroll
                for (int e = 0; e < 4; ++e) atomicAdd(dst + e, val[e]);
            } else {
                float4 o = *reinterpret_cast<const float4*>(dst);
                o.x += val[0]; o.y += val[1]; o.z += val[2]; o.w += val[3];
                *reinterpret_cast<float4*>(dst) = o;
            }
        }
    }
#ifdef SH_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    SH_T();  // fold done
}

// Deterministic route, second pass: dx[u] += sum over the shell positions p != u with clamp(p) = u of their rows of the
// position buffer, in a fixed order (z outermost), one rounding.  Threads walk the boundary voxels only: first the two z
// planes, then the y planes between them, then the x planes inside both.
template <typename T>
__global__ void __launch_bounds__(256)
conv3_shell_fold_kernel(const float* __restrict__ sbuf, T* __restrict__ d1, int D1, T* __restrict__ d2, ShellRegions R, int X,
                        int Y, int Z, int N) {
    const int groups = N >> 3;
    const int64_t nz = 2 * (int64_t)X * Y, ny = Z > 2 ? 2 * (int64_t)X * (Z - 2) : 0, nx = (Z > 2 && Y > 2) ? 2 * (int64_t)(Y - 2) * (Z - 2) : 0;
    const int64_t per_sample = (Z == 1 ? nz / 2 : nz) + (Y == 1 ? ny / 2 : ny) + (X == 1 ? nx / 2 : nx);
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)R.B * per_sample * groups) return;
    const int cg = (int)(idx % groups);
    int64_t j = idx / groups;
    const int b = (int)(j / per_sample);
    j -= (int64_t)b * per_sample;
    int u[3];  // (x, y, z)
    const int64_t nzz = Z == 1 ? nz / 2 : nz, nyy = Y == 1 ? ny / 2 : ny;
    if (j < nzz) {
        const int f = (int)(j / ((int64_t)X * Y)); const int64_t rem = j % ((int64_t)X * Y);
        u[2] = f ? Z - 1 : 0; u[0] = (int)(rem / Y); u[1] = (int)(rem % Y);
    } else if (j < nzz + nyy) {
        j -= nzz;
        const int f = (int)(j / ((int64_t)X * (Z - 2))); const int64_t rem = j % ((int64_t)X * (Z - 2));
        u[1] = f ? Y - 1 : 0; u[0] = (int)(rem / (Z - 2)); u[2] = 1 + (int)(rem % (Z - 2));
    } else {
        j -= nzz + nyy;
        const int f = (int)(j / ((int64_t)(Y - 2) * (Z - 2))); const int64_t rem = j % ((int64_t)(Y - 2) * (Z - 2));
        u[0] = f ? X - 1 : 0; u[1] = 1 + (int)(rem / (Z - 2)); u[2] = 1 + (int)(rem % (Z - 2));
    }
    const int E[3] = {X, Y, Z};
    int lo[3], hi[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        lo[a] = u[a] == 0 ? -1 : u[a];
        hi[a] = u[a] == E[a] - 1 ? E[a] : u[a];
    }
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int pz = lo[2]; pz <= hi[2]; ++pz)
        for (int py = lo[1]; py <= hi[1]; ++py)
            for (int px = lo[0]; px <= hi[0]; ++px) {
                if (px == u[0] && py == u[1] && pz == u[2]) continue;  // the voxel itself: the main term
                // owner region of the position: z faces own every (x, y); y faces the rest with z inside; x faces the rest
                int r, face, a1, a2;
                if (pz < 0 || pz >= Z) { r = 0; face = pz >= Z; a1 = px; a2 = py; }
                else if (py < 0 || py >= Y) { r = 1; face = py >= Y; a1 = px; a2 = pz; }
                else { r = 2; face = px >= X; a1 = py; a2 = pz; }
                const ShellView& g = R.v[r];
                const int64_t row = R.pstart[r] + ((int64_t)(b * 2 + face) * g.ext[0] + (a1 - g.org[0])) * g.ext[1] + (a2 - g.org[1]);
                const float* p = sbuf + row * N + cg * 8;
                const float4 va = *reinterpret_cast<const float4*>(p), vb = *reinterpret_cast<const float4*>(p + 4);
                acc[0] += va.x; acc[1] += va.y; acc[2] += va.z; acc[3] += va.w;
                acc[4] += vb.x; acc[5] += vb.y; acc[6] += vb.z; acc[7] += vb.w;
            }
    const int64_t vox = (((int64_t)b * X + u[0]) * Y + u[1]) * Z + u[2];
    const int n = cg * 8;
    T* dst = n < D1 ? d1 + vox * D1 + n : d2 + vox * (N - D1) + (n - D1);
    Vec8<T> o;
    o.load(dst);
#pragma unroll
    for (int e = 0; e < 8; ++e) o.v[e] += acc[e];
    o.store(dst);
}

template <int MODE, int NT, int S>
static int shell_go(const void* dy, const void* wb, void* d1, int D1, void* d2, const ShellRegions& R, int K, int N,
                    int64_t lo_offset, float* sbuf, hipStream_t st) {
    constexpr int BN = NT * 32;
    constexpr int PARTS = MODE == SH_SPLIT ? 2 : 1;
    size_t lds = (size_t)PARTS * 2 * S * (SH_H * SH_SZ * 16 + 64) + (size_t)PARTS * 2 * S * (9 * BN * 16 + 64);
    if (lds < (size_t)SH_P * SH_P * BN * 4) lds = (size_t)SH_P * SH_P * BN * 4;  // the fold's fp32 tile
    auto kern = conv3_shell_kernel<MODE, NT, S>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)R.start[3], N / BN), dim3(256), lds, st, dy, wb, d1, D1, d2, R, K, N, lo_offset, sbuf);
    return tdx_launch_status();
}

// dx[clamp(p)] += g[p] over the halo shell.  dy: [B][X][Y][Z][K]; wb: the packed data-gradient operand of
// tdx_conv3_pack_weight for (K -> N); dx split over d1 (N channels [0, D1)) and d2.  mode: SH_BF16 / SH_F32 / SH_SPLIT.
size_t conv3_shell_buffer_bytes(int B, int X, int Y, int Z, int N) {  // the deterministic route's position buffer
    const size_t pos = 2 * ((size_t)(X + 2) * (Y + 2) + (size_t)(X + 2) * Z + (size_t)Y * Z) * B;
    return pos * N * sizeof(float);
}

// sbuf: nullptr, or conv3_shell_buffer_bytes() of scratch for the deterministic route (TDX_SHELL_DETERMINISTIC=1)
int conv3_shell_launch(const void* dy, const void* wb, void* d1, int D1, void* d2, int B, int X, int Y, int Z, int K, int N,
                       int mode, hipStream_t st, void* sbuf_) {
    const int E[3] = {X, Y, Z}, str[3] = {Y * Z, Z, 1}, tapw[3] = {9, 3, 1};
    // region r: clamped global axis ax[r][0], in-plane axes ax[r][1], ax[r][2]; in-plane extents include the shell
    // of the axes whose faces come later in the list (so every shell position is owned by exactly one region)
    static const int ax[3][3] = {{2, 0, 1}, {1, 0, 2}, {0, 1, 2}};
    static const int wide[3][2] = {{1, 1}, {1, 0}, {0, 0}};
    ShellRegions R;
    R.B = B;
    int total = 0, prows = 0;
    for (int r = 0; r < 3; ++r) {
        ShellView& v = R.v[r];
        for (int k = 0; k < 3; ++k) { v.E[k] = E[ax[r][k]]; v.st[k] = str[ax[r][k]]; v.ws[k] = tapw[ax[r][k]]; }
        for (int k = 0; k < 2; ++k) {
            v.org[k] = wide[r][k] ? -1 : 0;
            v.ext[k] = v.E[k + 1] + (wide[r][k] ? 2 : 0);
            v.nb[k] = ceil_div(v.ext[k], SH_P);
        }
        R.start[r] = total;
        total += B * 2 * v.nb[0] * v.nb[1];
        R.pstart[r] = prows;
        prows += B * 2 * v.ext[0] * v.ext[1];
    }
    R.start[3] = total;
    R.pstart[3] = prows;
    const char* det_env = getenv("TDX_SHELL_DETERMINISTIC");
    const bool det = (det_env && atoi(det_env) != 0) || tdx_deterministic();
    float* sbuf = (sbuf_ != nullptr && det && (N % 8) == 0) ? (float*)sbuf_ : nullptr;
    const int64_t lo_offset = (int64_t)27 * K * N;
    // 64-wide channel tiles where the launch fills the chip without them; two K slices per iteration (half the
    // barriers, twice the loads in flight: the deep levels walk K = 512 with one or two workgroups per CU) where the
    // registers allow and K divides
    const bool wide_n = (N % 64) == 0 && (int64_t)total * (N / 64) >= 1024;
    const int kc2 = (mode == SH_F32 ? 8 : 16) * 2;
    const bool two = (K % kc2) == 0 && !(mode == SH_SPLIT && wide_n);
#define SH_GO(M, NTV, SV) shell_go<M, NTV, SV>(dy, wb, d1, D1, d2, R, K, N, M == SH_SPLIT ? lo_offset : 0, sbuf, st)
#define SH_PICK(M) (wide_n ? (two ? SH_GO(M, 2, 2) : SH_GO(M, 2, 1)) : (two ? SH_GO(M, 1, 2) : SH_GO(M, 1, 1)))
    int rc;
    if (mode == SH_BF16) rc = SH_PICK(SH_BF16);
    else if (mode == SH_F16) rc = SH_PICK(SH_F16);
    else if (mode == SH_F32) rc = SH_PICK(SH_F32);
    else if (wide_n) rc = SH_GO(SH_SPLIT, 2, 1);
    else rc = two ? SH_GO(SH_SPLIT, 1, 2) : SH_GO(SH_SPLIT, 1, 1);
#undef SH_PICK
#undef SH_GO
    if (rc != TDX_OK || sbuf == nullptr) return rc;
    const int64_t nz = 2 * (int64_t)X * Y, ny = Z > 2 ? 2 * (int64_t)X * (Z - 2) : 0, nx = (Z > 2 && Y > 2) ? 2 * (int64_t)(Y - 2) * (Z - 2) : 0;
    const int64_t per_sample = (Z == 1 ? nz / 2 : nz) + (Y == 1 ? ny / 2 : ny) + (X == 1 ? nx / 2 : nx);
    const int64_t threads = (int64_t)B * per_sample * (N / 8);
    if (mode == SH_F16)
        hipLaunchKernelGGL(conv3_shell_fold_kernel<f16>, dim3(ceil_div(threads, 256)), dim3(256), 0, st, sbuf, (f16*)d1, D1, (f16*)d2, R, X, Y, Z, N);
    else if (mode == SH_BF16)
        hipLaunchKernelGGL(conv3_shell_fold_kernel<bf16>, dim3(ceil_div(threads, 256)), dim3(256), 0, st, sbuf, (bf16*)d1, D1, (bf16*)d2, R, X, Y, Z, N);
    else
        hipLaunchKernelGGL(conv3_shell_fold_kernel<float>, dim3(ceil_div(threads, 256)), dim3(256), 0, st, sbuf, (float*)d1, D1, (float*)d2, R, X, Y, Z, N);
    return tdx_launch_status();
}
