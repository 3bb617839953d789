// Persistent LDS-DMA "ring" form of the bf16 MFMA implicit-GEMM 3x3x3 convolution for gfx950: forward (replicate
// padding) and the main term of the data gradient (zero padding) on the two finest U-Net levels.
//
// Why (profiles/r10_conv3_stamps.txt: s_memtime stamps of the brick kernel tdx_conv3_mfma.hip, level-0 64 -> 64 layer):
// of a brick workgroup's 48 k cycles, 8.8 k are its prologue (nothing to do until the first slice arrives from HBM),
// 4 x 1.3 k are ds_write phases (global -> VGPR -> LDS), 3.1 k the epilogue, and its four MFMA phases last 6.9 k cycles
// each instead of the 3.5 k they take alone on the SIMDs, because the two co-resident workgroups fall into step: both
// in their MFMA phase, then both staging.  The matrix pipe idles half the time with no resource saturated.
//
// This kernel keeps the matrix-core loop of the brick kernel (same LDS images, fragment addresses, MFMA order: results
// are bit-identical) and changes everything around it:
//   * ONE persistent 8-wave workgroup per CU walks its own list of 8 x 8 x 8 bricks (512 voxels: wave w = x plane w,
//     two 32-voxel M tiles x NT 32-channel N tiles per wave).  No prologue per brick: the next brick's first slice is
//     already in flight while the current brick computes.
//   * Staging is LDS-DMA only (global_load_lds_dwordx4: no staging registers, no ds_write phase).  A K slice's brick
//     (10 x 10 x 10 halo'd voxels x 16 channels, 38 KiB pieces) is double-buffered; its weights arrive in three 9-tap
//     groups (one per tap x offset) through a ring of RG_WSLOTS slots, two units ahead of their use.  A "unit" = one
//     (slice, tap group) = 36 (NT = 2) MFMAs per wave behind ONE s_barrier; the DMA instructions of the units ahead
//     are issued between the MFMAs of the current one.  Arrival is tracked with counted s_waitcnt vmcnt(N): every wave
//     issues the same number of DMA instructions per unit (missing pieces are dummies into a scratch KiB), so N is a
//     compile-time constant per (tap group, first-slice-after-an-epilogue) case.
//   * The weights are re-streamed from L2 once per 512 voxels instead of once per 256.
//   * Epilogue per brick: accumulators (+ bias) -> bf16 -> a wave-private 4 KiB LDS tile (in the brick buffer that
//     just became free) -> whole 128-B voxel rows to HBM; GroupNorm moments are accumulated in registers ACROSS the
//     bricks of a sample and flushed (LDS reduce over the 8 waves + one f64 atomic per channel and moment) only when
//     the sample changes: two flushes per workgroup instead of one per brick.
//   * Workgroup -> bricks: every XCD owns one contiguous range of bricks (halo voxels shared through one L2); inside
//     an XCD the 32 workgroups interleave, and the N tiles of one brick run on neighbouring workgroups at the same time.
//
// Used when the grid fills 8 x 8 x 8 bricks well (tdx_conv3_ring_supported); everything else stays on the brick kernel.
#include "tdx_common.h"
#include "tdx_conv3.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define RG_HY 10
#define RG_SZ 12                        // padded z stride of the LDS brick image (conflict-free 16-B fragment reads)
#define RG_ENT (10 * 10 * RG_SZ)        // 1200 entries per half plane
#define RG_APIECES 19                   // 1-KiB DMA pieces per half plane (1216 entries)
#define RG_APLANE (RG_APIECES * 1024)
#define RG_ABUF (2 * RG_APLANE)         // one brick slice: channels 0-7 | 8-15
#define RG_BPW 5                        // brick pieces per wave and slice (8 x 5 = 40 >= 38)
#define RG_WSLOTS 3                     // weight-group ring (lookahead 2 units)
#define RG_WAVES 8

struct RingArgs {
    const bf16* x1; const bf16* x2; int C1, C2;
    const bf16* wp; const float* bias; bf16* y;
    int B, X, Y, Z, Cout;
    int nbx, nby, nbz, ntn;              // bricks per axis, N tiles
    double* gn_acc;
    bf16* d1; bf16* d2; int D1; const bf16* a1; const bf16* a2;   // data gradient: dx split over two tensors, fused addends
    const void* zeros;                   // >= 16 zero bytes (zero padding)
    unsigned long long* stamps;          // diagnostic builds only
};

// One LDS-DMA instruction: every lane copies 16 B from its own global address to LDS byte address lds + 16 * lane.
// Inline assembly, not the builtin: the compiler would order every later ds_read behind the copy with vmcnt(0).
__device__ __forceinline__ void rg_dma(const void* gsrc, unsigned lds) {
    lds = __builtin_amdgcn_readfirstlane(lds);
    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds) : "memory");
}
// the same with a uniform 64-bit base and a 32-bit per-lane byte offset
__device__ __forceinline__ void rg_dma_off(const void* sbase, unsigned voff, unsigned lds) {
    lds = __builtin_amdgcn_readfirstlane(lds);
    asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds) : "memory");
}
#define RG_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
__device__ __forceinline__ void rg_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's LDS stores are performed before the others proceed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// Diagnostic builds only (tools/micro/ring_stamp.hip defines RG_STAMPS): s_memtime stamps of bricks RG_STAMP_B0 .. +1
// of every wave, kept in spare LDS and dumped to A.stamps at the end; the product build carries none of it.
#ifdef RG_STAMPS
#define RG_NSTAMP 160
#define RG_T()                                                                                        \
    do {                                                                                              \
        if (ord >= RG_STAMP_B0 && ord < RG_STAMP_B0 + 2) {                                            \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                               \
            if (lane == 0 && nst < RG_NSTAMP) sStamp[wave * RG_NSTAMP + nst] = t_;                    \
            ++nst;                                                                                    \
        }                                                                                             \
    } while (0)
#else
#define RG_T() do { } while (0)
#endif

template <int BN>
__device__ __forceinline__ int rg_tile_addr(int v, int c) {  // wave-private [32 voxels][BN] bf16 tile, 16-B chunk c of row v
    if (BN == 64) return v * 128 + ((c ^ (v & 7)) << 4);
    return v * 64 + ((c ^ ((v >> 1) & 3)) << 4);
}

template <int NT, bool ZP>
__global__ void __launch_bounds__(512, 2) conv3_ring_kernel(RingArgs A) {
    constexpr int BN = NT * 32;
    constexpr int WPIECES = (9 * BN * 16 + 1023) / 1024;  // per half plane of a 9-tap group: 9 / 5
    constexpr int WPLANE = WPIECES * 1024;
    constexpr int WSLOT = 2 * WPLANE;
    constexpr int WPW = (2 * WPIECES + RG_WAVES - 1) / RG_WAVES;  // weight pieces per wave and unit: 3 / 2
    constexpr int CH = BN / 8;                            // 16-B chunks per output row
    constexpr int VPI = 64 / CH;                          // voxels per store instruction
    constexpr int NST = 2 * (32 / VPI);                   // global stores per wave and brick: 8 / 4
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char* sA = smem;                             // [2][RG_ABUF]
    unsigned char* sW = smem + 2 * RG_ABUF;               // [RG_WSLOTS][WSLOT]
    unsigned char* sD = sW + RG_WSLOTS * WSLOT;           // 1 KiB sink of the dummy pieces
    float* sBias = reinterpret_cast<float*>(sD + 1024);   // [BN]
    float* sRed = sBias + BN;                             // [RG_WAVES][BN][2]
#ifdef RG_STAMPS
    unsigned long long* sStamp = reinterpret_cast<unsigned long long*>(sRed + RG_WAVES * BN * 2);
    int nst = 0;
#endif
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)smem;
    const unsigned ldsA = lds0, ldsW = lds0 + 2 * RG_ABUF, ldsD = ldsW + RG_WSLOTS * WSLOT;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;

    // ---- this workgroup's bricks: XCD x owns [lo, hi); its 32 workgroups = (32 / ntn) brick lanes x ntn N tiles
    const int hw = blockIdx.x, xcd = hw & 7, slot = hw >> 3;
    const int ntile = slot % A.ntn, kx = slot / A.ntn, KX = 32 / A.ntn;
    const int nbricks = A.B * A.nbx * A.nby * A.nbz;
    const int lo = xcd * (nbricks >> 3) + min(xcd, nbricks & 7), hi = lo + (nbricks >> 3) + (xcd < (nbricks & 7) ? 1 : 0);
    const int nmine = hi - lo > kx ? (hi - lo - kx + KX - 1) / KX : 0;
    if (nmine == 0) return;
    const int n0 = ntile * BN;
    const int Cin = A.C1 + A.C2, nsl = Cin >> 4;
    const int YZ = A.Y * A.Z, V = A.X * YZ;

    if (tid < BN) sBias[tid] = A.bias ? A.bias[n0 + tid] : 0.f;
    rg_barrier();

    // ---- per-lane DMA geometry (fixed for the kernel)
    // brick piece i of this wave: pi = wave * 5 + i -> (half, entries 64 q .. 64 q + 63)
    int hxyz[RG_BPW];
#pragma unroll
    for (int i = 0; i < RG_BPW; ++i) {
        const int pi = min(wave * RG_BPW + i, 2 * RG_APIECES - 1);
        const int e = min((pi % RG_APIECES) * 64 + lane, RG_ENT - 1);
        const int hx = e / (RG_HY * RG_SZ), rem = e - hx * (RG_HY * RG_SZ);
        const int hy = rem / RG_SZ, hz = min(rem - hy * RG_SZ, 9);
        hxyz[i] = hx | (hy << 8) | (hz << 16) | ((pi / RG_APIECES) << 24);
    }
    // weight piece j of this wave: pj = wave * WPW + j -> (half, entries 64 p ..): lane byte offset inside the group
    unsigned wlane[WPW];
#pragma unroll
    for (int j = 0; j < WPW; ++j) {
        const int pj = min(wave * WPW + j, 2 * WPIECES - 1);
        const int half = pj / WPIECES, e = min((pj % WPIECES) * 64 + lane, 9 * BN - 1);
        wlane[j] = (unsigned)(((e / BN) * A.Cout + (e % BN)) * 32 + half * 16);
    }

    // ---- fragment geometry (as tdx_conv3_mfma.hip): M tile mt of wave w: x = w, y = 4 mt + (r & 3), z = r >> 2
    int a_h[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) a_h[mt] = ((wave + 1) * RG_HY + (4 * mt + (r & 3) + 1)) * RG_SZ + ((r >> 2) + 1);
    int b_off[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) b_off[nt] = hh * WPLANE + (nt * 32 + r) * 16;

    // ---- issue cursors
    int vox[RG_BPW];       // source voxel (inside the sample, clamped; -1: zero) of this lane's pieces, for `vb`'s brick
    int vb_b = 0;          // sample of the brick `vox` describes
    auto brick_coords = [&](int ord, int& b, int& o0, int& o1, int& o2) {
        int id = lo + kx + KX * ord;
        const int bz = id % A.nbz; id /= A.nbz;
        const int by = id % A.nby; id /= A.nby;
        const int bx = id % A.nbx; id /= A.nbx;
        b = id; o0 = bx * 8; o1 = by * 8; o2 = bz * 8;
    };
    auto plan_brick = [&](int ord) {
        int b, o0, o1, o2;
        brick_coords(min(ord, nmine - 1), b, o0, o1, o2);
        vb_b = b;
#pragma unroll
        for (int i = 0; i < RG_BPW; ++i) {
            int s0 = o0 + (hxyz[i] & 0xff) - 1, s1 = o1 + ((hxyz[i] >> 8) & 0xff) - 1, s2 = o2 + ((hxyz[i] >> 16) & 0xff) - 1;
            bool ok = true;
            if (ZP) ok = s0 >= 0 && s0 < A.X && s1 >= 0 && s1 < A.Y && s2 >= 0 && s2 < A.Z;
            else { s0 = min(max(s0, 0), A.X - 1); s1 = min(max(s1, 0), A.Y - 1); s2 = min(max(s2, 0), A.Z - 1); }
            vox[i] = ok ? (s0 * A.Y + s1) * A.Z + s2 : -1;
        }
    };
    // piece i of brick slice s of the planned brick -> buffer `buf` (one LDS-DMA instruction)
    auto issue_brick_piece = [&](int s, int buf, int i) {
        const int k0 = s << 4;
        const bf16* xs; int Cs, kk;
        if (k0 < A.C1) { xs = A.x1; Cs = A.C1; kk = k0; } else { xs = A.x2; Cs = A.C2; kk = k0 - A.C1; }
        const bf16* base = xs + ((int64_t)vb_b * V) * Cs + kk;  // uniform
        const int pi = wave * RG_BPW + i;
        const int half = hxyz[i] >> 24;
        const unsigned dst = pi < 2 * RG_APIECES ? ldsA + buf * RG_ABUF + half * RG_APLANE + (pi % RG_APIECES) * 1024 : ldsD;
        if (ZP) {
            const bf16* src = vox[i] >= 0 ? base + (int64_t)vox[i] * Cs + half * 8 : reinterpret_cast<const bf16*>(A.zeros);
            rg_dma(src, dst);
        } else {
            rg_dma_off(base, (unsigned)(vox[i] * Cs + half * 8) * 2u, dst);
        }
    };
    // piece j of weight group (slice s, tap x offset g) -> ring slot
    auto issue_weight_piece = [&](int s, int g, int wslot, int j) {
        const bf16* base = A.wp + ((int64_t)(s * 27 + g * 9) * A.Cout + n0) * 16;  // uniform
        const int pj = wave * WPW + j;
        const unsigned dst = pj < 2 * WPIECES ? ldsW + wslot * WSLOT + (pj / WPIECES) * WPLANE + (pj % WPIECES) * 1024 : ldsD;
        rg_dma_off(base, wlane[j], dst);
    };
    auto issue_brick = [&](int s, int buf) {
#pragma unroll
        for (int i = 0; i < RG_BPW; ++i) issue_brick_piece(s, buf, i);
    };
    auto issue_weights = [&](int s, int g, int wslot) {
#pragma unroll
        for (int j = 0; j < WPW; ++j) issue_weight_piece(s, g, wslot, j);
    };

    // ---- GroupNorm moments of this workgroup's bricks of the current sample (lane: chunk lane % CH, 8 channels)
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    f32x2 p1[4], p2[4];  // packed pairs (channels 2 e, 2 e + 1): one v_pk_add_f32 + one v_pk_fma_f32 per bf16 pair
#pragma unroll
    for (int e = 0; e < 4; ++e) p1[e] = p2[e] = f32x2{0.f, 0.f};
    auto flush_stats = [&](int b) {
        float s1[8], s2[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { s1[2 * e] = p1[e].x; s1[2 * e + 1] = p1[e].y; s2[2 * e] = p2[e].x; s2[2 * e + 1] = p2[e].y; }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
            for (int o = CH; o < 64; o <<= 1) { s1[e] += __shfl_xor(s1[e], o, 64); s2[e] += __shfl_xor(s2[e], o, 64); }
        }
        if (lane < CH) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                sRed[(wave * BN + lane * 8 + e) * 2] = s1[e];
                sRed[(wave * BN + lane * 8 + e) * 2 + 1] = s2[e];
            }
        }
        rg_barrier();
        if (tid < BN * 2) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < RG_WAVES; ++w) t += sRed[w * BN * 2 + tid];
            const int rep = blockIdx.x & (TDX_GN_REPLICAS - 1);
            atomicAdd(&A.gn_acc[(((size_t)rep * A.B + b) * A.Cout + n0) * 2 + tid], (double)t);
        }
        rg_barrier();
#pragma unroll
        for (int e = 0; e < 4; ++e) p1[e] = p2[e] = f32x2{0.f, 0.f};
    };

    // ---- prologue: brick 0 / slice 0, weight units 0 and 1
    plan_brick(0);
    issue_brick(0, 0);
    issue_weights(0, 0, 0);
    issue_weights(0, 1, 1);
    // cursors of what is issued next: brick slice instance qi (slice qs of brick qb), weight unit uw (slice ws, group wg)
    int qs = 1, qb = 0;
    if (qs == nsl) { qs = 0; qb = 1; plan_brick(1); }
    int ws = 0, wg = 2, uw = 2;
    int q = 0, u = 0;  // slice instance / unit being computed

    f32x16 acc[NT][2];

    for (int ord = 0; ord < nmine; ++ord) {
        int b, o0, o1, o2;
        brick_coords(ord, b, o0, o1, o2);
        // accumulators start from the bias (lane (r, hh): channels nt*32 + 8 j + 4 hh + (0..3) in registers 4 j .. 4 j + 3)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 bv = *reinterpret_cast<const float4*>(sBias + nt * 32 + 8 * j + 4 * hh);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    acc[nt][mt][4 * j] = bv.x; acc[nt][mt][4 * j + 1] = bv.y; acc[nt][mt][4 * j + 2] = bv.z; acc[nt][mt][4 * j + 3] = bv.w;
                }
            }

        for (int s = 0; s < nsl; ++s, ++q) {
            const unsigned char* bufA = sA + (q & 1) * RG_ABUF + hh * RG_APLANE;
#pragma unroll
            for (int g = 0; g < 3; ++g, ++u) {
                // ---- arrival of this unit's weights (and, at g = 0, of the slice's brick): counted waits, see header.
                // `after` = stores of the previous brick's epilogue sit between the awaited copies and the youngest ones
                const bool first = ord == 0 && s == 0;   // nothing but the prologue's copies is in flight: drain
                const bool after = s == 0;
                RG_T();  // unit top
                if (first) RG_VMCNT(0);
                else if (g == 0) { if (after) RG_VMCNT(WPW + NST); else RG_VMCNT(WPW); }
                else if (g == 1) { if (after) RG_VMCNT(WPW + RG_BPW + NST); else RG_VMCNT(WPW + RG_BPW); }
                else RG_VMCNT(WPW + RG_BPW);
                RG_T();  // this wave's copies have landed
                rg_barrier();
                RG_T();  // everybody's have

                const unsigned char* slotW = sW + (u % RG_WSLOTS) * WSLOT;
                const unsigned char* xa[2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) xa[mt] = bufA + (a_h[mt] + (g - 1) * RG_HY * RG_SZ) * 16;

                bf16x8 xf[2][2], wf[2][NT];
                auto read_frags = [&](int t9, int fb) {
                    const int toff = ((t9 / 3 - 1) * RG_SZ + (t9 % 3 - 1)) * 16;
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) xf[fb][mt] = *reinterpret_cast<const bf16x8*>(xa[mt] + toff);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) wf[fb][nt] = *reinterpret_cast<const bf16x8*>(slotW + t9 * (BN * 16) + b_off[nt]);
                };
                read_frags(0, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2 + NT, 0);
#pragma unroll
                for (int t9 = 0; t9 < 9; ++t9) {
                    if (t9 + 1 < 9) read_frags(t9 + 1, (t9 + 1) & 1);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt)
                            acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[t9 & 1][nt], xf[t9 & 1][mt], acc[nt][mt], 0, 0, 0);
                    if (t9 + 1 < 9) {
#pragma unroll
                        for (int k = 0; k < 2 * NT; ++k) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            if (k < 2 + NT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        }
                    } else {
                        __builtin_amdgcn_sched_group_barrier(0x008, 2 * NT, 0);
                    }
                    // the copies of the units ahead ride between the taps' MFMAs, ONE LDS-DMA instruction per tap (a burst
                    // of them stalls both waves of a SIMD in the address queue while the matrix pipe drains): weights
                    // of unit u + 2 behind taps 0 .. WPW - 1, the next slice's brick (g = 0 only) behind the following 5
                    if (t9 < WPW) issue_weight_piece(ws, wg, uw % RG_WSLOTS, t9);
                    if (t9 == WPW - 1) {
                        ++uw;
                        if (++wg == 3) { wg = 0; if (++ws == nsl) ws = 0; }
                    }
                    if (g == 0 && t9 >= WPW && t9 < WPW + RG_BPW) issue_brick_piece(qs, (q + 1) & 1, t9 - WPW);
                    if (g == 0 && t9 == 8) {
                        if (++qs == nsl) { qs = 0; ++qb; plan_brick(qb); }
                    }
                }
            }
        }

        // ---------------- epilogue of the brick.  Lane (r, hh) of wave w holds, for M tile mt, voxel
        // (w, 4 mt + (r & 3), r >> 2) and channels nt*32 + 8 j + 4 hh + (0..3) in accumulator registers 4 j .. 4 j + 3.
        RG_T();  // last MFMA issued
        rg_barrier();  // every wave is done with the last slice's brick buffer: it now holds the output tiles
        RG_T();
        unsigned char* tile = sA + ((q - 1) & 1) * RG_ABUF + wave * 4096;
        const int last_b = b;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int vw = (r & 3) * 8 + (r >> 2);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ch = nt * 32 + 8 * j + 4 * hh;
                    const unsigned lo2 = pack_bf16x2(acc[nt][mt][4 * j], acc[nt][mt][4 * j + 1]);
                    const unsigned hi2 = pack_bf16x2(acc[nt][mt][4 * j + 2], acc[nt][mt][4 * j + 3]);
                    *reinterpret_cast<uint2*>(tile + rg_tile_addr<BN>(vw, ch >> 3) + (ch & 7) * 2) = make_uint2(lo2, hi2);
                }
#pragma unroll
            for (int i = 0; i < 32 / VPI; ++i) {
                const int v = lane / CH + VPI * i, cidx = lane % CH;
                const int c0 = o0 + wave, c1 = o1 + 4 * mt + (v >> 3), c2 = o2 + (v & 7);
                if (c0 < A.X && c1 < A.Y && c2 < A.Z) {
                    uint4 val = *reinterpret_cast<const uint4*>(tile + rg_tile_addr<BN>(v, cidx));
                    const int64_t ov = (int64_t)b * V + c0 * YZ + c1 * A.Z + c2;
                    const int n = n0 + cidx * 8;
                    if (ZP) {
                        // data gradient: dx split over the two inputs of a concatenated conv, plus the gradient that
                        // arrives over the block's residual path
                        const bool lo1 = n < A.D1;
                        bf16* dst = lo1 ? A.d1 + ov * A.D1 + n : A.d2 + ov * (A.Cout - A.D1) + (n - A.D1);
                        const bf16* asrc = lo1 ? (A.a1 ? A.a1 + ov * A.D1 + n : nullptr)
                                               : (A.a2 ? A.a2 + ov * (A.Cout - A.D1) + (n - A.D1) : nullptr);
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
                        *reinterpret_cast<uint4*>(A.y + ov * A.Cout + n) = val;
                        if (A.gn_acc != nullptr) {
                            const unsigned wds[4] = {val.x, val.y, val.z, val.w};
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const f32x2 lh = {__uint_as_float(wds[e] << 16), __uint_as_float(wds[e] & 0xffff0000u)};
                                p1[e] += lh;
                                p2[e] = __builtin_elementwise_fma(lh, lh, p2[e]);
                            }
                        }
                    }
                }
            }
        }
        RG_T();  // epilogue stores issued
        if (!ZP && A.gn_acc != nullptr) {
            int nb = -1;
            if (ord + 1 < nmine) { int t0, t1, t2; brick_coords(ord + 1, nb, t0, t1, t2); }
            if (nb != last_b) flush_stats(last_b);
        }
    }
    RG_VMCNT(0);  // the dummy / wrapped copies of the last units must not outlive the workgroup's LDS
#ifdef RG_STAMPS
    if (lane == 0 && A.stamps != nullptr) {
        unsigned long long* rec = A.stamps + ((size_t)blockIdx.x * RG_WAVES + wave) * (RG_NSTAMP + 1);
        rec[0] = nst;
        for (int i = 0; i < RG_NSTAMP; ++i) rec[1 + i] = sStamp[wave * RG_NSTAMP + i];
    }
#endif
}

bool conv3_ring_supported(int C1, int C2, int Cout, int B, int X, int Y, int Z) {
    const char* env = getenv("TDX_CONV3_RING");  // A/B switch, read per call: 0 off, 1 auto (default), 2 whenever legal
    const int mode = env ? atoi(env) : 1;
    if (mode == 0 || !conv3_mfma_supported(C1, C2, Cout)) return false;
    // whole bricks only: the counted vmcnt waits assume that every wave issues all of its epilogue stores
    if ((X % 8) || (Y % 8) || (Z % 8)) return false;
    const int ntn = Cout % 64 == 0 ? Cout / 64 : Cout / 32;
    if (ntn > 32 || (32 % ntn) != 0) return false;
    if ((int64_t)X * Y * Z * 1024 >= (1ll << 31)) return false;  // 32-bit per-lane byte offsets
    if (mode == 2) return true;
    // persistent one-workgroup-per-CU launch: the grid must fill the 8 x 8 x 8 bricks and give every CU several of them
    const int64_t nb = (int64_t)B * ceil_div(X, 8) * ceil_div(Y, 8) * ceil_div(Z, 8);
    const double fill = (double)B * X * Y * Z / (512.0 * nb);
    return fill >= 0.9 && nb * ntn >= 3 * 256;
}

#ifdef RG_STAMPS
static unsigned long long* rg_stamp_buffer = nullptr;
#endif

template <int NT, bool ZP>
static int ring_go(const RingArgs& a, hipStream_t st) {
    constexpr int BN = NT * 32;
    constexpr int WPIECES = (9 * BN * 16 + 1023) / 1024;
    size_t lds = (size_t)2 * RG_ABUF + (size_t)RG_WSLOTS * 2 * WPIECES * 1024 + 1024 + BN * 4 + RG_WAVES * BN * 2 * 4;
#ifdef RG_STAMPS
    lds += (size_t)RG_WAVES * RG_NSTAMP * 8;
#endif
    auto kern = conv3_ring_kernel<NT, ZP>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(256), dim3(512), lds, st, a);
    return tdx_launch_status();
}

// forward (zero_pad = false: y, bias, gn_acc) or main term of the data gradient (zero_pad = true: d1 / d2 / a1 / a2);
// TDX_ESHAPE = not a case for this kernel, take the brick kernel
int conv3_ring_launch(const void* x1, int C1, const void* x2, int C2, const void* wp, const float* bias, void* y, int B, int X,
                      int Y, int Z, int Cout, bool zero_pad, hipStream_t st, double* gn_acc, void* d1, int D1, void* d2,
                      const void* a1, const void* a2) {
    if (!conv3_ring_supported(C1, C2, Cout, B, X, Y, Z)) return TDX_ESHAPE;
    if (zero_pad && (tdx_scratch_ptr() == nullptr || tdx_scratch_bytes() < 16)) return TDX_ESHAPE;  // zero source
    const int NT = Cout % 64 == 0 ? 2 : 1;
    RingArgs a;
    a.x1 = (const bf16*)x1; a.x2 = (const bf16*)x2; a.C1 = C1; a.C2 = C2;
    a.wp = (const bf16*)wp; a.bias = bias; a.y = (bf16*)y;
    a.B = B; a.X = X; a.Y = Y; a.Z = Z; a.Cout = Cout;
    a.nbx = ceil_div(X, 8); a.nby = ceil_div(Y, 8); a.nbz = ceil_div(Z, 8); a.ntn = Cout / (32 * NT);
    a.gn_acc = gn_acc;
    a.d1 = (bf16*)d1; a.d2 = (bf16*)d2; a.D1 = D1; a.a1 = (const bf16*)a1; a.a2 = (const bf16*)a2;
    a.zeros = tdx_scratch_ptr();
    a.stamps = nullptr;
#ifdef RG_STAMPS
    a.stamps = rg_stamp_buffer;
#endif
    if (NT == 2) return zero_pad ? ring_go<2, true>(a, st) : ring_go<2, false>(a, st);
    return zero_pad ? ring_go<1, true>(a, st) : ring_go<1, false>(a, st);
}

extern "C" int tdx_conv3_uses_ring(int C1, int C2, int Cout, int B, int X, int Y, int Z) {
    return conv3_ring_supported(C1, C2, Cout, B, X, Y, Z) ? 1 : 0;
}
