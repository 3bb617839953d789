// Persistent LDS-DMA "ring" form of the bf16 MFMA implicit-GEMM 3x3x3 convolution for gfx950: forward (replicate
// padding) and the main term of the data gradient (zero padding) on the two finest U-Net levels.
//
// Why (profiles/r10_conv3_stamps.txt: s_memtime stamps of the brick kernel tdx_conv3_mfma.hip, level-0 64 -> 64 layer):
// of a brick workgroup's 48 k cycles, 8.8 k are its prologue (nothing to do until the first slice arrives from HBM),
// 4 x 1.3 k are ds_write phases (global -> VGPR -> LDS), 3.1 k the epilogue, and its four MFMA phases last 6.9 k cycles
// each instead of the 3.5 k they take alone on the SIMDs, because the two co-resident workgroups fall into step: both
// in their MFMA phase, then both staging.  The matrix pipe idles half the time with no resource saturated.
//
// Structure:
//   * ONE persistent 8-wave workgroup per CU walks its own list of bricks.  A brick is 8 MT/2 x 8 x 8 voxels; wave w
//     owns MT/2 x planes = MT 32-voxel M tiles x NT 32-channel N tiles, MT x NT = 4 (64-wide output tiles: 8 x 8 x 8
//     bricks, 2 x 2; 32-wide ones: 16 x 8 x 8 bricks, 4 x 1, so that a weight fragment still feeds four MFMAs).  No
//     prologue per brick: the next brick's first operands are in flight while the current brick computes.
//   * K is walked in units of EIGHT input channels x ALL taps: one v_mfma_f32_32x32x16_bf16 contracts the 8 channels of
//     tap 2p (k = 0-7, lanes 0-31) and of tap 2p + 1 (k = 8-15, lanes 32-63), 14 steps per unit with a 28th tap whose
//     x operand is a zeroed LDS entry (+3.7 % MFMAs).  That makes a unit's operands small -- brick: one 16-B entry per
//     halo voxel (19 / 34 KiB), weights: 28 taps x BN x 16 B (28 / 14 KiB) -- so two brick buffers, a ring of three
//     weight slots AND dedicated output tiles fit the CU's 160 KiB, and a unit is 56 MFMAs per wave behind ONE
//     s_barrier.  (First version: 16 channels x 9 taps = 36-MFMA units.  Its stamps, profiles/r10_ring_stamps.txt,
//     showed that the two waves of a SIMD do not interleave their MFMAs: the older one wins every arbitration, runs its
//     unit alone at 40 cycles per MFMA, then waits ~1000 cycles at the barrier while the younger one does the same.
//     Hence longer units and fragment reads TWO K steps ahead, so that a wave alone keeps the pipe at 32 cycles per MFMA.)
//   * Staging is LDS-DMA only (global_load_lds_dwordx4: no staging registers, no ds_write phase), ONE instruction
//     behind each of the first K steps: the next unit's brick, then the weights of the unit after it.  Arrival is
//     tracked with counted s_waitcnt vmcnt(N): every wave issues the same number of DMA instructions per unit
//     (missing pieces are dummies into a scratch KiB), so N is a compile-time constant.
//   * Epilogue per brick, no barrier: accumulators (started from the bias) -> bf16 -> a wave-private LDS tile -> whole
//     voxel rows to HBM; GroupNorm moments are accumulated in registers ACROSS the bricks of a sample and flushed (LDS
//     reduce over the 8 waves + one f64 atomic per channel and moment) only when the sample changes.
//   * Workgroup -> bricks: every XCD owns one contiguous range of bricks (halo voxels shared through one L2); inside
//     an XCD the 32 workgroups interleave, and the N tiles of one brick run on neighbouring workgroups at the same time.
//
// Round 6: the operand format is a template argument (HF: bf16 or fp16 words in the same layouts, H16<HF> in tdx_common.h), and so
// is the brick depth (Z4: 8 MT x 8 x 4 bricks for grids like 48 x 16 x 12 -- see RingShape).
//
// Same products and the same fp32 accumulation as the brick kernel up to summation order (taps in pairs, bias first).
// Used when the grid's whole bricks fill the chip and leave at most 2 voxels per axis (conv3_ring_supported; those
// remainder slabs go to the thin-brick kernel in a second launch); everything else stays on the brick kernel.
#include "tdx_common.h"
#include "tdx_conv3.h"
#include <stdlib.h>
#include <algorithm>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define RG_HY 10
#define RG_SZ 12                        // padded z stride of the LDS brick image (conflict-free 16-B fragment reads)
#define RG_SZ4 6                        // z stride of the 4-deep brick's image: no padding needed (RingShape)
#define RG_WSLOTS 3                     // weight ring (lookahead 2 units)
#define RG_WAVES 8
#define RG_STEPS 14                     // K steps per unit: tap pairs (2p, 2p + 1); tap 27 is the zero dummy
#ifndef RG_DEFAULT_LOADERS
#define RG_DEFAULT_LOADERS 1            // loader waves by default? (TDX_RING_LOADERS overrides per call)
#endif

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
    unsigned keep;  // M0 is compiler-reserved: saved and restored inside the statement instead of a clobber the compiler does not honour
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds) : "memory");
}
// the same with a uniform 64-bit base and a 32-bit per-lane byte offset
__device__ __forceinline__ void rg_dma_off(const void* sbase, unsigned voff, unsigned lds) {
    lds = __builtin_amdgcn_readfirstlane(lds);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds) : "memory");
}
#define RG_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
__device__ __forceinline__ void rg_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's LDS stores are performed before the others proceed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// Diagnostic builds only (tools/micro/ring_stamp.hip defines RG_STAMPS): s_memtime stamps of bricks RG_STAMP_B0 .. +1
// of every wave, kept in LDS (the statistics scratch: such builds run without the statistics flush) and dumped to
// A.stamps at the end; the product build carries none of it.
#ifdef RG_STAMPS
#define RG_NSTAMP 64
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

// LW = 0: the 8 computing waves issue the copies themselves (one per K step).  LW = 4: four extra LOADER waves (one more
// wave per SIMD) issue every copy, wait for their arrival and meet the computing waves at the unit barrier; the
// computing waves then carry no vector-memory instruction in their MFMA stream (an LDS-DMA instruction costs the issuing
// wave ~150 cycles there: profiles/r10_ring_stamps.txt, ablations).
// Z4 (round 6): bricks 4 voxels deep along z, for grids whose z extent is a multiple of 4 but not of 8 and that have too few
// 8-deep bricks to fill a persistent launch -- level 2 of the benchmark grid, 48 x 16 x 12 (reference ddpm.py:358), which ran
// on the brick kernel at 0.26 of the MFMA peak.  An M tile is then a whole 8 (y) x 4 (z) plane of the brick, a wave owns MT
// planes, the brick is 8 MT x 8 x 4 (64-wide tiles: 16 x 8 x 4, 32-wide: 32 x 8 x 4: the same 512 voxels as before).  The
// image keeps the 8-deep layout's y stride of 12 entries -- the one pattern of 8 (y) x 4 (z) fragment lanes that
// tools/micro/lds_pattern_probe.hip measures conflict-free (y = r & 7, z = r >> 3 on entries 12 y + z; every layout with a y
// stride of 6 .. 11 costs +42 % per ds_read_b128 under any permutation of the lane bits) -- and puts TWO x planes into each
// 12-entry row (entries 0-5: even halo plane, 6-11: odd), so the image of an 18-plane brick stays 17 KiB and fits beside the
// weight ring: entry(hx, hy, hz) = (hx >> 1) 120 + 12 hy + 6 (hx & 1) + hz.  A tap's x step then depends on the parity of the
// fragment's plane (+6 / -114 from an even halo plane, +114 / -6 from an odd one): two tap-offset tables, picked per M tile at
// compile time (a wave's planes are w XP + mt with XP even).
template <int NT, int LW = 0, bool Z4 = false>
struct RingShape {
    static constexpr int IW = LW ? LW : 8;                  // waves that issue copies
    static constexpr int BN = NT * 32;
    static constexpr int MT = 4 / NT;                       // M tiles per wave
    static constexpr int XP = Z4 ? MT : MT / 2;             // x planes per wave
    static constexpr int BX = 8 * XP;                       // brick extent along x
    static constexpr int BZ = Z4 ? 4 : 8;                   // ... along z (y: 8)
    static constexpr int HZ = BZ + 2;
    static constexpr int SZ = Z4 ? RG_SZ4 : RG_SZ;          // z stride of the image
    static constexpr int ENT = Z4 ? (BX + 2) / 2 * RG_HY * RG_SZ   // (two x planes per 12-entry row)
                                  : (BX + 2) * RG_HY * SZ;         // LDS entries of a brick image (16 B = 8 channels of a voxel)
    static constexpr int APIECES = (ENT + 63) / 64;         // 1-KiB DMA pieces: 19 / 34
    static constexpr int ABUF = APIECES * 1024;
    static constexpr int BPW = (APIECES + IW - 1) / IW;     // brick pieces per issuing wave and unit: 3 / 5 (LW = 4: 5 / 9)
    static constexpr int WPIECES = 28 * BN * 16 / 1024;     // 28 / 14
    static constexpr int WSLOT = WPIECES * 1024;
    static constexpr int WPW = (WPIECES + IW - 1) / IW;     // weight pieces per issuing wave and unit: 4 / 2 (LW = 4: 7 / 4)
    static constexpr int TILE = 32 * BN * 2;                // wave-private output tile: 4 / 2 KiB
    static constexpr int CH = BN / 8;                       // 16-B chunks per output row
    static constexpr int VPI = 64 / CH;                     // voxels per store instruction
    static constexpr int NST = MT * (32 / VPI);             // global stores per wave and brick: 8 / 8
    static constexpr size_t LDS = (size_t)2 * ABUF + (size_t)RG_WSLOTS * WSLOT + (size_t)RG_WAVES * TILE + 1024 /* dummy sink */ +
                                  64 /* zero entry */ + BN * 4 /* bias */ + (size_t)RG_WAVES * BN * 2 * 4 /* statistics */;
};

template <int NT, bool ZP, int LW, bool HF, bool Z4>
__global__ void __launch_bounds__(512 + 64 * LW, LW ? 3 : 2) conv3_ring_kernel(RingArgs A) {
    typedef RingShape<NT, LW, Z4> S;
    typedef H16<HF> H;                  // operand format: bf16 or fp16 words (the pointers of RingArgs are raw 16-bit rows)
    typedef typename H::T HT;
    constexpr int BN = S::BN, MT = S::MT, XP = S::XP, BPW = S::BPW, WPW = S::WPW, CH = S::CH, VPI = S::VPI, NST = S::NST;
    static_assert(LW > 0 || BPW + WPW < RG_STEPS, "one DMA instruction per K step");
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char* sA = smem;                                   // [2][ABUF]
    unsigned char* sW = sA + 2 * S::ABUF;                       // [RG_WSLOTS][WSLOT]: [28 taps][BN] x 16 B
    unsigned char* sT = sW + RG_WSLOTS * S::WSLOT;              // [RG_WAVES][TILE]
    unsigned char* sD = sT + RG_WAVES * S::TILE;                // 1 KiB sink of the dummy pieces
    unsigned char* sZ = sD + 1024;                              // 64 B of zeros: x operand of the 28th tap
    float* sBias = reinterpret_cast<float*>(sZ + 64);           // [BN]
    float* sRed = sBias + BN;                                   // [RG_WAVES][BN][2]
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)smem;
    const unsigned ldsA = lds0, ldsW = lds0 + 2 * S::ABUF, ldsD = ldsW + RG_WSLOTS * S::WSLOT + RG_WAVES * S::TILE;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const bool loader = LW > 0 && wave >= RG_WAVES;
    const int iw = LW ? wave - RG_WAVES : wave;                 // index among the issuing waves
#ifdef RG_STAMPS
    // NT = 2: the statistics scratch is exactly 8 waves x 64 stamps; NT = 1: behind it (ring_go adds the bytes)
    unsigned long long* sStamp = reinterpret_cast<unsigned long long*>(NT == 2 ? sRed : sRed + RG_WAVES * BN * 2);
    int nst = 0;
#endif

    // ---- this workgroup's bricks: XCD x owns [lo, hi); its gridDim.x / 8 workgroups (32 on a full chip; fewer under
    // TDX_PERSISTENT_CUS) = KX brick lanes x ntn N tiles
    const int hw = blockIdx.x, xcd = hw & 7, slot = hw >> 3;
    const int ntile = slot % A.ntn, kx = slot / A.ntn, KX = (int)(gridDim.x >> 3) / A.ntn;
    const int nbricks = A.B * A.nbx * A.nby * A.nbz;
    const int lo = xcd * (nbricks >> 3) + min(xcd, nbricks & 7), hi = lo + (nbricks >> 3) + (xcd < (nbricks & 7) ? 1 : 0);
    const int nmine = hi - lo > kx ? (hi - lo - kx + KX - 1) / KX : 0;
    if (nmine == 0) return;
    const int n0 = ntile * BN;
    const int Cin = A.C1 + A.C2, nun = Cin >> 3;  // units per brick
    const int YZ = A.Y * A.Z, V = A.X * YZ;

    if (tid < BN) sBias[tid] = A.bias ? A.bias[n0 + tid] : 0.f;
    if (tid < 16) reinterpret_cast<unsigned*>(sZ)[tid] = 0u;
    rg_barrier();

    // ---- per-lane DMA geometry (fixed for the kernel)
    // brick piece i of this wave: pi = wave * BPW + i -> entries 64 pi .. 64 pi + 63 of the image
    int hxyz[BPW];
#pragma unroll
    for (int i = 0; i < BPW; ++i) {
        const int pi = min(max(iw, 0) * BPW + i, S::APIECES - 1);
        const int e = min(pi * 64 + lane, S::ENT - 1);
        int hx, hy, hz;
        if (Z4) {  // entry = (hx >> 1) 120 + 12 hy + 6 (hx & 1) + hz
            const int xp = e / (RG_HY * RG_SZ), rem = e - xp * (RG_HY * RG_SZ), q = rem % RG_SZ;
            hy = rem / RG_SZ; hx = 2 * xp + q / RG_SZ4; hz = q % RG_SZ4;
        } else {
            hx = e / (RG_HY * S::SZ);
            const int rem = e - hx * (RG_HY * S::SZ);
            hy = rem / S::SZ; hz = min(rem - hy * S::SZ, S::HZ - 1);
        }
        hxyz[i] = hx | (hy << 8) | (hz << 16);
    }
    // weight piece j of this wave: pj = wave * WPW + j -> entries 64 pj .. of [28 taps][BN]; tap 27 copies tap 26 (its x
    // operand is zero; the weights only have to be finite).  Byte offset inside the 16-channel slice of the packed operand
    unsigned wlane[WPW];
#pragma unroll
    for (int j = 0; j < WPW; ++j) {
        const int pj = min(max(iw, 0) * WPW + j, S::WPIECES - 1);
        const int e = pj * 64 + lane;
        wlane[j] = (unsigned)((min(e / BN, 26) * A.Cout + (e % BN)) * 32);
    }

    // ---- fragment geometry: M tile mt of wave w: x plane w XP + mt / 2, y = 4 (mt % 2) + (r & 3), z = r >> 2
    // (Z4: x plane w XP + mt, y = r & 7, z = r >> 3);
    // K step p: lanes 0-31 read tap 2p, lanes 32-63 tap 2p + 1 (the 28th: the zero entry)
    const int ly = Z4 ? (r & 7) : (r & 3), lz = Z4 ? (r >> 3) : (r >> 2);  // voxel of this lane inside an M tile
    int a_h[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int px = Z4 ? wave * XP + mt : wave * XP + mt / 2, py = Z4 ? ly : 4 * (mt % 2) + ly;
        const int hx = px + 1;
        a_h[mt] = Z4 ? ((hx >> 1) * (RG_HY * RG_SZ) + (py + 1) * RG_SZ + (hx & 1) * RG_SZ4 + (lz + 1)) * 16
                     : ((hx * RG_HY + (py + 1)) * S::SZ + (lz + 1)) * 16;
    }
    // tap offsets; Z4: [0] for fragments on an ODD halo plane (M tiles mt even: hx = w XP + mt + 1), [1] for an even one
    int xo[Z4 ? 2 : 1][RG_STEPS];
#pragma unroll
    for (int p = 0; p < RG_STEPS; ++p) {
        const int t = min(2 * p + hh, 26);
        const int dx = t / 9 - 1, dy = (t / 3) % 3 - 1, dz = t % 3 - 1;
        if (Z4) {
            const int step_odd = dx > 0 ? RG_HY * RG_SZ - RG_SZ4 : (dx < 0 ? -RG_SZ4 : 0);   // from an odd plane
            const int step_even = dx > 0 ? RG_SZ4 : (dx < 0 ? RG_SZ4 - RG_HY * RG_SZ : 0);   // from an even plane
            xo[0][p] = (step_odd + dy * RG_SZ + dz) * 16;
            xo[Z4 ? 1 : 0][p] = (step_even + dy * RG_SZ + dz) * 16;
        } else {
            xo[0][p] = ((dx * RG_HY + dy) * S::SZ + dz) * 16;
        }
    }
    int b_off[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) b_off[nt] = (hh * BN + nt * 32 + r) * 16;

    // ---- issue cursors
    int vox[BPW];          // source voxel (inside the sample, clamped; -1: zero) of this lane's pieces, for the planned brick
    int vb_b = 0;          // sample of the planned brick
    auto brick_coords = [&](int ord, int& b, int& o0, int& o1, int& o2) {
        int id = lo + kx + KX * ord;
        const int bz = id % A.nbz; id /= A.nbz;
        const int by = id % A.nby; id /= A.nby;
        const int bx = id % A.nbx; id /= A.nbx;
        b = id; o0 = bx * S::BX; o1 = by * 8; o2 = bz * S::BZ;
    };
    auto plan_brick = [&](int ord) {
        int b, o0, o1, o2;
        brick_coords(min(ord, nmine - 1), b, o0, o1, o2);
        vb_b = b;
#pragma unroll
        for (int i = 0; i < BPW; ++i) {
            int s0 = o0 + (hxyz[i] & 0xff) - 1, s1 = o1 + ((hxyz[i] >> 8) & 0xff) - 1, s2 = o2 + ((hxyz[i] >> 16) & 0xff) - 1;
            bool ok = true;
            if (ZP) ok = s0 >= 0 && s0 < A.X && s1 >= 0 && s1 < A.Y && s2 >= 0 && s2 < A.Z;
            else { s0 = min(max(s0, 0), A.X - 1); s1 = min(max(s1, 0), A.Y - 1); s2 = min(max(s2, 0), A.Z - 1); }
            vox[i] = ok ? (s0 * A.Y + s1) * A.Z + s2 : -1;
        }
    };
    // piece i of the planned brick's 8-channel slice c8 -> buffer `buf` (one LDS-DMA instruction)
    auto issue_brick_piece = [&](int c8, int buf, int i) {
        const int k0 = c8 << 3;
        const bf16* xs; int Cs, kk;
        if (k0 < A.C1) { xs = A.x1; Cs = A.C1; kk = k0; } else { xs = A.x2; Cs = A.C2; kk = k0 - A.C1; }
        const bf16* base = xs + ((int64_t)vb_b * V) * Cs + kk;  // uniform
        const int pi = iw * BPW + i;
        const unsigned dst = pi < S::APIECES ? ldsA + buf * S::ABUF + pi * 1024 : ldsD;
        if (ZP) {
            const bf16* src = vox[i] >= 0 ? base + (int64_t)vox[i] * Cs : reinterpret_cast<const bf16*>(A.zeros);
            rg_dma(src, dst);
        } else {
            rg_dma_off(base, (unsigned)(vox[i] * Cs) * 2u, dst);
        }
    };
    // piece j of the weights of 8-channel slice c8 -> ring slot
    auto issue_weight_piece = [&](int c8, int wslot, int j) {
        const bf16* base = A.wp + ((int64_t)((c8 >> 1) * 27) * A.Cout + n0) * 16 + (c8 & 1) * 8;  // uniform
        const int pj = iw * WPW + j;
        const unsigned dst = pj < S::WPIECES ? ldsW + wslot * S::WSLOT + pj * 1024 : ldsD;
        rg_dma_off(base, wlane[j], dst);
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

    // ---- prologue (issuing waves): brick 0 / slice 0 and the weights of units 0 and 1
    // cursors of what is issued next: slice qs of the planned brick qb; weights of unit uw (slice ws)
    int qs = 1, qb = 0, ws = 2 % nun, uw = 2;
    int u = 0;  // unit being computed
    if (LW == 0 || loader) {
        plan_brick(0);
#pragma unroll
        for (int i = 0; i < BPW; ++i) issue_brick_piece(0, 0, i);
#pragma unroll
        for (int j = 0; j < WPW; ++j) issue_weight_piece(0, 0, j);
#pragma unroll
        for (int j = 0; j < WPW; ++j) issue_weight_piece(1 % nun, 1, j);
        if (qs == nun) { qs = 0; qb = 1; plan_brick(1); }
    }
    if (LW > 0 && loader) {
        // ---- loader waves: per unit, wait for the unit's operands (the only younger copies are the next unit's weights),
        // meet the computing waves, then issue the next unit's brick and the weights of the unit after it.  They join
        // every barrier the computing waves execute (the two of a statistics flush included).
        for (int ord = 0; ord < nmine; ++ord) {
            for (int c8 = 0; c8 < nun; ++c8, ++u) {
                RG_VMCNT(WPW);
                rg_barrier();
#pragma unroll
                for (int i = 0; i < BPW; ++i) issue_brick_piece(qs, (u + 1) & 1, i);
#pragma unroll
                for (int j = 0; j < WPW; ++j) issue_weight_piece(ws, uw % RG_WSLOTS, j);
                ++uw;
                if (++ws == nun) ws = 0;
                if (++qs == nun) { qs = 0; ++qb; plan_brick(qb); }
            }
#ifndef RG_STAMPS
            if (!ZP && A.gn_acc != nullptr) {
                int b, nb = -1, t0, t1, t2;
                brick_coords(ord, b, t0, t1, t2);
                if (ord + 1 < nmine) brick_coords(ord + 1, nb, t0, t1, t2);
                if (nb != b) { rg_barrier(); rg_barrier(); }
            }
#endif
        }
        RG_VMCNT(0);  // the wrapped copies of the last units must not outlive the workgroup's LDS
        return;
    }
    bool stores_behind = false;  // the previous brick's epilogue stores were issued after this unit's operands

    f32x16 acc[NT][MT];

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
                for (int mt = 0; mt < MT; ++mt) {
                    acc[nt][mt][4 * j] = bv.x; acc[nt][mt][4 * j + 1] = bv.y; acc[nt][mt][4 * j + 2] = bv.z; acc[nt][mt][4 * j + 3] = bv.w;
                }
            }

        for (int c8 = 0; c8 < nun; ++c8, ++u) {
            // ---- arrival of this unit's brick (issued one unit ago) and weights (two units ago): the only younger
            // copies are the weights of the next unit; the stores of a brick's epilogue sit behind its successor's
            // first unit's operands
            RG_T();  // unit top
            if (LW == 0) { if (stores_behind) RG_VMCNT(WPW + NST); else RG_VMCNT(WPW); }
            stores_behind = false;
            RG_T();  // this wave's copies have landed
            rg_barrier();
            RG_T();  // everybody's have

            const unsigned char* bufA = sA + (u & 1) * S::ABUF;
            const unsigned char* slotW = sW + (u % RG_WSLOTS) * S::WSLOT;
            const unsigned char* xa[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) xa[mt] = bufA + a_h[mt];

            // fragments are read TWO K steps ahead of the MFMAs that use them (three register sets): one wave alone has
            // to keep the matrix pipe busy -- the two waves of a SIMD take turns, they do not interleave
            bf16x8 xf[3][MT], wf[3][NT];
            auto read_frags = [&](int p, int fb) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const unsigned char* src = xa[mt] + xo[Z4 ? (mt & 1) : 0][p];
                    if (p == RG_STEPS - 1) src = hh ? sZ : src;  // the 28th tap multiplies zeros
                    xf[fb][mt] = *reinterpret_cast<const bf16x8*>(src);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) wf[fb][nt] = *reinterpret_cast<const bf16x8*>(slotW + p * (2 * BN * 16) + b_off[nt]);
            };
            read_frags(0, 0);
            read_frags(1, 1);
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * (MT + NT), 0);
#pragma unroll
            for (int p = 0; p < RG_STEPS; ++p) {
#if defined(RG_ABL) && (RG_ABL & 2)     // (ablation: no fragment reads after the first two steps)
                if (false)
#endif
                if (p + 2 < RG_STEPS) read_frags(p + 2, (p + 2) % 3);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        acc[nt][mt] = H::mfma(wf[p % 3][nt], xf[p % 3][mt], acc[nt][mt]);
                if (p + 2 < RG_STEPS) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (k < MT + NT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                } else {
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                }
                // the copies of the units ahead ride between the K steps, ONE LDS-DMA instruction per step (a burst of
                // them stalls the wave in the address queue while the matrix pipe drains): the next unit's brick behind
                // steps 0 .. BPW - 1, the weights of the unit after it behind the following WPW
#if !defined(RG_ABL) || !(RG_ABL & 1)  // (diagnostic ablation builds drop the copies: timing only)
                if (LW == 0 && p < BPW) issue_brick_piece(qs, (u + 1) & 1, p);
                if (LW == 0 && p >= BPW && p < BPW + WPW) issue_weight_piece(ws, uw % RG_WSLOTS, p - BPW);
#endif
                if (LW == 0 && p == BPW + WPW) {
                    ++uw;
                    if (++ws == nun) ws = 0;
                    if (++qs == nun) { qs = 0; ++qb; plan_brick(qb); }
                }
            }
        }

        // ---------------- epilogue of the brick (no barrier: wave-private tiles).  Lane (r, hh) of wave w holds, for
        // M tile mt, voxel (w XP + mt / 2, 4 (mt % 2) + (r & 3), r >> 2) (Z4: (w XP + mt, ly, lz)) and channels
        // nt*32 + 8 j + 4 hh + (0..3) in accumulator registers 4 j .. 4 j + 3.  Tile row = voxel in (y, z) order.
        RG_T();  // last MFMA issued
        unsigned char* tile = sT + wave * S::TILE;
        const int last_b = b;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int vw = ly * S::BZ + lz;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ch = nt * 32 + 8 * j + 4 * hh;
                    const unsigned lo2 = H::pack2(acc[nt][mt][4 * j], acc[nt][mt][4 * j + 1]);
                    const unsigned hi2 = H::pack2(acc[nt][mt][4 * j + 2], acc[nt][mt][4 * j + 3]);
                    *reinterpret_cast<uint2*>(tile + rg_tile_addr<BN>(vw, ch >> 3) + (ch & 7) * 2) = make_uint2(lo2, hi2);
                }
#pragma unroll
            for (int i = 0; i < 32 / VPI; ++i) {
                const int v = lane / CH + VPI * i, cidx = lane % CH;
                const int c0 = o0 + (Z4 ? wave * XP + mt : wave * XP + mt / 2);
                const int c1 = o1 + (Z4 ? (v >> 2) : 4 * (mt % 2) + (v >> 3)), c2 = o2 + (Z4 ? (v & 3) : (v & 7));
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
                        Vec8<HT> va, vb;
                        va.load(reinterpret_cast<const HT*>(&val));
                        vb.load(reinterpret_cast<const HT*>(asrc));
#pragma unroll
                        for (int e = 0; e < 8; ++e) va.v[e] += vb.v[e];
                        va.store(reinterpret_cast<HT*>(dst));
                    } else {
                        *reinterpret_cast<uint4*>(dst) = val;
                    }
                } else {
                    *reinterpret_cast<uint4*>(A.y + ov * A.Cout + n) = val;
                    if (A.gn_acc != nullptr) {
                        const unsigned wds[4] = {val.x, val.y, val.z, val.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const f32x2 lh = {H::lo(wds[e]), H::hi(wds[e])};
                            p1[e] += lh;
                            p2[e] = __builtin_elementwise_fma(lh, lh, p2[e]);
                        }
                    }
                }
            }
        }
        stores_behind = true;
        RG_T();  // epilogue stores issued
#ifndef RG_STAMPS
        if (!ZP && A.gn_acc != nullptr) {
            int nb = -1;
            if (ord + 1 < nmine) { int t0, t1, t2; brick_coords(ord + 1, nb, t0, t1, t2); }
            if (nb != last_b) flush_stats(last_b);
        }
#endif
    }
    if (LW == 0) RG_VMCNT(0);  // the wrapped copies of the last units must not outlive the workgroup's LDS
#ifdef RG_STAMPS
    if (lane == 0 && A.stamps != nullptr) {
        unsigned long long* rec = A.stamps + ((size_t)blockIdx.x * RG_WAVES + wave) * (RG_NSTAMP + 1);
        rec[0] = nst;
        for (int i = 0; i < RG_NSTAMP; ++i) rec[1 + i] = sStamp[wave * RG_NSTAMP + i];
    }
#endif
}

static bool ring_z8_supported(int C1, int C2, int Cout, int B, int X, int Y, int Z, int mode);
static bool ring_z4_supported(int C1, int C2, int Cout, int B, int X, int Y, int Z, int mode) {
    const char* env = getenv("TDX_RING_Z4");  // A/B switch, read per call: 0 = no 4-deep bricks (round 5's dispatch), 2 = prefer them
    if (env && atoi(env) == 0) return false;
    const int NT = Cout % 64 == 0 ? 2 : 1, bx = NT == 2 ? 16 : 32;
    if (X < bx || Y < 8 || Z < 4 || (X % bx) > 2 || (Y % 8) > 2 || (Z % 4) > 2) return false;
    const int ntn = Cout / (32 * NT);
    if (ntn > tdx_persistent_cus() / 8) return false;
    if ((int64_t)X * Y * Z * 1024 >= (1ll << 31)) return false;
    if (mode == 2) return true;
    if (NT == 1 && std::max(C1, C2) > 32) return false;  // (as for the 8-deep bricks)
    // every CU should get work: (brick, N tile) pairs >= 3/4 of the CUs; a pair walks all of K (>= 16 units at level 2)
    const int64_t nb = (int64_t)B * (X / bx) * (Y / 8) * (Z / 4);
    return nb * ntn >= 192 && (C1 + C2) >= 64;
}
// brick shape of the ring kernel for this call: 0 = not a case for it, 8 = 8-deep bricks, 4 = 4-deep bricks (Z4)
static int ring_brick_depth(int C1, int C2, int Cout, int B, int X, int Y, int Z) {
    const char* env = getenv("TDX_CONV3_RING");  // A/B switch, read per call: 0 off, 1 auto (default), 2 whenever legal
    const int mode = env ? atoi(env) : 1;
    if (mode == 0 || !conv3_mfma_supported(C1, C2, Cout)) return 0;
    const char* z4 = getenv("TDX_RING_Z4");  // 2: 4-deep bricks wherever they are legal (tests)
    if (z4 && atoi(z4) == 2 && ring_z4_supported(C1, C2, Cout, B, X, Y, Z, 2)) return 4;
    if (ring_z8_supported(C1, C2, Cout, B, X, Y, Z, mode)) return 8;
    return ring_z4_supported(C1, C2, Cout, B, X, Y, Z, mode) ? 4 : 0;
}
bool conv3_ring_supported(int C1, int C2, int Cout, int B, int X, int Y, int Z) {
    return ring_brick_depth(C1, C2, Cout, B, X, Y, Z) != 0;
}
static bool ring_z8_supported(int C1, int C2, int Cout, int B, int X, int Y, int Z, int mode) {
    const int NT = Cout % 64 == 0 ? 2 : 1, bx = NT == 2 ? 8 : 16;
    // The kernel walks WHOLE bricks (every wave issues all of its epilogue stores: the counted vmcnt waits of the
    // loader-less form assume it).  A grid may leave 1-2 voxels per axis beyond its whole bricks: those remainder slabs go
    // to the thin-brick kernel of tdx_conv3_mfma.hip in a second launch (conv3_ring_launch).  Larger remainders (level 2 of
    // the benchmark grid, 48 x 16 x 12) stay on the brick kernels altogether.
    if (X < bx || Y < 8 || Z < 8 || (X % bx) > 2 || (Y % 8) > 2 || (Z % 8) > 2) return false;
    const char* rag = getenv("TDX_RING_RAGGED");  // A/B switch: 0 = whole-brick grids only (round 3's rule)
    if (rag && atoi(rag) == 0 && ((X % bx) || (Y % 8) || (Z % 8))) return false;
    const int ntn = Cout / (32 * NT);
    if (ntn > tdx_persistent_cus() / 8) return false;  // an XCD's workgroups = brick lanes x N tiles
    if ((int64_t)X * Y * Z * 1024 >= (1ll << 31)) return false;  // 32-bit per-lane byte offsets
    if (mode == 2) return true;
    // 32-wide output tiles: every 8-channel unit re-fetches a 16-B piece of each halo voxel's row, one cache line per
    // lane; with rows of 128 B and more (C >= 64 per input tensor) that address traffic arrives late (128 -> 32 at
    // 192 x 64 x 48: 1.07 ms against the brick kernel's 0.85; 32 -> 32: 0.212 against 0.227)
    if (NT == 1 && std::max(C1, C2) > 32) return false;
    // persistent one-workgroup-per-CU launch: every CU should get several bricks
    const int64_t nb = (int64_t)B * (X / bx) * (Y / 8) * (Z / 8);
    const char* mi = getenv("TDX_RING_MIN_ITEMS");  // A/B knob, read per call: (brick, N tile) pairs a launch must have
    return nb * ntn >= (mi ? atoi(mi) : 3 * 256);
}

#ifdef RG_STAMPS
static unsigned long long* rg_stamp_buffer = nullptr;
#endif

template <int NT, bool ZP, int LW, bool HF, bool Z4 = false>
static int ring_go(const RingArgs& a, hipStream_t st) {
    size_t lds = RingShape<NT, LW, Z4>::LDS;
#ifdef RG_STAMPS
    if (NT == 1) lds += (size_t)RG_WAVES * RG_NSTAMP * 8;
#endif
    auto kern = conv3_ring_kernel<NT, ZP, LW, HF, Z4>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    // one workgroup per CU on (up to) TDX_PERSISTENT_CUS CUs, the same number on every XCD, a multiple of the N tiles
    const int px = tdx_persistent_cus() / 8, grid = 8 * (px - px % a.ntn);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512 + 64 * LW), lds, st, a);
    return tdx_launch_status();
}

// forward (zero_pad = false: y, bias, gn_acc) or main term of the data gradient (zero_pad = true: d1 / d2 / a1 / a2);
// TDX_ESHAPE = not a case for this kernel, take the brick kernel
int conv3_ring_launch(const void* x1, int C1, const void* x2, int C2, const void* wp, const float* bias, void* y, int B, int X,
                      int Y, int Z, int Cout, bool zero_pad, hipStream_t st, double* gn_acc, void* d1, int D1, void* d2,
                      const void* a1, const void* a2, bool hf) {
    const int depth = ring_brick_depth(C1, C2, Cout, B, X, Y, Z);
    if (depth == 0) return TDX_ESHAPE;
    if (zero_pad && (tdx_scratch_ptr() == nullptr || tdx_scratch_bytes() < 16)) return TDX_ESHAPE;  // zero source
    const int NT = Cout % 64 == 0 ? 2 : 1;
    const int bxr = (NT == 2 ? 8 : 16) * (depth == 4 ? 2 : 1);  // brick extent along x
    RingArgs a;
    a.x1 = (const bf16*)x1; a.x2 = (const bf16*)x2; a.C1 = C1; a.C2 = C2;
    a.wp = (const bf16*)wp; a.bias = bias; a.y = (bf16*)y;
    a.B = B; a.X = X; a.Y = Y; a.Z = Z; a.Cout = Cout;
    a.nbx = X / bxr; a.nby = Y / 8; a.nbz = Z / depth; a.ntn = Cout / (32 * NT);
    a.gn_acc = gn_acc;
    a.d1 = (bf16*)d1; a.d2 = (bf16*)d2; a.D1 = D1; a.a1 = (const bf16*)a1; a.a2 = (const bf16*)a2;
    a.zeros = tdx_scratch_ptr();
    a.stamps = nullptr;
#ifdef RG_STAMPS
    a.stamps = rg_stamp_buffer;
#endif
    const char* env = getenv("TDX_RING_LOADERS");  // A/B switch: 0 = the computing waves issue the copies themselves
    const bool lw = env ? atoi(env) != 0 : RG_DEFAULT_LOADERS;
    int rc;
    if (depth == 4) {  // 4-deep bricks: the loader-wave form only
#define RG_Z4(NTV) (hf ? (zero_pad ? ring_go<NTV, true, 4, true, true>(a, st) : ring_go<NTV, false, 4, true, true>(a, st)) \
                       : (zero_pad ? ring_go<NTV, true, 4, false, true>(a, st) : ring_go<NTV, false, 4, false, true>(a, st)))
        rc = NT == 2 ? RG_Z4(2) : RG_Z4(1);
#undef RG_Z4
    } else if (hf) {  // fp16 operands: the loader-wave form only (the loader-less form is an A/B build of the bf16 kernels)
        if (NT == 2) rc = zero_pad ? ring_go<2, true, 4, true>(a, st) : ring_go<2, false, 4, true>(a, st);
        else rc = zero_pad ? ring_go<1, true, 4, true>(a, st) : ring_go<1, false, 4, true>(a, st);
    } else if (NT == 2) {
        if (lw) rc = zero_pad ? ring_go<2, true, 4, false>(a, st) : ring_go<2, false, 4, false>(a, st);
        else rc = zero_pad ? ring_go<2, true, 0, false>(a, st) : ring_go<2, false, 0, false>(a, st);
    } else {
        if (lw) rc = zero_pad ? ring_go<1, true, 4, false>(a, st) : ring_go<1, false, 4, false>(a, st);
        else rc = zero_pad ? ring_go<1, true, 0, false>(a, st) : ring_go<1, false, 0, false>(a, st);
    }
    if (rc != TDX_OK || ((X % bxr) == 0 && (Y % 8) == 0 && (Z % depth) == 0)) return rc;
    // the 1-2 voxel remainder slabs beyond the whole bricks: thin 2 x 16 x 8 bricks, all slabs in one launch (same
    // operands, same epilogue incl. the statistics accumulators and the data gradient's split / addends)
    const int beyond[3] = {X - X % bxr, Y - Y % 8, Z - Z % depth};
    const Conv3Geom g = {B, X, Y, Z, X, Y, Z, 0};
    return conv3_mfma_launch(x1, C1, x2, C2, wp, bias, y, g, Cout, zero_pad, st, gn_acc, d1, D1, d2, a1, a2, nullptr, beyond, hf);
}

extern "C" int tdx_conv3_uses_ring(int C1, int C2, int Cout, int B, int X, int Y, int Z) {
    return conv3_ring_supported(C1, C2, Cout, B, X, Y, Z) ? 1 : 0;
}
extern "C" int tdx_conv3_ring_brick_depth(int C1, int C2, int Cout, int B, int X, int Y, int Z) {
    return ring_brick_depth(C1, C2, Cout, B, X, Y, Z);
}
