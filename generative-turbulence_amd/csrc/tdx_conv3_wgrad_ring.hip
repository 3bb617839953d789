// bf16 MFMA weight gradient of the replicate-padded 3x3x3 convolution, producer / consumer form (gfx950): 8 computing
// waves + 4 loader waves per workgroup, one workgroup per CU.  Output tiles 64 wide (NTN = 2; described below) or 32 wide
// (NTN = 1, Cout % 64 != 0: the level-0 layers of the benchmark model).
//
//   dW[tap][ci][co] = sum_v x[clamp(v + tap)][ci] * dy[v][co]
//
// The brick kernel (tdx_conv3_wgrad_mfma.hip) gives one wave per SIMD all 7 of its taps x 2 N tiles (224 accumulator
// registers) and lets that wave do everything: global -> VGPR -> LDS staging, 4 vector-ALU instructions of address
// arithmetic per MFMA, two barriers per brick.  With nothing else resident on the SIMD every one of those instructions
// is time the matrix pipe idles: MFMA busy 45 % (profiles/r09bf16_summary.md).  A DMA-staged copy of that kernel lost
// (an LDS-DMA instruction blocks its issuing wave for ~140 cycles: DESIGN 3.3).  This kernel cuts the work differently,
// the way tdx_conv3_ring.hip does for the forward:
//   * the workgroup's 32 (ci) x 64 (co) x 27-tap tile is 54 accumulator tiles of 32 x 32; the 8 COMPUTING waves own
//     7, 7, 7, 7, 7, 7, 6, 6 consecutive (tap, N tile) pairs = 112 accumulator registers, so that two of them fit a SIMD
//     next to a loader wave (<= 168 registers per lane) and take turns on the matrix pipe: 14 / 14 / 13 / 13 MFMAs per
//     SIMD and K step, as before.  A wave's pairs span 4 taps: 4 x-fragment reads + 2 dy-fragment reads per K step for
//     7 MFMAs; the x fragments of step s + 1 are read into the register set of the tap whose MFMAs have just been
//     issued (one step = ~230 cycles of cover), the dy fragments are double-buffered by step parity.
//   * the 4 LOADER waves stage brick i + 1 by LDS-DMA (global_load_lds_dwordx4, 18 instructions each) into the other
//     buffer pair while brick i computes, wait for arrival (s_waitcnt vmcnt(0)) and meet the computing waves at the ONE
//     barrier per brick.  The computing waves' instruction stream is MFMAs and fragment reads only.
//   * the bias gradient comes from the matrix pipe: waves 6 and 7 own only 6 pairs, their seventh slot multiplies an
//     all-ones x fragment with the dy fragment of N tile 0 / 1: every row of that tile is sum_v dy[v][co].
//
// Zero rows (dy rows of voxels outside a ragged brick; the missing channels of a half-filled last ci tile) are copied
// from the zero block at the head of the scratch arena (tdx_set_scratch).  Same per-workgroup sums in the same order as
// the brick kernel (K steps in brick order).
//
// NTN = 1 (round 5): the 32 x 32 x 27-tap tile is 27 accumulator tiles, 4 / 4 / 4 / 3 / 3 / 3 / 3 / 3 taps per computing wave
// (4 + 3 MFMAs per SIMD and K step), wave 7's spare slot sums the bias gradient; one dy plane per buffer.  Every x fragment
// feeds ONE MFMA here, so a K step reads 10 fragments per wave for 3-4 MFMAs (LDS array ~70 % busy at full MFMA rate): it
// gains less than the 64-wide form did: 32 -> 32 at 192 x 64 x 48 x 6: 0.255 -> 0.218 ms, 128 -> 32: 0.90 -> 0.82 ms,
// training step -0.22 ms (TDX_WGRAD_RING=2 restores the brick kernel for these layers).
#include "tdx_common.h"
#include "tdx_conv3.h"
#include <stdlib.h>
#include <algorithm>
#include <type_traits>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

#define WR_BX 4
#define WR_BY 8
#define WR_BZ 8
#define WR_HY 10
#define WR_HZ 10
#define WR_NVOX 256
#define WR_NSTEPS 16                                  // K steps of 16 voxels
#define WR_NHALO ((WR_BX + 2) * WR_HY * WR_HZ)         // 600 halo'd voxels, 64-B rows (32 channels)
#define WR_XPIECES ((WR_NHALO * 4 + 63) / 64)          // 38 DMA pieces of 1 KiB
#define WR_XBUF (WR_XPIECES * 1024)
#define WR_XPW ((WR_XPIECES + 3) / 4)                  // 10 per loader wave
#define WR_GPLANE (WR_NVOX * 64)                       // one 32-channel dy plane: 16 pieces
#define WR_CW 8                                        // computing waves
// NTN = N tiles of 32 output channels per workgroup (2: 64-wide, 1: 32-wide): dy planes per buffer, dy pieces per loader
// wave, accumulator tiles per computing wave
#define WR_GBUF(NTN) ((NTN) * WR_GPLANE)
#define WR_GPW(NTN) (4 * (NTN))
#define WR_SLOTS(NTN) ((NTN) == 2 ? 7 : 4)

struct WgradRingView {
    int B;
    int E[3];     // extents in the kernel's local axes (brick 4 x 8 x 8)
    int s[3];     // voxel strides
    int ws[3];    // weight-tap strides: global tap = sum_k (e_k + 1) * ws[k]
    int nb[3];    // bricks per axis
    int batch;    // voxels per sample
};

__device__ __forceinline__ bf16x8 wr_tr_frag(const unsigned char* lo, const unsigned char* hi) {
    s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lo));
    s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(hi));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 r = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, r);
}

// one LDS-DMA instruction (inline assembly: see tdx_conv3_ring.hip)
__device__ __forceinline__ void wr_dma(const void* gsrc, unsigned lds) {
    lds = __builtin_amdgcn_readfirstlane(lds);
    unsigned keep;  // M0 is compiler-reserved: saved and restored inside the statement (no "m0" clobber)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds) : "memory");
}
__device__ __forceinline__ void wr_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// HF: operand format (H16<HF>: bf16 or fp16 words behind the bf16-typed pointers)
template <int NTN, bool HF>
__global__ void __launch_bounds__(768, 3)
conv3_wgrad_ring_kernel(const bf16* __restrict__ x1, int C1, const bf16* __restrict__ x2, int C2, const bf16* __restrict__ dy,
                        float* __restrict__ dwp, float* __restrict__ dbias, WgradRingView gv, int Cout, int nsplit, int n_ci_tiles,
                        int64_t slab_stride, const void* __restrict__ zeros) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char* sX = smem;                        // [2][WR_XBUF]
    unsigned char* sG = smem + 2 * WR_XBUF;           // [2][WR_GBUF(NTN)]
    constexpr int GBUF = WR_GBUF(NTN), GPW = WR_GPW(NTN), SLOTS = WR_SLOTS(NTN);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Cin = C1 + C2;
    const int tile = blockIdx.x / nsplit, split = blockIdx.x - tile * nsplit;
    const int ci0 = (tile % n_ci_tiles) * 32, co0 = (tile / n_ci_tiles) * (32 * NTN);
    const int nbricks = gv.B * gv.nb[0] * gv.nb[1] * gv.nb[2];

    if (wave >= WR_CW) {
        // =========================================================== loader waves
        const int lw = wave - WR_CW;
        const bf16* xs;
        int Cs, cbase;
        if (ci0 < C1) { xs = x1; Cs = C1; cbase = ci0; } else { xs = x2; Cs = C2; cbase = ci0 - C1; }
        // x piece i of this wave: chunks pc = 64 (lw XPW + i) + lane = (halo voxel, 16-B quarter of its 64-B row); the two
        // slots beyond the image re-copy its last piece (same bytes to the same place)
        int xh[WR_XPW];  // hx | hy << 8 | hz << 16 | quarter << 24 | channels exist << 30
#pragma unroll
        for (int i = 0; i < WR_XPW; ++i) {
            const int pc = min(min(lw * WR_XPW + i, WR_XPIECES - 1) * 64 + lane, WR_NHALO * 4 - 1);
            const int hv = pc >> 2, q4 = pc & 3;
            const int hx = hv / (WR_HY * WR_HZ), rem = hv - hx * (WR_HY * WR_HZ);
            const int hy = rem / WR_HZ, hz = rem - hy * WR_HZ;
            xh[i] = hx | (hy << 8) | (hz << 16) | (q4 << 24) | ((cbase + q4 * 8 < Cs) ? (1 << 30) : 0);
        }
        // dy piece j of this wave: gp = lw * GPW + j -> plane gp / 16, chunks e = (gp % 16) * 64 + lane = (voxel e >> 2, quarter e & 3)
        const int g_plane = (lw * GPW) / 16, g_p0 = (lw * GPW) % 16;
        const int g_e0 = g_p0 * 64 + lane;
        auto issue = [&](int brick, int buf) {
            int bb = brick;
            const int bz = bb % gv.nb[2]; bb /= gv.nb[2];
            const int by = bb % gv.nb[1]; bb /= gv.nb[1];
            const int bx = bb % gv.nb[0]; bb /= gv.nb[0];
#pragma unroll
            for (int i = 0; i < WR_XPW; ++i) {
                const int sx = min(max(bx * WR_BX + (xh[i] & 0xff) - 1, 0), gv.E[0] - 1);
                const int sy = min(max(by * WR_BY + ((xh[i] >> 8) & 0xff) - 1, 0), gv.E[1] - 1);
                const int sz = min(max(bz * WR_BZ + ((xh[i] >> 16) & 0xff) - 1, 0), gv.E[2] - 1);
                const int64_t vox = (int64_t)bb * gv.batch + sx * gv.s[0] + sy * gv.s[1] + sz * gv.s[2];
                const bf16* src = (xh[i] >> 30) & 1 ? xs + vox * Cs + cbase + ((xh[i] >> 24) & 3) * 8 : reinterpret_cast<const bf16*>(zeros);
                wr_dma(src, lds0 + buf * WR_XBUF + min(lw * WR_XPW + i, WR_XPIECES - 1) * 1024);
            }
#pragma unroll
            for (int j = 0; j < GPW; ++j) {
                const int e = g_e0 + 64 * j, v = e >> 2, c4 = e & 3;
                const int vx = bx * WR_BX + (v >> 6), vy = by * WR_BY + ((v >> 3) & 7), vz = bz * WR_BZ + (v & 7);
                const bool ok = vx < gv.E[0] && vy < gv.E[1] && vz < gv.E[2];
                const int64_t vox = (int64_t)bb * gv.batch + vx * gv.s[0] + vy * gv.s[1] + vz * gv.s[2];
                const bf16* src = ok ? dy + vox * Cout + co0 + g_plane * 32 + c4 * 8 : reinterpret_cast<const bf16*>(zeros);
                wr_dma(src, lds0 + 2 * WR_XBUF + buf * GBUF + g_plane * WR_GPLANE + (g_p0 + j) * 1024);
            }
        };
        int brick = split, it = 0;
        if (brick < nbricks) issue(brick, 0);
        for (; brick < nbricks; brick += nsplit, ++it) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // brick `it` has landed
            wr_barrier();                                     // ... and the computing waves are done with brick it - 1
            if (brick + nsplit < nbricks) issue(brick + nsplit, (it + 1) & 1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    // =============================================================== computing waves
    // NTN = 2: wave w owns the (tap, N tile) pairs T = first .. first + cnt - 1 (T = 2 tap + nt), spanning taps a .. a + 3.
    // NTN = 1: T = tap; waves 0-2 own 4 taps, waves 3-7 own 3 (27 = 3 x 4 + 5 x 3), wave 7's fourth slot sums the bias
    // gradient: 4 + 3 MFMAs per SIMD and K step whichever two computing waves share it.
    const int first = NTN == 2 ? (wave < 7 ? 7 * wave : 48) : (wave < 3 ? 4 * wave : 12 + 3 * (wave - 3));
    const int cnt = NTN == 2 ? (wave < 6 ? 7 : 6) : (wave < 3 ? 4 : 3);
    const int a0 = NTN == 2 ? first >> 1 : first;
    const bool odd = NTN == 2 && (first & 1) != 0;  // slot 0 is N tile 1 of tap a; else slots (0, 1) are tap a

    // fragment lane geometry (tdx_conv3_wgrad_mfma.hip): a K step is 16 voxels; lane group g of 16 lanes reads voxel rows
    // 8 kh + q and + 4, columns 16 (g & 1) + 4 p .. + 3
    const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
    const int col_off = (16 * (g & 1) + 4 * p) * 2;
    const int kh = g >> 1;
    int a_off[4];  // byte offset of this lane's x fragment at K step 0 for the wave's four taps
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int tap = min(a0 + t, 26);
        const int ex = tap / 9 - 1, ey = (tap / 3) % 3 - 1, ez = tap % 3 - 1;
        a_off[t] = ((WR_HY + kh + 1) * WR_HZ + (q + 1) + (ex * WR_HY + ey) * WR_HZ + ez) * 64 + col_off;
    }
    const int b_row = (8 * kh + q) * 64 + col_off;

    f32x16 acc[SLOTS];
#pragma unroll
    for (int i = 0; i < SLOTS; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    // the bias gradient comes from a slot a wave does not need for a tap: it multiplies ones with a dy N tile
    const bool bias_slot = NTN == 2 ? cnt == 6 : wave == 7;   // NTN = 2: waves 6, 7 (N tile wave - 6); NTN = 1: wave 7
    const bf16x8 ones = H16<HF>::ones();

    // The brick loop, specialised at compile time (NTN = 2: on the parity of the wave's first pair = which slots share an x
    // fragment; NTN = 1: on whether the wave's fourth slot is in use).  The K-step loop stays ROLLED, two steps per trip (dy
    // fragment sets by step parity): fully unrolled, the scheduler hoists the fragment reads far ahead and spills hundreds
    // of registers.
    auto run = [&](auto odd_c) {
        constexpr bool ODD = decltype(odd_c)::value;  // NTN = 1: "the fourth slot is used"
        int it = 0;
        for (int brick = split; brick < nbricks; brick += nsplit, ++it) {
            wr_barrier();  // brick `it` is in LDS (the loaders waited for it), everybody is done with brick it - 1
            const unsigned char* bX = sX + (it & 1) * WR_XBUF;
            const unsigned char* bG = sG + (it & 1) * GBUF + b_row;
            auto step_off = [&](int s) { return ((s >> 2) * WR_HY + 2 * (s & 3)) * WR_HZ * 64; };
            auto read_a = [&](int s, int t) {
                const unsigned char* ap = bX + a_off[t] + step_off(s);
                return wr_tr_frag(ap, ap + 4 * 64);
            };
            auto read_b = [&](int s, int nt) {
                const unsigned char* bp = bG + nt * WR_GPLANE + s * (16 * 64);
                return wr_tr_frag(bp, bp + 4 * 64);
            };
            bf16x8 A[4], Bq[2][NTN];
#pragma unroll
            for (int t = 0; t < 4; ++t) A[t] = read_a(0, t);
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt) Bq[0][nt] = read_b(0, nt);

            // one K step: the x fragment of tap t for the next step is read right behind the last MFMA that uses the
            // current one; the dy fragments of the next step go to the other parity set.  (Past the last step the "next"
            // reads fetch step 15 again: valid LDS, never used.)
            auto step = [&](int s, int cur) {
                const int sn = min(s + 1, WR_NSTEPS - 1), nx = cur ^ 1;
#pragma unroll
                for (int nt = 0; nt < NTN; ++nt) Bq[nx][nt] = read_b(sn, nt);
                __builtin_amdgcn_sched_group_barrier(0x100, 2 * NTN, 0);
                if constexpr (NTN == 2) {
                    // slot 6's operands: the wave's last pair, or the bias column sums (waves 6, 7)
                    const bf16x8 a6 = bias_slot ? ones : A[3];
                    const bf16x8 b6 = bias_slot ? (wave == 7 ? Bq[cur][1] : Bq[cur][0]) : (ODD ? Bq[cur][1] : Bq[cur][0]);
                    if (!ODD) {
                        acc[0] = H16<HF>::mfma(A[0], Bq[cur][0], acc[0]);
                        acc[1] = H16<HF>::mfma(A[0], Bq[cur][1], acc[1]);
                        A[0] = read_a(sn, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                        acc[2] = H16<HF>::mfma(A[1], Bq[cur][0], acc[2]);
                        acc[3] = H16<HF>::mfma(A[1], Bq[cur][1], acc[3]);
                        A[1] = read_a(sn, 1);
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                        acc[4] = H16<HF>::mfma(A[2], Bq[cur][0], acc[4]);
                        acc[5] = H16<HF>::mfma(A[2], Bq[cur][1], acc[5]);
                        A[2] = read_a(sn, 2);
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                        acc[6] = H16<HF>::mfma(a6, b6, acc[6]);
                        A[3] = read_a(sn, 3);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    } else {
                        acc[0] = H16<HF>::mfma(A[0], Bq[cur][1], acc[0]);
                        A[0] = read_a(sn, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                        acc[1] = H16<HF>::mfma(A[1], Bq[cur][0], acc[1]);
                        acc[2] = H16<HF>::mfma(A[1], Bq[cur][1], acc[2]);
                        A[1] = read_a(sn, 1);
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                        acc[3] = H16<HF>::mfma(A[2], Bq[cur][0], acc[3]);
                        acc[4] = H16<HF>::mfma(A[2], Bq[cur][1], acc[4]);
                        A[2] = read_a(sn, 2);
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                        acc[5] = H16<HF>::mfma(A[3], Bq[cur][0], acc[5]);
                        acc[6] = H16<HF>::mfma(a6, b6, acc[6]);
                        A[3] = read_a(sn, 3);
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    }
                } else {
                    // one slot per tap: every x fragment feeds one MFMA and is re-read for the next step right behind it
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        acc[t] = H16<HF>::mfma(A[t], Bq[cur][0], acc[t]);
                        A[t] = read_a(sn, t);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    }
                    if (ODD) {  // the fourth tap (waves 0-2) or the bias column sums (wave 7)
                        acc[3] = H16<HF>::mfma(bias_slot ? ones : A[3], Bq[cur][0], acc[3]);
                        A[3] = read_a(sn, 3);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    }
                }
            };
#pragma unroll 1
            for (int s2 = 0; s2 < WR_NSTEPS / 2; ++s2) {
                step(2 * s2, 0);
                step(2 * s2 + 1, 1);
            }
        }
    };
    if (NTN == 2 ? odd : (cnt == 4 || bias_slot)) run(std::true_type{}); else run(std::false_type{});

    // ---- merge: D[row = ci][col = co]; lane holds col (lane & 31), rows (i & 3) + 8 (i >> 2) + 4 (lane >> 5)
    const int r = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
        const int T = first + i;
        if (i < cnt) {
            const int ltap = T / NTN, nt = T % NTN;  // tap in local axes -> tap of the weight tensor
            const int tap = (ltap / 9) * gv.ws[0] + ((ltap / 3) % 3) * gv.ws[1] + (ltap % 3) * gv.ws[2];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ci = ci0 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                if (ci >= Cin) continue;  // half-filled last tile
                float* dst = &dwp[((int64_t)tap * Cin + ci) * Cout + co0 + nt * 32 + r];
                if (slab_stride) dst[(int64_t)split * slab_stride] = acc[i][e];
                else atomicAdd(dst, acc[i][e]);
            }
        } else if (bias_slot && i == SLOTS - 1 && dbias != nullptr && ci0 == 0 && hh == 0) {
            // the all-ones slot: every row of the tile is the column sum of a dy N tile (NTN = 2: tile wave - 6)
            atomicAdd(&dbias[co0 + (NTN == 2 ? (wave - 6) * 32 : 0) + r], acc[i][0]);
        }
    }
}

bool conv3_wgrad_ring_supported(int C1, int C2, int Cout) {
    const char* env = getenv("TDX_WGRAD_RING");  // A/B switch, read per call: 0 = off
    if (env && atoi(env) == 0) return false;
    return conv3_wgrad_mfma_supported(C1, C2, Cout) && (Cout % 32) == 0 && tdx_scratch_ptr() != nullptr && tdx_scratch_bytes() >= 16;
}

// same contract as conv3_wgrad_mfma_launch; TDX_ESHAPE = not a case for this kernel
int conv3_wgrad_ring_launch(const void* x1, int C1, const void* x2, int C2, const void* dy, float* dwp, float* dbias, int B, int X,
                            int Y, int Z, int Cout, hipStream_t st, float* slabs, int max_slabs, int* nslab_out, bool hf) {
    if (!conv3_wgrad_ring_supported(C1, C2, Cout)) return TDX_ESHAPE;
    const int Cin = C1 + C2;
    // local axes: brick 4 x 8 x 8; the short axis goes where it leaves the fewest bricks
    const int E[3] = {X, Y, Z}, gs[3] = {Y * Z, Z, 1}, gw[3] = {9, 3, 1};
    const int cand[3][3] = {{0, 1, 2}, {1, 0, 2}, {2, 0, 1}};
    int best = 0;
    int64_t best_n = -1;
    for (int c = 0; c < 3; ++c) {
        const int64_t n = (int64_t)ceil_div(E[cand[c][0]], WR_BX) * ceil_div(E[cand[c][1]], WR_BY) * ceil_div(E[cand[c][2]], WR_BZ);
        if (best_n < 0 || n < best_n) { best_n = n; best = c; }
    }
    WgradRingView g;
    g.B = B; g.batch = X * Y * Z;
    const int bdim[3] = {WR_BX, WR_BY, WR_BZ};
    for (int k = 0; k < 3; ++k) {
        const int a = cand[best][k];
        g.E[k] = E[a]; g.s[k] = gs[a]; g.ws[k] = gw[a]; g.nb[k] = ceil_div(E[a], bdim[k]);
    }
    const int nbricks = B * g.nb[0] * g.nb[1] * g.nb[2];
    const int NTN = (Cout % 64) == 0 ? 2 : 1;
    {
        const char* env = getenv("TDX_WGRAD_RING");  // A/B switch: 2 = 64-wide tiles only (round 4's rule)
        if (NTN == 1 && env && atoi(env) == 2) return TDX_ESHAPE;
    }
    const int n_ci = (Cin + 31) / 32, n_co = Cout / (32 * NTN);
    const int ntiles = n_ci * n_co;
    const int cus = tdx_persistent_cus();
    int nsplit = cus >= 256 ? (256 + ntiles - 1) / ntiles : std::max(cus / ntiles, 1);  // one workgroup per CU (at most `cus`)
    if (nsplit > nbricks) nsplit = nbricks;
    if (nsplit < 1) nsplit = 1;
    // a workgroup should walk several bricks, or the double buffering has nothing to overlap
    if (nbricks < 4 * nsplit) return TDX_ESHAPE;
    const size_t lds = (size_t)2 * WR_XBUF + (size_t)2 * WR_GBUF(NTN);
    // TDX_DETERMINISTIC: never the atomic merge -- hold the K splits to the slabs the workspace has (added in order by the unpack kernel)
    if (tdx_deterministic() && slabs != nullptr && nsplit > max_slabs) nsplit = max_slabs > 0 ? max_slabs : 1;
    const bool use_slabs = slabs != nullptr && nsplit <= max_slabs;
    const int64_t slab_stride = use_slabs ? (int64_t)27 * Cin * Cout : 0;
    float* out = use_slabs ? slabs : dwp;
    if (nslab_out) *nslab_out = use_slabs ? nsplit : 0;
    static bool attr_set[2][2] = {{false, false}, {false, false}};
#define WR_GO(NTNV, HFV)                                                                                                          \
    do {                                                                                                                          \
        auto kern = conv3_wgrad_ring_kernel<NTNV, HFV>;                                                                           \
        if (!attr_set[NTNV - 1][HFV]) {                                                                                           \
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);          \
            if (e != hipSuccess) return (int)e;                                                                                   \
            attr_set[NTNV - 1][HFV] = true;                                                                                       \
        }                                                                                                                         \
        hipLaunchKernelGGL(kern, dim3((unsigned)(ntiles * nsplit)), dim3(768), lds, st, (const bf16*)x1, C1, (const bf16*)x2, C2, \
                           (const bf16*)dy, out, dbias, g, Cout, nsplit, n_ci, slab_stride, tdx_scratch_ptr());                   \
    } while (0)
    if (NTN == 2) { if (hf) WR_GO(2, true); else WR_GO(2, false); }
    else { if (hf) WR_GO(1, true); else WR_GO(1, false); }
#undef WR_GO
    return tdx_launch_status();
}
