// bf16 MFMA weight gradient of the replicate-padded 3x3x3 convolution, LDS-DMA double-buffered form (gfx950).
//
//   dW[tap][ci][co] = sum_v x[clamp(v + tap)][ci] * dy[v][co]
//
// Same GEMM cut as tdx_conv3_wgrad_mfma.hip -- one 4-wave workgroup per CU (one wave per SIMD, 224 accumulator
// registers) owns a 32 (ci) x 64 (co) tile of all 27 taps, wave w the taps w, w + 4, ..., and walks a strided subset of
// 4 x 8 x 8-voxel bricks; fragments are transposed LDS reads (ds_read_b64_tr_b16) of voxel-major 64-B rows -- but the
// bricks are staged by LDS-DMA (global_load_lds_dwordx4) into TWO buffer pairs: brick i + 1 lands while the 224 MFMAs
// per wave of brick i issue, one DMA instruction every dozen MFMAs.  The brick kernel stages global -> VGPR -> LDS: with
// one wave per SIMD its 32 ds_write_b128 per thread and brick (96 KB at ~80 B/clk), the wait for the loads and the
// second barrier are all exposed (MFMA pipe busy 45 %, profiles/r09bf16_summary.md); here a brick costs one
// s_waitcnt vmcnt(0) + one s_barrier.
//
// The bias gradient comes from the matrix pipe, as in tdx_conv3_wgrad_small.hip: wave 3 owns only 6 taps, its seventh
// accumulator slot multiplies an all-ones A fragment, so every row of that tile is sum_v dy[v][co].
//
// Zero rows (dy rows of voxels outside a ragged brick; the missing channels of a half-filled last ci tile) are copied
// from the zero block at the head of the scratch arena (tdx_set_scratch).  64-wide output tiles only (Cout % 64 == 0);
// everything else stays on tdx_conv3_wgrad_mfma.hip.
#include "tdx_common.h"
#include "tdx_conv3.h"
#include <stdlib.h>

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
#define WR_XPW ((WR_XPIECES + 3) / 4)                  // 10 per wave
#define WR_GPLANE (WR_NVOX * 64)                       // one 32-channel dy plane: 16 pieces
#define WR_GBUF (2 * WR_GPLANE)
#define WR_GPW 8                                       // dy pieces per wave
#define WR_TAPS 7

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
    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds) : "memory", "m0");
}

__global__ void __launch_bounds__(256, 1)
conv3_wgrad_ring_kernel(const bf16* __restrict__ x1, int C1, const bf16* __restrict__ x2, int C2, const bf16* __restrict__ dy,
                        float* __restrict__ dwp, float* __restrict__ dbias, WgradRingView gv, int Cout, int nsplit, int n_ci_tiles,
                        int64_t slab_stride, const void* __restrict__ zeros) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char* sX = smem;                        // [2][WR_XBUF]
    unsigned char* sG = smem + 2 * WR_XBUF;           // [2][WR_GBUF]
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Cin = C1 + C2;
    const int tile = blockIdx.x / nsplit, split = blockIdx.x - tile * nsplit;
    const int ci0 = (tile % n_ci_tiles) * 32, co0 = (tile / n_ci_tiles) * 64;
    const bf16* xs;
    int Cs, cbase;
    if (ci0 < C1) { xs = x1; Cs = C1; cbase = ci0; } else { xs = x2; Cs = C2; cbase = ci0 - C1; }
    const int nbricks = gv.B * gv.nb[0] * gv.nb[1] * gv.nb[2];

    // ---- fragment lane geometry (tdx_conv3_wgrad_mfma.hip): a K step is 16 voxels; lane group g of 16 lanes reads
    // voxel rows 8 kh + q and + 4, columns 16 (g & 1) + 4 p .. + 3
    const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
    const int col_off = (16 * (g & 1) + 4 * p) * 2;
    const int kh = g >> 1;

    f32x16 acc[WR_TAPS][2];
#pragma unroll
    for (int t = 0; t < WR_TAPS; ++t)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][nt][i] = 0.f;
    int a_off[WR_TAPS];  // byte offset of this lane's fragment at K step 0 for each of the wave's taps
#pragma unroll
    for (int t = 0; t < WR_TAPS; ++t) {
        const int tap = min(wave + 4 * t, 26);
        const int ex = tap / 9 - 1, ey = (tap / 3) % 3 - 1, ez = tap % 3 - 1;
        a_off[t] = ((WR_HY + kh + 1) * WR_HZ + (q + 1) + (ex * WR_HY + ey) * WR_HZ + ez) * 64 + col_off;
    }
    const bool do_bias = dbias != nullptr && ci0 == 0;
    const bool ones_slot = wave == 3;  // taps 3, 7, ..., 23: slot 6 is free
    bf16x8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (__bf16)1.0f;

    // ---- per-lane DMA geometry.  x piece i of this wave: chunks pc = 64 (wave XPW + i) + lane = (halo voxel, 16-B
    // quarter of its 64-B row); dy piece j: plane (wave XPW.. see below), chunks (voxel, quarter)
    int xh[WR_XPW];      // hx | hy << 8 | hz << 16 | quarter << 24 | valid << 31 (valid: the channels exist)
#pragma unroll
    for (int i = 0; i < WR_XPW; ++i) {
        // the 2 slots beyond the image re-copy its last piece (same bytes to the same place)
        const int pc = min(min(wave * WR_XPW + i, WR_XPIECES - 1) * 64 + lane, WR_NHALO * 4 - 1);
        const int hv = pc >> 2, q4 = pc & 3;
        const int hx = hv / (WR_HY * WR_HZ), rem = hv - hx * (WR_HY * WR_HZ);
        const int hy = rem / WR_HZ, hz = rem - hy * WR_HZ;
        xh[i] = hx | (hy << 8) | (hz << 16) | (q4 << 24) | ((cbase + q4 * 8 < Cs) ? (1 << 30) : 0);
    }
    // dy piece j of this wave: gp = wave * 8 + j -> plane gp / 16, chunks e = (gp % 16) * 64 + lane = (voxel e >> 2, quarter e & 3)
    const int g_plane = (wave * WR_GPW) / 16;           // the wave's 8 pieces lie in one plane
    const int g_e0 = ((wave * WR_GPW) % 16) * 64 + lane;  // chunk of piece 0; piece j: + 64 j  (voxel + 16 j)

    auto issue_x = [&](int brick, int buf, int i) {
        int bb = brick;
        const int bz = bb % gv.nb[2]; bb /= gv.nb[2];
        const int by = bb % gv.nb[1]; bb /= gv.nb[1];
        const int bx = bb % gv.nb[0]; bb /= gv.nb[0];
        const int sx = min(max(bx * WR_BX + (xh[i] & 0xff) - 1, 0), gv.E[0] - 1);
        const int sy = min(max(by * WR_BY + ((xh[i] >> 8) & 0xff) - 1, 0), gv.E[1] - 1);
        const int sz = min(max(bz * WR_BZ + ((xh[i] >> 16) & 0xff) - 1, 0), gv.E[2] - 1);
        const int64_t vox = (int64_t)bb * gv.batch + sx * gv.s[0] + sy * gv.s[1] + sz * gv.s[2];
        const bf16* src = (xh[i] >> 30) & 1 ? xs + vox * Cs + cbase + ((xh[i] >> 24) & 3) * 8 : reinterpret_cast<const bf16*>(zeros);
        const int pi = wave * WR_XPW + i;
        wr_dma(src, lds0 + buf * WR_XBUF + min(pi, WR_XPIECES - 1) * 1024);
    };
    auto issue_g = [&](int brick, int buf, int j) {
        int bb = brick;
        const int bz = bb % gv.nb[2]; bb /= gv.nb[2];
        const int by = bb % gv.nb[1]; bb /= gv.nb[1];
        const int bx = bb % gv.nb[0]; bb /= gv.nb[0];
        const int e = g_e0 + 64 * j, v = e >> 2, c4 = e & 3;
        const int vx = bx * WR_BX + (v >> 6), vy = by * WR_BY + ((v >> 3) & 7), vz = bz * WR_BZ + (v & 7);
        const bool ok = vx < gv.E[0] && vy < gv.E[1] && vz < gv.E[2];
        const int64_t vox = (int64_t)bb * gv.batch + vx * gv.s[0] + vy * gv.s[1] + vz * gv.s[2];
        const bf16* src = ok ? dy + vox * Cout + co0 + g_plane * 32 + c4 * 8 : reinterpret_cast<const bf16*>(zeros);
        wr_dma(src, lds0 + 2 * WR_XBUF + buf * WR_GBUF + g_plane * WR_GPLANE + (((wave * WR_GPW) % 16) + j) * 1024);
    };

    int brick = split, it = 0;
    if (brick < nbricks) {
#pragma unroll
        for (int i = 0; i < WR_XPW; ++i) issue_x(brick, 0, i);
#pragma unroll
        for (int j = 0; j < WR_GPW; ++j) issue_g(brick, 0, j);
    }
    for (; brick < nbricks; brick += nsplit, ++it) {
        const int buf = it & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of the brick have landed ...
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                     // ... everybody's have, and the previous brick's reads are done
        asm volatile("" ::: "memory");
        const int next = brick + nsplit;
        const bool more = next < nbricks;
        const unsigned char* bX = sX + buf * WR_XBUF;
        const unsigned char* bG = sG + buf * WR_GBUF;

        // K step s: voxels (x = s >> 2, y = 2 (s & 3) + kh, z = q (+4)).  One wave per SIMD: while step s issues its
        // 7 x 2 MFMAs from one register set, the fragments of step s + 1 are read into the other; the next brick's
        // 18 DMA instructions ride behind the steps, at most two per step
        auto read_b = [&](int s, bf16x8 (&bf)[2]) {
            const unsigned char* bp = bG + (16 * s + 8 * kh + q) * 64 + col_off;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) bf[nt] = wr_tr_frag(bp + nt * WR_GPLANE, bp + nt * WR_GPLANE + 4 * 64);
        };
        auto step_off = [&](int s) { return ((s >> 2) * WR_HY + 2 * (s & 3)) * WR_HZ * 64; };
        auto read_a = [&](int soff, int t) {
            const unsigned char* ap = bX + a_off[t] + soff;
            return wr_tr_frag(ap, ap + 4 * 64);
        };
        bf16x8 A[2][WR_TAPS], Bf[2][2];
#pragma unroll
        for (int t = 0; t < WR_TAPS; ++t) A[0][t] = read_a(0, t);
        read_b(0, Bf[0]);
#pragma unroll
        for (int s = 0; s < WR_NSTEPS; ++s) {
            const int cur = s & 1, nxt = cur ^ 1;
            const int sn = min(s + 1, WR_NSTEPS - 1), off_n = step_off(sn);
#pragma unroll
            for (int t = 0; t < WR_TAPS; ++t) {
                A[nxt][t] = read_a(off_n, t);
                if (t == 0) read_b(sn, Bf[nxt]);
                const bf16x8 af = (t == WR_TAPS - 1 && ones_slot) ? ones : A[cur][t];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[t][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, Bf[cur][nt], acc[t][nt], 0, 0, 0);
                if (t == 0) __builtin_amdgcn_sched_group_barrier(0x100, 2 + 4, 0);
                else __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            }
            if (more) {
                // 10 x pieces behind steps 0 .. 9, 8 dy pieces behind steps 4 .. 11
                if (s < WR_XPW) issue_x(next, buf ^ 1, s);
                if (s >= 4 && s < 4 + WR_GPW) issue_g(next, buf ^ 1, s - 4);
            }
        }
    }

    // ---- merge: D[row = ci][col = co]; lane holds col (lane & 31), rows (i&3) + 8 (i>>2) + 4 (lane>>5)
    const int r = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int t = 0; t < WR_TAPS; ++t) {
        const int ltap = wave + 4 * t;  // tap in local axes -> tap of the weight tensor
        if (ltap < 27) {
            const int tap = (ltap / 9) * gv.ws[0] + ((ltap / 3) % 3) * gv.ws[1] + (ltap % 3) * gv.ws[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int ci = ci0 + (i & 3) + 8 * (i >> 2) + 4 * hh;
                    if (ci >= Cin) continue;  // half-filled last tile
                    float* dst = &dwp[((int64_t)tap * Cin + ci) * Cout + co0 + nt * 32 + r];
                    if (slab_stride) dst[(int64_t)split * slab_stride] = acc[t][nt][i];
                    else atomicAdd(dst, acc[t][nt][i]);
                }
        } else if (do_bias && hh == 0) {
            // the all-ones slot: every row of the tile is the column sum of dy
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) atomicAdd(&dbias[co0 + nt * 32 + r], acc[t][nt][0]);
        }
    }
}

bool conv3_wgrad_ring_supported(int C1, int C2, int Cout) {
    const char* env = getenv("TDX_WGRAD_RING");  // A/B switch, read per call: 0 = off
    if (env && atoi(env) == 0) return false;
    return conv3_wgrad_mfma_supported(C1, C2, Cout) && (Cout % 64) == 0 && tdx_scratch_ptr() != nullptr && tdx_scratch_bytes() >= 16;
}

// same contract as conv3_wgrad_mfma_launch; TDX_ESHAPE = not a case for this kernel
int conv3_wgrad_ring_launch(const void* x1, int C1, const void* x2, int C2, const void* dy, float* dwp, float* dbias, int B, int X,
                            int Y, int Z, int Cout, hipStream_t st, float* slabs, int max_slabs, int* nslab_out) {
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
    const int n_ci = (Cin + 31) / 32, n_co = Cout / 64;
    const int ntiles = n_ci * n_co;
    int nsplit = (256 + ntiles - 1) / ntiles;  // one workgroup per CU
    if (nsplit > nbricks) nsplit = nbricks;
    if (nsplit < 1) nsplit = 1;
    // a workgroup should walk several bricks, or the double buffering has nothing to overlap
    if (nbricks < 4 * nsplit) return TDX_ESHAPE;
    const size_t lds = (size_t)2 * WR_XBUF + (size_t)2 * WR_GBUF;
    const bool use_slabs = slabs != nullptr && nsplit <= max_slabs;
    const int64_t slab_stride = use_slabs ? (int64_t)27 * Cin * Cout : 0;
    float* out = use_slabs ? slabs : dwp;
    if (nslab_out) *nslab_out = use_slabs ? nsplit : 0;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)conv3_wgrad_ring_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(conv3_wgrad_ring_kernel, dim3((unsigned)(ntiles * nsplit)), dim3(256), lds, st, (const bf16*)x1, C1,
                       (const bf16*)x2, C2, (const bf16*)dy, out, dbias, g, Cout, nsplit, n_ci, slab_stride, tdx_scratch_ptr());
    return tdx_launch_status();
}
