// bf16 MFMA weight gradient of the replicate-padded 3x3x3 convolution on SMALL grids (the deep U-Net levels:
// 24 x 8 x 6 and 12 x 4 x 3 voxels, 256-1024 channels).  gfx950.
//
//   dW[tap][ci][co] = sum_{b, v} x[b, clamp(v + tap)][ci] * dy[b, v][co]
//
// The brick kernel (tdx_conv3_wgrad_mfma.hip) reduces over 4 x 8 x 8 bricks: a 12 x 4 x 3 sample fills 28 % of the two
// bricks it needs, so 72 % of the K steps multiply zeros, and every brick costs a global -> register -> LDS round trip
// that one wave per SIMD cannot hide.  Here K is packed: a "row group" is up to 256 voxels of one or more whole samples
// (or an x slab of one), dense, in steps of 16; the x operand lives in LDS as a clamp-filled image with a one-voxel
// rim, so tap t of voxel v is the image entry of v plus a per-tap constant, and a small LDS table maps a K row to its
// image entry (any voxel can sit in any K row).  Staging is LDS-DMA (global_load_lds_dwordx4), double-buffered: group
// i + 1 lands while the MFMAs of group i issue.  Everything else is the brick kernel's scheme: one workgroup owns a
// 32 (ci) x 32 NT (co) tile of all 27 taps, wave w the taps w, w + 4, ...; transposed fragment reads
// (ds_read_b64_tr_b16) of voxel-major rows; partial tiles go to per-split slabs (summed by the unpack kernel) or are
// merged with f32 atomics.
#include "tdx_common.h"
#include "tdx_conv3.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

#define WS_MAXROWS 256        // voxels per row group (16 K steps)
#define WS_TAPS_PER_WAVE 7

struct WsGeom {
    int B, E[3];
    int nbg;        // samples per row group (1 when a sample is cut into x slabs)
    int xs, gx;     // x planes per group, x slabs per sample
    int Ix, Iy, Iz; // LDS image per sample of a group: (xs + 2)(E1 + 2)(E2 + 2) entries
    int ngroups;
    int xbytes;     // bytes of one x image buffer (a full group, whole 1-KiB DMA pieces)
    int grows;      // dy rows per buffer (multiple of 16)
};

__device__ __forceinline__ bf16x8 ws_tr_frag(const unsigned char* lo, const unsigned char* hi) {
    s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lo));
    s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(hi));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 r = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, r);
}

// One LDS-DMA instruction (global_load_lds_dwordx4: every lane copies 16 B from its own global address to LDS byte
// address `lds` + 16 lane), issued as inline assembly: the compiler treats the builtin form as a store to LDS that any
// later ds_read might alias and puts s_waitcnt vmcnt(0) in front of the NEXT fragment read -- which serialises the
// copy of group i + 1 with the MFMAs of group i, the opposite of double buffering.  The kernel orders the copies
// itself (s_waitcnt vmcnt(0) + barrier before a buffer is read); it issues no other vector-memory loads.
__device__ __forceinline__ void ws_dma16(const void* gsrc, const unsigned char* lds) {
    const unsigned a = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) const void*)lds);
    unsigned keep;  // M0 is compiler-reserved: saved and restored inside the statement (no "m0" clobber)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(a) : "memory");
}

// HF: operand format (H16<HF>: bf16 or fp16 words behind the bf16-typed pointers)
template <int NT, bool HF>
__global__ void __launch_bounds__(256, 1)
conv3_wgrad_small_kernel(const bf16* __restrict__ x1, int C1, const bf16* __restrict__ x2, int C2, const bf16* __restrict__ dy,
                         float* __restrict__ dwp, float* __restrict__ dbias, WsGeom g, int Cout, int nsplit, int n_ci_tiles,
                         int64_t slab_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int gbytes = NT * g.grows * 64;                 // one dy buffer: [NT planes][rows][32 ch]
    unsigned char* sX = smem;                              // [2][xbytes]: [entry][32 ch] of 64-B rows
    unsigned char* sG = smem + 2 * g.xbytes;               // [2][gbytes]
    unsigned* tab = reinterpret_cast<unsigned*>(sG + 2 * gbytes);  // [WS_MAXROWS] K row -> byte offset of its image entry
    unsigned* tabx = tab + WS_MAXROWS;                     // [xbytes / 64] image entry -> (ix | sample << 10 | clamped yz << 20)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Cin = C1 + C2;
    const int tile = blockIdx.x / nsplit, split = blockIdx.x - tile * nsplit;
    const int ci0 = (tile % n_ci_tiles) * 32, co0 = (tile / n_ci_tiles) * (32 * NT);
    const bf16* xs_;
    int Cs, cbase;
    if (ci0 < C1) { xs_ = x1; Cs = C1; cbase = ci0; } else { xs_ = x2; Cs = C2; cbase = ci0 - C1; }

    const int plane = g.E[1] * g.E[2], V = g.E[0] * plane;
    const int img = g.Ix * g.Iy * g.Iz;
    const int per_sample = g.xs * plane;                   // rows of one sample in a full group

    // ---- K row -> image entry (the same for every group: a ragged last group only has fewer valid rows)
    for (int r = tid; r < WS_MAXROWS; r += 256) {
        int ent = g.Iy * g.Iz + g.Iz + 1;                  // rows beyond the group: any staged entry (their dy rows are zero)
        if (r < g.nbg * per_sample) {
            const int bl = r / per_sample, rem = r - bl * per_sample;
            const int lx = rem / plane, rem2 = rem - lx * plane;
            const int ly = rem2 / g.E[2], lz = rem2 - ly * g.E[2];
            ent = ((bl * g.Ix + lx + 1) * g.Iy + ly + 1) * g.Iz + lz + 1;
        }
        tab[r] = (unsigned)ent * 64u;
    }

    // image entry -> where its source voxel is, in a form that is the same for every group (the per-group part is the
    // sample base and the x origin): keeps the integer divisions out of the per-group staging code
    for (int e = tid; e < (g.xbytes >> 6); e += 256) {
        const int bl = e / img, rem = e - bl * img;
        const int ix = rem / (g.Iy * g.Iz), rem2 = rem - ix * (g.Iy * g.Iz);
        const int iy = rem2 / g.Iz, iz = rem2 - iy * g.Iz;
        const int s1 = min(max(iy - 1, 0), g.E[1] - 1), s2 = min(max(iz - 1, 0), g.E[2] - 1);
        tabx[e] = (unsigned)ix | ((unsigned)bl << 10) | ((unsigned)(s1 * g.E[2] + s2) << 20);
    }
    __syncthreads();

    // ---- fragment lane geometry (as tdx_conv3_wgrad_mfma.hip): a K step is 16 rows; lane group g4 of 16 lanes reads
    // rows 8 kh + q and + 4, columns 16 (g4 & 1) + 4 p .. + 3
    const int g4 = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
    const int col_off = (16 * (g4 & 1) + 4 * p) * 2;
    const int kh = g4 >> 1;

    f32x16 acc[WS_TAPS_PER_WAVE][NT];
#pragma unroll
    for (int t = 0; t < WS_TAPS_PER_WAVE; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][nt][i] = 0.f;
    int toff[WS_TAPS_PER_WAVE];  // byte offset of this wave's taps inside the image
#pragma unroll
    for (int t = 0; t < WS_TAPS_PER_WAVE; ++t) {
        const int tap = min(wave + 4 * t, 26);
        const int ex = tap / 9 - 1, ey = (tap / 3) % 3 - 1, ez = tap % 3 - 1;
        toff[t] = ((ex * g.Iy + ey) * g.Iz + ez) * 64 + col_off;
    }
    // Bias gradient = column sums of dy: wave 3 owns only 6 taps (3, 7, ..., 23), so its seventh accumulator slot
    // multiplies an all-ones A fragment instead of a duplicate tap: every row of that tile is sum_v dy[v][co].
    const bool do_bias = dbias != nullptr && ci0 == 0;
    const bool ones_slot = wave == 3;
    const bf16x8 ones = H16<HF>::ones();

    // ---- staging of one row group by LDS-DMA: a piece = 64 lanes x 16 B = 16 rows of 64 B
    auto group_shape = [&](int grp, int& b0, int& x0, int& nb, int& nx) {
        b0 = (grp / g.gx) * g.nbg; x0 = (grp % g.gx) * g.xs;
        nb = min(g.nbg, g.B - b0); nx = min(g.xs, g.E[0] - x0);
    };
    auto stage = [&](int grp, int buf) {
        int b0, x0, nb, nx;
        group_shape(grp, b0, x0, nb, nx);
        unsigned char* dX = sX + buf * g.xbytes;
        const int npx = g.xbytes >> 10;
        for (int pc = wave; pc < npx; pc += 4) {
            const int e = pc * 16 + (lane >> 2), q4 = lane & 3;
            // every entry of the buffer is written (fragment reads of padded K rows and of the rim must find finite
            // values): entries beyond this group's samples repeat the pattern of its last sample
            const unsigned info = tabx[e];
            const int ix = info & 1023, bl = min((int)((info >> 10) & 1023), nb - 1), yz = info >> 20;
            const int s0 = min(max(x0 + ix - 1, 0), g.E[0] - 1);
            const int64_t vox = (int64_t)(b0 + bl) * V + s0 * plane + yz;
            const bf16* src = xs_ + vox * Cs + cbase + q4 * 8;
            ws_dma16(src, dX + pc * 1024);
        }
        unsigned char* dG = sG + buf * gbytes;
        const int nrows = nb * nx * plane, npg = (g.grows >> 4) * NT;
        for (int pc = wave; pc < npg; pc += 4) {
            const int pl = pc / (g.grows >> 4), rp = pc - pl * (g.grows >> 4);
            const int row = rp * 16 + (lane >> 2), q4 = lane & 3;
            unsigned char* dst = dG + pl * g.grows * 64 + rp * 1024;
            if (row < nrows) {
                // the rows of a group are contiguous voxels: whole samples b0 .., or planes x0 .. of sample b0
                const int64_t vox = (int64_t)b0 * V + (int64_t)x0 * plane + row;
                const bf16* src = dy + vox * Cout + co0 + pl * 32 + q4 * 8;
                ws_dma16(src, dst);
            } else {
                *reinterpret_cast<uint4*>(dst + lane * 16) = make_uint4(0, 0, 0, 0);  // K padding: zero dy rows
            }
        }
    };

    int grp = split, buf = 0;
    if (grp < g.ngroups) stage(grp, 0);
    for (; grp < g.ngroups; grp += nsplit, buf ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // this group has landed; the other buffer's fragment reads are done
        if (grp + nsplit < g.ngroups) stage(grp + nsplit, buf ^ 1);
        int b0, x0, nb, nx;
        group_shape(grp, b0, x0, nb, nx);
        // a sample-slab of a ragged last slab has nx < xs planes: its rows are the first nx * plane of the table's
        // order only when nbg == 1 (slab mode); in sample mode every group has whole samples (nx == xs)
        const int nsteps = (nb * nx * plane + 15) >> 4;
        const unsigned char* X = sX + buf * g.xbytes;
        const unsigned char* G = sG + buf * gbytes + col_off;
        auto read_b = [&](int s, bf16x8 (&bf)[NT]) {
            const unsigned char* bp = G + (16 * s + 8 * kh + q) * 64;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bf[nt] = ws_tr_frag(bp + nt * g.grows * 64, bp + nt * g.grows * 64 + 4 * 64);
        };
        // fragments of step s + 1 are read while the MFMAs of step s issue (two register sets); the table entries of
        // a step are fetched one step before its fragments, so no fragment read waits for its address
        bf16x8 A0[WS_TAPS_PER_WAVE], A1[WS_TAPS_PER_WAVE], B0[NT], B1[NT];
        const int krow = 8 * kh + q;
        unsigned e1 = tab[krow], e2 = tab[krow + 4];
        auto read_a = [&](int s_next, bf16x8 (&A)[WS_TAPS_PER_WAVE]) {
#pragma unroll
            for (int t = 0; t < WS_TAPS_PER_WAVE - 1; ++t) A[t] = ws_tr_frag(X + e1 + toff[t], X + e2 + toff[t]);
            A[WS_TAPS_PER_WAVE - 1] = ones_slot ? ones : ws_tr_frag(X + e1 + toff[WS_TAPS_PER_WAVE - 1], X + e2 + toff[WS_TAPS_PER_WAVE - 1]);
            e1 = tab[16 * s_next + krow];
            e2 = tab[16 * s_next + krow + 4];
        };
        auto mfma_step = [&](const bf16x8 (&A)[WS_TAPS_PER_WAVE], const bf16x8 (&Bf)[NT]) {
#pragma unroll
            for (int t = 0; t < WS_TAPS_PER_WAVE; ++t)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[t][nt] = H16<HF>::mfma(A[t], Bf[nt], acc[t][nt]);
        };
        constexpr int NRD = 2 * WS_TAPS_PER_WAVE + 2 + 2 * NT, NMF = WS_TAPS_PER_WAVE * NT;
        read_a(min(1, nsteps - 1), A0);
        read_b(0, B0);
#pragma unroll 1
        for (int s = 0; s < nsteps; s += 2) {
            const int s1 = min(s + 1, nsteps - 1), s2 = min(s + 2, nsteps - 1), s3 = min(s + 3, nsteps - 1);
            read_a(s2, A1);  // fragments of step s1 (its entries were fetched a step ago); then the entries of s2
            read_b(s1, B1);
            mfma_step(A0, B0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, NMF - 1, 0);
            if (s + 1 < nsteps) {
                read_a(s3, A0);
                read_b(s2, B0);
                mfma_step(A1, B1);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, NMF - 1, 0);
            }
        }
    }

    // ---- merge: D[row = ci][col = co]; lane holds col (lane & 31), rows (i & 3) + 8 (i >> 2) + 4 (lane >> 5)
    const int r = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int t = 0; t < WS_TAPS_PER_WAVE; ++t) {
        const int tap = wave + 4 * t;
        if (tap < 27) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int ci = ci0 + (i & 3) + 8 * (i >> 2) + 4 * hh;
                    float* dst = &dwp[((int64_t)tap * Cin + ci) * Cout + co0 + nt * 32 + r];
                    if (slab_stride) dst[(int64_t)split * slab_stride] = acc[t][nt][i];
                    else atomicAdd(dst, acc[t][nt][i]);
                }
        }
    }
    if (do_bias && ones_slot && hh == 0) {  // row 0 of the all-ones tile
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) atomicAdd(&dbias[co0 + nt * 32 + r], acc[WS_TAPS_PER_WAVE - 1][nt][0]);
    }
}

// Launch if this is a small-grid case; TDX_ESHAPE otherwise (the caller then takes the brick kernel).
int conv3_wgrad_small_launch(const void* x1, int C1, const void* x2, int C2, const void* dy, float* dwp, float* dbias, int B,
                             int X, int Y, int Z, int Cout, hipStream_t st, float* slabs, int max_slabs, int* nslab_out, bool hf) {
    static const int mode = getenv("TDX_WGRAD_SMALL") ? atoi(getenv("TDX_WGRAD_SMALL")) : 1;  // A/B switch
    if (mode == 0) return TDX_ESHAPE;
    const int Cin = C1 + C2;
    if ((C1 % 32) || (C2 % 32) || (Cout % 32) || Cin < 128) return TDX_ESHAPE;
    const char* env_rows = getenv("TDX_WGRAD_SMALL_ROWS");  // row gate (tests, A/B runs; read per call)
    const int64_t rows_total = (int64_t)B * X * Y * Z;
    if (rows_total > (env_rows ? atoi(env_rows) : 8000)) return TDX_ESHAPE;
    WsGeom g;
    g.B = B; g.E[0] = X; g.E[1] = Y; g.E[2] = Z;
    const int plane = Y * Z, V = X * plane;
    if (plane > WS_MAXROWS) return TDX_ESHAPE;
    if (V <= WS_MAXROWS) { g.nbg = std::min(B, WS_MAXROWS / V); g.xs = X; g.gx = 1; }
    else { g.nbg = 1; g.xs = WS_MAXROWS / plane; g.gx = ceil_div(X, g.xs); }
    g.Ix = g.xs + 2; g.Iy = Y + 2; g.Iz = Z + 2;
    const int NT = (Cout % 64 == 0) ? 2 : 1;
    auto sizes = [&]() {
        const int entries = g.nbg * g.Ix * g.Iy * g.Iz;
        g.xbytes = ((entries * 64 + 1023) >> 10) << 10;
        g.grows = ((g.nbg * g.xs * plane + 15) >> 4) << 4;
        return (size_t)2 * g.xbytes + (size_t)2 * NT * g.grows * 64 + (WS_MAXROWS + (g.xbytes >> 6)) * sizeof(unsigned);
    };
    size_t lds = sizes();
    while (lds > 160 * 1024 && (g.nbg > 1 || g.xs > 1)) {  // shrink the group until two buffers fit
        if (g.nbg > 1) --g.nbg; else { --g.xs; g.gx = ceil_div(X, g.xs); g.Ix = g.xs + 2; }
        lds = sizes();
    }
    if (lds > 160 * 1024) return TDX_ESHAPE;
    g.ngroups = ceil_div(B, g.nbg) * g.gx;
    const int n_ci = Cin / 32, n_co = Cout / (32 * NT);
    const int ntiles = n_ci * n_co;
    int nsplit = std::max(1, std::min(g.ngroups, ceil_div(256, ntiles)));
    // TDX_DETERMINISTIC: never the atomic merge -- hold the K splits to the slabs the workspace has (added in order by the unpack kernel)
    if (tdx_deterministic() && slabs != nullptr && nsplit > max_slabs) nsplit = max_slabs > 0 ? max_slabs : 1;
    const bool use_slabs = slabs != nullptr && nsplit <= max_slabs;
    const int64_t slab_stride = use_slabs ? (int64_t)27 * Cin * Cout : 0;
    float* out = use_slabs ? slabs : dwp;
    if (nslab_out) *nslab_out = use_slabs ? nsplit : 0;
    dim3 grid((unsigned)(ntiles * nsplit));
#define WS_LAUNCH(NTV, HFV)                                                                                           \
    do {                                                                                                              \
        auto kern = conv3_wgrad_small_kernel<NTV, HFV>;                                                               \
        static size_t attr = 0;                                                                                       \
        if (lds > attr) {                                                                                             \
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return (int)e;                                                                       \
            attr = lds;                                                                                               \
        }                                                                                                             \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, (const bf16*)x1, C1, (const bf16*)x2, C2, (const bf16*)dy, out, \
                           dbias, g, Cout, nsplit, n_ci, slab_stride);                                                \
    } while (0)
    if (NT == 2) { if (hf) WS_LAUNCH(2, true); else WS_LAUNCH(2, false); }
    else { if (hf) WS_LAUNCH(1, true); else WS_LAUNCH(1, false); }
#undef WS_LAUNCH
    return tdx_launch_status();
}
