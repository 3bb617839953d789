// Brick geometry shared by the fp32-tensor MFMA conv kernels (tdx_conv3_mfma_f32.hip, tdx_conv3_mfma_split.hip).
//
// A workgroup owns one brick of 256 output voxels: 4 x 8 x 8, or the "thin" 2 x 16 x 8 used for the 1-2 voxel
// remainder slabs that a grid leaves next to the exactly tiled main region (194 x 66 x 50: 3087 main bricks
// against 2304 + 215 with thin slabs).  Bricks live in LOCAL axes: local axis k is global axis perm[k], so the
// short brick edge can be laid along whichever grid axis leaves the fewest bricks, and the thin edge along a
// slab's thin axis; weight taps are re-indexed through ws[].  One launch carries up to three regions (the
// three slabs of a grid go out together).  Since the data gradient runs on the ORIGINAL grid (its halo-shell term
// is tdx_conv3_shell.hip) the slabs only occur on grids like the reference's real 194 x 50 x 50.
#pragma once
#include "tdx_common.h"
#include "tdx_conv3.h"

struct BrickView {
    int B;
    int Ei[3], Eo[3];    // global input / output extents
    int perm[3];         // local axis k = global axis perm[k]
    int org[3], ext[3];  // region origin / extent in output coordinates (local axes)
    int nb[3];           // bricks per local axis
    int ws[3];           // weight-tap stride of local axis k: {9, 3, 1}[perm[k]]
    int off;             // output voxel o reads input voxel o + off + e
};

struct BrickRegions {
    BrickView v[3];
    int start[4];  // first block id of region r (start[n] = number of blocks)
    int n;
};

// brick shapes: 0 = 4 x 8 x 8 (two 32-voxel M tiles per wave), 1 = thin 2 x 16 x 8 (remainder slabs),
// 2 = big 8 x 8 x 8 (four M tiles per wave: for 32-channel output tiles, so that a weight fragment still
// feeds four MFMAs and the weights are staged once per 512 voxels)
#define BRICK_MAIN 0
#define BRICK_THIN 1
#define BRICK_BIG 2
template <int SHAPE>
struct Brick {
    static constexpr int BX = SHAPE == BRICK_THIN ? 2 : (SHAPE == BRICK_BIG ? 8 : 4), BY = SHAPE == BRICK_THIN ? 16 : 8, BZ = 8;
    static constexpr int MT = SHAPE == BRICK_BIG ? 4 : 2;  // M tiles (32 voxels) per wave
    static constexpr int NVOX = BX * BY * BZ;
    static constexpr int HX = BX + 2, HY = BY + 2, HZ = BZ + 2;
    static constexpr int SZ = 12;  // padded z stride of the LDS image (conflict-free 16-B fragment reads)
    static constexpr int NHALO = HX * HY * HZ, ENTRIES = HX * HY * SZ;

    // wave w, M tile mt, lane row r -> local voxel inside the brick
    __device__ __forceinline__ static void lane_voxel(int wave, int mt, int r, int& lx, int& ly, int& lz) {
        if (SHAPE == BRICK_THIN) { lx = wave >> 1; ly = 8 * (wave & 1) + 4 * mt + (r & 3); }
        else if (SHAPE == BRICK_BIG) { lx = 2 * wave + (mt >> 1); ly = 4 * (mt & 1) + (r & 3); }
        else { lx = wave; ly = 4 * mt + (r & 3); }
        lz = r >> 2;
    }
    __device__ __forceinline__ static int tile_index(int lx, int ly, int lz) { return (lx * BY + ly) * BZ + lz; }
    __device__ __forceinline__ static void tile_voxel(int v, int& lx, int& ly, int& lz) {
        lz = v % BZ; ly = (v / BZ) % BY; lx = v / (BZ * BY);
    }
};

// block id -> region, sample and brick origin (local output coordinates)
template <int SHAPE>
__device__ __forceinline__ BrickView brick_decode(const BrickRegions& R, int bid, int& b, int (&o)[3]) {
    // the view is selected by value (uniform selects): indexing R.v[] dynamically would push the struct to scratch
    BrickView g = R.v[0];
    int first = 0;
    if (R.n > 1 && bid >= R.start[1]) { g = R.v[1]; first = R.start[1]; }
    if (R.n > 2 && bid >= R.start[2]) { g = R.v[2]; first = R.start[2]; }
    bid -= first;
    const int b2 = bid % g.nb[2]; bid /= g.nb[2];
    const int b1 = bid % g.nb[1]; bid /= g.nb[1];
    const int b0 = bid % g.nb[0]; bid /= g.nb[0];
    b = bid;
    o[0] = g.org[0] + b0 * Brick<SHAPE>::BX; o[1] = g.org[1] + b1 * Brick<SHAPE>::BY; o[2] = g.org[2] + b2 * Brick<SHAPE>::BZ;
    return g;
}

// linear input voxel (inside one sample) read by halo position (hx, hy, hz) of the brick at o, or -1 for a
// zero (ZERO_PAD: outside the grid; otherwise positions are clamped = replicate padding)
template <bool ZERO_PAD, bool PERM>
__device__ __forceinline__ int brick_halo_source(const BrickView& g, const int (&o)[3], int hx, int hy, int hz) {
    const int h[3] = {hx, hy, hz};
    // stride of local axis k in the input = stride of global axis perm[k] (identity when !PERM: compile-time)
    const int gs[3] = {g.Ei[1] * g.Ei[2], g.Ei[2], 1};
    int lin = 0;
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        int s = o[k] + h[k] - 1 + g.off;
        const int ax = PERM ? g.perm[k] : k;
        const int E = PERM ? (ax == 0 ? g.Ei[0] : ax == 1 ? g.Ei[1] : g.Ei[2]) : g.Ei[k];
        const int st = PERM ? (ax == 0 ? gs[0] : ax == 1 ? gs[1] : gs[2]) : gs[k];
        if (ZERO_PAD) ok = ok && s >= 0 && s < E;
        else s = min(max(s, 0), E - 1);
        lin += s * st;
    }
    return ok ? lin : -1;
}

// global output coordinates of local voxel (lx, ly, lz) of the brick at o; false if outside the region
template <bool PERM>
__device__ __forceinline__ bool brick_out_coords(const BrickView& g, const int (&o)[3], int lx, int ly, int lz, int (&c)[3]) {
    const int l[3] = {o[0] + lx, o[1] + ly, o[2] + lz};
    const bool ok = l[0] < g.org[0] + g.ext[0] && l[1] < g.org[1] + g.ext[1] && l[2] < g.org[2] + g.ext[2];
    if (PERM) {
        // c[perm[k]] = l[k] without dynamically indexed registers
#pragma unroll
        for (int a = 0; a < 3; ++a) c[a] = g.perm[0] == a ? l[0] : (g.perm[1] == a ? l[1] : l[2]);
    } else {
        c[0] = l[0]; c[1] = l[1]; c[2] = l[2];
    }
    return ok;
}

// row of the weight image (global tap index) of local tap (ex, ey, ez) in -1..1
__device__ __forceinline__ int brick_tap(const BrickView& g, int ex, int ey, int ez) {
    return (ex + 1) * g.ws[0] + (ey + 1) * g.ws[1] + (ez + 1) * g.ws[2];
}

// ---------------------------------------------------------------------------------------------- host side
static inline void brick_dims(int shape, int bd[3]) {
    bd[0] = shape == BRICK_THIN ? 2 : (shape == BRICK_BIG ? 8 : 4); bd[1] = shape == BRICK_THIN ? 16 : 8; bd[2] = 8;
}

static inline void brick_fill_view(BrickView& v, const Conv3Geom& g, const int perm[3], const int org_g[3], const int ext_g[3],
                                   int shape) {
    static const int tapw[3] = {9, 3, 1};
    int bd[3];
    brick_dims(shape, bd);
    v.B = g.B; v.off = g.off;
    v.Ei[0] = g.Xi; v.Ei[1] = g.Yi; v.Ei[2] = g.Zi;
    v.Eo[0] = g.Xo; v.Eo[1] = g.Yo; v.Eo[2] = g.Zo;
    for (int k = 0; k < 3; ++k) {
        v.perm[k] = perm[k];
        v.org[k] = org_g[perm[k]];
        v.ext[k] = ext_g[perm[k]];
        v.nb[k] = ceil_div(v.ext[k], bd[k]);
        v.ws[k] = tapw[perm[k]];
    }
}

static inline int64_t brick_count(const int ext_g[3], const int perm[3], int shape) {
    int bd[3];
    brick_dims(shape, bd);
    int64_t n = 1;
    for (int k = 0; k < 3; ++k) n *= ceil_div(ext_g[perm[k]], bd[k]);
    return n;
}

// Regions of one conv call: `main` (4 x 8 x 8 bricks, one region) and `thin` (2 x 16 x 8 bricks, 0-3 slab
// regions).  Zero-padded data gradient on big grids: remainders of 1-2 voxels along an axis become thin slabs.
// Otherwise the whole grid with the brick orientation that leaves the fewest bricks.
static inline void brick_plan(const Conv3Geom& g, bool zero_pad, bool allow_thin, BrickRegions& main, BrickRegions& thin,
                              int main_shape = BRICK_MAIN) {
    static const int cand[3][3] = {{0, 1, 2}, {1, 0, 2}, {2, 0, 1}};  // which global axis gets the short (4 / 2) edge
    const int Eo[3] = {g.Xo, g.Yo, g.Zo};
    int bd[3];
    brick_dims(main_shape, bd);
    int Em[3] = {Eo[0], Eo[1], Eo[2]};  // extent of the main region
    bool slab[3] = {false, false, false};
    const int org0[3] = {0, 0, 0};
    thin.n = 0;
    int mperm = 0;
    // thin slabs pay on big grids only (an extra, mostly empty launch costs more than ragged bricks on the
    // deep U-Net levels): at least 1024 main bricks
    const int64_t full = (int64_t)g.B * ceil_div(Eo[0], 4) * ceil_div(Eo[1], 8) * ceil_div(Eo[2], 8);  // in 256-voxel bricks
    if (zero_pad && allow_thin && full >= 1024) {
        for (int a = 0; a < 3; ++a) {
            const int rem = Eo[a] % bd[a];
            if ((rem == 1 || rem == 2) && Eo[a] > bd[a]) { slab[a] = true; Em[a] = Eo[a] - rem; }
        }
    }
    if (!slab[0] && !slab[1] && !slab[2]) {
        int64_t best = -1;
        for (int c = 0; c < 3; ++c) {
            const int64_t n = brick_count(Em, cand[c], main_shape);
            if (best < 0 || n < best) { best = n; mperm = c; }
        }
    }
    main.n = 1;
    brick_fill_view(main.v[0], g, cand[mperm], org0, Em, main_shape);
    main.start[0] = 0;
    main.start[1] = (int)((int64_t)g.B * main.v[0].nb[0] * main.v[0].nb[1] * main.v[0].nb[2]);
    int nblk = 0;
    for (int a = 0; a < 3; ++a) {
        if (!slab[a]) continue;
        // slab a: axis a in [Em[a], Eo[a]); axes before a restricted to the main extent, axes after a full
        int org[3], ext[3];
        for (int k = 0; k < 3; ++k) { org[k] = 0; ext[k] = (k < a) ? Em[k] : Eo[k]; }
        org[a] = Em[a]; ext[a] = Eo[a] - Em[a];
        const int p = (a + 1) % 3, q = (a + 2) % 3;
        const int perm1[3] = {a, p, q}, perm2[3] = {a, q, p};
        const int* perm = brick_count(ext, perm1, BRICK_THIN) <= brick_count(ext, perm2, BRICK_THIN) ? perm1 : perm2;
        BrickView& v = thin.v[thin.n];
        brick_fill_view(v, g, perm, org, ext, BRICK_THIN);
        thin.start[thin.n] = nblk;
        nblk += (int)((int64_t)g.B * v.nb[0] * v.nb[1] * v.nb[2]);
        ++thin.n;
    }
    thin.start[thin.n] = nblk;
}
